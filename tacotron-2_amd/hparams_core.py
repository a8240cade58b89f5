"""A small stand-in for tf.contrib.training.HParams (TensorFlow is not a dependency of this tree).

Supports what the reference's CLIs use (train.py:35, synthesize.py:15, hparams.py:376-379):
attribute access, ``parse("a=1,b=[2,3],c=foo")`` with values typed by the default's type,
``values()``, ``set_hparam``, ``add_hparam``.  Unlike TF's class, keys whose default is None can be
overridden (the value is parsed as int / float / bool / str in that order) -- SURVEY.md App. C-16.
"""
import re


def _parse_scalar(text, proto):
    t = text.strip()
    if isinstance(proto, bool):
        if t in ('True', 'true', '1'):
            return True
        if t in ('False', 'false', '0'):
            return False
        raise ValueError('could not parse %r as bool' % text)
    if isinstance(proto, int):
        return int(t)
    if isinstance(proto, float):
        return float(t)
    if isinstance(proto, str):
        return t.strip('\'"')
    # untyped default (None): best effort
    if t in ('None', 'none'):
        return None
    for conv in (int, float):
        try:
            return conv(t)
        except ValueError:
            pass
    if t in ('True', 'true'):
        return True
    if t in ('False', 'false'):
        return False
    return t.strip('\'"')


_ASSIGN = re.compile(r'\s*([A-Za-z_][A-Za-z0-9_]*)\s*=\s*(\[[^\]]*\]|[^,\[\]]*)\s*(?:,|$)')


class HParams(object):
    def __init__(self, **kwargs):
        object.__setattr__(self, '_keys', [])
        for k, v in kwargs.items():
            self.add_hparam(k, v)

    def add_hparam(self, name, value):
        if name in self._keys:
            raise ValueError('Hyperparameter name is reserved/duplicated: %s' % name)
        self._keys.append(name)
        object.__setattr__(self, name, value)

    def set_hparam(self, name, value):
        if name not in self._keys:
            raise ValueError('Unknown hyperparameter: %s' % name)
        object.__setattr__(self, name, value)

    def values(self):
        return {k: getattr(self, k) for k in self._keys}

    def __contains__(self, name):
        return name in self._keys

    def parse(self, text):
        """Override values from 'name=value,name=[v1,v2],...'. Returns self."""
        if not text:
            return self
        pos = 0
        while pos < len(text):
            m = _ASSIGN.match(text, pos)
            if not m or m.end() == pos:
                raise ValueError('Could not parse hparams string at: %r' % text[pos:])
            name, raw = m.group(1), m.group(2)
            if name not in self._keys:
                raise ValueError('Unknown hyperparameter type for %s' % name)
            cur = getattr(self, name)
            if raw.startswith('['):
                items = [s for s in raw[1:-1].split(',') if s.strip() != '']
                proto = cur[0] if isinstance(cur, (list, tuple)) and len(cur) else None
                val = [_parse_scalar(s, proto) for s in items]
                if isinstance(cur, tuple):
                    val = tuple(val)
            else:
                if isinstance(cur, (list, tuple)):
                    raise ValueError('Must pass a list for multi-valued parameter: %s' % name)
                val = _parse_scalar(raw, cur)
            object.__setattr__(self, name, val)
            pos = m.end()
        return self

    def __repr__(self):
        return 'HParams(%s)' % ', '.join('%s=%r' % (k, getattr(self, k)) for k in self._keys)
