#!/usr/bin/env python3
"""Build libwavenet_mi355.so for gfx950 with hipcc (cross-compiles without a GPU).

    python tacotron-2_amd/csrc/build.py [--force]

One hipcc invocation per translation unit (in parallel), then one link.  The .so is written next to
the sources (git-ignored, but it travels to the GPU box with the gpurun snapshot).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ['wn_api.hip', 'wn_misc.hip', 'wn_train.hip', 'wn_synth.hip', 'wn_synth_pipe.hip', 'wn_f32.hip', 'wn_synth_f32.hip']
HEADERS = ['wn_common.h', 'wn_tile.h', 'wn_wgrad.h', 'wn_mulaw_tables.h', os.path.join('..', '..', 'include', 'wavenet_mi355.h')]
LIB = os.path.join(HERE, 'libwavenet_mi355.so')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++20', '-fPIC', '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function', '-Wno-unused-value', '-Wno-inline-asm']


LAST_BUILD = {}      # translation unit -> 'compiled' | 'reused' (object newer than its source, every header and this script); 'link' likewise


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


def _compile(src):
    obj = os.path.join(HERE, src.replace('.hip', '.o'))
    newest = max([_mtime(os.path.join(HERE, src))] + [_mtime(os.path.join(HERE, h)) for h in HEADERS] + [_mtime(__file__)])
    if _mtime(obj) > newest:
        LAST_BUILD[src] = 'reused'
        return obj, ''
    LAST_BUILD[src] = 'compiled'
    cmd = ['hipcc'] + FLAGS + ['-c', os.path.join(HERE, src), '-o', obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('hipcc failed for %s:\n%s\n%s' % (src, r.stdout, r.stderr))
    return obj, r.stderr


def build(force=False, verbose=True):
    if force:
        for s in SOURCES:
            o = os.path.join(HERE, s.replace('.hip', '.o'))
            if os.path.exists(o):
                os.remove(o)
    with concurrent.futures.ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        res = list(ex.map(_compile, SOURCES))
    objs = [o for o, _ in res]
    warn = ''.join(w for _, w in res)
    if verbose and warn.strip():
        print(warn, file=sys.stderr)
    LAST_BUILD['link'] = 'reused'
    if _mtime(LIB) < max(_mtime(o) for o in objs):
        LAST_BUILD['link'] = 'linked'
        cmd = ['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('link failed:\n%s\n%s' % (r.stdout, r.stderr))
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    print('build mode: ' + ', '.join('%s %s' % kv for kv in sorted(LAST_BUILD.items())))
