"""Reader for TensorFlow-1.x checkpoints ("tensor bundles": ``<prefix>.index`` + ``<prefix>.data-00000-of-0000N``) so that a
``wavenet_model.ckpt-N`` trained with the reference can be loaded into the flat parameter buffer of this tree
 -- pure Python + numpy, no TensorFlow.

Format (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/table*, i.e. the LevelDB table format):
  * ``.index`` is an SSTable: data blocks of prefix-compressed (key, value) entries followed by a restart array, each block
    trailed by 1 byte compression type (0 = none -- what BundleWriter uses; 1 = snappy is rejected here) and a 4-byte masked
    crc32c; an index block mapping separator keys to block handles; a 48-byte footer (metaindex handle, index handle, padding,
    magic 0xdb4775248b80fb57).
  * key ""        -> BundleHeaderProto  {1: num_shards, 2: endianness, 3: version}
  * key <tensor>  -> BundleEntryProto   {1: dtype, 2: TensorShapeProto{2: Dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c}
  * the data shards hold the raw little-endian tensor bytes at [offset, offset + size).
The reference saves its variables under the names of their exponential-moving-average shadows
(``<variable op name>/ExponentialMovingAverage``, wavenet_vocoder/train.py:67-83) plus ``global_step``.

Targeted writers: ``tf.train.Saver`` of TensorFlow 1.0 - 1.15 with its default ``write_version = SaverDef.V2`` (what the reference's
``train.py:67-87`` uses on TF 1.x): uncompressed index blocks, one or several data shards.  Not supported: the pre-1.0 single-file
V1 format (``write_version = V1``), snappy-compressed index blocks (rejected with a message), TF-2 object-graph checkpoints.

Verification status: no TensorFlow-written checkpoint is available offline.  The reader is exercised (a) against a fixture
assembled byte by byte from the format documents by an INDEPENDENT script (oracle/gen_tf_bundle_fixture.py: own bit-serial CRC-32C
pinned to the RFC 3720 known answers, own varint / proto / block / footer assembly; tests/golden/tf_bundle/), (b) against files
produced by ``write_bundle`` below (round trip at model size), and it verifies every checksum the format carries (table blocks,
tensor data), so a misparse cannot go unnoticed.  It has NOT been run on a file written by TensorFlow itself.
"""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9
_DTYPES = {DT_FLOAT: np.dtype('<f4'), DT_INT32: np.dtype('<i4'), DT_INT64: np.dtype('<i8'), 2: np.dtype('<f8')}


# ------------------------------------------------------------------ varints / protobuf wire format
def _read_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]; pos += 1
        result |= (b & 0x7f) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _write_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _parse_proto(buf):
    """-> list of (field number, wire type, value); value = int (varint / fixed) or bytes (length-delimited)."""
    fields, pos = [], 0
    while pos < len(buf):
        key, pos = _read_varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _read_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError('unsupported protobuf wire type %d' % wt)
        fields.append((fn, wt, v))
    return fields


def _parse_entry(buf):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None)
    for fn, wt, v in _parse_proto(buf):
        if fn == 1: e['dtype'] = v
        elif fn == 2:
            for f2, _, v2 in _parse_proto(v):
                if f2 == 2:                                    # Dim
                    size = 0
                    for f3, _, v3 in _parse_proto(v2):
                        if f3 == 1: size = v3 if v3 < (1 << 63) else v3 - (1 << 64)
                    e['shape'].append(int(size))
        elif fn == 3: e['shard_id'] = v
        elif fn == 4: e['offset'] = v
        elif fn == 5: e['size'] = v
        elif fn == 6: e['crc32c'] = v
    return e


# ------------------------------------------------------------------ LevelDB table
def _block_handle(buf, pos):
    off, pos = _read_varint(buf, pos)
    size, pos = _read_varint(buf, pos)
    return (off, size), pos


def _read_block(data, handle):
    off, size = handle
    if off + size + 5 > len(data):
        raise ValueError('table block [%d, %d) runs past the end of the index file' % (off, off + size + 5))
    contents = data[off:off + size]
    ctype = data[off + size]
    if ctype != 0:
        raise NotImplementedError('compressed table block (type %d): tensor bundles are written uncompressed' % ctype)
    want = struct.unpack_from('<I', data, off + size + 1)[0]           # masked crc32c over contents + type byte (table format.cc)
    if _mask_crc(crc32c(data[off:off + size + 1])) != want:
        raise ValueError('checksum mismatch in the table block at offset %d of the checkpoint index (corrupt file)' % off)
    return contents


def _block_entries(block):
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _read_varint(block, pos)
        non_shared, pos = _read_varint(block, pos)
        vlen, pos = _read_varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared]); pos += non_shared
        val = bytes(block[pos:pos + vlen]); pos += vlen
        yield key, val


def read_index(index_path):
    """-> (header dict, {tensor name: entry dict})"""
    data = open(index_path, 'rb').read()
    if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError('%s is not a TensorFlow checkpoint index (bad table magic)' % index_path)
    footer = data[-48:]
    _, pos = _block_handle(footer, 0)                           # metaindex (unused)
    index_handle, _ = _block_handle(footer, pos)
    header, entries = {}, {}
    for _, handle_bytes in _block_entries(_read_block(data, index_handle)):
        handle, _ = _block_handle(handle_bytes, 0)
        for k, v in _block_entries(_read_block(data, handle)):
            if k == b'':
                for fn, _, val in _parse_proto(v):
                    if fn == 1: header['num_shards'] = val
                    elif fn == 2: header['endianness'] = val
            else:
                entries[k.decode('utf-8')] = _parse_entry(v)
    header.setdefault('num_shards', 1)
    if header.get('endianness', 0) != 0:
        raise NotImplementedError('big-endian bundle')
    return header, entries


def load_checkpoint(prefix, verify=True):
    """``prefix`` as TF uses it (e.g. .../wavenet_model.ckpt-100000).  -> {variable name: numpy array}.  verify: check the per-tensor
    masked crc32c the bundle stores (pure-Python CRC: ~1 s per 10 MB; the table-block checksums of the index are always checked)."""
    header, entries = read_index(prefix + '.index')
    n = header['num_shards']
    shards = {}
    out = {}
    for name, e in entries.items():
        if e['dtype'] not in _DTYPES:
            continue                                            # strings etc. are of no use here
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, n), dtype=np.uint8, mode='r')
        raw = np.asarray(shards[sid][e['offset']:e['offset'] + e['size']])
        if raw.size != e['size']:
            raise ValueError('tensor %s: data shard %d ends before offset %d + size %d' % (name, sid, e['offset'], e['size']))
        if verify and e['crc32c'] is not None and _mask_crc(crc32c(raw.tobytes())) != e['crc32c']:
            raise ValueError('checksum mismatch in the data of tensor %s (corrupt checkpoint shard)' % name)
        arr = raw.view(_DTYPES[e['dtype']])
        out[name] = arr.reshape(e['shape']).copy()
    return out


# ------------------------------------------------------------------ reference variable names -> this tree's tensor names
_EMA = '/ExponentialMovingAverage'


def map_reference_name(name):
    """TF variable (or EMA shadow) name of the reference's WaveNet graph -> tensor name of wn_tensor_info, or None.
    Matching is by the layers' own (unique) names, so any enclosing variable scopes (``WaveNet_model/inference/...``) are ignored."""
    if name.endswith(_EMA):
        name = name[:-len(_EMA)]
    leaf = name.rsplit('/', 1)[-1]
    if leaf not in ('kernel', 'bias', 'g', 'gc_embedding'):
        return None
    if leaf == 'gc_embedding':
        return 'gc_embedding'
    m = re.search(r'residual_block_(causal|cin|gin|skip|out)_conv_ResidualConv1DGLU_(\d+)/', name)
    if m:
        return 'ResidualConv1DGLU_%s/residual_block_%s_conv/%s' % (m.group(2), m.group(1), leaf)
    m = re.search(r'(ConvTranspose2D|ConvTranspose1D|ResizeConvolution|SubPixelConvolution)_layer_(\d+)/', name)
    if m:
        return 'local_conditioning_upsampling_%d/%s' % (int(m.group(2)) + 1, leaf)
    m = re.search(r'(final_convolution_[12]|input_convolution)/(?:[^/]+/)*%s$' % leaf, name)
    if m:
        return '%s/%s' % (m.group(1), leaf)
    return None


def load_reference_checkpoint(prefix, layout):
    """Fill a flat fp32 parameter vector in the engine's ``layout`` (name -> (shape, offset)) from a reference checkpoint.
    Returns (flat numpy array, global_step or None, sorted list of layout names that were NOT found)."""
    tensors = load_checkpoint(prefix)
    total = max(off + int(np.prod(shape)) for shape, off in layout.values())
    flat = np.zeros((total + 7) // 8 * 8, dtype=np.float32)
    found = set()
    # prefer the EMA shadows' raw variables as the reference stores them (train.py:75-83), fall back to plain names
    for name in sorted(tensors, key=lambda n: (not n.endswith(_EMA), n)):
        tgt = map_reference_name(name)
        if tgt is None or tgt not in layout or tgt in found:
            continue
        shape, off = layout[tgt]
        arr = tensors[name]
        if tuple(arr.shape) != tuple(shape):
            raise ValueError('checkpoint tensor %s has shape %s, the model expects %s for %s' % (name, arr.shape, tuple(shape), tgt))
        flat[off:off + arr.size] = arr.astype(np.float32).reshape(-1)
        found.add(tgt)
    step = tensors.get('global_step')
    return flat, (int(step) if step is not None else None), sorted(set(layout) - found)


# ------------------------------------------------------------------ writer (tests; same spec)
def _crc32c_table():
    poly, table = 0x82f63b78, []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ poly if c & 1 else c >> 1
        table.append(c)
    return table


_CRC_TABLE = _crc32c_table()


def crc32c(data, crc=0):
    crc ^= 0xffffffff
    for b in data:
        crc = _CRC_TABLE[(crc ^ b) & 0xff] ^ (crc >> 8)
    return crc ^ 0xffffffff


def _mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xffffffff) + 0xa282ead8) & 0xffffffff


def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b''
    for i, (k, v) in enumerate(items):
        if i % restart_interval == 0:
            restarts.append(len(out)); shared = 0
        else:
            shared = 0
            while shared < min(len(last), len(k)) and last[shared] == k[shared]:
                shared += 1
        out += _write_varint(shared) + _write_varint(len(k) - shared) + _write_varint(len(v)) + k[shared:] + v
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_bundle(prefix, tensors, entries_per_block=8):
    """Write {name: numpy array} as a one-shard tensor bundle (used by the tests to exercise the reader on the spec)."""
    data = bytearray()
    items = [(b'', b'\x08\x01' + b'\x10\x00' + b'\x1a\x02\x08\x01')]       # header: num_shards=1, little endian, version{producer=1}
    for name in sorted(tensors):
        a = np.asarray(tensors[name])                # (ascontiguousarray would turn a scalar into shape [1])
        dt = {np.dtype('float32'): DT_FLOAT, np.dtype('int32'): DT_INT32, np.dtype('int64'): DT_INT64}[a.dtype]
        raw = a.tobytes(order='C')
        shape = b''.join(b'\x12' + _write_varint(len(d)) + d for d in (b'\x08' + _write_varint(int(s)) for s in a.shape))
        e = b'\x08' + _write_varint(dt) + b'\x12' + _write_varint(len(shape)) + shape
        if len(data):
            e += b'\x20' + _write_varint(len(data))
        e += b'\x28' + _write_varint(len(raw)) + b'\x35' + struct.pack('<I', _mask_crc(crc32c(raw)))
        items.append((name.encode('utf-8'), e))
        data += raw
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))
    out, index_items = bytearray(), []

    def emit(block):
        handle = _write_varint(len(out)) + _write_varint(len(block))
        out.extend(block + b'\x00' + struct.pack('<I', _mask_crc(crc32c(block + b'\x00'))))
        return handle
    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        index_items.append((chunk[-1][0], emit(_build_block(chunk))))
    meta = emit(_build_block([]))
    idx = emit(_build_block(index_items, restart_interval=1))
    footer = meta + idx
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', TABLE_MAGIC)
    out.extend(footer)
    with open(prefix + '.index', 'wb') as f:
        f.write(bytes(out))
