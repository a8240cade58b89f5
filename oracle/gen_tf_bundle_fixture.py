"""Hand-assembled TensorFlow-1 tensor-bundle fixture (tests/golden/tf_bundle/fixture.ckpt-7.{index,data-00000-of-00001}).

Written INDEPENDENTLY of tacotron-2_amd/wavenet_vocoder/tf_checkpoint.py (it imports nothing from it, and uses a bit-serial CRC
instead of that file's table) from the format documents, so that the product's reader is not only tested against its own writer:
  * LevelDB table format (tensorflow/core/lib/io/format.{h,cc}, table_builder.cc; doc/table_format.md of LevelDB):
      data block   = entries {varint shared, varint non_shared, varint value_len, key delta, value} ... restart offsets (fixed32 each,
                     one per 16 entries), fixed32 restart count
      block trailer= 1 byte compression type (0 = none) + fixed32 MASKED crc32c over block contents + type byte
      index block  = one entry per data block: key >= last key of the block, value = BlockHandle {varint offset, varint size}
      footer       = metaindex BlockHandle, index BlockHandle, zero padding to 40 bytes, fixed64 magic 0xdb4775248b80fb57
      masked crc   = ((crc >> 15) | (crc << 17)) + 0xa282ead8   (mod 2^32)
  * tensor bundle (tensorflow/core/protobuf/tensor_bundle.proto, util/tensor_bundle/tensor_bundle.cc):
      key ""       -> BundleHeaderProto {1: num_shards = 1, 3: VersionDef {1: producer = 1}}    (endianness LITTLE = 0 is the default: omitted)
      key <name>   -> BundleEntryProto  {1: dtype, 2: TensorShapeProto {2: Dim {1: size}}*, 4: offset, 5: size, 6: fixed32 MASKED crc32c of the bytes}
      (proto3: zero-valued fields -- shard_id 0, offset 0 -- are not serialised)
Known answers checked while building: CRC-32C of "123456789" = 0xE3069283 and of 32 zero bytes = 0x8A9136AA (RFC 3720 B.4); the
footer ends with the bytes 57 fb 80 8b 24 75 47 db.
Tensors: two variables stored under their EMA shadow names as the reference saves them (train.py:75-83) with a long common key
prefix (exercises the prefix compression), and the int64 global_step.
"""
import os
import struct

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(os.environ.get('WN_GOLDEN_DIR') or os.path.join(ROOT, 'tests', 'golden'), 'tf_bundle')


def crc32c(data):
    crc = 0xFFFFFFFF
    for b in data:
        crc ^= b
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def masked(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    out = b''
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80]); v >>= 7
    return out + bytes([v])


def block(entries):
    """entries: sorted [(key bytes, value bytes)], restart interval 16 -> (contents + trailer)."""
    body, restarts, last = b'', [], b''
    for i, (k, v) in enumerate(entries):
        if i % 16 == 0:
            restarts.append(len(body)); shared = 0
        else:
            shared = 0
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        body += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        last = k
    if not entries:
        restarts = [0]
    body += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
    return body + b'\x00' + struct.pack('<I', masked(crc32c(body + b'\x00')))


def entry_proto(dtype, shape, offset, raw):
    shp = b''.join(b'\x12' + varint(len(d)) + d for d in (b'\x08' + varint(s) for s in shape))
    p = b'\x08' + varint(dtype) + b'\x12' + varint(len(shp)) + shp
    if offset:
        p += b'\x20' + varint(offset)
    p += b'\x28' + varint(len(raw)) + b'\x35' + struct.pack('<I', masked(crc32c(raw)))
    return p


def main():
    assert crc32c(b'123456789') == 0xE3069283 and crc32c(bytes(32)) == 0x8A9136AA
    rng = np.random.RandomState(5)
    tensors = [
        ('WaveNet_model/inference/final_convolution_1/bias/ExponentialMovingAverage', 1, rng.randn(8).astype('<f4')),
        ('WaveNet_model/inference/final_convolution_1/kernel/ExponentialMovingAverage', 1, rng.randn(1, 8, 8).astype('<f4')),
        ('global_step', 9, np.array(7, dtype='<i8')),
    ]
    tensors.sort(key=lambda t: t[0].encode())
    data, entries = b'', [(b'', b'\x08\x01\x1a\x02\x08\x01')]
    for name, dt, arr in tensors:
        raw = arr.tobytes()
        entries.append((name.encode(), entry_proto(dt, arr.shape, len(data), raw)))
        data += raw
    db = block(entries)                                             # one data block at offset 0
    mb = block([])                                                  # empty metaindex block
    data_handle = varint(0) + varint(len(db) - 5)
    ib = block([(entries[-1][0] + b'\x00', data_handle)])           # separator key >= the block's last key
    meta_off, idx_off = len(db), len(db) + len(mb)
    footer = varint(meta_off) + varint(len(mb) - 5) + varint(idx_off) + varint(len(ib) - 5)
    footer += bytes(40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    index = db + mb + ib + footer
    assert index[-8:] == bytes.fromhex('57fb808b247547db') and len(footer) == 48
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'fixture.ckpt-7.index'), 'wb') as f:
        f.write(index)
    with open(os.path.join(OUT, 'fixture.ckpt-7.data-00000-of-00001'), 'wb') as f:
        f.write(data)
    np.savez(os.path.join(OUT, 'expected.npz'), **{n: a for n, _, a in tensors})
    print('tf_bundle fixture: index %d B, data %d B, %d tensors' % (len(index), len(data), len(tensors)))


if __name__ == '__main__':
    main()
