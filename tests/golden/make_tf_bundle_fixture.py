#!/usr/bin/env python
"""Assembles tests/golden/tf_bundle/model.ckpt-7.{index,data-00000-of-00002,data-00001-of-00002} BYTE BY BYTE from the
published TensorFlow tensor-bundle / LevelDB-table format, independently of sse_amd.tf_checkpoint.write_bundle
(own varint, own CRC-32C, own block builder), so that the reader is checked against a second reading of the format:

  * two data shards (entries carry shard_id 0 and 1, header num_shards = 2);
  * three data blocks; keys inside a block are prefix-compressed against their predecessor with the LevelDB restart
    interval of 16 (one restart per block here) -- e.g. "shared_encoder/rnn/basic_lstm_cell/bias" shares 36 bytes with
    ".../kernel" -- plus a metaindex block, an index block whose keys are the blocks' last keys, and the 48-byte footer;
  * every block trailer = type byte 0 + masked CRC-32C over (block + type byte); every entry carries the masked crc32c
    of its tensor bytes (BundleEntryProto field 6, fixed32), dtype (DT_FLOAT = 1, DT_INT64 = 9), shape, offset, size.

TensorFlow itself cannot be installed in the build container: this is NOT a TensorFlow-written file.
Run from the repo root:  python tests/golden/make_tf_bundle_fixture.py
"""
import os
import struct

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tf_bundle")


def varint(v):
    out = b""
    while v >= 0x80:
        out += bytes([(v & 0x7F) | 0x80])
        v >>= 7
    return out + bytes([v])


def crc32c(data):                                       # bitwise, no table: deliberately not the reader's code
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def mask(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def tensors():
    rng = np.random.RandomState(7)
    t = {"shared_encoder/rnn/basic_lstm_cell/kernel": rng.standard_normal((11, 12)).astype(np.float32),
         "shared_encoder/rnn/basic_lstm_cell/bias": rng.standard_normal((12,)).astype(np.float32),
         "shared_encoder/rnn/basic_lstm_cell/kernel/Adagrad": np.full((11, 12), 0.1, np.float32),
         "shared_encoder/rnn/basic_lstm_cell/bias/Adagrad": np.full((12,), 0.1, np.float32),
         "shared_encoder/src_M": rng.standard_normal((3, 5)).astype(np.float32),
         "shared_encoder/tgt_M": rng.standard_normal((3, 5)).astype(np.float32),
         "word_embedding": rng.uniform(-0.25, 0.25, size=(9, 8)).astype(np.float32),
         "word_embedding/Adagrad": np.full((9, 8), 0.1, np.float32),
         "global_step": np.array(7, np.int64),
         "learning_rate": np.array(0.81, np.float32)}
    return t


def entry(dtype, shape, shard, offset, size, crc_masked):
    dims = b""
    for d in shape:
        dim = b"\x08" + varint(d)                       # TensorShapeProto.Dim.size = 1
        dims += b"\x12" + varint(len(dim)) + dim        # TensorShapeProto.dim = 2
    out = b"\x08" + varint(dtype)                       # dtype = 1
    out += b"\x12" + varint(len(dims)) + dims           # shape = 2
    if shard:
        out += b"\x18" + varint(shard)                  # shard_id = 3 (proto3: 0 is not written)
    if offset:
        out += b"\x20" + varint(offset)                 # offset = 4
    out += b"\x28" + varint(size)                       # size = 5
    out += b"\x35" + struct.pack("<I", crc_masked)      # crc32c = 6, fixed32
    return out


def block(pairs):
    body, prev = b"", b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i > 0:                                       # restart interval 16 > entries per block: one restart at 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        body += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    return body + struct.pack("<I", 0) + struct.pack("<I", 1)


def main():
    os.makedirs(HERE, exist_ok=True)
    t = tensors()
    names = sorted(t)                                   # table keys are sorted bytewise
    shard_blobs = [b"", b""]
    pairs = [(b"", b"\x08\x02" + b"\x1a\x02\x08\x01")]  # BundleHeaderProto: num_shards = 2, version { producer: 1 }
    for i, n in enumerate(names):
        a = t[n]
        raw = a.tobytes()
        sid = i % 2                                     # alternate the shards
        dtype = 1 if a.dtype == np.float32 else 9
        pairs.append((n.encode(), entry(dtype, a.shape, sid, len(shard_blobs[sid]), len(raw), mask(crc32c(raw)))))
        shard_blobs[sid] += raw
    for sid in range(2):
        with open(os.path.join(HERE, "model.ckpt-7.data-%05d-of-00002" % sid), "wb") as f:
            f.write(shard_blobs[sid])
    out, index_pairs = b"", []
    for chunk in (pairs[0:4], pairs[4:8], pairs[8:]):
        b = block(chunk)
        index_pairs.append((chunk[-1][0], varint(len(out)) + varint(len(b))))
        out += b + b"\x00" + struct.pack("<I", mask(crc32c(b + b"\x00")))
    meta = block([])
    meta_handle = varint(len(out)) + varint(len(meta))
    out += meta + b"\x00" + struct.pack("<I", mask(crc32c(meta + b"\x00")))
    body = b""
    for k, v in index_pairs:                            # index block: restart interval 1 (every key whole)
        body += varint(0) + varint(len(k)) + varint(len(v)) + k + v
    restarts, pos = [], 0
    for k, v in index_pairs:
        restarts.append(pos)
        pos += len(varint(0) + varint(len(k)) + varint(len(v)) + k + v)
    idx = body + b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    idx_handle = varint(len(out)) + varint(len(idx))
    out += idx + b"\x00" + struct.pack("<I", mask(crc32c(idx + b"\x00")))
    footer = meta_handle + idx_handle
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    with open(os.path.join(HERE, "model.ckpt-7.index"), "wb") as f:
        f.write(out)
    np.savez(os.path.join(HERE, "expected.npz"), **{k.replace("/", "|"): v for k, v in t.items()})
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
