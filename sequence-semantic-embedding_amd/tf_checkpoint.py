"""Reader for TensorFlow V2 checkpoints ("tensor bundles": `<prefix>.index` + `<prefix>.data-0000N-of-0000M`), so
that a model directory trained by the reference (`saver.save`, sse_train.py:205,212,232; `tf.train.Saver` default
write_version = V2) loads without TensorFlow:

    python -m sse_amd.tf_checkpoint models-classification            # writes <ckpt>.npz next to every checkpoint
    Saver.restore(None, "models-classification/SSE-LSTM.ckpt-4000")  # reads .index/.data directly when no .npz exists

The variable names ARE the interface: the library addresses weights by their TF names (`word_embedding`,
`source_encoder/rnn/basic_lstm_cell/kernel`, `.../Adagrad`, `learning_rate`, `global_step`), so conversion is a
dictionary copy.

Format (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, tensorflow/core/lib/io/table*.cc -- a LevelDB
table): the .index file is a sorted string table -- data blocks of prefix-compressed (key, value) entries with a
restart array, an index block mapping last-keys to block handles, a 48-byte footer (metaindex handle, index handle,
magic 0xdb4775248b80fb57); every block is followed by a 1-byte compression type and a 4-byte masked CRC.  Key "" holds
a BundleHeaderProto, every other key a BundleEntryProto {1: dtype, 2: shape {2: dim {1: size}}, 3: shard_id, 4: offset,
5: size, 6: crc32c}; tensor bytes are raw little-endian at [offset, offset + size) of data shard `shard_id`.
Every table block's masked CRC-32C (LevelDB: crc32c(block + type byte), rotated right by 15 plus 0xa282ead8) and every
tensor's masked crc32c (entry field 6) are VERIFIED on read, and `offset + size` is checked against the shard length: a
truncated or corrupted checkpoint raises instead of loading garbage.  The CRC itself is pinned by the RFC 3720
known-answer vectors (tests/test_serving_and_checkpoint.py).
NOT validated against a TensorFlow-written file in the build container (TensorFlow cannot be installed there); the
unit tests round-trip through `write_bundle` below and read a byte-for-byte hand-assembled two-shard fixture
(tests/golden/tf_bundle/, built by tests/golden/make_tf_bundle_fixture.py from the format description alone).  `tools/tf_checkpoint_to_npz.py`
is the alternative for a machine that has TensorFlow (`tf.train.load_checkpoint`).  Snappy-compressed blocks (not
what BundleWriter emits) and V1 single-file checkpoints are rejected with a clear error.
"""
import glob
import os
import struct
import sys

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}   # types.proto DataType


def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _read_block(data, offset, size):
    if offset + size + 5 > len(data):
        raise ValueError("checkpoint index is truncated: block [%d, %d) + 5-byte trailer exceeds %d bytes"
                         % (offset, offset + size, len(data)))
    ctype = data[offset + size]
    if ctype != 0:
        raise ValueError("checkpoint index block is compressed (type %d): only uncompressed tensor bundles are "
                         "supported; convert with tools/tf_checkpoint_to_npz.py on a machine with TensorFlow" % ctype)
    want = struct.unpack_from("<I", data, offset + size + 1)[0]
    got = _masked_crc(data[offset:offset + size + 1])              # contents + the type byte (table/format.cc)
    if got != want:
        raise ValueError("checkpoint index block at offset %d fails its CRC-32C (stored %08x, computed %08x): the file is "
                         "corrupt" % (offset, want, got))
    return data[offset:offset + size]


def _block_entries(block):
    """(key, value) pairs of one table block (prefix-compressed keys, restart array at the end)."""
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * num_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def _proto_fields(buf):
    """Minimal protobuf wire-format walk: yields (field number, wire type, value)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _parse_entry(buf):
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "slices": False, "crc32c": None}
    for field, wt, v in _proto_fields(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            for f2, _, v2 in _proto_fields(v):
                if f2 == 2:                                   # TensorShapeProto.dim
                    size = 0
                    for f3, _, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = v3 if v3 < (1 << 63) else v3 - (1 << 64)
                    e["shape"].append(size)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:                                      # fixed32, masked crc32c of the tensor bytes
            e["crc32c"] = struct.unpack("<I", bytes(v))[0]
        elif field == 7:
            e["slices"] = True
    return e


def is_tf_checkpoint(prefix):
    return os.path.exists(prefix + ".index")


def read_bundle(prefix):
    """{variable name: ndarray} of the checkpoint `<prefix>.index` / `<prefix>.data-*`."""
    if not os.path.exists(prefix + ".index"):
        if os.path.exists(prefix) and not os.path.isdir(prefix):
            raise ValueError("%s looks like a V1 (single-file) TensorFlow checkpoint; only V2 bundles are supported" % prefix)
        raise FileNotFoundError(prefix + ".index")
    data = open(prefix + ".index", "rb").read()
    if len(data) < 48 or struct.unpack_from("<Q", data, len(data) - 8)[0] != TABLE_MAGIC:
        raise ValueError("%s.index is not a TensorFlow tensor-bundle index (bad table magic)" % prefix)
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)          # metaindex handle (unused)
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries, num_shards = {}, 1
    for _, handle in _block_entries(_read_block(data, idx_off, idx_size)):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for key, value in _block_entries(_read_block(data, off, size)):
            if key == b"":
                for field, _, v in _proto_fields(value):
                    if field == 1:
                        num_shards = v
                    elif field == 2 and v != 0:
                        raise ValueError("big-endian tensor bundle not supported")
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    shards = {}
    out = {}
    for name, e in entries.items():
        if e["slices"]:
            raise ValueError("variable %s is stored as slices (partitioned variable): not supported" % name)
        if e["dtype"] not in DTYPES:
            continue                                          # e.g. string tensors of a SaverDef: not a variable
        sid = e["shard_id"]
        if sid not in shards:
            shards[sid] = np.memmap("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), dtype=np.uint8, mode="r")
        if e["offset"] + e["size"] > shards[sid].shape[0]:
            raise ValueError("variable %s: bytes [%d, %d) lie outside data shard %d (%d bytes): truncated checkpoint"
                             % (name, e["offset"], e["offset"] + e["size"], sid, shards[sid].shape[0]))
        raw = bytes(shards[sid][e["offset"]:e["offset"] + e["size"]])
        got_crc = _masked_crc(raw) if e["crc32c"] is not None else None
        if got_crc != e["crc32c"]:
            raise ValueError("variable %s fails its CRC-32C (stored %08x, computed %08x): the data shard is corrupt"
                             % (name, e["crc32c"], got_crc))
        want_bytes = int(np.prod(e["shape"], dtype=np.int64)) * np.dtype(DTYPES[e["dtype"]]).itemsize
        if want_bytes != e["size"]:
            raise ValueError("variable %s: %d bytes stored for shape %s" % (name, e["size"], e["shape"]))
        out[name] = np.frombuffer(raw, dtype=DTYPES[e["dtype"]]).reshape(e["shape"]).copy()
    return out


def to_npz_arrays(variables):
    """TF names -> the arrays Saver.restore expects (float32 variables / slots, learning_rate, global_step)."""
    out = {}
    for name, a in variables.items():
        if name == "global_step":
            out[name] = np.int64(a)
        elif name == "learning_rate":
            out[name] = np.float32(a)
        elif a.dtype == np.float32 and a.ndim >= 1:
            out[name] = a
    return out


def convert(prefix):
    arrays = to_npz_arrays(read_bundle(prefix))
    tmp = prefix + ".tmp.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, prefix + ".npz")
    return prefix + ".npz", sorted(arrays)


# --------------------------------------------------------------------------------------------------------------------
# writer (tests; also lets a model trained here be handed to tooling that expects the bundle layout)

def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _crc32c(data, _table=[]):
    """CRC-32C of a bytes-like object: the C routine of csrc/index_io.cpp through libsse_host.so (host only: no HIP
    runtime, no torch import) when it is built, else the table loop below (same values; seconds per embedding table)."""
    try:
        from . import _lib
        buf = data if isinstance(data, bytes) else bytes(data)   # (the reader hands over bytes: no copy)
        return int(_lib.load_host_library().sse_crc32c(buf, len(buf), 0))
    except (OSError, RuntimeError, AttributeError):            # library not built: pure-Python fallback of a host-only checksum
        data = bytes(data)
    if not _table:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _table.append(c)
    crc = 0xFFFFFFFF
    for b in data:
        crc = _table[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _masked_crc(data):
    c = _crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _build_block(pairs, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(pairs):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts) or 1)
    return bytes(out)


def _entry_proto(dtype_code, shape, offset, size, crc):
    dims = b"".join(b"\x12" + _put_varint(len(d)) + d for d in (b"\x08" + _put_varint(s) for s in shape))
    return (b"\x08" + _put_varint(dtype_code) + b"\x12" + _put_varint(len(dims)) + dims + b"\x20" + _put_varint(offset)
            + b"\x28" + _put_varint(size) + b"\x35" + struct.pack("<I", crc))


def write_bundle(prefix, variables, entries_per_block=7):
    """Write {name: ndarray} as a one-shard V2 bundle (uncompressed table, several data blocks)."""
    codes = {np.dtype(v): k for k, v in DTYPES.items()}
    blob, pairs = bytearray(), [(b"", b"\x08\x01\x1a\x02\x08\x01")]          # header: num_shards 1, version {producer 1}
    for name in sorted(variables):
        a = np.asarray(variables[name])
        raw = a.tobytes()          # C order
        pairs.append((name.encode("utf-8"), _entry_proto(codes[a.dtype], a.shape, len(blob), len(raw), _masked_crc(raw))))
        blob += raw
    with open("%s.data-00000-of-00001" % prefix, "wb") as f:
        f.write(bytes(blob))
    out, index_pairs = bytearray(), []
    for i in range(0, len(pairs), entries_per_block):
        chunk = pairs[i:i + entries_per_block]
        block = _build_block(chunk)
        index_pairs.append((chunk[-1][0], _put_varint(len(out)) + _put_varint(len(block))))
        out += block + b"\x00" + struct.pack("<I", _masked_crc(block + b"\x00"))
    meta = _build_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out += meta + b"\x00" + struct.pack("<I", _masked_crc(meta + b"\x00"))
    idx = _build_block(index_pairs, restart_interval=1)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out += idx + b"\x00" + struct.pack("<I", _masked_crc(idx + b"\x00"))
    footer = meta_handle + idx_handle
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if len(argv) != 1:
        raise SystemExit("usage: python -m sse_amd.tf_checkpoint <model_dir | checkpoint prefix>")
    target = argv[0]
    prefixes = [p[:-len(".index")] for p in sorted(glob.glob(os.path.join(target, "*.index")))] if os.path.isdir(target) else [target]
    if not prefixes:
        raise SystemExit("no *.index files under %s" % target)
    for p in prefixes:
        path, names = convert(p)
        print("%s: %d arrays (%s ...)" % (path, len(names), ", ".join(names[:3])))


if __name__ == "__main__":
    main()
