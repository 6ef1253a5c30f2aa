"""Reader for TensorFlow "bundle v2" checkpoints (`<prefix>.index` + `<prefix>.data-00000-of-00001`).

The reference restores its trained weights with `tf.train.Saver().restore(sess, checkpoint)`
(/root/reference/src/process_functional.py:32,43) where `checkpoint` is the `--resume` prefix
(/root/reference/src/match.py:21,132).  TensorFlow is not part of this stack, so this module parses the
two files itself:

* `.index` is an uncompressed LevelDB-style sorted table: data block(s) of prefix-compressed
  `(shared, non_shared, value_len, key_suffix, value)` entries, then a metaindex block, an index block and a
  48-byte footer holding the two block handles and the table magic 0xdb4775248b80fb57.
* every value (except key "" = BundleHeaderProto) is a BundleEntryProto:
  field 1 dtype, field 2 TensorShapeProto, field 3 shard_id, field 4 offset, field 5 size,
  field 6 fixed32 masked crc32c of the tensor bytes.
* `.data-*` is the raw little-endian tensor bytes at those offsets.

Only what the MC-CNN "fast" network needs is supported: float32 tensors in shard 0, no slices.
"""
import os
import struct

import numpy as np

_TABLE_MAGIC = 0xdb4775248b80fb57
_DT_FLOAT = 1


def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not (b & 0x80):
            return result, pos
        shift += 7


def _read_block(buf, offset, size):
    """Returns the list of (key, value) in one table block (compression type must be 0)."""
    block = buf[offset:offset + size]
    ctype = buf[offset + size]
    if ctype != 0:
        raise ValueError("compressed checkpoint index blocks are not supported (type %d)" % ctype)
    num_restarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * num_restarts
    pos = 0
    key = b""
    out = []
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        value_len, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + value_len])))
        pos += value_len
    return out


def _parse_proto(buf):
    """Minimal protobuf wire parser -> {field_number: [raw values]}."""
    fields = {}
    pos = 0
    n = len(buf)
    while pos < n:
        tag, pos = _varint(buf, pos)
        fno, wt = tag >> 3, tag & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            val = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        fields.setdefault(fno, []).append(val)
    return fields


def _parse_shape(buf):
    dims = []
    for d in _parse_proto(buf).get(2, []):
        size = _parse_proto(d).get(1, [0])[0]
        dims.append(int(size))
    return dims


# ---- CRC32C (Castagnoli), table driven; TF stores the "masked" value --------------------------------------
_CRC_TABLE = None


def _crc32c(data):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl.append(c)
        _CRC_TABLE = tbl
    crc = 0xFFFFFFFF
    tbl = _CRC_TABLE
    for b in data:
        crc = tbl[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def _mask_crc(crc):
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def read_index(prefix):
    """Returns {name: dict(dtype, shape, shard, offset, size, crc32c)} for every tensor in the bundle."""
    with open(prefix + ".index", "rb") as f:
        buf = f.read()
    if len(buf) < 48:
        raise ValueError("checkpoint index too short")
    footer = buf[-48:]
    magic = struct.unpack_from("<Q", footer, 40)[0]
    if magic != _TABLE_MAGIC:
        raise ValueError("not a TensorFlow bundle index (bad table magic)")
    pos = 0
    _meta_off, pos = _varint(footer, pos)
    _meta_sz, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_sz, pos = _varint(footer, pos)
    entries = {}
    for _k, handle in _read_block(buf, idx_off, idx_sz):
        boff, hp = _varint(handle, 0)
        bsz, hp = _varint(handle, hp)
        for key, val in _read_block(buf, boff, bsz):
            if key == b"":
                continue  # BundleHeaderProto
            p = _parse_proto(val)
            entries[key.decode()] = dict(
                dtype=p.get(1, [0])[0],
                shape=_parse_shape(p[2][0]) if 2 in p else [],
                shard=p.get(3, [0])[0],
                offset=p.get(4, [0])[0],
                size=p.get(5, [0])[0],
                crc32c=p.get(6, [None])[0],
            )
    return entries


def load_checkpoint(prefix, verify_crc=True, skip_slots=True):
    """Returns {variable name: float32 ndarray}.  Optimizer slot variables (`.../Momentum`) are skipped."""
    entries = read_index(prefix)
    data_path = prefix + ".data-00000-of-00001"
    with open(data_path, "rb") as f:
        data = f.read()
    out = {}
    for name, e in sorted(entries.items()):
        if skip_slots and name.endswith("/Momentum"):
            continue
        if e["dtype"] != _DT_FLOAT or e["shard"] != 0:
            raise ValueError("tensor %s: only float32 tensors in shard 0 are supported" % name)
        raw = data[e["offset"]:e["offset"] + e["size"]]
        if len(raw) != e["size"]:
            raise ValueError("tensor %s: data file truncated" % name)
        if verify_crc and e["crc32c"] is not None:
            if _mask_crc(_crc32c(raw)) != e["crc32c"]:
                raise ValueError("tensor %s: crc32c mismatch" % name)
        arr = np.frombuffer(raw, dtype="<f4").reshape(e["shape"]).copy()
        out[name] = arr
    return out


def load_fast_net_weights(checkpoint):
    """Weights of the 5-layer 'fast' MC-CNN as a list of (w_hwio, bias) float32 pairs.

    `checkpoint` may be a TF bundle prefix (the reference's --resume value) or an .npz written by
    `save_npz` with keys conv{k}/weights, conv{k}/biases.
    """
    if checkpoint is None:
        raise ValueError("a checkpoint is required (the reference's Saver.restore(None) fails the same way, "
                         "process_functional.py:43)")
    if checkpoint.endswith(".npz"):
        blob = dict(np.load(checkpoint))
    elif os.path.isfile(checkpoint + ".index"):
        blob = load_checkpoint(checkpoint)
    else:
        raise FileNotFoundError("no checkpoint at %r (.npz or TF bundle prefix expected)" % checkpoint)
    layers = []
    k = 1
    while "conv%d/weights" % k in blob:
        layers.append((np.ascontiguousarray(blob["conv%d/weights" % k], dtype=np.float32),
                       np.ascontiguousarray(blob["conv%d/biases" % k], dtype=np.float32)))
        k += 1
    if not layers:
        raise ValueError("checkpoint %r holds no conv<k>/weights variables" % checkpoint)
    return layers


def save_npz(path, layers):
    blob = {}
    for k, (w, b) in enumerate(layers, start=1):
        blob["conv%d/weights" % k] = w
        blob["conv%d/biases" % k] = b
    np.savez(path, **blob)
