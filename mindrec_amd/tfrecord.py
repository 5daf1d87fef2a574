"""TFRecord files of `tf.train.Example` records, read and written without TensorFlow -- the second on-disk format the reference's
reader takes (models/wide_deep/src/datasets.py:226-271: `ds.TFRecordDataset(dataset_files, schema(feat_ids int32, feat_vals
float32, label float32), num_shards, shard_id, shard_equal_rows=True)`, every row packing 1000 samples), besides MindRecord
(MindSpore's own container, which cannot be restated).  Both formats are public:

  file    = record*
  record  = uint64 length | uint32 masked_crc32c(length) | byte data[length] | uint32 masked_crc32c(data)      (little endian)
  data    = protobuf Example { Features features = 1 }            Features { map<string, Feature> feature = 1 }
  Feature = oneof { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3 }
            FloatList { repeated float value = 1 [packed] }       Int64List { repeated int64 value = 1 [packed] }

Host-side numpy / pure Python: data preparation, not the hot path."""
import struct

import numpy as np

_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.zeros(256, np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1          # CRC-32C (Castagnoli), reflected
            t[i] = c
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data):
    data = bytes(data)
    if len(data) >= 256:              # (records are ~200 KB: the library's slicing-by-8 host routine; pure Python below, for the test box without it)
        try:
            import ctypes
            from . import _lib
            out = ctypes.c_uint32(0)
            _lib.call("mrec_crc32c_host", data, len(data), ctypes.byref(out))
            return int(out.value)
        except (ImportError, OSError):
            pass
    t = _crc_table()
    c = 0xFFFFFFFF
    for b in bytes(data):
        c = int(t[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ---- protobuf wire format, the four message types above ---------------------------------------------------------------------------
def _varint(n):
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _read_varint(buf, i):
    shift = val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _ld(field, payload):                      # a length-delimited field
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def encode_example(columns):
    """{name: 1-D array} -> serialized Example.  Integer arrays become Int64List, floating ones FloatList, bytes BytesList."""
    feats = b""
    for name in sorted(columns):
        v = columns[name]
        if isinstance(v, (bytes, bytearray)):
            feature = _ld(1, _ld(1, bytes(v)))
        else:
            a = np.asarray(v).reshape(-1)
            if a.dtype.kind == "f":
                feature = _ld(2, _ld(1, a.astype("<f4").tobytes()))
            elif a.dtype.kind in "iub":
                feature = _ld(3, _ld(1, b"".join(_varint(int(x)) for x in a)))
            else:
                raise TypeError(f"column {name!r}: unsupported dtype {a.dtype}")
        feats += _ld(1, _ld(1, name.encode()) + _ld(2, feature))
    return _ld(1, feats)


def _fields(buf):
    i, n = 0, len(buf)
    while i < n:
        key, i = _read_varint(buf, i)
        field, wt = key >> 3, key & 7
        if wt == 2:
            ln, i = _read_varint(buf, i)
            yield field, wt, buf[i:i + ln]
            i += ln
        elif wt == 0:
            v, i = _read_varint(buf, i)
            yield field, wt, v
        elif wt == 5:
            yield field, wt, buf[i:i + 4]
            i += 4
        elif wt == 1:
            yield field, wt, buf[i:i + 8]
            i += 8
        else:
            raise ValueError(f"unsupported wire type {wt}")


def _decode_feature(buf):
    for field, wt, val in _fields(buf):
        if field == 2:                                   # FloatList
            parts = []
            for f2, w2, v2 in _fields(val):
                if f2 == 1:
                    parts.append(np.frombuffer(bytes(v2), "<f4"))      # packed (or one unpacked fixed32)
            return np.concatenate(parts) if parts else np.zeros(0, np.float32)
        if field == 3:                                   # Int64List
            out = []
            for f2, w2, v2 in _fields(val):
                if f2 != 1:
                    continue
                if w2 == 0:
                    out.append(v2)
                else:
                    j = 0
                    while j < len(v2):
                        x, j = _read_varint(v2, j)
                        out.append(x)
            a = np.array(out, dtype=np.uint64).astype(np.int64)
            return a
        if field == 1:                                   # BytesList
            return [bytes(v2) for f2, w2, v2 in _fields(val) if f2 == 1]
    return np.zeros(0, np.float32)


def decode_example(data):
    """serialized Example -> {name: int64 / float32 array, or list of bytes}"""
    out = {}
    for f, _, features in _fields(memoryview(data)):
        if f != 1:
            continue
        for f1, _, entry in _fields(features):
            if f1 != 1:
                continue
            name, value = None, None
            for f2, _, v in _fields(entry):
                if f2 == 1:
                    name = bytes(v).decode()
                elif f2 == 2:
                    value = _decode_feature(v)
            if name is not None:
                out[name] = value
    return out


# ---- files ------------------------------------------------------------------------------------------------------------------------
def write_file(path, examples):
    """examples: iterable of {name: array}.  Returns the number of records written."""
    n = 0
    with open(path, "wb") as f:
        for ex in examples:
            data = encode_example(ex)
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))
            n += 1
    return n


def read_file(path, verify=False):
    """Yields the serialized Examples of one file (verify: check both checksums of every record)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError(f"{path}: truncated record header")
            (ln,), (c1,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            data = f.read(ln)
            tail = f.read(4)
            if len(data) < ln or len(tail) < 4:
                raise IOError(f"{path}: truncated record")
            if verify and (c1 != masked_crc32c(head[:8]) or struct.unpack("<I", tail)[0] != masked_crc32c(data)):
                raise IOError(f"{path}: record checksum mismatch")
            yield data


def count_records(path):
    n = 0
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if len(head) < 12:
                return n
            f.seek(struct.unpack("<Q", head[:8])[0] + 4, 1)
            n += 1
