"""TFRecord files of tf.train.Example rows without TensorFlow (mindrec_amd/tfrecord.py), the Criteo writer on top of it, and
compat's `mindspore.dataset.Schema` / `TFRecordDataset` reading them the way the reference's reader does
(models/wide_deep/src/datasets.py:226-271)."""
import os
import struct
import sys

import numpy as np
import pytest

from mindrec_amd import criteo, tfrecord


def test_crc32c_and_framing_known_answers(tmp_path):
    assert tfrecord.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    assert tfrecord.crc32c(b"") == 0
    # the masking TFRecord applies: rotate right by 15, add a constant
    c = tfrecord.crc32c(b"abc")
    assert tfrecord.masked_crc32c(b"abc") == (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    p = str(tmp_path / "one.tfrecord")
    assert tfrecord.write_file(p, [{"a": np.array([1, -2, 3], np.int64), "b": np.array([0.5, 1.5], np.float32), "c": b"xyz"}]) == 1
    raw = open(p, "rb").read()
    (ln,) = struct.unpack("<Q", raw[:8])
    assert len(raw) == 8 + 4 + ln + 4 and tfrecord.count_records(p) == 1
    (data,) = list(tfrecord.read_file(p, verify=True))
    ex = tfrecord.decode_example(data)
    assert ex["a"].tolist() == [1, -2, 3] and ex["a"].dtype == np.int64
    assert ex["b"].tolist() == [0.5, 1.5] and ex["b"].dtype == np.float32 and ex["c"] == [b"xyz"]
    # a hand-assembled Example in protobuf's wire format: features{feature{key:"x" value{int64_list{value:[7, 300]}}}}  (packed varints)
    inner = b"\x0a\x01x" + b"\x12\x07" + (b"\x1a\x05" + b"\x0a\x03" + b"\x07\xac\x02")
    msg = b"\x0a" + bytes([len(b"\x0a" + bytes([len(inner)]) + inner)]) + b"\x0a" + bytes([len(inner)]) + inner
    assert tfrecord.decode_example(msg)["x"].tolist() == [7, 300]
    # ... and an UNPACKED repeated field (one varint per tag), which older writers emit
    inner = b"\x0a\x01y" + b"\x12\x09" + (b"\x1a\x07" + b"\x08\x07" + b"\x08\xac\x02" + b"\x08\x00")
    msg = b"\x0a" + bytes([len(b"\x0a" + bytes([len(inner)]) + inner)]) + b"\x0a" + bytes([len(inner)]) + inner
    assert tfrecord.decode_example(msg)["y"].tolist() == [7, 300, 0]
    corrupt = bytearray(raw)
    corrupt[20] ^= 1
    open(p, "wb").write(bytes(corrupt))
    with pytest.raises(IOError):
        list(tfrecord.read_file(p, verify=True))


def test_criteo_tfrecords_through_the_mindspore_style_reader(tmp_path):
    compat = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "compat"))
    if compat not in sys.path:
        sys.path.insert(0, compat)
    import mindspore.common.dtype as mstype
    import mindspore.dataset as ds
    rng = np.random.default_rng(1)
    n, F, L = 5 * 20 + 7, 39, 20                                              # 5 whole records of 20 samples; the tail is dropped
    ids = rng.integers(0, 1000, size=(n, F)).astype(np.int32)
    wts = rng.random((n, F)).astype(np.float32)
    label = (rng.random(n) < 0.3).astype(np.float32)
    assert criteo.write_tfrecords(str(tmp_path), "train", ids, wts, label, records_per_file=2, line_per_sample=L) == 5
    files = sorted(os.path.join(str(tmp_path), f) for f in os.listdir(str(tmp_path)) if "train" in f and "tfrecord" in f)
    assert len(files) == 3
    schema = ds.Schema()
    schema.add_column("feat_ids", de_type=mstype.int32)
    schema.add_column("feat_vals", de_type=mstype.float32)
    schema.add_column("label", de_type=mstype.float32)
    d = ds.TFRecordDataset(dataset_files=files, shuffle=False, schema=schema, num_parallel_workers=8)
    assert d.get_dataset_size() == 5
    rows = list(d)
    assert rows[0][0].dtype == np.int32 and rows[0][1].dtype == np.float32
    assert np.array_equal(np.stack([r[0] for r in rows]).reshape(-1, F), ids[:100])
    assert np.array_equal(np.stack([r[1] for r in rows]).reshape(-1, F), wts[:100])
    assert np.array_equal(np.stack([r[2] for r in rows]).reshape(-1), label[:100])
    # batch of 2 records + the reference's kind of map: [B, 39], [B, 39], [B, 1]
    b = d.batch(2, drop_remainder=True).map(operations=lambda x, y, z: (np.array(x).reshape(2 * L, F), np.array(y).reshape(2 * L, F), np.array(z).reshape(2 * L, 1)),
                                           input_columns=["feat_ids", "feat_vals", "label"])
    out = list(b)
    assert len(out) == 2 and out[1][0].shape == (40, F) and np.array_equal(out[1][0], ids[40:80])
    # row shards with equal rows: 5 rows over 2 shards -> 2 each, row i to shard i mod 2
    a0 = list(ds.TFRecordDataset(files, schema=schema, shuffle=False, num_shards=2, shard_id=0, shard_equal_rows=True))
    a1 = list(ds.TFRecordDataset(files, schema=schema, shuffle=False, num_shards=2, shard_id=1, shard_equal_rows=True))
    assert len(a0) == len(a1) == 2 and np.array_equal(a0[1][2], label[40:60]) and np.array_equal(a1[0][2], label[20:40])
    # a seeded shuffle: the same permutation for the same seed and epoch, another one after reset()
    ds.config.set_seed(5)
    s1 = ds.TFRecordDataset(files, schema=schema, shuffle=True)
    first, again = [r[2][0] for r in s1], [r[2][0] for r in s1]
    assert first == again and sorted(np.concatenate([r[2] for r in s1]).tolist()) == sorted(label[:100].tolist())
    with pytest.raises(NotImplementedError, match="MindRecord"):
        ds.MindDataset(["x.mindrecord"], columns_list=["feat_ids"])
