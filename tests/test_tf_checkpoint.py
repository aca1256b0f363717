"""TF V2 checkpoint (tensor bundle) reader, exercised against bundles assembled by the INDEPENDENT encoder below: it shares no code with
dc_tts_amd/tf_checkpoint.py (own varints, own protobuf fields, own table blocks with prefix-compressed keys and restart arrays, own bit-serial
CRC-32C pinned to the RFC 3720 vectors, LevelDB-style SHORTENED separator keys in the index block, any number of shards).  A tiny bundle it
produced is committed under tests/golden/tf_bundle/ and must be reproduced byte for byte.  TensorFlow itself cannot be installed here, so no
TF-written file exists: see dc_tts_amd/tf_checkpoint.py for the format references.

STATUS: UNPINNED.  Everything below was written from the published TensorFlow sources as remembered, not from a file TensorFlow wrote; it stays that way until a
TF-written bundle exists.  Where each byte of the encoder comes from (TensorFlow source tree, r1.x):

| bytes written by `write_bundle` / `_block` | TensorFlow source it restates |
|---|---|
| file names `<prefix>.index`, `<prefix>.data-%05d-of-%05d` | `core/util/tensor_bundle/naming.cc` (`MetaFilename`, `DataFilename`) |
| key `""` -> header, every other key = variable name, keys in sorted order | `core/util/tensor_bundle/tensor_bundle.cc` (`kHeaderEntryKey`, `BundleWriter::Finish`: a `std::map` of entries fed to `table::TableBuilder`) |
| `BundleHeaderProto {num_shards = 1, endianness = 2 (LITTLE = 0 omitted), version = 3 {producer = 1}}` | `core/protobuf/tensor_bundle.proto`, `core/framework/versions.proto`, `tensor_bundle.cc: kTensorBundleVersion` |
| `BundleEntryProto {dtype = 1, shape = 2, shard_id = 3, offset = 4, size = 5, crc32c = 6 (fixed32), slices = 7}` | `core/protobuf/tensor_bundle.proto` |
| `TensorShapeProto {dim = 2 {size = 1}}`, `DT_FLOAT = 1, DT_INT32 = 3, DT_INT64 = 9` | `core/framework/tensor_shape.proto`, `core/framework/types.proto` |
| tensor bytes raw, little-endian, C order at `[offset, offset + size)` of the shard; `crc32c` = MASKED CRC-32C of exactly those bytes | `tensor_bundle.cc: WriteTensor / BundleWriter::Add` (`crc32c::Mask(crc32c::Value(...))`) |
| mask = rotate right by 15, add `0xa282ead8` | `core/lib/hash/crc32c.h` (`Mask`, `kMaskDelta`) |
| block = entries `(varint32 shared, varint32 non_shared, varint32 value_len, key suffix, value)`, then `fixed32` restart offsets, then `fixed32` restart count | `core/lib/io/block_builder.cc` (`BlockBuilder::Add / Finish`); a restart every `block_restart_interval` (default 16) entries: `core/lib/io/table_options.h` |
| block trailer = 1 byte compression type (`kNoCompression = 0`, `kSnappyCompression = 1`) + `fixed32` masked CRC-32C of block + type byte | `core/lib/io/table_builder.cc` (`WriteRawBlock`), `core/lib/io/format.h` (`kBlockTrailerSize = 5`) |
| index block: one entry per data block, key = a separator `>=` the block's last key and `<` the next block's first (shortened), value = `BlockHandle` | `table_builder.cc` (`TableBuilder::Add`: `FindShortestSeparator`; `Finish`: `FindShortSuccessor`), restart interval 1 for the index block |
| `BlockHandle` = `varint64 offset, varint64 size` (size excludes the trailer) | `core/lib/io/format.cc` (`BlockHandle::EncodeTo`) |
| footer = metaindex handle, index handle, zero padding to 40 bytes, `fixed64` magic `0xdb4775248b80fb57` (48 bytes) | `format.cc` (`Footer::EncodeTo`), `format.h` (`kTableMagicNumber`, `Footer::kEncodedLength`) |

What TensorFlow is free to choose -- data-block size (`table::Options::block_size`, hence keys per block), the restart interval, the number of shards, how far a
separator key is shortened -- is fuzzed below (`test_reader_under_parameters_a_writer_may_choose`)."""
import os
import struct

import numpy as np
import pytest

from dc_tts_amd import tf_checkpoint as C

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tf_bundle")
MAGIC = 0xDB4775248B80FB57


def crc32c_bits(data, crc=0):
    """CRC-32C (Castagnoli) bit by bit from the polynomial -- no table, nothing shared with the product's implementation."""
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


def masked(c):
    """tensorflow/core/lib/hash/crc32c.h Mask(): rotate right by 15, add a constant."""
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def shortest_separator(a, b):
    """leveldb BytewiseComparator::FindShortestSeparator(a, b): a key k with a <= k < b, as short as the rule allows (what a real table
    builder stores in the index block instead of the block's last key)."""
    n = min(len(a), len(b)); i = 0
    while i < n and a[i] == b[i]:
        i += 1
    if i < n and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def _vint(n):
    out = bytearray()
    while True:
        b = n & 0x7F; n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b); return bytes(out)


def _field(num, wt, payload):
    return _vint((num << 3) | wt) + payload


def _block(entries, restart_interval=4):
    """LevelDB block with prefix compression and restart points."""
    buf, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(buf))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        buf += _vint(shared) + _vint(len(k) - shared) + _vint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_bundle(prefix, tensors, keys_per_block=5, with_crc=True, num_shards=1, sliced=(), block_ctype=0, restart_interval=4, tensor_crc=None):
    """Minimal TF tensor-bundle writer: uncompressed table blocks, masked crc32c everywhere.  num_shards > 1 deals the tensors
    round-robin over shard files; `sliced` names get a BundleEntryProto.slices field; block_ctype != 0 marks the data blocks as
    compressed (the bytes stay raw: only the reader's diagnostic is exercised)."""
    datas = [bytearray() for _ in range(num_shards)]; entries = []
    dt_enum = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9}
    for i, name in enumerate(sorted(tensors)):
        a = np.asarray(tensors[name]); raw = a.tobytes()                  # (tobytes is C order; ascontiguousarray would make a scalar 1-d)
        sid = i % num_shards; data = datas[sid]
        shape = b"".join(_field(2, 2, _vint(len(d)) + d) for d in (_field(1, 0, _vint(s)) for s in a.shape))
        e = _field(1, 0, _vint(dt_enum[a.dtype])) + _field(2, 2, _vint(len(shape)) + shape) + _field(3, 0, _vint(sid)) + \
            _field(4, 0, _vint(len(data))) + _field(5, 0, _vint(len(raw)))
        if with_crc:
            e += _field(6, 5, struct.pack("<I", masked((tensor_crc or crc32c_bits)(raw))))
        if name in sliced:
            e += _field(7, 2, _vint(0))                     # repeated TensorSliceProto slices = 7 (an empty message is enough)
        entries.append((name.encode(), e)); data += raw
    header = _field(1, 0, _vint(num_shards)) + _field(2, 0, _vint(0)) + _field(3, 2, _vint(2) + _field(1, 0, _vint(1)))
    entries = [(b"", header)] + entries
    out = bytearray(); index = []
    def emit(block, ctype=0):
        off = len(out); out.extend(block); trailer = bytes([ctype])
        out.extend(trailer + struct.pack("<I", masked(crc32c_bits(block + trailer))))
        return _vint(off) + _vint(len(block))
    for i in range(0, len(entries), keys_per_block):
        chunk = entries[i:i + keys_per_block]
        nxt = entries[i + keys_per_block][0] if i + keys_per_block < len(entries) else None
        sep = shortest_separator(chunk[-1][0], nxt) if nxt is not None else chunk[-1][0] + b"\xff"     # (last block: FindShortSuccessor-like)
        index.append((sep, emit(_block(chunk, restart_interval), block_ctype)))
    meta = emit(_block([]))
    idx = emit(_block(index, restart_interval=1))
    footer = meta + idx
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", MAGIC)
    open(prefix + ".index", "wb").write(bytes(out))
    for sid, data in enumerate(datas):
        open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "wb").write(bytes(data))


def test_crc32c_known_answers():
    """RFC 3720 B.4 test vectors + the standard check value, for the product's CRC and for the test's own bit-serial one."""
    inc, dec = bytes(range(32)), bytes(range(31, -1, -1))
    for f in (C.crc32c, crc32c_bits):
        assert f(b"123456789") == 0xE3069283
        assert f(b"\x00" * 32) == 0x8A9136AA
        assert f(b"\xff" * 32) == 0x62A8AB43
        assert f(inc) == 0x46DD794E
        assert f(dec) == 0x113FDB5C
    assert masked(0) == C.mask_crc(0) == 0xA282EAD8 and masked(0xFFFFFFFF) == C.mask_crc(0xFFFFFFFF)
    rng = np.random.default_rng(5)
    for n in (1, 7, 64, 1000):
        a = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert C.crc32c(a) == crc32c_bits(a) and C.mask_crc(C.crc32c(a)) == masked(crc32c_bits(a))


def _fixture_tensors():
    """A miniature of the two-directory layout (synthesize.py:32-40) in ONE bundle: scoped names with shared prefixes that straddle block
    boundaries, optimizer slots that must be ignored by a name-restricted restore, a scalar, an int32 step, four dtypes, two shards."""
    rng = np.random.default_rng(2024)
    t = {}
    for i in (1, 2, 3, 10, 11):
        t[f"Text2Mel/AudioEnc/HC_{i}/conv1d/kernel"] = rng.standard_normal((3, 4, 8)).astype(np.float32)
        t[f"Text2Mel/AudioEnc/HC_{i}/conv1d/kernel/Adam"] = rng.standard_normal((3, 4, 8)).astype(np.float32)
        t[f"Text2Mel/AudioEnc/HC_{i}/conv1d/kernel/Adam_1"] = rng.standard_normal((3, 4, 8)).astype(np.float32)
        t[f"Text2Mel/AudioEnc/HC_{i}/H1/gamma"] = rng.standard_normal(4).astype(np.float32)
    t["SSRN/D_4/conv2d_transpose/kernel"] = rng.standard_normal((1, 3, 5, 5)).astype(np.float32)
    t["SSRN/C_16/normalize/beta"] = rng.standard_normal(1025).astype(np.float32)
    t["beta1_power"] = np.asarray(0.9 ** 3, np.float32)
    t["gs/global_step"] = np.asarray(2, np.int32)
    t["aux/int64"] = np.arange(-3, 4, dtype=np.int64)
    return t


def test_committed_bundle_fixture(tmp_path):
    """tests/golden/tf_bundle/model.{index,data-0000?-of-00002}: assembled by the independent encoder above (3 keys per block -> an eight-block
    index with shortened separator keys, restart interval 2, two shards, bit-serial CRCs), committed, reproduced here byte for byte, and read by
    the product's reader -- every tensor, every checksum."""
    t = _fixture_tensors()
    p = str(tmp_path / "model")
    write_bundle(p, t, keys_per_block=3, num_shards=2, restart_interval=2)
    names = ["model.index", "model.data-00000-of-00002", "model.data-00001-of-00002"]
    if os.environ.get("DCTTS_WRITE_FIXTURES"):
        os.makedirs(GOLD, exist_ok=True)
        for n in names:
            open(os.path.join(GOLD, n), "wb").write(open(os.path.join(str(tmp_path), n), "rb").read())
    for n in names:
        assert open(os.path.join(GOLD, n), "rb").read() == open(os.path.join(str(tmp_path), n), "rb").read(), n
    header, entries = C.read_index(os.path.join(GOLD, "model.index"))
    assert header["num_shards"] == 2 and set(entries) == set(t) and {e["shard_id"] for e in entries.values()} == {0, 1}
    got = C.read_checkpoint(os.path.join(GOLD, "model"), verify=True, verify_tensors=True)
    assert set(got) == set(t)
    for k in t:
        assert got[k].dtype == t[k].dtype and got[k].shape == t[k].shape and np.array_equal(got[k], t[k]), k
    only = C.read_checkpoint(os.path.join(GOLD, "model"), [k for k in t if k.startswith("Text2Mel/") and "Adam" not in k])
    assert len(only) == 10


def test_round_trip_and_errors(tmp_path):
    rng = np.random.default_rng(0)
    T = {"Text2Mel/TextEnc/embed_1/lookup_table": rng.standard_normal((32, 128)).astype(np.float32),
         "Text2Mel/TextEnc/C_2/conv1d/kernel": rng.standard_normal((1, 128, 512)).astype(np.float32),
         "Text2Mel/TextEnc/C_2/conv1d/kernel/Adam": np.zeros((1, 128, 512), np.float32),
         "Text2Mel/TextEnc/C_2/conv1d/bias": rng.standard_normal(512).astype(np.float32),
         "gs/global_step": np.array(800000, np.int32),
         "SSRN/D_4/conv2d_transpose/kernel": rng.standard_normal((1, 3, 16, 16)).astype(np.float32)}
    prefix = str(tmp_path / "model_gs_800k")
    write_bundle(prefix, T)
    header, entries = C.read_index(prefix + ".index")
    assert header["num_shards"] == 1 and set(entries) == set(T)
    assert entries["SSRN/D_4/conv2d_transpose/kernel"]["shape"] == (1, 3, 16, 16)
    got = C.read_checkpoint(prefix)
    for k in T:
        np.testing.assert_array_equal(got[k], T[k])
    sub = C.read_checkpoint(prefix, ["Text2Mel/TextEnc/C_2/conv1d/bias"])
    assert list(sub) == ["Text2Mel/TextEnc/C_2/conv1d/bias"]
    with pytest.raises(C.CheckpointError):
        C.read_checkpoint(prefix, ["no/such/variable"])
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); raw[10] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(C.CheckpointError):                  # corrupted tensor bytes: crc32c mismatch
        C.read_checkpoint(prefix)
    (tmp_path / "checkpoint").write_text('model_checkpoint_path: "model_gs_800k"\nall_model_checkpoint_paths: "model_gs_800k"\n')
    assert C.latest_checkpoint(str(tmp_path)) == prefix


def test_load_reference_weights_layout(tmp_path, weights):
    """A full synthetic 'trained' checkpoint pair laid out like hp.logdir-1 / hp.logdir-2 -> the Engine's weight dict."""
    from dc_tts_amd.hyperparams import hp
    logdir = str(tmp_path / "LJ01")
    for suffix, scope in (("-1", "Text2Mel/"), ("-2", "SSRN/")):
        d = logdir + suffix; os.makedirs(d)
        T = {k: v for k, v in weights.items() if k.startswith(scope)}
        T[next(iter(T)) + "/Adam_1"] = np.zeros(3, np.float32)           # optimizer slot: must be ignored
        T["gs/global_step"] = np.array(1, np.int32)
        write_bundle(os.path.join(d, "model_gs_1k"), T, keys_per_block=16, with_crc=False)
        open(os.path.join(d, "checkpoint"), "w").write('model_checkpoint_path: "model_gs_1k"\n')
    W = C.load_reference_weights(logdir, hp)
    assert set(W) == set(weights)
    for k in weights:
        np.testing.assert_array_equal(W[k], weights[k])


def test_unsupported_layouts_fail_with_named_errors(tmp_path):
    """What a real TF checkpoint may contain beyond the plain single-shard bundle: multi-shard bundles are read; compressed index
    blocks, partitioned variables and missing shard files raise CheckpointError naming the problem (never a silent mis-read)."""
    rng = np.random.default_rng(1)
    T = {f"SSRN/C_{i}/conv1d/bias": rng.standard_normal(8 + i).astype(np.float32) for i in range(1, 8)}
    p = str(tmp_path / "multi")
    write_bundle(p, T, num_shards=3)
    got = C.read_checkpoint(p)
    for k in T:
        np.testing.assert_array_equal(got[k], T[k])
    os.remove(p + ".data-00001-of-00003")
    with pytest.raises(C.CheckpointError, match="shard file .* is missing"):
        C.read_checkpoint(p)
    p = str(tmp_path / "snappy")
    write_bundle(p, T, block_ctype=1)
    with pytest.raises(C.CheckpointError, match="snappy-compressed"):
        C.read_checkpoint(p)
    p = str(tmp_path / "sliced")
    write_bundle(p, T, sliced={"SSRN/C_3/conv1d/bias"})
    with pytest.raises(C.CheckpointError, match="partitioned"):
        C.read_checkpoint(p)
    assert set(C.read_checkpoint(p, ["SSRN/C_1/conv1d/bias"])) == {"SSRN/C_1/conv1d/bias"}     # untouched variables still load


def test_product_writer_round_trip_and_training_checkpoints(tmp_path, weights):
    """dc_tts_amd.tf_checkpoint.write_checkpoint / save_checkpoint (what a training run leaves behind, train.py:158) read back by the
    reader, incl. the logdir-1 / logdir-2 layout synthesize.py:32-40 restores from."""
    rng = np.random.default_rng(3)
    t = {"a/b": rng.normal(size=(3, 5)).astype(np.float32), "a/c": np.arange(7, dtype=np.int64), "z": np.float32(2.5).reshape(()),
         "m/n/o": rng.normal(size=(2, 3, 4)).astype(np.float64)}
    C.write_checkpoint(str(tmp_path / "x" / "model"), t, keys_per_block=2)
    back = C.read_checkpoint(str(tmp_path / "x" / "model"))
    assert set(back) == set(t) and all(np.array_equal(back[n], t[n]) and back[n].dtype == t[n].dtype for n in t)
    logdir = str(tmp_path / "logdir" / "LJ01")
    t2m = {n: v for n, v in weights.items() if n.startswith("Text2Mel/")}
    ssrn = {n: v for n, v in weights.items() if n.startswith("SSRN/")}
    p1 = C.save_checkpoint(logdir + "-1", t2m, 12345, {"Adam": {n: np.zeros_like(v) for n, v in t2m.items()}})
    C.save_checkpoint(logdir + "-2", ssrn, 200000)
    assert os.path.basename(p1) == "model_gs_012k" and C.latest_checkpoint(logdir + "-1") == p1
    assert int(C.read_checkpoint(p1, ["gs/global_step"])["gs/global_step"]) == 12345
    W = C.load_reference_weights(logdir)
    assert set(W) == set(weights) and all(np.array_equal(W[n], weights[n]) for n in weights)


def test_crc32c_vector_path_equals_byte_loop():
    """Buffers of 64 KB and more take the numpy path (chunks advanced together, folded with the zero-byte operator)."""
    from dc_tts_amd import tf_checkpoint as C
    assert C.crc32c(b"123456789") == 0xE3069283                           # the CRC-32C check value
    rng = np.random.default_rng(11)
    for n in (65536, 65537, 99991, 1 << 18, (1 << 20) + 3):
        a = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        ref = C._crc_scalar(a, 0xFFFFFFFF) ^ 0xFFFFFFFF
        assert C.crc32c(a) == ref
        k = n // 3
        assert C.crc32c(a[k:], C.crc32c(a[:k])) == ref                    # continuation across the two paths
    z = bytes(200000)
    assert C.crc32c(z) == C._crc_scalar(z, 0xFFFFFFFF) ^ 0xFFFFFFFF


def test_index_keys_bound_their_blocks_for_a_seeking_reader(tmp_path):
    """Table format: the index key of a data block is >= the block's last key and < the next block's first key, so that a reader that SEEKS
    (TensorFlow's BundleReader) lands in the right block.  With Adam slots the keys 'X' < 'X/Adam' < 'X/Adam_1' straddle block boundaries,
    where 'last key + 0xff' would overshoot ('X\\xff' > 'X/Adam'); the writer must stay legal there, and store beta1_power / beta2_power = beta ** (step + 1)."""
    import struct
    from dc_tts_amd import tf_checkpoint as C
    rng = np.random.default_rng(0)
    names = [f"SSRN/C_{i}/conv1d/kernel" for i in range(1, 30)]
    var = {n: rng.standard_normal((3, 2)).astype(np.float32) for n in names}
    slots = {"Adam": {n: np.zeros((3, 2), np.float32) for n in names}, "Adam_1": {n: np.ones((3, 2), np.float32) for n in names}}
    prefix = C.save_checkpoint(str(tmp_path), var, 2000, slots)
    data = open(prefix + ".index", "rb").read()
    footer = data[-48:]
    pos = 0
    _, pos = C._varint(footer, pos); _, pos = C._varint(footer, pos)
    io, pos = C._varint(footer, pos); isz, pos = C._varint(footer, pos)
    blocks = []
    for ikey, handle in C._block_entries(C._read_block(data, io, isz, True)):
        bo, p2 = C._varint(handle, 0); bs, _ = C._varint(handle, p2)
        keys = [k for k, _ in C._block_entries(C._read_block(data, bo, bs, True))]
        blocks.append((ikey, keys))
    # tiny blocks so that every kind of boundary occurs
    C.write_checkpoint(str(tmp_path / "small"), {**var, **{n + "/Adam": v for n, v in slots["Adam"].items()}, **{n + "/Adam_1": v for n, v in slots["Adam_1"].items()}}, keys_per_block=2)
    data2 = open(str(tmp_path / "small") + ".index", "rb").read()
    f2 = data2[-48:]; pos = 0
    _, pos = C._varint(f2, pos); _, pos = C._varint(f2, pos)
    io, pos = C._varint(f2, pos); isz, pos = C._varint(f2, pos)
    for ikey, handle in C._block_entries(C._read_block(data2, io, isz, True)):
        bo, p2 = C._varint(handle, 0); bs, _ = C._varint(handle, p2)
        blocks.append((ikey, [k for k, _ in C._block_entries(C._read_block(data2, bo, bs, True))]))
    assert len(blocks) > 40
    prev_file_last = None
    for i, (ikey, keys) in enumerate(blocks):
        assert keys == sorted(keys) and ikey >= keys[-1]
        if i + 1 < len(blocks) and blocks[i + 1][1][0] > keys[-1]:          # same file: the next block's first key
            assert ikey < blocks[i + 1][1][0], (ikey, blocks[i + 1][1][0])
    t = C.read_checkpoint(prefix)
    # tf.train.AdamOptimizer creates beta1_power = beta1 and multiplies it once per update: after t updates it holds beta1 ** (t + 1)
    assert abs(float(t["beta1_power"]) - 0.9 ** 2001) < 1e-12 and abs(float(t["beta2_power"]) - 0.999 ** 2001) < 1e-6
    assert int(t["gs/global_step"]) == 2000


def test_reader_under_parameters_a_writer_may_choose(tmp_path):
    """Hypothesis fuzz over what a conforming writer is free to choose (block size -> keys per block, restart interval, shard count, variable names with long
    shared prefixes -> separator shortening, scalar / empty-dimension / int tensors): the reader returns every tensor bit for bit, whole and by name."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import HealthCheck, given, settings
    from hypothesis import strategies as st
    name_part = st.sampled_from(["Text2Mel", "SSRN", "TextEnc", "AudioDec", "HC_1", "HC_10", "HC_11", "C_2", "conv1d", "kernel", "bias", "gamma", "beta", "Adam", "Adam_1", "a", "b", "z" * 9])
    names = st.lists(st.lists(name_part, min_size=1, max_size=4).map("/".join), min_size=1, max_size=14, unique=True)
    shape = st.lists(st.integers(0, 5), min_size=0, max_size=3)
    case = {"n": 0}

    @settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])
    @given(names=names, shapes=st.lists(shape, min_size=14, max_size=14), kinds=st.lists(st.integers(0, 2), min_size=14, max_size=14),
           kpb=st.integers(1, 9), ri=st.integers(1, 17), shards=st.integers(1, 4), seed=st.integers(0, 2 ** 31 - 1))
    def run(names, shapes, kinds, kpb, ri, shards, seed):
        rng = np.random.default_rng(seed)
        tensors = {}
        for i, n in enumerate(names):
            dt = (np.float32, np.int32, np.int64)[kinds[i]]
            a = rng.normal(size=shapes[i]).astype(np.float32) if dt is np.float32 else rng.integers(-9, 9, size=shapes[i]).astype(dt)
            tensors[n] = a
        case["n"] += 1
        prefix = str(tmp_path / f"fz{case['n']}")
        write_bundle(prefix, tensors, keys_per_block=kpb, restart_interval=ri, num_shards=shards)
        got = C.read_checkpoint(prefix)
        assert sorted(got) == sorted(tensors)
        for n, a in tensors.items():
            assert got[n].dtype == a.dtype and got[n].shape == a.shape and got[n].tobytes() == a.tobytes(), n
        pick = sorted(tensors)[:: max(1, len(tensors) // 3)]
        sub = C.read_checkpoint(prefix, names=pick)
        assert sorted(sub) == pick and all(sub[n].tobytes() == tensors[n].tobytes() for n in pick)
        for f in os.listdir(tmp_path):
            if f.startswith(f"fz{case['n']}."):
                os.remove(tmp_path / f)

    run()
