"""Reader (and a minimal writer) for TensorFlow V2 checkpoints ("tensor bundles": `<prefix>.index` + `<prefix>.data-XXXXX-of-YYYYY`), without
TensorFlow -- so the reference's trained weights (`synthesize.py:32-40` restores `Text2Mel/*` from `logdir-1` and `SSRN/*`
from `logdir-2`) can feed `dc_tts_amd.engine.Engine` directly.

Formats implemented from their public specifications.  UNPINNED: TensorFlow is not installable here and no TF-written bundle is
reachable (no network), so this reader is exercised against bundles assembled by an independent encoder in tests/test_tf_checkpoint.py
(own CRC pinned to RFC 3720, shortened separator keys, restart arrays, several shards; a committed byte-exact fixture under
tests/golden/tf_bundle/) and, on the GPU, end to end: checkpoint directories -> load_reference_weights -> Engine -> synthesize.  What it does not implement fails with a named CheckpointError instead of mis-reading: compressed
(snappy) table blocks, partitioned (sliced) variables, big-endian bundles, missing shard files; multi-shard bundles are read.
  * `.index` is a LevelDB-style sorted string table (tensorflow/core/lib/io/table): 48-byte footer = metaindex BlockHandle,
    index BlockHandle (varint64 offset + size each), zero padding, magic 0xdb4775248b80fb57; every block is followed by a
    5-byte trailer (compression type, masked crc32c); block entries are (shared, non_shared, value_len) varint32 triples with
    prefix-compressed keys, then a restart array.  The index block maps to data blocks; data-block values are protobufs.
  * key "" -> BundleHeaderProto {num_shards = 1, endianness = 2, version = 3}; every other key is a variable name ->
    BundleEntryProto {dtype = 1, shape = 2 (TensorShapeProto: dim = 2 {size = 1}), shard_id = 3, offset = 4, size = 5,
    crc32c = 6 (fixed32), slices = 7}.  Tensor bytes are raw little-endian at [offset, offset + size) of the shard file.
"""
import os
import re
import struct
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}     # DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64


class CheckpointError(ValueError):
    pass


# ----------------------------------------------------------------------------- crc32c (Castagnoli), TF's masking
def _make_crc_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
        tab.append(c)
    return tab


_CRC_TABLE = _make_crc_table()
_CRC_TABLE16 = None


def _crc_scalar(data, c: int) -> int:
    for b in data:
        c = _CRC_TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c


def _apply(cols, c: int) -> int:
    """A GF(2)-linear map on the 32-bit register, given by the images `cols` of its 32 unit vectors."""
    r, k = 0, 0
    while c:
        if c & 1:
            r ^= cols[k]
        c >>= 1
        k += 1
    return r


def _zero_operator(n: int):
    """The register update for n zero bytes is linear: its 32 columns, by repeated squaring of the one-byte operator."""
    result = [1 << k for k in range(32)]
    power = [_crc_scalar(b"\0", 1 << k) for k in range(32)]
    while n:
        if n & 1:
            result = [_apply(power, col) for col in result]
        power = [_apply(power, col) for col in power]
        n >>= 1
    return result


def _crc_vector(data: bytes, c0: int) -> int:
    """update(c0, data) for a long buffer with numpy: update(c0, data) = Z_n(c0) ^ update(0, data) (the update is affine in the register),
    leading zero bytes leave a zero register unchanged, so the buffer is front-padded to K equal chunks whose registers advance together
    (one table gather per byte position, K lanes wide) and are then folded pairwise: update(0, a + b) = Z_len(b)(update(0, a)) ^ update(0, b)."""
    global _CRC_TABLE16
    if _CRC_TABLE16 is None:                                              # two bytes per step: T16[v] = the register v advanced past two zero bytes
        t8 = np.array(_CRC_TABLE, np.uint32)
        v = np.arange(65536, dtype=np.uint32)
        v = t8[v & 0xFF] ^ (v >> 8)
        _CRC_TABLE16 = t8[v & 0xFF] ^ (v >> 8)
    n = len(data)
    K = 1 << max(0, (n // 2048).bit_length() - 1)
    L = (-(-n // K) + 1) // 2 * 2
    buf = np.zeros(K * L, np.uint8)
    buf[K * L - n:] = np.frombuffer(data, np.uint8)
    cols = np.ascontiguousarray(buf.view("<u2").reshape(K, L // 2).T)     # byte pair j of every chunk, contiguous
    reg = np.zeros(K, np.uint32)
    for j in range(L // 2):
        reg = _CRC_TABLE16[(reg ^ cols[j]) & 0xFFFF] ^ (reg >> 16)
    op = _zero_operator(L)                                                # advance a register past one chunk
    while reg.size > 1:
        a, b = reg[0::2], reg[1::2]
        za = np.zeros_like(a)
        for k in range(32):
            za ^= np.where((a >> np.uint32(k)) & np.uint32(1), np.uint32(op[k]), np.uint32(0))
        reg = za ^ b
        op = [_apply(op, col) for col in op]                              # ... past two
    return _apply(_zero_operator(n), c0) ^ int(reg[0])


def crc32c(data: bytes, crc: int = 0) -> int:
    """CRC-32C (Castagnoli, reflected 0x82F63B78) of `data`, continuing from `crc`.  Long buffers (checkpoint tensors are up to 25 MB,
    a network with its Adam slots ~300 MB) take the vectorised path: ~115 MB/s instead of ~9 MB/s for the byte loop."""
    c0 = crc ^ 0xFFFFFFFF
    if len(data) < (1 << 16):
        return _crc_scalar(data, c0) ^ 0xFFFFFFFF
    return _crc_vector(bytes(data) if not isinstance(data, (bytes, bytearray, memoryview)) else data, c0) ^ 0xFFFFFFFF


def mask_crc(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


# ----------------------------------------------------------------------------- varints / protobuf wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result, shift = 0, 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]; pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise CheckpointError("varint too long")


def _proto_fields(buf: bytes) -> Iterable[Tuple[int, int, object]]:
    """Yield (field_number, wire_type, value) of one protobuf message (varint / fixed64 / bytes / fixed32)."""
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]; pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]; pos += 4
        else:
            raise CheckpointError(f"unsupported protobuf wire type {wt}")
        yield field, wt, v


def _parse_shape(buf: bytes) -> Tuple[int, ...]:
    dims = []
    for f, _, v in _proto_fields(buf):
        if f == 2:                                   # repeated Dim dim = 2
            size = 0
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1:
                    size = v2 if v2 < (1 << 63) else v2 - (1 << 64)
            dims.append(size)
    return tuple(dims)


def _parse_entry(buf: bytes) -> dict:
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for f, _, v in _proto_fields(buf):
        if f == 1: e["dtype"] = v
        elif f == 2: e["shape"] = _parse_shape(v)
        elif f == 3: e["shard_id"] = v
        elif f == 4: e["offset"] = v
        elif f == 5: e["size"] = v
        elif f == 6: e["crc32c"] = v
        elif f == 7: e["sliced"] = True
    return e


# ----------------------------------------------------------------------------- sorted string table (.index)
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
    if offset + size + 5 > len(data):
        raise CheckpointError("block handle outside the file")
    body, ctype = data[offset:offset + size], data[offset + size]
    if verify:
        stored = struct.unpack_from("<I", data, offset + size + 1)[0]
        if mask_crc(crc32c(data[offset:offset + size + 1])) != stored:
            raise CheckpointError("block checksum mismatch")
    if ctype != 0:
        kind = {1: "snappy"}.get(ctype, f"type {ctype}")
        raise CheckpointError(f"{kind}-compressed table block at offset {offset}: only uncompressed index blocks are supported "
                              "(TF's BundleWriter writes them uncompressed; re-save the checkpoint with a stock tf.train.Saver)")
    return body


def _block_entries(block: bytes) -> Iterable[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise CheckpointError("short block")
    nrestarts = struct.unpack_from("<I", block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * nrestarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]; pos += non_shared
        yield key, block[pos:pos + vlen]; pos += vlen


def read_index(path: str, verify: bool = True) -> Tuple[dict, Dict[str, dict]]:
    """Parse `<prefix>.index` -> (header dict, {variable name: entry dict})."""
    data = open(path, "rb").read()
    if len(data) < 48:
        raise CheckpointError("index file shorter than a table footer")
    footer = data[-48:]
    if struct.unpack_from("<Q", footer, 40)[0] != TABLE_MAGIC:
        raise CheckpointError("bad table magic: not a TF V2 checkpoint index")
    pos = 0
    _mo, pos = _varint(footer, pos); _ms, pos = _varint(footer, pos)
    io, pos = _varint(footer, pos); isz, pos = _varint(footer, pos)
    header, entries = {"num_shards": 1, "endianness": 0, "version": None}, {}
    for _k, handle in _block_entries(_read_block(data, io, isz, verify)):
        bo, p2 = _varint(handle, 0); bs, _ = _varint(handle, p2)
        for key, val in _block_entries(_read_block(data, bo, bs, verify)):
            if key == b"":
                for f, _, v in _proto_fields(val):
                    if f == 1: header["num_shards"] = v
                    elif f == 2: header["endianness"] = v
            else:
                entries[key.decode("utf-8")] = _parse_entry(val)
    if header["endianness"] != 0:
        raise CheckpointError("big-endian bundles are not supported")
    return header, entries


def read_checkpoint(prefix: str, names: Optional[Iterable[str]] = None, verify: bool = True,
                    verify_tensors: bool = True) -> Dict[str, np.ndarray]:
    """Load variables of the bundle `<prefix>` (e.g. logdir/LJ01-1/model_gs_800k) as numpy arrays.
    `verify` checks the index blocks' checksums; `verify_tensors` additionally checks every tensor's crc32c (numpy, ~115 MB/s:
    two seconds for the 200 MB of network weights; load_reference_weights leaves it off)."""
    header, entries = read_index(prefix + ".index", verify)
    want = set(names) if names is not None else None
    shards: Dict[int, bytes] = {}
    out: Dict[str, np.ndarray] = {}
    for name, e in entries.items():
        if want is not None and name not in want:
            continue
        if e["sliced"]:
            raise CheckpointError(f"{name}: partitioned (sliced) variable -- the bundle stores it as several slices "
                                  "(BundleEntryProto.slices); the synthesis path's variables are never partitioned, so this is not a dc_tts checkpoint")
        if e["dtype"] not in _DTYPES:
            raise CheckpointError(f"{name}: unsupported dtype enum {e['dtype']}")
        sid = e["shard_id"]
        if sid not in shards:
            if not 0 <= sid < header["num_shards"]:
                raise CheckpointError(f"{name}: shard_id {sid} outside the header's num_shards = {header['num_shards']}")
            shard_path = "%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"])
            if not os.path.exists(shard_path):
                raise CheckpointError(f"{name}: shard file {os.path.basename(shard_path)} is missing (multi-shard bundle: all "
                                      f"{header['num_shards']} .data-* files must sit next to the .index)")
            shards[sid] = open(shard_path, "rb").read()
        raw = shards[sid][e["offset"]:e["offset"] + e["size"]]
        dt = np.dtype(_DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"])) if e["shape"] else 1
        if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
            raise CheckpointError(f"{name}: size {e['size']} does not match shape {e['shape']}")
        if verify_tensors and e["crc32c"] is not None and mask_crc(crc32c(raw)) != e["crc32c"]:
            raise CheckpointError(f"{name}: tensor checksum mismatch")
        out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    if want is not None and want - set(out):
        raise CheckpointError(f"variables missing from {prefix}: {sorted(want - set(out))[:3]}")
    return out


def latest_checkpoint(logdir: str) -> str:
    """tf.train.latest_checkpoint: the `checkpoint` state file names the newest prefix (synthesize.py:34,40)."""
    state = os.path.join(logdir, "checkpoint")
    m = re.search(r'^model_checkpoint_path:\s*"(.*)"', open(state).read(), re.M)
    if not m:
        raise CheckpointError(f"no model_checkpoint_path in {state}")
    p = m.group(1)
    return p if os.path.isabs(p) else os.path.join(logdir, p)


def load_reference_weights(logdir: str, hp=None) -> Dict[str, np.ndarray]:
    """What synthesize.py:32-40 restores: `Text2Mel/*` trainable variables from `<logdir>-1`, `SSRN/*` from `<logdir>-2`
    (optimizer slots such as `.../Adam` are ignored).  Returns the dict `dc_tts_amd.engine.Engine` takes."""
    from .hyperparams import hp as _hp
    from .layers import variable_shapes
    from .weights import check_weights
    hp = hp or _hp
    spec = variable_shapes(hp)
    t2m = [n for n in spec if n.startswith("Text2Mel/")]
    ssrn = [n for n in spec if n.startswith("SSRN/")]
    W = read_checkpoint(latest_checkpoint(logdir + "-1"), t2m, verify_tensors=False)
    W.update(read_checkpoint(latest_checkpoint(logdir + "-2"), ssrn, verify_tensors=False))
    W = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in W.items()}
    check_weights(W, hp)
    return W


# ------------------------------------------------------------------------------------------------ writer
# What `sv.saver.save(sess, logdir + '/model_gs_...')` leaves behind (train.py:158), in the subset of the format the reader above
# understands: one shard, uncompressed table blocks, masked crc32c on every block and tensor.  UNPINNED like the reader: no TensorFlow
# here to read these files back; the round trip through read_checkpoint is what tests/test_tf_checkpoint.py checks.
def _vint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _pb(num: int, wire: int, payload: bytes) -> bytes:
    return _vint((num << 3) | wire) + payload


def _table_block(entries, restart_interval: int = 16) -> bytes:
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
    if not restarts:
        restarts = [0]
    for r in restarts:
        buf += struct.pack("<I", r)
    buf += struct.pack("<I", len(restarts))
    return bytes(buf)


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray], keys_per_block: int = 64) -> None:
    """Write `<prefix>.index` and `<prefix>.data-00000-of-00001` holding `tensors` (float32 / float64 / int32 / int64 arrays)."""
    enum = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}
    data, entries = bytearray(), []
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        if not a.flags.c_contiguous:                      # (np.ascontiguousarray would turn a scalar into shape (1,))
            a = a.copy()
        if a.dtype not in enum:
            raise CheckpointError(f"{name}: dtype {a.dtype} cannot be written")
        raw = a.tobytes()
        dims = b"".join(_pb(2, 2, _vint(len(d)) + d) for d in (_pb(1, 0, _vint(s)) for s in a.shape))
        e = _pb(1, 0, _vint(enum[a.dtype])) + _pb(2, 2, _vint(len(dims)) + dims) + _pb(3, 0, _vint(0)) + _pb(4, 0, _vint(len(data))) + \
            _pb(5, 0, _vint(len(raw))) + _pb(6, 5, struct.pack("<I", mask_crc(crc32c(raw))))
        entries.append((name.encode(), e))
        data += raw
    header = _pb(1, 0, _vint(1)) + _pb(2, 0, _vint(0)) + _pb(3, 2, _vint(2) + _pb(1, 0, _vint(1)))      # num_shards 1, little endian, version {producer 1}
    entries = [(b"", header)] + entries
    out, index = bytearray(), []

    def emit(block: bytes) -> bytes:
        off = len(out)
        out.extend(block)
        out.extend(b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00"))))
        return _vint(off) + _vint(len(block))

    for i in range(0, len(entries), keys_per_block):
        chunk = entries[i:i + keys_per_block]
        # index key of a block: >= its last key and < the first key of the next block (table format).  The last key itself is always legal;
        # last_key + b"\xff" is NOT when the next key extends the last one past a byte below 0xff ("X" + "\xff" > "X/Adam").
        index.append((chunk[-1][0], emit(_table_block(chunk))))
    footer = emit(_table_block([])) + emit(_table_block(index, restart_interval=1))
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))


def save_checkpoint(logdir: str, variables: Dict[str, np.ndarray], global_step: int, slots: Optional[Dict[str, Dict[str, np.ndarray]]] = None) -> str:
    """train.py:158  sv.saver.save(sess, logdir + '/model_gs_{}k'.format(gs // 1000)): the trainable variables, `gs/global_step`
    (train.py:82) and, when given, the optimizer slots under TensorFlow's names (`<variable>/Adam`, `<variable>/Adam_1`), plus the
    `checkpoint` state file tf.train.latest_checkpoint reads (synthesize.py:34,40).  Returns the prefix."""
    prefix = os.path.join(logdir, "model_gs_{}k".format(str(global_step // 1000).zfill(3)))
    t = {n: np.asarray(v) for n, v in variables.items()}
    t["gs/global_step"] = np.asarray(global_step, dtype=np.int32)
    for suffix, d in (slots or {}).items():
        for n, v in d.items():
            t[n + "/" + suffix] = np.asarray(v)
    if slots:
        # tf.train.AdamOptimizer's two non-slot variables: created as beta (not 1) and multiplied by beta once per update, so after
        # t = global_step updates they hold beta ** (t + 1) -- what a TF process that restores this file uses for the NEXT step's bias correction
        t["beta1_power"] = np.asarray(0.9 ** (global_step + 1), dtype=np.float32)
        t["beta2_power"] = np.asarray(0.999 ** (global_step + 1), dtype=np.float32)
    write_checkpoint(prefix, t)
    with open(os.path.join(logdir, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "{0}"\nall_model_checkpoint_paths: "{0}"\n'.format(os.path.basename(prefix)))
    return prefix
