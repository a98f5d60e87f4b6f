"""The zstd block encoder (longtail_amd/csrc/zstd_block_core.h) instantiated on the host (oracle/zstd_model.c) against the
REFERENCE decoder: ZStdCompressionAPI_Decompress -> ZSTD_decompressDCtx (lib/zstd/longtail_zstd.c:144-177).  The same
source runs one wavefront per piece on the GPU (k_zstd.hip) and must give these bytes (tests/test_gpu_codecs.py), so
this file pins the format work -- literals section / Huffman tree description / 4 streams, NCount, FSE tables and the
sequence bit-stream -- without a GPU.  Needs oracle/_ref (built by `make -C oracle` where /root/reference exists)."""
import ctypes as C

import numpy as np
import pytest

from tests._libs import have_ref, oracle as get_oracle, ref as get_ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")

CODEC_ZSTD = 1


@pytest.fixture(scope="module")
def model():
    o = get_oracle()
    d = o.dll
    d.ltz_model_bound.restype = C.c_size_t
    d.ltz_model_bound.argtypes = [C.c_size_t]
    d.ltz_model_compress.restype = C.c_int
    d.ltz_model_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    d.ltz_model_encode_block.restype = C.c_uint32
    d.ltz_model_encode_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    d.ltz_model_debug.restype = None
    d.ltz_model_debug.argtypes = [C.c_uint32]
    d.ltz_model_sub_blocks.restype = None
    d.ltz_model_sub_blocks.argtypes = [C.c_int]
    return o


@pytest.fixture(params=[0, 1], ids=["pieces", "subblocks"], autouse=True)
def layout(request, model):
    """both layouts of a 128 KiB piece: one block (zb_encode_block) / one block per 4 KiB unit (zb_encode_piece_sub)"""
    model.dll.ltz_model_sub_blocks(request.param)
    yield request.param
    model.dll.ltz_model_sub_blocks(0)


def compress(o, b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(b, dtype=np.uint8)
    cap = o.dll.ltz_model_bound(len(b))
    out = np.zeros(cap + 8, np.uint8)
    n = C.c_size_t(0)
    assert o.dll.ltz_model_compress(b.ctypes.data, len(b), out.ctypes.data, cap, C.byref(n)) == 0
    return out[: n.value].copy()


def roundtrip(o, b: np.ndarray) -> int:
    c = compress(o, b)
    err, out = get_ref().decompress(CODEC_ZSTD, c, len(b))
    assert err == 0 and len(out) == len(b) and (out == b).all()
    return len(c)


@pytest.mark.parametrize("kind", [0, 1, 2, 11, 12, 13])
@pytest.mark.parametrize("n", [0, 1, 100, 255, 256, 257, 5000, 131071, 131072, 131073, 400000])
def test_synthetic_kinds_decode_with_reference(model, kind, n):
    roundtrip(model, model.synth(n, 1000 + n + kind, kind))


def test_compressible_data_gets_smaller_than_lz4(model):
    b = model.synth(2 << 20, 5, 1)
    assert roundtrip(model, b) < len(model.lz4_compress(b))


def _alphabets():
    rng = np.random.default_rng(11)
    yield "two symbols", rng.choice(2, 3000, p=[0.9, 0.1]).astype(np.uint8)
    yield "deep tree (length limit)", (rng.geometric(0.5, 200000) - 1).clip(0, 40).astype(np.uint8)
    yield "128 symbols flat (direct weights, all equal)", rng.integers(0, 128, 100000).astype(np.uint8)
    yield "129 symbols (FSE-compressed weights)", rng.integers(0, 129, 100000).astype(np.uint8)
    yield "high bytes only", (rng.geometric(0.3, 50000).clip(1, 30) + 200).astype(np.uint8)
    yield "gaussian bytes", (np.abs(rng.normal(128, 20, 300000)).astype(np.int64) % 256).astype(np.uint8)
    yield "nibbles then noise", np.concatenate([rng.integers(0, 16, 100000), rng.integers(0, 256, 100000)]).astype(np.uint8)
    yield "text", np.frombuffer((b"the quick brown fox jumps over the lazy dog. " * 4000), np.uint8).copy()
    t = rng.integers(0, 256, (4096, 8)).astype(np.uint8)
    yield "vocabulary of 8-byte tokens", t[rng.integers(0, 4096, 60000)].reshape(-1)
    yield "long literal runs and long matches", np.concatenate([rng.integers(0, 256, 70000).astype(np.uint8)] * 3)


@pytest.mark.parametrize("name,data", list(_alphabets()), ids=[n for n, _ in _alphabets()])
def test_literal_and_sequence_shapes(model, name, data):
    roundtrip(model, data)


@pytest.mark.parametrize("flags", [1, 2, 4, 6])
def test_every_table_mode(model, flags):
    """1: raw literals; 2: never the predefined distributions; 4: only predefined / RLE."""
    model.dll.ltz_model_debug(flags)
    try:
        for kind, n in [(1, 300000), (11, 5000), (12, 70000), (13, 131072)]:
            roundtrip(model, model.synth(n, 9 + n, kind))
    finally:
        model.dll.ltz_model_debug(0)


def test_encode_block_from_records_maximum_sequence_count(model, layout):
    """32 units x 1024 sequences of 4 bytes (one block: the 3-byte Number_of_Sequences form, >= 0x7F00; sub-blocks: 1024 each)."""
    nunits = 32
    meta = np.zeros((nunits, 4), np.uint32)
    lits = np.zeros((nunits, 4096), np.uint8)
    recs = np.zeros((nunits, 1024), np.uint64)
    pat = np.array([1, 2, 3, 4], np.uint8)
    for u in range(nunits):
        first_lit = 4 if u == 0 else 0
        nseq = 1023 if u == 0 else 1024
        recs[u, :nseq] = 4 << 16 | 4 << 32
        recs[u, 0] |= first_lit
        lits[u, :first_lit] = pat[:first_lit]
        meta[u] = (nseq, first_lit, 0, 0)
    out = np.zeros(140000, np.uint8)
    n = model.dll.ltz_model_encode_block(meta.ctypes.data, lits.ctypes.data, recs.ctypes.data, nunits, 131072, out.ctypes.data)
    assert 0 < n < 131072
    raw = np.tile(pat, 32768)
    head = np.frombuffer(bytes([0x28, 0xB5, 0x2F, 0xFD, 0xE0]) + (131072).to_bytes(8, "little"), np.uint8)
    if layout:  # the sub-blocks bring their headers; Last_Block goes on the last one
        model.dll.ltz_model_last_sub.restype = C.POINTER(C.c_uint16)
        sub = model.dll.ltz_model_last_sub()
        out[n - 3 - (sub[nunits - 1] & 0x7FFF)] |= 1
        frame = np.concatenate([head, out[:n]])
    else:
        frame = np.concatenate([head, np.frombuffer((1 | 2 << 1 | n << 3).to_bytes(3, "little"), np.uint8), out[:n]])
    err, dec = get_ref().decompress(CODEC_ZSTD, frame, len(raw))
    assert err == 0 and (dec == raw).all()


# ---- decoder (zstd_decode_core.h, host instantiation) ----
def model_decode(o, frame: np.ndarray, cap: int):
    o.dll.ltz_model_decompress.restype = C.c_int
    o.dll.ltz_model_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    out = np.zeros(cap + 8, np.uint8)
    n = C.c_size_t(0)
    err = o.dll.ltz_model_decompress(frame.ctypes.data, len(frame), out.ctypes.data, cap, C.byref(n))
    return err, out[: n.value]


@pytest.mark.parametrize("setting", range(5))
def test_decoder_reads_reference_encoder_output(model, setting):
    """Frames from the reference's ZSTD_compressCCtx at longtail's five settings (levels 3, 3, 22, 8, 22): Huffman with
    direct and FSE-coded weights, treeless blocks, all FSE modes incl. Repeat, repeat offsets, 1- and 4-stream literals."""
    r = get_ref()
    st = r.dll.refh_zstd_type(setting)
    rng = np.random.default_rng(setting)
    datas = [model.synth(n, 5 + n + k, k) for k in (0, 1, 2, 11, 12, 13) for n in (0, 1, 100, 5000, 131072, 131073, 400000, 3 << 20)]
    datas.append((np.abs(rng.normal(128, 20, 700000)).astype(np.int64) % 256).astype(np.uint8))
    datas.append(np.frombuffer(b"the quick brown fox jumps over the lazy dog. " * 9000, np.uint8).copy())
    for d in datas:
        err, out = model_decode(model, r.compress(1, st, d), len(d))
        assert err == 0 and len(out) == len(d) and (out == d).all()


def test_decoder_reads_own_encoder_and_concatenated_frames(model):
    a, b = model.synth(300000, 1, 1), model.synth(70000, 2, 12)
    fa, fb = compress(model, a), compress(model, b)
    skippable = np.frombuffer(bytes([0x5A, 0x2A, 0x4D, 0x18, 2, 0, 0, 0, 9, 9]), np.uint8)
    err, out = model_decode(model, np.concatenate([fa, skippable, fb]), len(a) + len(b))
    assert err == 0 and (out == np.concatenate([a, b])).all()
    assert model_decode(model, fa, len(a) - 1)[0] != 0  # destination too small
    assert model_decode(model, fa[:-1].copy(), len(a))[0] != 0  # truncated


def test_decoder_never_accepts_what_the_reference_rejects(model):
    """Differential fuzz on mutated reference frames: whenever the model accepts, the reference accepts with identical
    bytes.  (The converse does not hold: the reference's fast 4-stream Huffman loop does not verify that each literal
    stream is consumed exactly, huf_decompress.c:872-887, RFC 8878 §4.2.2 says it must be.)"""
    r = get_ref()
    rng = np.random.default_rng(9)
    both = 0
    for kind, n in ((1, 200000), (11, 30000), (12, 100000), (13, 60000), (1, 3000)):
        b = model.synth(n, 31 + n, kind)
        for w in (0, 2):
            c = r.compress(1, r.dll.refh_zstd_type(w), b)
            for _ in range(150):
                x = c.copy()
                if rng.integers(0, 4) == 0:
                    x = x[: rng.integers(0, len(x) + 1)].copy()
                else:
                    for _ in range(int(rng.integers(1, 4))):
                        x[rng.integers(0, len(x))] ^= np.uint8(1 << rng.integers(0, 8))
                e2, o2 = model_decode(model, x, n)
                if e2 == 0:
                    e1, o1 = r.decompress(1, x, n)
                    assert e1 == 0 and len(o1) == len(o2) and (o1 == o2).all()
                    both += 1
    assert both > 100


def test_cooperative_fse_table_builder_equals_serial(model):
    """zd_build_fse_par (all lanes; the sequence tables of k_zstd_prepare / k_zstd_decode) against zd_build_fse (the construction of
    zstd_decompress_block.c:484-603) on random normalised distributions with low-probability (-1) and absent symbols."""
    d = model.dll
    d.ltz_model_fse_tables.restype = C.c_int
    d.ltz_model_fse_tables.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(77)
    for it in range(600):
        tl = int(rng.integers(5, 10))
        size = 1 << tl
        nsym = int(rng.integers(1, min(53, size) + 1))
        nlow = int(rng.integers(0, min(nsym, 12))) if it % 3 else 0
        npos = int(rng.integers(1, nsym - nlow + 1)) if nsym > nlow else 0
        if npos == 0:
            nlow, npos = nsym - 1, 1
        rest = size - nlow
        cuts = np.sort(rng.choice(np.arange(1, rest), npos - 1, replace=False)) if npos > 1 else np.zeros(0, np.int64)
        counts = np.diff(np.concatenate(([0], cuts, [rest])))
        vals = np.concatenate((counts, -np.ones(nlow, np.int64), np.zeros(nsym - nlow - npos, np.int64)))
        rng.shuffle(vals)
        if vals[-1] == 0:  # the last symbol is present by definition of maxsym
            k = int(np.flatnonzero(vals)[0])
            vals[-1], vals[k] = vals[k], 0
        norm = vals.astype(np.int16)
        a = np.zeros(4 * size, np.uint8)
        b = np.ones(4 * size, np.uint8)
        r = d.ltz_model_fse_tables(norm.ctypes.data, nsym - 1, tl, a.ctypes.data, b.ctypes.data)
        assert r == 0, (it, tl, norm)
        assert np.array_equal(a, b), (it, tl, norm)
    # distributions that do not fill the table are rejected by both
    norm = np.array([10, 10, 3], np.int16)
    a = np.zeros(4 * 32, np.uint8)
    assert d.ltz_model_fse_tables(norm.ctypes.data, 2, 5, a.ctypes.data, a.ctypes.data) == 3


def _rep_piece(seed, nunits=8):
    """Sequences whose offsets repeat in every way the format has a code for -- the last, the one before, the third, the last minus
    one, with and without literals in front -- executed into the raw bytes they describe; per 4 KiB unit as the match finder delivers
    them (records {literals | match length << 16 | offset << 32}, the unit's literal bytes, meta {nseq, nlit, tail, 0})."""
    rng = np.random.default_rng(seed)
    raw = np.zeros(nunits * 4096, np.uint8)
    meta = np.zeros((nunits, 4), np.uint32)
    lits = np.zeros((nunits, 4096), np.uint8)
    recs = np.zeros((nunits, 1024), np.uint64)
    pos = 0
    for u in range(nunits):
        end, nlit, nseq, hist = (u + 1) * 4096, 0, 0, []
        while True:
            lit = int(rng.choice([0, 0, 0, 1, 2, 5, 17]))
            ml = int(rng.integers(3, 40))
            if pos + lit + ml > end - 8 or nseq == 1024:
                break
            raw[pos : pos + lit] = rng.integers(0, 256, lit)
            lits[u, nlit : nlit + lit] = raw[pos : pos + lit]
            pos += lit
            nlit += lit
            pick = int(rng.integers(0, 6))
            off = None
            if hist and pick < 5:
                off = [hist[-1], hist[-2] if len(hist) > 1 else hist[-1], hist[-3] if len(hist) > 2 else hist[-1], hist[-1] - 1, hist[-1]][pick]
            if not off or off < 1 or off > pos:
                off = int(rng.integers(1, pos + 1)) if pos else None
            if off is None:  # nothing to copy from yet: literals only
                raw[pos : pos + ml] = rng.integers(0, 256, ml)
                lits[u, nlit : nlit + ml] = raw[pos : pos + ml]
                pos += ml
                nlit += ml
                continue
            for k in range(ml):
                raw[pos + k] = raw[pos + k - off]
            pos += ml
            # the literals in front of this match are all the literal bytes since the last match
            recs[u, nseq] = (nlit - int(sum(int(r & 0xFFFF) for r in recs[u, :nseq]))) | (ml << 16) | (off << 32)
            nseq += 1
            hist.append(off)
        tail = end - pos
        raw[pos:end] = rng.integers(0, 256, tail)
        lits[u, nlit : nlit + tail] = raw[pos:end]
        nlit += tail
        pos = end
        meta[u] = (nseq, nlit, tail, 0)
    return raw, meta, lits, recs


@pytest.mark.parametrize("seed", range(6))
def test_block_local_repeat_offset_codes(model, layout, seed):
    """Sub-block layout: repeat-offset codes for history entries set inside the block (zb_encode_piece_sub, ZB_F_REPCODES).  The
    reference decoder (which carries the real history across blocks) regenerates the bytes, and the codes pay: the piece is smaller
    than with plain offsets."""
    if not layout:
        pytest.skip("the one-block-per-piece layout writes plain offsets")
    d = model.dll
    d.ltz_model_flags.restype = None
    d.ltz_model_flags.argtypes = [C.c_uint32]
    raw, meta, lits, recs = _rep_piece(seed)
    nunits = len(meta)
    head = np.frombuffer(bytes([0x28, 0xB5, 0x2F, 0xFD, 0xE0]) + (len(raw)).to_bytes(8, "little"), np.uint8)
    sizes = {}
    try:
        for flags in (1, 0):
            d.ltz_model_flags(flags)
            out = np.zeros(140000, np.uint8)
            n = d.ltz_model_encode_block(meta.ctypes.data, lits.ctypes.data, recs.ctypes.data, nunits, len(raw), out.ctypes.data)
            assert 0 < n < len(raw)
            d.ltz_model_last_sub.restype = C.POINTER(C.c_uint16)
            sub = d.ltz_model_last_sub()
            out[n - 3 - (sub[nunits - 1] & 0x7FFF)] |= 1
            frame = np.concatenate([head, out[:n]])
            err, dec = get_ref().decompress(CODEC_ZSTD, frame, len(raw))
            assert err == 0 and len(dec) == len(raw) and (dec == raw).all(), flags
            err2, got = model_decode(model, frame, len(raw))  # the decoder core of this library (serial instantiation)
            assert err2 == 0 and len(got) == len(raw) and (got == raw).all()
            sizes[flags] = n
    finally:
        d.ltz_model_flags(0)
    assert sizes[1] < sizes[0], sizes


def _builder_corners():
    """the statistics of tests/test_gpu_codecs.py::test_zstd_table_builders_all_lanes_equal_the_serial_ones (the builders' corners)"""
    rng = np.random.default_rng(77)
    n = 262144 + 4096 + 37
    yield "two literal symbols", rng.integers(0, 2, n).astype(np.uint8) * 200 + 7
    yield "256 symbols equally often", rng.permutation(np.arange(n, dtype=np.int64) % 256).astype(np.uint8)
    yield "geometric counts (depths far above 11)", np.minimum(rng.geometric(0.5, n) - 1, 60).astype(np.uint8) * 3 + 1
    yield "geometric, high bytes (FSE-compressed tree)", (255 - np.minimum(rng.geometric(0.35, n) - 1, 200)).astype(np.uint8)
    rare = np.full(n, 65, np.uint8)
    rare[rng.integers(0, n, 300)] = rng.integers(0, 256, 300).astype(np.uint8)
    yield "one value with rare others", rare
    rec = rng.integers(0, 256, 16).astype(np.uint8)
    yield "one match length, one literal length (RLE tables)", np.concatenate([np.concatenate([rec, rng.integers(0, 256, 1).astype(np.uint8)]) for _ in range(n // 17)])
    few = rng.integers(0, 256, n).astype(np.uint8)
    for k in range(0, n - 70000, 65536):
        few[k + 5000 : k + 5040] = few[k + 100 : k + 140]
    yield "a handful of sequences (predefined tables)", few
    varied = rng.integers(0, 256, n).astype(np.uint8)
    pos = 3000
    while pos + 400 < n:
        ln = int(rng.integers(4, 300))
        src = int(rng.integers(max(0, pos - 60000), pos - ln)) if pos - ln > 0 else 0
        varied[pos : pos + ln] = varied[src : src + ln]
        pos += ln + int(rng.integers(1, 40))
    yield "many lengths at many distances", varied


def _lane_inputs():
    for kind in (0, 1, 2, 11, 12, 13):
        for n in (100, 5000, 131072, 400000):
            yield f"kind{kind}-{n}", ("synth", n, 2000 + n + kind, kind)
    for name, data in _alphabets():
        yield name, ("data", data)
    for name, data in _builder_corners():
        yield name, ("data", data)


@pytest.mark.parametrize("name,what", list(_lane_inputs()), ids=[n for n, _ in _lane_inputs()])
def test_sixty_four_lanes_on_the_host_write_the_one_lane_models_bytes(model, name, what):
    """oracle/zstd_model_lanes.c: zstd_block_core.h instantiated with ZB_LANES = 64 on the host (64 fibers that meet where the wave
    meets) -- the code the KERNEL compiles, with its all-lanes table builders (zb_normalize_par, zb_build_enc_table_par,
    zb_huffman_build_par) that the one-lane model never runs.  Its frames must be the one-lane model's, byte for byte, and decode with
    the reference: a divergence of the all-lanes code shows without a GPU (the GPU test of the same claim:
    tests/test_gpu_codecs.py::test_zstd_table_builders_all_lanes_equal_the_serial_ones)."""
    d = model.dll
    d.ltz_model_lanes64.restype = None
    d.ltz_model_lanes64.argtypes = [C.c_int]
    data = model.synth(what[1], what[2], what[3]) if what[0] == "synth" else what[1]
    d.ltz_model_sub_blocks(1)
    d.ltz_lanes_bad_site.restype = C.c_uint32
    d.ltz_lanes_bad_site.argtypes = [C.c_int]
    try:
        one = compress(model, data)
        for order in (1, 2):  # the lanes run 0 .. 63 / 63 .. 0 between two meeting points
            d.ltz_model_lanes64(order)
            b = np.ascontiguousarray(data, dtype=np.uint8)
            cap = d.ltz_model_bound(len(b))
            many = np.zeros(cap + 8, np.uint8)
            n = C.c_size_t(0)
            rc = d.ltz_model_compress(b.ctypes.data, len(b), many.ctypes.data, cap, C.byref(n))
            assert rc == 0, (f"{name}: the 64 lanes did not keep step (zstd_block_core.h lines {d.ltz_lanes_bad_site(0) & 0x7FFFFFFF} / "
                             f"{d.ltz_lanes_bad_site(1) & 0x7FFFFFFF}): a collective in divergent control flow, or a value read and written "
                             "by different lanes without a ZB_SYNC_LDS between")
            many = many[: n.value]
            assert len(one) == len(many) and (one == many).all(), f"{name}: 64 lanes (order {order}) wrote {len(many)} bytes, one lane {len(one)}"
    finally:
        d.ltz_model_lanes64(0)
        d.ltz_model_sub_blocks(0)
    err, out = get_ref().decompress(CODEC_ZSTD, many, len(data))
    assert err == 0 and len(out) == len(data) and (out == np.ascontiguousarray(data, dtype=np.uint8)).all()
