"""-m gpu: HIP LZ4 / ZStd block codecs through the C ABI.  Payloads must decode to the original bytes with the
oracle's LZ4 decoder and -- when oracle/_ref is present -- with the REFERENCE decoders; the HIP LZ4 decoder must
agree with LZ4_decompress_safe on reference-produced payloads."""
import numpy as np
import pytest
import torch

from tests._libs import have_ref, ref as get_ref
from tests.gpu_util import layout, to_device, u32

pytestmark = pytest.mark.gpu


def gpu_lz4(gpu, blocks, seg_log2=0, caps=None):
    dev, offs = to_device(blocks)
    bounds = [len(b) + len(b) // 255 + 16 for b in blocks]
    caps = caps or bounds
    d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
    dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    sizes = u32(gpu.lz4_compress_blocks(dev, offs, [len(b) for b in blocks], dst, d_offs, caps, seg_log2))
    host = dst.cpu().numpy()
    return [host[o : o + int(s)].copy() for o, s in zip(d_offs, sizes)], sizes


def check_lz4_payload(oracle, data, payload):
    assert len(payload) > 0
    assert len(payload) <= len(data) + len(data) // 255 + 16
    n, out = oracle.lz4_decompress(payload, len(data))  # exact capacity, like DecompressBlock (compressblockstore.c:321)
    assert n == len(data), f"oracle decoder rejected payload ({n})"
    assert (out == data).all()
    if have_ref():
        r = get_ref()
        err, out2 = r.decompress(0, payload, len(data))
        assert err == 0 and len(out2) == len(data) and (out2 == data).all(), "reference LZ4_decompress_safe rejected payload"


@pytest.mark.parametrize("seg_log2", [0, 10, 11, 13])
def test_lz4_roundtrip_kinds(gpu, oracle, seg_log2):
    blocks = []
    for kind in (0, 1, 2):
        for n in [1, 5, 11, 12, 13, 14, 64, 4096, 8191, 8192, 8193, 32767, 32768, 32769, 65535, 65536, 65537, 65541, 65548, 200000,
                  (1 << 20) + 7]:
            blocks.append(oracle.synth(n, 50 + n + kind, kind))
    payloads, _ = gpu_lz4(gpu, blocks, seg_log2)
    for b, p in zip(blocks, payloads):
        check_lz4_payload(oracle, b, p)


def test_lz4_empty_block(gpu, oracle):
    payloads, sizes = gpu_lz4(gpu, [np.zeros(0, np.uint8), oracle.synth(100, 1, 1)])
    assert int(sizes[0]) == 1 and payloads[0][0] == 0  # lz4.c:1361-1371
    check_lz4_payload(oracle, oracle.synth(100, 1, 1), payloads[1])


def test_lz4_reference_test_block(gpu, oracle, golden):
    lb = golden["tests"]["lz4_block"]  # the block of test/test.cpp:2092-2192
    blk = np.concatenate([np.full(n, v, np.uint8) for n, v in lb["runs"]])
    (p,), _ = gpu_lz4(gpu, [blk])
    check_lz4_payload(oracle, blk, p)
    assert len(p) < 200  # two long runs must collapse to a handful of sequences


def test_lz4_structured_inputs(gpu, oracle):
    rng = np.random.default_rng(2)
    text = np.frombuffer((b"the quick brown fox jumps over the lazy dog, " * 30000)[: 1 << 20], dtype=np.uint8).copy()
    lowent = rng.integers(0, 4, 1 << 20, dtype=np.uint8)
    rep = rng.integers(0, 256, 1 << 20, dtype=np.uint8)
    for i in range(0, len(rep) - 300, 997):
        rep[i + 100 : i + 300] = rep[i : i + 200]
    tail_match = np.concatenate([rng.integers(0, 256, 70000, dtype=np.uint8), np.zeros(40, np.uint8)])  # matches right up to the end
    blocks = [text, lowent, rep, tail_match, np.zeros(65536 + 3, np.uint8), np.zeros(32768 * 3 + 11, np.uint8)]
    payloads, _ = gpu_lz4(gpu, blocks)
    for b, p in zip(blocks, payloads):
        check_lz4_payload(oracle, b, p)
    assert len(payloads[0]) < len(text) // 10
    assert len(payloads[4]) < 2000


def test_lz4_ratio_vs_reference(gpu, oracle, golden):
    """Not a parity requirement, a sanity bound: the segmented GPU parse must stay in the reference's neighbourhood."""
    for kind, size, seed, ref_size in golden["vec"]["lz4_ref_sizes"]:
        d = oracle.synth(int(size), int(seed), int(kind))
        (p,), _ = gpu_lz4(gpu, [d])
        check_lz4_payload(oracle, d, p)
        assert len(p) <= int(ref_size) * (1.25 if size >= (1 << 20) else 1.45) + 1024, (kind, size, len(p), ref_size)


def test_lz4_many_blocks_and_capacity(gpu, oracle):
    blocks = [oracle.synth(int(n), 700 + i, i % 3) for i, n in enumerate(np.random.default_rng(4).integers(1, 400000, 60))]
    payloads, _ = gpu_lz4(gpu, blocks)
    for b, p in zip(blocks, payloads):
        check_lz4_payload(oracle, b, p)
    # too small a destination -> size 0 (LZ4CompressionAPI_Compress turns that into ENOMEM, longtail_lz4.c:70-74)
    rnd = oracle.synth(100000, 5, 0)
    _, sizes = gpu_lz4(gpu, [rnd, rnd], caps=[100, 100000 + 100000 // 255 + 16])
    assert int(sizes[0]) == 0 and int(sizes[1]) > 100000


def test_lz4_gpu_decoder_on_reference_payloads(gpu, oracle):
    blocks = [oracle.synth(n, 900 + n, k) for k in (0, 1, 2) for n in (0, 1, 13, 100, 70000, 300000)]
    comps = [oracle.lz4_compress(b) for b in blocks]  # bit-exact with LZ4_compress_fast (tests/test_oracle_vs_ref.py)
    dev, offs = to_device(comps)
    d_offs, total = layout(blocks)
    dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(c) for c in comps], dst, d_offs, [len(b) for b in blocks]))
    host = dst.cpu().numpy()
    for b, o, s in zip(blocks, d_offs, sizes):
        assert int(s) == len(b)
        assert (host[o : o + len(b)] == b).all()
    # malformed input is reported, not decoded
    bad = comps[4].copy()
    bad[len(bad) // 2 :] = 0
    dev, offs = to_device([bad])
    sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(bad)], dst, [0], [len(blocks[4])]))
    n, _ = oracle.lz4_decompress(bad, len(blocks[4]))
    assert (int(sizes[0]) == 0xFFFFFFFF) == (n < 0)


@pytest.fixture(params=[1, 2, 0], ids=["subblocks", "subblocks+repcodes", "pieces"])
def zmode(request, oracle, monkeypatch):
    """The layouts of the encoder's frames: a run of sub-blocks per 128 KiB piece with a directory (the default), the same with
    block-local repeat-offset codes (LTHIP_ZSTD_REP=1: version-3 trailer, the lane decoder carries a block-local history) and one
    block per piece (LTHIP_ZSTD_SUB=0); the host model follows."""
    import ctypes as C

    from longtail_amd.lib import load_ablations

    oracle.dll.ltz_model_sub_blocks.argtypes = [C.c_int]
    oracle.dll.ltz_model_sub_blocks.restype = None
    oracle.dll.ltz_model_flags.argtypes = [C.c_uint32]
    oracle.dll.ltz_model_flags.restype = None
    monkeypatch.setenv("LTHIP_ZSTD_SUB", "1" if request.param else "0")
    monkeypatch.setenv("LTHIP_ZSTD_REP", "1" if request.param == 2 else "0")
    if request.param != 1:  # (the two other layouts exist in the ablation build only: zgpu below)
        load_ablations().dll.lthip_debug_reload_env()
    oracle.dll.ltz_model_sub_blocks(1 if request.param else 0)
    oracle.dll.ltz_model_flags(1 if request.param == 2 else 0)
    yield request.param
    oracle.dll.ltz_model_sub_blocks(0)
    oracle.dll.ltz_model_flags(0)
    monkeypatch.delenv("LTHIP_ZSTD_REP")
    monkeypatch.delenv("LTHIP_ZSTD_SUB")
    if request.param != 1:
        load_ablations().dll.lthip_debug_reload_env()


@pytest.fixture
def zgpu(request, zmode):
    """The context a zmode test encodes with: the PRODUCT library for the default layout, the ablation build for the two others."""
    return request.getfixturevalue("gpu" if zmode == 1 else "gpu_abl")


def gpu_zstd(gpu, blocks, quality=0):
    dev, offs = to_device(blocks)
    caps = [len(b) + (len(b) >> 8) + 64 for b in blocks]
    d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
    dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    sizes = u32(gpu.zstd_compress_blocks(dev, offs, [len(b) for b in blocks], dst, d_offs, caps, quality=quality))
    host = dst.cpu().numpy()
    return [host[o : o + int(s)].copy() for o, s in zip(d_offs, sizes)]


def zstd_pieces(frame: np.ndarray):
    """[(type, payload bytes)] per 128 KiB piece of a single-segment frame with an 8-byte content size (the layouts k_zstd.hip
    writes).  A piece is one block -- or, in frames that end with the "LTP\\2" directory, a run of sub-blocks (one per 4 KiB unit):
    then type 2 and the payload is the run WITH its block headers, Last_Block cleared (what zb_encode_piece_sub returns)."""
    assert bytes(frame[:5]) == bytes([0x28, 0xB5, 0x2F, 0xFD, 0xE0])
    content = int.from_bytes(bytes(frame[5:13]), "little")
    nunits = (content + 4095) // 4096
    pos, blocks = 13, []
    while True:
        h = int(frame[pos]) | int(frame[pos + 1]) << 8 | int(frame[pos + 2]) << 16
        last, typ, size = h & 1, (h >> 1) & 3, h >> 3
        n = 1 if typ == 1 else size
        blocks.append((typ, pos, 3 + n))
        pos += 3 + n
        if last:
            break
    tail = bytes(frame[pos:])
    magic = bytes([0x5D, 0x2A, 0x4D, 0x18])
    if tail[:4] == magic and tail[8:12] in (b"LTP\x02", b"LTP\x03"):
        # sub-block frames: skippable frame with the directory, one u16 per unit
        assert len(tail) == 12 + 2 * nunits and int.from_bytes(tail[4:8], "little") == 4 + 2 * nunits
        d = np.frombuffer(tail[12:], dtype="<u2")
        out, k = [], 0
        for u0 in range(0, nunits, 32):
            nu = min(32, nunits - u0)
            if d[u0] >= 0xFFFE:
                assert (d[u0 : u0 + nu] == d[u0]).all()
                typ, p0, n = blocks[k]
                assert typ == (0 if d[u0] == 0xFFFF else 1)
                out.append((typ, frame[p0 + 3 : p0 + n]))
                k += 1
            else:
                p0 = blocks[k][1]
                for u in range(nu):
                    typ, _, n = blocks[k + u]
                    assert n == 3 + (int(d[u0 + u]) & 0x7FFF) and typ == (0 if d[u0 + u] & 0x8000 else 2)
                end = blocks[k + nu - 1][1] + blocks[k + nu - 1][2]
                run = frame[p0:end].copy()
                run[blocks[k + nu - 1][1] - p0] &= 0xFE
                out.append((2, run))
                k += nu
        assert k == len(blocks)
        return out
    # frames of two or more pieces end with the independence marker: a skippable frame (k_zstd.hip z_write_trailer)
    if tail:
        assert len(blocks) >= 2 and tail == magic + bytes([4, 0, 0, 0]) + b"LTP\x01"
    return [(typ, frame[p0 + 3 : p0 + n]) for typ, p0, n in blocks]


def test_zstd_frames_decode_with_reference(zgpu, oracle, ref, zmode):
    gpu = zgpu
    blocks = [oracle.synth(n, 40 + n, k) for k in (0, 1, 2, 11, 12, 13) for n in (0, 1, 100, 5000, 131071, 131072, 131073, 400000)]
    blocks.append(oracle.synth((8 << 20) + 12345, 7, 1))
    rng = np.random.default_rng(3)
    blocks.append((np.abs(rng.normal(128, 20, 700000)).astype(np.int64) % 256).astype(np.uint8))  # Huffman, FSE-coded weights
    blocks.append(np.frombuffer(b"the quick brown fox jumps over the lazy dog. " * 9000, np.uint8).copy())
    frames = gpu_zstd(gpu, blocks)
    for b, f in zip(blocks, frames):
        assert len(f) > 0
        err, out = ref.decompress(1, f, len(b))
        assert err == 0 and len(out) == len(b) and (out == b).all()
    # all-zero input collapses to RLE blocks
    z = [i for i, b in enumerate(blocks) if len(b) == 400000 and not b.any()]
    assert z and len(frames[z[0]]) < 100 + (12 + 2 * 98 if zmode else 0)  # (+ the directory: one u16 per 4 KiB unit)


def test_zstd_compresses(zgpu, oracle, zmode):
    gpu = zgpu
    """Compressed_Blocks are really produced, and beat the LZ4 payload of the same parse on entropy-codable data."""
    blocks = [oracle.synth(2 << 20, 3, k) for k in (1, 11, 12, 13)]
    frames = gpu_zstd(gpu, blocks)
    lz4, _ = gpu_lz4(gpu, blocks)
    for b, f, l in zip(blocks, frames, lz4):
        kinds = [t for t, _ in zstd_pieces(f)]
        assert kinds.count(2) >= len(kinds) // 2
        assert len(f) < len(l)


def _model_src(oracle):
    import ctypes as C

    d = oracle.dll
    d.ltz_model_encode_block_src.restype = C.c_uint32
    d.ltz_model_encode_block_src.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return d


def _check_against_model(gpu, d, blocks, frames):
    """every Compressed_Block of `frames` == the host model run on the GPU match finder's units; Raw where the model says so"""
    unit0 = 0
    checked = 0
    for b, f in zip(blocks, frames):
        b = np.ascontiguousarray(b)
        nunits = (len(b) + 4095) // 4096
        meta, lits, recs = gpu.zstd_debug_units(unit0, nunits)
        for i, (typ, payload) in enumerate(zstd_pieces(f)):
            raw = min(131072, len(b) - i * 131072)
            nu = (raw + 4095) // 4096
            out = np.zeros(140000, np.uint8)
            m, l, r = (np.ascontiguousarray(a[i * 32 : i * 32 + nu]) for a in (meta, lits, recs))
            l[m[:, 0] == 0] = 0x5A  # units without a sequence have no literal buffer: their literals are the source bytes
            piece = b[i * 131072 : i * 131072 + raw]
            n = d.ltz_model_encode_block_src(m.ctypes.data, l.ctypes.data, r.ctypes.data, nu, raw, piece.ctypes.data, out.ctypes.data)
            if typ == 2:
                assert n == len(payload) and (out[:n] == payload).all()
                checked += 1
            elif typ == 0:
                assert n == 0
        unit0 += nunits
    return checked


def test_zstd_entropy_stage_is_bit_exact_with_host_model(zgpu, oracle, zmode):
    gpu = zgpu
    """k_zstd_encode and oracle/zstd_model.c compile the SAME zstd_block_core.h (64 lanes vs 1): fed with the GPU match
    finder's own output, the host model must reproduce every Compressed_Block byte for byte."""
    d = _model_src(oracle)
    rng = np.random.default_rng(5)
    blocks = [oracle.synth(n, 90 + n, k) for k, n in ((1, 1 << 20), (11, 300000), (12, 131072), (13, 200000), (0, 140000))]
    blocks.append((np.abs(rng.normal(128, 20, 400000)).astype(np.int64) % 256).astype(np.uint8))
    frames = gpu_zstd(gpu, blocks)
    assert _check_against_model(gpu, d, blocks, frames) >= 10


def test_zstd_table_builders_all_lanes_equal_the_serial_ones(zgpu, oracle, ref, zmode):
    gpu = zgpu
    """The kernel builds the Huffman code and the three FSE tables with all lanes (zb_huffman_build_par, zb_normalize_par,
    zb_build_enc_table_par), the one-lane host model with the serial builders: literal and sequence statistics that reach the
    builders' corners -- two literal symbols, all 256 equally often, geometric counts (depths far above 11: the length limit and
    the FSE-compressed tree description), a single literal value with rare others; sequences with one code only (RLE tables),
    with a few (predefined tables: the "less than one" cells), and with many distinct lengths and offsets -- must give the same
    bytes as the model and decode with the reference."""
    d = _model_src(oracle)
    rng = np.random.default_rng(77)
    n = 262144 + 4096 + 37
    two = rng.integers(0, 2, n).astype(np.uint8) * 200 + 7
    flat = rng.permutation(np.arange(n, dtype=np.int64) % 256).astype(np.uint8)
    geo = np.minimum(rng.geometric(0.5, n) - 1, 60).astype(np.uint8) * 3 + 1  # P(k) ~ 2^-k: Huffman depths up to ~18
    geo_hi = (255 - np.minimum(rng.geometric(0.35, n) - 1, 200)).astype(np.uint8)  # > 128 weights: FSE-compressed tree
    rare = np.full(n, 65, np.uint8)
    rare[rng.integers(0, n, 300)] = rng.integers(0, 256, 300).astype(np.uint8)
    # sequences: a 16-byte record repeated with one random byte in between (one match length, one literal length: RLE tables);
    # a handful of sequences per piece (predefined tables); matches of many lengths at many distances
    rec = rng.integers(0, 256, 16).astype(np.uint8)
    same = np.concatenate([np.concatenate([rec, rng.integers(0, 256, 1).astype(np.uint8)]) for _ in range(n // 17)])
    few = rng.integers(0, 256, n).astype(np.uint8)
    for k in range(0, n - 70000, 65536):
        few[k + 5000 : k + 5040] = few[k + 100 : k + 140]
    varied = rng.integers(0, 256, n).astype(np.uint8)
    pos = 3000
    while pos + 400 < n:
        ln = int(rng.integers(4, 300))
        src = int(rng.integers(max(0, pos - 60000), pos - ln)) if pos - ln > 0 else 0
        varied[pos : pos + ln] = varied[src : src + ln]
        pos += ln + int(rng.integers(1, 40))
    blocks = [two, flat, geo, geo_hi, rare, same, few, varied]
    frames = gpu_zstd(gpu, blocks)
    assert _check_against_model(gpu, d, blocks, frames) >= 12
    for b, f in zip(blocks, frames):
        err, out = ref.decompress(1, f, len(b))
        assert err == 0 and (out == b).all()
    assert len(frames[0]) < n // 2 and len(frames[4]) < n // 20  # (two symbols; one symbol with rare others)


def test_zstd_unaligned_sources_and_literals_read_from_the_source(zgpu, oracle, ref, zmode):
    gpu = zgpu
    """Units without a sequence keep their literals in the source (no copy by the match finder), wherever the block lies:
    blocks at odd device offsets, with entropy-codable literals but no matches (Huffman from the source), noise (sampled
    early-out -> Raw, bytes placed by the match finder), and a mix, must equal the host model and decode with the reference."""
    d = _model_src(oracle)
    rng = np.random.default_rng(11)
    skew = (np.abs(rng.normal(128, 12, 500001)).astype(np.int64) % 256).astype(np.uint8)
    noise = rng.integers(0, 256, 400003, dtype=np.uint8)
    half = np.concatenate([rng.integers(0, 256, 200000, dtype=np.uint8), oracle.synth(300000, 4, 1), skew[:150000]])
    blocks = [skew, noise, half, oracle.synth(131072 * 2 + 5, 3, 12)]
    for shift in (1, 2, 3, 7):
        offs, pos = [], shift
        for b in blocks:
            offs.append(pos)
            pos += len(b) + shift + 8
        host = np.zeros(pos + 64, np.uint8)
        for o, b in zip(offs, blocks):
            host[o : o + len(b)] = b
        dev = torch.from_numpy(host).cuda()
        caps = [len(b) + (len(b) >> 8) + 64 for b in blocks]
        d_offs, total = layout([np.zeros(c + shift, np.uint8) for c in caps], align=1)
        d_offs = [o + shift for o in d_offs]
        dst = torch.full((total + 64 + shift,), 0xEE, dtype=torch.uint8, device="cuda")
        sizes = u32(gpu.zstd_compress_blocks(dev, offs, [len(b) for b in blocks], dst, d_offs, caps))
        out = dst.cpu().numpy()
        frames = [out[o : o + int(n)].copy() for o, n in zip(d_offs, sizes)]
        kinds = [[t for t, _ in zstd_pieces(f)] for f in frames]
        assert kinds[0].count(2) == len(kinds[0]) and kinds[1].count(0) == len(kinds[1])  # all Huffman / all Raw
        assert _check_against_model(gpu, d, blocks, frames) >= 8
        for b, f in zip(blocks, frames):
            err, back = ref.decompress(1, f, len(b))
            assert err == 0 and len(back) == len(b) and (back == b).all()


def gpu_zstd_decode(gpu, frames, caps):
    dev, offs = to_device(frames)
    d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
    dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    sizes = u32(gpu.zstd_decompress_blocks(dev, offs, [len(f) for f in frames], dst, d_offs, caps))
    host = dst.cpu().numpy()
    return [None if int(s) == 0xFFFFFFFF else host[o : o + int(s)].copy() for o, s in zip(d_offs, sizes)]


def test_zstd_decoder_reads_reference_and_own_frames(zgpu, oracle, ref, zmode):
    gpu = zgpu
    """The HIP zstd decoder against frames from the REFERENCE encoder (all five longtail settings: levels 3, 3, 22, 8, 22)
    and from the HIP encoder: decoded bytes identical to the original."""
    rng = np.random.default_rng(8)
    datas = [oracle.synth(n, 60 + n, k) for k in (0, 1, 2, 11, 12, 13) for n in (0, 1, 100, 5000, 131072, 131073, 400000)]
    datas.append(oracle.synth((8 << 20) + 77, 9, 1))
    datas.append((np.abs(rng.normal(128, 20, 700000)).astype(np.int64) % 256).astype(np.uint8))
    datas.append(np.frombuffer(b"the quick brown fox jumps over the lazy dog. " * 9000, np.uint8).copy())
    frames, raws = [], []
    for d in datas:
        for w in range(5):
            frames.append(ref.compress(1, ref.dll.refh_zstd_type(w), d))
            raws.append(d)
    own = gpu_zstd(gpu, datas)
    frames += own
    raws += datas
    # two frames back to back + a skippable frame in front: ZSTD_decompressDCtx accepts concatenations (zstd_decompress.c:1068)
    skippable = np.frombuffer(bytes([0x50, 0x2A, 0x4D, 0x18, 3, 0, 0, 0, 1, 2, 3]), np.uint8)
    frames.append(np.concatenate([skippable, frames[10], own[9]]))
    raws.append(np.concatenate([raws[10], datas[9]]))
    outs = gpu_zstd_decode(gpu, frames, [len(r) for r in raws])
    for f, r, o in zip(frames, raws, outs):
        assert o is not None and len(o) == len(r) and (o == r).all()


def test_zstd_decoder_agrees_with_host_model_and_reference_on_damaged_frames(zgpu, oracle, ref, zmode):
    gpu = zgpu
    """Same source on host and device (zstd_decode_core.h): identical verdict and bytes on mutated frames; whatever the
    decoder accepts the reference accepts with the same bytes (it is stricter than the reference's fast Huffman path,
    which does not check that a literal stream is consumed exactly, so the converse is not required)."""
    import ctypes as C

    d = oracle.dll
    d.ltz_model_decompress.restype = C.c_int
    d.ltz_model_decompress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
    rng = np.random.default_rng(4)
    frames, caps = [], []
    for kind, n in ((1, 200000), (11, 30000), (12, 100000), (13, 60000), (1, 3000)):
        b = oracle.synth(n, 31 + n, kind)
        for w in (0, 2):
            c = ref.compress(1, ref.dll.refh_zstd_type(w), b)
            for _ in range(60):
                x = c.copy()
                if rng.integers(0, 4) == 0:
                    x = x[: rng.integers(0, len(x) + 1)].copy()
                else:
                    for _ in range(int(rng.integers(1, 4))):
                        x[rng.integers(0, len(x))] ^= np.uint8(1 << rng.integers(0, 8))
                frames.append(x)
                caps.append(n if rng.integers(0, 3) else int(rng.integers(0, n + 1)))
    outs = gpu_zstd_decode(gpu, frames, caps)
    accepted = 0
    for f, cap, o in zip(frames, caps, outs):
        buf = np.zeros(cap + 8, np.uint8)
        m = C.c_size_t(0)
        e = d.ltz_model_decompress(f.ctypes.data, len(f), buf.ctypes.data, cap, C.byref(m))
        assert (e != 0) == (o is None)
        if o is not None:
            accepted += 1
            assert len(o) == m.value and (o == buf[: m.value]).all()
            if len(o) == cap:  # the reference wants the exact capacity case or larger; compare when it can succeed
                err, r_out = ref.decompress(1, f, cap)
                assert err == 0 and (r_out == o).all()
    assert 20 < accepted < len(frames)


MARKER = bytes([0x5D, 0x2A, 0x4D, 0x18, 4, 0, 0, 0]) + b"LTP\x01"  # k_zstd.hip z_write_trailer


def test_zstd_piece_decoder_is_the_serial_decoder_on_own_frames_and_strict_on_false_promises(gpu_abl, oracle, ref, monkeypatch):
    gpu = gpu_abl  # (the switches this test sets exist in the ablation build only)
    """Frames of two or more pieces end with the independence marker and are decoded piece by piece on separate waves.
    (1) same bytes as the serial decoder (LTHIP_ZSTD_DBG=1) and as the reference on the encoder's own frames, also damaged ones:
        the piece decoder never ACCEPTS what the serial one rejects, and what it accepts is identical;
    (2) a frame that carries the marker without keeping its promise -- the REFERENCE encoder's output (matches across blocks,
        repeat offsets, repeated tables) with the marker appended -- is rejected or decoded exactly like the reference, never
        decoded differently."""
    rng = np.random.default_rng(12)
    monkeypatch.setenv("LTHIP_ZSTD_SUB", "0")  # frames of one block per piece (the sub-block layout has its own test below)
    datas = [oracle.synth(n, 90 + n, k) for k, n in ((1, 131073), (1, 600000), (11, 400000), (12, 300000), (13, 262144), (0, 500000),
                                                    (2, 400000), (1, (8 << 20) + 5))]
    own = gpu_zstd(gpu, datas)
    assert all(bytes(f[-12:]) == MARKER for f in own)  # >= 2 pieces each
    frames, caps = list(own), [len(d) for d in datas]
    for f, d in zip(own[:6], datas[:6]):
        for _ in range(25):
            x = f.copy()
            k = rng.integers(0, 4)
            if k == 0:
                x = np.concatenate([x[: rng.integers(13, len(x) - 12)], x[-12:]])  # truncated, marker kept
            else:
                for _ in range(int(rng.integers(1, 4))):
                    x[rng.integers(0, len(x) - 12)] ^= np.uint8(1 << rng.integers(0, 8))  # bit flips before the marker
            frames.append(x)
            caps.append(len(d))
    pieces = gpu_zstd_decode(gpu, frames, caps)
    monkeypatch.setenv("LTHIP_ZSTD_DBG", "1")
    gpu.lib.dll.lthip_debug_reload_env()  # (the library caches its switches)
    serial = gpu_zstd_decode(gpu, frames, caps)
    monkeypatch.delenv("LTHIP_ZSTD_DBG")
    gpu.lib.dll.lthip_debug_reload_env()
    for i, (f, cap, p_out, s_out) in enumerate(zip(frames, caps, pieces, serial)):
        if i < len(own):
            assert p_out is not None and (p_out == datas[i]).all() and (s_out == datas[i]).all()
        if p_out is not None:
            assert s_out is not None and len(p_out) == len(s_out) and (p_out == s_out).all(), i
    assert sum(o is None for o in pieces[len(own):]) > 20  # the damage is really detected
    # (2) false promises
    lying, raws = [], []
    for d in datas[:5]:
        for w in (0, 2, 3):
            lying.append(np.concatenate([ref.compress(1, ref.dll.refh_zstd_type(w), d), np.frombuffer(MARKER, np.uint8)]))
            raws.append(d)
    outs = gpu_zstd_decode(gpu, lying, [len(r) for r in raws])
    for f, r, o in zip(lying, raws, outs):
        err, r_out = ref.decompress(1, f, len(r))
        assert err == 0 and (r_out == r).all()          # the reference skips the marker and decodes the frame
        assert o is None or (len(o) == len(r) and (o == r).all())


def test_zstd_sub_block_decoder_is_the_serial_decoder_and_strict(gpu_abl, oracle, ref, monkeypatch):
    gpu = gpu_abl  # (the switches this test sets exist in the ablation build only)
    """Frames in the sub-block layout (one zstd block per 4 KiB unit, directory in the trailer) are decoded with one lane per block
    (k_zstd_sub_entropy + k_zstd_execute<true>).
    (1) the encoder's own frames: same bytes as the serial decoder (LTHIP_ZSTD_DBG=1) and the reference;
    (2) damaged frames -- bit flips anywhere (blocks, headers, directory), truncations with the trailer kept, directory entries
        swapped: never accepted where the serial decoder rejects, and identical bytes where accepted;
    (3) a directory that describes the blocks correctly on a frame that does not keep the layout's promises (the reference
        encoder's frames: repeat offsets, matches across blocks) is given back to the serial decoder: decoded like the reference."""
    rng = np.random.default_rng(21)
    monkeypatch.setenv("LTHIP_ZSTD_SUB", "1")
    datas = [oracle.synth(n, 190 + n, k) for k, n in ((1, 131073), (1, 600000), (11, 400000), (12, 300000), (13, 262144), (0, 500000),
                                                     (2, 400000), (1, 4097), (12, 100), (1, (8 << 20) + 5))]
    half = np.concatenate([rng.integers(0, 256, 70000, dtype=np.uint8), oracle.synth(200000, 4, 1), rng.integers(0, 256, 9000, dtype=np.uint8)])
    datas.append(half)  # raw sub-blocks inside compressed pieces, raw pieces
    own = gpu_zstd(gpu, datas)
    for f, d in zip(own, datas):
        nu = (len(d) + 4095) // 4096
        assert bytes(f[-(12 + 2 * nu) :][8:12]) == b"LTP\x02"
    frames, caps = list(own), [len(d) for d in datas]
    for f, d in zip(own[:7] + own[10:], datas[:7] + datas[10:]):
        nu = (len(d) + 4095) // 4096
        tl = 12 + 2 * nu
        for _ in range(30):
            x = f.copy()
            k = rng.integers(0, 5)
            if k == 0:
                x = np.concatenate([x[: rng.integers(13, len(x) - tl)], x[-tl:]])  # truncated, trailer kept
            elif k == 1 and nu >= 2:
                a, b = rng.integers(0, nu, 2)
                e = x[-2 * nu :].view("<u2")
                e[a], e[b] = e[b], e[a]  # directory entries swapped
            elif k == 2:
                x[len(x) - 2 * nu + rng.integers(0, 2 * nu)] ^= np.uint8(1 << rng.integers(0, 8))  # a directory bit
            else:
                for _ in range(int(rng.integers(1, 4))):
                    x[rng.integers(0, len(x) - tl)] ^= np.uint8(1 << rng.integers(0, 8))  # bit flips before the trailer
            frames.append(x)
            caps.append(len(d))
    fast = gpu_zstd_decode(gpu, frames, caps)
    monkeypatch.setenv("LTHIP_ZSTD_DBG", "1")
    gpu.lib.dll.lthip_debug_reload_env()  # (the library caches its switches)
    serial = gpu_zstd_decode(gpu, frames, caps)
    monkeypatch.delenv("LTHIP_ZSTD_DBG")
    gpu.lib.dll.lthip_debug_reload_env()
    for i, (f, cap, p_out, s_out) in enumerate(zip(frames, caps, fast, serial)):
        if i < len(own):
            assert p_out is not None and (p_out == datas[i]).all() and s_out is not None and (s_out == datas[i]).all()
            err, r_out = ref.decompress(1, f, cap)
            assert err == 0 and (r_out == datas[i]).all()
        assert (p_out is None) == (s_out is None), i
        if p_out is not None:
            assert len(p_out) == len(s_out) and (p_out == s_out).all(), i
    assert sum(o is None for o in fast[len(own):]) > 20  # the damage is really detected
    # (3) a truthful directory over a block of another encoder: the reference's single block of a 4 KiB input (repeat-offset codes,
    # its own choice of table modes) under this library's frame header, directory = [its size]
    lying, raws = [], []
    for d in datas[:7]:
        d = d[:4096]
        for w in (0, 2, 3):
            f = ref.compress(1, ref.dll.refh_zstd_type(w), d)
            fhd = int(f[4])
            pos = 5 + (0 if fhd & 0x20 else 1) + (0, 1, 2, 4)[fhd & 3] + ((1, 2, 4, 8)[fhd >> 6] if (fhd >> 6) or (fhd & 0x20) else 0)
            h = int(f[pos]) | int(f[pos + 1]) << 8 | int(f[pos + 2]) << 16
            if not (h & 1):
                continue  # (more than one block: not a layout the directory can describe)
            n = 1 if (h >> 1) & 3 == 1 else h >> 3
            typ = (h >> 1) & 3
            e = np.array([0xFFFF if typ == 0 else 0xFFFE if typ == 1 else n], "<u2")
            t = bytes([0x5D, 0x2A, 0x4D, 0x18]) + (4 + 2).to_bytes(4, "little") + b"LTP\x02" + e.tobytes()
            head = bytes([0x28, 0xB5, 0x2F, 0xFD, 0xE0]) + len(d).to_bytes(8, "little")
            lying.append(np.frombuffer(head + bytes(f[pos : pos + 3 + n]) + t, np.uint8).copy())
            raws.append(d)
    assert len(lying) >= 10
    outs = gpu_zstd_decode(gpu, lying, [len(r) for r in raws])
    for f, r, o in zip(lying, raws, outs):
        err, r_out = ref.decompress(1, f, len(r))
        assert err == 0 and (r_out == r).all()  # the reference skips the trailer and decodes the frame
        assert o is not None and len(o) == len(r) and (o == r).all()


def test_zstd_reference_frames_block_parallel_is_the_serial_decoder(gpu_abl, oracle, ref, monkeypatch):
    gpu = gpu_abl  # (the switches this test sets exist in the ablation build only)
    """Frames of the REFERENCE encoder (blocks that depend on each other: window = the frame, repeat offsets, repeated tables,
    treeless literals) are decoded block-parallel: every block's streams on a wave / lane of its own (k_zstd_blk_entropy,
    k_zstd_blk_sequences), then a payload's blocks in order on one wave (k_zstd_execute_payload); whatever that path declines goes to
    the serial decoder.  All five longtail settings, sizes from a few KiB to 8 MiB + 5, raw and RLE blocks inside frames; and 300
    damaged frames: same verdict and bytes as the serial decoder (LTHIP_ZSTD_DBG=1), undamaged ones equal to the data."""
    rng = np.random.default_rng(33)
    datas = [oracle.synth(n, 290 + n, k) for k, n in ((1, 131073), (1, 600000), (11, 400000), (12, 300000), (13, 262144), (2, 400000),
                                                     (1, 5000), (12, (8 << 20) + 5))]
    datas.append(np.concatenate([rng.integers(0, 256, 200000, dtype=np.uint8), oracle.synth(300000, 4, 1), np.zeros(150000, np.uint8)]))
    frames, caps, truth = [], [], []
    for i, d in enumerate(datas):
        for w in range(5):
            if len(d) > (1 << 20) and w not in (0, 3):
                continue  # (the slow settings only on the smaller inputs)
            frames.append(ref.compress(1, ref.dll.refh_zstd_type(w), d))
            caps.append(len(d))
            truth.append(d)
    clean = len(frames)
    for f, d in list(zip(frames[:clean], truth[:clean]))[:30]:
        for _ in range(10):
            x = f.copy()
            k = rng.integers(0, 4)
            if k == 0:
                x = x[: rng.integers(1, len(x))]
            elif k == 1:
                a = int(rng.integers(0, len(x)))
                x[a : a + int(rng.integers(1, 40))] = int(rng.integers(0, 256))
            else:
                for _ in range(int(rng.integers(1, 4))):
                    x[rng.integers(0, len(x))] ^= np.uint8(1 << rng.integers(0, 8))
            frames.append(x)
            caps.append(len(d))
            truth.append(None)
    clean_out = gpu_zstd_decode(gpu, frames[:clean], caps[:clean])
    n_pay, n_blocks, n_back, _ = gpu.zstd_last_decode_stats()
    assert n_pay == clean and n_back == 0 and n_blocks >= sum((len(t) + 131071) // 131072 for t in truth[:clean])  # none went the serial way
    assert all(o is not None and (o == t).all() for o, t in zip(clean_out, truth[:clean]))
    fast = gpu_zstd_decode(gpu, frames, caps)
    # the sequences of a call with few blocks are walked by the scalar unit (k_zstd_blk_seq_scalar): forced here for all of them, and
    # switched off, so that both ways see the damaged frames too
    monkeypatch.setenv("LTHIP_ZSTD_SEQ_SCALAR", "2")
    gpu.lib.dll.lthip_debug_reload_env()  # (the library caches its switches)
    scalar = gpu_zstd_decode(gpu, frames, caps)
    monkeypatch.setenv("LTHIP_ZSTD_SEQ_SCALAR", "0")
    gpu.lib.dll.lthip_debug_reload_env()
    lanes = gpu_zstd_decode(gpu, frames, caps)
    monkeypatch.delenv("LTHIP_ZSTD_SEQ_SCALAR")
    monkeypatch.setenv("LTHIP_ZSTD_DBG", "1")
    gpu.lib.dll.lthip_debug_reload_env()
    serial = gpu_zstd_decode(gpu, frames, caps)
    monkeypatch.delenv("LTHIP_ZSTD_DBG")
    gpu.lib.dll.lthip_debug_reload_env()
    for way in (fast, scalar, lanes):
        for i, (f_out, s_out, t) in enumerate(zip(way, serial, truth)):
            if t is not None:
                assert f_out is not None and len(f_out) == len(t) and (f_out == t).all(), i
            assert (f_out is None) == (s_out is None), i
            if f_out is not None:
                assert len(f_out) == len(s_out) and (f_out == s_out).all(), i
    assert sum(o is None for o in fast[clean:]) > 50  # the damage is really detected


def test_lz4_gpu_decoder_differential_fuzz(gpu, oracle):
    """Damaged payloads: the HIP decoder accepts exactly what the oracle's strict LZ4_decompress_safe restatement accepts
    (lz4.c:2215-2435), with the same size and bytes -- truncations, bit flips, zeroed tails, wrong capacities."""
    rng = np.random.default_rng(77)
    raws = [oracle.synth(n, 300 + n, k) for k, n in ((1, 70000), (1, 9000), (2, 5000), (0, 3000), (11, 40000), (12, 20000))]
    raws.append(np.frombuffer(b"abcdefgh" * 3000 + b"x" * 70000 + bytes(range(256)) * 40, np.uint8).copy())  # long matches, far offsets
    far = rng.integers(0, 256, 30000, dtype=np.uint8)
    raws.append(np.concatenate([far, rng.integers(0, 256, 20000, dtype=np.uint8), far]))  # offsets beyond the 8 KiB ring
    cases = []
    for raw in raws:
        comp = oracle.lz4_compress(raw)
        cases.append((comp, len(raw)))
        for _ in range(24):
            c = comp.copy()
            kind = rng.integers(0, 5)
            if kind == 0 and len(c) > 2:
                c = c[: rng.integers(1, len(c))]
            elif kind == 1:
                for _ in range(rng.integers(1, 4)):
                    c[rng.integers(0, len(c))] ^= 1 << rng.integers(0, 8)
            elif kind == 2:
                c[rng.integers(0, len(c)) :] = 0
            elif kind == 3:
                c[rng.integers(0, len(c))] = 255
            cap = len(raw) if kind != 4 else max(0, len(raw) + int(rng.integers(-20, 20)))
            cases.append((c, cap))
    comps = [c for c, _ in cases]
    caps = [cap for _, cap in cases]
    dev, offs = to_device(comps)
    d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
    dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(c) for c in comps], dst, d_offs, caps))
    host = dst.cpu().numpy()
    accepted = 0
    for (c, cap), o, s in zip(cases, d_offs, sizes):
        n, out = oracle.lz4_decompress(c, cap)
        if n < 0:
            assert int(s) == 0xFFFFFFFF
        else:
            assert int(s) == n and (host[o : o + n] == out[:n]).all()
            accepted += 1
    assert 8 <= accepted < len(cases)


def test_lz4_gpu_decoder_structured_cases(gpu, oracle):
    """The decoder's internal boundaries, one by one: overlapping matches of every small period (lane-modulo path), periods
    around 64 (step path), offsets around the 8 KiB output ring and at the format's maximum, matches longer than a ring
    segment, literal runs longer than the payload window, output sizes around the 2 KiB flush granule and odd alignments."""
    rng = np.random.default_rng(123)
    raws = []
    for period in (1, 2, 3, 7, 31, 62, 63, 64, 65, 127, 2047, 2048, 2049, 4095, 4096, 4097, 8191, 8192, 8193, 16384, 65534, 65535):
        seed = rng.integers(0, 256, period, dtype=np.uint8)
        reps = max(3, 150000 // period)
        raws.append(np.concatenate([np.tile(seed, reps), rng.integers(0, 256, 13, dtype=np.uint8)]))
    raws.append(np.concatenate([rng.integers(0, 256, 20000, dtype=np.uint8), np.zeros(300000, np.uint8), rng.integers(0, 256, 5000, dtype=np.uint8)]))
    for n in (2047, 2048, 2049, 4095, 4096, 4097, 6143, 6144, 6145):  # flush granule
        raws.append(np.concatenate([np.zeros(n - 20, np.uint8), rng.integers(0, 256, 20, dtype=np.uint8)]))
    comps = [oracle.lz4_compress(r) for r in raws]
    for shift in (0, 1, 5, 15):  # source / destination alignment relative to 16 bytes
        offs, pos = [], shift
        for c in comps:
            offs.append(pos)
            pos += len(c) + 16 + shift
        host = np.zeros(pos + 64, np.uint8)
        for o, c in zip(offs, comps):
            host[o : o + len(c)] = c
        dev = torch.from_numpy(host).cuda()
        d_offs, total = layout([np.zeros(len(r) + 32, np.uint8) for r in raws])
        d_offs = [o + shift for o in d_offs]
        dst = torch.full((total + 64 + shift,), 0xCD, dtype=torch.uint8, device="cuda")
        sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(c) for c in comps], dst, d_offs, [len(r) for r in raws]))
        out = dst.cpu().numpy()
        for r, o, s in zip(raws, d_offs, sizes):
            assert int(s) == len(r)
            assert (out[o : o + len(r)] == r).all()
            assert (out[o + len(r) : o + len(r) + 8] == 0xCD).all()  # nothing written past the block


def test_lz4_block_parallel_decoder(gpu, oracle):
    """Blocks of two or more 64 KiB units take the block-parallel path (tile walk, link pass, one wave per unit): payloads with a
    sliding window (the reference's parse: every unit reads what the unit before wrote), payloads of the HIP encoder (independent
    groups), incompressible and run-length blocks (one sequence spanning many units), sizes around the unit, odd alignments, and a
    call that mixes them with blocks of the wave-per-block decoder.  Sizes and bytes must be the oracle decoder's."""
    rng = np.random.default_rng(2024)
    raws = []
    for kind, n in ((1, 131072), (1, 131073), (1, 200000), (11, 65536 * 3), (12, 65536 * 5 + 1), (0, 300001), (1, (3 << 20) + 5), (13, 1 << 20)):
        raws.append(oracle.synth(n, 900 + n % 1000, kind))
    raws.append(np.zeros(700000, np.uint8))  # one match across ten units
    raws.append(np.concatenate([rng.integers(0, 256, 100000, dtype=np.uint8), np.zeros(400000, np.uint8), oracle.synth(200000, 5, 1)]))
    far = rng.integers(0, 256, 60000, dtype=np.uint8)
    raws.append(np.concatenate([far, oracle.synth(5000, 6, 1), far, far[:30000], oracle.synth(70000, 7, 12), far]))  # offsets near 64 KiB across units
    raws.append(oracle.synth(4000, 8, 1))  # a small block in the same call: wave-per-block decoder
    raws.append(oracle.synth(131071, 9, 1))  # one byte short of two units
    comps_ref = [oracle.lz4_compress(r) for r in raws]
    comps_hip, _ = gpu_lz4(gpu, raws)
    for comps in (comps_ref, comps_hip):
        for shift in (0, 3):
            offs, pos = [], shift
            for c in comps:
                offs.append(pos)
                pos += len(c) + 16 + shift
            host = np.zeros(pos + 64, np.uint8)
            for o, c in zip(offs, comps):
                host[o : o + len(c)] = c
            dev = torch.from_numpy(host).cuda()
            d_offs, total = layout([np.zeros(len(r) + 32, np.uint8) for r in raws])
            d_offs = [o + shift for o in d_offs]
            dst = torch.full((total + 64 + shift,), 0xCD, dtype=torch.uint8, device="cuda")
            sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(c) for c in comps], dst, d_offs, [len(r) for r in raws]))
            out = dst.cpu().numpy()
            for i, (r, o, s) in enumerate(zip(raws, d_offs, sizes)):
                assert int(s) == len(r), (i, int(s), len(r))
                assert (out[o : o + len(r)] == r).all(), i
                assert (out[o + len(r) : o + len(r) + 8] == 0xCD).all()  # nothing written past the block


def test_lz4_block_parallel_decoder_differential_fuzz(gpu, oracle):
    """Damaged payloads of several units: accepted exactly when the oracle's strict decoder accepts them, with its size and bytes."""
    rng = np.random.default_rng(4711)
    raws = [oracle.synth(n, 40 + k, k) for k, n in ((1, 400000), (11, 262144), (12, 150000), (0, 140000))]
    raws.append(np.concatenate([oracle.synth(100000, 3, 1), np.zeros(300000, np.uint8), oracle.synth(50000, 4, 1)]))
    hip, _ = gpu_lz4(gpu, raws)
    cases = []
    for raw, own in zip(raws, hip):
        for comp in (oracle.lz4_compress(raw), own):
            cases.append((comp, len(raw)))
            for _ in range(16):
                c = comp.copy()
                kind = rng.integers(0, 6)
                if kind == 0:
                    c = c[: rng.integers(1, len(c))]
                elif kind == 1:
                    for _ in range(rng.integers(1, 4)):
                        c[rng.integers(0, len(c))] ^= 1 << rng.integers(0, 8)
                elif kind == 2:
                    c[rng.integers(0, len(c)) :] = 0
                elif kind == 3:
                    c[rng.integers(0, len(c))] = 255
                elif kind == 4:
                    a = rng.integers(0, len(c) - 2)
                    c[a : a + 2] = 0  # a zero offset (or a token) somewhere
                cap = len(raw) if kind != 5 else max(131072, len(raw) + int(rng.integers(-70000, 70000)))
                cases.append((c, cap))
    comps = [c for c, _ in cases]
    caps = [cap for _, cap in cases]
    dev, offs = to_device(comps)
    d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
    dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
    sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(c) for c in comps], dst, d_offs, caps))
    host = dst.cpu().numpy()
    accepted = 0
    for i, ((c, cap), o, s) in enumerate(zip(cases, d_offs, sizes)):
        n, out = oracle.lz4_decompress(c, cap)
        if n < 0:
            assert int(s) == 0xFFFFFFFF, (i, int(s))
        else:
            assert int(s) == n, (i, int(s), n)
            assert (host[o : o + n] == out[:n]).all(), i
            accepted += 1
    assert 10 <= accepted < len(cases)


@pytest.mark.gpu
def test_origin_execution_is_the_chained_execution(gpu_abl, oracle, ref, monkeypatch):
    gpu = gpu_abl  # (the switches this test sets exist in the ablation build only)
    """Payloads whose matches cross the pieces they are decoded in -- LZ4 blocks with a sliding window (the reference's parse), zstd
    frames of the reference encoder -- are executed on ORIGINS (origin_exec.h: k_lz4_po_trace/_gather, k_zstd_fr_reps/_chain/_trace/
    _gather).  Chain-heavy data (every token / record / line repeats an earlier one: the closure of what depends on the piece before
    is the whole piece), a block of 8 MiB + 5, repeat offsets at block starts, raw and RLE blocks inside frames, two arena budgets
    (several groups per call).  Same sizes and bytes as round 2's execution (a unit waits for the unit before: LTHIP_LZ4_PD_WAIT=1;
    a frame's blocks one after the other on one wave: LTHIP_ZSTD_DBG=16) and as the data."""
    rng = np.random.default_rng(77)
    raws = [oracle.synth(n, 500 + k, k) for k, n in ((12, (8 << 20) + 5), (11, 3 << 20), (13, 2 << 20), (1, 5 << 20), (12, 131073), (2, 1 << 20))]
    raws.append(np.concatenate([oracle.synth(300000, 9, 12), rng.integers(0, 256, 200000, dtype=np.uint8), oracle.synth(300000, 9, 12),
                                np.zeros(200000, np.uint8), oracle.synth(300000, 9, 12)]))
    lz = [oracle.lz4_compress(r) for r in raws]
    zs = [ref.compress(1, ref.zstd_default, r) for r in raws]
    caps = [len(r) for r in raws]

    def run_lz4():
        dev, offs = to_device(lz)
        d_offs, total = layout([np.zeros(c, np.uint8) for c in caps])
        dst = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        sizes = u32(gpu.lz4_decompress_blocks(dev, offs, [len(c) for c in lz], dst, d_offs, caps))
        host = dst.cpu().numpy()
        return [host[o : o + int(s)].copy() for o, s in zip(d_offs, sizes)]

    results = {}
    for budget in ("1", "4096"):  # MiB of origins in flight: one block per group / all of them
        monkeypatch.setenv("LTHIP_ORIGIN_MIB", budget)
        gpu.lib.dll.lthip_debug_reload_env()
        results["lz4", budget] = run_lz4()
        results["zstd", budget] = gpu_zstd_decode(gpu, zs, caps)
        n_pay, n_blocks, n_back, _ = gpu.zstd_last_decode_stats()
        assert n_pay == len(zs) and n_back == 0 and n_blocks >= sum((c + 131071) // 131072 for c in caps)
    monkeypatch.setenv("LTHIP_LZ4_PD_WAIT", "1")
    monkeypatch.setenv("LTHIP_ZSTD_DBG", "16")
    gpu.lib.dll.lthip_debug_reload_env()
    results["lz4", "chained"] = run_lz4()
    results["zstd", "chained"] = gpu_zstd_decode(gpu, zs, caps)
    for key, outs in results.items():
        for i, (o, r) in enumerate(zip(outs, raws)):
            assert o is not None and len(o) == len(r) and (o == r).all(), (key, i)


@pytest.mark.gpu
def test_lane_parser_shared_table_tag_runs_out(gpu_abl, oracle, monkeypatch):
    gpu = gpu_abl  # (the switches this test sets exist in the ablation build only)
    """The lane parser's shared history table is never cleared between the groups of a workgroup: entries carry a group tag that
    counts down from 0xFFFF and the table is cleared when it runs out (after 65535 groups = 4 GiB per workgroup).  LTHIP_LZ4_DBG bit 26
    makes it run out every third group: same bytes as without, and they decode."""
    # (one persistent workgroup per CU: more than 3 x 256 groups of 64 KiB with redundancy, so that every workgroup sees a fourth group)
    raws = [oracle.synth(24 << 20, 70 + k, k) for k in (1, 11, 12)] + [oracle.synth((8 << 20) + 12345, 9, 11)]
    plain, _ = gpu_lz4(gpu, raws)
    monkeypatch.setenv("LTHIP_LZ4_DBG", str(1 << 26))
    short, _ = gpu_lz4(gpu, raws)
    monkeypatch.delenv("LTHIP_LZ4_DBG")
    for a, b, r in zip(plain, short, raws):
        assert len(a) == len(b) and (a == b).all()
        n, out = oracle.lz4_decompress(b, len(r))
        assert n == len(r) and (out[:n] == r).all()


def test_lane_parser_tickets_give_the_payloads_of_the_fixed_stride(gpu_abl, oracle, monkeypatch):
    gpu = gpu_abl  # (the switches this test sets exist in the ablation build only)
    """The lane parser's workgroups DRAW the listed groups (a ticket each, one group ahead) instead of striding over the list; a
    payload is a function of its block, whoever parsed its groups and in which order: blocks that are no multiples of the 64 KiB
    groups, the block list forwards and backwards, tickets and LTHIP_LZ4_DBG bit 27 (the fixed stride) -- the same bytes."""
    raws = [oracle.synth((6 << 20) - 4099 * k - 1, 30 + k, (1, 11, 12, 13)[k % 4]) for k in range(12)]
    drawn, _ = gpu_lz4(gpu, raws)
    back, _ = gpu_lz4(gpu, raws[::-1])
    monkeypatch.setenv("LTHIP_LZ4_DBG", str(1 << 27))
    gpu.lib.dll.lthip_debug_reload_env()
    strided, _ = gpu_lz4(gpu, raws)
    monkeypatch.delenv("LTHIP_LZ4_DBG")
    gpu.lib.dll.lthip_debug_reload_env()
    for a, b, c, r in zip(drawn, back[::-1], strided, raws):
        assert len(a) == len(b) == len(c) and (a == b).all() and (a == c).all()
        n, out = oracle.lz4_decompress(a, len(r))
        assert n == len(r) and (out[:n] == r).all()


def test_zstd_repcode_frames_stay_on_the_lane_decoder(gpu_abl, oracle, ref, monkeypatch):
    gpu = gpu_abl  # (the switches this test sets exist in the ablation build only)
    """LTHIP_ZSTD_REP=1: the frames say so in their trailer (version 3), their blocks use repeat-offset codes only for history entries
    the block itself has set, the reference decodes them, and this library's lane-per-block decoder resolves the codes itself --
    none of the payloads is handed back to the serial decoder.  Smaller than the same frames with plain offsets."""
    monkeypatch.setenv("LTHIP_ZSTD_SUB", "1")
    blocks = [oracle.synth(1 << 20, 5, k) for k in (1, 11, 12, 13)]
    rng = np.random.default_rng(12)
    rec = rng.integers(0, 256, (64, 40)).astype(np.uint8)  # 40-byte records from a small dictionary, one field varying: offsets repeat
    rows = rec[rng.integers(0, 64, 20000)].copy()
    rows[:, 7] = rng.integers(0, 256, 20000)
    blocks.append(rows.reshape(-1))
    sizes = {}
    for rep in ("0", "1"):
        monkeypatch.setenv("LTHIP_ZSTD_REP", rep)
        gpu.lib.dll.lthip_debug_reload_env()
        frames = gpu_zstd(gpu, blocks)
        for b, f in zip(blocks, frames):
            nu = (len(b) + 4095) // 4096
            assert bytes(f[-(12 + 2 * nu) :][8:12]) == (b"LTP\x03" if rep == "1" else b"LTP\x02")
            err, out = ref.decompress(1, f, len(b))
            assert err == 0 and len(out) == len(b) and (out == b).all()
        got = gpu_zstd_decode(gpu, frames, [len(b) for b in blocks])
        assert all(g is not None and (g == b).all() for g, b in zip(got, blocks))
        stats = gpu.zstd_last_decode_stats()
        assert stats[0] == len(blocks) and stats[2] == 0, stats  # nothing went back to the serial decoder
        sizes[rep] = [len(f) for f in frames]
    monkeypatch.delenv("LTHIP_ZSTD_REP")
    gpu.lib.dll.lthip_debug_reload_env()
    assert sizes["1"][-1] < sizes["0"][-1] and sum(sizes["1"]) <= sum(sizes["0"])


def test_zstd_settings_are_three_parses(gpu, oracle, ref, monkeypatch):
    """LTHIP_ZSTD_Q_DEFAULT / _HIGH / _MAX ('ztd1'/'ztd2', 'ztd4', 'ztd3'/'ztd5' of lib/zstd/longtail_zstd.c:11-28): "high" gives every
    redundant 32 KiB half of a piece the half in front of it as history (k_lz4_pair_halves) -- inside the piece: its pieces stay
    independent --, "max" gives it to EVERY redundant half but a block's first (history across the pieces: trailer version 4, the
    decoder runs such a frame's pieces as a chain) and reads the private table again after a step's inserts.  Both are smaller than
    the default on every synthetic kind with structure, all three decode with the reference, and with this library's lane-per-block
    decoder WITHOUT a payload going back to the serial one."""
    blocks = [oracle.synth((2 << 20) + 4097 * k, 70 + k, k) for k in (1, 11, 12, 13)]
    blocks.append(oracle.synth(300000, 5, 12))
    blocks.append(oracle.synth(1 << 20, 6, 0))  # incompressible: the same at every setting
    sizes = []
    for q in (0, 1, 2):
        frames = gpu_zstd(gpu, blocks, quality=q)
        for b, f in zip(blocks, frames):
            err, out = ref.decompress(1, f, len(b))
            assert err == 0 and len(out) == len(b) and (out == b).all()
        got = gpu_zstd_decode(gpu, frames, [len(b) for b in blocks])
        assert all(g is not None and (g == b).all() for g, b in zip(got, blocks))
        stats = gpu.zstd_last_decode_stats()
        assert stats[0] == len(blocks) and stats[2] == 0, stats
        sizes.append([len(f) for f in frames])
    for i in range(5):
        assert sizes[1][i] < sizes[0][i] and sizes[2][i] < sizes[0][i], (i, [s[i] for s in sizes])
    assert sizes[0][5] == sizes[1][5] == sizes[2][5]
    # "tokens": the vocabulary's first occurrences inside a half are what the history buys -- more than 8 %, and more than 8 % again
    # when a piece's first half has it too
    assert sizes[1][2] < 0.92 * sizes[0][2] and sizes[2][2] < 0.92 * sizes[1][2]


def test_zstd_max_setting_frames_are_chains_of_pieces(gpu, gpu_abl, oracle, ref, monkeypatch):
    """'max' ('ztd3' / 'ztd5'): matches reach into the piece before, the frame says so (directory trailer version 4) and
    k_zstd_execute<true> runs its pieces in order -- a piece waits for the flag of the piece in front of it; the items of a call are
    taken in piece-major order (k_zstd_rows) so that all frames' chains run side by side, also across the rounds of a large call.
    (1) many frames of ragged sizes in one call, more pieces than one round holds: the data, no payload back to the serial decoder;
    (2) the version byte forged: a chain frame marked independent (2) goes back to the serial decoder and decodes, an independent
        frame marked as a chain (4) decodes as a chain;
    (3) damaged chain frames: never accepted where the serial decoder (ablation build, LTHIP_ZSTD_DBG=1) rejects, same bytes where
        accepted -- a piece that fails tells the pieces behind it."""
    rng = np.random.default_rng(33)
    sizes = [(8 << 20) + 77, 131072, 131073, 4097, 100, 600000, (3 << 20) + 5] + [8 << 20] * 132  # 132 x 64 + ... pieces: more than one round of 8192
    kinds = [1, 12, 11, 13, 1, 12, 1] + [1, 11, 12] * 44
    datas = [oracle.synth(n, 500 + i, k) for i, (n, k) in enumerate(zip(sizes, kinds))]
    assert sum((n + 131071) // 131072 for n in sizes) > 8192 + 64
    frames = gpu_zstd(gpu, datas, quality=2)
    for f, d in zip(frames, datas):
        nu = (len(d) + 4095) // 4096
        assert bytes(f[-(12 + 2 * nu) :][8:12]) == b"LTP\x04"
    for f, d in list(zip(frames, datas))[:9]:
        err, out = ref.decompress(1, f, len(d))
        assert err == 0 and len(out) == len(d) and (out == d).all()
    got = gpu_zstd_decode(gpu, frames, [len(d) for d in datas])
    assert all(g is not None and len(g) == len(d) and (g == d).all() for g, d in zip(got, datas))
    stats = gpu.zstd_last_decode_stats()
    assert stats[0] == len(datas) and stats[2] == 0, stats
    # (2) forged version bytes
    forged, raws = [], []
    for f, d in list(zip(frames, datas))[:7]:
        x = f.copy()
        x[len(x) - 2 * ((len(d) + 4095) // 4096) - 1] = 2
        forged.append(x)
        raws.append(d)
    plain = gpu_zstd(gpu, datas[:7], quality=1)
    for f, d in zip(plain, datas[:7]):
        x = f.copy()
        k = len(x) - 2 * ((len(d) + 4095) // 4096) - 1
        assert x[k] == 2
        x[k] = 4
        forged.append(x)
        raws.append(d)
    got = gpu_zstd_decode(gpu, forged, [len(d) for d in raws])
    assert all(g is not None and len(g) == len(d) and (g == d).all() for g, d in zip(got, raws))
    stats = gpu.zstd_last_decode_stats()
    assert stats[2] >= 3, stats  # (the large chain frames marked independent: cross-piece offsets are not what version 2 promises)
    # (3) damage
    bad, caps = [], []
    for f, d in list(zip(frames, datas))[:8]:
        nu = (len(d) + 4095) // 4096
        tl = 12 + 2 * nu
        for _ in range(20):
            x = f.copy()
            k = rng.integers(0, 4)
            if k == 0 and len(x) - tl > 14:
                x = np.concatenate([x[: rng.integers(13, len(x) - tl)], x[-tl:]])
            elif k == 1:
                x[len(x) - 2 * nu + rng.integers(0, 2 * nu)] ^= np.uint8(1 << rng.integers(0, 8))
            else:
                for _ in range(int(rng.integers(1, 4))):
                    x[rng.integers(0, len(x) - tl)] ^= np.uint8(1 << rng.integers(0, 8))
            bad.append(x)
            caps.append(len(d))
    fast = gpu_zstd_decode(gpu, bad, caps)
    monkeypatch.setenv("LTHIP_ZSTD_DBG", "1")
    gpu_abl.lib.dll.lthip_debug_reload_env()
    serial = gpu_zstd_decode(gpu_abl, bad, caps)
    monkeypatch.delenv("LTHIP_ZSTD_DBG")
    gpu_abl.lib.dll.lthip_debug_reload_env()
    for i, (p_out, s_out) in enumerate(zip(fast, serial)):
        assert (p_out is None) == (s_out is None), i
        if p_out is not None:
            assert len(p_out) == len(s_out) and (p_out == s_out).all(), i
    assert sum(o is None for o in fast) > 20
