#!/usr/bin/env python3
"""Generates tests/golden/* from the REFERENCE (run in the build container only; needs /root/reference and
oracle/_ref/liblongtail_ref.so).  The outputs are DATA: inputs + expected outputs.

  chunker.input          the reference's own 1 MiB chunker test file (test/testdata/chunker.input), copied verbatim
  reference_tests.json   expectations the reference's own tests hold:
                           - BLAKE3 KAT                    test/test.cpp:465-474
                           - 20 chunk ranges               test/test.cpp:3423-3445 (extracted from the file by regex)
                           - LZ4 payload of 1147x13+4711x77 test/test.cpp:2092-2192 pins its size to 38 bytes; the bytes are
                             produced here by the reference encoder
  ref_vectors.npz        outputs of the reference library on seeded synthetic inputs (include/longtail_synth.h):
                           chunk lengths + chunk hashes for several (kind, size, target) cases, BLAKE3 of prefixes of
                           many lengths, NextChunkFromBuffer lengths, reference LZ4 sizes.
"""
import json
import re
import shutil
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests._libs import GOLDEN, oracle, ref  # noqa: E402

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")


def params(target):
    return max(48, target // 8), max(48, target // 2), max(48, target * 2)


def main():
    GOLDEN.mkdir(parents=True, exist_ok=True)
    o, r = oracle(), ref()
    shutil.copyfile(REF / "test/testdata/chunker.input", GOLDEN / "chunker.input")

    test_cpp = (REF / "test/test.cpp").read_text()
    # --- BLAKE3 KAT ---
    m = re.search(r'TEST\(Longtail, Longtail_Blake3\)(.*?)\n}', test_cpp, re.S)
    body = m.group(1)
    kat_str = re.search(r'const char\* test_string = "(.*?)";', body).group(1)
    kat_val = int(re.search(r'ASSERT_EQ\((0x[0-9a-fA-F]+)', body).group(1), 16)
    # --- chunk ranges ---
    m = re.search(r'TEST\(Longtail, ChunkerLargeFile\)(.*?)\n}', test_cpp, re.S)
    ranges = [(int(a), int(b)) for a, b in re.findall(r'\{\s*\(const uint8_t\*\)0,\s*(\d+),\s*(\d+)\}', m.group(1))]
    assert len(ranges) == 20
    # --- LZ4 block ---
    blk = np.concatenate([np.full(1147, 13, np.uint8), np.full(4711, 77, np.uint8)])
    payload = r.compress(0, r.lz4_type, blk)
    assert len(payload) == 38

    json.dump(
        {
            "blake3_kat": {"string_plus_nul": kat_str, "hash_hex": "%016x" % kat_val, "source": "test/test.cpp:465-474"},
            "chunker_input": {"min": 16384, "avg": 65536, "max": 262144, "ranges": ranges,
                              "source": "test/test.cpp:3409-3445"},
            "lz4_block": {"runs": [[1147, 13], [4711, 77]], "payload_hex": payload.tobytes().hex(),
                          "source": "test/test.cpp:2092-2192 (size 38 pinned by the stats assert)"},
            "blake3_id": r.dll.refh_blake3_id(),
            "lz4_type": r.lz4_type,
            "zstd_types": [int(r.dll.refh_zstd_type(i)) for i in range(5)],
        },
        open(GOLDEN / "reference_tests.json", "w"),
        indent=1,
    )

    vec = {}
    # --- chunk + hash cases ---
    cases = []
    for kind, size, target, seed in [
        (0, 8 << 20, 65536, 11), (1, 8 << 20, 65536, 12), (2, 1 << 20, 65536, 13), (0, 3 << 20, 32768, 14),
        (1, 2 << 20, 16384, 15), (0, 300000, 16, 16), (0, 1 << 20, 131072, 17), (1, (4 << 20) + 12345, 65536, 18),
        (0, 49, 65536, 19), (0, 8193, 65536, 20), (0, 131073, 65536, 21), (0, 100, 16, 22),
    ]:
        data = o.synth(size, seed, kind)
        mn, av, mx = params(target)
        offs, lens, hashes = r.chunk_and_hash(data, mn, av, mx)
        name = f"chunk_k{kind}_n{size}_t{target}_s{seed}"
        vec[name + "_lens"] = lens
        vec[name + "_hashes"] = hashes
        cases.append([name, kind, size, target, seed])
        fb = r.chunk_from_buffer(data, mn, av, mx)
        vec[name + "_frombuf"] = fb
    vec["chunk_cases"] = np.array(json.dumps(cases))
    # --- BLAKE3 of prefixes ---
    data = o.synth(300000, 99, 0)
    lengths = [0, 1, 2, 3, 4, 31, 32, 33, 63, 64, 65, 127, 128, 129, 1000, 1023, 1024, 1025, 2047, 2048, 2049, 3071, 3072, 3073,
               4096, 4097, 5000, 8191, 8192, 8193, 16384, 31744, 32768, 65536, 65537, 100000, 131071, 131072, 131073, 200000,
               262144, 300000]
    vec["blake3_lengths"] = np.array(lengths, np.uint64)
    vec["blake3_hashes"] = np.array([r.blake3(data[:n]) for n in lengths], np.uint64)
    # unaligned starts
    vec["blake3_unaligned"] = np.array([r.blake3(data[s:s + 70000]) for s in range(1, 9)], np.uint64)
    # --- reference LZ4 sizes (informational: our encoder must round-trip, not match) ---
    lz = []
    for kind, size, seed in [(0, 1 << 20, 31), (1, 1 << 20, 32), (2, 1 << 20, 33), (1, 8 << 20, 34), (1, 70000, 35)]:
        d = o.synth(size, seed, kind)
        lz.append([kind, size, seed, len(r.compress(0, r.lz4_type, d))])
    vec["lz4_ref_sizes"] = np.array(lz, np.int64)
    np.savez_compressed(GOLDEN / "ref_vectors.npz", **vec)
    print("wrote", sorted(p.name for p in GOLDEN.iterdir()))


if __name__ == "__main__":
    main()
