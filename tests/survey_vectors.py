"""The xorshift64 streams of SURVEY.md §8(c) and the values the survey probed on the REFERENCE for them (chunk count, first
chunks as (offset, length, BLAKE3-64)): reference-produced known answers that pin oracle and kernels alike."""
from functools import lru_cache

import numpy as np

SEED = 0x9E3779B97F4A7C15
# (bytes, target_chunk_size) -> (chunk count, [(offset, len, hash)...] of the first chunks)
EXPECTED = {
    (1000, 65536): (1, [(0, 1000, 0xA3CC33B545916DBF)]),
    (1 << 20, 32768): (59, [(0, 56265, 0x6758F9CAB9EA154A)]),
    (64 << 20, 65536): (2087, [(0, 14477, 0x068FE0BF0FE45E0D), (14477, 8405, 0x6A367DD25058A42B), (22882, 9685, 0xA52C91FD560ED461),
                               (32567, 18426, 0xEC7F0588BCBC69C1)]),
}


@lru_cache(maxsize=None)
def xorshift_stream(nbytes: int) -> np.ndarray:
    """s ^= s << 13; s ^= s >> 7; s ^= s << 17; the state AFTER each step as 8 little-endian bytes."""
    words = (nbytes + 7) // 8
    out = np.empty(words, np.uint64)
    s, mask = SEED, (1 << 64) - 1
    for i in range(words):
        s ^= (s << 13) & mask
        s ^= s >> 7
        s ^= (s << 17) & mask
        out[i] = s
    return out.view(np.uint8)[:nbytes].copy()
