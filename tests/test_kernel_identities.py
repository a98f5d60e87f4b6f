"""Arithmetic identities the HIP kernels rely on, re-checked on the CPU (no GPU, no library: the constants are restated here with the
place that uses them)."""
import numpy as np


def test_lz4_length_bytes_of_16_bit_lengths():
    """k_lz4.hip lz4_len_bytes16: the number of extra length bytes of an LZ4 literal / match length (lib/lz4/ext/lz4.c:1140-1160: 255-runs
    behind a token nibble of 15) is  len >= 15 ? (len - 15) / 255 + 1 : 0 ; the lane parser computes it for lengths below 65 536 as
    (len + 240) * 0x8081 >> 23 -- no branch, a 24-bit multiply."""
    l = np.arange(0, 65536, dtype=np.uint64)
    want = np.where(l >= 15, (l - np.minimum(l, 15)) // 255 + 1, 0)
    want[l < 15] = 0
    got = ((l + 240) * 0x8081) >> 23
    assert (got == want).all()
    assert int(((l + 240) * 0x8081).max()) < 1 << 32  # the product of v_mul_u32_u24 fits its 32 bits


def test_padded_window_rows_are_conflict_free_for_the_lanes_own_positions():
    """k_lz4.hip, padded LDS window: rows of 32 dwords at a pitch of 32 + LZ4_ROW_DUP = 35; the 32 lanes of a half-wave stand 16 dwords
    apart (64-byte sub-units) and must land on 32 different banks."""
    pitch = 35
    banks = {((16 * j) // 32 * pitch + (16 * j) % 32) % 32 for j in range(32)}
    assert len(banks) == 32


def test_adaptive_state_shift_register_equals_the_counter_it_replaced():
    """lz4_lane_parse2: 'three probe rounds without a hit at an unaligned position stop the one-byte steps, the first such hit brings them
    back' was a counter and a flag (round 5, first version); it is a shift register on the scalar unit now: bit k = such a hit k rounds ago
    (bit 0 also the unit's start), dense while the low three bits are not all zero."""
    rng = np.random.default_rng(3)
    for _ in range(200):
        hits = rng.random(60) < rng.choice([0.02, 0.2, 0.6])
        quiet, dense_w = 0, True  # the counter formulation
        qhist, dense_now = 1, True
        for h in hits:
            assert dense_w == dense_now
            if h:
                quiet, dense_w = 0, True
            else:
                quiet += 1
                if quiet == 3:
                    dense_w = False
            qhist = ((qhist << 1) | int(h)) & 0xFFFFFFFF
            dense_now = (qhist & 7) != 0
