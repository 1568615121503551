"""The "uni" LDS image of the backward attention kernels (viewformer_amd/csrc/attention_train_bf16.hip: dma_uni_piece, round 6): ONE 8 KB image of a
64 x 64 bf16 tile serves the row reads (ds_read_b128 A fragments) AND the transposing reads (ds_read_b64_tr_b16).  This file restates the kernel's
address arithmetic in Python and checks, on the CPU, what the kernel relies on:
  * the DMA's source permutation fills every 16-byte cell of the image exactly once;
  * the row-read and the transposing-read addresses the lanes compute are the cells of the (row, chunk) they mean to read;
  * both read patterns are free of LDS bank conflicts (16 lanes x 16 bytes of a ds_read_b128 pass in 16 distinct 16-byte slots of the 256-byte bank line;
    a 32-lane pass of the transposing read inside ONE bank line).
The GPU tests (tests/test_train.py, tests/test_hip_ring_stress.py) check the results; this one documents and pins the layout."""
import itertools

import numpy as np


def uni_m(j):
    return (j & 1) | (((j >> 2) & 1) << 1)


def cell(r, c):
    """16-byte cell of tile row r (0..63), 16-byte chunk c (0..7) of its 128-byte row"""
    return 16 * ((r >> 2) * 2 + (c >> 2)) + 4 * (r & 3) + ((c & 3) ^ uni_m(r >> 2))


def dma_source(pi, lane):
    """(row, chunk) that lane `lane` of DMA piece `pi` fetches; the LDS side is lane-linear: it lands in cell 64 pi + lane"""
    j, d, slot = 2 * pi + (lane >> 5), (lane >> 4) & 1, lane & 15
    return 4 * j + (slot >> 2), 4 * d + ((slot & 3) ^ uni_m(j))


def row_read_byte(lane, u, ks):
    """byte address of the ds_read_b128 of lane `lane` for the tile's 32-row half u and k-step ks (the kernels' `uni_row + u * 4096 + ...`)"""
    half, l31 = lane >> 5, lane & 31
    uni_row = 512 * (l31 >> 2) + 64 * (l31 & 3)
    return uni_row + u * 4096 + (ks >> 1) * 256 + ((((ks & 1) * 2 + half) ^ uni_m(l31 >> 2)) << 4)


def tr_read_byte(lane, u, ks2, d, second):
    """byte address of one ds_read_b64_tr_b16 of lane `lane`: rows 32 u + 16 ks2 (+ 8 for the second read of the pair) + 4 half + ((lane & 15) >> 2),
    feature half d, 8-byte piece lane & 3 of the 32-byte column block (lane >> 4) & 1 (the kernels' `uni_tr[ks2] + u * 4096 + ks2 * 2048 + d * 256`, gap 1024)"""
    half = lane >> 5
    cq = ((lane >> 4) & 1) * 2 + ((lane & 3) >> 1)
    m = half if ks2 == 0 else (half | 2)
    base = 512 * half + 64 * ((lane & 15) >> 2) + 16 * (cq ^ m) + 8 * (lane & 1)
    return base + u * 4096 + ks2 * 2048 + d * 256 + (1024 if second else 0)


def test_cell_is_a_bijection_and_the_dma_fills_it():
    cells = {cell(r, c) for r in range(64) for c in range(8)}
    assert cells == set(range(512))
    for pi in range(8):
        for lane in range(64):
            r, c = dma_source(pi, lane)
            assert 0 <= r < 64 and 0 <= c < 8
            assert cell(r, c) == 64 * pi + lane                     # the lane-linear LDS side of the DMA


def test_row_reads_hit_their_row_and_chunk_without_bank_conflicts():
    for u, ks in itertools.product(range(2), range(4)):
        for lane in range(64):
            half, l31 = lane >> 5, lane & 31
            assert row_read_byte(lane, u, ks) == 16 * cell(32 * u + l31, 2 * ks + half)
        # the hardware serves a ds_read_b128 in 16-lane passes over rows {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} of each half-wave
        for half in range(2):
            for rows in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
                slots = {(row_read_byte(32 * half + r, u, ks) % 256) // 16 for r in rows}
                assert len(slots) == 16


def test_transposing_reads_hit_their_rows_and_stay_inside_one_bank_line():
    for u, ks2, d, second in itertools.product(range(2), range(2), range(2), (False, True)):
        lines = {0: set(), 1: set()}
        for lane in range(64):
            half = lane >> 5
            row = 32 * u + 16 * ks2 + (8 if second else 0) + 4 * half + ((lane & 15) >> 2)
            col_byte = 64 * d + ((lane >> 4) & 1) * 32 + (lane & 3) * 8          # byte inside the tile's 128-byte row
            want = 16 * cell(row, col_byte >> 4) + (col_byte & 15)
            got = tr_read_byte(lane, u, ks2, d, second)
            assert got == want
            lines[half].add(got // 256)
        assert all(len(v) == 1 for v in lines.values())              # a 32-lane pass = 4 rows x 64 bytes = one 256-byte bank line
        # ... and the 32 lanes of a pass cover that line exactly once
        for half in range(2):
            b = sorted(tr_read_byte(32 * half + l, u, ks2, d, second) % 256 for l in range(32))
            assert b == list(range(0, 256, 8))


def test_m_separates_the_row_groups_of_a_pass():
    """the property the conflict-freeness of the row reads rests on: m() is injective on {0, 3, 5, 6} and on {1, 2, 4, 7} (r >> 2 of the two 16-lane passes)"""
    for grp in ([0, 3, 5, 6], [1, 2, 4, 7]):
        assert len({uni_m(j) for j in grp}) == 4
        assert len({uni_m(j + 8) for j in grp}) == 4                 # (the tile's second 32-row half)
    assert np.all([uni_m(j) < 4 for j in range(16)])
