"""CPU: host-side decisions of `vitlens_hip.ops` that need no GPU - the k-slice choice of the token-major weight-gradient GEMM
(`tn_splits`) against the preconditions its C entry point enforces (vl_gemm.hip: every slice >= 4 steps of 64 tokens)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-lens_amd"))


def _c_side_accepts(nk, splits):
    ln = (nk + splits - 1) // splits
    eff = (nk + ln - 1) // ln
    return ln >= 4 and nk - (eff - 1) * ln >= 4 and eff <= splits


def test_tn_split_choice_is_accepted_by_the_kernel_and_fills_the_chip():
    from vitlens_hip.ops import tn_splits
    # the C3 micro-batch (257 * 256 tokens = 1028 steps): c_fc / c_proj 64 tiles, in_proj 48, out_proj 16
    assert tn_splits(1028, 64) == 4 and tn_splits(1028, 48) == 4 and tn_splits(1028, 16) == 16
    assert tn_splits(1028, 256) == 1                       # enough tiles: no split
    assert tn_splits(64, 16) == 0                          # 16 tiles would need 16 slices of 4 steps: below the 16-step floor
    assert tn_splits(20, 1024) == 1 and tn_splits(15, 1024) == 0
    for nk in range(1, 1400, 7):
        for tiles in (1, 3, 12, 16, 48, 64, 100, 192, 256, 1024):
            s = tn_splits(nk, tiles)
            if s:
                assert s in (1, 2, 4, 8, 16) and tiles * s >= 192 and nk // s >= 16
                assert _c_side_accepts(nk, s), (nk, tiles, s)
                smaller = [q for q in (1, 2, 4, 8, 16) if q < s and tiles * q >= 192 and nk // q >= 16 and _c_side_accepts(nk, q)]
                assert not smaller, (nk, tiles, s, smaller)
