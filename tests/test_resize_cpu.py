"""Properties of the oracle's cv::resize restatement (oracle/orc_resize.py; "parity unpinned" against OpenCV itself,
see its header): what any bilinear resize with OpenCV's coefficient scheme must satisfy."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc_resize  # noqa: E402


def u8(x):
    return np.rint(x * 255).astype(np.uint8).transpose(0, 2, 3, 1)


def test_identity_constant_and_area_shortcut():
    rng = np.random.default_rng(0)
    f = rng.integers(0, 256, (2, 9, 13, 3), dtype=np.uint8)
    assert np.array_equal(u8(orc_resize.resize_u8_to_chw(f, 13, 9)), f), "same-size resize must reproduce the frame"
    c = np.full((1, 7, 11, 3), 93, np.uint8)
    for (w, h) in ((5, 3), (22, 14), (16, 16)):
        assert np.all(u8(orc_resize.resize_u8_to_chw(c, w, h)) == 93), "a constant frame stays constant"
    # exact 2x shrink = OpenCV's fast-area shortcut: rounded mean of each 2x2 block
    g = rng.integers(0, 256, (1, 8, 12, 1), dtype=np.uint8)
    want = (g[:, 0::2, 0::2].astype(int) + g[:, 0::2, 1::2] + g[:, 1::2, 0::2] + g[:, 1::2, 1::2] + 2) >> 2
    assert np.array_equal(u8(orc_resize.resize_u8_to_chw(g, 6, 4)), want.astype(np.uint8))
    # channel swap is a pure permutation
    a = orc_resize.resize_u8_to_chw(f, 20, 10, swap_rb=False)
    b = orc_resize.resize_u8_to_chw(f, 20, 10, swap_rb=True)
    assert np.array_equal(a[:, ::-1], b)
    # monotone in the input (bilinear weights are non-negative)
    lo = orc_resize.resize_u8_to_chw(f // 2, 17, 5)
    hi = orc_resize.resize_u8_to_chw(f, 17, 5)
    assert np.all(lo <= hi)
