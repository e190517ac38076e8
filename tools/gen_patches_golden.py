#!/usr/bin/env python3
"""Golden vectors that pin the per-channel im2col to TensorFlow's DOCUMENTED behaviour.

TensorFlow is not installed in this image, so `tf.image.extract_patches` cannot be run.  What can be
committed is what TensorFlow publishes: the worked examples in the docstring of `tf.image.extract_patches`
(https://www.tensorflow.org/api_docs/python/tf/image/extract_patches, identical text in TensorFlow 2.4 ... 2.16,
the versions `requirements.txt` of the reference admits: tensorflow>=2.4.0), copied here as literal arrays, plus
cases worked out BY HAND from the padding rule TensorFlow documents for `SAME`
(https://www.tensorflow.org/api_docs/python/tf/nn#notes_on_padding_2: out = ceil(in / stride),
pad_total = max((out - 1) * stride + k_eff - in, 0), pad_before = pad_total // 2, the rest after; k_eff = k + (k-1)(rate-1)).
Every expected array below is written out literally -- none is computed by this repository's own im2col.

The reference feeds one channel at a time, `images[B, H, W, 1]`, and reshapes the result `[B, oh, ow, kh*kw]` to
`(B*oh*ow, kh*kw)` (scripts/quantized_network.py:158-179), stored transposed `(kh*kw, B*oh*ow)` (:789-797); the
fixtures keep TensorFlow's `[B, oh, ow, kh*kw]` layout and the tests apply that reshape.

    python tools/gen_patches_golden.py        -> tests/golden/extract_patches_doc.npz
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cases = {}


def case(name, images, ksizes, strides, rates, padding, expected, source):
    images = np.asarray(images, dtype=np.float32)
    expected = np.asarray(expected, dtype=np.float32)
    assert images.ndim == 4 and images.shape[3] == 1 and expected.ndim == 4
    cases[name] = dict(images=images, ksizes=np.asarray(ksizes, np.int64), strides=np.asarray(strides, np.int64),
                       rates=np.asarray(rates, np.int64), same=np.asarray(1 if padding == "SAME" else 0, np.int64),
                       expected=expected, source=np.asarray(source))


# ---- published: the docstring examples of tf.image.extract_patches ------------------------------------------------
# "images is a 1 x 10 x 10 x 1 array that contains the numbers 1 through 100"
n = 10
images10 = [[[[x * n + y + 1] for y in range(n)] for x in range(n)]]
case("tfdoc_3x3_stride5_valid", images10, (3, 3), (5, 5), (1, 1), "VALID",
     [[[[1, 2, 3, 11, 12, 13, 21, 22, 23],
        [6, 7, 8, 16, 17, 18, 26, 27, 28]],
       [[51, 52, 53, 61, 62, 63, 71, 72, 73],
        [56, 57, 58, 66, 67, 68, 76, 77, 78]]]],
     "tf.image.extract_patches docstring, first example (ksizes [1,3,3,1], strides [1,5,5,1], rates [1,1,1,1], VALID)")
case("tfdoc_3x3_stride5_rate2_valid", images10, (3, 3), (5, 5), (2, 2), "VALID",
     [[[[1, 3, 5, 21, 23, 25, 41, 43, 45],
        [6, 8, 10, 26, 28, 30, 46, 48, 50]],
       [[51, 53, 55, 71, 73, 75, 91, 93, 95],
        [56, 58, 60, 76, 78, 80, 96, 98, 100]]]],
     "tf.image.extract_patches docstring, second example (same, rates [1,2,2,1])")

# ---- worked by hand from the documented SAME rule ------------------------------------------------------------------
# 10 x 10, k 3, stride 5, SAME: out = ceil(10/5) = 2, pad_total = max((2-1)*5 + 3 - 10, 0) = 0 -> no padding at all,
# the patches are those of the VALID call
case("same_rule_3x3_stride5", images10, (3, 3), (5, 5), (1, 1), "SAME",
     [[[[1, 2, 3, 11, 12, 13, 21, 22, 23],
        [6, 7, 8, 16, 17, 18, 26, 27, 28]],
       [[51, 52, 53, 61, 62, 63, 71, 72, 73],
        [56, 57, 58, 66, 67, 68, 76, 77, 78]]]],
     "SAME rule by hand: pad_total = 0")

# 4 x 4 (even), k 3, stride 2, SAME: out = 2, pad_total = (2-1)*2 + 3 - 4 = 1 -> pad_before 0, pad_after 1
# (the asymmetric case: the zero row / column is at the BOTTOM / RIGHT only)
images4 = [[[[r * 4 + c + 1] for c in range(4)] for r in range(4)]]       # 1..16
case("same_rule_even_3x3_stride2", images4, (3, 3), (2, 2), (1, 1), "SAME",
     [[[[1, 2, 3, 5, 6, 7, 9, 10, 11],
        [3, 4, 0, 7, 8, 0, 11, 12, 0]],
       [[9, 10, 11, 13, 14, 15, 0, 0, 0],
        [11, 12, 0, 15, 16, 0, 0, 0, 0]]]],
     "SAME rule by hand: pad_before 0, pad_after 1")

# 3 x 3, k 3, stride 1, SAME: out 3, pad_total 2 -> one ring of zeros
images3 = [[[[r * 3 + c + 1] for c in range(3)] for r in range(3)]]       # 1..9
case("same_rule_3x3_stride1", images3, (3, 3), (1, 1), (1, 1), "SAME",
     [[[[0, 0, 0, 0, 1, 2, 0, 4, 5], [0, 0, 0, 1, 2, 3, 4, 5, 6], [0, 0, 0, 2, 3, 0, 5, 6, 0]],
       [[0, 1, 2, 0, 4, 5, 0, 7, 8], [1, 2, 3, 4, 5, 6, 7, 8, 9], [2, 3, 0, 5, 6, 0, 8, 9, 0]],
       [[0, 4, 5, 0, 7, 8, 0, 0, 0], [4, 5, 6, 7, 8, 9, 0, 0, 0], [5, 6, 0, 8, 9, 0, 0, 0, 0]]]],
     "SAME rule by hand: pad 1 / 1")

# 5 x 5, k 3, rate 2 (k_eff 5), stride 1, SAME: out 5, pad_total = 4 -> 2 before, 2 after; taps at offsets -2, 0, +2
images5 = [[[[r * 5 + c + 1] for c in range(5)] for r in range(5)]]       # 1..25
exp5 = np.zeros((1, 5, 5, 9), dtype=np.float32)
hand5 = {                                                                   # (oy, ox): the nine taps, written out
    (0, 0): [0, 0, 0, 0, 1, 3, 0, 11, 13],
    (0, 2): [0, 0, 0, 1, 3, 5, 11, 13, 15],
    (2, 2): [1, 3, 5, 11, 13, 15, 21, 23, 25],
    (4, 4): [13, 15, 0, 23, 25, 0, 0, 0, 0],
    (2, 0): [0, 1, 3, 0, 11, 13, 0, 21, 23],
    (1, 3): [0, 0, 0, 7, 9, 0, 17, 19, 0],
}
for (oy, ox), taps in hand5.items():
    exp5[0, oy, ox] = taps
mask5 = np.zeros((1, 5, 5, 9), dtype=np.float32)
for (oy, ox) in hand5:
    mask5[0, oy, ox] = 1
case("same_rule_3x3_rate2", images5, (3, 3), (1, 1), (2, 2), "SAME", exp5, "SAME rule by hand, dilation 2: six output positions written out")
cases["same_rule_3x3_rate2"]["mask"] = mask5                              # only the hand-worked positions are compared

# 5 x 6 (non-square), k (2, 3), strides (2, 1), SAME: rows out 3, pad_total_h = (3-1)*2 + 2 - 5 = 1 -> 0 / 1;
# cols out 6, pad_total_w = 5 + 3 - 6 = 2 -> 1 / 1
images56 = [[[[r * 6 + c + 1] for c in range(6)] for r in range(5)]]      # 1..30
exp56 = np.zeros((1, 3, 6, 6), dtype=np.float32)
hand56 = {
    (0, 0): [0, 1, 2, 0, 7, 8],
    (0, 5): [5, 6, 0, 11, 12, 0],
    (1, 2): [14, 15, 16, 20, 21, 22],
    (2, 0): [0, 25, 26, 0, 0, 0],
    (2, 5): [29, 30, 0, 0, 0, 0],
}
mask56 = np.zeros_like(exp56)
for (oy, ox), taps in hand56.items():
    exp56[0, oy, ox] = taps
    mask56[0, oy, ox] = 1
case("same_rule_2x3_strides21", images56, (2, 3), (2, 1), (1, 1), "SAME", exp56, "SAME rule by hand, non-square kernel / image / strides")
cases["same_rule_2x3_strides21"]["mask"] = mask56

# VALID with a stride that does not divide: 7 x 7, k 3, stride 2 -> out = ceil((7 - 3 + 1) / 2) = 3
images7 = [[[[r * 7 + c + 1] for c in range(7)] for r in range(7)]]       # 1..49
exp7 = np.zeros((1, 3, 3, 9), dtype=np.float32)
for oy in range(3):
    for ox in range(3):
        exp7[0, oy, ox] = [(2 * oy + ky) * 7 + (2 * ox + kx) + 1 for ky in range(3) for kx in range(3)]
case("valid_rule_3x3_stride2_7x7", images7, (3, 3), (2, 2), (1, 1), "VALID", exp7,
     "VALID by the documented rule: patch (oy, ox) starts at (2 oy, 2 ox); values follow from the 1..49 numbering")

flat = {}
for name, c in cases.items():
    for k, v in c.items():
        flat[f"{name}__{k}"] = v
out = os.path.join(ROOT, "tests", "golden", "extract_patches_doc.npz")
np.savez_compressed(out, **flat)
print("wrote", out, "with", len(cases), "cases")
