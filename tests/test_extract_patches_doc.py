"""Row a7 (per-channel im2col, scripts/quantized_network.py:158-179) pinned to TensorFlow's DOCUMENTED behaviour:
tests/golden/extract_patches_doc.npz holds the worked examples of the `tf.image.extract_patches` docstring and cases
worked out by hand from TensorFlow's documented SAME / VALID rules (tools/gen_patches_golden.py; every expected array is
a literal there).  Checked against them:

* CPU: tests/_im2col_ref.py, the NumPy statement the other conv tests use as their reference;
* GPU: gpfq_extract_patches (explicit patch matrices), the Gram records the implicit-im2col kernels form straight from
  the channel planes (gpfq_conv_channel_records), and the conv layer driver end to end against the C oracle run on the
  DOCUMENTED patches.
"""
import os

import numpy as np
import pytest

from _im2col_ref import patches as ref_patches

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "extract_patches_doc.npz")


def _cases():
    z = np.load(GOLDEN)
    names = sorted({k.split("__")[0] for k in z.files})
    out = []
    for n in names:
        c = {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(n + "__")}
        c["name"] = n
        out.append(c)
    return out


CASES = _cases()
IDS = [c["name"] for c in CASES]


def _matrix(tf_layout):
    """TensorFlow's [B, oh, ow, kh*kw] -> the reference's transposed patch matrix (kh*kw, B*oh*ow) (:175-179, :789-797)."""
    k = tf_layout.shape[3]
    return np.ascontiguousarray(tf_layout.reshape(-1, k).T)


def _args(c):
    kh, kw = (int(v) for v in c["ksizes"])
    sh, sw = (int(v) for v in c["strides"])
    rh, rw = (int(v) for v in c["rates"])
    return kh, kw, sh, sw, rh, rw, "SAME" if int(c["same"]) else "VALID"


def test_fixture_has_the_published_examples():
    names = set(IDS)
    assert {"tfdoc_3x3_stride5_valid", "tfdoc_3x3_stride5_rate2_valid"} <= names and len(names) >= 8
    doc = next(c for c in CASES if c["name"] == "tfdoc_3x3_stride5_valid")
    assert doc["expected"].shape == (1, 2, 2, 9) and doc["expected"][0, 1, 1].tolist() == [56, 57, 58, 66, 67, 68, 76, 77, 78]


@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_numpy_reference_matches_documented_patches(c):
    kh, kw, sh, sw, rh, rw, padding = _args(c)
    got = ref_patches(c["images"], 0, kh, kw, sh, sw, rh, rw, padding)
    want = _matrix(c["expected"])
    assert got.shape == want.shape
    if "mask" in c:
        m = _matrix(c["mask"]).astype(bool)
        assert np.array_equal(got[m], want[m])
    else:
        assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("c", CASES, ids=IDS)
def test_gpu_extract_patches_matches_documented_patches(c):
    import torch
    from quantized_neural_networks_amd import hip
    kh, kw, sh, sw, rh, rw, padding = _args(c)
    dev = torch.device("cuda")
    # the documented image as channel 1 of a 3-channel NHWC tensor (the kernel picks one channel out of Cin)
    img = torch.from_numpy(c["images"]).to(dev)
    act = torch.cat([img * 0 - 7, img, img * 3], dim=3).contiguous()
    got = hip.extract_patches(act, 1, (kh, kw), (sh, sw), (rh, rw), padding).cpu().numpy()
    want = _matrix(c["expected"])
    assert got.shape == want.shape
    assert hip.patch_out_dim(c["images"].shape[1], kh, sh, rh, padding == "SAME") == c["expected"].shape[1]
    assert hip.patch_out_dim(c["images"].shape[2], kw, sw, rw, padding == "SAME") == c["expected"].shape[2]
    if "mask" in c:
        m = _matrix(c["mask"]).astype(bool)
        assert np.array_equal(got[m], want[m])
    else:
        assert np.array_equal(got, want)


FULL = [c for c in CASES if "mask" not in c]


@pytest.mark.gpu
@pytest.mark.parametrize("c", FULL, ids=[c["name"] for c in FULL])
def test_gpu_implicit_im2col_records_match_documented_patches(c):
    """The Gram records the conv kernels accumulate straight from the channel planes (no patch matrix) equal the Gram
    matrices of the DOCUMENTED patches: G1[t][s] = <Xq_t, X_s>, G2[t][s] = <Xq_t, Xq_s> (lower triangle), nx2[s] = <X_s, X_s>.
    Small integers: exact in float64."""
    import torch
    from quantized_neural_networks_amd import hip
    kh, kw, sh, sw, rh, rw, padding = _args(c)
    dev = torch.device("cuda")
    K = kh * kw
    reps = 3                                                       # the same documented image three times: sums triple
    img = torch.from_numpy(c["images"]).to(dev)[..., 0]           # [1][H][W]
    planes_w = img.repeat(reps, 1, 1).unsqueeze(0).contiguous()   # [nch = 1][n][H][W]
    planes_q = (2.0 * planes_w).contiguous()                      # zeros of the padding stay zeros
    try:
        rec, neg = hip.conv_channel_records(planes_w, planes_q, (kh, kw), (sh, sw), (rh, rw), padding)
    except hip.GpfqError as e:
        pytest.skip(f"no plane kernel for this shape: {e}")
    rec = rec.cpu().numpy()[0]
    P = _matrix(c["expected"]).astype(np.float64)
    G = reps * (P @ P.T)
    G1 = rec[:K * K * 2].reshape(K, K, 2)[:, :, 0]
    G2 = rec[:K * K * 2].reshape(K, K, 2)[:, :, 1]
    nx = rec[K * K * 2:]
    low = np.tril(np.ones((K, K), dtype=bool))
    assert np.array_equal(G1[low], (2.0 * G)[low]) and np.array_equal(G2[low], (4.0 * G)[low])
    assert np.array_equal(nx, np.diag(G)) and int(neg.cpu()[0]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("c", FULL, ids=[c["name"] for c in FULL])
def test_gpu_conv_layer_on_documented_images_matches_oracle_on_documented_patches(c, oracle_mod):
    """Whole conv driver (planes -> records -> decisions) on the documented images; the oracle walks the documented
    patch matrices.  (channel, filter) pairs bit for bit (:185-233)."""
    import torch
    from quantized_neural_networks_amd import layer
    kh, kw, sh, sw, rh, rw, padding = _args(c)
    dev = torch.device("cuda")
    K, F, reps = kh * kw, 5, 64
    r = np.random.default_rng(5)
    scale = np.float32(1.0 / c["images"].max())
    img = c["images"] * scale                                       # [1][H][W][1] in (0, 1]
    jit = (1.0 + 0.25 * r.random((reps, 1, 1, 1))).astype(np.float32)
    act_w = (img * jit).astype(np.float32)                          # 64 differently scaled copies: patches scale alike
    act_q = (act_w * np.float32(0.875)).astype(np.float32)
    W = (r.standard_normal((kh, kw, 1, F)) / K).astype(np.float32)
    Wd = torch.from_numpy(W).to(dev)
    alphabet, _ = layer.layer_alphabet(Wd, np.linspace(-1, 1, 8), 4.0)
    out = layer.quantize_conv2d(Wd, torch.from_numpy(act_w).to(dev), torch.from_numpy(act_q).to(dev), alphabet,
                                strides=(sh, sw), padding=padding, rate=(rh, rw), want_resid=False)
    P1 = _matrix(c["expected"]) * scale                             # documented patches of the unit image
    Pw = np.concatenate([(P1 * j).astype(np.float32) for j in jit.reshape(-1)], axis=1)
    Pq = (Pw * np.float32(0.875)).astype(np.float32)
    Q = out["Q"].cpu().numpy()
    for f in range(F):
        qo, _, _ = oracle_mod.neuron(W[:, :, 0, f].reshape(-1), Pw, Pq, alphabet)
        assert np.array_equal(Q[:, :, 0, f].reshape(-1), qo.astype(np.float32)), f"filter {f}"
