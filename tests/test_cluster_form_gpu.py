"""The block kernel's cluster form (gpfq_blk.hip, round 5): rows beyond 5120 samples are cut into 1024-sample slices, one workgroup
each, whose decision wavefronts exchange their partial dot products once per slot.  Same contract as every dense kernel
(scripts/quantized_network.py:91-121): bit-exact indices, values and residual vectors against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RESID_RTOL = 1e-5


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def hip():
    from quantized_neural_networks_amd import hip as h
    h.load()
    return h


def _synthetic(N, m, C, seed=0):
    W = (np.random.default_rng(seed).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
    G = np.random.default_rng(seed + 1).standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    Xq = np.maximum(G + 0.1 * np.random.default_rng(seed + 2).standard_normal((N, m)), 0).astype(np.float32)
    return W, X, Xq


def _run(hip, W, X, Xq, alphabet, want_u=True):
    r = hip.quantize_neurons(_dev(X), _dev(Xq), _dev(W.T), alphabet, want_u=want_u, path=1)
    torch.cuda.synchronize()
    assert hip.cluster_timeouts(r) == 0
    return r, {k: (v.cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in r.items()}


# (threshold, m): the option moves the threshold down so that clusters of two and three slices run too; 9000 / 16384: more slices than one
# batch of the exchange's reads; 5121 and 8192: as dispatched
@pytest.mark.parametrize("threshold,m", [(1024, 1100), (1024, 2049), (2048, 3000), (1, 5121), (1, 6000), (1, 8192), (1, 9000), (1, 16384), (1, 20001), (1, 28672)])
@pytest.mark.parametrize("C,levels,nl", [(5, 3, 0), (70, 2, 4), (200, 4, 2), (33, 16, 1), (70, 3, 0), (33, 16, 4)])
def test_cluster_form_vs_oracle(hip, oracle_mod, threshold, m, C, levels, nl):
    N = 23 if m > 6000 else 37                                    # (not a multiple of the block of four steps)
    W, X, Xq = _synthetic(N, m, C, seed=m + C)
    Xq[N - 2] = 0                                                 # a dead row: rule (i)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, levels), 2)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    try:
        hip.set_option("blk_cluster", threshold)
        hip.set_option("blk_cluster_nl", nl)                       # neurons per lane of a workgroup: by width (0: 4 neurons per workgroup here), 1, 2, 4
        r, out = _run(hip, W, X, Xq, alphabet)
        name = hip.last_dense_kernel()
    finally:
        hip.set_option("blk_cluster", 1)
        hip.set_option("blk_cluster_nl", 0)
    assert "cluster form" in name, name
    assert np.array_equal(out["idx"], idx)
    assert np.array_equal(out["Q"], Q.astype(np.float32))
    np.testing.assert_allclose(out["resid"], resid, rtol=RESID_RTOL)
    for j in (0, C - 1):
        _, _, u = oracle_mod.neuron(W[:, j], X, Xq, alphabet)
        assert np.array_equal(out["u"][j], u), j                 # the residual vector, bit for bit, across the slices


def test_cluster_form_off_keeps_the_wide_kernel(hip, oracle_mod):
    N, m, C = 19, 6000, 9
    W, X, Xq = _synthetic(N, m, C, seed=5)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    _, idx, _ = oracle_mod.layer(W, X, Xq, alphabet)
    try:
        hip.set_option("blk_cluster", 0)
        _, out = _run(hip, W, X, Xq, alphabet, want_u=False)
        assert "cluster form" not in hip.last_dense_kernel()
        assert np.array_equal(out["idx"], idx)
    finally:
        hip.set_option("blk_cluster", 1)
    _, out = _run(hip, W, X, Xq, alphabet, want_u=False)
    assert "cluster form" in hip.last_dense_kernel()
    assert np.array_equal(out["idx"], idx)


@pytest.mark.parametrize("m", [5200, 8192])
def test_cluster_form_slow_path_is_exercised(hip, oracle_mod, m):
    """Weights on the alphabet's boundaries against identical activations of both networks (tests/test_hip_parity.py,
    test_role_split_slow_path_is_exercised): the look-ahead cannot certify many decisions, and the exact dot products of the slow path
    are sums over ALL slices -- a second exchange per round."""
    r0 = np.random.default_rng(29)
    N, C = 48, 40
    G = r0.standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    step = 0.125
    alphabet = step * np.arange(-3, 4, dtype=np.float64)
    W = (step / 2 * r0.integers(-7, 8, (N, C))).astype(np.float32)
    Q, idx, resid = oracle_mod.layer(W, X, X, alphabet)
    r, out = _run(hip, W, X, X, alphabet)
    assert "cluster form" in hip.last_dense_kernel()
    assert np.array_equal(out["idx"], idx) and np.array_equal(out["Q"], Q.astype(np.float32))
    assert hip.exact_fallbacks(r) > 0
    # the symmetric form ({-a, 0, a}, {-a, a}): its slow path reads the row itself from memory, at the slice's offset
    Xq = np.maximum(G + 0.25 * r0.standard_normal((N, m)), 0).astype(np.float32)
    for members in (3, 2):
        alph = step * (np.arange(-1, 2, dtype=np.float64) if members == 3 else np.array([-1.0, 1.0]))
        Ws = (step / 2 * r0.integers(-3, 4, (N, C))).astype(np.float32)
        for Xq_ in (Xq, X):
            Qs, idxs, _ = oracle_mod.layer(Ws, X, Xq_, alph)
            r, out = _run(hip, Ws, X, Xq_, alph)
            assert np.array_equal(out["idx"], idxs), members
            assert np.array_equal(out["Q"], Qs.astype(np.float32))
            if Xq_ is X:
                assert hip.exact_fallbacks(r) > 0


def test_cluster_form_full_width_layer(hip, oracle_mod):
    """Dense(1024 -> 4096) on 8192 samples, ternary: every cluster of the launch (256 x 8 workgroups: eight rounds of the chip) against
    the oracle on ALL 4096 neurons (about 5 s of oracle on the box's host threads); the launch contains slow-path decisions, whose exact
    dot products are sums over the eight slices."""
    N, m, C = 1024, 8192, 4096
    W, X, Xq = _synthetic(N, m, C, seed=11)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    _, idx, resid = oracle_mod.layer(W, X, Xq, alphabet, threads=oracle_mod.num_threads())
    r, out = _run(hip, W, X, Xq, alphabet, want_u=False)
    assert "cluster form" in hip.last_dense_kernel()
    assert np.array_equal(out["idx"], idx)
    np.testing.assert_allclose(out["resid"], resid, rtol=RESID_RTOL)
    assert hip.exact_fallbacks(r) > 0


@pytest.mark.parametrize("cmap", [0, 1])
@pytest.mark.parametrize("m", [3000, 5121, 8192])
def test_cluster_form_both_workgroup_maps(hip, oracle_mod, cmap, m):
    """Workgroup id -> (cluster, slice): inside one XCD's queue (0) or consecutive ids (1: the slices of a cluster go round the XCDs and
    the exchange crosses them); the default picks by the slice count.  Same bits either way; enough clusters for several rounds of the chip
    would take a wide layer -- here 40 clusters of 3 / 6 / 8 slices."""
    N, C = 29, 640
    W, X, Xq = _synthetic(N, m, C, seed=m + cmap)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    _, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    try:
        hip.set_option("blk_cluster", 1024)
        hip.set_option("blk_cluster_map", cmap)
        _, out = _run(hip, W, X, Xq, alphabet, want_u=False)
        assert "cluster form" in hip.last_dense_kernel()
    finally:
        hip.set_option("blk_cluster", 1)
        hip.set_option("blk_cluster_map", -1)
    assert np.array_equal(out["idx"], idx)
    np.testing.assert_allclose(out["resid"], resid, rtol=RESID_RTOL)


@pytest.mark.parametrize("m", [2049, 2500, 3072])
@pytest.mark.parametrize("levels,waves", [(3, -1), (3, 8), (16, -1), (16, 11), (2, 11)])
def test_four_slices_of_768_samples_vs_oracle(hip, oracle_mod, m, levels, waves):
    """Round 6: rows of 2049..3072 samples in layers wider than 2048 neurons run as FOUR 768-sample slices of the cluster form (64 clusters
    per round instead of 85 three-slice clusters: whole rounds) -- `<4,24,4>` workgroups with eleven sweep wavefronts for symmetric
    alphabets, eight otherwise (option blk_cluster768 forces either, 0 gives the classic one-step shape back).  Indices, values, residual
    norms and residual VECTORS against the oracle; the classic shape gives the same tensors."""
    N, C = 21, 2100
    W, X, Xq = _synthetic(N, m, C, seed=m + levels)
    Xq[N - 3] = 0                                                 # a dead row: rule (i)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, levels), 2)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    try:
        hip.set_option("blk_cluster768", waves)
        r, out = _run(hip, W, X, Xq, alphabet)
        name = hip.last_dense_kernel()
        hip.set_option("blk_cluster768", 0)
        r0, out0 = _run(hip, W, X, Xq, alphabet, want_u=False)
        name0 = hip.last_dense_kernel()
    finally:
        hip.set_option("blk_cluster768", -1)
    assert "cluster form" in name and "cluster form" not in name0
    assert np.array_equal(out["idx"], idx) and np.array_equal(out0["idx"], idx)
    assert np.array_equal(out["Q"], Q.astype(np.float32))
    np.testing.assert_allclose(out["resid"], resid, rtol=RESID_RTOL)
    for j in (0, 1033, C - 1):
        _, _, u = oracle_mod.neuron(W[:, j], X, Xq, alphabet)
        assert np.array_equal(out["u"][j], u), j                 # the residual vector, bit for bit, across the four slices


def test_four_slices_of_768_only_where_they_pay(hip, oracle_mod):
    """... and only there: 2048 neurons or fewer, or rows of at most 2048 / more than 3072 samples, keep their shapes."""
    for N, C, m, want in ((9, 2048, 2500, False), (9, 2100, 2048, False), (9, 2100, 3073, False), (9, 2100, 2500, True)):
        W, X, Xq = _synthetic(N, m, C, seed=C + m)
        alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
        _, idx, _ = oracle_mod.layer(W, X, Xq, alphabet)
        r, out = _run(hip, W, X, Xq, alphabet, want_u=False)
        ws = hip.load().gpfq_workspace_bytes(N, m, C, 1)
        assert np.array_equal(out["idx"], idx)
        # (four slices of 768 samples: 3072 padded samples per record row -- the workspace tells the shapes apart)
        four = hip.load().gpfq_workspace_bytes(N, m, C, 1) == hip.load().gpfq_workspace_bytes(N, 2500, 2100, 1)
        assert ws > 0 and (four or not want), (N, C, m)
