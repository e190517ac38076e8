"""CPU: pin the oracle (C and NumPy restatements) to the golden vectors produced by running the
reference itself (tools/gen_golden.py).  Bit-exact on values, indices and residual vectors."""
import numpy as np
import pytest

DENSE = ["dense_ternary", "dense_2bit", "dense_3bit", "dense_4bit", "dense_first", "dense_m1", "dense_ragged",
         "dense_mid", "dense_mid4"]
EDGE = ["edge_dead_even", "edge_dead_odd", "edge_disjoint", "edge_ties", "edge_thresholds"]
CONV = ["conv_3x3", "conv_7x7", "conv_1x1", "conv_3x3_first"]


@pytest.mark.parametrize("case", DENSE)
def test_dense_c_oracle(oracle_mod, golden, case):
    g = golden("dense")[case]
    Q, idx, resid = oracle_mod.layer(g["W"], g["X"], g["Xq"], g["alphabet"])
    assert np.array_equal(Q.T, g["Q"])
    assert np.array_equal(idx.T, g["idx"])
    np.testing.assert_allclose(resid, g["resid"], rtol=1e-12)
    for j in range(g["W"].shape[1]):
        q, i, u = oracle_mod.neuron(g["W"][:, j], g["X"], g["Xq"], g["alphabet"])
        assert np.array_equal(u, g["U"][j])          # elementwise-exact residual


@pytest.mark.parametrize("case", [c for c in DENSE if "mid" not in c])
def test_dense_numpy_oracle(oracle_mod, golden, case):
    g = golden("dense")[case]
    for j in range(min(2, g["W"].shape[1])):
        q, i, u = oracle_mod.neuron_numpy(g["W"][:, j], g["X"], g["Xq"], g["alphabet"])
        assert np.array_equal(q, g["Q"][:, j]) and np.array_equal(i, g["idx"][:, j]) and np.array_equal(u, g["U"][j])


@pytest.mark.parametrize("case", DENSE)
def test_alphabet_radius(oracle_mod, golden, case):
    g = golden("dense")[case]
    M = len(g["alphabet"])
    alphabet, rad = oracle_mod.layer_alphabet(g["W"], np.linspace(-1, 1, M), float(g["scalar"]))
    assert rad == g["rad"] and np.array_equal(alphabet, g["alphabet"])


@pytest.mark.parametrize("case", EDGE)
def test_edge_cases(oracle_mod, golden, case):
    g = golden("edge")[case]
    Q, idx, resid = oracle_mod.layer(g["W"], g["X"], g["Xq"], g["alphabet"])
    assert np.array_equal(Q.T, g["Q"]) and np.array_equal(idx.T, g["idx"])
    np.testing.assert_allclose(resid, g["resid"], rtol=1e-12, atol=1e-300)
    q, i, u = oracle_mod.neuron_numpy(g["W"][:, 0], g["X"], g["Xq"], g["alphabet"])
    assert np.array_equal(q, g["Q"][:, 0])


def test_edge_semantics(golden):
    e = golden("edge")
    # rule (i): literal 0 although 0 is not in an even alphabet; index -1
    assert (e["edge_dead_even"]["Q"][[0, 5, 6, 39], :] == 0).all() and 0.0 not in e["edge_dead_even"]["alphabet"]
    assert (e["edge_dead_even"]["idx"][[0, 5, 6, 39], :] == -1).all()
    # ties go to the lower index
    assert e["edge_ties"]["Q"][0, 0] == 0.0 and e["edge_ties"]["Q"][0, 1] == -1.0


@pytest.mark.parametrize("case", CONV)
def test_conv_filters(oracle_mod, golden, case):
    g = golden("conv")[case]
    kh, kw, F = g["Wc"].shape
    for f in range(F):
        q, i, u = oracle_mod.neuron(g["Wc"][:, :, f].reshape(-1), g["X"], g["Xq"], g["alphabet"])
        assert np.array_equal(q.reshape(kh, kw), g["Q"][:, :, f])
        assert np.array_equal(i.reshape(kh, kw), g["idx"][:, :, f])
        np.testing.assert_allclose(np.linalg.norm(u), g["resid"][f], rtol=1e-12)


def test_bit_round(oracle_mod, golden):
    g = golden("bit_round")
    for M in (3, 4, 8, 16):
        a = g[f"alphabet_M{M}"]
        for t, want in zip(g[f"t64_M{M}"], g[f"round64_M{M}"]):
            assert a[oracle_mod.nearest(t, a)] == want == a[oracle_mod.nearest_numpy(t, a)]
        for t, want in zip(g[f"t32_M{M}"], g[f"round32_M{M}"]):
            assert a[oracle_mod.nearest(np.float64(t), a)] == want
    assert g["tie_pos"] == 0.0 and g["tie_neg"] == -1.0


@pytest.mark.parametrize("case", ["net_mlp_full", "net_mlp_partial", "net_mlp_nobias_ignore"])
def test_network_layers(oracle_mod, golden, case):
    """Every quantized layer of the reference's whole-network run, from the activations it recorded."""
    g = golden("network")[case]
    nlayers = len(g["dims"]) - 1
    for k in range(nlayers):
        if k in set(g["ignore"].tolist()):
            assert np.array_equal(g[f"Q{k}"], g[f"W{k}"])                 # untouched
            continue
        alphabet, _ = oracle_mod.layer_alphabet(g[f"W{k}"], g["alphabet"], float(g["scalar"]))
        Q, idx, _ = oracle_mod.layer(g[f"W{k}"], g[f"wX{k}"], g[f"qX{k}"], alphabet)
        assert np.array_equal(Q.T.astype(np.float32), g[f"Q{k}"])
        if bool(g["use_bias"]):
            assert np.array_equal(g[f"b{k}"], g[f"qb{k}"])                # bias carried over


def test_partial_batch_quirk_in_golden(golden):
    g = golden("network")["net_mlp_partial"]          # 40 samples, batch 16 -> 48 columns
    wX = g["wX0"]
    assert wX.shape == (12, 48)
    x = g["x"]
    assert np.array_equal(wX[:, 0:16], x[0:16].T)
    assert np.array_equal(wX[:, 16:24], x[32:40].T)     # last (partial) batch lands at 2*8
    assert np.array_equal(wX[:, 24:32], x[24:32].T)
    assert (wX[:, 32:48] == 0).all()                    # zero tail


def test_median_abs(oracle_mod):
    r = np.random.default_rng(0)
    for n in (1, 2, 5, 6, 1001, 4096):
        W = r.standard_normal(n).astype(np.float32)
        assert oracle_mod.median_abs(W) == np.median(np.abs(W))


def test_oracle_under_address_and_undefined_behaviour_sanitizers(tmp_path):
    """The C oracle (test infrastructure, CPU only -- the GPU pool offers no sanitizer) built with -fsanitize=address,undefined and driven over
    ragged layers, zero rows, m = 0, partial neuron ranges and NULL outputs (oracle/sanitize_drive.c): no report."""
    import os
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    exe = tmp_path / "drive"
    cmd = [gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-ffp-contract=off", "-fopenmp", "-o", str(exe),
           os.path.join(root, "oracle", "sanitize_drive.c"), os.path.join(root, "oracle", "gpfq_oracle.c"), "-lm"]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if built.returncode != 0:
        pytest.skip("this gcc has no sanitizer runtime: " + built.stderr[-300:])
    ran = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="2"))
    assert ran.returncode == 0 and "ok" in ran.stdout and "runtime error" not in ran.stderr and "AddressSanitizer" not in ran.stderr, ran.stderr[-2000:]
