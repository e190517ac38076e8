"""Round 6: the Dense layer driver with nothing crossing to the host -- the layer alphabet rad * linspace(-1, 1, M)
(scripts/quantized_network.py:544-545) formed and kept in DEVICE memory (gpfq_layer_alphabet_device), the kernel reading the Keras
kernel itself and writing Q / the indices in the layout set_weights takes (:562, :570) -- and the fail-safe handling of the block
kernel's deferred failures (a cluster exchange that times out; a degenerate radius): noticed BEFORE the result is used, logged, and the
layer rerun through the classic kernels, as the reference logs and re-raises at once (:563-565).  Everything against the oracle, bit for bit."""
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RESID_RTOL = 1e-5      # north star: 1e-5 relative on the float residual norms (observed ~1e-15)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(scope="module")
def hip():
    from quantized_neural_networks_amd import hip as h
    h.load()
    return h


@pytest.fixture(scope="module")
def layer():
    from quantized_neural_networks_amd import layer as l
    return l


def _synthetic(N, m, C, seed=0):
    W = (np.random.default_rng(seed).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
    G = np.random.default_rng(seed + 1).standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    Xq = np.maximum(G + 0.1 * np.random.default_rng(seed + 2).standard_normal((N, m)), 0).astype(np.float32)
    return W, X, Xq


@pytest.mark.parametrize("levels,scalar", [(3, 3), (2, 1.5), (4, 2), (16, 5), (15, 4), (64, 7)])
def test_device_alphabet_is_the_reference_product(hip, layer, oracle_mod, levels, scalar):
    """rad and every member: the float64 products of :544-545, bit for bit, without the median leaving the device."""
    W = (np.random.default_rng(levels).standard_normal((301, 77)) / 17).astype(np.float32)
    unit = np.linspace(-1, 1, levels)
    want_alphabet, want_rad = oracle_mod.layer_alphabet(W, unit, scalar)
    d = layer.layer_alphabet_device(_dev(W), unit, scalar)
    raw = d.buf.cpu().numpy()
    rad = raw[:8].view(np.float64)[0]
    members = raw[128:128 + 8 * levels].view(np.float64)
    assert rad == want_rad and d.rad() == want_rad
    assert np.array_equal(members, want_alphabet)
    assert np.array_equal(d.values(), want_alphabet)
    host_alphabet, host_rad = layer.layer_alphabet(_dev(W), unit, scalar)
    assert host_rad == want_rad and np.array_equal(host_alphabet, want_alphabet)


# (N, C, m, levels): widths that are no multiple of the 8- / 16-neuron output groups, every neurons-per-workgroup class, rows on either side
# of the shape thresholds, the cluster form (1537+ samples in a narrow layer, 3073+ anywhere)
SHAPES = [(61, 70, 512, 3), (40, 33, 1000, 16), (37, 16, 300, 4), (29, 5, 700, 3), (23, 1, 900, 2), (33, 130, 1024, 3), (21, 600, 768, 3),
          (19, 1100, 1024, 15), (17, 2100, 1024, 3), (26, 40, 1500, 3), (22, 200, 2048, 16), (18, 24, 2100, 3), (15, 70, 3100, 4), (13, 9, 5008, 8)]


@pytest.mark.parametrize("N,C,m,levels", SHAPES)
def test_dense_layer_device_alphabet_vs_oracle(hip, layer, oracle_mod, N, C, m, levels):
    W, X, Xq = _synthetic(N, m, C, seed=N + C)
    Xq[N - 2] = 0                                                 # a dead row: rule (i), the literal 0
    unit = np.linspace(-1, 1, levels)
    alphabet, rad = oracle_mod.layer_alphabet(W, unit, 3)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    Wd, Xd, Xqd = _dev(W), _dev(X), _dev(Xq)
    assert hip.dense_layer_supported(N, m, C, unit)
    d = layer.layer_alphabet_device(Wd, unit, 3)
    out = layer.quantize_dense(Wd, Xd, Xqd, d)
    assert "gpfq_blk_kernel" in hip.last_dense_kernel()
    assert hip.call_status(out) == 0
    assert out["Q"].shape == (N, C) and out["idx"].shape == (N, C) and out["idx"].dtype == torch.int8
    assert np.array_equal(out["idx"].cpu().numpy(), idx.T)       # Keras layout [N][C], written by the kernel's own flush
    assert np.array_equal(out["Q"].cpu().numpy(), Q.T.astype(np.float32))
    np.testing.assert_allclose(out["resid"].cpu().numpy(), resid, rtol=RESID_RTOL)
    # ... and the host alphabet's path (neuron-major copy, kernel, assembly pass) gives the same tensors
    ref = layer.quantize_dense(Wd, Xd, Xqd, alphabet)
    assert torch.equal(ref["Q"], out["Q"]) and torch.equal(ref["idx"], out["idx"]) and torch.equal(ref["resid"], out["resid"])
    assert d.rad() == rad


def test_dense_layer_shard_of_a_layer_neuron_major(hip, layer, oracle_mod):
    """Neurons [lo, hi) of a wider layer (what one rank of a process group quantizes), neuron-major indices for the all-gather, then the
    device-alphabet assembly: equal to the whole layer's columns."""
    N, C, m = 45, 150, 800
    W, X, Xq = _synthetic(N, m, C, seed=3)
    unit = np.linspace(-1, 1, 3)
    alphabet, _ = oracle_mod.layer_alphabet(W, unit, 2)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    Wd, Xd, Xqd = _dev(W), _dev(X), _dev(Xq)
    d = layer.layer_alphabet_device(Wd, unit, 2)
    shards = []
    for lo, hi in ((0, 64), (64, 128), (128, 150)):
        r = hip.quantize_dense_layer(Xd, Xqd, Wd, d, lo, hi, keras_out=False, want_values=False)
        assert hip.call_status(r) == 0
        assert np.array_equal(r["idx"].cpu().numpy(), idx[lo:hi])
        np.testing.assert_allclose(r["resid"].cpu().numpy(), resid[lo:hi], rtol=RESID_RTOL)
        shards.append(r["idx"])
    for bits in (8, 2):
        gathered = torch.cat(shards)
        if bits == 2:
            gathered, b = hip.pack_indices(gathered, 3)
            assert b == 2
        Qk, Ik = hip.assemble_kernel_device(gathered.contiguous(), d, bits=bits, N=N)
        assert np.array_equal(Ik.cpu().numpy(), idx.T) and np.array_equal(Qk.cpu().numpy(), Q.T.astype(np.float32))
    # a shard written straight into the whole layer's Keras-layout tensors leaves the other columns alone
    r = hip.quantize_dense_layer(Xd, Xqd, Wd, d, 64, 128)
    assert np.array_equal(r["idx"][:, 64:128].cpu().numpy(), idx[64:128].T)
    assert np.array_equal(r["Q"][:, 64:128].cpu().numpy(), Q[64:128].T.astype(np.float32))


def test_degenerate_radius_is_caught_on_the_device_and_the_layer_rerun(hip, layer, oracle_mod):
    """More than half of the kernel is zero: median(|W|) = 0, the alphabet is {-0, 0, 0} (:544-545) -- no arithmetic progression.  The
    host cannot know (the median never left the device): the kernel writes nothing and raises the call's alphabet word, and the layer
    driver reruns the layer with the host alphabet; the reference's result for such a layer (q = first member = -0.0) comes out."""
    N, C, m = 31, 20, 600
    W, X, Xq = _synthetic(N, m, C, seed=9)
    W[np.random.default_rng(1).random(W.shape) < 0.6] = 0
    unit = np.linspace(-1, 1, 3)
    alphabet, rad = oracle_mod.layer_alphabet(W, unit, 3)
    assert rad == 0
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    Wd, Xd, Xqd = _dev(W), _dev(X), _dev(Xq)
    d = layer.layer_alphabet_device(Wd, unit, 3)
    raw = hip.quantize_dense_layer(Xd, Xqd, Wd, d)
    assert hip.call_status(raw) == hip.GPFQ_ERR_ALPHABET
    assert "radius" in hip.load().gpfq_last_error().decode()
    out = layer.quantize_dense(Wd, Xd, Xqd, d)
    assert np.array_equal(out["idx"].cpu().numpy(), idx.T)
    assert np.array_equal(out["Q"].cpu().numpy(), Q.T.astype(np.float32))
    np.testing.assert_allclose(out["resid"].cpu().numpy(), resid, rtol=RESID_RTOL)


def test_shapes_without_a_block_kernel_fall_back_to_the_host_alphabet(hip, layer, oracle_mod):
    N, C, m = 25, 12, 200                                          # rows of at most 256 samples: the row-group kernels
    W, X, Xq = _synthetic(N, m, C, seed=4)
    unit = np.linspace(-1, 1, 4)
    alphabet, _ = oracle_mod.layer_alphabet(W, unit, 2)
    _, idx, _ = oracle_mod.layer(W, X, Xq, alphabet)
    assert not hip.dense_layer_supported(N, m, C, unit)
    d = layer.layer_alphabet_device(_dev(W), unit, 2)
    out = layer.quantize_dense(_dev(W), _dev(X), _dev(Xq), d)
    assert np.array_equal(out["idx"].cpu().numpy(), idx.T)
    with pytest.raises(hip.GpfqError):
        hip.quantize_dense_layer(_dev(X), _dev(Xq), _dev(W), d)


# ---- the cluster form's fail-safe -------------------------------------------------------------------------------
@pytest.fixture
def faulty_cluster(hip):
    """One slice of the first cluster never publishes: every exchange of that cluster times out (after 40 ms instead of 3 s)."""
    hip.set_option("blk_cluster_fault", 1)
    hip.set_option("blk_cluster_timeout_ms", 40)
    yield
    hip.set_option("blk_cluster_fault", 0)
    hip.set_option("blk_cluster_timeout_ms", 3000)
    hip.set_option("sync_errors", 0)


def test_forced_exchange_timeout_is_reported_by_the_c_abi(hip, oracle_mod, faulty_cluster):
    N, C, m = 14, 24, 3500                                         # four slices
    W, X, Xq = _synthetic(N, m, C, seed=2)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    r = hip.quantize_neurons(_dev(X), _dev(Xq), _dev(W.T), alphabet, path=1)
    assert "cluster form" in hip.last_dense_kernel()
    assert hip.call_status(r) == hip.GPFQ_ERR_CLUSTER_TIMEOUT and hip.cluster_timeouts(r) != 0
    # synchronous error reporting: the call itself returns the code
    hip.set_option("sync_errors", 1)
    with pytest.raises(hip.GpfqError, match="timed out"):
        hip.quantize_neurons(_dev(X), _dev(Xq), _dev(W.T), alphabet, path=1)
    hip.set_option("sync_errors", 0)
    # without the fault the same call is clean
    hip.set_option("blk_cluster_fault", 0)
    r = hip.quantize_neurons(_dev(X), _dev(Xq), _dev(W.T), alphabet, path=1)
    assert hip.call_status(r) == 0
    _, idx, _ = oracle_mod.layer(W, X, Xq, alphabet)
    assert np.array_equal(r["idx"].cpu().numpy(), idx)


@pytest.mark.parametrize("device_alphabet", [False, True])
def test_forced_exchange_timeout_falls_back_to_the_classic_kernels(hip, layer, oracle_mod, faulty_cluster, device_alphabet):
    """The layer driver reads the status BEFORE anything uses Q, logs the layer and reruns it with blk_cluster = 0: the result is the
    oracle's, and the option is back to its default afterwards."""
    N, C, m = 14, 40, 3500
    W, X, Xq = _synthetic(N, m, C, seed=6)
    unit = np.linspace(-1, 1, 3)
    alphabet, _ = oracle_mod.layer_alphabet(W, unit, 3)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    Wd, Xd, Xqd = _dev(W), _dev(X), _dev(Xq)
    a = layer.layer_alphabet_device(Wd, unit, 3) if device_alphabet else alphabet
    logged = []
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        out = layer.quantize_dense(Wd, Xd, Xqd, a, log=logged.append)
    assert any("timed out" in str(w.message) for w in caught) and any("timed out" in msg for msg in logged)
    assert "cluster form" not in hip.last_dense_kernel()
    assert np.array_equal(out["idx"].cpu().numpy(), idx.T)
    assert np.array_equal(out["Q"].cpu().numpy(), Q.T.astype(np.float32))
    np.testing.assert_allclose(out["resid"].cpu().numpy(), resid, rtol=RESID_RTOL)
    # the next (healthy) layer takes the cluster form again
    hip.set_option("blk_cluster_fault", 0)
    out = layer.quantize_dense(Wd, Xd, Xqd, a)
    assert "cluster form" in hip.last_dense_kernel()
    assert np.array_equal(out["idx"].cpu().numpy(), idx.T)


def test_class_surface_survives_a_timed_out_exchange(hip, oracle_mod, faulty_cluster):
    """quantize_network() on a two-layer MLP whose first layer takes the cluster form: the fault is logged with the layer's index and the
    network that comes out equals the one of a healthy run -- no garbage reaches set_weights or the next layer's activations."""
    from quantized_neural_networks_amd import keras_shim as ks
    from quantized_neural_networks_amd import quantized_network as qn

    class ListLogger:
        def __init__(self):
            self.lines = []

        def info(self, msg):
            self.lines.append(msg)

    n, d0, d1, d2 = 3300, 12, 24, 5                                # 3300 samples: four slices
    x = np.random.default_rng(0).standard_normal((n, d0)).astype(np.float32)

    def run():
        net = ks.Sequential([ks.Dense(d1, activation="relu", input_shape=(d0,)), ks.Dense(d2)], seed=3)
        logger = ListLogger()
        q = qn.QuantizedNeuralNetwork(network=net, batch_size=n, get_data=qn.MNISTSequence(x, np.zeros((n, 1)), n),
                                      logger=logger, bits=np.log2(3), alphabet_scalar=2)
        q.quantize_network()
        return [np.asarray(w.cpu() if isinstance(w, torch.Tensor) else w).copy() for w in q.quantized_net.get_weights()], logger.lines

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        faulty, lines = run()
    assert any("timed out" in line and "Layer 0" in line for line in lines), [l for l in lines if "Layer" in l][:5]
    hip.set_option("blk_cluster_fault", 0)
    healthy, lines = run()
    assert not any("timed out" in line for line in lines)
    for a, b in zip(faulty, healthy):
        assert np.array_equal(a, b)


def test_partitioned_chip_keeps_the_classic_shapes(hip, oracle_mod):
    """blk_chip_ok = 0 (what the library answers itself when the device is not the whole 8 x 32-CU chip): no cluster form anywhere."""
    N, C, m = 12, 9, 6000
    W, X, Xq = _synthetic(N, m, C, seed=8)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    _, idx, _ = oracle_mod.layer(W, X, Xq, alphabet)
    try:
        hip.set_option("blk_chip_ok", 0)
        r = hip.quantize_neurons(_dev(X), _dev(Xq), _dev(W.T), alphabet, path=1)
        assert "cluster form" not in hip.last_dense_kernel()
        assert np.array_equal(r["idx"].cpu().numpy(), idx)
    finally:
        hip.set_option("blk_chip_ok", -1)
    r = hip.quantize_neurons(_dev(X), _dev(Xq), _dev(W.T), alphabet, path=1)
    assert "cluster form" in hip.last_dense_kernel()
    assert np.array_equal(r["idx"].cpu().numpy(), idx)


def test_two_streams_of_cluster_launches_are_serialised_and_correct(hip, oracle_mod):
    """Cluster-form launches on two streams at once: the library orders them (one in flight per device); both results are the oracle's."""
    N, C, m = 10, 300, 4100
    W, X, Xq = _synthetic(N, m, C, seed=12)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    _, idx, _ = oracle_mod.layer(W, X, Xq, alphabet)
    Xd, Xqd, Wt = _dev(X), _dev(Xq), _dev(W.T)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    results = []
    for rep in range(3):
        for s in streams:
            with torch.cuda.stream(s):
                results.append(hip.quantize_neurons(Xd, Xqd, Wt, alphabet, path=1))
    torch.cuda.synchronize()
    for r in results:
        assert hip.cluster_timeouts(r) == 0
        assert np.array_equal(r["idx"].cpu().numpy(), idx)


@pytest.mark.parametrize("N,C,m,levels", [(3, 20, 512, 3), (37, 70, 1000, 16), (64, 600, 1024, 3), (41, 33, 2048, 4), (19, 10, 3000, 3), (13, 9, 5008, 8),
                                          (29, 2100, 768, 3), (45, 24, 1536, 2)])
def test_record_prepass_in_runs_equals_one_record_per_workgroup(hip, oracle_mod, N, C, m, levels):
    """Round 6: the block kernel's record pre-pass takes eight consecutive records per workgroup (every row read three times instead of
    eighteen).  Same layer through both forms (option blk_prep_run): indices, values, residual vectors and the count of exact-fallback
    decisions are equal, and equal to the oracle's."""
    W, X, Xq = _synthetic(N, m, C, seed=N * m)
    alphabet, _ = oracle_mod.layer_alphabet(W, np.linspace(-1, 1, levels), 3)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    outs = []
    try:
        for run in (0, 8, 4, 13):                                  # (4 .. 16: runs of that many records whatever the walk's length; 1, the default, only from 2048 steps)
            hip.set_option("blk_prep_run", run)
            with hip.option("blk_cluster", 0):
                r = hip.quantize_neurons(_dev(X), _dev(Xq), _dev(W.T), alphabet, want_u=True, path=1)
            assert "gpfq_blk_kernel" in hip.last_dense_kernel()
            outs.append((r["idx"].cpu().numpy(), r["Q"].cpu().numpy(), r["u"].cpu().numpy(), r["resid"].cpu().numpy(), hip.exact_fallbacks(r)))
    finally:
        hip.set_option("blk_prep_run", 1)
    for other in outs[1:]:
        for a, b in zip(outs[0][:3], other[:3]):
            assert np.array_equal(a, b)
        assert np.array_equal(other[0], idx) and np.array_equal(other[1], Q.astype(np.float32))
        np.testing.assert_allclose(other[3], resid, rtol=RESID_RTOL)
        if m <= 1024:
            assert outs[0][4] == other[4]                          # one chunk per thread: the same order of additions, the same records bit for bit


@pytest.mark.parametrize("N,C,m,levels", [(61, 70, 512, 3), (40, 33, 1000, 16), (33, 600, 1024, 2), (18, 24, 2100, 3), (15, 70, 3100, 3), (29, 5, 200, 3)])
def test_overlapped_layer_driver_equals_the_one_stream_form(hip, layer, oracle_mod, N, C, m, levels):
    """layer.quantize_dense_layer(overlap=True): the median and the alphabet on a second stream beside the row norms and the record pre-pass
    (gpfq_dense_layer_prepare / _run; for symmetric alphabets the records are scaled in place once the alphabet exists), with every way of
    saying when the kernel W was complete -- unknown (a fork wait), an event, "long ago" (no wait).  Same tensors as the one-stream call and
    as the oracle, also for a shape without a block-pipelined kernel (200 samples: the fallback inside)."""
    W, X, Xq = _synthetic(N, m, C, seed=N + m)
    Xq[1] = 0
    unit = np.linspace(-1, 1, levels)
    alphabet, rad = oracle_mod.layer_alphabet(W, unit, 3)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    Wd, Xd, Xqd = _dev(W), _dev(X), _dev(Xq)
    ready = torch.cuda.Event()
    ready.record()
    torch.cuda.synchronize()                                       # (kernel_ready=True below is then the truth)
    outs = [layer.quantize_dense_layer(Wd, Xd, Xqd, unit, 3, overlap=ov, kernel_ready=kr)
            for ov, kr in ((True, None), (False, None), (True, True), (True, ready), (True, True))]
    for out in outs:
        assert np.array_equal(out["idx"].cpu().numpy(), idx.T) and np.array_equal(out["Q"].cpu().numpy(), Q.T.astype(np.float32))
        np.testing.assert_allclose(out["resid"].cpu().numpy(), resid, rtol=RESID_RTOL)
        assert out["alphabet"].rad() == rad
    assert torch.equal(outs[0]["resid"], outs[1]["resid"])


def test_class_surface_runs_dense_layers_on_the_device_path(hip, layer, monkeypatch):
    """QuantizedNeuralNetwork.quantize_network() (scripts/quantized_network.py:576-590) takes its Dense layers through the same driver
    as bench.py's step: device-resident alphabet from the prefetched median, the Keras kernel read in place, no assembly pass and -- the
    host knowing the radius is good and the launch not being the cluster form -- no status wait.  The network, the radii and the residual
    norms equal those of the host-alphabet path, bit for bit."""
    from quantized_neural_networks_amd import keras_shim as ks
    from quantized_neural_networks_amd import quantized_network as qn

    n, d0, d1, d2 = 600, 40, 33, 7
    x = np.random.default_rng(0).standard_normal((n, d0)).astype(np.float32)
    calls = dict(device=0, assemble=0, status=0)
    real_layer, real_asm, real_status = hip.quantize_dense_layer, hip.assemble_kernel, hip.call_status

    def run(device_path):
        net = ks.Sequential([ks.Dense(d1, activation="relu", input_shape=(d0,)), ks.Dense(d2)], seed=5)
        q = qn.QuantizedNeuralNetwork(network=net, batch_size=n, get_data=qn.MNISTSequence(x, np.zeros((n, 1)), n),
                                      bits=2, alphabet_scalar=2.5)
        if not device_path:
            q._layer_alphabet_device = lambda layer_idx, rad: None
        q.quantize_network()
        ws = [np.asarray(w.cpu() if isinstance(w, torch.Tensor) else w).copy() for w in q.quantized_net.get_weights()]
        stats = {k: (float(v["rad"]), np.asarray(v["resid"]), np.asarray(v["idx"])) for k, v in q.last_layer_stats.items()}
        return ws, stats

    monkeypatch.setattr(hip, "quantize_dense_layer", lambda *a, **k: (calls.__setitem__("device", calls["device"] + 1), real_layer(*a, **k))[1])
    monkeypatch.setattr(hip, "assemble_kernel", lambda *a, **k: (calls.__setitem__("assemble", calls["assemble"] + 1), real_asm(*a, **k))[1])
    monkeypatch.setattr(hip, "call_status", lambda *a, **k: (calls.__setitem__("status", calls["status"] + 1), real_status(*a, **k))[1])
    dev_ws, dev_stats = run(True)
    assert calls == dict(device=2, assemble=0, status=0), calls
    host_ws, host_stats = run(False)
    assert calls["device"] == 2 and calls["assemble"] == 2
    assert len(dev_ws) == len(host_ws)
    for a, b in zip(dev_ws, host_ws):
        assert np.array_equal(a, b)
    assert dev_stats.keys() == host_stats.keys()
    for k in dev_stats:
        assert dev_stats[k][0] == host_stats[k][0]
        np.testing.assert_allclose(dev_stats[k][1], host_stats[k][1], rtol=RESID_RTOL)
        assert np.array_equal(dev_stats[k][2], host_stats[k][2])


@pytest.mark.parametrize("N,C,m,levels,run", [(37, 70, 1000, 16, 8), (64, 600, 1024, 3, 5), (2100, 40, 1024, 3, 1), (33, 20, 772, 4, 16), (21, 9, 1022, 3, 8),
                                              (41, 33, 2048, 4, 8), (13, 9, 5008, 8, 8), (50, 17, 1024, 3, 0),
                                              (156, 2050, 292, 4, 8), (40, 33, 512, 3, 8), (25, 70, 768, 3, 4), (31, 18, 260, 2, 16)])
def test_row_norms_inside_the_record_prepass(hip, layer, oracle_mod, N, C, m, levels, run):
    """gpfq_quantize_dense_layer without the caller's row norms forms them inside the record pre-pass where that reproduces the row-norm
    kernel's sums bit for bit (runs of records, rows padded to exactly 1024 samples, m % 4 == 0: the first four cases) and by that kernel
    otherwise (a ragged row, long rows, the cluster form, one record per workgroup, rows of fewer than 769 samples -- whole wavefronts of the
    pre-pass then sit the row out, which the fuzzer found at 292 samples).  Option blk_prep_norms = 0 always takes the kernel: same indices,
    values, residual norms, the same count of exact-fallback decisions -- and the call's counter block is zeroed either way."""
    W, X, Xq = _synthetic(N, m, C, seed=N + 3 * m)
    Xq[N // 2] = 0                                                 # a zero row: norm 0, the reference's first rule
    unit = np.linspace(-1, 1, levels)
    alphabet, rad = oracle_mod.layer_alphabet(W, unit, 2.5)
    Q, idx, resid = oracle_mod.layer(W, X, Xq, alphabet)
    Wd, Xd, Xqd = _dev(W), _dev(X), _dev(Xq)
    dalpha = layer.layer_alphabet_device(Wd, unit, 2.5)
    outs = []
    try:
        hip.set_option("blk_prep_run", run)
        for norms in (1, 0, 1):
            hip.set_option("blk_prep_norms", norms)
            r = hip.quantize_dense_layer(Xd, Xqd, Wd, dalpha, keras_out=False, want_values=True)
            assert hip.call_status(r) == 0
            outs.append((r["idx"].cpu().numpy(), r["Q"].cpu().numpy(), r["resid"].cpu().numpy(), hip.exact_fallbacks(r)))
        # the caller's own norms (gpfq_row_norms): the third way to the same records
        r = hip.quantize_dense_layer(Xd, Xqd, Wd, dalpha, nrm32=hip.row_norms(Xqd), keras_out=False, want_values=True)
        outs.append((r["idx"].cpu().numpy(), r["Q"].cpu().numpy(), r["resid"].cpu().numpy(), hip.exact_fallbacks(r)))
    finally:
        hip.set_option("blk_prep_run", 1)
        hip.set_option("blk_prep_norms", 1)
    for o in outs:
        assert np.array_equal(o[0], idx) and np.array_equal(o[1], Q.astype(np.float32))
        np.testing.assert_allclose(o[2], resid, rtol=RESID_RTOL)
        assert np.array_equal(o[2], outs[0][2]) and o[3] == outs[0][3]
