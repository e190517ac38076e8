"""GPU: the BASELINE.json configurations at FULL size, each with (a) size-independent properties over the whole
layer -- determinism, power-of-two scale, independence of the units under permutation / splitting, fixed point --
and (b) the C oracle on a sample of the same run (>= 8 neurons / >= 4 (channel, filter) pairs), asserted bit for bit.

  cfg2  Dense(4096 -> 4096), m = 1024, ternary, scalar 3           oracle on ALL 4096 neurons (scripts/quantized_network.py:91-121)
  cfg3  VGG16 fc2 Dense(4096 -> 4096) (oracle on ALL neurons) and fc1 Dense(25088 -> 4096) (512 neurons), m = 2048, 16 levels, scalar 5
  (round 5: the whole-layer checks also assert that the run contained exact-dot-product decisions -- the certified kernels' slow path)
  cfg5  ResNet50 conv1 7x7/2 VALID on 4096 x 230 x 230 x 3 (m = 51.4 M columns: int64 indexing, the > 2^18-row
        error-bound constants, 10 GB patch matrices) and 3x3 64 -> 64 @ 56 x 56 (m = 12.8 M), ternary, scalar 3 (:185-233)

The conv samples are additionally cross-checked against the streaming kernel (the reference's verbatim flow with the
residual in HBM) on the GPU-built patch matrices -- a second implementation at full size.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from quantized_neural_networks_amd import hip as h
    h.load()
    return h


def _dense_inputs(N, m, C, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    W = torch.randn((N, C), device=dev, generator=g) / np.sqrt(N)
    G = torch.randn((N, m), device=dev, generator=g)
    X = torch.relu(G)
    Xq = torch.relu(G + 0.1 * torch.randn((N, m), device=dev, generator=g))
    return W, X, Xq


def _dense_oracle_sample(oracle_mod, W, X, Xq, alphabet, out, neurons):
    """The C oracle on a sample of neurons of the same layer: indices bit for bit, residual norms to 1e-5."""
    Wh, Xh, Xqh = W.cpu().numpy(), X.cpu().numpy(), Xq.cpu().numpy()
    for j0, j1 in neurons:
        _, io, ro = oracle_mod.layer(Wh, Xh, Xqh, alphabet, j0, j1, threads=oracle_mod.num_threads())
        ig = out["idx"][:, j0:j1].t().cpu().numpy()
        bad = int((ig != io).any(axis=1).sum())
        assert bad == 0, f"{bad} of {j1 - j0} neurons differ from the oracle"
        np.testing.assert_allclose(out["resid"][j0:j1].cpu().numpy(), ro, rtol=1e-5)


def _dense_properties(layer, W, X, Xq, alphabet, base, M, fixed_point=True):
    dev = W.device
    again = layer.quantize_dense(W, X, Xq, alphabet)
    assert torch.equal(base["idx"], again["idx"]) and torch.equal(base["resid"], again["resid"])          # determinism
    a32 = torch.from_numpy(alphabet.astype(np.float32)).to(dev)
    assert int(base["idx"].min()) >= 0 and int(base["idx"].max()) < M
    assert torch.equal(base["Q"], a32[base["idx"].long()])
    s = 2.0 ** -5                                                                                          # scale
    scaled = layer.quantize_dense(W * s, X, Xq, alphabet * s)
    assert torch.equal(scaled["idx"], base["idx"]) and torch.equal(scaled["resid"], base["resid"] * s)
    C = W.shape[1]                                                                                         # independence
    perm = torch.randperm(C, device=dev, generator=torch.Generator(device=dev).manual_seed(11))
    permuted = layer.quantize_dense(W[:, perm].contiguous(), X, Xq, alphabet)
    assert torch.equal(permuted["idx"], base["idx"][:, perm])
    cut = C // 3
    lo = layer.quantize_dense(W[:, :cut].contiguous(), X, Xq, alphabet)
    hi = layer.quantize_dense(W[:, cut:].contiguous(), X, Xq, alphabet)
    assert torch.equal(torch.cat([lo["idx"], hi["idx"]], dim=1), base["idx"])
    if fixed_point:                                                                                        # fixed point
        k = torch.randint(0, M, W.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(12))
        fixed = layer.quantize_dense(a32[k].contiguous(), X, X, alphabet)
        assert torch.equal(fixed["idx"].long(), k) and float(fixed["resid"].abs().max()) == 0.0


def _whole_layer_vs_oracle(hip, oracle_mod, layer, W, X, Xq, alphabet, out, j0=0, j1=None):
    """EVERY neuron of [j0, j1) against the C oracle (OpenMP over all host cores), indices bit for bit and residual norms to
    1e-5 -- and the same launch through the raw binding must have taken exact-dot-product decisions (the slow path of the
    certified kernels): a whole-layer check that provably contains them.  Returns (seconds, fallbacks)."""
    import time
    j1 = W.shape[1] if j1 is None else j1
    Wh, Xh, Xqh = W.cpu().numpy(), X.cpu().numpy(), Xq.cpu().numpy()
    t0 = time.perf_counter()
    _, io, ro = oracle_mod.layer(Wh, Xh, Xqh, alphabet, j0, j1, threads=oracle_mod.num_threads())
    secs = time.perf_counter() - t0
    ig = out["idx"][:, j0:j1].t().cpu().numpy()
    bad = int((ig != io).any(axis=1).sum())
    assert bad == 0, f"{bad} of {j1 - j0} neurons differ from the oracle"
    np.testing.assert_allclose(out["resid"][j0:j1].cpu().numpy(), ro, rtol=1e-5)
    raw = hip.quantize_neurons(X, Xq, hip.neuron_major(W.contiguous(), j0, j1), alphabet)
    assert torch.equal(raw["idx"].t(), out["idx"][:, j0:j1])
    return secs, hip.exact_fallbacks(raw)


def test_cfg2_dense_whole_layer_vs_oracle(hip, oracle_mod):
    """The headline layer against the oracle, ALL 4096 neurons = 16.8 M decisions (2-3 s of oracle on the GPU box's host
    cores), slow-path decisions included (18 of them in the bench's layer; asserted > 0 here)."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    W, X, Xq = _dense_inputs(4096, 1024, 4096, dev, 21)
    alphabet, rad = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    alphabet_o, rad_o = oracle_mod.layer_alphabet(W.cpu().numpy(), np.linspace(-1, 1, 3), 3)
    assert rad == rad_o and np.array_equal(alphabet, alphabet_o)
    out = layer.quantize_dense(W, X, Xq, alphabet)
    assert "gpfq_blk_kernel" in hip.last_dense_kernel()          # the default kernel of this shape is the one checked
    secs, fb = _whole_layer_vs_oracle(hip, oracle_mod, layer, W, X, Xq, alphabet, out)
    print(f"cfg2 whole layer: oracle {secs:.1f} s on {oracle_mod.num_threads()} threads, exact fallbacks {fb}")
    assert fb > 0, "the whole-layer run held no slow-path decision: the check does not cover the exact fallback"


def test_cfg3_vgg16_fc2_full_size(hip, oracle_mod):
    """VGG16 fc2 at full size: properties + ALL 4096 neurons against the oracle (2 x cfg2's oracle work)."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    W, X, Xq = _dense_inputs(4096, 2048, 4096, dev, 31)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 16), 5)
    out = layer.quantize_dense(W, X, Xq, alphabet)
    _dense_properties(layer, W, X, Xq, alphabet, out, 16)
    secs, fb = _whole_layer_vs_oracle(hip, oracle_mod, layer, W, X, Xq, alphabet, out)
    print(f"cfg3 fc2 whole layer: oracle {secs:.1f} s, exact fallbacks {fb}")
    assert fb > 0, "the whole-layer run held no slow-path decision"


def test_cfg3_vgg16_fc1_full_size(hip, oracle_mod):
    """Keras VGG16 `fc1`: 25088 -> 4096 (102.8 M weights, 25088 sequential steps per neuron), m = 2048, 16 levels.
    Oracle on 512 neurons (12.8 M decisions; the whole layer would be a minute of host time): the first and the last 256."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    W, X, Xq = _dense_inputs(25088, 2048, 4096, dev, 32)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 16), 5)
    out = layer.quantize_dense(W, X, Xq, alphabet)
    _dense_properties(layer, W, X, Xq, alphabet, out, 16)
    fb = 0
    for j0, j1 in ((0, 256), (3840, 4096)):
        secs, f = _whole_layer_vs_oracle(hip, oracle_mod, layer, W, X, Xq, alphabet, out, j0, j1)
        fb += f
        print(f"cfg3 fc1 neurons {j0}..{j1}: oracle {secs:.1f} s, exact fallbacks {f}")
    # the oracle-checked launches themselves took exact-dot-product decisions (30 of them when this was written: seeded inputs, a fixed
    # order of additions) -- the comparison covers the slow path of the narrow shapes ...
    assert fb > 0, "the oracle-checked launches held no slow-path decision"
    # ... and the whole layer's own launch (the wide-layer shape, 4096 neurons) took some too and gives the tensor the layer driver returned
    raw = hip.quantize_neurons(X, Xq, hip.neuron_major(W.contiguous(), 0, W.shape[1]), alphabet, want_values=False)
    assert torch.equal(raw["idx"].t(), out["idx"]) and hip.exact_fallbacks(raw) > 0


# ---- conv -------------------------------------------------------------------------------------------------------
def _conv_inputs(n, H, Wd, cin, cout, k, dev, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    act_w = torch.rand((n, H, Wd, cin), device=dev, generator=g)
    act_q = torch.relu(act_w + 0.05 * torch.randn((n, H, Wd, cin), device=dev, generator=g))
    W = torch.randn((k, k, cin, cout), device=dev, generator=g) / k
    return act_w, act_q, W


def _host_gib_available():
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / (1 << 20)
    except OSError:
        pass
    return 0.0


def _conv_sample(hip, oracle_mod, act_w, act_q, W, alphabet, out, k, stride, padding, pairs, host_gib_needed):
    """(channel, filter) pairs of the same run against the streaming kernel (verbatim flow, residual in HBM) and
    against the C oracle, both on the GPU-built patch matrices of the channel."""
    Wh = W.cpu().numpy()
    K = k * k
    for c, filters in pairs:
        Pw = hip.extract_patches(act_w, c, (k, k), (stride, stride), (1, 1), padding)
        Pq = hip.extract_patches(act_q, c, (k, k), (stride, stride), (1, 1), padding)
        Wt = W[:, :, c, :].reshape(K, -1).t()[filters].contiguous()             # [F][K], row-major filters (:215)
        ref = hip.quantize_neurons(Pw, Pq, Wt, alphabet, path=hip.GPFQ_PATH_STREAM)
        got = out["idx"][:, :, c, :].reshape(K, -1).t()[filters]
        assert torch.equal(got, ref["idx"]), f"channel {c}: Gram path differs from the streaming kernel"
        if _host_gib_available() < host_gib_needed:
            pytest.skip(f"host has < {host_gib_needed} GiB available for the oracle's copy of the patch matrices")
        Pwh, Pqh = Pw.cpu().numpy(), Pq.cpu().numpy()
        del Pw, Pq
        Wf = np.ascontiguousarray(Wh[:, :, c, :].reshape(K, -1)[:, filters])    # Keras layout [K][F] for oracle.layer
        _, io, _ = oracle_mod.layer(Wf, Pwh, Pqh, alphabet, 0, len(filters), threads=len(filters))
        assert np.array_equal(got.cpu().numpy(), io), f"channel {c}: (channel, filter) pairs differ from the oracle"
        del Pwh, Pqh


def _conv_whole_tensor(hip, oracle_mod, act_w, act_q, W, alphabet, out, k, stride, padding, oracle_channels, oracle_filters):
    """EVERY (channel, filter) pair of the layer's index tensor against the streaming kernel -- the verbatim element-wise flow with the
    residual in HBM (scripts/quantized_network.py:185-233), itself compared with the oracle below and throughout the suite -- on the
    GPU-built patch matrices of each channel (full-size conv parity was a sample of 4-8 pairs of up to 262 144); and
    len(oracle_channels) x oracle_filters pairs (>= 64) of the same tensor against the C oracle."""
    K = k * k
    cin, cout = W.shape[2], W.shape[3]
    Wt_all = W.permute(2, 3, 0, 1).reshape(cin, cout, K).contiguous()           # [Cin][F][K]: row-major filters (:215)
    got_all = out["idx"].permute(2, 3, 0, 1).reshape(cin, cout, K)
    Pw = Pq = None
    bad = 0
    for c in range(cin):
        Pw = hip.extract_patches(act_w, c, (k, k), (stride, stride), (1, 1), padding, out=Pw)
        Pq = Pw if act_q is act_w else hip.extract_patches(act_q, c, (k, k), (stride, stride), (1, 1), padding, out=Pq)
        ref = hip.quantize_neurons(Pw, Pq, Wt_all[c], alphabet, path=hip.GPFQ_PATH_STREAM, want_values=False, want_resid=False)
        bad += int((ref["idx"] != got_all[c]).any(dim=1).sum())
        if c in oracle_channels and _host_gib_available() >= 4:
            fs = np.linspace(0, cout - 1, oracle_filters).astype(int)
            Wf = np.ascontiguousarray(Wt_all[c].cpu().numpy()[fs].T)            # Keras layout [K][F] for oracle.layer
            _, io, _ = oracle_mod.layer(Wf, Pw.cpu().numpy(), Pq.cpu().numpy(), alphabet, 0, len(fs), threads=min(len(fs), oracle_mod.num_threads()))
            assert np.array_equal(got_all[c].cpu().numpy()[fs], io), f"channel {c}: (channel, filter) pairs differ from the oracle"
    assert bad == 0, f"{bad} of {cin * cout} (channel, filter) walks differ from the streaming kernel"
    return cin * cout


def _conv_slack_ab(hip, layer, W, act_w, act_q, alphabet, base, kw):
    """The WHOLE index tensor (every (channel, filter) pair: 1 K - 262 K walks) with the Gram path's certification bound
    doubled (`gram_slack_log2` = 1: wider than any bound the path ever shipped with) and multiplied by 8: a decision the
    production bound certified wrongly would have to survive bounds 2x and 8x as wide too, or show up here as a
    difference -- a chain flagged under the wider bound is decided from its exact dot products."""
    try:
        for s in (1, 3):
            hip.set_option("gram_slack_log2", s)
            wide = layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)
            assert torch.equal(wide["idx"], base["idx"]), f"gram_slack_log2 = {s}: the index tensor moved"
            assert torch.equal(wide["Q"], base["Q"])
    finally:
        hip.set_option("gram_slack_log2", 0)


def _conv_properties(layer, W, act_w, act_q, alphabet, base, M, kw):
    dev = W.device
    assert torch.equal(base["idx"], layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)["idx"])        # determinism
    s = 2.0 ** 4
    assert torch.equal(layer.quantize_conv2d(W * s, act_w, act_q, alphabet * s, **kw)["idx"], base["idx"])  # scale
    cout = W.shape[3]
    pf = torch.randperm(cout, device=dev, generator=torch.Generator(device=dev).manual_seed(41))
    perm = layer.quantize_conv2d(W[:, :, :, pf].contiguous(), act_w, act_q, alphabet, **kw)                # independence
    assert torch.equal(perm["idx"], base["idx"][:, :, :, pf])
    half = cout // 2
    lo = layer.quantize_conv2d(W[..., :half].contiguous(), act_w, act_q, alphabet, **kw)
    hi = layer.quantize_conv2d(W[..., half:].contiguous(), act_w, act_q, alphabet, **kw)
    assert torch.equal(torch.cat([lo["idx"], hi["idx"]], dim=3), base["idx"])
    a32 = torch.from_numpy(alphabet.astype(np.float32)).to(dev)                                            # fixed point
    kk = torch.randint(0, M, W.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(42))
    fixed = layer.quantize_conv2d(a32[kk].contiguous(), act_w, act_w, alphabet, **kw)
    assert torch.equal(fixed["idx"].long(), kk)


def test_cfg5_resnet50_conv1_full_size(hip, oracle_mod):
    """ResNet50 conv1: 7x7 / 2, VALID on the zero-padded 230 x 230 input, 3 -> 64, 4096 images: 51.4 M columns."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    act_w, act_q, W = _conv_inputs(4096, 230, 230, 3, 64, 7, dev, 51)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    kw = dict(strides=(2, 2), padding="VALID", rate=(1, 1), want_resid=False)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)
    assert tuple(out["idx"].shape) == (7, 7, 3, 64)
    assert hip.patch_out_dim(230, 7, 2, 1, False) == 112
    _conv_properties(layer, W, act_w, act_q, alphabet, out, 3, kw)
    _conv_slack_ab(hip, layer, W, act_w, act_q, alphabet, out, kw)
    # fewer channels than GPUs (3 < 8): the records over eight unequal image shards, summed in rank order and in reverse -- what an
    # all-reduce may do --, then every rank's second half.  The row norms come from a fixed-order pass over the activations
    # (launch_canonical_norms), so the whole tensor equals the one-call result whatever the order
    bounds = [0, 500, 1012, 1536, 2048, 2500, 3072, 3584, 4096]
    recs, negs = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        rec_, neg_ = hip.conv_channel_records(hip.channel_planes(act_w[lo:hi].contiguous(), 0, 3),
                                              hip.channel_planes(act_q[lo:hi].contiguous(), 0, 3), (7, 7), (2, 2), (1, 1), "VALID")
        recs.append(rec_); negs.append(neg_)
    neg = torch.stack(negs).max(dim=0).values
    fwd = recs[0].clone()
    for r_ in recs[1:]:
        fwd += r_
    bwd = recs[-1].clone()
    for r_ in recs[-2::-1]:
        bwd += r_
    pw_all, pq_all = hip.channel_planes(act_w, 0, 3), hip.channel_planes(act_q, 0, 3)
    Wt_all = W.permute(2, 3, 0, 1).reshape(3, 64, 49).contiguous()
    want = out["Q"].permute(2, 3, 0, 1).reshape(3, 64, 49)
    for rec in (fwd, bwd):
        idx = torch.empty((3, 64, 49), dtype=hip.index_dtype(len(alphabet)), device=dev)
        Qs = torch.empty((3, 64, 49), dtype=torch.float32, device=dev)
        unc = torch.zeros((3, 64), dtype=torch.int32, device=dev)
        hip.conv_channels_from_records(rec, neg, pw_all, pq_all, Wt_all, alphabet, (7, 7), (2, 2), (1, 1), "VALID", idx, Qs, unc)
        assert int(unc.sum()) == 0 and torch.equal(Qs, want)
    del pw_all, pq_all, recs
    # 49 x 51.4 M x 4 B = 10 GB per patch matrix: two on the device, two copies on the host for the oracle
    _conv_sample(hip, oracle_mod, act_w, act_q, W, alphabet, out, 7, 2, "VALID", [(1, [0, 21, 42, 63])], host_gib_needed=48)


def test_cfg5_resnet50_conv1_signed_input(hip, oracle_mod):
    """ResNet50 conv1 as the reference's ImageNet driver would feed it (quantize_pretrained_imagenet.py:12: `resnet_preprocess_input`
    handed to ImageNetSequence): caffe-style images -- BGR, ImageNet channel means subtracted, so SIGNED values in about
    [-124, 152] -- zero-padded by 3 to 230 x 230, and BOTH networks see the same tensor (conv1 is the first layer: wX is qX,
    :478-481).  With negative activations the absolute inner products of the certification come from Cauchy-Schwarz, not from
    the Gram entries: a looser bound, more stopped chains, more repair -- parity and the layer time under exactly
    that regime.  Oracle + streaming kernel on four (channel, filter) pairs of two channels, the x 2 / x 8 slack A/B on the
    whole tensor, `reruns` and the layer time printed."""
    import time
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(53)
    img = torch.rand((4096, 224, 224, 3), device=dev, generator=g) * 255.0
    img -= torch.tensor([103.939, 116.779, 123.68], device=dev)
    act = torch.nn.functional.pad(img, (0, 0, 3, 3, 3, 3))           # ZeroPadding2D(3): the border is exactly zero
    del img
    assert tuple(act.shape) == (4096, 230, 230, 3) and float(act.min()) < -100.0
    W = torch.randn((7, 7, 3, 64), device=dev, generator=g) / 7
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    kw = dict(strides=(2, 2), padding="VALID", rate=(1, 1), want_resid=False)
    out = layer.quantize_conv2d(W, act, act, alphabet, **kw)                    # act_q IS act_w, as for a first layer
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    again = layer.quantize_conv2d(W, act, act, alphabet, **kw)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    assert torch.equal(again["idx"], out["idx"])
    print(f"conv1 on signed (caffe-preprocessed) images, first-layer form: {ms:.2f} ms, reruns {int(out['reruns'])}")
    assert int(out["reruns"]) <= 8
    _conv_slack_ab(hip, layer, W, act, act, alphabet, out, kw)
    # ... and with DIFFERENT analog / quantized inputs of the same signed kind (what a signed layer deeper in a network would see)
    act_q = act + 0.5 * torch.randn(act.shape, device=dev, generator=g) * (act != 0)
    out2 = layer.quantize_conv2d(W, act, act_q, alphabet, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    layer.quantize_conv2d(W, act, act_q, alphabet, **kw)
    torch.cuda.synchronize()
    print(f"conv1 on signed images, analog != quantized inputs: {(time.perf_counter() - t0) * 1e3:.2f} ms, reruns {int(out2['reruns'])}")
    _conv_slack_ab(hip, layer, W, act, act_q, alphabet, out2, kw)
    _conv_sample(hip, oracle_mod, act, act_q, W, alphabet, out2, 7, 2, "VALID", [(0, [3, 40])], host_gib_needed=48)
    del act_q, out2
    _conv_sample(hip, oracle_mod, act, act, W, alphabet, out, 7, 2, "VALID", [(2, [0, 63])], host_gib_needed=48)


def test_cfg5_resnet50_conv3x3_56_full_size(hip, oracle_mod):
    """ResNet50 conv2_x 3x3 SAME 64 -> 64 @ 56 x 56, 4096 images: 12.8 M columns, 4096 (channel, filter) pairs."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    act_w, act_q, W = _conv_inputs(4096, 56, 56, 64, 64, 3, dev, 52)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    kw = dict(strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)
    assert int(out["reruns"]) <= 4
    _conv_properties(layer, W, act_w, act_q, alphabet, out, 3, kw)
    _conv_slack_ab(hip, layer, W, act_w, act_q, alphabet, out, kw)
    _conv_sample(hip, oracle_mod, act_w, act_q, W, alphabet, out, 3, 1, "SAME", [(0, [0, 1, 62, 63]), (63, [5, 17, 33, 60])],
                 host_gib_needed=8)
