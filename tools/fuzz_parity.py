"""Randomised parity sweep on the GPU: random dense layers (all on-chip kernels, streaming, Gram path) and random conv
layers (kernel size, stride, rate, padding, image size, channel / filter counts, alphabets, sparse or signed
activations) against the CPU oracle, bit for bit.  Not part of the test suite (minutes of GPU time).
usage: fuzz_parity.py [seconds] [seed]      (FUZZ_LONG_P=0.5: more of the long-row conv cases)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from quantized_neural_networks_amd import hip, layer
from _im2col_ref import patches as ref_patches
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = torch.device("cuda")
t_end = time.time() + budget
n_dense = n_conv = n_long = 0
LONG_P = float(os.environ.get("FUZZ_LONG_P", "0.05"))   # share of conv cases with 2^18 .. 2^22 patch columns
bad = []


def activations(shape, kind):
    g = rng.standard_normal(shape)
    if kind == "relu":
        a = np.maximum(g, 0)
    elif kind == "sparse":
        a = np.maximum(g - 1.5, 0)                  # ~7 % non-zero
    elif kind == "signed":
        a = g
    else:
        a = rng.random(shape)
    return a.astype(np.float32)


while time.time() < t_end:
    bits = rng.choice([1, np.log2(3), np.log2(3), 2, 3, 4, 5, 6, 7, 8, np.log2(129)])   # up to 256 members (int16 indices beyond 64); 1 and log2(3): symmetric forms
    M = int(round(2 ** bits))
    scalar = float(rng.choice([1, 2, 3, 5]))
    if rng.random() < 0.5:
        # ---- dense ------------------------------------------------------------------------
        N = int(rng.integers(1, 200)); C = int(rng.integers(1, 60)) if rng.random() < 0.7 else int(rng.choice([rng.integers(500, 700), rng.integers(1000, 1100), rng.integers(2040, 2060)]))
        m = int(rng.choice([rng.integers(1, 300), rng.integers(300, 3000), rng.integers(3000, 5200), rng.integers(3000, 30000)]))
        if rng.random() < 0.15:                     # long walks (Gram path: one wavefront per neuron)
            N = int(rng.integers(200, 1025)); m = int(rng.integers(1, 2500))
        if rng.random() < 0.12:                     # 129..2048 neurons on rows of 257..1024 samples: the four-group matrix shapes (round 4)
            N = int(rng.integers(1, 60)); C = int(rng.choice([rng.integers(129, 400), rng.integers(1000, 1050), rng.integers(1900, 2049)])); m = int(rng.integers(257, 1025))
        if rng.random() < 0.15:                     # rows of 2049..16384 samples: the block kernel's cluster form (1024-sample slices over several workgroups)
            N = int(rng.integers(1, 60)); C = int(rng.choice([rng.integers(1, 60), rng.integers(100, 700), rng.integers(1000, 1400)])); m = int(rng.choice([rng.integers(2049, 5200), rng.integers(5121, 16385)]))
        if rng.random() < 0.12:                     # wide layers on long rows: 16 neurons per workgroup over eleven sweep wavefronts
            N = int(rng.integers(1, 40)); C = int(rng.integers(2049, 2400)); m = int(rng.integers(1025, 2049))
        if rng.random() < 0.08:                     # round 6: rows of 2049..3072 samples in layers wider than 2048 neurons: four 768-sample slices of the cluster form
            N = int(rng.integers(1, 30)); C = int(rng.integers(2049, 2300)); m = int(rng.integers(2049, 3073))
        kind = rng.choice(["relu", "sparse", "signed", "uniform"])
        X = activations((N, m), kind)
        Xq = X if rng.random() < 0.2 else (X + 0.1 * rng.standard_normal((N, m)).astype(np.float32) * (X != 0 if kind == "sparse" else 1)).astype(np.float32)
        if kind in ("relu", "sparse", "uniform"):
            Xq = np.maximum(Xq, 0).astype(np.float32)
        if rng.random() < 0.3:
            Xq[int(rng.integers(0, N))] = 0.0       # a dead row: rule (i)
        W = (rng.standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
        alphabet, _ = oracle.layer_alphabet(W, np.linspace(-1, 1, M), scalar)
        Qo, io, ro = oracle.layer(W, X, Xq, alphabet)
        path = int(rng.choice([0, 1, 2, 3])) if m <= hip.GPFQ_ONCHIP_MAX_M else int(rng.choice([0, 2, 3]))
        if N > 200:
            path = 3
        opts = {}
        if path in (0, 1) and rng.random() < 0.5:   # the role-split kernels (one step / a block of steps per slot) wherever they apply
            opts = dict(pipe=int(rng.choice([1, 2])), blk_sweep_waves=int(rng.choice([0, 8, 11])), blk_wide_groups=int(rng.choice([1, 1, 0])), blk_four_groups=int(rng.choice([1, 1, 0])),
                        blk_pair_groups=int(rng.choice([1, 1, 0])), blk_single_groups=int(rng.choice([1, 1, 0])), blk_quad_groups=int(rng.choice([2, 2, 0, 1])), blk_quad_waves=int(rng.choice([0, 0, 7, 8])),
                        variant=int(rng.choice([0, 0, 32])),
                        blk_cluster=int(rng.choice([1, 1, 0, 1024, 2048])), blk_cluster_map=int(rng.choice([-1, 0, 1])), blk_cluster_nl=int(rng.choice([0, 0, 1, 2, 4])),
                        blk_cluster768=int(rng.choice([-1, -1, 0, 8, 11])))   # cluster form (round 5): by shape, off, from 1025 / 2049 samples up; both workgroup maps
        elif path == 1:
            opts = dict(lanes_per_neuron=int(rng.choice([0, 1, 16, 32, 64])), waves_per_neuron=int(rng.choice([0, 0, 2, 4, 8, 16])),
                        onchip_mode=int(rng.integers(0, 2)))
        try:
            for k, v in opts.items():
                hip.set_option(k, v)
            r = hip.quantize_neurons(torch.from_numpy(X).to(dev), torch.from_numpy(Xq).to(dev), torch.from_numpy(W.T.copy()).to(dev),
                                     alphabet, path=path)
        finally:
            for k in opts:
                hip.set_option(k, {"onchip_mode": 1, "pipe": -1, "blk_sweep_waves": 0, "blk_quad_waves": 0, "blk_wide_groups": 1, "blk_four_groups": 1, "blk_pair_groups": 1, "blk_single_groups": 1, "blk_quad_groups": 2, "blk_cluster": 1, "blk_cluster_map": -1, "blk_cluster_nl": 0, "blk_cluster768": -1}.get(k, 0))
        ok = np.array_equal(r["idx"].cpu().numpy(), io) and np.allclose(r["resid"].cpu().numpy(), ro, rtol=1e-5, atol=0)
        n_dense += 1
        if not ok:
            bad.append(("dense", N, m, C, M, scalar, kind, path, opts, hip.last_dense_kernel()))
        # round 6: the same layer through the layer driver with the alphabet formed and kept on the device (median -> rad * alphabet on the
        # device, the Keras kernel read in place, the record pre-pass in runs of records / on a second stream), every shape -- the ones
        # without a block-pipelined kernel fall back to the host alphabet inside
        if M <= 64 and rng.random() < 0.6:
            popts = dict(blk_prep_run=int(rng.choice([1, 1, 0, 4, 8])), blk_cluster=int(rng.choice([1, 1, 0, 1024])))
            try:
                for k, v in popts.items():
                    hip.set_option(k, v)
                out = layer.quantize_dense_layer(torch.from_numpy(W).to(dev), torch.from_numpy(X).to(dev), torch.from_numpy(Xq).to(dev),
                                                 np.linspace(-1, 1, M), scalar, overlap=bool(rng.random() < 0.5), kernel_ready=[None, True][int(rng.random() < 0.5)])
            finally:
                hip.set_option("blk_prep_run", 1); hip.set_option("blk_cluster", 1)
            ok = (np.array_equal(out["idx"].cpu().numpy(), io.T) and np.array_equal(out["Q"].cpu().numpy(), Qo.T.astype(np.float32))
                  and np.allclose(out["resid"].cpu().numpy(), ro, rtol=1e-5, atol=0) and hip.call_status(out) == 0)
            n_dense += 1
            if not ok:
                bad.append(("dense layer driver, device alphabet", N, m, C, M, scalar, kind, popts, hip.last_dense_kernel()))
    else:
        # ---- conv ---------------------------------------------------------------------------
        kh = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 6, 7, 8])); kw = kh if rng.random() < 0.7 else int(rng.choice([1, 2, 3, 5, 7, 11, 17]))
        if kh * kw > 64:
            kw = 64 // kh
        stride = int(rng.choice([1, 1, 1, 2])); rate = int(rng.choice([1, 1, 1, 2])) if stride == 1 else 1
        padding = str(rng.choice(["SAME", "VALID"]))
        lo_h, lo_w = max(kh + (kh - 1) * (rate - 1), 4), max(kw + (kw - 1) * (rate - 1), 4)
        H = int(rng.integers(lo_h, max(30, lo_h + 4))); Wd = int(rng.integers(lo_w, max(30, lo_w + 4)))
        cin = int(rng.integers(1, 6)); F = int(rng.integers(1, 7))
        s2_case = rng.random() < 0.08                # 7x7 / 2 / VALID on images of 32+: the shift sums of the parity classes (gpfq_gram_s2.hip)
        if s2_case:
            kh = kw = 7; stride = 2; rate = 1; padding = "VALID"
            H = int(rng.integers(32, 72)); Wd = int(rng.integers(32, 72)); cin = int(rng.integers(1, 5)); F = int(rng.integers(1, 4))
        hip.set_option("conv_planes_free", int(rng.choice([1, 1, 0])))   # 0: that form fed from channel planes
        hip.set_option("conv_nhwc_halves", int(rng.choice([1, 1, 0])))   # 0: shards of <= 32 channels one image per wavefront
        nhwc_case = (kh, kw, stride, rate, padding) == (3, 3, 1, 1, "SAME") and rng.random() < 0.5
        if nhwc_case:                               # 32+ channels: the shift form straight from the NHWC activations (LDS-DMA ring)
            cin = int(rng.choice([rng.integers(32, 150), rng.integers(8, 32)])); F = int(rng.integers(1, 4))
        # long rows (round 4): 2^18 < columns <= 2^22 -- the range where the Gram path's certification bound grows with the row length
        # (gpfq_gram.hip, launch_gram_decide); the oracle walks GPU-built patch matrices of a few (channel, filter) pairs
        long_case = rng.random() < LONG_P
        if long_case:
            kh, kw, stride, rate, padding = [(3, 3, 1, 1, "SAME"), (3, 3, 1, 1, "SAME"), (7, 7, 2, 1, "VALID"), (5, 5, 1, 1, "SAME"), (3, 3, 2, 1, "SAME"),
                                             (3, 3, 1, 1, "VALID")][int(rng.integers(0, 6))]
            H = int(rng.integers(40, 120)); Wd = int(rng.integers(40, 120))
            cin = int(rng.choice([1, 2, 3, 3, 40])); F = int(rng.integers(1, 4))
            s2_case = nhwc_case = False
        oh, ow = hip.patch_out_dim(H, kh, stride, rate, padding == "SAME"), hip.patch_out_dim(Wd, kw, stride, rate, padding == "SAME")
        if oh * ow == 0:
            continue
        n = int(rng.choice([rng.integers(1, 20), -(-hip.GPFQ_GRAM_MIN_M // (oh * ow)) + int(rng.integers(1, 40))]))
        if long_case:
            cols = int(2 ** rng.uniform(18.0, 22.0 if cin <= 3 else 20.5)) + 1
            n = -(-cols // (oh * ow))
        if nhwc_case and not long_case:
            n = min(n, 12)                          # (the oracle walks every channel on the host)
            if cin < 32:
                n = 16                              # narrow shards take the NHWC form by image groups (2 .. 16 divide n)
        if s2_case and not long_case:
            n = int(rng.integers(1, 24))
        kind = rng.choice(["relu", "sparse", "signed", "uniform"])
        act_w = activations((n, H, Wd, cin), kind)
        first = rng.random() < 0.15
        act_q = act_w if first else (act_w + 0.05 * rng.standard_normal(act_w.shape).astype(np.float32) * (act_w != 0 if kind == "sparse" else 1)).astype(np.float32)
        if not first and kind != "signed":
            act_q = np.maximum(act_q, 0).astype(np.float32)
        Wk = (rng.standard_normal((kh, kw, cin, F)) / np.sqrt(kh * kw)).astype(np.float32)
        Wt = torch.from_numpy(Wk).to(dev)
        alphabet, _ = layer.layer_alphabet(Wt, np.linspace(-1, 1, M), scalar)
        aw = torch.from_numpy(act_w).to(dev); aq = aw if first else torch.from_numpy(act_q).to(dev)
        want_resid = bool(rng.random() < 0.3)
        hip.set_option("conv_shift", int(rng.choice([1, 2])))          # 2: the shift form of 3x3 / 1 / SAME layers at every image size
        out = layer.quantize_conv2d(Wt, aw, aq, alphabet, strides=(stride, stride), padding=padding, rate=(rate, rate), want_resid=want_resid)
        Q = out["Q"].cpu().numpy()
        ok = True
        for c in (range(cin) if cin <= 8 else sorted(set(int(v) for v in rng.integers(0, cin, 2 if long_case else 6)) | {0, cin - 1})):
            if long_case:                            # patch matrices built on the GPU (gpfq_extract_patches: pinned by tests/test_extract_patches_doc.py)
                Pw = hip.extract_patches(aw, c, (kh, kw), (stride, stride), (rate, rate), padding).cpu().numpy()
                Pq = Pw if first else hip.extract_patches(aq, c, (kh, kw), (stride, stride), (rate, rate), padding).cpu().numpy()
            else:
                Pw = ref_patches(act_w, c, kh, kw, stride, stride, rate, rate, padding)
                Pq = ref_patches(act_q, c, kh, kw, stride, stride, rate, rate, padding)
            for f in range(F):
                qo, _, uo = oracle.neuron(Wk[:, :, c, f].reshape(-1), Pw, Pq, alphabet)
                ok &= np.array_equal(Q[:, :, c, f].reshape(-1), qo.astype(np.float32))
                if want_resid:
                    ok &= bool(np.isclose(out["resid"][c, f].item(), np.linalg.norm(uo), rtol=1e-5, atol=0))
        n_conv += 1
        n_long += int(long_case)
        if not ok:
            bad.append(("conv", n, H, Wd, cin, F, kh, kw, stride, rate, padding, kind, first, M, scalar, want_resid))
    if bad:
        break
print(f"dense cases {n_dense}, conv cases {n_conv} (of which {n_long} with 2^18 .. 2^22 columns), mismatches {len(bad)}")
for b in bad:
    print("MISMATCH", b)
sys.exit(1 if bad else 0)
