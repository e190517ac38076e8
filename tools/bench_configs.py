"""Single-GPU timings of the BASELINE.json configs other than the headline (parity-test cases, not
bench lines): cfg1 MNIST MLP layer, cfg3 VGG16 fc2/fc1, cfg4 CIFAR10 CNN conv + dense layers, with
synthetic activations of the right shapes.  Writes gpurun_out/configs.json
(copied to profiles/r0N/configs.json).

    python tools/bench_configs.py [--skip-fc1] [--resnet] [--alphabet-in-layer] [--shapes] [--check profiles/r05/configs.json]

--shapes: also the kernel-only times of the dense shapes behind gpfq_blk.hip's dispatch table (blk_shape): the latency-bound widths,
the multi-GPU shards of the north-star layer, the long-row shapes.
--check FILE (round 5, the perf guard): after measuring, every record is compared with the record of the same name in FILE (a committed
profiles/r0N/configs.json); a layer more than 8 % (and more than 20 us) slower than its committed time fails the run (exit 1) -- a
regression in one of the ~25 hand-tuned shapes no longer passes silently.  Box-to-box noise is +-3 %.
"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
import oracle


def synthetic(N, m, C, dev):
    W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
    g = torch.Generator(device=dev).manual_seed(1)
    G = torch.randn((N, m), device=dev, generator=g)
    X = torch.relu(G)
    Xq = torch.relu(G + 0.1 * torch.randn((N, m), device=dev, generator=g))
    return W, X, Xq


def time_dense(name, N, m, C, bits, scalar, dev, check=8):
    W, X, Xq = synthetic(N, m, C, dev)
    Wd = torch.from_numpy(W).to(dev)
    unit = np.linspace(-1, 1, int(round(2 ** bits)))
    best = 1e9
    alphabet, rad = layer.layer_alphabet(Wd, unit, scalar)       # (before the timed region, as the class surface forms it: see time_conv)
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if "--alphabet-in-layer" in sys.argv:
            alphabet, rad = layer.layer_alphabet(Wd, unit, scalar)
        out = layer.quantize_dense(Wd, X, Xq, alphabet)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    # parity of a few neurons against the oracle on the same tensors
    Xh, Xqh = X.cpu().numpy(), Xq.cpu().numpy()
    _, io, ro = oracle.layer(W, Xh, Xqh, alphabet, 0, check)
    bad = int((out["idx"][:, :check].t().cpu().numpy() != io).any(axis=1).sum())
    rel = float(np.max(np.abs(out["resid"][:check].cpu().numpy() - ro) / ro))
    rec = dict(config=name, kind="dense", N=N, m=m, C=C, M=len(unit), ms=best * 1e3, weights_per_s=N * C / best,
               algorithmic_GBps=N * C * (8 * m + 8) / best / 1e9, neurons_checked=check,
               neurons_with_index_mismatch=bad, max_resid_rel_err=rel)
    print(json.dumps(rec), flush=True)
    return rec


def time_conv(name, cin, cout, hw, n, bits, scalar, dev, check=2, k=3, stride=1, padding="SAME", reps=4, first=False):
    g = torch.Generator(device=dev).manual_seed(2)
    act_w = torch.rand((n, hw, hw, cin), device=dev, generator=g)
    # first: the layer is the network's first, both networks see the data itself (scripts/quantized_network.py:478-481)
    act_q = act_w if first else torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device=dev, generator=g))
    W = torch.randn((k, k, cin, cout), device=dev, generator=g) / k
    unit = np.linspace(-1, 1, int(round(2 ** bits)))
    best = 1e9
    # (round 4: the class surface queues the medians of ALL layers' analog kernels before the first layer and reads them back with one
    #  host wait -- QuantizedNeuralNetwork._prefetch_medians -- so a layer's time no longer contains the alphabet's 80 us of host wait:
    #  the alphabet is formed before the timed region here too; --alphabet-in-layer times it inside, as rounds 1-3 did)
    alphabet, rad = layer.layer_alphabet(W, unit, scalar)
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if "--alphabet-in-layer" in sys.argv:
            alphabet, rad = layer.layer_alphabet(W, unit, scalar)
        out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(stride, stride), padding=padding, rate=(1, 1),
                                    want_resid=False)      # as the class surface calls it (residual norms are diagnostics)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    # parity: channel 0, first `check` filters, oracle on the GPU-built patch matrices
    Pw = hip.extract_patches(act_w, 0, (k, k), (stride, stride), (1, 1), padding)
    Pq = hip.extract_patches(act_q, 0, (k, k), (stride, stride), (1, 1), padding)
    m = Pw.shape[1]
    Wh = W.cpu().numpy()
    bad = 0
    if m * k * k <= 2_000_000_000:               # keep the host copy of the patch matrices reasonable
        Pw, Pq = Pw.cpu().numpy(), Pq.cpu().numpy()
        for f in range(check):
            q, _, _ = oracle.neuron(Wh[:, :, 0, f].reshape(-1), Pw, Pq, alphabet)
            bad += int(not np.array_equal(out["Q"][:, :, 0, f].cpu().numpy().reshape(-1), q.astype(np.float32)))
    else:
        check = 0
    rec = dict(config=name, kind=f"conv{k}x{k}/{stride}", Cin=cin, Cout=cout, m=int(m), M=len(unit), ms=best * 1e3,
               weights_per_s=k * k * cin * cout / best, filters_checked=check, filters_with_mismatch=bad)
    print(json.dumps(rec), flush=True)
    del act_w, act_q, out
    torch.cuda.empty_cache()
    return rec


def time_shape(N, C, m, bits, scalar, dev, reps=5):
    """Kernel-only time (pre-pass included) of one dense shape through the raw binding, best of `reps`, oracle on 4 neurons."""
    W, X, Xq = synthetic(N, m, C, dev)
    Wd = torch.from_numpy(W).to(dev)
    unit = np.linspace(-1, 1, int(round(2 ** bits)))
    alphabet, _ = layer.layer_alphabet(Wd, unit, scalar)
    Wt = Wd.t().contiguous()
    nrm = hip.row_norms(Xq)
    out = hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm, want_values=False)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm, want_values=False); b.record()
        torch.cuda.synchronize(); best = min(best, a.elapsed_time(b))
    _, io, _ = oracle.layer(W, X.cpu().numpy(), Xq.cpu().numpy(), alphabet, 0, 4)
    bad = int((out["idx"][:4].cpu().numpy() != io).any(axis=1).sum())
    rec = dict(config=f"shape {N} x {C} on {m} samples, M={len(unit)}", kind="dense kernel", N=N, m=m, C=C, M=len(unit), ms=best,
               kernel=hip.last_dense_kernel()[:40], neurons_checked=4, neurons_with_index_mismatch=bad)
    print(json.dumps(rec), flush=True)
    return rec


def check_against(recs, path, rel=0.08, abs_ms=0.020):
    """The perf guard: records by name against a committed configs.json."""
    ref = {r["config"]: r for r in json.load(open(path))["records"] if "ms" in r}
    worse, seen = [], 0
    for r in recs:
        old = ref.get(r.get("config"))
        if old is None or "ms" not in r:
            continue
        seen += 1
        if r["ms"] > old["ms"] * (1 + rel) and r["ms"] > old["ms"] + abs_ms:
            worse.append(f"{r['config']}: {r['ms']:.3f} ms against {old['ms']:.3f} committed (+{100 * (r['ms'] / old['ms'] - 1):.0f} %)")
    print(f"perf guard: {seen} records compared with {path}, {len(worse)} more than {100 * rel:.0f} % slower")
    for w in worse:
        print("  SLOWER: " + w)
    return not worse


def main():
    dev = torch.device("cuda", 0)
    recs = []
    recs.append(time_dense("cfg1 MNIST MLP Dense(784->128), m=512, 4-bit, scalar 5", 784, 512, 128, 4, 5, dev))
    recs.append(time_dense("cfg2 Dense(4096->4096), m=1024, ternary, scalar 3 (layer driver incl. median/assemble)", 4096, 1024, 4096, np.log2(3), 3, dev))
    recs.append(time_dense("cfg3 VGG16 fc2 Dense(4096->4096), m=2048, 4-bit, scalar 5", 4096, 2048, 4096, 4, 5, dev))
    if "--skip-fc1" not in sys.argv:
        recs.append(time_dense("cfg3 VGG16 fc1 Dense(25088->4096), m=2048, 4-bit, scalar 5", 25088, 2048, 4096, 4, 5, dev, check=4))
    recs.append(time_dense("cfg3 VGG16 predictions Dense(4096->1000), m=2048, 4-bit, scalar 5", 4096, 2048, 1000, 4, 5, dev))
    total = 0.0
    for idx, (cin, cout, hw) in {0: (3, 32, 32), 2: (32, 32, 32), 6: (32, 64, 16), 8: (64, 64, 16), 12: (64, 128, 8), 14: (128, 128, 8)}.items():
        r = time_conv(f"cfg4 CIFAR10 CNN conv layer {idx} ({cin}->{cout} @{hw}x{hw}), 5008 images, 3-bit, scalar 4", cin, cout, hw, 5008, 3, 4, dev)
        total += r["ms"]; recs.append(r)
    for name, N, C in [("cfg4 CIFAR10 CNN Dense(2048->128), m=5008", 2048, 128), ("cfg4 CIFAR10 CNN Dense(128->10), m=5008", 128, 10)]:
        r = time_dense(name + ", 3-bit, scalar 4", N, 5008, C, 3, 4, dev, check=4)
        total += r["ms"]; recs.append(r)
    recs.append(dict(config="cfg4 CIFAR10 CNN, all 6 conv + 2 dense layers (quantization only, synthetic activations)", ms=total))
    print(json.dumps(recs[-1]))
    if "--resnet" in sys.argv:
        # cfg5: representative Keras-ResNet50 conv layers, 4096 synthetic calibration images, ternary;
        # one layer of each spatial size (the net has 3/4/6/3 of the 3x3 ones) + conv1 (7x7/2 on the padded 230x230 input)
        n = 4096
        recs.append(time_conv("cfg5 ResNet50 conv1 7x7/2 VALID (3->64 @230x230 padded input), 4096 images, ternary, scalar 3",
                              3, 64, 230, n, np.log2(3), 3, dev, k=7, stride=2, padding="VALID", reps=4))
        recs.append(time_conv("cfg5 ResNet50 conv1 7x7/2 VALID as the FIRST layer it is (both networks see the images), 4096 images, ternary, scalar 3 [not in the total]",
                              3, 64, 230, n, np.log2(3), 3, dev, k=7, stride=2, padding="VALID", reps=4, first=True))
        conv1_ms = recs[-2]["ms"]
        # every other conv layer of the net by distinct shape (cin, cout, input size, kernel, stride) x how often it occurs:
        # 16 3x3 layers, 36 1x1 layers (four of them the stride-2 first convolutions of a stage, four the stride-2 shortcuts)
        inventory = [(64, 64, 56, 1, 1, 1), (64, 64, 56, 3, 1, 3), (64, 256, 56, 1, 1, 4), (256, 64, 56, 1, 1, 2),
                     (256, 128, 56, 1, 2, 1), (256, 512, 56, 1, 2, 1), (128, 128, 28, 3, 1, 4), (128, 512, 28, 1, 1, 4), (512, 128, 28, 1, 1, 3),
                     (512, 256, 28, 1, 2, 1), (512, 1024, 28, 1, 2, 1), (256, 256, 14, 3, 1, 6), (256, 1024, 14, 1, 1, 6), (1024, 256, 14, 1, 1, 5),
                     (1024, 512, 14, 1, 2, 1), (1024, 2048, 14, 1, 2, 1), (512, 512, 7, 3, 1, 3), (512, 2048, 7, 1, 1, 3), (2048, 512, 7, 1, 1, 2)]
        total5, count5 = conv1_ms, 1
        for cin, cout, hw, k, stride, times in inventory:
            r = time_conv(f"cfg5 ResNet50 {k}x{k}/{stride} conv ({cin}->{cout} @{hw}x{hw}) x{times}, 4096 images, ternary, scalar 3",
                          cin, cout, hw, n, np.log2(3), 3, dev, k=k, stride=stride, reps=5)
            r["occurrences"] = times
            total5 += times * r["ms"]; count5 += times
            recs.append(r)
        recs.append(dict(config=f"cfg5 ResNet50, all {count5} conv layers (quantization only, synthetic activations of each layer's shape, one GPU)", ms=total5))
        print(json.dumps(recs[-1]))
    if "--shapes" in sys.argv:
        L3, L4 = float(np.log2(3)), 4.0
        for N, C, m, bits, scalar in [(4096, 512, 1024, L3, 3), (4096, 1024, 1024, L3, 3), (4096, 2048, 1024, L3, 3), (4096, 4096, 1024, L3, 3),
                                      (4096, 128, 1024, L3, 3), (4096, 10, 1024, L3, 3), (4096, 256, 1024, L3, 3),
                                      (4096, 4096, 768, L3, 3), (4096, 4096, 512, L3, 3), (4096, 1024, 768, L3, 3), (4096, 1024, 512, L3, 3),
                                      (4096, 4096, 1000, L4, 5), (4096, 4096, 1536, L4, 5), (4096, 4096, 2048, L4, 5), (4096, 2048, 1536, L4, 5),
                                      (4096, 1000, 2048, L4, 5), (4096, 512, 2048, L4, 5), (4096, 128, 2048, L4, 5),
                                      (4096, 4096, 3000, L4, 5), (4096, 4096, 4096, L4, 5), (4096, 1024, 3000, L4, 5), (4096, 4096, 5008, 3.0, 4),
                                      (2048, 128, 5008, 3.0, 4), (784, 128, 512, L4, 5), (4096, 4096, 8192, L3, 3),
                                      # round 5, the cluster form's own shapes: 8-neuron workgroups, slice counts that do not divide 8, the longest rows
                                      (4096, 512, 4096, L4, 5), (4096, 4096, 6000, L4, 5), (4096, 4096, 12000, L3, 3), (4096, 1024, 28672, L3, 3)]:
            recs.append(time_shape(N, C, m, bits, scalar, dev))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(dict(note="tools/bench_configs.py on one MI355X; best of 4-5 runs per layer; whole layer driver "
                        "(norms, kernel, assemble" + (", alphabet median" if "--alphabet-in-layer" in sys.argv else
                                                      "; the alphabet's median is formed up front, as the class surface does since round 4")
                        + ") with inputs resident in HBM", records=recs),
              open("gpurun_out/configs.json", "w"), indent=1)
    if "--check" in sys.argv:
        if not check_against(recs, sys.argv[sys.argv.index("--check") + 1]):
            sys.exit(1)


if __name__ == "__main__":
    main()
