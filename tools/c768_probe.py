#!/usr/bin/env python3
"""Round 6: rows of 2049..3072 samples in wide layers -- the classic one-step shape (option blk_cluster768 = 0) against four 768-sample slices of
the cluster form with 8 / 11 sweep wavefronts; kernel-only times by the library's events, results compared with each other."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_networks_amd import hip
dev = torch.device("cuda", 0)
for (N, C, m, M) in ((4096, 4096, 3000, 16), (4096, 4096, 3000, 3), (4096, 2560, 3072, 8), (300, 2100, 2500, 4)):
    g = torch.Generator(device=dev).manual_seed(3)
    G = torch.randn((N, m), device=dev, generator=g)
    X, Xq = torch.relu(G), torch.relu(G + 0.1 * torch.randn((N, m), device=dev, generator=g))
    W = torch.randn((N, C), device=dev, generator=g) / np.sqrt(N)
    Wt = hip.neuron_major(W)
    alphabet = 3 * float(W.abs().median()) * np.linspace(-1, 1, M)
    nrm = hip.row_norms(Xq)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    ref = None
    for opt in (0, 8, 11):
        hip.set_option("blk_cluster768", opt)
        ks = []
        for _ in range(4):
            hip.set_main_kernel_events(*ev)
            r = hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm)
            torch.cuda.synchronize()
            hip.set_main_kernel_events(None, None)
            ks.append(ev[0].elapsed_time(ev[1]))
        same = "" if ref is None else (f"  indices equal to option 0: {bool(torch.equal(ref['idx'], r['idx']))}, values: {bool(torch.equal(ref['Q'], r['Q']))}, "
                                       f"residual norms max rel diff {float(((ref['resid'] - r['resid']).abs() / ref['resid']).max()):.1e}")
        ref = r if ref is None else ref
        print(f"{N} x {C} on {m} samples, M={M}, blk_cluster768={opt:2d}: kernel {np.median(ks):.3f} ms  status {hip.call_status(r)}  fallbacks {hip.exact_fallbacks(r)}{same}  [{hip.last_dense_kernel()[:40]}]")
hip.set_option("blk_cluster768", 11)
