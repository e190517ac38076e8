#!/usr/bin/env python3
"""Rows of 1025..2048 samples in wide layers (BASELINE cfg3's fc2: 4096 x 4096 on 2048 samples): the classic two-step shape against two
1024-sample slices of the cluster form (option blk_cluster = 1024); kernel-only times by the library's events after ten warm launches (the
chip's clock ramp, profiles/r06/clock_ramp.txt), results compared with each other."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_networks_amd import hip
dev = torch.device("cuda", 0)
shapes = ((4096, 4096, 2048, 16), (4096, 4096, 2048, 3), (4096, 4096, 1536, 16), (4096, 1000, 2048, 16), (4096, 4096, 2000, 16))
for (N, C, m, M) in shapes:
    g = torch.Generator(device=dev).manual_seed(3)
    G = torch.randn((N, m), device=dev, generator=g)
    X, Xq = torch.relu(G), torch.relu(G + 0.1 * torch.randn((N, m), device=dev, generator=g))
    W = torch.randn((N, C), device=dev, generator=g) / np.sqrt(N)
    Wt = hip.neuron_major(W)
    alphabet = 3 * float(W.abs().median()) * np.linspace(-1, 1, M)
    nrm = hip.row_norms(Xq)
    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    for e in ev:
        e.record()
    ref = None
    for opt in (1, 1024, 1, 1024):
        hip.set_option("blk_cluster", opt)
        for _ in range(6):
            r = hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm)
        ks = []
        for _ in range(5):
            hip.set_main_kernel_events(*ev)
            r = hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm)
            hip.set_main_kernel_events(None, None)
            torch.cuda.synchronize()
            ks.append(ev[0].elapsed_time(ev[1]))
        same = "" if ref is None else f"  indices equal: {bool(torch.equal(ref['idx'], r['idx']))}, values: {bool(torch.equal(ref['Q'], r['Q']))}"
        ref = r if ref is None else ref
        print(f"{N} x {C} on {m} samples, M={M}, blk_cluster={opt:4d}: kernel {np.median(ks):.3f} ms (min {np.min(ks):.3f})  status {hip.call_status(r)}{same}  [{hip.last_dense_kernel()[:28]}]")
hip.set_option("blk_cluster", 1)
