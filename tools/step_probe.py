#!/usr/bin/env python3
"""Round 6: what a Dense layer's step costs outside the recurrence kernel -- the median of |W|, the device alphabet, the row norms and
the record pre-pass (option blk_prep_run = 0 / 4 / 1), each as the average of back-to-back launches between two events.
    python tools/step_probe.py [N C m]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_networks_amd import hip, layer  # noqa: E402

N, C, m = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (4096, 4096, 1024)))
dev = torch.device("cuda", 0)
g = np.random.default_rng(1).standard_normal((N, m))
X = torch.from_numpy(np.maximum(g, 0).astype(np.float32)).to(dev)
Xq = torch.from_numpy(np.maximum(g + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)).to(dev)
W = torch.from_numpy((np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)).to(dev)
unit = np.linspace(-1, 1, 3)


def timed(name, fn, reps=50):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    print(f"{name:60s} {a.elapsed_time(b) / reps * 1e3:9.1f} us")


timed("median_abs (device scalar)", lambda: hip.median_abs(W.reshape(-1), on_device=True))
med = hip.median_abs(W.reshape(-1), on_device=True)
assert float(med.item()) == float(np.median(np.abs(W.cpu().numpy())))
timed("layer_alphabet_device from the median", lambda: hip.layer_alphabet_device(med, unit, 3.0))
timed("median + alphabet", lambda: layer.layer_alphabet_device(W, unit, 3.0))
timed("row_norms", lambda: hip.row_norms(Xq))
d = layer.layer_alphabet_device(W, unit, 3.0)
nrm = hip.row_norms(Xq)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
for run in (0, 4, 5, 6, 8, 9, 12, 16, 1):
    hip.set_option("blk_prep_run", run)
    ks, cs = [], []
    for _ in range(8):
        hip.set_main_kernel_events(*ev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = hip.quantize_dense_layer(X, Xq, W, d, nrm32=nrm)
        b.record(); torch.cuda.synchronize()
        hip.set_main_kernel_events(None, None)
        ks.append(ev[0].elapsed_time(ev[1])); cs.append(a.elapsed_time(b))
    print(f"blk_prep_run={run}: call - kernel = {1e3 * (np.median(cs) - np.median(ks)):7.1f} us   (kernel {np.median(ks):.4f} ms, status {hip.call_status(r)})")
hip.set_option("blk_prep_run", 1)
