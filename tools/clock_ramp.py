"""After an idle period the chip runs the same kernel slower and comes back over several launches: the north-star layer's recurrence kernel,
30 launches back to back after 0 / 1 / 10 / 100 ms of idle, each launch's own duration (the library's events on the dispatch).
usage: clock_ramp.py   (on the GPU box)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_networks_amd import hip, layer
N, m, C = 4096, 1024, 4096
r = np.random.default_rng(0)
G = r.standard_normal((N, m))
X = torch.from_numpy(np.maximum(G, 0).astype(np.float32)).cuda()
Xq = torch.from_numpy(np.maximum(G + 0.1 * r.standard_normal((N, m)), 0).astype(np.float32)).cuda()
W = torch.from_numpy((r.standard_normal((N, C)) / 64).astype(np.float32)).cuda()
unit = np.linspace(-1, 1, 3)
d = layer.layer_alphabet_device(W, unit, 3.0)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
for a, b in ev:
    a.record(); b.record()
for idle_ms in (0, 1, 10, 100, 1000):
    for _ in range(12):
        hip.quantize_dense_layer(X, Xq, W, d)
    torch.cuda.synchronize()
    time.sleep(idle_ms / 1e3)
    for a, b in ev:
        hip.set_main_kernel_events(a, b)
        hip.quantize_dense_layer(X, Xq, W, d)
    hip.set_main_kernel_events(None, None)
    torch.cuda.synchronize()
    ks = [a.elapsed_time(b) for a, b in ev]
    print(f"idle {idle_ms:5d} ms, then 30 launches: " + " ".join(f"{k:.3f}" for k in ks[:12]) + f" ... last ten avg {np.mean(ks[-10:]):.3f} ms")
