#!/usr/bin/env python3
"""Round 6: what the headline kernel costs by where its weights come from and where its outputs go (kernel-only times by the library's
events, medians of 12 launches, one box): neuron-major weights / the Keras kernel in place; neuron-major indices only (round 5's bench step),
indices + values, or both in the Keras layout by the kernel's own flush (the KOUT instantiation)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_networks_amd import hip, layer
N, C, m = 4096, 4096, 1024
dev = torch.device("cuda", 0)
g = np.random.default_rng(1).standard_normal((N, m))
X = torch.from_numpy(np.maximum(g, 0).astype(np.float32)).to(dev)
Xq = torch.from_numpy(np.maximum(g + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)).to(dev)
W = torch.from_numpy((np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)).to(dev)
unit = np.linspace(-1, 1, 3)
d = layer.layer_alphabet_device(W, unit, 3.0)
alphabet = d.values()
Wt = hip.neuron_major(W)
nrm = hip.row_norms(Xq)
ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
variants = {
    "neuron-major W, neuron-major idx only (round 5's step)": lambda: hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm, want_values=False),
    "neuron-major W, neuron-major idx + Q": lambda: hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm, want_values=True),
    "Keras W in place, neuron-major idx only": lambda: hip.quantize_dense_layer(X, Xq, W, d, nrm32=nrm, keras_out=False, want_values=False),
    "Keras W in place, Keras-layout idx + Q (KOUT)": lambda: hip.quantize_dense_layer(X, Xq, W, d, nrm32=nrm, keras_out=True),
    "Keras W in place, Keras-layout Q only (KOUT)": lambda: hip.quantize_dense_layer(X, Xq, W, d, nrm32=nrm, keras_out=True, want_idx=False),
}
for rep in range(2):
    for name, fn in variants.items():
        ks = []
        for _ in range(12):
            hip.set_main_kernel_events(*ev)
            fn()
            torch.cuda.synchronize()
            hip.set_main_kernel_events(None, None)
            ks.append(ev[0].elapsed_time(ev[1]))
        print(f"{name:58s} kernel {np.median(ks):.4f} ms (min {np.min(ks):.4f})")
