"""Where a conv layer's wall time goes that is not kernel time: host-side seconds per call of the layer driver's pieces
(time.perf_counter without synchronisation = launch / Python cost; with synchronisation = including the GPU work).
usage: host_overhead_probe.py"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer

def timeit(fn, reps=50, sync=False):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
        if sync: torch.cuda.synchronize()
    if not sync: torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6

g = torch.Generator(device="cuda").manual_seed(2)
for (n, hw, cin, cout, k) in [(4096, 14, 256, 1024, 1), (4096, 14, 256, 256, 3)]:
    act_w = torch.rand((n, hw, hw, cin), device="cuda", generator=g)
    act_q = torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device="cuda", generator=g))
    W = torch.randn((k, k, cin, cout), device="cuda", generator=g) / k
    unit = np.linspace(-1, 1, 3)
    alphabet, rad = layer.layer_alphabet(W, unit, 3)
    kw = dict(strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    print(f"== {k}x{k} {cin}->{cout} @{hw}x{hw}")
    print(f"  layer_alphabet (median, host waits for it): {timeit(lambda: layer.layer_alphabet(W, unit, 3), sync=True):7.1f} us per call")
    print(f"  quantize_conv2d, calls back to back (launch-bound or GPU-bound, whichever is longer): {timeit(lambda: layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)):7.1f} us")
    print(f"  quantize_conv2d, synchronised after every call: {timeit(lambda: layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw), sync=True):7.1f} us")
    def both():
        a, _ = layer.layer_alphabet(W, unit, 3)
        layer.quantize_conv2d(W, act_w, act_q, a, **kw)
    print(f"  both, synchronised (what tools/bench_configs.py times): {timeit(both, sync=True):7.1f} us")
    if k == 1:
        W2 = W.reshape(cin, cout)
        print(f"  hip.quantize_conv1x1 alone, back to back: {timeit(lambda: hip.quantize_conv1x1(act_q, W2, alphabet, (1, 1))):7.1f} us")
        print(f"  torch.full((Cin, F), nan, f64): {timeit(lambda: torch.full((cin, cout), float('nan'), dtype=torch.float64, device='cuda')):7.1f} us")
