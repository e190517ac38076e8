"""One 3x3 / stride 1 / SAME layer through the NHWC shift form, a few runs (for kernel traces and counter passes).
usage: nhwc_probe.py n H W cin cout [runs=3] [slots]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
n, H, W, cin, cout = [int(v) for v in sys.argv[1:6]]
runs = int(sys.argv[6]) if len(sys.argv) > 6 else 3
g = torch.Generator(device="cuda").manual_seed(2)
act_w = torch.relu(torch.randn((n, H, W, cin), device="cuda", generator=g))
act_q = torch.relu(act_w + 0.05 * torch.randn((n, H, W, cin), device="cuda", generator=g))
Wk = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / 3
alphabet, rad = layer.layer_alphabet(Wk, np.linspace(-1, 1, 3), 3)
if len(sys.argv) > 7:
    hip.set_option("conv_nhwc_slots", int(sys.argv[7]))
best = 1e9
for it in range(runs):
    torch.cuda.synchronize(); t0 = time.time()
    out = layer.quantize_conv2d(Wk, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    torch.cuda.synchronize(); best = min(best, time.time() - t0)
print(f"3x3 {cin}->{cout} @{H}x{W} n={n}{(' slots ' + sys.argv[7]) if len(sys.argv) > 7 else ''}: {best*1e3:.2f} ms, host reruns {int(out['reruns'])}")
