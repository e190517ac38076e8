"""Any conv layer shape through the layer driver, matrix-core records against vector-unit tiles (variant bit 2).
usage: conv_shape_probe.py n H W cin cout kh kw stride [SAME|VALID]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
n, H, W, cin, cout, kh, kw, stride = (int(v) for v in sys.argv[1:9])
padding = sys.argv[9] if len(sys.argv) > 9 else "SAME"
g = torch.Generator(device="cuda").manual_seed(2)
act_w = torch.relu(torch.randn((n, H, W, cin), device="cuda", generator=g))
act_q = torch.relu(act_w + 0.05 * torch.randn((n, H, W, cin), device="cuda", generator=g))
Wk = torch.randn((kh, kw, cin, cout), device="cuda", generator=g) / np.sqrt(kh * kw)
alphabet, rad = layer.layer_alphabet(Wk, np.linspace(-1, 1, 3), 3)
ref = None
for variant in (0, 4):
    hip.set_option("variant", variant)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        out = layer.quantize_conv2d(Wk, act_w, act_q, alphabet, strides=(stride, stride), padding=padding, rate=(1, 1), want_resid=False)
        torch.cuda.synchronize(); dt = time.time() - t0
    same = "" if ref is None else f", equal to variant 0: {bool(torch.equal(ref, out['Q']))}"
    ref = out["Q"] if ref is None else ref
    print(f"{kh}x{kw}/{stride} {cin}->{cout} @{H}x{W} n={n} variant {variant}: {dt*1e3:.2f} ms, host reruns {int(out['reruns'])}{same}")
hip.set_option("variant", 0)
