"""One conv layer through the layer driver (any kernel size / stride): time and the kernels that ran.
usage: conv_probe.py n H W cin cout k stride [SAME|VALID] [levels]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
a = sys.argv[1:]
n, H, W, cin, cout, k, stride = (int(v) for v in a[:7])
padding = a[7] if len(a) > 7 else "SAME"
levels = int(a[8]) if len(a) > 8 else 3
g = torch.Generator(device="cuda").manual_seed(2)
act_w = torch.relu(torch.randn((n, H, W, cin), device="cuda", generator=g))
act_q = torch.relu(act_w + 0.05 * torch.randn((n, H, W, cin), device="cuda", generator=g))
Wk = torch.randn((k, k, cin, cout), device="cuda", generator=g) / k
alphabet, rad = layer.layer_alphabet(Wk, np.linspace(-1, 1, levels), 3)
best = 1e9
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = layer.quantize_conv2d(Wk, act_w, act_q, alphabet, strides=(stride, stride), padding=padding, rate=(1, 1), want_resid=False)
    torch.cuda.synchronize(); best = min(best, time.time() - t0)
gb = 2 * act_w.numel() * 4 / 1e9
print(f"{k}x{k}/{stride} {padding} {cin}->{cout} @{H}x{W} n={n}: {best*1e3:.2f} ms, host reruns {int(out.get('reruns', 0))}; activations {gb:.2f} GB "
      f"= {gb / best / 1e3:.2f} TB/s if read once")
