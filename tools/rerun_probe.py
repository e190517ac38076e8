"""How many chains of a conv layer the Gram path cannot certify, per stage (diagnostics).
usage: rerun_probe.py cin cout hw n bits scalar"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
cin, cout, hw, n = (int(v) for v in sys.argv[1:5])
bits, scalar = float(sys.argv[5]), float(sys.argv[6])
g = torch.Generator(device="cuda").manual_seed(2)
act_w = torch.rand((n, hw, hw, cin), device="cuda", generator=g)
act_q = torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device="cuda", generator=g))
W = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / 3
unit = np.linspace(-1, 1, int(round(2 ** bits)))
alphabet, rad = layer.layer_alphabet(W, unit, scalar)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"Cin={cin} Cout={cout} hw={hw} n={n} M={len(unit)}: {dt*1e3:.2f} ms, host reruns {int(out['reruns'])} of {cin*cout} chains")
