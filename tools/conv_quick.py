"""Timing of the conv layer driver at BASELINE cfg4 shapes (CIFAR10 CNN, 5008 images, 3-bit), fused
3x3 path against the per-channel patch path.
usage: conv_quick.py [layers...] [--strip S] [--nofallback]   layers in {0, 2, 6, 8, 12, 14}"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer

SHAPES = {0: (3, 32, 32), 2: (32, 32, 32), 6: (32, 64, 16), 8: (64, 64, 16), 12: (64, 128, 8), 14: (128, 128, 8)}
args = [a for i, a in enumerate(sys.argv[1:], 1) if a.isdigit() and sys.argv[i - 1] not in ("--strip", "--shape")]
layers = [int(a) for a in args] or [2]
if "--strip" in sys.argv:
    hip.set_option("conv_strip", int(sys.argv[sys.argv.index("--strip") + 1]))
n = 5008
if "--shape" in sys.argv:                                   # --shape cin,cout,hw,n  (e.g. 512,512,7,4096)
    cin, cout, hw, n = (int(v) for v in sys.argv[sys.argv.index("--shape") + 1].split(","))
    SHAPES[99] = (cin, cout, hw)
    layers = [99]
for L in layers:
    cin, cout, hw = SHAPES[L]
    g = torch.Generator(device="cuda").manual_seed(0)
    act_w = torch.rand((n, hw, hw, cin), device="cuda", generator=g)
    act_q = torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device="cuda", generator=g))
    W = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / 3
    alphabet, rad = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
    res = {}
    for fused in ((1,) if "--nofallback" in sys.argv else (1, 0)):
        hip.set_option("conv_fused", fused)
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
            torch.cuda.synchronize(); dt = time.time() - t0
        res[fused] = (dt, out["idx"].clone())
    hip.set_option("conv_fused", 1)
    same = torch.equal(res[1][1], res[0][1]) if 0 in res else None
    print(f"layer {L}: Cin={cin} Cout={cout} m={n*hw*hw}: reruns {int(out['reruns'])} fused {res[1][0]*1e3:.2f} ms"
          + (f"  patches {res[0][0]*1e3:.2f} ms  identical={same}" if 0 in res else ""), flush=True)
