"""Timing of the conv layer driver at BASELINE cfg4 shapes (CIFAR10 CNN, 5008 columns, 3-bit).
usage: conv_quick.py [layer]   layer in {0, 2, 6, 8, 12, 14}, default 2"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer

SHAPES = {0: (3, 32, 32), 2: (32, 32, 32), 6: (32, 64, 16), 8: (64, 64, 16), 12: (64, 128, 8), 14: (128, 128, 8)}
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cin, cout, hw = SHAPES[L]
n = 5008
g = torch.Generator(device="cuda").manual_seed(0)
act_w = torch.rand((n, hw, hw, cin), device="cuda", generator=g)
act_q = torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device="cuda", generator=g))
W = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / 3
alphabet, rad = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1))
    torch.cuda.synchronize(); dt = time.time() - t0
    m = n * hw * hw
    nw = 9 * cin * cout
    print(f"layer {L}: Cin={cin} Cout={cout} m={m}: {dt*1e3:.1f} ms  {nw/dt:.3e} weights/s  "
          f"u-traffic {cin*cout*10*m*16/dt/1e9:.0f} GB/s", flush=True)
# per-phase timing for one channel
torch.cuda.synchronize(); t0 = time.time()
Pw = hip.extract_patches(act_w, 0, (3, 3), (1, 1), (1, 1), "SAME")
Pq = hip.extract_patches(act_q, 0, (3, 3), (1, 1), (1, 1), "SAME")
torch.cuda.synchronize(); t1 = time.time()
Wt = W[:, :, 0, :].reshape(9, cout).t().contiguous()
r = hip.quantize_neurons(Pw, Pq, Wt, alphabet)
torch.cuda.synchronize(); t2 = time.time()
print(f"one channel: patches {1e3*(t1-t0):.2f} ms, quantize {1e3*(t2-t1):.2f} ms")
