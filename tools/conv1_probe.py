"""ResNet50 conv1 (7x7 / stride 2, VALID on the padded 230x230 input, 3 -> 64) through the layer driver.
usage: conv1_probe.py [n_images] [--first] [--host-inputs]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
g = torch.Generator(device="cuda").manual_seed(2)
if "--host-inputs" in sys.argv:
    # inputs made on the HOST and uploaded, so that a kernel trace of this probe holds the layer driver's kernels only
    r = np.random.default_rng(2)
    a = r.random((n, 230, 230, 3), dtype=np.float32)
    act_w = torch.from_numpy(a).cuda()
    if "--first" in sys.argv:
        act_q = act_w
    else:
        a += np.float32(0.05) * r.standard_normal((n, 230, 230, 3), dtype=np.float32)
        act_q = torch.from_numpy(np.maximum(a, 0, out=a)).cuda()
    del a
    W = torch.from_numpy(r.standard_normal((7, 7, 3, 64), dtype=np.float32) / 7).cuda()
else:
    act_w = torch.rand((n, 230, 230, 3), device="cuda", generator=g)
    # --first: both networks see the same input, as for the first layer of a network (G2 = G1: half the MFMA work)
    act_q = act_w if "--first" in sys.argv else torch.relu(act_w + 0.05 * torch.randn((n, 230, 230, 3), device="cuda", generator=g))
    W = torch.randn((7, 7, 3, 64), device="cuda", generator=g) / 7
alphabet, rad = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(2, 2), padding="VALID", rate=(1, 1), want_resid=False)
    torch.cuda.synchronize(); dt = time.time() - t0
    print(f"conv1 n={n}: {dt*1e3:.1f} ms, host reruns {int(out['reruns'])}")
