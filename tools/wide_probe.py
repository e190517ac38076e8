import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
N, C, m = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2048, 128, 5008)
g = torch.Generator(device="cuda").manual_seed(1)
W = torch.randn((N, C), device="cuda", generator=g) / np.sqrt(N)
G = torch.randn((N, m), device="cuda", generator=g)
X = torch.relu(G); Xq = torch.relu(G + 0.1 * torch.randn((N, m), device="cuda", generator=g))
alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
Wt = W.t().contiguous()
nrm = hip.row_norms(Xq)
ref = None
for wpn in (0, 4, 6, 8, 10, 12, 16):
    hip.set_option("waves_per_neuron", wpn)
    best = 1e9
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.time()
        r = hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm, path=hip.GPFQ_PATH_ONCHIP)
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    same = "" if ref is None else f" equal: {bool(torch.equal(ref, r['idx']))}"
    ref = r["idx"] if ref is None else ref
    print(f"waves_per_neuron={wpn}: {best*1e3:.2f} ms [{hip.last_dense_kernel()[:40]}]{same}")
hip.set_option("waves_per_neuron", 0)
