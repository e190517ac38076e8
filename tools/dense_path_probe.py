"""Dense layer, on-chip kernels (path 1) against the Gram path (path 3) over a range of row lengths.
usage: dense_path_probe.py N C m1 [m2 ...]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
N, C = int(sys.argv[1]), int(sys.argv[2])
for m in (int(v) for v in sys.argv[3:]):
    g = torch.Generator(device="cuda").manual_seed(1)
    G = torch.randn((N, m), device="cuda", generator=g)
    X = torch.relu(G)
    Xq = torch.relu(G + 0.1 * torch.randn((N, m), device="cuda", generator=g))
    W = torch.randn((N, C), device="cuda", generator=g) / np.sqrt(N)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    Wt = W.t().contiguous()
    res = {}
    for path in (1, 3):
        if path == 1 and m > hip.GPFQ_ONCHIP_MAX_M:
            continue
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            r = hip.quantize_neurons(X, Xq, Wt, alphabet, path=path, want_values=False, want_resid=None)
            torch.cuda.synchronize(); dt = time.time() - t0
        res[path] = (dt, r["idx"])
    same = torch.equal(res[1][1], res[3][1]) if 1 in res else None
    print(f"N={N} C={C} m={m}: on-chip {res[1][0]*1e3 if 1 in res else float('nan'):.2f} ms, Gram {res[3][0]*1e3:.2f} ms, equal {same}")
