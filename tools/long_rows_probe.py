import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
for (N, C, m) in [(4096, 4096, 3000), (4096, 4096, 4096), (2048, 1024, 4096), (4096, 4096, 8192)]:
    g = np.random.default_rng(0)
    W = (g.standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
    G = g.standard_normal((N, m)).astype(np.float32)
    X = np.maximum(G, 0); Xq = np.maximum(G + 0.1 * g.standard_normal((N, m)).astype(np.float32), 0)
    Wd, Xd, Xqd = (torch.from_numpy(a).cuda() for a in (W, X, Xq))
    alphabet, _ = layer.layer_alphabet(Wd, np.linspace(-1, 1, 3), 3)
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        out = layer.quantize_dense(Wd, Xd, Xqd, alphabet)
        torch.cuda.synchronize(); best = min(best, time.time() - t0)
    print(f"N={N} C={C} m={m}: {best*1e3:.2f} ms  kernel {hip.last_dense_kernel()[:60]}")
