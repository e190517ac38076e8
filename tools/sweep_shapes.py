"""Kernel-variant sweep over layer shapes (tuning hooks); prints ms per variant."""
import sys, time, itertools
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip

def run(N, m, C, M):
    W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
    G = np.random.default_rng(1).standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
    alphabet = 3 * float(np.median(np.abs(W))) * np.linspace(-1, 1, M)
    Xd, Xqd, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(Xq).cuda(), torch.from_numpy(W.T.copy()).cuda()
    nrm = hip.row_norms(Xqd)
    ref = None
    out = []
    for lpn, var in [(0, 0), (1, 0), (32, 0), (64, 0)]:
        hip.set_option("lanes_per_neuron", lpn); hip.set_option("variant", var)
        best = 1e9
        try:
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.time()
                r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm)
                torch.cuda.synchronize(); best = min(best, time.time() - t0)
        except Exception as e:
            out.append(f"lpn{lpn}v{var}: ERR"); continue
        if ref is None: ref = r["idx"].clone()
        out.append(f"lpn{lpn}v{var}: {best*1e3:7.2f}{'' if torch.equal(ref, r['idx']) else ' MISMATCH'}")
    print(f"N={N} m={m} C={C} M={M} | " + " | ".join(out), flush=True)

shapes = [(1024, 1024, 256, 3), (1024, 1024, 100, 3), (4096, 1024, 10, 3), (1024, 2048, 1000, 16),
          (1024, 2048, 4096, 16), (1024, 512, 4096, 16), (1024, 256, 4096, 3), (1024, 128, 4096, 3), (1024, 1024, 1024, 3),
          (1024, 1024, 512, 3), (1024, 1024, 2048, 16), (784, 512, 128, 16), (1024, 1024, 8192, 3), (1024, 1536, 4096, 3)]
for s in shapes:
    run(*s)
