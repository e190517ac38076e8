"""Sweep workgroup size / tile height of the on-chip kernel at cfg2 (tuning hooks)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip
N = C = 4096; m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024; M = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
G = np.random.default_rng(1).standard_normal((N, m))
X = np.maximum(G, 0).astype(np.float32)
Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
alphabet = 3 * float(np.median(np.abs(W))) * np.linspace(-1, 1, M)
Xd, Xqd, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(Xq).cuda(), torch.from_numpy(W.T.copy()).cuda()
nrm = hip.row_norms(Xqd)
ref = None
import itertools
for gw, var, ts in itertools.product((16, 32, 64), (0, 1), (8, 4)):
    if True:
        hip.set_option("lanes_per_neuron", gw); hip.set_option("tile_steps", ts); hip.set_option("variant", var)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm)
            torch.cuda.synchronize(); best = min(best, time.time() - t0)
        if ref is None: ref = r["idx"].clone()
        print(f"variant={var} lanes_per_neuron={gw:2d} tile_steps={ts:2d}: {best*1e3:.2f} ms  same={bool(torch.equal(ref, r['idx']))}", flush=True)
