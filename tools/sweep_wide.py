"""Wide-kernel sweep: narrow layers (forced wavefront split) and long rows vs the other paths."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip

def run(N, m, C, M, settings):
    W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
    G = np.random.default_rng(1).standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
    alphabet = 3 * float(np.median(np.abs(W))) * np.linspace(-1, 1, M)
    Xd, Xqd, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(Xq).cuda(), torch.from_numpy(W.T.copy()).cuda()
    nrm = hip.row_norms(Xqd)
    ref, out = None, []
    for name, wpn, path in settings:
        hip.set_option("waves_per_neuron", wpn)
        best = 1e9
        for it in range(3):
            torch.cuda.synchronize(); t0 = time.time()
            r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm, path=path)
            torch.cuda.synchronize(); best = min(best, time.time() - t0)
        if ref is None: ref = r["idx"].clone()
        out.append(f"{name}: {best*1e3:7.2f}{'' if torch.equal(ref, r['idx']) else ' MISMATCH'}")
    hip.set_option("waves_per_neuron", 0)
    print(f"N={N} m={m} C={C} M={M} | " + " | ".join(out), flush=True)

narrow = [("rows32", 0, 1), ("wide2", 2, 1), ("wide4", 4, 1), ("wide8", 8, 1), ("wide16", 16, 1)]
for C in (4096, 2048, 1024, 512, 256, 64, 10):
    run(1024, 1024, C, 3, narrow)
run(1024, 2048, 512, 16, narrow)
long_rows = [("wide(auto)", 0, 1), ("stream", 0, 2)]
run(2048, 5008, 128, 8, long_rows)
run(784, 16384, 500, 3, long_rows)
run(512, 4096, 4096, 3, long_rows)
