"""Narrow dense layers with long rows (cfg4's Dense(2048->128), m = 5008): wavefronts per neuron and the
direct (register-prefetch) mode of the wide kernel against its LDS-staged mode (option variant = 2, i.e. bit 1).
usage: narrow_quick.py [N m C M]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip

N, m, C, M = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (2048, 5008, 128, 8)
W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
G = np.random.default_rng(1).standard_normal((N, m))
X = np.maximum(G, 0).astype(np.float32)
Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
alphabet = 3 * float(np.median(np.abs(W))) * np.linspace(-1, 1, M)
Xd, Xqd, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(Xq).cuda(), torch.from_numpy(W.T.copy()).cuda()
nrm = hip.row_norms(Xqd)
ref = None
for wpn in (0, 2, 4, 5, 8, 10, 16):
    for variant in (0, 2):
        hip.set_option("waves_per_neuron", wpn)
        hip.set_option("variant", variant)
        best = 1e9
        try:
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.time()
                r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm, path=1)
                torch.cuda.synchronize(); best = min(best, time.time() - t0)
        except hip.GpfqError as e:
            print(f"wpn={wpn} variant={variant}: {e}")
            continue
        if ref is None:
            ref = r["idx"].clone()
        mode = {0: "direct", 2: "lds   "}[variant]                  # variant bit 2: LDS-staged rows
        print(f"N={N} m={m} C={C} M={M} wpn={wpn:2d} {mode}: {best*1e3:7.2f} ms  "
              f"{best/N*1e6:.2f} us/step  same={bool(torch.equal(ref, r['idx']))}", flush=True)
hip.set_option("waves_per_neuron", 0)
hip.set_option("variant", 0)
