"""Row-group kernel at BASELINE cfg2: lanes per neuron x neurons per workgroup x tile steps.
usage: lpn_quick.py [N C m M]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip

N, C, m, M = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (4096, 4096, 1024, 3)
W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
G = np.random.default_rng(1).standard_normal((N, m))
X = np.maximum(G, 0).astype(np.float32)
Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
alphabet = 3 * float(np.median(np.abs(W))) * np.linspace(-1, 1, M)
Xd, Xqd, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(Xq).cuda(), torch.from_numpy(W.T.copy()).cuda()
nrm = hip.row_norms(Xqd)
ref = None
for lpn, gs, ts in [(32, 0, 0), (32, 8, 0), (32, 0, 4), (16, 16, 0), (16, 16, 4), (64, 0, 0)]:
    hip.set_option("lanes_per_neuron", lpn); hip.set_option("group_waves", gs); hip.set_option("tile_steps", ts)
    best = 1e9
    try:
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm)
            torch.cuda.synchronize(); best = min(best, time.time() - t0)
    except hip.GpfqError as e:
        print(f"lpn={lpn} gs={gs} ts={ts}: {e}")
        continue
    if ref is None:
        ref = r["idx"].clone()
    print(f"lpn={lpn:2d} neurons/wg={gs or 16:2d} tile_steps={ts}: {best*1e3:7.3f} ms  same={bool(torch.equal(ref, r['idx']))}", flush=True)
for k in ("lanes_per_neuron", "group_waves", "tile_steps"):
    hip.set_option(k, 0)
