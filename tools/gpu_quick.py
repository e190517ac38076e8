"""Quick on-GPU timing of the dense kernel at BASELINE cfg2 (N=C=4096, m=1024, ternary)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip
import oracle

N = C = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
M = int(sys.argv[3]) if len(sys.argv) > 3 else 3
scalar = 3
W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
G = np.random.default_rng(1).standard_normal((N, m))
X = np.maximum(G, 0).astype(np.float32)
Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
alphabet, rad = oracle.layer_alphabet(W, np.linspace(-1, 1, M), scalar)
Xd, Xqd, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(Xq).cuda(), torch.from_numpy(W.T.copy()).cuda()
nrm = hip.row_norms(Xqd)
for mode in (0, 1):
    hip.set_option("onchip_mode", mode)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm)
        torch.cuda.synchronize(); dt = time.time() - t0
        print(f"mode {mode} iter {it}: {dt*1e3:.2f} ms  {N*C/dt:.3e} weights/s  alg-GB/s {N*C*(8*m+8)/dt/1e9:.1f}  fallbacks {hip.exact_fallbacks(r)}")
nchk = 32
t0 = time.time(); Qo, io, ro = oracle.layer(W, X, Xq, alphabet, 0, nchk); dt = time.time() - t0
print(f"oracle {nchk} neurons {dt:.2f}s threads={oracle.num_threads()} -> {nchk*N/dt:.3e} weights/s")
idx = r["idx"][:nchk].cpu().numpy()
print("idx mismatching neurons:", int((idx != io).any(axis=1).sum()), "max resid rel err",
      float(np.max(np.abs(r["resid"][:nchk].cpu().numpy() - ro) / ro)))
