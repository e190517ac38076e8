"""Gram-path timing for one conv channel (N=9, m=5.1M, 32 filters): tile-shape variants."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip
N, m, C = 9, 5128192, 32
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, m), device="cuda", generator=g)
Xq = torch.relu(X + 0.05 * torch.randn((N, m), device="cuda", generator=g))
Wt = torch.randn((C, N), device="cuda", generator=g) / 3
alphabet = 0.5 * np.linspace(-1, 1, 8)
plan = hip.GramPlan(N, m, C, alphabet, X.device)
idx = torch.empty((C, N), dtype=torch.int8, device="cuda"); Q = torch.empty((C, N), device="cuda")
res = torch.empty(C, dtype=torch.float64, device="cuda"); unc = torch.empty(C, dtype=torch.int32, device="cuda")
ref = None
for var in (0, 1, 3):
    hip.set_option("variant", var)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        plan.run(X, Xq, Wt, idx, Q, res, unc)
        torch.cuda.synchronize(); dt = time.time() - t0
    if ref is None: ref = idx.clone()
    print(f"variant {var}: {dt*1e3:.3f} ms  same={bool(torch.equal(ref, idx))} uncertified={int(unc.sum())}")
hip.set_option("variant", 0)
