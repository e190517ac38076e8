"""Gram path on patch matrices in memory (gpfq_quantize_neurons_gram): one conv channel's worth,
N = 9 rows of m = 5.1 M columns, 32 filters.  usage: gram_quick.py [N m C]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip
N, m, C = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (9, 5128192, 32)
g = torch.Generator(device="cuda").manual_seed(0)
X = torch.rand((N, m), device="cuda", generator=g)
Xq = torch.relu(X + 0.05 * torch.randn((N, m), device="cuda", generator=g))
Wt = torch.randn((C, N), device="cuda", generator=g) / 3
alphabet = 0.5 * np.linspace(-1, 1, 8)
plan = hip.GramPlan(N, m, C, alphabet, X.device)
idx = torch.empty((C, N), dtype=torch.int8, device="cuda"); Q = torch.empty((C, N), device="cuda")
res = torch.empty(C, dtype=torch.float64, device="cuda"); unc = torch.empty(C, dtype=torch.int32, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    plan.run(X, Xq, Wt, idx, Q, res, unc)
    torch.cuda.synchronize(); dt = time.time() - t0
print(f"N={N} m={m} C={C}: {dt*1e3:.3f} ms (Gram record + decide + residual replay), still flagged {int((unc != 0).sum())}")
