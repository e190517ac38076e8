import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
N, C, m = 784, 500, 25000
for kind in ("relu", "uniform"):
    g = torch.Generator(device="cuda").manual_seed(1)
    if kind == "relu":
        G = torch.randn((N, m), device="cuda", generator=g); X = torch.relu(G); Xq = torch.relu(G + 0.1 * torch.randn((N, m), device="cuda", generator=g))
    else:
        X = torch.rand((N, m), device="cuda", generator=g); Xq = torch.relu(X + 0.05 * torch.randn((N, m), device="cuda", generator=g))
    W = torch.randn((N, C), device="cuda", generator=g) / np.sqrt(N)
    Wt = W.t().contiguous()
    for M in (3, 16):
        alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, M), 3 if M == 3 else 5)
        for sl in (0, -1, -3):
            hip.set_option("gram_slack_log2", sl)
            for it in range(3):
                torch.cuda.synchronize(); t0 = time.time()
                r = hip.quantize_neurons(X, Xq, Wt, alphabet, path=3, want_values=False, want_resid=None)
                torch.cuda.synchronize(); dt = time.time() - t0
            print(f"{kind} M={M} slack 2^{sl}: {dt*1e3:.2f} ms, host reruns {r['uncertified']}")
hip.set_option("gram_slack_log2", 0)
