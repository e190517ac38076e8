import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
g = torch.Generator(device="cuda").manual_seed(2)
for (n, hw, cin, cout) in [(5008, 32, 3, 32), (5008, 32, 16, 32), (5008, 32, 32, 32), (5008, 16, 32, 64), (4096, 56, 8, 8)]:
    act_w = torch.rand((n, hw, hw, cin), device="cuda", generator=g)
    act_q = torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device="cuda", generator=g))
    W = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / 3
    alphabet, rad = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
    res = []
    for nh in (1, 0):
        hip.set_option("conv_nhwc", nh)
        best = 1e9
        for it in range(6):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        res.append(best * 1e3)
    hip.set_option("conv_nhwc", 1)
    print(f"3x3 {cin}->{cout} @{hw}x{hw} n={n}: NHWC form {res[0]:.2f} ms, planes form {res[1]:.2f} ms")
