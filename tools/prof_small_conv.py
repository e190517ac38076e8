import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
cin = cout = 128; hw = 8; n = 5008
g = torch.Generator(device="cuda").manual_seed(0)
act_w = torch.rand((n, hw, hw, cin), device="cuda", generator=g)
act_q = torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device="cuda", generator=g))
W = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / 3
alphabet, rad = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
for it in range(2):
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1))
torch.cuda.synchronize()
