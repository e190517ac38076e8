"""3x3 / stride 1 / SAME conv layers through the layer driver: shift form (option conv_shift = 1, default) against the
per-output-position form (0), results compared bit for bit.
usage: conv3x3_probe.py [n H W cin cout]... [--host-inputs] [--shift-only]   (default: ResNet50's 3x3 layers at 4096 images)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip, layer
HOST_INPUTS = "--host-inputs" in sys.argv
SHIFT_ONLY = "--shift-only" in sys.argv          # (traces: the shipped form only)
args = [int(v) for v in sys.argv[1:] if not v.startswith("--")]
shapes = [tuple(args[i:i + 5]) for i in range(0, len(args), 5)] or [(4096, 56, 56, 64, 64), (4096, 28, 28, 128, 128),
                                                                     (4096, 14, 14, 256, 256), (4096, 7, 7, 512, 512)]
for n, H, W, cin, cout in shapes:
    g = torch.Generator(device="cuda").manual_seed(2)
    if HOST_INPUTS:
        # inputs made on the HOST and uploaded: a kernel trace of this probe then holds the layer driver's kernels only (until round 3 the
        # RNG / element-wise kernels of the input generation were 40 % of the committed traces)
        r = np.random.default_rng(2)
        a = np.maximum(r.standard_normal((n, H, W, cin), dtype=np.float32), 0)
        act_w = torch.from_numpy(a).cuda()
        a += np.float32(0.05) * r.standard_normal((n, H, W, cin), dtype=np.float32)
        act_q = torch.from_numpy(np.maximum(a, 0, out=a)).cuda()
        del a
        Wk = torch.from_numpy((r.standard_normal((3, 3, cin, cout), dtype=np.float32) / 3)).cuda()
    else:
        act_w = torch.relu(torch.randn((n, H, W, cin), device="cuda", generator=g))
        act_q = torch.relu(act_w + 0.05 * torch.randn((n, H, W, cin), device="cuda", generator=g))
        Wk = torch.randn((3, 3, cin, cout), device="cuda", generator=g) / 3
    alphabet, rad = layer.layer_alphabet(Wk, np.linspace(-1, 1, 3), 3)
    ref = None
    for shift in ((1,) if SHIFT_ONLY else (0, 1)):
        hip.set_option("conv_shift", shift)
        best = 1e9
        for it in range(4):
            torch.cuda.synchronize(); t0 = time.time()
            out = layer.quantize_conv2d(Wk, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
            torch.cuda.synchronize(); best = min(best, time.time() - t0)
        same = "" if ref is None else f", equal to the other form: {bool(torch.equal(ref, out['Q']))}"
        ref = out["Q"] if ref is None else ref
        print(f"3x3 {cin}->{cout} @{H}x{W} n={n} conv_shift={shift}: {best*1e3:.2f} ms, host reruns {int(out['reruns'])}{same}")
    del act_w, act_q
hip.set_option("conv_shift", 1)
