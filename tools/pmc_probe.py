"""Run the cfg2 dense kernel a few times (for rocprofv3 --pmc passes).
usage: pmc_probe.py M mode lanes_per_neuron variant tile_steps blk_sweep_waves      (env PMC_C / PMC_M: another width / row length)"""
import os
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import hip
arg = lambda i, d: int(sys.argv[i]) if len(sys.argv) > i else d
N = 4096; C = int(os.environ.get("PMC_C", "4096")); m = int(os.environ.get("PMC_M", "1024")); M = arg(1, 3)
W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
G = np.random.default_rng(1).standard_normal((N, m))
X = np.maximum(G, 0).astype(np.float32)
Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
rad = 3 * float(np.median(np.abs(W)))
alphabet = rad * np.linspace(-1, 1, M)
Xd, Xqd, Wt = torch.from_numpy(X).cuda(), torch.from_numpy(Xq).cuda(), torch.from_numpy(W.T.copy()).cuda()
hip.set_option("onchip_mode", arg(2, 1)); hip.set_option("lanes_per_neuron", arg(3, 0))
hip.set_option("variant", arg(4, 0)); hip.set_option("tile_steps", arg(5, 0)); hip.set_option("blk_sweep_waves", arg(6, 0))
nrm = hip.row_norms(Xqd)
if os.environ.get("PMC_LAYER", "0") != "0":
    # round 6: the kernel as bench.py's step runs it -- device-resident alphabet, the Keras kernel read in place, Q and the indices
    # written in the Keras layout by the kernel's own flush (the KOUT instantiation)
    from quantized_neural_networks_amd import layer
    Wd = torch.from_numpy(W).cuda()
    d = layer.layer_alphabet_device(Wd, np.linspace(-1, 1, M), 3)
    for _ in range(3):
        r = hip.quantize_dense_layer(Xd, Xqd, Wd, d, nrm32=nrm)
else:
    for _ in range(3):
        r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm)
torch.cuda.synchronize()
