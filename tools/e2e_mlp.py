"""End-to-end QuantizedNeuralNetwork.quantize_network() on the MNIST MLP of train_mnist_mlp.py:61-73
(Flatten, Dense(relu)+BatchNormalization per hidden width, Dense(softmax); random weights) with one batch of
`m` synthetic calibration samples as quantize_pretrained_mlp.py:73 feeds it.
usage: e2e_mlp.py [m] [widths, comma separated] [bits] [--profile] [--host-alphabet]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import keras_shim as ks, quantized_network as qn

args = [a for a in sys.argv[1:] if not a.startswith("--")]
m = int(args[0]) if len(args) > 0 else 25000
widths = [int(v) for v in args[1].split(",")] if len(args) > 1 else [500, 300, 100]
bits = float(args[2]) if len(args) > 2 else 4.0


class NullLogger:
    def __init__(self):
        self.n = 0

    def info(self, msg):
        self.n += 1


def build():
    L = [ks.Flatten(input_shape=(28, 28))]
    for w in widths:
        L += [ks.Dense(w, activation="relu"), ks.BatchNormalization()]
    L.append(ks.Dense(10, activation="softmax"))
    return ks.Sequential(L)


r = np.random.default_rng(0)
x = r.random((m, 28, 28)).astype(np.float32)
y = np.zeros((m, 10), dtype=np.float32)
for it in range(6 if "--more" in sys.argv else 3):
    net = build()
    log = NullLogger()
    q = qn.QuantizedNeuralNetwork(network=net, batch_size=m, get_data=qn.MNISTSequence(x, y, m), logger=log, bits=bits,
                                  alphabet_scalar=5)
    if "--host-alphabet" in sys.argv:                  # (A/B: round 5's Dense driver -- host alphabet, neuron-major copy, assembly pass)
        q._layer_alphabet_device = lambda layer_idx, rad: None
    torch.cuda.synchronize(); t0 = time.time()
    if it == 2 and "--profile" in sys.argv:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        q.quantize_network()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
    else:
        q.quantize_network()
    torch.cuda.synchronize(); total = time.time() - t0
    nw = sum(int(np.prod(l.get_weights()[0].shape)) for l in net.layers if l.__class__.__name__ == "Dense")
    print(f"run {it}: MLP 784-{'-'.join(map(str, widths))}-10, m={m}, {2**bits:.0f}-level alphabet: quantize_network() "
          f"{total*1e3:.1f} ms, {nw} weights, {nw/total:.3e} weights/s, {log.n} log lines")
