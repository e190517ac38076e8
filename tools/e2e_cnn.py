"""End-to-end QuantizedCNN.quantize_network() on BASELINE cfg4: the CIFAR10 CNN of train_cifar10_cnn.py:63-86
(random weights), 5000 synthetic calibration images in batches of 16 (=> 5008 columns with the partial-batch
quirk), 3 bits, alphabet_scalar 4.  Prints where the wall time goes: activation capture vs quantization.
usage: e2e_cnn.py [n_images] [batch] [--bn] [--profile] [--host-alphabet]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import keras_shim as ks, quantized_network as qn

_pos = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(_pos[0]) if len(_pos) > 0 else 5000
batch = int(_pos[1]) if len(_pos) > 1 else 16


class TimingLogger:
    def __init__(self):
        self.t0 = time.time()
        self.lines = []

    def info(self, msg):
        self.lines.append((time.time() - self.t0, msg))


def build():
    L = [ks.Conv2D(32, (3, 3), activation="relu", padding="same", input_shape=(32, 32, 3)), ks.BatchNormalization(),
         ks.Conv2D(32, (3, 3), activation="relu", padding="same"), ks.BatchNormalization(), ks.MaxPooling2D((2, 2)), ks.Dropout(0.2),
         ks.Conv2D(64, (3, 3), activation="relu", padding="same"), ks.BatchNormalization(),
         ks.Conv2D(64, (3, 3), activation="relu", padding="same"), ks.BatchNormalization(), ks.MaxPooling2D((2, 2)), ks.Dropout(0.3),
         ks.Conv2D(128, (3, 3), activation="relu", padding="same"), ks.BatchNormalization(),
         ks.Conv2D(128, (3, 3), activation="relu", padding="same"), ks.BatchNormalization(), ks.MaxPooling2D((2, 2)), ks.Dropout(0.4),
         ks.Flatten(), ks.Dense(128, activation="relu"), ks.BatchNormalization(), ks.Dropout(0.5), ks.Dense(10, activation="softmax")]
    net = ks.Sequential(L)
    if "--bn" in sys.argv:          # trained-looking BatchNorm statistics: the conv inputs become signed
        g = np.random.default_rng(5)
        for layer in net.layers:
            if layer.__class__.__name__ == "BatchNormalization":
                c = layer.get_weights()[0].shape[0]
                layer.set_weights([g.uniform(0.5, 1.5, c), g.normal(0, 0.3, c), g.normal(0.3, 0.3, c), g.uniform(0.5, 1.5, c)])
    return net


r = np.random.default_rng(0)
x = r.random((n, 32, 32, 3)).astype(np.float32)
y = np.zeros((n, 10), dtype=np.float32)
for it in range(4):
    net = build()
    log = TimingLogger()
    q = qn.QuantizedCNN(network=net, batch_size=batch, get_data=qn.CIFAR10Sequence(x, y, batch), logger=log, bits=3, alphabet_scalar=4)
    if "--host-alphabet" in sys.argv:         # (A/B: round 5's Dense driver -- host alphabet, neuron-major copy, assembly pass)
        q._layer_alphabet_device = lambda layer_idx, rad: None
    q.lookahead_capture = it >= 2            # runs 2, 3: with the analog look-ahead on the second stream (default on a single GPU: off)
    torch.cuda.synchronize(); t0 = time.time()
    if it == 1 and "--profile" in sys.argv:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        q.quantize_network()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats("quantized_neural_networks_amd|method|built-in", 40)
    else:
        q.quantize_network()
    torch.cuda.synchronize(); total = time.time() - t0
    print(f"run {it}: quantize_network() {total*1e3:.1f} ms for {n} images, batch {batch}" + ("  (analog look-ahead on)" if q.lookahead_capture else ""))
# phase split of the last run from the log: "Feeding input data ... done. X seconds."
feed = 0.0
lines = [m for _, m in log.lines]
for i, m in enumerate(lines):
    if "Feeding input data" in m and i + 1 < len(lines) and "done." in lines[i + 1]:
        feed += float(lines[i + 1].split("done.")[1].split("seconds")[0])
print("host reruns per conv layer:", {k: v.get("reruns") for k, v in q.last_layer_stats.items() if "reruns" in v})
print(f"activation capture (a8): {feed*1e3:.1f} ms; everything else (quantization, host copies, logging): {(total-feed)*1e3:.1f} ms")
