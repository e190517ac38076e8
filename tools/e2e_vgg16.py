"""End-to-end on the reference's published ImageNet experiment shape (model_metrics/ternary_vgg16_experiment.csv:
VGG16, Dense layers only (is_quantize_conv2d = FALSE), q_train_size = 1500, ternary, quantization_time ~ 15 300 s on
the authors' CPU box): Keras-VGG16 architecture with random weights, synthetic 224x224x3 images.
usage: e2e_vgg16.py [n_images] [batch] [capture chunk]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import keras_shim as ks, quantized_network as qn

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 500


class Log:
    def __init__(self):
        self.t0 = time.time()

    def info(self, msg):
        if "Neuron" not in msg and "-v" in sys.argv:
            print(f"[{time.time()-self.t0:7.2f}s] {msg}", flush=True)


def vgg16():
    L = []
    first = True
    for filters, reps in ((64, 2), (128, 2), (256, 3), (512, 3), (512, 3)):
        for _ in range(reps):
            kw = dict(input_shape=(224, 224, 3)) if first else {}
            first = False
            L.append(ks.Conv2D(filters, (3, 3), activation="relu", padding="same", **kw))
        L.append(ks.MaxPooling2D((2, 2), strides=(2, 2)))
    L += [ks.Flatten(), ks.Dense(4096, activation="relu"), ks.Dense(4096, activation="relu"), ks.Dense(1000, activation="softmax")]
    return ks.Sequential(L)


def run(tag):
    torch.manual_seed(0)
    net = vgg16()
    log = Log()
    q = qn.QuantizedCNN(network=net, batch_size=batch, get_data=qn.CIFAR10Sequence(x, y, batch), logger=log, bits=np.log2(3),
                        alphabet_scalar=3, is_quantize_conv2d=False)
    q._capture_chunk = chunk                 # bound the temporaries of the 224x224 conv layers
    cap = [0.0]
    inner = q._get_layer_data_generator

    def timed_capture(*a, **k):              # activation capture (forward passes) vs the quantization proper
        torch.cuda.synchronize(); t = time.time()
        out = inner(*a, **k)
        torch.cuda.synchronize(); cap[0] += time.time() - t
        return out
    q._get_layer_data_generator = timed_capture
    torch.cuda.synchronize(); t0 = time.time()
    q.quantize_network()
    torch.cuda.synchronize(); dt = time.time() - t0
    nw = sum(int(np.prod(l.get_weights()[0].shape)) for l in net.layers if l.__class__.__name__ == "Dense")
    print(f"[{tag}] VGG16 Dense layers (fc1 25088->4096, fc2, predictions; {nw} weights), {n} images: quantize_network() {dt:.2f} s "
          f"= activation capture {cap[0]:.2f} s (one fp32 forward pass of {n} images through 13 conv layers, PyTorch/MIOpen) + quantization "
          f"{dt - cap[0]:.2f} s (reference, published: ~15300 s); peak GPU memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    return dt


chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 250
x = np.random.default_rng(0).random((n, 224, 224, 3), dtype=np.float32)
y = np.zeros((n, 1), dtype=np.float32)
# First call in a process: MIOpen searches a solver for each of the 13 conv shapes (seconds each; the choice is cached for the
# process and, where the user database is writable, on disk).  That is a one-time cost per machine, reported separately: the
# second run is what every later network of these shapes costs.
run("cold: includes MIOpen's first-call solver search")
run("warm")
