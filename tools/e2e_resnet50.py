"""BASELINE cfg5 end to end: `QuantizedCNN.quantize_network()` on Keras-ResNet50's topology (53 Conv2D layers + the classifier; random
weights with trained-looking BatchNormalization statistics, synthetic mean-subtracted 224x224x3 images), ternary, alphabet_scalar 3 --
the reference's ImageNet driver flow (quantize_pretrained_imagenet.py:156-168) with the conv layers ON.  Round 5: the activation
capture walks the graph incrementally (live tensors of both networks, quantized_network._capture_incremental_graph) instead of
re-running two truncated models from the input for each of the 54 layers in 16-image batches (:456-462, :483-484).
usage: e2e_resnet50.py [n_images=4096] [batch=16] [--reference-capture n]   (--reference-capture: also time the reference's scheme on n images)"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from quantized_neural_networks_amd import keras_shim as K, quantized_network as qn

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if len(args) > 0 else 4096
batch = int(args[1]) if len(args) > 1 else 16


class Log:
    def info(self, msg):
        pass


def make_net(seed=1):
    net = K.ResNet50(input_shape=(224, 224, 3), classes=1000, seed=seed)
    g = np.random.default_rng(2)
    for layer in net.layers:
        if layer.__class__.__name__ == "BatchNormalization":
            c = layer.get_weights()[0].shape[0]
            layer.set_weights([g.uniform(0.8, 1.2, c), g.normal(0.1, 0.1, c), g.normal(0, 0.05, c), g.uniform(0.02, 0.06, c)])
    return net


def forward_all(net, x, chunk=512):
    """One forward pass of the whole network over all images (chunks of the capture grid), outputs dropped."""
    torch.cuda.synchronize(); t = time.time()
    for i in range(0, x.shape[0], chunk):
        net.predict_on_batch(x[i:i + chunk])
    torch.cuda.synchronize()
    return time.time() - t


def run(tag, x, incremental=True):
    net = make_net()
    q = qn.QuantizedCNN(network=net, batch_size=batch, get_data=qn.CIFAR10Sequence(x, np.zeros((len(x), 1), np.float32), batch),
                        logger=Log(), bits=np.log2(3), alphabet_scalar=3)
    q.incremental_capture = incremental
    cap = [0.0]
    inner = q._get_layer_data_generator

    def timed_capture(*a, **k):
        torch.cuda.synchronize(); t = time.time()
        out = inner(*a, **k)
        torch.cuda.synchronize(); cap[0] += time.time() - t
        return out
    q._get_layer_data_generator = timed_capture
    torch.cuda.reset_peak_memory_stats()
    torch.cuda.synchronize(); t0 = time.time()
    q.quantize_network()
    torch.cuda.synchronize(); dt = time.time() - t0
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    xd = q._raw_inputs()[0]
    del q._frontier
    fa = forward_all(net, xd); fq = forward_all(q.quantized_net, xd)
    nconv = sum(l.__class__.__name__ == "Conv2D" for l in net.layers)
    print(f"[{tag}] ResNet50 ({nconv} conv layers + classifier), {len(x)} images, batch {batch}: quantize_network() {dt:.3f} s = capture {cap[0]:.3f} s + "
          f"quantization and host {dt - cap[0]:.3f} s; one forward pass over all images: analog {fa:.3f} s, quantized {fq:.3f} s "
          f"(2 x both + 40 ms = {2 * (fa + fq) + 0.04:.3f} s); peak HBM {peak:.1f} GiB", flush=True)
    return q


g = np.random.default_rng(0)
# caffe-style preprocessed images (resnet_preprocess_input: BGR, ImageNet channel means subtracted): signed values
x = (g.random((n, 224, 224, 3), dtype=np.float32) * 255.0 - np.array([103.939, 116.779, 123.68], np.float32)).astype(np.float32)
run("cold: includes MIOpen's first-call solver search", x)
run("warm", x)
for a in sys.argv[1:]:
    if a.startswith("--reference-capture"):
        k = int(sys.argv[sys.argv.index(a) + 1]) if a == "--reference-capture" else int(a.split("=")[1])
        run(f"warm, incremental, {k} images", x[:k])
        run(f"reference's scheme (two truncated models per layer, batches of {batch}), {k} images", x[:k], incremental=False)
