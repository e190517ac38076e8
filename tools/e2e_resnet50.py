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


@torch.no_grad()
def make_net(x_cal, seed=1):
    """Random weights, but BatchNormalization statistics that a trained network would have: every BN layer's moving mean / variance
    are the statistics of ITS input over a calibration subset (a graph walk of the analog network), gamma in [0.8, 1.2], small
    beta.  With arbitrary statistics a 50-layer network multiplies its activations by a constant factor per block and overflows
    float32 half way down (first version of this tool: the quantized network's inputs of conv5_block3 were inf / NaN, every walk
    there went through the exact fallback, and the run measured that fallback)."""
    net = K.ResNet50(input_shape=(224, 224, 3), classes=1000, seed=seed)
    g = np.random.default_rng(2)
    inbound, last_use = net.graph_tables()
    vals = {0: torch.from_numpy(x_cal).to(net.device)}
    for k, layer in enumerate(net.layers):
        if k == 0:
            continue
        xs = [vals[p] for p in inbound[k]]
        if layer.__class__.__name__ == "BatchNormalization":
            c = xs[0].shape[-1]
            mu = xs[0].mean(dim=(0, 1, 2)); var = xs[0].var(dim=(0, 1, 2), unbiased=False).clamp_min(1e-6)
            layer.set_weights([g.uniform(0.8, 1.2, c).astype(np.float32), g.normal(0.05, 0.05, c).astype(np.float32), mu, var])
        one = len(xs) == 1 and not isinstance(layer.inbound_nodes[0].inbound_layers, (list, tuple))
        vals[k] = layer.call(xs[0] if one else xs)
        for p in inbound[k]:
            if last_use[p] == k:
                del vals[p]
    return net


def forward_all(net, x, chunk=512):
    """One forward pass of the whole network over all images (chunks of the capture grid), outputs dropped."""
    torch.cuda.synchronize(); t = time.time()
    for i in range(0, x.shape[0], chunk):
        net.predict_on_batch(x[i:i + chunk])
    torch.cuda.synchronize()
    return time.time() - t


def run(tag, x, incremental=True):
    net = make_net(x[:min(len(x), 64)])
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
    per_layer = {}
    inner_conv, inner_dense = q._quantize_conv2D_layer_parallel_jit, q._quantize_dense_layer

    def timed_layer(fn):
        def wrapped(k):
            torch.cuda.synchronize(); t = time.time(); c0 = cap[0]
            fn(k)
            torch.cuda.synchronize(); per_layer[k] = (time.time() - t) - (cap[0] - c0)      # the layer without its capture
        return wrapped
    if "--per-layer" in sys.argv:
        q._quantize_conv2D_layer_parallel_jit = timed_layer(inner_conv)
        q._quantize_dense_layer = timed_layer(inner_dense)
    torch.cuda.reset_peak_memory_stats()
    prof = None
    if "--profile" in sys.argv and tag.startswith("warm"):
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    torch.cuda.synchronize(); t0 = time.time()
    q.quantize_network()
    torch.cuda.synchronize(); dt = time.time() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    xd = q._raw_inputs()[0]
    q.__dict__.pop("_frontier", None)
    fa = forward_all(net, xd); fq = forward_all(q.quantized_net, xd)
    nconv = sum(l.__class__.__name__ == "Conv2D" for l in net.layers)
    print(f"[{tag}] ResNet50 ({nconv} conv layers + classifier), {len(x)} images, batch {batch}: quantize_network() {dt:.3f} s = capture {cap[0]:.3f} s + "
          f"quantization and host {dt - cap[0]:.3f} s; one forward pass over all images: analog {fa:.3f} s, quantized {fq:.3f} s "
          f"(2 x both + 40 ms = {2 * (fa + fq) + 0.04:.3f} s); peak HBM {peak:.1f} GiB", flush=True)
    if per_layer:
        worst = sorted(per_layer.items(), key=lambda kv: -kv[1])[:8]
        rer = {k: int(q.last_layer_stats[k].get("reruns", 0) or 0) for k in per_layer}
        print(f"    quantization without capture, all layers: {sum(per_layer.values()):.3f} s; reruns through the streaming kernel: {sum(rer.values())}; slowest: "
              + ", ".join(f"{net.layers[k].name} {v * 1e3:.1f} ms (reruns {rer[k]})" for k, v in worst), flush=True)
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
