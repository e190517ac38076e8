"""GPU: graph (functional-API) networks through the class surface.  The reference reaches every layer with ONE inbound
layer through truncated ``Model(inputs=net.layers[0].input, outputs=[inbound.output])`` objects
(scripts/quantized_network.py:432-462), which covers Keras-ResNet50's convolutions (quantize_pretrained_imagenet.py:10
imports it; BASELINE cfg5): skip connections, layers with several consumers, ``inbound_layers`` that is a layer or a list.
Conv results are checked against the C oracle on the activations the class captured (patches by tests/_im2col_ref.py)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _im2col_ref import patches as ref_patches  # noqa: E402

pytestmark = pytest.mark.gpu


class Quiet:
    def info(self, msg):
        pass


def _record_captures(q):
    rec = {}
    orig = q._get_layer_data_generator

    def wrapped(layer_idx, transpose=False):
        wX, qX = orig(layer_idx, transpose)
        rec[layer_idx] = (wX, qX)
        return wX, qX

    q._get_layer_data_generator = wrapped
    return rec


def _check_conv_layer(q, rec, idx, oracle_mod, pairs):
    """(channel, filter) pairs of conv layer idx against the oracle on the captured inputs of that layer."""
    layer = q.trained_net.layers[idx]
    W = layer.get_weights()[0]
    Q = q.quantized_net.layers[idx].get_weights()[0]
    st = q.last_layer_stats[idx]
    alphabet, rad = oracle_mod.layer_alphabet(W, q.alphabet, q.alphabet_scalar)
    assert rad == st["rad"] and np.array_equal(alphabet, st["alphabet"])
    wX, qX = (t.cpu().numpy() for t in rec[idx])
    kh, kw = layer.kernel_size
    rh, rw = layer.dilation_rate
    for c, f in pairs:
        Pw = ref_patches(wX, c, kh, kw, layer.strides[0], layer.strides[1], rh, rw, layer.padding.upper())
        Pq = ref_patches(qX, c, kh, kw, layer.strides[0], layer.strides[1], rh, rw, layer.padding.upper())
        qo, _, _ = oracle_mod.neuron(W[:, :, c, f].reshape(-1), Pw, Pq, alphabet)
        assert np.array_equal(Q[:, :, c, f].reshape(-1), qo.astype(np.float32)), (layer.name, c, f)


def test_small_graph_with_skip_connection(oracle_mod):
    from quantized_neural_networks_amd import keras_shim as K, quantized_network as qn
    x = K.Input((12, 12, 3))
    a = K.Conv2D(6, 3, padding="same", activation="relu", name="stem")(x)          # two consumers: the branch and the shortcut
    b = K.Conv2D(6, 3, padding="same", name="branch")(a)
    b = K.BatchNormalization(name="branch_bn")(b)
    s = K.Add(name="add")([a, b])                                                   # inbound_layers is a LIST here
    y = K.Activation("relu", name="out")(s)
    z = K.Conv2D(4, 3, strides=2, padding="same", name="down")(y)                   # single inbound layer: Activation
    z = K.GlobalAveragePooling2D(name="gap")(z)
    o = K.Dense(5, activation="softmax", name="head")(z)
    net = K.Model(x, o, seed=3)
    assert [l.__class__.__name__ for l in net.layers][:2] == ["InputLayer", "Conv2D"]
    add = next(l for l in net.layers if l.name == "add")
    assert isinstance(add.inbound_nodes[0].inbound_layers, list) and len(add.inbound_nodes[0].inbound_layers) == 2
    assert not isinstance(net.layers[1].inbound_nodes[0].inbound_layers, list)
    r = np.random.default_rng(0)
    X = r.random((40, 12, 12, 3)).astype(np.float32)
    y0 = np.zeros((40, 5), dtype=np.float32)
    q = qn.QuantizedCNN(network=net, batch_size=16, get_data=qn.CIFAR10Sequence(X, y0, 16), logger=Quiet(), bits=3, alphabet_scalar=4)
    assert q._incremental_capture_possible() and q._graph_capture_possible()   # round 5: graph networks walk their live tensors
    rec = _record_captures(q)
    q.quantize_network()
    names = {l.name: k for k, l in enumerate(net.layers)}
    for name in ("stem", "branch", "down"):
        _check_conv_layer(q, rec, names[name], oracle_mod, [(0, 0), (1, 2), (2, 3)])
    # capture layout: 40 samples in batches of 16 -> 48 rows, the partial-last-batch quirk (:491-495)
    assert rec[names["down"]][0].shape[0] == 48
    # the head's inputs come through both paths of the graph; its result equals the oracle's on the captured inputs
    k = names["head"]
    Wh = net.layers[k].get_weights()[0]
    alphabet, _ = oracle_mod.layer_alphabet(Wh, q.alphabet, q.alphabet_scalar)
    Qo, _, _ = oracle_mod.layer(Wh, rec[k][0].cpu().numpy(), rec[k][1].cpu().numpy(), alphabet)
    assert np.array_equal(q.quantized_net.layers[k].get_weights()[0], Qo.T.astype(np.float32))
    # the quantized graph still evaluates, and differs from the analog one
    pa, pq = net.predict_on_batch(X[:8]), q.quantized_net.predict_on_batch(X[:8])
    assert tuple(pq.shape) == (8, 5) and not torch.equal(pa, pq)


def test_resnet50_topology_end_to_end_reduced_images(oracle_mod, tmp_path):
    """BASELINE cfg5's network: all 53 Conv2D layers + the classifier of Keras-ResNet50's topology, ternary, scalar 3,
    at a reduced image count and size (24 images of 64 x 64; the full-size layers are tests/test_fullsize_configs.py)."""
    from quantized_neural_networks_amd import keras_shim as K, quantized_network as qn
    net = K.ResNet50(input_shape=(64, 64, 3), classes=10, seed=1)
    convs = [k for k, l in enumerate(net.layers) if l.__class__.__name__ == "Conv2D"]
    assert len(convs) == 53 and sum(l.__class__.__name__ == "Dense" for l in net.layers) == 1
    # "trained-looking" BatchNormalization statistics so that activations do not die out in a random network
    g = np.random.default_rng(2)
    for layer in net.layers:
        if layer.__class__.__name__ == "BatchNormalization":
            c = layer.get_weights()[0].shape[0]
            layer.set_weights([g.uniform(0.8, 1.2, c), g.normal(0.1, 0.1, c), g.normal(0, 0.05, c), g.uniform(0.02, 0.06, c)])
    X = g.random((24, 64, 64, 3)).astype(np.float32)
    q = qn.QuantizedCNN(network=net, batch_size=8, get_data=qn.CIFAR10Sequence(X, np.zeros((24, 10), np.float32), 8),
                        logger=Quiet(), bits=np.log2(3), alphabet_scalar=3)
    rec = _record_captures(q)
    q.quantize_network()
    names = {l.name: k for k, l in enumerate(net.layers)}
    for k in convs + [names["predictions"]]:
        Wq = q.quantized_net.layers[k].get_weights()[0]
        assert len(np.unique(Wq)) <= 3, net.layers[k].name                       # ternary: -rad, 0, +rad
        assert not np.array_equal(Wq, net.layers[k].get_weights()[0])
        assert np.array_equal(q.quantized_net.layers[k].get_weights()[1], net.layers[k].get_weights()[1])   # bias carried over
    # oracle samples: conv1 (7x7/2 VALID after ZeroPadding), a 3x3 'same', a strided 1x1 projection shortcut, the last 3x3
    _check_conv_layer(q, rec, names["conv1_conv"], oracle_mod, [(0, 0), (2, 63)])
    _check_conv_layer(q, rec, names["conv2_block1_2_conv"], oracle_mod, [(0, 0), (63, 5)])
    _check_conv_layer(q, rec, names["conv3_block1_0_conv"], oracle_mod, [(7, 100), (255, 511)])
    _check_conv_layer(q, rec, names["conv5_block3_2_conv"], oracle_mod, [(11, 400)])
    # N4: the quantized graph network survives save_model / load_model on the GPU (quantize_pretrained_imagenet.py:180-191)
    path = tmp_path / "quantized_resnet50"
    K.save_model(q.quantized_net, path)
    back = K.load_model(path, device="cuda")
    assert torch.equal(back.predict_on_batch(X[:4]), q.quantized_net.predict_on_batch(X[:4]))


def test_graph_incremental_capture_equals_the_recomputation_path_on_the_gpu():
    """The incremental graph capture (live tensors of both networks, one evaluation per layer) against the reference's scheme (two
    truncated Models per layer, re-run from the input: scripts/quantized_network.py:456-462) on the small ResNet50 topology, on the
    GPU: every captured activation tensor and every quantized kernel bit for bit.  One feeder batch holds all images and one
    capture chunk holds them too, so both schemes hand MIOpen / the GEMMs the same shapes (a convolution's algorithm, and with it
    the last bits, can depend on the batch size; the partial-batch layout is compared on the CPU: tests/test_graph_capture_cpu.py)."""
    from quantized_neural_networks_amd import keras_shim as K, quantized_network as qn
    net = K.ResNet50(input_shape=(64, 64, 3), classes=10, seed=1)
    g = np.random.default_rng(2)
    for layer in net.layers:
        if layer.__class__.__name__ == "BatchNormalization":
            c = layer.get_weights()[0].shape[0]
            layer.set_weights([g.uniform(0.8, 1.2, c), g.normal(0.1, 0.1, c), g.normal(0, 0.05, c), g.uniform(0.02, 0.06, c)])
    X = (g.random((24, 64, 64, 3)) * 255 - 110).astype(np.float32)              # signed, like preprocessed images
    # (MIOpen may choose a convolution solver by the memory it finds free; the deterministic setting pins the choice so that two
    #  evaluations of one layer on one input in one process give the same bits -- what this comparison rests on)
    det = torch.backends.cudnn.deterministic
    torch.backends.cudnn.deterministic = True
    runs = []
    for incremental in (True, False):
        q = qn.QuantizedCNN(network=net, batch_size=24, get_data=qn.CIFAR10Sequence(X, np.zeros((24, 10), np.float32), 24),
                            logger=Quiet(), bits=np.log2(3), alphabet_scalar=3)
        q.incremental_capture = incremental
        assert q._incremental_capture_possible() == incremental
        rec = _record_captures(q)
        q.quantize_network()
        runs.append((q, rec))
    torch.backends.cudnn.deterministic = det
    (qi, ri), (qr, rr) = runs
    assert sorted(ri) == sorted(rr) and len(ri) == 54
    for k in sorted(ri):
        for which, a, b in (("analog", ri[k][0], rr[k][0]), ("quantized", ri[k][1], rr[k][1])):
            assert torch.equal(a, b), (f"{which} inputs of layer {k} ({net.layers[k].name}) differ: max |diff| "
                                       f"{float((a - b).abs().max()):.3e} on values up to {float(a.abs().max()):.3e}")
        assert np.array_equal(qi.quantized_net.layers[k].get_weights()[0], qr.quantized_net.layers[k].get_weights()[0]), net.layers[k].name
