"""CPU: the incremental activation capture on GRAPH networks (round 5) against the reference's scheme -- two truncated Models per layer,
re-run from the input in the feeder's batches (scripts/quantized_network.py:456-462, :483-484) -- on the torch-backed shim with the device
set to the CPU (the capture is host logic + torch plumbing; no HIP call).  "Quantization" is simulated by overwriting a layer's kernel in
the quantized network through _update_weights, exactly where quantize_network() would."""
import numpy as np
import pytest
import torch

from quantized_neural_networks_amd import keras_shim as K, quantized_network as qn


class _Quiet:
    def info(self, msg):
        pass


def _skip_net():
    x = K.Input((10, 10, 3))
    a = K.Conv2D(5, 3, padding="same", activation="relu", name="stem")(x)
    b = K.Conv2D(5, 3, padding="same", name="branch")(a)
    b = K.BatchNormalization(name="branch_bn")(b)
    s = K.Add(name="add")([a, b])
    y = K.Activation("relu", name="out")(s)
    z = K.Conv2D(4, 3, strides=2, padding="same", name="down")(y)
    z = K.GlobalAveragePooling2D(name="gap")(z)
    o = K.Dense(6, activation="softmax", name="head")(z)
    return K.Model(x, o, seed=3, device="cpu")


def _pair(net, X, batch, **kw):
    mk = lambda: qn.QuantizedCNN(network=net, batch_size=batch, get_data=qn.CIFAR10Sequence(X, np.zeros((len(X), 1), np.float32), batch),
                                 logger=_Quiet(), bits=2, alphabet_scalar=2, device="cpu", **kw)
    inc, ref = mk(), mk()
    ref.incremental_capture = False
    assert inc._incremental_capture_possible() and inc._graph_capture_possible() and not ref._incremental_capture_possible()
    return inc, ref


def _walk(inc, ref, transpose_dense=True):
    """Capture every quantizable layer front to back with both schemes, 'quantizing' each in between; every pair of tensors equal."""
    net = inc.trained_net
    n_checked = 0
    for k, layer in enumerate(net.layers):
        name = layer.__class__.__name__
        if name not in ("Conv2D", "Dense"):
            continue
        tr = name == "Dense" and transpose_dense
        wi, qi = inc._get_layer_data_generator(k, transpose=tr)
        wr, qr = ref._get_layer_data_generator(k, transpose=tr)
        assert wi.shape == wr.shape and torch.equal(wi, wr), (k, layer.name)
        assert torch.equal(qi, qr), (k, layer.name)
        n_checked += 1
        W = layer.get_weights()[0]
        Q = (np.sign(W) * np.median(np.abs(W))).astype(np.float32)        # a stand-in for the quantized kernel
        for q in (inc, ref):
            q._update_weights(k, Q)
    return n_checked


def test_graph_capture_equals_truncated_models_on_a_skip_connection():
    net = _skip_net()
    X = np.random.default_rng(0).random((40, 10, 10, 3)).astype(np.float32)
    inc, ref = _pair(net, X, 16)                       # 40 samples in batches of 16: the partial-last-batch layout (:491-495) too
    assert _walk(inc, ref) == 4
    # the frontier holds live tensors only: after the last capture (the head, behind the pooling) one tensor per network
    fr = inc._frontier
    assert fr["graph"] and len(fr["w"]) == 1 and len(fr["q"]) == 1


def test_graph_capture_equals_truncated_models_with_fixed_partial_batch():
    net = _skip_net()
    X = np.random.default_rng(1).random((20, 10, 10, 3)).astype(np.float32)
    inc, ref = _pair(net, X, 8, fix_partial_batch=True)
    _walk(inc, ref)


def test_graph_capture_resnet50_topology_small():
    """Keras-ResNet50's graph (53 conv layers + the classifier, projection shortcuts, 16 Adds) on 6 images of 32 x 32."""
    net = K.ResNet50(input_shape=(32, 32, 3), classes=7, device="cpu", seed=1)
    g = np.random.default_rng(2)
    for layer in net.layers:
        if layer.__class__.__name__ == "BatchNormalization":
            c = layer.get_weights()[0].shape[0]
            layer.set_weights([g.uniform(0.8, 1.2, c), g.normal(0.1, 0.1, c), g.normal(0, 0.05, c), g.uniform(0.02, 0.06, c)])
    X = g.random((6, 32, 32, 3)).astype(np.float32)
    inc, ref = _pair(net, X, 3)
    inc._capture_chunk = 4                             # two chunks per layer: the chunked calls of merging layers too
    ref._capture_chunk = 4
    assert _walk(inc, ref) == 54
    # live set while walking a bottleneck block: never more than the block input (shortcut) + the branch tensors
    assert max(len(inc._frontier["w"]), len(inc._frontier["q"])) <= 3


def test_feeder_subclass_with_its_own_getitem_is_iterated_batch_by_batch():
    """a Sequence subclass that overrides __getitem__ (here: scaling) must not take the one-array fast path."""
    class Scaled(qn.CIFAR10Sequence):
        def __getitem__(self, idx):
            bx, by = super().__getitem__(idx)
            return bx * 0.5, by

    net = _skip_net()
    X = np.random.default_rng(3).random((12, 10, 10, 3)).astype(np.float32)
    q = qn.QuantizedCNN(network=net, batch_size=4, get_data=Scaled(X, np.zeros((12, 1), np.float32), 4), logger=_Quiet(), device="cpu")
    raw, sizes = q._raw_inputs()
    assert sizes == [4, 4, 4] and torch.equal(raw, torch.from_numpy(X * 0.5))
    plain = qn.QuantizedCNN(network=net, batch_size=4, get_data=qn.CIFAR10Sequence(X, np.zeros((12, 1), np.float32), 4), logger=_Quiet(), device="cpu")
    assert torch.equal(plain._raw_inputs()[0], torch.from_numpy(X))


def test_lazy_stats_hand_out_host_arrays_on_every_access_path():
    import pickle
    st = qn._LazyStats(rad=1.5, idx=torch.arange(6, dtype=torch.int8).reshape(2, 3), resid=torch.ones(3, dtype=torch.float64))
    assert isinstance(dict(st)["idx"], np.ndarray) and isinstance(st.copy()["resid"], np.ndarray)
    back = pickle.loads(pickle.dumps(st))
    assert isinstance(back["idx"], np.ndarray) and back["rad"] == 1.5
    assert all(isinstance(v, (float, np.ndarray)) for v in {**st}.values())
