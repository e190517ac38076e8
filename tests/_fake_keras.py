"""A NumPy duck-type of the slice of tf.keras the quantizer touches -- test helper only.

What a real ``tf.keras`` model hands the quantizer: layers whose ``get_weights()`` return NumPy arrays and whose
``set_weights()`` take them, truncated ``Model(inputs=, outputs=[...])`` objects with ``predict_on_batch`` returning
NumPy arrays, ``clone_model``.  The tests bind ``Model`` / ``clone_model`` of the quantizer module to these (exactly
what ``from tensorflow.keras.models import Model, clone_model`` would do on a box with TensorFlow), so the host-side
branches -- kernel upload from ``get_weights()``, ``set_weights([Q as ndarray, bias])``, activations arriving as
NumPy arrays -- run against the HIP path.  No ``forward_upto`` / ``_weights``: the quantizer must not mistake it for
its own torch-backed shim.
"""
import numpy as np


class _Tensor:
    """Symbolic handle: 'output of layer k' (k = -1 is the network input)."""

    def __init__(self, net, k):
        self.net, self.k = net, k


class _Node:
    def __init__(self, inbound_layers):
        self.inbound_layers = inbound_layers


class Dense:
    def __init__(self, kernel, bias=None, activation="linear"):
        self.kernel = np.asarray(kernel, dtype=np.float32)
        self.bias = None if bias is None else np.asarray(bias, dtype=np.float32)
        self.use_bias = bias is not None
        self.activation = activation
        self.input_shape = (None, self.kernel.shape[0])
        self.inbound_nodes = []
        self.input = self.output = None
        self.set_calls = []                       # what set_weights was handed (types are asserted by the tests)

    def get_weights(self):
        return [self.kernel.copy()] + ([self.bias.copy()] if self.use_bias else [])

    def set_weights(self, ws):
        self.set_calls.append([type(w) for w in ws])
        self.kernel = np.asarray(ws[0]).astype(np.float32)          # Keras casts to the variable dtype
        if self.use_bias:
            self.bias = np.asarray(ws[1]).astype(np.float32)

    def call(self, x):
        y = x.astype(np.float32) @ self.kernel
        if self.use_bias:
            y = y + self.bias
        if self.activation == "relu":
            y = np.maximum(y, np.float32(0))
        return y.astype(np.float32)

    def _clone(self):
        return Dense(self.kernel.copy(), None if self.bias is None else self.bias.copy(), self.activation)


class Sequential:
    def __init__(self, layers):
        self.layers = list(layers)
        prev = None
        for k, layer in enumerate(self.layers):
            layer.input = _Tensor(self, k - 1)
            layer.output = _Tensor(self, k)
            layer.inbound_nodes = [_Node(prev)] if prev is not None else [_Node([])]
            prev = layer

    def get_weights(self):
        out = []
        for layer in self.layers:
            out += layer.get_weights()
        return out

    def set_weights(self, ws):
        i = 0
        for layer in self.layers:
            n = 2 if layer.use_bias else 1
            layer.set_weights(ws[i:i + n])
            i += n

    def _run(self, x, k):
        x = np.asarray(x, dtype=np.float32)
        for layer in self.layers[:k + 1]:
            x = layer.call(x)
        return x


class Model:
    """Model(inputs=<net input>, outputs=[layer.output, ...]) -> truncated network."""

    def __init__(self, inputs=None, outputs=None):
        self.inputs, self.outputs = inputs, outputs

    def predict_on_batch(self, x):
        res = [t.net._run(x, t.k) for t in self.outputs]
        return res[0] if len(res) == 1 else res


def clone_model(net):
    """Keras clone_model re-initialises the weights; callers copy them over."""
    clone = Sequential([l._clone() for l in net.layers])
    for l in clone.layers:
        l.kernel = np.zeros_like(l.kernel)
    return clone
