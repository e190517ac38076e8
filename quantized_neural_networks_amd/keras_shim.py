"""A small torch-backed stand-in for the slice of ``tf.keras`` the quantizer touches.

TensorFlow is not available on the MI355X image, and the reference's drivers hand the quantizer a
Keras ``Model``.  This module provides duck-typed ``Sequential`` / ``Model`` / ``clone_model`` and
the layer types of the reference's networks (train_mnist_mlp.py, train_cifar10_cnn.py, VGG16),
executing inference with torch ops on the GPU so calibration activations never leave HBM.
``quantized_network.py`` uses real ``tf.keras`` when it is importable and this shim otherwise; it
only relies on the attribute set below, which both provide:

    network.layers, layer.__class__.__name__, layer.get_weights()/set_weights(), layer.use_bias,
    layer.inbound_nodes[0].inbound_layers, layer.input_shape, layers[0].input, layer.output,
    Model(inputs=, outputs=[...]).predict_on_batch(x), clone_model(network),
    network.get_weights()/set_weights(); conv: layer.strides, layer.padding, layer.dilation_rate.

Data format is channels-last (NHWC), weights use Keras layouts (Dense [in][out], Conv2D
[kh][kw][Cin][Cout], DepthwiseConv2D [kh][kw][Cin][mult]); this is plumbing, not the hot path.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _default_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


class _SymbolicTensor:
    """'output of layer k of network net' (k = -1: the network input)."""

    def __init__(self, net, k):
        self.net, self.k = net, k


class _Node:
    def __init__(self, inbound_layers):
        self.inbound_layers = inbound_layers


def _pair(v):
    return (int(v), int(v)) if np.isscalar(v) else (int(v[0]), int(v[1]))


def _same_pads(size, k, s, d):
    """TF 'SAME': out = ceil(size/s); total pad split floor-before / rest-after."""
    out = -(-size // s)
    keff = k + (k - 1) * (d - 1)
    total = max((out - 1) * s + keff - size, 0)
    return total // 2, total - total // 2


def _activation(name):
    if name in (None, "linear"):
        return lambda x: x
    if name == "relu":
        return torch.relu
    if name == "softmax":
        return lambda x: torch.softmax(x, dim=-1)
    if name == "sigmoid":
        return torch.sigmoid
    if name == "tanh":
        return torch.tanh
    raise ValueError(f"unsupported activation {name!r}")


class Layer:
    use_bias = False

    def __init__(self, name=None):
        self.name = name
        self.built = False
        self.inbound_nodes = []
        self.input = self.output = None
        self.input_shape = self.output_shape = None
        self._weights = []            # list of torch tensors (float32) in Keras order
        self.device = None

    # -- Keras surface ------------------------------------------------------------------
    def get_weights(self):
        return [w.detach().cpu().numpy().copy() for w in self._weights]

    def set_weights(self, weights):
        if len(weights) != len(self._weights):
            raise ValueError(f"{self.__class__.__name__} expects {len(self._weights)} arrays, got {len(weights)}")
        new = []
        for cur, w in zip(self._weights, weights):
            t = w if isinstance(w, torch.Tensor) else torch.from_numpy(np.asarray(w))
            t = t.to(device=cur.device, dtype=torch.float32)      # Keras casts to the variable dtype
            if tuple(t.shape) != tuple(cur.shape):
                raise ValueError(f"shape mismatch {tuple(t.shape)} vs {tuple(cur.shape)}")
            new.append(t.contiguous())
        self._weights = new

    # -- shim internals -----------------------------------------------------------------
    def build(self, input_shape, device, rng):
        self.input_shape = input_shape
        self.device = device
        self.output_shape = self.compute_output_shape(input_shape)
        self.built = True

    def compute_output_shape(self, s):
        return s

    def call(self, x):
        return x

    def config(self):
        return {}

    def clone(self):
        return self.__class__(**self.config())


class InputLayer(Layer):
    def __init__(self, input_shape=None, name=None):
        super().__init__(name)
        self._decl = tuple(input_shape) if input_shape is not None else None

    def config(self):
        return dict(input_shape=self._decl)


class Dense(Layer):
    def __init__(self, units, activation=None, use_bias=True, input_shape=None, name=None):
        super().__init__(name)
        self.units, self.activation, self.use_bias = int(units), activation, bool(use_bias)
        self._decl = tuple(input_shape) if input_shape is not None else None

    def config(self):
        return dict(units=self.units, activation=self.activation, use_bias=self.use_bias, input_shape=self._decl)

    def build(self, input_shape, device, rng):
        super().build(input_shape, device, rng)
        fan_in = int(input_shape[-1])
        lim = math.sqrt(6.0 / (fan_in + self.units))                      # glorot_uniform
        k = torch.from_numpy(rng.uniform(-lim, lim, (fan_in, self.units)).astype(np.float32)).to(device)
        self._weights = [k] + ([torch.zeros(self.units, device=device)] if self.use_bias else [])

    def compute_output_shape(self, s):
        return tuple(s[:-1]) + (self.units,)

    def call(self, x):
        y = x @ self._weights[0]
        if self.use_bias:
            y = y + self._weights[1]
        return _activation(self.activation)(y)


class Conv2D(Layer):
    def __init__(self, filters, kernel_size, strides=(1, 1), padding="valid", dilation_rate=(1, 1),
                 activation=None, use_bias=True, input_shape=None, name=None):
        super().__init__(name)
        self.filters = int(filters)
        self.kernel_size, self.strides, self.dilation_rate = _pair(kernel_size), _pair(strides), _pair(dilation_rate)
        self.padding, self.activation, self.use_bias = padding, activation, bool(use_bias)
        self._decl = tuple(input_shape) if input_shape is not None else None

    def config(self):
        return dict(filters=self.filters, kernel_size=self.kernel_size, strides=self.strides, padding=self.padding,
                    dilation_rate=self.dilation_rate, activation=self.activation, use_bias=self.use_bias,
                    input_shape=self._decl)

    def _kernel_shape(self, cin):
        return self.kernel_size + (cin, self.filters)

    def build(self, input_shape, device, rng):
        super().build(input_shape, device, rng)
        cin = int(input_shape[-1])
        shape = self._kernel_shape(cin)
        fan_in = shape[0] * shape[1] * cin
        fan_out = shape[0] * shape[1] * shape[3]
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        k = torch.from_numpy(rng.uniform(-lim, lim, shape).astype(np.float32)).to(device)
        nb = self._out_channels(cin)
        self._weights = [k] + ([torch.zeros(nb, device=device)] if self.use_bias else [])

    def _out_channels(self, cin):
        return self.filters

    def _spatial(self, size, axis):
        k, s, d = self.kernel_size[axis], self.strides[axis], self.dilation_rate[axis]
        if self.padding.lower() == "same":
            return -(-size // s)
        keff = k + (k - 1) * (d - 1)
        return max(-(-(size - keff + 1) // s), 0)

    def compute_output_shape(self, s):
        return (s[0], self._spatial(s[1], 0), self._spatial(s[2], 1), self._out_channels(int(s[3])))

    def _torch_weight(self):
        return self._weights[0].permute(3, 2, 0, 1), 1                      # OIHW, groups

    def call(self, x):
        x = x.permute(0, 3, 1, 2)                                           # NHWC -> NCHW (a channels-last view)
        pad = (0, 0)
        if self.padding.lower() == "same":
            pt, pb = _same_pads(x.shape[2], self.kernel_size[0], self.strides[0], self.dilation_rate[0])
            pl, pr = _same_pads(x.shape[3], self.kernel_size[1], self.strides[1], self.dilation_rate[1])
            if pt == pb and pl == pr:
                pad = (pt, pl)                                              # symmetric: the conv pads, no padded copy
            else:
                x = F.pad(x, (pl, pr, pt, pb))
        w, groups = self._torch_weight()
        y = F.conv2d(x, w.contiguous(), self._weights[1] if self.use_bias else None, stride=self.strides,
                     padding=pad, dilation=self.dilation_rate, groups=groups)
        return _activation(self.activation)(y.permute(0, 2, 3, 1).contiguous())


class DepthwiseConv2D(Conv2D):
    def __init__(self, kernel_size, strides=(1, 1), padding="valid", depth_multiplier=1, dilation_rate=(1, 1),
                 activation=None, use_bias=True, input_shape=None, name=None):
        super().__init__(0, kernel_size, strides, padding, dilation_rate, activation, use_bias, input_shape, name)
        self.depth_multiplier = int(depth_multiplier)

    def config(self):
        c = super().config()
        c.pop("filters")
        c["depth_multiplier"] = self.depth_multiplier
        return c

    def _kernel_shape(self, cin):
        return self.kernel_size + (cin, self.depth_multiplier)

    def _out_channels(self, cin):
        return cin * self.depth_multiplier

    def _torch_weight(self):
        kh, kw, cin, mult = self._weights[0].shape
        # Keras output channel order is c*mult + j, which is torch's grouped layout with groups = cin
        w = self._weights[0].permute(2, 3, 0, 1).reshape(cin * mult, 1, kh, kw)
        return w, cin


class Flatten(Layer):
    def __init__(self, input_shape=None, name=None):
        super().__init__(name)
        self._decl = tuple(input_shape) if input_shape is not None else None

    def config(self):
        return dict(input_shape=self._decl)

    def compute_output_shape(self, s):
        return (s[0], int(np.prod(s[1:])))

    def call(self, x):
        return x.reshape(x.shape[0], -1)


class _Pool2D(Layer):
    def __init__(self, pool_size=(2, 2), strides=None, padding="valid", name=None):
        super().__init__(name)
        self.pool_size = _pair(pool_size)
        self.strides = _pair(strides) if strides is not None else self.pool_size
        self.padding = padding

    def config(self):
        return dict(pool_size=self.pool_size, strides=self.strides, padding=self.padding)

    def compute_output_shape(self, s):
        def o(size, k, st):
            return -(-size // st) if self.padding.lower() == "same" else (size - k) // st + 1
        return (s[0], o(s[1], self.pool_size[0], self.strides[0]), o(s[2], self.pool_size[1], self.strides[1]), s[3])


class MaxPooling2D(_Pool2D):
    def call(self, x):
        x = x.permute(0, 3, 1, 2)
        if self.padding.lower() == "same":
            pt, pb = _same_pads(x.shape[2], self.pool_size[0], self.strides[0], 1)
            pl, pr = _same_pads(x.shape[3], self.pool_size[1], self.strides[1], 1)
            x = F.pad(x, (pl, pr, pt, pb), value=float("-inf"))
        return F.max_pool2d(x, self.pool_size, self.strides).permute(0, 2, 3, 1).contiguous()


class AveragePooling2D(_Pool2D):
    def call(self, x):
        if self.padding.lower() == "same":
            raise NotImplementedError("AveragePooling2D(padding='same') is not provided by the shim")
        return F.avg_pool2d(x.permute(0, 3, 1, 2), self.pool_size, self.strides).permute(0, 2, 3, 1).contiguous()


class ZeroPadding2D(Layer):
    def __init__(self, padding=(1, 1), name=None):
        super().__init__(name)
        p = padding
        if np.isscalar(p):
            p = ((p, p), (p, p))
        elif np.isscalar(p[0]):
            p = ((p[0], p[0]), (p[1], p[1]))
        self.padding = tuple(tuple(int(v) for v in q) for q in p)

    def config(self):
        return dict(padding=self.padding)

    def compute_output_shape(self, s):
        return (s[0], s[1] + sum(self.padding[0]), s[2] + sum(self.padding[1]), s[3])

    def call(self, x):
        (t, b), (l, r) = self.padding
        return F.pad(x, (0, 0, l, r, t, b))


class BatchNormalization(Layer):
    """Inference mode only (moving statistics), which is what predict_on_batch uses."""

    def __init__(self, epsilon=1e-3, name=None):
        super().__init__(name)
        self.epsilon = float(epsilon)

    def config(self):
        return dict(epsilon=self.epsilon)

    def build(self, input_shape, device, rng):
        super().build(input_shape, device, rng)
        c = int(input_shape[-1])
        self._weights = [torch.ones(c, device=device), torch.zeros(c, device=device),
                         torch.zeros(c, device=device), torch.ones(c, device=device)]   # gamma, beta, mean, var

    def call(self, x):
        g, b, mu, var = self._weights
        return (x - mu) * (g / torch.sqrt(var + self.epsilon)) + b


class Activation(Layer):
    def __init__(self, activation, name=None):
        super().__init__(name)
        self.activation = activation

    def config(self):
        return dict(activation=self.activation)

    def call(self, x):
        return _activation(self.activation)(x)


class ReLU(Layer):
    def call(self, x):
        return torch.relu(x)


class Dropout(Layer):
    def __init__(self, rate=0.5, name=None):
        super().__init__(name)
        self.rate = rate

    def config(self):
        return dict(rate=self.rate)


class Sequential:
    """Keras-like Sequential: ``network.layers`` lists exactly the added layers (an explicit
    InputLayer is not listed, as in Keras)."""

    def __init__(self, layers=None, input_shape=None, device=None, seed=0):
        self.layers = []
        self._input_shape = tuple(input_shape) if input_shape is not None else None
        self.device = torch.device(device) if device is not None else _default_device()
        self._rng = np.random.default_rng(seed)
        self.built = False
        for layer in layers or []:
            self.add(layer)

    def add(self, layer):
        if isinstance(layer, InputLayer):
            self._input_shape = layer._decl
            return
        if not self.layers and self._input_shape is None:
            decl = getattr(layer, "_decl", None)
            if decl is None:
                raise ValueError("the first layer needs input_shape= (or pass input_shape to Sequential)")
            self._input_shape = decl
        prev_shape = self.layers[-1].output_shape if self.layers else (None,) + tuple(self._input_shape)
        k = len(self.layers)
        layer.build(prev_shape, self.device, self._rng)
        layer.input = _SymbolicTensor(self, k - 1)
        layer.output = _SymbolicTensor(self, k)
        layer.inbound_nodes = [_Node(self.layers[-1] if self.layers else [])]
        if layer.name is None:
            layer.name = f"{layer.__class__.__name__.lower()}_{k}"
        self.layers.append(layer)
        self.built = True

    @property
    def input_shape(self):
        return (None,) + tuple(self._input_shape)

    @property
    def input(self):
        return self.layers[0].input

    def get_weights(self):
        out = []
        for layer in self.layers:
            out += layer.get_weights()
        return out

    def set_weights(self, weights):
        i = 0
        for layer in self.layers:
            n = len(layer._weights)
            layer.set_weights(weights[i:i + n])
            i += n
        if i != len(weights):
            raise ValueError("weight list length mismatch")

    def _as_tensor(self, x):
        if isinstance(x, torch.Tensor):
            return x.to(device=self.device, dtype=torch.float32)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).to(self.device)

    @torch.no_grad()
    def forward_upto(self, x, k):
        x = self._as_tensor(x)
        for layer in self.layers[:k + 1]:
            x = layer.call(x)
        return x

    def predict_on_batch(self, x):
        return self.forward_upto(x, len(self.layers) - 1)

    def predict(self, x, batch_size=32, verbose=0):
        outs = [self.predict_on_batch(x[i:i + batch_size]) for i in range(0, len(x), batch_size)]
        return torch.cat(outs).cpu().numpy()

    def compile(self, *args, **kwargs):
        return None

    def evaluate(self, x, y, batch_size=256, verbose=0):
        """(loss, accuracy) with categorical cross-entropy on one-hot labels, like the drivers' use."""
        correct, loss, n = 0, 0.0, len(x)
        for i in range(0, n, batch_size):
            p = self.predict_on_batch(x[i:i + batch_size]).double()
            t = torch.from_numpy(np.asarray(y[i:i + batch_size])).to(p.device).double()
            loss += float(-(t * torch.log(p.clamp_min(1e-12))).sum())
            correct += int((p.argmax(dim=1) == t.argmax(dim=1)).sum())
        return loss / max(n, 1), correct / max(n, 1)


class Model:
    """``Model(inputs=net.layers[0].input, outputs=[layer.output, ...])``: a truncated view."""

    def __init__(self, inputs=None, outputs=None):
        self.inputs = inputs
        self._single = not isinstance(outputs, (list, tuple))
        self.outputs = [outputs] if self._single else list(outputs)

    def predict_on_batch(self, x):
        res = [t.net.forward_upto(x, t.k) for t in self.outputs]
        return res[0] if (self._single or len(res) == 1) else res


def clone_model(net):
    """Same architecture, freshly initialised weights (Keras semantics); callers copy weights over."""
    clone = Sequential(input_shape=net._input_shape, device=net.device, seed=12345)
    for layer in net.layers:
        clone.add(layer.clone())
    return clone


def save_model(model, filepath):
    """Stand-in for ``tf.keras.models.save_model`` as the reference's drivers call it on ``quantized_net``
    (quantize_pretrained_mlp.py:87-95, _imagenet.py:180-191): architecture (layer classes + configs) and weights
    in ONE ``.npz`` file (no pickling; ``load_model`` rebuilds the network on the current device)."""
    import json
    arch = dict(input_shape=list(model._input_shape), layers=[dict(cls=l.__class__.__name__, name=l.name, config=l.config())
                                                              for l in model.layers])
    arrays = {"__arch__": np.frombuffer(json.dumps(arch).encode("utf-8"), dtype=np.uint8)}
    for k, layer in enumerate(model.layers):
        for j, w in enumerate(layer.get_weights()):
            arrays[f"w{k}_{j}"] = w
    path = str(filepath)
    with open(path if path.endswith(".npz") else path + ".npz", "wb") as f:
        np.savez(f, **arrays)


def load_model(filepath, device=None):
    """Inverse of ``save_model``."""
    import json
    path = str(filepath)
    with np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False) as z:
        arch = json.loads(bytes(z["__arch__"]).decode("utf-8"))
        known = {c.__name__: c for c in (Dense, Conv2D, DepthwiseConv2D, Flatten, MaxPooling2D, AveragePooling2D,
                                         ZeroPadding2D, BatchNormalization, Activation, ReLU, Dropout)}
        net = Sequential(input_shape=tuple(arch["input_shape"]), device=device)
        for k, spec in enumerate(arch["layers"]):
            if spec["cls"] not in known:
                raise ValueError(f"load_model: unknown layer class {spec['cls']!r}")
            def tup(v):
                return tuple(tup(e) for e in v) if isinstance(v, list) else v
            cfg = {key: tup(v) for key, v in spec["config"].items()}
            layer = known[spec["cls"]](**cfg)
            layer.name = spec["name"]
            net.add(layer)
            n = len(layer._weights)
            if n:
                layer.set_weights([z[f"w{k}_{j}"] for j in range(n)])
    return net
