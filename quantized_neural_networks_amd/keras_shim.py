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


class _FTensor:
    """Functional-API symbolic tensor: 'the output of `layer`' (Keras' KerasTensor), with its static shape."""

    def __init__(self, layer, shape):
        self.layer, self.shape = layer, tuple(shape)


_creation_counter = [0]


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
        c = self.__class__(**self.config())
        c.name = self.name
        return c

    # -- functional API: layer(x) / layer([a, b]) -----------------------------------------
    def __call__(self, inputs):
        """Connect the layer to symbolic tensor(s); weights are created when a Model is built over the graph."""
        multi = isinstance(inputs, (list, tuple))
        tensors = list(inputs) if multi else [inputs]
        if not all(isinstance(t, _FTensor) for t in tensors):
            raise TypeError("functional API: call layers on tensors made by Input() / other layers")
        if self.inbound_nodes:
            raise NotImplementedError("the shim connects every layer once (no shared layers)")
        shapes = [t.shape for t in tensors]
        self.input_shape = shapes if multi else shapes[0]
        self.output_shape = self.compute_output_shape(self.input_shape)
        self.input = tensors if multi else tensors[0]
        self.output = _FTensor(self, self.output_shape)
        inbound = [t.layer for t in tensors]
        self.inbound_nodes = [_Node(inbound if multi else inbound[0])]       # Keras: the layer itself when there is one
        _creation_counter[0] += 1
        self._order = _creation_counter[0]
        return self.output


class InputLayer(Layer):
    def __init__(self, input_shape=None, name=None):
        super().__init__(name)
        self._decl = tuple(input_shape) if input_shape is not None else None

    def config(self):
        return dict(input_shape=self._decl)


def Input(shape, name=None):
    """Functional API entry point: a symbolic input of shape (None,) + shape."""
    layer = InputLayer(input_shape=tuple(shape), name=name or "input_1")
    layer.input_shape = layer.output_shape = (None,) + tuple(shape)
    layer.inbound_nodes = [_Node([])]
    _creation_counter[0] += 1
    layer._order = _creation_counter[0]
    layer.input = layer.output = _FTensor(layer, layer.output_shape)
    return layer.output


class Add(Layer):
    """Element-wise sum of its inputs (ResNet shortcuts)."""

    def compute_output_shape(self, s):
        return tuple(s[0])

    def call(self, xs):
        y = xs[0]
        for x in xs[1:]:
            y = y + x
        return y


class GlobalAveragePooling2D(Layer):
    def compute_output_shape(self, s):
        return (s[0], s[3])

    def call(self, x):
        return x.mean(dim=(1, 2))


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
        # (the feature count spelled out: a rank's block of samples may be EMPTY -- six capture chunks over eight ranks -- and
        #  reshape(0, -1) is ambiguous)
        return x.reshape(x.shape[0], int(np.prod(x.shape[1:])))


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
        # TensorFlow's own inference form (tf.nn.batch_normalization: inv = rsqrt(variance + epsilon) * scale;
        # x * inv + (offset - mean * inv)): per-channel scale and shift, ONE pass over the activations (torch.addcmul) instead of
        # the three of (x - mean) * inv + offset -- 5 ms of the CIFAR10 CNN's 45 ms capture-and-quantize run went there
        g, b, mu, var = self._weights
        inv = g / torch.sqrt(var + self.epsilon)
        return torch.addcmul(b - mu * inv, x, inv)


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
    """``Model(inputs, outputs)`` over symbolic tensors, as the reference uses it:

    * tensors of a ``Sequential`` (``net.layers[0].input`` / ``layer.output``): a truncated view of that network
      (scripts/quantized_network.py:456-462);
    * tensors of the functional API (``Input(...)``, ``layer(x)``, ``Add()([a, b])``): the graph network whose
      ``layers`` lists, in creation order, every layer between ``inputs`` and ``outputs`` (``InputLayer`` first, as
      Keras does).  Weights are created the first time a Model is built over unbuilt layers; a second Model over the
      same tensors (the truncated models of the activation capture) shares them.
    """

    def __init__(self, inputs=None, outputs=None, device=None, seed=0, name=None):
        self.inputs = inputs
        self._single = not isinstance(outputs, (list, tuple))
        self.outputs = [outputs] if self._single else list(outputs)
        self.name = name
        self._functional = all(isinstance(t, _FTensor) for t in self.outputs)
        if not self._functional:
            return
        ins = list(inputs) if isinstance(inputs, (list, tuple)) else [inputs]
        if len(ins) != 1 or not isinstance(ins[0], _FTensor):
            raise NotImplementedError("functional Model: one input tensor")
        self._input = ins[0]
        # every layer the outputs depend on, in creation (= topological) order
        seen, stack = {}, [t.layer for t in self.outputs]
        while stack:
            layer = stack.pop()
            if id(layer) in seen:
                continue
            seen[id(layer)] = layer
            inbound = layer.inbound_nodes[0].inbound_layers
            stack.extend(inbound if isinstance(inbound, (list, tuple)) else [inbound])
        self.layers = sorted(seen.values(), key=lambda l: l._order)
        if self._input.layer not in self.layers:
            raise ValueError("functional Model: outputs do not depend on the input")
        self.device = torch.device(device) if device is not None else next(
            (l.device for l in self.layers if l.built and l.device is not None), _default_device())
        rng = np.random.default_rng(seed)
        taken = {l.name for l in self.layers if l.name}
        for k, layer in enumerate(self.layers):
            if not layer.built:
                layer.build(layer.input_shape, self.device, rng)
            if layer.name is None:
                base = f"{layer.__class__.__name__.lower()}_{k}"
                while base in taken:
                    base += "_"
                layer.name = base
                taken.add(base)
        self._input_shape = tuple(self._input.shape[1:])
        self.built = True

    def graph_tables(self):
        """(inbound, last_use) of a graph network, by position in ``self.layers`` (creation = topological order): ``inbound[k]`` lists
        the positions of the layers whose outputs layer k consumes (empty for the InputLayer), ``last_use[k]`` is the position of
        the last layer that consumes layer k's output (``len(self.layers)`` for a model output: never released).  What an
        incremental walk over the graph needs to keep exactly the live tensors (quantized_network._capture_incremental_graph)."""
        if not self._functional:
            raise NotImplementedError("graph_tables of a truncated view")
        tabs = getattr(self, "_tables", None)
        if tabs is None:
            pos = {id(l): k for k, l in enumerate(self.layers)}
            inbound, last = [], list(range(len(self.layers)))
            for k, layer in enumerate(self.layers):
                inb = layer.inbound_nodes[0].inbound_layers
                idx = [pos[id(p)] for p in (inb if isinstance(inb, (list, tuple)) else [inb])]
                inbound.append(idx)
                for p in idx:
                    last[p] = max(last[p], k)
            for t in self.outputs:
                last[pos[id(t.layer)]] = len(self.layers)
            tabs = self._tables = (inbound, last)
        return tabs

    # -- Keras surface of a full model ---------------------------------------------------
    @property
    def input(self):
        return self._input if self._functional else self.inputs

    @property
    def input_shape(self):
        return (None,) + tuple(self._input_shape)

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
    def _run_graph(self, x):
        """Values of self.outputs for input batch x: one pass over the layers in order, tensors freed after their
        last consumer."""
        vals = {id(self._input.layer): self._as_tensor(x)}
        last_use = {}
        for k, layer in enumerate(self.layers):
            inbound = layer.inbound_nodes[0].inbound_layers
            for p in (inbound if isinstance(inbound, (list, tuple)) else [inbound]):
                last_use[id(p)] = k
        keep = {id(t.layer) for t in self.outputs}
        for k, layer in enumerate(self.layers):
            if id(layer) in vals:
                continue
            inbound = layer.inbound_nodes[0].inbound_layers
            if isinstance(inbound, (list, tuple)):
                vals[id(layer)] = layer.call([vals[id(p)] for p in inbound])
            else:
                vals[id(layer)] = layer.call(vals[id(inbound)])
            for p in (inbound if isinstance(inbound, (list, tuple)) else [inbound]):
                if last_use.get(id(p)) == k and id(p) not in keep:
                    del vals[id(p)]
        return [vals[id(t.layer)] for t in self.outputs]

    def predict_on_batch(self, x):
        if self._functional:
            res = self._run_graph(x)
        else:
            res = [t.net.forward_upto(x, t.k) for t in self.outputs]
        return res[0] if (self._single or len(res) == 1) else res

    def predict(self, x, batch_size=32, verbose=0):
        outs = [self.predict_on_batch(x[i:i + batch_size]) for i in range(0, len(x), batch_size)]
        return torch.cat(outs).cpu().numpy()

    def compile(self, *args, **kwargs):
        return None

    def evaluate(self, x, y, batch_size=256, verbose=0):
        return Sequential.evaluate(self, x, y, batch_size, verbose)


def clone_model(net):
    """Same architecture, freshly initialised weights (Keras semantics); callers copy weights over."""
    if isinstance(net, Model):
        if not net._functional:
            raise NotImplementedError("clone_model of a truncated view")
        mapping = {}
        for layer in net.layers:
            if isinstance(layer, InputLayer):
                mapping[id(layer)] = Input(layer._decl, name=layer.name)
                continue
            inbound = layer.inbound_nodes[0].inbound_layers
            new = layer.clone()
            if isinstance(inbound, (list, tuple)):
                mapping[id(layer)] = new([mapping[id(p)] for p in inbound])
            else:
                mapping[id(layer)] = new(mapping[id(inbound)])
        outs = [mapping[id(t.layer)] for t in net.outputs]
        return Model(inputs=mapping[id(net._input.layer)], outputs=outs[0] if net._single else outs, device=net.device,
                     seed=12345, name=net.name)
    clone = Sequential(input_shape=net._input_shape, device=net.device, seed=12345)
    for layer in net.layers:
        clone.add(layer.clone())
    return clone


def ResNet50(input_shape=(224, 224, 3), classes=1000, device=None, seed=0):
    """The topology of ``tf.keras.applications.ResNet50`` (v1, channels-last; the model quantize_pretrained_imagenet.py:10
    imports): conv1 7x7/2 on the zero-padded input, 3x3/2 max-pool, stages of 3 / 4 / 6 / 3 bottleneck blocks
    (1x1, 3x3 'same', 1x1 convolutions, BatchNormalization after each, projection shortcut in every stage's first block,
    stride 2 on the first 1x1 and on the shortcut of stages 3-5), global average pooling, softmax classifier -- 53
    Conv2D layers and one Dense, with Keras' layer names.  Weights are random (no checkpoints on this image)."""
    def block(x, filters, stride, conv_shortcut, name):
        if conv_shortcut:
            sc = Conv2D(4 * filters, 1, strides=stride, name=name + "_0_conv")(x)
            sc = BatchNormalization(epsilon=1.001e-5, name=name + "_0_bn")(sc)
        else:
            sc = x
        y = Conv2D(filters, 1, strides=stride, name=name + "_1_conv")(x)
        y = BatchNormalization(epsilon=1.001e-5, name=name + "_1_bn")(y)
        y = Activation("relu", name=name + "_1_relu")(y)
        y = Conv2D(filters, 3, padding="same", name=name + "_2_conv")(y)
        y = BatchNormalization(epsilon=1.001e-5, name=name + "_2_bn")(y)
        y = Activation("relu", name=name + "_2_relu")(y)
        y = Conv2D(4 * filters, 1, name=name + "_3_conv")(y)
        y = BatchNormalization(epsilon=1.001e-5, name=name + "_3_bn")(y)
        y = Add(name=name + "_add")([sc, y])
        return Activation("relu", name=name + "_out")(y)

    def stack(x, filters, blocks, stride1, name):
        x = block(x, filters, stride1, True, name + "_block1")
        for i in range(2, blocks + 1):
            x = block(x, filters, 1, False, f"{name}_block{i}")
        return x

    inp = Input(input_shape, name="input_1")
    x = ZeroPadding2D(3, name="conv1_pad")(inp)
    x = Conv2D(64, 7, strides=2, name="conv1_conv")(x)
    x = BatchNormalization(epsilon=1.001e-5, name="conv1_bn")(x)
    x = Activation("relu", name="conv1_relu")(x)
    x = ZeroPadding2D(1, name="pool1_pad")(x)
    x = MaxPooling2D(3, strides=2, name="pool1_pool")(x)
    x = stack(x, 64, 3, 1, "conv2")
    x = stack(x, 128, 4, 2, "conv3")
    x = stack(x, 256, 6, 2, "conv4")
    x = stack(x, 512, 3, 2, "conv5")
    x = GlobalAveragePooling2D(name="avg_pool")(x)
    out = Dense(classes, activation="softmax", name="predictions")(x)
    return Model(inp, out, device=device, seed=seed, name="resnet50")


_LAYER_CLASSES = None


def _layer_classes():
    global _LAYER_CLASSES
    if _LAYER_CLASSES is None:
        _LAYER_CLASSES = {c.__name__: c for c in (Dense, Conv2D, DepthwiseConv2D, Flatten, MaxPooling2D, AveragePooling2D,
                                                  ZeroPadding2D, BatchNormalization, Activation, ReLU, Dropout, Add,
                                                  GlobalAveragePooling2D)}
    return _LAYER_CLASSES


def save_model(model, filepath):
    """Stand-in for ``tf.keras.models.save_model`` as the reference's drivers call it on ``quantized_net``
    (quantize_pretrained_mlp.py:87-95, _imagenet.py:180-191): architecture (layer classes + configs, and for graph
    networks every layer's inbound layers) and weights in ONE ``.npz`` file (no pickling; ``load_model`` rebuilds the
    network on the current device)."""
    import json
    functional = isinstance(model, Model)
    specs = []
    for l in model.layers:
        spec = dict(cls=l.__class__.__name__, name=l.name, config=l.config())
        if functional:
            inbound = l.inbound_nodes[0].inbound_layers
            spec["inbound"] = [p.name for p in inbound] if isinstance(inbound, (list, tuple)) else [inbound.name]
            spec["multi"] = isinstance(inbound, (list, tuple)) and not isinstance(l, InputLayer)
        specs.append(spec)
    arch = dict(input_shape=list(model._input_shape), layers=specs, functional=functional)
    if functional:
        arch["outputs"] = [t.layer.name for t in model.outputs]
        arch["single"] = model._single
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

    def tup(v):
        return tuple(tup(e) for e in v) if isinstance(v, list) else v

    with np.load(path if path.endswith(".npz") else path + ".npz", allow_pickle=False) as z:
        arch = json.loads(bytes(z["__arch__"]).decode("utf-8"))
        known = _layer_classes()
        if arch.get("functional"):
            tensors, layers = {}, []
            for spec in arch["layers"]:
                cfg = {key: tup(v) for key, v in spec["config"].items()}
                if spec["cls"] == "InputLayer":
                    tensors[spec["name"]] = Input(cfg["input_shape"], name=spec["name"])
                    layers.append(tensors[spec["name"]].layer)
                    continue
                if spec["cls"] not in known:
                    raise ValueError(f"load_model: unknown layer class {spec['cls']!r}")
                layer = known[spec["cls"]](**cfg)
                layer.name = spec["name"]
                ins = [tensors[n] for n in spec["inbound"]]
                tensors[spec["name"]] = layer(ins if spec["multi"] else ins[0])
                layers.append(layer)
            outs = [tensors[n] for n in arch["outputs"]]
            net = Model(layers[0].output, outs[0] if arch["single"] else outs, device=device)
            assert [l.name for l in net.layers] == [s["name"] for s in arch["layers"]]
            for k, layer in enumerate(net.layers):
                n = len(layer._weights)
                if n:
                    layer.set_weights([z[f"w{k}_{j}"] for j in range(n)])
            return net
        net = Sequential(input_shape=tuple(arch["input_shape"]), device=device)
        for k, spec in enumerate(arch["layers"]):
            if spec["cls"] not in known:
                raise ValueError(f"load_model: unknown layer class {spec['cls']!r}")
            cfg = {key: tup(v) for key, v in spec["config"].items()}
            layer = known[spec["cls"]](**cfg)
            layer.name = spec["name"]
            net.add(layer)
            n = len(layer._weights)
            if n:
                layer.set_weights([z[f"w{k}_{j}"] for j in range(n)])
    return net
