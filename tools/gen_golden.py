#!/opt/conda/bin/python3.9
"""Generate golden input/output vectors by RUNNING the reference implementation.

Run in the build container only (the reference never travels to the GPU box):

    /opt/conda/bin/python3.9 tools/gen_golden.py

It needs the legacy-casting interpreter (numpy 1.26 / scipy 1.7 / h5py 3.3) that
matches the reference's era.  The reference module
/root/reference/scripts/quantized_network.py is loaded *by path* with the
``tensorflow`` package stubbed out (its hot path never calls TensorFlow), and its
own functions are executed on seeded inputs:

  * ``_bit_round_parallel``                    (reference :40-57)
  * ``_quantize_neuron_parallel``              (reference :91-121, real h5py files)
  * ``_quantize_filter2D_parallel_jit``        (reference :185-233, real h5py files)
  * ``QuantizedNeuralNetwork.quantize_network`` (reference :576-590) driven through a
    duck-typed Dense-only fake Keras, which pins rad/alphabet, the transposed
    activation layout, the partial-last-batch quirk and bias carry-over.

Only data (inputs + expected outputs) is written, to tests/golden/*.npz.
"""
import importlib.util
import os
import sys
import tempfile
import types

import numpy as np

assert np.__version__.startswith("1."), "run me with the legacy-casting oracle interpreter"

REF = "/root/reference/scripts/quantized_network.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# --------------------------------------------------------------------------------------
# Fake Keras: just enough surface for the reference's Dense orchestration.
# --------------------------------------------------------------------------------------
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
        self.input = None
        self.output = None

    def get_weights(self):
        return [self.kernel.copy()] + ([self.bias.copy()] if self.use_bias else [])

    def set_weights(self, ws):
        # Keras casts whatever it is given to the variable dtype (float32).
        self.kernel = np.asarray(ws[0]).astype(np.float32)
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

    def forward_upto(self, x, k):
        x = np.asarray(x, dtype=np.float32)
        for layer in self.layers[:k + 1]:
            x = layer.call(x)
        return x


class Model:
    """Model(inputs=<net input>, outputs=[layer.output, ...]) -> truncated network."""

    def __init__(self, inputs=None, outputs=None):
        self.inputs, self.outputs = inputs, outputs

    def predict_on_batch(self, x):
        res = [t.net.forward_upto(x, t.k) for t in self.outputs]
        return res[0] if len(res) == 1 else res


def clone_model(net):
    # Keras clone_model re-initialises weights; the reference overwrites them right after.
    clone = Sequential([l._clone() for l in net.layers])
    for l in clone.layers:
        l.kernel = np.zeros_like(l.kernel)
    return clone


class Sequence:
    pass


def _load_reference():
    tf = types.ModuleType("tensorflow")
    tf.convert_to_tensor = lambda x: x
    keras = types.ModuleType("tensorflow.keras")
    utils = types.ModuleType("tensorflow.keras.utils")
    utils.Sequence = Sequence
    backend = types.ModuleType("tensorflow.keras.backend")
    backend.function = lambda *a, **k: None
    models = types.ModuleType("tensorflow.keras.models")
    models.Model = Model
    models.clone_model = clone_model
    image = types.ModuleType("tensorflow.image")
    image.extract_patches = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError("TF absent"))
    for name, mod in [("tensorflow", tf), ("tensorflow.keras", keras), ("tensorflow.keras.utils", utils),
                      ("tensorflow.keras.backend", backend), ("tensorflow.keras.models", models),
                      ("tensorflow.image", image)]:
        sys.modules[name] = mod
    spec = importlib.util.spec_from_file_location("ref_quantized_network", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_quantized_network"] = mod
    spec.loader.exec_module(mod)
    return mod


ref = _load_reference()
import h5py  # noqa: E402  (after the reference import so failures point at the right thing)


# --------------------------------------------------------------------------------------
# Input generators
# --------------------------------------------------------------------------------------
def relu_like(seed, N, m, noise=0.1, first_layer=False):
    r = np.random.default_rng(seed)
    G = r.standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    if first_layer:
        return X, X.copy()
    Xq = np.maximum(G + noise * r.standard_normal((N, m)), 0).astype(np.float32)
    return X, Xq


def layer_alphabet(W, bits, scalar):
    """Exactly the reference's expressions (:396, :544-545), evaluated by legacy numpy."""
    alphabet = np.linspace(-1, 1, num=int(round(2 ** bits)))
    rad = scalar * np.median(np.abs(W.flatten()))
    return rad * alphabet, rad


def replay_residual(w, X, Xq, q):
    """Residual the reference carries (:115, :119) and then discards; replayed with the same
    expressions under the same interpreter so the golden file can pin ||u||."""
    u = np.zeros(X.shape[1])
    for t in range(X.shape[0]):
        u += w[t] * X[t, :] - q[t] * Xq[t, :]
    return u


def to_index(q, alphabet):
    """Alphabet index of each reference output value; -1 = the literal 0 of rule (i) when 0 is
    not an alphabet member."""
    idx = np.full(q.shape, -1, dtype=np.int8)
    for k, a in enumerate(alphabet):
        idx[q == a] = k
    bad = (idx < 0) & (q != 0.0)
    assert not bad.any()
    return idx


def run_neuron(w, X, Xq, alphabet, workdir):
    fn = os.path.join(workdir, "neuron.h5")
    with h5py.File(fn, "w") as hf:
        hf.create_dataset("wX", shape=X.shape)    # default dtype f4, as the reference does (:487)
        hf.create_dataset("qX", shape=Xq.shape)
        hf["wX"][...] = X
        hf["qX"][...] = Xq
    q = ref._quantize_neuron_parallel(w, fn, alphabet)
    os.remove(fn)
    return q


def run_filter(filt, X, Xq, alphabet, workdir, channel_idx=3):
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        fn = "patch.h5"
        with h5py.File(fn, "w") as hf:
            hf.create_dataset(f"wX_channel{channel_idx}", data=X, chunks=True, maxshape=(None, None))
            hf.create_dataset(f"qX_channel{channel_idx}", data=Xq, chunks=True, maxshape=(None, None))
        q = ref._quantize_filter2D_parallel_jit(filt, channel_idx, fn, alphabet)
        os.remove(fn)
    finally:
        os.chdir(cwd)
    return q


def dense_cases(workdir):
    """Per-neuron recurrence (reference :91-121): several neurons per case."""
    specs = [
        # name,            N,   m,  bits,        scalar, seed, first, n_neurons
        ("dense_ternary",  96,  64, np.log2(3),  3,      10,   False, 6),
        ("dense_2bit",     64,  48, 2,           2,      11,   False, 4),
        ("dense_3bit",    128, 100, 3,           4,      12,   False, 4),
        ("dense_4bit",    256, 128, 4,           5,      13,   False, 4),
        ("dense_first",    80,  64, np.log2(3),  3,      14,   True,  4),
        ("dense_m1",       16,   1, 3,           2,      15,   False, 3),
        ("dense_ragged",   33,  77, 4,           5,      16,   False, 3),
        ("dense_mid",    1024, 256, np.log2(3),  3,      17,   False, 6),
        ("dense_mid4",    768, 192, 4,           5,      18,   False, 4),
    ]
    out = {}
    for name, N, m, bits, scalar, seed, first, C in specs:
        r = np.random.default_rng(seed + 1000)
        W = (r.standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
        X, Xq = relu_like(seed, N, m, first_layer=first)
        alphabet, rad = layer_alphabet(W, bits, scalar)
        Q = np.zeros((N, C))
        U = np.zeros((C, m))
        for j in range(C):
            Q[:, j] = run_neuron(W[:, j], X, Xq, alphabet, workdir)
            U[j] = replay_residual(W[:, j], X, Xq, Q[:, j])
        out[name] = dict(W=W, X=X, Xq=Xq, alphabet=alphabet, rad=np.float64(rad),
                         Q=Q, idx=to_index(Q, alphabet), resid=np.linalg.norm(U, axis=1), U=U,
                         bits=np.float64(bits), scalar=np.float64(scalar))
    return out


def edge_cases(workdir):
    out = {}
    # (a) dead features: all-zero quantized rows -> rule (i), literal 0 even for even alphabets.
    for name, bits in [("edge_dead_even", 2), ("edge_dead_odd", np.log2(3))]:
        N, m, C = 40, 32, 3
        r = np.random.default_rng(77)
        W = (r.standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
        X, Xq = relu_like(21, N, m)
        Xq[[0, 5, 6, 39], :] = 0          # includes t=0 and the last step
        X[[5, 17], :] = 0                 # analog row dead too / only analog dead
        alphabet, rad = layer_alphabet(W, bits, 2)
        Q = np.stack([run_neuron(W[:, j], X, Xq, alphabet, workdir) for j in range(C)], axis=1)
        resid = np.array([np.linalg.norm(replay_residual(W[:, j], X, Xq, Q[:, j])) for j in range(C)])
        out[name] = dict(W=W, X=X, Xq=Xq, alphabet=alphabet, rad=np.float64(rad), Q=Q,
                         idx=to_index(Q, alphabet), resid=resid)
    # (b) disjoint supports: <Xq_t, u> == 0 exactly although u != 0 -> rule (ii) (plain MSQ).
    N, m, C = 12, 16, 2
    X = np.zeros((N, m), dtype=np.float32)
    r = np.random.default_rng(5)
    for t in range(N):
        X[t, (t % 4) * 4:(t % 4) * 4 + 4] = r.random(4).astype(np.float32) + 0.5
    Xq = X.copy()
    W = (r.standard_normal((N, C))).astype(np.float32)
    alphabet, rad = layer_alphabet(W, 3, 2)
    Q = np.stack([run_neuron(W[:, j], X, Xq, alphabet, workdir) for j in range(C)], axis=1)
    resid = np.array([np.linalg.norm(replay_residual(W[:, j], X, Xq, Q[:, j])) for j in range(C)])
    out["edge_disjoint"] = dict(W=W, X=X, Xq=Xq, alphabet=alphabet, rad=np.float64(rad), Q=Q,
                                idx=to_index(Q, alphabet), resid=resid)
    # (c) exact ties: power-of-two data so the projection lands exactly midway between two
    #     alphabet members; first index must win (np.argmin).
    alphabet = np.array([-1.0, 0.0, 1.0])
    X = np.array([[1, 0, 0, 0], [1, 0, 0, 0], [0, 2, 0, 0], [1, 1, 0, 0], [2, 0, 0, 0]], dtype=np.float32)
    Xq = X.copy()
    W = np.array([[0.5, -0.5], [0.5, -0.5], [0.5, 1.5], [-0.5, 0.5], [0.25, 0.75]], dtype=np.float32)
    Q = np.stack([run_neuron(W[:, j], X, Xq, alphabet, workdir) for j in range(2)], axis=1)
    resid = np.array([np.linalg.norm(replay_residual(W[:, j], X, Xq, Q[:, j])) for j in range(2)])
    out["edge_ties"] = dict(W=W, X=X, Xq=Xq, alphabet=alphabet, rad=np.float64(1.0), Q=Q,
                            idx=to_index(Q, alphabet), resid=resid)
    # (d) tiny-norm rows straddling the 1e-16 (f32 norm) and 1e-10 (|<Xq,u>|) thresholds.
    N, m, C = 10, 8, 2
    r = np.random.default_rng(9)
    X = (r.random((N, m)) + 0.25).astype(np.float32)
    Xq = X.copy()
    Xq[2, :] = np.float32(1e-18)
    Xq[3, :] = np.float32(3e-17)
    Xq[4, :] = np.float32(5e-17)
    Xq[6, :] = np.float32(1e-12)
    Xq[7, :] = np.float32(1e-9)
    W = r.standard_normal((N, C)).astype(np.float32)
    alphabet, rad = layer_alphabet(W, 2, 2)
    Q = np.stack([run_neuron(W[:, j], X, Xq, alphabet, workdir) for j in range(C)], axis=1)
    resid = np.array([np.linalg.norm(replay_residual(W[:, j], X, Xq, Q[:, j])) for j in range(C)])
    out["edge_thresholds"] = dict(W=W, X=X, Xq=Xq, alphabet=alphabet, rad=np.float64(rad), Q=Q,
                                  idx=to_index(Q, alphabet), resid=resid)
    return out


def conv_cases(workdir):
    """Per-(channel, filter) recurrence (reference :185-233) on [kh*kw, m] patch matrices."""
    out = {}
    for name, kh, kw, m, bits, scalar, seed, F in [
        ("conv_3x3", 3, 3, 1500, 3, 4, 31, 5),
        ("conv_7x7", 7, 7, 640, np.log2(3), 3, 32, 3),
        ("conv_1x1", 1, 1, 300, 4, 5, 33, 4),
        ("conv_3x3_first", 3, 3, 900, 2, 2, 34, 3),
    ]:
        K = kh * kw
        r = np.random.default_rng(seed + 500)
        Wc = (r.standard_normal((kh, kw, F)) / np.sqrt(K)).astype(np.float32)
        X, Xq = relu_like(seed, K, m, first_layer=name.endswith("first"))
        if name == "conv_1x1":
            Xq[:, :] = Xq  # plain; rule (ii) fires at t=0 => MSQ
        alphabet, rad = layer_alphabet(Wc, bits, scalar)
        Q = np.zeros((kh, kw, F))
        resid = np.zeros(F)
        for f in range(F):
            Q[:, :, f] = run_filter(Wc[:, :, f], X, Xq, alphabet, workdir)
            resid[f] = np.linalg.norm(replay_residual(Wc[:, :, f].reshape(-1), X, Xq, Q[:, :, f].reshape(-1)))
        out[name] = dict(Wc=Wc, X=X, Xq=Xq, alphabet=alphabet, rad=np.float64(rad), Q=Q,
                         idx=to_index(Q, alphabet), resid=resid)
    return out


def bit_round_cases():
    out = {}
    r = np.random.default_rng(3)
    for M in (3, 4, 8, 16):
        alphabet = 0.37 * np.linspace(-1, 1, num=M)
        t64 = np.concatenate([r.standard_normal(64) * 0.4, alphabet, (alphabet[:-1] + alphabet[1:]) / 2,
                              [-5.0, 5.0, 0.0, -0.0]])
        t32 = (r.standard_normal(64) * 0.4).astype(np.float32)
        out[f"round64_M{M}"] = np.array([ref._bit_round_parallel(t, alphabet) for t in t64])
        out[f"round32_M{M}"] = np.array([ref._bit_round_parallel(t, alphabet) for t in t32])
        out[f"t64_M{M}"] = t64
        out[f"t32_M{M}"] = t32
        out[f"alphabet_M{M}"] = alphabet
    # the documented tie examples
    a = np.array([-1.0, 0.0, 1.0])
    out["tie_pos"] = np.float64(ref._bit_round_parallel(0.5, a))
    out["tie_neg"] = np.float64(ref._bit_round_parallel(-0.5, a))
    return out


class ArraySequence(Sequence):
    """Same behaviour as the reference's MNISTSequence (:235-263), kept local so the generator
    does not depend on Keras' Sequence base class."""

    def __init__(self, x, y, batch_size):
        self.x, self.y, self.batch_size = x, y, batch_size

    def __len__(self):
        return int(np.ceil(len(self.x) / self.batch_size))

    def __getitem__(self, i):
        return (np.array(self.x[i * self.batch_size:(i + 1) * self.batch_size]),
                np.array(self.y[i * self.batch_size:(i + 1) * self.batch_size]))


class ListLogger:
    def __init__(self):
        self.lines = []

    def info(self, msg):
        self.lines.append(msg)


def network_cases(workdir):
    """Whole-network Dense orchestration through the reference class (process pool included)."""
    out = {}
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        for name, dims, n_samples, batch, bits, scalar, use_bias, ignore in [
            ("net_mlp_full", (20, 16, 12, 5), 48, 16, np.log2(3), 3, True, []),
            ("net_mlp_partial", (12, 10, 6), 40, 16, 3, 2, True, []),      # 40 % 16 != 0 -> quirk
            ("net_mlp_nobias_ignore", (10, 8, 8, 4), 32, 32, 2, 2, False, [1]),
        ]:
            r = np.random.default_rng(abs(hash(name)) % 2 ** 31 if False else sum(map(ord, name)))
            layers = []
            for a, b in zip(dims[:-1], dims[1:]):
                K = (r.standard_normal((a, b)) / np.sqrt(a)).astype(np.float32)
                bias = (0.1 * r.standard_normal(b)).astype(np.float32) if use_bias else None
                layers.append(Dense(K, bias, activation="relu" if b != dims[-1] else "linear"))
            net = Sequential(layers)
            x = r.random((n_samples, dims[0])).astype(np.float32)
            y = np.zeros((n_samples, 1), dtype=np.float32)
            logger = ListLogger()
            qn = ref.QuantizedNeuralNetwork(network=net, batch_size=batch, get_data=ArraySequence(x, y, batch),
                                            logger=logger, ignore_layers=ignore, bits=bits, alphabet_scalar=scalar)
            recorded = {}
            orig_capture = qn._get_layer_data_generator

            def capture(layer_idx, transpose=False, _orig=orig_capture, _rec=recorded):
                # record the exact HDF5 contents each layer was quantized against
                fn = _orig(layer_idx, transpose)
                with h5py.File(fn, "r") as hf:
                    _rec[layer_idx] = (hf["wX"][...], hf["qX"][...])
                return fn

            qn._get_layer_data_generator = capture
            qn.quantize_network()
            qn._get_layer_data_generator = orig_capture
            case = dict(x=x, batch=np.int64(batch), bits=np.float64(bits), scalar=np.float64(scalar),
                        dims=np.array(dims), use_bias=np.bool_(use_bias), ignore=np.array(ignore, dtype=np.int64),
                        alphabet=qn.alphabet)
            for k, (la, lq) in enumerate(zip(net.layers, qn.quantized_net.layers)):
                case[f"W{k}"] = la.kernel
                case[f"Q{k}"] = lq.kernel                # float32 after Keras' cast
                if k in recorded:
                    case[f"wX{k}"], case[f"qX{k}"] = recorded[k]
                if use_bias:
                    case[f"b{k}"] = la.bias
                    case[f"qb{k}"] = lq.bias
            # Also pin the activation layout of the last quantized layer's inputs (quirk check):
            last = max(i for i in range(len(net.layers)) if i not in ignore)
            fn = qn._get_layer_data_generator(last, transpose=True)
            with h5py.File(fn, "r") as hf:
                case["wX_last"] = hf["wX"][...]
                case["qX_last"] = hf["qX"][...]
            os.remove(fn)
            case["last"] = np.int64(last)
            case["n_log_neuron_lines"] = np.int64(sum("quantized successfully." in l and "Neuron" in l
                                                      for l in logger.lines))
            out[name] = case
    finally:
        os.chdir(cwd)
    return out


def known_answer_settings():
    """The known answer the reference's tests/settings.py was written to check (never did):
    2->3->2 bias-free linear net with all-ones kernels, DATA=[[1,0],[0,2]], batch_size 1."""
    cwd = os.getcwd()
    net = Sequential([Dense(np.ones((2, 3)), None, "linear"), Dense(np.ones((3, 2)), None, "linear")])
    data = np.array([[1, 0], [0, 2]], dtype=np.float32)
    labels = np.array([[3], [6]], dtype=np.float32)
    qn = ref.QuantizedNeuralNetwork(network=net, batch_size=1, get_data=ArraySequence(data, labels, 1),
                                    logger=ListLogger())
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            fn = qn._get_layer_data_generator(1, transpose=True)
            with h5py.File(fn, "r") as hf:
                wX1, qX1 = hf["wX"][...], hf["qX"][...]
            fn0 = qn._get_layer_data_generator(0, transpose=True)
            with h5py.File(fn0, "r") as hf:
                wX0 = hf["wX"][...]
        finally:
            os.chdir(cwd)
    return dict(data=data, labels=labels, wX0=wX0, wX1=wX1, qX1=qX1,
                out=net.forward_upto(data, 1))


def main():
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as workdir:
        groups = {
            "dense": dense_cases(workdir),
            "edge": edge_cases(workdir),
            "conv": conv_cases(workdir),
            "network": network_cases(workdir),
        }
    for gname, cases in groups.items():
        flat = {}
        for cname, arrays in cases.items():
            for k, v in arrays.items():
                flat[f"{cname}__{k}"] = v
        np.savez_compressed(os.path.join(OUT, f"{gname}.npz"), **flat)
        print(gname, sorted(cases))
    np.savez_compressed(os.path.join(OUT, "bit_round.npz"), **bit_round_cases())
    np.savez_compressed(os.path.join(OUT, "settings_known_answer.npz"), **known_answer_settings())
    print("numpy", np.__version__, "-> wrote", OUT)


if __name__ == "__main__":
    main()
