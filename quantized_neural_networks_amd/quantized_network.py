"""Drop-in for the reference's ``scripts/quantized_network.py`` module surface.

Same public names, constructor signatures, attributes and log lines as the reference
(``QuantizedNeuralNetwork`` :331-590, ``QuantizedCNN`` :592-883, the three ``Sequence`` feeders
:235-329, ``_bit_round_parallel`` :40-57), so ``quantize_pretrained_{mlp,cnn,imagenet}.py`` can do
``from quantized_network import ...`` unchanged.  What differs is where the work
happens: calibration activations are captured straight into GPU tensors (no HDF5 files), and each
layer's greedy per-neuron loop is one call into the HIP kernels behind ``include/gpfq.h`` instead
of a process pool of Python workers.

The network object is only touched through the Keras attribute set of SURVEY A.4, so a real
``tf.keras`` model works when TensorFlow is installed; otherwise the torch-backed
``keras_shim`` supplies ``Model`` / ``clone_model``.

Extra keyword-only constructor arguments (not in the reference): ``device`` (torch device of this
process' GPU) and ``process_group`` (a ``torch.distributed`` group over which the neurons of every
layer -- and the samples of the activation capture in between -- are sharded), and ``fix_partial_batch`` to opt out of the reference's
partial-last-batch layout quirk (:491-495).
"""
import logging
from collections import namedtuple
from math import ceil
from time import time

import numpy as np
import torch

from . import hip, layer as _layer

try:  # real Keras when present (never on the MI355X image)
    from tensorflow.keras.models import Model, clone_model   # type: ignore
    from tensorflow.keras.utils import Sequence as _SequenceBase   # type: ignore
    HAVE_TF = True
except Exception:  # pragma: no cover - exercised on every box without TensorFlow
    from .keras_shim import Model, clone_model
    _SequenceBase = object
    HAVE_TF = False

SegmentedData = namedtuple("SegmentedData", ["wX_seg", "qX_seg"])


def _bit_round_parallel(t, alphabet):
    """Nearest member of the (scaled) alphabet to ``t``; the first one on ties (reference :40-57).
    Host-side scalar helper the drivers call per weight for their MSQ baseline; ``msq_quantize``
    below is the vectorised GPU form."""
    alphabet = np.asarray(alphabet)
    return alphabet[np.argmin(np.abs(alphabet - t))]


def msq_quantize(W, alphabet, device=None):
    """Whole-kernel MSQ on the GPU: equals ``[_bit_round_parallel(w, alphabet) for w in W.flatten()]``
    reshaped to ``W.shape`` (as float32, what Keras stores)."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    Wd = torch.from_numpy(np.ascontiguousarray(W, dtype=np.float32)).to(dev)
    Q, _ = hip.msq_round(Wd, np.asarray(alphabet, dtype=np.float64))
    return Q.cpu().numpy()


# ------------------------------------------------------------------------------------------
# data feeders (reference :235-329) -- same names and behaviour
# ------------------------------------------------------------------------------------------
class _SliceSequence(_SequenceBase):
    def __init__(self, x_set, y_set, batch_size):
        self.x, self.y = x_set, y_set
        self.batch_size = batch_size

    def __len__(self):
        return ceil(len(self.x) / self.batch_size)

    def _slice(self, idx):
        lo, hi = idx * self.batch_size, (idx + 1) * self.batch_size
        return self.x[lo:hi], self.y[lo:hi]

    def __getitem__(self, idx):
        bx, by = self._slice(idx)
        return np.asarray(bx), np.asarray(by)       # views: the reference copies here (np.array), nothing mutates them

    def _all_inputs(self):
        """(all inputs as ONE array, batch sizes): what iterating __getitem__ over the batches and concatenating gives, without the
        copies -- the activation capture takes the calibration set to the GPU in one piece (quantized_network._raw_inputs)."""
        n, bs = len(self.x), self.batch_size
        return np.asarray(self.x), [min(bs, n - lo) for lo in range(0, n, bs)]


class MNISTSequence(_SliceSequence):
    pass


class CIFAR10Sequence(_SliceSequence):
    pass


class ImageNetSequence(_SliceSequence):
    """x_set holds paths of ``.npy`` images; each is loaded and run through ``preprocess_func``."""

    def __init__(self, x_set, y_set, batch_size, preprocess_func):
        super().__init__(x_set, y_set, batch_size)
        self.preprocess_func = preprocess_func

    def __getitem__(self, idx):
        bx, by = self._slice(idx)
        return np.array([self.preprocess_func(np.load(f)) for f in bx]), np.array(by)

    _all_inputs = None                              # (files are loaded and preprocessed batch by batch)


# ------------------------------------------------------------------------------------------
class _LazyStats(dict):
    """last_layer_stats[layer]: rad, alphabet, resid, idx (, reruns).  The index tensor and the residual norms stay on the GPU until
    someone reads them (they are diagnostics: 103 MB of indices for VGG16's fc1 crossed PCIe after every layer until round 3);
    the first read copies to the host and keeps the NumPy array."""

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
            dict.__setitem__(self, key, v)
        return v

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    # (dict(stats), stats.copy(), pickle and ** read the underlying storage, not __getitem__: hand them host arrays -- ADVICE r04.
    #  Until a layer's statistics are read its index tensor stays in HBM: N x C bytes per Dense layer, 103 MB for VGG16's fc1.)
    def to_host(self):
        for k in list(self.keys()):
            self[k]
        return self

    def copy(self):
        return dict(self.to_host())

    def __iter__(self):
        self.to_host()
        return dict.__iter__(self)

    def __reduce__(self):
        return (dict, (dict(self.to_host()),))


class QuantizedNeuralNetwork:
    """Wrapper around a Keras-style model that quantizes its Dense layers (reference :331-590)."""

    def __init__(self, network, batch_size, get_data, mini_batch_size=32, logger=None, ignore_layers=[],
                 bits=np.log2(3), alphabet_scalar=1, *, device=None, process_group=None, fix_partial_batch=False):
        # batch_size and mini_batch_size are accepted and ignored, as in the reference (:371-400):
        # the sample count comes from get_data alone.
        self.get_data = get_data
        self.trained_net = network
        self.quantized_net = clone_model(network)
        self.quantized_net.set_weights(network.get_weights())
        self.alphabet_scalar = alphabet_scalar
        self.layer_dims = {
            layer_idx: layer.get_weights()[0].shape
            for layer_idx, layer in enumerate(network.layers)
            if layer.__class__.__name__ == "Dense"
        }
        self.bits = bits
        self.alphabet = np.linspace(-1, 1, num=int(round(2 ** (bits))))
        self.logger = logger
        self.ignore_layers = ignore_layers
        self._init_device(device, process_group, fix_partial_batch)

    # -- MI355X plumbing ------------------------------------------------------------------
    def _init_device(self, device, process_group, fix_partial_batch):
        if device is None:
            if not torch.cuda.is_available():
                raise hip.GpfqError("no GPU visible: the quantizer's hot path is HIP-only (no CPU fallback)")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.process_group = process_group
        self.fix_partial_batch = fix_partial_batch
        self.last_layer_stats = {}          # layer_idx -> dict(rad, alphabet, resid) for inspection/tests

    def _log(self, msg):
        if self.logger:
            self.logger.info(msg)
        else:
            print(msg)

    def _to_device(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float32)
        return torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float32)).to(self.device)

    # -- activation capture (reference :408-502) --------------------------------------------
    def _get_layer_data_generator(self, layer_idx, transpose=False):
        """Inputs of layer ``layer_idx`` in the analog (wX) and quantized (qX) networks for all
        calibration batches, as two GPU tensors ``(wX, qX)`` -- the reference's HDF5 datasets
        without the file.  Shape ``(len(get_data)*batch_size, *layer_input_shape)``, reversed when
        ``transpose`` (feature-major for Dense).  Batch b is written at offset b*(its own size), as
        in the reference (:491-495): with a partial last batch that overwrites earlier columns and
        leaves a zero tail.  ``fix_partial_batch=True`` writes batches back to back instead."""
        if self._incremental_capture_possible():
            return self._capture_incremental(layer_idx, transpose)
        if layer_idx == 0:
            inbound_analog = inbound_quant = None
        else:
            nodes_a = self.trained_net.layers[layer_idx].inbound_nodes
            nodes_q = self.quantized_net.layers[layer_idx].inbound_nodes
            if len(nodes_a) > 1 or len(nodes_q) > 1:
                raise NotImplementedError(f"layer {layer_idx} has several inbound nodes")
            inbound_analog, inbound_quant = nodes_a[0].inbound_layers, nodes_q[0].inbound_layers
            if not isinstance(inbound_analog, (list, tuple)):
                inbound_analog, inbound_quant = [inbound_analog], [inbound_quant]
            if len(inbound_analog) != 1 or len(inbound_quant) != 1:
                raise NotImplementedError(f"layer {layer_idx} has {len(inbound_analog)} inbound layers")

        layer = self.trained_net.layers[layer_idx]
        in_shape = layer.input_shape
        data_shape = tuple(in_shape[1:]) if in_shape[0] is None else tuple(in_shape)
        if layer_idx > 0:
            prev_trained = Model(inputs=self.trained_net.layers[0].input,
                                 outputs=[l.output for l in inbound_analog])
            prev_quant = Model(inputs=self.quantized_net.layers[0].input,
                               outputs=[l.output for l in inbound_quant])

        n_batches = self.get_data.__len__()
        num_images = n_batches * self.get_data.batch_size
        shape = (num_images,) + data_shape
        if transpose:
            shape = shape[::-1]
        wX_all = torch.zeros(shape, dtype=torch.float32, device=self.device)
        qX_all = wX_all if layer_idx == 0 else torch.zeros(shape, dtype=torch.float32, device=self.device)
        written = 0
        for b in range(n_batches):
            mini_batch = self.get_data.__getitem__(b)[0]
            if layer_idx == 0:
                wX = qX = self._to_device(mini_batch)
            else:
                wX = self._to_device(prev_trained.predict_on_batch(mini_batch))
                qX = self._to_device(prev_quant.predict_on_batch(mini_batch))
            k = wX.shape[0]
            lo = written if self.fix_partial_batch else b * k
            if transpose:
                perm = tuple(range(wX.dim() - 1, -1, -1))
                wX_all[..., lo:lo + k] = wX.permute(perm)
                if layer_idx != 0:
                    qX_all[..., lo:lo + k] = qX.permute(perm)
            else:
                wX_all[lo:lo + k] = wX
                if layer_idx != 0:
                    qX_all[lo:lo + k] = qX
            written += k
        return wX_all, qX_all

    # -- activation capture without recomputation (torch-backed Sequential networks) -------------------
    # The reference re-runs both truncated networks from the input for every layer and every batch
    # (:483-484): O(L^2) layer evaluations in 16-sample batches.  Layers are quantized front to back, so
    # the activations at the input of layer l follow from those at the input of the previously captured
    # layer by running only the layers in between -- with the weights layer l' has NOW (it was quantized
    # after its inputs were captured).  Samples are pushed through in large chunks; the reference's
    # batch structure only decides where columns land (including the partial-last-batch quirk).
    #
    # With a process group the SAMPLES are partitioned over the ranks for this part (shard_capture, on by default): every
    # rank pushes only its block of whole chunks through the layers in between, and the blocks are all-gathered when a
    # layer's inputs are needed (every rank walks all samples of its neurons / channels).  The chunk grid does not
    # depend on the number of ranks, so each chunk has the same shape as in a single-process run; the captured
    # activations are the same bits whenever the forward kernels are deterministic across processes (GEMMs are; MIOpen's
    # timed solver search per process can change the last bits of a convolution).
    _capture_chunk = None       # None: by the size of a sample (_chunk_samples); an int fixes it
    shard_capture = True

    def _chunk_samples(self):
        """Samples per forward chunk of the incremental capture: 2^25 input elements' worth, at least 512 and at most 16384 -- 16384 for
        the MNIST MLP (784 features), 8192 for CIFAR10 images (the reference's 5000 calibration images are one piece: no
        concatenation after every layer), 512 for ImageNet ones.  (Fixed at 512 until round 3: the MNIST run's
        25000 samples went through every layer in 49 pieces, 294 launches of ~10 us of host time each for 7 ms of kernels.)
        With a process group that shards the capture the chunk is capped so that every rank gets work: the largest power of two
        <= n / (2 world), at least 128 (with the uncapped grid 5000 CIFAR10 images were ONE chunk -- rank 0 ran every
        forward pass and the gather moved world x 8192 padded rows).  The grid depends on the GROUP SIZE, not on the rank: all ranks
        of a run cut the same chunks; a single-process run that is to be compared bit for bit with a sharded one pins
        `_capture_chunk` to the sharded run's value (chunk shapes can decide which GEMM / convolution kernel runs)."""
        if self._capture_chunk is not None:
            return int(self._capture_chunk)
        raw, _ = self._raw_inputs()
        feat = max(int(np.prod(raw.shape[1:])), 1)
        c = 512
        while c < 16384 and 2 * c * feat <= (1 << 25):
            c *= 2
        world, _ = _layer._group_info(self.process_group)
        if world > 1 and self.shard_capture:
            cap = 128
            while 2 * cap * 2 * world <= raw.shape[0]:
                cap *= 2
            c = min(c, cap)
        return c

    def _incremental_capture_possible(self):
        if not getattr(self, "incremental_capture", True) or len(self.trained_net.layers) != len(self.quantized_net.layers):
            return False
        if hasattr(self.trained_net, "forward_upto") and hasattr(self.quantized_net, "forward_upto"):
            return True
        return self._graph_capture_possible()

    def _graph_capture_possible(self):
        """Graph (functional-API) networks of the torch-backed shim: both networks expose their graph tables and every layer is a
        shim layer (a real tf.keras Model keeps the reference's truncated-Model capture)."""
        return all(getattr(n, "_functional", False) and hasattr(n, "graph_tables") and not hasattr(n, "forward_upto")
                   for n in (self.trained_net, self.quantized_net))

    def _raw_inputs(self):
        if getattr(self, "_raw", None) is None:
            whole = getattr(self.get_data, "_all_inputs", None)
            # (only when the feeder's batches ARE slices of its array: a subclass that overrides __getitem__ / _slice -- scaling,
            #  augmentation, lazy loading -- is iterated batch by batch as the reference does: ADVICE r04)
            cls = type(self.get_data)
            if whole is not None and not (getattr(cls, "__getitem__", None) is _SliceSequence.__getitem__
                                          and getattr(cls, "_slice", None) is _SliceSequence._slice):
                whole = None
            if whole is not None:
                # this module's own Sequences slice one array: the batches back to back ARE that array (no 313-way concatenate)
                arr, sizes = whole()
                self._raw = (self._to_device(arr), sizes)
            else:
                batches = [np.asarray(self.get_data.__getitem__(b)[0]) for b in range(self.get_data.__len__())]
                sizes = [int(a.shape[0]) for a in batches]
                self._raw = (self._to_device(batches[0] if len(batches) == 1 else np.concatenate(batches, axis=0)), sizes)
        return self._raw

    # layers that act on every element (or every sample's own row) with no reduction over samples and no workspace: called on the
    # whole block of samples at once (chunks + torch.cat would copy a 13 GB ResNet50 activation a second time for nothing)
    _WHOLE_BLOCK_LAYERS = {"Activation", "ReLU", "Add", "BatchNormalization", "ZeroPadding2D", "Dropout", "Flatten", "InputLayer"}

    @torch.no_grad()
    def _advance(self, layer, x):
        """layer.call on a block of samples in chunks of the capture grid; x: a tensor, or a list of tensors for a merging layer."""
        step = self._chunk_samples()
        multi = isinstance(x, (list, tuple))
        n = (x[0] if multi else x).shape[0]
        if n <= step or layer.__class__.__name__ in self._WHOLE_BLOCK_LAYERS:
            return layer.call(x)
        if multi:
            return torch.cat([layer.call([t[i:i + step] for t in x]) for i in range(0, n, step)])
        return torch.cat([layer.call(x[i:i + step]) for i in range(0, n, step)])

    def _capture_shard(self, n):
        """(world, lo, hi, blocks): this rank's block [lo, hi) of the n samples and every rank's block (`blocks[r]` = (lo_r, hi_r)) --
        whole chunks of the grid, dealt as evenly as whole chunks allow (the first n_chunks mod world ranks hold one more);
        world == 1 when the capture is not sharded."""
        world, rank = _layer._group_info(self.process_group)
        if world == 1 or not self.shard_capture:
            return 1, 0, n, [(0, n)]
        chunk = self._chunk_samples()
        n_chunks = -(-n // chunk)
        base, extra = divmod(n_chunks, world)
        blocks, c = [], 0
        for r in range(world):
            c_hi = c + base + (1 if r < extra else 0)
            blocks.append((min(c * chunk, n), min(c_hi * chunk, n)))
            c = c_hi
        return world, blocks[rank][0], blocks[rank][1], blocks

    def _gather_samples(self, x, n, world, blocks):
        """The ranks' sample blocks -> all n samples on every rank: one all-gather of blocks padded to the LARGEST block (at most one
        chunk more than the smallest), then the blocks' own rows back to back (the counts follow from the grid: no exchange)."""
        if world == 1:
            return x
        import torch.distributed as dist
        per = max(hi - lo for lo, hi in blocks)
        if x.shape[0] == per:
            pad = x.contiguous()
        else:
            pad = torch.zeros((per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
            pad[:x.shape[0]] = x
        out = torch.empty((world * per,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, pad, group=self.process_group)
        if all(hi - lo == per for lo, hi in blocks):
            return out[:n]
        return torch.cat([out[r * per:r * per + (hi - lo)] for r, (lo, hi) in enumerate(blocks) if hi > lo])

    # Look-ahead of the ANALOG network.  The analog activations at the input of the next layer to be quantized do not depend on
    # the layer being quantized now (only the quantized network's do: :461-462), so as soon as a layer's inputs are captured the
    # analog frontier is advanced to the next quantized layer on a second HIP stream -- under the quantization kernels of this
    # layer and, with a process group, under its index all-gather.  A latency-bound dense walk leaves most of the chip idle; a
    # conv layer's Gram kernels do not, and then the two simply share the GPU.  Same values: the same kernels on the same inputs.
    # Measured on one MI355X (tools/e2e_cnn.py, the CIFAR10 CNN at 5000 images): 48.0 ms without, 51 ms with -- on a single GPU the
    # conv layers' kernels fill the chip and the second stream only adds launches.  It should pay with a process group of more than
    # one rank (the all-gather and the shorter per-rank walks leave the GPU waiting), but that has only ever run with several ranks
    # sharing ONE GPU (tests/test_multirank_gpu.py forces it on and off: same bits): until a multi-GPU run has covered it the
    # default is OFF (round 5); `lookahead_capture = True` switches it on.
    lookahead_capture = None

    def _lookahead_enabled(self):
        return bool(self.lookahead_capture)

    def _next_quantized_layer(self, layer_idx):
        """Index of the next layer quantize_network() will quantize after `layer_idx`, or None."""
        for k in range(layer_idx + 1, len(self.trained_net.layers)):
            if self._will_quantize(k):
                return k
        return None

    def _will_quantize(self, k):
        return self.trained_net.layers[k].__class__.__name__ == "Dense" and k not in self.ignore_layers

    def _start_lookahead(self, fr, layer_idx):
        self._ahead = None
        nxt = self._next_quantized_layer(layer_idx)
        if (nxt is None or self.device.type != "cuda" or fr["w"].device.type != "cuda" or not self._lookahead_enabled()):
            return
        side = getattr(self, "_side_stream", None)
        if side is None:
            side = self._side_stream = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream(self.device)
        side.wait_stream(main)                                    # the frontier it starts from is being produced on `main`
        fr["w"].record_stream(side)                               # (and must outlive the side stream's reads if it is dropped early)
        tl = self.trained_net.layers
        with torch.cuda.stream(side):
            w = fr["w"]
            for k in range(fr["k"] + 1, nxt):
                w = self._advance(tl[k], w)
            ev = torch.cuda.Event()
            ev.record(side)
        self._ahead = dict(base=fr["w"], k0=fr["k"], k=nxt - 1, w=w, event=ev)

    def _take_lookahead(self, fr, layer_idx):
        """The analog activations at layer_idx - 1 if the side stream has them for THIS frontier, else None."""
        ahead, self._ahead = getattr(self, "_ahead", None), None
        if ahead is None or ahead["base"] is not fr["w"] or ahead["k0"] != fr["k"] or ahead["k"] != layer_idx - 1:
            return None
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ahead["event"])
        ahead["w"].record_stream(main)                            # allocated on the side stream, consumed (and later freed) on this one
        return ahead["w"]

    def _capture_incremental_graph(self, layer_idx, transpose):
        """The incremental capture on a GRAPH network (Keras-ResNet50's topology: skip connections, layers with several consumers).
        The reference rebuilds two truncated Models per layer and re-runs them from the input in 16-image batches
        (:456-462, :483-484): O(L^2) layer evaluations -- for ResNet50's 53 conv layers tens of seconds of forward passes around
        37 ms of quantization.  Here the frontier is the set of LIVE tensors of both networks after layer k (every output that a
        later layer still consumes: a residual branch stays alive until its Add), keyed by layer position; capturing layer l's
        inputs runs layers k+1 .. l-1 once, each on the outputs of its inbound layers, and releases a tensor after its last
        consumer.  While the two networks still agree (same input tensors, same weights) they share one tensor."""
        raw, sizes = self._raw_inputs()
        n = raw.shape[0]
        world, lo, hi, blocks = self._capture_shard(n)
        tl, ql = self.trained_net.layers, self.quantized_net.layers
        inbound, last_use = self.trained_net.graph_tables()
        if self.quantized_net.graph_tables()[0] != inbound:
            raise NotImplementedError("the quantized network's graph differs from the analog one's")
        src = inbound[layer_idx]
        if len(src) != 1:
            raise NotImplementedError(f"layer {layer_idx} has {len(src)} inbound layers")
        fr = getattr(self, "_frontier", None)
        if fr is None or not fr.get("graph") or fr["k"] > layer_idx - 1:
            mine = raw[lo:hi]
            fr = dict(graph=True, k=0, w={0: mine}, q={0: mine})    # layer 0 is the InputLayer: its output is the data
        fw, fq = fr["w"], fr["q"]
        for k in range(fr["k"] + 1, layer_idx):
            xw, xq = [fw[p] for p in inbound[k]], [fq[p] for p in inbound[k]]
            shared = all(a is b for a, b in zip(xw, xq)) and all(torch.equal(a, b) for a, b in zip(tl[k]._weights, ql[k]._weights))
            one = len(xw) == 1 and not isinstance(tl[k].inbound_nodes[0].inbound_layers, (list, tuple))
            w = self._advance(tl[k], xw[0] if one else xw)
            q = w if shared else self._advance(ql[k], xq[0] if one else xq)
            fw[k], fq[k] = w, q
            for p in inbound[k]:
                if last_use[p] == k:
                    fw.pop(p, None)
                    fq.pop(p, None)
            fr["k"] = k
        self._frontier = fr
        w_in, q_in = fw[src[0]], fq[src[0]]
        wX = self._assemble_capture(self._gather_samples(w_in, n, world, blocks), sizes, transpose)
        qX = wX if q_in is w_in else self._assemble_capture(self._gather_samples(q_in, n, world, blocks), sizes, transpose)
        return wX, qX

    def _capture_incremental(self, layer_idx, transpose):
        if not hasattr(self.trained_net, "forward_upto"):
            return self._capture_incremental_graph(layer_idx, transpose)
        raw, sizes = self._raw_inputs()
        n = raw.shape[0]
        world, lo, hi, blocks = self._capture_shard(n)
        fr = getattr(self, "_frontier", None)
        if fr is None or fr["k"] > layer_idx - 1:
            mine = raw[lo:hi]
            fr = dict(k=-1, w=mine, q=mine)                     # outputs of "layer -1" = the data itself
        tl, ql = self.trained_net.layers, self.quantized_net.layers

        steps = range(fr["k"] + 1, layer_idx)
        # (once per step: torch.equal on device tensors is a device sync)
        _same = {k: all(torch.equal(a, b) for a, b in zip(tl[k]._weights, ql[k]._weights)) for k in steps} if fr["q"] is fr["w"] else {}

        def same_weights(k):
            return _same.get(k, False)

        w_ahead = self._take_lookahead(fr, layer_idx) if len(steps) else None
        # the look-ahead holds the END of the analog chain only: it serves when the quantized chain either equals it all the
        # way (nothing quantized yet) or parts from it at the first step (the layer quantized last) -- front-to-back order
        # gives nothing else; otherwise the chains are walked together as before
        if w_ahead is not None and fr["q"] is fr["w"] and not all(same_weights(k) for k in steps) and same_weights(steps[0]):
            w_ahead = None
        if w_ahead is not None:
            if fr["q"] is fr["w"] and all(same_weights(k) for k in steps):
                q = w_ahead
            else:
                q = fr["q"]
                for k in steps:
                    q = self._advance(ql[k], q)
            fr = dict(k=layer_idx - 1, w=w_ahead, q=q)
        else:
            for k in steps:
                same = fr["q"] is fr["w"] and same_weights(k)
                w = self._advance(tl[k], fr["w"])
                q = w if same else self._advance(ql[k], fr["q"])
                fr = dict(k=k, w=w, q=q)
        self._frontier = fr
        full_w = raw if fr["k"] < 0 else self._gather_samples(fr["w"], n, world, blocks)
        wX = self._assemble_capture(full_w, sizes, transpose)
        # both networks still agree up to here (first quantized layer): one tensor, as for layer 0 (:478-481)
        if fr["q"] is fr["w"]:
            qX = wX
        else:
            qX = self._assemble_capture(self._gather_samples(fr["q"], n, world, blocks), sizes, transpose)
        self._start_lookahead(fr, layer_idx)                      # the analog network runs on while this layer is quantized
        return wX, qX

    def _assemble_capture(self, act, sizes, transpose):
        """Columns in the reference's layout (:491-495): batch b lands at offset b*(its own size)."""
        bs = self.get_data.batch_size
        n_batches = len(sizes)
        if not transpose and act.shape[0] == n_batches * bs and all(k == bs for k in sizes):
            return act                                           # full batches back to back ARE the captured block (read-only downstream)
        shape = (n_batches * bs,) + tuple(act.shape[1:])
        out = torch.zeros(shape[::-1] if transpose else shape, dtype=torch.float32, device=self.device)
        perm = tuple(range(act.dim() - 1, -1, -1))

        def put(lo, src):
            if transpose:
                out[..., lo:lo + src.shape[0]] = src.permute(perm)
            else:
                out[lo:lo + src.shape[0]] = src
        nfull = 0
        while nfull < n_batches and sizes[nfull] == bs:
            nfull += 1
        if nfull:
            put(0, act[:nfull * bs])                             # full batches: offset = running count
        start = nfull * bs
        written = start
        for b in range(nfull, n_batches):                        # partial batches (normally just the last)
            k = sizes[b]
            put(written if self.fix_partial_batch else b * k, act[start:start + k])
            start += k
            written += k
        return out

    def _update_weights(self, layer_idx, Q):
        """Install Q in the quantized network; the bias is carried over from the analog one (:504-521)."""
        fr = getattr(self, "_frontier", None)
        if fr is not None and fr["k"] >= layer_idx:
            self._frontier = None                                 # captured activations downstream of this layer are stale
        src, dst = self.trained_net.layers[layer_idx], self.quantized_net.layers[layer_idx]
        on_device = isinstance(getattr(dst, "_weights", None), list)      # torch-backed layer: weights never leave HBM
        if isinstance(Q, torch.Tensor) and not on_device:
            Q = Q.cpu().numpy()                                           # a real Keras layer wants arrays
        if src.use_bias:
            bias = src._weights[1] if on_device and isinstance(getattr(src, "_weights", None), list) else src.get_weights()[1]
            dst.set_weights([Q, bias])
        else:
            dst.set_weights([Q])

    def _kernel_on_device(self, layer):
        """The layer's kernel as a float32 tensor on the quantizer's device (no host round trip for torch-backed layers)."""
        ws = getattr(layer, "_weights", None)
        if isinstance(ws, list) and ws and isinstance(ws[0], torch.Tensor) and ws[0].device == self.device:
            return ws[0]
        return self._to_device(layer.get_weights()[0])

    def _layer_alphabet(self, Wd, layer_idx=None):
        # (:544-545) -- the median of |W| depends on the ANALOG kernel alone: quantize_network() queues the medians of all its layers
        # up front and reads them back with ONE host wait (_prefetch_medians) instead of one per layer
        med = getattr(self, "_medians", {}).pop(layer_idx, None) if layer_idx is not None else None
        if med is not None:
            rad = np.float64(self.alphabet_scalar) * np.float64(med)          # legacy-NumPy typing, as layer.layer_alphabet
            return rad * np.asarray(self.alphabet, dtype=np.float64), rad
        return _layer.layer_alphabet(Wd, self.alphabet, self.alphabet_scalar, self.process_group)

    def _prefetch_medians(self, layer_indices):
        """median(|W|) of the given layers' analog kernels, all queued before the first host wait (layers whose kernel is on the
        GPU and small enough for the one-GPU select; the others compute theirs when their turn comes)."""
        self._medians, self._medians_dev = {}, {}
        todo = []
        for k in layer_indices:
            Wd = self._kernel_on_device(self.trained_net.layers[k])
            world, _ = _layer._group_info(self.process_group)
            if Wd.is_cuda and Wd.numel() > 0 and (world == 1 or Wd.numel() < _layer._SHARDED_MEDIAN_MIN):
                todo.append((k, hip.median_abs(Wd.detach().reshape(-1), on_device=True)))
        for (k, t), v in zip(todo, hip.medians_to_host([t for _, t in todo])):
            self._medians[k] = v
            self._medians_dev[k] = t

    def _layer_alphabet_device(self, layer_idx, rad):
        """The same alphabet resident on the device (hip.DeviceAlphabet), formed from the prefetched DEVICE median: the Dense layer then
        runs as bench.py's step does -- the kernel reads the Keras kernel in place and writes Q and the indices in its layout, no
        neuron-major copy and no assembly pass.  The host already holds rad (last_layer_stats, the log), so nothing waits for it."""
        t = getattr(self, "_medians_dev", {}).pop(layer_idx, None)
        if t is None or not 1 <= len(self.alphabet) <= 64:
            return None
        d = hip.layer_alphabet_device(t, self.alphabet, self.alphabet_scalar)
        d._rad = np.float64(rad)
        d.radius_ok = bool(np.isfinite(rad) and rad > 0)          # known good on the host: no deferred alphabet status to wait for
        return d

    # -- Dense layer (reference :523-574) ---------------------------------------------------
    def _quantize_layer_parallel(self, layer_idx):
        Wd = self._kernel_on_device(self.trained_net.layers[layer_idx])
        N_ell, N_ell_plus_1 = Wd.shape
        self._log("\tFeeding input data through hidden layers...")
        tic = time()
        wX, qX = self._get_layer_data_generator(layer_idx, transpose=True)
        self._log(f"\tdone. {time()-tic:2f} seconds.")

        layer_alphabet, rad = self._layer_alphabet(Wd, layer_idx)
        dalpha = self._layer_alphabet_device(layer_idx, rad)

        self._log("\tQuantizing neurons (in parallel)...")
        tic = time()
        try:
            # residual norms are diagnostics (last_layer_stats): kept where the kernel holds the residual anyway
            # (a deferred failure of the kernel -- the cluster form's exchange timing out -- is noticed, logged and repaired INSIDE this
            #  call, before Q exists: nothing unchecked reaches set_weights, as nothing does in the reference, :563-565)
            out = _layer.quantize_dense(Wd, wX, qX, layer_alphabet if dalpha is None else dalpha, group=self.process_group, want_resid=None,
                                        log=lambda msg: self._log(f"\t\tLayer {layer_idx}: {msg}"))
            Q = out["Q"]
        except Exception as exc:
            self._log(f"\t\tLayer {layer_idx} generated an exception: {exc}")
            raise exc
        self._log_units("\t\tNeuron {} of " + f"{N_ell_plus_1} quantized successfully.", N_ell_plus_1)
        self._update_weights(layer_idx, Q)
        self._log(f"\tdone. {time()-tic:.2f} seconds.")
        self.last_layer_stats[layer_idx] = _LazyStats(rad=rad, alphabet=layer_alphabet, resid=out["resid"], idx=out["idx"])

    # The reference logs one record per neuron / filter as its futures complete (:567, :716).  Here all of a layer's units complete
    # together; the records still go out ONE PER UNIT (handlers and formatters that count or prefix records see what they saw), unless
    # the logger has INFO disabled (then nothing is formatted at all: 4096 logger calls cost about as much as the kernel that
    # quantized the neurons) or `batch_unit_log = True` asks for one multi-line record per layer.
    batch_unit_log = False

    def _log_units(self, fmt, n):
        lg = self.logger
        if lg is not None and hasattr(lg, "isEnabledFor") and not lg.isEnabledFor(logging.INFO):
            return
        if self.batch_unit_log:
            self._log("\n".join(fmt.format(i) for i in range(n)))
        else:
            for i in range(n):
                self._log(fmt.format(i))

    def quantize_network(self):
        """Quantizes all Dense layers that are not in ``ignore_layers``, in order (:576-590)."""
        num_layers = len(self.trained_net.layers)
        self._prefetch_medians([k for k, layer in enumerate(self.trained_net.layers)
                                if layer.__class__.__name__ == "Dense" and k not in self.ignore_layers])
        for layer_idx, layer in enumerate(self.trained_net.layers):
            if layer.__class__.__name__ == "Dense" and layer_idx not in self.ignore_layers:
                tic = time()
                self._log(f"Quantizing layer {layer_idx} (in parallel) of {num_layers}...")
                self._quantize_layer_parallel(layer_idx)
                self._log(f"Layer {layer_idx} of {num_layers} quantized successfully in {time() - tic:.2f} seconds.")


class QuantizedCNN(QuantizedNeuralNetwork):
    """Adds Conv2D / DepthwiseConv2D quantization (reference :592-883).  As in the reference the
    constructor does not chain to the parent's: there is no ``ignore_layers`` and no ``layer_dims``."""

    def __init__(self, network, batch_size, get_data, mini_batch_size=32, logger=None, bits=np.log2(3),
                 alphabet_scalar=1, patch_mini_batch_size=5000, is_quantize_conv2d=True, *,
                 device=None, process_group=None, fix_partial_batch=False):
        self.get_data = get_data
        self.trained_net = network
        self.quantized_net = clone_model(network)
        self.quantized_net.set_weights(network.get_weights())
        self.patch_mini_batch_size = patch_mini_batch_size      # kept for drop-in; patches are built on the GPU
        self.is_quantize_conv2d = is_quantize_conv2d
        self.alphabet_scalar = alphabet_scalar
        self.bits = bits
        self.alphabet = np.linspace(-1, 1, num=int(round(2 ** (bits))))
        self.logger = logger
        self._init_device(device, process_group, fix_partial_batch)

    def _will_quantize(self, k):
        name = self.trained_net.layers[k].__class__.__name__
        return name == "Dense" or (name in {"Conv2D", "DepthwiseConv2D"} and self.is_quantize_conv2d)

    def _quantize_dense_layer(self, layer_idx):
        super()._quantize_layer_parallel(layer_idx)

    def _quantize_conv2D_layer_parallel_jit(self, layer_idx):
        """Every (input channel, filter) pair of the kernel is quantized as an independent neuron of
        kh*kw weights against that channel's patch matrix (:815-867, :652-727).  The reference's
        (1,1)-filter shortcut is dead code (:835-842), so 1x1 kernels take the general path too."""
        self._log("\tFeeding input data through hidden layers...")
        tic = time()
        wX, qX = self._get_layer_data_generator(layer_idx)
        self._log(f"\tdone. {time()-tic:.2f} seconds.")

        layer = self.trained_net.layers[layer_idx]
        try:
            rate = layer.dilation_rate
        except Exception:
            rate = None
        Wd = self._kernel_on_device(layer)
        alphabet, rad = self._layer_alphabet(Wd, layer_idx)                        # (:831-832)
        num_channels = Wd.shape[-2]
        tic = time()
        self._log(f"\t\tBuilding patch arrays and quantizing channel filters for {num_channels} channels...")
        try:
            out = _layer.quantize_conv2d(Wd, wX, qX, alphabet, strides=tuple(layer.strides),
                                         padding=layer.padding.upper(), rate=tuple(rate) if rate else None,
                                         group=self.process_group,
                                         want_resid=False)      # residual norms are diagnostics: skip their replay
            Q = out["Q"]
        except Exception as exc:
            self._log(f"\t\t\tLayer {layer_idx} generated an exception: {exc}")
            raise Exception
        self._log(f"\t\tdone. {time()-tic:.2f} seconds.")
        self._update_weights(layer_idx, Q)
        self.last_layer_stats[layer_idx] = _LazyStats(rad=rad, alphabet=alphabet, resid=out["resid"], idx=out["idx"],
                                                      reruns=int(out.get("reruns", 0)))

    def quantize_network(self):
        num_layers = len(self.trained_net.layers)
        self._prefetch_medians([k for k, layer in enumerate(self.trained_net.layers)
                                if layer.__class__.__name__ == "Dense"
                                or (layer.__class__.__name__ in {"Conv2D", "DepthwiseConv2D"} and self.is_quantize_conv2d)])
        for layer_idx, layer in enumerate(self.trained_net.layers):
            if layer.__class__.__name__ == "Dense":
                self._log(f"Quantizing (Dense) layer {layer_idx} of {num_layers}...")
                tic = time()
                self._quantize_dense_layer(layer_idx)
                self._log(f"done. {time() - tic:.2f} seconds.")
            if layer.__class__.__name__ in {"Conv2D", "DepthwiseConv2D"} and self.is_quantize_conv2d:
                self._log(f"Quantizing ({layer.__class__.__name__}) layer {layer_idx} of {num_layers}...")
                tic = time()
                self._quantize_conv2D_layer_parallel_jit(layer_idx)
                self._log(f"done. {time() - tic:.2f} seconds.")
