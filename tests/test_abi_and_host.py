"""CPU: the C-ABI library builds for gfx950, loads, exports every symbol include/gpfq.h declares
(no compute calls without a GPU), and the host-side logic around it behaves."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from quantized_neural_networks_amd import build, hip
    build.build()
    return hip.load()


def test_header_symbols_exported(lib):
    from quantized_neural_networks_amd import hip
    header = open(os.path.join(ROOT, "include", "gpfq.h")).read()
    declared = set(re.findall(r"\b(gpfq_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(hip.SYMBOLS), declared ^ set(hip.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_version_and_errors_without_gpu(lib):
    assert lib.gpfq_version() == 306
    # argument validation happens before any launch: safe without a device
    a = (ctypes.c_double * 3)(-1.0, 0.0, 1.0)
    rc = lib.gpfq_quantize_neurons(None, None, 4, None, None, 4, a, 3, 1, 4, 8, 2, None, None, None, None, None, 0, 0, None)
    assert rc == -1 and b"NULL" in lib.gpfq_last_error()
    big = (ctypes.c_double * 257)(*range(257))
    rc = lib.gpfq_quantize_neurons(None, None, 4, None, None, 4, big, 257, -1, 4, 8, 2, None, None, None, None, None, 0, 0, None)
    assert rc == -2 and b"GPFQ_MAX_ALPHABET" in lib.gpfq_last_error()
    # 65..256 members (bits 7, 8 of the reference, scripts/quantized_network.py:396) are taken, with int16 indices
    assert lib.gpfq_quantize_neurons(None, None, 8, None, None, 4, big, 256, -1, 4, 8, 0, None, None, None, None, None, 0, 0, None) == 0
    assert [lib.gpfq_index_bits(M) for M in (1, 3, 4, 15, 16, 64, 65, 256, 257)] == [2, 2, 4, 4, 8, 8, 16, 16, 0]
    rc = lib.gpfq_quantize_neurons(None, None, 2, None, None, 4, a, 3, 1, 4, 8, 2, None, None, None, None, None, 0, 0, None)
    assert rc == -1
    assert lib.gpfq_quantize_neurons(None, None, 8, None, None, 4, a, 3, 1, 4, 8, 0, None, None, None, None, None, 0, 0, None) == 0
    # on-chip workspace: counter + row statistics (256-aligned) + iteration records of the pipelined kernel:
    # (N + 1 + 16) records of 128 + 16 * 1024 bytes
    assert lib.gpfq_workspace_bytes(9, 1024, 8, 0) == 512 + (9 + 17) * (128 + 16 * 1024)
    assert lib.gpfq_workspace_bytes(9, 100000, 8, 0) >= 8 * 100000 * 8
    assert lib.gpfq_workspace_bytes(9, 1024, 8, 2) > 0
    # an empty calibration set sizes a workspace without dividing by its slice count (SIGFPE in blk_shape)
    for path in (0, 1, 2):
        assert lib.gpfq_workspace_bytes(4, 0, 4, path) >= 0
    assert lib.gpfq_dense_layer_workspace_bytes(4, 0, 4) >= 0
    # round 6: the device-resident layer alphabet's entry points validate before they launch
    unit = (ctypes.c_double * 3)(-1.0, 0.0, 1.0)
    assert lib.gpfq_dense_layer_supported(64, 1024, 32, unit, 3) == 1
    assert lib.gpfq_dense_layer_supported(64, 200, 32, unit, 3) == 0        # rows of at most 256 samples: no block-pipelined kernel
    assert lib.gpfq_dense_layer_supported(64, 0, 32, unit, 3) == 0
    crooked = (ctypes.c_double * 3)(-1.0, 0.3, 1.0)
    assert lib.gpfq_dense_layer_supported(64, 1024, 32, crooked, 3) == 0    # not an arithmetic progression
    assert lib.gpfq_dense_layer_workspace_bytes(64, 1024, 32) > lib.gpfq_workspace_bytes(64, 1024, 32, 1)
    assert lib.gpfq_layer_alphabet_device(None, 3.0, unit, 3, None, None) == -1
    assert lib.gpfq_layer_alphabet_from_kernel(None, 16, 3.0, unit, 3, None, None, None, 0, None) == -1
    assert lib.gpfq_quantize_dense_layer(None, None, 1024, None, None, 32, 0, 32, None, unit, 3, 64, 1024, None, None, 1, 32, None, None, 0, None) == -1
    assert lib.gpfq_call_status(None, None) == -1


def test_patch_out_dim(lib):
    f = lib.gpfq_patch_out_dim
    assert f(32, 3, 1, 1, 1) == 32 and f(32, 3, 1, 1, 0) == 30
    assert f(224, 7, 2, 1, 0) == 109 and f(230, 7, 2, 1, 0) == 112 and f(224, 3, 2, 1, 1) == 112
    assert f(10, 3, 1, 2, 0) == 6 and f(2, 3, 1, 1, 0) == 0


def test_no_cpu_fallback():
    import torch
    from quantized_neural_networks_amd import hip
    X = torch.zeros((4, 8))
    with pytest.raises(hip.GpfqError):
        hip.quantize_neurons(X, X, torch.zeros((2, 4)), [-1.0, 0.0, 1.0])
    with pytest.raises(hip.GpfqError):
        hip.row_norms(X)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "quantized_neural_networks_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, re.M), f
                assert "gpfq_oracle" not in src, f


def test_auto_path_choice():
    """What GPFQ_PATH_AUTO resolves to (host logic only): on chip for rows that fit, Gram records for long rows with
    affordable walks (the reference's MNIST run), streaming for the rest."""
    from quantized_neural_networks_amd import hip
    ON, ST, GR = hip.GPFQ_PATH_ONCHIP, hip.GPFQ_PATH_STREAM, hip.GPFQ_PATH_GRAM
    assert hip.auto_path(4096, 1024, 4096) == ON                  # cfg2
    assert hip.auto_path(784, 512, 128) == ON                     # cfg1
    assert hip.auto_path(2048, 5008, 128) == ON                   # cfg4's dense head: walk too long for records
    assert hip.auto_path(784, 25000, 500) == GR                   # the MNIST MLP run: 25000 samples per row
    assert hip.auto_path(9, 5128192, 32) == GR                    # a conv channel's patch matrix
    assert hip.auto_path(128, 8193, 1000) == GR and hip.auto_path(128, 8193, 100) == ON     # wide layers cross over earlier
    assert hip.auto_path(4096, 50000, 64) == ST                   # long rows AND long walks
    assert hip.auto_path(784, 25000, 500, want_u=True) == ON      # residual vectors wanted: the kernels that hold u
    assert hip.auto_path(784, 40000, 500, want_u=True) == ST


def test_shard_bounds():
    from quantized_neural_networks_amd.layer import shard_bounds
    for n, w in [(4096, 8), (1000, 8), (3, 8), (10, 4), (0, 2)]:
        cover = []
        for r in range(w):
            lo, hi = shard_bounds(n, w, r)
            assert 0 <= lo <= hi <= n
            cover += list(range(lo, hi))
        assert cover == list(range(n))


def test_bit_round_host(golden):
    from quantized_neural_networks_amd.quantized_network import _bit_round_parallel
    g = golden("bit_round")
    for M in (3, 4, 8, 16):
        a = g[f"alphabet_M{M}"]
        got = np.array([_bit_round_parallel(t, a) for t in g[f"t64_M{M}"]])
        assert np.array_equal(got, g[f"round64_M{M}"])
        got = np.array([_bit_round_parallel(t, a) for t in g[f"t32_M{M}"]])
        assert np.array_equal(got, g[f"round32_M{M}"])


def test_sequences():
    from quantized_neural_networks_amd.quantized_network import CIFAR10Sequence, MNISTSequence
    x, y = np.arange(40).reshape(10, 4), np.arange(10)
    s = MNISTSequence(x, y, 4)
    assert len(s) == 3 and s.batch_size == 4
    assert np.array_equal(s[2][0], x[8:10]) and np.array_equal(s[0][1], y[0:4])
    assert len(CIFAR10Sequence(x, y, 5)) == 2


def test_module_surface():
    import quantized_network as qn
    for name in ["QuantizedNeuralNetwork", "QuantizedCNN", "MNISTSequence", "CIFAR10Sequence", "ImageNetSequence",
                 "_bit_round_parallel"]:
        assert hasattr(qn, name)
    import inspect
    sig = inspect.signature(qn.QuantizedNeuralNetwork.__init__)
    assert list(sig.parameters)[:9] == ["self", "network", "batch_size", "get_data", "mini_batch_size", "logger",
                                        "ignore_layers", "bits", "alphabet_scalar"]
    sig = inspect.signature(qn.QuantizedCNN.__init__)
    assert list(sig.parameters)[:10] == ["self", "network", "batch_size", "get_data", "mini_batch_size", "logger",
                                         "bits", "alphabet_scalar", "patch_mini_batch_size", "is_quantize_conv2d"]


def test_keras_shim_cpu():
    """The shim's layers against plain torch/numpy references (CPU tensors)."""
    import torch
    from quantized_neural_networks_amd import keras_shim as ks
    net = ks.Sequential([ks.Conv2D(4, 3, padding="same", activation="relu", input_shape=(8, 8, 3)),
                         ks.MaxPooling2D(), ks.Conv2D(5, 3, strides=2, padding="same"), ks.Flatten(),
                         ks.Dense(7, activation="relu"), ks.Dense(3, activation="softmax")], device="cpu")
    assert [l.__class__.__name__ for l in net.layers] == ["Conv2D", "MaxPooling2D", "Conv2D", "Flatten", "Dense", "Dense"]
    assert net.layers[4].input_shape == (None, 2 * 2 * 5) and net.layers[0].input_shape == (None, 8, 8, 3)
    x = np.random.default_rng(0).random((6, 8, 8, 3)).astype(np.float32)
    y = net.predict_on_batch(x)
    assert tuple(y.shape) == (6, 3) and torch.allclose(y.sum(1), torch.ones(6), atol=1e-5)
    clone = ks.clone_model(net)
    assert not np.array_equal(clone.get_weights()[0], net.get_weights()[0])
    clone.set_weights(net.get_weights())
    assert torch.equal(clone.predict_on_batch(x), y)
    trunc = ks.Model(inputs=net.layers[0].input, outputs=[net.layers[3].output])
    assert tuple(trunc.predict_on_batch(x).shape) == (6, 20)
    assert net.layers[5].inbound_nodes[0].inbound_layers is net.layers[4]
    # Dense against numpy
    d = ks.Sequential([ks.Dense(3, use_bias=False, input_shape=(2,))], device="cpu")
    d.layers[0].set_weights([np.ones((2, 3))])
    assert np.array_equal(d.predict_on_batch(np.array([[1, 0], [0, 2]], dtype=np.float32)).numpy(),
                          np.array([[1, 1, 1], [2, 2, 2]], dtype=np.float32))


def test_keras_shim_save_and_load(tmp_path):
    """save_model / load_model (what the drivers do with quantized_net, quantize_pretrained_mlp.py:87-95): one .npz with
    architecture and weights; the reloaded network predicts the same bits."""
    import torch
    from quantized_neural_networks_amd import keras_shim as ks
    net = ks.Sequential([ks.InputLayer(input_shape=(8, 8, 2)), ks.ZeroPadding2D(((1, 0), (0, 1))),
                         ks.Conv2D(4, (3, 2), strides=(2, 1), padding="same", activation="relu", use_bias=False),
                         ks.BatchNormalization(), ks.MaxPooling2D(), ks.Flatten(), ks.Dropout(0.3),
                         ks.Dense(5, activation="softmax")], device="cpu")
    x = np.random.default_rng(1).random((3, 8, 8, 2)).astype(np.float32)
    ks.save_model(net, tmp_path / "quantized_model")
    again = ks.load_model(tmp_path / "quantized_model", device="cpu")
    assert [l.name for l in again.layers] == [l.name for l in net.layers]
    assert all(np.array_equal(a, b) for a, b in zip(again.get_weights(), net.get_weights()))
    assert torch.equal(again.predict_on_batch(x), net.predict_on_batch(x))


def test_activation_capture_host_logic():
    """The cached-frontier capture equals the reference's per-batch recomputation from the input (CPU tensors):
    same column layout including the partial-last-batch quirk, with and without fix_partial_batch, transposed
    (Dense) and not (conv), and the two networks share one tensor until their weights differ."""
    import torch
    from quantized_neural_networks_amd import keras_shim as ks
    from quantized_neural_networks_amd.quantized_network import CIFAR10Sequence, QuantizedCNN

    def build():
        return ks.Sequential([ks.Conv2D(3, 3, padding="same", activation="relu", input_shape=(6, 6, 2)), ks.BatchNormalization(),
                              ks.MaxPooling2D(), ks.Conv2D(4, 3, padding="valid"), ks.Flatten(), ks.Dense(5, activation="relu"),
                              ks.Dense(2)], device="cpu", seed=4)

    x = np.random.default_rng(0).random((10, 6, 6, 2)).astype(np.float32)        # 10 samples, batches of 4: 4 + 4 + 2
    y = np.zeros((10, 1), dtype=np.float32)
    for fix in (False, True):
        qs = []
        for incremental in (True, False):
            q = QuantizedCNN(network=build(), batch_size=4, get_data=CIFAR10Sequence(x, y, 4), logger=None, device="cpu",
                             fix_partial_batch=fix)
            q.incremental_capture = incremental
            # make the quantized network differ from layer 3 on, as it would after quantizing that layer
            w = q.quantized_net.layers[3].get_weights()
            q.quantized_net.layers[3].set_weights([w[0] * 0.5] + w[1:])
            qs.append(q)
        for k, transpose in [(0, False), (1, False), (3, False), (5, True), (6, True)]:
            got = qs[0]._get_layer_data_generator(k, transpose)
            want = qs[1]._get_layer_data_generator(k, transpose)
            for a, b in zip(got, want):
                assert a.shape == b.shape and torch.equal(a == 0, b == 0)
                assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
            assert (got[0] is got[1]) == (k <= 3)                                 # shared until the weights differ
            n_cols = got[0].shape[-1] if transpose else got[0].shape[0]
            assert n_cols == 12                                                   # 3 batches x batch_size columns
            tail = got[0][..., 10:] if transpose else got[0][10:]
            head = got[0][..., 4:6] if transpose else got[0][4:6]
            if fix:
                assert (tail == 0).all()
            elif k > 0:
                assert (tail == 0).all() and not (head == 0).all()                # batch 2 landed at offset 2*2 = 4


def test_header_is_plain_c(tmp_path):
    """include/gpfq.h is a C header: it must compile as C99 without any HIP/C++ context."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "gpfq.h"\nint main(void) { int (*f)(const char *, int) = gpfq_set_option; return f ? GPFQ_MAX_ALPHABET - 64 : 1; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])


def test_accuracy_regression_entry_point_skips_cleanly_without_data(tmp_path, capsys):
    """the regression against model_metrics/*.csv needs the data set and the analog weights, which neither
    the reference checkout nor this image holds: the entry point must say so and exit 0 (no GPU touched)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("accuracy_regression", os.path.join(ROOT, "examples", "accuracy_regression.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    pub = tmp_path / "mnist_model_metrics.csv"
    pub.write_text(",data_set,analog_model,serialized_quantized_model,q_train_size,bits,alphabet_scalar,analog_test_acc,sd_test_acc,msq_test_acc\n"
                   "2020-07-15_1,mnist,a,b,25000,1.584962500721156,2,0.9824,0.9548,0.9330\n")
    assert mod.main(["--published", str(pub), "--dataset", str(tmp_path / "mnist.npz"), "--model", str(tmp_path / "m.npz")]) == 0
    assert "skipped" in capsys.readouterr().out
    assert mod.main(["--published", str(tmp_path / "none.csv")]) == 0


def test_cluster_form_options_and_workspace_sizing(lib):
    """Host side of the block kernel's cluster form (no launch): the options validate their values, and gpfq_workspace_bytes() follows
    them -- one record stream per 1024-sample slice, the exchange buffers per group of neurons."""
    from quantized_neural_networks_amd import hip
    ws = lambda N, m, C: int(lib.gpfq_workspace_bytes(N, m, C, hip.GPFQ_PATH_ONCHIP))
    try:
        assert lib.gpfq_set_option(b"blk_cluster", 500) != 0 and b"blk_cluster" in lib.gpfq_last_error()
        assert lib.gpfq_set_option(b"blk_cluster", -1) != 0
        for v in (0, 1, 1024, 4096):
            assert lib.gpfq_set_option(b"blk_cluster", v) == 0
        for key in (b"blk_cluster_nl", b"blk_cluster_map"):
            assert lib.gpfq_set_option(key, 0) == 0
        # default dispatch: 8192 samples = eight slices of the headline shape's records (16 B per sample and step + headers)
        assert lib.gpfq_set_option(b"blk_cluster", 1) == 0
        w8192, w1024 = ws(4096, 8192, 4096), ws(4096, 1024, 4096)
        assert w8192 > 7 * w1024 and w8192 < 10 * w1024
        assert ws(4096, 8192, 4096) == w8192                         # a pure function of its arguments and the options
        assert ws(4096, 8192, 8192) > w8192                          # more neurons: more exchange buffers
        # cluster form off: rows beyond 5120 samples need no block-kernel records at all
        assert lib.gpfq_set_option(b"blk_cluster", 0) == 0
        assert ws(4096, 8192, 4096) < w1024
        # ... and forced from 1025 samples up, a 2048-sample row takes two slices' records
        assert lib.gpfq_set_option(b"blk_cluster", 1024) == 0
        assert ws(4096, 2048, 4096) > 1.9 * w1024
    finally:
        lib.gpfq_set_option(b"blk_cluster", 1)
        lib.gpfq_set_option(b"blk_cluster_nl", 0)
        lib.gpfq_set_option(b"blk_cluster_map", -1)


def test_round6_options_and_median_workspace(lib):
    """Host side of round 6's late additions (no launch): the options are known to gpfq_set_option, an unknown key is refused with its
    name in the message, and the two-pass median's workspace holds its two fine histograms AND the three coarse ones the picks read."""
    try:
        for key, values in ((b"blk_prep_norms", (0, 1)), (b"blk_prep_run", (0, 1, 4, 16)), (b"blk_cluster768", (-1, 0, 8, 11)), (b"sync_errors", (0, 1))):
            for v in values:
                assert lib.gpfq_set_option(key, v) == 0, (key, v)
        assert lib.gpfq_set_option(b"blk_prep_norm", 1) != 0 and b"blk_prep_norm" in lib.gpfq_last_error()
    finally:
        for key, v in ((b"blk_prep_norms", 1), (b"blk_prep_run", 1), (b"blk_cluster768", -1), (b"sync_errors", 0)):
            lib.gpfq_set_option(key, v)
    fine = 4 * (2 * (1 << 15) + 2 * (1 << 16))
    assert int(lib.gpfq_median_abs_workspace_bytes_for(1 << 24)) == 128 + fine + 4 * 3 * 1024
    assert int(lib.gpfq_median_abs_workspace_bytes_for(1 << 24)) == int(lib.gpfq_median_abs_workspace_bytes_for(1000))   # (any kernel: a function of nothing but the form)
