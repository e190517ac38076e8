"""GPU, several ranks on ONE device (gloo carries the collectives through host memory): the real HIP path
under sharding -- neurons of a Dense layer, input channels of a conv layer (3x3 from planes, 5x5 implicit
im2col), image shards with all-reduced Gram records when there are fewer channels than ranks, index packing for
the all-gather, the median of the
alphabet radius with its counting sharded over the ranks -- reassembles the single-process result bit for bit.
(The multi-GPU runs use the same code with backend nccl = RCCL; only the transport differs.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs(case, dev):
    g = torch.Generator(device=dev).manual_seed(11)
    if case in ("dense", "dense_8bit", "dense_long", "dense_device", "dense_device_long", "dense_device_wide"):
        N, m, C = (120, 3100, 70) if case in ("dense_long", "dense_device_long") else ((40, 1024, 4500) if case == "dense_device_wide" else (300, 1024, 70))   # 70 neurons: uneven shards; dense_long: rows of the block kernel's cluster form (four slices per rank's launch)
        W = torch.randn((N, C), device=dev, generator=g) / np.sqrt(N)
        G = torch.randn((N, m), device=dev, generator=g)
        return dict(W=W, X=torch.relu(G), Xq=torch.relu(G + 0.1 * torch.randn((N, m), device=dev, generator=g)),
                    bits=8 if case == "dense_8bit" else 2)    # 8 bits: int16 indices, gathered as bytes
    if case == "dense_big_median":
        N, m, C = 2100, 256, 2048                                # 4.3 M weights: the sharded median path
        W = torch.randn((N, C), device=dev, generator=g) / np.sqrt(N)
        G = torch.randn((N, m), device=dev, generator=g)
        return dict(W=W[:, :], X=torch.relu(G), Xq=torch.relu(G + 0.1 * torch.randn((N, m), device=dev, generator=g)), bits=np.log2(3),
                    neurons=64)
    cin, k = {"conv3x3": (5, 3), "conv5x5": (4, 5), "conv_filters": (1, 3), "conv_columns7": (2, 7), "conv3x3_8bit": (5, 3),
              "conv_filters_8bit": (1, 3), "conv3x3_64": (64, 3), "conv3x3_12": (12, 3)}[case]
    act_w = torch.rand((40, 24, 24, cin), device=dev, generator=g)
    act_q = torch.relu(act_w + 0.05 * torch.randn(act_w.shape, device=dev, generator=g))
    W = torch.randn((k, k, cin, 6), device=dev, generator=g) / k
    return dict(W=W, act_w=act_w, act_q=act_q, bits=8 if case.endswith("8bit") else 3)


class _Quiet:
    def info(self, msg):
        pass


def _run_network(case, dev, group, lookahead=None):
    """The class surface over a process group: neurons / channels sharded per layer AND the samples of the activation
    capture in between (quantized_network.py: shard_capture).  A capture chunk of 8 samples spreads the 44 samples (6
    chunks, the last one partial, and a partial last batch) over the ranks; the chunk grid does not depend on the
    number of ranks, so the single-process run must give the same bits."""
    from quantized_neural_networks_amd import keras_shim as ks, quantized_network as qn
    r = np.random.default_rng(5)
    if case == "network_resnet":
        # ResNet50's topology in small (keras_shim.ResNet50, quantize_pretrained_imagenet.py:10): conv1 7x7 / 2 on the zero-padded
        # 3-channel input (fewer channels than 8 ranks: image shards + all-reduced records), max-pool, a bottleneck block with a
        # projection shortcut on 64 channels (shards of 8 at world size 8) and one with stride 2, global pooling, a 5-way classifier
        # (fewer neurons than ranks: empty shards)
        def block(x, filters, stride, name):
            sc = ks.Conv2D(4 * filters, 1, strides=stride, name=name + "_0_conv")(x)
            sc = ks.BatchNormalization(epsilon=1.001e-5, name=name + "_0_bn")(sc)
            y = ks.Conv2D(filters, 1, strides=stride, name=name + "_1_conv")(x)
            y = ks.Activation("relu", name=name + "_1_relu")(ks.BatchNormalization(epsilon=1.001e-5, name=name + "_1_bn")(y))
            y = ks.Conv2D(filters, 3, padding="same", name=name + "_2_conv")(y)
            y = ks.Activation("relu", name=name + "_2_relu")(ks.BatchNormalization(epsilon=1.001e-5, name=name + "_2_bn")(y))
            y = ks.BatchNormalization(epsilon=1.001e-5, name=name + "_3_bn")(ks.Conv2D(4 * filters, 1, name=name + "_3_conv")(y))
            return ks.Activation("relu", name=name + "_out")(ks.Add(name=name + "_add")([sc, y]))
        inp = ks.Input((38, 38, 3), name="input_1")
        x = ks.Conv2D(64, 7, strides=2, name="conv1_conv")(ks.ZeroPadding2D(3, name="conv1_pad")(inp))
        x = ks.Activation("relu", name="conv1_relu")(ks.BatchNormalization(epsilon=1.001e-5, name="conv1_bn")(x))
        x = ks.MaxPooling2D(3, strides=2, name="pool1_pool")(ks.ZeroPadding2D(1, name="pool1_pad")(x))
        x = block(x, 16, 1, "conv2_block1")
        x = block(x, 16, 2, "conv3_block1")
        out = ks.Dense(5, activation="softmax", name="predictions")(ks.GlobalAveragePooling2D(name="avg_pool")(x))
        net = ks.Model(inp, out, seed=4, name="resnet_small")
        x = r.random((44, 38, 38, 3)).astype(np.float32)
        q = qn.QuantizedCNN(network=net, batch_size=16, get_data=qn.CIFAR10Sequence(x, np.zeros((44, 5), np.float32), 16),
                            logger=_Quiet(), bits=np.log2(3), alphabet_scalar=3, process_group=group)
    elif case == "network_cnn":
        net = ks.Sequential([
            ks.Conv2D(4, 3, padding="same", activation="relu", input_shape=(12, 12, 3)),
            ks.Conv2D(5, 3, strides=2, padding="same", activation="relu"),
            ks.DepthwiseConv2D(3, padding="valid", depth_multiplier=2, use_bias=False),
            ks.Conv2D(3, 1, padding="valid"),
            ks.Flatten(),
            ks.Dense(6, activation="softmax"),
        ], seed=3)
        x = r.random((44, 12, 12, 3)).astype(np.float32)
        q = qn.QuantizedCNN(network=net, batch_size=16, get_data=qn.CIFAR10Sequence(x, np.zeros((44, 6), np.float32), 16),
                            logger=_Quiet(), bits=3, alphabet_scalar=4, process_group=group)
    else:
        n = 700 if case == "network_mlp_grid" else 44
        net = ks.Sequential([ks.Dense(40, activation="relu", input_shape=(30,)), ks.Dense(24, activation="relu"), ks.Dense(5)], seed=2)
        x = r.random((n, 30)).astype(np.float32)
        q = qn.QuantizedNeuralNetwork(network=net, batch_size=16, get_data=qn.MNISTSequence(x, np.zeros((n, 1)), 16),
                                      logger=_Quiet(), bits=2, alphabet_scalar=2, process_group=group)
    q._capture_chunk = 8
    if case == "network_mlp_grid":
        # the DEFAULT chunk grid under a process group (nothing pinned): 700 samples over 2 ranks are capped to 128-sample
        # chunks (6 chunks: 3 per rank, the last one short); the single-process run it is compared with pins the same 128
        q._capture_chunk = None if group is not None else 128
        if group is not None:
            assert q._chunk_samples() == 128
            world, lo, hi, blocks = q._capture_shard(n)
            assert [b[1] - b[0] for b in blocks] == [384, 316]
    if lookahead is not None:
        q.lookahead_capture = lookahead                          # (None: the default -- off since round 5)
    captured = []
    orig = q._get_layer_data_generator

    def wrapped(layer_idx, transpose=False):
        wX, qX = orig(layer_idx, transpose)
        captured.append((layer_idx, wX.cpu().numpy(), qX.cpu().numpy()))
        return wX, qX

    q._get_layer_data_generator = wrapped
    q.quantize_network()
    res = {}
    for k, layer in enumerate(q.quantized_net.layers):
        ws = layer.get_weights()
        if ws:
            res[f"Q{k}"] = np.asarray(ws[0])
    for k, wX, qX in captured:
        res[f"wX{k}"], res[f"qX{k}"] = wX, qX
    if group is not None:                                        # this rank really advanced only its block of samples
        n_all = q._raw_inputs()[0].shape[0]
        world, lo, hi, per = q._capture_shard(n_all)
        assert world > 1 and hi - lo < n_all
        fr = getattr(q, "_frontier", None)
        if fr is not None:                                       # (a graph network's frontier is the dict of its live tensors)
            live = list(fr["w"].values()) if isinstance(fr["w"], dict) else [fr["w"]]
            assert live and all(t.shape[0] == hi - lo for t in live)
    return res


def _run(case, dev, group):
    from quantized_neural_networks_amd import layer
    if case.startswith("network"):
        name, _, la = case.partition("+")                        # "network_mlp+la0" / "+la1": the analog look-ahead stream forced off / on
        return _run_network(name, dev, group, {"": None, "la0": False, "la1": True}[la])
    d = _inputs(case, dev)
    unit = np.linspace(-1, 1, int(round(2 ** d["bits"])))
    if case == "dense_big_median":
        layer._SHARDED_MEDIAN_MIN = 1 << 22                      # (the sharded counting protocol at a test's size: the default threshold is 32 M weights since round 6)
    if case.startswith("dense_device"):
        # round 6: the layer driver with the alphabet formed and kept on the device (median -> rad * alphabet -> shard's kernel reading the
        # Keras kernel -> all-gather of packed indices -> assembly from the device alphabet), the pre-pass on a second stream; _wide: shards
        # of the 16-neuron shape at world size 2 (the Keras-layout flush on one GPU, neuron-major shards under a group)
        out = layer.quantize_dense_layer(d["W"], d["X"], d["Xq"], unit, 3, group=group, overlap=(case != "dense_device_long"))
        res = {k: v.cpu().numpy() for k, v in out.items() if k in ("Q", "idx", "resid")}
        res["rad"] = np.float64(out["alphabet"].rad())
        ref_alphabet, ref_rad = layer.layer_alphabet(d["W"], unit, 3, group)
        assert res["rad"] == ref_rad
        if group is None:                                        # ... equals the host alphabet's layer driver
            ref = layer.quantize_dense(d["W"], d["X"], d["Xq"], ref_alphabet)
            assert torch.equal(ref["Q"], out["Q"]) and torch.equal(ref["idx"], out["idx"]) and torch.equal(ref["resid"], out["resid"])
        return res
    alphabet, rad = layer.layer_alphabet(d["W"], unit, 3, group)
    if case.startswith("dense"):
        W = d["W"][:, :d["neurons"]].contiguous() if "neurons" in d else d["W"]
        out = layer.quantize_dense(W, d["X"], d["Xq"], alphabet, group=group)
    else:
        out = layer.quantize_conv2d(d["W"], d["act_w"], d["act_q"], alphabet, strides=(1, 1), padding="SAME", rate=(1, 1),
                                    group=group, want_resid=False)
    res = {k: v.cpu().numpy() for k, v in out.items() if isinstance(v, torch.Tensor) and k != "reruns"}
    res["rad"] = np.float64(rad)
    return res


def _worker(rank, world, port, case, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = _run(case, torch.device("cuda", 0), dist.group.WORLD)
    np.savez(os.path.join(result_dir, f"{case}_{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("case,world", [("dense", 2), ("dense", 3), ("dense_long", 2), ("dense_big_median", 2), ("conv3x3", 2), ("conv5x5", 3),
                                        ("dense_device", 2), ("dense_device", 8), ("dense_device_long", 2), ("dense_device_wide", 2),
                                        ("dense_8bit", 3), ("conv3x3_8bit", 2), ("conv_filters_8bit", 2),
                                        ("network_mlp", 2), ("network_cnn", 2), ("network_cnn", 3), ("network_mlp_grid", 2),
                                        ("conv_filters", 2), ("conv_columns7", 3),    # fewer channels than ranks: image shards
                                        # world size 8 (the north-star's): 70 neurons / 12 and 64 channels over eight ranks, 5 / 2 / 1
                                        # channels (fewer than ranks: records over image shards), the class surface with 6 sample
                                        # chunks over 8 ranks (empty sample shards), ResNet50's topology in small
                                        ("dense", 8), ("dense_8bit", 8), ("conv3x3_12", 8), ("conv3x3_64", 8), ("conv3x3", 8),
                                        ("conv_columns7", 8), ("conv_filters", 8), ("network_mlp", 8), ("network_cnn", 8),
                                        ("network_resnet", 8)])
def test_ranks_sharing_one_gpu(case, world, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"{case}_{r}.npz") for r in range(world)]
    single = _run(case, torch.device("cuda", 0), None)
    if case in ("network_cnn", "network_resnet"):
        # (round 5: a graph network's capture is incremental and sharded by samples like a Sequential one's -- each rank advances the
        #  LIVE tensors of its block of samples and the blocks are all-gathered when a layer's inputs are needed)
        # The ranks hold the same gathered activations and take the same decisions: bit-identical to EACH OTHER.  Against the
        # single-process run the captured activations are the same up to the forward kernels' determinism: MIOpen picks a
        # convolution solver by a timed search the first time a process meets a shape, so two processes may run different
        # solvers on the same chunk (last-bit differences, seen about once in five runs); a GPFQ decision that sits within such
        # a difference of a boundary may then differ.  GEMM layers (network_mlp, the layer-level cases) are compared bit for bit.
        for k in res[0].files:
            for r in range(1, world):
                assert np.array_equal(res[r][k], res[0][k]), (k, r)
        for k, v in single.items():
            if k.startswith(("wX", "qX")):
                np.testing.assert_allclose(res[0][k], v, rtol=1e-4, atol=1e-6, err_msg=k)
            else:
                assert res[0][k].shape == v.shape and np.mean(res[0][k] != v) < 0.02, k
        return
    for k, v in single.items():
        if k == "resid" and not case.startswith("dense"):
            continue                                             # NaN placeholders when residual norms are not requested
        for r in range(world):
            assert np.array_equal(res[r][k], v), (k, r)


@pytest.mark.parametrize("case", ["network_mlp", "network_cnn"])
def test_lookahead_stream_on_equals_off_under_a_process_group(case, tmp_path):
    """The analog look-ahead capture (a second HIP stream, on by default with more than one rank) against the same two ranks with
    it forced off: captured activations and quantized kernels of every layer, bit for bit (same kernels on the same inputs, in
    the same processes' order of first use)."""
    import torch.multiprocessing as mp
    out = {}
    for la in ("la0", "la1"):
        name = f"{case}+{la}"
        mp.spawn(_worker, args=(2, _free_port(), name, str(tmp_path)), nprocs=2, join=True)
        out[la] = [np.load(tmp_path / f"{name}_{r}.npz") for r in range(2)]
    for r in range(2):
        assert sorted(out["la0"][r].files) == sorted(out["la1"][r].files)
    for k in out["la0"][0].files:
        if case == "network_mlp":                                # GEMM layers: deterministic across processes
            assert np.array_equal(out["la0"][0][k], out["la1"][0][k]), k
        else:                                                    # (MIOpen's per-process solver choice: see test_ranks_sharing_one_gpu)
            a, b = out["la0"][0][k], out["la1"][0][k]
            assert a.shape == b.shape
            if k.startswith(("wX", "qX")):
                np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-6, err_msg=k)
            else:
                assert np.mean(a != b) < 0.02, k
        for la in ("la0", "la1"):                                # and the two ranks of a run agree exactly, either way
            assert np.array_equal(out[la][0][k], out[la][1][k]), (la, k)
