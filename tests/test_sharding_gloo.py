"""CPU, world_size 2 and 8, gloo: the neuron/channel sharding and the single all-gather per layer
reassemble exactly the unsharded result.  The collective plumbing is what is under test (the HIP kernels are covered
by the -m gpu tests): THIS TEST replaces the entry points of the HIP binding (`hip.quantize_neurons`, ...) by
oracle-backed stand-ins on CPU tensors in its own worker processes; the product has no hook for that -- layer.py calls
the binding directly and the binding refuses CPU tensors."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _standin_quantize(X, Xq, Wt, alphabet, **kw):
    import oracle
    W = Wt.numpy().T.copy()
    Q, idx, resid = oracle.layer(W, X.numpy(), Xq.numpy(), np.asarray(alphabet))
    idx = idx.astype(np.int8 if len(alphabet) <= 64 else np.int16)         # the binding's index element types
    return dict(Q=torch.from_numpy(Q.astype(np.float32)), idx=torch.from_numpy(idx), resid=torch.from_numpy(resid), u=None)


def _standin_assemble(qidx, alphabet, want_idx=True, bits=None, N=None):
    assert bits in (None, 8, 16)
    a = np.asarray(alphabet, dtype=np.float64)
    k = qidx.numpy().astype(np.int64)
    Q = np.where(k < 0, 0.0, a[np.maximum(k, 0)]).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(Q.T)), torch.from_numpy(np.ascontiguousarray(qidx.numpy().T))


def _standin_patches(act, channel, kernel_size, strides, rate, padding, out=None):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _im2col_ref import patches
    rh, rw = rate if rate else (1, 1)
    return torch.from_numpy(patches(act.numpy(), channel, kernel_size[0], kernel_size[1], strides[0], strides[1], rh, rw, padding))


def _standin_pack(qidx, M):
    return qidx, 8 if M <= 64 else 16       # plain int8 / int16 indices, no packing (the packing kernel is covered on the GPU)


def _standin_planes(act, c_lo, c_hi):
    return act[..., c_lo:c_hi].permute(3, 0, 1, 2).contiguous()


def _standin_neuron_major(W, lo=0, hi=None):
    return W[:, lo:hi].t().contiguous()


# ---- round 6: the device-resident alphabet's entry points (layer.quantize_dense_layer under a process group) ----
def _standin_median(W, meanwhile=None, on_device=False):
    return torch.from_numpy(np.array([np.median(np.abs(W.numpy()))], dtype=np.float32))


def _standin_layer_alphabet_device(median32, unit_alphabet, alphabet_scalar):
    from quantized_neural_networks_amd import hip
    buf = torch.zeros(hip.GPFQ_DEVICE_ALPHABET_BYTES, dtype=torch.uint8)
    buf[:8] = torch.from_numpy(np.array([np.float64(alphabet_scalar) * np.float64(median32.numpy()[0])]).view(np.uint8))   # the legacy-NumPy product (:544)
    return hip.DeviceAlphabet(buf, unit_alphabet, alphabet_scalar)


def _standin_dense_layer(X, Xq, W, dalpha, lo=0, hi=None, nrm32=None, keras_out=True, want_values=True, want_idx=True, want_resid=True, prepared=None):
    import oracle
    hi = W.shape[1] if hi is None else hi
    Wn = np.ascontiguousarray(W.numpy()[:, lo:hi])
    if hi > lo:
        Q, idx, resid = oracle.layer(Wn, X.numpy(), Xq.numpy(), dalpha.values())
    else:
        Q, idx, resid = np.zeros((0, W.shape[0])), np.zeros((0, W.shape[0]), np.int8), np.zeros(0)
    out = dict(idx=torch.from_numpy(idx.astype(np.int8)), Q=torch.from_numpy(Q.astype(np.float32)), resid=torch.from_numpy(resid), u=None,
               workspace=torch.zeros(16, dtype=torch.uint8))
    if keras_out:
        out["idx"], out["Q"] = out["idx"].t().contiguous(), out["Q"].t().contiguous()
    return out


def _standin_assemble_device(qidx, dalpha, want_idx=True, bits=8, N=None):
    return _standin_assemble(qidx, dalpha.values(), want_idx, bits, N)


def _install_standins():
    """Swap the binding's entry points for CPU stand-ins in THIS process; returns the originals."""
    sys.path.insert(0, ROOT)
    from quantized_neural_networks_amd import hip
    names = dict(quantize_neurons=_standin_quantize, extract_patches=_standin_patches, assemble_kernel=_standin_assemble,
                 pack_indices=_standin_pack, channel_planes=_standin_planes, neuron_major=_standin_neuron_major,
                 median_abs=_standin_median, layer_alphabet_device=_standin_layer_alphabet_device,
                 layer_alphabet_from_kernel=lambda W, unit, scalar: _standin_layer_alphabet_device(_standin_median(W), unit, scalar),
                 dense_layer_supported=lambda N, m, C, unit: True, quantize_dense_layer=_standin_dense_layer,
                 assemble_kernel_device=_standin_assemble_device, call_status=lambda result: 0, last_dense_kernel=lambda: "stand-in")
    keep = {k: getattr(hip, k) for k in names}
    for k, v in names.items():
        setattr(hip, k, v)
    return hip, keep


def _worker(rank, world, port, case, result_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quantized_neural_networks_amd import layer
    _install_standins()
    group = dist.group.WORLD                # sharding needs an explicit group (None = this process alone)
    r = np.random.default_rng(7)
    if case.startswith("dense"):
        N, m, C = 24, 40, 7                                  # 7 neurons over 2 ranks: uneven shards
        W = (r.standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
        G = r.standard_normal((N, m))
        X = np.maximum(G, 0).astype(np.float32)
        Xq = np.maximum(G + 0.1 * r.standard_normal((N, m)), 0).astype(np.float32)
        alphabet = 0.3 * np.linspace(-1, 1, _dense_members(case))
        if case == "dense_device":
            # round 6: median -> device alphabet -> shard -> all-gather of the indices -> assembly from the device alphabet (one stream:
            # the second stream of the GPU form has no CPU counterpart)
            out = layer.quantize_dense_layer(torch.from_numpy(W), torch.from_numpy(X), torch.from_numpy(Xq), np.linspace(-1, 1, 4), 2.0,
                                             group=group, overlap=False)
            out = {k: v for k, v in out.items() if k in ("Q", "idx", "resid")}
        else:
            out = layer.quantize_dense(torch.from_numpy(W), torch.from_numpy(X), torch.from_numpy(Xq), alphabet, group=group)
    else:
        Cin = 3 if case == "conv_channels" else 1            # Cin < world -> filters are sharded instead
        act = r.random((4, 6, 6, Cin)).astype(np.float32)
        actq = (act + 0.05 * r.random(act.shape)).astype(np.float32)
        W = (r.standard_normal((3, 3, Cin, 5)) / 3).astype(np.float32)
        alphabet = 0.25 * np.linspace(-1, 1, 3)
        out = layer.quantize_conv2d(torch.from_numpy(W), torch.from_numpy(act), torch.from_numpy(actq), alphabet,
                                    strides=(1, 1), padding="SAME", rate=(1, 1), group=group)
    np.savez(os.path.join(result_dir, f"{case}_{rank}.npz"), **{k: v.numpy() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


def _dense_members(case):
    return 200 if case == "dense_int16" else 4              # 200 members: int16 indices, gathered as bytes


@pytest.mark.parametrize("case,world", [("dense", 2), ("dense_int16", 2), ("dense_device", 2), ("dense_device", 8), ("conv_channels", 2), ("conv_filters", 2),
                                        # the north-star's world size: 7 neurons / 3 channels / 5 filters over EIGHT ranks -- empty
                                        # shards, `per`-padded gathers, fewer channels than ranks (filters are sharded instead)
                                        ("dense", 8), ("dense_int16", 8), ("conv_channels", 8), ("conv_filters", 8)])
def test_sharded_equals_unsharded(case, world, tmp_path, oracle_mod):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    res = [np.load(tmp_path / f"{case}_{r}.npz") for r in range(world)]
    for k in res[0].files:
        for r in range(1, world):
            assert np.array_equal(res[0][k], res[r][k]), f"ranks 0 and {r} disagree on {k}"
    # unsharded reference: same stand-ins, no process group
    sys.path.insert(0, ROOT)
    from quantized_neural_networks_amd import layer
    hip, keep = _install_standins()
    try:
        r = np.random.default_rng(7)
        if case.startswith("dense"):
            N, m, C = 24, 40, 7
            W = (r.standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
            G = r.standard_normal((N, m))
            X = np.maximum(G, 0).astype(np.float32)
            Xq = np.maximum(G + 0.1 * r.standard_normal((N, m)), 0).astype(np.float32)
            if case == "dense_device":
                out = layer.quantize_dense_layer(torch.from_numpy(W), torch.from_numpy(X), torch.from_numpy(Xq), np.linspace(-1, 1, 4), 2.0, overlap=False)
                # ... which is the host alphabet's result for rad = 2 * median(|W|)
                ref = layer.quantize_dense(torch.from_numpy(W), torch.from_numpy(X), torch.from_numpy(Xq),
                                           np.float64(2.0) * np.float64(np.median(np.abs(W))) * np.linspace(-1, 1, 4))
                assert torch.equal(ref["Q"], out["Q"]) and torch.equal(ref["idx"], out["idx"])
                out = {k: v for k, v in out.items() if k in ("Q", "idx", "resid")}
            else:
                out = layer.quantize_dense(torch.from_numpy(W), torch.from_numpy(X), torch.from_numpy(Xq),
                                           0.3 * np.linspace(-1, 1, _dense_members(case)))
        else:
            Cin = 3 if case == "conv_channels" else 1
            act = r.random((4, 6, 6, Cin)).astype(np.float32)
            actq = (act + 0.05 * r.random(act.shape)).astype(np.float32)
            W = (r.standard_normal((3, 3, Cin, 5)) / 3).astype(np.float32)
            out = layer.quantize_conv2d(torch.from_numpy(W), torch.from_numpy(act), torch.from_numpy(actq),
                                        0.25 * np.linspace(-1, 1, 3), strides=(1, 1), padding="SAME", rate=(1, 1))
    finally:
        for k, v in keep.items():
            setattr(hip, k, v)
    for k, v in out.items():
        assert np.array_equal(res[0][k], v.numpy()), k
    if case == "dense_int16":
        assert out["idx"].dtype == torch.int16 and res[0]["idx"].dtype == np.int16 and int(out["idx"].max()) > 127
