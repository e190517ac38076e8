"""GPU: size-independent properties at BASELINE.json's full sizes, where the CPU oracle is too slow to check
every unit (the oracle checks a sample of the same runs in bench.py / tools/bench_configs.py):

* fixed point      -- weights that are alphabet members, with identical analog and quantized data, quantize
                      to themselves (u stays 0, every step is rule (ii));
* scale            -- scaling weights and alphabet by a power of two changes nothing (exact in binary fp);
* independence     -- units (neurons / (channel, filter) pairs) do not interact: quantizing a permuted or a
                      split set of units gives the permuted / concatenated result bit for bit;
* determinism      -- two runs give identical bits.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from quantized_neural_networks_amd import hip as h
    h.load()
    return h


def _dense_cfg2(dev):
    N = C = 4096
    m = 1024
    g = torch.Generator(device=dev).manual_seed(7)
    W = torch.randn((N, C), device=dev, generator=g) / np.sqrt(N)
    G = torch.randn((N, m), device=dev, generator=g)
    X = torch.relu(G)
    Xq = torch.relu(G + 0.1 * torch.randn((N, m), device=dev, generator=g))
    return W, X, Xq


def test_dense_cfg2_properties(hip):
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    W, X, Xq = _dense_cfg2(dev)
    unit = np.linspace(-1, 1, 3)
    alphabet, rad = layer.layer_alphabet(W, unit, 3)
    base = layer.quantize_dense(W, X, Xq, alphabet)
    again = layer.quantize_dense(W, X, Xq, alphabet)
    assert torch.equal(base["idx"], again["idx"]) and torch.equal(base["resid"], again["resid"])          # determinism
    assert set(np.unique(base["idx"].cpu().numpy()).tolist()) <= {0, 1, 2}
    assert torch.equal(base["Q"], torch.from_numpy(alphabet.astype(np.float32)).to(dev)[base["idx"].long()])

    # scale: 2^-7 on weights and alphabet
    s = 2.0 ** -7
    scaled = layer.quantize_dense(W * s, X, Xq, alphabet * s)
    assert torch.equal(scaled["idx"], base["idx"])
    assert torch.equal(scaled["resid"], base["resid"] * s)

    # independence: a permutation of the neurons, and the two halves quantized separately
    perm = torch.randperm(W.shape[1], device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    permuted = layer.quantize_dense(W[:, perm].contiguous(), X, Xq, alphabet)
    assert torch.equal(permuted["idx"], base["idx"][:, perm]) and torch.equal(permuted["resid"], base["resid"][perm])
    half = W.shape[1] // 2
    lo = layer.quantize_dense(W[:, :half].contiguous(), X, Xq, alphabet)
    hi = layer.quantize_dense(W[:, half:].contiguous(), X, Xq, alphabet)
    assert torch.equal(torch.cat([lo["idx"], hi["idx"]], dim=1), base["idx"])

    # fixed point: alphabet-valued weights, identical data for both networks
    a32 = torch.from_numpy(alphabet.astype(np.float32)).to(dev)
    k = torch.randint(0, 3, W.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(2))
    fixed = layer.quantize_dense(a32[k].contiguous(), X, X, alphabet)
    assert torch.equal(fixed["idx"].long(), k)
    assert float(fixed["resid"].abs().max()) == 0.0


def test_conv_cfg4_properties(hip):
    """cfg4's 32 -> 32 layer at 32 x 32, 5008 columns-worth of images, 3 bits (1024 (channel, filter) pairs)."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(3)
    n, hw, cin, cout = 5008, 32, 32, 32
    act_w = torch.rand((n, hw, hw, cin), device=dev, generator=g)
    act_q = torch.relu(act_w + 0.05 * torch.randn((n, hw, hw, cin), device=dev, generator=g))
    W = torch.randn((3, 3, cin, cout), device=dev, generator=g) / 3
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
    kw = dict(strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    base = layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)
    assert torch.equal(base["idx"], layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)["idx"])        # determinism
    assert int(base["reruns"]) <= 2

    s = 2.0 ** 5
    assert torch.equal(layer.quantize_conv2d(W * s, act_w, act_q, alphabet * s, **kw)["idx"], base["idx"])   # scale

    # independence: permuting input channels (of kernel and both activations) permutes the result's channel axis,
    # permuting filters permutes its filter axis
    pc = torch.randperm(cin, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    pf = torch.randperm(cout, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    perm = layer.quantize_conv2d(W[:, :, pc][:, :, :, pf].contiguous(), act_w[..., pc].contiguous(),
                                 act_q[..., pc].contiguous(), alphabet, **kw)
    assert torch.equal(perm["idx"], base["idx"][:, :, pc][:, :, :, pf])

    # fixed point
    a32 = torch.from_numpy(alphabet.astype(np.float32)).to(dev)
    k = torch.randint(0, 8, W.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(6))
    fixed = layer.quantize_conv2d(a32[k].contiguous(), act_w, act_w, alphabet, **kw)
    assert torch.equal(fixed["idx"].long(), k)
