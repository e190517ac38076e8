"""GPU: the remaining layers of BASELINE cfg4 and cfg5 at FULL size, each with asserted oracle samples.

  cfg4  CIFAR10 CNN (train_cifar10_cnn.py:63-86), 5008 columns-worth of images (5000 samples, batch 16: the partial-batch
        layout of scripts/quantized_network.py:467-495), 3 bits, scalar 4: all six 3x3 SAME conv layers
        (3->32 and 32->32 @32x32, 32->64 and 64->64 @16x16, 64->128 and 128->128 @8x8) and the two Dense layers
        (2048->128, 128->10) on rows of 5008 samples.
  cfg5  ResNet50, 4096 images, ternary, scalar 3: the 3x3 layers @28x28x128, @14x14x256, @7x7x512 (the NHWC form's
        128 / 256 / 512-channel shapes) and a strided 1x1 layer (conv3_block1_0_conv: 1x1/2, 256 -> 512 @56x56).

Every layer: >= 4 (channel, filter) pairs / neurons of the same run against the C oracle, bit for bit
(scripts/quantized_network.py:185-233, :91-121); the conv samples also against the streaming kernel on GPU-built patch
matrices, as tests/test_fullsize_configs.py does for conv1 and the 56x56 layer.
"""
import numpy as np
import pytest
import torch

from test_fullsize_configs import _conv_inputs, _conv_sample, _conv_slack_ab, _conv_whole_tensor, _dense_inputs, _dense_oracle_sample

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from quantized_neural_networks_amd import hip as h
    h.load()
    return h


CFG4_CONV = [  # (cin, cout, H = W), train_cifar10_cnn.py:64-78
    (3, 32, 32), (32, 32, 32), (32, 64, 16), (64, 64, 16), (64, 128, 8), (128, 128, 8),
]


@pytest.mark.parametrize("cin,cout,hw", CFG4_CONV)
def test_cfg4_conv_layer_full_size(hip, oracle_mod, cin, cout, hw):
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    act_w, act_q, W = _conv_inputs(5008, hw, hw, cin, cout, 3, dev, 400 + cin + cout + hw)
    if cin == 3:
        act_q = act_w                                   # a first layer: both networks see the data itself (:478-481)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
    kw = dict(strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)
    assert tuple(out["idx"].shape) == (3, 3, cin, cout) and int(out["reruns"]) <= 4
    assert torch.equal(out["idx"], layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)["idx"])         # determinism
    _conv_slack_ab(hip, layer, W, act_w, act_q, alphabet, out, kw)
    pairs = [(0, [0, cout - 1]), (cin - 1, [1, cout // 2])] if cin > 1 else [(0, [0, 1, cout // 2, cout - 1])]
    _conv_sample(hip, oracle_mod, act_w, act_q, W, alphabet, out, 3, 1, "SAME", pairs, host_gib_needed=4)


@pytest.mark.parametrize("cin,cout,hw", CFG4_CONV)
def test_cfg4_conv_layer_whole_tensor(hip, oracle_mod, cin, cout, hw):
    """Round 6: every (channel, filter) pair of each cfg4 conv layer (96 ... 16 384 walks over 0.3 ... 5.1 M columns) -- the Gram path's
    index tensor against the verbatim streaming kernel, channel by channel; 64+ pairs per layer against the C oracle."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    act_w, act_q, W = _conv_inputs(5008, hw, hw, cin, cout, 3, dev, 400 + cin + cout + hw)
    if cin == 3:
        act_q = act_w
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    chans = set(range(cin)) if cin == 3 else {0, cin // 3, cin // 2, cin - 1}
    n = _conv_whole_tensor(hip, oracle_mod, act_w, act_q, W, alphabet, out, 3, 1, "SAME", chans, 22 if cin == 3 else 16)
    assert n == cin * cout


@pytest.mark.parametrize("ch,hw", [(64, 56), (128, 28), (256, 14), (512, 7)])
def test_cfg5_resnet50_conv3x3_whole_tensor(hip, oracle_mod, ch, hw):
    """Round 6: all 4 096 / 16 384 / 65 536 / 262 144 (channel, filter) walks of ResNet50's four 3x3 layer shapes at 4096 images (12.8 M ...
    0.2 M columns) against the streaming kernel; 64 pairs against the oracle."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    act_w, act_q, W = _conv_inputs(4096, hw, hw, ch, ch, 3, dev, 500 + ch)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    _conv_whole_tensor(hip, oracle_mod, act_w, act_q, W, alphabet, out, 3, 1, "SAME", {0, ch // 3, ch // 2, ch - 1}, 16)


def test_cfg5_resnet50_conv1_whole_tensor(hip, oracle_mod):
    """Round 6: all 192 (channel, filter) walks of ResNet50's conv1 (7x7 / 2 VALID on the padded 230 x 230 input, 49 steps over 51.4 M columns
    each) against the streaming kernel on the GPU-built 10 GB patch matrices."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    act_w, act_q, W = _conv_inputs(4096, 230, 230, 3, 64, 7, dev, 51)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(2, 2), padding="VALID", rate=(1, 1), want_resid=False)
    _conv_whole_tensor(hip, oracle_mod, act_w, act_q, W, alphabet, out, 7, 2, "VALID", set(), 0)


@pytest.mark.parametrize("N,C", [(2048, 128), (128, 10)])
def test_cfg4_dense_layer_full_size(hip, oracle_mod, N, C):
    """Dense(2048 -> 128) and Dense(128 -> 10) on rows of 5008 samples (cfg4's layers 19 and 22): every neuron against the
    oracle."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    W, X, Xq = _dense_inputs(N, 5008, C, dev, 430 + C)
    X[:, 4992:] = 0; Xq[:, 4992:] = 0                   # the partial-batch layout leaves the last 16 columns zero (:491-495)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 8), 4)
    out = layer.quantize_dense(W, X, Xq, alphabet)
    again = layer.quantize_dense(W, X, Xq, alphabet)
    assert torch.equal(out["idx"], again["idx"]) and torch.equal(out["resid"], again["resid"])
    _dense_oracle_sample(oracle_mod, W, X, Xq, alphabet, out, [(0, C)])


@pytest.mark.parametrize("ch,hw", [(128, 28), (256, 14), (512, 7)])
def test_cfg5_resnet50_conv3x3_full_size(hip, oracle_mod, ch, hw):
    """ResNet50 conv3_x / conv4_x / conv5_x 3x3 SAME layers at 4096 images: 3.2 M / 0.8 M / 0.2 M columns,
    16 K / 65 K / 262 K (channel, filter) pairs, through the NHWC form."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    act_w, act_q, W = _conv_inputs(4096, hw, hw, ch, ch, 3, dev, 500 + ch)
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    kw = dict(strides=(1, 1), padding="SAME", rate=(1, 1), want_resid=False)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)
    assert int(out["reruns"]) <= 8
    assert torch.equal(out["idx"], layer.quantize_conv2d(W, act_w, act_q, alphabet, **kw)["idx"])         # determinism
    s = 2.0 ** 3
    assert torch.equal(layer.quantize_conv2d(W * s, act_w, act_q, alphabet * s, **kw)["idx"], out["idx"])  # scale
    _conv_slack_ab(hip, layer, W, act_w, act_q, alphabet, out, kw)
    _conv_sample(hip, oracle_mod, act_w, act_q, W, alphabet, out, 3, 1, "SAME",
                 [(0, [0, ch - 1]), (ch // 2 + 1, [3]), (ch - 1, [ch // 3])], host_gib_needed=4)


def test_cfg5_resnet50_conv1x1_stride2_full_size(hip, oracle_mod):
    """ResNet50 conv3_block1_0_conv: 1x1 / 2, 256 -> 512 on 4096 x 56 x 56 x 256 (13 GB per tensor): one-step walks, i.e.
    nearest(alphabet, w) unless the channel is dead on the strided grid (:83-87) -- with a dead channel, a channel alive only off
    the grid, and a channel that wakes up in the last image (beyond the prefix the dead-channel probe reads)."""
    from quantized_neural_networks_amd import layer
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(77)
    act_q = torch.rand((4096, 56, 56, 256), device=dev, generator=g)
    act_w = act_q                                        # (the analog activations do not enter a one-step walk's decision)
    act_q[..., 5] = 0
    act_q[..., 9] = 0; act_q[:, 1::2, :, 9] = 1.0        # alive only off the stride-2 grid: dead for this layer
    act_q[..., 200] = 0; act_q[4095, 54, 54, 200] = 0.25  # wakes up at the very last sampled position
    W = torch.randn((1, 1, 256, 512), device=dev, generator=g) / 16
    alphabet, _ = layer.layer_alphabet(W, np.linspace(-1, 1, 3), 3)
    out = layer.quantize_conv2d(W, act_w, act_q, alphabet, strides=(2, 2), padding="VALID", rate=(1, 1), want_resid=False)
    Q = out["Q"]
    assert float(Q[0, 0, 5].abs().max()) == 0.0 and float(Q[0, 0, 9].abs().max()) == 0.0
    msq, _ = hip.msq_round(W, alphabet)
    live = torch.ones(256, dtype=torch.bool, device=dev); live[5] = live[9] = False
    assert torch.equal(Q[0, 0, live], msq[0, 0, live])                   # every live channel (200 included): plain MSQ (rule (ii))
    _conv_sample(hip, oracle_mod, act_w, act_q, W, alphabet, out, 1, 2, "VALID",
                 [(0, [0, 511]), (5, [7]), (9, [8]), (200, [100, 300]), (255, [256])], host_gib_needed=2)
