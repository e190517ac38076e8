"""GPU: the driver flows (examples/quantize_mlp.py, examples/quantize_cnn.py = the reference's
scripts/quantize_pretrained_mlp.py:66-153 and _cnn.py:66-159 on this image's stand-ins for TensorFlow and the data
sets) executed at small size: they run, append one metrics row per parameter setting with exactly the reference's
columns (:119-133 / :124-140) and its append semantics (header with the first row only, :138-153), and the models they
save load back on the GPU to the same network."""
import csv
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

MLP_COLUMNS = ["data_set", "analog_model", "serialized_quantized_model", "q_train_size", "bits", "alphabet_scalar",
               "analog_test_acc", "sd_test_acc", "msq_test_acc", "quantization_time"]          # quantize_pretrained_mlp.py:119-133
CNN_COLUMNS = ["data_set", "serialized_model", "q_train_size", "ignore_layers", "bits", "alphabet_scalar",
               "analog_test_acc", "sd_test_acc", "msq_test_acc", "quantization_time"]          # quantize_pretrained_cnn.py:124-140


def _run(script, *args):
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    return res.stdout


def _rows(path):
    with open(path, newline="") as f:
        return list(csv.reader(f))


def test_mlp_driver_flow_metrics_rows_and_saved_models(tmp_path):
    out = tmp_path / "mnist_model_metrics.csv"
    save = tmp_path / "quantized_models"
    _run("quantize_mlp.py", "--samples", "640", "--scalars", "2", "3", "--widths", "96", "48", "--csv", str(out), "--save-dir", str(save))
    rows = _rows(out)
    assert rows[0] == [""] + MLP_COLUMNS                       # pandas writes the unnamed index (the time stamp) first
    assert len(rows) == 3 and all(len(r) == len(rows[0]) for r in rows)      # one header, one row per scalar
    assert [float(r[6]) for r in rows[1:]] == [2.0, 3.0] and all(float(r[4]) == 640 for r in rows[1:])
    assert all(0.0 <= float(r[8]) <= 1.0 and float(r[10]) > 0 for r in rows[1:])
    # a second run APPENDS to the same file (mode="a"), header included, as the reference's does
    _run("quantize_mlp.py", "--samples", "640", "--scalars", "4", "--widths", "96", "48", "--csv", str(out))
    rows = _rows(out)
    assert len(rows) == 5 and rows[3] == [""] + MLP_COLUMNS and float(rows[4][6]) == 4.0
    # saved quantized networks load back (GPU) and hold ternary kernels scaled by the layer radius
    import torch
    from quantized_neural_networks_amd import keras_shim
    files = sorted(os.listdir(save))
    assert len(files) == 2
    net = keras_shim.load_model(os.path.join(save, files[0]), device="cuda")
    x = np.random.default_rng(0).random((32, 28, 28)).astype(np.float32)
    y = net.predict_on_batch(x)
    assert y.is_cuda and tuple(y.shape) == (32, 10) and torch.allclose(y.sum(1), torch.ones(32, device=y.device), atol=1e-5)
    dense = [l for l in net.layers if l.__class__.__name__ == "Dense"]
    assert len(dense) == 3 and all(len(np.unique(l.get_weights()[0])) <= 3 for l in dense)


def test_cnn_driver_flow_metrics_rows_and_saved_models(tmp_path):
    out = tmp_path / "cifar10_model_metrics.csv"
    save = tmp_path / "quantized_models"
    _run("quantize_cnn.py", "--samples", "200", "--test-samples", "100", "--bits", "3", "--scalars", "3", "4", "--csv", str(out),
         "--save-dir", str(save))
    rows = _rows(out)
    assert rows[0] == [""] + CNN_COLUMNS and len(rows) == 3
    assert [float(r[6]) for r in rows[1:]] == [3.0, 4.0] and all(r[4] == "[]" and float(r[5]) == 3.0 for r in rows[1:])
    assert all(r[2].startswith("quantized_cifar10_scaler") for r in rows[1:])
    from quantized_neural_networks_amd import keras_shim
    files = sorted(os.listdir(save))
    assert len(files) == 2
    net = keras_shim.load_model(os.path.join(save, files[1]), device="cuda")
    convs = [l for l in net.layers if l.__class__.__name__ == "Conv2D"]
    assert len(convs) == 6 and all(len(np.unique(l.get_weights()[0])) <= 8 for l in convs)     # 3 bits: 8 levels
    y = net.predict_on_batch(np.random.default_rng(1).random((8, 32, 32, 3)).astype(np.float32))
    assert tuple(y.shape) == (8, 10)


def test_accuracy_regression_against_a_published_table(tmp_path):
    """examples/accuracy_regression.py end to end on the GPU: a data set file in Keras' mnist.npz layout, an analog model
    in the shim's save format and a metrics table in the reference's schema; it must reproduce accuracies it published
    itself (GPFQ is deterministic) and flag a table that disagrees."""
    import pandas as pd
    from quantized_neural_networks_amd import keras_shim as K
    from quantized_network import MNISTSequence, QuantizedNeuralNetwork
    r = np.random.default_rng(4)
    model = K.Sequential(seed=4)
    model.add(K.Flatten(input_shape=(8, 8)))
    model.add(K.Dense(64, activation="relu"))
    model.add(K.Dense(10, activation="softmax"))
    x_train = (r.random((600, 8, 8)) * 255).astype(np.uint8)
    x_test = (r.random((300, 8, 8)) * 255).astype(np.uint8)
    y_train = r.integers(0, 10, 600)
    y_test = model.predict(x_test.astype(np.float32) / 255.0).argmax(1)             # labels = the analog net's own answers
    np.savez(tmp_path / "mnist.npz", x_train=x_train, y_train=y_train, x_test=x_test, y_test=y_test)
    K.save_model(model, tmp_path / "analog")
    rows = []
    quiet = type("Quiet", (), {"info": staticmethod(lambda m: None)})()
    Xf, Yf = x_train.astype(np.float32) / 255.0, np.eye(10, dtype=np.float32)[y_train]
    for scalar in (2, 3):
        q = QuantizedNeuralNetwork(network=model, batch_size=400, get_data=MNISTSequence(Xf[:400], Yf[:400], batch_size=400),
                                   logger=quiet, bits=np.log2(3), alphabet_scalar=scalar)
        q.quantize_network()
        _, acc = q.quantized_net.evaluate(x_test.astype(np.float32) / 255.0, np.eye(10, dtype=np.float32)[y_test])
        rows.append(dict(data_set="mnist", analog_model="analog", serialized_quantized_model=f"q{scalar}", q_train_size=400,
                         bits=np.log2(3), alphabet_scalar=scalar, analog_test_acc=1.0, sd_test_acc=acc, msq_test_acc=0.0))
    pd.DataFrame(rows, index=["r0", "r1"]).to_csv(tmp_path / "published.csv")
    args = ["--published", str(tmp_path / "published.csv"), "--dataset", str(tmp_path / "mnist.npz"), "--model", str(tmp_path / "analog.npz")]
    out = _run("accuracy_regression.py", *args, "--tolerance", "0.0")
    assert "accuracy regression passed" in out and out.count("diff +0.0000") == 2
    bad = pd.read_csv(tmp_path / "published.csv", index_col=0)
    bad["sd_test_acc"] = (bad["sd_test_acc"] + 0.5) % 1.0
    bad.to_csv(tmp_path / "published_bad.csv")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "accuracy_regression.py"), "--published",
                          str(tmp_path / "published_bad.csv"), *args[2:]], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 1 and "FAILED" in res.stdout


IMAGENET_COLUMNS = ["data_set", "serialized_model", "quantized_model", "is_quantize_conv2d", "q_train_size", "valid_size", "bits",
                    "alphabet_scalar", "analog_test_top1_acc", "analog_test_top5_acc", "gpfq_test_top1_acc", "gpfq_test_top5_acc",
                    "msq_test_top1_acc", "msq_test_top5_acc", "quantization_time", "np_seed", "tf_seed"]   # quantize_pretrained_imagenet.py:233-252


@pytest.mark.parametrize("model,extra", [("vgg", []), ("resnet50", ["--quantize-conv2d"])])
def test_imagenet_driver_flow_metrics_rows(tmp_path, model, extra):
    """examples/quantize_imagenet.py = scripts/quantize_pretrained_imagenet.py:86-269: .npy images + y_val.npy, np.random.choice
    splits, ImageNetSequence, QuantizedCNN(..., patch_mini_batch_size=1000, is_quantize_conv2d=...), top-1 / top-5 of the analog,
    GPFQ and MSQ networks, the saved model, one metrics row per setting with the reference's 17 columns and append semantics.
    "vgg": Dense layers only, the reference's default; "resnet50": the functional ResNet50 topology with its Conv2D layers."""
    out = tmp_path / "ILSVRC2012_model_metrics.csv"
    save = tmp_path / "quantized_models"
    data = tmp_path / "data"
    log = _run("quantize_imagenet.py", "--model", model, "--image-size", "32", "--classes", "10", "--q-train-size", "48", "--valid-size", "64",
               "--scalars", "2", "3", "--csv", str(out), "--save-dir", str(save), "--data-dir", str(data), *extra)
    rows = _rows(out)
    assert rows[0] == [""] + IMAGENET_COLUMNS
    assert len(rows) == 3 and all(len(r) == len(rows[0]) for r in rows)
    col = {name: i + 1 for i, name in enumerate(IMAGENET_COLUMNS)}
    for r, scalar in zip(rows[1:], (2.0, 3.0)):
        assert float(r[col["alphabet_scalar"]]) == scalar and int(r[col["q_train_size"]]) == 48 and int(r[col["valid_size"]]) == 64
        assert r[col["is_quantize_conv2d"]] == str(bool(extra))
        assert float(r[col["analog_test_top1_acc"]]) == 1.0 and float(r[col["analog_test_top5_acc"]]) == 1.0   # labels = its own top-1
        for name in ("gpfq_test_top1_acc", "gpfq_test_top5_acc", "msq_test_top1_acc", "msq_test_top5_acc"):
            assert 0.0 <= float(r[col[name]]) <= 1.0
        assert float(r[col["gpfq_test_top5_acc"]]) >= float(r[col["gpfq_test_top1_acc"]])
        assert float(r[col["quantization_time"]]) > 0 and r[col["np_seed"]] == "0" and r[col["tf_seed"]] == "0"
        assert r[col["quantized_model"]].startswith(f"quantized_{'vgg16' if model == 'vgg' else 'resnet50'}_scaler")
    assert len(os.listdir(data / "preprocessed_val")) == 48 + 64 + 16 and (data / "y_val.npy").exists()
    assert sorted(os.listdir(save)) == sorted(r[col["quantized_model"]] + ".npz" for r in rows[1:]) or len(os.listdir(save)) == 2
    assert "appended 2 rows" in log
    # the saved network loads back on the GPU with quantized (ternary) kernels in the layers the driver asked for
    from quantized_neural_networks_amd import keras_shim
    net = keras_shim.load_model(os.path.join(save, sorted(os.listdir(save))[0]), device="cuda")
    kinds = ("Dense", "Conv2D") if extra else ("Dense",)
    q_layers = [l for l in net.layers if l.__class__.__name__ in kinds]
    assert q_layers and all(len(np.unique(l.get_weights()[0])) <= 3 for l in q_layers)
