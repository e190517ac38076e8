#!/usr/bin/env python3
"""The reference's CNN experiment flow (scripts/quantize_pretrained_cnn.py:50-160) on the MI355X build.

Same steps and the same calls into the `quantized_network` module -- `CIFAR10Sequence(..., batch_size=16)`,
`QuantizedCNN(...).quantize_network()`, evaluation of the quantized net, the MSQ baseline with the layer radius
`alphabet_scalar * median(|W|)` for Dense and Conv2D layers, one metrics row per parameter setting written as CSV
with the reference's column schema (model_metrics/cifar10_model_metrics_*.csv) -- but with what this image has:
no TensorFlow and no CIFAR10 files, so the network (train_cifar10_cnn.py:63-86) is built with the torch-backed
Keras shim, its weights are random (BatchNormalization statistics included), data are synthetic, and the three
accuracy columns hold agreement with the analog network's own predictions (analog = 1 by construction).

    python examples/quantize_cnn.py [--samples 5000] [--bits 1.585 3] [--scalars 2 3 4] [--csv out.csv]
"""
import argparse
import os
import sys
from collections import namedtuple
from itertools import product
from time import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quantized_network import CIFAR10Sequence, QuantizedCNN, msq_quantize  # noqa: E402
from quantized_neural_networks_amd import keras_shim as keras  # noqa: E402

ParamConfig = namedtuple("ParamConfig", "pretrained_model, data_set, q_train_size, ignore_layers, bits, alphabet_scalar")
# columns of the reference's CIFAR10 metrics file, in its order (quantize_pretrained_cnn.py:124-140); the index is the run's time stamp
METRICS_COLUMNS = ["data_set", "serialized_model", "q_train_size", "ignore_layers", "bits", "alphabet_scalar",
                   "analog_test_acc", "sd_test_acc", "msq_test_acc", "quantization_time"]


def build_model(seed=0):
    K = keras
    model = K.Sequential(seed=seed)
    model.add(K.Conv2D(32, (3, 3), activation="relu", padding="same", input_shape=(32, 32, 3)))
    model.add(K.BatchNormalization())
    model.add(K.Conv2D(32, (3, 3), activation="relu", padding="same"))
    model.add(K.BatchNormalization())
    model.add(K.MaxPooling2D((2, 2)))
    model.add(K.Dropout(0.2))
    for width, drop in ((64, 0.3), (128, 0.4)):
        model.add(K.Conv2D(width, (3, 3), activation="relu", padding="same"))
        model.add(K.BatchNormalization())
        model.add(K.Conv2D(width, (3, 3), activation="relu", padding="same"))
        model.add(K.BatchNormalization())
        model.add(K.MaxPooling2D((2, 2)))
        model.add(K.Dropout(drop))
    model.add(K.Flatten())
    model.add(K.Dense(128, activation="relu"))
    model.add(K.BatchNormalization())
    model.add(K.Dropout(0.5))
    model.add(K.Dense(10, activation="softmax"))
    g = np.random.default_rng(seed + 1)                    # "trained" BatchNormalization statistics
    for layer in model.layers:
        if layer.__class__.__name__ == "BatchNormalization":
            c = layer.get_weights()[0].shape[0]
            layer.set_weights([g.uniform(0.5, 1.5, c), g.normal(0, 0.3, c), g.normal(0.3, 0.3, c), g.uniform(0.5, 1.5, c)])
    return model


def agreement(net, ref_out, x):
    pred = net.predict(x, batch_size=500)
    return float(np.mean(pred.argmax(1) == ref_out.argmax(1)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=5000, help="q_train_size (reference: 5000)")
    ap.add_argument("--bits", type=float, nargs="+", default=[np.log2(3)])
    ap.add_argument("--scalars", type=float, nargs="+", default=[2, 3, 4])
    ap.add_argument("--csv", default=None, help="append the metrics rows here (reference schema and append semantics)")
    ap.add_argument("--save-dir", default=None, help="save every quantized model there (quantize_pretrained_cnn.py:97-100)")
    ap.add_argument("--test-samples", type=int, default=2000)
    args = ap.parse_args()

    rng = np.random.default_rng(0)
    model = build_model()
    X_train = rng.random((args.samples, 32, 32, 3)).astype(np.float32)
    X_test = rng.random((args.test_samples, 32, 32, 3)).astype(np.float32)
    y_train = np.zeros((args.samples, 10), dtype=np.float32)
    y_test = model.predict(X_test, batch_size=500)                       # analog network's outputs
    quiet = type("Quiet", (), {"info": staticmethod(lambda m: None)})()

    rows = []
    grid = product(["synthetic_cnn"], ["synthetic-cifar10"], [args.samples], [[]], args.bits, args.scalars)
    for idx, params in enumerate(ParamConfig(*c) for c in grid):
        get_data = CIFAR10Sequence(X_train[0:params.q_train_size], y_train[0:params.q_train_size], batch_size=16)
        my_quant_net = QuantizedCNN(network=model, batch_size=params.q_train_size, get_data=get_data, logger=quiet,
                                    bits=params.bits, alphabet_scalar=params.alphabet_scalar)
        tic = time()
        my_quant_net.quantize_network()
        quantization_time = time() - tic
        q_accuracy = agreement(my_quant_net.quantized_net, y_test, X_test)

        # MSQ net: same radius as the corresponding layer of the greedy network (quantize_pretrained_cnn.py:104-117)
        MSQ_model = keras.clone_model(model)
        MSQ_model.set_weights(model.get_weights())
        for layer_idx, layer in enumerate(model.layers):
            if layer.__class__.__name__ in ("Dense", "Conv2D"):
                W, b = layer.get_weights()
                rad = params.alphabet_scalar * np.median(np.abs(W.flatten()))
                MSQ_model.layers[layer_idx].set_weights([msq_quantize(W, rad * my_quant_net.alphabet), b])
        MSQ_accuracy = agreement(MSQ_model, y_test, X_test)

        import pandas as pd
        stamp = str(pd.Timestamp.now()).replace(" ", "_").replace(":", "").replace(".", "")
        model_name = f"quantized_cifar10_scaler{params.alphabet_scalar}_{params.bits}bits_{stamp}"
        if args.save_dir:
            os.makedirs(args.save_dir, exist_ok=True)
            keras.save_model(my_quant_net.quantized_net, os.path.join(args.save_dir, model_name))
        trial_metrics = pd.DataFrame({
            "data_set": params.data_set, "serialized_model": model_name, "q_train_size": params.q_train_size,
            "ignore_layers": [params.ignore_layers], "bits": params.bits, "alphabet_scalar": params.alphabet_scalar,
            "analog_test_acc": 1.0, "sd_test_acc": q_accuracy, "msq_test_acc": MSQ_accuracy,
            "quantization_time": quantization_time}, index=[stamp])
        if args.csv:                                    # header with the first row only, rows appended (:146-159)
            trial_metrics.to_csv(args.csv, mode="a", header=(idx == 0))
        rows.append(trial_metrics)
        print(f"bits {params.bits:.3f} scalar {params.alphabet_scalar:g}: quantization_time {quantization_time:.3f} s, "
              f"agreement with the analog net: GPFQ {q_accuracy:.4f}, MSQ {MSQ_accuracy:.4f}", flush=True)

    if args.csv:
        print(f"appended {len(rows)} rows to {args.csv}")


if __name__ == "__main__":
    main()
