#!/usr/bin/env python3
"""The reference's MLP experiment flow (scripts/quantize_pretrained_mlp.py:66-153) on the MI355X build.

Same steps and the same calls into the `quantized_network` module -- build the calibration feeder,
`QuantizedNeuralNetwork(...).quantize_network()`, evaluate the quantized net, build the MSQ baseline with
the layer radius `alphabet_scalar * median(|W|)`, collect one metrics row per parameter setting -- but with
what this image has: no TensorFlow and no MNIST files, so the network (the reference's 784-500-300-10 MLP
with BatchNormalization, train_mnist_mlp.py:60-73) is built with the torch-backed Keras shim, weights are
random, data are synthetic, and "accuracy" is agreement with the analog network's own predictions.

    python examples/quantize_mlp.py [--samples 25000] [--scalars 2 3 4]
"""
import argparse
import os
import sys
from collections import namedtuple
from itertools import product
from time import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quantized_network import MNISTSequence, QuantizedNeuralNetwork, msq_quantize  # noqa: E402
from quantized_neural_networks_amd import keras_shim as keras  # noqa: E402

ParamConfig = namedtuple("ParamConfig", "data_set, bits, alphabet_scalar")


# columns of the reference's MNIST metrics file, in its order (quantize_pretrained_mlp.py:119-133); the index is the run's time stamp
METRICS_COLUMNS = ["data_set", "analog_model", "serialized_quantized_model", "q_train_size", "bits", "alphabet_scalar",
                   "analog_test_acc", "sd_test_acc", "msq_test_acc", "quantization_time"]


def build_model(seed=0, widths=(500, 300)):
    model = keras.Sequential(seed=seed)
    model.add(keras.Flatten(input_shape=(28, 28)))
    for width in widths:
        model.add(keras.Dense(width, activation="relu", use_bias=True))
        model.add(keras.BatchNormalization())
    model.add(keras.Dense(10, activation="softmax"))
    return model


def agreement(net, ref_out, x):
    """(argmax agreement with the analog net, relative L2 error of the output probabilities)."""
    pred = net.predict(x, batch_size=1000)
    return (float(np.mean(pred.argmax(1) == ref_out.argmax(1))),
            float(np.linalg.norm(pred - ref_out) / np.linalg.norm(ref_out)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=25000, help="calibration samples (the reference's quant_train_size)")
    ap.add_argument("--scalars", type=float, nargs="+", default=[2, 3, 4])
    ap.add_argument("--save-dir", default=None, help="save every quantized model there (the reference's serialized_models/)")
    ap.add_argument("--csv", default=None, help="append one metrics row per setting there, the reference's schema and "
                                                "append semantics (quantize_pretrained_mlp.py:119-153)")
    ap.add_argument("--widths", type=int, nargs="+", default=[500, 300], help="hidden layer widths")
    args = ap.parse_args()

    rng = np.random.default_rng(0)
    model = build_model(widths=tuple(args.widths))
    X_train = rng.random((args.samples, 28, 28)).astype(np.float32)
    X_test = rng.random((2000, 28, 28)).astype(np.float32)
    y_test = model.predict(X_test, batch_size=1000)                     # analog network's outputs
    y_train = np.zeros((args.samples, 10), dtype=np.float32)

    rows = []
    for idx, params in enumerate(ParamConfig(*c) for c in product(["synthetic-mnist"], [np.log2(3)], args.scalars)):
        get_data = MNISTSequence(X_train, y_train, batch_size=args.samples)
        my_quant_net = QuantizedNeuralNetwork(network=model, batch_size=args.samples, get_data=get_data,
                                              logger=type("Quiet", (), {"info": staticmethod(lambda m: None)})(),
                                              bits=params.bits, alphabet_scalar=params.alphabet_scalar)
        tic = time()
        my_quant_net.quantize_network()
        quantization_time = time() - tic
        q_acc = agreement(my_quant_net.quantized_net, y_test, X_test)
        if args.save_dir:                               # quantize_pretrained_mlp.py:87-95 (save_model of quantized_net)
            os.makedirs(args.save_dir, exist_ok=True)
            keras.save_model(my_quant_net.quantized_net,
                             os.path.join(args.save_dir, f"Quantized_MLP_scaler{params.alphabet_scalar}_bits{params.bits:.3f}"))

        # MSQ baseline: same radius as the corresponding GPFQ layer (quantize_pretrained_mlp.py:97-112)
        MSQ_model = keras.clone_model(model)
        MSQ_model.set_weights(model.get_weights())
        for layer_idx, layer in enumerate(model.layers):
            if layer.__class__.__name__ in ("Dense", "Conv2D"):
                W, b = model.layers[layer_idx].get_weights()
                rad = params.alphabet_scalar * np.median(np.abs(W.flatten()))
                MSQ_model.layers[layer_idx].set_weights([msq_quantize(W, rad * my_quant_net.alphabet), b])
        msq_acc = agreement(MSQ_model, y_test, X_test)
        if args.csv:                                    # one row per setting, header with the first (:138-153)
            import pandas as pd
            stamp = str(pd.Timestamp.now()).replace(" ", "_").replace(":", "").replace(".", "")
            trial_metrics = pd.DataFrame({
                "data_set": params.data_set, "analog_model": "synthetic_mlp",
                "serialized_quantized_model": f"quantized_mnist_scaler{params.alphabet_scalar}_{stamp}",
                "q_train_size": args.samples, "bits": params.bits, "alphabet_scalar": params.alphabet_scalar,
                "analog_test_acc": 1.0, "sd_test_acc": q_acc[0], "msq_test_acc": msq_acc[0],
                "quantization_time": quantization_time}, index=[stamp])
            trial_metrics.to_csv(args.csv, mode="a", header=(idx == 0))
        n_weights = sum(int(np.prod(s)) for s in my_quant_net.layer_dims.values())
        rows.append((params.alphabet_scalar, quantization_time, n_weights / quantization_time, q_acc, msq_acc))

    print(f"{'scalar':>6} {'quant time (s)':>15} {'weights/s':>12} {'GPFQ agree':>11} {'GPFQ rel.err':>13} {'MSQ agree':>10} {'MSQ rel.err':>12}")
    for r in rows:
        print(f"{r[0]:6g} {r[1]:15.3f} {r[2]:12.3e} {r[3][0]:11.4f} {r[3][1]:13.4f} {r[4][0]:10.4f} {r[4][1]:12.4f}")


if __name__ == "__main__":
    main()
