#!/usr/bin/env python3
"""Accuracy regression against the reference's published tables.

The reference publishes one row per parameter setting in model_metrics/*.csv (columns of
scripts/quantize_pretrained_mlp.py:119-133 / _cnn.py:124-140: bits, alphabet_scalar, q_train_size, analog_test_acc,
sd_test_acc, msq_test_acc).  Given the data set those runs used and the analog model they started from, this script
repeats every row's quantization on the MI355X path and compares the test accuracy with the published `sd_test_acc`.

    python examples/accuracy_regression.py --published model_metrics/mnist_model_metrics_2020-07-15_121741452415.csv \\
           --dataset mnist.npz --model analog_mnist_mlp.npz [--cnn] [--tolerance 0.02] [--rows 3]

  --dataset   .npz with x_train, y_train, x_test, y_test (the layout of Keras' mnist.npz / a dump of cifar10.load_data());
              pixels are scaled to [0, 1] as the reference's drivers do
  --model     the ANALOG network in keras_shim.save_model format (weights of the reference's trained network copied in)

Neither ships with the reference checkout (its analog CIFAR10 weights blob is missing) nor with this image
(no network): when a path is absent the script says so and exits 0, so it can sit in a pipeline.  The GPFQ result is a
deterministic function of weights, calibration data and alphabet, so with the reference's inputs the accuracies agree
to the last evaluation digit except where a forward pass (GPU matmul vs the reference's CPU TensorFlow) moves a decision;
--tolerance bounds that.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--published", required=True, help="a metrics CSV of the reference (model_metrics/*.csv)")
    ap.add_argument("--dataset", default=None)
    ap.add_argument("--model", default=None)
    ap.add_argument("--cnn", action="store_true", help="QuantizedCNN with batches of 16 (quantize_pretrained_cnn.py) instead of QuantizedNeuralNetwork")
    ap.add_argument("--tolerance", type=float, default=0.02)
    ap.add_argument("--rows", type=int, default=0, help="only the first N rows (0 = all)")
    args = ap.parse_args(argv)

    for what, path in (("published metrics", args.published), ("data set", args.dataset), ("analog model", args.model)):
        if not path or not os.path.exists(path):
            print(f"accuracy regression skipped: no {what} at {path!r}")
            return 0

    import pandas as pd
    from quantized_network import CIFAR10Sequence, MNISTSequence, QuantizedCNN, QuantizedNeuralNetwork
    from quantized_neural_networks_amd import keras_shim as keras

    table = pd.read_csv(args.published, index_col=0)
    if args.rows:
        table = table.iloc[:args.rows]
    with np.load(args.dataset) as z:
        X_train, y_train, X_test, y_test = z["x_train"], z["y_train"], z["x_test"], z["y_test"]
    scale = 255.0 if X_train.dtype == np.uint8 else 1.0
    X_train, X_test = X_train.astype(np.float32) / scale, X_test.astype(np.float32) / scale
    classes = int(max(y_train.max(), y_test.max())) + 1
    onehot = lambda y: np.eye(classes, dtype=np.float32)[np.asarray(y).reshape(-1).astype(np.int64)]
    y_train, y_test = onehot(y_train), onehot(y_test)
    model = keras.load_model(args.model)
    _, analog = model.evaluate(X_test, y_test)
    quiet = type("Quiet", (), {"info": staticmethod(lambda m: None)})()

    worst, lines = 0.0, []
    for stamp, row in table.iterrows():
        n = int(row["q_train_size"])
        if args.cnn:
            q = QuantizedCNN(network=model, batch_size=n, get_data=CIFAR10Sequence(X_train[:n], y_train[:n], batch_size=16), logger=quiet,
                             bits=float(row["bits"]), alphabet_scalar=float(row["alphabet_scalar"]))
        else:
            q = QuantizedNeuralNetwork(network=model, batch_size=n, get_data=MNISTSequence(X_train[:n], y_train[:n], batch_size=n),
                                       logger=quiet, bits=float(row["bits"]), alphabet_scalar=float(row["alphabet_scalar"]))
        q.quantize_network()
        _, acc = q.quantized_net.evaluate(X_test, y_test)
        diff = acc - float(row["sd_test_acc"])
        worst = max(worst, abs(diff))
        lines.append(f"{stamp}  bits {float(row['bits']):.3f} scalar {float(row['alphabet_scalar']):g}: published sd_test_acc "
                     f"{float(row['sd_test_acc']):.4f} (analog {float(row['analog_test_acc']):.4f}), here {acc:.4f} (analog {analog:.4f}), diff {diff:+.4f}")
    print("\n".join(lines))
    ok = worst <= args.tolerance
    print(f"accuracy regression {'passed' if ok else 'FAILED'}: worst |diff| {worst:.4f}, tolerance {args.tolerance}")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
