#!/usr/bin/env python3
"""The reference's ImageNet experiment flow (scripts/quantize_pretrained_imagenet.py:86-269) on the MI355X build.

Same steps and the same calls into the `quantized_network` module: preprocessed validation images as one `.npy` file per
image and the labels in `y_val.npy` (what scripts/preprocess_imagenet.py leaves behind), the random split into
quantization-training / validation / test indices with `np.random.choice` (:113-118), `ImageNetSequence(paths, labels,
batch_size=16, preprocess_func=...)`, `QuantizedCNN(network=model, batch_size=q_train_size, get_data=...,
patch_mini_batch_size=1000, is_quantize_conv2d=...)`, `quantize_network()`, top-1 / top-5 accuracy of the analog, GPFQ and
MSQ networks on the validation split (:74-88, :135-141, :187-191, :226-230), the MSQ baseline with the layer radius
`alphabet_scalar * median(|W|)` (:201-221), the quantized model saved under its time-stamped name (:182-185), and one
metrics row per parameter setting with the reference's 17 columns, appended to a CSV with the header on the first row only
(:233-252, :283-291).

What this image lacks is replaced and said so: no TensorFlow and no ILSVRC2012 files, so the network is the torch-backed Keras
shim's ResNet50 (Keras' topology: 53 Conv2D + 1 Dense, skip connections) or a small VGG-style stack with RANDOM weights, the
"preprocessed images" are synthetic arrays written to --data-dir, and the labels are the analog network's own top-1
predictions on them (so the analog accuracies are 1 by construction and the other columns measure agreement with it).

    python examples/quantize_imagenet.py [--model resnet50|vgg] [--image-size 64] [--q-train-size 64] [--valid-size 128]
                                         [--quantize-conv2d] [--bits 1.585] [--scalars 2 3] [--csv out.csv] [--save-dir DIR]
"""
import argparse
import os
import sys
import tempfile
from collections import namedtuple
from glob import glob
from itertools import product
from pathlib import Path
from time import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from quantized_network import ImageNetSequence, QuantizedCNN, msq_quantize  # noqa: E402
from quantized_neural_networks_amd import keras_shim as keras  # noqa: E402

ParamConfig = namedtuple("ParamConfig", "pretrained_model, preprocess_func, data_set, q_train_size, bits, alphabet_scalar, "
                                        "valid_size, is_quantize_conv2d")            # quantize_pretrained_imagenet.py:66-69
# columns of the reference's ImageNet metrics file, in its order (:233-252); the index is the run's time stamp
METRICS_COLUMNS = ["data_set", "serialized_model", "quantized_model", "is_quantize_conv2d", "q_train_size", "valid_size", "bits",
                   "alphabet_scalar", "analog_test_top1_acc", "analog_test_top5_acc", "gpfq_test_top1_acc", "gpfq_test_top5_acc",
                   "msq_test_top1_acc", "msq_test_top5_acc", "quantization_time", "np_seed", "tf_seed"]
np_seed = 0
tf_seed = 0


def top_k_accuracy(y_true, y_pred, k=1):
    """One-hot y_true, scores y_pred (the reference's NumPy branch, :86-88)."""
    argsorted_y = np.argsort(y_pred)[:, -k:]
    return float(np.any(argsorted_y.T == y_true.argmax(axis=1), axis=0).mean())


def to_categorical(y, num_classes):
    out = np.zeros((len(y), num_classes), dtype=np.float32)
    out[np.arange(len(y)), y] = 1.0
    return out


def resnet50(image_size, classes):
    return keras.ResNet50(input_shape=(image_size, image_size, 3), classes=classes, seed=1), "resnet50"


def vgg(image_size, classes):
    """VGG16's shape in small: conv stacks, three Dense layers on top (the reference quantizes only those by default)."""
    K = keras
    net = K.Sequential(seed=1)
    net.add(K.Conv2D(16, (3, 3), activation="relu", padding="same", input_shape=(image_size, image_size, 3)))
    net.add(K.Conv2D(16, (3, 3), activation="relu", padding="same"))
    net.add(K.MaxPooling2D((2, 2)))
    net.add(K.Conv2D(32, (3, 3), activation="relu", padding="same"))
    net.add(K.Conv2D(32, (3, 3), activation="relu", padding="same"))
    net.add(K.MaxPooling2D((2, 2)))
    net.add(K.Flatten())
    net.add(K.Dense(256, activation="relu"))
    net.add(K.Dense(256, activation="relu"))
    net.add(K.Dense(classes, activation="softmax"))
    return net, "vgg16"


def predict(net, generator):
    """model.predict(generator): every batch of the Sequence through the network."""
    return np.concatenate([net.predict(generator[b][0], batch_size=64) for b in range(len(generator))])


def write_synthetic_dataset(data_dir, n_images, image_size, net, classes, rng):
    """What preprocess_imagenet.py leaves behind: preprocessed_val/*.npy and y_val.npy (labels: the analog net's own top-1)."""
    proc = Path(data_dir) / "preprocessed_val"
    proc.mkdir(parents=True, exist_ok=True)
    imgs = rng.random((n_images, image_size, image_size, 3)).astype(np.float32) * 255.0
    for i, im in enumerate(imgs):
        np.save(proc / f"ILSVRC2012_val_{i:08d}.npy", im)
    y = net.predict(np.stack([preprocess(im) for im in imgs]), batch_size=64).argmax(1)
    np.save(Path(data_dir) / "y_val.npy", y.astype(np.int64))
    return proc


def preprocess(x):
    """Stand-in for keras.applications.*.preprocess_input: zero-centre each colour channel (caffe mode without the BGR swap)."""
    return (np.asarray(x, dtype=np.float32) - np.array([103.939, 116.779, 123.68], dtype=np.float32)) / 64.0


def quantize_network(parameters, dir_data, dir_processed_images, logger, save_dir, classes):
    import pandas as pd
    model, model_name_analog = parameters.pretrained_model
    np.random.seed(np_seed)                                               # the seeds for splitting training and testing (:104-106)

    image_paths = np.array(sorted(glob(str(dir_processed_images / "*.npy"))))     # labels follow the sorted paths (:108-111)
    num_images = len(image_paths)
    y = np.load(str(dir_data / "y_val.npy"))
    train_idxs = np.random.choice(range(num_images), size=parameters.q_train_size, replace=False)
    valid_test_idxs = list(set(range(num_images)).difference(set(train_idxs)))
    valid_idxs = np.random.choice(valid_test_idxs, size=parameters.valid_size, replace=False)
    train_paths, valid_paths = image_paths[train_idxs], image_paths[valid_idxs]
    y_train, y_valid = to_categorical(y[train_idxs], classes), to_categorical(y[valid_idxs], classes)

    valid_generator = ImageNetSequence(valid_paths, y_valid, batch_size=16, preprocess_func=parameters.preprocess_func)
    y_valid_pred_analog = predict(model, valid_generator)
    top1_analog = top_k_accuracy(y_valid, y_valid_pred_analog, k=1)
    top5_analog = top_k_accuracy(y_valid, y_valid_pred_analog, k=5)

    quantization_train_generator = ImageNetSequence(train_paths, y_train, batch_size=16, preprocess_func=parameters.preprocess_func)
    my_quant_net = QuantizedCNN(network=model, batch_size=parameters.q_train_size, get_data=quantization_train_generator,
                                logger=logger, bits=parameters.bits, alphabet_scalar=parameters.alphabet_scalar,
                                patch_mini_batch_size=1000, is_quantize_conv2d=parameters.is_quantize_conv2d)
    tic = time()
    my_quant_net.quantize_network()
    quantization_time = time() - tic

    model_timestamp = str(pd.Timestamp.now()).replace(" ", "_").replace(":", "").replace(".", "")
    model_name = f"quantized_{model_name_analog}_scaler{parameters.alphabet_scalar}_{parameters.bits}bits_{model_timestamp}".replace(".", "")
    if save_dir:
        os.makedirs(save_dir, exist_ok=True)
        keras.save_model(my_quant_net.quantized_net, os.path.join(save_dir, model_name))

    y_valid_pred_gpfq = predict(my_quant_net.quantized_net, valid_generator)
    top1_gpfq = top_k_accuracy(y_valid, y_valid_pred_gpfq, k=1)
    top5_gpfq = top_k_accuracy(y_valid, y_valid_pred_gpfq, k=5)

    # MSQ net: the same radius as the corresponding layer of the GPFQ network (:201-221)
    MSQ_model = keras.clone_model(model)
    MSQ_model.set_weights(model.get_weights())
    for layer_idx, layer in enumerate(model.layers):
        if layer.__class__.__name__ == "Dense" or (parameters.is_quantize_conv2d and layer.__class__.__name__ == "Conv2D"):
            ws = layer.get_weights()
            W = ws[0]
            rad = parameters.alphabet_scalar * np.median(np.abs(W.flatten()))
            Q = msq_quantize(W, rad * my_quant_net.alphabet)
            MSQ_model.layers[layer_idx].set_weights([Q] + list(ws[1:]))
    y_valid_pred_msq = predict(MSQ_model, valid_generator)
    top1_msq = top_k_accuracy(y_valid, y_valid_pred_msq, k=1)
    top5_msq = top_k_accuracy(y_valid, y_valid_pred_msq, k=5)

    return pd.DataFrame({
        "data_set": parameters.data_set, "serialized_model": model_name_analog, "quantized_model": model_name,
        "is_quantize_conv2d": parameters.is_quantize_conv2d, "q_train_size": parameters.q_train_size,
        "valid_size": parameters.valid_size, "bits": parameters.bits, "alphabet_scalar": parameters.alphabet_scalar,
        "analog_test_top1_acc": top1_analog, "analog_test_top5_acc": top5_analog,
        "gpfq_test_top1_acc": top1_gpfq, "gpfq_test_top5_acc": top5_gpfq,
        "msq_test_top1_acc": top1_msq, "msq_test_top5_acc": top5_msq,
        "quantization_time": quantization_time, "np_seed": np_seed, "tf_seed": tf_seed}, index=[model_timestamp])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", choices=["resnet50", "vgg"], default="vgg")
    ap.add_argument("--image-size", type=int, default=64)
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--q-train-size", type=int, default=64, help="reference: 1500")
    ap.add_argument("--valid-size", type=int, default=128, help="reference: 20000")
    ap.add_argument("--quantize-conv2d", action="store_true", help="is_quantize_conv2d (reference default: Dense layers only)")
    ap.add_argument("--bits", type=float, nargs="+", default=[np.log2(3)])
    ap.add_argument("--scalars", type=float, nargs="+", default=[2.0])
    ap.add_argument("--csv", default=None, help="append the metrics rows here (reference schema and append semantics)")
    ap.add_argument("--save-dir", default=None, help="save every quantized model there (:182-185)")
    ap.add_argument("--data-dir", default=None, help="where the synthetic preprocessed_val/*.npy and y_val.npy go (default: a temp dir)")
    args = ap.parse_args()

    rng = np.random.default_rng(0)
    builder = resnet50 if args.model == "resnet50" else vgg
    model = builder(args.image_size, args.classes)
    tmp = None
    if args.data_dir is None:
        tmp = tempfile.TemporaryDirectory()
        args.data_dir = tmp.name
    dir_data = Path(args.data_dir)
    dir_processed = write_synthetic_dataset(dir_data, args.q_train_size + args.valid_size + 16, args.image_size, model[0], args.classes, rng)
    quiet = type("Quiet", (), {"info": staticmethod(lambda m: None)})()

    grid = product([model], [preprocess], ["ILSVRC2012-synthetic"], [args.q_train_size], args.bits, args.scalars, [args.valid_size],
                   [bool(args.quantize_conv2d)])
    n_rows = 0
    for idx, params in enumerate(ParamConfig(*c) for c in grid):
        trial_metrics = quantize_network(params, dir_data, dir_processed, quiet, args.save_dir, args.classes)
        if args.csv:                                    # header with the first row only, rows appended (:283-291)
            trial_metrics.to_csv(args.csv, mode="a", header=(idx == 0))
        n_rows += 1
        r = trial_metrics.iloc[0]
        print(f"{r['serialized_model']} bits {params.bits:.3f} scalar {params.alphabet_scalar:g} conv2d {params.is_quantize_conv2d}: "
              f"quantization_time {r['quantization_time']:.3f} s; top-1 / top-5 agreement with the analog net: "
              f"GPFQ {r['gpfq_test_top1_acc']:.3f} / {r['gpfq_test_top5_acc']:.3f}, MSQ {r['msq_test_top1_acc']:.3f} / {r['msq_test_top5_acc']:.3f}",
              flush=True)
    if args.csv:
        print(f"appended {n_rows} rows to {args.csv}")
    if tmp is not None:
        tmp.cleanup()


if __name__ == "__main__":
    main()
