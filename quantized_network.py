"""`from quantized_network import ...` -- the module name the reference's drivers import
(scripts/quantize_pretrained_mlp.py:16, _cnn.py:10, _imagenet.py:15).  Re-exports the MI355X build."""
from quantized_neural_networks_amd.quantized_network import (  # noqa: F401
    CIFAR10Sequence,
    ImageNetSequence,
    MNISTSequence,
    QuantizedCNN,
    QuantizedNeuralNetwork,
    SegmentedData,
    _bit_round_parallel,
    msq_quantize,
)
