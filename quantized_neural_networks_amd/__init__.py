"""MI355X-native GPFQ (greedy path-following quantization) hot path.

Drop-in for the reference's ``scripts/quantized_network.py`` module surface
(``QuantizedNeuralNetwork`` / ``QuantizedCNN``), with the per-neuron greedy loop running as
hand-written HIP kernels for gfx950 behind a C ABI (``include/gpfq.h``).
"""
__version__ = "0.1.0"
