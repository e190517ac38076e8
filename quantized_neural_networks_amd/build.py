"""Builds libgpfq_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container; the
resulting .so travels to the GPU box with the source tree (it is git-ignored, not gpurun-ignored).
"""
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libgpfq_hip.so")
SOURCES = ["gpfq_capi.hip", "gpfq_onchip.hip", "gpfq_rows.hip", "gpfq_wide.hip", "gpfq_stream.hip", "gpfq_gram.hip", "gpfq_gram_image.hip", "gpfq_gram_conv.hip", "gpfq_gram_mfma.hip",
           "gpfq_misc.hip"]
HEADERS = ["gpfq_device.hpp", "gpfq_launch.hpp", "gpfq_gram_tile.hpp", os.path.join("..", "..", "include", "gpfq.h")]

# -ffp-contract=off: the float32 products/subtraction of the residual update must round
# separately (reference numerics); float64 accumulations use explicit fma().
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-shared", "-std=c++17", "-Wall"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile the HIP library if missing or older than its sources; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
