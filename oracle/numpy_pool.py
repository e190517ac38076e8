"""The reference-shaped CPU baseline -- TEST / BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py).

``python -m oracle.numpy_pool <dir> <workers>``: the NumPy restatement of the reference's neuron loop
(``oracle.neuron_numpy`` = /root/reference/scripts/quantized_network.py:91-121 with every cast explicit) over a
``multiprocessing`` pool, one neuron per task -- the shape of the reference's own fan-out
(``ProcessPoolExecutor`` + ``executor.submit(_quantize_neuron_parallel, W[:, j], ...)``, :549-556), with in-memory
arrays instead of the reference's HDF5 reads (a conservative baseline: BASELINE.md 3).

A process of its own, started by ``bench.py`` as a child: it never touches the GPU, so forking the pool is safe, and the
BLAS threads are pinned to one per worker by the environment the parent passes.  Reads ``W.npy`` ([N][n] Keras layout,
the sample's columns), ``X.npy``, ``Xq.npy``, ``alphabet.npy`` from <dir>; writes ``idx.npy`` ([n][N] int16) and
``result.json`` ({"seconds": wall time of the pool's map, "workers", "neurons"}).
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

_G = {}


def _one(j):
    from oracle import neuron_numpy
    _, idx, u = neuron_numpy(_G["W"][:, j], _G["X"], _G["Xq"], _G["alphabet"])
    return j, idx, float(np.sqrt(np.dot(u, u)))


def main():
    d, workers = sys.argv[1], int(sys.argv[2])
    for k in ("W", "X", "Xq", "alphabet"):
        _G[k] = np.load(os.path.join(d, k + ".npy"))
    n = _G["W"].shape[1]
    idx = np.zeros((n, _G["W"].shape[0]), np.int16)
    resid = np.zeros(n, np.float64)
    with mp.get_context("fork").Pool(workers) as pool:
        pool.map(abs, range(workers))                   # (workers started and warm before the clock)
        t0 = time.perf_counter()
        for j, ix, r in pool.imap_unordered(_one, range(n), chunksize=1):
            idx[j], resid[j] = ix, r
        dt = time.perf_counter() - t0
    np.save(os.path.join(d, "idx.npy"), idx)
    np.save(os.path.join(d, "resid.npy"), resid)
    with open(os.path.join(d, "result.json"), "w") as f:
        json.dump({"seconds": dt, "workers": workers, "neurons": n}, f)


if __name__ == "__main__":
    main()
