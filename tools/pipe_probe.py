#!/usr/bin/env python3
"""Pipelined dense kernel (csrc/gpfq_pipe.hip) against the row-group kernel and the oracle: bit-equality of
indices / residual norms / residual vectors, exact-fallback counts and kernel time per variant.

    python tools/pipe_probe.py [N C m bits scalar]      (default: BASELINE cfg2 4096 4096 1024 log2(3) 3)
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quantized_neural_networks_amd import hip, layer  # noqa: E402
import oracle  # noqa: E402


def synth(N, m, C):
    W = (np.random.default_rng(0).standard_normal((N, C)) / np.sqrt(N)).astype(np.float32)
    G = np.random.default_rng(1).standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    Xq = np.maximum(G + 0.1 * np.random.default_rng(2).standard_normal((N, m)), 0).astype(np.float32)
    return W, X, Xq


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in ev]
    return min(ts), float(np.mean(ts))


def main():
    args = sys.argv[1:]
    N, C, m = (int(args[0]), int(args[1]), int(args[2])) if len(args) >= 3 else (4096, 4096, 1024)
    bits = float(args[3]) if len(args) > 3 else float(np.log2(3))
    scalar = float(args[4]) if len(args) > 4 else 3.0
    n_or = int(args[5]) if len(args) > 5 else 64
    dev = torch.device("cuda:0")
    W, X, Xq = synth(N, m, C)
    Wd, Xd, Xqd = (torch.from_numpy(a).to(dev) for a in (W, X, Xq))
    M = int(round(2 ** bits))
    alphabet, rad = layer.layer_alphabet(Wd, np.linspace(-1, 1, M), scalar)
    Wt = Wd.t().contiguous()
    nrm = hip.row_norms(Xqd)
    want_u = C * m * 8 <= (1 << 31)

    def run(**opts):
        for k, v in opts.items():
            hip.set_option(k, v)
        r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm, want_u=want_u, path=hip.GPFQ_PATH_ONCHIP)
        torch.cuda.synchronize()
        return r

    hip.set_option("blk_four_groups", int(os.environ.get("BLK_FOUR", "1")))   # <= 1024 neurons on rows of 769..2048 samples: 4 neurons per workgroup
    hip.set_option("blk_pair_groups", int(os.environ.get("BLK_PAIRS", "1")))  # <= 512 neurons: 2 neurons per workgroup
    hip.set_option("blk_single_groups", int(os.environ.get("BLK_SINGLE", "1")))  # <= 128 neurons: 1 neuron per workgroup
    hip.set_option("blk_quad_groups", int(os.environ.get("BLK_QUAD", "2")))   # <= 2048 neurons on rows of 257..1024 samples: four neuron groups x 1 / 2 neurons per lane (fused matrix form); 1: 129..2048 only; 2: narrower layers too
    hip.set_option("blk_quad_waves", int(os.environ.get("BLK_QUAD_NW", "0")))  # four-group narrow shapes on rows <= 768 samples: 7 / 8 sweep wavefronts (0: by shape)
    hip.set_option("blk_cluster_map", int(os.environ.get("BLK_CLUSTER_MAP", "-1")))
    hip.set_option("blk_cluster_nl", int(os.environ.get("BLK_CLUSTER_NL", "0")))    # cluster form: neurons per lane (0: by width)
    hip.set_option("blk_cluster", int(os.environ.get("BLK_CLUSTER", "1")))     # cluster form: 1 = rows beyond 5120 samples, 0 = off, >= 1024: rows beyond that
    hip.set_option("blk_wide_groups", int(os.environ.get("BLK_WIDE", "1")))   # rows > 1024 samples, > 2048 neurons: 16 neurons per workgroup
    base = run(pipe=0)
    print(f"shape N={N} C={C} m={m} M={M}; row-group kernel fallbacks={hip.exact_fallbacks(base)}")
    if n_or:
        oracle.build()
        t0 = time.perf_counter()
        Qo, io, ro = oracle.layer(W, X, Xq, alphabet, 0, n_or, threads=oracle.num_threads())
        print(f"oracle {n_or} neurons {time.perf_counter() - t0:.1f}s; row-group kernel idx mismatches:",
              int((base["idx"][:n_or].cpu().numpy() != io).sum()))
    tmin, tavg = timed(lambda: hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm, want_values=False, path=hip.GPFQ_PATH_ONCHIP))
    print(f"  old kernel: min {tmin:.3f} ms avg {tavg:.3f} ms (incl. pre-pass launches)")
    variants = [int(v) for v in os.environ.get("PIPE_VARIANTS", "0").split(",")]
    modes = [int(v) for v in os.environ.get("PIPE_MODES", "1,2").split(",")]       # 1: one step per slot, 2: blocks of steps
    sweeps = [int(v) for v in os.environ.get("PIPE_SWEEPS", "11,8").split(",")]  # mode 2: sweep wavefronts of the 16-neuron shapes
    for mode, variant, ts, sw in [(md, v, ts, sw) for md in modes for v in variants for ts in ((0, 2, 1) if md == 1 else (0,))
                                  for sw in (sweeps if md == 2 else (11,))]:
        try:
            r = run(pipe=mode, tile_steps=ts, variant=16 * variant, blk_sweep_waves=sw)
        except hip.GpfqError as e:
            print(f"  pipe ts={ts}: {e}")
            continue
        bad_i = int((r["idx"] != base["idx"]).sum())
        bad_q = int((r["Q"] != base["Q"]).sum())
        rel_r = float(((r["resid"] - base["resid"]).abs() / base["resid"].clamp_min(1e-300)).max())
        bad_u = int((r["u"] != base["u"]).sum()) if want_u else -1
        fb = hip.exact_fallbacks(r)
        if hip.cluster_timeouts(r):
            print("  !! cluster form: an exchange timed out")
        if os.environ.get("GPFQ_DIAG") and mode == 2:
            st = r["workspace"][64:64 + 48 * 8].view(torch.int64).cpu().numpy()
            ns = max(int(st[5]), 1)
            for name, o in (("sweep wave 0 (few pairs)", 0), ("sweep wave 7 (most pairs)", 8)):
                print(f"    {name}: cycles per slot: dma issue {st[o]/ns:.0f}, phase U {st[o+1]/ns:.0f}, phase D+fold {st[o+2]/ns:.0f}, "
                      f"dma wait {st[o+3]/ns:.0f}, barrier {st[o+4]/ns:.0f}, barrier -> next slot {st[o+6]/ns:.0f}  ({ns} slots)")
            print("    slot top -> barrier arrival, sweep wavefronts 0..: " + " ".join(f"{st[32 + w]/ns:.0f}" for w in range(sw)))
            print(f"    decision wave: work {st[16]/ns:.0f} (prologue {st[18]/ns:.0f}, chain + certification {st[19]/ns:.0f}), barrier {st[17]/ns:.0f}, barrier -> next slot {st[20]/ns:.0f} cycles per slot")
        tmin, tavg = timed(lambda: hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm, want_values=False,
                                                        path=hip.GPFQ_PATH_ONCHIP))
        print(f"  pipe mode={mode} variant={variant} ts={ts or 'auto'} sweeps={sw} [{hip.last_dense_kernel()[:30]}]: min {tmin:.3f} ms avg {tavg:.3f} ms  "
              f"mismatch idx={bad_i} Q={bad_q} u={bad_u} max resid rel diff={rel_r:.2e}  exact fallbacks={fb}")
    hip.set_option("pipe", -1); hip.set_option("tile_steps", 0); hip.set_option("variant", 0); hip.set_option("blk_sweep_waves", 0)


if __name__ == "__main__":
    main()
