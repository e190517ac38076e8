#!/usr/bin/env python3
"""Headline benchmark: weights quantized per second for a whole Dense layer (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY 8d "cfg2"): one Dense(4096 -> 4096) layer, 1024
calibration samples, ternary alphabet (M = 3), alphabet_scalar = 3, synthetic ReLU-like activations.
One "step" = the whole layer driver on inputs already resident in HBM: alphabet radius (median of
|W|), the row-norm pre-pass, the greedy kernel for every neuron, the all-gather (N > 1) and the
transposes back to the Keras kernel layout.

N > 1: one process per GPU.  Default = BASELINE.json's north-star workload: the FIXED 4096 x 4096 layer, its
neurons split over the N ranks ("strong" scaling) and one RCCL all-gather of the packed indices.  Below about
1000 neurons per GPU the walk is latency-bound (N sequential steps of ~1 us), so this curve flattens by
construction; `--scaling weak` gives every rank its own 4096-neuron shard of a Dense(4096 -> 4096*N) layer.

Prints ONE JSON line on rank 0 (contract in the task statement), with
  `roofline`      the dominant kernel against the resource that binds it -- FP64-rate vector issue: algorithmic flops
                  (6 m per weight) / the kernel's HIP-event time / the FP64 vector peak (spec) -- plus,
                  labelled as secondary, the 8d HBM-equivalent figure, the compulsory traffic and the PMC-measured one;
  `cpu_baseline`  the C oracle port on all host cores over a bounded sample of the same layer;
  `cpu_baseline_numpy`  the reference-shaped one: the NumPy restatement over a process pool (a child process);
  `long_rows`     a secondary record outside the timed region: the same layer on --long-rows (8192) calibration samples, i.e. through
                  the block kernel's cluster form -- kernel time by the library's own events and the same flop roofline.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector, spec (= half of the 157.3 TFLOP/s FP32 vector peak of MI355X_MICROARCH.md)
# what the vector units were measured to issue on this chip (tools/ubench/issue_cycles.hip): a float64-rate
# wavefront instruction every 5.63 shader cycles per SIMD -> 64 lanes * 2 flop / 5.63 * 1024 SIMDs * 2.4 GHz
FP64_MEASURED_ISSUE_TFLOPS = 64 * 2 / 5.63 * 1024 * 2.4e9 / 1e12


def synthetic_layer(N, m, C, c_lo, c_hi, seed=0):
    """SURVEY 8d recipe.  W columns [c_lo, c_hi) are generated per 4096-column block so a shard
    does not need the whole kernel on the host; block 0 is exactly rng(0) of the 1-GPU case."""
    rng_g = np.random.default_rng(seed + 1)
    G = rng_g.standard_normal((N, m))
    X = np.maximum(G, 0).astype(np.float32)
    Xq = np.maximum(G + 0.1 * np.random.default_rng(seed + 2).standard_normal((N, m)), 0).astype(np.float32)
    return X, Xq


def weight_block(N, block, width, seed=0):
    return (np.random.default_rng(seed + 1000003 * block).standard_normal((N, width)) / np.sqrt(N)).astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clock-warmup", dest="clock_warmup", type=int, default=10,
                    help="untimed steps BEFORE the --warmup steps: after an idle period this chip runs the same kernel 14 %% slower and comes back over "
                         "~8 launches (3.28, 3.24, 3.15, 3.08, 3.03, 2.96, 2.93, 2.86 ms, then flat: profiles/r06/clock_ramp.txt) -- the timed region "
                         "should see the steady clock whatever W the caller picks; 0 = none.  Reported in the line as clock_warmup_steps")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="N > 1: strong = the fixed --n x --c layer split over the ranks (north star); weak = --c neurons per rank")
    ap.add_argument("--n", "--fan-in", dest="n", type=int, default=4096, help="fan-in N (rows of W)")
    ap.add_argument("--c", "--neurons", dest="c", type=int, default=4096, help="neurons: total (strong, default) / per GPU (weak)")
    ap.add_argument("--m", "--samples", dest="m", type=int, default=1024, help="calibration samples")
    ap.add_argument("--bits", type=float, default=float(np.log2(3)))
    ap.add_argument("--alphabet-scalar", type=float, default=3.0)
    ap.add_argument("--cpu-sample", type=int, default=512, help="neurons timed on the host cores (0 = skip)")
    ap.add_argument("--long-rows", type=int, default=8192,
                    help="also time the same layer on this many calibration samples (a secondary record: the block kernel's cluster form); 0 = skip")
    ap.add_argument("--overlap", dest="overlap", action="store_true",
                    help="(the default; N > 1: where the median is not sharded over the ranks) the median of |W| + alphabet on a second HIP stream -- they depend on the analog kernel alone, which is "
                         "complete long before the step, as a trained network's is: no fork wait --, row norms + record pre-pass on this one "
                         "(gpfq_dense_layer_prepare), ONE join, then the alphabet-dependent rest (gpfq_dense_layer_run): layer.quantize_dense_layer("
                         "overlap=True, kernel_ready=True).  Same box: 2.97 -> 2.94 ms per step, kernel + 0.10 -> kernel + 0.07 (profiles/r06/overlap_ab.txt)")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false", help="one stream: median, alphabet, row norms, record pre-pass, kernel in sequence")
    ap.set_defaults(overlap=True)
    ap.add_argument("--numpy-sample", type=int, default=-1,
                    help="neurons of the NumPy process-pool baseline (the reference-shaped one); -1 = 2 x host cores, 0 = skip")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible (the HIP path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # one process per GPU; GPFQ_DIST_BACKEND=gloo lets the N > 1 code path be exercised with several ranks on a
    # single GPU (functional check only -- the collectives then go through host memory)
    backend = os.environ.get("GPFQ_DIST_BACKEND", "nccl")
    local_dev = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)

    import torch.distributed as dist
    group = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
        group = dist.group.WORLD           # the layer drivers shard only over an EXPLICIT group

    from quantized_neural_networks_amd import hip, layer

    N, m = args.n, args.m
    C_total = args.c * world if args.scaling == "weak" else args.c
    M = int(round(2 ** args.bits))
    unit_alphabet = np.linspace(-1, 1, num=M)

    # ---- synthetic inputs, resident in HBM before the timed region -------------------------
    X, Xq = synthetic_layer(N, m, C_total, 0, 0)
    nblocks = -(-C_total // args.c)
    W = np.concatenate([weight_block(N, b, min(args.c, C_total - b * args.c)) for b in range(nblocks)], axis=1)
    Xd, Xqd = torch.from_numpy(X).to(dev), torch.from_numpy(Xq).to(dev)
    Wd = torch.from_numpy(W).to(dev)

    kernel_ms = []
    kname = [""]
    knames = [""] * args.steps                 # per timed step: the inner events are only recorded by the block-pipelined family
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    ev_ag = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]   # N > 1: the all-gather
    # the recurrence kernel alone: events the C library launches that kernel with (gpfq_set_main_kernel_events -> hipExtLaunchKernelGGL: the
    # dispatch's own start and end, nothing extra in the queue); `ev` brackets the library call around it
    ev_k = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for pair in ev_k:                          # (torch creates an event's handle at its first record(): here, not inside a timed step)
        pair[0].record(); pair[1].record()

    st = {"C_total": C_total, "Wd": Wd}      # the layer being stepped (the weak-scaling companion swaps in its own)

    # The layer driver with nothing crossing to the host (round 6): the median of |W| stays a device scalar, one single-thread kernel forms
    # rad * alphabet in device memory (layer.layer_alphabet_device), the block-pipelined kernel reads the Keras kernel itself and (one GPU)
    # writes Q and the indices in the Keras layout: no host wait, no neuron-major copy, no assembly pass inside a step.  Shapes that kernel
    # does not take (a --m below 257, ...) keep round 5's step: host alphabet, neuron-major copy, assembly.
    lo0, hi0 = layer.shard_bounds(C_total, world, rank)
    device_path = hip.dense_layer_supported(N, m, max(hi0 - lo0, 1), unit_alphabet)
    statuses = []

    # A recorded event is a barrier packet: ~6 us of idle queue.  The events around the library CALL (a secondary figure, and the fallback
    # for kernel families the inner events do not cover) are therefore recorded inside the timed region only when they are that fallback;
    # for the block-pipelined family they come from a few extra steps after it (st["call_events"]).
    ev_rec = [False] * args.steps

    def call_events(i_timed):
        return i_timed is not None and (st.get("call_events") or not kname[0].startswith("gpfq_blk_kernel"))

    def step(i_timed=None):
        C_total, Wd = st["C_total"], st["Wd"]
        lo, hi = layer.shard_bounds(C_total, world, rank)
        if not device_path:
            return step_host(i_timed, C_total, Wd, lo, hi)
        # --overlap (default): the layer's two independent halves on two HIP streams, as layer.quantize_dense_layer(overlap=True,
        # kernel_ready=True) runs them: median of |W| + alphabet (kernel only) on the side stream, row norms + record pre-pass
        # (activations only) on this one, one join, then the recurrence.  --no-overlap: one stream.
        # (st["alphabet_pre"], secondary figure only: the alphabet formed before the loop, as _prefetch_medians does for a network)
        ws = None
        # (N > 1: wherever the median is not sharded over the ranks -- kernels below layer._SHARDED_MEDIAN_MIN elements, the north-star layer among them:
        #  every rank selects it itself, no collective on the side stream)
        if args.overlap and st.get("alphabet_pre") is None and (world == 1 or Wd.numel() < layer._SHARDED_MEDIAN_MIN):
            main, side = torch.cuda.current_stream(dev), layer._side_stream(dev)
            with torch.cuda.stream(side):
                dalpha = layer.layer_alphabet_device(Wd, unit_alphabet, args.alphabet_scalar, None)
            dalpha.buf.record_stream(main)
            ws = hip.dense_layer_workspace(N, m, hi - lo, dev)
            if call_events(i_timed):
                ev[i_timed][0].record()      # (before the pre-pass: an event between the pre-pass and the rest would sit on the step's critical path)
            hip.dense_layer_prepare(Xd, Xqd, unit_alphabet, hi - lo, ws)
            main.wait_stream(side)
        else:
            dalpha = st.get("alphabet_pre") or layer.layer_alphabet_device(Wd, unit_alphabet, args.alphabet_scalar, group)   # N > 1: counting sharded over ranks
        nrm = None                            # (the row norms are formed inside the layer call: its launch also zeroes the call's counter block)
        if i_timed is not None:
            hip.set_main_kernel_events(*ev_k[i_timed])
            if ws is None and call_events(i_timed):
                ev[i_timed][0].record()      # same stream the kernel is launched on (torch current stream)
        r = hip.quantize_dense_layer(Xd, Xqd, Wd, dalpha, lo, hi, nrm32=nrm, keras_out=(world == 1), want_values=(world == 1), prepared=ws)
        if i_timed is not None:
            if call_events(i_timed):
                ev[i_timed][1].record()
                ev_rec[i_timed] = True
            hip.set_main_kernel_events(None, None)
            statuses.append(r["workspace"][:16])     # (a view: the deferred status words of every timed step, read after the loop)
        kname[0] = hip.last_dense_kernel()
        if i_timed is not None:
            knames[i_timed] = kname[0]
        if world > 1:
            # one all-gather of the indices (packed to 2 bits per weight for the ternary alphabet), then values + transpose to the Keras layout
            packed, bits = hip.pack_indices(r["idx"], M)
            if i_timed is not None:
                ev_ag[i_timed][0].record()
            gathered = layer.all_gather_units(packed, C_total, group)
            if i_timed is not None:
                ev_ag[i_timed][1].record()
                st["gather_bytes"] = gathered.numel() * gathered.element_size()
            Q, idx = hip.assemble_kernel_device(gathered.contiguous(), dalpha, bits=bits, N=N)
        else:
            Q, idx = r["Q"], r["idx"]
        return Q, idx, r

    def step_host(i_timed, C_total, Wd, lo, hi):
        pre = {}

        def alphabet_free_work():            # queued behind the median kernels, runs while the host waits for the radius
            pre["Wt"] = hip.neuron_major(Wd, lo, hi)
            pre["nrm"] = hip.row_norms(Xqd)

        if st.get("alphabet_pre") is not None:
            alphabet = st["alphabet_pre"]
            alphabet_free_work()
        else:
            alphabet, _ = layer.layer_alphabet(Wd, unit_alphabet, args.alphabet_scalar, group, alphabet_free_work)   # N > 1: counting sharded over ranks
        Wt, nrm = pre["Wt"], pre["nrm"]
        if i_timed is not None:
            hip.set_main_kernel_events(*ev_k[i_timed])
            if call_events(i_timed):
                ev[i_timed][0].record()
        r = hip.quantize_neurons(Xd, Xqd, Wt, alphabet, nrm32=nrm, want_values=False)
        if i_timed is not None:
            if call_events(i_timed):
                ev[i_timed][1].record()
                ev_rec[i_timed] = True
            hip.set_main_kernel_events(None, None)
        kname[0] = hip.last_dense_kernel()
        if i_timed is not None:
            knames[i_timed] = kname[0]
        if world > 1:
            packed, bits = hip.pack_indices(r["idx"], M)
            if i_timed is not None:
                ev_ag[i_timed][0].record()
            gathered = layer.all_gather_units(packed, C_total, group)
            if i_timed is not None:
                ev_ag[i_timed][1].record()
                st["gather_bytes"] = gathered.numel() * gathered.element_size()
            Q, idx = hip.assemble_kernel(gathered.contiguous(), alphabet, bits=bits, N=N)
        else:
            Q, idx = hip.assemble_kernel(r["idx"], alphabet)
        return Q, idx, r

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.clock_warmup + args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        Q, idx, last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    kernel_name = kname[0]
    # (only the block-pipelined family takes the inner events; a step whose kernel is not of that family falls back to ITS call events,
    #  which were then recorded inside the timed region: ADVICE r04)
    kernel_ms = [ev_k[i][0].elapsed_time(ev_k[i][1]) if knames[i].startswith("gpfq_blk_kernel") else ev[i][0].elapsed_time(ev[i][1])
                 for i in range(args.steps)]
    # Secondary figure (never `value`): the same steps with the layer's radius formed BEFORE the loop -- what a layer costs inside
    # QuantizedNeuralNetwork.quantize_network(), which queues the medians of all layers up front (_prefetch_medians)
    ms_prefetched = None
    # deferred status of every timed step (a timed-out cluster exchange, a degenerate device alphabet): read now, after the timed region
    bad_status = int(torch.stack([w.view(torch.int32)[2:4] for w in statuses]).ne(0).sum().item()) if statuses else 0
    if world == 1:
        st["alphabet_pre"] = (layer.layer_alphabet_device(Wd, unit_alphabet, args.alphabet_scalar, None) if device_path
                              else layer.layer_alphabet(Wd, unit_alphabet, args.alphabet_scalar, None)[0])
        for _ in range(max(1, args.clock_warmup)):
            step()
        fence()
        t0p = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        ms_prefetched = (time.perf_counter() - t0p) / args.steps * 1e3
        st["alphabet_pre"] = None
    # N > 1: what the collective saw -- backend, the world size of the group the all-gather ran on, the all-gather's own
    # duration (HIP events around it on its stream; it waits for the slowest rank's kernel, so rank 0's figure includes the
    # skew) and every rank's kernel time, gathered over the same group
    collective = None
    if world > 1:
        ag_ms = [a.elapsed_time(b) for a, b in ev_ag]
        mine = torch.tensor([float(np.mean(kernel_ms)), float(np.mean(ag_ms)), float(np.min(ag_ms))], dtype=torch.float64, device=dev)
        every = torch.empty(dist.get_world_size(group) * 3, dtype=torch.float64, device=dev)   # (concatenated form: gloo takes no other)
        dist.all_gather_into_tensor(every, mine, group=group)
        every = every.cpu().numpy().reshape(-1, 3)
        collective = {"backend": dist.get_backend(group) + (" (RCCL over xGMI)" if dist.get_backend(group) == "nccl" else ""),
                      "world_size": dist.get_world_size(group), "op": "all_gather_into_tensor of packed alphabet indices, one per layer",
                      "gathered_bytes_per_rank": int(st.get("gather_bytes", 0)),
                      "allgather_ms": float(every[:, 1].max()), "allgather_ms_min_over_steps_per_rank": [float(v) for v in every[:, 2]],
                      "allgather_ms_per_rank": [float(v) for v in every[:, 1]],
                      "kernel_ms_per_rank": [float(v) for v in every[:, 0]],
                      "devices": sorted({torch.cuda.get_device_name(dev)})}

    # the events around the library call: from the timed steps where they were recorded there, else from a few more steps now
    if not all(ev_rec):
        st["call_events"] = True
        for i in range(min(args.steps, 5)):
            step(i)
        fence()
        st["call_events"] = False
    call_ms = [ev[i][0].elapsed_time(ev[i][1]) for i in range(args.steps) if ev_rec[i]]
    # N > 1, strong scaling (the default: the north-star layer is fixed): the same run also steps the weak-scaling layer --
    # args.c neurons PER GPU -- so that one launch of the driver's command shows both regimes.  `value` stays the strong one.
    weak = None
    if world > 1 and args.scaling == "strong":
        Cw = args.c * world
        Ww = np.concatenate([weight_block(N, b, args.c) for b in range(world)], axis=1)
        st["C_total"], st["Wd"] = Cw, torch.from_numpy(Ww).to(dev)
        for _ in range(args.clock_warmup + args.warmup):
            step()
        fence()
        t0w = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        ew = time.perf_counter() - t0w
        tw = torch.tensor([ew], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        ew = float(tw.item())
        weak = {"value": N * Cw * args.steps / ew, "unit": "weights/s", "ms_per_step": ew / args.steps * 1e3, "scaling": "weak",
                "workload": f"Dense({N}->{Cw}): {args.c} neurons per GPU, same samples and alphabet (whole-job aggregate over {world} GPUs)"}
        st["C_total"], st["Wd"] = C_total, Wd

    weights_per_step = N * C_total
    value = weights_per_step * args.steps / elapsed
    lo, hi = layer.shard_bounds(C_total, world, rank)
    k_avg_s = float(np.mean(kernel_ms)) / 1e3
    C_local = hi - lo
    alg_flops = 6.0 * m * N * C_local                # per launch on this rank: 2m for <Xq, u + wX>, 4m for the update
    achieved_tf = alg_flops / k_avg_s / 1e12
    alg_bytes = (8 * m + 8) * N * C_local            # 8d's HBM-equivalent: one f32 row of X and of Xq per weight, w in, q out
    hbm_equiv = alg_bytes / k_avg_s / 1e9
    compulsory = (2 * N * m + 2 * N * C_local) * 4   # X, Xq read once; W in, Q out
    traffic = _recorded_traffic(N, m, C_local, kernel_name)

    if rank == 0:
        out = {
            "metric": "weights_quantized_per_sec", "value": value, "unit": "weights/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_warmup_steps": args.clock_warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_medians_prefetched": ms_prefetched, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "step": ("device-resident alphabet: median(|W|) -> rad * alphabet on the device"
                     + (" on a second HIP stream (it depends on the analog kernel alone: no fork wait, one join) beside the row norms and the record pre-pass"
                        if args.overlap and (world == 1 or N * C_total < layer._SHARDED_MEDIAN_MIN) else ", row norms, record pre-pass")
                     + ", then the block-pipelined kernel reading the Keras kernel and writing Q / indices in the Keras layout; no host wait inside a step" if device_path else
                     "host alphabet (one host wait per step), neuron-major copy, row norms, record pre-pass, kernel, assembly pass"),
            "deferred_status_nonzero_steps": bad_status,
            "config": {
                "workload": (f"Dense({N}->{C_total}) whole-layer GPFQ, m={m} calibration samples, M={M} alphabet, "
                             f"alphabet_scalar={args.alphabet_scalar:g}"
                             + (" (BASELINE.json configs[1], the north-star layer)" if (N, C_total, m, M) == (4096, 4096, 1024, 3) else "")
                             + (f"; neurons split over {world} GPUs ({args.scaling} scaling"
                                + ("; a neuron is a chain of N dependent steps of ~1 us, so the fixed layer is latency-bound per GPU: expect a flat curve"
                                   if args.scaling == "strong" else "") + ")" if world > 1 else "")),
                "N": N, "C": C_total, "m": m, "M": M, "neurons_per_gpu": C_local,
                "sharding": "neurons (columns of W) contiguous over ranks; one all-gather of packed indices per layer" if world > 1 else "none",
                "arithmetic": "f32 products / f64 residual and dot products (reference's mixed flow)",
            },
            "roofline": {
                "bound": "valu_fp64",
                "kernel": kernel_name,
                "achieved": achieved_tf, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved_tf / FP64_VECTOR_PEAK_TFLOPS,
                "traffic": traffic["hbm_bytes_per_launch"] if traffic else None,
                "kernel_ms_avg": k_avg_s * 1e3, "kernel_ms_min": float(np.min(kernel_ms)),
                # events around the library call that launches the kernel: with --overlap (default, one GPU) the alphabet-dependent half only --
                # one in-place pass over the records for the symmetric alphabet + this kernel; --no-overlap: row norms + record pre-pass + this kernel
                "call_ms_avg": float(np.mean(call_ms)),
                # the same fraction on the bracket rounds 1-3 quoted (events around the whole call): comparable across rounds
                "frac_call": alg_flops / (float(np.mean(call_ms)) / 1e3) / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "algorithmic_flops_per_launch": alg_flops,
                "frac_of_measured_issue_rate": achieved_tf / FP64_MEASURED_ISSUE_TFLOPS,
                "measured_issue_rate_tflops": FP64_MEASURED_ISSUE_TFLOPS,
                "hbm_equiv": {"achieved_gbs": hbm_equiv, "peak_gbs": HBM_PEAK_GBS, "frac": hbm_equiv / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_launch": alg_bytes,
                              "note": "SURVEY 8d's (8m+8) B per weight; > 1 means the rows are served on chip, not faster than HBM"},
                "compulsory_bytes_per_launch": compulsory,
                "traffic_gbs": (traffic["hbm_bytes_per_launch"] / k_avg_s / 1e9) if traffic else None,
                "traffic_source": (traffic["source"] if traffic else
                                   "no committed rocprofv3 --pmc pass for this kernel and shape (profiles/traffic.json)"),
                "note": "skinny dot products with the residual on chip: the binding resource is FP64-rate vector issue "
                        "(per weight and sample one f64 fma, one f32->f64 convert, one f64 add, three f32 ops on two samples each -- two for the symmetric ternary alphabet), not HBM; "
                        "flops per launch = 6 m N C_local (SURVEY 8d); duration = the HIP events this kernel's launch carries on its stream -- its dispatch's own "
                        "start and end (gpfq_set_main_kernel_events -> hipExtLaunchKernelGGL; call_ms_avg: events around the library call that launches it, "
                        "recorded in five extra steps after the timed region)",
            },
        }
        if collective is not None:
            out["collective"] = collective
        if weak is not None:
            out["weak_scaling_companion"] = weak
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"], out["parity_sample"] = _cpu_baseline(W, X, Xq, unit_alphabet, args, idx, last)
            if args.numpy_sample != 0:
                try:
                    out["cpu_baseline_numpy"] = _cpu_baseline_numpy(W, X, Xq, unit_alphabet, args, idx)
                except Exception as exc:                          # a secondary record must not cost the headline line
                    out["cpu_baseline_numpy"] = {"error": repr(exc)[:300]}
        if world == 1 and args.long_rows > m:
            out["long_rows"] = _long_rows(N, C_total, args.long_rows, unit_alphabet, args, dev)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _long_rows(N, C, m_long, unit_alphabet, args, dev):
    """Secondary record, outside the timed region of the headline: the same Dense(N -> C) layer on `m_long` calibration samples -- rows
    that do not fit one workgroup's registers and run as the block kernel's CLUSTER FORM (1024-sample slices over several workgroups,
    partial dot products exchanged once per slot: DESIGN 4.1).  Kernel time from the library's own events, three launches; never fails
    the bench line (an exception is reported in place of the numbers)."""
    try:
        from quantized_neural_networks_amd import hip, layer
        g = np.random.default_rng(7).standard_normal((N, m_long)).astype(np.float32)
        X = torch.from_numpy(np.maximum(g, 0)).to(dev)
        Xq = torch.from_numpy(np.maximum(g + 0.1 * np.random.default_rng(8).standard_normal((N, m_long)).astype(np.float32), 0)).to(dev)
        del g
        Wd = torch.from_numpy(weight_block(N, 0, C)).to(dev)
        alphabet, _ = layer.layer_alphabet(Wd, unit_alphabet, args.alphabet_scalar, None)
        Wt = hip.neuron_major(Wd, 0, C)
        nrm = hip.row_norms(Xq)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(3)]
        hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm, want_values=False)                     # warm-up
        for a, b in evs:
            hip.set_main_kernel_events(a, b)
            r = hip.quantize_neurons(X, Xq, Wt, alphabet, nrm32=nrm, want_values=False)
            hip.set_main_kernel_events(None, None)
        torch.cuda.synchronize()
        name = hip.last_dense_kernel()
        ms = [a.elapsed_time(b) for a, b in evs] if name.startswith("gpfq_blk_kernel") else []
        if not ms:
            return {"m": m_long, "kernel": name, "note": "not the block kernel: no inner events"}
        flops = 6.0 * m_long * N * C
        k = float(np.mean(ms)) / 1e3
        return {"workload": f"Dense({N}->{C}) on m={m_long} calibration samples, same alphabet", "kernel": name[:120], "kernel_ms_avg": k * 1e3,
                "kernel_ms_min": float(np.min(ms)), "achieved_tflops": flops / k / 1e12, "frac_of_fp64_vector_peak": flops / k / 1e12 / FP64_VECTOR_PEAK_TFLOPS,
                "weights_per_s_kernel": N * C / k, "cluster_timeouts": hip.cluster_timeouts(r), "exact_fallbacks": hip.exact_fallbacks(r)}
    except Exception as exc:                                          # a secondary record must not cost the headline line
        return {"m": m_long, "error": repr(exc)[:300]}


def _recorded_traffic(N, m, C_local, kernel_name):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json: FETCH_SIZE x 2 +
    WRITE_SIZE, separate --pmc passes of this same command), if one was recorded for this shape AND this kernel
    family; else None.  The counters cannot be read inside the timed run, so this is a record, and says so."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
        for r in rec.get("records", []):
            if (r["N"], r["m"], r["C"]) == (N, m, C_local) and r.get("kernel", "").split("<")[0].split(" ")[0] in kernel_name:
                return {"hbm_bytes_per_launch": r["hbm_bytes_per_launch"],
                        "source": "profiles/traffic.json (" + r.get("profile", "committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes") + ")"}
    except Exception:
        pass
    return None


def _cpu_baseline(W, X, Xq, unit_alphabet, args, idx_gpu, last):
    """The C oracle port (oracle/gpfq_oracle.c, OpenMP over neurons) on this box's host cores over a
    bounded sample of the same layer; also the parity check of the GPU result on that sample."""
    import oracle
    oracle.build()
    alphabet, _ = oracle.layer_alphabet(W, unit_alphabet, args.alphabet_scalar)
    n = min(args.cpu_sample, W.shape[1])
    threads = oracle.num_threads()
    t0 = time.perf_counter()
    Qo, io, ro = oracle.layer(W, X, Xq, alphabet, 0, n, threads=threads)
    dt = time.perf_counter() - t0
    ig = idx_gpu[:, :n].t().cpu().numpy()
    rg = last["resid"][:n].cpu().numpy()
    bad = int((ig != io).any(axis=1).sum())
    rel = float(np.max(np.abs(rg - ro) / np.maximum(ro, 1e-300)))
    cpu = {"value": n * W.shape[0] / dt, "unit": "weights/s", "cores": threads, "kind": "port",
           "sample": f"{n} of {W.shape[1]} neurons of the same layer (independent, equal cost), "
                     f"{dt:.2f} s wall on {threads} threads, in-memory arrays (no HDF5)"}
    parity = {"neurons_checked": n, "neurons_with_index_mismatch": bad, "max_resid_rel_err": rel}
    return cpu, parity


def _cpu_baseline_numpy(W, X, Xq, unit_alphabet, args, idx_gpu):
    """The reference-shaped baseline (north_star: "the reference's NumPy/TensorFlow CPU path timed on the same box's host cores";
    BASELINE.md 3): the NumPy restatement of _quantize_neuron_parallel (oracle.neuron_numpy, explicit casts) over a process pool
    of os.cpu_count() workers, one neuron per task as the reference's executor.submit fan-out (:549-556), in-memory arrays
    instead of HDF5, on >= 2 x cores neurons of the same layer.  Runs in a CHILD process that never touches the GPU (forking a
    pool from this GPU-initialised process would not be safe); its indices are compared with the GPU's."""
    import shutil
    import subprocess
    import tempfile
    import oracle
    cores = os.cpu_count() or 1
    n = min(W.shape[1], 2 * cores if args.numpy_sample < 0 else args.numpy_sample)
    alphabet, _ = oracle.layer_alphabet(W, unit_alphabet, args.alphabet_scalar)
    d = tempfile.mkdtemp(prefix="gpfq_numpy_pool_")
    try:
        np.save(os.path.join(d, "W.npy"), np.ascontiguousarray(W[:, :n]))
        np.save(os.path.join(d, "X.npy"), X)
        np.save(os.path.join(d, "Xq.npy"), Xq)
        np.save(os.path.join(d, "alphabet.npy"), alphabet)
        # (a CPU-only child: no profiler preload or tool library rides along when bench.py itself runs under rocprofv3)
        env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER"))}
        env.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1", PYTHONPATH=ROOT)
        subprocess.run([sys.executable, "-m", "oracle.numpy_pool", d, str(cores)], check=True, cwd=ROOT, env=env, timeout=600)
        with open(os.path.join(d, "result.json")) as f:
            res = json.load(f)
        io = np.load(os.path.join(d, "idx.npy"))
    finally:
        shutil.rmtree(d, ignore_errors=True)
    bad = int((idx_gpu[:, :n].t().cpu().numpy() != io).any(axis=1).sum())
    return {"value": n * W.shape[0] / res["seconds"], "unit": "weights/s", "cores": cores, "kind": "port (NumPy, process pool)",
            "sample": f"{n} of {W.shape[1]} neurons of the same layer, one neuron per task over {cores} worker processes "
                      f"(1 BLAS thread each), {res['seconds']:.2f} s wall, in-memory arrays (the reference reads HDF5 per step: slower)",
            "neurons_with_index_mismatch_vs_gpu": bad}


if __name__ == "__main__":
    main()
