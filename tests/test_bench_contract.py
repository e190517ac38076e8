"""The driver's contract with bench.py: one JSON line on stdout with the agreed keys (GPU box only; a reduced
layer so the check takes seconds -- the numbers of record come from the default invocation)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--n", "256", "--c", "512", "--m", "512", "--cpu-sample", "32", "--numpy-sample", "16"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["higher_is_better"] is True
    assert out["clock_warmup_steps"] == 10 and out["deferred_status_nonzero_steps"] == 0       # (untimed steps before the warm-up steps: the chip's clock ramp)
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["vs_baseline"] is None and "workload" in out["config"]
    roof = out["roofline"]
    for key in ["bound", "achieved", "peak", "unit", "frac", "traffic"]:
        assert key in roof, key
    # the binding resource of the dominant kernel is named, and the fraction is a fraction
    assert roof["bound"] in ("valu_fp64", "hbm", "mfma") and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0 < roof["frac"] <= 1.0 and roof["unit"] in ("TFLOP/s", "GB/s") and roof["kernel"]
    assert roof["kernel_ms_avg"] > 0 and roof["algorithmic_flops_per_launch"] == 6.0 * 512 * 256 * 512
    assert roof["compulsory_bytes_per_launch"] == (2 * 256 * 512 + 2 * 256 * 512) * 4 and "hbm_equiv" in roof
    # the kernel's own events (its dispatch's start and end) lie inside the events around the whole call (five extra steps after the timed region)
    assert 0 < roof["kernel_ms_min"] <= roof["kernel_ms_avg"] <= roof["call_ms_avg"]
    cpu = out["cpu_baseline"]
    for key in ["value", "unit", "cores", "kind", "sample"]:
        assert key in cpu, key
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0
    assert out["parity_sample"]["neurons_with_index_mismatch"] == 0
    # round 5: the reference-shaped baseline (NumPy restatement over a process pool, a child process) beside the C port, checked
    # against the GPU's indices; and the fraction on the rounds-1-3 bracket (events around the whole call) beside `frac`
    npb = out["cpu_baseline_numpy"]
    assert npb["value"] > 0 and npb["cores"] >= 1 and npb["neurons_with_index_mismatch_vs_gpu"] == 0 and "process" in npb["kind"]
    assert 0 < roof["frac_call"] <= roof["frac"]


@pytest.mark.gpu
def test_main_kernel_events_bracket_the_recurrence_kernel_alone():
    """gpfq_set_main_kernel_events: the block-pipelined kernel is launched with the caller's two events (its own start and end) -- inside the
    events around the gpfq_quantize_neurons call, which also holds the record pre-pass; cleared, nothing is recorded; a kernel
    family that does not take part leaves them alone."""
    import numpy as np
    import torch
    from quantized_neural_networks_amd import hip
    r = np.random.default_rng(5)
    N, m, C = 512, 1024, 1024
    X = torch.from_numpy(r.random((N, m)).astype(np.float32)).cuda()
    Wt = torch.from_numpy(r.standard_normal((C, N)).astype(np.float32)).cuda()
    alphabet = np.array([-0.7, 0.0, 0.7])
    mk = lambda: torch.cuda.Event(enable_timing=True)
    hip.quantize_neurons(X, X, Wt, alphabet, want_values=False)                      # warm
    assert hip.last_dense_kernel().startswith("gpfq_blk_kernel")
    a, b, k0, k1 = mk(), mk(), mk(), mk()
    hip.set_main_kernel_events(k0, k1)
    try:
        a.record()
        ref = hip.quantize_neurons(X, X, Wt, alphabet, want_values=False)
        b.record()
    finally:
        hip.set_main_kernel_events(None, None)
    torch.cuda.synchronize()
    inner, outer = k0.elapsed_time(k1), a.elapsed_time(b)
    assert 0 < inner < outer
    assert a.elapsed_time(k0) > 0 and k1.elapsed_time(b) >= 0                        # a < k0 < k1 <= b on the stream
    # cleared: a further call records nothing into the old events, and results never depend on the hook
    again = hip.quantize_neurons(X, X, Wt, alphabet, want_values=False)
    torch.cuda.synchronize()
    assert k0.elapsed_time(k1) == inner and torch.equal(ref["idx"], again["idx"])


@pytest.mark.gpu
def test_bench_two_ranks_over_gloo_split_the_fixed_layer():
    """The N > 1 path of bench.py as the driver launches it (torch.distributed.run, one process per rank), with the
    collectives on gloo so that two ranks can share this box's one GPU: the default workload is the FIXED layer split
    over the ranks (strong scaling, the north-star curve), one JSON line from rank 0."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, GPFQ_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--fan-in", "256", "--neurons", "512", "--samples", "512"]      # (the launcher's parser would take "--n" for its own)
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["C"] == 512 and out["config"]["neurons_per_gpu"] == 256 and "split over 2 GPUs" in out["config"]["workload"]
    assert "cpu_baseline" not in out                     # rank 0 at N = 1 only
    assert 0 < out["roofline"]["frac"] <= 1.0
    weak = out["weak_scaling_companion"]                 # the same launch also steps --neurons per GPU (weak scaling)
    assert weak["scaling"] == "weak" and weak["value"] > 0 and "Dense(256->1024)" in weak["workload"]
    col = out["collective"]                              # what the collective saw: backend, group size, its own time, every rank's kernel
    assert col["backend"].startswith("gloo") and col["world_size"] == 2
    assert len(col["kernel_ms_per_rank"]) == 2 and all(v > 0 for v in col["kernel_ms_per_rank"])
    assert col["allgather_ms"] > 0 and len(col["allgather_ms_per_rank"]) == 2
    assert col["gathered_bytes_per_rank"] == 512 * 256 * 2 // 8      # 512 neurons x 256 weights at 2 bits (ternary)


@pytest.mark.gpu
def test_bench_eight_ranks_over_gloo_strong_and_weak():
    """World size 8 (the north-star's), functionally: eight ranks share this box's one GPU, the collectives go over gloo.  The fixed
    layer (520 neurons: 65 per rank, not a multiple of the 16-neuron workgroups) is split over the ranks, the per-rank arrays
    of the `collective` object have eight entries, and the same launch steps the weak-scaling companion."""
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, GPFQ_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--fan-in", "256", "--neurons", "520", "--samples", "512"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=1500, env=env)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["value"] > 0
    assert out["config"]["C"] == 520 and out["config"]["neurons_per_gpu"] == 65 and "split over 8 GPUs" in out["config"]["workload"]
    assert "cpu_baseline" not in out and 0 < out["roofline"]["frac"] <= 1.0
    weak = out["weak_scaling_companion"]
    assert weak["scaling"] == "weak" and weak["value"] > 0 and "Dense(256->4160)" in weak["workload"]
    col = out["collective"]
    assert col["backend"].startswith("gloo") and col["world_size"] == 8
    assert len(col["kernel_ms_per_rank"]) == 8 and all(v > 0 for v in col["kernel_ms_per_rank"])
    assert col["allgather_ms"] > 0 and len(col["allgather_ms_per_rank"]) == 8
    assert col["gathered_bytes_per_rank"] == 8 * 65 * 256 * 2 // 8     # eight 65-neuron shards x 256 weights at 2 bits
