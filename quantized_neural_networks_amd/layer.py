"""Device-side layer drivers: what the reference's ``_quantize_layer_parallel``
(scripts/quantized_network.py:523-574) and ``_quantize_conv2D_layer_parallel_jit`` (:815-867) do
between "activations and weights are available" and "Q is handed to set_weights", on GPU tensors.

Multi-GPU: the independent units -- neurons of a Dense layer, (input-channel, filter)
pairs of a conv layer -- are partitioned contiguously over the ranks of a ``torch.distributed``
process group (one process per GPU, backend "nccl" = RCCL over xGMI).  Every rank holds the full
activation matrices and the full analog kernel, quantizes its shard, and ONE all-gather per layer
reassembles the quantized kernel; there is no other communication.
"""
import numpy as np
import torch

from . import hip


# ------------------------------------------------------------------------------------------
# alphabet radius
# ------------------------------------------------------------------------------------------
# Below this many weights one GPU selects the median faster than three dependent all-reduces take.  Round 6: the one-GPU select is two reads
# of the data in two launches (70 us at 16.8 M weights), the sharded protocol three reads of a 1 / world share + three all-reduces of 16 KiB
# (tens of microseconds EACH over RCCL, and every one waits for the slowest rank): the break-even moved from 4 M to about 32 M weights --
# the north-star layer (16.8 M) selects locally on every rank, VGG16's fc1 (102.8 M) shards its counting.
_SHARDED_MEDIAN_MIN = 1 << 25


def median_abs(W, group=None, meanwhile=None):
    """np.median(np.abs(W.flatten())) for float32 W (:544, :831) as a float32 value: the middle
    element, or for an even count the float32 mean of the two middle elements (NumPy semantics;
    torch.median would return the lower one).  With a process group (every rank holds W) each rank counts
    one slice of the flattened kernel and the histograms are summed over the ranks -- the same value.
    `meanwhile`: see hip.median_abs (called exactly once, also for an empty kernel)."""
    if W.numel() == 0:
        if meanwhile is not None:
            meanwhile()
        return np.float32(np.nan)
    flat = W.detach().reshape(-1)
    n = flat.numel()
    world, rank = _group_info(group)
    if world == 1 or n < _SHARDED_MEDIAN_MIN or not flat.is_cuda:
        return hip.median_abs(flat, meanwhile)
    import torch.distributed as dist
    per = -(-n // (4 * world)) * 4                       # slices start on multiples of 4 elements
    lo, hi = min(rank * per, n), min((rank + 1) * per, n)
    return hip.median_abs_sharded(flat[lo:hi], n, lambda t: dist.all_reduce(t, group=group), meanwhile)


def layer_alphabet_device(W, alphabet, alphabet_scalar, group=None):
    """The layer alphabet of :544-545 formed and kept ON THE DEVICE (hip.DeviceAlphabet): the median of |W| stays a device scalar and one
    single-thread kernel forms rad = float64(alphabet_scalar) * float64(median) and rad * alphabet -- nothing waits for the host.
    quantize_dense() takes it in place of the host alphabet wherever the block-pipelined kernel runs (hip.dense_layer_supported)."""
    flat = W.detach().reshape(-1)
    n = flat.numel()
    if n == 0:
        raise hip.GpfqError("layer_alphabet_device: empty kernel")
    world, rank = _group_info(group)
    if world == 1 or n < _SHARDED_MEDIAN_MIN:
        return hip.layer_alphabet_from_kernel(flat, alphabet, alphabet_scalar)      # median + alphabet: one call, two launches
    else:
        import torch.distributed as dist
        per = -(-n // (4 * world)) * 4
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        med = hip.median_abs_sharded(flat[lo:hi], n, lambda t: dist.all_reduce(t, group=group), on_device=True)
    return hip.layer_alphabet_device(med, alphabet, alphabet_scalar)


def layer_alphabet(W, alphabet, alphabet_scalar, group=None, meanwhile=None):
    """(rad * alphabet, rad) with the reference's legacy-NumPy typing (:544-545): the python
    scalar times the float32 median is a float64 product.  `meanwhile()` may queue GPU work that does not need the
    alphabet; it runs while the host waits for the median."""
    rad = np.float64(alphabet_scalar) * np.float64(median_abs(W, group, meanwhile))
    return rad * np.asarray(alphabet, dtype=np.float64), rad


# ------------------------------------------------------------------------------------------
# sharding helpers
# ------------------------------------------------------------------------------------------
def shard_bounds(n_units, world_size, rank):
    """Contiguous partition of n_units over world_size ranks: [lo, hi) of `rank`."""
    per = -(-n_units // world_size)
    lo = min(rank * per, n_units)
    return lo, min(lo + per, n_units)


_warned_unsharded = False


def _warn_unsharded_once():
    """group=None inside an initialised multi-rank job: every rank quantizes the whole layer on its own.  That is the
    documented meaning, but a caller who expected the default group gets N redundant copies of the
    work and no error -- say so once."""
    global _warned_unsharded
    if _warned_unsharded:
        return
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        _warned_unsharded = True
        import warnings
        warnings.warn("quantized_neural_networks_amd.layer: torch.distributed is initialised with "
                      f"{dist.get_world_size()} ranks but no process group was passed: this rank quantizes the whole layer "
                      "alone (pass group=dist.group.WORLD / process_group=... to shard the neurons over the ranks)",
                      RuntimeWarning, stacklevel=3)


def _group_info(group):
    """(world, rank) of an EXPLICIT process group; ``None`` means "this process alone" even inside an initialised
    torch.distributed job (a rank-0-only quantization under torchrun must not wait for collectives the other ranks
    never enter).  Pass ``dist.group.WORLD`` to shard over all ranks."""
    if group is None:
        _warn_unsharded_once()
        return 1, 0
    import torch.distributed as dist
    return dist.get_world_size(group), dist.get_rank(group)


def all_gather_units(local, n_units, group=None):
    """Reassemble a [units_local, ...] shard into [n_units, ...] on every rank with one all-gather.
    Shards are padded to the common size ceil(n_units / world) so a single
    all_gather_into_tensor (ncclAllGather on RCCL) moves them."""
    import torch.distributed as dist
    world, rank = _group_info(group)
    if world == 1:
        return local
    if local.dtype == torch.int16 and local.dim() >= 2:
        # RCCL (like NCCL and gloo) has no 16-bit integer type: the indices of 65..256-member alphabets travel as bytes
        return all_gather_units(local.contiguous().view(torch.uint8), n_units, group).view(torch.int16)
    per = -(-n_units // world)
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    return out[:n_units]


# ------------------------------------------------------------------------------------------
# Dense layer
# ------------------------------------------------------------------------------------------
def _log_failure(log, msg):
    import warnings
    warnings.warn(msg, RuntimeWarning, stacklevel=3)
    if log is not None:
        log(msg)


def quantize_neurons_checked(X, Xq, Wt, alphabet, log=None, **kw):
    """hip.quantize_neurons with the block kernel's deferred failure handled at the reference's granularity (the reference logs the
    failing unit and re-raises at once, scripts/quantized_network.py:563-565): when the call went through the CLUSTER FORM -- several
    workgroups per group of neurons that exchange partial dot products, which rests on their being co-resident -- its status word is read
    (one host wait) BEFORE the result is used; a timed-out exchange is logged and the same neurons are rerun at once through the classic
    kernels (option blk_cluster = 0), and only a failure of that run raises.  Every other kernel family has no deferred failure and no wait."""
    r = hip.quantize_neurons(X, Xq, Wt, alphabet, **kw)
    if "cluster form" not in hip.last_dense_kernel():
        return r
    st = hip.call_status(r)
    if st == 0:
        return r
    _log_failure(log, f"quantize_neurons: the cluster form's exchange timed out on {Wt.shape[0]} neurons x {X.shape[1]} samples "
                      f"(status {st}); rerunning them through the classic kernels")
    with hip.option("blk_cluster", 0):
        r = hip.quantize_neurons(X, Xq, Wt, alphabet, **kw)
        st = hip.call_status(r)
    if st != 0:
        raise hip.GpfqError(f"quantize_neurons failed again without the cluster form (status {st})")
    r["cluster_fallback"] = True
    return r


_side_streams = {}


def _side_stream(device):
    st = _side_streams.get(device.index)
    if st is None:
        st = _side_streams[device.index] = torch.cuda.Stream(device=device)
    return st


def quantize_dense_layer(W, X, Xq, unit_alphabet, alphabet_scalar, group=None, want_resid=True, log=None, check=True, overlap=False,
                         kernel_ready=None):
    """The body of _quantize_layer_parallel (scripts/quantized_network.py:523-574) from "the activations are there" to the tensors
    set_weights takes, with nothing crossing to the host: median of |W| -> rad * alphabet on the device (:544-545), row norms, record
    pre-pass, the recurrence reading the Keras kernel in place.

    overlap=True runs its two independent halves on two HIP streams: the median and the alphabet depend on the kernel W alone and go to a
    side stream; the row norms and the record pre-pass depend on the activations alone and stay on this one (gpfq_dense_layer_prepare);
    ONE join, then the alphabet-dependent rest (gpfq_dense_layer_run).  kernel_ready says when W was complete: None -- unknown, the side
    stream first waits for everything outstanding on this one (always safe; the wait then sits on the longer of the two chains and the
    overlap buys little); a torch.cuda.Event -- the side stream waits for that; True -- W was complete before anything now outstanding
    here was queued (a trained network's analog kernel: the reference never writes it), no wait at all: the median of layer k + 1 then
    also fills the tail of layer k.  Measured at the north-star layer: 3.05 -> 3.00 ms per layer (profiles/r06/overlap_ab.txt).
    (Round 6's first scheme -- the pre-pass on the side stream behind a fork wait -- measured a wash and is gone.)

    Same tensors as quantize_dense(W, X, Xq, rad * unit_alphabet), bit for bit.  Returns its dict + "alphabet" (the hip.DeviceAlphabet)."""
    N, C = W.shape
    world, rank = _group_info(group)
    lo, hi = shard_bounds(C, world, rank)
    m = X.shape[1]
    side_ok = world == 1 or W.numel() < _SHARDED_MEDIAN_MIN      # (the sharded median's collectives stay on the caller's stream)
    if not (overlap and side_ok and W.numel() and m > 0 and hi > lo and hip.dense_layer_supported(N, m, hi - lo, unit_alphabet)):
        dalpha = layer_alphabet_device(W, unit_alphabet, alphabet_scalar, group)
        out = quantize_dense(W, X, Xq, dalpha, group=group, want_resid=want_resid, log=log, check=check)
        out["alphabet"] = dalpha
        return out
    main = torch.cuda.current_stream(W.device)
    side = _side_stream(W.device)
    if kernel_ready is None:
        side.wait_stream(main)
    elif kernel_ready is not True:
        side.wait_event(kernel_ready)
    with torch.cuda.stream(side):
        dalpha = layer_alphabet_device(W, unit_alphabet, alphabet_scalar, None)
    W.record_stream(side)                                         # (the allocator's bookkeeping: both streams use W and the alphabet block)
    dalpha.buf.record_stream(main)
    ws = hip.dense_layer_workspace(N, m, hi - lo, W.device)
    hip.dense_layer_prepare(X, Xq, unit_alphabet, hi - lo, ws)
    main.wait_stream(side)
    out = quantize_dense(W, X, Xq, dalpha, group=group, want_resid=want_resid, log=log, check=check, prepared=ws)
    out["alphabet"] = dalpha
    return out


def quantize_dense(W, X, Xq, alphabet, group=None, want_resid=True, log=None, check=True, prepared=None):
    """Quantize every neuron (column) of a Dense kernel.

    W        f32 [N][C]  Keras kernel layout (row = input feature), on the GPU
    X, Xq    f32 [N][m]  feature-major analog / quantized activations (the transposed wX, qX)
    alphabet f64 [M]     the layer alphabet rad * linspace(-1, 1, M) -- or a hip.DeviceAlphabet (layer_alphabet_device): the radius then
                         never crosses to the host, the kernel reads W in its Keras layout and, on one GPU, writes Q and the indices
                         in it (no neuron-major copy, no assembly pass); shapes the block-pipelined kernel does not take fall back
                         to the host alphabet (one read-back)

    Returns dict(Q f32 [N][C], idx i8 [N][C], resid f64 [C]) on every rank.  want_resid=None: residual norms only
    where the kernel holds the residual anyway (NaN from the Gram path, which would replay it in an extra pass).
    check=False (device alphabets; benchmarks): the deferred status of the launch is NOT read here -- no host wait at all; the result
    carries "workspace" and the caller owes hip.call_status(result) before it trusts Q.
    """
    N, C = W.shape
    world, rank = _group_info(group)
    lo, hi = shard_bounds(C, world, rank)
    dalpha = alphabet if isinstance(alphabet, hip.DeviceAlphabet) else None
    if dalpha is not None and not (X.shape[1] > 0 and hip.dense_layer_supported(N, X.shape[1], max(hi - lo, 1), dalpha.unit)):
        alphabet, dalpha = dalpha.values(), None                   # (no block-pipelined kernel for this shape: the host alphabet's paths)
    Wc = W.contiguous()
    if dalpha is not None:
        r = hip.quantize_dense_layer(X, Xq, Wc, dalpha, lo, hi, keras_out=(world == 1), want_values=(world == 1), want_resid=want_resid,
                                     prepared=prepared)           # (prepared: quantize_dense_layer's side stream has run the pre-pass into this workspace)
        # (the status costs one host wait: skipped where nothing deferred can have happened -- the caller knows the radius is a finite
        #  positive number, DeviceAlphabet.radius_ok, and the launch was not the cluster form)
        need = check and not (getattr(dalpha, "radius_ok", False) and "cluster form" not in hip.last_dense_kernel())
        st = hip.call_status(r) if need else 0
        if st == hip.GPFQ_ERR_CLUSTER_TIMEOUT:
            _log_failure(log, f"Dense layer {N} x {C}: the cluster form's exchange timed out; rerunning the layer through the classic kernels")
            with hip.option("blk_cluster", 0):
                r = hip.quantize_dense_layer(X, Xq, Wc, dalpha, lo, hi, keras_out=(world == 1), want_values=(world == 1), want_resid=want_resid)
                st = hip.call_status(r)
        if st == hip.GPFQ_ERR_ALPHABET:
            # radius 0 / infinite / NaN (the median of a kernel that is mostly zeros): only the host alphabet's kernels take such a one
            alphabet, dalpha = dalpha.values(), None
        elif st != 0:
            raise hip.GpfqError(f"quantize_dense: status {st} after the fallback")
    if dalpha is not None:
        if world == 1:
            out = dict(Q=r["Q"], idx=r["idx"], workspace=r["workspace"], cluster_err=torch.zeros(1, dtype=torch.int32, device=W.device))
            if want_resid is not False:
                out["resid"] = r["resid"]
            return out
        packed, bits = hip.pack_indices(r["idx"], len(dalpha))
        Q, idx = hip.assemble_kernel_device(all_gather_units(packed, C, group).contiguous(), dalpha, bits=bits, N=N)
        out = dict(Q=Q, idx=idx, workspace=r["workspace"], cluster_err=torch.zeros(1, dtype=torch.int32, device=W.device))
        if want_resid is not False:
            out["resid"] = all_gather_units(r["resid"], C, group)
        return out
    Wt = hip.neuron_major(Wc, lo, hi)                            # neuron-major shard [C_local][N]
    deferred = None
    if hi > lo:
        if world > 1:
            # (a rank whose exchange timed out repairs its shard BEFORE the all-gather: the other ranks never see garbage and nobody
            #  has to agree on anything)
            r = quantize_neurons_checked(X, Xq, Wt, alphabet, log=log, want_values=False, want_resid=want_resid)
        else:
            # one GPU: the assembly pass is queued behind the kernel first and the status read after it -- the host's wait then costs no
            # bubble between the two (cfg4's Dense(128 -> 10): 0.06 ms of a 0.15 ms layer); nothing is RETURNED unchecked
            r = hip.quantize_neurons(X, Xq, Wt, alphabet, want_values=False, want_resid=want_resid)
            deferred = r if "cluster form" in hip.last_dense_kernel() else None
        i_loc, res_loc = r["idx"], r["resid"]
    else:
        i_loc = torch.empty((0, N), dtype=hip.index_dtype(len(alphabet)), device=W.device)
        res_loc = torch.empty((0,), dtype=torch.float64, device=W.device)
    # only the indices travel over xGMI -- packed to 2 or 4 bits per weight when the alphabet allows;
    # values are looked up while transposing to the Keras layout
    if world > 1:
        packed, bits = hip.pack_indices(i_loc, len(alphabet))
        Q, idx = hip.assemble_kernel(all_gather_units(packed, C, group).contiguous(), alphabet, bits=bits, N=N)
    else:
        Q, idx = hip.assemble_kernel(i_loc, alphabet)
        if deferred is not None and hip.call_status(deferred) != 0:
            _log_failure(log, f"Dense layer {N} x {C}: the cluster form's exchange timed out; rerunning the layer through the classic kernels")
            with hip.option("blk_cluster", 0):
                r = hip.quantize_neurons(X, Xq, Wt, alphabet, want_values=False, want_resid=want_resid)
                if hip.call_status(r) != 0:
                    raise hip.GpfqError("quantize_dense failed again without the cluster form")
            res_loc = r["resid"]
            Q, idx = hip.assemble_kernel(r["idx"], alphabet)
    # (cluster_err: kept for callers of round 5's interface -- a timed-out exchange no longer leaves this function, see quantize_neurons_checked)
    out = dict(Q=Q, idx=idx, cluster_err=torch.zeros(1, dtype=torch.int32, device=W.device))
    if want_resid is not False:
        out["resid"] = all_gather_units(res_loc, C, group)
    return out


# ------------------------------------------------------------------------------------------
# Conv2D layer
# ------------------------------------------------------------------------------------------
def _quantize_conv1x1(W, act_q, alphabet, strides):
    """1x1 kernels: every (channel, filter) pair is a ONE-step walk.  u = 0, hence <Xq_0, u> = 0 and rule
    (ii) (:86-87) returns nearest(alphabet, w) -- plain MSQ -- unless the channel's (sub-sampled) quantized
    activations are identically zero, where rule (i) (:83-84) returns the literal 0.  So the layer needs the
    per-channel norms and one MSQ pass instead of Cin patch matrices; cheaper than any all-gather, so it is
    not sharded.  (The reference reaches the same values through its general path: its (1,1) shortcut is
    dead code, :835-842.)  The residual norms are not formed (NaN)."""
    _, _, Cin, F = W.shape
    sh, sw = strides
    # all the layer needs of its activations is whether each channel's float32-rounded row norm is below 1e-16: partial sums
    # of squares only grow, so a prefix of the positions settles every live channel and only what is left undecided is
    # summed in full (hip.channel_dead; SAME == VALID for k = 1) -- no pass over the whole NHWC tensor
    Q, idx = hip.quantize_conv1x1(act_q.contiguous(), W.reshape(Cin, F), alphabet, (sh, sw))     # one library call, no host round trip
    resid = torch.full((Cin, F), float("nan"), dtype=torch.float64, device=W.device)
    return dict(Q=Q.reshape(1, 1, Cin, F), idx=idx.reshape(1, 1, Cin, F), resid=resid)


def quantize_conv2d(W, act_w, act_q, alphabet, strides, padding, rate, group=None, want_resid=True):
    """Quantize a Conv2D / DepthwiseConv2D kernel channel by channel.

    W          f32 [kh][kw][Cin][F]   Keras kernel layout
    act_w/q    f32 NHWC [n][H][W][Cin] analog / quantized layer inputs
    Each (input channel c, filter f) pair is an independent neuron of kh*kw weights whose data are
    the rows of channel c's patch matrix (:652-727); channels are partitioned over the ranks.  When Cin < world the
    Gram records are formed over image shards and all-reduced (or, where that does not apply, the filters of each
    channel are partitioned instead).  Sharded runs exchange ONE all-gather per layer: the alphabet indices of the
    rank's (channel, filter) pairs, packed to 2 / 4 bits per weight as the dense path's are; values are looked up
    locally (+ one all-gather of the residual norms when they are requested).

    Returns dict(Q f32 [kh][kw][Cin][F], idx i8 same shape, resid f64 [Cin][F]).
    """
    kh, kw, Cin, F = W.shape
    K = kh * kw
    dev = W.device
    if K == 1 and not want_resid:
        return _quantize_conv1x1(W, act_q, alphabet, strides)
    world, rank = _group_info(group)
    by_channel = Cin >= world
    Qc = torch.zeros((Cin, F, K), dtype=torch.float32, device=dev)
    Ic = torch.zeros((Cin, F, K), dtype=hip.index_dtype(len(alphabet)), device=dev)
    Rc = torch.full((Cin, F), 0.0 if want_resid else float("nan"), dtype=torch.float64, device=dev)
    c_lo, c_hi = shard_bounds(Cin, world, rank) if by_channel else (0, Cin)
    f_lo, f_hi = (0, F) if by_channel else shard_bounds(F, world, rank)
    Pw = Pq = None
    reruns = 0                                 # filters of this rank rerun through the exact path (diagnostics)
    # channel-major copies [Cin][n][H][W] of this rank's channels: the per-channel gather then reads
    # contiguous planes instead of one float out of every Cin (one transposing pass per layer)
    same = act_q is act_w
    act_w = act_w.contiguous()
    act_q = act_w if same else act_q.contiguous()
    # 3 x 3 / stride 1 / SAME shards of 32+ channels (narrower ones where image groups fill the lanes) read the NHWC tensors directly; everything else goes through the planes
    nhwc = (not want_resid and by_channel and (kh, kw) == (3, 3) and tuple(strides) == (1, 1) and tuple(rate or (1, 1)) == (1, 1)
            and str(padding).upper() == "SAME" and len(alphabet) <= hip.GPFQ_MAX_ALPHABET
            and hip.conv3x3_nhwc_supported(act_w.shape[0], act_w.shape[1], act_w.shape[2], c_hi - c_lo))
    # ... and so do the layers of the 7 x 7 / stride 2 / VALID shift-sum form (ResNet50's conv1): its requests gather the bands out of the
    # interleaved rows
    nhwc_any = (not nhwc and not want_resid and by_channel and len(alphabet) <= hip.GPFQ_MAX_ALPHABET
                and hip.conv_channels_nhwc_supported(act_w.shape[0], act_w.shape[1], act_w.shape[2], c_hi - c_lo, (kh, kw), strides,
                                                     rate, padding))
    cm = {}

    def planes():
        if not cm:
            cm["w"] = hip.channel_planes(act_w, c_lo, c_hi)
            cm["q"] = cm["w"] if same else hip.channel_planes(act_q, c_lo, c_hi)
        return cm["w"], cm["q"]

    def patches(c):
        nonlocal Pw, Pq
        if nhwc or nhwc_any:                                           # (rare reruns: that channel's slice alone)
            Pw = hip.extract_patches(act_w[..., c:c + 1].contiguous(), 0, (kh, kw), strides, rate, padding, out=Pw)
            Pq = Pw if same else hip.extract_patches(act_q[..., c:c + 1].contiguous(), 0, (kh, kw), strides, rate, padding, out=Pq)
            return
        cm_w, cm_q = planes()
        Pw = hip.extract_patches(cm_w[c - c_lo].unsqueeze(-1), 0, (kh, kw), strides, rate, padding, out=Pw)
        Pq = Pw if same else hip.extract_patches(cm_q[c - c_lo].unsqueeze(-1), 0, (kh, kw), strides, rate, padding, out=Pq)

    def rerun_flagged(Unc, with_resid):
        """The (channel, filter) pairs the Gram path left flagged, through the verbatim flow on the channel's patch matrices: ONE
        exact call per CHANNEL for all its flagged filters (round 5).  Normally about one pair in 10^4; but where the quantized
        network's activations have drifted orders of magnitude away from the analog ones (the last blocks of a 50-layer network
        whose 1 x 1 layers are plain MSQ: ResNet50's conv5_block3_2_conv flagged 179 000 of its 262 144 pairs) the prediction's
        bound is wider than the alphabet's spacing for most walks -- one call per PAIR was 33 s of launches for that layer."""
        flagged = torch.nonzero(Unc)                                   # one sync per layer
        if flagged.numel() == 0:
            return 0
        by_c = {}
        for c, f in flagged.tolist():
            by_c.setdefault(c, []).append(f)
        for c, fs in by_c.items():
            patches(c)
            fsel = torch.tensor(fs, dtype=torch.long, device=dev)
            path = hip.GPFQ_PATH_ONCHIP if Pw.shape[1] <= hip.GPFQ_ONCHIP_MAX_M else hip.GPFQ_PATH_STREAM
            r = quantize_neurons_checked(Pw, Pq, Wt_all[c].index_select(0, fsel).contiguous(), alphabet, path=path)     # (rows of up to 28672 samples: possibly the cluster form)
            Qc[c, fsel], Ic[c, fsel] = r["Q"], r["idx"]
            if with_resid:
                Rc[c, fsel] = r["resid"]
        return int(flagged.shape[0])

    # neuron-major filters [Cin][F][K]: row-major flattening of each kh x kw filter (:215), t = ky*kw + kx
    Wt_all = W.permute(2, 3, 0, 1).reshape(Cin, F, K).contiguous()
    rh, rw = rate if rate else (1, 1)
    same_pad = str(padding).upper() == "SAME"
    cols = (act_w.shape[0] * hip.patch_out_dim(act_w.shape[1], kh, strides[0], rh, same_pad)
            * hip.patch_out_dim(act_w.shape[2], kw, strides[1], rw, same_pad))
    replicated = False                         # every rank already holds the whole result (column-sharded records)
    if not by_channel and not want_resid and K <= hip.GPFQ_GRAM_AUTO_MAX_N and act_w.shape[0] >= world:
        # Fewer input channels than ranks (an image input has 3): the Gram records are sums over the patch columns, so
        # every rank forms them over its share of the IMAGES, one all-reduce of Cin * (2 K^2 + K) doubles sums them,
        # and every rank finishes (decide + repair, milliseconds) from the same records -- no gather afterwards.
        # Certified decisions do not depend on the summation order of the records.  Both phases must have a kernel
        # for their shape (the first sees this rank's images, the second all of them): all ranks agree on that first.
        import torch.distributed as dist
        n_lo, n_hi = shard_bounds(act_w.shape[0], world, rank)
        rec = neg = None
        try:
            pw = hip.channel_planes(act_w[n_lo:n_hi].contiguous(), 0, Cin)
            pq = pw if same else hip.channel_planes(act_q[n_lo:n_hi].contiguous(), 0, Cin)
            rec, neg = hip.conv_channel_records(pw, pq, (kh, kw), strides, rate, padding)
            supported = 1 if hip.conv_records_supported(act_w.shape[0], act_w.shape[1], act_w.shape[2], Cin, (kh, kw), strides,
                                                        rate, padding) else 0
        except hip.GpfqError:
            supported = 0
        if rec is None:
            rec = torch.zeros((Cin, K * K * 2 + K), dtype=torch.float64, device=dev)
            neg = torch.zeros((Cin,), dtype=torch.int32, device=dev)
        ok = torch.tensor([supported], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)          # all ranks take the same branch
        if int(ok.item()):
            dist.all_reduce(rec, op=dist.ReduceOp.SUM, group=group)
            dist.all_reduce(neg, op=dist.ReduceOp.MAX, group=group)
            Unc = torch.zeros((Cin, F), dtype=torch.int32, device=dev)
            hip.conv_channels_from_records(rec, neg, *planes(), Wt_all, alphabet, (kh, kw), strides, rate, padding, Ic, Qc, Unc)
            reruns += rerun_flagged(Unc, False)                            # the same (rare) pairs on every rank
            replicated = True
    # the whole-shard Gram call for every conv layer (one launch chain instead of a Python loop over the
    # channels); with residual norms requested it builds patch matrices, which only pays for long ones
    whole_shard = K <= hip.GPFQ_GRAM_AUTO_MAX_N and cols > (hip.GPFQ_GRAM_MIN_M if want_resid else 0)
    if replicated or f_hi <= f_lo:
        pass
    elif whole_shard:
        # Gram path, the whole channel loop (:844-860) in one library call: no per-channel allocation,
        # Python or sync; the filters whose decision chain could not be certified are collected once
        Unc = torch.zeros((Cin, F), dtype=torch.int32, device=dev)
        if (f_lo, f_hi) == (0, F) and nhwc:
            hip.quantize_conv3x3_nhwc(act_w, act_q, c_lo, c_hi, Wt_all[c_lo:c_hi], alphabet, Ic[c_lo:c_hi], Qc[c_lo:c_hi], Unc[c_lo:c_hi])
        elif (f_lo, f_hi) == (0, F) and nhwc_any:
            hip.quantize_conv_channels_nhwc(act_w, act_q, c_lo, c_hi, Wt_all[c_lo:c_hi], alphabet, (kh, kw), strides, rate, padding,
                                            Ic[c_lo:c_hi], Qc[c_lo:c_hi], Unc[c_lo:c_hi])
        elif (f_lo, f_hi) == (0, F):
            hip.quantize_conv_channels(*planes(), Wt_all[c_lo:c_hi], alphabet, (kh, kw), strides, rate, padding,
                                       Ic[c_lo:c_hi], Qc[c_lo:c_hi], Rc[c_lo:c_hi] if want_resid else None, Unc[c_lo:c_hi])
        else:                                  # filters split over ranks (Cin < world): every rank walks all channels
            Wt_f = Wt_all[:, f_lo:f_hi].contiguous()
            i_f = torch.empty((Cin, f_hi - f_lo, K), dtype=hip.index_dtype(len(alphabet)), device=dev)
            q_f = torch.empty((Cin, f_hi - f_lo, K), dtype=torch.float32, device=dev)
            r_f = torch.full((Cin, f_hi - f_lo), float("nan"), dtype=torch.float64, device=dev)
            u_f = torch.empty((Cin, f_hi - f_lo), dtype=torch.int32, device=dev)
            hip.quantize_conv_channels(*planes(), Wt_f, alphabet, (kh, kw), strides, rate, padding, i_f, q_f,
                                       r_f if want_resid else None, u_f)
            Ic[:, f_lo:f_hi], Qc[:, f_lo:f_hi], Rc[:, f_lo:f_hi], Unc[:, f_lo:f_hi] = i_f, q_f, r_f, u_f
        reruns = rerun_flagged(Unc, True)                             # one sync per layer; ~1 filter in 10^4
    else:
        for c in range(c_lo, c_hi):
            patches(c)
            r = quantize_neurons_checked(Pw, Pq, Wt_all[c, f_lo:f_hi], alphabet)
            Qc[c, f_lo:f_hi] = r["Q"]
            Ic[c, f_lo:f_hi] = r["idx"]
            Rc[c, f_lo:f_hi] = r["resid"]
    if world > 1 and not replicated:
        # ONE all-gather of the packed alphabet indices ((channel, filter) pairs are rows of K weights); the float32
        # values never travel: every rank looks them up while laying the kernel out ([K][pairs] is Keras' [kh][kw][Cin][F])
        if by_channel:
            rows = Ic[c_lo:c_hi].reshape(-1, K).contiguous()                       # (c, f) rows of this rank's channels
            packed, bits = hip.pack_indices(rows, len(alphabet))
            g = all_gather_units(packed.reshape(c_hi - c_lo, F * packed.shape[1]), Cin, group)   # units = channels (a shard may be empty)
            Qk, Ik = hip.assemble_kernel(g.reshape(Cin * F, packed.shape[1]).contiguous(), alphabet, bits=bits, N=K)
            Q = Qk.reshape(kh, kw, Cin, F)
            idx = Ik.reshape(kh, kw, Cin, F)
            if want_resid:
                Rc = all_gather_units(Rc[c_lo:c_hi].contiguous(), Cin, group)
        else:
            rows = Ic[:, f_lo:f_hi].transpose(0, 1).reshape(-1, K).contiguous()    # (f, c) rows of this rank's filters
            packed, bits = hip.pack_indices(rows, len(alphabet))
            g = all_gather_units(packed.reshape(f_hi - f_lo, Cin * packed.shape[1]), F, group)   # units = filters
            Qk, Ik = hip.assemble_kernel(g.reshape(F * Cin, packed.shape[1]).contiguous(), alphabet, bits=bits, N=K)
            Q = Qk.reshape(kh, kw, F, Cin).permute(0, 1, 3, 2).contiguous()
            idx = Ik.reshape(kh, kw, F, Cin).permute(0, 1, 3, 2).contiguous()
            if want_resid:
                Rc = all_gather_units(Rc[:, f_lo:f_hi].transpose(0, 1).contiguous(), F, group).transpose(0, 1)
        return dict(Q=Q, idx=idx, resid=Rc.contiguous(), reruns=torch.tensor(reruns))
    Q = Qc.reshape(Cin, F, kh, kw).permute(2, 3, 0, 1).contiguous()
    idx = Ic.reshape(Cin, F, kh, kw).permute(2, 3, 0, 1).contiguous()
    return dict(Q=Q, idx=idx, resid=Rc.contiguous(), reruns=torch.tensor(reruns))
