"""ctypes binding of libgpfq_hip.so (the C ABI of include/gpfq.h) on PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns device memory and the HIP stream; every compute call goes
through the C ABI with raw device pointers.  There is NO CPU fallback: if the library cannot be
loaded, or a tensor is not on a GPU, the call raises.
"""
import ctypes
import os

import torch

from . import build as _build

GPFQ_PATH_AUTO, GPFQ_PATH_ONCHIP, GPFQ_PATH_STREAM = 0, 1, 2
GPFQ_PATH_GRAM = 3                 # binding-level selector: gpfq_quantize_neurons_gram + exact rerun of flagged neurons
GPFQ_GRAM_AUTO_MAX_N = 64          # conv layers with kh*kw up to this take the whole-shard Gram call
GPFQ_GRAM_MAX_N = 1024             # longest walk the Gram path takes (include/gpfq.h)
GPFQ_MAX_ALPHABET = 256       # more than 64 members: int16 indices (index_dtype)
GPFQ_ONCHIP_MAX_M = 28672          # longest row whose residual stays in registers (include/gpfq.h)
GPFQ_GRAM_MIN_M = 16384
GPFQ_DEVICE_ALPHABET_BYTES = 1024
GPFQ_LAYOUT_NEURON_MAJOR, GPFQ_LAYOUT_KERAS = 0, 1
GPFQ_ERR_CLUSTER_TIMEOUT, GPFQ_ERR_ALPHABET = -6, -7

# every symbol include/gpfq.h declares: (restype, argtypes)
_i64, _int, _vp, _sz = ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
_dp = ctypes.POINTER(ctypes.c_double)
SYMBOLS = {
    "gpfq_version": (_int, []),
    "gpfq_last_error": (ctypes.c_char_p, []),
    "gpfq_device_count": (_int, []),
    "gpfq_row_norms": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "gpfq_workspace_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "gpfq_last_dense_kernel": (ctypes.c_char_p, []),
    "gpfq_set_main_kernel_events": (_int, [_vp, _vp]),
    "gpfq_set_option": (_int, [ctypes.c_char_p, _int]),
    "gpfq_quantize_neurons": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _dp, _int, _int, _i64, _i64, _i64,
                                     _vp, _vp, _vp, _vp, _vp, _sz, _int, _vp]),
    "gpfq_call_status": (_int, [_vp, _vp]),
    "gpfq_layer_alphabet_device": (_int, [_vp, ctypes.c_double, _dp, _int, _vp, _vp]),
    "gpfq_layer_alphabet_from_kernel": (_int, [_vp, _i64, ctypes.c_double, _dp, _int, _vp, _vp, _vp, _sz, _vp]),
    "gpfq_dense_layer_supported": (_int, [_i64, _i64, _i64, _dp, _int]),
    "gpfq_dense_layer_keras_out_supported": (_int, [_i64, _i64, _i64, _dp, _int]),
    "gpfq_dense_layer_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gpfq_quantize_dense_layer": (_int, [_vp, _vp, _i64, _vp, _vp, _i64, _i64, _i64, _vp, _dp, _int, _i64, _i64,
                                         _vp, _vp, _int, _i64, _vp, _vp, _sz, _vp]),
    "gpfq_dense_layer_prepare": (_int, [_vp, _vp, _i64, _vp, _dp, _int, _i64, _i64, _i64, _vp, _sz, _vp]),
    "gpfq_dense_layer_run": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _dp, _int, _i64, _i64,
                                    _vp, _vp, _int, _i64, _vp, _vp, _sz, _vp]),
    "gpfq_assemble_kernel_device": (_int, [_vp, _int, _vp, _int, _i64, _i64, _vp, _vp, _vp]),
    "gpfq_gram_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "gpfq_quantize_neurons_gram": (_int, [_vp, _vp, _i64, _vp, _int, _vp, _i64, _dp, _int, _int, _i64, _i64, _i64,
                                          _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpfq_channel_planes": (_int, [_vp, _i64, _i64, _i64, _i64, _vp, _vp]),
    "gpfq_conv3x3_nhwc_supported": (_int, [_i64, _i64, _i64, _i64]),
    "gpfq_conv3x3_nhwc_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i64]),
    "gpfq_quantize_conv3x3_nhwc": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _vp, _dp, _int, _int, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpfq_channel_sumsq_workspace_bytes": (_sz, [_i64]),
    "gpfq_channel_sumsq": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _vp, _sz, _vp]),
    "gpfq_channel_dead_workspace_bytes": (_sz, [_i64]),
    "gpfq_channel_dead": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _int, _i64, _vp, _vp, _sz, _vp]),
    "gpfq_conv_channels_nhwc_supported": (_int, [_i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int]),
    "gpfq_quantize_conv_channels_nhwc": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int,
                                                 _vp, _vp, _int, _int, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpfq_conv1x1_workspace_bytes": (_sz, [_i64]),
    "gpfq_quantize_conv1x1": (_int, [_vp, _i64, _i64, _i64, _i64, _int, _int, _vp, _i64, _vp, _int, _vp, _vp, _vp, _sz, _vp]),
    "gpfq_conv_channels_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int, _i64, _int]),
    "gpfq_quantize_conv_channels": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int,
                                           _vp, _dp, _int, _int, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpfq_conv_records_supported": (_int, [_i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int]),
    "gpfq_conv_channel_records": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int,
                                         _vp, _vp, _vp, _sz, _vp]),
    "gpfq_quantize_conv_channels_from_records": (_int, [_vp, _vp, _vp, _vp, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int,
                                                        _int, _int, _vp, _dp, _int, _int, _i64, _vp, _vp, _vp, _vp, _sz, _vp]),
    "gpfq_msq_round": (_int, [_vp, _i64, _dp, _int, _vp, _vp, _vp]),
    "gpfq_index_bits": (_int, [_int]),
    "gpfq_pack_indices": (_int, [_vp, _i64, _i64, _int, _vp, _vp]),
    "gpfq_assemble_kernel": (_int, [_vp, _int, _dp, _int, _i64, _i64, _vp, _vp, _vp]),
    "gpfq_median_abs_workspace_bytes": (_sz, []),
    "gpfq_median_abs_workspace_bytes_for": (_sz, [_i64]),
    "gpfq_median_abs": (_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    "gpfq_median_abs_begin": (_int, [_i64, _vp, _sz, _vp]),
    "gpfq_median_abs_count": (_int, [_vp, _i64, _i64, _int, _vp, _vp]),
    "gpfq_median_abs_pick": (_int, [_i64, _int, _vp, _vp]),
    "gpfq_median_abs_end": (_int, [_i64, _vp, _vp, _vp]),
    "gpfq_patch_out_dim": (_i64, [_i64, _i64, _i64, _i64, _int]),
    "gpfq_extract_patches": (_int, [_vp, _i64, _i64, _i64, _i64, _i64, _int, _int, _int, _int, _int, _int, _int,
                                    _vp, _i64, _vp]),
}

_lib = None


class GpfqError(RuntimeError):
    pass


def lib_path():
    return _build.LIB


ABI_VERSION = 306                  # gpfq_version() of the library this binding was written against


def load():
    """Load libgpfq_hip.so (must already be built in-tree: __graft_entry__.build() / build.py).

    A library built from other sources or flags than the tree holds (content hash, build.tree_hash) is rebuilt when
    hipcc is at hand and refused otherwise; so is one whose gpfq_version() differs from ABI_VERSION."""
    global _lib
    if _lib is None:
        path = lib_path()
        if os.path.exists(path) and _build.built_hash() != _build.tree_hash():
            try:
                _build.build()
            except Exception as e:       # no hipcc here: do not run a stale binary silently
                raise GpfqError(f"{path} was built from other sources than this tree and cannot be rebuilt: {e}")
        if not os.path.exists(path):
            raise GpfqError(f"{path} is missing: run `python -m quantized_neural_networks_amd.build` "
                            "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        lib = ctypes.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)      # AttributeError if the ABI and the header disagree
            fn.restype, fn.argtypes = res, args
        if lib.gpfq_version() != ABI_VERSION:
            raise GpfqError(f"{path} reports ABI version {lib.gpfq_version()}, this binding expects {ABI_VERSION}")
        _lib = lib
    return _lib


def _check(rc, what):
    if rc != 0:
        raise GpfqError(f"{what} failed ({rc}): {load().gpfq_last_error().decode()}")


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise GpfqError(f"{name} must be a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise GpfqError(f"{name} must be {dtype}, got {t.dtype}")
    return t


def _rows(t, name):
    """2-D tensor with unit inner stride -> (ptr, rows, cols, pitch)."""
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise GpfqError(f"{name} must be 2-D with contiguous rows")
    pitch = t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)
    return t.data_ptr(), t.shape[0], t.shape[1], max(pitch, t.shape[1])


def index_dtype(M):
    """Element type of the index outputs for an alphabet of M members: int8 up to 64, int16 up to GPFQ_MAX_ALPHABET."""
    return torch.int8 if int(M) <= 64 else torch.int16


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _alphabet(alphabet):
    a = [float(v) for v in alphabet]
    if not 1 <= len(a) <= GPFQ_MAX_ALPHABET:
        raise GpfqError(f"alphabet size {len(a)} not in [1, {GPFQ_MAX_ALPHABET}]")
    arr = (ctypes.c_double * len(a))(*a)
    zero_idx = -1
    for k, v in enumerate(a):
        if v == 0.0:
            zero_idx = k
    return arr, len(a), zero_idx


def row_norms(Xq):
    """float32-rounded Euclidean norm of every row of Xq (f32 [N][m]) -> f32 [N]."""
    _dev(Xq, torch.float32, "Xq")
    ptr, N, m, ld = _rows(Xq, "Xq")
    out = torch.empty(N, dtype=torch.float32, device=Xq.device)
    with torch.cuda.device(Xq.device):
        _check(load().gpfq_row_norms(ptr, N, m, ld, out.data_ptr(), _stream()), "gpfq_row_norms")
    return out


def auto_path(N, m, C, want_u=False, M=0):
    """What GPFQ_PATH_AUTO resolves to for C neurons of N weights over rows of m samples.

    Rows beyond GPFQ_GRAM_MIN_M samples: the Gram path (N x N records + scalar recurrences) beats walking such rows
    step by step whenever the records are affordable (N <= GPFQ_GRAM_MAX_N); the reference's MNIST run is this case.
    Measured in round 2 (a probe since pruned; the numbers are in DESIGN.md, section 4, long walks): with many neurons the crossover already comes at about 8192 samples --
    N = 128, C = 1000, m = 8192: 0.94 ms on chip, 0.41 ms there; N = 784, C = 500, m = 12288: 5.5 vs 4.0 ms.
    Otherwise the residual stays on chip up to GPFQ_ONCHIP_MAX_M samples and streams through HBM beyond
    (the library chooses between those two itself)."""
    long_rows = m > GPFQ_GRAM_MIN_M or (m > GPFQ_GRAM_MIN_M // 2 and C * m >= 5_000_000)
    # (alphabets beyond 64 members have no wavefront-per-neuron chain for walks beyond 64 steps: they stay element-wise)
    if not want_u and long_rows and N <= GPFQ_GRAM_MAX_N and not (M > 64 and N > 64):
        return GPFQ_PATH_GRAM
    return GPFQ_PATH_ONCHIP if m <= GPFQ_ONCHIP_MAX_M else GPFQ_PATH_STREAM


def quantize_neurons(X, Xq, Wt, alphabet, nrm32=None, want_u=False, path=GPFQ_PATH_AUTO, want_values=True,
                     want_resid=True):
    """Greedy recurrence for all C neurons (rows of Wt [C][N]) against X, Xq [N][m].

    Returns dict(idx=i8 [C][N], Q=f32 [C][N], resid=f64 [C], u=f64 [C][m] or None).
    want_resid=None: residual norms only where they come for free (the kernels that hold u); the Gram path, which
    would replay the residual in an extra pass, then returns NaN for them.
    """
    _dev(X, torch.float32, "X"); _dev(Xq, torch.float32, "Xq"); _dev(Wt, torch.float32, "Wt")
    xp, N, m, ld = _rows(X, "X")
    xqp, N2, m2, ld2 = _rows(Xq, "Xq")
    wp, C, Nw, ldw = _rows(Wt, "Wt")
    if (N2, m2) != (N, m) or Nw != N:
        raise GpfqError(f"shape mismatch: X {tuple(X.shape)}, Xq {tuple(Xq.shape)}, Wt {tuple(Wt.shape)}")
    if ld2 != ld:
        raise GpfqError("X and Xq must share one row pitch")
    arr, M, zero_idx = _alphabet(alphabet)
    dev = X.device
    if path == GPFQ_PATH_GRAM or (path == GPFQ_PATH_AUTO and auto_path(N, m, C, want_u, M) == GPFQ_PATH_GRAM):
        return _quantize_neurons_gram(X, Xq, Wt, alphabet, nrm32, want_values, bool(want_resid))
    if nrm32 is None:
        nrm32 = row_norms(Xq)
    _dev(nrm32, torch.float32, "nrm32")
    idx = torch.empty((C, N), dtype=index_dtype(M), device=dev)
    Q = torch.empty((C, N), dtype=torch.float32, device=dev) if want_values else None
    resid = torch.empty(C, dtype=torch.float64, device=dev)
    lib = load()
    nbytes = lib.gpfq_workspace_bytes(N, m, C, path)
    streaming = path == GPFQ_PATH_STREAM or (path == GPFQ_PATH_AUTO and m > GPFQ_ONCHIP_MAX_M)
    # (the streaming path keeps its residual in the workspace, which gpfq_workspace_bytes sizes for it: a second
    #  C x m float64 tensor here would double the largest allocation of the call)
    u = torch.empty((C, m), dtype=torch.float64, device=dev) if want_u else None
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gpfq_quantize_neurons(xp, xqp, ld, nrm32.data_ptr(), wp, ldw, arr, M, zero_idx, N, m, C,
                                       idx.data_ptr(), Q.data_ptr() if Q is not None else None, resid.data_ptr(),
                                       u.data_ptr() if u is not None else None,
                                       ws.data_ptr(), nbytes, path, _stream())
    _check(rc, "gpfq_quantize_neurons")
    # on-chip path: the first 8 workspace bytes count decisions re-derived exactly (diagnostics)
    return dict(idx=idx, Q=Q, resid=resid, u=u if want_u else None, workspace=None if streaming else ws)


def _quantize_neurons_gram(X, Xq, Wt, alphabet, nrm32, want_values=True, want_resid=True):
    """Gram-matrix path (short walks over long rows) + exact rerun of the uncertified neurons."""
    xp, N, m, ld = _rows(X, "X")
    xqp, _, _, _ = _rows(Xq, "Xq")
    wp, C, _, ldw = _rows(Wt, "Wt")
    arr, M, zero_idx = _alphabet(alphabet)
    dev = X.device
    lib = load()
    idx = torch.empty((C, N), dtype=index_dtype(M), device=dev)
    Q = torch.empty((C, N), dtype=torch.float32, device=dev)
    resid = torch.empty(C, dtype=torch.float64, device=dev) if want_resid else \
        torch.full((C,), float("nan"), dtype=torch.float64, device=dev)
    unc = torch.empty(C, dtype=torch.int32, device=dev)
    compute_norms = nrm32 is None            # the row norms are the Gram diagonal: no separate pass
    if compute_norms:
        nrm32 = torch.empty(N, dtype=torch.float32, device=dev)
    _dev(nrm32, torch.float32, "nrm32")
    nbytes = lib.gpfq_gram_workspace_bytes(N, m, C)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gpfq_quantize_neurons_gram(xp, xqp, ld, nrm32.data_ptr(), 1 if compute_norms else 0, wp, ldw, arr, M, zero_idx, N, m, C,
                                            idx.data_ptr(), Q.data_ptr(), resid.data_ptr() if want_resid else None,
                                            unc.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_quantize_neurons_gram")
    bad = torch.nonzero(unc).flatten()                    # one sync per call
    if bad.numel():
        r2 = quantize_neurons(X, Xq, Wt[bad].contiguous(), alphabet, nrm32=nrm32, path=GPFQ_PATH_STREAM)
        idx[bad], Q[bad], resid[bad] = r2["idx"], r2["Q"], r2["resid"]
    return dict(idx=idx, Q=Q if want_values else None, resid=resid, u=None, workspace=None,
                uncertified=int(bad.numel()))


class GramPlan:
    """Repeated gpfq_quantize_neurons_gram calls of one shape (the channels of a conv layer) without
    per-call allocation or synchronisation: scratch, the row-norm buffer and the alphabet are set up
    once; run() writes into caller-provided (views of) output tensors and flags uncertified neurons."""

    def __init__(self, N, m, C, alphabet, device):
        self.N, self.m, self.C, self.dev = int(N), int(m), int(C), device
        self.arr, self.M, self.zero_idx = _alphabet(alphabet)
        self.lib = load()
        self.nbytes = self.lib.gpfq_gram_workspace_bytes(self.N, self.m, self.C)
        self.ws = torch.empty(max(self.nbytes, 16), dtype=torch.uint8, device=device)
        self.nrm = torch.empty(max(self.N, 1), dtype=torch.float32, device=device)

    def run(self, X, Xq, Wt, idx, Q, resid, unc):
        xp, N, m, ld = _rows(X, "X")
        xqp, _, _, ld2 = _rows(Xq, "Xq")
        wp, C, Nw, ldw = _rows(Wt, "Wt")
        if (N, m, C, Nw) != (self.N, self.m, self.C, self.N) or ld2 != ld:
            raise GpfqError("GramPlan.run: shape differs from the plan")
        for t, dt, shape in ((idx, index_dtype(self.M), (C, N)), (Q, torch.float32, (C, N)), (resid, torch.float64, (C,)),
                             (unc, torch.int32, (C,))):
            if t.dtype != dt or tuple(t.shape) != shape or not t.is_contiguous() or not t.is_cuda:
                raise GpfqError("GramPlan.run: outputs must be contiguous GPU tensors of the planned shape")
        with torch.cuda.device(self.dev):
            rc = self.lib.gpfq_quantize_neurons_gram(xp, xqp, ld, self.nrm.data_ptr(), 1, wp, ldw, self.arr, self.M,
                                                     self.zero_idx, N, m, C, idx.data_ptr(), Q.data_ptr(),
                                                     resid.data_ptr(), unc.data_ptr(), self.ws.data_ptr(), self.nbytes,
                                                     _stream())
        _check(rc, "gpfq_quantize_neurons_gram")


def channel_planes(act, c_lo, c_hi):
    """NHWC f32 [n][H][W][Cin] -> channel-major f32 [c_hi - c_lo][n][H][W] (gpfq_channel_planes)."""
    _dev(act, torch.float32, "act")
    if act.dim() != 4 or not act.is_contiguous():
        raise GpfqError("channel_planes needs a contiguous NHWC tensor")
    n, H, W, Cin = act.shape
    out = torch.empty((c_hi - c_lo, n, H, W), dtype=torch.float32, device=act.device)
    with torch.cuda.device(act.device):
        rc = load().gpfq_channel_planes(act.data_ptr(), n * H * W, Cin, c_lo, c_hi - c_lo, out.data_ptr(), _stream())
    _check(rc, "gpfq_channel_planes")
    return out


def channel_sumsq(act, strides=(1, 1)):
    """f64 [Cin]: squared norms of the channels of NHWC f32 activations over the positions a (1, 1) kernel with these strides
    visits (gpfq_channel_sumsq): one pass over the tensor, no channel-major copy."""
    _dev(act, torch.float32, "act")
    if act.dim() != 4 or not act.is_contiguous():
        raise GpfqError("channel_sumsq needs a contiguous NHWC tensor")
    n, H, W, Cin = act.shape
    lib = load()
    nbytes = lib.gpfq_channel_sumsq_workspace_bytes(Cin)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=act.device)
    out = torch.empty(Cin, dtype=torch.float64, device=act.device)
    with torch.cuda.device(act.device):
        rc = lib.gpfq_channel_sumsq(act.data_ptr(), n, H, W, Cin, int(strides[0]), int(strides[1]), out.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_channel_sumsq")
    return out


def channel_dead(act, strides=(1, 1), prefix_positions=0):
    """bool [Cin]: channels of NHWC f32 activations whose float32-rounded norm over the positions a (1, 1) kernel with these
    strides visits is below 1e-16 (gpfq_channel_dead): a prefix of the positions decides every live channel, only channels
    still undecided are summed in full.  No sync."""
    _dev(act, torch.float32, "act")
    if act.dim() != 4 or not act.is_contiguous():
        raise GpfqError("channel_dead needs a contiguous NHWC tensor")
    n, H, W, Cin = act.shape
    lib = load()
    nbytes = lib.gpfq_channel_dead_workspace_bytes(Cin)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=act.device)
    out = torch.empty(Cin, dtype=torch.int32, device=act.device)
    with torch.cuda.device(act.device):
        rc = lib.gpfq_channel_dead(act.data_ptr(), n, H, W, Cin, int(strides[0]), int(strides[1]), int(prefix_positions),
                                   out.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_channel_dead")
    return out != 0


def quantize_conv1x1(act_q, W2, alphabet, strides=(1, 1)):
    """A conv layer of kernel_size (1, 1) in one call (gpfq_quantize_conv1x1): act_q NHWC f32, W2 f32 [Cin][F] ->
    (Q f32 [Cin][F], idx i8 / i16 [Cin][F]); dead channels take 0 / the alphabet's zero index.  No sync."""
    _dev(act_q, torch.float32, "act_q")
    _dev(W2, torch.float32, "W2")
    if act_q.dim() != 4 or not act_q.is_contiguous() or W2.dim() != 2 or not W2.is_contiguous() or W2.shape[0] != act_q.shape[3]:
        raise GpfqError("quantize_conv1x1 needs a contiguous NHWC tensor and a contiguous [Cin][F] kernel")
    n, H, W, Cin = act_q.shape
    F = W2.shape[1]
    arr, M, _ = _alphabet(alphabet)
    lib = load()
    nbytes = lib.gpfq_conv1x1_workspace_bytes(Cin)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=act_q.device)
    Q = torch.empty((Cin, F), dtype=torch.float32, device=act_q.device)
    idx = torch.empty((Cin, F), dtype=index_dtype(M), device=act_q.device)
    with torch.cuda.device(act_q.device):
        rc = lib.gpfq_quantize_conv1x1(act_q.data_ptr(), n, H, W, Cin, int(strides[0]), int(strides[1]), W2.data_ptr(), F, arr, M,
                                       Q.data_ptr(), idx.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_quantize_conv1x1")
    return Q, idx


def neuron_major(W, lo=0, hi=None):
    """Keras kernel W f32 [N][C] -> the neuron-major shard Wt f32 [hi - lo][N] = W[:, lo:hi].T (what the pool hands its workers,
    scripts/quantized_network.py:553-556).  The same LDS-tiled transposing copy as channel_planes (gpfq_channel_planes with the
    neurons as "channels"): ~30 us for a 4096 x 4096 kernel against ~65 us for torch's strided copy."""
    _dev(W, torch.float32, "W")
    if W.dim() != 2 or not W.is_contiguous():
        raise GpfqError("neuron_major needs a contiguous [N][C] kernel")
    N, C = W.shape
    hi = C if hi is None else hi
    out = torch.empty((hi - lo, N), dtype=torch.float32, device=W.device)
    if hi > lo and N > 0:
        with torch.cuda.device(W.device):
            rc = load().gpfq_channel_planes(W.data_ptr(), N, C, lo, hi - lo, out.data_ptr(), _stream())
        _check(rc, "gpfq_channel_planes")
    return out


def quantize_conv_channels(act_w_cm, act_q_cm, Wt_all, alphabet, kernel_size, strides, rate, padding,
                           idx, Q, resid, unc):
    """All channels of a conv layer shard in one library call (gpfq_quantize_conv_channels).
    act_*_cm: channel-major f32 [nch][n][H][W]; Wt_all f32 [nch][F][K]; outputs are caller tensors
    idx i8 / Q f32 [nch][F][K], resid f64 [nch][F], unc i32 [nch][F].  No sync; returns nothing."""
    for t, dt in ((act_w_cm, torch.float32), (act_q_cm, torch.float32), (Wt_all, torch.float32),
                  (idx, index_dtype(len(alphabet))), (Q, torch.float32), (resid, torch.float64), (unc, torch.int32)):
        if t is None:
            continue                                  # resid is optional (skips the exact replay)
        _dev(t, dt, "tensor")
        if not t.is_contiguous():
            raise GpfqError("quantize_conv_channels needs contiguous tensors")
    nch, n, H, W = act_w_cm.shape
    kh, kw = kernel_size
    sh, sw = strides
    rh, rw = rate if rate else (1, 1)
    same = 1 if str(padding).upper() == "SAME" else 0
    F, K = Wt_all.shape[1], Wt_all.shape[2]
    if (tuple(act_q_cm.shape) != (nch, n, H, W) or Wt_all.shape[0] != nch or K != kh * kw or tuple(idx.shape) != (nch, F, K)
            or tuple(Q.shape) != (nch, F, K) or (resid is not None and tuple(resid.shape) != (nch, F))
            or tuple(unc.shape) != (nch, F)):
        raise GpfqError("quantize_conv_channels: shape mismatch")
    arr, M, zero_idx = _alphabet(alphabet)
    lib = load()
    nbytes = lib.gpfq_conv_channels_workspace_bytes(n, H, W, nch, kh, kw, sh, sw, rh, rw, same, F, 0 if resid is None else 1)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=act_w_cm.device)
    with torch.cuda.device(act_w_cm.device):
        rc = lib.gpfq_quantize_conv_channels(act_w_cm.data_ptr(), act_q_cm.data_ptr(), n, H, W, nch, kh, kw, sh, sw, rh, rw,
                                             same, Wt_all.data_ptr(), arr, M, zero_idx, F, idx.data_ptr(), Q.data_ptr(),
                                             resid.data_ptr() if resid is not None else None, unc.data_ptr(),
                                             ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_quantize_conv_channels")


def conv_channels_nhwc_supported(n, H, W, nch, kernel_size, strides, rate, padding):
    """Whether quantize_conv_channels_nhwc has a kernel for this layer shape (today: 7 x 7 / stride 2 / VALID, the shift-sum form)."""
    rh, rw = rate if rate else (1, 1)
    return bool(load().gpfq_conv_channels_nhwc_supported(int(n), int(H), int(W), int(nch), int(kernel_size[0]), int(kernel_size[1]),
                                                        int(strides[0]), int(strides[1]), int(rh), int(rw),
                                                        1 if str(padding).upper() == "SAME" else 0))


def quantize_conv_channels_nhwc(act_w, act_q, c_lo, c_hi, Wt_all, alphabet, kernel_size, strides, rate, padding, idx, Q, unc):
    """Channels [c_lo, c_hi) of a conv layer whose kernel reads the NHWC activations itself (gpfq_quantize_conv_channels_nhwc): no
    channel-major copy.  act_* f32 [n][H][W][Cin]; Wt_all f32 [nch][F][K]; outputs are caller tensors idx / Q [nch][F][K],
    unc i32 [nch][F].  No sync; returns nothing."""
    for t, dt in ((act_w, torch.float32), (act_q, torch.float32), (Wt_all, torch.float32), (idx, index_dtype(len(alphabet))),
                  (Q, torch.float32), (unc, torch.int32)):
        _dev(t, dt, "tensor")
        if not t.is_contiguous():
            raise GpfqError("quantize_conv_channels_nhwc needs contiguous tensors")
    n, H, W, Cin = act_w.shape
    nch = c_hi - c_lo
    kh, kw = kernel_size
    sh, sw = strides
    rh, rw = rate if rate else (1, 1)
    same = 1 if str(padding).upper() == "SAME" else 0
    F, K = Wt_all.shape[1], Wt_all.shape[2]
    if (tuple(act_q.shape) != (n, H, W, Cin) or not 0 <= c_lo <= c_hi <= Cin or Wt_all.shape[0] != nch or K != kh * kw
            or tuple(idx.shape) != (nch, F, K) or tuple(Q.shape) != (nch, F, K) or tuple(unc.shape) != (nch, F)):
        raise GpfqError("quantize_conv_channels_nhwc: shape mismatch")
    arr, M, zero_idx = _alphabet(alphabet)
    lib = load()
    nbytes = lib.gpfq_conv_channels_workspace_bytes(n, H, W, nch, kh, kw, sh, sw, rh, rw, same, F, 0)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=act_w.device)
    with torch.cuda.device(act_w.device):
        rc = lib.gpfq_quantize_conv_channels_nhwc(act_w.data_ptr(), act_q.data_ptr(), n, H, W, Cin, c_lo, nch, kh, kw, sh, sw, rh, rw,
                                                  same, Wt_all.data_ptr(), arr, M, zero_idx, F, idx.data_ptr(), Q.data_ptr(),
                                                  unc.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_quantize_conv_channels_nhwc")


def conv3x3_nhwc_supported(n, H, W, nch):
    """Whether quantize_conv3x3_nhwc takes a shard of nch channels of n images of H x W."""
    return bool(load().gpfq_conv3x3_nhwc_supported(int(n), int(H), int(W), int(nch)))


def quantize_conv3x3_nhwc(act_w, act_q, c_lo, c_hi, Wt_all, alphabet, idx, Q, unc):
    """Channels [c_lo, c_hi) of a 3 x 3 / stride 1 / SAME conv layer straight from the NHWC activations
    (gpfq_quantize_conv3x3_nhwc): no channel-major copy.  act_* f32 [n][H][W][Cin]; Wt_all f32 [nch][F][9]; outputs are caller
    tensors idx / Q [nch][F][9], unc i32 [nch][F].  No sync; returns nothing."""
    for t, dt in ((act_w, torch.float32), (act_q, torch.float32), (Wt_all, torch.float32), (idx, index_dtype(len(alphabet))),
                  (Q, torch.float32), (unc, torch.int32)):
        _dev(t, dt, "tensor")
        if not t.is_contiguous():
            raise GpfqError("quantize_conv3x3_nhwc needs contiguous tensors")
    n, H, W, Cin = act_w.shape
    nch = c_hi - c_lo
    F = Wt_all.shape[1]
    if (tuple(act_q.shape) != (n, H, W, Cin) or tuple(Wt_all.shape) != (nch, F, 9) or tuple(idx.shape) != (nch, F, 9)
            or tuple(Q.shape) != (nch, F, 9) or tuple(unc.shape) != (nch, F)):
        raise GpfqError("quantize_conv3x3_nhwc: shape mismatch")
    arr, M, zero_idx = _alphabet(alphabet)
    lib = load()
    nbytes = lib.gpfq_conv3x3_nhwc_workspace_bytes(n, H, W, nch, F)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=act_w.device)
    with torch.cuda.device(act_w.device):
        rc = lib.gpfq_quantize_conv3x3_nhwc(act_w.data_ptr(), act_q.data_ptr(), n, H, W, Cin, c_lo, nch, Wt_all.data_ptr(), arr, M, zero_idx,
                                            F, idx.data_ptr(), Q.data_ptr(), unc.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_quantize_conv3x3_nhwc")


def conv_records_supported(n, H, W, nch, kernel_size, strides, rate, padding):
    """Whether conv_channel_records / conv_channels_from_records have a plane kernel for n images of H x W."""
    kh, kw = kernel_size
    sh, sw = strides
    rh, rw = rate if rate else (1, 1)
    same = 1 if str(padding).upper() == "SAME" else 0
    return bool(load().gpfq_conv_records_supported(int(n), int(H), int(W), int(nch), kh, kw, sh, sw, rh, rw, same))


def conv_channel_records(act_w_cm, act_q_cm, kernel_size, strides, rate, padding):
    """First half of quantize_conv_channels for column-sharded multi-GPU runs: the Gram records of all channels over
    THESE images (planes f32 [nch][n][H][W]).  Returns (records f64 [nch][K*K*2 + K], negflags i32 [nch]); both are
    summed / maximised over the ranks before conv_channels_from_records.  No sync."""
    for t in (act_w_cm, act_q_cm):
        _dev(t, torch.float32, "planes")
        if not t.is_contiguous():
            raise GpfqError("conv_channel_records needs contiguous planes")
    nch, n, H, W = act_w_cm.shape
    kh, kw = kernel_size
    sh, sw = strides
    rh, rw = rate if rate else (1, 1)
    same = 1 if str(padding).upper() == "SAME" else 0
    K = kh * kw
    dev = act_w_cm.device
    records = torch.zeros((nch, K * K * 2 + K), dtype=torch.float64, device=dev)
    negflags = torch.zeros((nch,), dtype=torch.int32, device=dev)
    if n == 0:
        return records, negflags
    lib = load()
    nbytes = lib.gpfq_conv_channels_workspace_bytes(n, H, W, nch, kh, kw, sh, sw, rh, rw, same, 0, 0)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.gpfq_conv_channel_records(act_w_cm.data_ptr(), act_q_cm.data_ptr(), n, H, W, nch, kh, kw, sh, sw, rh, rw, same,
                                           records.data_ptr(), negflags.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_conv_channel_records")
    return records, negflags


def conv_channels_from_records(records, negflags, act_w_cm, act_q_cm, Wt_all, alphabet, kernel_size, strides, rate, padding,
                               idx, Q, unc):
    """Second half: decide (and repair) all channels from summed records; planes of ALL images.  No sync."""
    nch, n, H, W = act_w_cm.shape
    kh, kw = kernel_size
    sh, sw = strides
    rh, rw = rate if rate else (1, 1)
    same = 1 if str(padding).upper() == "SAME" else 0
    F, K = Wt_all.shape[1], Wt_all.shape[2]
    if (tuple(records.shape) != (nch, K * K * 2 + K) or tuple(negflags.shape) != (nch,) or K != kh * kw
            or tuple(idx.shape) != (nch, F, K) or tuple(Q.shape) != (nch, F, K) or tuple(unc.shape) != (nch, F)):
        raise GpfqError("conv_channels_from_records: shape mismatch")
    for t, dt in ((records, torch.float64), (negflags, torch.int32), (act_w_cm, torch.float32), (act_q_cm, torch.float32),
                  (Wt_all, torch.float32), (idx, index_dtype(len(alphabet))), (Q, torch.float32), (unc, torch.int32)):
        _dev(t, dt, "tensor")
        if not t.is_contiguous():
            raise GpfqError("conv_channels_from_records needs contiguous tensors")
    arr, M, zero_idx = _alphabet(alphabet)
    lib = load()
    nbytes = lib.gpfq_conv_channels_workspace_bytes(n, H, W, nch, kh, kw, sh, sw, rh, rw, same, F, 0)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=act_w_cm.device)
    with torch.cuda.device(act_w_cm.device):
        rc = lib.gpfq_quantize_conv_channels_from_records(records.data_ptr(), negflags.data_ptr(), act_w_cm.data_ptr(),
                                                          act_q_cm.data_ptr(), n, H, W, nch, kh, kw, sh, sw, rh, rw, same,
                                                          Wt_all.data_ptr(), arr, M, zero_idx, F, idx.data_ptr(), Q.data_ptr(),
                                                          unc.data_ptr(), ws.data_ptr(), nbytes, _stream())
    _check(rc, "gpfq_quantize_conv_channels_from_records")


class DeviceAlphabet:
    """The layer alphabet rad * unit (scripts/quantized_network.py:544-545) held in DEVICE memory (gpfq_layer_alphabet_device): formed by
    one single-thread kernel from the float32 device scalar median(|W|), so that no launch of the layer waits for the radius to cross to
    the host.  `buf`: the GPFQ_DEVICE_ALPHABET_BYTES block the kernels read; `unit`: linspace(-1, 1, M) on the host (M, the zero member
    and the symmetric form are decided from it).  rad() / values() read the radius back (one sync: logging, the reference's return value)."""

    def __init__(self, buf, unit, alphabet_scalar):
        import numpy as np
        self.buf, self.unit, self.alphabet_scalar = buf, np.asarray(unit, dtype=np.float64), float(alphabet_scalar)
        self._rad = None
        self.radius_ok = False        # set by a caller that KNOWS the radius is finite and positive (it read the median): quantize_dense then
                                      # has no deferred alphabet status to wait for

    def __len__(self):
        return len(self.unit)

    def rad(self):
        import numpy as np
        if self._rad is None:
            self._rad = np.float64(self.buf[:8].view(torch.float64).cpu().item())
        return self._rad

    def values(self):
        """rad * unit as float64: the host alphabet of the same layer (bit-identical to the device's members: the same float64 products)."""
        return self.rad() * self.unit


def layer_alphabet_device(median32, unit_alphabet, alphabet_scalar):
    """DeviceAlphabet of a layer from the float32 DEVICE scalar median32 = median(|W|) (median_abs(..., on_device=True)).  No sync."""
    import numpy as np
    _dev(median32, torch.float32, "median32")
    unit = np.asarray(unit_alphabet, dtype=np.float64)
    if not 1 <= len(unit) <= 64:
        raise GpfqError(f"device-resident alphabets hold 1..64 members, got {len(unit)}")
    arr = (ctypes.c_double * len(unit))(*[float(v) for v in unit])
    buf = torch.empty(GPFQ_DEVICE_ALPHABET_BYTES, dtype=torch.uint8, device=median32.device)
    with torch.cuda.device(median32.device):
        _check(load().gpfq_layer_alphabet_device(median32.data_ptr(), float(alphabet_scalar), arr, len(unit), buf.data_ptr(), _stream()),
               "gpfq_layer_alphabet_device")
    return DeviceAlphabet(buf, unit, alphabet_scalar)


def layer_alphabet_from_kernel(W, unit_alphabet, alphabet_scalar):
    """DeviceAlphabet of a layer straight from its float32 kernel W (any shape, contiguous): median(|W|) and the alphabet in one library
    call, the alphabet formed by the last workgroup of the median's second pass (gpfq_layer_alphabet_from_kernel).  No sync."""
    import numpy as np
    _dev(W, torch.float32, "W")
    Wc = W.contiguous()
    unit = np.asarray(unit_alphabet, dtype=np.float64)
    if not 1 <= len(unit) <= 64:
        raise GpfqError(f"device-resident alphabets hold 1..64 members, got {len(unit)}")
    if Wc.data_ptr() % 16 != 0:
        return layer_alphabet_device(median_abs(Wc.reshape(-1), on_device=True), unit, alphabet_scalar)
    arr = (ctypes.c_double * len(unit))(*[float(v) for v in unit])
    lib = load()
    nbytes = lib.gpfq_median_abs_workspace_bytes_for(Wc.numel())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
    buf = torch.empty(GPFQ_DEVICE_ALPHABET_BYTES, dtype=torch.uint8, device=W.device)
    with torch.cuda.device(W.device):
        _check(lib.gpfq_layer_alphabet_from_kernel(Wc.data_ptr(), Wc.numel(), float(alphabet_scalar), arr, len(unit), buf.data_ptr(), None,
                                                   ws.data_ptr(), nbytes, _stream()), "gpfq_layer_alphabet_from_kernel")
    return DeviceAlphabet(buf, unit, alphabet_scalar)


def dense_layer_supported(N, m, C, unit_alphabet):
    """Whether quantize_dense_layer (device-resident alphabet, the block-pipelined kernel) takes this shape and unit alphabet."""
    arr = (ctypes.c_double * len(unit_alphabet))(*[float(v) for v in unit_alphabet])
    return bool(load().gpfq_dense_layer_supported(int(N), int(m), int(C), arr, len(unit_alphabet)))


def _dense_layer_args(X, Xq, W, unit, lo, hi):
    _dev(X, torch.float32, "X"); _dev(Xq, torch.float32, "Xq")
    xp, N, m, ld = _rows(X, "X")
    xqp, N2, m2, ld2 = _rows(Xq, "Xq")
    if (N2, m2) != (N, m) or ld2 != ld:
        raise GpfqError(f"shape mismatch: X {tuple(X.shape)}, Xq {tuple(Xq.shape)} (one row pitch)")
    Ctot = None
    if W is not None:
        _dev(W, torch.float32, "W")
        if W.dim() != 2 or not W.is_contiguous() or W.shape[0] != N:
            raise GpfqError(f"W {tuple(W.shape)} must be the contiguous [N][C] Keras kernel of X's {N} input features")
        Ctot = W.shape[1]
        hi = Ctot if hi is None else hi
        if not 0 <= lo <= hi <= Ctot:
            raise GpfqError(f"neuron range [{lo}, {hi}) outside the layer's {Ctot}")
    arr = (ctypes.c_double * len(unit))(*[float(v) for v in unit])
    return xp, xqp, ld, N, m, Ctot, hi, arr


def dense_layer_workspace(N, m, C, device):
    """The workspace of a quantize_dense_layer / dense_layer_prepare + dense_layer_run call (status words in its first 16 bytes)."""
    nbytes = load().gpfq_dense_layer_workspace_bytes(int(N), int(m), int(C))
    return torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)


def dense_layer_prepare(X, Xq, unit_alphabet, C, ws, nrm32=None):
    """The alphabet-independent half of quantize_dense_layer (gpfq_dense_layer_prepare): status block, row norms, record pre-pass, on the
    current stream, into `ws` (dense_layer_workspace).  Needs neither the kernel nor the alphabet: a caller overlaps it with the median."""
    xp, xqp, ld, N, m, _, _, arr = _dense_layer_args(X, Xq, None, unit_alphabet, 0, None)
    if nrm32 is not None:
        _dev(nrm32, torch.float32, "nrm32")
    with torch.cuda.device(X.device):
        rc = load().gpfq_dense_layer_prepare(xp, xqp, ld, nrm32.data_ptr() if nrm32 is not None else None, arr, len(unit_alphabet), N, m, int(C),
                                             ws.data_ptr(), ws.numel(), _stream())
    _check(rc, "gpfq_dense_layer_prepare")


def quantize_dense_layer(X, Xq, W, dalpha, lo=0, hi=None, nrm32=None, keras_out=True, want_values=True, want_idx=True, want_resid=True,
                         prepared=None):
    """Neurons [lo, hi) of a Dense layer in one library call (gpfq_quantize_dense_layer): W f32 [N][C] is the Keras kernel itself, the
    alphabet a DeviceAlphabet.  keras_out: Q f32 / idx i8 are whole-layer [N][C] tensors of which columns lo..hi are written (the layout
    set_weights takes, scripts/quantized_network.py:562, :570); else this shard's neuron-major [hi - lo][N].
    prepared: the workspace a dense_layer_prepare call for the same X, Xq and hi - lo neurons has filled (the call is then
    gpfq_dense_layer_run: the alphabet-dependent half only).
    Returns dict(Q, idx, resid f64 [hi - lo], workspace); call_status(result) is the deferred error check.  No sync."""
    xp, xqp, ld, N, m, Ctot, hi, arr = _dense_layer_args(X, Xq, W, dalpha.unit, lo, hi)
    C = hi - lo
    M = len(dalpha)
    dev = X.device
    lib = load()
    # The kernel writes the Keras layout itself in the 16-neuron four-step shapes; elsewhere its outputs are neuron-major and one assembly
    # pass (members from the device alphabet) lays them out
    direct = keras_out and C > 0 and bool(lib.gpfq_dense_layer_keras_out_supported(N, m, C, arr, M))
    via_assembly = keras_out and not direct
    shape = (N, Ctot) if direct else (C, N)
    idx = torch.empty(shape, dtype=torch.int8, device=dev) if (want_idx or via_assembly) else None
    Q = torch.empty(shape, dtype=torch.float32, device=dev) if (want_values and not via_assembly) else None
    resid = torch.empty(C, dtype=torch.float64, device=dev) if want_resid is not False else None
    if C == 0:                                            # (an empty shard: nothing launched, a clean status block)
        if keras_out:
            idx = torch.empty((N, Ctot), dtype=torch.int8, device=dev) if want_idx else None
            Q = torch.empty((N, Ctot), dtype=torch.float32, device=dev) if want_values else None
        return dict(idx=idx, Q=Q, resid=resid, u=None, workspace=torch.zeros(16, dtype=torch.uint8, device=dev))
    ws = prepared if prepared is not None else dense_layer_workspace(N, m, C, dev)
    with torch.cuda.device(dev):
        out_args = (idx.data_ptr() if idx is not None else None, Q.data_ptr() if Q is not None else None,
                    GPFQ_LAYOUT_KERAS if direct else GPFQ_LAYOUT_NEURON_MAJOR, Ctot if direct else N, resid.data_ptr() if resid is not None else None,
                    ws.data_ptr(), ws.numel(), _stream())
        if prepared is not None:
            rc = lib.gpfq_dense_layer_run(xp, xqp, ld, W.data_ptr(), Ctot, lo, C, dalpha.buf.data_ptr(), arr, M, N, m, *out_args)
        else:
            if nrm32 is not None:
                _dev(nrm32, torch.float32, "nrm32")
            rc = lib.gpfq_quantize_dense_layer(xp, xqp, ld, nrm32.data_ptr() if nrm32 is not None else None,
                                               W.data_ptr(), Ctot, lo, C, dalpha.buf.data_ptr(), arr, M, N, m, *out_args)
    _check(rc, "gpfq_dense_layer_run" if prepared is not None else "gpfq_quantize_dense_layer")
    if via_assembly:
        Qk, Ik = assemble_kernel_device(idx, dalpha, want_idx=want_idx)
        if (lo, hi) != (0, Ctot):                         # a shard of a wider layer: its columns of whole-layer tensors, as the direct form writes them
            Qf = torch.empty((N, Ctot), dtype=torch.float32, device=dev) if want_values else None
            If = torch.empty((N, Ctot), dtype=torch.int8, device=dev) if want_idx else None
            if Qf is not None:
                Qf[:, lo:hi] = Qk
            if If is not None:
                If[:, lo:hi] = Ik
            Qk, Ik = Qf, If
        return dict(idx=Ik, Q=Qk if want_values else None, resid=resid, u=None, workspace=ws)
    return dict(idx=idx, Q=Q, resid=resid, u=None, workspace=ws)


def call_status(result):
    """Deferred errors of an asynchronous dense call (gpfq_call_status; forces a device sync): 0, GPFQ_ERR_CLUSTER_TIMEOUT (an exchange
    of the block kernel's cluster form timed out: the outputs are invalid) or GPFQ_ERR_ALPHABET (a device-resident alphabet whose radius
    was 0 / infinite / NaN: nothing was computed).  The layer drivers check it before a layer's result is used (layer.quantize_dense)."""
    ws = result.get("workspace") if isinstance(result, dict) else result
    if ws is None or ws.numel() < 16:
        return 0
    with torch.cuda.device(ws.device):
        return int(load().gpfq_call_status(ws.data_ptr(), _stream()))


def assemble_kernel_device(qidx, dalpha, want_idx=True, bits=8, N=None):
    """assemble_kernel with the members read from a DeviceAlphabet (gpfq_assemble_kernel_device): [C][N] int8 indices or rows packed by
    pack_indices (bits = 2 / 4, pass N) -> (Q f32 [N][C] in Keras layout, idx i8 [N][C])."""
    _dev(qidx, torch.int8 if bits == 8 else torch.uint8, "qidx")
    if qidx.dim() != 2 or not qidx.is_contiguous():
        raise GpfqError("qidx must be a contiguous 2-D tensor")
    C = qidx.shape[0]
    if bits >= 8:
        N = qidx.shape[1]
    elif N is None or qidx.shape[1] != (N * bits + 7) // 8:
        raise GpfqError("packed indices need N, with ceil(N*bits/8) bytes per row")
    Q = torch.empty((N, C), dtype=torch.float32, device=qidx.device)
    idx_t = torch.empty((N, C), dtype=torch.int8, device=qidx.device) if want_idx else None
    with torch.cuda.device(qidx.device):
        _check(load().gpfq_assemble_kernel_device(qidx.data_ptr(), bits, dalpha.buf.data_ptr(), len(dalpha), N, C, Q.data_ptr(),
                                                  idx_t.data_ptr() if idx_t is not None else None, _stream()),
               "gpfq_assemble_kernel_device")
    return Q, idx_t


def last_dense_kernel():
    """Name of the dense kernel family the last quantize_neurons() call dispatched (diagnostics)."""
    return load().gpfq_last_dense_kernel().decode()


def set_main_kernel_events(start=None, stop=None):
    """Measurement hook (gpfq_set_main_kernel_events): two torch.cuda.Event(enable_timing=True) that take the start and the end of the
    block-pipelined dense kernel's own dispatch -- the recurrence without the pre-passes of the same call (the kernel is launched with
    them: hipExtLaunchKernelGGL, no extra packet in the queue); start.elapsed_time(stop) afterwards.  None, None clears.  torch creates an
    event's handle at its first record(): events that have none are recorded once here (a benchmark does that before its timed region)."""
    if start is None or stop is None:
        _check(load().gpfq_set_main_kernel_events(None, None), "gpfq_set_main_kernel_events")
        return
    for e in (start, stop):
        if not e.cuda_event:
            e.record()
    _check(load().gpfq_set_main_kernel_events(ctypes.c_void_p(start.cuda_event), ctypes.c_void_p(stop.cuda_event)),
           "gpfq_set_main_kernel_events")


def exact_fallbacks(result):
    """How many decisions of an on-chip quantize_neurons() call were re-derived with the exact dot
    product (forces a device sync; diagnostics only)."""
    ws = result.get("workspace")
    return 0 if ws is None or ws.numel() < 8 else int(ws[:8].view(torch.int64).item())


def cluster_timeouts(result):
    """Nonzero when an exchange of the block kernel's cluster form (rows beyond 5120 samples: several workgroups per group of
    neurons) gave up waiting for a slice that never arrived -- the results of that call are then invalid (forces a device
    sync; diagnostics and tests)."""
    ws = result.get("workspace")
    return 0 if ws is None or ws.numel() < 16 else int(ws[8:12].view(torch.int32).item())


_OPTION_DEFAULTS = {"blk_cluster768": -1, "blk_prep_run": 1, "blk_prep_norms": 1, "blk_cluster": 1, "blk_cluster_nl": 0, "blk_cluster_map": -1, "blk_chip_ok": -1, "blk_cluster_timeout_ms": 3000,
                    "blk_cluster_fault": 0, "sync_errors": 0}
_options = {}


def set_option(key, value):
    _check(load().gpfq_set_option(key.encode(), int(value)), f"gpfq_set_option({key})")
    _options[key] = int(value)


class option:
    """`with hip.option(key, value):` -- a process-wide library option for the duration of a block, then back to what this binding last
    set it to (or the library's default).  Results never depend on options, only which kernel runs."""

    def __init__(self, key, value):
        self.key, self.value = key, int(value)

    def __enter__(self):
        self.prev = _options.get(self.key, _OPTION_DEFAULTS.get(self.key))
        if self.prev is None:
            raise GpfqError(f"hip.option: no known default for '{self.key}'")
        set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.prev)
        return False


def msq_round(W, alphabet):
    """Nearest alphabet member of every weight (first index on ties) -> (Q f32, idx i8 / i16), W's shape."""
    _dev(W, torch.float32, "W")
    Wc = W.contiguous()
    arr, M, _ = _alphabet(alphabet)
    Q = torch.empty_like(Wc)
    idx = torch.empty(Wc.shape, dtype=index_dtype(M), device=W.device)
    with torch.cuda.device(W.device):
        _check(load().gpfq_msq_round(Wc.data_ptr(), Wc.numel(), arr, M, Q.data_ptr(), idx.data_ptr(), _stream()),
               "gpfq_msq_round")
    return Q, idx


def index_bits(M):
    """Bits per index on the wire for an alphabet of M members (8 = plain int8, 16 = plain int16: no packing)."""
    return int(load().gpfq_index_bits(int(M)))


def pack_indices(qidx, M):
    """[C][N] indices -> (packed u8 [C][ceil(N*bits/8)], bits); bits == 8 / 16 returns qidx (int8 / int16) itself."""
    _dev(qidx, index_dtype(M), "qidx")
    if qidx.dim() != 2 or not qidx.is_contiguous():
        raise GpfqError("qidx must be a contiguous [C][N] tensor")
    bits = index_bits(M)
    if bits >= 8:
        return qidx, bits
    C, N = qidx.shape
    packed = torch.empty((C, (N * bits + 7) // 8), dtype=torch.uint8, device=qidx.device)
    with torch.cuda.device(qidx.device):
        _check(load().gpfq_pack_indices(qidx.data_ptr(), N, C, bits, packed.data_ptr(), _stream()), "gpfq_pack_indices")
    return packed, bits


def assemble_kernel(qidx, alphabet, want_idx=True, bits=None, N=None):
    """[C][N] int8 / int16 indices (bits = 8 / 16; None = by the alphabet) or rows packed by pack_indices (bits = 2/4,
    pass N) -> (Q f32 [N][C] in Keras layout, idx i8 / i16 [N][C])."""
    if bits is None:
        bits = 8 if len(alphabet) <= 64 else 16
    _dev(qidx, torch.int16 if bits == 16 else torch.int8 if bits == 8 else torch.uint8, "qidx")
    if qidx.dim() != 2 or not qidx.is_contiguous():
        raise GpfqError("qidx must be a contiguous 2-D tensor")
    C = qidx.shape[0]
    if bits >= 8:
        N = qidx.shape[1]
    elif N is None or qidx.shape[1] != (N * bits + 7) // 8:
        raise GpfqError("packed indices need N, with ceil(N*bits/8) bytes per row")
    arr, M, _ = _alphabet(alphabet)
    Q = torch.empty((N, C), dtype=torch.float32, device=qidx.device)
    idx_t = torch.empty((N, C), dtype=index_dtype(M), device=qidx.device) if want_idx else None
    with torch.cuda.device(qidx.device):
        _check(load().gpfq_assemble_kernel(qidx.data_ptr(), bits, arr, M, N, C, Q.data_ptr(),
                                           idx_t.data_ptr() if idx_t is not None else None, _stream()),
               "gpfq_assemble_kernel")
    return Q, idx_t


_pinned = {}


def _read_scalar(out, meanwhile=None):
    """The float32 device scalar `out` on the host.  `meanwhile` (a callable, optional) is run after the copy has been queued
    and before the host waits for it: kernel launches that do not depend on the value (the layer's transposes and row
    norms) keep the GPU busy while the host wakes up, instead of a launch bubble after every median."""
    import numpy as np
    key = (out.device.index, torch.cuda.current_stream(out.device).cuda_stream)
    slot = _pinned.get(key)
    if slot is None:
        slot = _pinned[key] = (torch.empty(1, dtype=torch.float32).pin_memory(), torch.cuda.Event())
    host, ev = slot
    host.copy_(out, non_blocking=True)
    ev.record()
    if meanwhile is not None:
        meanwhile()
    ev.synchronize()
    return np.float32(host[0].item())


def median_abs(W, meanwhile=None, on_device=False):
    """np.median(np.abs(W.flatten())) of a float32 GPU tensor as a numpy float32 (exact select).
    on_device=True: the float32 device scalar [1] instead, no host wait (the class surface queues the medians of ALL its layers
    up front -- they depend on the analog kernels alone -- and reads them back with one copy: medians_to_host)."""
    _dev(W, torch.float32, "W")
    Wc = W.contiguous()
    lib = load()
    nbytes = lib.gpfq_median_abs_workspace_bytes_for(Wc.numel())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=W.device)
    out = torch.empty(1, dtype=torch.float32, device=W.device)
    with torch.cuda.device(W.device):
        _check(lib.gpfq_median_abs(Wc.data_ptr(), Wc.numel(), out.data_ptr(), ws.data_ptr(), nbytes, _stream()),
               "gpfq_median_abs")
        if on_device:
            return out
        return _read_scalar(out, meanwhile)


def medians_to_host(scalars):
    """float32 device scalars (median_abs(..., on_device=True)) -> list of numpy float32, one copy and one host wait for all."""
    import numpy as np
    if not scalars:
        return []
    host = torch.cat([t.reshape(1) for t in scalars]).cpu().numpy()
    return [np.float32(v) for v in host]


GPFQ_MEDIAN_HIST_OFFSET, GPFQ_MEDIAN_HIST_WORDS = 64, 4096


def median_abs_sharded(W_local, n_total, all_reduce_sum, meanwhile=None, on_device=False):
    """The same median when every rank counts one slice of the layer's n_total weights (W_local: this rank's
    contiguous float32 slice; slices partition the flattened kernel).  `all_reduce_sum(t)` must sum the int32
    tensor t in place over the ranks (dist.all_reduce): three 16 KiB all-reduces per median."""
    import numpy as np
    _dev(W_local, torch.float32, "W_local")
    Wc = W_local.contiguous()
    lib = load()
    nbytes = lib.gpfq_median_abs_workspace_bytes()
    ws = torch.empty(nbytes, dtype=torch.uint8, device=W_local.device)
    hist = ws[GPFQ_MEDIAN_HIST_OFFSET:GPFQ_MEDIAN_HIST_OFFSET + 4 * GPFQ_MEDIAN_HIST_WORDS].view(torch.int32)
    out = torch.empty(1, dtype=torch.float32, device=W_local.device)
    with torch.cuda.device(W_local.device):
        _check(lib.gpfq_median_abs_begin(n_total, ws.data_ptr(), nbytes, _stream()), "gpfq_median_abs_begin")
        for p in range(3):
            _check(lib.gpfq_median_abs_count(Wc.data_ptr(), Wc.numel(), n_total, p, ws.data_ptr(), _stream()), "gpfq_median_abs_count")
            all_reduce_sum(hist)
            _check(lib.gpfq_median_abs_pick(n_total, p, ws.data_ptr(), _stream()), "gpfq_median_abs_pick")
        _check(lib.gpfq_median_abs_end(n_total, ws.data_ptr(), out.data_ptr(), _stream()), "gpfq_median_abs_end")
        if on_device:
            return out
        return _read_scalar(out, meanwhile)


def patch_out_dim(size, k, stride, rate, same):
    return int(load().gpfq_patch_out_dim(size, k, stride, rate, 1 if same else 0))


def extract_patches(act, channel, kernel_size, strides, rate, padding, out=None):
    """Patch matrix [kh*kw][n*oh*ow] of one channel of NHWC activations (TF extract_patches order)."""
    _dev(act, torch.float32, "act")
    if act.dim() != 4 or not act.is_contiguous():
        raise GpfqError("act must be a contiguous NHWC tensor")
    n, H, W, Cin = act.shape
    kh, kw = kernel_size
    sh, sw = strides
    rh, rw = rate if rate else (1, 1)
    same = str(padding).upper() == "SAME"
    if not same and str(padding).upper() != "VALID":
        raise GpfqError(f"unknown padding {padding!r}")
    oh, ow = patch_out_dim(H, kh, sh, rh, same), patch_out_dim(W, kw, sw, rw, same)
    cols = n * oh * ow
    if out is None:
        out = torch.empty((kh * kw, cols), dtype=torch.float32, device=act.device)
    ptr, r, c, ldp = _rows(out, "out")
    if r != kh * kw or c != cols:
        raise GpfqError("out has the wrong shape")
    with torch.cuda.device(act.device):
        _check(load().gpfq_extract_patches(act.data_ptr(), n, H, W, Cin, channel, kh, kw, sh, sw, rh, rw,
                                           1 if same else 0, out.data_ptr(), ldp, _stream()), "gpfq_extract_patches")
    return out
