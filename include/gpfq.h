/*
 * gpfq.h -- C ABI of the MI355X (gfx950) greedy path-following quantizer (GPFQ) hot path.
 *
 * The reference (elybrand/quantized_neural_networks) is pure Python and has no FFI; the boundary
 * this library replaces is its process-pool task signature
 *
 *     executor.submit(_quantize_neuron_parallel, W[:, j], hf_filename, layer_alphabet)
 *                                               scripts/quantized_network.py:553-556
 *     executor.submit(_quantize_filter2D_parallel_jit, channel_filters[:,:,f], channel_idx,
 *                     channel_hf_filename, alphabet)
 *                                               scripts/quantized_network.py:707-712
 *
 * i.e. "one weight vector + the two activation matrices + the scaled alphabet -> one quantized
 * weight vector".  Here one call quantizes ALL neurons of a layer shard (or all filters of one
 * input channel) on the GPU.  INTEGRATION.md shows the ctypes binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain C, no C++/torch/HIP types in signatures; `stream` is a hipStream_t passed as void*
 *     (NULL = the default stream).  All launches are asynchronous on that stream.
 *   - every pointer marked [device] must be device-accessible memory owned by the caller; the
 *     library never allocates, frees or retains caller memory.  Scratch is caller-provided and
 *     sized by gpfq_workspace_bytes().
 *   - [host] pointers are read during the call only (copied into kernel arguments).
 *   - return value: GPFQ_OK (0) or a negative GPFQ_ERR_* code; gpfq_last_error() returns a
 *     thread-local human-readable message for the last failing call.  Nothing throws.
 *   - re-entrant per stream: calls on different streams with disjoint outputs/workspaces may
 *     overlap.
 *
 * Data layout (all row-major, "feature-major" as the reference's transposed HDF5 datasets,
 * scripts/quantized_network.py:467-470, :789-797):
 *   X, Xq   f32 [N][ld]   row t = direction t of the analog / quantized network's walk over the m
 *                         calibration samples (wX[t,:], qX[t,:]); ld >= m is the row pitch in elements
 *   Wt      f32 [C][ldw]  neuron-major weights: row j = W[:, j] (ldw >= N)
 *   qidx    i8  [C][N]    alphabet index chosen for weight t of neuron j; `zero_idx` semantics below.
 *                         Alphabets of more than 64 members (bits 7 and 8 of the reference's `bits`, :396)
 *                         have i16 elements instead: every `void *qidx` below is int8_t* for M <= 64 and
 *                         int16_t* for 64 < M <= GPFQ_MAX_ALPHABET (gpfq_index_bits(M) == 16).
 *   Qt      f32 [C][N]    the quantized weights, (float)alphabet[qidx] (Keras stores float32)
 *   resid   f64 [C]       ||u_final||_2 per neuron (the reference discards u, :121; emitted for parity checks)
 *   u_out   f64 [C][m]    optional final residual vectors (NULL to skip)
 */
#ifndef GPFQ_H
#define GPFQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPFQ_OK                 0
#define GPFQ_ERR_INVALID_ARG   (-1)   /* NULL pointer, negative size, bad pitch ...            */
#define GPFQ_ERR_UNSUPPORTED   (-2)   /* e.g. alphabet larger than GPFQ_MAX_ALPHABET           */
#define GPFQ_ERR_WORKSPACE     (-3)   /* workspace missing or smaller than gpfq_workspace_bytes */
#define GPFQ_ERR_LAUNCH        (-4)   /* HIP reported an error; text in gpfq_last_error()      */
#define GPFQ_ERR_NO_DEVICE     (-5)   /* no gfx950 device visible                              */
#define GPFQ_ERR_CLUSTER_TIMEOUT (-6) /* gpfq_call_status: an exchange of the block kernel's cluster form timed out, outputs invalid */
#define GPFQ_ERR_ALPHABET      (-7)   /* gpfq_call_status: the device-resident alphabet was degenerate (radius 0 / inf / NaN), nothing computed */

#define GPFQ_MAX_ALPHABET 256         /* alphabet members per call (bits <= 8, scripts/quantized_network.py:396: int(round(2**bits)));
                                         more than 64 members: int16 indices, the kernels that hold four alphabet registers per lane */

/* gpfq_quantize_neurons `path` selector */
#define GPFQ_PATH_AUTO      0   /* rows beyond GPFQ_GRAM_MIN_M samples with walks of <= GPFQ_GRAM_MAX_N steps (and no u_out):
                                   the Gram path + a rerun of what it flags (ONE stream synchronisation and a small D2H copy inside
                                   the call: not capturable into a hipGraph -- gpfq_set_option("auto_gram", 0) keeps AUTO on the
                                   asynchronous kernels, or call gpfq_quantize_neurons_gram and handle its `uncertified` flags
                                   yourself); else on-chip residual when m fits the registers, else streaming */
#define GPFQ_PATH_ONCHIP    1   /* residual u lives in VGPRs (of up to 16 wavefronts per neuron), rows staged through LDS (m <= GPFQ_ONCHIP_MAX_M) */
#define GPFQ_PATH_STREAM    2   /* residual u lives in HBM (any m; conv patch matrices)          */

#define GPFQ_ONCHIP_MAX_M 28672   /* 16 wavefronts x 64 lanes x 28 residual elements (float64) per lane */
#define GPFQ_GRAM_MIN_M   16384   /* conv layers / short walks take the Gram path above this many columns */

int         gpfq_version(void);
const char *gpfq_last_error(void);
/* number of visible HIP devices whose architecture is gfx950 (0 if none / no driver) */
int         gpfq_device_count(void);

/*
 * Pre-pass over the quantized-network activations: nrm32[t] = (float)sqrt(sum_i (double)Xq[t][i]^2).
 * Replaces the two scipy.linalg.norm(X_tilde, 2) calls made per weight at
 * scripts/quantized_network.py:83 and :89 (BLAS snrm2 -> float32-rounded norm); the value only
 * depends on t, so it is computed once per layer instead of 2*C times.
 *   Xq [device] f32 [N][ld], nrm32 [device] f32 [N].
 */
int gpfq_row_norms(const float *Xq, int64_t N, int64_t m, int64_t ld, float *nrm32, void *stream);

/* Bytes of scratch gpfq_quantize_neurons needs for this shape on this path.  The on-chip path
 * keeps per-row statistics there (32*N + 64 bytes, rounded up to 256; after the call the first 8 bytes
 * hold, as a uint64, how many decisions were re-derived with the exact dot product -- diagnostics
 * only) and the slot records of the block-pipelined kernel
 * (about (N + 9) * (384 + 16 * padded m) bytes for rows of up to 5120 samples: per step the row statistics, the Gram band and the
 * zero-padded operand rows, plus a compact second copy of the headers);
 * with less it still runs -- on the row-group kernels, or without any workspace in the reference's
 * verbatim flow (same results, slower). */
size_t gpfq_workspace_bytes(int64_t N, int64_t m, int64_t C, int path);

/* Name of the dense kernel family the calling thread's last gpfq_quantize_neurons dispatched ("" before
 * the first call); diagnostics for benchmarks and tests. */
const char *gpfq_last_dense_kernel(void);

/* Measurement hook: two hipEvent_t of the caller (NULL, NULL to clear; per calling thread; created with timing enabled) that take the
 * start and the end of a dense layer's MAIN kernel -- the recurrence, without the pre-passes that share the call -- so that a
 * benchmark times exactly the kernel its roofline statement names (bench.py): hipEventElapsedTime(start, stop) afterwards.  The
 * kernel is launched WITH the events (hipExtLaunchKernelGGL: the dispatch's own timestamps, the figure rocprofv3 reports; no
 * extra packet in the queue).  Only the block-pipelined kernel family (gpfq_blk_kernel) uses them; with any other family the
 * events are left as they were.  Results never depend on it. */
int gpfq_set_main_kernel_events(void *start_event, void *stop_event);

/*
 * Process-wide tuning/test hooks; results never depend on them, only which kernel family runs.  Each option is one atomic
 * integer: a call running on another thread sees the old or the new value of each, never a torn one -- the calls stay
 * re-entrant per stream whatever is set while they run.
 *   "onchip_mode"  1 (default) certified-prediction mode, 0 verbatim reference flow
 *   "tile_steps"   LDS tile height in steps (power of two <= 64), 0 = heuristic
 *   "group_waves"  neurons (wavefronts) per workgroup 1..16 of the wave-per-neuron kernel, 0 = heuristic
 *   "lanes_per_neuron"  0 = heuristic, 16/32/64 = row-group kernel with that many lanes per neuron,
 *                  1 = wave-per-neuron kernel
 *   "variant"      bit 0: row-group kernel without the float64 copy of Xq in LDS; bit 1: wide kernel with
 *                  LDS-staged rows instead of register prefetch; bit 2: Gram records of walks longer than 64
 *                  steps on the vector units instead of the matrix cores (v_mfma_f64_16x16x4_f64);
 *                  bit 4: pipelined kernel issues its LDS-DMA spread over the steps of a tile
 *   "pipe"         role-split dense kernels (alphabets <= 64): -1 (default) the block form (gpfq_blk.hip; rows of 257..28672
 *                  samples) where measured faster -- layers of 512+ neurons, and any width for rows of 769+ samples --,
 *                  0 never, 1 one step per slot (gpfq_pipe.hip, rows up to 2048) whenever it applies, 2 the block form
 *                  whenever it applies
 *   "blk_sweep_waves"   0 (default: by shape -- eleven for rows of 769..1024 samples, eight for shorter rows), 8 or 11: sweep
 *                  wavefronts per workgroup of the block form's 16-neuron four-step shapes (with the decision wavefront two or
 *                  three wavefronts per SIMD; same bits: DESIGN.md)
 *   "blk_quad_waves"    0 (default: by shape), 7 or 8: sweep wavefronts per workgroup of the four-group narrow shapes on rows of at most
 *                  768 samples (seven = two wavefronts per SIMD with the decision wavefront; the default for layers of at most 1024 neurons)
 *   "blk_quad_groups"   2 (default): layers of at most 2048 neurons on rows of 257..1024 samples take four neuron groups per sweep
 *                  wavefront with one or two neurons per lane (4 / 8 neurons per workgroup, dot products on the matrix unit);
 *                  1: only layers of 129..2048 neurons; 0: the one- / two-group shapes of round 3
 *   "blk_cluster"       the block form's CLUSTER FORM (a row cut into 1024-sample slices, one workgroup each, partial dot products exchanged
 *                  once per slot; rows of up to GPFQ_ONCHIP_MAX_M samples): 1 (default) by shape -- every row beyond 3072 samples, rows of 1537..3072
 *                  samples in layers that are one round of the chip --, 0 off (rows beyond 5120 samples then take the wide kernel), a row
 *                  length >= 1024: every row beyond it (tests)
 *   "blk_cluster_nl"    that form's neurons per workgroup: 0 (default) 8 where the layer is then one round, else 16; 1 / 2 / 4 force 4 / 8 / 16
 *   "blk_cluster_map"   its workgroup id -> (cluster, slice) map: 0 a cluster's slices side by side in one XCD's queue, 1 consecutive ids
 *                  (the slices go round the XCDs), -1 (default) = 1: the chip holds 256 / slices whole clusters per round under map 1 and
 *                  8 x (32 / slices) under map 0, never more (and where the slice count divides 8 an XCD streams one slice's records
 *                  instead of all of them); map 0 only when forced (tests, A/B); same bits
 *   "blk_four_groups"   1 (default): layers of at most 1024 neurons on rows of 769..2048 samples take 4 neurons per workgroup; 0: 8
 *   "blk_wide_groups"   1 (default): rows of 1025..2048 samples in layers of more than 2048 neurons take 16 neurons per workgroup
 *                  (eleven sweep wavefronts, one round of workgroups); 0: 8 neurons per workgroup as narrower layers do
 *   "waves_per_neuron"  2..16: force the wide kernel (one neuron over that many wavefronts), 0 = heuristic
 *                  (rows longer than 2048 samples, and layers too narrow to fill the chip)
 *   "gram_slack_log2"   Gram paths: error bounds multiplied by 2^value (tests force the repair/rerun branches)
 *   "conv_s2"      1 (default): 7x7 / stride 2 / VALID conv layers form their Gram records from shift sums of the parity classes of the
 *                  channel planes (gpfq_gram_s2.hip); 0: the matrix-core kernel
 *   "auto_gram"    1 (default): GPFQ_PATH_AUTO may divert long rows to the Gram path (one stream synchronisation inside the call);
 *                  0: AUTO only picks between the asynchronous on-chip and streaming kernels
 *   "conv_fused"   1 (default): conv layers read their patch rows from the channel planes; 0: per-channel patch matrices
 *   "conv_nhwc"    1 (default): 3x3 / stride 1 / SAME layers read the NHWC activations directly (shards of 32+ channels, and of 8 to 31 where 2, 4 or 8 divides the image count: see "conv_nhwc_halves")
 *                  (gpfq_quantize_conv3x3_nhwc); 0: channel planes first
 *   "conv_nhwc_slots"   that form's workgroups per launch (default 8192: many short one-wavefront workgroups balance themselves)
 *   "conv_nhwc_halves"  1 (default): in shards of at most 32 / 16 / 8 channels the lanes the shard leaves idle walk further
 *                  halves / quarters / eighths of the images (where that count divides the number of images); 0: they idle
 *   "conv_planes_free"  1 (default): 7x7 / stride 2 / VALID layers read the NHWC activations themselves
 *                  (gpfq_quantize_conv_channels_nhwc); 0: channel planes first
 *   "blk_pair_groups", "blk_single_groups"  1 (default): the block form takes two neurons per workgroup in layers of at most 512
 *                  neurons, one in layers of at most 128; 0: four / two
 *   "conv_strip"   plane-correlation kernel: output positions per lane (0 = heuristic, 1, 2 or 4)
 *   "conv_shift"   1 (default): 3x3 / stride 1 / SAME layers on images of 20 x 20 or more (shards of 8+ channels) accumulate shift sums (27 FMAs per
 *                  position, border classes apart); 0: the per-output-position records (99 FMAs) that VALID layers and small
 *                  images use; 2: the shift form for every image of 4 x 4 or more (tests)
 */
int gpfq_set_option(const char *key, int value);
/* Round 6 additions to the option list above:
 *   "blk_prep_run"  1 (default): the block kernel's record pre-pass takes runs of 4 .. 16 records per workgroup (each row read about four
 *                  times instead of eighteen) for walks of 2048+ steps, the run length by the number of records; 0: one record per
 *                  workgroup (the same records); 4 .. 16: runs of that many records whatever the walk's length (tests, A/B)
 *   "blk_prep_norms"  1 (default): gpfq_quantize_dense_layer / _prepare called without the caller's row norms form them INSIDE the record
 *                  pre-pass where that reproduces gpfq_row_norms' sums bit for bit (runs of records, rows of 769 .. 1024 samples,
 *                  m a multiple of four) -- one launch less; 0: always by the row-norm kernel in front of the pre-pass (tests, A/B)
 *   "blk_cluster768"  -1 (default): rows of 2049..3072 samples in layers wider than 2048 neurons run as FOUR 768-sample slices of the cluster
 *                  form (64 clusters per round: whole rounds), eleven sweep wavefronts for symmetric alphabets, eight otherwise; 8 / 11 force
 *                  the count; 0: the classic one-step shape of rounds 4-5
 *   "blk_chip_ok"   -1 (default): the cluster form asks the device whether it is the whole chip its workgroup maps assume (256 compute
 *                  units = 8 XCDs x 32, no compute-unit mask in the environment) and is not used otherwise; 0 / 1 force the answer (tests)
 *   "blk_cluster_timeout_ms"  how long an exchange of the cluster form waits for a slice that does not arrive (default 3000)
 *   "blk_cluster_fault"       tests: 1 = one slice of the first cluster never publishes -- every exchange of that cluster times out
 *   "sync_errors"   1: gpfq_quantize_neurons (block-pipelined kernel) and gpfq_quantize_dense_layer wait for their launches and return
 *                  gpfq_call_status() of the call; 0 (default): asynchronous, the caller checks gpfq_call_status itself
 */

/*
 * Deferred errors of an asynchronous dense call (gpfq_quantize_neurons on the on-chip path, gpfq_quantize_dense_layer): waits for
 * `stream` and reads the call's status words at the start of its workspace -- bytes 0..7 exact-fallback decisions (diagnostics),
 * bytes 8..11 != 0: GPFQ_ERR_CLUSTER_TIMEOUT, bytes 12..15 != 0: GPFQ_ERR_ALPHABET.  The reference logs the failing unit and re-raises
 * at once (scripts/quantized_network.py:563-565); the layer drivers of this package call this BEFORE a layer's result reaches
 * set_weights and rerun the layer through the classic kernels (quantized_neural_networks_amd/layer.py).
 */
int gpfq_call_status(const void *workspace, void *stream);

/*
 * The hot path: run the greedy recurrence for C independent neurons.
 * Replaces C calls of _quantize_neuron_parallel (scripts/quantized_network.py:91-121) -- or, with
 * N = kh*kw and C = number of filters, of _quantize_filter2D_parallel_jit (:185-233) -- including
 * _quantize_weight_parallel (:59-89) and _bit_round_parallel (:40-57) per weight:
 *
 *   u = 0 (f64[m]);  for t in 0..N-1:
 *     if nrm32[t] < 1e-16:            q = 0 (literal; index zero_idx)            (:83-84)
 *     elif |<Xq_t, u>| < 1e-10:       q = nearest(alphabet, (double)w_t)          (:86-87)
 *     else:                           q = nearest(alphabet, <Xq_t, u + w_t*X_t> / nrm32[t]^2)   (:89)
 *     u += (double)( f32(w_t*X_t) - f32((float)q * Xq_t) )                        (:119)
 *
 * with the reference's float32/float64 flow reproduced per element (DESIGN.md "numerics").
 * nearest() = first index of min |alphabet[k] - t| (np.argmin tie rule).
 *
 *   alphabet [host] f64 [M], 1 <= M <= GPFQ_MAX_ALPHABET  (the layer alphabet rad*linspace(-1,1,M), :545)
 *   zero_idx : index written to qidx for the literal 0 of rule (i); pass the index of an exact
 *              0.0 alphabet member, or -1 if there is none (even M).
 *   nrm32    [device] from gpfq_row_norms.
 *   qidx, Qt, resid [device] outputs; any of them may be NULL.  u_out [device] optional.
 *   workspace [device], workspace_bytes >= gpfq_workspace_bytes(N, m, C, path).
 */
int gpfq_quantize_neurons(const float *X, const float *Xq, int64_t ld, const float *nrm32,
                          const float *Wt, int64_t ldw,
                          const double *alphabet, int M, int zero_idx,
                          int64_t N, int64_t m, int64_t C,
                          void *qidx, float *Qt, double *resid, double *u_out,
                          void *workspace, size_t workspace_bytes, int path, void *stream);

/*
 * The Dense layer driver as ONE call with nothing crossing to the host (round 6).  Replaces the body of _quantize_layer_parallel
 * (scripts/quantized_network.py:523-574) between "the activations are there" and set_weights:
 *
 *   rad = alphabet_scalar * median(|W|); layer_alphabet = rad * alphabet          (:544-545)   gpfq_median_abs + gpfq_layer_alphabet_device
 *   Q[:, j] = _quantize_neuron_parallel(W[:, j], ..., layer_alphabet) for every j  (:549-562)   gpfq_quantize_dense_layer
 *
 * gpfq_layer_alphabet_device forms, ON THE DEVICE, rad = float64(alphabet_scalar) * float64(*median32) (the reference's legacy-NumPy
 * typing of :544), the members rad * unit_alphabet[k] (float64 products, :545) and what the block-pipelined kernel derives from them,
 * into `dev_alphabet` (GPFQ_DEVICE_ALPHABET_BYTES of device memory, 16-byte aligned; float64 rad at byte 0, the M float64 members from
 * byte 128) -- no launch of the layer waits for the radius to reach the host.
 *   median32       [device] f32 [1]   from gpfq_median_abs (or the sharded protocol)
 *   unit_alphabet  [host]   f64 [M]   linspace(-1, 1, M) (:396), 1 <= M <= 64
 *
 * gpfq_quantize_dense_layer quantizes neurons c_lo .. c_lo + C of the layer:
 *   W      [device] f32 [N][ldc]   the Keras kernel itself (row = input feature; no neuron-major copy)
 *   nrm32  [device] f32 [N] from gpfq_row_norms, or NULL: computed here
 *   qidx, Q [device] outputs (either may be NULL): GPFQ_LAYOUT_KERAS: the whole layer's [N][ldo] arrays in the layout set_weights takes
 *          (:562, :570), of which columns c_lo .. c_lo + C are written; GPFQ_LAYOUT_NEURON_MAJOR: this shard's [C][N] (what the all-gather
 *          of a multi-GPU run exchanges; gpfq_assemble_kernel_device lays the gathered indices out)
 *   resid  [device] f64 [C] residual norms (may be NULL)
 *   workspace [device] >= gpfq_dense_layer_workspace_bytes(N, m, C), 16-byte aligned; its first 16 bytes are the call's status words
 *          (gpfq_call_status)
 * Only the block-pipelined kernel reads a device-resident alphabet: gpfq_dense_layer_supported() says whether this shape has one (rows of
 * 257..GPFQ_ONCHIP_MAX_M samples, M <= 64, linspace-like unit alphabet); otherwise GPFQ_ERR_UNSUPPORTED, and the caller reads the radius
 * back and uses gpfq_quantize_neurons.  A radius of 0 (more than half of the kernel is zero), infinity or NaN is only seen on the device:
 * the kernel then writes nothing and raises the call's alphabet word -- gpfq_call_status returns GPFQ_ERR_ALPHABET.
 */
#define GPFQ_DEVICE_ALPHABET_BYTES 1024
#define GPFQ_LAYOUT_NEURON_MAJOR 0
#define GPFQ_LAYOUT_KERAS        1
int gpfq_layer_alphabet_device(const float *median32, double alphabet_scalar, const double *unit_alphabet, int M,
                               void *dev_alphabet, void *stream);
/* ... or both steps of :544-545 in one call: median(|W|) and, by the last workgroup of the median's second pass, the alphabet -- no launch
 * between the two.  W [device] f32 [n], 16-byte aligned; median_out [device] f32 [1] (may be NULL); workspace >=
 * gpfq_median_abs_workspace_bytes_for(n). */
int gpfq_layer_alphabet_from_kernel(const float *W, int64_t n, double alphabet_scalar, const double *unit_alphabet, int M,
                                    void *dev_alphabet, float *median_out, void *workspace, size_t workspace_bytes, void *stream);
int gpfq_dense_layer_supported(int64_t N, int64_t m, int64_t C, const double *unit_alphabet, int M);
/* ... and whether the kernel of that shape writes GPFQ_LAYOUT_KERAS outputs itself (the 16-neuron four-step shapes: layers wider than 2048
 * neurons on rows of up to 1024 samples -- the decision wavefront has the slack there); elsewhere gpfq_quantize_dense_layer takes
 * GPFQ_LAYOUT_NEURON_MAJOR only (GPFQ_ERR_UNSUPPORTED otherwise) and gpfq_assemble_kernel_device lays the result out in one pass */
int gpfq_dense_layer_keras_out_supported(int64_t N, int64_t m, int64_t C, const double *unit_alphabet, int M);
size_t gpfq_dense_layer_workspace_bytes(int64_t N, int64_t m, int64_t C);
int gpfq_quantize_dense_layer(const float *X, const float *Xq, int64_t ld, const float *nrm32,
                              const float *W, int64_t ldc, int64_t c_lo, int64_t C,
                              const void *dev_alphabet, const double *unit_alphabet, int M,
                              int64_t N, int64_t m,
                              int8_t *qidx, float *Q, int out_layout, int64_t ldo, double *resid,
                              void *workspace, size_t workspace_bytes, void *stream);

/*
 * gpfq_quantize_dense_layer in two halves, for callers that overlap them: the median of |W| and the alphabet depend on the kernel alone, the
 * row norms and the record pre-pass on the activations alone.  gpfq_dense_layer_prepare (status block, row norms when nrm32 is NULL, record
 * pre-pass) may run on one stream while gpfq_median_abs + gpfq_layer_alphabet_device run on another; gpfq_dense_layer_run (for symmetric
 * alphabets one in-place pass over the records, then the kernel) follows once both are done, on the same workspace.  Same results as the
 * one-call form, bit for bit (quantized_neural_networks_amd/layer.py: quantize_dense_layer).
 */
int gpfq_dense_layer_prepare(const float *X, const float *Xq, int64_t ld, const float *nrm32, const double *unit_alphabet, int M,
                             int64_t N, int64_t m, int64_t C, void *workspace, size_t workspace_bytes, void *stream);
int gpfq_dense_layer_run(const float *X, const float *Xq, int64_t ld,
                         const float *W, int64_t ldc, int64_t c_lo, int64_t C,
                         const void *dev_alphabet, const double *unit_alphabet, int M,
                         int64_t N, int64_t m,
                         int8_t *qidx, float *Q, int out_layout, int64_t ldo, double *resid,
                         void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same recurrence for SHORT walks over LONG rows (conv: N = kh*kw steps against patch matrices of
 * m = n_img*oh*ow columns, _quantize_filter2D_parallel_jit, scripts/quantized_network.py:185-233), done on
 * N x N Gram matrices of the rows instead of on the rows: the patch data are read once per call instead of
 * once per neuron and step.  Every decision is certified against a rigorous bound on the float32 roundings
 * the Gram formulation skips; chains that stop at an uncertifiable step (about 1 in 10^4) are repaired on the
 * device from the exact element-wise dot products of that step (lists of up to 1024 chains; two rounds,
 * twelve for walks longer than 64 steps -- dense layers whose rows are too long to stay on chip); whatever is
 * still open afterwards (practically nothing) is flagged non-zero in `uncertified` and MUST be rerun by the
 * caller through gpfq_quantize_neurons -- the outputs of flagged neurons are undefined, those of unflagged
 * neurons equal the exact flow's.
 *   N <= GPFQ_GRAM_MAX_N, m < 2^30.  Arguments as gpfq_quantize_neurons; uncertified [device] i32 [C].
 *   resid (optional) is produced by an exact element-wise replay of the residual with the chosen q.
 *   compute_norms != 0: nrm32 is an OUTPUT, filled with (float)sqrt(<Xq_t,Xq_t>) from the Gram diagonal (the
 *   row norms come for free here; pass the same array on to the rerun of flagged neurons); else an input.
 *   Option "gram_slack_log2" (gpfq_set_option) multiplies the error bounds by 2^value (tests).
 *   Walks longer than 64 steps (dense layers: the reference's MNIST run, 25000 samples per row) build their
 *   records on the matrix cores (v_mfma_f64_16x16x4_f64; needs ld % 4 == 0 and 16-byte aligned X, Xq, otherwise
 *   the vector-unit kernel runs) and walk them with one wavefront per neuron.
 */
#define GPFQ_GRAM_MAX_N 1024
size_t gpfq_gram_workspace_bytes(int64_t N, int64_t m, int64_t C);
int gpfq_quantize_neurons_gram(const float *X, const float *Xq, int64_t ld, float *nrm32, int compute_norms,
                               const float *Wt, int64_t ldw,
                               const double *alphabet, int M, int zero_idx,
                               int64_t N, int64_t m, int64_t C,
                               void *qidx, float *Qt, double *resid, int32_t *uncertified,
                               void *workspace, size_t workspace_bytes, void *stream);

/*
 * Memoryless scalar quantization of n weights: Q[i] = (float)alphabet[nearest((double)W[i])].
 * Replaces the per-weight Python loop `[_bit_round_parallel(w, layer_alphabet) for w in W.flatten()]`
 * of the drivers' MSQ baseline (scripts/quantize_pretrained_mlp.py:109, _cnn.py:114, _imagenet.py:219).
 *   W [device] f32 [n]; Q [device] f32 [n] (may be NULL); qidx [device] i8 [n] (i16 for M > 64; may be NULL).
 */
int gpfq_msq_round(const float *W, int64_t n, const double *alphabet, int M,
                   float *Q, void *qidx, void *stream);

/*
 * Assemble the quantized kernel in Keras layout from the neuron-major indices:
 *   Q[t][j] = (float)alphabet[qidx[j][t]]   (0.0f for the literal-zero index -1),  qidx_t[t][j] = qidx[j][t].
 * Replaces the `Q[:, neuron_idx] = future.result()` assembly loop (scripts/quantized_network.py:557-562);
 * with neurons sharded over GPUs only the indices need to be all-gathered -- packed to 2 or 4 bits per
 * weight by gpfq_pack_indices when the alphabet allows (code = index + 1, 0 = the literal zero; every
 * neuron row padded to whole bytes, ceil(N*bits/8) bytes per row, so shards stay row-aligned).
 *   gpfq_index_bits(M): 2 for M <= 3, 4 for M <= 15, 8 for M <= 64 (plain int8 indices, no packing), 16 beyond
 *   (plain int16 indices).
 *   qidx [device] i8 [C][N] (bits = 8), i16 [C][N] (bits = 16) or packed u8 [C][ceil(N*bits/8)];
 *   Q [device] f32 [N][C] (may be NULL); qidx_t [device] [N][C] of the element type of `bits` (i8 for packed input; may be NULL).
 */
int gpfq_index_bits(int M);
int gpfq_pack_indices(const int8_t *qidx, int64_t N, int64_t C, int bits, uint8_t *packed, void *stream);
int gpfq_assemble_kernel(const void *qidx, int bits, const double *alphabet, int M, int64_t N, int64_t C,
                         float *Q, void *qidx_t, void *stream);
/* ... with the members read from a device-resident alphabet (gpfq_layer_alphabet_device; M <= 64: bits = 2, 4 or 8) */
int gpfq_assemble_kernel_device(const void *qidx, int bits, const void *dev_alphabet, int M, int64_t N, int64_t C,
                                float *Q, void *qidx_t, void *stream);

/*
 * median(|W|) of n float32 weights with NumPy's semantics (float32 result; even n -> float32 mean of
 * the two middle values).  Replaces `median(abs(W.flatten()))` of the alphabet radius
 * (scripts/quantized_network.py:544, :831).  Exact radix select, no sort.
 *   W [device] f32 [n]; median_out [device] f32 [1]; workspace [device] >= gpfq_median_abs_workspace_bytes().
 */
size_t gpfq_median_abs_workspace_bytes(void);
/* ... or, with room for the candidate list that lets the third pass read a few per cent of the data instead of all of it (round 6; the
 * call uses whatever the workspace holds beyond the minimum): */
size_t gpfq_median_abs_workspace_bytes_for(int64_t n);
int gpfq_median_abs(const float *W, int64_t n, float *median_out, void *workspace, size_t workspace_bytes,
                    void *stream);

/*
 * The same median with the counting sharded over ranks (every rank holds the layer's n_total weights and counts
 * its own slice; multi-GPU runs would otherwise repeat the whole select on every GPU).  Protocol, identical on
 * all ranks:  begin;  for pass = 0, 1, 2: { count(slice, pass);  sum the GPFQ_MEDIAN_HIST_WORDS 32-bit counters
 * at workspace + GPFQ_MEDIAN_HIST_OFFSET over the ranks (one all-reduce);  pick(pass) };  end -> median_out.
 * Slices must partition the n_total elements (start them on multiples of 4 elements for the 16-byte loads).
 */
#define GPFQ_MEDIAN_HIST_OFFSET 64
#define GPFQ_MEDIAN_HIST_WORDS  4096
int gpfq_median_abs_begin(int64_t n_total, void *workspace, size_t workspace_bytes, void *stream);
int gpfq_median_abs_count(const float *W_local, int64_t n_local, int64_t n_total, int pass, void *workspace, void *stream);
int gpfq_median_abs_pick(int64_t n_total, int pass, void *workspace, void *stream);
int gpfq_median_abs_end(int64_t n_total, void *workspace, float *median_out, void *stream);

/*
 * Per-channel im2col: the patch matrices of ONE input channel for the analog and quantized
 * activations, transposed to feature-major [kh*kw][n*oh*ow].  Replaces _build_patch_array
 * (scripts/quantized_network.py:729-809) + _segment_data2D (:123-183), i.e.
 * tf.image.extract_patches(images[B,H,W,1], sizes=[1,kh,kw,1], strides=[1,sh,sw,1],
 * rates=[1,rh,rw,1], padding) reshaped to (B*oh*ow, kh*kw) and stored transposed.
 *   act  [device] f32 NHWC [n][H][W][Cin]; channel c is gathered.
 *   same_padding: 1 = TF "SAME" (pad_before = floor(total/2)), 0 = "VALID".
 *   P    [device] f32 [kh*kw][ldp], ldp >= n*oh*ow; column order (image, oy, ox) row-major.
 * oh/ow as TF defines them; query with gpfq_patch_out_dim.
 */
int64_t gpfq_patch_out_dim(int64_t in, int64_t k, int64_t stride, int64_t rate, int same_padding);
int gpfq_extract_patches(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c,
                         int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                         float *P, int64_t ldp, void *stream);

/*
 * Channel planes of NHWC activations: planes[c][p] = act[p][c_lo + c] for p < npos = n*H*W, c < nch -- the
 * `[..., channel_idx]` slices of scripts/quantized_network.py:769-770 for a shard of channels in one pass
 * (the input layout of gpfq_quantize_conv_channels).  act [device] f32 [npos][Cin]; planes [device] f32 [nch][npos].
 */
int gpfq_channel_planes(const float *act, int64_t npos, int64_t Cin, int64_t c_lo, int64_t nch, float *planes, void *stream);

/*
 * Squared channel norms of NHWC activations over the positions a (1, 1) kernel with strides (sh, sw) visits:
 * sumsq[c] = sum over img, y % sh == 0, x % sw == 0 of act[img][y][x][c]^2 in float64 -- the squared norm of the single row
 * of channel c's patch matrix for kernel_size (1, 1) (scripts/quantized_network.py:769-797, :83).  All a 1 x 1 conv layer
 * needs from its activations: its (channel, filter) walks have one step, u = 0, so the decision is nearest(alphabet, w)
 * unless that norm is below 1e-16 (:83-87).  act [device] f32 [n][H][W][Cin]; sumsq [device] f64 [Cin]; one pass.
 */
size_t gpfq_channel_sumsq_workspace_bytes(int64_t Cin);
int gpfq_channel_sumsq(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, double *sumsq,
                       void *workspace, size_t workspace_bytes, void *stream);

/*
 * The one bit per channel a 1 x 1 conv layer consumes of its activations: dead[c] = 1 iff the float32-rounded norm of
 * channel c over the positions a (1, 1) kernel with strides (sh, sw) visits is below 1e-16 (rule (i),
 * scripts/quantized_network.py:83-84) -- the comparison gpfq_channel_sumsq's output would be put to, without reading the
 * whole tensor: sums of squares only grow, so a channel whose sum over the first `prefix_positions` positions
 * (0 = about 8 MiB worth) already exceeds 4e-32 is live; only the channels still undecided after the prefix get their
 * full sum (a strided pass over those channels alone; skipped on the device when there are none).  No host round trip.
 * act [device] f32 [n][H][W][Cin]; dead [device] i32 [Cin].
 */
size_t gpfq_channel_dead_workspace_bytes(int64_t Cin);
int gpfq_channel_dead(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, int64_t prefix_positions,
                      int32_t *dead, void *workspace, size_t workspace_bytes, void *stream);

/*
 * A whole conv layer of kernel_size (1, 1) in one call: every (channel, filter) pair is a ONE-step walk -- u = 0, so rule (ii)
 * (scripts/quantized_network.py:86-87) returns nearest(alphabet, w) unless the channel is dead on the strided grid, where rule
 * (i) (:83-84) returns the literal 0 (index: the alphabet's zero member, -1 if it has none).  gpfq_channel_dead + the MSQ pass
 * with its mask, queued back to back: the values _quantize_conv2D_layer_parallel_jit reaches through its general path (:835-842
 * is dead code there) for such a layer.
 * act_q [device] f32 [n][H][W][Cin]; Wt [device] f32 [Cin][F] (the Keras kernel [1][1][Cin][F] as it lies); Q f32 / qidx i8 (i16
 * beyond 64 members) [Cin][F], either may be NULL; workspace gpfq_conv1x1_workspace_bytes(Cin) bytes, 16-byte aligned.
 */
size_t gpfq_conv1x1_workspace_bytes(int64_t Cin);
int gpfq_quantize_conv1x1(const float *act_q, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, const float *Wt, int64_t F,
                          const double *alphabet, int M, float *Q, void *qidx, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The channel loop of a conv layer in one call: for each of `nch` input channels build the two patch
 * matrices and run gpfq_quantize_neurons_gram for that channel's F filters -- the body of
 * `for channel_idx in range(num_channels)` in _quantize_conv2D_layer_parallel_jit
 * (scripts/quantized_network.py:844-860) with _quantize_channel_parallel_jit (:652-727) inlined, all
 * launches asynchronous on `stream`, patch buffers reused from channel to channel.
 *   act_w, act_q [device] f32 CHANNEL-MAJOR [nch][n][H][W] (analog / quantized layer inputs of these channels;
 *                the same pointer for a first layer);  Wt [device] f32 [nch][F][kh*kw], each filter flattened
 *                row-major (:215);  outputs qidx/Qt [nch][F][kh*kw], resid [nch][F] (may be NULL),
 *                uncertified i32 [nch][F] (see gpfq_quantize_neurons_gram: flagged pairs must be rerun).
 *   Needs kh*kw <= GPFQ_GRAM_MAX_N and n*oh*ow < 2^30.
 * When resid == NULL no patch matrix is materialised: row t = (ky, kx) of a patch matrix is the channel
 * plane sampled at (oy*sh + ky*rh - pad_top, ox*sw + kx*rw - pad_left), so the Gram records of all channels
 * are accumulated straight from the planes in one launch (a plane-correlation kernel for 3x3 / stride 1;
 * implicit im2col for every other shape: 16x16 blocks on the matrix cores for 6 <= kh*kw <= 64 -- strided 3x3,
 * 5x5, 7x7 --, register tiles on the vector units otherwise), followed by one batched decide
 * launch and the device-side repair of uncertified chains (option "conv_fused" = 0 switches back to the
 * per-channel patch matrices; results are identical).
 * The workspace size depends on whether resid is requested (want_resid = resid != NULL).
 */
size_t gpfq_conv_channels_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int sh, int sw,
                                          int rh, int rw, int same_padding, int64_t F, int want_resid);
int gpfq_quantize_conv_channels(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch,
                                int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                                const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                                void *qidx, float *Qt, double *resid, int32_t *uncertified,
                                void *workspace, size_t workspace_bytes, void *stream);

/*
 * The channel loop of a 3 x 3 / stride 1 / SAME conv layer straight from the NHWC activations Keras hands over
 * (`layer_data[..., channel_idx]`, scripts/quantized_network.py:769-770, without the channel-major copy): the shift-form Gram
 * records with the lanes along the channels, then the batched decide of gpfq_quantize_conv_channels.  Same results.
 *   act_w, act_q [device] f32 NHWC [n][H][W][Cin]; the call takes channels [c_lo, c_lo + nch) (a rank's shard);
 *   Wt [nch][F][9], outputs qidx / Qt [nch][F][9], uncertified [nch][F] as gpfq_quantize_conv_channels (no residual norms).
 *   gpfq_conv3x3_nhwc_supported: 1 if this form takes the shape (images of 4 x 4 or more, 32+ channels in the shard -- or 8 to 31 on an image count that 2, 4 or 8 divides so that image groups fill at least half of a wavefront's lanes,
 *   options "conv_fused" and "conv_nhwc" on), else 0: use gpfq_channel_planes + gpfq_quantize_conv_channels.
 */
int gpfq_conv3x3_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t nch);
size_t gpfq_conv3x3_nhwc_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch, int64_t F);
int gpfq_quantize_conv3x3_nhwc(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c_lo, int64_t nch,
                               const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                               void *qidx, float *Qt, int32_t *uncertified, void *workspace, size_t workspace_bytes, void *stream);

/*
 * gpfq_quantize_conv_channels for layers whose kernel reads the NHWC activations ITSELF -- no channel-major copy (gpfq_channel_planes)
 * beforehand.  gpfq_conv_channels_nhwc_supported says which: today the 7 x 7 / stride 2 / VALID shapes of the shift-sum form
 * (ResNet50's conv1 on its padded input), whose bands are gathered out of the interleaved rows by the LDS-DMA requests.
 * act_w / act_q [device] f32 [n][H][W][Cin], the shard is channels [c_lo, c_lo + nch); everything else as
 * gpfq_quantize_conv_channels (workspace: gpfq_conv_channels_workspace_bytes of the same shape, want_resid = 0; residual norms
 * are not formed).  Replaces the same loop of _quantize_conv2D_layer_parallel_jit (scripts/quantized_network.py:729-809).
 */
int gpfq_conv_channels_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int sh, int sw, int rh, int rw,
                                      int same_padding);
int gpfq_quantize_conv_channels_nhwc(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c_lo,
                                     int64_t nch, int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                                     const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                                     void *qidx, float *Qt, int32_t *uncertified, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same call in two halves, for layers with fewer input channels than GPUs (an image input has 3): the Gram
 * records are SUMS over the patch columns, so every GPU forms them over its share of the images
 * (gpfq_conv_channel_records on the planes of those images), the records are summed over the GPUs (an all-reduce of
 * nch * (K*K*2 + K) doubles, K = kh*kw; the flags by maximum) and every GPU finishes from the summed records
 * (gpfq_quantize_conv_channels_from_records on the planes of ALL images, which the repair of uncertified chains
 * reads).  Certified decisions do not depend on the order in which the records were summed, and the one number whose
 * last bit would -- the float32 row norm, a rounded square root of a sum of squares -- is not taken from the records
 * for nch <= 15 (images of up to 2^19 pixels): both forms recompute it from act_q of ALL images in a summation order
 * fixed by n, H, W and the kernel geometry, so the results are those of the one-call form bit for bit, whatever the
 * number of GPUs up to 16 and however the images were split.  (Beyond 15 channels a layer shards by channel; if it is
 * run in two halves nevertheless the norm is the rounded square root of the summed record entry, and a sum within one
 * part in 10^16 of a float32 rounding boundary can round the other way than in the one-call form.)
 *   gpfq_conv_records_supported: 1 when BOTH halves have a plane kernel for this shape (the halves see different
 *   image counts: ask for each before committing all GPUs to this form), else 0.
 *   records [device] f64 [nch][K*K*2 + K], negflags [device] i32 [nch]; workspace as for
 *   gpfq_quantize_conv_channels with want_resid = 0 (F = 0 for the records half).
 *   GPFQ_ERR_UNSUPPORTED for kernel shapes that need patch matrices in memory (kh*kw > 256 ...): shard by channel.
 */
int gpfq_conv_records_supported(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int sh, int sw, int rh, int rw,
                                int same_padding);
int gpfq_conv_channel_records(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch,
                              int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                              double *records, int32_t *negflags, void *workspace, size_t workspace_bytes, void *stream);
int gpfq_quantize_conv_channels_from_records(const double *records, const int32_t *negflags,
                                             const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch,
                                             int kh, int kw, int sh, int sw, int rh, int rw, int same_padding,
                                             const float *Wt, const double *alphabet, int M, int zero_idx, int64_t F,
                                             void *qidx, float *Qt, int32_t *uncertified,
                                             void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GPFQ_H */
