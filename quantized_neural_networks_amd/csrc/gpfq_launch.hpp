// Host-side launch interfaces between the C ABI (gpfq_capi.hip) and the kernel files.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "gpfq_device.hpp"

namespace gpfq {

// Per-row quantities of the certified mode (gpfq_onchip.hip), produced by launch_row_stats:
//   G      = <Xq_t, X_t>                         (f64)
//   rden   = 1 / (f32-rounded ||Xq_t||)^2        (0 for rows that take rule (i))
//   cbound = 2^-23 * sum_i |Xq_ti X_ti| * rden   bound on the f32 product roundings, per unit |w|
//   cabs   = 2^-149 * sum_i |Xq_ti| * rden       the same for products that round in the subnormal range
struct RowStats {
    double G, rden, cbound, cabs;
};

struct OnchipArgs {
    const float *X, *Xq;
    int64_t ld;
    const float *nrm32;
    const RowStats *stats = nullptr;
    unsigned long long *fallback_count = nullptr;
    int mode = 0;          // 0 = exact flow, 1 = certified (needs stats)
    int lpn = 0;           // lanes per neuron of the row-group kernel: 16/32/64, 1 = wave kernel, 0 = heuristic
    int wpn = 0;           // wavefronts per neuron of the wide kernel (forced split), 0 = heuristic
    const float *Wt;
    int64_t ldw;
    AlphabetArg A;
    const AlphabetBig *big = nullptr;   // alphabets of 65..256 members: A is unused and qidx points at int16 elements
    int64_t N, m, C;
    int8_t *qidx;
    float *Qt;
    double *resid;
    double *u_out;
    int ts_override = 0;   // tuning hooks (bench/tests); 0 = heuristic
    int nw_override = 0;
    int variant = 0;
};

struct StreamArgs {
    const float *X, *Xq;
    int64_t ld;
    const float *nrm32;
    const float *Wt;
    int64_t ldw;
    AlphabetArg A;
    const AlphabetBig *big = nullptr;   // alphabets of 65..256 members: A is unused and qidx points at int16 elements
    int64_t N, m, C;
    int8_t *qidx;
    float *Qt;
    double *resid;
    double *u_out;      // may be NULL: then the residual lives in the workspace
    void *workspace;
    size_t workspace_bytes;
};

// Pipelined dense kernel (gpfq_pipe.hip): rows of up to 2048 samples, alphabets of up to 64 members.
struct PipeArgs {
    const float *X, *Xq;
    int64_t ld;
    const float *nrm32;
    const float *Wt;
    int64_t ldw;
    AlphabetArg A;
    int64_t N, m, C;
    int8_t *qidx;
    float *Qt;
    double *resid;
    double *u_out;
    void *workspace;                       // pipe_workspace_bytes(N, m), 16-byte aligned
    unsigned long long *fallback_count = nullptr;
    int ts_override = 0;                   // tuning hook (bench/tests); 0 = heuristic
    int variant = 0;                       // tuning / timing experiments (PipeK::flags)
    // Block form only (round 6, gpfq_quantize_dense_layer): the weights and the outputs in the layer's own (Keras) layout, the alphabet in
    // device memory.  Weight of neuron j at step t: Wt[j ldw + t ldt]; output element (j, t) at j o_sj + t o_st (o_st == 1: neuron-major
    // [C][N] whatever o_sj says).  dev_alpha != NULL: a DevAlphabet formed on the device (launch_alphabet_device) -- A then holds the UNIT
    // alphabet linspace(-1, 1, M), from which the host takes M, zero_idx and the choice of the symmetric form.
    int64_t ldt = 1;
    int64_t o_sj = 0, o_st = 1;
    const DevAlphabet *dev_alpha = nullptr;
    int zero_counters = 0;                 // 1: launch_blk zeroes the call's 64-byte counter block (fallback_count) itself, with the alphabet's store
    int phase = 0;                         // 0: the whole call; 1: the alphabet-independent half (record pre-pass); 2: the rest (launch_blk)
    // nrm32 == NULL with nrm32_out set (gpfq_quantize_dense_layer without the caller's norms): launch_blk forms the row norms itself -- inside the
    // record pre-pass where that reproduces the row-norm kernel's sums bit for bit (runs of records, rows of 1024 padded samples, m % 4 == 0),
    // else by that kernel into nrm32_out -- and zeroes the call's counter block (fallback_count) with them
    float *nrm32_out = nullptr;
};
// Block form (gpfq_blk.hip): B steps per slot; same arguments.
// The layer alphabet formed on the device from the float32 median of |W| (device scalar): rad = alphabet_scalar * median, members
// rad * unit[k]; dev_alphabet: GPFQ_DEVICE_ALPHABET_BYTES of device memory (a DevAlphabet).
hipError_t launch_alphabet_device(const float *median32, double alphabet_scalar, const AlphabetArg &unit, void *dev_alphabet, hipStream_t stream);
bool blk_supported(const PipeArgs &a);
bool blk_keras_out_supported(int64_t m, int64_t C);   // PipeArgs::o_st != 1 is taken (the 16-neuron four-step shapes; elsewhere neuron-major + one assembly pass)
size_t blk_workspace_bytes(int64_t N, int64_t m, int64_t C);   // (C: the cluster form's exchange buffers are per 16 neurons)
hipError_t launch_blk(const PipeArgs &a, hipStream_t stream);
void blk_set_four_groups(int on);   // 4-neuron workgroups for layers of at most 1024 neurons (speed only)
void blk_set_single_groups(int on); // 1-neuron workgroups for layers of at most 128 neurons (speed only)
void blk_set_pair_groups(int on);   // 2-neuron workgroups for layers of at most 512 neurons (speed only)
void blk_set_wide_groups(int on);   // 16-neuron workgroups for rows beyond 1024 samples (speed only)
void blk_set_quad_groups(int on);   // four neuron groups x 1 / 2 neurons per lane for layers of at most 2048 neurons on rows of 257..1024 samples; 2 (default): every such layer, 1: 129..2048 neurons only, 0: off (speed only)
void blk_set_quad_waves(int nw);    // sweep wavefronts of the four-group narrow shapes on rows of at most 768 samples: 0 (default) = by shape (seven for layers of at most 1024 neurons, else eight), 7 or 8 force it (speed only)
void blk_set_cluster_nl(int v);     // cluster form: neurons per lane of a workgroup, 0 (default) = by width, 1 / 2 / 4 force it (speed only)
void blk_set_cluster_map(int v);    // cluster form: workgroup id -> (cluster, slice): -1 (default) by the slice count, 0 = a cluster inside one XCD, 1 = consecutive ids (speed only)
void blk_set_cluster(int v);        // cluster form (rows cut into 1024-sample slices over several workgroups, up to 16384 samples): 1 (default) = by shape, 0 = off, v >= 1024 = every row beyond v samples (speed only)
void blk_set_prep_norms(int v);     // 1 (default): the dense-layer call's row norms formed inside the record pre-pass where bit-identical; 0: always by the row-norm kernel
void blk_set_prep_run(int v);       // 1 (default): the record pre-pass in runs of eight records per workgroup; 0: one record per workgroup (same records)
void blk_set_cluster768(int v);     // rows of 2049..3072 samples in layers wider than 2048 neurons as four 768-sample slices: -1 (default) yes, 8 / 11 force the sweep wavefronts, 0 = the classic one-step shape (speed only)
void blk_set_chip_ok(int v);        // -1 (default): the cluster form asks the device whether it is the whole 8 x 32-CU chip; 0 / 1: forced (tests)
void blk_set_cluster_timeout_ms(int v);  // cluster form: how long an exchange waits for a missing slice (default 3000 ms)
void blk_set_cluster_fault(int v);  // tests: 1 = one slice never publishes (forces the timeout and the caller's fallback)
void blk_set_sweep_waves(int nw);   // sweep wavefronts of the 16-neuron four-step shapes: 0 (default) = by shape (eleven for rows of 769..1024 samples, eight below), 8 or 11 force it (speed only)
bool pipe_supported(const PipeArgs &a);
size_t pipe_workspace_bytes(int64_t N, int64_t m);
hipError_t launch_pipe(const PipeArgs &a, hipStream_t stream);

// Raises `kernel`'s dynamic-LDS limit on the current device to `bytes` unless an earlier launch already did (gpfq_capi.hip).
hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes);

// Which dense kernel family the last gpfq_quantize_neurons call of this thread dispatched (diagnostics: gpfq_last_dense_kernel).
void note_dense_kernel(const char *name);

// Measurement hook (gpfq_set_main_kernel_events): two events of the calling thread that take the start and the end of a dense layer's
// MAIN kernel -- the recurrence itself, without its pre-passes -- so that a benchmark can time exactly the kernel its roofline statement
// names.  The block-pipelined kernel (gpfq_blk.hip) is launched with them (hipExtLaunchKernelGGL: the dispatch's own timestamps, nothing
// extra in the queue -- two hipEventRecord calls around the launch were two barrier packets, ~6 us of idle queue each, inside every timed
// step); other families leave the events alone.  False: no events set.
bool main_kernel_events(hipEvent_t *start, hipEvent_t *stop);

hipError_t launch_onchip(const OnchipArgs &a, hipStream_t stream);
bool rows_supported(const OnchipArgs &a, int lpn);
hipError_t launch_rows(const OnchipArgs &a, int lpn, hipStream_t stream);
hipError_t launch_wide(const OnchipArgs &a, int W, hipStream_t stream);

size_t stream_workspace_bytes(int64_t N, int64_t m, int64_t C, bool need_u);
hipError_t launch_stream(const StreamArgs &a, hipStream_t stream);

struct GramArgs {
    const float *X, *Xq;
    int64_t ld;
    const float *nrm32;
    float *nrm32_out = nullptr;   // non-NULL: fill the row norms from the Gram diagonal first (nrm32 == nrm32_out)
    const float *Wt;
    int64_t ldw;
    AlphabetArg A;
    const AlphabetBig *big = nullptr;   // alphabets of 65..256 members: A is unused and qidx points at int16 elements
    int64_t N, m, C;
    int8_t *qidx;
    float *Qt;
    double *resid;          // may be NULL (skips the exact replay of the residual)
    int32_t *uncertified;   // [C]: 1 = decision chain not certified, rerun through the exact path
    void *workspace;
    double slack = 1.0;     // multiplies the error bounds (tests)
    int variant = 0;        // bit 2: records of long walks on the vector units instead of the matrix cores
};

size_t gram_workspace_bytes(int64_t N, int64_t m, int64_t C);
hipError_t launch_gram(const GramArgs &a, hipStream_t stream);

// Gram records of long walks (N > 64) on the matrix cores (gpfq_gram_mfma.hip); partial records as gpfq_gram.hip's.
bool gram_mfma_supported(const float *X, const float *Xq, int64_t ld, int64_t N);
int64_t gram_mfma_walkers(int64_t N, int64_t m);
hipError_t launch_gram_mfma(const float *X, const float *Xq, int64_t ld, int64_t N, int64_t m, double *part, int *negflag,
                            hipStream_t stream);

// Batched decide step (one launch for all channels of a conv shard): element strides per channel.
struct DecideBatch {
    int64_t nch = 1, gram_cs = 0, nrm_cs = 0, w_cs = 0, out_cs = 0, unc_cs = 0, hist_cs = 0;
};
// Gram records (gpfq_gram.hip): [N][N][2] (G1, G2; lower triangle) + [N] (squared norms of the X rows).
// part: [nch][nparts] records -> gram: [nch] records, nrm32 (may be NULL): [nch][N].
hipError_t launch_gram_reduce(const double *part, int64_t nparts, int N, double *gram, float *nrm32, int64_t nch,
                              hipStream_t stream);
// Where the patch rows of the exact repair pass come from: a patch matrix in memory, or the channel planes
// (row (ky, kx), column (b, oy, ox) is plane[b][oy*sh + ky*rh - pt][ox*sw + kx*rw - pl], zero outside the image).
struct FixSrc {
    const float *X, *Xq;
    int64_t ld, m;               // m = columns (n*oh*ow for planes)
    int planes;                  // 0: X/Xq are [N][ld] patch rows; 1: channel images, element (ch, img, y, x) at ch * plane + ((img * H + y) * W + x) * pix
    int64_t plane;               // channel stride: floats per channel plane ([nch][n][H][W]), or 1 for NHWC
    int64_t pix;                 // pixel stride: 1 for channel planes, Cin for NHWC
    int n, H, W, oh, ow;
    int kw, sh, sw, rh, rw, pt, pl;
};
size_t gram_fix_bytes();
// Row norms of the patch rows of layers with few channel images, in a summation order fixed by the layer's dimensions (so that
// image-sharded records + all-reduce and the one-call form decide from the same norms); overwrites nrm32[ch * nrm_cs + t].
// No-op unless canonical_norms_apply(); fix_ws as for launch_gram_decide (used before it).
constexpr int kCanonNormMaxChannels = 15;
bool canonical_norms_apply(const FixSrc &src, int64_t nch);
hipError_t launch_canonical_norms(const FixSrc &src, int N, int64_t nch, float *nrm32, int64_t nrm_cs, void *fix_ws, hipStream_t stream);
// decide pass + (src != NULL) two rounds of device-side repair of the chains it could not certify;
// fix_ws: gram_fix_bytes() of scratch.  Chains still flagged afterwards are the caller's to rerun.
hipError_t launch_gram_decide(const double *gram, const float *nrm32, const float *Wt, int64_t ldw, const AlphabetArg &A,
                              int N, int64_t C, double slack, int8_t *qidx, float *Qt, int32_t *uncertified,
                              float *q32_hist, const DecideBatch &bs, const FixSrc *src, void *fix_ws, const int *negflag,
                              hipStream_t stream, const AlphabetBig *big = nullptr);
// big != NULL: alphabets of 65..256 members -- A is unused, qidx points at int16 elements.
// negflag (may be NULL): one int per channel, zeroed by the caller before its Gram kernel runs and set by
// that kernel when it meets a negative element of X or Xq; 0 lets the decide step use the Gram entries
// themselves as the absolute inner products of its error bound (post-ReLU inputs) instead of Cauchy-Schwarz.

// Fused conv path for 3x3 / stride 1 / rate 1 kernels (gpfq_gram_image.hip): the Gram matrices of all
// channels straight from the channel planes, then one batched decide launch -- no patch matrices.
struct ImageGramArgs {
    const float *act_w, *act_q;   // channel-major planes [nch][n][H][W]
    int64_t n, H, W, nch;
    int pad;                      // 1 = SAME (one ring of zeros), 0 = VALID
    const float *Wt;              // [nch][F][9]
    AlphabetArg A;
    const AlphabetBig *big = nullptr;   // as GramArgs
    int64_t F;
    int8_t *qidx;                 // [nch][F][9]
    float *Qt;                    // [nch][F][9]
    int32_t *uncertified;         // [nch][F]
    void *workspace;
    double slack = 1.0;
    int variant = 0;              // tuning hook: forces the strip length (1, 2, 4)
    int64_t nhwc_cin = 0;         // launch_gram_image_nhwc: act_w / act_q are NHWC tensors (offset to the shard's first channel) of this many channels
    int shift_form = 1;           // SAME padding: shift sums (27 FMAs per position) instead of per-output-position records (99); 0 = never
    // phase 1: stop after the Gram records (written to `records` [nch][171] f64 and `negflags` [nch] i32);
    // phase 2: take the records from there instead of forming them (column-sharded multi-GPU runs sum them in between)
    int phase = 0;
    double *records = nullptr;
    int32_t *negflags = nullptr;
};
// Any other kernel shape / stride / rate (gpfq_gram_conv.hip): the register-tile Gram kernel with implicit
// im2col staging from the channel planes, all channels of the shard in one launch, then the batched decide.
struct ConvGramArgs {
    const float *act_w, *act_q;   // channel-major planes [nch][n][H][W]
    int64_t n, H, W, nch;
    int kh, kw, sh, sw, rh, rw, pt, pl;
    int64_t oh, ow;
    const float *Wt;              // [nch][F][kh*kw]
    AlphabetArg A;
    const AlphabetBig *big = nullptr;   // as GramArgs
    int64_t F;
    int8_t *qidx;                 // [nch][F][kh*kw]
    float *Qt;
    int32_t *uncertified;         // [nch][F]
    void *workspace;
    double slack = 1.0;
    int variant = 0;        // bit 2: vector-unit tiles instead of the matrix cores for 16 < kh*kw <= 64
    int phase = 0;          // as ImageGramArgs: 1 = records only, 2 = from records ([nch][K*K*2 + K] f64, [nch] i32)
    double *records = nullptr;
    int32_t *negflags = nullptr;
    double *s2_part = nullptr;   // gram_s2_workspace_bytes of scratch behind the workspace proper (7x7 / 2 layers; NULL: the matrix-core kernel)
    int64_t pix = 1;             // > 1: act_w / act_q are NHWC tensors of this many channels, offset to the shard's first channel (7x7 / 2 shift-sum form only)
};
bool gram_conv_supported(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int64_t oh, int64_t ow);
size_t gram_conv_workspace_bytes(int64_t K, int64_t nch, int64_t F, int64_t m);
hipError_t launch_gram_conv(const ConvGramArgs &a, hipStream_t stream);
// 7x7 / stride 2 / VALID (ResNet50's conv1): the Gram records from shift sums of the parity classes of the planes (gpfq_gram_s2.hip)
bool gram_s2_supported(int64_t n, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int rh, int rw, int pt, int pl, int64_t nch);
size_t gram_s2_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch);
hipError_t launch_gram_s2(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch, double *part,
                          double *gram, float *nrm32, int *negflag, hipStream_t stream, int64_t pix = 1);
void image_set_nhwc_halves(int on);  // NHWC 3x3 form, <= 32 channels: the idle half of a wavefront walks the second half of the images (speed only)
void image_set_nhwc_slots(int n);    // NHWC 3x3 form: workgroups per launch (speed only)
void conv_set_s2(int on);           // the shift-sum form for 7x7 / 2 layers (speed only; 0: the matrix-core kernel)

bool gram_image_supported(int64_t n, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int rh, int rw, int same_padding);
size_t gram_image_workspace_bytes(int64_t nch, int64_t F);
hipError_t launch_gram_image(const ImageGramArgs &a, hipStream_t stream);
// The same from NHWC activations (ImageGramArgs::nhwc_cin; 3 x 3 / stride 1 / SAME, phase 0 only): no channel-major copy
bool gram_image_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t nch);
size_t gram_image_nhwc_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch, int64_t F);
hipError_t launch_gram_image_nhwc(const ImageGramArgs &a, hipStream_t stream);

hipError_t launch_row_stats(const float *X, const float *Xq, int64_t N, int64_t m, int64_t ld, const float *nrm32,
                            RowStats *stats, hipStream_t stream);
hipError_t launch_row_norms(const float *Xq, int64_t N, int64_t m, int64_t ld, float *nrm32, hipStream_t stream, unsigned *zero16 = nullptr);
// zero16 != NULL: the launch also zeroes those sixteen 32-bit words (a call's counter block)
// big != NULL (65..256 members): A is unused, index arrays hold int16 elements (assemble: bits = 16)
// dead / row_len / zero_idx: weights [rows][row_len]; rows with dead[row] != 0 take Q = 0, index zero_idx (a 1 x 1 conv layer's dead channels)
hipError_t launch_msq(const float *W, int64_t n, const AlphabetArg &A, float *Q, int8_t *qidx, hipStream_t stream,
                      const AlphabetBig *big = nullptr, const int32_t *dead = nullptr, int64_t row_len = 1, int zero_idx = -1);
hipError_t launch_assemble(const int8_t *qidx, const AlphabetArg &A, int64_t N, int64_t C, int bits, float *Q, int8_t *idxT,
                           hipStream_t stream, const AlphabetBig *big = nullptr, const DevAlphabet *dev = nullptr);
// dev != NULL: the members are read from a DevAlphabet in device memory (A: its size only)
hipError_t launch_pack(const int8_t *qidx, int64_t N, int64_t C, int bits, uint8_t *packed, hipStream_t stream);
size_t median_workspace_bytes();
size_t median_workspace_bytes_fast(int64_t n);   // ... with room for the one-GPU form's candidate list (gpfq_median_abs_workspace_bytes_for)
size_t channel_sumsq_workspace_bytes(int64_t Cin);
hipError_t launch_channel_sumsq(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, double *out,
                                void *workspace, hipStream_t stream);
size_t channel_dead_workspace_bytes(int64_t Cin);
hipError_t launch_channel_dead(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, int32_t *dead,
                               void *workspace, int64_t prefix_positions, hipStream_t stream);
hipError_t launch_median_abs(const float *W, int64_t n, float *out, void *workspace, hipStream_t stream, size_t workspace_bytes,
                             void *alpha_out = nullptr, double alphabet_scalar = 0.0, const AlphabetArg *unit = nullptr, int want_sym = 0);
// alpha_out != NULL (two-pass form only): the last workgroup of the second pass also forms the layer's DevAlphabet from the median
// (unit: the unit alphabet, passed on by value; want_sym: the symmetric-form instantiations will be launched)
bool blk_unit_wants_sym(const AlphabetArg &unit);   // gpfq_blk.hip: whether the symmetric form applies to rad * unit
hipError_t launch_median_begin(int64_t n_total, void *workspace, hipStream_t stream);
hipError_t launch_median_count(const float *W_local, int64_t n_local, int64_t n_total, int pass, void *workspace, hipStream_t stream);
hipError_t launch_median_pick(int64_t n_total, int pass, void *workspace, hipStream_t stream);
hipError_t launch_median_end(int64_t n_total, void *workspace, float *out, hipStream_t stream);
hipError_t launch_channel_planes(const float *act, int64_t npos, int64_t Cin, int64_t c_lo, int64_t nch, float *planes,
                                 hipStream_t stream);
hipError_t launch_extract_patches(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c,
                                  int kh, int kw, int sh, int sw, int rh, int rw, int pad_top, int pad_left,
                                  int64_t oh, int64_t ow, float *P, int64_t ldp, hipStream_t stream);

}  // namespace gpfq
