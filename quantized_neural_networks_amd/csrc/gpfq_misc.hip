// Small kernels around the hot loop: the per-row norm pre-pass, memoryless scalar quantization,
// and the per-channel im2col that lays conv activations out as feature-major patch matrices.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

// ---- row norms -----------------------------------------------------------------------------
// nrm32[t] = (float)sqrt(sum_i (double)Xq[t][i]^2): scipy.linalg.norm(X_tilde, 2) on float32 data
// (scripts/quantized_network.py:83, :89) = BLAS snrm2, a float32-rounded norm.
// One workgroup per row; lanes stride the row with 16-B loads; fixed-order reduction.
__global__ void __launch_bounds__(256)
gpfq_row_norms_kernel(const float *__restrict__ Xq, int64_t m, int64_t ld, int vec, float *__restrict__ nrm32, unsigned *__restrict__ zero16)
{
    __shared__ double sm[4];
    if (zero16 && blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0u;     // (gpfq_quantize_dense_layer: the call's counter block, no memset of its own)
    const float *row = Xq + (int64_t)blockIdx.x * ld;
    double s = 0.0;
    if (vec) {
        const int64_t m4 = m / 4;
        const float4 *r4 = reinterpret_cast<const float4 *>(row);
        for (int64_t i = threadIdx.x; i < m4; i += 256) {
            const float4 v = r4[i];
            s = fma((double)v.x, (double)v.x, s);
            s = fma((double)v.y, (double)v.y, s);
            s = fma((double)v.z, (double)v.z, s);
            s = fma((double)v.w, (double)v.w, s);
        }
        for (int64_t i = m4 * 4 + threadIdx.x; i < m; i += 256) s = fma((double)row[i], (double)row[i], s);
    } else {
        for (int64_t i = threadIdx.x; i < m; i += 256) s = fma((double)row[i], (double)row[i], s);
    }
    s = wave_sum(s);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) nrm32[blockIdx.x] = (float)sqrt(sm[0] + sm[1] + sm[2] + sm[3]);
}

hipError_t launch_row_norms(const float *Xq, int64_t N, int64_t m, int64_t ld, float *nrm32, hipStream_t stream, unsigned *zero16)
{
    if (N == 0) return zero16 ? hipMemsetAsync(zero16, 0, 64, stream) : hipSuccess;
    const int vec = (ld % 4 == 0) && ((uintptr_t)Xq % 16 == 0);
    hipLaunchKernelGGL(gpfq_row_norms_kernel, dim3((unsigned)N), dim3(256), 0, stream, Xq, m, ld, vec, nrm32, zero16);
    return hipGetLastError();
}

// ---- channel norms straight from NHWC activations (1x1 conv layers) -----------------------------
// sumsq[c] = sum over the sampled positions (img, y, x) with y % sh == 0, x % sw == 0 of (double)act[img][y][x][c]^2: the
// squared row norm of channel c's 1 x 1 patch matrix (scripts/quantized_network.py:769-797 with kernel_size (1, 1): one row
// per channel, one column per output position) without building it.  A 1 x 1 layer needs nothing else from its activations
// (layer.py: _quantize_conv1x1); the channel-major copy + one workgroup per row that did this before read the tensor three
// times at 0.6 TB/s (64 rows of 12.8 M samples: 64 busy CUs).  Here the positions are split over the workgroups, a thread owns
// the same four (or one) channels at every position it visits, and the per-workgroup partial sums are added in a fixed order.
template <int VEC>
__global__ void __launch_bounds__(256)
gpfq_channel_sumsq_kernel(const float *__restrict__ act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw,
                          int64_t oh, int64_t ow, int64_t p_limit, double *__restrict__ partial)
{
    __shared__ double sm[256][VEC];
    const int64_t G = Cin / VEC;                                  // channel groups of a position
    const int64_t g0 = (int64_t)blockIdx.y * 256;                 // this workgroup's first group (G > 256: chunks of 256)
    const int gw = (int)(G - g0 < 256 ? G - g0 : 256);            // groups of this chunk
    const int pp = 256 / gw;                                      // positions in flight per pass
    const int g = threadIdx.x % gw, slot = threadIdx.x / gw;
    const int64_t P = n * oh * ow < p_limit ? n * oh * ow : p_limit;   // (p_limit: the first positions only, gpfq_channel_dead)
    const int64_t per = (P + gridDim.x - 1) / gridDim.x;
    const int64_t p_lo = (int64_t)blockIdx.x * per, p_hi = p_lo + per < P ? p_lo + per : P;
    double acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.0;
    if (slot < pp) {
        for (int64_t p = p_lo + slot; p < p_hi; p += pp) {
            int64_t pix = p;                                      // stride 1: the positions are the pixels, in order
            if (sh != 1 || sw != 1) {
                const int64_t img = p / (oh * ow);
                const unsigned rem = (unsigned)(p - img * oh * ow), y = rem / (unsigned)ow, x = rem - y * (unsigned)ow;
                pix = (img * H + (int64_t)y * sh) * W + (int64_t)x * sw;
            }
            const float *src = act + pix * Cin + (g0 + g) * VEC;
            if constexpr (VEC == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(src);
                acc[0] = fma((double)v.x, (double)v.x, acc[0]); acc[1] = fma((double)v.y, (double)v.y, acc[1]);
                acc[2] = fma((double)v.z, (double)v.z, acc[2]); acc[3] = fma((double)v.w, (double)v.w, acc[3]);
            } else {
                acc[0] = fma((double)src[0], (double)src[0], acc[0]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) sm[threadIdx.x][k] = acc[k];
    __syncthreads();
    if (threadIdx.x < gw) {                                       // slots of a group, in order
        double tot[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) tot[k] = 0.0;
        for (int q = 0; q < pp; ++q)
#pragma unroll
            for (int k = 0; k < VEC; ++k) tot[k] += sm[q * gw + threadIdx.x][k];
#pragma unroll
        for (int k = 0; k < VEC; ++k) partial[(int64_t)blockIdx.x * Cin + (g0 + threadIdx.x) * VEC + k] = tot[k];
    }
}

__global__ void __launch_bounds__(256)
gpfq_channel_sumsq_final_kernel(const double *__restrict__ partial, int64_t Cin, int nblocks, double *__restrict__ out)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Cin) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * Cin + c];
    out[c] = s;
}

constexpr int kSumsqBlocks = 1024;
size_t channel_sumsq_workspace_bytes(int64_t Cin) { return (size_t)kSumsqBlocks * (size_t)Cin * sizeof(double); }

hipError_t launch_channel_sumsq(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, double *out,
                                void *workspace, hipStream_t stream)
{
    const int64_t oh = (H + sh - 1) / sh, ow = (W + sw - 1) / sw, P = n * oh * ow;
    const bool vec = (Cin % 4 == 0) && ((uintptr_t)act % 16 == 0);
    const int64_t G = vec ? Cin / 4 : Cin;
    int64_t nb = (P + 63) / 64;                                   // at least 64 positions per workgroup
    const int64_t chunks = (G + 255) / 256;
    const int64_t cap = kSumsqBlocks / chunks > 0 ? kSumsqBlocks / chunks : 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    double *partial = static_cast<double *>(workspace);
    if (vec)
        hipLaunchKernelGGL(gpfq_channel_sumsq_kernel<4>, dim3((unsigned)nb, (unsigned)chunks), dim3(256), 0, stream, act, n, H, W, Cin, sh, sw,
                           oh, ow, P, partial);
    else
        hipLaunchKernelGGL(gpfq_channel_sumsq_kernel<1>, dim3((unsigned)nb, (unsigned)chunks), dim3(256), 0, stream, act, n, H, W, Cin, sh, sw,
                           oh, ow, P, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gpfq_channel_sumsq_final_kernel, dim3((unsigned)((Cin + 255) / 256)), dim3(256), 0, stream, partial, Cin, (int)nb, out);
    return hipGetLastError();
}

// ---- dead channels of a 1x1 conv layer without reading the whole tensor -----------------------------
// A 1 x 1 layer consumes ONE bit per channel of its activations: whether the float32-rounded norm of the channel's one-row
// patch matrix is below 1e-16 (rule (i), scripts/quantized_network.py:83-84).  Partial sums of squares only grow, so a channel
// whose sum over the FIRST positions already exceeds kLiveSumsq = 4e-32 (norm 2e-16: beyond 1e-16 by more than any rounding of
// the float64 sum, the square root or the float32 conversion) is live whatever follows.  Phase 1 sums a prefix of the
// positions (gpfq_channel_sumsq_kernel with a position limit: <= 8 MiB read), phase 2 lists the channels still undecided,
// phase 3 -- only when the list is not empty -- forms the full sums of the listed channels alone (positions over the
// workgroups, a strided 4-byte read per position and listed channel), phase 4 compares.  Same bits as comparing the full
// norms: only the comparison is consumed.  36 of ResNet50's 53 conv layers are 1 x 1: 33 ms of full passes -> launch latency.
constexpr double kLiveSumsq = 4e-32;
constexpr int kDeadBlocks = 512;         // workgroups of phase 3 (a partial sum per workgroup and listed channel)
constexpr int kDeadGroup = 8;            // listed channels accumulated together in one pass over the positions

struct DeadCtl { int n_undecided; int pad[15]; };

__global__ void __launch_bounds__(256)
gpfq_channel_dead_list_kernel(const double *__restrict__ partial, int64_t Cin, int nblocks, bool complete,
                              int32_t *__restrict__ dead, DeadCtl *__restrict__ ctl, int32_t *__restrict__ list)
{
    // one workgroup: the list keeps the channels in ascending order (fixed summation order downstream: deterministic)
    __shared__ int base;
    __shared__ int wcount[4];
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int64_t c0 = 0; c0 < Cin; c0 += 256) {
        const int64_t c = c0 + threadIdx.x;
        bool und = false;
        if (c < Cin) {
            double s = 0.0;
            for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * Cin + c];
            const bool live = s >= kLiveSumsq;
            // the prefix was the whole tensor: decide here with the norm pre-pass's rounding, float32(sqrt(sum)) (:83)
            if (complete) dead[c] = ((double)(float)sqrt(s) < 1e-16) ? 1 : 0;
            else { dead[c] = 0; und = !live; }
        }
        const unsigned long long bal = __ballot(und);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) wcount[wave] = __popcll(bal);
        __syncthreads();
        int off = base;
        for (int w = 0; w < wave; ++w) off += wcount[w];
        if (und) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = (int32_t)c;
        __syncthreads();
        if (threadIdx.x == 0) base += wcount[0] + wcount[1] + wcount[2] + wcount[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) ctl->n_undecided = base;
}

__global__ void __launch_bounds__(256)
gpfq_channel_dead_scan_kernel(const float *__restrict__ act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw,
                              int64_t oh, int64_t ow, const DeadCtl *__restrict__ ctl, const int32_t *__restrict__ list,
                              double *__restrict__ partial2)
{
    const int nu = ctl->n_undecided;
    if (nu == 0) return;                                          // (the usual case: every channel was live in the prefix)
    __shared__ double sm[4][kDeadGroup];
    const int64_t P = n * oh * ow;
    const int64_t per = (P + gridDim.x - 1) / gridDim.x;
    const int64_t p_lo = (int64_t)blockIdx.x * per, p_hi = p_lo + per < P ? p_lo + per : P;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int u0 = 0; u0 < nu; u0 += kDeadGroup) {
        int64_t ch[kDeadGroup];
        double acc[kDeadGroup];
#pragma unroll
        for (int k = 0; k < kDeadGroup; ++k) { ch[k] = u0 + k < nu ? list[u0 + k] : -1; acc[k] = 0.0; }
        for (int64_t p = p_lo + threadIdx.x; p < p_hi; p += 256) {
            int64_t pix = p;
            if (sh != 1 || sw != 1) {
                const int64_t img = p / (oh * ow);
                const unsigned rem = (unsigned)(p - img * oh * ow), y = rem / (unsigned)ow, x = rem - y * (unsigned)ow;
                pix = (img * H + (int64_t)y * sh) * W + (int64_t)x * sw;
            }
            const float *src = act + pix * Cin;
#pragma unroll
            for (int k = 0; k < kDeadGroup; ++k)
                if (ch[k] >= 0) { const double v = (double)src[ch[k]]; acc[k] = fma(v, v, acc[k]); }
        }
#pragma unroll
        for (int k = 0; k < kDeadGroup; ++k) {
            const double s = wave_sum(acc[k]);
            if (lane == 0) sm[wave][k] = s;
        }
        __syncthreads();
        if (threadIdx.x < kDeadGroup && u0 + (int)threadIdx.x < nu)
            partial2[(int64_t)blockIdx.x * Cin + u0 + threadIdx.x] = (sm[0][threadIdx.x] + sm[1][threadIdx.x]) + (sm[2][threadIdx.x] + sm[3][threadIdx.x]);
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
gpfq_channel_dead_final_kernel(const double *__restrict__ partial2, int64_t Cin, int nblocks, const DeadCtl *__restrict__ ctl,
                               const int32_t *__restrict__ list, int32_t *__restrict__ dead)
{
    const int nu = ctl->n_undecided;
    for (int u = blockIdx.x * 256 + threadIdx.x; u < nu; u += gridDim.x * 256) {
        double s = 0.0;
        for (int b = 0; b < nblocks; ++b) s += partial2[(int64_t)b * Cin + u];
        dead[list[u]] = ((double)(float)sqrt(s) < 1e-16) ? 1 : 0;
    }
}

static size_t dead_al(size_t x) { return (x + 255) & ~(size_t)255; }
size_t channel_dead_workspace_bytes(int64_t Cin)
{
    // [phase-1 partial sums][control][list][phase-3 partial sums]
    return dead_al(channel_sumsq_workspace_bytes(Cin)) + dead_al(sizeof(DeadCtl)) + dead_al((size_t)Cin * sizeof(int32_t)) +
           dead_al((size_t)kDeadBlocks * (size_t)Cin * sizeof(double));
}

hipError_t launch_channel_dead(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int sh, int sw, int32_t *dead,
                               void *workspace, int64_t prefix_positions, hipStream_t stream)
{
    const int64_t oh = (H + sh - 1) / sh, ow = (W + sw - 1) / sw, P = n * oh * ow;
    char *ws = static_cast<char *>(workspace);
    double *partial = reinterpret_cast<double *>(ws);   ws += dead_al(channel_sumsq_workspace_bytes(Cin));
    DeadCtl *ctl = reinterpret_cast<DeadCtl *>(ws);     ws += dead_al(sizeof(DeadCtl));
    int32_t *list = reinterpret_cast<int32_t *>(ws);    ws += dead_al((size_t)Cin * sizeof(int32_t));
    double *partial2 = reinterpret_cast<double *>(ws);
    // prefix: about 8 MiB of the tensor, at least 256 positions (0 = that default; tests pass short prefixes)
    int64_t P1 = prefix_positions > 0 ? prefix_positions : (int64_t)(1 << 21) / (Cin > 0 ? Cin : 1);
    if (prefix_positions <= 0 && P1 < 256) P1 = 256;
    if (P1 > P) P1 = P;
    const bool vec = (Cin % 4 == 0) && ((uintptr_t)act % 16 == 0);
    const int64_t G = vec ? Cin / 4 : Cin;
    int64_t nb = (P1 + 63) / 64;
    const int64_t chunks = (G + 255) / 256;
    const int64_t cap = kSumsqBlocks / chunks > 0 ? kSumsqBlocks / chunks : 1;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    if (vec)
        hipLaunchKernelGGL(gpfq_channel_sumsq_kernel<4>, dim3((unsigned)nb, (unsigned)chunks), dim3(256), 0, stream, act, n, H, W, Cin, sh, sw,
                           oh, ow, P1, partial);
    else
        hipLaunchKernelGGL(gpfq_channel_sumsq_kernel<1>, dim3((unsigned)nb, (unsigned)chunks), dim3(256), 0, stream, act, n, H, W, Cin, sh, sw,
                           oh, ow, P1, partial);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gpfq_channel_dead_list_kernel, dim3(1), dim3(256), 0, stream, partial, Cin, (int)nb, P1 == P, dead, ctl, list);
    if ((e = hipGetLastError()) != hipSuccess || P1 == P) return e;
    int64_t nb2 = (P + 255) / 256;
    if (nb2 > kDeadBlocks) nb2 = kDeadBlocks;
    hipLaunchKernelGGL(gpfq_channel_dead_scan_kernel, dim3((unsigned)nb2), dim3(256), 0, stream, act, n, H, W, Cin, sh, sw, oh, ow, ctl, list,
                       partial2);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    hipLaunchKernelGGL(gpfq_channel_dead_final_kernel, dim3(8), dim3(256), 0, stream, partial2, Cin, (int)nb2, ctl, list, dead);
    return hipGetLastError();
}

// ---- row statistics of the certified mode (see RowStats) -----------------------------------
__global__ void __launch_bounds__(256)
gpfq_row_stats_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t m, int64_t ld,
                      const float *__restrict__ nrm32, RowStats *__restrict__ stats)
{
    __shared__ double sm[3][4];
    const float *rx = X + (int64_t)blockIdx.x * ld, *rq = Xq + (int64_t)blockIdx.x * ld;
    double g = 0.0, a = 0.0, s1 = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 256) {
        const double xq = (double)rq[i], pr = xq * (double)rx[i];     // exact product of two f32
        g += pr; a += fabs(pr); s1 += fabs(xq);
    }
    g = wave_sum(g); a = wave_sum(a); s1 = wave_sum(s1);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { sm[0][wave] = g; sm[1][wave] = a; sm[2][wave] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        g  = sm[0][0] + sm[0][1] + sm[0][2] + sm[0][3];
        a  = sm[1][0] + sm[1][1] + sm[1][2] + sm[1][3];
        s1 = sm[2][0] + sm[2][1] + sm[2][2] + sm[2][3];
        const double nrm = (double)nrm32[blockIdx.x];
        RowStats st;
        st.G = g;
        st.rden = nrm < 1e-16 ? 0.0 : 1.0 / (nrm * nrm);
        st.cbound = 0x1p-23 * a * st.rden * (1.0 + 0x1p-20);
        st.cabs = 0x1p-149 * s1 * st.rden * (1.0 + 0x1p-20);
        stats[blockIdx.x] = st;
    }
}

hipError_t launch_row_stats(const float *X, const float *Xq, int64_t N, int64_t m, int64_t ld, const float *nrm32,
                            RowStats *stats, hipStream_t stream)
{
    if (N == 0) return hipSuccess;
    hipLaunchKernelGGL(gpfq_row_stats_kernel, dim3((unsigned)N), dim3(256), 0, stream, X, Xq, m, ld, nrm32, stats);
    return hipGetLastError();
}

// ---- MSQ -----------------------------------------------------------------------------------
// Q[i] = alphabet[argmin |alphabet - (double)W[i]|], first index on ties
// (_bit_round_parallel applied per weight, scripts/quantize_pretrained_mlp.py:109).
template <class Alph, class Idx>
__global__ void __launch_bounds__(256)
gpfq_msq_kernel(const float *__restrict__ W, int64_t n, Alph A, float *__restrict__ Q, Idx *__restrict__ qidx,
                const int32_t *__restrict__ dead, int64_t row_len, int zero_idx)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        // (the one-step walks of a 1 x 1 conv layer: weights [channel][filter], a dead channel takes the literal 0 of rule (i))
        if (dead && dead[i / row_len]) {
            if (Q) Q[i] = 0.f;
            if (qidx) qidx[i] = (Idx)zero_idx;
            continue;
        }
        const double t = (double)W[i];
        int best = 0;
        double dbest = fabs(A.a[0] - t);
        for (int k = 1; k < A.M; ++k) {
            const double d = fabs(A.a[k] - t);
            if (d < dbest) { dbest = d; best = k; }
        }
        if (Q) Q[i] = (float)A.a[best];
        if (qidx) qidx[i] = (Idx)best;
    }
}

hipError_t launch_msq(const float *W, int64_t n, const AlphabetArg &A, float *Q, int8_t *qidx, hipStream_t stream,
                      const AlphabetBig *big, const int32_t *dead, int64_t row_len, int zero_idx)
{
    if (n == 0) return hipSuccess;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (row_len < 1) row_len = 1;
    if (big)
        hipLaunchKernelGGL((gpfq_msq_kernel<AlphabetBig, int16_t>), dim3((unsigned)blocks), dim3(256), 0, stream, W, n, *big, Q,
                           reinterpret_cast<int16_t *>(qidx), dead, row_len, zero_idx);
    else
        hipLaunchKernelGGL((gpfq_msq_kernel<AlphabetArg, int8_t>), dim3((unsigned)blocks), dim3(256), 0, stream, W, n, A, Q, qidx,
                           dead, row_len, zero_idx);
    return hipGetLastError();
}

// ---- per-channel im2col ----------------------------------------------------------------------
// P[ky*kw + kx][(b*oh + oy)*ow + ox] = act[b][oy*sh + ky*rh - pad_top][ox*sw + kx*rw - pad_left][c]
// (zero outside the image): tf.image.extract_patches on one channel, reshaped to
// (B*oh*ow, kh*kw) and stored transposed (scripts/quantized_network.py:158-179, :789-797).
__global__ void __launch_bounds__(256)
gpfq_patches_kernel(const float *__restrict__ act, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c,
                    int kh, int kw, int sh, int sw, int rh, int rw, int pad_top, int pad_left,
                    int64_t oh, int64_t ow, float *__restrict__ P, int64_t ldp)
{
    const int64_t cols = n * oh * ow;
    const int r = blockIdx.y;                    // patch row = ky*kw + kx
    const int ky = r / kw, kx = r - ky * kw;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; col < cols; col += stride) {
        const int64_t ox = col % ow;
        const int64_t oy = (col / ow) % oh;
        const int64_t b  = col / (ow * oh);
        const int64_t iy = oy * sh + (int64_t)ky * rh - pad_top;
        const int64_t ix = ox * sw + (int64_t)kx * rw - pad_left;
        float v = 0.f;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = act[((b * H + iy) * W + ix) * Cin + c];
        P[(int64_t)r * ldp + col] = v;
    }
}

hipError_t launch_extract_patches(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c,
                                  int kh, int kw, int sh, int sw, int rh, int rw, int pad_top, int pad_left,
                                  int64_t oh, int64_t ow, float *P, int64_t ldp, hipStream_t stream)
{
    const int64_t cols = n * oh * ow;
    if (cols == 0 || kh * kw == 0) return hipSuccess;
    int64_t bx = (cols + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(gpfq_patches_kernel, dim3((unsigned)bx, (unsigned)(kh * kw)), dim3(256), 0, stream,
                       act, n, H, W, Cin, c, kh, kw, sh, sw, rh, rw, pad_top, pad_left, oh, ow, P, ldp);
    return hipGetLastError();
}

// ---- channel planes: NHWC activations -> channel-major [nch][n*H*W] ------------------------------
// planes[c][p] = act[p][c_lo + c]: the `[..., channel_idx]` slices of scripts/quantized_network.py:769-770
// for a whole shard of channels in one pass.  256 positions x 32 channels per workgroup through LDS:
// reads run along the channels of a position, writes along the positions of a channel.
__global__ void __launch_bounds__(256)
gpfq_planes_kernel(const float *__restrict__ act, int64_t npos, int64_t Cin, int64_t c_lo, int64_t nch,
                   float *__restrict__ planes)
{
    __shared__ float tile[256][33];
    const int64_t p0 = (int64_t)blockIdx.x * 256;
    const int cb = blockIdx.y * 32;
    const int tc = (int)(nch - cb < 32 ? nch - cb : 32);
    const int np = (int)(npos - p0 < 256 ? npos - p0 : 256);
    const float *src = act + p0 * Cin + c_lo + cb;
    if (tc == 32) {
        for (int i = threadIdx.x; i < np * 32; i += 256) tile[i >> 5][i & 31] = src[(int64_t)(i >> 5) * Cin + (i & 31)];
    } else {
        for (int i = threadIdx.x; i < np * tc; i += 256) {
            const int p = i / tc, c = i - p * tc;
            tile[p][c] = src[(int64_t)p * Cin + c];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = wave; c < tc; c += 4) {
        float *dst = planes + (int64_t)(cb + c) * npos + p0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int p = lane + 64 * k;
            if (p < np) dst[p] = tile[p][c];
        }
    }
}

// Few channels in all (image inputs: 1..4): the LDS tile would be nearly empty -- one thread per position instead,
// its CIN values read as one contiguous group, every plane written along the positions.
template <int CIN>
__global__ void __launch_bounds__(256)
gpfq_planes_few_kernel(const float *__restrict__ act, int64_t npos, float *__restrict__ planes)
{
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < npos; p += (int64_t)gridDim.x * 256) {
        float v[CIN];
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[c] = act[p * CIN + c];
#pragma unroll
        for (int c = 0; c < CIN; ++c) planes[(int64_t)c * npos + p] = v[c];
    }
}

hipError_t launch_channel_planes(const float *act, int64_t npos, int64_t Cin, int64_t c_lo, int64_t nch, float *planes,
                                 hipStream_t stream)
{
    if (npos == 0 || nch == 0) return hipSuccess;
    if (c_lo == 0 && nch == Cin && Cin <= 4) {
        int64_t blocks = (npos + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        const dim3 grid((unsigned)blocks);
        if (Cin == 1) hipLaunchKernelGGL(gpfq_planes_few_kernel<1>, grid, dim3(256), 0, stream, act, npos, planes);
        else if (Cin == 2) hipLaunchKernelGGL(gpfq_planes_few_kernel<2>, grid, dim3(256), 0, stream, act, npos, planes);
        else if (Cin == 3) hipLaunchKernelGGL(gpfq_planes_few_kernel<3>, grid, dim3(256), 0, stream, act, npos, planes);
        else hipLaunchKernelGGL(gpfq_planes_few_kernel<4>, grid, dim3(256), 0, stream, act, npos, planes);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gpfq_planes_kernel, dim3((unsigned)((npos + 255) / 256), (unsigned)((nch + 31) / 32)), dim3(256), 0, stream,
                       act, npos, Cin, c_lo, nch, planes);
    return hipGetLastError();
}

// ---- kernel assembly: [C][N] indices -> Keras-layout [N][C] values (+ indices) ------------------
// Q[t][j] = (float)alphabet[qidx[j][t]] (0 for the literal-zero index -1), i.e. `Q[:, neuron_idx] =
// future.result()` for every neuron (scripts/quantized_network.py:562) fused with the transpose from
// the kernels' neuron-major layout.  32x32 tiles through LDS, both sides coalesced.
// `bits` = 8: qidx holds one int8 index per weight (16: one int16, alphabets of 65..256 members); 2 or 4: rows
// packed by gpfq_pack_kernel (code = index + 1, 0 = the literal zero), row pitch NB = ceil(N*bits/8) bytes.
template <class Alph, class Idx>
__global__ void __launch_bounds__(256)
gpfq_assemble_kernel(const Idx *__restrict__ qidx, Alph A, int64_t N, int64_t C, int bits,
                     float *__restrict__ Q, Idx *__restrict__ idxT, int64_t jtile0)
{
    __shared__ Idx tile[32][33];
    const int64_t t0 = (int64_t)blockIdx.x * 32, j0 = ((int64_t)blockIdx.y + jtile0) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int64_t NB = (N * bits + 7) / 8;
    for (int r = ty; r < 32; r += 8) {
        const int64_t j = j0 + r, t = t0 + tx;
        Idx k = 0;
        if (j < C && t < N) {
            if (bits >= 8) k = qidx[j * N + t];
            else {
                const unsigned byte = reinterpret_cast<const unsigned char *>(qidx)[j * NB + (t * bits) / 8];
                k = (Idx)((int)((byte >> ((t * bits) & 7)) & ((1u << bits) - 1u)) - 1);
            }
        }
        tile[r][tx] = k;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t t = t0 + r, j = j0 + tx;
        if (t < N && j < C) {
            const int k = tile[tx][r];
            if (Q) Q[t * C + j] = (k >= 0 && k < A.M) ? (float)A.a[k] : 0.f;
            if (idxT) idxT[t * C + j] = (Idx)k;
        }
    }
}

// The common case -- one int8 index per weight, N and C multiples of four -- in 64 x 64 tiles with 4-byte index reads and
// 16-byte value writes (the 32 x 32 form reads single bytes: 43 -> 27 us for 4096 x 4096).
template <class Alph>
__global__ void __launch_bounds__(256)
gpfq_assemble64_kernel(const int8_t *__restrict__ qidx, Alph A, int64_t N, int64_t C, float *__restrict__ Q, int8_t *__restrict__ idxT,
                       int64_t jtile0)
{
    __shared__ int8_t tile[64][68];                               // [neuron][step], rows 4-byte aligned
    const int64_t t0 = (int64_t)blockIdx.x * 64, j0 = ((int64_t)blockIdx.y + jtile0) * 64;
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;       // 16 x 16
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int r = r0 + 16 * pass;
        const int64_t j = j0 + r, t = t0 + 4 * c4;
        unsigned w = 0;
        if (j < C && t < N) w = *reinterpret_cast<const unsigned *>(qidx + j * N + t);
        *reinterpret_cast<unsigned *>(&tile[r][4 * c4]) = w;
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int tt = r0 + 16 * pass;
        const int64_t t = t0 + tt, j = j0 + 4 * c4;
        if (t < N && j < C) {
            int k[4];
            float q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                k[e] = tile[4 * c4 + e][tt];
                q[e] = (k[e] >= 0 && k[e] < A.M) ? (float)A.a[k[e]] : 0.f;
            }
            if (Q) *reinterpret_cast<float4 *>(Q + t * C + j) = make_float4(q[0], q[1], q[2], q[3]);
            if (idxT) *reinterpret_cast<unsigned *>(idxT + t * C + j) = (unsigned)(k[0] & 0xff) | ((unsigned)(k[1] & 0xff) << 8) |
                                                                       ((unsigned)(k[2] & 0xff) << 16) | ((unsigned)(k[3] & 0xff) << 24);
        }
    }
}

hipError_t launch_assemble(const int8_t *qidx, const AlphabetArg &A, int64_t N, int64_t C, int bits, float *Q, int8_t *idxT,
                           hipStream_t stream, const AlphabetBig *big, const DevAlphabet *dev)
{
    if (N == 0 || C == 0) return hipSuccess;
    const AlphabetRef R{dev ? dev->a : nullptr, A.M};             // (dev: the members stay in device memory -- only the address is taken here)
    // the neuron tiles ride on grid.y (at most 65535): kernels of more rows -- a multi-GPU 1 x 1 conv layer has Cin * F of them,
    // 2^21 for ResNet50's conv5_block1_0_conv -- take several launches, each told its first tile
    constexpr int64_t kMaxY = 65535;
    if (!big && bits == 8 && N % 4 == 0 && C % 4 == 0 && (uintptr_t)qidx % 4 == 0 && (uintptr_t)Q % 16 == 0 && (uintptr_t)idxT % 4 == 0) {
        const int64_t tiles = (C + 63) / 64;
        for (int64_t j = 0; j < tiles; j += kMaxY) {
            const int64_t ny = tiles - j < kMaxY ? tiles - j : kMaxY;
            if (dev) hipLaunchKernelGGL(gpfq_assemble64_kernel<AlphabetRef>, dim3((unsigned)((N + 63) / 64), (unsigned)ny), dim3(256), 0, stream, qidx, R, N, C, Q, idxT, j);
            else hipLaunchKernelGGL(gpfq_assemble64_kernel<AlphabetArg>, dim3((unsigned)((N + 63) / 64), (unsigned)ny), dim3(256), 0, stream, qidx, A, N, C, Q, idxT, j);
        }
        return hipGetLastError();
    }
    const int64_t tiles = (C + 31) / 32;
    for (int64_t j = 0; j < tiles; j += kMaxY) {
        const int64_t ny = tiles - j < kMaxY ? tiles - j : kMaxY;
        const dim3 grid((unsigned)((N + 31) / 32), (unsigned)ny);
        if (big)
            hipLaunchKernelGGL((gpfq_assemble_kernel<AlphabetBig, int16_t>), grid, dim3(256), 0, stream,
                               reinterpret_cast<const int16_t *>(qidx), *big, N, C, bits, Q, reinterpret_cast<int16_t *>(idxT), j);
        else if (dev)
            hipLaunchKernelGGL((gpfq_assemble_kernel<AlphabetRef, int8_t>), grid, dim3(256), 0, stream, qidx, R, N, C, bits, Q, idxT, j);
        else
            hipLaunchKernelGGL((gpfq_assemble_kernel<AlphabetArg, int8_t>), grid, dim3(256), 0, stream, qidx, A, N, C, bits, Q, idxT, j);
    }
    return hipGetLastError();
}

// Pack the indices of each neuron row into `bits`-wide codes (index + 1; 0 = literal zero) so that the
// all-gather over xGMI moves 2 or 4 bits per weight instead of 8.  One thread per output byte.
__global__ void __launch_bounds__(256)
gpfq_pack_kernel(const int8_t *__restrict__ qidx, int64_t N, int64_t C, int bits, uint8_t *__restrict__ packed)
{
    const int64_t NB = (N * bits + 7) / 8;
    const int per = 8 / bits;
    const int64_t total = C * NB;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int64_t j = o / NB, b = o - j * NB;
        unsigned byte = 0;
        for (int e = 0; e < per; ++e) {
            const int64_t t = b * per + e;
            if (t < N) byte |= (unsigned)((int)qidx[j * N + t] + 1) << (e * bits);
        }
        packed[o] = (uint8_t)byte;
    }
}

hipError_t launch_pack(const int8_t *qidx, int64_t N, int64_t C, int bits, uint8_t *packed, hipStream_t stream)
{
    const int64_t total = C * ((N * bits + 7) / 8);
    if (total == 0) return hipSuccess;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gpfq_pack_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, qidx, N, C, bits, packed);
    return hipGetLastError();
}

// ---- median of |W| ---------------------------------------------------------------------------
// np.median(np.abs(W.flatten())) on float32 data (scripts/quantized_network.py:544, :831) without a
// sort: exact radix select on the bit patterns of |w| (monotone as unsigned integers), three
// histogram passes of 11 + 11 + 10 bits.  For an even count NumPy returns the float32 mean of the two
// middle values, so both are selected -- in the SAME three passes over the data: the two ranks are
// adjacent, so they nearly always share the class being refined (one histogram serves both); when
// they have parted, the pass counts two classes from the one read.
struct SelState {
    unsigned prefix;            // bits of the answer fixed so far
    unsigned pad;
    unsigned long long k;       // rank still to be resolved inside the prefix class
};

constexpr int kSelBins = 2048;
constexpr int kSelThreads = 1024;

// Few, large workgroups: every workgroup ends with one global atomic per non-empty bin, and in the first pass all of them
// hit the same ~40 bins (same-address atomics serialise in L2: with 1024 workgroups that tail was two thirds of the pass).
__global__ void __launch_bounds__(kSelThreads)
gpfq_select_hist_kernel(const float *__restrict__ W, int64_t n, const SelState *__restrict__ st, int nsel,
                        int shift, int nbits, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[2][kSelBins];
    const unsigned p0 = st[0].prefix, p1 = nsel > 1 ? st[1].prefix : p0;
    const bool split = p1 != p0;
    for (int b = threadIdx.x; b < kSelBins; b += kSelThreads) { h[0][b] = 0; h[1][b] = 0; }
    __syncthreads();
    const unsigned hi_mask = (shift + nbits >= 32) ? 0u : ~((1u << (shift + nbits)) - 1u);
    const unsigned bin_mask = (1u << nbits) - 1u;
    auto count = [&](float w) {
        const unsigned key = __float_as_uint(w) & 0x7fffffffu, cls = key & hi_mask, bin = (key >> shift) & bin_mask;
        if (cls == p0) atomicAdd(&h[0][bin], 1u);
        else if (split && cls == p1) atomicAdd(&h[1][bin], 1u);
    };
    const int64_t stride = (int64_t)gridDim.x * kSelThreads;
    const int64_t n4 = ((uintptr_t)W % 16 == 0) ? n / 4 : 0;
    const float4 *W4 = reinterpret_cast<const float4 *>(W);
    for (int64_t i = (int64_t)blockIdx.x * kSelThreads + threadIdx.x; i < n4; i += stride) {
        const float4 v = W4[i];
        count(v.x); count(v.y); count(v.z); count(v.w);
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kSelThreads + threadIdx.x; i < n; i += stride) count(W[i]);
    __syncthreads();
    // (while the two order statistics share their class, the second histogram stays empty and the pick reads the first)
    for (int b = threadIdx.x; b < kSelBins; b += kSelThreads) {
        const unsigned c0 = h[0][b], c1 = h[1][b];
        if (c0) atomicAdd(&hist[b], c0);
        if (c1) atomicAdd(&hist[kSelBins + b], c1);
    }
}

// Locate the bin holding rank k (parallel: 256 threads x 8 bins) for each order statistic in turn, fix its bits, clear the
// histograms.  `out` != NULL (last pass): the median itself -- for an even count the float32 mean of the two middle values.
__global__ void __launch_bounds__(256)
gpfq_select_pick_kernel(unsigned *__restrict__ hist, SelState *__restrict__ st, int nsel, int shift, int nbits,
                        float *__restrict__ out)
{
    __shared__ unsigned h[kSelBins];
    __shared__ unsigned long long part[256];
    __shared__ int found_bin;
    __shared__ unsigned long long found_rank;
    const int bins = 1 << nbits;
    const int tid = threadIdx.x;
    const bool split = nsel > 1 && st[1].prefix != st[0].prefix;
    __syncthreads();                                          // (everyone has read the prefixes before thread 0 moves them)
    for (int sel = 0; sel < nsel; ++sel) {
        const unsigned *src = hist + ((sel && split) ? kSelBins : 0);
        for (int b = tid; b < kSelBins; b += 256) h[b] = b < bins ? src[b] : 0u;
        if (tid == 0) { found_bin = bins - 1; found_rank = 0; }
        __syncthreads();
        unsigned long long mine = 0;
        for (int b = 0; b < 8; ++b) mine += h[tid * 8 + b];
        // exclusive prefix of `mine` over the 256 threads: shuffles inside each wavefront, the four wavefront totals through LDS
        unsigned long long incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(incl, off);
            if ((tid & 63) >= off) incl += up;
        }
        if ((tid & 63) == 63) part[tid >> 6] = incl;
        __syncthreads();
        unsigned long long before = incl - mine;
        for (int w = 0; w < (tid >> 6); ++w) before += part[w];
        const unsigned long long k = st[sel].k;
        if (before <= k && k < before + mine) {           // exactly one thread (or none if k is out of range)
            unsigned long long acc = before;
            int b = tid * 8;
            for (; b < tid * 8 + 7; ++b) {
                if (acc + h[b] > k) break;
                acc += h[b];
            }
            found_bin = b;
            found_rank = k - acc;
        }
        __syncthreads();
        if (tid == 0) {
            st[sel].k = found_rank;
            st[sel].prefix = st[sel].prefix | ((unsigned)found_bin << shift);
        }
        __syncthreads();
    }
    for (int b = tid; b < 2 * kSelBins; b += 256) hist[b] = 0u;
    if (out && tid == 0) {
        const float a = __uint_as_float(st[0].prefix);
        *out = nsel > 1 ? __fdiv_rn(__fadd_rn(a, __uint_as_float(st[1].prefix)), 2.0f) : a;
    }
}

__global__ void __launch_bounds__(256)
gpfq_select_init_kernel(SelState *st, unsigned *hist, unsigned long long k0, unsigned long long k1)
{
    for (int b = threadIdx.x; b < 2 * kSelBins; b += 256) hist[b] = 0u;
    if (threadIdx.x == 0) {
        st[0].prefix = 0; st[0].pad = 0; st[0].k = k0;
        st[1].prefix = 0; st[1].pad = 0; st[1].k = k1;
    }
}

__global__ void gpfq_select_finish_kernel(const SelState *st, int even, float *out)
{
    const float a = __uint_as_float(st[0].prefix);
    if (!even) { *out = a; return; }
    const float b = __uint_as_float(st[1].prefix);
    *out = __fdiv_rn(__fadd_rn(a, b), 2.0f);          // float32 mean of the two middle values
}

size_t median_workspace_bytes() { return 2 * sizeof(SelState) + 2 * kSelBins * sizeof(unsigned) + 64; }

static const int kSelShifts[3] = {21, 10, 0}, kSelWidths[3] = {11, 11, 10};

static SelState *sel_state(void *workspace) { return static_cast<SelState *>(workspace); }
static unsigned *sel_hist(void *workspace) { return reinterpret_cast<unsigned *>(static_cast<char *>(workspace) + 64); }
static int sel_count(int64_t n_total) { return (n_total % 2) == 0 ? 2 : 1; }

// The select with the elements partitioned over ranks (every rank holds the whole layer, each counts one slice): per pass
// the ranks histogram their slices, the caller sums the histograms over the ranks (2 x 8 KiB, one all-reduce), and every
// rank picks the same bins.  launch_median_abs is the one-rank sequence of the same kernels.
hipError_t launch_median_begin(int64_t n_total, void *workspace, hipStream_t stream)
{
    const bool even = (n_total % 2) == 0;
    const unsigned long long k0 = even ? (unsigned long long)(n_total / 2 - 1) : (unsigned long long)(n_total / 2);
    hipLaunchKernelGGL(gpfq_select_init_kernel, dim3(1), dim3(256), 0, stream, sel_state(workspace), sel_hist(workspace), k0,
                       (unsigned long long)(n_total / 2));
    return hipGetLastError();
}

hipError_t launch_median_count(const float *W_local, int64_t n_local, int64_t n_total, int pass, void *workspace, hipStream_t stream)
{
    if (n_local <= 0) return hipSuccess;
    int64_t blocks = (n_local + kSelThreads * 16 - 1) / (kSelThreads * 16);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(gpfq_select_hist_kernel, dim3((unsigned)blocks), dim3(kSelThreads), 0, stream,
                       W_local, n_local, sel_state(workspace), sel_count(n_total), kSelShifts[pass], kSelWidths[pass], sel_hist(workspace));
    return hipGetLastError();
}

static hipError_t median_pick(int64_t n_total, int pass, void *workspace, float *out, hipStream_t stream)
{
    hipLaunchKernelGGL(gpfq_select_pick_kernel, dim3(1), dim3(256), 0, stream, sel_hist(workspace), sel_state(workspace),
                       sel_count(n_total), kSelShifts[pass], kSelWidths[pass], out);
    return hipGetLastError();
}

hipError_t launch_median_pick(int64_t n_total, int pass, void *workspace, hipStream_t stream)
{
    return median_pick(n_total, pass, workspace, nullptr, stream);
}

hipError_t launch_median_end(int64_t n_total, void *workspace, float *out, hipStream_t stream)
{
    hipLaunchKernelGGL(gpfq_select_finish_kernel, dim3(1), dim3(1), 0, stream, sel_state(workspace), sel_count(n_total) - 1, out);
    return hipGetLastError();
}

// ---- one-GPU form (round 6): two passes over the data, two launches ----------------------------------------------------------
// The same exact select with the 31 key bits cut 15 + 16 instead of 11 + 11 + 10: a workgroup's histogram is 128 KiB of this chip's
// 160 KiB LDS -- 32768 counters in pass A; 65536 SIXTEEN-bit counters, two to a word, in pass B, where a workgroup takes at most 65535
// elements so that no counter can carry into its neighbour -- and each pass's pick is done by the LAST workgroup of the pass's own launch
// to arrive (a ticket behind the histogram merges, which are agent-scope atomics), not by a launch of its own.  4096 x 4096 weights: seven
// launches and three reads of the data (68 us) -> memset + two launches, two reads (profiles/r06).  The sharded protocol above keeps its
// three 16 KiB histograms: those cross the ranks.
struct Sel2Ctl {
    unsigned done, pad0, pad1, pad2;
};
constexpr int kSel2A = 1 << 15, kSel2B = 1 << 16;
// workspace: [SelState x 2, 64 B][ctl 64 B][histA 2 x 32768 u32][histB 2 x 65536 u32][coarseA 1024 u32][coarseB 2 x 1024 u32]
static Sel2Ctl *sel2_ctl(void *ws) { return reinterpret_cast<Sel2Ctl *>(static_cast<char *>(ws) + 64); }
static unsigned *sel2_hist_a(void *ws) { return reinterpret_cast<unsigned *>(static_cast<char *>(ws) + 128); }
static unsigned *sel2_hist_b(void *ws) { return sel2_hist_a(ws) + 2 * kSel2A; }
static unsigned *sel2_coarse_a(void *ws) { return sel2_hist_b(ws) + 2 * kSel2B; }
static unsigned *sel2_coarse_b(void *ws) { return sel2_coarse_a(ws) + kSelThreads; }
size_t median_workspace_bytes_fast(int64_t) { return 128 + (size_t)(2 * kSel2A + 2 * kSel2B + 3 * kSelThreads) * sizeof(unsigned); }

// PASS 0: bins = key >> 16 of every element; PASS 1: bins = key & 0xffff of the elements whose high bits are the class(es) pass 0 picked.
template <int PASS>
__global__ void __launch_bounds__(kSelThreads)
gpfq_median2_kernel(const float *__restrict__ W, int64_t n, int64_t per_block, SelState *__restrict__ st, unsigned *hist, unsigned *coarse,
                    Sel2Ctl *__restrict__ ctl, int nsel, unsigned long long k0, unsigned long long k1, float *__restrict__ out,
                    DevAlphabet *alpha_out, double alphabet_scalar, AlphabetArg unit, int want_sym)
{
    extern __shared__ unsigned h[];                               // PASS 0: [32768] counters; PASS 1: [32768] words of two 16-bit counters (128 KiB either way)
    __shared__ unsigned long long part[16];
    __shared__ int is_last;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int WORDS = kSel2A;
    constexpr int CSH = PASS == 0 ? 5 : 6;                        // fine bin >> CSH = its owner (32768 / 65536 bins over 1024 threads)                                 // per histogram, either pass
#ifdef GPFQ_MEDIAN_STAMPS
    unsigned long long *stamps = reinterpret_cast<unsigned long long *>(ctl) + 2 + PASS * 3;      // diagnostic build: [pass][3] in the control block's spare words
    const unsigned long long ts0 = __builtin_amdgcn_s_memrealtime();
#endif
    const unsigned p0 = PASS == 0 ? 0u : st[0].prefix, p1 = (PASS == 0 || nsel < 2) ? p0 : st[1].prefix;
    const bool split = p1 != p0;
    const int nh = split ? 2 : 1;                                 // (the two middle values of an even count in different classes -- rare: the range is read once per class)
    unsigned pcur = p0;
    auto count = [&](float w) {
        const unsigned key = __float_as_uint(w) & 0x7fffffffu;
        if (PASS == 0) atomicAdd(&h[key >> 16], 1u);
        else {
            const unsigned cls = key & 0xffff0000u, lo = key & 0xffffu;
            if (cls == pcur) atomicAdd(&h[lo >> 1], 1u << (16 * (lo & 1u)));
        }
    };
    // Grid-stride over 16-byte reads, at most 16 per thread (65536 elements per workgroup): consecutive workgroups read consecutive
    // 16 KiB.  Pass 1's 16-bit counters must not carry into their neighbours: a workgroup counts at most 65532 elements in LDS -- thread 0's
    // first read and the tail that is no whole read go straight to the global histogram.
    const int64_t n4 = n / 4;                                     // (W is 16-byte aligned: launch_median_abs)
    const float4 *W4 = reinterpret_cast<const float4 *>(W);
    const int64_t stride4 = (int64_t)gridDim.x * kSelThreads;
    auto count_global = [&](float w, int hs) {                    // (pass 1 only)
        const unsigned key = __float_as_uint(w) & 0x7fffffffu, cls = key & 0xffff0000u, lo = key & 0xffffu;
        if (cls == (hs ? p1 : p0)) { atomicAdd(&hist[hs * kSel2B + lo], 1u); atomicAdd(&coarse[hs * kSelThreads + (lo >> CSH)], 1u); }
    };
    for (int hs = 0; hs < nh; ++hs) {
        pcur = hs ? p1 : p0;
        for (int b = tid; b < WORDS; b += kSelThreads) h[b] = 0;
        __syncthreads();
        int64_t i = (int64_t)blockIdx.x * kSelThreads + tid;
        if (PASS == 1 && tid == 0 && i < n4) {
            const float4 v = W4[i];
            count_global(v.x, hs); count_global(v.y, hs); count_global(v.z, hs); count_global(v.w, hs);
            i += stride4;
        }
        for (; i < n4; i += stride4) {
            const float4 v = W4[i];
            count(v.x); count(v.y); count(v.z); count(v.w);
        }
        for (int64_t j = n4 * 4 + (int64_t)blockIdx.x * kSelThreads + tid; j < n; j += stride4) {      // (at most three elements)
            if (PASS == 1) count_global(W[j], hs);
            else {
                const unsigned bin = (__float_as_uint(W[j]) & 0x7fffffffu) >> 16;
                atomicAdd(&hist[bin], 1u); atomicAdd(&coarse[bin >> CSH], 1u);
            }
        }
        __syncthreads();
        // merge: the fine counts in rounds of 1024 consecutive words (a wavefront's atomics of a round fall into four lines: one owner's
        // words per thread made every one of them 64 lines, 92 us of pass 0) ...
        if (PASS == 0) {
            for (int b = tid; b < WORDS; b += kSelThreads) {
                const unsigned cnt = h[b];
                if (cnt) atomicAdd(&hist[b], cnt);
            }
        } else {
            for (int b = tid; b < WORDS; b += kSelThreads) {
                const unsigned cnt = h[b];
                if (cnt & 0xffffu) atomicAdd(&hist[hs * kSel2B + 2 * b], cnt & 0xffffu);
                if (cnt >> 16) atomicAdd(&hist[hs * kSel2B + 2 * b + 1], cnt >> 16);
            }
        }
        // ... and the coarse count of thread t's own bins -- the 32 words from 32 t: 32 bins of pass 0, 64 sixteen-bit bins of pass 1 -- as
        // a sum in a register (the words of a wavefront's lanes taken rotated by the lane: 32 different banks per round)
        {
            constexpr int WPT = WORDS / kSelThreads;
            unsigned tot = 0;
#pragma unroll 8
            for (int j = 0; j < WPT; ++j) {
                const unsigned cnt = h[tid * WPT + ((j + lane) & (WPT - 1))];
                tot += PASS == 0 ? cnt : (cnt & 0xffffu) + (cnt >> 16);
            }
            if (tot) atomicAdd(&coarse[hs * kSelThreads + tid], tot);
        }
        __syncthreads();
    }
#ifdef GPFQ_MEDIAN_STAMPS
    if (blockIdx.x == 0 && tid == 0) stamps[0] = __builtin_amdgcn_s_memrealtime() - ts0;          // workgroup 0: zeroing + reads + merge
#endif
    // ---- the last workgroup to get here picks (each wavefront waits for its own merges -- agent-scope atomics -- to be acknowledged: a
    // workgroup-scope release, no cache maintenance; an agent-scope fence per thread writes the L2 back thousands of times per pass) ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) is_last = __hip_atomic_fetch_add(&ctl->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (!is_last) return;
#ifdef GPFQ_MEDIAN_STAMPS
    const unsigned long long ts1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) stamps[1] = ts1 - ts0;                                                         // the last workgroup: from its start to its ticket
#endif
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (PASS == 0 && tid == 0) {
        st[0].prefix = 0; st[0].pad = 0; st[0].k = k0;
        st[1].prefix = 0; st[1].pad = 0; st[1].k = k1;
    }
    __syncthreads();
    constexpr int BINS = PASS == 0 ? kSel2A : kSel2B, PER = BINS / kSelThreads;      // 32 / 64 bins per thread
    // One scan of the histogram serves BOTH order statistics while they share their class (always in pass 0, nearly always in pass 1:
    // the two middle ranks of an even count are adjacent); when they have parted, the second class's histogram is scanned for the second.
    // (Two scans for two ranks in one histogram made the pick 12 / 20 us of a 26 / 36 us pass: `tools/median_probe.py`.)
    __shared__ int f_thread[2];
    __shared__ unsigned long long f_before[2];
    unsigned long long rank_of[2];
    rank_of[0] = st[0].k; rank_of[1] = nsel > 1 ? st[1].k : 0;
    for (int scan = 0; scan < (nsel > 1 && split ? 2 : 1); ++scan) {
        const unsigned *src = hist + (scan ? kSel2B : 0), *csrc = coarse + (scan ? kSelThreads : 0);
        const int s_lo = scan, s_hi = (nsel > 1 && !split) ? 2 : scan + 1;       // the order statistics this scan serves
        if (tid < 2) { f_thread[tid] = -1; f_before[tid] = 0; }
        // (plain 16-byte loads: the acquire fence above has dropped this workgroup's stale lines, and every merge was acknowledged before
        //  its workgroup took a ticket -- PER single agent-scope loads in sequence cost 50-90 us per pick)
        // Thread tid owns the PER consecutive bins from tid * PER; their sum is word tid of the pass's COARSE histogram, which every
        // workgroup merged beside the fine one: one 4 KiB read instead of the whole 128 / 256 KiB histogram, whose lines -- last written by
        // agent-scope atomics of all eight XCDs -- one compute unit pulled in at 20 GB/s (7 / 12.6 us of pick in a 22 / 30 us pass,
        // whatever the access pattern: profiles/r06/median_probe.txt).
        const unsigned long long mine = csrc[tid];
        unsigned long long incl = mine;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        __syncthreads();
        if (lane == 63) part[tid >> 6] = incl;
        __syncthreads();
        unsigned long long before = incl - mine;
        for (int w = 0; w < (tid >> 6); ++w) before += part[w];
        for (int sel = s_lo; sel < s_hi; ++sel)
            if (before <= rank_of[sel] && rank_of[sel] < before + mine) { f_thread[sel] = tid; f_before[sel] = before; }   // one thread per rank (none if it is out of range)
        __syncthreads();
        // the bin inside that thread's PER (<= 64) bins: one parallel read and a scan over the lanes of a wavefront per order statistic
        const int wsel = s_lo + (tid >> 6);
        if (wsel < s_hi) {
            const int T = f_thread[wsel];
            const unsigned c = (T >= 0 && lane < PER) ? src[T * PER + lane] : 0u;
            unsigned long long inc = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned long long up = __shfl_up(inc, off);
                if (lane >= off) inc += up;
            }
            const unsigned long long kk = rank_of[wsel] - f_before[wsel], excl = inc - c;
            const bool hit = T >= 0 && excl <= kk && kk < inc;
            const unsigned long long any = __ballot(hit);
            if (hit || (any == 0ull && lane == 0)) {              // (a rank beyond the data -- cannot happen -- keeps the last bin, rank 0)
                const int fb = hit ? T * PER + lane : BINS - 1;
                st[wsel].k = hit ? kk - excl : 0ull;
                st[wsel].prefix = st[wsel].prefix | ((unsigned)fb << (PASS == 0 ? 16 : 0));
            }
        }
        __syncthreads();
    }
#ifdef GPFQ_MEDIAN_STAMPS
    if (tid == 0) stamps[2] = __builtin_amdgcn_s_memrealtime() - ts1;                            // the pick
#endif
    if (tid == 0) {
        ctl->done = 0;
        if (PASS == 1) {
            const float a = __uint_as_float(st[0].prefix);
            const float med = nsel > 1 ? __fdiv_rn(__fadd_rn(a, __uint_as_float(st[1].prefix)), 2.0f) : a;   // float32 mean of the two middle values
            if (out) *out = med;
            // (gpfq_layer_alphabet_from_kernel: the layer alphabet rad * unit, :544-545, formed right here -- no launch of its own)
            if (alpha_out) form_device_alphabet(alpha_out, med, alphabet_scalar, unit, want_sym);
        }
    }
}

hipError_t launch_median_abs(const float *W, int64_t n, float *out, void *workspace, hipStream_t stream, size_t workspace_bytes,
                             void *alpha_out, double alphabet_scalar, const AlphabetArg *unit, int want_sym)
{
    const AlphabetArg no_unit{};
    if (workspace_bytes < median_workspace_bytes_fast(n) || (uintptr_t)W % 16 != 0) {
        // the minimal workspace of rounds 1-5 (gpfq_median_abs_workspace_bytes()), or a kernel that cannot be read 16 bytes at a time: the three-pass sequence
        hipError_t e = launch_median_begin(n, workspace, stream);
        for (int p = 0; p < 3 && e == hipSuccess; ++p) {
            e = launch_median_count(W, n, n, p, workspace, stream);
            if (e == hipSuccess) e = median_pick(n, p, workspace, p == 2 ? out : nullptr, stream);
        }
        return (e == hipSuccess && alpha_out) ? hipErrorInvalidValue : e;      // (the fused alphabet needs the two-pass form: the caller checks first)
    }
    hipError_t e = hipMemsetAsync(workspace, 0, median_workspace_bytes_fast(n), stream);    // state, control words, both histograms
    if (e != hipSuccess) return e;
    const bool even = (n % 2) == 0;
    const unsigned long long k0 = even ? (unsigned long long)(n / 2 - 1) : (unsigned long long)(n / 2), k1 = (unsigned long long)(n / 2);
    // contiguous ranges of a multiple of four elements, one workgroup per compute unit for a layer-sized kernel (a histogram is most of a
    // compute unit's LDS: a 257th workgroup would wait for a whole round)
    int64_t per = (n + 255) / 256;
    per = (per + 4095) & ~(int64_t)4095;                          // whole rounds of a workgroup's 1024 16-byte reads, at most 16 of them (pass 1's 16-bit counters)
    if (per < 4096) per = 4096;
    if (per > 65536) per = 65536;
    const unsigned grid = (unsigned)((n + per - 1) / per);
    const size_t lds = (size_t)kSel2A * sizeof(unsigned);         // one 128 KiB histogram per workgroup, either pass
    e = ensure_dynamic_lds((const void *)gpfq_median2_kernel<0>, kSel2A * sizeof(unsigned));
    if (e == hipSuccess) e = ensure_dynamic_lds((const void *)gpfq_median2_kernel<1>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(gpfq_median2_kernel<0>, dim3(grid), dim3(kSelThreads), kSel2A * sizeof(unsigned), stream, W, n, per, sel_state(workspace),
                       sel2_hist_a(workspace), sel2_coarse_a(workspace), sel2_ctl(workspace), sel_count(n), k0, k1, (float *)nullptr,
                       (DevAlphabet *)nullptr, 0.0, no_unit, 0);
    hipLaunchKernelGGL(gpfq_median2_kernel<1>, dim3(grid), dim3(kSelThreads), lds, stream, W, n, per, sel_state(workspace),
                       sel2_hist_b(workspace), sel2_coarse_b(workspace), sel2_ctl(workspace), sel_count(n), k0, k1, out,
                       static_cast<DevAlphabet *>(alpha_out), alphabet_scalar, unit ? *unit : no_unit, want_sym);
    return hipGetLastError();
}

}  // namespace gpfq
