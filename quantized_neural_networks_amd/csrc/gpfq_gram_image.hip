// Conv2D hot path for 3x3 / stride 1 / rate 1 kernels without patch matrices.
//
// The reference builds, per input channel, the two patch matrices (scripts/quantized_network.py:729-809,
// :123-183) and runs every filter's 9-step recurrence on their rows (:185-233).  The decisions only need
// inner products of patch rows (gpfq_gram.hip), and row t = (ky, kx) of a patch matrix is the channel plane
// shifted by (ky - pad, kx - pad) with zeros outside the image, so those inner products are correlations of
// the plane with itself at pairs of shifts.  This file accumulates the Gram records of ALL channels of a
// shard straight from the channel planes in one launch:
//
//     G1[t][s] = <Xq_t, X_s>,  G2[t][s] = <Xq_t, Xq_s>  (s <= t),   nx2[s] = <X_s, X_s>
//
// i.e. 99 float64 FMAs per output position and channel, ~9x less HBM traffic than extracting patches.
// The kernel is bound by the FP64 VALU rate.
//
// Work split: a workgroup of 4 wavefronts stages a band of output rows (their input rows with the zero
// ring, float32) in LDS.  Wavefronts work in pairs on the same strips: a lane handles a strip of S
// consecutive output positions of one row from a 3 x (S+2) register window, and the two wavefronts of a
// pair own the columns s in {0,3,5,7,8} and {1,2,4,6} of the record (49 + 50 sums), which keeps a wavefront
// under 128 accumulator registers and two of them resident per SIMD.
#include <atomic>
#include <type_traits>

#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"
#include "gpfq_roles.hpp"

namespace gpfq {

constexpr int kImgThreads = 256;
constexpr int kImgStripsPerBand = 256;      // 2 sweeps of the 2 x 64 strip lanes per staged band
constexpr int kImgBlocks = 1024;            // upper bound of co-resident workgroups (sizes the partials)
constexpr int kImgMaxLds = 64 * 1024;
constexpr int64_t kRec9 = 9 * 9 * 2 + 9;

struct ImgParams {
    const float *act_w, *act_q;
    int64_t plane;        // n*H*W floats per channel
    int n, H, W, pad;
    int rpad, cpad;       // zero rows above an image / zero columns left of it in the staged band (pad, pad; shift form: 2, 2)
    int PH, oh, SPR;      // padded plane height, output rows per image, strips per output row
    int grows;            // n*oh output rows in total
    int RB, nbands;       // output rows per band
    int LP, lrows, lpr_log2;
    float inv_SPR, inv_oh, inv_PH;
    int same_act;
    int nbx;
    double *part;         // [nch][nbx*2] Gram records (N = 9)
    int *negflag;         // [nch], set when a channel has a negative activation
};

// floor(v / d) for 0 <= v < 2^24 (exact in float): float estimate, one correction step.
__device__ __forceinline__ int div_small(int v, int d, float inv, int &rem)
{
    int q = (int)((float)v * inv);
    const int r = v - q * d;
    const int fix = (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);      // branch-free: the estimate is off by at most one
    rem = r - fix * d;
    return q + fix;
}

__host__ __device__ constexpr bool in_group(int g, int s)
{
    return g == 0 ? (s == 0 || s == 3 || s == 5 || s == 7 || s == 8) : (s == 1 || s == 2 || s == 4 || s == 6);
}

template <int S>
__device__ __forceinline__ void load_window_row(const float *p, float (&v)[S + 2])
{
    if constexpr (S == 4) {                        // p is 16-byte aligned (LP % 4 == 0, strips start at 4*sx)
        const float4 a = *reinterpret_cast<const float4 *>(p);
        const float2 b = *reinterpret_cast<const float2 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y;
    } else if constexpr (S == 2) {                 // 8-byte aligned
        const float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
#pragma unroll
        for (int i = 0; i < S + 2; ++i) v[i] = p[i];
    }
}

// Stage padded rows [v0, v0 + nv) of channel blockIdx.y (both planes) into LDS: row r of image b's padded plane is image row
// r - rpad (zeros outside), columns shifted by cpad.  Thread role: column piece st_c4 of rows st_row0, st_row0 + st_step, ...
__device__ __forceinline__ void stage_band(const ImgParams &p, const float *__restrict__ pw, const float *__restrict__ pq,
                                           float *__restrict__ lw, float *__restrict__ lq, int v0, int nv, unsigned &signs)
{
    const int L4 = p.LP >> 2;
    const int st_c4 = threadIdx.x & ((1 << p.lpr_log2) - 1), st_row0 = threadIdx.x >> p.lpr_log2;
    const int st_step = kImgThreads >> p.lpr_log2;
    const int st_ix = 4 * st_c4 - p.cpad;
    const bool st_inside = st_ix >= 0 && st_ix + 3 < p.W;
        __syncthreads();                                   // the previous band has been consumed
        if (st_c4 < L4) {
            // this thread's column piece is fixed; walk its rows st_row0, st_row0 + st_step, ... of the band
            int r;
            int b = div_small(v0 + st_row0, p.PH, p.inv_PH, r);
            for (int row = st_row0; row < nv; row += st_step) {
                const int iy = r - p.rpad;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
                if (b < p.n && iy >= 0 && iy < p.H) {
                    const unsigned o = ((unsigned)b * p.H + iy) * p.W + st_ix;      // plane < 2^30 floats
                    if (st_inside) {
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        const f4u t = *reinterpret_cast<const f4u *>(pw + o);
                        a = make_float4(t.x, t.y, t.z, t.w);
                        if (!p.same_act) {
                            const f4u u = *reinterpret_cast<const f4u *>(pq + o);
                            c = make_float4(u.x, u.y, u.z, u.w);
                        }
                    } else {
                        if (st_ix >= 0 && st_ix < p.W) { a.x = pw[o]; c.x = pq[o]; }
                        if (st_ix + 1 >= 0 && st_ix + 1 < p.W) { a.y = pw[o + 1]; c.y = pq[o + 1]; }
                        if (st_ix + 2 >= 0 && st_ix + 2 < p.W) { a.z = pw[o + 2]; c.z = pq[o + 2]; }
                        if (st_ix + 3 >= 0 && st_ix + 3 < p.W) { a.w = pw[o + 3]; c.w = pq[o + 3]; }
                    }
                }
                *reinterpret_cast<float4 *>(lw + row * p.LP + 4 * st_c4) = a;
                if (!p.same_act) *reinterpret_cast<float4 *>(lq + row * p.LP + 4 * st_c4) = c;
                neg_track(signs, a);
                neg_track(signs, c);
                r += st_step;
                while (r >= p.PH) { r -= p.PH; ++b; }
            }
        }
}

template <int S, int G>
__device__ __forceinline__ void image_gram_body(const ImgParams &p, float *lds)
{
    Gram9 acc;                                       // only the columns of group G are ever touched
    gram9_zero(acc);
    const int lane = threadIdx.x & 63, pair = threadIdx.x >> 7;
    const float *pw = p.act_w + (int64_t)blockIdx.y * p.plane;
    const float *pq = p.act_q + (int64_t)blockIdx.y * p.plane;
    float *lw = lds;
    float *lq = p.same_act ? lds : lds + (size_t)p.lrows * p.LP;
    unsigned signs = 0;                                  // neg_track() of everything this thread staged

    for (int band = blockIdx.x; band < p.nbands; band += gridDim.x) {
        const int g0 = band * p.RB;
        const int rows = min(p.RB, p.grows - g0);
        int oy0, oy1;
        const int b0 = div_small(g0, p.oh, p.inv_oh, oy0);
        const int b1 = div_small(g0 + rows - 1, p.oh, p.inv_oh, oy1);
        const int v0 = b0 * p.PH + oy0;
        const int nv = b1 * p.PH + oy1 + 2 - v0 + 1;
        stage_band(p, pw, pq, lw, lq, v0, nv, signs);
        __syncthreads();
        const int nstrips = rows * p.SPR;
        for (int q = pair * 64 + lane; q < nstrips; q += 128) {
            int sx, oy;
            const int grow = div_small(q, p.SPR, p.inv_SPR, sx);
            const int b = div_small(g0 + grow, p.oh, p.inv_oh, oy);
            const int off = (b * p.PH + oy - v0) * p.LP + S * sx;
            // 3 x (S+2) windows: the quantized plane in float64, the analog one converted where it is used
            double wq[3][S + 2];
            float wx[3][S + 2];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float r[S + 2];
                load_window_row<S>(lq + off + ky * p.LP, r);
#pragma unroll
                for (int i = 0; i < S + 2; ++i) wq[ky][i] = (double)r[i];
                load_window_row<S>(lw + off + ky * p.LP, wx[ky]);
            }
#pragma unroll
            for (int e = 0; e < S; ++e) {
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    if (in_group(G, s)) {
                        const double x = (double)wx[s / 3][s % 3 + e], y = wq[s / 3][s % 3 + e];
                        acc.nx[s] = fma(x, x, acc.nx[s]);
#pragma unroll
                        for (int t = s; t < 9; ++t) {
                            const double a = wq[t / 3][t % 3 + e];
                            acc.g[t * (t + 1) / 2 + s][0] = fma(a, x, acc.g[t * (t + 1) / 2 + s][0]);
                            acc.g[t * (t + 1) / 2 + s][1] = fma(a, y, acc.g[t * (t + 1) / 2 + s][1]);
                        }
                    }
                }
            }
        }
    }
    if (__ballot(neg_seen(signs)) && lane == 0) atomicOr(p.negflag + blockIdx.y, 1);   // a negative activation was seen
    // the two wavefronts of a pair fill disjoint columns of one record
    double *out = p.part + (((int64_t)blockIdx.y * p.nbx + blockIdx.x) * 2 + pair) * kRec9;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        if (in_group(G, s)) {
#pragma unroll
            for (int t = s; t < 9; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double v = wave_sum(acc.g[t * (t + 1) / 2 + s][k]);
                    if (lane == 0) out[(t * 9 + s) * 2 + k] = v;
                }
            const double v = wave_sum(acc.nx[s]);
            if (lane == 0) out[162 + s] = v;
        }
    }
}

template <int S>
__global__ void __launch_bounds__(kImgThreads, 2)
gpfq_gram_image_kernel(ImgParams p)
{
    extern __shared__ __attribute__((aligned(16))) float img_lds[];
    // the wavefront index is uniform: both instantiations run the same staging code and barriers
    const int group = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 1);
    if (group == 0) image_gram_body<S, 0>(p, img_lds);
    else image_gram_body<S, 1>(p, img_lds);
}


// ---- the shift form (SAME padding) ------------------------------------------------------------------------
// Row t = (ky, kx) of a patch matrix is the plane shifted by (ky - 1, kx - 1), so with A = o + (ky_t - 1, kx_t - 1)
//
//     G1[t][s] = sum over output positions o of q[o + offs_t - 1] x[o + offs_s - 1] = sum_A q[A] x[A + d],   d = offs_s - offs_t,
//
// over the input positions A whose output position o = A - offs_t + 1 lies in the image (x, q read as zero outside).  For A at
// least one step away from the border every t qualifies, and the sum depends on (t, s) only through d: the lower triangle s <= t
// needs the 13 shifts d = (dy, dx) with dy in {-2, -1} (any dx) or dy = 0, dx <= 0, so an interior position costs
// 13 + 13 + 1 = 27 float64 FMAs (C1[d] += q[A] x[A+d], C2[d] += q[A] q[A+d], C3 += x[A]^2) instead of the 99 of the
// per-output-position form above.  Border positions (first / last row or column) qualify only for some t -- top row: ky_t <= 1,
// bottom row: ky_t >= 1, likewise the columns -- so they are accumulated per border CLASS (8 classes: {top, middle, bottom} x
// {left, middle, right} without the interior) by DEDICATED THREADS of the same kernel, from the band the workgroup has in LDS
// anyway (a separate pass over the border columns would touch every cache line of the planes again: a row of 56 floats is two
// lines): a thread keeps one role for the whole launch, so its private sums belong to one class.
// The record is assembled as
//
//     G1[t][s] = sum over the classes c in which t qualifies of C1_c[d(t, s)]      (G2 likewise;  nx2[s]: classes in which s qualifies).
//
// Everything is ADDED (never a full sum minus a border): a patch row that is identically zero keeps an exactly zero norm (rule (i),
// :83-84), and the float32 row norms (float)sqrt(G2[t][t]) come from sums of squares as before.
constexpr int kShiftN = 27;            // C1[13], C2[13], C3

template <int N>
__device__ __forceinline__ void load_row_f(const float *p, float (&v)[N])
{
    if constexpr (N == 8) {                        // S = 4: 16-byte aligned
        const float4 a = *reinterpret_cast<const float4 *>(p), b = *reinterpret_cast<const float4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (N == 6) {                 // S = 2: 8-byte aligned
        const float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + 2),
                     c = *reinterpret_cast<const float2 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
    } else {
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[i];
    }
}

// The staged band has two zero rows above every image and two zero columns on either side.  A band is RB image rows with
// RB * (SPR + 2) + 2 * (SPR + 2) <= 256, and every thread has ONE role:
//   [0, RB*SPR)            a strip of S positions of band row t / SPR, middle columns enabled        (class middle-middle; rows
//   next 2*RB              the strip that holds the first (even) / last (odd) column of band row j/2,  0 and H-1 disabled)
//                          only that position enabled                                                 (middle-left / -right)
//   next SPR + 2           the strips of the band's FIRST image rows (y = 0): SPR with the middle columns, then the two corners
//   next SPR + 2           the same for the band's LAST image rows (y = H - 1)
// All roles run the same instruction stream on a 3 x (S + 4) window (rows y-2 .. y, columns x0-2 .. x0+S+1); they differ in which
// positions of the strip are enabled.  Middle rows are one item per thread and band; a band holds rows 0 / H-1 of at most
// RB / H + 1 images, which their threads loop over.
// SAME_ACT: both networks see the same planes (a first layer): C2 = C1, only C1 is accumulated.
template <int S, bool SAME_ACT>
__global__ void __launch_bounds__(kImgThreads, 2)
gpfq_gram_shift_kernel(ImgParams p)
{
    extern __shared__ __attribute__((aligned(16))) float img_lds[];
    double c1[13], c2[13], c3 = 0.0;
#pragma unroll
    for (int i = 0; i < 13; ++i) { c1[i] = 0.0; c2[i] = 0.0; }
    const int tid = threadIdx.x, lane = tid & 63;
    const float *pw = p.act_w + (int64_t)blockIdx.y * p.plane;
    const float *pq = p.act_q + (int64_t)blockIdx.y * p.plane;
    float *lw = img_lds;
    float *lq = p.same_act ? img_lds : img_lds + (size_t)p.lrows * p.LP;
    unsigned signs = 0;

    // role of this thread: cls = 3 * cy + cx (cy: 0 first row, 1 middle rows, 2 last row; cx likewise), strip sx, enabled columns
    const int nI = p.RB * p.SPR, nC = 2 * p.RB, nR = p.SPR + 2;
    int cy = 1, cx = 1, my_row = 0, sx = 0;
    bool active = true;
    if (tid < nI) my_row = div_small(tid, p.SPR, p.inv_SPR, sx);
    else if (tid < nI + nC) { my_row = (tid - nI) >> 1; cx = ((tid - nI) & 1) ? 2 : 0; }
    else if (tid < nI + nC + 2 * nR) {
        int j = tid - nI - nC;
        cy = j < nR ? 0 : 2;
        j -= j < nR ? 0 : nR;
        if (j < p.SPR) sx = j; else cx = j == p.SPR ? 0 : 2;
    } else active = false;
    if (cx == 0) sx = 0;
    if (cx == 2) sx = p.SPR - 1;
    const int x0 = S * sx;
    const int xlo = cx == 1 ? 1 : (cx == 0 ? 0 : p.W - 1), xhi = cx == 1 ? p.W - 2 : xlo;

    for (int band = blockIdx.x; band < p.nbands; band += gridDim.x) {
        const int g0 = band * p.RB;
        const int rows = min(p.RB, p.grows - g0);
        int oy0, oy1;
        const int b0 = div_small(g0, p.oh, p.inv_oh, oy0);
        const int b1 = div_small(g0 + rows - 1, p.oh, p.inv_oh, oy1);
        const int v0 = b0 * p.PH + oy0;                     // padded row of image row oy0 - 2
        const int nv = b1 * p.PH + oy1 + 2 - v0 + 1;
        stage_band(p, pw, pq, lw, lq, v0, nv, signs);
        __syncthreads();
        // band rows of this thread: its own one (middle rows), or every image row 0 / H-1 in the band
        int r = cy == 1 ? my_row : (cy == 0 ? (oy0 == 0 ? 0 : p.H - oy0) : p.H - 1 - oy0);
        const int rstep = cy == 1 ? p.RB : p.H;
        for (; active && r < rows; r += rstep) {
            int y;
            const int b = div_small(g0 + r, p.oh, p.inv_oh, y);
            const int off = (b * p.PH + y - v0) * p.LP + x0;
            const bool rowin = cy == 1 ? (y >= 1 && y <= p.H - 2) : true;      // (rows 0 / H-1 belong to their own threads)
            double qa[S];
            {   // own row (dy = 0): q[A], x[A] and the shifts dx = -2, -1, 0
                float xr[S + 4], qr[S + 4];
                load_row_f<S + 4>(lw + off + 2 * p.LP, xr);
                load_row_f<S + 4>(lq + off + 2 * p.LP, qr);
                double xd[S + 2], qd[S + 2];
#pragma unroll
                for (int i = 0; i < S + 2; ++i) { xd[i] = (double)xr[i]; qd[i] = (double)qr[i]; }
#pragma unroll
                for (int e = 0; e < S; ++e) {
                    const bool in = rowin && x0 + e >= xlo && x0 + e <= xhi;
                    qa[e] = in ? (double)qr[e + 2] : 0.0;
                    const double xm = in ? (double)xr[e + 2] : 0.0;
                    c3 = fma(xm, xm, c3);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        c1[10 + j] = fma(qa[e], xd[e + j], c1[10 + j]);
                        if (!SAME_ACT) c2[10 + j] = fma(qa[e], qd[e + j], c2[10 + j]);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {                   // rows y - 2, y - 1: dx = -2 .. 2
                float xr[S + 4], qr[S + 4];
                load_row_f<S + 4>(lw + off + k * p.LP, xr);
                load_row_f<S + 4>(lq + off + k * p.LP, qr);
                double xd[S + 4], qd[S + 4];
#pragma unroll
                for (int i = 0; i < S + 4; ++i) { xd[i] = (double)xr[i]; qd[i] = (double)qr[i]; }
#pragma unroll
                for (int e = 0; e < S; ++e)
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        c1[k * 5 + j] = fma(qa[e], xd[e + j], c1[k * 5 + j]);
                        if (!SAME_ACT) c2[k * 5 + j] = fma(qa[e], qd[e + j], c2[k * 5 + j]);
                    }
            }
        }
    }
    if (__ballot(neg_seen(signs)) && lane == 0) atomicOr(p.negflag + blockIdx.y, 1);
    // class sums of this workgroup through the (now idle) band LDS, nine accumulators at a time, summed in thread order
    double *sc = reinterpret_cast<double *>(img_lds);       // [9 values + the class][256]
    const int cls = active ? 3 * cy + cx : -1;
    double acc[kShiftN];
#pragma unroll
    for (int i = 0; i < 13; ++i) { acc[i] = c1[i]; acc[13 + i] = SAME_ACT ? c1[i] : c2[i]; }
    acc[26] = c3;
    double *out = p.part + ((int64_t)blockIdx.y * p.nbx + blockIdx.x) * 9 * kShiftN;
#pragma unroll
    for (int round = 0; round < 3; ++round) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 9; ++i) sc[i * kImgThreads + tid] = acc[round * 9 + i];
        sc[9 * kImgThreads + tid] = (double)cls;
        __syncthreads();
        if (tid < 81) {
            const int c = tid / 9, i = tid - 9 * c;
            double v = 0.0;
            for (int l = 0; l < kImgThreads; ++l) v += (int)sc[9 * kImgThreads + l] == c ? sc[i * kImgThreads + l] : 0.0;
            out[c * kShiftN + round * 9 + i] = v;
        }
    }
}

// Class sums -> the N = 9 Gram record of a channel (layout of gpfq_gram.hip) + the float32 row norms.
__global__ void __launch_bounds__(256)
gpfq_gram_shift_combine_kernel(const double *__restrict__ part, int nparts, double *__restrict__ gram, float *__restrict__ nrm32)
{
    __shared__ double T[9][kShiftN];
    const int64_t ch = blockIdx.x;
    // one wavefront per entry of the workgroups' [class][27] partials (few channels mean hundreds of workgroups per channel);
    // lane l takes workgroups l, l + 64, ...: a fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int idx = wave; idx < 9 * kShiftN; idx += 4) {
        double v = 0.0;
        for (int k = lane; k < nparts; k += 64) v += part[(ch * nparts + k) * 9 * kShiftN + idx];
        v = wave_sum(v);
        if (lane == 0) T[idx / kShiftN][idx % kShiftN] = v;
    }
    __syncthreads();
    // row t = (ky, kx) qualifies in class (cy, cx) unless the class is the top row and ky = 2, the bottom row and ky = 0, ...
    auto qualifies = [](int c, int t) {
        const int cy = c / 3, cx = c - 3 * cy, ky = t / 3, kx = t - 3 * ky;
        return !(cy == 0 && ky == 2) && !(cy == 2 && ky == 0) && !(cx == 0 && kx == 2) && !(cx == 2 && kx == 0);
    };
    for (int e = threadIdx.x; e < (int)kRec9; e += 256) {
        double v = 0.0;
        int t = -1, s = -1, k = -1;
        if (e < 162) {
            k = e & 1; t = (e >> 1) / 9; s = (e >> 1) - 9 * t;
            if (s <= t) {
                const int dy = s / 3 - t / 3, dx = s % 3 - t % 3;            // in the lower half-plane
                const int j = (dy < 0 ? (dy + 2) * 5 + dx + 2 : 10 + dx + 2) + 13 * k;
                for (int c = 0; c < 9; ++c) v += qualifies(c, t) ? T[c][j] : 0.0;
            }
        } else {
            s = e - 162;
            for (int c = 0; c < 9; ++c) v += qualifies(c, s) ? T[c][26] : 0.0;
        }
        gram[ch * kRec9 + e] = v;
        if (nrm32 && k == 1 && s == t) nrm32[ch * 9 + t] = (float)sqrt(v);
    }
}

// ---- the shift form straight from NHWC activations ------------------------------------------------------------
// The same 27 sums per input position and the same nine classes, with the LANES along the channels: a wavefront (a workgroup of
// 64 threads) holds one position of 64 channels at a time (256 contiguous bytes per access) and belongs to ONE class for the whole
// launch -- which positions a class holds is the same for every channel, so the class logic is scalar.  An item is one image x
// one strip of kNhwcStrip columns, walked DOWN the rows with the 3 x (strip + 4) neighbourhood of both tensors in registers as
// float64: the two rows above a position were the wavefront's own rows a moment ago, so an element is requested once per strip
// (+ the halo columns).  That window and the 27 accumulators leave two wavefronts per SIMD, too few to cover HBM latency with
// register loads (the first version: 8.5 ms for a layer the planes form does in 5.9): the rows come through an LDS RING filled
// by LDS-DMA (global_load_lds_dword: no registers), kNhwcDepth - 1 rows ahead of the arithmetic.
// Partial sums: [channel][slot][27], slots grouped by class (NhwcParams::slot_off), added per class in slot order.
struct NhwcParams {
    const float *act_w, *act_q;           // NHWC, offset to the first channel of the shard
    int64_t cin;                          // channels of the tensor (the pixel stride)
    int n, H, W, nch;
    int nslots, slot_off[10];             // slots of class c: [slot_off[c], slot_off[c + 1])
    int halves;                           // image groups G = 1, 2, 4, 8 or 16: shards of at most 64 / G channels -- lanes [g 64/G, (g + 1) 64/G) of a wavefront walk
                                          // the g-th G-th of the images (G divides n): no lane idles because the shard is narrow
    int n_walk;                           // images a lane walks: n / G
    int parts;                            // workgroups per channel of the first combine stage (<= kNhwcParts): ~256 slots each
    unsigned half_off;                    // bytes from an image of one group to its partner in the next
    double *part;
    int *negflag;
};
constexpr int kNhwcStrip = 5;             // positions of an image row per item (the window is kNhwcStrip + 4 columns wide)
constexpr int kNhwcDepth = 4;             // rows of the LDS ring

// The walk of one class.  SW = kNhwcStrip for the classes of the middle columns; SW = 1 for the border-column classes, whose one
// position per row needs five columns, not nine: what bounds this kernel is the rate at which the CU's vector-memory path takes
// LDS-DMA requests (profiles/r03/nhwc_pair_experiment.txt), so a class requests no column and no row it does not sum over --
// rows above the image (zeros) are neither requested nor walked.
template <bool SAME_ACT, int SW>
__device__ __forceinline__ void nhwc_class_walk(const NhwcParams &p, float *ring_base, int cls)
{
    constexpr int NC = SW + 4, D = kNhwcDepth, NT = SAME_ACT ? 1 : 2;
    static_assert(NC == 9 || NC == 5, "the request helpers take rows of nine or five columns");
    float (*ring)[NT][NC][64] = reinterpret_cast<float (*)[NT][NC][64]>(ring_base);
    const int lane = threadIdx.x;
    // (G image groups: at most 64 / G channels, lane group g holds the same channels of the image g n / G further on -- the walk is per
    //  lane, only the address of its pixel differs: a constant in the lane's request offset)
    const int lg = 64 / p.halves;                                  // lanes per image group
    const int hsel = p.halves > 1 ? lane / lg : 0;
    const int64_t ch = p.halves > 1 ? lane - hsel * lg : (int64_t)blockIdx.y * 64 + lane;
    const bool live = ch < p.nch;
    const int k = blockIdx.x - p.slot_off[cls], nk = p.slot_off[cls + 1] - p.slot_off[cls];
    const int cy = cls / 3, cx = cls - 3 * cy;
    const int H = p.H, W = p.W;
    const int ya = cy == 1 ? 1 : (cy == 0 ? 0 : H - 1), yb = cy == 1 ? H - 2 : ya;
    const int xlo = cx == 1 ? 1 : (cx == 0 ? 0 : W - 1), xhi = cx == 1 ? W - 2 : xlo;
    const int ns = (xhi - xlo + SW) / SW;
    const int64_t nitems = (int64_t)p.n_walk * ns;
    // requests: the address of a column is SCALAR (row base + column offset, glds4_row9); the lane adds its channel
    const unsigned voff = (unsigned)(live ? ch : 0) * 4u + (unsigned)hsel * p.half_off;
    const unsigned ring_addr = lds_addr(ring_base);
    const int nr = yb - ya + 3;                                     // rows ya - 2 .. yb of the walk, j = 0 .. nr - 1
    const int j0 = ya < 2 ? 2 - ya : 0;                             // rows j < j0 lie above the image
    const int64_t row_bytes = (int64_t)W * p.cin * 4;

    double c1[13], c2[13], c3 = 0.0;
#pragma unroll
    for (int i = 0; i < 13; ++i) { c1[i] = 0.0; c2[i] = 0.0; }
    unsigned sg = 0;                                                // OR of the bit patterns read: the sign bit says "a negative value"

    // items k, k + nk, ...: the slots that run side by side hold neighbouring strips of the same images
    for (int64_t item = k; item < nitems; item += nk) {
        const int64_t img = item / ns;
        const int xa = xlo + SW * (int)(item - img * ns);          // first position; window columns xa - 2 .. xa + SW + 1
        // row base of walk row 0, less the 2048 that keeps the column offsets below unsigned
        const unsigned long long iw = (unsigned long long)(uintptr_t)(p.act_w + (img * H + (ya - 2)) * (int64_t)W * p.cin) - 2048;
        const unsigned long long iq = (unsigned long long)(uintptr_t)(p.act_q + (img * H + (ya - 2)) * (int64_t)W * p.cin) - 2048;
        // per item: byte offset of window column i (clamped into the image: a column outside is requested from a valid address and
        // read as zero by the consumer) from the row base, less the 256 i the request's immediate adds back, plus the 2048
        unsigned coff[NC];
        unsigned cvalid = 0;
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int col = xa - 2 + i;
            cvalid |= (col >= 0 && col < W) ? 1u << i : 0u;
            const int cc = col < 0 ? 0 : (col >= W ? W - 1 : col);
            coff[i] = (unsigned)((long long)cc * p.cin * 4 - 256 * i + 2048);
        }
        // request row j of the walk into ring slot (j - j0) % D: exactly NT * NC requests per row (static wait counts)
        auto request = [&](int j) {
            const unsigned long long rw = iw + (unsigned long long)((long long)j * row_bytes);
            const unsigned long long rq_ = iq + (unsigned long long)((long long)j * row_bytes);
            const unsigned dst = ring_addr + (unsigned)((((j - j0) % D) * NT) * NC * 256);
            if constexpr (NC == 9) {
                glds4_row9((unsigned)rw, (unsigned)(rw >> 32), coff, voff, dst);
                if (!SAME_ACT) glds4_row9((unsigned)rq_, (unsigned)(rq_ >> 32), coff, voff, dst + NC * 256);
            } else {
                glds4_row5((unsigned)rw, (unsigned)(rw >> 32), coff, voff, dst);
                if (!SAME_ACT) glds4_row5((unsigned)rq_, (unsigned)(rq_ >> 32), coff, voff, dst + NC * 256);
            }
        };
        float rx[NC], rq[NC];
        auto consume = [&](int j) {                                 // wait for row j, read it
            const int ahead = nr - 1 - j < D - 2 ? nr - 1 - j : D - 2;              // rows requested after it so far (row j + D - 1 follows this read)
            if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NC * NT) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(1 * NC * NT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (cvalid == (1u << NC) - 1) {                         // (uniform) every column inside the image: nothing to mask
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    rx[i] = ring[(j - j0) % D][0][i][lane];
                    if (!SAME_ACT) rq[i] = ring[(j - j0) % D][NT - 1][i][lane];
                }
            } else {
#pragma unroll
                for (int i = 0; i < NC; ++i) {
                    const bool ok = (cvalid >> i) & 1u;
                    const float a = ring[(j - j0) % D][0][i][lane];
                    rx[i] = ok ? a : 0.f;
                    if (!SAME_ACT) { const float b = ring[(j - j0) % D][NT - 1][i][lane]; rq[i] = ok ? b : 0.f; }
                }
            }
        };
        static_assert(D == 4, "the wait counts above are written for a ring of four rows");
#pragma unroll
        for (int j = 0; j < D - 1; ++j)
            if (j0 + j < nr) request(j0 + j);
        double Bx[3][NC], Bq[3][NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) { Bx[1][i] = Bq[1][i] = 0.0; Bx[2][i] = Bq[2][i] = 0.0; }
        // rows ya - 2, ya - 1 (j = 0, 1), where inside the image, fill the two buffers above; each consumed row frees its ring slot for row j + D
        if (j0 == 0) {
            consume(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (0 + D - 1 < nr) request(0 + D - 1);
#pragma unroll
            for (int i = 0; i < NC; ++i) { Bx[1][i] = (double)rx[i]; Bq[1][i] = SAME_ACT ? Bx[1][i] : (double)rq[i]; }
        }
        if (j0 <= 1) {
            consume(1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (1 + D - 1 < nr) request(1 + D - 1);
#pragma unroll
            for (int i = 0; i < NC; ++i) { Bx[2][i] = (double)rx[i]; Bq[2][i] = SAME_ACT ? Bx[2][i] : (double)rq[i]; }
        }
        // one row: ROT names the buffers (row y in B[ROT], y - 1 in B[ROT + 2], y - 2 in B[ROT + 1], indices mod 3)
        auto step = [&](int j, auto rot_tag) {
            constexpr int R0 = decltype(rot_tag)::value, R1 = (R0 + 2) % 3, R2 = (R0 + 1) % 3;
            consume(j);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the ring slot is read before it is requested again
            if (j + D - 1 < nr) request(j + D - 1);
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                Bx[R0][i] = (double)rx[i];
                Bq[R0][i] = SAME_ACT ? Bx[R0][i] : (double)rq[i];
                sg |= __float_as_uint(rx[i]) | (SAME_ACT ? 0u : __float_as_uint(rq[i]));
            }
#pragma unroll
            for (int e = 0; e < SW; ++e) {
                if (xa + e <= xhi) {                                // (uniform: the last strip of a row may be short)
                    const double qa = Bq[R0][e + 2], xm = Bx[R0][e + 2];
                    c3 = fma(xm, xm, c3);
#pragma unroll
                    for (int jj = 0; jj < 3; ++jj) {                 // dy = 0, dx = jj - 2
                        c1[10 + jj] = fma(qa, Bx[R0][e + jj], c1[10 + jj]);
                        if (!SAME_ACT) c2[10 + jj] = fma(qa, Bq[R0][e + jj], c2[10 + jj]);
                    }
#pragma unroll
                    for (int jj = 0; jj < 5; ++jj) {                 // dy = -2, -1, dx = jj - 2
                        c1[jj] = fma(qa, Bx[R2][e + jj], c1[jj]);
                        c1[5 + jj] = fma(qa, Bx[R1][e + jj], c1[5 + jj]);
                        if (!SAME_ACT) { c2[jj] = fma(qa, Bq[R2][e + jj], c2[jj]); c2[5 + jj] = fma(qa, Bq[R1][e + jj], c2[5 + jj]); }
                    }
                }
            }
        };
        for (int j = 2; j < nr; j += 3) {
            step(j, std::integral_constant<int, 0>{});
            if (j + 1 < nr) step(j + 1, std::integral_constant<int, 1>{});
            if (j + 2 < nr) step(j + 2, std::integral_constant<int, 2>{});
        }
    }
    if (live) {
        if (sg >> 31) atomicOr(p.negflag + ch, 1);
        double *out = p.part + ((ch * p.halves + hsel) * p.nslots + blockIdx.x) * kShiftN;        // [channel][image group][slot][27]
#pragma unroll
        for (int i = 0; i < 13; ++i) { out[i] = c1[i]; out[13 + i] = SAME_ACT ? c1[i] : c2[i]; }
        out[26] = c3;
    }
}

template <bool SAME_ACT>
__global__ void __launch_bounds__(64)
gpfq_gram_shift_nhwc_kernel(NhwcParams p)
{
    __shared__ float ring[kNhwcDepth * (SAME_ACT ? 1 : 2) * (kNhwcStrip + 4) * 64];
    int cls = 0;
    while ((int)blockIdx.x >= p.slot_off[cls + 1]) ++cls;
    if (cls % 3 == 1) nhwc_class_walk<SAME_ACT, kNhwcStrip>(p, ring, cls);
    else nhwc_class_walk<SAME_ACT, 1>(p, ring, cls);
}

// Class sums of the NHWC form, first stage: a launch has thousands of slots (nhwc_slots), 27 doubles each per channel -- a channel's
// partials are megabytes.  Workgroup (channel, part) adds the slots of its sixteenth of the slot range, class by class, into
// tpart[channel][part][9][27].
constexpr int kNhwcParts = 16;            // at most: nhwc_parts()
__global__ void __launch_bounds__(256)
gpfq_gram_shift_partsum_nhwc_kernel(NhwcParams p, double *__restrict__ tpart)
{
    // 64 slots at a time: their 64 x 27 doubles are one contiguous piece of memory, copied to LDS with whole-line requests (a lane per
    // (slot, sum) straight from memory would fetch every 64-byte sector 27 times over); wavefront w then adds the sums i = w, w + 4, ...
    // over the 64 slots of the piece -- lane = slot -- and the pieces in order: a fixed order.  Pieces do not straddle classes.
    __shared__ double tile[64 * kShiftN];
    const int64_t ch = blockIdx.x;
    const int part = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per = (p.nslots + p.parts - 1) / p.parts;
    const int k0 = part * per, k1 = k0 + per < p.nslots ? k0 + per : p.nslots;
    double *out = tpart + (ch * kNhwcParts + part) * 9 * kShiftN;
    const double *src = p.part + ch * p.halves * p.nslots * kShiftN;
    for (int c = 0; c < 9; ++c) {
        const int lo = p.slot_off[c] > k0 ? p.slot_off[c] : k0, hi = p.slot_off[c + 1] < k1 ? p.slot_off[c + 1] : k1;
        double acc[(kShiftN + 3) / 4];
#pragma unroll
        for (int q = 0; q < (kShiftN + 3) / 4; ++q) acc[q] = 0.0;
        for (int h = 0; h < p.halves; ++h)                           // (the image groups, one after the other)
        for (int kb = lo; kb < hi; kb += 64) {
            const int cnt = hi - kb < 64 ? hi - kb : 64;
            __syncthreads();
            for (int e = threadIdx.x; e < cnt * kShiftN; e += 256) tile[e] = src[((int64_t)h * p.nslots + kb) * kShiftN + e];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < (kShiftN + 3) / 4; ++q) {
                const int i = wave + 4 * q;
                const double x = (i < kShiftN && lane < cnt) ? tile[lane * kShiftN + i] : 0.0;
                acc[q] += wave_sum(x);
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < (kShiftN + 3) / 4; ++q)
                if (wave + 4 * q < kShiftN) out[c * kShiftN + wave + 4 * q] = acc[q];
        }
    }
}

// Second stage: the parts in order -> the N = 9 Gram record of a channel + the float32 row norms (as gpfq_gram_shift_combine_kernel).
__global__ void __launch_bounds__(256)
gpfq_gram_shift_combine_nhwc_kernel(NhwcParams p, const double *__restrict__ tpart, double *__restrict__ gram, float *__restrict__ nrm32)
{
    __shared__ double T[9][kShiftN];
    const int64_t ch = blockIdx.x;
    for (int idx = threadIdx.x; idx < 9 * kShiftN; idx += (int)blockDim.x) {
        double v = 0.0;
        for (int q = 0; q < p.parts; ++q) v += tpart[(ch * kNhwcParts + q) * 9 * kShiftN + idx];
        T[idx / kShiftN][idx % kShiftN] = v;
    }
    __syncthreads();
    auto qualifies = [](int c, int t) {
        const int cy = c / 3, cx = c - 3 * cy, ky = t / 3, kx = t - 3 * ky;
        return !(cy == 0 && ky == 2) && !(cy == 2 && ky == 0) && !(cx == 0 && kx == 2) && !(cx == 2 && kx == 0);
    };
    for (int e = threadIdx.x; e < (int)kRec9; e += (int)blockDim.x) {
        double v = 0.0;
        int t = -1, s = -1, k = -1;
        if (e < 162) {
            k = e & 1; t = (e >> 1) / 9; s = (e >> 1) - 9 * t;
            if (s <= t) {
                const int dy = s / 3 - t / 3, dx = s % 3 - t % 3;
                const int j = (dy < 0 ? (dy + 2) * 5 + dx + 2 : 10 + dx + 2) + 13 * k;
                for (int c = 0; c < 9; ++c) v += qualifies(c, t) ? T[c][j] : 0.0;
            }
        } else {
            s = e - 162;
            for (int c = 0; c < 9; ++c) v += qualifies(c, s) ? T[c][26] : 0.0;
        }
        gram[ch * kRec9 + e] = v;
        if (nrm32 && k == 1 && s == t) nrm32[ch * 9 + t] = (float)sqrt(v);
    }
}

static inline size_t al256i(size_t x) { return (x + 255) & ~(size_t)255; }

static int image_strip(int64_t ow, int variant)
{
    if ((variant == 1 || variant == 2 || variant == 4) && ow % variant == 0) return variant;
    return ow % 4 == 0 ? 4 : ow % 2 == 0 ? 2 : 1;
}

// shift: the band layout of the shift form (SAME only): two zero rows above every image, none below, two zero columns on either side.
static bool image_plan(int64_t n, int64_t H, int64_t W, int pad, int variant, ImgParams *out, int *S_out, size_t *lds_out, bool shift = false)
{
    const int64_t oh = H + 2 * pad - 2, ow = W + 2 * pad - 2;
    if (oh <= 0 || ow <= 0 || n <= 0) return false;
    // div_small() is exact for dividends below 2^24 (float holds them exactly)
    if (n * (H + 2 * pad) >= (1LL << 24) || n * oh * ow >= (1LL << 30) || n * H * W >= (1LL << 30)) return false;
    if (shift && (pad != 1 || H < 4 || W < 4)) return false;          // (launch_gram_image asks for it from 20 x 20 up)
    const int colpad = shift ? 4 : 2 * pad;
    if (W + colpad > 1020) return false;                   // a staged row is at most one piece per thread
    const int S = image_strip(ow, variant);
    ImgParams p{};
    p.n = (int)n; p.H = (int)H; p.W = (int)W; p.pad = pad;
    p.rpad = shift ? 2 : pad; p.cpad = shift ? 2 : pad;
    p.plane = n * H * W;
    p.PH = (int)(H + 2 * pad); p.oh = (int)oh; p.SPR = (int)(ow / S);
    p.grows = (int)(n * oh);
    p.RB = shift ? kImgThreads / (p.SPR + 2) - 2 : kImgStripsPerBand / p.SPR;       // shift: one item per thread and band (see the kernel)
    if (shift && p.RB < 1) return false;
    if (p.RB < 1) p.RB = 1;
    if (p.RB > p.grows) p.RB = p.grows;
    p.nbands = (p.grows + p.RB - 1) / p.RB;
    p.LP = (int)((W + colpad + 3) & ~(int64_t)3);
    p.lrows = p.RB + 2 * ((p.RB - 1) / p.oh + 2);
    int lg = 0; while ((1 << lg) < p.LP / 4) ++lg;
    p.lpr_log2 = lg;
    p.inv_SPR = 1.0f / (float)p.SPR; p.inv_oh = 1.0f / (float)p.oh; p.inv_PH = 1.0f / (float)p.PH;
    size_t lds = (size_t)2 * p.lrows * p.LP * sizeof(float);
    if (lds > (size_t)kImgMaxLds) return false;
    if (shift && lds < (size_t)kImgThreads * 10 * sizeof(double)) lds = (size_t)kImgThreads * 10 * sizeof(double);   // the final class sums
    if (out) *out = p;
    if (S_out) *S_out = S;
    if (lds_out) *lds_out = lds;
    return true;
}

bool gram_image_supported(int64_t n, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int rh, int rw, int same_padding)
{
    if (kh != 3 || kw != 3 || sh != 1 || sw != 1 || rh != 1 || rw != 1) return false;
    return image_plan(n, H, W, same_padding ? 1 : 0, 0, nullptr, nullptr, nullptr);
}

size_t gram_image_workspace_bytes(int64_t nch, int64_t F)
{
    size_t b = 0;
    b += al256i((size_t)(nch + kImgBlocks) * 2 * kRec9 * sizeof(double));   // partials: nch * ceil(kImgBlocks / nch) * 2 at most
    b += al256i((size_t)nch * kRec9 * sizeof(double));                      // Gram records
    b += al256i((size_t)nch * 9 * sizeof(float));                           // row norms
    b += al256i((size_t)nch * F * 9 * sizeof(float));                       // chosen values per filter and step
    b += gram_fix_bytes();                                                  // device-side repair of uncertified chains
    b += al256i((size_t)nch * sizeof(int));                                 // "channel has negative activations" flags
    return b;
}

hipError_t launch_gram_image(const ImageGramArgs &a, hipStream_t stream)
{
    if (a.nch == 0 || (a.F == 0 && a.phase != 1)) return hipSuccess;
    ImgParams p;
    int S;
    size_t lds;
    // SAME padding: the shift form (27 instead of 99 FMAs per position) wherever its band layout fits.  Measured on ResNet50's
    // 3x3 layers at 4096 images (tools/conv3x3_probe.py, kernel alone): 56 x 56: 6.8 -> 3.1 ms, 28 x 28: 2.8 -> 1.6, 14 x 14: 1.63 ->
    // 1.57, 7 x 7: 1.5 -> 1.7 (a third of the positions are on the border and the bands are short); the CIFAR10 CNN's layers at 5008
    // images (tools/conv3x3_probe.py, whole layer): 32 x 32, 32 channels: 1.93 -> 1.49 ms; 24 x 24: 1.70 -> 1.50; 20 x 20: 2.24 ->
    // 2.02; 16 x 16 loses; 3 channels at 32 x 32 (hundreds of short workgroups per channel, each with its class sums to reduce):
    // 0.34 -> 0.50.  So: from 20 x 20 up, with 8 channels or more in the shard.
    // shift_form = 2 forces it for every size it can take (tests).
    const bool shift = a.shift_form && (a.shift_form == 2 || (a.H >= 20 && a.W >= 20 && a.nch >= 8)) && image_plan(a.n, a.H, a.W, a.pad, a.variant, &p, &S, &lds, true);
    if (!shift && !image_plan(a.n, a.H, a.W, a.pad, a.variant, &p, &S, &lds)) return hipErrorInvalidValue;
    p.act_w = a.act_w; p.act_q = a.act_q; p.same_act = a.act_w == a.act_q;
    // one round of the chip: as many workgroups as are co-resident
    const void *fn = shift ? (S == 4 ? (const void *)gpfq_gram_shift_kernel<4, false> : S == 2 ? (const void *)gpfq_gram_shift_kernel<2, false>
                                                                                        : (const void *)gpfq_gram_shift_kernel<1, false>)
                           : (S == 4 ? (const void *)gpfq_gram_image_kernel<4> : S == 2 ? (const void *)gpfq_gram_image_kernel<2>
                                                                                 : (const void *)gpfq_gram_image_kernel<1>);
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kImgThreads, lds) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    int64_t resident = (int64_t)per_cu * cus;
    if (resident > kImgBlocks) resident = kImgBlocks;
    int64_t nbx = (resident + a.nch - 1) / a.nch;
    if (nbx > p.nbands) nbx = p.nbands;
    p.nbx = (int)nbx;
    char *ws = static_cast<char *>(a.workspace);
    double *part = reinterpret_cast<double *>(ws);  ws += al256i((size_t)(a.nch + kImgBlocks) * 2 * kRec9 * sizeof(double));
    double *gram = reinterpret_cast<double *>(ws);  ws += al256i((size_t)a.nch * kRec9 * sizeof(double));
    float *nrm = reinterpret_cast<float *>(ws);     ws += al256i((size_t)a.nch * 9 * sizeof(float));
    float *q32h = reinterpret_cast<float *>(ws);    ws += al256i((size_t)a.nch * a.F * 9 * sizeof(float));
    void *fixws = ws;                               ws += gram_fix_bytes();
    int *negflag = reinterpret_cast<int *>(ws);
    hipError_t e;
    if (a.phase == 2) {                                    // records formed elsewhere (and summed over the column shards)
        e = hipMemcpyAsync(negflag, a.negflags, (size_t)a.nch * sizeof(int), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        e = launch_gram_reduce(a.records, 1, 9, gram, nrm, a.nch, stream);
        if (e != hipSuccess) return e;
    } else {
        hipError_t e0 = hipMemsetAsync(negflag, 0, (size_t)a.nch * sizeof(int), stream);
        if (e0 != hipSuccess) return e0;
        p.part = part;
        p.negflag = negflag;
        const dim3 grid((unsigned)nbx, (unsigned)a.nch), block(kImgThreads);
        if (shift) {
            // partials: [nch][nbx][9 classes][27], inside the area sized for the per-output-position form's records
#define GPFQ_SHIFT(S_)                                                                                                   \
            do {                                                                                                             \
                if (p.same_act) hipLaunchKernelGGL((gpfq_gram_shift_kernel<S_, true>), grid, block, lds, stream, p);         \
                else hipLaunchKernelGGL((gpfq_gram_shift_kernel<S_, false>), grid, block, lds, stream, p);                   \
            } while (0)
            switch (S) {
            case 4:  GPFQ_SHIFT(4); break;
            case 2:  GPFQ_SHIFT(2); break;
            default: GPFQ_SHIFT(1); break;
            }
#undef GPFQ_SHIFT
            hipLaunchKernelGGL(gpfq_gram_shift_combine_kernel, dim3((unsigned)a.nch), dim3(256), 0, stream, part, (int)nbx, gram, nrm);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        } else {
        switch (S) {
        case 4:  hipLaunchKernelGGL(gpfq_gram_image_kernel<4>, grid, block, lds, stream, p); break;
        case 2:  hipLaunchKernelGGL(gpfq_gram_image_kernel<2>, grid, block, lds, stream, p); break;
        default: hipLaunchKernelGGL(gpfq_gram_image_kernel<1>, grid, block, lds, stream, p); break;
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = launch_gram_reduce(part, nbx * 2, 9, gram, nrm, a.nch, stream);
        if (e != hipSuccess) return e;
        }
        if (a.phase == 1) {
            e = hipMemcpyAsync(a.records, gram, (size_t)a.nch * kRec9 * sizeof(double), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return e;
            return hipMemcpyAsync(a.negflags, negflag, (size_t)a.nch * sizeof(int), hipMemcpyDeviceToDevice, stream);
        }
    }
    DecideBatch bs;
    bs.nch = a.nch; bs.gram_cs = kRec9; bs.nrm_cs = 9; bs.w_cs = a.F * 9; bs.out_cs = a.F * 9; bs.unc_cs = a.F; bs.hist_cs = a.F * 9;
    FixSrc src{};
    src.X = a.act_w; src.Xq = a.act_q; src.ld = 0; src.planes = 1; src.plane = p.plane; src.pix = 1;
    src.n = p.n; src.H = p.H; src.W = p.W; src.oh = p.oh; src.ow = p.SPR * S;
    src.kw = 3; src.sh = src.sw = src.rh = src.rw = 1; src.pt = src.pl = p.pad;
    src.m = (int64_t)p.grows * src.ow;
    e = launch_canonical_norms(src, 9, a.nch, nrm, 9, fixws, stream);           // few channels: norms in an order fixed by the dimensions
    if (e != hipSuccess) return e;
    return launch_gram_decide(gram, nrm, a.Wt, 9, a.A, 9, a.F, a.slack, a.qidx, a.Qt, a.uncertified, q32h, bs, &src, fixws, negflag,
                              stream, a.big);
}

// ---- NHWC entry: 3 x 3, stride 1, SAME, all channels of the shard in one launch chain ----
static std::atomic<int> g_nhwc_halves{1};              // shards of <= 32 channels: lanes 32-63 walk the second half of the images (option conv_nhwc_halves)
void image_set_nhwc_halves(int on) { g_nhwc_halves.store(on ? 1 : 0, std::memory_order_relaxed); }
static std::atomic<int> g_nhwc_slots{8192};            // workgroups of a launch (option conv_nhwc_slots: experiment switch)
void image_set_nhwc_slots(int n) { g_nhwc_slots.store(n < 256 ? 256 : (n > 65536 ? 65536 : n), std::memory_order_relaxed); }
// Image groups of a shard of nch channels over n images: the largest G in {16, 8, 4, 2} with nch <= 64 / G that divides n (1: none)
static int nhwc_image_groups(int64_t n, int64_t nch)
{
    if (!g_nhwc_halves.load(std::memory_order_relaxed)) return 1;
    for (int g = 16; g >= 2; g >>= 1)
        if (nch * g <= 64 && n % g == 0 && n >= g) return g;
    return 1;
}

static void nhwc_slots(int64_t n, int64_t H, int64_t W, int64_t nch, NhwcParams &p)
{
    // Slots per class in proportion to its work (items x row steps per item x requests per step), about 8192 one-wavefront
    // workgroups in the launch -- four times what the chip holds at once: a workgroup belongs to one class and the classes' costs
    // per step are only estimated below, so many short workgroups, picked up by the CUs as they come free, balance what two long
    // rounds do not (kernel at 64 ch @56x56 / 128 @28x28 / 256 @14x14 / 512 @7x7, 4096 images: 3.45 / 2.22 / 1.36 / 0.93 ms with 2048
    // slots, 2.36 / 1.27 / 0.84 / 0.65 with 4096, 1.93 / 1.08 / 0.69 / 0.49 with 8192, 2.14 / 1.06 / 0.65 / 0.43 with 16384); the
    // partial sums -- 27 doubles per slot and channel, 113 MB per launch -- are what more slots cost (two-stage combine below)
    const int64_t groups = (nch + 63) / 64;
    int64_t total = g_nhwc_slots.load(std::memory_order_relaxed) / groups;
    if (total < 256) total = 256;
    const int64_t ns_mid = (W - 2 + kNhwcStrip - 1) / kNhwcStrip;
    double work[9];
    int64_t items[9];
    double wsum = 0.0;
    for (int c = 0; c < 9; ++c) {
        const int cy = c / 3, cx = c % 3;
        const int64_t rows = cy == 1 ? H - 2 : 1, strips = cx == 1 ? ns_mid : 1;
        items[c] = n * strips;
        // row steps of an item (rows above the image are not walked) x the columns a step requests
        const int64_t steps = cy == 0 ? 1 : (cy == 1 ? rows + 1 : 3);
        work[c] = (double)items[c] * (double)steps * (cx == 1 ? kNhwcStrip + 4 : 7);   // (a border-column step: five of the nine requests, a fifth of the sums, the same waits)
        wsum += work[c];
    }
    p.slot_off[0] = 0;
    for (int c = 0; c < 9; ++c) {
        int64_t nb = (int64_t)((double)total * work[c] / wsum + 0.5);
        if (nb < 1) nb = 1;
        if (nb > items[c]) nb = items[c];
        p.slot_off[c + 1] = p.slot_off[c] + (int)nb;
    }
    p.nslots = p.slot_off[9];
}

bool gram_image_nhwc_supported(int64_t n, int64_t H, int64_t W, int64_t nch)
{
    // (64 channels fill the lanes of a wavefront; with 32 -- half of them idle -- this form still beats planes + the LDS-staged kernel since
    //  round 3: the CIFAR10 CNN's 32 -> 32 @32x32 layer on 5008 images 1.55 -> 0.95 ms, 32 -> 64 @16x16 0.66 -> 0.46)
    // narrower shards: only where the image groups put their lanes to work (an image count the group count divides)
    // (and from 8 channels up: 3 -> 32 @32x32 on 5008 images takes 0.45 ms here against 0.34 through planes -- sixteen groups' partials and
    //  4-byte requests 12 bytes apart --, 8 -> 8 @56x56 0.74 against 0.96, 16 -> 32 @32x32 0.56 against 0.89)
    if (nch < 32 && (nch < 8 || nhwc_image_groups(n, nch) * nch < 32)) return false;
    return n > 0 && H >= 4 && W >= 4 && nch >= 1 && n * H * W < (1LL << 30) && H < 32768 && W < 32768;
}

size_t gram_image_nhwc_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch, int64_t F)
{
    NhwcParams p{};
    nhwc_slots(n, H, W, nch, p);
    size_t b = 0;
    b += al256i((size_t)nch * p.nslots * (nch <= 4 ? 16 : nch <= 8 ? 8 : nch <= 16 ? 4 : nch <= 32 ? 2 : 1) * kShiftN * sizeof(double));
    b += al256i((size_t)nch * kRec9 * sizeof(double));
    b += al256i((size_t)nch * 9 * sizeof(float));
    b += al256i((size_t)nch * F * 9 * sizeof(float));
    b += gram_fix_bytes();
    b += al256i((size_t)nch * sizeof(int));
    b += al256i((size_t)nch * kNhwcParts * 9 * kShiftN * sizeof(double));     // the first stage's part sums
    return b;
}

hipError_t launch_gram_image_nhwc(const ImageGramArgs &a, hipStream_t stream)
{
    if (a.nch == 0 || a.F == 0) return hipSuccess;
    if (!gram_image_nhwc_supported(a.n, a.H, a.W, a.nch) || a.pad != 1 || a.nhwc_cin < a.nch) return hipErrorInvalidValue;
    NhwcParams p{};
    p.act_w = a.act_w; p.act_q = a.act_q; p.cin = a.nhwc_cin;
    p.n = (int)a.n; p.H = (int)a.H; p.W = (int)a.W; p.nch = (int)a.nch;
    // narrow shards: the lanes a shard of at most 32 / 16 / 8 / 4 channels leaves idle walk further G-ths of the images
    p.halves = nhwc_image_groups(a.n, a.nch);
    if ((uint64_t)a.n * a.H * a.W * a.nhwc_cin * sizeof(float) >= (1ull << 32)) p.halves = 1;      // (the group offset is a 32-bit lane offset)
    p.n_walk = (int)(a.n / p.halves);
    p.half_off = p.halves > 1 ? (unsigned)((uint64_t)p.n_walk * a.H * a.W * a.nhwc_cin * sizeof(float)) : 0u;
    nhwc_slots(p.n_walk, a.H, a.W, a.nch, p);
    char *ws = static_cast<char *>(a.workspace);
    p.part = reinterpret_cast<double *>(ws);        ws += al256i((size_t)a.nch * p.nslots * (a.nch <= 4 ? 16 : a.nch <= 8 ? 8 : a.nch <= 16 ? 4 : a.nch <= 32 ? 2 : 1) * kShiftN * sizeof(double));
    double *gram = reinterpret_cast<double *>(ws);  ws += al256i((size_t)a.nch * kRec9 * sizeof(double));
    float *nrm = reinterpret_cast<float *>(ws);     ws += al256i((size_t)a.nch * 9 * sizeof(float));
    float *q32h = reinterpret_cast<float *>(ws);    ws += al256i((size_t)a.nch * a.F * 9 * sizeof(float));
    void *fixws = ws;                               ws += gram_fix_bytes();
    int *negflag = reinterpret_cast<int *>(ws);     ws += al256i((size_t)a.nch * sizeof(int));
    double *tpart = reinterpret_cast<double *>(ws);
    p.negflag = negflag;
    hipError_t e = hipMemsetAsync(negflag, 0, (size_t)a.nch * sizeof(int), stream);
    if (e != hipSuccess) return e;
    const dim3 grid((unsigned)p.nslots, (unsigned)((a.nch + 63) / 64));
    if (a.act_w == a.act_q) hipLaunchKernelGGL((gpfq_gram_shift_nhwc_kernel<true>), grid, dim3(64), 0, stream, p);
    else hipLaunchKernelGGL((gpfq_gram_shift_nhwc_kernel<false>), grid, dim3(64), 0, stream, p);
    p.parts = p.nslots / 256 < 1 ? 1 : (p.nslots / 256 > kNhwcParts ? kNhwcParts : p.nslots / 256);
    hipLaunchKernelGGL(gpfq_gram_shift_partsum_nhwc_kernel, dim3((unsigned)a.nch, (unsigned)p.parts), dim3(256), 0, stream, p, tpart);
    hipLaunchKernelGGL(gpfq_gram_shift_combine_nhwc_kernel, dim3((unsigned)a.nch), dim3(256), 0, stream, p, tpart, gram, nrm);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    DecideBatch bs;
    bs.nch = a.nch; bs.gram_cs = kRec9; bs.nrm_cs = 9; bs.w_cs = a.F * 9; bs.out_cs = a.F * 9; bs.unc_cs = a.F; bs.hist_cs = a.F * 9;
    FixSrc src{};
    src.X = a.act_w; src.Xq = a.act_q; src.ld = 0; src.planes = 1; src.plane = 1; src.pix = a.nhwc_cin;   // channel stride 1, pixel stride Cin
    src.n = p.n; src.H = p.H; src.W = p.W; src.oh = p.H; src.ow = p.W;
    src.kw = 3; src.sh = src.sw = src.rh = src.rw = 1; src.pt = src.pl = 1;
    src.m = (int64_t)a.n * a.H * a.W;
    e = launch_canonical_norms(src, 9, a.nch, nrm, 9, fixws, stream);           // few channels: norms in an order fixed by the dimensions
    if (e != hipSuccess) return e;
    return launch_gram_decide(gram, nrm, a.Wt, 9, a.A, 9, a.F, a.slack, a.qidx, a.Qt, a.uncertified, q32h, bs, &src, fixws, negflag,
                              stream, a.big);
}

}  // namespace gpfq
