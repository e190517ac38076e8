// Gram-matrix GPFQ path for SHORT walks over LONG rows: the conv case of
// _quantize_filter2D_parallel_jit (scripts/quantized_network.py:185-233), N = kh*kw steps (9, 25, 49)
// against patch matrices of m = n_img*oh*ow = 10^5..10^7 columns, shared by all C filters of a channel.
//
// The streaming kernel reads and writes every filter's residual u once per step (16 B per
// filter*column*step of HBM traffic).  Here the patch rows are read ONCE per channel to form the N x N
// Gram matrices
//     G1[t][s] = <Xq_t, X_s>     G2[t][s] = <Xq_t, Xq_s>     (+ A1, A2: the same with absolute values)
// and every filter then runs its N-step recurrence on scalars:
//     <Xq_t, u_{t-1}>  ~  sum_{s<t} ( w_s G1[t][s] - q_s G2[t][s] ).
// The reference forms u element-wise with three float32 roundings per step (:228) and accumulates it in
// float64, so the identity above holds only up to
//     |error| <= c * B_t,   B_t = sum_{s<t} ( |w_s| A1[t][s] + |q_s| A2[t][s] ),   c = 2^-21
// (2^-24 each for the two products and the subtraction, float64 accumulation of u and of the dot
// products, m < 2^30, and a factor ~2 of slack).  A decision is accepted only if the predicted
// quotient is farther from every decision boundary of the alphabet than that bound allows
// (and the rule-(ii) test |<Xq_t,u>| < 1e-10 is decided the same way); otherwise the filter is
// flagged `uncertified` and the caller reruns it through the exact element-wise path
// (gpfq_quantize_neurons).  Accepted decisions are therefore provably the exact flow's.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

constexpr int kGramThreads = 256;
constexpr int kGramCH = 256;               // columns staged per chunk

// Gram tiles with register accumulators.  Block (x, ty, sz) owns rows t in [t0, t0 + 4*TB), t0 = 4*TB*ty
// (wave w: TB of them) against rows s in [s0, s0 + SB), s0 = SB*sz, and walks the column chunks
// x, x + gridDim.x, ...: each chunk of 256 columns of the 4*TB + 2*SB rows it needs is staged in LDS once,
// every lane then feeds 4 columns into its TB*SB*4 float64 accumulators (products of two f32 are exact in
// f64).  Only at the end are the accumulators reduced across the wave and written as one partial per block:
//     part[x][t][s][0..3] = <Xq_t,X_s>, <Xq_t,Xq_s>, <|Xq_t|,|X_s|>, <|Xq_t|,|Xq_s|>  over the block's columns.
template <int TB, int SB>
__global__ void __launch_bounds__(kGramThreads)
gpfq_gram_tile_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int N, int64_t m,
                      int64_t nchunks, double *__restrict__ part)
{
    __shared__ __attribute__((aligned(16))) float lqt[4 * TB][kGramCH];
    __shared__ __attribute__((aligned(16))) float lxs[SB][kGramCH];
    __shared__ __attribute__((aligned(16))) float lqs[SB][kGramCH];
    const int t0 = blockIdx.y * 4 * TB, s0 = blockIdx.z * SB;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)Xq % 16 == 0);
    double acc[TB][SB][4];
#pragma unroll
    for (int a = 0; a < TB; ++a)
#pragma unroll
        for (int s = 0; s < SB; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[a][s][k] = 0.0;

    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int64_t c0 = ch * kGramCH;
        __syncthreads();
        // stage the 4*TB + 2*SB row segments as 16-byte pieces (zero beyond N / m)
        for (int idx = threadIdx.x; idx < (4 * TB + 2 * SB) * (kGramCH / 4); idx += kGramThreads) {
            const int r = idx / (kGramCH / 4), c = (idx - r * (kGramCH / 4)) * 4;
            const int64_t col = c0 + c;
            const float *src;
            float *dst;
            int row;
            if (r < 4 * TB)           { row = t0 + r;               src = Xq; dst = &lqt[r][c]; }
            else if (r < 4 * TB + SB) { row = s0 + r - 4 * TB;      src = X;  dst = &lxs[r - 4 * TB][c]; }
            else                      { row = s0 + r - 4 * TB - SB; src = Xq; dst = &lqs[r - 4 * TB - SB][c]; }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < N && col < m) {
                const float *g = src + (int64_t)row * ld + col;
                if (vec && col + 4 <= m) v = *reinterpret_cast<const float4 *>(g);
                else {
                    v.x = g[0];
                    if (col + 1 < m) v.y = g[1];
                    if (col + 2 < m) v.z = g[2];
                    if (col + 3 < m) v.w = g[3];
                }
            }
            *reinterpret_cast<float4 *>(dst) = v;
        }
        __syncthreads();
        // lane l feeds columns 4l..4l+3 of the chunk: one 16-byte LDS read per row
        float4 qt4[TB];
#pragma unroll
        for (int a = 0; a < TB; ++a) qt4[a] = *reinterpret_cast<const float4 *>(&lqt[wave * TB + a][4 * lane]);
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            const float4 xs4 = *reinterpret_cast<const float4 *>(&lxs[s][4 * lane]);
            const float4 qs4 = *reinterpret_cast<const float4 *>(&lqs[s][4 * lane]);
            const float xsv[4] = {xs4.x, xs4.y, xs4.z, xs4.w}, qsv[4] = {qs4.x, qs4.y, qs4.z, qs4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double xs = (double)xsv[e], qs = (double)qsv[e];
                const double xsa = fabs(xs), qsa = fabs(qs);
#pragma unroll
                for (int a = 0; a < TB; ++a) {
                    const float qtf = e == 0 ? qt4[a].x : e == 1 ? qt4[a].y : e == 2 ? qt4[a].z : qt4[a].w;
                    const double qt = (double)qtf, qta = fabs(qt);
                    acc[a][s][0] = fma(qt, xs, acc[a][s][0]);
                    acc[a][s][1] = fma(qt, qs, acc[a][s][1]);
                    acc[a][s][2] = fma(qta, xsa, acc[a][s][2]);
                    acc[a][s][3] = fma(qta, qsa, acc[a][s][3]);
                }
            }
        }
    }
#pragma unroll
    for (int a = 0; a < TB; ++a) {
        const int t = t0 + wave * TB + a;
#pragma unroll
        for (int s = 0; s < SB; ++s)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = wave_sum(acc[a][s][k]);
                if (lane == 0 && t < N && s0 + s < N)
                    part[(((int64_t)blockIdx.x * N + t) * N + (s0 + s)) * 4 + k] = v;
            }
    }
}

// gram[t][s][k] = sum over the partial blocks in block order (deterministic); one wavefront per entry.
__global__ void __launch_bounds__(256)
gpfq_gram_reduce_kernel(const double *__restrict__ part, int64_t nparts, int N, double *__restrict__ gram)
{
    const int64_t total = (int64_t)N * N * 4;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (e >= total) return;
    double v = 0.0;
    for (int64_t c = threadIdx.x & 63; c < nparts; c += 64) v += part[c * total + e];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) gram[e] = v;
}

// nrm32[t] = (float)sqrt(<Xq_t, Xq_t>): the float32-rounded row norm (:83, :89) from the Gram diagonal.
__global__ void gpfq_gram_norms_kernel(const double *__restrict__ gram, int N, float *__restrict__ nrm32)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N) nrm32[t] = (float)sqrt(gram[((int64_t)t * N + t) * 4 + 1]);
}

// One thread per neuron: the N-step recurrence on the Gram matrices with certified decisions.
__global__ void __launch_bounds__(64)
gpfq_gram_decide_kernel(const double *__restrict__ gram, const float *__restrict__ nrm32,
                        const float *__restrict__ Wt, int64_t ldw, AlphabetArg A, int N, int64_t C,
                        double slack, int8_t *__restrict__ qidx, float *__restrict__ Qt,
                        int32_t *__restrict__ uncertified, float *__restrict__ q32_hist)
{
    const int64_t j = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (j >= C) return;
    const float *w = Wt + j * ldw;
    float *qh = q32_hist + j * N;                       // this neuron's chosen values (f32), for s < t
    const double c = 0x1p-21 * slack;                   // slack = 1 in production; tests shrink margins with it
    bool ok = true;
    for (int t = 0; t < N; ++t) {
        int idx = A.zero_idx;
        float q32 = 0.f;
        const float nrm = nrm32[t];
        if (!((double)nrm < 1e-16)) {                                                  // not rule (i)
            double acc = 0.0, B = 0.0;
            for (int s = 0; s < t; ++s) {
                const double *g = gram + ((int64_t)t * N + s) * 4;
                const double ws = (double)w[s], qs = (double)qh[s];
                acc += ws * g[0] - qs * g[1];
                B += fabs(ws) * g[2] + fabs(qs) * g[3];
            }
            const double err0 = c * B;
            double tq;
            double delta;
            if (fabs(acc) + err0 < 1e-10) {                                            // certainly rule (ii)
                tq = (double)w[t];
                delta = 0.0;
            } else if (fabs(acc) - err0 >= 1e-10) {                                    // certainly rule (iii)
                const double *g = gram + ((int64_t)t * N + t) * 4;
                const double wt = (double)w[t];
                const double denom = (double)nrm * (double)nrm;
                tq = (acc + wt * g[0]) / denom;
                delta = (err0 + 0x1p-23 * fabs(wt) * g[2] * slack) / denom + 0x1p-44 * fabs(tq);
            } else {
                ok = false;                                                            // cannot tell (ii) from (iii)
                break;
            }
            // first minimum of |a_k - tq| and the runner-up distance
            double d1 = fabs(A.a[0] - tq), d2 = __longlong_as_double(0x7ff0000000000000LL);
            int best = 0;
            for (int k = 1; k < A.M; ++k) {
                const double d = fabs(A.a[k] - tq);
                if (d < d1) { d2 = d1; d1 = d; best = k; }
                else if (d < d2) d2 = d;
            }
            if (!(0.5 * (d2 - d1) > delta)) { ok = false; break; }                      // too close to a boundary (or NaN)
            idx = best;
            q32 = (float)A.a[best];
        }
        qh[t] = q32;
        if (qidx) qidx[j * N + t] = (int8_t)idx;
        if (Qt) Qt[j * N + t] = q32;
    }
    uncertified[j] = ok ? 0 : 1;
}

// Exact replay of the residual for known decisions: u = sum_t f32(f32(w_t X_t) - f32(q_t Xq_t)) with the
// reference's element-wise flow (:228), never stored -- only its squared norm leaves the chip.
constexpr int kReplayNG = 8;
__global__ void __launch_bounds__(256)
gpfq_replay_norm_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int N, int64_t m, int64_t C,
                        const float *__restrict__ Wt, int64_t ldw, const float *__restrict__ q32_hist,
                        int64_t nchunks, double *__restrict__ part)
{
    __shared__ double sm[4];
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int64_t j0 = (int64_t)blockIdx.y * kReplayNG;
    double u[kReplayNG][4];
#pragma unroll
    for (int g = 0; g < kReplayNG; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) u[g][e] = 0.0;
    const bool vec = (ld % 4 == 0) && ((uintptr_t)X % 16 == 0) && ((uintptr_t)Xq % 16 == 0);
    for (int t = 0; t < N; ++t) {
        float x[4] = {0.f, 0.f, 0.f, 0.f}, xq[4] = {0.f, 0.f, 0.f, 0.f};
        if (i0 < m) {
            if (vec && i0 + 4 <= m) {
                const float4 a = *reinterpret_cast<const float4 *>(X + (int64_t)t * ld + i0);
                const float4 b = *reinterpret_cast<const float4 *>(Xq + (int64_t)t * ld + i0);
                x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; xq[0] = b.x; xq[1] = b.y; xq[2] = b.z; xq[3] = b.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i0 + e < m) { x[e] = X[(int64_t)t * ld + i0 + e]; xq[e] = Xq[(int64_t)t * ld + i0 + e]; }
            }
        }
#pragma unroll
        for (int g = 0; g < kReplayNG; ++g) {
            const int64_t j = j0 + g;
            if (j < C) {
                const float w = Wt[j * ldw + t], q = q32_hist[j * N + t];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[g][e] += (double)__fsub_rn(__fmul_rn(w, x[e]), __fmul_rn(q, xq[e]));
            }
        }
    }
    for (int g = 0; g < kReplayNG; ++g) {
        const int64_t j = j0 + g;
        if (j >= C) break;
        double s = 0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e) s = fma(u[g][e], u[g][e], s);
        s = wave_sum(s);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) part[j * nchunks + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
    }
}

__global__ void __launch_bounds__(64)
gpfq_replay_finish_kernel(const double *__restrict__ part, int64_t nchunks, int64_t C, double *__restrict__ resid)
{
    const int64_t j = blockIdx.x;
    double s = 0.0;
    for (int64_t c = threadIdx.x; c < nchunks; c += 64) s += part[j * nchunks + c];
    s = wave_sum(s);
    if (threadIdx.x == 0) resid[j] = sqrt(s);
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

constexpr int kGramBlocksX = 512;          // column-chunk walkers per (t-set, s-set)

static int64_t gram_parts(int64_t m)
{
    const int64_t nchunks = (m + kGramCH - 1) / kGramCH;
    return nchunks < kGramBlocksX ? (nchunks > 0 ? nchunks : 1) : kGramBlocksX;
}

size_t gram_workspace_bytes(int64_t N, int64_t m, int64_t C)
{
    const int64_t rchunks = (m + 1023) / 1024;
    size_t b = 0;
    b += al256((size_t)gram_parts(m) * N * N * 4 * sizeof(double));   // Gram partials
    b += al256((size_t)N * N * 4 * sizeof(double) + 8);               // Gram matrices
    b += al256((size_t)C * N * sizeof(float));                        // chosen values (f32) per neuron and step
    b += al256((size_t)C * rchunks * sizeof(double));                 // replay partials
    return b;
}

hipError_t launch_gram(const GramArgs &a, hipStream_t stream)
{
    const int64_t nchunks = (a.m + kGramCH - 1) / kGramCH;
    const int64_t nparts = gram_parts(a.m);
    const int64_t rchunks = (a.m + 1023) / 1024;
    char *ws = static_cast<char *>(a.workspace);
    double *part = reinterpret_cast<double *>(ws);  ws += al256((size_t)nparts * a.N * a.N * 4 * sizeof(double));
    double *gram = reinterpret_cast<double *>(ws);  ws += al256((size_t)a.N * a.N * 4 * sizeof(double) + 8);
    float *q32h  = reinterpret_cast<float *>(ws);   ws += al256((size_t)a.C * a.N * sizeof(float));
    double *rpart = reinterpret_cast<double *>(ws);

    const int N = (int)a.N;
    if (a.m > 0 && N > 0) {
        if (N <= 9 && a.variant == 1) {
            hipLaunchKernelGGL((gpfq_gram_tile_kernel<1, 9>), dim3((unsigned)nparts, (unsigned)((N + 3) / 4), 1), dim3(kGramThreads), 0, stream,
                               a.X, a.Xq, a.ld, N, a.m, nchunks, part);
        } else if (N <= 9 && a.variant == 2) {
            hipLaunchKernelGGL((gpfq_gram_tile_kernel<2, 9>), dim3((unsigned)nparts, (unsigned)((N + 7) / 8), 1), dim3(kGramThreads), 0, stream,
                               a.X, a.Xq, a.ld, N, a.m, nchunks, part);
        } else if (N <= 9) {
            hipLaunchKernelGGL((gpfq_gram_tile_kernel<3, 9>), dim3((unsigned)nparts, 1, 1), dim3(kGramThreads), 0, stream,
                               a.X, a.Xq, a.ld, N, a.m, nchunks, part);
        } else {
            hipLaunchKernelGGL((gpfq_gram_tile_kernel<2, 12>), dim3((unsigned)nparts, (unsigned)((N + 7) / 8), (unsigned)((N + 11) / 12)),
                               dim3(kGramThreads), 0, stream, a.X, a.Xq, a.ld, N, a.m, nchunks, part);
        }
        hipLaunchKernelGGL(gpfq_gram_reduce_kernel, dim3((unsigned)((a.N * a.N * 4 + 3) / 4)), dim3(256), 0, stream,
                           part, nparts, N, gram);
    } else {
        hipError_t e = hipMemsetAsync(gram, 0, (size_t)a.N * a.N * 4 * sizeof(double) + 8, stream);
        if (e != hipSuccess) return e;
    }
    if (a.nrm32_out && N > 0)
        hipLaunchKernelGGL(gpfq_gram_norms_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, stream, gram, N, a.nrm32_out);
    hipLaunchKernelGGL(gpfq_gram_decide_kernel, dim3((unsigned)((a.C + 63) / 64)), dim3(64), 0, stream,
                       gram, a.nrm32, a.Wt, a.ldw, a.A, N, a.C, a.slack, a.qidx, a.Qt, a.uncertified, q32h);
    if (a.resid) {
        if (a.m > 0 && N > 0) {
            hipLaunchKernelGGL(gpfq_replay_norm_kernel, dim3((unsigned)rchunks, (unsigned)((a.C + kReplayNG - 1) / kReplayNG)),
                               dim3(256), 0, stream, a.X, a.Xq, a.ld, N, a.m, a.C, a.Wt, a.ldw, q32h, rchunks, rpart);
            hipLaunchKernelGGL(gpfq_replay_finish_kernel, dim3((unsigned)a.C), dim3(64), 0, stream, rpart, rchunks, a.C, a.resid);
        } else {
            hipError_t e = hipMemsetAsync(a.resid, 0, (size_t)a.C * sizeof(double), stream);
            if (e != hipSuccess) return e;
        }
    }
    return hipGetLastError();
}

}  // namespace gpfq
