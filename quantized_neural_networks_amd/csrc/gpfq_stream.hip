// Streaming GPFQ path: the float64 residual u of every neuron lives in HBM.  Used when a row of
// the activation matrices is too long for one wavefront's registers (m > 2048): the conv patch
// matrices of _quantize_filter2D_parallel_jit (scripts/quantized_network.py:185-233, m = n_img*oh*ow
// up to 10^6..10^7, N = kh*kw steps) and Dense layers calibrated on very many samples.
//
// One step t of the recurrence (:219-228) is two launches:
//   step_kernel(t)   for every (neuron, chunk of m): apply step t-1's update to u (read-modify-write)
//                    and, in the same pass, accumulate this chunk's share of <Xq_t, u> and
//                    <Xq_t, u + w_t*X_t>;  partial sums go to the workspace.
//   decide_kernel(t) one wavefront per neuron: sum the partials in a fixed order, apply
//                    _quantize_weight_parallel's three rules (:83-89), publish q_t.
// so u is read and written once per step: 16 B per (neuron, sample, step) of HBM traffic, plus the
// four activation rows, which a workgroup keeps in registers and reuses for NG neurons.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

constexpr int kStreamThreads = 256;
constexpr int kEPT   = 4;                            // elements per thread (one float4 / two double2)
constexpr int kChunk = kStreamThreads * kEPT;        // 1024 samples per workgroup
constexpr int kNG    = 8;                            // neurons sharing one load of the rows

__device__ __forceinline__ double block_sum(double v, double *sm)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += sm[k];
    return s;
}

__device__ __forceinline__ void load4(const float *__restrict__ row, int64_t i0, int64_t m, bool vec, float (&v)[kEPT])
{
    if (vec && i0 + kEPT <= m) {
        const float4 f = *reinterpret_cast<const float4 *>(row + i0);
        v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
#pragma unroll
        for (int e = 0; e < kEPT; ++e) v[e] = (i0 + e < m) ? row[i0 + e] : 0.f;
    }
}

// t in [0, N]: t == 0 has no update to apply, t == N has no dot products to form (it applies the
// last update and accumulates ||u||^2 instead).
__global__ void __launch_bounds__(kStreamThreads)
gpfq_stream_step_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld,
                        const float *__restrict__ Wt, int64_t ldw, int64_t N, int64_t m, int64_t C,
                        int64_t t, const float *__restrict__ q32_prev, double *__restrict__ u,
                        double *__restrict__ partials, int64_t nchunks, int vec)
{
    __shared__ double sm[kStreamThreads / 64];
    const int64_t chunk = blockIdx.x;
    const int64_t i0 = chunk * kChunk + (int64_t)threadIdx.x * kEPT;
    const int64_t j0 = (int64_t)blockIdx.y * kNG;
    const bool has_prev = t > 0, has_cur = t < N;

    float xp[kEPT], xqp[kEPT], xc[kEPT], xqc[kEPT];
#pragma unroll
    for (int e = 0; e < kEPT; ++e) xp[e] = xqp[e] = xc[e] = xqc[e] = 0.f;
    if (i0 < m) {
        if (has_prev) { load4(X + (t - 1) * ld, i0, m, vec, xp); load4(Xq + (t - 1) * ld, i0, m, vec, xqp); }
        if (has_cur)  { load4(X + t * ld, i0, m, vec, xc);       load4(Xq + t * ld, i0, m, vec, xqc); }
    }

    for (int g = 0; g < kNG; ++g) {
        const int64_t j = j0 + g;
        if (j >= C) break;
        double uu[kEPT];
        double *up = u + j * m + i0;
        // current residual: zero before the first update (u = zeros(m), :213)
#pragma unroll
        for (int e = 0; e < kEPT; ++e) uu[e] = 0.0;
        if (t > 1 && i0 < m) {
#pragma unroll
            for (int e = 0; e < kEPT; ++e) if (i0 + e < m) uu[e] = up[e];
        }
        if (has_prev) {
            const float wp = Wt[j * ldw + t - 1];
            const float qp = q32_prev[j];
#pragma unroll
            for (int e = 0; e < kEPT; ++e) {
                const float p = __fmul_rn(wp, xp[e]);
                const float r = __fmul_rn(qp, xqp[e]);
                uu[e] += (double)__fsub_rn(p, r);              // :228
            }
            if (i0 < m) {
#pragma unroll
                for (int e = 0; e < kEPT; ++e) if (i0 + e < m) up[e] = uu[e];
            }
        }
        double a0 = 0.0, a1 = 0.0;
        if (has_cur) {
            const float w = Wt[j * ldw + t];
#pragma unroll
            for (int e = 0; e < kEPT; ++e) {
                const double xd = (double)xqc[e];
                const double v = uu[e] + (double)__fmul_rn(w, xc[e]);
                a0 = fma(xd, uu[e], a0);                       // <Xq_t, u>          (:86)
                a1 = fma(xd, v, a1);                           // <Xq_t, u + w*X_t>  (:89)
            }
        } else {
#pragma unroll
            for (int e = 0; e < kEPT; ++e) a0 = fma(uu[e], uu[e], a0);   // ||u||^2 share
        }
        a0 = block_sum(a0, sm);
        if (has_cur) a1 = block_sum(a1, sm);
        if (threadIdx.x == 0) {
            partials[(j * nchunks + chunk) * 2 + 0] = a0;
            partials[(j * nchunks + chunk) * 2 + 1] = a1;
        }
    }
}

// AR = alphabet registers per lane: 1 (up to 64 members, int8 indices) or 4 (up to 256 members, int16 indices).
template <int AR>
__global__ void __launch_bounds__(64)
gpfq_stream_decide_kernel(const float *__restrict__ nrm32, const float *__restrict__ Wt, int64_t ldw,
                          AlphabetT<64 * AR> A, int64_t N, int64_t C, int64_t t,
                          const double *__restrict__ partials, int64_t nchunks,
                          float *__restrict__ q32_prev, typename IndexOf<AR>::type *__restrict__ qidx, float *__restrict__ Qt,
                          double *__restrict__ resid)
{
    const int64_t j = blockIdx.x;
    const int lane = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    for (int64_t c = lane; c < nchunks; c += 64) {
        s0 += partials[(j * nchunks + c) * 2 + 0];
        s1 += partials[(j * nchunks + c) * 2 + 1];
    }
    s0 = wave_sum(s0);
    if (t == N) {                                     // epilogue: residual norm
        if (lane == 0 && resid) resid[j] = sqrt(s0);
        return;
    }
    s1 = wave_sum(s1);
    const float w = Wt[j * ldw + t];
    const float nrm = nrm32[t];
    const Decision dec = decide<AR>(w, nrm, s0, s1, alpha_lanes<AR>(A, lane), A.M, A.zero_idx, A.ascending != 0);
    if (lane == 0) {
        const float q32 = (float)dec.q;
        q32_prev[j] = q32;
        if (qidx) qidx[j * N + t] = (typename IndexOf<AR>::type)dec.idx;
        if (Qt)   Qt[j * N + t] = q32;
    }
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t stream_workspace_bytes(int64_t N, int64_t m, int64_t C, bool need_u)
{
    (void)N;
    const int64_t nchunks = (m + kChunk - 1) / kChunk;
    size_t b = 0;
    if (need_u) b += align256((size_t)C * (size_t)m * sizeof(double));
    b += align256((size_t)C * (size_t)nchunks * 2 * sizeof(double));
    b += align256((size_t)C * sizeof(float));
    return b;
}

hipError_t launch_stream(const StreamArgs &a, hipStream_t stream)
{
    const int64_t nchunks = (a.m + kChunk - 1) / kChunk;
    char *ws = static_cast<char *>(a.workspace);
    double *u = a.u_out;
    if (!u) { u = reinterpret_cast<double *>(ws); ws += align256((size_t)a.C * (size_t)a.m * sizeof(double)); }
    double *partials = reinterpret_cast<double *>(ws);
    ws += align256((size_t)a.C * (size_t)nchunks * 2 * sizeof(double));
    float *q32_prev = reinterpret_cast<float *>(ws);

    const int vec = (a.ld % 4 == 0) && ((uintptr_t)a.X % 16 == 0) && ((uintptr_t)a.Xq % 16 == 0);
    const dim3 sgrid((unsigned)nchunks, (unsigned)((a.C + kNG - 1) / kNG));
    for (int64_t t = 0; t <= a.N; ++t) {
        hipLaunchKernelGGL(gpfq_stream_step_kernel, sgrid, dim3(kStreamThreads), 0, stream,
                           a.X, a.Xq, a.ld, a.Wt, a.ldw, a.N, a.m, a.C, t, q32_prev, u, partials, nchunks, vec);
        if (a.big)
            hipLaunchKernelGGL(gpfq_stream_decide_kernel<4>, dim3((unsigned)a.C), dim3(64), 0, stream,
                               a.nrm32, a.Wt, a.ldw, *a.big, a.N, a.C, t, partials, nchunks, q32_prev,
                               reinterpret_cast<int16_t *>(a.qidx), a.Qt, a.resid);
        else
            hipLaunchKernelGGL(gpfq_stream_decide_kernel<1>, dim3((unsigned)a.C), dim3(64), 0, stream,
                               a.nrm32, a.Wt, a.ldw, a.A, a.N, a.C, t, partials, nchunks, q32_prev,
                               a.qidx, a.Qt, a.resid);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    if (a.N == 0 && u) {
        // no steps: the residual is identically zero
        return hipMemsetAsync(u, 0, (size_t)a.C * (size_t)a.m * sizeof(double), stream);
    }
    return hipSuccess;
}

}  // namespace gpfq
