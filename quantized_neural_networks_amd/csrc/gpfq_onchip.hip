// On-chip GPFQ kernel: one wavefront per neuron, the float64 residual u lives in VGPRs for the
// whole N-step walk, activation rows are staged through LDS once per workgroup and shared by
// all its neurons.
//
// Replaces _quantize_neuron_parallel / _quantize_filter2D_parallel_jit
// (scripts/quantized_network.py:91-121, :185-233) for m <= 64*EPL <= 2048.
//
// Work decomposition (cfg2: N = C = 4096, m = 1024):
//   grid  = C / NW workgroups, NW wavefronts (= neurons) each; 256 workgroups of 16 waves fill the
//           256 CUs with 4 waves per SIMD, so every neuron of the layer is resident at once and the
//           kernel is one pass of N sequential steps.
//   lane l of a wave owns elements {256c + 4l + e} of u (EPL = m/64 float64 values = 2*EPL VGPRs).
//   per tile of TS steps the workgroup copies rows X[t0..t0+TS), Xq[t0..t0+TS) (2*TS*m*4 B) from
//   global/L2 into LDS with 16-B loads; each wave then reads its 16-B slices with ds_read_b128.
//   HBM sees each row once per XCD; the kernel is bound by FP64-rate VALU issue and the per-step
//   wave-uniform work (reduction, decision), not by HBM.
//
// Two arithmetic modes, identical results:
//   MODE_EXACT      the reference's flow verbatim: <Xq_t, u> and <Xq_t, u + f32(w*X_t)> are both
//                   accumulated element by element (:86, :89).
//   MODE_CERTIFIED  only <Xq_t, u> is accumulated.  The second dot product is predicted as
//                   <Xq_t, u> + w * G_t with G_t = <Xq_t, X_t> from the pre-pass; it differs from the
//                   exact one only by the float32 rounding of the products w*X_ti, which is bounded by
//                   2^-24 |w| A_t, A_t = sum_i |Xq_ti X_ti|.  If the predicted quotient is farther from
//                   every decision boundary of the alphabet than that bound (plus float64 slack), the
//                   decision is provably the exact flow's; otherwise (about once per 10^7 weights) the
//                   wave recomputes the exact dot product.  The residual update is always the exact
//                   element-wise flow, so u is bit-identical in both modes.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

template <int EPL>
struct Lanes {
    static constexpr int VW  = EPL >= 4 ? 4 : EPL;   // contiguous elements per lane per chunk
    static constexpr int NCH = EPL / VW;             // chunks of 64*VW elements
    static constexpr int MP  = 64 * EPL;             // padded row length held by one wave
    __device__ static __forceinline__ int elem(int lane, int c, int e) { return 64 * VW * c + VW * lane + e; }
};

template <int EPL>
__device__ __forceinline__ void lds_read_row(const float *row, int lane, float (&dst)[EPL])
{
    using L = Lanes<EPL>;
#pragma unroll
    for (int c = 0; c < L::NCH; ++c) {
        const float *p = row + L::elem(lane, c, 0);
        if constexpr (L::VW == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            dst[4 * c + 0] = v.x; dst[4 * c + 1] = v.y; dst[4 * c + 2] = v.z; dst[4 * c + 3] = v.w;
        } else if constexpr (L::VW == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(p);
            dst[2 * c + 0] = v.x; dst[2 * c + 1] = v.y;
        } else {
            dst[c] = *p;
        }
    }
}

// Copy rows [t0, t0+ts) of a [N][ld] f32 matrix into LDS rows of pitch MP, zero-filling i >= m.
template <int MP>
__device__ __forceinline__ void stage_rows(const float *__restrict__ G, int64_t ld, int64_t t0, int ts,
                                           int64_t N, int m, bool vec4, float *lds, int tid, int nthreads)
{
    if (vec4) {
        constexpr int Q = MP / 4;
        for (int idx = tid; idx < ts * Q; idx += nthreads) {
            const int s = idx / Q, i4 = (idx - s * Q) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t0 + s < N && i4 < m) v = *reinterpret_cast<const float4 *>(G + (t0 + s) * ld + i4);
            *reinterpret_cast<float4 *>(lds + s * MP + i4) = v;
        }
    } else {
        for (int idx = tid; idx < ts * MP; idx += nthreads) {
            const int s = idx / MP, i = idx - s * MP;
            float v = 0.f;
            if (t0 + s < N && i < m) v = G[(t0 + s) * ld + i];
            lds[s * MP + i] = v;
        }
    }
}

enum { MODE_EXACT = 0, MODE_CERTIFIED = 1 };

// EPL = 32 needs ~200 VGPRs: cap the workgroup at 8 waves so the allocator may use 256.
// AR = alphabet registers per lane: 1 (up to 64 members, int8 indices) or 4 (up to 256 members, int16 indices).
template <int EPL, int MODE, int AR>
__global__ void __launch_bounds__(EPL >= 32 ? 512 : 1024)
gpfq_onchip_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld,
                   const float *__restrict__ nrm32, const RowStats *__restrict__ stats,
                   const float *__restrict__ Wt, int64_t ldw,
                   AlphabetT<64 * AR> A, int64_t N, int m, int64_t C, int TS, int vec4,
                   typename IndexOf<AR>::type *__restrict__ qidx, float *__restrict__ Qt,
                   double *__restrict__ resid, double *__restrict__ u_out,
                   unsigned long long *__restrict__ fallback_count)
{
    using L = Lanes<EPL>;
    using Idx = typename IndexOf<AR>::type;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsX  = lds;                  // [TS][MP]
    float *ldsXq = lds + TS * L::MP;     // [TS][MP]

    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t j = (int64_t)blockIdx.x * (nthreads >> 6) + wave;   // this wave's neuron
    const bool active = j < C;

    const AlphaLanes<AR> a_lane = alpha_lanes<AR>(A, lane);
    const bool ascending = A.ascending != 0;

    double u[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) u[e] = 0.0;   // zeros(m), :115

    int   my_idx = 0;      // lane (t & 63) keeps step t's outputs until the 64-step flush
    float my_q   = 0.f;
    unsigned n_fallback = 0;

    // Per-step wave-uniform operands (w_t, ||Xq_t||, row statistics) travel through the scalar
    // cache and are fetched one step ahead, so they cost SGPRs instead of VGPRs and their latency
    // hides behind the previous step.
    const int64_t jw = active ? j : 0;
    const float *__restrict__ wrow = Wt + jw * ldw;
    float w_next = 0.f, nrm_next = 0.f;
    RowStats st_next = {0.0, 0.0, 0.0, 0.0};
    if (N > 0) {
        w_next = wrow[0];
        nrm_next = nrm32[0];
        if (MODE == MODE_CERTIFIED) st_next = stats[0];
    }

    for (int64_t t0 = 0; t0 < N; t0 += TS) {
        __syncthreads();   // previous tile fully consumed
        stage_rows<L::MP>(X,  ld, t0, TS, N, m, vec4, ldsX,  tid, nthreads);
        stage_rows<L::MP>(Xq, ld, t0, TS, N, m, vec4, ldsXq, tid, nthreads);
        __syncthreads();
        if (!active) continue;

        const int ts = (int)((N - t0) < TS ? (N - t0) : TS);
        for (int s = 0; s < ts; ++s) {
            const int64_t t = t0 + s;
            const float w = w_next, nrm = nrm_next;
            const RowStats st = st_next;
            if (t + 1 < N) {
                w_next = wrow[t + 1];
                nrm_next = nrm32[t + 1];
                if (MODE == MODE_CERTIFIED) st_next = stats[t + 1];
            }

            float x[EPL], xq[EPL];
            lds_read_row<EPL>(ldsXq + s * L::MP, lane, xq);

            Decision dec;
            if (MODE == MODE_EXACT) {
                lds_read_row<EPL>(ldsX + s * L::MP, lane, x);
                // <Xq_t, u> (:86) and <Xq_t, u + w*X_t> (:89); two accumulators per sum break the
                // dependent-FMA chain.
                double d0a = 0.0, d0b = 0.0, d1a = 0.0, d1b = 0.0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float p = __fmul_rn(w, x[e]);            // f32 product
                    const double xd = (double)xq[e];
                    const double v  = u[e] + (double)p;            // f64 add
                    if (e & 1) { d0b = fma(xd, u[e], d0b); d1b = fma(xd, v, d1b); }
                    else       { d0a = fma(xd, u[e], d0a); d1a = fma(xd, v, d1a); }
                }
                double dot_u, dot_uw;
                wave_sum2(d0a + d0b, d1a + d1b, dot_u, dot_uw);
                dec = decide<AR>(w, nrm, dot_u, dot_uw, a_lane, A.M, A.zero_idx, ascending);
            } else {
                double d0a = 0.0, d0b = 0.0;
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e & 1) d0b = fma((double)xq[e], u[e], d0b);
                    else       d0a = fma((double)xq[e], u[e], d0a);
                }
                // X_t is only needed by the update (and the rare exact fallback): issue its LDS
                // reads now so they complete under the reduction and the decision
                lds_read_row<EPL>(ldsX + s * L::MP, lane, x);
                const double dot_u = wave_sum(d0a + d0b);
                if ((double)nrm < 1e-16) {                                     // :83-84
                    dec.idx = A.zero_idx; dec.q = 0.0;
                } else {
                    if (fabs(dot_u) < 1e-10) {                                 // :86-87
                        dec.idx = nearest<AR>((double)w, a_lane, A.M, ascending);
                    } else {
                        const double wd = (double)w;
                        const double wg = wd * st.G;
                        const double tq = (dot_u + wg) * st.rden;              // predicted quotient (:89)
                        // modelling error of the prediction (in quotient units) + float64 slack
                        const double delta = fabs(wd) * st.cbound + st.cabs + 0x1p-44 * (fabs(dot_u) + fabs(wg)) * st.rden;
                        double margin;
                        dec.idx = nearest_margin<AR>(tq, a_lane, A.M, ascending, margin);
                        if (!(margin > delta)) {                               // rare: exact :89
                            ++n_fallback;
                            double a = 0.0, b = 0.0;
#pragma unroll
                            for (int e = 0; e < EPL; ++e) {
                                const double v = u[e] + (double)__fmul_rn(w, x[e]);
                                if (e & 1) b = fma((double)xq[e], v, b);
                                else       a = fma((double)xq[e], v, a);
                            }
                            const double te = wave_sum(a + b) / ((double)nrm * (double)nrm);
                            dec.idx = nearest<AR>(te, a_lane, A.M, ascending);
                        }
                    }
                    dec.q = alpha_get<AR>(a_lane.v, dec.idx);
                }
            }

            // u += w*X_t - q*Xq_t  (:119): f32 products, f32 subtraction, f64 accumulate
            const float q32 = (float)dec.q;
            if (q32 == 0.0f) {
                // f32(0 * xq) = +-0 and p - (+-0) = p: the increment is the product itself
#pragma unroll
                for (int e = 0; e < EPL; ++e) u[e] += (double)__fmul_rn(w, x[e]);
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    const float p = __fmul_rn(w, x[e]);
                    const float r = __fmul_rn(q32, xq[e]);
                    u[e] += (double)__fsub_rn(p, r);
                }
            }

            if (lane == (int)(t & 63)) { my_idx = dec.idx; my_q = q32; }
            if (((t + 1) & 63) == 0 || t + 1 == N) {
                const int64_t base = t & ~(int64_t)63;
                if (lane <= (int)(t & 63)) {
                    if (qidx) qidx[j * N + base + lane] = (Idx)my_idx;
                    if (Qt)   Qt[j * N + base + lane]   = my_q;
                }
            }
        }
    }

    if (!active) return;
    if (MODE == MODE_CERTIFIED && fallback_count && n_fallback && lane == 0)
        atomicAdd(fallback_count, (unsigned long long)n_fallback);
    if (resid) {
        double ss = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ss = fma(u[e], u[e], ss);
        ss = wave_sum(ss);
        if (lane == 0) resid[j] = sqrt(ss);
    }
    if (u_out) {
#pragma unroll
        for (int c = 0; c < L::NCH; ++c)
#pragma unroll
            for (int e = 0; e < L::VW; ++e) {
                const int i = L::elem(lane, c, e);
                if (i < m) u_out[j * (int64_t)m + i] = u[c * L::VW + e];
            }
    }
}

template <int EPL, int MODE, int AR>
static hipError_t launch_epl_ar(const OnchipArgs &a, const AlphabetT<64 * AR> &A, hipStream_t stream)
{
    constexpr int MP = 64 * EPL;
    // neurons (waves) per workgroup: enough workgroups to cover the 256 CUs, at most 16 waves
    int nw = 16;
    if (EPL >= 32) nw = 8;                       // ~200 VGPRs -> 2 waves/SIMD
    while (nw > 1 && (a.C + nw - 1) / nw < 256) nw >>= 1;
    // tile: as many steps as fit ~64 KiB of LDS for both matrices, a divisor of 64
    int ts = 64;
    while (ts > 1 && (size_t)2 * ts * MP * sizeof(float) > 64 * 1024) ts >>= 1;
    if (a.ts_override > 0) ts = a.ts_override;
    if (a.nw_override > 0) nw = a.nw_override;
    const size_t lds_bytes = (size_t)2 * ts * MP * sizeof(float);
    const bool vec4 = (a.ld % 4 == 0) && (a.m % 4 == 0) && ((uintptr_t)a.X % 16 == 0) && ((uintptr_t)a.Xq % 16 == 0);
    const unsigned grid = (unsigned)((a.C + nw - 1) / nw);
    hipError_t e = ensure_dynamic_lds((const void *)gpfq_onchip_kernel<EPL, MODE, AR>, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gpfq_onchip_kernel<EPL, MODE, AR>), dim3(grid), dim3(nw * 64), lds_bytes, stream,
                       a.X, a.Xq, a.ld, a.nrm32, a.stats, a.Wt, a.ldw, A, a.N, (int)a.m, a.C, ts, vec4 ? 1 : 0,
                       reinterpret_cast<typename IndexOf<AR>::type *>(a.qidx), a.Qt, a.resid, a.u_out, a.fallback_count);
    return hipGetLastError();
}

template <int EPL, int MODE>
static hipError_t launch_epl(const OnchipArgs &a, hipStream_t stream)
{
    if (a.big) return launch_epl_ar<EPL, MODE, 4>(a, *a.big, stream);      // 65..256 members: int16 indices
    return launch_epl_ar<EPL, MODE, 1>(a, a.A, stream);
}

template <int MODE>
static hipError_t launch_mode(const OnchipArgs &a, hipStream_t stream)
{
    if (a.m <= 64)   return launch_epl<1, MODE>(a, stream);
    if (a.m <= 128)  return launch_epl<2, MODE>(a, stream);
    if (a.m <= 256)  return launch_epl<4, MODE>(a, stream);
    if (a.m <= 512)  return launch_epl<8, MODE>(a, stream);
    if (a.m <= 1024) return launch_epl<16, MODE>(a, stream);
    return launch_epl<32, MODE>(a, stream);
}

hipError_t launch_onchip(const OnchipArgs &a, hipStream_t stream)
{
    // rows too long for one wavefront (or a forced split): one neuron over several wavefronts
    // ... or a layer too narrow to fill the chip with one or two neurons per wavefront: its steps are
    // latency-bound, and splitting a neuron over 2-4 wavefronts shortens them (round-1 sweep:
    // C = 512, m = 1024: 1.41 -> 1.11 ms per 1024 steps; m = 2048: 2.57 -> 1.45 ms)
    const bool narrow = a.wpn == 0 && a.lpn == 0 && a.m >= 512 && a.C <= 1024;
    if (a.wpn > 1 || a.m > 2048 || narrow) {
        int W = a.wpn > 1 ? a.wpn : (int)((a.m + 1023) / 1024);
        if (narrow && a.m <= 2048) {
            // 4 (8 above 512 neurons) elements per lane, 2..4 wavefronts: multiples of 4 elements per lane keep the
            // register-prefetch mode of the wide kernel (round-1 sweep: m = 512, C = 128: 0.92 -> 0.77 us/step)
            W = (int)(a.m / (64 * (a.C > 512 ? 8 : 4)));
            W = W < 2 ? 2 : W > 4 ? 4 : W;
        }
        if (a.wpn == 0 && a.m > 2048 && a.m <= 8192 && a.C <= 512) {
            // few neurons on long rows leave most of the chip idle and a step is bound by the instructions each wavefront
            // spends around its sweep: 8 elements per lane over 8..16 wavefronts instead of 16 over half as many
            // (tools/wide_probe.py, one deciding wavefront per step: 2048 x 128, m = 5008: 3.29 -> 2.76 ms; m = 3000: 2.79 -> 2.32)
            W = (int)((a.m + 511) / 512);
            W = W < 8 ? 8 : W;
        }
        while (W < 16 && (a.m + 64 * (int64_t)W - 1) / (64 * (int64_t)W) > 16) ++W;
        if (W > 16) W = 16;                          // rows beyond 16384: the long-row form, up to 28 elements per lane
        note_dense_kernel("gpfq_wide_kernel (one neuron over several wavefronts)");
        return launch_wide(a, W, stream);
    }
    if (a.mode == MODE_CERTIFIED && a.stats) {
        int lpn = a.big ? 1 : a.lpn;         // 65..256 members: the wavefront-per-neuron kernel (4 alphabet registers per lane)
        if (lpn == 0) lpn = 32;                      // measured best on cfg2/cfg3-like layers (round-1 sweep over shapes)
        while (lpn >= 16 && lpn <= 64 && !rows_supported(a, lpn)) lpn *= 2;
        if (lpn >= 16 && lpn <= 64) {
            note_dense_kernel(lpn == 16 ? "gpfq_rows_kernel<16> (row-group kernel, 4 neurons per wavefront)"
                                        : lpn == 32 ? "gpfq_rows_kernel<32> (row-group kernel, 2 neurons per wavefront)"
                                                    : "gpfq_rows_kernel<64> (row-group kernel, 1 neuron per wavefront)");
            return launch_rows(a, lpn, stream);
        }
        note_dense_kernel("gpfq_onchip_kernel<certified> (wavefront per neuron)");
        return launch_mode<MODE_CERTIFIED>(a, stream);
    }
    note_dense_kernel("gpfq_onchip_kernel<exact> (wavefront per neuron, verbatim flow)");
    return launch_mode<MODE_EXACT>(a, stream);
}

}  // namespace gpfq
