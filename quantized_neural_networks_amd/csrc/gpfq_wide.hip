// "Wide" on-chip GPFQ kernel: one neuron spread over W wavefronts of a workgroup.
//
// Same recurrence and numerics as gpfq_onchip_kernel<EPL, MODE_EXACT> (the reference's verbatim
// flow, scripts/quantized_network.py:59-121), for the two situations the one-wave-per-neuron layouts
// handle badly:
//   * rows longer than one wavefront's registers (2048 < m <= 16384): the residual still never
//     leaves the chip, instead of streaming u through HBM with two launches per step;
//   * narrow layers / small shards (C of a few hundred): the element sweeps of one step are split
//     over W waves, so the per-step latency -- which is all that matters when the chip is not full --
//     shrinks.
// Each wave owns 64*EPL consecutive elements of u, forms its share of the two dot products, and the
// W partial pairs meet in LDS with ONE workgroup barrier per step (slots alternate by step parity);
// every wave then takes the (identical) decision itself and updates its own elements.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"
#include "gpfq_roles.hpp"

namespace gpfq {

template <int EPL>
__device__ __forceinline__ void wide_read_row(const float *row, int lane, float (&dst)[EPL])
{
    constexpr int VW = EPL >= 4 ? 4 : EPL;
#pragma unroll
    for (int c = 0; c < EPL / VW; ++c) {
        const float *p = row + 64 * VW * c + VW * lane;
        if constexpr (VW == 4) {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            dst[4 * c + 0] = v.x; dst[4 * c + 1] = v.y; dst[4 * c + 2] = v.z; dst[4 * c + 3] = v.w;
        } else if constexpr (VW == 2) {
            const float2 v = *reinterpret_cast<const float2 *>(p);
            dst[2 * c + 0] = v.x; dst[2 * c + 1] = v.y;
        } else {
            dst[c] = *p;
        }
    }
}

// AR = alphabet registers per lane: 1 (up to 64 members, int8 indices) or 4 (up to 256 members, int16 indices).
template <int EPL, int AR>
__global__ void __launch_bounds__(1024)
gpfq_wide_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld,
                 const float *__restrict__ nrm32, const float *__restrict__ Wt, int64_t ldw,
                 AlphabetT<64 * AR> A, int64_t N, int m, int64_t C, int TS, int W, int G, int vec4,
                 typename IndexOf<AR>::type *__restrict__ qidx, float *__restrict__ Qt,
                 double *__restrict__ resid, double *__restrict__ u_out)
{
    constexpr int VW = EPL >= 4 ? 4 : EPL;
    const int MP = 64 * EPL * W;                        // padded row length covered by a neuron's W waves
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *ldsX  = lds;                                  // [TS][MP]
    float *ldsXq = lds + (size_t)TS * MP;                // [TS][MP]
    double *red  = reinterpret_cast<double *>(lds + (size_t)2 * TS * MP);   // [2][G][W][2]

    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = wave / W, part = wave - g * W;         // neuron within the workgroup, slice of the row
    const int64_t j = (int64_t)blockIdx.x * G + g;
    const bool active = j < C;                            // inactive waves still walk (barriers) but write nothing
    const float *__restrict__ wrow = Wt + (active ? j : 0) * ldw;

    const AlphaLanes<AR> a_lane = alpha_lanes<AR>(A, lane);
    const bool ascending = A.ascending != 0;

    double u[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) u[e] = 0.0;            // zeros(m), :115
    int   my_idx = 0;
    float my_q   = 0.f;

    float w_next = 0.f, nrm_next = 0.f;
    if (N > 0) { w_next = wrow[0]; nrm_next = nrm32[0]; }

    const int total4 = TS * (MP / 4);
    const int rot = (int)((blockIdx.x * 331u) % (unsigned)(total4 > 0 ? total4 : 1));   // see gpfq_rows.hip

    for (int64_t t0 = 0; t0 < N; t0 += TS) {
        __syncthreads();                                  // previous tile fully consumed
        if (vec4) {
            const int Q = MP / 4;
            for (int idx0 = tid; idx0 < total4; idx0 += nthreads) {
                int idx = idx0 + rot;
                if (idx >= total4) idx -= total4;
                const int s = idx / Q, i4 = (idx - s * Q) * 4;
                float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vq = vx;
                if (t0 + s < N && i4 < m) {
                    vx = *reinterpret_cast<const float4 *>(X  + (t0 + s) * ld + i4);
                    vq = *reinterpret_cast<const float4 *>(Xq + (t0 + s) * ld + i4);
                }
                *reinterpret_cast<float4 *>(ldsX  + (size_t)s * MP + i4) = vx;
                *reinterpret_cast<float4 *>(ldsXq + (size_t)s * MP + i4) = vq;
            }
        } else {
            for (int idx = tid; idx < TS * MP; idx += nthreads) {
                const int s = idx / MP, i = idx - s * MP;
                float vx = 0.f, vq = 0.f;
                if (t0 + s < N && i < m) { vx = X[(t0 + s) * ld + i]; vq = Xq[(t0 + s) * ld + i]; }
                ldsX[idx] = vx;
                ldsXq[idx] = vq;
            }
        }
        __syncthreads();

        const int ts = (int)((N - t0) < TS ? (N - t0) : TS);
        for (int s = 0; s < ts; ++s) {
            const int64_t t = t0 + s;
            const float w = w_next, nrm = nrm_next;
            if (t + 1 < N) { w_next = wrow[t + 1]; nrm_next = nrm32[t + 1]; }

            float x[EPL], xq[EPL];
            const size_t off = (size_t)s * MP + (size_t)part * 64 * EPL;
            wide_read_row<EPL>(ldsXq + off, lane, xq);
            wide_read_row<EPL>(ldsX + off, lane, x);

            // this wave's share of <Xq_t, u> (:86) and <Xq_t, u + f32(w*X_t)> (:89)
            double d0a = 0.0, d0b = 0.0, d1a = 0.0, d1b = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const float p = __fmul_rn(w, x[e]);
                const double xd = (double)xq[e];
                const double v  = u[e] + (double)p;
                if (e & 1) { d0b = fma(xd, u[e], d0b); d1b = fma(xd, v, d1b); }
                else       { d0a = fma(xd, u[e], d0a); d1a = fma(xd, v, d1a); }
            }
            double dot_u, dot_uw;
            wave_sum2(d0a + d0b, d1a + d1b, dot_u, dot_uw);
            // the W shares meet in LDS; slots alternate with the step parity so one barrier suffices
            double *slot = red + ((size_t)(t & 1) * G + g) * W * 2;
            if (lane == 0) { slot[2 * part] = dot_u; slot[2 * part + 1] = dot_uw; }
            slot_barrier();                                    // (LDS only: not the output stores __syncthreads() would wait for)
            dot_u = 0.0; dot_uw = 0.0;
            for (int p2 = 0; p2 < W; ++p2) { dot_u += slot[2 * p2]; dot_uw += slot[2 * p2 + 1]; }   // fixed order

            const Decision dec = decide<AR>(w, nrm, dot_u, dot_uw, a_lane, A.M, A.zero_idx, ascending);

            // u += w*X_t - q*Xq_t  (:119)
            const float q32 = (float)dec.q;
            if (q32 == 0.0f) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) u[e] += (double)__fmul_rn(w, x[e]);
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e)
                    u[e] += (double)__fsub_rn(__fmul_rn(w, x[e]), __fmul_rn(q32, xq[e]));
            }

            if (lane == (int)(t & 63)) { my_idx = dec.idx; my_q = q32; }
            if (((t + 1) & 63) == 0 || t + 1 == N) {
                const int64_t base = t & ~(int64_t)63;
                if (active && part == 0 && lane <= (int)(t & 63)) {
                    if (qidx) qidx[j * N + base + lane] = (typename IndexOf<AR>::type)my_idx;
                    if (Qt)   Qt[j * N + base + lane]   = my_q;
                }
            }
        }
    }

    if (resid) {
        double ss = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ss = fma(u[e], u[e], ss);
        ss = wave_sum(ss);
        __syncthreads();                                  // all step slots consumed
        if (lane == 0) red[(size_t)g * W + part] = ss;
        __syncthreads();
        if (active && part == 0 && lane == 0) {
            double s2 = 0.0;
            for (int p2 = 0; p2 < W; ++p2) s2 += red[(size_t)g * W + p2];
            resid[j] = sqrt(s2);
        }
    }
    if (u_out && active) {
#pragma unroll
        for (int c = 0; c < EPL / VW; ++c)
#pragma unroll
            for (int e = 0; e < VW; ++e) {
                const int i = part * 64 * EPL + 64 * VW * c + VW * lane + e;
                if (i < m) u_out[j * (int64_t)m + i] = u[c * VW + e];
            }
    }
}

// One neuron per workgroup (narrow layers: nothing to share through LDS): every wavefront reads its slice of
// the rows X_t, Xq_t straight into registers, one step ahead of their use, so no staging sits on the
// step-to-step critical path; the only LDS traffic is the W pairs of partial dot products per step.
//
// PREFETCH = false is the long-row form (16384 < m <= 28672: 36..56 elements per lane of 8 wavefronts, 256 VGPRs
// each): the Xq slice is requested whole at the top of its step and kept; the X slice streams through a ring of
// four 4-element pieces, requested four pieces ahead of their use, once for the dot products and once more (from
// L2) for the update; u never leaves the chip (the alternative streams it through HBM with two launches per
// step).  `aligned` = 0 reads element-wise (any pitch, any m).  (Holding both slices, or 4 wavefronts of 512
// registers, spills: only 256 registers are directly addressable.)
template <int EPL>
__device__ __forceinline__ void wide_fetch(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int64_t t,
                                           int base, int lane, int m, bool aligned, float (&x)[EPL], float (&xq)[EPL])
{
    static_assert(EPL % 4 == 0, "direct mode reads 16-byte pieces");
#pragma unroll
    for (int c = 0; c < EPL / 4; ++c) {
        const int i = base + 256 * c + 4 * lane;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (aligned) {                                             // m % 4 == 0: a piece is inside or outside
            if (i < m) {
                a = *reinterpret_cast<const float4 *>(X + t * ld + i);
                b = *reinterpret_cast<const float4 *>(Xq + t * ld + i);
            }
        } else {
            const float *px = X + t * ld + i, *pq = Xq + t * ld + i;
            if (i < m)     { a.x = px[0]; b.x = pq[0]; }
            if (i + 1 < m) { a.y = px[1]; b.y = pq[1]; }
            if (i + 2 < m) { a.z = px[2]; b.z = pq[2]; }
            if (i + 3 < m) { a.w = px[3]; b.w = pq[3]; }
        }
        x[4 * c] = a.x; x[4 * c + 1] = a.y; x[4 * c + 2] = a.z; x[4 * c + 3] = a.w;
        xq[4 * c] = b.x; xq[4 * c + 1] = b.y; xq[4 * c + 2] = b.z; xq[4 * c + 3] = b.w;
    }
}

template <int EPL>
__device__ __forceinline__ void wide_fetch1(const float *__restrict__ P, int64_t ld, int64_t t, int base, int lane, int m,
                                            bool aligned, float (&v)[EPL])
{
#pragma unroll
    for (int c = 0; c < EPL / 4; ++c) {
        const int i = base + 256 * c + 4 * lane;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (aligned) {
            if (i < m) a = *reinterpret_cast<const float4 *>(P + t * ld + i);
        } else {
            const float *p = P + t * ld + i;
            if (i < m)     a.x = p[0];
            if (i + 1 < m) a.y = p[1];
            if (i + 2 < m) a.z = p[2];
            if (i + 3 < m) a.w = p[3];
        }
        v[4 * c] = a.x; v[4 * c + 1] = a.y; v[4 * c + 2] = a.z; v[4 * c + 3] = a.w;
    }
}

#ifdef GPFQ_WIDE_STAMPS
#define WSTAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define WSTAMP(var) do { } while (0)
#endif

template <int EPL, bool PREFETCH, int AR>
__global__ void __launch_bounds__((!PREFETCH || EPL >= 16) ? 512 : 1024)  // 16+ elements per lane need > 128 VGPRs
gpfq_wide_direct_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld,
                        const float *__restrict__ nrm32, const float *__restrict__ Wt, int64_t ldw,
                        AlphabetT<64 * AR> A, int64_t N, int m, int64_t C, int W, int aligned,
                        typename IndexOf<AR>::type *__restrict__ qidx, float *__restrict__ Qt,
                        double *__restrict__ resid, double *__restrict__ u_out)
{
    __shared__ double red[2][16][2];                      // [step parity][wave][dot_u, dot_uw]
    __shared__ float qdec[2];                             // [step parity] the decision, from wavefront 0
    const int lane = threadIdx.x & 63;
    const int part = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j = blockIdx.x;
    const float *__restrict__ wrow = Wt + j * ldw;
    const int base = part * 64 * EPL;

    const AlphaLanes<AR> a_lane = alpha_lanes<AR>(A, lane);
    const bool ascending = A.ascending != 0;

    double u[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) u[e] = 0.0;            // zeros(m), :115
    int   my_idx = 0;
    float my_q   = 0.f;

    float w_next = 0.f, nrm_next = 0.f;
    float xn[PREFETCH ? EPL : 1], xqn[PREFETCH ? EPL : 1];
    if (N > 0) {
        w_next = wrow[0]; nrm_next = nrm32[0];
        if constexpr (PREFETCH) wide_fetch<EPL>(X, Xq, ld, 0, base, lane, m, aligned != 0, xn, xqn);
    }
    unsigned long long w0 = 0, w1 = 0, w2 = 0, w3 = 0, w4 = 0, w5 = 0, a_pf = 0, a_dot = 0, a_red = 0, a_dec = 0, a_upd = 0;
    (void)w0; (void)w1; (void)w2; (void)w3; (void)w4; (void)w5; (void)a_pf; (void)a_dot; (void)a_red; (void)a_dec; (void)a_upd;
    for (int64_t t = 0; t < N; ++t) {
        WSTAMP(w0);
        const float w = w_next, nrm = nrm_next;
        // PREFETCH: the step's row slices are already in registers.  Long-row form: the Xq slice is requested whole
        // at the top of the step and kept for both sweeps; the X slice streams through a ring of kRing 4-element
        // pieces, requested kRing pieces ahead of their use, once per sweep (the second time from L2).
        constexpr int NCH = EPL / 4, kRing = NCH < 4 ? NCH : 4;
        float x[PREFETCH ? EPL : 4 * kRing], xq[EPL];
        if constexpr (PREFETCH) {
#pragma unroll
            for (int e = 0; e < EPL; ++e) { x[e] = xn[e]; xq[e] = xqn[e]; }
        } else {
            wide_fetch1<EPL>(Xq, ld, t, base, lane, m, aligned != 0, xq);
#pragma unroll
            for (int c = 0; c < kRing; ++c)
                wide_fetch1<4>(X, ld, t, base + 256 * c, lane, m, aligned != 0, *reinterpret_cast<float (*)[4]>(&x[4 * c]));
        }
        if (t + 1 < N) {
            w_next = wrow[t + 1]; nrm_next = nrm32[t + 1];
            if constexpr (PREFETCH) wide_fetch<EPL>(X, Xq, ld, t + 1, base, lane, m, aligned != 0, xn, xqn);
        }
        WSTAMP(w1);
        // this wave's share of <Xq_t, u> (:86) and <Xq_t, u + f32(w*X_t)> (:89)
        double d0a = 0.0, d0b = 0.0, d1a = 0.0, d1b = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = 4 * c + k, r = PREFETCH ? e : 4 * (c % kRing) + k;
                const float p = __fmul_rn(w, x[r]);
                const double xd = (double)xq[e];
                const double v  = u[e] + (double)p;
                if (e & 1) { d0b = fma(xd, u[e], d0b); d1b = fma(xd, v, d1b); }
                else       { d0a = fma(xd, u[e], d0a); d1a = fma(xd, v, d1a); }
            }
            if constexpr (!PREFETCH) {
                __builtin_amdgcn_sched_barrier(0);                 // the piece is consumed: its slot can be refilled
                if (c + kRing < NCH)
                    wide_fetch1<4>(X, ld, t, base + 256 * (c + kRing), lane, m, aligned != 0,
                                   *reinterpret_cast<float (*)[4]>(&x[4 * (c % kRing)]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // The update sweep re-reads X (from L2) through a pointer the compiler cannot match with the first sweep's:
        // otherwise it merges the two reads and keeps the whole slice in registers across the decision.
        const float *X2 = X;
        if constexpr (!PREFETCH) {
            asm volatile("" : "+s"(X2));
#pragma unroll
            for (int c = 0; c < kRing; ++c)                        // its first pieces are in flight during the decision
                wide_fetch1<4>(X2, ld, t, base + 256 * c, lane, m, aligned != 0, *reinterpret_cast<float (*)[4]>(&x[4 * c]));
        }
        WSTAMP(w2);
        double dot_u, dot_uw;
        wave_sum2(d0a + d0b, d1a + d1b, dot_u, dot_uw);
        double (*slot)[2] = red[t & 1];                   // slots alternate with the step parity: one barrier per step
        if (lane == 0) { slot[part][0] = dot_u; slot[part][1] = dot_uw; }
        // s_waitcnt lgkmcnt(0) + s_barrier, NOT __syncthreads(): that also waits for vmcnt(0), i.e. for the NEXT step's rows
        // requested a few hundred cycles ago -- a full L2 / HBM round trip on the critical path of every step (in-kernel
        // stamps, Dense(2048->128) on rows of 5008 samples: 1780 of a step's 3520 cycles)
        slot_barrier();
        WSTAMP(w3);
        // From five wavefronts per neuron on (they share SIMDs) ONE of them sums the shares and decides while the others wait at a
        // second barrier, instead of all repeating the ~150 dependent instructions on the SIMDs they share (W / 4 wavefronts per
        // SIMD: the step took W / 4 times the decision).  Up to four wavefronts each has a SIMD of its own and the repeated
        // decision is cheaper than a second barrier (Dense(784->128), m = 512, two wavefronts: 0.80 vs 0.87 ms).
        float qsel = 0.f;
        if (W <= 4 || part == 0) {
            dot_u = 0.0; dot_uw = 0.0;
            for (int p2 = 0; p2 < W; ++p2) { dot_u += slot[p2][0]; dot_uw += slot[p2][1]; }   // fixed order
            const Decision dec = decide<AR>(w, nrm, dot_u, dot_uw, a_lane, A.M, A.zero_idx, ascending);
            qsel = (float)dec.q;
            if (lane == (int)(t & 63)) { my_idx = dec.idx; my_q = qsel; }
        }
        if (W > 4) {
            if (part == 0 && lane == 0) qdec[t & 1] = qsel;
            slot_barrier();
            qsel = qdec[t & 1];                           // (slots alternate with the step parity, like the shares')
        }
        // u += w*X_t - q*Xq_t  (:119)
        const float q32 = qsel;
        WSTAMP(w4);
        if constexpr (PREFETCH) {
            if (q32 == 0.0f) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) u[e] += (double)__fmul_rn(w, x[e]);
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e)
                    u[e] += (double)__fsub_rn(__fmul_rn(w, x[e]), __fmul_rn(q32, xq[e]));
            }
        } else {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {      // q = 0: f32(q*xq) = +-0 and p - (+-0) = p, the same sum
                    const int r = 4 * (c % kRing) + k;
                    u[4 * c + k] += (double)__fsub_rn(__fmul_rn(w, x[r]), __fmul_rn(q32, xq[4 * c + k]));
                }
                __builtin_amdgcn_sched_barrier(0);
                if (c + kRing < NCH)
                    wide_fetch1<4>(X2, ld, t, base + 256 * (c + kRing), lane, m, aligned != 0,
                                   *reinterpret_cast<float (*)[4]>(&x[4 * (c % kRing)]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        WSTAMP(w5);
#ifdef GPFQ_WIDE_STAMPS
        a_pf += w1 - w0; a_dot += w2 - w1; a_red += w3 - w2; a_dec += w4 - w3; a_upd += w5 - w4;
#endif
        if (((t + 1) & 63) == 0 || t + 1 == N) {
            const int64_t b0 = t & ~(int64_t)63;
            if (part == 0 && lane <= (int)(t & 63)) {
                if (qidx) qidx[j * N + b0 + lane] = (typename IndexOf<AR>::type)my_idx;
                if (Qt)   Qt[j * N + b0 + lane]   = my_q;
            }
        }
    }
#ifdef GPFQ_WIDE_STAMPS
    if (blockIdx.x == 0 && lane == 0 && part == 0 && N > 0)
        printf("wide direct kernel EPL=%d W=%d: cycles per step: prefetch issue %llu, dot %llu, reduce+barrier %llu, sum+decide %llu, update %llu\n",
               EPL, W, a_pf / N, a_dot / N, a_red / N, a_dec / N, a_upd / N);
#endif

    if (resid) {
        double ss = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ss = fma(u[e], u[e], ss);
        ss = wave_sum(ss);
        __syncthreads();                                  // all step slots consumed
        if (lane == 0) red[0][part][0] = ss;
        __syncthreads();
        if (part == 0 && lane == 0) {
            double s2 = 0.0;
            for (int p2 = 0; p2 < W; ++p2) s2 += red[0][p2][0];
            resid[j] = sqrt(s2);
        }
    }
    if (u_out) {
#pragma unroll
        for (int c = 0; c < EPL / 4; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = base + 256 * c + 4 * lane + e;
                if (i < m) u_out[j * (int64_t)m + i] = u[c * 4 + e];
            }
    }
}

template <int EPL, int AR>
static hipError_t launch_wide_epl_ar(const OnchipArgs &a, const AlphabetT<64 * AR> &A, int W, hipStream_t stream)
{
    auto *qidx = reinterpret_cast<typename IndexOf<AR>::type *>(a.qidx);
    const int MP = 64 * EPL * W;
    int G = 16 / W;                                       // neurons per workgroup (<= 16 wavefronts)
    if (G < 1) G = 1;
    while (G > 1 && (a.C + G - 1) / G < 256) G >>= 1;     // narrow layers: spread the neurons over the CUs
    if constexpr (EPL % 4 == 0) {
        // register-prefetch mode, one neuron per workgroup: faster than sharing LDS-staged rows between the
        // neurons of a workgroup at every shape measured (round-1 sweep of narrow layers), the rows come from L2 anyway
        const bool aligned = (a.ld % 4 == 0) && (a.m % 4 == 0) && ((uintptr_t)a.X % 16 == 0) && ((uintptr_t)a.Xq % 16 == 0);
        if (aligned && !(a.variant & 2) && (EPL < 16 || W <= 8)) {
            hipLaunchKernelGGL((gpfq_wide_direct_kernel<EPL, true, AR>), dim3((unsigned)a.C), dim3(64 * W), 0, stream,
                               a.X, a.Xq, a.ld, a.nrm32, a.Wt, a.ldw, A, a.N, (int)a.m, a.C, W, 1,
                               qidx, a.Qt, a.resid, a.u_out);
            return hipGetLastError();
        }
    }
    int ts = 16;
    while (ts > 1 && (size_t)2 * ts * MP * sizeof(float) > 96 * 1024) ts >>= 1;
    if (a.ts_override > 0 && (size_t)2 * a.ts_override * MP * sizeof(float) <= 150 * 1024) ts = a.ts_override;
    const size_t lds_bytes = (size_t)2 * ts * MP * sizeof(float) + (size_t)2 * G * W * 2 * sizeof(double);
    const bool vec4 = (a.ld % 4 == 0) && (a.m % 4 == 0) && ((uintptr_t)a.X % 16 == 0) && ((uintptr_t)a.Xq % 16 == 0);
    const unsigned grid = (unsigned)((a.C + G - 1) / G);
    hipError_t e = ensure_dynamic_lds((const void *)gpfq_wide_kernel<EPL, AR>, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gpfq_wide_kernel<EPL, AR>), dim3(grid), dim3(64 * W * G), lds_bytes, stream,
                       a.X, a.Xq, a.ld, a.nrm32, a.Wt, a.ldw, A, a.N, (int)a.m, a.C, ts, W, G, vec4 ? 1 : 0,
                       qidx, a.Qt, a.resid, a.u_out);
    return hipGetLastError();
}

template <int EPL>
static hipError_t launch_wide_epl(const OnchipArgs &a, int W, hipStream_t stream)
{
    if (a.big) return launch_wide_epl_ar<EPL, 4>(a, *a.big, W, stream);    // 65..256 members: int16 indices
    return launch_wide_epl_ar<EPL, 1>(a, a.A, W, stream);
}

// W = wavefronts per neuron (2..16); elements per lane follow from m.
hipError_t launch_wide(const OnchipArgs &a, int W, hipStream_t stream)
{
    if (W < 1) W = 1;
    if (W > 16) W = 16;
    const int64_t per_lane = (a.m + 64 * (int64_t)W - 1) / (64 * (int64_t)W);
    if (per_lane <= 1)  return launch_wide_epl<1>(a, W, stream);
    if (per_lane <= 2)  return launch_wide_epl<2>(a, W, stream);
    if (per_lane <= 4)  return launch_wide_epl<4>(a, W, stream);
    if (per_lane <= 8)  return launch_wide_epl<8>(a, W, stream);
    if (per_lane <= 16) return launch_wide_epl<16>(a, W, stream);
    const int64_t per_lane8 = (a.m + 511) / 512;
    if (W == 16 && per_lane8 <= 56) {                     // long rows: 8 wavefronts, rows streamed through a register ring
        const int aligned = (a.ld % 4 == 0) && (a.m % 4 == 0) && ((uintptr_t)a.X % 16 == 0) && ((uintptr_t)a.Xq % 16 == 0);
#define GPFQ_LONG(EPL_)                                                                                                   \
        do {                                                                                                              \
            if (a.big)                                                                                                    \
                hipLaunchKernelGGL((gpfq_wide_direct_kernel<EPL_, false, 4>), dim3((unsigned)a.C), dim3(512), 0, stream,  \
                                   a.X, a.Xq, a.ld, a.nrm32, a.Wt, a.ldw, *a.big, a.N, (int)a.m, a.C, 8, aligned,         \
                                   reinterpret_cast<int16_t *>(a.qidx), a.Qt, a.resid, a.u_out);                          \
            else                                                                                                          \
                hipLaunchKernelGGL((gpfq_wide_direct_kernel<EPL_, false, 1>), dim3((unsigned)a.C), dim3(512), 0, stream,  \
                                   a.X, a.Xq, a.ld, a.nrm32, a.Wt, a.ldw, a.A, a.N, (int)a.m, a.C, 8, aligned,            \
                                   a.qidx, a.Qt, a.resid, a.u_out);                                                       \
        } while (0)
        if (per_lane8 <= 36) GPFQ_LONG(36);
        else if (per_lane8 <= 40) GPFQ_LONG(40);
        else if (per_lane8 <= 44) GPFQ_LONG(44);
        else if (per_lane8 <= 48) GPFQ_LONG(48);
        else if (per_lane8 <= 52) GPFQ_LONG(52);
        else GPFQ_LONG(56);
#undef GPFQ_LONG
        return hipGetLastError();
    }
    return hipErrorInvalidValue;
}

}  // namespace gpfq
