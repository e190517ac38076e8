// Row-group GPFQ kernel: LPN lanes per neuron (16, 32 or 64), 64/LPN neurons per wavefront, and
// the whole per-step decision done lane-parallel in VGPRs -- no SGPR round trips, no scalar
// branches on the common path.  Same contract and results as gpfq_onchip_kernel (certified mode);
// it exists because the on-chip kernel spends about half of its VALU issue on per-step work that
// is identical in all 64 lanes (the reduction, the quotient, the alphabet search).  Here that work
// is shared by 64/LPN neurons: each 16-lane DPP row owns a copy of the alphabet (M <= 16) and takes
// the decision for the neuron it belongs to, and the reductions are in-row rotations (+ one
// v_permlane16/32_swap per doubling) that leave the sum in every lane of the neuron.
//
// Replaces _quantize_neuron_parallel / _quantize_filter2D_parallel_jit
// (scripts/quantized_network.py:91-121, :185-233); numerics exactly as in gpfq_onchip.hip:
// the residual update is the reference's element-wise f32/f64 flow, the decision uses the
// certified prediction <Xq_t,u> + w*G_t and falls back to the exact dot product of :89 when the
// predicted quotient is within the error bound of a decision boundary.
//
// Layout: a workgroup owns 16 neurons (= LPN/4 wavefronts) and stages TS-step tiles of X and Xq
// (f32) plus the 16 x TS weights of its neurons in LDS.  Lane l of a neuron's LPN lanes holds
// elements {4*LPN*c + 4*l + e} of u, e < 4, c < EPL/4 (EPL = padded m / LPN float64 values).
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

constexpr int kMaxGroupNeurons = 16;   // neurons per workgroup (fewer when the layer is too narrow to fill 256 CUs)

// x + (x rotated by N lanes inside each row of 16): one level of an in-row all-reduce.
template <int ROR>
__device__ __forceinline__ double row_ror_add(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x120 + ROR, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x120 + ROR, 0xF, 0xF, true);
    return x + __hiloint2double(hi, lo);
}

// Sum over the LPN lanes of each neuron, delivered to every one of them (bitwise identical in all
// lanes: each level adds the same two operands on both sides).
template <int LPN>
__device__ __forceinline__ double group_allreduce(double x)
{
    x = row_ror_add<8>(x);
    x = row_ror_add<4>(x);
    x = row_ror_add<2>(x);
    x = row_ror_add<1>(x);
    if constexpr (LPN >= 32) {   // rows (0,1) and (2,3)
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        x = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
    if constexpr (LPN >= 64) {   // halves
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(x), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(x), false, false);
        x = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
    return x;
}

// OR over the 16 lanes of each row, delivered to all of them.
__device__ __forceinline__ unsigned row_or(unsigned x)
{
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true);
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x124, 0xF, 0xF, true);
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x122, 0xF, 0xF, true);
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x121, 0xF, 0xF, true);
    return x;
}

// Partial (per-lane) sums of <Xq_row, u>: four accumulators break the dependent-FMA chain.
template <int LPN, int EPL, bool XQD>
__device__ __forceinline__ double dot_partial(const double (&u)[EPL], const float *rowq, const double *rowd)
{
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int c = 0; c < EPL / 4; ++c) {
        if constexpr (XQD) {
            const double2 qa = *reinterpret_cast<const double2 *>(rowd + 4 * LPN * c);
            const double2 qb = *reinterpret_cast<const double2 *>(rowd + 4 * LPN * c + 2 * LPN);
            a0 = fma(qa.x, u[4 * c + 0], a0); a1 = fma(qa.y, u[4 * c + 1], a1);
            a2 = fma(qb.x, u[4 * c + 2], a2); a3 = fma(qb.y, u[4 * c + 3], a3);
        } else {
            const float4 q4 = *reinterpret_cast<const float4 *>(rowq + 4 * LPN * c);
            a0 = fma((double)q4.x, u[4 * c + 0], a0); a1 = fma((double)q4.y, u[4 * c + 1], a1);
            a2 = fma((double)q4.z, u[4 * c + 2], a2); a3 = fma((double)q4.w, u[4 * c + 3], a3);
        }
    }
    return (a0 + a1) + (a2 + a3);
}

// u += f32(w*X_t) - f32(q*Xq_t)  (:119): f32 products, f32 subtraction, f64 accumulate.  ZERO: every
// lane's q is 0, so the increment is the product itself (f32(0*xq) = +-0, p - (+-0) = p).
// (Fusing this sweep with the next step's dot product was measured slower: register pressure.  Packed float32
// math -- v_pk_mul_f32 / v_pk_add_f32 for the products and the difference -- is bit-identical and changes nothing:
// 5.44 vs 5.39 ms, the float64-rate instructions set the pace.)
template <int LPN, int EPL, bool ZERO>
__device__ __forceinline__ void update_residual(double (&u)[EPL], float w, float q32, const float *rowx, const float *rowq)
{
#pragma unroll
    for (int c = 0; c < EPL / 4; ++c) {
        const float4 x4 = *reinterpret_cast<const float4 *>(rowx + 4 * LPN * c);
        if constexpr (ZERO) {
            u[4 * c + 0] += (double)__fmul_rn(w, x4.x);
            u[4 * c + 1] += (double)__fmul_rn(w, x4.y);
            u[4 * c + 2] += (double)__fmul_rn(w, x4.z);
            u[4 * c + 3] += (double)__fmul_rn(w, x4.w);
        } else {
            const float4 q4 = *reinterpret_cast<const float4 *>(rowq + 4 * LPN * c);
            u[4 * c + 0] += (double)__fsub_rn(__fmul_rn(w, x4.x), __fmul_rn(q32, q4.x));
            u[4 * c + 1] += (double)__fsub_rn(__fmul_rn(w, x4.y), __fmul_rn(q32, q4.y));
            u[4 * c + 2] += (double)__fsub_rn(__fmul_rn(w, x4.z), __fmul_rn(q32, q4.z));
            u[4 * c + 3] += (double)__fsub_rn(__fmul_rn(w, x4.w), __fmul_rn(q32, q4.w));
        }
    }
}

// XQD: additionally stage Xq converted to float64 (conversion done once per workgroup instead of
// once per neuron) for the dot product; the f32 copy stays for the update of non-zero decisions.
template <int LPN, int EPL, bool XQD>
__global__ void __launch_bounds__(LPN * 16)     // (4 waves per SIMD at <= 128 VGPRs measured slower: 6.1 vs 5.4 ms;
                                                // 16 lanes per neuron with 32 neurons per workgroup, 2 waves per SIMD: 8.9 ms)
gpfq_rows_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld,
                 const float *__restrict__ nrm32, const RowStats *__restrict__ stats,
                 const float *__restrict__ Wt, int64_t ldw,
                 AlphabetArg A, int64_t N, int m, int64_t C, int TS, int GS, int vec4,
                 int8_t *__restrict__ qidx, float *__restrict__ Qt,
                 double *__restrict__ resid, double *__restrict__ u_out,
                 unsigned long long *__restrict__ fallback_count)
{
    constexpr int NPW = 64 / LPN;        // neurons per wavefront
    constexpr int MP  = LPN * EPL;       // padded row length
    constexpr int NCH = EPL / 4;         // 16-byte chunks per lane
    static_assert(EPL % 4 == 0, "EPL must be a multiple of 4");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float  *ldsX  = lds;                                   // [TS][MP]
    float  *ldsXq = lds + (size_t)TS * MP;                 // [TS][MP]
    double *ldsXqd = reinterpret_cast<double *>(lds + (size_t)2 * TS * MP);   // [TS][MP] f64, two planes per chunk
    float  *ldsW  = lds + (size_t)(XQD ? 4 : 2) * TS * MP; // [16][TS]
    double *ldsA  = reinterpret_cast<double *>(ldsW + kMaxGroupNeurons * TS);   // [16] alphabet (slow path)

    const int tid = threadIdx.x, nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int sub  = lane / LPN;                           // neuron within the wavefront
    const int l    = lane % LPN;                           // lane within the neuron
    const int k    = lane & 15;                            // alphabet slot within the DPP row
    const int gn   = wave * NPW + sub;                     // neuron within the workgroup
    const int64_t j = (int64_t)blockIdx.x * GS + gn;
    const bool active = j < C;

    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    const int M = A.M;
    const double a      = k < M ? A.a[k] : kNaN;
    const double a_next = k + 1 < M ? A.a[k + 1] : kInf;
    const double a_prev = (k > 0 && k <= M) ? A.a[k - 1] : -kInf;
    const bool ascending = A.ascending != 0;
    if (tid < 16) ldsA[tid] = tid < M ? A.a[tid] : kNaN;

    double u[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) u[e] = 0.0;              // zeros(m), :115

    int   my_idx = 0;
    float my_q   = 0.f;
    unsigned n_fallback = 0;

    // wave-uniform per-step operands through the scalar cache, one step ahead
    float nrm_next = 0.f;
    RowStats st_next = {0.0, 0.0, 0.0, 0.0};
    if (N > 0) { nrm_next = nrm32[0]; st_next = stats[0]; }

    const int elem0 = 4 * l;                               // + 4*LPN*c + e

    for (int64_t t0 = 0; t0 < N; t0 += TS) {
        __syncthreads();                                   // previous tile fully consumed
        // stage rows [t0, t0+TS) of X and Xq, zero-filled beyond m / N
        if (vec4) {
            constexpr int Q = MP / 4;
            // All workgroups walk the same rows at about the same time; starting each one at a
            // different 16-byte segment of the tile spreads their simultaneous requests over the L2
            // channels instead of queueing them on the channel that holds the tile's first lines
            // (staging alone: 1.17 -> 0.96 ms at cfg2).
            const int total = TS * Q;
            const int rot = (int)((blockIdx.x * 331u) % (unsigned)total);
            for (int idx0 = tid; idx0 < total; idx0 += nthreads) {
                int idx = idx0 + rot;
                if (idx >= total) idx -= total;
                const int s = idx / Q, i4 = (idx - s * Q) * 4;
                float4 vx = make_float4(0.f, 0.f, 0.f, 0.f), vq = vx;
                if (t0 + s < N && i4 < m) {
                    vx = *reinterpret_cast<const float4 *>(X  + (t0 + s) * ld + i4);
                    vq = *reinterpret_cast<const float4 *>(Xq + (t0 + s) * ld + i4);
                }
                *reinterpret_cast<float4 *>(ldsX  + s * MP + i4) = vx;
                *reinterpret_cast<float4 *>(ldsXq + s * MP + i4) = vq;
                if (XQD) {
                    // chunk c of 4*LPN elements: plane A = elements (4l, 4l+1), plane B = (4l+2, 4l+3)
                    const int cc = i4 / (4 * LPN), ll = (i4 - cc * 4 * LPN) / 4;
                    double *base = ldsXqd + (size_t)s * MP + 4 * LPN * cc + 2 * ll;
                    *reinterpret_cast<double2 *>(base)           = make_double2((double)vq.x, (double)vq.y);
                    *reinterpret_cast<double2 *>(base + 2 * LPN) = make_double2((double)vq.z, (double)vq.w);
                }
            }
        } else {
            for (int idx = tid; idx < TS * MP; idx += nthreads) {
                const int s = idx / MP, i = idx - s * MP;
                float vx = 0.f, vq = 0.f;
                if (t0 + s < N && i < m) { vx = X[(t0 + s) * ld + i]; vq = Xq[(t0 + s) * ld + i]; }
                ldsX[s * MP + i] = vx;
                ldsXq[s * MP + i] = vq;
                if (XQD) {
                    const int cc = i / (4 * LPN), r = i - cc * 4 * LPN, ll = r / 4, e = r & 3;
                    ldsXqd[(size_t)s * MP + 4 * LPN * cc + (e >> 1) * 2 * LPN + 2 * ll + (e & 1)] = (double)vq;
                }
            }
        }
        for (int idx = tid; idx < GS * TS; idx += nthreads) {
            const int n = idx / TS, s = idx - n * TS;
            const int64_t jn = (int64_t)blockIdx.x * GS + n;
            ldsW[idx] = (jn < C && t0 + s < N) ? Wt[jn * ldw + t0 + s] : 0.f;
        }
        __syncthreads();

        const int ts = (int)((N - t0) < TS ? (N - t0) : TS);
        for (int s = 0; s < ts; ++s) {
            const int64_t t = t0 + s;
            const float nrm = nrm_next;
            const RowStats st = st_next;
            if (t + 1 < N) { nrm_next = nrm32[t + 1]; st_next = stats[t + 1]; }

            const float w = ldsW[gn * TS + s];
            const float *rowq = ldsXq + s * MP + elem0;
            const float *rowx = ldsX  + s * MP + elem0;

            // ---- <Xq_t, u>  (:86) -------------------------------------------------------------
            const double dot_u = group_allreduce<LPN>(dot_partial<LPN, EPL, XQD>(u, rowq, ldsXqd + (size_t)s * MP + 2 * l));

            // ---- decision (:83-89, :57), one copy per 16-lane row ----------------------------------
            float q32 = 0.f;
            int   idx = A.zero_idx;
            if (!((double)nrm < 1e-16)) {                                        // not rule (i)
                const double wd = (double)w;
                const double wg = wd * st.G;
                const double tq = (dot_u + wg) * st.rden;                        // predicted quotient
                const bool   msq = fabs(dot_u) < 1e-10;                          // rule (ii)
                const double tt = msq ? wd : tq;
                // twice the modelling error of the prediction (quotient units) + float64 slack
                const double delta2 = 2.0 * (fabs(wd) * st.cbound + st.cabs) + 0x1p-43 * (fabs(dot_u) + fabs(wg)) * st.rden;
                // (a per-lane scan of the members for alphabets of <= 4 instead of the 16-lane search + broadcast
                //  measured slower: 5.7 vs 5.5 ms ternary -- f64 compares and 64-bit selects cost more than the DPP ops)
                const double d  = fabs(a - tt), dn = fabs(a_next - tt), dp = fabs(a_prev - tt);
                const bool c_lt = a < tt, n_lt = a_next < tt;
                const bool is_lo = c_lt && !n_lt;                                // k = p-1: last member below t
                const bool is_p0 = (k == 0) && !c_lt;                            // t at or below the whole alphabet (or NaN)
                const bool use_hi = is_lo && !(d <= dn);                         // tie -> lower index
                const int    idx_l = k + (use_hi ? 1 : 0);
                const double q_l   = use_hi ? a_next : a;
                // twice the distance of t from the boundary between the winner and the runner-up
                double m2 = (k + 1 < M) ? fabs(dn - d) : (dp - d);
                // first-index rule: a lower member at the same distance would win instead
                const bool plateau = is_lo && !use_hi && k > 0 && !(dp > d);
                const bool cert = !plateau && (msq || m2 > delta2) && ascending;
                const bool decider = is_lo || is_p0;
                unsigned w1 = decider ? __float_as_uint((float)q_l) : 0u;
                unsigned w2 = decider ? ((unsigned)idx_l | (cert ? 0u : 0x100u) | 0x200u) : 0u;
                w1 = row_or(w1);
                w2 = row_or(w2);
                q32 = __uint_as_float(w1);
                idx = (int)(w2 & 0xffu);
                const bool redo = (w2 & 0x300u) != 0x200u;                       // not certified (or no decider)
                if (__ballot(redo) != 0ull) {
                    // rare: exact <Xq_t, u + f32(w*X_t)> (:89) and a plain first-minimum scan
                    double ea = 0.0, eb = 0.0;
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const float4 q4 = *reinterpret_cast<const float4 *>(rowq + 4 * LPN * c);
                        const float4 x4 = *reinterpret_cast<const float4 *>(rowx + 4 * LPN * c);
                        ea = fma((double)q4.x, u[4 * c + 0] + (double)__fmul_rn(w, x4.x), ea);
                        eb = fma((double)q4.y, u[4 * c + 1] + (double)__fmul_rn(w, x4.y), eb);
                        ea = fma((double)q4.z, u[4 * c + 2] + (double)__fmul_rn(w, x4.z), ea);
                        eb = fma((double)q4.w, u[4 * c + 3] + (double)__fmul_rn(w, x4.w), eb);
                    }
                    const double te = group_allreduce<LPN>(ea + eb) / ((double)nrm * (double)nrm);
                    const double t2 = msq ? wd : te;
                    int bi = 0;
                    double bq = ldsA[0], bd = fabs(bq - t2);
                    for (int kk = 1; kk < M; ++kk) {
                        const double ak = ldsA[kk], dk = fabs(ak - t2);
                        if (dk < bd) { bd = dk; bi = kk; bq = ak; }
                    }
                    if (redo) { idx = bi; q32 = (float)bq; }
                    n_fallback += (redo && l == 0 && active) ? 1u : 0u;
                }
            }

            // ---- u += w*X_t - q*Xq_t  (:119): f32 products, f32 subtraction, f64 accumulate -------
            if (__ballot(q32 != 0.0f) == 0ull)                    // every neuron of the wave chose 0
                update_residual<LPN, EPL, true>(u, w, q32, rowx, rowq);
            else
                update_residual<LPN, EPL, false>(u, w, q32, rowx, rowq);

            // ---- outputs: lane (t mod LPN) of the neuron keeps step t until the LPN-step flush -------
            const int slot = (int)(t & (LPN - 1));
            if (l == slot) { my_idx = idx; my_q = q32; }
            if (slot == LPN - 1 || t + 1 == N) {
                const int64_t base = t - slot;
                if (active && l <= slot) {
                    if (qidx) qidx[j * N + base + l] = (int8_t)my_idx;
                    if (Qt)   Qt[j * N + base + l]   = my_q;
                }
            }
        }
    }

    if (fallback_count && n_fallback) atomicAdd(fallback_count, (unsigned long long)n_fallback);   // rare
    if (resid) {
        double ss = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) ss = fma(u[e], u[e], ss);
        ss = group_allreduce<LPN>(ss);
        if (active && l == 0) resid[j] = sqrt(ss);
    }
    if (u_out && active) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int i = 4 * LPN * c + elem0 + e;
                if (i < m) u_out[j * (int64_t)m + i] = u[4 * c + e];
            }
    }
}

template <int LPN, int EPL, bool XQD>
static hipError_t launch_rows_xqd(const OnchipArgs &a, hipStream_t stream)
{
    constexpr int MP = LPN * EPL;
    constexpr size_t per_elem = XQD ? 16 : 8;              // LDS bytes per staged element
    int ts = 64;
    while (ts > 1 && (size_t)ts * MP * per_elem > 128 * 1024) ts >>= 1;
    if (a.ts_override > 0) ts = a.ts_override;
    while (ts > 1 && (size_t)ts * MP * per_elem > 150 * 1024) ts >>= 1;
    const size_t lds_bytes = (size_t)ts * MP * per_elem + (size_t)kMaxGroupNeurons * ts * sizeof(float) + 16 * sizeof(double);
    // neurons per workgroup: 16 (smaller groups re-stage the same rows more often and measured slower
    // even on narrow layers; the tuning hook can still lower it)
    int gs = kMaxGroupNeurons;
    if (a.nw_override > 0 && a.nw_override <= 16 && a.nw_override >= 64 / LPN && !(a.nw_override & (a.nw_override - 1))) gs = a.nw_override;
    const bool vec4 = (a.ld % 4 == 0) && (a.m % 4 == 0) && ((uintptr_t)a.X % 16 == 0) && ((uintptr_t)a.Xq % 16 == 0);
    const unsigned grid = (unsigned)((a.C + gs - 1) / gs);
    hipError_t e = ensure_dynamic_lds((const void *)gpfq_rows_kernel<LPN, EPL, XQD>, lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((gpfq_rows_kernel<LPN, EPL, XQD>), dim3(grid), dim3(LPN * gs), lds_bytes, stream,
                       a.X, a.Xq, a.ld, a.nrm32, a.stats, a.Wt, a.ldw, a.A, a.N, (int)a.m, a.C, ts, gs, vec4 ? 1 : 0,
                       a.qidx, a.Qt, a.resid, a.u_out, a.fallback_count);
    return hipGetLastError();
}

template <int LPN, int EPL>
static hipError_t launch_rows_inst(const OnchipArgs &a, hipStream_t stream)
{
    if (a.variant & 1) return launch_rows_xqd<LPN, EPL, false>(a, stream);
    return launch_rows_xqd<LPN, EPL, true>(a, stream);
}

template <int LPN>
static hipError_t launch_rows_lpn(const OnchipArgs &a, hipStream_t stream)
{
    const int64_t per_lane = (a.m + LPN - 1) / LPN;
    if (per_lane <= 4)  return launch_rows_inst<LPN, 4>(a, stream);
    if (per_lane <= 8)  return launch_rows_inst<LPN, 8>(a, stream);
    if (per_lane <= 16) return launch_rows_inst<LPN, 16>(a, stream);
    if (per_lane <= 32) return launch_rows_inst<LPN, 32>(a, stream);
    if constexpr (LPN < 64) {
        if (per_lane <= 64) return launch_rows_inst<LPN, 64>(a, stream);
    }
    return hipErrorInvalidValue;
}

bool rows_supported(const OnchipArgs &a, int lpn)
{
    if (a.A.M > 16 || !a.stats) return false;
    const int64_t per_lane = (a.m + lpn - 1) / lpn;
    return per_lane <= (lpn == 64 ? 32 : 64);
}

hipError_t launch_rows(const OnchipArgs &a, int lpn, hipStream_t stream)
{
    if (lpn == 16) return launch_rows_lpn<16>(a, stream);
    if (lpn == 32) return launch_rows_lpn<32>(a, stream);
    return launch_rows_lpn<64>(a, stream);
}

}  // namespace gpfq
