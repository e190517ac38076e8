// Pipelined GPFQ kernel: the dense default for rows of up to 2048 samples.
//
// Replaces _quantize_neuron_parallel / _quantize_filter2D_parallel_jit
// (scripts/quantized_network.py:91-121, :185-233); same contract and the same bits as
// gpfq_rows_kernel / gpfq_onchip_kernel, which it supersedes on the layers it takes.
//
// What bound the row-group kernel (profiles/r01/sq_counters_rows_kernel.txt): its step is one
// serial chain  dot -> all-reduce -> decision -> update,  its LDS reads ran three deep behind
// scalar-cache waits, every neuron re-read every staged row from LDS, and staging was a
// synchronous copy between two barriers.  Here:
//
//  * One-step look-ahead breaks the chain.  Iteration t sweeps the residual ONCE: it applies the
//    update of step t-1 (u += f32(w X_{t-1}) - f32(q Xq_{t-1}), the reference's element-wise flow,
//    :119) and in the same pass accumulates D_{t+1} = <Xq_{t+1}, u_{t-1}>.  Decision t does not
//    wait for that sweep: it uses D_t (accumulated one iteration earlier, = <Xq_t, u_{t-2}>) plus
//    the contribution of step t-1's increment, which is known in closed form from two entries of
//    the Gram band,  <Xq_t, u_{t-1}> = D_t + w_{t-1} <Xq_t, X_{t-1}> - q_{t-1} <Xq_t, Xq_{t-1}>,
//    up to the float32 roundings of that increment.  Those are bounded rigorously
//    (|d_i - (w x_i - q xq_i)| <= 2^-23 (1 + 2^-24) (|w x_i| + |q xq_i|), subnormal products
//    2^-149), so the predicted quotient is either farther from every decision boundary than the
//    bound -- the decision is then provably the reference's -- or the wave falls back to the exact
//    dot products of :86/:89 on the completed u_{t-1} (about one decision in 10^6).  Rule (ii)'s
//    |<Xq_t,u>| < 1e-10 test is certified the same way (exact when the increment is orthogonal to
//    Xq_t element by element, e.g. at t = 0 or on disjoint supports).  The residual itself is
//    always updated by the exact element-wise flow, so u is bit-identical.
//  * Neuron blocking per lane.  A wavefront owns NPL neurons and every lane holds the same
//    EPL = m/64 sample positions of all of them, so each LDS value (X, Xq, and Xq as float64) is
//    read once for NPL neurons, and the per-step reduction and decision are shared: a packed
//    butterfly (v_permlane32_swap / v_permlane16_swap) leaves neuron n's sum in "its" 64/NPL
//    lanes, whose 16-lane DPP rows each hold a copy of the alphabet and take the decision
//    lane-parallel as the row-group kernel did.
//  * The pre-pass lays the operands out per ITERATION: record t = [row statistics of step t]
//    [X_{t-1}] [Xq_{t-1}] [Xq_{t+1} as float64, plane-swizzled for conflict-free ds_read_b128],
//    zero-padded to 64*EPL samples.  A tile of TS records is one contiguous block that the
//    workgroup streams into the other LDS buffer with global_load_lds_dwordx4 (LDS-DMA: no VGPRs,
//    no ds_write, no conversion in the hot loop) while it works on the current one; the only
//    barrier is at the tile boundary.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

namespace {

constexpr int kRecBytes = 128;

// Step record (first 128 bytes of an iteration record); all float64.
struct PipeRec {
    double rden;   // 1 / (f32-rounded ||Xq_t||)^2, 0 for rows that take rule (i)
    double G;      // <Xq_t, X_t>
    double cb;     // 2^-23 sum|Xq_t X_t| rden   (f32 rounding of w_t X_t, per unit |w_t|)
    double ca;     // 2^-149 sum|Xq_t| rden      (the same in the subnormal range)
    double H1;     // <Xq_t, X_{t-1}>
    double H2;     // <Xq_t, Xq_{t-1}>
    double E1;     // 2^-23 (1+2^-20) sum|Xq_t X_{t-1}|    (absolute, per unit |w_{t-1}|)
    double E2;     // 2^-23 (1+2^-20) sum|Xq_t Xq_{t-1}|   (absolute, per unit |q_{t-1}|)
    double Ea;     // 2^-149 (1+2^-20) sum|Xq_t|           (absolute; subnormal products)
    double nrm;    // (double) f32-rounded ||Xq_t||
    double pad[6];
};
static_assert(sizeof(PipeRec) == kRecBytes, "record header is 128 bytes");

__host__ __device__ constexpr int64_t pipe_rec_bytes(int epl) { return kRecBytes + 16 * 64 * (int64_t)epl; }

// ---- pre-pass ---------------------------------------------------------------------------------
// One workgroup per iteration record t in [0, nrec): rows t-1 (f32 copies), t (statistics) and t+1
// (float64 copy of Xq) of the caller's matrices.
__global__ void __launch_bounds__(256)
gpfq_pipe_prep_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int64_t N, int m, int epl,
                      const float *__restrict__ nrm32, char *__restrict__ recs)
{
    __shared__ double sm[7][4];
    const int64_t t = blockIdx.x;
    const int MP = 64 * epl;
    char *rb = recs + t * pipe_rec_bytes(epl);
    float  *ox  = reinterpret_cast<float *>(rb + kRecBytes);
    float  *oq  = ox + MP;
    double *od  = reinterpret_cast<double *>(rb + kRecBytes + 8 * (int64_t)MP);
    const bool has_prev = t >= 1 && t - 1 < N, has_cur = t < N, has_next = t + 1 < N;
    const float *px = X + (t - 1) * ld, *pq = Xq + (t - 1) * ld;
    const float *cx = X + t * ld, *cq = Xq + t * ld;
    const float *nq = Xq + (t + 1) * ld;
    double g = 0.0, a = 0.0, s1 = 0.0, h1 = 0.0, h2 = 0.0, e1 = 0.0, e2 = 0.0;
    for (int i = threadIdx.x; i < MP; i += 256) {
        const bool in = i < m;
        const float xp = (has_prev && in) ? px[i] : 0.f, qp = (has_prev && in) ? pq[i] : 0.f;
        ox[i] = xp;
        oq[i] = qp;
        const float qn = (has_next && in) ? nq[i] : 0.f;
        const int c = i >> 8, r = i & 255, l = r >> 2, e = r & 3;
        od[(((c * 2 + (e >> 1)) * 64 + l) << 1) + (e & 1)] = (double)qn;
        if (has_cur && in) {
            const double q = (double)cq[i], x = (double)cx[i];
            const double pr = q * x;                       // products of two f32 are exact in f64
            g += pr; a += fabs(pr); s1 += fabs(q);
            const double p1 = q * (double)xp, p2 = q * (double)qp;
            h1 += p1; e1 += fabs(p1);
            h2 += p2; e2 += fabs(p2);
        }
    }
    g = wave_sum(g); a = wave_sum(a); s1 = wave_sum(s1);
    h1 = wave_sum(h1); h2 = wave_sum(h2); e1 = wave_sum(e1); e2 = wave_sum(e2);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sm[0][wave] = g; sm[1][wave] = a; sm[2][wave] = s1; sm[3][wave] = h1; sm[4][wave] = h2; sm[5][wave] = e1; sm[6][wave] = e2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double v[7];
        for (int k = 0; k < 7; ++k) v[k] = (sm[k][0] + sm[k][1]) + (sm[k][2] + sm[k][3]);
        PipeRec rec{};
        const double nrm = has_cur ? (double)nrm32[t] : 0.0;
        const double up = 1.0 + 0x1p-20;
        rec.nrm = nrm;
        rec.rden = nrm < 1e-16 ? 0.0 : 1.0 / (nrm * nrm);
        rec.G = v[0];
        rec.cb = 0x1p-23 * v[1] * rec.rden * up;
        rec.ca = 0x1p-149 * v[2] * rec.rden * up;
        rec.H1 = v[3];
        rec.H2 = v[4];
        rec.E1 = 0x1p-23 * v[5] * up;
        rec.E2 = 0x1p-23 * v[6] * up;
        rec.Ea = 0x1p-149 * v[2] * up;
        *reinterpret_cast<PipeRec *>(rb) = rec;
    }
}

// ---- cross-lane plumbing of the hot loop -------------------------------------------------------
template <int ROR>
__device__ __forceinline__ double ror_add(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x120 + ROR, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x120 + ROR, 0xF, 0xF, true);
    return x + __hiloint2double(hi, lo);
}

// v_permlane32_swap on a float64 pair: returns (x' + y') with x' = [x.lo32, y.lo32], y' = [x.hi32, y.hi32]:
// lanes 0-31 get x[l] + x[l+32], lanes 32-63 get y[l-32] + y[l].
__device__ __forceinline__ double fold32(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// v_permlane16_swap: x' = [x.r0, y.r0, x.r2, y.r2], y' = [x.r1, y.r1, x.r3, y.r3]; returns x' + y':
// rows 0, 2 get x.r0 + x.r1 / x.r2 + x.r3, rows 1, 3 get y.r0 + y.r1 / y.r2 + y.r3.
__device__ __forceinline__ double fold16(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// Sums of NPL per-lane values over the 64 lanes; neuron n's total lands (bitwise identical) in every one of "its"
// 64/NPL lanes: lanes [n*64/NPL, (n+1)*64/NPL).
template <int NPL>
__device__ __forceinline__ double packed_allreduce(const double (&a)[NPL])
{
    double x;
    if constexpr (NPL == 4) {
        const double s02 = fold32(a[0], a[2]);       // halves: a0 | a2
        const double s13 = fold32(a[1], a[3]);       //         a1 | a3
        x = fold16(s02, s13);                        // rows:   a0, a1, a2, a3
    } else if constexpr (NPL == 2) {
        const double s = fold32(a[0], a[1]);         // halves: a0 | a1
        x = fold16(s, s);
    } else {
        const double s = fold32(a[0], a[0]);
        x = fold16(s, s);
    }
    x = ror_add<8>(x);
    x = ror_add<4>(x);
    x = ror_add<2>(x);
    x = ror_add<1>(x);
    return x;
}

// OR over the lanes of each neuron's group (GL = 16, 32 or 64 lanes), delivered to all of them.
template <int GL>
__device__ __forceinline__ unsigned group_or(unsigned x)
{
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x128, 0xF, 0xF, true);
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x124, 0xF, 0xF, true);
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x122, 0xF, 0xF, true);
    x |= (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x121, 0xF, 0xF, true);
    if constexpr (GL >= 32) {
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);
        x = r[0] | r[1];
    }
    if constexpr (GL >= 64) {
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
        x = r[0] | r[1];
    }
    return x;
}

// LDS-DMA (global_load_lds): lane l's 16 (4) bytes land at lds_dst + 16 l (4 l); lds_dst is a wave-uniform LDS byte
// address.  Issued from inline asm so that hipcc does not count it: with the builtin form it puts s_waitcnt vmcnt(0)
// in front of the next ds_read of ANY LDS address and the prefetch of the next tile would stop the current one
// (cdna_hip_programming.md 5.7).  The tile loop waits for the DMA itself (dma_wait) before its barrier.  M0 is
// saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void *g, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *g, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

// u += f32(w x) - f32(q xq) (:119) for four consecutive samples, then acc += xqd * u (float64).
// The float32 products and the subtraction are written on two-element vectors, so they become v_pk_mul_f32 /
// v_pk_add_f32 on adjacent registers: each half rounds exactly as the scalar instruction does (no contraction:
// the translation unit is built with -ffp-contract=off), and a lone wavefront on a SIMD issues two float32
// results per slot instead of one.
typedef float pk2 __attribute__((ext_vector_type(2)));

template <bool ZERO>
__device__ __forceinline__ void sweep4(double *u, double *acc, float w, float q, const float4 &x4, const float4 &q4,
                                       const double2 &da, const double2 &db)
{
    const pk2 w2 = {w, w}, q2 = {q, q};
    pk2 d01 = w2 * pk2{x4.x, x4.y}, d23 = w2 * pk2{x4.z, x4.w};
    if constexpr (!ZERO) {
        const pk2 r01 = q2 * pk2{q4.x, q4.y}, r23 = q2 * pk2{q4.z, q4.w};
        d01 = d01 - r01;
        d23 = d23 - r23;
    }
    u[0] += (double)d01.x; u[1] += (double)d01.y; u[2] += (double)d23.x; u[3] += (double)d23.y;
    acc[0] = fma(da.x, u[0], acc[0]);
    acc[1] = fma(da.y, u[1], acc[1]);
    acc[2] = fma(db.x, u[2], acc[2]);
    acc[3] = fma(db.y, u[3], acc[3]);
}

}  // namespace

// NPL neurons per wavefront, EPL samples per lane (row length 64*EPL), NW wavefronts per workgroup.
// BRANCHY: a neuron whose previous decision was 0 skips the q*Xq half of the increment (scalar branch per neuron).
template <int NPL, int EPL, int NW, bool BRANCHY>
__global__ void __launch_bounds__(64 * NW)
gpfq_pipe_kernel(const char *__restrict__ recs, const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld,
                 const float *__restrict__ Wt, int64_t ldw, AlphabetArg A, int64_t N, int m, int64_t C, int TS,
                 int8_t *__restrict__ qidx, float *__restrict__ Qt, double *__restrict__ resid, double *__restrict__ u_out,
                 unsigned long long *__restrict__ fallback_count)
{
    constexpr int MP   = 64 * EPL;
    constexpr int NCH  = EPL / 4;                  // 256-sample chunks: one float4 (and two double2) per lane each
    constexpr int GL   = 64 / NPL;                 // lanes of a neuron's decision group
    constexpr int NEUR = NW * NPL;                 // neurons per workgroup
    constexpr int64_t RB = kRecBytes + 16 * (int64_t)MP;
    static_assert(EPL % 4 == 0 && (NPL == 1 || NPL == 2 || NPL == 4), "layout");

    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tile_bytes = TS * (int)RB;
    char  *ldsT = lds;                                                            // [2][TS][RB]
    float *ldsW = reinterpret_cast<float *>(lds + 2 * (size_t)tile_bytes);        // [2][NEUR][TS]
    double *ldsA = reinterpret_cast<double *>(ldsW + 2 * NEUR * TS);              // [64] alphabet (exact path)

    const unsigned ldsT_addr = lds_addr(ldsT), ldsW_addr = lds_addr(ldsW);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp  = lane / GL;                    // neuron of this lane's decision group
    const int ka   = lane % GL;                    // alphabet slot of this lane
    const int64_t j0 = (int64_t)blockIdx.x * NEUR + (int64_t)wave * NPL;
    const int64_t jg = j0 + grp;                   // the group's neuron
    const bool g_active = jg < C;

    const double kInf = __longlong_as_double(0x7ff0000000000000LL);
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    const int M = A.M;
    const double a      = ka < M ? A.a[ka] : kNaN;
    const double a_next = ka + 1 < M ? A.a[ka + 1] : kInf;
    const double a_prev = (ka > 0 && ka <= M) ? A.a[ka - 1] : -kInf;
    const bool ascending = A.ascending != 0;
    if (tid < 64) ldsA[tid] = tid < M ? A.a[tid] : kNaN;
    for (int i = tid; i < 2 * NEUR * TS; i += 64 * NW) ldsW[i] = 0.f;

    double u[NPL][EPL];
#pragma unroll
    for (int n = 0; n < NPL; ++n)
#pragma unroll
        for (int e = 0; e < EPL; ++e) u[n][e] = 0.0;       // zeros(m), :115

    // loop-carried: per-lane partial sums of D_t, the previous step's weight / decision of every neuron
    double part[NPL];
    float  wprev[NPL], qprev[NPL];
#pragma unroll
    for (int n = 0; n < NPL; ++n) { part[n] = 0.0; wprev[n] = 0.f; qprev[n] = 0.f; }
    float wg_prev = 0.f, qg_prev = 0.f;            // the same two for this lane's group, as lane values

    int   my_idx = 0;
    float my_q   = 0.f;
    unsigned n_fallback = 0;

    const int64_t nrec = N + 1;                    // iterations 0..N (iteration N only applies update N-1)
    const int ntiles = (int)((nrec + TS - 1) / TS);

    auto load_tile = [&](int k) {
        const char *src = recs + (int64_t)k * tile_bytes;
        const unsigned dst = ldsT_addr + (unsigned)(k & 1) * (unsigned)tile_bytes;
        for (int off = wave * 1024; off < tile_bytes; off += NW * 1024)
            if (off + lane * 16 < tile_bytes) glds16(src + off + lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)(dst + off)));
        const unsigned dw = ldsW_addr + (unsigned)((k & 1) * NEUR * TS * 4);
        for (int i0 = wave * 64; i0 < NEUR * TS; i0 += NW * 64) {
            const int i = i0 + lane;
            const int n = i / TS, s = i - n * TS;
            const int64_t jn = (int64_t)blockIdx.x * NEUR + n, t = (int64_t)k * TS + s;
            if (i < NEUR * TS && jn < C && t < N) glds4(Wt + jn * ldw + t, (unsigned)__builtin_amdgcn_readfirstlane((int)(dw + 4 * i0)));
        }
    };

    __syncthreads();                               // ldsW zeros before the first DMA lands on top of them
    load_tile(0);
    dma_wait();
    __syncthreads();

    for (int k = 0; k < ntiles; ++k) {
        if (k + 1 < ntiles) load_tile(k + 1);      // streams into the other buffer while this tile is worked on
        const char  *tb = ldsT + (size_t)(k & 1) * tile_bytes;
        const float *tw = ldsW + (k & 1) * NEUR * TS;
        const int ts = (int)((nrec - (int64_t)k * TS) < TS ? (nrec - (int64_t)k * TS) : TS);
        for (int s = 0; s < ts; ++s) {
            const int64_t t = (int64_t)k * TS + s;
            const char *rb = tb + (size_t)s * RB;
            const PipeRec *rec = reinterpret_cast<const PipeRec *>(rb);
            const float  *bx = reinterpret_cast<const float *>(rb + kRecBytes) + 4 * lane;
            const float  *bq = bx + MP;
            const double *bd = reinterpret_cast<const double *>(rb + kRecBytes + 8 * (size_t)MP) + 2 * lane;

            // ---- sweep: u_{t-1} = u_{t-2} + increment of step t-1; next[n] = partial <Xq_{t+1}, u_{t-1}> ----
            double acc[NPL][4];
#pragma unroll
            for (int n = 0; n < NPL; ++n) { acc[n][0] = 0.0; acc[n][1] = 0.0; acc[n][2] = 0.0; acc[n][3] = 0.0; }
            if constexpr (BRANCHY) {
                constexpr int CG = NCH < 4 ? NCH : 4;          // chunks held in registers at a time
#pragma unroll
                for (int c0 = 0; c0 < NCH; c0 += CG) {
                    float4 x4[CG], q4[CG];
                    double2 da[CG], db[CG];
#pragma unroll
                    for (int c = 0; c < CG; ++c) {
                        x4[c] = *reinterpret_cast<const float4 *>(bx + 256 * (c0 + c));
                        q4[c] = *reinterpret_cast<const float4 *>(bq + 256 * (c0 + c));
                        da[c] = *reinterpret_cast<const double2 *>(bd + 256 * (c0 + c));
                        db[c] = *reinterpret_cast<const double2 *>(bd + 256 * (c0 + c) + 128);
                    }
#pragma unroll
                    for (int n = 0; n < NPL; ++n) {
                        if ((__float_as_uint(qprev[n]) << 1) == 0u) {
#pragma unroll
                            for (int c = 0; c < CG; ++c)
                                sweep4<true>(&u[n][4 * (c0 + c)], acc[n], wprev[n], qprev[n], x4[c], q4[c], da[c], db[c]);
                        } else {
#pragma unroll
                            for (int c = 0; c < CG; ++c)
                                sweep4<false>(&u[n][4 * (c0 + c)], acc[n], wprev[n], qprev[n], x4[c], q4[c], da[c], db[c]);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    const float4 x4 = *reinterpret_cast<const float4 *>(bx + 256 * c);
                    const float4 q4 = *reinterpret_cast<const float4 *>(bq + 256 * c);
                    const double2 da = *reinterpret_cast<const double2 *>(bd + 256 * c);
                    const double2 db = *reinterpret_cast<const double2 *>(bd + 256 * c + 128);
#pragma unroll
                    for (int n = 0; n < NPL; ++n)
                        sweep4<false>(&u[n][4 * c], acc[n], wprev[n], qprev[n], x4, q4, da, db);
                }
            }
            if (t >= N) break;                                 // iteration N: the last update only

            // ---- D_t for this lane's group: all-reduce of the partial sums of the previous sweep ----
            const double D = packed_allreduce<NPL>(part);
#pragma unroll
            for (int n = 0; n < NPL; ++n) part[n] = (acc[n][0] + acc[n][1]) + (acc[n][2] + acc[n][3]);

            // ---- decision t (:83-89, :57), one copy per 16-lane row of the group; branch-free -----
            const float wg = tw[(wave * NPL + grp) * TS + s];   // w_t of the group's neuron
            const double nrm = rec->nrm, rden = rec->rden, rG = rec->G, rcb = rec->cb, rca = rec->ca;
            const double rH1 = rec->H1, rH2 = rec->H2, rE1 = rec->E1, rE2 = rec->E2, rEa = rec->Ea;
            const bool rule1 = nrm < 1e-16;                                      // rule (i): literal 0
            const double wd = (double)wg, wpd = (double)wg_prev, qpd = (double)qg_prev;
            const double corr = fma(wpd, rH1, -(qpd * rH2));
            const double du = D + corr;                                          // predicted <Xq_t, u_{t-1}>
            const bool   inc = ((__float_as_uint(wg_prev) | __float_as_uint(qg_prev)) << 1) != 0u;
            const double eps = fma(fabs(wpd), rE1, fabs(qpd) * rE2) + (inc ? rEa : 0.0);
            const bool   du_exact = eps == 0.0;                                  // increment orthogonal to Xq_t element-wise
            const bool   msq = du_exact & (fabs(du) < 1e-10);                    // rule (ii), certain
            const bool   sure = du_exact | (fabs(du) - eps >= 1e-10);            // ... or certainly not rule (ii)
            const double wG = wd * rG;
            const double tq = (du + wG) * rden;                                  // predicted quotient
            const double tt = msq ? wd : tq;
            // twice the modelling error of the prediction (quotient units) + float64 slack
            const double delta2 = 2.0 * (fma(fabs(wd), rcb, rca) + eps * rden)
                                  + 0x1p-43 * (fabs(D) + fabs(corr) + fabs(wG)) * rden;
            const double d  = fabs(a - tt), dn = fabs(a_next - tt), dp = fabs(a_prev - tt);
            const bool c_lt = a < tt, n_lt = a_next < tt;
            const bool is_lo = c_lt & !n_lt;                                     // last member below t
            const bool is_p0 = (ka == 0) & !c_lt;                                // t at or below the whole alphabet (or NaN)
            const bool use_hi = is_lo & !(d <= dn);                              // tie -> lower index
            const int    idx_l = ka + (use_hi ? 1 : 0);
            const double q_l   = use_hi ? a_next : a;
            const double m2 = (ka + 1 < M) ? fabs(dn - d) : (dp - d);            // twice the distance from the boundary
            const bool plateau = is_lo & !use_hi & (ka > 0) & !(dp > d);         // first-index rule would pick a lower member
            const bool cert = !plateau & (msq | (m2 > delta2)) & ascending & sure;
            const bool decider = (is_lo | is_p0) & !rule1;
            unsigned w1 = decider ? __float_as_uint((float)q_l) : 0u;
            unsigned w2 = decider ? ((unsigned)idx_l | (cert ? 0u : 0x100u) | 0x200u) : 0u;
            w1 = group_or<GL>(w1);
            w2 = group_or<GL>(w2);
            float q32 = __uint_as_float(w1);                                     // rule (i): no decider, 0
            int   idx = rule1 ? A.zero_idx : (int)(w2 & 0xffu);
            const bool redo = !rule1 & ((w2 & 0x300u) != 0x200u);                // not certified (or no decider)
            if (__ballot(redo) != 0ull) {
                // rare: the reference's two dot products on the completed u_{t-1} (:86, :89), plain first-minimum scan
                double eu[NPL], ew[NPL];
                float  wn[NPL];
#pragma unroll
                for (int n = 0; n < NPL; ++n) {
                    eu[n] = 0.0; ew[n] = 0.0;
                    wn[n] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wg), n * GL));
                }
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 256 * c + 4 * lane + e;
                        float xq = 0.f, xx = 0.f;
                        if (i < m) { xq = Xq[t * ld + i]; xx = X[t * ld + i]; }
#pragma unroll
                        for (int n = 0; n < NPL; ++n) {
                            eu[n] = fma((double)xq, u[n][4 * c + e], eu[n]);
                            ew[n] = fma((double)xq, u[n][4 * c + e] + (double)__fmul_rn(wn[n], xx), ew[n]);
                        }
                    }
                const double dot_u = packed_allreduce<NPL>(eu);
                const double dot_w = packed_allreduce<NPL>(ew);
                const double te = dot_w / (nrm * nrm);
                const double t2 = fabs(dot_u) < 1e-10 ? (double)wg : te;
                int bi = 0;
                double bq2 = ldsA[0], bd2 = fabs(bq2 - t2);
                for (int kk = 1; kk < M; ++kk) {
                    const double ak = ldsA[kk], dk = fabs(ak - t2);
                    if (dk < bd2) { bd2 = dk; bi = kk; bq2 = ak; }
                }
                if (redo) { idx = bi; q32 = (float)bq2; }
                n_fallback += (redo && ka == 0 && g_active) ? 1u : 0u;
            }

            // ---- hand q_t, w_t to the next sweep (wave-uniform per neuron) and to the next decision (per group) ----
#pragma unroll
            for (int n = 0; n < NPL; ++n) {
                wprev[n] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wg), n * GL));
                qprev[n] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(q32), n * GL));
            }
            wg_prev = wg;
            qg_prev = q32;

            // ---- outputs: lane (t mod 16) of the group keeps step t until the 16-step flush ----
            const int slot = (int)(t & 15);
            if (ka == slot) { my_idx = idx; my_q = q32; }
            if (slot == 15 || t + 1 == N) {
                const int64_t base = t - slot;
                if (g_active && ka <= slot) {
                    if (qidx) qidx[jg * N + base + ka] = (int8_t)my_idx;
                    if (Qt)   Qt[jg * N + base + ka]   = my_q;
                }
            }
        }
        dma_wait();                                // this wave's share of the next tile has landed ...
        __syncthreads();                           // ... everyone's has, and everyone is done with this one
    }

    if (fallback_count && n_fallback) atomicAdd(fallback_count, (unsigned long long)n_fallback);   // rare
    if (resid) {
        double ss[NPL];
#pragma unroll
        for (int n = 0; n < NPL; ++n) {
            double s = 0.0;
#pragma unroll
            for (int e = 0; e < EPL; ++e) s = fma(u[n][e], u[n][e], s);
            ss[n] = s;
        }
        const double tot = packed_allreduce<NPL>(ss);
        if (g_active && ka == 0) resid[jg] = sqrt(tot);
    }
    if (u_out) {
#pragma unroll
        for (int n = 0; n < NPL; ++n) {
            const int64_t jn = j0 + n;
            if (jn < C) {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int i = 256 * c + 4 * lane + e;
                        if (i < m) u_out[jn * (int64_t)m + i] = u[n][4 * c + e];
                    }
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------
static int pipe_epl(int64_t m)
{
    if (m <= 256) return 4;
    if (m <= 512) return 8;
    if (m <= 1024) return 16;
    if (m <= 2048) return 32;
    return 0;
}

static int pipe_tile_steps(int epl, int neur, int64_t N)
{
    const int64_t rb = pipe_rec_bytes(epl) + 4 * (int64_t)neur;
    int ts = (int)((150 * 1024 - 1024) / (2 * rb));
    if (ts > 16) ts = 16;
    if (ts > N + 1) ts = (int)(N + 1);
    return ts < 1 ? 1 : ts;
}

bool pipe_supported(const PipeArgs &a)
{
    if (pipe_epl(a.m) == 0 || a.N < 1 || a.m < 1) return false;
    if (a.A.M > 64) return false;
    return a.N + 64 < (1LL << 31) / 64;             // tile offsets stay in int range
}

size_t pipe_workspace_bytes(int64_t N, int64_t m)
{
    const int epl = pipe_epl(m);
    if (!epl) return 0;
    // iteration records 0..N, padded to whole tiles of up to 16 records
    return (size_t)(N + 1 + 16) * (size_t)pipe_rec_bytes(epl);
}

template <int NPL, int EPL, int NW, bool BR>
static hipError_t launch_pipe_inst(const PipeArgs &a, hipStream_t stream)
{
    constexpr int NEUR = NW * NPL;
    int ts = pipe_tile_steps(EPL, NEUR, a.N);
    if (a.ts_override > 0 && a.ts_override < ts) ts = a.ts_override;
    const size_t lds_bytes = 2 * (size_t)ts * pipe_rec_bytes(EPL) + 2 * (size_t)NEUR * ts * sizeof(float) + 64 * sizeof(double);
    const unsigned grid = (unsigned)((a.C + NEUR - 1) / NEUR);
    auto *kern = gpfq_pipe_kernel<NPL, EPL, NW, BR>;
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds_bytes, stream, static_cast<const char *>(a.workspace), a.X, a.Xq, a.ld,
                       a.Wt, a.ldw, a.A, a.N, (int)a.m, a.C, ts, a.qidx, a.Qt, a.resid, a.u_out, a.fallback_count);
    return hipGetLastError();
}

template <int NPL, int NW>
static hipError_t launch_pipe_epl(const PipeArgs &a, int epl, hipStream_t stream)
{
    const bool br = !(a.variant & 1);
    switch (epl) {
    case 4:  return br ? launch_pipe_inst<NPL, 4, NW, true>(a, stream)  : launch_pipe_inst<NPL, 4, NW, false>(a, stream);
    case 8:  return br ? launch_pipe_inst<NPL, 8, NW, true>(a, stream)  : launch_pipe_inst<NPL, 8, NW, false>(a, stream);
    case 16: return br ? launch_pipe_inst<NPL, 16, NW, true>(a, stream) : launch_pipe_inst<NPL, 16, NW, false>(a, stream);
    default: return br ? launch_pipe_inst<NPL, 32, NW, true>(a, stream) : launch_pipe_inst<NPL, 32, NW, false>(a, stream);
    }
}

hipError_t launch_pipe(const PipeArgs &a, hipStream_t stream)
{
    const int epl = pipe_epl(a.m);
    if (!epl) return hipErrorInvalidValue;
    // pre-pass: iteration records (statistics + padded operand rows), tiles padded with zero records
    const int64_t nrec = a.N + 1 + 16;
    hipLaunchKernelGGL(gpfq_pipe_prep_kernel, dim3((unsigned)nrec), dim3(256), 0, stream, a.X, a.Xq, a.ld, a.N, (int)a.m, epl,
                       a.nrm32, static_cast<char *>(a.workspace));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int npl = a.npl;
    if (npl == 0) npl = a.A.M <= 16 ? 4 : (a.A.M <= 32 ? 2 : 1);
    if (a.A.M > 16 && npl > 2) npl = 2;
    if (a.A.M > 32) npl = 1;
    if (npl == 4) return launch_pipe_epl<4, 4>(a, epl, stream);
    if (npl == 2) return launch_pipe_epl<2, 8>(a, epl, stream);
    return launch_pipe_epl<1, 8>(a, epl, stream);
}

}  // namespace gpfq
