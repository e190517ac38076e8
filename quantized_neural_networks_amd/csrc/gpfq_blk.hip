// Block-pipelined GPFQ kernel: gpfq_pipe.hip's roles (eight sweep wavefronts over the sample axis, one decision
// wavefront per workgroup), with B steps per slot instead of one.
//
// Replaces _quantize_neuron_parallel / _quantize_filter2D_parallel_jit (scripts/quantized_network.py:91-121, :185-233);
// same contract and the same bits as the other dense kernels.
//
// Why blocks.  With one step per slot (gpfq_pipe.hip) 45 % of a slot is not arithmetic: the fold of the partial dot
// products, the LDS hand-off of (w, q) and of the partial sums, and the barrier are paid per STEP and sit on the
// critical path of every sweep wavefront (profiles/r02: 1375 of 2900 cycles per step with the sweep removed).  Here a
// slot covers a block of B steps:
//
//  * sweep, slot b: applies the B updates of block b-1 to the residual (the reference's element-wise f32/f64 flow,
//    :119, one after the other) and then accumulates the B dot products <Xq_t, u> of block b+1's rows against the
//    residual as it stands after block b-1.  One fold sequence per row (pipelined), one hand-off, one barrier per block.
//  * decision wavefront, slot b: takes the B decisions of block b in order.  Step t = bB + s uses
//        <Xq_t, u_{t-1}> = D_t + sum over t' in [ (b-1)B, t ) of ( w_t' <Xq_t, X_t'> - q_t' <Xq_t, Xq_t'> )  + roundings,
//    D_t from the sweep one slot earlier (residual after block b-2), the sum from the pre-pass's Gram BAND of width
//    2B-1 (<Xq_t, X_t'>, <Xq_t, Xq_t'> and the absolute sums that bound the float32 roundings of those increments:
//    |d_i - (w x_i - q xq_i)| <= 2^-23 (1 + 2^-24) (|w x_i| + |q xq_i|), subnormal products 2^-149).  A decision whose
//    predicted quotient is farther from every boundary than the accumulated bound is provably the reference's; the
//    first one of a neuron that is not stops that neuron's chain for the block.
//  * slow path (about one decision in 10^6): after the slot's barrier, while any neuron is stopped at step S (the
//    smallest such S), the sweep wavefronts form the reference's two exact dot products of step bB + S (:86, :89) on the
//    residual they hold, replaying the block's already decided updates into temporaries (never into u), the decision
//    wavefront takes the exact decision and resumes the chain; two extra barriers per round.
//  The residual itself is only ever updated by the exact element-wise flow with final decisions, so u is bit-identical.
//
// Layout: slot record t = [statistics of step t + Gram band of row t][X_{t-B}][Xq_{t-B}][Xq_{t+B} as float64],
// zero-padded; a tile = the B records of a slot, streamed into the other LDS buffer by LDS-DMA during the slot -- one 1 KiB
// piece at a time from inside the sweep's update phase (scalar addressing): issued back to back at the top of the slot the
// pieces filled the vector-memory queue and stalled every sweep ~1000 cycles per slot.
//
// Shapes (blk_shape): G neuron groups per sweep wavefront (4G neurons per workgroup), S sample pairs per k-lane over the
// sweep wavefronts, B steps per slot, NSW sweep wavefronts -- <4,16|24|32,4> x 8 for rows of up to 512 / 768 / 1024 samples,
// <2,24|32,2> x 8 up to 1536 / 2048 (8 neurons per workgroup), <4,48|64,2> x 11 for the same rows in layers of more than
// 2048 neurons (16 neurons per workgroup: one round of workgroups instead of two, half the folds and decisions per weight:
// 8.2 -> 6.5 ms at 4096 x 4096 x 2048), and <2,16,4> x 8 (8 neurons per workgroup: half the sweep per slot) for rows of
// 769..1024 samples when the layer has at most 2048 neurons, <1,8,4> / <1,12|16,2> x 8 (4 neurons per workgroup) for rows of 769..2048
// samples when it has at most 1024, and one step per slot -- <2,48|64,1> x 11, <1,24|32,1> x 8 -- for rows of 2049..4096 samples.
// What bounds a slot: the sweeps (nine sample pairs on three of
// the SIMDs) for <4,32,4> and the B = 2 shapes, the decision wavefront's chain everywhere else; it runs at raised priority
// (s_setprio) because it shares its SIMD with two sweep wavefronts, and the sweeps lower theirs as they progress through the
// slot.  Symmetric alphabets take the SYM instantiation (BlkK::sym_a).  profiles/r02/blk_phase_stamps.txt has the per-phase
// cycle counts (diagnostic build: GPFQ_DIAG="-DGPFQ_BLK_STAMPS").
#include <atomic>
#include <cmath>
#include <cstring>
#include <type_traits>

#include <cstdlib>
#include <mutex>
#include "gpfq_device.hpp"
#include <hip/hip_ext.h>
#include "gpfq_launch.hpp"
#include "gpfq_roles.hpp"

namespace gpfq {

namespace {

struct BlkStats {          // first 64 bytes of a record header (all float64); see gpfq_pipe.hip's PipeRec
    double rden, G, cb, ca, Ea, nrm;
    double sE1, sE2;       // round 5: sums of the band's E1 / E2 over ALL its distances (rounded up): the quick certification's bound
};
struct BandEntry {         // distance d = 1 .. 2B-1 (cluster form: 3B-1) at header offset 64 + 32 (d - 1)
    double H1, H2, E1, E2; // <Xq_t, X_{t-d}>, <Xq_t, Xq_{t-d}>, 2^-23 (1+2^-20) sum|Xq_t X_{t-d}|, ... sum|Xq_t Xq_{t-d}|
};
static_assert(sizeof(BlkStats) == 64 && sizeof(BandEntry) == 32, "header layout");

// (cluster form, CL: the partial dot products travel between the slices for a slot, so a decision's D is TWO slots old and the band
//  covers the 3B - 1 increments then pending)
__host__ __device__ constexpr int blk_band(int B, bool CL = false) { return CL ? 3 * B - 1 : 2 * B - 1; }
__host__ __device__ constexpr int blk_hdr_bytes(int B, bool CL = false) { return (64 + blk_band(B, CL) * 32 + 127) & ~127; }
// The row t + B travels as float64 (the sweep's dot products then need no conversion) -- except in the one-step-per-slot shapes,
// which are bound by the record stream itself: there it stays float32 (12 instead of 16 bytes per sample) and is converted in the sweep.
// -- and in the shapes with one neuron group per sweep wavefront (G = 1: 4 or 2 neurons per workgroup, the narrow layers), which put
// up to 256 workgroups on the chip that EACH pull the whole record stream out of the L2s: 256 x 66 KiB per slot of four steps was
// 10 TB/s, the bound of those shapes (profiles/r03/blk_phase_stamps.txt); a lane converts its two samples for two or four neurons.
__host__ __device__ constexpr bool blk_row64(int G, int B) { return B > 1 && G > 1; }
// (record skews against LDS bank conflicts were measured twice and bought nothing: DESIGN_HISTORY.md, "gpfq_blk.hip notes" 1)
__host__ __device__ constexpr int64_t blk_rec_bytes(int64_t mp, int B, int G, bool CL = false) { return blk_hdr_bytes(B, CL) + (blk_row64(G, B) ? 16 : 12) * mp; }

// ---- pre-pass ---------------------------------------------------------------------------------
// One workgroup per slot record t: statistics of row t, its Gram band against rows t-1 .. t-(2B-1), float32 copies
// of rows t-B, the float64 copy of Xq row t+B; zero-padded to mp samples.  One pass over the samples with all 3 + 4 (2B-1)
// sums in registers and ONE workgroup reduction.  The launch is bound by its float64 arithmetic (about 60 operations per
// sample of a row: seven band distances x four sums): 49 us for 4096 rows of 1024 samples.
template <int B, bool R64, bool CL = false>
__global__ void __launch_bounds__(256)
gpfq_blk_prep_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int64_t N, int m, int mp,
                     const float *__restrict__ nrm32, char *__restrict__ recs, char *__restrict__ hdrs,
                     const DevAlphabet *__restrict__ alpha, int sym, int nsl, int64_t slice_bytes)
{
    // (symmetric form, BlkK: the records hold f32(a32 Xq) in place of Xq; a32 comes from the device alphabet)
    const float sym_a = sym ? alpha->sym_a : 0.f;
    // (cluster form, nsl > 1: mp = nsl record rows of mp / nsl samples each, slice s's record stream slice_bytes behind slice s - 1's;
    //  the statistics and the Gram band are the whole row's and go into every slice's record)
    constexpr int GREC = R64 ? 2 : 1;                         // (any G with this row format: the record size depends on nothing else)
    constexpr int ND = blk_band(B, CL), NV = 3 + 4 * ND;
    constexpr int DN = CL ? 2 * B : B;                        // the float64 row of a record: row t + DN (cluster form: two slots ahead)
    __shared__ double sm[4][((NV + 3) & ~3) + 1];
    const int64_t t = blockIdx.x;
    constexpr int hdr = blk_hdr_bytes(B, CL);
    const int mps = nsl > 1 ? mp / nsl : mp;                  // samples of a record row
    char *rb = recs + t * blk_rec_bytes(mps, B, GREC, CL);
    float  *ox = reinterpret_cast<float *>(rb + hdr);
    float  *oq = ox + mps;
    double *od = reinterpret_cast<double *>(rb + hdr + 8 * (int64_t)mps);
    const bool has_prev = t >= B && t - B < N, has_cur = t < N, has_next = t + DN < N;
    const float *px = X + (t - B) * ld, *pq = Xq + (t - B) * ld;
    const float *cx = X + t * ld, *cq = Xq + t * ld;
    const float *nq = Xq + (t + DN) * ld;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double up = 1.0 + 0x1p-20;

    constexpr int NP = (NV + 3) & ~3;
    double v[NP];                                             // G, sum|Xq X|, sum|Xq|, then (H1, H2, sum|.|, sum|.|) per distance
#pragma unroll
    for (int k = 0; k < NP; ++k) v[k] = 0.0;
    auto one = [&](float xc, float qc_, const float (&bx)[ND], const float (&bq)[ND]) {   // one sample of row t against its band
        const double q = (double)qc_, pr = q * (double)xc;       // products of two f32 are exact in f64
        v[0] += pr; v[1] += fabs(pr); v[2] += fabs(q);
#pragma unroll
        for (int d = 1; d <= ND; ++d) {
            const double p1 = q * (double)bx[d - 1], p2 = q * (double)bq[d - 1];
            v[3 + 4 * (d - 1) + 0] += p1;       v[3 + 4 * (d - 1) + 1] += p2;
            v[3 + 4 * (d - 1) + 2] += fabs(p1); v[3 + 4 * (d - 1) + 3] += fabs(p2);
        }
    };
    const bool vec = (ld % 4 == 0) && (((uintptr_t)X | (uintptr_t)Xq) % 16 == 0);
    if (vec) {
        // four consecutive samples per thread, 16-byte accesses: a quarter of the memory instructions (rows absent from the
        // walk -- before the first, after the last -- read as zeros and add nothing)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 4 * threadIdx.x; i < mp; i += 1024) {
            auto row = [&](const float *base, int64_t r) -> float4 {
                if (r < 0 || r >= N || i >= m) return z4;
                float4 w = *reinterpret_cast<const float4 *>(base + r * ld + i);
                if (i + 3 >= m) { w.y = i + 1 < m ? w.y : 0.f; w.z = i + 2 < m ? w.z : 0.f; w.w = 0.f; }   // (m % 4 != 0: the row's tail)
                return w;
            };
            const float4 xp = row(X, t - B), qp = row(Xq, t - B), qn = row(Xq, t + DN);
            const float4 xc = row(X, t), qc4 = row(Xq, t);
            float4 bx4[ND], bq4[ND];
#pragma unroll
            for (int d = 1; d <= ND; ++d) { bx4[d - 1] = row(X, t - d); bq4[d - 1] = row(Xq, t - d); }
            const int sl = nsl > 1 ? i / mps : 0, il = i - sl * mps;      // slice and sample of the slice (mps is a multiple of four)
            const int64_t so = (int64_t)sl * slice_bytes;
            float *oxs = reinterpret_cast<float *>(reinterpret_cast<char *>(ox) + so), *oqs = reinterpret_cast<float *>(reinterpret_cast<char *>(oq) + so);
            double *ods = reinterpret_cast<double *>(reinterpret_cast<char *>(od) + so);
            *reinterpret_cast<float4 *>(oxs + il) = xp;
            *reinterpret_cast<float4 *>(oqs + il) = sym_a != 0.f ? make_float4(__fmul_rn(sym_a, qp.x), __fmul_rn(sym_a, qp.y), __fmul_rn(sym_a, qp.z), __fmul_rn(sym_a, qp.w)) : qp;
            if constexpr (R64) {
                *reinterpret_cast<double2 *>(ods + il) = make_double2((double)qn.x, (double)qn.y);
                *reinterpret_cast<double2 *>(ods + il + 2) = make_double2((double)qn.z, (double)qn.w);
            } else {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(ods) + il) = qn;
            }
            float b1[ND], b2[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = bx4[d].x; b2[d] = bq4[d].x; }
            one(xc.x, qc4.x, b1, b2);
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = bx4[d].y; b2[d] = bq4[d].y; }
            one(xc.y, qc4.y, b1, b2);
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = bx4[d].z; b2[d] = bq4[d].z; }
            one(xc.z, qc4.z, b1, b2);
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = bx4[d].w; b2[d] = bq4[d].w; }
            one(xc.w, qc4.w, b1, b2);
        }
    } else {
        for (int i = threadIdx.x; i < mp; i += 256) {
            const bool in = i < m;
            const int sl = nsl > 1 ? i / mps : 0, il = i - sl * mps;
            const int64_t so = (int64_t)sl * slice_bytes;
            float *oxs = reinterpret_cast<float *>(reinterpret_cast<char *>(ox) + so), *oqs = reinterpret_cast<float *>(reinterpret_cast<char *>(oq) + so);
            double *ods = reinterpret_cast<double *>(reinterpret_cast<char *>(od) + so);
            oxs[il] = (has_prev && in) ? px[i] : 0.f;
            oqs[il] = (has_prev && in) ? (sym_a != 0.f ? __fmul_rn(sym_a, pq[i]) : pq[i]) : 0.f;   // symmetric form: f32(a Xq), see BlkK::sym_a
            if constexpr (R64) ods[il] = (double)((has_next && in) ? nq[i] : 0.f);
            else reinterpret_cast<float *>(ods)[il] = (has_next && in) ? nq[i] : 0.f;
            if (has_cur && in) {
                float b1[ND], b2[ND];
#pragma unroll
                for (int d = 1; d <= ND; ++d) {
                    b1[d - 1] = t - d >= 0 ? X[(t - d) * ld + i] : 0.f;
                    b2[d - 1] = t - d >= 0 ? Xq[(t - d) * ld + i] : 0.f;
                }
                one(cx[i], cq[i], b1, b2);
            }
        }
    }
    // wavefront sums four values at a time: halves and rows fold by swaps (row r ends up with value r of the group), then
    // rotations inside the rows -- 7 additions per group instead of 24 for four separate butterflies (the launch is bound by
    // exactly this arithmetic: ~4 samples per thread against 31 sums)
#pragma unroll
    for (int g = 0; g < NP / 4; ++g) {
        double x = fold16(fold32(v[4 * g], v[4 * g + 2]), fold32(v[4 * g + 1], v[4 * g + 3]));
        x = ror_add<8>(x); x = ror_add<4>(x); x = ror_add<2>(x); x = ror_add<1>(x);
        if ((lane & 15) == 0) sm[wave][4 * g + (lane >> 4)] = x;
    }
    __syncthreads();
    auto total = [&](int k) { return (sm[0][k] + sm[1][k]) + (sm[2][k] + sm[3][k]); };
    if (threadIdx.x == 0) {
        BlkStats st{};
        const double nrm = has_cur ? (double)nrm32[t] : 0.0;
        const double s1 = total(1), s2 = total(2);
        st.nrm = nrm;
        st.rden = nrm < 1e-16 ? 0.0 : 1.0 / (nrm * nrm);
        st.G = total(0);
        st.cb = 0x1p-23 * s1 * st.rden * up;
        st.ca = 0x1p-149 * s2 * st.rden * up;
        st.Ea = 0x1p-149 * s2 * up;
        double e1 = 0.0, e2 = 0.0;                                // (the band entries' own E1, E2: same products, summed; `up` twice covers the sums' roundings)
#pragma unroll
        for (int d = 1; d <= ND; ++d) { e1 += total(5 + 4 * (d - 1)); e2 += total(6 + 4 * (d - 1)); }
        st.sE1 = 0x1p-23 * e1 * up * up; st.sE2 = 0x1p-23 * e2 * up * up;
        for (int sl = 0; sl < (nsl > 1 ? nsl : 1); ++sl) *reinterpret_cast<BlkStats *>(rb + (int64_t)sl * slice_bytes) = st;
        *reinterpret_cast<BlkStats *>(hdrs + t * hdr) = st;
    } else if (threadIdx.x <= ND) {
        const int d = threadIdx.x;
        BandEntry e;
        e.H1 = total(3 + 4 * (d - 1)); e.H2 = total(4 + 4 * (d - 1));
        e.E1 = 0x1p-23 * total(5 + 4 * (d - 1)) * up; e.E2 = 0x1p-23 * total(6 + 4 * (d - 1)) * up;
        for (int sl = 0; sl < (nsl > 1 ? nsl : 1); ++sl) *reinterpret_cast<BandEntry *>(rb + (int64_t)sl * slice_bytes + 64 + 32 * (d - 1)) = e;
        *reinterpret_cast<BandEntry *>(hdrs + t * hdr + 64 + 32 * (d - 1)) = e;
    }
}

// The same pre-pass for a RUN of 8 .. 16 consecutive records per workgroup (round 6; classic form, 16-byte aligned rows): row t of X and Xq is
// an operand of records t .. t + ND (the band) and of records t + B, t - B (the operand rows), so one workgroup per record pulls every
// row out of the L2s about 2 ND + 4 = 18 times -- 295 MB through the L2s for the headline layer's 32 MB, which with the one-pass-per-
// workgroup latency is the launch's 49 us.  Here a workgroup keeps the band's rows of its 4 samples per thread in registers as a sliding
// window and steps through RUN records: three row reads per record (t of both matrices, t + B of Xq), the wavefronts' partial sums of
// every record parked in LDS and ONE workgroup reduction for the whole run.  Same records, bit for bit, for rows of up to 1024 padded
// samples (one chunk per thread: the same order of additions); longer rows add their chunks' wavefront sums in chunk order.
constexpr int kPrepRunMax = 16;               // records per workgroup: 4 .. 16, chosen per launch (launch_blk)
template <int B, bool R64>
__global__ void __launch_bounds__(256)
gpfq_blk_prep_run_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int64_t N, int m, int mp, int64_t nrec, int RUN,
                         const float *__restrict__ nrm32, char *__restrict__ recs, char *__restrict__ hdrs,
                         const DevAlphabet *__restrict__ alpha, int sym, unsigned *__restrict__ zero16)
{
    constexpr int GREC = R64 ? 2 : 1;
    constexpr int ND = blk_band(B), NV = 3 + 4 * ND, NP = (NV + 3) & ~3;
    constexpr int hdr = blk_hdr_bytes(B);
    __shared__ double sm[kPrepRunMax][4][NP + 1];
    // nrm32 == NULL (rows of exactly ONE 16-byte piece per thread -- 1024 padded samples --, m % 4 == 0: launch_blk): the row norms are formed here, as gpfq_row_norms_kernel forms
    // them -- the same four products per thread in the same order, the same wavefront sum, the same sum of the four wavefronts -- bit for bit
    __shared__ double nsm[kPrepRunMax][4];
    const bool own_norms = nrm32 == nullptr;
    if (zero16 && blockIdx.x == 0 && threadIdx.x < 16) zero16[threadIdx.x] = 0u;    // (the call's counter block: gpfq_quantize_dense_layer)
    const float sym_a = sym ? alpha->sym_a : 0.f;
    const int64_t t0 = (int64_t)blockIdx.x * RUN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double up = 1.0 + 0x1p-20;
    for (int k = threadIdx.x; k < RUN * 4 * (NP + 1); k += 256) (&sm[0][0][0])[k] = 0.0;     // (RUN: a launch parameter, <= kPrepRunMax)
    __syncthreads();
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t rbytes = blk_rec_bytes(mp, B, GREC);
    for (int i = 4 * threadIdx.x; i < mp; i += 1024) {
        auto row = [&](const float *base, int64_t r) -> float4 {
            if (r < 0 || r >= N || i >= m) return z4;
            float4 w = *reinterpret_cast<const float4 *>(base + r * ld + i);
            if (i + 3 >= m) { w.y = i + 1 < m ? w.y : 0.f; w.z = i + 2 < m ? w.z : 0.f; w.w = 0.f; }   // (m % 4 != 0: the row's tail)
            return w;
        };
        float4 wx[ND], wq[ND];                                    // the window: rows t - 1 .. t - ND of the record in hand
#pragma unroll
        for (int d = 1; d <= ND; ++d) { wx[d - 1] = row(X, t0 - d); wq[d - 1] = row(Xq, t0 - d); }
        float4 xc = row(X, t0), qc4 = row(Xq, t0), qn = row(Xq, t0 + B);
        // (a run-time trip count: the window slides by register moves -- 56 of them per record, next to ~330 arithmetic instructions in a
        //  launch that waits for memory)
#pragma unroll 1
        for (int r = 0; r < RUN; ++r) {
            const int64_t t = t0 + r;
            // (the next record's three rows: requested before this record's arithmetic)
            const float4 xc1 = r + 1 < RUN ? row(X, t + 1) : z4, qc1 = r + 1 < RUN ? row(Xq, t + 1) : z4, qn1 = r + 1 < RUN ? row(Xq, t + 1 + B) : z4;
            if (t < nrec) {
                char *rb = recs + t * rbytes;
                float *ox = reinterpret_cast<float *>(rb + hdr), *oq = ox + mp;
                const float4 xp = wx[B - 1], qp = wq[B - 1];      // rows t - B
                *reinterpret_cast<float4 *>(ox + i) = xp;
                *reinterpret_cast<float4 *>(oq + i) = sym_a != 0.f ? make_float4(__fmul_rn(sym_a, qp.x), __fmul_rn(sym_a, qp.y), __fmul_rn(sym_a, qp.z), __fmul_rn(sym_a, qp.w)) : qp;
                if constexpr (R64) {
                    double *od = reinterpret_cast<double *>(rb + hdr + 8 * (int64_t)mp);
                    *reinterpret_cast<double2 *>(od + i) = make_double2((double)qn.x, (double)qn.y);
                    *reinterpret_cast<double2 *>(od + i + 2) = make_double2((double)qn.z, (double)qn.w);
                } else {
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(rb + hdr + 8 * (int64_t)mp) + i) = qn;
                }
            }
            if (own_norms) {
                double s = 0.0;
                s = fma((double)qc4.x, (double)qc4.x, s);
                s = fma((double)qc4.y, (double)qc4.y, s);
                s = fma((double)qc4.z, (double)qc4.z, s);
                s = fma((double)qc4.w, (double)qc4.w, s);
                s = wave_sum(s);
                if (lane == 0) nsm[r][wave] = s;
            }
            double v[NP];                                         // G, sum|Xq X|, sum|Xq|, then (H1, H2, sum|.|, sum|.|) per distance
#pragma unroll
            for (int k = 0; k < NP; ++k) v[k] = 0.0;
            auto one = [&](float xcs, float qcs, const float (&bx)[ND], const float (&bq)[ND]) {
                const double q = (double)qcs, pr = q * (double)xcs;
                v[0] += pr; v[1] += fabs(pr); v[2] += fabs(q);
#pragma unroll
                for (int d = 1; d <= ND; ++d) {
                    const double p1 = q * (double)bx[d - 1], p2 = q * (double)bq[d - 1];
                    v[3 + 4 * (d - 1) + 0] += p1;       v[3 + 4 * (d - 1) + 1] += p2;
                    v[3 + 4 * (d - 1) + 2] += fabs(p1); v[3 + 4 * (d - 1) + 3] += fabs(p2);
                }
            };
            float b1[ND], b2[ND];
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = wx[d].x; b2[d] = wq[d].x; }
            one(xc.x, qc4.x, b1, b2);
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = wx[d].y; b2[d] = wq[d].y; }
            one(xc.y, qc4.y, b1, b2);
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = wx[d].z; b2[d] = wq[d].z; }
            one(xc.z, qc4.z, b1, b2);
#pragma unroll
            for (int d = 0; d < ND; ++d) { b1[d] = wx[d].w; b2[d] = wq[d].w; }
            one(xc.w, qc4.w, b1, b2);
#pragma unroll
            for (int g = 0; g < NP / 4; ++g) {
                double x = fold16(fold32(v[4 * g], v[4 * g + 2]), fold32(v[4 * g + 1], v[4 * g + 3]));
                x = ror_add<8>(x); x = ror_add<4>(x); x = ror_add<2>(x); x = ror_add<1>(x);
                if ((lane & 15) == 0) sm[r][wave][4 * g + (lane >> 4)] += x;     // (this thread's slot in every chunk)
            }
            // slide the window
#pragma unroll
            for (int d = ND - 1; d >= 1; --d) { wx[d] = wx[d - 1]; wq[d] = wq[d - 1]; }
            wx[0] = xc; wq[0] = qc4;
            xc = xc1; qc4 = qc1; qn = qn1;
        }
    }
    __syncthreads();
    // the run's headers: thread (r, e) -- e = 0 the statistics, e = d the band entry at distance d
    for (int w = threadIdx.x; w < RUN * (ND + 1); w += 256) {
        const int r = w / (ND + 1), e = w - r * (ND + 1);
        const int64_t t = t0 + r;
        if (t >= nrec) continue;
        auto total = [&](int k) { return (sm[r][0][k] + sm[r][1][k]) + (sm[r][2][k] + sm[r][3][k]); };
        char *rb = recs + t * rbytes;
        if (e == 0) {
            BlkStats st{};
            const double nrm = t < N ? (own_norms ? (double)(float)sqrt(nsm[r][0] + nsm[r][1] + nsm[r][2] + nsm[r][3]) : (double)nrm32[t]) : 0.0;
            const double s1 = total(1), s2 = total(2);
            st.nrm = nrm;
            st.rden = nrm < 1e-16 ? 0.0 : 1.0 / (nrm * nrm);
            st.G = total(0);
            st.cb = 0x1p-23 * s1 * st.rden * up;
            st.ca = 0x1p-149 * s2 * st.rden * up;
            st.Ea = 0x1p-149 * s2 * up;
            double e1 = 0.0, e2 = 0.0;
#pragma unroll
            for (int d = 1; d <= ND; ++d) { e1 += total(5 + 4 * (d - 1)); e2 += total(6 + 4 * (d - 1)); }
            st.sE1 = 0x1p-23 * e1 * up * up; st.sE2 = 0x1p-23 * e2 * up * up;
            *reinterpret_cast<BlkStats *>(rb) = st;
            *reinterpret_cast<BlkStats *>(hdrs + t * hdr) = st;
        } else {
            const int d = e;
            BandEntry be;
            be.H1 = total(3 + 4 * (d - 1)); be.H2 = total(4 + 4 * (d - 1));
            be.E1 = 0x1p-23 * total(5 + 4 * (d - 1)) * up; be.E2 = 0x1p-23 * total(6 + 4 * (d - 1)) * up;
            *reinterpret_cast<BandEntry *>(rb + 64 + 32 * (d - 1)) = be;
            *reinterpret_cast<BandEntry *>(hdrs + t * hdr + 64 + 32 * (d - 1)) = be;
        }
    }
}

// Symmetric form, two-phase calls (gpfq_dense_layer_prepare / _run): the pre-pass ran before the alphabet existed and left Xq_{t-B} as it
// is; this pass multiplies those rows by a32 in place (the same single float32 product the fused pre-pass forms).  grid (records, slices).
__global__ void __launch_bounds__(256)
gpfq_blk_scale_kernel(char *__restrict__ recs, int64_t rec_bytes, int hdr, int mps, int64_t slice_bytes, const DevAlphabet *__restrict__ alpha)
{
    const float a = alpha->sym_a;
    float4 *row = reinterpret_cast<float4 *>(recs + (int64_t)blockIdx.y * slice_bytes + (int64_t)blockIdx.x * rec_bytes + hdr + 4 * (int64_t)mps);
    for (int i = threadIdx.x; i < mps / 4; i += 256) {
        const float4 v = row[i];
        row[i] = make_float4(__fmul_rn(a, v.x), __fmul_rn(a, v.y), __fmul_rn(a, v.z), __fmul_rn(a, v.w));
    }
}

// Who writes a block's outputs from the LDS ring to memory (round 5).  The sweep wavefronts, block b - 1 at the top of slot b, one wavefront per
// slot in turn, wherever the decision wavefront IS the slot -- the four-group shapes with one neuron per lane: its own flush (eight slots'
// worth at a time, 64-bit address arithmetic on operands hipcc spills at three wavefronts per SIMD) cost those shapes 9-12 % while the sweeps
// idle at the barrier.  Everywhere else the decision wavefront, which has the slack there (profiles/r05/sweep_side_flush*.txt).
// (and the cluster form: its decision wavefront has no register to spare for the flush's addresses)
template <int G, int NL, bool CL = false> constexpr bool blk_sweep_flush() { return (G == 4 && NL == 1) || CL; }

// LDS carve-up (byte offsets), shared by host and device.
struct BlkLds {
    int tile_bytes, tile_pitch, off_w, off_d, off_wq, off_x2, off_e, off_out, off_ctl, off_zero, off_hr, hr_pitch, total;
};
constexpr int kOutSteps = 32;                                    // steps of outputs staged in LDS (a ring: block b's leave it in slot b + 1)

// Partial sums have one slot per sweep wavefront, rounded up to the decision wavefront's 64 / NB sub-lanes per neuron
// (sub-lane r adds slots r, r + R, ...; a slot no wavefront writes stays zero).
__host__ __device__ constexpr int blk_sublanes(int nb) { return 64 / nb > 8 ? 8 : 64 / nb; }   // R: lanes of the decision wavefront per neuron
__host__ __device__ constexpr int blk_slots(int nsw, int nb) { return (nsw + blk_sublanes(nb) - 1) / blk_sublanes(nb) * blk_sublanes(nb); }

__host__ __device__ inline BlkLds blk_lds(int mp, int nb, int B, int nsw, int G, bool CL = false)
{
    BlkLds L;
    const int nw = blk_slots(nsw, nb);
    L.tile_bytes = B * (int)blk_rec_bytes(mp, B, G, CL);
    L.tile_pitch = (L.tile_bytes + 1023) & ~1023;               // the DMA moves whole 1 KiB pieces
    int o = 2 * L.tile_pitch;
    L.off_w = o;    o += 2 * nb * B * 4;    o = (o + 15) & ~15; // [2][NB][B] f32        weights of the block
    L.off_d = o;    o += 2 * nw * B * nb * 8;                   // [2][NW][B][NB] f64    partial dot products
    L.off_wq = o;   o += 2 * nb * B * 8;                        // [2][NB][B] (w, q) f32 decisions of the block
    L.off_x2 = o;   o += nw * nb * 16;                          // [NW][NB] (f64, f64)   exact partials (slow path)
    L.off_e = o;    o += 68 * 8;                                // [2 + 64 + 2] f64      -inf, -inf, alphabet, +inf, +inf
    L.off_out = o;  o += nb * kOutSteps * 8;                    // [NB][32] (idx i32, q f32) until a sweep wavefront writes them out
    L.off_ctl = o;  o += 16;                                    // [2] (by slot parity) smallest step a neuron of the block is stopped at, -1: none; [2] dummy
    L.off_zero = o; o += 32;                                    // zeros (band entries a step does not have)
    o = (o + 15) & ~15;
    L.hr_pitch = (B * blk_hdr_bytes(B, CL) + 1023) & ~1023;     // the headers of a tile, whole 1 KiB DMA pieces
    L.off_hr = o;   o += 3 * L.hr_pitch;                        // [3] header ring: tile k in buffer k % 3 (lands two slots ahead)
    L.total = o;
    return L;
}

struct BlkK {
    const char *recs;
    const char *hdrs;           // the record headers once more, compact: [record][blk_hdr_bytes(B)] (the decision wavefront's copy)
    const float *Wt;
    int64_t ldw, ldt;           // weight of neuron j at step t: Wt[j ldw + t ldt] -- neuron-major rows (ldt = 1) or the Keras kernel itself (ldw = 1, ldt = its row pitch)
    int64_t N, C;
    int m, M, zero_idx, nblk;   // nblk = ceil(N / B) blocks, nblk + 1 slots
    int8_t *qidx;
    float *Qt;
    int64_t o_sj, o_st;         // output element (neuron j, step t) at j o_sj + t o_st: neuron-major [C][N] (N, 1) or Keras layout [N][ldo] (1, ldo)
    const DevAlphabet *alpha;   // the layer alphabet in device memory (gpfq_alphabet_setup_kernel); alpha->ok == 0: nothing runs, *alpha_err is raised
    int *alpha_err;
    double *resid, *u_out;
    unsigned long long *fallback_count;
    unsigned long long *stamps;     // diagnostic build only (GPFQ_BLK_STAMPS): per-phase shader cycles of two wavefronts
    // Symmetric alphabets {-a, 0, a} and {-a, a} (the reference's default, bits = log2(3): quantize_pretrained_mlp.py:40): every
    // decision is q = sg * a32 with sg in {-1, 0, 1}, so f32(q * xq) = sg * f32(a32 * xq) EXACTLY and the update's
    // f32(w x) - f32(q xq) is ONE fused multiply-add on the pre-scaled row: fma(-sg, f32(a32 xq), f32(w x)) rounds once, the
    // subtraction's rounding.  The pre-pass stores f32(a32 * Xq) in place of Xq and the decisions publish -sg in place of q:
    // 8 instead of 12 packed float32 instructions per sample pair of four neurons.  0: the general form.
    // (DevAlphabet::sym_a; the SYM instantiations.)
    const float *Xq;                // (symmetric form's slow path)
    int64_t ldx;
    // The alphabet as an arithmetic progression (blk_uniform; DevAlphabet::a0, step, inv, c0): member k = a0 + k step in float64, whose
    // float32 rounding is float32(alphabet[k]) for every k (checked where the DevAlphabet is formed); inv = 1 / step, c0 = -a0 / step
    // (0, 0 for a single member).  The chain of decisions finds its candidate index and value by arithmetic; the certification looks the
    // true members up.
    // ... up to one float32 ulp: alphabet[k] = rad (2k - M + 1) / (M - 1) is a small rational multiple of a float32 median, and such
    // values sit EXACTLY on float32 rounding ties a few per cent of the time -- the float64 product rad * linspace[k] and the fused
    // a0 + k step then round to different neighbours (2 of 16 members of a typical 4-bit alphabet).  Bit k of DevAlphabet::plus / minus:
    // float32(alphabet[k]) is the next float32 above / below (in the integer order of the bit patterns) float32(a0 + k step).
    unsigned char pw[12];       // sample pairs per k-lane of sweep wavefront w (BlkSplit: a launch parameter)
    // Cluster form (round 5, rows beyond what one workgroup's registers hold; CL instantiations only): `nsl` workgroups -- the SLICES of a
    // cluster -- hold MP samples each of the same NB neurons; slice s streams its own records (recs + s slice_bytes: the same layout
    // over its samples, the headers' statistics and Gram band over the WHOLE row) and the decision wavefronts exchange their partial dot
    // products through `mbox` once per slot (cl_exchange), then take the same decisions from the same bits.  0: the classic form.
    int nsl;
    int cl_map;                 // 0: a cluster's slices in one XCD's queue; 1: consecutive workgroup ids (see gpfq_blk_kernel)
    int64_t slice_bytes;
    unsigned long long *mbox;   // [cluster][2][nsl][64 lanes][4] 8-byte words (payload32 | tag32 << 32), zeroed before the launch
    int64_t u_ld;               // row pitch of u_out (the whole row's sample count)
    double slack;               // float64 slack of a predicted dot product, relative: 2^-43 per 1024 samples of a row
    int *cl_err;                // set when an exchange timed out (a slice of the cluster never arrived): results are then invalid
    unsigned long long cl_timeout;  // ... after this many ticks of s_memrealtime (100 MHz; 3 s unless the option blk_cluster_timeout_ms says otherwise)
    int cl_fault;               // tests (option blk_cluster_fault): slice 1 of cluster 0 never publishes
};

// Diagnostic builds only (GPFQ_DIAG="-DGPFQ_BLK_DIAG -DGPFQ_BLK_STAMPS ...", quantized_neural_networks_amd/build.py): gpfq_blk_diag.hpp holds
// the in-kernel phase stamps and the two A/B switches of round 4's matrix form.  The shipped translation unit compiles without it.
#ifdef GPFQ_BLK_DIAG
#include "gpfq_blk_diag.hpp"
#else
constexpr bool kNoMfmaD = false, kNoFused = false;
#define STAMP(var) do { } while (0)
#define STAMP_DO(...)
#endif

// ---- cluster form: the exchange between the decision wavefronts of a cluster's slices (CL instantiations) ----
// Every slice publishes KV float64 values per lane and reads all slices' back, summed in slice order -- the same bits in every slice, so the
// slices take the same decisions without another word passing between them.  A value travels as two 8-byte words (32 bits of payload, the
// exchange's sequence number above them: the LL protocol of the collectives libraries), so a word is valid or visibly stale by itself and no
// fence orders anything; agent-scope accesses (sc1) meet in the memory side of the L2s.  Two buffers by the parity of the sequence
// number: a slice publishes exchange k + 1 only after it has read everybody's k, i.e. after everybody has read k - 1.  The slices of a
// cluster are co-resident by construction of the grid (gpfq_blk_kernel); should one never arrive, the wait gives up after ~3 s of
// s_memrealtime, raises BlkK::cl_err and the launch runs to its end on garbage instead of hanging the device.
struct ClState { int64_t cl; int slice; unsigned seq; bool dead; int64_t rec_off; int m_sl; };   // rec_off: the slice's record stream; m_sl: its samples of the row
typedef unsigned cl_u32x4 __attribute__((ext_vector_type(4)));

// Loads of the exchange: 16 bytes per lane, issued back to back and waited for together; device scope (sc1), like the stores -- right
// wherever the slices of a cluster run.  (Under map 0 they sit in ONE XCD, whose L2 every one of their stores passes through, so polling
// that L2 would do there -- but sc0 loads hit the CU's own cache and never see the data, with or without `buffer_inv sc0`, and behind
// `buffer_inv sc1` they cost more than they save: 40.2 against 28.1 ms at 4096 x 4096 on 8192 samples, profiles/r05/cluster_form.txt;
// an sc1 store drops its line from the L2s anyway, MI355X_MICROARCH.md.)
__device__ __forceinline__ cl_u32x4 cl_load16(const unsigned long long *p)
{
    cl_u32x4 r;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(r) : "v"(p) : "memory");
    return r;
}

// publish: the lane's KV values of exchange cs.seq + 1 (which this call makes the current one)
template <int KV>
__device__ __forceinline__ void cl_publish(const BlkK &K, ClState &cs, int lane, const double (&v)[KV])
{
    const unsigned seq = ++cs.seq;
    if (K.cl_fault && cs.cl == 0 && cs.slice == 1) return;          // (tests: the slice that never arrives)
    unsigned long long *buf = K.mbox + ((cs.cl * 2 + (int64_t)(seq & 1u)) * (int64_t)K.nsl) * 256;       // [slice][lane][4]
    unsigned long long *mine = buf + ((int64_t)cs.slice * 64 + lane) * 4;
#pragma unroll
    for (int k = 0; k < KV; ++k) {
        const cl_u32x4 w = {(unsigned)__double2loint(v[k]), seq, (unsigned)__double2hiint(v[k]), seq};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(mine + 2 * k), "v"(w) : "memory");
    }
}

// gather: every slice's values of the current exchange, added in slice order (CLB slices per batch of loads)
template <int KV, int CLB>
__device__ __forceinline__ void cl_gather(const BlkK &K, ClState &cs, int lane, double (&v)[KV])
{
    const unsigned seq = cs.seq;
    const unsigned long long *buf = K.mbox + ((cs.cl * 2 + (int64_t)(seq & 1u)) * (int64_t)K.nsl) * 256;
    double tot[KV];
#pragma unroll
    for (int k = 0; k < KV; ++k) tot[k] = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int s0 = 0; s0 < K.nsl; s0 += CLB) {
        cl_u32x4 w[CLB][KV];
        for (;;) {
#pragma unroll
            for (int i = 0; i < CLB; ++i) {
                const unsigned long long *p = buf + ((int64_t)(s0 + i < K.nsl ? s0 + i : cs.slice) * 64 + lane) * 4;    // (beyond the cluster: the own words, ignored below)
#pragma unroll
                for (int k = 0; k < KV; ++k) w[i][k] = cl_load16(p + 2 * k);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool ok = true;
#pragma unroll
            for (int i = 0; i < CLB; ++i)
#pragma unroll
                for (int k = 0; k < KV; ++k) {
                    asm volatile("" : "+v"(w[i][k]));               // (the loads have landed: nothing below moves above the wait)
                    ok &= (w[i][k].y == seq) & (w[i][k].w == seq);
                }
            if (__ballot(!ok) == 0ull || cs.dead) break;
            if (__builtin_amdgcn_s_memrealtime() - t0 > K.cl_timeout) {
                cs.dead = true;
                if (lane == 0 && K.cl_err) __hip_atomic_store(K.cl_err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int i = 0; i < CLB; ++i)
#pragma unroll
            for (int k = 0; k < KV; ++k) {
                const double x = __hiloint2double((int)w[i][k].z, (int)w[i][k].x);
                tot[k] += (s0 + i < K.nsl) ? x : 0.0;
            }
    }
#pragma unroll
    for (int k = 0; k < KV; ++k) v[k] = tot[k];
}

template <int KV, int CLB>
__device__ __forceinline__ void cl_exchange(const BlkK &K, ClState &cs, int lane, double (&v)[KV])
{
    cl_publish<KV>(K, cs, lane, v);
    cl_gather<KV, CLB>(K, cs, lane, v);
}

// Sample-pair split over the sweep wavefronts: PairSplit for eight of them (the decision wavefront, wavefront 8, shares
// SIMD 0 with wavefronts 0 and 4); with eleven the twelve wavefronts of a workgroup are three per SIMD.
template <int S, int NSW> struct BlkSplit { static constexpr const int *pw = PairSplit<S>::pw; };
template <> struct BlkSplit<4, 4>   { static constexpr int pw_[4] = {1, 1, 1, 1}; static constexpr const int *pw = pw_; };   // rows of up to 512 samples, four sweep wavefronts
// (eleven sweep wavefronts: wavefronts w, w + 4, w + 8 share SIMD w, the decision wavefront is the third of SIMD 3, whose sweep
//  wavefronts 3 and 7 therefore hold five pairs where the other SIMDs hold nine)
template <> struct BlkSplit<32, 11> { static constexpr int pw_[11] = {3, 3, 3, 3, 3, 3, 3, 2, 3, 3, 3}; static constexpr const int *pw = pw_; };
template <> struct BlkSplit<24, 11> { static constexpr int pw_[11] = {2, 2, 2, 2, 2, 2, 2, 1, 3, 3, 3}; static constexpr const int *pw = pw_; };

// (the eleven-wavefront 48- and 64-pair tables come from measured sweeps, tools/split_sweep.sh: DESIGN_HISTORY.md, "gpfq_blk.hip notes" 2)
template <> struct BlkSplit<64, 11> { static constexpr int pw_[11] = {4, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6}; static constexpr const int *pw = pw_; };
template <> struct BlkSplit<48, 11> { static constexpr int pw_[11] = {4, 5, 4, 4, 5, 4, 4, 5, 4, 4, 5}; static constexpr const int *pw = pw_; };
template <> struct BlkSplit<16, 11> { static constexpr int pw_[11] = {2, 2, 1, 1, 2, 2, 1, 1, 1, 1, 2}; static constexpr const int *pw = pw_; };

// The four-group narrow shapes on rows of 769..1024 samples (G = 4, one or two neurons per lane, eight sweep wavefronts): their slot is the
// chain of decisions, and every pair on the two sweep wavefronts that share the decision wavefront's SIMD lengthens it: 2,5,5,4,2,5,5,4 is
// the measured optimum (1.42 against 1.56-1.72 ms for the other splits at 4096 x 512; four sweep wavefronts, or the decision wavefront
// alone on its SIMD, are slower: DESIGN_HISTORY.md, "gpfq_blk.hip notes" 3).
struct BlkSplitQuad32 { static constexpr int pw[8] = {2, 5, 5, 4, 2, 5, 5, 4}; };
// (four neurons per workgroup on rows of 1537..2048 samples, two steps per slot -- cfg3's predictions layer, 4096 x 1000 on 2048 samples:
//  1,2,2,3,2,2,2,2 2.70-2.75 ms against PairSplit<16>'s 2.81-2.87; the other splits tried there were slower than either)
struct BlkSplitFour16 { static constexpr int pw[8] = {1, 2, 2, 3, 2, 2, 2, 2}; };
// Round 5: the four-group narrow shapes on rows of at most 768 samples take SEVEN sweep wavefronts -- eight wavefronts per workgroup, two per
// SIMD, 256 registers per wavefront; the decision wavefront, wavefront 7, shares SIMD 3 with sweep wavefront 3, which holds the fewest pairs.
// 4096 x 1024 on 768 samples 1.39 -> 1.20-1.22 ms, on 512 samples 1.26 -> 1.19, cfg1's Dense(784 -> 128) 0.267 -> 0.239; rows of 769..1024
// samples keep eight (their five-pair wavefronts at two per SIMD become the slot: 1.39 / 1.40-1.44 ms at 4096 x 512, 1.93 / 2.08 at 4096 x
// 2048): profiles/r05/quad_seven_wavefronts_ab.txt.
template <int S> struct BlkSplit7;
template <> struct BlkSplit7<32> { static constexpr int pw[7] = {5, 5, 5, 2, 5, 5, 5}; };
template <> struct BlkSplit7<24> { static constexpr int pw[7] = {4, 4, 4, 1, 4, 4, 3}; };
template <> struct BlkSplit7<16> { static constexpr int pw[7] = {3, 3, 2, 1, 3, 2, 2}; };
template <int G, int S, int NSW, int NL> constexpr const int *blk_split()
{
    if constexpr (NSW == 7) return BlkSplit7<S>::pw;
    else
    if constexpr (G == 4 && NL < 4 && S == 32 && NSW == 8) return BlkSplitQuad32::pw;
    else if constexpr (G == 1 && NL == 4 && S == 16 && NSW == 8) return BlkSplitFour16::pw;
    else return BlkSplit<S, NSW>::pw;
}

template <int G, int S, int NSW, int NL> constexpr bool blk_split_has(int k)
{
    for (int w = 0; w < NSW; ++w)
        if (blk_split<G, S, NSW, NL>()[w] == k) return true;
    return false;
}

// ---- sweep wavefront -------------------------------------------------------------------------------
template <int G, int PW, int MP, int B, int NSW, bool SYM, int NL, int CLM>
__device__ __forceinline__ void blk_sweep_role(const BlkK &K, char *lds_generic, const BlkLds &L, int wave, int lane, int pbase, const ClState &cs)
{
    constexpr bool CL = CLM != 0;
    constexpr int NB = NL * G, KQ = 64 / G, NW = blk_slots(NSW, NB);   // NL neurons per lane (4; 2 in the narrow-layer shapes)
    constexpr int RSH = NL == 4 ? 0 : (NL == 2 ? 1 : 2);          // folded sums: neuron i of the lane ends up in rows i << RSH .. of the wavefront
    constexpr int HDR = blk_hdr_bytes(B, CL);
    constexpr int RB = (int)blk_rec_bytes(MP, B, G, CL);
    lchar *lds = (lchar *)lds_generic;
    const int ng = lane & (G - 1), kq = lane / G, row = lane >> 4;
    const bool writer = (lane & 15 & ~(G - 1)) == 0 && (row & ((1 << RSH) - 1)) == 0;   // one lane per (neuron, ng) publishes the folded sums
    const int nloc = NL * ng;                                     // first of this lane's NL neurons
    const int nrow = row >> RSH;                                  // the neuron (of the lane's NL) whose folded sums this row holds
    const int64_t jbase = cs.cl * NB;                             // (classic form: the workgroup's neuron block, see gpfq_blk_kernel)
    const unsigned ldsT_addr = lds_addr(lds_generic), ldsW_addr = lds_addr(lds_generic + L.off_w);
    const int64_t N = K.N;
    const int nslots = K.nblk + 1;
    const int o_x  = HDR + 8 * (pbase + kq);                      // float2 x   [pair]  (row t - B of record t)
    int o_q = o_x + 4 * MP;                                       // float2 xq  [pair]
    // (opaque to the compiler: it would fuse the x and xq reads of a pair -- 4 MP bytes apart -- into ONE ds_read2st64_b64, which the
    //  LDS serves as two 32-bank accesses in 8 cycles; two ds_read_b64 take 2 cycles each: MI355X_MICROARCH.md, LDS)
    asm volatile("" : "+v"(o_q));
    constexpr int DB = blk_row64(G, B) ? 16 : 8;                  // bytes of a sample pair of row t + B
    const int o_d  = HDR + 8 * MP + DB * (pbase + kq);           // double2 (float2) xqd[pair]  (row t + B)
    using DRaw = std::conditional_t<blk_row64(G, B), double2, float2>;   // as it sits in the record; converted where it is consumed
    auto ld_d = [&](int off) -> DRaw { return lds_ld<DRaw>(lds, off); };
    auto to_d2 = [](const DRaw &v) -> double2 { return make_double2((double)v.x, (double)v.y); };
    const int o_wq = L.off_wq + nloc * B * 8;
    const int o_dw = L.off_d + (wave * B * NB + nloc + nrow) * 8;
    const int o_x2 = L.off_x2 + (wave * NB + nloc + nrow) * 16;

    double u[NL][2 * PW];
#pragma unroll
    for (int i = 0; i < NL; ++i)
#pragma unroll
        for (int e = 0; e < 2 * PW; ++e) u[i][e] = 0.0;          // zeros(m), :115

    // The record tile of the next slot streams into the other LDS buffer in 1 KiB LDS-DMA pieces, piece k of this wavefront
    // being piece wave + 8 k of the tile (the last piece may run into the next record: harmless).  The pieces are issued ONE
    // AT A TIME from inside phase U (two issue points per step), with the base address stepped in scalar registers: a
    // wavefront that issues its eight or nine pieces back to back waits ~1000 cycles per slot for the vector-memory queue
    // (the texture addresser moves 64 bytes per clock: 66 KiB per slot), and so does the wavefront it shares its SIMD with --
    // spread out, the queue never fills and the transfers hide under the arithmetic (measured: 4.86 -> 4.51 ms at
    // 4096 x 4096 x 1024, and with the decision wavefront's priority raised 4.17).
    constexpr int NPIECES = (B * RB + 1023) >> 10;
    constexpr int PER_MIN = NPIECES / NSW;               // every wavefront has at least this many pieces per tile
    constexpr int PTS = PW >= 2 ? 2 : 1;                          // issue points per step of phase U (its first pairs)
    constexpr int PPP = PER_MIN / (PTS * B) > 0 ? PER_MIN / (PTS * B) : 1;   // pieces per issue point
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_piece = [&](int b1, int k) {
        const int pc = wave + NSW * k;
        glds16_s(K.recs + (CL ? cs.rec_off : (int64_t)0) + (int64_t)b1 * L.tile_bytes + ((int64_t)pc << 10), lane16,
                 ldsT_addr + (unsigned)(b1 & 1) * (unsigned)L.tile_pitch + ((unsigned)pc << 10));
    };
    auto load_weights = [&](int b1) {
        const unsigned dw = ldsW_addr + (unsigned)((b1 & 1) * NB * B * 4);
        for (int i0 = wave * 64; i0 < NB * B; i0 += NSW * 64) {
            const int i = i0 + lane;
            const int n = i / B, s = i - n * B;
            const int64_t jn = jbase + n, t = (int64_t)b1 * B + s;
            if (i < NB * B && jn < K.C && t < N) glds4(K.Wt + jn * K.ldw + t * K.ldt, dw + 4 * (unsigned)i0);
        }
    };

    // The record headers once more, compact, two slots AHEAD into a ring of three: the decision wavefront reads tile b + 1's
    // at the end of slot b, when the LDS is quiet (see blk_decision_role).  One 1 KiB piece per wavefront (tile k -> buffer k % 3).
    constexpr int NHP = (B * HDR + 1023) >> 10;
    auto load_headers = [&](int b2, int buf) {
        if (wave < NHP && b2 <= K.nblk)
            glds16_s(K.hdrs + (int64_t)b2 * B * HDR + ((int64_t)wave << 10), lane16,
                     lds_addr(lds_generic + L.off_hr) + (unsigned)buf * (unsigned)L.hr_pitch + ((unsigned)wave << 10));
    };
    for (int k = 0; wave + NSW * k < NPIECES; ++k) issue_piece(0, k);
    load_weights(0);
    load_headers(0, 0); load_headers(1, 1);
    dma_wait();
    slot_barrier();
    int hbuf = 2;                                                 // (b + 2) % 3
    unsigned long long st0 = 0, st1 = 0, st2 = 0, st3 = 0, st4 = 0, st5 = 0, acc_dma = 0, acc_u = 0, acc_d = 0, acc_w = 0, acc_b = 0, acc_t = 0;
    (void)st0; (void)st1; (void)st2; (void)st3; (void)st4; (void)st5; (void)acc_dma; (void)acc_u; (void)acc_d; (void)acc_w; (void)acc_b; (void)acc_t;

    // Few pairs per slot (kPreloadAll): EVERY operand of a slot -- the rows of the B updates, the block's (w, q), the rows of the B
    // dot products -- is requested right after the barrier that ends the slot before, together with the control word of the slow
    // path, and held in registers: one LDS round trip per slot instead of three in sequence (control word, phase U's operands, phase
    // D's; 400-900 cycles each with the LDS busiest right after the barrier -- profiles/r03/blk_phase_stamps.txt).
    // Phase D on the matrix unit (round 4; G = 4 neuron groups x B = 4 steps x NL = 4 neurons per lane, row t + B as float64), and with it
    // the two phases FUSED pair by pair (kFused): see the slot loop.
    // (NL = 4, 2 or 1 neurons per lane: 16, 8 or 4 per workgroup; four steps per slot only -- with two, half of every 4 x 4 x 4 block idles and
    //  the form measured slower than the one- / two-group shapes: DESIGN_HISTORY.md, "gpfq_blk.hip notes" 4)
    constexpr bool kMfmaD = G == 4 && B == 4 && blk_row64(G, B) && !kNoMfmaD;
    constexpr bool kFused = kMfmaD && !kNoFused;
    constexpr bool kPreloadAll = PW * B <= 8 && PW <= 5 && !kFused;
    // ... where the registers allow it (u and the operands of a slot together): otherwise they are requested at the top of
    // phase U as in round 2
    constexpr bool kHoist = kPreloadAll && PW * B <= 4;
    constexpr int PB = kPreloadAll ? B : 1, PP = kPreloadAll ? PW : 1;
    // (the rows of the dot products too when they are few registers: otherwise phase D requests them itself, as before)
    constexpr bool kPreD = kHoist && PW * B * (blk_row64(G, B) ? 4 : 2) <= 16 && !kMfmaD;
    constexpr int DBn = kPreD ? B : 1, DPn = kPreD ? PW : 1;
    float2 xs[PB][PP], qs[PB][PP], wqa[PB][NL];
    DRaw ds[DBn][DPn];
    auto preload_wq = [&](int b) {                                // (w, q) of block b - 1, for slot b's updates
        const int pbq = ((b - 1) & 1) * NB * B * 8;
#pragma unroll
        for (int s = 0; s < PB; ++s)
#pragma unroll
            for (int n = 0; n < NL; ++n) wqa[s][n] = lds_ld<float2>(lds, o_wq + pbq + (n * B + s) * 8);
    };
    auto preload_rows = [&](int b) {
        const int tbase = (b & 1) * L.tile_pitch;
#pragma unroll
        for (int s = 0; s < PB; ++s)
#pragma unroll
            for (int p = 0; p < PP; ++p) {
                xs[s][p] = lds_ld<float2>(lds, tbase + s * RB + o_x + 8 * p * KQ);
                qs[s][p] = lds_ld<float2>(lds, tbase + s * RB + o_q + 8 * p * KQ);
                if constexpr (kPreD) ds[s][p] = ld_d(tbase + s * RB + o_d + DB * p * KQ);
            }
    };
    if constexpr (kHoist) { preload_wq(0); preload_rows(0); }

    // The slow path of block bb (rare; after the barrier that ends slot bb): while any neuron of the block is stopped at step S, the exact dot
    // products of step bb B + S on the residual as it stands BEFORE block bb.
    auto slow_path = [&](int b, int ctl) {
        for (;;) {
            const int S = __builtin_amdgcn_readfirstlane(ctl);
            if (S < 0) break;
            // u is the residual BEFORE block b.  Exact <Xq_t, u_{t-1}> and <Xq_t, u_{t-1} + f32(w_t X_t)> (:86, :89) for
            // t = bB + S: the block's first S updates replayed into temporaries; rows of block b are records of tile b+1.
            const int nb_ = ((b + 1) & 1) * L.tile_pitch;
            const int cbq = (b & 1) * NB * B * 8;
            double eu[NL], ew[NL];
#pragma unroll
            for (int n = 0; n < NL; ++n) { eu[n] = 0.0; ew[n] = 0.0; }
            // (one neuron and one pair at a time: the slow path must not cost the hot loop registers)
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                const float w = lds_ld<float2>(lds, o_wq + cbq + (n * B + S) * 8).x;
#pragma unroll 1
                for (int p = 0; p < PW; ++p) {
                    double t0 = u[n][0], t1 = u[n][1];
#pragma unroll
                    for (int pp = 1; pp < PW; ++pp) { t0 = (p == pp) ? u[n][2 * pp] : t0; t1 = (p == pp) ? u[n][2 * pp + 1] : t1; }
                    for (int j = 0; j < S; ++j) {
                        const int rb = nb_ + j * RB;
                        const float2 x2 = lds_ld<float2>(lds, rb + o_x + 8 * p * KQ), q2 = lds_ld<float2>(lds, rb + o_q + 8 * p * KQ);
                        const float2 wq = lds_ld<float2>(lds, o_wq + cbq + (n * B + j) * 8);
                        if constexpr (SYM) {
                            t0 += (double)__fmaf_rn(wq.y, q2.x, __fmul_rn(wq.x, x2.x));
                            t1 += (double)__fmaf_rn(wq.y, q2.y, __fmul_rn(wq.x, x2.y));
                        } else {
                            t0 += (double)__fsub_rn(__fmul_rn(wq.x, x2.x), __fmul_rn(wq.y, q2.x));
                            t1 += (double)__fsub_rn(__fmul_rn(wq.x, x2.y), __fmul_rn(wq.y, q2.y));
                        }
                    }
                    const int rb = nb_ + S * RB;
                    const float2 x2 = lds_ld<float2>(lds, rb + o_x + 8 * p * KQ);
                    float2 q2;
                    if constexpr (SYM) {                              // the record holds a32 * Xq_t: the row itself from memory (rare path)
                        const int i0 = 2 * (pbase + p * KQ + kq);
                        const float *xr = K.Xq + ((int64_t)b * B + S) * K.ldx + (CL ? (int64_t)cs.slice * MP : (int64_t)0);   // (cluster form: the slice's samples)
                        const int mrow = CL ? cs.m_sl : K.m;
                        q2.x = i0 < mrow ? xr[i0] : 0.f;
                        q2.y = i0 + 1 < mrow ? xr[i0 + 1] : 0.f;
                    } else {
                        q2 = lds_ld<float2>(lds, rb + o_q + 8 * p * KQ);
                    }
                    eu[n] = fma((double)q2.x, t0, eu[n]);
                    eu[n] = fma((double)q2.y, t1, eu[n]);
                    ew[n] = fma((double)q2.x, t0 + (double)__fmul_rn(w, x2.x), ew[n]);
                    ew[n] = fma((double)q2.y, t1 + (double)__fmul_rn(w, x2.y), ew[n]);
                }
            }
            const double vu = fold_klanes_n<G, NL>(eu), vw = fold_klanes_n<G, NL>(ew);
            if (writer) lds_st<double2>(lds, o_x2, make_double2(vu, vw));
            slot_barrier();                                       // partials published
            slot_barrier();                                       // chains resumed, control word rewritten
            ctl = lds_ld<int>(lds, L.off_ctl + 4 * (b & 1));
        }
    };
    for (int b = 0; b < nslots; ++b) {
        STAMP(st0);
        STAMP_DO(if (b) acc_t += st0 - st5;)   // from the barrier to the top of the next slot: control word, next operands
        // (blk_sweep_flush shapes) block b - 1's outputs -- final since the barrier, slow path included -- go from the LDS ring to memory
        // here, by one sweep wavefront per slot in turn: lane = (neuron, step), a byte and a float each
        if (blk_sweep_flush<G, NL, CL>() && b >= 1 && wave == b % NSW && lane < NB * B && (!CL || cs.slice == 0)) {
            const int nn = lane / B, sidx = lane % B;
            const int64_t t = (int64_t)(b - 1) * B + sidx, jn = jbase + nn;
            if (t < N && jn < K.C) {
                const int2 v = lds_ld<int2>(lds, L.off_out + (nn * kOutSteps + (int)(t % kOutSteps)) * 8);
                if (K.qidx) K.qidx[jn * N + t] = (int8_t)v.x;
                if (K.Qt) K.Qt[jn * N + t] = __int_as_float(v.y);
            }
        }
        if (b + 1 < nslots) load_weights(b + 1);                  // the block's weights: one piece
        load_headers(b + 2, hbuf); hbuf = hbuf == 2 ? 0 : hbuf + 1;
        const int bn = b + 1 < nslots ? b + 1 : b;                // (the last slot rewrites its own tile with the same bytes)
        STAMP(st1);
        const int tbase = (b & 1) * L.tile_pitch;
        const int pbq = ((b - 1) & 1) * NB * B * 8;               // (w, q) of block b-1

        // one sample pair of one step: f32 products and subtraction on two samples at once (v_pk_mul_f32 / v_pk_add_f32 or, for
        // symmetric alphabets, v_pk_fma_f32), conversions, float64 additions
        auto update_pair = [&](int p, const float2 &x2, const float2 &q2, const float (&wv)[NL], const float (&qv)[NL]) {
            const pk2 xv = {x2.x, x2.y}, qx = {q2.x, q2.y};
            // f32 products and subtraction on two samples at once (v_pk_mul_f32 / v_pk_add_f32): each half rounds
            // exactly as the scalar instruction (no contraction: -ffp-contract=off).  Written stage by stage over the
            // four neurons so that no instruction consumes the result of the one right before it (hipcc pads those
            // packed-to-scalar dependences with s_nop, which cost issue slots)
            pk2 pr[NL], rr[NL], dd[NL];
            double c0[NL], c1[NL];
#pragma unroll
            for (int n = 0; n < NL; ++n) pr[n] = pk2{wv[n], wv[n]} * xv;
            if constexpr (SYM) {                      // qv = -sg, qx = f32(a xq): v_pk_fma_f32, one rounding
#pragma unroll
                for (int n = 0; n < NL; ++n) dd[n] = __builtin_elementwise_fma(pk2{qv[n], qv[n]}, qx, pr[n]);
            } else {
#pragma unroll
                for (int n = 0; n < NL; ++n) rr[n] = pk2{qv[n], qv[n]} * qx;
#pragma unroll
                for (int n = 0; n < NL; ++n) dd[n] = pr[n] - rr[n];
            }
#pragma unroll
            for (int n = 0; n < NL; ++n) { c0[n] = (double)dd[n].x; c1[n] = (double)dd[n].y; }
#pragma unroll
            for (int n = 0; n < NL; ++n) { u[n][2 * p] += c0[n]; u[n][2 * p + 1] += c1[n]; }
        };
        // ---- phase U: the B updates of block b-1, in order: u += f32(w x) - f32(q xq)  (:119) ----
        // Software-pipelined by hand: the operands of the NEXT pair (and the next step's four (w, q)) are requested before
        // the arithmetic of the current one (sched_barrier keeps hipcc from sinking the requests to their first use,
        // where every pair would wait out a full LDS round trip).
        if constexpr (kFused) {
            // ---- round 4: updates and dot products fused, PAIR by pair ----
            // For each sample pair p: the B updates of block b-1 on u[.][2p], u[.][2p+1] (element-wise flow, in step order), then that
            // pair's share of the B x 16 dot products of block b+1 on the matrix unit (see the kMfmaD branch of phase D below for
            // the lane maps) -- issued two per step UNDER the next pair's updates, so phase D no longer exists as a phase: as one
            // it cost a slot an LDS round trip nothing hid, a row of matrix instructions issued at the lowest priority while the
            // other wavefronts of the SIMD were still updating, and the fold behind it (profiles/r04/blk_phase_stamps.txt: 980 -
            // 2300 cycles per slot for 640 - 1150 cycles of matrix work).  The block's sixteen (w, q) are read once per slot.
            // (The slot's first requests stay BEHIND the wait for the slow path's control word: hoisted in front of it -- with the slow
            //  path inlined, round 4, or as an out-of-line call without a spill, round 5 -- the kernel is 10-20 % slower:
            //  DESIGN_HISTORY.md, "gpfq_blk.hip notes" 5; profiles/r05/blk_variants_ab.txt.)
            const int rbm = tbase + ng * RB + o_d;
            // Operand rows are requested PF pair-steps ahead of their use: one everywhere -- except in the seven-sweep-wavefront shapes (two
            // wavefronts per SIMD, 256 registers), where a pair-step's handful of instructions no longer covers an LDS round trip and the
            // sweep's own chain of them would become the slot: two (round 5).
            // (two and four pair-steps ahead in the eight-wavefront narrow shapes too: no change, 1.28-1.30 ms at 4096 x 512 on 1024 samples
            //  either way -- profiles/r05/cluster_form.txt)
            constexpr int PF = NSW == 7 ? 2 : 1, NPS = PW * B;
            float2 fwq[B][NL], xbuf[PF], qbuf[PF];
            double2 dcur;
            auto row_off = [&](int i, bool q) { return tbase + (i % B) * RB + (q ? o_q : o_x) + 8 * (i / B) * KQ; };   // pair-step i = pair * B + step
            auto first_requests = [&]() {
#pragma unroll
                for (int s = 0; s < B; ++s)
#pragma unroll
                    for (int n = 0; n < NL; ++n) fwq[s][n] = lds_ld<float2>(lds, o_wq + pbq + (n * B + s) * 8);
#pragma unroll
                for (int i = 0; i < PF; ++i)
                    if (i < NPS) { xbuf[i] = lds_ld<float2>(lds, row_off(i, false)); qbuf[i] = lds_ld<float2>(lds, row_off(i, true)); }
                dcur = lds_ld<double2>(lds, rbm);
            };
            first_requests();
            double2 dprev = make_double2(0.0, 0.0);
            double acc[NL];
#pragma unroll
            for (int n = 0; n < NL; ++n) acc[n] = 0.0;
            auto mfma_pair = [&](int pp, int i, const double2 &dd) {      // matrix instruction i = 0 .. 2 NL - 1 of pair pp
                const int n = i % NL, e = i / NL;
                acc[n] = __builtin_amdgcn_mfma_f64_4x4x4f64(u[n][2 * pp + e], e ? dd.y : dd.x, acc[n], 0, 0, 0);
            };
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                // issue priority falls with progress through the slot (the laggard of a SIMD is served first, see below)
                {
                    // (priorities by wavefront AGE instead of progress measured slower: DESIGN_HISTORY.md, "gpfq_blk.hip notes" 6)
                    const int pr = 2 - (3 * p) / PW;
                    if (p == 0 || pr != 2 - (3 * (p - 1)) / PW) {
                        const int v = pr;
                        if (v >= 2) __builtin_amdgcn_s_setprio(2);
                        else if (v == 1) __builtin_amdgcn_s_setprio(1);
                        else __builtin_amdgcn_s_setprio(0);
                    }
                }
                if (p > 0) dcur = lds_ld<double2>(lds, rbm + DB * p * KQ);       // (consumed under the NEXT pair's updates, or at the end)
#pragma unroll
                for (int st = 0; st < B; ++st) {
                    const int ips = p * B + st;                                  // this pair-step; its operands sit in buffer ips % PF
                    const float2 x2 = xbuf[ips % PF], q2 = qbuf[ips % PF];
                    const int pn = (ips + 1) / B, sn = (ips + 1) % B;            // (the GPFQ_BLK_X_LDS2 experiment re-reads the next one)
                    (void)pn; (void)sn;
                    if (ips + PF < NPS) {
                        xbuf[ips % PF] = lds_ld<float2>(lds, row_off(ips + PF, false));
                        qbuf[ips % PF] = lds_ld<float2>(lds, row_off(ips + PF, true));
                    }
                    if (p < PTS) {
#pragma unroll
                        for (int i = 0; i < PPP; ++i) {
                            const int k = (PTS * st + p) * PPP + i;
                            if (PTS * B * PPP <= PER_MIN || k < PER_MIN) issue_piece(bn, k);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    float wv[NL], qv[NL];
#pragma unroll
                    for (int n = 0; n < NL; ++n) { wv[n] = fwq[st][n].x; qv[n] = fwq[st][n].y; }
                    update_pair(p, x2, q2, wv, qv);
                    if (p > 0) {                                   // the previous pair's 2 NL matrix instructions, spread over this pair's B steps
#pragma unroll
                        for (int i = 0; i < 2 * NL; ++i)
                            if (i * B / (2 * NL) == st) mfma_pair(p - 1, i, dprev);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                dprev = dcur;
            }
            for (int k = PTS * B * PPP < PER_MIN ? PTS * B * PPP : PER_MIN; wave + NSW * k < NPIECES; ++k) issue_piece(bn, k);   // the rest
            STAMP(st2);
#pragma unroll
            for (int i = 0; i < 2 * NL; ++i) mfma_pair(PW - 1, i, dprev);
#pragma unroll
            for (int n = 0; n < NL; ++n) { acc[n] = ror_add<8>(acc[n]); acc[n] = ror_add<4>(acc[n]); }   // blocks: lane bits 2, 3
            if (b + 1 < nslots && (lane & 12) == 0) {             // one lane per (step, neuron group)
                const int od = L.off_d + ((((((b + 1) & 1) * NW + wave) * B) + (lane & 3)) * NB + NL * (lane >> 4)) * 8;
                if constexpr (NL == 4) {
                    lds_st<double2>(lds, od, make_double2(acc[0], acc[1]));
                    lds_st<double2>(lds, od + 16, make_double2(acc[2], acc[3]));
                } else if constexpr (NL == 2) {
                    lds_st<double2>(lds, od, make_double2(acc[0], acc[NL - 1]));
                } else {
                    lds_st<double>(lds, od, acc[0]);
                }
            }
        } else {
        if constexpr (kPreloadAll) {
            // One or two pairs per step: the pipelined loop below keeps ONE pair of operands in flight, and a step's arithmetic
            // (32 instructions per pair) is shorter than an LDS round trip with twelve wavefronts on the LDS -- the sweep of a
            // 4-neuron workgroup took 600 cycles per (pair, step) against 316 for the arithmetic.  So every operand of the slot is
            // requested up front (at most 8 pairs + the block's (w, q): 64 registers) and the steps run back to back.
            // (operands: requested after the last barrier where the registers allow it, see preload_rows)
            if constexpr (!kHoist) { preload_wq(b); preload_rows(b); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int s = 0; s < B; ++s) {
                if (s == 0) __builtin_amdgcn_s_setprio(2);
                if (s == B / 2) __builtin_amdgcn_s_setprio(1);
                float wv[NL], qv[NL];
#pragma unroll
                for (int n = 0; n < NL; ++n) { wv[n] = wqa[s][n].x; qv[n] = wqa[s][n].y; }
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    if (p < PTS) {
#pragma unroll
                        for (int i = 0; i < PPP; ++i) {
                            const int k = (PTS * s + p) * PPP + i;
                            if (PTS * B * PPP <= PER_MIN || k < PER_MIN) issue_piece(bn, k);
                        }
                    }
                    update_pair(p, xs[s][p], qs[s][p], wv, qv);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else
        {
            float2 x2n = lds_ld<float2>(lds, tbase + o_x), q2n = lds_ld<float2>(lds, tbase + o_q);
            float2 wqn[NL];
#pragma unroll
            for (int n = 0; n < NL; ++n) wqn[n] = lds_ld<float2>(lds, o_wq + pbq + (n * B) * 8);
            // two steps per trip for the B = 4 shapes: the rotation of the prefetched (w, q) of the next step is then a renaming
            // instead of eight v_mov per step (3.76 -> 3.64 ms at 4096 x 4096 x 1024); the B = 2 shapes would unroll completely
            // and run out of registers
#pragma clang loop unroll_count(B == 4 ? 2 : 1)
            for (int s = 0; s < B; ++s) {
                // Issue priority falls with progress through the slot (2, 1, 1, 0 over the halves of the two phases; the decision
                // wavefront keeps 3).  At equal priority the SIMD serves its OLDEST ready wavefront first: the first-launched
                // sweep ran at full single-wavefront rate, finished early and left the youngest to run the tail of the slot alone,
                // at half the issue rate (profiles/r02/blk_phase_stamps.txt: phase U of equal shares took 3440 / 5250 / 7200
                // cycles by age).  With the laggard preferred the wavefronts of a SIMD reach the barrier together.
                if (s == 0) __builtin_amdgcn_s_setprio(2);
                if (s == B / 2) __builtin_amdgcn_s_setprio(1);
                float wv[NL], qv[NL];
#pragma unroll
                for (int n = 0; n < NL; ++n) { wv[n] = wqn[n].x; qv[n] = wqn[n].y; }
                const int rb = tbase + s * RB;
                const int sn = s + 1 < B ? s + 1 : s;             // (the last step re-requests its own: harmless)
#pragma unroll
                for (int n = 0; n < NL; ++n) wqn[n] = lds_ld<float2>(lds, o_wq + pbq + (n * B + sn) * 8);
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    const float2 x2 = x2n, q2 = q2n;
                    const int rbn = p + 1 < PW ? rb : tbase + sn * RB, pn = p + 1 < PW ? p + 1 : 0;
                    x2n = lds_ld<float2>(lds, rbn + o_x + 8 * pn * KQ);
                    q2n = lds_ld<float2>(lds, rbn + o_q + 8 * pn * KQ);
                    if (p < PTS) {
#pragma unroll
                        for (int i = 0; i < PPP; ++i) {
                            const int k = (PTS * s + p) * PPP + i;
                            if (PTS * B * PPP <= PER_MIN || k < PER_MIN) issue_piece(bn, k);   // (short tiles: fewer pieces than issue points)
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    update_pair(p, x2, q2, wv, qv);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        for (int k = PTS * B * PPP < PER_MIN ? PTS * B * PPP : PER_MIN; wave + NSW * k < NPIECES; ++k) issue_piece(bn, k);   // the rest
        STAMP(st2);
        // ---- phase D: this wavefront's share of <Xq_t, u> for the B rows of block b+1 ----
        if constexpr (kMfmaD) {
            // Round 4: the B x 16 dot products of a slot are a (steps x samples) . (samples x neurons) product, and
            // v_mfma_f64_4x4x4_4b_f64 forms four independent 4 x 4 x 4 blocks of it per issue: D_b[i][j] += sum_k A_b[i][k] B_b[k][j], lane
            // maps (tools/ubench/mfma_f64_4x4x4.hip, profiles/r04/ubench_mfma_f64_4x4x4.txt) A: i + 4 b + 16 k, B: j + 4 b + 16 k,
            // D: j + 4 b + 16 i.  With the sweep's lane = ng + 4 kq that is
            //   A = u[n][e] AS IT STANDS: i = ng (the lane's neuron group: neuron 4 i + n), block b = kq & 3, k = kq >> 2
            //       (the four k-lanes b, b + 4, b + 8, b + 12: the contraction runs over their samples),
            //   B = Xq_{step j}[the lane's own sample]: lane (ng, kq) reads the row of STEP ng instead of every step's
            //       (one ds_read_b128 per pair instead of four),
            //   D = one value per lane: step j = lane & 3, neuron 4 (lane >> 4) + n, partial over the k-lanes = b mod 4.
            // One issue does the 256 multiply-adds of four v_fma_f64 (16 cycles: the vector unit's own float64 rate -- the MFMA
            // occupies it, the ubench's costs add -- but ONE issue slot instead of four at the 5.6 cycles two wavefronts per SIMD
            // sustain), the sum over samples happens inside the instruction, and what is left of the k-lane fold (16 values over 16
            // lanes: 84 instructions per slot) is the sum over the four blocks: two row rotations per accumulator.
            // The order of the float64 additions differs from the vector form's (as that one's differs from BLAS ddot's): the
            // certification's 2^-43 term is the slack for exactly that, and the residual never sees these sums.
            if (b + 1 < nslots) {
                __builtin_amdgcn_s_setprio(0);
                double acc[NL];
#pragma unroll
                for (int n = 0; n < NL; ++n) acc[n] = 0.0;
                const int rbm = tbase + ng * RB + o_d;
                double2 d2n = lds_ld<double2>(lds, rbm);
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    const double2 d2 = d2n;
                    if (p + 1 < PW) d2n = lds_ld<double2>(lds, rbm + DB * (p + 1) * KQ);
#pragma unroll
                    for (int n = 0; n < NL; ++n) acc[n] = __builtin_amdgcn_mfma_f64_4x4x4f64(u[n][2 * p], d2.x, acc[n], 0, 0, 0);
#pragma unroll
                    for (int n = 0; n < NL; ++n) acc[n] = __builtin_amdgcn_mfma_f64_4x4x4f64(u[n][2 * p + 1], d2.y, acc[n], 0, 0, 0);
                }
#pragma unroll
                for (int n = 0; n < NL; ++n) { acc[n] = ror_add<8>(acc[n]); acc[n] = ror_add<4>(acc[n]); }   // blocks: lane bits 2, 3
                if ((lane & 12) == 0) {                           // one lane per (step, neuron group)
                    const int od = L.off_d + ((((((b + 1) & 1) * NW + wave) * B) + (lane & 3)) * NB + NL * (lane >> 4)) * 8;
#pragma unroll
                    for (int n = 0; n < NL; ++n) lds_st<double>(lds, od + 8 * n, acc[n]);
                }
            }
        } else
        if (b + 1 < nslots && kPreloadAll) {
            DRaw dsl[B][PW];
#pragma unroll
            for (int r = 0; r < B; ++r)
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    if constexpr (kPreD) dsl[r][p] = ds[r][p];
                    else dsl[r][p] = ld_d(tbase + r * RB + o_d + DB * p * KQ);
                }
            if constexpr (!kPreD) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < B; ++r) {
                if (r == B / 2) __builtin_amdgcn_s_setprio(0);
                double acc[NL];
#pragma unroll
                for (int n = 0; n < NL; ++n) acc[n] = 0.0;
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    const double2 d2 = to_d2(dsl[r][p]);
#pragma unroll
                    for (int n = 0; n < NL; ++n) {
                        acc[n] = fma(d2.x, u[n][2 * p], acc[n]);
                        acc[n] = fma(d2.y, u[n][2 * p + 1], acc[n]);
                    }
                }
                const double v = fold_klanes_n<G, NL>(acc);
                if (writer) lds_st<double>(lds, o_dw + ((((b + 1) & 1) * NW) * B + r) * NB * 8, v);
            }
        } else
        if (b + 1 < nslots) {
            DRaw d2n = ld_d(tbase + o_d);
#pragma clang loop unroll_count(B == 4 ? 2 : 1)
            for (int r = 0; r < B; ++r) {
                if (r == B / 2) __builtin_amdgcn_s_setprio(0);
                double acc[NL];
#pragma unroll
                for (int n = 0; n < NL; ++n) acc[n] = 0.0;
                const int rb = tbase + r * RB;
                const int rn = r + 1 < B ? r + 1 : r;
#pragma unroll
                for (int p = 0; p < PW; ++p) {
                    const double2 d2 = to_d2(d2n);
                    const int rbn = p + 1 < PW ? rb : tbase + rn * RB, pn = p + 1 < PW ? p + 1 : 0;
                    d2n = ld_d(rbn + o_d + DB * pn * KQ);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < NL; ++n) {
                        acc[n] = fma(d2.x, u[n][2 * p], acc[n]);
                        acc[n] = fma(d2.y, u[n][2 * p + 1], acc[n]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                const double v = fold_klanes_n<G, NL>(acc);
                if (writer) lds_st<double>(lds, o_dw + ((((b + 1) & 1) * NW) * B + r) * NB * 8, v);
            }
        }
        }   // (!kFused)
        STAMP(st3);
        dma_wait();                                               // this wavefront's share of the next tile has landed
        STAMP(st4);
        slot_barrier();
        STAMP(st5);
        STAMP_DO(acc_dma += st1 - st0; acc_u += st2 - st1; acc_d += st3 - st2; acc_w += st4 - st3; acc_b += st5 - st4;)

        // ---- slow path: neurons of block b stopped at an uncertifiable step (rare) ----
        {
        int ctl = lds_ld<int>(lds, L.off_ctl + 4 * (b & 1));      // (requested first: it is waited for alone)
        if constexpr (kHoist) {
            if (b + 1 < nslots) { preload_wq(b + 1); preload_rows(b + 1); }
        }
        if (__builtin_amdgcn_readfirstlane(ctl) >= 0) {
            slow_path(b, ctl);
            if constexpr (kHoist) {
                if (b + 1 < nslots) { preload_wq(b + 1); preload_rows(b + 1); }    // (the slow path has rewritten decisions of block b)
            }
        }
        }
    }

    STAMP_DO(
    if (K.stamps && blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 7)) {
        unsigned long long *o = K.stamps + (wave == 0 ? 0 : 8);
        o[0] = acc_dma; o[1] = acc_u; o[2] = acc_d; o[3] = acc_w; o[4] = acc_b; o[5] = (unsigned long long)nslots; o[6] = acc_t;
    }
    if (K.stamps && blockIdx.x == 0 && lane == 0) K.stamps[32 + wave] = acc_dma + acc_u + acc_d + acc_w;   // slot top -> arrival at the barrier, every sweep wavefront

    )
    // ---- epilogue: residual norms through the same partial-sum path, residual vectors straight to memory ----
    if (K.resid) {
        double ss[NL];
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            double q = 0.0;
#pragma unroll
            for (int e = 0; e < 2 * PW; ++e) q = fma(u[n][e], u[n][e], q);
            ss[n] = q;
        }
        const double v = fold_klanes_n<G, NL>(ss);
        if (writer) lds_st<double>(lds, o_dw, v);
    }
    slot_barrier();
    if (K.u_out) {
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int64_t jn = jbase + nloc + n;
            if (jn < K.C) {
#pragma unroll
                for (int p = 0; p < PW; ++p)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int i = 2 * (pbase + p * KQ + kq) + e;
                        if constexpr (CL) {
                            if (i < cs.m_sl) K.u_out[jn * K.u_ld + (int64_t)cs.slice * MP + i] = u[n][2 * p + e];
                        } else {
                            if (i < K.m) K.u_out[jn * (int64_t)K.m + i] = u[n][2 * p + e];
                        }
                    }
            }
        }
    }
}

// ---- decision wavefront ------------------------------------------------------------------------------
template <int G, int MP, int B, int NSW, bool SYM, int NL, int CLM, bool KOUT = false>
__device__ __forceinline__ void blk_decision_role(const BlkK &K, char *lds_generic, const BlkLds &L, int lane, ClState &cs)
{
    constexpr bool CL = CLM != 0;
    constexpr int NB = NL * G, R = blk_sublanes(NB), NW = blk_slots(NSW, NB);
    constexpr int RB = (int)blk_rec_bytes(MP, B, G, CL);
    lchar *lds = (lchar *)lds_generic;
    // neuron of the workgroup, sub-lane.  Four-neuron workgroups use 32 lanes; the other half shadows the last neuron (same reads,
    // same decisions, same stores to the same addresses) and is left out of the counters
    const bool shadow = lane / R >= NB;
    const int n = shadow ? NB - 1 : lane / R, r = lane % R;
    const int64_t wg = cs.cl;                                     // the workgroup's neurons: wg NB .. (cluster form: the cluster's; classic form: its neuron block)
    const int64_t jn = wg * NB + n;
    const bool active = jn < K.C;
    const int64_t N = K.N;
    const int M = K.M;
    const int nslots = K.nblk + 1;
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    const int o_d   = L.off_d + (r * B * NB + n) * 8;             // partial sums of sweep wavefronts r, r + R, ...
    const int o_x2  = L.off_x2 + (r * NB + n) * 16;
    const int o_w   = L.off_w + n * B * 4;
    const int o_wq  = L.off_wq + n * B * 8;
    const int o_out = L.off_out + n * kOutSteps * 8;

    // alphabet members r, r + R, ... of this sub-lane (NaN beyond M: never counted); larger alphabets loop over LDS
    const bool in_regs = M <= 4 * R;

    // the alphabet's progression (DevAlphabet): read ONCE, here -- inside the slot loop these would be loads from memory on the chain of
    // decisions (the stores of the loop may alias them for all the compiler knows), 10-25 % of a narrow layer's time
    const double al_a0 = K.alpha->a0, al_step = K.alpha->step, al_inv = K.alpha->inv, al_c0 = K.alpha->c0;
    const unsigned long long al_plus = K.alpha->plus, al_minus = K.alpha->minus;
    const double sym_top = SYM ? lds_ld<double>(lds, L.off_e + 8 * (2 + M - 1)) : 0.0;      // a;  a / 2 (0 for {-a, a})
    const double sym_hb = (SYM && M == 3) ? 0.5 * sym_top : 0.0;
    float wprev[B], qprev[B];                                     // block b-1 (final), this neuron
    float wprev2[B], qprev2[B];                                   // (cluster form) block b-2: D is two slots old there
#pragma unroll
    for (int j = 0; j < B; ++j) { wprev[j] = 0.f; qprev[j] = 0.f; wprev2[j] = 0.f; qprev2[j] = 0.f; }
    double DmCur = 0.0;                                           // (cluster form) D of the own step of the slot's block, over all slices: gathered a slot ago
    (void)wprev2; (void)qprev2; (void)DmCur;
    unsigned long long n_fallback = 0;
    auto wave_min_stop = [&](int v) -> int {                    // min over the wavefront of values in [0, B]
        int mn = B;
#pragma unroll
        for (int k = B - 1; k >= 0; --k)
            if (__ballot(v == k) != 0ull) mn = k;
        return mn;
    };

    // (not blk_sweep_flush shapes) Steps [t0, t1) of this workgroup's outputs from the LDS ring to memory: a lane takes 8 consecutive steps
    // of one neuron (32-byte runs of indices, 128-byte runs of values per neuron)
    constexpr bool kOwnFlush = !blk_sweep_flush<G, NL, CL>();
    auto flush = [&](int64_t t0, int64_t t1) {
        if constexpr (KOUT) {
            // Keras layout [N][ldo] (round 6: the layer's outputs as set_weights takes them, scripts/quantized_network.py:562, :570 -- no
            // transposing pass behind the kernel): a lane takes NJ consecutive NEURONS of one step -- NJ indices in one store, NJ values
            // in NJ / 4; the lanes of the wavefront are consecutive steps, so the reads of the LDS ring do not conflict
            constexpr int NJ = NB < 8 ? NB : 8, GJ = NB / NJ;
            for (int e = lane; e < GJ * kOutSteps; e += 64) {
                const int g = e / kOutSteps, c = e % kOutSteps;
                const int64_t ts = t0 + c, j0 = wg * NB + g * NJ;
                if (ts >= t1 || j0 >= K.C || (CL && cs.slice != 0)) continue;
                int idxj[NJ]; float qj[NJ];
#pragma unroll
                for (int k = 0; k < NJ; ++k) {
                    const int2 v = lds_ld<int2>(lds, L.off_out + ((g * NJ + k) * kOutSteps + (int)(ts % kOutSteps)) * 8);
                    idxj[k] = v.x; qj[k] = __int_as_float(v.y);
                }
                const int64_t o = ts * K.o_st + j0;
                const bool whole = j0 + NJ <= K.C && K.o_sj == 1;
                if (K.qidx) {
                    if (whole && NJ >= 4 && ((uintptr_t)(K.qidx + o) & (NJ - 1)) == 0) {
                        unsigned lo = 0, hi = 0;
#pragma unroll
                        for (int k = 0; k < NJ; ++k) { if (k < 4) lo |= (unsigned)(idxj[k] & 0xff) << (8 * k); else hi |= (unsigned)(idxj[k] & 0xff) << (8 * (k - 4)); }
                        if constexpr (NJ == 8) *reinterpret_cast<uint2 *>(K.qidx + o) = make_uint2(lo, hi);
                        else *reinterpret_cast<unsigned *>(K.qidx + o) = lo;
                    } else {
#pragma unroll
                        for (int k = 0; k < NJ; ++k) if (j0 + k < K.C) K.qidx[ts * K.o_st + (j0 + k) * K.o_sj] = (int8_t)idxj[k];
                    }
                }
                if (K.Qt) {
                    if (whole && NJ >= 4 && ((uintptr_t)(K.Qt + o) & 15) == 0) {
#pragma unroll
                        for (int k = 0; k + 3 < NJ; k += 4) *reinterpret_cast<float4 *>(K.Qt + o + k) = make_float4(qj[k], qj[k + 1], qj[k + 2], qj[k + 3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < NJ; ++k) if (j0 + k < K.C) K.Qt[ts * K.o_st + (j0 + k) * K.o_sj] = qj[k];
                    }
                }
            }
            return;
        }
        for (int e = lane; e < NB * (kOutSteps / 8); e += 64) {
            const int nn = e / (kOutSteps / 8), c = e % (kOutSteps / 8);
            const int64_t j = wg * NB + nn, ts = t0 + 8 * c;
            if (j >= K.C || ts >= t1 || (CL && cs.slice != 0)) continue;      // (cluster form: slice 0 writes the cluster's outputs)
            int idx8[8]; float q8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int2 v = lds_ld<int2>(lds, L.off_out + (nn * kOutSteps + (int)((ts + k) % kOutSteps)) * 8);
                idx8[k] = v.x; q8[k] = __int_as_float(v.y);
            }
            if (ts + 8 <= t1 && ((j * N + ts) & 7) == 0) {
                if (K.qidx) {
                    unsigned lo = 0, hi = 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) { lo |= (unsigned)(idx8[k] & 0xff) << (8 * k); hi |= (unsigned)(idx8[4 + k] & 0xff) << (8 * k); }
                    *reinterpret_cast<uint2 *>(K.qidx + j * N + ts) = make_uint2(lo, hi);
                }
                if (K.Qt) {
                    float4 *dst = reinterpret_cast<float4 *>(K.Qt + j * N + ts);
                    if (((uintptr_t)dst & 15) == 0) {
                        dst[0] = make_float4(q8[0], q8[1], q8[2], q8[3]);
                        dst[1] = make_float4(q8[4], q8[5], q8[6], q8[7]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) K.Qt[j * N + ts + k] = q8[k];
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (ts + k < t1) {
                        if (K.qidx) K.qidx[j * N + ts + k] = (int8_t)idx8[k];
                        if (K.Qt)   K.Qt[j * N + ts + k]   = q8[k];
                    }
            }
        }
    };
    int64_t flushed = 0, flush_hi = 0;                            // steps [0, flushed) are in memory; [flushed, flush_hi) are due

    // The record HEADERS a slot's decisions read (row statistics + Gram band: a few hundred bytes per step, about thirty LDS reads
    // per lane) were requested from the tile in LDS right after the slot's barrier -- exactly when every sweep wavefront requests
    // its operands: 2300 of the decision wavefront's 3500 cycles per slot were that queue (profiles/r03/blk_phase_stamps.txt).
    // They do not depend on anything the slot computes (global loads one slot ahead came back just as late, behind the DMA pieces:
    // DESIGN_HISTORY.md, "gpfq_blk.hip notes" 8).  So the headers travel once more, compact (1.5 KiB per tile), by LDS-DMA into a ring of three buffers TWO slots ahead,
    // and this wavefront reads tile b + 1's at the END of slot b, when the LDS is quiet: after the barrier only the sweeps'
    // partial sums and the block's weights are read.
    const int smh = lane & (B - 1);                               // this lane's step of a block
    // Seven-sweep-wavefront shapes (two wavefronts per SIMD, 256 registers each): the next tile's headers are requested behind the
    // chain and arrive under the certification.  Everywhere else that placement costs registers this role does not have at three
    // wavefronts per SIMD (hipcc spills 19-29 of them: cfg1's Dense(784 -> 128) 0.355 against 0.262 ms), and they are requested
    // behind the slot's stores instead (below).
    constexpr bool kHdrChain = NSW == 7;
    constexpr int BI = B > 1 ? B - 1 : 1, HDRB = blk_hdr_bytes(B, CL);
    // Only what the predicted dot products need: the bounds (E1, E2, cb, ca, Ea) are read from the tile in LDS while the chain
    // computes -- they are consumed by the certification after it, off the critical path.
    double2 g01 = make_double2(0.0, 0.0);                         // own step: (1/nrm^2, G)
    double2 ghp[B];                                               // own step against block b-1: (H1, H2) at distance B + sm - j
    double2 ghi[BI];                                              // own step against this block's steps j < sm (zeros beyond)
    // (what every sub-lane needs of the OTHER steps -- 1/nrm^2, G and <Xq_s, Xq_j> -- is fetched from the sub-lane that owns the
    //  step by DPP instead of being kept in another 28 registers across the slot)
    auto prefetch_headers = [&](int buf) {                        // the tile in buffer `buf` of the header ring
        const int rm = L.off_hr + buf * L.hr_pitch + smh * HDRB;
        g01 = lds_ld<double2>(lds, rm);
#pragma unroll
        for (int j = 0; j < B; ++j) ghp[j] = lds_ld<double2>(lds, rm + 64 + 32 * (B + smh - j - 1));
#pragma unroll
        for (int j = 0; j + 1 < B; ++j) ghi[j] = lds_ld<double2>(lds, j < smh ? rm + 64 + 32 * (smh - j - 1) : L.off_zero);
    };
    slot_barrier();                                               // (tile 0 and the headers of tiles 0, 1 landed)
    // (cluster form: the headers are read at the top of the slot that uses them -- its chain no longer waits for the sweeps' partial sums,
    //  so the post-barrier LDS burst costs it nothing, and 32 registers held across the barrier and the exchange made hipcc spill there)
    if constexpr (!CL) prefetch_headers(0);
    int hnext = 1;                                                // (b + 1) % 3
    // The B dependent decisions of a slot are a latency chain on a SIMD that two sweep wavefronts keep busy: at equal priority
    // every instruction of the chain waits its turn behind theirs (9640 cycles per slot, the longest path of the workgroup);
    // ahead of them it takes 6800 and the sweeps, which have the slack, fill the gaps.
    __builtin_amdgcn_s_setprio(3);

    unsigned long long dt0 = 0, dt1 = 0, dt2 = 0, dta = 0, dtb = 0, dacc_work = 0, dacc_bar = 0, dacc_pro = 0, dacc_chain = 0, dacc_tail = 0;
    (void)dt0; (void)dt1; (void)dt2; (void)dta; (void)dtb; (void)dacc_work; (void)dacc_bar; (void)dacc_pro; (void)dacc_chain; (void)dacc_tail;
    for (int b = 0; b < nslots; ++b) {
        STAMP(dt0);
        STAMP_DO(if (b) dacc_tail += dt0 - dt2;)   // after the barrier
        const int tbase = (b & 1) * L.tile_pitch;
        const int cbq = (b & 1) * NB * B * 8;
        float  wc[B], qc[B];                                      // this block: weights, decisions (as float32 values)
        double D[B];
        double DmT = 0.0;                                         // (cluster form) the own step's D over all slices: the slow path's D[s]
        (void)DmT;
        int stop = B;                                             // first step of the block this neuron could not certify
        // ---- the slot's decisions, in three parts (round 3).  Until round 2 a decision was ~100 instructions issued B times in
        // sequence by this one wavefront (3300 cycles per slot of four: the floor of every narrow layer).  Only a sliver of that is
        // the DEPENDENT part -- the next decision needs nothing of this one but q -- so:
        //  (1) prologue, one step per sub-lane (sub-lane sm = lane mod B owns step sm): D of the own step (the sweeps' partial sums),
        //      everything the pending increments of block b-1 and the WEIGHTS of this block's earlier steps contribute to the
        //      predicted dot product and to its error bound -- none of it depends on this block's decisions;
        //  (2) the chain, identical in all sub-lanes of a neuron: du_s = A_s - sum_{j<s} q_j H2[s][j], the quotient, the nearest member
        //      by ARITHMETIC on the uniform alphabet (k = rint((t - a_0) / step), clamped; q = a_0 + k step -- the host has checked that
        //      this reproduces float32(alphabet[k]) for every k: blk_uniform) -- about 16 instructions per step, no table, no LDS;
        //  (3) certification, one step per sub-lane again: the error bound, the margins against the TRUE table neighbours
        //      alphabet[k-1], alphabet[k], alphabet[k+1], rule (ii)'s certainty.  The first step of a neuron that fails stops its chain
        //      for the block exactly as before (slow path below): what the chain guessed from there on is discarded.
        //  Every LDS read is issued before the arithmetic; the stores come last (a store between two reads orders them).
        double cPm = 0.0, ePm = 0.0;                                            // own step: block b-1's increments (also the slow path's)
        int ctl_now = -1;                                                       // the control word this wavefront publishes (it never reads it back)
        unsigned anyP = 0u;
#pragma unroll
        for (int j = 0; j < B; ++j) anyP |= __float_as_uint(wprev[j]) | __float_as_uint(qprev[j]);
        if constexpr (CL) {
#pragma unroll
            for (int j = 0; j < B; ++j) anyP |= __float_as_uint(wprev2[j]) | __float_as_uint(qprev2[j]);
        }
        const int nvalid = (int)min((int64_t)B, N - (int64_t)b * B);            // steps beyond N pad the last block: no-ops
        const int oslot0 = (int)(((int64_t)b * B) % kOutSteps);
        const int o_dummy = L.off_ctl + 8;
        auto quad_bcast = [&](double x, int s) -> double {       // value of lane (4 * (lane / 4) + s)
            int lo = __double2loint(x), hi = __double2hiint(x);
            if (s == 0) { lo = __builtin_amdgcn_mov_dpp(lo, 0x00, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x00, 0xF, 0xF, true); }
            if (s == 1) { lo = __builtin_amdgcn_mov_dpp(lo, 0x55, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0x55, 0xF, 0xF, true); }
            if (s == 2) { lo = __builtin_amdgcn_mov_dpp(lo, 0xAA, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xAA, 0xF, 0xF, true); }
            if (s == 3) { lo = __builtin_amdgcn_mov_dpp(lo, 0xFF, 0xF, 0xF, true); hi = __builtin_amdgcn_mov_dpp(hi, 0xFF, 0xF, 0xF, true); }
            return __hiloint2double(hi, lo);
        };
        if (b < K.nblk) {
            const int sm = smh;                                   // this lane's step (B is a power of two <= 4)
            // ---- (1) reads
            double dp[NW];                                        // partial sums of the own step, every sweep wavefront's slot
#pragma unroll
            for (int w = 0; w < NW; ++w) dp[w] = lds_ld<double>(lds, L.off_d + ((((b & 1) * NW + w) * B + sm) * NB + n) * 8);
#pragma unroll
            for (int s = 0; s < B; ++s) wc[s] = lds_ld<float>(lds, o_w + (b & 1) * NB * B * 4 + 4 * s);
            const float w_own_raw = lds_ld<float>(lds, o_w + (b & 1) * NB * B * 4 + 4 * sm);
            double2 hp2[CL ? B : 1];                              // (cluster form) own step against block b-2, distance 2B + sm - j: from the tile, not kept across the slots
            if constexpr (CL) {
                prefetch_headers(b % 3);                          // tile b's headers: buffer b % 3 of the ring
#pragma unroll
                for (int j = 0; j < B; ++j) hp2[j] = lds_ld<double2>(lds, tbase + sm * RB + 64 + 32 * (2 * B + sm - j - 1));
            }
            if constexpr (kOwnFlush) {
                if (flush_hi > flushed) {                         // (outputs of earlier slots: under the latency of the reads above)
                    flush(flushed, flush_hi);
                    flushed = flush_hi;
                }
            }
            // (the record headers: requested before the last barrier, see prefetch_headers)
            double2 hp[B], hi_[BI];
            const double2 o01 = g01;
#pragma unroll
            for (int j = 0; j + 1 < B; ++j) hi_[j] = ghi[j];
#pragma unroll
            for (int s = 0; s < B; ++s) hp[s] = ghp[s];
            __builtin_amdgcn_sched_barrier(0);
            // ---- (1) arithmetic
            double wd[B];
#pragma unroll
            for (int s = 0; s < B; ++s) {
                wc[s] = s < nvalid ? wc[s] : 0.f;
                qc[s] = 0.f;
                wd[s] = (double)wc[s];
            }
            double Dm;                                            // D of the own step: fixed-order tree over the wavefronts' slots
            {
                double t[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) t[w] = dp[w];
#pragma unroll
                for (int span = 1; span < NW; span *= 2)
#pragma unroll
                    for (int w = 0; w + span < NW; w += 2 * span) t[w] += t[w + span];
                Dm = t[0];
            }
#pragma unroll
            for (int j = 0; j < B; ++j) {
                const double wj = (double)wprev[j], qj = (double)qprev[j];
                cPm = fma(wj, hp[j].x, cPm); cPm = fma(-qj, hp[j].y, cPm);
            }
            if constexpr (CL) {                                   // block b-2's increments are pending too
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    const double wj = (double)wprev2[j], qj = (double)qprev2[j];
                    cPm = fma(wj, hp2[j].x, cPm); cPm = fma(-qj, hp2[j].y, cPm);
                }
            }
            double Aw = cPm;                                      // + this block's weights before the own step
#pragma unroll
            for (int j = 0; j + 1 < B; ++j) Aw = fma(wd[j], hi_[j].x, Aw);
            if constexpr (CL) {
                // cluster form: what the sweeps left in LDS is this slice's share of block b + 1's D (the records carry row t + 2B): it leaves
                // for the other slices now and is gathered at the END of the slot, behind the stores, while this wavefront would wait at the
                // barrier anyway; this block's D came back a slot ago (DmCur).  lane = (neuron, step).
                static_assert(!CL || (B == 4 && R % 4 == 0), "cluster form: a quad of lanes = the four steps of a neuron's block");
                __builtin_amdgcn_sched_barrier(0);
                if (b + 1 < K.nblk) {
                    const double xv[1] = {Dm};
                    cl_publish<1>(K, cs, lane, xv);
                }
                Dm = DmCur;
                DmT = Dm;
                __builtin_amdgcn_sched_barrier(0);
            }
            const double Am = Dm + Aw;
            double A[B];
#pragma unroll
            for (int s = 0; s < B; ++s) A[s] = quad_bcast(Am, s);
            // the bounds of the own step, from the tile in LDS: requested now, consumed after the chain
            const int rbm = tbase + sm * RB;
            const double2 o23 = lds_ld<double2>(lds, rbm + 16);
            const double rEa = lds_ld<double>(lds, rbm + 32);
            const double2 o67 = lds_ld<double2>(lds, rbm + 48);   // (sE1, sE2): the quick certification's bound (below)
            STAMP(dta);
            // ---- (2) the chain
            const double u_a0 = al_a0, u_step = al_step, u_inv = al_inv, u_c0 = al_c0, u_kmax = (double)(M - 1);
            const double u_amax = fmax(fabs(u_a0), fabs(fma(u_kmax, u_step, u_a0))) * (1.0 + 0x1p-20);   // >= every |member| (and its float32 rounding)
            const unsigned long long u_plus = al_plus, u_minus = al_minus;
            const int u_zero = K.zero_idx;
            const float sym_a32 = (float)sym_top;
            auto pick = [&](double tt, double &kd) -> float {     // nearest member of the uniform alphabet, by arithmetic
                if constexpr (SYM && NB <= 8) {
                    // {-a, 0, a} / {-a, a}, exactly symmetric (blk_sym_a): two comparisons with the boundaries -a/2, a/2 (or 0) instead of
                    // the progression's arithmetic -- ~10 instructions fewer per decision on the wavefront whose chain is the floor of the
                    // narrow shapes (4096 x 512 on 1024 samples 1.416 -> 1.38-1.39 ms, same box).  Not in the 16-neuron shapes, whose
                    // slot is the sweeps': there the same change measured 2.5 % SLOWER (3.04-3.10 against 2.98-3.01 ms).
                    // (A tie goes to the lower index as argmin does; a decision that close is never certified anyway.)
                    // (the chain restated by hand for these alphabets was slower: DESIGN_HISTORY.md, "gpfq_blk.hip notes" 10)
                    const bool up = tt > sym_hb, mid = tt > -sym_hb;
                    kd = up ? u_kmax : (mid ? 1.0 : 0.0);
                    return up ? sym_a32 : (mid ? 0.f : -sym_a32);
                }
                kd = fmin(fmax(rint(fma(tt, u_inv, u_c0)), 0.0), u_kmax);
                const int ki = (int)kd;
                const int adj = (int)((unsigned)(u_plus >> ki) & 1u) - (int)((unsigned)(u_minus >> ki) & 1u);
                const float v = __int_as_float(__float_as_int((float)fma(kd, u_step, u_a0)) + adj);   // float32(alphabet[k]), see BlkK
                return ki == u_zero ? 0.f : v;
            };
            float q32s[B];
            double qd[B];
#pragma unroll
            for (int s = 0; s < B; ++s) {
                // (1/nrm^2, G and <Xq_s, Xq_j> of step s from the sub-lane that owns it)
                const double rden_s = quad_bcast(o01.x, s), wG_s = wd[s] * quad_bcast(o01.y, s);
                double du = A[s];
#pragma unroll
                for (int j = 0; j + 1 < B; ++j)
                    if (j < s) du = fma(-qd[j], quad_bcast(hi_[j].y, s), du);
                const double tq = (du + wG_s) * rden_s;
                const double tt = fabs(du) < 1e-10 ? wd[s] : tq;
                double kd;
                float q32 = pick(tt, kd);
                q32 = rden_s == 0.0 ? 0.f : q32;                  // rule (i): the pre-pass stores 1 / nrm^2 = 0 for nrm < 1e-16
                q32s[s] = q32;
                qd[s] = (double)q32;
            }
            // The own step's predicted dot product, from the lane's OWN band entries (zeros beyond the own step: a product with zero
            // adds nothing), in the chain's order -- the same bits as the chain's du of step sm, without selecting it out of the four
            // (round 5: two conditional moves per step of the chain, on the wavefront whose instruction count is a narrow layer's time).
            double du_m = Am;
#pragma unroll
            for (int j = 0; j + 1 < B; ++j) du_m = fma(-qd[j], hi_[j].y, du_m);
            // (seven-sweep-wavefront shapes, two wavefronts per SIMD and 256 registers each: the next tile's headers are requested
            //  HERE, behind the chain, and arrive under the certification; see kHdrChain)
            if constexpr (kHdrChain) {
                if (b + 1 < K.nblk) prefetch_headers(hnext);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- (3) certification of the own step
            const bool valid_m = sm < nvalid;
            const float w_m = valid_m ? w_own_raw : 0.f;
            const double wdm = (double)w_m, rden = o01.x, wGm = wdm * o01.y;
            const double rcb = o23.x, rca = o23.y;
            const bool rule1 = rden == 0.0;
            const bool small = fabs(du_m) < 1e-10;
            const double tq_m = (du_m + wGm) * rden;
            const double tt_m = small ? wdm : tq_m;
            double kd_m;
            const float qk32 = pick(tt_m, kd_m);                  // (the chain's arithmetic on the chain's operands: the same bits)
            const int ki = (int)kd_m;
            double a_lo, a_k, a_hi;                               // a[k-1], a[k], a[k+1] (-inf / +inf beyond the ends)
            if constexpr (SYM) {
                // {-a, 0, a} / {-a, a}, exactly symmetric in float64 (blk_sym_a): the members ARE (k - (M - 1) / 2) * gap with gap = a or
                // 2 a, every one of these products and sums exact -- no table read behind the pick (a dependent LDS round trip on the
                // wavefront whose latency chain is a narrow layer's slot: round 5)
                const double kInfD = __longlong_as_double(0x7ff0000000000000LL);
                const double gap = M == 3 ? sym_top : 2.0 * sym_top;
                a_k = fma(kd_m, gap, -sym_top);
                a_lo = ki == 0 ? -kInfD : a_k - gap;
                a_hi = ki == M - 1 ? kInfD : a_k + gap;
            } else
            {
                const int oe = L.off_e + 8 * (1 + ki);            // table with two sentinels on either side
                a_lo = lds_ld<double>(lds, oe); a_k = lds_ld<double>(lds, oe + 8); a_hi = lds_ld<double>(lds, oe + 16);
            }
            const double d_k = fabs(a_k - tt_m), d_lo = fabs(a_lo - tt_m), d_hi = fabs(a_hi - tt_m);
            const double slack43 = (CL ? K.slack : 0x1p-43) * (fabs(Dm) + fabs(du_m - Dm) + fabs(wGm)) * rden;      // float64 slack of the prediction
            const double base2 = 2.0 * fma(fabs(wdm), rcb, rca) + slack43;
            // The certification proper needs eps = sum over the pending increments of |w_j| E1 + |q_j| E2 (fourteen products on seven
            // band entries read from the tile, each at a sub-lane-dependent address): a third of this wavefront's instructions -- and
            // this wavefront IS the slot of every narrow layer.  Round 5: first a bound on that bound -- the largest pending |w| times
            // the sum of ALL the band's E1 plus the alphabet's largest |member| times the sum of its E2 (two numbers the pre-pass
            // leaves in the header: BlkStats::sE1 / sE2) -- which certifies all but about one slot in 10^4; only when some lane fails
            // it does the whole wavefront take the exact bound below (and with it rule (ii)'s certain case, which the quick test
            // never accepts: eps_q > 0 whenever the row has any overlap with its band).  eps_q >= eps term by term, so a decision
            // the quick test certifies is one the exact test certifies.
            bool ok;
            float wmx = fmaxf(fabsf(wprev[0]), fabsf(wc[0]));
#pragma unroll
            for (int j = 1; j < B; ++j) wmx = fmaxf(wmx, fmaxf(fabsf(wprev[j]), fabsf(wc[j])));
            if constexpr (CL) {
#pragma unroll
                for (int j = 0; j < B; ++j) wmx = fmaxf(wmx, fabsf(wprev2[j]));
            }
            constexpr int NPEND = blk_band(B, CL);                // increments that may be pending at a decision
            const double eps_q = fma((double)wmx, o67.x, u_amax * o67.y) + (double)NPEND * rEa;
            const bool sure_q = fabs(du_m) - eps_q >= 1e-10;                     // certainly not rule (ii)
            const double delta2_q = base2 + 2.0 * eps_q * rden;
            const bool far_q = (d_lo - d_k > delta2_q) & (d_hi - d_k > delta2_q);
            ok = rule1 | (far_q & sure_q) | !valid_m;
            if (__ballot(!ok) != 0ull)
            {
                double2 ep[B], ei_[BI];
#pragma unroll
                for (int j = 0; j < B; ++j) ep[j] = lds_ld<double2>(lds, rbm + 64 + 32 * (B + sm - j - 1) + 16);
#pragma unroll
                for (int j = 0; j + 1 < B; ++j) ei_[j] = lds_ld<double2>(lds, j < sm ? rbm + 64 + 32 * (sm - j - 1) + 16 : L.off_zero);
#pragma unroll
                for (int j = 0; j < B; ++j) {
                    ePm = fma(fabs((double)wprev[j]), ep[j].x, ePm); ePm = fma(fabs((double)qprev[j]), ep[j].y, ePm);
                }
                if constexpr (CL) {
#pragma unroll
                    for (int j = 0; j < B; ++j) {
                        const double2 e2 = lds_ld<double2>(lds, rbm + 64 + 32 * (2 * B + sm - j - 1) + 16);
                        ePm = fma(fabs((double)wprev2[j]), e2.x, ePm); ePm = fma(fabs((double)qprev2[j]), e2.y, ePm);
                    }
                }
                double eps = ePm;
#pragma unroll
                for (int j = 0; j + 1 < B; ++j) { eps = fma(fabs(wd[j]), ei_[j].x, eps); eps = fma(fabs(qd[j]), ei_[j].y, eps); }
                unsigned any_m = anyP;                                // any nonzero pending (w, q) before the own step
#pragma unroll
                for (int j = 0; j + 1 < B; ++j) any_m |= j < sm ? (__float_as_uint(wc[j]) | __float_as_uint(q32s[j])) : 0u;
                // (each of the up to 2B-1 pending increments may lose up to Ea = 2^-149 sum|Xq_t| to subnormal float32 products)
                eps += ((any_m << 1) != 0u) ? (double)NPEND * rEa : 0.0;
                const bool du_exact = eps == 0.0;                     // every pending increment orthogonal to Xq_t element-wise
                const bool msq = du_exact & small;                    // rule (ii), certain
                const bool sure = du_exact | (fabs(du_m) - eps >= 1e-10);   // ... or certainly not rule (ii)
                // twice the modelling error of the prediction (quotient units) + float64 slack
                const double delta2 = base2 + 2.0 * eps * rden;
                const bool far = (d_lo - d_k > delta2) & (d_hi - d_k > delta2);      // beyond the bound from both boundaries
                const bool first = (d_k < d_lo) & (d_k <= d_hi);                     // exact t (rule (ii)): argmin's first minimum (:57)
                const bool cert = (msq ? first : far) & sure;
                ok = rule1 | cert | !valid_m;
            }
            // first step of each neuron that is not certain: the R sub-lanes of a neuron sit side by side, sub-lane sm at bit sm
            const unsigned long long bad = __ballot(!ok);
            const float q_m = rule1 ? 0.f : qk32;
            const bool st = r < B;                                // (sub-lanes B.. repeat sub-lanes 0..B-1)
            if (bad == 0ull) {
                // every step of every neuron certified (all but about one slot in 10^4): no chain stops, nothing to select
                STAMP(dtb);
#pragma unroll
                for (int s = 0; s < B; ++s) qc[s] = q32s[s];
                const float qpub = SYM ? (q_m > 0.f ? -1.f : (q_m < 0.f ? 1.f : 0.f)) : q_m;
                lds_st<float2>(lds, st ? o_wq + cbq + 8 * sm : o_dummy, make_float2(w_m, qpub));
                lds_st<int2>(lds, (st & valid_m) ? o_out + ((oslot0 + sm) % kOutSteps) * 8 : o_dummy,
                             make_int2(rule1 ? K.zero_idx : ki, __float_as_int(q_m)));
                ctl_now = -1;
                if (lane == 0) lds_st<int>(lds, L.off_ctl + 4 * (b & 1), -1);
            } else
            {
            const unsigned field = (unsigned)(bad >> (lane & ~(R - 1))) & ((1u << B) - 1u);
            stop = field ? __builtin_ctz(field) : B;
            STAMP(dtb);
#pragma unroll
            for (int s = 0; s < B; ++s) qc[s] = s < stop ? q32s[s] : 0.f;
            const bool keep = sm < stop;
            const float q_st = keep ? q_m : 0.f;
            // what the sweeps multiply the (scaled) Xq row with: q, or minus its sign; (w, 0) for a step still to be decided
            const float qpub = SYM ? (q_st > 0.f ? -1.f : (q_st < 0.f ? 1.f : 0.f)) : q_st;
            lds_st<float2>(lds, st ? o_wq + cbq + 8 * sm : o_dummy, make_float2(w_m, qpub));
            lds_st<int2>(lds, (st & keep & valid_m) ? o_out + ((oslot0 + sm) % kOutSteps) * 8 : o_dummy,
                         make_int2(rule1 ? K.zero_idx : ki, __float_as_int(q_m)));
            // smallest stop over the workgroup's active neurons: scalar, from the ballot
            const unsigned long long badA = __ballot(!ok & active);
            int smin = B;
#pragma unroll
            for (int k = B - 1; k >= 0; --k) {
                unsigned long long pat = 0ull;
#pragma unroll
                for (int g = 0; g < 64 / B; ++g) pat |= 1ull << (g * B + k);
                if (badA & pat) smin = k;
            }
            ctl_now = smin < B ? smin : -1;
            if (lane == 0) lds_st<int>(lds, L.off_ctl + 4 * (b & 1), ctl_now);
            }
            // The next tile's headers (landed a slot ago) are requested HERE, when the LDS is quiet -- but no longer waited for at the
            // slot's barrier: the stores above must have reached the LDS before the barrier, the reads need not have returned.  So
            // the stores are waited for first (an LDS store completes in tens of cycles), then the reads are issued, and this
            // wavefront's barrier is a bare s_barrier: the reads' round trip -- which every sweep wavefront of a narrow layer sat
            // out at that barrier: the decision wavefront IS their slot -- overlaps the barrier and the next slot's first reads.
            // (Round 5.  Requested behind the chain instead they hide as well, but 32 more live registers through the certification
            //  made hipcc spill 19-29 registers in this role at three wavefronts per SIMD: cfg1's Dense(784 -> 128) 0.355 against
            //  0.262 ms; requested at the top of the slot they land in the post-barrier burst: +8 %.  profiles/r05/.)
            if constexpr (!kHdrChain) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if constexpr (!CL) { if (b + 1 < K.nblk) prefetch_headers(hnext); }
            }
            if constexpr (CL) {
                // every slice's share of block b + 1's D, added in slice order: the same bits in every slice (published above)
                if (b + 1 < K.nblk) {
                    double xv[1] = {0.0};
                    // (9+ slices in the 16-neuron shape: sixteen per batch of loads -- a batch is a trip to memory, and two of them no longer
                    //  fit into this wavefront's wait at the barrier: 4096 x 4096 on 16384 samples 51.0 -> 48.0 ms, 4096 x 1024 on 28672
                    //  37.3 -> 33.1; the 4- / 8-neuron shapes, whose slot is this wavefront's, spill with the second instantiation)
                    if (NL == 4 && K.nsl > 8) cl_gather<1, NL == 4 ? 16 : 8>(K, cs, lane, xv);
                    else cl_gather<1, 8>(K, cs, lane, xv);
                    DmCur = xv[0];
                }
            }
        } else {
            STAMP(dta);
            STAMP(dtb);
            if (lane == 0) lds_st<int>(lds, L.off_ctl + 4 * (b & 1), -1);
        }

        // One decision (:83-89, :57) in full, from the record in LDS: the slow path's resumed chains (and what the hot chain above
        // is checked against by the parity tests).  commit == this lane's chain is still running.  Returns false when not certifiable.
        auto decide = [&](int s, bool commit, auto inregs_tag) -> bool {
            constexpr bool IN_REGS = decltype(inregs_tag)::value;
            double am[4];                                                        // (slow path only: not kept across the slots)
#pragma unroll
            for (int q = 0; q < 4; ++q) am[q] = (IN_REGS && r + q * R < M) ? lds_ld<double>(lds, L.off_e + 8 * (2 + r + q * R)) : kNaN;
            const int rb = tbase + s * RB;
            const double2 r01 = lds_ld<double2>(lds, rb), r23 = lds_ld<double2>(lds, rb + 16), r45 = lds_ld<double2>(lds, rb + 32);
            const double rden = r01.x, rG = r01.y, rcb = r23.x, rca = r23.y, rEa = r45.x, nrm = r45.y;
            const bool rule1 = nrm < 1e-16;                                      // rule (i): literal 0
            // not yet applied increments: block b-1 (formed above by sub-lane s) and this block's steps before s (distance s - j)
            double corr = quad_bcast(cPm, s), eps = quad_bcast(ePm, s);
            unsigned anyinc = anyP;
#pragma unroll
            for (int j = 0; j < B; ++j) {
                if (j < s) {
                    const int d = s - j;
                    const double2 h = lds_ld<double2>(lds, rb + 64 + 32 * (d - 1)), e = lds_ld<double2>(lds, rb + 64 + 32 * (d - 1) + 16);
                    const double wj = (double)wc[j], qj = (double)qc[j];
                    corr = fma(wj, h.x, corr); corr = fma(-qj, h.y, corr);
                    eps = fma(fabs(wj), e.x, eps); eps = fma(fabs(qj), e.y, eps);
                    anyinc |= __float_as_uint(wc[j]) | __float_as_uint(qc[j]);
                }
            }
            // (each of the up to 2B-1 pending increments may lose up to Ea = 2^-149 sum|Xq_t| to subnormal float32 products)
            eps += ((anyinc << 1) != 0u) ? (double)blk_band(B, CL) * rEa : 0.0;
            const double wd = (double)wc[s];
            const double du = D[s] + corr;                                       // predicted <Xq_t, u_{t-1}>
            const bool   du_exact = eps == 0.0;                                  // every pending increment orthogonal to Xq_t element-wise
            const bool   msq = du_exact & (fabs(du) < 1e-10);                    // rule (ii), certain
            const bool   sure = du_exact | (fabs(du) - eps >= 1e-10);            // ... or certainly not rule (ii)
            const double wG = wd * rG;
            const double tq = (du + wG) * rden;                                  // predicted quotient
            const double tt = msq ? wd : tq;
            // twice the modelling error of the prediction (quotient units) + float64 slack
            const double delta2 = 2.0 * (fma(fabs(wd), rcb, rca) + eps * rden)
                                  + (CL ? K.slack : 0x1p-43) * (fabs(D[s]) + fabs(corr) + fabs(wG)) * rden;
            int idx_l;
            double q_l;
            bool cert;
            if constexpr (SYM) {
                // {-a, 0, a} / {-a, a}, exactly symmetric in float64 (blk_sym_a): the boundaries are -a/2 and a/2 (or 0), exact, and
                // the nearest member follows from two comparisons -- a tie goes to the lower index as argmin does (:57), though a
                // decision that close is never certified.  Twice the distance from the nearer boundary is the margin.
                const bool up = tt > sym_hb, mid = tt > -sym_hb;
                idx_l = up ? M - 1 : (mid ? 1 : 0);
                q_l = up ? sym_top : (mid ? 0.0 : -sym_top);
                const double m2 = 2.0 * fmin(fabs(tt - sym_hb), fabs(tt + sym_hb));
                cert = (msq | (m2 > delta2)) & sure;
            } else {
            int c = 0;                                                           // members below t, counted by the R sub-lanes
            if constexpr (IN_REGS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) c += (am[q] < tt) ? 1 : 0;
            } else {
                for (int kk = r; kk < M; kk += R) c += (lds_ld<double>(lds, L.off_e + 8 * (2 + kk)) < tt) ? 1 : 0;
            }
            const int p = sub_sumi<R>(c);
            const int oe = L.off_e + 8 * p;
            const double lolo = lds_ld<double>(lds, oe), lo = lds_ld<double>(lds, oe + 8), hi = lds_ld<double>(lds, oe + 16),
                         hihi = lds_ld<double>(lds, oe + 24);
            const double d_lo = fabs(lo - tt), d_hi = fabs(hi - tt), d_ll = fabs(lolo - tt), d_hh = fabs(hihi - tt);
            const bool at0 = p == 0, atM = p == M;
            const bool use_hi = at0 | (!atM & !(d_lo <= d_hi));                  // tie -> lower index
            idx_l = use_hi ? p : p - 1;
            q_l   = use_hi ? hi : lo;
            const double m2_in = fabs(d_hi - d_lo), m2_lo = d_hh - d_hi, m2_hi = d_ll - d_lo;
            const double m2 = at0 ? m2_lo : (atM ? m2_hi : m2_in);               // twice the distance from the boundary
            const bool plateau = !use_hi & (p >= 2) & !(d_ll > d_lo);            // a lower member at the same distance would win
            cert = !plateau & (msq | (m2 > delta2)) & sure;
            }
            const bool valid = s < nvalid;
            const float q32 = (rule1 | !valid) ? 0.f : (float)q_l;
            const int   idx = rule1 ? K.zero_idx : idx_l;
            const bool ok = rule1 | cert | !valid;
            const bool sel = commit & ok;
            qc[s] = sel ? q32 : qc[s];
            wc[s] = (sel & !valid) ? 0.f : wc[s];
            const bool st = sel & (r == 0);                                      // stores that are not wanted land in the dummy slot
            // what the sweeps multiply the (scaled) Xq row with: q, or minus its sign
            const float qpub = SYM ? (q32 > 0.f ? -1.f : (q32 < 0.f ? 1.f : 0.f)) : q32;
            lds_st<float2>(lds, st ? o_wq + cbq + 8 * s : o_dummy, make_float2(wc[s], qpub));
            lds_st<int2>(lds, (st & valid) ? o_out + ((oslot0 + s) % kOutSteps) * 8 : o_dummy, make_int2(idx, __float_as_int(q32)));
            return ok;
        };
        using T_ = std::true_type;
        using F_ = std::false_type;
        hnext = hnext == 2 ? 0 : hnext + 1;
        STAMP(dt1);
        if (kHdrChain || b >= K.nblk) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the last slot's only store; chain-prefetch shapes: the stores)
        asm volatile("s_barrier" ::: "memory");
        STAMP(dt2);
        STAMP_DO(dacc_work += dt1 - dt0; dacc_bar += dt2 - dt1; dacc_pro += dta - dt0; dacc_chain += dtb - dta;)

        // ---- slow path: exact decision of the stopped step, then the chain resumes ----
        for (;;) {
            const int S = ctl_now;                                // (wave-uniform: formed from ballots)
            if (S < 0) break;
            slot_barrier();                                       // exact partials published
            // (the hot path only forms the own step's D: here every sub-lane needs all of them, sub-lane r adding slots r, r + R, ...)
            if constexpr (CL) {                                   // (the sums over all slices, from the sub-lanes that own the steps)
#pragma unroll
                for (int s = 0; s < B; ++s) D[s] = quad_bcast(DmT, s);
            } else {
#pragma unroll
            for (int s = 0; s < B; ++s) {
                double d = 0.0;
#pragma unroll
                for (int q = 0; q < NW / R; ++q) d += lds_ld<double>(lds, o_d + ((((b & 1) * NW + q * R) * B) + s) * NB * 8);
                D[s] = sub_sum<R>(d);
            }
            }
            double du = 0.0, dw = 0.0;
#pragma unroll
            for (int q = 0; q < NW / R; ++q) {
                const double2 v = lds_ld<double2>(lds, o_x2 + q * R * NB * 16);
                du += v.x; dw += v.y;
            }
            du = sub_sum<R>(du); dw = sub_sum<R>(dw);
            if constexpr (CL) {                                   // the exact dot products over all slices
                double xv[2] = {du, dw};
                cl_exchange<2, 4>(K, cs, lane, xv);
                du = xv[0]; dw = xv[1];
            }
            const bool mine = active && stop == S;
            if (mine) {
                const int64_t t = (int64_t)b * B + S;
                const double nrm = lds_ld<double2>(lds, tbase + S * RB + 32).y;
                float wS = 0.f;
#pragma unroll
                for (int s = 0; s < B; ++s) wS = (s == S) ? wc[s] : wS;
                const double te = dw / (nrm * nrm);
                const double t2 = fabs(du) < 1e-10 ? (double)wS : te;
                int bi = 0;
                double bq = lds_ld<double>(lds, L.off_e + 16), bdist = fabs(bq - t2);
                for (int kk = 1; kk < M; ++kk) {
                    const double ak = lds_ld<double>(lds, L.off_e + 8 * (2 + kk)), dk = fabs(ak - t2);
                    if (dk < bdist) { bdist = dk; bi = kk; bq = ak; }
                }
                const float q32 = (float)bq;
#pragma unroll
                for (int s = 0; s < B; ++s) qc[s] = (s == S) ? q32 : qc[s];
                if (r == 0) {
                    lds_st<float2>(lds, o_wq + cbq + 8 * S, make_float2(wS, SYM ? (q32 > 0.f ? -1.f : (q32 < 0.f ? 1.f : 0.f)) : q32));
                    lds_st<int2>(lds, o_out + (int)(t % kOutSteps) * 8, make_int2(bi, __float_as_int(q32)));
                    if (!shadow) ++n_fallback;
                }
            }
            // resume the chains that were stopped at S
            int stop2 = mine ? B : stop;
#pragma unroll
            for (int s = 0; s < B; ++s) {
                if (s > S) {                                      // (uniform)
                    const bool ok = in_regs ? decide(s, mine && stop2 == B, T_{}) : decide(s, mine && stop2 == B, F_{});
                    if (mine && stop2 == B && !ok) stop2 = s;
                }
            }
            stop = stop2;
            const int smin = wave_min_stop(active ? stop : B);
            ctl_now = smin < B ? smin : -1;
            if (lane == 0) lds_st<int>(lds, L.off_ctl + 4 * (b & 1), ctl_now);
            slot_barrier();                                       // chains resumed, control word rewritten
        }

        // block b is final: it becomes "the previous block"; its outputs leave the LDS ring in the next slot (a sweep wavefront's job) or,
        // where this wavefront flushes, when the ring is full -- at the top of a slot, behind that slot's first LDS reads
#pragma unroll
        for (int j = 0; j < B; ++j) {
            if constexpr (CL) { wprev2[j] = wprev[j]; qprev2[j] = qprev[j]; }
            wprev[j] = (b < K.nblk) ? wc[j] : 0.f; qprev[j] = (b < K.nblk) ? qc[j] : 0.f;
        }
        if constexpr (kOwnFlush) {
            const int64_t done = min((int64_t)(b + 1) * B, N);
            if (b < K.nblk && (done - flushed >= kOutSteps - B + 1 || done == N)) { flush_hi = done; }
        }
    }
    if constexpr (kOwnFlush) {
        if (flush_hi > flushed) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); flush(flushed, flush_hi); }
    }

    if (K.fallback_count && n_fallback && (!CL || cs.slice == 0)) atomicAdd(K.fallback_count, n_fallback);   // rare
    STAMP_DO(if (K.stamps && blockIdx.x == 0 && lane == 0) { K.stamps[16] = dacc_work; K.stamps[17] = dacc_bar; K.stamps[18] = dacc_pro; K.stamps[19] = dacc_chain; K.stamps[20] = dacc_tail; })
    slot_barrier();                                               // residual-norm partials published
    if (K.resid) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < NW / R; ++q) tot += lds_ld<double>(lds, o_d + q * R * B * NB * 8);
        tot = sub_sum<R>(tot);
        if constexpr (CL) {
            double xv[1] = {tot};
            cl_exchange<1, 8>(K, cs, lane, xv);
            tot = xv[0];
        }
        if (active && r == 0 && (!CL || cs.slice == 0)) K.resid[jn] = sqrt(tot);
    }
}

}  // namespace

// G neuron groups per sweep wavefront (4G neurons per workgroup), S sample pairs per k-lane over the NSW sweep
// wavefronts (rows of MP = (128 / G) * S samples), B steps per slot.
// NL neurons per lane of a sweep wavefront: NL * G neurons per workgroup (4; 2 for layers of at most 512 neurons, which
// then fill twice the CUs with half the element-wise work per slot -- a narrow layer is bound by the time of ONE slot).
// CL (cluster form): workgroup id -> (cluster, slice) with a cluster's slices side by side in ONE XCD's queue (workgroups go to the
// XCDs round-robin by id): id = ((cluster / 8) nsl + slice) 8 + cluster % 8.  Workgroups are dispatched in id order, so whenever a
// slice is resident every slice before it in the queue is resident or done -- the oldest cluster with work left is always complete on
// the chip (nsl <= 28 <= the 32 CUs of an XCD) and the exchange cannot deadlock, whatever the mapping of ids to XCDs really is.
// KOUT: the outputs in the Keras layout [N][ldo] (BlkK::o_st), written by the decision wavefront's flush -- instantiated for the 16-neuron
// four-step shapes only (blk_kout_shape): there that wavefront has the slack; in every other shape the extra code cost registers the
// hot loops do not have (SGPR and VGPR spills, 4-6 % on the narrow shapes), and their outputs stay neuron-major (one assembly pass follows).
template <int G, int S, int B, int NSW, bool SYM, int NL, int CLM = 0, bool KOUT = false>
__global__ void __launch_bounds__(64 * (NSW + 1))
gpfq_blk_kernel(BlkK K)
{
    constexpr bool CL = CLM != 0;
    constexpr int NB = NL * G, KQ = 64 / G, MP = 2 * KQ * S;
    // The alphabet lives in device memory (DevAlphabet).  One that is not an arithmetic progression the chain can index -- only possible when
    // it was formed on the device from a median that is zero or not finite -- runs nothing: the call's alphabet word is raised instead and the
    // caller reruns the layer with a host alphabet (gpfq_call_status, layer.quantize_dense).  Uniform over the launch: no exchange is left waiting.
    if (K.alpha->ok == 0) {
        if (K.alpha_err && blockIdx.x == 0 && threadIdx.x == 0) *K.alpha_err = 1;
        return;
    }
    ClState cs{0, 0, 0u, false, 0, 0};
    if constexpr (CL) {
        const int id = (int)blockIdx.x, j = id >> 3;
        if (K.cl_map == 0) {                                      // the slices of a cluster side by side in ONE XCD's queue
            cs.slice = j % K.nsl;
            cs.cl = (int64_t)(j / K.nsl) * 8 + (id & 7);
        } else {                                                  // consecutive ids: the slices of a cluster go round the XCDs
            cs.cl = id / K.nsl;
            cs.slice = id - (int)cs.cl * K.nsl;
        }
        if (cs.cl * NB >= K.C) return;                            // (the last group of eight clusters may be short)
        cs.rec_off = (int64_t)cs.slice * K.slice_bytes;
        const int left = K.m - cs.slice * MP;
        cs.m_sl = left < 0 ? 0 : (left > MP ? MP : left);
    }
    if constexpr (!CL) {
        // Classic form: workgroup id -> neuron block, XCD-aware (round 6).  Workgroups go to the eight XCDs round-robin by id, so with the
        // identity map the two workgroups that share a 128-byte line of the Keras kernel's row (32 neurons) -- and of the Keras-layout
        // outputs -- sit on DIFFERENT XCDs: every line of W crosses into two L2s and every output line is written back in halves.  Giving
        // XCD x the x-th eighth of the neuron blocks keeps neighbours in one L2.  (Launches whose workgroup count is no multiple of 8 keep
        // the identity.)
        const unsigned nb = gridDim.x, b = blockIdx.x;
        cs.cl = (nb & 7u) == 0u ? (int64_t)((b & 7u) * (nb >> 3) + (b >> 3)) : (int64_t)b;
    }
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const BlkLds L = blk_lds(MP, NB, B, NSW, G, CL);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- LDS initialisation: zero weights / partials / (w, q) / control, alphabet table with sentinels ----
    for (int i = tid; i < (L.off_e - L.off_w) / 4; i += blockDim.x) reinterpret_cast<int *>(lds + L.off_w)[i] = 0;
    for (int i = tid; i < (L.total - L.off_out) / 4; i += blockDim.x) reinterpret_cast<int *>(lds + L.off_out)[i] = 0;
    if (tid < 68) {
        const double kInf = __longlong_as_double(0x7ff0000000000000LL);
        const int k = tid - 2;
        reinterpret_cast<double *>(lds + L.off_e)[tid] = k < 0 ? -kInf : (k < K.M ? K.alpha->a[k] : kInf);
    }
    if (tid < 2) reinterpret_cast<int *>(lds + L.off_ctl)[tid] = -1;
    __syncthreads();

    if (wave < NSW) {
        int pbase = 0, pw = 1;
#pragma unroll
        for (int w = 0; w < NSW; ++w) { pbase += (w < wave) ? KQ * (int)K.pw[w] : 0; pw = (w == wave) ? (int)K.pw[w] : pw; }
        // (one instantiation of the role per pair count that the shape's split holds)
#define GPFQ_BLK_ROLE(PW_)                                                                                             \
        if constexpr (blk_split_has<G, S, NSW, NL>(PW_)) {                                                                    \
            if (pw == PW_) blk_sweep_role<G, PW_, MP, B, NSW, SYM, NL, CLM>(K, lds, L, wave, lane, pbase, cs);          \
        }
        GPFQ_BLK_ROLE(1) GPFQ_BLK_ROLE(2) GPFQ_BLK_ROLE(3) GPFQ_BLK_ROLE(4) GPFQ_BLK_ROLE(5) GPFQ_BLK_ROLE(6)
#undef GPFQ_BLK_ROLE
    } else {
        blk_decision_role<G, MP, B, NSW, SYM, NL, CLM, KOUT>(K, lds, L, lane, cs);
    }
}

// ---- host side ------------------------------------------------------------------------------------
struct BlkShape { int G, S, B, mp, NW, NL, NS; };  // NL: neurons per lane (4 G or 2 G neurons per workgroup); NS: 0 = classic; >= 2: slices of the cluster form (mp = samples of a slice)
static std::atomic<int> g_blk_single{1};  // one neuron per workgroup for layers of at most 128 neurons (blk_set_single_groups)
void blk_set_single_groups(int on) { g_blk_single.store(on ? 1 : 0, std::memory_order_relaxed); }
static std::atomic<int> g_blk_nw{0};      // sweep wavefronts of the 16-neuron B = 4 shapes: 8, 11, or 0 = by shape (blk_set_sweep_waves)
static std::atomic<int> g_blk_four{1};    // 4-neuron workgroups for layers of at most 1024 neurons on rows of 769..2048 samples
void blk_set_four_groups(int on) { g_blk_four.store(on ? 1 : 0, std::memory_order_relaxed); }
static std::atomic<int> g_blk_wide{1};    // 16-neuron workgroups for rows beyond 1024 samples in layers wider than 2048 neurons
void blk_set_wide_groups(int on) { g_blk_wide.store(on ? 1 : 0, std::memory_order_relaxed); }
void blk_set_sweep_waves(int nw) { g_blk_nw.store(nw == 11 ? 11 : (nw == 8 ? 8 : 0), std::memory_order_relaxed); }

// C: neurons of the call.  Rows of 769..1024 samples are the one shape whose slot is bound by the sweeps (nine sample pairs
// on three of the SIMDs), not by the chain of decisions: up to 2048 neurons -- one round of 256 workgroups with 8 neurons
// each -- the sweeps are halved by giving a workgroup 8 neurons instead of 16 (4096 x 2048, m = 1024: 4.0 -> 2.9 ms).
static std::atomic<int> g_blk_quad{2};    // four neuron groups x 1 / 2 neurons per lane for layers of at most 2048 neurons on rows of 257..1024 samples (1: 129..2048 only)
void blk_set_quad_groups(int on) { g_blk_quad.store(on < 0 ? 0 : (on > 2 ? 2 : on), std::memory_order_relaxed); }
static std::atomic<int> g_blk_quad_nw{0}; // sweep wavefronts of the four-group narrow shapes on rows of at most 768 samples: 7, 8, or 0 = by shape
void blk_set_quad_waves(int nw) { g_blk_quad_nw.store(nw == 7 ? 7 : (nw == 8 ? 8 : 0), std::memory_order_relaxed); }
static std::atomic<int> g_blk_pairs{1};   // two-neuron workgroups for layers of at most 512 neurons
void blk_set_pair_groups(int on) { g_blk_pairs.store(on ? 1 : 0, std::memory_order_relaxed); }

// Cluster form (round 5): long rows -- up to 28672 samples -- are cut into slices of 1024 samples, one workgroup of the headline shape
// <4,32,4> x 11 each, that exchange their partial dot products once per slot (cl_publish / cl_gather).  Option blk_cluster: 1 (default)
// = by row length and width (blk_shape), 0 = off (rows beyond 5120 samples then keep the several-wavefronts-per-neuron kernel), a value
// from 1024 up = every row beyond that many samples (tests, A/B).
static std::atomic<int> g_blk_cluster_nl{0};      // cluster form: neurons per lane, 0 = by width; 1 / 2 / 4 force it (option blk_cluster_nl)
void blk_set_cluster_nl(int v) { g_blk_cluster_nl.store(v == 1 || v == 2 || v == 4 ? v : 0, std::memory_order_relaxed); }
static std::atomic<int> g_blk_prep_run{1};         // 1 (default): the record pre-pass takes runs of 4 .. 16 records per workgroup for walks of 2048+ steps; 0: one record per workgroup; 4 .. 16: runs of that many at any length (option blk_prep_run: A/B, tests)
void blk_set_prep_run(int v) { g_blk_prep_run.store(v >= 4 && v <= 16 ? v : (v ? 1 : 0), std::memory_order_relaxed); }
static std::atomic<int> g_blk_prep_norms{1};       // 1 (default): gpfq_quantize_dense_layer's row norms inside the record pre-pass where that is bit-identical; 0: always the row-norm kernel (option blk_prep_norms: A/B, tests)
void blk_set_prep_norms(int v) { g_blk_prep_norms.store(v ? 1 : 0, std::memory_order_relaxed); }
static std::atomic<int> g_blk_chip{-1};            // -1: ask the device; 0 / 1 force the answer of blk_chip_ok (option blk_chip_ok: tests)
void blk_set_chip_ok(int v) { g_blk_chip.store(v < 0 ? -1 : (v ? 1 : 0), std::memory_order_relaxed); }
static std::atomic<int> g_blk_cl_timeout_ms{3000};  // how long an exchange of the cluster form waits for a slice before it gives up (option blk_cluster_timeout_ms)
void blk_set_cluster_timeout_ms(int v) { g_blk_cl_timeout_ms.store(v < 1 ? 1 : v, std::memory_order_relaxed); }
static std::atomic<int> g_blk_cl_fault{0};         // tests: 1 = slice 1 of cluster 0 never publishes (forces the exchange's timeout; option blk_cluster_fault)
void blk_set_cluster_fault(int v) { g_blk_cl_fault.store(v ? 1 : 0, std::memory_order_relaxed); }
static std::atomic<int> g_blk_cluster768{-1};      // rows of 2049..3072 samples in wide layers as four 768-sample slices: -1 (default) yes, 0 off, 8 / 11 force the sweep wavefronts (option blk_cluster768)
void blk_set_cluster768(int v) { g_blk_cluster768.store(v == 0 ? 0 : (v == 8 ? 8 : (v == 11 ? 11 : -1)), std::memory_order_relaxed); }
static std::atomic<int> g_blk_cluster{1};
void blk_set_cluster(int v) { g_blk_cluster.store(v <= 0 ? 0 : (v < 1024 ? 1 : v), std::memory_order_relaxed); }
constexpr int64_t kClusterMaxM = 28672;      // = GPFQ_ONCHIP_MAX_M: 28 slices, still inside one XCD's 32 CUs
// The cluster form's co-residency argument (gpfq_blk_kernel) is about THIS chip: 256 compute units in 8 XCDs of 32, the workgroups of a
// launch handed to the XCDs round-robin and started in order.  A device that shows anything else -- a partition of the chip (CPX / DPX / QPX
// modes: 32 / 128 / 64 compute units), a compute-unit mask from the environment -- does not get the cluster shapes: its long rows keep the
// classic shapes and the several-wavefronts-per-neuron kernel.  Asked once per device; no device (the CPU build container
// sizing a workspace): the full chip is assumed.
static bool blk_chip_ok()
{
    static std::atomic<int> cache[64];                       // per device: 0 unknown, 1 ok, 2 not
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return true; }
    if (dev < 0 || dev >= 64) return false;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int cus = 0;
        bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus == 256;
        for (const char *name : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK", "HSA_CU_MASK_SKIP_INIT"})
            if (const char *e = getenv(name)) ok = ok && e[0] == 0;
        v = ok ? 1 : 2;
        cache[dev].store(v, std::memory_order_relaxed);
    }
    const int forced = g_blk_chip.load(std::memory_order_relaxed);      // (tests: option blk_chip_ok = 0 pretends the chip is partitioned)
    return forced >= 0 ? forced != 0 : v == 1;
}
// Workgroup id -> (cluster, slice).  Workgroups go to the XCDs round-robin by id and every XCD starts its own in order.  Map 0 puts the
// slices of a cluster side by side in ONE XCD's queue (the exchange stays inside that XCD's L2 domain); but an XCD's 32 CUs then hold
// 32 / nsl whole clusters and 32 % nsl slices of the next one, which wait a whole round for their mates -- and so does every round after
// (three slices: four rounds' time for three rounds of work).  Map 1, consecutive ids, strands at most one cluster of the whole chip.
static std::atomic<int> g_blk_cluster_map{-1};     // -1: by the number of slices (blk_cluster_map below); 0 / 1 force a map (option blk_cluster_map)
void blk_set_cluster_map(int v) { g_blk_cluster_map.store(v < 0 ? -1 : (v ? 1 : 0), std::memory_order_relaxed); }
static int blk_cluster_map(int nsl, int64_t clusters)
{
    const int v = g_blk_cluster_map.load(std::memory_order_relaxed);
    if (v >= 0) return v;
    // The two maps take the same time at every slice count measured (profiles/r05/cluster_form.txt) -- the exchange hides behind the chain of
    // decisions across XCDs as well as inside one.  But where the slice count divides 8 (or is 16), consecutive ids put slice s of EVERY
    // cluster on XCD s mod 8: an XCD's 32 workgroups then stream one or two record streams instead of all of them, and the launch's HBM
    // traffic is the headline's per slice (FETCH_SIZE at 4096 x 4096 on 8192 samples: 44.4 GB under map 0) -- so map 1 there.
    if (nsl > 0 && (8 % nsl == 0 || nsl % 8 == 0)) return 1;
    // Other slice counts: whole clusters per round -- an XCD's 32 CUs hold 32 / nsl of them under map 0 (the rest of its CUs wait a round
    // for their mates), the chip's 256 hold 256 / nsl under map 1.  256 / nsl >= 8 (32 / nsl) for every nsl, so map 1 never needs more
    // rounds than map 0 (the comparison this function used to make always came out the same way): the default is map 1 at
    // every slice count; map 0 -- and its per-XCD deadlock-freedom argument -- runs only when the option forces it (tests, A/B).
    // (Measured, profiles/r05/cluster_form.txt: the two tie at 3, 5, 6, 7 slices of 256 clusters; 20 slices: 32 rounds against 22,
    //  122.0 against 83.9 ms at 4096 x 4096 on 20000 samples; 28 slices of 64 clusters: 40.3 against 37.3 ms.)
    (void)clusters;
    return 1;
}

static BlkShape blk_shape(int64_t m, int64_t C)
{
    if (m <= 0 || C <= 0) return {0, 0, 0, 0, 0, 0};               // (an empty calibration set has no shape: the slice count below would divide by zero)
    {
        // By default (1): every row beyond 3072 samples -- a slot of the cluster form is the headline shape's (its exchange hides behind the
        // chain of decisions), the classic shapes lose a third of that rate from 2049 samples up --, and rows of 1537..3072 samples
        // wherever the layer is ONE round of the chip (an XCD holds 32 / slices clusters at a time: up to 2048 neurons at two slices,
        // 1280 at three): profiles/r05/cluster_form.txt.
        const int clm = g_blk_cluster.load(std::memory_order_relaxed);
        const int ns_ = (int)((m + 1023) / 1024);
        const bool one_round = ns_ <= 32 && ((C + 15) / 16 + 7) / 8 <= 32 / ns_;      // with 16 neurons per workgroup
        const bool take = blk_chip_ok() && (clm == 1 ? (m > 3072 || (m > 1536 && one_round && (m > 2048 || C > 128))) : (clm > 1 && m > clm));   // (at most 128 neurons on rows of at most 2048 samples: the one-neuron workgroups stay ahead, 1.98 / 2.07 ms)
        // Round 6: rows of 2049..3072 samples in layers wider than 2048 neurons -- three 1024-sample slices would be 85 clusters per round
        // (four rounds for 3.01 rounds of work: 12.6 ms at 4096 x 4096 on 3000 samples, behind the classic one-step shape's 11.5) -- as FOUR
        // slices of 768 samples: 64 clusters per round, four whole rounds of a shorter slot.  (Other ragged lengths gain nothing from
        // shorter slices: a 768-sample slot takes 0.875 of a 1024-sample one)
        if (clm == 1 && blk_chip_ok() && m > 2048 && m <= 3072 && C > 2048 && g_blk_cluster768.load(std::memory_order_relaxed) != 0)
            return {4, 24, 4, 768, 11, 4, 4};                      // (eight or eleven sweep wavefronts: launch_blk, by the alphabet's form)
        if (take && m <= kClusterMaxM) {
            // neurons per workgroup: the fewest (4, 8, 16) with which the layer is still ONE round of the chip -- a slot of the 4- and
            // 8-neuron shapes is the decision wavefront's (with the exchange's flight exposed), a slot of the 16-neuron shape the sweeps'
            const int ns = (int)((m + 1023) / 1024);
            const int cn = g_blk_cluster_nl.load(std::memory_order_relaxed);
            // (measured, profiles/r05/cluster_form.txt: 8 neurons per workgroup wherever the layer is then still one round -- an XCD holds
            //  32 / ns clusters at a time --, 16 otherwise; 4 per workgroup never beat 8)
            int nl = 4;
            if (cn == 1 || cn == 2) nl = cn;
            else if (cn == 0 && ((C + 7) / 8 + 7) / 8 <= 32 / ns) nl = 2;
            return {4, 32, 4, 1024, nl == 4 ? 11 : 8, nl, ns};
        }
    }
    // Round 4 (dot products on the matrix unit, fused with the updates pair by pair): rows of 769..1024 samples take ELEVEN sweep wavefronts
    // (three per SIMD: 2.97 against 3.13 ms at 4096 x 4096 x 1024 -- the per-wavefront fold that made eleven lose until round 3 is gone);
    // shorter rows keep eight (768 samples: 2.53 against 2.63 ms; 512: 1.96 against 1.98)
    const int nw_opt = g_blk_nw.load(std::memory_order_relaxed);
    const int nw4 = nw_opt ? nw_opt : 8, nw4_long = nw_opt ? nw_opt : 11;
    // Layers of at most 512 neurons: TWO neurons per workgroup (two per lane of the sweep wavefronts).  A narrow layer is bound by
    // the time of one slot, a slot by the instructions its workgroup issues (profiles/r03/blk_phase_stamps.txt): half the neurons
    // are half the element-wise work per slot on twice the CUs.  Rows of up to 5120 samples (cfg4's Dense(2048 -> 128) on 5008).
    // Layers of at most 128 neurons: ONE neuron per workgroup -- the 64 workgroups of the two-neuron form leave three CUs in four idle,
    // and a slot's sweeps are half as long again (cfg1's Dense(784 -> 128), cfg4's Dense(2048 -> 128) on 5008 samples)
    // Round 4: layers of 129..2048 neurons on rows of 257..1024 samples take FOUR neuron groups per sweep wavefront with one or two
    // neurons per lane (4 or 8 per workgroup): the fused matrix form of the 16-neuron shape.  The one-group shapes below fold every
    // step's partial dot products over all 64 lanes (six dependent stages per row); here the matrix instruction leaves two row
    // rotations per slot, and the short rows no longer put 16 neurons on a workgroup (a layer of 1024 neurons filled 64 CUs):
    // 4096 x 512 on 1024 samples 1.74 -> 1.56 ms, x 1024 2.10 -> 1.65, x 2048 2.44 -> 1.97; 4096 x 1024 on 768 samples 2.69 -> 1.48,
    // on 512 samples 1.96 -> 1.38 (profiles/r04/latency_shapes.txt).  At most 128 neurons: 1.57 both ways at first; with the pair split of
    // BlkSplitQuad32 the four-group form is ahead there too (4096 x 128 on 1024 samples 1.58 -> 1.40 ms, on 768 1.62 -> 1.39, on 512 1.37 ->
    // 1.33; 4096 x 10 1.62 -> 1.50; cfg1's Dense(784 -> 128) 0.279 -> 0.271) and is the default for every layer of at most 2048 neurons.
    const int quad = g_blk_quad.load(std::memory_order_relaxed);         // 2 (default): every layer of at most 2048 neurons; 1: 129..2048 only; 0: off
    if (quad && C <= 2048 && (quad >= 2 || C > 128) && m > 256 && m <= 1024) {
        const int nl = C > 1024 ? 2 : 1;
        const int qw = g_blk_quad_nw.load(std::memory_order_relaxed);
        const int qnw = qw ? qw : (nl == 1 ? 7 : 8);                  // (seven sweep wavefronts: see BlkSplit7; rows beyond 768 samples keep eight)
        if (m <= 512) return {4, 16, 4, 512, qnw, nl};
        if (m <= 768) return {4, 24, 4, 768, qnw, nl};
        return {4, 32, 4, 1024, qw == 7 ? 7 : 8, nl};                 // (rows of 769..1024 samples: seven only when the option forces it)
    }
    if (C <= 128 && g_blk_pairs.load(std::memory_order_relaxed) != 0 && g_blk_single.load(std::memory_order_relaxed) != 0) {
        if (m > 256 && m <= 512) return {1, 4, 4, 512, 4, 1};
        if (m > 512 && m <= 1024) return {1, 8, 4, 1024, 8, 1};
        if (m > 1024 && m <= 1536) return {1, 12, 2, 1536, 8, 1};
        if (m > 1536 && m <= 2048) return {1, 16, 2, 2048, 8, 1};
        if (m > 2048 && m <= 3072) return {1, 24, 1, 3072, 8, 1};
        if (m > 3072 && m <= 4096) return {1, 32, 1, 4096, 8, 1};
        if (m > 4096 && m <= 5120) return {1, 40, 1, 5120, 8, 1};    // (eleven sweep wavefronts of 2-4 pairs: 2.17 against 2.06 ms at 2048 x 128 on 5008 samples)
    }
    if (C <= 512 && g_blk_pairs.load(std::memory_order_relaxed) != 0) {
        if (m > 256 && m <= 512) return {1, 4, 4, 512, 4, 2};
        if (m > 512 && m <= 1024) return {1, 8, 4, 1024, 8, 2};
        if (m > 1024 && m <= 1536) return {1, 12, 2, 1536, 8, 2};
        if (m > 1536 && m <= 2048) return {1, 16, 2, 2048, 8, 2};
        if (m > 2048 && m <= 3072) return {1, 24, 1, 3072, 8, 2};
        if (m > 3072 && m <= 4096) return {1, 32, 1, 4096, 8, 2};
        if (m > 4096 && m <= 5120) return {1, 40, 1, 5120, 8, 2};
    }
    if (m > 256 && m <= 512) return {4, 16, 4, 512, nw4, 4};
    if (m > 512 && m <= 768) return {4, 24, 4, 768, nw4, 4};
    // layers of at most 1024 neurons on rows of 769+ samples: FOUR neurons per workgroup (a quarter of the 16-neuron sweep per slot;
    // 256 workgroups hold 1024 neurons) -- the slot is then the chain of decisions
    const bool four = C <= 1024 && g_blk_four.load(std::memory_order_relaxed) != 0;
    if (four && m > 768 && m <= 1024) return {1, 8, 4, 1024, 8, 4};
    if (four && m > 1024 && m <= 1536) return {1, 12, 2, 1536, 8, 4};
    if (four && m > 1536 && m <= 2048) return {1, 16, 2, 2048, 8, 4};
    if (m > 768 && m <= 1024) return C <= 2048 ? BlkShape{2, 16, 4, 1024, 8, 4} : BlkShape{4, 32, 4, 1024, nw4_long, 4};
    // rows beyond 1024 samples: 8 neurons per workgroup and eight sweep wavefronts -- or, in layers of more than 2048 neurons (where
    // that takes two rounds of workgroups), 16 neurons over eleven sweep wavefronts: one round, half the decisions and folds per weight
    // (a 16-neuron workgroup takes 1.6 x as long as an 8-neuron one: 6.5 against 4.1 ms for 4096 steps of 2048 samples; rounds of 256)
    const int64_t rounds16 = (C + 4095) / 4096, rounds8 = (C + 2047) / 2048;
    const bool wide = C > 2048 && 8 * rounds16 <= 5 * rounds8 && g_blk_wide.load(std::memory_order_relaxed) != 0;
    if (m > 1024 && m <= 1536) return wide ? BlkShape{4, 48, 2, 1536, 11, 4} : BlkShape{2, 24, 2, 1536, 8, 4};
    if (m > 1536 && m <= 2048) return wide ? BlkShape{4, 64, 2, 2048, 11, 4} : BlkShape{2, 32, 2, 2048, 8, 4};
    // rows of 2049..4096 samples: a record is 48 / 64 KiB, so a slot is ONE step (B = 1) -- 8 neurons over eleven sweep wavefronts
    // (six 64-sample pairs each), or 4 neurons over eight for layers of at most 1024 neurons.  Still four to five times the
    // several-wavefronts-per-neuron kernel these rows had (4096 x 4096 x 4096: 38 ms)
    if (m > 2048 && m <= 3072) return four ? BlkShape{1, 24, 1, 3072, 8, 4} : BlkShape{2, 48, 1, 3072, 11, 4};
    if (m > 3072 && m <= 4096) return four ? BlkShape{1, 32, 1, 4096, 8, 4} : BlkShape{2, 64, 1, 4096, 11, 4};
    // rows of 4097..5120 samples (the reference's CIFAR10 runs calibrate on 5000 images, quantize_pretrained_cnn.py:78): a 60 KiB
    // record, 4 neurons per workgroup at any width -- a layer of 4096 neurons takes four rounds of workgroups, still several times
    // the several-wavefronts-per-neuron kernel these rows had
    if (m > 4096 && m <= 5120) return {1, 40, 1, 5120, 8, 4};
    return {0, 0, 0, 0, 0, 0};
}

// (blk_uniform, blk_sym_of, form_device_alphabet: gpfq_device.hpp -- shared with the median's last workgroup, which forms the alphabet itself)

// A host alphabet's DevAlphabet, computed on the host (launch_blk), stored into the call's workspace.
__global__ void gpfq_alphabet_store_kernel(DevAlphabet *out, DevAlphabet D, unsigned *counters)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = D;
    if (counters && threadIdx.x < 16) counters[threadIdx.x] = 0u;     // (the call's counter block, in the same launch: no memset of its own)
}

// The layer alphabet formed ON the device (gpfq_layer_alphabet_device): rad = float64(alphabet_scalar) * float64(float32 median) -- the
// reference's legacy-NumPy product (:544: python scalar times np.float32 is a float64 product) --, members rad * unit[k] (:545: float64
// products), and the progression the chain of decisions indexes.  `want_sym`: the caller will launch the symmetric-form instantiations
// (the unit alphabet is {-1, 0, 1} / {-1, 1}): an alphabet that is then not exactly symmetric is not ok.
__global__ void gpfq_alphabet_device_kernel(DevAlphabet *out, const float *__restrict__ median32, double alphabet_scalar, AlphabetArg unit, int want_sym)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    form_device_alphabet(out, median32[0], alphabet_scalar, unit, want_sym);
}

// workspace: [records of slots 0..nblk, + one record of DMA over-read][compact headers of the same records, + 2 KiB of over-read]
static size_t blk_recs_bytes(int64_t nblk, const BlkShape &sh)          // (cluster form: of ONE slice)
{
    return ((size_t)((nblk + 1) * sh.B + 1) * (size_t)blk_rec_bytes(sh.mp, sh.B, sh.G, sh.NS != 0) + 255) & ~(size_t)255;
}
// cluster form: the exchange buffers, [cluster][2][slice][64 lanes][4 words]; clusters in whole groups of eight
static int64_t blk_clusters(int64_t C, const BlkShape &sh) { return (((C + sh.NL * sh.G - 1) / (sh.NL * sh.G)) + 7) / 8 * 8; }
static size_t blk_mbox_bytes(int64_t C, const BlkShape &sh) { return sh.NS > 1 ? (size_t)blk_clusters(C, sh) * 2 * sh.NS * 64 * 4 * 8 : 0; }
static size_t blk_hdrs_bytes(int64_t nblk, int B, bool CL = false) { return (size_t)((nblk + 1) * B + 1) * (size_t)blk_hdr_bytes(B, CL) + 2048; }
static size_t blk_hdrs_off(int64_t nblk, const BlkShape &sh) { return blk_recs_bytes(nblk, sh) * (size_t)(sh.NS ? sh.NS : 1); }     // compact headers: behind the last record stream
static size_t blk_mbox_off(int64_t nblk, const BlkShape &sh) { return blk_hdrs_off(nblk, sh) + ((blk_hdrs_bytes(nblk, sh.B, sh.NS != 0) + 255) & ~(size_t)255); }

// Whether the kernel itself writes Keras-layout outputs for this shape (the 16-neuron four-step shapes: blk_kout_shape)
bool blk_keras_out_supported(int64_t m, int64_t C)
{
    const BlkShape sh = blk_shape(m, C);
    return sh.G == 4 && sh.B == 4 && sh.NL == 4 && sh.NS == 0;
}

bool blk_supported(const PipeArgs &a)
{
    const BlkShape sh = blk_shape(a.m, a.C);
    if (sh.G == 0 || a.N < 1 || a.m < 1) return false;
    if (a.A.M > 64 || !a.A.ascending) return false;
    DevAlphabet D{};
    D.M = a.A.M;
    for (int k = 0; k < a.A.M; ++k) D.a[k] = a.A.a[k];
    if (!blk_uniform(D)) return false;      // (other alphabets keep the row-group kernels)
    return a.N + 64 < (1LL << 31) / 64;
}

constexpr size_t kAlphaBlock = 1024;        // the call's DevAlphabet, in front of the records (a host alphabet is stored there; a device alphabet is the caller's block)
static_assert(sizeof(DevAlphabet) <= kAlphaBlock, "alphabet block");

size_t blk_workspace_bytes(int64_t N, int64_t m, int64_t Cn)
{
    if (m <= 0 || N < 0) return 0;          // (an empty calibration set: blk_shape has no shape for it -- and must not be asked)
    // (the record layout depends on the row length and, through the steps per slot, on the width class of the layer: the largest)
    size_t need = 0;
    for (int64_t C : {(int64_t)1 << 30, (int64_t)2048, (int64_t)1024, (int64_t)512, Cn > 0 ? Cn : (int64_t)1}) {   // (and the call's own width: the cluster form's neurons per workgroup)
        const BlkShape sh = blk_shape(m, C);
        if (!sh.G) continue;
        const int64_t nblk = (N + sh.B - 1) / sh.B;
        // (cluster form: a record stream per slice, the compact headers once, the exchange buffers + the error word)
        const size_t b = blk_recs_bytes(nblk, sh) * (size_t)(sh.NS ? sh.NS : 1) + ((blk_hdrs_bytes(nblk, sh.B, sh.NS != 0) + 255) & ~(size_t)255) + blk_mbox_bytes(Cn, sh);
        if (b > need) need = b;
    }
    return need ? need + kAlphaBlock : 0;
}

// a32 of a symmetric alphabet {-a, 0, a} or {-a, a} (DevAlphabet::sym_a), else 0.  PipeArgs::variant bit 1 (option "variant" bit 5) keeps the general form (A/B timing).
// (device alphabets: a.A is the UNIT alphabet linspace(-1, 1, M) -- symmetric exactly when rad * unit is, for a finite rad > 0)
static float blk_sym_a(const PipeArgs &a)
{
    if (a.variant & 2) return 0.f;
    return blk_sym_of(a.A.a, a.A.M);
}

constexpr bool blk_kout_shape(int G, int B, int NL, int CLM) { return G == 4 && B == 4 && NL == 4 && CLM == 0; }

template <int G, int S, int B, int NSW, bool SYM, int NL, int CLM = 0, bool KOUT = false>
static hipError_t launch_blk_sym(const PipeArgs &a, const BlkShape &sh, const DevAlphabet *alpha, hipStream_t stream)
{
    constexpr bool CL = CLM != 0;
    if constexpr (!KOUT && blk_kout_shape(G, B, NL, CLM)) {
        if (a.o_st != 1) return launch_blk_sym<G, S, B, NSW, SYM, NL, CLM, true>(a, sh, alpha, stream);
    }
    if (!KOUT && a.o_st != 1) return hipErrorInvalidValue;        // (Keras-layout outputs: blk_keras_out_supported said no)
    constexpr int NB = NL * G;
    const BlkLds L = blk_lds(sh.mp, NB, B, NSW, G, CL);
    const unsigned grid = CL ? (unsigned)(blk_clusters(a.C, sh) * sh.NS) : (unsigned)((a.C + NB - 1) / NB);
    auto *kern = gpfq_blk_kernel<G, S, B, NSW, SYM, NL, CLM, KOUT>;
    hipError_t e = ensure_dynamic_lds((const void *)kern, (size_t)L.total);
    if (e != hipSuccess) return e;
    BlkK K;
    const int64_t nblk_ = (a.N + B - 1) / B;
    char *const wbase = static_cast<char *>(a.workspace) + kAlphaBlock;       // records, compact headers, exchange buffers: behind the alphabet block
    K.recs = wbase; K.Wt = a.Wt; K.ldw = a.ldw; K.ldt = a.ldt;
    K.hdrs = K.recs + blk_hdrs_off(nblk_, sh);
    K.alpha = alpha;
    K.alpha_err = a.fallback_count ? reinterpret_cast<int *>(a.fallback_count + 1) + 1 : nullptr;   // (fourth 32-bit word of the call's counter block, zeroed with it)
    K.nsl = CL ? sh.NS : 0; K.cl_map = blk_cluster_map(sh.NS, (a.C + NB - 1) / NB); K.slice_bytes = (int64_t)blk_recs_bytes(nblk_, sh); K.mbox = nullptr; K.cl_err = nullptr;
    K.u_ld = a.m; K.slack = 0x1p-43 * (double)(sh.NS > 1 ? sh.NS : 1);
    K.cl_timeout = 100000ull * (unsigned long long)g_blk_cl_timeout_ms.load(std::memory_order_relaxed); K.cl_fault = g_blk_cl_fault.load(std::memory_order_relaxed);
    if constexpr (CLM == 1) {
        char *mb = wbase + blk_mbox_off(nblk_, sh);
        const size_t mbytes = blk_mbox_bytes(a.C, sh);
        e = hipMemsetAsync(mb, 0, mbytes, stream);                 // sequence numbers start at 1: a zero word is "not yet written"
        if (e != hipSuccess) return e;
        K.mbox = reinterpret_cast<unsigned long long *>(mb);
        K.cl_err = a.fallback_count ? reinterpret_cast<int *>(a.fallback_count + 1) : nullptr;   // (second word of the call's counter block, zeroed with it)
    }
    K.N = a.N; K.C = a.C; K.m = (int)a.m; K.M = a.A.M; K.zero_idx = a.A.zero_idx; K.nblk = (int)((a.N + B - 1) / B);
    K.qidx = a.qidx; K.Qt = a.Qt; K.resid = a.resid; K.u_out = a.u_out; K.fallback_count = a.fallback_count;
    K.o_sj = a.o_st == 1 ? a.N : a.o_sj; K.o_st = a.o_st;             // neuron-major [C][N] unless the caller asked for the Keras layout
    K.stamps = a.fallback_count ? a.fallback_count + 8 : nullptr;      // (diagnostic build: the unused row-statistics area behind the counter block)
    K.Xq = a.Xq; K.ldx = a.ld;
    {
        const int *pw = blk_split<G, S, NSW, NL>();
        for (int w = 0; w < 12; ++w) K.pw[w] = (unsigned char)(w < NSW ? pw[w] : 0);
    }
    hipEvent_t ev0, ev1;
    if (main_kernel_events(&ev0, &ev1))    // (a benchmark's events on this launch alone, when it asked for them: the dispatch's own start and end)
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(64 * (NSW + 1)), (unsigned)L.total, stream, ev0, ev1, 0, K);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (NSW + 1)), (size_t)L.total, stream, K);
    return hipGetLastError();
}

// (the eleven-wavefront variants of the shapes that also exist with eight are kept in the general form only: an experiment switch)
constexpr bool blk_has_sym(int S, int NSW) { (void)S; (void)NSW; return true; }

template <int G, int S, int B, int NSW = 8, int NL = 4>
static hipError_t launch_blk_inst(const PipeArgs &a, const BlkShape &sh, const DevAlphabet *alpha, hipStream_t stream)
{
    // (sweep-bound shapes gain 3-4 %, the others nothing)
    if constexpr (blk_has_sym(S, NSW)) {
        if (blk_sym_a(a) != 0.f) return launch_blk_sym<G, S, B, NSW, true, NL>(a, sh, alpha, stream);
    }
    return launch_blk_sym<G, S, B, NSW, false, NL>(a, sh, alpha, stream);
}

// One cluster-form launch in flight per device: the co-residency of a cluster's slices is argued for ONE such launch on the
// chip; a second one on another stream (the class surface's look-ahead capture runs on one) would share the compute units with it and
// could strand slices of both.  So a cluster launch on a stream other than the last one's first waits, on the device, for that one to end
// (an event recorded behind every cluster launch); launches on one stream are ordered anyway.  The guard holds the lock from the wait to
// the record, so two host threads cannot interleave theirs.
struct ClusterLaunchGuard {
    static std::mutex &mu() { static std::mutex m; return m; }
    struct Last { hipEvent_t ev = nullptr; hipStream_t stream = nullptr; bool any = false; };
    static Last &last(int dev) { static Last L[64]; return L[dev & 63]; }
    std::unique_lock<std::mutex> lock;
    hipStream_t stream;
    int dev = 0;
    hipError_t err = hipSuccess;
    explicit ClusterLaunchGuard(hipStream_t s) : lock(mu()), stream(s)
    {
        err = hipGetDevice(&dev);
        if (err != hipSuccess) return;
        Last &l = last(dev);
        if (!l.ev) err = hipEventCreateWithFlags(&l.ev, hipEventDisableTiming);
        if (err == hipSuccess && l.any && l.stream != stream) err = hipStreamWaitEvent(stream, l.ev, 0);
    }
    ~ClusterLaunchGuard()
    {
        if (err != hipSuccess) return;
        Last &l = last(dev);
        if (hipEventRecord(l.ev, stream) == hipSuccess) { l.any = true; l.stream = stream; }
    }
};

// The call's DevAlphabet: the caller's block (a.dev_alpha, formed on the device: gpfq_quantize_dense_layer), or the host alphabet a.A
// with its progression computed here and stored in front of the records by one single-thread kernel.
static hipError_t blk_alphabet(const PipeArgs &a, const DevAlphabet **alpha, hipStream_t stream)
{
    if (a.dev_alpha) {
        *alpha = a.dev_alpha;
        return a.zero_counters && a.fallback_count ? hipMemsetAsync(a.fallback_count, 0, 64, stream) : hipSuccess;
    }
    DevAlphabet D{};
    D.M = a.A.M; D.zero_idx = a.A.zero_idx;
    D.rad = std::nan("");
    for (int k = 0; k < a.A.M && k < 64; ++k) D.a[k] = a.A.a[k];
    if (!blk_uniform(D)) return hipErrorInvalidValue;               // (blk_supported said otherwise)
    D.sym_a = blk_sym_a(a);
    D.ok = 1;
    DevAlphabet *dst = static_cast<DevAlphabet *>(a.workspace);
    hipLaunchKernelGGL(gpfq_alphabet_store_kernel, dim3(1), dim3(64), 0, stream, dst, D,
                       a.zero_counters ? reinterpret_cast<unsigned *>(a.fallback_count) : (unsigned *)nullptr);
    *alpha = dst;
    return hipGetLastError();
}

bool blk_unit_wants_sym(const AlphabetArg &unit)
{
    PipeArgs a{};
    a.A = unit;
    return blk_sym_a(a) != 0.f;
}

hipError_t launch_alphabet_device(const float *median32, double alphabet_scalar, const AlphabetArg &unit, void *dev_alphabet, hipStream_t stream)
{
    PipeArgs a{};
    a.A = unit;
    hipLaunchKernelGGL(gpfq_alphabet_device_kernel, dim3(1), dim3(64), 0, stream, static_cast<DevAlphabet *>(dev_alphabet), median32,
                       alphabet_scalar, unit, blk_sym_a(a) != 0.f ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_blk(const PipeArgs &a, hipStream_t stream)
{
    const BlkShape sh = blk_shape(a.m, a.C);
    if (!sh.G) return hipErrorInvalidValue;
    const int64_t nblk = (a.N + sh.B - 1) / sh.B;
    const int64_t nrec = (nblk + 1) * sh.B + 1;
    const bool r64 = blk_row64(sh.G, sh.B);
    // a.phase: 0 = the whole call; 1 = the alphabet-independent half (the record pre-pass, rows unscaled); 2 = the rest (symmetric form: the
    // records' Xq rows scaled in place, then the kernel) -- gpfq_dense_layer_prepare / _run, which a caller runs on two streams
    const bool do_prep = a.phase != 2, do_run = a.phase != 1;
    const DevAlphabet *alpha = nullptr;
    hipError_t e = do_run ? blk_alphabet(a, &alpha, stream) : hipSuccess;
    if (e != hipSuccess) return e;
    char *const wbase = static_cast<char *>(a.workspace) + kAlphaBlock;
    const int sym_shape = (sh.NS || blk_has_sym(sh.S, sh.NW)) && blk_sym_a(a) != 0.f ? 1 : 0;
    const int sym_prep = a.phase == 0 ? sym_shape : 0;            // (two-phase calls: the scaling follows in phase 2)
    if (a.phase == 2 && sym_shape) {
        const int mps = sh.mp;
        hipLaunchKernelGGL(gpfq_blk_scale_kernel, dim3((unsigned)nrec, (unsigned)(sh.NS ? sh.NS : 1)), dim3(256), 0, stream, wbase,
                           (int64_t)blk_rec_bytes(mps, sh.B, sh.G, sh.NS != 0), blk_hdr_bytes(sh.B, sh.NS != 0), mps, sh.NS ? (int64_t)blk_recs_bytes(nblk, sh) : (int64_t)0, alpha);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    const bool vec_rows = (a.ld % 4 == 0) && (((uintptr_t)a.X | (uintptr_t)a.Xq) % 16 == 0);
    const bool run_form = !sh.NS && vec_rows && (g_blk_prep_run.load(std::memory_order_relaxed) > 1 || (g_blk_prep_run.load(std::memory_order_relaxed) == 1 && nrec >= 2048));
    // (the row norms, when the caller left them to this call: inside the pre-pass where it forms the row-norm kernel's very sums, else by that kernel)
    const float *nrm32 = a.nrm32;
    unsigned *zero16 = nullptr;
    if (do_prep && !nrm32 && a.nrm32_out) {
        zero16 = reinterpret_cast<unsigned *>(a.fallback_count);
        // (mp == 1024 exactly: every thread of the pre-pass's workgroups then holds one 16-byte piece of a row -- with fewer, whole wavefronts sit
        //  the loop out and the wavefront sum would read lanes that never ran: tools/fuzz_parity.py found that at 292 samples)
        const bool fuse = run_form && sh.mp == 1024 && a.m % 4 == 0 && g_blk_prep_norms.load(std::memory_order_relaxed) != 0;
        if (!fuse) {
            e = launch_row_norms(a.Xq, a.N, a.m, a.ld, a.nrm32_out, stream, zero16);
            if (e != hipSuccess) return e;
            nrm32 = a.nrm32_out; zero16 = nullptr;
        }
    }
    if (sh.NS) {                                                   // cluster form: NS slices of the headline shape
        note_dense_kernel("gpfq_blk_kernel, cluster form (rows cut into 1024-sample slices: one workgroup of 8 or 11 sweep wavefronts + 1 decision wavefront per slice, partial dot products exchanged once per slot)");
        const int sym = sym_shape;
        if (do_prep) {
            hipLaunchKernelGGL((gpfq_blk_prep_kernel<4, true, true>), dim3((unsigned)nrec), dim3(256), 0, stream, a.X, a.Xq, a.ld, a.N, (int)a.m, sh.mp * sh.NS,
                               nrm32, wbase, wbase + blk_hdrs_off(nblk, sh), alpha, sym_prep,
                               sh.NS, (int64_t)blk_recs_bytes(nblk, sh));
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        if (!do_run) return hipSuccess;
        ClusterLaunchGuard one_at_a_time(stream);
        if (one_at_a_time.err != hipSuccess) return one_at_a_time.err;
#define GPFQ_BLK_CL(S_, NSW_, NL_, M_) (sym ? launch_blk_sym<4, S_, 4, NSW_, true, NL_, M_>(a, sh, alpha, stream) : launch_blk_sym<4, S_, 4, NSW_, false, NL_, M_>(a, sh, alpha, stream))
        if (sh.S == 24) {
            // four 768-sample slices (blk_shape).  Sweep wavefronts, measured at 4096 x 4096 on 3000 samples (tools/c768_probe.py): the symmetric
            // form 10.29 ms with eleven against 10.68 with eight, the general form 11.02 with eight against 11.25 with eleven
            const int forced = g_blk_cluster768.load(std::memory_order_relaxed);
            const int nw = forced == 8 || forced == 11 ? forced : (sym ? 11 : 8);
            return nw == 8 ? GPFQ_BLK_CL(24, 8, 4, 1) : GPFQ_BLK_CL(24, 11, 4, 1);
        }
        if (sh.NL == 4) return GPFQ_BLK_CL(32, 11, 4, 1);
        return sh.NL == 1 ? GPFQ_BLK_CL(32, 8, 1, 1) : GPFQ_BLK_CL(32, 8, 2, 1);
#undef GPFQ_BLK_CL
    }
    auto *prep = sh.B == 4 ? (r64 ? gpfq_blk_prep_kernel<4, true> : gpfq_blk_prep_kernel<4, false>)
                           : (sh.B == 2 ? (r64 ? gpfq_blk_prep_kernel<2, true> : gpfq_blk_prep_kernel<2, false>) : gpfq_blk_prep_kernel<1, false>);
    const int sym = sym_prep;   // (exactly the launches launch_blk_inst gives the symmetric form)
    const bool vec = (a.ld % 4 == 0) && (((uintptr_t)a.X | (uintptr_t)a.Xq) % 16 == 0);
    if (!do_prep) {
    } else if (vec && (g_blk_prep_run.load(std::memory_order_relaxed) > 1 || (g_blk_prep_run.load(std::memory_order_relaxed) == 1 && nrec >= 2048))) {
        // (walks of fewer than 2048 steps: runs of records would be fewer workgroups than the chip has compute units, each a chain of eight
        //  records -- a Dense(128 -> 10) layer's pre-pass 0.08 ms longer; those keep one record per workgroup)
        // runs of 4 .. 16 records per workgroup (gpfq_blk_prep_run_kernel); the one-record form keeps the rows it cannot read 16 bytes at a time.
        // The kernel holds four workgroups per compute unit (116 registers): 1024 at a time.  A workgroup is a chain of 1 + run round trips to
        // memory, so the run length is the one in 4 .. 16 under which the launch is the fewest rounds x (1 + run) -- the SHORTEST run that
        // still fits one round where there is one (4101 records of the headline layer: runs of five, 821 workgroups).
        const int opt = g_blk_prep_run.load(std::memory_order_relaxed);
        int run = 8;
        if (opt >= 4 && opt <= kPrepRunMax) run = opt;
        else {
            int64_t best = -1;
            for (int rl = 4; rl <= kPrepRunMax; ++rl) {
                const int64_t wgs = (nrec + rl - 1) / rl, cost = ((wgs + 1023) / 1024) * (1 + rl);
                if (best < 0 || cost < best) { best = cost; run = rl; }
            }
        }
        auto *prun = sh.B == 4 ? (r64 ? gpfq_blk_prep_run_kernel<4, true> : gpfq_blk_prep_run_kernel<4, false>)
                               : (sh.B == 2 ? (r64 ? gpfq_blk_prep_run_kernel<2, true> : gpfq_blk_prep_run_kernel<2, false>) : gpfq_blk_prep_run_kernel<1, false>);
        hipLaunchKernelGGL(prun, dim3((unsigned)((nrec + run - 1) / run)), dim3(256), 0, stream, a.X, a.Xq, a.ld, a.N, (int)a.m, sh.mp, nrec, run,
                           nrm32, wbase, wbase + blk_recs_bytes(nblk, sh), alpha, sym, zero16);
    } else {
        hipLaunchKernelGGL(prep, dim3((unsigned)nrec), dim3(256), 0, stream, a.X, a.Xq, a.ld, a.N, (int)a.m, sh.mp,
                           nrm32, wbase, wbase + blk_recs_bytes(nblk, sh), alpha, sym, 1, (int64_t)0);
    }
    e = hipGetLastError();
    if (e != hipSuccess || !do_run) return e;
    // (four neuron groups with one or two neurons per lane: the narrow forms of the fused matrix shape)
    if (sh.G == 4 && sh.NL < 4 && sh.NW == 7) {                    // (rows of at most 768 samples: blk_shape)
        if (sh.S == 16) return sh.NL == 1 ? launch_blk_inst<4, 16, 4, 7, 1>(a, sh, alpha, stream) : launch_blk_inst<4, 16, 4, 7, 2>(a, sh, alpha, stream);
        if (sh.S == 24) return sh.NL == 1 ? launch_blk_inst<4, 24, 4, 7, 1>(a, sh, alpha, stream) : launch_blk_inst<4, 24, 4, 7, 2>(a, sh, alpha, stream);
        return sh.NL == 1 ? launch_blk_inst<4, 32, 4, 7, 1>(a, sh, alpha, stream) : launch_blk_inst<4, 32, 4, 7, 2>(a, sh, alpha, stream);
    }
    if (sh.G == 4 && sh.NL < 4) {
        if (sh.S == 16) return sh.NL == 1 ? launch_blk_inst<4, 16, 4, 8, 1>(a, sh, alpha, stream) : launch_blk_inst<4, 16, 4, 8, 2>(a, sh, alpha, stream);
        if (sh.S == 24) return sh.NL == 1 ? launch_blk_inst<4, 24, 4, 8, 1>(a, sh, alpha, stream) : launch_blk_inst<4, 24, 4, 8, 2>(a, sh, alpha, stream);
        // (eleven sweep wavefronts buy these shapes nothing: 4096 x 512 on 1024 samples 1.77 against 1.62 ms, 4096 x 2048 1.94 / 1.95 in round 4;
        //  measured again on round 5's kernels, one neuron per lane: 1.326 against 1.285 -- profiles/r05/cluster_form.txt)
        return sh.NL == 1 ? launch_blk_inst<4, 32, 4, 8, 1>(a, sh, alpha, stream) : launch_blk_inst<4, 32, 4, 8, 2>(a, sh, alpha, stream);
    }
    if (sh.NL == 1) {                                              // one-neuron workgroups (layers of at most 128 neurons)
        if (sh.B == 4) return sh.S == 4 ? launch_blk_inst<1, 4, 4, 4, 1>(a, sh, alpha, stream) : launch_blk_inst<1, 8, 4, 8, 1>(a, sh, alpha, stream);
        if (sh.B == 2) return sh.S == 12 ? launch_blk_inst<1, 12, 2, 8, 1>(a, sh, alpha, stream) : launch_blk_inst<1, 16, 2, 8, 1>(a, sh, alpha, stream);
        if (sh.S == 24) return launch_blk_inst<1, 24, 1, 8, 1>(a, sh, alpha, stream);
        return sh.S == 32 ? launch_blk_inst<1, 32, 1, 8, 1>(a, sh, alpha, stream) : launch_blk_inst<1, 40, 1, 8, 1>(a, sh, alpha, stream);
    }
    if (sh.NL == 2) {                                              // two-neuron workgroups (layers of at most 512 neurons)
        if (sh.B == 4) return sh.S == 4 ? launch_blk_inst<1, 4, 4, 4, 2>(a, sh, alpha, stream) : launch_blk_inst<1, 8, 4, 8, 2>(a, sh, alpha, stream);
        if (sh.B == 2) return sh.S == 12 ? launch_blk_inst<1, 12, 2, 8, 2>(a, sh, alpha, stream) : launch_blk_inst<1, 16, 2, 8, 2>(a, sh, alpha, stream);
        if (sh.S == 24) return launch_blk_inst<1, 24, 1, 8, 2>(a, sh, alpha, stream);
        return sh.S == 32 ? launch_blk_inst<1, 32, 1, 8, 2>(a, sh, alpha, stream) : launch_blk_inst<1, 40, 1, 8, 2>(a, sh, alpha, stream);
    }
    if (sh.B == 1 && sh.S == 40) return launch_blk_inst<1, 40, 1>(a, sh, alpha, stream);
    if (sh.B == 1) {
        if (sh.G == 2) return sh.S == 64 ? launch_blk_inst<2, 64, 1, 11>(a, sh, alpha, stream) : launch_blk_inst<2, 48, 1, 11>(a, sh, alpha, stream);
        return sh.S == 32 ? launch_blk_inst<1, 32, 1>(a, sh, alpha, stream) : launch_blk_inst<1, 24, 1>(a, sh, alpha, stream);
    }
    if (sh.G == 4 && sh.S == 64) return launch_blk_inst<4, 64, 2, 11>(a, sh, alpha, stream);
    if (sh.G == 4 && sh.S == 48) return launch_blk_inst<4, 48, 2, 11>(a, sh, alpha, stream);
    if (sh.G == 4 && sh.NW == 11) {
        if (sh.S == 16) return launch_blk_inst<4, 16, 4, 11>(a, sh, alpha, stream);
        if (sh.S == 24) return launch_blk_inst<4, 24, 4, 11>(a, sh, alpha, stream);
        return launch_blk_inst<4, 32, 4, 11>(a, sh, alpha, stream);
    }
    if (sh.G == 4) {
        if (sh.S == 16) return launch_blk_inst<4, 16, 4>(a, sh, alpha, stream);
        if (sh.S == 24) return launch_blk_inst<4, 24, 4>(a, sh, alpha, stream);
        return launch_blk_inst<4, 32, 4>(a, sh, alpha, stream);
    }
    if (sh.G == 1) {
        if (sh.S == 8) return launch_blk_inst<1, 8, 4>(a, sh, alpha, stream);
        if (sh.S == 12) return launch_blk_inst<1, 12, 2>(a, sh, alpha, stream);
        return launch_blk_inst<1, 16, 2>(a, sh, alpha, stream);
    }
    if (sh.B == 4) return launch_blk_inst<2, 16, 4>(a, sh, alpha, stream);
    if (sh.S == 24) return launch_blk_inst<2, 24, 2>(a, sh, alpha, stream);
    return launch_blk_inst<2, 32, 2>(a, sh, alpha, stream);
}

}  // namespace gpfq
