// Pipelined GPFQ kernel: the dense default for rows of up to 2048 samples.
//
// Replaces _quantize_neuron_parallel / _quantize_filter2D_parallel_jit
// (scripts/quantized_network.py:91-121, :185-233); same contract and the same bits as
// gpfq_rows_kernel / gpfq_onchip_kernel, which it supersedes on the layers it takes.
//
// What bound the row-group kernel (profiles/r01/sq_counters_rows_kernel.txt, tools/ubench/issue_cycles.hip):
// instruction issue.  On gfx950 a float64-rate instruction (v_add_f64, v_fma_f64, v_cvt_f64_f32, and the packed
// float32 ones) costs 5.6 shader cycles of a SIMD and a wavefront issues at most one vector instruction per 7.5
// cycles, so a step costs (instructions per step) x ~5 cycles -- and more than half of the row-group kernel's
// instructions were not the element-wise floor (3 packed f32 + 2 cvt + 2 add + 2 fma per two samples of a neuron)
// but the per-step reduction and decision, executed once per wavefront for its two neurons, eight times per CU.
// Here that work is done ONCE per workgroup and step:
//
//  * Roles.  A workgroup owns NB = 4G neurons (16 at m <= 1024) and has eight SWEEP wavefronts and one DECISION
//    wavefront.  The sweep wavefronts split the SAMPLE axis: wavefront w owns PW_w sample pairs per k-lane for
//    all NB neurons (lane = (neuron group ng, k-lane kq), four neurons per lane, so every LDS value is read once for
//    four neurons); PW_w is uneven so that the SIMD that also hosts the decision wavefront gets less sweep work.
//    Per step a sweep wavefront applies the residual update of the previous step (the reference's element-wise
//    f32/f64 flow, :119), accumulates its share of the next dot product in the same pass, folds the four sums over
//    its k-lanes (packed v_permlane32/16_swap butterfly + two DPP rotations) and leaves NB partial sums in LDS.
//    The decision wavefront (lane = (neuron, sub-lane)) adds the eight partials, takes the decision of all NB
//    neurons in one instruction stream (alphabet search by counting, lane-local) and publishes (w_t, q_t).
//    One s_barrier per step.
//  * One-step look-ahead makes the two roles concurrent.  The sweep of slot t applies update t-1 and accumulates
//    D_{t+1} = <Xq_{t+1}, u_{t-1}>.  Decision t (same slot) uses D_t = <Xq_t, u_{t-2}> from the slot before plus
//    the contribution of step t-1's increment in closed form from two entries of the Gram band,
//    <Xq_t, u_{t-1}> = D_t + w_{t-1} <Xq_t, X_{t-1}> - q_{t-1} <Xq_t, Xq_{t-1}>, up to the float32 roundings of
//    that increment.  Those are bounded rigorously (|d_i - (w x_i - q xq_i)| <= 2^-23 (1 + 2^-24) (|w x_i| + |q xq_i|),
//    subnormal products 2^-149), so the predicted quotient is either farther from every decision boundary than the
//    bound -- the decision is then provably the reference's -- or the step is flagged and redone from the exact
//    dot products of :86/:89 on the completed residual (slow path: two extra barriers, about one decision in 10^6).
//    Rule (ii)'s |<Xq_t,u>| < 1e-10 test is certified the same way (exact when the increment is orthogonal to Xq_t
//    element by element, e.g. at t = 0 or on disjoint supports).  The residual itself is always updated by the exact
//    element-wise flow, so u is bit-identical.
//  * The pre-pass lays the operands out per SLOT: record t = [row statistics of step t][X_{t-1}][Xq_{t-1}]
//    [Xq_{t+1} as float64], zero-padded.  A tile of TS records is one contiguous block that the sweep wavefronts
//    stream into the other LDS buffer with global_load_lds_dwordx4 (LDS-DMA: no VGPRs, no ds_write, no conversion
//    in the hot loop) while the current one is worked on.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"
#include "gpfq_roles.hpp"

namespace gpfq {

namespace {

constexpr int kRecBytes = 128;

// Step record (first 128 bytes of a slot record); all float64.
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

__host__ __device__ constexpr int64_t pipe_rec_bytes(int64_t mp) { return kRecBytes + 16 * mp; }

// ---- pre-pass ---------------------------------------------------------------------------------
// One workgroup per slot record t in [0, nrec): rows t-1 (f32 copies), t (statistics) and t+1 (float64 copy of Xq)
// of the caller's matrices, zero-padded to mp samples.
__global__ void __launch_bounds__(256)
gpfq_pipe_prep_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int64_t N, int m, int mp,
                      const float *__restrict__ nrm32, char *__restrict__ recs)
{
    __shared__ double sm[7][4];
    const int64_t t = blockIdx.x;
    char *rb = recs + t * pipe_rec_bytes(mp);
    float  *ox  = reinterpret_cast<float *>(rb + kRecBytes);
    float  *oq  = ox + mp;
    double *od  = reinterpret_cast<double *>(rb + kRecBytes + 8 * (int64_t)mp);
    const bool has_prev = t >= 1 && t - 1 < N, has_cur = t < N, has_next = t + 1 < N;
    const float *px = X + (t - 1) * ld, *pq = Xq + (t - 1) * ld;
    const float *cx = X + t * ld, *cq = Xq + t * ld;
    const float *nq = Xq + (t + 1) * ld;
    double g = 0.0, a = 0.0, s1 = 0.0, h1 = 0.0, h2 = 0.0, e1 = 0.0, e2 = 0.0;
    for (int i = threadIdx.x; i < mp; i += 256) {
        const bool in = i < m;
        const float xp = (has_prev && in) ? px[i] : 0.f, qp = (has_prev && in) ? pq[i] : 0.f;
        ox[i] = xp;
        oq[i] = qp;
        od[i] = (double)((has_next && in) ? nq[i] : 0.f);
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

// LDS carve-up (byte offsets), shared by host and device.
struct PipeLds {
    int tile_bytes, tile_pitch, off_w, off_d, off_wq, off_x2, off_e, off_out, off_flag, total;
};
__host__ __device__ inline PipeLds pipe_lds(int mp, int nb, int ts)
{
    PipeLds L;
    L.tile_bytes = ts * (int)pipe_rec_bytes(mp);
    L.tile_pitch = (L.tile_bytes + 1023) & ~1023;               // the DMA moves whole 1 KiB pieces
    int o = 2 * L.tile_pitch;
    L.off_w = o;    o += 2 * nb * ts * 4;   o = (o + 15) & ~15;     // [2][NB][TS] f32   weights of the tile
    L.off_d = o;    o += 2 * kSweepWaves * nb * 8;                  // [2][8][NB] f64    partial dot products
    L.off_wq = o;   o += 2 * nb * 8;                                // [2][NB] (w, q) f32 of the step just decided
    L.off_x2 = o;   o += kSweepWaves * nb * 16;                     // [8][NB] (f64, f64) exact partials (slow path)
    L.off_e = o;    o += 68 * 8;                                    // [2 + 64 + 2] f64  -inf, -inf, alphabet, +inf, +inf
    L.off_out = o;  o += 2 * nb * ts * 8;   o = (o + 15) & ~15;     // [2][NB][TS] (idx i32, q f32) until the tile's flush
    L.off_flag = o; o += 16;                                        // [2] i32 flagged decisions of the slot
    L.total = o;
    return L;
}

struct PipeK {
    const char *recs;
    const float *X, *Xq;
    int64_t ld;
    const float *Wt;
    int64_t ldw;
    int64_t N, C;
    int m, TS, M, zero_idx;
    int flags;             // tuning: bit 0 = spread the DMA issue over the slots of a tile (measured slower: an LDS-DMA
                           // piece costs its wavefront 100+ cycles of issue wherever it stands)
    int8_t *qidx;
    float *Qt;
    double *resid, *u_out;
    unsigned long long *fallback_count;
};

// ---- sweep wavefront -------------------------------------------------------------------------------
template <int G, int PW, int MP>
__device__ __forceinline__ void sweep_role(const PipeK &K, char *lds_generic, const PipeLds &L, int wave, int lane, int pbase)
{
    constexpr int NB = 4 * G, KQ = 64 / G;
    constexpr int RB = (int)pipe_rec_bytes(MP);
    lchar *lds = (lchar *)lds_generic;
    const int ng = lane & (G - 1), kq = lane / G, row = lane >> 4;
    const bool writer = (lane & 15 & ~(G - 1)) == 0;             // one lane per (row, ng) publishes the folded sums
    const int nloc = 4 * ng;                                      // first of this lane's four neurons
    const int64_t jbase = (int64_t)blockIdx.x * NB;
    const unsigned ldsT_addr = lds_addr(lds_generic), ldsW_addr = lds_addr(lds_generic + L.off_w);
    const int64_t N = K.N;
    const int TS = K.TS;
    const int64_t nrec = N + 1;                                   // slots 0..N (slot N only applies update N-1)
    const int ntiles = (int)((nrec + TS - 1) / TS);
    // per-lane LDS byte offsets that do not change: operand slices inside a record, the (w, q) quadruple, partial sums
    const int o_x  = kRecBytes + 8 * (pbase + kq);               // float2 x   [pair]
    const int o_q  = o_x + 4 * MP;                                // float2 xq  [pair]
    const int o_d  = kRecBytes + 8 * MP + 16 * (pbase + kq);     // double2 xqd[pair]
    const int o_wq = L.off_wq + nloc * 8;
    const int o_dw = L.off_d + (wave * NB + nloc + row) * 8;
    const int o_x2 = L.off_x2 + (wave * NB + nloc + row) * 16;

    double u[4][2 * PW];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 2 * PW; ++e) u[i][e] = 0.0;          // zeros(m), :115

    // LDS-DMA of tile k, 1 KiB pieces: this wavefront's pieces are wave, wave + 8, ...; `part` of `parts` (the
    // pieces are spread over the slots of the tile before so that no slot starts with a burst of issue work)
    const int npieces = (L.tile_bytes + 1023) >> 10;              // the last piece may run into the next record: harmless
    auto load_tile = [&](int k, int part, int parts) {
        const char *src = K.recs + (int64_t)k * L.tile_bytes + lane * 16;
        const unsigned dst = ldsT_addr + (unsigned)(k & 1) * (unsigned)L.tile_pitch;
        int j = 0;
        for (int pc = wave; pc < npieces; pc += kSweepWaves, ++j)
            if (j % parts == part) glds16(src + ((size_t)pc << 10), dst + ((unsigned)pc << 10));
        if (part == 0) {
            const unsigned dw = ldsW_addr + (unsigned)((k & 1) * NB * TS * 4);
            for (int i0 = wave * 64; i0 < NB * TS; i0 += kSweepWaves * 64) {
                const int i = i0 + lane;
                const int n = i / TS, s = i - n * TS;
                const int64_t jn = jbase + n, t = (int64_t)k * TS + s;
                if (i < NB * TS && jn < K.C && t < N) glds4(K.Wt + jn * K.ldw + t, dw + 4 * (unsigned)i0);
            }
        }
    };

    load_tile(0, 0, 1);
    dma_wait();
    slot_barrier();

    for (int k = 0; k < ntiles; ++k) {
        const int tbase = (k & 1) * L.tile_pitch;
        const int ts = (int)((nrec - (int64_t)k * TS) < TS ? (nrec - (int64_t)k * TS) : TS);
        const int parts = (ts > 1 && (K.flags & 1)) ? ts - 1 : 1; // next tile's DMA: in slot 0, or spread over slots 0 .. ts-2
        for (int s = 0; s < ts; ++s) {
            const int64_t t = (int64_t)k * TS + s;
            const int pb = (int)((t - 1) & 1);                    // buffer of step t-1's (w, q) and flag
            const int rb = tbase + s * RB;
            const int flag = lds_ld<int>(lds, L.off_flag + 4 * pb);
            float4 wq01 = lds_ld<float4>(lds, o_wq + pb * NB * 8), wq23 = lds_ld<float4>(lds, o_wq + pb * NB * 8 + 16);
            if (k + 1 < ntiles && s < parts) load_tile(k + 1, s, parts);   // streams into the other buffer meanwhile
            if (__builtin_amdgcn_readfirstlane(flag) != 0) {
                // slow path: step t-1 has decisions the look-ahead could not certify.  u is u_{t-2} here, exactly
                // what :86 / :89 take: this wavefront's share of <Xq, u> and <Xq, u + f32(w X)> for its neurons.
                const int64_t tq = t - 1;
                const float wv[4] = {wq01.x, wq01.z, wq23.x, wq23.z};
                double eu[4] = {0.0, 0.0, 0.0, 0.0}, ew[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int p = 0; p < PW; ++p)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int i = 2 * (pbase + p * KQ + kq) + e;
                        float xq = 0.f, xx = 0.f;
                        if (i < K.m) { xq = K.Xq[tq * K.ld + i]; xx = K.X[tq * K.ld + i]; }
#pragma unroll
                        for (int n = 0; n < 4; ++n) {
                            eu[n] = fma((double)xq, u[n][2 * p + e], eu[n]);
                            ew[n] = fma((double)xq, u[n][2 * p + e] + (double)__fmul_rn(wv[n], xx), ew[n]);
                        }
                    }
                const double vu = fold_klanes<G>(eu), vw = fold_klanes<G>(ew);
                if (writer) lds_st<double2>(lds, o_x2, make_double2(vu, vw));
                slot_barrier();                                   // partials published
                slot_barrier();                                   // decisions of step t-1 rewritten
                wq01 = lds_ld<float4>(lds, o_wq + pb * NB * 8);
                wq23 = lds_ld<float4>(lds, o_wq + pb * NB * 8 + 16);
            }
            const float wv[4] = {wq01.x, wq01.z, wq23.x, wq23.z}, qv[4] = {wq01.y, wq01.w, wq23.y, wq23.w};

            // ---- sweep: u_{t-1} = u_{t-2} + increment of step t-1; acc = share of <Xq_{t+1}, u_{t-1}> ----
            double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                // operands of the pair: X_{t-1}, Xq_{t-1}, Xq_{t+1} (f64)  (requesting all pairs of the slot up front
                // needs 40 more registers than the 168 a nine-wavefront workgroup leaves per lane: spills)
                const float2 x2 = lds_ld<float2>(lds, rb + o_x + 8 * p * KQ), q2 = lds_ld<float2>(lds, rb + o_q + 8 * p * KQ);
                const double2 d2 = lds_ld<double2>(lds, rb + o_d + 16 * p * KQ);
                const pk2 xv = {x2.x, x2.y}, qx = {q2.x, q2.y};
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    // f32 products and subtraction on two samples at once (v_pk_mul_f32 / v_pk_add_f32): each half rounds
                    // exactly as the scalar instruction (no contraction: -ffp-contract=off)
                    const pk2 d = pk2{wv[n], wv[n]} * xv - pk2{qv[n], qv[n]} * qx;
                    u[n][2 * p]     += (double)d.x;
                    u[n][2 * p + 1] += (double)d.y;
                    acc[n] = fma(d2.x, u[n][2 * p], acc[n]);
                    acc[n] = fma(d2.y, u[n][2 * p + 1], acc[n]);
                }
            }
            if (t < N) {
                const double v = fold_klanes<G>(acc);
                if (writer) lds_st<double>(lds, o_dw + (int)((t + 1) & 1) * kSweepWaves * NB * 8, v);
            }
            if (s + 1 == ts) dma_wait();                          // this wavefront's share of the next tile has landed
            slot_barrier();
        }
    }

    // ---- epilogue: residual norms through the same partial-sum path, residual vectors straight to memory ----
    if (K.resid) {
        double ss[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            double q = 0.0;
#pragma unroll
            for (int e = 0; e < 2 * PW; ++e) q = fma(u[n][e], u[n][e], q);
            ss[n] = q;
        }
        const double v = fold_klanes<G>(ss);
        if (writer) lds_st<double>(lds, o_dw, v);
    }
    slot_barrier();
    if (K.u_out) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int64_t jn = jbase + nloc + n;
            if (jn < K.C) {
#pragma unroll
                for (int p = 0; p < PW; ++p)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int i = 2 * (pbase + p * KQ + kq) + e;
                        if (i < K.m) K.u_out[jn * (int64_t)K.m + i] = u[n][2 * p + e];
                    }
            }
        }
    }
}

// ---- decision wavefront ------------------------------------------------------------------------------
template <int G, int MP>
__device__ __forceinline__ void decision_role(const PipeK &K, char *lds_generic, const PipeLds &L, int lane)
{
    constexpr int NB = 4 * G, R = 64 / NB;
    constexpr int RB = (int)pipe_rec_bytes(MP);
    lchar *lds = (lchar *)lds_generic;
    const int n = lane / R, r = lane % R;                         // neuron of the workgroup, sub-lane
    const int64_t jn = (int64_t)blockIdx.x * NB + n;
    const bool active = jn < K.C;
    const int64_t N = K.N;
    const int TS = K.TS, M = K.M;
    const int64_t nrec = N + 1;
    const int ntiles = (int)((nrec + TS - 1) / TS);
    const double kNaN = __longlong_as_double(0x7ff8000000000000LL);
    // per-lane LDS byte offsets that do not change
    const int o_d   = L.off_d + (r * NB + n) * 8;                 // partial sums of sweep wavefronts r, r + R, ...
    const int o_x2  = L.off_x2 + (r * NB + n) * 16;
    const int o_w   = L.off_w + n * TS * 4;
    const int o_wq  = L.off_wq + n * 8;
    const int o_out = L.off_out + n * TS * 8;

    // alphabet members r, r + R, ... of this sub-lane (NaN beyond M: never counted); larger alphabets loop over LDS
    const bool in_regs = M <= 4 * R;
    double am[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) am[q] = (r + q * R < M) ? lds_ld<double>(lds, L.off_e + 8 * (2 + r + q * R)) : kNaN;

    float wprev = 0.f, qprev = 0.f;
    bool  redo_prev = false;
    double nrm_prev = 0.0;
    unsigned long long n_fallback = 0;
    int64_t flushed = -1;

    // outputs of a tile wait in LDS until the first slot of the next tile (a flagged step is rewritten before)
    auto flush = [&](int64_t k) {
        const int ob = L.off_out + (int)(k & 1) * NB * TS * 8;
        for (int e = lane; e < NB * TS; e += 64) {
            const int nn = e / TS, s = e - nn * TS;
            const int64_t j = (int64_t)blockIdx.x * NB + nn, t = k * TS + s;
            if (j < K.C && t < N) {
                const int2 v = lds_ld<int2>(lds, ob + e * 8);
                if (K.qidx) K.qidx[j * N + t] = (int8_t)v.x;
                if (K.Qt)   K.Qt[j * N + t]   = __int_as_float(v.y);
            }
        }
    };

    slot_barrier();                                               // (tile 0 landed)

    for (int k = 0; k < ntiles; ++k) {
        const int tbase = (k & 1) * L.tile_pitch;
        const int wbase = (k & 1) * NB * TS * 4, obase = (k & 1) * NB * TS * 8;
        const int ts = (int)((nrec - (int64_t)k * TS) < TS ? (nrec - (int64_t)k * TS) : TS);
        for (int s = 0; s < ts; ++s) {
            const int64_t t = (int64_t)k * TS + s;
            const int pb = (int)((t - 1) & 1), cb = (int)(t & 1);
            const int flag = lds_ld<int>(lds, L.off_flag + 4 * pb);
            // requests of the slot that depend on nothing decided here: D_t partials, record t, w_t
            double part[kSweepWaves / R];
#pragma unroll
            for (int q = 0; q < kSweepWaves / R; ++q) part[q] = lds_ld<double>(lds, o_d + (cb * kSweepWaves + q * R) * NB * 8);
            const int rb = tbase + s * RB;
            const double2 r01 = lds_ld<double2>(lds, rb), r23 = lds_ld<double2>(lds, rb + 16), r45 = lds_ld<double2>(lds, rb + 32),
                          r67 = lds_ld<double2>(lds, rb + 48), r89 = lds_ld<double2>(lds, rb + 64);
            const float wg = lds_ld<float>(lds, o_w + wbase + 4 * s);
            if (__builtin_amdgcn_readfirstlane(flag) != 0) {
                // slow path: redo the flagged decisions of step t-1 from the exact dot products (:86, :89)
                slot_barrier();                                   // partials published
                double du = 0.0, dw = 0.0;
#pragma unroll
                for (int q = 0; q < kSweepWaves / R; ++q) {
                    const double2 v = lds_ld<double2>(lds, o_x2 + q * R * NB * 16);
                    du += v.x; dw += v.y;
                }
                du = sub_sum<R>(du); dw = sub_sum<R>(dw);
                if (redo_prev) {
                    const double te = dw / (nrm_prev * nrm_prev);
                    const double t2 = fabs(du) < 1e-10 ? (double)wprev : te;
                    int bi = 0;
                    double bq = lds_ld<double>(lds, L.off_e + 16), bdist = fabs(bq - t2);
                    for (int kk = 1; kk < M; ++kk) {
                        const double ak = lds_ld<double>(lds, L.off_e + 8 * (2 + kk)), dk = fabs(ak - t2);
                        if (dk < bdist) { bdist = dk; bi = kk; bq = ak; }
                    }
                    qprev = (float)bq;
                    if (r == 0) {
                        const int64_t tq = t - 1;
                        lds_st<int2>(lds, o_out + (int)((tq / TS) & 1) * NB * TS * 8 + (int)(tq % TS) * 8, make_int2(bi, __float_as_int(qprev)));
                        lds_st<float2>(lds, o_wq + pb * NB * 8, make_float2(wprev, qprev));
                        if (active) ++n_fallback;
                    }
                }
                slot_barrier();                                   // decisions rewritten
            }
            if (s == 0 && k >= 1) { flush(k - 1); flushed = k - 1; }   // the previous tile's outputs are final now
            if (t < N) {
                // ---- D_t: the partial sums of the previous slot ----
                double D = part[0];
#pragma unroll
                for (int q = 1; q < kSweepWaves / R; ++q) D += part[q];
                D = sub_sum<R>(D);

                // ---- decision t (:83-89, :57) for neuron n; branch-free ----
                const double rden = r01.x, rG = r01.y, rcb = r23.x, rca = r23.y, rH1 = r45.x, rH2 = r45.y;
                const double rE1 = r67.x, rE2 = r67.y, rEa = r89.x, nrm = r89.y;
                const bool rule1 = nrm < 1e-16;                                      // rule (i): literal 0
                const double wd = (double)wg, wpd = (double)wprev, qpd = (double)qprev;
                const double corr = fma(wpd, rH1, -(qpd * rH2));
                const double du = D + corr;                                          // predicted <Xq_t, u_{t-1}>
                const bool   inc = ((__float_as_uint(wprev) | __float_as_uint(qprev)) << 1) != 0u;
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
                // p = number of members below t (ascending alphabet), counted by the R sub-lanes
                int c = 0;
                if (in_regs) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) c += (am[q] < tt) ? 1 : 0;
                } else {
                    for (int kk = r; kk < M; kk += R) c += (lds_ld<double>(lds, L.off_e + 8 * (2 + kk)) < tt) ? 1 : 0;
                }
                const int p = sub_sumi<R>(c);
                // neighbours of t in the table -inf, -inf, a_0 .. a_{M-1}, +inf, +inf (all four requested at once)
                const int oe = L.off_e + 8 * p;
                const double lolo = lds_ld<double>(lds, oe), lo = lds_ld<double>(lds, oe + 8), hi = lds_ld<double>(lds, oe + 16),
                             hihi = lds_ld<double>(lds, oe + 24);
                const double d_lo = fabs(lo - tt), d_hi = fabs(hi - tt), d_ll = fabs(lolo - tt), d_hh = fabs(hihi - tt);
                const bool at0 = p == 0, atM = p == M;
                const bool use_hi = at0 | (!atM & !(d_lo <= d_hi));                  // tie -> lower index
                const int    idx_l = use_hi ? p : p - 1;
                const double q_l   = use_hi ? hi : lo;
                // twice the distance of t from the boundary between the winner and the runner-up
                const double m2_in = fabs(d_hi - d_lo), m2_lo = d_hh - d_hi, m2_hi = d_ll - d_lo;
                const double m2 = at0 ? m2_lo : (atM ? m2_hi : m2_in);
                // first-index rule: a lower member at the same distance would win instead (repeated members)
                const bool plateau = !use_hi & (p >= 2) & !(d_ll > d_lo);
                const bool cert = !plateau & (msq | (m2 > delta2)) & sure;
                const float q32 = rule1 ? 0.f : (float)q_l;
                const int   idx = rule1 ? K.zero_idx : idx_l;
                const bool  redo = !rule1 & !cert;

                if (r == 0) {
                    lds_st<float2>(lds, o_wq + cb * NB * 8, make_float2(wg, q32));
                    lds_st<int2>(lds, o_out + obase + 8 * s, make_int2(idx, __float_as_int(q32)));
                }
                const unsigned long long rb_ = __ballot(redo & active & (r == 0));
                if (lane == 0) lds_st<int>(lds, L.off_flag + 4 * cb, (int)__popcll(rb_));
                wprev = wg; qprev = q32; redo_prev = redo; nrm_prev = nrm;
            }
            slot_barrier();
        }
    }
    if ((N - 1) / TS > flushed) flush((N - 1) / TS);              // the tile of the last step

    if (K.fallback_count && n_fallback) atomicAdd(K.fallback_count, n_fallback);   // rare
    slot_barrier();                                               // residual-norm partials published
    if (K.resid) {
        double tot = 0.0;
#pragma unroll
        for (int q = 0; q < kSweepWaves / R; ++q) tot += lds_ld<double>(lds, o_d + q * R * NB * 8);
        tot = sub_sum<R>(tot);
        if (active && r == 0) K.resid[jn] = sqrt(tot);
    }
}

}  // namespace

// G neuron groups per sweep wavefront (4G neurons per workgroup), S sample pairs per k-lane over the eight sweep
// wavefronts: rows of MP = (128 / G) * S samples.
template <int G, int S>
__global__ void __launch_bounds__(64 * (kSweepWaves + 1))
gpfq_pipe_kernel(PipeK K, AlphabetArg A)
{
    constexpr int NB = 4 * G, KQ = 64 / G, MP = 2 * KQ * S;
    using PS = PairSplit<S>;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const PipeLds L = pipe_lds(MP, NB, K.TS);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- LDS initialisation: zero weights / partials / (w, q) / flags, alphabet table with sentinels ----
    for (int i = tid; i < (L.off_e - L.off_w) / 4; i += blockDim.x) reinterpret_cast<int *>(lds + L.off_w)[i] = 0;
    for (int i = tid; i < (L.total - L.off_out) / 4; i += blockDim.x) reinterpret_cast<int *>(lds + L.off_out)[i] = 0;
    if (tid < 68) {
        const double kInf = __longlong_as_double(0x7ff0000000000000LL);
        const int k = tid - 2;
        reinterpret_cast<double *>(lds + L.off_e)[tid] = k < 0 ? -kInf : (k < A.M ? A.a[k] : kInf);
    }
    __syncthreads();

    if (wave < kSweepWaves) {
        int pbase = 0;
#pragma unroll
        for (int w = 0; w < kSweepWaves; ++w) pbase += (w < wave) ? KQ * PS::pw[w] : 0;
        const int pw = PS::pw[wave & 7];
        // one instantiation per distinct pair count of the split
        if (pw == 1) { if constexpr (S == 16 || S == 24) sweep_role<G, 1, MP>(K, lds, L, wave, lane, pbase); }
        else if (pw == 2) sweep_role<G, 2, MP>(K, lds, L, wave, lane, pbase);
        else if (pw == 3) sweep_role<G, 3, MP>(K, lds, L, wave, lane, pbase);
        else if (pw == 4) { if constexpr (S >= 24) sweep_role<G, 4, MP>(K, lds, L, wave, lane, pbase); }
        else { if constexpr (S == 32) sweep_role<G, 5, MP>(K, lds, L, wave, lane, pbase); }
    } else {
        decision_role<G, MP>(K, lds, L, lane);
    }
}

// ---- host side ------------------------------------------------------------------------------------
struct PipeShape { int G, S, mp; };

static PipeShape pipe_shape(int64_t m)
{
    // smallest padded row that holds m, more neurons per workgroup first
    static const PipeShape shapes[] = {{16, 16, 128}, {16, 24, 192}, {16, 32, 256}, {8, 24, 384}, {8, 32, 512},
                                       {4, 24, 768}, {4, 32, 1024}, {2, 24, 1536}, {2, 32, 2048}};
    for (const PipeShape &s : shapes)
        if (m <= s.mp) return s;
    return {0, 0, 0};
}

static int pipe_tile_steps(const PipeShape &sh, int64_t N)
{
    int ts = 16;
    while (ts > 1 && pipe_lds(sh.mp, 4 * sh.G, ts).total > 158 * 1024) --ts;
    if (ts > N + 1) ts = (int)(N + 1);
    return ts < 1 ? 1 : ts;
}

bool pipe_supported(const PipeArgs &a)
{
    if (pipe_shape(a.m).G == 0 || a.N < 1 || a.m < 1) return false;
    if (a.A.M > 64 || !a.A.ascending) return false;
    return a.N + 64 < (1LL << 31) / 64;             // tile offsets stay in int range
}

size_t pipe_workspace_bytes(int64_t N, int64_t m)
{
    const PipeShape sh = pipe_shape(m);
    if (!sh.G) return 0;
    // slot records 0..N, padded to whole tiles of up to 16 records
    return (size_t)(N + 1 + 16) * (size_t)pipe_rec_bytes(sh.mp);
}

template <int G, int S>
static hipError_t launch_pipe_inst(const PipeArgs &a, const PipeShape &sh, hipStream_t stream)
{
    constexpr int NB = 4 * G;
    int ts = pipe_tile_steps(sh, a.N);
    if (a.ts_override > 0 && a.ts_override < ts) ts = a.ts_override;
    const PipeLds L = pipe_lds(sh.mp, NB, ts);
    const unsigned grid = (unsigned)((a.C + NB - 1) / NB);
    auto *kern = gpfq_pipe_kernel<G, S>;
    hipError_t e = ensure_dynamic_lds((const void *)kern, (size_t)L.total);
    if (e != hipSuccess) return e;
    PipeK K;
    K.recs = static_cast<const char *>(a.workspace); K.X = a.X; K.Xq = a.Xq; K.ld = a.ld; K.Wt = a.Wt; K.ldw = a.ldw;
    K.N = a.N; K.C = a.C; K.m = (int)a.m; K.TS = ts; K.M = a.A.M; K.zero_idx = a.A.zero_idx; K.flags = a.variant & 1;
    K.qidx = a.qidx; K.Qt = a.Qt; K.resid = a.resid; K.u_out = a.u_out; K.fallback_count = a.fallback_count;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (kSweepWaves + 1)), (size_t)L.total, stream, K, a.A);
    return hipGetLastError();
}

hipError_t launch_pipe(const PipeArgs &a, hipStream_t stream)
{
    const PipeShape sh = pipe_shape(a.m);
    if (!sh.G) return hipErrorInvalidValue;
    // pre-pass: slot records (statistics + padded operand rows), tiles padded with zero records
    const int64_t nrec = a.N + 1 + 16;
    hipLaunchKernelGGL(gpfq_pipe_prep_kernel, dim3((unsigned)nrec), dim3(256), 0, stream, a.X, a.Xq, a.ld, a.N, (int)a.m, sh.mp,
                       a.nrm32, static_cast<char *>(a.workspace));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    switch (sh.G * 100 + sh.S) {
    case 1616: return launch_pipe_inst<16, 16>(a, sh, stream);
    case 1624: return launch_pipe_inst<16, 24>(a, sh, stream);
    case 1632: return launch_pipe_inst<16, 32>(a, sh, stream);
    case 824:  return launch_pipe_inst<8, 24>(a, sh, stream);
    case 832:  return launch_pipe_inst<8, 32>(a, sh, stream);
    case 424:  return launch_pipe_inst<4, 24>(a, sh, stream);
    case 432:  return launch_pipe_inst<4, 32>(a, sh, stream);
    case 224:  return launch_pipe_inst<2, 24>(a, sh, stream);
    default:   return launch_pipe_inst<2, 32>(a, sh, stream);
    }
}

}  // namespace gpfq
