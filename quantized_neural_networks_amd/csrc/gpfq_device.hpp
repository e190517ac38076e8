// Device-side helpers shared by the GPFQ kernels (gfx950 / CDNA4, wave64 only).
//
// Numerics contract (DESIGN.md "numerics"): the per-element arithmetic that defines the
// residual u is reproduced exactly as the reference's legacy-NumPy expression does it
// (scripts/quantized_network.py:89, :119):
//     p = f32(w * X_i)            float32 product, rounded once
//     r = f32((float)q * Xq_i)    q rounded to float32 first, float32 product
//     d = f32(p - r)              float32 subtraction
//     u_i += (double)d            float64 accumulation
// The translation unit is compiled with -ffp-contract=off and the f32 steps use the
// never-contracted __fmul_rn/__fsub_rn intrinsics, so no v_fma_f32 / v_fmac_f32 can fuse them.
// Reductions (dot products, norms) accumulate in float64 with explicit fma(); their summation
// order differs from BLAS (as any two BLAS builds differ from each other) at the 1e-16 level.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gpfq {

constexpr int kWave = 64;

// ---- cross-lane plumbing -------------------------------------------------------------------
// dpp_ctrl encodings (CDNA ISA): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
// bound_ctrl:1 makes lanes without a source read 0, so no "old" operand has to be initialised.
template <int CTRL>
__device__ __forceinline__ double dpp_fetch0(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double x, int lane)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float readlane_f32(float x, int lane)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

// Inclusive scan inside each row of 16 lanes: lane 15 of a row ends with the row total.
__device__ __forceinline__ double row_scan(double x)
{
    x += dpp_fetch0<0x111>(x);   // row_shr:1
    x += dpp_fetch0<0x112>(x);   // row_shr:2
    x += dpp_fetch0<0x114>(x);   // row_shr:4
    x += dpp_fetch0<0x118>(x);   // row_shr:8
    return x;
}

// Sum of x over the 64 lanes of the wavefront, returned wave-uniform (read from lane 63).
// 4 in-row scan steps + 2 row broadcasts; fixed order, so results are run-to-run identical.
__device__ __forceinline__ double wave_sum(double x)
{
    x = row_scan(x);
    x += dpp_fetch0<0x142>(x);   // row_bcast:15 -> lane 31 = r0+r1, lane 63 = r2+r3 (+ unused others)
    x += dpp_fetch0<0x143>(x);   // row_bcast:31 -> lane 63 = r0+r1+r2+r3
    return readlane_f64(x, 63);
}

// Two sums for the price of one: v_permlane32_swap folds the upper half-wave of `a` onto its lower
// half and the lower half-wave of `b` onto its upper half, after which lanes 0-31 hold a's 32
// pair-sums and lanes 32-63 hold b's; one row scan + one row_bcast:15 finishes both.
__device__ __forceinline__ void wave_sum2(double a, double b, double &sum_a, double &sum_b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
    double x = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    x = row_scan(x);
    x += dpp_fetch0<0x142>(x);   // lane 31 = rows 0+1 (a), lane 63 = rows 2+3 (b)
    sum_a = readlane_f64(x, 31);
    sum_b = readlane_f64(x, 63);
}

// ---- alphabet rounding -------------------------------------------------------------------
// nearest(): index of the FIRST minimum of |alphabet[k] - t| evaluated in float64 -- the result of
// alphabet[argmin(abs(alphabet - t))], scripts/quantized_network.py:57.
//
// Lane k < M holds a_lane = alphabet[k]; lanes >= M hold NaN (every comparison false).
// Ascending alphabets (the only kind rad*linspace(-1,1,M) with rad >= 0 produces): since
// fl(a - t) is monotone in a, the global minimum of |fl(a_k - t)| is attained next to the
// position of t, so it is min(d[p-1], d[p]) with p = #{k : a_k < t}; the first index attaining
// that value is then found with one ballot.  Non-ascending alphabets take the plain scan.
// t is wave-uniform; the result is wave-uniform.
__device__ __forceinline__ int nearest(double t, double a_lane, int M, bool ascending)
{
    const double d = fabs(a_lane - t);
    if (ascending) {
        const unsigned long long lt = __ballot(a_lane < t);
        const int p = __popcll(lt);
        const int lo = p > 0 ? p - 1 : 0;
        const int hi = p < M ? p : M - 1;
        const double dlo = readlane_f64(d, lo);
        const double dhi = readlane_f64(d, hi);
        const double dmin = dlo <= dhi ? dlo : dhi;
        const unsigned long long eq = __ballot(d == dmin);
        return eq ? (int)__ffsll((long long)eq) - 1 : 0;   // all-NaN distances: np.argmin -> 0
    }
    int best = 0;
    double dbest = readlane_f64(d, 0);
    for (int k = 1; k < M; ++k) {
        const double dk = readlane_f64(d, k);
        if (dk < dbest) { dbest = dk; best = k; }
    }
    return best;
}

// nearest() plus the distance of t from the closest decision boundary (the midpoint between the
// chosen member and the runner-up), used to certify decisions taken from an approximated t.
// margin < 0 means "cannot certify" (non-ascending alphabet, ties/plateaus).
__device__ __forceinline__ int nearest_margin(double t, double a_lane, int M, bool ascending, double &margin)
{
    margin = -1.0;
    if (!ascending) return nearest(t, a_lane, M, false);
    const double d = fabs(a_lane - t);
    const unsigned long long lt = __ballot(a_lane < t);
    const int p = __popcll(lt);
    const int lo = p > 0 ? p - 1 : 0;
    const int hi = p < M ? p : M - 1;
    const double dlo = readlane_f64(d, lo);
    const double dhi = readlane_f64(d, hi);
    const double dmin = dlo <= dhi ? dlo : dhi;
    const unsigned long long eq = __ballot(d == dmin);
    const int idx = eq ? (int)__ffsll((long long)eq) - 1 : 0;
    if (lo != hi) {
        if (idx == lo || idx == hi) margin = 0.5 * fabs(dlo - dhi);
    } else if (M == 1) {
        margin = __longlong_as_double(0x7ff0000000000000LL);
    } else if (idx == lo) {
        const double dn = readlane_f64(d, lo == 0 ? 1 : M - 2);      // t lies outside the alphabet's range
        margin = 0.5 * (dn - dmin);
    }
    return idx;
}

struct Decision {
    int    idx;   // alphabet index, or zero_idx for the literal 0 of rule (i)
    double q;     // float64 value the reference stores in q[t]
};

// _quantize_weight_parallel, scripts/quantized_network.py:59-89, given the two wave-reduced dot
// products.  dot_u  = <Xq_t, u>,  dot_uw = <Xq_t, u + w*X_t>,  nrm = f32-rounded ||Xq_t||.
__device__ __forceinline__ Decision decide(float w, float nrm, double dot_u, double dot_uw,
                                           double a_lane, int M, int zero_idx, bool ascending)
{
    Decision r;
    if ((double)nrm < 1e-16) {                                  // :83-84
        r.idx = zero_idx;
        r.q = 0.0;
        return r;
    }
    double t;
    if (fabs(dot_u) < 1e-10) t = (double)w;                     // :86-87
    else t = dot_uw / ((double)nrm * (double)nrm);              // :89 (IEEE f64 division)
    r.idx = nearest(t, a_lane, M, ascending);
    r.q = readlane_f64(a_lane, r.idx);
    return r;
}

// ---- "has this channel a negative activation?" ----------------------------------------------------
// Running maximum of the raw bit patterns: every negative non-zero float is above 0x80000000 (= -0.0,
// which counts as non-negative here), every non-negative one below.
__device__ __forceinline__ void neg_track(unsigned &acc, float v) { acc = max(acc, __float_as_uint(v)); }
__device__ __forceinline__ void neg_track(unsigned &acc, const float4 &v)
{
    acc = max(max(acc, max(__float_as_uint(v.x), __float_as_uint(v.y))), max(__float_as_uint(v.z), __float_as_uint(v.w)));
}
__device__ __forceinline__ bool neg_seen(unsigned acc) { return acc > 0x80000000u; }

// ---- Gram accumulators of the 3x3 conv case (gpfq_gram.hip, gpfq_gram_image.hip) -----------------
// One column (sample) of the 9 patch rows: q[t] = Xq_t, x[s] = X_s in float64 (products of two float32
// values are exact in float64).  Only what the decide step reads: the lower triangle s <= t of
// <Xq_t,X_s> and <Xq_t,Xq_s> (packed index t(t+1)/2 + s), and the squared norms of the X rows.
struct Gram9 {
    double g[45][2];
    double nx[9];
};

__device__ __forceinline__ void gram9_zero(Gram9 &a)
{
#pragma unroll
    for (int i = 0; i < 45; ++i) { a.g[i][0] = 0.0; a.g[i][1] = 0.0; }
#pragma unroll
    for (int s = 0; s < 9; ++s) a.nx[s] = 0.0;
}

__device__ __forceinline__ void gram9_add(Gram9 &a, const double (&q)[9], const double (&x)[9])
{
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        a.nx[s] = fma(x[s], x[s], a.nx[s]);
#pragma unroll
        for (int t = s; t < 9; ++t) {
            a.g[t * (t + 1) / 2 + s][0] = fma(q[t], x[s], a.g[t * (t + 1) / 2 + s][0]);
            a.g[t * (t + 1) / 2 + s][1] = fma(q[t], q[s], a.g[t * (t + 1) / 2 + s][1]);
        }
    }
}

template <int CAP>
struct AlphabetT {
    double a[CAP];
    int    M;
    int    zero_idx;
    int    ascending;
};
using AlphabetArg = AlphabetT<64>;     // by-value kernel argument of every kernel family (bits <= 6)
using AlphabetBig = AlphabetT<256>;    // GPFQ_MAX_ALPHABET: the kernels that take 65..256 members (bits 7, 8) write int16 indices

// The layer alphabet IN DEVICE MEMORY (round 6; include/gpfq.h: GPFQ_DEVICE_ALPHABET_BYTES, gpfq_layer_alphabet_device): everything a
// launch of the block-pipelined dense kernel needs of `rad * linspace(-1, 1, M)` (scripts/quantized_network.py:544-545), formed on the
// device from the float32 median of |W| -- or stored from a host alphabet -- by gpfq_alphabet_setup_kernel (gpfq_blk.hip), so that no
// launch waits for the radius to cross to the host.  `ok` = 0: not a strictly ascending arithmetic progression the chain of decisions
// can index by arithmetic (rad = 0, infinite or NaN: the median of a kernel that is mostly zeros): the kernel then writes nothing and
// raises the call's alphabet word (workspace bytes 12..15), and the caller reruns the layer with a host alphabet.
struct DevAlphabet {
    double rad;                          // float64(alphabet_scalar) * float64(float32 median): the reference's legacy-NumPy product (:544); NaN when stored from a host alphabet
    double a0, step, inv, c0;            // member k = a0 + k step; inv = 1 / step, c0 = -a0 / step (blk_uniform, gpfq_blk.hip)
    unsigned long long plus, minus;      // bit k: float32(a[k]) is the float32 above / below float32(fma(k, step, a0))
    float sym_a;                         // float32(a[M - 1]) of an exactly symmetric {-a, 0, a} / {-a, a}, else 0
    int M, zero_idx, ok;
    double amax;                         // max |member|
    double pad[6];
    double a[64];                        // the members (float64), a[k] = rad * unit[k]
};
static_assert(sizeof(DevAlphabet) == 128 + 64 * 8, "DevAlphabet layout (include/gpfq.h documents the offsets of rad and a[])");

// an alphabet by reference (members in device memory): the kernels templated on their alphabet argument index it like AlphabetT
struct AlphabetRef {
    const double *a;
    int M;
};

// The alphabet as an arithmetic progression, if it is one in the sense the chain of decisions needs: strictly ascending and
// float32(fma(k, step, a0)) == float32(alphabet[k]) for every k (up to the one-ulp corrections of D.plus / D.minus), with
// step = (a[M-1] - a[0]) / (M - 1) -- the same fused operation the kernel performs.  Everything the reference builds
// (rad * linspace(-1, 1, M), scripts/quantized_network.py:396, :545) is, for a finite rad > 0.  Host and device (round 6: the device forms
// the alphabet itself from the layer's median, gpfq_alphabet_device_kernel): fills D's progression fields from D.a[0..M), true if it is one.
__host__ __device__ inline bool blk_fin(double x) { return fabs(x) <= 1.7976931348623157e308; }     // finite (false for NaN)
__host__ __device__ inline bool blk_uniform(DevAlphabet &D)
{
    const int M = D.M;
    const double *a = D.a;
    D.plus = D.minus = 0ull;
    D.a0 = D.step = D.inv = D.c0 = 0.0;
    if (M < 1 || M > 64 || !blk_fin(a[0]) || !blk_fin(a[M - 1])) return false;
    D.a0 = a[0];
    D.amax = fmax(fabs(a[0]), fabs(a[M - 1]));
    if (M == 1) return true;
    D.step = (a[M - 1] - a[0]) / (double)(M - 1);
    if (!(D.step > 0.0) || !blk_fin(D.step)) return false;
    D.inv = 1.0 / D.step;
    D.c0 = -D.a0 * D.inv;
    if (!blk_fin(D.inv) || !blk_fin(D.c0)) return false;
    for (int k = 0; k < M; ++k) {
        if (k > 0 && !(a[k - 1] < a[k])) return false;
        if (a[k] == 0.0) {                                     // (the member 0 is returned as such, BlkK::zero_idx, if the progression passes through it)
            if (!(fabs(fma((double)k, D.step, D.a0)) <= 0x1p-40 * D.amax)) return false;
        } else {
            const float want = (float)a[k], have = (float)fma((double)k, D.step, D.a0);
            const int64_t d = (int64_t)__builtin_bit_cast(int32_t, want) - (int64_t)__builtin_bit_cast(int32_t, have);
            if (d == 1) D.plus |= 1ull << k;
            else if (d == -1) D.minus |= 1ull << k;
            else if (d != 0) return false;
        }
        // and the index arithmetic finds a member from its own value (monotone rounding does the rest)
        if (rint(fma(a[k], D.inv, D.c0)) != (double)k) return false;
    }
    return true;
}

// a32 of an exactly symmetric alphabet {-a, 0, a} or {-a, a} (DevAlphabet::sym_a; the SYM instantiations), else 0
__host__ __device__ inline float blk_sym_of(const double *a, int M)
{
    if (M != 2 && M != 3) return 0.f;
    // (exactly symmetric as float64 too: the decisions' nearest-member search takes its boundaries as -a/2 and a/2)
    if (a[0] != -a[M - 1] || (M == 3 && a[1] != 0.0)) return 0.f;
    const float hi = (float)a[M - 1];
    if (!(hi > 0.f) || !blk_fin((double)hi)) return 0.f;
    return hi;
}


// The layer alphabet formed ON the device: rad = float64(alphabet_scalar) * float64(float32 median) -- the reference's legacy-NumPy
// product (:544: python scalar times np.float32 is a float64 product) --, members rad * unit[k] (:545: float64 products), and the
// progression the chain of decisions indexes.  `want_sym`: the caller will launch the symmetric-form instantiations (the unit alphabet is
// {-1, 0, 1} / {-1, 1}): an alphabet that is then not exactly symmetric is not ok.  One thread (gpfq_alphabet_device_kernel; the last
// workgroup of the one-GPU median, gpfq_median2_kernel).
__device__ inline void form_device_alphabet(DevAlphabet *out, float median32, double alphabet_scalar, const AlphabetArg &unit, int want_sym)
{
    DevAlphabet D{};
    D.M = unit.M; D.zero_idx = unit.zero_idx;
    D.rad = alphabet_scalar * (double)median32;
    for (int k = 0; k < unit.M && k < 64; ++k) D.a[k] = D.rad * unit.a[k];
    bool ok = blk_fin(D.rad) && blk_uniform(D);
    D.sym_a = blk_sym_of(D.a, D.M);
    if (want_sym && D.sym_a == 0.f) ok = false;
    // (the literal zero of rule (i) and the member 0: the caller's zero_idx is the unit alphabet's -- rad * 0 = 0 for every finite rad)
    if (ok && D.zero_idx >= 0 && D.a[D.zero_idx] != 0.0) ok = false;
    D.ok = ok ? 1 : 0;
    *out = D;
}

__device__ __forceinline__ double alphabet_lane(const AlphabetArg &A, int lane)
{
    return lane < A.M ? A.a[lane] : __longlong_as_double(0x7ff8000000000000LL);
}

// ---- alphabets of up to 64 * AR members across the lanes ----------------------------------------------
// Member k lives in register k / 64 of lane k % 64 (NaN beyond M); AR = 1 is the layout above.  nearest() /
// nearest_margin() / decide() below are the same procedures with the position count, the two neighbour
// distances and the first-minimum search taken over AR registers.
template <int AR>
struct AlphaLanes {
    double v[AR];
};
template <int AR, int CAP>
__device__ __forceinline__ AlphaLanes<AR> alpha_lanes(const AlphabetT<CAP> &A, int lane)
{
    static_assert(64 * AR <= CAP, "alphabet registers beyond the argument's capacity");
    AlphaLanes<AR> al;
#pragma unroll
    for (int r = 0; r < AR; ++r) al.v[r] = 64 * r + lane < A.M ? A.a[64 * r + lane] : __longlong_as_double(0x7ff8000000000000LL);
    return al;
}
// value of member k (wave-uniform k) of a lane-distributed array
template <int AR>
__device__ __forceinline__ double alpha_get(const double (&v)[AR], int k)
{
    double x = readlane_f64(v[0], k & 63);
#pragma unroll
    for (int r = 1; r < AR; ++r)
        if ((k >> 6) == r) x = readlane_f64(v[r], k & 63);
    return x;
}
template <int AR>
__device__ __forceinline__ int first_equal(const double (&d)[AR], double dmin)
{
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        const unsigned long long eq = __ballot(d[r] == dmin);
        if (eq) return 64 * r + (int)__ffsll((long long)eq) - 1;
    }
    return 0;                                                   // all-NaN distances: np.argmin -> 0
}
template <int AR>
__device__ __forceinline__ int nearest(double t, const AlphaLanes<AR> &al, int M, bool ascending)
{
    double d[AR];
#pragma unroll
    for (int r = 0; r < AR; ++r) d[r] = fabs(al.v[r] - t);
    if (ascending) {
        int p = 0;
#pragma unroll
        for (int r = 0; r < AR; ++r) p += __popcll(__ballot(al.v[r] < t));
        const int lo = p > 0 ? p - 1 : 0;
        const int hi = p < M ? p : M - 1;
        const double dlo = alpha_get<AR>(d, lo);
        const double dhi = alpha_get<AR>(d, hi);
        return first_equal<AR>(d, dlo <= dhi ? dlo : dhi);
    }
    int best = 0;
    double dbest = alpha_get<AR>(d, 0);
    for (int k = 1; k < M; ++k) {
        const double dk = alpha_get<AR>(d, k);
        if (dk < dbest) { dbest = dk; best = k; }
    }
    return best;
}
template <int AR>
__device__ __forceinline__ int nearest_margin(double t, const AlphaLanes<AR> &al, int M, bool ascending, double &margin)
{
    margin = -1.0;
    if (!ascending) return nearest<AR>(t, al, M, false);
    double d[AR];
    int p = 0;
#pragma unroll
    for (int r = 0; r < AR; ++r) {
        d[r] = fabs(al.v[r] - t);
        p += __popcll(__ballot(al.v[r] < t));
    }
    const int lo = p > 0 ? p - 1 : 0;
    const int hi = p < M ? p : M - 1;
    const double dlo = alpha_get<AR>(d, lo);
    const double dhi = alpha_get<AR>(d, hi);
    const double dmin = dlo <= dhi ? dlo : dhi;
    const int idx = first_equal<AR>(d, dmin);
    if (lo != hi) {
        if (idx == lo || idx == hi) margin = 0.5 * fabs(dlo - dhi);
    } else if (M == 1) {
        margin = __longlong_as_double(0x7ff0000000000000LL);
    } else if (idx == lo) {
        const double dn = alpha_get<AR>(d, lo == 0 ? 1 : M - 2);   // t lies outside the alphabet's range
        margin = 0.5 * (dn - dmin);
    }
    return idx;
}
template <int AR>
__device__ __forceinline__ Decision decide(float w, float nrm, double dot_u, double dot_uw,
                                           const AlphaLanes<AR> &al, int M, int zero_idx, bool ascending)
{
    Decision r;
    if ((double)nrm < 1e-16) {                                  // :83-84
        r.idx = zero_idx;
        r.q = 0.0;
        return r;
    }
    double t;
    if (fabs(dot_u) < 1e-10) t = (double)w;                     // :86-87
    else t = dot_uw / ((double)nrm * (double)nrm);              // :89 (IEEE f64 division)
    r.idx = nearest<AR>(t, al, M, ascending);
    r.q = alpha_get<AR>(al.v, r.idx);
    return r;
}

// Index element of the outputs: int8 for alphabets of up to 64 members, int16 beyond (include/gpfq.h: gpfq_index_bits).
template <int AR> struct IndexOf { using type = int8_t; };
template <> struct IndexOf<4> { using type = int16_t; };

}  // namespace gpfq
