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

// ---- DPP plumbing -------------------------------------------------------------------------
// dpp_ctrl encodings (CDNA ISA): row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    // lanes whose source is outside the row / masked rows receive `old` = 0
    int rlo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
    int rhi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(rhi, rlo);
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

// Sum of x over the 64 lanes of the wavefront, returned wave-uniform (read from lane 63).
// 4 in-row scan steps + 2 row broadcasts; deterministic order.
__device__ __forceinline__ double wave_sum(double x)
{
    x += dpp_fetch<0x111, 0xF>(x);   // row_shr:1
    x += dpp_fetch<0x112, 0xF>(x);   // row_shr:2
    x += dpp_fetch<0x114, 0xF>(x);   // row_shr:4
    x += dpp_fetch<0x118, 0xF>(x);   // row_shr:8   -> lane 15 of each row = row total
    x += dpp_fetch<0x142, 0xA>(x);   // row_bcast:15 into rows 1,3
    x += dpp_fetch<0x143, 0xC>(x);   // row_bcast:31 into rows 2,3 -> lane 63 = wave total
    return readlane_f64(x, 63);
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

struct AlphabetArg {
    double a[64];       // GPFQ_MAX_ALPHABET
    int    M;
    int    zero_idx;
    int    ascending;
};

__device__ __forceinline__ double alphabet_lane(const AlphabetArg &A, int lane)
{
    return lane < A.M ? A.a[lane] : __longlong_as_double(0x7ff8000000000000LL);
}

}  // namespace gpfq
