// Register tile of a Gram record shared by the patch-matrix kernel (gpfq_gram.hip) and the implicit-im2col
// kernel (gpfq_gram_conv.hip): a workgroup of 4 wavefronts owns rows t in [t0, t0 + 4*TB) (wavefront w: TB of
// them) against rows s in [s0, s0 + SB); a chunk of 256 columns of the 4*TB + 2*SB rows it needs
// (Xq_t | X_s | Xq_s) sits in LDS, lane l feeds columns 4l..4l+3 into TB*SB*2 float64 accumulators.
#pragma once

#include "gpfq_device.hpp"

namespace gpfq {

constexpr int kGramThreads = 256;
constexpr int kGramCH = 256;               // columns staged per chunk

__host__ __device__ inline int64_t gram_record(int64_t N) { return N * N * 2 + N; }

template <int TB, int SB>
struct GramTile {
    static constexpr int R = 4 * TB + 2 * SB;          // staged rows
    static constexpr int J = (R + 3) / 4;              // rows staged per wavefront
    double acc[TB][SB][2], accn[SB];

    __device__ __forceinline__ void zero()
    {
#pragma unroll
        for (int a = 0; a < TB; ++a)
#pragma unroll
            for (int s = 0; s < SB; ++s) { acc[a][s][0] = 0.0; acc[a][s][1] = 0.0; }
#pragma unroll
        for (int s = 0; s < SB; ++s) accn[s] = 0.0;
    }

    // one staged chunk; `norms`: this wavefront also accumulates <X_s, X_s> for the SB columns
    __device__ __forceinline__ void accumulate(const float (*lrow)[kGramCH], int wave, int lane, bool norms)
    {
        float4 qt4[TB];
#pragma unroll
        for (int a = 0; a < TB; ++a) qt4[a] = *reinterpret_cast<const float4 *>(&lrow[wave * TB + a][4 * lane]);
        // the LDS reads run one column row ahead of the FMAs; the scheduling barrier keeps the compiler from
        // hoisting all 2*SB reads (and their conversions) to the top, which costs more registers than there are
        float4 xs_next = *reinterpret_cast<const float4 *>(&lrow[4 * TB][4 * lane]);
        float4 qs_next = *reinterpret_cast<const float4 *>(&lrow[4 * TB + SB][4 * lane]);
#pragma unroll
        for (int s = 0; s < SB; ++s) {
            const float4 xs4 = xs_next, qs4 = qs_next;
            if (s + 1 < SB) {
                xs_next = *reinterpret_cast<const float4 *>(&lrow[4 * TB + s + 1][4 * lane]);
                qs_next = *reinterpret_cast<const float4 *>(&lrow[4 * TB + SB + s + 1][4 * lane]);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float xsv[4] = {xs4.x, xs4.y, xs4.z, xs4.w}, qsv[4] = {qs4.x, qs4.y, qs4.z, qs4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double xs = (double)xsv[e], qs = (double)qsv[e];
#pragma unroll
                for (int a = 0; a < TB; ++a) {
                    const float qtf = e == 0 ? qt4[a].x : e == 1 ? qt4[a].y : e == 2 ? qt4[a].z : qt4[a].w;
                    const double qt = (double)qtf;
                    acc[a][s][0] = fma(qt, xs, acc[a][s][0]);
                    acc[a][s][1] = fma(qt, qs, acc[a][s][1]);
                }
            }
        }
        if (norms) {
#pragma unroll
            for (int s = 0; s < SB; ++s) {
                const float4 xs4 = *reinterpret_cast<const float4 *>(&lrow[4 * TB + s][4 * lane]);
                accn[s] = fma((double)xs4.x, (double)xs4.x, accn[s]);
                accn[s] = fma((double)xs4.y, (double)xs4.y, accn[s]);
                accn[s] = fma((double)xs4.z, (double)xs4.z, accn[s]);
                accn[s] = fma((double)xs4.w, (double)xs4.w, accn[s]);
            }
        }
    }

    // wave sums -> this workgroup's partial record
    __device__ __forceinline__ void store(double *__restrict__ out, int N, int t0, int s0, int wave, int lane, bool norms)
    {
#pragma unroll
        for (int a = 0; a < TB; ++a) {
            const int t = t0 + wave * TB + a;
#pragma unroll
            for (int s = 0; s < SB; ++s)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double v = wave_sum(acc[a][s][k]);
                    if (lane == 0 && t < N && s0 + s < N) out[((int64_t)t * N + (s0 + s)) * 2 + k] = v;
                }
        }
        if (norms) {
#pragma unroll
            for (int s = 0; s < SB; ++s) {
                const double v = wave_sum(accn[s]);
                if (lane == 0 && s0 + s < N) out[(int64_t)N * N * 2 + s0 + s] = v;
            }
        }
    }
};

// Tiles of the lower triangle in row-major order: linear index -> (ty, sz).  Row ty holds the column tiles
// sz with SB*sz <= 4*TB*ty + 4*TB - 1 (capped at the last column tile).
template <int TB, int SB>
__host__ __device__ inline int tile_row_count(int ty, int N)
{
    const int nsz = (N + SB - 1) / SB;
    const int c = (4 * TB * ty + 4 * TB - 1) / SB + 1;
    return c < nsz ? c : nsz;
}

template <int TB, int SB>
__host__ __device__ inline int tile_count(int N)
{
    const int nty = (N + 4 * TB - 1) / (4 * TB);
    int total = 0;
    for (int ty = 0; ty < nty; ++ty) total += tile_row_count<TB, SB>(ty, N);
    return total;
}

template <int TB, int SB>
__device__ __forceinline__ void tile_decode(int e, int N, int &ty, int &sz)
{
    ty = 0;
    for (;;) {
        const int c = tile_row_count<TB, SB>(ty, N);
        if (e < c) break;
        e -= c;
        ++ty;
    }
    sz = e;
}

}  // namespace gpfq
