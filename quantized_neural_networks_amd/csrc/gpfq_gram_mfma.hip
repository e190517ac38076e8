// Gram records of LONG walks (64 < N <= 1024: dense layers whose rows are too long for the on-chip residual,
// scripts/quantized_network.py:60-124 on the reference's MNIST run with 25000 samples) on the matrix cores.
//
// G1 = Xq X^T and G2 = Xq Xq^T are float64 GEMMs over float32 data (every product of two float32 values is
// exact in float64), so 64 x 64 tiles of the lower triangle are accumulated with v_mfma_f64_16x16x4_f64: the
// register-tile kernel (gpfq_gram.hip) reads 26 LDS values per 48 FMAs and ends up LDS-bound at 24 TFLOP/s
// for N = 784, the matrix cores take one LDS value per 16 FMAs.
//
// Workgroup (x, e): tile e of the lower triangle (rows t0.. against rows s0..), column chunks x, x + gridDim.x, ...
// A chunk of 64 columns of the 192 rows (Xq_t | X_s | Xq_s) is staged in LDS through registers that are
// requested one chunk ahead.  Wavefront w owns rows t0 + 16w .. + 15 against all 64 s: 2 x 4 accumulator blocks.
//
// Operand layout of v_mfma_f64_16x16x4_f64 (D = A B + C, A 16x4, B 4x16): lane l holds A[l & 15][l >> 4] and
// B[l >> 4][l & 15]; D register r of lane l is D[(l >> 4) + 4r][l & 15].  Both operands are "row (l & 15) of a
// row-major matrix at k-column (l >> 4)" here, so one float4 read per 16 x 16 block serves four MFMAs (the
// order in which the columns are summed is free).
//
// The partial records have the layout of gpfq_gram.hip (one per column walker), summed by gpfq_gram_reduce_kernel.
#include "gpfq_device.hpp"
#include "gpfq_gram_tile.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

constexpr int kMfmaT = 64;                 // tile edge
constexpr int kMfmaCH = 64;                // columns per staged chunk
constexpr int kMfmaLD = kMfmaCH + 4;       // LDS row stride in floats (16-byte aligned, rows 4 banks apart)
constexpr int kMfmaStage = 3 * kMfmaT * (kMfmaCH / 4) / kGramThreads;   // float4 per thread and chunk: 12

typedef double mfma_acc __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(kGramThreads, 2)
gpfq_gram_mfma_kernel(const float *__restrict__ X, const float *__restrict__ Xq, int64_t ld, int N, int64_t m,
                      int64_t nchunks, double *__restrict__ part, int *__restrict__ negflag)
{
    __shared__ __attribute__((aligned(16))) float lds[3 * kMfmaT][kMfmaLD];
    int ty = 0, sz = blockIdx.y;
    while (sz > ty) { sz -= ty + 1; ++ty; }                                // lower triangle, row-major
    const int t0 = ty * kMfmaT, s0 = sz * kMfmaT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool norms = t0 + kMfmaT >= N;                                   // the last tile row meets every column tile
    unsigned signs = 0;

    // staging role: float4 column group c4 of rows r0, r0 + 16, ... of the 192 staged rows
    const int c4 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    const float *src[kMfmaStage];
#pragma unroll
    for (int p = 0; p < kMfmaStage; ++p) {
        const int r = r0 + 16 * p, set = r / kMfmaT, within = r % kMfmaT;
        const int row = (set == 0 ? t0 : s0) + within;
        src[p] = row < N ? (set == 1 ? X : Xq) + (int64_t)row * ld + 4 * c4 : nullptr;
    }
    float4 pre[kMfmaStage];
    auto request = [&](int64_t ch) __attribute__((always_inline)) {
        const int64_t col = ch * kMfmaCH + 4 * c4;
#pragma unroll
        for (int p = 0; p < kMfmaStage; ++p) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src[p] && col < m) {
                v = *reinterpret_cast<const float4 *>(src[p] + ch * kMfmaCH);      // ld % 4 == 0: inside the row's storage
                if (col + 4 > m) {                                                 // ragged end of the row
                    if (col + 1 >= m) v.y = 0.f;
                    if (col + 2 >= m) v.z = 0.f;
                    v.w = 0.f;
                }
            }
            pre[p] = v;
        }
    };

    mfma_acc g1[4], g2[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) { g1[b] = mfma_acc{0.0, 0.0, 0.0, 0.0}; g2[b] = mfma_acc{0.0, 0.0, 0.0, 0.0}; }
    double nx = 0.0;                                                       // <X_s, X_s>: row 16w + (l & 15), quarter l >> 4
    const int fr = lane & 15, fk = lane >> 4;

    if ((int64_t)blockIdx.x < nchunks) request(blockIdx.x);
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < kMfmaStage; ++p) {
            *reinterpret_cast<float4 *>(&lds[r0 + 16 * p][4 * c4]) = pre[p];
            neg_track(signs, pre[p]);
        }
        __syncthreads();
        if (ch + gridDim.x < nchunks) request(ch + gridDim.x);             // in flight while the matrix cores work
#pragma unroll
        for (int jj = 0; jj < kMfmaCH / 16; ++jj) {
            const int col = 16 * jj + 4 * fk;
            const float4 a4 = *reinterpret_cast<const float4 *>(&lds[16 * wave + fr][col]);
            float4 bx[4], bq[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                bx[b] = *reinterpret_cast<const float4 *>(&lds[kMfmaT + 16 * b + fr][col]);
                bq[b] = *reinterpret_cast<const float4 *>(&lds[2 * kMfmaT + 16 * b + fr][col]);
            }
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double a = (double)av[e];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float xf = e == 0 ? bx[b].x : e == 1 ? bx[b].y : e == 2 ? bx[b].z : bx[b].w;
                    const float qf = e == 0 ? bq[b].x : e == 1 ? bq[b].y : e == 2 ? bq[b].z : bq[b].w;
                    g1[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (double)xf, g1[b], 0, 0, 0);
                    g2[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, (double)qf, g2[b], 0, 0, 0);
                }
            }
            if (norms) {
                const float4 x4 = *reinterpret_cast<const float4 *>(&lds[kMfmaT + 16 * wave + fr][col]);
                nx = fma((double)x4.x, (double)x4.x, nx);
                nx = fma((double)x4.y, (double)x4.y, nx);
                nx = fma((double)x4.z, (double)x4.z, nx);
                nx = fma((double)x4.w, (double)x4.w, nx);
            }
        }
    }
    if (__ballot(neg_seen(signs)) && lane == 0) atomicOr(negflag, 1);      // a negative element was seen

    double *out = part + (int64_t)blockIdx.x * gram_record(N);
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int s = s0 + 16 * b + fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = t0 + 16 * wave + fk + 4 * r;
            if (t < N && s < N) *reinterpret_cast<double2 *>(&out[((int64_t)t * N + s) * 2]) = make_double2(g1[b][r], g2[b][r]);
        }
    }
    if (norms) {
        nx += __shfl_xor(nx, 16);
        nx += __shfl_xor(nx, 32);
        const int s = s0 + 16 * wave + fr;
        if (fk == 0 && s < N) out[(int64_t)N * N * 2 + s] = nx;
    }
}

bool gram_mfma_supported(const float *X, const float *Xq, int64_t ld, int64_t N)
{
    return N > kMfmaT && ld % 4 == 0 && (uintptr_t)X % 16 == 0 && (uintptr_t)Xq % 16 == 0;
}

static int64_t mfma_tiles(int64_t N)
{
    const int64_t nt = (N + kMfmaT - 1) / kMfmaT;
    return nt * (nt + 1) / 2;
}

// column walkers (= partial records): about four workgroups per CU over all tiles
int64_t gram_mfma_walkers(int64_t N, int64_t m)
{
    const int64_t nchunks = (m + kMfmaCH - 1) / kMfmaCH;
    int64_t w = (1024 + mfma_tiles(N) - 1) / mfma_tiles(N);
    if (w > nchunks) w = nchunks;
    return w > 0 ? w : 1;
}

hipError_t launch_gram_mfma(const float *X, const float *Xq, int64_t ld, int64_t N, int64_t m, double *part, int *negflag,
                            hipStream_t stream)
{
    const int64_t nchunks = (m + kMfmaCH - 1) / kMfmaCH;
    hipLaunchKernelGGL(gpfq_gram_mfma_kernel, dim3((unsigned)gram_mfma_walkers(N, m), (unsigned)mfma_tiles(N)), dim3(kGramThreads),
                       0, stream, X, Xq, ld, (int)N, m, nchunks, part, negflag);
    return hipGetLastError();
}

}  // namespace gpfq
