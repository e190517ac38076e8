// Micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 against v_fma_f64 (per SIMD), and a layout check of the
// f64 MFMA operand/result maps with exact integer data.
// Build: hipcc -O3 --offload-arch=gfx950 -o mfma_f64_rate mfma_f64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int U = 8;   // independent accumulator tiles per wavefront

__global__ void __launch_bounds__(256) k_mfma(double *out, int iters, double seed)
{
    double4_t acc[U];
    for (int i = 0; i < U; ++i) acc[i] = double4_t{0, 0, 0, 0};
    const double a = seed + threadIdx.x, b = seed * 0.5 + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < U; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[0] = s;
}

// MODE 0: both multiplicands are the same registers in every instruction; 1: one of them changes from one
// instruction to the next (8 different registers); 2: both change.
template <int MODE>
__global__ void __launch_bounds__(256) k_fma(double *out, int iters, double seed)
{
    double acc[4 * U];
    for (int i = 0; i < 4 * U; ++i) acc[i] = i;
    double av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = seed + threadIdx.x + i; bv[i] = seed * 0.5 + threadIdx.x - i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4 * U; ++i)
            asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(av[MODE == 2 ? i % 8 : 0]), "v"(bv[MODE >= 1 ? (i + 3) % 8 : 0]));
    }
    double s = 0;
    for (int i = 0; i < 4 * U; ++i) s += acc[i];
    if (s == 123.456) out[0] = s;
}

// D = A(16x4) * B(4x16): lane l supplies A[l&15][l>>4] and B[l>>4][l&15]; result register r of lane l is
// D[(l>>4) + 4r][l&15] (cdna_hip_programming.md) -- verified here with small integers.
__global__ void k_layout(int *bad)
{
    const int l = threadIdx.x;
    const double a = (l & 15) * 10 + (l >> 4);          // A[i][k] = 10 i + k
    const double b = (l >> 4) * 100 + (l & 15);         // B[k][j] = 100 k + j
    double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, double4_t{0, 0, 0, 0}, 0, 0, 0);
    int nbad = 0;
    for (int r = 0; r < 4; ++r) {
        const int i = (l >> 4) + 4 * r, j = l & 15;
        double want = 0;
        for (int k = 0; k < 4; ++k) want += (10.0 * i + k) * (100.0 * k + j);
        if (d[r] != want) ++nbad;
    }
    if (nbad) atomicAdd(bad, nbad);
}

int main()
{
    double *out; hipMalloc(&out, 8);
    int *bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, bad);
    int hb = -1; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("layout check: %d wrong results (0 = maps are as documented)\n", hb);
    const int iters = 20000;
    const char *names[4] = {"v_mfma_f64_16x16x4            ", "v_fma_f64, same multiplicands ", "v_fma_f64, one changing       ",
                            "v_fma_f64, both changing      "};
    for (int waves = 1; waves <= 4; ++waves) {
        for (int which = 0; which < 4; ++which) {
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            dim3 grid(256 * waves), block(256);
            auto launch = [&](int n) {
                if (which == 0) hipLaunchKernelGGL(k_mfma, grid, block, 0, 0, out, n, 1.0);
                else if (which == 1) hipLaunchKernelGGL(k_fma<0>, grid, block, 0, 0, out, n, 1.0);
                else if (which == 2) hipLaunchKernelGGL(k_fma<1>, grid, block, 0, 0, out, n, 1.0);
                else hipLaunchKernelGGL(k_fma<2>, grid, block, 0, 0, out, n, 1.0);
            };
            launch(10);
            hipEventRecord(a);
            launch(iters);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double fma_per_wave = which == 0 ? (double)iters * U * 1024 : (double)iters * 4 * U * 64;
            const double total = fma_per_wave * 256 * waves * 4;      // waves in flight: grid * 4 per block
            printf("%s %d wave(s) per SIMD: %.3f ms, %.1f TFLOP/s (%.1f FMA/clk/SIMD at 2.4 GHz)\n",
                   names[which], waves, ms, 2 * total / ms / 1e9, total / (ms * 1e-3) / 2.4e9 / 1024);
        }
    }
    return 0;
}
