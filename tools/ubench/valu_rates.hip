// Micro-benchmark: per-SIMD issue cost of the VALU instructions the GPFQ inner loop uses.
// Build: hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int U = 16;   // independent chains per lane

template <int OP>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float seed)
{
    float f[U]; double d[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { f[i] = seed + i + threadIdx.x; d[i] = (double)f[i] * 1.000001; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % U]));
            if (OP == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(d[(i + 1) % U]), "v"(d[(i + 2) % U]));
            if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 4) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i]));
            if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(seed));
            if (OP == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % U]));
            if (OP == 7) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f[i]) : "v"(f[(i + 1) % U]));
            if (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % U]));
            if (OP == 9) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) % U]));
        }
    }
    float s = 0; 
#pragma unroll
    for (int i = 0; i < U; ++i) s += f[i] + (float)d[i];
    if (s == 123.456f) out[0] = s;
}

template <int OP> float run(int waves_per_simd, int iters, float *out)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    dim3 grid(256), block(waves_per_simd * 4 * 64);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, 10, 1.0f);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, out, iters, 1.0f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 4));
    const char *names[] = {"v_mul_f32", "v_add_f64", "v_fma_f64", "v_cvt_f64_f32", "v_cvt_f32_f64", "v_fma_f32", "v_mul_f64", "v_mov_b32_dpp", "v_pk_mul_f32", "v_pk_add_f32"};
    const int iters = 20000;
    for (int w : {1, 2, 4}) {
        float ms[10];
        ms[0] = run<0>(w, iters, out); ms[1] = run<1>(w, iters, out); ms[2] = run<2>(w, iters, out);
        ms[3] = run<3>(w, iters, out); ms[4] = run<4>(w, iters, out); ms[5] = run<5>(w, iters, out);
        ms[6] = run<6>(w, iters, out); ms[7] = run<7>(w, iters, out); ms[8] = run<8>(w, iters, out); ms[9] = run<9>(w, iters, out);
        for (int o = 0; o < 10; ++o) {
            double ns_per_inst = ms[o] * 1e6 / ((double)iters * U * w);   // per wave-instruction per SIMD
            printf("waves/SIMD=%d %-16s %8.3f ms  %6.3f ns/inst/SIMD  (~%.2f cyc @2.4GHz)\n", w, names[o], ms[o], ns_per_inst, ns_per_inst * 2.4);
        }
    }
    return 0;
}
