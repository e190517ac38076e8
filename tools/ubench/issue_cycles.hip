// Micro-benchmark: shader-clock cycles per wavefront instruction (s_memtime inside the kernel, so the
// result does not depend on the clock the chip holds), at 1 to 4 wavefronts per SIMD, for the
// instructions and the instruction MIX of the GPFQ sweep (csrc/gpfq_pipe.hip).
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -o /tmp/issue_cycles issue_cycles.hip && /tmp/issue_cycles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

constexpr int U = 8;   // independent chains per lane

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, float *out, int iters, float seed)
{
    float f[2 * U]; double d[U], e[U];
#pragma unroll
    for (int i = 0; i < U; ++i) { f[2 * i] = seed + i + threadIdx.x; f[2 * i + 1] = seed * i; d[i] = (double)f[2 * i] * 1.000001; e[i] = d[i] * 0.5; }
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (OP == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
            if (OP == 2) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(d[i]) : "v"(e[i]));
            if (OP == 3) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i]));
            if (OP == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
            if (OP == 5) asm volatile("v_mov_b32_dpp %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(f[i]) : "v"(f[(i + 1) % U]));
            if (OP == 6) {   // sweep mix for 2 samples: 3 packed f32, 2 cvt, 2 f64 add, 2 f64 fma  (independent across i)
                asm volatile("v_pk_mul_f32 %0, %1, %1\n\tv_pk_mul_f32 %2, %3, %3\n\tv_pk_add_f32 %0, %0, %2"
                             : "+v"(*(double *)&f[2 * i]), "+v"(e[i]), "+v"(e[(i + 1) % U]), "+v"(e[(i + 2) % U]));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e[i]) : "v"(f[2 * i]));
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e[(i + 3) % U]) : "v"(f[2 * i + 1]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[(i + 4) % U]) : "v"(e[(i + 3) % U]));
                asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(e[(i + 5) % U]) : "v"(d[i]));
                asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(e[(i + 6) % U]) : "v"(d[(i + 4) % U]));
            }
            if (OP == 7) {   // dependent pair cvt -> add, pairs independent
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e[i]) : "v"(f[i]));
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(e[i]));
            }
            if (OP == 8) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[0]) : "v"(e[i]));          // one dependent chain
            if (OP == 9) asm volatile("v_cmp_lt_f64 vcc, %0, %1\n\tv_cndmask_b32 %2, %2, %3, vcc" : : "v"(d[i]), "v"(e[i]), "v"(f[i]), "v"(f[i + 1]) : "vcc");
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = 0;
#pragma unroll
    for (int i = 0; i < U; ++i) s += f[2 * i] + f[2 * i + 1] + (float)d[i] + (float)e[i];
    if (s == 123.456f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP> double run(int waves_per_simd, int iters, unsigned long long *cyc, float *out, int inst_per_iter)
{
    dim3 grid(256), block(waves_per_simd * 4 * 64);
    const int nw = 256 * waves_per_simd * 4;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, cyc, out, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    // cycles per instruction PER SIMD: a wave's lifetime / its instructions / waves sharing the SIMD
    return (double)h[nw / 2] / ((double)iters * inst_per_iter) / waves_per_simd;
}

int main()
{
    unsigned long long *cyc; float *out;
    hipMalloc(&cyc, 8192 * 8); hipMalloc(&out, 4);
    // warm up the clocks
    for (int r = 0; r < 200; ++r) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, cyc, out, 20000, 1.0f);
    hipDeviceSynchronize();
    const int iters = 20000;
    const char *names[] = {"v_mul_f32", "v_add_f64", "v_fma_f64", "v_cvt_f64_f32", "v_pk_mul_f32", "v_mov_b32_dpp",
                           "sweep mix (9 inst / 2 samples)", "cvt->add dependent pairs", "v_add_f64 one chain", "v_cmp_f64+cndmask"};
    const int per[] = {U, U, U, U, U, U, 9 * U, 2 * U, U, 2 * U};
    for (int w : {1, 2, 3, 4}) {
        double c[10];
        c[0] = run<0>(w, iters, cyc, out, per[0]); c[1] = run<1>(w, iters, cyc, out, per[1]); c[2] = run<2>(w, iters, cyc, out, per[2]);
        c[3] = run<3>(w, iters, cyc, out, per[3]); c[4] = run<4>(w, iters, cyc, out, per[4]); c[5] = run<5>(w, iters, cyc, out, per[5]);
        c[6] = run<6>(w, iters, cyc, out, per[6]); c[7] = run<7>(w, iters, cyc, out, per[7]); c[8] = run<8>(w, iters, cyc, out, per[8]);
        c[9] = run<9>(w, iters, cyc, out, per[9]);
        for (int o = 0; o < 10; ++o) printf("waves/SIMD=%d %-34s %6.2f shader cycles per instruction per SIMD\n", w, names[o], c[o]);
    }
    return 0;
}
