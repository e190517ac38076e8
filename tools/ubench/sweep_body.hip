// Micro-benchmark: the block kernel's phase U / phase D loop bodies (csrc/gpfq_blk.hip) in isolation, identical wavefronts,
// 1..4 wavefronts per SIMD, operands from registers or from LDS (one pair prefetched ahead, as in the kernel).
// Prints shader cycles per vector-ALU instruction per SIMD.
// hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o /tmp/sweep_body sweep_body.hip && /tmp/sweep_body
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float pk2 __attribute__((ext_vector_type(2)));

template <int PW, int MODE>   // MODE 0: U from registers, 1: U from LDS, 2: D from registers, 3: D from LDS
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, double *out, int iters, float seed)
{
    __shared__ float2 sx[4096];
    __shared__ double2 sd[2048];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sx[i] = make_float2(seed + i, seed - i);
    for (int i = threadIdx.x; i < 2048; i += blockDim.x) sd[i] = make_double2(seed + i, seed - i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double u[4][2 * PW];
    for (int n = 0; n < 4; ++n) for (int e = 0; e < 2 * PW; ++e) u[n][e] = seed * (n + e);
    float wv[4], qv[4];
    for (int n = 0; n < 4; ++n) { wv[n] = seed + n; qv[n] = seed - n; }
    double acc[4] = {0, 0, 0, 0};
    const int base = (wave * 64 + (lane >> 2)) & 1023;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (MODE < 2) {
        float2 x2n = sx[base], q2n = sx[base + 1024];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                const float2 x2 = x2n, q2 = q2n;
                if (MODE == 1) { x2n = sx[(base + 16 * (p + 1) + it) & 1023]; q2n = sx[1024 + ((base + 16 * (p + 1) + it) & 1023)]; }
                else { x2n.x += 1.0f; q2n.y += 1.0f; }
                __builtin_amdgcn_sched_barrier(0);
                const pk2 xv = {x2.x, x2.y}, qx = {q2.x, q2.y};
                pk2 pr[4], rr[4], dd[4];
                double c0[4], c1[4];
#pragma unroll
                for (int n = 0; n < 4; ++n) pr[n] = pk2{wv[n], wv[n]} * xv;
#pragma unroll
                for (int n = 0; n < 4; ++n) rr[n] = pk2{qv[n], qv[n]} * qx;
#pragma unroll
                for (int n = 0; n < 4; ++n) dd[n] = pr[n] - rr[n];
#pragma unroll
                for (int n = 0; n < 4; ++n) { c0[n] = (double)dd[n].x; c1[n] = (double)dd[n].y; }
#pragma unroll
                for (int n = 0; n < 4; ++n) { u[n][2 * p] += c0[n]; u[n][2 * p + 1] += c1[n]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        double2 d2n = sd[base];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int p = 0; p < PW; ++p) {
                const double2 d2 = d2n;
                if (MODE == 3) d2n = sd[(base + 16 * (p + 1) + it) & 1023];
                else d2n.x += 1.0;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    acc[n] = fma(d2.x, u[n][2 * p], acc[n]);
                    acc[n] = fma(d2.y, u[n][2 * p + 1], acc[n]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = acc[0] + acc[1] + acc[2] + acc[3];
    for (int n = 0; n < 4; ++n) for (int e = 0; e < 2 * PW; ++e) s += u[n][e];
    if (s == 123.456) out[0] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int PW, int MODE> double run(int wps, int iters, unsigned long long *cyc, double *out)
{
    dim3 grid(256), block(wps * 4 * 64);
    const int nw = 256 * wps * 4;
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((k<PW, MODE>), grid, block, 0, 0, cyc, out, iters, 1.0f);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nw);
    hipMemcpy(h.data(), cyc, nw * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const int per = MODE < 2 ? 28 * PW : 8 * PW;      // vector-ALU instructions of the body (without the operand bookkeeping)
    return (double)h[nw / 2] / ((double)iters * per) / wps;
}

int main()
{
    unsigned long long *cyc; double *out;
    hipMalloc(&cyc, 8192 * 8); hipMalloc(&out, 8);
    for (int r = 0; r < 100; ++r) hipLaunchKernelGGL((k<3, 0>), dim3(256), dim3(512), 0, 0, cyc, out, 5000, 1.0f);
    hipDeviceSynchronize();
    const int iters = 5000;
    const char *names[] = {"phase U, registers", "phase U, LDS operands", "phase D, registers", "phase D, LDS operands"};
    for (int w : {1, 2, 3, 4}) {
        double c[4] = {run<3, 0>(w, iters, cyc, out), run<3, 1>(w, iters, cyc, out), run<3, 2>(w, iters, cyc, out), run<3, 3>(w, iters, cyc, out)};
        for (int o = 0; o < 4; ++o) printf("waves/SIMD=%d PW=3 %-24s %6.2f shader cycles per VALU instruction per SIMD\n", w, names[o], c[o]);
    }
    return 0;
}
