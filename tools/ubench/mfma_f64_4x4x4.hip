// Micro-benchmark (round 4): v_mfma_f64_4x4x4_4b_f64 -- (a) the lane maps of its A, B and D operands, found with one-hot
// operands (nothing is assumed), and (b) what it costs a SIMD that is otherwise busy with the GPFQ sweep's vector
// instructions: cycles per loop trip of {NM MFMAs + NV vector instructions} at 1..4 wavefronts per SIMD, by s_memtime.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma444 mfma_f64_4x4x4.hip && /tmp/mfma444
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

// ---- (a) layout: for every pair (la, lb) of one-hot lanes, the lanes of D that receive the product ----
__global__ void k_layout(unsigned long long *hit)
{
    const int l = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            const unsigned long long m = __ballot(d != 0.0);
            if (l == 0) hit[la * 64 + lb] = m;
        }
}

// ---- (b) rate: NM MFMAs (NA accumulators round-robin) + NV vector f64 adds per trip ----
template <int NM, int NV, int NA>
__global__ void __launch_bounds__(1024) k_rate(unsigned long long *cyc, double *out, int iters, double seed)
{
    double acc[NA], a[4], b[4], v[8], e[8];
#pragma unroll
    for (int i = 0; i < NA; ++i) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = seed + threadIdx.x + i; b[i] = seed * 0.5 - i; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed * i; e[i] = seed + i; }
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        constexpr int TOT = NM + NV;
#pragma unroll
        for (int i = 0, im = 0, iv = 0; i < TOT; ++i) {
            // spread the MFMAs evenly among the vector instructions
            const bool mf = NM > 0 && (NV == 0 || (i * NM) / TOT != ((i + 1) * NM) / TOT);
            if (mf) {
                // (the builtin, not inline asm: hipcc then pads the MFMA hazards itself)
                acc[im % NA] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[im % 4], b[(im + 1) % 4], acc[im % NA], 0, 0, 0);
                asm volatile("" : "+v"(acc[im % NA]));
                ++im;
            } else {
                asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[iv % 8]) : "v"(e[iv % 8]));
                ++iv;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0;
#pragma unroll
    for (int i = 0; i < NA; ++i) s += acc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 123.456) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NM, int NV, int NA> void rate(unsigned long long *cyc, double *out)
{
    const int iters = 2000;
    printf("  %2d MFMA (%d acc) + %2d v_add_f64 per trip:", NM, NA, NV);
    for (int wps = 1; wps <= 4; ++wps) {
        dim3 grid(256), block(wps * 4 * 64);
        const int nw = 256 * wps * 4;
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k_rate<NM, NV, NA>), grid, block, 0, 0, cyc, out, iters, 1.0);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per_trip = (double)h[nw / 2] / iters;
        // SIMD cycles per trip of ONE wavefront's work = wavefront cycles per trip / wavefronts per SIMD
        printf("  %dw %7.1f (%6.1f/SIMD)", wps, per_trip, per_trip / wps);
    }
    printf("\n");
}

int main()
{
    unsigned long long *hit; hipMalloc(&hit, 64 * 64 * 8);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, hit);
    std::vector<unsigned long long> h(64 * 64);
    hipMemcpy(h.data(), hit, 64 * 64 * 8, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_4x4x4_4b_f64 one-hot map: A lane la x B lane lb -> D lane (only pairs that meet)\n");
    for (int la = 0; la < 64; ++la) {
        printf("  A%2d:", la);
        for (int lb = 0; lb < 64; ++lb) {
            const unsigned long long m = h[la * 64 + lb];
            if (!m) continue;
            printf(" B%2d->D", lb);
            for (int l = 0; l < 64; ++l) if (m >> l & 1) printf("%d,", l);
        }
        printf("\n");
    }
    // hypothesis check: A[i][k] of block b at lane i + 4k + 16b, B[k][j] at lane j + 4k + 16b, D[i][j] at lane j + 4i + 16b
    int bad1 = 0, bad2 = 0;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            {   // H1
                const int ia = la & 3, ka = (la >> 2) & 3, ba = la >> 4, jb = lb & 3, kb = (lb >> 2) & 3, bb = lb >> 4;
                const unsigned long long want = (ka == kb && ba == bb) ? 1ull << (jb + 4 * ia + 16 * ba) : 0ull;
                bad1 += want != h[la * 64 + lb];
            }
            {   // H2: i + 4b + 16k (the 16x16x4 map with the rows taken as (block, i))
                const int ia = la & 3, ba = (la >> 2) & 3, ka = la >> 4, jb = lb & 3, bb = (lb >> 2) & 3, kb = lb >> 4;
                const unsigned long long want = (ka == kb && ba == bb) ? 1ull << (jb + 4 * ba + 16 * ia) : 0ull;
                bad2 += want != h[la * 64 + lb];
            }
        }
    printf("H1 (A: i + 4k + 16b, B: j + 4k + 16b, D: j + 4i + 16b): %d mismatches\n", bad1);
    printf("H2 (A: i + 4b + 16k, B: j + 4b + 16k, D: j + 4b + 16i): %d mismatches\n", bad2);

    unsigned long long *cyc; hipMalloc(&cyc, 256 * 16 * 8);
    double *out; hipMalloc(&out, 8);
    printf("cycles per trip, median wavefront (and per SIMD = / wavefronts per SIMD):\n");
    rate<8, 0, 8>(cyc, out);
    rate<8, 0, 4>(cyc, out);
    rate<8, 0, 1>(cyc, out);
    rate<0, 24, 1>(cyc, out);
    rate<2, 24, 2>(cyc, out);
    rate<4, 24, 4>(cyc, out);
    rate<8, 24, 4>(cyc, out);
    rate<8, 24, 8>(cyc, out);
    rate<8, 48, 4>(cyc, out);
    rate<8, 96, 4>(cyc, out);
    rate<0, 96, 1>(cyc, out);
    return 0;
}
