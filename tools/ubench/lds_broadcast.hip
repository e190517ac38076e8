// Micro-benchmark: LDS read cost when several lanes of a wavefront read the SAME address (the sweep's operands: the four
// neuron-group lanes of a k-lane share x / xq), by access width and by which lanes share.  Four wavefronts per SIMD issue
// batches of independent reads; prints shader cycles per read instruction per CU (sixteen wavefronts share the LDS).
// hipcc -O3 --offload-arch=gfx950 -o /tmp/lds_broadcast lds_broadcast.hip && /tmp/lds_broadcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int W>   // W: 0 = b32, 1 = b64, 2 = b128, 3 = read2st64_b64
__global__ void __launch_bounds__(1024) k(unsigned long long *cyc, float *out, int iters, int pat)
{
    __shared__ __attribute__((aligned(16))) float s[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) s[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int unit = W == 0 ? 4 : (W == 2 ? 16 : 8);
    int idx = lane;                                   // pattern 0: all lanes distinct, contiguous
    if (pat == 1) idx = lane >> 2;                    // four consecutive lanes share
    if (pat == 2) idx = lane & 15;                    // lanes l, l+16, l+32, l+48 share
    if (pat == 3) idx = lane >> 1;                    // two consecutive lanes share
    if (pat == 4) idx = lane & 31;                    // lanes l, l+32 share
    if (pat == 5) idx = 0;                            // all lanes one address
    if (pat == 6) idx = (lane & 7) | ((lane >> 5) << 3);   // lanes l, l+8, l+16, l+24 share (inside each half)
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)s + idx * unit + ((threadIdx.x >> 6) & 3) * 2048;
    unsigned long long t0, t1;
    float acc = 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        if (W == 0) {
            float a, b, c, d;
            asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:1024\n\tds_read_b32 %2, %4 offset:512\n\tds_read_b32 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
            acc += a + b + c + d;
        } else if (W == 1) {
            double a, b, c, d;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %4 offset:1024\n\tds_read_b64 %2, %4 offset:512\n\tds_read_b64 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
            acc += (float)(a + b + c + d);
        } else if (W == 2) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 a, b, c, d;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:512\n\tds_read_b128 %3, %4 offset:1536\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
            acc += a.x + b.y + c.z + d.w;
        } else {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 a, b, c, d;
            asm volatile("ds_read2st64_b64 %0, %4 offset1:8\n\tds_read2st64_b64 %1, %4 offset0:1 offset1:9\n\tds_read2st64_b64 %2, %4 offset0:2 offset1:10\n\tds_read2st64_b64 %3, %4 offset0:3 offset1:11\n\ts_waitcnt lgkmcnt(0)"
                         : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(addr) : "memory");
            acc += a.x + b.y + c.z + d.w;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    if (acc == 123.456f) out[0] = acc;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int W> double run(int pat, unsigned long long *cyc, float *out)
{
    const int iters = 20000;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<W>, dim3(256), dim3(1024), 0, 0, cyc, out, iters, pat);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(4096);
    hipMemcpy(h.data(), cyc, 4096 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    return (double)h[2048] / (iters * 4.0) / 16.0;    // sixteen wavefronts per CU issue concurrently
}

int main()
{
    unsigned long long *cyc; float *out;
    hipMalloc(&cyc, 4096 * 8); hipMalloc(&out, 4);
    const char *pats[] = {"all lanes distinct", "4 consecutive lanes share", "lanes l, l+16, l+32, l+48 share", "2 consecutive lanes share",
                          "lanes l, l+32 share", "all lanes one address", "lanes l, l+8, l+16, l+24 share"};
    const char *ws[] = {"ds_read_b32", "ds_read_b64", "ds_read_b128", "ds_read2st64_b64"};
    for (int p = 0; p < 7; ++p) {
        double c[4] = {run<0>(p, cyc, out), run<1>(p, cyc, out), run<2>(p, cyc, out), run<3>(p, cyc, out)};
        for (int w = 0; w < 4; ++w) printf("%-34s %-18s %6.2f cycles per instruction per CU\n", pats[p], ws[w], c[w]);
    }
    return 0;
}
