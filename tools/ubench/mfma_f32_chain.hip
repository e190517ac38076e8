// Micro-benchmark (round 5): could the sweep's float32 work -- p = f32(w x), d = f32(p - sg f32(a xq)) for symmetric alphabets --
// move to v_mfma_f32_16x16x4_f32?  Two questions:
//  (a) SEMANTICS.  With A[sample][k] = (x, axq, 0, 0) and B[k][neuron] = (w, -sg, 0, 0), C = 0, is D[sample][neuron] bit for bit
//      fmaf(axq, -sg, fmaf(x, w, 0)) -- the reference's two roundings in the reference's order -- for every element?  The lane maps
//      assumed: A lane l = A[l % 16][l / 16], B lane l = B[l / 16][l % 16], D lane l reg r = D[4 (l / 16) + r][l % 16] (checked by
//      the comparison itself: a wrong map mismatches everywhere).  Also tried: the opposite order fmaf(x, w, f32(axq * -sg)).
//  (b) COST BESIDE THE FP64 VECTOR WORK.  Per 16-sample tile and step the sweep would issue 1 f32 MFMA + 4 v_cvt_f64_f32 + 4 v_add_f64
//      (+ a quarter of a v_mfma_f64_4x4x4 per step); cycles per trip of {NM f32 MFMAs + NV float64-rate vector instructions
//      (+ ND f64 MFMAs)} at 1..4 wavefronts per SIMD.  If the f32 matrix instruction runs beside the vector unit the trip costs
//      max(matrix, vector); if it occupies the same unit (as the f64 one does: profiles/r04) the costs add.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma32 tools/ubench/mfma_f32_chain.hip && /tmp/mfma32
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>

typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void k_sem(const float *x, const float *axq, const float *w, const float *sg, float *out)
{
    // one wavefront per tile: tile t has 16 samples and 16 neurons
    const int l = threadIdx.x, t = blockIdx.x;
    const int i = l & 15, k = l >> 4;
    const float a = k == 0 ? x[t * 16 + i] : (k == 1 ? axq[t * 16 + i] : 0.f);
    const float b = k == 0 ? w[t * 16 + i] : (k == 1 ? sg[t * 16 + i] : 0.f);
    f4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(t * 16 + 4 * k + r) * 16 + i] = c[r];       // [tile][sample 4k + r][neuron i]
}

template <int NM, int NV, int ND>
__global__ void __launch_bounds__(1024) k_rate(unsigned long long *cyc, double *out, int iters, float seed)
{
    f4 acc[4];
    double dacc[4], v[8], e[8];
    float a[4], b[4], cv[8];
    double da[4], db[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { acc[i] = f4{0.f, 0.f, 0.f, 0.f}; dacc[i] = 0.0; a[i] = seed + threadIdx.x + i; b[i] = seed * 0.5f - i; da[i] = a[i]; db[i] = b[i]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed * i; e[i] = seed + i; cv[i] = seed * i + 0.25f; }
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        constexpr int TOT = NM + NV + ND;
#pragma unroll
        for (int i = 0, im = 0, iv = 0, id = 0; i < TOT; ++i) {
            const bool mf = NM > 0 && (i * NM) / TOT != ((i + 1) * NM) / TOT;
            const bool md = !mf && ND > 0 && (i * ND) / TOT != ((i + 1) * ND) / TOT;
            if (mf) {
                acc[im % 4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[im % 4], b[(im + 1) % 4], acc[im % 4], 0, 0, 0);
                asm volatile("" : "+v"(acc[im % 4]));
                ++im;
            } else if (md) {
                dacc[id % 4] = __builtin_amdgcn_mfma_f64_4x4x4f64(da[id % 4], db[(id + 1) % 4], dacc[id % 4], 0, 0, 0);
                asm volatile("" : "+v"(dacc[id % 4]));
                ++id;
            } else {
                // the sweep's vector mix: conversions and float64 additions, alternating
                if (iv & 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[iv % 8]) : "v"(e[iv % 8]));
                else asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(e[iv % 8]) : "v"(cv[iv % 8]));
                ++iv;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    double s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + dacc[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + e[i];
    if (s == 123.456) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NM, int NV, int ND> void rate(unsigned long long *cyc, double *out)
{
    const int iters = 2000;
    printf("  %2d f32 MFMA + %2d vector (cvt/add f64) + %2d f64 MFMA per trip:", NM, NV, ND);
    for (int wps = 1; wps <= 4; ++wps) {
        dim3 grid(256), block(wps * 4 * 64);
        const int nw = 256 * wps * 4;
        for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k_rate<NM, NV, ND>), grid, block, 0, 0, cyc, out, iters, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per_trip = (double)h[nw / 2] / iters;
        printf("  %dw %7.1f (%6.1f/SIMD)", wps, per_trip, per_trip / wps);
    }
    printf("\n");
}

int main()
{
    // ---- (a) semantics ----
    const int T = 4096;
    std::mt19937 g(7);
    std::vector<float> x(T * 16), axq(T * 16), w(T * 16), sg(T * 16), out(T * 256);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    for (int i = 0; i < T * 16; ++i) {
        const int e1 = (int)(g() % 24) - 12, e2 = (int)(g() % 24) - 12;
        x[i] = (g() % 5 == 0) ? 0.f : std::ldexp(U(g) + 0.5f, e1);          // ReLU-like: zeros and positive values over 24 binades
        axq[i] = (g() % 5 == 0) ? 0.f : std::ldexp(U(g) + 0.5f, e1 + (int)(g() % 3) - 1);
        w[i] = std::ldexp(U(g) - 0.5f, e2 / 3);
        sg[i] = (float)((int)(g() % 3) - 1);
    }
    float *dx, *dq, *dw, *ds, *dout;
    hipMalloc(&dx, T * 64); hipMalloc(&dq, T * 64); hipMalloc(&dw, T * 64); hipMalloc(&ds, T * 64); hipMalloc(&dout, T * 1024);
    hipMemcpy(dx, x.data(), T * 64, hipMemcpyHostToDevice); hipMemcpy(dq, axq.data(), T * 64, hipMemcpyHostToDevice);
    hipMemcpy(dw, w.data(), T * 64, hipMemcpyHostToDevice); hipMemcpy(ds, sg.data(), T * 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_sem, dim3(T), dim3(64), 0, 0, dx, dq, dw, ds, dout);
    hipMemcpy(out.data(), dout, T * 1024, hipMemcpyDeviceToHost);
    long bad_ref = 0, bad_rev = 0, bad_fused = 0, nz = 0;
    for (int t = 0; t < T; ++t)
        for (int s = 0; s < 16; ++s)
            for (int n = 0; n < 16; ++n) {
                const float X = x[t * 16 + s], Q = axq[t * 16 + s], W = w[t * 16 + n], S = sg[t * 16 + n];
                const float got = out[(t * 16 + s) * 16 + n];
                const float p = X * W;                                         // f32 product (rounded)
                const float ref = std::fmaf(Q, S, p);                          // the reference's flow for symmetric alphabets
                const float rev = std::fmaf(X, W, Q * S);
                const float fused = (float)((double)X * (double)W + (double)Q * (double)S);   // one rounding of the exact sum
                unsigned ug, ur, uv, uf;
                std::memcpy(&ug, &got, 4); std::memcpy(&ur, &ref, 4); std::memcpy(&uv, &rev, 4); std::memcpy(&uf, &fused, 4);
                // (+0 and -0 are one value here: the increment is added to a float64 residual)
                bad_ref += !(got == ref || (ug == ur));
                bad_rev += !(got == rev || (ug == uv));
                bad_fused += !(got == fused || (ug == uf));
                nz += got != 0.f;
            }
    printf("(a) v_mfma_f32_16x16x4_f32, A = (x, axq, 0, 0), B = (w, sg, 0, 0), %ld elements (%ld nonzero):\n", (long)T * 256, nz);
    printf("    mismatches vs fmaf(axq, sg, f32(x w)) [the reference's order]: %ld\n", bad_ref);
    printf("    mismatches vs fmaf(x, w, f32(axq sg))  [opposite order]:        %ld\n", bad_rev);
    printf("    mismatches vs one rounding of the exact sum:                    %ld\n", bad_fused);

    // ---- (b) cost ----
    unsigned long long *cyc; hipMalloc(&cyc, 256 * 16 * 8);
    double *o; hipMalloc(&o, 8);
    printf("(b) cycles per trip, median wavefront (and per SIMD = / wavefronts per SIMD):\n");
    rate<8, 0, 0>(cyc, o);
    rate<0, 64, 0>(cyc, o);
    rate<0, 0, 8>(cyc, o);
    rate<8, 64, 0>(cyc, o);        // the proposed tile-step mix x 8: 1 f32 MFMA per 8 vector instructions
    rate<8, 64, 2>(cyc, o);        // + the dot products' f64 MFMAs (a quarter per tile-step)
    rate<0, 64, 2>(cyc, o);
    rate<4, 64, 0>(cyc, o);
    rate<16, 64, 0>(cyc, o);
    return 0;
}
