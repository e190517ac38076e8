// Conv2D hot path for 3x3 / stride 1 / rate 1 kernels without patch matrices.
//
// The reference builds, per input channel, the two patch matrices (scripts/quantized_network.py:729-809,
// :123-183) and runs every filter's 9-step recurrence on their rows (:185-233).  The decisions only need
// inner products of patch rows (gpfq_gram.hip), and row t = (ky, kx) of a patch matrix is the channel plane
// shifted by (ky - pad, kx - pad) with zeros outside the image, so those inner products are correlations of
// the plane with itself at pairs of shifts.  This file accumulates the Gram records of ALL channels of a
// shard straight from the channel planes in one launch:
//
//     G1[t][s] = <Xq_t, X_s>,  G2[t][s] = <Xq_t, Xq_s>  (s <= t),   nx2[s] = <X_s, X_s>
//
// i.e. 99 float64 FMAs per output position and channel, ~9x less HBM traffic than extracting patches.
// The kernel is bound by the FP64 VALU rate.
//
// Work split: a workgroup of 4 wavefronts stages a band of output rows (their input rows with the zero
// ring, float32) in LDS.  Wavefronts work in pairs on the same strips: a lane handles a strip of S
// consecutive output positions of one row from a 3 x (S+2) register window, and the two wavefronts of a
// pair own the columns s in {0,3,5,7,8} and {1,2,4,6} of the record (49 + 50 sums), which keeps a wavefront
// under 128 accumulator registers and two of them resident per SIMD.
#include "gpfq_device.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

constexpr int kImgThreads = 256;
constexpr int kImgStripsPerBand = 256;      // 2 sweeps of the 2 x 64 strip lanes per staged band
constexpr int kImgBlocks = 1024;            // upper bound of co-resident workgroups (sizes the partials)
constexpr int kImgMaxLds = 64 * 1024;
constexpr int64_t kRec9 = 9 * 9 * 2 + 9;

struct ImgParams {
    const float *act_w, *act_q;
    int64_t plane;        // n*H*W floats per channel
    int n, H, W, pad;
    int PH, oh, SPR;      // padded plane height, output rows per image, strips per output row
    int grows;            // n*oh output rows in total
    int RB, nbands;       // output rows per band
    int LP, lrows, lpr_log2;
    float inv_SPR, inv_oh, inv_PH;
    int same_act;
    int nbx;
    double *part;         // [nch][nbx*2] Gram records (N = 9)
    int *negflag;         // [nch], set when a channel has a negative activation
};

// floor(v / d) for 0 <= v < 2^24 (exact in float): float estimate, one correction step.
__device__ __forceinline__ int div_small(int v, int d, float inv, int &rem)
{
    int q = (int)((float)v * inv);
    const int r = v - q * d;
    const int fix = (r >= d ? 1 : 0) - (r < 0 ? 1 : 0);      // branch-free: the estimate is off by at most one
    rem = r - fix * d;
    return q + fix;
}

__host__ __device__ constexpr bool in_group(int g, int s)
{
    return g == 0 ? (s == 0 || s == 3 || s == 5 || s == 7 || s == 8) : (s == 1 || s == 2 || s == 4 || s == 6);
}

template <int S>
__device__ __forceinline__ void load_window_row(const float *p, float (&v)[S + 2])
{
    if constexpr (S == 4) {                        // p is 16-byte aligned (LP % 4 == 0, strips start at 4*sx)
        const float4 a = *reinterpret_cast<const float4 *>(p);
        const float2 b = *reinterpret_cast<const float2 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y;
    } else if constexpr (S == 2) {                 // 8-byte aligned
        const float2 a = *reinterpret_cast<const float2 *>(p), b = *reinterpret_cast<const float2 *>(p + 2);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    } else {
#pragma unroll
        for (int i = 0; i < S + 2; ++i) v[i] = p[i];
    }
}

template <int S, int G>
__device__ __forceinline__ void image_gram_body(const ImgParams &p, float *lds)
{
    Gram9 acc;                                       // only the columns of group G are ever touched
    gram9_zero(acc);
    const int lane = threadIdx.x & 63, pair = threadIdx.x >> 7;
    const float *pw = p.act_w + (int64_t)blockIdx.y * p.plane;
    const float *pq = p.act_q + (int64_t)blockIdx.y * p.plane;
    float *lw = lds;
    float *lq = p.same_act ? lds : lds + (size_t)p.lrows * p.LP;
    const int L4 = p.LP >> 2;
    // staging role of this thread: column piece st_c4 of rows st_row0 + k*st_step
    const int st_c4 = threadIdx.x & ((1 << p.lpr_log2) - 1), st_row0 = threadIdx.x >> p.lpr_log2;
    const int st_step = kImgThreads >> p.lpr_log2;
    const int st_ix = 4 * st_c4 - p.pad;
    const bool st_inside = st_ix >= 0 && st_ix + 3 < p.W;
    unsigned signs = 0;                                  // neg_track() of everything this thread staged

    for (int band = blockIdx.x; band < p.nbands; band += gridDim.x) {
        const int g0 = band * p.RB;
        const int rows = min(p.RB, p.grows - g0);
        int oy0, oy1;
        const int b0 = div_small(g0, p.oh, p.inv_oh, oy0);
        const int b1 = div_small(g0 + rows - 1, p.oh, p.inv_oh, oy1);
        const int v0 = b0 * p.PH + oy0;
        const int nv = b1 * p.PH + oy1 + 2 - v0 + 1;
        __syncthreads();                                   // the previous band has been consumed
        if (st_c4 < L4) {
            // this thread's column piece is fixed; walk its rows st_row0, st_row0 + st_step, ... of the band
            int r;
            int b = div_small(v0 + st_row0, p.PH, p.inv_PH, r);
            for (int row = st_row0; row < nv; row += st_step) {
                const int iy = r - p.pad;
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
                if (b < p.n && iy >= 0 && iy < p.H) {
                    const unsigned o = ((unsigned)b * p.H + iy) * p.W + st_ix;      // plane < 2^30 floats
                    if (st_inside) {
                        typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
                        const f4u t = *reinterpret_cast<const f4u *>(pw + o);
                        a = make_float4(t.x, t.y, t.z, t.w);
                        if (!p.same_act) {
                            const f4u u = *reinterpret_cast<const f4u *>(pq + o);
                            c = make_float4(u.x, u.y, u.z, u.w);
                        }
                    } else {
                        if (st_ix >= 0 && st_ix < p.W) { a.x = pw[o]; c.x = pq[o]; }
                        if (st_ix + 1 >= 0 && st_ix + 1 < p.W) { a.y = pw[o + 1]; c.y = pq[o + 1]; }
                        if (st_ix + 2 >= 0 && st_ix + 2 < p.W) { a.z = pw[o + 2]; c.z = pq[o + 2]; }
                        if (st_ix + 3 >= 0 && st_ix + 3 < p.W) { a.w = pw[o + 3]; c.w = pq[o + 3]; }
                    }
                }
                *reinterpret_cast<float4 *>(lw + row * p.LP + 4 * st_c4) = a;
                if (!p.same_act) *reinterpret_cast<float4 *>(lq + row * p.LP + 4 * st_c4) = c;
                neg_track(signs, a);
                neg_track(signs, c);
                r += st_step;
                while (r >= p.PH) { r -= p.PH; ++b; }
            }
        }
        __syncthreads();
        const int nstrips = rows * p.SPR;
        for (int q = pair * 64 + lane; q < nstrips; q += 128) {
            int sx, oy;
            const int grow = div_small(q, p.SPR, p.inv_SPR, sx);
            const int b = div_small(g0 + grow, p.oh, p.inv_oh, oy);
            const int off = (b * p.PH + oy - v0) * p.LP + S * sx;
            // 3 x (S+2) windows: the quantized plane in float64, the analog one converted where it is used
            double wq[3][S + 2];
            float wx[3][S + 2];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float r[S + 2];
                load_window_row<S>(lq + off + ky * p.LP, r);
#pragma unroll
                for (int i = 0; i < S + 2; ++i) wq[ky][i] = (double)r[i];
                load_window_row<S>(lw + off + ky * p.LP, wx[ky]);
            }
#pragma unroll
            for (int e = 0; e < S; ++e) {
#pragma unroll
                for (int s = 0; s < 9; ++s) {
                    if (in_group(G, s)) {
                        const double x = (double)wx[s / 3][s % 3 + e], y = wq[s / 3][s % 3 + e];
                        acc.nx[s] = fma(x, x, acc.nx[s]);
#pragma unroll
                        for (int t = s; t < 9; ++t) {
                            const double a = wq[t / 3][t % 3 + e];
                            acc.g[t * (t + 1) / 2 + s][0] = fma(a, x, acc.g[t * (t + 1) / 2 + s][0]);
                            acc.g[t * (t + 1) / 2 + s][1] = fma(a, y, acc.g[t * (t + 1) / 2 + s][1]);
                        }
                    }
                }
            }
        }
    }
    if (__ballot(neg_seen(signs)) && lane == 0) atomicOr(p.negflag + blockIdx.y, 1);   // a negative activation was seen
    // the two wavefronts of a pair fill disjoint columns of one record
    double *out = p.part + (((int64_t)blockIdx.y * p.nbx + blockIdx.x) * 2 + pair) * kRec9;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        if (in_group(G, s)) {
#pragma unroll
            for (int t = s; t < 9; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const double v = wave_sum(acc.g[t * (t + 1) / 2 + s][k]);
                    if (lane == 0) out[(t * 9 + s) * 2 + k] = v;
                }
            const double v = wave_sum(acc.nx[s]);
            if (lane == 0) out[162 + s] = v;
        }
    }
}

template <int S>
__global__ void __launch_bounds__(kImgThreads, 2)
gpfq_gram_image_kernel(ImgParams p)
{
    extern __shared__ __attribute__((aligned(16))) float img_lds[];
    // the wavefront index is uniform: both instantiations run the same staging code and barriers
    const int group = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6) & 1);
    if (group == 0) image_gram_body<S, 0>(p, img_lds);
    else image_gram_body<S, 1>(p, img_lds);
}

static inline size_t al256i(size_t x) { return (x + 255) & ~(size_t)255; }

static int image_strip(int64_t ow, int variant)
{
    if ((variant == 1 || variant == 2 || variant == 4) && ow % variant == 0) return variant;
    return ow % 4 == 0 ? 4 : ow % 2 == 0 ? 2 : 1;
}

static bool image_plan(int64_t n, int64_t H, int64_t W, int pad, int variant, ImgParams *out, int *S_out, size_t *lds_out)
{
    const int64_t oh = H + 2 * pad - 2, ow = W + 2 * pad - 2;
    if (oh <= 0 || ow <= 0 || n <= 0) return false;
    // div_small() is exact for dividends below 2^24 (float holds them exactly)
    if (n * (H + 2 * pad) >= (1LL << 24) || n * oh * ow >= (1LL << 30) || n * H * W >= (1LL << 30)) return false;
    if (W + 2 * pad > 1020) return false;                  // a staged row is at most one piece per thread
    const int S = image_strip(ow, variant);
    ImgParams p{};
    p.n = (int)n; p.H = (int)H; p.W = (int)W; p.pad = pad;
    p.plane = n * H * W;
    p.PH = (int)(H + 2 * pad); p.oh = (int)oh; p.SPR = (int)(ow / S);
    p.grows = (int)(n * oh);
    p.RB = kImgStripsPerBand / p.SPR; if (p.RB < 1) p.RB = 1;
    if (p.RB > p.grows) p.RB = p.grows;
    p.nbands = (p.grows + p.RB - 1) / p.RB;
    p.LP = (int)((W + 2 * pad + 3) & ~(int64_t)3);
    p.lrows = p.RB + 2 * ((p.RB - 1) / p.oh + 2);
    int lg = 0; while ((1 << lg) < p.LP / 4) ++lg;
    p.lpr_log2 = lg;
    p.inv_SPR = 1.0f / (float)p.SPR; p.inv_oh = 1.0f / (float)p.oh; p.inv_PH = 1.0f / (float)p.PH;
    const size_t lds = (size_t)2 * p.lrows * p.LP * sizeof(float);
    if (lds > (size_t)kImgMaxLds) return false;
    if (out) *out = p;
    if (S_out) *S_out = S;
    if (lds_out) *lds_out = lds;
    return true;
}

bool gram_image_supported(int64_t n, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int rh, int rw, int same_padding)
{
    if (kh != 3 || kw != 3 || sh != 1 || sw != 1 || rh != 1 || rw != 1) return false;
    return image_plan(n, H, W, same_padding ? 1 : 0, 0, nullptr, nullptr, nullptr);
}

size_t gram_image_workspace_bytes(int64_t nch, int64_t F)
{
    size_t b = 0;
    b += al256i((size_t)(nch + kImgBlocks) * 2 * kRec9 * sizeof(double));   // partials: nch * ceil(kImgBlocks / nch) * 2 at most
    b += al256i((size_t)nch * kRec9 * sizeof(double));                      // Gram records
    b += al256i((size_t)nch * 9 * sizeof(float));                           // row norms
    b += al256i((size_t)nch * F * 9 * sizeof(float));                       // chosen values per filter and step
    b += gram_fix_bytes();                                                  // device-side repair of uncertified chains
    b += al256i((size_t)nch * sizeof(int));                                 // "channel has negative activations" flags
    return b;
}

hipError_t launch_gram_image(const ImageGramArgs &a, hipStream_t stream)
{
    if (a.nch == 0 || (a.F == 0 && a.phase != 1)) return hipSuccess;
    ImgParams p;
    int S;
    size_t lds;
    if (!image_plan(a.n, a.H, a.W, a.pad, a.variant, &p, &S, &lds)) return hipErrorInvalidValue;
    p.act_w = a.act_w; p.act_q = a.act_q; p.same_act = a.act_w == a.act_q;
    // one round of the chip: as many workgroups as are co-resident
    const void *fn = S == 4 ? (const void *)gpfq_gram_image_kernel<4> : S == 2 ? (const void *)gpfq_gram_image_kernel<2>
                                                                       : (const void *)gpfq_gram_image_kernel<1>;
    int per_cu = 0, dev = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, kImgThreads, lds) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
        cus = 256;
    int64_t resident = (int64_t)per_cu * cus;
    if (resident > kImgBlocks) resident = kImgBlocks;
    int64_t nbx = (resident + a.nch - 1) / a.nch;
    if (nbx > p.nbands) nbx = p.nbands;
    p.nbx = (int)nbx;
    char *ws = static_cast<char *>(a.workspace);
    double *part = reinterpret_cast<double *>(ws);  ws += al256i((size_t)(a.nch + kImgBlocks) * 2 * kRec9 * sizeof(double));
    double *gram = reinterpret_cast<double *>(ws);  ws += al256i((size_t)a.nch * kRec9 * sizeof(double));
    float *nrm = reinterpret_cast<float *>(ws);     ws += al256i((size_t)a.nch * 9 * sizeof(float));
    float *q32h = reinterpret_cast<float *>(ws);    ws += al256i((size_t)a.nch * a.F * 9 * sizeof(float));
    void *fixws = ws;                               ws += gram_fix_bytes();
    int *negflag = reinterpret_cast<int *>(ws);
    hipError_t e;
    if (a.phase == 2) {                                    // records formed elsewhere (and summed over the column shards)
        e = hipMemcpyAsync(negflag, a.negflags, (size_t)a.nch * sizeof(int), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        e = launch_gram_reduce(a.records, 1, 9, gram, nrm, a.nch, stream);
        if (e != hipSuccess) return e;
    } else {
        hipError_t e0 = hipMemsetAsync(negflag, 0, (size_t)a.nch * sizeof(int), stream);
        if (e0 != hipSuccess) return e0;
        p.part = part;
        p.negflag = negflag;
        const dim3 grid((unsigned)nbx, (unsigned)a.nch), block(kImgThreads);
        switch (S) {
        case 4:  hipLaunchKernelGGL(gpfq_gram_image_kernel<4>, grid, block, lds, stream, p); break;
        case 2:  hipLaunchKernelGGL(gpfq_gram_image_kernel<2>, grid, block, lds, stream, p); break;
        default: hipLaunchKernelGGL(gpfq_gram_image_kernel<1>, grid, block, lds, stream, p); break;
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = launch_gram_reduce(part, nbx * 2, 9, gram, nrm, a.nch, stream);
        if (e != hipSuccess) return e;
        if (a.phase == 1) {
            e = hipMemcpyAsync(a.records, gram, (size_t)a.nch * kRec9 * sizeof(double), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return e;
            return hipMemcpyAsync(a.negflags, negflag, (size_t)a.nch * sizeof(int), hipMemcpyDeviceToDevice, stream);
        }
    }
    DecideBatch bs;
    bs.nch = a.nch; bs.gram_cs = kRec9; bs.nrm_cs = 9; bs.w_cs = a.F * 9; bs.out_cs = a.F * 9; bs.unc_cs = a.F; bs.hist_cs = a.F * 9;
    FixSrc src{};
    src.X = a.act_w; src.Xq = a.act_q; src.ld = 0; src.planes = 1; src.plane = p.plane;
    src.n = p.n; src.H = p.H; src.W = p.W; src.oh = p.oh; src.ow = p.SPR * S;
    src.kw = 3; src.sh = src.sw = src.rh = src.rw = 1; src.pt = src.pl = p.pad;
    src.m = (int64_t)p.grows * src.ow;
    return launch_gram_decide(gram, nrm, a.Wt, 9, a.A, 9, a.F, a.slack, a.qidx, a.Qt, a.uncertified, q32h, bs, &src, fixws, negflag,
                              stream, a.big);
}

}  // namespace gpfq
