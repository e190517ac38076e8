// Conv2D hot path for any kernel shape, stride and rate without patch matrices (implicit im2col).
//
// Row t = (ky, kx) of a channel's patch matrix (scripts/quantized_network.py:729-809, :123-183) at column
// (b, oy, ox) is plane[b][oy*sh + ky*rh - pt][ox*sw + kx*rw - pl] (zero outside the image).  The register-tile
// Gram kernel of gpfq_gram.hip stages its 256-column chunks of patch rows in LDS; here the chunks are
// gathered straight from the channel planes, for all channels of a shard in one launch (grid: column
// walkers x lower-triangle tiles x channels), so a 7x7/2 layer no longer writes and re-reads 20 GB of patch
// data per channel.  The 3x3 / stride-1 case has its own kernel (gpfq_gram_image.hip).
#include "gpfq_device.hpp"
#include "gpfq_gram_tile.hpp"
#include "gpfq_launch.hpp"

namespace gpfq {

struct ConvParams {
    const float *act_w, *act_q;
    int64_t plane;               // n*H*W floats per channel (< 2^31)
    int H, W, kw, sh, sw, rh, rw, pt, pl, oh, ow;
    int K, m;                    // kh*kw patch rows, n*oh*ow columns (< 2^30)
    int nchunks;
    int same_act;
    int padded;                  // some tap of some column falls outside the image
    int d64[3], dch[3];          // (images, rows, columns) decomposition of 64 columns / of the chunk stride
    int nparts;
    double *part;                // [nch][nparts] Gram records
    int *negflag;                // [nch], set when a channel has a negative activation
};

// (b, oy, ox) += d with carries; every component of d is below its modulus.
__device__ __forceinline__ void col_advance(int &b, int &oy, int &ox, const int (&d)[3], int oh, int ow)
{
    ox += d[2];
    int c = ox >= ow ? 1 : 0;
    ox -= c * ow;
    oy += d[1] + c;
    c = oy >= oh ? 1 : 0;
    oy -= c * oh;
    b += d[0] + c;
}

template <int TB, int SB>
__global__ void __launch_bounds__(kGramThreads, 2)
gpfq_gram_conv_kernel(ConvParams p)
{
    using Tile = GramTile<TB, SB>;
    constexpr int R = Tile::R, J = Tile::J;
    __shared__ __attribute__((aligned(16))) float lrow[R][kGramCH];
    int ty, sz;
    // grid = (tiles, column walkers, channels): the tiles of one walker are neighbours in launch order, so the
    // plane region they all gather from is in L2 at the same time
    tile_decode<TB, SB>(blockIdx.x, p.K, ty, sz);
    const int walker = blockIdx.y, nwalk = gridDim.y;
    const int t0 = ty * 4 * TB, s0 = sz * SB;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // uniform: row taps live in SGPRs
    const bool norms = (t0 + 4 * TB >= p.K) && wave == 0;
    const float *pw = p.act_w + (int64_t)blockIdx.z * p.plane;
    const float *pq = p.same_act ? pw : p.act_q + (int64_t)blockIdx.z * p.plane;
    Tile tile;
    tile.zero();

    // wavefront w stages rows w, w + 4, ... of the tile: their taps, once
    const float *rsrc[J];
    int rdy[J], rdx[J];
    bool rok[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int r = wave + 4 * j;
        int row = -1;
        const float *base = pq;
        if (r < 4 * TB) row = t0 + r;
        else if (r < 4 * TB + SB) { row = s0 + r - 4 * TB; base = pw; }
        else if (r < R) row = s0 + r - 4 * TB - SB;
        rok[j] = row >= 0 && row < p.K;
        const int ky = rok[j] ? row / p.kw : 0, kx = rok[j] ? row - ky * p.kw : 0;
        rsrc[j] = base;
        rdy[j] = ky * p.rh;
        rdx[j] = kx * p.rw;
    }

    // lane l gathers columns c0 + l + 64e (e = 0..3) of every chunk; the accumulate step reads them back as
    // columns 4l..4l+3 of the LDS rows
    int col = walker * kGramCH + lane;
    int b = col / (p.oh * p.ow);
    int oy = (col - b * p.oh * p.ow) / p.ow;
    int ox = col - (b * p.oh + oy) * p.ow;
    // All gathers of a chunk are issued back to back (clamped addresses, no branches around the loads) and
    // masked afterwards, one chunk ahead of its use: their latencies overlap each other and the FMAs.
    static_assert(J <= 8, "one mask bit per gathered value");
    float v[J][4];
    unsigned vmask = 0, signs = 0;
    auto gather = [&]() {
        int base[4], iy0[4], ix0[4];
        bool cok[4];
        int bb = b, yy = oy, xx = ox;
        vmask = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cok[e] = col + 64 * e < p.m;
            iy0[e] = yy * p.sh - p.pt;
            ix0[e] = xx * p.sw - p.pl;
            base[e] = (bb * p.H + iy0[e]) * p.W + ix0[e];
            if (e < 3) col_advance(bb, yy, xx, p.d64, p.oh, p.ow);
        }
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int roff = rdy[j] * p.W + rdx[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool ok = rok[j] && cok[e];
                if (p.padded)
                    ok = ok && (unsigned)(iy0[e] + rdy[j]) < (unsigned)p.H && (unsigned)(ix0[e] + rdx[j]) < (unsigned)p.W;
                v[j][e] = rsrc[j][ok ? base[e] + roff : 0];       // masked when it is written to LDS: no wait here
                vmask |= (ok ? 1u : 0u) << (4 * j + e);
            }
        }
    };
    if (walker < p.nchunks) gather();
    for (int ch = walker; ch < p.nchunks; ch += nwalk) {
        __syncthreads();                                   // the previous chunk has been consumed
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const int r = wave + 4 * j;
            if (r < R) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float val = (vmask >> (4 * j + e)) & 1u ? v[j][e] : 0.f;
                    lrow[r][lane + 64 * e] = val;
                    neg_track(signs, val);
                }
            }
        }
        __syncthreads();
        col += p.dch[0] * p.oh * p.ow + p.dch[1] * p.ow + p.dch[2];
        col_advance(b, oy, ox, p.dch, p.oh, p.ow);
        if (ch + nwalk < p.nchunks) gather();              // in flight during the FMAs below
        tile.accumulate(lrow, wave, lane, norms);
    }
    if (__ballot(neg_seen(signs)) && lane == 0) atomicOr(p.negflag + blockIdx.z, 1);   // a negative activation was seen
    tile.store(p.part + ((int64_t)blockIdx.z * p.nparts + walker) * gram_record(p.K), p.K, t0, s0, wave, lane, norms);
}

static inline size_t al256v(size_t x) { return (x + 255) & ~(size_t)255; }

constexpr int kConvTB = 2, kConvSB = 8;      // 8 x 8 tiles: least wasted work on the diagonal, no register spills
constexpr int kConvBlocksTarget = 4096;      // workgroups per launch to aim for (several rounds of the chip)
constexpr int kConvMaxWalkers = 512;

static int64_t conv_walkers(int64_t K, int64_t nch, int64_t m)
{
    const int64_t nchunks = (m + kGramCH - 1) / kGramCH;
    const int64_t tiles = tile_count<kConvTB, kConvSB>((int)K);
    int64_t x = (kConvBlocksTarget + tiles * nch - 1) / (tiles * nch);
    if (x > kConvMaxWalkers) x = kConvMaxWalkers;
    if (x > nchunks) x = nchunks;
    return x < 1 ? 1 : x;
}

bool gram_conv_supported(int64_t n, int64_t H, int64_t W, int64_t nch, int kh, int kw, int64_t oh, int64_t ow)
{
    const int64_t K = (int64_t)kh * kw;
    if (K < 1 || K > 256 || n <= 0 || oh <= 0 || ow <= 0 || nch <= 0 || nch > 65535) return false;
    return n * H * W < (1LL << 31) - (1LL << 20) && n * oh * ow < (1LL << 30);
}

size_t gram_conv_workspace_bytes(int64_t K, int64_t nch, int64_t F, int64_t m)
{
    size_t b = 0;
    b += al256v((size_t)nch * conv_walkers(K, nch, m) * gram_record(K) * sizeof(double));   // partial records
    b += al256v((size_t)nch * gram_record(K) * sizeof(double));                             // Gram records
    b += al256v((size_t)nch * K * sizeof(float));                                           // row norms
    b += al256v((size_t)nch * F * K * sizeof(float));                                       // chosen values per filter and step
    b += gram_fix_bytes();
    b += al256v((size_t)nch * sizeof(int));                                                 // negative-activation flags
    return b;
}

static void decompose(int64_t v, int64_t oh, int64_t ow, int (&d)[3])
{
    d[0] = (int)(v / (oh * ow));
    v -= (int64_t)d[0] * oh * ow;
    d[1] = (int)(v / ow);
    d[2] = (int)(v - (int64_t)d[1] * ow);
}

hipError_t launch_gram_conv(const ConvGramArgs &a, hipStream_t stream)
{
    if (a.nch == 0 || a.F == 0) return hipSuccess;
    const int64_t K = (int64_t)a.kh * a.kw, m = a.n * a.oh * a.ow;
    if (!gram_conv_supported(a.n, a.H, a.W, a.nch, a.kh, a.kw, a.oh, a.ow)) return hipErrorInvalidValue;
    const int64_t nparts = conv_walkers(K, a.nch, m);
    ConvParams p{};
    p.act_w = a.act_w; p.act_q = a.act_q; p.plane = a.n * a.H * a.W;
    p.H = (int)a.H; p.W = (int)a.W; p.kw = a.kw; p.sh = a.sh; p.sw = a.sw; p.rh = a.rh; p.rw = a.rw;
    p.pt = a.pt; p.pl = a.pl; p.oh = (int)a.oh; p.ow = (int)a.ow;
    p.K = (int)K; p.m = (int)m;
    p.nchunks = (int)((m + kGramCH - 1) / kGramCH);
    p.same_act = a.act_w == a.act_q;
    // taps reach from -pt to (oh-1)*sh + (kh-1)*rh - pt (rows), likewise for columns
    p.padded = a.pt > 0 || a.pl > 0 || (a.oh - 1) * a.sh + (int64_t)(a.kh - 1) * a.rh - a.pt >= a.H
               || (a.ow - 1) * a.sw + (int64_t)(a.kw - 1) * a.rw - a.pl >= a.W;
    decompose(64, a.oh, a.ow, p.d64);
    decompose((int64_t)kGramCH * nparts, a.oh, a.ow, p.dch);
    p.nparts = (int)nparts;
    char *ws = static_cast<char *>(a.workspace);
    double *part = reinterpret_cast<double *>(ws);  ws += al256v((size_t)a.nch * nparts * gram_record(K) * sizeof(double));
    double *gram = reinterpret_cast<double *>(ws);  ws += al256v((size_t)a.nch * gram_record(K) * sizeof(double));
    float *nrm = reinterpret_cast<float *>(ws);     ws += al256v((size_t)a.nch * K * sizeof(float));
    float *q32h = reinterpret_cast<float *>(ws);    ws += al256v((size_t)a.nch * a.F * K * sizeof(float));
    void *fixws = ws;                               ws += gram_fix_bytes();
    int *negflag = reinterpret_cast<int *>(ws);
    hipError_t e0 = hipMemsetAsync(negflag, 0, (size_t)a.nch * sizeof(int), stream);
    if (e0 != hipSuccess) return e0;
    p.part = part;
    p.negflag = negflag;
    hipLaunchKernelGGL((gpfq_gram_conv_kernel<kConvTB, kConvSB>),
                       dim3((unsigned)tile_count<kConvTB, kConvSB>((int)K), (unsigned)nparts, (unsigned)a.nch), dim3(kGramThreads), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    e = launch_gram_reduce(part, nparts, (int)K, gram, nrm, a.nch, stream);
    if (e != hipSuccess) return e;
    DecideBatch bs;
    bs.nch = a.nch; bs.gram_cs = gram_record(K); bs.nrm_cs = K; bs.w_cs = a.F * K; bs.out_cs = a.F * K; bs.unc_cs = a.F;
    bs.hist_cs = a.F * K;
    FixSrc src{};
    src.X = a.act_w; src.Xq = a.act_q; src.ld = 0; src.m = m; src.planes = 1; src.plane = p.plane;
    src.n = (int)a.n; src.H = p.H; src.W = p.W; src.oh = p.oh; src.ow = p.ow;
    src.kw = a.kw; src.sh = a.sh; src.sw = a.sw; src.rh = a.rh; src.rw = a.rw; src.pt = a.pt; src.pl = a.pl;
    return launch_gram_decide(gram, nrm, a.Wt, K, a.A, (int)K, a.F, a.slack, a.qidx, a.Qt, a.uncertified, q32h, bs, &src, fixws, negflag,
                              stream);
}

}  // namespace gpfq
