// Conv2D hot path for any kernel shape, stride and rate without patch matrices (implicit im2col).
//
// Row t = (ky, kx) of a channel's patch matrix (scripts/quantized_network.py:729-809, :123-183) at column
// (b, oy, ox) is plane[b][oy*sh + ky*rh - pt][ox*sw + kx*rw - pl] (zero outside the image).  The register-tile
// Gram kernel of gpfq_gram.hip stages its 256-column chunks of patch rows in LDS; here the chunks are
// gathered straight from the channel planes, for all channels of a shard in one launch (grid: column
// walkers x lower-triangle tiles x channels), so a 7x7/2 layer no longer writes and re-reads 20 GB of patch
// data per channel.  The 3x3 / stride-1 case has its own kernel (gpfq_gram_image.hip).
#include <atomic>

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
    const float *zero;           // one 0.0f in device memory: what taps outside the image read (matrix-core kernel)
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

// ---- 6 <= K <= 64 (strided 3x3, 5x5, 7x7 kernels): the same records on the matrix cores ------------------------
// The register-tile kernel above gathers the rows of every 8 x 8 tile again (28 tiles x 24 rows for a 7x7 kernel:
// the gathers, not the FMAs, set its pace).  Here one workgroup keeps ALL K rows of a 128-column chunk in LDS
// (gathered once, one chunk ahead) and forms the 16 x 16 blocks of the lower triangle with
// v_mfma_f64_16x16x4_f64 (operand layout: gpfq_gram_mfma.hip).  Wavefronts 0,1 accumulate G1 = Xq X^T (+ the
// squared norms of X) over alternate 16-column groups, wavefronts 2,3 G2 = Xq Xq^T; when both networks see the
// same input (first layer) G2 = G1 and all four wavefronts split the groups of G1.  One partial record per
// (walker, column-group split).
// REM1: K = 16 NB + 1 (7x7 = 49): a block row of the matrix cores for ONE row would be 40 % of the MFMA work; the
// last row goes through the vector units instead (lane s owns the pair (K-1, s), each wavefront a quarter of
// the chunk's columns; the four partial sums meet in LDS at the end).
// Taps that fall outside the image (or the matrix) read a zero word instead of being masked afterwards.
constexpr int kCmCH = 128;                 // columns per staged chunk
constexpr int kCmLD = kCmCH + 4;           // LDS row stride in floats
constexpr int kCmSplit = 4;                // partial records per walker (upper bound: 4 when act_w == act_q, else 2)

typedef double cm_acc __attribute__((ext_vector_type(4)));

// PADDED: some tap of some column falls outside the image (SAME padding ...): only then are the taps bounds-checked.
template <int NB, bool REM1, bool PADDED>
__global__ void __launch_bounds__(kGramThreads, 2)     // two workgroups per CU: one's gathers and LDS traffic under the other's MFMAs
gpfq_gram_conv_mfma_kernel(ConvParams p)
{
    constexpr int ROWS = 16 * NB + (REM1 ? 4 : 0);  // staged rows per set (a multiple of 4: wavefront w stages rows w, w + 4, ...)
    constexpr int NBLK = NB * (NB + 1) / 2;
    constexpr int JH = ROWS / 4;
    __shared__ __attribute__((aligned(16))) float lrow[2][ROWS][kCmLD];     // set 0: X rows, set 1: Xq rows
    __shared__ int tap_off[ROWS], tap_yx[ROWS];
    const int walker = blockIdx.x, nwalk = gridDim.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float *pw = p.act_w + (int64_t)blockIdx.z * p.plane;
    const float *pq = p.act_q + (int64_t)blockIdx.z * p.plane;
    const int64_t zoff[2] = {p.zero - pw, p.zero - pq};      // the zero word as an element offset from either plane
    const bool same = p.same_act != 0;
    for (int r = threadIdx.x; r < ROWS; r += kGramThreads) {
        const int ky = r < p.K ? r / p.kw : 0, kx = r < p.K ? r - ky * p.kw : 0;
        tap_off[r] = ky * p.rh * p.W + kx * p.rw;
        tap_yx[r] = ((ky * p.rh) << 16) | (kx * p.rw);
    }
    __syncthreads();

    // lane l gathers columns c0 + l and c0 + l + 64 of every chunk, for rows wave, wave + 4, ... of both sets
    int col = walker * kCmCH + lane;
    int b = col / (p.oh * p.ow);
    int oy = (col - b * p.oh * p.ow) / p.ow;
    int ox = col - (b * p.oh + oy) * p.ow;
    float v[2][JH][2];
    unsigned signs = 0;
    auto gather = [&]() __attribute__((always_inline)) {
        int base[2], iy0[2], ix0[2];
        bool cok[2];
        int bb = b, yy = oy, xx = ox;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            cok[e] = col + 64 * e < p.m;
            iy0[e] = yy * p.sh - p.pt;
            ix0[e] = xx * p.sw - p.pl;
            base[e] = (bb * p.H + iy0[e]) * p.W + ix0[e];
            if (e == 0) col_advance(bb, yy, xx, p.d64, p.oh, p.ow);
        }
#pragma unroll
        for (int set = 0; set < 2; ++set) {
            if (set == 1 && same) break;                                   // one input: only the X rows are staged
            const float *src = set ? pq : pw;
#pragma unroll
            for (int j = 0; j < JH; ++j) {
                const int row = wave + 4 * j;
                const int off = tap_off[row], yx = tap_yx[row];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    // branch-free and without a predicate that outlives the load: a tap outside the image (or the matrix) reads the
                    // zero word at p.zero.  (Short-circuit conditions and a pointer select turned every element into a saveexec /
                    // branch ladder with the 52 predicates spilled to VGPR lanes: ~1500 instructions per chunk and wavefront.)
                    const bool in = !PADDED || (((unsigned)(iy0[e] + (yx >> 16)) < (unsigned)p.H) & ((unsigned)(ix0[e] + (yx & 0xffff)) < (unsigned)p.W));
                    const bool ok = (row < p.K) & cok[e] & in;
                    const int64_t idx = ok ? (int64_t)(base[e] + off) : zoff[set];
                    v[set][j][e] = src[idx];
                }
            }
        }
    };

    const int which = same ? 0 : wave >> 1;                                // 0: G1 (+ norms), 1: G2
    const int ksplit = same ? 4 : 2, kidx = same ? wave : wave & 1;
    const float (*asrc)[kCmLD] = lrow[same ? 0 : 1];                       // Xq rows
    const float (*bsrc)[kCmLD] = which == 0 ? lrow[0] : lrow[1];           // X rows (G1) or Xq rows (G2)
    const int fr = lane & 15, fk = lane >> 4;
    cm_acc acc[NBLK];
#pragma unroll
    for (int i = 0; i < NBLK; ++i) acc[i] = cm_acc{0.0, 0.0, 0.0, 0.0};
    double nx[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) nx[i] = 0.0;
    double r1 = 0.0, r2 = 0.0, rn = 0.0;                                   // REM1: <Xq_T, X_s>, <Xq_T, Xq_s>, <X_s, X_s>, T = 16 NB
    const int rs = lane <= 16 * NB ? lane : 16 * NB;                       // REM1: this lane's s (clamped; lanes > T idle)

    if (walker < p.nchunks) gather();
    for (int ch = walker; ch < p.nchunks; ch += nwalk) {
        __syncthreads();                                   // the previous chunk has been consumed
#pragma unroll
        for (int set = 0; set < 2; ++set) {
            if (set == 1 && same) break;
#pragma unroll
            for (int j = 0; j < JH; ++j)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    lrow[set][wave + 4 * j][lane + 64 * e] = v[set][j][e];
                    neg_track(signs, v[set][j][e]);
                }
        }
        __syncthreads();
        col += p.dch[0] * p.oh * p.ow + p.dch[1] * p.ow + p.dch[2];
        col_advance(b, oy, ox, p.dch, p.oh, p.ow);
        if (ch + nwalk < p.nchunks) gather();              // in flight while the matrix cores work
        for (int g = kidx; g < kCmCH / 16; g += ksplit) {
            const int c = 16 * g + 4 * fk;
            float4 a4[NB], b4[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                a4[i] = *reinterpret_cast<const float4 *>(&asrc[16 * i + fr][c]);
                b4[i] = *reinterpret_cast<const float4 *>(&bsrc[16 * i + fr][c]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double bd[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) bd[i] = (double)(e == 0 ? b4[i].x : e == 1 ? b4[i].y : e == 2 ? b4[i].z : b4[i].w);
#pragma unroll
                for (int bt = 0; bt < NB; ++bt) {
                    const double ad = (double)(e == 0 ? a4[bt].x : e == 1 ? a4[bt].y : e == 2 ? a4[bt].z : a4[bt].w);
#pragma unroll
                    for (int bs = 0; bs <= bt; ++bs)
                        acc[bt * (bt + 1) / 2 + bs] = __builtin_amdgcn_mfma_f64_16x16x4f64(ad, bd[bs], acc[bt * (bt + 1) / 2 + bs], 0, 0, 0);
                }
                if (which == 0) {
#pragma unroll
                    for (int i = 0; i < NB; ++i) nx[i] = fma(bd[i], bd[i], nx[i]);
                }
            }
        }
        if constexpr (REM1) {
            // the last row on the vector units: this wavefront's quarter of the columns, lane = s
            const float (*xs)[kCmLD] = lrow[0];
#pragma unroll
            for (int c = 32 * wave; c < 32 * wave + 32; c += 4) {
                const float4 qt = *reinterpret_cast<const float4 *>(&asrc[16 * NB][c]);   // broadcast
                const float4 x4 = *reinterpret_cast<const float4 *>(&xs[rs][c]);
                const float4 q4 = *reinterpret_cast<const float4 *>(&asrc[rs][c]);
                r1 = fma((double)qt.x, (double)x4.x, r1); r1 = fma((double)qt.y, (double)x4.y, r1);
                r1 = fma((double)qt.z, (double)x4.z, r1); r1 = fma((double)qt.w, (double)x4.w, r1);
                r2 = fma((double)qt.x, (double)q4.x, r2); r2 = fma((double)qt.y, (double)q4.y, r2);
                r2 = fma((double)qt.z, (double)q4.z, r2); r2 = fma((double)qt.w, (double)q4.w, r2);
                rn = fma((double)x4.x, (double)x4.x, rn); rn = fma((double)x4.y, (double)x4.y, rn);
                rn = fma((double)x4.z, (double)x4.z, rn); rn = fma((double)x4.w, (double)x4.w, rn);
            }
        }
    }
    if (__ballot(neg_seen(signs)) && lane == 0) atomicOr(p.negflag + blockIdx.z, 1);   // a negative activation was seen

    const int64_t rec = gram_record(p.K);
    double *out = p.part + ((int64_t)blockIdx.z * p.nparts + (int64_t)walker * ksplit + kidx) * rec;
#pragma unroll
    for (int bt = 0; bt < NB; ++bt)
#pragma unroll
        for (int bs = 0; bs <= bt; ++bs)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int t = 16 * bt + fk + 4 * r, s = 16 * bs + fr;
                if (t < p.K && s <= t) {
                    const double val = acc[bt * (bt + 1) / 2 + bs][r];
                    double *o = out + ((int64_t)t * p.K + s) * 2;
                    if (which == 0) o[0] = val;
                    if (which == 1 || same) o[1] = val;
                }
            }
    if (which == 0) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            double x = nx[i];
            x += __shfl_xor(x, 16);
            x += __shfl_xor(x, 32);
            if (fk == 0 && 16 * i + fr < p.K) out[(int64_t)p.K * p.K * 2 + 16 * i + fr] = x;
        }
    }
    if constexpr (REM1) {
        // row T = 16 NB: the four column quarters meet in LDS; the walker's first record takes the sums, its
        // other records zeros (every record is summed entry by entry afterwards)
        __syncthreads();
        double *red = reinterpret_cast<double *>(&lrow[0][0][0]);          // [4 waves][3][64]
        red[(wave * 3 + 0) * 64 + lane] = r1;
        red[(wave * 3 + 1) * 64 + lane] = r2;
        red[(wave * 3 + 2) * 64 + lane] = rn;
        __syncthreads();
        const int T = 16 * NB;
        if (lane <= T && wave < ksplit) {
            double s1 = 0.0, s2 = 0.0, sn = 0.0;
            if (wave == 0) {
#pragma unroll
                for (int w = 0; w < 4; ++w) { s1 += red[(w * 3 + 0) * 64 + lane]; s2 += red[(w * 3 + 1) * 64 + lane]; sn += red[(w * 3 + 2) * 64 + lane]; }
            }
            double *o = p.part + ((int64_t)blockIdx.z * p.nparts + (int64_t)walker * ksplit + wave) * rec;
            o[((int64_t)T * p.K + lane) * 2 + 0] = s1;
            o[((int64_t)T * p.K + lane) * 2 + 1] = same ? s1 : s2;
            if (lane == T) o[(int64_t)p.K * p.K * 2 + T] = sn;
        }
    }
}

// measured against the register-tile kernel: 3x3/2 (K = 9) 37 -> 20 ms, 5x5 18 -> 6 ms, 7x7/2 100 -> 37 ms; 2x2/2 loses (1.7 vs 1.3 ms)
static bool conv_mfma_shape(int64_t K) { return K >= 6 && K <= 64; }

static int64_t conv_mfma_walkers(int64_t nch, int64_t m)
{
    const int64_t nchunks = (m + kCmCH - 1) / kCmCH;
    int64_t x = (1024 + nch - 1) / nch;                  // about four workgroups per CU in all
    if (x > nchunks) x = nchunks;
    return x < 1 ? 1 : x;
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

static std::atomic<int> g_conv_s2{1};
void conv_set_s2(int on) { g_conv_s2.store(on ? 1 : 0, std::memory_order_relaxed); }

size_t gram_conv_workspace_bytes(int64_t K, int64_t nch, int64_t F, int64_t m)
{
    size_t b = 0;
    int64_t parts = conv_walkers(K, nch, m);                                                // partial records (either kernel)
    if (conv_mfma_shape(K) && conv_mfma_walkers(nch, m) * kCmSplit > parts) parts = conv_mfma_walkers(nch, m) * kCmSplit;
    b += al256v((size_t)nch * parts * gram_record(K) * sizeof(double));
    b += al256v((size_t)nch * gram_record(K) * sizeof(double));                             // Gram records
    b += al256v((size_t)nch * K * sizeof(float));                                           // row norms
    b += al256v((size_t)nch * F * K * sizeof(float));                                       // chosen values per filter and step
    b += gram_fix_bytes();
    b += al256v((size_t)nch * sizeof(int));                                                 // negative-activation flags
    b += 256;                                                                               // a zero word for out-of-image taps
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
    if (a.nch == 0 || (a.F == 0 && a.phase != 1)) return hipSuccess;
    const int64_t K = (int64_t)a.kh * a.kw, m = a.n * a.oh * a.ow;
    if (!gram_conv_supported(a.n, a.H, a.W, a.nch, a.kh, a.kw, a.oh, a.ow)) return hipErrorInvalidValue;
    const bool mfma = conv_mfma_shape(K) && !(a.variant & 4);
    const bool same_act = a.act_w == a.act_q;
    const int64_t walkers = mfma ? conv_mfma_walkers(a.nch, m) : conv_walkers(K, a.nch, m);
    const int64_t nparts = mfma ? walkers * (same_act ? 4 : 2) : walkers;
    int64_t maxparts = conv_walkers(K, a.nch, m);
    if (conv_mfma_shape(K) && conv_mfma_walkers(a.nch, m) * kCmSplit > maxparts) maxparts = conv_mfma_walkers(a.nch, m) * kCmSplit;
    const int64_t chunk = mfma ? kCmCH : kGramCH;
    ConvParams p{};
    p.act_w = a.act_w; p.act_q = a.act_q; p.plane = a.n * a.H * a.W;
    p.H = (int)a.H; p.W = (int)a.W; p.kw = a.kw; p.sh = a.sh; p.sw = a.sw; p.rh = a.rh; p.rw = a.rw;
    p.pt = a.pt; p.pl = a.pl; p.oh = (int)a.oh; p.ow = (int)a.ow;
    p.K = (int)K; p.m = (int)m;
    p.nchunks = (int)((m + chunk - 1) / chunk);
    p.same_act = same_act;
    // taps reach from -pt to (oh-1)*sh + (kh-1)*rh - pt (rows), likewise for columns
    p.padded = a.pt > 0 || a.pl > 0 || (a.oh - 1) * a.sh + (int64_t)(a.kh - 1) * a.rh - a.pt >= a.H
               || (a.ow - 1) * a.sw + (int64_t)(a.kw - 1) * a.rw - a.pl >= a.W;
    decompose(64, a.oh, a.ow, p.d64);
    decompose(chunk * walkers, a.oh, a.ow, p.dch);
    p.nparts = (int)nparts;
    char *ws = static_cast<char *>(a.workspace);
    double *part = reinterpret_cast<double *>(ws);  ws += al256v((size_t)a.nch * maxparts * gram_record(K) * sizeof(double));
    double *gram = reinterpret_cast<double *>(ws);  ws += al256v((size_t)a.nch * gram_record(K) * sizeof(double));
    float *nrm = reinterpret_cast<float *>(ws);     ws += al256v((size_t)a.nch * K * sizeof(float));
    float *q32h = reinterpret_cast<float *>(ws);    ws += al256v((size_t)a.nch * a.F * K * sizeof(float));
    void *fixws = ws;                               ws += gram_fix_bytes();
    int *negflag = reinterpret_cast<int *>(ws);     ws += al256v((size_t)a.nch * sizeof(int));
    p.zero = reinterpret_cast<const float *>(ws);
    hipError_t e0 = hipMemsetAsync(negflag, 0, al256v((size_t)a.nch * sizeof(int)) + 256, stream);
    if (e0 != hipSuccess) return e0;
    p.part = part;
    p.negflag = negflag;
    hipError_t e;
    if (a.phase == 2) {                                    // records formed elsewhere (and summed over the column shards)
        e = hipMemcpyAsync(negflag, a.negflags, (size_t)a.nch * sizeof(int), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;
        e = launch_gram_reduce(a.records, 1, (int)K, gram, nrm, a.nch, stream);
        if (e != hipSuccess) return e;
    } else if ((g_conv_s2.load(std::memory_order_relaxed) || a.pix > 1) && a.s2_part &&
               gram_s2_supported(a.n, a.H, a.W, a.kh, a.kw, a.sh, a.sw, a.rh, a.rw, a.pt, a.pl, a.nch)) {
        // 7x7 / stride 2 / VALID: shift sums of the parity classes of the planes instead of every (t, s) product (gpfq_gram_s2.hip)
        e = launch_gram_s2(a.act_w, a.act_q, a.n, a.H, a.W, a.nch, a.s2_part, gram, nrm, negflag, stream, a.pix);
        if (e != hipSuccess) return e;
        if (a.phase == 1) {
            e = hipMemcpyAsync(a.records, gram, (size_t)a.nch * gram_record(K) * sizeof(double), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return e;
            return hipMemcpyAsync(a.negflags, negflag, (size_t)a.nch * sizeof(int), hipMemcpyDeviceToDevice, stream);
        }
    } else {
        if (a.pix > 1) return hipErrorInvalidValue;        // the tile kernels read channel planes
        if (mfma) {
            const dim3 grid((unsigned)walkers, 1, (unsigned)a.nch);
#define GPFQ_CM(NB_, REM_)                                                                                                          \
            do {                                                                                                                    \
                if (p.padded) hipLaunchKernelGGL((gpfq_gram_conv_mfma_kernel<NB_, REM_, true>), grid, dim3(kGramThreads), 0, stream, p);  \
                else hipLaunchKernelGGL((gpfq_gram_conv_mfma_kernel<NB_, REM_, false>), grid, dim3(kGramThreads), 0, stream, p);          \
            } while (0)
            if (K <= 16) GPFQ_CM(1, false);
            else if (K == 17) GPFQ_CM(1, true);
            else if (K <= 32) GPFQ_CM(2, false);
            else if (K == 33) GPFQ_CM(2, true);
            else if (K <= 48) GPFQ_CM(3, false);
            else if (K == 49) GPFQ_CM(3, true);
            else GPFQ_CM(4, false);
#undef GPFQ_CM
        } else {
            hipLaunchKernelGGL((gpfq_gram_conv_kernel<kConvTB, kConvSB>),
                               dim3((unsigned)tile_count<kConvTB, kConvSB>((int)K), (unsigned)nparts, (unsigned)a.nch), dim3(kGramThreads), 0, stream, p);
        }
        e = hipGetLastError();
        if (e != hipSuccess) return e;
        e = launch_gram_reduce(part, nparts, (int)K, gram, nrm, a.nch, stream);
        if (e != hipSuccess) return e;
        if (a.phase == 1) {
            e = hipMemcpyAsync(a.records, gram, (size_t)a.nch * gram_record(K) * sizeof(double), hipMemcpyDeviceToDevice, stream);
            if (e != hipSuccess) return e;
            return hipMemcpyAsync(a.negflags, negflag, (size_t)a.nch * sizeof(int), hipMemcpyDeviceToDevice, stream);
        }
    }
    DecideBatch bs;
    bs.nch = a.nch; bs.gram_cs = gram_record(K); bs.nrm_cs = K; bs.w_cs = a.F * K; bs.out_cs = a.F * K; bs.unc_cs = a.F;
    bs.hist_cs = a.F * K;
    FixSrc src{};
    src.X = a.act_w; src.Xq = a.act_q; src.ld = 0; src.m = m; src.planes = 1; src.plane = a.pix > 1 ? 1 : p.plane; src.pix = a.pix > 1 ? a.pix : 1;
    src.n = (int)a.n; src.H = p.H; src.W = p.W; src.oh = p.oh; src.ow = p.ow;
    src.kw = a.kw; src.sh = a.sh; src.sw = a.sw; src.rh = a.rh; src.rw = a.rw; src.pt = a.pt; src.pl = a.pl;
    e = launch_canonical_norms(src, (int)K, a.nch, nrm, K, fixws, stream);      // few channels: norms in an order fixed by the dimensions
    if (e != hipSuccess) return e;
    return launch_gram_decide(gram, nrm, a.Wt, K, a.A, (int)K, a.F, a.slack, a.qidx, a.Qt, a.uncertified, q32h, bs, &src, fixws, negflag,
                              stream, a.big);
}

}  // namespace gpfq
