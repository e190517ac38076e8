// Gram records of a 7x7 / stride 2 / VALID conv layer (ResNet50's conv1 on the zero-padded input) from SHIFT SUMS of the
// parity classes of each channel's image -- the stride-2 analogue of gpfq_gram_shift_kernel (gpfq_gram_image.hip).
//
// Replaces, for that kernel shape, the patch-matrix inner products behind _quantize_filter2D_parallel_jit
// (scripts/quantized_network.py:185-233; patch rows: :123-183, :729-809): same records, same decide step.
//
// Patch row t = (ky, kx), column (b, oy, ox) of a stride-2 VALID layer is plane[b][2 oy + ky][2 ox + kx].  Split the plane into its four
// parity classes p = (y mod 2, x mod 2), D_p[i][j] = plane[2i + py][2j + px]: row t reads D_{p_t}[oy + ay_t][ox + ax_t] with p_t = (ky mod 2,
// kx mod 2), a_t = (ky div 2, kx div 2) -- a STRIDE-1 shift of a decimated plane.  Hence with A = o + a_t
//
//     <Xq_t, X_s> = sum over A in W_t of Dq_{p_t}[A] * Dx_{p_s}[A + d],   d = a_s - a_t,   W_t = [ay_t, ay_t + oh) x [ax_t, ax_t + ow):
//
// the summand depends on (t, s) only through the class pair and the shift d; only the WINDOW W_t depends on t, and all windows of a
// class share their interior.  The matrix-core kernel (gpfq_gram_conv_mfma_kernel) forms every (t, s) entry from scratch: 2 x 1225
// products per output position = ~580 per input position; here an input position costs 16 roles x 14 = 224 products.
//
//  * Roles: a workgroup of 16 wavefronts, wavefront w <-> (class of s: w mod 4, row shift dy = -(w div 4)).  Lower triangle only
//    (s <= t in row-major (ky, kx) order) needs dy in [-3, 0] and all dx in [-3, 3].  A lane accumulates, for its positions A of ITS class
//    p_t (lane div 16), the seven dx sums of Dq_{p_t}[A] Dx_{p_s}[A + (dy, dx)] and of Dq_{p_t}[A] Dq_{p_s}[A + (dy, dx)] (+ sum Dx[A]^2 in role 0).
//  * Regions: position A belongs to window W_t iff ay_t <= A_y < ay_t + oh (same for x).  Rows [3, oh) and columns [3, 3 + 4 nstrip) belong
//    to every window (the interior); each of the 3 + 3 remaining rows and 3 + (ow - 4 nstrip) remaining columns is its own region.  A launch
//    covers ONE row region (the interior rows, or one border row); within it the interior columns go in strips of four over all 16 lanes
//    of a class (accumulator set A) and border column j is walked by lanes j and j + 8 of each class (set B): every accumulator belongs
//    to one region, and gpfq_gram_s2_combine_kernel ADDS the regions in which a row t qualifies -- never a full sum minus a border,
//    so exactly zero norms (rule (i), :83-84) stay exactly zero.
//  * A band of decimated rows (+ 3 above) of all four classes of both tensors is staged in LDS de-interleaved, 4 zero columns left
//    and >= 4 right of every row, zero rows above the plane: no bounds checks in the sweeps.  The band's full rows are one run of the
//    channel's pixels -- contiguous floats of a channel plane, or every Cin-th float of the NHWC tensor itself (S2Params::pix: the
//    layer then needs no channel planes at all) -- copied by LDS-DMA into a raw area under the previous band's sums and scattered
//    LDS -> LDS into the classes.
#include "gpfq_device.hpp"
#include "gpfq_gram_tile.hpp"
#include "gpfq_launch.hpp"
#include "gpfq_roles.hpp"

namespace gpfq {

namespace {

constexpr int kS2D = 3;                // largest tap shift in a decimated plane: 7 taps -> a in 0..3
constexpr int kS2K = 7;
constexpr int kS2L = 4;                // zero floats left of a staged row: column j at index j + 4 puts a strip's operand positions (4 s ..) on a 16-byte boundary
constexpr int kS2Threads = 1024;       // 16 wavefronts = 4 classes of s x 4 row shifts
constexpr int kS2Acc = 15;             // 7 (G1) + 7 (G2) + sum x^2
constexpr int kS2Slots = 9;            // regions per launch and class: interior columns + up to 8 border columns
constexpr int kS2MaxLaunch = 8;
#ifndef GPFQ_S2_SKIP
#define GPFQ_S2_SKIP 0                  // diagnostic builds: 1 no border columns, 2 no interior strips, 3 classes scattered once only, 4 no staging at all
#endif

struct S2Params {
    const float *act_w, *act_q;
    int64_t plane;                     // floats from one channel to the next: n * H * W for channel planes, 1 for NHWC tensors
    int64_t pix;                       // floats from one pixel to the next: 1 for channel planes, Cin for NHWC tensors
    int n, H, W, Wd;                   // Wd = (W + 1) / 2 decimated columns
    int same_act;
    int nreg;                          // row regions of the launch (blockIdx.y): the interior rows, then one border row each
    int r0s[kS2MaxLaunch], r1s[kS2MaxLaunch];   // position rows [r0, r1) of a region
    int nwgs[kS2MaxLaunch];            // workgroups of a region (blockIdx.x beyond: nothing to do)
    int64_t offs[kS2MaxLaunch];        // its partials inside a channel's block
    int nstrip;                        // interior columns [3, 3 + 4 nstrip)
    int nbc, bc[8];                    // border columns
    int LP;                            // LDS row pitch in floats (column j at index j + kS2L)
    int RB;                            // position rows per band
    int cls;                           // floats per class in LDS: (RB + 3) rows of LP, rounded up to 64
    int rawcap;                        // floats per tensor in the raw area: 2 (RB + 3) full rows
    double *part;                      // partials of channel 0: per region [nwg][4 classes][kS2Slots][16 roles][kS2Acc]
    int64_t part_cs;                   // doubles from one channel's partials (all regions) to the next channel's
    int *negflag;
};

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
template <int OFF> __device__ __forceinline__ void lds_rd128(v4f &v, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void lds_rd64(v2f &v, unsigned addr)
{
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
}

template <bool SAME>                  // both networks see the same planes (a first layer): G2 = G1, one tensor staged
__global__ void __launch_bounds__(kS2Threads)
gpfq_gram_s2_kernel(S2Params p)
{
    extern __shared__ __attribute__((aligned(16))) float s2_lds[];
    const int reg = blockIdx.y;
    if ((int)blockIdx.x >= p.nwgs[reg]) return;
    const int r0 = p.r0s[reg], r1 = p.r1s[reg];
    const int RB = p.RB;                                            // position rows per band
    const int nbands = (r1 - r0 + RB - 1) / RB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ps = wave & 3, dyi = wave >> 2;                       // role: class of s, row shift -dyi
    const int pt = lane >> 4, slot = lane & 15;                     // class of t, lane slot within the class
    const int LP = p.LP;
    // class pitch: a multiple of 256 bytes.  A ds_read_b128 is served in groups of 16 lanes -- 8 of one class with strips {0-3, 12-15}
    // and 8 of the next with strips {4-11} (MI355X_MICROARCH.md, LDS) -- over 64 banks: with equal class bases (mod 256 B) the 16
    // strips of a group are the 16 slots of a bank row (a pitch of 128 B more, as in the first version, made every own-row read 2-way).
    const int cls_stride = p.cls;
    float *Lx = s2_lds;                                             // [4][LR][LP]
    float *Lq = SAME ? s2_lds : s2_lds + 4 * cls_stride;
    const unsigned lx_base = lds_addr(Lx), lq_base = lds_addr(Lq);
    float *raw = s2_lds + (SAME ? 4 : 8) * cls_stride;              // [tensors][rawcap]: the band's full rows as they lie in the plane
    const float *pw = p.act_w + (int64_t)blockIdx.z * p.plane;
    const float *pq = p.act_q + (int64_t)blockIdx.z * p.plane;

    double a1[7], a2[7], a3 = 0.0;                                  // set A: interior columns
    double b1[7], b2[7], b3 = 0.0;                                  // set B: border column (slot mod 8)
#pragma unroll
    for (int i = 0; i < 7; ++i) { a1[i] = a2[i] = b1[i] = b2[i] = 0.0; }
    unsigned signs = 0;

    // pads (left 4, right >= 4 floats of every row) are zero for the whole launch: zero everything once
    for (int i = tid; i < (SAME ? 4 : 8) * cls_stride; i += kS2Threads) s2_lds[i] = 0.f;
    __syncthreads();

    const int items = p.n * nbands;
    // Staging of band `item`: its full rows [2 (y0 - 3), 2 y1) are ONE contiguous piece of the plane -- LDS-DMA (4 bytes per lane,
    // no registers, no alignment demands) copies it to the raw area while the previous band is being summed; the de-interleave
    // pass (LDS -> LDS) then scatters it into the four classes.  Rows above / below the plane are not read (zero-filled below).
    auto band_rows = [&](int item, int &img, int &y0, int &rows, int &fs, int &fe) {
        img = item / nbands;
        const int band = item - img * nbands;
        y0 = r0 + band * RB;
        rows = min(r1, y0 + RB) - y0;
        const int f0 = 2 * (y0 - kS2D);
        fs = max(f0, 0); fe = min(f0 + 2 * (rows + kS2D), p.H);
    };
    auto issue = [&](int item) {
        int img, y0, rows, fs, fe;
        band_rows(item, img, y0, rows, fs, fe);
        const int total = (fe - fs) * p.W;
        const int64_t at = ((int64_t)img * p.H + fs) * p.W * p.pix;
#pragma unroll
        for (int tz = 0; tz < (SAME ? 1 : 2); ++tz) {
            const float *src = (tz ? pq : pw) + at;
            const unsigned dst = lds_addr(raw + tz * p.rawcap);
            // (NHWC tensors: the piece is every pix-th float -- a gathering request, the LDS image is the same)
            for (int c = wave * 64; c < total && GPFQ_S2_SKIP != 4; c += kS2Threads) {
                if (c + lane < total) glds4(src + (int64_t)(c + lane) * p.pix, dst + 4u * (unsigned)c);
            }
        }
    };
    if ((int)blockIdx.x < items) issue(blockIdx.x);
    for (int it = blockIdx.x; it < items; it += p.nwgs[reg]) {
        int img, y0, rows, fs, fe;
        band_rows(it, img, y0, rows, fs, fe);
        const int nfull = 2 * (rows + kS2D), f0 = 2 * (y0 - kS2D);
        dma_wait();
        __syncthreads();                                            // the raw rows have landed; the previous band's sums are done
#pragma unroll
        for (int tz = 0; tz < (SAME ? 1 : 2) && (GPFQ_S2_SKIP < 3 || it == (int)blockIdx.x); ++tz) {
            const float *rw_ = raw + tz * p.rawcap;
            float *dst = tz ? Lq : Lx;
            for (int fr = wave; fr < nfull; fr += kS2Threads / 64) {
                const int f = f0 + fr, li = fr >> 1, py = fr & 1;
                const bool rin = f >= fs && f < fe;
                const float *rr = rw_ + (f - fs) * p.W;
                float *dd = dst + py * 2 * cls_stride + li * LP + kS2L;
                for (int c0 = 0; c0 < p.W; c0 += 256) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = c0 + 64 * j + lane;
                        v[j] = (rin && c < p.W) ? rr[c] : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int c = c0 + 64 * j + lane;
                        if (c < p.W) { dd[(c & 1) * cls_stride + (c >> 1)] = v[j]; neg_track(signs, v[j]); }
                    }
                }
            }
        }
        __syncthreads();                                            // the classes are ready, the raw area is free
        if (it + p.nwgs[reg] < items) issue(it + p.nwgs[reg]);
        // ---- set A: strips of four interior columns, all 16 lanes of a class ----
        // (item k = row * nstrip + strip, k = slot, slot + 16, ...: stepped without a division)
        int row = 0, strip = slot;
        while (strip >= p.nstrip) { strip -= p.nstrip; ++row; }
        int oq = pt * cls_stride + (row + kS2D) * LP + kS2L + 4 * strip;          // own row: positions 3 + 4 strip .. are floats 3 .. 6 from here
        int ow_ = oq + (ps - pt) * cls_stride - dyi * LP;                         // operand row: positions (3 + 4 strip) - 3 ..
        const int wrap = LP - 4 * p.nstrip;                                    // from the end of a row's strips to the next row's first
        const int drow = 16 / p.nstrip, dstrip = 16 - drow * p.nstrip, dstep = drow * LP + 4 * dstrip;
        for (; row < rows && GPFQ_S2_SKIP != 2;) {
            // (reads issued by hand: left to itself the compiler narrows the own-row pair to ds_read2_b32 -- 32-bank addressing, 4-way
            //  conflicts between the lanes of two classes -- and waits for all eight before the first product)
            const unsigned aq = lq_base + 4u * (unsigned)oq, ax = lx_base + 4u * (unsigned)ow_;
            v4f q0, q1, x0, x1, w0, w1;
            v2f x2, w2;
            lds_rd128<0>(q0, aq); lds_rd128<16>(q1, aq);
            lds_rd128<0>(x0, ax); lds_rd128<16>(x1, ax); lds_rd64<32>(x2, ax);
            if (!SAME) {
                const unsigned aw = lq_base + 4u * (unsigned)ow_;
                lds_rd128<0>(w0, aw); lds_rd128<16>(w1, aw); lds_rd64<32>(w2, aw);
                asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(q0), "+v"(q1), "+v"(x0), "+v"(x1), "+v"(x2));
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q0), "+v"(q1), "+v"(x0), "+v"(x1), "+v"(x2));
            }
            const double qa[4] = {(double)q0.w, (double)q1.x, (double)q1.y, (double)q1.z};
            {
                const double xw[10] = {(double)x0.x, (double)x0.y, (double)x0.z, (double)x0.w, (double)x1.x, (double)x1.y, (double)x1.z,
                                       (double)x1.w, (double)x2.x, (double)x2.y};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int d = 0; d < 7; ++d) a1[d] = fma(qa[e], xw[e + d], a1[d]);
            }
            if (!SAME) {
                // (the first Gram's sums are operands of the wait: they stay in front of it, under the second operand's latency)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(a1[0]), "+v"(a1[1]), "+v"(a1[2]), "+v"(a1[3]),
                             "+v"(a1[4]), "+v"(a1[5]), "+v"(a1[6]));
                const double qw[10] = {(double)w0.x, (double)w0.y, (double)w0.z, (double)w0.w, (double)w1.x, (double)w1.y, (double)w1.z,
                                       (double)w1.w, (double)w2.x, (double)w2.y};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int d = 0; d < 7; ++d) a2[d] = fma(qa[e], qw[e + d], a2[d]);
            }
            if (wave == 0) {                                        // squared norms of the X rows: the own-class X at the own positions
                const float4 *U4 = reinterpret_cast<const float4 *>(__builtin_assume_aligned(Lx, 16)) + (oq >> 2);
                const float4 u0 = U4[0], u1 = U4[1];
                const double xa[4] = {(double)u0.w, (double)u1.x, (double)u1.y, (double)u1.z};
#pragma unroll
                for (int e = 0; e < 4; ++e) a3 = fma(xa[e], xa[e], a3);
            }
            strip += dstrip; row += drow; oq += dstep; ow_ += dstep;          // 16 items on
            const bool w = strip >= p.nstrip;
            strip -= w ? p.nstrip : 0; row += w ? 1 : 0; oq += w ? wrap : 0; ow_ += w ? wrap : 0;
        }
        // ---- set B: border column (slot mod 8), the rows split between lanes slot and slot + 8 ----
        if ((slot & 7) < p.nbc && GPFQ_S2_SKIP != 1) {
            const int cx = p.bc[slot & 7];
            for (int row = slot >> 3; row < rows; row += 2) {
                const int oq = pt * cls_stride + (row + kS2D) * LP + kS2L + cx;
                const int ow_ = ps * cls_stride + (row + kS2D - dyi) * LP + kS2L + cx - kS2D;
                const double qa = (double)Lq[oq];
#pragma unroll
                for (int d = 0; d < 7; ++d) b1[d] = fma(qa, (double)Lx[ow_ + d], b1[d]);
                if (!SAME) {
#pragma unroll
                    for (int d = 0; d < 7; ++d) b2[d] = fma(qa, (double)Lq[ow_ + d], b2[d]);
                }
                if (wave == 0) { const double xa = (double)Lx[oq]; b3 = fma(xa, xa, b3); }
            }
        }
        __syncthreads();
    }
    if (__ballot(neg_seen(signs)) && lane == 0) atomicOr(p.negflag + blockIdx.z, 1);

    // ---- per-workgroup partial sums: set A summed over the 16 lanes of a class (a DPP row), set B over lanes j, j + 8 ----
    double *out = p.part + (int64_t)blockIdx.z * p.part_cs + p.offs[reg] + ((int64_t)blockIdx.x * 4 + pt) * kS2Slots * 16 * kS2Acc;
    auto row_sum = [](double v) { v = ror_add<8>(v); v = ror_add<4>(v); v = ror_add<2>(v); v = ror_add<1>(v); return v; };
#pragma unroll
    for (int d = 0; d < 7; ++d) {
        const double s1 = row_sum(a1[d]), s2 = row_sum(SAME ? a1[d] : a2[d]);
        const double t1 = ror_add<8>(b1[d]), t2 = ror_add<8>(SAME ? b1[d] : b2[d]);
        if (slot == 0) { out[(0 * 16 + wave) * kS2Acc + d] = s1; out[(0 * 16 + wave) * kS2Acc + 7 + d] = s2; }
        if (slot < 8) { out[((1 + slot) * 16 + wave) * kS2Acc + d] = t1; out[((1 + slot) * 16 + wave) * kS2Acc + 7 + d] = t2; }
    }
    {
        const double s3 = row_sum(a3), t3 = ror_add<8>(b3);
        if (slot == 0) out[(0 * 16 + wave) * kS2Acc + 14] = s3;
        if (slot < 8) out[((1 + slot) * 16 + wave) * kS2Acc + 14] = t3;
    }
}

struct S2Launch { int64_t off; int nwg; int r0, r1; };              // partials of a launch, its position rows
struct S2Combine {
    S2Launch L[kS2MaxLaunch];
    int nlaunch;
    int oh, ow, nstrip, nbc, bc[8];
    int64_t part_per_channel;                                       // doubles
};

// First stage of the combine: a launch region's workgroup blocks ([4 classes][kS2Slots][16 roles][kS2Acc] doubles each, ~1300 per
// channel at 4096 images) added in chunks of kS2Chunk, in place into the chunk's first block.  The second stage reads ONE double per
// block and entry, 69 KB apart -- over all blocks that was 88 MB per channel fetched a sector at a time (0.37 ms of conv1's 12.7);
// here the blocks are read once, whole lines at a time.
constexpr int kS2Chunk = 32;
constexpr int kS2Block = 4 * kS2Slots * 16 * kS2Acc;
__global__ void __launch_bounds__(256)
gpfq_gram_s2_chunks_kernel(double *__restrict__ part, S2Combine c)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kS2Block) return;
    const int64_t ch = blockIdx.z;
    int l = 0, j = blockIdx.y;                                      // chunk j of launch l
    while (l < c.nlaunch && j >= (c.L[l].nwg + kS2Chunk - 1) / kS2Chunk) { j -= (c.L[l].nwg + kS2Chunk - 1) / kS2Chunk; ++l; }
    if (l >= c.nlaunch) return;
    const int g0 = j * kS2Chunk, g1 = g0 + kS2Chunk < c.L[l].nwg ? g0 + kS2Chunk : c.L[l].nwg;
    double *p = part + ch * c.part_per_channel + c.L[l].off + i;
    double v = p[(int64_t)g0 * kS2Block];
    for (int g = g0 + 1; g < g1; ++g) v += p[(int64_t)g * kS2Block];
    p[(int64_t)g0 * kS2Block] = v;
}

// Regions -> the K = 49 Gram record of a channel (layout of gpfq_gram.hip: [t][s][G1, G2] lower triangle, then the squared X-row norms)
// + the float32 row norms.  Entry (t, s): every region in which row t qualifies, in launch / slot / workgroup order.
__global__ void __launch_bounds__(256)
gpfq_gram_s2_combine_kernel(const double *__restrict__ part, S2Combine c, double *__restrict__ gram, float *__restrict__ nrm32)
{
    constexpr int K = kS2K * kS2K;
    const int64_t ch = blockIdx.y;
    const int64_t rec = gram_record(K);
    const int lane = threadIdx.x & 63;
    const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // one wavefront per entry: lanes split the workgroups' partials
    if (e >= rec) return;
    int t, s, k;
    if (e < (int64_t)K * K * 2) { k = (int)(e & 1); t = (int)((e >> 1) / K); s = (int)((e >> 1) - (int64_t)t * K); }
    else { t = s = (int)(e - (int64_t)K * K * 2); k = 2; }
    double v = 0.0;
    if (s <= t) {
        const int kyt = t / kS2K, kxt = t - kyt * kS2K, kys = s / kS2K, kxs = s - kys * kS2K;
        const int ptc = (kyt & 1) * 2 + (kxt & 1), psc = (kys & 1) * 2 + (kxs & 1);
        const int ayt = kyt >> 1, axt = kxt >> 1, dy = (kys >> 1) - ayt, dx = (kxs >> 1) - axt;
        // (s <= t in row-major order of (ky, kx) implies dy <= 0; k == 2: the x^2 sum of role 0)
        const int role = k == 2 ? 0 : psc + 4 * (-dy);
        const int acc = k == 2 ? 14 : (dx + kS2D) + 7 * k;
        const double *pc = part + ch * c.part_per_channel;
        for (int l = 0; l < c.nlaunch; ++l) {
            const S2Launch &L = c.L[l];
            // a launch is the interior rows [3, oh) -- in every window -- or ONE border row r: in W_t iff ayt <= r < ayt + oh
            const bool rows_in = (L.r1 - L.r0 > 1) || (L.r0 >= ayt && L.r0 < ayt + c.oh);
            if (!rows_in) continue;
            for (int sl = 0; sl < kS2Slots; ++sl) {
                if (sl > 0) {
                    if (sl - 1 >= c.nbc) break;
                    const int col = c.bc[sl - 1];
                    if (!(col >= axt && col < axt + c.ow)) continue;
                }
                double w = 0.0;
                for (int g = lane * kS2Chunk; g < L.nwg; g += 64 * kS2Chunk)        // (the chunk sums: gpfq_gram_s2_chunks_kernel)
                    w += pc[L.off + ((((int64_t)g * 4 + ptc) * kS2Slots + sl) * 16 + role) * kS2Acc + acc];
                v += wave_sum(w);
            }
        }
    }
    if (lane == 0) {
        gram[ch * rec + e] = v;
        if (nrm32 && k == 1 && s == t) nrm32[ch * K + t] = (float)sqrt(v);
    }
}

struct S2Plan {
    int oh, ow, Wd, nstrip, nbc, bc[8], LP, RB, cls, rawcap;
    int nlaunch;
    int r0[kS2MaxLaunch], r1[kS2MaxLaunch], nwg[kS2MaxLaunch];
    int64_t off[kS2MaxLaunch];
    int64_t part_per_channel;
    size_t lds;
};

bool s2_plan(int64_t n, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int rh, int rw, int pt, int pl, bool same_act, S2Plan *P)
{
    if (kh != kS2K || kw != kS2K || sh != 2 || sw != 2 || rh != 1 || rw != 1 || pt != 0 || pl != 0) return false;
    if (H < 32 || W < 32 || H > 4096 || W > 4096 || n < 1 || n * H * W >= (1LL << 40)) return false;
    const int oh = (int)((H - kS2K) / 2 + 1), ow = (int)((W - kS2K) / 2 + 1);
    if (oh <= 2 * kS2D + 1 || ow <= 2 * kS2D + 4) return false;
    S2Plan q{};
    q.oh = oh; q.ow = ow; q.Wd = (int)((W + 1) / 2);
    q.nstrip = (ow - kS2D) / 4;
    q.nbc = 0;
    for (int c = 0; c < kS2D; ++c) q.bc[q.nbc++] = c;
    for (int c = kS2D + 4 * q.nstrip; c < ow + kS2D; ++c) { if (q.nbc >= 8) return false; q.bc[q.nbc++] = c; }
    q.LP = (kS2L + q.Wd + kS2D + 4 + 3) & ~3;                        // 4 zeros left, the row, >= 7 zeros right
    // pitch = 4 (mod 32) floats: the border-column walks read one column of two consecutive rows per class (ds_read_b32, 32 banks):
    // with a pitch of 0 (mod 32) -- 128 floats for ResNet50's conv1 -- the two rows of every read share their banks (2-way)
    while ((q.LP & 31) != 4) q.LP += 4;
    // the longest band whose classes (4 per tensor) and raw rows fit the 160 KiB of a CU
    const int ntz = same_act ? 1 : 2;
    q.RB = 0;
    for (int rb = 32; rb >= 4; rb >>= 1) {
        const int rawcap = (int)((2 * (rb + kS2D) * W + 63) & ~(int64_t)63);
        const int cls = ((rb + kS2D) * q.LP + 63) & ~63;
        const size_t lds = ((size_t)ntz * 4 * cls + (size_t)ntz * rawcap) * sizeof(float);
        if (lds <= 158 * 1024) { q.RB = rb; q.cls = cls; q.rawcap = rawcap; q.lds = lds; break; }
    }
    if (!q.RB) return false;
    // launches: the interior rows, then every border row
    int l = 0;
    int64_t off = 0;
    auto add = [&](int r0, int r1, int nwg) {
        q.r0[l] = r0; q.r1[l] = r1; q.nwg[l] = nwg; q.off[l] = off;
        off += (int64_t)nwg * 4 * kS2Slots * 16 * kS2Acc;
        ++l;
    };
    const int64_t items = n * ((oh - kS2D + q.RB - 1) / q.RB);
    add(kS2D, oh, (int)(items < 512 ? items : 512));
    for (int r = 0; r < kS2D; ++r) add(r, r + 1, (int)(n < 128 ? n : 128));
    for (int r = oh; r < oh + kS2D; ++r) add(r, r + 1, (int)(n < 128 ? n : 128));
    q.nlaunch = l;
    q.part_per_channel = off;
    *P = q;
    return true;
}

}  // namespace

// nch: channels of the call.  The plan launches a fixed 512 + 6 * 128 workgroups PER CHANNEL, each writing 4 * 9 * 16 * 15 doubles of
// partial sums (88 MB per channel): right for an image input (ResNet50's conv1: 3 channels, 265 MB), 5.6 GB and 82 K workgroups for a
// 64-channel layer -- those stay on the matrix-core kernel.
constexpr int64_t kS2MaxChannels = 8;
bool gram_s2_supported(int64_t n, int64_t H, int64_t W, int kh, int kw, int sh, int sw, int rh, int rw, int pt, int pl, int64_t nch)
{
    if (nch > kS2MaxChannels) return false;
    S2Plan P;
    return s2_plan(n, H, W, kh, kw, sh, sw, rh, rw, pt, pl, false, &P);
}

size_t gram_s2_workspace_bytes(int64_t n, int64_t H, int64_t W, int64_t nch)
{
    S2Plan P;
    if (!s2_plan(n, H, W, kS2K, kS2K, 2, 2, 1, 1, 0, 0, false, &P)) return 0;
    return ((size_t)nch * (size_t)P.part_per_channel * sizeof(double) + 255) & ~(size_t)255;
}

// Gram records [nch][gram_record(49)] + float32 row norms of all channels; `part` = gram_s2_workspace_bytes bytes.
// pix = 1: act_* are channel planes [nch][n][H][W]; pix = Cin > 1: NHWC tensors offset to the shard's first channel.
hipError_t launch_gram_s2(const float *act_w, const float *act_q, int64_t n, int64_t H, int64_t W, int64_t nch, double *part,
                          double *gram, float *nrm32, int *negflag, hipStream_t stream, int64_t pix)
{
    const bool same_act = act_w == act_q;
    S2Plan P;
    if (!s2_plan(n, H, W, kS2K, kS2K, 2, 2, 1, 1, 0, 0, same_act, &P)) return hipErrorInvalidValue;
    hipError_t attr = same_act ? ensure_dynamic_lds((const void *)gpfq_gram_s2_kernel<true>, P.lds)
                               : ensure_dynamic_lds((const void *)gpfq_gram_s2_kernel<false>, P.lds);
    if (attr != hipSuccess) return attr;
    S2Params p{};
    p.act_w = act_w; p.act_q = act_q; p.plane = pix > 1 ? 1 : n * H * W; p.pix = pix > 1 ? pix : 1; p.n = (int)n; p.H = (int)H; p.W = (int)W; p.Wd = P.Wd;
    p.same_act = same_act ? 1 : 0;
    p.nstrip = P.nstrip; p.nbc = P.nbc;
    for (int i = 0; i < 8; ++i) p.bc[i] = P.bc[i];
    p.LP = P.LP; p.RB = P.RB; p.cls = P.cls; p.rawcap = P.rawcap; p.negflag = negflag; p.part_cs = P.part_per_channel;
    S2Combine c{};
    c.nlaunch = P.nlaunch; c.oh = P.oh; c.ow = P.ow; c.nstrip = P.nstrip; c.nbc = P.nbc;
    for (int i = 0; i < 8; ++i) c.bc[i] = P.bc[i];
    c.part_per_channel = P.part_per_channel;
    int maxwg = 0;
    p.nreg = P.nlaunch; p.part = part;
    for (int l = 0; l < P.nlaunch; ++l) {
        p.r0s[l] = P.r0[l]; p.r1s[l] = P.r1[l]; p.nwgs[l] = P.nwg[l]; p.offs[l] = P.off[l];
        c.L[l].off = P.off[l]; c.L[l].nwg = P.nwg[l]; c.L[l].r0 = P.r0[l]; c.L[l].r1 = P.r1[l];
        if (P.nwg[l] > maxwg) maxwg = P.nwg[l];
    }
    // ONE launch: blockIdx.y = row region (the border rows' workgroups run beside the interior's instead of after it)
    if (same_act) hipLaunchKernelGGL(gpfq_gram_s2_kernel<true>, dim3((unsigned)maxwg, (unsigned)P.nlaunch, (unsigned)nch), dim3(kS2Threads), P.lds, stream, p);
    else hipLaunchKernelGGL(gpfq_gram_s2_kernel<false>, dim3((unsigned)maxwg, (unsigned)P.nlaunch, (unsigned)nch), dim3(kS2Threads), P.lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    int nchunks = 0;
    for (int l = 0; l < P.nlaunch; ++l) nchunks += (P.nwg[l] + kS2Chunk - 1) / kS2Chunk;
    hipLaunchKernelGGL(gpfq_gram_s2_chunks_kernel, dim3((unsigned)((kS2Block + 255) / 256), (unsigned)nchunks, (unsigned)nch), dim3(256), 0, stream, part, c);
    const int64_t rec = gram_record(kS2K * kS2K);
    hipLaunchKernelGGL(gpfq_gram_s2_combine_kernel, dim3((unsigned)((rec + 3) / 4), (unsigned)nch), dim3(256), 0, stream, part, c, gram, nrm32);
    return hipGetLastError();
}

}  // namespace gpfq
