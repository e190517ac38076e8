// Cross-lane plumbing, LDS-DMA and typed LDS access shared by the role-split dense kernels
// (gpfq_pipe.hip: one step per slot; gpfq_blk.hip: a block of steps per slot).  gfx950 / wave64 only.
#pragma once

#include "gpfq_device.hpp"

namespace gpfq {
namespace {

constexpr int kSweepWaves = 8;

// ---- cross-lane plumbing of the hot loop -------------------------------------------------------
template <int ROR>
__device__ __forceinline__ double ror_add(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), 0x120 + ROR, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), 0x120 + ROR, 0xF, 0xF, true);
    return x + __hiloint2double(hi, lo);
}

// v_permlane32_swap on a float64 pair: returns (x' + y') with x' = [x.lo32, y.lo32], y' = [x.hi32, y.hi32]:
// lanes 0-31 get x[l] + x[l+32], lanes 32-63 get y[l-32] + y[l].
__device__ __forceinline__ double fold32(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// v_permlane16_swap: x' = [x.r0, y.r0, x.r2, y.r2], y' = [x.r1, y.r1, x.r3, y.r3]; returns x' + y':
// rows 0, 2 get x.r0 + x.r1 / x.r2 + x.r3, rows 1, 3 get y.r0 + y.r1 / y.r2 + y.r3.
__device__ __forceinline__ double fold16(double x, double y)
{
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}

// LDS-DMA (global_load_lds): lane l's 16 (4) bytes land at lds_dst + 16 l (4 l); lds_dst is a wave-uniform LDS byte
// address.  Issued from inline asm so that hipcc does not count it: with the builtin form it puts s_waitcnt vmcnt(0)
// in front of the next ds_read of ANY LDS address and the prefetch of the next tile would stop the current one
// (cdna_hip_programming.md 5.7).  The tile loop waits for the DMA itself (dma_wait) before its barrier.  M0 is
// saved and restored inside the statement.
__device__ __forceinline__ void glds16(const void *g, unsigned lds_dst_uniform)
{
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *g, unsigned lds_dst_uniform)
{
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(lds_dst) : "memory");
}
// The same with a wave-uniform base address in scalar registers and a 32-bit per-lane byte offset (global_load_lds ..., v, s[..]):
// stepping the base from piece to piece is scalar arithmetic, so an LDS-DMA piece in the middle of a vector-bound loop costs
// the loop one vector-memory issue and no vector ALU work.
__device__ __forceinline__ void glds16_s(const void *base_uniform, unsigned lane_off, unsigned lds_dst_uniform)
{
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    const unsigned long long a = (unsigned long long)(uintptr_t)base_uniform;
    const unsigned long long sa = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                                  (unsigned)__builtin_amdgcn_readfirstlane((int)a);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane_off), "s"(sa), "s"(lds_dst) : "memory");
}
// NC LDS-DMA requests of 4 bytes per lane with ONE write of M0 and SCALAR addresses: request i reads row + off[i] + lane_off and lands at
// lds_dst + 256 i + 4 lane.  The instruction's immediate offset (256 i) moves both addresses, so off[i] is (byte offset of element i
// from the row base) - 256 i + 2048 and `row` the row base - 2048: every off[i] is then an unsigned 32-bit number and the 64-bit
// address of a request is two scalar additions (into s[96:99], named in the clobber list: the load wants an aligned register pair) --
// no vector instruction per request (a per-lane 64-bit address costs two to three).
__device__ __forceinline__ void glds4_row9(unsigned row_lo, unsigned row_hi, const unsigned (&off)[9], unsigned lane_off, unsigned lds_dst_uniform)
{
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\t"
                 "s_add_u32 s96, %[rlo], %[o0]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97]\n\t"
                 "s_add_u32 s98, %[rlo], %[o1]\n\ts_addc_u32 s99, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[98:99] offset:256\n\t"
                 "s_add_u32 s96, %[rlo], %[o2]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97] offset:512\n\t"
                 "s_add_u32 s98, %[rlo], %[o3]\n\ts_addc_u32 s99, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[98:99] offset:768\n\t"
                 "s_add_u32 s96, %[rlo], %[o4]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97] offset:1024\n\t"
                 "s_add_u32 s98, %[rlo], %[o5]\n\ts_addc_u32 s99, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[98:99] offset:1280\n\t"
                 "s_add_u32 s96, %[rlo], %[o6]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97] offset:1536\n\t"
                 "s_add_u32 s98, %[rlo], %[o7]\n\ts_addc_u32 s99, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[98:99] offset:1792\n\t"
                 "s_add_u32 s96, %[rlo], %[o8]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97] offset:2048\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [vo] "v"(lane_off), [dst] "s"(lds_dst), [rlo] "s"(row_lo), [rhi] "s"(row_hi), [o0] "s"(off[0]), [o1] "s"(off[1]), [o2] "s"(off[2]),
                   [o3] "s"(off[3]), [o4] "s"(off[4]), [o5] "s"(off[5]), [o6] "s"(off[6]), [o7] "s"(off[7]), [o8] "s"(off[8])
                 : "memory", "scc", "s96", "s97", "s98", "s99");
}
__device__ __forceinline__ void glds4_row5(unsigned row_lo, unsigned row_hi, const unsigned (&off)[5], unsigned lane_off, unsigned lds_dst_uniform)
{
    const unsigned lds_dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_dst_uniform);
    unsigned keep;
    asm volatile("s_mov_b32 %[keep], m0\n\ts_mov_b32 m0, %[dst]\n\t"
                 "s_add_u32 s96, %[rlo], %[o0]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97]\n\t"
                 "s_add_u32 s98, %[rlo], %[o1]\n\ts_addc_u32 s99, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[98:99] offset:256\n\t"
                 "s_add_u32 s96, %[rlo], %[o2]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97] offset:512\n\t"
                 "s_add_u32 s98, %[rlo], %[o3]\n\ts_addc_u32 s99, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[98:99] offset:768\n\t"
                 "s_add_u32 s96, %[rlo], %[o4]\n\ts_addc_u32 s97, %[rhi], 0\n\tglobal_load_lds_dword %[vo], s[96:97] offset:1024\n\t"
                 "s_mov_b32 m0, %[keep]"
                 : [keep] "=&s"(keep)
                 : [vo] "v"(lane_off), [dst] "s"(lds_dst), [rlo] "s"(row_lo), [rhi] "s"(row_hi), [o0] "s"(off[0]), [o1] "s"(off[1]), [o2] "s"(off[2]),
                   [o3] "s"(off[3]), [o4] "s"(off[4])
                 : "memory", "scc", "s96", "s97", "s98", "s99");
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)p;
}

// ---- more cross-lane plumbing -------------------------------------------------------------------
// Four per-lane values summed over the k-lanes of a sweep wavefront (lane = ng + G * kq): packed butterfly over lane
// bits 5 and 4 (rows), then rotations inside the rows for the k-lane bits below 4.  Afterwards row i of the wavefront
// holds the sums of the lane's neuron i, lane-in-row ng + G * j (any j) the one of neuron group ng.
template <int G>
__device__ __forceinline__ double fold_klanes(const double (&a)[4])
{
    const double s02 = fold32(a[0], a[2]);           // halves: a0 | a2
    const double s13 = fold32(a[1], a[3]);           //         a1 | a3
    double x = fold16(s02, s13);                     // rows:   a0, a1, a2, a3
    if constexpr (G <= 8) x = ror_add<8>(x);
    if constexpr (G <= 4) x = ror_add<4>(x);
    if constexpr (G <= 2) x = ror_add<2>(x);
    if constexpr (G <= 1) x = ror_add<1>(x);
    return x;
}

// The same for NL = 4, 2 or 1 values per lane.  Two values: one half-wave swap packs them (lanes 0-31 / 32-63), the row swap
// of the packed value with itself folds rows 0+1 and 2+3 -- afterwards rows 0, 1 hold value 0 and rows 2, 3 value 1.  One value:
// both swaps with itself, every row holds it.
template <int G, int NL>
__device__ __forceinline__ double fold_klanes_n(const double (&a)[NL])
{
    if constexpr (NL == 4) {
        return fold_klanes<G>(a);
    } else {
        double x = NL == 2 ? fold32(a[0], a[NL - 1]) : fold32(a[0], a[0]);
        x = fold16(x, x);
        if constexpr (G <= 8) x = ror_add<8>(x);
        if constexpr (G <= 4) x = ror_add<4>(x);
        if constexpr (G <= 2) x = ror_add<2>(x);
        if constexpr (G <= 1) x = ror_add<1>(x);
        return x;
    }
}

template <int CTRL>
__device__ __forceinline__ double dpp_add(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return x + __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ int dpp_addi(int x) { return x + __builtin_amdgcn_mov_dpp(x, CTRL, 0xF, 0xF, true); }

// Sum over the R adjacent sub-lanes of a neuron in the decision wavefront (R = 1, 2, 4, 8), identical bits in all of
// them: quad_perm [1,0,3,2] (0xB1), quad_perm [2,3,0,1] (0x4E), row_half_mirror (0x141).
template <int R> __device__ __forceinline__ double sub_sum(double x)
{
    if constexpr (R >= 2) x = dpp_add<0xB1>(x);
    if constexpr (R >= 4) x = dpp_add<0x4E>(x);
    if constexpr (R >= 8) x = dpp_add<0x141>(x);
    return x;
}
template <int R> __device__ __forceinline__ int sub_sumi(int x)
{
    if constexpr (R >= 2) x = dpp_addi<0xB1>(x);
    if constexpr (R >= 4) x = dpp_addi<0x4E>(x);
    if constexpr (R >= 8) x = dpp_addi<0x141>(x);
    return x;
}

// One barrier per slot: LDS writes of this wavefront done (lgkmcnt), then s_barrier.  Raw instructions: __syncthreads()
// would also wait for the output stores (vmcnt) every step.  The "memory" clobber keeps LDS accesses on their side.
__device__ __forceinline__ void slot_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

typedef float pk2 __attribute__((ext_vector_type(2)));

// Sample-pair split over the eight sweep wavefronts (pairs per k-lane).  Wavefronts 0 and 4 share their SIMD with the
// decision wavefront (a workgroup's wavefronts go to the SIMDs in cyclic order: speed only), so they get fewer.
template <int S> struct PairSplit;
template <> struct PairSplit<32> { static constexpr int pw[8] = {2, 4, 4, 4, 3, 5, 5, 5}; };
template <> struct PairSplit<24> { static constexpr int pw[8] = {1, 3, 3, 3, 2, 4, 4, 4}; };
template <> struct PairSplit<16> { static constexpr int pw[8] = {1, 2, 2, 2, 1, 3, 3, 2}; };
template <> struct PairSplit<12> { static constexpr int pw[8] = {1, 2, 2, 1, 1, 2, 2, 1}; };
template <> struct PairSplit<8>  { static constexpr int pw[8] = {1, 1, 1, 1, 1, 1, 1, 1}; };
template <> struct PairSplit<40> { static constexpr int pw[8] = {5, 5, 5, 5, 5, 5, 5, 5}; };

// LDS through 32-bit address-space-3 pointers: offsets stay 32-bit integer arithmetic (generic pointers into the
// dynamic LDS array cost 64-bit adds and multiplies per access in the hot loop).
typedef __attribute__((address_space(3))) char lchar;
typedef float  nf2 __attribute__((ext_vector_type(2)));
typedef float  nf4 __attribute__((ext_vector_type(4)));
typedef double nd2 __attribute__((ext_vector_type(2)));
typedef int    ni2 __attribute__((ext_vector_type(2)));
// (HIP's float2 / double2 ... are classes and cannot be read through an address-space pointer: native vectors underneath)
template <typename T> struct LdsNative { using type = T; };
template <> struct LdsNative<float2>  { using type = nf2; };
template <> struct LdsNative<float4>  { using type = nf4; };
template <> struct LdsNative<double2> { using type = nd2; };
template <> struct LdsNative<int2>    { using type = ni2; };
template <typename T> __device__ __forceinline__ T lds_ld(lchar *base, int off)
{
    using NT = typename LdsNative<T>::type;
    const NT v = *reinterpret_cast<__attribute__((address_space(3))) const NT *>(base + off);
    T out;
    __builtin_memcpy(&out, &v, sizeof(T));
    return out;
}
template <typename T> __device__ __forceinline__ void lds_st(lchar *base, int off, const T &v)
{
    using NT = typename LdsNative<T>::type;
    NT nv;
    __builtin_memcpy(&nv, &v, sizeof(T));
    *reinterpret_cast<__attribute__((address_space(3))) NT *>(base + off) = nv;
}

}  // namespace
}  // namespace gpfq
