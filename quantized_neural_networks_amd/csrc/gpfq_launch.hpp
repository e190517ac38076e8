// Host-side launch interfaces between the C ABI (gpfq_capi.hip) and the kernel files.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "gpfq_device.hpp"

namespace gpfq {

// Per-row quantities of the certified mode (gpfq_onchip.hip), produced by launch_row_stats:
//   G      = <Xq_t, X_t>                         (f64)
//   rden   = 1 / (f32-rounded ||Xq_t||)^2        (0 for rows that take rule (i))
//   cbound = 2^-23 * sum_i |Xq_ti X_ti| * rden   bound on the f32 product roundings, per unit |w|
//   cabs   = 2^-149 * sum_i |Xq_ti| * rden       the same for products that round in the subnormal range
struct RowStats {
    double G, rden, cbound, cabs;
};

struct OnchipArgs {
    const float *X, *Xq;
    int64_t ld;
    const float *nrm32;
    const RowStats *stats = nullptr;
    unsigned long long *fallback_count = nullptr;
    int mode = 0;          // 0 = exact flow, 1 = certified (needs stats)
    int lpn = 0;           // lanes per neuron of the row-group kernel: 16/32/64, 1 = wave kernel, 0 = heuristic
    int wpn = 0;           // wavefronts per neuron of the wide kernel (forced split), 0 = heuristic
    const float *Wt;
    int64_t ldw;
    AlphabetArg A;
    int64_t N, m, C;
    int8_t *qidx;
    float *Qt;
    double *resid;
    double *u_out;
    int ts_override = 0;   // tuning hooks (bench/tests); 0 = heuristic
    int nw_override = 0;
    int variant = 0;
};

struct StreamArgs {
    const float *X, *Xq;
    int64_t ld;
    const float *nrm32;
    const float *Wt;
    int64_t ldw;
    AlphabetArg A;
    int64_t N, m, C;
    int8_t *qidx;
    float *Qt;
    double *resid;
    double *u_out;      // may be NULL: then the residual lives in the workspace
    void *workspace;
    size_t workspace_bytes;
};

hipError_t launch_onchip(const OnchipArgs &a, hipStream_t stream);
bool rows_supported(const OnchipArgs &a, int lpn);
hipError_t launch_rows(const OnchipArgs &a, int lpn, hipStream_t stream);
hipError_t launch_wide(const OnchipArgs &a, int W, hipStream_t stream);

size_t stream_workspace_bytes(int64_t N, int64_t m, int64_t C, bool need_u);
hipError_t launch_stream(const StreamArgs &a, hipStream_t stream);

struct GramArgs {
    const float *X, *Xq;
    int64_t ld;
    const float *nrm32;
    float *nrm32_out = nullptr;   // non-NULL: fill the row norms from the Gram diagonal first (nrm32 == nrm32_out)
    const float *Wt;
    int64_t ldw;
    AlphabetArg A;
    int64_t N, m, C;
    int8_t *qidx;
    float *Qt;
    double *resid;          // may be NULL (skips the exact replay of the residual)
    int32_t *uncertified;   // [C]: 1 = decision chain not certified, rerun through the exact path
    void *workspace;
    double slack = 1.0;     // multiplies the error bounds (tests)
    int variant = 0;        // tuning hook: Gram tile shape
};

size_t gram_workspace_bytes(int64_t N, int64_t m, int64_t C);
hipError_t launch_gram(const GramArgs &a, hipStream_t stream);

hipError_t launch_row_stats(const float *X, const float *Xq, int64_t N, int64_t m, int64_t ld, const float *nrm32,
                            RowStats *stats, hipStream_t stream);
hipError_t launch_row_norms(const float *Xq, int64_t N, int64_t m, int64_t ld, float *nrm32, hipStream_t stream);
hipError_t launch_msq(const float *W, int64_t n, const AlphabetArg &A, float *Q, int8_t *qidx, hipStream_t stream);
hipError_t launch_assemble(const int8_t *qidx, const AlphabetArg &A, int64_t N, int64_t C, int bits, float *Q, int8_t *idxT,
                           hipStream_t stream);
hipError_t launch_pack(const int8_t *qidx, int64_t N, int64_t C, int bits, uint8_t *packed, hipStream_t stream);
size_t median_workspace_bytes();
hipError_t launch_median_abs(const float *W, int64_t n, float *out, void *workspace, hipStream_t stream);
hipError_t launch_extract_patches(const float *act, int64_t n, int64_t H, int64_t W, int64_t Cin, int64_t c,
                                  int kh, int kw, int sh, int sw, int rh, int rw, int pad_top, int pad_left,
                                  int64_t oh, int64_t ow, float *P, int64_t ldp, hipStream_t stream);

}  // namespace gpfq
