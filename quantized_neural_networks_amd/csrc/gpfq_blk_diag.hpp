// Diagnostic switches of gpfq_blk.hip -- included ONLY by diagnostic builds (-DGPFQ_BLK_DIAG; never the shipped library):
//   -DGPFQ_BLK_STAMPS    in-kernel phase stamps (s_memtime) of two sweep wavefronts and the decision wavefront of workgroup 0, accumulated
//                        per slot and left in the unused row-statistics area behind the call's counter block (tools/pipe_probe.py prints them)
//   -DGPFQ_BLK_NO_MFMA   phase D (the block dot products) on the vector unit in every shape: the A/B of round 4's matrix form
//   -DGPFQ_BLK_NO_FUSED  matrix-unit phase D as a phase of its own behind the updates: round 4's first form
// The marginal-cost experiments of rounds 4-5 (every matrix / vector / LDS / DMA instruction issued twice, the barrier removed, pair splits
// from the environment, the flush forced to one side) were removed in round 6; their numbers are in profiles/r04 and profiles/r05, their
// code in the history (commit 1050215 and before).
#pragma once

#ifdef GPFQ_BLK_NO_MFMA
constexpr bool kNoMfmaD = true;
#else
constexpr bool kNoMfmaD = false;
#endif
#ifdef GPFQ_BLK_NO_FUSED
constexpr bool kNoFused = true;
#else
constexpr bool kNoFused = false;
#endif

#ifdef GPFQ_BLK_STAMPS
#define STAMP(var) do { __builtin_amdgcn_sched_barrier(0); var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } while (0)
#define STAMP_DO(...) __VA_ARGS__
#else
#define STAMP(var) do { } while (0)
#define STAMP_DO(...)
#endif
