// consts.h -- sizes shared by the kernels (kernels.h) and the pure-host planner (plan.h); no HIP types here.
#pragma once
#include <stdint.h>

namespace dsh {

constexpr uint32_t kTile = 128;   // sketches per tile side in k_pair_counts
constexpr uint32_t kListCap = 512;  // capacity of a sketch's register list (entries): upper tail (emax <= 255) + lower tail (elow <= 255)
constexpr uint32_t kMaxListSide = 255;  // cap of either tail (their per-value counts are bytes)
constexpr uint32_t kMaxBuckets = 1u << 15;  // buckets of a column block's index: (position group, tail)
constexpr int kMaxPLds = 17;   // largest p whose registers fit a workgroup's LDS (k_sketch)
constexpr int kMaxPReg32 = 15;  // up to here k_sketch keeps one 32-bit word per register in LDS (ds_max_i32; 128 KiB at p = 15, a 1 024-lane workgroup per CU); packed bytes above
                                // (A/B profiles/rd6d, rd6e: p = 10 +11 %, 12-13 +18 %, 14 +4 %; 15 would be -32 %: one workgroup per CU)
constexpr int kMaxPCompare = 24;  // the compare path takes every p the sketches can have
constexpr int kMaxP = 24;      // largest p for sketching / cardinalities / up- and download (positions are 24-bit)
constexpr uint32_t kSketchSub = 8192;  // bases per sub-chunk (256 threads x 32 start positions)

}  // namespace dsh
