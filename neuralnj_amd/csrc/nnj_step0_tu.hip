// nnj_step0_tu.hip -- the two all-pairs kernels of step 0, k_pair_alpha<1, 8> and k_pair_score<1, 8> (nnj_scorer.hpp), in a
// translation unit of their own, so that they can be compiled with the backend's max-ilp scheduling strategy (-mllvm
// -amdgpu-sched-strategy=max-ilp, neuralnj_amd/build.py).  Round 5 measured that strategy over the whole library
// (profiles/r05/ab_sched_strategies.txt, ab_step0_tu.txt): it orders the MFMA blocks of these two kernels better (14.85 -> 13.8 and
// 33.1 -> 32.7 ms per 256-tree rollout; the same instructions, bit-identical results) and makes k_tok1p and the group kernels of
// the NJ step spill -- a per-kernel choice, which the toolchain offers per translation unit only.  The kernel source is the one
// of nnj_scorer.hpp, included inside an anonymous namespace (internal linkage: nothing here collides with the same templates
// in nnj_api.hip; NNJ_STEP0_TU leaves the file's other kernels out); the launchers take the two parameter structs as bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <type_traits>

#define NNJ_STEP0_TU 1
namespace {
#include "nnj_scorer.hpp"
}

hipError_t nnj_launch_pair_alpha_1_8(unsigned grid, size_t lds, hipStream_t st, const void* rowset, const void* scorerw,
                                     const int* ij_prev, float* alpha_part, int mode, int n, int C, int npairs, int ppad,
                                     int cs, int nsc, int npg, int B) {
  RowSet rs;
  ScorerW sw;
  memcpy(&rs, rowset, sizeof(rs));
  memcpy(&sw, scorerw, sizeof(sw));
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_alpha<1, 8>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_pair_alpha<1, 8>), dim3(grid), dim3(512), lds, st, rs, sw, ij_prev, alpha_part, mode, n, C, npairs,
                     ppad, cs, nsc, npg, B);
  return hipGetLastError();
}

hipError_t nnj_launch_pair_score_1_8(unsigned grid, size_t lds, hipStream_t st, const void* rowset, const void* scorerw,
                                     const int* ij_prev, const float* alpha, const uint8_t* mask, float* score_part, int mode,
                                     int n, int C, int npairs, int ppad, int cs, int has_ctx, int nsc, int npg, int B) {
  RowSet rs;
  ScorerW sw;
  memcpy(&rs, rowset, sizeof(rs));
  memcpy(&sw, scorerw, sizeof(sw));
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_pair_score<1, 8>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL((k_pair_score<1, 8>), dim3(grid), dim3(512), lds, st, rs, sw, ij_prev, alpha, mask, score_part, mode, n, C,
                     npairs, ppad, cs, has_ctx, nsc, npg, B);
  return hipGetLastError();
}
