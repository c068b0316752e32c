// Cycles per MFMA instruction on one SIMD (one wave per SIMD, back-to-back independent accumulators) for the f16 shapes a
// head dimension of 8 could use: 32x32x16 (the shape the kernels use), 16x16x32, the CDNA3-era 32x32x8 and 16x16x16, 4x4x4.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_rate tools/mfma_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
#define REP 64
template <int KIND>
__global__ void k(float* out, long long* cyc) {
  f16x8 a8, b8; f16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(i * 0.5f); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  f32x4 d0 = {0}, d1 = {0}, d2 = {0}, d3 = {0};
  long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int r = 0; r < REP; ++r) {
    if constexpr (KIND == 0) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, c3, 0, 0, 0);
    } else if constexpr (KIND == 1) {
      d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, d3, 0, 0, 0);
    } else if constexpr (KIND == 2) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, c3, 0, 0, 0);
    } else if constexpr (KIND == 3) {
      d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, d3, 0, 0, 0);
    } else {
      d0 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, d1, 0, 0, 0);
      d2 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, d3, 0, 0, 0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  for (int i = 0; i < 4; ++i) s += d0[i] + d1[i] + d2[i] + d3[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; long long* cyc; long long h;
  (void)hipMalloc(&out, 1 << 20); (void)hipMalloc(&cyc, 8);
  const char* names[5] = {"32x32x16_f16", "16x16x32_f16", "32x32x8f16", "16x16x16f16", "4x4x4f16"};
  const double flops[5] = {2.0 * 32 * 32 * 16, 2.0 * 16 * 16 * 32, 2.0 * 32 * 32 * 8, 2.0 * 16 * 16 * 16, 2.0 * 4 * 4 * 4 * 16};
  for (int kind = 0; kind < 5; ++kind) {
    for (int rep = 0; rep < 2; ++rep) {
      if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, out, cyc);
      if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, out, cyc);
      if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, out, cyc);
      if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, out, cyc);
      if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, out, cyc);
      (void)hipDeviceSynchronize();
    }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (4.0 * REP);
    printf("%-14s %6.1f counter ticks per MFMA  (%.0f flop per instruction, %.0f flop per tick)\n", names[kind], per, flops[kind], flops[kind] / per);
  }
  return 0;
}
