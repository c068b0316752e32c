// Sustained f16 MFMA rate of the WHOLE chip on random operands by instruction shape: 32x32x16 against 16x16x32 (MI355X_MICROARCH.md
// "DVFS give-back" item 7: the chip lowers its clock under matrix load, and the clock it holds depends on the shape).  Every CU
// runs WAVES waves (one or two per SIMD) of back-to-back MFMAs on rotating random operands for ~1 s; reported: TFLOP/s by wall
// clock and the in-kernel clock (s_memtime ticks per s_memrealtime 100 MHz tick).  DUTY < 100 inserts s_sleep so that the matrix
// pipe is busy only part of the time (the row-attention kernels keep it ~57 % busy).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_power tools/mfma_power.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int SLEEP, int MODE = 1>
__global__ __launch_bounds__(512) void k(const f16x8* __restrict__ ops, float* out, unsigned long long* clk, int iters) {
  f16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = ops[(threadIdx.x * 8 + i) & 4095]; b[i] = ops[(threadIdx.x * 8 + 4 + i + blockIdx.x) & 4095]; }
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  f32x4 d[16];
  for (int i = 0; i < 16; ++i) d[i] = (f32x4){0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) c[i] = MODE == 2 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], c[i], 0, 0, 0) : MODE == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i + 2 * r + 1) & 3], c[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[r], c[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) d[i] = MODE == 2 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0], b[0], d[i], 0, 0, 0) : MODE == 0 ? __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 3], b[(i + 2 * r + 1) & 3], d[i], 0, 0, 0) : __builtin_amdgcn_mfma_f32_16x16x32_f16(a[(i + r) & 3], b[(i >> 2) ^ r], d[i], 0, 0, 0);
    }
    if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += c[i][j];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) s += d[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND, int SLEEP, int MODE = 1>
void run(const char* name, int waves, const f16x8* ops, float* out, unsigned long long* clk, int zero) {
  const int nblk = 256, iters = 1 << 17;                      // 16 MFMAs of 32x32x16 or 32 of 16x16x32 per iteration: 2^19 flop x ...
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL((k<KIND, SLEEP, MODE>), dim3(nblk), dim3(64 * waves), 0, 0, ops, out, clk, iters);
    (void)hipDeviceSynchronize();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    unsigned long long h[2 * 256];
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int i = 0; i < nblk; ++i) ghz += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    const double flop = (double)nblk * waves * iters * 16.0 * 2.0 * 32 * 32 * 16;
    if (rep == 2)
      printf("%-14s %s mode %d waves/CU %2d sleep %2d: %7.3f s  %7.1f TFLOP/s  in-kernel clock %.3f GHz\n", name, zero ? "ZERO  " : "random", MODE, waves, SLEEP, s,
             flop / s * 1e-12, ghz / nblk);
  }
}

int main() {
  f16x8* ops; float* out; unsigned long long* clk;
  (void)hipMalloc(&ops, 4096 * sizeof(f16x8)); (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 2 * 256 * 8);
  for (int zero = 0; zero < 2; ++zero) {
    _Float16 h[4096 * 8];
    srand(7);
    for (int i = 0; i < 4096 * 8; ++i) h[i] = zero ? (_Float16)0.f : (_Float16)(((rand() & 0xffff) / 32768.0f - 1.0f) * 4.0f);
    (void)hipMemcpy(ops, h, sizeof(h), hipMemcpyHostToDevice);
    run<0, 0>("32x32x16_f16", 4, ops, out, clk, zero);
    run<1, 0>("16x16x32_f16", 4, ops, out, clk, zero);
    run<0, 0>("32x32x16_f16", 8, ops, out, clk, zero);
    run<1, 0>("16x16x32_f16", 8, ops, out, clk, zero);
    run<0, 4>("32x32x16_f16", 8, ops, out, clk, zero);
    run<1, 4>("16x16x32_f16", 8, ops, out, clk, zero);
    if (!zero) {     // mode 0: both operands change with every MFMA; 1 (above): B constant over four; 2: the SAME random operands every time
      run<0, 0, 0>("32x32x16_f16", 8, ops, out, clk, zero);
      run<0, 0, 2>("32x32x16_f16", 8, ops, out, clk, zero);
      run<1, 0, 0>("16x16x32_f16", 8, ops, out, clk, zero);
      run<1, 0, 2>("16x16x32_f16", 8, ops, out, clk, zero);
    }
  }
  return 0;
}
