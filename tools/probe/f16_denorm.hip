// probe: do v_cvt / v_mfma_f32_32x32x16_f16 keep fp16 denormals?  (hipcc --offload-arch=gfx950 f16_denorm.hip -o f16_denorm)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float a, float* out, unsigned* bits) {
  const _Float16 ha = (_Float16)a;                 // conversion of a value in the fp16 subnormal range
  h8 A, B;
  for (int i = 0; i < 8; ++i) { A[i] = ha; B[i] = (_Float16)1.0f; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; unsigned short u; __builtin_memcpy(&u, &ha, 2); bits[0] = u; out[1] = (float)ha; }
}
int main() {
  float* d; unsigned* b; hipMalloc(&d, 8); hipMalloc(&b, 4);
  for (float a : {3.814697265625e-06f /*2^-18*/, 5.9604645e-08f /*2^-24*/, 1e-5f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, d, b);
    float h[2]; unsigned hb; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost); hipMemcpy(&hb, b, 4, hipMemcpyDeviceToHost);
    printf("a=%g  fp16 bits=0x%04x back=%g  mfma(sum of 16 a*1)=%g  expected=%g\n", a, hb, h[1], h[0], 16.0 * h[1]);
  }
  return 0;
}
