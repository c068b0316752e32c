// nnj_train.hip -- libnnj_train_hip.so (include/nnj_train.h): forward and backward kernels of the operators of the
// reference's Finetune mode, gfx950.  A first, unfused path: fp32 arithmetic (contractions on the fp32 matrix pipe), one
// kernel per operator; the graph is kept by torch.autograd.Function objects on the host (neuralnj_amd/train_ops.py).
// No CPU fallback.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/nnj_train.h"

namespace {
char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define CHK_LAUNCH()                                                                              \
  do {                                                                                            \
    hipError_t e_ = hipGetLastError();                                                            \
    if (e_ != hipSuccess) return fail(-3, "%s: %s", __func__, hipGetErrorString(e_));             \
  } while (0)
inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

// ------------------------------------------------------------------ strided batched GEMM
// 128 x 64 output tile per workgroup of four waves, each wave 32 rows x 64 columns in two 32x32 accumulators of
// v_mfma_f32_32x32x2_f32 (exact fp32 products and sums, like the fmaf chain torch's fp32 GEMM is compared with); k in
// steps of 16 through LDS.  Loads walk whichever of (m, k) / (k, n) has unit stride fastest, so nn.Linear operands,
// their transposes and the head-strided attention operands are all read in contiguous pieces.  A first version of
// this kernel did the products on the vector pipe (4 x 4 outputs per thread): 85 % of a Finetune episode's GPU time.
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int TM = 128, TN = 64;
// TK = k-extent of one LDS stage: 16 (12.5 KB of LDS).  A 64-long stage for the short contractions (k = 64 of every
// nn.Linear: one barrier pair instead of four) was measured and dropped: 50 KB of LDS per workgroup and the four-row
// scatter of the k-fast operand into LDS made every shape slower (GEMM time of an episode 62 -> 88 ms).
template <int TK>
__global__ __launch_bounds__(256) void k_gemm(nnjt_gemm g) {
  __shared__ __attribute__((aligned(16))) float As[TK][TM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[TK][TN + 4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int b1 = blockIdx.z / g.nb2, b2 = blockIdx.z % g.nb2;
  const float* A = g.A + b1 * g.sAb1 + b2 * g.sAb2;
  const float* B = g.B + b1 * g.sBb1 + b2 * g.sBb2;
  float* C = g.C + b1 * g.sCb1 + b2 * g.sCb2;
  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  const bool a_kfast = g.sAk == 1, b_kfast = g.sBk == 1;
  // 16-byte loads along the unit-stride direction where the operand allows it (aligned base, other stride a multiple
  // of four floats): the products of the model are short (k = 64 for every nn.Linear), so the operand loads, not the
  // MFMAs, are most of this kernel; scalar loads otherwise
  typedef float f4 __attribute__((ext_vector_type(4)));
  const bool a_vec = ((a_kfast && g.sAm % 4 == 0) || (g.sAm == 1 && g.sAk % 4 == 0)) && ((uintptr_t)A % 16 == 0);
  const bool b_vec = ((b_kfast && g.sBn % 4 == 0) || (g.sBn == 1 && g.sBk % 4 == 0)) && ((uintptr_t)B % 16 == 0);
  for (int k0 = 0; k0 < g.K; k0 += TK) {
    if (a_vec) {
#pragma unroll
      for (int i = 0; i < TM * TK / 4 / 256; ++i) {
        const int e = tid + 256 * i;                         // TM * TK / 4 groups of four
        if (a_kfast) {                                       // four consecutive k of one row
          const int m = e / (TK / 4), k = 4 * (e % (TK / 4));
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (m0 + m < g.M && k0 + k + 3 < g.K) v = *reinterpret_cast<const f4*>(A + (int64_t)(m0 + m) * g.sAm + (k0 + k));
          else if (m0 + m < g.M)
            for (int t = 0; t < 4; ++t) if (k0 + k + t < g.K) v[t] = A[(int64_t)(m0 + m) * g.sAm + (k0 + k + t)];
          As[k][m] = v[0]; As[k + 1][m] = v[1]; As[k + 2][m] = v[2]; As[k + 3][m] = v[3];
        } else {                                             // four consecutive rows of one k
          const int m = 4 * (e % (TM / 4)), k = e / (TM / 4);
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (k0 + k < g.K && m0 + m + 3 < g.M) v = *reinterpret_cast<const f4*>(A + (int64_t)(k0 + k) * g.sAk + (m0 + m));
          else if (k0 + k < g.K)
            for (int t = 0; t < 4; ++t) if (m0 + m + t < g.M) v[t] = A[(int64_t)(k0 + k) * g.sAk + (m0 + m + t)];
          *reinterpret_cast<f4*>(&As[k][m]) = v;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TM * TK / 256; ++i) {
        const int e = tid + 256 * i;
        const int m = a_kfast ? e / TK : e % TM, k = a_kfast ? e % TK : e / TM;
        const bool ok = m0 + m < g.M && k0 + k < g.K;
        As[k][m] = ok ? A[(int64_t)(m0 + m) * g.sAm + (int64_t)(k0 + k) * g.sAk] : 0.f;
      }
    }
    if (b_vec) {
#pragma unroll
      for (int i = 0; i < TN * TK / 4 / 256; ++i) {
        const int e = tid + 256 * i;                         // TN * TK / 4 groups of four
        if (b_kfast) {
          const int n = e / (TK / 4), k = 4 * (e % (TK / 4));
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (n0 + n < g.N && k0 + k + 3 < g.K) v = *reinterpret_cast<const f4*>(B + (int64_t)(n0 + n) * g.sBn + (k0 + k));
          else if (n0 + n < g.N)
            for (int t = 0; t < 4; ++t) if (k0 + k + t < g.K) v[t] = B[(int64_t)(n0 + n) * g.sBn + (k0 + k + t)];
          Bs[k][n] = v[0]; Bs[k + 1][n] = v[1]; Bs[k + 2][n] = v[2]; Bs[k + 3][n] = v[3];
        } else {
          const int n = 4 * (e % (TN / 4)), k = e / (TN / 4);
          f4 v = {0.f, 0.f, 0.f, 0.f};
          if (k0 + k < g.K && n0 + n + 3 < g.N) v = *reinterpret_cast<const f4*>(B + (int64_t)(k0 + k) * g.sBk + (n0 + n));
          else if (k0 + k < g.K)
            for (int t = 0; t < 4; ++t) if (n0 + n + t < g.N) v[t] = B[(int64_t)(k0 + k) * g.sBk + (n0 + n + t)];
          *reinterpret_cast<f4*>(&Bs[k][n]) = v;
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < TN * TK / 256; ++i) {
        const int e = tid + 256 * i;
        const int n = b_kfast ? e / TK : e % TN, k = b_kfast ? e % TK : e / TN;
        const bool ok = n0 + n < g.N && k0 + k < g.K;
        Bs[k][n] = ok ? B[(int64_t)(k0 + k) * g.sBk + (int64_t)(n0 + n) * g.sBn] : 0.f;
      }
    }
    __syncthreads();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int kk = 0; kk < TK; kk += 2) {                     // (rows of the stage beyond K hold zeros)
      const float a = As[kk + kh][32 * wave + l31];          // A[m = lane & 31][k = lane >> 5]
      const float bl = Bs[kk + kh][l31], bh = Bs[kk + kh][32 + l31];
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bl, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bh, acc[1], 0, 0, 0);
    }
#endif
    __syncthreads();
  }
  // C/D layout: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * kh, n = n0 + 32 * t + l31;
      if (m < g.M && n < g.N) {
        float* c = C + (int64_t)m * g.sCm + (int64_t)n * g.sCn;
        float v = g.alpha * acc[t][r];
        if (g.bias) v += g.bias[n];
        *c = g.beta == 0.f ? v : v + g.beta * *c;
      }
    }
}

// ------------------------------------------------------------------ weight gradient of a 64 -> 64 nn.Linear
// dW[m][n] = sum_tokens dy[token][m] x[token][n]: a 64 x 64 output and a contraction over up to millions of tokens --
// in a batch of Finetune episodes 45 % of the GEMM time went here, through the general kernel's LDS stages with one
// workgroup per CU.  Both operands lie token-major with 64 contiguous features, which is exactly what
// v_mfma_f32_32x32x2_f32 wants lane by lane (lane & 31 = feature, lane >> 5 = which of the step's two tokens): every
// lane loads its four operand values (two feature halves of dy, two of x) straight from global memory, 128-byte
// segments per half wave, no LDS and no barrier in the loop.  A workgroup owns `per_block` tokens, a quarter per wave;
// the four waves' tiles are added in LDS in wave order (deterministic) and written as one part; nnjt_sum_rows adds the
// parts.
__global__ __launch_bounds__(256) void k_wgrad64(const float* __restrict__ dy, const float* __restrict__ x,
                                                 float* __restrict__ parts, int64_t rows, int64_t per_block) {
  __shared__ float red[64 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int64_t b0 = (int64_t)blockIdx.x * per_block;
  const int64_t b1 = b0 + per_block < rows ? b0 + per_block : rows;
  const int64_t per_wave = ((per_block / 4) + 1) & ~(int64_t)1;          // even: a step takes two tokens
  const int64_t k0 = b0 + wave * per_wave;
  const int64_t k1 = k0 + per_wave < b1 ? k0 + per_wave : b1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int U = 8;                                       // token pairs in flight per lane
  for (int64_t t = k0; t < k1; t += 2 * U) {
    float a0[U], a1[U], c0[U], c1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t tok = t + 2 * u + kh;
      const bool ok = tok < k1;
      const float* dp = dy + (ok ? tok : k0) * 64 + l31;
      const float* xp = x + (ok ? tok : k0) * 64 + l31;
      a0[u] = ok ? dp[0] : 0.f;
      a1[u] = ok ? dp[32] : 0.f;
      c0[u] = ok ? xp[0] : 0.f;
      c1[u] = ok ? xp[32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], c0[u], acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], c1[u], acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], c0[u], acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], c1[u], acc[1][1], 0, 0, 0);
    }
  }
#endif
  // C/D layout: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh, n = 32 * j + l31;
            if (w == 0) red[m * 64 + n] = acc[i][j][r];
            else red[m * 64 + n] += acc[i][j][r];
          }
    }
    __syncthreads();
  }
  float* out = parts + (int64_t)blockIdx.x * 4096;
  for (int e = tid; e < 4096; e += 256) out[e] = red[e];
}

// ------------------------------------------------------------------ skinny products: C[M, N] = alpha A[M, K] B[K, N]
// M, K <= 64 and N = sites x features (65,536 at 1024 sites): the pair scorer's x_g = alpha_rows x state (model.py:148)
// and its gradient with respect to the state.  Bound by streaming B in and C out; the general kernel's 128 x 64 tile is
// mostly empty here (1.2 TB/s).  A (a few thousand floats, any strides) sits in LDS as [k][m]; B goes from global memory
// straight into the MFMA lanes (lane & 31 = column: 128-byte segments; lane >> 5 = which of the step's two k); a wave
// owns 64 columns, a workgroup 256.  MT = 1 when M <= 32.
template <int MT>
__global__ __launch_bounds__(256) void k_skinny(const float* __restrict__ A, int64_t sAm, int64_t sAk, int64_t bsA,
                                                const float* __restrict__ B, int64_t bsB, float* __restrict__ C,
                                                int64_t bsC, int M, int K, int64_t N, float alpha) {
  __shared__ float As[64][64];                               // [k][m], zero beyond (K, M)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, kh = lane >> 5;
  const int64_t b = blockIdx.y;
  A += b * bsA;
  B += b * bsB;
  C += b * bsC;
  for (int e = tid; e < 64 * 64; e += 256) {
    const int k = e >> 6, m = e & 63;
    As[k][m] = (k < K && m < M) ? A[(int64_t)m * sAm + (int64_t)k * sAk] : 0.f;
  }
  __syncthreads();
  const int64_t col0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  if (col0 >= N) return;
  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int U = 8;                                       // k pairs in flight per lane
  for (int k0 = 0; k0 < K; k0 += 2 * U) {
    float b0[U], b1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * u + kh;
      const bool ok = k < K;
      const float* bp = B + (int64_t)(ok ? k : 0) * N + col0 + l31;
      b0[u] = ok ? bp[0] : 0.f;
      b1[u] = ok ? bp[32] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 2 * u + kh;                         // (< 64 + 16: As rows beyond K are never read: k < K or b = 0)
      const int kc = k < 64 ? k : 63;
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const float a = As[kc][32 * i + l31];
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0[u], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1[u], acc[i][1], 0, 0, 0);
      }
    }
  }
#endif
  // C/D layout: register r of a lane = row (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column lane & 31
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m < M) C[(int64_t)m * N + col0 + 32 * j + l31] = alpha * acc[i][j][r];
      }
}

// ------------------------------------------------------------------ bias, column sums
__global__ void k_add_bias(float* y, const float* __restrict__ bias, int64_t n, int cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += bias[i % cols];
}
// out[c] += sum_r x[r, c] (cols <= 256): a workgroup owns `rows_per_block` rows; its 256 threads are 256 / cols row groups
// x cols columns, so every iteration reads 256 consecutive floats; the groups meet in LDS and the workgroup adds ONE
// atomic per column (the first version -- 64 rows per workgroup, thread = column -- spent 0.5 ms of atomics on the
// same 64 addresses for the 1.25 M token rows of the encoder's bias gradients).
__global__ __launch_bounds__(256) void k_colsum(const float* __restrict__ x, float* out, int64_t rows, int cols,
                                                int64_t rows_per_block) {
  __shared__ float part[256];
  const int G = 256 / cols;
  const int c = threadIdx.x % cols, rg = threadIdx.x / cols;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float s = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (rg < G) {
    int64_t r = r0 + rg;
    for (; r + 3 * G < r1; r += 4 * G) {                     // four loads in flight per lane
      s += x[r * cols + c];
      s1 += x[(r + G) * cols + c];
      s2 += x[(r + 2 * G) * cols + c];
      s3 += x[(r + 3 * G) * cols + c];
    }
    for (; r < r1; r += G) s += x[r * cols + c];
    s = (s + s1) + (s2 + s3);
  }
  part[threadIdx.x] = s;
  __syncthreads();
  if ((int)threadIdx.x < cols) {
    float t = 0.f;
    for (int g2 = 0; g2 < G; ++g2) t += part[g2 * cols + threadIdx.x];
    atomicAdd(out + threadIdx.x, t);
  }
}

// out[c] = sum_r x[r, c] for any number of columns: the second step of a contraction cut into pieces.  A workgroup owns
// 64 columns; its four waves each add every fourth row (four loads in flight per lane), then the four partial sums
// meet in LDS in a fixed order -- deterministic.  (First version: one thread per column walking all rows alone, 40 us
// for the 256 pieces of a weight gradient: 22 % of a Finetune episode's GPU time.)
__global__ __launch_bounds__(256) void k_sum_rows(const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                  int64_t cols) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int64_t r = rg;
    for (; r + 12 < rows; r += 16) {
      s0 += x[r * cols + c];
      s1 += x[(r + 4) * cols + c];
      s2 += x[(r + 8) * cols + c];
      s3 += x[(r + 12) * cols + c];
    }
    for (; r < rows; r += 4) s0 += x[r * cols + c];
  }
  part[rg][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rg == 0 && c < cols) out[c] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// ------------------------------------------------------------------ LayerNorm over D <= 64 features: one wavefront per row
// (lane = feature; the lanes beyond a narrower model's width add nothing to the sums)
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, float* __restrict__ y,
                                                float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const bool in = lane < D;
  const float inv_d = 1.0f / (float)D;
  const float v = in ? x[r * D + lane] : 0.f;
  float s = v;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float mu = s * inv_d;
  const float d = in ? v - mu : 0.f;
  float q = d * d;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) q += __shfl_xor(q, o);
  const float rs = rsqrtf(q * inv_d + 1e-5f);
  if (in) y[r * D + lane] = d * rs * gamma[lane] + beta[lane];
  if (lane == 0) { mean[r] = mu; rstd[r] = rs; }
}
// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma; dgamma += sum dy * xhat; dbeta += sum dy.
// A workgroup walks 64 rows per wave and adds its partial sums with one atomic per feature and wave.
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x,
                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, float* __restrict__ dx,
                                                float* dgamma, float* dbeta, int64_t rows, int D) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool in = lane < D;
  const float inv_d = 1.0f / (float)D;
  const float gm = in ? gamma[lane] : 0.f;
  float ag = 0.f, ab = 0.f;
  const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
  for (int64_t r = r0; r < r0 + 64 && r < rows; ++r) {
    const float xh = in ? (x[r * D + lane] - mean[r]) * rstd[r] : 0.f;
    const float d = in ? dy[r * D + lane] : 0.f;
    const float g = d * gm;
    float s1 = g, s2 = g * xh;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (in) dx[r * D + lane] = rstd[r] * (g - s1 * inv_d - xh * s2 * inv_d);
    ag += d * xh;
    ab += d;
  }
  if (in) {
    atomicAdd(dgamma + lane, ag);
    atomicAdd(dbeta + lane, ab);
  }
}

// ------------------------------------------------------------------ elementwise
__global__ void k_gelu_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = x[i]; y[i] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }
}
__global__ void k_gelu_bwd(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dx, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float v = x[i];
    const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * v * v);
    dx[i] = dy[i] * (cdf + v * pdf);
  }
}
__device__ __forceinline__ float sigm(float t) { return 1.0f / (1.0f + expf(-t)); }
__global__ void k_gate_fwd(const float* __restrict__ h, const float* __restrict__ a, const float* __restrict__ b,
                           float* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float z = sigm(h[i]); out[i] = z * a[i] + (1.0f - z) * b[i]; }
}
__global__ void k_gate_bwd(const float* __restrict__ dout, const float* __restrict__ h, const float* __restrict__ a,
                           const float* __restrict__ b, float* __restrict__ dh, float* __restrict__ da,
                           float* __restrict__ db, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const float z = sigm(h[i]), d = dout[i];
    dh[i] = d * (a[i] - b[i]) * z * (1.0f - z);
    da[i] = d * z;
    db[i] = d * (1.0f - z);
  }
}
__global__ void k_axpby(float a, const float* x, float b, const float* y, float* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a * x[i] + (y ? b * y[i] : 0.f);
}
__global__ void k_rowscale(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out,
                           int64_t n, int cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] * s[i / cols];
}
// nn.Dropout in training mode.  The keep decision of element i is a pure function of (seed, offset + i): a SplitMix64
// finaliser of the counter, its top 24 bits against p -- no generator state on the device, any launch shape gives the
// same mask.  y = x * keep / (1 - p); the byte mask is what the backward needs.
__device__ __forceinline__ uint32_t mix24(uint64_t seed, uint64_t ctr) {
  uint64_t z = seed + (ctr + 1) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 40);
}
__global__ void k_dropout_fwd(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ keep, int64_t n,
                              uint32_t thresh, float scale, uint64_t seed, uint64_t offset) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const bool k = mix24(seed, offset + (uint64_t)i) >= thresh;
  keep[i] = k ? 1 : 0;
  y[i] = k ? x[i] * scale : 0.f;
}
__global__ void k_dropout_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ keep, float* __restrict__ dx,
                              int64_t n, float scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dx[i] = keep[i] ? dy[i] * scale : 0.f;
}
__global__ void k_fill_where(float* x, const uint8_t* __restrict__ sel, float value, int64_t n, int inner, int sel_rows,
                             int cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t r = i / cols;
  const int c = (int)(i % cols);
  if (sel[((r / inner) % sel_rows) * cols + c]) x[i] = value;
}

// ------------------------------------------------------------------ softmax over the last dim: one wavefront per row
__global__ __launch_bounds__(256) void k_softmax_fwd(const float* __restrict__ x, const uint8_t* __restrict__ keep,
                                                     float* __restrict__ y, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * cols;
  const uint8_t* kr = keep ? keep + r * cols : nullptr;
  float m = -INFINITY;
  for (int c = lane; c < cols; c += 64)
    if (!kr || kr[c]) m = fmaxf(m, xr[c]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.f;
  for (int c = lane; c < cols; c += 64)
    if (!kr || kr[c]) s += expf(xr[c] - m);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  const float inv = 1.0f / s;
  for (int c = lane; c < cols; c += 64) y[r * cols + c] = (!kr || kr[c]) ? expf(xr[c] - m) * inv : 0.f;
}
__global__ __launch_bounds__(256) void k_softmax_bwd(const float* __restrict__ dy, const float* __restrict__ y,
                                                     float* __restrict__ dx, int64_t rows, int cols) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < cols; c += 64) s += y[r * cols + c] * dy[r * cols + c];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  for (int c = lane; c < cols; c += 64) dx[r * cols + c] = y[r * cols + c] * (dy[r * cols + c] - s);
}

// ------------------------------------------------------------------ row gathers
__global__ void k_gather_rows(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ out,
                              int n, int p, int64_t width) {
  const int b = blockIdx.z, q = blockIdx.y;
  const int64_t row = idx[(int64_t)b * p + q];
  const float* s = src + ((int64_t)b * n + row) * width;
  float* o = out + ((int64_t)b * p + q) * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (int64_t)gridDim.x * blockDim.x) o[i] = s[i];
}
__global__ void k_scatter_rows_add(const float* __restrict__ dout, const int64_t* __restrict__ idx, float* dsrc, int n,
                                   int p, int64_t width) {
  const int b = blockIdx.z, q = blockIdx.y;
  const int64_t row = idx[(int64_t)b * p + q];
  float* s = dsrc + ((int64_t)b * n + row) * width;
  const float* o = dout + ((int64_t)b * p + q) * width;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < width; i += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(s + i, o[i]);
}

// ------------------------------------------------------------------ 5-d permuted copy
struct Perm5 { int64_t d[5], s[5]; };
__global__ void k_permute5(const float* in, float* out, Perm5 p, int64_t n, int inverse) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int64_t rem = i, off = 0;
#pragma unroll
  for (int k = 4; k >= 0; --k) { off += (rem % p.d[k]) * p.s[k]; rem /= p.d[k]; }
  if (inverse) const_cast<float*>(in)[off] = out[i];
  else out[i] = in[off];
}
}  // namespace

extern "C" {
int nnjt_abi_version(void) { return 2; }
const char* nnjt_last_error(void) { return g_err; }

int nnjt_gemm_run(const nnjt_gemm* g, void* stream) {
  if (!g || !g->A || !g->B || !g->C) return fail(-1, "nnjt_gemm_run: null operand");
  if (g->M <= 0 || g->N <= 0 || g->K <= 0 || g->nb1 <= 0 || g->nb2 <= 0) return fail(-1, "nnjt_gemm_run: empty shape");
  const int64_t nb = (int64_t)g->nb1 * g->nb2;
  if (nb > 65535) return fail(-1, "nnjt_gemm_run: more than 65535 batch entries (%lld)", (long long)nb);
  const dim3 grid((g->N + TN - 1) / TN, (g->M + TM - 1) / TM, (unsigned)nb);
  hipLaunchKernelGGL(k_gemm<16>, grid, dim3(256), 0, static_cast<hipStream_t>(stream), *g);
  CHK_LAUNCH();
  return 0;
}
int nnjt_skinny_gemm(const float* A, int64_t sAm, int64_t sAk, int64_t bsA, const float* B, int64_t bsB, float* C,
                     int64_t bsC, int32_t nb, int32_t M, int32_t K, int64_t N, float alpha, void* stream) {
  if (!A || !B || !C) return fail(-1, "nnjt_skinny_gemm: null");
  if (nb <= 0 || nb > 65535 || M <= 0 || M > 64 || K <= 0 || K > 64 || N <= 0 || N % 64 != 0)
    return fail(-1, "nnjt_skinny_gemm: needs 1 <= M, K <= 64 and N a multiple of 64 (M %d, K %d, N %lld)", M, K, (long long)N);
  const dim3 grid((unsigned)((N + 255) / 256), (unsigned)nb);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (M <= 32) hipLaunchKernelGGL(k_skinny<1>, grid, dim3(256), 0, st, A, sAm, sAk, bsA, B, bsB, C, bsC, M, K, N, alpha);
  else hipLaunchKernelGGL(k_skinny<2>, grid, dim3(256), 0, st, A, sAm, sAk, bsA, B, bsB, C, bsC, M, K, N, alpha);
  CHK_LAUNCH();
  return 0;
}
int nnjt_wgrad64(const float* dy, const float* x, float* parts, int64_t rows, int64_t per_block, void* stream) {
  if (!dy || !x || !parts) return fail(-1, "nnjt_wgrad64: null");
  if (rows <= 0 || per_block < 8 || per_block % 8 != 0) return fail(-1, "nnjt_wgrad64: per_block must be a positive multiple of 8");
  const int64_t nparts = (rows + per_block - 1) / per_block;
  if (nparts > 65535 * 16) return fail(-1, "nnjt_wgrad64: too many parts");
  hipLaunchKernelGGL(k_wgrad64, dim3((unsigned)nparts), dim3(256), 0, static_cast<hipStream_t>(stream), dy, x, parts, rows,
                     per_block);
  CHK_LAUNCH();
  return 0;
}
int nnjt_add_bias(float* y, const float* bias, int64_t rows, int32_t cols, void* stream) {
  if (!y || !bias) return fail(-1, "nnjt_add_bias: null");
  const int64_t n = rows * cols;
  hipLaunchKernelGGL(k_add_bias, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), y, bias, n, cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_colsum(const float* x, float* out, int64_t rows, int32_t cols, void* stream) {
  if (!x || !out || cols <= 0 || cols > 256) return fail(-1, "nnjt_colsum: null, or not 1..256 columns");
  if (rows <= 0) return 0;
  const int G = 256 / cols;
  int64_t per = (rows + 1023) / 1024;                       // at most ~1024 workgroups (64 K atomics on the largest input),
  if (per < 16 * G) per = 16 * G;                           // at least 16 rows per row group
  per = (per + G - 1) / G * G;
  hipLaunchKernelGGL(k_colsum, dim3((unsigned)((rows + per - 1) / per)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x, out, rows, cols, per);
  CHK_LAUNCH();
  return 0;
}
int nnjt_sum_rows(const float* x, float* out, int64_t rows, int64_t cols, void* stream) {
  if (!x || !out || rows <= 0 || cols <= 0) return fail(-1, "nnjt_sum_rows: bad argument");
  hipLaunchKernelGGL(k_sum_rows, dim3(blocks_for(cols, 64)), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, rows, cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int64_t rows, int32_t cols, void* stream) {
  if (cols < 1 || cols > 64) return fail(-2, "nnjt_layernorm: 1 .. 64 features (one lane per feature), got %d", cols);
  hipLaunchKernelGGL(k_ln_fwd, dim3(blocks_for(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma, beta, y,
                     mean, rstd, rows, (int)cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t cols, void* stream) {
  if (cols < 1 || cols > 64) return fail(-2, "nnjt_layernorm: 1 .. 64 features, got %d", cols);
  hipLaunchKernelGGL(k_ln_bwd, dim3(blocks_for(rows, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, x, gamma,
                     mean, rstd, dx, dgamma, dbeta, rows, (int)cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
  hipLaunchKernelGGL(k_gelu_fwd, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, n);
  CHK_LAUNCH();
  return 0;
}
int nnjt_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream) {
  hipLaunchKernelGGL(k_gelu_bwd, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, x, dx, n);
  CHK_LAUNCH();
  return 0;
}
int nnjt_gate_fwd(const float* h, const float* a, const float* b, float* out, int64_t n, void* stream) {
  hipLaunchKernelGGL(k_gate_fwd, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), h, a, b, out, n);
  CHK_LAUNCH();
  return 0;
}
int nnjt_gate_bwd(const float* dout, const float* h, const float* a, const float* b, float* dh, float* da, float* db,
                  int64_t n, void* stream) {
  hipLaunchKernelGGL(k_gate_bwd, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dout, h, a, b, dh,
                     da, db, n);
  CHK_LAUNCH();
  return 0;
}
int nnjt_softmax_fwd(const float* x, const uint8_t* keep, float* y, int64_t rows, int32_t cols, void* stream) {
  hipLaunchKernelGGL(k_softmax_fwd, dim3(blocks_for(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, keep, y,
                     rows, cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_softmax_bwd(const float* dy, const float* y, float* dx, int64_t rows, int32_t cols, void* stream) {
  hipLaunchKernelGGL(k_softmax_bwd, dim3(blocks_for(rows, 4)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, y, dx,
                     rows, cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_axpby(float a, const float* x, float b, const float* y, float* out, int64_t n, void* stream) {
  if (!x || !out) return fail(-1, "nnjt_axpby: null");
  hipLaunchKernelGGL(k_axpby, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a, x, b, y, out, n);
  CHK_LAUNCH();
  return 0;
}
int nnjt_rowscale(const float* x, const float* s, float* out, int64_t rows, int32_t cols, void* stream) {
  const int64_t n = rows * cols;
  hipLaunchKernelGGL(k_rowscale, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, s, out, n, cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_dropout_fwd(const float* x, float* y, uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset,
                     void* stream) {
  if (!x || !y || !keep) return fail(-1, "nnjt_dropout_fwd: null");
  if (!(p >= 0.f && p < 1.f)) return fail(-1, "nnjt_dropout_fwd: p = %g outside [0, 1)", (double)p);
  if (n <= 0) return 0;
  const uint32_t thresh = (uint32_t)((double)p * 16777216.0);            // drop when the 24-bit draw is below p * 2^24
  hipLaunchKernelGGL(k_dropout_fwd, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, keep, n,
                     thresh, 1.0f / (1.0f - p), seed, offset);
  CHK_LAUNCH();
  return 0;
}
int nnjt_dropout_bwd(const float* dy, const uint8_t* keep, float* dx, int64_t n, float p, void* stream) {
  if (!dy || !dx || !keep) return fail(-1, "nnjt_dropout_bwd: null");
  if (!(p >= 0.f && p < 1.f)) return fail(-1, "nnjt_dropout_bwd: p = %g outside [0, 1)", (double)p);
  if (n <= 0) return 0;
  hipLaunchKernelGGL(k_dropout_bwd, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dy, keep, dx, n,
                     1.0f / (1.0f - p));
  CHK_LAUNCH();
  return 0;
}
int nnjt_fill_where(float* x, const uint8_t* sel, float value, int64_t rows, int32_t inner, int32_t sel_rows,
                    int32_t cols, void* stream) {
  const int64_t n = rows * cols;
  hipLaunchKernelGGL(k_fill_where, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x, sel, value, n,
                     inner, sel_rows, cols);
  CHK_LAUNCH();
  return 0;
}
int nnjt_gather_rows(const float* src, const int64_t* idx, float* out, int32_t B, int32_t n, int32_t p, int64_t width,
                     void* stream) {
  if (p > 65535 || B > 65535) return fail(-1, "nnjt_gather_rows: p and B up to 65535");
  const unsigned gx = (unsigned)(width >= 256 * 64 ? 64 : (width + 255) / 256);
  hipLaunchKernelGGL(k_gather_rows, dim3(gx, (unsigned)p, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                     idx, out, n, p, width);
  CHK_LAUNCH();
  return 0;
}
int nnjt_scatter_rows_add(const float* dout, const int64_t* idx, float* dsrc, int32_t B, int32_t n, int32_t p,
                          int64_t width, void* stream) {
  if (p > 65535 || B > 65535) return fail(-1, "nnjt_scatter_rows_add: p and B up to 65535");
  const unsigned gx = (unsigned)(width >= 256 * 64 ? 64 : (width + 255) / 256);
  hipLaunchKernelGGL(k_scatter_rows_add, dim3(gx, (unsigned)p, (unsigned)B), dim3(256), 0, static_cast<hipStream_t>(stream),
                     dout, idx, dsrc, n, p, width);
  CHK_LAUNCH();
  return 0;
}
int nnjt_permute5(const float* in, float* out, const int64_t* dims, const int64_t* strides, int32_t inverse, void* stream) {
  Perm5 p;
  int64_t n = 1;
  for (int k = 0; k < 5; ++k) { p.d[k] = dims[k]; p.s[k] = strides[k]; n *= dims[k]; }
  if (n <= 0) return fail(-1, "nnjt_permute5: empty");
  hipLaunchKernelGGL(k_permute5, dim3(blocks_for(n, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), in, out, p, n,
                     inverse);
  CHK_LAUNCH();
  return 0;
}
}
