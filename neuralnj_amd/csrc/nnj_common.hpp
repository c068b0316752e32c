// nnj_common.hpp -- device helpers shared by the gfx950 kernels of libnnj_hip.so.
//
// Register layout used by every token-local stage ("feature-major tile"):
//   a wave owns NT tiles of 32 tokens; lane l serves token (l & 31) of each tile and
//   lane-half hh = l >> 5.  A 64-feature vector of a token lives in two f32x16
//   accumulators a[0], a[1]; element reg = 4*g + t of a[mt] is feature
//        f = 32*mt + 8*g + 4*hh + t.
//   This is exactly the C/D layout of v_mfma_f32_32x32x2_f32 (col = lane&31,
//   row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)), so the output of one linear layer is
//   directly the B operand of the next one (no LDS round trip), and it is also the
//   natural order of 16-byte loads: chunk (2*(4*mt+g) + hh) of the token's row.
//
// OUT^T[out x tokens] = W[out x in] * IN^T[in x tokens]:  A = W from LDS (one
// ds_read_b128 feeds 4 MFMAs), B = IN from registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define NNJ_D 64          // embed_dim the kernels are specialised for
#define NNJ_NHEAD 8           // heads
#define NNJ_DH 8          // head dim
#define NNJ_F 256         // FFN hidden
// bits of the sticky per-handle status word (nnj_numeric_status)
#define NNJ_FLAG_NONFINITE 1          // a pair-score table held a non-finite score
#define NNJ_FLAG_BARRIER_TIMEOUT 2    // an LDS-counter barrier between partner waves gave up waiting
#define NNJ_FLAG_MERGE_WEIGHTS 4      // two-pass step: a merge's attention weights had no source and no fallback was launched
#define NNJ_FLAG_BAD_MERGE 8          // tree likelihood: a caller's merge list held a pair with i >= j or a position outside the live list

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// exp via v_exp_f32: relative error ~1e-7*|x|; every use here is exp(x), x <= 0 (softmax
// terms, sigmoid, erfc tail), so the absolute error stays below fp32 rounding of the sums.
__device__ __forceinline__ float nnj_exp(float x) { return __expf(x); }
// 1/x and 1/sqrt(x) by v_rcp_f32 / v_rsq_f32 (1 ulp): an IEEE division costs ~10 VALU instructions
__device__ __forceinline__ float nnj_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float nnj_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float sigmoidf_(float x) {
  // 1/(1+e^-x): e^-x overflowing to +inf gives rcp(inf) = 0, the correct limit
  return nnj_rcp(1.0f + nnj_exp(-x));
}
// sigmoid of t given xs = -log2(e) * t (the scorer's W_h, b_h, W_g, b_g are stored pre-scaled): one
// multiplication less per element in kernels that are bound by vector issue
__device__ __forceinline__ float sigmoid_l2(float xs) { return nnj_rcp(1.0f + __builtin_amdgcn_exp2f(xs)); }
// nn.GELU() default (exact erf form), select free:  x Phi(x) = x/2 + |x| (1/2 - h(|x|)),
// h(a) = erfc(a / sqrt2) / 2 by Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7 on erfc, a few fp32 ulps of the
// result) with the polynomial coefficients halved: 11 vector + 2 transcendental instructions (|x| is a source
// modifier).  The earlier form max(x, 0) - |x| h cost four more: fmaxf on a matrix-pipe result gets a
// canonicalising v_max in front of it, and its explicit z = |x| / sqrt2 one multiplication.  Measured with GELU
// knocked out (results wrong, time only): k_ffn16 24.4 -> 16.4 ms, k_pair_score 35.4 -> 28.0, incremental scores
// 59.1 -> 49.3 per rollout of 256 -- these kernels are bound by vector issue, every instruction here counts.
__device__ __forceinline__ float gelu_erf(float x) {
  const float ax = fabsf(x);
  const float t = nnj_rcp(ax * (0.3275911f * 0.70710678118654752440f) + 1.0f);
  float p = 0.5f * 1.061405429f;
  p = p * t - 0.5f * 1.453152027f;
  p = p * t + 0.5f * 1.421413741f;
  p = p * t - 0.5f * 0.284496736f;
  p = p * t + 0.5f * 0.254829592f;
  const float y = ax * 0.84932180028801904272f;             // sqrt(log2(e) / 2): exp(-x^2 / 2) = exp2(-y^2)
  const float e = __builtin_amdgcn_exp2f(-(y * y));
  const float q = 0.5f - (p * t) * e;                       // 1/2 - erfc(|x|/sqrt2) / 2
  return ax * q + 0.5f * x;
}

// The same GELU on TWO values with packed fp32 arithmetic (v_pk_fma_f32 / v_pk_mul_f32: two elements per 4-cycle issue
// slot): 2 v_and + 10 packed + 4 transcendental instructions per pair -- 40 issue cycles per element against 60 -- and
// s += gelu(x) . w for the scorer's last layer as one more packed fma into a two-lane accumulator.  Same operations
// as gelu_erf per element (up to which of the last two products the compiler fuses).
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2v gelu_erf2(f32x2v x) {
  const f32x2v ax = __builtin_elementwise_abs(x);
  const f32x2v one = {1.0f, 1.0f}, half = {0.5f, 0.5f};
  const f32x2v k = {0.3275911f * 0.70710678118654752440f, 0.3275911f * 0.70710678118654752440f};
  const f32x2v u = __builtin_elementwise_fma(ax, k, one);
  const f32x2v t = {nnj_rcp(u[0]), nnj_rcp(u[1])};
  const f32x2v c4 = {-0.5f * 1.453152027f, -0.5f * 1.453152027f}, c3 = {0.5f * 1.421413741f, 0.5f * 1.421413741f};
  const f32x2v c2 = {-0.5f * 0.284496736f, -0.5f * 0.284496736f}, c1 = {0.5f * 0.254829592f, 0.5f * 0.254829592f};
  const f32x2v c5 = {0.5f * 1.061405429f, 0.5f * 1.061405429f};
  f32x2v p = __builtin_elementwise_fma(c5, t, c4);
  p = __builtin_elementwise_fma(p, t, c3);
  p = __builtin_elementwise_fma(p, t, c2);
  p = __builtin_elementwise_fma(p, t, c1);
  const f32x2v ys = {0.84932180028801904272f, 0.84932180028801904272f};
  const f32x2v y = ax * ys;
  const f32x2v y2 = y * y;
  const f32x2v e = {__builtin_amdgcn_exp2f(-y2[0]), __builtin_amdgcn_exp2f(-y2[1])};
  const f32x2v q = __builtin_elementwise_fma(-(p * t), e, half);
  return __builtin_elementwise_fma(x, half, ax * q);
}
// Two sigmoids / gate mixes at a time with PACKED fp32 arithmetic (round 5).  The compiler packs the fused multiply-adds
// of these chains by itself but leaves the `1 + 2^x` additions and the `a - b` differences scalar (k_inc_score_w<3>: 108
// v_add_f32 + 96 v_sub_f32 per site next to 265 v_pk_fma_f32): as packed FMAs the loop body has 8 % fewer vector
// instructions (1053 -> 967).  Same values up to the fusion: the mixes are explicit FMAs now where the compiler had emitted a
// packed multiply and a packed add in places (one rounding fewer; the tables move in the seventh digit: the e64 scan of
// profiles/r05 is of this code).  What it buys is small -- at two waves per SIMD a
// scalar fp32 instruction already runs at the pipe's rate, a packed one saves issue slots only: incremental scores 49.2 ->
// 48.3 ms per rollout; in the 32-pair kernels of step 0 (25 % fewer vector instructions in the gate block) nothing, so
// those keep the scalar form (profiles/r05/ab_packed_gates.txt).
// (a - b and e + 1 are written as fma(a, 1, -b) / fma(e, 1, 1): exactly the same values, and the packed FMA is the one form
// the compiler keeps packed -- plain vector differences come out as two v_sub_f32 again)
// The multiplier 1.0 is made opaque to the optimiser (no instruction; identical calls are merged and hoisted), as the -1.0 of
// split2 is: otherwise fma(a, 1, -b) is folded back into a - b and scalarised.
__device__ __forceinline__ f32x2v opaque_ones() {
  float k = 1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(k));
#endif
  return (f32x2v){k, k};
}
__device__ __forceinline__ f32x2v pk_sub(f32x2v a, f32x2v b) { return __builtin_elementwise_fma(a, opaque_ones(), -b); }
__device__ __forceinline__ f32x2v sigmoid_l2x2(f32x2v xs) {
  const f32x2v e = {__builtin_amdgcn_exp2f(xs[0]), __builtin_amdgcn_exp2f(xs[1])};
  const f32x2v d = __builtin_elementwise_fma(e, opaque_ones(), (f32x2v){1.0f, 1.0f});
  return (f32x2v){nnj_rcp(d[0]), nnj_rcp(d[1])};
}
// x += sigmoid(g) * (xg - x)  for four features (g = -log2(e) x the gate's pre-activation): (1-w) x + w x_g, model.py:150-153
__device__ __forceinline__ void gate_mix4(f32x4& x, const f32x4& xg, const f32x4& g) {
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x2v w = sigmoid_l2x2((f32x2v){g[2 * p], g[2 * p + 1]});
    const f32x2v xv = {x[2 * p], x[2 * p + 1]}, gv = {xg[2 * p], xg[2 * p + 1]};
    const f32x2v r = __builtin_elementwise_fma(w, pk_sub(gv, xv), xv);
    x[2 * p] = r[0]; x[2 * p + 1] = r[1];
  }
}
// out = b + sigmoid(z) * (a - b) for four features: z x_i + (1-z) x_j, model.py:105-108
__device__ __forceinline__ f32x4 gate_sel4(const f32x4& a, const f32x4& b, const f32x4& z) {
  f32x4 o;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const f32x2v w = sigmoid_l2x2((f32x2v){z[2 * p], z[2 * p + 1]});
    const f32x2v av = {a[2 * p], a[2 * p + 1]}, bv = {b[2 * p], b[2 * p + 1]};
    const f32x2v r = __builtin_elementwise_fma(w, pk_sub(av, bv), bv);
    o[2 * p] = r[0]; o[2 * p + 1] = r[1];
  }
  return o;
}
// s2 += gelu(x[0..3]) * w[0..3] (two-lane accumulator; the caller adds the lanes once per site)
__device__ __forceinline__ void gelu_dot4(f32x2v& s2, const f32x4& x, const f32x4& w) {
  const f32x2v g0 = gelu_erf2((f32x2v){x[0], x[1]}), g1 = gelu_erf2((f32x2v){x[2], x[3]});
  s2 = __builtin_elementwise_fma(g0, (f32x2v){w[0], w[1]}, s2);
  s2 = __builtin_elementwise_fma(g1, (f32x2v){w[2], w[3]}, s2);
}

// ---- LDS-DMA: 16 bytes per lane HBM -> LDS with no register staging (global_load_lds_dwordx4).
// LDS destination = wave-uniform base + lane*16; the per-lane SOURCE address is free.
// (The builtins exist only in the device pass; the host pass of hipcc sees empty bodies.)
__device__ __forceinline__ void lds_dma16(const float* gsrc, float* lds_base_uniform) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_global_load_lds(gsrc, lds_base_uniform, 16, 0, 0);
#endif
}
__device__ __forceinline__ void wait_vmem_all() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// ---- hand-issued LDS reads (the compiler would drain in-flight LDS-DMA with vmcnt(0) before every
// ds_read it issues itself).  Protocol (cdna_hip_programming.md 5.7, form ii): issue a batch with
// lds_read_*, later name every destination in lds_wait_* before the first consumer.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ void lds_read_b64(f32x2& d, unsigned byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(byte_addr), "n"(OFF));
#endif
}
template <int OFF>
__device__ __forceinline__ void lds_read_b128(f32x4& d, unsigned byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(byte_addr), "n"(OFF));
#endif
}
template <int OFF>
__device__ __forceinline__ void lds_read_b32(float& d, unsigned byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(byte_addr), "n"(OFF));
#endif
}
__device__ __forceinline__ void lds_wait_all() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);     // register-only MFMAs must not be hoisted above the wait (rule 18)
#endif
}
// makes a value "defined here": no consumer of x can be scheduled above this point
template <typename T>
__device__ __forceinline__ void pin_after_wait(T& x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(x));
#endif
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {   // LDS byte offset of a __shared__ pointer
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// ---- LDS weight image: W[out][in] row-major, 16-byte chunks XOR-swizzled by row so
// that a ds_read_b128 whose 16-lane groups read 16 different rows at the same logical
// chunk is bank-conflict free.  IN must be a multiple of 64 floats.
__device__ __forceinline__ int wswz(int row, int chunk) { return chunk ^ (row & 15); }

// cooperative copy of a [rows][IN] fp32 matrix from global to the swizzled LDS image
template <int IN>
__device__ __forceinline__ void stage_weight(float* lds, const float* __restrict__ g, int rows,
                                             int tid, int nthreads) {
  constexpr int CH = IN / 4;
  for (int i = tid; i < rows * CH; i += nthreads) {
    const int r = i / CH, c = i % CH;
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + (size_t)r * IN + 4 * c);
    *reinterpret_cast<f32x4*>(lds + r * IN + 4 * wswz(r, c)) = v;
  }
}
// sub-block variant: rows [r0, r0+rows) and input columns [c0, c0+INSUB) of a [*, IN] matrix
template <int IN, int INSUB>
__device__ __forceinline__ void stage_weight_sub(float* lds, const float* __restrict__ g, int r0, int rows,
                                                 int c0, int tid, int nthreads) {
  constexpr int CH = INSUB / 4;
  for (int i = tid; i < rows * CH; i += nthreads) {
    const int r = i / CH, c = i % CH;
    const f32x4 v = *reinterpret_cast<const f32x4*>(g + (size_t)(r0 + r) * IN + c0 + 4 * c);
    *reinterpret_cast<f32x4*>(lds + r * INSUB + 4 * wswz(r, c)) = v;
  }
}

// ---- linear layer on feature-major tiles.
//   out[nt][mt] (MT tiles of 32 out-features) = bias + W * in   (or out += W * in),  in has KT tiles.
//   W: LDS image [32*MT][LDW] (XOR-swizzled 16-byte chunks when SWZ); bias readable with 16-byte loads.
//   The A fragments are software pipelined (fragment s+1 is read before the MFMAs of fragment s) and
//   consecutive MFMA groups alternate between the MT independent accumulators.  Per accumulator the
//   k order is ascending, so results do not depend on the interleaving.
template <int MT, int KT, int NT, int LDW, bool SWZ, bool ACC, bool BIAS>
__device__ __forceinline__ void linear_core(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT],
                                            const float* W, const float* bias, int lane) {
  const int row = lane & 31, hh = lane >> 5;
  if constexpr (!ACC) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BIAS) b4 = *reinterpret_cast<const f32x4*>(bias + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          out[nt][mt][4 * g + 0] = b4[0]; out[nt][mt][4 * g + 1] = b4[1];
          out[nt][mt][4 * g + 2] = b4[2]; out[nt][mt][4 * g + 3] = b4[3];
        }
      }
  }
  constexpr int NSTEP = KT * 4 * MT;          // step s: kt = s / (4*MT), g = (s / MT) % 4, mt = s % MT
  auto frag = [&](int s) {
    const int kt = s / (4 * MT), g = (s / MT) % 4, mt = s % MT;
    const int wrow = 32 * mt + row;
    const int chunk = (32 * kt + 8 * g + 4 * hh) >> 2;
    return *reinterpret_cast<const f32x4*>(W + wrow * LDW + 4 * (SWZ ? wswz(wrow, chunk) : chunk));
  };
  f32x4 a[2];
  a[0] = frag(0);
  static_for<0, NSTEP>([&](auto si) {
    constexpr int s = decltype(si)::value;
    constexpr int kt = s / (4 * MT), g = (s / MT) % 4, mt = s % MT;
    if constexpr (s + 1 < NSTEP) a[(s + 1) & 1] = frag(s + 1);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) out[nt][mt] = mfma32(a[s & 1][t], in[nt][kt][4 * g + t], out[nt][mt]);
  });
}
// out = bias + W*in (bias must not be null)
template <int MT, int KT, int NT, int LDW = 32 * KT, bool SWZ = true>
__device__ __forceinline__ void linear_T(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT],
                                         const float* W, const float* bias, int lane) {
  linear_core<MT, KT, NT, LDW, SWZ, false, true>(out, in, W, bias, lane);
}
// out = W*in
template <int MT, int KT, int NT, int LDW = 32 * KT, bool SWZ = true>
__device__ __forceinline__ void linear_T_nb(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT],
                                            const float* W, int lane) {
  linear_core<MT, KT, NT, LDW, SWZ, false, false>(out, in, W, nullptr, lane);
}
// out += W*in
template <int MT, int KT, int NT, int LDW = 32 * KT, bool SWZ = true>
__device__ __forceinline__ void linear_T_acc(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT],
                                             const float* W, int lane) {
  linear_core<MT, KT, NT, LDW, SWZ, true, false>(out, in, W, nullptr, lane);
}

// ---- fp32 linear layers on the 16-bit matrix pipe ("f16x3").
// v_mfma_f32_32x32x16_f16 moves 16x the k of v_mfma_f32_32x32x2_f32 per cycle.  Every fp32 operand is split
// into two fp16 pieces, x = h + m: h = fp16(x) (round to nearest), m = fp16(x - h) (the subtraction is exact).
// h + m carries 22 of the 24 significand bits (fp16 denormals are kept by the conversion and by the MFMA on
// gfx950 -- tools/probe/f16_denorm.hip -- so small values degrade gracefully: absolute error <= 3e-8), and a
// product is accumulated in fp32 from the three piece products hh + hm + mh (mm is below 2^-22 of the
// product).  Measured on random GEMMs the result is within 1.3-1.7x of the error of a plain fp32 GEMM
// (which has its own accumulation rounding) as long as the operands are not uniformly below ~1e-2 in
// magnitude -- true of every operand here (weights, LayerNorm outputs, activations, softmax weights, q scaled
// by 0.05) -- at 3/16 of the fp32-MFMA cycles and 4 bytes per stored operand element.  Operand magnitudes must
// stay below 65504 (they do: the largest are the S rows, O(10)).  The C/D layout of the 32x32x16 instruction
// is the feature-major tile above, and its B operand of k-step G of a 32-feature tile is registers
// 8G..8G+7 of that tile: the layer chaining of linear_core carries over unchanged.
// (Identifiers still say "b6"/"6": the scheme started as a 3-piece bf16 split with six products.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int NPL = 2;                                     // pieces (= planes of a stored operand) per fp32 value
struct Frag3 { u32x4 h, m; };                              // 8 fp16 per piece

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// round to nearest, lo -> bits 15:0.  A plain cast: the compiler sees the conversion and inserts the
// VALU -> MFMA hazard waits itself, which it cannot do for inline asm.
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {
  const f16x2 v = {(_Float16)lo, (_Float16)hi};
  return __builtin_bit_cast(unsigned, v);
}
// hides a value's provenance from the optimiser (no instruction): without it the compiler re-converts each
// half separately instead of unpacking the pair it already has
__device__ __forceinline__ void opaque(unsigned& x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#endif
}
// -1.0f the optimiser cannot see through (no instruction; identical calls are merged and hoisted out of loops)
__device__ __forceinline__ float opaque_neg_one() {
  float k = -1.0f;
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+s"(k));
#endif
  return k;
}
// two fp32 -> two packed fp16 pairs (h, m): THREE instructions -- v_cvt_pk_f16_f32 for h, then the residuals
// a - h.lo, b - h.hi computed in fp32 and rounded to fp16 by v_fma_mixlo_f16 / v_fma_mixhi_f16, which take the fp16
// halves of h as they stand and write the halves of m.  Written as `a - (float)h` the compiler emits two
// v_cvt_f32_f16, a v_pk_add_f32 (plus the wait state its packed result needs) and a second v_cvt_pk: six issue
// slots per pair, and the operand splits were a fifth of the vector work of the token kernels (knocked out, time
// only: k_tok1p 36.4 -> 29.9 ms per rollout, incremental alpha 31.7 -> 28.4, k_row_pv 36.0 -> 33.3).  The mixed-
// precision FMA is selected only for fma(fpext(h), k, a) with a multiplier the optimiser cannot fold -- hence the
// opaque -1 -- and only while the two residuals are not paired by the SLP vectoriser: the low half passes through an
// empty asm, so the high one is a lone fpround(fma) next to a finished low half (the v_fma_mixhi pattern).  The
// compiler schedules these like any other VALU instruction (hazard wait states included); results are bit-identical
// to the subtraction (the fma is exact).
// MIX = false: the residuals as plain subtractions (two v_cvt_f32_f16, one v_pk_add_f32, a second v_cvt_pk).  The
// mixed-precision instructions issue at half rate, so the gain is the removed wait states and registers rather than
// the instruction count, and it depends on the kernel: measured per rollout of 256, k_tok1p 35.4 -> 34.0 ms and the
// incremental alpha kernels 32.5 -> 30.4, but the incremental SCORE kernels 56.9 -> 58.8 -- those keep this form.
template <bool MIX = true>
__device__ __forceinline__ void split2(float a, float b, unsigned& h, unsigned& m) {
  h = cvt_pk_f16(a, b);
  opaque(h);
  const f16x2 hv = __builtin_bit_cast(f16x2, h);
  if constexpr (!MIX) {
    const f32x2v r = pk_sub((f32x2v){a, b}, (f32x2v){(float)hv[0], (float)hv[1]});      // exact either way
    m = cvt_pk_f16(r[0], r[1]);
    return;
  }
  const float k = opaque_neg_one();
  _Float16 m0 = (_Float16)__builtin_fmaf((float)hv[0], k, a);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(m0));
#endif
  const f16x2 mv = {m0, (_Float16)__builtin_fmaf((float)hv[1], k, b)};
  m = __builtin_bit_cast(unsigned, mv);
}
// registers base..base+7 of a feature-major tile -> the fragment of one k-step
template <int BASE, bool MIX = true>
__device__ __forceinline__ void split8(Frag3& o, const f32x16& x) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    unsigned h, m;
    split2<MIX>(x[BASE + 2 * p], x[BASE + 2 * p + 1], h, m);
    o.h[p] = h; o.m[p] = m;
  }
}
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c,
                                                 0, 0, 0);
}
// c += A*B from the three leading piece products (m.h, h.m, h.h), smallest first
// (a fourth piece product m.m was measured in round 4 -- profiles/r04/noise_variants.txt, mm4_cost.txt: -11 % trees/s, no systematic gain)
__device__ __forceinline__ f32x16 mfma_b6(const Frag3& a, const Frag3& b, f32x16 c) {
  c = mfma_f16(a.m, b.h, c);
  c = mfma_f16(a.h, b.m, c);
  c = mfma_f16(a.h, b.h, c);
  return c;
}

// ---- hand-issued fragment reads and counted waits for kernels whose operand images arrive by LDS-DMA
template <int OFF>
__device__ __forceinline__ void lds_read_frag(u32x4& d, unsigned byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(byte_addr), "n"(OFF));
#endif
}
template <int N>
__device__ __forceinline__ void lds_wait_le() {     // at most N LDS reads still in flight (they return in order)
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmem_le() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
#endif
}
// Workgroup barrier WITHOUT the fence of __syncthreads(): the fence drains vmcnt to 0, i.e. it would wait
// for the LDS-DMA of the tiles that are meant to stay in flight across the barrier.  The callers wait for
// exactly the pieces they need (counted vmcnt) before it; LDS is only read by hand-issued reads after it.
__device__ __forceinline__ void barrier_nofence() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_barrier" ::: "memory");
#endif
}
__device__ __forceinline__ void pin_frag(Frag3& f) {
  pin_after_wait(f.h); pin_after_wait(f.m);
}
// LDS weight image for the f16x3 layers: two planes (h, m) of [rows][IN] fp16.  Within a row the
// in-features are permuted so that the eight a lane needs for one k-step are one 16-byte chunk:
// chunk q = 4*kt + 2*G + hh holds features 32kt + 16G + 8j + 4hh + t (j = 0,1; t = 0..3) in order
// (j, t); chunks are XOR-swizzled by row & 7 (ds_read_b128 of 32 rows x 2 chunks: conflict free).
// Plane stride = rows*IN*2 bytes.  One matrix takes rows*IN*6 bytes = 1.5x its fp32 image.
// A ds_read_b128 is served in four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32) over 64
// banks = sixteen 16-byte slots per 256 bytes; a group reads 16 rows at one logical chunk.  The XOR below
// sends those 16 rows to 16 different slots for every row length (rows of 64, 128 and >= 256 bytes).
template <int CH>
__device__ __forceinline__ int wswz6(int row, int chunk) {
  return CH >= 16 ? chunk ^ (row & 15) : (CH == 8 ? chunk ^ ((row >> 1) & 7) : chunk ^ ((row >> 2) & 3));
}
// `g` is a [*, GLD] fp32 matrix; rows [0, rows) x columns [c0, c0+IN) of it go to image rows
// [row0, row0+rows) of an image of `img_rows` rows (several matrices can share one image).
template <int IN, int GLD = IN>
__device__ __forceinline__ void stage_weight_b6(float* lds, const float* __restrict__ g, int rows, int tid,
                                                int nthreads, int row0 = 0, int img_rows = 0, int c0 = 0) {
  constexpr int CH = IN / 8;
  u32x4* img = reinterpret_cast<u32x4*>(lds);
  const int plane = (img_rows ? img_rows : rows) * CH;       // in 16-byte units
  for (int i = tid; i < rows * CH; i += nthreads) {
    const int r = i / CH, q = i % CH;
    const int f0 = 32 * (q >> 2) + 16 * ((q >> 1) & 1) + 4 * (q & 1);
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(g + (size_t)r * GLD + c0 + f0);
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(g + (size_t)r * GLD + c0 + f0 + 8);
    Frag3 f;
    unsigned h, m;
    split2(v0[0], v0[1], h, m); f.h[0] = h; f.m[0] = m;
    split2(v0[2], v0[3], h, m); f.h[1] = h; f.m[1] = m;
    split2(v1[0], v1[1], h, m); f.h[2] = h; f.m[2] = m;
    split2(v1[2], v1[3], h, m); f.h[3] = h; f.m[3] = m;
    const int ir = row0 + r;
    const int o = ir * CH + wswz6<CH>(ir, q);
    img[o] = f.h; img[plane + o] = f.m;
  }
}
// the same image of the TRANSPOSE of a [IN][rows] fp32 matrix (image row r = column r of g)
template <int IN>
__device__ __forceinline__ void stage_weight_b6_T(float* lds, const float* __restrict__ g, int rows, int tid,
                                                  int nthreads) {
  constexpr int CH = IN / 8;
  u32x4* img = reinterpret_cast<u32x4*>(lds);
  const int plane = rows * CH;
  for (int i = tid; i < rows * CH; i += nthreads) {
    const int r = i / CH, q = i % CH;
    const int f0 = 32 * (q >> 2) + 16 * ((q >> 1) & 1) + 4 * (q & 1);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = g[(size_t)(f0 + (e & 3) + 8 * (e >> 2)) * rows + r];
    Frag3 f;
    unsigned h, m;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      split2(v[2 * e], v[2 * e + 1], h, m);
      f.h[e] = h; f.m[e] = m;
    }
    const int o = r * CH + wswz6<CH>(r, q);
    img[o] = f.h; img[plane + o] = f.m;
  }
}
// floats of LDS one staged matrix occupies
__host__ __device__ constexpr int b6_floats(int rows, int in) { return rows * in * NPL / 2; }
constexpr int IMG64 = b6_floats(64, 64);                   // floats of a [64][64] operand image

// out[nt][mt] = bias + W*in (or += W*in), W = a stage_weight_b6 image of [32*MT][32*KT].
template <int MT, int KT, int NT, bool ACC, bool BIAS, bool LEAN = false, bool MIX = true>
__device__ __forceinline__ void linear_core_b6(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT],
                                               const float* W, const float* bias, int lane) {
  const int row = lane & 31, hh = lane >> 5;
  if constexpr (!ACC) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BIAS) b4 = *reinterpret_cast<const f32x4*>(bias + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          out[nt][mt][4 * g + 0] = b4[0]; out[nt][mt][4 * g + 1] = b4[1];
          out[nt][mt][4 * g + 2] = b4[2]; out[nt][mt][4 * g + 3] = b4[3];
        }
      }
  }
  constexpr int CH = 4 * KT;                                 // 16-byte chunks per image row
  constexpr int PLANE = 32 * MT * CH;
  const u32x4* img = reinterpret_cast<const u32x4*>(W);
  constexpr int NSTEP = 2 * KT * MT;                         // step s: ks = s / MT (k-step), mt = s % MT
  auto frag = [&](int s, Frag3& a) {
    const int ks = s / MT, mt = s % MT;
    const int wrow = 32 * mt + row;
    const int o = wrow * CH + wswz6<CH>(wrow, 2 * ks + hh);
    a.h = img[o]; a.m = img[PLANE + o];
  };
  if constexpr (LEAN) {
    // register-lean form for kernels at two waves per SIMD (the partner wave fills the gaps): one A and
    // one B fragment live at a time, 24 registers instead of 48
    static_for<0, 2 * KT>([&](auto ki) {
      constexpr int ks = decltype(ki)::value;
      Frag3 b[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) split8<8 * (ks & 1), MIX>(b[nt], in[nt][ks >> 1]);
      static_for<0, MT>([&](auto mi) {
        constexpr int mt = decltype(mi)::value;
        Frag3 a;
        frag(ks * MT + mt, a);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) out[nt][mt] = mfma_b6(a, b[nt], out[nt][mt]);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
  } else {
  // Software pipeline, one scheduling region per k-step: the MFMAs of k-step ks run beside the split of
  // the B fragment of ks+1 (VALU) and the A reads of the next step (LDS); the region boundary keeps the
  // splits of later steps from being hoisted (their live ranges would not fit the register file).
  Frag3 a[2];
  Frag3 b[2][NT];
  frag(0, a[0]);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) split8<0, MIX>(b[0][nt], in[nt][0]);
  static_for<0, 2 * KT>([&](auto ki) {
    constexpr int ks = decltype(ki)::value;
    static_for<0, MT>([&](auto mi) {
      constexpr int mt = decltype(mi)::value;
      constexpr int s = ks * MT + mt;
      if constexpr (s + 1 < NSTEP) frag(s + 1, a[(s + 1) & 1]);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) out[nt][mt] = mfma_b6(a[s & 1], b[ks & 1][nt], out[nt][mt]);
    });
    if constexpr (ks + 1 < 2 * KT) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) split8<8 * ((ks + 1) & 1), MIX>(b[(ks + 1) & 1][nt], in[nt][(ks + 1) >> 1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
  }
}
template <int MT, int KT, int NT, bool LEAN = false, bool MIX = true>
__device__ __forceinline__ void linear6_T(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT], const float* W,
                                          const float* bias, int lane) {
  linear_core_b6<MT, KT, NT, false, true, LEAN, MIX>(out, in, W, bias, lane);
}
template <int MT, int KT, int NT, bool LEAN = false>
__device__ __forceinline__ void linear6_T_nb(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT], const float* W,
                                             int lane) {
  linear_core_b6<MT, KT, NT, false, false, LEAN>(out, in, W, nullptr, lane);
}
template <int MT, int KT, int NT, bool LEAN = false>
__device__ __forceinline__ void linear6_T_acc(f32x16 (&out)[NT][MT], const f32x16 (&in)[NT][KT], const float* W,
                                              int lane) {
  linear_core_b6<MT, KT, NT, true, false, LEAN>(out, in, W, nullptr, lane);
}

// out = W * in for an input that is ALREADY split into k-step fragments (the softmax kernels write the context
// weights alpha as fp16 pieces: the consumers' site loops carry no split of them).  W = image of [32*MT][16*KS2].
template <int MT, int KS2>
__device__ __forceinline__ void linear6_pre(f32x16 (&out)[1][MT], const Frag3 (&b)[KS2], const float* W, int lane) {
  const int row = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[0][mt][r] = 0.f;
  constexpr int CH = 2 * KS2;
  constexpr int PLANE = 32 * MT * CH;
  const u32x4* img = reinterpret_cast<const u32x4*>(W);
  static_for<0, KS2>([&](auto ki) {
    constexpr int ks = decltype(ki)::value;
    static_for<0, MT>([&](auto mi) {
      constexpr int mt = decltype(mi)::value;
      const int wrow = 32 * mt + row;
      const int o = wrow * CH + wswz6<CH>(wrow, 2 * ks + hh);
      Frag3 a;
      a.h = img[o]; a.m = img[PLANE + o];
      out[0][mt] = mfma_b6(a, b[ks], out[0][mt]);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}
// alpha planes: the two fp16 pieces of a softmax weight, `plane` fp16 elements apart
__device__ __forceinline__ void alpha_store(float* alpha, long plane, size_t idx, float a) {
  _Float16* ah = reinterpret_cast<_Float16*>(alpha);
  const _Float16 h = (_Float16)a;
  ah[idx] = h;
  ah[plane + idx] = (_Float16)(a - (float)h);
}
// fragment of k-step ks for a 16-token-tile consumer: 8 consecutive r' (natural order)
__device__ __forceinline__ void alpha_frag16(Frag3& f, const float* alpha, long plane, size_t idx) {
  const unsigned short* ah = reinterpret_cast<const unsigned short*>(alpha);
  f.h = *reinterpret_cast<const u32x4*>(ah + idx);
  f.m = *reinterpret_cast<const u32x4*>(ah + plane + idx);
}
// ... for a 32-token-tile consumer: k-step ks of row `row_idx` (element offset of the row): r' = 16ks + 8j + 4hh + t
__device__ __forceinline__ void alpha_frag32(Frag3& f, const float* alpha, long plane, size_t row_idx, int ks, int hh,
                                             bool ok) {
  const unsigned short* ah = reinterpret_cast<const unsigned short*>(alpha) + row_idx + 16 * ks + 4 * hh;
  uint2 h0 = make_uint2(0u, 0u), h1 = h0, m0 = h0, m1 = h0;
  if (ok) {
    h0 = *reinterpret_cast<const uint2*>(ah); h1 = *reinterpret_cast<const uint2*>(ah + 8);
    m0 = *reinterpret_cast<const uint2*>(ah + plane); m1 = *reinterpret_cast<const uint2*>(ah + plane + 8);
  }
  f.h = (u32x4){h0.x, h0.y, h1.x, h1.y};
  f.m = (u32x4){m0.x, m0.y, m1.x, m1.y};
}

// ---- 16-token tiles.  lane = (token l&15, feature quarter kq = l>>4); a 64-feature vector is four f32x4 tiles,
// element r of tile mt = feature 16*mt + 4*kq + r -- the C/D layout of v_mfma_f32_16x16x32_f16 -- so a tensor
// costs 16 registers per lane (32 with the 32-token tiles): kernels whose dependency chains are latency bound
// run three or four waves per SIMD on it.  The B operand of k-step ks (32 features) is tiles 2ks, 2ks+1:
// k-slot 8*kq + 4*u + r = feature 32*ks + 16*u + 4*kq + r; weight images store their columns in that order.
struct V64 { f32x4 t[4]; };

__device__ __forceinline__ f32x4 mfma16_f16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0,
                                                 0);
}
__device__ __forceinline__ f32x4 mfma16_b6(const Frag3& a, const Frag3& b, f32x4 c) {
  c = mfma16_f16(a.m, b.h, c);
  c = mfma16_f16(a.h, b.m, c);
  c = mfma16_f16(a.h, b.h, c);
  return c;
}
// eight values -> one k-step fragment (k-slot order: a[0..3], b[0..3])
template <bool MIX = true>
__device__ __forceinline__ void split_8(Frag3& o, const f32x4& a, const f32x4& b) {
  unsigned h, m;
  split2<MIX>(a[0], a[1], h, m); o.h[0] = h; o.m[0] = m;
  split2<MIX>(a[2], a[3], h, m); o.h[1] = h; o.m[1] = m;
  split2<MIX>(b[0], b[1], h, m); o.h[2] = h; o.m[2] = m;
  split2<MIX>(b[2], b[3], h, m); o.h[3] = h; o.m[3] = m;
}
__device__ __forceinline__ void load_v64(V64& v, const float* p, int kq) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) v.t[mt] = *reinterpret_cast<const f32x4*>(p + 16 * mt + 4 * kq);
}
// LDS image of a [rows][64] fp32 matrix for the 16-token linears: two planes of [rows][8 chunks of 16 B],
// chunk q = 4*ks + kg holds in-features 32ks + 16u + 4kg + r in (u, r) order, XOR-swizzled like the 32-token
// images (a ds_read_b128 of 16 rows x 4 chunks is conflict free).
__device__ __forceinline__ void stage_weight_t16(float* lds, const float* __restrict__ g, int rows, int tid,
                                                 int nthreads, bool transposed = false, int ld = 64, int c0 = 0) {
  u32x4* img = reinterpret_cast<u32x4*>(lds);
  const int plane = rows * 8;
  for (int i = tid; i < rows * 8; i += nthreads) {
    const int r = i >> 3, q = i & 7;
    const int f0 = 32 * (q >> 2) + 4 * (q & 3);
    f32x4 v0, v1;
    if (!transposed) {
      v0 = *reinterpret_cast<const f32x4*>(g + (size_t)r * ld + c0 + f0);
      v1 = *reinterpret_cast<const f32x4*>(g + (size_t)r * ld + c0 + f0 + 16);
    } else {                                               // image row r = column r of g
#pragma unroll
      for (int e = 0; e < 4; ++e) { v0[e] = g[(size_t)(f0 + e) * rows + r]; v1[e] = g[(size_t)(f0 + 16 + e) * rows + r]; }
    }
    Frag3 f;
    split_8(f, v0, v1);
    const int o = r * 8 + wswz6<8>(r, q);
    img[o] = f.h; img[plane + o] = f.m;
  }
}
// out (MT tiles of 16 rows) = bias + W*in or += W*in; W = a [16*MT][64] image of the layout above
// BIAS is a compile-time switch: a run-time `if (bias)` puts every bias load into a basic block of its own and
// serialises their latency.  Callers keep small per-feature vectors (biases, s_out.2 weights) in LDS (scorer_consts).
template <int MT, bool ACC, bool BIAS = true>
__device__ __forceinline__ void linear_t16(f32x4 (&out)[MT], const V64& in, const float* W, const float* bias,
                                           int lane) {
  const int l15 = lane & 15, kq = lane >> 4;
  if constexpr (!ACC) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if constexpr (BIAS) out[mt] = *reinterpret_cast<const f32x4*>(bias + 16 * mt + 4 * kq);
      else out[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  const u32x4* img = reinterpret_cast<const u32x4*>(W);
  constexpr int PLANE = 16 * MT * 8;
  static_for<0, 2>([&](auto ki) {
    constexpr int ks = decltype(ki)::value;
    Frag3 b;
    split_8(b, in.t[2 * ks], in.t[2 * ks + 1]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int row = 16 * mt + l15;
      const int o = row * 8 + wswz6<8>(row, 4 * ks + kq);
      Frag3 a;
      a.h = img[o]; a.m = img[PLANE + o];
      out[mt] = mfma16_b6(a, b, out[mt]);
    }
    __builtin_amdgcn_sched_barrier(0);
  });
}

// The same product with the weight fragments READ BY HAND, PF (k-step, row tile) steps (2: measured against 3 and 4, which cost registers) ahead of the MFMAs that use
// them.  The compiler's own schedule issues each ds_read_b128 pair right in front of its three MFMAs and waits for
// it at once (lgkmcnt(0) some 40 times per tile and site in the pair-scorer chains): with two or three waves per
// SIMD that LDS latency is the largest single stall of those kernels.  Here the reads of step s + PF are issued
// behind the MFMAs of step s and the wait in front of step s is counted.  `bfr[ks]` = the B fragments (the caller
// splits; fragments that exist already -- the S rows of the scorer -- are passed as they are).
// Only for images that are complete before the call and are not written while it runs (weights staged once per
// workgroup): the compiler does not order its own LDS stores against these asm reads.
template <int MT, int PF = 2, typename PRE>
__device__ __forceinline__ void linear_t16p_core(f32x4 (&out)[MT], const Frag3 (&bfr)[2], const float* W, int lane,
                                                 PRE&& pre) {
  const int l15 = lane & 15, kq = lane >> 4;
  constexpr int NS = 2 * MT;
  constexpr int PLANE = 16 * MT * 8 * 16;                  // bytes
  const int sw = (l15 >> 1) & 7;                           // wswz6<8>(16 mt + l15, q) = q ^ ((l15 >> 1) & 7) for every mt
  const unsigned ad[2] = {lds_addr(W) + (unsigned)(l15 * 8 + (kq ^ sw)) * 16u,
                          lds_addr(W) + (unsigned)(l15 * 8 + ((4 + kq) ^ sw)) * 16u};
  Frag3 a[PF];
  auto issue = [&](auto si) {
    constexpr int s = decltype(si)::value;
    if constexpr (s < NS) {
      constexpr int ks = s / MT, mt = s % MT;
      lds_read_frag<mt * 2048>(a[s % PF].h, ad[ks]);
      lds_read_frag<mt * 2048 + PLANE>(a[s % PF].m, ad[ks]);
    }
  };
  static_for<0, PF>([&](auto si) { issue(si); });
  pre();                                                   // the caller's work that hides the first reads' latency (operand split)
  static_for<0, NS>([&](auto si) {
    constexpr int s = decltype(si)::value;
    constexpr int ks = s / MT, mt = s % MT;
    constexpr int ahead = (s + PF - 1 < NS - 1 ? s + PF - 1 : NS - 1) - s;      // steps whose reads are younger than step s
    lds_wait_le<2 * ahead>();
    pin_frag(a[s % PF]);
    out[mt] = mfma16_b6(a[s % PF], bfr[ks], out[mt]);
    issue(std::integral_constant<int, s + PF>{});
  });
}
// NT token tiles through ONE 64 x 64 linear: every weight fragment is read once and multiplies the NT tiles' B
// fragments -- a third of the fragment reads at NT = 3, and between two waits the matrix pipe has 3 NT products on NT
// independent accumulators (piece products outermost, so consecutive MFMAs never depend on each other; per accumulator
// the order is the one of mfma16_b6: bit-identical results).
template <int NT, int PF = 2, typename PRE>
__device__ __forceinline__ void linear_t16p_multi(V64 (&out)[NT], Frag3 (&bfr)[NT][2], const float* W, int lane,
                                                  PRE&& pre) {
  const int l15 = lane & 15, kq = lane >> 4;
  constexpr int MT = 4, NS = 2 * MT;
  constexpr int PLANE = 16 * MT * 8 * 16;                  // bytes
  const int sw = (l15 >> 1) & 7;
  const unsigned ad[2] = {lds_addr(W) + (unsigned)(l15 * 8 + (kq ^ sw)) * 16u,
                          lds_addr(W) + (unsigned)(l15 * 8 + ((4 + kq) ^ sw)) * 16u};
  Frag3 a[PF];
  auto issue = [&](auto si) {
    constexpr int s = decltype(si)::value;
    if constexpr (s < NS) {
      constexpr int ks = s / MT, mt = s % MT;
      lds_read_frag<mt * 2048>(a[s % PF].h, ad[ks]);
      lds_read_frag<mt * 2048 + PLANE>(a[s % PF].m, ad[ks]);
    }
  };
  static_for<0, PF>([&](auto si) { issue(si); });
  static_for<0, NS>([&](auto si) {
    constexpr int s = decltype(si)::value;
    constexpr int ks = s / MT, mt = s % MT;
    constexpr int ahead = (s + PF - 1 < NS - 1 ? s + PF - 1 : NS - 1) - s;
    if constexpr (mt == 0) pre(std::integral_constant<int, ks>{});     // the caller forms the B fragments of k-step ks (pure vector work)
    lds_wait_le<2 * ahead>();
    pin_frag(a[s % PF]);
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t].t[mt] = mfma16_f16(a[s % PF].m, bfr[t][ks].h, out[t].t[mt]);
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t].t[mt] = mfma16_f16(a[s % PF].h, bfr[t][ks].m, out[t].t[mt]);
#pragma unroll
    for (int t = 0; t < NT; ++t) out[t].t[mt] = mfma16_f16(a[s % PF].h, bfr[t][ks].h, out[t].t[mt]);
    issue(std::integral_constant<int, s + PF>{});
  });
}
template <int MT, bool ACC, bool BIAS = true, int PF = 2, bool MIX = true>
__device__ __forceinline__ void linear_t16p(f32x4 (&out)[MT], const V64& in, const float* W, const float* bias,
                                            int lane) {
  const int kq = lane >> 4;
  if constexpr (!ACC) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if constexpr (BIAS) out[mt] = *reinterpret_cast<const f32x4*>(bias + 16 * mt + 4 * kq);
      else out[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  Frag3 b[2];
  linear_t16p_core<MT, PF>(out, b, W, lane, [&] {
    split_8<MIX>(b[0], in.t[0], in.t[1]);
    split_8<MIX>(b[1], in.t[2], in.t[3]);
  });
}

// layer norm of a V64 (the four lanes of a token hold 16 features each).
// dt = the model's embed_dim (a multiple of 8, <= 64): a narrower model runs on these 64-feature kernels with its
// tensors zero-padded (nnj_load_weights); the statistics are the ones of its dt features -- the padding adds nothing to the
// sum and is left out of the variance -- and gamma = beta = 0 there keeps the padded features at zero.  dt = 64: as before.
__device__ __forceinline__ void layer_norm_v64(V64& y, const V64& x, const float* gamma, const float* beta, int kq, int dt) {
  float s = 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) s += (x.t[mt][0] + x.t[mt][1]) + (x.t[mt][2] + x.t[mt][3]);
  s += __shfl_xor(s, 16);
  s += __shfl_xor(s, 32);
  const float inv_d = 1.0f / (float)dt;
  const float mean = s * inv_d;
  float v = 0.f;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const bool in = 16 * mt + 4 * kq < dt;
#pragma unroll
    for (int e = 0; e < 4; ++e) { const float d = x.t[mt][e] - mean; v += in ? d * d : 0.f; }
  }
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  const float inv = nnj_rsqrt(v * inv_d + 1e-5f);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 16 * mt + 4 * kq);
    const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 16 * mt + 4 * kq);
#pragma unroll
    for (int e = 0; e < 4; ++e) y.t[mt][e] = (x.t[mt][e] - mean) * inv * gm[e] + bt[e];
  }
}

// ---- token I/O: 64 features of one token <-> two accumulators (zeros when !valid).
// The loads are UNCONDITIONAL (a branch per load would put every load in its own basic block and
// serialise the memory latency): `p` must be a readable address even when !valid -- callers clamp
// the index -- and the result is zeroed by a select.
__device__ __forceinline__ void load_token64(f32x16 (&a)[2], const float* p, bool valid, int hh) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 32 * mt + 8 * g + 4 * hh);
      a[mt][4 * g + 0] = valid ? v[0] : 0.f; a[mt][4 * g + 1] = valid ? v[1] : 0.f;
      a[mt][4 * g + 2] = valid ? v[2] : 0.f; a[mt][4 * g + 3] = valid ? v[3] : 0.f;
    }
}
__device__ __forceinline__ void store_token64(const f32x16 (&a)[2], float* p, bool valid, int hh) {
  if (!valid) return;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v = {a[mt][4 * g + 0], a[mt][4 * g + 1], a[mt][4 * g + 2], a[mt][4 * g + 3]};
      *reinterpret_cast<f32x4*>(p + 32 * mt + 8 * g + 4 * hh) = v;
    }
}

// LayerNorm over the features of the lane's token (eps 1e-5, biased variance); gamma/beta readable with 16-byte loads.
// dt: see layer_norm_v64.
__device__ __forceinline__ void layer_norm64(f32x16 (&y)[2], const f32x16 (&x)[2],
                                             const float* gamma, const float* beta, int hh, int dt) {
  float s = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[mt][r];
  s += __shfl_xor(s, 32);
  const float inv_d = 1.0f / (float)dt;
  const float mean = s * inv_d;
  float v = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const bool in = 32 * mt + 8 * g < dt;               // features 32 mt + 8 g + 4 hh + t (dt is a multiple of 8)
#pragma unroll
      for (int t = 0; t < 4; ++t) { const float d = x[mt][4 * g + t] - mean; v += in ? d * d : 0.f; }
    }
  v += __shfl_xor(v, 32);
  const float inv = nnj_rsqrt(v * inv_d + 1e-5f);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 32 * mt + 8 * g + 4 * hh);
      const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
      for (int t = 0; t < 4; ++t) y[mt][4 * g + t] = (x[mt][4 * g + t] - mean) * inv * gm[t] + bt[t];
    }
}

// flat index of pair (i,j), i<j, among combinations(range(n),2)
__host__ __device__ __forceinline__ int pair_index(int n, int i, int j) {
  return i * n - i * (i + 1) / 2 + (j - i - 1);
}
__host__ __device__ __forceinline__ int num_pairs(int n) { return n * (n - 1) / 2; }
// inverse: flat p -> (i,j)
__device__ __forceinline__ void pair_from_index(int n, int p, int& i, int& j) {
  // rows before i hold i*n - i(i+1)/2 pairs; solve by float estimate then fix up
  int ii = (int)((2.0f * n - 1.0f - sqrtf((2.0f * n - 1.0f) * (2.0f * n - 1.0f) - 8.0f * (float)p)) * 0.5f);
  if (ii < 0) ii = 0;
  if (ii > n - 2) ii = n - 2;
  while (ii > 0 && ii * n - ii * (ii + 1) / 2 > p) --ii;
  while (ii < n - 2 && (ii + 1) * n - (ii + 1) * (ii + 2) / 2 <= p) ++ii;
  i = ii;
  j = p - (ii * n - ii * (ii + 1) / 2) + ii + 1;
}
