// nnj_scorer16.hpp -- the incremental NJ-step scorer on 16-token tiles: alpha for every n (k_inc_alpha16); scores
// for n <= 16 (k_inc_score16<1>: twelve one-tile waves) and for 33..48 rows (k_inc_score_w<3>: one wave per site
// walks three tiles = 48 instead of 64 padded pairs).  Measured: with 17..32 and 49..64 rows the 32-pair kernels
// of nnj_scorer.hpp are as fast or faster; there they are used.
//
// The per-site dependency chain of the pair scorer (gate -> image -> x_g -> W_g -> mix -> s_out -> GELU) is
// latency bound at two waves per SIMD, and the 32-token feature-major tile (32 registers per 64-feature
// vector) leaves no room for a third.  Here a wave owns 16 pairs: lane = (pair l&15, feature quarter
// kq = l>>4), a 64-feature vector is four f32x4 tiles, element r of tile mt = feature 16*mt + 4*kq + r -- the
// C/D layout of v_mfma_f32_16x16x32_f16 -- so every tensor costs 16 registers, twelve waves fit a CU
// (three per SIMD) and a step pads its n-1 pairs to a multiple of 16 instead of 32.
// The B operand of k-step ks (32 features) is tiles 2ks, 2ks+1: k-slot 8*kq + 4*u + r = feature
// 32*ks + 16*u + 4*kq + r; weight images store their columns in that order (stage_weight_t16).
// Everything is f16x3 (nnj_common.hpp).
#pragma once
#include "nnj_scorer.hpp"

// barrier of the NG waves that share a site slot (LDS counter; see k_tok1p's pair_barrier)
template <int NG>
__device__ __forceinline__ void group_barrier_lds(int* cnt, int& epoch, int* flag) {
  epoch += NG;
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) atomicAdd(cnt, 1);
  int spins = 0;
  for (; *reinterpret_cast<volatile int*>(cnt) < epoch && spins < (1 << 22); ++spins) __builtin_amdgcn_s_sleep(1);
  if (spins == (1 << 22) && (threadIdx.x & 63) == 0) atomicOr(flag, NNJ_FLAG_BARRIER_TIMEOUT);   // never silent
  asm volatile("" ::: "memory");
}

struct Inc16 {            // per-lane description of the step: pair (m, r)
  int r, slot_r, slot_m;
  float sgn;              // +1 if r < m (the pair is (r, m)), -1 otherwise
};
__device__ __forceinline__ Inc16 inc16(const RowSet& rs, const int* ij_prev, int b, int n, int r, int qn = 0) {
  Inc16 L;
  const int m = min(max(ij_prev[2 * b], 0), n - 1);
  L.r = r;                                               // qn: q, the row's index among the rows other than m (q_to_r)
  L.slot_m = slot_of(rs, b, m);
  const int pos = qn ? q_to_r(r, m) : r;
  L.slot_r = slot_of(rs, b, pos < n ? pos : 0);          // lanes beyond the rows read row 0 (never used)
  L.sgn = r < m ? 1.0f : -1.0f;                          // (q < m <=> r < m)
  return L;
}
// x = S_m + sigmoid(U_r - U_m + s*b) (S_r - S_m)   (see inc_gate).  The accumulators of U_r = W_h S_r start at
// s*b - U_m (gate_init16) instead of zero: the MFMAs add U_r on top and the gate is one sigmoid of the accumulator --
// per element one fused multiply-add instead of a zero move, a subtraction and an addition.
__device__ __forceinline__ void gate_init16(V64& ur, const V64& um, const float* bh, float sgn, int kq) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(bh + 16 * mt + 4 * kq);
#pragma unroll
    for (int e = 0; e < 4; ++e) ur.t[mt][e] = sgn * b4[e] - um.t[mt][e];
  }
}
__device__ __forceinline__ void gate16(V64& x, const V64& sr, const V64& ur, const V64& sm) {
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    x.t[mt] = gate_sel4(sr.t[mt], sm.t[mt], ur.t[mt]);      // S_m + sigmoid(.) (S_r - S_m), two features per instruction
    __builtin_amdgcn_sched_barrier(0);
  }
}

constexpr int T16_WAVES = 12;

// ------------------------------------------------------------------ k_inc_alpha16
// alpha partials of the new pairs (see k_inc_alpha: (A^T x).S_r, K' never read).  NG waves (16 pairs each)
// share a site: together they write its 16*NG S rows as a weight-like image [2 planes][16*NG r][64 d] (each
// lane one 16-byte chunk per k-step and plane) and each multiplies its x' with all rows.
// part[b][sc*NSLOT+slot][pair r][r'].
// NW waves per workgroup: 12 (three per SIMD) or, where the registers allow it, 16 / 15 (four per SIMD).
template <int NG, int NW = T16_WAVES>
__global__ __launch_bounds__(64 * NW) void k_inc_alpha16(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                                float* __restrict__ alpha_part, int n, int C, int cs,
                                                                int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NSLOT = NW / NG;
  constexpr int IMG = 16 * NG * 64 * NPL / 2;              // floats of an image
  float* At_l = smem;                                      // A^T, IMG64 floats
  float* Wh_l = smem + IMG64;                              // W_h
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave % NSLOT, tl = wave / NSLOT;
  float* img = smem + 2 * IMG64 + slot * IMG;
  int* cnt0 = reinterpret_cast<int*>(smem + 2 * IMG64 + NSLOT * IMG);
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(At_l, w.imgAt, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  float* cv = smem + 2 * IMG64 + NSLOT * IMG + 16;
  stage_scorer_consts(cv, w, tid);
  if (tid < NSLOT) cnt0[tid] = 0;
  __syncthreads();
  int* cnt = cnt0 + slot;
  int epoch = 0;
  const Inc16 L = inc16(rs, ij_prev, b, n, 16 * tl + l15);
  const size_t bo = (size_t)b * rs.bstride;
  const float* Sr = rs.S + bo + (size_t)L.slot_r * C * 64;
  const float* Sm = rs.S + bo + (size_t)L.slot_m * C * 64;
  const float* Um = rs.U + bo + (size_t)L.slot_m * C * 64;
  f32x4 acc[NG];
#pragma unroll
  for (int mt = 0; mt < NG; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4* im4 = reinterpret_cast<u32x4*>(img);
  constexpr int PL = 16 * NG * 8;
  V64 sr;
  int c = c0 + slot;
  if (c < c1) load_v64(sr, Sr + (size_t)c * 64, kq);
  for (; c < c1; c += NSLOT) {
    asm volatile("" ::: "memory");
    V64 x;
    // The fp16 pieces of S_r serve twice: as this row of the image and as the B operand of U_r = W_h S_r, which
    // is recomputed here (24 MFMAs on an idle pipe) instead of being read: the kernel is HBM bound and the
    // cached U rows were half of its traffic.  (U_m, one row per site, is still read.)
    Frag3 sf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
    {
      V64 sm, um, ur;
      load_v64(sm, Sm + (size_t)c * 64, kq);
      load_v64(um, Um + (size_t)c * 64, kq);
      gate_init16(ur, um, cv, L.sgn, kq);
      linear_t16p_core<4>(ur.t, sf, Wh_l, lane, [] {});
      gate16(x, sr, ur, sm);
    }
    if constexpr (NG > 1) group_barrier_lds<NG>(cnt, epoch, status);     // everyone is done with the previous image
    // row r of the image: chunk 4*ks + kq = this lane's tiles 2ks, 2ks+1
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int o = L.r * 8 + wswz6<8>(L.r, 4 * ks + kq);
      im4[o] = sf[ks].h; im4[PL + o] = sf[ks].m;
    }
    if constexpr (NG > 1) group_barrier_lds<NG>(cnt, epoch, status);     // all rows are in the image
    const int cn = c + NSLOT < c1 ? c + NSLOT : c;               // prefetch behind the MFMAs (last: harmless reload)
    load_v64(sr, Sr + (size_t)cn * 64, kq);
    V64 xp;
    linear_t16p<4, false, false>(xp.t, x, At_l, nullptr, lane);          // x' = A^T x
    linear_t16<NG, true, false>(acc, xp, img, nullptr, lane);           // acc[r'][pair] += S_r' . x'
  }
  // one partial set per WORKGROUP: the slots add their tiles into one LDS tile [64 pairs][64 r'] in slot order
  // (fixed order: bitwise reproducible), the images are dead by then
  __syncthreads();
  float* red = smem + 2 * IMG64;                           // 4096 floats (NSLOT * IMG >= 12288)
  for (int i = tid; i < 4096; i += 64 * NW) red[i] = 0.f;
  __syncthreads();
  for (int s_ = 0; s_ < NSLOT; ++s_) {
    if (slot == s_) {
#pragma unroll
      for (int mt = 0; mt < NG; ++mt) {
        f32x4* d4 = reinterpret_cast<f32x4*>(red + L.r * 64 + 16 * mt + 4 * kq);
        *d4 = *d4 + acc[mt];
      }
    }
    __syncthreads();
  }
  float* dst = alpha_part + ((size_t)b * gridDim.x + sc) * 4096;
  for (int i = tid; i < 1024; i += 64 * NW)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(red)[i];
}

// ------------------------------------------------------------------ k_inc_score16
// scores of the new pairs.  The NG waves of a site build its transposed image S^T [2 planes][64 d][16*NG r']
// (natural r' order; A operand of x_g^T = S^T alpha^T) together, each the columns of its 16 rows.
// part[b][sc*NSLOT+slot][pair r].
template <int NC>                                           // NC = 16-column groups of an image row (1, 2 or 4)
__device__ __forceinline__ int tswz(int d, int chunk) {     // conflict-free chunk swizzle per row length
  return NC == 4 ? chunk ^ ((d >> 1) & 7) : (NC == 2 ? chunk ^ ((-(d >> 2)) & 3) : chunk);
}
template <int NG, bool CTX>
__global__ __launch_bounds__(64 * T16_WAVES) void k_inc_score16(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                                const float* __restrict__ alpha,
                                                                const uint8_t* __restrict__ mask,
                                                                float* __restrict__ score_part, int n, int C, int cs,
                                                                int qn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NSLOT = T16_WAVES / NG;
  constexpr int NC = NG == 3 ? 4 : NG;                     // image geometry: 16*NC columns (3 waves use the 64-column one)
  constexpr int CH = 2 * NC;                               // 16-byte chunks per image row (8 r' each)
  constexpr int KSX = NC == 4 ? 2 : 1;                     // k-steps of the x_g GEMM (32 r' each)
  constexpr int IMG = 64 * 16 * NC * NPL / 2;              // floats of an image
  float* Wg_l = smem;
  float* S0_l = smem + IMG64;
  float* Wh_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave % NSLOT, tl = wave / NSLOT;
  float* img = smem + 3 * IMG64 + slot * IMG;
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * T16_WAVES);
  stage_image_t16(S0_l, w.imgS0, tid, 64 * T16_WAVES);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * T16_WAVES);
  float* cv = smem + 3 * IMG64 + NSLOT * IMG + 16;
  stage_scorer_consts(cv, w, tid);
  if constexpr (NG == 3) {                                 // columns 48..63 are never written: they meet alpha = 0 but must be finite
    for (int i = tid; i < NSLOT * IMG; i += 64 * T16_WAVES) smem[3 * IMG64 + i] = 0.f;
  }
  __syncthreads();
  const Inc16 L = inc16(rs, ij_prev, b, n, 16 * tl + l15, qn);
  const size_t bo = (size_t)b * rs.bstride;
  const float* Sr = rs.S + bo + (size_t)L.slot_r * C * 64;
  const float* Sm = rs.S + bo + (size_t)L.slot_m * C * 64;
  const float* Um = rs.U + bo + (size_t)L.slot_m * C * 64;
  const size_t ap = ((size_t)b * 64 + L.r) * 64 + 8 * kq;       // element offset into the alpha planes
  const long apl = (long)gridDim.y * 4096;
  // column r of the image: chunk r>>3, element r&7
  unsigned short* t16 = reinterpret_cast<unsigned short*>(img);
  constexpr int RL = 16 * NC;                              // fp16 per image row
  constexpr int PLH = 64 * RL;                             // plane stride in fp16
  const int wchunk = 2 * tl + (l15 >> 3), we = l15 & 7;
  const u32x4* im4 = reinterpret_cast<const u32x4*>(img);
  constexpr int PL4 = 64 * CH;                             // plane stride in 16-byte units
  float score = 0.f;
  // the waves that share an image meet at WORKGROUP barriers (measured faster here than LDS-counter group
  // barriers): every wave runs the same number of iterations, the surplus ones on the last site with weight 0
  const int c_end = NG > 1 ? c0 + (c1 - c0 + NSLOT - 1) / NSLOT * NSLOT : c1;
  for (int c_ = c0 + slot; c_ < c_end; c_ += NSLOT) {
    asm volatile("" ::: "memory");
    const bool act = c_ < c1;
    const int c = act ? c_ : c1 - 1;
    const float mc = (!act || mask[(size_t)b * C + c]) ? 0.f : 1.f;      // seq_mask (model.py:96); first load of the iteration
    // no register prefetch of the next site: three waves per SIMD hide the row loads, and the 32 registers
    // would push the kernel over the 168 of that occupancy
    V64 sr;
    load_v64(sr, Sr + (size_t)c * 64, kq);
    // alpha[pair][r'] of this lane's pair, k-slot 8kq + j of k-step ks = r' = 32ks + 8kq + j (exactly 0 beyond
    // the live rows): issued first, it lands behind the gate and the image
    Frag3 al[KSX];                                               // pre-split by k_alpha_softmax
    if constexpr (CTX) {
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) alpha_frag16(al[ks], alpha, apl, ap + 32 * ks);
    }
    V64 x;
    // fp16 pieces of S_r: B operand of U_r = W_h S_r (recomputed: the cached U rows were half of the kernel's HBM
    // reads) and, transposed, the columns of the image
    Frag3 sf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) split_8<false>(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
    {
      V64 sm, um, ur;
      load_v64(sm, Sm + (size_t)c * 64, kq);
      load_v64(um, Um + (size_t)c * 64, kq);
      gate_init16(ur, um, cv, L.sgn, kq);
      linear_t16p_core<4>(ur.t, sf, Wh_l, lane, [] {});
      gate16(x, sr, ur, sm);
    }
    if constexpr (CTX) {
      if constexpr (NG > 1) __syncthreads();                     // everyone is done with the previous image
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const unsigned h = sf[mt >> 1].h[2 * (mt & 1) + pr], m = sf[mt >> 1].m[2 * (mt & 1) + pr];
          const int d0 = 16 * mt + 4 * kq + 2 * pr, d1 = d0 + 1;
          const int o0 = d0 * RL + 8 * tswz<NC>(d0, wchunk) + we, o1 = d1 * RL + 8 * tswz<NC>(d1, wchunk) + we;
          t16[o0] = (unsigned short)h; t16[o1] = (unsigned short)(h >> 16);
          t16[PLH + o0] = (unsigned short)m; t16[PLH + o1] = (unsigned short)(m >> 16);
          __builtin_amdgcn_sched_barrier(0);                     // bounded live ranges (three waves per SIMD: 168 registers)
        }
      if constexpr (NG > 1) __syncthreads();                     // all columns are in the image
    }
    if constexpr (CTX) {
      V64 xg, g;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xg.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      static_for<0, KSX>([&](auto ki) {
        constexpr int ks = decltype(ki)::value;
        const Frag3& bfr = al[ks];
        // lanes whose chunk lies beyond a short row read the row's last chunk instead: finite data against
        // alpha values that are exactly 0
        const int lc = min(4 * ks + kq, CH - 1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int d = 16 * mt + l15;
          const int o = d * CH + tswz<NC>(d, lc);
          Frag3 a;
          a.h = im4[o]; a.m = im4[PL4 + o];
          xg.t[mt] = mfma16_b6(a, bfr, xg.t[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      linear_t16p<4, false, true, 2, false>(g.t, xg, Wg_l, cv + 64, lane);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        gate_mix4(x.t[mt], xg.t[mt], g.t[mt]);          // (1-w)*x + w*x_g
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    V64 s1;
    linear_t16p<4, false, true, 2, false>(s1.t, x, S0_l, cv + 128, lane);
    f32x2v s2 = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 16 * mt + 4 * kq);
      gelu_dot4(s2, s1.t[mt], w4);
      __builtin_amdgcn_sched_barrier(0);
    }
    float s = s2[0] + s2[1];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    score += (s + w.s2b) * mc;
  }
  // one partial set per WORKGROUP: the slots' sums meet in LDS (the images are dead) and are added in slot order
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // [NSLOT][64]
  if (16 * NG < 64 && tl == 0) red[slot * 64 + lane] = 0.f;                 // pair rows without a wave
  __syncthreads();
  if (kq == 0) red[slot * 64 + L.r] = score;
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < NSLOT; ++s_) v += red[s_ * 64 + tid];
    score_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  }
}

// -DNNJ_STAMP: a DIAGNOSTIC build (never shipped: tools/stamp_run.py) that stamps the site loop of k_inc_score_w with
// s_memtime and adds the per-phase cycle counts of every wave into g_stamp (cdna_hip_programming.md section 7, "In-kernel
// stamps"): [0] iterations, [1] row loads until they have landed, [2] phase A (U_r, gate, image) of all tiles,
// [3 + t] phase B (x_g, W_g, mix, s_out, GELU) of tile t.
#ifdef NNJ_STAMP
__device__ unsigned long long g_stamp[8];
__device__ __forceinline__ unsigned long long stamp_now() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  return __builtin_amdgcn_s_memtime();
}
#endif
// ------------------------------------------------------------------ k_inc_score_w
// scores of the new pairs with ONE WAVE PER SITE: the wave walks the NT 16-pair tiles of its site itself, so the
// site image S^T [2 planes][64 d][16*NT r'] is private to the wave and nothing in the loop synchronises with
// another wave (k_inc_score16<3> pays two twelve-wave barriers per site).  Eight waves per workgroup (two per
// SIMD, <= 256 registers): the latency of one tile's chain is covered by the other tiles of the same wave --
// the phases below are loops over the tiles, i.e. NT independent chains -- and by the second wave.
// Used for 33..48 rows (NT = 3: 48 instead of 64 padded pairs).  part[b][sc][pair r].
template <int NT, bool CTX, int NW = 8>
__global__ __launch_bounds__(64 * NW) void k_inc_score_w(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                     const float* __restrict__ alpha,
                                                     const uint8_t* __restrict__ mask,
                                                     float* __restrict__ score_part, int n, int C, int cs, int qn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WPF = 2;                                   // fragment reads ahead
  constexpr int RL = 16 * NT;                              // fp16 per image row
  constexpr int CH = 2 * NT;                               // 16-byte chunks per image row (8 r' each)
  constexpr int KSX = (NT + 1) / 2;                        // k-steps of the x_g GEMM (32 r' each)
  constexpr int IMG = 64 * RL * NPL / 2;                   // floats of an image
  constexpr int PLH = 64 * RL;                             // plane stride in fp16
  constexpr int PL4 = 64 * CH;                             // plane stride in 16-byte units
  float* Wg_l = smem;
  float* S0_l = smem + IMG64;
  float* Wh_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* img = smem + 3 * IMG64 + wave * IMG;
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * NW);
  stage_image_t16(S0_l, w.imgS0, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  float* cv = smem + 3 * IMG64 + NW * IMG;
  stage_scorer_consts(cv, w, tid);
  // the alpha pieces of this alignment's pairs sit in LDS for the lifetime of the workgroup (two planes of
  // [16 NT pairs][64 r'] fp16, 16-byte chunks XOR-swizzled by the pair): read from L2 per tile and site they were
  // one exposed round trip in front of every tile's chain
  const long apl = (long)gridDim.y * 4096;
  constexpr int APL = 16 * NT * 64;                        // fp16 elements of a plane in LDS
  unsigned short* alds = reinterpret_cast<unsigned short*>(cv + SCORER_CONSTS);
  if constexpr (CTX) {
    const unsigned short* ag = reinterpret_cast<const unsigned short*>(alpha) + (size_t)b * 4096;
    for (int i = tid; i < 2 * 16 * NT * 8; i += 64 * NW) {             // 16-byte chunks of both planes
      const int pl = i / (16 * NT * 8), rc = i % (16 * NT * 8), r = rc >> 3, ch = rc & 7;
      *reinterpret_cast<u32x4*>(alds + pl * APL + r * 64 + 8 * (ch ^ (r & 7))) =
          *reinterpret_cast<const u32x4*>(ag + pl * apl + r * 64 + 8 * ch);
    }
  }
  __syncthreads();
  const size_t bo = (size_t)b * rs.bstride;
  const float* Sr[NT];
  float sgn[NT], score[NT];
  int rr[NT];
  const float *Sm, *Um;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const Inc16 L = inc16(rs, ij_prev, b, n, 16 * t + l15, qn);
    Sr[t] = rs.S + bo + (size_t)L.slot_r * C * 64;
    sgn[t] = L.sgn;
    rr[t] = L.r;
    score[t] = 0.f;
    if (t == 0) {
      Sm = rs.S + bo + (size_t)L.slot_m * C * 64;
      Um = rs.U + bo + (size_t)L.slot_m * C * 64;
    }
  }
  // chunk swizzle of the image rows: none needed for 96-byte rows (see below), tswz for 32/64/128-byte rows
  auto wsw = [](int d, int chunk) { if constexpr (NT == 3) return chunk; else return tswz<NT>(d, chunk); };
  unsigned short* t16 = reinterpret_cast<unsigned short*>(img);
  const u32x4* im4 = reinterpret_cast<const u32x4*>(img);
#ifdef NNJ_STAMP
  unsigned long long st_n = 0, st_load = 0, st_a = 0, st_b[4] = {0, 0, 0, 0};
#endif
  // Round 5: the rows of the NEXT site are requested as soon as the gate phase has consumed the current ones -- the
  // registers are dead from there on -- so their HBM latency runs behind the chain phase (64 % of a site's cycles)
  // instead of in front of the gate phase (stamps of round 4: 11 % of a site's cycles passed until the loads had landed).
  V64 sr[NT], sm, um;
  unsigned mraw = 0;
  auto request = [&](int c) {
    mraw = mask[(size_t)b * C + c];                                     // seq_mask (model.py:96)
#pragma unroll
    for (int t = 0; t < NT; ++t) load_v64(sr[t], Sr[t] + (size_t)c * 64, kq);
    load_v64(sm, Sm + (size_t)c * 64, kq);
    load_v64(um, Um + (size_t)c * 64, kq);
  };
  if (c0 + wave < c1) request(c0 + wave);
  for (int c = c0 + wave; c < c1; c += NW) {
    asm volatile("" ::: "memory");
#ifdef NNJ_STAMP
    const unsigned long long T0 = stamp_now();
#endif
    const float mc = mraw ? 0.f : 1.f;
    V64 x[NT];
    {
#ifdef NNJ_STAMP
      const unsigned long long T1 = stamp_now();
      st_load += T1 - T0; ++st_n;
#endif
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        // fp16 pieces of S_r: B operand of U_r = W_h S_r (recomputed, see k_inc_alpha16) and, transposed, the
        // columns 16t + l15 of the image
        Frag3 sf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) split_8<false>(sf[ks], sr[t].t[2 * ks], sr[t].t[2 * ks + 1]);
        V64 ur;
        // (U_r read from the row store instead -- 24 MFMAs and 16 fragment reads less, one more row of traffic per tile --
        // measured equal to slower: scores 48.4 -> 48.7 ms per rollout with only the 41..48 tier changed)
        gate_init16(ur, um, cv, sgn[t], kq);
        linear_t16p_core<4, WPF>(ur.t, sf, Wh_l, lane, [] {});
        gate16(x[t], sr[t], ur, sm);
        if constexpr (CTX) {
          const int wchunk = 2 * t + (l15 >> 3), we = l15 & 7;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const unsigned h = sf[mt >> 1].h[2 * (mt & 1) + pr], m = sf[mt >> 1].m[2 * (mt & 1) + pr];
              const int d0 = 16 * mt + 4 * kq + 2 * pr, d1 = d0 + 1;
              const int o0 = d0 * RL + 8 * wsw(d0, wchunk) + we, o1 = d1 * RL + 8 * wsw(d1, wchunk) + we;
              t16[o0] = (unsigned short)h; t16[o1] = (unsigned short)(h >> 16);
              t16[PLH + o0] = (unsigned short)m; t16[PLH + o1] = (unsigned short)(m >> 16);
            }
        }
        // bounded live ranges: one tile's pieces at a time, and the weight fragments are re-read per tile (kept
        // across the tiles they would take 64 registers per matrix)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    request(c + NW < c1 ? c + NW : c);                      // (last site: a harmless reload; unconditional: no branch)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#ifdef NNJ_STAMP
    unsigned long long TP = stamp_now();
    st_a += TP - T0;                                        // (includes the loads: [2] - [1] is phase A proper)
#endif
    // x_g^T = S^T alpha^T, W_g, mix, s_out per tile.  Image rows are 16*NT fp16 (96 bytes at NT = 3: rows d and
    // d+8 share banks, the 16 lanes of a read group hold 8 distinct d per chunk parity -- conflict free without
    // a swizzle).  Lanes whose chunk lies beyond a short row read the row's last chunk: finite data against
    // alpha = 0.
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if constexpr (CTX) {
        V64 xg, g;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xg.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
          Frag3 bfr;
          {
            const int ar = 16 * t + l15;
            const unsigned short* ap_ = alds + ar * 64 + 8 * ((4 * ks + kq) ^ (ar & 7));
            bfr.h = *reinterpret_cast<const u32x4*>(ap_);
            bfr.m = *reinterpret_cast<const u32x4*>(ap_ + APL);
          }
          const int lc = min(4 * ks + kq, CH - 1);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int d = 16 * mt + l15;
            const int o = d * CH + wsw(d, lc);
            Frag3 a;
            a.h = im4[o]; a.m = im4[PL4 + o];
            xg.t[mt] = mfma16_b6(a, bfr, xg.t[mt]);
          }
        }
        linear_t16p<4, false, true, WPF, false>(g.t, xg, Wg_l, cv + 64, lane);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gate_mix4(x[t].t[mt], xg.t[mt], g.t[mt]);          // (1-w)*x + w*x_g
      }
      V64 s1;
      linear_t16p<4, false, true, WPF, false>(s1.t, x[t], S0_l, cv + 128, lane);
      f32x2v s2 = {0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 16 * mt + 4 * kq);
        gelu_dot4(s2, s1.t[mt], w4);
      }
      float s = s2[0] + s2[1];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      score[t] += (s + w.s2b) * mc;
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#ifdef NNJ_STAMP
      { const unsigned long long TN = stamp_now(); st_b[t < 4 ? t : 3] += TN - TP; TP = TN; }
#endif
    }
  }
#ifdef NNJ_STAMP
  if (lane == 0) {
    atomicAdd(&g_stamp[0], st_n); atomicAdd(&g_stamp[1], st_load); atomicAdd(&g_stamp[2], st_a);
    for (int t = 0; t < NT && t < 4; ++t) atomicAdd(&g_stamp[3 + t], st_b[t]);
  }
#endif
  // one partial set per WORKGROUP: the waves' sums meet in LDS (the images are dead) and are added in wave order
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // [NW][64]
  red[wave * 64 + lane] = 0.f;                             // pair rows without a tile
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) red[wave * 64 + rr[t]] = score[t];
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < NW; ++s_) v += red[s_ * 64 + tid];
    score_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  }
}

// ------------------------------------------------------------------ k_inc_score_wi
// k_inc_score_w with the tiles of a site walked STAGE BY STAGE in groups of NI instead of tile by tile: each of the
// four GEMMs of the chain (U_r, x_g, W_g, s_out.0) runs over the NI tiles of a group at once (linear_t16p_multi), so a
// weight or image fragment is read from LDS once per group instead of once per tile and the matrix pipe always has NI
// independent accumulator chains; the vector stages between them (split, gate, mix, GELU) are NI independent streams
// as well.  NI = NT = 3 does not fit 256 registers (45 spilled, 36 ms per rollout against 27): three tiles run as a
// group of two and a single.  Same image, same operation order per accumulator: results are bit-identical to
// k_inc_score_w.
template <int NT, int NI = 2, int NW = 8>
__global__ __launch_bounds__(64 * NW) void k_inc_score_wi(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                      const float* __restrict__ alpha,
                                                      const uint8_t* __restrict__ mask,
                                                      float* __restrict__ score_part, int n, int C, int cs, int qn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WPF = 2;
  constexpr int RL = 16 * NT;
  constexpr int CH = 2 * NT;
  constexpr int KSX = (NT + 1) / 2;
  constexpr int IMG = 64 * RL * NPL / 2;
  constexpr int PLH = 64 * RL;
  constexpr int PL4 = 64 * CH;
  constexpr int NGRP = (NT + NI - 1) / NI;
  float* Wg_l = smem;
  float* S0_l = smem + IMG64;
  float* Wh_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* img = smem + 3 * IMG64 + wave * IMG;
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * NW);
  stage_image_t16(S0_l, w.imgS0, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  float* cv = smem + 3 * IMG64 + NW * IMG;
  stage_scorer_consts(cv, w, tid);
  const long apl = (long)gridDim.y * 4096;
  constexpr int APL = 16 * NT * 64;
  unsigned short* alds = reinterpret_cast<unsigned short*>(cv + SCORER_CONSTS);
  {
    const unsigned short* ag = reinterpret_cast<const unsigned short*>(alpha) + (size_t)b * 4096;
    for (int i = tid; i < 2 * 16 * NT * 8; i += 64 * NW) {
      const int pl = i / (16 * NT * 8), rc = i % (16 * NT * 8), r = rc >> 3, ch = rc & 7;
      *reinterpret_cast<u32x4*>(alds + pl * APL + r * 64 + 8 * (ch ^ (r & 7))) =
          *reinterpret_cast<const u32x4*>(ag + pl * apl + r * 64 + 8 * ch);
    }
  }
  __syncthreads();
  const size_t bo = (size_t)b * rs.bstride;
  const float* Sr[NT];
  float sgn[NT], score[NT];
  int rr[NT];
  const float *Sm, *Um;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const Inc16 L = inc16(rs, ij_prev, b, n, 16 * t + l15, qn);
    Sr[t] = rs.S + bo + (size_t)L.slot_r * C * 64;
    sgn[t] = L.sgn;
    rr[t] = L.r;
    score[t] = 0.f;
    if (t == 0) {
      Sm = rs.S + bo + (size_t)L.slot_m * C * 64;
      Um = rs.U + bo + (size_t)L.slot_m * C * 64;
    }
  }
  auto wsw = [](int d, int chunk) { if constexpr (NT == 3) return chunk; else return tswz<NT>(d, chunk); };
  unsigned short* t16 = reinterpret_cast<unsigned short*>(img);
  const u32x4* im4 = reinterpret_cast<const u32x4*>(img);
  // the rows of the next site are requested behind the gate phase of the current one (see k_inc_score_w)
  V64 srall[NT], sm, um;
  unsigned mraw = 0;
  auto request = [&](int c) {
    mraw = mask[(size_t)b * C + c];
#pragma unroll
    for (int t = 0; t < NT; ++t) load_v64(srall[t], Sr[t] + (size_t)c * 64, kq);
    load_v64(sm, Sm + (size_t)c * 64, kq);
    load_v64(um, Um + (size_t)c * 64, kq);
  };
  if (c0 + wave < c1) request(c0 + wave);
  for (int c = c0 + wave; c < c1; c += NW) {
    asm volatile("" ::: "memory");
    const float mc = mraw ? 0.f : 1.f;
    V64 x[NT];
    {
      static_for<0, NGRP>([&](auto gi) {
        constexpr int T0 = decltype(gi)::value * NI, N = (NT - T0 < NI ? NT - T0 : NI);
        V64 ur[N];
        Frag3 sf[N][2];
#pragma unroll
        for (int u = 0; u < N; ++u) gate_init16(ur[u], um, cv, sgn[T0 + u], kq);
        linear_t16p_multi<N, WPF>(ur, sf, Wh_l, lane, [&](auto ki) {
          constexpr int ks = decltype(ki)::value;
#pragma unroll
          for (int u = 0; u < N; ++u) split_8<false>(sf[u][ks], srall[T0 + u].t[2 * ks], srall[T0 + u].t[2 * ks + 1]);
        });
#pragma unroll
        for (int u = 0; u < N; ++u) {
          const int wchunk = 2 * (T0 + u) + (l15 >> 3), we = l15 & 7;
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const unsigned h = sf[u][mt >> 1].h[2 * (mt & 1) + pr], m = sf[u][mt >> 1].m[2 * (mt & 1) + pr];
              const int d0 = 16 * mt + 4 * kq + 2 * pr, d1 = d0 + 1;
              const int o0 = d0 * RL + 8 * wsw(d0, wchunk) + we, o1 = d1 * RL + 8 * wsw(d1, wchunk) + we;
              t16[o0] = (unsigned short)h; t16[o1] = (unsigned short)(h >> 16);
              t16[PLH + o0] = (unsigned short)m; t16[PLH + o1] = (unsigned short)(m >> 16);
            }
        }
#pragma unroll
        for (int u = 0; u < N; ++u) gate16(x[T0 + u], srall[T0 + u], ur[u], sm);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      });
    }
    request(c + NW < c1 ? c + NW : c);                      // (last site: a harmless reload; unconditional: no branch)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NGRP>([&](auto gi) {
      constexpr int T0 = decltype(gi)::value * NI, N = (NT - T0 < NI ? NT - T0 : NI);
      // x_g^T = S^T alpha^T of the group's tiles: one image fragment per (k-step, d tile), N alpha fragments per k-step
      V64 xg[N];
#pragma unroll
      for (int u = 0; u < N; ++u)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xg[u].t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) {
        Frag3 bfr[N];
#pragma unroll
        for (int u = 0; u < N; ++u) {
          const int ar = 16 * (T0 + u) + l15;
          const unsigned short* ap_ = alds + ar * 64 + 8 * ((4 * ks + kq) ^ (ar & 7));
          bfr[u].h = *reinterpret_cast<const u32x4*>(ap_);
          bfr[u].m = *reinterpret_cast<const u32x4*>(ap_ + APL);
        }
        const int lc = min(4 * ks + kq, CH - 1);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int d = 16 * mt + l15;
          const int o = d * CH + wsw(d, lc);
          Frag3 a;
          a.h = im4[o]; a.m = im4[PL4 + o];
#pragma unroll
          for (int u = 0; u < N; ++u) xg[u].t[mt] = mfma16_f16(a.m, bfr[u].h, xg[u].t[mt]);
#pragma unroll
          for (int u = 0; u < N; ++u) xg[u].t[mt] = mfma16_f16(a.h, bfr[u].m, xg[u].t[mt]);
#pragma unroll
          for (int u = 0; u < N; ++u) xg[u].t[mt] = mfma16_f16(a.h, bfr[u].h, xg[u].t[mt]);
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      {
        V64 g[N];
        Frag3 bx[N][2];
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) g[u].t[mt] = *reinterpret_cast<const f32x4*>(cv + 64 + 16 * mt + 4 * kq);
        linear_t16p_multi<N, WPF>(g, bx, Wg_l, lane, [&](auto ki) {
          constexpr int ks = decltype(ki)::value;
#pragma unroll
          for (int u = 0; u < N; ++u) split_8<false>(bx[u][ks], xg[u].t[2 * ks], xg[u].t[2 * ks + 1]);
        });
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) gate_mix4(x[T0 + u].t[mt], xg[u].t[mt], g[u].t[mt]);          // (1-w)*x + w*x_g
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      {
        V64 s1[N];
        Frag3 bx[N][2];
#pragma unroll
        for (int u = 0; u < N; ++u)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) s1[u].t[mt] = *reinterpret_cast<const f32x4*>(cv + 128 + 16 * mt + 4 * kq);
        linear_t16p_multi<N, WPF>(s1, bx, S0_l, lane, [&](auto ki) {
          constexpr int ks = decltype(ki)::value;
#pragma unroll
          for (int u = 0; u < N; ++u) split_8<false>(bx[u][ks], x[T0 + u].t[2 * ks], x[T0 + u].t[2 * ks + 1]);
        });
#pragma unroll
        for (int u = 0; u < N; ++u) {
          f32x2v s2 = {0.f, 0.f};
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 16 * mt + 4 * kq);
            gelu_dot4(s2, s1[u].t[mt], w4);
          }
          float s = s2[0] + s2[1];
          s += __shfl_xor(s, 16);
          s += __shfl_xor(s, 32);
          score[T0 + u] += (s + w.s2b) * mc;
        }
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  __syncthreads();
  float* red = smem + 3 * IMG64;
  red[wave * 64 + lane] = 0.f;
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) red[wave * 64 + rr[t]] = score[t];
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < NW; ++s_) v += red[s_ * 64 + tid];
    score_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  }
}
