// nnj_scorer.hpp -- gfx950 kernels of the neural NJ loop: pair scorer, merged-row
// aggregate, score-table assemble + argmax (restates reference model.py:90-209,
// environment.py:760-835, utils.py:213-251, finetune_rl_search.py:140-160).
//
// Algebra (exact in real arithmetic, fp32 rounding differs from the reference by O(eps)):
//   h   = W_h (x_i - x_j) + b_h          = U_i - U_j + b_h,          U_r  = W_h S_r
//   q.k = (W_q x + b_q).(W_k S_r + b_k)  = x . K'_r + beta_r,        K'_r = A S_r + a0,
//         A = W_q^T W_k, a0 = W_q^T b_k, beta_r = u . sum_c S_r[c] + C * t0,
//         u = W_k^T b_q, t0 = b_q . b_k
// so the per-pair W_h and W_q GEMMs of the reference become per-ROW transforms that are
// cached next to each row (U, K', beta partials) and computed once when a row is born.
//
// Rows live in slots: S/U/Kp [B][slots][C][64]; `live[b][p]` maps the reference's row
// position p (0..n-1, sorted) to a slot; a merge writes the new row into slot(live[i])
// and deletes position j (reference environment.py:764-768).
#pragma once
#include "nnj_common.hpp"

struct ScorerW {
  const float *Wh, *bh;        // h_linear_last (weight [64][64], bias)
  const float *Wg, *bg;        // g_linear_last
  const float *A, *a0, *u;     // derived: W_q^T W_k, W_q^T b_k, W_k^T b_q
  float t0;                    // b_q . b_k
  const float *S0, *s0, *s2w;  // s_out.0 weight/bias, s_out.2 weight
  float s2b;                   // s_out.2 bias
  // round 4: the four 64 x 64 operands of the 16-token kernels as ready LDS images (stage_weight_t16's layout: two fp16
  // planes, swizzled), built ONCE on the device when the weights are loaded (k_build_scorer_images): a workgroup copies
  // 16 KiB per matrix instead of loading, splitting and storing it -- at one alignment per rollout the split was a
  // quarter of every NJ-step kernel.  imgAt = the image of A^T.
  const float *imgAt, *imgWh, *imgWg, *imgS0;
};
// copy of a ready [64][64] operand image (IMG64 floats) into LDS
__device__ __forceinline__ void stage_image_t16(float* lds, const float* __restrict__ gimg, int tid, int nthreads) {
  const f32x4* g4 = reinterpret_cast<const f32x4*>(gimg);
  f32x4* l4 = reinterpret_cast<f32x4*>(lds);
  for (int i = tid; i < IMG64 / 4; i += nthreads) l4[i] = g4[i];
}
// the images themselves (one workgroup; `out` = 4 x IMG64 floats: A^T | W_h | W_g | S0)
#ifndef NNJ_STEP0_TU           // (nnj_step0_tu.hip includes this file for k_pair_alpha and k_pair_score alone)
__global__ void k_build_scorer_images(ScorerW w, float* __restrict__ out) {
  stage_weight_t16(out, w.A, 64, threadIdx.x, blockDim.x, true);
  stage_weight_t16(out + IMG64, w.Wh, 64, threadIdx.x, blockDim.x);
  stage_weight_t16(out + 2 * IMG64, w.Wg, 64, threadIdx.x, blockDim.x);
  stage_weight_t16(out + 3 * IMG64, w.S0, 64, threadIdx.x, blockDim.x);
}
#endif

struct RowSet {                // where the rows of one call live
  const float* S;              // [B][slots][C][64]
  const float* U;
  const float* Kp;
  const float* beta_part;      // [B][slots][ntile32]
  long bstride;                // floats between batch elements (slots*C*64)
  const int* live;             // [B][live_stride] position -> slot, or nullptr = identity
  int live_stride;
  int ntile32;                 // beta partials per row (>= ceil(C/32); entries not written by the row's producer are 0)
};

__device__ __forceinline__ int slot_of(const RowSet& rs, int b, int pos) {
  return rs.live ? rs.live[b * rs.live_stride + pos] : pos;
}
// The two-pass step (nnj_step2.hpp) numbers the n-1 rows OTHER than the freshly merged row m as q = 0..n-2 (pairs,
// image rows, alpha planes, score partials); position in the list:
__device__ __forceinline__ int q_to_r(int q, int m) { return q + (q >= m ? 1 : 0); }

// The per-feature vectors of the scorer (b_h, b_g, s_out.0 bias, s_out.2 weight: 4 x 64 floats) live in LDS for the
// lifetime of a workgroup: read from global memory inside the site loop they were one L2 round trip each, several of
// them serialised by the scheduling fences that bound the live ranges (k_inc_score16: four `s_waitcnt vmcnt(0)` per
// site on the s_out.2 weights alone), and their 64-bit addresses cost registers (k_inc_score_w<3>: 256 registers + 92
// bytes of scratch -> 247 and none; incremental scores 72.5 -> 63.6 ms, alpha 34.6 -> 31.7 ms per rollout).
constexpr int SCORER_CONSTS = 256;
__device__ __forceinline__ void stage_scorer_consts(float* cv, const ScorerW& w, int tid) {
  if (tid < 64) { cv[tid] = w.bh[tid]; cv[64 + tid] = w.bg[tid]; cv[128 + tid] = w.s0[tid]; cv[192 + tid] = w.s2w[tid]; }
}

enum { PAIRS_FULL = 0, PAIRS_INCR = 1 };

// pair -> (position i, position j); returns false for padding lanes / the self pair
__device__ __forceinline__ bool pair_of(int mode, int n, int p, int npairs, const int* ij_prev, int b,
                                        int& pi, int& pj) {
  pi = 0; pj = 1;
  if (p >= npairs) return false;
  if (mode == PAIRS_FULL) { pair_from_index(n, p, pi, pj); return true; }
  const int ip = min(max(ij_prev[2 * b], 0), n - 1);
  if (p == ip) return false;             // the reference scores (i,i) but never reads it (model.py:186-197)
  pi = p < ip ? p : ip;
  pj = p < ip ? ip : p;
  return true;
}

// ---- per-site images of all n rows in LDS (swizzled [64][64] images), staged through registers
// in two halves (T14 split): site_load issues the global loads of the NEXT site before the MFMAs of
// the current one, site_store writes them to the other LDS buffer afterwards.
//   img_a <- A-operand source rows (K' for phase A), img_s <- S, img_u <- U,
//   img_t <- S transposed [d][r] (phase B)
template <int NTHR>
struct SiteRegs { f32x4 a[1024 / NTHR], s[1024 / NTHR], u[1024 / NTHR]; };
// where this thread's pieces of the site images come from: site invariant, looked up ONCE per workgroup (inside
// site_load the live-list lookup was a dependent L2 round trip in front of every row load)
template <int NTHR>
struct SiteOff { size_t base[1024 / NTHR]; };
template <int NTHR>
__device__ __forceinline__ void site_offsets(SiteOff<NTHR>& O, const RowSet& rs, int b, int n, int C, int tid) {
#pragma unroll
  for (int k = 0; k < 1024 / NTHR; ++k) {
    const int i = tid + NTHR * k;
    const int r = i >> 4, ch = i & 15;
    O.base[k] = (size_t)b * rs.bstride + (size_t)slot_of(rs, b, r < n ? r : 0) * C * 64 + 4 * ch;
  }
}
template <bool WITH_A, int NTHR>
__device__ __forceinline__ void site_load(SiteRegs<NTHR>& R, const SiteOff<NTHR>& O, const RowSet& rs, int n, int c,
                                          const float* srcA, int tid) {
  // loads only: the values are not touched here (rows >= n read row 0, always there; they are zeroed in site_store).
  // A select on the loaded value at this point is a USE: the compiler then waits for the loads right here and the
  // next site's rows are not in flight behind the MFMAs of the current one at all (it did: `s_waitcnt vmcnt(0)`
  // straight after the loads in the ISA of both step-0 kernels).
#pragma unroll
  for (int k = 0; k < 1024 / NTHR; ++k) {
    const size_t off = O.base[k] + (size_t)c * 64;
    R.s[k] = *reinterpret_cast<const f32x4*>(rs.S + off);
    R.u[k] = *reinterpret_cast<const f32x4*>(rs.U + off);
    if (WITH_A) R.a[k] = *reinterpret_cast<const f32x4*>(srcA + off);
  }
}
template <bool WITH_A, bool WITH_T, int NTHR>
__device__ __forceinline__ void site_store(const SiteRegs<NTHR>& R, float* img_a, float* img_s, float* img_u,
                                           float* img_t, int tid, int n) {
#pragma unroll
  for (int k = 0; k < 1024 / NTHR; ++k) {
    const int i = tid + NTHR * k;
    const int r = i >> 4, ch = i & 15;
    const int sw = 4 * wswz(r, ch);
    const bool live = r < n;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    const f32x4 vs = live ? R.s[k] : z, vu = live ? R.u[k] : z;
    *reinterpret_cast<f32x4*>(img_s + r * 64 + sw) = vs;
    *reinterpret_cast<f32x4*>(img_u + r * 64 + sw) = vu;
    if (WITH_A) {
      const f32x4 va = live ? R.a[k] : z;
      // K' as an f16x3 A image [2 planes][64 rows r][64 d] (the layout of stage_weight_b6): this thread's four
      // features d = 4ch..4ch+3 are one half of chunk q = 4*kt + 2*G + hh  (d = 32kt + 16G + 8j + 4hh + t)
      const int q = 4 * (ch >> 3) + 2 * ((ch >> 2) & 1) + (ch & 1), j = (ch >> 1) & 1;
      unsigned h01, m01, h23, m23;
      split2(va[0], va[1], h01, m01);
      split2(va[2], va[3], h23, m23);
      uint2* img = reinterpret_cast<uint2*>(img_a) + (r * 8 + wswz6<8>(r, q)) * 2 + j;   // 8-byte units
      img[0] = make_uint2(h01, h23);
      img[64 * 8 * 2] = make_uint2(m01, m23);
    }
    if (WITH_T) {
      // S^T as an f16x3 A image: [2 planes][64 d][64 r'] fp16, row d = the r' order of stage_weight_b6
      unsigned short* t16 = reinterpret_cast<unsigned short*>(img_t);
      const int q = 4 * (r >> 5) + 2 * ((r >> 4) & 1) + ((r >> 2) & 1), e = 4 * ((r >> 3) & 1) + (r & 3);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        unsigned h, m;
        split2(vs[2 * pr], vs[2 * pr + 1], h, m);
        const int d0 = 4 * ch + 2 * pr, d1 = d0 + 1;
        const int o0 = d0 * 64 + 8 * wswz6<8>(d0, q) + e, o1 = d1 * 64 + 8 * wswz6<8>(d1, q) + e;
        t16[o0] = (unsigned short)h; t16[o1] = (unsigned short)(h >> 16);
        t16[4096 + o0] = (unsigned short)m; t16[4096 + o1] = (unsigned short)(m >> 16);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
}

// gate: x = z*x_i + (1-z)*x_j, z = sigmoid(U_i - U_j + b_h)  (model.py:105-108)
__device__ __forceinline__ void gate_tile(f32x16 (&x)[2], const float* img_s, const float* img_u,
                                          const float* bh, int pi, int pj, int hh) {
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = 2 * (4 * mt + g) + hh;
      const f32x4 si = *reinterpret_cast<const f32x4*>(img_s + pi * 64 + 4 * wswz(pi, ch));
      const f32x4 sj = *reinterpret_cast<const f32x4*>(img_s + pj * 64 + 4 * wswz(pj, ch));
      const f32x4 ui = *reinterpret_cast<const f32x4*>(img_u + pi * 64 + 4 * wswz(pi, ch));
      const f32x4 uj = *reinterpret_cast<const f32x4*>(img_u + pj * 64 + 4 * wswz(pj, ch));
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bh + 4 * ch);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float z = sigmoid_l2(ui[t] - uj[t] + b4[t]);
        x[mt][4 * g + t] = sj[t] + z * (si[t] - sj[t]);     // z*x_i + (1-z)*x_j
      }
      if (g & 1) __builtin_amdgcn_sched_barrier(0);         // at most 8 row pieces in flight: bounded live range
    }
}

// ------------------------------------------------------------------ k_pair_alpha
// Workgroup -> (site chunk, pair group, batch element) of the all-pairs kernels.  The pair groups of one
// (site chunk, b) read the same rows: they get consecutive slots of ONE XCD (blockIdx % 8 = the XCD of the
// round-robin dispatch), so they run together and the rows they share are served by that XCD's L2 instead of
// being fetched once per pair group.  false: padding workgroup, nothing to do.
__device__ __forceinline__ bool pair_block(int nsc, int npg, int B, int& sc, int& pg, int& b) {
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int grp = (slot / npg) * 8 + xcd;
  pg = slot % npg;
  sc = grp % nsc;
  b = grp / nsc;
  return grp < nsc * B;
}
__host__ inline unsigned pair_grid(int nsc, int npg, int B) { return (unsigned)(((nsc * B + 7) / 8) * 8 * npg); }

// Phase A: alpha_part[b][sc][pair][r] = sum_{c in chunk sc} x_pair[c,:] . K'_r[c,:]
// grid (nsc, pair groups, B); 4 waves x TPW tiles of 32 pairs.
template <int TPW, int NW>
__global__ __launch_bounds__(64 * NW) void k_pair_alpha(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                    float* __restrict__ alpha_part, int mode, int n, int C,
                                                    int npairs, int ppad, int cs, int nsc, int npg, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // 2 x [img_k (f16x3) | img_s | img_u]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  int sc, pg, b;
  if (!pair_block(nsc, npg, B, sc, pg, b)) return;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  float* cv = smem + 2 * (IMG64 + 8192);
  stage_scorer_consts(cv, w, tid);
  int pi[TPW], pj[TPW];
  bool any = false;
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int p = ((pg * NW + wave) * TPW + tt) * 32 + (lane & 31);
    pair_of(mode, n, p, npairs, ij_prev, b, pi[tt], pj[tt]);
    any = any || (((pg * NW + wave) * TPW + tt) * 32 < npairs);
  }
  f32x16 acc[TPW][1][2];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[tt][0][mt][r] = 0.f;
  SiteRegs<64 * NW> R;
  SiteOff<64 * NW> RO;
  site_offsets<64 * NW>(RO, rs, b, n, C, tid);
  if (c0 < c1) {
    site_load<true, 64 * NW>(R, RO, rs, n, c0, rs.Kp, tid);
    site_store<true, false, 64 * NW>(R, smem, smem + IMG64, smem + IMG64 + 4096, nullptr, tid, n);
  }
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    float* cur = smem + ((c - c0) & 1) * (IMG64 + 8192);
    float* nxt = smem + (((c - c0) & 1) ^ 1) * (IMG64 + 8192);
    const bool more = c + 1 < c1;
    if (more) site_load<true, 64 * NW>(R, RO, rs, n, c + 1, rs.Kp, tid);       // in flight behind the MFMAs
    if (any) {
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        f32x16 x[1][2];
        gate_tile(x[0], cur + IMG64, cur + IMG64 + 4096, cv, pi[tt], pj[tt], hh);
        linear6_T_acc<2, 2, 1, true>(acc[tt], x, cur, lane);
      }
    }
    if (more) site_store<true, false, 64 * NW>(R, nxt, nxt + IMG64, nxt + IMG64 + 4096, nullptr, tid, n);
    __syncthreads();
  }
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int p = ((pg * NW + wave) * TPW + tt) * 32 + (lane & 31);
    if (p < ppad) {
      float* dst = alpha_part + (((size_t)b * nsc + sc) * ppad + p) * 64;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = {acc[tt][0][mt][4 * g], acc[tt][0][mt][4 * g + 1], acc[tt][0][mt][4 * g + 2],
                     acc[tt][0][mt][4 * g + 3]};
          *reinterpret_cast<f32x4*>(dst + 32 * mt + 8 * g + 4 * hh) = v;
        }
    }
  }
}

#ifndef NNJ_STEP0_TU           // nnj_api.hip only
// ------------------------------------------------------------------ k_alpha_softmax
// alpha[b][pair][r] = softmax_r( (sum_sc part + beta_r) / sqrt(64*C) ), rows i, j of the
// pair and r >= n excluded (model.py:118-146).  One wave per pair, lane = r.
__global__ __launch_bounds__(256) void k_alpha_softmax(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                       const float* __restrict__ alpha_part,
                                                       float* __restrict__ alpha, int mode, int n, int C,
                                                       int npairs, int ppad, int nsc,
                                                       const float* __restrict__ beta_slot, int nslot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (p >= ppad) return;
  int pi, pj;
  const bool valid = pair_of(mode, n, p, npairs, ij_prev, b, pi, pj);
  float a = 0.f;
#pragma unroll 8          // independent loads in flight; the additions stay in order
  for (int sc = 0; sc < nsc; ++sc) a += alpha_part[(((size_t)b * nsc + sc) * ppad + p) * 64 + lane];
  // per-row bias of the logits: summed once per row by k_beta_sum (every pair's wave used to add the row's C / 32 partials
  // itself: four times the loads of the logits' own partials at 4096 sites)
  const float beta = lane < n ? beta_slot[(size_t)b * nslot + slot_of(rs, b, lane)] : 0.f;
  a = (a + beta) * (1.0f / sqrtf(64.0f * (float)C));
  const bool in = valid && lane < n && lane != pi && lane != pj;
  float v = in ? a : -INFINITY;
  float mx = v;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float e = in ? expf(v - mx) : 0.f;
  float s = e;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  // written as its two fp16 pieces (planes): the B operand of x_g = S^T alpha^T as the consumers load it
  alpha_store(alpha, (long)gridDim.y * ppad * 64, ((size_t)b * ppad + p) * 64 + lane, (s > 0.f) ? e / s : 0.f);
}

#endif  // NNJ_STEP0_TU
// ------------------------------------------------------------------ k_pair_score
// Phase B: per (pair, site): x_g = sum_r alpha_r S_r ; g = W_g x_g + b_g ; w = sigmoid(g);
// x = (1-w) x + w x_g ; s = s_out(x) ; score_part[b][sc][pair] = sum_{c in chunk} mask_c s
// (model.py:148-153, 93-97).  has_ctx = (n > 2) (model.py:111).
template <int TPW, int NW>
__global__ __launch_bounds__(64 * NW) void k_pair_score(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                    const float* __restrict__ alpha,
                                                    const uint8_t* __restrict__ mask,
                                                    float* __restrict__ score_part, int mode, int n, int C,
                                                    int npairs, int ppad, int cs, int has_ctx, int nsc, int npg, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // Wg | S0 (f16x3 images) | 2 x [img_t | img_s | img_u]
  float* Wg_l = smem;
  float* S0_l = smem + b6_floats(64, 64);
  float* ring = smem + 2 * b6_floats(64, 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  int sc, pg, b;
  if (!pair_block(nsc, npg, B, sc, pg, b)) return;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_weight_b6<64>(Wg_l, w.Wg, 64, tid, 64 * NW);
  stage_weight_b6<64>(S0_l, w.S0, 64, tid, 64 * NW);
  float* cv = smem + 2 * IMG64 + 2 * (IMG64 + 8192);
  stage_scorer_consts(cv, w, tid);
  int pi[TPW], pj[TPW];
  bool any = false;
  Frag3 af[TPW][4];                                        // alpha of the lane's pair, as k-step fragments (pre-split)
  float score[TPW];
#pragma unroll
  for (int tt = 0; tt < TPW; ++tt) {
    const int tile = (pg * NW + wave) * TPW + tt;
    const int p = tile * 32 + (lane & 31);
    pair_of(mode, n, p, npairs, ij_prev, b, pi[tt], pj[tt]);
    any = any || (tile * 32 < npairs);
    score[tt] = 0.f;
    const bool ld = has_ctx && p < ppad;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      alpha_frag32(af[tt][ks], alpha, (long)B * ppad * 64, ((size_t)b * ppad + (ld ? p : 0)) * 64, ks, hh, ld);
  }
  SiteRegs<64 * NW> R;
  SiteOff<64 * NW> RO;
  site_offsets<64 * NW>(RO, rs, b, n, C, tid);
  if (c0 < c1) {
    site_load<false, 64 * NW>(R, RO, rs, n, c0, nullptr, tid);
    site_store<false, true, 64 * NW>(R, nullptr, ring + IMG64, ring + IMG64 + 4096, ring, tid, n);
  }
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    float* cur = ring + ((c - c0) & 1) * (IMG64 + 8192);
    float* nxt = ring + (((c - c0) & 1) ^ 1) * (IMG64 + 8192);
    const float* img_t = cur;                          // operand image (IMG64 floats)
    const float* img_s = cur + IMG64;
    const float* img_u = cur + IMG64 + 4096;
    const bool more = c + 1 < c1;
    // seq_mask (model.py:96).  Loaded BEFORE the next site's rows are requested: vector-memory waits are in order, a
    // byte fetched after them would wait for all of them and the prefetch would overlap nothing.  `mask` is never
    // null here (the launcher substitutes a zero-filled buffer): no branch, no basic block of its own.
    const float mc = mask[(size_t)b * C + c] ? 0.f : 1.f;
    if (more) site_load<false, 64 * NW>(R, RO, rs, n, c + 1, nullptr, tid);     // in flight behind the MFMAs
    if (any) {
#pragma unroll
      for (int tt = 0; tt < TPW; ++tt) {
        f32x16 x[1][2];
        gate_tile(x[0], img_s, img_u, cv, pi[tt], pj[tt], hh);
        if (has_ctx) {
          f32x16 xg[1][2], g[1][2];
          linear6_pre<2, 4>(xg, af[tt], img_t, lane);
          linear6_T<2, 2, 1, true>(g, xg, Wg_l, cv + 64, lane);
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float wg = sigmoid_l2(g[0][mt][r]);
              x[0][mt][r] += wg * (xg[0][mt][r] - x[0][mt][r]);   // (1-w)*x + w*x_g
            }
        }
        f32x16 s1[1][2];
        linear6_T<2, 2, 1, true>(s1, x, S0_l, cv + 128, lane);
        f32x2v s2 = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 32 * mt + 8 * g4 + 4 * hh);
            const f32x4 x4 = {s1[0][mt][4 * g4], s1[0][mt][4 * g4 + 1], s1[0][mt][4 * g4 + 2], s1[0][mt][4 * g4 + 3]};
            gelu_dot4(s2, x4, w4);
          }
        float s = s2[0] + s2[1];
        s += __shfl_xor(s, 32);
        score[tt] += (s + w.s2b) * mc;
      }
    }
    if (more) site_store<false, true, 64 * NW>(R, nullptr, nxt + IMG64, nxt + IMG64 + 4096, nxt, tid, n);
    __syncthreads();
  }
  if (hh == 0) {
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
      const int p = ((pg * NW + wave) * TPW + tt) * 32 + (lane & 31);
      if (p < ppad) score_part[((size_t)b * nsc + sc) * ppad + p] = score[tt];
    }
  }
}

#ifndef NNJ_STEP0_TU           // nnj_api.hip only
// ------------------------------------------------------------------ incremental NJ step
// The n-1 new pairs (m, r) of a step, m = position of the freshly merged row.  Lane = partner
// row r (NT tiles of 32), so every lane streams ITS OWN rows S_r, U_r straight from HBM
// (16-byte pieces) and all lanes share S_m, U_m.  Each wave owns whole sites (c = c0+wave,
// +4, ...): no workgroup barrier in the loop, four sites in flight per CU, the next site's
// loads are issued before the current site's MFMAs.
struct IncLane {
  int m, slot_m;
  int r[2], slot_r[2];
  bool valid[2], r_first[2];     // r_first: r < m, i.e. the pair is (r, m) not (m, r)
};
template <int NT>
__device__ __forceinline__ IncLane inc_lane(const RowSet& rs, const int* ij_prev, int b, int n, int lane,
                                            int r0 = 0, int qn = 0) {
  IncLane L;
  L.m = min(max(ij_prev[2 * b], 0), n - 1);
  L.slot_m = slot_of(rs, b, L.m);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = r0 + 32 * nt + (lane & 31);             // qn: q, the row's index among the rows other than m
    L.r[nt] = r;
    L.valid[nt] = qn ? r < n - 1 : (r < n && r != L.m);
    const int pos = qn ? q_to_r(r, L.m) : r;
    L.slot_r[nt] = slot_of(rs, b, pos < n ? pos : 0);
    L.r_first[nt] = r < L.m;                               // (q < m <=> r < m)
  }
  return L;
}
template <int NT>
struct IncRaw { f32x16 sr[NT][2], ur[NT][2]; };      // this lane's row (sr prefetched one site ahead; ur = W_h sr recomputed)
struct IncShared { f32x16 sm[2], um[2]; };            // the merged row m (same for all lanes; loaded just in time)

__device__ __forceinline__ void inc_load_shared(IncShared& sh, const RowSet& rs, const IncLane& L, size_t bo, int C,
                                                int c, int hh) {
  const size_t om = bo + ((size_t)L.slot_m * C + c) * 64;
  load_token64(sh.sm, rs.S + om, true, hh);
  load_token64(sh.um, rs.U + om, true, hh);
}
template <int NT>
__device__ __forceinline__ void inc_load(IncRaw<NT>& raw, const RowSet& rs, const IncLane& L, size_t bo, int n,
                                         int C, int c, int hh) {
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const size_t o = bo + ((size_t)L.slot_r[nt] * C + c) * 64;
    // lanes beyond the live rows read row 0 (slot_r is clamped) and are NOT zeroed: their pairs are never
    // used, and as image rows/columns they only ever meet attention weights that are exactly 0
    load_token64(raw.sr[nt], rs.S + o, true, hh);
  }
}
// x = z*x_i + (1-z)*x_j with (i,j) = sort(m, r), z = sigmoid(U_i - U_j + b)  (model.py:105-108, 186-197).
// With 1 - sigmoid(t) = sigmoid(-t) both orders are one expression,
//     x = S_m + sigmoid(U_r - U_m + s*b) * (S_r - S_m),   s = +1 if r < m else -1,
// which needs no per-element selects (8 VALU instructions per element).
template <int NT>
__device__ __forceinline__ void inc_gate(f32x16 (&x)[NT][2], const IncRaw<NT>& raw, const IncShared& sh,
                                         const IncLane& L, const float* bh, int hh) {
  float sgn[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) sgn[nt] = L.r_first[nt] ? 1.0f : -1.0f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(bh + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k = 4 * g + t;
          const float z = sigmoid_l2((raw.ur[nt][mt][k] - sh.um[mt][k]) + sgn[nt] * b4[t]);
          x[nt][mt][k] = sh.sm[mt][k] + z * (raw.sr[nt][mt][k] - sh.sm[mt][k]);
        }
    }
}

// Phase A (alpha partials of the new pairs) lives in nnj_scorer16.hpp (k_inc_alpha16): 16-pair tiles, and no
// K' at all --  x . K'_r = (A^T x) . S_r + x . a0, whose last term is the same for every r and cancels in the
// softmax over r (k_alpha_softmax); the S rows are in registers anyway (the gate needs them).

// barrier of the two waves that share a site slot (LDS counter; see k_tok1p's pair_barrier)
__device__ __forceinline__ void pair_barrier_lds(int* cnt, int& epoch, int* flag) {
  epoch += 2;
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) atomicAdd(cnt, 1);
  int spins = 0;
  for (; *reinterpret_cast<volatile int*>(cnt) < epoch && spins < (1 << 22); ++spins) __builtin_amdgcn_s_sleep(1);
  if (spins == (1 << 22) && (threadIdx.x & 63) == 0) atomicOr(flag, NNJ_FLAG_BARRIER_TIMEOUT);   // never silent
  asm volatile("" ::: "memory");
}

// Phase B: scores of the new pairs.  Eight waves, one 32-pair tile each (two waves per SIMD: the partner
// wave's MFMAs run beside this wave's VALU work).  The transposed site image S_c^T -- A operand of
// x_g^T = S_c^T alpha^T, an f16x3 image [64 d][32*KT r'] -- is written from the rows the lanes hold.
// KT = 1 (n <= 32): every wave owns whole sites (c = c0 + wave, +8, ...), image private to the wave.
// KT = 2 (n > 32): waves w and w+4 share site slot w&3 and build the 64-column image together, each the
// columns of its 32 rows, between two pair barriers on an LDS counter (see k_tok1p).
// part[b][sc*NSLOT+slot][pair r].
template <int KT, bool CTX>
__global__ __launch_bounds__(512) void k_inc_score(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                   const float* __restrict__ alpha,
                                                   const uint8_t* __restrict__ mask,
                                                   float* __restrict__ score_part, int n, int C, int cs,
                                                   int* __restrict__ status, int qn) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wg_l = smem;                                      // operand images
  float* S0_l = smem + IMG64;
  float* Wh_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, tok = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NSLOT = 8 / KT;                            // sites in flight per workgroup
  const int slot = wave % NSLOT, tl = wave / NSLOT;        // tl: which 32 rows of the image this wave holds
  constexpr int TCH = 4 * KT;                              // 16-byte chunks per image row
  constexpr int IMG = b6_floats(64, 32 * KT);
  float* img_t = smem + 3 * IMG64 + slot * IMG;
  int* cnt = reinterpret_cast<int*>(smem + 3 * IMG64 + NSLOT * IMG) + slot;
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_weight_b6<64>(Wg_l, w.Wg, 64, tid, 512);
  stage_weight_b6<64>(S0_l, w.S0, 64, tid, 512);
  stage_weight_b6<64>(Wh_l, w.Wh, 64, tid, 512);
  float* cv = smem + 3 * IMG64 + NSLOT * IMG + 16;
  stage_scorer_consts(cv, w, tid);
  if (tid < NSLOT) reinterpret_cast<int*>(smem + 3 * IMG64 + NSLOT * IMG)[tid] = 0;
  __syncthreads();
  int epoch = 0;
  const IncLane L = inc_lane<1>(rs, ij_prev, b, n, lane, 32 * tl, qn);
  const int r = L.r[0];
  const size_t bo = (size_t)b * rs.bstride;
  // position of column r in an image row (the r' order of stage_weight_b6)
  const int q = 4 * tl + 2 * ((tok >> 4) & 1) + ((tok >> 2) & 1), e = 4 * ((tok >> 3) & 1) + (tok & 3);
  float score = 0.f;
  IncRaw<1> raw;
  int c = c0 + slot;
  if (c < c1) inc_load<1>(raw, rs, L, bo, n, C, c, hh);
  for (; c < c1; c += NSLOT) {
    asm volatile("" ::: "memory");
    const float mc = mask[(size_t)b * C + c] ? 0.f : 1.f;             // seq_mask (model.py:96); first load of the iteration
    // alpha[pair][r'] (only r' < 32*KT can be non-zero) is re-read per site from L2 (keeping it in registers
    // next to the prefetched rows would spill); issued first, it lands behind the gate and the image
    Frag3 at[2 * KT];                                      // pre-split k-step fragments (see k_alpha_softmax)
    if constexpr (CTX) {
#pragma unroll
      for (int ks = 0; ks < 2 * KT; ++ks)
        alpha_frag32(at[ks], alpha, (long)gridDim.y * 4096, ((size_t)b * 64 + r) * 64, ks, hh, true);
    }
    f32x16 x[1][2];
    {
      // fp16 pieces of S_r: the B operand of U_r = W_h S_r (recomputed on the idle matrix pipe: the cached U rows
      // were half of the kernel's HBM reads) and, transposed, the columns of the image -- written before the gate
      // so that the pieces are dead by then
      Frag3 sf[4];
      static_for<0, 4>([&](auto ki) {
        constexpr int ks = decltype(ki)::value;
        split8<8 * (ks & 1), false>(sf[ks], raw.sr[0][ks >> 1]);
      });
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < 16; ++k) raw.ur[0][mt][k] = 0.f;
      const u32x4* wh4 = reinterpret_cast<const u32x4*>(Wh_l);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const int wrow = 32 * mt + tok;
          const int o = wrow * 8 + wswz6<8>(wrow, 2 * ks + hh);
          Frag3 a;
          a.h = wh4[o]; a.m = wh4[64 * 8 + o];
          raw.ur[0][mt] = mfma_b6(a, sf[ks], raw.ur[0][mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (CTX) {
        if constexpr (KT == 2) pair_barrier_lds(cnt, epoch, status);   // the partner is done with the previous site's image
        unsigned short* t16 = reinterpret_cast<unsigned short*>(img_t);
        constexpr int PL = 64 * 32 * KT;                       // plane stride in fp16 elements
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
              const unsigned h = sf[2 * mt + (g >> 1)].h[2 * (g & 1) + pr], m = sf[2 * mt + (g >> 1)].m[2 * (g & 1) + pr];
              const int d0 = 32 * mt + 8 * g + 4 * hh + 2 * pr, d1 = d0 + 1;
              const int o0 = d0 * (32 * KT) + 8 * wswz6<TCH>(d0, q) + e, o1 = d1 * (32 * KT) + 8 * wswz6<TCH>(d1, q) + e;
              t16[o0] = (unsigned short)h; t16[o1] = (unsigned short)(h >> 16);
              t16[PL + o0] = (unsigned short)m; t16[PL + o1] = (unsigned short)(m >> 16);
            }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      IncShared sh;
      inc_load_shared(sh, rs, L, bo, C, c, hh);
      inc_gate<1>(x, raw, sh, L, cv, hh);
    }
    if constexpr (CTX && KT == 2) pair_barrier_lds(cnt, epoch, status);   // all 64 columns are in the image
    const int cn = c + NSLOT;
    inc_load<1>(raw, rs, L, bo, n, C, cn < c1 ? cn : c, hh);           // prefetch behind the MFMAs (last: harmless reload)
    if constexpr (CTX) {
      f32x16 xg[1][2], g[1][2];
      linear6_pre<2, 2 * KT>(xg, at, img_t, lane);
      linear6_T<2, 2, 1, true, false>(g, xg, Wg_l, cv + 64, lane);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          const float wg = sigmoid_l2(g[0][mt][k]);
          x[0][mt][k] += wg * (xg[0][mt][k] - x[0][mt][k]);   // (1-w)*x + w*x_g
        }
    }
    f32x16 s1[1][2];
    linear6_T<2, 2, 1, true, false>(s1, x, S0_l, cv + 128, lane);
    f32x2v s2 = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 32 * mt + 8 * g + 4 * hh);
        const f32x4 x4 = {s1[0][mt][4 * g], s1[0][mt][4 * g + 1], s1[0][mt][4 * g + 2], s1[0][mt][4 * g + 3]};
        gelu_dot4(s2, x4, w4);
      }
    float s = s2[0] + s2[1];
    s += __shfl_xor(s, 32);
    score += (s + w.s2b) * mc;
  }
  // one partial set per WORKGROUP: the slots' sums meet in LDS (the images are dead) and are added in slot order
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // [NSLOT][64]
  if (hh == 0) red[slot * 64 + r] = score;
  if (KT == 1 && lane >= 32) red[slot * 64 + lane] = 0.f;  // pair rows 32..63 have no wave
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < NSLOT; ++s_) v += red[s_ * 64 + tid];
    score_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  }
}

// ------------------------------------------------------------------ row transforms
// Tokens = sites.  For one row tile held feature-major in registers, write
// U = W_h S, K' = A S + a0 and the beta partial of the wave's 32 sites.
__device__ __forceinline__ void row_transforms(const f32x16 (&s)[1][2], const float* Wh_l, const float* A_l,
                                               const ScorerW& w, float* U_row, float* Kp_row, float* beta_dst,
                                               int c, bool valid, int lane) {
  const int hh = lane >> 5;
  f32x16 o[1][2];
  linear6_T_nb<2, 2, 1>(o, s, Wh_l, lane);
  store_token64(o[0], U_row + (size_t)c * 64, valid, hh);
  linear6_T<2, 2, 1>(o, s, A_l, w.a0, lane);
  store_token64(o[0], Kp_row + (size_t)c * 64, valid, hh);
  float d = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 u4 = *reinterpret_cast<const f32x4*>(w.u + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
      for (int t = 0; t < 4; ++t) d += u4[t] * s[0][mt][4 * g + t];
    }
  if (!valid) d = 0.f;
#pragma unroll
  for (int o2 = 32; o2 >= 1; o2 >>= 1) d += __shfl_xor(d, o2);
  if (lane == 0) *beta_dst = d;
}

// k_row_xf: transforms of existing rows. grid (ceil(C/128), rows, B); wave = 32 sites.
__global__ __launch_bounds__(256) void k_row_xf(const float* __restrict__ S, float* __restrict__ U,
                                                float* __restrict__ Kp, float* __restrict__ beta_part,
                                                ScorerW w, long bstride, int slots, int C, int ntile32, int bstr) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wh_l = smem;
  float* A_l = smem + 4096;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  stage_weight_b6<64>(Wh_l, w.Wh, 64, tid, 256);
  stage_weight_b6<64>(A_l, w.A, 64, tid, 256);
  __syncthreads();
  const int tile = blockIdx.x * 4 + wave, row = blockIdx.y, b = blockIdx.z;
  if (tile >= ntile32) return;
  const int c = tile * 32 + (lane & 31);
  const bool valid = c < C;
  const size_t roff = (size_t)b * bstride + (size_t)row * C * 64;
  f32x16 s[1][2];
  load_token64(s[0], S + roff + (size_t)(valid ? c : 0) * 64, valid, lane >> 5);
  row_transforms(s, Wh_l, A_l, w, U + roff, Kp + roff, beta_part + ((size_t)b * slots + row) * bstr + tile,
                 c, valid, lane);
}

// ------------------------------------------------------------------ merged-row aggregate
// k_agg_alpha: alpha partials of the ONE merged pair per batch element (VALU stream).
// part[b][chunk][r] = sum_{c in chunk} x[c,:] . K'_r[c,:]; chunk = 16 sites; rp = row stride of part (64, or 128 / 256
// while more than 64 rows are live).
__global__ __launch_bounds__(256) void k_agg_alpha(RowSet rs, ScorerW w, const int* __restrict__ ij,
                                                   float* __restrict__ part, int n, int C, int rp) {
  __shared__ __attribute__((aligned(16))) float xl[1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = blockIdx.x, b = blockIdx.y;
  const int pi = min(max(ij[2 * b], 0), n - 1), pj = min(max(ij[2 * b + 1], 0), n - 1);   // never index outside the rows
  const size_t bo = (size_t)b * rs.bstride;
  const size_t oi = bo + (size_t)slot_of(rs, b, pi) * C * 64, oj = bo + (size_t)slot_of(rs, b, pj) * C * 64;
  const int e = tid * 4;                       // element within the chunk: site e/64, feature e%64
  const int c = chunk * 16 + (e >> 6);
  f32x4 x = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const size_t off = (size_t)c * 64 + (e & 63);
    const f32x4 si = *reinterpret_cast<const f32x4*>(rs.S + oi + off);
    const f32x4 sj = *reinterpret_cast<const f32x4*>(rs.S + oj + off);
    const f32x4 ui = *reinterpret_cast<const f32x4*>(rs.U + oi + off);
    const f32x4 uj = *reinterpret_cast<const f32x4*>(rs.U + oj + off);
    const f32x4 b4 = *reinterpret_cast<const f32x4*>(w.bh + (e & 63));
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float z = sigmoid_l2(ui[t] - uj[t] + b4[t]);
      x[t] = sj[t] + z * (si[t] - sj[t]);
    }
  }
  *reinterpret_cast<f32x4*>(xl + e) = x;
  // the live-list lookups of all rows at once (in front of every row's loads they were a dependent round trip each)
  __shared__ int slots[256];
  if (tid < n) slots[tid] = slot_of(rs, b, tid);
  __syncthreads();
  const int nch = gridDim.x;
  // this lane's part of x in registers; two rows per iteration: eight 16-byte loads of a wave in flight
  f32x4 xv[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ee = (k * 64 + lane) * 4;
    ok[k] = chunk * 16 + (ee >> 6) < C;
    xv[k] = *reinterpret_cast<const f32x4*>(xl + ee);
  }
  const size_t coff = (size_t)chunk * 16 * 64 + (size_t)lane * 4;
  for (int r = wave; r < n; r += 8) {
    const int r2 = r + 4 < n ? r + 4 : r;                    // (last odd row: loaded twice, stored once)
    const float* k0 = rs.Kp + bo + (size_t)slots[r] * C * 64 + coff;
    const float* k1 = rs.Kp + bo + (size_t)slots[r2] * C * 64 + coff;
    f32x4 kv0[4], kv1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      kv0[k] = ok[k] ? *reinterpret_cast<const f32x4*>(k0 + k * 256) : (f32x4){0.f, 0.f, 0.f, 0.f};
      kv1[k] = ok[k] ? *reinterpret_cast<const f32x4*>(k1 + k * 256) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    float acc0 = 0.f, acc1 = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                             // same order of the sums as one row at a time
      acc0 += xv[k][0] * kv0[k][0] + xv[k][1] * kv0[k][1] + xv[k][2] * kv0[k][2] + xv[k][3] * kv0[k][3];
      acc1 += xv[k][0] * kv1[k][0] + xv[k][1] * kv1[k][1] + xv[k][2] * kv1[k][2] + xv[k][3] * kv1[k][3];
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { acc0 += __shfl_xor(acc0, o); acc1 += __shfl_xor(acc1, o); }
    if (lane == 0) {
      part[((size_t)b * nch + chunk) * rp + r] = acc0;
      if (r2 != r) part[((size_t)b * nch + chunk) * rp + r2] = acc1;
    }
  }
}

// k_agg_finish: softmax of the merged pair's alpha, x_g, gate, new row; writes the row
// (and, when U_out != nullptr, its cached transforms).  Tokens = sites; grid (ceil(C/128), B).
//   out_row(b) = out_base + b*out_bstride + out_slot(b)*C*64 where out_slot = live[pi]
//   (in place, environment.py:764-768) or 0 for the dense one-row output of nnj_aggregate.
//   SPLIT (small batches, where 128-site workgroups would not fill the chip): the workgroup owns ONE 32-site
//   tile and its four waves split the rows of the x_g sum (every 4th row each, partial sums added in wave order
//   through LDS); wave 0 finishes the tile.
template <bool SPLIT, int RPT>                            // RPT: 64-row groups of the alpha vector (1; 2 or 4 above 64 live rows)
__global__ __launch_bounds__(256) void k_agg_finish(RowSet rs, ScorerW w, const int* __restrict__ ij,
                                                    const float* __restrict__ part, int nch, float* S_out,
                                                    float* U_out, float* Kp_out, float* beta_out,
                                                    long out_bstride, int out_slots, int in_place, int n, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wg_l = smem;
  float* Wh_l = smem + 4096;
  float* A_l = smem + 8192;
  float* al = smem + 12288;       // alpha[64 * RPT] (256 floats reserved), then 1024 floats of reduction scratch
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int b = blockIdx.y;
  const int pi = min(max(ij[2 * b], 0), n - 1), pj = min(max(ij[2 * b + 1], 0), n - 1);
  stage_weight_b6<64>(Wg_l, w.Wg, 64, tid, 256);
  if (U_out) {
    stage_weight_b6<64>(Wh_l, w.Wh, 64, tid, 256);
    stage_weight_b6<64>(A_l, w.A, 64, tid, 256);
  }
  const bool has_ctx = n > 2;                       // model.py:111
  constexpr int RP = 64 * RPT;
  {
    // alpha logits of the merged pair: the chunk partials of k_agg_alpha and the rows' beta partials are summed by
    // all four waves (every 4th term each, all loads of a wave in flight at once; the serial loop of one wave was
    // most of this kernel's latency at small batch), the four sums meet in LDS in wave order.  Lane = row mod 64.
    float* red4 = al + 256;                         // [4][RP]
    bool in[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int r = 64 * k + lane;
      in[k] = has_ctx && r < n && r != pi && r != pj;
      float s = 0.f;
      if (in[k]) {
#pragma unroll 16
        for (int ch = wave; ch < nch; ch += 4) s += part[((size_t)b * nch + ch) * RP + r];
        const float* bp = rs.beta_part + ((size_t)b * (rs.bstride / ((long)C * 64)) + slot_of(rs, b, r)) * rs.ntile32;
        float beta = 0.f;
#pragma unroll 8
        for (int t = wave; t < rs.ntile32; t += 4) beta += bp[t];
        s += beta;
      }
      red4[wave * RP + r] = s;
    }
    __syncthreads();
    if (wave == 0) {
      float a[RPT];
      float mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const int r = 64 * k + lane;
        a[k] = -INFINITY;
        if (in[k]) {
          const float tot = ((red4[r] + red4[RP + r]) + red4[2 * RP + r]) + red4[3 * RP + r] + (float)C * w.t0;
          a[k] = tot * (1.0f / sqrtf(64.0f * (float)C));
        }
        mx = fmaxf(mx, a[k]);
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
      float se = 0.f;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        a[k] = in[k] ? expf(a[k] - mx) : 0.f;
        se += a[k];
      }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
#pragma unroll
      for (int k = 0; k < RPT; ++k) al[64 * k + lane] = (se > 0.f) ? a[k] / se : 0.f;
    }
  }
  __syncthreads();
  const int tile = SPLIT ? blockIdx.x : blockIdx.x * 4 + wave;
  if (tile * 32 >= C) return;
  const int c = tile * 32 + (lane & 31);
  const bool valid = c < C;
  const size_t bo = (size_t)b * rs.bstride;
  const int slot_i = slot_of(rs, b, pi), slot_j = slot_of(rs, b, pj);
  if (SPLIT && !has_ctx && wave != 0) return;          // nothing to split
  f32x16 x[1][2], xg[1][2];
  {
    f32x16 si[2], sj[2], ui[2], uj[2];
    load_token64(si, rs.S + bo + ((size_t)slot_i * C + (valid ? c : 0)) * 64, valid, hh);
    load_token64(sj, rs.S + bo + ((size_t)slot_j * C + (valid ? c : 0)) * 64, valid, hh);
    load_token64(ui, rs.U + bo + ((size_t)slot_i * C + (valid ? c : 0)) * 64, valid, hh);
    load_token64(uj, rs.U + bo + ((size_t)slot_j * C + (valid ? c : 0)) * 64, valid, hh);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(w.bh + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const float z = sigmoid_l2(ui[mt][4 * g + t] - uj[mt][4 * g + t] + b4[t]);
          x[0][mt][4 * g + t] = sj[mt][4 * g + t] + z * (si[mt][4 * g + t] - sj[mt][4 * g + t]);
        }
      }
  }
  if (has_ctx) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) xg[0][mt][r] = 0.f;
#pragma unroll 4                                   // several rows' loads in flight (rows i, j have weight 0)
    for (int r = SPLIT ? wave : 0; r < n; r += SPLIT ? 4 : 1) {
      const float a = al[r];
      f32x16 sr[2];
      load_token64(sr, rs.S + bo + ((size_t)slot_of(rs, b, r) * C + (valid ? c : 0)) * 64, valid, hh);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) xg[0][mt] += a * sr[mt];
    }
    if (SPLIT) {
      float* red = al + 256;                           // [4 waves][64 lanes][32] (the alpha scratch is dead)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(red + ((wave * 64 + lane) * 32 + 16 * mt + 4 * g)) =
              (f32x4){xg[0][mt][4 * g], xg[0][mt][4 * g + 1], xg[0][mt][4 * g + 2], xg[0][mt][4 * g + 3]};
      __syncthreads();
      if (wave != 0) return;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v = *reinterpret_cast<const f32x4*>(red + (lane * 32 + 16 * mt + 4 * g));
#pragma unroll
          for (int w2 = 1; w2 < 4; ++w2) v += *reinterpret_cast<const f32x4*>(red + ((w2 * 64 + lane) * 32 + 16 * mt + 4 * g));
          xg[0][mt][4 * g] = v[0]; xg[0][mt][4 * g + 1] = v[1]; xg[0][mt][4 * g + 2] = v[2]; xg[0][mt][4 * g + 3] = v[3];
        }
    }
    f32x16 g[1][2];
    linear6_T<2, 2, 1>(g, xg, Wg_l, w.bg, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float wg = sigmoid_l2(g[0][mt][r]);
        x[0][mt][r] += wg * (xg[0][mt][r] - x[0][mt][r]);
      }
  }
  const int oslot = in_place ? slot_i : 0;
  const size_t oo = (size_t)b * out_bstride + (size_t)oslot * C * 64;
  store_token64(x[0], S_out + oo + (size_t)c * 64, valid, hh);
  if (U_out)
    row_transforms(x, Wh_l, A_l, w, U_out + oo, Kp_out + oo,
                   beta_out + ((size_t)b * out_slots + oslot) * rs.ntile32 + tile, c, valid, lane);
}

// What the two-pass step (nnj_step2.hpp) needs from the table kernel besides the pick: the attention weights of the
// NEXT merge (the merged pair's logits exist already when the pick is one of this step's new pairs or the candidate
// named one step ago) and the candidate of the next step.  All null for the callers that do not use it.
struct StepOut {
  const float* lam;           // [B][64 q][64 q'] summed logits of this step's new pairs (k_step_softmax), or null
  const float* beta_slot;     // [B][slots] per-row bias of the attention logits (sum of the row's beta partials + C t0)
  int nslot;
  const float* acand_part;    // [B][nblk][64] logits of the candidate against the rows (k_step_alpha), or null
  int nblk;
  const int* cand_cur;        // [B][2] the candidate those logits belong to, positions of THIS table; (-1,-1) = none
  const float* alpha0;        // step 0 (all pairs, <= 64 rows): k_pair_alpha's partials [B][nsc0][ppad0][64], or null
  int nsc0, ppad0;
  float* am;                  // out [B][64]: weights of the merge just picked, by position AFTER it (0 at i)
  int* need;                  // out [B]: 1 = no source applied, the fallback kernels must compute `am`
  int* cand_next;             // out [B][2]: best entry of this table without rows i, j, by position after the merge
  int* cand_run;              // out [B]: 1 = that pair is not the one whose x' row exists (k_pair_xp must run)
  float inv_scale;            // 1 / sqrt(64 C)
  int fallback;               // 1 = the caller launches the fallback kernels where `need` is set; 0 = a pick without a
                              // source is an internal error (sticky status bit NNJ_FLAG_MERGE_WEIGHTS: never silent)
  int rep0;                   // 1 = step 0 of B replicas of ONE alignment (sampled rollouts, finetune_rl_search.py:338-427):
                              // the all-pairs kernels ran for alignment 0 only, its partials serve every replica's table
};

// ------------------------------------------------------------------ table assemble + argmax
// One workgroup per batch element.  new/first scores = fixed-order sum of score parts;
// table via the old->new index map (utils.py:227-247) from logits_prev; argmax with
// first-maximal-index tie break (finetune_rl_search.py:145); flat -> (i,j)
// (environment.py:457-462).
__global__ __launch_bounds__(256) void k_assemble_argmax(const float* __restrict__ score_part, int nsc, int ppad,
                                                         const float* __restrict__ logits_prev,
                                                         const int* ij_prev,
                                                         float* __restrict__ logits_out,
                                                         float* __restrict__ trace_out, long trace_bstride,
                                                         const int* __restrict__ forced, long forced_bstride,
                                                         int* __restrict__ merges_out, long merges_bstride,
                                                         float* __restrict__ gap_out, long gap_bstride,
                                                         int* ij_cur, int mode, int n,
                                                         const float* __restrict__ uniforms, long u_bstride,
                                                         float inv_temp, int* __restrict__ nonfinite,
                                                         const int* __restrict__ live_cur, int* __restrict__ live_next,
                                                         int live_stride, int qn, StepOut so) {
  __shared__ float newsc[256];
  __shared__ int picked_j, picked_i;
  // the two-pass step asks for the best entry WITHOUT the rows of the pick: the pass over the table keeps every
  // entry's rows (and value) in LDS so that the second pass is a few LDS reads per thread
  constexpr int PCACHE = 2304;
  __shared__ unsigned short pcode[PCACHE];
  __shared__ float pval[PCACHE];
  // the rollout keeps two live lists: this kernel also writes the NEXT one (the current list without position j:
  // environment.py:764-768), which used to be a launch of its own per step (k_update_live; 5 us of the 100 us of a
  // step at batch 1).  The current entries are loaded here, their latency hides behind the table pass.
  int lv0 = 0, lv1 = 0;
  if (live_next && threadIdx.x < n - 1) {
    lv0 = live_cur[(size_t)blockIdx.x * live_stride + threadIdx.x];
    lv1 = live_cur[(size_t)blockIdx.x * live_stride + threadIdx.x + 1];
  }
  __shared__ float red_v[8];
  __shared__ int red_i[4];
  const int tid = threadIdx.x, b = blockIdx.x;
  const size_t bsrc = so.rep0 ? 0 : b;                      // whose all-pairs partials this table is assembled from
  const int np = num_pairs(n), np_prev = num_pairs(n + 1);
  int ip = 0, jp = 0;
  if (mode == PAIRS_INCR) { ip = ij_prev[2 * b]; jp = ij_prev[2 * b + 1]; }
  // Round 4 (one alignment per rollout: this kernel is a chain of dependent L2 round trips, 11.8 us of a 46 us step):
  // everything the two-pass tail needs that does NOT depend on the pick is fetched here, behind the table pass -- the
  // carried candidate's pair, its logits' partial sums (used only if the pick turns out to be that pair: same sums, same
  // order), the per-row bias of every position.
  // Sums of MANY partial sets (one alignment per rollout: 256 workgroups of the step kernels, 64 values each): thread
  // (phase tid >> 4, column group tid & 15) adds every 16th set, four columns at a time -- 16 independent 16-byte loads in
  // flight per thread and sum, ONE round trip where four threads per column took two to four -- and the 16 phase sums of a
  // column are added in phase order (fixed order: reproducible).  psum[which][phase][column].
  __shared__ float psum[2][16][64];
  auto phase_sums = [&](const float* part, int nset, int which) {
    const int cg = tid & 15, ph = tid >> 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
    for (int k = ph; k < nset; k += 16) a += *reinterpret_cast<const f32x4*>(part + (size_t)k * 64 + 4 * cg);
    *reinterpret_cast<f32x4*>(&psum[which][ph][4 * cg]) = a;
  };
  auto phase_total = [&](int which, int col) {
    float v = psum[which][0][col];
#pragma unroll
    for (int ph = 1; ph < 16; ++ph) v += psum[which][ph][col];
    return v;
  };
  int cc0 = -2, cc1 = -2;
  float a_cand = 0.f, beta_r = 0.f;
  const bool cand_phases = so.am && so.acand_part && so.cand_cur && so.nblk >= 32;
  if (so.am) {
    if (so.acand_part && so.cand_cur) {
      cc0 = so.cand_cur[2 * b]; cc1 = so.cand_cur[2 * b + 1];
      if (cand_phases) phase_sums(so.acand_part + (size_t)b * so.nblk * 64, so.nblk, 1);
      else {
        const int r = tid & 63, part = tid >> 6;
        if (r < n) {
          const int col = (mode == PAIRS_INCR && r == ip) ? 63 : r - ((mode == PAIRS_INCR && r > ip) ? 1 : 0);
#pragma unroll 16
          for (int k = part; k < so.nblk; k += 4) a_cand += so.acand_part[((size_t)b * so.nblk + k) * 64 + col];
        }
      }
    }
    if (tid < 64 && tid < n) beta_r = so.beta_slot[(size_t)b * so.nslot + live_cur[(size_t)b * live_stride + tid]];
  }
  if (mode == PAIRS_INCR) {
    if (qn && nsc >= 32 && ppad == 64) {
      phase_sums(score_part + (size_t)b * nsc * 64, nsc, 0);
      __syncthreads();
      if (tid < n) newsc[tid] = phase_total(0, tid - (tid > ip ? 1 : 0));
    } else if (qn && nsc > 8) {
      // several partials and at most 64 new scores: four threads per score, every 4th partial each, the four sums added
      // in order
      const int r = tid & 63, part = tid >> 6;
      float s = 0.f;
      if (r < n) {
        const int src = r - (r > ip ? 1 : 0);
#pragma unroll 32
        for (int sc = part; sc < nsc; sc += 4) s += score_part[((size_t)b * nsc + sc) * ppad + src];
      }
      pval[PCACHE - 256 + tid] = s;
      __syncthreads();
      if (tid < n) newsc[tid] = ((pval[PCACHE - 256 + tid] + pval[PCACHE - 192 + tid]) + pval[PCACHE - 128 + tid]) + pval[PCACHE - 64 + tid];
    } else if (tid < n) {
      // qn: the partials are indexed by q, the partner's index among the rows other than the merged one (nnj_step2.hpp)
      const int src = qn ? tid - (tid > ip ? 1 : 0) : tid;
      float s = 0.f;
#pragma unroll 8
      for (int sc = 0; sc < nsc; ++sc) s += score_part[((size_t)b * nsc + sc) * ppad + src];
      newsc[tid] = s;
    }
    __syncthreads();
  }
  // one pass: every thread keeps the best (first maximal index wins) and the runner-up VALUE of its entries; the
  // triples are merged by wavefront shuffles (6 steps) and once across the four waves through LDS -- two workgroup
  // barriers instead of the 16 + 8 of a shared-memory tree
  float best = -INFINITY, second = -INFINITY;
  int besti = 0x7fffffff;
  const bool pc = so.cand_next != nullptr && np <= PCACHE;
  for (int p = tid; p < np; p += 256) {
    float v;
    if (mode == PAIRS_FULL) {
      v = 0.f;
#pragma unroll 8
      for (int sc = 0; sc < nsc; ++sc) v += score_part[(bsrc * nsc + sc) * ppad + p];
      if (pc) {
        int ii, jj;
        pair_from_index(n, p, ii, jj);
        pcode[p] = (unsigned short)(ii << 8 | jj);
      }
    } else {
      int ii, jj;
      pair_from_index(n, p, ii, jj);
      if (ii == ip) v = newsc[jj];
      else if (jj == ip) v = newsc[ii];
      else v = logits_prev[(size_t)b * np_prev + pair_index(n + 1, ii + (ii >= jp), jj + (jj >= jp))];
      if (pc) pcode[p] = (unsigned short)(ii << 8 | jj);
    }
    if (pc) pval[p] = v;
    logits_out[(size_t)b * np + p] = v;
    if (trace_out) trace_out[(size_t)b * trace_bstride + p] = v;
    if (v > best) { second = best; best = v; besti = p; }       // p increases: a later equal value never replaces
    else second = fmaxf(second, v);
    // a score that is not finite means an operand left the range of the fp16 pieces (nnj_common.hpp): sticky flag,
    // read back by nnj_numeric_status -- the library must not return a wrong tree silently
    if (!(fabsf(v) <= 3.402823466e38f) && nonfinite) atomicOr(nonfinite, NNJ_FLAG_NONFINITE);
  }
  auto merge = [](float& b1, int& i1, float& s1, float b2, int i2, float s2) {
    if (b2 > b1 || (b2 == b1 && i2 < i1)) { s1 = fmaxf(b1, s2); b1 = b2; i1 = i2; }
    else s1 = fmaxf(s1, b2);
  };
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float b2 = __shfl_xor(best, o), s2 = __shfl_xor(second, o);
    const int i2 = __shfl_xor(besti, o);
    merge(best, besti, second, b2, i2, s2);
  }
  if ((tid & 63) == 0) { red_v[tid >> 6] = best; red_i[tid >> 6] = besti; red_v[4 + (tid >> 6)] = second; }
  __syncthreads();
  best = red_v[0]; besti = red_i[0]; second = red_v[4];
#pragma unroll
  for (int wv = 1; wv < 4; ++wv) merge(best, besti, second, red_v[wv], red_i[wv], red_v[4 + wv]);
  const int bi = (besti >= 0 && besti < np) ? besti : 0;   // all-NaN table: stay in range
  const float bv = best;
  int pick = bi;
  if (uniforms) {
    // Categorical(logits / temperature).sample() by inverse CDF on a supplied uniform u in [0,1)
    // (finetune_rl_search.py:147): smallest k with sum_{p<=k} e_p > u * sum_p e_p, fp64, flat pair order.
    // Parallel: thread t owns the contiguous entries [t*chunk, (t+1)*chunk); local fp64 sums, an exclusive scan of the
    // 256 sums (wavefront shuffles + the four wave totals through LDS), then the FIRST thread whose range crosses
    // the target walks its entries.  (The running sum is associated per chunk, not strictly left to right: it can
    // differ from a sequential sum in the last fp64 bit, i.e. only for a uniform within 1e-16 of a CDF boundary.)
    __shared__ double wsum[4];
    __shared__ int first_hit;
    if (tid == 0) first_hit = 0x7fffffff;
    const float* lg = logits_out + (size_t)b * np;
    const int chunk = (np + 255) / 256;
    const int p0 = min(tid * chunk, np), p1 = min(p0 + chunk, np);
    double local = 0.0;
    for (int p = p0; p < p1; ++p) local += exp((double)(lg[p] - bv) * (double)inv_temp);
    double incl = local;                                   // inclusive scan inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const double up = __shfl_up(incl, o);
      if ((tid & 63) >= o) incl += up;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    double base = 0.0, total = 0.0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) { if (wv < (tid >> 6)) base += wsum[wv]; total += wsum[wv]; }
    const double before = base + incl - local;             // sum of all entries of lower flat index
    const double target = (double)uniforms[(size_t)b * u_bstride] * total;
    if (p1 > p0 && before + local > target) atomicMin(&first_hit, tid);
    __syncthreads();
    const int hit = first_hit;
    if (hit == 0x7fffffff) pick = np - 1;                  // rounding: target >= total
    else if (tid == hit) {
      double run = before;
      int k = p1 - 1;
      for (int p = p0; p < p1; ++p) {
        run += exp((double)(lg[p] - bv) * (double)inv_temp);
        if (run > target) { k = p; break; }
      }
      first_hit = -k - 1;                                  // publish the pick to thread 0
    }
    __syncthreads();
    if (hit != 0x7fffffff) pick = -first_hit - 1;
  }
  if (tid == 0) {
    int ci, cj;
    pair_from_index(n, pick, ci, cj);
    if (merges_out) { merges_out[(size_t)b * merges_bstride] = ci; merges_out[(size_t)b * merges_bstride + 1] = cj; }
    if (gap_out) gap_out[(size_t)b * gap_bstride] = np > 1 ? bv - second : 0.f;
    if (forced) {
      const int fi = forced[(size_t)b * forced_bstride], fj = forced[(size_t)b * forced_bstride + 1];
      if (fi >= 0 && fi < fj && fj < n) { ci = fi; cj = fj; }       // out-of-range forcing is ignored
    }
    ij_cur[2 * b] = ci; ij_cur[2 * b + 1] = cj;
    picked_j = min(max(cj, 1), n - 1);
    picked_i = min(max(ci, 0), n - 2);
  }
  if (live_next) {
    __syncthreads();
    if (tid < n - 1) live_next[(size_t)b * live_stride + tid] = tid >= picked_j ? lv1 : lv0;
  }
  if (!so.am) return;
  // ---- the two-pass step: weights of the merge just picked, candidate of the next step
  __syncthreads();
  const int pi_ = picked_i, pj_ = picked_j;
  // source of the merged pair's logits a_r (nnj_step2.hpp): 1 = a pair scored in this step, 2 = the candidate
  int src = 0, qs = 0;
  if (mode == PAIRS_INCR && so.lam && (pi_ == ip || pj_ == ip)) {
    src = 1;
    const int rs_ = pi_ == ip ? pj_ : pi_;
    qs = rs_ - (rs_ > ip ? 1 : 0);
  } else if (so.acand_part && so.cand_cur && cc0 == pi_ && cc1 == pj_) {
    src = 2;
  } else if (mode == PAIRS_FULL && so.alpha0) {
    src = 3;                                               // step 0: every pair was scored in this step
  }
  if (src == 3) {
    // the picked pair's logits: the partials of k_pair_alpha (x . K'_r: they differ from (A^T x) . S_r by a term that
    // is the same for every r), four threads per row
    const int r = tid & 63, part = tid >> 6;
    const size_t pp = (size_t)pair_index(n, pi_, pj_);
    float a = 0.f;
    if (r < n)
#pragma unroll 16
      for (int k = part; k < so.nsc0; k += 4) a += so.alpha0[((bsrc * so.nsc0 + k) * so.ppad0 + pp) * 64 + r];
    pval[PCACHE - 256 + tid] = a;
    __syncthreads();
  }
  if (src == 2 && !cand_phases) {
    // the candidate's logits: the partials of the k_step_alpha workgroups, four threads per row (every 4th block
    // each; summed at the top of the kernel), the four sums added in order by wave 0 below
    pval[PCACHE - 256 + tid] = a_cand;                      // (entries the table never reaches: np <= 2080 when src == 2)
    __syncthreads();
  }
  if (tid < 64) {
    const int r = tid;
    const bool in = src != 0 && n > 2 && r < n && r != pi_ && r != pj_;
    float v = -INFINITY;
    if (in) {
      float a;
      if (src == 1) a = so.lam[((size_t)b * 64 + qs) * 64 + (r - (r > ip ? 1 : 0))];
      else if (src == 2 && cand_phases)                      // (psum[1] is complete: several workgroup barriers since)
        a = phase_total(1, (mode == PAIRS_INCR && r == ip) ? 63 : r - ((mode == PAIRS_INCR && r > ip) ? 1 : 0));
      else a = ((pval[PCACHE - 256 + r] + pval[PCACHE - 192 + r]) + pval[PCACHE - 128 + r]) + pval[PCACHE - 64 + r];   // src 2, 3
      v = (a + beta_r) * so.inv_scale;
    }
    float mx = v;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = in ? expf(v - mx) : 0.f;
    float se = e;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
    so.am[(size_t)b * 64 + tid] = 0.f;
    __builtin_amdgcn_wave_barrier();
    if (in) so.am[(size_t)b * 64 + (r - (r > pj_ ? 1 : 0))] = se > 0.f ? e / se : 0.f;
    if (tid == 0) {
      so.need[b] = (src == 0 && n > 2) ? 1 : 0;
      if (src == 0 && n > 2 && !so.fallback && nonfinite) atomicOr(nonfinite, NNJ_FLAG_MERGE_WEIGHTS);
    }
  }
  if (!so.cand_next) return;
  // best entry of this table that survives the merge (first maximal index: the order of the surviving pairs is the
  // order of the next table's old part) -- the only OLD pair the next argmax can pick
  float cb = -INFINITY;
  int cbi = 0x7fffffff;
  for (int p = tid; p < np; p += 256) {
    int ii, jj;
    float v;
    if (pc) { const int code = pcode[p]; ii = code >> 8; jj = code & 255; v = pval[p]; }
    else { pair_from_index(n, p, ii, jj); v = logits_out[(size_t)b * np + p]; }
    if (ii == pi_ || ii == pj_ || jj == pi_ || jj == pj_) continue;
    if (v > cb) { cb = v; cbi = p; }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const float b2 = __shfl_xor(cb, o);
    const int i2 = __shfl_xor(cbi, o);
    if (b2 > cb || (b2 == cb && i2 < cbi)) { cb = b2; cbi = i2; }
  }
  __syncthreads();                                          // (red_v / red_i of the argmax are dead)
  if ((tid & 63) == 0) { red_v[tid >> 6] = cb; red_i[tid >> 6] = cbi; }
  __syncthreads();
  if (tid == 0) {
    cb = red_v[0]; cbi = red_i[0];
#pragma unroll
    for (int wv = 1; wv < 4; ++wv)
      if (red_v[wv] > cb || (red_v[wv] == cb && red_i[wv] < cbi)) { cb = red_v[wv]; cbi = red_i[wv]; }
    int na = -1, nb = -1;
    if (cbi >= 0 && cbi < np) {
      int a_, b_;
      pair_from_index(n, cbi, a_, b_);
      na = a_ - (a_ > pj_ ? 1 : 0);
      nb = b_ - (b_ > pj_ ? 1 : 0);
    }
    // the x' row of the candidate stays valid while the candidate is the same pair of rows
    int keep = 0;
    if (so.cand_cur && na >= 0) {
      const int ca = so.cand_cur[2 * b], cc = so.cand_cur[2 * b + 1];
      if (ca >= 0 && ca != pi_ && ca != pj_ && cc != pi_ && cc != pj_)
        keep = (ca - (ca > pj_ ? 1 : 0) == na && cc - (cc > pj_ ? 1 : 0) == nb) ? 1 : 0;
    }
    so.cand_next[2 * b] = na; so.cand_next[2 * b + 1] = nb;
    so.cand_run[b] = (na >= 0 && !keep) ? 1 : 0;
  }
}

// Topology key of a finished rollout for the duplicate filter of the sampling mode (reference utils.py:76 compares
// `topo_repr`, the rooted topology string whose children are ordered by their smallest leaf): a 64-bit hash of the
// same equivalence -- leaves hash their index, a join hashes the COMMUTATIVE sum of its children -- so two merge
// lists get the same key iff they build the same rooted, unordered, leaf-labelled tree (up to 2^-64 collisions).
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {    // splitmix64 finaliser
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27; x *= 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__global__ void k_topology_hash(const int* __restrict__ merges, unsigned long long* __restrict__ keys, int B, int T) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  unsigned long long hsh[256];                              // current rows (positions), T <= 256
  for (int i = 0; i < T; ++i) hsh[i] = mix64(0x9e3779b97f4a7c15ull * (unsigned long long)(i + 1));
  const int* m = merges + (size_t)b * (T - 1) * 2;
  int n = T;
  for (int s = 0; s < T - 1; ++s, --n) {
    const int i = min(max(m[2 * s], 0), n - 1), j = min(max(m[2 * s + 1], 0), n - 1);
    hsh[i] = mix64(hsh[i] + hsh[j] + 0x632be59bd9b4e019ull);
    for (int p = j; p < n - 1; ++p) hsh[p] = hsh[p + 1];   // position j is deleted (environment.py:764-768)
  }
  keys[b] = hsh[0];
}

// argmax of a given table (finetune_rl_search.py:145,159-160): ij_out [B][2], gap [B] optional
__global__ __launch_bounds__(256) void k_select_pair(const float* __restrict__ logits, int* __restrict__ ij_out,
                                                     float* __restrict__ gap_out, int n) {
  __shared__ float red_v[256];
  __shared__ int red_i[256];
  const int tid = threadIdx.x, b = blockIdx.x;
  const int np = num_pairs(n);
  const float* l = logits + (size_t)b * np;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int p = tid; p < np; p += 256) {
    const float v = l[p];
    if (v > best || (v == best && p < besti)) { best = v; besti = p; }
  }
  red_v[tid] = best; red_i[tid] = besti;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) {
      const float v2 = red_v[tid + s]; const int i2 = red_i[tid + s];
      if (v2 > red_v[tid] || (v2 == red_v[tid] && i2 < red_i[tid])) { red_v[tid] = v2; red_i[tid] = i2; }
    }
    __syncthreads();
  }
  const int bi = (red_i[0] >= 0 && red_i[0] < np) ? red_i[0] : 0;   // all-NaN table: stay in range
  const float bv = red_v[0];
  __syncthreads();
  float second = -INFINITY;
  for (int p = tid; p < np; p += 256)
    if (p != bi) second = fmaxf(second, l[p]);
  red_v[tid] = second;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (tid < s) red_v[tid] = fmaxf(red_v[tid], red_v[tid + s]);
    __syncthreads();
  }
  if (tid == 0) {
    int ci, cj;
    pair_from_index(n, bi, ci, cj);
    ij_out[2 * b] = ci; ij_out[2 * b + 1] = cj;
    if (gap_out) gap_out[b] = np > 1 ? bv - red_v[0] : 0.f;
  }
}

// live list: drop position j (merged row stays in position i's slot) -- environment.py:764-768
__global__ void k_update_live(int* __restrict__ live, int live_stride, const int* __restrict__ ij, int B, int n) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int pj = min(max(ij[2 * b + 1], 1), n - 1);
  int* l = live + (size_t)b * live_stride;
  for (int p = pj; p < n - 1; ++p) l[p] = l[p + 1];
}
__global__ void k_init_live(int* __restrict__ live, int live_stride, int B, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * live_stride) return;
  live[i] = i % live_stride;
}

// utils.get_score_indices_to_prev on the device (utils.py:213-251): idx int64 [B][P(n)]
__global__ void k_index_map(const int* __restrict__ ij_prev, long long* __restrict__ idx, int B, int n) {
  const int np = num_pairs(n);
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * np) return;
  const int b = (int)(i / np), p = (int)(i % np);
  const int ip = ij_prev[2 * b], jp = ij_prev[2 * b + 1];
  int ii, jj;
  pair_from_index(n, p, ii, jj);
  long long v;
  if (ii == ip) v = num_pairs(n + 1) + jj;
  else if (jj == ip) v = num_pairs(n + 1) + ii;
  else v = pair_index(n + 1, ii + (ii >= jp), jj + (jj >= jp));
  idx[i] = v;
}

// dense compaction for the API-compatible env.step: out[b][t] = (t == i ? merged : state[b][src])
__global__ void k_compact_rows(const float* __restrict__ state, const float* __restrict__ merged,
                               const int* __restrict__ ij, float* __restrict__ out, int n, long row_f4) {
  const int t = blockIdx.y, b = blockIdx.z;     // output row t in [0, n-1)
  const int pi = min(max(ij[2 * b], 0), n - 1), pj = min(max(ij[2 * b + 1], 0), n - 1);
  const int src = t < pj ? t : t + 1;
  const f32x4* s = (t == pi) ? reinterpret_cast<const f32x4*>(merged) + (size_t)b * row_f4
                             : reinterpret_cast<const f32x4*>(state) + ((size_t)b * n + src) * row_f4;
  f32x4* d = reinterpret_cast<f32x4*>(out) + ((size_t)b * (n - 1) + t) * row_f4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < row_f4; i += (long)gridDim.x * blockDim.x) d[i] = s[i];
}

// dense view of the live rows of a slot-layout state: out[b][p] = S[b][live[b][p]]  (environment.py:833-835)
__global__ void k_gather_rows(const float* __restrict__ S, const int* __restrict__ live, int live_stride,
                              float* __restrict__ out, int n, long row_f4, long b_f4) {
  const int p = blockIdx.y, b = blockIdx.z;
  const f32x4* src = reinterpret_cast<const f32x4*>(S) + (size_t)b * b_f4 + (size_t)live[(size_t)b * live_stride + p] * row_f4;
  f32x4* d = reinterpret_cast<f32x4*>(out) + ((size_t)b * n + p) * row_f4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < row_f4; i += (long)gridDim.x * blockDim.x) d[i] = src[i];
}

// state of batch element 0 copied to elements 1..B-1 (encode once, replicate: the sampling mode runs
// B rollouts of ONE alignment; the reference re-encodes it for every rollout)
__global__ void k_replicate(float* __restrict__ buf, long per_b_f4) {
  const f32x4* src = reinterpret_cast<const f32x4*>(buf);
  f32x4* dst = reinterpret_cast<f32x4*>(buf) + (size_t)(blockIdx.y + 1) * per_b_f4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < per_b_f4; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void k_replicate_u8(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int B, int L) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)B * L) dst[i] = src[i % L];
}
#endif  // NNJ_STEP0_TU
