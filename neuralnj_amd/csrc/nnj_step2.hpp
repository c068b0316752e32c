// nnj_step2.hpp -- the NJ step of the device-resident rollout with TWO passes over the live rows instead of four
// (restates reference environment.py:760-835 + model.py:102-155 for the merge and model.py:184-201 for the new
// pairs, like nnj_scorer.hpp / nnj_scorer16.hpp whose kernels it replaces inside nnj_rollout_*).
//
// A step used to stream the live rows four times: K' for the attention logits of the merged pair (k_agg_alpha), S for
// its context and the new row (k_agg_finish), S for the attention logits of the n-1 new pairs (k_inc_alpha16), S for
// their scores (k_inc_score*); each pass needs a softmax over all sites of the one before.  Two of the four go:
//
//  * the logits of the MERGED pair are never computed by a pass of their own.  aggregate(x_i, x_j) inside the scorer
//    (model.py:93 -> 102) and inside env.step (environment.py:791) is the same function of the same rows, and its
//    logits  a_r = sum_c (A^T x_ij[c]) . S_r[c]  (x . a0 and every other r-independent term cancel in the softmax
//    over r) do not depend on the step at which they are evaluated.  The pair picked by the argmax is either
//      - one of the n-1 pairs scored in THIS step: its logits against every other row are the row of k_step_alpha's
//        output that k_step_softmax has just summed (`lam`), or
//      - the best pair of the older part of the table, known one step ahead (k_assemble_argmax names it: `cand`);
//        k_step_alpha carries its logits along (one x' row per alignment, one dot product per row and site);
//    k_assemble_argmax turns whichever applies into the attention weights `am` of the next merge.  Anything else
//    (forced or sampled picks outside those two, the first step) takes the fallback: k_pair_xp + k_agg_dot +
//    k_agg_am, per alignment, only where needed.
//  * the merged row is produced INSIDE the alpha pass of the new pairs: per site the waves that hold the rows add up
//    am_r S_r[c] (products through LDS, column sums), one wave finishes the row -- x_ij gate, W_g, mix, W_h --
//    writes S_m[c], U_m[c] in place and hands them to the others through LDS.
//
// Numbering inside a step with n rows after the merge, merged row at position m: the n-1 OTHER rows are q = 0..n-2,
// r = q + (q >= m); pairs, image rows, alpha planes and score partials are all indexed by q (the pair (m, m) the
// reference computes and never reads is gone: n = 17, 33, 49 need one 16-pair tile less).
#pragma once
#include "nnj_scorer16.hpp"

struct StepIO {
  const int* ij;              // [B][2] the merge (i, j), positions of the list BEFORE the merge (n+1 rows)
  const int* live_old;        // [B][live_stride] that list (slot of j; the new list is rs.live)
  const float* am;            // [B][64] attention weights of the merged pair by NEW position (0 at m and beyond n)
  float* alpha_part;          // [B][blocks][64 q][64 q'] logits of the new pairs, partial over the sites
  float* S_w;                 // rs.S, rs.U, rs.beta_part, writable (the merged row goes in place)
  float* U_w;
  float* beta_w;
  int beta_n;                 // beta partials per row in use (entries beyond the writer's block count are zeroed)
  const int* cand;            // [B][2] best OLD pair of the coming table by new position, (-1, -1) = none
  const int* cand_run;        // [B] 1 = the x' row in Xc belongs to another pair: the workgroups rewrite their sites first
  float* Xc;                  // [B][C][64] x' = A^T x of that pair
  float* acand_part;          // [B][blocks][64]: [q] = sum_c x'_cand . S_r(q), [63] = ... . S_m
};

// beta_slot[b][slot] = sum of the row's beta partials + C t0 (the per-row bias of the attention logits: it changes only
// when the row does).  One thread per row; run once when the two-pass steps take over from the other kernels.
__global__ void k_beta_sum(const float* __restrict__ beta_part, int stride, int nb, float ct0, float* __restrict__ beta_slot,
                           int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const float* bp = beta_part + (size_t)i * stride;
  float s = 0.f;
#pragma unroll 32          // (32 loads in flight, added in order: one thread walks up to C / 32 partials)
  for (int t = 0; t < nb; ++t) s += bp[t];
  beta_slot[i] = s + ct0;
}

// x' = A^T gate(S_a, S_b) of ONE designated pair per alignment (positions `pair[b]`, list `live`), tokens = sites.
// Used for the candidate pair (skipped while `skip[b]` says the row in Xout is still the right one) and by the
// fallback for the merged pair.  grid (ceil(C/128), B), 4 waves x 32 sites.
__global__ __launch_bounds__(256) void k_pair_xp(const float* __restrict__ S, const float* __restrict__ U, long bstride,
                                                 const int* __restrict__ live, int live_stride, ScorerW w,
                                                 const int* __restrict__ pair, const int* __restrict__ run,
                                                 float* __restrict__ Xout, int n, int C) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int b = blockIdx.y;
  if (run && !run[b]) return;
  const int pa = pair[2 * b], pb = pair[2 * b + 1];
  if (pa < 0 || pb < 0 || pa >= n || pb >= n) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  stage_weight_b6_T<64>(smem, w.A, 64, tid, 256);
  __syncthreads();
  const int c = (blockIdx.x * 4 + wave) * 32 + (lane & 31);
  if ((blockIdx.x * 4 + wave) * 32 >= C) return;
  const bool valid = c < C;
  const size_t bo = (size_t)b * bstride;
  const size_t oa = bo + ((size_t)live[(size_t)b * live_stride + pa] * C + (valid ? c : 0)) * 64;
  const size_t ob = bo + ((size_t)live[(size_t)b * live_stride + pb] * C + (valid ? c : 0)) * 64;
  f32x16 x[1][2];
  {
    f32x16 si[2], sj[2], ui[2], uj[2];
    load_token64(si, S + oa, valid, hh); load_token64(sj, S + ob, valid, hh);
    load_token64(ui, U + oa, valid, hh); load_token64(uj, U + ob, valid, hh);
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
  f32x16 o[1][2];
  linear6_T_nb<2, 2, 1>(o, x, smem, lane);
  store_token64(o[0], Xout + ((size_t)b * C + c) * 64, valid, hh);
}

// ------------------------------------------------------------------ k_step_alpha
// Per site c (NG waves of 16 rows share it, lane = (row l15, feature quarter kq), row q = 16 tl + l15):
//   1. products am_r S_r[c] of the wave's rows -> LDS, column sums of its 16 rows -> xgp[tl]
//   2. wave 0 of the site, one feature per lane: x_g = sum xgp, x_ij = gate(S_i, S_j), g = W_g x_g + b_g (matrix
//      pipe: the vector as the B operand of all 16 columns), S_m = x_ij + sigmoid(g) (x_g - x_ij), U_m = W_h S_m;
//      S_m[c], U_m[c] to HBM (slot of i, in place) and to LDS; beta partial u . S_m; x'_cand . S_m
//   3. everyone: x = gate(S_m, S_r), x' = A^T x, acc[q'][q] += S_q' . x' against the site's image (as k_inc_alpha16),
//      and x'_cand[c] . S_r[c]
template <int NG, int NW>
__global__ __launch_bounds__(64 * NW) void k_step_alpha(RowSet rs, ScorerW w, StepIO io, int n, int C, int cs,
                                                        int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NSLOT = NW / NG;
  constexpr int IMG = 16 * NG * 64 * NPL / 2;              // floats of a site image = of the products of its rows
  constexpr int XS = 64 * NG + 64 * 6;                     // per-slot scratch: xgp[NG][64] | xg | g | S_m | U_m | x'_cand | spare
  float* At_l = smem;                                      // A^T
  float* Wh_l = smem + IMG64;
  float* Wg_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave % NSLOT, tl = wave / NSLOT;        // (the waves of a site on different SIMDs -- slot = wave / NG -- measured slower: 54.9 vs 52.4 ms per rollout)
  float* img = smem + 3 * IMG64 + slot * IMG;
  float* xs = smem + 3 * IMG64 + NSLOT * IMG + slot * XS;
  int* cnt0 = reinterpret_cast<int*>(smem + 3 * IMG64 + NSLOT * IMG + NSLOT * XS);
  float* cv = reinterpret_cast<float*>(cnt0 + 16);
  float* epi = cv + SCORER_CONSTS;                         // [NSLOT][64 + 2] sums of the epilogue
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(At_l, w.imgAt, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * NW);
  stage_scorer_consts(cv, w, tid);
  if (tid < NSLOT) cnt0[tid] = 0;
  __syncthreads();
  int* cnt = cnt0 + slot;
  int epoch = 0;
  const int m = min(max(io.ij[2 * b], 0), n - 1);
  const int j_old = min(max(io.ij[2 * b + 1], 0), n);
  const int q = 16 * tl + l15;
  const bool qv = q < n - 1;
  const int r = qv ? q_to_r(q, m) : (m == 0 ? 1 : 0);        // lanes beyond the rows read a row that is not written here
  const float sgn = r < m ? 1.0f : -1.0f;
  const float aw = qv ? io.am[(size_t)b * 64 + r] : 0.f;
  const size_t bo = (size_t)b * rs.bstride;
  const int slot_m = slot_of(rs, b, m);
  const int slot_j = io.live_old[(size_t)b * rs.live_stride + j_old];
  const float* Sr = rs.S + bo + (size_t)slot_of(rs, b, r) * C * 64;
  const size_t om = bo + (size_t)slot_m * C * 64, oj = bo + (size_t)slot_j * C * 64;
  const bool pairs = n > 2;                                // with two rows left the one new pair has no context (model.py:111)
  const int ca = io.cand ? io.cand[2 * b] : -1;
  const bool has_cand = ca >= 0 && pairs;
  float* Xc = io.Xc + (size_t)b * C * 64;
  if (has_cand && io.cand_run[b]) {
    // the candidate changed: x' = A^T gate(S_a, S_b) of its rows for this workgroup's sites, 16 sites per wave and pass
    // (tokens = sites), written to Xc and read back below (the lines are in nobody's L1 yet)
    const int cb2 = io.cand[2 * b + 1];
    const float* Sa = rs.S + bo + (size_t)slot_of(rs, b, ca) * C * 64;
    const float* Sb = rs.S + bo + (size_t)slot_of(rs, b, cb2) * C * 64;
    const float* Ua = rs.U + bo + (size_t)slot_of(rs, b, ca) * C * 64;
    const float* Ub = rs.U + bo + (size_t)slot_of(rs, b, cb2) * C * 64;
    for (int cc0 = c0 + 16 * wave; cc0 < c1; cc0 += 16 * NW) {
      const int cc = cc0 + l15;
      const bool ok = cc < c1;
      const size_t o = (size_t)(ok ? cc : c1 - 1) * 64;
      V64 sa, sb, ua, ub, x;
      load_v64(sa, Sa + o, kq); load_v64(sb, Sb + o, kq);
      load_v64(ua, Ua + o, kq); load_v64(ub, Ub + o, kq);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(cv + 16 * mt + 4 * kq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = sigmoid_l2(ua.t[mt][e] - ub.t[mt][e] + b4[e]);
          x.t[mt][e] = sb.t[mt][e] + z * (sa.t[mt][e] - sb.t[mt][e]);
        }
      }
      V64 xp;
      lds_wait_all();
      linear_t16p<4, false, false>(xp.t, x, At_l, nullptr, lane);
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(Xc + o + 16 * mt + 4 * kq) = xp.t[mt];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  const float u_l = w.u[lane];                             // wave 0 of a site: one feature per lane
  const float bh_l = cv[lane];
  f32x4 acc[NG];
#pragma unroll
  for (int mt = 0; mt < NG; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cacc = 0.f, beta_acc = 0.f, cm_acc = 0.f;
  u32x4* im4 = reinterpret_cast<u32x4*>(img);
  constexpr int PL = 16 * NG * 8;
  float* xgp = xs;
  float* v_xg = xs + 64 * NG;
  float* v_g = v_xg + 64;
  float* v_sm = v_g + 64;
  float* v_um = v_sm + 64;
  float* v_xc = v_um + 64;
  V64 sr;
  float si = 0.f, sj = 0.f, ui = 0.f, uj = 0.f, xc_l = 0.f;
  int c = c0 + slot;
  if (c < c1) {
    load_v64(sr, Sr + (size_t)c * 64, kq);
    if (tl == 0) {
      si = rs.S[om + (size_t)c * 64 + lane]; sj = rs.S[oj + (size_t)c * 64 + lane];
      ui = rs.U[om + (size_t)c * 64 + lane]; uj = rs.U[oj + (size_t)c * 64 + lane];
      if (has_cand) xc_l = Xc[(size_t)c * 64 + lane];
    }
  }
  for (; c < c1; c += NSLOT) {
    asm volatile("" ::: "memory");
    // ---- 1. am_r S_r[c]: the wave's 16 rows as fp32 rows of the (dead) image buffer, 16-byte chunk ch of row k stored
    // at chunk ch ^ k: the row stores and the column reads are conflict free and a column read is one XOR with a
    // constant away from the lane's base address; then the column sums of those rows
    if constexpr (NG > 1) group_barrier_lds<NG>(cnt, epoch, status);     // everyone is done with the previous site's image
    {
      float* pw = img + (16 * tl + l15) * 64;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 p = sr.t[mt] * aw;
        *reinterpret_cast<f32x4*>(pw + 4 * ((4 * mt + kq) ^ l15)) = p;
      }
      asm volatile("" ::: "memory");
      const float* pr = img + 16 * tl * 64;
      float s = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += pr[k * 64 + (lane ^ (4 * k))];
      xgp[tl * 64 + lane] = s;
    }
    if constexpr (NG > 1) group_barrier_lds<NG>(cnt, epoch, status);     // all column sums are in xgp
    else asm volatile("" ::: "memory");
    // ---- 2. the merged row of this site
    if (tl == 0) {
      float xg = 0.f;
#pragma unroll
      for (int t = 0; t < NG; ++t) xg += xgp[t * 64 + lane];
      const float z = sigmoid_l2(ui - uj + bh_l);
      const float xij = sj + z * (si - sj);
      v_xg[lane] = xg;
      asm volatile("" ::: "memory");
      V64 g;
      {
        V64 xv;
        load_v64(xv, v_xg, kq);
        Frag3 bf[2];
        split_8(bf[0], xv.t[0], xv.t[1]);
        split_8(bf[1], xv.t[2], xv.t[3]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) g.t[mt] = *reinterpret_cast<const f32x4*>(cv + 64 + 16 * mt + 4 * kq);
        lds_wait_all();
        linear_t16p_core<4>(g.t, bf, Wg_l, lane, [] {});
      }
      if (l15 == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(v_g + 16 * mt + 4 * kq) = g.t[mt];
      }
      asm volatile("" ::: "memory");
      const float wg = sigmoid_l2(v_g[lane]);
      const float smd = xij + wg * (xg - xij);
      io.S_w[om + (size_t)c * 64 + lane] = smd;
      beta_acc += u_l * smd;
      cm_acc += xc_l * smd;
      v_sm[lane] = smd;
      if (has_cand) v_xc[lane] = xc_l;
      asm volatile("" ::: "memory");
      V64 um;
      {
        V64 xv;
        load_v64(xv, v_sm, kq);
        Frag3 bf[2];
        split_8(bf[0], xv.t[0], xv.t[1]);
        split_8(bf[1], xv.t[2], xv.t[3]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) um.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        lds_wait_all();
        linear_t16p_core<4>(um.t, bf, Wh_l, lane, [] {});
      }
      if (l15 == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          *reinterpret_cast<f32x4*>(v_um + 16 * mt + 4 * kq) = um.t[mt];
          *reinterpret_cast<f32x4*>(io.U_w + om + (size_t)c * 64 + 16 * mt + 4 * kq) = um.t[mt];
        }
      }
      // next site's rows of the merged pair (behind this site's work)
      const int cn = c + NSLOT < c1 ? c + NSLOT : c;
      si = rs.S[om + (size_t)cn * 64 + lane]; sj = rs.S[oj + (size_t)cn * 64 + lane];
      ui = rs.U[om + (size_t)cn * 64 + lane]; uj = rs.U[oj + (size_t)cn * 64 + lane];
      if (has_cand) xc_l = Xc[(size_t)cn * 64 + lane];
    }
    if constexpr (NG > 1) group_barrier_lds<NG>(cnt, epoch, status);     // S_m, U_m (and x'_cand) of the site are in LDS
    else asm volatile("" ::: "memory");
    if (!pairs) {
      const int cn = c + NSLOT < c1 ? c + NSLOT : c;
      load_v64(sr, Sr + (size_t)cn * 64, kq);
      continue;
    }
    // ---- 3. the new pairs (m, r)
    // fp16 pieces of S_r: image row, and B operand of U_r = W_h S_r (split here, not in front of the merged row: the
    // pieces would be live across it)
    Frag3 sf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
    V64 x;
    {
      V64 sm, um, ur;                                      // (U_r = W_h S_r issued by the partners of wave 0 while they wait for
      load_v64(sm, v_sm, kq);                              //  the merged row: measured slower, 56.0 vs 52.7 ms per rollout)
      load_v64(um, v_um, kq);
      gate_init16(ur, um, cv, sgn, kq);
      lds_wait_all();
      linear_t16p_core<4>(ur.t, sf, Wh_l, lane, [] {});
      gate16(x, sr, ur, sm);
    }
    if (has_cand) {                                        // x'_cand[c] . S_r[c] (this lane's 16 features)
      V64 xc;
      load_v64(xc, v_xc, kq);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) cacc += xc.t[mt][e] * sr.t[mt][e];
    }
    // row q of the image: chunk 4*ks + kq = this lane's tiles 2ks, 2ks+1 (the products are dead: every wave has
    // summed its own rows, and the only other reader of this buffer is behind the next barrier)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int o = q * 8 + wswz6<8>(q, 4 * ks + kq);
      im4[o] = sf[ks].h; im4[PL + o] = sf[ks].m;
    }
    if constexpr (NG > 1) group_barrier_lds<NG>(cnt, epoch, status);     // all rows are in the image
    const int cn = c + NSLOT < c1 ? c + NSLOT : c;                         // prefetch behind the MFMAs (last: harmless reload)
    load_v64(sr, Sr + (size_t)cn * 64, kq);
    V64 xp;
    linear_t16p<4, false, false>(xp.t, x, At_l, nullptr, lane);          // x' = A^T x
    linear_t16<NG, true, false>(acc, xp, img, nullptr, lane);           // acc[q'][pair] += S_q' . x'
  }
  // ---- epilogue: one partial set per WORKGROUP, the slots added in slot order (bitwise reproducible)
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // 4096 floats (NSLOT * IMG >= 4096)
  for (int i = tid; i < 4096; i += 64 * NW) red[i] = 0.f;
  cacc += __shfl_xor(cacc, 16);
  cacc += __shfl_xor(cacc, 32);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { beta_acc += __shfl_xor(beta_acc, o); cm_acc += __shfl_xor(cm_acc, o); }
  __syncthreads();
  if (kq == 0) epi[slot * 66 + q] = cacc;
  if (tl == 0 && lane == 0) { epi[slot * 66 + 64] = beta_acc; epi[slot * 66 + 65] = cm_acc; }
  for (int s_ = 0; s_ < NSLOT; ++s_) {
    if (slot == s_) {
#pragma unroll
      for (int mt = 0; mt < NG; ++mt) {
        f32x4* d4 = reinterpret_cast<f32x4*>(red + q * 64 + 16 * mt + 4 * kq);
        *d4 = *d4 + acc[mt];
      }
    }
    __syncthreads();
  }
  float* dst = io.alpha_part + ((size_t)b * gridDim.x + sc) * 4096;
  for (int i = tid; i < 1024; i += 64 * NW)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(red)[i];
  if (tid < 64) {
    float v = 0.f;
    const int idx = tid == 63 ? 65 : (tid < 16 * NG ? tid : -1);         // (q <= 62: entry 63 carries the merged row's)
    if (idx >= 0)
      for (int s_ = 0; s_ < NSLOT; ++s_) v += epi[s_ * 66 + idx];
    if (io.acand_part) io.acand_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  } else if (tid == 64) {
    float v = 0.f;
    for (int s_ = 0; s_ < NSLOT; ++s_) v += epi[s_ * 66 + 64];
    float* bp = io.beta_w + ((size_t)b * (rs.bstride / ((long)C * 64)) + slot_m) * rs.ntile32;
    bp[sc] = v;
    if (sc == 0)
      for (int k = gridDim.x; k < io.beta_n; ++k) bp[k] = 0.f;          // entries of a row that had more partials
  }
}

// ------------------------------------------------------------------ k_step_alpha_w
// The same step with ONE WAVE PER SITE walking the NT 16-row tiles itself (as k_inc_score_w does for the scores):
// products, column sums, the merged row and the site's image are private to the wave, so nothing in the loop
// synchronises with another wave -- k_step_alpha pays four group barriers per site and its partners of wave 0 idle
// while it finishes the merged row (measured: waiting, not issue, bounds that kernel: VALU 0.15, matrix pipe 0.29 of
// the cycles).  NW waves per workgroup at <= 512 / ceil(NW / 4) registers; the other waves of the SIMD cover a wave's
// dependent chains.
// IL: step 3 over the NT tiles STAGE BY STAGE (linear_t16p_multi: every weight / image fragment read once per site, NT
// independent accumulator chains; see k_inc_score_wi) and the image rows written from the pieces the U_r product uses
// (no second split).  Bit-identical results.
template <int NT, int NW, bool IL = false>
__global__ __launch_bounds__(64 * NW) void k_step_alpha_w(RowSet rs, ScorerW w, StepIO io, int n, int C, int cs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int IMG = 16 * NT * 64 * NPL / 2;              // floats of a site image = of the products of its rows
  constexpr int XS = 64 * 5;                               // per-wave scratch: xg | g | S_m | U_m | spare
  float* At_l = smem;                                      // A^T
  float* Wh_l = smem + IMG64;
  float* Wg_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* img = smem + 3 * IMG64 + wave * IMG;
  float* xs = smem + 3 * IMG64 + NW * IMG + wave * XS;
  float* cv = smem + 3 * IMG64 + NW * IMG + NW * XS;
  float* epi = cv + SCORER_CONSTS;                         // [NW][64 + 2] sums of the epilogue
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(At_l, w.imgAt, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * NW);
  stage_scorer_consts(cv, w, tid);
  __syncthreads();
  const int m = min(max(io.ij[2 * b], 0), n - 1);
  const int j_old = min(max(io.ij[2 * b + 1], 0), n);
  const size_t bo = (size_t)b * rs.bstride;
  const int slot_m = slot_of(rs, b, m);
  const int slot_j = io.live_old[(size_t)b * rs.live_stride + j_old];
  const size_t om = bo + (size_t)slot_m * C * 64, oj = bo + (size_t)slot_j * C * 64;
  const float* Sr[NT];
  float sgn[NT], aw[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q = 16 * t + l15;
    const bool qv = q < n - 1;
    const int r = qv ? q_to_r(q, m) : (m == 0 ? 1 : 0);      // lanes beyond the rows read a row that is not written here
    sgn[t] = r < m ? 1.0f : -1.0f;
    aw[t] = qv ? io.am[(size_t)b * 64 + r] : 0.f;
    Sr[t] = rs.S + bo + (size_t)slot_of(rs, b, r) * C * 64;
  }
  const int ca = io.cand ? io.cand[2 * b] : -1;
  const bool has_cand = ca >= 0;
  float* Xc = io.Xc + (size_t)b * C * 64;
  if (has_cand && io.cand_run[b]) {
    // the candidate changed: x' = A^T gate(S_a, S_b) of its rows for this workgroup's sites (see k_step_alpha)
    const int cb2 = io.cand[2 * b + 1];
    const float* Sa = rs.S + bo + (size_t)slot_of(rs, b, ca) * C * 64;
    const float* Sb = rs.S + bo + (size_t)slot_of(rs, b, cb2) * C * 64;
    const float* Ua = rs.U + bo + (size_t)slot_of(rs, b, ca) * C * 64;
    const float* Ub = rs.U + bo + (size_t)slot_of(rs, b, cb2) * C * 64;
    for (int cc0 = c0 + 16 * wave; cc0 < c1; cc0 += 16 * NW) {
      const int cc = cc0 + l15;
      const bool ok = cc < c1;
      const size_t o = (size_t)(ok ? cc : c1 - 1) * 64;
      V64 sa, sb, ua, ub, x;
      load_v64(sa, Sa + o, kq); load_v64(sb, Sb + o, kq);
      load_v64(ua, Ua + o, kq); load_v64(ub, Ub + o, kq);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(cv + 16 * mt + 4 * kq);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float z = sigmoid_l2(ua.t[mt][e] - ub.t[mt][e] + b4[e]);
          x.t[mt][e] = sb.t[mt][e] + z * (sa.t[mt][e] - sb.t[mt][e]);
        }
      }
      V64 xp;
      lds_wait_all();
      linear_t16p<4, false, false>(xp.t, x, At_l, nullptr, lane);
      if (ok) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(Xc + o + 16 * mt + 4 * kq) = xp.t[mt];
      }
    }
    __threadfence_block();
    __syncthreads();
  }
  const float u_l = w.u[lane];
  const float bh_l = cv[lane];
  f32x4 acc[NT][NT];                                       // [pair tile][row tile]
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float cacc[NT], beta_acc = 0.f, cm_acc = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) cacc[t] = 0.f;
  u32x4* im4 = reinterpret_cast<u32x4*>(img);
  constexpr int PL = 16 * NT * 8;
  float* v_xg = xs;
  float* v_g = xs + 64;
  float* v_sm = xs + 128;
  float* v_um = xs + 192;
  float* v_xc = xs + 256;
  for (int c = c0 + wave; c < c1; c += NW) {
    asm volatile("" ::: "memory");
    V64 sr[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) load_v64(sr[t], Sr[t] + (size_t)c * 64, kq);
    const float si = rs.S[om + (size_t)c * 64 + lane], sj = rs.S[oj + (size_t)c * 64 + lane];
    const float ui = rs.U[om + (size_t)c * 64 + lane], uj = rs.U[oj + (size_t)c * 64 + lane];
    const float xc_l = has_cand ? Xc[(size_t)c * 64 + lane] : 0.f;
    // ---- 1. am_r S_r[c] as fp32 rows of the (dead) image buffer (16-byte chunk ch of row k at chunk ch ^ (k & 15)),
    // then the column sums over all rows: x_g, one feature per lane
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float* pw = img + (16 * t + l15) * 64;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 p = sr[t].t[mt] * aw[t];
        *reinterpret_cast<f32x4*>(pw + 4 * ((4 * mt + kq) ^ l15)) = p;
      }
    }
    asm volatile("" ::: "memory");
    float xg = 0.f;
#pragma unroll
    for (int k8 = 0; k8 < 2 * NT; ++k8) {                  // eight reads in flight at a time (all 16 NT would take as many registers)
#pragma unroll
      for (int k = 8 * k8; k < 8 * k8 + 8; ++k) xg += img[k * 64 + (lane ^ (4 * (k & 15)))];
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- 2. the merged row of this site (see k_step_alpha)
    const float z = sigmoid_l2(ui - uj + bh_l);
    const float xij = sj + z * (si - sj);
    v_xg[lane] = xg;
    asm volatile("" ::: "memory");
    {
      V64 g, sm, um;
      {
        V64 xv;
        load_v64(xv, v_xg, kq);
        Frag3 bf[2];
        split_8(bf[0], xv.t[0], xv.t[1]);
        split_8(bf[1], xv.t[2], xv.t[3]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) g.t[mt] = *reinterpret_cast<const f32x4*>(cv + 64 + 16 * mt + 4 * kq);
        lds_wait_all();
        linear_t16p_core<4>(g.t, bf, Wg_l, lane, [] {});
      }
      if (l15 == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(v_g + 16 * mt + 4 * kq) = g.t[mt];
      }
      asm volatile("" ::: "memory");
      const float wg = sigmoid_l2(v_g[lane]);
      const float smd = xij + wg * (xg - xij);
      io.S_w[om + (size_t)c * 64 + lane] = smd;
      beta_acc += u_l * smd;
      cm_acc += xc_l * smd;
      v_sm[lane] = smd;
      asm volatile("" ::: "memory");
      load_v64(sm, v_sm, kq);
      Frag3 bf[2];
      split_8(bf[0], sm.t[0], sm.t[1]);
      split_8(bf[1], sm.t[2], sm.t[3]);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) um.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      lds_wait_all();
      linear_t16p_core<4>(um.t, bf, Wh_l, lane, [] {});    // every column is U_m: the lane holds its 16 features as it stands
      if (l15 == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          *reinterpret_cast<f32x4*>(v_um + 16 * mt + 4 * kq) = um.t[mt];
          *reinterpret_cast<f32x4*>(io.U_w + om + (size_t)c * 64 + 16 * mt + 4 * kq) = um.t[mt];
        }
      }
    }
    if (n <= 2) continue;                                  // with two rows left the one new pair has no context (model.py:111)
    // ---- 3. the new pairs (m, r).  First the image rows of all tiles (fp16 pieces of S_r), then tile by tile: the
    // pieces again (split twice: kept, they and the x of every tile would not fit the registers), U_r = W_h S_r, gate,
    // x' = A^T x, the products with the image; and x'_cand[c] . S_r[c]
    if (has_cand) v_xc[lane] = xc_l;
    if constexpr (IL) {
      V64 x[NT];
      {
        V64 ur[NT];
        Frag3 sf[NT][2];
        {
          V64 um;
          load_v64(um, v_um, kq);
#pragma unroll
          for (int t = 0; t < NT; ++t) gate_init16(ur[t], um, cv, sgn[t], kq);
        }
        lds_wait_all();
        linear_t16p_multi<NT>(ur, sf, Wh_l, lane, [&](auto ki) {
          constexpr int ks = decltype(ki)::value;
#pragma unroll
          for (int t = 0; t < NT; ++t) split_8(sf[t][ks], sr[t].t[2 * ks], sr[t].t[2 * ks + 1]);
        });
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int q = 16 * t + l15;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const int o = q * 8 + wswz6<8>(q, 4 * ks + kq);
            im4[o] = sf[t][ks].h; im4[PL + o] = sf[t][ks].m;
          }
        }
        V64 sm;
        load_v64(sm, v_sm, kq);
#pragma unroll
        for (int t = 0; t < NT; ++t) gate16(x[t], sr[t], ur[t], sm);
      }
      if (has_cand) {
        V64 xcv;
        load_v64(xcv, v_xc, kq);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 4; ++e) cacc[t] += xcv.t[mt][e] * sr[t].t[mt][e];
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      V64 xp[NT];
      {
        Frag3 bx[NT][2];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) xp[t].t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        lds_wait_all();
        linear_t16p_multi<NT>(xp, bx, At_l, lane, [&](auto ki) {
          constexpr int ks = decltype(ki)::value;
#pragma unroll
          for (int t = 0; t < NT; ++t) split_8(bx[t][ks], x[t].t[2 * ks], x[t].t[2 * ks + 1]);
        });
      }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      static_for<0, 2>([&](auto ki) {
        constexpr int ks = decltype(ki)::value;
        Frag3 bp[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) split_8(bp[t], xp[t].t[2 * ks], xp[t].t[2 * ks + 1]);
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
          const int row = 16 * mt + l15;
          const int o = row * 8 + wswz6<8>(row, 4 * ks + kq);
          Frag3 a;
          a.h = im4[o]; a.m = im4[PL + o];
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][mt] = mfma16_f16(a.m, bp[t].h, acc[t][mt]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][mt] = mfma16_f16(a.h, bp[t].m, acc[t][mt]);
#pragma unroll
          for (int t = 0; t < NT; ++t) acc[t][mt] = mfma16_f16(a.h, bp[t].h, acc[t][mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      });
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      continue;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      Frag3 sf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr[t].t[2 * ks], sr[t].t[2 * ks + 1]);
      const int q = 16 * t + l15;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o = q * 8 + wswz6<8>(q, 4 * ks + kq);
        im4[o] = sf[ks].h; im4[PL + o] = sf[ks].m;
      }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      V64 x;
      {
        Frag3 sf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr[t].t[2 * ks], sr[t].t[2 * ks + 1]);
        V64 ur;
        {
          V64 um;                                          // S_m, U_m of the site are re-read from LDS per tile: kept in
          load_v64(um, v_um, kq);                          // registers across the tiles they cost 32 of the 256
          gate_init16(ur, um, cv, sgn[t], kq);
        }
        lds_wait_all();
        linear_t16p_core<4>(ur.t, sf, Wh_l, lane, [] {});
        V64 sm;
        load_v64(sm, v_sm, kq);
        gate16(x, sr[t], ur, sm);
      }
      if (has_cand) {
        V64 xcv;
        load_v64(xcv, v_xc, kq);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) cacc[t] += xcv.t[mt][e] * sr[t].t[mt][e];
      }
      V64 xp;
      lds_wait_all();
      linear_t16p<4, false, false>(xp.t, x, At_l, nullptr, lane);         // x' = A^T x
      linear_t16<NT, true, false>(acc[t], xp, img, nullptr, lane);        // acc[q'][pair] += S_q' . x'
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- epilogue: one partial set per WORKGROUP, the waves added in wave order (bitwise reproducible)
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // 4096 floats (NW * IMG >= 4096)
  for (int i = tid; i < 4096; i += 64 * NW) red[i] = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) { cacc[t] += __shfl_xor(cacc[t], 16); cacc[t] += __shfl_xor(cacc[t], 32); }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { beta_acc += __shfl_xor(beta_acc, o); cm_acc += __shfl_xor(cm_acc, o); }
  __syncthreads();
  if (kq == 0) {
#pragma unroll
    for (int t = 0; t < NT; ++t) epi[wave * 66 + 16 * t + l15] = cacc[t];
  }
  if (lane == 0) { epi[wave * 66 + 64] = beta_acc; epi[wave * 66 + 65] = cm_acc; }
  for (int s_ = 0; s_ < NW; ++s_) {
    if (wave == s_) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < NT; ++mt) {
          f32x4* d4 = reinterpret_cast<f32x4*>(red + (16 * t + l15) * 64 + 16 * mt + 4 * kq);
          *d4 = *d4 + acc[t][mt];
        }
    }
    __syncthreads();
  }
  float* dst = io.alpha_part + ((size_t)b * gridDim.x + sc) * 4096;
  for (int i = tid; i < 1024; i += 64 * NW)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(red)[i];
  if (tid < 64) {
    float v = 0.f;
    const int idx = tid == 63 ? 65 : (tid < 16 * NT ? tid : -1);         // (q <= 62: entry 63 carries the merged row's)
    if (idx >= 0)
      for (int s_ = 0; s_ < NW; ++s_) v += epi[s_ * 66 + idx];
    if (io.acand_part) io.acand_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  } else if (tid == 64) {
    float v = 0.f;
    for (int s_ = 0; s_ < NW; ++s_) v += epi[s_ * 66 + 64];
    float* bp = io.beta_w + ((size_t)b * (rs.bstride / ((long)C * 64)) + slot_m) * rs.ntile32;
    bp[sc] = v;
    if (sc == 0)
      for (int k = gridDim.x; k < io.beta_n; ++k) bp[k] = 0.f;          // entries of a row that had more partials
  }
}

// ------------------------------------------------------------------ k_step_softmax
// alpha[b][q][q'] = softmax_q'((lam + beta_r(q')) / sqrt(64 C)) over the rows other than m and r(q); lam = the summed
// partials of k_step_alpha, kept for the next merge.  The last wave of the grid also sums the beta partials
// k_step_alpha wrote for the merged row into beta_slot (no pair of this step has m in its context; the table kernel
// and later steps do).  One wave per pair q, lane = q'.  grid (16, B).
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_step_softmax(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                      const float* __restrict__ alpha_part, float* __restrict__ alpha,
                                                      float* __restrict__ lam, float* __restrict__ beta_slot, int nblk,
                                                      int n, int C) {
  // SPLIT (small batches: k_step_alpha ran many workgroups per alignment, i.e. many partials): one pair per
  // workgroup, its four waves sum every 4th partial each and the four sums are added in wave order; grid (64, B)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = SPLIT ? blockIdx.x : blockIdx.x * 4 + wave, b = blockIdx.y;
  const int m = min(max(ij_prev[2 * b], 0), n - 1);
  const int nrow = (int)(rs.bstride / ((long)C * 64));
  if (q == 63 && (!SPLIT || wave == 0)) {                  // (q <= 62 are pairs: n - 1 <= 63)
    const float* bp = rs.beta_part + ((size_t)b * nrow + slot_of(rs, b, m)) * rs.ntile32;
    float s = 0.f;
    for (int t = lane; t < nblk; t += 64) s += bp[t];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) beta_slot[(size_t)b * nrow + slot_of(rs, b, m)] = s + (float)C * w.t0;
  }
  float a = 0.f;
  if (SPLIT) {
    // thread (phase tid >> 4, column group tid & 15) adds every 16th partial set, four columns at a time: 16 independent
    // 16-byte loads in flight (one round trip for 256 sets), the 16 phase sums of a column added in phase order
    __shared__ float psum[16][64];
    {
      const int cg = threadIdx.x & 15, ph = threadIdx.x >> 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 16
      for (int sc = ph; sc < nblk; sc += 16)
        v += *reinterpret_cast<const f32x4*>(alpha_part + (((size_t)b * nblk + sc) * 64 + q) * 64 + 4 * cg);
      *reinterpret_cast<f32x4*>(&psum[ph][4 * cg]) = v;
    }
    __syncthreads();
    if (wave != 0) return;
    a = psum[0][lane];
#pragma unroll
    for (int ph = 1; ph < 16; ++ph) a += psum[ph][lane];
  } else {
#pragma unroll 8          // independent loads in flight; the additions stay in order
    for (int sc = 0; sc < nblk; ++sc) a += alpha_part[(((size_t)b * nblk + sc) * 64 + q) * 64 + lane];
  }
  lam[((size_t)b * 64 + q) * 64 + lane] = a;
  const bool in = q < n - 1 && lane < n - 1 && lane != q;
  float v = -INFINITY;
  if (in) v = (a + beta_slot[(size_t)b * nrow + slot_of(rs, b, q_to_r(lane, m))]) * (1.0f / sqrtf(64.0f * (float)C));
  float mx = v;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  const float e = in ? expf(v - mx) : 0.f;
  float s = e;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  alpha_store(alpha, (long)gridDim.y * 4096, ((size_t)b * 64 + q) * 64 + lane, (s > 0.f) ? e / s : 0.f);
}

// ------------------------------------------------------------------ fallback for the merged pair's weights
// part[b][chunk][r] = sum_{c in chunk of 16 sites} x'[c] . S_r[c]  for every row r of the list BEFORE the merge,
// x' = the row k_pair_xp wrote; only alignments with need[b] != 0.  grid (ceil(C/16), B).
__global__ __launch_bounds__(256) void k_agg_dot(RowSet rs, const float* __restrict__ Xp, const int* __restrict__ need,
                                                 float* __restrict__ part, int n, int C, int rp) {
  const int chunk = blockIdx.x, b = blockIdx.y;
  if (need && !need[b]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ int slots[256];
  if (tid < n) slots[tid] = slot_of(rs, b, tid);
  __syncthreads();
  f32x4 xv[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int ee = (k * 64 + lane) * 4;
    ok[k] = chunk * 16 + (ee >> 6) < C;
    xv[k] = ok[k] ? *reinterpret_cast<const f32x4*>(Xp + ((size_t)b * C + chunk * 16) * 64 + ee) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const size_t bo = (size_t)b * rs.bstride;
  const size_t coff = (size_t)chunk * 16 * 64 + (size_t)lane * 4;
  const int nch = gridDim.x;
  for (int r = wave; r < n; r += 4) {
    const float* k0 = rs.S + bo + (size_t)slots[r] * C * 64 + coff;
    f32x4 kv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) kv[k] = ok[k] ? *reinterpret_cast<const f32x4*>(k0 + k * 256) : (f32x4){0.f, 0.f, 0.f, 0.f};
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += xv[k][0] * kv[k][0] + xv[k][1] * kv[k][1] + xv[k][2] * kv[k][2] + xv[k][3] * kv[k][3];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) part[((size_t)b * nch + chunk) * rp + r] = acc;
  }
}
// am[b][r'] (positions AFTER the merge of (i, j): position j removed, 0 at i) = softmax over the rows other than i, j of
// (sum of the chunk partials + beta_r) / sqrt(64 C).  One wave per alignment; n = rows before the merge (<= 65).
__global__ __launch_bounds__(64) void k_agg_am(RowSet rs, ScorerW w, const int* __restrict__ ij, const int* __restrict__ need,
                                               const float* __restrict__ part, int nch, int rp,
                                               const float* __restrict__ beta_slot, float* __restrict__ am, int n, int C) {
  const int b = blockIdx.x, lane = threadIdx.x;
  if (need && !need[b]) return;
  const int pi = min(max(ij[2 * b], 0), n - 1), pj = min(max(ij[2 * b + 1], 0), n - 1);
  const int nrow = (int)(rs.bstride / ((long)C * 64));
  float v[2];
  bool in[2];
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = 64 * k + lane;
    in[k] = n > 2 && r < n && r != pi && r != pj;
    v[k] = -INFINITY;
    if (in[k]) {
      float s = 0.f;
#pragma unroll 16
      for (int ch = 0; ch < nch; ++ch) s += part[((size_t)b * nch + ch) * rp + r];
      v[k] = (s + beta_slot[(size_t)b * nrow + slot_of(rs, b, r)]) * (1.0f / sqrtf(64.0f * (float)C));
    }
    mx = fmaxf(mx, v[k]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float se = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) { v[k] = in[k] ? expf(v[k] - mx) : 0.f; se += v[k]; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) se += __shfl_xor(se, o);
  am[(size_t)b * 64 + lane] = 0.f;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int r = 64 * k + lane;
    if (in[k]) am[(size_t)b * 64 + (r - (r > pj ? 1 : 0))] = se > 0.f ? v[k] / se : 0.f;
  }
}

// nnj_step (include/nnj.h) on the two-pass kernels.  In a rollout the table kernel of a step prepares the NEXT merge (its
// attention weights `am`, the candidate, the next live list) for the pair it has just picked; through the step API the
// CALLER names the merge of the next call.  One thread per alignment: the pair is brought into range like the table kernel
// does, compared with the pick the stored weights belong to -- another pair (or the first step of a session: `first`)
// sets `need`, so the fallback kernels compute the weights, and drops the candidate, whose positions assumed the pick --
// and the list after the merge is written beside the one before it (environment.py:764-768: position j leaves).
__global__ void k_step_prepare(int* __restrict__ ij, const int* __restrict__ pick, int first, int* __restrict__ need,
                               int* __restrict__ cand_cur, int* __restrict__ cand_run, const int* __restrict__ live_old,
                               int* __restrict__ live_new, int live_stride, int B, int n1) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int j = min(max(ij[2 * b + 1], 1), n1 - 1);
  const int i = min(max(ij[2 * b], 0), j - 1);
  ij[2 * b] = i; ij[2 * b + 1] = j;
  const bool same = !first && pick[2 * b] == i && pick[2 * b + 1] == j;
  if (!same) { need[b] = 1; cand_cur[2 * b] = -1; cand_cur[2 * b + 1] = -1; cand_run[b] = 0; }
  const int* lo = live_old + (size_t)b * live_stride;
  int* ln = live_new + (size_t)b * live_stride;
  for (int p = 0; p < n1 - 1; ++p) ln[p] = lo[p + (p >= j ? 1 : 0)];
}
