// nnj_step_g.hpp -- the alpha pass of the two-pass NJ step (nnj_step2.hpp: merge + attention logits of the new pairs)
// on SHARED 16-pair tiles and with the merged rows of a group of sites finished as ONE tile (round 4).
//
// k_step_alpha_w walks ceil(P / 16) tiles per site and finishes the merged row of every site on its own: x_g by vector
// products and column sums through LDS, then W_g and W_h as matrix products of which ONE column of sixteen is used, with
// five LDS round trips between one-feature-per-lane and tile layout on the way -- the knock-outs of round 3 put that
// chain at a fifth of the kernel.  Here a wave owns G consecutive sites at a time (as k_inc_score_g does):
//   1. the rows of all G sites go to the sites' images as fp16 pieces (row major, read both ways: nnj_scorer_g.hpp);
//   2. x_g of the G sites is ONE matrix product per site, x_g^T = S^T am^T, with the image read transposed and the
//      weights am as the B operand of the columns that stand for that site (column j of the chain's tile = site j mod G,
//      zeros elsewhere): the vector products, their LDS rows and the column sums are gone;
//   3. the chain -- x_ij gate, W_g, mix, W_h -- runs once per GROUP in tile layout (16 features per lane): the matrix
//      products serve G sites at once and nothing goes through LDS until S_m / U_m are handed to the pair columns;
//   4. the pairs of the G sites share tiles (columns q2 = g P + p); their products with the image rows use the image of
//      the column's site (a tile with columns of two sites accumulates one set per site; the epilogue picks).
#pragma once
#include "nnj_scorer_g.hpp"

// NT pair tiles per group of G sites (G a power of two), images of IR rows (>= P, multiple of 8), NW waves.
// Requires n >= 3 and G (n - 1) <= 16 NT with the groupings tile_glo / tile_ghi know.
template <int NT, int G, int IR, int NW, int PF = 2>
__global__ __launch_bounds__(64 * NW) void k_step_alpha_g(RowSet rs, ScorerW w, StepIO io, int n, int C, int cs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int KSX = (IR + 31) / 32;                      // k-steps of the x_g product (32 rows each)
  constexpr int NRT = (IR + 15) / 16;                      // row tiles (q') of the logits
  constexpr int PLB = IR * 128;                            // bytes of an image plane
  constexpr int SITEB = 2 * PLB;
  constexpr int IMGF = G * SITEB / 4;                      // floats of a wave's images
  constexpr int SLACK = (32 * KSX - IR) * 32 + 32 + 64;    // zeroed slack behind the last image + the dump line (nnj_scorer_g.hpp)
  constexpr int MAXS = tile_ghi<NT, G>(NT == 3 ? 1 : 0) - tile_glo<NT, G>(NT == 3 ? 1 : 0) + 1;   // sites a tile can hold
  constexpr int XS = 3 * 64 * G;                           // per-wave hand-over: S_m | U_m | x'_cand of the G sites
  static_assert((G & (G - 1)) == 0 && G <= 16, "G: a power of two");
  float* At_l = smem;                                      // A^T
  float* Wh_l = smem + IMG64;
  float* Wg_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* imgs = smem + 3 * IMG64;
  float* xs = imgs + NW * IMGF + SLACK + wave * XS;
  float* cv = imgs + NW * IMGF + SLACK + NW * XS;          // b_h | b_g | (s_out vectors, unused here) | u
  float* uv = cv + SCORER_CONSTS;
  unsigned short* amp = reinterpret_cast<unsigned short*>(uv + 64);   // am as fp16 pieces: [2 planes][64 q | 64 zeros]
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  const int P = n - 1;
  stage_image_t16(At_l, w.imgAt, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * NW);
  stage_scorer_consts(cv, w, tid);
  const int m = min(max(io.ij[2 * b], 0), n - 1);
  if (tid < 64) {
    uv[tid] = w.u[tid];
    const float a = tid < P ? io.am[(size_t)b * 64 + q_to_r(tid, m)] : 0.f;
    const _Float16 h = (_Float16)a;
    const _Float16 lo = (_Float16)(a - (float)h);
    amp[tid] = __builtin_bit_cast(unsigned short, h);
    amp[64 + tid] = 0;
    amp[128 + tid] = __builtin_bit_cast(unsigned short, lo);
    amp[192 + tid] = 0;
  }
  for (int i = tid; i < NW * IMGF + SLACK; i += 64 * NW) imgs[i] = 0.f;   // rows beyond the pairs: finite (they meet zeros)
  __syncthreads();
  const int j_old = min(max(io.ij[2 * b + 1], 0), n);
  const size_t bo = (size_t)b * rs.bstride;
  const int slot_m = slot_of(rs, b, m);
  const int slot_j = io.live_old[(size_t)b * rs.live_stride + j_old];
  const size_t om = bo + (size_t)slot_m * C * 64, oj = bo + (size_t)slot_j * C * 64;
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
  const unsigned img0 = lds_addr(imgs) + (unsigned)wave * (unsigned)(G * SITEB);
  char* imw = reinterpret_cast<char*>(imgs) + wave * (G * SITEB);
  // ---- per tile and lane: the column (site g of the group, pair p) -- as k_inc_score_g
  const float* Sr[NT];
  float sgn[NT], cacc[NT];
  int gq[NT], pq[NT];
  unsigned wo[NT], wm[NT];
  bool valid[NT];
  const unsigned dump = (unsigned)((NW - wave) * (G * SITEB)) + (unsigned)((32 * KSX - IR) * 128 + 128);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q2 = 16 * t + l15;
    int g = 0;
#pragma unroll
    for (int k = 1; k < G; ++k) g += q2 >= k * P ? 1 : 0;
    g += q2 >= G * P ? 1 : 0;
    valid[t] = g < G;
    if (!valid[t]) g = 0;
    const int p = valid[t] ? q2 - g * P : 0;
    const int r = q_to_r(p, m);
    gq[t] = g; pq[t] = p;
    sgn[t] = r < m ? 1.0f : -1.0f;
    cacc[t] = 0.f;
    Sr[t] = rs.S + bo + (size_t)slot_of(rs, b, r) * C * 64;
    wo[t] = valid[t] ? (unsigned)(g * SITEB + p * 128) + 8u * (unsigned)(kq ^ img_bp(p & 15)) : dump + 8u * (unsigned)kq;
    wm[t] = valid[t] ? (unsigned)PLB : 128u;
  }
  unsigned ro[2];                                          // transposed reads: rows 8 kq + 4 sec + (l15 >> 2), granule l15 & 3
#pragma unroll
  for (int sec = 0; sec < 2; ++sec) {
    const int row = 8 * kq + 4 * sec + (l15 >> 2);
    ro[sec] = (unsigned)(row * 128) + 8u * (unsigned)((l15 & 3) ^ img_bp(row & 15));
  }
  // row reads (A operand of the logits: rows q' = 16 mt + l15, k-step ks = granules 8 ks + 4 u + kq, u = 0, 1)
  const unsigned rr = (unsigned)(l15 * 128) + 8u * (unsigned)(kq ^ img_bp(l15));
  const int gs = l15 & (G - 1);                            // the chain's tile: column j stands for site j mod G
  f32x4 acc[NT][MAXS][NRT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int k = 0; k < MAXS; ++k)
#pragma unroll
      for (int mt = 0; mt < NRT; ++mt) acc[t][k][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float beta_acc = 0.f, cm_acc = 0.f;
  float* v_sm = xs;                                        // [G][64]
  float* v_um = xs + 64 * G;
  float* v_xc = xs + 128 * G;
  for (int cg = c0 + G * wave; cg < c1; cg += G * NW) {
    asm volatile("" ::: "memory");
    // ---- 1. the rows of the group's sites -> images (fp16 pieces)
    V64 srk[1];                                            // one tile per group (NT == 1): the rows stay in registers for step 4
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int cs_ = min(cg + gq[t], c1 - 1);
      V64 sr;
      load_v64(sr, Sr[t] + (size_t)cs_ * 64, kq);
      if constexpr (NT == 1) srk[0] = sr;
      Frag3 sf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned o = wo[t] ^ (32u * (2 * ks + u));
          *reinterpret_cast<u32x2*>(imw + o) = (u32x2){sf[ks].h[2 * u], sf[ks].h[2 * u + 1]};
          *reinterpret_cast<u32x2*>(imw + o + wm[t]) = (u32x2){sf[ks].m[2 * u], sf[ks].m[2 * u + 1]};
        }
    }
    asm volatile("" ::: "memory");
    // ---- 2. x_g of the G sites: column j of the tile = site j mod G
    V64 xg;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xg.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    static_for<0, G>([&](auto gic) {
      constexpr int gi = decltype(gic)::value;
      Frag3 bfr[KSX];
      const unsigned short* ap_ = amp + (gs == gi ? 0 : 64) + 8 * kq;
#pragma unroll
      for (int ks = 0; ks < KSX; ++ks) {
        bfr[ks].h = *reinterpret_cast<const u32x4*>(ap_ + 32 * ks);
        bfr[ks].m = *reinterpret_cast<const u32x4*>(ap_ + 32 * ks + 128);
      }
      xg_from_image<KSX, PLB, PF>(xg.t, img0 + (unsigned)(gi * SITEB), ro, bfr);
    });
    // ---- 3. the merged rows of the group in tile layout (restates k_step_alpha's step 2)
    {
      const bool sv = cg + gs < c1;
      const size_t co = (size_t)min(cg + gs, c1 - 1) * 64;
      V64 smd;                                             // (x_ij first: built in two stages so that only two of the
      {                                                    //  four rows of the merged pair are in registers at a time)
        V64 ui, uj;
        load_v64(ui, rs.U + om + co, kq); load_v64(uj, rs.U + oj + co, kq);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(cv + 16 * mt + 4 * kq);
#pragma unroll
          for (int e = 0; e < 4; ++e) smd.t[mt][e] = sigmoid_l2(ui.t[mt][e] - uj.t[mt][e] + b4[e]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        V64 si, sj;
        load_v64(si, rs.S + om + co, kq); load_v64(sj, rs.S + oj + co, kq);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) smd.t[mt][e] = sj.t[mt][e] + smd.t[mt][e] * (si.t[mt][e] - sj.t[mt][e]);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        V64 gg;
        linear_t16p<4, false, true>(gg.t, xg, Wg_l, cv + 64, lane);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gate_mix4(smd.t[mt], xg.t[mt], gg.t[mt]);          // (1-w)*x + w*x_g
      }
      __builtin_amdgcn_sched_barrier(0);
      const bool writer = l15 < G && sv;                   // one replica of every site's column writes
      if (writer) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(io.S_w + om + co + 16 * mt + 4 * kq) = smd.t[mt];
      }
      if (l15 < G) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(v_sm + gs * 64 + 16 * mt + 4 * kq) = smd.t[mt];
      }
      {
        float bs = 0.f, cms = 0.f;
        V64 xc;
        if (has_cand) load_v64(xc, Xc + co, kq);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 u4 = *reinterpret_cast<const f32x4*>(uv + 16 * mt + 4 * kq);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            bs += u4[e] * smd.t[mt][e];
            if (has_cand) cms += xc.t[mt][e] * smd.t[mt][e];
          }
        }
        if (writer) { beta_acc += bs; cm_acc += cms; }
        if (has_cand && l15 < G) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(v_xc + gs * 64 + 16 * mt + 4 * kq) = xc.t[mt];
        }
      }
      V64 um;
      linear_t16p<4, false, false>(um.t, smd, Wh_l, nullptr, lane);
      if (writer) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(io.U_w + om + co + 16 * mt + 4 * kq) = um.t[mt];
      }
      if (l15 < G) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(v_um + gs * 64 + 16 * mt + 4 * kq) = um.t[mt];
      }
    }
    asm volatile("" ::: "memory");
    // ---- 4. the new pairs (m, r): gate, x' = A^T x, logits against the rows of the column's site
    static_for<0, NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      V64 sr;
      if constexpr (NT == 1) sr = srk[0];
      else load_v64(sr, Sr[t] + (size_t)min(cg + gq[t], c1 - 1) * 64, kq);
      V64 x;
      {
        Frag3 sf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
        V64 ur;
        {
          V64 um;
          load_v64(um, v_um + gq[t] * 64, kq);
          gate_init16(ur, um, cv, sgn[t], kq);
        }
        lds_wait_all();
        linear_t16p_core<4>(ur.t, sf, Wh_l, lane, [] {});
        V64 sm;
        load_v64(sm, v_sm + gq[t] * 64, kq);
        gate16(x, sr, ur, sm);
      }
      if (has_cand) {
        V64 xcv;
        load_v64(xcv, v_xc + gq[t] * 64, kq);
        float d = 0.f;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int e = 0; e < 4; ++e) d += xcv.t[mt][e] * sr.t[mt][e];
        cacc[t] += (valid[t] && cg + gq[t] < c1) ? d : 0.f;
      }
      V64 xp;
      lds_wait_all();
      linear_t16p<4, false, false>(xp.t, x, At_l, nullptr, lane);         // x' = A^T x
      if (!(valid[t] && cg + gq[t] < c1)) {                               // padding columns and sites beyond the chunk add nothing
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xp.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      Frag3 bx[2];
      split_8(bx[0], xp.t[0], xp.t[1]);
      split_8(bx[1], xp.t[2], xp.t[3]);
      static_for<tile_glo<NT, G>(t), tile_ghi<NT, G>(t) + 1>([&](auto gic) {
        constexpr int gi = decltype(gic)::value;
        constexpr int k = gi - tile_glo<NT, G>(t);
        const char* ib = imw + gi * SITEB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int mt = 0; mt < NRT; ++mt) {
            const u32x2 h0 = *reinterpret_cast<const u32x2*>(ib + (rr ^ (unsigned)(64 * ks)) + 2048 * mt);
            const u32x2 h1 = *reinterpret_cast<const u32x2*>(ib + (rr ^ (unsigned)(64 * ks + 32)) + 2048 * mt);
            const u32x2 m0 = *reinterpret_cast<const u32x2*>(ib + (rr ^ (unsigned)(64 * ks)) + 2048 * mt + PLB);
            const u32x2 m1 = *reinterpret_cast<const u32x2*>(ib + (rr ^ (unsigned)(64 * ks + 32)) + 2048 * mt + PLB);
            Frag3 a;
            a.h = (u32x4){h0[0], h0[1], h1[0], h1[1]};
            a.m = (u32x4){m0[0], m0[1], m1[0], m1[1]};
            acc[t][k][mt] = mfma16_b6(a, bx[ks], acc[t][k][mt]);
          }
      });
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // ---- epilogue: one partial set per WORKGROUP.  Per wave the columns of a pair are added site by site (a column takes
  // the accumulator set of ITS site), then the waves in wave order: bitwise reproducible
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // [64 q][64 q'] (the images are dead)
  float* epi = red + 4096;                                 // [NW][66]
  for (int i = tid; i < 4096 + NW * 66; i += 64 * NW) red[i] = 0.f;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { beta_acc += __shfl_xor(beta_acc, o); cm_acc += __shfl_xor(cm_acc, o); }
#pragma unroll
  for (int t = 0; t < NT; ++t) { cacc[t] += __shfl_xor(cacc[t], 16); cacc[t] += __shfl_xor(cacc[t], 32); }
  __syncthreads();
  if (lane == 0) { epi[wave * 66 + 64] = beta_acc; epi[wave * 66 + 65] = cm_acc; }
  for (int gi = 0; gi < G; ++gi) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (kq == 0 && valid[t] && gq[t] == gi) epi[wave * 66 + pq[t]] += cacc[t];
    asm volatile("" ::: "memory");
  }
  for (int s_ = 0; s_ < NW; ++s_) {
    if (wave == s_) {
      static_for<0, G>([&](auto gic) {
        constexpr int gi = decltype(gic)::value;
        static_for<0, NT>([&](auto tc) {
          constexpr int t = decltype(tc)::value;
          if constexpr (gi >= tile_glo<NT, G>(t) && gi <= tile_ghi<NT, G>(t)) {
            constexpr int k = gi - tile_glo<NT, G>(t);
            if (valid[t] && gq[t] == gi) {
#pragma unroll
              for (int mt = 0; mt < NRT; ++mt) {
                f32x4 v = acc[t][k][mt];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (16 * mt + 4 * kq + e < P) ? v[e] : 0.f;   // rows beyond the pairs: zeros
                f32x4* d4 = reinterpret_cast<f32x4*>(red + pq[t] * 64 + 16 * mt + 4 * kq);
                *d4 = *d4 + v;
              }
            }
          }
        });
        asm volatile("" ::: "memory");
      });
    }
    __syncthreads();
  }
  float* dst = io.alpha_part + ((size_t)b * gridDim.x + sc) * 4096;
  for (int i = tid; i < 1024; i += 64 * NW)
    reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(red)[i];
  if (tid < 64) {
    float v = 0.f;
    const int idx = tid == 63 ? 65 : (tid < 16 * NT ? tid : -1);           // (q <= 62: entry 63 carries the merged row's)
    if (idx >= 0)
      for (int s_ = 0; s_ < NW; ++s_) v += epi[s_ * 66 + idx];
    if (io.acand_part) io.acand_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  } else if (tid == 64) {
    float v = 0.f;
    for (int s_ = 0; s_ < NW; ++s_) v += epi[s_ * 66 + 64];
    float* bp = io.beta_w + ((size_t)b * (rs.bstride / ((long)C * 64)) + slot_m) * rs.ntile32;
    bp[sc] = v;
    if (sc == 0)
      for (int k = gridDim.x; k < io.beta_n; ++k) bp[k] = 0.f;            // entries of a row that had more partials
  }
}

// floats of dynamic LDS k_step_alpha_g<NT, G, IR, NW> needs (the epilogue's 4096 + 66 NW floats alias the images)
constexpr int step_alpha_g_lds(int NT, int G, int IR, int NW) {
  const int body = NW * G * IR * 64 + ((32 * ((IR + 31) / 32) - IR) * 32 + 32 + 64) + NW * 3 * 64 * G + SCORER_CONSTS + 64 + 128;
  const int epi = 4096 + NW * 66;
  return 3 * IMG64 + (body > epi ? body : epi);
}
