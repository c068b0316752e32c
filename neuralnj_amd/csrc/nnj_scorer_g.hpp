// nnj_scorer_g.hpp -- scores of the new pairs of an NJ step with SHARED 16-pair tiles (round 4).
//
// k_inc_score16 / k_inc_score_w give every site its own ceil(P / 16) tiles (P = n - 1 new pairs): over the 48 steps of a
// 50-taxon rollout 23 % of the pair columns are padding, and the time of these kernels follows the number of tiles, not
// the number of pairs (0.54 ms per step for every P <= 16, 1.10 for 17..32, 1.82 for 33..48 at a batch of 256).  Here a
// wave owns G consecutive sites at a time and lays their G * P pairs out as ONE column list q2 = g * P + p over
// NT = ceil(G P / 16) tiles: P <= 4 -> four sites per tile, P <= 8 -> two, 17..24 pairs -> three tiles for two sites
// instead of four.  Everything of the chain is column local (gate, W_g, mix, s_out, GELU: the weights are shared) except
// the context x_g^T = S^T alpha^T, whose A operand is the image of the COLUMN's site: a tile that holds columns of
// several sites runs that product once per site with the alpha fragments of the other sites' columns replaced by zeros
// (more MFMAs on a pipe that idles, no vector work).
//
// The site image is stored ROW MAJOR -- [2 planes][IR rows r'][64 d] fp16, a lane writes the pieces of its own row as
// four 8-byte granules per plane -- and read TRANSPOSED by ds_read_b64_tr_b16 (cdna_hip_programming.md T10): the
// 16 scalar ds_write_b16 per plane, tile and lane of the older kernels (and their address arithmetic) are gone.
// Granule (row r, unit v = feature / 4) sits at unit v ^ bp(r & 15), bp = the row's low four bits with bits 1 and 2
// exchanged: the 16 rows of a store group land on 16 different granule positions (conflict free on the 32-bank write
// path) and the eight rows {r0 .. r0+3, r0+8 .. r0+11} a 32-lane half of a transposed read covers land on all 64 banks.
#pragma once
#include "nnj_step2.hpp"

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int OFF>
__device__ __forceinline__ void lds_read_tr16(u32x2& d, unsigned byte_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(byte_addr), "n"(OFF) : "memory");
#endif
}
__device__ __forceinline__ int img_bp(int r) { return (r & 9) | ((r & 2) << 1) | ((r & 4) >> 1); }

// xg[mt] += S^T alpha^T for one site image at LDS byte address `base`.  `ro` = the lane's two read offsets (sec = 0, 1:
// rows 8 kq + 4 sec + (l15 >> 2), granule l15 & 3, swizzled; the row tile mt is an XOR of bits 5..6), `bfr` = the alpha
// fragments of the lane's column (zeros where the column belongs to another site).  Reads PF row tiles ahead.
template <int N>
__device__ __forceinline__ void lds_wait_le_nb() {   // counted LDS wait without a scheduling barrier: the consumers below are
#if defined(__HIP_DEVICE_COMPILE__)                  // pinned behind it one by one (pin_frag), everything else may move
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
#endif
}
template <int KSX, int PLB, int PF>
__device__ __forceinline__ void xg_from_image(f32x4 (&xg)[4], unsigned base, const unsigned (&ro)[2],
                                              const Frag3 (&bfr)[KSX]) {
  constexpr int NS = 4 * KSX;                              // step s: ks = s / 4, mt = s % 4
  u32x2 ah[PF][2], am[PF][2];                              // [buffer][sec]
  auto issue = [&](auto si) {
    constexpr int s = decltype(si)::value;
    if constexpr (s < NS) {
      constexpr int ks = s / 4, mt = s % 4;
      const unsigned a0 = base + (ro[0] ^ (32u * mt)), a1 = base + (ro[1] ^ (32u * mt));
      lds_read_tr16<ks * 4096>(ah[s % PF][0], a0);
      lds_read_tr16<ks * 4096>(ah[s % PF][1], a1);
      lds_read_tr16<ks * 4096 + PLB>(am[s % PF][0], a0);
      lds_read_tr16<ks * 4096 + PLB>(am[s % PF][1], a1);
    }
  };
  static_for<0, PF>([&](auto si) { issue(si); });
  static_for<0, NS>([&](auto si) {
    constexpr int s = decltype(si)::value;
    constexpr int ks = s / 4, mt = s % 4;
    constexpr int ahead = (s + PF - 1 < NS - 1 ? s + PF - 1 : NS - 1) - s;      // steps whose reads are younger than step s
    lds_wait_le_nb<4 * ahead>();
    Frag3 a;
    a.h = (u32x4){ah[s % PF][0][0], ah[s % PF][0][1], ah[s % PF][1][0], ah[s % PF][1][1]};
    a.m = (u32x4){am[s % PF][0][0], am[s % PF][0][1], am[s % PF][1][0], am[s % PF][1][1]};
    pin_frag(a);
    xg[mt] = mfma16_b6(a, bfr[ks], xg[mt]);
    issue(std::integral_constant<int, s + PF>{});
  });
}

// which sites of a group can own columns of tile t (compile time: the launch guarantees the pair range of a tier):
// one site per group: itself; one tile per group: all G; three tiles for two sites (17..24 pairs): {0}, {0,1}, {1}
template <int NT, int G>
__host__ __device__ constexpr int tile_glo(int t) { return (G == 1 || NT == 1) ? 0 : (t == 2 ? 1 : 0); }
template <int NT, int G>
__host__ __device__ constexpr int tile_ghi(int t) { return G == 1 ? 0 : (NT == 1 ? G - 1 : (t == 0 ? 0 : 1)); }

// NT tiles per group of G sites, images of IR rows (IR >= P, a multiple of 8), NW waves per workgroup.
// part[b][sc][pair p].  Requires n >= 3 (a context exists) and G (n - 1) <= 16 NT.
template <int NT, int G, int IR, int NW, int PF = 2>
__global__ __launch_bounds__(64 * NW) void k_inc_score_g(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                        const float* __restrict__ alpha,
                                                        const uint8_t* __restrict__ mask,
                                                        float* __restrict__ score_part, int n, int C, int cs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WPF = 2;
  constexpr int KSX = (IR + 31) / 32;                      // k-steps of the x_g product (32 r' each)
  constexpr int PLB = IR * 128;                            // bytes of an image plane
  constexpr int SITEB = 2 * PLB;                           // bytes of a site image
  constexpr int IMGF = G * SITEB / 4;                      // floats of a wave's images
  constexpr int SLACK = (32 * KSX - IR) * 32 + 32 + 64;    // floats behind the last image: zeroed slack (rows IR .. 32 KSX - 1 of it) + 256 B where padding columns write
  static_assert(G == 1 || NT == 1 || (NT == 3 && G == 2), "tile_glo / tile_ghi know these groupings");
  constexpr int APR = 16 * NT;                             // alpha rows in LDS (+ a zero row at APR)
  constexpr int APN = APR + 1;                             // rows of an alpha plane in LDS
  constexpr int APL = APN * 64;                            // fp16 elements of an alpha plane
  float* Wg_l = smem;
  float* S0_l = smem + IMG64;
  float* Wh_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* imgs = smem + 3 * IMG64;
  float* cv = imgs + NW * IMGF + SLACK;
  unsigned short* alds = reinterpret_cast<unsigned short*>(cv + SCORER_CONSTS);
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  const int P = n - 1;
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * NW);
  stage_image_t16(S0_l, w.imgS0, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  stage_scorer_consts(cv, w, tid);
  // image rows beyond the pairs are never written: they meet alpha = 0 and must be finite
  for (int i = tid; i < NW * IMGF + SLACK; i += 64 * NW) imgs[i] = 0.f;
  {
    const long apl = (long)gridDim.y * 4096;
    const unsigned short* ag = reinterpret_cast<const unsigned short*>(alpha) + (size_t)b * 4096;
    for (int i = tid; i < 2 * APN * 8; i += 64 * NW) {                     // 16-byte chunks of both planes
      const int pl = i / (APN * 8), rc = i % (APN * 8), r = rc >> 3, ch = rc & 7;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (r < APR) v = *reinterpret_cast<const u32x4*>(ag + pl * apl + r * 64 + 8 * ch);
      *reinterpret_cast<u32x4*>(alds + pl * APL + r * 64 + 8 * (ch ^ (r & 7))) = v;
    }
  }
  __syncthreads();
  const size_t bo = (size_t)b * rs.bstride;
  const int m = min(max(ij_prev[2 * b], 0), n - 1);
  const float* Sm = rs.S + bo + (size_t)slot_of(rs, b, m) * C * 64;
  const float* Um = rs.U + bo + (size_t)slot_of(rs, b, m) * C * 64;
  const unsigned img0 = lds_addr(imgs) + (unsigned)wave * (unsigned)(G * SITEB);
  char* imw = reinterpret_cast<char*>(imgs) + wave * (G * SITEB);
  // ---- per tile and lane: the column (site g of the group, pair p)
  const float* Sr[NT];
  float sgn[NT], score[NT];
  int gq[NT], pq[NT];
  unsigned wo[NT], wm[NT];                                 // image write offset of the lane's row; distance of its m plane
  bool valid[NT];
  const unsigned dump = (unsigned)((NW - wave) * (G * SITEB)) + (unsigned)((32 * KSX - IR) * 128 + 128);   // (from imw)
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int q2 = 16 * t + l15;
    int g = 0;
#pragma unroll
    for (int k = 1; k < G; ++k) g += q2 >= k * P ? 1 : 0;
    g += q2 >= G * P ? 1 : 0;                              // g == G: a padding column
    valid[t] = g < G;
    if (!valid[t]) g = 0;
    const int p = valid[t] ? q2 - g * P : 0;
    const int r = q_to_r(p, m);
    gq[t] = g; pq[t] = p;
    sgn[t] = r < m ? 1.0f : -1.0f;
    score[t] = 0.f;
    Sr[t] = rs.S + bo + (size_t)slot_of(rs, b, r) * C * 64;
    // a padding column writes its (finite, never read) pieces to the dump line instead of branching around the stores
    wo[t] = valid[t] ? (unsigned)(g * SITEB + p * 128) + 8u * (unsigned)(kq ^ img_bp(p & 15)) : dump + 8u * (unsigned)kq;
    wm[t] = valid[t] ? (unsigned)PLB : 128u;
  }
  // read bases of the transposed fragments (site 0 of the group): row 8 kq + 4 sec + (l15 >> 2), unit l15 & 3
  unsigned ro[2];
#pragma unroll
  for (int sec = 0; sec < 2; ++sec) {
    const int row = 8 * kq + 4 * sec + (l15 >> 2);
    ro[sec] = (unsigned)(row * 128) + 8u * (unsigned)((l15 & 3) ^ img_bp(row & 15));
  }
  for (int cg = c0 + G * wave; cg < c1; cg += G * NW) {
    asm volatile("" ::: "memory");
    V64 x[NT];
    float mc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int cs_ = min(cg + gq[t], c1 - 1);                             // (sites beyond the chunk: clamped, weight 0)
      mc[t] = (!valid[t] || cg + gq[t] >= c1 || mask[(size_t)b * C + cs_]) ? 0.f : 1.f;   // seq_mask (model.py:96)
      V64 sr, sm, um;
      load_v64(sr, Sr[t] + (size_t)cs_ * 64, kq);
      load_v64(sm, Sm + (size_t)cs_ * 64, kq);
      load_v64(um, Um + (size_t)cs_ * 64, kq);
      Frag3 sf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) split_8<false>(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
      V64 ur;
      gate_init16(ur, um, cv, sgn[t], kq);
      linear_t16p_core<4, WPF>(ur.t, sf, Wh_l, lane, [] {});
      gate16(x[t], sr, ur, sm);
      // the lane's row of its site's image: features 32 ks + 16 u + 4 kq .. + 3 = granule 4 (2 ks + u) + kq
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned o = wo[t] ^ (32u * (2 * ks + u));
          *reinterpret_cast<u32x2*>(imw + o) = (u32x2){sf[ks].h[2 * u], sf[ks].h[2 * u + 1]};
          *reinterpret_cast<u32x2*>(imw + o + wm[t]) = (u32x2){sf[ks].m[2 * u], sf[ks].m[2 * u + 1]};
        }
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- per tile: x_g^T = S^T alpha^T (once per site the tile's columns belong to), W_g, mix, s_out
    static_for<0, NT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      V64 xg, gg;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xg.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      static_for<tile_glo<NT, G>(t), tile_ghi<NT, G>(t) + 1>([&](auto gic) {
        constexpr int gi = decltype(gic)::value;
        Frag3 bfr[KSX];
        const int ar = (valid[t] && gq[t] == gi) ? pq[t] : APR;           // other sites' columns: the zero row
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
          const unsigned short* ap_ = alds + ar * 64 + 8 * ((4 * ks + kq) ^ (ar & 7));
          bfr[ks].h = *reinterpret_cast<const u32x4*>(ap_);
          bfr[ks].m = *reinterpret_cast<const u32x4*>(ap_ + APL);
        }
        xg_from_image<KSX, PLB, PF>(xg.t, img0 + (unsigned)(gi * SITEB), ro, bfr);
      });
      linear_t16p<4, false, true, WPF, false>(gg.t, xg, Wg_l, cv + 64, lane);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) gate_mix4(x[t].t[mt], xg.t[mt], gg.t[mt]);          // (1-w)*x + w*x_g
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
      score[t] += (s + w.s2b) * mc[t];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  // one partial set per WORKGROUP: per wave the columns of a pair are added site by site, then the waves in wave order
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // [NW][64] (the images are dead)
  red[wave * 64 + lane] = 0.f;
  __syncthreads();
  for (int gi = 0; gi < G; ++gi) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (kq == 0 && valid[t] && gq[t] == gi) red[wave * 64 + pq[t]] += score[t];
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < NW; ++s_) v += red[s_ * 64 + tid];
    score_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  }
}

// ------------------------------------------------------------------ k_inc_score_s5: 33..40 pairs, FIVE tiles per TWO sites
// k_inc_score_w<3> gives each site three tiles, the third at most half full.  Here a wave owns two consecutive sites A, B
// and five tiles: A0 A1 (pairs 0..31 of A), S (columns 0..7 = pairs 32..39 of A, columns 8..15 = pairs 32..39 of B),
// B0 B1.  Two 40-row images at once would take 160 KB for eight waves; instead the images of A and B share ONE 48-row
// buffer IN TURN: rows 0..31 hold the site being worked on, rows 32..39 the tail of A, rows 40..47 the tail of B (both
// written with tile S).  Order per site pair: gate phase of A0 A1 S -> chains of A0 A1 and the A half of x_g(S) ->
// gate phase of B0 B1 (rows 0..31 overwritten) -> the B half of x_g(S), chain of S -> chains of B0 B1.  For site B the
// k-slots 8..15 of k-step 1 (buffer rows 40..47) carry alpha[32..39]: its alpha fragments of that k-step are read one
// chunk down and the lanes of slots 0..7 (A's tail) take the zero row.  The tile pairs (A0 A1), (B0 B1) run stage by
// stage (linear_t16p_multi, see k_inc_score_wi), the image is the transposed-read one of k_inc_score_g.
// Requires 33 <= n - 1 <= 40.  part[b][sc][pair p].
template <int N, int KSX, int PLB, int PF>
__device__ __forceinline__ void xg_from_image_multi(V64 (&xg)[N], unsigned base, const unsigned (&ro)[2],
                                                    const Frag3 (&bfr)[N][KSX]) {
  constexpr int NS = 4 * KSX;
  u32x2 ah[PF][2], am[PF][2];
  auto issue = [&](auto si) {
    constexpr int s = decltype(si)::value;
    if constexpr (s < NS) {
      constexpr int ks = s / 4, mt = s % 4;
      const unsigned a0 = base + (ro[0] ^ (32u * mt)), a1 = base + (ro[1] ^ (32u * mt));
      lds_read_tr16<ks * 4096>(ah[s % PF][0], a0);
      lds_read_tr16<ks * 4096>(ah[s % PF][1], a1);
      lds_read_tr16<ks * 4096 + PLB>(am[s % PF][0], a0);
      lds_read_tr16<ks * 4096 + PLB>(am[s % PF][1], a1);
    }
  };
  static_for<0, PF>([&](auto si) { issue(si); });
  static_for<0, NS>([&](auto si) {
    constexpr int s = decltype(si)::value;
    constexpr int ks = s / 4, mt = s % 4;
    constexpr int ahead = (s + PF - 1 < NS - 1 ? s + PF - 1 : NS - 1) - s;
    lds_wait_le_nb<4 * ahead>();
    Frag3 a;
    a.h = (u32x4){ah[s % PF][0][0], ah[s % PF][0][1], ah[s % PF][1][0], ah[s % PF][1][1]};
    a.m = (u32x4){am[s % PF][0][0], am[s % PF][0][1], am[s % PF][1][0], am[s % PF][1][1]};
    pin_frag(a);
#pragma unroll
    for (int u = 0; u < N; ++u) xg[u].t[mt] = mfma16_f16(a.m, bfr[u][ks].h, xg[u].t[mt]);
#pragma unroll
    for (int u = 0; u < N; ++u) xg[u].t[mt] = mfma16_f16(a.h, bfr[u][ks].m, xg[u].t[mt]);
#pragma unroll
    for (int u = 0; u < N; ++u) xg[u].t[mt] = mfma16_f16(a.h, bfr[u][ks].h, xg[u].t[mt]);
    issue(std::integral_constant<int, s + PF>{});
  });
}

template <int NW, int PF = 2>
__global__ __launch_bounds__(64 * NW) void k_inc_score_s5(RowSet rs, ScorerW w, const int* __restrict__ ij_prev,
                                                         const float* __restrict__ alpha,
                                                         const uint8_t* __restrict__ mask,
                                                         float* __restrict__ score_part, int n, int C, int cs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int WPF = 2;
  constexpr int IR = 48, KSX = 2;
  constexpr int PLB = IR * 128;                            // bytes of an image plane
  constexpr int SITEB = 2 * PLB;                           // bytes of a wave's buffer
  constexpr int IMGF = SITEB / 4;
  constexpr int SLACK = (32 * KSX - IR) * 32 + 32 + 64;    // as k_inc_score_g: zeroed rows behind the last buffer + the dump line
  constexpr int APR = 48, APN = APR + 1, APL = APN * 64;   // alpha rows in LDS (+ a zero row at APR)
  float* Wg_l = smem;
  float* S0_l = smem + IMG64;
  float* Wh_l = smem + 2 * IMG64;
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* imgs = smem + 3 * IMG64;
  float* cv = imgs + NW * IMGF + SLACK;
  unsigned short* alds = reinterpret_cast<unsigned short*>(cv + SCORER_CONSTS);
  const int sc = blockIdx.x, b = blockIdx.y;
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  const int P = n - 1;
  stage_image_t16(Wg_l, w.imgWg, tid, 64 * NW);
  stage_image_t16(S0_l, w.imgS0, tid, 64 * NW);
  stage_image_t16(Wh_l, w.imgWh, tid, 64 * NW);
  stage_scorer_consts(cv, w, tid);
  for (int i = tid; i < NW * IMGF + SLACK; i += 64 * NW) imgs[i] = 0.f;      // tail rows no pair writes must be finite
  {
    const long apl = (long)gridDim.y * 4096;
    const unsigned short* ag = reinterpret_cast<const unsigned short*>(alpha) + (size_t)b * 4096;
    for (int i = tid; i < 2 * APN * 8; i += 64 * NW) {
      const int pl = i / (APN * 8), rc = i % (APN * 8), r = rc >> 3, ch = rc & 7;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (r < APR) v = *reinterpret_cast<const u32x4*>(ag + pl * apl + r * 64 + 8 * ch);
      *reinterpret_cast<u32x4*>(alds + pl * APL + r * 64 + 8 * (ch ^ (r & 7))) = v;
    }
  }
  __syncthreads();
  const size_t bo = (size_t)b * rs.bstride;
  const int m = min(max(ij_prev[2 * b], 0), n - 1);
  const float* Sm = rs.S + bo + (size_t)slot_of(rs, b, m) * C * 64;
  const float* Um = rs.U + bo + (size_t)slot_of(rs, b, m) * C * 64;
  const unsigned img0 = lds_addr(imgs) + (unsigned)wave * (unsigned)SITEB;
  char* imw = reinterpret_cast<char*>(imgs) + wave * SITEB;
  // full tiles t = 0, 1: pair 16 t + l15 (of site A or B); tile S: pair 32 + (l15 & 7) of site l15 >> 3
  const float* Sr[3];
  float sgn[3];
  unsigned wo[3];
  const int siteS = l15 >> 3, pS = 32 + (l15 & 7);
  const bool validS = pS < P;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    const int p = t < 2 ? 16 * t + l15 : (validS ? pS : 0);
    const int r = q_to_r(p, m);
    sgn[t] = r < m ? 1.0f : -1.0f;
    Sr[t] = rs.S + bo + (size_t)slot_of(rs, b, r) * C * 64;
    const int R = t < 2 ? p : pS + 8 * siteS;                               // buffer row
    wo[t] = (unsigned)(R * 128) + 8u * (unsigned)(kq ^ img_bp(R & 15));
  }
  const unsigned dump = (unsigned)((NW - wave) * SITEB) + (unsigned)((32 * KSX - IR) * 128 + 128);   // (from imw)
  const unsigned woS = validS ? wo[2] : dump + 8u * (unsigned)kq;
  const unsigned wmS = validS ? (unsigned)PLB : 128u;
  unsigned ro[2];
#pragma unroll
  for (int sec = 0; sec < 2; ++sec) {
    const int row = 8 * kq + 4 * sec + (l15 >> 2);
    ro[sec] = (unsigned)(row * 128) + 8u * (unsigned)((l15 & 3) ^ img_bp(row & 15));
  }
  float score[3] = {0.f, 0.f, 0.f};
  // gate phase of the two full tiles of site c: x, and rows 0..31 of the buffer
  auto gate_pair = [&](V64 (&x)[2], int c) {
    V64 sr[2], sm, um, ur[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) load_v64(sr[t], Sr[t] + (size_t)c * 64, kq);
    load_v64(sm, Sm + (size_t)c * 64, kq);
    load_v64(um, Um + (size_t)c * 64, kq);
    Frag3 sf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) gate_init16(ur[t], um, cv, sgn[t], kq);
    linear_t16p_multi<2, WPF>(ur, sf, Wh_l, lane, [&](auto ki) {
      constexpr int ks = decltype(ki)::value;
#pragma unroll
      for (int t = 0; t < 2; ++t) split_8<false>(sf[t][ks], sr[t].t[2 * ks], sr[t].t[2 * ks + 1]);
    });
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const unsigned o = wo[t] ^ (32u * (2 * ks + u));
          *reinterpret_cast<u32x2*>(imw + o) = (u32x2){sf[t][ks].h[2 * u], sf[t][ks].h[2 * u + 1]};
          *reinterpret_cast<u32x2*>(imw + o + PLB) = (u32x2){sf[t][ks].m[2 * u], sf[t][ks].m[2 * u + 1]};
        }
#pragma unroll
    for (int t = 0; t < 2; ++t) gate16(x[t], sr[t], ur[t], sm);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // chains of the two full tiles against the buffer as it stands; SHIFT: the site is B (tail rows at 40..47)
  auto chain_pair = [&](V64 (&x)[2], float mcs, auto shift_c) {
    constexpr bool SHIFT = decltype(shift_c)::value;
    V64 xg[2];
    {
      Frag3 bfr[2][KSX];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) xg[t].t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSX; ++ks) {
          int ar = 16 * t + l15, ch = 4 * ks + kq;
          if (SHIFT && ks == 1) { if (kq == 0) ar = APR; else ch -= 1; }
          const unsigned short* ap_ = alds + ar * 64 + 8 * (ch ^ (ar & 7));
          bfr[t][ks].h = *reinterpret_cast<const u32x4*>(ap_);
          bfr[t][ks].m = *reinterpret_cast<const u32x4*>(ap_ + APL);
        }
      }
      xg_from_image_multi<2, KSX, PLB, PF>(xg, img0, ro, bfr);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    {
      V64 g[2];
      Frag3 bx[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) g[t].t[mt] = *reinterpret_cast<const f32x4*>(cv + 64 + 16 * mt + 4 * kq);
      linear_t16p_multi<2, WPF>(g, bx, Wg_l, lane, [&](auto ki) {
        constexpr int ks = decltype(ki)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) split_8<false>(bx[t][ks], xg[t].t[2 * ks], xg[t].t[2 * ks + 1]);
      });
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gate_mix4(x[t].t[mt], xg[t].t[mt], g[t].t[mt]);          // (1-w)*x + w*x_g
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    {
      V64 s1[2];
      Frag3 bx[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) s1[t].t[mt] = *reinterpret_cast<const f32x4*>(cv + 128 + 16 * mt + 4 * kq);
      linear_t16p_multi<2, WPF>(s1, bx, S0_l, lane, [&](auto ki) {
        constexpr int ks = decltype(ki)::value;
#pragma unroll
        for (int t = 0; t < 2; ++t) split_8<false>(bx[t][ks], x[t].t[2 * ks], x[t].t[2 * ks + 1]);
      });
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x2v s2 = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 16 * mt + 4 * kq);
          gelu_dot4(s2, s1[t].t[mt], w4);
        }
        float s = s2[0] + s2[1];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        score[t] += (s + w.s2b) * mcs;
      }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // half of x_g(S): the columns of site `sb` against the buffer as it stands
  auto xg_half = [&](V64& xg, auto site_c) {
    constexpr int sb = decltype(site_c)::value;
    Frag3 bfr[KSX];
#pragma unroll
    for (int ks = 0; ks < KSX; ++ks) {
      int ar = (validS && siteS == sb) ? pS : APR, ch = 4 * ks + kq;
      if (sb == 1 && ks == 1) { if (kq == 0) ar = APR; else ch -= 1; }
      const unsigned short* ap_ = alds + ar * 64 + 8 * (ch ^ (ar & 7));
      bfr[ks].h = *reinterpret_cast<const u32x4*>(ap_);
      bfr[ks].m = *reinterpret_cast<const u32x4*>(ap_ + APL);
    }
    xg_from_image<KSX, PLB, PF>(xg.t, img0, ro, bfr);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  for (int cg = c0 + 2 * wave; cg < c1; cg += 2 * NW) {
    asm volatile("" ::: "memory");
    const int cA = cg, cB = min(cg + 1, c1 - 1);
    const bool hasB = cg + 1 < c1;
    const float mcA = mask[(size_t)b * C + cA] ? 0.f : 1.f;                  // seq_mask (model.py:96)
    const float mcB = (!hasB || mask[(size_t)b * C + cB]) ? 0.f : 1.f;
    const float mcS = !validS ? 0.f : (siteS == 0 ? mcA : mcB);
    V64 xgS;
    const int cS = siteS == 0 ? cA : cB;
    {
      V64 x[2];
      gate_pair(x, cA);
      {                                                    // the tails of both sites' images (tile S's rows); its gate phase comes later
        V64 sr;
        load_v64(sr, Sr[2] + (size_t)cS * 64, kq);
        Frag3 sf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) split_8<false>(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const unsigned o = woS ^ (32u * (2 * ks + u));
            *reinterpret_cast<u32x2*>(imw + o) = (u32x2){sf[ks].h[2 * u], sf[ks].h[2 * u + 1]};
            *reinterpret_cast<u32x2*>(imw + o + wmS) = (u32x2){sf[ks].m[2 * u], sf[ks].m[2 * u + 1]};
          }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      chain_pair(x, mcA, std::false_type{});
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) xgS.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xg_half(xgS, std::integral_constant<int, 0>{});
    {
      V64 x[2];
      gate_pair(x, cB);                                    // rows 0..31 now hold site B
      xg_half(xgS, std::integral_constant<int, 1>{});
      {                                                    // tile S: gate phase (rows re-read: kept from above they would cost 16 registers
        V64 xS;                                            // across two gate phases and a chain), then its chain
        {
          V64 sr, sm, um, ur;
          load_v64(sr, Sr[2] + (size_t)cS * 64, kq);
          load_v64(sm, Sm + (size_t)cS * 64, kq);
          load_v64(um, Um + (size_t)cS * 64, kq);
          Frag3 sf[2];
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) split_8<false>(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
          gate_init16(ur, um, cv, sgn[2], kq);
          linear_t16p_core<4, WPF>(ur.t, sf, Wh_l, lane, [] {});
          gate16(xS, sr, ur, sm);
        }
        V64 gg;
        linear_t16p<4, false, true, WPF, false>(gg.t, xgS, Wg_l, cv + 64, lane);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) gate_mix4(xS.t[mt], xgS.t[mt], gg.t[mt]);          // (1-w)*x + w*x_g
        V64 s1;
        linear_t16p<4, false, true, WPF, false>(s1.t, xS, S0_l, cv + 128, lane);
        f32x2v s2 = {0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 16 * mt + 4 * kq);
          gelu_dot4(s2, s1.t[mt], w4);
        }
        float s = s2[0] + s2[1];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        score[2] += (s + w.s2b) * mcS;
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
      chain_pair(x, mcB, std::true_type{});
    }
  }
  // one partial set per WORKGROUP: per wave tile S's two columns of a pair are added A then B, then the waves in order
  __syncthreads();
  float* red = smem + 3 * IMG64;                           // [NW][64] (the buffers are dead)
  red[wave * 64 + lane] = 0.f;
  __syncthreads();
  if (kq == 0) {
    red[wave * 64 + l15] = score[0];
    red[wave * 64 + 16 + l15] = score[1];
  }
  for (int gi = 0; gi < 2; ++gi) {
    if (kq == 0 && validS && siteS == gi) red[wave * 64 + pS] += score[2];
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  if (tid < 64) {
    float v = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < NW; ++s_) v += red[s_ * 64 + tid];
    score_part[((size_t)b * gridDim.x + sc) * 64 + tid] = v;
  }
}

// floats of dynamic LDS k_inc_score_g<NT, G, IR, NW> needs
constexpr int inc_score_g_lds(int NT, int G, int IR, int NW) {
  return 3 * IMG64 + NW * G * IR * 64 + ((32 * ((IR + 31) / 32) - IR) * 32 + 32 + 64) + SCORER_CONSTS + 2 * (16 * NT + 1) * 32;
}
