// nnj_scorer_wide.hpp -- the pair scorer for MORE THAN 64 live rows (up to 256): alignments of 100 taxa (the
// reference's bundled evaluation set) and BASELINE configs[4] (200 x 4096).  The kernels of nnj_scorer.hpp /
// nnj_scorer16.hpp keep all rows of a site in 64-row LDS images and are used as soon as a rollout is down to 64
// rows; above that these three kernels score "star" pair sets
//        { (m, r) : r in the live rows }            one set per batch element (the incremental step: m = the
//                                                    freshly merged row, reference model.py:184-201)
//        { (m, r) : r > m },  m = m0 .. m0+M-1       the all-pairs table of step 0 as n-1 stars
//                                                    (reference model.py:168-181)
// with the same arithmetic as the 16-pair kernels (f16x3 GEMMs, V64 tiles, gate16, K'-free alpha -- see
// k_inc_alpha16): a workgroup of 8 waves owns one (site chunk, m, batch element); wave w owns the pair tiles
// t = w + 8*pt (pt < PT, rows 16t .. 16t+15), RP = 128*PT padded rows, PT = 1 (n <= 128) or 2 (n <= 256).  Per
// site the waves write the rows they hold into ONE LDS image (alpha: S as a weight-like image [RP r'][64 d];
// score: S^T [64 d][RP r']) between two workgroup barriers and then multiply their own pairs with all of it.
// Work per star is n pairs x n context rows; a star re-reads the n rows of its sites (L2 / MALL resident:
// consecutive workgroups are the stars of one site chunk).
//   alpha_part [b][mi][RP pair r][sc][RP r']   alpha [b][mi][RP][RP]   score_part [b][mi][sc][RP]
//   (the partials of a pair row are contiguous: the softmax kernel streams nsc x RP floats per row; with the site chunk
//    outermost its reads were a power-of-two stride of RP x RP floats apart and met in the same channels)
#pragma once
#include "nnj_scorer16.hpp"

struct WideGeom {
  int PT, RP;          // pair tiles per wave, padded rows (128 * PT)
  int M;               // stars of the launch group (1: incremental; n-1: all pairs)
  int MB;              // stars per launch (workspace bound), launches = ceil(M / MB)
  int nsc, cs;         // site chunks, sites per chunk
};

// star of this workgroup: merged row position m, and whether lane row r is a wanted pair
struct WideStar { int m, slot_m, mi; };
__device__ __forceinline__ WideStar wide_star(const RowSet& rs, const int* ij_prev, int m0, int b, int n) {
  WideStar s;
  s.mi = blockIdx.y;
  s.m = ij_prev ? min(max(ij_prev[2 * b], 0), n - 1) : m0 + (int)blockIdx.y;
  s.slot_m = slot_of(rs, b, s.m);
  return s;
}
// tile t (rows 16t..16t+15) holds a wanted pair of the star: r != m, r < n, and r > m for the all-pairs stars
__device__ __forceinline__ bool wide_tile_active(int t, int m, int n, bool full) {
  const int lo = 16 * t, hi = min(16 * t + 15, n - 1);
  if (lo >= n) return false;
  return full ? hi > m : true;
}

// ------------------------------------------------------------------ k_wide_alpha
// PT = 2: the two pair tiles of a wave are accumulated by TWO workgroups (blockIdx.x = sc * PT + pth): both build the
// whole image, each keeps the 16 accumulators of one of its tiles (128 accumulators per lane spill).
template <int PT>
__global__ __launch_bounds__(512) void k_wide_alpha(RowSet rs, ScorerW w, const int* __restrict__ ij_prev, int m0,
                                                    float* __restrict__ alpha_part, int n, int C, int cs, int nsc) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RP = 128 * PT, NGT = 8 * PT;               // padded rows, 16-row tiles of the image
  float* At_l = smem;                                      // A^T, IMG64 floats
  float* Wh_l = smem + IMG64;                              // W_h
  float* img = smem + 2 * IMG64;                           // [2 planes][RP r'][8 chunks of 16 B]
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sc = blockIdx.x / PT, pth = blockIdx.x % PT, b = blockIdx.z;
  const bool full = ij_prev == nullptr;
  const WideStar st = wide_star(rs, ij_prev, m0, b, n);
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(At_l, w.imgAt, tid, 512);
  stage_image_t16(Wh_l, w.imgWh, tid, 512);
  float* cv = smem + 2 * IMG64 + 64 * RP;
  stage_scorer_consts(cv, w, tid);
  const size_t bo = (size_t)b * rs.bstride;
  const float* Sm = rs.S + bo + (size_t)st.slot_m * C * 64;
  const float* Um = rs.U + bo + (size_t)st.slot_m * C * 64;
  const float* Sr[PT];
  int rr[PT];
  float sgn[PT];
  bool act[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int t = wave + 8 * pt;
    rr[pt] = 16 * t + l15;
    Sr[pt] = rs.S + bo + (size_t)slot_of(rs, b, rr[pt] < n ? rr[pt] : 0) * C * 64;   // beyond the rows: row 0 (finite, unused)
    sgn[pt] = rr[pt] < st.m ? 1.0f : -1.0f;
    act[pt] = wide_tile_active(t, st.m, n, full);
  }
  f32x4 acc[NGT];
#pragma unroll
  for (int mt = 0; mt < NGT; ++mt) acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  u32x4* im4 = reinterpret_cast<u32x4*>(img);
  constexpr int PL = RP * 8;                               // plane stride in 16-byte units
  const u32x4* wh4 = reinterpret_cast<const u32x4*>(Wh_l);
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    asm volatile("" ::: "memory");
    V64 x[PT];
    V64 sm, um;
    load_v64(sm, Sm + (size_t)c * 64, kq);
    load_v64(um, Um + (size_t)c * 64, kq);
    __syncthreads();                                       // every wave is done with the previous site's image
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      V64 sr, ur;
      load_v64(sr, Sr[pt] + (size_t)c * 64, kq);
      Frag3 sf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
      // row r of the image: chunk 4*ks + kq = this lane's tiles 2ks, 2ks+1 (the operand order of linear_t16)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int o = rr[pt] * 8 + wswz6<8>(rr[pt], 4 * ks + kq);
        im4[o] = sf[ks].h; im4[PL + o] = sf[ks].m;
      }
      // the pair vectors exist for the tiles that hold a wanted pair only (wave-uniform): a tile of rows r <= m of an
      // all-pairs star, or beyond the rows, contributes its rows to the image and nothing else
      if (!act[pt]) continue;
      gate_init16(ur, um, cv, sgn[pt], kq);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int row = 16 * mt + l15;
          const int o = row * 8 + wswz6<8>(row, 4 * ks + kq);
          Frag3 a;
          a.h = wh4[o]; a.m = wh4[64 * 8 + o];
          ur.t[mt] = mfma16_b6(a, sf[ks], ur.t[mt]);      // U_r = W_h S_r, recomputed (see k_inc_alpha16)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      gate16(x[pt], sr, ur, sm);
    }
    __syncthreads();                                       // all RP rows are in the image
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      if (pt != pth || !act[pt]) continue;                 // wave-uniform
      V64 xp;
      linear_t16<4, false, false>(xp.t, x[pt], At_l, nullptr, lane);          // x' = A^T x
      linear_t16<NGT, true, false>(acc, xp, img, nullptr, lane);              // acc[r'][pair] += S_r' . x'
    }
  }
  // partial sums of this site chunk: [pair r][r'], the lane's four r' of tile mt at 16mt + 4kq
  float* dst = alpha_part + ((size_t)b * gridDim.y + st.mi) * ((size_t)nsc * RP * RP) + (size_t)sc * RP;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    if (pt != pth || !act[pt]) continue;
#pragma unroll
    for (int mt = 0; mt < NGT; ++mt)
      *reinterpret_cast<f32x4*>(dst + (size_t)rr[pt] * nsc * RP + 16 * mt + 4 * kq) = acc[mt];
  }
}

// ------------------------------------------------------------------ k_wide_softmax
// alpha[b][mi][r][r'] = softmax_r'( (sum_sc part + beta_r') / sqrt(64 C) ) over the live rows other than m and r
// (model.py:118-146); pairs the star does not want are written as zeros.  One wave per pair row, lane = r' mod 64.
__global__ __launch_bounds__(256) void k_wide_softmax(RowSet rs, ScorerW w, const int* __restrict__ ij_prev, int m0,
                                                      const float* __restrict__ alpha_part,
                                                      float* __restrict__ alpha, int n, int C, int RP, int nsc,
                                                      const float* __restrict__ beta_slot, int nslot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + wave, b = blockIdx.z;
  if (r >= RP) return;
  const bool full = ij_prev == nullptr;
  const WideStar st = wide_star(rs, ij_prev, m0, b, n);
  const size_t star = (size_t)b * gridDim.y + st.mi;
  const size_t orow = (star * RP + r) * RP;                 // alpha is written as two fp16 planes (pre-split)
  const long apl = (long)gridDim.z * gridDim.y * RP * RP;
  const bool wanted = r < n && r != st.m && (!full || r > st.m);
  const int nk = RP / 64;
  if (!wanted) {
    for (int k = 0; k < nk; ++k) alpha_store(alpha, apl, orow + 64 * k + lane, 0.f);
    return;
  }
  const float inv = 1.0f / sqrtf(64.0f * (float)C);
  float a[4];
  float mx = -INFINITY;
  for (int k = 0; k < nk; ++k) {
    const int rp = 64 * k + lane;
    float s = 0.f;
#pragma unroll 8          // independent loads in flight; the additions stay in order
    for (int sc = 0; sc < nsc; ++sc) s += alpha_part[((star * RP + r) * nsc + sc) * RP + rp];
    const bool in = rp < n && rp != st.m && rp != r;
    const float beta = rp < n ? beta_slot[(size_t)b * nslot + slot_of(rs, b, rp)] : 0.f;   // (k_beta_sum, once per row)
    a[k] = in ? (s + beta) * inv : -INFINITY;
    mx = fmaxf(mx, a[k]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float sum = 0.f;
  for (int k = 0; k < nk; ++k) {
    a[k] = a[k] == -INFINITY ? 0.f : expf(a[k] - mx);
    sum += a[k];
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
  for (int k = 0; k < nk; ++k) alpha_store(alpha, apl, orow + 64 * k + lane, sum > 0.f ? a[k] / sum : 0.f);
}

// ------------------------------------------------------------------ k_wide_score
template <int PT>
__global__ __launch_bounds__(512) void k_wide_score(RowSet rs, ScorerW w, const int* __restrict__ ij_prev, int m0,
                                                    const float* __restrict__ alpha, const uint8_t* __restrict__ mask,
                                                    float* __restrict__ score_part, int n, int C, int cs) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RP = 128 * PT;
  constexpr int CH = RP / 8;                               // 16-byte chunks per image row (8 r' each): 16 or 32
  constexpr int KSX = RP / 32;                             // k-steps of the x_g GEMM
  constexpr int PLH = 64 * RP;                             // plane stride in fp16
  constexpr int PL4 = 64 * CH;                             // plane stride in 16-byte units
  float* Wg_l = smem;
  float* S0_l = smem + IMG64;
  float* Wh_l = smem + 2 * IMG64;
  float* img = smem + 3 * IMG64;                           // S^T: [2 planes][64 d][RP r'] fp16
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sc = blockIdx.x, b = blockIdx.z;
  const bool full = ij_prev == nullptr;
  const WideStar st = wide_star(rs, ij_prev, m0, b, n);
  const int c0 = sc * cs, c1 = min(C, c0 + cs);
  stage_image_t16(Wg_l, w.imgWg, tid, 512);
  stage_image_t16(S0_l, w.imgS0, tid, 512);
  stage_image_t16(Wh_l, w.imgWh, tid, 512);
  float* cv = smem + 3 * IMG64 + 64 * RP;
  stage_scorer_consts(cv, w, tid);
  const size_t bo = (size_t)b * rs.bstride;
  const float* Sm = rs.S + bo + (size_t)st.slot_m * C * 64;
  const float* Um = rs.U + bo + (size_t)st.slot_m * C * 64;
  const size_t star = (size_t)b * gridDim.y + st.mi;
  const float* Sr[PT];
  size_t ap[PT];                                           // element offsets into the alpha planes
  const long apl = (long)gridDim.z * gridDim.y * RP * RP;
  int rr[PT];
  float sgn[PT], score[PT];
  bool act[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int t = wave + 8 * pt;
    rr[pt] = 16 * t + l15;
    Sr[pt] = rs.S + bo + (size_t)slot_of(rs, b, rr[pt] < n ? rr[pt] : 0) * C * 64;
    ap[pt] = (star * RP + rr[pt]) * RP + 8 * kq;
    sgn[pt] = rr[pt] < st.m ? 1.0f : -1.0f;
    act[pt] = wide_tile_active(t, st.m, n, full);
    score[pt] = 0.f;
  }
  unsigned short* t16 = reinterpret_cast<unsigned short*>(img);
  const u32x4* im4 = reinterpret_cast<const u32x4*>(img);
  const u32x4* wh4 = reinterpret_cast<const u32x4*>(Wh_l);
  __syncthreads();
  for (int c = c0; c < c1; ++c) {
    asm volatile("" ::: "memory");
    const float mc = mask[(size_t)b * C + c] ? 0.f : 1.f;                // seq_mask (model.py:96); first load of the iteration
    V64 x[PT];
    V64 sm, um;
    load_v64(sm, Sm + (size_t)c * 64, kq);
    load_v64(um, Um + (size_t)c * 64, kq);
    __syncthreads();                                       // every wave is done with the previous site's image
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      V64 sr, ur;
      load_v64(sr, Sr[pt] + (size_t)c * 64, kq);
      Frag3 sf[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) split_8(sf[ks], sr.t[2 * ks], sr.t[2 * ks + 1]);
      // column r of the image: chunk r >> 3, element r & 7 (natural r' order)
      const int wchunk = rr[pt] >> 3, we = rr[pt] & 7;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const unsigned h = sf[mt >> 1].h[2 * (mt & 1) + pr], m = sf[mt >> 1].m[2 * (mt & 1) + pr];
          const int d0 = 16 * mt + 4 * kq + 2 * pr, d1 = d0 + 1;
          const int o0 = d0 * RP + 8 * wswz6<CH>(d0, wchunk) + we, o1 = d1 * RP + 8 * wswz6<CH>(d1, wchunk) + we;
          t16[o0] = (unsigned short)h; t16[o1] = (unsigned short)(h >> 16);
          t16[PLH + o0] = (unsigned short)m; t16[PLH + o1] = (unsigned short)(m >> 16);
        }
      if (!act[pt]) continue;                              // (see k_wide_alpha: no pair vector for a tile without a wanted pair)
      gate_init16(ur, um, cv, sgn[pt], kq);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int row = 16 * mt + l15;
          const int o = row * 8 + wswz6<8>(row, 4 * ks + kq);
          Frag3 a;
          a.h = wh4[o]; a.m = wh4[64 * 8 + o];
          ur.t[mt] = mfma16_b6(a, sf[ks], ur.t[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      gate16(x[pt], sr, ur, sm);
    }
    __syncthreads();                                       // all RP columns are in the image
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      if (!act[pt]) continue;                              // wave-uniform
      V64 xg, g;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) xg.t[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // x_g^T = S^T alpha^T: k-slot 8kq + j of k-step ks = r' = 32ks + 8kq + j (alpha is exactly 0 beyond the rows)
#pragma unroll 2
      for (int ks = 0; ks < KSX; ++ks) {
        Frag3 bfr;
        alpha_frag16(bfr, alpha, apl, ap[pt] + 32 * ks);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int d = 16 * mt + l15;
          const int o = d * CH + wswz6<CH>(d, 4 * ks + kq);
          Frag3 a;
          a.h = im4[o]; a.m = im4[PL4 + o];
          xg.t[mt] = mfma16_b6(a, bfr, xg.t[mt]);
        }
      }
      linear_t16<4, false>(g.t, xg, Wg_l, cv + 64, lane);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) gate_mix4(x[pt].t[mt], xg.t[mt], g.t[mt]);          // (1-w)*x + w*x_g
      V64 s1;
      linear_t16<4, false>(s1.t, x[pt], S0_l, cv + 128, lane);
      f32x2v s2 = {0.f, 0.f};
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(cv + 192 + 16 * mt + 4 * kq);
        gelu_dot4(s2, s1.t[mt], w4);
      }
      float s = s2[0] + s2[1];
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      score[pt] += (s + w.s2b) * mc;
    }
  }
  if (kq == 0) {
    float* dst = score_part + (star * gridDim.x + sc) * RP;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) dst[rr[pt]] = act[pt] ? score[pt] : 0.f;
  }
}

// all-pairs table of step 0 from the stars: full[b][pair_index(n, m, r)] = sum_sc score_part[b][m - m0][sc][r], r > m
__global__ void k_wide_gather_full(const float* __restrict__ score_part, float* __restrict__ full, int n, int RP,
                                   int nsc, int m0, int mcount) {
  const int b = blockIdx.y;
  const int np = num_pairs(n);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < mcount * RP; i += gridDim.x * blockDim.x) {
    const int mi = i / RP, r = i % RP, m = m0 + mi;
    if (r <= m || r >= n) continue;
    float s = 0.f;
    for (int sc = 0; sc < nsc; ++sc) s += score_part[(((size_t)b * mcount + mi) * nsc + sc) * RP + r];
    full[(size_t)b * np + pair_index(n, m, r)] = s;
  }
}
