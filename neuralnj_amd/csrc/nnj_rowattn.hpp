// nnj_rowattn.hpp -- tied row attention of the MSA encoder on the fp16 matrix pipe
// (restates reference axial_attention.py:6-138: q,k,v projections, q scaling and padding,
//  attention over the C alignment columns with head dimension E = R*8, softmax over keys).
//
// The attention of one (batch element, head) is two fp32-accurate GEMMs with contraction lengths
// E = 8R (scores) and C (context).  Both run as "f16x3" (nnj_common.hpp): operands stored as two
// fp16 planes, three piece products per fp32 product, fp32 accumulation.  A flash-style single kernel
// would have to keep a query block stationary at 4 bytes per element (51 KiB per 32 queries at R = 50)
// next to its 208 context accumulators, or re-stream K and V per small query block; the scores therefore
// make one round trip through HBM (288 GB; 8.6 GB per layer at B = 256, C = 1024):
//
//   k_qkv6    LN -> q,k,v projections -> Q6, K6 (row-major operand tiles) and V6 (transposed operand tiles),
//             already split into planes and already in the LDS image order of the consumers
//   k_row_s   S^T = K Q^T per 256x256 block, key classes applied (padded key -> fill, beyond C -> -inf),
//             written as 32x32 C-layout register images + per-tile row maxima
//   k_row_pv  per 128 queries: P = exp(S - max) (split in registers), O^T = V^T P^T, ctx = O / sum(P)
//
// Operand tile ("A tile"): [2 planes][rows][16 k] fp16 = 32 bytes per row and plane; the two 16-byte
// halves of a row (k-slots 0-7, 8-15) are stored swapped when bit 3 of the row index is set, which makes
// the ds_read_b128 of 32 rows x one half conflict free (16 lanes of a read group hit 16 different
// 16-byte slots).  One MFMA k-step consumes one tile row per lane: lane (row r, half HH) reads its 8
// fp16 with one ds_read_b128 per plane.
//   Q6/K6: [bh][ks][row block of 256][plane][256 rows c][32 B]   k-slot 8*(r&1) + d, e = 8r + d, ks = r>>1
//   V6   : [bh][k16][plane][ET*32 rows e][32 B]                  k-slot 8*HH + 4*jj + t for key
//                                                                 16*k16 + 8*jj + 4*HH + t
//          (the key order of registers 8G..8G+7 of a 32x32 C-layout tile, so that P^T is a B operand as is)
//   S    : [bh][qt][kt][4][64 lanes][4] fp32: tile (32 keys x 32 queries) of S^T as the accumulator registers
//          (group g = registers 4g..4g+3) of the wave that computed it; M: [bh][qt][kt][32 queries] tile maxima.
#pragma once
#include "nnj_encoder.hpp"

struct Ra6 {
  int C, T;        // alignment columns (keys = queries), rows
  int KS;          // 16-wide k-steps of the head dimension: ceil(8T/16)
  int Cp;          // C rounded up to 256
  int ET;          // 32-row tiles of the head dimension in V6 (= nech * ETc)
  int ETc;         // ... per e-chunk of k_row_pv (bucketed, <= 16: its context accumulators)
  int nech;        // e-chunks: k_row_pv workgroups per query block (1 up to 64 rows; 100 rows: 2 x 13 tiles; 200: 4 x 13)
  int nrb;         // Cp / 256
  int nt32;        // Cp / 32
  int nk16;        // Cp / 16
  int Epad;        // row length of ctx
  size_t qk_bh;    // bytes of Q6 (= K6) per (b, h)
  size_t v_bh;     // bytes of V6 per (b, h)
  size_t s_bh;     // floats of S per (b, h)
  size_t m_bh;     // floats of M per (b, h)
};
// B: the batch the geometry is for.  A small batch leaves k_row_pv's grid (B x 8 heads x C / 128 query blocks) far below the
// chip's 256 CUs -- one alignment of 1024 columns: 64 workgroups walking all keys with 13 accumulator tiles each -- so its
// head dimension is cut into more e-chunks than the registers ask for (every chunk recomputes the probabilities and owns
// its slice of the context rows: results are bit-identical for every cut).
inline Ra6 ra6_geom(int T, int C, int Epad, int B) {
  Ra6 g;
  g.C = C; g.T = T; g.Epad = Epad;
  g.KS = (8 * T + 15) / 16;
  g.Cp = (C + 255) / 256 * 256;
  static const int buckets[] = {1, 2, 4, 6, 7, 8, 10, 13, 16};
  const int need = (16 * g.KS + 31) / 32;
  g.nech = (need + 15) / 16;
  {
    const long wgs = (long)B * 8 * (g.Cp / 128);
    while (g.nech < 4 && wgs * g.nech < 256 && need >= 4 * g.nech) g.nech *= 2;
  }
  // 33..64 rows (9..16 head tiles): TWO e-chunks at every batch -- two workgroups of <= 8 accumulator tiles per CU, i.e. two
  // waves per SIMD, instead of one of 13 (50 rows: 7 + 7 tiles; twice the V6 DMA and the exponentials, 8 % padding, and
  // still 35.5 -> 34.1 ms per rollout: one wave's DMA issue and barrier waits sit under the other workgroup's MFMAs;
  // the same two halves inside ONE eight-wave workgroup were slower, 38.5: profiles/r04/ab_row_pv_chunks.txt).
  if (g.nech < 2 && need >= 9 && need <= 16) g.nech = 2;
  const int per = (need + g.nech - 1) / g.nech;
  g.ETc = 16;
  for (int k : buckets) if (per <= k) { g.ETc = k; break; }
  g.ET = g.nech * g.ETc;
  g.nrb = g.Cp / 256; g.nt32 = g.Cp / 32; g.nk16 = g.Cp / 16;
  g.qk_bh = (size_t)g.KS * g.nrb * NPL * 8192;
  g.v_bh = (size_t)g.nk16 * NPL * g.ET * 32 * 32;
  g.s_bh = (size_t)g.nt32 * g.nt32 * 1024;
  g.m_bh = (size_t)g.nt32 * g.nt32 * 32;
  return g;
}

// key classes of a batch element: 0 = key, 1 = padded key (axial_attention.py:99-103), 2 = beyond C
__global__ void k_key_classes(const uint8_t* __restrict__ mask, uint8_t* __restrict__ cls, int B, int C, int Cp) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * Cp) return;
  const int b = (int)(i / Cp), c = (int)(i % Cp);
  cls[i] = c >= C ? 2 : ((mask && mask[(size_t)b * C + c]) ? 1 : 0);
}

__device__ __forceinline__ unsigned dpp_quad_xor1(unsigned x) {   // value of lane ^ 1
  return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned dpp_quad_xor2(unsigned x) {   // value of lane ^ 2
  return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0x4E, 0xF, 0xF, true);
}

// ------------------------------------------------------------------ k_qkv6
// Persistent, 8 waves.  A wave owns 2 rows x 16 columns of the alignment (lane & 31 = 16*(r&1) + (c&15),
// column block aligned to 16): with this shape every store instruction below fills whole 32-byte sectors
// of both the row-major (Q6, K6) and the transposed (V6) tiles.  Tokens outside the alignment (odd R,
// C % 16 != 0) are written as zeros: they pad the contraction of the score GEMM.
__global__ __launch_bounds__(512) void k_qkv6(const float* __restrict__ x, const uint8_t* __restrict__ mask, AttnW wn,
                                              uint8_t* __restrict__ Q6, uint8_t* __restrict__ K6,
                                              uint8_t* __restrict__ V6, Ra6 g, int B) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                          // one f16x3 image [Wq | Wk | Wv] of 192 rows: y is split once
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  stage_weight_b6<64>(Wl, wn.Wq, 64, tid, 512, 0, 192);
  stage_weight_b6<64>(Wl, wn.Wk, 64, tid, 512, 64, 192);
  stage_weight_b6<64>(Wl, wn.Wv, 64, tid, 512, 128, 192);
  // LayerNorm scale / shift and the three bias vectors in LDS (persistent workgroup): see k_ffn16
  float* cq = smem + b6_floats(192, 64);     // ln_w | ln_b | bq | bk | bv, 64 floats each
  if (tid < 64) { cq[tid] = wn.ln_w[tid]; cq[64 + tid] = wn.ln_b[tid]; cq[128 + tid] = wn.bq[tid];
                  cq[192 + tid] = wn.bk[tid]; cq[256 + tid] = wn.bv[tid]; }
  __syncthreads();
  const int R = g.T, C = g.C;
  const int RP = (R + 1) / 2, CB = (C + 15) / 16;
  const int tiles_per_b = RP * CB, groups_per_b = (tiles_per_b + 7) / 8;
  const int ngroups = groups_per_b * B;
  const int lam = lane & 31, c16 = lam & 15, rr = lam >> 4;
  const unsigned sel = (lane & 1) ? 0x03020706u : 0x05040100u;      // see the 4x4 transpose below
  const bool bit1 = (lane >> 1) & 1;
  // token address of this lane in group g_ (tokens outside the alignment read token (0,0) of the batch element)
  auto tok_addr = [&](int g_) {
    const int b_ = g_ / groups_per_b;
    const int tile_ = min((g_ % groups_per_b) * 8 + wave, tiles_per_b - 1);
    const int r_ = 2 * (tile_ / CB) + rr, c_ = 16 * (tile_ % CB) + c16;
    const bool v_ = r_ < R && c_ < C;
    return x + (((size_t)b_ * R + (v_ ? r_ : 0)) * C + (v_ ? c_ : 0)) * 64;
  };
  f32x16 xr[2];
  if ((int)blockIdx.x < ngroups) load_token64(xr, tok_addr(blockIdx.x), true, hh);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int b = grp / groups_per_b;
    const int tile = (grp % groups_per_b) * 8 + wave;
    const bool tile_ok = tile < tiles_per_b;                         // wave-uniform
    const int rp = tile / CB, cb = tile % CB;
    const int r = 2 * rp + rr, c = 16 * cb + c16;
    const bool valid = tile_ok && r < R && c < C;
    asm volatile("" ::: "memory");      // keep the parameter loads inside the loop (hoisted, they would occupy ~200 VGPRs)
    f32x16 y[1][2], o[1][6];
    layer_norm64(y[0], xr, cq, cq + 64, hh, wn.dt);
    // the next group's tokens go into the registers LayerNorm has just consumed: their HBM latency runs
    // behind the GEMM and the stores
    if (grp + (int)gridDim.x < ngroups) load_token64(xr, tok_addr(grp + gridDim.x), true, hh);
    if (!tile_ok) continue;
    const bool padded = mask && valid && mask[(size_t)b * C + c];
    // q goes to the operand planes UNSCALED (zero for padded sites, axial_attention.py:78-82): scaled by
    // dh^-0.5 / sqrt(R) its low fp16 piece would sit in the fp16 denormal range from about 50 rows on and the pair
    // would carry 19-20 instead of 22 significand bits; k_row_s applies the scale to the fp32 logits instead
    const float qscale = padded ? 0.0f : 1.0f;
    linear6_T_nb<6, 2, 1>(o, y, Wl, lane);
    const int rb = c >> 8;
    const size_t qk_row = (size_t)(c & 255) * 32 + 16 * (rr ^ ((c >> 3) & 1)) + 8 * hh;
    const size_t v_col = 16 * (((c16 >> 2) & 1) ^ rr) + 8 * (c16 >> 3);
    const int e_row = 8 * r + 4 * hh + (lane & 3);                   // the V6 row this lane stores after the transpose
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int h = 4 * mt + gq;
        const size_t bh = (size_t)b * NNJ_NHEAD + h;
        // ---- q and k: 8 bytes (d = 4hh..4hh+3) of the 16-byte half (row r) of column c's tile row
#pragma unroll
        for (int ten = 0; ten < 2; ++ten) {
          const float* bias = ten ? cq + 192 : cq + 128;
          const float scale = ten ? 1.0f : qscale;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * mt + 8 * gq + 4 * hh);
          float v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = valid ? (b4[t] + o[0][2 * ten + mt][4 * gq + t]) * scale : 0.f;
          unsigned p01[NPL], p23[NPL];
          split2(v[0], v[1], p01[0], p01[1]);
          split2(v[2], v[3], p23[0], p23[1]);
          uint8_t* dst = (ten ? K6 : Q6) + bh * g.qk_bh + ((size_t)(rp * g.nrb + rb) * NPL) * 8192 + qk_row;
#pragma unroll
          for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(dst + (size_t)p * 8192) = make_uint2(p01[p], p23[p]);
        }
        // ---- v: transposed.  The lane holds four rows e (t = 0..3) of ONE key; a 4x4 transpose over the
        // quad (four consecutive keys) gives it one row e = 8r + 4hh + (lane&3) of FOUR keys = 8 bytes.
        {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(cq + 256 + 32 * mt + 8 * gq + 4 * hh);
          float v[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) v[t] = valid ? b4[t] + o[0][4 + mt][4 * gq + t] : 0.f;
          unsigned p01[NPL], p23[NPL];
          split2(v[0], v[1], p01[0], p01[1]);
          split2(v[2], v[3], p23[0], p23[1]);
          uint8_t* dst = V6 + bh * g.v_bh + ((size_t)cb * NPL * (g.ET * 32) + e_row) * 32 + v_col;
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            // step 1 (lane ^ 1): even lanes collect element t=0 (and 2) of the key pair, odd lanes t=1 (and 3)
            const unsigned x01 = dpp_quad_xor1(p01[p]), x23 = dpp_quad_xor1(p23[p]);
            const unsigned q_lo = __builtin_amdgcn_perm(x01, p01[p], sel);
            const unsigned q_hi = __builtin_amdgcn_perm(x23, p23[p], sel);
            // step 2 (lane ^ 2): lanes 0,1 keep t = 0,1 and receive the other key pair; lanes 2,3 keep t = 2,3
            const unsigned recv = dpp_quad_xor2(bit1 ? q_lo : q_hi);
            const unsigned keep = bit1 ? q_hi : q_lo;
            *reinterpret_cast<uint2*>(dst + (size_t)p * (g.ET * 32) * 32) =
                bit1 ? make_uint2(recv, keep) : make_uint2(keep, recv);
          }
        }
      }
  }
}

// ------------------------------------------------------------------ k_row_s
// S^T block [256 keys x QW queries] of one (b, h): 4 waves, wave = 128 keys x QW/2 queries = 4 x QW/64 MFMA
// tiles, contraction over the KS operand tiles of 16.  Operand tiles stream HBM/L2 -> LDS by LDS-DMA into a
// three-stage ring (stage = K tile 16 KiB | Q tile QW/16 KiB); fragments are read with hand-issued ds_read_b128
// (the compiler would drain the in-flight DMA before reads of its own); the B fragment of query tile j+1 is in
// flight behind the 12 MFMAs of group j.  One workgroup barrier per k-step.
// QW = 128 is the instantiation used (wave = 128 x 64, 128 accumulator registers, three 24-KiB stages): TWO
// workgroups per CU, which are not synchronised with each other -- one computes while the other issues DMA
// pieces, waits at its barrier or stores its tiles.
template <int QW>
struct RsShape {
  static constexpr int NJ = QW / 64;                                 // query tiles per wave
  static constexpr int NST = 3;                                      // 3 x 24 KiB stages: two workgroups per CU still fit
  static constexpr unsigned QPL = QW * 32u;                          // bytes of one plane of the Q tile
  static constexpr unsigned KTB = NPL * 8192u;                       // bytes of a K tile (256 rows x 16 k, NPL planes)
  static constexpr unsigned STAGE = KTB + NPL * QPL;
  static constexpr int NPK = NPL * 8;                                // 1-KiB pieces of the K tile
  static constexpr int NP = (NPK + NPL * QW / 32) / 4;               // DMA pieces per wave and k-step
};
// CHK > 0 (used above 64 alignment rows, with QW = 64): CHUNKED accumulation.  A logit is a sum of 8R products; on
// the matrix pipe every MFMA rounds the running fp32 sum once, i.e. 3 x R/2 roundings at the magnitude of the logit
// itself -- measured 1.1e-4 absolute on logits of magnitude 100 at R = 200 (stress weights), against 2e-5 for the
// reference, whose no-grad path adds per-chunk partial sums (axial_attention.py:35-64).  With CHK the k-steps are
// accumulated CHK at a time into a zeroed accumulator and the chunk sums are added to the total by the vector
// unit: the many roundings happen on small partial sums, only KS/CHK on the large one (same error as the
// reference's order).  The second accumulator set is paid for by the narrower wave tile (128 keys x 32 queries).
#ifndef NNJ_RS_CHK
#define NNJ_RS_CHK 8
#endif
// (Register staging of the operand tiles instead of LDS-DMA, an error-free fold of the chunk sums and one fused kernel for
// scores and context were built and measured in round 4 -- profiles/r04/ab_rowattn_regstage.txt, noise_variants.txt,
// ab_rowfused_*.txt; none faster / closer: the sources are in the git history and tools/experiments/.)
template <int QW, int CHK = 0>
__global__ __launch_bounds__(256, 2) void k_row_s(const uint8_t* __restrict__ Q6, const uint8_t* __restrict__ K6,
                                                             const uint8_t* __restrict__ cls, float* __restrict__ S,
                                                             float* __restrict__ M, Ra6 g, int nbh, float fill, float qs) {
  using SH = RsShape<QW>;
  constexpr int NJ = SH::NJ, NP = SH::NP, NST = SH::NST;
  constexpr unsigned QPL = SH::QPL, STAGE = SH::STAGE, KTB = SH::KTB;
  constexpr int NPK = SH::NPK;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, HH = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = g.Cp / QW;
  const int tiles_per_bh = g.nrb * nqb;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;            // all blocks of one (b,h) on one XCD (one L2)
  const int bh = (slot / tiles_per_bh) * 8 + xcd;
  if (bh >= nbh) return;
  const int tile = slot % tiles_per_bh, kb = tile / nqb, qb = tile % nqb;
  const int kh = wave >> 1, qh = wave & 1;                           // key half, query half
  const int KS = g.KS;
  const int q0 = qb * QW;                                            // first query of the block
  const uint8_t* Kt = K6 + (size_t)bh * g.qk_bh + (size_t)kb * NPL * 8192;
  const uint8_t* Qt = Q6 + (size_t)bh * g.qk_bh + (size_t)(q0 >> 8) * NPL * 8192 + (size_t)(q0 & 255) * 32;
  const size_t ks_stride = (size_t)g.nrb * NPL * 8192;
  // DMA piece i (of NP per wave and tile pair): 1 KiB of the K tile (pieces 0..23 of the workgroup) or of one
  // plane of the Q tile.  An LDS-DMA instruction costs 60-185 issue cycles: the pieces of the next tile are
  // spread over the MFMA groups of a k-step instead of being issued in one burst.
  auto issue_piece = [&](auto pi, int ks, int stage) {
    constexpr int i = decltype(pi)::value;
    const int kk = ks < KS ? ks : KS - 1;                            // past the end: harmless reload into a free stage
    const int I = wave * NP + i;                                     // wave-uniform
    const uint8_t* src;
    if (I < NPK) {
      src = Kt + I * 1024;
    } else {
      const int J = I - NPK, pl = J / (QW / 32), pc = J % (QW / 32); // plane, 1-KiB piece of the plane
      src = Qt + (size_t)pl * 8192 + pc * 1024;
    }
    uint8_t* dst = reinterpret_cast<uint8_t*>(smem) + stage * STAGE + I * 1024;
    lds_dma16(reinterpret_cast<const float*>(src + (size_t)kk * ks_stride + lane * 16), reinterpret_cast<float*>(dst));
  };
  auto issue = [&](int ks, int stage) {
    static_for<0, NP>([&](auto pi) { issue_piece(pi, ks, stage); });
  };
  const unsigned half = 16u * (unsigned)(HH ^ ((l31 >> 3) & 1));
  const unsigned aA = lds_addr(smem) + (unsigned)(kh * 128 + l31) * 32u + half;
  const unsigned aB = lds_addr(smem) + KTB + (unsigned)(qh * 32 * NJ + l31) * 32u + half;
  f32x16 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  f32x16 tot[CHK > 0 ? 4 : 1][CHK > 0 ? NJ : 1];          // chunk sums are folded into this (CHK > 0)
  if constexpr (CHK > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  }
  Frag3 A[4], Bf[2];
  auto readA = [&](unsigned base) {
    static_for<0, 4>([&](auto ii) {
      constexpr int i = decltype(ii)::value;
      lds_read_frag<i * 1024>(A[i].h, base);
      lds_read_frag<i * 1024 + 8192>(A[i].m, base);
    });
  };
  auto readB = [&](Frag3& bfr, unsigned base, auto jj) {
    constexpr int j = decltype(jj)::value;
    lds_read_frag<j * 1024>(bfr.h, base);
    lds_read_frag<j * 1024 + QPL>(bfr.m, base);
  };
  issue(0, 0);
  if constexpr (NST == 3) issue(1, 1);
  auto kstep = [&](int ks) {
    const unsigned so = (unsigned)(ks % NST) * STAGE;
    if constexpr (NST == 3) wait_vmem_le<NP>(); else wait_vmem_le<0>();   // tile ks has landed
    barrier_nofence();                       // ... for every wave; every wave is done with tile ks-1
    const int st2 = (ks + NST - 1) % NST;    // the stage of tile ks-1 takes tile ks+NST-1
    readA(aA + so);
    readB(Bf[0], aB + so, std::integral_constant<int, 0>{});
    static_for<0, NJ>([&](auto jj) {
      constexpr int j = decltype(jj)::value;
      lds_wait_all();
      if constexpr (j == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pin_frag(A[i]);
      }
      pin_frag(Bf[j & 1]);
      if constexpr (j + 1 < NJ) readB(Bf[(j + 1) & 1], aB + so, std::integral_constant<int, j + 1>{});
      static_for<0, 4>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        acc[i][j] = mfma_b6(A[i], Bf[j & 1], acc[i][j]);
        // pieces behind the first MFMA groups; a k-step with few groups takes two per group
        constexpr int grp = 4 * j + i, per = (NP + 4 * NJ - 2) / (4 * NJ - 1);
        static_for<0, per>([&](auto ee) {
          constexpr int piece = grp * per + decltype(ee)::value;
          if constexpr (piece < NP) issue_piece(std::integral_constant<int, piece>{}, ks + NST - 1, st2);
        });
      });
    });
    if constexpr (CHK > 0) {
      if ((ks + 1) % CHK == 0 || ks + 1 == KS) {           // fold the chunk sum into the total, restart the chunk
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            tot[i][j] += acc[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
          }
      }
    }
  };
  for (int ks = 0; ks + 1 < KS; ks += 2) {
    kstep(ks);
    kstep(ks + 1);
  }
  if (KS & 1) kstep(KS - 1);
  if constexpr (CHK > 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) acc[i][j] = tot[i][j];
  }
  wait_vmem_le<0>();                         // nothing of the ring may still be landing when the workgroup ends
  // ---- epilogue: key classes, tile maxima, register images
  const int b = bh / NNJ_NHEAD;
  const uint8_t* cl = cls + (size_t)b * g.Cp + kb * 256 + kh * 128 + 4 * HH;
  float* Sb = S + (size_t)bh * g.s_bh;
  float* Mb = M + (size_t)bh * g.m_bh;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned is1 = 0, is2 = 0;               // bit r: key of accumulator register r is padded / beyond C
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const unsigned cw = *reinterpret_cast<const unsigned*>(cl + 32 * i + 8 * gq);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const unsigned c8 = (cw >> (8 * t)) & 0xffu;
        is1 |= (c8 == 1u ? 1u : 0u) << (4 * gq + t);
        is2 |= (c8 == 2u ? 1u : 0u) << (4 * gq + t);
      }
    }
    const int kt = kb * 8 + kh * 4 + i;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int qt = q0 / 32 + qh * NJ + j;
      float tmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float s = acc[i][j][r] * qs;           // align_scaling (axial_attention.py:31-33, 77) x log2(e), see k_qkv6
        s = ((is1 >> r) & 1u) ? fill : s;
        s = ((is2 >> r) & 1u) ? -INFINITY : s;
        acc[i][j][r] = s;
        tmax = fmaxf(tmax, s);
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const size_t t = (size_t)qt * g.nt32 + kt;
      if (HH == 0) Mb[t * 32 + l31] = tmax;
      float* dst = Sb + t * 1024 + lane * 4;   // image [4 register groups][64 lanes][4]: every store is 1 KiB contiguous
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
        *reinterpret_cast<f32x4*>(dst + 256 * gq) =
            (f32x4){acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]};
    }
  }
}

// ------------------------------------------------------------------ k_row_pv
// Context of 128 queries of one (b, h): 4 waves x 32 queries, O^T[e x query] in ET accumulators per wave.
// Per 16 keys: the V6 tile (ET*3 KiB) arrives by LDS-DMA (four-stage ring, one barrier per two k-steps), the
// lane's 8 probabilities exp(S - max) -- S read back as the register image k_row_s wrote, the next tile
// prefetched -- are split into a B fragment in registers, and ET A fragments stream from LDS through a
// two-deep software pipeline (hand-issued reads).  sum(P) is carried per lane; ctx = O / sum.
// More than 16 head tiles (more than 64 alignment rows): the head dimension is cut into g.nech chunks of ET tiles,
// one workgroup per (query block, chunk) -- each recomputes the probabilities (the exponentials are cheap next to
// the ET x 3 MFMAs per 16 keys) and owns its slice of the context rows; the chunks of a query block sit on
// consecutive workgroup slots of one XCD, so the score images they share are served by its L2.
template <int ET>
__global__ __launch_bounds__(256) void k_row_pv(const uint8_t* __restrict__ V6, const float* __restrict__ S,
                                                const float* __restrict__ M, float* __restrict__ ctx, Ra6 g,
                                                int nbh) {
  constexpr unsigned TILE = ET * NPL * 1024u;                         // bytes of one V6 tile
  constexpr unsigned STG = (TILE + 4095u) / 4096u * 4096u;            // stage size: whole KiB per wave
  constexpr int NIW = STG / 4096;                                     // DMA instructions per wave and tile
  constexpr int NST = 4 * STG <= 163840 ? 4 : 3;                      // ring stages (160 KiB of LDS)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, HH = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = g.Cp / 128, nech = g.nech;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / (nqb * nech)) * 8 + xcd;
  if (bh >= nbh) return;
  const int qb = (slot % (nqb * nech)) / nech, ech = slot % nech, qt = qb * 4 + wave;
  const int nk16 = g.nk16;
  const size_t tile_g = (size_t)g.ET * NPL * 1024u;                   // bytes of a whole V6 tile (all chunks) in HBM
  const size_t plane_g = (size_t)g.ET * 1024u;
  const uint8_t* Vt = V6 + (size_t)bh * g.v_bh + (size_t)ech * ET * 1024u;
  auto issue_piece = [&](auto pi, int k, int stage) {              // DMA piece i of NIW per wave and tile
    constexpr int i = decltype(pi)::value;
    const int kk = k < nk16 ? k : nk16 - 1;
    const unsigned I = (unsigned)(wave * NIW + i);                    // wave-uniform
    // piece I = (plane I / ET, head tile I % ET) of this chunk; beyond the tile: filler into the stage's pad
    const size_t so = I * 1024u < TILE ? (size_t)(I / ET) * plane_g + (size_t)(I % ET) * 1024u : 0u;
    lds_dma16(reinterpret_cast<const float*>(Vt + (size_t)kk * tile_g + so + lane * 16),
              reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(smem) + stage * STG + I * 1024u));
  };
  auto issue = [&](int k, int stage) {
    static_for<0, NIW>([&](auto pi) { issue_piece(pi, k, stage); });
  };
  const float* Sq = S + (size_t)bh * g.s_bh + (size_t)qt * g.nt32 * 1024 + lane * 4;
  const float* Mq = M + (size_t)bh * g.m_bh + (size_t)qt * g.nt32 * 32 + l31;
  float m = -INFINITY;
  for (int kt = 0; kt < g.nt32; ++kt) m = fmaxf(m, Mq[(size_t)kt * 32]);
  auto loadS = [&](f32x16& s, int kt) {
    const float* p = Sq + (size_t)kt * 1024;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(p + 256 * gq);
      s[4 * gq] = v[0]; s[4 * gq + 1] = v[1]; s[4 * gq + 2] = v[2]; s[4 * gq + 3] = v[3];
    }
  };
  f32x16 acc[ET];
#pragma unroll
  for (int t = 0; t < ET; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float lsum = 0.f;
  f32x16 s_cur, s_nxt;
  issue(0, 0);
  loadS(s_cur, 0);
  issue(1, 1);
  s_nxt = s_cur;
  const unsigned aA = lds_addr(smem) + (unsigned)l31 * 32u + 16u * (unsigned)(HH ^ ((l31 >> 3) & 1));
  Frag3 bfr[2];
  auto kstep = [&](int k, auto par) {
    constexpr int P = decltype(par)::value;    // k & 1
    if constexpr (NST == 4) {
      // Four-stage ring, ONE barrier per two k-steps (= per S image): at an even k the tiles k and k+1 (issued
      // during k-2 and k-1) and the S image prefetched at k-2 are waited for (vmcnt 0: they are everything
      // this wave has in flight), the barrier makes that true for every wave and tells that tiles k-2, k-1 are
      // read: their stages take tiles k+2 (issued during k) and k+3 (during k+1).
      if constexpr (P == 0) {
        wait_vmem_le<0>();
        barrier_nofence();
      }
    } else {
      // Three stages (the 16-tile head dimension): one barrier per k-step.  Vector memory operations retire in
      // order; younger than tile k's pieces are the pieces of tile k+1 and, on odd k, the 4 loads of the next S
      // image issued at the top of k-1.
      if constexpr (P == 1) wait_vmem_le<NIW + 4>(); else wait_vmem_le<NIW>();
      barrier_nofence();
    }
    if constexpr (P == 0) {
      if ((k >> 1) + 1 < g.nt32) loadS(s_nxt, (k >> 1) + 1);
    }
    const int st2 = (k + 2) % NST;              // its pieces go out between the MFMA groups below
    if constexpr (P == 0) {
      // probabilities of the 32 keys of this S image; registers 0-7 feed this k-step, 8-15 the next
      f32x16 p;
#pragma unroll
      for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s_cur[r] - m); lsum += p[r]; }   // logits in log2 units (k_row_s)
      split8<0>(bfr[0], p);
      split8<8>(bfr[1], p);
    }
    const Frag3& bf = bfr[P];
    const unsigned so = aA + (unsigned)(k % NST) * STG;
    // A fragments: three buffers, two tiles ahead (one group of 6 MFMAs = 192 cycles is less than an LDS
    // round trip under load)
    Frag3 a[3];
    auto rd = [&](auto ti) {
      constexpr int t = decltype(ti)::value;
      lds_read_frag<t * 1024>(a[t % 3].h, so);
      lds_read_frag<t * 1024 + ET * 1024>(a[t % 3].m, so);
    };
    rd(std::integral_constant<int, 0>{});
    if constexpr (ET > 1) rd(std::integral_constant<int, 1>{});
    static_for<0, ET>([&](auto ti) {
      constexpr int t = decltype(ti)::value;
      if constexpr (t + 1 < ET) lds_wait_le<NPL>(); else lds_wait_all();    // fragment t is in (t+1 may be landing)
      pin_frag(a[t % 3]);
      if constexpr (t + 2 < ET) rd(std::integral_constant<int, t + 2>{});
      acc[t] = mfma_b6(a[t % 3], bf, acc[t]);
      if constexpr (t < NIW) issue_piece(std::integral_constant<int, t>{}, k + 2, st2);
    });
    if constexpr (P == 1) s_cur = s_nxt;
  };
  for (int k = 0; k < nk16; k += 2) {          // nk16 is even (Cp is a multiple of 256)
    kstep(k, std::integral_constant<int, 0>{});
    kstep(k + 1, std::integral_constant<int, 1>{});
  }
  wait_vmem_le<0>();
  // ---- epilogue: ctx[b][h][q][e] = O / sum(P), e = 32t + 8g + 4HH + 0..3
  lsum += __shfl_xor(lsum, 32);
  const float inv = nnj_rcp(lsum);
  const int q = qt * 32 + l31;
  if (q < g.C) {
    float* dst = ctx + ((size_t)bh * g.C + q) * g.Epad;
    const int E = 8 * g.T;
#pragma unroll
    for (int t = 0; t < ET; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int e = 32 * (ech * ET + t) + 8 * gq + 4 * HH;
        if (e < E)
          *reinterpret_cast<f32x4*>(dst + e) = (f32x4){acc[t][4 * gq] * inv, acc[t][4 * gq + 1] * inv,
                                                       acc[t][4 * gq + 2] * inv, acc[t][4 * gq + 3] * inv};
      }
  }
}
