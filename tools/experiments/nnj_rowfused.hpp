// nnj_rowfused.hpp -- tied row attention of up to 64 alignment rows as ONE kernel per layer (round 4).
//
// k_row_s + k_row_pv send the scores of a layer through HBM (8.6 GB written, 12 GB fetched per layer at B = 256,
// C = 1024) because a flash-style kernel cannot keep a query block stationary next to its context accumulators at TWO
// waves per SIMD (nnj_rowattn.hpp).  At ONE wave per SIMD -- where k_row_pv has been running all along -- the 512-register
// file holds both: a wave owns 32 queries, S^T [256 keys x 32 queries] = 128 accumulators and O^T [E x 32 queries]
// = 16 ET accumulators (208 at 50 rows).  Per 256-key block:
//   S phase   the contraction over the KS operand tiles of the head dimension (k_row_s's loop with a 256 x 32 wave tile:
//             the K tile of a k-step is shared by the four waves, each reads its own 32 rows of the Q tile);
//   softmax   scale, key classes, running maximum per query (online softmax: O and the sum are rescaled when the
//             maximum grows), P = exp2(S - max) in place;
//   P.V phase k_row_pv's loop over the sixteen 16-key V6 tiles of the block, P registers as the B operand as they stand.
// Operand tiles arrive by LDS-DMA into two three-stage rings (K|Q tiles: 24 KiB stages; V6 tiles: ET x 2 KiB); the
// first tiles of the next phase are issued during the last two steps of the running one, so the DMA never drains.
// The Q tile of the block is re-streamed once per key block (from L2); no score ever leaves the registers.
#pragma once
#include "nnj_rowattn.hpp"

// NKT = 32-key tiles of an inner key block: 8 (the whole 256-row block of the K6 layout: 128 S accumulators) or 4 (its
// halves: 64 -- with the 208 context accumulators of 50 rows the eight-tile form spills).
template <int ET, int NKT = 4>
__global__ __launch_bounds__(256) void k_row_fused(const uint8_t* __restrict__ Q6, const uint8_t* __restrict__ K6,
                                                   const uint8_t* __restrict__ V6, const uint8_t* __restrict__ cls,
                                                   const uint8_t* __restrict__ cls_blk, float* __restrict__ ctx, Ra6 g,
                                                   int nbh, float fill, float qs) {
  static_assert(NKT == 4 || NKT == 8, "inner key block: 128 or 256 keys");
  constexpr int NIB = 8 / NKT;                                        // inner blocks per 256-row block of K6
  constexpr int PVS = 2 * NKT;                                        // P.V steps (16 keys each) of an inner block
  constexpr unsigned KPL = NKT * 1024u;                               // bytes of one plane of the K part of a stage
  constexpr unsigned KTB = NPL * KPL;                                 // bytes of the K part (32 NKT rows x 16 k, NPL planes)
  constexpr unsigned QPL = 128u * 32u;                                // bytes of one plane of the Q tile (128 queries)
  constexpr unsigned SSTG = KTB + NPL * QPL;                          // stage of the S ring
  constexpr int NPS = (NPL * NKT + NPL * 4) / 4;                      // DMA pieces per wave and S step
  constexpr unsigned VTILE = ET * NPL * 1024u;
  constexpr unsigned VSTG = (VTILE + 4095u) / 4096u * 4096u;
  constexpr int NPV = VSTG / 4096;                                    // DMA pieces per wave and P.V step
  constexpr int NSTV = 3 * SSTG + 3 * VSTG <= 163840u ? 3 : 2;       // V ring stages (160 KiB of LDS)
  static_assert(NSTV == 3, "two V stages need another prefetch distance: not instantiated");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  uint8_t* lds = reinterpret_cast<uint8_t*>(smem);
  const unsigned sring = lds_addr(smem), vring = sring + 3 * SSTG;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, HH = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nqb = g.Cp / 128;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;             // all blocks of one (b,h) on one XCD (one L2)
  const int bh = (slot / nqb) * 8 + xcd;
  if (bh >= nbh) return;
  const int qb = slot % nqb, q0 = qb * 128;
  const int KS = g.KS, nrb = g.nrb;
  const uint8_t* Kbh = K6 + (size_t)bh * g.qk_bh;
  const uint8_t* Qt = Q6 + (size_t)bh * g.qk_bh + (size_t)(q0 >> 8) * NPL * 8192 + (size_t)(q0 & 255) * 32;
  const size_t ks_stride = (size_t)nrb * NPL * 8192;
  const uint8_t* Vbh = V6 + (size_t)bh * g.v_bh;
  const size_t vtile_g = (size_t)g.ET * NPL * 1024u, vplane_g = (size_t)g.ET * 1024u;
  const int nib = nrb * NIB;                                          // inner key blocks
  const int SG = nib * KS, VG = nib * PVS;                            // S steps / P.V steps of the whole block row
  // S step sg = ib * KS + ks -> ring stage sg % 3; pieces I = wave * NPS + i: NPL NKT of the K rows, 8 of the Q tile
  auto issue_s = [&](auto pi, int sg) {
    constexpr int i = decltype(pi)::value;
    const int sgc = sg < SG ? sg : SG - 1;                            // past the end: harmless reload into a free stage
    const int ib = sgc / KS, ks = sgc - ib * KS;
    const int kb = ib / NIB, hb = ib % NIB;
    const int I = wave * NPS + i;
    const uint8_t* src;
    if (I < NPL * NKT) {
      const int pl = I / NKT, pc = I % NKT;                           // plane, 1-KiB piece (32 rows) of the inner block's rows
      src = Kbh + (size_t)kb * NPL * 8192 + (size_t)pl * 8192 + (size_t)(hb * NKT + pc) * 1024;
    } else {
      const int J = I - NPL * NKT, pl = J / 4, pc = J % 4;
      src = Qt + (size_t)pl * 8192 + pc * 1024;
    }
    lds_dma16(reinterpret_cast<const float*>(src + (size_t)ks * ks_stride + lane * 16),
              reinterpret_cast<float*>(lds + (unsigned)(sg % 3) * SSTG + I * 1024));
  };
  // P.V step vg = ib * PVS + k16 (= the V6 tile index) -> ring stage vg % 3 of the V ring
  auto issue_v = [&](auto pi, int vg) {
    constexpr int i = decltype(pi)::value;
    const int vgc = vg < VG ? vg : VG - 1;
    const unsigned I = (unsigned)(wave * NPV + i);
    const size_t so = I * 1024u < VTILE ? (size_t)(I / ET) * vplane_g + (size_t)(I % ET) * 1024u : 0u;
    lds_dma16(reinterpret_cast<const float*>(Vbh + (size_t)vgc * vtile_g + so + lane * 16),
              reinterpret_cast<float*>(lds + 3 * SSTG + (unsigned)(vg % 3) * VSTG + I * 1024u));
  };
  const unsigned half = 16u * (unsigned)(HH ^ ((l31 >> 3) & 1));
  const unsigned aK = sring + (unsigned)l31 * 32u + half;
  const unsigned aQ = sring + KTB + (unsigned)(wave * 32 + l31) * 32u + half;
  const unsigned aV = vring + (unsigned)l31 * 32u + half;
  f32x16 O[ET];
#pragma unroll
  for (int t = 0; t < ET; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[t][r] = 0.f;
  float m = -INFINITY, lsum = 0.f;
  const int b = bh / NNJ_NHEAD;
  unsigned blkmask = 0;                                               // bit kb: block kb holds padded keys / keys beyond C
  for (int kb = 0; kb < nrb; ++kb) blkmask |= (cls_blk[(size_t)b * nrb + kb] ? 1u : 0u) << (kb & 31);
  if (nrb > 32) blkmask = 0xffffffffu;                                // (more than 8192 columns: every block takes the class path)
  static_for<0, NPS>([&](auto pi) { issue_s(pi, 0); });
  static_for<0, NPS>([&](auto pi) { issue_s(pi, 1); });
  for (int ib = 0; ib < nib; ++ib) {
    const int kb = ib / NIB, hb = ib % NIB;
    f32x16 S[NKT];
#pragma unroll
    for (int i = 0; i < NKT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[i][r] = 0.f;
    // ================================================================= S phase
    for (int ks = 0; ks < KS; ++ks) {
      const int sg = ib * KS + ks;
      const unsigned so = (unsigned)(sg % 3) * SSTG;
      // tile sg has landed; younger: the pieces of the step issued one step ago (an S tile, or the block's first V tile)
      if (ks == KS - 1) wait_vmem_le<NPV>(); else wait_vmem_le<NPS>();
      barrier_nofence();
      Frag3 A[NKT], Bq;
      static_for<0, 4>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        lds_read_frag<i * 1024>(A[i].h, aK + so);
        lds_read_frag<i * 1024 + KPL>(A[i].m, aK + so);
      });
      lds_read_frag<0>(Bq.h, aQ + so);
      lds_read_frag<QPL>(Bq.m, aQ + so);
      lds_wait_all();
#pragma unroll
      for (int i = 0; i < 4; ++i) pin_frag(A[i]);
      pin_frag(Bq);
      static_for<4, NKT>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        lds_read_frag<i * 1024>(A[i].h, aK + so);
        lds_read_frag<i * 1024 + KPL>(A[i].m, aK + so);
      });
      // the tile two steps ahead goes into the stage of tile sg - 1: behind the first MFMA groups.  The last two steps of
      // the phase issue the block's first two V6 tiles instead (the S tiles of the next block come from the P.V phase).
      static_for<0, 4>([&](auto ii) {
        constexpr int i = decltype(ii)::value;
        S[i] = mfma_b6(A[i], Bq, S[i]);
        if (ks < KS - 2) {
          static_for<0, 2>([&](auto ee) {
            constexpr int piece = 2 * i + decltype(ee)::value;
            if constexpr (piece < NPS) issue_s(std::integral_constant<int, piece>{}, sg + 2);
          });
        } else {
          static_for<0, 2>([&](auto ee) {
            constexpr int piece = 2 * i + decltype(ee)::value;
            if constexpr (piece < NPV) issue_v(std::integral_constant<int, piece>{}, ib * PVS + (ks - (KS - 2)));
          });
        }
      });
      if constexpr (NKT > 4) {
        lds_wait_all();
#pragma unroll
        for (int i = 4; i < NKT; ++i) pin_frag(A[i]);
        static_for<4, NKT>([&](auto ii) {
          constexpr int i = decltype(ii)::value;
          S[i] = mfma_b6(A[i], Bq, S[i]);
        });
      }
    }
    // ================================================================= softmax of the block (online)
    {
      float tmax = -INFINITY;
      if (((blkmask >> (kb & 31)) & 1u) == 0) {                        // (wave uniform)
#pragma unroll
        for (int i = 0; i < NKT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) { S[i][r] *= qs; tmax = fmaxf(tmax, S[i][r]); }
      } else {
        // padded keys (-> fill) or keys beyond the alignment (-> -inf) in this block: the class bytes of the lane's keys.
        // (rare path: these loads drain the DMA ring once)
        const uint8_t* cl = cls + (size_t)b * g.Cp + kb * 256 + hb * (32 * NKT) + 4 * HH;
#pragma unroll
        for (int i = 0; i < NKT; ++i) {
          unsigned is1 = 0, is2 = 0;
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
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float s = S[i][r] * qs;
            s = ((is1 >> r) & 1u) ? fill : s;
            s = ((is2 >> r) & 1u) ? -INFINITY : s;
            S[i][r] = s;
            tmax = fmaxf(tmax, s);
          }
        }
      }
      tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
      const float mn = fmaxf(m, tmax);                                 // finite: every block holds a key of the alignment
      if (ib > 0 && __any(mn > m)) {                                   // the maximum of some query grew: rescale what is summed
        const float al = __builtin_amdgcn_exp2f(m - mn);
        lsum *= al;
#pragma unroll
        for (int t = 0; t < ET; ++t) O[t] *= al;
      }
      m = mn;
#pragma unroll
      for (int i = 0; i < NKT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(S[i][r] - m);         // logits in log2 units (qs carries log2 e)
          S[i][r] = p;
          lsum += p;
        }
    }
    // ================================================================= P.V phase
    // A run-time loop over the key tiles of the block (two 16-key steps each): the tile in use is always S[0], the others
    // move up one place after it (48 moves per 78 MFMAs) -- fully unrolled, the sixteen steps' address arithmetic was hoisted
    // and the kernel spilled.
    for (int j2 = 0; j2 < NKT; ++j2) {
      static_for<0, 2>([&](auto hh2) {
        constexpr int G = decltype(hh2)::value;                           // registers 8G .. 8G+7 of the tile
        const int k16 = 2 * j2 + G;
        const int vg = ib * PVS + k16;
        const unsigned so = (unsigned)(vg % 3) * VSTG;
        // tile vg has landed; younger: the pieces issued one step ago (a V tile, or the next block's first S tile)
        if (k16 == PVS - 1) wait_vmem_le<NPS>(); else wait_vmem_le<NPV>();
        barrier_nofence();
        Frag3 pf;
        split8<8 * G>(pf, S[0]);
        Frag3 a[3];
        auto rd = [&](auto ti) {
          constexpr int t = decltype(ti)::value;
          lds_read_frag<t * 1024>(a[t % 3].h, aV + so);
          lds_read_frag<t * 1024 + ET * 1024>(a[t % 3].m, aV + so);
        };
        rd(std::integral_constant<int, 0>{});
        if constexpr (ET > 1) rd(std::integral_constant<int, 1>{});
        const bool tail = k16 >= PVS - 2;                                 // the last two steps issue the next block's S tiles
        static_for<0, ET>([&](auto ti) {
          constexpr int t = decltype(ti)::value;
          if constexpr (t + 1 < ET) lds_wait_le<NPL>(); else lds_wait_all();
          pin_frag(a[t % 3]);
          if constexpr (t + 2 < ET) rd(std::integral_constant<int, t + 2>{});
          O[t] = mfma_b6(a[t % 3], pf, O[t]);
          if constexpr (t < NPV || t < NPS) {
            if (!tail) {
              if constexpr (t < NPV) issue_v(std::integral_constant<int, t>{}, vg + 2);
            } else {
              if constexpr (t < NPS) issue_s(std::integral_constant<int, t>{}, (ib + 1) * KS + (k16 - (PVS - 2)));
            }
          }
        });
      });
#pragma unroll
      for (int i = 0; i + 1 < NKT; ++i) S[i] = S[i + 1];
    }
  }
  wait_vmem_le<0>();                         // nothing of the rings may still be landing when the workgroup ends
  // ---- epilogue: ctx[b][h][q][e] = O / sum(P), e = 32t + 8g + 4HH + 0..3
  lsum += __shfl_xor(lsum, 32);
  const float inv = nnj_rcp(lsum);
  const int q = q0 + wave * 32 + l31;
  if (q < g.C) {
    float* dst = ctx + ((size_t)bh * g.C + q) * g.Epad;
    const int E = 8 * g.T;
#pragma unroll
    for (int t = 0; t < ET; ++t)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int e = 32 * t + 8 * gq + 4 * HH;
        if (e < E)
          *reinterpret_cast<f32x4*>(dst + e) = (f32x4){O[t][4 * gq] * inv, O[t][4 * gq + 1] * inv,
                                                       O[t][4 * gq + 2] * inv, O[t][4 * gq + 3] * inv};
      }
  }
}

// cls_blk[b][kb] = OR of the key classes of the 256 keys of block kb (0: every key is a plain key of the alignment)
__global__ void k_key_class_blocks(const uint8_t* __restrict__ cls, uint8_t* __restrict__ cls_blk, int total, int Cp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;      // i = b * nrb + kb
  if (i >= total) return;
  const int nrb = Cp / 256, b = i / nrb, kb = i % nrb;
  const uint8_t* c = cls + (size_t)b * Cp + kb * 256;
  unsigned v = 0;
  for (int k = 0; k < 256; k += 4) v |= *reinterpret_cast<const unsigned*>(c + k);
  cls_blk[i] = (uint8_t)((v | (v >> 8) | (v >> 16) | (v >> 24)) & 0xffu);
}
