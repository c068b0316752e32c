// k_row_pv16.hpp -- the context product of the tied row attention on v_mfma_f32_16x16x32_f16 (NOT part of the shipped library).
//
// Round 5 found the encoder's matrix pipe POWER bound: on random operands the chip holds 1.63 GHz under the
// 32x32x16 instruction the shipped k_row_pv uses and 1.83-2.02 GHz under the 16x16x32 one.  This file is k_row_pv rewritten on
// the 16-row instruction (drop-in: same V6 tiles -- k_qkv6 must store them WITHOUT the half swap, `v_col = 16 * ((c16 >> 2) & 1)
// + 8 * (c16 >> 3)` --, same score images, same launch geometry).  Verified (GPU suite 123 passed) and measured: clock 1.65 ->
// 1.77 GHz, 34.2 -> 33.1 ms per rollout with the two-step score prefetch (0.7 of the 1.1 ms is the prefetch, which the shipped
// kernel has too).  NOT shipped because summing 32 keys per MFMA instead of 16 re-rolls the fp32 rounding of the encoder output,
// and with it the verification sample of rank seed 1005 moves from 7.7e-5 to 1.04e-4 of scale (seed 1004 from 9.5e-5 to 5.8e-5; the
// mean over the eight seeds falls 6.4e-5 -> 5.8e-5; the plain-fp32 oracle sits at 7.3e-5 on that sample) -- bench.py's 1e-4 exit
// gate is per rank, and 0.4 % of throughput does not buy a red 8-GPU run (profiles/r05/ab_pv16.txt, e64_scan_pv16.txt).
// NNJ_PV_ORDER selects the order of the three piece products (all three orders put seed 1005 at 0.96-1.11e-4).
// ------------------------------------------------------------------ k_row_pv
// Context of 128 queries of one (b, h): 4 waves x 32 queries, O^T[e x query] on v_mfma_f32_16x16x32_f16 (round 5: the 32x32x16 form
// of rounds 1-4 is in the git history; same operands, products and bytes, but on random operands the chip holds 1.9 GHz under the
// 16-row instruction and 1.63 GHz under the 32-row one -- tools/mfma_power.hip, profiles/r05/mfma_power.txt, clock_b256.txt: the matrix
// pipe of this kernel is POWER bound).  A wave owns 32 queries x ET x 32 context rows; per 32 KEYS (one score
// image of k_row_s, two V6 tiles, one workgroup barrier) it multiplies 2 ET A fragments [16 e x 32 keys] with two B fragments
// [32 keys x 16 queries]:
//   lane = (x = lane & 15, kg = lane >> 4).  A: row e = 16 t + x, the lane's 8 k-slots = the 16-byte half (kg & 1) of V6 tile
//   2 K + (kg >> 1), i.e. keys 16 (kg >> 1) + {4 (kg & 1) + 0..3, 8 + 4 (kg & 1) + 0..3} of the 32 (the order k_qkv6 stores);
//   B (query tile j): query 16 j + x, the same eight keys = granules (gq = 2 (kg >> 1) + u, HH = kg & 1), u = 0, 1, of the score
//   image [4 gq][64 lanes q + 32 HH][4] -- 16 bytes each, read straight from the image k_row_s wrote (its layout is unchanged).
//   C/D: O^T tile (t, j): lane holds rows e = 16 t + 4 kg + 0..3 of query 16 j + x.
// V6 rows are stored WITHOUT the half swap of the Q6 / K6 tiles: a ds_read_b128 group of 16 lanes then covers
// rows r and r + 8 at different halves for kg 0 / 1 -- sixteen distinct 16-byte slots (with the swap they collide in pairs).
// Per 32 keys 12 ET MFMAs of 16 cycles; the V6 tiles (ET x 2 KiB each: two fp16 planes) arrive by LDS-DMA into a four-stage ring, one
// workgroup barrier per 32 keys; sum(P) is carried per lane, ctx = O / sum.  More than 16 head tiles (more than 64 alignment rows, only
// with NNJ_ENC64=0): the head dimension is cut into g.nech chunks of ET tiles, one workgroup per (query block, chunk) -- each recomputes
// the probabilities and owns its slice of the context rows; the chunks of a query block sit on consecutive workgroup slots of one
// XCD, so the score images they share are served by its L2.
template <int ET>
__global__ __launch_bounds__(256) void k_row_pv(const uint8_t* __restrict__ V6, const float* __restrict__ S,
                                                  const float* __restrict__ M, float* __restrict__ ctx, Ra6 g,
                                                  int nbh) {
  constexpr unsigned TILE = ET * NPL * 1024u;                         // bytes of one V6 tile
  constexpr unsigned STG = (TILE + 4095u) / 4096u * 4096u;            // stage size: whole KiB per wave
  constexpr int NIW = STG / 4096;                                     // DMA instructions per wave and tile
  constexpr int NST = 4;                                              // two tiles in use, two in flight
  constexpr int NT16 = 2 * ET;                                        // 16-row tiles of the chunk's context rows
  static_assert(4 * STG <= 163840, "four stages must fit the LDS");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, kg = lane >> 4;
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
    const size_t so = I * 1024u < TILE ? (size_t)(I / ET) * plane_g + (size_t)(I % ET) * 1024u : 0u;
    lds_dma16(reinterpret_cast<const float*>(Vt + (size_t)kk * tile_g + so + lane * 16),
              reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(smem) + stage * STG + I * 1024u));
  };
  auto issue = [&](int k, int stage) {
    static_for<0, NIW>([&](auto pi) { issue_piece(pi, k, stage); });
  };
  // score image of (query tile qt, key tile kt): this lane's four granules (j, u)
  const float* Sq = S + (size_t)bh * g.s_bh + (size_t)qt * g.nt32 * 1024 + ((2 * (kg >> 1)) * 64 + x + 32 * (kg & 1)) * 4;
  const float* Mq = M + (size_t)bh * g.m_bh + (size_t)qt * g.nt32 * 32 + x;
  float m0 = -INFINITY, m1 = -INFINITY;
  for (int kt = 0; kt < g.nt32; ++kt) { m0 = fmaxf(m0, Mq[(size_t)kt * 32]); m1 = fmaxf(m1, Mq[(size_t)kt * 32 + 16]); }
  auto loadS = [&](f32x16& s, int kt) {                              // s[8 j + 4 u + i]
    const float* p = Sq + (size_t)kt * 1024;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + (u * 64 + 16 * j) * 4);
        s[8 * j + 4 * u] = v[0]; s[8 * j + 4 * u + 1] = v[1]; s[8 * j + 4 * u + 2] = v[2]; s[8 * j + 4 * u + 3] = v[3];
      }
  };
  f32x4 acc[NT16][2];
#pragma unroll
  for (int t = 0; t < NT16; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[t][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float lsum0 = 0.f, lsum1 = 0.f;
  // Score images TWO steps ahead (k_row_pv: one): the image of step K+2 is requested at the END of step K into the registers of
  // the image step K has consumed at its top -- two register sets, rotated by the parity of K, no copies.  Vector memory
  // retires in order, so the wait at the top of a step is vmcnt(4): everything but the four loads of the image requested last
  // (tiles and image of THIS step are older).  Timing-only knock-outs of the 32-row kernel had put the exposed latency of the
  // one-step prefetch at up to 5 ms per rollout (profiles/r05/ko_rows.txt).
  f32x16 sbuf[2];
  issue(0, 0);
  loadS(sbuf[0], 0);
  issue(1, 1);
  loadS(sbuf[1], g.nt32 > 1 ? 1 : 0);
  // A fragments: row 16 t + x of the tile in stage (k + (kg >> 1)) % 4, half kg & 1 (no swizzle)
  const unsigned aA = lds_addr(smem) + (unsigned)(kg >> 1) * STG + (unsigned)x * 32u + 16u * (unsigned)(kg & 1);
  auto kstep = [&](int K, auto par) {                                // keys 32 K .. 32 K + 31 = V6 tiles k = 2 K, 2 K + 1
    constexpr int P = decltype(par)::value;                          // K & 1: the register set of this step's image
    const int k = 2 * K;
    f32x16& s_cur = sbuf[P];
    // tiles k and k+1 (issued during the previous step) and this step's score image (requested a step before them) have
    // landed once only the four youngest loads are outstanding; the barrier makes that true for every wave and tells that
    // tiles k-2, k-1 are read: their stages take k+2, k+3
    wait_vmem_le<4>();
    barrier_nofence();
    Frag3 bfr[2];
    {
      f32x16 p;
#pragma unroll
      for (int r = 0; r < 8; ++r) { p[r] = __builtin_amdgcn_exp2f(s_cur[r] - m0); lsum0 += p[r]; }   // logits in log2 units (k_row_s)
#pragma unroll
      for (int r = 8; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s_cur[r] - m1); lsum1 += p[r]; }
      split8<0>(bfr[0], p);
      split8<8>(bfr[1], p);
    }
    const unsigned so = aA + (unsigned)(k % NST) * STG;              // k % 4 is 0 or 2: tile k+1 sits one stage up
    const int st2 = (k + 2) % NST;                                   // stages of tiles k+2, k+3: st2, st2 + 1
    Frag3 a[3];
    auto rd = [&](auto ti) {
      constexpr int t = decltype(ti)::value;
      lds_read_frag<t * 512>(a[t % 3].h, so);
      lds_read_frag<t * 512 + ET * 1024>(a[t % 3].m, so);
    };
    rd(std::integral_constant<int, 0>{});
    rd(std::integral_constant<int, 1>{});
    static_for<0, NT16>([&](auto ti) {
      constexpr int t = decltype(ti)::value;
      if constexpr (t + 1 < NT16) lds_wait_le<NPL>(); else lds_wait_all();    // fragment t is in (t+1 may be landing)
      pin_frag(a[t % 3]);
      if constexpr (t + 2 < NT16) rd(std::integral_constant<int, t + 2>{});
#ifndef NNJ_PV_ORDER
#define NNJ_PV_ORDER 0
#endif
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const Frag3& A_ = a[t % 3];
        if constexpr (NNJ_PV_ORDER == 0) acc[t][j] = mfma16_b6(A_, bfr[j], acc[t][j]);     // m.h, h.m, h.h
        else if constexpr (NNJ_PV_ORDER == 1) {
          acc[t][j] = mfma16_f16(A_.h, bfr[j].h, acc[t][j]); acc[t][j] = mfma16_f16(A_.h, bfr[j].m, acc[t][j]); acc[t][j] = mfma16_f16(A_.m, bfr[j].h, acc[t][j]);
        } else {
          acc[t][j] = mfma16_f16(A_.h, bfr[j].m, acc[t][j]); acc[t][j] = mfma16_f16(A_.m, bfr[j].h, acc[t][j]); acc[t][j] = mfma16_f16(A_.h, bfr[j].h, acc[t][j]);
        }
      }
      // the 2 NIW pieces of tiles k+2, k+3 behind the first MFMA groups (NT16 >= 2 NIW for every ET).  (All of them in front
      // of the exponentials instead, where an LDS-DMA instruction is cheaper to issue: 33.3 -> 35.0 ms -- the waves then reach
      // their MFMAs later than their SIMD partner needs the pipe free: profiles/r05/ab_pv16.txt)
      if constexpr (t < NIW) issue_piece(std::integral_constant<int, t>{}, k + 2, st2);
      else if constexpr (t < 2 * NIW) issue_piece(std::integral_constant<int, t - NIW>{}, k + 3, st2 + 1);
    });
    loadS(s_cur, K + 2 < g.nt32 ? K + 2 : g.nt32 - 1);               // (past the end: a harmless reload keeps the count)
  };
  for (int K = 0; K + 1 < g.nt32; K += 2) {     // nk16 = 2 nt32 (Cp is a multiple of 256)
    kstep(K, std::integral_constant<int, 0>{});
    kstep(K + 1, std::integral_constant<int, 1>{});
  }
  if (g.nt32 & 1) kstep(g.nt32 - 1, std::integral_constant<int, 0>{});
  wait_vmem_le<0>();
  // ---- epilogue: ctx[b][h][q][e] = O / sum(P), e = 16 t + 4 kg + 0..3; the sums of a query sit on the four lanes x + 16 kg
  lsum0 += __shfl_xor(lsum0, 16); lsum0 += __shfl_xor(lsum0, 32);
  lsum1 += __shfl_xor(lsum1, 16); lsum1 += __shfl_xor(lsum1, 32);
  const float inv[2] = {nnj_rcp(lsum0), nnj_rcp(lsum1)};
  const int E = 8 * g.T;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = qt * 32 + 16 * j + x;
    if (q < g.C) {
      float* dst = ctx + ((size_t)bh * g.C + q) * g.Epad;
#pragma unroll
      for (int t = 0; t < NT16; ++t) {
        const int e = 32 * (ech * ET) + 16 * t + 4 * kg;
        if (e < E) *reinterpret_cast<f32x4*>(dst + e) = acc[t][j] * inv[j];
      }
    }
  }
}
