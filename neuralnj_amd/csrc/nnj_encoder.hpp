// nnj_encoder.hpp -- gfx950 kernels of the axial MSA encoder
// (restates reference model.py:67-88, msa_modules.py:62-151, axial_attention.py:6-255).
//
// Tiling: one wave owns one alignment column (b, c) and all R rows of it ("column
// wave", tokens = rows, NT = ceil(R/32) tiles of 32 tokens on the lanes); a workgroup
// is 4 waves = 4 adjacent columns, so every row contributes 1 KiB contiguous bytes.
// Token-local stages chain in registers (nnj_common.hpp); weights sit in LDS.
//
//   k_embed_qkv : embed (6-entry LUT of the site codes) -> x ; LN -> q,k,v (row attn, layer 0)
//   k_row_attn  : tied row attention, flash style, head dim E = R*8, per (b, h, 64 queries)
//   k_tok1      : ctx -> out_proj -> +x ; LN -> q,k,v -> column attention -> out_proj -> +x
//   k_ffn/k_qkv : LN -> fc1 -> GELU -> fc2 -> +x ; LN -> q,k,v of the next layer's row attn (persistent,
//                 flat token tiling)
//
// HBM layouts: x [B,R,C,64]; Q/K/V/ctx head-major [B,8,C,Epad] with e = r*8 + d,
// Epad = roundup(R*8,16) (zero padded) -- a (b,h) slice is a plain [C x Epad] matrix.
#pragma once
#include "nnj_common.hpp"

struct AttnW {            // pointers into the packed device weights (row-major [out][in])
  const float *Wk, *bk, *Wv, *bv, *Wq, *bq, *Wo, *bo, *ln_w, *ln_b;
};
struct FfnW {
  const float *W1, *b1, *W2, *b2, *ln_w, *ln_b;
};
struct EmbedW {           // embed = Linear(4 -> 64), GELU, Linear(64 -> 64)  (reference model.py:39-43)
  const float *E0, *e0, *E2, *e2;
};

// ------------------------------------------------------------------ QKV epilogue
// Writes one of q/k/v (feature-major registers of the wave's column) to the
// head-major buffer.  Rows r >= R inside the zero-padded Epad range are written as 0.
template <int NT>
__device__ __forceinline__ void store_headmajor(const f32x16 (&v)[NT][2], float* dst, int b, int c, int C,
                                                int R, int Epad, float scale, int lane, int r0 = 0) {
  const int tok = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = r0 + 32 * nt + tok;
    if (r * 8 >= Epad) continue;
    const bool live = r < R;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int h = 4 * mt + g;
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (live) {
          o[0] = v[nt][mt][4 * g + 0] * scale; o[1] = v[nt][mt][4 * g + 1] * scale;
          o[2] = v[nt][mt][4 * g + 2] * scale; o[3] = v[nt][mt][4 * g + 3] * scale;
        }
        *reinterpret_cast<f32x4*>(dst + (((size_t)b * NNJ_NHEAD + h) * C + c) * Epad + r * 8 + 4 * hh) = o;
      }
  }
}

// LN + q,k,v projections of the row attention, written head-major.
// q is pre-multiplied by head_dim^-0.5 / sqrt(R) and zeroed at padded columns
// (reference axial_attention.py:31-33,77-82).
template <int NT>
__device__ __forceinline__ void row_qkv_stage(const f32x16 (&x)[NT][2], const AttnW& w, const float* Wq_l,
                                              const float* Wk_l, const float* Wv_l, float* Q, float* K,
                                              float* V, int b, int c, int C, int R, int Epad, bool padded,
                                              int lane, int r0 = 0) {
  const int hh = lane >> 5;
  f32x16 y[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) layer_norm64(y[nt], x[nt], w.ln_w, w.ln_b, hh);
  const float qscale = padded ? 0.0f : (rsqrtf((float)NNJ_DH) / sqrtf((float)R));
  f32x16 o[NT][2];
  linear_T<2, 2, NT>(o, y, Wq_l, w.bq, lane);
  store_headmajor<NT>(o, Q, b, c, C, R, Epad, qscale, lane, r0);
  linear_T<2, 2, NT>(o, y, Wk_l, w.bk, lane);
  store_headmajor<NT>(o, K, b, c, C, R, Epad, 1.0f, lane, r0);
  linear_T<2, 2, NT>(o, y, Wv_l, w.bv, lane);
  store_headmajor<NT>(o, V, b, c, C, R, Epad, 1.0f, lane, r0);
}

// ------------------------------------------------------------------ k_embed_qkv
// codes uint8 [B,R,L] (patch_size 1: C == L); lut [6][64] = embed MLP of the six site
// vectors (reference model.py:39-43,76-77 evaluated on phydata.py:38-46's vectors).
template <int NT>
__global__ __launch_bounds__(256) void k_embed_qkv(const uint8_t* __restrict__ codes,
                                                   const float* __restrict__ onehot, EmbedW ew,
                                                   const float* __restrict__ lut,
                                                   const uint8_t* __restrict__ mask, float* __restrict__ x,
                                                   float* __restrict__ Q, float* __restrict__ K,
                                                   float* __restrict__ V, AttnW w, int B, int R, int C,
                                                   int Epad, int do_qkv) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wq_l = smem;
  float* Wk_l = smem + 4096;
  float* Wv_l = smem + 8192;
  float* E2_l = smem + 12288;     // only staged for the general float input
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (onehot) stage_weight<64>(E2_l, ew.E2, 64, tid, 256);
  if (do_qkv) {
    stage_weight<64>(Wq_l, w.Wq, 64, tid, 256);
    stage_weight<64>(Wk_l, w.Wk, 64, tid, 256);
    stage_weight<64>(Wv_l, w.Wv, 64, tid, 256);
  }
  __syncthreads();
  const long col = (long)blockIdx.x * 4 + wave;   // (b, c) flattened
  if (col >= (long)B * C) return;
  const int b = (int)(col / C), c = (int)(col % C);
  const int tok = lane & 31, hh = lane >> 5;
  f32x16 xr[NT][2];
  if (onehot) {
    // general float input [B,R,L,4]: the embed MLP itself (first Linear on the VALU, second by MFMA)
    f32x16 t1[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = 32 * nt + tok;
      const bool valid = r < R;
      const f32x4 oh = *reinterpret_cast<const f32x4*>(onehot + (((size_t)b * R + (valid ? r : 0)) * C + c) * 4);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(ew.e0 + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(ew.E0 + (32 * mt + 8 * g + 4 * hh + t) * 4);
            const float s = b4[t] + wv[0] * oh[0] + wv[1] * oh[1] + wv[2] * oh[2] + wv[3] * oh[3];
            t1[nt][mt][4 * g + t] = valid ? gelu_erf(s) : 0.f;
          }
        }
    }
    linear_T<2, 2, NT>(xr, t1, E2_l, ew.e2, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = 32 * nt + tok;
      if (!(r < R)) { xr[nt][0] = (f32x16)(0.f); xr[nt][1] = (f32x16)(0.f); }
      store_token64(xr[nt], x + (((size_t)b * R + r) * C + c) * 64, r < R, hh);
    }
  } else {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = 32 * nt + tok;
      const bool valid = r < R;
      int code = codes[((size_t)b * R + (valid ? r : 0)) * C + c];
      if (!valid || code > 5) code = 5;
      load_token64(xr[nt], lut + code * 64, valid, hh);
      store_token64(xr[nt], x + (((size_t)b * R + r) * C + c) * 64, valid, hh);
    }
  }
  if (do_qkv) {
    const bool padded = mask && mask[(size_t)b * C + c];
    row_qkv_stage<NT>(xr, w, Wq_l, Wk_l, Wv_l, Q, K, V, b, c, C, R, Epad, padded, lane);
  }
}

// ------------------------------------------------------------------ k_row_attn
// Tied row attention for one (b, h) and 64 query columns: standard attention over the
// C columns with head dimension Epad = 16*NTE (reference axial_attention.py:97-114), online
// softmax, fp32 MFMA 16x16x4.  Computes S^T = K Q^T so that the probabilities come out
// in the A-operand layout of the P*V product.
//   * K/V tiles of 16 keys stream HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4) into a
//     two-stage ring: the DMA of tile t+1 is in flight while tile t is multiplied; one
//     workgroup barrier per tile.
//   * LDS rows are padded to LD = roundup(Epad-4,32)+4 floats (LD % 32 == 4 makes the float2
//     K reads and the scalar V reads bank-conflict free); DMA lanes that land in the pad read a
//     harmless address.
//   * MFMA operands are read from LDS by hand-issued ds_reads in batches, software pipelined:
//     the next batch is issued before the current batch's MFMAs and waited for after them.
//   * the key-padding mask of the batch element sits in LDS (no global load inside the loop).
#define RA_KEYS 16        // keys per tile
template <int NTE>
struct RaShape {
  static constexpr int Epad = 16 * NTE;
  static constexpr int LD = (Epad - 4 + 31) / 32 * 32 + 4;
  static constexpr int CPR = LD / 4, CPE = Epad / 4;              // 16-byte chunks per LDS row / global row
  static constexpr int KI = (RA_KEYS * CPR + 63) / 64;            // 1-KiB DMA instructions per K (or V) tile
  static constexpr int NDMA = (2 * KI + 3) / 4;                   // LDS-DMA instructions per wave per tile
  static constexpr int STAGE_F = NDMA * 4 * 256;                  // floats per ring stage: [K: KI KiB | V: KI KiB]
  static constexpr int NS = 2 * NTE;                              // float2 k-steps of S^T
};

template <int NTE>
__global__ __launch_bounds__(256) void k_row_attn(const float* __restrict__ Q, const float* __restrict__ K,
                                                  const float* __restrict__ V,
                                                  const uint8_t* __restrict__ mask, float* __restrict__ ctx,
                                                  int B, int C, float fill) {
  using SH = RaShape<NTE>;
  constexpr int Epad = SH::Epad, LD = SH::LD, NS = SH::NS, NDMA = SH::NDMA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware mapping: all query tiles of one (b,h) share blockIdx % 8 (one L2)
  const int nq = (C + 63) / 64;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int bh = (slot / nq) * 8 + xcd;
  const int qt = slot % nq;
  if (bh >= B * NNJ_NHEAD) return;       // whole workgroup exits together
  const int b = bh / NNJ_NHEAD;
  const size_t base = (size_t)bh * C * Epad;
  const int l15 = lane & 15, kq = lane >> 4;
  const int q0 = qt * 64 + wave * 16;
  const int qi = q0 + l15;               // this lane's query (B operand column)
  const bool qvalid = qi < C;
  const int nkt = (C + RA_KEYS - 1) / RA_KEYS;
  unsigned char* maskl = reinterpret_cast<unsigned char*>(smem + 2 * SH::STAGE_F);   // [nkt*16]

  // key classes for the whole (b) row: 0 = key, 1 = padded key (axial_attention.py:99-103), 2 = beyond C
  for (int j = tid; j < nkt * RA_KEYS; j += 256)
    maskl[j] = j >= C ? 2 : ((mask && mask[(size_t)b * C + j]) ? 1 : 0);

  // DMA plan: the stage image is [K tile | V tile], each KI whole 1-KiB instructions, so every
  // instruction has ONE wave-uniform source base (K or V tile start) plus a 32-bit per-lane offset
  // (0 for lanes that land in the row pad: they re-read the tile's first bytes, harmless).
  unsigned voff[NDMA];
#pragma unroll
  for (int i = 0; i < NDMA; ++i) {
    const int I = wave * NDMA + i;                      // wave-uniform instruction index in the stage
    const int q = (I < SH::KI ? I : I - SH::KI) * 64 + lane;
    const int row = q / SH::CPR, cc = q - row * SH::CPR;
    voff[i] = (row < RA_KEYS && cc < SH::CPE) ? (unsigned)(row * Epad + 4 * cc) : 0u;
  }
  auto issue_tile = [&](int kt, int stage) {
    const int j0 = kt * RA_KEYS;
    float* dst = smem + stage * SH::STAGE_F + wave * NDMA * 256;
    const float* Kt = K + base + (size_t)j0 * Epad;     // wave-uniform tile bases
    const float* Vt = V + base + (size_t)j0 * Epad;
    if (j0 + RA_KEYS <= C) {                            // (wave-uniform) full tile: base + 32-bit offset
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const int I = wave * NDMA + i;
        if (I < 2 * SH::KI) lds_dma16((I < SH::KI ? Kt : Vt) + voff[i], dst + i * 256);
      }
    } else {                                            // last tile of an alignment with C % 16 != 0
#pragma unroll
      for (int i = 0; i < NDMA; ++i) {
        const int I = wave * NDMA + i;
        if (I < 2 * SH::KI) {
          unsigned o = voff[i];
          if ((j0 + (int)(o / Epad)) >= C) o = 0u;      // keys beyond the alignment: finite filler
          lds_dma16((I < SH::KI ? Kt : Vt) + o, dst + i * 256);
        }
      }
    }
  };

  // Q fragment: lane (query, kq) holds Q[query][8s + 2kq + u]
  float qf[2 * NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const float2 v = *reinterpret_cast<const float2*>(Q + base + (size_t)(qvalid ? qi : 0) * Epad + 8 * s + 2 * kq);
    qf[2 * s] = qvalid ? v.x : 0.f; qf[2 * s + 1] = qvalid ? v.y : 0.f;
  }
  f32x4 O[NTE];
#pragma unroll
  for (int t = 0; t < NTE; ++t) O[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_run = 0.f;

  // LDS byte addresses of this lane's operand streams inside a stage
  const unsigned smem_b = lds_addr(smem);
  const unsigned ka0 = smem_b + (unsigned)(l15 * LD + 2 * kq) * 4u;                    // K[key l15][2kq + 8s]
  const unsigned va0 = smem_b + (unsigned)(SH::KI * 256 + 4 * kq * LD + l15) * 4u;     // V[4kq + r][l15 + 16t]
  const unsigned ma0 = lds_addr(maskl) + 4u * kq;

  constexpr int KB = NS < 10 ? NS : 10;            // float2 reads per S batch
  constexpr int NKB = (NS + KB - 1) / KB;

  issue_tile(0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    wait_vmem_all();                      // my pieces of tile kt have landed
    __syncthreads();                      // everyone's pieces landed; everyone is done with tile kt-1
    if (kt + 1 < nkt) issue_tile(kt + 1, (kt + 1) & 1);
    const unsigned sb_ = (unsigned)((kt & 1) * SH::STAGE_F) * 4u;
    const unsigned ka = ka0 + sb_, va = va0 + sb_;
    const int j0 = kt * RA_KEYS;

    // ---- S^T[key x query]: A = K (lane: key l15, kq) from LDS, B = Q regs; two accumulation chains
    f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sbb = {0.f, 0.f, 0.f, 0.f};
    f32x2 kA[KB], kB[KB];
    float mk4;                             // 4 key-class bytes of this lane's keys
    lds_read_b32<0>(mk4, ma0 + (unsigned)j0);
    static_for<0, KB>([&](auto i) { lds_read_b64<32 * decltype(i)::value>(kA[decltype(i)::value], ka); });
    lds_wait_all();
    pin_after_wait(mk4);
#pragma unroll
    for (int i = 0; i < KB; ++i) pin_after_wait(kA[i]);
    static_for<0, NKB>([&](auto bi) {
      constexpr int b0 = decltype(bi)::value * KB;
      constexpr int nb = (NS - b0) < KB ? (NS - b0) : KB;
      constexpr int n0 = b0 + KB;
      constexpr int nn = n0 >= NS ? 0 : ((NS - n0) < KB ? (NS - n0) : KB);
      f32x2 (&cur)[KB] = (decltype(bi)::value & 1) ? kB : kA;
      f32x2 (&nxt)[KB] = (decltype(bi)::value & 1) ? kA : kB;
      static_for<0, nn>([&](auto i) { lds_read_b64<32 * (n0 + decltype(i)::value)>(nxt[decltype(i)::value], ka); });
#pragma unroll
      for (int i = 0; i < nb; ++i) {
        sa = mfma16(cur[i][0], qf[2 * (b0 + i)], sa);
        sbb = mfma16(cur[i][1], qf[2 * (b0 + i) + 1], sbb);
      }
      lds_wait_all();
#pragma unroll
      for (int i = 0; i < nn; ++i) pin_after_wait(nxt[i]);
    });
    // first V batch goes out now; it lands behind the softmax arithmetic
    float vA[NTE], vB[NTE];
    static_for<0, NTE>([&](auto t) { lds_read_b32<64 * decltype(t)::value>(vA[decltype(t)::value], va); });

    // st[r] = S[query = l15][key = j0 + 4*kq + r]
    const unsigned mbits = __float_as_uint(mk4);
    float st[4];
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned cls = (mbits >> (8 * r)) & 0xffu;
      float s = sa[r] + sbb[r];
      s = cls == 2 ? -INFINITY : (cls == 1 ? fill : s);
      st[r] = s;
      tmax = fmaxf(tmax, s);
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float sc = nnj_exp(m_run - m_new);                 // 0 on the first tile (m_run = -inf)
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float p = nnj_exp(st[r] - m_new);
      st[r] = p;
      psum += p;
    }
    psum += __shfl_xor(psum, 16);
    psum += __shfl_xor(psum, 32);
    l_run = l_run * sc + psum;
    m_run = m_new;
    // rescale O rows (row 4*kq+reg of the C/D layout is query 4*kq+reg; its scale lives in the lanes
    // with (lane & 15) == that query); skipped when no running maximum moved in this wave
    if (!__all(sc == 1.0f)) {
      float scr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[r] = __shfl(sc, 4 * kq + r);
#pragma unroll
      for (int t = 0; t < NTE; ++t) { O[t][0] *= scr[0]; O[t][1] *= scr[1]; O[t][2] *= scr[2]; O[t][3] *= scr[3]; }
    }
    lds_wait_all();
#pragma unroll
    for (int t = 0; t < NTE; ++t) pin_after_wait(vA[t]);
    // ---- O[query x e] += P[query x key] V[key x e]; A = P regs, B = V from LDS (one key row per batch)
    static_for<0, 4>([&](auto ri) {
      constexpr int r = decltype(ri)::value;
      float (&cur)[NTE] = (r & 1) ? vB : vA;
      float (&nxt)[NTE] = (r & 1) ? vA : vB;
      if constexpr (r + 1 < 4)
        static_for<0, NTE>([&](auto t) {
          lds_read_b32<((r + 1) * LD + 16 * decltype(t)::value) * 4>(nxt[decltype(t)::value], va);
        });
#pragma unroll
      for (int t = 0; t < NTE; ++t) O[t] = mfma16(st[r], cur[t], O[t]);
      lds_wait_all();
      if constexpr (r + 1 < 4) {
#pragma unroll
        for (int t = 0; t < NTE; ++t) pin_after_wait(nxt[t]);
      }
    });
  }
  // normalise and store: O[t][reg] is (query q0 + 4*kq + reg, e = 16 t + l15)
  float linv[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) linv[r] = nnj_rcp(__shfl(l_run, 4 * kq + r));
#pragma unroll
  for (int t = 0; t < NTE; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qrow = q0 + 4 * kq + r;
      if (qrow < C) ctx[base + (size_t)qrow * Epad + 16 * t + l15] = O[t][r] * linv[r];
    }
  }
}

// ------------------------------------------------------------------ k_tok1
// Row-attention output projection + residual, then the whole column-attention block
// (reference msa_modules.py:109-125 around axial_attention.py:119-138 and 141-255).
//   LDS: W[0..3] four 64x64 weight images (Wo_row -> later Wo_col, Wq, Wk, Wv) 64 KiB
//        + per wave K,V of one head-half [64][32] x 2 = 16 KiB  (x4 waves)
template <int NT>
__global__ __launch_bounds__(256) void k_tok1(const float* __restrict__ ctx, const uint8_t* __restrict__ mask,
                                              float* __restrict__ x, AttnW wr, AttnW wc, int B, int R, int C,
                                              int Epad, int skip_col) {
  // persistent: one workgroup per CU, the five 64x64 weight images staged ONCE (80 KiB), then a loop
  // over groups of 4 columns (one column per wave) with no workgroup barrier inside
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W0 = smem;               // Wo_row
  float* Wq_l = smem + 4096;
  float* Wk_l = smem + 8192;
  float* Wv_l = smem + 12288;
  float* Wo_l = smem + 16384;     // Wo_col
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float* kvl = smem + 20480 + wave * 2048;   // V image of one head-half [64][32], wave private
  stage_weight<64>(W0, wr.Wo, 64, tid, 256);
  stage_weight<64>(Wq_l, wc.Wq, 64, tid, 256);
  stage_weight<64>(Wk_l, wc.Wk, 64, tid, 256);
  stage_weight<64>(Wv_l, wc.Wv, 64, tid, 256);
  stage_weight<64>(Wo_l, wc.Wo, 64, tid, 256);
  __syncthreads();
  const int tok = lane & 31, hh = lane >> 5;
  const long ncols = (long)B * C;
  for (long col = (long)blockIdx.x * 4 + wave; col < ncols; col += (long)gridDim.x * 4) {
  asm volatile("" ::: "memory");            // keep bias / LayerNorm parameter loads inside the loop (see k_ffn)
  const int b = (int)(col / C), c = (int)(col % C);
  const bool padded = mask && mask[(size_t)b * C + c];

  f32x16 xr[NT][2];
  {
    // ---- row attention: out_proj(context) + residual
    f32x16 cx[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = 32 * nt + tok;
      const bool valid = r < R;
      load_token64(xr[nt], x + (((size_t)b * R + (valid ? r : 0)) * C + c) * 64, valid, hh);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int h = 4 * mt + g;
          const f32x4 v = *reinterpret_cast<const f32x4*>(ctx + (((size_t)b * NNJ_NHEAD + h) * C + c) * Epad +
                                                          (valid ? r : 0) * 8 + 4 * hh);
          cx[nt][mt][4 * g + 0] = valid ? v[0] : 0.f; cx[nt][mt][4 * g + 1] = valid ? v[1] : 0.f;
          cx[nt][mt][4 * g + 2] = valid ? v[2] : 0.f; cx[nt][mt][4 * g + 3] = valid ? v[3] : 0.f;
        }
    }
    f32x16 o[NT][2];
    linear_T<2, 2, NT>(o, cx, W0, wr.bo, lane);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) xr[nt][mt] += o[nt][mt];
  }
  if (skip_col & 3) {                        // 1: debug tap (state after the row-attention block); 2: timing ablation
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int r = 32 * nt + tok;
      store_token64(xr[nt], x + (((size_t)b * R + r) * C + c) * 64, r < R, hh);
    }
    continue;
  }

  // ---- column attention
  f32x16 cx[NT][2];
  if (R == 1) {
    // single position: output = out_proj(v_proj(x)) (axial_attention.py:198-209)
    f32x16 y[NT][2];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) layer_norm64(y[nt], xr[nt], wc.ln_w, wc.ln_b, hh);
    linear_T<2, 2, NT>(cx, y, Wv_l, wc.bv, lane);
  } else {
    const float scaling = rsqrtf((float)NNJ_DH);      // axial_attention.py:214
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {                  // heads 4*hf .. 4*hf+3
      f32x16 qh[NT][1], kh[NT][1], vh[NT][1];
      {
        f32x16 y[NT][2];                              // LayerNorm recomputed per head-half: cheaper than 64 live VGPRs
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) layer_norm64(y[nt], xr[nt], wc.ln_w, wc.ln_b, hh);
      linear_T<1, 2, NT>(qh, y, Wq_l + hf * 32 * 64, wc.bq + 32 * hf, lane);
      linear_T<1, 2, NT>(kh, y, Wk_l + hf * 32 * 64, wc.bk + 32 * hf, lane);
      linear_T<1, 2, NT>(vh, y, Wv_l + hf * 32 * 64, wc.bv + 32 * hf, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      // publish v of this head-half for all rows of the column (wave-private LDS image [row][32])
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int r = 32 * nt + tok;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v4 = {vh[nt][0][4 * g], vh[nt][0][4 * g + 1], vh[nt][0][4 * g + 2], vh[nt][0][4 * g + 3]};
          *reinterpret_cast<f32x4*>(kvl + r * 32 + 8 * g + 4 * hh) = v4;
        }
      }
      // per head: S^T[key j x query i] = K_h Q_h^T straight from the projection registers (the q/k
      // accumulators ARE the B/A operands: lane-half hh carries d = 4hh+t), softmax over j inside the
      // lane (+ one exchange with the other half), P.V on the VALU with V broadcast from LDS
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x16 sc_[NT][NT];                       // [query tile][key tile]
#pragma unroll
        for (int it = 0; it < NT; ++it)
#pragma unroll
          for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sc_[it][jt][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
              sc_[it][jt] = mfma32(kh[jt][0][4 * g + t], qh[it][0][4 * g + t] * scaling, sc_[it][jt]);
          }
#pragma unroll
        for (int it = 0; it < NT; ++it) {
          // element reg of tile jt is key j = 32*jt + (reg&3) + 8*(reg>>2) + 4*hh
          float m = -INFINITY;
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int j = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * hh;
              float v = sc_[it][jt][r];
              if (padded) v = -10000.0f;           // every key of a padded column (axial_attention.py:220-224)
              if (j >= R) v = -INFINITY;
              sc_[it][jt][r] = v;
              m = fmaxf(m, v);
            }
          m = fmaxf(m, __shfl_xor(m, 32));
          float l = 0.f, o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float p = nnj_exp(sc_[it][jt][r] - m);
              l += p;
              const float* vp = kvl + (32 * jt + (r & 3) + 8 * (r >> 2) + 4 * hh) * 32 + 8 * g;
              const f32x4 v0 = *reinterpret_cast<const f32x4*>(vp);
              const f32x4 v1 = *reinterpret_cast<const f32x4*>(vp + 4);
#pragma unroll
              for (int t = 0; t < 4; ++t) { o8[t] += p * v0[t]; o8[4 + t] += p * v1[t]; }
            }
          l += __shfl_xor(l, 32);
          const float inv = nnj_rcp(l);
          // this lane keeps d = 4hh+t: add the partner half's partial sums for those d
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float send = hh ? o8[t] : o8[4 + t];
            const float recv = __shfl_xor(send, 32);
            cx[it][hf][4 * g + t] = ((hh ? o8[4 + t] : o8[t]) + recv) * inv;
          }
          __builtin_amdgcn_sched_barrier(0);        // keep the heads' live ranges apart
        }
      }
    }
  }
  f32x16 o[NT][2];
  linear_T<2, 2, NT>(o, cx, Wo_l, wc.bo, lane);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = 32 * nt + tok;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xr[nt][mt] += o[nt][mt];
    store_token64(xr[nt], x + (((size_t)b * R + r) * C + c) * 64, r < R, hh);
  }
  }   // persistent column loop
}

// ------------------------------------------------------------------ k_tok1p
// k_tok1 for 32 < R <= 64 with TWO waves per column (one 32-row tile each), 8 waves per workgroup:
// two waves per SIMD, so the VALU stream (softmax, P.V) issues at the full rate and one wave's VALU
// overlaps the other's MFMAs.  The K and V projections of a head-half are exchanged through a
// per-column LDS image; the two waves of a column meet at a pair barrier built on an LDS counter
// (never a workgroup barrier: the four column pairs run unsynchronised).  Persistent: the five
// weight images are staged once (80 KiB) + 4 x (K image [64][36] + V image [64][32]) = 149 KiB.
__device__ __forceinline__ void pair_barrier(int* cnt, int& epoch) {
  // both waves of the pair arrive (LDS executes a wave's operations in order: its image writes are
  // in place before its increment), then wait until the counter shows both arrivals of this epoch
  epoch += 2;
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) atomicAdd(cnt, 1);
  // bounded spin: a lost partner ends in wrong numbers (caught by the parity tests), never in a hung GPU
  for (int spins = 0; *reinterpret_cast<volatile int*>(cnt) < epoch && spins < (1 << 22); ++spins) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(512) void k_tok1p(const float* __restrict__ ctx, const uint8_t* __restrict__ mask,
                                               float* __restrict__ x, AttnW wr, AttnW wc, int B, int R, int C,
                                               int Epad, int skip_col) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W0 = smem;               // Wo_row
  float* Wq_l = smem + 4096;
  float* Wk_l = smem + 8192;
  float* Wv_l = smem + 12288;
  float* Wo_l = smem + 16384;     // Wo_col
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave & 3, rt = wave >> 2;            // column slot of the workgroup, row tile of the column
  float* kimg = smem + 20480 + slot * (64 * 36 + 64 * 32);   // K image [64][36]
  float* vimg = kimg + 64 * 36;                              // V image [64][32]
  int* cnt = reinterpret_cast<int*>(smem + 20480 + 4 * (64 * 36 + 64 * 32)) + slot;
  stage_weight<64>(W0, wr.Wo, 64, tid, 512);
  stage_weight<64>(Wq_l, wc.Wq, 64, tid, 512);
  stage_weight<64>(Wk_l, wc.Wk, 64, tid, 512);
  stage_weight<64>(Wv_l, wc.Wv, 64, tid, 512);
  stage_weight<64>(Wo_l, wc.Wo, 64, tid, 512);
  if (tid < 4) reinterpret_cast<int*>(smem + 20480 + 4 * (64 * 36 + 64 * 32))[tid] = 0;
  __syncthreads();
  int epoch = 0;
  const int tok = lane & 31, hh = lane >> 5;
  const int r = 32 * rt + tok;                          // this lane's row of the column
  const bool valid = r < R;
  const long ncols = (long)B * C;
  const float scaling = rsqrtf((float)NNJ_DH);         // axial_attention.py:214
  for (long col = (long)blockIdx.x * 4 + slot; col < ncols; col += (long)gridDim.x * 4) {
    asm volatile("" ::: "memory");                      // keep parameter loads inside the loop (see k_ffn)
    const int b = (int)(col / C), c = (int)(col % C);
    const bool padded = mask && mask[(size_t)b * C + c];
    float* xp = x + (((size_t)b * R + (valid ? r : 0)) * C + c) * 64;
    f32x16 xr[1][2];
    {
      // ---- row attention: out_proj(context) + residual
      f32x16 cx[1][2], o[1][2];
      load_token64(xr[0], xp, valid, hh);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int h = 4 * mt + g;
          const f32x4 v = *reinterpret_cast<const f32x4*>(ctx + (((size_t)b * NNJ_NHEAD + h) * C + c) * Epad +
                                                          (valid ? r : 0) * 8 + 4 * hh);
          cx[0][mt][4 * g + 0] = valid ? v[0] : 0.f; cx[0][mt][4 * g + 1] = valid ? v[1] : 0.f;
          cx[0][mt][4 * g + 2] = valid ? v[2] : 0.f; cx[0][mt][4 * g + 3] = valid ? v[3] : 0.f;
        }
      linear_T<2, 2, 1>(o, cx, W0, wr.bo, lane);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) xr[0][mt] += o[0][mt];
    }
    if (skip_col & 3) { store_token64(xr[0], xp, valid, hh); continue; }

    // ---- column attention over the 2 x 32 rows of the column
    f32x16 cx[1][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {                    // heads 4*hf .. 4*hf+3
      f32x16 qh[1][1], kh[1][1], vh[1][1];
      {
        f32x16 y[1][2];
        layer_norm64(y[0], xr[0], wc.ln_w, wc.ln_b, hh);
        linear_T<1, 2, 1>(qh, y, Wq_l + hf * 32 * 64, wc.bq + 32 * hf, lane);
        linear_T<1, 2, 1>(kh, y, Wk_l + hf * 32 * 64, wc.bk + 32 * hf, lane);
        linear_T<1, 2, 1>(vh, y, Wv_l + hf * 32 * 64, wc.bv + 32 * hf, lane);
      }
      pair_barrier(cnt, epoch);                         // the partner has finished reading the previous images
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 k4 = {kh[0][0][4 * g], kh[0][0][4 * g + 1], kh[0][0][4 * g + 2], kh[0][0][4 * g + 3]};
        const f32x4 v4 = {vh[0][0][4 * g], vh[0][0][4 * g + 1], vh[0][0][4 * g + 2], vh[0][0][4 * g + 3]};
        *reinterpret_cast<f32x4*>(kimg + r * 36 + 8 * g + 4 * hh) = k4;
        *reinterpret_cast<f32x4*>(vimg + r * 32 + 8 * g + 4 * hh) = v4;
      }
      pair_barrier(cnt, epoch);                         // both row tiles' K and V are in the images
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // S^T[key j x query i]: A = K image rows (lane = key), B = this wave's q registers (lane = query)
        f32x16 sc_[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const f32x4 ka = *reinterpret_cast<const f32x4*>(kimg + (32 * jt + tok) * 36 + 8 * g + 4 * hh);
#pragma unroll
          for (int k = 0; k < 16; ++k) sc_[jt][k] = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) sc_[jt] = mfma32(ka[t], qh[0][0][4 * g + t] * scaling, sc_[jt]);
        }
        // element k of tile jt is key j = 32*jt + (k&3) + 8*(k>>2) + 4*hh
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const int j = 32 * jt + (k & 3) + 8 * (k >> 2) + 4 * hh;
            float v = sc_[jt][k];
            if (padded) v = -10000.0f;                  // every key of a padded column (axial_attention.py:220-224)
            if (j >= R) v = -INFINITY;
            sc_[jt][k] = v;
            m = fmaxf(m, v);
          }
        m = fmaxf(m, __shfl_xor(m, 32));
        float l = 0.f, o8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            const float p = nnj_exp(sc_[jt][k] - m);
            l += p;
            const float* vp = vimg + (32 * jt + (k & 3) + 8 * (k >> 2) + 4 * hh) * 32 + 8 * g;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(vp);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(vp + 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) { o8[t] += p * v0[t]; o8[4 + t] += p * v1[t]; }
          }
        l += __shfl_xor(l, 32);
        const float inv = nnj_rcp(l);
#pragma unroll
        for (int t = 0; t < 4; ++t) {                   // this lane keeps d = 4hh+t of the head
          const float send = hh ? o8[t] : o8[4 + t];
          const float recv = __shfl_xor(send, 32);
          cx[0][hf][4 * g + t] = ((hh ? o8[4 + t] : o8[t]) + recv) * inv;
        }
      }
    }
    f32x16 o[1][2];
    linear_T<2, 2, 1>(o, cx, Wo_l, wc.bo, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xr[0][mt] += o[0][mt];
    store_token64(xr[0], xp, valid, hh);
  }
}

// ------------------------------------------------------------------ persistent token kernels
// One workgroup per CU, weights staged into LDS ONCE, then a loop over 256-token groups (flat
// (column,row) token order, 8 waves x 32 tokens, two waves per SIMD) with the next group's tokens
// prefetched behind the MFMAs.  No barrier inside the loop.  (The per-group weight re-staging of a
// non-persistent kernel moved 2.7x more bytes than the activations themselves.)
//   k_ffn : LN -> fc1 -> GELU -> fc2 -> +x            LDS 128 KiB: W1 [256][64] | W2 [64][256]
//   k_qkv : LN -> q,k,v of the row attention            LDS  48 KiB: Wq | Wk | Wv
__device__ __forceinline__ void flat_token(int t, int R, int C, int& c, int& r, bool& valid) {
  valid = t < R * C;
  c = valid ? t / R : 0;
  r = valid ? t - c * R : 0;
}

__global__ __launch_bounds__(512) void k_ffn(float* __restrict__ x, float* __restrict__ tmp, FfnW wf, int B, int R,
                                             int C, int groups_per_b) {
  // bf16x6 images of the two weight matrices take 192 KiB, more than the LDS: the hidden layer is done in
  // two halves, each one pass of this workgroup over ITS token groups with the half's weights resident
  // (W1 rows [128p,128p+128) | W2 columns [128p,128p+128): 96 KiB).  Pass 0 leaves b2 + W2a gelu(..) in
  // `tmp` (same token layout as x; every lane re-reads only what it wrote itself), pass 1 adds the other
  // half and the residual.  The additions happen in the same order as in a single pass.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1l = smem;                         // two [64][64] images (hidden units 64q..64q+63 of the half)
  float* W2l = smem + 2 * b6_floats(64, 64); // two [64][64] images
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  const int ngroups = groups_per_b * B;
  if ((int)blockIdx.x >= ngroups) return;
  int c, r; bool valid;
  auto addr = [&](int g_) {
    const int b = g_ / groups_per_b;
    flat_token(((g_ % groups_per_b) * 8 + wave) * 32 + (lane & 31), R, C, c, r, valid);
    return (((size_t)b * R + r) * C + c) * 64;
  };
  static_for<0, 2>([&](auto pi) {
    constexpr int pass = decltype(pi)::value;
    if (pass) __syncthreads();               // everyone is done with the first half's images
#pragma unroll
    for (int q = 0; q < 2; ++q) {              // four [64][64] images: W1 rows / W2 columns 128*pass + 64*q ..
      stage_weight_b6<64>(W1l + q * b6_floats(64, 64), wf.W1 + (size_t)(pass * 128 + q * 64) * 64, 64, tid, 512);
      stage_weight_b6<64, 256>(W2l + q * b6_floats(64, 64), wf.W2, 64, tid, 512, 0, 0, 128 * pass + 64 * q);
    }
    __syncthreads();
    // No register prefetch of the next group: with the bf16x6 fragments next to the hidden tile it would
    // spill, and a spilled prefetch is waited for at once.  Two waves per SIMD: one computes while the
    // other waits for its tokens.
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
      const size_t xo = addr(grp);
      const bool cur_valid = valid;
      // compiler memory barrier: keeps the loop-invariant bias / LayerNorm parameter loads INSIDE the loop
      // (hoisted, they would occupy ~200 VGPRs for the whole kernel and spill)
      asm volatile("" ::: "memory");
      f32x16 y[1][2], out[1][2];
      {
        f32x16 xr[2];
        load_token64(xr, x + xo, cur_valid, hh);
        layer_norm64(y[0], xr, wf.ln_w, wf.ln_b, hh);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
          if (!pass) b4 = *reinterpret_cast<const f32x4*>(wf.b2 + 32 * mt + 8 * g + 4 * hh);
          out[0][mt][4 * g] = b4[0]; out[0][mt][4 * g + 1] = b4[1]; out[0][mt][4 * g + 2] = b4[2]; out[0][mt][4 * g + 3] = b4[3];
        }
#pragma unroll
      for (int q = 0; q < 2; ++q) {              // 64 hidden units at a time
        f32x16 hdn[1][2];
        linear6_T<2, 2, 1>(hdn, y, W1l + q * b6_floats(64, 64), wf.b1 + pass * 128 + q * 64, lane);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int k = 0; k < 16; ++k) hdn[0][mt][k] = gelu_erf(hdn[0][mt][k]);
        linear6_T_acc<2, 2, 1>(out, hdn, W2l + q * b6_floats(64, 64), lane);
      }
      if (pass) {
        // x + (first half + second half): the token (L2) and the first half's sum are re-read here rather
        // than kept in 64 registers through the GEMMs
        f32x16 xo_[2], t0[2];
        load_token64(xo_, x + xo, cur_valid, hh);
        load_token64(t0, tmp + xo, cur_valid, hh);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) xo_[mt] += t0[mt] + out[0][mt];
        store_token64(xo_, x + xo, cur_valid, hh);
      } else {
        store_token64(out[0], tmp + xo, cur_valid, hh);
      }
    }
  });
}

__global__ __launch_bounds__(512) void k_qkv(const float* __restrict__ x, const uint8_t* __restrict__ mask, AttnW wn,
                                             float* __restrict__ Q, float* __restrict__ K, float* __restrict__ V,
                                             int B, int R, int C, int Epad, int groups_per_b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Wl = smem;                          // one bf16x6 image [Wq | Wk | Wv] of 192 rows: y is split once
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hh = lane >> 5;
  stage_weight_b6<64>(Wl, wn.Wq, 64, tid, 512, 0, 192);
  stage_weight_b6<64>(Wl, wn.Wk, 64, tid, 512, 64, 192);
  stage_weight_b6<64>(Wl, wn.Wv, 64, tid, 512, 128, 192);
  __syncthreads();
  const int ngroups = groups_per_b * B;
  const float qs = rsqrtf((float)NNJ_DH) / sqrtf((float)R);
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int b = grp / groups_per_b;
    int c, r; bool valid;
    flat_token(((grp % groups_per_b) * 8 + wave) * 32 + (lane & 31), R, C, c, r, valid);
    asm volatile("" ::: "memory");      // keep the parameter loads inside the loop (see k_ffn)
    f32x16 xr[2], y[1][2], o[1][6];
    load_token64(xr, x + (((size_t)b * R + r) * C + c) * 64, valid, hh);
    layer_norm64(y[0], xr, wn.ln_w, wn.ln_b, hh);
    const bool padded = mask && mask[(size_t)b * C + c];
    const float qscale = padded ? 0.0f : qs;
    linear6_T_nb<6, 2, 1>(o, y, Wl, lane);
    // lane (token, hh) owns d = 4hh..4hh+3 of every head: 16-byte pieces at [b][h][c][r*8 + 4hh]
    auto put = [&](float* dst, const float* bias, int m0, float scale) {
      if (!valid) return;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int h = 4 * mt + g;
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * mt + 8 * g + 4 * hh);
          f32x4 v = {(b4[0] + o[0][m0 + mt][4 * g]) * scale, (b4[1] + o[0][m0 + mt][4 * g + 1]) * scale,
                     (b4[2] + o[0][m0 + mt][4 * g + 2]) * scale, (b4[3] + o[0][m0 + mt][4 * g + 3]) * scale};
          *reinterpret_cast<f32x4*>(dst + (((size_t)b * NNJ_NHEAD + h) * C + c) * Epad + r * 8 + 4 * hh) = v;
        }
    };
    put(Q, wn.bq, 0, qscale);
    put(K, wn.bk, 2, 1.0f);
    put(V, wn.bv, 4, 1.0f);
  }
}
