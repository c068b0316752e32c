// nnj_encoder.hpp -- gfx950 kernels of the axial MSA encoder
// (restates reference model.py:67-88, msa_modules.py:62-151, axial_attention.py:6-255).
//
// Tiling of the token-local stages: one wave owns one alignment column (b, c) and all R rows of it
// ("column wave", tokens = rows, NT = ceil(R/32) tiles of 32 tokens on the lanes), or -- k_ffn16 -- 16 tokens
// of a flat (column, row) order.  Token-local stages chain in registers (nnj_common.hpp); weights sit in LDS.
//
//   k_embed     : embed (6-entry LUT of the site codes, or the embed MLP on float input) -> x
//   k_tok1p     : ctx -> out_proj -> +x ; LN -> q,k,v -> column attention -> out_proj -> +x  (f16x3; one wave per
//                 column for R <= 32, two for R <= 64)
//   k_ffn16     : LN -> fc1 -> GELU -> fc2 -> +x   (persistent, flat token tiling, 16-token tiles, f16x3)
// The tied row attention (q,k,v projections, scores, context) lives in nnj_rowattn.hpp.
//
// HBM layouts: x [B,R,C,64]; ctx head-major [B,8,C,Epad] with e = r*8 + d, Epad = roundup(R*8,16).
#pragma once
#include "nnj_common.hpp"

struct AttnW {            // pointers into the packed device weights (row-major [out][in])
  const float *Wk, *bk, *Wv, *bv, *Wq, *bq, *Wo, *bo, *ln_w, *ln_b;
  int dt;                 // the model's embed_dim (<= 64; narrower models run zero-padded, see layer_norm64)
};
struct FfnW {
  const float *W1, *b1, *W2, *b2, *ln_w, *ln_b;
  int dt;
};
struct EmbedW {           // embed = Linear(4 K -> 64), GELU, Linear(64 -> 64)  (reference model.py:39-43; K = patch_size)
  const float *E0, *e0, *E2, *e2;
};

// ------------------------------------------------------------------ k_embed
// codes uint8 [B,R,L], L = C * K sites (K = patch_size: a token is K consecutive sites, reference model.py:76
// 'b r (c k) e -> b r c (k e)').  K = 1: lut [6][64] = the embed MLP of the six site vectors (reference
// model.py:39-43,76-77 evaluated on phydata.py:38-46's vectors).  K > 1: ptab [K][6][64] = the first Linear's
// contribution of site i of a patch carrying code v (its bias in entry i = 0), summed over the K sites, GELU, then the
// second Linear on the matrix pipe.  Float input [B,R,L,4] (any values): the embed MLP itself.
__global__ __launch_bounds__(256) void k_embed(const uint8_t* __restrict__ codes,
                                               const float* __restrict__ onehot, EmbedW ew,
                                               const float* __restrict__ lut, const float* __restrict__ ptab,
                                               float* __restrict__ x, int B, int R, int C, int K) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* E2_l = smem;             // only staged for the general float input and for patches
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool mlp = onehot != nullptr || K > 1;
  if (mlp) stage_weight<64>(E2_l, ew.E2, 64, tid, 256);
  __syncthreads();
  const long col = (long)blockIdx.x * 4 + wave;   // (b, c) flattened
  if (col >= (long)B * C) return;
  const int b = (int)(col / C), c = (int)(col % C);
  const int tok = lane & 31, hh = lane >> 5;
  const size_t L = (size_t)C * K;
  // a wave owns one alignment column and walks its rows 32 at a time (any number of rows)
  for (int r0 = 0; r0 < R; r0 += 32) {
    const int r = r0 + tok;
    const bool valid = r < R;
    f32x16 xr[1][2];
    if (onehot) {
      // general float input [B,R,L,4]: the embed MLP itself (first Linear on the VALU, second by MFMA)
      f32x16 t1[1][2];
      const float* ohp = onehot + (((size_t)b * R + (valid ? r : 0)) * L + (size_t)c * K) * 4;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(ew.e0 + 32 * mt + 8 * g + 4 * hh);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float* wrow = ew.E0 + (size_t)(32 * mt + 8 * g + 4 * hh + t) * 4 * K;
            float s = b4[t];
            for (int i = 0; i < K; ++i) {
              const f32x4 oh = *reinterpret_cast<const f32x4*>(ohp + 4 * i);
              const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 4 * i);
              s += wv[0] * oh[0] + wv[1] * oh[1] + wv[2] * oh[2] + wv[3] * oh[3];
            }
            t1[0][mt][4 * g + t] = valid ? gelu_erf(s) : 0.f;
          }
        }
      linear_T<2, 2, 1>(xr, t1, E2_l, ew.e2, lane);
    } else if (K > 1) {
      f32x16 t1[1][2];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < 16; ++k) t1[0][mt][k] = 0.f;
      const uint8_t* cp = codes + ((size_t)b * R + (valid ? r : 0)) * L + (size_t)c * K;
      for (int i = 0; i < K; ++i) {
        int code = cp[i];
        if (code > 5) code = 5;
        f32x16 add[2];
        load_token64(add, ptab + ((size_t)i * 6 + code) * 64, true, hh);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) t1[0][mt] += add[mt];
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < 16; ++k) t1[0][mt][k] = valid ? gelu_erf(t1[0][mt][k]) : 0.f;
      linear_T<2, 2, 1>(xr, t1, E2_l, ew.e2, lane);
    } else {
      int code = codes[((size_t)b * R + (valid ? r : 0)) * C + c];
      if (!valid || code > 5) code = 5;
      load_token64(xr[0], lut + code * 64, valid, hh);
    }
    store_token64(xr[0], x + (((size_t)b * R + r) * C + c) * 64, valid, hh);
  }
}
// mask of the tokens of a patched alignment: batch_seq_mask[:, ::patch_size] (reference model.py:79, 167)
__global__ void k_patch_mask(const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, int B, int C, int K) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)B * C) out[i] = mask[(i / C) * ((long)C * K) + (i % C) * K];
}

// ------------------------------------------------------------------ k_tok1p
// Row-attention output projection + residual, then the whole column-attention block
// (reference msa_modules.py:109-125 around axial_attention.py:119-138 and 141-255).
// A column (b, c) is owned by NWC waves, one 32-row tile each (NWC = 1 for R <= 32, 2 for R <= 64), 8 waves per
// workgroup: two waves per SIMD, so one wave's VALU (softmax) overlaps the other's MFMAs.  The K and V
// projections of a head-half go through per-column LDS images; with NWC = 2 the two waves of a column meet at a
// pair barrier built on an LDS counter (never a workgroup barrier: the column slots run unsynchronised).
// Persistent: the five weight images are staged once (80 KiB) + 8/NWC x (K image + V^T image) = 149 KiB.
template <int NWC>
__device__ __forceinline__ void pair_barrier(int* cnt, int& epoch, int* flag) {
  // all NWC waves of the column arrive (LDS executes a wave's operations in order: its image writes are
  // in place before its increment), then wait until the counter shows every arrival of this epoch
  epoch += NWC;
  asm volatile("" ::: "memory");
  if ((threadIdx.x & 63) == 0) atomicAdd(cnt, 1);
  // bounded spin: a lost partner never hangs the GPU; it sets the sticky status bit NNJ_FLAG_BARRIER_TIMEOUT
  // (nnj_numeric_status), so the wrong numbers that follow are never returned silently
  int spins = 0;
  for (; *reinterpret_cast<volatile int*>(cnt) < epoch && spins < (1 << 22); ++spins) __builtin_amdgcn_s_sleep(1);
  if (spins == (1 << 22) && (threadIdx.x & 63) == 0) atomicOr(flag, NNJ_FLAG_BARRIER_TIMEOUT);
  asm volatile("" ::: "memory");
}

// SK (NWC <= 2): the last eight keys of the last 32-key tile are beyond the rows for EVERY lane (R <= 32 NWC - 8, e.g. the
// 50 rows of the bench: keys 56..63) -- their maximum / exponential / sum / split instructions are compiled out (round 4:
// the knock-outs of profiles/r04/ko_tok1p.txt put the softmax at 6.2 of the kernel's 34.8 ms; an eighth of it was padding)
template <int NWC, bool SK = false>                     // waves per column: 1 (R <= 32), 2 (<= 64), 4 (<= 128), 8 (<= 256)
__global__ __launch_bounds__(512) void k_tok1p(const float* __restrict__ ctx, const uint8_t* __restrict__ mask,
                                               float* __restrict__ x, AttnW wr, AttnW wc, int B, int R, int C,
                                               int Epad, int skip_col, int* __restrict__ status) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int QKV = b6_floats(96, 64);                // q | k | v rows of one head half: one operand image
  static_assert(IMG64 == 4096 && QKV == 6144, "the LDS layout below assumes 4-byte operand elements");
  float* W0 = smem;               // Wo_row
  float* Wqkv_l = smem + 4096;    // two [96][64] images
  float* Wo_l = smem + 16384;     // Wo_col
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // column slot of the workgroup, row tile of the column.  Consecutive waves land on different SIMDs: the two
  // waves of a column (which run in lockstep) sit on two SIMDs, and each SIMD hosts waves of two different,
  // unsynchronised columns -- one's VALU phases overlap the other's MFMA phases
  // (NWC = 1: every wave owns a column, no barrier at all.)
  constexpr int NSLOT = 8 / NWC;                        // columns in flight per workgroup
  constexpr int KR = 32 * NWC;                          // rows (= keys) of the column images
  constexpr int SLOTF = KR * 32 + KR * 32;              // floats of a slot: K image (two fp16 planes) + V^T image
  constexpr int KPL = KR * 64;                          // bytes of a K plane: [KR keys][4 heads x 2 halves][4 fp16]
  constexpr int VCH = 4 * NWC;                          // 16-byte chunks of a V^T image row (8 keys each)
  constexpr int VPL = 32 * KR * 2;                      // bytes of a V^T plane
  const int slot = wave / NWC, rt = wave % NWC;
  // K image: two fp16 planes of [KR keys][8 chunks of 8 B]; chunk 2g + hh = the four k values d = 4hh..4hh+3 of
  // head g -- the k-slots 8hh..8hh+3 of ONE 32x32x16 MFMA per (key tile, head, piece product); slots 8hh+4..8hh+7 are
  // zero on both operands (the head dimension is 8).  Chunks are XOR-swizzled by (key >> 2) & 7: the ds_read_b64 of 32
  // keys at one logical chunk is conflict free.  (QK^T used to be four v_mfma_f32_32x32x2_f32 of 64 cycles each per
  // key tile and head: 4096 of the 11000 matrix-pipe cycles of a column; as three f16 piece products it is 1536.)
  uint8_t* kimg = reinterpret_cast<uint8_t*>(smem + 20480 + slot * SLOTF);
  // V^T image: two fp16 planes of [32 (head, d)][KR keys], keys in fragment order (see the P.V loop)
  uint8_t* vimg = kimg + 2 * KPL;
  int* cnt = reinterpret_cast<int*>(smem + 20480 + NSLOT * SLOTF) + slot;
  // f16x3 operand images (4 B per element: the same 16 KiB per matrix as an fp32 image)
  stage_weight_b6<64>(W0, wr.Wo, 64, tid, 512);
  stage_weight_b6<64>(Wo_l, wc.Wo, 64, tid, 512);
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    stage_weight_b6<64>(Wqkv_l + hf * QKV, wc.Wq + hf * 32 * 64, 32, tid, 512, 0, 96);
    stage_weight_b6<64>(Wqkv_l + hf * QKV, wc.Wk + hf * 32 * 64, 32, tid, 512, 32, 96);
    stage_weight_b6<64>(Wqkv_l + hf * QKV, wc.Wv + hf * 32 * 64, 32, tid, 512, 64, 96);
  }
  if (tid < NSLOT) reinterpret_cast<int*>(smem + 20480 + NSLOT * SLOTF)[tid] = 0;
  // bias and LayerNorm vectors in LDS (persistent workgroup; see k_ffn16): row out_proj bias | column ln_w | ln_b |
  // bq | bk | bv | column out_proj bias, 64 floats each
  float* ct = smem + 20480 + NSLOT * SLOTF + 16;
  if (tid < 64) { ct[tid] = wr.bo[tid]; ct[64 + tid] = wc.ln_w[tid]; ct[128 + tid] = wc.ln_b[tid]; ct[192 + tid] = wc.bq[tid];
                  ct[256 + tid] = wc.bk[tid]; ct[320 + tid] = wc.bv[tid]; ct[384 + tid] = wc.bo[tid]; }
  __syncthreads();
  int epoch = 0;
  const int tok = lane & 31, hh = lane >> 5;
  const int r = 32 * rt + tok;                          // this lane's row of the column
  const bool valid = r < R;
  const long ncols = (long)B * C;
  // axial_attention.py:214, times log2(e): the column logits are in log2 units, a probability is one v_exp_f32 (2^x)
  const float scaling = rsqrtf((float)NNJ_DH) * 1.4426950408889634f;
  // The token and its row-attention context of the NEXT column are loaded (raw: rows beyond R read row 0, they
  // are masked as keys and never stored) while the output projection of the current one runs: the two gathers
  // were 12-15k exposed cycles per column at the top of the loop.
  const int rc_ = valid ? r : 0;
  auto load_col = [&](long col_, f32x16 (&xo)[2], f32x16 (&co)[2]) {
    const int b_ = (int)(col_ / C), c_ = (int)(col_ % C);
    load_token64(xo, x + (((size_t)b_ * R + rc_) * C + c_) * 64, true, hh);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int h = 4 * mt + g;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ctx + (((size_t)b_ * NNJ_NHEAD + h) * C + c_) * Epad + rc_ * 8 + 4 * hh);
        co[mt][4 * g + 0] = v[0]; co[mt][4 * g + 1] = v[1]; co[mt][4 * g + 2] = v[2]; co[mt][4 * g + 3] = v[3];
      }
  };
  const long cstride = (long)gridDim.x * NSLOT;
  f32x16 xn[2], cn[2];
  {
    const long col0 = (long)blockIdx.x * NSLOT + slot;
    if (col0 < ncols) load_col(col0, xn, cn);
  }
  for (long col = (long)blockIdx.x * NSLOT + slot; col < ncols; col += cstride) {
    asm volatile("" ::: "memory");                      // keep parameter loads inside the loop (hoisted, they would occupy ~200 VGPRs)
    const int b = (int)(col / C), c = (int)(col % C);
    const bool padded = mask && mask[(size_t)b * C + c];
    const float qscale = padded ? 0.f : scaling;
    float* xp = x + (((size_t)b * R + rc_) * C + c) * 64;
    f32x16 xr[1][2];
    {
      // ---- row attention: out_proj(context) + residual
      f32x16 cx[1][2], o[1][2];
      xr[0][0] = xn[0]; xr[0][1] = xn[1];
      cx[0][0] = cn[0]; cx[0][1] = cn[1];
      linear6_T<2, 2, 1, true>(o, cx, W0, ct, lane);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) xr[0][mt] += o[0][mt];
    }
    if (skip_col & 3) {
      store_token64(xr[0], xp, valid, hh);
      load_col(col + cstride < ncols ? col + cstride : col, xn, cn);
      continue;
    }

    // ---- column attention over the 2 x 32 rows of the column
    f32x16 cx[1][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {                    // heads 4*hf .. 4*hf+3
      // q, k feature-major (lane = row); v TOKEN-major (operands swapped: lane = (head, d), registers = rows) so
      // that its columns go to the V^T image as 8-byte stores.  One pass over y: its fp16 split is shared.
      f32x16 qh, kh, vT;
      {
        f32x16 y[2];
        layer_norm64(y, xr[0], ct + 64, ct + 128, hh, wc.dt);
        const float bvl = ct[320 + 32 * hf + tok];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 q4 = *reinterpret_cast<const f32x4*>(ct + 192 + 32 * hf + 8 * g + 4 * hh);
          const f32x4 k4 = *reinterpret_cast<const f32x4*>(ct + 256 + 32 * hf + 8 * g + 4 * hh);
#pragma unroll
          for (int t = 0; t < 4; ++t) { qh[4 * g + t] = q4[t]; kh[4 * g + t] = k4[t]; vT[4 * g + t] = bvl; }
        }
        const u32x4* img = reinterpret_cast<const u32x4*>(Wqkv_l + hf * QKV);
        static_for<0, 4>([&](auto ki) {
          constexpr int ks = decltype(ki)::value;
          Frag3 bfr;
          split8<8 * (ks & 1)>(bfr, y[ks >> 1]);
          static_for<0, 3>([&](auto mi) {
            constexpr int mt = decltype(mi)::value;
            const int wrow = 32 * mt + tok;
            const int o = wrow * 8 + wswz6<8>(wrow, 2 * ks + hh);
            Frag3 afr;
            afr.h = img[o]; afr.m = img[96 * 8 + o];
            if constexpr (mt == 0) qh = mfma_b6(afr, bfr, qh);
            else if constexpr (mt == 1) kh = mfma_b6(afr, bfr, kh);
            else vT = mfma_b6(bfr, afr, vT);
          });
          __builtin_amdgcn_sched_barrier(0);
        });
      }
      if constexpr (NWC >= 2) pair_barrier<NWC>(cnt, epoch, status);  // the partners have finished reading the previous images
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        {
          unsigned kh01, km01, kh23, km23;
          split2(kh[4 * g], kh[4 * g + 1], kh01, km01);
          split2(kh[4 * g + 2], kh[4 * g + 3], kh23, km23);
          uint8_t* kd = kimg + r * 64 + 8 * ((2 * g + hh) ^ ((r >> 2) & 7));
          *reinterpret_cast<uint2*>(kd) = make_uint2(kh01, kh23);
          *reinterpret_cast<uint2*>(kd + KPL) = make_uint2(km01, km23);
        }
        // vT registers 4g..4g+3: rows 32rt + 8g + 4hh + 0..3 of feature `tok`.  Key 32jt + 16u + 8v + 4hh + w sits at
        // position 32jt + 16u + 8hh + 4v + w of the image row: the 8 keys a lane multiplies in one k-step (16 keys)
        // are one 16-byte chunk, chunk index 4jt + 2u + hh, in the order the probability registers have
        unsigned h01, m01, h23, m23;
        split2(vT[4 * g], vT[4 * g + 1], h01, m01);
        split2(vT[4 * g + 2], vT[4 * g + 3], h23, m23);
        const int chunk = 4 * rt + 2 * (g >> 1) + hh;
        uint8_t* dst = vimg + tok * (16 * VCH) + 16 * wswz6<VCH>(tok, chunk) + 8 * (g & 1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(dst + VPL) = make_uint2(m01, m23);
      }
      if constexpr (NWC >= 2) pair_barrier<NWC>(cnt, epoch, status);  // every row tile's K and V are in the images
      if constexpr (NWC <= 2) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // S^T[key j x query i]: A = K image rows (lane = key), B = this wave's q registers (lane = query)
        f32x16 sc_[NWC];
        Frag3 qf;                                          // this lane's four q values of head g, scaled, as pieces
        {
          unsigned h01, m01, h23, m23;
          split2(qh[4 * g] * qscale, qh[4 * g + 1] * qscale, h01, m01);
          split2(qh[4 * g + 2] * qscale, qh[4 * g + 3] * qscale, h23, m23);
          qf.h = (u32x4){h01, h23, 0u, 0u};
          qf.m = (u32x4){m01, m23, 0u, 0u};
        }
#pragma unroll
        for (int jt = 0; jt < NWC; ++jt) {
          const int kr = 32 * jt + tok;
          const uint8_t* ks_ = kimg + kr * 64 + 8 * ((2 * g + hh) ^ ((kr >> 2) & 7));
          const uint2 ah = *reinterpret_cast<const uint2*>(ks_), am = *reinterpret_cast<const uint2*>(ks_ + KPL);
          Frag3 kf;
          kf.h = (u32x4){ah.x, ah.y, 0u, 0u};
          kf.m = (u32x4){am.x, am.y, 0u, 0u};
#pragma unroll
          for (int k = 0; k < 16; ++k) sc_[jt][k] = 0.f;
          sc_[jt] = mfma_b6(kf, qf, sc_[jt]);
        }
        // element k of tile jt is key j = 32*jt + (k&3) + 8*(k>>2) + 4*hh.  A padded column gets the same score
        // for every key (axial_attention.py:220-224: -10000 everywhere): its q is scaled by 0 instead, the
        // softmax of equal scores is the same.  Keys beyond the rows exist only in the second tile (R > 32).
        float m = -INFINITY;
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < NWC; ++jt)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            if (SK && jt == NWC - 1 && k >= 12) continue;       // keys 32 jt + 24 .. + 31: beyond the rows in every lane
            float v = sc_[jt][k];
            if (jt == NWC - 1 && 32 * jt + (k & 3) + 8 * (k >> 2) + 4 * hh >= R) v = -INFINITY;
            sc_[jt][k] = v;
            m = fmaxf(m, v);
          }
        m = fmaxf(m, __shfl_xor(m, 32));
#pragma unroll
        for (int jt = 0; jt < NWC; ++jt)
#pragma unroll
          for (int k = 0; k < 16; ++k) {
            if (SK && jt == NWC - 1 && k >= 12) { sc_[jt][k] = 0.f; continue; }
            const float p = __builtin_amdgcn_exp2f(sc_[jt][k] - m);
            sc_[jt][k] = p;
            l += p;
          }
        l += __shfl_xor(l, 32);
        // O^T[(head, d) x query] = V^T P^T on the fp16 pipe: the probability registers 8u..8u+7 of tile jt are the B
        // operand of k-step 2jt + u as they stand; A = the whole V^T image (all four heads of the half: only the
        // rows of head g mean anything against P_g, i.e. registers 4g..4g+3 of the product -- d = 4hh + t)
        f32x16 o;
#pragma unroll
        for (int k = 0; k < 16; ++k) o[k] = 0.f;
        static_for<0, 2 * NWC>([&](auto si) {
          constexpr int ks = decltype(si)::value;
          Frag3 pf, vf;
          if constexpr (SK && ks == 2 * NWC - 1) {
            // the last k-step: probabilities 12..15 of the tile are exact zeros (keys beyond the rows): no split
            unsigned h, m_;
            split2(sc_[ks >> 1][8], sc_[ks >> 1][9], h, m_); pf.h[0] = h; pf.m[0] = m_;
            split2(sc_[ks >> 1][10], sc_[ks >> 1][11], h, m_); pf.h[1] = h; pf.m[1] = m_;
            pf.h[2] = 0u; pf.h[3] = 0u; pf.m[2] = 0u; pf.m[3] = 0u;
          } else {
            split8<8 * (ks & 1)>(pf, sc_[ks >> 1]);
          }
          const uint8_t* src = vimg + tok * (16 * VCH) + 16 * wswz6<VCH>(tok, 2 * ks + hh);
          vf.h = *reinterpret_cast<const u32x4*>(src);
          vf.m = *reinterpret_cast<const u32x4*>(src + VPL);
          o = mfma_b6(vf, pf, o);
        });
        const float inv = nnj_rcp(l);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          float ov = o[0];                              // register 4g + t, g a compile-time constant after unrolling
#pragma unroll
          for (int k = 1; k < 16; ++k) ov = (k == 4 * g + t) ? o[k] : ov;
          cx[0][hf][4 * g + t] = ov * inv;
        }
      }
      } else {
        // More than two row tiles (R > 64): one key tile at a time with a running maximum (online softmax), so the
        // register cost does not grow with the number of tiles.  Same operands as above.
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float m = -INFINITY, l = 0.f;
          f32x16 o;
#pragma unroll
          for (int k = 0; k < 16; ++k) o[k] = 0.f;
          Frag3 qf;
          {
            unsigned h01, m01, h23, m23;
            split2(qh[4 * g] * qscale, qh[4 * g + 1] * qscale, h01, m01);
            split2(qh[4 * g + 2] * qscale, qh[4 * g + 3] * qscale, h23, m23);
            qf.h = (u32x4){h01, h23, 0u, 0u};
            qf.m = (u32x4){m01, m23, 0u, 0u};
          }
#pragma unroll 1
          for (int jt = 0; jt < NWC; ++jt) {
            const int kr = 32 * jt + tok;
            const uint8_t* ks_ = kimg + kr * 64 + 8 * ((2 * g + hh) ^ ((kr >> 2) & 7));
            const uint2 ah = *reinterpret_cast<const uint2*>(ks_), am = *reinterpret_cast<const uint2*>(ks_ + KPL);
            Frag3 kf;
            kf.h = (u32x4){ah.x, ah.y, 0u, 0u};
            kf.m = (u32x4){am.x, am.y, 0u, 0u};
            f32x16 sc;
#pragma unroll
            for (int k = 0; k < 16; ++k) sc[k] = 0.f;
            sc = mfma_b6(kf, qf, sc);
            float mt = -INFINITY;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              float v = sc[k];
              if (32 * jt + (k & 3) + 8 * (k >> 2) + 4 * hh >= R) v = -INFINITY;     // keys beyond the rows
              sc[k] = v;
              mt = fmaxf(mt, v);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mn = fmaxf(m, mt);                  // finite from the first tile on (rows 0..31 exist)
            const float corr = __builtin_amdgcn_exp2f(m - mn);             // first tile: exp(-inf) = 0 on l = 0, o = 0
            l *= corr;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
              o[k] *= corr;
              const float p = __builtin_amdgcn_exp2f(sc[k] - mn);          // masked keys: exp(-inf) = 0
              sc[k] = p;
              l += p;
            }
            m = mn;
            static_for<0, 2>([&](auto ui) {
              constexpr int u = decltype(ui)::value;
              Frag3 pf, vf;
              split8<8 * u>(pf, sc);
              const uint8_t* src = vimg + tok * (16 * VCH) + 16 * wswz6<VCH>(tok, 2 * (2 * jt + u) + hh);
              vf.h = *reinterpret_cast<const u32x4*>(src);
              vf.m = *reinterpret_cast<const u32x4*>(src + VPL);
              o = mfma_b6(vf, pf, o);
            });
          }
          l += __shfl_xor(l, 32);
          const float inv = nnj_rcp(l);
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            float ov = o[0];
#pragma unroll
            for (int k = 1; k < 16; ++k) ov = (k == 4 * g + t) ? o[k] : ov;
            cx[0][hf][4 * g + t] = ov * inv;
          }
        }
      }
    }
    load_col(col + cstride < ncols ? col + cstride : col, xn, cn);      // (last column: harmless reload; unconditional, so
    f32x16 o[1][2];                                                      //  the old values are dead across the iteration)
    linear6_T<2, 2, 1, true>(o, cx, Wo_l, ct + 384, lane);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) xr[0][mt] += o[0][mt];
    store_token64(xr[0], xp, valid, hh);
  }
}

// ------------------------------------------------------------------ k_ffn16 (persistent)
// One workgroup per CU, weights staged into LDS once per pass, then a loop over 256-token groups (flat
// (column,row) token order).  No barrier inside the loop.  (The per-group weight re-staging of a
// non-persistent kernel moved 2.7x more bytes than the activations themselves.)
__device__ __forceinline__ void flat_token(int t, int R, int C, int& c, int& r, bool& valid) {
  valid = t < R * C;
  c = valid ? t / R : 0;
  r = valid ? t - c * R : 0;
}

// FFN on 16-token tiles: 16 waves x 16 tokens = 256-token groups, 16 registers per tensor, four waves per SIMD
// (<= 128 registers): the LayerNorm -> fc1 -> GELU -> fc2 chain of one wave is latency bound, the other three
// fill its gaps (no register prefetch needed).  Both weight matrices are resident as eight [64][64] operand
// images (128 KiB); the hidden layer is walked 64 units at a time and never leaves the registers.
constexpr int FFN_PF = 2;     // fragment reads in flight ahead of the MFMAs (k_ffn16 runs at 128 registers)
__global__ __launch_bounds__(1024) void k_ffn16(float* __restrict__ x, FfnW wf, int B, int R, int C, int groups_per_b) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* W1l = smem;                         // four [64][64] images: hidden units 64q..64q+63
  float* W2l = smem + 4 * IMG64;             // four [64][64] images: hidden columns 64q..64q+63
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  const int ngroups = groups_per_b * B;
  if ((int)blockIdx.x >= ngroups) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    stage_weight_t16(W1l + q * IMG64, wf.W1 + (size_t)(q * 64) * 64, 64, tid, 1024);
    stage_weight_t16(W2l + q * IMG64, wf.W2, 64, tid, 1024, false, 256, 64 * q);
  }
  // LayerNorm scale / shift and the two bias vectors stay in LDS for the lifetime of the (persistent) workgroup: read
  // from global memory inside the loop they are 28 small L2 round trips per 16-token tile
  float* cf = smem + 8 * IMG64;              // ln_w[64] | ln_b[64] | b2[64] | b1[256]
  if (tid < 64) { cf[tid] = wf.ln_w[tid]; cf[64 + tid] = wf.ln_b[tid]; cf[128 + tid] = wf.b2[tid]; }
  if (tid < 256) cf[192 + tid] = wf.b1[tid];
  __syncthreads();
  const int nq = (4 * wf.dt + 63) / 64;      // hidden blocks of 64 in use: the ffn width is 4 embed_dim (model.py:28)
  for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
    const int b = grp / groups_per_b;
    int c, r; bool valid;
    flat_token(((grp % groups_per_b) * 16 + wave) * 16 + l15, R, C, c, r, valid);
    float* xp = x + (((size_t)b * R + r) * C + c) * 64;
    asm volatile("" ::: "memory");          // keep the parameter loads inside the loop (hoisted, they would occupy ~200 VGPRs)
    V64 xr, y, out;
    load_v64(xr, xp, kq);
    layer_norm_v64(y, xr, cf, cf + 64, kq, wf.dt);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) out.t[mt] = *reinterpret_cast<const f32x4*>(cf + 128 + 16 * mt + 4 * kq);
    // y is split ONCE (inside the loop it was split again for each of the four hidden blocks); both products read
    // their weight fragments by hand ahead of the MFMAs, at two base addresses + immediate offsets (the compiler's own
    // reads recomputed a swizzled address per fragment: 34 vector adds per 36 reads)
    Frag3 yf[2];
    split_8(yf[0], y.t[0], y.t[1]);
    split_8(yf[1], y.t[2], y.t[3]);
#pragma unroll 1
    for (int q = 0; q < nq; ++q) {             // 64 hidden units at a time (a narrower model's padded blocks are skipped)
      V64 hdn;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) hdn.t[mt] = *reinterpret_cast<const f32x4*>(cf + 192 + q * 64 + 16 * mt + 4 * kq);
      linear_t16p_core<4, FFN_PF>(hdn.t, yf, W1l + q * IMG64, lane, [] {});
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const f32x2v g0 = gelu_erf2((f32x2v){hdn.t[mt][0], hdn.t[mt][1]}), g1 = gelu_erf2((f32x2v){hdn.t[mt][2], hdn.t[mt][3]});
        hdn.t[mt] = (f32x4){g0[0], g0[1], g1[0], g1[1]};
      }
      linear_t16p<4, true, false, FFN_PF>(out.t, hdn, W2l + q * IMG64, nullptr, lane);
    }
    if (!valid) continue;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4*>(xp + 16 * mt + 4 * kq) = xr.t[mt] + out.t[mt];
  }
}
