// nnj_encoder64.hpp -- the axial MSA encoder in fp64 for alignments of MORE THAN 64 rows (gfx950).
// (restates reference model.py:67-88, msa_modules.py:62-151, axial_attention.py:6-255)
//
// Why it exists: above 64 rows the sharpened-weight fixtures amplify the encoder's fp32-level noise
// 10-20x in the pair scorer -- profiles/r03/cfg5_margin.txt: the step-0 table of 200 x 4096 is 2.9e-4 of scale from the
// fp64 evaluation with THIS library's f16x3 encoder, 2.6e-4 with the plain-fp32 oracle's encoder output (24-bit operands,
// fp32 accumulation: the reference's own arithmetic) and 6.3e-6 with an fp64 encoder output in front of the same scorer.
// No operand split cures that (the fp32 oracle has exact operands); only better-than-fp32 accumulation and vector
// arithmetic throughout does.  gfx950 has native fp64 on both pipes (v_mfma_f64_16x16x4_f64, 78.6 TFLOP/s dense), so
// the > 64-row path runs the encoder in fp64 and rounds the embeddings to fp32 ONCE, at the end.  The <= 64-row kernels
// (BASELINE configs[1..3]) are untouched.
//
// Structure: one tiled fp64 GEMM (k64_gemm) with epilogue functors does every contraction -- the q/k/v projections
// (written straight into the [head][e = r*8+d][site] operand planes of the tied attention), the tied logits
// S = Q^T K (TN form), P V (NT form), the output projections and both FFN layers -- beside five small kernels:
// embed, LayerNorm, row softmax, column attention (R x R per column and head, 8-wide heads: vector fp64), store.
// Everything is per alignment (the host loops over the batch with ONE alignment's workspace).
//
// MFMA f64 16x16x4 maps (cdna_hip_programming.md section 3): A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// C/D: col = lane & 15, row = (lane >> 4) + 4 * reg.
#pragma once
#include "nnj_common.hpp"

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f64x4 mfma64(double a, double b, f64x4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------ k64_gemm
// C[m][n] = sum_k A(m, k) B(n, k), batched over blockIdx.z, handed element by element to the epilogue functor.
//   TA = false: A stored [M][lda], k contiguous;  TA = true: A stored [K][lda], m contiguous.  TB likewise for B / n.
// Tile BM x BN x 16, 256 threads = 2 x 2 waves, each wave (BM/2) x (BN/2) = TM x TN MFMA tiles of 16 x 16.
// The next k-tile travels global -> registers while the current one is multiplied from LDS (two barriers per k-tile).
// LDS images: k-contiguous operands as [row][8 granules of 16 B], granule g at g ^ ((row >> 1) & 7): the fragment read
// (16 rows x 2 k per half-wave) is conflict free; m-contiguous operands as [16 k][BM + 16] (stride = 16 mod 32 doubles).
// Any M, N, K: rows beyond M / N are clamped on load (their results are never stored), k beyond K is zero filled.
// lda / ldb and the base pointers must keep the 16-byte granules aligned (even leading dimensions).
struct Gemm64 {
  const double* A; long lda, sA;      // sA / sB: elements between consecutive batches (blockIdx.z)
  const double* B; long ldb, sB;
  int M, N, K;
};

template <int BR, bool TR>
struct Tile64 {
  static constexpr int GRAN = BR * 8;                       // 16-byte granules per 16-k tile
  static constexpr int PER = GRAN / 256;                    // per thread
  static constexpr int LD = BR + 16;                        // m-contiguous image: doubles per k row
  static constexpr int DOUBLES = TR ? 16 * LD : BR * 16;
  static_assert(GRAN % 256 == 0, "tile rows must be a multiple of 32");
  f64x2 r[PER];
  // global -> registers: k-tile starting at k0
  __device__ __forceinline__ void load(const double* __restrict__ base, long ld, int row0, int rows, int k0, int K, int tid) {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = tid + 256 * i;
      if constexpr (!TR) {
        const int rr = q >> 3, gq = q & 7;
        int row = row0 + rr;
        if (row >= rows) row = rows - 1;
        const int k = k0 + 2 * gq;
        f64x2 v = *reinterpret_cast<const f64x2*>(base + (long)row * ld + (k + 1 < K ? k : 0));
        // (the clamped address keeps the load inside the row; the values are zeroed below when k is past K)
        if (k + 1 >= K) {
          const double lo = k < K ? base[(long)row * ld + k] : 0.0;
          v = (f64x2){lo, 0.0};
        }
        r[i] = v;
      } else {
        constexpr int GR = BR / 2;                            // granules per k row
        const int kk = q / GR, gq = q % GR;
        const int k = k0 + kk;
        int m = row0 + 2 * gq;
        if (m >= rows) m = 0;                                 // a clamped pair: only feeds results that are never stored
        // (m + 1 == rows reads the pad element of the row -- ld >= roundup(rows, 2) -- which feeds a result that is never stored)
        f64x2 v = {0.0, 0.0};
        if (k < K) v = *reinterpret_cast<const f64x2*>(base + (long)k * ld + m);
        r[i] = v;
      }
    }
  }
  __device__ __forceinline__ void store(double* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int q = tid + 256 * i;
      if constexpr (!TR) {
        const int rr = q >> 3, gq = q & 7;
        *reinterpret_cast<f64x2*>(lds + rr * 16 + ((gq ^ ((rr >> 1) & 7)) << 1)) = r[i];
      } else {
        constexpr int GR = BR / 2;
        const int kk = q / GR, gq = q % GR;
        *reinterpret_cast<f64x2*>(lds + kk * LD + 2 * gq) = r[i];
      }
    }
  }
  // fragment element (row rr of the tile, k of the tile)
  __device__ __forceinline__ static double frag(const double* lds, int rr, int k) {
    if constexpr (!TR) return lds[rr * 16 + (((k >> 1) ^ ((rr >> 1) & 7)) << 1) + (k & 1)];
    else return lds[k * LD + rr];
  }
};

template <int BM, int BN, bool TA, bool TB, typename EPI>
__global__ __launch_bounds__(256, 2) void k64_gemm(Gemm64 g, EPI epi) {
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  using TileA = Tile64<BM, TA>;
  using TileB = Tile64<BN, TB>;
  double* As = smem64;
  double* Bs = smem64 + TileA::DOUBLES;
  constexpr int TM = BM / 32, TN = BN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int z = blockIdx.z;
  const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
  const double* A = g.A + (long)z * g.sA;
  const double* B = g.B + (long)z * g.sB;
  f64x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f64x4){0.0, 0.0, 0.0, 0.0};
  TileA ta;
  TileB tb;
  const int nk = (g.K + 15) / 16;
  ta.load(A, g.lda, m_blk, g.M, 0, g.K, tid);
  tb.load(B, g.ldb, n_blk, g.N, 0, g.K, tid);
  const int l15 = lane & 15, kq = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    ta.store(As, tid);
    tb.store(Bs, tid);
    __syncthreads();
    if (kt + 1 < nk) {
      ta.load(A, g.lda, m_blk, g.M, 16 * (kt + 1), g.K, tid);
      tb.load(B, g.ldb, n_blk, g.N, 16 * (kt + 1), g.K, tid);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = TileA::frag(As, wm * (BM / 2) + 16 * i + l15, 4 * ks + kq);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = TileB::frag(Bs, wn * (BN / 2) + 16 * j + l15, 4 * ks + kq);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = mfma64(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m_blk + wm * (BM / 2) + 16 * i + kq + 4 * r;
        const int n = n_blk + wn * (BN / 2) + 16 * j + l15;
        if (m < g.M && n < g.N) epi(z, m, n, acc[i][j][r]);
      }
}
template <int BM, int BN, bool TA, bool TB>
constexpr size_t gemm64_lds() { return (size_t)(Tile64<BM, TA>::DOUBLES + Tile64<BN, TB>::DOUBLES) * sizeof(double); }

// ------------------------------------------------------------------ epilogues
// q/k/v of the tied row attention (axial_attention.py:66-95), GEMM with M = 192 stacked output features (q | k | v) and
// N = the R*C tokens of one alignment (token t = r*C + c): planes [head][e = r*8 + d][site c], ld = Cp.
// q is scaled by dh^-0.5 / sqrt(R) and zeroed on padded sites (:77-82).
struct EpiRowQkv {
  double *Q, *K, *V; const double* bias; const uint8_t* mask; int C; long Cp; long plane; double scaling;
  __device__ __forceinline__ void operator()(int, int m, int n, double v) const {
    const int which = m >> 6, f = m & 63, hd = f >> 3, d = f & 7;
    const int r = n / C, c = n - r * C;
    v += bias[m];
    if (which == 0) v = (mask && mask[c]) ? 0.0 : v * scaling;
    double* dst = which == 0 ? Q : (which == 1 ? K : V);
    dst[(long)hd * plane + (long)(r * 8 + d) * Cp + c] = v;
  }
};
// tied logits (axial_attention.py:97-103): keys of padded sites get the fill of every row chunk, summed (:35-64)
struct EpiLogits {
  double* S; long ld, plane; const uint8_t* mask; double fill;
  __device__ __forceinline__ void operator()(int z, int m, int n, double v) const {
    S[(long)z * plane + (long)m * ld + n] = (mask && mask[n]) ? fill : v;
  }
};
// context (axial_attention.py:114): z = head, m = query site i, n = e = r*8 + d  ->  ctx[token r*C + i][head*8 + d]
struct EpiCtx {
  double* ctx; int C;
  __device__ __forceinline__ void operator()(int z, int m, int n, double v) const {
    ctx[((long)(n >> 3) * C + m) * 64 + z * 8 + (n & 7)] = v;
  }
};
// x[token][f] += acc + bias[f]  (output projections, fc2: the residual of NormalizedResidualBlock, msa_modules.py:109-125)
struct EpiResid {
  double* x; const double* bias;
  __device__ __forceinline__ void operator()(int, int m, int n, double v) const { x[(long)m * 64 + n] += v + bias[n]; }
};
// column attention's q | k | v, token-major [token][192]; q scaled by dh^-0.5 (axial_attention.py:214)
struct EpiColQkv {
  double* out; const double* bias; double scaling;
  __device__ __forceinline__ void operator()(int, int m, int n, double v) const {
    v += bias[n];
    if (n < 64) v *= scaling;
    out[(long)m * 192 + n] = v;
  }
};
__device__ __forceinline__ double gelu64(double x) { return 0.5 * x * (1.0 + erf(x * 0.70710678118654752440)); }
// fc1 + exact-erf GELU (msa_modules.py:140-151)
struct EpiGelu {
  double* out; const double* bias;
  __device__ __forceinline__ void operator()(int, int m, int n, double v) const { out[(long)m * 256 + n] = gelu64(v + bias[n]); }
};

// ------------------------------------------------------------------ small kernels
// embed (model.py:39-43, 76-77) of ONE alignment: codes uint8 [R][L] or float one-hot [R][L][4], L = C K sites.
// lut [6][64] (K = 1) and ptab [K][6][64] are the fp64 tables nnj_load_weights builds.  One workgroup = 4 tokens x 64 features.
struct Embed64W { const double *E0, *e0, *E2, *e2, *lut, *ptab; };
__global__ __launch_bounds__(256) void k64_embed(const uint8_t* __restrict__ codes, const float* __restrict__ onehot,
                                                 Embed64W w, double* __restrict__ x, int R, int C, int K) {
  __shared__ double t1[4][64];
  const int f = threadIdx.x & 63, slot = threadIdx.x >> 6;
  const long tok = (long)blockIdx.x * 4 + slot, ntok = (long)R * C;
  const bool valid = tok < ntok;
  const long t = valid ? tok : ntok - 1;
  const int r = (int)(t / C), c = (int)(t % C);
  const long L = (long)C * K;
  if (!onehot && K == 1) {
    int code = codes[(long)r * L + c];
    if (code > 5) code = 5;
    if (valid) x[t * 64 + f] = w.lut[code * 64 + f];
    return;
  }
  double s;
  if (onehot) {
    s = w.e0[f];
    const float* ohp = onehot + ((long)r * L + (long)c * K) * 4;
    for (int i = 0; i < 4 * K; ++i) s += w.E0[(long)f * 4 * K + i] * (double)ohp[i];
  } else {
    s = 0.0;
    const uint8_t* cp = codes + (long)r * L + (long)c * K;
    for (int i = 0; i < K; ++i) {
      int code = cp[i];
      if (code > 5) code = 5;
      s += w.ptab[((long)i * 6 + code) * 64 + f];
    }
  }
  t1[slot][f] = gelu64(s);
  __syncthreads();
  double o = w.e2[f];
  for (int j = 0; j < 64; ++j) o += w.E2[f * 64 + j] * t1[slot][j];
  if (valid) x[t * 64 + f] = o;
}

// y = LayerNorm(x) over the model's dt features (eps 1e-5, biased variance, msa_modules.py:107); 16 lanes per token.
__global__ __launch_bounds__(256) void k64_layernorm(const double* __restrict__ x, double* __restrict__ y,
                                                     const double* __restrict__ gamma, const double* __restrict__ beta,
                                                     long ntok, int dt) {
  const long tok = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int q = threadIdx.x & 15;
  const long t = tok < ntok ? tok : ntok - 1;
  const f64x2 v0 = *reinterpret_cast<const f64x2*>(x + t * 64 + 4 * q);
  const f64x2 v1 = *reinterpret_cast<const f64x2*>(x + t * 64 + 4 * q + 2);
  double s = (v0[0] + v0[1]) + (v1[0] + v1[1]);
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o);
  const double mean = s / (double)dt;
  const double d[4] = {v0[0] - mean, v0[1] - mean, v1[0] - mean, v1[1] - mean};
  double var = 0.0;
#pragma unroll
  for (int e = 0; e < 4; ++e) var += (4 * q + e < dt) ? d[e] * d[e] : 0.0;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) var += __shfl_xor(var, o);
  const double inv = 1.0 / sqrt(var / (double)dt + 1e-5);
  if (tok < ntok) {
#pragma unroll
    for (int e = 0; e < 4; ++e) y[t * 64 + 4 * q + e] = d[e] * inv * gamma[4 * q + e] + beta[4 * q + e];
  }
}

// softmax over the keys of every row of S [rows][ld] in place (axial_attention.py:54 / 132); one workgroup per row
__global__ __launch_bounds__(256) void k64_softmax_rows(double* __restrict__ S, long ld, int C) {
  __shared__ double red[4];
  double* row = S + (long)blockIdx.x * ld;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double mx = -1.0e300;
  for (int j = tid; j < C; j += 256) mx = fmax(mx, row[j]);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) mx = fmax(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
  __syncthreads();
  double s = 0.0;
  for (int j = tid; j < C; j += 256) {
    const double e = exp(row[j] - mx);
    row[j] = e;
    s += e;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) s += __shfl_xor(s, o);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const double inv = 1.0 / ((red[0] + red[1]) + (red[2] + red[3]));
  for (int j = tid; j < C; j += 256) row[j] *= inv;
}

// column attention of one alignment column (axial_attention.py:190-255): qkv [token r*C + c][192] (q pre-scaled) ->
// ctx [token][64].  One workgroup per (column, PAIR of heads); thread (row group rg = tid >> 1, head tid & 1 of the pair)
// owns the NU query rows rg, rg + 128 of its head: every key / value vector it reads from LDS (the column's k | v of the
// two heads, [R][32] doubles = 51 KB at 200 rows: three workgroups per CU; broadcast reads) feeds NU rows' products.
// Two passes over the keys (maximum, then exponentials): one exp per (query, key, head).  Every key of a padded column
// carries the fill -10000 (:220-224): the softmax is uniform.
template <int NU>
__global__ __launch_bounds__(256) void k64_col_attention(const double* __restrict__ qkv, double* __restrict__ ctx,
                                                         const uint8_t* __restrict__ mask, int R, int C) {
  extern __shared__ __attribute__((aligned(16))) double kv[];          // [R][32]: k[16] | v[16] of heads 2 hy, 2 hy + 1
  const int c = blockIdx.x, hy = blockIdx.y, tid = threadIdx.x;
  for (int i = tid; i < R * 16; i += 256) {
    const int r = i >> 4, p = i & 15;                                  // p < 8: k pairs, else v pairs
    const long src = ((long)r * C + c) * 192 + (p < 8 ? 64 : 128) + 16 * hy + 2 * (p & 7);
    *reinterpret_cast<f64x2*>(kv + r * 32 + 2 * p) = *reinterpret_cast<const f64x2*>(qkv + src);
  }
  __syncthreads();
  const bool padded = mask && mask[c];
  const int hd = tid & 1, rg = tid >> 1;
  double q[NU][8], mx[NU], sum[NU], acc[NU][8];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = rg + 128 * u;
    const double* qrow = qkv + ((long)(i < R ? i : 0) * C + c) * 192 + 16 * hy + 8 * hd;
#pragma unroll
    for (int d = 0; d < 8; ++d) { q[u][d] = qrow[d]; acc[u][d] = 0.0; }
    mx[u] = padded ? -10000.0 : -1.0e300;
    sum[u] = 0.0;
  }
  if (!padded)
    for (int j = 0; j < R; ++j) {
      const double* kj = kv + j * 32 + hd * 8;
      double k[8];
#pragma unroll
      for (int d = 0; d < 8; ++d) k[d] = kj[d];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        double s = 0.0;
#pragma unroll
        for (int d = 0; d < 8; ++d) s += q[u][d] * k[d];
        mx[u] = fmax(mx[u], s);
      }
    }
  for (int j = 0; j < R; ++j) {
    const double* kj = kv + j * 32 + hd * 8;
    double k[8], v[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) { k[d] = kj[d]; v[d] = kj[16 + d]; }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      double s = 0.0;
#pragma unroll
      for (int d = 0; d < 8; ++d) s += q[u][d] * k[d];
      const double p = padded ? 1.0 : exp(s - mx[u]);
      sum[u] += p;
#pragma unroll
      for (int d = 0; d < 8; ++d) acc[u][d] += p * v[d];
    }
  }
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int i = rg + 128 * u;
    if (i < R) {
      double* orow = ctx + ((long)i * C + c) * 64 + 16 * hy + 8 * hd;
      const double inv = 1.0 / sum[u];
#pragma unroll
      for (int d = 0; d < 8; ++d) orow[d] = acc[u][d] * inv;
    }
  }
}

// the embeddings leave the fp64 encoder rounded to fp32 once
__global__ void k64_store_f32(const double* __restrict__ x, float* __restrict__ out, long n) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i + 1 < n) {
    const f64x2 v = *reinterpret_cast<const f64x2*>(x + i);
    *reinterpret_cast<f32x2*>(out + i) = (f32x2){(float)v[0], (float)v[1]};
  } else if (i < n) {
    out[i] = (float)x[i];
  }
}
