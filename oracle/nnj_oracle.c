/*
 * nnj_oracle.c -- CPU restatement of the NeuralNJ Argmax hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle and the timed CPU baseline
 * ("port") of bench.py.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product (neuralnj_amd/, libnnj_hip.so) never
 * does.  Parity status: PINNED -- checked in tests/test_oracle_golden.py against
 * vectors captured from the reference itself (tests/golden/gen_golden.py).
 *
 * It follows the reference's algorithm literally (same op order, no algebraic
 * refactoring), each function citing the reference file:line it restates.  Compile
 * with -DREAL=float (default, the reference's precision) or -DREAL=double (a
 * higher-precision arbiter for near-tie diagnostics).  API arrays are always fp32.
 *
 * Layouts: state [B,n,C,D]; onehot [B,T,L,V]; mask [B,L] (1 = padded site);
 * logits [B,P(n)] in itertools.combinations order.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define NNJO_OK 0
#define NNJO_ERR_ARG (-1)
#define NNJO_ERR_NO_WEIGHTS (-3)

typedef struct {
  int32_t vocab_size, patch_size, embed_dim, num_heads, num_layers, device;
} nnjo_config;

typedef struct {
  /* all weight matrices stored TRANSPOSED [in][out] for axpy-style inner loops */
  real *Wk, *bk, *Wv, *bv, *Wq, *bq, *Wo, *bo, *ln_w, *ln_b;
} attn_w;
typedef struct {
  attn_w row, col;
  real *W1, *b1, *W2, *b2, *ln_w, *ln_b;
} layer_w;

typedef struct nnjo_handle {
  nnjo_config cfg;
  int D, H, F, V, K, nl;
  int have_w;
  layer_w* layers;
  real *E0, *e0, *E2, *e2;           /* embed.0, embed.2 */
  real *Wh, *bh, *Wg, *bg, *Wgq, *bgq, *Wgk, *bgk; /* aggregate layer */
  real *S0, *s0, *s2w, *s2b;         /* s_out.0, s_out.2 */
  char err[256];
} nnjo_handle;

static char g_err[256];

/* ------------------------------------------------------------------ utils */
static real* ralloc(size_t n) {
  real* p = (real*)malloc((n ? n : 1) * sizeof(real));
  if (!p) { fprintf(stderr, "nnj_oracle: out of memory (%zu reals)\n", n); abort(); }
  return p;
}
static inline size_t npairs(int n) { return (size_t)n * (size_t)(n - 1) / 2; }
/* flat index of pair (i,j), i<j, among combinations(range(n),2)
 * (reference environment.py:457-462) */
static inline int64_t pair_index(int n, int i, int j) {
  return (int64_t)i * n - (int64_t)i * (i + 1) / 2 + (j - i - 1);
}
static inline real gelu_erf(real x) { /* nn.GELU() default = exact erf form */
  return (real)0.5 * x * ((real)1 + (real)erf((double)x * 0.70710678118654752440));
}
static inline real sigmoidr(real x) { return (real)1 / ((real)1 + (real)exp(-(double)x)); }

/* out[r][:] = b + in[r][:] @ Wt   (Wt is [in_dim][out_dim]) */
static void linear_rows(real* out, const real* in, const real* Wt, const real* b,
                        size_t rows, int in_dim, int out_dim) {
#pragma omp parallel for schedule(static)
  for (size_t r = 0; r < rows; ++r) {
    real* o = out + r * (size_t)out_dim;
    const real* x = in + r * (size_t)in_dim;
    if (b) for (int j = 0; j < out_dim; ++j) o[j] = b[j];
    else for (int j = 0; j < out_dim; ++j) o[j] = 0;
    for (int i = 0; i < in_dim; ++i) {
      const real xi = x[i];
      const real* w = Wt + (size_t)i * out_dim;
      for (int j = 0; j < out_dim; ++j) o[j] += xi * w[j];
    }
  }
}
/* same, single row, serial (for use inside parallel regions) */
static inline void linear_one(real* o, const real* x, const real* Wt, const real* b,
                              int in_dim, int out_dim) {
  if (b) for (int j = 0; j < out_dim; ++j) o[j] = b[j];
  else for (int j = 0; j < out_dim; ++j) o[j] = 0;
  for (int i = 0; i < in_dim; ++i) {
    const real xi = x[i];
    const real* w = Wt + (size_t)i * out_dim;
    for (int j = 0; j < out_dim; ++j) o[j] += xi * w[j];
  }
}

/* torch.nn.LayerNorm(D), eps 1e-5, biased variance (reference msa_modules.py:107) */
static inline void layer_norm_one(real* y, const real* x, const real* w, const real* b, int D) {
  real mean = 0;
  for (int d = 0; d < D; ++d) mean += x[d];
  mean /= (real)D;
  real var = 0;
  for (int d = 0; d < D; ++d) { real t = x[d] - mean; var += t * t; }
  var /= (real)D;
  const real inv = (real)1 / (real)sqrt((double)var + 1e-5);
  for (int d = 0; d < D; ++d) y[d] = (x[d] - mean) * inv * w[d] + b[d];
}

/* ------------------------------------------------------------- lifecycle */
int nnjo_abi_version(void) { return 1; }

int nnjo_num_params(const nnjo_config* c, size_t* n) {
  if (!c || !n) return NNJO_ERR_ARG;
  size_t D = c->embed_dim, F = 4 * D, V = c->vocab_size, K = c->patch_size;
  size_t per_layer = 2 * (4 * (D * D + D) + 2 * D) + (F * D + F) + (D * F + D) + 2 * D;
  *n = per_layer * c->num_layers + (D * V * K + D) + (D * D + D) + 4 * (D * D + D) +
       (D * D + D) + (D + 1);
  return NNJO_OK;
}

int nnjo_create(const nnjo_config* cfg, nnjo_handle** out) {
  if (!cfg || !out) { snprintf(g_err, sizeof g_err, "nnjo_create: null argument"); return NNJO_ERR_ARG; }
  if (cfg->embed_dim <= 0 || cfg->num_heads <= 0 || cfg->embed_dim % cfg->num_heads ||
      cfg->num_layers < 0 || cfg->vocab_size <= 0 || cfg->patch_size <= 0) {
    snprintf(g_err, sizeof g_err, "nnjo_create: bad config");
    return NNJO_ERR_ARG;
  }
  nnjo_handle* h = (nnjo_handle*)calloc(1, sizeof *h);
  h->cfg = *cfg;
  h->D = cfg->embed_dim; h->H = cfg->num_heads; h->F = 4 * cfg->embed_dim;
  h->V = cfg->vocab_size; h->K = cfg->patch_size; h->nl = cfg->num_layers;
  h->layers = (layer_w*)calloc((size_t)(h->nl ? h->nl : 1), sizeof(layer_w));
  *out = h;
  return NNJO_OK;
}

static void free_attn(attn_w* a) {
  free(a->Wk); free(a->bk); free(a->Wv); free(a->bv); free(a->Wq); free(a->bq);
  free(a->Wo); free(a->bo); free(a->ln_w); free(a->ln_b);
}
int nnjo_destroy(nnjo_handle* h) {
  if (!h) return NNJO_OK;
  if (h->have_w) {
    for (int l = 0; l < h->nl; ++l) {
      free_attn(&h->layers[l].row); free_attn(&h->layers[l].col);
      free(h->layers[l].W1); free(h->layers[l].b1); free(h->layers[l].W2);
      free(h->layers[l].b2); free(h->layers[l].ln_w); free(h->layers[l].ln_b);
    }
    free(h->E0); free(h->e0); free(h->E2); free(h->e2);
    free(h->Wh); free(h->bh); free(h->Wg); free(h->bg); free(h->Wgq); free(h->bgq);
    free(h->Wgk); free(h->bgk); free(h->S0); free(h->s0); free(h->s2w); free(h->s2b);
  }
  free(h->layers);
  free(h);
  return NNJO_OK;
}
const char* nnjo_last_error(const nnjo_handle* h) { return h ? h->err : g_err; }

/* read weight[out][in] from the packed stream, store transposed [in][out] */
static real* take_wT(const float** p, int out_dim, int in_dim) {
  real* w = ralloc((size_t)out_dim * in_dim);
  for (int o = 0; o < out_dim; ++o)
    for (int i = 0; i < in_dim; ++i) w[(size_t)i * out_dim + o] = (real)(*p)[(size_t)o * in_dim + i];
  *p += (size_t)out_dim * in_dim;
  return w;
}
static real* take_v(const float** p, int n) {
  real* v = ralloc((size_t)n);
  for (int i = 0; i < n; ++i) v[i] = (real)(*p)[i];
  *p += n;
  return v;
}
static void take_attn(const float** p, attn_w* a, int D) {
  /* state_dict order: k_proj, v_proj, q_proj, out_proj, layer_norm
   * (reference axial_attention.py:24-28, msa_modules.py:107) */
  a->Wk = take_wT(p, D, D); a->bk = take_v(p, D);
  a->Wv = take_wT(p, D, D); a->bv = take_v(p, D);
  a->Wq = take_wT(p, D, D); a->bq = take_v(p, D);
  a->Wo = take_wT(p, D, D); a->bo = take_v(p, D);
  a->ln_w = take_v(p, D);   a->ln_b = take_v(p, D);
}

int nnjo_load_weights(nnjo_handle* h, const float* packed, size_t n) {
  if (!h || !packed) return NNJO_ERR_ARG;
  size_t need; nnjo_num_params(&h->cfg, &need);
  if (n != need) { snprintf(h->err, sizeof h->err, "load_weights: got %zu floats, need %zu", n, need); return NNJO_ERR_ARG; }
  if (h->have_w) { snprintf(h->err, sizeof h->err, "load_weights: already loaded"); return NNJO_ERR_ARG; }
  const float* p = packed;
  const int D = h->D, F = h->F;
  for (int l = 0; l < h->nl; ++l) {
    layer_w* L = &h->layers[l];
    take_attn(&p, &L->row, D);
    take_attn(&p, &L->col, D);
    L->W1 = take_wT(&p, F, D); L->b1 = take_v(&p, F);
    L->W2 = take_wT(&p, D, F); L->b2 = take_v(&p, D);
    L->ln_w = take_v(&p, D);   L->ln_b = take_v(&p, D);
  }
  h->E0 = take_wT(&p, D, h->V * h->K); h->e0 = take_v(&p, D);
  h->E2 = take_wT(&p, D, D);           h->e2 = take_v(&p, D);
  h->Wh = take_wT(&p, D, D);  h->bh = take_v(&p, D);   /* h_linear_last */
  h->Wg = take_wT(&p, D, D);  h->bg = take_v(&p, D);   /* g_linear_last */
  h->Wgq = take_wT(&p, D, D); h->bgq = take_v(&p, D);  /* g_attn_q */
  h->Wgk = take_wT(&p, D, D); h->bgk = take_v(&p, D);  /* g_attn_k */
  h->S0 = take_wT(&p, D, D);  h->s0 = take_v(&p, D);   /* s_out.0 */
  h->s2w = take_v(&p, D);     h->s2b = take_v(&p, 1);  /* s_out.2 */
  if ((size_t)(p - packed) != need) { snprintf(h->err, sizeof h->err, "load_weights: internal size mismatch"); return NNJO_ERR_ARG; }
  h->have_w = 1;
  return NNJO_OK;
}

/* --------------------------------------------------------------- encoder */
/* All encoder tensors here are [B][R][C][D] (b-major).  The reference works in
 * [R,C,B,D] (model.py:81); every op is independent per b, so only indexing differs. */

/* RowSelfAttention (tied across rows) -- reference axial_attention.py:6-138.
 * y = LayerNorm(x) already applied; returns out_proj(context) added to x. */
static void row_attention(const nnjo_handle* h, const attn_w* w, real* x, const uint8_t* maskC,
                          int B, int R, int C) {
  const int D = h->D, H = h->H, dh = D / H;
  const size_t N = (size_t)R * C;
  /* align_scaling: head_dim^-0.5 / sqrt(num_rows)  (axial_attention.py:31-33) */
  const real scaling = (real)(pow((double)dh, -0.5) / sqrt((double)R));
  /* no-grad chunking (axial_attention.py:35-64,127-128): when R*C > max_tokens_per_msa (1024, model.py:34)
   * the logits are computed per chunk of max_rows rows -- each chunk's einsum, then its own
   * masked_fill(-10000) -- and the chunks are ADDED (`attns += attn_weights`, :52).  Restated in that
   * order: the partial sums of a chunk (max_rows*dh terms) are rounded before they meet the running total,
   * which is also what keeps the 8R-term sum at fp32 accuracy for 100 rows and more. */
  int max_rows = R;
  if ((long)R * C > 1024) { max_rows = 1024 / C; if (max_rows < 1) max_rows = 1; }

  real* y = ralloc(N * D);
  real* q = ralloc(N * D);
  real* k = ralloc(N * D);
  real* v = ralloc(N * D);
  real* ctx = ralloc(N * D);
  real* out = ralloc(N * D);
  const int RD = R * dh;
  real* Qh = ralloc((size_t)C * RD);       /* [i][(r,d)] */
  real* KhT = ralloc((size_t)RD * C);      /* [(r,d)][j] */
  real* Vh = ralloc((size_t)C * RD);       /* [j][(r,d)] */
  real* P = ralloc((size_t)C * C);

  for (int b = 0; b < B; ++b) {
    real* xb = x + (size_t)b * N * D;
    const uint8_t* mb = maskC ? maskC + (size_t)b * C : NULL;
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < N; ++t) layer_norm_one(y + t * D, xb + t * D, w->ln_w, w->ln_b, D);
    linear_rows(q, y, w->Wq, w->bq, N, D, D);
    linear_rows(k, y, w->Wk, w->bk, N, D, D);
    linear_rows(v, y, w->Wv, w->bv, N, D, D);
    /* q *= scaling; q *= 1 - padding_mask  (axial_attention.py:77-82) */
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < N; ++t) {
      const int c = (int)(t % C);
      const real m = (mb && mb[c]) ? (real)0 : (real)1;
      for (int d = 0; d < D; ++d) q[t * D + d] = q[t * D + d] * scaling * m;
    }
    for (int hh = 0; hh < H; ++hh) {
#pragma omp parallel for schedule(static)
      for (int c = 0; c < C; ++c)
        for (int r = 0; r < R; ++r)
          for (int d = 0; d < dh; ++d) {
            const size_t src = ((size_t)r * C + c) * D + hh * dh + d;
            Qh[(size_t)c * RD + r * dh + d] = q[src];
            KhT[(size_t)(r * dh + d) * C + c] = k[src];
            Vh[(size_t)c * RD + r * dh + d] = v[src];
          }
      /* attn_weights[h,b,i,j] = sum_{r,d} q[r,i,b,h,d] k[r,j,b,h,d]  (axial_attention.py:97) */
#pragma omp parallel for schedule(static)
      for (int i = 0; i < C; ++i) {
        real* Pi = P + (size_t)i * C;
        real Pc[C];
        for (int j = 0; j < C; ++j) Pi[j] = 0;
        for (int r0 = 0; r0 < R; r0 += max_rows) {                /* one chunk of rows (axial_attention.py:45) */
          const int r1 = r0 + max_rows < R ? r0 + max_rows : R;
          for (int j = 0; j < C; ++j) Pc[j] = 0;
          for (int e = r0 * dh; e < r1 * dh; ++e) {
            const real qe = Qh[(size_t)i * RD + e];
            const real* kr = KhT + (size_t)e * C;
            for (int j = 0; j < C; ++j) Pc[j] += qe * kr[j];
          }
          /* masked_fill(padding_mask[:,0], -10000) on key positions, per chunk (axial_attention.py:99-103) */
          if (mb) for (int j = 0; j < C; ++j) if (mb[j]) Pc[j] = (real)-10000.0;
          for (int j = 0; j < C; ++j) Pi[j] += Pc[j];              /* attns += attn_weights (:52) */
        }
        /* softmax over j (axial_attention.py:54 / 132) */
        real mx = Pi[0];
        for (int j = 1; j < C; ++j) if (Pi[j] > mx) mx = Pi[j];
        /* The two reductions over the C keys (softmax denominator, context) use a wide accumulator and are
         * rounded to `real` once: the reference's softmax / bmm kernels reduce in blocked, vectorised order,
         * whose rounding error is far below that of a sequential fp32 sum of C = 1024 terms (measured on the
         * 50 x 1024 fixtures: 1.0e-4 of the score scale for the sequential sum against 3e-5 for the
         * reference's own tables, both against the fp64 build).  The checker must not be noisier than the
         * code it restates; every stored tensor stays `real`. */
        double s = 0;
        for (int j = 0; j < C; ++j) { Pi[j] = (real)exp((double)(Pi[j] - mx)); s += (double)Pi[j]; }
        const real inv = (real)(1.0 / s);
        for (int j = 0; j < C; ++j) Pi[j] *= inv;
        /* context[r,i,b,h,d] = sum_j P[h,b,i,j] v[r,j,b,h,d]  (axial_attention.py:114) */
        double acc[RD];
        for (int e = 0; e < RD; ++e) acc[e] = 0;
        for (int j = 0; j < C; ++j) {
          const real pj = Pi[j];
          const real* vr = Vh + (size_t)j * RD;
          for (int e = 0; e < RD; ++e) acc[e] += (double)pj * (double)vr[e];   /* unrounded product, as an FMA */
        }
        for (int r = 0; r < R; ++r)
          for (int d = 0; d < dh; ++d)
            ctx[((size_t)r * C + i) * D + hh * dh + d] = (real)acc[r * dh + d];
      }
    }
    linear_rows(out, ctx, w->Wo, w->bo, N, D, D);
    /* residual (msa_modules.py:109-125; dropout is identity in eval) */
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < N * D; ++t) xb[t] += out[t];
  }
  free(y); free(q); free(k); free(v); free(ctx); free(out); free(Qh); free(KhT); free(Vh); free(P);
}

/* ColumnSelfAttention -- reference axial_attention.py:141-255. */
static void col_attention(const nnjo_handle* h, const attn_w* w, real* x, const uint8_t* maskC,
                          int B, int R, int C) {
  const int D = h->D, H = h->H, dh = D / H;
  const size_t N = (size_t)R * C;
  const real scaling = (real)pow((double)dh, -0.5); /* axial_attention.py:214 */
#pragma omp parallel
  {
    real* y = ralloc((size_t)R * D);
    real* q = ralloc((size_t)R * D);
    real* k = ralloc((size_t)R * D);
    real* v = ralloc((size_t)R * D);
    real* ctx = ralloc((size_t)R * D);
    real* o = ralloc((size_t)D);
    real* p = ralloc((size_t)R);
#pragma omp for schedule(static) collapse(2)
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c) {
        real* xb = x + (size_t)b * N * D;
        const int padded = maskC && maskC[(size_t)b * C + c];
        for (int r = 0; r < R; ++r) {
          layer_norm_one(y + (size_t)r * D, xb + ((size_t)r * C + c) * D, w->ln_w, w->ln_b, D);
          linear_one(v + (size_t)r * D, y + (size_t)r * D, w->Wv, w->bv, D, D);
        }
        if (R == 1) {
          /* single position shortcut (axial_attention.py:198-209) */
          for (int d = 0; d < D; ++d) ctx[d] = v[d];
        } else {
          for (int r = 0; r < R; ++r) {
            linear_one(q + (size_t)r * D, y + (size_t)r * D, w->Wq, w->bq, D, D);
            linear_one(k + (size_t)r * D, y + (size_t)r * D, w->Wk, w->bk, D, D);
            for (int d = 0; d < D; ++d) q[(size_t)r * D + d] *= scaling;
          }
          for (int hh = 0; hh < H; ++hh)
            for (int i = 0; i < R; ++i) {
              /* einsum("icnhd,jcnhd->hcnij") (axial_attention.py:216) */
              for (int j = 0; j < R; ++j) {
                real s = 0;
                for (int d = 0; d < dh; ++d) s += q[(size_t)i * D + hh * dh + d] * k[(size_t)j * D + hh * dh + d];
                /* masked_fill(padding_mask, -10000): every key of a padded column (axial_attention.py:220-224) */
                p[j] = padded ? (real)-10000.0 : s;
              }
              real mx = p[0];
              for (int j = 1; j < R; ++j) if (p[j] > mx) mx = p[j];
              real sum = 0;
              for (int j = 0; j < R; ++j) { p[j] = (real)exp((double)(p[j] - mx)); sum += p[j]; }
              const real inv = (real)1 / sum;
              /* einsum("hcnij,jcnhd->icnhd") (axial_attention.py:234) */
              for (int d = 0; d < dh; ++d) {
                real acc = 0;
                for (int j = 0; j < R; ++j) acc += (p[j] * inv) * v[(size_t)j * D + hh * dh + d];
                ctx[(size_t)i * D + hh * dh + d] = acc;
              }
            }
        }
        for (int r = 0; r < R; ++r) {
          linear_one(o, ctx + (size_t)r * D, w->Wo, w->bo, D, D);
          real* xr = xb + ((size_t)r * C + c) * D;
          for (int d = 0; d < D; ++d) xr[d] += o[d];
        }
      }
    free(y); free(q); free(k); free(v); free(ctx); free(o); free(p);
  }
}

/* FeedForwardNetwork inside NormalizedResidualBlock -- reference msa_modules.py:109-151 */
static void ffn_block(const nnjo_handle* h, const layer_w* L, real* x, size_t tokens) {
  const int D = h->D, F = h->F;
#pragma omp parallel
  {
    real* y = ralloc((size_t)D);
    real* u = ralloc((size_t)F);
    real* o = ralloc((size_t)D);
#pragma omp for schedule(static)
    for (size_t t = 0; t < tokens; ++t) {
      real* xt = x + t * D;
      layer_norm_one(y, xt, L->ln_w, L->ln_b, D);
      linear_one(u, y, L->W1, L->b1, D, F);
      for (int f = 0; f < F; ++f) u[f] = gelu_erf(u[f]);
      linear_one(o, u, L->W2, L->b2, F, D);
      for (int d = 0; d < D; ++d) xt[d] += o[d];
    }
    free(y); free(u); free(o);
  }
}

static void copy_out(float* dst, const real* src, size_t n) {
  for (size_t i = 0; i < n; ++i) dst[i] = (float)src[i];
}

/* PhyloATTN.encode_zxr -- reference model.py:67-88.
 * onehot float [B,R,L,V]; mask [B,L] or NULL; state_out float [B,R,C,D].
 * taps: NULL or 4 optional float buffers [B,R,C,D] receiving the tensor after
 * embed, after layer-0 row attention, layer-0 column attention, layer-0 FFN. */
int nnjo_encode(nnjo_handle* h, const float* onehot, const uint8_t* mask, float* state_out,
                int32_t B, int32_t R, int32_t L, float* const* taps) {
  if (!h || !onehot || !state_out || B <= 0 || R <= 0 || L <= 0) return NNJO_ERR_ARG;
  if (!h->have_w) { snprintf(h->err, sizeof h->err, "encode: weights not loaded"); return NNJO_ERR_NO_WEIGHTS; }
  const int D = h->D, V = h->V, K = h->K;
  if (L % K) { snprintf(h->err, sizeof h->err, "encode: L=%d not divisible by patch_size=%d", L, K); return NNJO_ERR_ARG; }
  const int C = L / K; /* patch_num (model.py:72) */
  const size_t N = (size_t)B * R * C;
  real* x = ralloc(N * D);
  /* 'b r (c k) e -> b r c (k e)' then embed = Linear, GELU, Linear (model.py:76-77,39-43) */
#pragma omp parallel
  {
    real* in = ralloc((size_t)V * K);
    real* t1 = ralloc((size_t)D);
#pragma omp for schedule(static)
    for (size_t t = 0; t < N; ++t) {
      const float* src = onehot + t * (size_t)(V * K);
      for (int i = 0; i < V * K; ++i) in[i] = (real)src[i];
      linear_one(t1, in, h->E0, h->e0, V * K, D);
      for (int d = 0; d < D; ++d) t1[d] = gelu_erf(t1[d]);
      linear_one(x + t * D, t1, h->E2, h->e2, D, D);
    }
    free(in); free(t1);
  }
  if (taps && taps[0]) copy_out(taps[0], x, N * D);
  /* batch_seq_mask[:, ::patch_size] repeated over rows (model.py:79) */
  uint8_t* maskC = NULL;
  if (mask) {
    maskC = (uint8_t*)malloc((size_t)B * C);
    for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) maskC[(size_t)b * C + c] = mask[(size_t)b * L + (size_t)c * K] ? 1 : 0;
  }
  for (int l = 0; l < h->nl; ++l) {
    const layer_w* Lw = &h->layers[l];
    /* AxialTransformerLayer.forward: row -> column -> ffn (msa_modules.py:62-91) */
    row_attention(h, &Lw->row, x, maskC, B, R, C);
    if (l == 0 && taps && taps[1]) copy_out(taps[1], x, N * D);
    col_attention(h, &Lw->col, x, maskC, B, R, C);
    if (l == 0 && taps && taps[2]) copy_out(taps[2], x, N * D);
    ffn_block(h, Lw, x, N);
    if (l == 0 && taps && taps[3]) copy_out(taps[3], x, N * D);
  }
  copy_out(state_out, x, N * D);
  free(x); free(maskC);
  return NNJO_OK;
}

/* ------------------------------------------------------- pair scorer core */
/* One pair: PhyloATTN.aggregate (model.py:102-155) and, if `score`, the rest of
 * decode_gg (model.py:90-99).  S = rows of this batch element [n][C][D] (real),
 * kS = g_attn_k(S) [n][C][D] (only if n > 2).  xi,xj point at rows i and j of S.
 * x_out [C][D] may be NULL.  Returns the score (sum over unmasked sites). */
static real pair_core(const nnjo_handle* h, const real* S, const real* kS, int n, int C,
                      int i, int j, const uint8_t* maskC, real* x_out, int score,
                      real* x, real* xg, real* alpha) {
  const int D = h->D;
  const real* xi = S + (size_t)i * C * D;
  const real* xj = S + (size_t)j * C * D;
  real tmp[D], hh[D];
  /* h = h_linear_last(x_i - x_j); z = sigmoid(h); x = z*x_i + (1-z)*x_j (model.py:105-108) */
  for (int c = 0; c < C; ++c) {
    for (int d = 0; d < D; ++d) tmp[d] = xi[(size_t)c * D + d] - xj[(size_t)c * D + d];
    linear_one(hh, tmp, h->Wh, h->bh, D, D);
    for (int d = 0; d < D; ++d) {
      const real z = sigmoidr(hh[d]);
      x[(size_t)c * D + d] = z * xi[(size_t)c * D + d] + ((real)1 - z) * xj[(size_t)c * D + d];
    }
  }
  if (n > 2) { /* model.py:111 */
    /* q = g_attn_q(x); alpha[r] = sum_{c,d} q*k / sqrt(D*patch_num) (model.py:112-118) */
    for (int r = 0; r < n; ++r) alpha[r] = 0;
    for (int c = 0; c < C; ++c) {
      linear_one(tmp, x + (size_t)c * D, h->Wgq, h->bgq, D, D);
      for (int r = 0; r < n; ++r) {
        const real* kr = kS + ((size_t)r * C + c) * D;
        real s = 0;
        for (int d = 0; d < D; ++d) s += tmp[d] * kr[d];
        alpha[r] += s;
      }
    }
    const real inv_scale = (real)(1.0 / sqrt((double)D * (double)C));
    for (int r = 0; r < n; ++r) alpha[r] *= inv_scale;
    /* alpha[..., i] += -inf; alpha[..., j] += -inf; softmax over r (model.py:120-146) */
    real mx = -INFINITY;
    for (int r = 0; r < n; ++r) if (r != i && r != j && alpha[r] > mx) mx = alpha[r];
    real sum = 0;
    for (int r = 0; r < n; ++r) {
      alpha[r] = (r == i || r == j) ? (real)0 : (real)exp((double)(alpha[r] - mx));
      sum += alpha[r];
    }
    for (int r = 0; r < n; ++r) alpha[r] /= sum;
    /* x_global_res = sum_r alpha*v; g = g_linear_last(.); w = sigmoid(g);
     * x = (1-w)*x + w*x_global_res (model.py:148-153) */
    for (int c = 0; c < C; ++c) {
      real* g = xg + (size_t)c * D;
      for (int d = 0; d < D; ++d) g[d] = 0;
      for (int r = 0; r < n; ++r) {
        const real a = alpha[r];
        const real* vr = S + ((size_t)r * C + c) * D;
        for (int d = 0; d < D; ++d) g[d] += a * vr[d];
      }
      linear_one(tmp, g, h->Wg, h->bg, D, D);
      for (int d = 0; d < D; ++d) {
        const real w = sigmoidr(tmp[d]);
        x[(size_t)c * D + d] = ((real)1 - w) * x[(size_t)c * D + d] + w * g[d];
      }
    }
  }
  if (x_out) memcpy(x_out, x, (size_t)C * D * sizeof(real));
  if (!score) return 0;
  /* scores = sum_c seq_mask_c * s_out(x)[c]; s_out = Linear, GELU, Linear(D->1) (model.py:93-97,56-60) */
  real total = 0;
  for (int c = 0; c < C; ++c) {
    linear_one(tmp, x + (size_t)c * D, h->S0, h->s0, D, D);
    real s = h->s2b[0];
    for (int d = 0; d < D; ++d) s += gelu_erf(tmp[d]) * h->s2w[d];
    const real m = (maskC && maskC[c]) ? (real)0 : (real)1;
    total += s * m;
  }
  return total;
}

static real* to_real(const float* src, size_t n) {
  real* p = ralloc(n);
  for (size_t i = 0; i < n; ++i) p[i] = (real)src[i];
  return p;
}

/* Scores a list of pairs of one batch element.  pairs int32 [np][2] (i<=j allowed:
 * the reference also scores the self pair, model.py:186-197). */
static void score_pairs_one(const nnjo_handle* h, const real* S, int n, int C, const uint8_t* maskC,
                            const int32_t* pairs, int np, real* scores_out) {
  const int D = h->D;
  real* kS = NULL;
  if (n > 2) { kS = ralloc((size_t)n * C * D); linear_rows(kS, S, h->Wgk, h->bgk, (size_t)n * C, D, D); }
#pragma omp parallel
  {
    real* x = ralloc((size_t)C * D);
    real* xg = ralloc((size_t)C * D);
    real* alpha = ralloc((size_t)n);
#pragma omp for schedule(dynamic, 1)
    for (int p = 0; p < np; ++p)
      scores_out[p] = pair_core(h, S, kS, n, C, pairs[2 * p], pairs[2 * p + 1], maskC, NULL, 1, x, xg, alpha);
    free(x); free(xg); free(alpha);
  }
  free(kS);
}

static uint8_t* mask_cols(const uint8_t* mask, int B, int L, int K) {
  if (!mask) return NULL;
  const int C = L / K;
  uint8_t* m = (uint8_t*)malloc((size_t)B * C);
  for (int b = 0; b < B; ++b) for (int c = 0; c < C; ++c) m[(size_t)b * C + c] = mask[(size_t)b * L + (size_t)c * K] ? 1 : 0;
  return m;
}

/* decode_zxr, logits_prev is None -- reference model.py:168-181 */
int nnjo_pair_scores_full(nnjo_handle* h, const float* state, const uint8_t* mask, float* logits_out,
                          int32_t B, int32_t n, int32_t L) {
  if (!h || !state || !logits_out || B <= 0 || n < 2) return NNJO_ERR_ARG;
  if (!h->have_w) return NNJO_ERR_NO_WEIGHTS;
  const int D = h->D, C = L / h->K;
  const int np = (int)npairs(n);
  int32_t* pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)np);
  int t = 0;
  for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) { pairs[2 * t] = i; pairs[2 * t + 1] = j; ++t; }
  uint8_t* maskC = mask_cols(mask, B, L, h->K);
  real* sc = ralloc((size_t)np);
  for (int b = 0; b < B; ++b) {
    real* S = to_real(state + (size_t)b * n * C * D, (size_t)n * C * D);
    score_pairs_one(h, S, n, C, maskC ? maskC + (size_t)b * C : NULL, pairs, np, sc);
    for (int p = 0; p < np; ++p) logits_out[(size_t)b * np + p] = (float)sc[p];
    free(S);
  }
  free(sc); free(pairs); free(maskC);
  return NNJO_OK;
}

/* utils.get_score_indices_to_prev for one batch element -- reference utils.py:213-251.
 * n = current number of rows (nb_seq); (ip,jp) = previous merge; out[P(n)] indexes
 * cat(logits_prev[P(n+1)], new_scores[n]). */
void nnjo_index_map_one(int32_t n, int32_t ip, int32_t jp, int64_t* out) {
  const int64_t len_prev = (int64_t)npairs(n + 1);
  int64_t t = 0;
  for (int ii = 0; ii < n; ++ii)
    for (int jj = ii + 1; jj < n; ++jj, ++t) {
      int64_t v;
      if (ii < ip) {
        if (jj < ip) v = pair_index(n + 1, ii, jj);
        else if (jj == ip) v = len_prev + ii;
        else if (jj < jp) v = pair_index(n + 1, ii, jj);
        else v = pair_index(n + 1, ii, jj + 1);
      } else if (ii == ip) {
        v = len_prev + jj;
      } else if (ii < jp) {
        if (jj < jp) v = pair_index(n + 1, ii, jj);
        else v = pair_index(n + 1, ii, jj + 1);
      } else {
        v = pair_index(n + 1, ii + 1, jj + 1);
      }
      out[t] = v;
    }
}

int nnjo_score_index_map(const int32_t* ij_prev, int64_t* idx_out, int32_t B, int32_t n) {
  if (!ij_prev || !idx_out || n < 2) return NNJO_ERR_ARG;
  const size_t np = npairs(n);
  for (int b = 0; b < B; ++b) nnjo_index_map_one(n, ij_prev[2 * b], ij_prev[2 * b + 1], idx_out + (size_t)b * np);
  return NNJO_OK;
}

/* decode_zxr, logits_prev given -- reference model.py:184-201: scores the n pairs
 * sort(i_prev, r), r = 0..n-1, then gather(cat(logits_prev, new), idx_map). */
int nnjo_pair_scores_incr(nnjo_handle* h, const float* state, const uint8_t* mask,
                          const int32_t* ij_prev, const float* logits_prev, float* logits_out,
                          int32_t B, int32_t n, int32_t L, float* new_scores_out /* [B,n] or NULL */) {
  if (!h || !state || !ij_prev || !logits_prev || !logits_out || B <= 0 || n < 2) return NNJO_ERR_ARG;
  if (!h->have_w) return NNJO_ERR_NO_WEIGHTS;
  const int D = h->D, C = L / h->K;
  const size_t np = npairs(n), np_prev = npairs(n + 1);
  uint8_t* maskC = mask_cols(mask, B, L, h->K);
  int32_t* pairs = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)n);
  real* sc = ralloc((size_t)n);
  int64_t* idx = (int64_t*)malloc(sizeof(int64_t) * np);
  for (int b = 0; b < B; ++b) {
    const int ip = ij_prev[2 * b];
    for (int r = 0; r < n; ++r) { pairs[2 * r] = r < ip ? r : ip; pairs[2 * r + 1] = r < ip ? ip : r; }
    real* S = to_real(state + (size_t)b * n * C * D, (size_t)n * C * D);
    score_pairs_one(h, S, n, C, maskC ? maskC + (size_t)b * C : NULL, pairs, n, sc);
    free(S);
    if (new_scores_out) for (int r = 0; r < n; ++r) new_scores_out[(size_t)b * n + r] = (float)sc[r];
    nnjo_index_map_one(n, ip, ij_prev[2 * b + 1], idx);
    for (size_t p = 0; p < np; ++p) {
      const int64_t s = idx[p];
      logits_out[(size_t)b * np + p] = s < (int64_t)np_prev ? logits_prev[(size_t)b * np_prev + s]
                                                            : (float)sc[s - (int64_t)np_prev];
    }
  }
  free(idx); free(sc); free(pairs); free(maskC);
  return NNJO_OK;
}

/* agent.aggregate(subtree_i, subtree_j, (ii,jj), batchwise_ij_indices=True) as called
 * by PhyInferEnv.step -- reference environment.py:822-831, model.py:102-155 with the
 * context rows stashed by the preceding decode_zxr (model.py:166). out [B,1,C,D]. */
int nnjo_aggregate(nnjo_handle* h, const float* state, const int32_t* ij, float* out_row,
                   int32_t B, int32_t n, int32_t L) {
  if (!h || !state || !ij || !out_row || B <= 0 || n < 2) return NNJO_ERR_ARG;
  if (!h->have_w) return NNJO_ERR_NO_WEIGHTS;
  const int D = h->D, C = L / h->K;
  for (int b = 0; b < B; ++b) {
    real* S = to_real(state + (size_t)b * n * C * D, (size_t)n * C * D);
    real* kS = NULL;
    if (n > 2) { kS = ralloc((size_t)n * C * D); linear_rows(kS, S, h->Wgk, h->bgk, (size_t)n * C, D, D); }
    real* x = ralloc((size_t)C * D);
    real* xg = ralloc((size_t)C * D);
    real* alpha = ralloc((size_t)n);
    real* xo = ralloc((size_t)C * D);
    pair_core(h, S, kS, n, C, ij[2 * b], ij[2 * b + 1], NULL, xo, 0, x, xg, alpha);
    copy_out(out_row + (size_t)b * C * D, xo, (size_t)C * D);
    free(S); free(kS); free(x); free(xg); free(alpha); free(xo);
  }
  return NNJO_OK;
}

/* Tensor half of PhyInferEnv.step -- reference environment.py:760-835:
 * base_indices = [0..n-1] with slot i -> n (the appended merged row), slot j popped. */
int nnjo_env_step(nnjo_handle* h, const float* state, const int32_t* ij, float* state_out,
                  int32_t B, int32_t n, int32_t L) {
  if (!h || !state || !ij || !state_out || n < 3) return NNJO_ERR_ARG;
  const int D = h->D, C = L / h->K;
  const size_t row = (size_t)C * D;
  float* merged = (float*)malloc(sizeof(float) * (size_t)B * row);
  int rc = nnjo_aggregate(h, state, ij, merged, B, n, L);
  if (rc) { free(merged); return rc; }
  for (int b = 0; b < B; ++b) {
    const int i = ij[2 * b], j = ij[2 * b + 1];
    int t = 0;
    for (int r = 0; r < n; ++r) {
      if (r == j) continue;
      const float* src = (r == i) ? merged + (size_t)b * row : state + ((size_t)b * n + r) * row;
      memcpy(state_out + ((size_t)b * (n - 1) + t) * row, src, row * sizeof(float));
      ++t;
    }
  }
  free(merged);
  return NNJO_OK;
}

/* argmax(logits, -1), first maximal index; flat -> (i,j) via combinations order
 * -- reference finetune_rl_search.py:145,159-160, environment.py:457-462 */
int nnjo_select_pair(const float* logits, int32_t* ij_out, float* top2_gap, int32_t B, int32_t n) {
  if (!logits || !ij_out || n < 2) return NNJO_ERR_ARG;
  const size_t np = npairs(n);
  for (int b = 0; b < B; ++b) {
    const float* l = logits + (size_t)b * np;
    size_t best = 0;
    for (size_t p = 1; p < np; ++p) if (l[p] > l[best]) best = p;
    float second = -INFINITY;
    for (size_t p = 0; p < np; ++p) if (p != best && l[p] > second) second = l[p];
    if (top2_gap) top2_gap[b] = np > 1 ? l[best] - second : 0.0f;
    size_t t = 0; int fi = 0, fj = 1;
    for (int i = 0; i < n; ++i) {
      const size_t cnt = (size_t)(n - 1 - i);
      if (best < t + cnt) { fi = i; fj = i + 1 + (int)(best - t); break; }
      t += cnt;
    }
    ij_out[2 * b] = fi; ij_out[2 * b + 1] = fj;
  }
  return NNJO_OK;
}

/* Categorical(logits / temperature).sample() by inverse CDF on a supplied uniform (the reference's
 * RNG stream, finetune_rl_search.py:147, is not reproducible across devices; the stream is an input):
 * smallest k with sum_{p<=k} e_p > u * sum_p e_p, e_p = exp((l_p - max) / temperature), fp64, flat order. */
static size_t sample_index(const float* l, size_t np, float u, float inv_temp) {
  float mx = l[0];
  for (size_t p = 1; p < np; ++p) if (l[p] > mx) mx = l[p];
  double total = 0.0;
  for (size_t p = 0; p < np; ++p) total += exp((double)(l[p] - mx) * (double)inv_temp);
  const double target = (double)u * total;
  double run = 0.0;
  for (size_t p = 0; p < np; ++p) {
    run += exp((double)(l[p] - mx) * (double)inv_temp);
    if (run > target) return p;
  }
  return np - 1;
}

/* reinforce_rollout, eval branch -- reference finetune_rl_search.py:78-189.
 * Argmax (uniforms == NULL, :145) or sampling (:147) with caller-supplied uniforms [B,T-1].
 * forced_merges: NULL or int32 [B,T-1,2] applied instead of the chosen pair.
 * merges_out int32 [B,T-1,2] = the chosen pair of each step (argmax or sample);
 * logits_trace: NULL or float [B, sum_{n=T..2} P(n)]; top2_gap: NULL or [B,T-1];
 * state_out: NULL or [B,T,C,D] encoder output. */
static int rollout_impl(nnjo_handle* h, const float* onehot, const uint8_t* mask,
                        int32_t B, int32_t T, int32_t L, const int32_t* forced_merges,
                        const float* uniforms, float inv_temp,
                        int32_t* merges_out, float* logits_trace, float* top2_gap, float* state_out) {
  if (!h || !onehot || !merges_out || B <= 0 || T < 2) return NNJO_ERR_ARG;
  if (!h->have_w) return NNJO_ERR_NO_WEIGHTS;
  const int D = h->D, C = L / h->K;
  const size_t row = (size_t)C * D;
  float* state = (float*)malloc(sizeof(float) * (size_t)B * T * row);
  float* state2 = (float*)malloc(sizeof(float) * (size_t)B * T * row);
  int rc = nnjo_encode(h, onehot, mask, state, B, T, L, NULL); /* :108-112 */
  if (rc) { free(state); free(state2); return rc; }
  if (state_out) memcpy(state_out, state, sizeof(float) * (size_t)B * T * row);
  size_t total = 0;
  for (int n = T; n >= 2; --n) total += npairs(n);
  float* logits = (float*)malloc(sizeof(float) * npairs(T) * (size_t)B);
  float* logits_prev = (float*)malloc(sizeof(float) * npairs(T) * (size_t)B);
  int32_t* ij = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)B);
  int32_t* ij_apply = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)B);
  float* gap = (float*)malloc(sizeof(float) * (size_t)B);
  size_t off = 0;
  for (int step = 0, n = T; n >= 2; ++step, --n) {
    const size_t np = npairs(n);
    if (step == 0) rc = nnjo_pair_scores_full(h, state, mask, logits, B, n, L);         /* :126, model.py:168-181 */
    else rc = nnjo_pair_scores_incr(h, state, mask, ij_apply, logits_prev, logits, B, n, L, NULL); /* :121-126 */
    if (rc) break;
    nnjo_select_pair(logits, ij, gap, B, n);                                             /* :145,159-160 */
    for (int b = 0; b < B; ++b) {
      if (uniforms) {                                                                    /* :147 */
        size_t k = sample_index(logits + (size_t)b * np, np, uniforms[(size_t)b * (T - 1) + step], inv_temp);
        size_t t = 0;
        for (int i = 0; i < n; ++i) {
          const size_t cnt = (size_t)(n - 1 - i);
          if (k < t + cnt) { ij[2 * b] = i; ij[2 * b + 1] = i + 1 + (int)(k - t); break; }
          t += cnt;
        }
      }
      merges_out[((size_t)b * (T - 1) + step) * 2] = ij[2 * b];
      merges_out[((size_t)b * (T - 1) + step) * 2 + 1] = ij[2 * b + 1];
      if (top2_gap) top2_gap[(size_t)b * (T - 1) + step] = gap[b];
      if (logits_trace) memcpy(logits_trace + (size_t)b * total + off, logits + (size_t)b * np, sizeof(float) * np);
      for (int k = 0; k < 2; ++k)
        ij_apply[2 * b + k] = forced_merges ? forced_merges[((size_t)b * (T - 1) + step) * 2 + k] : ij[2 * b + k];
    }
    off += np;
    if (n > 2) {                                                                          /* env.step :164 */
      rc = nnjo_env_step(h, state, ij_apply, state2, B, n, L);
      if (rc) break;
      float* t = state; state = state2; state2 = t;
    }
    { float* t = logits; logits = logits_prev; logits_prev = t; }                         /* :175 */
  }
  free(state); free(state2); free(logits); free(logits_prev); free(ij); free(ij_apply); free(gap);
  return rc;
}

int nnjo_rollout_argmax(nnjo_handle* h, const float* onehot, const uint8_t* mask,
                        int32_t B, int32_t T, int32_t L, const int32_t* forced_merges,
                        int32_t* merges_out, float* logits_trace, float* top2_gap, float* state_out) {
  return rollout_impl(h, onehot, mask, B, T, L, forced_merges, NULL, 1.0f, merges_out, logits_trace, top2_gap, state_out);
}

/* eval, argmax=False branch (RL_Search): onehot [B,T,L,V] holds the B alignments (replicas of one
 * alignment are simply repeated by the caller); uniforms float [B,T-1]. */
int nnjo_rollout_sample(nnjo_handle* h, const float* onehot, const uint8_t* mask,
                        int32_t B, int32_t T, int32_t L, const float* uniforms, float temperature,
                        int32_t* merges_out, float* logits_trace) {
  if (!uniforms || !(temperature > 0.f)) return NNJO_ERR_ARG;
  return rollout_impl(h, onehot, mask, B, T, L, NULL, uniforms, 1.0f / temperature, merges_out, logits_trace, NULL, NULL);
}

int nnjo_set_threads(int32_t n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}
