// nnj_api.hip -- C ABI (include/nnj.h) of libnnj_hip.so: handle, weights, workspace
// carving and the launch sequences of the encoder and the neural NJ loop.
// gfx950 only; no CPU fallback: every compute entry point launches HIP kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <initializer_list>
#include <vector>

#include "../../include/nnj.h"
#include "nnj_encoder.hpp"
#include "nnj_encoder64.hpp"
#include "nnj_rowattn.hpp"
#include "nnj_scorer.hpp"
#include "nnj_scorer16.hpp"
#include "nnj_scorer_wide.hpp"

// nnj_step0_tu.hip: the two all-pairs kernels of step 0, compiled with a scheduling strategy of their own
hipError_t nnj_launch_pair_alpha_1_8(unsigned grid, size_t lds, hipStream_t st, const void* rowset, const void* scorerw,
                                     const int* ij_prev, float* alpha_part, int mode, int n, int C, int npairs, int ppad,
                                     int cs, int nsc, int npg, int B);
hipError_t nnj_launch_pair_score_1_8(unsigned grid, size_t lds, hipStream_t st, const void* rowset, const void* scorerw,
                                     const int* ij_prev, const float* alpha, const uint8_t* mask, float* score_part, int mode,
                                     int n, int C, int npairs, int ppad, int cs, int has_ctx, int nsc, int npg, int B);
#include "nnj_step_g.hpp"
#include "nnj_likelihood.hpp"

namespace {

enum ProfKind {
  PK_EMBED = 0, PK_TOK1, PK_FFN, PK_ROW_XF, PK_PAIR_ALPHA, PK_ALPHA_SOFTMAX,
  PK_PAIR_SCORE, PK_ASSEMBLE, PK_AGG_ALPHA, PK_AGG_FINISH, PK_MISC, PK_PAIR_ALPHA_INCR, PK_PAIR_SCORE_INCR,
  PK_ROW_QKV, PK_ROW_S, PK_ROW_PV, PK_STEP_SMALL,
  PK_COUNT
};
const char* const kProfNames[PK_COUNT] = {
    "k_embed", "k_tok1", "k_ffn", "k_row_xf", "k_pair_alpha", "k_alpha_softmax",
    "k_pair_score", "k_assemble_argmax", "k_agg_alpha", "k_agg_finish", "misc", "k_pair_alpha_incr",
    "k_pair_score_incr", "k_qkv6", "k_row_s", "k_row_pv", "k_step_small"};

char g_err[512] = "";

struct LayerOff {
  size_t row[10], col[10];   // Wk,bk,Wv,bv,Wq,bq,Wo,bo,ln_w,ln_b
  size_t W1, b1, W2, b2, ln_w, ln_b;
};

}  // namespace

struct nnj_handle {
  nnj_config cfg;
  bool have_w = false;
  float* d_w = nullptr;          // packed weights followed by derived tensors
  float* d_wimg = nullptr;       // the scorer's four 64 x 64 operands as ready LDS images (k_build_scorer_images)
  // the fp64 encoder of alignments of more than 64 rows (nnj_encoder64.hpp): every packed tensor as doubles at its
  // packed offset, the embed tables unrounded, then per layer the stacked q|k|v matrices of both attentions
  double* d_w64 = nullptr;
  struct Qkv64 { size_t Wrow, brow, Wcol, bcol; };
  std::vector<Qkv64> lo64;
  int enc64 = 1;                 // NNJ_ENC64=0: the f16x3 encoder above 64 rows too; =2: the fp64 encoder at EVERY row count (both: A/B of the parity tables only)
  size_t n_packed = 0;
  std::vector<LayerOff> lo;
  size_t oE0, oe0, oE2, oe2, oWh, obh, oWg, obg, oWgq, obgq, oWgk, obgk, oS0, os0, os2w, os2b;
  size_t oA, oa0, ou, olut, optab;      // derived (after the packed block)
  size_t oWhS, obhS, oWgS, obgS; // derived: -log2(e) x (W_h, b_h, W_g, b_g): the sigmoids take exp2 arguments directly
  float t0 = 0.f, s2b = 0.f;
  int debug_stop = 0;            // encoder debug tap (nnj_debug_encoder_stop)
  int num_cu = 256;              // compute units of the device (persistent-kernel grid size)
  int* d_flag = nullptr;         // sticky "a pair score was not finite" flag (nnj_numeric_status)
  int concurrency = 2;           // sub-batches of a rollout that run on streams of their own (nnj_set_concurrency)
  int two_pass = 1;              // rollouts run the two-pass NJ step (nnj_step2.hpp); NNJ_TWO_PASS=0 selects the four-pass kernels
  int two_pass_cand = 1;         // ... and carry the candidate pair's logits along (NNJ_TWO_PASS_CAND=0: fallback pass instead)
  int step_w = 1;                // one wave per site in the alpha pass of 17..48 rows (NNJ_STEP_W=0: groups of waves)
  // small batches: the ~370 launches of a rollout are captured once into a hipGraph and replayed while the call's
  // arguments stay the same (launch-bound regime: BASELINE configs[1], one alignment per rollout)
  struct GraphKey {
    const void *codes, *mask, *forced, *uniforms, *merges, *trace, *gap, *state, *ws;
    int B, T, L, n_encode, debug_stop; float inv_temp;
    bool operator==(const GraphKey& o) const {
      return codes == o.codes && mask == o.mask && forced == o.forced && uniforms == o.uniforms && merges == o.merges &&
             trace == o.trace && gap == o.gap && state == o.state && ws == o.ws && B == o.B && T == o.T && L == o.L &&
             n_encode == o.n_encode && debug_stop == o.debug_stop && inv_temp == o.inv_temp;
    }
  };
  // nnj_step keeps the rows of an ongoing loop in slot layout inside the caller's workspace, with their cached
  // transforms (U, K', beta), exactly like the rollout: consecutive steps re-transform nothing
  struct StepSession {
    bool valid = false; const void* ws = nullptr; const float* last_out = nullptr; int B = 0, T0 = 0, L = 0, n = 0;
    // nnj_step on the two-pass kernels (nnj_step2.hpp): which of the two live lists / candidate slots is current,
    // whether the last table kernel left am / need / cand for the next merge, and whether every row still has its K'
    // (the two-pass step does not produce K' of merged rows: the four-pass entry points then start from the dense tensor)
    int live_idx = 0, cand_idx = 0; bool tp = false, kp_valid = true;
  };
  StepSession sess;
  GraphKey gkey{}, gcand{};      // key of the instantiated graph; key of the previous small-batch call
  hipGraphExec_t gexec = nullptr;
  int use_graph = 1;
  hipStream_t sub[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  char err[512] = "";
  // profiling
  bool prof = false;
  std::vector<hipEvent_t> ev;
  std::vector<int> ev_kind;
  size_t ev_used = 0;
  int64_t prof_dropped = 0;      // launches that could not be recorded (event creation failed): reported, never silent
  double prof_ms[PK_COUNT] = {0};
  int64_t prof_n[PK_COUNT] = {0};
};

namespace {

int fail(nnj_handle* h, int code, const char* fmt, ...) {
  char* dst = h ? h->err : g_err;
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(dst, 512, fmt, ap);
  va_end(ap);
  return code;
}

#define HIPCHK(h, expr)                                                                       \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(h, NNJ_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

size_t count_params(const nnj_config& c) {
  size_t D = c.embed_dim, F = 4 * D, V = c.vocab_size, K = c.patch_size;
  size_t per_layer = 2 * (4 * (D * D + D) + 2 * D) + (F * D + F) + (D * F + D) + 2 * D;
  return per_layer * c.num_layers + (D * V * K + D) + (D * D + D) + 4 * (D * D + D) + (D * D + D) + (D + 1);
}

AttnW attn_ptrs(const nnj_handle* h, const size_t (&o)[10]) {
  const float* w = h->d_w;
  return AttnW{w + o[0], w + o[1], w + o[2], w + o[3], w + o[4], w + o[5], w + o[6], w + o[7], w + o[8], w + o[9], h->cfg.embed_dim};
}
FfnW ffn_ptrs(const nnj_handle* h, const LayerOff& l) {
  const float* w = h->d_w;
  return FfnW{w + l.W1, w + l.b1, w + l.W2, w + l.b2, w + l.ln_w, w + l.ln_b, h->cfg.embed_dim};
}
ScorerW scorer_ptrs(const nnj_handle* h) {
  const float* w = h->d_w;
  ScorerW s;
  s.Wh = w + h->oWhS; s.bh = w + h->obhS; s.Wg = w + h->oWgS; s.bg = w + h->obgS;   // pre-scaled by -log2(e)
  s.A = w + h->oA; s.a0 = w + h->oa0; s.u = w + h->ou; s.t0 = h->t0;
  s.S0 = w + h->oS0; s.s0 = w + h->os0; s.s2w = w + h->os2w; s.s2b = h->s2b;
  s.imgAt = h->d_wimg; s.imgWh = h->d_wimg + IMG64; s.imgWg = h->d_wimg + 2 * IMG64; s.imgS0 = h->d_wimg + 3 * IMG64;
  return s;
}

// ---- profiling: one event pair per launch, tagged with the kernel kind
struct Scope {
  nnj_handle* h; hipStream_t st; int kind; bool on;
  Scope(nnj_handle* h_, hipStream_t st_, int kind_) : h(h_), st(st_), kind(kind_), on(false) {
    if (!h->prof) return;
    if (h->ev_used + 2 > h->ev.size()) {          // the pool grows on demand: no launch goes unrecorded
      const size_t old = h->ev.size();
      h->ev.resize(old + 2048);
      h->ev_kind.resize(h->ev.size() / 2, 0);
      for (size_t i = old; i < h->ev.size(); ++i)
        if (hipEventCreate(&h->ev[i]) != hipSuccess) { h->ev.resize(i & ~(size_t)1); h->prof_dropped++; break; }
    }
    if (h->ev_used + 2 <= h->ev.size()) {
      on = true;
      hipEventRecord(h->ev[h->ev_used], st);
    } else {
      h->prof_dropped++;
    }
  }
  ~Scope() {
    if (on) {
      hipEventRecord(h->ev[h->ev_used + 1], st);
      h->ev_kind[h->ev_used / 2] = kind;
      h->ev_used += 2;
    }
  }
};

template <typename K>
int set_lds(nnj_handle* h, K kernel, size_t bytes) {
  if (bytes > 48 * 1024)
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return NNJ_OK;
}

// ---- shapes the kernels cover
int check_shape(nnj_handle* h, int B, int T, int L) {
  if (B <= 0 || T < 1 || L <= 0) return fail(h, NNJ_ERR_ARG, "bad shape B=%d T=%d L=%d", B, T, L);
  if (T > 256) return fail(h, NNJ_ERR_UNSUPPORTED, "T=%d: this build covers up to 256 rows (eight waves per alignment column, 256-row scorer images)", T);
  return NNJ_OK;
}

struct EncDims { int Epad, NT; size_t hm; };   // hm = floats of the head-major context buffer
EncDims enc_dims(int B, int T, int C) {
  EncDims d;
  d.Epad = (T * 8 + 15) / 16 * 16;             // row length of ctx [B,8,C,Epad]
  d.NT = T <= 32 ? 1 : (T <= 64 ? 2 : (T <= 128 ? 4 : 8));   // waves per column in k_tok1p
  d.hm = (size_t)B * NNJ_NHEAD * C * d.Epad;
  return d;
}

// pair-scorer launch geometry
struct PairGeom {
  int npairs, tpw, pg, ppad, nsc, nsc_a, cs, blocks;     // nsc / nsc_a: partial sets of the scores / of alpha
  const float* score_src = nullptr;                      // where k_assemble_argmax reads the partials (nullptr: w.score_part)
};

// more than 64 live rows: star launches of nnj_scorer_wide.hpp
WideGeom wide_geom(int n, int B, int C, bool full) {
  WideGeom g;
  g.PT = n <= 128 ? 1 : 2;
  g.RP = 128 * g.PT;
  g.M = full ? n - 1 : 1;
  long nsc = (512 + (long)g.M * B - 1) / ((long)g.M * B);            // ~2 workgroups per CU
  const long max_nsc = (C + 15) / 16;                                 // at least 16 sites per workgroup (weight staging)
  if (nsc > max_nsc) nsc = max_nsc;
  // at most 128 sites per workgroup: the alpha logits are sums over 64 C products, accumulated by the matrix pipe in
  // fp32 site after site; over thousands of sites the roundings of the running sum reach 1e-4 of the scores
  // (measured at 200 x 4096).  Chunk sums of <= 128 sites, added in k_wide_softmax, keep them at the fp32 level.
  if (nsc < (C + 127) / 128) nsc = (C + 127) / 128;
  if (nsc < 1) nsc = 1;
  g.cs = (int)((C + nsc - 1) / nsc);
  g.nsc = (C + g.cs - 1) / g.cs;
  const size_t per_star = (size_t)B * g.nsc * g.RP * g.RP;            // floats of alpha partials per star
  size_t mb = ((size_t)1 << 29) / per_star;                           // <= 2 GiB of partials per launch
  if (mb < 1) mb = 1;
  g.MB = (int)std::min<size_t>(mb, (size_t)g.M);
  return g;
}
PairGeom pair_geom(int mode, int n, int B, int C) {
  PairGeom g;
  if (mode == PAIRS_INCR) {
    // wave-private kernels: grid (nsc_blocks, B), 4 partial sets per block; nsc counts PARTIALS
    g.npairs = n; g.tpw = 1; g.pg = 1; g.ppad = 64;
    int blocks = (1024 + B - 1) / B;           // ~1024 workgroups per launch (measured flat from 512 to 3072)
    // small batches: down to 4 sites per workgroup -- one alignment of 1024 sites then runs 256 workgroups with four busy
    // waves each, one per SIMD of every CU, instead of 128 with eight (B = 1: 3.53 -> 3.48 ms per tree; 2 sites: 4.6)
    constexpr int mincs = 4;
    const int max_blocks = (C + mincs - 1) / mincs;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks < 1) blocks = 1;
    g.cs = (C + blocks - 1) / blocks;
    blocks = (C + g.cs - 1) / g.cs;
    // one partial set per workgroup (the site slots of a workgroup are summed in LDS, in slot order)
    g.blocks = blocks;
    g.nsc_a = blocks;
    g.nsc = blocks;
    return g;
  }
  g.npairs = mode == PAIRS_FULL ? n * (n - 1) / 2 : n;
  g.tpw = 1;                       // 8 tiles per workgroup: 8 waves x 1 tile (alpha) or 4 waves x 2 tiles (score)
  const int tiles = (g.npairs + 31) / 32;
  g.pg = (tiles + 8 * g.tpw - 1) / (8 * g.tpw);
  g.ppad = g.pg * 8 * g.tpw * 32;
  int nsc = (2048 + g.pg * B - 1) / (g.pg * B);
  const int max_nsc = (C + 7) / 8;
  if (nsc > max_nsc) nsc = max_nsc;
  if (nsc < 1) nsc = 1;
  g.cs = (C + nsc - 1) / nsc;
  g.nsc = (C + g.cs - 1) / g.cs;
  g.nsc_a = g.nsc; g.blocks = g.nsc;
  return g;
}

// beta partials per row: one per 32 sites from k_row_xf / k_agg_finish, one per workgroup from k_step_alpha (which
// may run more workgroups per alignment than there are 32-site tiles when the batch is small); unused entries are 0
int beta_stride(int B, int C) {
  const int nt32 = (C + 31) / 32;
  return std::max(nt32, pair_geom(PAIRS_INCR, 2, B, C).blocks);
}

// workspace regions of the NJ loop (after the state/slots buffer), in floats
struct LoopWs {
  size_t U, Kp, beta, alpha_part, alpha, score_part, full, agg_part, logits0, logits1, merged, live, ij, zmask;
  size_t lam, beta_slot, acand, Xc, am, need, cand, cand_run, mrep, pick, end; // the two-pass step (nnj_step2.hpp); replicated mask
};
LoopWs loop_ws(int B, int T, int C) {
  LoopWs w;
  const size_t rows = (size_t)B * T * C * 64;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += align_up(n, 64); return r; };
  w.U = take(rows);
  w.Kp = take(rows);
  w.beta = take((size_t)B * T * beta_stride(B, C));
  size_t ap = 0, al = 0, sp = 0, fu = 0;
  const int Tn = std::min(T, 64);                                  // the 64-row kernels
  for (int mode = 0; mode < 2; ++mode)
    for (int n : {Tn, std::min(Tn, 48), std::min(Tn, 32), std::min(Tn, 16)})     // the incremental geometry changes at n = 48, 32 and 16
      for (int Bg : {B, 1}) {              // Bg = 1: the one-alignment launch of sampled replicas (rollout_core, rep0)
        if (n < 2) continue;
        PairGeom g = pair_geom(mode, n, Bg, C);
        ap = std::max(ap, (size_t)Bg * g.nsc_a * g.ppad * 64);
        al = std::max(al, (size_t)Bg * g.ppad * 64);
        sp = std::max(sp, (size_t)Bg * g.nsc * g.ppad);
      }
  if (T > 64) {                                                    // the star kernels (more than 64 live rows)
    for (int full = 0; full < 2; ++full)
      for (int n : {T, std::min(T, 128)})
        // Bg = 1: the first table of sampled replicas of ONE alignment is launched for alignment 0 only (rollout_core,
        // rep0), with the geometry of a batch of one -- whose star blocks (MB) can be larger than B x MB(B)
        for (int Bg : {B, 1}) {
          const WideGeom g = wide_geom(n, Bg, C, full != 0);
          ap = std::max(ap, (size_t)Bg * g.MB * g.nsc * g.RP * g.RP);
          al = std::max(al, (size_t)Bg * g.MB * g.RP * g.RP);
          sp = std::max(sp, (size_t)Bg * g.MB * g.nsc * g.RP);
        }
    fu = (size_t)B * T * (T - 1) / 2;
  }
  w.alpha_part = take(ap);
  w.alpha = take(al);
  w.score_part = take(sp);
  w.full = take(fu);
  w.agg_part = take((size_t)B * ((C + 15) / 16) * (T > 128 ? 256 : (T > 64 ? 128 : 64)));
  w.logits0 = take((size_t)B * T * (T - 1) / 2 + 1);
  w.logits1 = take((size_t)B * T * (T - 1) / 2 + 1);
  w.merged = take((size_t)B * C * 64);
  w.live = take((size_t)2 * B * T);            // two lists (current / next step)
  w.ij = take((size_t)B * 2 + 2);
  w.zmask = take(((size_t)B * C + 3) / 4);       // all-false site mask for callers that pass none (bytes)
  w.lam = take((size_t)B * 4096);
  w.beta_slot = take((size_t)B * T);
  w.acand = take((size_t)B * pair_geom(PAIRS_INCR, 2, B, C).blocks * 64);
  w.Xc = take((size_t)B * C * 64);
  w.am = take((size_t)B * 64);
  w.need = take((size_t)B);
  w.cand = take((size_t)B * 4);                  // two lists of (a, b): this step's candidate / the next one's
  w.cand_run = take((size_t)B);
  w.mrep = take(((size_t)B * C + 3) / 4);          // site mask of a replicated alignment (bytes)
  w.pick = take((size_t)B * 2);                    // nnj_step: the pick the stored merge weights belong to
  w.end = o;
  return w;
}

// A launch geometry against the regions loop_ws reserved (the workspace may have been sized for another batch than the one a
// launch is for): never write past a region silently.
int check_pair_fit(nnj_handle* h, const LoopWs& w, const PairGeom& g, int B, const char* what) {
  if ((size_t)B * g.nsc_a * g.ppad * 64 > w.alpha - w.alpha_part || (size_t)B * g.ppad * 64 > w.score_part - w.alpha ||
      (size_t)B * g.nsc * g.ppad > w.full - w.score_part)
    return fail(h, NNJ_ERR_WORKSPACE, "%s (pairs=%d, B=%d, partial sets=%d) does not fit the workspace regions", what, g.npairs, B, g.nsc);
  return NNJ_OK;
}

// Zero fill as a KERNEL.  The small-batch rollout is captured into a hipGraph and replayed (rollout_impl): with the three
// hipMemsetAsync calls of the path captured as memset nodes every second replay returned wrong score tables for C >= 256
// (round 4: tools/graph_check2.py, profiles/r04/graph_check.txt -- the encoder output right, the tables wrong: the zero
// mask / the beta tail were not zero when the scorer read them); as kernel nodes they are ordered like everything else.
__global__ void k_zero_bytes(uint8_t* __restrict__ p, size_t n) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i + 16 <= n) *reinterpret_cast<u32x4*>(p + i) = (u32x4){0u, 0u, 0u, 0u};
  else for (size_t k = i; k < n; ++k) p[k] = 0;
}
// (p 16-byte aligned: every caller passes a workspace region)
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {
  if (!bytes) return hipSuccess;
  const size_t threads = (bytes + 15) / 16;
  hipLaunchKernelGGL(k_zero_bytes, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, static_cast<uint8_t*>(p), bytes);
  return hipGetLastError();
}

// the scorer kernels read the site mask unconditionally (no branch in their site loops): a caller without a mask
// gets a zero-filled one from the workspace
int scorer_mask(nnj_handle* h, const uint8_t* mask, float* base, const LoopWs& w, int B, int C, hipStream_t st,
                const uint8_t** out) {
  if (mask) { *out = mask; return NNJ_OK; }
  uint8_t* z = reinterpret_cast<uint8_t*>(base + w.zmask);
  HIPCHK(h, zero_async(z, (size_t)B * C, st));
  *out = z;
  return NNJ_OK;
}

// encoder scratch (floats, after the state): ctx | Q6 | K6 | V6 | S | M | key classes
struct EncWs { size_t ctx, q6, k6, v6, s, m, cls, end; };
EncWs enc_ws(int B, int T, int C) {
  const EncDims d = enc_dims(B, T, C);
  const Ra6 g = ra6_geom(T, C, d.Epad, B);
  const size_t nbh = (size_t)B * NNJ_NHEAD;
  EncWs w;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += align_up(n, 64); return r; };
  w.ctx = take(d.hm);
  w.q6 = take(nbh * g.qk_bh / 4);
  w.k6 = take(nbh * g.qk_bh / 4);
  w.v6 = take(nbh * g.v_bh / 4);
  w.s = take(nbh * g.s_bh);
  w.m = take(nbh * g.m_bh);
  w.cls = take(((size_t)B * g.Cp + 3) / 4 + 1);                          // key classes
  w.end = o;
  return w;
}

// scratch of the fp64 encoder (more than 64 rows), in DOUBLES, for ONE alignment (the host loops over the batch):
// x | y = LayerNorm(x) | big = { Q, K, V planes [8][R*8][Cp] + ctx  |  column q|k|v [token][192] + ctx  |  FFN hidden [token][256] } | S [8][C][Cp]
struct Enc64Ws { size_t x, y, big, ctx, s, end; long Cp, plane; };
Enc64Ws enc64_ws(int T, int C) {
  Enc64Ws w;
  w.Cp = (C + 1) / 2 * 2;
  w.plane = (long)T * 8 * w.Cp;
  const size_t tok64 = (size_t)T * C * 64;
  const size_t planes = 3 * 8 * (size_t)w.plane;
  size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += align_up(n, 32); return r; };
  w.x = take(tok64);
  w.y = take(tok64);
  w.big = take(std::max(planes, 3 * tok64) + 32 + tok64);
  w.ctx = w.big + align_up(std::max(planes, 3 * tok64), 32);           // the context of either attention, behind its operands
  w.s = take((size_t)8 * C * w.Cp);
  w.end = o;
  return w;
}
size_t ws_floats_one(int B, int T, int C) {
  const size_t state = align_up((size_t)B * T * C * 64, 64);
  // (NNJ_ENC64=2, a diagnostic: the fp64 encoder below 65 rows too -- its scratch must then exist at every shape)
  static const bool enc64_always = getenv("NNJ_ENC64") && atoi(getenv("NNJ_ENC64")) == 2;
  const size_t enc = std::max(enc_ws(B, T, C).end, (T > 64 || enc64_always) ? 2 * enc64_ws(T, C).end : (size_t)0);
  const size_t loop = loop_ws(B, T, C).end;
  return state + std::max(enc, loop) + 256;
}
// a rollout may run as up to NNJ_MAX_SUB independent sub-batches on streams of their own (nnj_set_concurrency):
// the workspace holds either the whole batch or every split of it
constexpr int NNJ_MAX_SUB = 4;
// the tail of the workspace holds the token mask of a patched alignment (patch_size > 1: mask[:, ::patch_size])
size_t pm_floats(int B, int C) { return align_up(((size_t)B * C + 3) / 4, 64); }
size_t ws_floats(int B, int T, int C) {
  size_t need = ws_floats_one(B, T, C);
  for (int ns = 2; ns <= NNJ_MAX_SUB; ++ns)
    if (B >= ns) need = std::max(need, (size_t)ns * align_up(ws_floats_one((B + ns - 1) / ns, T, C), 64));
  return align_up(need, 64) + pm_floats(B, C);
}

// ------------------------------------------------------------------ encoder launches
template <int NT>
int launch_encoder(nnj_handle* h, const uint8_t* codes, const float* onehot, const uint8_t* mask, float* x,
                   float* scratch, int B, int T, int C, hipStream_t st) {
  const EncDims d = enc_dims(B, T, C);
  const EncWs ew_ = enc_ws(B, T, C);
  const Ra6 g6 = ra6_geom(T, C, d.Epad, B);
  float* ctx = scratch + ew_.ctx;
  uint8_t* Q6 = reinterpret_cast<uint8_t*>(scratch + ew_.q6);
  uint8_t* K6 = reinterpret_cast<uint8_t*>(scratch + ew_.k6);
  uint8_t* V6 = reinterpret_cast<uint8_t*>(scratch + ew_.v6);
  float* Sbuf = scratch + ew_.s;
  float* Mbuf = scratch + ew_.m;
  uint8_t* cls = reinterpret_cast<uint8_t*>(scratch + ew_.cls);
  const int nbh = B * NNJ_NHEAD;
  const int nl = h->cfg.num_layers;
  const unsigned colblocks = (unsigned)(((size_t)B * C + 3) / 4);
  const float* lut = h->d_w + h->olut;
  {
    Scope sc(h, st, PK_EMBED);
    const size_t lds = 4096 * sizeof(float);
    if (int rc = set_lds(h, k_embed, lds)) return rc;
    const float* wp = h->d_w;
    const EmbedW ew{wp + h->oE0, wp + h->oe0, wp + h->oE2, wp + h->oe2};
    hipLaunchKernelGGL(k_embed, dim3(colblocks), dim3(256), lds, st, codes, onehot, ew, lut, h->d_w + h->optab, x, B, T, C,
                       h->cfg.patch_size);
    hipLaunchKernelGGL(k_key_classes, dim3((unsigned)(((size_t)B * g6.Cp + 255) / 256)), dim3(256), 0, st, mask, cls, B,
                       C, g6.Cp);
    // V6 keys beyond the alignment meet probabilities that are exactly 0: they only have to be finite
    if (g6.Cp != C) HIPCHK(h, zero_async(V6, (size_t)nbh * g6.v_bh, st));
  }
  // reference no-grad chunking: one masked_fill(-10000) per row chunk, summed (axial_attention.py:35-64)
  int nchunks = 1;
  if ((long)T * C > 1024) { int max_rows = 1024 / C; if (max_rows < 1) max_rows = 1; nchunks = (T + max_rows - 1) / max_rows; }
  // k_row_s writes the logits (and their tile maxima) times log2(e): the probabilities of k_row_pv are one v_exp_f32
  // (2^x) each, without the multiplication exp(x) costs per key and query
  const float LOG2E = 1.4426950408889634f;
  const float fill = -10000.0f * (float)nchunks * LOG2E;
  for (int l = 0; l < nl; ++l) {
    {
      Scope sc(h, st, PK_ROW_QKV);
      const size_t lds_qkv = (size_t)(b6_floats(192, 64) + 320) * sizeof(float);
      if (int rc = set_lds(h, k_qkv6, lds_qkv)) return rc;
      const long tiles = (long)((T + 1) / 2) * ((C + 15) / 16);
      const long ngroups = ((tiles + 7) / 8) * B;
      const unsigned grid = (unsigned)std::min<long>(ngroups, 2L * h->num_cu);
      hipLaunchKernelGGL(k_qkv6, dim3(grid), dim3(512), lds_qkv, st, (const float*)x, mask,
                         attn_ptrs(h, h->lo[l].row), Q6, K6, V6, g6, B);
    }
    {
      Scope sc(h, st, PK_ROW_S);
      const float qs = LOG2E / (sqrtf((float)NNJ_DH) * sqrtf((float)T));
      if (T <= 64) {
        constexpr int QW = 128;
        const size_t lds = (size_t)RsShape<QW>::NST * RsShape<QW>::STAGE;
        if (int rc = set_lds(h, k_row_s<QW>, lds)) return rc;
        const unsigned grid = (unsigned)((long)(nbh + 7) / 8 * 8 * g6.nrb * (g6.Cp / QW));
        hipLaunchKernelGGL(k_row_s<QW>, dim3(grid), dim3(256), lds, st, (const uint8_t*)Q6, (const uint8_t*)K6,
                           (const uint8_t*)cls, Sbuf, Mbuf, g6, nbh, fill, qs);
      } else {              // more than 64 rows (only with NNJ_ENC64=0): chunked accumulation of the 8R-term logits (see k_row_s)
        constexpr int QW = 64, CHK = NNJ_RS_CHK;
        const size_t lds = (size_t)RsShape<QW>::NST * RsShape<QW>::STAGE;
        if (int rc = set_lds(h, (k_row_s<QW, CHK>), lds)) return rc;
        const unsigned grid = (unsigned)((long)(nbh + 7) / 8 * 8 * g6.nrb * (g6.Cp / QW));
        hipLaunchKernelGGL((k_row_s<QW, CHK>), dim3(grid), dim3(256), lds, st, (const uint8_t*)Q6, (const uint8_t*)K6,
                           (const uint8_t*)cls, Sbuf, Mbuf, g6, nbh, fill, qs);
      }
    }
    {
      Scope sc(h, st, PK_ROW_PV);
      const unsigned grid = (unsigned)((nbh + 7) / 8 * 8 * (g6.Cp / 128) * g6.nech);
#define NNJ_PV_CASE(N)                                                                                   \
  case N: {                                                                                              \
    const size_t stg = (N * NPL * 1024 + 4095) / 4096 * 4096, lds = (4 * stg <= 163840 ? 4 : 3) * stg;        \
    if (int rc = set_lds(h, k_row_pv<N>, lds)) return rc;                                                \
    hipLaunchKernelGGL(k_row_pv<N>, dim3(grid), dim3(256), lds, st, (const uint8_t*)V6, (const float*)Sbuf, \
                       (const float*)Mbuf, ctx, g6, nbh);                                                \
  } break;
      switch (g6.ETc) {
        NNJ_PV_CASE(1) NNJ_PV_CASE(2) NNJ_PV_CASE(4) NNJ_PV_CASE(6) NNJ_PV_CASE(7) NNJ_PV_CASE(8) NNJ_PV_CASE(10) NNJ_PV_CASE(13)
        NNJ_PV_CASE(16)
        default: return fail(h, NNJ_ERR_UNSUPPORTED, "row attention: no instantiation for %d head tiles", g6.ETc);
      }
#undef NNJ_PV_CASE
    }
    {
      Scope sc(h, st, PK_TOK1);
      const int dbg = ((l == 0 && h->debug_stop == 1) ? 1 : 0) | (h->debug_stop >= 16 ? (h->debug_stop >> 4) << 1 : 0);
      // NT waves per column (32 rows each), 8 / NT columns in flight per workgroup; persistent
      constexpr int NSLOT = 8 / NT;
      const unsigned grid1 = (unsigned)std::min<long>((long)(((size_t)B * C + NSLOT - 1) / NSLOT), (long)h->num_cu);
      const size_t lds = (size_t)(5 * 4096 + NSLOT * (32 * NT * 32 + 32 * NT * 32) + 16 + 448) * sizeof(float);
      if (NT <= 2 && T <= 32 * NT - 8) {                  // the last eight keys are padding in every lane: k_tok1p<NT, SK>
        if (int rc = set_lds(h, (k_tok1p<NT, (NT <= 2)>), lds)) return rc;
        hipLaunchKernelGGL((k_tok1p<NT, (NT <= 2)>), dim3(grid1), dim3(512), lds, st, ctx, mask, x, attn_ptrs(h, h->lo[l].row),
                           attn_ptrs(h, h->lo[l].col), B, T, C, d.Epad, dbg, h->d_flag);
      } else {
        if (int rc = set_lds(h, k_tok1p<NT>, lds)) return rc;
        hipLaunchKernelGGL(k_tok1p<NT>, dim3(grid1), dim3(512), lds, st, ctx, mask, x, attn_ptrs(h, h->lo[l].row),
                           attn_ptrs(h, h->lo[l].col), B, T, C, d.Epad, dbg, h->d_flag);
      }
    }
    if (l == 0 && (h->debug_stop == 1 || h->debug_stop == 2)) break;
    {
      Scope sc(h, st, PK_FFN);
      const size_t lds = (size_t)16384 * sizeof(float);
      {   // persistent FFN / QKV kernels over flat 256-token groups, one workgroup per CU
        const size_t lds_ffn = (size_t)(8 * IMG64 + 448) * sizeof(float);   // W1 and W2 whole (8 images of 64x64) + LN / bias vectors
        const int groups_per_b = (T * C + 255) / 256;
        const long ngroups = (long)groups_per_b * B;
        const unsigned grid = (unsigned)std::min<long>(ngroups, h->num_cu);
        if (int rc = set_lds(h, k_ffn16, lds_ffn)) return rc;
        hipLaunchKernelGGL(k_ffn16, dim3(grid), dim3(1024), lds_ffn, st, x, ffn_ptrs(h, h->lo[l]), B, T, C, groups_per_b);
      }
    }
  }
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

// ---- more than 64 rows: the encoder in fp64 (nnj_encoder64.hpp), one alignment at a time
template <int BM, int BN, bool TA, bool TB, typename EPI>
int gemm64(nnj_handle* h, const Gemm64& g, const EPI& epi, int batches, hipStream_t st) {
  const size_t lds = gemm64_lds<BM, BN, TA, TB>();
  if (int rc = set_lds(h, (k64_gemm<BM, BN, TA, TB, EPI>), lds)) return rc;
  const long gx = (g.N + BN - 1) / BN, gy = (g.M + BM - 1) / BM;
  if (gy > 65535 || batches > 65535) return fail(h, NNJ_ERR_UNSUPPORTED, "fp64 encoder: %d x %d exceeds the launch grid", g.M, g.N);
  hipLaunchKernelGGL((k64_gemm<BM, BN, TA, TB, EPI>), dim3((unsigned)gx, (unsigned)gy, (unsigned)batches), dim3(256), lds, st, g, epi);
  return NNJ_OK;
}
int run_encoder64(nnj_handle* h, const uint8_t* codes, const float* onehot, const uint8_t* mask, float* x_out,
                  float* scratch, int B, int T, int C, hipStream_t st) {
  const Enc64Ws ew = enc64_ws(T, C);
  double* base = reinterpret_cast<double*>(scratch);
  double *x = base + ew.x, *y = base + ew.y, *big = base + ew.big, *ctx = base + ew.ctx, *S = base + ew.s;
  const double* w = h->d_w64;
  const int K = h->cfg.patch_size, dt = h->cfg.embed_dim, nl = h->cfg.num_layers;
  const long ntok = (long)T * C, L = (long)C * K;
  const int RD = T * 8;
  const unsigned ln_grid = (unsigned)((ntok * 16 + 255) / 256);
  // reference no-grad chunking: one masked_fill(-10000) per row chunk, summed (axial_attention.py:35-64)
  int nchunks = 1;
  if ((long)T * C > 1024) { int max_rows = 1024 / C; if (max_rows < 1) max_rows = 1; nchunks = (T + max_rows - 1) / max_rows; }
  const double fill = -10000.0 * nchunks;
  const double row_scale = 1.0 / (sqrt((double)NNJ_DH) * sqrt((double)T)), col_scale = 1.0 / sqrt((double)NNJ_DH);
  const size_t lds_col = (size_t)T * 32 * sizeof(double);
  const int nu = (T + 127) / 128;                      // query rows per thread of the column-attention kernel
  if (int rc = nu == 1 ? set_lds(h, k64_col_attention<1>, lds_col) : set_lds(h, k64_col_attention<2>, lds_col)) return rc;
  for (int b = 0; b < B; ++b) {
    const uint8_t* mb = mask ? mask + (size_t)b * C : nullptr;
    {
      Scope sc(h, st, PK_EMBED);
      const Embed64W e{w + h->oE0, w + h->oe0, w + h->oE2, w + h->oe2, w + h->olut, w + h->optab};
      hipLaunchKernelGGL(k64_embed, dim3((unsigned)((ntok + 3) / 4)), dim3(256), 0, st,
                         codes ? codes + (size_t)b * T * L : nullptr, onehot ? onehot + (size_t)b * T * L * 4 : nullptr, e, x,
                         T, C, K);
    }
    for (int l = 0; l < nl; ++l) {
      const LayerOff& lo = h->lo[l];
      const nnj_handle::Qkv64& q64 = h->lo64[l];
      double *Qp = big, *Kp = big + 8 * ew.plane, *Vp = big + 16 * ew.plane;
      {   // tied row attention: LN -> q,k,v planes
        Scope sc(h, st, PK_ROW_QKV);
        hipLaunchKernelGGL(k64_layernorm, dim3(ln_grid), dim3(256), 0, st, (const double*)x, y, w + lo.row[8], w + lo.row[9], ntok, dt);
        const Gemm64 g{w + q64.Wrow, 64, 0, y, 64, 0, 192, (int)ntok, 64};
        const EpiRowQkv epi{Qp, Kp, Vp, w + q64.brow, mb, C, ew.Cp, ew.plane, row_scale};
        if (int rc = gemm64<64, 128, false, false>(h, g, epi, 1, st)) return rc;
      }
      {   // S = Q^T K per head, softmax over the keys
        Scope sc(h, st, PK_ROW_S);
        const Gemm64 g{Qp, ew.Cp, ew.plane, Kp, ew.Cp, ew.plane, C, C, RD};
        const EpiLogits epi{S, ew.Cp, (long)C * ew.Cp, mb, fill};
        if (int rc = gemm64<128, 128, true, true>(h, g, epi, 8, st)) return rc;
        hipLaunchKernelGGL(k64_softmax_rows, dim3((unsigned)(8 * C)), dim3(256), 0, st, S, ew.Cp, C);
      }
      {   // ctx = P V
        Scope sc(h, st, PK_ROW_PV);
        const Gemm64 g{S, ew.Cp, (long)C * ew.Cp, Vp, ew.Cp, ew.plane, C, RD, C};
        const EpiCtx epi{ctx, C};
        if (int rc = gemm64<128, 128, false, false>(h, g, epi, 8, st)) return rc;
      }
      {
        Scope sc(h, st, PK_TOK1);
        {   // row out-projection + residual
          const Gemm64 g{ctx, 64, 0, w + lo.row[6], 64, 0, (int)ntok, 64, 64};
          const EpiResid epi{x, w + lo.row[7]};
          if (int rc = gemm64<128, 64, false, false>(h, g, epi, 1, st)) return rc;
        }
        if (!(l == 0 && h->debug_stop == 1)) {
          // column attention: LN -> q|k|v -> R x R attention per column and head -> out-projection + residual
          hipLaunchKernelGGL(k64_layernorm, dim3(ln_grid), dim3(256), 0, st, (const double*)x, y, w + lo.col[8], w + lo.col[9], ntok, dt);
          const Gemm64 g{y, 64, 0, w + q64.Wcol, 64, 0, (int)ntok, 192, 64};
          const EpiColQkv epi{big, w + q64.bcol, col_scale};
          if (int rc = gemm64<128, 64, false, false>(h, g, epi, 1, st)) return rc;
          const dim3 cgrid((unsigned)C, 4);
          if (nu == 1) hipLaunchKernelGGL(k64_col_attention<1>, cgrid, dim3(256), lds_col, st, (const double*)big, ctx, mb, T, C);
          else hipLaunchKernelGGL(k64_col_attention<2>, cgrid, dim3(256), lds_col, st, (const double*)big, ctx, mb, T, C);
          const Gemm64 g2{ctx, 64, 0, w + lo.col[6], 64, 0, (int)ntok, 64, 64};
          const EpiResid epi2{x, w + lo.col[7]};
          if (int rc = gemm64<128, 64, false, false>(h, g2, epi2, 1, st)) return rc;
        }
      }
      if (l == 0 && (h->debug_stop == 1 || h->debug_stop == 2)) break;
      {
        Scope sc(h, st, PK_FFN);
        hipLaunchKernelGGL(k64_layernorm, dim3(ln_grid), dim3(256), 0, st, (const double*)x, y, w + lo.ln_w, w + lo.ln_b, ntok, dt);
        const Gemm64 g{y, 64, 0, w + lo.W1, 64, 0, (int)ntok, 256, 64};
        const EpiGelu epi{big, w + lo.b1};
        if (int rc = gemm64<128, 128, false, false>(h, g, epi, 1, st)) return rc;
        const Gemm64 g2{big, 256, 0, w + lo.W2, 256, 0, (int)ntok, 64, 256};
        const EpiResid epi2{x, w + lo.b2};
        if (int rc = gemm64<128, 64, false, false>(h, g2, epi2, 1, st)) return rc;
      }
    }
    {
      Scope sc(h, st, PK_MISC);
      const long n = ntok * 64;
      hipLaunchKernelGGL(k64_store_f32, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, st, (const double*)x,
                         x_out + (size_t)b * ntok * 64, n);
    }
  }
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

int run_encoder(nnj_handle* h, const uint8_t* codes, const uint8_t* mask, float* x, float* scratch, int B, int T,
                int C, hipStream_t st, const float* onehot = nullptr) {
  if ((T > 64 && h->enc64) || h->enc64 == 2) return run_encoder64(h, codes, onehot, mask, x, scratch, B, T, C, st);
  switch (enc_dims(B, T, C).NT) {
    case 1: return launch_encoder<1>(h, codes, onehot, mask, x, scratch, B, T, C, st);
    case 2: return launch_encoder<2>(h, codes, onehot, mask, x, scratch, B, T, C, st);
    case 4: return launch_encoder<4>(h, codes, onehot, mask, x, scratch, B, T, C, st);
    default: return launch_encoder<8>(h, codes, onehot, mask, x, scratch, B, T, C, st);
  }
}

// ------------------------------------------------------------------ NJ-loop launches
int launch_row_xf(nnj_handle* h, const float* S, float* U, float* Kp, float* beta, long bstride, int slots,
                  int rows, int B, int C, hipStream_t st) {
  // (beta entries beyond the 32-site tiles of a row stay 0: see beta_stride)
  HIPCHK(h, zero_async(beta, (size_t)B * slots * beta_stride(B, C) * sizeof(float), st));
  Scope sc(h, st, PK_ROW_XF);
  const int nt32 = (C + 31) / 32;
  const size_t lds = 2 * 4096 * sizeof(float);
  hipLaunchKernelGGL(k_row_xf, dim3((unsigned)((nt32 + 3) / 4), (unsigned)rows, (unsigned)B), dim3(256), lds, st, S,
                     U, Kp, beta, scorer_ptrs(h), bstride, slots, C, nt32, beta_stride(B, C));
  return NNJ_OK;
}

// more than 64 live rows: the star kernels of nnj_scorer_wide.hpp (incremental: one star per batch element;
// all pairs: n-1 stars in launches of at most MB)
// per-row bias of the attention logits for the softmax kernels of the four-pass / all-pairs path: beta_slot[b][slot] = sum of
// the row's beta partials + C t0, the sum every pair's wave used to repeat (same order: the same bits).  The two-pass steps
// keep beta_slot up to date themselves and never run between these launches and their softmax.
int row_slots(const RowSet& rs, int C) { return (int)(rs.bstride / ((long)C * 64)); }
void launch_beta_sum(nnj_handle* h, const RowSet& rs, float* base, const LoopWs& w, int B, int C, hipStream_t st) {
  Scope sc(h, st, PK_STEP_SMALL);
  const int total = B * row_slots(rs, C);
  hipLaunchKernelGGL(k_beta_sum, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rs.beta_part, rs.ntile32, rs.ntile32,
                     (float)C * scorer_ptrs(h).t0, base + w.beta_slot, total);
}
int launch_pair_scores_wide(nnj_handle* h, const RowSet& rs, const int* ij_prev, const uint8_t* mask, float* base,
                            const LoopWs& w, int mode, int n, int B, int C, PairGeom& g, hipStream_t st) {
  const bool full = mode == PAIRS_FULL;
  const WideGeom wg = wide_geom(n, B, C, full);
  const ScorerW sw = scorer_ptrs(h);
  const int* ijp = full ? nullptr : ij_prev;
  // the regions were sized by loop_ws for the batch the workspace belongs to; a launch for another batch size (the
  // one-alignment form of sampled replicas) must fit them too -- never write past a region silently
  if ((size_t)B * wg.MB * wg.nsc * wg.RP * wg.RP > w.alpha - w.alpha_part ||
      (size_t)B * wg.MB * wg.RP * wg.RP > w.score_part - w.alpha || (size_t)B * wg.MB * wg.nsc * wg.RP > w.full - w.score_part)
    return fail(h, NNJ_ERR_WORKSPACE, "star launch (n=%d, B=%d, C=%d) does not fit the workspace regions", n, B, C);
  const size_t lds_a = (size_t)(2 * IMG64 + 64 * wg.RP + SCORER_CONSTS) * sizeof(float);
  const size_t lds_s = (size_t)(3 * IMG64 + 64 * wg.RP + SCORER_CONSTS) * sizeof(float);
  launch_beta_sum(h, rs, base, w, B, C, st);
  for (int m0 = 0; m0 < wg.M; m0 += wg.MB) {
    const int mc = std::min(wg.MB, wg.M - m0);
    const dim3 grid((unsigned)wg.nsc, (unsigned)mc, (unsigned)B);
    {
      Scope sc(h, st, full ? PK_PAIR_ALPHA : PK_PAIR_ALPHA_INCR);
      if (wg.PT == 1) {
        if (int rc = set_lds(h, k_wide_alpha<1>, lds_a)) return rc;
        hipLaunchKernelGGL(k_wide_alpha<1>, grid, dim3(512), lds_a, st, rs, sw, ijp, m0, base + w.alpha_part, n, C, wg.cs,
                           wg.nsc);
      } else {
        if (int rc = set_lds(h, k_wide_alpha<2>, lds_a)) return rc;
        hipLaunchKernelGGL(k_wide_alpha<2>, dim3(grid.x * 2, grid.y, grid.z), dim3(512), lds_a, st, rs, sw, ijp, m0,
                           base + w.alpha_part, n, C, wg.cs, wg.nsc);
      }
    }
    {
      Scope sc(h, st, PK_ALPHA_SOFTMAX);
      hipLaunchKernelGGL(k_wide_softmax, dim3((unsigned)(wg.RP / 4), (unsigned)mc, (unsigned)B), dim3(256), 0, st, rs, sw,
                         ijp, m0, base + w.alpha_part, base + w.alpha, n, C, wg.RP, wg.nsc,
                         (const float*)(base + w.beta_slot), row_slots(rs, C));
    }
    {
      Scope sc(h, st, full ? PK_PAIR_SCORE : PK_PAIR_SCORE_INCR);
      if (wg.PT == 1) {
        if (int rc = set_lds(h, k_wide_score<1>, lds_s)) return rc;
        hipLaunchKernelGGL(k_wide_score<1>, grid, dim3(512), lds_s, st, rs, sw, ijp, m0, base + w.alpha, mask,
                           base + w.score_part, n, C, wg.cs);
      } else {
        if (int rc = set_lds(h, k_wide_score<2>, lds_s)) return rc;
        hipLaunchKernelGGL(k_wide_score<2>, grid, dim3(512), lds_s, st, rs, sw, ijp, m0, base + w.alpha, mask,
                           base + w.score_part, n, C, wg.cs);
      }
    }
    if (full) {
      Scope sc(h, st, PK_MISC);
      hipLaunchKernelGGL(k_wide_gather_full, dim3((unsigned)((mc * wg.RP + 255) / 256), (unsigned)B), dim3(256), 0, st,
                         base + w.score_part, base + w.full, n, wg.RP, wg.nsc, m0, mc);
    }
  }
  g.npairs = full ? n * (n - 1) / 2 : n;
  if (full) { g.nsc = 1; g.ppad = n * (n - 1) / 2; g.score_src = base + w.full; }
  else { g.nsc = wg.nsc; g.ppad = wg.RP; g.score_src = base + w.score_part; }
  return NNJ_OK;
}

// scores of the pairs of `mode` into score_part (and alpha scratch)
int launch_pair_scores(nnj_handle* h, const RowSet& rs, const int* ij_prev, const uint8_t* mask, float* base,
                       const LoopWs& w, int mode, int n, int B, int C, PairGeom& g, hipStream_t st) {
  if (n > 64) return launch_pair_scores_wide(h, rs, ij_prev, mask, base, w, mode, n, B, C, g, st);
  g = pair_geom(mode, n, B, C);
  if (int rc = check_pair_fit(h, w, g, B, "pair-score launch")) return rc;
  g.score_src = base + w.score_part;
  const ScorerW sw = scorer_ptrs(h);
  const int has_ctx = n > 2 ? 1 : 0;
  if (mode == PAIRS_INCR) {
    const int ng = n > 48 ? 4 : (n > 32 ? 3 : (n > 16 ? 2 : 1));          // 16-row tiles (= waves) that share a site
    const dim3 grid((unsigned)g.blocks, (unsigned)B);
    const dim3 blk16(64 * T16_WAVES);
    if (has_ctx) {
      {
        Scope sc(h, st, PK_PAIR_ALPHA_INCR);
        // 16 waves per workgroup (four per SIMD) for one and two tiles per site: 0.330 -> 0.290 and 0.656 -> 0.633 ms per
        // launch; with three / four tiles the 128-register limit spills (1.06 -> 1.13, 1.33 -> 1.72 ms): 12 waves there
#define NNJ_IA(NG, NW)                                                                                          \
  case NG: {                                                                                                    \
    const size_t lds = (size_t)(2 * IMG64 + NW * 512 * NPL + 16 + SCORER_CONSTS) * sizeof(float);              \
    if (int rc = set_lds(h, (k_inc_alpha16<NG, NW>), lds)) return rc;                                           \
    hipLaunchKernelGGL((k_inc_alpha16<NG, NW>), grid, dim3(64 * NW), lds, st, rs, sw, ij_prev, base + w.alpha_part, n, C, \
                       g.cs, h->d_flag);                                                                        \
  } break;
        switch (ng) { NNJ_IA(1, 16) NNJ_IA(2, 16) NNJ_IA(3, 12) NNJ_IA(4, 12) }
#undef NNJ_IA
      }
      launch_beta_sum(h, rs, base, w, B, C, st);             // (a scope of its own: profiling scopes do not nest)
      {
        Scope sc(h, st, PK_ALPHA_SOFTMAX);
        hipLaunchKernelGGL(k_alpha_softmax, dim3((unsigned)(g.ppad / 4), (unsigned)B), dim3(256), 0, st, rs, sw, ij_prev,
                           base + w.alpha_part, base + w.alpha, mode, n, C, g.npairs, g.ppad, g.nsc_a,
                           (const float*)(base + w.beta_slot), row_slots(rs, C));
      }
    }
    {
      Scope sc(h, st, PK_PAIR_SCORE_INCR);
      if (n <= 16) {
        // 16-pair tiles: half the padding of the 32-pair kernels (measured 2x faster here); above 16 rows the
        // group barriers of the shared image cost more than the padding saves
        const size_t lds = (size_t)(3 * IMG64 + T16_WAVES * 512 * NPL + 16 + SCORER_CONSTS) * sizeof(float);
        if (has_ctx) {
          if (int rc = set_lds(h, k_inc_score16<1, true>, lds)) return rc;
          hipLaunchKernelGGL((k_inc_score16<1, true>), grid, blk16, lds, st, rs, sw, ij_prev, base + w.alpha, mask,
                             base + w.score_part, n, C, g.cs, 0);
        } else {
          if (int rc = set_lds(h, k_inc_score16<1, false>, lds)) return rc;
          hipLaunchKernelGGL((k_inc_score16<1, false>), grid, blk16, lds, st, rs, sw, ij_prev, base + w.alpha, mask,
                             base + w.score_part, n, C, g.cs, 0);
        }
      } else if (n > 32 && n <= 48) {                      // three 16-row tiles: 48 instead of 64 padded pairs,
        const size_t lds = (size_t)(3 * IMG64 + 8 * (64 * 48 * NPL / 2) + SCORER_CONSTS + 3 * 1024) * sizeof(float);   // one wave per site; + alpha pieces
        if (int rc = set_lds(h, k_inc_score_w<3, true>, lds)) return rc;
        hipLaunchKernelGGL((k_inc_score_w<3, true>), grid, dim3(512), lds, st, rs, sw, ij_prev, base + w.alpha, mask,
                           base + w.score_part, n, C, g.cs, 0);
      } else if (n > 32) {
        const size_t lds = (size_t)(3 * IMG64 + 4 * b6_floats(64, 64) + 16 + SCORER_CONSTS) * sizeof(float);
        if (int rc = set_lds(h, k_inc_score<2, true>, lds)) return rc;
        hipLaunchKernelGGL((k_inc_score<2, true>), grid, dim3(512), lds, st, rs, sw, ij_prev, base + w.alpha, mask,
                           base + w.score_part, n, C, g.cs, h->d_flag, 0);
      } else {                                             // 17..32: the 32-pair kernel (k_inc_score_w<2> with 8 or 12 waves measured 2-3 ms per rollout slower)
        const size_t lds = (size_t)(3 * IMG64 + 8 * b6_floats(64, 32) + 16 + SCORER_CONSTS) * sizeof(float);
        if (int rc = set_lds(h, k_inc_score<1, true>, lds)) return rc;
        hipLaunchKernelGGL((k_inc_score<1, true>), grid, dim3(512), lds, st, rs, sw, ij_prev, base + w.alpha, mask,
                           base + w.score_part, n, C, g.cs, h->d_flag, 0);
      }
    }
    return NNJ_OK;
  }
  const dim3 grid(pair_grid(g.nsc, g.pg, B));       // XCD-aware 1-D mapping (nnj_scorer.hpp pair_block)
  if (has_ctx) {
    {
      Scope sc(h, st, PK_PAIR_ALPHA);
      const size_t lds = (2 * (IMG64 + 8192) + SCORER_CONSTS) * sizeof(float);
      // (k_pair_alpha<1, 8> and k_pair_score<1, 8> live in nnj_step0_tu.hip)
      HIPCHK(h, nnj_launch_pair_alpha_1_8(grid.x, lds, st, &rs, &sw, ij_prev, base + w.alpha_part, mode, n, C, g.npairs, g.ppad,
                                          g.cs, g.nsc, g.pg, B));
    }
    launch_beta_sum(h, rs, base, w, B, C, st);
    {
      Scope sc(h, st, PK_ALPHA_SOFTMAX);
      hipLaunchKernelGGL(k_alpha_softmax, dim3((unsigned)(g.ppad / 4), (unsigned)B), dim3(256), 0, st, rs, sw, ij_prev,
                         base + w.alpha_part, base + w.alpha, mode, n, C, g.npairs, g.ppad, g.nsc,
                         (const float*)(base + w.beta_slot), row_slots(rs, C));
    }
  }
  {
    Scope sc(h, st, PK_PAIR_SCORE);
    const size_t lds = (size_t)(2 * IMG64 + 2 * (IMG64 + 8192) + SCORER_CONSTS) * sizeof(float);
    HIPCHK(h, nnj_launch_pair_score_1_8(grid.x, lds, st, &rs, &sw, ij_prev, base + w.alpha, mask, base + w.score_part, mode, n, C,
                                        g.npairs, g.ppad, g.cs, has_ctx, g.nsc, g.pg, B));
  }
  return NNJ_OK;
}

// ---- the two-pass NJ step (nnj_step2.hpp).  `rs` lists the rows AFTER the merge of `ij` (n rows, merged row at
// position ij[0], whose slot still holds S_i); live_old = the list before it.  Produces the merged row in place and
// the score partials of the n-1 new pairs (q numbering) in w.score_part.
struct Step2 {
  bool first;         // the previous step ran the other kernels: the per-row biases are summed from their partials
  bool fallback;      // the previous table kernel may have left `need` set: launch the per-alignment fallback
  bool cand;          // carry the candidate pair's logits along
  int* cand_cur;      // [B][2]
};
int launch_step2(nnj_handle* h, RowSet rs, const int* live_old, const int* ij, const uint8_t* mask, float* base,
                 const LoopWs& w, const Step2& o, int n, int B, int C, PairGeom& g, hipStream_t st) {
  const ScorerW sw = scorer_ptrs(h);
  float* S_w = const_cast<float*>(rs.S);
  float* U_w = const_cast<float*>(rs.U);
  g = pair_geom(PAIRS_INCR, n, B, C);
  if (int rc = check_pair_fit(h, w, g, B, "NJ-step launch")) return rc;
  g.score_src = base + w.score_part;
  int* need = reinterpret_cast<int*>(base + w.need);
  int* cand_run = reinterpret_cast<int*>(base + w.cand_run);
  const int T = rs.live_stride;
  if (o.first) {
    Scope sc(h, st, PK_STEP_SMALL);
    hipLaunchKernelGGL(k_beta_sum, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, st, rs.beta_part, rs.ntile32,
                       rs.ntile32, (float)C * sw.t0, base + w.beta_slot, B * T);
  }
  if (o.fallback) {
    // weights of the merge for the alignments whose pick was neither a pair of the last step nor the candidate
    Scope sc(h, st, PK_STEP_SMALL);
    RowSet ro = rs;
    ro.live = live_old;
    const size_t lds = (size_t)IMG64 * sizeof(float);
    hipLaunchKernelGGL(k_pair_xp, dim3((unsigned)((C + 127) / 128), (unsigned)B), dim3(256), lds, st, rs.S, rs.U, rs.bstride,
                       live_old, T, sw, ij, (const int*)need, base + w.merged, n + 1, C);
    const int nch = (C + 15) / 16, rp = T > 128 ? 256 : (T > 64 ? 128 : 64);
    hipLaunchKernelGGL(k_agg_dot, dim3((unsigned)nch, (unsigned)B), dim3(256), 0, st, ro, (const float*)(base + w.merged),
                       (const int*)need, base + w.agg_part, n + 1, C, rp);
    hipLaunchKernelGGL(k_agg_am, dim3((unsigned)B), dim3(64), 0, st, ro, sw, ij, (const int*)need,
                       (const float*)(base + w.agg_part), nch, rp, (const float*)(base + w.beta_slot), base + w.am, n + 1, C);
  }
  const bool cand = o.cand && n > 2;
  StepIO io;
  io.ij = ij; io.live_old = live_old; io.am = base + w.am; io.alpha_part = base + w.alpha_part;
  io.S_w = S_w; io.U_w = U_w; io.beta_w = const_cast<float*>(rs.beta_part); io.beta_n = rs.ntile32;
  io.cand = cand ? o.cand_cur : nullptr; io.cand_run = cand_run; io.Xc = base + w.Xc; io.acand_part = base + w.acand;
  const int np = n - 1;                                    // pairs = rows other than the merged one
  const int ng = np > 48 ? 4 : (np > 32 ? 3 : (np > 16 ? 2 : 1));
  const dim3 grid((unsigned)g.blocks, (unsigned)B);
  {
    Scope sc(h, st, PK_PAIR_ALPHA_INCR);
#define NNJ_SA(NG, NW)                                                                                          \
  case NG: {                                                                                                    \
    const size_t lds = (size_t)(3 * IMG64 + NW * 1024 + (NW / NG) * (64 * NG + 64 * 6) + 16 + SCORER_CONSTS +   \
                                (NW / NG) * 66 + 16) * sizeof(float);                                           \
    if (int rc = set_lds(h, (k_step_alpha<NG, NW>), lds)) return rc;                                            \
    hipLaunchKernelGGL((k_step_alpha<NG, NW>), grid, dim3(64 * NW), lds, st, rs, sw, io, n, C, g.cs, h->d_flag); \
  } break;
#define NNJ_SW(NT, NW, IL)                                                                                      \
  {                                                                                                             \
    const size_t lds = (size_t)(3 * IMG64 + NW * (1024 * NT + 64 * 5) + SCORER_CONSTS + NW * 66 + 16) * sizeof(float); \
    if (int rc = set_lds(h, (k_step_alpha_w<NT, NW, IL>), lds)) return rc;                                      \
    hipLaunchKernelGGL((k_step_alpha_w<NT, NW, IL>), grid, dim3(64 * NW), lds, st, rs, sw, io, n, C, g.cs);     \
  }
    // Tiers (every arm that was measured and lost is in the git history and under profiles/r04, r05):
    //   <= 4 pairs / 5..8 pairs with enough sites per wave: shared tiles, four / two sites per tile group (nnj_step_g.hpp);
    //   17..32 pairs: one wave per site, its two tiles stage by stage; 33..48: one wave per site, tile by tile;
    //   everything else (<= 16 pairs in small workgroups, more than 48 pairs): groups of waves share a site (k_step_alpha).
#define NNJ_AG(NT, G, IR, NW)                                                                                   \
  {                                                                                                            \
    const size_t lds = (size_t)step_alpha_g_lds(NT, G, IR, NW) * sizeof(float);                                \
    if (int rc = set_lds(h, (k_step_alpha_g<NT, G, IR, NW>), lds)) return rc;                                  \
    hipLaunchKernelGGL((k_step_alpha_g<NT, G, IR, NW>), grid, dim3(64 * NW), lds, st, rs, sw, io, n, C, g.cs); \
  }
    const bool grp2 = g.cs >= 16, grp4 = g.cs >= 32;      // (see the score kernels below)
    if (n > 2 && np <= 4 && grp4) NNJ_AG(1, 4, 8, 8)
    else if (n > 2 && np > 4 && np <= 8 && grp2) NNJ_AG(1, 2, 8, 8)
    else
#undef NNJ_AG
    if (h->step_w && ng == 2) NNJ_SW(2, 8, true)
    else if (h->step_w && ng == 3) NNJ_SW(3, 8, false)
    else {
      switch (ng) { NNJ_SA(1, 12) NNJ_SA(2, 12) NNJ_SA(3, 12) NNJ_SA(4, 12) }
    }
#undef NNJ_SW
#undef NNJ_SA
  }
  const bool has_ctx = n > 2;
  if (has_ctx) {
    Scope sc(h, st, PK_ALPHA_SOFTMAX);
    if (g.blocks > 8)
      hipLaunchKernelGGL(k_step_softmax<true>, dim3(64, (unsigned)B), dim3(256), 0, st, rs, sw, ij, (const float*)(base + w.alpha_part),
                         base + w.alpha, base + w.lam, base + w.beta_slot, g.blocks, n, C);
    else
      hipLaunchKernelGGL(k_step_softmax<false>, dim3(16, (unsigned)B), dim3(256), 0, st, rs, sw, ij, (const float*)(base + w.alpha_part),
                         base + w.alpha, base + w.lam, base + w.beta_slot, g.blocks, n, C);
  }
  {
    Scope sc(h, st, PK_PAIR_SCORE_INCR);
    const dim3 blk16(64 * T16_WAVES);
    // Tiers: shared 16-pair tiles where the pairs of one site leave a tile part empty and a wave has enough
    // sites (nnj_scorer_g.hpp: <= 4 pairs four sites per tile, 5..8 two, 17..24 three tiles per two sites, 33..40 five tiles
    // per two sites); otherwise one wave per site (17..32: two tiles stage by stage; 33..48: tile by tile), 16-pair tiles
    // with one wave each up to 16 pairs, the 32-pair kernel above 48.
#define NNJ_SG(NT, G, IR, NW, ...)                                                                              \
  {                                                                                                            \
    const size_t lds = (size_t)inc_score_g_lds(NT, G, IR, NW) * sizeof(float);                                 \
    if (int rc = set_lds(h, (k_inc_score_g<NT, G, IR, NW, ##__VA_ARGS__>), lds)) return rc;                    \
    hipLaunchKernelGGL((k_inc_score_g<NT, G, IR, NW, ##__VA_ARGS__>), grid, dim3(64 * NW), lds, st, rs, sw, ij, \
                       base + w.alpha, mask, base + w.score_part, n, C, g.cs);                                 \
  }
    // (a wave of these kernels walks the G sites of a group one after the other: where a workgroup has fewer than G sites
    // per wave -- one alignment per rollout: 4 sites per workgroup -- the one-site-per-wave kernels are the shorter path)
    const bool grp2 = g.cs >= 16, grp4 = g.cs >= 32;
    if (has_ctx && np <= 4 && grp4) NNJ_SG(1, 4, 8, 8)
    else if (has_ctx && np <= 8 && np > 4 && grp2) NNJ_SG(1, 2, 8, 12)
    else if (np > 16 && np <= 24 && grp2) NNJ_SG(3, 2, 24, 8)
    else if (np > 32 && np <= 40 && grp2) {
      // 33..40 pairs: five tiles per two sites through one image buffer in turn (k_inc_score_s5)
      const size_t lds = (size_t)inc_score_g_lds(3, 1, 48, 8) * sizeof(float);
      if (int rc = set_lds(h, (k_inc_score_s5<8>), lds)) return rc;
      hipLaunchKernelGGL((k_inc_score_s5<8>), grid, dim3(512), lds, st, rs, sw, ij, base + w.alpha, mask,
                         base + w.score_part, n, C, g.cs);
    }
    else
#undef NNJ_SG
    if (np <= 16) {
      const size_t lds = (size_t)(3 * IMG64 + T16_WAVES * 512 * NPL + 16 + SCORER_CONSTS) * sizeof(float);
      if (has_ctx) {
        if (int rc = set_lds(h, k_inc_score16<1, true>, lds)) return rc;
        hipLaunchKernelGGL((k_inc_score16<1, true>), grid, blk16, lds, st, rs, sw, ij, base + w.alpha, mask,
                           base + w.score_part, n, C, g.cs, 1);
      } else {
        if (int rc = set_lds(h, k_inc_score16<1, false>, lds)) return rc;
        hipLaunchKernelGGL((k_inc_score16<1, false>), grid, blk16, lds, st, rs, sw, ij, base + w.alpha, mask,
                           base + w.score_part, n, C, g.cs, 1);
      }
    } else if (np > 32 && np <= 48) {
      // three tiles, tile by tile (stage by stage they need ~300 registers: all three at once spills at two waves per SIMD
      // -- profiles/r04/ab_stage_by_stage.txt -- and at ONE wave per SIMD with 512 registers the lost second wave costs
      // more than the interleave gains: 52.0 -> 55.5 ms per rollout, profiles/r05/ab_one_wave_three_tiles.txt)
      const size_t lds = (size_t)(3 * IMG64 + 8 * (64 * 48 * NPL / 2) + SCORER_CONSTS + 3 * 1024) * sizeof(float);
      if (int rc = set_lds(h, k_inc_score_w<3, true>, lds)) return rc;
      hipLaunchKernelGGL((k_inc_score_w<3, true>), grid, dim3(512), lds, st, rs, sw, ij, base + w.alpha, mask,
                         base + w.score_part, n, C, g.cs, 1);
    } else if (np > 32) {
      const size_t lds = (size_t)(3 * IMG64 + 4 * b6_floats(64, 64) + 16 + SCORER_CONSTS) * sizeof(float);
      if (int rc = set_lds(h, k_inc_score<2, true>, lds)) return rc;
      hipLaunchKernelGGL((k_inc_score<2, true>), grid, dim3(512), lds, st, rs, sw, ij, base + w.alpha, mask,
                         base + w.score_part, n, C, g.cs, h->d_flag, 1);
    } else {
      // 17..32 pairs: one wave per site walks its two 16-pair tiles stage by stage (k_inc_score_wi<2>)
      const size_t lds = (size_t)(3 * IMG64 + 8 * (64 * 32 * NPL / 2) + SCORER_CONSTS + 2 * 1024) * sizeof(float);
      if (int rc = set_lds(h, (k_inc_score_wi<2>), lds)) return rc;
      hipLaunchKernelGGL((k_inc_score_wi<2>), grid, dim3(512), lds, st, rs, sw, ij, base + w.alpha, mask,
                         base + w.score_part, n, C, g.cs, 1);
    }
  }
  return NNJ_OK;
}

int launch_aggregate(nnj_handle* h, const RowSet& rs, const int* ij, float* base, const LoopWs& w, float* S_out,
                     float* U_out, float* Kp_out, float* beta_out, long out_bstride, int out_slots, int in_place,
                     int n, int B, int C, hipStream_t st) {
  const ScorerW sw = scorer_ptrs(h);
  const int nch = (C + 15) / 16;
  const int rpt = n <= 64 ? 1 : (n <= 128 ? 2 : 4);             // 64-row groups of the alpha vector
  if (n > 2) {
    Scope sc(h, st, PK_AGG_ALPHA);
    hipLaunchKernelGGL(k_agg_alpha, dim3((unsigned)nch, (unsigned)B), dim3(256), 0, st, rs, sw, ij, base + w.agg_part, n, C,
                       64 * rpt);
  }
  {
    Scope sc(h, st, PK_AGG_FINISH);
    const bool split = (long)B * ((C + 127) / 128) < (long)h->num_cu;   // small batches: one 32-site tile per workgroup, rows split over its waves
    const size_t lds = (size_t)(3 * 4096 + 256 + (split ? 8192 : 1024)) * sizeof(float);
    const dim3 grid(split ? (unsigned)((C + 31) / 32) : (unsigned)((C + 127) / 128), (unsigned)B);
#define NNJ_AGG(SP, RPT)                                                                                       \
  {                                                                                                            \
    if (int rc = set_lds(h, k_agg_finish<SP, RPT>, lds)) return rc;                                            \
    hipLaunchKernelGGL((k_agg_finish<SP, RPT>), grid, dim3(256), lds, st, rs, sw, ij, base + w.agg_part, nch, S_out, \
                       U_out, Kp_out, beta_out, out_bstride, out_slots, in_place, n, C);                       \
  }
    if (split) { if (rpt == 1) NNJ_AGG(true, 1) else if (rpt == 2) NNJ_AGG(true, 2) else NNJ_AGG(true, 4) }
    else { if (rpt == 1) NNJ_AGG(false, 1) else if (rpt == 2) NNJ_AGG(false, 2) else NNJ_AGG(false, 4) }
#undef NNJ_AGG
  }
  return NNJ_OK;
}

// dense-state helper for the API-compatible entry points: transforms of all n rows
RowSet dense_rowset(nnj_handle* h, const float* state, float* base, const LoopWs& w, int B, int n, int C,
                    hipStream_t st) {
  RowSet rs;
  rs.S = state; rs.U = base + w.U; rs.Kp = base + w.Kp; rs.beta_part = base + w.beta;
  rs.bstride = (long)n * C * 64; rs.live = nullptr; rs.live_stride = 0; rs.ntile32 = beta_stride(B, C);
  launch_row_xf(h, state, base + w.U, base + w.Kp, base + w.beta, rs.bstride, n, n, B, C, st);
  return rs;
}

// ---- dense state tensors of a model narrower than 64 features (nnj_create): [.., embed_dim] at the C ABI, [.., 64] with
// zero padding inside.  Two staging tensors of the workspace's whole-state size sit behind the regular workspace.
__global__ void k_widen(const float* __restrict__ in, float* __restrict__ out, long ntok, int dt) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;                  // one float4 of the wide tensor
  if (i >= ntok * 16) return;
  const long tok = i >> 4; const int f = (int)(i & 15) * 4;
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  if (f < dt) v = *reinterpret_cast<const f32x4*>(in + tok * dt + f);           // dt is a multiple of 8
  *reinterpret_cast<f32x4*>(out + tok * 64 + f) = v;
}
__global__ void k_narrow(const float* __restrict__ in, float* __restrict__ out, long ntok, int dt) {
  const int q = dt / 4;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;                  // one float4 of the narrow tensor
  if (i >= ntok * q) return;
  const long tok = i / q; const int f = (int)(i % q) * 4;
  *reinterpret_cast<f32x4*>(out + tok * dt + f) = *reinterpret_cast<const f32x4*>(in + tok * 64 + f);
}
inline bool narrow_model(const nnj_handle* h) { return h->cfg.embed_dim != NNJ_D; }
size_t wide_floats(int B, int T, int C) { return align_up((size_t)B * T * C * 64, 64); }
size_t ws_floats_model(const nnj_handle* h, int B, int T, int C) {
  return ws_floats(B, T, C) + (narrow_model(h) ? 2 * wide_floats(B, T, C) : 0);
}
// staging tensor k (0: inputs, 1: outputs) of a workspace laid out for (B, T, C)
float* wide_buf(void* ws, int B, int T, int C, int k) {
  return static_cast<float*>(ws) + ws_floats(B, T, C) + (size_t)k * wide_floats(B, T, C);
}
void widen_into(const nnj_handle* h, const float* in, float* out, long ntok, hipStream_t st) {
  hipLaunchKernelGGL(k_widen, dim3((unsigned)((ntok * 16 + 255) / 256)), dim3(256), 0, st, in, out, ntok, h->cfg.embed_dim);
}
// the tensor the kernels read for a caller's dense input of ntok tokens
const float* dense_in(const nnj_handle* h, const float* user, long ntok, void* ws, int B, int T, int C, hipStream_t st) {
  if (!narrow_model(h)) return user;
  float* w = wide_buf(ws, B, T, C, 0);
  widen_into(h, user, w, ntok, st);
  return w;
}
// the tensor the kernels write for a caller's dense output, and the copy that hands it over afterwards
float* dense_out(const nnj_handle* h, float* user, void* ws, int B, int T, int C) {
  return narrow_model(h) ? wide_buf(ws, B, T, C, 1) : user;
}
void dense_out_done(const nnj_handle* h, const float* wide, float* user, long ntok, hipStream_t st) {
  if (!narrow_model(h)) return;
  const long n4 = ntok * (h->cfg.embed_dim / 4);
  hipLaunchKernelGGL(k_narrow, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, wide, user, ntok, h->cfg.embed_dim);
}

int need_ws(nnj_handle* h, void* ws, size_t ws_bytes, int B, int T, int C) {
  h->sess.valid = false;                 // every workspace user overwrites the regions a step session lives in
  if (!ws) return fail(h, NNJ_ERR_ARG, "workspace pointer is null");
  const size_t need = ws_floats_model(h, B, T, C) * sizeof(float);
  if (ws_bytes < need) return fail(h, NNJ_ERR_WORKSPACE, "workspace too small: %zu < %zu bytes", ws_bytes, need);
  if ((uintptr_t)ws % 256) return fail(h, NNJ_ERR_ARG, "workspace must be 256-byte aligned");
  return NNJ_OK;
}

// sites -> tokens: a token is patch_size consecutive sites (reference model.py:72-76); the kernels work on tokens
int to_tokens(nnj_handle* h, int L_sites, int32_t* C) {
  const int K = h->cfg.patch_size;
  if (L_sites <= 0 || L_sites % K) return fail(h, NNJ_ERR_ARG, "L=%d sites is not a positive multiple of patch_size=%d", L_sites, K);
  *C = L_sites / K;
  return NNJ_OK;
}
// the mask of the tokens: the caller's [B, L] mask as it is for patch_size 1, else mask[:, ::patch_size] (reference
// model.py:79, 167) written to the tail of the (already validated) workspace
int token_mask(nnj_handle* h, const uint8_t** mask, void* ws, int B, int T, int C, hipStream_t st) {
  const int K = h->cfg.patch_size;
  if (K == 1 || !*mask) return NNJ_OK;
  uint8_t* pm = reinterpret_cast<uint8_t*>(static_cast<float*>(ws) + ws_floats(B, T, C) - pm_floats(B, C));
  hipLaunchKernelGGL(k_patch_mask, dim3((unsigned)(((size_t)B * C + 255) / 256)), dim3(256), 0, st, *mask, pm, B, C, K);
  *mask = pm;
  return NNJ_OK;
}

// ---- sessions over the dense-state entry points (nnj_pair_scores_full / _incr, nnj_aggregate, nnj_env_step, nnj_step)
// The reference's loop hands the dense state [B,n,C,D] from call to call.  The library keeps the rows of the tensor
// it has last SEEN (nnj_pair_scores_full) or RETURNED (nnj_env_step, nnj_step) in slot layout inside the caller's
// workspace, next to their cached transforms (U, K', beta) and the live list -- the rollout's own state.  A call
// whose `state` is that very tensor (same pointer, workspace, B, L and row count; the caller has not written to it)
// continues the session: no row is transformed again and the merge runs in place, like inside nnj_rollout_argmax.
// Anything else runs stateless from the dense tensor, as before.
struct SessView { float* S; float* base; LoopWs w; int* live; int* ijs; RowSet rs; };
bool sess_matches(const nnj_handle* h, const void* ws, const float* state, int B, int n, int L, bool need_kp = true) {
  const nnj_handle::StepSession& ss = h->sess;
  return ss.valid && ss.ws == ws && ss.last_out == state && ss.B == B && ss.L == L && ss.n == n && (ss.kp_valid || !need_kp);
}
SessView sess_view(void* ws, int B, int T0, int C, int live_idx = 0) {
  SessView v;
  v.S = static_cast<float*>(ws);
  v.base = v.S + align_up((size_t)B * T0 * C * 64, 64);
  v.w = loop_ws(B, T0, C);
  v.live = reinterpret_cast<int*>(v.base + v.w.live) + (size_t)live_idx * B * T0;
  v.ijs = reinterpret_cast<int*>(v.base + v.w.ij);
  v.rs.S = v.S; v.rs.U = v.base + v.w.U; v.rs.Kp = v.base + v.w.Kp; v.rs.beta_part = v.base + v.w.beta;
  v.rs.bstride = (long)T0 * C * 64; v.rs.live = v.live; v.rs.live_stride = T0; v.rs.ntile32 = beta_stride(B, C);
  return v;
}
// start a session from a dense tensor of T0 rows: copy into the slots, identity live list, row transforms
int sess_begin(nnj_handle* h, const SessView& v, const float* state, int B, int T0, int C, hipStream_t st) {
  h->sess.live_idx = 0; h->sess.cand_idx = 0; h->sess.tp = false; h->sess.kp_valid = true;
  if (narrow_model(h)) widen_into(h, state, v.S, (long)B * T0 * C, st);
  else HIPCHK(h, hipMemcpyAsync(v.S, state, (size_t)B * T0 * C * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
  {
    Scope sc(h, st, PK_MISC);
    hipLaunchKernelGGL(k_init_live, dim3((unsigned)((B * T0 + 255) / 256)), dim3(256), 0, st, v.live, T0, B, T0);
  }
  return launch_row_xf(h, v.S, v.base + v.w.U, v.base + v.w.Kp, v.base + v.w.beta, (long)T0 * C * 64, T0, T0, B, C, st);
}
void sess_set(nnj_handle* h, const void* ws, const float* dense, int B, int T0, int L, int n) {
  nnj_handle::StepSession& ss = h->sess;                  // (live_idx, cand_idx, tp, kp_valid: kept)
  ss.valid = true; ss.ws = ws; ss.last_out = dense; ss.B = B; ss.T0 = T0; ss.L = L; ss.n = n;
}

// Every entry point selects the handle's device; the caller's current device (torch's, in a single process that drives
// several GPUs) is put back when the call returns.
struct DevGuard {
  int prev = -1;
  DevGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
  ~DevGuard() { int cur = -1; if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev); }
};

int ready(nnj_handle* h) {
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  if (!h->have_w) return fail(h, NNJ_ERR_NO_WEIGHTS, "weights not loaded");
  HIPCHK(h, hipSetDevice(h->cfg.device));
  return NNJ_OK;
}

}  // namespace

// A model of embed_dim Dt < 64 (heads of 8 features) as a 64-feature model: every matrix and vector of the packed weights
// zero-padded.  Exact: padded features are zero in every tensor of the path (a Linear's padded outputs are 0 + bias 0,
// GELU(0) = 0, the gates mix 0 with 0, LayerNorm leaves them out of its statistics -- layer_norm64 -- and scales them by
// gamma = 0); the padded heads attend uniformly over values that are zero.  The one place the width itself enters is the
// scale of the aggregate's logits, 1 / sqrt(embed_dim * patch_num) (model.py:110): the kernels divide by sqrt(64 * patch_num),
// so g_attn_q (weight and bias) carries sqrt(64 / Dt).
void pad_weights(const nnj_config& c, const float* p, std::vector<float>& out) {
  nnj_config c64 = c;
  c64.embed_dim = NNJ_D; c64.num_heads = NNJ_NHEAD;
  out.assign(count_params(c64), 0.f);
  const size_t Dt = c.embed_dim, Ft = 4 * Dt, D = NNJ_D, F = NNJ_F, K4 = 4 * (size_t)c.patch_size;
  size_t src = 0, dst = 0;
  auto mat = [&](size_t rt, size_t ct, size_t rp, size_t cp, double f = 1.0) {
    for (size_t r = 0; r < rt; ++r)
      for (size_t q = 0; q < ct; ++q) out[dst + r * cp + q] = (float)(f * (double)p[src + r * ct + q]);
    src += rt * ct; dst += rp * cp;
  };
  auto lin = [&](double f = 1.0) { mat(Dt, Dt, D, D, f); mat(1, Dt, 1, D, f); };
  for (int l = 0; l < c.num_layers; ++l) {
    for (int a = 0; a < 2; ++a) {
      for (int k = 0; k < 4; ++k) lin();
      mat(1, Dt, 1, D); mat(1, Dt, 1, D);
    }
    mat(Ft, Dt, F, D); mat(1, Ft, 1, F); mat(Dt, Ft, D, F); mat(1, Dt, 1, D); mat(1, Dt, 1, D); mat(1, Dt, 1, D);
  }
  mat(Dt, K4, D, K4); mat(1, Dt, 1, D); lin();             // embed
  lin(); lin();                                             // h_linear_last, g_linear_last
  lin(sqrt((double)D / (double)Dt));                        // g_attn_q
  lin();                                                    // g_attn_k
  lin(); mat(1, Dt, 1, D); mat(1, 1, 1, 1);                 // s_out
}

// ====================================================================== C ABI
extern "C" {

int nnj_abi_version(void) { return NNJ_ABI_VERSION; }

int nnj_num_params(const nnj_config* cfg, size_t* n) {
  if (!cfg || !n) return fail(nullptr, NNJ_ERR_ARG, "nnj_num_params: null argument");
  *n = count_params(*cfg);
  return NNJ_OK;
}

int nnj_create(const nnj_config* cfg, nnj_handle** out) {
  if (!cfg || !out) return fail(nullptr, NNJ_ERR_ARG, "nnj_create: null argument");
  // embed_dim: 64 natively; a narrower model (a multiple of 8 with heads of 8 features, e.g. the reference's own defaults
  // utils.py:45-52: 32 features, 4 heads) runs on the same kernels zero-padded to 64 features / 8 heads (nnj_load_weights)
  if (cfg->embed_dim < 8 || cfg->embed_dim > NNJ_D || cfg->embed_dim % 8 || cfg->num_heads * NNJ_DH != cfg->embed_dim ||
      cfg->patch_size < 1 || cfg->patch_size > 16 || cfg->vocab_size != 4 || cfg->num_layers < 0)
    return fail(nullptr, NNJ_ERR_UNSUPPORTED,
                "this build covers embed_dim 8..64 (multiples of 8) with num_enc_heads = embed_dim / 8, patch_size 1..16, "
                "vocab_size=4 (got %d,%d,%d,%d)",
                cfg->embed_dim, cfg->num_heads, cfg->patch_size, cfg->vocab_size);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, NNJ_ERR_NO_DEVICE, "no HIP device visible");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, NNJ_ERR_ARG, "device %d out of range (%d devices)", cfg->device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, NNJ_ERR_HIP, "hipGetDeviceProperties failed");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, NNJ_ERR_NO_DEVICE, "device %d is %s; libnnj_hip.so is built for gfx950 only", cfg->device, prop.gcnArchName);
  nnj_handle* h = new nnj_handle();
  h->cfg = *cfg;
  if (const char* e = getenv("NNJ_TWO_PASS")) h->two_pass = atoi(e);
  if (const char* e = getenv("NNJ_TWO_PASS_CAND")) h->two_pass_cand = atoi(e);
  if (const char* e = getenv("NNJ_STEP_W")) h->step_w = atoi(e);
  if (const char* e = getenv("NNJ_ENC64")) h->enc64 = atoi(e);
  if (const char* e = getenv("NNJ_GRAPH")) h->use_graph = atoi(e);     // NNJ_GRAPH=0: small-batch rollouts are never replayed from a hipGraph
  h->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipSetDevice(cfg->device) != hipSuccess || hipMalloc(&h->d_flag, sizeof(int)) != hipSuccess ||
      hipMemset(h->d_flag, 0, sizeof(int)) != hipSuccess) {
    delete h;
    return fail(nullptr, NNJ_ERR_HIP, "nnj_create: device allocation failed");
  }
  *out = h;
  return NNJ_OK;
}

int nnj_destroy(nnj_handle* h) {
  if (!h) return NNJ_OK;
  hipSetDevice(h->cfg.device);
  if (h->d_w) hipFree(h->d_w);
  if (h->d_wimg) hipFree(h->d_wimg);
  if (h->d_w64) hipFree(h->d_w64);
  if (h->d_flag) hipFree(h->d_flag);
  if (h->gexec) hipGraphExecDestroy(h->gexec);
  for (int k = 0; k < 4; ++k) {
    if (h->sub[k]) hipStreamDestroy(h->sub[k]);
    if (h->ev_join[k]) hipEventDestroy(h->ev_join[k]);
  }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  for (hipEvent_t e : h->ev) hipEventDestroy(e);
  delete h;
  return NNJ_OK;
}

const char* nnj_last_error(const nnj_handle* h) { return h ? h->err : g_err; }

int nnj_load_weights(nnj_handle* h, const float* p, size_t n) {
  if (!h || !p) return fail(h, NNJ_ERR_ARG, "nnj_load_weights: null argument");
  size_t need = count_params(h->cfg);
  if (n != need) return fail(h, NNJ_ERR_ARG, "nnj_load_weights: got %zu floats, config needs %zu", n, need);
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  const size_t D = NNJ_D, F = NNJ_F;
  std::vector<float> widened;
  if (h->cfg.embed_dim != NNJ_D) {           // a narrower model: every tensor zero-padded into the 64-feature layout
    pad_weights(h->cfg, p, widened);
    p = widened.data();
    need = widened.size();
  }
  size_t o = 0;
  auto take = [&](size_t k) { size_t r = o; o += k; return r; };
  h->lo.assign(h->cfg.num_layers, LayerOff());
  for (int l = 0; l < h->cfg.num_layers; ++l) {
    LayerOff& L = h->lo[l];
    for (int a = 0; a < 2; ++a) {
      size_t* t = a == 0 ? L.row : L.col;
      for (int k = 0; k < 4; ++k) { t[2 * k] = take(D * D); t[2 * k + 1] = take(D); }
      t[8] = take(D); t[9] = take(D);
    }
    L.W1 = take(F * D); L.b1 = take(F); L.W2 = take(D * F); L.b2 = take(D); L.ln_w = take(D); L.ln_b = take(D);
  }
  const size_t K = (size_t)h->cfg.patch_size;
  h->oE0 = take(D * 4 * K); h->oe0 = take(D); h->oE2 = take(D * D); h->oe2 = take(D);
  h->oWh = take(D * D); h->obh = take(D); h->oWg = take(D * D); h->obg = take(D);
  h->oWgq = take(D * D); h->obgq = take(D); h->oWgk = take(D * D); h->obgk = take(D);
  h->oS0 = take(D * D); h->os0 = take(D); h->os2w = take(D); h->os2b = take(1);
  if (o != need) return fail(h, NNJ_ERR_ARG, "nnj_load_weights: internal layout mismatch");
  h->n_packed = need;
  // derived tensors, computed in double and rounded once (see nnj_scorer.hpp header)
  const size_t base = align_up(need, 64);
  h->oA = base; h->oa0 = base + D * D; h->ou = h->oa0 + D; h->olut = h->ou + D;
  h->oWhS = h->olut + 6 * D; h->obhS = h->oWhS + D * D; h->oWgS = h->obhS + D; h->obgS = h->oWgS + D * D;
  h->optab = h->obgS + D;                                  // [K][6][D] first-layer contributions of the sites of a patch
  const size_t total = h->optab + K * 6 * D;
  std::vector<float> host(total, 0.f);
  memcpy(host.data(), p, need * sizeof(float));
  const float *Wq = p + h->oWgq, *bq = p + h->obgq, *Wk = p + h->oWgk, *bk = p + h->obgk;
  for (size_t d1 = 0; d1 < D; ++d1) {
    for (size_t d2 = 0; d2 < D; ++d2) {
      double s = 0;
      for (size_t e = 0; e < D; ++e) s += (double)Wq[e * D + d1] * (double)Wk[e * D + d2];
      host[h->oA + d1 * D + d2] = (float)s;
    }
    double s0 = 0;
    for (size_t e = 0; e < D; ++e) s0 += (double)Wq[e * D + d1] * (double)bk[e];
    host[h->oa0 + d1] = (float)s0;
    double su = 0;
    for (size_t e = 0; e < D; ++e) su += (double)bq[e] * (double)Wk[e * D + d1];
    host[h->ou + d1] = (float)su;
  }
  {
    const double nl2e = -1.4426950408889634;           // sigmoid(t) = 1 / (1 + 2^(-log2(e) t))
    for (size_t i = 0; i < D * D; ++i) {
      host[h->oWhS + i] = (float)(nl2e * (double)p[h->oWh + i]);
      host[h->oWgS + i] = (float)(nl2e * (double)p[h->oWg + i]);
    }
    for (size_t i = 0; i < D; ++i) {
      host[h->obhS + i] = (float)(nl2e * (double)p[h->obh + i]);
      host[h->obgS + i] = (float)(nl2e * (double)p[h->obg + i]);
    }
  }
  double t0 = 0;
  for (size_t e = 0; e < D; ++e) t0 += (double)bq[e] * (double)bk[e];
  h->t0 = (float)t0;
  h->s2b = p[h->os2b];
  // embed LUT of the six site vectors (model.py:39-43 on phydata.py:38-46)
  static const int onehot[6][4] = {{1,0,0,0},{0,1,0,0},{0,0,1,0},{0,0,0,1},{1,1,1,1},{0,0,0,0}};
  const float *E0 = p + h->oE0, *e0 = p + h->oe0, *E2 = p + h->oE2, *e2 = p + h->oe2;
  for (size_t i = 0; i < K; ++i)                            // patch table: site i of a patch carrying `code` (+ the bias once)
    for (int code = 0; code < 6; ++code)
      for (size_t j = 0; j < D; ++j) {
        double s = i == 0 ? (double)e0[j] : 0.0;
        for (int v = 0; v < 4; ++v) s += (double)E0[j * 4 * K + 4 * i + v] * onehot[code][v];
        host[h->optab + (i * 6 + code) * D + j] = (float)s;
      }
  for (int code = 0; code < 6 && K == 1; ++code) {
    double t1[NNJ_D];
    for (size_t j = 0; j < D; ++j) {
      double s = e0[j];
      for (int v = 0; v < 4; ++v) s += (double)E0[j * 4 + v] * onehot[code][v];
      t1[j] = 0.5 * s * (1.0 + erf(s * 0.70710678118654752440));
    }
    for (size_t f = 0; f < D; ++f) {
      double s = e2[f];
      for (size_t j = 0; j < D; ++j) s += (double)E2[f * D + j] * t1[j];
      host[h->olut + code * D + f] = (float)s;
    }
  }
  // fp64 copy for the > 64-row encoder: the packed tensors converted exactly, the embed tables unrounded, stacked q|k|v
  std::vector<double> host64(host.begin(), host.end());
  for (size_t i = 0; i < K; ++i)
    for (int code = 0; code < 6; ++code)
      for (size_t j = 0; j < D; ++j) {
        double s = i == 0 ? (double)e0[j] : 0.0;
        for (int v = 0; v < 4; ++v) s += (double)E0[j * 4 * K + 4 * i + v] * onehot[code][v];
        host64[h->optab + (i * 6 + code) * D + j] = s;
      }
  for (int code = 0; code < 6 && K == 1; ++code) {
    double t1[NNJ_D];
    for (size_t j = 0; j < D; ++j) {
      double s = e0[j];
      for (int v = 0; v < 4; ++v) s += (double)E0[j * 4 + v] * onehot[code][v];
      t1[j] = 0.5 * s * (1.0 + erf(s * 0.70710678118654752440));
    }
    for (size_t f = 0; f < D; ++f) {
      double s = e2[f];
      for (size_t j = 0; j < D; ++j) s += (double)E2[f * D + j] * t1[j];
      host64[h->olut + code * D + f] = s;
    }
  }
  h->lo64.assign(h->cfg.num_layers, nnj_handle::Qkv64());
  for (int l = 0; l < h->cfg.num_layers; ++l)
    for (int a = 0; a < 2; ++a) {
      const size_t* t = a == 0 ? h->lo[l].row : h->lo[l].col;         // Wk,bk,Wv,bv,Wq,bq,Wo,bo,ln_w,ln_b
      const size_t oW = host64.size();
      for (int which : {4, 0, 2}) host64.insert(host64.end(), p + t[which], p + t[which] + D * D);
      const size_t ob = host64.size();
      for (int which : {5, 1, 3}) host64.insert(host64.end(), p + t[which], p + t[which] + D);
      if (a == 0) { h->lo64[l].Wrow = oW; h->lo64[l].brow = ob; } else { h->lo64[l].Wcol = oW; h->lo64[l].bcol = ob; }
    }
  if (h->d_w64) { hipFree(h->d_w64); h->d_w64 = nullptr; }
  HIPCHK(h, hipMalloc(&h->d_w64, host64.size() * sizeof(double)));
  HIPCHK(h, hipMemcpy(h->d_w64, host64.data(), host64.size() * sizeof(double), hipMemcpyHostToDevice));
  if (h->d_w) { hipFree(h->d_w); h->d_w = nullptr; }
  if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }      // captured launches point at the old weights
  HIPCHK(h, hipMalloc(&h->d_w, total * sizeof(float)));
  HIPCHK(h, hipMemcpy(h->d_w, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
  if (!h->d_wimg) HIPCHK(h, hipMalloc(&h->d_wimg, 4 * (size_t)IMG64 * sizeof(float)));
  hipLaunchKernelGGL(k_build_scorer_images, dim3(1), dim3(512), 0, nullptr, scorer_ptrs(h), h->d_wimg);
  HIPCHK(h, hipDeviceSynchronize());
  h->have_w = true;
  return NNJ_OK;
}

// Host-only self check of the launch geometry (no device, no handle): every scorer launch of a rollout of B alignments of
// T rows and C tokens -- the batch itself and the one-alignment form of sampled replicas -- fits the workspace regions
// loop_ws(B, T, C) reserves.  0 = fits; otherwise the number of the first region that is too small.
int nnj_workspace_selfcheck(int32_t B, int32_t T, int32_t C) {
  if (B <= 0 || T < 2 || C <= 0 || T > 256) return NNJ_ERR_ARG;
  const LoopWs w = loop_ws(B, T, C);
  const size_t cap_ap = w.alpha - w.alpha_part, cap_al = w.score_part - w.alpha, cap_sp = w.full - w.score_part;
  for (int Bg : {B, 1})
    for (int n = T; n >= 2; --n) {
      if (Bg == 1 && n != T) break;                          // (only the first table is launched for one alignment)
      if (n > 64) {
        for (int full = 0; full < 2; ++full) {
          if ((full != 0) != (n == T)) continue;
          const WideGeom g = wide_geom(n, Bg, C, full != 0);
          if ((size_t)Bg * g.MB * g.nsc * g.RP * g.RP > cap_ap) return 1;
          if ((size_t)Bg * g.MB * g.RP * g.RP > cap_al) return 2;
          if ((size_t)Bg * g.MB * g.nsc * g.RP > cap_sp) return 3;
        }
      } else {
        const PairGeom g = pair_geom(n == T ? PAIRS_FULL : PAIRS_INCR, n, Bg, C);
        if ((size_t)Bg * g.nsc_a * g.ppad * 64 > cap_ap) return 1;
        if ((size_t)Bg * g.ppad * 64 > cap_al) return 2;
        if ((size_t)Bg * g.nsc * g.ppad > cap_sp) return 3;
      }
    }
  return 0;
}

#ifdef NNJ_STAMP
// diagnostic builds only (tools/stamp_run.py): read and clear the in-kernel stamp accumulators of k_inc_score_w
int nnj_debug_read_stamps(unsigned long long* out8) {
  if (hipDeviceSynchronize() != hipSuccess) return NNJ_ERR_HIP;
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamp), 8 * sizeof(unsigned long long)) != hipSuccess) return NNJ_ERR_HIP;
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z, sizeof(z)) != hipSuccess) return NNJ_ERR_HIP;
  return NNJ_OK;
}
#endif

int nnj_workspace_bytes(const nnj_handle* h, int32_t B, int32_t T, int32_t L_sites, size_t* bytes) {
  if (!h || !bytes || B <= 0 || T < 1 || L_sites <= 0) return fail(const_cast<nnj_handle*>(h), NNJ_ERR_ARG, "nnj_workspace_bytes: bad argument");
  int32_t L = 0;
  if (int rc = to_tokens(const_cast<nnj_handle*>(h), L_sites, &L)) return rc;
  *bytes = ws_floats_model(h, B, T, L) * sizeof(float);
  return NNJ_OK;
}

int nnj_encode(nnj_handle* h, const uint8_t* codes, const float* onehot, const uint8_t* mask, float* state_out,
               int32_t B, int32_t T, int32_t L_sites, void* ws, size_t ws_bytes, void* stream) {
  DevGuard dev_guard;
  if (int rc = ready(h)) return rc;
  int32_t L = 0;                                           // tokens (patches of sites)
  if (int rc = to_tokens(h, L_sites, &L)) return rc;
  if ((!codes && !onehot) || (codes && onehot) || !state_out)
    return fail(h, NNJ_ERR_ARG, "nnj_encode: exactly one of codes / onehot, and an output buffer, are required");
  if (int rc = check_shape(h, B, T, L)) return rc;
  if (int rc = need_ws(h, ws, ws_bytes, B, T, L)) return rc;
  if (int rc = token_mask(h, &mask, ws, B, T, L, static_cast<hipStream_t>(stream))) return rc;
  float* base = static_cast<float*>(ws);
  const size_t state = align_up((size_t)B * T * L * 64, 64);
  float* x = dense_out(h, state_out, ws, B, T, L);
  if (int rc = run_encoder(h, codes, mask, x, base + state, B, T, L, static_cast<hipStream_t>(stream), onehot)) return rc;
  dense_out_done(h, x, state_out, (long)B * T * L, static_cast<hipStream_t>(stream));
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

int nnj_pair_scores_full(nnj_handle* h, const float* state, const uint8_t* mask, float* logits_out, int32_t B,
                         int32_t n, int32_t L_sites, void* ws, size_t ws_bytes, void* stream) {
  DevGuard dev_guard;
  if (int rc = ready(h)) return rc;
  int32_t L = 0;                                           // tokens (patches of sites)
  if (int rc = to_tokens(h, L_sites, &L)) return rc;
  if (!state || !logits_out || n < 2) return fail(h, NNJ_ERR_ARG, "nnj_pair_scores_full: bad argument");
  if (int rc = check_shape(h, B, n, L)) return rc;
  if (int rc = need_ws(h, ws, ws_bytes, B, n, L)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (int rc = token_mask(h, &mask, ws, B, n, L, st)) return rc;
  // first decode of a loop: the rows go into a session (see sess_matches) that the following nnj_env_step /
  // nnj_pair_scores_incr / nnj_aggregate calls on the same tensors continue
  const SessView v = sess_view(ws, B, n, L);
  if (int rc = sess_begin(h, v, state, B, n, L, st)) return rc;
  PairGeom g;
  if (int rc = scorer_mask(h, mask, v.base, v.w, B, L, st, &mask)) return rc;
  if (int rc = launch_pair_scores(h, v.rs, v.ijs, mask, v.base, v.w, PAIRS_FULL, n, B, L, g, st)) return rc;
  {
    Scope sc(h, st, PK_ASSEMBLE);
    hipLaunchKernelGGL(k_assemble_argmax, dim3((unsigned)B), dim3(256), 0, st, g.score_src, g.nsc, g.ppad,
                       (const float*)nullptr, (const int*)nullptr, logits_out, (float*)nullptr, 0L, (const int*)nullptr,
                       0L, (int*)nullptr, 0L, (float*)nullptr, 0L, v.ijs, (int)PAIRS_FULL, n, (const float*)nullptr, 0L, 1.0f,
                       h->d_flag, (const int*)nullptr, (int*)nullptr, 0, 0, StepOut{});
  }
  HIPCHK(h, hipGetLastError());
  sess_set(h, ws, state, B, n, L, n);
  return NNJ_OK;
}

int nnj_pair_scores_incr(nnj_handle* h, const float* state, const uint8_t* mask, const int32_t* ij_prev,
                         const float* logits_prev, float* logits_out, int32_t B, int32_t n, int32_t L_sites, void* ws,
                         size_t ws_bytes, void* stream) {
  DevGuard dev_guard;
  if (int rc = ready(h)) return rc;
  int32_t L = 0;                                           // tokens (patches of sites)
  if (int rc = to_tokens(h, L_sites, &L)) return rc;
  if (!state || !ij_prev || !logits_prev || !logits_out || n < 2) return fail(h, NNJ_ERR_ARG, "nnj_pair_scores_incr: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool cont = sess_matches(h, ws, state, B, n, L);
  const int T0 = cont ? h->sess.T0 : n + 1;
  if (int rc = check_shape(h, B, T0, L)) return rc;
  if (int rc = need_ws(h, ws, ws_bytes, B, T0, L)) return rc;      // (invalidates the session: re-validated below)
  if (int rc = token_mask(h, &mask, ws, B, T0, L, st)) return rc;
  PairGeom g;
  const SessView v = sess_view(ws, B, T0, L);
  RowSet rs = v.rs;
  if (!cont) rs = dense_rowset(h, dense_in(h, state, (long)B * n * L, ws, B, T0, L, st), v.base, v.w, B, n, L, st);
  if (int rc = scorer_mask(h, mask, v.base, v.w, B, L, st, &mask)) return rc;
  if (int rc = launch_pair_scores(h, rs, ij_prev, mask, v.base, v.w, PAIRS_INCR, n, B, L, g, st)) return rc;
  {
    Scope sc(h, st, PK_ASSEMBLE);
    hipLaunchKernelGGL(k_assemble_argmax, dim3((unsigned)B), dim3(256), 0, st, g.score_src, g.nsc, g.ppad,
                       logits_prev, ij_prev, logits_out, (float*)nullptr, 0L, (const int*)nullptr, 0L, (int*)nullptr, 0L,
                       (float*)nullptr, 0L, v.ijs, (int)PAIRS_INCR, n, (const float*)nullptr, 0L, 1.0f, h->d_flag,
                       (const int*)nullptr, (int*)nullptr, 0, 0, StepOut{});
  }
  HIPCHK(h, hipGetLastError());
  if (cont) sess_set(h, ws, state, B, T0, L, n);
  return NNJ_OK;
}

int nnj_score_index_map(nnj_handle* h, const int32_t* ij_prev, int64_t* idx_out, int32_t B, int32_t n, void* stream) {
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (!ij_prev || !idx_out || B <= 0 || n < 2) return fail(h, NNJ_ERR_ARG, "nnj_score_index_map: bad argument");
  const long total = (long)B * (n * (n - 1) / 2);
  hipLaunchKernelGGL(k_index_map, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     ij_prev, reinterpret_cast<long long*>(idx_out), B, n);
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

int nnj_aggregate(nnj_handle* h, const float* state, const int32_t* ij, float* out_row, int32_t B, int32_t n,
                  int32_t L_sites, void* ws, size_t ws_bytes, void* stream) {
  DevGuard dev_guard;
  if (int rc = ready(h)) return rc;
  int32_t L = 0;                                           // tokens (patches of sites)
  if (int rc = to_tokens(h, L_sites, &L)) return rc;
  if (!state || !ij || !out_row || n < 2) return fail(h, NNJ_ERR_ARG, "nnj_aggregate: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool cont = sess_matches(h, ws, state, B, n, L);
  const int T0 = cont ? h->sess.T0 : n;
  if (int rc = check_shape(h, B, T0, L)) return rc;
  if (int rc = need_ws(h, ws, ws_bytes, B, T0, L)) return rc;
  const SessView v = sess_view(ws, B, T0, L);
  RowSet rs = v.rs;
  if (!cont) rs = dense_rowset(h, dense_in(h, state, (long)B * n * L, ws, B, T0, L, st), v.base, v.w, B, n, L, st);
  float* orow = dense_out(h, out_row, ws, B, T0, L);
  if (int rc = launch_aggregate(h, rs, ij, v.base, v.w, orow, nullptr, nullptr, nullptr, (long)L * 64, 1, 0, n, B, L, st)) return rc;
  dense_out_done(h, orow, out_row, (long)B * L, st);
  HIPCHK(h, hipGetLastError());
  if (cont) sess_set(h, ws, state, B, T0, L, n);           // the rows are untouched: the session goes on
  return NNJ_OK;
}

int nnj_env_step(nnj_handle* h, const float* state, const int32_t* ij, float* state_out, int32_t B, int32_t n,
                 int32_t L_sites, void* ws, size_t ws_bytes, void* stream) {
  DevGuard dev_guard;
  if (int rc = ready(h)) return rc;
  int32_t L = 0;                                           // tokens (patches of sites)
  if (int rc = to_tokens(h, L_sites, &L)) return rc;
  if (!state || !ij || !state_out || n < 3) return fail(h, NNJ_ERR_ARG, "nnj_env_step: bad argument (n must be >= 3)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool cont = sess_matches(h, ws, state, B, n, L);
  const int T0 = cont ? h->sess.T0 : n;
  if (int rc = check_shape(h, B, T0, L)) return rc;
  if (int rc = need_ws(h, ws, ws_bytes, B, T0, L)) return rc;      // (invalidates the session: re-validated below)
  const SessView v = sess_view(ws, B, T0, L);
  if (cont) {
    // in the session: merged row in place over slot(i) with its transforms, position j leaves the list, and the dense
    // tensor the reference's env.step returns (environment.py:833-835) is one gather of the live rows
    HIPCHK(h, hipMemcpyAsync(v.ijs, ij, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    if (int rc = launch_aggregate(h, v.rs, v.ijs, v.base, v.w, v.S, v.base + v.w.U, v.base + v.w.Kp, v.base + v.w.beta,
                                  v.rs.bstride, T0, 1, n, B, L, st)) return rc;
    Scope sc(h, st, PK_MISC);
    hipLaunchKernelGGL(k_update_live, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, v.live, T0, (const int*)v.ijs, B, n);
    float* so = dense_out(h, state_out, ws, B, T0, L);
    hipLaunchKernelGGL(k_gather_rows, dim3(16, (unsigned)(n - 1), (unsigned)B), dim3(256), 0, st, (const float*)v.S,
                       (const int*)v.live, T0, so, n - 1, (long)L * 16, (long)T0 * L * 16);
    dense_out_done(h, so, state_out, (long)B * (n - 1) * L, st);
    HIPCHK(h, hipGetLastError());
    sess_set(h, ws, state_out, B, T0, L, n - 1);
    return NNJ_OK;
  }
  const float* state_w = dense_in(h, state, (long)B * n * L, ws, B, T0, L, st);
  RowSet rs = dense_rowset(h, state_w, v.base, v.w, B, n, L, st);
  float* merged = v.base + v.w.merged;
  if (int rc = launch_aggregate(h, rs, ij, v.base, v.w, merged, nullptr, nullptr, nullptr, (long)L * 64, 1, 0, n, B, L, st)) return rc;
  {
    Scope sc(h, st, PK_MISC);
    const long row_f4 = (long)L * 16;
    float* so = dense_out(h, state_out, ws, B, T0, L);
    hipLaunchKernelGGL(k_compact_rows, dim3(16, (unsigned)(n - 1), (unsigned)B), dim3(256), 0, st, state_w, merged, ij,
                       so, n, row_f4);
    dense_out_done(h, so, state_out, (long)B * (n - 1) * L, st);
  }
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

int nnj_session_reset(nnj_handle* h) {
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  h->sess.valid = false;
  return NNJ_OK;
}

int nnj_select_pair(nnj_handle* h, const float* logits, int32_t* ij_out, float* top2_gap, int32_t B, int32_t n,
                    void* stream) {
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (!logits || !ij_out || B <= 0 || n < 2) return fail(h, NNJ_ERR_ARG, "nnj_select_pair: bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  Scope sc(h, st, PK_ASSEMBLE);
  hipLaunchKernelGGL(k_select_pair, dim3((unsigned)B), dim3(256), 0, st, logits, ij_out, top2_gap, n);
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

int nnj_step(nnj_handle* h, const float* state, const uint8_t* mask, const int32_t* ij, const float* logits_prev,
             const int32_t* forced_next, float* state_out, float* logits_out, int32_t* chosen_ij, float* top2_gap,
             int32_t B, int32_t n, int32_t L_sites, void* ws, size_t ws_bytes, void* stream) {
  DevGuard dev_guard;
  if (int rc = ready(h)) return rc;
  int32_t L = 0;                                           // tokens (patches of sites)
  if (int rc = to_tokens(h, L_sites, &L)) return rc;
  if (!state || !ij || !logits_prev || !state_out || !logits_out || !chosen_ij || n < 2)
    return fail(h, NNJ_ERR_ARG, "nnj_step: bad argument (n = rows after the merge, >= 2)");
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int C = L;
  // A loop of nnj_step calls is a SESSION: the rows live in slot layout in the caller's workspace next to their
  // cached transforms (the rollout's own state), so a step runs the rollout's kernels once -- merged-row aggregate in
  // place, live-list update, the n new scores, table + argmax -- and re-transforms nothing.  The session continues
  // when `state` is the tensor the previous call returned (same workspace, batch, sites and row count); any other
  // input starts a new one from the dense tensor (copy into slots + row transforms).  state_out is the dense
  // gather of the live rows the reference's env.step hands back (environment.py:833-835).
  // Up to 64 rows the step runs on the two-pass kernels of the rollout (nnj_step2.hpp): the merged row is produced
  // inside the alpha pass of the new pairs and its attention weights come from the previous call's table kernel when the
  // caller merges the pair that call picked (`chosen_ij`, the usual loop) -- any other pair takes the per-alignment
  // fallback.  These kernels do not maintain K' of the merged rows: the session then continues through nnj_step only
  // (the other dense-state entry points start again from the tensor they are given).
  const bool two_pass = h->two_pass != 0 && n + 1 <= 64 && n + 1 > 2;
  const bool cont = sess_matches(h, ws, state, B, n + 1, L, /*need_kp=*/!two_pass);
  const int T0 = cont ? h->sess.T0 : n + 1;
  const nnj_handle::StepSession prev = h->sess;            // (need_ws invalidates the session: re-validated below)
  if (int rc = check_shape(h, B, T0, L)) return rc;
  if (int rc = need_ws(h, ws, ws_bytes, B, T0, L)) return rc;
  if (int rc = token_mask(h, &mask, ws, B, T0, L, st)) return rc;
  if (two_pass && T0 <= 64) {
    int live_idx = cont ? prev.live_idx : 0, cand_idx = cont ? prev.cand_idx : 0;
    const bool have_am = cont && prev.tp;
    const SessView v = sess_view(ws, B, T0, C, live_idx);
    float* base = v.base;
    const LoopWs w = v.w;
    if (!cont) {
      if (int rc = sess_begin(h, v, state, B, T0, C, st)) return rc;
    }
    int* live_old = v.live;
    int* live_new = reinterpret_cast<int*>(base + w.live) + (size_t)(live_idx ^ 1) * B * T0;
    int* candbuf[2] = {reinterpret_cast<int*>(base + w.cand), reinterpret_cast<int*>(base + w.cand) + 2 * (size_t)B};
    int* need = reinterpret_cast<int*>(base + w.need);
    int* pick = reinterpret_cast<int*>(base + w.pick);
    const bool use_cand = h->two_pass_cand != 0;
    HIPCHK(h, hipMemcpyAsync(v.ijs, ij, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    {
      Scope sc(h, st, PK_STEP_SMALL);
      hipLaunchKernelGGL(k_step_prepare, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, v.ijs, (const int*)pick,
                         have_am ? 0 : 1, need, candbuf[cand_idx], reinterpret_cast<int*>(base + w.cand_run),
                         (const int*)live_old, live_new, T0, B, n + 1);
    }
    RowSet rs = v.rs;
    rs.live = live_new;
    if (int rc = scorer_mask(h, mask, base, w, B, C, st, &mask)) return rc;
    PairGeom g;
    Step2 o;
    o.first = !have_am;                                    // the per-row biases of a new session: summed from the partials
    o.fallback = true;                                     // (per alignment, only where `need` is set)
    o.cand = use_cand;
    o.cand_cur = candbuf[cand_idx];
    if (int rc = launch_step2(h, rs, live_old, v.ijs, mask, base, w, o, n, B, C, g, st)) return rc;
    StepOut so{};
    if (n - 1 >= 2) {                                      // the table kernel prepares the next call's merge
      so.lam = n > 2 ? base + w.lam : nullptr;
      so.beta_slot = base + w.beta_slot;
      so.nslot = T0;
      so.acand_part = (use_cand && n > 2) ? base + w.acand : nullptr;
      so.nblk = g.blocks;
      so.cand_cur = (use_cand && n > 2) ? candbuf[cand_idx] : nullptr;
      so.am = base + w.am;
      so.need = need;
      so.cand_next = use_cand ? candbuf[cand_idx ^ 1] : nullptr;
      so.cand_run = reinterpret_cast<int*>(base + w.cand_run);
      so.inv_scale = 1.0f / sqrtf(64.0f * (float)C);
      so.fallback = 1;
    }
    {
      Scope sc(h, st, PK_ASSEMBLE);
      hipLaunchKernelGGL(k_assemble_argmax, dim3((unsigned)B), dim3(256), 0, st, g.score_src, g.nsc, g.ppad, logits_prev,
                         (const int*)v.ijs, logits_out, (float*)nullptr, 0L, forced_next, 2L, (int*)nullptr, 0L, top2_gap, 1L,
                         chosen_ij, (int)PAIRS_INCR, n, (const float*)nullptr, 0L, 1.0f, h->d_flag, (const int*)live_new,
                         (int*)nullptr, T0, 1, so);
    }
    HIPCHK(h, hipMemcpyAsync(pick, chosen_ij, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    {
      Scope sc(h, st, PK_MISC);
      float* so_ = dense_out(h, state_out, ws, B, T0, C);
      hipLaunchKernelGGL(k_gather_rows, dim3(16, (unsigned)n, (unsigned)B), dim3(256), 0, st, (const float*)v.S,
                         (const int*)live_new, T0, so_, n, (long)C * 16, (long)T0 * C * 16);
      dense_out_done(h, so_, state_out, (long)B * n * C, st);
    }
    HIPCHK(h, hipGetLastError());
    sess_set(h, ws, state_out, B, T0, L, n);
    h->sess.live_idx = live_idx ^ 1; h->sess.cand_idx = cand_idx ^ 1;
    h->sess.tp = n - 1 >= 2; h->sess.kp_valid = false;
    return NNJ_OK;
  }
  const SessView v = sess_view(ws, B, T0, C);
  float* S = v.S;
  float* base = v.base;
  const LoopWs w = v.w;
  int* live = v.live;
  int* ijs = v.ijs;
  if (!cont) {
    if (int rc = sess_begin(h, v, state, B, T0, C, st)) return rc;
  }
  RowSet rs;
  rs.S = S; rs.U = base + w.U; rs.Kp = base + w.Kp; rs.beta_part = base + w.beta;
  rs.bstride = (long)T0 * C * 64; rs.live = live; rs.live_stride = T0; rs.ntile32 = beta_stride(B, C);
  HIPCHK(h, hipMemcpyAsync(ijs, ij, (size_t)B * 2 * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
  // env.step (environment.py:760-835): merged row into slot(i), position j leaves the list
  if (int rc = launch_aggregate(h, rs, ijs, base, w, S, base + w.U, base + w.Kp, base + w.beta, rs.bstride, T0, 1, n + 1,
                                B, C, st)) return rc;
  {
    Scope sc(h, st, PK_MISC);
    hipLaunchKernelGGL(k_update_live, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, live, T0, (const int*)ijs, B, n + 1);
  }
  // decode_zxr (model.py:184-201 + utils.py:213-251) and the argmax (finetune_rl_search.py:145,159-160)
  PairGeom g;
  if (int rc = scorer_mask(h, mask, base, w, B, C, st, &mask)) return rc;
  if (int rc = launch_pair_scores(h, rs, ijs, mask, base, w, PAIRS_INCR, n, B, C, g, st)) return rc;
  {
    Scope sc(h, st, PK_ASSEMBLE);
    hipLaunchKernelGGL(k_assemble_argmax, dim3((unsigned)B), dim3(256), 0, st, g.score_src, g.nsc, g.ppad, logits_prev,
                       (const int*)ijs, logits_out, (float*)nullptr, 0L, forced_next, 2L, (int*)nullptr, 0L, top2_gap, 1L,
                       chosen_ij, (int)PAIRS_INCR, n, (const float*)nullptr, 0L, 1.0f, h->d_flag, (const int*)nullptr,
                       (int*)nullptr, 0, 0, StepOut{});
  }
  {
    Scope sc(h, st, PK_MISC);
    float* so = dense_out(h, state_out, ws, B, T0, C);
    hipLaunchKernelGGL(k_gather_rows, dim3(16, (unsigned)n, (unsigned)B), dim3(256), 0, st, (const float*)S, (const int*)live,
                       T0, so, n, (long)C * 16, (long)T0 * C * 16);
    dense_out_done(h, so, state_out, (long)B * n * C, st);
  }
  HIPCHK(h, hipGetLastError());
  sess_set(h, ws, state_out, B, T0, L, n);
  return NNJ_OK;
}

static int rollout_core(nnj_handle* h, const uint8_t* codes, const uint8_t* mask_in, int32_t B, int32_t T, int32_t L,
                        int32_t n_encode, const int32_t* forced, const float* uniforms, float inv_temp,
                        int32_t* merges_out, float* trace, float* gap, float* state_out, void* ws, hipStream_t st) {
  const int C = L;
  float* S = static_cast<float*>(ws);
  const size_t state = align_up((size_t)B * T * C * 64, 64);
  float* base = S + state;
  const LoopWs w = loop_ws(B, T, C);
  const uint8_t* mask = mask_in;
  if (int rc = run_encoder(h, codes, mask_in, S, base, n_encode, T, C, st)) return rc;   // finetune_rl_search.py:108-112
  if (n_encode == 1 && B > 1) {
    Scope sc(h, st, PK_MISC);
    hipLaunchKernelGGL(k_replicate, dim3(64, (unsigned)(B - 1)), dim3(256), 0, st, S, (long)T * C * 16);
    if (mask_in) {
      uint8_t* mrep = reinterpret_cast<uint8_t*>(base + w.mrep);
      hipLaunchKernelGGL(k_replicate_u8, dim3((unsigned)(((size_t)B * L + 255) / 256)), dim3(256), 0, st, mask_in, mrep, B, L);
      mask = mrep;
    }
  }
  if (state_out && narrow_model(h)) dense_out_done(h, S, state_out, (long)B * T * C, st);
  else if (state_out) HIPCHK(h, hipMemcpyAsync(state_out, S, (size_t)B * T * C * 64 * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (int rc = scorer_mask(h, mask, base, w, B, C, st, &mask)) return rc;      // (after the encoder: the region is its scratch)
  // two live lists: k_assemble_argmax writes the next step's list (without position j) beside the one the merged-row
  // kernels of this step still read
  int* livebuf[2] = {reinterpret_cast<int*>(base + w.live), reinterpret_cast<int*>(base + w.live) + (size_t)B * T};
  int* ij = reinterpret_cast<int*>(base + w.ij);
  {
    Scope sc(h, st, PK_MISC);
    hipLaunchKernelGGL(k_init_live, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, st, livebuf[0], T, B, T);
  }
  launch_row_xf(h, S, base + w.U, base + w.Kp, base + w.beta, (long)T * C * 64, T, T, B, C, st);
  RowSet rs;
  rs.S = S; rs.U = base + w.U; rs.Kp = base + w.Kp; rs.beta_part = base + w.beta;
  rs.bstride = (long)T * C * 64; rs.live = livebuf[0]; rs.live_stride = T; rs.ntile32 = beta_stride(B, C);
  if (h->two_pass && T <= 64 && T > 2) {
    // the table kernel of step 0 already computes the weights of the first merge: it needs the per-row biases
    Scope sc(h, st, PK_STEP_SMALL);
    hipLaunchKernelGGL(k_beta_sum, dim3((unsigned)((B * T + 255) / 256)), dim3(256), 0, st, rs.beta_part, rs.ntile32,
                       rs.ntile32, (float)C * scorer_ptrs(h).t0, base + w.beta_slot, B * T);
  }
  size_t total = 0;
  for (int n = T; n >= 2; --n) total += (size_t)n * (n - 1) / 2;
  float* lg[2] = {base + w.logits0, base + w.logits1};
  size_t off = 0;
  // The two-pass step (nnj_step2.hpp) runs whenever at most 64 rows are left after a merge: the merge picked from a
  // table of n + 1 rows happens INSIDE the next step's alpha pass, not behind the table kernel.
  const bool two_pass = h->two_pass != 0;
  const bool use_cand = two_pass && h->two_pass_cand != 0;
  int* candbuf[2] = {reinterpret_cast<int*>(base + w.cand), reinterpret_cast<int*>(base + w.cand) + 2 * (size_t)B};
  bool prev_gave_am = false;                               // did the previous table kernel write am / need?
  bool prev_lam = false, prev_v2 = false, prev_alpha0 = false;
  for (int step = 0, n = T; n >= 2; ++step, --n) {
    const int mode = step == 0 ? PAIRS_FULL : PAIRS_INCR;
    PairGeom g;
    rs.live = livebuf[step & 1];
    const bool v2 = two_pass && step >= 1 && n <= 64 && prev_gave_am;
    const bool rep0 = step == 0 && n_encode == 1 && B > 1;
    if (v2) {
      Step2 o;
      // without forced / sampled picks the pick of a step >= 1 of this kind is always covered (a new pair or the
      // candidate); the first such step follows a table whose kernels carry no logits
      o.first = !prev_v2 && T > 64;                        // (up to 64 rows the biases were summed in front of step 0)
      o.fallback = !(prev_lam || prev_alpha0) || forced != nullptr || uniforms != nullptr || !use_cand;
      o.cand = use_cand;
      o.cand_cur = candbuf[step & 1];
      if (int rc = launch_step2(h, rs, livebuf[(step + 1) & 1], ij, mask, base, w, o, n, B, C, g, st)) return rc;
    } else {
      // sampled rollouts of ONE alignment: every replica's first table is the same numbers -- the all-pairs kernels
      // run once (alignment 0), k_assemble_argmax gives every replica its own pick from them
      if (int rc = launch_pair_scores(h, rs, ij, mask, base, w, mode, n, rep0 ? 1 : B, C, g, st)) return rc;   // :121-126
    }
    // does the NEXT step run as a two-pass step?  then this table kernel supplies the weights of its merge
    const bool next_v2 = two_pass && n - 1 >= 2 && n - 1 <= 64;
    StepOut so{};
    so.rep0 = rep0 ? 1 : 0;
    if (next_v2) {
      so.lam = (v2 && n > 2) ? base + w.lam : nullptr;
      so.beta_slot = base + w.beta_slot;
      so.nslot = T;
      so.acand_part = (v2 && use_cand && n > 2) ? base + w.acand : nullptr;
      so.nblk = g.blocks;
      so.cand_cur = (v2 && use_cand && n > 2) ? candbuf[step & 1] : nullptr;
      so.am = base + w.am;
      so.need = reinterpret_cast<int*>(base + w.need);
      so.cand_next = use_cand ? candbuf[(step + 1) & 1] : nullptr;
      so.cand_run = reinterpret_cast<int*>(base + w.cand_run);
      so.inv_scale = 1.0f / sqrtf(64.0f * (float)C);
      // (the fallback kernels of the NEXT step: same condition as Step2::fallback there)
      so.fallback = (!(((v2 && n > 2)) || (mode == PAIRS_FULL && n <= 64 && n > 2)) || forced != nullptr || uniforms != nullptr ||
                     !use_cand) ? 1 : 0;
      if (mode == PAIRS_FULL && n <= 64 && n > 2) {          // step 0 on the 64-row kernels: the pick's logits are in alpha_part
        so.alpha0 = base + w.alpha_part;
        so.nsc0 = g.nsc_a;
        so.ppad0 = g.ppad;
      }
    }
    {
      Scope sc(h, st, PK_ASSEMBLE);                                                                 // :140-160
      hipLaunchKernelGGL(k_assemble_argmax, dim3((unsigned)B), dim3(256), 0, st, g.score_src, g.nsc, g.ppad,
                         (const float*)lg[(step + 1) & 1], (const int*)ij, lg[step & 1], trace ? trace + off : nullptr,
                         (long)total, forced ? forced + 2 * step : nullptr, (long)(T - 1) * 2, merges_out + 2 * step,
                         (long)(T - 1) * 2, gap ? gap + step : nullptr, (long)(T - 1), ij, mode, n,
                         uniforms ? uniforms + step : nullptr, (long)(T - 1), inv_temp, h->d_flag,
                         (const int*)livebuf[step & 1], n > 2 ? livebuf[(step + 1) & 1] : (int*)nullptr, T, v2 ? 1 : 0, so);
    }
    off += (size_t)n * (n - 1) / 2;
    prev_gave_am = next_v2;
    prev_lam = so.lam != nullptr;
    prev_alpha0 = so.alpha0 != nullptr;
    prev_v2 = v2;
    if (n > 2 && !next_v2) {                                                                        // env.step :164
      if (int rc = launch_aggregate(h, rs, ij, base, w, S, base + w.U, base + w.Kp, base + w.beta, rs.bstride, T, 1, n,
                                    B, C, st)) return rc;
    }
  }
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

// The alignments of a batch are independent: a rollout of B >= 64 alignments runs as `concurrency`
// sub-batches on streams of the library's own, forked from and joined to the caller's stream by events (the call
// stays asynchronous on `stream`, nothing else changes for the caller).  The sub-batches drift apart by a kernel or
// two, so HBM-bound launches of one (merged-row passes, q/k/v planes) overlap matrix- and vector-bound launches of
// another, and the tail of every launch is filled by the other sub-batch.
static int rollout_impl(nnj_handle* h, const uint8_t* codes, const uint8_t* mask_in, int32_t B, int32_t T, int32_t L_sites,
                        int32_t n_encode, const int32_t* forced, const float* uniforms, float inv_temp,
                        int32_t* merges_out, float* trace, float* gap, float* state_out, void* ws, size_t ws_bytes,
                        void* stream) {
  DevGuard dev_guard;
  if (int rc = ready(h)) return rc;
  if (!codes || !merges_out || T < 2) return fail(h, NNJ_ERR_ARG, "rollout: bad argument");
  if (n_encode != B && n_encode != 1) return fail(h, NNJ_ERR_ARG, "rollout: n_encode must be 1 or B");
  int32_t L = 0;                                           // tokens (patches of sites); codes stay [*, T, L_sites]
  if (int rc = to_tokens(h, L_sites, &L)) return rc;
  if (int rc = check_shape(h, B, T, L)) return rc;
  if (int rc = need_ws(h, ws, ws_bytes, B, T, L)) return rc;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (h->cfg.patch_size > 1 && mask_in) {                  // mask of the tokens, n_encode rows, in the workspace tail
    uint8_t* pm = reinterpret_cast<uint8_t*>(static_cast<float*>(ws) + ws_floats(B, T, L) - pm_floats(B, L));
    hipLaunchKernelGGL(k_patch_mask, dim3((unsigned)(((size_t)n_encode * L + 255) / 256)), dim3(256), 0, st, mask_in, pm,
                       n_encode, L, h->cfg.patch_size);
    mask_in = pm;
  }
  int ns = std::min(h->concurrency, NNJ_MAX_SUB);
  if (n_encode != B || B < 64 || ns < 2) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool graph_ok = h->use_graph && B <= 16 && !h->prof &&
                          (st == nullptr || (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone));
    if (!graph_ok)
      return rollout_core(h, codes, mask_in, B, T, L, n_encode, forced, uniforms, inv_temp, merges_out, trace, gap,
                          state_out, ws, st);
    // the legacy default stream cannot be captured: the graph then lives on a stream of the handle, forked from and
    // joined to the caller's stream by events
    hipStream_t gs = st;
    if (st == nullptr) {
      if (!h->sub[0]) HIPCHK(h, hipStreamCreateWithFlags(&h->sub[0], hipStreamNonBlocking));
      if (!h->ev_fork) HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
      if (!h->ev_join[0]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[0], hipEventDisableTiming));
      gs = h->sub[0];
      HIPCHK(h, hipEventRecord(h->ev_fork, st));
      HIPCHK(h, hipStreamWaitEvent(gs, h->ev_fork, 0));
    }
    const nnj_handle::GraphKey key{codes, mask_in, forced, uniforms, merges_out, trace, gap, state_out, ws,
                                   B, T, L, n_encode, h->debug_stop, inv_temp};
    static const bool gtrace = getenv("NNJ_GRAPH_TRACE") != nullptr;     // diagnostics: which branch a small-batch call takes
    if (gtrace) fprintf(stderr, "[nnj graph] B=%d: %s\n", B, (h->gexec && key == h->gkey) ? "replay" : (key == h->gcand ? "capture" : "plain"));
    if (!(h->gexec && key == h->gkey) && !(key == h->gcand)) {
      // first call with these arguments: plain launches; a graph is built only when a call repeats (a caller that
      // passes fresh buffers every time would otherwise pay a capture per call)
      h->gcand = key;
      if (int rc = rollout_core(h, codes, mask_in, B, T, L, n_encode, forced, uniforms, inv_temp, merges_out, trace, gap,
                                state_out, ws, gs)) return rc;
    } else if (!h->gexec || !(key == h->gkey)) {
      if (h->gexec) { hipGraphExecDestroy(h->gexec); h->gexec = nullptr; }
      // warm run outside the capture: kernel attributes (dynamic LDS sizes) are set by the launch helpers
      if (int rc = rollout_core(h, codes, mask_in, B, T, L, n_encode, forced, uniforms, inv_temp, merges_out, trace, gap,
                                state_out, ws, gs)) return rc;
      HIPCHK(h, hipStreamBeginCapture(gs, hipStreamCaptureModeThreadLocal));
      const int rc = rollout_core(h, codes, mask_in, B, T, L, n_encode, forced, uniforms, inv_temp, merges_out, trace,
                                  gap, state_out, ws, gs);
      hipGraph_t graph = nullptr;
      const hipError_t e = hipStreamEndCapture(gs, &graph);
      if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
      if (e != hipSuccess || !graph) return fail(h, NNJ_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
      const hipError_t e2 = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
      hipGraphDestroy(graph);
      if (e2 != hipSuccess) { h->gexec = nullptr; return fail(h, NNJ_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e2)); }
      h->gkey = key;                       // the warm run above already produced this call's results
    } else {
      HIPCHK(h, hipGraphLaunch(h->gexec, gs));
    }
    if (gs != st) {
      HIPCHK(h, hipEventRecord(h->ev_join[0], gs));
      HIPCHK(h, hipStreamWaitEvent(st, h->ev_join[0], 0));
    }
    return NNJ_OK;
  }
  if (!h->ev_fork) HIPCHK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
  for (int k = 0; k < ns; ++k) {
    if (!h->sub[k]) HIPCHK(h, hipStreamCreateWithFlags(&h->sub[k], hipStreamNonBlocking));
    if (!h->ev_join[k]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming));
  }
  size_t total = 0;
  for (int n = T; n >= 2; --n) total += (size_t)n * (n - 1) / 2;
  const int per = (B + ns - 1) / ns;
  const size_t sub_floats = align_up(ws_floats_one(per, T, L), 64);
  HIPCHK(h, hipEventRecord(h->ev_fork, st));
  for (int k = 0; k < ns; ++k) {
    const int b0 = k * per, bn = std::min(per, B - b0);
    if (bn <= 0) break;
    HIPCHK(h, hipStreamWaitEvent(h->sub[k], h->ev_fork, 0));
    if (int rc = rollout_core(h, codes + (size_t)b0 * T * L_sites, mask_in ? mask_in + (size_t)b0 * L : nullptr, bn, T, L, bn,
                              forced ? forced + (size_t)b0 * (T - 1) * 2 : nullptr,
                              uniforms ? uniforms + (size_t)b0 * (T - 1) : nullptr, inv_temp,
                              merges_out + (size_t)b0 * (T - 1) * 2, trace ? trace + (size_t)b0 * total : nullptr,
                              gap ? gap + (size_t)b0 * (T - 1) : nullptr,
                              state_out ? state_out + (size_t)b0 * T * L * h->cfg.embed_dim : nullptr,
                              static_cast<float*>(ws) + (size_t)k * sub_floats, h->sub[k])) return rc;
    HIPCHK(h, hipEventRecord(h->ev_join[k], h->sub[k]));
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_join[k], 0));
  }
  return NNJ_OK;
}

int nnj_set_concurrency(nnj_handle* h, int32_t streams) {
  if (!h || streams < 1 || streams > NNJ_MAX_SUB) return fail(h, NNJ_ERR_ARG, "nnj_set_concurrency: 1..%d", NNJ_MAX_SUB);
  h->concurrency = streams;
  return NNJ_OK;
}

int nnj_rollout_argmax(nnj_handle* h, const uint8_t* codes, const uint8_t* mask, int32_t B, int32_t T, int32_t L,
                       const int32_t* forced, int32_t* merges_out, float* trace, float* gap, float* state_out,
                       void* ws, size_t ws_bytes, void* stream) {
  return rollout_impl(h, codes, mask, B, T, L, B, forced, nullptr, 1.0f, merges_out, trace, gap, state_out, ws, ws_bytes,
                      stream);
}

int nnj_rollout_sample(nnj_handle* h, const uint8_t* codes, const uint8_t* mask, int32_t B, int32_t T, int32_t L,
                       int32_t n_encode, const float* uniforms, float temperature, int32_t* merges_out, float* trace,
                       void* ws, size_t ws_bytes, void* stream) {
  if (!uniforms || !(temperature > 0.f)) return fail(h, NNJ_ERR_ARG, "nnj_rollout_sample: uniforms and a positive temperature are required");
  return rollout_impl(h, codes, mask, B, T, L, n_encode, nullptr, uniforms, 1.0f / temperature, merges_out, trace, nullptr,
                      nullptr, ws, ws_bytes, stream);
}

int nnj_topology_hash(nnj_handle* h, const int32_t* merges, int32_t B, int32_t T, uint64_t* keys_out, void* stream) {
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (!merges || !keys_out || B <= 0 || T < 2 || T > 256) return fail(h, NNJ_ERR_ARG, "nnj_topology_hash: bad argument");
  hipLaunchKernelGGL(k_topology_hash, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), merges,
                     reinterpret_cast<unsigned long long*>(keys_out), B, T);
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------- tree likelihood (nnj_likelihood.hpp)
namespace {

// regularised lower incomplete gamma P(a, x): series / continued fraction (Numerical Recipes 6.2 forms)
double gammp(double a, double x) {
  if (x <= 0.0) return 0.0;
  const double gln = lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 1000; ++n) { ap += 1.0; del *= x / ap; sum += del; if (fabs(del) < fabs(sum) * 1e-16) break; }
    return sum * exp(-x + a * log(x) - gln);
  }
  double b = x + 1.0 - a, c = 1.0 / 1e-300, d = 1.0 / b, hh = d;
  for (int i = 1; i < 1000; ++i) {
    const double an = -i * (i - a);
    b += 2.0;
    d = an * d + b; if (fabs(d) < 1e-300) d = 1e-300;
    c = b + an / c; if (fabs(c) < 1e-300) c = 1e-300;
    d = 1.0 / d;
    const double del = d * c;
    hh *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return 1.0 - exp(-x + a * log(x) - gln) * hh;
}
// mean rate of each of ncat equiprobable categories of Gamma(alpha, alpha) (Yang 1994)
void gamma_rates(double alpha, int ncat, double* rates) {
  if (!(alpha > 0.0) || ncat <= 1) { for (int i = 0; i < ncat; ++i) rates[i] = 1.0; return; }
  std::vector<double> cut(ncat + 1, 0.0);
  for (int i = 1; i < ncat; ++i) {                         // quantile by bisection on P(alpha, alpha x) = i / ncat
    const double p = (double)i / ncat;
    double lo = 0.0, hi = 1.0;
    while (gammp(alpha, alpha * hi) < p) hi *= 2.0;
    for (int it = 0; it < 200; ++it) { const double mid = 0.5 * (lo + hi); (gammp(alpha, alpha * mid) < p ? lo : hi) = mid; }
    cut[i] = 0.5 * (lo + hi);
  }
  double prev = 0.0;
  for (int i = 0; i < ncat; ++i) {
    const double cur = i + 1 < ncat ? gammp(alpha + 1.0, alpha * cut[i + 1]) : 1.0;
    rates[i] = (cur - prev) * ncat;
    prev = cur;
  }
}
// eigen system of the GTR rate matrix: Q_ij = r_ij pi_j, normalised to one expected substitution per unit time;
// B = Pi^1/2 Q Pi^-1/2 is symmetric: cyclic Jacobi, then U = Pi^-1/2 V, Uinv = V^T Pi^1/2
int build_model(const nnj_subst_model* m, LikModel& md, char* err) {
  static const int pair[4][4] = {{-1, 0, 1, 2}, {0, -1, 3, 4}, {1, 3, -1, 5}, {2, 4, 5, -1}};   // AC AG AT CG CT GT
  if (m->ncat < 1 || m->ncat > LIK_MAXCAT) { snprintf(err, 512, "substitution model: ncat must be 1..%d", LIK_MAXCAT); return NNJ_ERR_ARG; }
  if (!(m->pinv >= 0.0 && m->pinv < 1.0)) { snprintf(err, 512, "substitution model: pinv must be in [0, 1)"); return NNJ_ERR_ARG; }
  double pi[4], fs = 0.0;
  for (int i = 0; i < 4; ++i) { if (!(m->freqs[i] > 0.0)) { snprintf(err, 512, "substitution model: base frequencies must be positive"); return NNJ_ERR_ARG; } fs += m->freqs[i]; }
  for (int i = 0; i < 4; ++i) pi[i] = m->freqs[i] / fs;
  for (int i = 0; i < 6; ++i) if (!(m->rates[i] > 0.0)) { snprintf(err, 512, "substitution model: exchange rates must be positive"); return NNJ_ERR_ARG; }
  double Q[4][4], mu = 0.0;
  for (int i = 0; i < 4; ++i) {
    double row = 0.0;
    for (int j = 0; j < 4; ++j) if (i != j) { Q[i][j] = m->rates[pair[i][j]] * pi[j]; row += Q[i][j]; }
    Q[i][i] = -row;
    mu += pi[i] * row;
  }
  double A[4][4], V[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) { A[i][j] = sqrt(pi[i]) * Q[i][j] / mu / sqrt(pi[j]); V[i][j] = i == j ? 1.0 : 0.0; }
  for (int i = 0; i < 4; ++i) for (int j = i + 1; j < 4; ++j) { const double sm = 0.5 * (A[i][j] + A[j][i]); A[i][j] = A[j][i] = sm; }
  for (int sweep = 0; sweep < 100; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += A[p][q] * A[p][q];
    if (off < 1e-32) break;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (fabs(A[p][q]) < 1e-300) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < 4; ++k) { const double akp = A[k][p], akq = A[k][q]; A[k][p] = c * akp - sn * akq; A[k][q] = sn * akp + c * akq; }
        for (int k = 0; k < 4; ++k) { const double apk = A[p][k], aqk = A[q][k]; A[p][k] = c * apk - sn * aqk; A[q][k] = sn * apk + c * aqk; }
        for (int k = 0; k < 4; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - sn * vkq; V[k][q] = sn * vkp + c * vkq; }
      }
  }
  for (int k = 0; k < 4; ++k) md.lam[k] = A[k][k];
  for (int i = 0; i < 4; ++i)
    for (int k = 0; k < 4; ++k) { md.U[i * 4 + k] = V[i][k] / sqrt(pi[i]); md.Uinv[k * 4 + i] = V[i][k] * sqrt(pi[i]); }
  for (int i = 0; i < 4; ++i) md.freqs[i] = pi[i];
  md.ncat = m->ncat; md.pinv = m->pinv;
  for (int i = 0; i < LIK_MAXCAT; ++i) md.rates[i] = 1.0;
  gamma_rates(m->alpha, m->ncat, md.rates);
  return NNJ_OK;
}

struct LikWs { size_t prog, colour, brlen, brlen_new, brlen_try, pmat, inv, down, outer, site, ll, ll_try, step, end; };   // in doubles
LikWs lik_ws(int B, int nA, int T, int L, int nc) {
  LikWs w; size_t o = 0;
  auto take = [&](size_t n) { size_t r = o; o += align_up(n, 32); return r; };
  const size_t NN = 2 * (size_t)T - 2;
  w.prog = take(((size_t)B * (T - 1) * 2 + 1) / 2);
  w.colour = take(((size_t)B * NN + 1) / 2);
  w.brlen = take(B * NN); w.brlen_new = take(B * NN); w.brlen_try = take(B * NN);
  w.pmat = take(B * NN * nc * 16);
  w.inv = take((size_t)nA * L);
  w.down = take((size_t)B * (T - 1) * nc * 4 * L);
  w.outer = take(B * NN * nc * 4 * L);
  w.site = take((size_t)B * L);
  w.ll = take(B); w.ll_try = take(B); w.step = take(B);
  w.end = o;
  return w;
}

// accept a tried sweep per tree: ll_try >= ll -> take it (step = 0: done); else halve the step
__global__ void k_lik_accept(double* __restrict__ ll, const double* __restrict__ ll_try, double* __restrict__ step,
                             double* __restrict__ brlen, const double* __restrict__ brlen_try, int B, int NN) {
  const int b = blockIdx.x;
  const bool active = step[b] > 0.0;
  const bool better = active && ll_try[b] >= ll[b];
  __syncthreads();
  if (better) for (int i = threadIdx.x; i < NN; i += blockDim.x) brlen[(size_t)b * NN + i] = brlen_try[(size_t)b * NN + i];
  if (threadIdx.x == 0 && active) {
    if (better) { ll[b] = ll_try[b]; step[b] = 0.0; } else step[b] *= 0.5;
  }
}
__global__ void k_fill(double* __restrict__ p, double v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

int lik_eval(nnj_handle* h, const uint8_t* codes, int nA, const LikModel& md, const LikWs& w, double* base,
             const double* brlen, double* ll_out, int B, int T, int L, hipStream_t st) {
  const int NN = 2 * T - 2, nc = md.ncat;
  const int total = B * NN * nc;
  hipLaunchKernelGGL(k_lik_pmats, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, brlen, md, base + w.pmat, total);
  hipLaunchKernelGGL(k_lik_down, dim3((unsigned)((L + 127) / 128), (unsigned)B, (unsigned)nc), dim3(128), 0, st, codes, nA,
                     reinterpret_cast<const int*>(base + w.prog), (const double*)(base + w.pmat), md, base + w.down, T, L);
  hipLaunchKernelGGL(k_lik_site_ll, dim3((unsigned)((L + 127) / 128), (unsigned)B), dim3(128), 0, st,
                     (const double*)(base + w.inv), nA, md, (const double*)(base + w.down), base + w.site, T, L);
  hipLaunchKernelGGL(k_lik_sum_sites, dim3((unsigned)B), dim3(256), 0, st, (const double*)(base + w.site), ll_out, L);
  return NNJ_OK;
}

int lik_common(nnj_handle* h, const uint8_t* codes, int nA, const uint8_t* mask, const int32_t* merges, const float* brlen_in,
               const nnj_subst_model* model, int B, int T, int L, void* ws, size_t ws_bytes, LikModel& md, LikWs& w,
               double*& base, hipStream_t st) {
  // (the caller holds the DevGuard and has selected the handle's device: it covers every launch of the entry point)
  if (!codes || !merges || !model || B <= 0 || T < 2 || T > 256 || L <= 0 || (nA != 1 && nA != B))
    return fail(h, NNJ_ERR_ARG, "tree likelihood: bad argument (2 <= T <= 256, n_align = 1 or B)");
  if (int rc = build_model(model, md, h->err)) return rc;
  w = lik_ws(B, nA, T, L, md.ncat);
  if (!ws || ws_bytes < w.end * sizeof(double) || (uintptr_t)ws % 256)
    return fail(h, NNJ_ERR_WORKSPACE, "tree likelihood: workspace too small or misaligned (%zu bytes needed)", w.end * sizeof(double));
  h->sess.valid = false;
  base = static_cast<double*>(ws);
  int* prog = reinterpret_cast<int*>(base + w.prog);
  hipLaunchKernelGGL(k_lik_program, dim3((unsigned)B), dim3(256), 0, st, merges, prog,
                     reinterpret_cast<int*>(base + w.colour), B, T, h->d_flag);
  const int ne = B * (T - 1) * 2;
  hipLaunchKernelGGL(k_lik_brlen_init, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, (const int*)prog, brlen_in, 0.1,
                     base + w.brlen, B, T);
  hipLaunchKernelGGL(k_lik_invariant, dim3((unsigned)((nA * L + 255) / 256)), dim3(256), 0, st, codes, mask, md, base + w.inv,
                     nA, T, L);
  return NNJ_OK;
}

}  // namespace

extern "C" {

int nnj_lik_workspace_bytes(int32_t B, int32_t n_align, int32_t T, int32_t L, int32_t ncat, size_t* bytes) {
  if (!bytes || B <= 0 || T < 2 || L <= 0 || ncat < 1 || ncat > LIK_MAXCAT) return NNJ_ERR_ARG;
  *bytes = lik_ws(B, n_align, T, L, ncat).end * sizeof(double);
  return NNJ_OK;
}

int nnj_tree_loglik(nnj_handle* h, const uint8_t* codes, int32_t n_align, const uint8_t* mask, const int32_t* merges,
                    const float* brlen, const nnj_subst_model* model, int32_t B, int32_t T, int32_t L, double* loglik_out,
                    void* ws, size_t ws_bytes, void* stream) {
  LikModel md; LikWs w; double* base = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  DevGuard dev_guard;                                      // for the whole call: every launch below goes to the handle's device
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (!loglik_out) return fail(h, NNJ_ERR_ARG, "nnj_tree_loglik: null output");
  if (int rc = lik_common(h, codes, n_align, mask, merges, brlen, model, B, T, L, ws, ws_bytes, md, w, base, st)) return rc;
  lik_eval(h, codes, n_align, md, w, base, base + w.brlen, loglik_out, B, T, L, st);
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

int nnj_lik_model_probe(nnj_handle* h, const nnj_subst_model* model, double t, double* Q_out, double* rates_out,
                        double* P_out) {
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (!model || !Q_out || !rates_out || !P_out || !(t >= 0.0)) return fail(h, NNJ_ERR_ARG, "nnj_lik_model_probe: bad argument");
  LikModel md;
  if (int rc = build_model(model, md, h->err)) return rc;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += md.U[i * 4 + k] * md.lam[k] * md.Uinv[k * 4 + j];
      Q_out[i * 4 + j] = s;
    }
  for (int c = 0; c < md.ncat; ++c) rates_out[c] = md.rates[c];
  double* d = nullptr;
  HIPCHK(h, hipMalloc(&d, (size_t)(1 + md.ncat * 16) * sizeof(double)));
  hipError_t e = hipMemcpy(d, &t, sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_lik_pmats, dim3(1), dim3(256), 0, nullptr, (const double*)d, md, d + 1, md.ncat);
    e = hipMemcpy(P_out, d + 1, (size_t)md.ncat * 16 * sizeof(double), hipMemcpyDeviceToHost);
  }
  hipFree(d);
  if (e != hipSuccess) return fail(h, NNJ_ERR_HIP, "nnj_lik_model_probe: %s", hipGetErrorString(e));
  return NNJ_OK;
}

int nnj_tree_optimize(nnj_handle* h, const uint8_t* codes, int32_t n_align, const uint8_t* mask, const int32_t* merges,
                      const float* brlen_in, const nnj_subst_model* model, int32_t sweeps, int32_t B, int32_t T, int32_t L,
                      float* brlen_out, double* loglik_out, void* ws, size_t ws_bytes, void* stream) {
  LikModel md; LikWs w; double* base = nullptr;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  DevGuard dev_guard;                                      // for the whole call: every launch below goes to the handle's device
  HIPCHK(h, hipSetDevice(h->cfg.device));
  if (!loglik_out || sweeps < 0) return fail(h, NNJ_ERR_ARG, "nnj_tree_optimize: bad argument");
  if (int rc = lik_common(h, codes, n_align, mask, merges, brlen_in, model, B, T, L, ws, ws_bytes, md, w, base, st)) return rc;
  const int NN = 2 * T - 2;
  const int* prog = reinterpret_cast<const int*>(base + w.prog);
  // the two child edges of the last join are one edge of the unrooted tree: folded for the sweeps, split on export
  const bool fold = getenv("NNJ_LIK_NOFOLD") == nullptr;
  if (sweeps > 0 && fold)
    hipLaunchKernelGGL(k_lik_root_fold, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, prog, base + w.brlen, B, T, 0);
  lik_eval(h, codes, n_align, md, w, base, base + w.brlen, base + w.ll, B, T, L, st);
  const int* colour = reinterpret_cast<const int*>(base + w.colour);
  for (int sw = 0; sw < sweeps; ++sw) {
    // one Gauss-Seidel pass = four colour steps (k_lik_program): partials on both sides of every edge from the current
    // lengths (pmat and down are current), one Newton-Raphson solve per edge of the colour, the step taken in full if
    // the likelihood does not drop, else halved (per tree, decided on the device), then the partials refreshed
    for (int c = 0; c < 4; ++c) {
      hipLaunchKernelGGL(k_lik_outer, dim3((unsigned)((L + 127) / 128), (unsigned)B, (unsigned)md.ncat), dim3(128), 0, st, codes, n_align, prog,
                         (const double*)(base + w.pmat), md, (const double*)(base + w.down), base + w.outer, T, L);
      hipLaunchKernelGGL(k_lik_newton, dim3((unsigned)NN, (unsigned)B), dim3(256), 0, st, codes, n_align,
                         (const double*)(base + w.inv), (const double*)(base + w.down), (const double*)(base + w.outer), md,
                         fold ? prog : (const int*)nullptr, colour, c, (const double*)(base + w.brlen), base + w.brlen_new, T, L, 12);
      hipLaunchKernelGGL(k_fill, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, base + w.step, 1.0, B);
      for (int tr = 0; tr < 2; ++tr) {
        hipLaunchKernelGGL(k_lik_blend, dim3((unsigned)((B * NN + 255) / 256)), dim3(256), 0, st, (const double*)(base + w.brlen),
                           (const double*)(base + w.brlen_new), (const double*)(base + w.step), base + w.brlen_try, B, NN);
        lik_eval(h, codes, n_align, md, w, base, base + w.brlen_try, base + w.ll_try, B, T, L, st);
        hipLaunchKernelGGL(k_lik_accept, dim3((unsigned)B), dim3(64), 0, st, base + w.ll, (const double*)(base + w.ll_try),
                           base + w.step, base + w.brlen, (const double*)(base + w.brlen_try), B, NN);
      }
      lik_eval(h, codes, n_align, md, w, base, base + w.brlen, base + w.ll, B, T, L, st);   // pmat / down of the accepted lengths
    }
  }
  HIPCHK(h, hipMemcpyAsync(loglik_out, base + w.ll, (size_t)B * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (brlen_out) {
    const int ne = B * (T - 1) * 2;
    if (sweeps > 0 && fold)
      hipLaunchKernelGGL(k_lik_root_fold, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, st, prog, base + w.brlen, B, T, 1);
    hipLaunchKernelGGL(k_lik_brlen_export, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, prog,
                       (const double*)(base + w.brlen), brlen_out, B, T);
  }
  HIPCHK(h, hipGetLastError());
  return NNJ_OK;
}

}  // extern "C"

extern "C" {
int nnj_debug_encoder_stop(nnj_handle* h, int32_t stage) {
  if (!h) return NNJ_ERR_ARG;
  h->debug_stop = stage;
  return NNJ_OK;
}

int nnj_profile_enable(nnj_handle* h, int32_t on) {
  if (!h) return fail(nullptr, NNJ_ERR_ARG, "null handle");
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  h->prof = on != 0;
  h->ev_used = 0;
  h->prof_dropped = 0;
  for (int k = 0; k < PK_COUNT; ++k) { h->prof_ms[k] = 0; h->prof_n[k] = 0; }
  return NNJ_OK;
}

int nnj_numeric_status(nnj_handle* h, int32_t* nonfinite_out, void* stream) {
  if (!h || !nonfinite_out) return fail(h, NNJ_ERR_ARG, "nnj_numeric_status: null argument");
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  hipStream_t st = static_cast<hipStream_t>(stream);
  int v = 0;
  HIPCHK(h, hipMemcpyAsync(&v, h->d_flag, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemsetAsync(h->d_flag, 0, sizeof(int), st));
  HIPCHK(h, hipStreamSynchronize(st));
  *nonfinite_out = v;
  return NNJ_OK;
}

int nnj_profile_dropped(nnj_handle* h, int64_t* dropped_out) {
  if (!h || !dropped_out) return fail(h, NNJ_ERR_ARG, "nnj_profile_dropped: null argument");
  *dropped_out = h->prof_dropped;
  return NNJ_OK;
}

int nnj_profile_kinds(void) { return PK_COUNT; }
const char* nnj_profile_kind_name(int32_t k) { return (k >= 0 && k < PK_COUNT) ? kProfNames[k] : ""; }

int nnj_profile_read(nnj_handle* h, double* ms_out, int64_t* launches_out, int32_t cap) {
  if (!h || !ms_out || !launches_out || cap < PK_COUNT) return fail(h, NNJ_ERR_ARG, "nnj_profile_read: bad argument");
  DevGuard dev_guard;
  HIPCHK(h, hipSetDevice(h->cfg.device));
  for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
    HIPCHK(h, hipEventSynchronize(h->ev[i + 1]));
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
    const int k = h->ev_kind[i / 2];
    h->prof_ms[k] += ms;
    h->prof_n[k] += 1;
  }
  h->ev_used = 0;
  for (int k = 0; k < PK_COUNT; ++k) {
    ms_out[k] = h->prof_ms[k]; launches_out[k] = h->prof_n[k];
    h->prof_ms[k] = 0; h->prof_n[k] = 0;
  }
  return NNJ_OK;
}

}  // extern "C"
