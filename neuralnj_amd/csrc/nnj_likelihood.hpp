// nnj_likelihood.hpp -- tree log-likelihood on the GPU: Felsenstein pruning under GTR+I+G for batches of trees given
// as merge lists.  It stands where the reference calls raxmlpy.optimize_brlen (raxml-ng /
// libpll, CPU) to rank sampled trees: environment.py:365-441, 625-672, finetune_rl_search.py:401-411.  The
// arithmetic of that call lives in un-vendored third-party code (parity UNPINNED): this file follows the
// published algorithm -- Felsenstein (1981) pruning; GTR with discrete-gamma rate heterogeneity (Yang 1994, mean
// rate per category) and a proportion of invariant sites; Newton-Raphson branch-length optimisation on the
// eigen-space "sum table" of an edge -- and is checked against closed forms and an independent numpy/scipy
// evaluation (oracle/lik_oracle.py, tests/test_likelihood.py).
//
// A tree is the merge list of a rollout: step s joins positions (i, j) of the current row list (merged node ->
// position i, position j deleted).  Node ids: leaves 0..T-1 (the alignment rows), the join of step s = T + s; the
// last join is the root of a rooted view whose two child edges form one edge of the unrooted tree.  Everything is
// fp64 (per-site likelihoods of 200-taxon trees underflow fp32; no scaling is needed in fp64 up to 256 taxa).
//   prog   int32 [B][T-1][2]        child node ids of every join
//   brlen  double [B][2T-2]         length of the edge ABOVE node id v (v < 2T-2; the root has none)
//   pmat   double [B][2T-2][ncat][16]
//   down   double [B][T-1][ncat*4][L]     D_v: partial likelihood of the data below join v, given the state at v
//   outer  double [B][2T-2][ncat*4][L]    O_v: ... of the data NOT below v, given the state at the parent of v
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct LikModel {                 // eigen system of the normalised GTR rate matrix, category rates, +I
  double U[16], Uinv[16], lam[4]; // Q = U diag(lam) Uinv
  double freqs[4];
  double rates[8];                // discrete-gamma category rates (mean 1)
  double pinv;
  int ncat;
};

constexpr int LIK_MAXCAT = 8;

// merges -> child node ids per join; one thread per tree.  Also the COLOUR of every edge for the branch-length sweeps
// (colour[b][v], edge above node v): 2 * (depth of v mod 2) + (0 / 1: first / second child of its join).  Two edges
// of one colour never share a node -- siblings differ in the second term, parent and child in the first -- so the
// edges of a colour can be optimised at once from the same partial likelihoods without fighting each other, and
// four colour steps are one Gauss-Seidel pass over the tree.
// One WORKGROUP per tree (round 4; it was one thread with a 256-entry private list in scratch memory: 658 us per call
// at 200 taxa -- a seventh of a Search round's model optimisation, which calls it for every trial value): the position
// list lives in LDS and "position j leaves the list" (environment.py:764-768) is a parallel shift.
// A malformed merge (i >= j, or a position outside the live list) is REPORTED: it sets NNJ_FLAG_BAD_MERGE in the handle's
// sticky status word (nnj_numeric_status); the program is still built (positions clamped, position i wins) so that the
// launch sequence stays defined, but its likelihoods must be discarded.
__global__ __launch_bounds__(256) void k_lik_program(const int* __restrict__ merges, int* __restrict__ prog,
                                                     int* __restrict__ colour, int B, int T, int* __restrict__ flag) {
  __shared__ int ids[256];
  __shared__ int pl[512];                                  // the tree's child ids per join (for the colour pass)
  __shared__ int cl[512];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= B) return;
  ids[tid] = tid;
  const int* m = merges + (size_t)b * (T - 1) * 2;
  int* p = prog + (size_t)b * (T - 1) * 2;
  __syncthreads();
  int n = T;
  for (int s = 0; s < T - 1; ++s, --n) {
    const int i = min(max(m[2 * s], 0), n - 1), j = min(max(m[2 * s + 1], 0), n - 1);
    if (tid == 0 && (m[2 * s] != i || m[2 * s + 1] != j || i >= j)) atomicOr(flag, NNJ_FLAG_BAD_MERGE);
    if (tid == 0) { pl[2 * s] = ids[i]; pl[2 * s + 1] = ids[j]; }
    int v = (tid >= j && tid < n - 1) ? ids[tid + 1] : ids[tid];
    if (tid == i) v = T + s;                                // (a malformed merge, i >= j: position i wins, and the flag is set)
    __syncthreads();
    ids[tid] = v;
    __syncthreads();
  }
  for (int e = tid; e < 2 * (T - 1); e += 256) p[e] = pl[e];
  if (!colour) return;
  // top-down: the depth parity of a join is kept in the colour entry of the join itself (the last join is the root)
  if (tid == 0) {
    for (int s = T - 2; s >= 0; --s) {
      const int par = s == T - 2 ? 0 : (cl[T + s] >> 1);    // depth parity of join T + s
      const int cp = par ^ 1;
      cl[pl[2 * s]] = 2 * cp; cl[pl[2 * s + 1]] = 2 * cp + 1;
    }
  }
  __syncthreads();
  int* col = colour + (size_t)b * (2 * T - 2);
  for (int e = tid; e < 2 * T - 2; e += 256) col[e] = cl[e];
}

// child-edge lengths in merge order [B][T-1][2] (or a constant) -> per-node lengths [B][2T-2]
__global__ void k_lik_brlen_init(const int* __restrict__ prog, const float* __restrict__ brlen_merge, double deflt,
                                 double* __restrict__ brlen, int B, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (T - 1) * 2) return;
  const int b = i / ((T - 1) * 2);
  const int v = prog[i];
  double t = brlen_merge ? (double)brlen_merge[i] : deflt;
  if (!(t > 1e-8)) t = 1e-8;
  brlen[(size_t)b * (2 * T - 2) + v] = t;
}
// per-node lengths back to merge order (float)
__global__ void k_lik_brlen_export(const int* __restrict__ prog, const double* __restrict__ brlen,
                                   float* __restrict__ brlen_merge, int B, int T) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (T - 1) * 2) return;
  const int b = i / ((T - 1) * 2);
  brlen_merge[i] = (float)brlen[(size_t)b * (2 * T - 2) + prog[i]];
}

// The two child edges of the last join are ONE edge of the unrooted tree (pulley principle: under a reversible model
// the likelihood depends on their sum only).  Optimised as two edges by a Jacobi sweep, each is solved with the other
// held at its old value -- t1' = s* - t2, t2' = s* - t1 -- and the sum is mirrored about the optimum (2 s* - s)
// instead of converging.  So the pair is kept FOLDED while lengths are optimised (all of the sum on the first child,
// 1e-8 on the second, which k_lik_newton leaves alone) and split evenly when lengths are exported.
__global__ void k_lik_root_fold(const int* __restrict__ prog, double* __restrict__ brlen, int B, int T, int split) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int* pr = prog + ((size_t)b * (T - 1) + (T - 2)) * 2;
  double* t = brlen + (size_t)b * (2 * T - 2);
  const double s = t[pr[0]] + t[pr[1]];
  if (split) { t[pr[0]] = 0.5 * s; t[pr[1]] = 0.5 * s; }
  else { t[pr[0]] = fmax(s - 1e-8, 1e-8); t[pr[1]] = 1e-8; }
}

// P(t) = U diag(exp(lam * rate_c * t)) Uinv for every (tree, node, category)
__global__ void k_lik_pmats(const double* __restrict__ brlen, LikModel md, double* __restrict__ pmat, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;     // (b, v, c)
  if (i >= total) return;
  const int c = i % md.ncat;
  const double t = brlen[i / md.ncat] * md.rates[c];
  double e[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) e[k] = exp(md.lam[k] * t);
  double* P = pmat + (size_t)i * 16;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) s += md.U[a * 4 + k] * e[k] * md.Uinv[k * 4 + d];
      P[a * 4 + d] = s < 0.0 ? 0.0 : s;
    }
}

// tip vector of a site code (0..3 = A,C,G,T; 4 = gap/N and 5 = padding: every state possible)
__device__ __forceinline__ void lik_tip(double (&x)[4], int code) {
#pragma unroll
  for (int k = 0; k < 4; ++k) x[k] = (code > 3 || code == k) ? 1.0 : 0.0;
}
// y[a] = sum_d P[a][d] x[d]
__device__ __forceinline__ void lik_apply(double (&y)[4], const double* __restrict__ P, const double (&x)[4]) {
#pragma unroll
  for (int a = 0; a < 4; ++a) y[a] = P[a * 4] * x[0] + P[a * 4 + 1] * x[1] + P[a * 4 + 2] * x[2] + P[a * 4 + 3] * x[3];
}

// inv[a][c] = sum of pi_k over the states k every taxon of site c is compatible with (0 for a variable site): the
// likelihood of the site under the invariant-sites component.  Masked (padded) sites get -1: "skip this site".
__global__ void k_lik_invariant(const uint8_t* __restrict__ codes, const uint8_t* __restrict__ mask, LikModel md,
                                double* __restrict__ inv, int nA, int T, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nA * L) return;
  const int a = i / L, c = i % L;
  if (mask && mask[(size_t)a * L + c]) { inv[i] = -1.0; return; }
  unsigned poss = 0xF;
  for (int r = 0; r < T; ++r) { const int code = codes[((size_t)a * T + r) * L + c]; if (code <= 3) poss &= 1u << code; }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) if (poss & (1u << k)) s += md.freqs[k];
  inv[i] = s;
}

// Post-order pass: one thread per (tree, site, RATE CATEGORY) walks the joins in merge order (children always come
// first).  Round 4: the categories are a grid dimension (they are independent until the site likelihood is formed) --
// one alignment of 4096 sites was 32 waves on 256 CUs, each a chain of 199 joins x 4 categories of dependent loads;
// k_lik_site_ll then forms site_ll [B][L] (0 for masked / padded sites) from the root partials.
// codes [nA][T][L], nA = 1 or B.  grid (ceil(L / 128), B, ncat).
__global__ __launch_bounds__(128) void k_lik_down(const uint8_t* __restrict__ codes, int n_align,
                                                  const int* __restrict__ prog, const double* __restrict__ pmat,
                                                  LikModel md, double* __restrict__ down, int T, int L) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, cat = blockIdx.z;
  if (c >= L) return;
  const int nc = md.ncat, NN = 2 * T - 2;
  const uint8_t* cd = codes + (size_t)(n_align == 1 ? 0 : b) * T * L + c;
  const int* pg = prog + (size_t)b * (T - 1) * 2;
  const double* Pb = pmat + (size_t)b * NN * nc * 16;
  double* Db = down + (size_t)b * (T - 1) * nc * 4 * L + c;
  for (int s = 0; s < T - 1; ++s) {
    const int ch[2] = {pg[2 * s], pg[2 * s + 1]};
    double acc[4] = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      const int v = ch[side];
      double x[4], y[4];
      if (v < T) lik_tip(x, cd[(size_t)v * L]);
      else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = Db[((size_t)(v - T) * nc * 4 + cat * 4 + k) * L];
      }
      lik_apply(y, Pb + ((size_t)v * nc + cat) * 16, x);
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] *= y[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) Db[((size_t)s * nc * 4 + cat * 4 + k) * L] = acc[k];      // (the root join too)
  }
}
// site likelihood: pinv * [constant site] + (1 - pinv) / ncat * sum_cat sum_k pi_k root_k (categories added in order)
__global__ __launch_bounds__(128) void k_lik_site_ll(const double* __restrict__ inv, int n_align, LikModel md,
                                                     const double* __restrict__ down, double* __restrict__ site_ll,
                                                     int T, int L) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (c >= L) return;
  const int nc = md.ncat;
  const double* Rb = down + ((size_t)b * (T - 1) + (T - 2)) * nc * 4 * L + c;
  double g = 0.0;
  for (int cat = 0; cat < nc; ++cat)
#pragma unroll
    for (int k = 0; k < 4; ++k) g += md.freqs[k] * Rb[(size_t)(cat * 4 + k) * L];
  g *= (1.0 - md.pinv) / (double)nc;
  const double iv = inv[(size_t)(n_align == 1 ? 0 : b) * L + c];
  g += md.pinv * fmax(iv, 0.0);
  site_ll[(size_t)b * L + c] = iv < 0.0 ? 0.0 : log(g);
}

// deterministic sum over sites: one workgroup per tree, fixed-order tree reduction
__global__ __launch_bounds__(256) void k_lik_sum_sites(const double* __restrict__ site_ll, double* __restrict__ out, int L) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  double s = 0.0;
  for (int c = tid; c < L; c += 256) s += site_ll[(size_t)b * L + c];
  red[tid] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) out[b] = red[0];
}

// Pre-order pass: O_v for every non-root node (see the header).  One thread per (tree, site), joins walked from the
// root down; U_p[i] = sum_k P_ik(t_p) O_p[k] (U_root = 1), O_v = (P(t_sibling) D_sibling) * U_p.
__global__ __launch_bounds__(128) void k_lik_outer(const uint8_t* __restrict__ codes, int n_align,
                                                   const int* __restrict__ prog, const double* __restrict__ pmat,
                                                   LikModel md, const double* __restrict__ down,
                                                   double* __restrict__ outer, int T, int L) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y, cat = blockIdx.z;     // (categories: a grid dimension)
  if (c >= L) return;
  const int nc = md.ncat, NN = 2 * T - 2;
  const uint8_t* cd = codes + (size_t)(n_align == 1 ? 0 : b) * T * L + c;
  const int* pg = prog + (size_t)b * (T - 1) * 2;
  const double* Pb = pmat + (size_t)b * NN * nc * 16;
  const double* Db = down + (size_t)b * (T - 1) * nc * 4 * L + c;
  double* Ob = outer + (size_t)b * NN * nc * 4 * L + c;
  for (int s = T - 2; s >= 0; --s) {
    const int p = T + s;                                   // this join; its own O_p was written by its parent's turn
    const int ch[2] = {pg[2 * s], pg[2 * s + 1]};
    {
      double up[4] = {1.0, 1.0, 1.0, 1.0};
      if (s != T - 2) {
        double op[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) op[k] = Ob[((size_t)p * nc * 4 + cat * 4 + k) * L];
        lik_apply(up, Pb + ((size_t)p * nc + cat) * 16, op);
      }
      double msg[2][4];                                    // P(t_child) D_child, for both children
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const int v = ch[side];
        double x[4];
        if (v < T) lik_tip(x, cd[(size_t)v * L]);
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) x[k] = Db[((size_t)(v - T) * nc * 4 + cat * 4 + k) * L];
        }
        lik_apply(msg[side], Pb + ((size_t)v * nc + cat) * 16, x);
      }
#pragma unroll
      for (int side = 0; side < 2; ++side)
#pragma unroll
        for (int k = 0; k < 4; ++k) Ob[((size_t)ch[side] * nc * 4 + cat * 4 + k) * L] = msg[1 - side][k] * up[k];
    }
  }
}

// Newton-Raphson on the length of ONE edge per workgroup (tree b, node v): the site likelihood as a function of t is
//   pinv * inv_s + (1 - pinv) / ncat * sum_cat sum_k S[cat][k] exp(lam_k r_cat t),  S = (pi O U)_k (Uinv D)_k
// ("sum table" of the edge, recomputed from O_v and D_v in every iteration: they stay in L2).  Each iteration reduces
// d/dt and d2/dt2 of the log-likelihood over the sites; t <- t - f'/f'' where concave, else a step along the
// gradient; a trust region of [t/4, 4t]; lengths stay in [1e-8, 100].  The edges of ONE colour (k_lik_program) are
// optimised at once from the same partials; the caller re-evaluates the likelihood, damps a step that does not
// improve it, refreshes the partials and goes on to the next colour (round 2 optimised ALL edges at once -- a Jacobi
// sweep -- and needed a dozen sweeps where this needs three: tools/lik_conv.py).
__global__ __launch_bounds__(256) void k_lik_newton(const uint8_t* __restrict__ codes, int n_align,
                                                    const double* __restrict__ inv, const double* __restrict__ down,
                                                    const double* __restrict__ outer, LikModel md,
                                                    const int* __restrict__ prog, const int* __restrict__ colour, int cur,
                                                    const double* __restrict__ brlen, double* __restrict__ brlen_new,
                                                    int T, int L, int iters) {
  __shared__ double red[2][256];
  const int v = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int nc = md.ncat, NN = 2 * T - 2;
  // not this step's colour, or the second child of the last join (folded into the first: k_lik_root_fold): unchanged
  if ((colour && colour[(size_t)b * NN + v] != cur) ||
      (prog && v == prog[((size_t)b * (T - 1) + (T - 2)) * 2 + 1])) {
    if (tid == 0) brlen_new[(size_t)b * NN + v] = brlen[(size_t)b * NN + v];
    return;
  }
  const uint8_t* cdb = codes + (size_t)(n_align == 1 ? 0 : b) * T * L;
  const double* ivb = inv + (size_t)(n_align == 1 ? 0 : b) * L;
  const double* Db = down + (size_t)b * (T - 1) * nc * 4 * L;
  const double* Ob = outer + (size_t)b * NN * nc * 4 * L;
  double t = brlen[(size_t)b * NN + v];
  const double w = (1.0 - md.pinv) / (double)nc;
  for (int it = 0; it < iters; ++it) {
    double g1 = 0.0, g2 = 0.0;
    for (int c = tid; c < L; c += 256) {
      const double iv = ivb[c];
      if (iv < 0.0) continue;                              // masked site
      double f = 0.0, f1 = 0.0, f2 = 0.0;
      for (int cat = 0; cat < nc; ++cat) {
        double o[4], d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = Ob[((size_t)v * nc * 4 + cat * 4 + k) * L + c];
        if (v < T) lik_tip(d, cdb[(size_t)v * L + c]);
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) d[k] = Db[((size_t)(v - T) * nc * 4 + cat * 4 + k) * L + c];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          double a = 0.0, bb = 0.0;
#pragma unroll
          for (int i = 0; i < 4; ++i) { a += md.freqs[i] * o[i] * md.U[i * 4 + k]; bb += md.Uinv[k * 4 + i] * d[i]; }
          const double lr = md.lam[k] * md.rates[cat];
          const double e = a * bb * exp(lr * t);
          f += e; f1 += lr * e; f2 += lr * lr * e;
        }
      }
      f = md.pinv * iv + w * f; f1 *= w; f2 *= w;
      const double r1 = f1 / f;
      g1 += r1;                                            // d/dt log f
      g2 += f2 / f - r1 * r1;                              // d2/dt2 log f
    }
    red[0][tid] = g1; red[1][tid] = g2;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
      if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
      __syncthreads();
    }
    g1 = red[0][0]; g2 = red[1][0];
    __syncthreads();
    double tn;
    if (g2 < 0.0) tn = t - g1 / g2;                        // Newton step towards the maximum
    else tn = g1 > 0.0 ? t * 2.0 : t * 0.5;                // not concave here: move along the gradient
    if (tn > 4.0 * t + 1e-3) tn = 4.0 * t + 1e-3;          // trust region
    if (tn < 0.25 * t) tn = 0.25 * t;
    tn = fmin(fmax(tn, 1e-8), 100.0);
    const bool done = fabs(tn - t) <= 1e-10 + 1e-7 * t;
    t = tn;
    if (done) break;                                       // (uniform: every thread holds the same t)
  }
  if (tid == 0) brlen_new[(size_t)b * NN + v] = t;
}

// t <- t_old + step * (t_new - t_old) for trees whose sweep is (re)tried with a damped step
__global__ void k_lik_blend(const double* __restrict__ t_old, const double* __restrict__ t_new, const double* __restrict__ step,
                            double* __restrict__ t_out, int B, int NN) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * NN) return;
  const double s = step[i / NN];
  t_out[i] = t_old[i] + s * (t_new[i] - t_old[i]);
}
