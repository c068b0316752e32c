/*
 * nnj.h -- C ABI of the MI355X-native NeuralNJ hot path (libnnj_hip.so).
 *
 * Drop-in boundary for the Argmax inference path of DingShizhe/NeuralNJ: the
 * axial-attention MSA encoder plus the iterative neural neighbour-joining loop.
 * Every entry point names the reference interface it replaces (file:line are
 * relative to the reference repository).  Plain pointers and sizes only; no
 * torch types.  All entry points return 0 on success or a negative nnj_status;
 * nnj_last_error() gives a message.  No C++ exception crosses this boundary and
 * nothing here calls exit().
 *
 * Memory: every `dev` pointer is HIP device memory on the handle's device; every
 * `host` pointer is ordinary host memory.  The library allocates only at
 * nnj_create / nnj_load_weights; all scratch comes from the caller-provided
 * workspace (size it with nnj_workspace_bytes).  Launches are asynchronous on
 * the `stream` argument (a hipStream_t passed as void*; NULL = default stream);
 * there is no hidden synchronisation.
 *
 * Layouts (row-major, last index fastest):
 *   codes   uint8  [B,T,L]     site code: 0..3 = A,C,G,T one-hot, 4 = gap/N =
 *                              [1,1,1,1], 5 = padding = [0,0,0,0]
 *                              (lossless form of reference phydata.py:38-46,57-77)
 *   onehot  float  [B,T,L,V]   the reference's own input (V = vocab_size)
 *   mask    uint8  [B,L]       1 = padded site (reference finetune_rl_search.py:93)
 *   state   float  [B,n,C,D]   row embeddings, C = L / patch_size
 *   logits  float  [B,P(n)]    P(n) = n(n-1)/2, pair order = itertools.combinations
 *                              = torch.triu_indices(n,n,1) order
 *                              (reference environment.py:457-462, model.py:176)
 *   merges  int32  [B,T-1,2]   chosen (i,j), i<j, row indices at that step
 */
#ifndef NNJ_H
#define NNJ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNJ_ABI_VERSION 1

typedef enum nnj_status {
  NNJ_OK = 0,
  NNJ_ERR_ARG = -1,         /* bad argument (null pointer, bad size)          */
  NNJ_ERR_UNSUPPORTED = -2, /* model/config shape the HIP kernels do not cover */
  NNJ_ERR_NO_WEIGHTS = -3,  /* compute call before nnj_load_weights           */
  NNJ_ERR_WORKSPACE = -4,   /* workspace too small                            */
  NNJ_ERR_HIP = -5,         /* a HIP runtime call failed                      */
  NNJ_ERR_NO_DEVICE = -6    /* no usable gfx950 device                        */
} nnj_status;

/* Mirrors cfgs.model.* read by PhyloATTN.__init__ (reference model.py:15-22). */
typedef struct nnj_config {
  int32_t vocab_size;  /* cfgs.model.vocab_size   (4)                      */
  int32_t patch_size;  /* cfgs.model.patch_size   (1 in the shipped yaml; 1..16 supported: a TOKEN of the state is
                          patch_size consecutive sites, model.py:72-79.  Every entry point takes L = SITES (a multiple
                          of patch_size), codes / one-hot input [*,T,L(,4)] and masks [*,L] per site; state tensors are
                          [B, rows, L / patch_size, 64]; the mask of a token is the mask of its first site.)        */
  int32_t embed_dim;   /* cfgs.model.embed_dim    (64 in the shipped yaml; 8..64 in multiples of 8 supported with
                          num_heads = embed_dim / 8, e.g. the reference's own default 32 / 4 heads, utils.py:45-52.
                          The kernels work on 64 features: a narrower model runs zero-padded -- exact, every padded
                          feature stays zero through the whole path -- and its dense state tensors keep their own
                          width [.., embed_dim] at this interface; ffn width = 4 * embed_dim, model.py:28)         */
  int32_t num_heads;   /* cfgs.model.num_enc_heads (8; must be embed_dim / 8: heads of 8 features)                */
  int32_t num_layers;  /* cfgs.model.num_enc_layers (6)                    */
  int32_t device;      /* HIP device ordinal                               */
} nnj_config;

typedef struct nnj_handle nnj_handle;

int nnj_abi_version(void);

/* PhyloATTN(cfgs).to(device) -- reference model.py:11-60, finetune_rl_search.py:482. */
int nnj_create(const nnj_config* cfg, nnj_handle** out);
int nnj_destroy(nnj_handle* h);
/* Message for the last failing call on this handle (h == NULL: last nnj_create). */
const char* nnj_last_error(const nnj_handle* h);

/* Number of fp32 parameters of the packed vector for this config. */
int nnj_num_params(const nnj_config* cfg, size_t* n);

/* load_state_dict(checkpoint['model_state_dict']) -- reference
 * finetune_rl_search.py:481-484.  `packed_host` is the concatenation of the 172
 * state_dict tensors in state_dict order. Synchronous. */
int nnj_load_weights(nnj_handle* h, const float* packed_host, size_t n);

/* Scratch bytes needed by any entry point below for shapes up to (B,T,L). */
int nnj_workspace_bytes(const nnj_handle* h, int32_t B, int32_t T, int32_t L, size_t* bytes);

/* Host-only self check of the launch geometry behind nnj_workspace_bytes (no device, no handle): does every scorer
 * launch of a rollout of B alignments x T rows x C tokens -- the batch itself and the one-alignment form
 * nnj_rollout_sample uses for the first table of replicas -- fit the workspace regions reserved for it?  Returns 0 if
 * so, NNJ_ERR_ARG for a bad shape, otherwise the (positive) number of the first region that is too small.  No
 * reference counterpart: the reference allocates per operator (model.py:168-201). */
int nnj_workspace_selfcheck(int32_t B, int32_t T, int32_t C);

/* PhyloATTN.encode_zxr -- reference model.py:67-88 (+ msa_modules.py:62-151,
 * axial_attention.py:6-255).  Exactly one of codes_dev / onehot_dev is non-NULL: codes_dev takes the
 * 6-entry LUT fast path; onehot_dev (float [B,T,L,4], any values) runs the embed MLP on the device. */
int nnj_encode(nnj_handle* h, const uint8_t* codes_dev, const float* onehot_dev,
               const uint8_t* mask_dev, float* state_out_dev,
               int32_t B, int32_t T, int32_t L,
               void* ws_dev, size_t ws_bytes, void* stream);

/* PhyloATTN.decode_zxr with logits_prev=None -- reference model.py:168-181
 * (decode_gg 90-99, aggregate 102-155): scores of all P(n) pairs. */
int nnj_pair_scores_full(nnj_handle* h, const float* state_dev, const uint8_t* mask_dev,
                         float* logits_out_dev, int32_t B, int32_t n, int32_t L,
                         void* ws_dev, size_t ws_bytes, void* stream);

/* PhyloATTN.decode_zxr with logits_prev given -- reference model.py:184-201
 * together with utils.get_score_indices_to_prev (utils.py:213-251): scores the n
 * pairs (i_prev, r), then assembles the P(n) table from logits_prev [B,P(n+1)] by
 * the old->new index map computed on the device from ij_prev [B,2]. */
int nnj_pair_scores_incr(nnj_handle* h, const float* state_dev, const uint8_t* mask_dev,
                         const int32_t* ij_prev_dev, const float* logits_prev_dev,
                         float* logits_out_dev, int32_t B, int32_t n, int32_t L,
                         void* ws_dev, size_t ws_bytes, void* stream);

/* utils.get_score_indices_to_prev alone -- reference utils.py:213-251:
 * idx_out int64 [B,P(n)] indexes cat(logits_prev[P(n+1)], new_scores[n]). */
int nnj_score_index_map(nnj_handle* h, const int32_t* ij_prev_dev, int64_t* idx_out_dev,
                        int32_t B, int32_t n, void* stream);

/* PhyloATTN.aggregate(x_i, x_j, (ii,jj), batchwise_ij_indices=True) as called from
 * PhyInferEnv.step -- reference environment.py:822-831, model.py:102-155.  Rows
 * ij_dev[b] = (i,j) of state [B,n,C,D] are merged with the other n-2 rows as
 * context; writes one row per batch element: out [B,1,C,D]. */
int nnj_aggregate(nnj_handle* h, const float* state_dev, const int32_t* ij_dev,
                  float* out_row_dev, int32_t B, int32_t n, int32_t L,
                  void* ws_dev, size_t ws_bytes, void* stream);

/* Tensor half of PhyInferEnv.step -- reference environment.py:760-835: aggregate
 * rows (i,j), put the merged row in slot i, drop slot j, shift later rows down.
 * state_out [B,n-1,C,D] must not alias state. */
int nnj_env_step(nnj_handle* h, const float* state_dev, const int32_t* ij_dev,
                 float* state_out_dev, int32_t B, int32_t n, int32_t L,
                 void* ws_dev, size_t ws_bytes, void* stream);

/* Sessions over the dense-state entry points.  The reference's loop (finetune_rl_search.py:121-164) hands the dense
 * state from call to call: decode_zxr(state) -> argmax -> env.step(state) -> decode_zxr(new state) ...  The library
 * keeps the rows of the tensor it last saw in nnj_pair_scores_full, or last returned from nnj_env_step / nnj_step, in
 * slot layout inside the caller's workspace, with their cached per-row transforms and the live list -- the state of
 * nnj_rollout_argmax's own loop.  nnj_pair_scores_incr, nnj_aggregate, nnj_env_step and nnj_step CONTINUE that session
 * when their state_dev is that very tensor: same pointer, same workspace pointer, B, L and row count, and the caller
 * has not written to it since.  They then transform no row again, merge in place and gather the dense output once
 * (the reference copies the whole state twice per step); results are bit-identical to nnj_rollout_argmax's.  Any
 * other input runs stateless from the dense tensor, as documented per function.  A caller that rewrites a tensor the
 * session is bound to (or frees it and reuses the address) must call nnj_session_reset first. */
int nnj_session_reset(nnj_handle* h);

/* argmax(logits,-1) and flat index -> (i,j) -- reference
 * finetune_rl_search.py:145,159-160 (first maximal index wins).
 * ij_out int32 [B,2]; top2_gap_out float [B] may be NULL. */
int nnj_select_pair(nnj_handle* h, const float* logits_dev, int32_t* ij_out_dev,
                    float* top2_gap_out_dev, int32_t B, int32_t n, void* stream);

/* One iteration of the reference's loop body in its device order -- environment.py:760-835 (merge the pair chosen in
 * the previous iteration), then model.py:184-201 + utils.py:213-251 (score the n new pairs, assemble the table), then
 * finetune_rl_search.py:145,159-160 (argmax) -- as ONE fused step with no host round trip: the kernels of
 * nnj_rollout_argmax's loop body, once each.
 *   state_dev        [B,n+1,C,D]   rows before the merge          ij_dev        int32 [B,2] the pair to merge
 *   logits_prev_dev  [B,P(n+1)]    the table ij_dev was chosen from
 *   forced_next_dev  NULL, or int32 [B,2]: returned in chosen_ij_dev instead of the argmax (teacher forcing)
 *   state_out_dev    [B,n,C,D]     logits_out_dev [B,P(n)]   chosen_ij_dev int32 [B,2]   top2_gap_dev [B] or NULL
 * n = number of rows AFTER the merge, n >= 2.
 * A loop of calls is a session (see "Sessions" above): the library keeps the rows in slot layout, with the cached
 * per-row transforms, inside the caller's workspace, so consecutive steps re-transform nothing.  The session continues
 * when state_dev is the tensor the library last saw or returned (same workspace pointer, B, L, and n+1 = the session's
 * row count); anything else -- another tensor, another workspace, an entry point other than the dense-state ones
 * (nnj_encode, nnj_rollout_*) on the same workspace in between -- starts a new session from the dense state_dev (copy
 * + row transforms).  state_out_dev is the dense tensor the reference's env.step returns.
 * Up to 64 rows the step runs the two-pass kernels of nnj_rollout_argmax's loop: the merged row is produced inside the
 * alpha pass of the new pairs and its attention weights are the ones the PREVIOUS call's table kernel prepared for the
 * pair it returned in chosen_ij_dev -- merging that pair (the usual loop) costs no extra pass; any other ij_dev is
 * served by per-alignment fallback kernels.  Such a session is continued by nnj_step only (the other dense-state entry
 * points start again from the tensor they are given).  Iterated from the encoder output it gives nnj_rollout_argmax's
 * merges and its tables to fp32 rounding (the first merge takes its weights from the fallback kernels, the rollout from
 * the all-pairs kernel's partial sums); above 64 rows, and on a handle created under NNJ_TWO_PASS=0, the four-pass
 * kernels, whose loop it reproduces bit for bit. */
int nnj_step(nnj_handle* h, const float* state_dev, const uint8_t* mask_dev, const int32_t* ij_dev,
             const float* logits_prev_dev, const int32_t* forced_next_dev, float* state_out_dev,
             float* logits_out_dev, int32_t* chosen_ij_dev, float* top2_gap_dev, int32_t B, int32_t n, int32_t L,
             void* ws_dev, size_t ws_bytes, void* stream);

/* The whole eval+argmax branch of reinforce_rollout -- reference
 * finetune_rl_search.py:78-189 -- resident on the device with no host sync:
 * encode, then T-1 x (score new pairs, assemble table, argmax, merge).
 *   forced_merges_dev : NULL, or int32 [B,T-1,2] merges to apply instead of the
 *                       argmax (teacher forcing for parity tests)
 *   merges_out_dev    : int32 [B,T-1,2] the argmax pair of every step
 *   logits_trace_dev  : NULL, or float [B, sum_{n=T..2} P(n)] per-step tables
 *   top2_gap_dev      : NULL, or float [B,T-1]
 *   state_out_dev     : NULL, or float [B,T,C,D] encoder output */
int nnj_rollout_argmax(nnj_handle* h, const uint8_t* codes_dev, const uint8_t* mask_dev,
                       int32_t B, int32_t T, int32_t L,
                       const int32_t* forced_merges_dev, int32_t* merges_out_dev,
                       float* logits_trace_dev, float* top2_gap_dev, float* state_out_dev,
                       void* ws_dev, size_t ws_bytes, void* stream);

/* Sampling twin of nnj_rollout_argmax -- the eval, argmax=False branch of reinforce_rollout used by
 * RL_Search ("NeuralNJ-MC", reference finetune_rl_search.py:147, 338-427): at
 * every step the pair is drawn from Categorical(logits / temperature) by inverse CDF (fp64, flat pair
 * order) on the caller's uniforms_dev float [B,T-1] in [0,1) -- the reference's own RNG stream is not
 * reproducible across devices, so the stream is an input.  n_encode = B: codes/mask hold B alignments;
 * n_encode = 1: ONE alignment (codes [1,T,L], mask [1,L]) is encoded once and replicated to the B
 * rollouts (the reference re-encodes it per rollout), and the all-pairs scores of the first step --
 * the same numbers for every rollout -- are computed once.  merges_out [B,T-1,2] = the sampled pairs.
 * Tree likelihood scoring of the sampled trees (raxml-ng) is outside this library. */
int nnj_rollout_sample(nnj_handle* h, const uint8_t* codes_dev, const uint8_t* mask_dev,
                       int32_t B, int32_t T, int32_t L, int32_t n_encode,
                       const float* uniforms_dev, float temperature,
                       int32_t* merges_out_dev, float* logits_trace_dev,
                       void* ws_dev, size_t ws_bytes, void* stream);

/* Duplicate filter of the sampling mode -- the reference keeps one tree per `topo_repr` string (utils.py:76, the
 * rooted topology with children ordered by their smallest leaf).  keys_out uint64 [B]: a 64-bit key of the same
 * equivalence computed on the device from the merge lists int32 [B,T-1,2] (leaf = hash of its index, join = hash of
 * the commutative sum of its children): equal keys <=> equal topo_repr, up to 2^-64 collisions. */
int nnj_topology_hash(nnj_handle* h, const int32_t* merges_dev, int32_t B, int32_t T, uint64_t* keys_out_dev,
                      void* stream);

/* ---- Tree likelihood on the GPU: stands where the reference calls raxmlpy.optimize_brlen
 * (raxml-ng / libpll on the CPU: environment.py:365-441, 625-672, finetune_rl_search.py:401-411) to score sampled
 * trees.  That arithmetic lives in un-vendored third-party code with no expected values in the reference's tests:
 * PARITY UNPINNED.  These entry points follow the published algorithms (Felsenstein pruning; GTR, discrete-gamma rate
 * heterogeneity with the mean rate per category, proportion of invariant sites; Newton-Raphson branch lengths) and
 * are checked against closed forms and an independent numpy/scipy evaluation.  Model parameters are inputs: they are
 * NOT optimised here (the reference passes opt_model=True to raxml-ng). */
typedef struct nnj_subst_model {
  double rates[6];   /* GTR exchangeabilities AC AG AT CG CT GT (GT = 1 by convention)  */
  double freqs[4];   /* base frequencies A C G T (normalised by the library)            */
  double alpha;      /* gamma shape; <= 0: no rate heterogeneity                        */
  double pinv;       /* proportion of invariant sites, [0, 1)                           */
  int32_t ncat;      /* discrete gamma categories, 1..8 (4 = "+G" / "+G4")              */
} nnj_subst_model;

/* Scratch bytes for the two entry points below (n_align = 1: all B trees are over ONE alignment; B: one each). */
int nnj_lik_workspace_bytes(int32_t B, int32_t n_align, int32_t T, int32_t L, int32_t ncat, size_t* bytes);

/* log-likelihood of B trees given as merge lists int32 [B,T-1,2] (the output of the rollout entry points).
 * brlen_dev float [B,T-1,2]: length of the edge from each join to its two children, in merge order (NULL: 0.1
 * everywhere); the two child edges of the last join form one edge of the unrooted tree.  codes uint8 [n_align,T,L]
 * (gap / N / padding = every state possible), mask uint8 [n_align,L] or NULL (1 = site ignored).
 * loglik_out_dev double [B].  fp64 throughout. */
int nnj_tree_loglik(nnj_handle* h, const uint8_t* codes_dev, int32_t n_align, const uint8_t* mask_dev,
                    const int32_t* merges_dev, const float* brlen_dev, const nnj_subst_model* model_host,
                    int32_t B, int32_t T, int32_t L, double* loglik_out_dev, void* ws_dev, size_t ws_bytes, void* stream);

/* Probe of the substitution model (how the likelihood kernels are pinned to the data the reference holds): the model
 * the entry points build from `model_host` -- the rate matrix Q = U diag(lam) U^-1 normalised to one expected
 * substitution per unit time (Q_out, host, 16 doubles, row-major A C G T) and the ncat discrete-gamma category rates
 * (rates_out, host, ncat doubles, mean 1 over the gamma part) -- and the transition matrices P(rate_c * t) of ONE
 * branch of length t exactly as the device kernel forms them (P_out, host, ncat x 16 doubles).  Synchronous.
 * The reference's bundled test set keeps, for each of its 1,152 simulated alignments, the IQ-TREE log of the
 * simulation (data_gen/data/test/<len>/<taxa>/<name>_raw.tre.log) with the GTR parameters, the normalised Q and the
 * category rates: tests/golden/iqtree_models.npz, checked by tests/test_likelihood.py through this entry point. */
int nnj_lik_model_probe(nnj_handle* h, const nnj_subst_model* model_host, double t, double* Q_out, double* rates_out,
                        double* P_out);

/* Branch-length optimisation, then the log-likelihood: `sweeps` rounds (the reference asks raxml-ng for iters=3) of
 * one Newton-Raphson solve per edge, all edges of all trees at once from the same partial likelihoods, the step per
 * tree halved until the likelihood does not drop.  brlen_out_dev float [B,T-1,2] or NULL. */
int nnj_tree_optimize(nnj_handle* h, const uint8_t* codes_dev, int32_t n_align, const uint8_t* mask_dev,
                      const int32_t* merges_dev, const float* brlen_in_dev, const nnj_subst_model* model_host,
                      int32_t sweeps, int32_t B, int32_t T, int32_t L, float* brlen_out_dev, double* loglik_out_dev,
                      void* ws_dev, size_t ws_bytes, void* stream);

/* Concurrency of the rollout entry points.  The alignments of a batch are independent; with streams = k (1..4,
 * default 2) a rollout of B >= 64 alignments is cut into k contiguous sub-batches that run on k streams owned by the
 * handle, forked from and joined to the caller's `stream` with events: the call is still asynchronous on `stream` and
 * results are identical to those of the sub-batches run alone (bit for bit; launch geometry depends on the
 * sub-batch size).  The reference has no counterpart (one device, one stream).  streams = 1 keeps every launch on the
 * caller's stream. */
int nnj_set_concurrency(nnj_handle* h, int32_t streams);

/* Numeric guard.  The GEMMs of this library split fp32 operands into two fp16 pieces: operand
 * magnitudes beyond 65504 overflow to infinity instead of being rounded as fp32 would.  Every pair-score table
 * the library writes is scanned by the kernel that writes it; a non-finite entry sets a sticky per-handle flag.
 * nnj_numeric_status synchronises `stream`, returns the flag in *nonfinite_out (1 = some table since the last
 * call held a non-finite score: results must be discarded) and clears it.  The reference has no counterpart
 * (it computes in plain fp32); callers that fetch results to the host should call this once per batch. */
int nnj_numeric_status(nnj_handle* h, int32_t* nonfinite_out, void* stream);
/* Bits of the word nnj_numeric_status returns (0 = nothing to report).  NNJ_STATUS_BARRIER_TIMEOUT: kernels in
 * which two or more waves of a workgroup build one LDS image meet at LDS-counter barriers with a bounded spin; a
 * wave that gave up waiting (it never happens on a healthy device) computed on an incomplete image, so the results
 * since the last call must be discarded exactly like non-finite ones.  NNJ_STATUS_MERGE_WEIGHTS: in a rollout the
 * attention weights of a merge come from the scorer's own logits of the picked pair (two-pass NJ step);
 * a pick for which neither source applied while the fallback pass was not scheduled is an internal error of the
 * library, reported here instead of returning a tree built on zero weights.  NNJ_STATUS_BAD_MERGE: a merge list handed
 * to nnj_tree_loglik / nnj_tree_optimize held a pair with i >= j or a position outside the live list (the lists live in
 * device memory, so they are validated by the kernel that reads them): the likelihoods of that call must be discarded. */
#define NNJ_STATUS_NONFINITE 1
#define NNJ_STATUS_BARRIER_TIMEOUT 2
#define NNJ_STATUS_MERGE_WEIGHTS 4
#define NNJ_STATUS_BAD_MERGE 8

/* Kernel timing for bench.py's roofline object: when enabled, every kernel launch of
 * the entry points is bracketed by a HIP event pair on the launch stream, tagged with
 * its kernel kind.  After the caller has synchronised the stream, nnj_profile_read
 * returns accumulated milliseconds and launch counts per kind (arrays of
 * nnj_profile_kinds() entries, `cap` = their length) and resets the counters. */
int nnj_profile_enable(nnj_handle* h, int32_t on);
int nnj_profile_kinds(void);
const char* nnj_profile_kind_name(int32_t kind);
int nnj_profile_read(nnj_handle* h, double* ms_out, int64_t* launches_out, int32_t cap);
/* Launches since nnj_profile_enable that could NOT be bracketed (the event pool grows on demand; this counts
 * failed event creations).  Non-zero means the per-kind times under-report: bench.py refuses to print them. */
int nnj_profile_dropped(nnj_handle* h, int64_t* dropped_out);

/* Test tap: make nnj_encode stop inside layer 0 (1 = after the row-attention block,
 * 2 = after the column-attention block, 0 = run everything). */
int nnj_debug_encoder_stop(nnj_handle* h, int32_t stage);

#ifdef __cplusplus
}
#endif
#endif /* NNJ_H */
