/* nnj_train.h -- C ABI of libnnj_train_hip.so: the differentiable operators of NeuralNJ's Finetune mode.
 *
 * The reference fine-tunes with torch.autograd straight through model.py / msa_modules.py / axial_attention.py
 * (reference finetune_rl_search.py:192-335: loss.backward() on the REINFORCE + entropy loss of a sampled rollout).  Here
 * every arithmetic operator of that forward pass has a hand-written gfx950 forward AND backward kernel; the graph
 * (which operator feeds which, gradient accumulation into shared rows) is kept by torch.autograd.Function objects in
 * neuralnj_amd/train_ops.py, one per operator, that call these entry points on raw device pointers.  All tensors are
 * contiguous fp32 on the device unless a stride is spelled out; `stream` is a hipStream_t.  Every function returns 0 or
 * a negative error code (nnjt_last_error() says why); nothing synchronises the stream.
 *
 * This is a first, UNFUSED path: fp32 arithmetic (the contractions on v_mfma_f32_32x32x2_f32), one kernel per operator,
 * activations kept by the caller.  It is separate from libnnj_hip.so (inference), which it neither links
 * nor changes.
 */
#ifndef NNJ_TRAIN_H
#define NNJ_TRAIN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int nnjt_abi_version(void);
const char* nnjt_last_error(void);

/* C[b1,b2][m,n] = alpha * sum_k A[b1,b2][m,k] * B[b1,b2][k,n] + bias[n] + beta * C[b1,b2][m,n]
 * (beta = 0: C is not read; bias may be NULL -- the bias of nn.Linear, added in the epilogue instead of a second pass)
 * Every operand is addressed by element strides: X[b1,b2][r,c] = X + b1*sXb1 + b2*sXb2 + r*sXr + c*sXc.
 * This one kernel is every contraction of the model: nn.Linear forward (reference msa_modules.py / model.py) and its
 * two backward products, the tied row-attention einsums (axial_attention.py:97,114), the column-attention einsums
 * (axial_attention.py:216,234) and the pair scorer's attention over rows (model.py:118,148). */
typedef struct nnjt_gemm {
  const float* A; const float* B; float* C;
  int32_t M, N, K, nb1, nb2;
  int64_t sAm, sAk, sAb1, sAb2;
  int64_t sBk, sBn, sBb1, sBb2;
  int64_t sCm, sCn, sCb1, sCb2;
  float alpha, beta;
  const float* bias;
} nnjt_gemm;
int nnjt_gemm_run(const nnjt_gemm* g, void* stream);

/* C[b][m, n] = alpha * sum_k A[b][m, k] * B[b][k, n] for 1 <= M, K <= 64 and a long N (a multiple of 64): the pair scorer's
 * x_g = alpha_rows x state over sites x features (reference model.py:148) and its gradient with respect to the state.
 * A[b][m, k] = A + b * bsA + m * sAm + k * sAk (either orientation); B[b] is [K, N] and C[b] is [M, N], both contiguous. */
int nnjt_skinny_gemm(const float* A, int64_t sAm, int64_t sAk, int64_t bsA, const float* B, int64_t bsB, float* C,
                     int64_t bsC, int32_t nb, int32_t M, int32_t K, int64_t N, float alpha, void* stream);

/* Weight gradient of a 64 -> 64 nn.Linear over `rows` tokens: parts[p][m * 64 + n] = sum over the tokens of block p of
 * dy[token][m] * x[token][n]  (dy, x: [rows, 64] contiguous; block p = tokens [p * per_block, (p + 1) * per_block),
 * per_block a multiple of 8; parts: [ceil(rows / per_block), 4096]).  nnjt_sum_rows over the parts gives dW (the
 * weight gradients of nn.Linear, reference msa_modules.py / model.py under torch.autograd). */
int nnjt_wgrad64(const float* dy, const float* x, float* parts, int64_t rows, int64_t per_block, void* stream);

/* y[r, :] += bias (rows x cols, in place) -- the bias of nn.Linear; colsum: out[c] += sum_r x[r, c] (its gradient). */
int nnjt_add_bias(float* y, const float* bias, int64_t rows, int32_t cols, void* stream);
int nnjt_colsum(const float* x, float* out, int64_t rows, int32_t cols, void* stream);

/* out[c] = sum_r x[r, c] (rows x cols, any number of columns; a fixed order of additions): the second step of a contraction
 * that was cut into pieces along k (weight gradients over millions of tokens, attention logits over sites x features). */
int nnjt_sum_rows(const float* x, float* out, int64_t rows, int64_t cols, void* stream);

/* nn.LayerNorm(64), eps 1e-5 (reference msa_modules.py:107): y = (x - mean) * rstd * gamma + beta over the last dim.
 * Forward keeps mean / rstd per row for the backward; dgamma / dbeta are ACCUMULATED (+=). */
int nnjt_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                       int64_t rows, int32_t cols, void* stream);
int nnjt_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                       float* dx, float* dgamma, float* dbeta, int64_t rows, int32_t cols, void* stream);

/* nn.GELU() (exact erf form; reference model.py:41,58, msa_modules.py:140). */
int nnjt_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int nnjt_gelu_bwd(const float* dy, const float* x, float* dx, int64_t n, void* stream);

/* The sigmoid gate of PhyloATTN.aggregate (reference model.py:105-108 and 150-153):
 *   out = z * a + (1 - z) * b,  z = sigmoid(h)      (first gate: h = W_h(x_i - x_j), a = x_i, b = x_j;
 *                                                    second: h = W_g x_g, a = x_g, b = x)
 * Backward writes dh, da, db (no accumulation). */
int nnjt_gate_fwd(const float* h, const float* a, const float* b, float* out, int64_t n, void* stream);
int nnjt_gate_bwd(const float* dout, const float* h, const float* a, const float* b, float* dh, float* da, float* db,
                  int64_t n, void* stream);

/* softmax over the last dim of [rows, cols]; entries whose keep[r*cols + c] == 0 (keep may be NULL) get probability 0
 * (the "+= -inf" of model.py:128-144).  Backward: dx = y * (dy - sum_c y * dy). */
int nnjt_softmax_fwd(const float* x, const uint8_t* keep, float* y, int64_t rows, int32_t cols, void* stream);
int nnjt_softmax_bwd(const float* dy, const float* y, float* dx, int64_t rows, int32_t cols, void* stream);

/* out = a * x + b * y (elementwise; residual adds, scalings, differences). out may alias x or y. */
int nnjt_axpby(float a, const float* x, float b, const float* y, float* out, int64_t n, void* stream);

/* out[r, :] = x[r, :] * s[r]  -- row scaling (zeroing q at padded sites, axial_attention.py:78-82; masked site sums). */
int nnjt_rowscale(const float* x, const float* s, float* out, int64_t rows, int32_t cols, void* stream);

/* nn.Dropout(p) in training mode (reference msa_modules.py:104-119 on every sublayer's output, :141-149 after the
 * feed-forward GELU, axial_attention.py:56,136,233 on the attention probabilities; p = 0.4, model.py:23):
 *   keep[i] = 1 with probability 1 - p, y[i] = x[i] * keep[i] / (1 - p);  backward dx = dy * keep / (1 - p).
 * The draw of element i is a pure function of (seed, offset + i) -- a counter-based generator, no device state: the
 * caller advances `offset` by n per call.  The stream of torch's own generator cannot be reproduced, so parity with the
 * reference in train() mode is distributional; given the same masks the arithmetic is the reference's. */
int nnjt_dropout_fwd(const float* x, float* y, uint8_t* keep, int64_t n, float p, uint64_t seed, uint64_t offset,
                     void* stream);
int nnjt_dropout_bwd(const float* dy, const uint8_t* keep, float* dx, int64_t n, float p, void* stream);

/* x[r, c] = value where sel[r_map(r), c] != 0: the key-padding fill of the attention logits
 * (axial_attention.py:99-103, 220-224): x is [outer, inner, cols], sel is [outer?, cols] picked by
 * sel_row = (r / inner) % sel_rows.  The backward of masked_fill zeroes the gradient there: call with value = 0. */
int nnjt_fill_where(float* x, const uint8_t* sel, float value, int64_t rows, int32_t inner, int32_t sel_rows,
                    int32_t cols, void* stream);

/* out[b, p, :] = src[b, idx[b, p], :]   (rows of `width` floats; src [B, n, width], idx int64 [B, p]) and its
 * gradient dsrc[b, idx[b,p], :] += dout[b, p, :] (atomic adds).  The pair gathers of decode_zxr (model.py:173-197),
 * and with width = 1 the table gather of model.py:201. */
int nnjt_gather_rows(const float* src, const int64_t* idx, float* out, int32_t B, int32_t n, int32_t p, int64_t width,
                     void* stream);
int nnjt_scatter_rows_add(const float* dout, const int64_t* idx, float* dsrc, int32_t B, int32_t n, int32_t p,
                          int64_t width, void* stream);

/* out (contiguous, dims d[0..4]) [i0,i1,i2,i3,i4] = in[sum_k i_k * stride[k]]; inverse = 1: the transposed copy
 * in[sum_k i_k * stride[k]] = out[...] (the gradient of a permutation).  einops.rearrange of the reference. */
int nnjt_permute5(const float* in, float* out, const int64_t* dims, const int64_t* strides, int32_t inverse, void* stream);

#ifdef __cplusplus
}
#endif
#endif
