/*
 * cppf_hip_experimental.h -- the EXPERIMENTAL part of libcppf_hip.so's C ABI (see cppf_hip.h for the stable part and the
 * conventions, which hold here too).  Nothing in this file is bound by the reference-facing modules' default paths; entries
 * may change or disappear without an ABI bump.  What is here and why it is kept:
 *   - launch forms superseded on the default paths but kept for A/B measurements (bench.py --separate-encode,
 *     --materialize-tuples, --mlp-arith native|split16) and as the two-kernel references the fused forms are tested against;
 *   - helpers of BASELINE configs[4]'s extensions (float16 feature table) that the reference does not have;
 *   - the logit prior of the synthetic benchmarks: THE REFERENCE HAS NO PRIOR (eval.py:225-235 draws from the network's own
 *     logits).  It exists because the reference's checkpoints are not available: random-init weights need a teacher for votes to
 *     cluster.  bench.py reports value_no_prior beside the headline;
 *   - a test hook.
 */
#ifndef CPPF_HIP_EXPERIMENTAL_H_
#define CPPF_HIP_EXPERIMENTAL_H_

#include "cppf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- the synthetic benchmarks' logit prior ------------------------------------------------------------------------------
 * cppf_decode_bins with logit_prior float32 [T, 6, nb] (same shape as logits; NULL = cppf_decode_bins) added to the logits before
 * the softmax with one float32 add each. */
int cppf_decode_bins_prior(int B, const float* logits, const float* logit_prior, int nb, const float* uniforms,
                           const float* pts, const int32_t* idx, int k, const int32_t* pt_off, const int32_t* tup_off,
                           int64_t total_tuples, const double* h_axes, int32_t* bins, float* scaled, float* scale,
                           float* tr, float* rot, void* stream);
/* cppf_reslayer_split_decode with a prior: logit_prior float32 [rows, 192] or NULL; or prior_pos float32 [rows, 6] /
 * prior_inv_sigma (instead of logit_prior, never both): the prior GENERATED in the epilogue -- prior[t, c, k] =
 * -0.5 ((k - prior_pos[t, c]) * prior_inv_sigma)^2, float32, these three operations in this order -- a Gaussian bump in logit
 * space around a per-coordinate bin position: 24 bytes per row read instead of 768 (the array form costs the launch 0.34 ms per
 * 1.28 M rows, all of it its 983 MB of reads).  Bit-identical to passing the array built with the same three operations. */
int cppf_reslayer_split_decode_prior(const float* x, int64_t ldx, int32_t k_in, int64_t rows, const void* wq, int64_t wq_bytes,
                                     const float* b1, const float* b0, const float* logit_prior, const float* prior_pos,
                                     float prior_inv_sigma, const float* uniforms, int32_t* bins, int32_t* sched, void* stream);

/* ---- a2 / a3 variants -------------------------------------------------------------------------------------------------- */
/* The descriptor half alone, on normals the caller already has (e.g. from cppf_estimate_normals). */
int cppf_shot352_from_normals(int B, const float* pts, const int32_t* pt_off, int64_t total_points,
                              const float* normals, float shot_r, float* out_shot, float* out_rf,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* out_half[i] = (float16) x[i], n elements (round to nearest even): the float16 feature table of BASELINE config 5 from the
 * point encoder's float32 output (cppf_encode_tuples_shot_f16 gathers it). */
int cppf_cast_f16(const float* x, void* out_half, int64_t n, void* stream);

/* Same row layout from a half-precision feature table (float16 [n, feat_dim], feat_dim % 8 == 0), float32 output
 * (BASELINE config 5, "fp16 features"; not in the reference -- equals the float32 entry on half-rounded features). */
int cppf_encode_tuples_shot_f16(int B, const float* pts, const float* normals, const void* feat_half, int feat_dim,
                                const int32_t* idx, int k, const int32_t* pt_off, const int32_t* tup_off,
                                int64_t total_tuples, float* out, void* stream);

/* a3' descriptor part (DINO model): replaces desc_pair_transform(cat_i desc_transform(desc[idx_i])) (train_dino.py:95-96)
 * by a gather-add of per-point products: tables float32[total_points, k, D] with tables[n][i] = W_i . d[n] (W_i = the i-th
 * column block of desc_pair_transform.weight, d = desc_transform(desc)), bias float32[D] or NULL;
 * out[t, out_col + c] = bias[c] + sum_i tables[pt_off[b] + idx[t,i]][i][c].  D % 4 == 0; out row stride in floats. */
int cppf_encode_tuples_dino(int B, const float* tables, int k, int D, const float* bias, const int32_t* idx,
                            const int32_t* pt_off, const int32_t* tup_off, int64_t total_tuples, float* out,
                            int out_stride, int out_col, void* stream);

/* cppf_kept_rows32 as int64[B, max_kept] (device): row = tup_off[b] + kept_tuple[tup_off[b] + j] for
 * j < kept_count[b], the scene's first tuple row otherwise -- the fixed-shape index a caller gathers per-pair tensors with
 * (e.g. the tuple features the scale head runs on, eval.py:272) without reading kept_count on the host. */
int cppf_kept_rows(int B, const int32_t* tup_off, const int32_t* kept_tuple, const int32_t* kept_count, int max_kept,
                   int64_t* rows, void* stream);

/* ---- superseded launch forms of the MLP ------------------------------------------------------------------------------- */
/* ---- 128-wide residual layer of the tuple / point encoders (train_shot.py:19-45 ResLayer with dim_in == dim_out == 128,
 * bn = dropout = False; five of the six layers of `tuple_encoder` and `shot_encoder`):
 *     x <- x + relu(x W1^T + b1) W2^T        in place on x float32[rows,128] (device, contiguous)
 * w1, w2: float32[128,128] row-major [out,in] (the nn.Linear weights), b1 float32[128]; fc2's bias is left to the caller
 * (it commutes with the residual stream).  One kernel on the f32 matrix cores: both products of a 32-row tile stay in
 * registers.  float32 in, float32 accumulate: results differ from a library GEMM's only by summation order. */
int cppf_reslayer128(float* x, int64_t rows, const float* w1, const float* b1, const float* w2, void* stream);

/* ---- the tuple encode feeding the tuple MLP without materialising its rows (train_shot.py:75-83 -> :100-111): the pair
 * features alone and the tuples' global point indices ...
 *   heads float32 [T, ld_heads >= 4 C(k,2)]: the first 4 C(k,2) columns of cppf_encode_tuples_shot's rows, bit for bit;
 *   gidx  int32 [T, k] = scene point base + idx
 * ... and the first ResLayer of the tuple encoder (a 128-wide projection layer, `chain` identity layers behind it) reading
 * row t = [heads[t, 0:head_cols] | table[gidx[t, 0]] | ... | table[gidx[t, slots-1]]] through its x-tile fetches: table
 * float32 [points, fdim] (the point encoder's output), fdim a power of two >= 8, head_cols % 8 == 0, slots <= 8; wq / b1 /
 * b0 / chain as cppf_reslayer_split for k_in = head_cols + slots * fdim.  Bit-identical to cppf_reslayer_split on the
 * materialised rows; saves writing and re-reading them (1.8 GB per 64 scenes). */
int cppf_encode_tuples_shot_heads(int B, const float* pts, const float* normals, const int32_t* idx, int k,
                                  const int32_t* pt_off, const int32_t* tup_off, int64_t total_tuples, float* heads,
                                  int32_t ld_heads, int32_t* gidx, void* stream);
int cppf_reslayer_split_gather(const float* heads, int64_t ld_heads, int32_t head_cols, const int32_t* gidx, int32_t slots,
                               const float* table, int32_t fdim, float* out, int64_t ldo, int32_t n_out, int64_t rows,
                               const void* wq, int64_t wq_bytes, const float* b1, const float* b0, int32_t chain,
                               int32_t* sched, void* stream);

/* The DINO model's two-kernel form of cppf_reslayer_split_sumencode:
 * cppf_encode_tuples_coord_heads: heads float32 [T, ld_heads >= round8(3 C(k,2))] = the coordinate columns of
 *   cppf_encode_tuples_coord's rows (train_dino.py:92), bit for bit, zero-padded to a multiple of 8 columns; gidx int32 [T, k] =
 *   scene point base + idx.
 * cppf_reslayer_split_sumgather: the first ResLayer on rows [heads | s(t)] with the per-point parts of its first products summed
 *   from tables float32 [points, slots, 256] (see cppf_reslayer_split_sumencode); also usable for the SHOT model (s(t) = the
 *   concatenated point features, train_shot.py:82: a re-slicing, no fold -- measured slower than the x-tile gather there).
 *   wq = cppf_reslayer_split_stream_bytes(head_cols, 128, 1, chain) bytes. */
int cppf_encode_tuples_coord_heads(int B, const float* pts, const int32_t* idx, int k, const int32_t* pt_off,
                                   const int32_t* tup_off, int64_t total_tuples, float* heads, int32_t ld_heads, int32_t* gidx,
                                   void* stream);
int cppf_reslayer_split_sumgather(const float* heads, int64_t ld_heads, int32_t head_cols, const int32_t* gidx, int32_t slots,
                                  const float* tables, int64_t ld_tables, float* out, int64_t ldo, int32_t n_out, int64_t rows,
                                  const void* wq, int64_t wq_bytes, const float* b1, const float* b0, int32_t chain,
                                  int32_t* sched, void* stream);

/* ---- the same ResLayer launches in f16x2 arithmetic (cppf2_amd.models.MLP_ARITH = "split16"; not the default): every
 * float32 operand as an fp16 pair hi + lo (RNE; 22-23 significant bits), the three products hi hi + hi lo + lo hi on the fp16
 * matrix cores, float32 accumulate -- half the matrix-core work of the exact bf16-triple form, error against float64 at the
 * level of a float32 GEMM's accumulation error (tests/test_mlp_split.py, bench.py mlp_error_vs_f64), but NOT exact products, and
 * fp16's range: activations must stay below 65504 in magnitude (larger ones become Inf -> NaN rows); activations whose lo piece
 * is subnormal keep an absolute resolution of 2^-25.  wq = fp16 (hi, lo) fragment pairs of weight_scale x the weights in the
 * fragment order of cppf_reslayer_split (cppf_reslayer_split16_stream_bytes bytes), b1 / b0 = weight_scale x the biases,
 * weight_scale a power of two chosen so that the largest |weight| x weight_scale is ~2^13.  One struct for all launch forms:
 * plain (x, out[, first_out = tap]); gather (gidx != NULL: x = heads with k_in head columns, table, slots, fdim; n_out = 128);
 * decode (uniforms != NULL: n_out = 192, chain = 0, bins out, logit_prior optional).  Unused members must be zero. */
typedef struct CppfReslayerSplit16Args {
  const float* x; int64_t ldx; int32_t k_in;
  float* out; int64_t ldo; int32_t n_out; int64_t rows;
  const void* wq; int64_t wq_bytes; const float* b1; const float* b0; int32_t chain;
  float weight_scale;
  float* first_out; int64_t ld_first;
  const int32_t* gidx; int32_t slots; const float* table; int32_t fdim;
  const float* logit_prior; const float* uniforms; int32_t* bins;
  void* stream;
  int32_t mode;            /* 0: the forms above; 1: plain Linear (cppf_linear_split: n_out % 256 == 0, b1 = bias or NULL, b0 NULL);
                              2: gather with per-point slot tables summed into the accumulators (cppf_reslayer_split_sumgather:
                              table = [points, slots, 256], ld_table its point pitch, fdim unused) */
  int64_t ld_table;
  int32_t* sched;          /* NULL or the block-scheduling counters (see cppf_reslayer_split; not used by mode 1) */
  const float* prior_pos;  /* decode: the generated prior of cppf_reslayer_split_decode (instead of logit_prior) or NULL */
  float prior_inv_sigma;
} CppfReslayerSplit16Args;
int64_t cppf_reslayer_split16_stream_bytes(int32_t k_in, int32_t n_out, int32_t proj, int32_t chain);
int cppf_reslayer_split16(const CppfReslayerSplit16Args* args);

/* Test hook: workgroups > 0 forces the number of persistent workgroups of every later cppf_reslayer_split* launch of this
 * process (the kernels' results do not depend on it; tests/test_mlp_split.py runs the counted-wait protocol at 1, 7 and all
 * CUs beside a saturating copy stream); 0 restores one workgroup per CU. */
int cppf_reslayer_split_debug_grid(int32_t workgroups);

#ifdef __cplusplus
}
#endif
#endif /* CPPF_HIP_EXPERIMENTAL_H_ */
