/*
 * cppf_hip.h -- the STABLE C ABI of libcppf_hip.so, the MI355X (gfx950) implementation of the
 * CPPF++ (qq456cvb/CPPF2) point-pair-feature voting hot path.
 *
 * ABI 11 (round 6) splits the interface in two.  THIS file is what a maintainer of the reference binds: the entry points the
 * drop-in modules (eval.py, train_*.py, dataset.py, utils/util.py, src_shot/build/shot.py), INTEGRATION.md and the default bench
 * call -- one per stage of the path, each with exactly the reference's semantics -- plus the workspace-size
 * queries.  Everything else the library exports -- launch forms kept for A/B measurements, the f16x2 arithmetic, the float16
 * table and weighted-vote extensions' helpers, the synthetic benchmarks' logit prior, test hooks -- is declared in
 * cppf_hip_experimental.h and may change without an ABI bump.  tests/test_abi.py holds both files to the ctypes tables of
 * cppf2_amd/_lib.py symbol by symbol; tests/test_zz_stable_abi_coverage_gpu.py asserts that a GPU test session calls every
 * stable symbol through ctypes.
 *
 * The reference has no FFI of its own for this path except one pybind11 module
 * (src_shot/shot.cpp:164-168); everything else is Python glue over torch ops.  Each entry
 * point below names the reference code it replaces (file:line under the reference root).
 * The Python binding (ctypes) lives in cppf2_amd/_lib.py; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *  - every function returns 0 on success or a negative CPPF_E* code; no C++ exception
 *    crosses the boundary; cppf_last_error_string() describes the last failure of the
 *    calling thread.
 *  - all pointers are DEVICE pointers unless the name starts with h_ (host).  The caller
 *    owns every buffer (inputs, outputs, workspace); the library allocates nothing.
 *  - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*); functions
 *    are re-entrant.  The only process-wide state is a per-device cache written once under a mutex on
 *    first use (CU count, the vote kernels' dynamic-LDS attribute); calls keep their state in the caller's
 *    workspace (cppf_shot_prepare -> cppf_shot_describe share one: same workspace, same stream, in that order).
 *  - batch-first: B independent scenes per call.  Ragged scenes are described by device
 *    offset arrays pt_off[B+1] (points) and tup_off[B+1] (tuples), int32, plus host-side
 *    maxima used only to size launches.
 *  - point indices are int32; tuples are rows of `k` indices (k = 2 + num_more, <= 8).
 */
#ifndef CPPF_HIP_H_
#define CPPF_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPPF_OK            0
#define CPPF_EINVAL       -1   /* bad argument (null pointer, size out of range) */
#define CPPF_EUNSUPPORTED -2   /* valid request this build does not implement */
#define CPPF_EHIP         -3   /* a HIP runtime call failed (see cppf_last_error_string) */
#define CPPF_ECAPACITY    -4   /* a caller-provided capacity is too small */

#define CPPF_ABI_VERSION 11

/* Per-scene voxel grid geometry: train_dino.py:172-173 (corners, grid_res). 32 bytes. */
typedef struct CppfSceneGrid {
  float   c0[3];    /* pc.min(0) */
  int32_t g[3];     /* ((max-min)/res).long() + 1 */
  int32_t ncell;    /* g[0]*g[1]*g[2], 0 if empty / overflow */
  int32_t flags;    /* bit0: empty scene; bit1: cell count overflows int32 */
} CppfSceneGrid;

/* Per-scene result record gathered across ranks (one RCCL all_gather). 160 bytes. */
typedef struct CppfSceneResult {
  int64_t  argmax;       /* flat index of the first maximum of the centre grid (C order) */
  double   t[3];         /* cand_world = c0 + cell*res, float64 like train_dino.py:213 */
  double   R[9];         /* row-major R_est, eval.py:298-313 */
  float    scale[3];     /* lower median of pred_scales over kept pairs, eval.py:309 */
  uint32_t peak;         /* vote count at argmax */
  int32_t  up_idx;       /* top-1 sphere bin of the "up" vote, eval.py:284 */
  int32_t  right_idx;    /* top-1 sphere bin of the "right" vote, eval.py:292 */
  int32_t  kept;         /* pairs surviving the back-vote filter, eval.py:258 */
  float    up_count;     /* counts[up_idx] */
  float    right_count;  /* counts[right_idx] */
  int32_t  flags;        /* CppfSceneGrid.flags | (bit2: grid larger than cells_cap) */
  int32_t  ncell;        /* cells of the scene's vote grid */
  int32_t  pad_[3];
} CppfSceneResult;

int         cppf_version(void);
const char* cppf_last_error_string(void);

/* ---- a1. tuple sampler: replaces np.random.randint(0, N, (T, k)) at eval.py:207,
 * train_shot.py:88, train_dino.py:103.  Philox4x32-10, counter (tuple, word-block, scene id, 0),
 * key = seed; scene id of scene b = scene_id_base + b*scene_id_stride, so the table does not
 * depend on how scenes are sharded over ranks.  out_idx: int32[tup_off[B], k]. */
int cppf_sample_tuples(int B, const int32_t* pt_off, const int32_t* tup_off, int max_t, int k,
                       uint64_t seed, int32_t scene_id_base, int32_t scene_id_stride,
                       int32_t* out_idx, void* stream);

/* Uniforms in [0,1) (24-bit) from the same generator, counter (row, word-block, scene id, stream_id):
 * stands in for the unseeded torch.multinomial draw of eval.py:229.  out: float32[tup_off[B], m], m <= 8. */
int cppf_philox_uniform(int B, const int32_t* tup_off, int max_t, int m, uint64_t seed,
                        int32_t scene_id_base, int32_t scene_id_stride, int32_t stream_id,
                        float* out, void* stream);

/* ---- a2. SHOT352 + normals: replaces shot.compute(pc, normal_r, shot_r) (src_shot/shot.cpp:45-100,
 * PCL 1.9.1 NormalEstimation + SHOTEstimation).  out_shot: float32[n,352], out_normal: float32[n,3];
 * rows PCL would leave NaN are written as NaN (callers zero them, eval.py:215-216).
 * workspace: cppf_shot352_workspace_bytes(B, total_points) bytes.
 * flags (every entry point that estimates normals): 0 = pcl::NormalEstimation's arithmetic (src_shot/shot.cpp:25-32, 66-72):
 * single-pass float32 sums of the raw coordinates and their products over the neighbours in (distance, index) order,
 * covariance = E[x x^T] - E[x] E[x]^T, closed-form pcl::eigen33, float32 flipNormalTowardsViewpoint -- what the reference's
 * trained checkpoints saw; CPPF_SHOT_F64_NORMALS = float64 covariance about the query point + Jacobi (the more accurate
 * normals of rounds 1-3: 0.03 deg median / 0.45 deg max away from PCL's arithmetic on the bench clouds).  The local frames and
 * the descriptor follow PCL's algorithm in both (float64 weighted covariance, SelfAdjointEigenSolver restated as Jacobi). */
#define CPPF_SHOT_F64_NORMALS 1
int64_t cppf_shot352_workspace_bytes(int B, int64_t total_points);
int cppf_shot352(int B, const float* pts, const int32_t* pt_off, int64_t total_points, float normal_r, float shot_r,
                 float* out_shot, float* out_normal, float* out_rf /* optional float32[n,9] local frames */,
                 void* workspace, int64_t workspace_bytes, int flags, void* stream);

/* shot.compute_color (src_shot/shot.cpp:102-161 -> pcl::SHOTColorEstimation, SHOT1344): colors float32[total_points,3] in
 * [0,1] (stored as uint8 = c * 255 truncated, like the wrapper), out_shot float32[total_points,1344] = 32 sectors x 11
 * shape slots followed by 32 sectors x 31 colour slots, L2-normalised over all 1344 entries; NaN rows as cppf_shot352.
 * Exported by the reference's module but called by none of its scripts; parity unpinned like cppf_shot352. */
int64_t cppf_shot1344_workspace_bytes(int B, int64_t total_points);
int cppf_shot1344(int B, const float* pts, const float* colors, const int32_t* pt_off, int64_t total_points,
                  float normal_r, float shot_r, float* out_shot, float* out_normal,
                  void* workspace, int64_t workspace_bytes, int flags, void* stream);
/* Two-call form of cppf_shot352 sharing one workspace: prepare = cell sort + covariances + eigen-solves (normals out,
 * local-frame axes kept in the workspace); describe = the histogram kernel alone.  describe must follow a prepare
 * on the same inputs / stream / workspace.  `normals` are the ones the descriptor reads (eval.py zeroes NaNs only
 * after shot.compute, so pass prepare's output unchanged).  nan_to_zero != 0 makes describe write 0 where shot.compute
 * writes NaN (invalid frames, < 5 neighbours): the np.nan_to_num of eval.py:215 folded into the store. */
int cppf_shot_prepare(int B, const float* pts, const int32_t* pt_off, int64_t total_points, float normal_r,
                      float shot_r, float* out_normal, void* workspace, int64_t workspace_bytes, int flags, void* stream);
int cppf_shot_describe(int B, const float* pts, const int32_t* pt_off, int64_t total_points, const float* normals,
                       float shot_r, int nan_to_zero, float* out_shot, float* out_rf, void* workspace,
                       int64_t workspace_bytes, void* stream);
/* x[isnan(x)] = 0 in place, n floats: eval.py:216 on the normals (after the descriptor has read them). */
int cppf_nan_to_zero(float* x, int64_t n, void* stream);
/* estimate_normal(pc, normal_r) (src_shot/shot.cpp:12-42).  Same workspace size as cppf_shot352. */
int cppf_estimate_normals(int B, const float* pts, const int32_t* pt_off, int64_t total_points, float normal_r,
                          float* out_normal, void* workspace, int64_t workspace_bytes, int flags, void* stream);

/* ---- a3. tuple encode: replaces BeyondCPPF.prepare_tuple_inputs.
 * SHOT model (train_shot.py:75-83): row = [p_i-p_j for i<j (C(k,2)*3) | max(n_i.n_j, -n_i.n_j) (C(k,2)) |
 *   feat[idx_0..k-1] (k*feat_dim)]; feat is the per-point output of shot_encoder, float32[n,feat_dim],
 *   feat_dim % 4 == 0.  out: float32[tup_off[B], C(k,2)*4 + k*feat_dim].
 * DINO model (train_dino.py:92): only the coordinate part, written at the start of rows of
 *   `out_stride` floats (the descriptor part: cppf_linear_split + cppf_reslayer_split_sumencode below). */
int cppf_encode_tuples_shot(int B, const float* pts, const float* normals, const float* feat, int feat_dim,
                            const int32_t* idx, int k, const int32_t* pt_off, const int32_t* tup_off,
                            int64_t total_tuples, float* out, void* stream);
int cppf_encode_tuples_coord(int B, const float* pts, const int32_t* idx, int k, const int32_t* pt_off,
                             const int32_t* tup_off, int64_t total_tuples, float* out, int out_stride,
                             void* stream);

/* ---- a4+a5. bin decode + vote parameters: replaces eval.py:225-240 (softmax, one multinomial draw per
 * (tuple, coord) by inverse CDF from `uniforms`, bin/(nb-1)-0.5, per-pair metric scale) fused with
 * generate_target_pairs(pred_pairs_scaled, a1, a2, a3) (dataset.py:118-135, centre 0).
 * logits: float32[T,6,nb] (nb <= 64); uniforms: float32[T,6]; h_axes: 9 doubles = the three axis vectors
 * in the positional order handed to generate_target_pairs (eval.py passes up, front, right).
 * Outputs (any may be NULL): bins int32[T,6], scaled float32[T,6] (pred_pairs_scaled), scale float32[T],
 * tr float32[T,2] (proj_len, dist2o), rot float32[T,3].
 * (The reference has no logit prior; the synthetic benchmarks' stand-in for trained weights is cppf_decode_bins_prior in
 * cppf_hip_experimental.h.) */
int cppf_decode_bins(int B, const float* logits, int nb, const float* uniforms, const float* pts,
                     const int32_t* idx, int k, const int32_t* pt_off, const int32_t* tup_off,
                     int64_t total_tuples, const double* h_axes, int32_t* bins, float* scaled, float* scale,
                     float* tr, float* rot, void* stream);

/* generate_target_pairs(point_pairs, a1, a2, a3, center) (dataset.py:118-135) for explicit pairs
 * float32[T,2,3]; centers: float64[B,3] per scene (NULL = zeros). */
int cppf_generate_target_pairs(int B, const float* pairs, const int32_t* tup_off, int64_t total_tuples,
                               const double* h_axes, const double* centers, float* tr, float* rot, void* stream);

/* ---- a6. centre Hough vote + argmax: replaces vote_center (train_dino.py:171-215).
 * cppf_scene_bounds fills CppfSceneGrid[B] (train_dino.py:172-173).
 * cppf_vote_center accumulates the votes and extracts the first maximum.
 *   tr: float32[T,2]; cos_tab/sin_tab: device float32[num_rots] (train_dino.py:194-195 table, built by the caller
 *     so that table values are an input, not a libm detail);
 *   grid / grid_off: optional uint32 output grid (all scenes concatenated, scene b at grid_off[b],
 *     int64 device offsets); pass NULL to keep the accumulator on-chip only.
 *   cells_cap: upper bound on ncell of any scene in the batch (sizes the launch; scenes above it get
 *     flags bit2 and no votes).  mode: 0 = auto, 1 = LDS-slab accumulation (per-slab rotation arcs),
 *     2 = global atomics, 3 = LDS-slab with the exhaustive rotation sweep (A/B reference).
 *     Mode 1 may be OR-ed with CPPF_VC_FRAMES_ONLY (compute the per-pair circle frames into the workspace and
 *     return) or CPPF_VC_FRAMES_READY (the workspace already holds them): the two-call form of the same work, for
 *     callers that time or overlap the vote kernel separately.
 *   vote_wt: NULL = every vote counts 1 (the reference).  Otherwise per-pair weights in [0, 4]; a vote adds
 *     round(w*256) to its cell (integer accumulation, deterministic) -- the uncertainty-weighted accumulator of
 *     BASELINE config 5, an extension the reference does not have; w == 1 yields 256 x the reference grid.
 *   workspace: cppf_vote_center_workspace_bytes(B, cells_cap, total_tuples) bytes.
 *   out_argmax int64[B], out_peak uint32[B] (0xFFFFFFFF for a scene above cells_cap), out_world float64[B,3]. */
#define CPPF_VC_FRAMES_ONLY 0x200
#define CPPF_VC_FRAMES_READY 0x100
int cppf_scene_bounds(int B, const float* pts, const int32_t* pt_off, float res, CppfSceneGrid* out, void* stream);
int64_t cppf_vote_center_workspace_bytes(int B, int64_t cells_cap, int64_t total_tuples);
int cppf_vote_center(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k,
                     const int32_t* tup_off, int max_t, int64_t total_tuples, const float* tr,
                     const float* vote_wt /* optional float32[T]: see below */, double res,
                     int num_rots, const float* cos_tab, const float* sin_tab, const CppfSceneGrid* grids,
                     uint32_t* grid, const int64_t* grid_off, int64_t cells_cap, int mode,
                     void* workspace, int64_t workspace_bytes,
                     int64_t* out_argmax, uint32_t* out_peak, double* out_world, void* stream);

/* ---- a7. back-vote ("noisy pair") filter + importance weights: replaces eval.py:251-275.
 *   centers: float64[B,3] voted centres (T_est); kidx/gamma: per scene, the order-statistic index and
 *   interpolation weight np.percentile(back_errs, ratio*100) uses (device arrays int32[B], float32[B];
 *   cppf2_amd.ops.percentile_params computes them exactly as NumPy does).
 *   Outputs: mask uint8[T]; kept_tuple int32 (scene b's kept tuples, in order, at tup_off[b]..),
 *   kept_count int32[B]; kept_wt float64 (imp_pair_wt, same layout as kept_tuple);
 *   kept_row0 int32 (first row of the pair in the flattened candidate list of vote_rotation, -1 if
 *   |ab| <= 1e-7, train_dino.py:223); back_errs float32[T] and thr float32[B] optional.
 *   workspace: cppf_backvote_workspace_bytes(total_points) bytes. */
int64_t cppf_backvote_workspace_bytes(int64_t total_points, int B);
int cppf_backvote_filter(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k,
                         const int32_t* tup_off, const float* tr, const double* centers, const double* h_axes,
                         const int32_t* kidx, const float* gamma, double imp_wt_margin, int num_rots,
                         uint8_t* mask, int32_t* kept_tuple, int32_t* kept_count, double* kept_wt,
                         int32_t* kept_row0, float* back_errs, float* thr,
                         void* workspace, int64_t workspace_bytes, void* stream);

/* ---- a8+a9. rotation vote: replaces vote_rotation (train_dino.py:218-239) fused with get_topk_dir
 * (eval.py:37-51): candidates are generated in registers and tested against the `S` sphere bins.
 *   rot: float32[T,3] (targets_rot); rot_col: which column holds the angle (0 = up, 2 = "right", eval.py:278,287);
 *   the kept_* arrays come from cppf_backvote_filter; max_kept >= max_b kept_count[b];
 *   sphere: float32[S,3]; cos_thr = float32(cos(2*angle_tol deg)); bmm_size: rows per float32
 *   accumulation chunk (eval.py:81, 100000).  bin_lut (optional, device int16[lut_rows*lut_cols*8]): for each cell of
 *   the partition {row = floor((1-y)*lut_rows/2), col = floor(atan2(z,x) mod 2pi * lut_cols/2pi)} of the unit sphere,
 *   up to 8 bin ids (-1 = empty) covering every bin whose cone can contain a direction of the cell
 *   (cppf2_amd.ops.build_bin_lut builds it for any bin set); NULL selects the exhaustive bin sweep.
 *   Outputs: counts float32[B,S], top_idx int32[B], top_count float32[B].
 *   workspace: cppf_rot_bins_workspace_bytes(B, S, max_kept, num_rots, bmm_size). */
int64_t cppf_rot_bins_workspace_bytes(int B, int S, int max_kept, int num_rots, int bmm_size);
int cppf_rot_bins(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k,
                  const int32_t* tup_off, const float* rot, int rot_col,
                  const int32_t* kept_tuple, const int32_t* kept_count, const double* kept_wt,
                  const int32_t* kept_row0, int max_kept, int num_rots, const float* cos_tab, const float* sin_tab,
                  const float* sphere, int S, float cos_thr, int bmm_size,
                  const int16_t* bin_lut, int lut_rows, int lut_cols,
                  float* counts, int32_t* top_idx, float* top_count,
                  void* workspace, int64_t workspace_bytes, void* stream);
/* Both rotation votes of eval.py:277-293 in one pass over the kept pairs (they share the pair frames and differ
 * only in the angle column: rot_col0 = 0 for the up vote, rot_col1 = 2 for the "right" vote).  Outputs are
 * axis-major: counts float32[2,B,S], top_idx int32[2,B], top_count float32[2,B]; per-axis results are identical to
 * two cppf_rot_bins calls.  Same workspace size. */
int cppf_rot_bins2(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k,
                   const int32_t* tup_off, const float* rot, int rot_col0, int rot_col1,
                   const int32_t* kept_tuple, const int32_t* kept_count, const double* kept_wt,
                   const int32_t* kept_row0, int max_kept, int num_rots, const float* cos_tab, const float* sin_tab,
                   const float* sphere, int S, float cos_thr, int bmm_size,
                   const int16_t* bin_lut, int lut_rows, int lut_cols,
                   float* counts, int32_t* top_idx, float* top_count,
                   void* workspace, int64_t workspace_bytes, void* stream);

/* Global tuple rows of the kept pairs, int32[B, max_kept] (device): row = tup_off[b] + kept_tuple[tup_off[b] + j] for
 * j < kept_count[b], the scene's first tuple row otherwise -- the fixed-shape index a caller gathers per-pair tensors with (the
 * tuple features the scale head runs on, eval.py:272; what cppf_reslayer_tail scatters through) without reading kept_count on
 * the host; a scene without tuples pads with row 0.  Padded entries are valid rows to read and are never written through (cppf_reslayer_tail skips
 * them by kept_count). */
int cppf_kept_rows32(int B, const int32_t* tup_off, const int32_t* kept_tuple, const int32_t* kept_count, int max_kept,
                     int32_t* rows, void* stream);

/* Stand-alone halves with the reference's own signatures (used by the drop-in wrappers):
 * vote_rotation -> up float32[n_valid, num_rots, 3] (valid pairs compacted in order), valid uint8[T];
 * get_topk_dir on explicit candidates float32[M,3] with weights float64[M] (NULL = ones). */
int cppf_vote_rotation(const float* pts, int n_points, const int32_t* idx, int k, int T, const float* rot_angle,
                       int num_rots, const float* cos_tab, const float* sin_tab,
                       float* up, uint8_t* valid, int32_t* n_valid, void* workspace, int64_t workspace_bytes,
                       void* stream);
int cppf_sphere_counts(const float* cand, int64_t M, const double* wt, const float* sphere, int S,
                       float cos_thr, int bmm_size, float* counts, void* workspace, int64_t workspace_bytes,
                       void* stream);

/* ---- a11. pose assembly: replaces eval.py:295-313 (Gram-Schmidt, cross product, median scale).
 *   h_up_axis / h_right_axis: column of R the up / right vote fills (np.where(cfg.up)[0][0]).
 *   pred_scales float32[T,3] (MLP scale head) may be NULL. */
int cppf_assemble_pose(int B, const float* sphere, const int32_t* up_idx, const float* up_count,
                       const int32_t* right_idx, const float* right_count, int up_axis, int right_axis,
                       const int64_t* argmax, const uint32_t* peak, const double* world,
                       const CppfSceneGrid* grids, const float* pred_scales, const int32_t* tup_off,
                       const int32_t* kept_tuple, const int32_t* kept_count,
                       CppfSceneResult* out, void* stream);

/* ---- step behind the path: online alignment refinement, replaces eval.py:319-355 (100 Adam steps on
 * the translation and a quaternion delta, lietorch SO3; mean L1 between the kept pairs' points in the object frame and
 * the predicted pair coordinates, y component only for up-symmetric categories).  scaled float32[T,2,3]
 * (pred_pairs_scaled of cppf_decode_bins); kept_* from cppf_backvote_filter; results: the records of
 * cppf_assemble_pose, whose t and R are replaced in place (flags bit3 set).  lr = 1e-2, steps = 100 in the reference.
 * PARITY UNPINNED (lietorch absent): pinned only by oracle/cppf_oracle.py:refine_pose. */
int cppf_refine_pose(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k,
                     const int32_t* tup_off, const float* scaled, const int32_t* kept_tuple,
                     const int32_t* kept_count, int y_only, int steps, float lr, CppfSceneResult* results,
                     void* stream);

/* ---- the ensemble score and selection (eval.py:358-372; BASELINE configs[2]: both models vote every instance).
 * cppf_alignment_loss: loss[b] (float64) = mean over the kept pairs' end points of clip(|(pc - t) @ R / scale_norm - pred|, 0,
 *   0.1) (all three coordinates, or y only for the up-symmetric categories, eval.py:360-361), pred = bins / (nb - 1) - 0.5
 *   (the un-scaled pair the model drew, eval.py:230), scale_norm = float32 norm of scale_src[b].scale (the DINO pass' record for
 *   both passes, eval.py:308-310; 1 where it is 0); NaN when no pair was kept.  results = the pass' records (t, R).
 * cppf_ensemble_select: out[b] = the record of the model with the smaller loss -- strict '<' against inf, model 0 (DINO)
 *   first; enable0 = geo_branch, enable1 = visual_branch (eval.py:367) -- carrying model 0's scale and the pick (-1: none)
 *   in pad_[0]; best[b] = its loss. */
int cppf_alignment_loss(int B, const float* pts, const int32_t* pt_off, const int32_t* idx, int k, const int32_t* tup_off,
                        const int32_t* bins, int nb, const int32_t* kept_tuple, const int32_t* kept_count, int y_only,
                        const CppfSceneResult* results, const CppfSceneResult* scale_src, double* loss, void* stream);
int cppf_ensemble_select(int B, const CppfSceneResult* rec0, const CppfSceneResult* rec1, const double* loss0,
                         const double* loss1, int enable0, int enable1, CppfSceneResult* out, double* best, void* stream);

/* ---- steps in front of the path ---------------------------------------------------------------
 * Back-projection of a masked depth map: replaces backproject() (utils/util.py:2586-2607) + the sign flip and
 * float32 cast at eval.py:185-189.  depth float32[H,W] in metres, mask uint8[H,W]; h_kinv = inverse intrinsics
 * (9 doubles, host).  Pixels are emitted in row-major order (np.where order): out_pts float32[cap,3],
 * out_rowcol int32[cap,2] (optional), out_count int32 (number of valid pixels; may exceed cap). */
int cppf_backproject(const float* depth, const uint8_t* mask, int H, int W, const double* h_kinv, int cap,
                     float* out_pts, int32_t* out_rowcol, int32_t* out_count, void* stream);
/* backproject() (utils/util.py:2586-2607) exactly as the reference returns it (ABI 9): depth float64[H,W], out_pts float64[cap,3]
 * with x and y NEGATED (the reference's own convention; its callers negate them back, eval.py:187-188) -- the same float64
 * operations in the same order, so the array is bit-identical to NumPy's.  What `utils.util.backproject` binds. */
int cppf_backproject64(const double* depth, const uint8_t* mask, int H, int W, const double* h_kinv, int cap,
                       double* out_pts, int32_t* out_rowcol, int32_t* out_count, void* stream);
/* One uniformly random point per `res` voxel (anchored at the cloud's min corner): replaces downsample()
 * (utils/util.py:39-46, open3d voxel_down_sample_and_trace + np.random.choice).  The draw is Philox(seed, point
 * index), so the kept set does not depend on thread order; out_idx int32[n] holds the kept point indices in
 * ascending order, out_count their number. */
int64_t cppf_voxel_downsample_workspace_bytes(int64_t n);
int cppf_voxel_downsample(const float* pts, int n, float res, uint64_t seed, int32_t* out_idx, int32_t* out_count,
                          void* workspace, int64_t workspace_bytes, void* stream);

/* DINO-branch feature plumbing: replaces interpolate_features (dataset.py:40-59) = grid_sample
 * (bilinear, zeros padding, align_corners=False) of the patch-token map desc at the pixel centres of pts float32[n,2]
 * (x, y), then L2 normalisation over the C channels.  desc is addressed as desc[c*stride_c + y*stride_y + x*stride_x]
 * (elements): patch-major tokens [h*w, C] -> (1, w*C, C); the reference's NCHW view -> (h*w, w, 1).  strides = the
 * reference's `strides` (pixels per token).  out: float32[n,C], or float16[n,C] when out_f16 (BASELINE config 5's
 * fp16 feature tables).  Needs 16*C bytes of LDS: C <= 4096. */
int cppf_interpolate_features(const float* desc, int C, int h, int w, int64_t stride_c, int64_t stride_y,
                              int64_t stride_x, const float* pts, int n, float strides, int normalize, void* out,
                              int out_f16, void* stream);

/* ---- a ResLayer with at most 8 outputs (the scale head's output layer ResLayer(64, 3), train_shot.py:67-71) in plain float32
 * (fmaf chains in index order: a float32 GEMM up to the summation order, independent of the row count), one thread per row:
 *     out[r] = skip(x[i]) + relu(x[i, :k_in] W1^T + b1) W2^T,   skip = x W0^T + b0 (b0 carries fc2's bias), or x when w0 == NULL
 * x float32 [rows, >= k_in], row stride ldx (multiple of 4, 16-byte aligned base), k_in % 4 == 0; w1, w0 float32 [n_out, k_in],
 * w2 float32 [n_out, n_out] (nn.Linear layout).  r = i, or scatter_rows[i] (int32) when given; with valid_count (int32
 * [rows / per_group]) entry i is written only if (i % per_group) < valid_count[i / per_group]: the kept-pair lists of
 * cppf_kept_rows32 (per_group = max_kept, valid_count = kept_count) scattered into a [total_tuples, n_out] buffer, real pairs
 * only, each exactly once (eval.py:272 reads the scale head's rows of the kept pairs). */
int cppf_reslayer_tail(const float* x, int64_t ldx, int32_t k_in, int32_t n_out, int64_t rows, const float* w1, const float* b1,
                       const float* w0, const float* b0, const float* w2, const int32_t* scatter_rows,
                       const int32_t* valid_count, int32_t per_group, float* out, int64_t ldo, void* stream);

/* ---- the ResLayers of the models (train_shot.py:19-45; bn = dropout = False), one or several per kernel, on the bf16
 * matrix cores in split-float32 arithmetic (every float32 operand is the exact sum of three bf16 values; the six largest of the
 * nine bf16 products, each exact in float32, per K step; float32 accumulate.  Error against float64: within 3 x a library
 * float32 GEMM's -- measured 2.4 x, tests/test_mlp_split.py holds e_split < 3 e_native + 2e-7 -- see cppf_mlp_split.hip):
 *     y = skip(x) + relu(x[:, :k_in] W1^T + b1) W2^T          skip(x) = x when b0 == NULL (then k_in == n_out; out may be
 *                                                              x itself: in place), x W0^T + b0 otherwise
 *     then `chain` times                       y <- y + relu(y W1_l^T + b1_l) W2_l^T   without leaving the registers
 * x float32 with row stride ldx >= k_in, out float32 with row stride ldo >= n_out (elements; both multiples of 4, base
 * pointers 16-byte aligned); k_in a multiple of 8 (columns of x beyond the layer's true dim_in must hold finite values;
 * their weights are zero in the stream); n_out in {64, 128, 192, 256}; b1 float32[(1 + chain) * n_out].  wq = the weights
 * pre-split into bf16 triples in the per-lane operand order the kernel streams (cppf2_amd.models.pack_split writes it;
 * cppf_reslayer_split_stream_bytes gives its size, -1 for unsupported shapes).  Each layer's fc2 bias is the caller's (it commutes
 * with the residual stream).
 * sched (ABI 10; every cppf_reslayer_split* entry point takes it in front of `stream`): NULL, or int32[2] device memory of the
 * caller's, ZERO before the first launch that uses it.  With it the launch's persistent workgroups claim their 128- / 256-row
 * blocks from a counter instead of taking a fixed share each: workgroups that start late or run slower -- another stream's
 * kernels on the chip, the last partial round of a static split -- take fewer blocks (tuple MLP 9.17 -> 8.94 ms).  A launch
 * leaves the two words zero again, so launches that cannot overlap (one stream) may share one buffer; launches on DIFFERENT
 * streams need different buffers.  Results never depend on it (a row block's arithmetic does not depend on who runs it). */
int64_t cppf_reslayer_split_stream_bytes(int32_t k_in, int32_t n_out, int32_t proj, int32_t chain);
int cppf_reslayer_split(const float* x, int64_t ldx, int32_t k_in, float* out, int64_t ldo, int32_t n_out, int64_t rows,
                        const void* wq, int64_t wq_bytes, const float* b1, const float* b0, int32_t chain, int32_t* sched,
                        void* stream);
/* The same with a second output: first_out float32 [rows, >= n_out] (row stride ld_first, a multiple of 4; 16-byte aligned; a
 * buffer of its own) receives the activation after the FIRST layer, out the one after the whole chain -- for a stack whose
 * intermediate result is read elsewhere (the SHOT / DINO models' 256-wide tuple features feed the scale head,
 * train_shot.py:112-114) while the identity layers behind it continue in registers. */
int cppf_reslayer_split_tap(const float* x, int64_t ldx, int32_t k_in, float* first_out, int64_t ld_first, float* out,
                            int64_t ldo, int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes, const float* b1,
                            const float* b0, int32_t chain, int32_t* sched, void* stream);

/* ---- prepare_tuple_inputs (train_shot.py:75-83) and the tuple encoder's first launch (train_shot.py:100-111) in ONE kernel: the
 * first ResLayer of the tuple encoder (a 128-wide projection layer, `chain` identity layers behind it) on rows
 * [40 pair features | table[pt_off[b] + idx[t, 0]] | ... | table[pt_off[b] + idx[t, 4]]] that are never written: the pair features of
 * a 5-point tuple -- the first 40 columns of cppf_encode_tuples_shot's rows, bit for bit -- are computed by the kernel's own lanes
 * from pts / normals float32 [points, 3] and the sampler's scene-local idx int32 [rows, 5] (pt_off / tup_off int32 [B + 1]) and go
 * straight into its x tiles; the descriptors are gathered from table float32 [points, fdim] (the point encoder's output; fdim a
 * power of two >= 8) by the x-tile fetches.  Bit-identical to cppf_reslayer_split on the materialised rows (1.8 GB per 64
 * scenes saved).  k = 5, n_out = 128; wq / b1 / b0 / chain as cppf_reslayer_split for k_in = 40 + 5 fdim.  (The two-kernel forms
 * of rounds 3-4, cppf_encode_tuples_shot_heads + cppf_reslayer_split_gather, are in cppf_hip_experimental.h.) */
int cppf_reslayer_split_encode(int B, const float* pts, const float* normals, const int32_t* idx, int32_t k,
                               const int32_t* pt_off, const int32_t* tup_off, const float* table, int32_t fdim, float* out,
                               int64_t ldo, int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes, const float* b1,
                               const float* b0, int32_t chain, int32_t* sched, void* stream);

/* ---- the DINO model's tuple encode without its rows, and every Linear of both models on the matrix cores (train_dino.py:86-97,
 * 128-133; reference call site eval.py:221).
 * cppf_linear_split: out[rows, n_out] = x[rows, :k_in] W^T + bias (bias may be NULL), a plain nn.Linear in the split arithmetic of
 *   cppf_reslayer_split; n_out a multiple of 256, k_in a multiple of 8; wq = cppf_linear_split_stream_bytes(k_in, n_out) bytes
 *   (cppf2_amd.models.pack_linear).  Replaces the library GEMMs of desc_transform (train_dino.py:86, 95) and of the per-slot
 *   slices of desc_pair_transform (train_dino.py:87, 96).
 * cppf_reslayer_split_sumencode: prepare_tuple_inputs (train_dino.py:91-97) + the tuple encoder's first ResLayer (128-wide projection
 *   layer, `chain` identity layers behind it) on rows [30 coordinate differences | s(t)] that are never written, where s(t) =
 *   desc_pair_transform of the concatenated desc_transform outputs of the tuple's points (train_dino.py:95-96) is linear in per-point
 *   vectors: the per-point parts of x W1^T and x W0^T come from tables float32 [points, 5, 256] (= [W1_i p | W0_i p] per point p and
 *   slot i; point pitch ld_tables floats) that cppf_linear_split wrote from host-folded weights, and are summed into the accumulators
 *   in slot order; the coordinate columns are built by the kernel's own lanes from pts float32 [points, 3] and the sampler's
 *   scene-local idx int32 [rows, 5] (pt_off / tup_off int32 [B + 1]) and are the only columns that run through the first product.
 *   wq = cppf_reslayer_split_stream_bytes(32, 128, 1, chain) bytes; b1 / b0 include the folded transforms' bias terms.  k = 5,
 *   n_out = 128.  Algebraically the reference's network; the [T, 286] rows are never formed.  (Two-kernel form:
 *   cppf_encode_tuples_coord_heads + cppf_reslayer_split_sumgather, cppf_hip_experimental.h.) */
int64_t cppf_linear_split_stream_bytes(int32_t k_in, int32_t n_out);
int cppf_linear_split(const float* x, int64_t ldx, int32_t k_in, float* out, int64_t ldo, int32_t n_out, int64_t rows,
                      const void* wq, int64_t wq_bytes, const float* bias, void* stream);
int cppf_reslayer_split_sumencode(int B, const float* pts, const int32_t* idx, int32_t k, const int32_t* pt_off,
                                  const int32_t* tup_off, const float* tables, int64_t ld_tables, float* out, int64_t ldo,
                                  int32_t n_out, int64_t rows, const void* wq, int64_t wq_bytes, const float* b1, const float* b0,
                                  int32_t chain, int32_t* sched, void* stream);

/* ---- the bin draw fused into the MLP's output layer (eval.py:225-229 behind train_shot.py:62-66): the 192-wide projection
 * ResLayer of the logit head (6 coordinates x 32 bins) with  bins[t, c] = inverse-CDF draw of softmax(logits[t, c, :]) at
 * uniforms[t, c]  as its epilogue -- cppf_decode_bins' arithmetic bit for bit -- so that the logits are never written;
 * cppf_decode_from_bins then computes what the second half of cppf_decode_bins computes (scaled pair, scale, targets_tr,
 * targets_rot) from the bins.  x / wq / b1 / b0 as cppf_reslayer_split (n_out = 192, chain = 0); uniforms float32 [rows, 6];
 * bins int32 [rows, 6].  (With a logit prior -- the synthetic benchmarks' stand-in for trained weights; the reference has
 * none -- cppf_reslayer_split_decode_prior, cppf_hip_experimental.h.) */
int cppf_reslayer_split_decode(const float* x, int64_t ldx, int32_t k_in, int64_t rows, const void* wq, int64_t wq_bytes,
                               const float* b1, const float* b0, const float* uniforms, int32_t* bins, int32_t* sched,
                               void* stream);
int cppf_decode_from_bins(int B, const int32_t* bins, int nb, const float* pts, const int32_t* idx, int k,
                          const int32_t* pt_off, const int32_t* tup_off, int64_t total_tuples, const double* h_axes,
                          float* scaled, float* scale, float* tr, float* rot, void* stream);

/* Batch mode (two HIP streams working on different batches): the cppf_reslayer_split* launches are
 * persistent -- one workgroup per CU holding the CU's whole register file -- so while one runs, no kernel of another stream
 * can start anywhere on the chip.  cppf_mlp_reserve_cus(n) makes every later launch of this process use n fewer CUs (never
 * fewer than half of them); the kernels are power-limited, so 32 fewer CUs cost them ~4 % while the other stream's voting /
 * descriptor kernels run beside them.  Use one CU per shader engine (CUs / 8 = 32 on an MI355X): workgroups are placed
 * round-robin over the engines, and one engine without a free CU stalls the other stream's launch (docs/measurements.md
 * 11.7).  0 (default) = one workgroup per CU.  Results never depend on it.  Returns the PREVIOUS reservation (>= 0; ABI 11) so
 * that nested users restore it instead of zeroing it; a negative argument only queries.  With the per-device cache named in the
 * conventions above this is the library's only process-wide state; it is an atomic integer, safe to set from any thread, and
 * applies to launches enqueued afterwards. */
int cppf_mlp_reserve_cus(int32_t cus);

#ifdef __cplusplus
}
#endif
#endif /* CPPF_HIP_H_ */
