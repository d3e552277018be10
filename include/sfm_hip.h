/* C ABI of libsfm_hip.so: the MI355X (gfx950) implementation of the robust-estimation hot
 * path of the Structure-from-Motion reference pipeline.
 *
 * The reference exposes this path as Python functions, not as an FFI; each entry point
 * below names the reference function (file:line under the reference checkout) whose
 * arithmetic it replaces.  The Python package structure-from-motion_amd/sfm_hip binds these
 * with ctypes and keeps the reference's function names and signatures on top.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless the parameter comment says "host".
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *    enqueue work; nothing here synchronises with the host.
 *  - Return value: 0 on success, otherwise a hipError_t value or SFM_EINVAL for rejected
 *    arguments; sfm_hip_error_string() turns it into text.
 *  - A correspondence is (x1, y1, x2, y2): K-normalised image coordinates of one match in
 *    view 1 and view 2, third homogeneous coordinate 1; `corr` is an array [n][4].
 *  - A hypothesis is a row of 12 int32 indices into `corr`: columns 0-7 feed the 8-point
 *    essential-matrix solver, columns 8-11 the 4-point homography solver
 *    (src/ransac_algorithm.py:94 and :116 draw them in this order).
 *  - 3x3 matrices are row-major [9]; cameras are row-major 3x4 [12].
 *  - Sample indices (`idx`, `row`) must lie in [0, n) (for the translation stage: in [0, points of that camera)).
 *    The kernels clamp every index into that range before they use it, so a bad index cannot fault the GPU; the model
 *    of such a hypothesis is then that of the clamped sample.  Only the record entry points reject a bad row.
 *  - `xmax` parameters are a HINT: max |coordinate| over corr as far as the caller knows it (0 = unknown).  The
 *    library measures the maximum itself while it prepares the match-side operands and uses the larger of the two, so
 *    an understated hint cannot void a bound.  Matches with a NaN, infinite or absurdly large (>= 1e6) coordinate have
 *    no usable bound: upper-bound entry points then return the trivial bound n, exact entry points evaluate every pair
 *    in float64.  Negative or NaN hints are rejected.
 *  - Thresholds follow the reference's comparisons: the epipolar test squares its threshold
 *    (src/ransac_algorithm.py:102), so its sign is irrelevant; the transfer and the reprojection test compare a norm
 *    with the threshold itself (:327, src/estimate_translation.py:127), so a non-positive one admits no match.
 */
#ifndef SFM_HIP_H
#define SFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFM_EINVAL 100001
#define SFM_ETOOLARGE 100002 /* sfm_eh_score_noalloc: n > SFM_NOALLOC_MAX_MATCHES */

const char* sfm_hip_error_string(int code);
int sfm_hip_abi_version(void);

/* Fused minimal solvers + inlier counting for nb hypotheses.
 * Replaces one pass of the loop body src/ransac_algorithm.py:94-102 and :116-121
 * (estimate_F_DLT :216-263, enforce_essential :176-190, epipolar_errors :193-213,
 * estimate_H_DLT :266-293, get_inlier_mask_H :296-329) for every hypothesis.
 * thr_e = err_threshold, thr_h = 3*err_threshold.  count_e/count_h: int32[nb]. */
int sfm_eh_score_f64(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                     int32_t* count_e, int32_t* count_h, void* stream);

/* float32 copy of the correspondences for sfm_eh_upper_f32 and sfm_eh_best: corr32 = float[n + 1][4], 16-byte aligned,
 * sfm_corr_f32_bytes(n) bytes: the n matches and one trailer row whose first word is the measured max |coordinate|. */
int64_t sfm_corr_f32_bytes(int32_t n);
int sfm_corr_to_f32(const double* corr, int32_t n, float* corr32, void* stream);

/* Float32 pre-filter for the same hypotheses: UPPER BOUNDS up_e/up_h (int32[nb]) of the counts
 * sfm_eh_score_f64 would return.  Models are solved in float64, scaled to max-abs 1, rounded to
 * float32 and scored against corr32 with upper-bound forms of the two tests whose slack is
 * derived per hypothesis from a float32 error bound; xmax: hint, see Conventions (corr32 carries the
 * measured value), safety >= 1 multiplies the bound (2 is the default of the Python layer).
 * one_sided != 0 selects cheaper, looser bounds: for E the symmetric test implies the one-sided
 * test "distance to the epipolar line in view 2 < sqrt(2) thr" (no E^T x2), for H the disc test
 * implies the per-component box test (no squares); 26 instead of 38 instructions per residual
 * pair, still up >= exact.
 * Use: take the hypothesis with the largest upper bound, get its exact count with
 * sfm_eh_score_f64, then re-score exactly every hypothesis whose upper bound reaches that
 * count; the first maximum of the exact counts is the winner of src/ransac_algorithm.py:105 /
 * :124 over the block (sfm_hip/ops.py:eh_best does exactly this). */
int sfm_eh_upper_f32(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e,
                     double thr_h, double xmax, double safety, int32_t one_sided, int32_t* up_e, int32_t* up_h,
                     void* stream);

/* The whole arg-max step as one call: the exact winners of a block of hypotheses, i.e. the first
 * maximum of the counts sfm_eh_score_f64 would return (largest count, lowest id), for E and for H.
 * Internally a cascade of upper bounds: float32 bounds for all hypotheses (loose = 0: the tight
 * forms; 1: the cheaper one-sided/box forms on the vector ALU; 2: the same forms with E x1 and H x1 on
 * the bf16 matrix cores, the fastest) -> exact count of the hypothesis with the largest bound -> candidates whose
 * bound reaches it -> tight float32 bounds when there are many -> float64 for the rest.  Replaces
 * the best-so-far bookkeeping of src/ransac_algorithm.py:102-106 and :121-125 over nb iterations
 * when only the final winner is wanted.
 * workspace: device scratch of at least sfm_eh_best_workspace_bytes(nb, n) bytes, 256-byte aligned
 * (nothing is allocated on the device here).  result (HOST, int64[8]): count_e, index_e, count_h, index_h,
 * candidates_e, candidates_h, re-bounded candidates, hypotheses scored in float64.
 * Unlike the other entry points this one synchronises `stream` (it reads a few words back to decide
 * the next level). */
int64_t sfm_eh_best_workspace_bytes(int64_t nb, int32_t n);
/* The same step split in two, so that a caller can keep several blocks of hypotheses in flight on different
 * streams: submit only enqueues work on `stream` (all bound levels for the usual case of at most 256 candidates
 * per class, and one copy of the outcome into host_pending: sfm_eh_best_pending_bytes() bytes of host memory,
 * page-locked if the call is to return before the work is done); collect waits for `stream`, reads host_pending
 * and, only when there were more candidates than that, runs the remaining levels.  Both take the same arguments;
 * one workspace and one host_pending per block in flight. */
int64_t sfm_eh_best_pending_bytes(void);
int sfm_eh_best_submit(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e,
                       double thr_h, double xmax, double safety, int32_t loose, void* workspace, int64_t workspace_bytes,
                       void* host_pending, void* stream);
int sfm_eh_best_collect(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e,
                        double thr_h, double xmax, double safety, int32_t loose, void* workspace, int64_t workspace_bytes,
                        const void* host_pending, int64_t* result, void* stream);
int sfm_eh_best(const double* corr, const float* corr32, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                double xmax, double safety, int32_t loose, void* workspace, int64_t workspace_bytes, int64_t* result,
                void* stream);

/* sfm_eh_score_f64 with the bulk of the residual tests on the bf16 matrix cores + float32 and IDENTICAL counts: the
 * bilinear form s = x2^T E x1, the squared line norms |(E x1)_{0,1}|^2, |(E^T x2)_{0,1}|^2 and the row H x1 come from MFMAs
 * of three-piece splits; a pair whose float32 test value lies inside the rigorous error band around the threshold is
 * re-evaluated with the float64 test inside the kernel (queued per wave, 64 pairs at a time).
 * Early loops (default): each test is implied by a cheaper one -- the symmetric epipolar test of src/ransac_algorithm.py:98-102
 * by its one-sided half (distance to the epipolar line in the second view alone, widened by the rigorous error band), the
 * disc test of :120-121, :320-327 by its x strip (|dx| below the radius) -- and a pair that fails the cheaper test is proven to be no float64
 * inlier without the full test; the few pairs per thousand that pass take the float64 test.  A wave whose hypotheses come from
 * all-inlier samples (most pairs pass) goes over to the full float32 tests.  The counts are the same either way.
 * sfm_eh_score_mixed_opt: options = SFM_EHX_EVERY_PAIR_FULL_TEST sends EVERY pair through the full symmetric-epipolar and
 * disc tests (no early loop; what bench.py's headline rate is measured on), 0 = sfm_eh_score_mixed.
 * xmax: hint, see Conventions.  workspace: sfm_eh_score_mixed_workspace_bytes(nb, n) bytes, 256-byte aligned. */
#define SFM_EHX_EVERY_PAIR_FULL_TEST 1
int64_t sfm_eh_score_mixed_workspace_bytes(int64_t nb, int32_t n);
int sfm_eh_score_mixed(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, double xmax,
                       void* workspace, int64_t workspace_bytes, int32_t* count_e, int32_t* count_h, void* stream);
int sfm_eh_score_mixed_opt(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h, double xmax,
                           int32_t options, void* workspace, int64_t workspace_bytes, int32_t* count_e, int32_t* count_h,
                           void* stream);

/* sfm_eh_score_mixed for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): same counts, no workspace argument, and no hipMalloc / hipFree or other
 * allocating call inside.  It has no counterpart in the reference.  The match-side operands (264 bytes per match) go into
 * a fixed array inside the library, SFM_NOALLOC_MAX_MATCHES * 264 bytes (8.25 MiB) that every GPU the library runs on
 * holds until the process ends; n beyond that capacity is rejected with SFM_ETOOLARGE.  All calls on one GPU share
 * that array: calls that overlap on one GPU (different streams or threads) race; calls on one stream are safe.
 * The array used is the one of the GPU that owns `stream` (of the current GPU for a NULL stream); the kernels are
 * launched on `stream`, so it should belong to the current device like the data pointers.  The first call on a GPU
 * looks up the array's address there (which may load the library's code object): make one call per GPU outside graph
 * capture before capturing one. */
#define SFM_NOALLOC_MAX_MATCHES 32768
int sfm_eh_score_noalloc(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double thr_e, double thr_h,
                         int32_t* count_e, int32_t* count_h, void* stream);

/* The models themselves: F_out (estimate_F_DLT, :216-263, before the singular values are
 * replaced), E_out (after enforce_essential, :176-190) and H_out (estimate_H_DLT, :266-293),
 * each double[nb][9] or NULL.  Same solvers as sfm_eh_score_f64. */
int sfm_eh_models_f64(const double* corr, int32_t n, const int32_t* idx, int64_t nb, double* F_out, double* E_out,
                      double* H_out, void* stream);

/* estimate_F_DLT (src/ransac_algorithm.py:216-263, homography = 0, n >= 8) or estimate_H_DLT (:266-293, homography = 1,
 * n >= 4) over ALL n given correspondences: out = double[9], row-major F (sign arbitrary, as the singular vector's) or
 * H divided by H[2][2].  One wave; for callers of those two functions with more than a minimal sample. */
int sfm_dlt_general_f64(const double* corr, int32_t n, int32_t homography, double* out, void* stream);

/* Inlier mask (uint8[n], 0/1) and count (int32[1], may be NULL) of one essential matrix:
 * the symmetric epipolar test of src/ransac_algorithm.py:98-102 / :133-138. */
int sfm_epipolar_mask_f64(const double* E, const double* corr, int32_t n, double thr, uint8_t* mask, int32_t* count,
                          void* stream);

/* Inlier mask and count of one homography: src/ransac_algorithm.py:296-329. */
int sfm_transfer_mask_f64(const double* H, const double* corr, int32_t n, double thr, uint8_t* mask, int32_t* count,
                          void* stream);

/* Point-to-epipolar-line distances d[n] of one matrix, src/ransac_algorithm.py:193-213.
 * x1, x2: the 3xN homogeneous point arrays, row stride ld doubles. */
int sfm_epipolar_errors_f64(const double* F, const double* x1, const double* x2, int32_t n, int64_t ld, double* d,
                            void* stream);

/* U diag(1,1,0) V^T of one 3x3 matrix, src/ransac_algorithm.py:176-190. */
int sfm_enforce_essential_f64(const double* F, double* E, void* stream);

/* Four motion candidates of an essential matrix, P2s double[4][12]:
 * src/ransac_algorithm.py:422-435 (SVD, determinant sign fix, [UWVt|+-u3], [UW'Vt|+-u3]). */
int sfm_essential_candidates_f64(const double* E, double* P2s, void* stream);

/* Chirality counts of four candidate second cameras over the correspondences selected by
 * `mask` (NULL = all): DLT triangulation against P1=[I|0] and the test
 * (P2 X)_z > 0 and (P1 X)_z > 0 of src/chirality_check.py:22-28 and :39-60.
 * counts: int32[4] (overwritten). */
int sfm_chirality4_f64(const double* P2s, const double* corr, const uint8_t* mask, int32_t n, int32_t* counts,
                       void* stream);

/* Homography -> two (R,t) solutions, src/ransac_algorithm.py:332-403 including the sign vote
 * over all n correspondences (:359-361); H itself is not modified.
 * out: double[43] = R1[9] t1[3] R2[9] t2[3] Ea[9] Eb[9] with E = [t]x R (:129-130), then 1.0 if the
 * vote negated H (the reference negates its argument in place in that case) else 0.0. */
int sfm_homography_to_rt_f64(const double* H, const double* corr, int32_t n, double* out, void* stream);

/* DLT triangulation of m points, src/utils.py:17-36.  P1, P2: double[12]; x1, x2: the first two
 * rows of the 3xM point arrays, row stride ld doubles; X: double[4][m] (row-major, last row 1). */
int sfm_triangulate_f64(const double* P1, const double* P2, const double* x1, const double* x2, int32_t m, int64_t ld,
                        double* X, void* stream);

/* Packs one camera's 2D-3D correspondences for the translation stages:
 * pts[j] = (R X_j, x_j) = double[6], with X promoted to double exactly as `R @ X` does in
 * src/estimate_translation.py:98,123 and src/refine_translation.py:125.
 * R: double[9]; X: [3][m] float32 (x_is_f64 = 0; what find_correspondences returns,
 * src/utils.py:90) or float64 (x_is_f64 = 1); x: double[3][m]. */
int sfm_pack_points_f64(const double* R, const void* X, int32_t x_is_f64, const double* x, int32_t m, double* pts,
                        void* stream);

/* 2-point translation RANSAC for ncam cameras in one launch:
 * src/estimate_translation.py:62-75 (estimate_T_2point :80-106, count_inliers :109-129).
 * pts: packed points of all cameras back to back; cam_off (HOST, int64[ncam+1]): first point
 * of each camera; idx: int32[ncam][nb][2], indices local to the camera; thr = 4*pixel/fx.
 * counts: int32[ncam][nb] workspace/output; best: uint64[ncam] = (count << 32) | (0xFFFFFFFF - index)
 * of the first maximum; T_best: double[ncam][3], the winner's translation (unspecified when
 * the count is 0: the reference returns None then). */
int sfm_t_ransac_f64(const double* pts, const int64_t* cam_off, int32_t ncam, const int32_t* idx, int64_t nb, double thr,
                     int32_t* counts, uint64_t* best, double* T_best, void* stream);

/* Inlier counts of nb given translations T double[nb][3] over one camera's packed points:
 * count_inliers, src/estimate_translation.py:109-129.  counts: int32[nb]. */
int sfm_t_count_f64(const double* pts, int32_t m, const double* T, int64_t nb, double thr, int32_t* counts, void* stream);

/* First maximum of an inlier-count array, folded into *best with max():
 * key = (count << 32) | (0xFFFFFFFF - (index_base + i)), so that a larger count wins and, among
 * equal counts, the lowest global hypothesis id: the strict `>` update of
 * src/estimate_translation.py:73 and src/ransac_algorithm.py:105.  The caller zeroes *best
 * (uint64[1]) before the first call; partial ranges and other GPUs' results combine by max. */
int sfm_argmax_count_i32(const int32_t* counts, int64_t nb, uint64_t index_base, uint64_t* best, void* stream);

/* float32 copy of packed points for sfm_t_upper_f32: pts32 float[m][8] = (RX, x, y, 0, 0, 0). */
int sfm_pts_to_f32(const double* pts, int64_t m, float* pts32, void* stream);

/* Float32 pre-filter of sfm_t_ransac_f64: UPPER BOUNDS up int32[ncam][nb] of the counts it would
 * compute (same arguments; T solved in float64, scored in float32 with a slack derived from the
 * float32 error bound).  rmax (HOST, double[ncam]): max |component of R X| per camera; xmax: max
 * |x|,|y| over all cameras; safety >= 1.  PRECONDITION (not checked, unlike the E/H entry points): rmax and xmax are
 * true maxima; understated values void the bound.  sfm_t_ransac_mixed measures its maxima itself and is the
 * entry point the package uses.  Use as sfm_eh_upper_f32: exact-score the leader, then
 * every hypothesis whose bound reaches its count (sfm_hip/ops.py:t_best). */
int sfm_t_upper_f32(const double* pts, const float* pts32, const int64_t* cam_off, int32_t ncam, const int32_t* idx,
                    int64_t nb, double thr, const double* rmax, double xmax, double safety, int32_t* up, void* stream);

/* Row-wise sfm_argmax_count_i32 (index_base 0): best uint64[nrows] is overwritten. */
int sfm_argmax_count_batched_i32(const int32_t* counts, int64_t nb, int32_t nrows, uint64_t* best, void* stream);

/* Translations of given samples only: T_out double[nb][3] (estimate_T_2point, :80-106). */
int sfm_t_models_f64(const double* pts, int32_t m, const int32_t* idx, int64_t nb, double* T_out, void* stream);

/* Fixed-damping Levenberg-Marquardt on T for ncam cameras in one launch:
 * src/refine_translation.py:57-104 with compute_reprojection_error :107-132, ComputeUpdate
 * :135-155 and projection_derivatives_wrt_T :158-200, including the reference's pairing of the
 * residual vector [all dx | all dy] with the interleaved Jacobian rows.
 * K: double[9]; T0: double[ncam][3]; work: double[2 * total_points] scratch;
 * T_out: double[ncam][3]; history: double[ncam][iters][3] or NULL. */
int sfm_lm_translation_f64(const double* K, const double* pts, const int64_t* cam_off, int32_t ncam, const double* T0,
                           int32_t iters, double mu, double* work, double* T_out, double* history, void* stream);

/* Index sets of a hypothesis range generated on the device with a counter-based generator:
 * the counterpart of the sample draws `np.random.randint(0, N, size=8)` / `size=4`
 * (src/ransac_algorithm.py:92,116) and `size=2` (src/estimate_translation.py:65) for runs that
 * shard a hypothesis range over several GPUs: word w of hypothesis id is
 * Philox4x32-10(counter=(id_lo, id_hi, w/4, 0), key=(seed_lo, seed_hi))[w%4] and the index is
 * (word * n) >> 32, so any split of [first_id, first_id + nb) yields the same sets.
 * width: 12 (E/H, idx int32[nb][12], 16-byte aligned) or 2 (translation, idx int32[nb][2]).
 * (Runs that must follow the reference's seeded global np.random stream draw on the host.) */
int sfm_index_sets_i32(uint64_t seed, uint64_t first_id, int64_t nb, int32_t n, int32_t width, int32_t* idx, void* stream);

/* Matrix-core form of the loose float32 bounds (same meaning and guarantees as sfm_eh_upper_f32 with
 * one_sided = 1: up >= exact count for every hypothesis).  The epipolar residual x2' E x1 and the squared line norm
 * |(E x1)_{0,1}|^2 of src/ransac_algorithm.py:206-212 (a bilinear and a quadratic form) and the product H x1 of :320 run
 * on the bf16 matrix cores from three-piece bf16 splits of the float64 models and coordinates; the comparisons and the
 * counts stay on the vector ALU.
 * corrm: match-side operands written by sfm_corr_to_mfma FOR THE SAME thr_e (the second view's coordinates are stored
 * divided by the square root of the slackened threshold constant); sfm_corr_mfma_bytes(n) bytes (136 per match, rounded
 * up to 32 matches, + 16), 16-byte aligned. */
int64_t sfm_corr_mfma_bytes(int32_t n);
int sfm_corr_to_mfma(const double* corr, int32_t n, double thr_e, void* corrm, void* stream);
int sfm_eh_upper_mfma_f32(const double* corr, const void* corrm, int32_t n, const int32_t* idx, int64_t nb, double thr_e,
                          double thr_h, double xmax, double safety, int32_t* up_e, int32_t* up_h, void* stream);
/* Measurement aid: only the launch of the epipolar test (part = 1: the inner loop of
 * src/ransac_algorithm.py:98-102) or only the transfer test (part = 2: :120-121); the other output is zero. */
int sfm_eh_upper_mfma_part_f32(const double* corr, const void* corrm, int32_t n, const int32_t* idx, int64_t nb, double thr_e,
                               double thr_h, double xmax, double safety, int32_t part, int32_t* up_e, int32_t* up_h,
                               void* stream);

/* sfm_t_ransac_f64 with the bulk of the residual tests on the bf16 matrix cores + float32 and IDENTICAL counts:
 * dx, dy and thr*pz of src/estimate_translation.py:123-127 are bilinear in (hypothesis, correspondence) and come from
 * one MFMA each per 32 x 32 pairs (three-piece splits of the float64 values); a pair whose float32 value lies within
 * the rigorous error band of the threshold (about one in 10^6) is re-evaluated in float64 inside the kernel.
 * workspace: device scratch of sfm_t_ransac_mixed_workspace_bytes(cam_off, ncam) bytes, 256-byte aligned. */
int64_t sfm_t_ransac_mixed_workspace_bytes(const int64_t* cam_off, int32_t ncam);
int sfm_t_ransac_mixed(const double* pts, const int64_t* cam_off, int32_t ncam, const int32_t* idx, int64_t nb, double thr,
                       void* workspace, int64_t workspace_bytes, int32_t* counts, uint64_t* best, double* T_best, void* stream);

/* Record events of the E/H loop as ONE call each: one kernel launch and one read-back (every workgroup repeats the
 * short serial part - models of the sample, sign vote, decomposition, motion candidates - and handles its own matches).  row: HOST int32[12], the index set of the recording iteration;
 * scratch: device double[sfm_record_scratch_doubles()]; host_out: host double[same size]; both calls synchronise.
 * sfm_e_record_f64 = src/ransac_algorithm.py:105-113: E of the sample, its inlier mask over all matches (mask, device
 * uint8[n]) and essential_to_RT over ALL matches; host_out: E at 0..8, the four candidate [R|t] at 18..65 (row-major
 * 3x4 each), their chirality counts as int32[4] at double slot 66, the mask count as int32 at slot 68.
 * sfm_h_record_f64 = :121-147: H of the sample, homography_to_RT, and for each of its two solutions the epipolar
 * mask of E = [t]x R (mask_a / mask_b) and essential_to_RT over the masked matches; host_out (slots 0..8 unspecified): H at 9..17,
 * the homography_to_RT output at 70..112, the second solution's candidates at 113..160, counts at 161, mask count at 163. */
int64_t sfm_record_scratch_doubles(void);
int sfm_e_record_f64(const double* corr, int32_t n, const int32_t* row, double thr, uint8_t* mask, double* scratch,
                     double* host_out, void* stream);
int sfm_h_record_f64(const double* corr, int32_t n, const int32_t* row, double thr, uint8_t* mask_a, uint8_t* mask_b,
                     double* scratch, double* host_out, void* stream);
/* nev homography record events in one launch.  The positions of the H record events of a block of iterations are known
 * as soon as its counts are (best_num_inliers_H only ever rises, src/ransac_algorithm.py:124-125: they are the strict
 * running maxima of the H counts), so the loop's host side computes them all at once instead of one round trip each.
 * rows: DEVICE int32[nev][12] (indices clamped into [0, n)); mask_a, mask_b: uint8[nev][n]; scratch (device) and
 * host_out (host): double[nev][sfm_record_scratch_doubles()], each block laid out as for sfm_h_record_f64.  Synchronises. */
int sfm_h_record_batch_f64(const double* corr, int32_t n, const int32_t* rows, int32_t nev, double thr, uint8_t* mask_a,
                           uint8_t* mask_b, double* scratch, double* host_out, void* stream);

/* ---- sample draws of the drop-in path (host only, no GPU involved) ---------------------------
 * out[0..count) = what np.random.randint(0, n, size=count) returns for NumPy's legacy global generator in the state
 * (key[624], *pos), which is advanced in place exactly as NumPy advances it: MT19937 words, each masked with the
 * smallest 2^k - 1 >= n - 1 and redrawn while above n - 1 (n = 1 consumes nothing).  Replaces the draws of
 * src/ransac_algorithm.py:94,116 and src/estimate_translation.py:64 at about 1 ns per integer instead of NumPy's
 * 5-25 ns.  key and pos are HOST pointers (the package passes the address of NumPy's own state); n < 2^31. */
int sfm_legacy_randint_fill(uint32_t* key, int32_t* pos, int64_t n, int64_t count, int32_t* out);

/* ---- descriptor matching (feature front-end next to the path) ------------------------------
 * Replaces `cv2.BFMatcher().knnMatch(desc_query, desc_train, k=2)` followed by the ratio test
 * `m.distance < ratio * n.distance` (src/extract_sift.py:117-123 with ratio 0.4,
 * src/utils.py:82-86 with ratio 0.75) for SIFT descriptors as OpenCV emits them: 128 integer-valued
 * components in [0, 255].  Squared distances are exact integers (int8 matrix cores); the reported
 * distances are their correctly rounded float32 square roots; ties go to the lowest train index.
 *
 * sfm_descriptors_to_i8: desc float32[m][128] (or uint8 when src_is_u8) -> desc8 int8[m][128]
 * (value - 128, 16-byte aligned), norm int32[m] (squared norm of the shifted vector); *bad (device
 * int32) is set to 1 when a component is not an integer in [0, 255] (results are then meaningless). */
int sfm_descriptors_to_i8(const void* desc, int32_t src_is_u8, int64_t m, int8_t* desc8, int32_t* norm, int32_t* bad, void* stream);

/* Bytes of device workspace sfm_knn2_ratio_i8 needs for nq queries against nt train descriptors. */
int64_t sfm_knn2_workspace_bytes(int32_t nq, int32_t nt);

/* For every query: train_idx int32[nq] (nearest train descriptor), dist1/dist2 float32[nq] (distance
 * to the nearest and to the second nearest), good uint8[nq] = (double)dist1 < ratio * (double)dist2,
 * *good_count (device int32) = number of good queries.  nt >= 2 (knnMatch with k=2 yields pairs only then). */
int sfm_knn2_ratio_i8(const int8_t* q8, const int32_t* qnorm, int32_t nq, const int8_t* t8, const int32_t* tnorm, int32_t nt,
                      double ratio, void* workspace, int64_t workspace_bytes, int32_t* train_idx, float* dist1, float* dist2,
                      uint8_t* good, int32_t* good_count, void* stream);

/* ---- SIFT detection and description (feature front-end next to the path) ---------------------
 * Replaces `cv2.SIFT_create().detectAndCompute(gray, None)` (src/extract_sift.py:104-114, src/utils.py:78-80): Lowe's
 * SIFT with OpenCV's default parameters (doubled base image, 3 layers per octave, sigma 1.6, contrast threshold 0.04,
 * edge threshold 10, descriptors 128 integers in [0, 255]).  OpenCV is not part of the reference's tree: results are
 * checked against a restatement of the published algorithm, not against cv2 itself.
 *
 * sfm_sift_workspace_bytes: device bytes for an image of w x h pixels (0 when the size is not supported).
 * sfm_sift_detect: gray uint8[h][w] (device).  Builds the scale space in the workspace and writes the oriented
 *   keypoints, in no particular order and BEFORE duplicate removal, as keypoints float32[capacity][6] =
 *   (x, y, size, angle in degrees, response, packed octave as integer bits; x, y, size in pixels of the DOUBLED
 *   image, as OpenCV holds them at that stage).  *count (host) = number found; SFM_ETOOLARGE when it exceeds capacity
 *   (the first `capacity` are valid).  Synchronises the stream.
 * sfm_sift_describe: descriptors uint8[n][128] for n final keypoints (device float32[n][6], coordinates of the input
 *   image, octave field repacked for firstOctave = -1, i.e. what OpenCV's KeyPoint holds after detection), from the
 *   scale space a preceding sfm_sift_detect of the same image left in the workspace.
 * sfm_sift_gaussian_taps (host only): the float32 taps of the blur for one sigma (ksize <= 64), for tests.
 * sfm_sift_octaves (host only): number of octaves and, per octave, width, height and the offset (in floats) of its
 *   six Gaussian images in the workspace. */
size_t sfm_sift_workspace_bytes(int32_t w, int32_t h);
int sfm_sift_detect(const uint8_t* gray, int32_t w, int32_t h, void* workspace, size_t workspace_bytes, float* keypoints, int32_t capacity,
                    int32_t* count, void* stream);
int sfm_sift_describe(const void* workspace, int32_t w, int32_t h, const float* keypoints, int32_t n, uint8_t* descriptors, void* stream);
int sfm_sift_gaussian_taps(double sigma, float* taps, int32_t* ksize);
int sfm_sift_octaves(int32_t w, int32_t h, int32_t* noct, int32_t* widths, int32_t* heights, int64_t* gauss_offsets);

#ifdef __cplusplus
}
#endif
#endif
