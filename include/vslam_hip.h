/*
 * vslam_hip.h -- C ABI of the MI355X (gfx950) hot path of scomup/visual_slam:
 * bundle-adjustment linearisation + per-landmark Schur elimination + LM, and the 256-d SuperPoint
 * descriptor matchers.  Plain pointers and sizes only; every pointer argument may be a host or a
 * device pointer unless stated otherwise (copies use hipMemcpyDefault).  All entry points return
 * VS_OK (0) or a negative vs_status; vs_last_error_string() describes the last failure on the
 * calling thread.  Nothing throws or exits across this boundary.  A handle is not thread-safe: use
 * one handle per (thread, GPU).  Calls are stream-ordered on the handle's stream and block until
 * their host-visible outputs are complete.
 *
 * Reference paths cited below are relative to the reference tree (scomup/visual_slam).
 */
#ifndef VSLAM_HIP_H_
#define VSLAM_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    VS_OK = 0,
    VS_ERR_INVALID = -1,      /* bad argument (null pointer, index out of range, duplicate edge) */
    VS_ERR_HIP = -2,          /* a HIP runtime call failed (no device, launch failure, ...) */
    VS_ERR_NOMEM = -3,
    VS_ERR_STATE = -4,        /* call order violated (e.g. solve before observations are set) */
    VS_ERR_NUMERIC = -5,      /* reduced camera system not positive definite */
    VS_ERR_UNSUPPORTED = -6   /* size beyond what this build's reduced-system solver handles */
} vs_status;

const char *vs_last_error_string(void);
/* Library version string and the GPU architecture the kernels were compiled for ("gfx950"). */
const char *vs_version(void);

/* ------------------------------------------------------------------------------------------
 * Bundle adjustment.  Replaces what the reference does between Optimizer::addReprojectionEdges
 * and ceres::Solve (src/core/optimizer.cc:60-145, src/core/tracking.cc:97-110, 139-152) with the
 * residual functor of src/core/reprojection_error.h:22-51.
 * Parameter layout: camera c = translation t[3c..3c+2] and (roll,pitch,yaw) rpy[3c..3c+2]
 * (PoseData, src/core/optimizer.h:17-22); point j = xyz[3j..3j+2] (optimizer.h:42).
 * Camera-block parameter order everywhere: (t0,t1,t2,roll,pitch,yaw).
 * ------------------------------------------------------------------------------------------ */
typedef struct vs_ba *vs_ba_handle;

typedef struct {
    int32_t max_num_iterations;        /* tracking.cc:102 sets 100 */
    double initial_trust_region_radius;/* Ceres default 1e4 */
    double max_trust_region_radius;    /* 1e16 */
    double min_trust_region_radius;    /* 1e-32 */
    double min_relative_decrease;      /* 1e-3 */
    double min_lm_diagonal;            /* 1e-6 */
    double max_lm_diagonal;            /* 1e32 */
    double function_tolerance;         /* 1e-6 */
    double gradient_tolerance;         /* 1e-10 */
    double parameter_tolerance;        /* 1e-8 */
    int32_t max_consecutive_invalid_steps; /* 5 */
} vs_lm_options;

typedef struct {
    int32_t iterations;
    int32_t successful_steps;
    int32_t termination;               /* 0 convergence, 1 no convergence, 2 failure */
    double initial_cost;
    double final_cost;
    double final_radius;
} vs_lm_summary;

/* Fills *o with the options the reference runs with: max_num_iterations = 100
 * (tracking.cc:102,144), everything else the Ceres defaults ceres::Solver::Options() carries. */
void vs_lm_default_options(vs_lm_options *o);

/* Creates a problem of fixed size on the current HIP device.  fx,fy,cx,cy: the four entries the
 * reference puts into K (tracking.cc:58-62), widened from float as reprojection_error.h:44 does. */
int vs_ba_create(vs_ba_handle *h, int32_t n_cameras, int32_t n_points, int64_t n_observations,
                 double fx, double fy, double cx, double cy);
int vs_ba_destroy(vs_ba_handle h);
/* hipStream_t to launch on (default: the null stream). */
int vs_ba_set_stream(vs_ba_handle h, void *hip_stream);

/* Optimizer::addPose (optimizer.cc:14-17): t, rpy are n_cameras x 3 doubles. */
int vs_ba_set_cameras(vs_ba_handle h, const double *t, const double *rpy);
/* Optimizer::addPoints (optimizer.cc:19-28): n_points x 3 doubles. */
int vs_ba_set_points(vs_ba_handle h, const double *xyz);
/* The residual blocks addReprojectionEdges adds (optimizer.cc:102-143), one per entry, in the
 * caller's order: camera index, point index, keypoint (u,v).  HOST pointers.  An edge may appear
 * at most once per (camera, point).  Builds the landmark-major / camera-major / camera-pair
 * index structures used by the kernels; call again only when the edge set changes. */
int vs_ba_set_observations(vs_ba_handle h, const int32_t *cam, const int32_t *pt, const double *uv);
/* Multi-GPU: as vs_ba_set_observations, but this handle linearises only the landmarks of shard
 * `rank` of `world` (contiguous landmark ranges balanced by observation count) while keeping the
 * global camera-pair block pattern, so vs_ba_reduce_buffer() has identical layout on every rank. */
int vs_ba_set_observations_sharded(vs_ba_handle h, const int32_t *cam, const int32_t *pt,
                                   const double *uv, int32_t rank, int32_t world);
/* SetParameterBlockConstant on every point (fix_points != 0; optimizer.cc:134-137, the
 * fix_camera_or_point = true mode of Tracking::solveCamera).  Cameras are always free
 * (the code freezing them is commented out, optimizer.cc:90-100). */
int vs_ba_set_fixed(vs_ba_handle h, int32_t fix_points);
/* HuberLoss(a) on the FIRST observation only, none on the rest (optimizer.cc:69,128);
 * a <= 0 disables it (default). */
int vs_ba_set_huber_first(vs_ba_handle h, double a);

/* (a) residual + Jacobian evaluation at the current parameters.  Any output may be NULL.
 * cost = 0.5 * sum rho(|r|^2); res: n_obs x 2; Jc: n_obs x 12 (2x6 row-major); Jp: n_obs x 6
 * (2x3 row-major), all in the caller's observation order, loss-corrected like Ceres does. */
int vs_ba_evaluate(vs_ba_handle h, double *cost, double *res, double *Jc, double *Jp);

/* (a)+(b) fused: linearise at the current parameters and form the reduced camera system
 *   S = U + diag(damp_c) - sum_j W_j (V_j + diag(damp_p_j))^-1 W_j^T,
 *   rhs = -g_c + sum_j W_j (V_j + diag(damp_p_j))^-1 g_p_j
 * on the device (J is never materialised).  damp_c: 6*n_cameras, damp_p: 3*n_points, or NULL = 0;
 * device pointers are used in place (no copy), host pointers are uploaded first.
 * This is the step the throughput metric times.  Results stay on the device. */
int vs_ba_linearize(vs_ba_handle h, const double *damp_c, const double *damp_p);
/* Stored upper-triangular 6x6 blocks of S: the n_cameras diagonal blocks first, then the
 * off-diagonal blocks (row < col) sorted by (row, col).  Pattern is fixed by the edge set. */
int vs_ba_block_count(vs_ba_handle h, int64_t *n_blocks);
int vs_ba_block_index(vs_ba_handle h, int32_t *block_row, int32_t *block_col);
/* Copies out what the last vs_ba_linearize produced.  S_blocks: n_blocks x 36 (6x6 row-major);
 * rhs: 6*n_cameras; cost: scalar.  Any may be NULL. */
int vs_ba_get_reduced_system(vs_ba_handle h, double *S_blocks, double *rhs, double *cost);
/* Per-landmark products of the last vs_ba_linearize: Vinv n_points x 6 (xx,xy,xz,yy,yz,zz of the
 * damped inverse), gp n_points x 3.  Any may be NULL. */
int vs_ba_get_landmark_blocks(vs_ba_handle h, double *Vinv6, double *gp);
/* Device buffer holding [S diagonal blocks | S off-diagonal blocks | rhs | diag(U) | g_c | cost]
 * as doubles, the quantity landmark shards sum across GPUs (all-reduce) before solving. */
int vs_ba_reduce_buffer(vs_ba_handle h, void **device_ptr, int64_t *n_doubles);

/* One Levenberg-Marquardt step at the current parameters with the given trust-region radius:
 * Jacobi scaling (fixed at the handle's first step), LM diagonal, Schur, reduced solve,
 * back-substitution.  dcam: n_cameras x 6, dpt: n_points x 3 (zeros when points are fixed),
 * model_cost_change = -(J d)^T (r + J d / 2).  Parameters are not modified. */
int vs_ba_lm_step(vs_ba_handle h, double radius, double *dcam, double *dpt,
                  double *model_cost_change);
/* ceres::Solve replacement (tracking.cc:110,152): runs LM to termination and updates the
 * handle's parameters in place. */
int vs_ba_solve(vs_ba_handle h, const vs_lm_options *opts, vs_lm_summary *summary);
/* Multi-GPU hook for callers that bring their own collective (e.g. torch.distributed): called by
 * vs_ba_lm_step / vs_ba_solve wherever a rank-global sum is needed (the reduce buffer, the step
 * scalars, one slot per rank for maxima, the final gather of the points); must sum `n` doubles at
 * device pointer `buf` across ranks, in place, and return 0.  The handle's stream is synchronised
 * before each call.  Every rank must install a hook (or a communicator) before vs_ba_solve. */
typedef int (*vs_allreduce_fn)(void *user, void *device_buf, int64_t n_doubles);
int vs_ba_set_allreduce(vs_ba_handle h, vs_allreduce_fn fn, void *user);

/* Native multi-GPU path (the reference is single-process; this is the landmark-sharded form of the
 * Optimizer behind src/core/optimizer.h:24-36): one process or thread per GPU, each with a
 * communicator over the same unique id.  Rank 0 calls vs_comm_unique_id and hands the
 * VS_COMM_ID_BYTES bytes to the other ranks by any means (a file, MPI, a socket); then every rank
 * calls vs_comm_create with its GPU current (a collective call: it returns once all ranks have
 * joined).  A handle given a communicator (vs_ba_set_comm) sums its reduce buffer across ranks with
 * an RCCL all-reduce enqueued on the handle's stream -- no host synchronisation, no callback -- and
 * takes every LM decision (step scalars, gradient norm) from rank-global values, so all ranks
 * iterate in lockstep and end with identical cameras; vs_ba_solve finally gathers the landmark
 * shards so that every rank holds all points.  librccl is opened at run time on first use. */
#define VS_COMM_ID_BYTES 128
typedef struct vs_comm *vs_comm_handle;
int vs_comm_unique_id(void *id_bytes);
int vs_comm_create(vs_comm_handle *c, int32_t rank, int32_t world, const void *id_bytes);
int vs_comm_destroy(vs_comm_handle c);
/* In-place sum of n doubles at a device pointer across the ranks, enqueued on hip_stream. */
int vs_comm_allreduce_sum(vs_comm_handle c, double *device_buf, int64_t n, void *hip_stream);
/* Every rank's n doubles at device_send end up at device_recv + rank * n on every rank (the
 * collective vs_ba_exchange_reduced is built on), enqueued on hip_stream. */
int vs_comm_allgather(vs_comm_handle c, const double *device_send, double *device_recv, int64_t n, void *hip_stream);
/* Combines the shards' partial reduce buffers (no counterpart in the reference, which is one
 * process): after it every rank's vs_ba_reduce_buffer() holds the sum over the ranks.  The
 * landmarks of a shard see one contiguous range of cameras, so a rank sends only the rows of that
 * range (diagonal blocks, the off-diagonal blocks of those rows, right-hand side, diag(U), g_c,
 * the cost): one all-gather of these segments through the communicator (RCCL, on the handle's
 * stream) or the gather callback, then every rank adds the pieces in rank order.  With neither, the
 * whole buffer goes through the all-reduce callback.  vs_ba_lm_step / vs_ba_solve call it
 * themselves; it is public so that a caller who drives vs_ba_linearize can finish the sum. */
int vs_ba_exchange_reduced(vs_ba_handle h);
/* Doubles this rank sends in vs_ba_exchange_reduced and doubles every rank receives (ranks x the
 * longest segment); zeros for an unsharded handle. */
int vs_ba_exchange_size(vs_ba_handle h, int64_t *send_doubles, int64_t *total_doubles);
/* Gather callback for callers that bring their own collective: every rank's n doubles at device
 * pointer `send` must arrive at `recv` + rank * n on every rank; returns 0.  The handle's stream is
 * synchronised before the call. */
typedef int (*vs_allgather_fn)(void *user, const void *send, void *recv, int64_t n_doubles);
int vs_ba_set_allgather(vs_ba_handle h, vs_allgather_fn fn, void *user);
/* NULL detaches.  The handle must have been given this rank's shard
 * (vs_ba_set_observations_sharded with the communicator's rank and world). */
int vs_ba_set_comm(vs_ba_handle h, vs_comm_handle c);

/* Measurement aid (no counterpart in the reference): runs vs_ba_linearize `iters` times with HIP
 * events between its kernels, on the handle's stream, and returns average milliseconds:
 * avg_ms[0..4] = camera prep, landmark pass, camera (diagonal) pass, camera-pair (off-diagonal)
 * pass, finalize; avg_ms[5] = first event to last. */
int vs_ba_profile_linearize(vs_ba_handle h, const double *damp_c, const double *damp_p,
                            int32_t iters, float *avg_ms);

/* Optimizer::getPose / getPoints payloads (optimizer.cc:51-59,187-190). */
int vs_ba_get_cameras(vs_ba_handle h, double *t, double *rpy);
int vs_ba_get_points(vs_ba_handle h, double *xyz);

/* ------------------------------------------------------------------------------------------
 * Descriptor matching.  Replaces SuperpointFrontend::twoWayMatching
 * (src/core/superpoint_frontend.cc:163-211) and getStereoMatching (:110-161):
 *   d(i,j) = sqrtf(2 - 2*clamp((float)dot(desc0_i, desc1_j), -1, 1)), dot accumulated in double;
 *   idx0[i] = first argmin_j, idx1[j] = first argmin_i; keep (i, idx0[i], d) iff
 *   idx1[idx0[i]] == i and d < nn_thresh; output ordered by i.
 * Descriptors are n x 256 row-major fp32 (superpoint_frontend.cc:97-107).
 * ratio < 1 additionally requires d < ratio * (second smallest d of row i); ratio >= 1 = off
 * (the reference has no ratio test).
 * ------------------------------------------------------------------------------------------ */
#define VS_DESC_DIM 256

typedef struct vs_matcher *vs_matcher_handle;

/* Batched matcher with device-resident buffers for up to max_pairs frame pairs of up to
 * max_n0 x max_n1 descriptors each. */
int vs_matcher_create(vs_matcher_handle *m, int32_t max_pairs, int32_t max_n0, int32_t max_n1);
int vs_matcher_destroy(vs_matcher_handle m);
int vs_matcher_set_stream(vs_matcher_handle m, void *hip_stream);
/* Loads pair `pair`.  xy0/xy1 (n x 2 int32 pixel coordinates, cv::Point) may both be NULL
 * (twoWayMatching) or both non-NULL (getStereoMatching's mask: entries with |y0-y1| > 20 or
 * x0 < x1 keep distance 1.0 and still compete, superpoint_frontend.cc:117,126-129). */
int vs_matcher_set_pair(vs_matcher_handle m, int32_t pair, const float *desc0, int32_t n0,
                        const float *desc1, int32_t n1, const int32_t *xy0, const int32_t *xy1);
/* Loads pairs [first_pair, first_pair + n_pairs) with one call: every pair is n0 x n1, pair p's
 * descriptors start at desc0 + p * stride0 and desc1 + p * stride1 (strides in floats, at least
 * n * 256), its keypoints (if given) at xy0 + p * 2 * n0 and xy1 + p * 2 * n1.  The batched form of
 * the call above (the reference matches one cv::Mat pair per call, superpoint_frontend.cc:163-170);
 * it is what a caller uses whose descriptors come out of the network as one (pairs, n, 256) array.
 * Host or device pointers in both calls.  With DEVICE pointers the descriptors are read by a kernel
 * enqueued on the matcher's stream and the call returns without waiting for it: the arrays must
 * stay untouched until work enqueued later on that stream has started (or the stream has been
 * synchronised).  With host pointers the call returns when the caller's arrays are free again. */
int vs_matcher_set_pairs(vs_matcher_handle m, int32_t first_pair, int32_t n_pairs, const float *desc0,
                         int64_t stride0, int32_t n0, const float *desc1, int64_t stride1, int32_t n1,
                         const int32_t *xy0, const int32_t *xy1);
/* Matches pairs [0, n_pairs) on the device.  Results stay on the device. */
int vs_matcher_run(vs_matcher_handle m, int32_t n_pairs, float nn_thresh, float ratio);
/* Copies out pair `pair`'s matches (each array sized >= n0 of that pair). */
int vs_matcher_get_matches(vs_matcher_handle m, int32_t pair, int32_t *out_i, int32_t *out_j,
                           float *out_d, int32_t *out_n);
/* Row / column nearest neighbours of the last run (idx0: n0, idx1: n1). */
int vs_matcher_get_nn(vs_matcher_handle m, int32_t pair, int32_t *idx0, int32_t *idx1);
/* Number of rows+columns whose nearest neighbour had to be re-decided by the exact fp64 scan
 * in the last run (diagnostic). */
int vs_matcher_get_rescans(vs_matcher_handle m, int64_t *n_rescanned);

/* One frame pair split over several GPUs by QUERY ROWS (no counterpart in the reference, whose
 * matcher is one loop nest, superpoint_frontend.cc:173-209; this is its row-sharded form): every
 * rank loads its slice of desc0 (rows row0 .. row0+n0-1 of the whole frame) and ALL of desc1 with
 * vs_matcher_set_pair, then calls vs_matcher_run_sharded instead of vs_matcher_run.  The row
 * nearest neighbours are local; the column nearest neighbours (the mutual check,
 * superpoint_frontend.cc:197-209) are combined across ranks as the minimum of packed
 * (distance bits, global row) keys -- equal distances resolve to the smaller row, the reference's
 * first minimum -- by an RCCL all-reduce on the matcher's stream (vs_matcher_set_comm) or by a
 * callback (vs_matcher_set_min_allreduce: must leave the element-wise minimum over ranks of the n
 * uint64 at device_buf in place and return 0; the stream is synchronised before the call).
 * vs_matcher_get_matches then returns this rank's matches with GLOBAL row numbers, ordered by
 * row; the ranks' lists concatenated in rank order are the whole pair's matches. */
typedef int (*vs_allreduce_min_u64_fn)(void *user, void *device_buf, int64_t n);
int vs_matcher_set_comm(vs_matcher_handle m, vs_comm_handle c);
int vs_matcher_set_min_allreduce(vs_matcher_handle m, vs_allreduce_min_u64_fn fn, void *user);
int vs_matcher_run_sharded(vs_matcher_handle m, int32_t n_pairs, int32_t row0, float nn_thresh, float ratio);

/* Measurement aid (no counterpart in the reference): vs_matcher_run `iters` times with HIP events
 * between its stages; avg_ms[0..3] = MFMA dot + top-2 kernel, partial merge, exact distance and
 * rescan kernels, mutual-check compaction; avg_ms[4] = first event to last. */
int vs_matcher_profile_run(vs_matcher_handle m, int32_t n_pairs, float nn_thresh, float ratio,
                           int32_t iters, float *avg_ms);

/* Measurement aid (no counterpart in the reference): the rate at which the current device issues
 * v_mfma_f32_32x32x16_f16 back to back from registers, one wave per SIMD on every CU, for about
 * `millis` milliseconds on random (random_operands != 0) or all-zero operands; *tflops = the
 * chip-wide fp16 rate of that loop.  On random operands the chip runs into its power limit and the
 * clock drops, so this is the ceiling a kernel on real data can reach, below the spec peak. */
int vs_debug_mfma_rate(int32_t millis, int32_t random_operands, float *tflops);

/* One-shot forms with the reference's call shape (host or device pointers). */
int vs_match_two_way(const float *desc0, int32_t n0, const float *desc1, int32_t n1,
                     float nn_thresh, float ratio, int32_t *out_i, int32_t *out_j, float *out_d,
                     int32_t *out_n);
int vs_match_stereo(const float *desc0, const int32_t *xy0, int32_t n0, const float *desc1,
                    const int32_t *xy1, int32_t n1, float nn_thresh, float ratio, int32_t *out_i,
                    int32_t *out_j, float *out_d, int32_t *out_n);

/* Allocation-free form of vs_match_two_way.  It has no counterpart in the reference: it is for
 * callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  Same results as vs_match_two_way, bit for bit.  All of
 * desc0, desc1, out_i, out_j, out_d (each >= n0 entries) and out_n are DEVICE pointers; the
 * descriptors are read in place.  n0, n1 <= VS_MATCH_NOALLOC_MAX_N, larger sizes return
 * VS_ERR_INVALID.  The kernels are enqueued on hip_stream (NULL = the null stream) and the call
 * returns without waiting: the outputs are complete when the stream reaches that point.  The
 * intermediate buffers live in one statically allocated device array (about 3.2 MB per GPU, held
 * until the process ends) that every call on a GPU shares: two calls that overlap on one GPU
 * (different streams or threads) race; calls on one stream are safe. */
#define VS_MATCH_NOALLOC_MAX_N 2048
int vs_match_two_way_noalloc(const float *desc0, int32_t n0, const float *desc1, int32_t n1,
                             float nn_thresh, float ratio, int32_t *out_i, int32_t *out_j,
                             float *out_d, int32_t *out_n, void *hip_stream);

/* ------------------------------------------------------------------------------------------
 * Per-frame tracking helpers beside the matcher.  Poses are transform::Rigid3f as 7 floats:
 * translation x, y, z then rotation quaternion w, x, y, z (src/transform/rigid_transform.h:101-103);
 * K4 = fx, fy, cx, cy (config/config.yaml:12-16).  Pixel coordinates are cv::Point (n x 2 int32).
 * Array and pose/K4 arguments are host or device pointers (the small ones are copied to the host
 * before use); the scalar counts n_removed / n_updated are HOST pointers.  Every call blocks until
 * its results are in the output arrays.
 * ------------------------------------------------------------------------------------------ */

/* Tracking::reprojection (src/core/tracking.cc:70-78) for n world points: float pinhole
 * projection of Tcw * Pw along +Y, truncated to integer pixels. */
int vs_reproject_points(const float *pose, const float *K4, const float *xyz, int32_t n,
                        int32_t *xy);

/* Tracking::searchProjection (src/core/tracking.cc:453-496): every tracked point i (world point
 * + descriptor), in index order, takes the keypoint j with the smallest descriptor distance
 * among those not yet claimed and within `radius` pixels of its reprojection (first minimum in
 * j order), if that distance is < accept (0.8 in the reference) and the reprojection lies inside
 * [0, 2cx] x [0, 2cy].  claimed (n_keys bytes, in/out) marks keypoints that own a tracked point
 * (Frame::tps() != nullptr); tracks[i] = j or -1. */
int vs_search_projection(const float *pose, const float *K4, const float *tp_xyz,
                         const float *tp_desc, int32_t n_tp, const int32_t *key_xy,
                         const float *key_desc, int32_t n_keys, float radius, double accept,
                         uint8_t *claimed, int32_t *tracks);

/* Tracking::removeOutlier for one frame (src/core/tracking.cc:357-372): tracks[i] = -1 where the
 * tracked keypoint and the integer reprojection are more than max_dist (20) pixels apart. */
int vs_remove_outliers(const float *pose, const float *K4, const float *tp_xyz, int32_t n_tp,
                       const int32_t *key_xy, int32_t n_keys, double max_dist, int32_t *tracks,
                       int32_t *n_removed);

/* Frame::computePoint3d (src/core/frame.cc:45-86) for all n0 left keypoints: nearest right
 * keypoint among those with |yl - yr| <= 20 and xl >= xr, disparity > 1, depth = bf / disparity,
 * back-projection through inv(K) and the inverse pose.  good[i] = 1 and xyz[3i..3i+2] = Pw where
 * the reference returns true; minloc (optional) = chosen right keypoint or -1. */
int vs_compute_points3d(const float *desc0, const int32_t *xy0, int32_t n0, const float *desc1,
                        const int32_t *xy1, int32_t n1, const float *K4, float baseline_fx,
                        const float *pose, uint8_t *good, float *xyz, int32_t *minloc);

/* Tracking::updateTrackedPoints (src/core/tracking.cc:378-451): two-view linear triangulation of
 * every tracked point seen in both the oldest (pose1, key_xy1, tracks1) and the newest (pose2,
 * key_xy2, tracks2) frame of the window.  tracks*[i] = keypoint index or -1.  A point is
 * re-triangulated when the rays' parallax cosine lies in [0, 0.9998] and the new point has depth
 * (camera +Y) > 0.2 in both frames; then xyz[3i..3i+2] is overwritten and updated[i] = 1
 * (updated may be NULL).  The float 4 x 4 system is built as the reference builds it; its null
 * vector (JacobiSVD::matrixV().col(3) in the reference, float) is computed in double. */
int vs_triangulate_tracks(const float *pose1, const float *pose2, const float *K4,
                          const int32_t *key_xy1, int32_t n_keys1, const int32_t *tracks1,
                          const int32_t *key_xy2, int32_t n_keys2, const int32_t *tracks2,
                          int32_t n_tp, float *xyz, uint8_t *updated, int32_t *n_updated);

#ifdef __cplusplus
}
#endif
#endif /* VSLAM_HIP_H_ */
