/* svo_hip_lockstep_c.h -- a C face of FrontendLockstep (svo_hip_lockstep.h) in libsvo_hip_host.so, for callers without a
 * C++ compiler at hand (bench.py drives it through ctypes).  Same conventions as include/svo_hip.h: int status, no
 * exception crosses the boundary, svohl_last_error() has the text (per calling thread).  One engine = one lock-step group
 * on one svoh_ctx; an engine's calls come from one thread at a time (several engines may run on several threads). */
#ifndef SVO_HIP_LOCKSTEP_C_H_
#define SVO_HIP_LOCKSTEP_C_H_

#include "../../include/svo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svohl_engine svohl_engine;
typedef struct svohl_pool svohl_pool;

/* worker threads shared by several engines (SharedPool, svo_hip_pool.h): pass the pool and a seed (the engine's first
 * stream index) to svohl_create_shared; destroy the pool after its engines */
int svohl_pool_create(int n_workers, svohl_pool** out);
/* ... or ONE pool of n_threads (the calling group's thread counts as one of them) that the engines take in turns, a phase
 * at a time (ExclusivePool): pass it to svohl_create_shared like the other kind */
int svohl_pool_create_exclusive(int n_threads, svohl_pool** out);
void svohl_pool_destroy(svohl_pool* p);

/* params_yaml: the reference's parameter file as text (the keys of svo_factory.cpp this library implements; NULL = the
 * defaults).  images_pinned != 0: the images passed to svohl_add_images live in svoh_host_alloc memory. */
int svohl_create(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* params_yaml,
                 double depth_min, double depth_mean, double depth_max, int kf_every, int n_workers, int images_pinned,
                 svohl_engine** out);
int svohl_create_shared(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* params_yaml,
                        double depth_min, double depth_mean, double depth_max, int kf_every, svohl_pool* pool, int seed, int images_pinned,
                        svohl_engine** out);
/* Streams that DIFFER (round 6; LockstepStreamOptions): one entry per stream in every array.  params_yaml[s]: stream s' parameter
 * file as text (NULL entry = defaults) -- what the streams' shared device calls take once (pyramid levels, grid, detector,
 * matcher / depth-filter switches) must agree, the call fails otherwise; depth_min_mean_max: 3 doubles per stream; kf_every,
 * min_tracked: the stream's keyframe rule.  pool may be NULL (then n_workers threads of the engine's own). */
int svohl_create_streams(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* const* params_yaml,
                         const double* depth_min_mean_max, const int* kf_every, const int* min_tracked, int n_workers, svohl_pool* pool, int seed,
                         int images_pinned, svohl_engine** out);
/* ... and a CAMERA per stream: cams / T_B_Cs hold n_streams entries (intrinsics, distortion and extrinsics may differ, width and height may
 * not: the streams' pyramids are one call).  cams[0] is the engine's camera: the one the reference's process-wide thresholds are taken from
 * (svo_hip::fixProcessWideThresholds, svo_hip_host.h). */
int svohl_create_streams_cameras(svoh_ctx* ctx, int n_streams, const svoh_camera* cams, const svoh_se3* T_B_Cs, const char* const* params_yaml,
                                 const double* depth_min_mean_max, const int* kf_every, const int* min_tracked, int n_workers, svohl_pool* pool, int seed,
                                 int images_pinned, svohl_engine** out);
void svohl_destroy(svohl_engine* e);
/* one frame of every stream (FrontendLockstep::addImages); images[s] == NULL: stream s has no frame this round; T_f_w_first:
 * n_streams poses, read for the streams whose first frame this is (may be NULL when no stream starts) */
int svohl_add_images(svohl_engine* e, const uint8_t* const* images, int pitch, const svoh_se3* T_f_w_first);
/* n_rounds calls of svohl_add_images in one: every stream gets image frame_of(k) of ONE sequence of n_frames images of
 * image_bytes each -- stream s reads ITS copy of the sequence at base + s * stream_stride (0: one shared copy, which the
 * device then serves from its caches after the first stream) -- k = k_first .. k_first + n_rounds - 1; frame_of walks 0 .. n-1, n-2 .. 1, 0, 1 ..: the
 * camera goes the path forth and back, so a run never has to restart).  round_ms (may be NULL): 7 doubles per round as
 * svohl_last_round.  For drivers whose own loop is slow (an interpreter holding a global lock between calls). */
int svohl_run_sequence(svohl_engine* e, const uint8_t* base, size_t image_bytes, size_t stream_stride, int n_frames, int pitch, long k_first, int n_rounds,
                       const svoh_se3* T_f_w_first, double* round_ms);
/* ... with a schedule per stream: stream s has a frame in round k iff k >= phase[s] and (k - phase[s]) % every[s] == 0; that frame is
 * its j-th, j = (k - phase[s]) / every[s], and shows image pingpong(start[s] + step[s] * j) of its copy of the sequence (pingpong: the
 * walk over 0 .. n_frames - 1 that turns round at both ends; step may be negative or larger than one).  T_f_w_first: n_streams poses,
 * read for a stream in the round of its first frame.  *frames_done (may be NULL): frames taken by all streams in these rounds. */
int svohl_run_schedule(svohl_engine* e, const uint8_t* base, size_t image_bytes, size_t stream_stride, int n_frames, int pitch, long k_first, int n_rounds,
                       const int* start, const int* step, const int* every, const int* phase, const svoh_se3* T_f_w_first, double* round_ms, long* frames_done);
int svohl_pose(svohl_engine* e, int stream, svoh_se3* T_f_w);
/* pyramid, align, reproject, pose, seeds, keyframe, total of the last round (ms) and its device calls */
int svohl_last_round(svohl_engine* e, double times_ms[7], int* device_calls);
/* the rows of `stream` completed since the last call: 7 integers each (frame, is_kf, n_aligned, n_reprojected,
 * n_after_pose_opt, n_seeds_updated, n_converged_seeds); at most max_rows are handed out, *n_rows = how many */
int svohl_completed_rows(svohl_engine* e, int stream, int max_rows, int64_t* rows, int* n_rows);
int svohl_finish(svohl_engine* e);
/* where the rounds' time went since the engine was made: up to max_phases sums (ms) of the group thread's phases, in the
 * order of svohl_phase_name(0 ..); *n_phases = how many there are */
int svohl_phase_times(svohl_engine* e, int max_phases, double* ms, int* n_phases);
const char* svohl_phase_name(int k);
const char* svohl_last_error(void);

/* ---- the stereo engine (FrontendLockstepStereo, svo_hip_lockstep_stereo.h; round 6): same conventions, svohs_last_error() has the text.
 * cams / T_B_C: the rig's two cameras.  svohs_run_sequence: n_rounds calls of addPairs in one -- stream s reads ITS copy of a sequence of n_pairs
 * stereo pairs (images left(0) right(0) left(1) right(1) ... of image_bytes each) at base + s * stream_stride, walking it forth and back
 * (pair_of(k) = 0 .. n-1, n-2 .. 1, 0 ...); prior_forward: 4 doubles (qw qx qy qz of R_imu(f)_imu(f-1)) per pair, or NULL; round_ms: one per round. */
typedef struct svohs_engine svohs_engine;
int svohs_create(svoh_ctx* ctx, int n_streams, const svoh_camera* cams, const svoh_se3* T_B_C, const char* params_yaml, int kf_every, double lambda_rot, int n_workers,
                 int images_pinned, svohs_engine** out);
/* ... with a RIG per stream: cams / T_B_C hold 2 x n_streams entries (stream s: [2 s], [2 s + 1]); one image size for all (StereoLockstepOptions::per_stream_rig);
 * stream 0's first camera is the one the reference's process-wide thresholds are taken from */
int svohs_create_rigs(svoh_ctx* ctx, int n_streams, const svoh_camera* cams, const svoh_se3* T_B_C, const char* params_yaml, int kf_every, double lambda_rot, int n_workers,
                      int images_pinned, svohs_engine** out);
void svohs_destroy(svohs_engine* e);
int svohs_run_sequence(svohs_engine* e, const uint8_t* base, size_t image_bytes, size_t stream_stride, int n_pairs, int pitch, long k_first, int n_rounds,
                       const svoh_se3* T_imu_world_first, const double* prior_forward, double* round_ms);
int svohs_pose(svohs_engine* e, int stream, svoh_se3* T_imu_world);
int svohs_phase_times(svohs_engine* e, double* ms /* 8: pyramids, finish seeds, align, reproject, pose, structure, keyframes, seed updates */);
int svohs_finish(svohs_engine* e);
const char* svohs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
