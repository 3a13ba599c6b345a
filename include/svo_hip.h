/*
 * svo_hip.h -- C ABI of the MI355X-native SVO direct front end (libsvo_hip.so).
 *
 * This is the drop-in boundary for the per-frame direct front end of
 * Jianxff/svo_pro_universal.  Host C++ (the adapter that subclasses the
 * reference's SparseImgAlignBase etc.) calls these entry
 * points; everything behind them is hand-written HIP for gfx950.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, POD structs only.
 *   - every function returns an int status: SVOH_OK (0) or a negative
 *     svoh_status; no exception or abort crosses the ABI.  The text of the last
 *     error of a context is available from svoh_last_error_string().
 *   - the caller owns all host buffers; the context owns device buffers.
 *   - frames (image pyramids) live on the device and are referenced by 64-bit
 *     handles returned from svoh_upload_pyramid()/svoh_build_pyramid().
 *   - matrices are column-major (Eigen default in the reference): a "2xN"
 *     array stores column i at [2*i, 2*i+1].
 *   - a context is single-threaded (one hipStream_t); use one context per
 *     calling thread, as the reference uses one Matcher per thread.
 *   - pointers in problem structs are host pointers unless the struct's
 *     mem_space field says SVOH_MEM_DEVICE.
 *
 * Each entry point cites (file:line, relative to the reference tree) the
 * reference interface it replaces.
 */
#ifndef SVO_HIP_H_
#define SVO_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5 -> 6): svoh_frame_view.features, svoh_align_camera.pos_seed_unit, svoh_feature_batch.feature_index changed
 * three struct layouts; _capi.load() and the host layer's constructors refuse a library of another version */
#define SVOH_ABI_VERSION 2
#define SVOH_MAX_LEVELS 8
#define SVOH_MAX_CAMS 4

typedef enum svoh_status {
  SVOH_OK = 0,
  SVOH_ERR_INVALID_ARGUMENT = -1,
  SVOH_ERR_HIP = -2,           /* a HIP runtime call failed (see error string) */
  SVOH_ERR_OUT_OF_MEMORY = -3,
  SVOH_ERR_BAD_HANDLE = -4,
  SVOH_ERR_UNSUPPORTED = -5,   /* e.g. patch size / camera model not built */
  SVOH_ERR_NO_DEVICE = -6
} svoh_status;

typedef enum svoh_mem_space {
  SVOH_MEM_HOST = 0,
  SVOH_MEM_DEVICE = 1,
  /* svoh_feature_batch only: the arrays are the context's own page-locked staging block, handed out by svoh_matcher_stage
   * and filled in place by the caller (by several host threads at once, if it likes): nothing is copied on the host,
   * neither on the way in nor on the way out */
  SVOH_MEM_STAGED = 2,
  /* images only (svoh_build_pyramid_multi): page-locked host memory of svoh_host_alloc, read by the device directly */
  SVOH_MEM_HOST_PINNED = 3
} svoh_mem_space;

typedef struct svoh_ctx svoh_ctx; /* opaque */
typedef uint64_t svoh_frame_t;    /* 0 is never a valid handle */

/* ---- context ---------------------------------------------------------- */

int svoh_abi_version(void);
/* device: HIP device ordinal.  Fails with SVOH_ERR_NO_DEVICE when no GPU. */
int svoh_create(int device, svoh_ctx** out_ctx);
int svoh_destroy(svoh_ctx* ctx);
/* never NULL; valid until the next call on ctx (ctx may be NULL: global msg) */
const char* svoh_last_error_string(const svoh_ctx* ctx);
/* blocks until all work queued on the context's stream is complete */
int svoh_synchronize(svoh_ctx* ctx);
/* the context's hipStream_t as an opaque pointer (for event timing) */
void* svoh_stream(svoh_ctx* ctx);

/* ---- geometry PODs ---------------------------------------------------- */

/* Rigid transform as minkindr's QuatTransformation: unit quaternion (w,x,y,z)
 * + translation.  (3rd/minkindr/include/kindr/minimal/quat-transformation.h) */
typedef struct svoh_se3 {
  double q[4]; /* w, x, y, z */
  double t[3];
} svoh_se3;

typedef enum svoh_distortion {
  SVOH_DISTORTION_NONE = 0,   /* vk::cameras::NoDistortion */
  SVOH_DISTORTION_RADTAN = 1  /* vk::cameras::RadialTangentialDistortion k1 k2 p1 p2 */
} svoh_distortion;

/* vk::cameras::PinholeProjection<Distortion>
 * (src/vikit/vikit_cameras/include/vikit/cameras/implementation/pinhole_projection.hpp:10-64) */
typedef struct svoh_camera {
  double fx, fy, cx, cy;
  double d[4];          /* k1 k2 p1 p2 for RADTAN, ignored for NONE */
  int32_t distortion;   /* svoh_distortion */
  int32_t width, height;
  int32_t reserved;
} svoh_camera;

/* Diagnostic / parity entry (a-15): the camera maths every kernel uses (csrc/svoh_math.h), evaluated ON THE DEVICE for
 * n points, so that the reference's own camera tests (src/vikit/vikit_cameras/test/test_cameras.cpp:83-121, 162-173)
 * can be restated against the device code and not only against the host compilation of the same header:
 *   px[2i..]     = PinholeProjection::project3(xyz[3i..])                        pinhole_projection.hpp:44-64
 *   J[6i..]      = its 2x3 Jacobian, row-major (diag(fx,fy) * distortion.jacobian(uv) * d(uv)/d(xyz))
 *   f_back[3i..] = PinholeProjection::backProject3(px[2i..]) (radtan: five fixed-point iterations)  :30-42
 * Host pointers; J and f_back may be NULL. */
int svoh_camera_maths(svoh_ctx* ctx, const svoh_camera* cam, int n, const double* xyz, double* px, double* J,
                      double* f_back);

/* ---- frames / image pyramid  (a-0) ------------------------------------ */

typedef enum svoh_halfsample_rounding {
  /* what the reference does on x86: the SSE2 double-rounding rule when
   * cols%16==0, else the scalar truncating rule, chosen per level
   * (src/vikit/vikit_common/src/vision.cpp:73-111) */
  SVOH_HALFSAMPLE_REFERENCE = 0,
  SVOH_HALFSAMPLE_SCALAR = 1,   /* always (a+b+c+d)/4           (vision.cpp:108) */
  SVOH_HALFSAMPLE_SSE2 = 2      /* always avg(avg(a,c),avg(b,d)) (vision.cpp:19-44) */
} svoh_halfsample_rounding;

/* Upload an existing host pyramid (n_levels row-major u8 images).
 * Replaces nothing in the reference: it is how Frame::img_pyr_
 * (src/svo_common/include/svo/common/frame.h:46) gets to the device. */
int svoh_upload_pyramid(svoh_ctx* ctx, int n_levels,
                        const uint8_t* const* level_data, const int* width,
                        const int* height, const int* pitch,
                        svoh_frame_t* out_frame);

/* Build the pyramid on the device from a level-0 image.
 * Replaces frame_utils::createImgPyramid (src/svo_common/src/frame.cpp:372-386)
 * -> vk::halfSample (src/vikit/vikit_common/src/vision.cpp:73-111).
 * img may be host or device memory (mem_space).  If host_levels_out is not
 * NULL it must hold n_levels pointers to tightly packed (pitch == width)
 * host buffers; levels 1..n-1 (and 0) are copied back so that the reference's
 * host-side Frame::img_pyr_ stays bit-identical to the device copy. */
int svoh_build_pyramid(svoh_ctx* ctx, const uint8_t* img, int width, int height,
                       int pitch, int mem_space, int n_levels, int rounding,
                       uint8_t* const* host_levels_out, svoh_frame_t* out_frame);

/* Batched variant: n_images level-0 images of identical size, image i at
 * img + i*image_stride bytes.  out_frames receives n_images handles. */
int svoh_build_pyramid_batch(svoh_ctx* ctx, const uint8_t* img, size_t image_stride,
                             int n_images, int width, int height, int pitch,
                             int mem_space, int n_levels, int rounding,
                             svoh_frame_t* out_frames);

/* Page-locked host memory (hipHostMalloc): what a camera driver's image buffers should be when many streams feed one
 * GPU -- svoh_build_pyramid_multi reads such images over PCIe by itself, with no staging copy on the host. */
int svoh_host_alloc(svoh_ctx* ctx, size_t bytes, void** out);
int svoh_host_free(svoh_ctx* ctx, void* p);

/* svoh_build_pyramid_batch for images that live at separate addresses (one per camera stream): imgs[i] = level 0 of
 * image i, all of one size and pitch.  mem_space: SVOH_MEM_HOST (pageable: one copy call per image), SVOH_MEM_HOST_PINNED
 * (memory of svoh_host_alloc: ONE gather kernel reads all images, at most 256 per call), SVOH_MEM_DEVICE.  The frames
 * share one device allocation, which lives until the last of them is released. */
int svoh_build_pyramid_multi(svoh_ctx* ctx, const uint8_t* const* imgs, int n_images, int width, int height, int pitch,
                             int mem_space, int n_levels, int rounding, svoh_frame_t* out_frames);

/* The same on a SECOND stream of the context, beside whatever the context's own stream is given afterwards: how the next
 * frames' images cross PCIe while the current frames' chain is still running.  Call it when the context's stream is idle
 * -- right behind a call that waited; it waits for the stream otherwise -- and call svoh_prefetch_fence before the first
 * use of the frames: the fence makes the context's stream wait (on the device, not the host) for the prefetch. */
int svoh_build_pyramid_multi_prefetch(svoh_ctx* ctx, const uint8_t* const* imgs, int n_images, int width, int height, int pitch,
                                      int mem_space, int n_levels, int rounding, svoh_frame_t* out_frames);
int svoh_prefetch_fence(svoh_ctx* ctx);

/* Copy one level of a device frame back to the host (tightly packed). */
int svoh_download_level(svoh_ctx* ctx, svoh_frame_t frame, int level,
                        uint8_t* out, int* out_width, int* out_height);
int svoh_frame_info(svoh_ctx* ctx, svoh_frame_t frame, int* n_levels,
                    int* width0, int* height0);
/* Gives the frame up (the reference drops a Frame's img_pyr_ with the Frame).  Returns at once: launches queued before
 * the call may still read the frame, and they will -- its memory goes to a small per-context pool (8 allocations) and is
 * only written again by a later svoh_build_pyramid / svoh_upload_pyramid of the same size, on the context's stream,
 * behind them; a frame that does not fit the pool is freed (which waits for the device).  The handle is invalid from
 * here on. */
int svoh_release_frame(svoh_ctx* ctx, svoh_frame_t frame);

/* What the context holds on the device right now: live frame handles, bytes of the frame
 * slabs they keep alive, bytes of the grow-only workspaces.  A host that forgets to release
 * frames (Frame objects die, their device pyramids do not) shows up here. */
typedef struct svoh_context_stats_t {
  int64_t live_frames;
  int64_t frame_bytes;
  int64_t workspace_bytes;
} svoh_context_stats_t;
int svoh_context_stats(svoh_ctx* ctx, svoh_context_stats_t* out);

/* Tuning and diagnostic knobs (launch geometry overrides: SVOH_ALIGN_THREADS, SVOH_ALIGN_ROWS, SVOH_ALIGN_LDS, SVOH_ALIGN_CLUSTER,
 * SVOH_ALIGN_WG_PER_CU, SVOH_MATCHER_G8, SVOH_SEED_BINNING, SVOH_KLT_BLOCK, SVOH_POSE_THREADS, SVOH_COPY_KERNEL; INTEGRATION.md lists
 * them) are read from the environment once, by svoh_create.  No launch path looks at the environment.  A process that
 * changes those variables afterwards (the test-suite does, to run every kernel geometry) asks for them to be read again: */
int svoh_reload_knobs(svoh_ctx* ctx);

/* How staged blocks travel between page-locked host memory and the device: 0 = the runtime's copies (hipMemcpyAsync), 1 (the
 * default) = copy kernels for blocks of 16 KB .. 1 MB, where they are the faster call for ONE stream, 2 = copy kernels always.
 * 2 is for a process that drives SEVERAL contexts on one device (lock-step groups): a runtime copy between two kernels is a
 * hand-over between the compute queue and the copy engine, and the machine takes only ~160 k such mixed dispatches per second
 * from all streams together (tools/svoh_dispatch_rate).  Same bytes either way.  SVOH_COPY_KERNEL in the environment sets the
 * same value when the context is made. */
int svoh_set_copy_policy(svoh_ctx* ctx, int policy);

/* Kernel timing.  Off by default: bracketing a launch with an event pair costs ~9 us of every call on an MI355X
 * (tools/svoh_call_overhead), a quarter of a small call.  When on (this call, or SVOH_KERNEL_TIMING=1 in the
 * environment of svoh_create), svoh_sparse_align_last_kernel_ms / _kernel_ms_history / svoh_last_kernel_ms report the
 * device time of the launches made since; when off they fail with SVOH_ERR_INVALID_ARGUMENT. */
int svoh_set_kernel_timing(svoh_ctx* ctx, int enabled);

/* ---- sparse image alignment  (a-1 ... a-8) ---------------------------- */

/* SparseImgAlignOptions + the solver options the reference hard-wires
 * (src/svo_img_align/include/svo/img_align/sparse_img_align_base.h:37-46,
 *  src/svo_img_align/src/sparse_img_align_base.cpp:35-42). */
typedef struct svoh_align_options {
  int32_t max_level;   /* 4 */
  int32_t min_level;   /* 1 (FrameHandlerBase sets 2: svo_factory.cpp:137-138) */
  int32_t patch_size;  /* 4 (sparse_img_align.cpp:31); 8 also built */
  int32_t max_iter;    /* 10 */
  double eps;          /* 5e-4 */
  int32_t estimate_illumination_gain;
  int32_t estimate_illumination_offset;
  int32_t use_distortion_jacobian;
  int32_t robustification; /* Tukey weights, b = 4.6851 */
  double weight_scale;     /* 10 */
} svoh_align_options;

/* SparseImgAlignBase::setWeightedPrior
 * (src/svo_img_align/src/sparse_img_align_base.cpp:44-62). */
typedef struct svoh_align_prior {
  int32_t have_prior;
  int32_t reserved;
  svoh_se3 T_prior;          /* T_cur_ref_prior (imu frames) */
  double alpha_prior, beta_prior;
  double lambda_rot, lambda_trans, lambda_alpha, lambda_beta;
} svoh_align_prior;

/* One camera of the (ref, cur) frame bundles, in the reference's own SoA
 * feature layout (src/svo_common/include/svo/common/frame.h:62-73). */
typedef struct svoh_align_camera {
  svoh_frame_t ref_frame;    /* pyramids, >= max_level+1 levels */
  svoh_frame_t cur_frame;
  svoh_camera cam;           /* Frame::cam() of this camera index */
  svoh_se3 ref_T_imu_cam;    /* ref_frame.T_imu_cam() */
  svoh_se3 ref_T_cam_imu;    /* ref_frame.T_cam_imu(): MUST be the inverse of ref_T_imu_cam (as in a reference Frame, which
                                derives both from one transformation; checked to 1e-9, SVOH_ERR_INVALID_ARGUMENT otherwise).
                                Only its rotation is read: the Jacobian rows take T_cam_imu * (T_imu_cam * xyz_ref) of
                                frame.h:342-357 to be xyz_ref itself */
  svoh_se3 cur_T_cam_imu;    /* cur_frame.T_cam_imu() */
  double ref_pos[3];         /* ref_frame.pos() = T_world_cam().getPosition() */
  int32_t n_features;        /* ref_frame.num_features_ */
  int32_t mem_space;         /* where px/f/pos_world/flags live */
  const double* px;          /* 2 x n  px_vec_ */
  const double* f;           /* 3 x n  f_vec_ (bearing vectors) */
  /* 3 x n: landmark_vec_[i]->pos_, or for seeds
   * seed_ref.keyframe->T_world_cam()*getSeedPosInFrame(seed_id)
   * (sparse_img_align.cpp:281-292); ignored where flags[i]==0 */
  const double* pos_world;
  /* n: 1 iff (landmark_vec_[i] || seed_ref_vec_[i].keyframe) && !isMapPoint(type_vec_[i])
   * (sparse_img_align.cpp:239-245) */
  const uint8_t* flags;
  /* NULL, or n entries (host memory; SVOH_MEM_HOST cameras only): pos_seed_unit[i] >= 0 says feature i is a seed whose position
   * is to be taken ON THE DEVICE from unit pos_seed_unit[i] of the seed batch that was last sent off in a deferred section
   * of this context (svoh_update_seeds_batch(_ex) + svoh_matcher_flush / _collect, host arrays or SVOH_MEM_STAGED; collected or not
   * -- its device block must not have been laid out for another seed batch since): T_world_keyframe x (f / mu) with the inverse depth the update leaves, the arithmetic of
   * seed_ref.keyframe->T_world_cam() * getSeedPosInFrame(seed_id) (sparse_img_align.cpp:281-292).  pos_world[i] is ignored for
   * such a feature.  The alignment of the next frame can then be queued before the host has seen the update's results. */
  const int32_t* pos_seed_unit;
} svoh_align_camera;

typedef struct svoh_align_problem {
  int32_t n_cams;
  int32_t reserved;
  svoh_align_camera cams[SVOH_MAX_CAMS];
  svoh_se3 T_icur_iref;      /* initial value: cur.T_imu_world * ref.T_imu_world^-1 */
  double alpha_init, beta_init;
  svoh_align_prior prior;
} svoh_align_problem;

typedef struct svoh_align_result {
  int32_t status;            /* 0 ok; 1 = no features to track (run() returns 0);
                                2 = solver stopped on NaN (state rolled back);
                                3 = a large problem was spread over several workgroups and one of them did not
                                    reach the device-side barrier within its bounded wait of ~0.1 s (the kernel gives up instead of
                                    hanging).  svoh_sparse_align_batch then repeats the launch with one
                                    workgroup per problem, so only enqueue / fetch callers can see this value */
  int32_t n_fts_to_track;    /* return value of SparseImgAlign::run */
  svoh_se3 T_icur_iref;      /* optimised state */
  double alpha, beta;
  int32_t iters[SVOH_MAX_LEVELS];     /* evaluateError calls per level */
  int32_t n_meas[SVOH_MAX_LEVELS];    /* residuals in the last evaluation  */
  double chi2[SVOH_MAX_LEVELS];       /* chi2/n_meas of the last evaluation */
  /* sum over every evaluateError call of the number of visible patches: the
   * "patch-iterations" the run executed (the unit of SURVEY.md 8(d)) */
  int64_t n_patch_iters;
} svoh_align_result;

/* Replaces SparseImgAlign::run (src/svo_img_align/src/sparse_img_align.cpp:34-113)
 * for n_problems independent (ref bundle, cur bundle) pairs in one launch:
 * feature selection (a-3), base caches (a-4), per-level Jacobians/ref patches
 * (a-5), residuals (a-6), normal equations (a-7), GN driver incl. prior,
 * LDLT and SE3 update (a-1, a-2) all run on the device with no host round trip.
 * The caller composes f->T_f_w_ = T_cam_imu * T_icur_iref * T_iref_world. */
int svoh_sparse_align_batch(svoh_ctx* ctx, const svoh_align_options* options,
                            int n_problems, const svoh_align_problem* problems,
                            svoh_align_result* results);

/* Split form of the above for callers that want to overlap: enqueue returns
 * as soon as the launch is queued on the context stream (kernel and the copy of
 * its results to pinned host memory); fetch blocks and hands out the results of
 * the last enqueue.  Several enqueue calls may be queued before one fetch: the
 * host-side preparation of a launch then overlaps the previous kernel. */
int svoh_sparse_align_enqueue(svoh_ctx* ctx, const svoh_align_options* options,
                              int n_problems, const svoh_align_problem* problems);
int svoh_sparse_align_fetch(svoh_ctx* ctx, int n_problems, svoh_align_result* results);

/* Launch geometry as a function of the PROBLEM instead of the launch.  svoh_sparse_align_enqueue picks workgroup size,
 * lanes per patch and workgroups per problem from what the whole launch looks like; the choices differ in the order of
 * their sums (poses agree to 1e-15, iteration counts are the same), so a problem's result bits depend on what else is in
 * its launch.  A host that runs many independent camera streams in lock step and wants every stream to reproduce its
 * single-stream run bit for bit asks, per problem, for the key of the geometry a launch of THAT problem alone gets, and
 * launches the problems of equal key together: _enqueue_keyed runs every problem in exactly that geometry whatever
 * n_problems is (it may go out as several launches; results queue up in problem order -- collect them with
 * svoh_sparse_align_fetch_all, in the order of the enqueue calls).  Keys are only meaningful to the context (and knob
 * settings) that made them. */
int svoh_sparse_align_geometry_key(svoh_ctx* ctx, const svoh_align_options* options, const svoh_align_problem* problem,
                                   int32_t* key);
/* Which geometry a problem "alone" gets.  0 (default): the one that is fastest for ONE problem of its size -- five classes below 512
 * patches (lanes per patch, 256 / 512 threads, the one-wave-per-SIMD build), so a lock-step round of streams of different sizes goes out
 * as up to four keyed launches one behind the other.  1: SHARED classes -- every problem below 512 patches (all cameras together) runs in
 * the 512-thread lane-per-patch geometry (one key; a rig's cameras side by side as before), larger ones keep the rule of their size: a
 * stream alone pays 0 - 5 % of its alignment (measured: sparse_align.hip, decide_geometry), a round of mixed streams is one or two launches.
 * The setting applies to svoh_sparse_align_geometry_key AND to launches of a single problem (svoh_sparse_align_batch / _enqueue with
 * n_problems = 1), so that a stream's single-stream run and its lock-step run agree bit for bit under either setting -- as long as both
 * use the same one. */
int svoh_set_align_geometry_classes(svoh_ctx* ctx, int shared);
int svoh_sparse_align_enqueue_keyed(svoh_ctx* ctx, const svoh_align_options* options, int n_problems,
                                    const svoh_align_problem* problems, int32_t key);
/* The results of EVERY launch queued since the last fetch / fetch_all, in launch
 * order; n_results must be their total number.  At most 2^18 results may be
 * queued (enqueue fails beyond that: fetch first). */
int svoh_sparse_align_fetch_all(svoh_ctx* ctx, int n_results, svoh_align_result* results);

/* Device time (ms, HIP events on the context stream) of the alignment kernel
 * of the last enqueue/batch call; valid after fetch/batch returned.  Fails when that launch was made with kernel
 * timing off (it never hands out the time of an older launch). */
int svoh_sparse_align_last_kernel_ms(svoh_ctx* ctx, float* ms);

/* Device times (ms) of the last n TIMED alignment launches, oldest first (the library keeps the event pairs of the
 * last 32; launches made while kernel timing was off leave no entry): for callers that queue several enqueue calls
 * before one fetch.  *n_out = entries written. */
int svoh_sparse_align_kernel_ms_history(svoh_ctx* ctx, int n, float* ms, int* n_out);

/* Diagnostic/parity entry: evaluate H (8x8 col-major), g (8), chi2, n_meas for
 * ONE problem at a given level and state, i.e. SparseImgAlign::evaluateError
 * (sparse_img_align.cpp:115-156) on a fresh level.  visibility (may be NULL)
 * receives one byte per selected feature, in selection order. */
int svoh_sparse_align_evaluate(svoh_ctx* ctx, const svoh_align_options* options,
                               const svoh_align_problem* problem, int level,
                               double* H64, double* g8, double* chi2,
                               int32_t* n_meas, uint8_t* visibility,
                               int32_t* n_selected);

/* ---- patch-split Gauss-Newton (second row) -------------- */

/* One alignment problem whose patches are split over several GPUs (or several
 * contexts): every participant holds the same frames and a share of the
 * features, computes the normal equations of its share at the common state,
 * the 74 doubles are summed over the participants (RCCL all-reduce of the
 * caller's device buffer), and every participant then applies the identical
 * update.  The reference already sums per-camera contributions into one H
 * (sparse_img_align.cpp:138-154); this extends the same sum across devices.
 * The state lives in DEVICE memory owned by the caller. */
typedef struct svoh_align_gn_state {
  svoh_se3 T_icur_iref;      /* current estimate (model of the GN solver) */
  double alpha, beta;
  svoh_se3 T_old;            /* state before the last update: restored when the solve fails */
  double alpha_old, beta_old;
  double I_prior[8];         /* prior information, fixed at iteration 0 of a level (applyPrior) */
  double chi2;               /* chi2 / n_meas of the sums the last update consumed */
  int32_t n_meas;            /* residuals in those sums */
  int32_t stop;              /* sticky: the solver met a NaN (mini_least_squares_solver.hpp:73-82) */
  int32_t level_done;        /* the last update converged (or stopped): leave the level */
  int32_t status;            /* as svoh_align_result.status */
} svoh_align_gn_state;

#define SVOH_ALIGN_SUMS_DOUBLES 74   /* H 8x8 column-major, g[8], sum of w*res^2, number of residuals */

/* Context-owned device buffers for one split run (a state and SVOH_ALIGN_SUMS_DOUBLES
 * doubles), for hosts that have no device allocator of their own; valid until
 * svoh_destroy.  Callers with their own device memory may pass that instead. */
int svoh_sparse_align_split_buffers(svoh_ctx* ctx, svoh_align_gn_state** d_state, double** d_sums);

/* Write the problem's initial state (T_icur_iref, alpha_init, beta_init) to d_state. */
int svoh_sparse_align_split_init(svoh_ctx* ctx, const svoh_align_problem* problem,
                                 svoh_align_gn_state* d_state);

/* SparseImgAlign::evaluateError (sparse_img_align.cpp:115-156) over THIS participant's
 * features at the state in d_state, on a fresh level (caches rebuilt):
 * d_sums[0..63] = H, [64..71] = g, [72] = sum of weighted squared residuals
 * (not yet divided), [73] = number of residuals.  The features are spread over
 * n_workgroups workgroups of this GPU (0 = about 256 patches each), whose blocks
 * are added in a fixed order.  Both pointers are device memory; the work is queued on the context stream and the call does not
 * block (svoh_synchronize before handing d_sums to another stream). */
int svoh_sparse_align_partial_sums(svoh_ctx* ctx, const svoh_align_options* options,
                                   const svoh_align_problem* problem, int level, int n_workgroups,
                                   const svoh_align_gn_state* d_state, double* d_sums);

/* One iteration of MiniLeastSquaresSolver::optimizeGaussNewton
 * (mini_least_squares_solver.hpp:42-107) on the summed normal equations:
 * applyPrior, pivoted LDLT, update, convergence test -- the same device code
 * the resident kernel runs.  iter = 0 starts a level.  Blocks; *h_state (may be
 * NULL) receives a copy of the updated state. */
int svoh_sparse_align_gn_update(svoh_ctx* ctx, const svoh_align_options* options,
                                const svoh_align_problem* problem, int level, int iter,
                                const double* d_sums, svoh_align_gn_state* d_state,
                                svoh_align_gn_state* h_state);

/* ---- KLT feature alignment (a-9) -------------------------------------- */

/* FeatureTrackerOptions klt_* (src/svo_tracker/include/svo/tracker/feature_tracking_types.h:15-29) */
typedef struct svoh_klt_options {
  int32_t max_level;                      /* 4 */
  int32_t min_level;                      /* 0 */
  int32_t patch_sizes[SVOH_MAX_LEVELS];   /* {16,16,16,8,8}, indexed by level, multiples of 8, <= 32 */
  int32_t max_iter;                       /* 30 */
  float min_update_squared;               /* 0.001 */
  int32_t reserved;
} svoh_klt_options;

/* Replaces the per-track loop of FeatureTracker::trackFrameBundle
 * (src/svo_tracker/src/feature_tracker.cpp:64-99), i.e. n_tracks calls of
 * feature_alignment::alignPyr2D (src/svo_direct/src/feature_alignment.cpp:761-973;
 * batch form alignPyr2DVec :732-758).  ref_frames[i]: pyramid holding track i's
 * template; px_ref: 2 x n int (ref_observation.getPx().cast<int>());
 * px_cur: 2 x n double, in = track.back().getPx(), out = aligned position;
 * status[i] = 1 iff alignPyr2D returned true.  Host pointers. */
int svoh_klt_track_batch(svoh_ctx* ctx, const svoh_klt_options* options, int n_tracks,
                         const svoh_frame_t* ref_frames, svoh_frame_t cur_frame,
                         const int32_t* px_ref, double* px_cur, uint8_t* status);

/* Same, with one current frame per track (cur_frames[i]): batches the tracks of
 * many frame bundles / camera streams into one launch. */
int svoh_klt_track_multi(svoh_ctx* ctx, const svoh_klt_options* options, int n_tracks,
                         const svoh_frame_t* ref_frames, const svoh_frame_t* cur_frames,
                         const int32_t* px_ref, double* px_cur, uint8_t* status);

/* Same tracks addressed through a frame table, the layout of choice when the per-track
 * arrays already live on the device: frames[n_frames] (host array of handles; all with
 * more than max_level levels and one common size per level), ref_frame_idx / cur_frame_idx
 * (n_tracks each) index into it.  mem_space = SVOH_MEM_HOST: the five per-track arrays are
 * host pointers, indices are validated, results are copied back (identical to
 * svoh_klt_track_multi).  SVOH_MEM_DEVICE: device pointers used in place on the context's
 * stream, a track with an out-of-range index gets status 0, and the call returns without
 * synchronising. */
int svoh_klt_track_indexed(svoh_ctx* ctx, const svoh_klt_options* options, int n_frames,
                           const svoh_frame_t* frames, int n_tracks,
                           const int32_t* ref_frame_idx, const int32_t* cur_frame_idx,
                           const int32_t* px_ref, double* px_cur, uint8_t* status, int mem_space);

/* Device time (ms) of the last KLT / matcher / seed-update kernel of this context. */
int svoh_last_kernel_ms(svoh_ctx* ctx, float* ms);
/* Work counters of that kernel (for the roofline accounting of SURVEY.md 8(d)):
 *   KLT:     [0] track-iterations at 16x16, [1] at 8x8, [2] templates built 16x16, [3] 8x8
 *   matcher: [0] affine warps done, [1] ZMSSD evaluations, [2] align1D/2D iterations,
 *            [3] seeds whose filter state was updated,
 *            [4] / [5] units with >= 5 / >= 10 alignment iterations, [6] / [7] units with >= 20 / >= 50
 *            ZMSSD evaluations (how uneven the units are) */
int svoh_last_kernel_counters(svoh_ctx* ctx, uint64_t out[8]);

/* ---- matcher and depth filter (a-10 ... a-14) -------------------------- */

/* svo::FeatureType (src/svo_common/include/svo/common/types.h:60-73) */
typedef enum svoh_feature_type {
  SVOH_FT_EDGELET_SEED = 0, SVOH_FT_CORNER_SEED = 1, SVOH_FT_MAPPOINT_SEED = 2,
  SVOH_FT_EDGELET_SEED_CONVERGED = 3, SVOH_FT_CORNER_SEED_CONVERGED = 4,
  SVOH_FT_MAPPOINT_SEED_CONVERGED = 5, SVOH_FT_EDGELET = 6, SVOH_FT_CORNER = 7,
  SVOH_FT_MAPPOINT = 8, SVOH_FT_FIXED_LANDMARK = 9, SVOH_FT_OUTLIER = 10
} svoh_feature_type;

/* Matcher::MatchResult (src/svo_direct/include/svo/direct/matcher.h:56-68) */
typedef enum svoh_match_result {
  SVOH_MATCH_SUCCESS = 0, SVOH_MATCH_FAIL_SCORE = 1, SVOH_MATCH_FAIL_TRIANGULATION = 2,
  SVOH_MATCH_FAIL_VISIBILITY = 3, SVOH_MATCH_FAIL_WARP = 4, SVOH_MATCH_FAIL_ALIGNMENT = 5,
  SVOH_MATCH_FAIL_RANGE = 6, SVOH_MATCH_FAIL_ANGLE = 7, SVOH_MATCH_FAIL_CLOSE_VIEW = 8,
  SVOH_MATCH_FAIL_LOCK = 9, SVOH_MATCH_FAIL_TOO_FAR = 10,
  SVOH_MATCH_NOT_RUN = 100  /* updateSeed returned before calling the matcher */
} svoh_match_result;

/* Matcher::Options (matcher.h:39-54); align_1d is chosen per feature by the callers */
typedef struct svoh_matcher_options {
  int32_t align_max_iter;                 /* 10 */
  int32_t max_epi_search_steps;           /* 100 (500 in StereoTriangulation) */
  int32_t subpix_refinement;              /* 1 */
  int32_t epi_search_edgelet_filtering;   /* 1 */
  int32_t scan_on_unit_sphere;            /* Matcher default 1; DepthFilterOptions default 0 */
  int32_t affine_est_offset;              /* 1 */
  int32_t affine_est_gain;                /* 0 */
  int32_t reserved;
  double epi_search_edgelet_max_angle;    /* 0.7 */
  double max_patch_diff_ratio;            /* 2.0 */
} svoh_matcher_options;

/* ---- resident feature columns -----------------------------------------------------------------------------------
 * The per-feature columns of a keyframe that do not change once the keyframe exists -- px_vec_, f_vec_, grad_vec_,
 * level_vec_ (src/svo_common/include/svo/common/frame.h:62-73) -- kept in HBM.  A matcher / depth-filter batch then names a
 * feature by (reference frame, index) instead of carrying its 60 bytes over PCIe every frame: the reprojector and the depth
 * filter go through the same few keyframes' features frame after frame (reprojector.cpp:131-306, depth_filter.cpp:200-233).
 * _upload: n_sets sets in one call (one staging block, one copy; the new keyframes of many camera streams); set k has
 * n[k] >= 0 features, px[k] = 2 x n[k], f[k] = 3 x n[k], grad[k] = 2 x n[k], level[k] = n[k] (host pointers, read before
 * the call returns).  A set is immutable; a keyframe that gains features gets a new set.  _release: the memory goes back
 * to the context's pool once every set uploaded with it has been released; batches queued before the release may still
 * read it (stream order). */
typedef uint64_t svoh_features_t;
int svoh_features_upload(svoh_ctx* ctx, int n_sets, const int32_t* n, const double* const* px, const double* const* f,
                         const double* const* grad, const int32_t* const* level, svoh_features_t* out);
int svoh_features_release(svoh_ctx* ctx, svoh_features_t features);

/* What the matcher reads of a Frame: pyramid, camera, pose */
typedef struct svoh_frame_view {
  svoh_frame_t frame;
  svoh_camera cam;
  svoh_se3 T_f_w;                         /* Frame::T_f_w_ (camera <- world) */
  double seed_mu_range;                   /* Frame::seed_mu_range_ (depth filter only) */
  int32_t id;                             /* Frame::id() */
  /* 0 everywhere but in ONE place: the current frames of a staged seed batch (SVOH_MEM_STAGED) queued from the hook of
   * svoh_optimize_pose_batch_hook.  k > 0 there means: T_f_w above holds the frame's T_cam_imu, and the pose the batch is
   * evaluated at is T_cam_imu * (T_imu_world of result k-1 of the pose batch just launched), composed ON THE DEVICE when
   * the batch's kernels start -- the depth filter's update can be sent off before the host has seen the optimised pose
   * (frame_handler_mono.cpp:120-158: optimizePose, then depth_filter_->updateSeeds with the optimised frame). */
  int32_t pose_result_index_plus1;
  /* reference frames only, 0 = none: the frame's resident columns, for batches that name their features by index
   * (svoh_feature_batch::feature_index) */
  svoh_features_t features;
} svoh_frame_view;

/* n features referencing one of n_ref_frames reference frames, SoA like Frame's
 * feature storage (frame.h:62-73) */
#define SVOH_BATCH_UNITS 0        /* every unit names its reference frame and feature (the arrays below) */
/* seed batches staged with SVOH_STAGE_RESIDENT_COLUMNS, or seed batches with DEVICE arrays (SVOH_MEM_DEVICE: px / f / grad / level /
 * ref_frame_idx NULL; always the packed geometry): the units ARE the reference frames' resident features -- unit order:
 * for r = 0 .. n_ref_frames - 1 in turn, feature 0 .. n_r - 1 of ref_frames[r].features (n = the sum of the sets' sizes, checked).
 * ref_frame_idx and feature_index need not be filled in (they are not read; the staged pointers are still passed); type,
 * cur_frame_idx, state and the outputs are per unit in that order; all seeds of a set go into ONE current frame, as updateSeeds'
 * do (cur_frame_idx must be the same for all units of a set: the packed geometry reads it at the set's first unit).  What it buys: a large batch (the packed geometry) is processed
 * in the TILE order of the reference pixels that svoh_features_upload computed once per keyframe -- no counting sort, no record
 * scatter, no un-sort per frame (depth_filter.cpp:200-251 updates every seed of every keyframe: this IS its unit list). */
#define SVOH_BATCH_WHOLE_SETS 1
typedef struct svoh_feature_batch {
  int32_t n;
  int32_t layout;                         /* SVOH_BATCH_UNITS (0) / SVOH_BATCH_WHOLE_SETS */
  const int32_t* ref_frame_idx;           /* n: index into the ref_frames array */
  const double* px;                       /* 2 x n  px_vec_ */
  const double* f;                        /* 3 x n  f_vec_ */
  const double* grad;                     /* 2 x n  grad_vec_ */
  const int32_t* level;                   /* n      level_vec_ */
  uint8_t* type;                          /* n      type_vec_ (svoh_feature_type), updated by update_seeds */
  /* optional multi-stream batching: feature i is matched into cur_frames[cur_frame_idx[i]];
   * NULL / 0 = every feature uses cur_frames[0] (the reference's one-frame call) */
  const int32_t* cur_frame_idx;           /* n or NULL */
  int32_t n_cur_frames;                   /* length of the cur_frame array passed to the call (0 = 1) */
  /* SVOH_MEM_HOST (0): every array above and every per-feature argument of the call is a
   * host pointer; the call stages them, synchronises and copies the results back.
   * SVOH_MEM_DEVICE: they are device pointers of the context's device, used in place on
   * the context's stream; out-of-range indices mark the feature SVOH_MATCH_NOT_RUN instead
   * of failing the call, and the call returns without synchronising (unless n_success is
   * requested): order later work on svoh_stream() or call svoh_synchronize(). */
  int32_t mem_space;
  /* SVOH_MEM_STAGED batches only, NULL otherwise: feature i is feature feature_index[i] of the resident columns of its
   * reference frame (ref_frames[ref_frame_idx[i]].features); px, f, grad and level above are then NULL -- the library
   * gathers them on the device ahead of the kernels.  An index outside the set marks the unit SVOH_MATCH_NOT_RUN. */
  const int32_t* feature_index;
} svoh_feature_batch;

/* Replaces n calls of Matcher::findMatchDirect (src/svo_direct/src/matcher.cpp:31-141),
 * as made by reprojector_utils::matchCandidate (src/svo/src/reprojector.cpp:384-460).
 * depth: n reference depths; px_cur: 2 x n, in = projection estimate, out = match;
 * outputs per feature: result (svoh_match_result), f_cur (3 x n, normalised bearing),
 * search_level, h_inv (align1D only), A_cur_ref (4 x n, col-major 2x2).  Optional
 * outputs may be NULL.  The caller replays the sequential grid-occupancy logic. */
int svoh_match_direct_batch(svoh_ctx* ctx, const svoh_matcher_options* options,
                            int n_ref_frames, const svoh_frame_view* ref_frames,
                            const svoh_frame_view* cur_frame, const svoh_feature_batch* features,
                            const double* depth, double* px_cur, int32_t* result,
                            double* f_cur, int32_t* search_level, double* h_inv, double* A_cur_ref);

/* The same with Matcher::Options::use_affine_warp_ == false (src/svo_direct/include/svo/direct/matcher.h:50,
 * matcher.cpp:67-81): the reference patch comes from warp::warpPixelwise (src/svo_direct/src/patch_warp.cpp:158-230),
 * which needs the landmark's position: landmark_xyz = 3 x n, world frame (ref_ftr.landmark->pos()), in the memory
 * space of the batch.  A_cur_ref and search_level are those of the affine warp, as in the reference.  Nothing in the
 * reference clears the flag; the branch is built for completeness and runs one lane per feature at every batch size.
 * Not allowed inside a deferred section. */
int svoh_match_direct_batch_pixelwise(svoh_ctx* ctx, const svoh_matcher_options* options, int n_ref_frames,
                                      const svoh_frame_view* ref_frames, const svoh_frame_view* cur_frame,
                                      const svoh_feature_batch* features, const double* depth, const double* landmark_xyz,
                                      double* px_cur, int32_t* result, double* f_cur, int32_t* search_level, double* h_inv,
                                      double* A_cur_ref);

/* Deferred section: between begin and collect, ONE svoh_match_direct_batch and ONE svoh_update_seeds_batch(_ex)
 * with host arrays are queued on the context's stream without a synchronisation; collect waits once and copies
 * every result to the caller's arrays (which must stay valid until then).  This is how one reprojection
 * (Reprojector::reprojectFrames: landmarks and converged seeds through findMatchDirect, unconverged seeds through
 * updateSeed) costs one round trip instead of one per call.  The two queued batches stage through buffers of their
 * own, so other calls on the context inside the section -- device-resident batches (stream-ordered, they run at
 * once), svoh_epipolar_match_batch, svoh_detect_features -- are allowed and leave the queued batches intact; a second
 * host-array batch of a kind already queued fails with "collect first". */
int svoh_matcher_begin_deferred(svoh_ctx* ctx);
int svoh_matcher_collect(svoh_ctx* ctx);
/* Sends the kernels of the batches queued so far in the open section to the device WITHOUT waiting (collect does
 * that by itself otherwise): the caller goes on with host work -- the next frame's image, its pyramid -- and collects
 * when it needs the results.  The section stays open; a batch of a kind that has been flushed cannot be queued again
 * before collect.  This is how the depth filter's seed update leaves the per-frame critical path
 * (DepthFilterHip::updateSeedsAsync). */
int svoh_matcher_flush(svoh_ctx* ctx);
/* In an open deferred section with a queued (not yet flushed) svoh_update_seeds_batch(_ex): replaces the CURRENT frame's
 * view -- its pose -- by `cur_frame` (same frame handle and camera).  The batch can then be queued, and its inputs
 * uploaded, BEFORE the pose it is evaluated at is known: everything else the seed update reads (the keyframes' seeds,
 * their states, the images) does not depend on the pose optimisation that runs meanwhile. */
int svoh_matcher_deferred_set_cur_frame(svoh_ctx* ctx, const svoh_frame_view* cur_frame);

/* ---- f-4: candidate projection of the reprojector ----------------------------------------------------------
 * Replaces the arithmetic of reprojector_utils::getCandidate / projectPointAndCheckVisibility
 * (src/svo/src/reprojector.cpp:489-543; Frame::isVisible, src/svo_common/src/frame.cpp:229-260) for n points of the
 * local map: point i is a world position (kind 0: v = landmark_vec_[i]->pos()) or a seed of keyframe kf[i] (kind 1:
 * v = f_vec_ column, mu = inverse depth; position T_world_kf[kf[i]] * (v / mu)).  Outputs per point: px (2 x n, the
 * projection into the current frame) and visible (1 = getCandidate would return a candidate at px).  Which keyframes
 * are visible, the landmark bookkeeping, std::sort and the ordered replay stay with the caller
 * (host/svo_hip_host.cpp: ReprojectorHip).
 *
 * The enqueue form does not synchronise.  With align_result_index >= 0 the current frame's pose is composed ON THE
 * DEVICE from result #align_result_index of the alignment results queued since the last fetch (svoh_sparse_align_enqueue
 * / _enqueue_keyed; the numbering of svoh_sparse_align_fetch_all -- for the usual single launch, the problem's index):
 *   T_f_w = T_cam_imu (first pose argument) * T_icur_iref (alignment result) * T_imu_world_ref
 * (sparse_img_align.cpp:100-107), so that one svoh_sparse_align_fetch delivers the pose AND the candidates of the
 * frame: the candidate projection costs no round trip of its own.  With align_result_index < 0 the first pose argument
 * is T_f_w itself.  collect copies the results out (n must be the queued call's n).  Host pointers; one queued call
 * at a time. */
int svoh_project_candidates_enqueue(svoh_ctx* ctx, const svoh_camera* cam, const svoh_se3* T_f_w_or_T_cam_imu,
                                    const svoh_se3* T_imu_world_ref, int align_result_index, int n_kf,
                                    const svoh_se3* T_world_kf, int n, const uint8_t* kind, const int32_t* kf,
                                    const double* v, const double* mu);
int svoh_project_candidates_collect(svoh_ctx* ctx, int n, double* px, uint8_t* visible);
/* blocking form with an explicit T_f_w */
int svoh_project_candidates(svoh_ctx* ctx, const svoh_camera* cam, const svoh_se3* T_f_w, int n_kf,
                            const svoh_se3* T_world_kf, int n, const uint8_t* kind, const int32_t* kf, const double* v,
                            const double* mu, double* px, uint8_t* visible);

/* The candidate projections of MANY current frames (camera streams in lock step) in one launch, staged in place:
 * _stage sizes the context's page-locked block for n_jobs jobs, n_kf_total keyframe poses and n_points_total points and
 * hands out its arrays; the caller fills them (job j owns keyframe poses [kf_begin, kf_begin + n_kf) and points
 * [point_begin, point_begin + n_points); job[i] = the job of point i; kf[i] counts from the job's kf_begin) -- several host
 * threads may fill disjoint parts at once --, _enqueue_staged uploads and launches without a wait, _wait blocks until the
 * results stand in out->px / out->visible (valid until the next _stage).  Arithmetic, per point, is that of
 * svoh_project_candidates_enqueue.  Not to be mixed with a queued svoh_project_candidates_enqueue call. */
typedef struct svoh_candidate_job {
  svoh_camera cam;
  svoh_se3 T_f_w_or_T_cam_imu;      /* as the first pose argument of svoh_project_candidates_enqueue */
  svoh_se3 T_imu_world_ref;         /* read when align_result_index >= 0 */
  int32_t align_result_index;       /* < 0: the first pose is T_f_w itself */
  int32_t kf_begin, n_kf;
  int32_t point_begin, n_points;
  int32_t reserved;
} svoh_candidate_job;
/* The form with resident columns (_stage_ranges): entry k of the keyframe table is ALSO a range of points -- the features
 * 0 .. n_points-1 of the keyframe's resident columns (svoh_features_upload), in order, as points point_begin .. of job
 * `job`.  Ranges lie back to back (range k+1 begins where range k ends).  Per point only kind and mu cross PCIe: a seed's
 * bearing vector is the f column of its feature; v is read for landmarks (kind 0) only, and uploaded only when there is one. */
typedef struct svoh_candidate_range {
  svoh_features_t features;
  int32_t point_begin, n_points;
  int32_t job;
  int32_t reserved;
} svoh_candidate_range;
typedef struct svoh_candidate_stage_t {
  svoh_candidate_job* jobs;         /* n_jobs */
  svoh_se3* T_world_kf;             /* n_kf_total */
  int32_t* job;                     /* n_points_total (NULL in the ranges form) */
  uint8_t* kind;                    /* n_points_total */
  int32_t* kf;                      /* n_points_total (NULL in the ranges form) */
  double* v;                        /* 3 x n_points_total */
  double* mu;                       /* n_points_total */
  double* px;                       /* out: 2 x n_points_total */
  uint8_t* visible;                 /* out: n_points_total */
  svoh_candidate_range* ranges;     /* n_kf_total (the ranges form), else NULL */
  /* the ranges form: n_points_total entries, preset to -1.  mu_unit[i] >= 0: the seed's inverse depth is read ON THE DEVICE from
   * unit mu_unit[i] of the seed batch last sent off on this context (as svoh_align_camera::pos_seed_unit; mu[i] is ignored), so
   * that the projection can be queued before the host has seen that update's results.  Enqueue with
   * svoh_project_candidates_enqueue_staged_units then. */
  int32_t* mu_unit;
} svoh_candidate_stage_t;
int svoh_project_candidates_stage(svoh_ctx* ctx, int n_jobs, int n_kf_total, int n_points_total, svoh_candidate_stage_t* out);
int svoh_project_candidates_stage_ranges(svoh_ctx* ctx, int n_jobs, int n_kf_total, int n_points_total, svoh_candidate_stage_t* out);
int svoh_project_candidates_enqueue_staged(svoh_ctx* ctx);
int svoh_project_candidates_enqueue_staged_units(svoh_ctx* ctx);   /* the ranges form with mu_unit entries >= 0 */
int svoh_project_candidates_wait(svoh_ctx* ctx);

/* The candidate SELECTION of reprojector_utils::matchCandidates (src/svo/src/reprojector.cpp:342-382) for many lists at once, the
 * matches taken as given: list l (a stream's pass; candidates in visiting order) is candidates [begin[l], begin[l+1]); candidate i
 * projects into grid cell cell[i] and its match succeeded iff success[i] != 0.  Per list: the first success of every cell that was
 * free wins the cell (an atomic minimum on the visiting index), candidates are visited up to the winner that fills the frame
 * (num_features[l] + winners so far >= max_n_features[l]) and only while their cell is free -- exactly the sequential loop, in
 * parallel.  occupancy: n_lists x n_cells bytes in / out (a list's grid before / after its pass); visited: one byte per candidate
 * (1 = the loop tried it: n_trials counts these, its side effects apply); n_consumed[l] = how many candidates the loop consumed
 * (candidates.erase); num_features in / out.  max_n_features[l] must be > 0 (with 0 the reference's loop ignores the grid).
 * Host pointers, blocking.  The reprojector's mirror keeps the selection on the host by default (it needs the lists sorted, and
 * the sort runs while the matcher kernel does: HISTORY.md, round 5); SVOH_REPROJ_DEVICE_SELECT=1 routes its replay through here. */
int svoh_select_matches_batch(svoh_ctx* ctx, int n_lists, const int32_t* begin, const int32_t* cell, const uint8_t* success,
                              int n_cells, uint8_t* occupancy, const int32_t* max_n_features, int32_t* num_features,
                              uint8_t* visited, int32_t* n_trials, int32_t* n_matches, int32_t* n_consumed);

/* A matcher batch staged in place (svoh_feature_batch.mem_space = SVOH_MEM_STAGED).  Inside an open deferred section:
 * _stage sizes the page-locked block of the section's direct (seeds = 0) or seed (seeds = 1) batch for n units and up to
 * max_frame_views reference + current frames and hands out every array of the batch; the caller fills the inputs in
 * place and then makes the usual call -- svoh_match_direct_batch / svoh_update_seeds_batch_ex -- with exactly these
 * pointers (feature arrays, depth / px_cur resp. state, and the outputs it wants: the others NULL).  The call copies
 * nothing; after svoh_matcher_collect the outputs stand where they were handed out (until the next _stage of that
 * kind).  cur_frame_idx is always present (n_cur_frames >= 1 current frames); index checks are the kernels' (a bad index
 * marks the unit SVOH_MATCH_NOT_RUN).  With want_match_outputs == 0 a seed batch has no px_cur / f_cur / search_level /
 * A_cur_ref arrays (NULL here) and brings back type, state, result and success only. */
typedef struct svoh_matcher_stage_t {
  int32_t* ref_frame_idx;  int32_t* cur_frame_idx;
  double* px;  double* f;  double* grad;  int32_t* level;  uint8_t* type;
  double* depth;           /* direct: in */
  double* px_cur;          /* direct: in / out; seeds: out */
  double* state;           /* seeds: in / out */
  int32_t* result;  uint8_t* success;  double* f_cur;  int32_t* search_level;  double* h_inv;  double* A_cur_ref;
  int32_t* feature_index;  /* SVOH_STAGE_RESIDENT_COLUMNS: in (px, f, grad, level above are NULL then) */
} svoh_matcher_stage_t;
/* `flags` (the argument was called want_match_outputs: 0 / 1 mean what they meant) */
#define SVOH_STAGE_MATCH_OUTPUTS 1     /* a seed batch brings px_cur / f_cur / search_level / A_cur_ref back as well */
#define SVOH_STAGE_RESIDENT_COLUMNS 2  /* features named by (ref_frame_idx, feature_index): see svoh_features_upload */
int svoh_matcher_stage(svoh_ctx* ctx, int seeds, int n, int max_frame_views, int flags, svoh_matcher_stage_t* out);

/* DepthFilterOptions used by updateSeed (src/svo_direct/include/svo/direct/depth_filter.h:40-100) */
typedef struct svoh_depth_filter_options {
  double seed_convergence_sigma2_thresh;      /* 200 */
  double mappoint_convergence_sigma2_thresh;  /* 500 */
  /* the function-local static of updateSeed: cur_frame.getAngleError(1.0) of the FIRST
   * frame ever passed (depth_filter.cpp:383-384): atan(1/(2fx)) + atan(1/(2fy)) */
  double px_error_angle;
  int32_t check_visibility;                   /* 1 */
  int32_t check_convergence;                  /* 0 */
  int32_t use_vogiatzis_update;               /* 1 */
  int32_t reserved;
} svoh_depth_filter_options;

/* Replaces the synchronous branch of DepthFilter::updateSeeds
 * (src/svo_direct/src/depth_filter.cpp:200-233): for every feature whose type isSeed(),
 * depth_filter_utils::updateSeed (:367-499) = visibility test, epipolar search
 * (Matcher::findEpipolarMatchDirect, matcher.cpp:157-241), computeTau (:580-596),
 * updateFilterVogiatzis/Gaussian (:501-578), convergence test (seed.h:143-151).
 * state: 4 x n [mu = 1/depth, sigma2, a, b] (invmu_sigma2_a_b_vec_), updated in place;
 * features->type updated in place (converged / outlier); success[i] = return value of
 * updateSeed; match_result (may be NULL) = svoh_match_result of the epipolar search.
 * Returns the number of successes in *n_success. */
int svoh_update_seeds_batch(svoh_ctx* ctx, const svoh_matcher_options* matcher_options,
                            const svoh_depth_filter_options* options,
                            int n_ref_frames, const svoh_frame_view* ref_frames,
                            const svoh_frame_view* cur_frame, const svoh_feature_batch* features,
                            double* state, uint8_t* success, int32_t* match_result,
                            int32_t* n_success);

/* What Matcher holds after the epipolar search of updateSeed and what
 * reprojector_utils::matchCandidate (src/svo/src/reprojector.cpp:403-413, 473-476) reads
 * from it when an unconverged seed is matched during reprojection: px_cur_, f_cur_,
 * search_level_, A_cur_ref_.  Every pointer may be NULL; arrays are n-long batches like
 * the outputs of svoh_match_direct_batch and live where features->mem_space says. */
typedef struct svoh_seed_match_outputs {
  double* px_cur;          /* 2 x n */
  double* f_cur;           /* 3 x n */
  int32_t* search_level;   /* n */
  double* A_cur_ref;       /* 4 x n, col-major 2x2 */
} svoh_seed_match_outputs;

/* svoh_update_seeds_batch that also returns the matcher state of every seed (outputs may
 * be NULL = svoh_update_seeds_batch).  Entries of seeds whose update returned before the
 * epipolar search (match_result SVOH_MATCH_NOT_RUN) read back as zeros from a host-resident
 * batch.  In a device-resident batch they are unspecified: the per-unit kernels (batches of up to
 * 49 152 seeds) write zeros there as well, the packed kernel of larger batches leaves them untouched. */
int svoh_update_seeds_batch_ex(svoh_ctx* ctx, const svoh_matcher_options* matcher_options,
                               const svoh_depth_filter_options* options,
                               int n_ref_frames, const svoh_frame_view* ref_frames,
                               const svoh_frame_view* cur_frame, const svoh_feature_batch* features,
                               double* state, uint8_t* success, int32_t* match_result,
                               int32_t* n_success, const svoh_seed_match_outputs* outputs);

/* ---- stereo seam: plain epipolar matches with their triangulated depth ---- */

/* Per-feature outputs of svoh_epipolar_match_batch: what Matcher::findEpipolarMatchDirect
 * returns and leaves in the Matcher (matcher.h:70-90).  result and depth are required,
 * the others may be NULL; arrays live where features->mem_space says. */
typedef struct svoh_epipolar_match_outputs {
  int32_t* result;         /* n      svoh_match_result */
  double* depth;           /* n      `depth` out-parameter (valid when result == SVOH_MATCH_SUCCESS) */
  double* px_cur;          /* 2 x n  Matcher::px_cur_ */
  double* f_cur;           /* 3 x n  Matcher::f_cur_ */
  int32_t* search_level;   /* n      Matcher::search_level_ */
  double* h_inv;           /* n      Matcher::h_inv_ (align1D) */
  double* A_cur_ref;       /* 4 x n  Matcher::A_cur_ref_, col-major 2x2 */
} svoh_epipolar_match_outputs;

/* Replaces n calls of
 *   Matcher::findEpipolarMatchDirect(ref_frame, cur_frame, T_cur_ref, ref_ftr, d_estimate_inv, d_min_inv, d_max_inv, depth)
 * (src/svo_direct/src/matcher.cpp:157-241), each on a fresh Matcher with options_.align_1d = isEdgelet(type),
 * which is how StereoTriangulation::compute (src/svo/src/stereo_triangulation.cpp:92-104) matches the new
 * features of the left frame into the right frame (max_epi_search_steps = 500 there).
 * T_cur_ref: n_ref_frames x n_cur_frames transforms (host array), entry [ref_idx * n_cur_frames + cur_idx], as the
 * stereo caller passes T_f1f0 = T_cam1_body * T_body_cam0; NULL = cur.T_f_w * ref.T_f_w^-1 (the 7-argument
 * overload, matcher.cpp:143-155).
 * d_inv_common = {d_estimate_inv, d_min_inv, d_max_inv} for every feature; d_inv (3 x n, may be NULL, lives
 * where features->mem_space says) overrides it per feature.
 * The caller replays its own sequential bookkeeping (the stereo loop stops at n_desired successes). */
int svoh_epipolar_match_batch(svoh_ctx* ctx, const svoh_matcher_options* options,
                              int n_ref_frames, const svoh_frame_view* ref_frames,
                              const svoh_frame_view* cur_frame, const svoh_se3* T_cur_ref,
                              const svoh_feature_batch* features, const double d_inv_common[3],
                              const double* d_inv, const svoh_epipolar_match_outputs* outputs);

/* ---- keyframe feature detector ----------------------- */

/* DetectorOptions (src/svo_direct/include/svo/direct/feature_detection_types.h:49-84) */
typedef struct svoh_detector_options {
  int32_t cell_size;            /* 30: at most one feature per cell_size x cell_size bucket */
  int32_t max_level;            /* 2 */
  int32_t min_level;            /* 0 */
  int32_t border;               /* 8 */
  int32_t detect_edgelets;      /* 0 = DetectorType::kFast, 1 = kFastGrad (corners, then edgelets in free cells) */
  int32_t reserved;
  double threshold_primary;     /* 10: FAST barrier and minimum corner score */
  double threshold_secondary;   /* 100: gradient magnitude threshold of the edgelet detector */
} svoh_detector_options;

/* Replaces FastDetector::detect / FastGradDetector::detect (src/svo_direct/src/feature_detection.cpp:113-194):
 * FAST-10 corners on levels min_level..max_level (fast_corner_detect_10 + fast_corner_score_10 + 3x3 non-maximum
 * suppression, src/fast_neon), best corner per free grid cell; optionally Scharr-gradient edgelets on level 1 in
 * the cells that are still free; fillFeatures' threshold / mask / sort by score.
 * occupancy: ceil(width/cell_size) * ceil(height/cell_size) bytes, non-zero = cell already holds a feature
 * (OccupandyGrid2D::fillWithKeypoints), may be NULL; not modified (the reference resets its grid after detect).
 * mask: level-0 sized u8 image (0 = no feature here) or NULL.  Outputs hold one entry per grid cell at most:
 * px (2 x n, level-0 pixels), score, level, grad (2 x n, unit vector), type (svoh_feature_type).  Host pointers.
 * Equal scores keep cell order (the reference's std::sort leaves their order to the library). */
int svoh_detect_features(svoh_ctx* ctx, svoh_frame_t frame, const svoh_detector_options* options,
                         const uint8_t* occupancy, const uint8_t* mask, int mask_pitch, int max_n_features,
                         double* px, double* score, int32_t* level, double* grad, uint8_t* type,
                         int32_t* n_features);

/* The detector for MANY frames of one size in one round trip, in two halves.  svoh_detect_cells_batch is the device
 * half: for every frame and grid cell the best corner (key: score << 32 | ~(level, y, x), as fd_utils::fastDetector's
 * per-cell best, 0 = none above threshold_primary) and, in the cells that neither `occupancy` nor a corner takes, the
 * best edgelet (key: float bits of the magnitude << 32 | ~(y, x) on level 1) with its histogram angle.  occupancy: n_frames
 * x n_cells bytes or NULL; the three outputs n_frames x n_cells each (host).  No mask (a masked-out corner would have to
 * free its cell between the two phases).  svoh_detect_fill_features is the host half, fd_utils::fillFeatures for corners
 * then edgelets of ONE frame from its n_cells entries of each array: pure host code without a context, callable from
 * any thread.  Both halves together give exactly svoh_detect_features' features (mask == NULL). */
int svoh_detect_cells_batch(svoh_ctx* ctx, int n_frames, const svoh_frame_t* frames, const svoh_detector_options* options,
                            const uint8_t* occupancy, uint64_t* corner_keys, uint64_t* edge_keys, float* edge_angles);
/* The device half in two steps: _enqueue queues uploads, kernels and the copy of the results and returns; _collect waits for
 * THAT batch (an event behind its copy -- not for work the caller has queued on the context since) and hands the arrays out.
 * One batch in flight per context; between the two the caller may make any other call.  This is how the detector of a round's
 * new keyframes runs ahead of the pose optimisation instead of behind the depth filter's update. */
int svoh_detect_cells_batch_enqueue(svoh_ctx* ctx, int n_frames, const svoh_frame_t* frames, const svoh_detector_options* options,
                                    const uint8_t* occupancy);
int svoh_detect_cells_batch_collect(svoh_ctx* ctx, uint64_t* corner_keys, uint64_t* edge_keys, float* edge_angles);
int svoh_detect_fill_features(const svoh_detector_options* options, int width, int height, const uint64_t* corner_keys,
                              const uint64_t* edge_keys, const float* edge_angles, int max_n_features, double* px,
                              double* score, int32_t* level, double* grad, uint8_t* type, int32_t* n_features);

/* feature_detection_utils::getAngleAtPixelUsingHistogram(img_pyr[level], px, 4) (feature_detection_utils.cpp:831-839, 947-1009) for n
 * pixels of possibly different frames and levels in one call: bins[k] = the dominant bin (0 .. 35) of the smoothed 36-bin histogram of the
 * 9x9 window's gradient directions; the angle is bins[k] * 2 pi / 36.  px: 2 x n integer pixels AT THAT LEVEL.  This is how
 * FrameHandlerBase::upgradeSeedsToFeatures refreshes the direction of an edgelet it upgrades (frame_handler_base.cpp:893-901).  Blocking. */
int svoh_histogram_angle_bins(svoh_ctx* ctx, int n_frames, const svoh_frame_t* frames, int n, const int32_t* frame_idx, const int32_t* level,
                              const int32_t* px, int32_t* bins);

/* ---- pose optimiser ----------------------------------- */

/* PoseOptimizer::ErrorType (src/svo/include/svo/pose_optimizer.h:33) */
typedef enum svoh_pose_error_type {
  SVOH_POSE_ERR_UNIT_PLANE = 0, SVOH_POSE_ERR_BEARING_DIFF = 1, SVOH_POSE_ERR_IMAGE_PLANE = 2
} svoh_pose_error_type;

typedef struct svoh_pose_options {
  int32_t max_iter;               /* 10  (PoseOptimizer::getDefaultSolverOptions, pose_optimizer.cpp:22-29) */
  int32_t error_type;             /* svoh_pose_error_type; kUnitPlane unless poseoptim_using_unit_sphere */
  double eps;                     /* 1e-6 */
  /* removeOutliers' threshold in the unit of the error type (pose_optimizer.cpp:211-218: reproj_thresh / focal
   * length on the unit plane, |2 sin(angle_error / 2)| for bearing differences, pixels on the image plane);
   * the reference computes it once per process in function-local statics -- the caller owns that quirk */
  double outlier_threshold;
  int32_t have_rotation_prior;    /* setRotationPrior (pose_optimizer.cpp:30-37) */
  int32_t reserved;
  double prior_lambda;
  double R_prior[4];              /* R_frame_world, quaternion w x y z */
} svoh_pose_options;

/* one frame of the bundle: the SoA columns PoseOptimizer reads (frame.h:62-73) + the 3-D point of every
 * feature as evaluateErrorImpl resolves it (landmark position or seed position, pose_optimizer.cpp:128-139) */
typedef struct svoh_pose_camera {
  svoh_camera cam;
  svoh_se3 T_cam_imu;
  int32_t n_features;
  int32_t reserved;
  const double* px;               /* 2 x n */
  const double* f;                /* 3 x n */
  const double* grad;             /* 2 x n */
  const int32_t* level;           /* n */
  const uint8_t* type;            /* n  svoh_feature_type (edgelets get the 1-D residual and 2x sigma) */
  const double* xyz_world;        /* 3 x n */
  const uint8_t* usable;          /* n: 1 = has a landmark or is a corner/edgelet seed; 0 = skipped */
  uint8_t* outlier;               /* n out (may be NULL): 1 = removeOutliers marks it kOutlier */
  double* final_error;            /* n out (may be NULL): unwhitened error / 2^level after the optimisation */
} svoh_pose_camera;

typedef struct svoh_pose_problem {
  int32_t n_cams;
  int32_t reserved;
  svoh_pose_camera cams[SVOH_MAX_CAMS];
  svoh_se3 T_imu_world;           /* frame_bundle->at(0)->T_imu_world() */
} svoh_pose_problem;

typedef struct svoh_pose_result {
  svoh_se3 T_imu_world;           /* optimised; the caller sets T_f_w_ = T_cam_imu * T_imu_world per frame */
  double measurement_sigma;       /* 1.48 * median of the start errors (MADScaleEstimator) */
  double reproj_error_before;     /* median start error, median final error (stats_, in the error type's unit) */
  double reproj_error_after;
  int32_t n_meas;
  int32_t n_deleted_edges, n_deleted_corners;
  int32_t iters;
  int32_t status;                 /* 0 ok, 1 no measurement, 2 solver stopped (singular system) */
  int32_t reserved;
} svoh_pose_result;

/* Replaces PoseOptimizer::run (src/svo/src/pose_optimizer.cpp:39-113) for n_problems frame bundles at once:
 * start errors -> MAD scale, Gauss-Newton on T_imu_world with Tukey weights (and the optional rotation prior),
 * removeOutliers.  run()'s return value is n_meas - n_deleted_edges - n_deleted_corners.  Host pointers. */
int svoh_optimize_pose_batch(svoh_ctx* ctx, const svoh_pose_options* options, int n_problems,
                             const svoh_pose_problem* problems, svoh_pose_result* results);
/* The same call with a hook: after the launch and the copy of its results have been queued, `after_launch(user)` runs on
 * the calling thread, and only then the call waits -- for an event behind ITS OWN work, not for the stream.  Whatever the
 * hook queues on the context (the depth filter's staging for the seed update that follows the pose optimisation:
 * DepthFilterHip::prepareUpdateSeeds) is uploaded while the pose kernel runs and is not waited for here.  The hook must
 * not call a blocking entry of this context. */
int svoh_optimize_pose_batch_hook(svoh_ctx* ctx, const svoh_pose_options* options, int n_problems,
                                  const svoh_pose_problem* problems, svoh_pose_result* results,
                                  void (*after_launch)(void* user), void* user);

/* The same for feature arrays that already live on the device (a multi-stream server, or the per-frame chain kept
 * resident): one set of arrays concatenated over all bundles in problem order and, inside a bundle, camera order --
 * bundle p's camera c owns the n_features entries after those of the cameras before it.  The pointers inside
 * `problems[].cams[]` are ignored (n_features, cam, T_cam_imu are read); outlier / final_error are written in place;
 * nothing but the descriptors and the per-bundle results crosses PCIe, so the call is bound by the kernel. */
typedef struct svoh_pose_packed_arrays {
  int64_t n_features_total;
  const double* px;               /* 2 x N */
  const double* f;                /* 3 x N */
  const double* grad;             /* 2 x N */
  const int32_t* level;           /* N */
  const uint8_t* type;            /* N */
  const double* xyz_world;        /* 3 x N */
  const uint8_t* usable;          /* N */
  uint8_t* outlier;               /* N out */
  double* final_error;            /* N out */
} svoh_pose_packed_arrays;
int svoh_optimize_pose_batch_packed(svoh_ctx* ctx, const svoh_pose_options* options, int n_problems,
                                    const svoh_pose_problem* problems, const svoh_pose_packed_arrays* arrays,
                                    svoh_pose_result* results);

/* Replaces Point::optimize (src/svo_common/src/point.cpp:248-325) for a batch of landmarks, as
 * FrameHandlerBase::optimizeStructure (src/svo/src/frame_handler_base.cpp:779-825) calls it for every
 * landmark of a new keyframe: 3-DoF Gauss-Newton on the landmark position over its observations
 * (bearing vector f of the observing feature and T_f_w of its frame), unit-plane residuals or, with
 * using_bearing_vector (omnidirectional cameras), unit-sphere residuals; at most n_iter iterations,
 * stop on growing error (step rolled back) or max|dp| <= 1e-10.  Landmark i owns observations
 * [obs_begin[i], obs_begin[i+1]); obs_view indexes T_f_w.  A landmark with fewer than two
 * observations is left unchanged.  pos is updated in place; iters (may be NULL) receives the
 * iterations started per landmark.  Host pointers. */
int svoh_optimize_points_batch(svoh_ctx* ctx, int n_iter, int using_bearing_vector, int n_views,
                               const svoh_se3* T_f_w, int n_points, const int32_t* obs_begin,
                               const int32_t* obs_view, const double* obs_f, double* pos, int32_t* iters);
/* The same in two halves: _enqueue stages the batch (buffers of its own), queues upload, kernel and copy back on the context's stream and
 * returns; _collect waits for THAT batch (an event behind its copy -- not for what the caller has queued since) and hands the positions
 * out; n_points must be the queued batch's.  One batch in flight per context.  The frame handler optimises a frame's landmarks right
 * after its pose (frame_handler_mono.cpp:157) and reads them again at the next frame: a driver queues the batch there and collects it
 * when the next frame's alignment points are resolved (FrontendLockstep). */
int svoh_optimize_points_batch_enqueue(svoh_ctx* ctx, int n_iter, int using_bearing_vector, int n_views,
                                       const svoh_se3* T_f_w, int n_points, const int32_t* obs_begin,
                                       const int32_t* obs_view, const double* obs_f, const double* pos);
int svoh_optimize_points_batch_collect(svoh_ctx* ctx, int n_points, double* pos, int32_t* iters);

#ifdef __cplusplus
}
#endif
#endif /* SVO_HIP_H_ */
