// svo_hip_lockstep.h -- many camera streams through the per-frame chain in LOCK STEP: one launch per stage for all of them.
//
// The reference runs one frame handler per camera stream; its only parallelism across streams is std::async per camera
// (src/svo/src/frame_handler_base.cpp:681-695).  On an MI355X one EuRoC-sized stream is a chain of latency-bound round
// trips that keeps < 1 % of the device busy, and a host thread per stream saturates on the launch path near 7 000
// frames/s.  FrontendLockstep takes ONE frame of EVERY stream at a time and runs the chain of
// FrameHandlerMono::processFrame (src/svo/src/frame_handler_mono.cpp:120-158, frame_handler_base.cpp:610-825) stage by
// stage:
//
//   pyramids            one gather + one pyramid launch for S images             svoh_build_pyramid_multi
//   sparse alignment    S problems, grouped by launch geometry                   svoh_sparse_align_enqueue_keyed
//    + candidate proj.  S jobs queued behind it, poses composed on the device    svoh_project_candidates_stage / _enqueue_staged
//   reprojection        one direct batch + one seed batch with S current frames  svoh_matcher_stage + svoh_match_direct_batch /
//                                                                                svoh_update_seeds_batch_ex (cur_frame_idx)
//   pose optimisation   S bundles                                                svoh_optimize_pose_batch
//   structure optim.    the landmarks of all streams' frames                     svoh_optimize_points_batch
//   depth filter        one seed batch over the keyframes of all streams         svoh_matcher_stage + svoh_update_seeds_batch
//   keyframes           the detector for every new keyframe of the round         svoh_detect_cells_batch
//
// Each stream's host work -- the walk over its keyframes, plan, sort, replay, pose staging, seed gathering -- is the
// SAME code a single stream runs (the phase interfaces of ReprojectorHip / PoseOptimizerHip / SparseImgAlignHip), on a
// small pool of worker threads that write straight into the context's page-locked staging blocks; only the thread that
// owns the context talks to the device.  Every alignment problem runs in the launch geometry it would get alone
// (svoh_sparse_align_geometry_key), every other kernel's result does not depend on what shares its launch: a stream's
// trajectory and counters are those of its single-stream run, byte for byte (tests/test_mini_frontend_gpu.py).
//
// Like tools/svoh_mini_frontend.cpp's single-stream chain this is an integration harness above the mirrors, NOT the
// reference's frame handler: no map, no initialiser (first pose and depth prior are given), keyframes by a fixed rule.
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "svo_hip_io.h"
#include "svo_hip_pool.h"

namespace svo_hip {

// What may differ between the streams of one engine (round 6).  Everything that decides the options of a device call the streams
// of a round SHARE -- the image size, the pyramid's levels, the grid, the detector / matcher / depth-filter switches --
// must be the same for all of them (the constructor checks and refuses); a stream's own are its camera (intrinsics, distortion model and
// coefficients, extrinsics: every physical camera has its own calibration -- same image size, since the streams' pyramids are one call), its feature budgets (max_fts: the
// reprojector's cap and, with max_seeds_ratio, the seeds per keyframe), its alignment options (streams of different options go
// into different launches), its keyframe rule and its depth prior.  The reference's counterpart: independent frame handlers, each
// built from its own parameter file (src/svo/include/svo/frame_handler_base.h:274-374, svo_factory.cpp:107-310).
struct LockstepStreamOptions {
  io::FrontendParams params;
  float depth_min = 1.f, depth_mean = 2.f, depth_max = 4.f;
  size_t kf_every = 8, min_tracked = 60;
  // the stream's own calibration (own_camera = false: LockstepOptions::cam / T_B_C); width and height must be those of LockstepOptions::cam
  bool own_camera = false;
  svoh_camera cam{};
  Transformation T_B_C{ { 1, 0, 0, 0 }, { 0, 0, 0 } };
};

struct LockstepOptions {
  // every stream's options when per_stream is empty; the options shared by all streams otherwise
  io::FrontendParams params;
  svoh_camera cam{};
  Transformation T_B_C{ { 1, 0, 0, 0 }, { 0, 0, 0 } };
  float depth_min = 1.f, depth_mean = 2.f, depth_max = 4.f;   // the depth prior of a new keyframe's seeds
  size_t kf_every = 8, min_tracked = 60;                     // the harness' keyframe rule
  int n_workers = 1;                                          // host threads, the caller included (a pool of the engine's own)
  // ... or, when set, no pool of its own: the engine's phases draw on worker threads shared with other engines (one
  // engine per lock-step group); item i of this engine prefers worker (i + shared_pool_seed) % workers
  std::shared_ptr<SharedPool> shared_pool;
  int shared_pool_seed = 0;
  // ... or one pool that the engines take in turns, a phase at a time (ExclusivePool)
  std::shared_ptr<ExclusivePool> exclusive_pool;
  bool pin_workers = false;                                   // bind them to CPUs of their own (WorkerPool)
  int images_mem_space = SVOH_MEM_HOST;                       // SVOH_MEM_HOST_PINNED: images live in svoh_host_alloc memory
  // a keyframe's constant feature columns are uploaded once (svoh_features_upload); the matcher and depth-filter batches of
  // the frames after it name features by index instead of carrying 60 bytes per feature over PCIe every frame
  bool resident_features = true;
  // The third candidate list of a stream (its unconverged seeds, up to a thousand matcher units) joins the round's batch ...
  //   kSpeculateAsBefore  only if the stream's frame before reached that pass (ReprojectorHip::reprojectFrames' own policy); a stream
  //                       that reaches an unplanned pass pauses its replay and gets a batch of its own
  //   kSpeculateAll       always          kSpeculateNever   never: every third pass goes through the paused replay (tests)
  enum Speculation { kSpeculateAsBefore = 0, kSpeculateAll = 1, kSpeculateNever = 2 } speculation = kSpeculateAsBefore;
  // landmarks (round 6): a frame selected as keyframe upgrades the seeds its features hang on to points (svo_hip::upgradeSeedsToFeatures,
  // frame_handler_base.cpp:828-920; the refreshed edgelet directions of all streams in one device call), and every frame's landmarks go
  // through the structure optimisation (frame_handler_mono.cpp:157) -- all streams' points in ONE svoh_optimize_points_batch per round.
  // false: the seed-only chain of round 5.
  bool landmarks = true;
  // one entry per stream, or empty: every stream runs with params / depth_* / kf_every / min_tracked above
  std::vector<LockstepStreamOptions> per_stream;
};

class FrontendLockstep {
 public:
  // what a stream's frame left behind (the columns of svoh_mini_frontend's frontend.csv)
  struct FrameRow { size_t k = 0; bool is_kf = false; size_t n_aligned = 0, n_reproj = 0, n_pose = 0, n_seed_upd = 0, n_converged = 0, n_struct = 0, n_landmarks = 0; };
  struct RoundTimes { double pyramid = 0, align = 0, reproject = 0, pose = 0, seeds = 0, keyframe = 0, total = 0; };   // ms, the round as a whole

  FrontendLockstep(svoh_ctx* ctx, int n_streams, const LockstepOptions& options);
  ~FrontendLockstep();
  FrontendLockstep(const FrontendLockstep&) = delete;
  FrontendLockstep& operator=(const FrontendLockstep&) = delete;

  int numStreams() const { return static_cast<int>(streams_.size()); }
  // One frame of every stream: images[s] = level 0 of stream s' image (all of the camera's size, `pitch` bytes per row), or
  // NULL: stream s has no frame in this round (cameras of different rates, a dropped frame, a stream that starts later or has
  // ended) and is left exactly as it is.  A stream's FIRST frame -- in whatever round it comes -- makes its first keyframe at
  // T_f_w_first[s] (needed in every round in which some stream starts); its k-th frame is numbered k whatever the round.
  // next_images (may be NULL): the images of the NEXT call, if the caller has them already -- they are then sent to the
  // device during this round, on a second stream beside the chain's work (svoh_build_pyramid_multi_prefetch), and the
  // next call, which must be given exactly these pointers as `images`, finds its pyramids made.
  void addImages(const uint8_t* const* images, int pitch, const Transformation* T_f_w_first, const uint8_t* const* next_images = nullptr);
  // the pose of stream s after the last addImages (T_f_w of its newest frame)
  const Transformation& pose(int s) const;
  // Rows are complete once the frame's depth-filter update has been finished, i.e. at the start of the next addImages (or
  // in finish()): completedRows(s) hands out, and forgets, the rows of stream s that became complete since the last call.
  std::vector<FrameRow> completedRows(int s);
  const RoundTimes& lastRoundTimes() const { return times_; }
  size_t keyframesAlive(int s) const;
  // waits for the depth-filter update in flight and completes the last rows
  void finish();
  // device calls per round of the last addImages, for the record (one per stage, whatever the number of streams)
  int lastRoundDeviceCalls() const { return device_calls_; }
  // where the rounds' time went, phase by phase (sums over all rounds since construction, ms), with the phases' names
  static constexpr int kNumPhases = 32;
  const double* phaseTimes() const { return phase_ms_; }
  static const char* phaseName(int k);

 private:
  struct Stream;
  void finishSeedUpdate();
  // the structure optimisation queued in the round before (side stream): wait for it, every stream's points take their positions
  void finishStructure();
  size_t structure_in_flight_ = 0;   // points of the batch in flight
  std::vector<int> structure_streams_;
  void startDetection(const std::vector<int>& which);
  void makeKeyframes(const std::vector<int>& which);
  // the detector batch of the round's new keyframes between its two halves
  struct DetectBatch {
    std::vector<int> started_for, streams;   // the streams it was started for; those among them whose frame has room for seeds (slot order)
    std::vector<uint8_t> occ;
    std::vector<uint64_t> ckeys, ekeys;
    std::vector<float> angles;
    bool in_flight = false;
  } detect_;
  const bool detect_ahead_ = true;   // the detector of a round's periodic keyframes ahead of the pose optimisation
  const bool align_ahead_ = true;    // the alignment queued ahead of the wait for the previous round's seed update
  bool speculate_never_ = false;
  bool speculate_all_ = false;   // plan every stream's unconverged-seed list whether or not its pass was reached on the frame before
  void drainReleases();
  void check(int rc, const char* what) const;

  // the engine's host phases: on its own pool, or on the pool it shares with other engines
  struct Runner {
    std::unique_ptr<WorkerPool> own;
    std::shared_ptr<SharedPool> shared;
    std::shared_ptr<ExclusivePool> exclusive;
    int seed = 0;
    void run(int n_items, const std::function<void(int)>& fn)
    {
      if (exclusive) exclusive->run(n_items, fn);
      else if (shared) shared->run(n_items, fn, seed);
      else own->run(n_items, fn);
    }
  };
  svoh_ctx* ctx_;
  LockstepOptions opt_;
  Runner pool_;
  std::vector<std::unique_ptr<Stream>> streams_;
  size_t round_ = 0;
  bool seeds_in_flight_ = false;
  const bool pose_chain_ = true;   // the seed update queued behind the pose kernel, its poses taken on the device
  RoundTimes times_;
  int device_calls_ = 0;
  double phase_ms_[kNumPhases] = {};
  // device pyramids to give up: frames die on whatever thread drops their last reference, the context is single-threaded
  std::mutex release_mu_;
  std::vector<svoh_frame_t> to_release_;
  std::vector<svoh_features_t> features_to_release_;
  // the seed update in flight: where each stream's slice of the staged batch lies
  svoh_matcher_stage_t seed_stage_{};
  // the next round's pyramids, if the caller handed its images in early
  std::vector<svoh_frame_t> prefetched_;
  std::vector<const uint8_t*> prefetched_from_;
  void prefetch(const uint8_t* const* next_images, int pitch);
};

}  // namespace svo_hip
