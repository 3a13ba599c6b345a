// svo_hip_lockstep_stereo.h -- many STEREO camera streams (BASELINE config 3 x config 5) through the per-pair chain of
// FrameHandlerStereo::processFrame (src/svo/src/frame_handler_stereo.cpp:57-205, frame_handler_base.cpp:610-825) in LOCK STEP:
// one pair of every stream at a time, every per-pair stage ONE launch for all of them.
//
//   pyramids            one call for the 2 S images                                       svoh_build_pyramid_multi
//   sparse alignment    S two-camera bundles (8 parameters: pose + illumination gain /    svoh_sparse_align_enqueue_keyed
//                       offset) with each stream's IMU rotation prior, grouped by geometry
//   reprojection        first every stream's left camera, then every stream's right one   svoh_matcher_stage + svoh_match_direct_batch /
//                       (a stream's right camera sees what its left one did to the        svoh_update_seeds_batch_ex, twice per round
//                       landmarks' statistics and to the seeds: reprojector.cpp:342-486)
//   pose optimisation   S rig bundles                                                     svoh_optimize_pose_batch
//   structure optim.    the left frames' landmarks of all streams, then the right ones    svoh_optimize_points_batch, twice per round
//   depth filter        the keyframes' seeds into every stream's left frame, then into    svoh_update_seeds_batch, twice per round (the
//                       its right one (the second update starts from the first's states)  second left in flight until the next round)
//   keyframes           the streams that make one: both detector runs, the stereo              svoh_detect_cells_batch, svoh_epipolar_match_batch,
//                       triangulation's epipolar searches, the refreshed edgelet           svoh_histogram_angle_bins, svoh_features_upload:
//                       directions, the resident columns -- one call each for all          one call each per round
//
// A stream's host work is the SAME code tools/svoh_mini_stereo.cpp runs for one stream (the mirrors' phase interfaces); every
// kernel's per-unit result does not depend on what shares its launch, every alignment problem runs in the launch geometry it would
// get alone: a stream's trajectory and counters are those of its single-stream run, byte for byte (tests/test_mini_stereo_gpu.py).
// Like svoh_mini_stereo this is an integration harness above the mirrors, NOT the reference's frame handler: no map (every live
// keyframe counts as overlapping), no initialiser (the first rig pose is given), keyframes by a fixed rule.
#pragma once

#include <deque>
#include <memory>
#include <utility>
#include <vector>

#include "svo_hip_io.h"
#include "svo_hip_pool.h"

namespace svo_hip {

struct StereoLockstepOptions {
  io::FrontendParams params;            // (illumination gain / offset are estimated, the matcher's gain as well: euroc_stereo_imu.yaml:30-31, as svoh_mini_stereo)
  std::vector<io::RigCamera> rig;       // the two cameras of every stream's rig ...
  // ... or a rig per stream (one entry per stream, two cameras each; empty: `rig` for all): intrinsics, distortion, extrinsics / baseline may
  // differ, the image size may not (the streams' pyramids are one call).  The reference's process-wide static thresholds are taken from
  // rig[0] (svo_hip::fixProcessWideThresholds)
  std::vector<std::vector<io::RigCamera>> per_stream_rig;
  size_t kf_every = 8;
  double lambda_rot = 0.5;              // img_align_prior_lambda_rot
  int n_workers = 1;                    // host threads, the caller included
  bool landmarks = true;                // upgradeSeedsToFeatures at keyframes, optimizeStructure every pair (as svoh_mini_stereo)
  int images_mem_space = SVOH_MEM_HOST; // SVOH_MEM_HOST_PINNED: the images live in svoh_host_alloc memory (one gather kernel reads all of them over PCIe)
  // a keyframe pair's constant feature columns are uploaded once (svoh_features_upload); the depth-filter batches of the pairs after it are
  // whole resident sets (SVOH_BATCH_WHOLE_SETS): per seed only state and type cross PCIe.  false: explicit columns every pair (tests)
  bool resident_features = true;
  bool speculate_all = false;           // every camera's third candidate list (its unconverged seeds) joins the round's batch whether or not its pass was reached before (tests)
};

class FrontendLockstepStereo {
 public:
  // what a stream's pair left behind (the counter columns of svoh_mini_stereo's frontend.csv)
  struct PairRow { size_t k = 0; bool is_kf = false; size_t n_aligned = 0, n_reproj = 0, n_pose = 0, n_seed_upd = 0, n_landmarks = 0; double alpha = 0, beta = 0; };
  FrontendLockstepStereo(svoh_ctx* ctx, int n_streams, const StereoLockstepOptions& options);
  ~FrontendLockstepStereo();
  FrontendLockstepStereo(const FrontendLockstepStereo&) = delete;
  FrontendLockstepStereo& operator=(const FrontendLockstepStereo&) = delete;
  int numStreams() const { return static_cast<int>(streams_.size()); }
  // One pair of every stream: left[s] / right[s] = level 0 of the two images (the cameras' size, `pitch` bytes per row; both NULL: stream s
  // has no pair this round).  A stream's first pair makes its first keyframes at T_imu_world_first[s].  imu_prior[s] (the array or an
  // entry may be NULL): R_imu(k)_imu(k-1) of the stream's new pair, the rotation the alignment's prior is built from.
  // next_left / next_right (may be NULL): the pairs of the NEXT call, when the caller knows them already -- their pyramids are built during
  // this round on a second stream, beside the chain's work (svoh_build_pyramid_multi_prefetch: 2 S images cross PCIe per round); the next
  // call must then bring exactly these images.
  void addPairs(const uint8_t* const* left, const uint8_t* const* right, int pitch, const Transformation* T_imu_world_first, const svoh::Quat* const* imu_prior,
                const uint8_t* const* next_left = nullptr, const uint8_t* const* next_right = nullptr);
  Transformation pose(int s) const;     // T_imu_world of stream s' newest pair
  // rows are complete once the pair's second seed update has been finished (at the start of the next addPairs, or in finish())
  std::vector<PairRow> completedRows(int s);
  size_t keyframesAlive(int s) const;
  void finish();
  int lastRoundDeviceCalls() const { return device_calls_; }
  size_t pausedPasses() const { return paused_passes_; }   // replays that reached a pass nobody had planned (since construction)
  // where the rounds' time went (ms summed since construction): pyramids, finish seeds, align, reproject, pose, structure, keyframes, seed updates
  static constexpr int kNumPhases = 8;
  const double* phaseTimes() const { return phase_ms_; }
  // ... and inside "align" and "reproject": align prep / launch / wait / finish, walk + plan, match stage + submit, match wait, replay
  static constexpr int kNumDetails = 8;
  const double* detailTimes() const { return detail_ms_; }
  static const char* detailName(int k) { static const char* n[] = { "align prep", "align launch", "align wait", "align finish", "walk + plan", "match stage + submit", "match wait", "replay" }; return k >= 0 && k < kNumDetails ? n[k] : ""; }
  static const char* phaseName(int k) { static const char* n[] = { "pyramids", "finish seeds", "align", "reproject", "pose", "structure", "keyframes", "seed updates" }; return k >= 0 && k < kNumPhases ? n[k] : ""; }

 private:
  struct Stream;
  void check(int rc, const char* what) const;
  void finishSecondSeedUpdate();
  void makeKeyframes(const std::vector<std::pair<int, size_t>>& which);
  void drainReleases();
  const std::vector<io::RigCamera>& rigOf(int s) const { return opt_.per_stream_rig.empty() ? opt_.rig : opt_.per_stream_rig[static_cast<size_t>(s)]; }
  // one depth-filter update of the tracking streams' visible keyframes into their camera c: blocking (collected at once) or left in flight
  void seedUpdate(const std::vector<int>& trk, int c, bool leave_in_flight);
  void collectSeedUpdate();
  svoh_ctx* ctx_;
  StereoLockstepOptions opt_;
  std::unique_ptr<WorkerPool> pool_;
  std::vector<std::unique_ptr<Stream>> streams_;
  int device_calls_ = 0;
  size_t paused_passes_ = 0;
  double phase_ms_[kNumPhases] = {};
  double detail_ms_[kNumDetails] = {};
  bool seeds_in_flight_ = false;
  // the seed batch in flight: its streams, and where it was staged (the context's page-locked area: valid until the next stage call)
  std::vector<int> sb_streams_;
  svoh_matcher_stage_t seed_stage_{};
  std::mutex release_mu_;
  std::vector<svoh_frame_t> to_release_;
  std::vector<svoh_features_t> features_to_release_;
  // the next round's pyramids, announced and under way (in the order left(s), right(s) of the streams that will have a pair)
  std::vector<svoh_frame_t> prefetched_;
  std::vector<const uint8_t*> prefetched_from_;
  void prefetch(const uint8_t* const* next_left, const uint8_t* const* next_right, int pitch);
};

}  // namespace svo_hip
