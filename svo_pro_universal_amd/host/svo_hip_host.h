// svo_hip_host.h -- C++ host layer above the C ABI (include/svo_hip.h) that
// mirrors the reference's interface for the sparse-image-alignment seam:
//
//   svo::SparseImgAlignBase / svo::SparseImgAlign
//       src/svo_img_align/include/svo/img_align/sparse_img_align_base.h:59-163
//       src/svo_img_align/include/svo/img_align/sparse_img_align.h:30-89
//   the calls FrameHandlerBase makes on it
//       src/svo/src/frame_handler_base.cpp:621-634 (reset, setWeightedPrior,
//       setMaxNumFeaturesToAlign, run) and :1247 (setCompensation)
//
// Same names, argument meaning and return values.  The reference's Frame /
// FrameBundle / Transformation depend on Eigen + OpenCV, which this image does
// not have, so light POD mirrors of exactly the members run() reads are
// defined here; the adapter that subclasses the real svo::SparseImgAlignBase is
// in INTEGRATION.md.  No CPU fallback: every call goes to libsvo_hip.so.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/svo_hip.h"
#include "../csrc/svoh_math.h"

namespace svo_hip {

namespace detail { struct SpeculativeMatches; struct Resolved; }   // svo_hip_host_internal.h

using Transformation = svoh::Rigid;  // minkindr QuatTransformation semantics (svoh_math.h)

// The members of svo::Frame that SparseImgAlign::run reads (frame.h:46-73, 252-306).
struct Frame {
  svoh_frame_t pyramid = 0;      // device copy of img_pyr_
  svoh_features_t features = 0;  // device copy of px_vec_ / f_vec_ / grad_vec_ / level_vec_ of a keyframe whose features are final (svoh_features_upload), or 0
  svoh_camera cam{};             // cam()
  Transformation T_f_w_{ {1, 0, 0, 0}, {0, 0, 0} };
  Transformation T_cam_imu_{ {1, 0, 0, 0}, {0, 0, 0} };
  Transformation T_imu_cam_{ {1, 0, 0, 0}, {0, 0, 0} };
  size_t num_features_ = 0;
  std::vector<double> px_vec_;       // 2 x n
  std::vector<double> f_vec_;        // 3 x n
  // landmark_vec_[i]->pos_ or the seed position in the world (sparse_img_align.cpp:281-292)
  std::vector<double> pos_world_;    // 3 x n
  // landmark or seed reference present and not a map point (sparse_img_align.cpp:239-245)
  std::vector<uint8_t> alignable_;   // n
  std::vector<int32_t> pos_seed_unit_;   // empty, or n: see resolveAlignmentPoints(frame, unit_of)
  // members read / written by the matcher and the depth filter (frame.h:62-73, 160-170)
  std::vector<double> grad_vec_;                 // 2 x n
  std::vector<int32_t> level_vec_;               // n
  std::vector<uint8_t> type_vec_;                // n  svo::FeatureType
  std::vector<double> invmu_sigma2_a_b_vec_;     // 4 x n  seed states
  double seed_mu_range_ = 0.0;
  int id_ = 0;
  int id() const { return id_; }
  // members the reprojector writes for a newly matched feature (frame.h:62-86); optional otherwise
  std::vector<double> score_vec_;                       // n
  std::vector<int> track_id_vec_;                       // n (FeatureTracker)
  std::vector<std::shared_ptr<struct Point>> landmark_vec_;   // n, nullptr = no landmark
  struct SeedRef { std::shared_ptr<Frame> keyframe; int seed_id = -1; };
  std::vector<SeedRef> seed_ref_vec_;                   // n
  // five landmarks (closest to the centre and to the four corners) that decide whether two frames have
  // overlapping fields of view (frame.h:49-50, frame.cpp:171-227); first = feature index or -1
  struct KeyPoint { int first = -1; svoh::Vec3 second{ 0, 0, 0 }; };
  KeyPoint key_pts_[5];
  void resetKeyPoints() { for (KeyPoint& k : key_pts_) k = KeyPoint(); }   // frame.cpp:218-221
  void setKeyPoints();
  double getSeedDepth(size_t idx) const { return 1.0 / invmu_sigma2_a_b_vec_[4 * idx]; }   // seed.h:110-113 (inverse depth)
  size_t numTrackedFeatures() const;                                      // frame.h:153-163
  bool isVisible(const svoh::Vec3& xyz_w, double* px /* 2, may be NULL */) const;   // frame.cpp:229-260
  // isVisible's cosine of the image corner's off-axis angle: a function of the intrinsics alone, which the
  // reference recomputes (an undistortion included) on every call; kept here until `cam` changes
  mutable svoh_camera min_cos_cam_{};
  mutable double min_cos_ = 0.0;
  mutable bool min_cos_valid_ = false;

  void set_T_cam_imu(const Transformation& T) { T_cam_imu_ = T; T_imu_cam_ = svoh::inverse(T); }  // frame.h:270-274
  const Transformation& T_cam_imu() const { return T_cam_imu_; }
  const Transformation& T_imu_cam() const { return T_imu_cam_; }
  Transformation T_imu_world() const { return svoh::mul(T_imu_cam_, T_f_w_); }  // frame.h:267
  svoh::Vec3 pos() const { return svoh::inverse(T_f_w_).t; }                    // frame.h:306
};
using FramePtr = std::shared_ptr<Frame>;

struct FrameBundle {
  std::vector<FramePtr> frames_;
  bool empty() const { return frames_.empty(); }
  size_t size() const { return frames_.size(); }
  const FramePtr& at(size_t i) const { return frames_.at(i); }
  using Ptr = std::shared_ptr<FrameBundle>;
};

// ---------------------------------------------------------------------------
// Device pyramids of host frames.  The reference's Frame owns img_pyr_ (cv::Mat) and dies with its last
// shared_ptr; its device copy must die with it or a long sequence leaks one slab per frame.  The cache keeps
// one handle per live frame object, keyed by the object's address with a weak_ptr as the liveness witness (no
// change to the Frame class needed), builds the device pyramid on first use and releases the handles of expired
// frames on every call (`sweep`).  The adapter of INTEGRATION.md uses exactly this class with svo::Frame.
// ---------------------------------------------------------------------------
class DeviceFrameCache {
 public:
  explicit DeviceFrameCache(svoh_ctx* ctx) : ctx_(ctx) {}
  ~DeviceFrameCache() { clear(); }
  DeviceFrameCache(const DeviceFrameCache&) = delete;
  DeviceFrameCache& operator=(const DeviceFrameCache&) = delete;
  // handle of `frame`'s device pyramid; `make` uploads / builds it when the frame is seen for the first time
  template <class FrameT, class Make>
  svoh_frame_t get(const std::shared_ptr<FrameT>& frame, Make make)
  {
    sweep();
    const void* key = frame.get();
    auto it = entries_.find(key);
    if (it != entries_.end() && !it->second.alive.expired()) return it->second.handle;
    if (it != entries_.end()) { release(it->second.handle); entries_.erase(it); }   // the address was reused by a new frame
    const svoh_frame_t h = make(*frame);
    entries_[key] = Entry{ std::weak_ptr<const void>(std::shared_ptr<const void>(frame, frame.get())), h };
    return h;
  }
  // release the device pyramids of frames that no longer exist; returns how many were released
  size_t sweep();
  void clear();
  size_t size() const { return entries_.size(); }

 private:
  struct Entry { std::weak_ptr<const void> alive; svoh_frame_t handle; };
  void release(svoh_frame_t h);
  svoh_ctx* ctx_;
  std::unordered_map<const void*, Entry> entries_;
};

// sparse_img_align_base.h:37-46
struct SparseImgAlignOptions {
  int max_level = 4;
  int min_level = 1;
  bool estimate_illumination_gain = false;
  bool estimate_illumination_offset = false;
  bool use_distortion_jacobian = false;
  bool robustification = false;
  double weight_scale = 10;
};

// the two solver options the reference sets (sparse_img_align_base.cpp:35-42)
struct SolverOptions {
  size_t max_iter = 10;
  double eps = 0.0005;
};

// Fills Frame::pos_world_ / alignable_ the way sparse_img_align.cpp:239-245 and :281-292 resolve them: the
// landmark's position, or the seed's position via seed_ref_vec_; neither, or a map point -> not alignable.
void resolveAlignmentPoints(Frame& frame);
// The same while a depth-filter update is still in flight on the device: unit_of(keyframe, seed_id) >= 0 names the unit of that
// staged batch that holds the seed -- the feature's position is then left to the device (Frame::pos_seed_unit_ ->
// svoh_align_camera::pos_seed_unit: it reads the inverse depth the update leaves), pos_world_ holds the value from the seed's
// state before the update and is not used for it.  Everything else as above.
void resolveAlignmentPoints(Frame& frame, const std::function<int32_t(const Frame& keyframe, size_t seed_id)>& unit_of);

class SparseImgAlignHip {
 public:
  using Ptr = std::shared_ptr<SparseImgAlignHip>;

  SparseImgAlignHip(svoh_ctx* ctx, SolverOptions solver_options, SparseImgAlignOptions options);

  static SolverOptions getDefaultSolverOptions() { return SolverOptions(); }

  // MiniLeastSquaresSolver::reset (mini_least_squares_solver.hpp:240-250): drops the prior
  void reset();
  void setWeightedPrior(const Transformation& T_cur_ref_prior, double alpha_prior, double beta_prior,
                        double lambda_rot, double lambda_trans, double lambda_alpha, double lambda_beta);
  void setMaxNumFeaturesToAlign(int num) { max_num_features_ = num; }  // ignored by the reference too (SURVEY gotcha 12)
  void setAlphaInitialValue(double a) { alpha_init_ = a; }
  void setBetaInitialValue(double b) { beta_init_ = b; }
  void setCompensation(bool do_compensation);
  void setPatchSize(int patch_size) { patch_size_ = patch_size; }

  // SparseImgAlign::run: optimises the pose of cur_frames (writes T_f_w_ of every
  // frame of the bundle) and returns the number of features tracked; 0 = none.
  // Throws std::runtime_error on an ABI error (the reference CHECK-aborts).
  size_t run(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames);
  // The same with a hook between the launch and the wait for it: after_enqueue(T_iref_world) runs while the
  // alignment kernel does, and device work it queues on the context (ReprojectorHip::enqueueCandidateProjection)
  // runs right behind the alignment and comes back with the same round trip.  lastRunRepeated(): the launch had to
  // be repeated (a cluster of workgroups gave up, svoh_sparse_align_batch's fallback), so whatever was queued behind
  // the first launch has used a pose that is not the result's and must be discarded by the caller.
  using AfterEnqueue = std::function<void(const Transformation& T_iref_world)>;
  size_t run(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames, const AfterEnqueue& after_enqueue);
  bool lastRunRepeated() const { return last_run_repeated_; }
  // run() in two halves, for a driver that puts the problems of MANY streams into one launch (FrontendLockstep):
  // prepareRun builds this stream's problem (and returns T_iref_world), the driver launches and fetches, finishRun does
  // what run() does with the result -- T_f_w_ of every current frame, alpha / beta reset -- and returns run()'s value.
  Transformation prepareRun(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames, svoh_align_options& opt, svoh_align_problem& pb) const
  {
    return buildProblem(ref_frames, cur_frames, 0, 1, opt, pb);
  }
  size_t finishRun(const svoh_align_result& result, const FrameBundle::Ptr& cur_frames, const Transformation& T_iref_world);

  // The same optimisation with the patches split over `world` participants (GPUs): this one takes share `rank`
  // of every camera's features, and between evaluateError and the solve the 74 doubles at d_sums (DEVICE memory)
  // are summed over the participants by the callback -- ncclAllReduce(d_sums, d_sums, n, ncclDouble, ncclSum, ...)
  // followed by a stream synchronisation in a multi-GPU host, nothing for world == 1.  Every participant computes
  // the same update and ends in the same state (second row).  Returns the number of patches
  // visible in the first evaluation, summed over the participants (0 = nothing to track).
  using SumOverParticipants = std::function<void(double* d_sums, size_t n)>;
  size_t runSplit(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames, int rank, int world,
                  const SumOverParticipants& sum_over_participants);

  // last run's statistics
  const svoh_align_result& lastResult() const { return last_; }

 private:
  Transformation buildProblem(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames, int rank, int world,
                              svoh_align_options& opt, svoh_align_problem& pb) const;
  svoh_ctx* ctx_;
  SolverOptions solver_options_;
  SparseImgAlignOptions options_;
  int patch_size_ = 4;
  int max_num_features_ = -1;
  double alpha_init_ = 0.0, beta_init_ = 0.0;
  svoh_align_prior prior_{};
  svoh_align_result last_{};
  bool last_run_repeated_ = false;
};

// ---------------------------------------------------------------------------
// Seam 2: depth filter.  Mirrors svo::DepthFilter::updateSeeds
// (src/svo_direct/include/svo/direct/depth_filter.h:158-160,
//  src/svo_direct/src/depth_filter.cpp:200-233) and the options it reads
// (depth_filter.h:40-100, matcher.h:39-54).
// ---------------------------------------------------------------------------
struct DepthFilterOptions {
  double seed_convergence_sigma2_thresh = 200.0;
  double mappoint_convergence_sigma2_thresh = 500.0;
  bool scan_epi_unit_sphere = false;   // svo_factory.cpp:261
  bool affine_est_offset = true;
  bool affine_est_gain = false;
  bool use_threaded_depthfilter = false;  // must stay false: the threaded variant races
};

class DetectorHip;
namespace depth_filter_utils {
// depth_filter_utils::initializeSeeds (depth_filter.cpp:255-365): detect features in the free cells of the
// detector's grid (the caller has marked the occupied ones) and append them to the frame as seeds with
// mu = 1 / depth_mean, sigma2 = mu_range^2 / 36, a = b = 10, mu_range = 1 / depth_min (seed.h:130-145)
void initializeSeeds(const FramePtr& frame, DetectorHip& feature_detector, size_t max_n_seeds, float depth_min, float depth_max,
                     float depth_mean);
// its second half: the detected features (px, score, level, grad, type) appended to the frame as seeds
void appendSeeds(const FramePtr& frame, const std::vector<double>& px, const std::vector<double>& score, const std::vector<int32_t>& level,
                 const std::vector<double>& grad, const std::vector<uint8_t>& type, float depth_min, float depth_mean);
}  // namespace depth_filter_utils

class DepthFilterHip {
 public:
  DepthFilterHip(svoh_ctx* ctx, const DepthFilterOptions& options);
  // returns the number of successfully updated seeds; updates invmu_sigma2_a_b_vec_ and
  // type_vec_ of the reference frames in place, exactly like the reference
  size_t updateSeeds(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame);
  // The same in two halves: updateSeedsAsync sends the update to the device and returns (the reference frames must
  // not be touched by anything that reads or writes their seeds until the update is finished -- appending features to
  // them is fine); finishUpdateSeeds waits for it, writes the states and types back and returns the success count.
  // Nothing in a frame's chain needs the updated seeds before the NEXT frame's alignment, so a caller that finishes
  // there takes the seed update (kernel + round trip) off the per-frame critical path.  The update holds the context's
  // ONE deferred matcher section: a ReprojectorHip::reprojectFrames on the same context in between finishes it first
  // (finishPendingSeedUpdate; the later finishUpdateSeeds then only hands out the count).  Caveat: anything that
  // synchronises the context's stream in between also waits for the update's kernel -- on a keyframe the detector
  // (svoh_detect_features) and frame releases do, so the saving is a non-keyframe saving.
  void updateSeedsAsync(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame);
  size_t finishUpdateSeeds();
  // The first half of updateSeedsAsync, for a caller that knows the update's inputs before it knows the current frame's
  // final pose (the per-frame chain: reprojection done, pose optimisation running): the seeds and their states are
  // staged and uploaded now; updateSeedsAsync(same frames, cur_frame) then only replaces the current frame's view and
  // sends the kernel off.  Between the two calls nothing but cur_frame's pose may change.
  void prepareUpdateSeeds(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame);
  // the unit of the update in flight (updateSeedsAsync, not finished yet) that holds seed `seed_id` of `keyframe`, or -1: for
  // resolveAlignmentPoints(frame, unit_of) -- the next frame's alignment queued before this update's results are back
  int32_t unitOfPendingSeed(const Frame& keyframe, size_t seed_id) const;
  bool updateInFlight() const { return async_open_ && !prepared_; }
  void finishUpdateSeedsEarly();   // what finishPendingSeedUpdate calls
  bool updatePending() const { return async_open_; }
  ~DepthFilterHip();
  DepthFilterHip(const DepthFilterHip&) = delete;
  DepthFilterHip& operator=(const DepthFilterHip&) = delete;
  svoh_matcher_options& getMatcherOptions() { return matcher_options_; }
  // the options updateSeeds passes to the device for an update into cur_frame (depth_filter.cpp:224-225; the
  // function-local static px_error_angle of updateSeed): for a driver that batches the updates of many streams itself
  svoh_depth_filter_options abiOptions(const Frame& cur_frame);
  // Matcher::MatchResult of every seed of the last call, in (frame, feature) order
  const std::vector<int32_t>& lastMatchResults() const { return last_results_; }

 private:
  svoh_ctx* ctx_;
  DepthFilterOptions options_;
  svoh_matcher_options matcher_options_{};
  bool have_px_error_angle_ = false;   // the function-local static of updateSeed (depth_filter.cpp:383-384)
  double px_error_angle_ = 0.0;
  std::vector<int32_t> last_results_;
  struct Pending {   // the arrays of the queued update: alive until finishUpdateSeeds
    std::vector<FramePtr> frames;
    std::vector<size_t> counts;
    std::vector<svoh_frame_view> refs;
    std::vector<int32_t> ref_idx, level;
    std::vector<double> px, f, grad, state;
    std::vector<uint8_t> type, success;
    int32_t n_success = 0;
  } pending_;
  bool async_open_ = false;
  bool prepared_ = false;          // prepareUpdateSeeds has queued the batch; updateSeedsAsync has not sent it off yet
  const Frame* prepared_cur_ = nullptr;
  bool finished_early_ = false;
  size_t early_count_ = 0;
  size_t finishUpdateSeedsNow();
  void queueUpdateSeeds(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame, bool send_off);
};

// Finishes the seed update that a DepthFilterHip has in flight on this context, if there is one (see updateSeedsAsync).
void finishPendingSeedUpdate(svoh_ctx* ctx);

// ---------------------------------------------------------------------------
// Seam 3: KLT.  Mirrors feature_alignment::alignPyr2DVec
// (src/svo_direct/include/svo/direct/feature_alignment.h:59-69): cv::Point2f becomes Point2f.
// ---------------------------------------------------------------------------
struct Point2f { float x, y; };
namespace feature_alignment {
void alignPyr2DVec(svoh_ctx* ctx, svoh_frame_t img_pyr_ref, svoh_frame_t img_pyr_cur, int max_level, int min_level,
                   const std::vector<int>& patch_sizes, int n_iter, float min_update_squared,
                   const std::vector<Point2f>& px_ref, std::vector<Point2f>& px_cur, std::vector<uint8_t>& status);
}

// FeatureTracker (src/svo_tracker/include/svo/tracker/feature_tracker.h,
// feature_tracking_types.h:15-139): trackFrameBundle with the tracks of the whole bundle in
// ONE batched launch; the bookkeeping (pushBack, terminated tracks, px / score / track id /
// bearing vectors of the new frame) is the reference's, run on the host afterwards.
struct FeatureTrackerOptions {   // feature_tracking_types.h:15-49
  int klt_max_level = 4;
  int klt_min_level = 0;
  std::vector<int> klt_patch_sizes = { 16, 16, 16, 8, 8 };
  int klt_max_iter = 30;
  double klt_min_update_squared = 0.001;
  bool klt_template_is_first_observation = true;
  size_t min_tracks_to_detect_new_features = 50;
  bool reset_before_detection = true;
};

class FeatureRef {
 public:
  FeatureRef(const FrameBundle::Ptr& frame_bundle, size_t frame_index, size_t feature_index)
      : frame_bundle_(frame_bundle), frame_index_(frame_index), feature_index_(feature_index) {}
  const FrameBundle::Ptr& getFrameBundle() const { return frame_bundle_; }
  size_t getFrameIndex() const { return frame_index_; }
  size_t getFeatureIndex() const { return feature_index_; }
  const double* getPx() const { return &getFrame()->px_vec_[2 * feature_index_]; }
  const FramePtr& getFrame() const { return frame_bundle_->at(frame_index_); }
 private:
  FrameBundle::Ptr frame_bundle_;
  size_t frame_index_, feature_index_;
};

class FeatureTrack {
 public:
  explicit FeatureTrack(int track_id) : track_id_(track_id) {}
  int getTrackId() const { return track_id_; }
  size_t size() const { return feature_track_.size(); }
  bool empty() const { return feature_track_.empty(); }
  const FeatureRef& front() const { return feature_track_.front(); }   // first observation
  const FeatureRef& back() const { return feature_track_.back(); }     // last observation
  const FeatureRef& at(size_t i) const { return feature_track_.at(i); }
  void pushBack(const FrameBundle::Ptr& frame_bundle, size_t frame_index, size_t feature_index)
  {
    feature_track_.emplace_back(frame_bundle, frame_index, feature_index);
  }
 private:
  int track_id_;
  std::vector<FeatureRef> feature_track_;
};
using FeatureTracks = std::vector<FeatureTrack>;

class DetectorHip;

class FeatureTrackerHip {
 public:
  FeatureTrackerHip(svoh_ctx* ctx, const FeatureTrackerOptions& options, size_t bundle_size);
  // one detector per camera of the bundle (the reference builds them from DetectorOptions in its constructor)
  void setDetectors(const std::vector<std::shared_ptr<DetectorHip>>& detectors) { detectors_ = detectors; }
  // feature_tracker.cpp:24-50: track, and when fewer than min_tracks_to_detect_new_features survive,
  // (optionally reset and) detect new features and start a track for each
  void trackAndDetect(const FrameBundle::Ptr& nframe_kp1);
  // feature_tracker.cpp:124-182 with the detectors set above
  size_t initializeNewTracks(const FrameBundle::Ptr& nframe);
  // feature_tracker.cpp:52-122; returns getTotalActiveTracks()
  size_t trackFrameBundle(const FrameBundle::Ptr& nframe_kp1);
  // The track-creating tail of initializeNewTracks (feature_tracker.cpp:168-178) for the features
  // [n_old, num_features_) of every frame of the bundle; detection itself is SURVEY.md 8(f-2).
  size_t initializeNewTracks(const FrameBundle::Ptr& nframe, const std::vector<size_t>& n_old_per_frame);
  const FeatureTracks& getActiveTracks(size_t frame_index) const { return active_tracks_.at(frame_index); }
  const FeatureTracks& getTerminatedTracks(size_t frame_index) const { return terminated_tracks_.at(frame_index); }
  size_t getTotalActiveTracks() const;
  void resetActiveTracks() { for (auto& t : active_tracks_) t.clear(); }
  void resetTerminatedTracks() { for (auto& t : terminated_tracks_) t.clear(); }
  void reset() { resetActiveTracks(); resetTerminatedTracks(); }
 private:
  svoh_ctx* ctx_;
  FeatureTrackerOptions options_;
  size_t bundle_size_;
  std::vector<FeatureTracks> active_tracks_, terminated_tracks_;
  std::vector<std::shared_ptr<DetectorHip>> detectors_;
  int next_track_id_ = 0;   // PointIdProvider::getNewPointId()
};

// ---------------------------------------------------------------------------
// Seam 2b: reprojector.  Mirrors reprojector_utils::matchCandidates
// (src/svo/include/svo/reprojector.h, src/svo/src/reprojector.cpp:342-382) and what it
// touches: OccupandyGrid2D (svo_common/include/svo/common/occupancy_grid_2d.h:10-113),
// Reprojector::Candidate / Statistics, Point::getCloseViewObs (point.cpp:83-129).
// The per-candidate matcher work of matchCandidate (:384-486) runs speculatively for all
// candidates in two batched launches; the reference's loop -- grid occupancy, n_trials /
// n_matches, num_features_, the early break, n_failed_reproj_ / n_succeeded_reproj_, the
// in-place seed updates -- is then replayed on the host in candidate order, and only the
// candidates that loop visits leave side effects.
// ---------------------------------------------------------------------------
struct Point {
  svoh::Vec3 pos_{ 0, 0, 0 };
  int id_ = 0;
  int n_failed_reproj_ = 0, n_succeeded_reproj_ = 0;
  struct Obs { std::weak_ptr<Frame> frame; size_t keypoint_index_ = 0; };
  std::vector<Obs> obs_;
  std::vector<int> last_projected_kf_id_ = std::vector<int>(SVOH_MAX_CAMS, -1);   // per camera (point.h)
  int last_structure_optim_ = 0;   // frame id of the last optimizeStructure that touched the point (point.h)
  const svoh::Vec3& pos() const { return pos_; }
  int id() const { return id_; }
  bool getCloseViewObs(const svoh::Vec3& framepos, FramePtr& ref_frame, size_t& ref_feature_index) const;
};
using PointPtr = std::shared_ptr<Point>;

struct OccupandyGrid2D {
  OccupandyGrid2D(int cell, int cols, int rows)
      : cell_size(cell), n_cols(cols), n_rows(rows), occupancy_(static_cast<size_t>(cols) * rows, false) {}
  static int getNCell(int n_pixels, int size);
  const int cell_size, n_cols, n_rows;
  std::vector<bool> occupancy_;
  void reset() { std::fill(occupancy_.begin(), occupancy_.end(), false); }
  size_t size() const { return occupancy_.size(); }
  bool isOccupied(size_t cell_index) const { return occupancy_.at(cell_index); }
  void setOccupied(size_t cell_index) { occupancy_.at(cell_index) = true; }
  int numOccupied() const;
  size_t getCellIndex(int x, int y, int scale = 1) const;
};

namespace reprojector {
struct Candidate {   // Reprojector::Candidate (reprojector.h)
  FramePtr ref_frame;
  size_t ref_index = 0;
  double cur_px[2] = { 0, 0 };
  int n_reproj = 0;      // n_succeeded_reproj_ - n_failed_reproj_ of the landmark
  uint8_t type = 0;      // svo::FeatureType of the reference feature when the candidate was made
  double score = 0.0;
  size_t n_obs = 0;
};
struct Statistics { size_t n_matches = 0, n_trials = 0; };
}  // namespace reprojector

// ReprojectorOptions / Reprojector::reprojectFrames (src/svo/include/svo/reprojector.h:26-165,
// src/svo/src/reprojector.cpp:27-306): candidate generation from the landmarks and seeds of
// the visible keyframes, three matchCandidates passes (landmarks, converged seeds, unconverged seeds).  The
// SVO_GLOBAL_MAP block (fixed landmarks, :51-129) is not compiled in the reference's default build and not mirrored.
struct ReprojectorOptions {
  size_t max_n_features_per_frame = 120;
  size_t cell_size = 30;
  size_t max_n_kfs = 5;
  bool reproject_unconverged_seeds = true;
  double max_unconverged_seeds_ratio = -1.0;
  size_t min_required_features = 0;
  double seed_sigma2_thresh = 200;
  bool remove_unconstrained_points = true;
  bool affine_est_offset = true;
  bool affine_est_gain = false;
};

class ReprojectorHip {
 public:
  ReprojectorHip(svoh_ctx* ctx, const ReprojectorOptions& options, size_t camera_index);
  void reprojectFrames(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs, std::vector<PointPtr>& trash_points);
  // f-4 on the device: the arithmetic of reprojector_utils::getCandidate -- seed position, projection into
  // cur_frame, visibility cone, image box, 8-pixel margin -- for EVERY feature of `kfs` (the local map: which of them
  // are visible is only known once the pose is), queued on the context without a wait (svoh_project_candidates_enqueue).
  // Called from SparseImgAlignHip::run's after_enqueue hook with align_result_index >= 0, the pose is composed on the
  // device from the alignment's result and the projection comes back with the alignment's own round trip; the next
  // reprojectFrames(cur_frame, ...) then takes the pixel and the verdict of every candidate of these keyframes from
  // the device instead of computing them (same bits: tests/cpp/test_host_reprojector.cpp), and walks, sorts and
  // replays as before.  discardCandidateProjection() drops a queued projection (lastRunRepeated()).
  void enqueueCandidateProjection(const FramePtr& cur_frame, const std::vector<FramePtr>& kfs, const Transformation* T_iref_world,
                                  int align_result_index);
  void discardCandidateProjection();
  bool doesFrameHaveEnoughFeatures(const FramePtr& frame) const
  {
    return options_.max_n_features_per_frame > 0 && frame->numTrackedFeatures() >= options_.max_n_features_per_frame;
  }
  ~ReprojectorHip();
  ReprojectorHip(const ReprojectorHip&) = delete;
  ReprojectorHip& operator=(const ReprojectorHip&) = delete;

  // ---- reprojectFrames in the phases it is made of.  reprojectFrames() runs them one after the other with the device
  // work on its own context; a driver of MANY camera streams (FrontendLockstep) runs each phase for all its streams --
  // the host phases on a pool of threads, the device phases as ONE launch for everybody.  Same code, same order per
  // stream: a stream's features, counters and side effects are those of its own reprojectFrames call.
  // (1) candidate projection: how many points / keyframe poses the projection of `kfs` has, then the inputs written
  //     where the caller says (its own page-locked block), then the results adopted from where the device left them
  //     (they are read in place by the next walkCandidates, and must stay valid until it returns)
  //     ranges != NULL (svoh_project_candidates_stage_ranges): every keyframe must have resident columns (Frame::features); its
  //     entry of the table is written as the range of its points -- they begin at point_offset of the whole launch and belong to
  //     `job` -- and per point only kind and mu (a landmark's position too) are written: kf stays untouched
  struct ProjectionArrays { svoh_se3* T_world_kf; uint8_t* kind; int32_t* kf; double* v; double* mu; svoh_candidate_range* ranges = nullptr; int32_t point_offset = 0, job = 0;
                           // mu_unit != NULL (ranges form): a seed point whose keyframe's seeds are in a depth-filter update still in flight
                           // names the unit of that batch (unit_of, as resolveAlignmentPoints) instead of carrying its inverse depth -- the
                           // driver must finish that update before it walks the candidates (walkCandidates trusts such points)
                           int32_t* mu_unit = nullptr; };
  void countCandidateProjection(const std::vector<FramePtr>& kfs, size_t* n_points, size_t* n_kf) const;
  void gatherCandidateProjection(const FramePtr& cur_frame, const std::vector<FramePtr>& kfs, const ProjectionArrays& into,
                                 const std::function<int32_t(const Frame& keyframe, size_t seed_id)>& unit_of = nullptr);
  void adoptCandidateProjection(const FramePtr& cur_frame, const double* px, const uint8_t* visible);
  // (2) grid and statistics reset, the walk over the visible keyframes' features: the three candidate lists
  void walkCandidates(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs, std::vector<PointPtr>& trash_points);
  //     ... or without the unconverged seeds (the longest list, and in the steady state one whose pass is not reached): for drivers
  //     of the paused replay -- planPausedPass(…, &visible_kfs) walks them when their pass does come up
  void walkCandidatesWithoutUnconverged(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs, std::vector<PointPtr>& trash_points);
  // (3) what every candidate of the first n_speculated lists matches against: the stream's direct and seed batch
  void planMatches(const FramePtr& cur_frame, int n_speculated, bool resident_features = false);
  detail::SpeculativeMatches& plannedMatches() { return *sm_; }
  // (4) sortCandidatesByReprojStats of the three lists (while the device works)
  void sortCandidateLists();
  // (5) the reference's three passes over finished batches; ctx_for_unspeculated: where a pass nobody planned is
  //     matched with a round trip of its own (NULL: there must be none -- n_speculated was 3)
  void replayMatches(const FramePtr& cur_frame, svoh_ctx* ctx_for_unspeculated);
  // The same for drivers that cannot let the replay make a device call of its own (FrontendLockstep's worker threads): a pass
  // that has to run and was not planned PAUSES the replay -- true is returned, frame, grid, counters and lists stay as they are.
  // planPausedPass plans that pass' list (plannedMatches() then holds its batch alone), the driver runs the batch, resumeReplay
  // goes on from there (and may pause again).  Passes in the reference's order either way: same results.
  void sortPlannedListsOnly(bool on) { sort_unplanned_lists_ = !on; }
  bool replayMatchesUntilUnplanned(const FramePtr& cur_frame);
  void planPausedPass(const FramePtr& cur_frame, bool resident_features = false, const std::vector<FramePtr>* visible_kfs = nullptr);
  bool resumeReplay(const FramePtr& cur_frame);
  // SVOH_REPROJ_DEVICE_SELECT=1 (read when the reprojector is made): the passes' selection -- which candidates are tried, where a
  // pass ends -- comes from svoh_select_matches_batch instead of the walk over the grid in replay(); same features, counters, grid
  // and side effects (tests/cpp/test_host_reprojector.cpp runs both).  Off by default: it is a round trip of its own.
  bool device_select_ = false;
  bool reachedUnconvergedPass() const { return reached_unconverged_; }

  std::unique_ptr<OccupandyGrid2D> grid_;
  std::vector<reprojector::Candidate> candidates_;
  reprojector::Statistics stats_;
  ReprojectorOptions options_;
 private:
  svoh_ctx* ctx_;
  size_t camera_index_;
  bool speculate_unconverged_ = false;   // was the unconverged-seed pass reached on the previous frame?
  bool reached_unconverged_ = false;
  bool replayPasses(const FramePtr& cur_frame, svoh_ctx* ctx_for_unspeculated, bool pause_at_unplanned);
  void walkLists(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs, std::vector<PointPtr>& trash_points, int which);
  void emitCandidate(const FramePtr& cur_frame, const FramePtr& ref_frame, const svoh::Rigid& T_world_ref, long proj_off, size_t proj_n, size_t i,
                     std::vector<reprojector::Candidate>& list);
  bool unconverged_pending_ = false;
  // sortCandidateLists sorts every list (reprojectFrames: an unplanned pass goes through matchCandidates, which expects its list
  // sorted) or, for drivers of the paused replay, the planned ones only (FrontendLockstep sets this to false)
  bool sort_unplanned_lists_ = true;
  int lists_sorted_ = 3;
  int replay_next_pass_ = 0; bool replay_paused_ = false, replay_stop_ = false; size_t replay_max_n_ = 0;
  // queued / collected device projection: per keyframe the offset of its first feature in the flat arrays
  const Frame* proj_frame_ = nullptr;
  int proj_frame_id_ = 0;
  bool proj_collected_ = false;
  // a keyframe of the queued projection as it was when queued: identity (address and id), its slice of the flat arrays,
  // how many features the slice covers, the pose it was projected with
  struct ProjKf { const Frame* frame; int id; size_t offset; size_t n_features; Transformation T_f_w; };
  std::vector<ProjKf> proj_kf_off_;
  // the projection's inputs and results, read through these (the vectors below, or the driver's page-locked block)
  const uint8_t* proj_kind_p_ = nullptr; const double* proj_v_p_ = nullptr; const double* proj_mu_p_ = nullptr; const int32_t* proj_unit_p_ = nullptr;
  const double* proj_px_p_ = nullptr; const uint8_t* proj_visible_p_ = nullptr;
  size_t proj_n_points_ = 0, proj_n_kf_ = 0;
  std::vector<uint8_t> proj_kind_, proj_visible_;
  std::vector<int32_t> proj_kf_;
  std::vector<double> proj_v_, proj_mu_, proj_px_;
  std::vector<svoh_se3> proj_T_world_kf_;
  // the candidate lists of the converged / unconverged seeds (candidates_ holds the landmarks'), the plan of the frame
  std::vector<reprojector::Candidate> converged_, unconverged_;
  std::unique_ptr<detail::SpeculativeMatches> sm_;
  std::vector<std::vector<detail::Resolved>> plan_rs_;
  int n_speculated_ = 3;
  bool have_proj_ = false;
};

namespace reprojector_utils {
// reprojector.cpp:309-339, 489-543
void sortCandidatesByReprojStats(std::vector<reprojector::Candidate>& candidates);
bool getCandidate(const FramePtr& cur_frame, const FramePtr& ref_frame, size_t ref_index, reprojector::Candidate& candidate);
bool getCandidate(const FramePtr& cur_frame, const FramePtr& ref_frame, size_t ref_index, reprojector::Candidate& candidate,
                  const svoh::Rigid* T_world_ref);   // with ref_frame->T_world_cam() already at hand
bool projectPointAndCheckVisibility(const FramePtr& frame, const svoh::Vec3& xyz, double* px);
void setGridCellsOccupied(const std::vector<reprojector::Candidate>& candidates, OccupandyGrid2D& grid);
// Appends every matched feature to `frame` (px_vec_, f_vec_, grad_vec_, level_vec_, type_vec_, score_vec_,
// invmu_sigma2_a_b_vec_, landmark_vec_, seed_ref_vec_ at index num_features_, then ++num_features_), updates
// the seeds of the reference frames and the landmarks' reprojection counters exactly as the sequential
// reference loop does, and erases the consumed candidates from the front of the list.
void matchCandidates(svoh_ctx* ctx, const FramePtr& frame, size_t max_n_features_per_frame, bool affine_est_offset,
                     bool affine_est_gain, std::vector<reprojector::Candidate>& candidates, OccupandyGrid2D& grid,
                     reprojector::Statistics& stats, double seed_sigma2_thresh);
// Three candidate lists (landmarks, converged seeds, unconverged seeds) matched with ONE round trip to the device:
// all matcher work is queued as one direct batch plus one seed batch, then before_pass(k, max_n) / the reference's
// loop over list k / after_pass(k) run for k = 0, 1, 2 until before_pass says stop (Reprojector::reprojectFrames).
// sort_in_flight: the lists come UNSORTED and are put through sortCandidatesByReprojStats here, between sending the
// matcher work off and waiting for it (the work of a candidate does not depend on its place in its list).
void matchCandidatesFused(svoh_ctx* ctx, const FramePtr& frame, bool affine_est_offset, bool affine_est_gain, double seed_sigma2_thresh,
                          std::vector<reprojector::Candidate>* lists[3], const std::function<bool(int pass, size_t& max_n)>& before_pass,
                          const std::function<void(int pass)>& after_pass, OccupandyGrid2D& grid, reprojector::Statistics stats[3],
                          int n_speculated = 3, bool sort_in_flight = false);   // lists [n_speculated, 3) are matched only when their pass is reached
// Matcher::MatchResult per candidate of the last call on this thread (-1 = never reached / cell taken,
// 1000 = landmark without a close view), for tests and statistics.
const std::vector<int32_t>& lastMatchResults();
}  // namespace reprojector_utils

// ---------------------------------------------------------------------------
// Keyframe feature detection.  Mirrors AbstractDetector / FastDetector /
// FastGradDetector (src/svo_direct/include/svo/direct/feature_detection.h,
// src/svo_direct/src/feature_detection.cpp:27-50, 113-194) and DetectorOptions
// (feature_detection_types.h:49-84).
// ---------------------------------------------------------------------------
enum class DetectorType { kFast, kFastGrad };
struct DetectorOptions {
  size_t cell_size = 30;
  int max_level = 2;
  int min_level = 0;
  int border = 8;
  DetectorType detector_type = DetectorType::kFast;
  double threshold_primary = 10.0;
  double threshold_secondary = 100.0;
};

class DetectorHip {
 public:
  DetectorHip(svoh_ctx* ctx, const DetectorOptions& options, int image_width, int image_height);
  // AbstractDetector::detect(const FramePtr&) (feature_detection.cpp:40-50): replaces the frame's features
  // by the detected ones (px, score, level, grad, type, normalised bearing vectors, empty seed / landmark slots)
  void detect(const FramePtr& frame);
  // the virtual detect(img_pyr, mask, max_n_features, px_vec, score_vec, level_vec, grad_vec, types_vec): appends
  void detect(svoh_frame_t img_pyr, const uint8_t* mask, int mask_pitch, size_t max_n_features, std::vector<double>& px_vec,
              std::vector<double>& score_vec, std::vector<int32_t>& level_vec, std::vector<double>& grad_vec,
              std::vector<uint8_t>& types_vec);
  // detect(img_pyr, ...) in two halves around ONE batched device call for many streams' keyframes
  // (svoh_detect_cells_batch): the options and this detector's occupancy bytes, then fillFeatures on the cells that came back
  svoh_detector_options abiOptions() const;
  void occupancyBytes(uint8_t* out /* grid_.size() */) const;
  void fillFromCells(const uint64_t* corner_keys, const uint64_t* edge_keys, const float* edge_angles, int width, int height, size_t max_n_features,
                     std::vector<double>& px_vec, std::vector<double>& score_vec, std::vector<int32_t>& level_vec, std::vector<double>& grad_vec,
                     std::vector<uint8_t>& types_vec);
  void resetGrid() { grid_.reset(); }
  // OccupandyGrid2D::fillWithKeypoints (occupancy_grid_2d.h:96-107): 2 x n pixel coordinates
  void fillGridWithKeypoints(const std::vector<double>& px_vec, size_t n);
  OccupandyGrid2D grid_;   // callers mark cells of existing features: grid_.fillWithKeypoints / setOccupied
 private:
  svoh_ctx* ctx_;
  DetectorOptions options_;
};

// ---------------------------------------------------------------------------
// Stereo seam.  Mirrors svo::StereoTriangulation (src/svo/include/svo/stereo_triangulation.h:14-40,
// src/svo/src/stereo_triangulation.cpp:23-140) as FrameHandlerStereo::makeKeyframe drives it
// (frame_handler_stereo.cpp:170-205): detect new features in the left frame, match each into the right frame along
// its epipolar line (Matcher::findEpipolarMatchDirect, 500 steps) and make landmarks of the successes.
// All candidates are matched in ONE svoh_epipolar_match_batch launch; the reference's loop (shuffled order,
// stop after n_desired successes, feature / landmark bookkeeping of both frames) is replayed on the host.
// ---------------------------------------------------------------------------
struct StereoTriangulationOptions {   // stereo_triangulation.h:14-20
  size_t triangulate_n_features = 120;
  double mean_depth_inv = 1.0 / 3.0;
  double min_depth_inv = 1.0 / 1.0;
  double max_depth_inv = 1.0 / 50.0;
};

class StereoTriangulationHip {
 public:
  StereoTriangulationOptions options_;
  std::shared_ptr<DetectorHip> feature_detector_;
  StereoTriangulationHip(svoh_ctx* ctx, const StereoTriangulationOptions& options, const std::shared_ptr<DetectorHip>& feature_detector);
  void compute(const FramePtr& frame0, const FramePtr& frame1);
  // ---- compute() in the phases it is made of, for a driver of MANY rigs (FrontendLockstepStereo) that runs the detector for all of them
  // in one call and all their epipolar searches in ONE svoh_epipolar_match_batch (a pair of frames per rig).  Same code, same order per rig.
  //   wantsFeatures(frame0)   false: "sufficient number of features", compute() returns at once
  //   prepare(...)            the detector's new features (of frame0, with this object's detector grid) appended to frame0, the visiting
  //                           order shuffled, room made in frame1; job: the new features are frame0's [n_old, n_old + n_new), the pair's
  //                           views and T_frame1_frame0 for the batch.  false: nothing to match
  //   finish(...)             the reference's loop over the batch's results of this pair (result / depth / px_cur / f_cur / A of its n_new units)
  struct Job { size_t n_old = 0, n_new = 0, n_desired = 0; std::vector<size_t> indices; svoh_frame_view v0{}, v1{}; svoh_se3 T_f1_f0{}; };
  bool wantsFeatures(const Frame& frame0) const;
  bool prepare(const FramePtr& frame0, const FramePtr& frame1, const std::vector<double>& new_px, const std::vector<double>& new_scores, const std::vector<int32_t>& new_levels,
               const std::vector<double>& new_grads, const std::vector<uint8_t>& new_types, Job* job);
  void finish(const FramePtr& frame0, const FramePtr& frame1, const Job& job, const int32_t* result, const double* depth, const double* px_cur, const double* f_cur, const double* A);
  static svoh_matcher_options matcherOptions();
  // The reference shuffles the corner and the edgelet part of the new indices with std::random_shuffle (rand()):
  // the default does the same; a caller that needs a reproducible order (tests) sets its own.
  std::function<void(std::vector<size_t>& indices, size_t n_corners)> shuffle_;
  // of the last compute(): the visiting order, Matcher::MatchResult per visited index (-1 = not reached), counts
  std::vector<size_t> last_indices_;
  std::vector<int32_t> last_results_;
  size_t last_n_succeeded_ = 0, last_n_failed_ = 0;
  int next_point_id_ = 0;   // PointIdProvider::getNewPointId() of the landmarks made here
 private:
  svoh_ctx* ctx_;
};

// ---------------------------------------------------------------------------
// Pose optimiser.  Mirrors svo::PoseOptimizer (src/svo/include/svo/pose_optimizer.h:20-80,
// src/svo/src/pose_optimizer.cpp:17-113, 198-307) as FrameHandlerBase::optimizePose drives it
// (frame_handler_base.cpp:746-790): setRotationPrior, run(frame_bundle, reproj_thresh_px).
// ---------------------------------------------------------------------------
class PoseOptimizerHip {
 public:
  enum class ErrorType { kUnitPlane, kBearingVectorDiff, kImagePlane };
  struct Statistics { double reproj_error_after = 0.0, reproj_error_before = 0.0; } stats_;
  explicit PoseOptimizerHip(svoh_ctx* ctx, SolverOptions solver_options = getDefaultSolverOptions());
  static SolverOptions getDefaultSolverOptions() { SolverOptions o; o.max_iter = 10; o.eps = 0.000001; return o; }
  void setErrorType(ErrorType type) { err_type_ = type; }
  void setRotationPrior(const svoh::Quat& R_frame_world, double lambda);
  void reset() { have_prior_ = false; }   // MiniLeastSquaresSolver::reset drops the prior
  // optimises frame_bundle->at(0)->T_imu_world(), writes T_f_w_ of every frame, marks outliers
  // (type_vec_[i] = kOutlier, landmark / seed reference dropped) and returns the number of remaining measurements
  size_t run(const FrameBundle::Ptr& frame_bundle, double reproj_thresh_px);
  // the same, with `after_launch` called between the launch and the wait for its results (svoh_optimize_pose_batch_hook):
  // host work that does not need the optimised pose -- DepthFilterHip::prepareUpdateSeeds -- runs beside the kernel
  size_t run(const FrameBundle::Ptr& frame_bundle, double reproj_thresh_px, const std::function<void()>& after_launch);
  size_t iterCount() const { return static_cast<size_t>(last_.iters); }
  const svoh_pose_result& lastResult() const { return last_; }
  double measurement_sigma_ = 1.0;
  // run() in two halves (FrontendLockstep: the bundles of many streams in ONE svoh_optimize_pose_batch): prepareRun builds
  // this stream's options and problem (the arrays it points to live in this object until finishRun), finishRun does what
  // run() does with the result and returns run()'s value.
  void prepareRun(const FrameBundle::Ptr& frame_bundle, double reproj_thresh_px, svoh_pose_options& o, svoh_pose_problem& pb);
  size_t finishRun(const FrameBundle::Ptr& frame_bundle, const svoh_pose_result& result);
 private:
  svoh_ctx* ctx_;
  SolverOptions solver_options_;
  ErrorType err_type_ = ErrorType::kUnitPlane;
  bool have_prior_ = false;
  double prior_lambda_ = 0.0;
  svoh::Quat R_prior_{ 1, 0, 0, 0 };
  svoh_pose_result last_{};
  std::vector<std::vector<double>> run_xyz_;                 // per camera, between prepareRun and finishRun
  std::vector<std::vector<uint8_t>> run_usable_, run_outlier_;
};

// ---------------------------------------------------------------------------
// Structure optimisation (second half).  FrameHandlerBase::optimizeStructure
// (frame_handler_base.cpp:779-825): every landmark of the new keyframe's frames that is not an edgelet gets
// Point::optimize(max_iter) (point.cpp:248-325) over all its observations, one batched device call per frame.
// Like the reference, max_n_pts > 0 only reorders the candidates (std::nth_element on last_structure_optim_) --
// its range-for runs over the whole container, not up to the partition point -- so every candidate is optimised
// and stamped; max_n_pts == 0 returns at once.  Returns the number of landmarks handed to the optimiser.
// ---------------------------------------------------------------------------
size_t optimizeStructure(svoh_ctx* ctx, const FrameBundle::Ptr& frames, int max_n_pts, int max_iter);
// The same in phases, for callers that put the landmarks of MANY frames (one per camera stream) into one device call
// (FrontendLockstep): gather() is optimizeStructure's selection and staging for ONE frame; the caller runs
// svoh_optimize_points_batch over the batch (or over several batches laid side by side: view indices are the batch's own)
// and hands the optimised positions to apply().
struct StructureBatch {
  std::vector<PointPtr> pts;
  int stamp = 0;                                  // the frame's id: last_structure_optim_ of every point handed in
  std::vector<svoh_se3> views;
  std::vector<int32_t> obs_begin, obs_view;       // obs_begin: pts.size() + 1 entries
  std::vector<double> obs_f, pos;
  // returns max_n_pts as the reference leaves it for the next frame of a bundle (frame_handler_base.cpp:806)
  int gather(const Frame& frame, int max_n_pts);
  void apply(const double* pos_out);
  size_t size() const { return pts.size(); }
};

// FrameHandlerBase::upgradeSeedsToFeatures (frame_handler_base.cpp:828-920), what the frame handler does to a frame it has
// selected as keyframe: every feature with a landmark adds itself to the landmark's observations; every feature that hangs on
// a seed of an older keyframe -- converged or not -- turns that seed into a landmark (a new Point at the seed's position, observed
// by the keyframe and by this frame; an existing one when another frame of the bundle made it already) and both features into
// corner / edgelet / map-point features; an upgraded edgelet's direction is refreshed from this frame's image
// (getAngleAtPixelUsingHistogram, :893-901).  Two steps, so that a driver of many streams makes ONE device call for the
// directions of all of them: the host part (returns the number of upgraded features, appends the upgraded edgelets' feature
// indices to *edgelets), then refreshEdgeletDirections for any number of frames.  next_point_id: the caller's counter of
// Point ids (Point::id(), point.cpp's global counter).
size_t upgradeSeedsToFeatures(const FramePtr& frame, int* next_point_id, std::vector<size_t>* edgelets);
void refreshEdgeletDirections(svoh_ctx* ctx, const std::vector<FramePtr>& frames, const std::vector<std::vector<size_t>>& edgelets);
inline size_t upgradeSeedsToFeatures(svoh_ctx* ctx, const FramePtr& frame, int* next_point_id)
{
  std::vector<std::vector<size_t>> e(1);
  const size_t n = upgradeSeedsToFeatures(frame, next_point_id, &e[0]);
  refreshEdgeletDirections(ctx, { frame }, e);
  return n;
}
// A keyframe leaves the map: its observations leave its landmarks (Map::removeKeyframe's loop over Point::removeObservation,
// map.cpp:64-93, point.cpp:60-66)
void removeObservationsOf(const Frame& frame);

// ---------------------------------------------------------------------------
// The part of svo::Map the reprojector's caller uses (second half): keyframe container and
// the overlap query of FrameHandlerBase::projectMapInFrame (frame_handler_base.cpp:653-661).  keyframes_ is the
// reference's own container type (map.h:23), so that iteration order -- which decides the order of equally
// distant keyframes and of the reprojection passes -- is the standard library's, as in the reference.
// ---------------------------------------------------------------------------
class Map {
 public:
  using Ptr = std::shared_ptr<Map>;
  using Keyframes = std::unordered_map<int, FramePtr>;   // frame id -> frame
  Keyframes keyframes_;
  int last_added_kf_id_ = -1;
  std::vector<int> sorted_keyframe_ids_;

  void reset() { keyframes_.clear(); sorted_keyframe_ids_.clear(); last_added_kf_id_ = -1; }   // map.cpp:21-27
  void addKeyframe(const FramePtr& new_keyframe, bool temporal_map);                            // map.cpp:99-109
  void removeKeyframe(int frame_id);                                                             // map.cpp:29-57 (container part)
  // keyframes with one of their key points visible in `frame`, with the distance between the camera centres
  void getOverlapKeyframes(const FramePtr& frame, std::vector<std::pair<FramePtr, double>>* close_kfs) const;   // map.cpp:111-133
  void getClosestNKeyframesWithOverlap(const FramePtr& cur_frame, size_t num_frames, std::vector<FramePtr>* close_kfs) const;   // :135-158
  FramePtr getClosestKeyframe(const FramePtr& frame) const;                                     // map.cpp:160-177
  FramePtr getFurthestKeyframe(const svoh::Vec3& pos) const;                                    // map.cpp:184-197
  FramePtr getKeyframeById(int id) const;                                                       // map.cpp:199-205
  void getSortedKeyframes(std::vector<FramePtr>& kfs_sorted) const;                             // map.cpp:207-218
  size_t size() const { return keyframes_.size(); }
};

// The function-local `static double px_error_angle` of depth_filter_utils::updateSeed
// (depth_filter.cpp:383-384): the first camera ever passed sets it for the whole process,
// for the depth filter and the reprojector alike.
double updateSeedPxErrorAngle(const Frame& cur_frame);
// The reference keeps three thresholds in function-local statics -- removeOutliers' two (pose_optimizer.cpp:211-212: reproj_thresh / focal
// length, and its bearing-angle form) and the depth filter's one-pixel angle (depth_filter.cpp:383-384) -- so the FIRST camera that reaches
// them fixes them for the whole process, whatever cameras come later.  The mirrors do the same.  A driver of several cameras of different
// focal lengths in one process decides which camera that is by calling this before anything else runs (FrontendLockstep does, with
// LockstepOptions::cam); calls after the statics are fixed change nothing.
void fixProcessWideThresholds(const svoh_camera& cam, double reproj_thresh_px);

}  // namespace svo_hip
