// svo_hip_lockstep.cpp -- FrontendLockstep (svo_hip_lockstep.h): many camera streams, one launch per stage.
#include "svo_hip_lockstep.h"

#include <pthread.h>
#include <sched.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "svo_hip_host_internal.h"

#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#define SVOH_CPU_RELAX() _mm_pause()
#else
#define SVOH_CPU_RELAX() do { } while (0)
#endif

namespace svo_hip {

// ---- FrontendLockstep ----------------------------------------------------------------------------------------------
namespace {
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

namespace {
enum Phase { kPhPyramid = 0, kPhFinishSeeds, kPhAlignPrep, kPhAlignLaunch, kPhProjGather, kPhAlignWait, kPhWalkPlan, kPhMatchStage, kPhMatchCopy, kPhMatchSubmit, kPhSort,
             kPhMatchWait, kPhReplay, kPhPoseCall, kPhPoseApply, kPhSeedStage, kPhSeedGather, kPhSeedSubmit, kPhKeyframe, kPhSeedWait, kPhDetectWait, kPhPrefetch, kPhStructure, kPhStructGather, kPhStructCall, kPhStructWait, kPhRest };
struct PhaseClock {
  double* acc; double t;
  explicit PhaseClock(double* a) : acc(a), t(now_ms()) {}
  void lap(int k) { const double n = now_ms(); acc[k] += n - t; t = n; }
};
}  // namespace

const char* FrontendLockstep::phaseName(int k)
{
  static const char* names[] = { "pyramid", "finish seeds", "align prep", "align launch", "projection gather", "align wait", "walk + plan", "match stage", "match copy",
                                 "match submit", "sort", "match wait", "replay + pose prep", "pose call", "pose apply", "seed stage", "seed gather", "seed submit", "keyframe", "seed wait", "detect wait", "prefetch", "structure", "structure gather", "structure call", "structure wait", "rest" };
  return k >= 0 && k < (int)(sizeof names / sizeof names[0]) ? names[k] : "";
}

struct FrontendLockstep::Stream {
  LockstepStreamOptions so;   // this stream's own options
  size_t k = 0;               // frames taken so far = the number of its next frame
  // the round in progress: has an image; has an image and a frame before it (goes through the chain); has its first image (makes its first keyframe)
  bool active = false, tracking = false, starting = false;
  int slot = -1;              // position among the round's tracking streams (index of its current frame in the batches' frame lists)
  SparseImgAlignHip img_align;
  ReprojectorHip reprojector;
  PoseOptimizerHip pose_optimizer;
  DetectorHip detector;
  std::deque<FramePtr> kfs;   // the last reprojector.max_n_kfs keyframes
  FramePtr last, frame;
  FrameBundle::Ptr b_last, b_cur;
  std::vector<FramePtr> visible;
  std::vector<PointPtr> trash;
  // the round in progress
  svoh_align_options align_opt{};
  svoh_align_problem align_pb{};
  Transformation T_iref_world{ { 1, 0, 0, 0 }, { 0, 0, 0 } };
  int32_t align_key = 0;
  int align_result = -1;
  size_t proj_points = 0, proj_kf = 0, proj_point_off = 0, proj_kf_off = 0;
  int proj_job = -1;
  size_t direct_off = 0, seeds_off = 0, ref_off = 0;
  bool do_pose = false, needs_more = false;
  svoh_pose_options pose_opt{};
  svoh_pose_problem pose_pb{};
  int pose_slot = -1;
  bool want_kf = false;
  int detect_slot = -1;
  int detect_max_n = 0;
  // the depth-filter update in flight
  std::vector<FramePtr> seed_frames;
  std::vector<size_t> seed_counts;
  size_t seed_off = 0, seed_n = 0;
  // landmarks
  StructureBatch structure;
  size_t structure_off = 0, structure_view_off = 0, structure_obs_off = 0;
  std::vector<size_t> upgraded_edgelets;
  int next_point_id = 0;
  // bookkeeping
  FrameRow row;
  bool row_open = false;
  std::vector<FrameRow> done_rows;

  Stream(svoh_ctx* ctx, const LockstepOptions& o, const LockstepStreamOptions& own, const ReprojectorOptions& ropt)
      : so(own), img_align(ctx, SparseImgAlignHip::getDefaultSolverOptions(), own.params.img_align), reprojector(ctx, ropt, 0), pose_optimizer(ctx),
        detector(ctx, own.params.detector, o.cam.width, o.cam.height) {}
};

void FrontendLockstep::check(int rc, const char* what) const
{
  if (rc != SVOH_OK) throw std::runtime_error(std::string(what) + ": " + svoh_last_error_string(ctx_));
}

FrontendLockstep::FrontendLockstep(svoh_ctx* ctx, int n_streams, const LockstepOptions& options)
    : ctx_(ctx), opt_(options)
{
  requireMatchingAbi();
  if (opt_.exclusive_pool) pool_.exclusive = opt_.exclusive_pool;
  else if (opt_.shared_pool) { pool_.shared = opt_.shared_pool; pool_.seed = opt_.shared_pool_seed; }
  else pool_.own.reset(new WorkerPool(options.n_workers < 1 ? 1 : options.n_workers, options.pin_workers));
  if (!ctx_) throw std::runtime_error("FrontendLockstep: NULL svoh_ctx (no CPU fallback exists)");
  if (n_streams < 1 || n_streams > 256) throw std::runtime_error("FrontendLockstep: n_streams out of range [1, 256]");
  // (round 6: the A/B switches of round 5 -- alignment / detector ahead, pose chain, copy policy, spin limits -- are gone from the library: their
  // numbers are in HISTORY round 5 and profiles/r05_*_ab.txt, the orders that won are the code.  What is left are two OPTIONS a caller sets:
  // resident_features and speculation, which the tests use to drive the explicit-column batches and the paused replay.)
  fixProcessWideThresholds(opt_.cam, 2.0);   // (streams with cameras of their own: the engine's camera is the one the reference's statics see first)
  speculate_all_ = opt_.speculation == LockstepOptions::kSpeculateAll;
  speculate_never_ = opt_.speculation == LockstepOptions::kSpeculateNever;
  opt_.params.depth_filter.use_threaded_depthfilter = false;   // the synchronous path
  if (!opt_.per_stream.empty() && opt_.per_stream.size() != static_cast<size_t>(n_streams)) throw std::runtime_error("FrontendLockstep: per_stream must hold one entry per stream (or none)");
  if (!opt_.per_stream.empty()) opt_.params = opt_.per_stream[0].params;   // the shared part is read from here
  opt_.params.depth_filter.use_threaded_depthfilter = false;
  for (int s = 0; s < n_streams; ++s) {
    LockstepStreamOptions so;
    if (opt_.per_stream.empty()) { so.params = opt_.params; so.depth_min = opt_.depth_min; so.depth_mean = opt_.depth_mean; so.depth_max = opt_.depth_max; so.kf_every = opt_.kf_every; so.min_tracked = opt_.min_tracked; }
    else so = opt_.per_stream[static_cast<size_t>(s)];
    so.params.depth_filter.use_threaded_depthfilter = false;
    if (so.kf_every < 1) throw std::runtime_error("FrontendLockstep: kf_every must be >= 1");
    if (so.own_camera && (so.cam.width != opt_.cam.width || so.cam.height != opt_.cam.height))
      throw std::runtime_error("FrontendLockstep: a stream's own camera must have the image size of the engine's (the streams' pyramids are one call); cameras of another size run as an engine of their own");
    // what the streams of a round share in ONE device call must be the same for all of them
    const io::FrontendParams &a = opt_.params, &b = so.params;
    const bool same = a.n_pyr_levels_to_build == b.n_pyr_levels_to_build && a.grid_size == b.grid_size && a.seed_sigma2_thresh == b.seed_sigma2_thresh &&
        a.reprojector_affine_est_offset == b.reprojector_affine_est_offset && a.reprojector_affine_est_gain == b.reprojector_affine_est_gain &&
        a.depth_filter.seed_convergence_sigma2_thresh == b.depth_filter.seed_convergence_sigma2_thresh &&
        a.depth_filter.mappoint_convergence_sigma2_thresh == b.depth_filter.mappoint_convergence_sigma2_thresh &&
        a.depth_filter.scan_epi_unit_sphere == b.depth_filter.scan_epi_unit_sphere && a.depth_filter.affine_est_offset == b.depth_filter.affine_est_offset &&
        a.depth_filter.affine_est_gain == b.depth_filter.affine_est_gain && a.detector.cell_size == b.detector.cell_size && a.detector.max_level == b.detector.max_level &&
        a.detector.min_level == b.detector.min_level && a.detector.border == b.detector.border && a.detector.detector_type == b.detector.detector_type &&
        a.detector.threshold_primary == b.detector.threshold_primary && a.detector.threshold_secondary == b.detector.threshold_secondary;
    if (!same) throw std::runtime_error("FrontendLockstep: stream " + std::to_string(s) + " differs from stream 0 in an option that the streams' shared device calls take once "
                                        "(pyramid levels, grid, detector, matcher / depth-filter switches): such streams belong in engines of their own");
    ReprojectorOptions ropt;
    ropt.max_n_features_per_frame = static_cast<size_t>(so.params.max_fts);
    ropt.cell_size = static_cast<size_t>(so.params.grid_size);
    ropt.seed_sigma2_thresh = so.params.seed_sigma2_thresh;
    ropt.affine_est_offset = so.params.reprojector_affine_est_offset;
    ropt.affine_est_gain = so.params.reprojector_affine_est_gain;
    streams_.emplace_back(new Stream(ctx_, opt_, so, ropt));
    streams_.back()->reprojector.sortPlannedListsOnly(true);
  }
}

FrontendLockstep::~FrontendLockstep()
{
  try { finish(); } catch (...) {}
  if (structure_in_flight_) { std::vector<double> pos(3 * structure_in_flight_); (void)svoh_optimize_points_batch_collect(ctx_, static_cast<int>(structure_in_flight_), pos.data(), nullptr); }
  for (auto& st : streams_) {
    for (const FramePtr& f : st->kfs) for (auto& sr : f->seed_ref_vec_) sr.keyframe.reset();   // break the self references
    if (st->last) for (auto& sr : st->last->seed_ref_vec_) sr.keyframe.reset();
  }
  streams_.clear();
  drainReleases();
  // a prefetch announced for a round that never came may still be writing its slab on the upload stream: the context's
  // stream waits for it (on the device) before the slab goes back to the pool, where the next build would reuse it unordered
  (void)svoh_prefetch_fence(ctx_);
  for (svoh_frame_t h : prefetched_) if (h) (void)svoh_release_frame(ctx_, h);
}

void FrontendLockstep::drainReleases()
{
  std::vector<svoh_frame_t> r;
  std::vector<svoh_features_t> fr;
  { std::lock_guard<std::mutex> lock(release_mu_); r.swap(to_release_); fr.swap(features_to_release_); }
  for (svoh_frame_t h : r) (void)svoh_release_frame(ctx_, h);
  for (svoh_features_t h : fr) (void)svoh_features_release(ctx_, h);
}

const Transformation& FrontendLockstep::pose(int s) const
{
  const Stream& st = *streams_.at(static_cast<size_t>(s));
  if (!st.last) throw std::runtime_error("FrontendLockstep::pose: no frame yet");
  return st.last->T_f_w_;
}

size_t FrontendLockstep::keyframesAlive(int s) const { return streams_.at(static_cast<size_t>(s))->kfs.size(); }

std::vector<FrontendLockstep::FrameRow> FrontendLockstep::completedRows(int s)
{
  std::vector<FrameRow> out;
  out.swap(streams_.at(static_cast<size_t>(s))->done_rows);
  return out;
}

void FrontendLockstep::finishStructure()
{
  if (!structure_in_flight_) return;
  const double tw = now_ms();
  std::vector<double> pos(3 * structure_in_flight_);
  check(svoh_optimize_points_batch_collect(ctx_, static_cast<int>(structure_in_flight_), pos.data(), nullptr), "svoh_optimize_points_batch_collect");
  ++device_calls_;
  structure_in_flight_ = 0;
  pool_.run(static_cast<int>(structure_streams_.size()), [&](int w) {
    Stream& st = *streams_[static_cast<size_t>(structure_streams_[static_cast<size_t>(w)])];
    st.structure.apply(pos.data() + 3 * st.structure_off);
  });
  structure_streams_.clear();
  phase_ms_[kPhStructWait] += now_ms() - tw;
}

void FrontendLockstep::finish()
{
  finishStructure();
  finishSeedUpdate();
  drainReleases();
}

// The depth-filter update sent off at the end of the last round: wait for it (usually long done), put every seed's state
// and type where the reference's loop leaves them (ref_frame.invmu_sigma2_a_b_vec_.col(i), type_vec_[i]), close the rows.
void FrontendLockstep::finishSeedUpdate()
{
  if (seeds_in_flight_) {
    seeds_in_flight_ = false;
    const double tw = now_ms();
    check(svoh_matcher_collect(ctx_), "svoh_matcher_collect");
    phase_ms_[kPhSeedWait] += now_ms() - tw;   // (part of "finish seeds")
    ++device_calls_;
    const svoh_matcher_stage_t& ss = seed_stage_;
    pool_.run(numStreams(), [&](int s) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      size_t off = st.seed_off, n_success = 0;
      for (size_t k = 0; k < st.seed_frames.size(); ++k) {
        Frame& r = *st.seed_frames[k];
        const size_t n = st.seed_counts[k];
        // (a seed that became a feature while its update was in flight -- a keyframe made between the update's launch and here
        // upgraded it -- keeps what the upgrade made of it: the same rule as DepthFilterHip::finishUpdateSeedsNow)
        for (size_t i = 0; i < n; ++i) {
          if (r.type_vec_[i] >= SVOH_FT_EDGELET) continue;
          n_success += ss.success[off + i];
          std::copy(ss.state + 4 * (off + i), ss.state + 4 * (off + i + 1), r.invmu_sigma2_a_b_vec_.begin() + 4 * i);
          r.type_vec_[i] = ss.type[off + i];
        }
        off += n;
      }
      st.seed_frames.clear(); st.seed_counts.clear(); st.seed_n = 0;
      if (st.row_open) st.row.n_seed_upd = n_success;
    });
  }
  pool_.run(numStreams(), [&](int s) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    if (!st.row_open) return;
    size_t n_conv = 0;
    for (const FramePtr& f : st.kfs)
      for (size_t i = 0; i < f->num_features_; ++i)
        n_conv += f->type_vec_[i] == SVOH_FT_CORNER_SEED_CONVERGED || f->type_vec_[i] == SVOH_FT_EDGELET_SEED_CONVERGED;
    st.row.n_converged = n_conv;
    st.done_rows.push_back(st.row);
    st.row_open = false;
  });
}

// make_keyframe of the harness in three steps, so that the detector's device work can run ahead of the pose optimisation:
//   startDetection(which)   the occupancy grids of the streams' current frames (pool), the detector on every frame that still has
//                           room for seeds ("Skip seed initialization" otherwise) queued in ONE device call, nothing waited for;
//                           what it reads -- the frames' pyramids and the features the reprojection left -- is final after the
//                           replay (the pose optimiser flags outliers, it removes nothing);
//   makeKeyframes(which)    collects the detection (or runs it now, for streams nobody started it for), then initializeSeeds'
//                           second half, the self references of the new seeds, the keyframe window, the resident columns.
void FrontendLockstep::startDetection(const std::vector<int>& which)
{
  detect_ = DetectBatch();
  if (which.empty()) return;
  const size_t n_cells = streams_[0]->detector.grid_.size();
  for (int s : which) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    st.detect_max_n = st.so.params.max_n_seeds_per_frame - static_cast<int>(st.frame->num_features_);
    st.detect_slot = -1;
    if (st.detect_max_n > 0) { st.detect_slot = static_cast<int>(detect_.streams.size()); detect_.streams.push_back(s); }
  }
  detect_.started_for = which;
  const size_t n = detect_.streams.size();
  detect_.occ.resize(n * n_cells);
  pool_.run(static_cast<int>(which.size()), [&](int w) {
    Stream& st = *streams_[static_cast<size_t>(which[static_cast<size_t>(w)])];
    st.detector.resetGrid();
    st.detector.fillGridWithKeypoints(st.frame->px_vec_, st.frame->num_features_);
    if (st.detect_slot >= 0) st.detector.occupancyBytes(detect_.occ.data() + static_cast<size_t>(st.detect_slot) * n_cells);
  });
  if (n == 0) return;
  std::vector<svoh_frame_t> frames(n);
  for (size_t i = 0; i < n; ++i) frames[i] = streams_[static_cast<size_t>(detect_.streams[i])]->frame->pyramid;
  const svoh_detector_options dopt = streams_[0]->detector.abiOptions();
  check(svoh_detect_cells_batch_enqueue(ctx_, static_cast<int>(n), frames.data(), &dopt, detect_.occ.data()), "svoh_detect_cells_batch_enqueue");
  ++device_calls_;
  detect_.in_flight = true;
}

void FrontendLockstep::makeKeyframes(const std::vector<int>& which)
{
  if (which.empty()) {
    if (detect_.in_flight) { detect_.in_flight = false; (void)svoh_detect_cells_batch_collect(ctx_, nullptr, nullptr, nullptr); }   // (cannot happen: started for a subset of `which`)
    return;
  }
  // the frame handler's step at a new keyframe (frame_handler_mono.cpp:186): the seeds a new keyframe's features hang on become landmarks --
  // the host part per stream on the pool, the refreshed directions of all streams' upgraded edgelets in ONE device call.  (What the
  // detector reads -- the features' pixels -- does not change: a detection started ahead stays what it is.)
  if (opt_.landmarks) {
    pool_.run(static_cast<int>(which.size()), [&](int w) {
      Stream& st = *streams_[static_cast<size_t>(which[static_cast<size_t>(w)])];
      st.upgraded_edgelets.clear();
      upgradeSeedsToFeatures(st.frame, &st.next_point_id, &st.upgraded_edgelets);
    });
    std::vector<FramePtr> fr;
    std::vector<std::vector<size_t>> ed;
    for (int s : which) { Stream& st = *streams_[static_cast<size_t>(s)]; if (!st.upgraded_edgelets.empty()) { fr.push_back(st.frame); ed.push_back(st.upgraded_edgelets); } }
    if (!fr.empty()) { refreshEdgeletDirections(ctx_, fr, ed); ++device_calls_; }
  }
  const size_t n_cells = streams_[0]->detector.grid_.size();
  // streams whose detection was not started ahead (the tracked-features rule fired after the pose optimisation): now, blocking
  std::vector<int> late;
  for (int s : which) if (std::find(detect_.started_for.begin(), detect_.started_for.end(), s) == detect_.started_for.end()) late.push_back(s);
  const size_t n_early = detect_.streams.size();
  detect_.ckeys.resize(n_early * n_cells); detect_.ekeys.resize(n_early * n_cells); detect_.angles.resize(n_early * n_cells);
  if (detect_.in_flight) {
    const double tw = now_ms();
    detect_.in_flight = false;
    check(svoh_detect_cells_batch_collect(ctx_, detect_.ckeys.data(), detect_.ekeys.data(), detect_.angles.data()), "svoh_detect_cells_batch_collect");
    phase_ms_[kPhDetectWait] += now_ms() - tw;   // (part of "keyframe")
  }
  if (!late.empty()) {
    DetectBatch early;
    std::swap(early, detect_);
    startDetection(late);
    // (slots of the late streams follow those of the early ones)
    const size_t n_late = detect_.streams.size();
    std::vector<uint64_t> ck(n_late * n_cells), ek(n_late * n_cells);
    std::vector<float> an(n_late * n_cells);
    if (detect_.in_flight) {
      const double tw = now_ms();
      detect_.in_flight = false;
      check(svoh_detect_cells_batch_collect(ctx_, ck.data(), ek.data(), an.data()), "svoh_detect_cells_batch_collect");
      phase_ms_[kPhDetectWait] += now_ms() - tw;
    }
    for (int s : detect_.streams) { Stream& st = *streams_[static_cast<size_t>(s)]; st.detect_slot += static_cast<int>(n_early); }
    early.ckeys.insert(early.ckeys.end(), ck.begin(), ck.end()); early.ekeys.insert(early.ekeys.end(), ek.begin(), ek.end()); early.angles.insert(early.angles.end(), an.begin(), an.end());
    std::swap(early, detect_);
  }
  const std::vector<uint64_t>&ckeys = detect_.ckeys, &ekeys = detect_.ekeys;
  const std::vector<float>& angles = detect_.angles;
  pool_.run(static_cast<int>(which.size()), [&](int w) {
    Stream& st = *streams_[static_cast<size_t>(which[static_cast<size_t>(w)])];
    const FramePtr& f = st.frame;
    const size_t n_old = f->num_features_;
    if (st.detect_slot >= 0) {
      const size_t i = static_cast<size_t>(st.detect_slot);
      std::vector<double> px, score, grad;
      std::vector<int32_t> level;
      std::vector<uint8_t> type;
      st.detector.fillFromCells(ckeys.data() + i * n_cells, ekeys.data() + i * n_cells, angles.data() + i * n_cells, opt_.cam.width, opt_.cam.height,
                                static_cast<size_t>(st.detect_max_n), px, score, level, grad, type);
      depth_filter_utils::appendSeeds(f, px, score, level, grad, type, st.so.depth_min, st.so.depth_mean);
    } else {
      st.detector.resetGrid();
    }
    // bootstrap stand-in for the initialiser's landmarks: a keyframe's own new seeds are usable for the
    // alignment of the next frame at their current depth estimate (self reference)
    for (size_t i = n_old; i < f->num_features_; ++i) { f->seed_ref_vec_[i].keyframe = f; f->seed_ref_vec_[i].seed_id = static_cast<int>(i); }
    st.kfs.push_back(f);
    while (st.kfs.size() > st.reprojector.options_.max_n_kfs) {
      for (auto& sr : st.kfs.front()->seed_ref_vec_) sr.keyframe.reset();   // break the self references
      removeObservationsOf(*st.kfs.front());                                 // (Map::removeKeyframe)
      st.kfs.pop_front();
    }
  });
  detect_ = DetectBatch();
  // the new keyframes' features are final now: their constant columns go to the device once, in one call
  if (opt_.resident_features) {
    const size_t m = which.size();
    std::vector<int32_t> n(m);
    std::vector<const double*> px(m), f(m), grad(m);
    std::vector<const int32_t*> level(m);
    std::vector<svoh_features_t> handles(m, 0);
    for (size_t w = 0; w < m; ++w) {
      const Frame& fr = *streams_[static_cast<size_t>(which[w])]->frame;
      n[w] = static_cast<int32_t>(fr.num_features_);
      px[w] = fr.px_vec_.data(); f[w] = fr.f_vec_.data(); grad[w] = fr.grad_vec_.data(); level[w] = fr.level_vec_.data();
    }
    check(svoh_features_upload(ctx_, static_cast<int>(m), n.data(), px.data(), f.data(), grad.data(), level.data(), handles.data()), "svoh_features_upload");
    ++device_calls_;
    for (size_t w = 0; w < m; ++w) streams_[static_cast<size_t>(which[w])]->frame->features = handles[w];
  }
}

void FrontendLockstep::prefetch(const uint8_t* const* next_images, int pitch)
{
  if (!next_images || !prefetched_.empty()) return;
  const int S = numStreams();
  std::vector<const uint8_t*> imgs;   // the streams that will have a frame
  for (int s = 0; s < S; ++s) if (next_images[s]) imgs.push_back(next_images[s]);
  if (imgs.empty()) return;
  std::vector<svoh_frame_t> handles(imgs.size(), 0);
  check(svoh_build_pyramid_multi_prefetch(ctx_, imgs.data(), static_cast<int>(imgs.size()), opt_.cam.width, opt_.cam.height, pitch, opt_.images_mem_space,
                                          opt_.params.n_pyr_levels_to_build, SVOH_HALFSAMPLE_REFERENCE, handles.data()), "svoh_build_pyramid_multi_prefetch");
  ++device_calls_;
  prefetched_.assign(static_cast<size_t>(S), 0);
  prefetched_from_.assign(next_images, next_images + S);
  for (int s = 0, i = 0; s < S; ++s) if (next_images[s]) prefetched_[static_cast<size_t>(s)] = handles[static_cast<size_t>(i++)];
}

void FrontendLockstep::addImages(const uint8_t* const* images, int pitch, const Transformation* T_f_w_first, const uint8_t* const* next_images)
{
  const int S = numStreams();
  device_calls_ = 0;
  times_ = RoundTimes();
  const double t0 = now_ms();
  PhaseClock pc(phase_ms_);
  // SVOH_LOCKSTEP_TRACE=<file>: one line per round with what each phase took (appended; several engines share the file)
  static const char* trace_path = getenv("SVOH_LOCKSTEP_TRACE");
  double phase_at_start[kNumPhases];
  if (trace_path) std::copy(phase_ms_, phase_ms_ + kNumPhases, phase_at_start);
  struct TraceLine {
    FrontendLockstep* e; const double* start; double t0;
    ~TraceLine()
    {
      if (!trace_path) return;
      std::string line = "engine " + std::to_string(reinterpret_cast<uintptr_t>(e) & 0xffff) + " round " + std::to_string(e->round_) + " total " + std::to_string(now_ms() - t0);
      for (int k = 0; k < kNumPhases && phaseName(k)[0]; ++k) { char b[64]; snprintf(b, sizeof b, " | %s %.3f", phaseName(k), e->phase_ms_[k] - start[k]); line += b; }
      line += "\n";
      if (FILE* f = fopen(trace_path, "a")) { fputs(line.c_str(), f); fclose(f); }
    }
  } trace_line{ this, phase_at_start, t0 };
  drainReleases();
  if (!images) throw std::runtime_error("FrontendLockstep::addImages: NULL images");

  // ---- who takes part in this round: the streams that have an image.  A stream with a frame before it goes through the chain
  // ("tracking", numbered by `slot` among the round's tracking streams); a stream's first image makes its first keyframe ("starting")
  std::vector<int> trk, starting;
  int n_active = 0;
  for (int s = 0; s < S; ++s) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    st.active = images[s] != nullptr;
    st.tracking = st.active && st.last;
    st.starting = st.active && !st.last;
    st.slot = -1;
    st.do_pose = false; st.needs_more = false; st.want_kf = false; st.proj_job = -1; st.pose_slot = -1; st.align_result = -1;
    if (st.tracking) { st.slot = static_cast<int>(trk.size()); trk.push_back(s); }
    if (st.starting) { if (!T_f_w_first) throw std::runtime_error("FrontendLockstep::addImages: a stream's first frame needs its pose (T_f_w_first)"); starting.push_back(s); }
    n_active += st.active;
  }
  const int nT = static_cast<int>(trk.size());
  if (!prefetched_.empty())   // made during the round before: for exactly the images that were announced
    for (int s = 0; s < S; ++s)
      if (prefetched_from_[static_cast<size_t>(s)] != images[s]) throw std::runtime_error("FrontendLockstep::addImages: not the images that were announced as next_images");

  // ---- pyramids of the round's images: one call
  if (n_active) {
    std::vector<svoh_frame_t> handles(static_cast<size_t>(S), 0);
    if (!prefetched_.empty()) {
      handles.swap(prefetched_);
      prefetched_.clear(); prefetched_from_.clear();
      check(svoh_prefetch_fence(ctx_), "svoh_prefetch_fence");
    } else {
      std::vector<const uint8_t*> imgs;
      for (int s = 0; s < S; ++s) if (images[s]) imgs.push_back(images[s]);
      std::vector<svoh_frame_t> made(imgs.size(), 0);
      check(svoh_build_pyramid_multi(ctx_, imgs.data(), static_cast<int>(imgs.size()), opt_.cam.width, opt_.cam.height, pitch, opt_.images_mem_space,
                                     opt_.params.n_pyr_levels_to_build, SVOH_HALFSAMPLE_REFERENCE, made.data()), "svoh_build_pyramid_multi");
      ++device_calls_;
      for (int s = 0, i = 0; s < S; ++s) if (images[s]) handles[static_cast<size_t>(s)] = made[static_cast<size_t>(i++)];
    }
    for (int s = 0; s < S; ++s) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (!st.active) continue;
      FramePtr frame(new Frame, [this](Frame* f) {
        if (f->pyramid || f->features) {
          std::lock_guard<std::mutex> lock(release_mu_);
          if (f->pyramid) to_release_.push_back(f->pyramid);
          if (f->features) features_to_release_.push_back(f->features);
        }
        delete f;
      });
      frame->pyramid = handles[static_cast<size_t>(s)];
      frame->cam = st.so.own_camera ? st.so.cam : opt_.cam;
      frame->set_T_cam_imu(svoh::inverse(st.so.own_camera ? st.so.T_B_C : opt_.T_B_C));
      frame->id_ = static_cast<int>(st.k);
      st.frame = frame;
    }
  }
  pc.lap(kPhPyramid);
  finishStructure();   // the points optimised behind the previous round: their positions are read from here on (alignment points, candidates)
  // The previous round's seed update: its results are needed from here on (alignment points, candidates) -- unless the alignment
  // is queued AHEAD of the wait: the only thing it needs of the update is the new inverse depth of the seeds its points hang on,
  // and the device has that (svoh_align_camera::pos_seed_unit reads the update's batch in place).  The wait for the update and
  // the host's share of finishing it then run beside the alignment kernel.
  const bool align_ahead = align_ahead_ && seeds_in_flight_ && nT > 0 && opt_.images_mem_space >= 0;
  if (!align_ahead) finishSeedUpdate();
  pc.lap(kPhFinishSeeds);
  const double t1 = now_ms();
  times_.pyramid = t1 - t0;

  // the round's end for every stream that had an image: its row opens, its frame becomes its last one
  auto close_round = [&]() {
    for (auto& stp : streams_) {
      Stream& st = *stp;
      if (!st.active) continue;
      st.row.is_kf = st.want_kf;
      st.row.n_landmarks = 0;
      for (size_t i = 0; i < st.frame->num_features_ && i < st.frame->landmark_vec_.size(); ++i) st.row.n_landmarks += st.frame->landmark_vec_[i] != nullptr;
      st.row_open = true;
      // (the frame before this one lives on in b_last until the stream's next alignment set-up replaces the bundles -- on the
      // pool: taking a dozen frames apart here, on the group's thread, is 0.05 ms of every round)
      st.last = st.frame; st.frame.reset();
      ++st.k;
    }
    drainReleases();
  };
  for (int s : starting) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    st.frame->T_f_w_ = T_f_w_first[s];
    st.row = FrameRow(); st.row.k = st.k;
    st.want_kf = true;
  }
  if (nT == 0) {   // nobody to track (the first round of all, or a round nobody has an image for)
    if (!starting.empty()) { startDetection(starting); makeKeyframes(starting); }
    close_round();
    prefetch(next_images, pitch);
    times_.keyframe = now_ms() - t1;
    times_.total = now_ms() - t0;
    ++round_;
    return;
  }
  const Frame& any_frame = *streams_[static_cast<size_t>(trk[0])]->frame;   // (the camera is the engine's: what depends on it is the same for every stream)

  // ---- 1. sparse image alignment against the last frame (frame_handler_base.cpp:610-643), every stream's problem in the
  // geometry it would get alone; behind it the candidate projection of every stream, its pose composed on the device
  pool_.run(S, [&](int s) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    if (!st.tracking) return;
    st.frame->T_f_w_ = st.last->T_f_w_;
    if (align_ahead)
      resolveAlignmentPoints(*st.last, [&st](const Frame& kf, size_t seed_id) -> int32_t {   // where the update in flight holds that seed
        size_t off = st.seed_off;
        for (size_t k = 0; k < st.seed_frames.size(); ++k) {
          if (st.seed_frames[k].get() == &kf) return seed_id < st.seed_counts[k] ? static_cast<int32_t>(off + seed_id) : -1;
          off += st.seed_counts[k];
        }
        return -1;   // a keyframe that has left the window: its seeds are not updated any more, the host's value stands
      });
    else resolveAlignmentPoints(*st.last);
    st.b_last.reset(new FrameBundle); st.b_cur.reset(new FrameBundle);
    st.b_last->frames_.push_back(st.last); st.b_cur->frames_.push_back(st.frame);
    st.img_align.reset();
    st.visible.assign(st.kfs.begin(), st.kfs.end());
    st.T_iref_world = st.img_align.prepareRun(st.b_last, st.b_cur, st.align_opt, st.align_pb);
    st.reprojector.countCandidateProjection(st.visible, &st.proj_points, &st.proj_kf);
  });
  pc.lap(kPhAlignPrep);
  std::vector<svoh_align_result> align_results(static_cast<size_t>(nT));
  std::vector<uint8_t> align_repeated(static_cast<size_t>(S), 0);
  svoh_candidate_stage_t cs{};
  {
    // groups of equal launch geometry AND equal options, in the order their first stream appears
    auto same_options = [](const svoh_align_options& a, const svoh_align_options& b) {
      return a.max_level == b.max_level && a.min_level == b.min_level && a.patch_size == b.patch_size && a.max_iter == b.max_iter && a.eps == b.eps &&
             a.estimate_illumination_gain == b.estimate_illumination_gain && a.estimate_illumination_offset == b.estimate_illumination_offset &&
             a.use_distortion_jacobian == b.use_distortion_jacobian && a.robustification == b.robustification && a.weight_scale == b.weight_scale;
    };
    std::vector<std::pair<int32_t, std::vector<int>>> groups;
    for (int s : trk) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      check(svoh_sparse_align_geometry_key(ctx_, &st.align_opt, &st.align_pb, &st.align_key), "svoh_sparse_align_geometry_key");
      size_t g = 0;
      while (g < groups.size() && !(groups[g].first == st.align_key && same_options(streams_[static_cast<size_t>(groups[g].second[0])]->align_opt, st.align_opt))) ++g;
      if (g == groups.size()) groups.emplace_back(st.align_key, std::vector<int>());
      groups[g].second.push_back(s);
    }
    int next_result = 0;
    std::vector<svoh_align_problem> pbs;
    for (const auto& g : groups) {
      pbs.clear();
      for (int s : g.second) { Stream& st = *streams_[static_cast<size_t>(s)]; pbs.push_back(st.align_pb); st.align_result = next_result++; }
      check(svoh_sparse_align_enqueue_keyed(ctx_, &streams_[static_cast<size_t>(g.second[0])]->align_opt, static_cast<int>(pbs.size()), pbs.data(), g.first),
            "svoh_sparse_align_enqueue_keyed");
      ++device_calls_;
    }
    pc.lap(kPhAlignLaunch);
    // ... and so is the candidate projection when its points are ranges over resident columns: a seed's inverse depth comes from the
    // update's batch on the device as well (svoh_candidate_stage_t::mu_unit), the wait moves behind both launches
    const bool proj_ahead = align_ahead && opt_.resident_features;
    if (align_ahead && !proj_ahead) { finishSeedUpdate(); pc.lap(kPhFinishSeeds); }
    // the candidate projections: one staged call for all streams that have a local map
    size_t n_points = 0, n_kf = 0;
    int n_jobs = 0;
    for (int s : trk) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      st.proj_point_off = n_points; st.proj_kf_off = n_kf;
      if (st.proj_points) { st.proj_job = n_jobs++; n_points += st.proj_points; n_kf += st.proj_kf; }
    }
    if (n_jobs) {
      if (opt_.resident_features) check(svoh_project_candidates_stage_ranges(ctx_, n_jobs, static_cast<int>(n_kf), static_cast<int>(n_points), &cs), "svoh_project_candidates_stage_ranges");
      else check(svoh_project_candidates_stage(ctx_, n_jobs, static_cast<int>(n_kf), static_cast<int>(n_points), &cs), "svoh_project_candidates_stage");
      pool_.run(S, [&](int s) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        if (st.proj_job < 0) return;
        svoh_candidate_job& jb = cs.jobs[st.proj_job];
        jb = svoh_candidate_job{};
        jb.cam = st.frame->cam;
        svoh::store_rigid(st.frame->T_cam_imu(), jb.T_f_w_or_T_cam_imu);
        svoh::store_rigid(st.T_iref_world, jb.T_imu_world_ref);
        jb.align_result_index = st.align_result;
        jb.kf_begin = static_cast<int32_t>(st.proj_kf_off); jb.n_kf = static_cast<int32_t>(st.proj_kf);
        jb.point_begin = static_cast<int32_t>(st.proj_point_off); jb.n_points = static_cast<int32_t>(st.proj_points);
        const size_t o = st.proj_point_off;
        ReprojectorHip::ProjectionArrays into{ cs.T_world_kf + st.proj_kf_off, cs.kind + o, cs.kf ? cs.kf + o : nullptr, cs.v + 3 * o, cs.mu + o };
        if (cs.ranges) { into.ranges = cs.ranges + st.proj_kf_off; into.point_offset = static_cast<int32_t>(o); into.job = st.proj_job; into.mu_unit = cs.mu_unit + o; }
        if (proj_ahead)
          st.reprojector.gatherCandidateProjection(st.frame, st.visible, into, [&st](const Frame& kf, size_t seed_id) -> int32_t {
            size_t off = st.seed_off;
            for (size_t k = 0; k < st.seed_frames.size(); ++k) {
              if (st.seed_frames[k].get() == &kf) return seed_id < st.seed_counts[k] ? static_cast<int32_t>(off + seed_id) : -1;
              off += st.seed_counts[k];
            }
            return -1;
          });
        else st.reprojector.gatherCandidateProjection(st.frame, st.visible, into);
        if (cs.job) for (size_t i = 0; i < st.proj_points; ++i) cs.job[o + i] = st.proj_job;
      });
      if (proj_ahead) check(svoh_project_candidates_enqueue_staged_units(ctx_), "svoh_project_candidates_enqueue_staged_units");
      else check(svoh_project_candidates_enqueue_staged(ctx_), "svoh_project_candidates_enqueue_staged");
      ++device_calls_;
    }
    pc.lap(kPhProjGather);
    if (proj_ahead) { finishSeedUpdate(); pc.lap(kPhFinishSeeds); }   // the previous round's seed update: waited for and written back beside the two kernels
    check(svoh_sparse_align_fetch_all(ctx_, nT, align_results.data()), "svoh_sparse_align_fetch_all");
    ++device_calls_;
    if (n_jobs) check(svoh_project_candidates_wait(ctx_), "svoh_project_candidates_wait");
    // a cluster of workgroups that never completed (status 3): the blocking entry repeats that problem with one
    // workgroup, and what was projected behind the first launch used a pose that is not the result's
    for (int s : trk) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (align_results[static_cast<size_t>(st.align_result)].status != 3) continue;
      check(svoh_sparse_align_batch(ctx_, &st.align_opt, 1, &st.align_pb, &align_results[static_cast<size_t>(st.align_result)]), "svoh_sparse_align_batch");
      align_repeated[static_cast<size_t>(s)] = 1;
      ++device_calls_;
    }
  }
  pc.lap(kPhAlignWait);
  const double t2 = now_ms();
  times_.align = t2 - t1;

  // ---- 2. reprojection (frame_handler_base.cpp:645-744): walk and plan per stream, ONE direct batch and ONE seed batch.
  // A stream's third list (the unconverged seeds: up to a thousand units) is planned only if its pass was reached on the stream's
  // frame before -- the policy of ReprojectorHip::reprojectFrames.  A stream that reaches a pass nobody planned pauses its replay;
  // those streams get one more batch of their own (below), the others lose nothing.
  pool_.run(S, [&](int s) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    if (!st.tracking) return;
    st.row = FrameRow(); st.row.k = st.k;
    st.row.n_aligned = st.img_align.finishRun(align_results[static_cast<size_t>(st.align_result)], st.b_cur, st.T_iref_world);
    if (st.proj_job >= 0 && !align_repeated[static_cast<size_t>(s)] && st.row.n_aligned != 0)
      st.reprojector.adoptCandidateProjection(st.frame, cs.px + 2 * st.proj_point_off, cs.visible + st.proj_point_off);
    else st.reprojector.discardCandidateProjection();
    st.trash.clear();
    const bool third = !speculate_never_ && (speculate_all_ || st.k <= 1 || st.reprojector.reachedUnconvergedPass());   // (of the frame before: the walk resets nothing of it)
    if (third) st.reprojector.walkCandidates(st.frame, st.visible, st.trash);
    else st.reprojector.walkCandidatesWithoutUnconverged(st.frame, st.visible, st.trash);   // (their turn comes with their pass, if it comes)
    st.reprojector.planMatches(st.frame, third ? 3 : 2, opt_.resident_features);
  });
  pc.lap(kPhWalkPlan);
  // one matcher round for the streams in `who`: their planned batches staged side by side, sent off, `meanwhile` on the host, collected,
  // every stream's outputs pointed at its slices
  svoh_matcher_stage_t ds{}, ss{};
  auto matcher_round = [&](const std::vector<int>& who, const std::function<void()>& meanwhile) {
    size_t n_direct = 0, n_seeds = 0, n_refs = 0;
    for (int s : who) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      detail::SpeculativeMatches& sm = st.reprojector.plannedMatches();
      st.direct_off = n_direct; st.seeds_off = n_seeds; st.ref_off = n_refs;
      n_direct += sm.direct.size(); n_seeds += sm.seeds.size(); n_refs += sm.frames.size();
    }
    ds = svoh_matcher_stage_t{}; ss = svoh_matcher_stage_t{};
    if (n_direct + n_seeds == 0) { if (meanwhile) meanwhile(); return; }
    const svoh_matcher_options mopt = detail::reprojectorMatcherOptions(opt_.params.reprojector_affine_est_offset, opt_.params.reprojector_affine_est_gain);
    const int max_views = static_cast<int>(n_refs) + nT + 1;
    check(svoh_matcher_begin_deferred(ctx_), "svoh_matcher_begin_deferred");
    struct CloseSection { svoh_ctx* c; bool armed; ~CloseSection() { if (armed) (void)svoh_matcher_collect(c); } } close_section{ ctx_, true };
    const int stage_flags = SVOH_STAGE_MATCH_OUTPUTS | (opt_.resident_features ? SVOH_STAGE_RESIDENT_COLUMNS : 0);
    if (n_direct) check(svoh_matcher_stage(ctx_, 0, static_cast<int>(n_direct), max_views, stage_flags, &ds), "svoh_matcher_stage");
    if (n_seeds) check(svoh_matcher_stage(ctx_, 1, static_cast<int>(n_seeds), max_views, stage_flags, &ss), "svoh_matcher_stage");
    std::vector<svoh_frame_view> refs(n_refs ? n_refs : 1), curs(static_cast<size_t>(nT));
    for (int s : trk) { const Stream& st = *streams_[static_cast<size_t>(s)]; curs[static_cast<size_t>(st.slot)] = detail::viewOf(*st.frame); }
    pc.lap(kPhMatchStage);
    pool_.run(static_cast<int>(who.size()), [&](int w) {
      const int s = who[static_cast<size_t>(w)];
      Stream& st = *streams_[static_cast<size_t>(s)];
      detail::SpeculativeMatches& sm = st.reprojector.plannedMatches();
      for (size_t k = 0; k < sm.frames.size(); ++k) refs[st.ref_off + k] = detail::viewOf(*sm.frames[k]);
      auto copy_batch = [&](const detail::Batch& b, const svoh_matcher_stage_t& g, size_t o) {
        const size_t m = b.size();
        if (!m) return;
        for (size_t i = 0; i < m; ++i) { g.ref_frame_idx[o + i] = b.ref_idx[i] + static_cast<int32_t>(st.ref_off); g.cur_frame_idx[o + i] = st.slot; }
        if (b.resident) memcpy(g.feature_index + o, b.fidx.data(), 4 * m);
        else {
          memcpy(g.px + 2 * o, b.px.data(), 16 * m); memcpy(g.f + 3 * o, b.f.data(), 24 * m); memcpy(g.grad + 2 * o, b.grad.data(), 16 * m);
          memcpy(g.level + o, b.level.data(), 4 * m);
        }
        memcpy(g.type + o, b.type.data(), m);
      };
      copy_batch(sm.direct, ds, st.direct_off);
      if (const size_t m = sm.direct.size()) { memcpy(ds.depth + st.direct_off, sm.direct.depth.data(), 8 * m); memcpy(ds.px_cur + 2 * st.direct_off, sm.direct.px_cur.data(), 16 * m); }
      copy_batch(sm.seeds, ss, st.seeds_off);
      if (const size_t m = sm.seeds.size()) memcpy(ss.state + 4 * st.seeds_off, sm.seeds.state.data(), 32 * m);
    });
    pc.lap(kPhMatchCopy);
    auto batch_of = [&](const svoh_matcher_stage_t& g, size_t n) {
      svoh_feature_batch fb{};
      fb.n = static_cast<int32_t>(n);
      fb.ref_frame_idx = g.ref_frame_idx; fb.cur_frame_idx = g.cur_frame_idx; fb.n_cur_frames = nT;
      fb.px = g.px; fb.f = g.f; fb.grad = g.grad; fb.level = g.level; fb.type = g.type; fb.feature_index = g.feature_index;
      fb.mem_space = SVOH_MEM_STAGED;
      return fb;
    };
    if (n_direct) {
      const svoh_feature_batch fb = batch_of(ds, n_direct);
      check(svoh_match_direct_batch(ctx_, &mopt, static_cast<int>(n_refs), refs.data(), curs.data(), &fb, ds.depth, ds.px_cur, ds.result, ds.f_cur, ds.search_level,
                                    nullptr, ds.A_cur_ref), "svoh_match_direct_batch");
    }
    if (n_seeds) {
      const svoh_feature_batch fb = batch_of(ss, n_seeds);
      const svoh_depth_filter_options o = detail::reprojectorSeedOptions(any_frame, opt_.params.seed_sigma2_thresh);
      const svoh_seed_match_outputs outs{ ss.px_cur, ss.f_cur, ss.search_level, ss.A_cur_ref };
      check(svoh_update_seeds_batch_ex(ctx_, &mopt, &o, static_cast<int>(n_refs), refs.data(), curs.data(), &fb, ss.state, ss.success, ss.result, nullptr, &outs),
            "svoh_update_seeds_batch_ex");
    }
    check(svoh_matcher_flush(ctx_), "svoh_matcher_flush");
    ++device_calls_;
    pc.lap(kPhMatchSubmit);
    if (meanwhile) meanwhile();
    close_section.armed = false;
    check(svoh_matcher_collect(ctx_), "svoh_matcher_collect");
    ++device_calls_;
    pc.lap(kPhMatchWait);
  };
  auto point_outputs = [&](Stream& st) {
    detail::SpeculativeMatches& sm = st.reprojector.plannedMatches();
    if (sm.direct.size()) {
      const size_t o = st.direct_off;
      sm.direct.out.result = ds.result + o; sm.direct.out.search_level = ds.search_level + o; sm.direct.out.px_cur = ds.px_cur + 2 * o;
      sm.direct.out.f_cur = ds.f_cur + 3 * o; sm.direct.out.A = ds.A_cur_ref + 4 * o; sm.direct.out.type = ds.type + o; sm.direct.out.success = ds.success + o;
    }
    if (sm.seeds.size()) {
      const size_t o = st.seeds_off;
      sm.seeds.out.result = ss.result + o; sm.seeds.out.search_level = ss.search_level + o; sm.seeds.out.px_cur = ss.px_cur + 2 * o;
      sm.seeds.out.f_cur = ss.f_cur + 3 * o; sm.seeds.out.A = ss.A_cur_ref + 4 * o; sm.seeds.out.state = ss.state + 4 * o; sm.seeds.out.type = ss.type + o;
      sm.seeds.out.success = ss.success + o;
    }
  };
  auto pose_prep = [&](Stream& st) {
    st.row.n_reproj = st.frame->num_features_;
    // 3. pose optimisation (frame_handler_base.cpp:746-790): this stream's bundle
    st.do_pose = st.frame->num_features_ >= 10;
    if (st.do_pose) st.pose_optimizer.prepareRun(st.b_cur, 2.0, st.pose_opt, st.pose_pb);
  };
  // sortCandidatesByReprojStats of every stream's three lists while the device works
  matcher_round(trk, [&]() {
    pool_.run(S, [&](int s) { Stream& st = *streams_[static_cast<size_t>(s)]; if (st.tracking) st.reprojector.sortCandidateLists(); });
    pc.lap(kPhSort);
  });
  // the context's stream is idle here: the next round's images start their way up now, beside the rest of this round
  { const double tw = now_ms(); prefetch(next_images, pitch); phase_ms_[kPhPrefetch] += now_ms() - tw; }   // (part of "replay + pose prep")
  // the reference's three passes per stream on its slices of the finished batches, then the stream's pose problem
  pool_.run(S, [&](int s) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    if (!st.tracking) return;
    point_outputs(st);
    st.needs_more = st.reprojector.replayMatchesUntilUnplanned(st.frame);
    if (!st.needs_more) pose_prep(st);
  });
  // streams whose replay stands before a pass that was not planned: that pass' list in a batch of their own, and on
  for (;;) {
    std::vector<int> more;
    for (int s : trk) if (streams_[static_cast<size_t>(s)]->needs_more) more.push_back(s);
    if (more.empty()) break;
    pool_.run(static_cast<int>(more.size()), [&](int w) {
      Stream& st = *streams_[static_cast<size_t>(more[static_cast<size_t>(w)])];
      st.reprojector.planPausedPass(st.frame, opt_.resident_features, &st.visible);
    });
    matcher_round(more, nullptr);
    pool_.run(static_cast<int>(more.size()), [&](int w) {
      Stream& st = *streams_[static_cast<size_t>(more[static_cast<size_t>(w)])];
      point_outputs(st);
      st.needs_more = st.reprojector.resumeReplay(st.frame);
      if (!st.needs_more) pose_prep(st);
    });
  }
  pc.lap(kPhReplay);
  const double t3 = now_ms();
  times_.reproject = t3 - t2;
  // the streams whose periodic keyframe falls on this frame (every stream counts its own frames), and the streams that start: their
  // detector goes to the device now, ahead of the pose optimisation, and is long done when the keyframes are made (section 5) --
  // behind the depth filter's update it would be waited for
  if (detect_ahead_) {
    std::vector<int> ahead = starting;
    for (int s : trk) { const Stream& st = *streams_[static_cast<size_t>(s)]; if (st.k % st.so.kf_every == 0) ahead.push_back(s); }
    std::sort(ahead.begin(), ahead.end());
    if (!ahead.empty()) { startDetection(ahead); pc.lap(kPhKeyframe); }
  }

  // ---- 3 + 4. pose optimisation (frame_handler_base.cpp:746-790) and depth filter (frame_handler_mono.cpp:125): the bundles of all
  // streams in one launch, and BEHIND it on the device the seeds of every stream's keyframes into its new frame as ONE batch.  The
  // seed batch is staged before the pose call and sent off from its hook: every current frame takes its pose from the pose
  // kernel's result on the device (svoh_frame_view::pose_result_index_plus1), so the update runs while the host is still waiting
  // for -- and then applying -- the poses, and it is finished at the start of the next round, before anything reads the seeds again.
  {
    std::vector<svoh_pose_problem> pbs;
    std::vector<int> who;
    for (int s : trk) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (st.do_pose) { st.pose_slot = static_cast<int>(pbs.size()); pbs.push_back(st.pose_pb); who.push_back(s); }
    }
    size_t n_total = 0, n_ref_total = 0;
    for (int s : trk) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      st.seed_frames = st.visible;
      st.seed_counts.clear();
      size_t n = 0;
      for (const FramePtr& f : st.seed_frames) { st.seed_counts.push_back(f->num_features_); n += f->num_features_; }
      if (n == 0) { st.seed_frames.clear(); st.seed_counts.clear(); }
      st.seed_off = n_total; st.seed_n = n; st.ref_off = n_ref_total;
      n_total += n; n_ref_total += st.seed_frames.size();
    }
    // (the order of round 4 -- pose, wait, then the seed batch with the poses the host has applied -- is kept for comparison)
    const bool chain = pose_chain_ && !pbs.empty() && n_total > 0;
    DepthFilterHip df(ctx_, opt_.params.depth_filter);   // (options only: the batch is the driver's)
    const svoh_depth_filter_options dfo = df.abiOptions(any_frame);
    const svoh_matcher_options df_mopt = df.getMatcherOptions();
    std::vector<svoh_frame_view> refs(n_ref_total ? n_ref_total : 1), curs(static_cast<size_t>(nT));
    struct CloseSection { svoh_ctx* c; bool armed; ~CloseSection() { if (armed) (void)svoh_matcher_collect(c); } } close_section{ ctx_, false };
    auto stage_seeds = [&](bool poses_from_device) {
      check(svoh_matcher_begin_deferred(ctx_), "svoh_matcher_begin_deferred");
      close_section.armed = true;
      check(svoh_matcher_stage(ctx_, 1, static_cast<int>(n_total), static_cast<int>(n_ref_total) + nT + 1, opt_.resident_features ? SVOH_STAGE_RESIDENT_COLUMNS : 0, &seed_stage_),
            "svoh_matcher_stage");
      const svoh_matcher_stage_t& g = seed_stage_;
      pc.lap(kPhSeedStage);
      pool_.run(S, [&](int s) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        if (!st.tracking) return;
        svoh_frame_view& cv = curs[static_cast<size_t>(st.slot)];
        cv = detail::viewOf(*st.frame);
        if (poses_from_device && st.pose_slot >= 0) {   // T_f_w := T_cam_imu * (result pose_slot of the batch in flight), on the device
          svoh::store_rigid(st.frame->T_cam_imu(), cv.T_f_w);
          cv.pose_result_index_plus1 = st.pose_slot + 1;
        }
        size_t off = st.seed_off;
        for (size_t k = 0; k < st.seed_frames.size(); ++k) {
          const Frame& r = *st.seed_frames[k];
          refs[st.ref_off + k] = detail::viewOf(r);
          const size_t n = st.seed_counts[k];
          // resident columns: the batch is every feature of every keyframe, frame after frame (SVOH_BATCH_WHOLE_SETS) -- nothing names a unit
          if (g.feature_index) for (size_t i = 0; i < n; ++i) g.cur_frame_idx[off + i] = st.slot;
          else {
            for (size_t i = 0; i < n; ++i) { g.ref_frame_idx[off + i] = static_cast<int32_t>(st.ref_off + k); g.cur_frame_idx[off + i] = st.slot; }
            memcpy(g.px + 2 * off, r.px_vec_.data(), 16 * n); memcpy(g.f + 3 * off, r.f_vec_.data(), 24 * n); memcpy(g.grad + 2 * off, r.grad_vec_.data(), 16 * n);
            memcpy(g.level + off, r.level_vec_.data(), 4 * n);
          }
          memcpy(g.type + off, r.type_vec_.data(), n);
          memcpy(g.state + 4 * off, r.invmu_sigma2_a_b_vec_.data(), 32 * n);
          off += n;
        }
      });
      pc.lap(kPhSeedGather);
    };
    auto submit_seeds = [&]() {
      const svoh_matcher_stage_t& g = seed_stage_;
      svoh_feature_batch fb{};
      fb.n = static_cast<int32_t>(n_total);
      fb.ref_frame_idx = g.ref_frame_idx; fb.cur_frame_idx = g.cur_frame_idx; fb.n_cur_frames = nT;
      fb.px = g.px; fb.f = g.f; fb.grad = g.grad; fb.level = g.level; fb.type = g.type; fb.feature_index = g.feature_index;
      fb.mem_space = SVOH_MEM_STAGED;
      fb.layout = g.feature_index ? SVOH_BATCH_WHOLE_SETS : SVOH_BATCH_UNITS;
      check(svoh_update_seeds_batch(ctx_, &df_mopt, &dfo, static_cast<int>(n_ref_total), refs.data(), curs.data(), &fb, g.state, g.success, g.result, nullptr),
            "svoh_update_seeds_batch");
      check(svoh_matcher_flush(ctx_), "svoh_matcher_flush");
      ++device_calls_;
      close_section.armed = false;
      seeds_in_flight_ = true;
    };
    if (chain) stage_seeds(true);
    const double t3b = now_ms();
    if (!pbs.empty()) {
      std::vector<svoh_pose_result> res(pbs.size());
      if (chain) {
        // (an exception cannot cross the C boundary: it is carried over it)
        struct Hook { const std::function<void()>* fn; std::exception_ptr error; };
        const std::function<void()> fn = submit_seeds;
        Hook hook{ &fn, nullptr };
        const int rc = svoh_optimize_pose_batch_hook(ctx_, &streams_[static_cast<size_t>(who[0])]->pose_opt, static_cast<int>(pbs.size()), pbs.data(), res.data(),
                                                     [](void* user) {
                                                       Hook* h = static_cast<Hook*>(user);
                                                       try { (*h->fn)(); } catch (...) { h->error = std::current_exception(); }
                                                     }, &hook);
        if (hook.error) std::rethrow_exception(hook.error);
        check(rc, "svoh_optimize_pose_batch_hook");
      } else {
        check(svoh_optimize_pose_batch(ctx_, &streams_[static_cast<size_t>(who[0])]->pose_opt, static_cast<int>(pbs.size()), pbs.data(), res.data()), "svoh_optimize_pose_batch");
      }
      ++device_calls_;
      pc.lap(kPhPoseCall);
      pool_.run(static_cast<int>(who.size()), [&](int w) {
        Stream& st = *streams_[static_cast<size_t>(who[static_cast<size_t>(w)])];
        st.row.n_pose = st.pose_optimizer.finishRun(st.b_cur, res[static_cast<size_t>(w)]);
      });
      pc.lap(kPhPoseApply);
    }
    // ---- 3b. structure optimisation (frame_handler_mono.cpp:157: optimizeStructure(new_frames_, max_pts, 5)): the landmarks of every
    // stream's frame, gathered per stream on the pool, ONE svoh_optimize_points_batch for all of them (a stream's views and points are a
    // slice of the call's; Point::optimize of one point does not see another), applied per stream.
    if (opt_.landmarks) {
      pool_.run(S, [&](int s) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        if (!st.tracking) return;
        st.structure.gather(*st.frame, st.so.params.structure_optimization_max_pts);
      });
      pc.lap(kPhStructGather);
      size_t n_pts = 0, n_views = 0, n_obs = 0;
      for (int s : trk) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        st.structure_off = n_pts; st.structure_view_off = n_views; st.structure_obs_off = n_obs;
        if (st.so.params.structure_optimization_max_pts == 0) st.structure.pts.clear();   // (optimizeStructure returns at once)
        n_pts += st.structure.size(); n_views += st.structure.size() ? st.structure.views.size() : 0; n_obs += st.structure.size() ? st.structure.obs_view.size() : 0;
      }
      if (n_pts) {
        std::vector<svoh_se3> views(n_views);
        std::vector<int32_t> obs_begin(n_pts + 1), obs_view(n_obs);
        std::vector<double> obs_f(3 * n_obs), pos(3 * n_pts);
        pool_.run(S, [&](int s) {
          Stream& st = *streams_[static_cast<size_t>(s)];
          if (!st.tracking || !st.structure.size()) return;
          const StructureBatch& b = st.structure;
          std::copy(b.views.begin(), b.views.end(), views.begin() + st.structure_view_off);
          for (size_t k = 0; k < b.size(); ++k) obs_begin[st.structure_off + k] = b.obs_begin[k] + static_cast<int32_t>(st.structure_obs_off);
          for (size_t k = 0; k < b.obs_view.size(); ++k) obs_view[st.structure_obs_off + k] = b.obs_view[k] + static_cast<int32_t>(st.structure_view_off);
          std::copy(b.obs_f.begin(), b.obs_f.end(), obs_f.begin() + 3 * st.structure_obs_off);
          std::copy(b.pos.begin(), b.pos.end(), pos.begin() + 3 * st.structure_off);
        });
        obs_begin[n_pts] = static_cast<int32_t>(n_obs);
        pc.lap(kPhStructure);
        // queued (behind the depth filter's update on the context's stream) and NOT waited for: nothing before the next round's alignment
        // set-up reads a point's position (the keyframe step makes new points and adds observations; it moves none), so the kernel runs
        // while the host is in the keyframe phase, and its results are taken at the next round's start (finishStructure).  (A stream of
        // its own was measured and is worse: with four groups' streams on the runtime's four hardware queues a side stream waits behind
        // another group's chain -- 0.13 - 0.30 ms at the next round's start.)
        check(svoh_optimize_points_batch_enqueue(ctx_, 5, 0, static_cast<int>(n_views), views.data(), static_cast<int>(n_pts), obs_begin.data(), obs_view.data(), obs_f.data(), pos.data()),
              "svoh_optimize_points_batch_enqueue");
        ++device_calls_;
        structure_in_flight_ = n_pts;
        for (int s : trk) { Stream& st = *streams_[static_cast<size_t>(s)]; if (st.structure.size()) { structure_streams_.push_back(s); st.row.n_struct = st.structure.size(); } }
        pc.lap(kPhStructCall);
      }
      pc.lap(kPhStructure);
    }
    const double t4 = now_ms();
    times_.pose = t4 - t3b;
    if (!chain && n_total) { stage_seeds(false); submit_seeds(); }
    pc.lap(kPhSeedSubmit);
    times_.seeds = (t3b - t3) + (now_ms() - t4);
  }
  const double t5 = now_ms();

  // ---- 5. keyframe rule (every stream by its own count and its own thresholds), the new keyframes' detector in one call
  {
    std::vector<int> which = starting;
    for (int s : trk) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      st.want_kf = st.k % st.so.kf_every == 0 || st.frame->numTrackedFeatures() < st.so.min_tracked;
      if (st.want_kf) which.push_back(s);
    }
    std::sort(which.begin(), which.end());
    makeKeyframes(which);
  }
  close_round();
  pc.lap(kPhKeyframe);
  const double t6 = now_ms();
  times_.keyframe = t6 - t5;
  times_.total = t6 - t0;
  ++round_;
}

}  // namespace svo_hip

// ---- C face (svo_hip_lockstep_c.h) ----------------------------------------------------------------------------------
#include "svo_hip_lockstep_c.h"

struct svohl_engine {
  std::unique_ptr<svo_hip::FrontendLockstep> fe;
  std::vector<std::vector<svo_hip::FrontendLockstep::FrameRow>> backlog;   // rows taken from the engine and not handed out yet
};

namespace {
thread_local std::string g_svohl_error = "no error";
template <class F>
int svohl_guard(F&& f)
{
  try { f(); return SVOH_OK; }
  catch (const std::bad_alloc&) { g_svohl_error = "out of host memory"; return SVOH_ERR_OUT_OF_MEMORY; }
  catch (const std::exception& e) { g_svohl_error = e.what(); return SVOH_ERR_INVALID_ARGUMENT; }
  catch (...) { g_svohl_error = "unknown exception"; return SVOH_ERR_INVALID_ARGUMENT; }
}
}  // namespace

struct svohl_pool { std::shared_ptr<svo_hip::SharedPool> pool; std::shared_ptr<svo_hip::ExclusivePool> exclusive; };

extern "C" {

const char* svohl_last_error(void) { return g_svohl_error.c_str(); }

int svohl_pool_create(int n_workers, svohl_pool** out)
{
  return svohl_guard([&] {
    if (!out || n_workers < 1 || n_workers > 1024) throw std::runtime_error("svohl_pool_create: bad arguments");
    std::unique_ptr<svohl_pool> p(new svohl_pool);
    p->pool.reset(new svo_hip::SharedPool(n_workers));
    *out = p.release();
  });
}

int svohl_pool_create_exclusive(int n_threads, svohl_pool** out)
{
  return svohl_guard([&] {
    if (!out || n_threads < 1 || n_threads > 1024) throw std::runtime_error("svohl_pool_create_exclusive: bad arguments");
    std::unique_ptr<svohl_pool> p(new svohl_pool);
    p->exclusive.reset(new svo_hip::ExclusivePool(n_threads));
    *out = p.release();
  });
}

void svohl_pool_destroy(svohl_pool* p) { try { delete p; } catch (...) {} }

static int svohl_create_impl(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* params_yaml, double depth_min, double depth_mean,
                             double depth_max, int kf_every, int n_workers, svohl_pool* pool, int seed, int images_pinned, svohl_engine** out);

int svohl_create(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* params_yaml, double depth_min, double depth_mean,
                 double depth_max, int kf_every, int n_workers, int images_pinned, svohl_engine** out)
{
  return svohl_create_impl(ctx, n_streams, cam, T_B_C, params_yaml, depth_min, depth_mean, depth_max, kf_every, n_workers, nullptr, 0, images_pinned, out);
}

int svohl_create_shared(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* params_yaml, double depth_min, double depth_mean,
                        double depth_max, int kf_every, svohl_pool* pool, int seed, int images_pinned, svohl_engine** out)
{
  if (!pool) { g_svohl_error = "svohl_create_shared: NULL pool"; return SVOH_ERR_INVALID_ARGUMENT; }
  return svohl_create_impl(ctx, n_streams, cam, T_B_C, params_yaml, depth_min, depth_mean, depth_max, kf_every, 1, pool, seed, images_pinned, out);
}

static int svohl_create_impl(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* params_yaml, double depth_min, double depth_mean,
                             double depth_max, int kf_every, int n_workers, svohl_pool* pool, int seed, int images_pinned, svohl_engine** out)
{
  return svohl_guard([&] {
    if (!out || !cam || !T_B_C) throw std::runtime_error("svohl_create: NULL argument");
    *out = nullptr;
    svo_hip::LockstepOptions lo;
    if (pool) { lo.shared_pool = pool->pool; lo.shared_pool_seed = seed; lo.exclusive_pool = pool->exclusive; }
    lo.params = svo_hip::io::frontendParamsFromYaml(params_yaml ? svo_hip::io::parseYaml(params_yaml) : svo_hip::io::YamlNode());
    lo.cam = *cam;
    lo.T_B_C = svoh::load_rigid(*T_B_C);
    lo.depth_min = static_cast<float>(depth_min); lo.depth_mean = static_cast<float>(depth_mean); lo.depth_max = static_cast<float>(depth_max);
    lo.kf_every = kf_every > 0 ? static_cast<size_t>(kf_every) : 8;
    lo.n_workers = n_workers;
    lo.images_mem_space = images_pinned ? SVOH_MEM_HOST_PINNED : SVOH_MEM_HOST;
    std::unique_ptr<svohl_engine> e(new svohl_engine);
    e->fe.reset(new svo_hip::FrontendLockstep(ctx, n_streams, lo));
    e->backlog.resize(static_cast<size_t>(n_streams));
    *out = e.release();
  });
}

static int svohl_create_streams_impl(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, bool cameras_per_stream, const char* const* params_yaml,
                                     const double* depth_min_mean_max, const int* kf_every, const int* min_tracked, int n_workers, svohl_pool* pool, int seed,
                                     int images_pinned, svohl_engine** out);
int svohl_create_streams(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, const char* const* params_yaml,
                         const double* depth_min_mean_max, const int* kf_every, const int* min_tracked, int n_workers, svohl_pool* pool, int seed,
                         int images_pinned, svohl_engine** out)
{
  return svohl_create_streams_impl(ctx, n_streams, cam, T_B_C, false, params_yaml, depth_min_mean_max, kf_every, min_tracked, n_workers, pool, seed, images_pinned, out);
}
int svohl_create_streams_cameras(svoh_ctx* ctx, int n_streams, const svoh_camera* cams, const svoh_se3* T_B_Cs, const char* const* params_yaml,
                                 const double* depth_min_mean_max, const int* kf_every, const int* min_tracked, int n_workers, svohl_pool* pool, int seed,
                                 int images_pinned, svohl_engine** out)
{
  return svohl_create_streams_impl(ctx, n_streams, cams, T_B_Cs, true, params_yaml, depth_min_mean_max, kf_every, min_tracked, n_workers, pool, seed, images_pinned, out);
}
static int svohl_create_streams_impl(svoh_ctx* ctx, int n_streams, const svoh_camera* cam, const svoh_se3* T_B_C, bool cameras_per_stream, const char* const* params_yaml,
                                     const double* depth_min_mean_max, const int* kf_every, const int* min_tracked, int n_workers, svohl_pool* pool, int seed,
                                     int images_pinned, svohl_engine** out)
{
  return svohl_guard([&] {
    if (!out || !cam || !T_B_C || !params_yaml || !depth_min_mean_max || !kf_every || !min_tracked || n_streams < 1) throw std::runtime_error("svohl_create_streams: NULL argument");
    *out = nullptr;
    svo_hip::LockstepOptions lo;
    if (pool) { lo.shared_pool = pool->pool; lo.shared_pool_seed = seed; lo.exclusive_pool = pool->exclusive; }
    lo.cam = *cam;
    lo.T_B_C = svoh::load_rigid(*T_B_C);
    lo.n_workers = n_workers;
    lo.images_mem_space = images_pinned ? SVOH_MEM_HOST_PINNED : SVOH_MEM_HOST;
    for (int s = 0; s < n_streams; ++s) {
      svo_hip::LockstepStreamOptions so;
      so.params = svo_hip::io::frontendParamsFromYaml(params_yaml[s] ? svo_hip::io::parseYaml(params_yaml[s]) : svo_hip::io::YamlNode());
      so.depth_min = static_cast<float>(depth_min_mean_max[3 * s]); so.depth_mean = static_cast<float>(depth_min_mean_max[3 * s + 1]); so.depth_max = static_cast<float>(depth_min_mean_max[3 * s + 2]);
      so.kf_every = kf_every[s] > 0 ? static_cast<size_t>(kf_every[s]) : 8;
      so.min_tracked = min_tracked[s] >= 0 ? static_cast<size_t>(min_tracked[s]) : 60;
      if (cameras_per_stream) { so.own_camera = true; so.cam = cam[s]; so.T_B_C = svoh::load_rigid(T_B_C[s]); }
      lo.per_stream.push_back(so);
    }
    std::unique_ptr<svohl_engine> e(new svohl_engine);
    e->fe.reset(new svo_hip::FrontendLockstep(ctx, n_streams, lo));
    e->backlog.resize(static_cast<size_t>(n_streams));
    *out = e.release();
  });
}

void svohl_destroy(svohl_engine* e) { try { delete e; } catch (...) {} }

int svohl_add_images(svohl_engine* e, const uint8_t* const* images, int pitch, const svoh_se3* T_f_w_first)
{
  return svohl_guard([&] {
    if (!e || !images) throw std::runtime_error("svohl_add_images: NULL argument");
    std::vector<svo_hip::Transformation> T;
    if (T_f_w_first) for (int s = 0; s < e->fe->numStreams(); ++s) T.push_back(svoh::load_rigid(T_f_w_first[s]));
    e->fe->addImages(images, pitch, T.empty() ? nullptr : T.data());
  });
}

int svohl_run_sequence(svohl_engine* e, const uint8_t* base, size_t image_bytes, size_t stream_stride, int n_frames, int pitch, long k_first, int n_rounds,
                       const svoh_se3* T_f_w_first, double* round_ms)
{
  return svohl_guard([&] {
    if (!e || !base || n_frames < 2 || n_rounds < 0 || k_first < 0) throw std::runtime_error("svohl_run_sequence: bad arguments");
    const int S = e->fe->numStreams();
    std::vector<svo_hip::Transformation> T;
    if (T_f_w_first) for (int s = 0; s < S; ++s) T.push_back(svoh::load_rigid(T_f_w_first[s]));
    std::vector<const uint8_t*> ptrs(static_cast<size_t>(S)), next(static_cast<size_t>(S));
    const long period = 2L * (n_frames - 1);
    const bool prefetch = true;   // (the next images of a replay are known: they go up during the round)
    for (long k = k_first; k < k_first + n_rounds; ++k) {
      const long m = k % period;
      const long f = m < n_frames ? m : period - m;
      const long m1 = (k + 1) % period;
      const long f1 = m1 < n_frames ? m1 : period - m1;
      for (int s = 0; s < S; ++s) {
        ptrs[static_cast<size_t>(s)] = base + static_cast<size_t>(s) * stream_stride + static_cast<size_t>(f) * image_bytes;
        next[static_cast<size_t>(s)] = base + static_cast<size_t>(s) * stream_stride + static_cast<size_t>(f1) * image_bytes;
      }
      // (the images of the round after this one are there already -- a replay: they go up during this round)
      e->fe->addImages(ptrs.data(), pitch, T.empty() ? nullptr : T.data(), prefetch ? next.data() : nullptr);
      if (round_ms) {
        const svo_hip::FrontendLockstep::RoundTimes& t = e->fe->lastRoundTimes();
        double* o = round_ms + 7 * (k - k_first);
        o[0] = t.pyramid; o[1] = t.align; o[2] = t.reproject; o[3] = t.pose; o[4] = t.seeds; o[5] = t.keyframe; o[6] = t.total;
      }
    }
  });
}

int svohl_run_schedule(svohl_engine* e, const uint8_t* base, size_t image_bytes, size_t stream_stride, int n_frames, int pitch, long k_first, int n_rounds,
                       const int* start, const int* step, const int* every, const int* phase, const svoh_se3* T_f_w_first, double* round_ms, long* frames_done)
{
  return svohl_guard([&] {
    if (!e || !base || n_frames < 2 || n_rounds < 0 || k_first < 0 || !start || !step || !every || !phase) throw std::runtime_error("svohl_run_schedule: bad arguments");
    const int S = e->fe->numStreams();
    for (int s = 0; s < S; ++s) if (every[s] < 1 || phase[s] < 0 || step[s] == 0) throw std::runtime_error("svohl_run_schedule: every >= 1, phase >= 0, step != 0");
    std::vector<svo_hip::Transformation> T;
    if (T_f_w_first) for (int s = 0; s < S; ++s) T.push_back(svoh::load_rigid(T_f_w_first[s]));
    std::vector<const uint8_t*> ptrs(static_cast<size_t>(S)), next(static_cast<size_t>(S));
    const long period = 2L * (n_frames - 1);
    const bool prefetch = true;   // (the next images of a replay are known: they go up during the round)
    auto image_of = [&](int s, long k) -> const uint8_t* {   // stream s' image in round k, or none
      if (k < phase[s] || (k - phase[s]) % every[s] != 0) return nullptr;
      const long j = (k - phase[s]) / every[s];
      long m = (static_cast<long>(start[s]) + static_cast<long>(step[s]) * j) % period;
      if (m < 0) m += period;
      const long f = m < n_frames ? m : period - m;
      return base + static_cast<size_t>(s) * stream_stride + static_cast<size_t>(f) * image_bytes;
    };
    long done = 0;
    for (long k = k_first; k < k_first + n_rounds; ++k) {
      for (int s = 0; s < S; ++s) { ptrs[static_cast<size_t>(s)] = image_of(s, k); next[static_cast<size_t>(s)] = image_of(s, k + 1); done += ptrs[static_cast<size_t>(s)] != nullptr; }
      e->fe->addImages(ptrs.data(), pitch, T.empty() ? nullptr : T.data(), prefetch ? next.data() : nullptr);
      if (round_ms) {
        const svo_hip::FrontendLockstep::RoundTimes& t = e->fe->lastRoundTimes();
        double* o = round_ms + 7 * (k - k_first);
        o[0] = t.pyramid; o[1] = t.align; o[2] = t.reproject; o[3] = t.pose; o[4] = t.seeds; o[5] = t.keyframe; o[6] = t.total;
      }
    }
    if (frames_done) *frames_done = done;
  });
}

int svohl_pose(svohl_engine* e, int stream, svoh_se3* T_f_w)
{
  return svohl_guard([&] {
    if (!e || !T_f_w) throw std::runtime_error("svohl_pose: NULL argument");
    svoh::store_rigid(e->fe->pose(stream), *T_f_w);
  });
}

int svohl_last_round(svohl_engine* e, double times_ms[7], int* device_calls)
{
  return svohl_guard([&] {
    if (!e) throw std::runtime_error("svohl_last_round: NULL argument");
    const svo_hip::FrontendLockstep::RoundTimes& t = e->fe->lastRoundTimes();
    if (times_ms) { times_ms[0] = t.pyramid; times_ms[1] = t.align; times_ms[2] = t.reproject; times_ms[3] = t.pose; times_ms[4] = t.seeds; times_ms[5] = t.keyframe; times_ms[6] = t.total; }
    if (device_calls) *device_calls = e->fe->lastRoundDeviceCalls();
  });
}

int svohl_completed_rows(svohl_engine* e, int stream, int max_rows, int64_t* rows, int* n_rows)
{
  return svohl_guard([&] {
    if (!e || !n_rows || (max_rows > 0 && !rows)) throw std::runtime_error("svohl_completed_rows: NULL argument");
    std::vector<svo_hip::FrontendLockstep::FrameRow>& b = e->backlog.at(static_cast<size_t>(stream));
    for (const auto& r : e->fe->completedRows(stream)) b.push_back(r);
    int n = 0;
    while (n < max_rows && static_cast<size_t>(n) < b.size()) {
      const auto& r = b[static_cast<size_t>(n)];
      int64_t* o = rows + 7 * n;
      o[0] = static_cast<int64_t>(r.k); o[1] = r.is_kf; o[2] = static_cast<int64_t>(r.n_aligned); o[3] = static_cast<int64_t>(r.n_reproj);
      o[4] = static_cast<int64_t>(r.n_pose); o[5] = static_cast<int64_t>(r.n_seed_upd); o[6] = static_cast<int64_t>(r.n_converged);
      ++n;
    }
    b.erase(b.begin(), b.begin() + n);
    *n_rows = n;
  });
}

int svohl_phase_times(svohl_engine* e, int max_phases, double* ms, int* n_phases)
{
  return svohl_guard([&] {
    if (!e || !n_phases || (max_phases > 0 && !ms)) throw std::runtime_error("svohl_phase_times: NULL argument");
    int n = 0;
    while (n < svo_hip::FrontendLockstep::kNumPhases && svo_hip::FrontendLockstep::phaseName(n)[0]) ++n;
    for (int k = 0; k < n && k < max_phases; ++k) ms[k] = e->fe->phaseTimes()[k];
    *n_phases = n;
  });
}

const char* svohl_phase_name(int k) { return svo_hip::FrontendLockstep::phaseName(k); }

int svohl_finish(svohl_engine* e)
{
  return svohl_guard([&] {
    if (!e) throw std::runtime_error("svohl_finish: NULL argument");
    e->fe->finish();
  });
}

}  // extern "C"
