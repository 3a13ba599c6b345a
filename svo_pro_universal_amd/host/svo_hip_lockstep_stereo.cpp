// svo_hip_lockstep_stereo.cpp -- FrontendLockstepStereo (svo_hip_lockstep_stereo.h): many stereo streams, one launch per per-pair stage.
#include "svo_hip_lockstep_stereo.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>

#include "svo_hip_host_internal.h"

namespace svo_hip {

struct FrontendLockstepStereo::Stream {
  SparseImgAlignHip img_align;
  ReprojectorHip rp0, rp1;
  ReprojectorHip* rp[2];
  PoseOptimizerHip pose_optimizer;
  DetectorHip seed_detector;
  std::shared_ptr<DetectorHip> tri_detector;
  StereoTriangulationHip stereo;
  unsigned shuffle_state = 12345u;   // (svoh_mini_stereo's reproducible order instead of rand())
  std::deque<FramePtr> kfs;
  FrameBundle::Ptr last, bundle;
  std::vector<FramePtr> visible;
  std::vector<PointPtr> trash;
  size_t k = 0;                      // pairs taken so far
  bool active = false, tracking = false, starting = false;
  int slot = -1;
  // the round in progress
  svoh_align_options align_opt{};
  svoh_align_problem align_pb{};
  Transformation T_iref_world{ { 1, 0, 0, 0 }, { 0, 0, 0 } };
  int32_t align_key = 0;
  int align_result = -1;
  size_t direct_off = 0, seeds_off = 0, ref_off = 0;
  bool needs_more = false;
  size_t n_reproj = 0;
  bool do_pose = false;
  svoh_pose_options pose_opt{};
  svoh_pose_problem pose_pb{};
  int pose_slot = -1;
  size_t n_pose = 0;
  StructureBatch structure;
  size_t structure_off = 0, structure_view_off = 0, structure_obs_off = 0;
  int structure_max_pts = 0;
  int next_point_id = 1 << 20;       // (the triangulation's points count from 0)
  // the seed update being built / in flight
  std::vector<FramePtr> seed_frames;
  std::vector<size_t> seed_counts;
  size_t seed_off = 0;
  PairRow row;
  bool row_open = false;
  std::vector<PairRow> done_rows;

  static ReprojectorOptions reprojector_options(const io::FrontendParams& p)
  {
    ReprojectorOptions ropt;
    ropt.max_n_features_per_frame = static_cast<size_t>(p.max_fts);
    ropt.cell_size = static_cast<size_t>(p.grid_size);
    ropt.seed_sigma2_thresh = p.seed_sigma2_thresh;
    ropt.affine_est_offset = p.reprojector_affine_est_offset;
    ropt.affine_est_gain = true;   // the stereo-imu configuration estimates the gain in the matcher as well
    return ropt;
  }
  static StereoTriangulationOptions triangulation_options()
  {
    StereoTriangulationOptions sto;
    sto.triangulate_n_features = 120;   // svo_factory.cpp:240
    return sto;
  }
  Stream(svoh_ctx* ctx, const StereoLockstepOptions& o)
      : img_align(ctx, SparseImgAlignHip::getDefaultSolverOptions(), o.params.img_align), rp0(ctx, reprojector_options(o.params), 0), rp1(ctx, reprojector_options(o.params), 1),
        pose_optimizer(ctx), seed_detector(ctx, o.params.detector, o.rig[0].cam.width, o.rig[0].cam.height),
        tri_detector(new DetectorHip(ctx, o.params.detector, o.rig[0].cam.width, o.rig[0].cam.height)), stereo(ctx, triangulation_options(), tri_detector)
  {
    rp[0] = &rp0; rp[1] = &rp1;
    rp0.sortPlannedListsOnly(true); rp1.sortPlannedListsOnly(true);
    stereo.shuffle_ = [this](std::vector<size_t>& idx, size_t n_corners) {
      auto rnd = [&]() { shuffle_state = shuffle_state * 1664525u + 1013904223u; return shuffle_state >> 8; };
      auto shuf = [&](size_t a, size_t b) { for (size_t i = b; i > a + 1; --i) std::swap(idx[i - 1], idx[a + rnd() % (i - a)]); };
      shuf(0, std::min(n_corners, idx.size())); shuf(std::min(n_corners, idx.size()), idx.size());
    };
  }
};

void FrontendLockstepStereo::check(int rc, const char* what) const
{
  if (rc != SVOH_OK) throw std::runtime_error(std::string(what) + ": " + svoh_last_error_string(ctx_));
}

FrontendLockstepStereo::FrontendLockstepStereo(svoh_ctx* ctx, int n_streams, const StereoLockstepOptions& options) : ctx_(ctx), opt_(options)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("FrontendLockstepStereo: NULL svoh_ctx (no CPU fallback exists)");
  if (n_streams < 1 || n_streams > 128) throw std::runtime_error("FrontendLockstepStereo: n_streams out of range [1, 128]");
  if (opt_.rig.size() != 2) throw std::runtime_error("FrontendLockstepStereo: a rig of two cameras is needed");
  if (opt_.rig[0].cam.width != opt_.rig[1].cam.width || opt_.rig[0].cam.height != opt_.rig[1].cam.height)
    throw std::runtime_error("FrontendLockstepStereo: the two cameras must have one image size (their pyramids are built in one call)");
  if (opt_.kf_every < 1) throw std::runtime_error("FrontendLockstepStereo: kf_every must be >= 1");
  if (!opt_.per_stream_rig.empty()) {
    if (opt_.per_stream_rig.size() != static_cast<size_t>(n_streams)) throw std::runtime_error("FrontendLockstepStereo: per_stream_rig must hold one rig per stream (or none)");
    for (const auto& r : opt_.per_stream_rig) {
      if (r.size() != 2) throw std::runtime_error("FrontendLockstepStereo: a stream's rig needs two cameras");
      for (const io::RigCamera& c : r)
        if (c.cam.width != opt_.rig[0].cam.width || c.cam.height != opt_.rig[0].cam.height)
          throw std::runtime_error("FrontendLockstepStereo: a stream's own cameras must have the image size of the engine's rig (the streams' pyramids are one call)");
    }
  }
  fixProcessWideThresholds(opt_.rig[0].cam, 2.0);   // (rigs of different focal lengths: the engine's first camera is the one the reference's statics see first)
  opt_.params.depth_filter.use_threaded_depthfilter = false;
  // euroc_stereo_imu.yaml:30-31: img_align_est_illumination_gain / _offset (as svoh_mini_stereo)
  opt_.params.img_align.estimate_illumination_gain = true;
  opt_.params.img_align.estimate_illumination_offset = true;
  pool_.reset(new WorkerPool(opt_.n_workers < 1 ? 1 : opt_.n_workers, false));
  for (int s = 0; s < n_streams; ++s) streams_.emplace_back(new Stream(ctx_, opt_));
}

FrontendLockstepStereo::~FrontendLockstepStereo()
{
  try { finish(); } catch (...) {}
  if (seeds_in_flight_) (void)svoh_matcher_collect(ctx_);
  // a prefetch announced for a round that never came may still be writing its slab on the upload stream
  (void)svoh_prefetch_fence(ctx_);
  for (svoh_frame_t h : prefetched_) if (h) (void)svoh_release_frame(ctx_, h);
  for (auto& st : streams_) {
    for (const FramePtr& f : st->kfs) for (auto& sr : f->seed_ref_vec_) sr.keyframe.reset();
    if (st->last) for (const FramePtr& f : st->last->frames_) for (auto& sr : f->seed_ref_vec_) sr.keyframe.reset();
  }
  streams_.clear();
  drainReleases();
}

void FrontendLockstepStereo::drainReleases()
{
  std::vector<svoh_frame_t> r;
  std::vector<svoh_features_t> fr;
  { std::lock_guard<std::mutex> lock(release_mu_); r.swap(to_release_); fr.swap(features_to_release_); }
  for (svoh_frame_t h : r) (void)svoh_release_frame(ctx_, h);
  for (svoh_features_t h : fr) (void)svoh_features_release(ctx_, h);
}

Transformation FrontendLockstepStereo::pose(int s) const
{
  const Stream& st = *streams_.at(static_cast<size_t>(s));
  if (!st.last) throw std::runtime_error("FrontendLockstepStereo::pose: no pair yet");
  return st.last->at(0)->T_imu_world();
}

size_t FrontendLockstepStereo::keyframesAlive(int s) const { return streams_.at(static_cast<size_t>(s))->kfs.size(); }

std::vector<FrontendLockstepStereo::PairRow> FrontendLockstepStereo::completedRows(int s)
{
  std::vector<PairRow> out;
  out.swap(streams_.at(static_cast<size_t>(s))->done_rows);
  return out;
}

void FrontendLockstepStereo::finish()
{
  finishSecondSeedUpdate();
  drainReleases();
}

namespace {
size_t num_landmarks(const Frame& f)
{
  size_t n = 0;
  for (size_t i = 0; i < f.num_features_ && i < f.landmark_vec_.size(); ++i) n += f.landmark_vec_[i] != nullptr;
  return n;
}
bool scene_depth(const Frame& f, double& d_med, double& d_min)   // frame_utils::getSceneDepth on the landmarks (as svoh_mini_stereo)
{
  std::vector<double> d;
  for (size_t i = 0; i < f.num_features_ && i < f.landmark_vec_.size(); ++i)
    if (f.landmark_vec_[i]) { const svoh::Vec3 p = svoh::transform(f.T_f_w_, f.landmark_vec_[i]->pos()); d.push_back(sqrt(p.x * p.x + p.y * p.y + p.z * p.z)); }
  if (d.empty()) return false;
  std::sort(d.begin(), d.end());
  d_med = d[d.size() / 2]; d_min = d.front();
  return true;
}
}  // namespace

// svoh_mini_stereo's make_keyframe for the streams `which` (stream, kf_id: which of the pair's frames is the keyframe), every device step
// ONE call for all of them -- the two detector runs, the epipolar searches of the stereo triangulation, the refreshed edgelet directions,
// the upload of the resident columns -- and the host steps between them per stream on the pool.  Per stream the steps and their order are
// those of the single-stream harness (frame_handler_stereo.cpp:146-175): upgradeSeedsToFeatures of the frame that is not the keyframe,
// StereoTriangulation::compute (detector in the left frame's free cells, epipolar match of every new feature, the loop up to n_desired),
// upgradeSeedsToFeatures of the keyframe, depth_filter_->addKeyframe (new seeds in its free cells), the keyframe window.
void FrontendLockstepStereo::makeKeyframes(const std::vector<std::pair<int, size_t>>& which)
{
  const size_t K = which.size();
  if (K == 0) return;
  auto stream_of = [&](size_t w) -> Stream& { return *streams_[static_cast<size_t>(which[w].first)]; };
  const int width = opt_.rig[0].cam.width, height = opt_.rig[0].cam.height;
  const size_t n_cells = stream_of(0).tri_detector->grid_.size();
  struct Work {
    std::vector<size_t> edgelets;
    bool tri = false, tri_match = false, seeds = false;
    int slot = -1;
    StereoTriangulationHip::Job job;
    size_t unit_off = 0;
    double d_med = 0, d_min = 0;
    int max_n_seeds = 0;
  };
  std::vector<Work> work(K);
  // one detector call for the frames `frame_of(w)` of the streams with `wanted(w)`: every cell's best corner / edgelet, by the streams' occupancy grids
  std::vector<uint8_t> occ;
  std::vector<uint64_t> ckeys, ekeys;
  std::vector<float> angles;
  auto detect_cells = [&](const std::function<bool(size_t)>& wanted, const std::function<svoh_frame_t(size_t)>& pyramid_of, DetectorHip& options_of) {
    std::vector<svoh_frame_t> frames;
    std::vector<uint8_t> packed;
    for (size_t w = 0; w < K; ++w) {
      work[w].slot = -1;
      if (!wanted(w)) continue;
      work[w].slot = static_cast<int>(frames.size());
      frames.push_back(pyramid_of(w));
      packed.insert(packed.end(), occ.begin() + static_cast<long>(w * n_cells), occ.begin() + static_cast<long>((w + 1) * n_cells));
    }
    const size_t n = frames.size();
    ckeys.assign(n * n_cells, 0); ekeys.assign(n * n_cells, 0); angles.assign(n * n_cells, 0.f);
    if (!n) return;
    const svoh_detector_options dopt = options_of.abiOptions();
    check(svoh_detect_cells_batch_enqueue(ctx_, static_cast<int>(n), frames.data(), &dopt, packed.data()), "svoh_detect_cells_batch_enqueue");
    check(svoh_detect_cells_batch_collect(ctx_, ckeys.data(), ekeys.data(), angles.data()), "svoh_detect_cells_batch_collect");
    ++device_calls_;
  };
  auto refresh = [&](const std::function<FramePtr(size_t)>& frame_of) {
    std::vector<FramePtr> fr;
    std::vector<std::vector<size_t>> ed;
    for (size_t w = 0; w < K; ++w) if (!work[w].edgelets.empty()) { fr.push_back(frame_of(w)); ed.push_back(work[w].edgelets); }
    if (!fr.empty()) { refreshEdgeletDirections(ctx_, fr, ed); ++device_calls_; }
  };

  // ---- the frame that is not the keyframe upgrades the seeds it hangs on first (:149-154); the triangulation detector's grid
  occ.assign(K * n_cells, 0);
  pool_->run(static_cast<int>(K), [&](int wi) {
    const size_t w = static_cast<size_t>(wi);
    Stream& st = stream_of(w);
    const FrameBundle::Ptr& b = st.bundle;
    if (opt_.landmarks) upgradeSeedsToFeatures(b->at(1 - which[w].second), &st.next_point_id, &work[w].edgelets);
    st.tri_detector->resetGrid();
    st.tri_detector->fillGridWithKeypoints(b->at(0)->px_vec_, b->at(0)->num_features_);
    work[w].tri = st.stereo.wantsFeatures(*b->at(0));
    if (work[w].tri) st.tri_detector->occupancyBytes(occ.data() + w * n_cells);
    else { st.stereo.last_indices_.clear(); st.stereo.last_results_.clear(); st.stereo.last_n_succeeded_ = st.stereo.last_n_failed_ = 0; }
  });
  refresh([&](size_t w) { return stream_of(w).bundle->at(1 - which[w].second); });
  // ---- stereo triangulation (:146-155): new features where the left frame has none, all streams' epipolar searches in one launch
  detect_cells([&](size_t w) { return work[w].tri; }, [&](size_t w) { return stream_of(w).bundle->at(0)->pyramid; }, *stream_of(0).tri_detector);
  pool_->run(static_cast<int>(K), [&](int wi) {
    const size_t w = static_cast<size_t>(wi);
    if (!work[w].tri) return;
    Stream& st = stream_of(w);
    const size_t i = static_cast<size_t>(work[w].slot);
    std::vector<double> px, score, grad;
    std::vector<int32_t> level;
    std::vector<uint8_t> type;
    st.tri_detector->fillFromCells(ckeys.data() + i * n_cells, ekeys.data() + i * n_cells, angles.data() + i * n_cells, width, height, n_cells, px, score, level, grad, type);
    work[w].tri_match = st.stereo.prepare(st.bundle->at(0), st.bundle->at(1), px, score, level, grad, type, &work[w].job);
  });
  {
    size_t n_units = 0, n_pairs = 0;
    for (size_t w = 0; w < K; ++w) if (work[w].tri_match) { work[w].unit_off = n_units; work[w].slot = static_cast<int>(n_pairs++); n_units += work[w].job.n_new; }
    if (n_units) {
      std::vector<svoh_frame_view> v0(n_pairs), v1(n_pairs);
      std::vector<svoh_se3> T(n_pairs * n_pairs);
      for (svoh_se3& t : T) svoh::store_rigid(Transformation{ { 1, 0, 0, 0 }, { 0, 0, 0 } }, t);   // (only a pair's own entry is read)
      std::vector<int32_t> ref_idx(n_units), cur_idx(n_units), level(n_units), result(n_units);
      std::vector<double> px(2 * n_units), f(3 * n_units), grad(2 * n_units), depth(n_units), px_cur(2 * n_units), f_cur(3 * n_units), A(4 * n_units);
      std::vector<uint8_t> type(n_units);
      for (size_t w = 0; w < K; ++w) {
        if (!work[w].tri_match) continue;
        const Frame& f0 = *stream_of(w).bundle->at(0);
        const StereoTriangulationHip::Job& j = work[w].job;
        const size_t p = static_cast<size_t>(work[w].slot), o = work[w].unit_off;
        v0[p] = j.v0; v1[p] = j.v1; T[p * n_pairs + p] = j.T_f1_f0;
        std::fill(ref_idx.begin() + static_cast<long>(o), ref_idx.begin() + static_cast<long>(o + j.n_new), static_cast<int32_t>(p));
        std::fill(cur_idx.begin() + static_cast<long>(o), cur_idx.begin() + static_cast<long>(o + j.n_new), static_cast<int32_t>(p));
        std::copy(f0.px_vec_.begin() + static_cast<long>(2 * j.n_old), f0.px_vec_.begin() + static_cast<long>(2 * (j.n_old + j.n_new)), px.begin() + static_cast<long>(2 * o));
        std::copy(f0.f_vec_.begin() + static_cast<long>(3 * j.n_old), f0.f_vec_.begin() + static_cast<long>(3 * (j.n_old + j.n_new)), f.begin() + static_cast<long>(3 * o));
        std::copy(f0.grad_vec_.begin() + static_cast<long>(2 * j.n_old), f0.grad_vec_.begin() + static_cast<long>(2 * (j.n_old + j.n_new)), grad.begin() + static_cast<long>(2 * o));
        std::copy(f0.level_vec_.begin() + static_cast<long>(j.n_old), f0.level_vec_.begin() + static_cast<long>(j.n_old + j.n_new), level.begin() + static_cast<long>(o));
        std::copy(f0.type_vec_.begin() + static_cast<long>(j.n_old), f0.type_vec_.begin() + static_cast<long>(j.n_old + j.n_new), type.begin() + static_cast<long>(o));
      }
      svoh_feature_batch fb{};
      fb.n = static_cast<int32_t>(n_units);
      fb.ref_frame_idx = ref_idx.data(); fb.cur_frame_idx = cur_idx.data(); fb.n_cur_frames = static_cast<int32_t>(n_pairs);
      fb.px = px.data(); fb.f = f.data(); fb.grad = grad.data(); fb.level = level.data(); fb.type = type.data();
      const StereoTriangulationOptions& so = stream_of(0).stereo.options_;
      const double d_inv[3] = { so.mean_depth_inv, so.min_depth_inv, so.max_depth_inv };
      svoh_epipolar_match_outputs out{};
      out.result = result.data(); out.depth = depth.data(); out.px_cur = px_cur.data(); out.f_cur = f_cur.data(); out.A_cur_ref = A.data();
      const svoh_matcher_options mo = StereoTriangulationHip::matcherOptions();
      check(svoh_epipolar_match_batch(ctx_, &mo, static_cast<int>(n_pairs), v0.data(), v1.data(), T.data(), &fb, d_inv, nullptr, &out), "svoh_epipolar_match_batch");
      ++device_calls_;
      pool_->run(static_cast<int>(K), [&](int wi) {
        const size_t w = static_cast<size_t>(wi);
        if (!work[w].tri_match) return;
        Stream& st = stream_of(w);
        const size_t o = work[w].unit_off;
        st.stereo.finish(st.bundle->at(0), st.bundle->at(1), work[w].job, result.data() + o, depth.data() + o, px_cur.data() + 2 * o, f_cur.data() + 3 * o, A.data() + 4 * o);
      });
    }
  }
  // ---- the keyframe's seeds become landmarks (upgradeSeedsToFeatures, :162), then new seeds in its free cells (depth_filter_->addKeyframe, :167-173)
  occ.assign(K * n_cells, 0);
  pool_->run(static_cast<int>(K), [&](int wi) {
    const size_t w = static_cast<size_t>(wi);
    Stream& st = stream_of(w);
    const FrameBundle::Ptr& b = st.bundle;
    const FramePtr& f = b->at(which[w].second);
    work[w].edgelets.clear();
    if (opt_.landmarks) upgradeSeedsToFeatures(f, &st.next_point_id, &work[w].edgelets);
    work[w].seeds = false;
    if (!scene_depth(*b->at(0), work[w].d_med, work[w].d_min)) return;
    st.seed_detector.resetGrid();
    st.seed_detector.fillGridWithKeypoints(f->px_vec_, f->num_features_);
    work[w].max_n_seeds = opt_.params.max_n_seeds_per_frame - static_cast<int>(f->num_features_);
    if (work[w].max_n_seeds <= 0) return;   // "Skip seed initialization. Have already enough features." (the grid stays as it is, as in initializeSeeds)
    work[w].seeds = true;
    st.seed_detector.occupancyBytes(occ.data() + w * n_cells);
  });
  refresh([&](size_t w) { return stream_of(w).bundle->at(which[w].second); });
  detect_cells([&](size_t w) { return work[w].seeds; }, [&](size_t w) { return stream_of(w).bundle->at(which[w].second)->pyramid; }, stream_of(0).seed_detector);
  pool_->run(static_cast<int>(K), [&](int wi) {
    const size_t w = static_cast<size_t>(wi);
    Stream& st = stream_of(w);
    const FrameBundle::Ptr& b = st.bundle;
    const FramePtr& f = b->at(which[w].second);
    if (work[w].seeds) {
      const size_t i = static_cast<size_t>(work[w].slot), n_old = f->num_features_;
      std::vector<double> px, score, grad;
      std::vector<int32_t> level;
      std::vector<uint8_t> type;
      st.seed_detector.fillFromCells(ckeys.data() + i * n_cells, ekeys.data() + i * n_cells, angles.data() + i * n_cells, width, height, static_cast<size_t>(work[w].max_n_seeds),
                                     px, score, level, grad, type);
      depth_filter_utils::appendSeeds(f, px, score, level, grad, type, static_cast<float>(0.5 * work[w].d_min), static_cast<float>(work[w].d_med));
      for (size_t k = n_old; k < f->num_features_; ++k) { f->seed_ref_vec_[k].keyframe = f; f->seed_ref_vec_[k].seed_id = static_cast<int>(k); }
    }
  });
  // ---- the frames' features are final now: their constant columns go to the device once (a keyframe's seeds are updated by every later pair)
  if (opt_.resident_features) {
    std::vector<int32_t> n;
    std::vector<const double*> px, fv, grad;
    std::vector<const int32_t*> level;
    std::vector<FramePtr> frames;
    for (size_t w = 0; w < K; ++w) {
      const FrameBundle::Ptr& b = stream_of(w).bundle;
      if (b->at(0)->num_features_ == 0 || b->at(1)->num_features_ == 0) continue;
      for (size_t c = 0; c < 2; ++c) {
        const FramePtr& fr = b->at(c);
        frames.push_back(fr);
        n.push_back(static_cast<int32_t>(fr->num_features_)); px.push_back(fr->px_vec_.data()); fv.push_back(fr->f_vec_.data()); grad.push_back(fr->grad_vec_.data()); level.push_back(fr->level_vec_.data());
      }
    }
    if (!frames.empty()) {
      std::vector<svoh_features_t> handles(frames.size(), 0);
      check(svoh_features_upload(ctx_, static_cast<int>(frames.size()), n.data(), px.data(), fv.data(), grad.data(), level.data(), handles.data()), "svoh_features_upload");
      ++device_calls_;
      for (size_t k = 0; k < frames.size(); ++k) frames[k]->features = handles[k];
    }
  }
  // ---- the keyframe window
  pool_->run(static_cast<int>(K), [&](int wi) {
    Stream& st = stream_of(static_cast<size_t>(wi));
    const FrameBundle::Ptr& b = st.bundle;
    const size_t max_kfs = 2 * st.rp0.options_.max_n_kfs;
    for (size_t c = 0; c < 2; ++c) {
      st.kfs.push_back(b->at(c));
      while (st.kfs.size() > max_kfs) {
        for (auto& sr : st.kfs.front()->seed_ref_vec_) sr.keyframe.reset();
        removeObservationsOf(*st.kfs.front());   // (Map::removeKeyframe)
        st.kfs.pop_front();
      }
    }
  });
}

// depth_filter_->updateSeeds(overlap_kfs, new_frames_->at(c)) (frame_handler_stereo.cpp:127-129) for every tracking stream in ONE batch: the
// units are DepthFilterHip::queueUpdateSeeds' (every feature of every visible keyframe), a stream's units name its own frames
void FrontendLockstepStereo::seedUpdate(const std::vector<int>& trk, int c, bool leave_in_flight)
{
  sb_streams_ = trk;
  // where every stream's units and reference frames go, then every stream fills its slices (pool)
  size_t n_units = 0, n_refs = 0;
  std::vector<size_t> ref_off(trk.size());
  bool resident = opt_.resident_features;
  for (size_t w = 0; w < trk.size(); ++w) {
    Stream& st = *streams_[static_cast<size_t>(trk[w])];
    st.seed_frames = st.visible;
    st.seed_counts.clear();
    st.seed_off = n_units;
    ref_off[w] = n_refs;
    for (const FramePtr& rf : st.seed_frames) { st.seed_counts.push_back(rf->num_features_); n_units += rf->num_features_; resident = resident && rf->features != 0; }
    n_refs += st.seed_frames.size();
  }
  if (n_units == 0) { for (int s : trk) { streams_[static_cast<size_t>(s)]->seed_frames.clear(); streams_[static_cast<size_t>(s)]->seed_counts.clear(); } return; }
  // the batch is built in the context's page-locked staging area (no copy between the host arrays and the transfer); with resident keyframe
  // columns (svoh_features_upload at the keyframe step) it is every feature of every visible keyframe, frame after frame (SVOH_BATCH_WHOLE_SETS):
  // per unit only the seed's state and type cross PCIe, and the kernel works in the tile order computed once per keyframe
  check(svoh_matcher_begin_deferred(ctx_), "svoh_matcher_begin_deferred");
  struct CloseSection { svoh_ctx* c; bool armed; ~CloseSection() { if (armed) (void)svoh_matcher_collect(c); } } close_section{ ctx_, true };
  check(svoh_matcher_stage(ctx_, 1, static_cast<int>(n_units), static_cast<int>(n_refs + trk.size()) + 1, resident ? SVOH_STAGE_RESIDENT_COLUMNS : 0, &seed_stage_), "svoh_matcher_stage");
  const svoh_matcher_stage_t& g = seed_stage_;
  std::vector<svoh_frame_view> refs(n_refs), curs(trk.size());
  pool_->run(static_cast<int>(trk.size()), [&](int w) {
    Stream& st = *streams_[static_cast<size_t>(trk[static_cast<size_t>(w)])];
    curs[static_cast<size_t>(st.slot)] = detail::viewOf(*st.bundle->at(static_cast<size_t>(c)));
    size_t off = st.seed_off;
    for (size_t k = 0; k < st.seed_frames.size(); ++k) {
      const Frame& r = *st.seed_frames[k];
      const int32_t ref = static_cast<int32_t>(ref_off[static_cast<size_t>(w)] + k);
      refs[static_cast<size_t>(ref)] = detail::viewOf(r);
      const size_t n = st.seed_counts[k];
      if (g.feature_index) for (size_t i = 0; i < n; ++i) g.cur_frame_idx[off + i] = st.slot;
      else {
        for (size_t i = 0; i < n; ++i) { g.ref_frame_idx[off + i] = ref; g.cur_frame_idx[off + i] = st.slot; }
        memcpy(g.px + 2 * off, r.px_vec_.data(), 16 * n); memcpy(g.f + 3 * off, r.f_vec_.data(), 24 * n); memcpy(g.grad + 2 * off, r.grad_vec_.data(), 16 * n);
        memcpy(g.level + off, r.level_vec_.data(), 4 * n);
      }
      memcpy(g.type + off, r.type_vec_.data(), n);
      memcpy(g.state + 4 * off, r.invmu_sigma2_a_b_vec_.data(), 32 * n);
      off += n;
    }
  });
  svoh_feature_batch fb{};
  fb.n = static_cast<int32_t>(n_units);
  fb.ref_frame_idx = g.ref_frame_idx; fb.cur_frame_idx = g.cur_frame_idx; fb.n_cur_frames = static_cast<int32_t>(trk.size());
  fb.px = g.px; fb.f = g.f; fb.grad = g.grad; fb.level = g.level; fb.type = g.type; fb.feature_index = g.feature_index;
  fb.mem_space = SVOH_MEM_STAGED;
  fb.layout = g.feature_index ? SVOH_BATCH_WHOLE_SETS : SVOH_BATCH_UNITS;
  DepthFilterHip df(ctx_, opt_.params.depth_filter);   // (options only: the batch is the driver's)
  const svoh_depth_filter_options dfo = df.abiOptions(*streams_[static_cast<size_t>(trk[0])]->bundle->at(static_cast<size_t>(c)));
  const svoh_matcher_options mopt = df.getMatcherOptions();
  check(svoh_update_seeds_batch(ctx_, &mopt, &dfo, static_cast<int>(refs.size()), refs.data(), curs.data(), &fb, g.state, g.success, g.result, nullptr), "svoh_update_seeds_batch");
  check(svoh_matcher_flush(ctx_), "svoh_matcher_flush");
  close_section.armed = false;
  ++device_calls_;
  seeds_in_flight_ = true;
  if (!leave_in_flight) collectSeedUpdate();
}

void FrontendLockstepStereo::collectSeedUpdate()
{
  if (!seeds_in_flight_) return;
  seeds_in_flight_ = false;
  check(svoh_matcher_collect(ctx_), "svoh_matcher_collect");
  ++device_calls_;
  const svoh_matcher_stage_t& g = seed_stage_;
  pool_->run(static_cast<int>(sb_streams_.size()), [&](int w) {
    Stream& st = *streams_[static_cast<size_t>(sb_streams_[static_cast<size_t>(w)])];
    size_t off = st.seed_off, n_applied = 0;
    for (size_t k = 0; k < st.seed_frames.size(); ++k) {
      Frame& r = *st.seed_frames[k];
      const size_t n = st.seed_counts[k];
      // (a seed that became a feature while its update was in flight keeps what the upgrade made of it: DepthFilterHip::finishUpdateSeedsNow's rule)
      for (size_t i = 0; i < n; ++i) {
        if (r.type_vec_[i] >= SVOH_FT_EDGELET) continue;
        std::copy(g.state + 4 * (off + i), g.state + 4 * (off + i + 1), r.invmu_sigma2_a_b_vec_.begin() + 4 * i);
        r.type_vec_[i] = g.type[off + i];
        n_applied += g.success[off + i];
      }
      off += n;
    }
    st.seed_frames.clear(); st.seed_counts.clear();
    st.row.n_seed_upd += n_applied;
  });
}

// the second camera's update of the round before: waited for, written back; the rows of that round are complete
void FrontendLockstepStereo::finishSecondSeedUpdate()
{
  collectSeedUpdate();
  for (auto& stp : streams_) {
    Stream& st = *stp;
    if (!st.row_open) continue;
    st.done_rows.push_back(st.row);
    st.row_open = false;
  }
}

void FrontendLockstepStereo::prefetch(const uint8_t* const* next_left, const uint8_t* const* next_right, int pitch)
{
  if (!next_left || !next_right || !prefetched_.empty()) return;
  const int S = numStreams();
  std::vector<const uint8_t*> imgs;
  for (int s = 0; s < S; ++s) {
    if ((next_left[s] == nullptr) != (next_right[s] == nullptr)) throw std::runtime_error("FrontendLockstepStereo::addPairs: a pair needs both images (next_left / next_right)");
    if (next_left[s]) { imgs.push_back(next_left[s]); imgs.push_back(next_right[s]); }
  }
  if (imgs.empty()) return;
  std::vector<svoh_frame_t> handles(imgs.size(), 0);
  check(svoh_build_pyramid_multi_prefetch(ctx_, imgs.data(), static_cast<int>(imgs.size()), opt_.rig[0].cam.width, opt_.rig[0].cam.height, pitch, opt_.images_mem_space,
                                          opt_.params.n_pyr_levels_to_build, SVOH_HALFSAMPLE_REFERENCE, handles.data()), "svoh_build_pyramid_multi_prefetch");
  ++device_calls_;
  prefetched_.swap(handles);   // (only once the call has succeeded: a failed announcement leaves nothing behind)
  prefetched_from_ = imgs;
}

void FrontendLockstepStereo::addPairs(const uint8_t* const* left, const uint8_t* const* right, int pitch, const Transformation* T_imu_world_first, const svoh::Quat* const* imu_prior,
                                      const uint8_t* const* next_left, const uint8_t* const* next_right)
{
  const int S = numStreams();
  device_calls_ = 0;
  // where a round's time goes (sums since construction, ms): pyramids, finish seeds, align, reproject, pose, structure, keyframes, seed updates
  double tp = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
  auto lap = [&](int k) { const double n = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); phase_ms_[k] += n - tp; tp = n; };
  double td = tp;
  auto detail = [&](int k) { const double n = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); detail_ms_[k] += n - td; td = n; };
  auto detail_start = [&]() { td = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  drainReleases();
  if (!left || !right) throw std::runtime_error("FrontendLockstepStereo::addPairs: NULL images");
  std::vector<int> trk, starting;
  std::vector<const uint8_t*> imgs;
  for (int s = 0; s < S; ++s) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    if ((left[s] == nullptr) != (right[s] == nullptr)) throw std::runtime_error("FrontendLockstepStereo::addPairs: a pair needs both images");
    st.active = left[s] != nullptr;
    st.tracking = st.active && st.last;
    st.starting = st.active && !st.last;
    st.slot = -1; st.do_pose = false; st.needs_more = false; st.pose_slot = -1; st.align_result = -1; st.n_reproj = 0; st.n_pose = 0;
    if (st.tracking) { st.slot = static_cast<int>(trk.size()); trk.push_back(s); }
    if (st.starting) { if (!T_imu_world_first) throw std::runtime_error("FrontendLockstepStereo::addPairs: a stream's first pair needs its pose"); starting.push_back(s); }
    if (st.active) { imgs.push_back(left[s]); imgs.push_back(right[s]); }
  }
  const int nT = static_cast<int>(trk.size());

  // ---- pyramids of the round's 2 x (active streams) images: one call
  if (!prefetched_.empty() && prefetched_from_ != imgs) throw std::runtime_error("FrontendLockstepStereo::addPairs: not the pairs that were announced as next_left / next_right");
  if (!imgs.empty()) {
    std::vector<svoh_frame_t> handles(imgs.size(), 0);
    if (!prefetched_.empty()) {   // made during the round before: for exactly the images that were announced
      handles.swap(prefetched_);
      prefetched_.clear(); prefetched_from_.clear();
      check(svoh_prefetch_fence(ctx_), "svoh_prefetch_fence");
    } else {
      check(svoh_build_pyramid_multi(ctx_, imgs.data(), static_cast<int>(imgs.size()), opt_.rig[0].cam.width, opt_.rig[0].cam.height, pitch, opt_.images_mem_space,
                                     opt_.params.n_pyr_levels_to_build, SVOH_HALFSAMPLE_REFERENCE, handles.data()), "svoh_build_pyramid_multi");
      ++device_calls_;
    }
    size_t at = 0;
    for (int s = 0; s < S; ++s) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (!st.active) continue;
      st.bundle.reset(new FrameBundle);
      for (int c = 0; c < 2; ++c) {
        FramePtr frame(new Frame, [this](Frame* f) {
          if (f->pyramid || f->features) {
            std::lock_guard<std::mutex> lock(release_mu_);
            if (f->pyramid) to_release_.push_back(f->pyramid);
            if (f->features) features_to_release_.push_back(f->features);
          }
          delete f;
        });
        frame->pyramid = handles[at++];
        frame->cam = rigOf(s)[static_cast<size_t>(c)].cam;
        frame->set_T_cam_imu(svoh::inverse(rigOf(s)[static_cast<size_t>(c)].T_B_C));
        frame->id_ = static_cast<int>(2 * st.k + static_cast<size_t>(c));
        st.bundle->frames_.push_back(frame);
      }
    }
  }
  lap(0);
  // the pair before: its second seed update is needed from here on (alignment points, candidates)
  finishSecondSeedUpdate();
  lap(1);

  auto close_round = [&]() {
    for (auto& stp : streams_) {
      Stream& st = *stp;
      if (!st.active) continue;
      st.row.n_landmarks = num_landmarks(*st.bundle->at(0)) + num_landmarks(*st.bundle->at(1));
      st.row_open = true;
      st.last = st.bundle; st.bundle.reset();
      ++st.k;
    }
    drainReleases();
  };

  // ---- first pairs: the rig's pose is given, stereo triangulation + seeds (StereoInit's stand-in, as svoh_mini_stereo's first pair)
  {
    std::vector<std::pair<int, size_t>> first;
    for (int s : starting) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      for (const FramePtr& f : st.bundle->frames_) f->T_f_w_ = svoh::mul(f->T_cam_imu(), T_imu_world_first[s]);
      st.row = PairRow(); st.row.k = st.k; st.row.is_kf = true;
      first.emplace_back(s, 0);
    }
    makeKeyframes(first);
    for (int s : starting) { Stream& st = *streams_[static_cast<size_t>(s)]; st.row.alpha = st.img_align.lastResult().alpha; st.row.beta = st.img_align.lastResult().beta; }
  }
  if (nT == 0) { prefetch(next_left, next_right, pitch); close_round(); return; }

  // ---- 1. sparse image alignment of the bundles, each with its stream's IMU rotation prior (frame_handler_base.cpp:610-643)
  detail_start();
  pool_->run(S, [&](int s) {
    Stream& st = *streams_[static_cast<size_t>(s)];
    if (!st.tracking) return;
    st.row = PairRow(); st.row.k = st.k;
    for (size_t c = 0; c < 2; ++c) { st.bundle->at(c)->T_f_w_ = st.last->at(c)->T_f_w_; resolveAlignmentPoints(*st.last->at(c)); }
    st.img_align.reset();
    if (imu_prior && imu_prior[s] && opt_.lambda_rot > 0) {
      Transformation T_prior{ *imu_prior[s], { 0, 0, 0 } };   // T_newimu_lastimu_prior: the rotation is what the weights use
      st.img_align.setWeightedPrior(T_prior, 0.0, 0.0, opt_.lambda_rot, 0.0, 0.0, 0.0);
    }
    st.T_iref_world = st.img_align.prepareRun(st.last, st.bundle, st.align_opt, st.align_pb);
    st.visible.assign(st.kfs.begin(), st.kfs.end());
  });
  detail(0);
  {
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
      check(svoh_sparse_align_enqueue_keyed(ctx_, &streams_[static_cast<size_t>(g.second[0])]->align_opt, static_cast<int>(pbs.size()), pbs.data(), g.first), "svoh_sparse_align_enqueue_keyed");
      ++device_calls_;
    }
    detail(1);
    std::vector<svoh_align_result> results(static_cast<size_t>(nT));
    check(svoh_sparse_align_fetch_all(ctx_, nT, results.data()), "svoh_sparse_align_fetch_all");
    ++device_calls_;
    detail(2);
    prefetch(next_left, next_right, pitch);   // (behind a call that waited: the context's stream is idle, the next pairs cross PCIe beside the rest of the round)
    for (int s : trk) {   // a cluster of workgroups that never completed (status 3): that problem again, one workgroup
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (results[static_cast<size_t>(st.align_result)].status != 3) continue;
      check(svoh_sparse_align_batch(ctx_, &st.align_opt, 1, &st.align_pb, &results[static_cast<size_t>(st.align_result)]), "svoh_sparse_align_batch");
      ++device_calls_;
    }
    pool_->run(S, [&](int s) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (!st.tracking) return;
      st.row.n_aligned = st.img_align.finishRun(results[static_cast<size_t>(st.align_result)], st.bundle, st.T_iref_world);
    });
    detail(3);
  }

  lap(2);
  // ---- 2. reprojection, per camera (frame_handler_base.cpp:645-744): first every stream's camera 0, then every stream's camera 1 -- a
  // stream's second camera sees what its first did to the landmarks' reprojection statistics and to the seeds it matched
  svoh_matcher_stage_t ds{}, ss{};
  for (int c = 0; c < 2; ++c) {
    const size_t cc = static_cast<size_t>(c);
    pool_->run(S, [&](int s) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (!st.tracking) return;
      ReprojectorHip& rp = *st.rp[c];
      rp.discardCandidateProjection();
      st.trash.clear();
      // the third list (the unconverged seeds: the longest, and in the steady state one whose pass is not reached) joins the batch only if the
      // camera's pass reached it in the pair before (ReprojectorHip::reprojectFrames' own policy); a replay that does reach an unplanned pass
      // pauses and gets a batch of its own below.  The passes run in the reference's order either way: same results.
      const bool third = opt_.speculate_all || st.k <= 1 || rp.reachedUnconvergedPass();
      if (third) rp.walkCandidates(st.bundle->at(cc), st.visible, st.trash);
      else rp.walkCandidatesWithoutUnconverged(st.bundle->at(cc), st.visible, st.trash);
      rp.planMatches(st.bundle->at(cc), third ? 3 : 2, false);
    });
    detail(4);
    auto matcher_round = [&](const std::vector<int>& who, bool sort_meanwhile) {
      size_t n_direct = 0, n_seeds = 0, n_refs = 0;
      for (int s : who) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        detail::SpeculativeMatches& sm = st.rp[c]->plannedMatches();
        st.direct_off = n_direct; st.seeds_off = n_seeds; st.ref_off = n_refs;
        n_direct += sm.direct.size(); n_seeds += sm.seeds.size(); n_refs += sm.frames.size();
      }
      ds = svoh_matcher_stage_t{}; ss = svoh_matcher_stage_t{};
      auto sort_lists = [&]() { pool_->run(S, [&](int s) { Stream& st = *streams_[static_cast<size_t>(s)]; if (st.tracking) st.rp[c]->sortCandidateLists(); }); };
      if (n_direct + n_seeds == 0) { if (sort_meanwhile) sort_lists(); return; }
      const svoh_matcher_options mopt = detail::reprojectorMatcherOptions(opt_.params.reprojector_affine_est_offset, true);
      const int max_views = static_cast<int>(n_refs) + nT + 1;
      check(svoh_matcher_begin_deferred(ctx_), "svoh_matcher_begin_deferred");
      struct CloseSection { svoh_ctx* c; bool armed; ~CloseSection() { if (armed) (void)svoh_matcher_collect(c); } } close_section{ ctx_, true };
      if (n_direct) check(svoh_matcher_stage(ctx_, 0, static_cast<int>(n_direct), max_views, SVOH_STAGE_MATCH_OUTPUTS, &ds), "svoh_matcher_stage");
      if (n_seeds) check(svoh_matcher_stage(ctx_, 1, static_cast<int>(n_seeds), max_views, SVOH_STAGE_MATCH_OUTPUTS, &ss), "svoh_matcher_stage");
      std::vector<svoh_frame_view> refs(n_refs ? n_refs : 1), curs(static_cast<size_t>(nT));
      for (int s : trk) { const Stream& st = *streams_[static_cast<size_t>(s)]; curs[static_cast<size_t>(st.slot)] = detail::viewOf(*st.bundle->at(cc)); }
      pool_->run(static_cast<int>(who.size()), [&](int w) {
        Stream& st = *streams_[static_cast<size_t>(who[static_cast<size_t>(w)])];
        detail::SpeculativeMatches& sm = st.rp[c]->plannedMatches();
        for (size_t k = 0; k < sm.frames.size(); ++k) refs[st.ref_off + k] = detail::viewOf(*sm.frames[k]);
        auto copy_batch = [&](const detail::Batch& b, const svoh_matcher_stage_t& g, size_t o) {
          const size_t m = b.size();
          if (!m) return;
          for (size_t i = 0; i < m; ++i) { g.ref_frame_idx[o + i] = b.ref_idx[i] + static_cast<int32_t>(st.ref_off); g.cur_frame_idx[o + i] = st.slot; }
          memcpy(g.px + 2 * o, b.px.data(), 16 * m); memcpy(g.f + 3 * o, b.f.data(), 24 * m); memcpy(g.grad + 2 * o, b.grad.data(), 16 * m);
          memcpy(g.level + o, b.level.data(), 4 * m);
          memcpy(g.type + o, b.type.data(), m);
        };
        copy_batch(sm.direct, ds, st.direct_off);
        if (const size_t m = sm.direct.size()) { memcpy(ds.depth + st.direct_off, sm.direct.depth.data(), 8 * m); memcpy(ds.px_cur + 2 * st.direct_off, sm.direct.px_cur.data(), 16 * m); }
        copy_batch(sm.seeds, ss, st.seeds_off);
        if (const size_t m = sm.seeds.size()) memcpy(ss.state + 4 * st.seeds_off, sm.seeds.state.data(), 32 * m);
      });
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
        check(svoh_match_direct_batch(ctx_, &mopt, static_cast<int>(n_refs), refs.data(), curs.data(), &fb, ds.depth, ds.px_cur, ds.result, ds.f_cur, ds.search_level, nullptr, ds.A_cur_ref),
              "svoh_match_direct_batch");
      }
      if (n_seeds) {
        const svoh_feature_batch fb = batch_of(ss, n_seeds);
        const svoh_depth_filter_options o = detail::reprojectorSeedOptions(*streams_[static_cast<size_t>(trk[0])]->bundle->at(cc), opt_.params.seed_sigma2_thresh);
        const svoh_seed_match_outputs outs{ ss.px_cur, ss.f_cur, ss.search_level, ss.A_cur_ref };
        check(svoh_update_seeds_batch_ex(ctx_, &mopt, &o, static_cast<int>(n_refs), refs.data(), curs.data(), &fb, ss.state, ss.success, ss.result, nullptr, &outs), "svoh_update_seeds_batch_ex");
      }
      check(svoh_matcher_flush(ctx_), "svoh_matcher_flush");
      ++device_calls_;
      if (sort_meanwhile) sort_lists();
      detail(5);
      close_section.armed = false;
      check(svoh_matcher_collect(ctx_), "svoh_matcher_collect");
      ++device_calls_;
      detail(6);
    };
    auto point_outputs = [&](Stream& st) {
      detail::SpeculativeMatches& sm = st.rp[c]->plannedMatches();
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
    matcher_round(trk, true);
    pool_->run(S, [&](int s) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (!st.tracking) return;
      point_outputs(st);
      st.needs_more = st.rp[c]->replayMatchesUntilUnplanned(st.bundle->at(cc));
    });
    for (;;) {   // the streams whose replay reached a pass that was not planned: that pass' list as a batch of their own, then on from there
      std::vector<int> more;
      for (int s : trk) if (streams_[static_cast<size_t>(s)]->needs_more) more.push_back(s);
      if (more.empty()) break;
      paused_passes_ += more.size();
      pool_->run(static_cast<int>(more.size()), [&](int w) { Stream& st = *streams_[static_cast<size_t>(more[static_cast<size_t>(w)])]; st.rp[c]->planPausedPass(st.bundle->at(cc), false, &st.visible); });
      matcher_round(more, false);
      pool_->run(static_cast<int>(more.size()), [&](int w) {
        Stream& st = *streams_[static_cast<size_t>(more[static_cast<size_t>(w)])];
        point_outputs(st);
        st.needs_more = st.rp[c]->resumeReplay(st.bundle->at(cc));
      });
    }
    for (int s : trk) { Stream& st = *streams_[static_cast<size_t>(s)]; st.n_reproj += st.bundle->at(cc)->num_features_; }
    detail(7);
  }

  lap(3);
  // ---- 3. pose optimisation of the rigs (frame_handler_base.cpp:746-790): one batch
  {
    std::vector<svoh_pose_problem> pbs;
    std::vector<int> who;
    pool_->run(S, [&](int s) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      if (!st.tracking) return;
      st.row.n_reproj = st.n_reproj;
      st.do_pose = st.n_reproj >= 10;
      if (st.do_pose) st.pose_optimizer.prepareRun(st.bundle, 2.0, st.pose_opt, st.pose_pb);
    });
    for (int s : trk) { Stream& st = *streams_[static_cast<size_t>(s)]; if (st.do_pose) { st.pose_slot = static_cast<int>(pbs.size()); pbs.push_back(st.pose_pb); who.push_back(s); } }
    if (!pbs.empty()) {
      std::vector<svoh_pose_result> res(pbs.size());
      check(svoh_optimize_pose_batch(ctx_, &streams_[static_cast<size_t>(who[0])]->pose_opt, static_cast<int>(pbs.size()), pbs.data(), res.data()), "svoh_optimize_pose_batch");
      ++device_calls_;
      pool_->run(static_cast<int>(who.size()), [&](int w) {
        Stream& st = *streams_[static_cast<size_t>(who[static_cast<size_t>(w)])];
        st.n_pose = st.pose_optimizer.finishRun(st.bundle, res[static_cast<size_t>(w)]);
        st.row.n_pose = st.n_pose;
      });
    }
  }

  lap(4);
  // ---- 3b. structure optimisation (frame_handler_stereo.cpp:114): optimizeStructure works through a bundle's frames one after the other (a
  // point both cameras see is optimised twice, the second time from the first's result): two batches, each the landmarks of one camera of all streams
  if (opt_.landmarks && opt_.params.structure_optimization_max_pts != 0) {
    for (int s : trk) streams_[static_cast<size_t>(s)]->structure_max_pts = opt_.params.structure_optimization_max_pts;
    for (size_t c = 0; c < 2; ++c) {
      pool_->run(S, [&](int s) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        if (!st.tracking) return;
        st.structure_max_pts = st.structure.gather(*st.bundle->at(c), st.structure_max_pts);
      });
      size_t n_pts = 0, n_views = 0, n_obs = 0;
      for (int s : trk) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        st.structure_off = n_pts; st.structure_view_off = n_views; st.structure_obs_off = n_obs;
        n_pts += st.structure.size(); n_views += st.structure.size() ? st.structure.views.size() : 0; n_obs += st.structure.size() ? st.structure.obs_view.size() : 0;
      }
      if (!n_pts) continue;
      std::vector<svoh_se3> views(n_views);
      std::vector<int32_t> obs_begin(n_pts + 1), obs_view(n_obs);
      std::vector<double> obs_f(3 * n_obs), pos(3 * n_pts);
      pool_->run(S, [&](int s) {
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
      check(svoh_optimize_points_batch(ctx_, 5, 0, static_cast<int>(n_views), views.data(), static_cast<int>(n_pts), obs_begin.data(), obs_view.data(), obs_f.data(), pos.data(), nullptr),
            "svoh_optimize_points_batch");
      ++device_calls_;
      pool_->run(S, [&](int s) {
        Stream& st = *streams_[static_cast<size_t>(s)];
        if (!st.tracking || !st.structure.size()) return;
        st.structure.apply(pos.data() + 3 * st.structure_off);
      });
    }
  }

  lap(5);
  // ---- 4. keyframe rule (svoh_mini_stereo's); a keyframe pair's step comes BEFORE its seed updates, as makeKeyframe does (:162-175)
  {
    std::vector<std::pair<int, size_t>> kf;
    for (int s : trk) {
      Stream& st = *streams_[static_cast<size_t>(s)];
      const bool kf_next = st.k % opt_.kf_every == 0 || st.n_pose < 60;
      if (kf_next) { kf.emplace_back(s, (st.k / opt_.kf_every) % 2); st.row.is_kf = true; }
      st.row.alpha = st.img_align.lastResult().alpha; st.row.beta = st.img_align.lastResult().beta;
    }
    makeKeyframes(kf);
  }

  lap(6);
  // ---- 5. depth filter, per camera (frame_handler_stereo.cpp:127-129): the first camera's update of all streams, waited for (the second
  // starts from its states), then the second camera's, left in flight until the next round
  seedUpdate(trk, 0, false);
  seedUpdate(trk, 1, true);
  close_round();
  lap(7);
}

}  // namespace svo_hip

// ---- C face (svo_hip_lockstep_c.h: svohs_*) ---------------------------------------------------------------------------
#include "svo_hip_lockstep_c.h"

struct svohs_engine { std::unique_ptr<svo_hip::FrontendLockstepStereo> fe; };

namespace {
thread_local std::string g_svohs_error = "no error";
template <class F>
int svohs_guard(F&& f)
{
  try { f(); return SVOH_OK; }
  catch (const std::bad_alloc&) { g_svohs_error = "out of host memory"; return SVOH_ERR_OUT_OF_MEMORY; }
  catch (const std::exception& e) { g_svohs_error = e.what(); return SVOH_ERR_INVALID_ARGUMENT; }
  catch (...) { g_svohs_error = "unknown exception"; return SVOH_ERR_INVALID_ARGUMENT; }
}
}  // namespace

extern "C" {

const char* svohs_last_error(void) { return g_svohs_error.c_str(); }

int svohs_create(svoh_ctx* ctx, int n_streams, const svoh_camera* cams, const svoh_se3* T_B_C, const char* params_yaml, int kf_every, double lambda_rot, int n_workers,
                 int images_pinned, svohs_engine** out)
{
  return svohs_guard([&] {
    if (!out || !cams || !T_B_C) throw std::runtime_error("svohs_create: NULL argument");
    *out = nullptr;
    svo_hip::StereoLockstepOptions lo;
    lo.params = svo_hip::io::frontendParamsFromYaml(params_yaml ? svo_hip::io::parseYaml(params_yaml) : svo_hip::io::YamlNode());
    for (int c = 0; c < 2; ++c) { svo_hip::io::RigCamera rc; rc.label = c ? "cam1" : "cam0"; rc.cam = cams[c]; rc.T_B_C = svoh::load_rigid(T_B_C[c]); lo.rig.push_back(rc); }
    lo.kf_every = kf_every > 0 ? static_cast<size_t>(kf_every) : 8;
    lo.lambda_rot = lambda_rot;
    lo.n_workers = n_workers;
    lo.images_mem_space = images_pinned ? SVOH_MEM_HOST_PINNED : SVOH_MEM_HOST;
    std::unique_ptr<svohs_engine> e(new svohs_engine);
    e->fe.reset(new svo_hip::FrontendLockstepStereo(ctx, n_streams, lo));
    *out = e.release();
  });
}

int svohs_create_rigs(svoh_ctx* ctx, int n_streams, const svoh_camera* cams, const svoh_se3* T_B_C, const char* params_yaml, int kf_every, double lambda_rot, int n_workers,
                      int images_pinned, svohs_engine** out)
{
  return svohs_guard([&] {
    if (!out || !cams || !T_B_C || n_streams < 1) throw std::runtime_error("svohs_create_rigs: NULL argument");
    *out = nullptr;
    svo_hip::StereoLockstepOptions lo;
    lo.params = svo_hip::io::frontendParamsFromYaml(params_yaml ? svo_hip::io::parseYaml(params_yaml) : svo_hip::io::YamlNode());
    for (int s = 0; s < n_streams; ++s) {
      std::vector<svo_hip::io::RigCamera> rig;
      for (int c = 0; c < 2; ++c) { svo_hip::io::RigCamera rc; rc.label = c ? "cam1" : "cam0"; rc.cam = cams[2 * s + c]; rc.T_B_C = svoh::load_rigid(T_B_C[2 * s + c]); rig.push_back(rc); }
      lo.per_stream_rig.push_back(rig);
    }
    lo.rig = lo.per_stream_rig[0];
    lo.kf_every = kf_every > 0 ? static_cast<size_t>(kf_every) : 8;
    lo.lambda_rot = lambda_rot;
    lo.n_workers = n_workers;
    lo.images_mem_space = images_pinned ? SVOH_MEM_HOST_PINNED : SVOH_MEM_HOST;
    std::unique_ptr<svohs_engine> e(new svohs_engine);
    e->fe.reset(new svo_hip::FrontendLockstepStereo(ctx, n_streams, lo));
    *out = e.release();
  });
}

void svohs_destroy(svohs_engine* e) { try { delete e; } catch (...) {} }

int svohs_run_sequence(svohs_engine* e, const uint8_t* base, size_t image_bytes, size_t stream_stride, int n_pairs, int pitch, long k_first, int n_rounds,
                       const svoh_se3* T_imu_world_first, const double* prior_forward, double* round_ms)
{
  return svohs_guard([&] {
    if (!e || !base || n_pairs < 2 || n_rounds < 0 || k_first < 0) throw std::runtime_error("svohs_run_sequence: bad arguments");
    const int S = e->fe->numStreams();
    std::vector<svo_hip::Transformation> T;
    if (T_imu_world_first) for (int s = 0; s < S; ++s) T.push_back(svoh::load_rigid(T_imu_world_first[s]));
    std::vector<const uint8_t*> left(static_cast<size_t>(S)), right(static_cast<size_t>(S)), next_left(static_cast<size_t>(S)), next_right(static_cast<size_t>(S));
    std::vector<const svoh::Quat*> prior(static_cast<size_t>(S), nullptr);
    const long period = 2L * (n_pairs - 1);
    auto pair_of = [&](long k) { const long m = k % period; return m < n_pairs ? m : period - m; };
    for (long k = k_first; k < k_first + n_rounds; ++k) {
      const long f = pair_of(k);
      // R_imu(new)_imu(old): walking forwards from pair f - 1 it is prior_forward[f]; walking backwards from pair f + 1 the inverse of prior_forward[f + 1]
      svoh::Quat q{ 1, 0, 0, 0 };
      bool have = false;
      if (prior_forward && k > 0) {
        const long fp = pair_of(k - 1);
        if (fp == f - 1) { const double* p = prior_forward + 4 * f; q = svoh::Quat{ p[0], p[1], p[2], p[3] }; have = true; }
        else if (fp == f + 1) { const double* p = prior_forward + 4 * (f + 1); q = svoh::Quat{ p[0], -p[1], -p[2], -p[3] }; have = true; }
      }
      for (int s = 0; s < S; ++s) {
        left[static_cast<size_t>(s)] = base + static_cast<size_t>(s) * stream_stride + static_cast<size_t>(2 * f) * image_bytes;
        right[static_cast<size_t>(s)] = base + static_cast<size_t>(s) * stream_stride + static_cast<size_t>(2 * f + 1) * image_bytes;
        prior[static_cast<size_t>(s)] = have ? &q : nullptr;
      }
      const double t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
      const bool has_next = k + 1 < k_first + n_rounds;   // (the next pairs of a replay are known: they go up during the round)
      if (has_next) {
        const long fn = pair_of(k + 1);
        for (int s = 0; s < S; ++s) {
          next_left[static_cast<size_t>(s)] = base + static_cast<size_t>(s) * stream_stride + static_cast<size_t>(2 * fn) * image_bytes;
          next_right[static_cast<size_t>(s)] = base + static_cast<size_t>(s) * stream_stride + static_cast<size_t>(2 * fn + 1) * image_bytes;
        }
      }
      e->fe->addPairs(left.data(), right.data(), pitch, T.empty() ? nullptr : T.data(), prior.data(), has_next ? next_left.data() : nullptr, has_next ? next_right.data() : nullptr);
      if (round_ms) round_ms[k - k_first] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
    }
  });
}

int svohs_pose(svohs_engine* e, int stream, svoh_se3* T_imu_world)
{
  return svohs_guard([&] {
    if (!e || !T_imu_world) throw std::runtime_error("svohs_pose: NULL argument");
    svoh::store_rigid(e->fe->pose(stream), *T_imu_world);
  });
}

int svohs_phase_times(svohs_engine* e, double* ms /* 8 */)
{
  return svohs_guard([&] {
    if (!e || !ms) throw std::runtime_error("svohs_phase_times: NULL argument");
    for (int k = 0; k < svo_hip::FrontendLockstepStereo::kNumPhases; ++k) ms[k] = e->fe->phaseTimes()[k];
  });
}

int svohs_finish(svohs_engine* e)
{
  return svohs_guard([&] {
    if (!e) throw std::runtime_error("svohs_finish: NULL argument");
    e->fe->finish();
  });
}

}  // extern "C"
