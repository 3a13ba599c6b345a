#include "svo_hip_host.h"
#include "svo_hip_host_internal.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <unordered_map>

namespace svo_hip {

SparseImgAlignHip::SparseImgAlignHip(svoh_ctx* ctx, SolverOptions solver_options, SparseImgAlignOptions options)
    : ctx_(ctx), solver_options_(solver_options), options_(options)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("SparseImgAlignHip: NULL svoh_ctx (no CPU fallback exists)");
  reset();
}

void SparseImgAlignHip::reset() { prior_ = svoh_align_prior{}; }

void SparseImgAlignHip::setWeightedPrior(const Transformation& T_cur_ref_prior, double alpha_prior,
                                         double beta_prior, double lambda_rot, double lambda_trans,
                                         double lambda_alpha, double lambda_beta)
{
  prior_.have_prior = 1;
  svoh::store_rigid(T_cur_ref_prior, prior_.T_prior);
  prior_.alpha_prior = alpha_prior;
  prior_.beta_prior = beta_prior;
  prior_.lambda_rot = lambda_rot;
  prior_.lambda_trans = lambda_trans;
  prior_.lambda_alpha = lambda_alpha;
  prior_.lambda_beta = lambda_beta;
}

void SparseImgAlignHip::setCompensation(bool do_compensation)
{
  options_.estimate_illumination_gain = do_compensation;
  options_.estimate_illumination_offset = do_compensation;
}

Transformation SparseImgAlignHip::buildProblem(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames,
                                               int rank, int world, svoh_align_options& opt, svoh_align_problem& pb) const
{
  if (!ref_frames || !cur_frames || ref_frames->empty() || ref_frames->size() != cur_frames->size())
    throw std::runtime_error("SparseImgAlignHip::run: bundles must be non-empty and of equal size");
  if (ref_frames->size() > SVOH_MAX_CAMS) throw std::runtime_error("SparseImgAlignHip::run: too many cameras");
  if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("SparseImgAlignHip: rank / world out of range");

  opt = svoh_align_options{};
  opt.max_level = options_.max_level;
  opt.min_level = options_.min_level;
  opt.patch_size = patch_size_;
  opt.max_iter = static_cast<int32_t>(solver_options_.max_iter);
  opt.eps = solver_options_.eps;
  opt.estimate_illumination_gain = options_.estimate_illumination_gain;
  opt.estimate_illumination_offset = options_.estimate_illumination_offset;
  opt.use_distortion_jacobian = options_.use_distortion_jacobian;
  opt.robustification = options_.robustification;
  opt.weight_scale = options_.weight_scale;

  pb = svoh_align_problem{};
  pb.n_cams = static_cast<int32_t>(ref_frames->size());
  for (size_t i = 0; i < ref_frames->size(); ++i) {
    const Frame& r = *ref_frames->at(i);
    const Frame& c = *cur_frames->at(i);
    svoh_align_camera& cam = pb.cams[i];
    cam.ref_frame = r.pyramid;
    cam.cur_frame = c.pyramid;
    cam.cam = r.cam;
    svoh::store_rigid(r.T_imu_cam(), cam.ref_T_imu_cam);
    svoh::store_rigid(r.T_cam_imu(), cam.ref_T_cam_imu);
    svoh::store_rigid(c.T_cam_imu(), cam.cur_T_cam_imu);
    const svoh::Vec3 p = r.pos();
    cam.ref_pos[0] = p.x; cam.ref_pos[1] = p.y; cam.ref_pos[2] = p.z;
    // this participant's share of the camera's features (the whole list when world == 1)
    const size_t lo = r.num_features_ * static_cast<size_t>(rank) / world;
    const size_t hi = r.num_features_ * static_cast<size_t>(rank + 1) / world;
    cam.n_features = static_cast<int32_t>(hi - lo);
    cam.mem_space = SVOH_MEM_HOST;
    cam.px = r.px_vec_.data() + 2 * lo;
    cam.f = r.f_vec_.data() + 3 * lo;
    cam.pos_world = r.pos_world_.data() + 3 * lo;
    cam.flags = r.alignable_.data() + lo;
    cam.pos_seed_unit = r.pos_seed_unit_.size() == r.num_features_ && r.num_features_ ? r.pos_seed_unit_.data() + lo : nullptr;
  }
  // T_iref_world_ and the optimisation variable (sparse_img_align.cpp:62, 74-75)
  const Transformation T_iref_world = ref_frames->at(0)->T_imu_world();
  const Transformation T_icur_iref = svoh::mul(cur_frames->at(0)->T_imu_world(), svoh::inverse(T_iref_world));
  svoh::store_rigid(T_icur_iref, pb.T_icur_iref);
  pb.alpha_init = alpha_init_;
  pb.beta_init = beta_init_;
  pb.prior = prior_;
  return T_iref_world;
}

size_t SparseImgAlignHip::finishRun(const svoh_align_result& result, const FrameBundle::Ptr& cur_frames, const Transformation& T_iref_world)
{
  last_ = result;
  if (last_.n_fts_to_track == 0) return 0;  // "no features to track" (sparse_img_align.cpp:53-57)
  // f->T_f_w_ = f->T_cam_imu() * state.T_icur_iref * T_iref_world_ (sparse_img_align.cpp:103-106)
  const Transformation T_opt = svoh::load_rigid(last_.T_icur_iref);
  for (const FramePtr& f : cur_frames->frames_) f->T_f_w_ = svoh::mul(svoh::mul(f->T_cam_imu(), T_opt), T_iref_world);
  alpha_init_ = 0.0;  // sparse_img_align.cpp:109-110
  beta_init_ = 0.0;
  return static_cast<size_t>(last_.n_fts_to_track);
}

size_t SparseImgAlignHip::run(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames)
{
  svoh_align_options opt;
  svoh_align_problem pb;
  const Transformation T_iref_world = buildProblem(ref_frames, cur_frames, 0, 1, opt, pb);
  svoh_align_result res{};
  const int rc = svoh_sparse_align_batch(ctx_, &opt, 1, &pb, &res);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_sparse_align_batch: ") + svoh_last_error_string(ctx_));
  return finishRun(res, cur_frames, T_iref_world);
}

size_t SparseImgAlignHip::run(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames, const AfterEnqueue& after_enqueue)
{
  svoh_align_options opt;
  svoh_align_problem pb;
  const Transformation T_iref_world = buildProblem(ref_frames, cur_frames, 0, 1, opt, pb);
  last_run_repeated_ = false;
  int rc = svoh_sparse_align_enqueue(ctx_, &opt, 1, &pb);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_sparse_align_enqueue: ") + svoh_last_error_string(ctx_));
  if (after_enqueue) after_enqueue(T_iref_world);
  rc = svoh_sparse_align_fetch(ctx_, 1, &last_);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_sparse_align_fetch: ") + svoh_last_error_string(ctx_));
  if (last_.status == 3) {   // a cluster that never completed: the blocking entry repeats the launch with one workgroup
    last_run_repeated_ = true;
    rc = svoh_sparse_align_batch(ctx_, &opt, 1, &pb, &last_);
    if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_sparse_align_batch: ") + svoh_last_error_string(ctx_));
  }
  const svoh_align_result res = last_;
  return finishRun(res, cur_frames, T_iref_world);
}

size_t SparseImgAlignHip::runSplit(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames, int rank, int world,
                                   const SumOverParticipants& sum_over_participants)
{
  svoh_align_options opt;
  svoh_align_problem pb;
  const Transformation T_iref_world = buildProblem(ref_frames, cur_frames, rank, world, opt, pb);
  auto check = [this](int rc, const char* what) {
    if (rc != SVOH_OK) throw std::runtime_error(std::string(what) + ": " + svoh_last_error_string(ctx_));
  };
  svoh_align_gn_state* d_state = nullptr;
  double* d_sums = nullptr;
  check(svoh_sparse_align_split_buffers(ctx_, &d_state, &d_sums), "svoh_sparse_align_split_buffers");
  check(svoh_sparse_align_split_init(ctx_, &pb, d_state), "svoh_sparse_align_split_init");
  last_ = svoh_align_result{};
  svoh_align_gn_state st{};
  long long first_visible = -1;
  // SparseImgAlign::run's level loop (sparse_img_align.cpp:80-96) around optimizeGaussNewton
  // (mini_least_squares_solver.hpp:42-107), with the sum over the participants between evaluateError and the solve
  for (int level = opt.max_level; level >= opt.min_level; --level) {
    for (int iter = 0; iter < opt.max_iter; ++iter) {
      check(svoh_sparse_align_partial_sums(ctx_, &opt, &pb, level, 0, d_state, d_sums), "svoh_sparse_align_partial_sums");
      check(svoh_synchronize(ctx_), "svoh_synchronize");
      if (sum_over_participants) sum_over_participants(d_sums, SVOH_ALIGN_SUMS_DOUBLES);
      check(svoh_sparse_align_gn_update(ctx_, &opt, &pb, level, iter, d_sums, d_state, &st), "svoh_sparse_align_gn_update");
      last_.iters[level] = iter + 1;
      last_.n_meas[level] = st.n_meas;
      last_.chi2[level] = st.chi2;
      last_.n_patch_iters += st.n_meas / (patch_size_ * patch_size_);
      if (first_visible < 0) first_visible = st.n_meas / (patch_size_ * patch_size_);
      if (st.level_done) break;
    }
  }
  last_.status = st.status;
  last_.T_icur_iref = st.T_icur_iref;
  last_.alpha = st.alpha;
  last_.beta = st.beta;
  // run() returns the number of selected features; the shares only know the patches that were visible in the
  // first evaluation, summed over the participants -- zero in the same cases
  last_.n_fts_to_track = static_cast<int32_t>(first_visible < 0 ? 0 : first_visible);
  if (last_.n_fts_to_track == 0) { last_.status = 1; return 0; }
  const Transformation T_opt = svoh::load_rigid(last_.T_icur_iref);
  for (const FramePtr& f : cur_frames->frames_) f->T_f_w_ = svoh::mul(svoh::mul(f->T_cam_imu(), T_opt), T_iref_world);
  alpha_init_ = 0.0;
  beta_init_ = 0.0;
  return static_cast<size_t>(last_.n_fts_to_track);
}

// ---- DeviceFrameCache -------------------------------------------------------------
void DeviceFrameCache::release(svoh_frame_t h)
{
  if (h && ctx_) (void)svoh_release_frame(ctx_, h);   // an unknown handle (context torn down first) is not an error here
}

size_t DeviceFrameCache::sweep()
{
  size_t n = 0;
  for (auto it = entries_.begin(); it != entries_.end();) {
    if (it->second.alive.expired()) { release(it->second.handle); it = entries_.erase(it); ++n; }
    else ++it;
  }
  return n;
}

void DeviceFrameCache::clear()
{
  for (auto& kv : entries_) release(kv.second.handle);
  entries_.clear();
}

// ---- DepthFilterHip -------------------------------------------------------------
// The context has ONE deferred matcher section.  A seed update sent off with updateSeedsAsync holds it until it is
// finished; anything else of this layer that needs the section on the same context (ReprojectorHip::reprojectFrames)
// finishes that update first instead of failing with "a deferred section is already open": the owner is looked up here.
namespace {
std::mutex g_async_seed_mu;
std::unordered_map<svoh_ctx*, DepthFilterHip*> g_async_seed_owner;
}  // namespace

void finishPendingSeedUpdate(svoh_ctx* ctx)
{
  DepthFilterHip* owner = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_async_seed_mu);
    auto it = g_async_seed_owner.find(ctx);
    if (it != g_async_seed_owner.end()) owner = it->second;
  }
  if (owner) owner->finishUpdateSeedsEarly();
}

DepthFilterHip::~DepthFilterHip()
{
  if (async_open_) { try { (void)finishUpdateSeedsNow(); } catch (...) {} }
}

DepthFilterHip::DepthFilterHip(svoh_ctx* ctx, const DepthFilterOptions& options) : ctx_(ctx), options_(options)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("DepthFilterHip: NULL svoh_ctx (no CPU fallback exists)");
  if (options_.use_threaded_depthfilter)
    throw std::runtime_error("DepthFilterHip: use_threaded_depthfilter must be false (parity needs the synchronous path)");
  // Matcher::Options defaults (matcher.h:39-54) + DepthFilter ctor (depth_filter.cpp:49-54)
  matcher_options_.align_max_iter = 10;
  matcher_options_.max_epi_search_steps = 100;
  matcher_options_.subpix_refinement = 1;
  matcher_options_.epi_search_edgelet_filtering = 1;
  matcher_options_.epi_search_edgelet_max_angle = 0.7;
  matcher_options_.max_patch_diff_ratio = 2.0;
  matcher_options_.scan_on_unit_sphere = options_.scan_epi_unit_sphere;
  matcher_options_.affine_est_offset = options_.affine_est_offset;
  matcher_options_.affine_est_gain = options_.affine_est_gain;
}

namespace detail {
svoh_frame_view viewOf(const Frame& f)
{
  svoh_frame_view v{};
  v.frame = f.pyramid;
  v.cam = f.cam;
  svoh::store_rigid(f.T_f_w_, v.T_f_w);
  v.seed_mu_range = f.seed_mu_range_;
  v.id = f.id();
  v.features = f.features;
  return v;
}
}  // namespace detail
static svoh_frame_view view_of(const Frame& f) { return detail::viewOf(f); }

size_t DepthFilterHip::updateSeeds(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame)
{
  updateSeedsAsync(ref_frames_with_seeds, cur_frame);
  return finishUpdateSeeds();
}

void DepthFilterHip::updateSeedsAsync(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame)
{
  if (prepared_) {
    // prepareUpdateSeeds has queued the batch: the current frame's final pose, and off it goes
    if (!cur_frame || cur_frame.get() != prepared_cur_ || ref_frames_with_seeds.size() != pending_.frames.size())
      throw std::runtime_error("DepthFilterHip::updateSeedsAsync: not the update that prepareUpdateSeeds has queued");
    prepared_ = false; prepared_cur_ = nullptr;
    if (!async_open_) return;   // (nothing to update: no seeds)
    const svoh_frame_view cur = view_of(*cur_frame);
    int rc = svoh_matcher_deferred_set_cur_frame(ctx_, &cur);
    if (rc == SVOH_OK) rc = svoh_matcher_flush(ctx_);
    if (rc != SVOH_OK) {
      const std::string msg = svoh_last_error_string(ctx_);
      (void)svoh_matcher_collect(ctx_);
      async_open_ = false;
      { std::lock_guard<std::mutex> lock(g_async_seed_mu); g_async_seed_owner.erase(ctx_); }
      pending_.frames.clear();
      throw std::runtime_error("DepthFilterHip::updateSeedsAsync (prepared): " + msg);
    }
    return;
  }
  queueUpdateSeeds(ref_frames_with_seeds, cur_frame, true);
}

int32_t DepthFilterHip::unitOfPendingSeed(const Frame& keyframe, size_t seed_id) const
{
  if (!async_open_ || prepared_) return -1;
  size_t off = 0;
  for (size_t k = 0; k < pending_.frames.size(); ++k) {
    if (pending_.frames[k].get() == &keyframe) return seed_id < pending_.counts[k] ? static_cast<int32_t>(off + seed_id) : -1;
    off += pending_.counts[k];
  }
  return -1;
}

void DepthFilterHip::prepareUpdateSeeds(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame)
{
  if (prepared_) throw std::runtime_error("DepthFilterHip::prepareUpdateSeeds: an update is prepared already");
  queueUpdateSeeds(ref_frames_with_seeds, cur_frame, false);
  prepared_ = true;
  prepared_cur_ = cur_frame.get();
}

svoh_depth_filter_options DepthFilterHip::abiOptions(const Frame& cur_frame)
{
  px_error_angle_ = updateSeedPxErrorAngle(cur_frame);
  have_px_error_angle_ = true;
  svoh_depth_filter_options o{};
  o.seed_convergence_sigma2_thresh = options_.seed_convergence_sigma2_thresh;
  o.mappoint_convergence_sigma2_thresh = options_.mappoint_convergence_sigma2_thresh;
  o.px_error_angle = px_error_angle_;
  o.check_visibility = 1; o.check_convergence = 0; o.use_vogiatzis_update = 1;  // depth_filter.cpp:224-225
  return o;
}

void DepthFilterHip::queueUpdateSeeds(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame, bool send_off)
{
  if (!cur_frame) throw std::runtime_error("DepthFilterHip::updateSeeds: NULL current frame");
  if (async_open_ || finished_early_) throw std::runtime_error("DepthFilterHip::updateSeedsAsync: the previous update has not been finished");
  px_error_angle_ = updateSeedPxErrorAngle(*cur_frame);
  have_px_error_angle_ = true;
  Pending& q = pending_;
  q.frames = ref_frames_with_seeds;
  q.counts.clear();
  q.refs.clear(); q.ref_idx.clear(); q.level.clear(); q.px.clear(); q.f.clear(); q.grad.clear(); q.state.clear(); q.type.clear();
  for (size_t k = 0; k < ref_frames_with_seeds.size(); ++k) {
    const Frame& r = *ref_frames_with_seeds[k];
    q.refs.push_back(view_of(r));
    const size_t n = r.num_features_;
    q.counts.push_back(n);
    q.ref_idx.insert(q.ref_idx.end(), n, static_cast<int32_t>(k));
    q.px.insert(q.px.end(), r.px_vec_.begin(), r.px_vec_.begin() + 2 * n);
    q.f.insert(q.f.end(), r.f_vec_.begin(), r.f_vec_.begin() + 3 * n);
    q.grad.insert(q.grad.end(), r.grad_vec_.begin(), r.grad_vec_.begin() + 2 * n);
    q.level.insert(q.level.end(), r.level_vec_.begin(), r.level_vec_.begin() + n);
    q.type.insert(q.type.end(), r.type_vec_.begin(), r.type_vec_.begin() + n);
    q.state.insert(q.state.end(), r.invmu_sigma2_a_b_vec_.begin(), r.invmu_sigma2_a_b_vec_.begin() + 4 * n);
  }
  const size_t n_total = q.ref_idx.size();
  last_results_.assign(n_total, SVOH_MATCH_NOT_RUN);
  q.n_success = 0;
  if (n_total == 0) { q.frames.clear(); return; }
  svoh_feature_batch fb{};
  fb.n = static_cast<int32_t>(n_total);
  fb.ref_frame_idx = q.ref_idx.data(); fb.px = q.px.data(); fb.f = q.f.data(); fb.grad = q.grad.data();
  fb.level = q.level.data(); fb.type = q.type.data();
  const svoh_depth_filter_options o = abiOptions(*cur_frame);
  const svoh_frame_view cur = view_of(*cur_frame);
  q.success.assign(n_total, 0);
  // queued in a deferred section and sent to the device at once (svoh_matcher_flush): the kernel runs while the caller
  // goes on; finishUpdateSeeds waits for it and puts the results where the reference's loop leaves them
  if (svoh_matcher_begin_deferred(ctx_) != SVOH_OK) throw std::runtime_error(std::string("svoh_matcher_begin_deferred: ") + svoh_last_error_string(ctx_));
  async_open_ = true;
  { std::lock_guard<std::mutex> lock(g_async_seed_mu); g_async_seed_owner[ctx_] = this; }
  int rc = svoh_update_seeds_batch(ctx_, &matcher_options_, &o, static_cast<int>(q.refs.size()), q.refs.data(), &cur, &fb,
                                   q.state.data(), q.success.data(), last_results_.data(), &q.n_success);
  if (rc == SVOH_OK && send_off) rc = svoh_matcher_flush(ctx_);
  if (rc != SVOH_OK) {
    const std::string msg = svoh_last_error_string(ctx_);
    (void)svoh_matcher_collect(ctx_);
    async_open_ = false;
    { std::lock_guard<std::mutex> lock(g_async_seed_mu); g_async_seed_owner.erase(ctx_); }
    q.frames.clear();
    throw std::runtime_error("svoh_update_seeds_batch: " + msg);
  }
}

void DepthFilterHip::finishUpdateSeedsEarly()
{
  if (!async_open_) return;
  if (prepared_) {
    // queued by prepareUpdateSeeds and not yet given its frame's pose: what it would compute is not an update of anything.
    // It is dropped, and NOT remembered as finished -- the updateSeedsAsync that follows finds nothing prepared and queues
    // the update afresh (it used to find "finished early" and throw, the frame's update lost).
    (void)finishUpdateSeedsNow();
    return;
  }
  early_count_ = finishUpdateSeedsNow();
  finished_early_ = true;
}

size_t DepthFilterHip::finishUpdateSeeds()
{
  if (finished_early_) {   // someone needed the context's deferred section in between (finishPendingSeedUpdate)
    finished_early_ = false;
    return early_count_;
  }
  return finishUpdateSeedsNow();
}

size_t DepthFilterHip::finishUpdateSeedsNow()
{
  Pending& q = pending_;
  if (!async_open_) { q.frames.clear(); prepared_ = false; prepared_cur_ = nullptr; return 0; }
  async_open_ = false;
  { std::lock_guard<std::mutex> lock(g_async_seed_mu); g_async_seed_owner.erase(ctx_); }
  if (svoh_matcher_collect(ctx_) != SVOH_OK) { q.frames.clear(); prepared_ = false; throw std::runtime_error(std::string("svoh_matcher_collect: ") + svoh_last_error_string(ctx_)); }
  if (prepared_) {   // queued by prepareUpdateSeeds and never given its frame's pose: what it computed is not an update of anything
    prepared_ = false; prepared_cur_ = nullptr;
    q.frames.clear();
    return 0;
  }
  // scatter back in place (ref_frame.invmu_sigma2_a_b_vec_.col(i), type_vec_[i]); a frame's block is the features
  // it had when the update was queued (features are only ever appended)
  size_t off = 0, n_applied = 0;
  for (size_t k = 0; k < q.frames.size(); ++k) {
    Frame& r = *q.frames[k];
    const size_t n = q.counts[k];
    // (a seed that became a feature while its update was in flight -- upgradeSeedsToFeatures at a keyframe made between the
    // update's launch and here -- keeps what the upgrade made of it; units that were no seeds when the update was queued
    // are not touched by the device either)
    for (size_t i = 0; i < n; ++i) {
      if (r.type_vec_[i] >= SVOH_FT_EDGELET) continue;
      std::copy(q.state.begin() + 4 * (off + i), q.state.begin() + 4 * (off + i + 1), r.invmu_sigma2_a_b_vec_.begin() + 4 * i);
      r.type_vec_[i] = q.type[off + i];
      n_applied += q.success[off + i];
    }
    off += n;
  }
  q.frames.clear();
  return n_applied;   // the updates that succeeded AND were applied (q.n_success also counts seeds that became features meanwhile)
}

// ---- FeatureTracker ---------------------------------------------------------------
FeatureTrackerHip::FeatureTrackerHip(svoh_ctx* ctx, const FeatureTrackerOptions& options, size_t bundle_size)
    : ctx_(ctx), options_(options), bundle_size_(bundle_size), active_tracks_(bundle_size), terminated_tracks_(bundle_size)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("FeatureTrackerHip: NULL svoh_ctx (no CPU fallback exists)");
}

size_t FeatureTrackerHip::getTotalActiveTracks() const
{
  size_t n = 0;
  for (const FeatureTracks& t : active_tracks_) n += t.size();
  return n;
}

size_t FeatureTrackerHip::initializeNewTracks(const FrameBundle::Ptr& nframe, const std::vector<size_t>& n_old_per_frame)
{
  if (!nframe || nframe->size() != bundle_size_ || n_old_per_frame.size() != bundle_size_)
    throw std::runtime_error("FeatureTrackerHip::initializeNewTracks: bundle size mismatch");
  for (size_t frame_index = 0; frame_index < bundle_size_; ++frame_index) {
    const FramePtr& frame = nframe->at(frame_index);
    FeatureTracks& tracks = active_tracks_[frame_index];
    if (frame->track_id_vec_.size() < frame->num_features_) frame->track_id_vec_.resize(frame->num_features_, -1);
    for (size_t feature_index = n_old_per_frame[frame_index]; feature_index < frame->num_features_; ++feature_index) {
      const int new_track_id = next_track_id_++;
      tracks.emplace_back(new_track_id);
      tracks.back().pushBack(nframe, frame_index, feature_index);
      frame->track_id_vec_[feature_index] = new_track_id;
    }
  }
  return getTotalActiveTracks();
}

void FeatureTrackerHip::trackAndDetect(const FrameBundle::Ptr& nframe_kp1)
{
  const size_t n_tracked = trackFrameBundle(nframe_kp1);
  if (n_tracked < options_.min_tracks_to_detect_new_features) {
    if (options_.reset_before_detection) {
      resetActiveTracks();
      for (const FramePtr& frame : nframe_kp1->frames_) {   // frame->clearFeatureStorage()
        frame->num_features_ = 0;
        frame->px_vec_.clear(); frame->f_vec_.clear(); frame->grad_vec_.clear(); frame->level_vec_.clear();
        frame->type_vec_.clear(); frame->score_vec_.clear(); frame->track_id_vec_.clear();
        frame->landmark_vec_.clear(); frame->seed_ref_vec_.clear(); frame->invmu_sigma2_a_b_vec_.clear();
      }
    }
    initializeNewTracks(nframe_kp1);
  }
}

size_t FeatureTrackerHip::initializeNewTracks(const FrameBundle::Ptr& nframe)
{
  if (!nframe || nframe->size() != bundle_size_ || detectors_.size() != bundle_size_)
    throw std::runtime_error("FeatureTrackerHip::initializeNewTracks: bundle / detector count mismatch");
  std::vector<size_t> n_old(bundle_size_);
  for (size_t frame_index = 0; frame_index < bundle_size_; ++frame_index) {
    const FramePtr& frame = nframe->at(frame_index);
    DetectorHip& det = *detectors_[frame_index];
    det.resetGrid();
    det.fillGridWithKeypoints(frame->px_vec_, frame->num_features_);
    n_old[frame_index] = frame->num_features_;
    // detect(frame->img_pyr_, frame->getMask(), grid_.size(), new_px, ...) and append; the frame-level overload
    // also computes the normalised bearing vectors (feature_tracker.cpp:143-166)
    frame->grad_vec_.resize(2 * frame->num_features_, 0.0);
    frame->level_vec_.resize(frame->num_features_, 0);
    frame->type_vec_.resize(frame->num_features_, SVOH_FT_CORNER);
    frame->score_vec_.resize(frame->num_features_, 0.0);
    std::vector<int> ids = frame->track_id_vec_;
    det.detect(frame);
    ids.resize(frame->num_features_, -1);
    frame->track_id_vec_ = ids;
  }
  return initializeNewTracks(nframe, n_old);
}

size_t FeatureTrackerHip::trackFrameBundle(const FrameBundle::Ptr& nframe_kp1)
{
  if (!nframe_kp1 || nframe_kp1->size() != bundle_size_) throw std::runtime_error("FeatureTrackerHip: bundle size mismatch");
  resetTerminatedTracks();

  // gather every track of every frame of the bundle (feature_tracker.cpp:64-80)
  std::vector<svoh_frame_t> ref_frames, cur_frames;
  std::vector<int32_t> px_ref;
  std::vector<double> px_cur;
  for (size_t frame_index = 0; frame_index < bundle_size_; ++frame_index) {
    const FramePtr& cur_frame = nframe_kp1->at(frame_index);
    for (const FeatureTrack& track : active_tracks_[frame_index]) {
      const FeatureRef& ref_observation = options_.klt_template_is_first_observation ? track.front() : track.back();
      ref_frames.push_back(ref_observation.getFrame()->pyramid);
      cur_frames.push_back(cur_frame->pyramid);
      px_ref.push_back(static_cast<int32_t>(ref_observation.getPx()[0]));  // getPx().cast<int>(): truncation
      px_ref.push_back(static_cast<int32_t>(ref_observation.getPx()[1]));
      px_cur.push_back(track.back().getPx()[0]);
      px_cur.push_back(track.back().getPx()[1]);
    }
  }
  const size_t n = ref_frames.size();
  std::vector<uint8_t> status(n, 0);
  if (n) {
    svoh_klt_options o{};
    o.max_level = options_.klt_max_level; o.min_level = options_.klt_min_level; o.max_iter = options_.klt_max_iter;
    o.min_update_squared = static_cast<float>(options_.klt_min_update_squared);
    for (size_t l = 0; l < options_.klt_patch_sizes.size() && l < SVOH_MAX_LEVELS; ++l) o.patch_sizes[l] = options_.klt_patch_sizes[l];
    const int rc = svoh_klt_track_multi(ctx_, &o, static_cast<int>(n), ref_frames.data(), cur_frames.data(), px_ref.data(),
                                        px_cur.data(), status.data());
    if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_klt_track_multi: ") + svoh_last_error_string(ctx_));
  }

  // the reference's bookkeeping, frame by frame and track by track (feature_tracker.cpp:86-118)
  size_t t = 0;
  for (size_t frame_index = 0; frame_index < bundle_size_; ++frame_index) {
    FeatureTracks& tracks = active_tracks_[frame_index];
    const FramePtr& cur_frame = nframe_kp1->at(frame_index);
    std::vector<double> new_keypoints, new_scores;
    std::vector<int> new_track_ids;
    FeatureTracks kept;
    size_t new_keypoints_counter = 0;
    for (size_t track_index = 0; track_index < tracks.size(); ++track_index, ++t) {
      FeatureTrack& track = tracks[track_index];
      const FeatureRef& ref_observation = options_.klt_template_is_first_observation ? track.front() : track.back();
      if (status[t]) {
        new_keypoints.push_back(px_cur[2 * t]); new_keypoints.push_back(px_cur[2 * t + 1]);
        const Frame& rf = *ref_observation.getFrame();
        new_scores.push_back(ref_observation.getFeatureIndex() < rf.score_vec_.size() ? rf.score_vec_[ref_observation.getFeatureIndex()] : 0.0);
        new_track_ids.push_back(track.getTrackId());
        track.pushBack(nframe_kp1, frame_index, new_keypoints_counter);
        ++new_keypoints_counter;
        kept.push_back(track);
      } else {
        terminated_tracks_[frame_index].push_back(track);
      }
    }
    tracks.swap(kept);
    // insert the new keypoints in the frame (resizeFeatureStorage + assignments)
    cur_frame->px_vec_ = new_keypoints;
    cur_frame->score_vec_ = new_scores;
    cur_frame->track_id_vec_ = new_track_ids;
    cur_frame->num_features_ = new_keypoints_counter;
    // frame_utils::computeNormalizedBearingVectors (frame.cpp:427-439)
    cur_frame->f_vec_.resize(3 * new_keypoints_counter);
    const svoh::CamModel cm = svoh::load_camera(cur_frame->cam);
    for (size_t i = 0; i < new_keypoints_counter; ++i) {
      svoh::Vec3 f = svoh::back_project3(cm, new_keypoints[2 * i], new_keypoints[2 * i + 1]);
      const double nn = sqrt(f.x * f.x + f.y * f.y + f.z * f.z);
      cur_frame->f_vec_[3 * i] = f.x / nn; cur_frame->f_vec_[3 * i + 1] = f.y / nn; cur_frame->f_vec_[3 * i + 2] = f.z / nn;
    }
  }
  return getTotalActiveTracks();
}

// ---- feature detector ---------------------------------------------------------------
DetectorHip::DetectorHip(svoh_ctx* ctx, const DetectorOptions& options, int image_width, int image_height)
    : grid_(static_cast<int>(options.cell_size), OccupandyGrid2D::getNCell(image_width, static_cast<int>(options.cell_size)),
            OccupandyGrid2D::getNCell(image_height, static_cast<int>(options.cell_size))),
      ctx_(ctx), options_(options)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("DetectorHip: NULL svoh_ctx (no CPU fallback exists)");
}

void DetectorHip::detect(svoh_frame_t img_pyr, const uint8_t* mask, int mask_pitch, size_t max_n_features,
                         std::vector<double>& px_vec, std::vector<double>& score_vec, std::vector<int32_t>& level_vec,
                         std::vector<double>& grad_vec, std::vector<uint8_t>& types_vec)
{
  const svoh_detector_options o = abiOptions();
  const size_t n_cells = grid_.size();
  std::vector<uint8_t> occ(n_cells);
  occupancyBytes(occ.data());
  std::vector<double> px(2 * n_cells), score(n_cells), grad(2 * n_cells);
  std::vector<int32_t> level(n_cells);
  std::vector<uint8_t> type(n_cells);
  int32_t n = 0;
  const int rc = svoh_detect_features(ctx_, img_pyr, &o, occ.data(), mask, mask_pitch,
                                      static_cast<int>(std::min<size_t>(max_n_features, n_cells)), px.data(), score.data(),
                                      level.data(), grad.data(), type.data(), &n);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_detect_features: ") + svoh_last_error_string(ctx_));
  px_vec.insert(px_vec.end(), px.begin(), px.begin() + 2 * n);
  score_vec.insert(score_vec.end(), score.begin(), score.begin() + n);
  level_vec.insert(level_vec.end(), level.begin(), level.begin() + n);
  grad_vec.insert(grad_vec.end(), grad.begin(), grad.begin() + 2 * n);
  types_vec.insert(types_vec.end(), type.begin(), type.begin() + n);
  resetGrid();   // FastDetector / FastGradDetector::detect end with resetGrid()
}

svoh_detector_options DetectorHip::abiOptions() const
{
  svoh_detector_options o{};
  o.cell_size = static_cast<int32_t>(options_.cell_size);
  o.max_level = options_.max_level; o.min_level = options_.min_level; o.border = options_.border;
  o.detect_edgelets = options_.detector_type == DetectorType::kFastGrad;
  o.threshold_primary = options_.threshold_primary; o.threshold_secondary = options_.threshold_secondary;
  return o;
}

void DetectorHip::occupancyBytes(uint8_t* out) const
{
  for (size_t k = 0; k < grid_.size(); ++k) out[k] = grid_.isOccupied(k);
}

void DetectorHip::fillFromCells(const uint64_t* corner_keys, const uint64_t* edge_keys, const float* edge_angles, int width, int height,
                                size_t max_n_features, std::vector<double>& px_vec, std::vector<double>& score_vec, std::vector<int32_t>& level_vec,
                                std::vector<double>& grad_vec, std::vector<uint8_t>& types_vec)
{
  const svoh_detector_options o = abiOptions();
  const size_t n_cells = grid_.size();
  std::vector<double> px(2 * n_cells), score(n_cells), grad(2 * n_cells);
  std::vector<int32_t> level(n_cells);
  std::vector<uint8_t> type(n_cells);
  int32_t n = 0;
  const int rc = svoh_detect_fill_features(&o, width, height, corner_keys, edge_keys, edge_angles, static_cast<int>(std::min<size_t>(max_n_features, n_cells)),
                                           px.data(), score.data(), level.data(), grad.data(), type.data(), &n);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_detect_fill_features: ") + svoh_last_error_string(nullptr));
  px_vec.insert(px_vec.end(), px.begin(), px.begin() + 2 * n);
  score_vec.insert(score_vec.end(), score.begin(), score.begin() + n);
  level_vec.insert(level_vec.end(), level.begin(), level.begin() + n);
  grad_vec.insert(grad_vec.end(), grad.begin(), grad.begin() + 2 * n);
  types_vec.insert(types_vec.end(), type.begin(), type.begin() + n);
  resetGrid();   // FastDetector / FastGradDetector::detect end with resetGrid()
}

void DetectorHip::fillGridWithKeypoints(const std::vector<double>& px_vec, size_t n)
{
  for (size_t i = 0; i < n; ++i)
    grid_.setOccupied(grid_.getCellIndex(static_cast<int>(px_vec[2 * i]), static_cast<int>(px_vec[2 * i + 1]), 1));
}

void DetectorHip::detect(const FramePtr& frame)
{
  if (!frame) throw std::runtime_error("DetectorHip::detect: NULL frame");
  // the reference passes the frame's own (empty or resized) members: the new features replace the old ones
  std::vector<double> px = std::move(frame->px_vec_), score = std::move(frame->score_vec_), grad = std::move(frame->grad_vec_);
  std::vector<int32_t> level = std::move(frame->level_vec_);
  std::vector<uint8_t> type = std::move(frame->type_vec_);
  px.resize(2 * frame->num_features_); score.resize(frame->num_features_); grad.resize(2 * frame->num_features_);
  level.resize(frame->num_features_); type.resize(frame->num_features_);
  detect(frame->pyramid, nullptr, 0, grid_.size(), px, score, level, grad, type);
  frame->px_vec_ = px; frame->score_vec_ = score; frame->grad_vec_ = grad; frame->level_vec_ = level; frame->type_vec_ = type;
  frame->num_features_ = level.size();
  frame->landmark_vec_.assign(frame->num_features_, nullptr);
  frame->seed_ref_vec_.assign(frame->num_features_, Frame::SeedRef());
  frame->invmu_sigma2_a_b_vec_.resize(4 * frame->num_features_);
  frame->f_vec_.resize(3 * frame->num_features_);
  const svoh::CamModel cm = svoh::load_camera(frame->cam);
  for (size_t i = 0; i < frame->num_features_; ++i) {   // frame_utils::computeNormalizedBearingVectors
    const svoh::Vec3 f = svoh::back_project3(cm, px[2 * i], px[2 * i + 1]);
    const double nn = sqrt(f.x * f.x + f.y * f.y + f.z * f.z);
    frame->f_vec_[3 * i] = f.x / nn; frame->f_vec_[3 * i + 1] = f.y / nn; frame->f_vec_[3 * i + 2] = f.z / nn;
  }
}

// ---- StereoTriangulationHip (stereo_triangulation.cpp:16-140) -----------------------------------
StereoTriangulationHip::StereoTriangulationHip(svoh_ctx* ctx, const StereoTriangulationOptions& options,
                                               const std::shared_ptr<DetectorHip>& feature_detector)
  : options_(options), feature_detector_(feature_detector), ctx_(ctx)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("StereoTriangulationHip: NULL svoh_ctx (no CPU fallback exists)");
  shuffle_ = [](std::vector<size_t>& indices, size_t n_corners) {
    // std::random_shuffle(first, last) as libstdc++ implements it on rand() (stereo_triangulation.cpp:77-78)
    auto rs = [](std::vector<size_t>::iterator first, std::vector<size_t>::iterator last) {
      if (first == last) return;
      for (auto i = first + 1; i != last; ++i) {
        auto j = first + std::rand() % ((i - first) + 1);
        if (i != j) std::iter_swap(i, j);
      }
    };
    rs(indices.begin(), indices.begin() + static_cast<long>(n_corners));
    rs(indices.begin() + static_cast<long>(n_corners), indices.end());
  };
}

namespace {
size_t numLandmarksOf(const Frame& f)   // Frame::numLandmarks (frame.cpp:144-151)
{
  size_t n = 0;
  for (size_t i = 0; i < f.num_features_ && i < f.landmark_vec_.size(); ++i) n += f.landmark_vec_[i] != nullptr;
  return n;
}
}  // namespace

bool StereoTriangulationHip::wantsFeatures(const Frame& frame0) const { return numLandmarksOf(frame0) < options_.triangulate_n_features; }   // else "sufficient number of features": no effect

svoh_matcher_options StereoTriangulationHip::matcherOptions()
{
  svoh_matcher_options mo{};   // Matcher::Options defaults (matcher.h:39-54) + :93-94
  mo.align_max_iter = 10; mo.max_epi_search_steps = 500; mo.subpix_refinement = 1; mo.epi_search_edgelet_filtering = 1;
  mo.scan_on_unit_sphere = 1; mo.affine_est_offset = 1; mo.affine_est_gain = 0;
  mo.epi_search_edgelet_max_angle = 0.7; mo.max_patch_diff_ratio = 2.0;
  return mo;
}

bool StereoTriangulationHip::prepare(const FramePtr& frame0, const FramePtr& frame1, const std::vector<double>& new_px, const std::vector<double>& new_scores,
                                     const std::vector<int32_t>& new_levels, const std::vector<double>& new_grads, const std::vector<uint8_t>& new_types, Job* job)
{
  last_indices_.clear(); last_results_.clear(); last_n_succeeded_ = last_n_failed_ = 0;
  *job = Job();
  if (!frame0 || !frame1) throw std::runtime_error("StereoTriangulationHip: NULL frame");
  const size_t n_landmarks0 = numLandmarksOf(*frame0);
  if (n_landmarks0 >= options_.triangulate_n_features) return false;
  const size_t n_new = new_levels.size();
  if (n_new == 0) return false;   // "Stereo Triangulation: No features detected."

  // add them to the first frame (:56-69)
  Frame& f0 = *frame0;
  const size_t n_old = f0.num_features_;
  const size_t n0 = n_old + n_new;
  f0.px_vec_.resize(2 * n_old); f0.px_vec_.insert(f0.px_vec_.end(), new_px.begin(), new_px.end());
  f0.grad_vec_.resize(2 * n_old); f0.grad_vec_.insert(f0.grad_vec_.end(), new_grads.begin(), new_grads.end());
  f0.score_vec_.resize(n_old); f0.score_vec_.insert(f0.score_vec_.end(), new_scores.begin(), new_scores.end());
  f0.level_vec_.resize(n_old); f0.level_vec_.insert(f0.level_vec_.end(), new_levels.begin(), new_levels.end());
  f0.type_vec_.resize(n_old); f0.type_vec_.insert(f0.type_vec_.end(), new_types.begin(), new_types.end());
  f0.f_vec_.resize(3 * n0);
  const svoh::CamModel cm0 = svoh::load_camera(f0.cam);
  for (size_t i = n_old; i < n0; ++i) {   // frame_utils::computeNormalizedBearingVectors
    const svoh::Vec3 f = svoh::back_project3(cm0, f0.px_vec_[2 * i], f0.px_vec_[2 * i + 1]);
    const double nn = sqrt(f.x * f.x + f.y * f.y + f.z * f.z);
    f0.f_vec_[3 * i] = f.x / nn; f0.f_vec_[3 * i + 1] = f.y / nn; f0.f_vec_[3 * i + 2] = f.z / nn;
  }
  f0.landmark_vec_.resize(n0); f0.seed_ref_vec_.resize(n0); f0.track_id_vec_.resize(n0, -1);
  f0.invmu_sigma2_a_b_vec_.resize(4 * n0);
  f0.num_features_ = n0;

  // visiting order: corners first, each part shuffled (:71-78)
  job->indices.resize(n_new);
  for (size_t k = 0; k < n_new; ++k) job->indices[k] = n_old + k;
  const size_t n_corners = static_cast<size_t>(std::count(new_types.begin(), new_types.end(), static_cast<uint8_t>(SVOH_FT_CORNER)));
  shuffle_(job->indices, n_corners);
  job->n_old = n_old; job->n_new = n_new;
  job->n_desired = options_.triangulate_n_features - n_landmarks0;

  Frame& f1 = *frame1;
  {  // reserve space for features in the second frame (:86-90)
    const size_t need = f1.num_features_ + job->n_desired;
    if (need > f1.landmark_vec_.size() || 2 * need > f1.px_vec_.size()) {
      f1.px_vec_.resize(2 * need); f1.f_vec_.resize(3 * need); f1.grad_vec_.resize(2 * need); f1.score_vec_.resize(need);
      f1.level_vec_.resize(need); f1.type_vec_.resize(need, SVOH_FT_OUTLIER); f1.landmark_vec_.resize(need);
      f1.seed_ref_vec_.resize(need); f1.track_id_vec_.resize(need, -1); f1.invmu_sigma2_a_b_vec_.resize(4 * need);
    }
  }
  svoh::store_rigid(svoh::mul(f1.T_cam_imu_, f0.T_imu_cam_), job->T_f1_f0);   // frame1->T_cam_body_ * frame0->T_body_cam_
  job->v0 = view_of(f0); job->v1 = view_of(f1);
  return true;
}

void StereoTriangulationHip::finish(const FramePtr& frame0, const FramePtr& frame1, const Job& job, const int32_t* result, const double* depth, const double* px_cur,
                                    const double* f_cur, const double* A)
{
  // the reference's loop (:95-137)
  Frame &f0 = *frame0, &f1 = *frame1;
  const size_t n_old = job.n_old, n_new = job.n_new, n_desired = job.n_desired;
  last_indices_ = job.indices;
  last_results_.assign(n_new, -1);
  const Transformation T_world_cam0 = svoh::inverse(f0.T_f_w_);
  size_t n_succeeded = 0, n_failed = 0;
  for (const size_t i_ref : job.indices) {
    const size_t k = i_ref - n_old;
    last_results_[k] = result[k];
    if (result[k] == SVOH_MATCH_SUCCESS) {
      const svoh::Vec3 p_cam = { f0.f_vec_[3 * i_ref] * depth[k], f0.f_vec_[3 * i_ref + 1] * depth[k], f0.f_vec_[3 * i_ref + 2] * depth[k] };
      PointPtr new_point = std::make_shared<Point>();
      new_point->pos_ = svoh::transform(T_world_cam0, p_cam);
      new_point->id_ = next_point_id_++;
      f0.landmark_vec_[i_ref] = new_point;
      f0.track_id_vec_[i_ref] = new_point->id();
      new_point->obs_.push_back(Point::Obs{ frame0, i_ref });
      const size_t i_cur = f1.num_features_;
      f1.type_vec_[i_cur] = f0.type_vec_[i_ref];
      f1.level_vec_[i_cur] = f0.level_vec_[i_ref];
      f1.px_vec_[2 * i_cur] = px_cur[2 * k]; f1.px_vec_[2 * i_cur + 1] = px_cur[2 * k + 1];
      for (int j = 0; j < 3; ++j) f1.f_vec_[3 * i_cur + j] = f_cur[3 * k + j];
      f1.score_vec_[i_cur] = f0.score_vec_[i_ref];
      double g0 = A[4 * k] * f0.grad_vec_[2 * i_ref] + A[4 * k + 2] * f0.grad_vec_[2 * i_ref + 1];
      double g1 = A[4 * k + 1] * f0.grad_vec_[2 * i_ref] + A[4 * k + 3] * f0.grad_vec_[2 * i_ref + 1];
      const double z = g0 * g0 + g1 * g1;
      if (z > 0.0) { const double nn = sqrt(z); g0 /= nn; g1 /= nn; }   // .normalized()
      f1.grad_vec_[2 * i_cur] = g0; f1.grad_vec_[2 * i_cur + 1] = g1;
      f1.landmark_vec_[i_cur] = new_point;
      f1.track_id_vec_[i_cur] = new_point->id();
      new_point->obs_.push_back(Point::Obs{ frame1, i_cur });
      f1.num_features_++;
      ++n_succeeded;
    } else {
      ++n_failed;
    }
    if (n_succeeded >= n_desired) break;
  }
  last_n_succeeded_ = n_succeeded; last_n_failed_ = n_failed;
}

void StereoTriangulationHip::compute(const FramePtr& frame0, const FramePtr& frame1)
{
  last_indices_.clear(); last_results_.clear(); last_n_succeeded_ = last_n_failed_ = 0;
  if (!frame0 || !frame1) throw std::runtime_error("StereoTriangulationHip::compute: NULL frame");
  if (!wantsFeatures(*frame0)) return;

  // detect new features (the detector's grid holds what the caller marked; every cell may deliver one)
  std::vector<double> new_px, new_scores, new_grads;
  std::vector<int32_t> new_levels;
  std::vector<uint8_t> new_types;
  const size_t max_n_features = feature_detector_->grid_.size();
  feature_detector_->detect(frame0->pyramid, nullptr, 0, max_n_features, new_px, new_scores, new_levels, new_grads, new_types);
  Job job;
  if (!prepare(frame0, frame1, new_px, new_scores, new_levels, new_grads, new_types, &job)) return;

  // every candidate through Matcher::findEpipolarMatchDirect in one launch (finish's loop only reads the results)
  Frame& f0 = *frame0;
  const size_t n_old = job.n_old, n_new = job.n_new;
  const svoh_matcher_options mo = matcherOptions();
  std::vector<int32_t> ref_idx(n_new, 0);
  svoh_feature_batch fb{};
  fb.n = static_cast<int32_t>(n_new);
  fb.ref_frame_idx = ref_idx.data();
  fb.px = f0.px_vec_.data() + 2 * n_old; fb.f = f0.f_vec_.data() + 3 * n_old; fb.grad = f0.grad_vec_.data() + 2 * n_old;
  fb.level = f0.level_vec_.data() + n_old; fb.type = f0.type_vec_.data() + n_old;
  const double d_inv[3] = { options_.mean_depth_inv, options_.min_depth_inv, options_.max_depth_inv };
  std::vector<int32_t> result(n_new);
  std::vector<double> depth(n_new), px_cur(2 * n_new), f_cur(3 * n_new), A(4 * n_new);
  svoh_epipolar_match_outputs out{};
  out.result = result.data(); out.depth = depth.data(); out.px_cur = px_cur.data(); out.f_cur = f_cur.data();
  out.A_cur_ref = A.data();
  const int rc = svoh_epipolar_match_batch(ctx_, &mo, 1, &job.v0, &job.v1, &job.T_f1_f0, &fb, d_inv, nullptr, &out);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_epipolar_match_batch: ") + svoh_last_error_string(ctx_));
  finish(frame0, frame1, job, result.data(), depth.data(), px_cur.data(), f_cur.data(), A.data());
}

// ---- pose optimiser -------------------------------------------------------------------
PoseOptimizerHip::PoseOptimizerHip(svoh_ctx* ctx, SolverOptions solver_options) : ctx_(ctx), solver_options_(solver_options)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("PoseOptimizerHip: NULL svoh_ctx (no CPU fallback exists)");
}

void PoseOptimizerHip::setRotationPrior(const svoh::Quat& R_frame_world, double lambda)
{
  R_prior_ = R_frame_world;
  prior_lambda_ = lambda;
  have_prior_ = true;
}

size_t PoseOptimizerHip::run(const FrameBundle::Ptr& frame_bundle, double reproj_thresh_px) { return run(frame_bundle, reproj_thresh_px, nullptr); }

namespace {
struct PoseOutlierThresholds { double uplane, bearing_diff; };
// (the reference's two function-local statics: whatever camera and threshold get here first fix them for the process)
const PoseOutlierThresholds& poseOutlierThresholds(const svoh_camera& cam, double reproj_thresh_px)
{
  static const PoseOutlierThresholds t = { reproj_thresh_px / std::fabs(cam.fx),
                                           std::fabs(2 * std::sin(0.5 * (std::atan(reproj_thresh_px / (2.0 * cam.fx)) + std::atan(reproj_thresh_px / (2.0 * cam.fy))))) };
  return t;
}
}  // namespace

void fixProcessWideThresholds(const svoh_camera& cam, double reproj_thresh_px)
{
  (void)poseOutlierThresholds(cam, reproj_thresh_px);
  Frame f;
  f.cam = cam;
  (void)updateSeedPxErrorAngle(f);
}

void PoseOptimizerHip::prepareRun(const FrameBundle::Ptr& frame_bundle, double reproj_thresh_px, svoh_pose_options& o, svoh_pose_problem& pb)
{
  if (!frame_bundle || frame_bundle->empty()) throw std::runtime_error("PoseOptimizer: FrameBundle is empty");   // CHECK
  const size_t nc = frame_bundle->size();
  if (nc > SVOH_MAX_CAMS) throw std::runtime_error("PoseOptimizerHip: too many cameras in the bundle");
  const Frame& f0 = *frame_bundle->at(0);
  // removeOutliers' thresholds are function-local statics: the first bundle ever optimised fixes them
  // (pose_optimizer.cpp:211-212)
  const double threshold_uplane = poseOutlierThresholds(f0.cam, reproj_thresh_px).uplane;
  const double threshold_bearing_diff = poseOutlierThresholds(f0.cam, reproj_thresh_px).bearing_diff;
  o = svoh_pose_options{};
  o.max_iter = static_cast<int32_t>(solver_options_.max_iter);
  o.eps = solver_options_.eps;
  o.error_type = err_type_ == ErrorType::kUnitPlane ? SVOH_POSE_ERR_UNIT_PLANE
                 : err_type_ == ErrorType::kBearingVectorDiff ? SVOH_POSE_ERR_BEARING_DIFF : SVOH_POSE_ERR_IMAGE_PLANE;
  o.outlier_threshold = err_type_ == ErrorType::kUnitPlane ? threshold_uplane
                        : err_type_ == ErrorType::kBearingVectorDiff ? threshold_bearing_diff : reproj_thresh_px;
  o.have_rotation_prior = have_prior_;
  o.prior_lambda = prior_lambda_;
  o.R_prior[0] = R_prior_.w; o.R_prior[1] = R_prior_.x; o.R_prior[2] = R_prior_.y; o.R_prior[3] = R_prior_.z;

  pb = svoh_pose_problem{};
  pb.n_cams = static_cast<int32_t>(nc);
  svoh::store_rigid(f0.T_imu_world(), pb.T_imu_world);
  std::vector<std::vector<double>>& xyz = run_xyz_;
  std::vector<std::vector<uint8_t>>&usable = run_usable_, &outlier = run_outlier_;
  xyz.resize(nc); usable.resize(nc); outlier.resize(nc);
  size_t n_features = 0;
  for (size_t c = 0; c < nc; ++c) {
    Frame& fr = *frame_bundle->at(c);
    const size_t n = fr.num_features_;
    n_features += n;
    xyz[c].assign(3 * n, 0.0); usable[c].assign(n, 0); outlier[c].assign(n, 0);
    fr.landmark_vec_.resize(n); fr.seed_ref_vec_.resize(n);
    for (size_t i = 0; i < n; ++i) {
      // evaluateErrorImpl (pose_optimizer.cpp:128-139): landmark position, or the seed's position in the world
      svoh::Vec3 p{ 0, 0, 0 };
      const uint8_t t = fr.type_vec_[i];
      if (fr.landmark_vec_[i]) p = fr.landmark_vec_[i]->pos_;
      else if ((t == SVOH_FT_CORNER_SEED || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_CORNER_SEED_CONVERGED ||
                t == SVOH_FT_EDGELET_SEED_CONVERGED) && fr.seed_ref_vec_[i].keyframe) {
        const Frame& kf = *fr.seed_ref_vec_[i].keyframe;
        const size_t k = static_cast<size_t>(fr.seed_ref_vec_[i].seed_id);
        const double depth = kf.getSeedDepth(k);
        const svoh::Vec3 in_kf{ kf.f_vec_[3 * k] * depth, kf.f_vec_[3 * k + 1] * depth, kf.f_vec_[3 * k + 2] * depth };
        p = svoh::transform(svoh::inverse(kf.T_f_w_), in_kf);   // T_world_cam() * getSeedPosInFrame
      } else continue;
      usable[c][i] = 1;
      xyz[c][3 * i] = p.x; xyz[c][3 * i + 1] = p.y; xyz[c][3 * i + 2] = p.z;
    }
    svoh_pose_camera& pc = pb.cams[c];
    pc.cam = fr.cam;
    svoh::store_rigid(fr.T_cam_imu(), pc.T_cam_imu);
    pc.n_features = static_cast<int32_t>(n);
    pc.px = fr.px_vec_.data(); pc.f = fr.f_vec_.data(); pc.grad = fr.grad_vec_.data(); pc.level = fr.level_vec_.data();
    pc.type = fr.type_vec_.data(); pc.xyz_world = xyz[c].data(); pc.usable = usable[c].data(); pc.outlier = outlier[c].data();
  }
  if (n_features == 0) throw std::runtime_error("PoseOptimizer: No features in frames");   // CHECK_GT
}

size_t PoseOptimizerHip::finishRun(const FrameBundle::Ptr& frame_bundle, const svoh_pose_result& result)
{
  last_ = result;
  const size_t nc = frame_bundle->size();
  const Frame& f0 = *frame_bundle->at(0);
  measurement_sigma_ = last_.measurement_sigma;
  const Transformation T_imu_world = svoh::load_rigid(last_.T_imu_world);
  for (size_t c = 0; c < nc; ++c) {
    Frame& fr = *frame_bundle->at(c);
    fr.T_f_w_ = svoh::mul(fr.T_cam_imu(), T_imu_world);
    for (size_t i = 0; i < fr.num_features_; ++i)
      if (run_outlier_[c][i]) {
        fr.type_vec_[i] = SVOH_FT_OUTLIER;
        fr.seed_ref_vec_[i].keyframe.reset();
        fr.landmark_vec_[i] = nullptr;
      }
  }
  const double error_scale = err_type_ == ErrorType::kUnitPlane ? std::fabs(f0.cam.fx) : 1.0;   // focal_length_
  stats_.reproj_error_before = last_.reproj_error_before * error_scale;
  stats_.reproj_error_after = last_.reproj_error_after * error_scale;
  return static_cast<size_t>(last_.n_meas - last_.n_deleted_edges - last_.n_deleted_corners);
}

size_t PoseOptimizerHip::run(const FrameBundle::Ptr& frame_bundle, double reproj_thresh_px, const std::function<void()>& after_launch)
{
  svoh_pose_options o;
  svoh_pose_problem pb;
  prepareRun(frame_bundle, reproj_thresh_px, o, pb);
  svoh_pose_result res{};
  int rc;
  if (after_launch) {
    // (an exception cannot cross the C boundary: it is carried over it)
    struct Hook { const std::function<void()>* fn; std::exception_ptr error; } hook{ &after_launch, nullptr };
    rc = svoh_optimize_pose_batch_hook(ctx_, &o, 1, &pb, &res, [](void* user) {
      Hook* h = static_cast<Hook*>(user);
      try { (*h->fn)(); } catch (...) { h->error = std::current_exception(); }
    }, &hook);
    if (hook.error) std::rethrow_exception(hook.error);
  } else {
    rc = svoh_optimize_pose_batch(ctx_, &o, 1, &pb, &res);
  }
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_optimize_pose_batch: ") + svoh_last_error_string(ctx_));
  return finishRun(frame_bundle, res);
}

void resolveAlignmentPoints(Frame& frame) { resolveAlignmentPoints(frame, nullptr); }

void resolveAlignmentPoints(Frame& frame, const std::function<int32_t(const Frame& keyframe, size_t seed_id)>& unit_of)
{
  const size_t n = frame.num_features_;
  frame.pos_world_.assign(3 * n, 0.0);
  frame.alignable_.assign(n, 0);
  frame.pos_seed_unit_.clear();
  if (unit_of) frame.pos_seed_unit_.assign(n, -1);
  for (size_t i = 0; i < n; ++i) {
    const uint8_t t = frame.type_vec_[i];
    if (t == SVOH_FT_MAPPOINT || t == SVOH_FT_MAPPOINT_SEED || t == SVOH_FT_MAPPOINT_SEED_CONVERGED) continue;
    svoh::Vec3 p;
    if (i < frame.landmark_vec_.size() && frame.landmark_vec_[i]) p = frame.landmark_vec_[i]->pos_;
    else if (i < frame.seed_ref_vec_.size() && frame.seed_ref_vec_[i].keyframe) {
      const Frame& kf = *frame.seed_ref_vec_[i].keyframe;
      const size_t k = static_cast<size_t>(frame.seed_ref_vec_[i].seed_id);
      const double depth = kf.getSeedDepth(k);
      p = svoh::transform(svoh::inverse(kf.T_f_w_),
                          svoh::Vec3{ kf.f_vec_[3 * k] * depth, kf.f_vec_[3 * k + 1] * depth, kf.f_vec_[3 * k + 2] * depth });
      if (unit_of) frame.pos_seed_unit_[i] = unit_of(kf, k);
    } else continue;
    frame.alignable_[i] = 1;
    frame.pos_world_[3 * i] = p.x; frame.pos_world_[3 * i + 1] = p.y; frame.pos_world_[3 * i + 2] = p.z;
  }
}

namespace depth_filter_utils {
void initializeSeeds(const FramePtr& frame, DetectorHip& feature_detector, size_t max_n_seeds, float depth_min, float /*depth_max*/,
                     float depth_mean)
{
  const int max_n_features = static_cast<int>(max_n_seeds) - static_cast<int>(frame->num_features_);
  if (max_n_features <= 0) return;   // "Skip seed initialization. Have already enough features."
  std::vector<double> px, score, grad;
  std::vector<int32_t> level;
  std::vector<uint8_t> type;
  feature_detector.detect(frame->pyramid, nullptr, 0, static_cast<size_t>(max_n_features), px, score, level, grad, type);
  appendSeeds(frame, px, score, level, grad, type, depth_min, depth_mean);
}

void appendSeeds(const FramePtr& frame, const std::vector<double>& px, const std::vector<double>& score, const std::vector<int32_t>& level,
                 const std::vector<double>& grad, const std::vector<uint8_t>& type, float depth_min, float depth_mean)
{
  const size_t n_old = frame->num_features_;
  const size_t n_new = level.size();
  const size_t n = n_old + n_new;
  frame->px_vec_.resize(2 * n_old); frame->px_vec_.insert(frame->px_vec_.end(), px.begin(), px.end());
  frame->grad_vec_.resize(2 * n_old); frame->grad_vec_.insert(frame->grad_vec_.end(), grad.begin(), grad.end());
  frame->score_vec_.resize(n_old); frame->score_vec_.insert(frame->score_vec_.end(), score.begin(), score.end());
  frame->level_vec_.resize(n_old); frame->level_vec_.insert(frame->level_vec_.end(), level.begin(), level.end());
  frame->type_vec_.resize(n_old);
  for (size_t i = 0; i < n_new; ++i) {
    if (type[i] == SVOH_FT_CORNER) frame->type_vec_.push_back(SVOH_FT_CORNER_SEED);
    else if (type[i] == SVOH_FT_EDGELET) frame->type_vec_.push_back(SVOH_FT_EDGELET_SEED);
    else throw std::runtime_error("initializeSeeds: unknown feature type");   // LOG(FATAL)
  }
  frame->f_vec_.resize(3 * n);
  const svoh::CamModel cm = svoh::load_camera(frame->cam);
  for (size_t i = n_old; i < n; ++i) {
    const svoh::Vec3 f = svoh::back_project3(cm, frame->px_vec_[2 * i], frame->px_vec_[2 * i + 1]);
    const double nn = sqrt(f.x * f.x + f.y * f.y + f.z * f.z);
    frame->f_vec_[3 * i] = f.x / nn; frame->f_vec_[3 * i + 1] = f.y / nn; frame->f_vec_[3 * i + 2] = f.z / nn;
  }
  frame->landmark_vec_.resize(n); frame->seed_ref_vec_.resize(n); frame->track_id_vec_.resize(n, -1);
  frame->num_features_ = n;
  frame->seed_mu_range_ = 1.0 / depth_min;                               // getMeanRangeFromDepthMinMax
  frame->invmu_sigma2_a_b_vec_.resize(4 * n);
  for (size_t i = n_old; i < n; ++i) {
    frame->invmu_sigma2_a_b_vec_[4 * i] = 1.0 / depth_mean;              // getMeanFromDepth
    frame->invmu_sigma2_a_b_vec_[4 * i + 1] = frame->seed_mu_range_ * frame->seed_mu_range_ / 36.0;   // getInitSigma2FromMuRange
    frame->invmu_sigma2_a_b_vec_[4 * i + 2] = 10.0;
    frame->invmu_sigma2_a_b_vec_[4 * i + 3] = 10.0;
  }
}
}  // namespace depth_filter_utils

double updateSeedPxErrorAngle(const Frame& cur_frame)
{
  // static double px_error_angle = cur_frame.getAngleError(1.0);  (depth_filter.cpp:383-384,
  // camera_geometry_base.hpp: atan(1/(2 fx)) + atan(1/(2 fy)))
  static const double px_error_angle = atan(1.0 / (2.0 * cur_frame.cam.fx)) + atan(1.0 / (2.0 * cur_frame.cam.fy));
  return px_error_angle;
}

// ---- reprojector ------------------------------------------------------------------
bool Point::getCloseViewObs(const svoh::Vec3& framepos, FramePtr& ref_frame, size_t& ref_feature_index) const
{
  double min_cos_angle = 0.0;
  svoh::Vec3 obs_dir{ framepos.x - pos_.x, framepos.y - pos_.y, framepos.z - pos_.z };
  {
    const double n = sqrt(obs_dir.x * obs_dir.x + obs_dir.y * obs_dir.y + obs_dir.z * obs_dir.z);
    if (n > 0.0) { obs_dir.x /= n; obs_dir.y /= n; obs_dir.z /= n; }
  }
  for (const Obs& obs : obs_) {
    FramePtr frame = obs.frame.lock();
    if (!frame) return false;
    const svoh::Vec3 fp = frame->pos();
    svoh::Vec3 dir{ fp.x - pos_.x, fp.y - pos_.y, fp.z - pos_.z };
    const double n = sqrt(dir.x * dir.x + dir.y * dir.y + dir.z * dir.z);
    if (n > 0.0) { dir.x /= n; dir.y /= n; dir.z /= n; }
    const double cos_angle = obs_dir.x * dir.x + obs_dir.y * dir.y + obs_dir.z * dir.z;
    if (cos_angle > min_cos_angle) {
      min_cos_angle = cos_angle;
      ref_frame = frame;
      ref_feature_index = obs.keypoint_index_;
    }
  }
  return !(min_cos_angle < 0.4);  // observations more than 60 degrees away are useless
}

int OccupandyGrid2D::getNCell(int n_pixels, int size)
{
  return static_cast<int>(std::ceil(static_cast<double>(n_pixels) / static_cast<double>(size)));
}
int OccupandyGrid2D::numOccupied() const { return static_cast<int>(std::count(occupancy_.begin(), occupancy_.end(), true)); }
size_t OccupandyGrid2D::getCellIndex(int x, int y, int scale) const
{
  // getCellIndex(Eigen::Vector2d(scale * x, scale * y)) (occupancy_grid_2d.h:82-94)
  const double px = scale * x, py = scale * y;
  return static_cast<size_t>(std::floor(py / cell_size) * n_cols + std::floor(px / cell_size));
}

// ---- Frame helpers the reprojector needs ----
static bool is_corner_edgelet_seed(uint8_t t)
{
  return t == SVOH_FT_CORNER_SEED || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_CORNER_SEED_CONVERGED || t == SVOH_FT_EDGELET_SEED_CONVERGED;
}
static bool is_map_point(uint8_t t) { return t == SVOH_FT_MAPPOINT || t == SVOH_FT_MAPPOINT_SEED || t == SVOH_FT_MAPPOINT_SEED_CONVERGED; }

size_t Frame::numTrackedFeatures() const
{
  size_t count = 0;
  for (size_t i = 0; i < num_features_; ++i) {
    const uint8_t t = type_vec_[i];
    const bool valid_landmark = i < landmark_vec_.size() && landmark_vec_[i] != nullptr;
    if ((valid_landmark && t != SVOH_FT_FIXED_LANDMARK && !is_map_point(t)) || is_corner_edgelet_seed(t)) ++count;
  }
  return count;
}

bool Frame::isVisible(const svoh::Vec3& xyz_w, double* px) const
{
  const svoh::Vec3 xyz_f = svoh::transform(T_f_w_, xyz_w);
  const svoh::CamModel cm = svoh::load_camera(cam);
  {   // pinhole: not farther off the optical axis than the image's top-left corner (frame.cpp:233-246)
    if (!min_cos_valid_ || memcmp(&min_cos_cam_, &cam, sizeof cam) != 0) {
      const svoh::Vec3 f_tl = svoh::back_project3(cm, 0.0, 0.0);
      const double n = sqrt(f_tl.x * f_tl.x + f_tl.y * f_tl.y + f_tl.z * f_tl.z);
      min_cos_ = f_tl.z / n;
      min_cos_cam_ = cam;
      min_cos_valid_ = true;
    }
    const double min_cos = min_cos_;
    const double nf = sqrt(xyz_f.x * xyz_f.x + xyz_f.y * xyz_f.y + xyz_f.z * xyz_f.z);
    const double cur_cos = xyz_f.z / nf;
    if (cur_cos < min_cos) return false;
  }
  // cam_->project3(xyz_f, &px).isKeypointVisible(): inside the image box (camera_geometry.hpp:29-38)
  double u, v;
  svoh::project3(cm, xyz_f, u, v);
  if (px) { px[0] = u; px[1] = v; }
  return u >= 0.0 && v >= 0.0 && u < static_cast<double>(cam.width) && v < static_cast<double>(cam.height);
}

// ---- Reprojector::reprojectFrames (reprojector.cpp:27-306) ----
ReprojectorHip::ReprojectorHip(svoh_ctx* ctx, const ReprojectorOptions& options, size_t camera_index)
    : options_(options), ctx_(ctx), camera_index_(camera_index)
{
  requireMatchingAbi();
  if (!ctx_) throw std::runtime_error("ReprojectorHip: NULL svoh_ctx (no CPU fallback exists)");
  if (camera_index_ >= SVOH_MAX_CAMS) throw std::runtime_error("ReprojectorHip: camera index out of range");
  device_select_ = getenv("SVOH_REPROJ_DEVICE_SELECT") != nullptr && atoi(getenv("SVOH_REPROJ_DEVICE_SELECT")) != 0;
}

namespace {
// per thread (one camera stream = one host thread in tools/svoh_mini_frontend): nothing is shared between streams
struct ReprojTiming {   // SVOH_REPROJ_TIMING=1: host / device split of reprojectFrames (MEDIANS per call, steady state), printed when the thread ends
  bool on = getenv("SVOH_REPROJ_TIMING") != nullptr;
  long skip = 5;        // the first calls pay one-time costs (code objects, first allocations): not part of the statistics
  // accumulators of the call in progress; closed into the sample lists by end_call()
  double t[6] = { 0, 0, 0, 0, 0, 0 };
  double kernel_ms = 0;
  double rt[4] = { 0, 0, 0, 0 };   // device round trip: direct batch call, seed batch call, collect, the rest
  std::vector<double> samples[11];
  long n = 0, n_direct = 0, n_seeds = 0, n_reached3 = 0, n_spec3 = 0;
  static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  void end_call()
  {
    if (skip > 0) { --skip; n = n_direct = n_seeds = n_reached3 = n_spec3 = 0; }
    else {
      for (int k = 0; k < 6; ++k) samples[k].push_back(t[k]);
      samples[6].push_back(kernel_ms);
      for (int k = 0; k < 3; ++k) samples[7 + k].push_back(rt[k]);
      samples[10].push_back(t[0] + t[1] + t[2] + t[3] + t[4] + t[5]);
    }
    for (double& v : t) v = 0;
    for (double& v : rt) v = 0;
    kernel_ms = 0;
  }
  static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
  ~ReprojTiming()
  {
    if (on && n)
      fprintf(stderr, "[reproject] median per call (ms): whole call %.3f = candidates %.3f, sort %.3f, plan %.3f, device round trip(s) %.3f (kernel %.3f), other (replay, grid) %.3f [round trip: stage direct %.3f, stage seeds %.3f, launch + wait %.3f] (%ld calls; per call %.0f direct + %.0f seed units speculated; unconverged pass reached %ld x, speculated %ld x)\n",
              med(samples[10]), med(samples[0]), med(samples[1]), med(samples[2]), med(samples[3]), med(samples[6]), med(samples[5]), med(samples[7]), med(samples[8]), med(samples[9]),
              n, (double)n_direct / n, (double)n_seeds / n, n_reached3, n_spec3);
  }
};
thread_local ReprojTiming g_reproj_timing;
}  // namespace

static bool same_pose(const Transformation& a, const Transformation& b)
{
  return a.q.w == b.q.w && a.q.x == b.q.x && a.q.y == b.q.y && a.q.z == b.q.z && a.t.x == b.t.x && a.t.y == b.t.y && a.t.z == b.t.z;
}

ReprojectorHip::~ReprojectorHip() = default;

void ReprojectorHip::countCandidateProjection(const std::vector<FramePtr>& kfs, size_t* n_points, size_t* n_kf) const
{
  size_t np = 0, nk = 0;
  for (const FramePtr& kf : kfs) {
    if (!kf || kf->num_features_ == 0) continue;
    np += kf->num_features_;
    ++nk;
  }
  if (n_points) *n_points = np;
  if (n_kf) *n_kf = nk;
}

void ReprojectorHip::gatherCandidateProjection(const FramePtr& cur_frame, const std::vector<FramePtr>& kfs, const ProjectionArrays& into,
                                               const std::function<int32_t(const Frame& keyframe, size_t seed_id)>& unit_of)
{
  if (unit_of && !(into.ranges && into.mu_unit)) throw std::runtime_error("ReprojectorHip::gatherCandidateProjection: seed units need the ranges form with mu_unit");
  proj_kf_off_.clear();
  size_t at = 0;
  int32_t k = 0;
  for (const FramePtr& kf : kfs) {
    if (!kf || kf->num_features_ == 0) continue;
    svoh::store_rigid(svoh::inverse(kf->T_f_w_), into.T_world_kf[k]);
    proj_kf_off_.push_back(ProjKf{ kf.get(), kf->id_, at, kf->num_features_, kf->T_f_w_ });
    if (into.ranges) {
      if (!kf->features) throw std::runtime_error("ReprojectorHip::gatherCandidateProjection: a keyframe without resident feature columns in the ranges form");
      into.ranges[k] = svoh_candidate_range{ kf->features, into.point_offset + static_cast<int32_t>(at), static_cast<int32_t>(kf->num_features_), into.job, 0 };
    }
    for (size_t i = 0; i < kf->num_features_; ++i, ++at) {
      const PointPtr& lm = i < kf->landmark_vec_.size() ? kf->landmark_vec_[i] : PointPtr();
      double* v = into.v + 3 * at;
      if (lm) {                                       // getCandidate: the landmark's position ...
        const svoh::Vec3 p = lm->pos();
        into.kind[at] = 0; v[0] = p.x; v[1] = p.y; v[2] = p.z; into.mu[at] = 1.0;
      } else {                                        // ... or T_world_cam() * getSeedPosInFrame(i)
        into.kind[at] = 1;
        if (!into.ranges) { v[0] = kf->f_vec_[3 * i]; v[1] = kf->f_vec_[3 * i + 1]; v[2] = kf->f_vec_[3 * i + 2]; }   // (the ranges form reads the resident f column)
        into.mu[at] = 4 * i < kf->invmu_sigma2_a_b_vec_.size() ? kf->invmu_sigma2_a_b_vec_[4 * i] : 1.0;
        if (unit_of) into.mu_unit[at] = unit_of(*kf, i);   // (>= 0: the device reads the update's result; mu above is the state before it)
      }
      if (!into.ranges) into.kf[at] = k;
    }
    ++k;
  }
  proj_n_points_ = at; proj_n_kf_ = static_cast<size_t>(k);
  proj_kind_p_ = into.kind; proj_v_p_ = into.v; proj_mu_p_ = into.mu; proj_unit_p_ = unit_of ? into.mu_unit : nullptr;
  proj_px_p_ = nullptr; proj_visible_p_ = nullptr;
  proj_frame_ = at ? cur_frame.get() : nullptr;
  proj_frame_id_ = cur_frame->id_;
  proj_collected_ = false;
}

void ReprojectorHip::adoptCandidateProjection(const FramePtr& cur_frame, const double* px, const uint8_t* visible)
{
  if (proj_frame_ != cur_frame.get() || proj_frame_id_ != cur_frame->id_) throw std::runtime_error("ReprojectorHip::adoptCandidateProjection: not the frame whose projection was gathered");
  proj_px_p_ = px; proj_visible_p_ = visible;
  proj_collected_ = true;
}

void ReprojectorHip::enqueueCandidateProjection(const FramePtr& cur_frame, const std::vector<FramePtr>& kfs, const Transformation* T_iref_world,
                                                int align_result_index)
{
  discardCandidateProjection();
  size_t n_points = 0, n_kf = 0;
  countCandidateProjection(kfs, &n_points, &n_kf);
  proj_kind_.resize(n_points); proj_kf_.resize(n_points); proj_v_.resize(3 * n_points); proj_mu_.resize(n_points); proj_T_world_kf_.resize(n_kf);
  gatherCandidateProjection(cur_frame, kfs, ProjectionArrays{ proj_T_world_kf_.data(), proj_kind_.data(), proj_kf_.data(), proj_v_.data(), proj_mu_.data() });
  const int n = static_cast<int>(n_points);
  if (n == 0) return;
  proj_frame_ = nullptr;   // (set again below, once the call is queued)
  svoh_se3 Ta, Tb;
  int rc;
  if (align_result_index >= 0 && T_iref_world) {
    svoh::store_rigid(cur_frame->T_cam_imu(), Ta);
    svoh::store_rigid(*T_iref_world, Tb);
    rc = svoh_project_candidates_enqueue(ctx_, &cur_frame->cam, &Ta, &Tb, align_result_index, static_cast<int>(proj_T_world_kf_.size()),
                                         proj_T_world_kf_.data(), n, proj_kind_.data(), proj_kf_.data(), proj_v_.data(), proj_mu_.data());
  } else {
    svoh::store_rigid(cur_frame->T_f_w_, Ta);
    rc = svoh_project_candidates_enqueue(ctx_, &cur_frame->cam, &Ta, nullptr, -1, static_cast<int>(proj_T_world_kf_.size()),
                                         proj_T_world_kf_.data(), n, proj_kind_.data(), proj_kf_.data(), proj_v_.data(), proj_mu_.data());
  }
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_project_candidates_enqueue: ") + svoh_last_error_string(ctx_));
  proj_frame_ = cur_frame.get();
  proj_frame_id_ = cur_frame->id_;
  proj_collected_ = false;
}

void ReprojectorHip::discardCandidateProjection()
{
  if (proj_frame_ && !proj_collected_ && proj_kind_p_ == proj_kind_.data()) {   // the queued call's results are dropped, the context is free for the next one
    proj_px_.resize(2 * proj_n_points_); proj_visible_.resize(proj_n_points_);
    (void)svoh_project_candidates_collect(ctx_, static_cast<int>(proj_n_points_), proj_px_.data(), proj_visible_.data());
  }
  proj_frame_ = nullptr;
  proj_collected_ = false;
}

void ReprojectorHip::walkCandidates(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs, std::vector<PointPtr>& trash_points)
{
  walkLists(cur_frame, visible_kfs, trash_points, 3);
}

void ReprojectorHip::walkCandidatesWithoutUnconverged(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs, std::vector<PointPtr>& trash_points)
{
  walkLists(cur_frame, visible_kfs, trash_points, 1);
}

// which: 1 = grid / statistics reset, landmarks (with their trash-list side effects) and converged seeds; 2 = the unconverged seeds
// only, appended to what a walk with 1 has left (same keyframes, same frame: the list is what one walk with 3 gives -- each list
// comes out in the order of its own loop of the reference either way, and nothing a landmark or converged-seed pass does changes
// a seed's inverse depth or type)
void ReprojectorHip::walkLists(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs, std::vector<PointPtr>& trash_points, int which)
{
  if (which == 2) {
    if (!unconverged_pending_) throw std::runtime_error("ReprojectorHip: no walk is waiting for its unconverged seeds");
    unconverged_pending_ = false;
    const bool have_proj = have_proj_ && proj_frame_ == cur_frame.get() && proj_frame_id_ == cur_frame->id_ && proj_collected_;
    std::vector<reprojector::Candidate>& unconverged = unconverged_;
    unconverged.clear();
    if (options_.reproject_unconverged_seeds)
      for (const FramePtr& ref_frame : visible_kfs) {
        const svoh::Rigid T_world_ref = svoh::inverse(ref_frame->T_f_w_);
        long proj_off = -1;
        size_t proj_n = 0;
        if (have_proj)
          for (const ProjKf& ko : proj_kf_off_)
            if (ko.frame == ref_frame.get() && ko.id == ref_frame->id_ && same_pose(ko.T_f_w, ref_frame->T_f_w_)) {
              proj_off = static_cast<long>(ko.offset); proj_n = ko.n_features; break;
            }
        for (size_t i = 0; i < ref_frame->num_features_; ++i) {
          const uint8_t type = ref_frame->type_vec_[i];
          if (type == SVOH_FT_CORNER_SEED || type == SVOH_FT_EDGELET_SEED) emitCandidate(cur_frame, ref_frame, T_world_ref, proj_off, proj_n, i, unconverged);
        }
      }
    if (have_proj) { proj_frame_ = nullptr; proj_collected_ = false; }
    return;
  }
  // device projection queued / adopted for this frame: pointer AND id (a frame destroyed with its projection still
  // queued may be followed by a new one at the same address)
  const bool have_proj = proj_frame_ == cur_frame.get() && proj_frame_id_ == cur_frame->id_ && proj_collected_;
  have_proj_ = have_proj;
  unconverged_pending_ = which == 1;
  if (options_.max_n_features_per_frame == 0) throw std::runtime_error("Reprojector: max_n_features_per_frame must be > 0");   // CHECK_GT
  if (!grid_)
    grid_.reset(new OccupandyGrid2D(static_cast<int>(options_.cell_size),
                                    OccupandyGrid2D::getNCell(cur_frame->cam.width, static_cast<int>(options_.cell_size)),
                                    OccupandyGrid2D::getNCell(cur_frame->cam.height, static_cast<int>(options_.cell_size))));
  grid_->reset();
  stats_ = reprojector::Statistics();

  // landmarks of the closest keyframes with overlap (:131-176), converged seeds (:201-241) and unconverged seeds
  // (:243-306): the three candidate lists depend on the visible keyframes and the current pose only, so one walk over
  // the keyframes' features gathers all of them (each list in the order its own loop of the reference gives), and they
  // are matched together
  candidates_.clear();
  std::vector<reprojector::Candidate>&converged = converged_, &unconverged = unconverged_;
  converged.clear(); unconverged.clear();
  for (const FramePtr& ref_frame : visible_kfs) {
    const svoh::Rigid T_world_ref = svoh::inverse(ref_frame->T_f_w_);
    // this keyframe's slice of the device projection, if it was part of it
    // ... and only as far as it still describes the keyframe: the projection took a snapshot of the keyframe's pose, of
    // the landmark positions and of the seeds' inverse depths when it was queued.  A keyframe whose pose has moved
    // since, a feature appended since, a landmark or seed that has changed since is computed here instead.
    long proj_off = -1;
    size_t proj_n = 0;
    if (have_proj)
      for (const ProjKf& ko : proj_kf_off_)
        if (ko.frame == ref_frame.get() && ko.id == ref_frame->id_ && same_pose(ko.T_f_w, ref_frame->T_f_w_)) {
          proj_off = static_cast<long>(ko.offset); proj_n = ko.n_features; break;
        }
    const size_t n_lm = ref_frame->landmark_vec_.size();
    for (size_t i = 0; i < ref_frame->num_features_; ++i) {
      const uint8_t type = ref_frame->type_vec_[i];
      if (i < n_lm && ref_frame->landmark_vec_[i] && type != SVOH_FT_OUTLIER && !is_map_point(type) && type != SVOH_FT_FIXED_LANDMARK) {
        const PointPtr& point = ref_frame->landmark_vec_[i];
        if (point->n_failed_reproj_ > 10) trash_points.push_back(point);
        else if (point->last_projected_kf_id_.at(camera_index_) != cur_frame->id_) {   // project a point only once
          point->last_projected_kf_id_[camera_index_] = cur_frame->id_;
          if (point->obs_.size() < 2 && options_.remove_unconstrained_points) trash_points.push_back(point);
          else emitCandidate(cur_frame, ref_frame, T_world_ref, proj_off, proj_n, i, candidates_);
        }
      }
      const bool conv = type == SVOH_FT_CORNER_SEED_CONVERGED || type == SVOH_FT_EDGELET_SEED_CONVERGED;
      const bool unconv = (type == SVOH_FT_CORNER_SEED || type == SVOH_FT_EDGELET_SEED) && options_.reproject_unconverged_seeds && (which & 2);
      if (conv || unconv) emitCandidate(cur_frame, ref_frame, T_world_ref, proj_off, proj_n, i, conv ? converged : unconverged);
    }
  }
  // the projection has been consumed (unless the unconverged seeds may still ask for it)
  if (have_proj && (which & 2)) { proj_frame_ = nullptr; proj_collected_ = false; }
}

// getCandidate for feature i of ref_frame, appended to `list` when it is visible.  The candidate is made in place (one reference
// to the keyframe taken per candidate: the walk visits thousands of features per frame and is on the frame's critical path)
void ReprojectorHip::emitCandidate(const FramePtr& cur_frame, const FramePtr& ref_frame, const svoh::Rigid& T_world_ref, long proj_off, size_t proj_n, size_t i,
                                   std::vector<reprojector::Candidate>& list)
{
  if (proj_off >= 0 && i < proj_n) {
    const size_t at = static_cast<size_t>(proj_off) + i;
    const Point* lm = i < ref_frame->landmark_vec_.size() ? ref_frame->landmark_vec_[i].get() : nullptr;
    bool fresh;
    if (lm) { const svoh::Vec3 p = lm->pos(); fresh = proj_kind_p_[at] == 0 && p.x == proj_v_p_[3 * at] && p.y == proj_v_p_[3 * at + 1] && p.z == proj_v_p_[3 * at + 2]; }
    else fresh = proj_kind_p_[at] == 1 && ((proj_unit_p_ && proj_unit_p_[at] >= 0) ||   // (read on the device from the update the driver has finished since)
                                           (4 * i < ref_frame->invmu_sigma2_a_b_vec_.size() ? ref_frame->invmu_sigma2_a_b_vec_[4 * i] : 1.0) == proj_mu_p_[at]);
    if (fresh) {
      if (!proj_visible_p_[at]) return;
      list.emplace_back();
      reprojector::Candidate& candidate = list.back();
      candidate.ref_frame = ref_frame; candidate.ref_index = i;
      candidate.cur_px[0] = proj_px_p_[2 * at]; candidate.cur_px[1] = proj_px_p_[2 * at + 1];
      candidate.n_reproj = lm ? lm->n_succeeded_reproj_ - lm->n_failed_reproj_ : 0;
      candidate.score = i < ref_frame->score_vec_.size() ? ref_frame->score_vec_[i] : 0.0;
      candidate.type = ref_frame->type_vec_[i];
      candidate.n_obs = lm ? lm->obs_.size() : 0u;
      return;
    }
  }
  reprojector::Candidate candidate;
  if (reprojector_utils::getCandidate(cur_frame, ref_frame, i, candidate, &T_world_ref)) list.push_back(std::move(candidate));
}

void ReprojectorHip::planMatches(const FramePtr& cur_frame, int n_speculated, bool resident_features)
{
  if (!sm_) sm_.reset(new detail::SpeculativeMatches);   // keeps its buffers from frame to frame
  sm_->clear();
  // resident_features: every unit names its feature by index into its reference frame's device columns (Frame::features); for
  // drivers that stage the batches themselves (FrontendLockstep) -- enqueue() below sends explicit columns
  sm_->setResident(resident_features);
  n_speculated_ = n_speculated < 0 ? 0 : (n_speculated > 3 ? 3 : n_speculated);
  std::vector<reprojector::Candidate>* lists[3] = { &candidates_, &converged_, &unconverged_ };
  plan_rs_.resize(3);
  for (int k = 0; k < 3; ++k) plan_rs_[k].clear();
  for (int k = 0; k < n_speculated_; ++k) plan_rs_[k] = sm_->plan(cur_frame, *lists[k]);
}

void ReprojectorHip::sortCandidateLists()
{
  // sortCandidatesByReprojStats of the three lists (reprojector.cpp:188, 224, 263) while the device works: what a
  // candidate is matched against does not depend on its place in the list, only the replay's visiting order does
  std::vector<reprojector::Candidate>* lists[3] = { &candidates_, &converged_, &unconverged_ };
  std::vector<uint32_t> order;
  // (a list that was not planned is sorted when -- if -- its pass is reached: planPausedPass, or matchCandidates' own sort... the
  // reference sorts it right before its pass as well, reprojector.cpp:263)
  const int n_sorted = sort_unplanned_lists_ ? 3 : n_speculated_;
  lists_sorted_ = n_sorted;
  for (int k = 0; k < n_sorted; ++k) {
    reprojector_utils::sortCandidatesWithOrder(*lists[k], &order);
    if (k < n_speculated_ && !order.empty()) {
      std::vector<detail::Resolved> sorted(order.size());
      for (size_t i = 0; i < order.size(); ++i) sorted[i] = std::move(plan_rs_[k][order[i]]);
      plan_rs_[k].swap(sorted);
    }
  }
}

void ReprojectorHip::replayMatches(const FramePtr& cur_frame, svoh_ctx* ctx_for_unspeculated)
{
  replay_next_pass_ = 0; replay_paused_ = false;
  (void)replayPasses(cur_frame, ctx_for_unspeculated, false);
}

bool ReprojectorHip::replayMatchesUntilUnplanned(const FramePtr& cur_frame)
{
  replay_next_pass_ = 0; replay_paused_ = false;
  return replayPasses(cur_frame, nullptr, true);
}

void ReprojectorHip::planPausedPass(const FramePtr& cur_frame, bool resident_features, const std::vector<FramePtr>* visible_kfs)
{
  if (!replay_paused_) throw std::runtime_error("ReprojectorHip::planPausedPass: the replay is not waiting for a pass");
  std::vector<reprojector::Candidate>* lists[3] = { &candidates_, &converged_, &unconverged_ };
  if (replay_next_pass_ == 2 && unconverged_pending_) {   // the walk left the unconverged seeds for now: their turn
    if (!visible_kfs) throw std::runtime_error("ReprojectorHip::planPausedPass: the unconverged seeds were not walked, and no keyframes to walk them over");
    std::vector<PointPtr> no_trash;
    walkLists(cur_frame, *visible_kfs, no_trash, 2);
    lists_sorted_ = lists_sorted_ < 2 ? lists_sorted_ : 2;
  }
  // (the batches of the passes that have been replayed are done with; the list is sorted now if it was not planned and so not sorted)
  if (replay_next_pass_ >= lists_sorted_) { reprojector_utils::sortCandidatesByReprojStats(*lists[replay_next_pass_]); lists_sorted_ = replay_next_pass_ + 1; }
  sm_->clear();
  sm_->setResident(resident_features);
  plan_rs_[replay_next_pass_] = sm_->plan(cur_frame, *lists[replay_next_pass_]);
  n_speculated_ = replay_next_pass_ + 1;
}

bool ReprojectorHip::resumeReplay(const FramePtr& cur_frame)
{
  if (!replay_paused_) throw std::runtime_error("ReprojectorHip::resumeReplay: the replay is not waiting for a pass");
  return replayPasses(cur_frame, nullptr, true);
}

// the passes from replay_next_pass_ on; pause_at_unplanned: a pass that has to run and was not planned stops the replay (true is
// returned, everything stays as it is) until planPausedPass + the caller's batch + resumeReplay
bool ReprojectorHip::replayPasses(const FramePtr& cur_frame, svoh_ctx* ctx_for_unspeculated, bool pause_at_unplanned)
{
  const size_t max_total_n_features = options_.max_n_features_per_frame;   // + max_n_fixed_lm, 0 without the global map
  std::vector<reprojector::Candidate>&converged = converged_, &unconverged = unconverged_;
  std::vector<reprojector::Candidate>* lists[3] = { &candidates_, &converged, &unconverged };
  reprojector::Statistics st[3];
  auto add = [&](const reprojector::Statistics& stt) { stats_.n_matches += stt.n_matches; stats_.n_trials += stt.n_trials; };
  auto before_pass = [&](int pass, size_t& max_n) -> bool {
    max_n = max_total_n_features;
    if (pass == 0) return true;
    if (pass == 1) {
      if (doesFrameHaveEnoughFeatures(cur_frame)) { reprojector_utils::setGridCellsOccupied(converged, *grid_); return false; }   // :226-231
      return true;
    }
    // pass 2: the feature budget of the unconverged seeds (:269-284)
    size_t max_allowed_total = max_total_n_features;
    if (options_.max_unconverged_seeds_ratio > 0) {
      const double min_lm_seeds_ratio = 1 - options_.max_unconverged_seeds_ratio;
      const size_t max_allowed_alternative = static_cast<size_t>(cur_frame->numTrackedFeatures() / min_lm_seeds_ratio);
      if (max_allowed_total > max_allowed_alternative) max_allowed_total = max_allowed_alternative;
    }
    if (max_allowed_total < options_.min_required_features) max_allowed_total = options_.min_required_features;
    max_n = max_allowed_total;
    return true;
  };
  bool& stop = replay_stop_;
  auto after_pass = [&](int pass) {
    add(st[pass]);
    if (pass == 0) {
      if (doesFrameHaveEnoughFeatures(cur_frame)) reprojector_utils::setGridCellsOccupied(candidates_, *grid_);   // :193-199
    } else if (pass == 1) {
      if (doesFrameHaveEnoughFeatures(cur_frame) || !options_.reproject_unconverged_seeds) {                        // :236-241
        reprojector_utils::setGridCellsOccupied(converged, *grid_);
        stop = true;
      }
    } else {
      if (doesFrameHaveEnoughFeatures(cur_frame)) reprojector_utils::setGridCellsOccupied(unconverged, *grid_);    // :300-305
    }
  };
  const bool resumed = replay_paused_;
  if (!resumed) { reached_unconverged_ = false; stop = false; }
  replay_paused_ = false;
  for (int k = replay_next_pass_; k < 3; ++k) {
    size_t max_n = 0;
    if (resumed && k == replay_next_pass_) max_n = replay_max_n_;   // (before_pass ran when the replay stopped here)
    else if (stop || !before_pass(k, max_n)) break;
    if (k >= n_speculated_ && pause_at_unplanned) {
      replay_next_pass_ = k; replay_max_n_ = max_n; replay_paused_ = true;
      return true;
    }
    if (k < n_speculated_) sm_->replay(cur_frame, max_n, *lists[k], plan_rs_[k], *grid_, st[k], device_select_ ? ctx_for_unspeculated : nullptr);
    else {
      // a pass nobody bet on: its own round trip
      if (!ctx_for_unspeculated) throw std::runtime_error("ReprojectorHip::replayMatches: a pass that was not planned, and no context to match it on");
      reprojector_utils::matchCandidates(ctx_for_unspeculated, cur_frame, max_n, options_.affine_est_offset, options_.affine_est_gain, *lists[k], *grid_, st[k],
                                         options_.seed_sigma2_thresh);
    }
    if (k == 2) reached_unconverged_ = true;
    after_pass(k);
  }
  // (the candidate lists are emptied on return: their buffers stay, the frame references go)
  candidates_.clear(); converged.clear(); unconverged.clear();
  sm_->clear();
  replay_next_pass_ = 0;
  return false;
}

void ReprojectorHip::reprojectFrames(const FramePtr& cur_frame, const std::vector<FramePtr>& visible_kfs,
                                     std::vector<PointPtr>& trash_points)
{
  const double ts0 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  // device projection queued for this frame (enqueueCandidateProjection): take it out of the context now
  const bool queued = proj_frame_ == cur_frame.get() && proj_frame_id_ == cur_frame->id_;
  if (queued && !proj_collected_) {
    proj_px_.resize(2 * proj_n_points_); proj_visible_.resize(proj_n_points_);
    if (svoh_project_candidates_collect(ctx_, static_cast<int>(proj_n_points_), proj_px_.data(), proj_visible_.data()) != SVOH_OK)
      throw std::runtime_error(std::string("svoh_project_candidates_collect: ") + svoh_last_error_string(ctx_));
    proj_px_p_ = proj_px_.data(); proj_visible_p_ = proj_visible_.data();
    proj_collected_ = true;
  } else if (!queued) {
    discardCandidateProjection();
  }
  // whatever happens below, no frame reference, no open deferred section and no stale projection stay behind
  struct Release {
    ReprojectorHip* r; svoh_ctx* c;
    ~Release()
    {
      r->proj_frame_ = nullptr; r->proj_collected_ = false;
      r->candidates_.clear(); r->converged_.clear(); r->unconverged_.clear();
      if (r->sm_) { if (r->sm_->in_flight) { r->sm_->in_flight = false; (void)svoh_matcher_collect(c); } r->sm_->clear(); }
    }
  } release{ this, ctx_ };
  walkCandidates(cur_frame, visible_kfs, trash_points);
  // the three sortCandidatesByReprojStats calls (:188, 224, 263) happen below, while the matcher work of the lists is on
  // the device
  const double ts1 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  if (g_reproj_timing.on) (void)svoh_set_kernel_timing(ctx_, 1);
  // The unconverged seeds are many (up to max_n_kfs x max_seeds per frame) and their pass is only reached when the
  // landmarks and converged seeds did not fill the frame: their matcher work joins the common round trip only if the
  // pass was reached on the previous frame (a wrong guess costs one extra round trip or some unused work, nothing else).
  // Reprojector::reprojectFrames' three matchCandidates passes (reprojector.cpp:177-306) with ONE round trip to the
  // device: none of the three candidate lists depends on a match result (only on the visible keyframes and the current
  // pose), so all three are planned first, their matcher work runs as one direct batch plus one seed batch queued back
  // to back, and then the reference's control flow -- pass, enough-features test, grid marking, early return, the
  // feature budget of the unconverged seeds -- is replayed on finished results.  Work of a pass the control flow never
  // reaches is wasted, never visible: replay() alone touches the frame, the grid, the points and the seeds.
  const int n_speculated = speculate_unconverged_ ? 3 : 2;
  planMatches(cur_frame, n_speculated);
  const double tp1 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  sm_->enqueue(ctx_, cur_frame, options_.affine_est_offset, options_.affine_est_gain, options_.seed_sigma2_thresh);
  const double tp2 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  sortCandidateLists();
  const double tp3 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  sm_->finish(ctx_);
  sm_->direct.useOwnOutputs(); sm_->seeds.useOwnOutputs();
  double tp4 = tp3;
  if (g_reproj_timing.on) {
    g_reproj_timing.n_direct += (long)sm_->direct.size(); g_reproj_timing.n_seeds += (long)sm_->seeds.size(); g_reproj_timing.n_spec3 += n_speculated == 3;
    float kms = 0.f;
    if (svoh_last_kernel_ms(ctx_, &kms) == SVOH_OK) g_reproj_timing.kernel_ms += kms;
    tp4 = ReprojTiming::now();
  }
  replayMatches(cur_frame, ctx_);
  speculate_unconverged_ = reached_unconverged_;
  if (g_reproj_timing.on) {
    g_reproj_timing.n_reached3 += reached_unconverged_;
    const double ts4 = ReprojTiming::now();
    g_reproj_timing.t[0] += ts1 - ts0;                          // walk
    g_reproj_timing.t[2] += tp1 - ts1;                          // plan
    g_reproj_timing.t[3] += (tp2 - tp1) + (tp4 - tp3);          // device round trip(s)
    g_reproj_timing.t[1] += tp3 - tp2;                          // sort
    g_reproj_timing.t[5] += ts4 - tp4;                          // replay, grid
    ++g_reproj_timing.n;
    g_reproj_timing.end_call();
  }
}

namespace reprojector_utils {
// order[k] = where the k-th candidate of the sorted list stood before (empty for lists of fewer than two)
void sortCandidatesWithOrder(std::vector<reprojector::Candidate>& candidates, std::vector<uint32_t>* order)
{
  // std::sort(candidates, type > , n_reproj > , score >) of reprojector.cpp:545-556.  The order of EQUAL candidates is
  // whatever libstdc++'s introsort leaves, and its moves depend on the comparison results only: sorting 24-byte keys
  // with the same comparator and applying the permutation gives the vector the reference's call gives, for less
  // memory traffic than sorting the 64-byte candidates.
  // The three-field comparison as ONE comparison of a 128-bit key: type in the top bits, then n_reproj with its sign bit
  // flipped (signed order = unsigned order), then the score's bits mapped so that unsigned order is the order of the
  // doubles (negative: all bits flipped; non-negative: sign bit set; -0.0 taken as +0.0, which `>` cannot tell apart).
  // a.k > b.k is then exactly the reference's lambda (reprojector.cpp:549-555) for every pair -- except when a score
  // is NaN, which no mapping can order like `>` does: such a list is sorted with the lambda itself.
  struct Key { unsigned __int128 k; uint32_t at; };
  thread_local std::vector<Key> keys;
  thread_local std::vector<reprojector::Candidate> sorted;
  const size_t n = candidates.size();
  if (order) order->clear();
  if (n < 2) return;
  keys.resize(n);
  bool any_nan = false;
  for (size_t i = 0; i < n; ++i) {
    const reprojector::Candidate& c = candidates[i];
    double sc = c.score;
    any_nan = any_nan || sc != sc;
    if (sc == 0.0) sc = 0.0;   // -0.0 -> +0.0
    uint64_t bits;
    memcpy(&bits, &sc, sizeof bits);
    const uint64_t lo = (bits >> 63) ? ~bits : (bits | (uint64_t(1) << 63));
    const uint64_t hi = (uint64_t(c.type) << 32) | uint64_t(static_cast<uint32_t>(c.n_reproj) ^ 0x80000000u);
    keys[i] = Key{ (static_cast<unsigned __int128>(hi) << 64) | lo, static_cast<uint32_t>(i) };
  }
  if (!any_nan) std::sort(keys.begin(), keys.end(), [](const Key& lhs, const Key& rhs) { return lhs.k > rhs.k; });
  else std::sort(keys.begin(), keys.end(), [&candidates](const Key& kl, const Key& kr) {
    const reprojector::Candidate& lhs = candidates[kl.at];
    const reprojector::Candidate& rhs = candidates[kr.at];
    return lhs.type > rhs.type || (lhs.type == rhs.type && lhs.n_reproj > rhs.n_reproj) ||
           (lhs.type == rhs.type && lhs.n_reproj == rhs.n_reproj && lhs.score > rhs.score);
  });
  sorted.clear();
  sorted.reserve(n);
  for (size_t i = 0; i < n; ++i) sorted.push_back(std::move(candidates[keys[i].at]));
  candidates.swap(sorted);
  sorted.clear();
  if (order) { order->resize(n); for (size_t i = 0; i < n; ++i) (*order)[i] = keys[i].at; }
}

void sortCandidatesByReprojStats(std::vector<reprojector::Candidate>& candidates) { sortCandidatesWithOrder(candidates, nullptr); }

bool projectPointAndCheckVisibility(const FramePtr& frame, const svoh::Vec3& xyz, double* px)
{
  if (!frame->isVisible(xyz, px)) return false;
  const int pxi0 = static_cast<int>(px[0]), pxi1 = static_cast<int>(px[1]);   // px->cast<int>()
  constexpr int kPatchSize = 8;                                               // isKeypointVisibleWithMargin
  return pxi0 >= kPatchSize && pxi1 >= kPatchSize && pxi0 < frame->cam.width - kPatchSize && pxi1 < frame->cam.height - kPatchSize;
}

bool getCandidate(const FramePtr& cur_frame, const FramePtr& ref_frame, size_t ref_index, reprojector::Candidate& candidate)
{
  return getCandidate(cur_frame, ref_frame, ref_index, candidate, nullptr);
}

// T_world_ref: ref_frame->T_world_cam() when the caller has it already (one inverse per keyframe, not per seed)
bool getCandidate(const FramePtr& cur_frame, const FramePtr& ref_frame, size_t ref_index, reprojector::Candidate& candidate,
                  const svoh::Rigid* T_world_ref)
{
  svoh::Vec3 xyz_world{ 0, 0, 0 };
  int n_reproj = 0;
  const PointPtr lm = ref_index < ref_frame->landmark_vec_.size() ? ref_frame->landmark_vec_[ref_index] : nullptr;
  if (lm) {
    xyz_world = lm->pos();
    n_reproj = lm->n_succeeded_reproj_ - lm->n_failed_reproj_;
  } else {
    const double depth = ref_frame->getSeedDepth(ref_index);   // T_world_cam() * getSeedPosInFrame(ref_index)
    const svoh::Vec3 in_f{ ref_frame->f_vec_[3 * ref_index] * depth, ref_frame->f_vec_[3 * ref_index + 1] * depth,
                           ref_frame->f_vec_[3 * ref_index + 2] * depth };
    xyz_world = T_world_ref ? svoh::transform(*T_world_ref, in_f) : svoh::transform(svoh::inverse(ref_frame->T_f_w_), in_f);
  }
  double px[2];
  if (!projectPointAndCheckVisibility(cur_frame, xyz_world, px)) return false;
  candidate = reprojector::Candidate();
  candidate.ref_frame = ref_frame; candidate.ref_index = ref_index;
  candidate.cur_px[0] = px[0]; candidate.cur_px[1] = px[1];
  candidate.n_reproj = n_reproj;
  candidate.score = ref_index < ref_frame->score_vec_.size() ? ref_frame->score_vec_[ref_index] : 0.0;
  candidate.type = ref_frame->type_vec_[ref_index];
  candidate.n_obs = lm ? lm->obs_.size() : 0u;
  return true;
}

void setGridCellsOccupied(const std::vector<reprojector::Candidate>& candidates, OccupandyGrid2D& grid)
{
  for (const reprojector::Candidate& c : candidates)
    grid.setOccupied(grid.getCellIndex(static_cast<int>(c.cur_px[0]), static_cast<int>(c.cur_px[1]), 1));
}

namespace {
thread_local std::vector<int32_t> g_last_results;
}
std::vector<int32_t>& g_last_results_ref() { return g_last_results; }
const std::vector<int32_t>& lastMatchResults() { return g_last_results; }
}  // namespace reprojector_utils

namespace {
bool is_edgelet(uint8_t t) { return t == SVOH_FT_EDGELET || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_EDGELET_SEED_CONVERGED; }
bool is_converged_seed(uint8_t t)
{
  return t == SVOH_FT_CORNER_SEED_CONVERGED || t == SVOH_FT_EDGELET_SEED_CONVERGED || t == SVOH_FT_MAPPOINT_SEED_CONVERGED;
}
bool is_unconverged_seed(uint8_t t)
{
  return t == SVOH_FT_CORNER_SEED || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_MAPPOINT_SEED;
}
template <class T>
void grow(std::vector<T>& v, size_t n, const T& fill = T()) { if (v.size() < n) v.resize(n, fill); }
}  // namespace

// ---- detail::SpeculativeMatches (svo_hip_host_internal.h) ----
namespace detail {
using reprojector_utils::g_last_results_ref;

svoh_matcher_options reprojectorMatcherOptions(bool affine_est_offset, bool affine_est_gain)
{
  // Matcher matcher; (defaults of matcher.h:39-54) + the two affine flags (reprojector.cpp:352-354)
  svoh_matcher_options mopt{};
  mopt.align_max_iter = 10; mopt.max_epi_search_steps = 100; mopt.subpix_refinement = 1;
  mopt.epi_search_edgelet_filtering = 1; mopt.scan_on_unit_sphere = 1;
  mopt.epi_search_edgelet_max_angle = 0.7; mopt.max_patch_diff_ratio = 2.0;
  mopt.affine_est_offset = affine_est_offset; mopt.affine_est_gain = affine_est_gain;
  return mopt;
}

svoh_depth_filter_options reprojectorSeedOptions(const Frame& cur_frame, double seed_sigma2_thresh)
{
  svoh_depth_filter_options o{};
  o.seed_convergence_sigma2_thresh = seed_sigma2_thresh;      // updateSeed(..., seed_sigma2_thresh, false, false):
  o.mappoint_convergence_sigma2_thresh = seed_sigma2_thresh;  // one threshold for every seed type here
  o.px_error_angle = updateSeedPxErrorAngle(cur_frame);
  o.check_visibility = 0; o.check_convergence = 0; o.use_vogiatzis_update = 1;
  return o;
}

int SpeculativeMatches::slot_of(const FramePtr& f)
{
  if (last_slot >= 0 && frames[static_cast<size_t>(last_slot)].get() == f.get()) return last_slot;   // runs of one keyframe's features
  for (size_t k = 0; k < frames.size(); ++k) if (frames[k] == f) return last_slot = static_cast<int>(k);
  frames.push_back(f);
  return last_slot = static_cast<int>(frames.size() - 1);
}

std::vector<Resolved> SpeculativeMatches::plan(const FramePtr& frame, const std::vector<reprojector::Candidate>& candidates)
{
  const size_t n = candidates.size();
  std::vector<Resolved> rs(n);
  direct.reserve_more(n); seeds.reserve_more(n);
  const svoh::Vec3 cur_pos = frame->pos();
  for (size_t i = 0; i < n; ++i) {
    const reprojector::Candidate& c = candidates[i];
    if (!c.ref_frame || c.ref_index >= c.ref_frame->num_features_) throw std::runtime_error("matchCandidates: bad candidate");
    Resolved& r = rs[i];
    r.batch_pos = -1; r.frame_slot = -1; r.point = nullptr; r.ref = nullptr; r.idx = 0;
    Point* lm = c.ref_index < c.ref_frame->landmark_vec_.size() ? c.ref_frame->landmark_vec_[c.ref_index].get() : nullptr;
    if (!lm) {
      r.ref = c.ref_frame.get(); r.idx = c.ref_index;
      if (is_converged_seed(c.type)) r.kind = kConvergedSeed;
      else if (is_unconverged_seed(c.type)) r.kind = kUnconvergedSeed;
      else throw std::runtime_error("matchCandidates: seed type unknown");  // CHECK(false) in the reference
      r.frame_slot = slot_of(c.ref_frame);
    } else {
      r.point = lm;
      FramePtr rf; size_t ri = 0;
      if (lm->getCloseViewObs(cur_pos, rf, ri)) { r.kind = kLandmark; r.ref = rf.get(); r.idx = ri; r.frame_slot = slot_of(rf); }
      else r.kind = kNoCloseView;
    }
    if (direct.resident && r.ref && !r.ref->features) throw std::runtime_error("matchCandidates: a reference frame without resident feature columns in a batch that names features by index");
    if (r.kind == kConvergedSeed || r.kind == kLandmark) {
      r.batch_pos = static_cast<int>(direct.size());
      direct.push(*r.ref, r.idx, r.frame_slot);
      if (r.kind == kConvergedSeed) direct.depth.push_back(r.ref->getSeedDepth(r.idx));
      else {
        const svoh::Vec3 p = r.ref->pos(), q = r.point->pos();  // (ref_frame->pos() - landmark->pos()).norm()
        direct.depth.push_back(sqrt((p.x - q.x) * (p.x - q.x) + (p.y - q.y) * (p.y - q.y) + (p.z - q.z) * (p.z - q.z)));
      }
      direct.px_cur.push_back(c.cur_px[0]); direct.px_cur.push_back(c.cur_px[1]);
    } else if (r.kind == kUnconvergedSeed) {
      r.batch_pos = static_cast<int>(seeds.size());
      seeds.push(*r.ref, r.idx, r.frame_slot);
      const double* st = &r.ref->invmu_sigma2_a_b_vec_[4 * r.idx];
      seeds.state.push_back(st[0]); seeds.state.push_back(st[1]); seeds.state.push_back(st[2]); seeds.state.push_back(st[3]);
    }
  }
  return rs;
}

void SpeculativeMatches::enqueue(svoh_ctx* ctx, const FramePtr& frame, bool affine_est_offset, bool affine_est_gain, double seed_sigma2_thresh)
{
  if (!direct.size() && !seeds.size()) return;
  if (direct.resident) throw std::runtime_error("SpeculativeMatches::enqueue: batches planned with resident_features are staged by their driver");
  const svoh_matcher_options mopt = reprojectorMatcherOptions(affine_est_offset, affine_est_gain);
  thread_local std::vector<svoh_frame_view> views;
  views.clear();
  for (const FramePtr& f : frames) views.push_back(viewOf(*f));
  const svoh_frame_view cur = viewOf(*frame);
  auto batch_of = [](Batch& b) {
    svoh_feature_batch fb{};
    fb.n = static_cast<int32_t>(b.size());
    fb.ref_frame_idx = b.ref_idx.data(); fb.px = b.px.data(); fb.f = b.f.data(); fb.grad = b.grad.data();
    fb.level = b.level.data(); fb.type = b.type.data();
    return fb;
  };
  auto fail = [&](const char* what) {
    const std::string msg = std::string(what) + ": " + svoh_last_error_string(ctx);
    (void)svoh_matcher_collect(ctx);   // leave no open section behind
    throw std::runtime_error(msg);
  };
  // a seed update still in flight (DepthFilterHip::updateSeedsAsync) holds the context's one deferred section: a batch
  // issued now would be queued INTO it instead of running -- finish the update first
  finishPendingSeedUpdate(ctx);
  if (svoh_matcher_begin_deferred(ctx) != SVOH_OK) throw std::runtime_error(std::string("svoh_matcher_begin_deferred: ") + svoh_last_error_string(ctx));
  svoh_feature_batch fbd{}, fbs{};
  const double tr0 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  if (direct.size()) {
    const size_t m = direct.size();
    // (outputs: svoh_matcher_collect writes every entry of each -- no fill needed)
    direct.result.resize(m); direct.search_level.resize(m); direct.f_cur.resize(3 * m); direct.A.resize(4 * m);
    fbd = batch_of(direct);
    const int rc = svoh_match_direct_batch(ctx, &mopt, static_cast<int>(views.size()), views.data(), &cur, &fbd,
                                           direct.depth.data(), direct.px_cur.data(), direct.result.data(),
                                           direct.f_cur.data(), direct.search_level.data(), nullptr, direct.A.data());
    if (rc != SVOH_OK) fail("svoh_match_direct_batch");
  }
  const double tr1 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  if (seeds.size()) {
    const size_t m = seeds.size();
    seeds.result.resize(m); seeds.search_level.resize(m); seeds.success.resize(m);
    seeds.px_cur.resize(2 * m); seeds.f_cur.resize(3 * m); seeds.A.resize(4 * m);
    fbs = batch_of(seeds);
    const svoh_depth_filter_options o = reprojectorSeedOptions(*frame, seed_sigma2_thresh);
    const svoh_seed_match_outputs outs{ seeds.px_cur.data(), seeds.f_cur.data(), seeds.search_level.data(), seeds.A.data() };
    const int rc = svoh_update_seeds_batch_ex(ctx, &mopt, &o, static_cast<int>(views.size()), views.data(), &cur, &fbs,
                                              seeds.state.data(), seeds.success.data(), seeds.result.data(), nullptr, &outs);
    if (rc != SVOH_OK) fail("svoh_update_seeds_batch_ex");
  }
  if (svoh_matcher_flush(ctx) != SVOH_OK) fail("svoh_matcher_flush");
  in_flight = true;
  if (g_reproj_timing.on) { const double tr2 = ReprojTiming::now(); g_reproj_timing.rt[0] += tr1 - tr0; g_reproj_timing.rt[1] += tr2 - tr1; }
}

void SpeculativeMatches::finish(svoh_ctx* ctx)
{
  if (!in_flight) return;
  in_flight = false;
  const double tr2 = g_reproj_timing.on ? ReprojTiming::now() : 0.0;
  if (svoh_matcher_collect(ctx) != SVOH_OK) throw std::runtime_error(std::string("svoh_matcher_collect: ") + svoh_last_error_string(ctx));
  if (g_reproj_timing.on) g_reproj_timing.rt[2] += ReprojTiming::now() - tr2;
}

void SpeculativeMatches::replay(const FramePtr& frame, size_t max_n_features_per_frame, std::vector<reprojector::Candidate>& candidates,
                                const std::vector<Resolved>& rs, OccupandyGrid2D& grid, reprojector::Statistics& stats, svoh_ctx* select_on)
{
  std::vector<int32_t>& g_last_results = g_last_results_ref();
  const size_t n = candidates.size();
  g_last_results.assign(n, -1);
  // the selection from the device (optional): visited[k] = the loop below tries candidate k; n_stop = where it ends
  std::vector<uint8_t> visited;
  size_t n_stop = n;
  if (select_on && max_n_features_per_frame > 0 && n > 0) {
    std::vector<int32_t> cell(n);
    std::vector<uint8_t> success(n), occ(grid.size());
    for (size_t k = 0; k < n; ++k) {
      cell[k] = static_cast<int32_t>(grid.getCellIndex(static_cast<int>(candidates[k].cur_px[0]), static_cast<int>(candidates[k].cur_px[1]), 1));
      const Resolved& r = rs[k];
      success[k] = (r.kind == kConvergedSeed || r.kind == kLandmark) ? direct.out.result[r.batch_pos] == SVOH_MATCH_SUCCESS
                   : r.kind == kUnconvergedSeed ? seeds.out.success[r.batch_pos] != 0 : 0;
    }
    for (size_t c = 0; c < occ.size(); ++c) occ[c] = grid.isOccupied(c) ? 1 : 0;
    const int32_t begin[2] = { 0, static_cast<int32_t>(n) };
    const int32_t max_n = static_cast<int32_t>(max_n_features_per_frame);
    int32_t n_feat = static_cast<int32_t>(frame->num_features_), n_trials = 0, n_matches = 0, n_consumed = 0;
    visited.assign(n, 0);
    if (svoh_select_matches_batch(select_on, 1, begin, cell.data(), success.data(), static_cast<int>(occ.size()), occ.data(), &max_n, &n_feat, visited.data(), &n_trials,
                                  &n_matches, &n_consumed) != SVOH_OK)
      throw std::runtime_error(std::string("svoh_select_matches_batch: ") + svoh_last_error_string(select_on));
    n_stop = static_cast<size_t>(n_consumed);
  }
  size_t i = 0;
  // The batches' outputs were written by the device a moment ago: every first touch of a line is a miss in all of the host's
  // caches.  Which candidates will be tried is only known when the loop gets there, so the result code of EVERY candidate a little
  // ahead is asked for early (one line each), and a successful match's pixel / bearing vector / warp when its code turns up.
  constexpr size_t kAhead = 24;
  auto prefetch_result = [&](size_t k2) {
    if (k2 >= n_stop) return;
    const Resolved& r2 = rs[k2];
    if (r2.batch_pos < 0) return;
    if (r2.kind == kUnconvergedSeed) __builtin_prefetch(&seeds.out.success[r2.batch_pos]);
    else if (r2.kind == kConvergedSeed || r2.kind == kLandmark) __builtin_prefetch(&direct.out.result[r2.batch_pos]);
  };
  for (size_t k2 = 0; k2 < kAhead; ++k2) prefetch_result(k2);
  for (size_t k = 0; k < n_stop; ++k) {
    reprojector::Candidate& c = candidates[k];
    const Resolved& r = rs[k];
    ++i;
    prefetch_result(k + kAhead);
    const size_t grid_index = grid.getCellIndex(static_cast<int>(c.cur_px[0]), static_cast<int>(c.cur_px[1]), 1);
    if (!visited.empty()) { if (!visited[k]) continue; }
    else if (max_n_features_per_frame > 0 && grid.isOccupied(grid_index)) continue;
    ++stats.n_trials;
    bool ok = false;
    const Batch* b = nullptr;
    if (r.kind == kConvergedSeed || r.kind == kLandmark) {
      b = &direct;
      const int res = direct.out.result[r.batch_pos];
      g_last_results[k] = res;
      ok = res == SVOH_MATCH_SUCCESS;
      c.cur_px[0] = direct.out.px_cur[2 * r.batch_pos]; c.cur_px[1] = direct.out.px_cur[2 * r.batch_pos + 1];  // Keypoint& px_cur
      if (r.kind == kLandmark) { if (ok) r.point->n_succeeded_reproj_ += 1; else r.point->n_failed_reproj_++; }
    } else if (r.kind == kUnconvergedSeed) {
      b = &seeds;
      g_last_results[k] = seeds.out.result[r.batch_pos];
      ok = seeds.out.success[r.batch_pos] != 0;
      // updateSeed changed the seed of the reference frame whether it succeeded or not
      std::copy(seeds.out.state + 4 * r.batch_pos, seeds.out.state + 4 * r.batch_pos + 4, r.ref->invmu_sigma2_a_b_vec_.begin() + 4 * r.idx);
      r.ref->type_vec_[r.idx] = seeds.out.type[r.batch_pos];
    } else {
      g_last_results[k] = 1000;
    }
    if (!ok) continue;
    // matchCandidate's tail (:455-486): fill the first free slot of the frame
    const size_t s = frame->num_features_;
    grow(frame->px_vec_, 2 * (s + 1)); grow(frame->f_vec_, 3 * (s + 1)); grow(frame->grad_vec_, 2 * (s + 1));
    grow(frame->level_vec_, s + 1); grow(frame->type_vec_, s + 1); grow(frame->score_vec_, s + 1);
    grow(frame->invmu_sigma2_a_b_vec_, 4 * (s + 1)); grow(frame->landmark_vec_, s + 1); grow(frame->seed_ref_vec_, s + 1);
    const int p = r.batch_pos;
    if (is_edgelet(c.type)) {
      const double* A = &b->out.A[4 * p];
      const double* g = &r.ref->grad_vec_[2 * r.idx];
      double g0 = A[0] * g[0] + A[2] * g[1], g1 = A[1] * g[0] + A[3] * g[1];
      const double z = g0 * g0 + g1 * g1;
      if (z > 0.0) { const double nn = sqrt(z); g0 /= nn; g1 /= nn; }
      frame->grad_vec_[2 * s] = g0; frame->grad_vec_[2 * s + 1] = g1;
    }
    frame->type_vec_[s] = c.type;
    frame->px_vec_[2 * s] = b->out.px_cur[2 * p]; frame->px_vec_[2 * s + 1] = b->out.px_cur[2 * p + 1];
    for (int j = 0; j < 3; ++j) frame->f_vec_[3 * s + j] = b->out.f_cur[3 * p + j];
    frame->level_vec_[s] = b->out.search_level[p];
    frame->score_vec_[s] = c.score;
    if (r.kind == kLandmark) frame->landmark_vec_[s] = c.ref_frame->landmark_vec_[c.ref_index];   // == r.point, as the shared_ptr
    else { frame->seed_ref_vec_[s].keyframe = c.ref_frame; frame->seed_ref_vec_[s].seed_id = static_cast<int>(c.ref_index); }
    std::copy(c.ref_frame->invmu_sigma2_a_b_vec_.begin() + 4 * c.ref_index,
              c.ref_frame->invmu_sigma2_a_b_vec_.begin() + 4 * c.ref_index + 4, frame->invmu_sigma2_a_b_vec_.begin() + 4 * s);
    ++stats.n_matches;
    ++frame->num_features_;
    grid.setOccupied(grid_index);
    if (max_n_features_per_frame > 0 && frame->num_features_ >= max_n_features_per_frame) break;
  }
  candidates.erase(candidates.begin(), candidates.begin() + static_cast<std::ptrdiff_t>(i));
}
}  // namespace detail

namespace reprojector_utils {
using detail::SpeculativeMatches;
using detail::Resolved;
static bool device_select_env()
{
  static const bool on = getenv("SVOH_REPROJ_DEVICE_SELECT") != nullptr && atoi(getenv("SVOH_REPROJ_DEVICE_SELECT")) != 0;
  return on;
}

void matchCandidates(svoh_ctx* ctx, const FramePtr& frame, size_t max_n_features_per_frame, bool affine_est_offset,
                     bool affine_est_gain, std::vector<reprojector::Candidate>& candidates, OccupandyGrid2D& grid,
                     reprojector::Statistics& stats, double seed_sigma2_thresh)
{
  if (!ctx) throw std::runtime_error("matchCandidates: NULL svoh_ctx (no CPU fallback exists)");
  if (!frame) throw std::runtime_error("matchCandidates: NULL frame");
  g_last_results.assign(candidates.size(), -1);
  if (candidates.empty()) return;
  SpeculativeMatches sm;
  const std::vector<Resolved> rs = sm.plan(frame, candidates);
  sm.run(ctx, frame, affine_est_offset, affine_est_gain, seed_sigma2_thresh);
  sm.direct.useOwnOutputs(); sm.seeds.useOwnOutputs();
  // (SVOH_REPROJ_DEVICE_SELECT=1: the loop's selection from svoh_select_matches_batch, as in ReprojectorHip::replayMatches)
  sm.replay(frame, max_n_features_per_frame, candidates, rs, grid, stats, device_select_env() ? ctx : nullptr);
}

// Reprojector::reprojectFrames' three matchCandidates passes (reprojector.cpp:177-306) with ONE round trip to the
// device (see ReprojectorHip::reprojectFrames, which is made of the same pieces): plan, send off, sort, wait, replay.
void matchCandidatesFused(svoh_ctx* ctx, const FramePtr& frame, bool affine_est_offset, bool affine_est_gain, double seed_sigma2_thresh,
                          std::vector<reprojector::Candidate>* lists[3], const std::function<bool(int pass, size_t& max_n)>& before_pass,
                          const std::function<void(int pass)>& after_pass, OccupandyGrid2D& grid, reprojector::Statistics stats[3],
                          int n_speculated, bool sort_in_flight)
{
  if (!ctx) throw std::runtime_error("matchCandidatesFused: NULL svoh_ctx (no CPU fallback exists)");
  thread_local SpeculativeMatches sm;   // keeps its buffers from frame to frame ...
  sm.clear();
  // ... but no frame reference past the call, and no deferred section left open if something throws in between
  struct Release { SpeculativeMatches& s; svoh_ctx* c; ~Release() { if (s.in_flight) { s.in_flight = false; (void)svoh_matcher_collect(c); } s.clear(); } } release{ sm, ctx };
  std::vector<Resolved> rs[3];
  for (int k = 0; k < 3 && k < n_speculated; ++k) rs[k] = sm.plan(frame, *lists[k]);
  sm.enqueue(ctx, frame, affine_est_offset, affine_est_gain, seed_sigma2_thresh);
  if (sort_in_flight) {
    thread_local std::vector<uint32_t> order;
    for (int k = 0; k < 3; ++k) {
      sortCandidatesWithOrder(*lists[k], &order);
      if (k < n_speculated && !order.empty()) {
        std::vector<Resolved> sorted(order.size());
        for (size_t i = 0; i < order.size(); ++i) sorted[i] = std::move(rs[k][order[i]]);
        rs[k].swap(sorted);
      }
    }
  }
  sm.finish(ctx);
  sm.direct.useOwnOutputs(); sm.seeds.useOwnOutputs();
  for (int k = 0; k < 3; ++k) {
    size_t max_n = 0;
    if (!before_pass(k, max_n)) break;
    if (k < n_speculated) sm.replay(frame, max_n, *lists[k], rs[k], grid, stats[k], device_select_env() ? ctx : nullptr);
    else matchCandidates(ctx, frame, max_n, affine_est_offset, affine_est_gain, *lists[k], grid, stats[k], seed_sigma2_thresh);   // a pass nobody bet on: its own round trip
    after_pass(k);
  }
}
}  // namespace reprojector_utils

// ---- structure optimisation -------------------------------------------------------
int StructureBatch::gather(const Frame& frame, int max_n_pts)
{
  pts.clear(); views.clear(); obs_begin.assign(1, 0); obs_view.clear(); obs_f.clear(); pos.clear();
  stamp = frame.id_;
  for (size_t i = 0; i < frame.num_features_; ++i) {
    if (i >= frame.landmark_vec_.size() || frame.landmark_vec_[i] == nullptr) continue;
    const uint8_t t = frame.type_vec_[i];   // isEdgelet (types.h:113-118)
    if (t == SVOH_FT_EDGELET || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_EDGELET_SEED_CONVERGED) continue;
    pts.push_back(frame.landmark_vec_[i]);
  }
  if (max_n_pts > 0) {
    // the reference partitions the candidates here and then still loops over all of them (:804-819)
    const size_t n = std::min(static_cast<size_t>(max_n_pts), pts.size());
    max_n_pts = static_cast<int>(n);
    std::nth_element(pts.begin(), pts.begin() + n, pts.end(),
                     [](const PointPtr& lhs, const PointPtr& rhs) { return lhs->last_structure_optim_ < rhs->last_structure_optim_; });
  }
  // views = the frames the observations live in
  std::vector<const Frame*> view_frames;
  pos.resize(3 * pts.size());
  for (size_t k = 0; k < pts.size(); ++k) {
    const Point& pt = *pts[k];
    pos[3 * k] = pt.pos_.x; pos[3 * k + 1] = pt.pos_.y; pos[3 * k + 2] = pt.pos_.z;
    for (const Point::Obs& obs : pt.obs_) {
      const FramePtr f = obs.frame.lock();
      if (!f) continue;   // "could not unlock weak_ptr<Frame> in Point::optimize": the observation is skipped
      size_t v = 0;
      while (v < view_frames.size() && view_frames[v] != f.get()) ++v;
      if (v == view_frames.size()) {
        view_frames.push_back(f.get());
        svoh_se3 T;
        svoh::store_rigid(f->T_f_w_, T);
        views.push_back(T);
      }
      obs_view.push_back(static_cast<int32_t>(v));
      for (int c = 0; c < 3; ++c) obs_f.push_back(f->f_vec_[3 * obs.keypoint_index_ + c]);
    }
    obs_begin.push_back(static_cast<int32_t>(obs_view.size()));
  }
  return max_n_pts;
}

void StructureBatch::apply(const double* pos_out)
{
  for (size_t k = 0; k < pts.size(); ++k) {
    pts[k]->pos_ = svoh::Vec3{ pos_out[3 * k], pos_out[3 * k + 1], pos_out[3 * k + 2] };
    pts[k]->last_structure_optim_ = stamp;
  }
}

size_t optimizeStructure(svoh_ctx* ctx, const FrameBundle::Ptr& frames, int max_n_pts, int max_iter)
{
  if (!ctx) throw std::runtime_error("optimizeStructure: NULL svoh_ctx (no CPU fallback exists)");
  if (!frames) throw std::runtime_error("optimizeStructure: NULL frame bundle");
  if (max_n_pts == 0) return 0;   // -1 = optimise all points (frame_handler_base.cpp:785-786)
  size_t n_total = 0;
  StructureBatch b;
  for (const FramePtr& frame : frames->frames_) {
    const bool optimize_on_sphere = false;   // Camera::Type::kOmni only (:794-796); svoh_camera has no such model
    max_n_pts = b.gather(*frame, max_n_pts);
    if (b.pts.empty()) continue;
    // Point::optimize returns before touching anything when obs_.size() < 2 (point.cpp:255-259); the device applies
    // the same rule to the observations it is given.  (The two differ only for a landmark whose stored observations
    // are >= 2 while fewer than two of their frames are still alive, which the reference reports as an error.)
    const int rc = svoh_optimize_points_batch(ctx, max_iter, optimize_on_sphere ? 1 : 0, static_cast<int>(b.views.size()), b.views.data(),
                                              static_cast<int>(b.pts.size()), b.obs_begin.data(), b.obs_view.data(), b.obs_f.data(),
                                              b.pos.data(), nullptr);
    if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_optimize_points_batch: ") + svoh_last_error_string(ctx));
    b.apply(b.pos.data());
    n_total += b.pts.size();
  }
  return n_total;
}

// ---- FrameHandlerBase::upgradeSeedsToFeatures (frame_handler_base.cpp:828-920) -----------------------------------
size_t upgradeSeedsToFeatures(const FramePtr& frame, int* next_point_id, std::vector<size_t>* edgelets)
{
  if (!frame || !next_point_id || !edgelets) throw std::runtime_error("upgradeSeedsToFeatures: NULL argument");
  Frame& fr = *frame;
  const size_t n = fr.num_features_;
  if (fr.landmark_vec_.size() < n) fr.landmark_vec_.resize(n);
  if (fr.seed_ref_vec_.size() < n) fr.seed_ref_vec_.resize(n);
  if (fr.track_id_vec_.size() < n) fr.track_id_vec_.resize(n, -1);
  auto is_corner = [](uint8_t t) { return t == SVOH_FT_CORNER || t == SVOH_FT_CORNER_SEED || t == SVOH_FT_CORNER_SEED_CONVERGED; };       // types.h:106-111
  auto is_map_pt = [](uint8_t t) { return t == SVOH_FT_MAPPOINT || t == SVOH_FT_MAPPOINT_SEED || t == SVOH_FT_MAPPOINT_SEED_CONVERGED; }; // :120-125
  auto is_edgelet = [](uint8_t t) { return t == SVOH_FT_EDGELET || t == SVOH_FT_EDGELET_SEED || t == SVOH_FT_EDGELET_SEED_CONVERGED; };  // :113-118
  size_t update_count = 0;
  for (size_t i = 0; i < n; ++i) {
    if (fr.landmark_vec_[i]) {
      // (corner / edgelet / map point, or a fixed landmark: either way) landmark_vec_[i]->addObservation(frame, i)
      fr.landmark_vec_[i]->obs_.push_back(Point::Obs{ frame, i });
    } else if (fr.seed_ref_vec_[i].keyframe) {
      Frame::SeedRef& ref = fr.seed_ref_vec_[i];
      Frame& kf = *ref.keyframe;
      const size_t sid = static_cast<size_t>(ref.seed_id);
      if (kf.landmark_vec_.size() < kf.num_features_) kf.landmark_vec_.resize(kf.num_features_);
      if (kf.track_id_vec_.size() < kf.num_features_) kf.track_id_vec_.resize(kf.num_features_, -1);
      // in the multi-camera case another frame of the bundle may have made this seed's point already
      PointPtr point = kf.landmark_vec_[sid];
      if (!point) {
        // xyz_world = T_world_cam * getSeedPosInFrame(seed_id) (seed.h:145-149: f * depth)
        const double depth = kf.getSeedDepth(sid);
        point = std::make_shared<Point>();
        point->pos_ = svoh::transform(svoh::inverse(kf.T_f_w_), svoh::Vec3{ kf.f_vec_[3 * sid] * depth, kf.f_vec_[3 * sid + 1] * depth, kf.f_vec_[3 * sid + 2] * depth });
        point->id_ = (*next_point_id)++;
        kf.landmark_vec_[sid] = point;
        kf.track_id_vec_[sid] = point->id();
        point->obs_.push_back(Point::Obs{ ref.keyframe, sid });
      }
      fr.landmark_vec_[i] = point;
      fr.track_id_vec_[i] = point->id();
      point->obs_.push_back(Point::Obs{ frame, i });
      const uint8_t kt = kf.type_vec_[sid];
      if (is_corner(kt)) { kf.type_vec_[sid] = SVOH_FT_CORNER; fr.type_vec_[i] = SVOH_FT_CORNER; }
      else if (is_map_pt(kt)) { kf.type_vec_[sid] = SVOH_FT_MAPPOINT; fr.type_vec_[i] = SVOH_FT_MAPPOINT; }
      else if (is_edgelet(kt)) { kf.type_vec_[sid] = SVOH_FT_EDGELET; fr.type_vec_[i] = SVOH_FT_EDGELET; edgelets->push_back(i); }
      else throw std::runtime_error("upgradeSeedsToFeatures: Seed-Type not known");   // CHECK(false)
      ++update_count;
      // (a self reference -- the harness' stand-in for the initialiser's landmarks -- is a reference like any other)
      ref.keyframe.reset();
      ref.seed_id = -1;
    }
  }
  return update_count;
}

void refreshEdgeletDirections(svoh_ctx* ctx, const std::vector<FramePtr>& frames, const std::vector<std::vector<size_t>>& edgelets)
{
  if (!ctx) throw std::runtime_error("refreshEdgeletDirections: NULL svoh_ctx (no CPU fallback exists)");
  if (frames.size() != edgelets.size()) throw std::runtime_error("refreshEdgeletDirections: one list per frame");
  std::vector<svoh_frame_t> handles;
  std::vector<int32_t> fidx, level, px;
  for (size_t k = 0; k < frames.size(); ++k) {
    if (edgelets[k].empty()) continue;
    const Frame& f = *frames[k];
    const int32_t slot = static_cast<int32_t>(handles.size());
    handles.push_back(f.pyramid);
    for (size_t i : edgelets[k]) {
      const int lv = f.level_vec_[i];
      fidx.push_back(slot); level.push_back(lv);
      // (frame->px_vec_.col(i) / (1 << level)).cast<int>(): truncation
      px.push_back(static_cast<int32_t>(f.px_vec_[2 * i] / static_cast<double>(1 << lv)));
      px.push_back(static_cast<int32_t>(f.px_vec_[2 * i + 1] / static_cast<double>(1 << lv)));
    }
  }
  if (fidx.empty()) return;
  std::vector<int32_t> bins(fidx.size());
  if (svoh_histogram_angle_bins(ctx, static_cast<int>(handles.size()), handles.data(), static_cast<int>(fidx.size()), fidx.data(), level.data(), px.data(), bins.data()) != SVOH_OK)
    throw std::runtime_error(std::string("svoh_histogram_angle_bins: ") + svoh_last_error_string(ctx));
  size_t at = 0;
  for (size_t k = 0; k < frames.size(); ++k)
    for (size_t i : edgelets[k]) {
      const double angle = static_cast<double>(bins[at++]) * 2.0 * 3.14159265358979323846 / 36.0;   // getDominantAngle (:1001-1009)
      frames[k]->grad_vec_[2 * i] = std::cos(angle);
      frames[k]->grad_vec_[2 * i + 1] = std::sin(angle);
    }
}

void removeObservationsOf(const Frame& frame)
{
  for (size_t i = 0; i < frame.num_features_ && i < frame.landmark_vec_.size(); ++i) {
    if (!frame.landmark_vec_[i]) continue;
    std::vector<Point::Obs>& obs = frame.landmark_vec_[i]->obs_;
    obs.erase(std::remove_if(obs.begin(), obs.end(), [&](const Point::Obs& o) { const FramePtr f = o.frame.lock(); return !f || f.get() == &frame; }), obs.end());
  }
}

// ---- key points and the keyframe map ----------------------------------------------
void Frame::setKeyPoints()
{
  // frame.cpp:171-227, including its use of cv on the u axis for the two left-hand quadrants
  const double cu = static_cast<double>(cam.width / 2);
  const double cv = static_cast<double>(cam.height / 2);
  auto px = [this](int i, int c) { return px_vec_[2 * static_cast<size_t>(i) + c]; };
  for (size_t i = 0; i < num_features_; ++i) {
    if (i >= landmark_vec_.size() || landmark_vec_[i] == nullptr || type_vec_[i] == SVOH_FT_OUTLIER) continue;
    const double u = px_vec_[2 * i], v = px_vec_[2 * i + 1];
    const KeyPoint here{ static_cast<int>(i), landmark_vec_[i]->pos_ };
    // center
    if (key_pts_[0].first == -1)
      key_pts_[0] = here;
    else if (std::max(std::fabs(u - cu), std::fabs(v - cv)) <
             std::max(std::fabs(px(key_pts_[0].first, 0) - cu), std::fabs(px(key_pts_[0].first, 1) - cv)))
      key_pts_[0] = here;
    // corners
    auto corner = [&](int k) {
      if (key_pts_[k].first == -1)
        key_pts_[k] = here;
      else if ((u - cu) * (v - cv) > (px(key_pts_[k].first, 0) - cu) * (px(key_pts_[k].first, 1) - cv))
        key_pts_[k] = here;
    };
    if (u >= cu && v >= cv) corner(1);
    if (u >= cu && v < cv) corner(2);
    if (u < cv && v < cv) corner(3);
    if (u < cv && v >= cv) corner(4);
  }
}

void Map::addKeyframe(const FramePtr& new_keyframe, bool temporal_map)
{
  keyframes_.insert(std::make_pair(new_keyframe->id(), new_keyframe));
  last_added_kf_id_ = new_keyframe->id();
  if (temporal_map) sorted_keyframe_ids_.push_back(new_keyframe->id());
}

void Map::removeKeyframe(int frame_id)
{
  auto it_kf = keyframes_.find(frame_id);
  if (it_kf == keyframes_.end()) return;   // "Cannot find the keyframe ..., will not do anything"
  const FramePtr frame = it_kf->second;
  for (size_t i = 0; i < frame->num_features_ && i < frame->landmark_vec_.size(); ++i) {
    if (!frame->landmark_vec_[i]) continue;
    // Point::removeObservation(frame_id) (point.cpp:60-66)
    std::vector<Point::Obs>& obs = frame->landmark_vec_[i]->obs_;
    obs.erase(std::remove_if(obs.begin(), obs.end(),
                             [&](const Point::Obs& o) { const FramePtr f = o.frame.lock(); return f && f->id_ == frame_id; }),
              obs.end());
  }
  keyframes_.erase(it_kf);
}

void Map::getOverlapKeyframes(const FramePtr& frame, std::vector<std::pair<FramePtr, double>>* close_kfs) const
{
  if (!close_kfs || !frame) throw std::runtime_error("Map::getOverlapKeyframes: NULL argument");
  const svoh::Vec3 tp = frame->T_f_w_.t;   // T_f_w_.getPosition(): the translation of T_f_w, not the camera centre
  for (const auto& kf : keyframes_) {
    for (const Frame::KeyPoint& keypoint : kf.second->key_pts_) {
      if (keypoint.first == -1) continue;
      if (frame->isVisible(keypoint.second, nullptr)) {
        const svoh::Vec3 tk = kf.second->T_f_w_.t;
        const double dx = tp.x - tk.x, dy = tp.y - tk.y, dz = tp.z - tk.z;
        close_kfs->push_back(std::make_pair(kf.second, std::sqrt(dx * dx + dy * dy + dz * dz)));
        break;   // this keyframe has an overlapping field of view
      }
    }
  }
}

void Map::getClosestNKeyframesWithOverlap(const FramePtr& cur_frame, size_t num_frames, std::vector<FramePtr>* close_kfs) const
{
  if (!close_kfs) throw std::runtime_error("Map::getClosestNKeyframesWithOverlap: NULL argument");
  std::vector<std::pair<FramePtr, double>> overlap_kfs;
  getOverlapKeyframes(cur_frame, &overlap_kfs);
  if (overlap_kfs.empty()) return;
  const size_t N = std::min(num_frames, overlap_kfs.size());
  std::nth_element(overlap_kfs.begin(), overlap_kfs.begin() + N, overlap_kfs.end(),
                   [](const std::pair<FramePtr, double>& lhs, const std::pair<FramePtr, double>& rhs) { return lhs.second < rhs.second; });
  overlap_kfs.resize(N);
  close_kfs->reserve(num_frames);
  for (const auto& p : overlap_kfs) close_kfs->push_back(p.first);
}

FramePtr Map::getClosestKeyframe(const FramePtr& frame) const
{
  std::vector<std::pair<FramePtr, double>> close_kfs;
  getOverlapKeyframes(frame, &close_kfs);
  if (close_kfs.empty()) return nullptr;
  std::sort(close_kfs.begin(), close_kfs.end(),
            [](const std::pair<FramePtr, double>& lhs, const std::pair<FramePtr, double>& rhs) { return lhs.second < rhs.second; });
  if (close_kfs.at(0).first != frame) return close_kfs.at(0).first;
  if (close_kfs.size() == 1) return nullptr;
  return close_kfs.at(1).first;
}

FramePtr Map::getFurthestKeyframe(const svoh::Vec3& pos) const
{
  FramePtr furthest_kf;
  double maxdist = 0.0;
  for (const auto& kf : keyframes_) {
    const svoh::Vec3 p = kf.second->pos();
    const double dx = p.x - pos.x, dy = p.y - pos.y, dz = p.z - pos.z;
    const double dist = std::sqrt(dx * dx + dy * dy + dz * dz);
    if (dist > maxdist) { maxdist = dist; furthest_kf = kf.second; }
  }
  return furthest_kf;
}

FramePtr Map::getKeyframeById(int id) const
{
  auto it_kf = keyframes_.find(id);
  return it_kf == keyframes_.end() ? nullptr : it_kf->second;
}

void Map::getSortedKeyframes(std::vector<FramePtr>& kfs_sorted) const
{
  kfs_sorted.reserve(keyframes_.size());
  for (const auto& kv : keyframes_) kfs_sorted.push_back(kv.second);
  std::sort(kfs_sorted.begin(), kfs_sorted.end(), [](const FramePtr& left, const FramePtr& right) { return left->id_ < right->id_; });
}

// ---- alignPyr2DVec ----------------------------------------------------------------
namespace feature_alignment {
void alignPyr2DVec(svoh_ctx* ctx, svoh_frame_t img_pyr_ref, svoh_frame_t img_pyr_cur, int max_level, int min_level,
                   const std::vector<int>& patch_sizes, int n_iter, float min_update_squared,
                   const std::vector<Point2f>& px_ref, std::vector<Point2f>& px_cur, std::vector<uint8_t>& status)
{
  const size_t n = px_ref.size();
  if (px_cur.size() != n) throw std::runtime_error("alignPyr2DVec: px_ref and px_cur differ in size");
  status.resize(n);
  if (n == 0) return;
  svoh_klt_options o{};
  o.max_level = max_level; o.min_level = min_level; o.max_iter = n_iter; o.min_update_squared = min_update_squared;
  for (size_t l = 0; l < patch_sizes.size() && l < SVOH_MAX_LEVELS; ++l) o.patch_sizes[l] = patch_sizes[l];
  std::vector<int32_t> pr(2 * n);
  std::vector<double> pc(2 * n);
  std::vector<svoh_frame_t> refs(n, img_pyr_ref);
  for (size_t i = 0; i < n; ++i) {
    pr[2 * i] = static_cast<int32_t>(px_ref[i].x);  // Eigen::Vector2i(px_ref[i].x, px_ref[i].y): truncation
    pr[2 * i + 1] = static_cast<int32_t>(px_ref[i].y);
    pc[2 * i] = px_cur[i].x; pc[2 * i + 1] = px_cur[i].y;
  }
  const int rc = svoh_klt_track_batch(ctx, &o, static_cast<int>(n), refs.data(), img_pyr_cur, pr.data(), pc.data(), status.data());
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_klt_track_batch: ") + svoh_last_error_string(ctx));
  for (size_t i = 0; i < n; ++i) { px_cur[i].x = static_cast<float>(pc[2 * i]); px_cur[i].y = static_cast<float>(pc[2 * i + 1]); }
}
}  // namespace feature_alignment

}  // namespace svo_hip
