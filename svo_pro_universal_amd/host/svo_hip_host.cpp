#include "svo_hip_host.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace svo_hip {

SparseImgAlignHip::SparseImgAlignHip(svoh_ctx* ctx, SolverOptions solver_options, SparseImgAlignOptions options)
    : ctx_(ctx), solver_options_(solver_options), options_(options)
{
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

size_t SparseImgAlignHip::run(const FrameBundle::Ptr& ref_frames, const FrameBundle::Ptr& cur_frames)
{
  if (!ref_frames || !cur_frames || ref_frames->empty() || ref_frames->size() != cur_frames->size())
    throw std::runtime_error("SparseImgAlignHip::run: bundles must be non-empty and of equal size");
  if (ref_frames->size() > SVOH_MAX_CAMS) throw std::runtime_error("SparseImgAlignHip::run: too many cameras");

  svoh_align_options opt{};
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

  svoh_align_problem pb{};
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
    cam.n_features = static_cast<int32_t>(r.num_features_);
    cam.mem_space = SVOH_MEM_HOST;
    cam.px = r.px_vec_.data();
    cam.f = r.f_vec_.data();
    cam.pos_world = r.pos_world_.data();
    cam.flags = r.alignable_.data();
  }
  // T_iref_world_ and the optimisation variable (sparse_img_align.cpp:62, 74-75)
  const Transformation T_iref_world = ref_frames->at(0)->T_imu_world();
  const Transformation T_icur_iref = svoh::mul(cur_frames->at(0)->T_imu_world(), svoh::inverse(T_iref_world));
  svoh::store_rigid(T_icur_iref, pb.T_icur_iref);
  pb.alpha_init = alpha_init_;
  pb.beta_init = beta_init_;
  pb.prior = prior_;

  const int rc = svoh_sparse_align_batch(ctx_, &opt, 1, &pb, &last_);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_sparse_align_batch: ") + svoh_last_error_string(ctx_));
  if (last_.n_fts_to_track == 0) return 0;  // "no features to track" (sparse_img_align.cpp:53-57)

  // f->T_f_w_ = f->T_cam_imu() * state.T_icur_iref * T_iref_world_ (sparse_img_align.cpp:103-106)
  const Transformation T_opt = svoh::load_rigid(last_.T_icur_iref);
  for (const FramePtr& f : cur_frames->frames_) f->T_f_w_ = svoh::mul(svoh::mul(f->T_cam_imu(), T_opt), T_iref_world);
  alpha_init_ = 0.0;  // sparse_img_align.cpp:109-110
  beta_init_ = 0.0;
  return static_cast<size_t>(last_.n_fts_to_track);
}

// ---- DepthFilterHip -------------------------------------------------------------
DepthFilterHip::DepthFilterHip(svoh_ctx* ctx, const DepthFilterOptions& options) : ctx_(ctx), options_(options)
{
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

static svoh_frame_view view_of(const Frame& f)
{
  svoh_frame_view v{};
  v.frame = f.pyramid;
  v.cam = f.cam;
  svoh::store_rigid(f.T_f_w_, v.T_f_w);
  v.seed_mu_range = f.seed_mu_range_;
  v.id = f.id();
  return v;
}

size_t DepthFilterHip::updateSeeds(const std::vector<FramePtr>& ref_frames_with_seeds, const FramePtr& cur_frame)
{
  if (!cur_frame) throw std::runtime_error("DepthFilterHip::updateSeeds: NULL current frame");
  if (!have_px_error_angle_) {  // static double px_error_angle = cur_frame.getAngleError(1.0)
    px_error_angle_ = atan(1.0 / (2.0 * cur_frame->cam.fx)) + atan(1.0 / (2.0 * cur_frame->cam.fy));
    have_px_error_angle_ = true;
  }
  std::vector<svoh_frame_view> refs;
  std::vector<int32_t> ref_idx, level;
  std::vector<double> px, f, grad, state;
  std::vector<uint8_t> type;
  for (size_t k = 0; k < ref_frames_with_seeds.size(); ++k) {
    const Frame& r = *ref_frames_with_seeds[k];
    refs.push_back(view_of(r));
    const size_t n = r.num_features_;
    ref_idx.insert(ref_idx.end(), n, static_cast<int32_t>(k));
    px.insert(px.end(), r.px_vec_.begin(), r.px_vec_.begin() + 2 * n);
    f.insert(f.end(), r.f_vec_.begin(), r.f_vec_.begin() + 3 * n);
    grad.insert(grad.end(), r.grad_vec_.begin(), r.grad_vec_.begin() + 2 * n);
    level.insert(level.end(), r.level_vec_.begin(), r.level_vec_.begin() + n);
    type.insert(type.end(), r.type_vec_.begin(), r.type_vec_.begin() + n);
    state.insert(state.end(), r.invmu_sigma2_a_b_vec_.begin(), r.invmu_sigma2_a_b_vec_.begin() + 4 * n);
  }
  const size_t n_total = ref_idx.size();
  last_results_.assign(n_total, SVOH_MATCH_NOT_RUN);
  if (n_total == 0) return 0;
  svoh_feature_batch fb{};
  fb.n = static_cast<int32_t>(n_total);
  fb.ref_frame_idx = ref_idx.data(); fb.px = px.data(); fb.f = f.data(); fb.grad = grad.data();
  fb.level = level.data(); fb.type = type.data();
  svoh_depth_filter_options o{};
  o.seed_convergence_sigma2_thresh = options_.seed_convergence_sigma2_thresh;
  o.mappoint_convergence_sigma2_thresh = options_.mappoint_convergence_sigma2_thresh;
  o.px_error_angle = px_error_angle_;
  o.check_visibility = 1; o.check_convergence = 0; o.use_vogiatzis_update = 1;  // depth_filter.cpp:224-225
  const svoh_frame_view cur = view_of(*cur_frame);
  std::vector<uint8_t> success(n_total);
  int32_t n_success = 0;
  const int rc = svoh_update_seeds_batch(ctx_, &matcher_options_, &o, static_cast<int>(refs.size()), refs.data(), &cur, &fb,
                                         state.data(), success.data(), last_results_.data(), &n_success);
  if (rc != SVOH_OK) throw std::runtime_error(std::string("svoh_update_seeds_batch: ") + svoh_last_error_string(ctx_));
  // scatter back in place (ref_frame.invmu_sigma2_a_b_vec_.col(i), type_vec_[i])
  size_t off = 0;
  for (const FramePtr& rp : ref_frames_with_seeds) {
    Frame& r = *rp;
    const size_t n = r.num_features_;
    std::copy(state.begin() + 4 * off, state.begin() + 4 * (off + n), r.invmu_sigma2_a_b_vec_.begin());
    std::copy(type.begin() + off, type.begin() + off + n, r.type_vec_.begin());
    off += n;
  }
  return static_cast<size_t>(n_success);
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
