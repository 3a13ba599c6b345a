#include "svo_hip_host.h"

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

}  // namespace svo_hip
