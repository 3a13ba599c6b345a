// test_host_pose.cpp -- PoseOptimizerHip::run (mirror of PoseOptimizer::run, pose_optimizer.cpp:39-113) in the
// call shape of FrameHandlerBase::optimizePose (frame_handler_base.cpp:746-790) against orc_optimize_pose:
// landmarks and seed references resolved to 3-D points, T_f_w_ written back, outliers marked kOutlier.
// Input: a dump written by tests/test_host_cpp_gpu.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../oracle/svo_oracle.h"
#include "../../svo_pro_universal_amd/host/svo_hip_host.h"

using namespace svo_hip;

#define CHECK(cond)                                                            \
  do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); return 1; } } while (0)

template <class T>
static std::vector<T> rd(FILE* f, size_t n)
{
  std::vector<T> v(n);
  if (n && fread(v.data(), sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}
static Transformation to_T(const double* v) { Transformation T{ { v[0], v[1], v[2], v[3] }, { v[4], v[5], v[6] } }; return T; }

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  std::vector<int32_t> hdr = rd<int32_t>(f, 4);   // w, h, n, error_type
  const int w = hdr[0], h = hdr[1], n = hdr[2], et = hdr[3];
  std::vector<double> camv = rd<double>(f, 9), T_cam_imu = rd<double>(f, 7), T_imu_world = rd<double>(f, 7), T_kf_w = rd<double>(f, 7);
  std::vector<double> px = rd<double>(f, 2 * (size_t)n), fv = rd<double>(f, 3 * (size_t)n), grad = rd<double>(f, 2 * (size_t)n),
                      xyz = rd<double>(f, 3 * (size_t)n);
  std::vector<int32_t> level = rd<int32_t>(f, n);
  std::vector<uint8_t> type = rd<uint8_t>(f, n), usable = rd<uint8_t>(f, n);
  fclose(f);
  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }
  svoh_camera cam{};
  cam.fx = camv[0]; cam.fy = camv[1]; cam.cx = camv[2]; cam.cy = camv[3];
  for (int i = 0; i < 4; ++i) cam.d[i] = camv[4 + i];
  cam.distortion = camv[8] != 0.0 ? SVOH_DISTORTION_RADTAN : SVOH_DISTORTION_NONE;
  cam.width = w; cam.height = h;

  // keyframe holding the seeds that half of the features refer to
  FramePtr kf(new Frame);
  kf->cam = cam; kf->T_f_w_ = to_T(T_kf_w.data()); kf->id_ = 1;
  FramePtr fr(new Frame);
  fr->cam = cam; fr->id_ = 2;
  fr->set_T_cam_imu(to_T(T_cam_imu.data()));
  fr->T_f_w_ = svoh::mul(fr->T_cam_imu(), to_T(T_imu_world.data()));
  fr->num_features_ = (size_t)n;
  fr->px_vec_ = px; fr->f_vec_ = fv; fr->grad_vec_ = grad; fr->level_vec_ = level; fr->type_vec_ = type;
  fr->landmark_vec_.resize(n); fr->seed_ref_vec_.resize(n);
  for (int i = 0; i < n; ++i) {
    if (!usable[i]) { fr->type_vec_[i] = SVOH_FT_OUTLIER; continue; }   // neither landmark nor seed: skipped by the optimiser
    const svoh::Vec3 X{ xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
    const uint8_t t = type[i];
    const bool seed = t == SVOH_FT_CORNER_SEED_CONVERGED || t == SVOH_FT_EDGELET_SEED_CONVERGED;
    if (seed) {
      // express the point as a seed of the keyframe: bearing vector and inverse depth
      const svoh::Vec3 pk = svoh::transform(kf->T_f_w_, X);
      const double d = sqrt(pk.x * pk.x + pk.y * pk.y + pk.z * pk.z);
      const size_t k = kf->num_features_++;
      kf->f_vec_.insert(kf->f_vec_.end(), { pk.x / d, pk.y / d, pk.z / d });
      kf->invmu_sigma2_a_b_vec_.insert(kf->invmu_sigma2_a_b_vec_.end(), { 1.0 / d, 1.0, 10.0, 10.0 });
      fr->seed_ref_vec_[i].keyframe = kf; fr->seed_ref_vec_[i].seed_id = (int)k;
    } else {
      PointPtr p(new Point);
      p->pos_ = X;
      fr->landmark_vec_[i] = p;
    }
  }
  FrameBundle::Ptr bundle(new FrameBundle);
  bundle->frames_.push_back(fr);

  // oracle on the flat problem (points of seeds recomputed from the keyframe like the mirror does)
  svoh_pose_options o{};
  o.max_iter = 10; o.eps = 1e-6; o.error_type = et;
  const double thresh_px = 2.0;
  o.outlier_threshold = et == SVOH_POSE_ERR_UNIT_PLANE ? thresh_px / fabs(cam.fx)
                        : et == SVOH_POSE_ERR_BEARING_DIFF ? fabs(2 * sin(0.5 * (atan(thresh_px / (2.0 * cam.fx)) + atan(thresh_px / (2.0 * cam.fy))))) : thresh_px;
  o.R_prior[0] = 1.0;
  std::vector<double> oxyz(3 * (size_t)n, 0.0);
  std::vector<uint8_t> ousable(n, 0), oout(n, 0);
  for (int i = 0; i < n; ++i) {
    if (!usable[i]) continue;
    ousable[i] = 1;
    svoh::Vec3 X{ xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
    if (fr->seed_ref_vec_[i].keyframe) {
      const size_t k = (size_t)fr->seed_ref_vec_[i].seed_id;
      const double depth = 1.0 / kf->invmu_sigma2_a_b_vec_[4 * k];
      X = svoh::transform(svoh::inverse(kf->T_f_w_), svoh::Vec3{ kf->f_vec_[3 * k] * depth, kf->f_vec_[3 * k + 1] * depth, kf->f_vec_[3 * k + 2] * depth });
    }
    oxyz[3 * i] = X.x; oxyz[3 * i + 1] = X.y; oxyz[3 * i + 2] = X.z;
  }
  svoh_pose_problem pb{};
  pb.n_cams = 1;
  svoh::store_rigid(fr->T_imu_world(), pb.T_imu_world);
  pb.cams[0].cam = cam; svoh::store_rigid(fr->T_cam_imu(), pb.cams[0].T_cam_imu); pb.cams[0].n_features = n;
  pb.cams[0].px = px.data(); pb.cams[0].f = fv.data(); pb.cams[0].grad = grad.data(); pb.cams[0].level = level.data();
  std::vector<uint8_t> otype = fr->type_vec_;
  pb.cams[0].type = otype.data(); pb.cams[0].xyz_world = oxyz.data(); pb.cams[0].usable = ousable.data(); pb.cams[0].outlier = oout.data();
  svoh_pose_result ores;
  orc_optimize_pose(&o, &pb, &ores);

  // the reference's calls (frame_handler_base.cpp:757-764)
  PoseOptimizerHip pose_optimizer(ctx);
  pose_optimizer.setErrorType(et == 0 ? PoseOptimizerHip::ErrorType::kUnitPlane
                              : et == 1 ? PoseOptimizerHip::ErrorType::kBearingVectorDiff : PoseOptimizerHip::ErrorType::kImagePlane);
  const size_t n_final = pose_optimizer.run(bundle, thresh_px);

  CHECK((int)n_final == ores.n_meas - ores.n_deleted_edges - ores.n_deleted_corners);
  CHECK(pose_optimizer.iterCount() == (size_t)ores.iters && pose_optimizer.measurement_sigma_ == ores.measurement_sigma);
  const Transformation T_f_w_expected = svoh::mul(fr->T_cam_imu(), svoh::load_rigid(ores.T_imu_world));
  double worst = 0;
  worst = fmax(worst, fabs(fr->T_f_w_.q.w - T_f_w_expected.q.w)); worst = fmax(worst, fabs(fr->T_f_w_.q.x - T_f_w_expected.q.x));
  worst = fmax(worst, fabs(fr->T_f_w_.q.y - T_f_w_expected.q.y)); worst = fmax(worst, fabs(fr->T_f_w_.q.z - T_f_w_expected.q.z));
  worst = fmax(worst, fabs(fr->T_f_w_.t.x - T_f_w_expected.t.x)); worst = fmax(worst, fabs(fr->T_f_w_.t.y - T_f_w_expected.t.y));
  worst = fmax(worst, fabs(fr->T_f_w_.t.z - T_f_w_expected.t.z));
  CHECK(worst < 1e-9);
  int n_marked = 0;
  for (int i = 0; i < n; ++i) {
    if (!usable[i]) continue;
    CHECK((fr->type_vec_[i] == SVOH_FT_OUTLIER) == (oout[i] != 0));
    if (oout[i]) { CHECK(!fr->landmark_vec_[i] && !fr->seed_ref_vec_[i].keyframe); ++n_marked; }
    else CHECK(fr->type_vec_[i] == type[i] && (fr->landmark_vec_[i] || fr->seed_ref_vec_[i].keyframe));
  }
  printf("pose: %d measurements, %d iterations, %d outliers marked, |T_f_w - oracle| %.2e, median error %.3f -> %.3f px\n", ores.n_meas,
         ores.iters, n_marked, worst, pose_optimizer.stats_.reproj_error_before * (et == 0 ? 1.0 : (et == 1 ? cam.fx : 1.0)),
         pose_optimizer.stats_.reproj_error_after * (et == 0 ? 1.0 : (et == 1 ? cam.fx : 1.0)));
  CHECK(n_marked > 0 && n_marked == ores.n_deleted_edges + ores.n_deleted_corners);
  svoh_destroy(ctx);
  printf("PASS\n");
  return 0;
}
