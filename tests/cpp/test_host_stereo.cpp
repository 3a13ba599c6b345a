// test_host_stereo.cpp -- StereoTriangulationHip::compute (mirror of svo::StereoTriangulation::compute,
// src/svo/src/stereo_triangulation.cpp:23-140, as FrameHandlerStereo::makeKeyframe calls it) against a sequential
// restatement with the oracle's detector and matcher: same new features, same visiting order, same successes,
// same landmarks and right-frame features, the early stop at n_desired.  Input: a stereo pair dumped by
// tests/test_host_cpp_gpu.py (left / right image of one synthetic scene, extrinsics, body pose).
#include <algorithm>
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
  std::vector<int32_t> hdr = rd<int32_t>(f, 4);  // w, h, triangulate_n_features, n_existing_landmarks
  const int w = hdr[0], h = hdr[1], n_want = hdr[2], n_have = hdr[3];
  std::vector<double> camv = rd<double>(f, 9), T_c0_b = rd<double>(f, 7), T_c1_b = rd<double>(f, 7), T_b_w = rd<double>(f, 7),
                      dinv = rd<double>(f, 3);
  std::vector<uint8_t> img0 = rd<uint8_t>(f, (size_t)w * h), img1 = rd<uint8_t>(f, (size_t)w * h);
  fclose(f);

  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }
  svoh_camera cam{};
  cam.fx = camv[0]; cam.fy = camv[1]; cam.cx = camv[2]; cam.cy = camv[3];
  for (int i = 0; i < 4; ++i) cam.d[i] = camv[4 + i];
  cam.distortion = camv[8] != 0.0 ? SVOH_DISTORTION_RADTAN : SVOH_DISTORTION_NONE;
  cam.width = w; cam.height = h;

  const int n_levels = 5;
  FramePtr f0(new Frame), f1(new Frame);
  CHECK(svoh_build_pyramid(ctx, img0.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, nullptr, &f0->pyramid) == SVOH_OK);
  CHECK(svoh_build_pyramid(ctx, img1.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, nullptr, &f1->pyramid) == SVOH_OK);
  f0->cam = cam; f1->cam = cam;
  f0->set_T_cam_imu(to_T(T_c0_b.data())); f1->set_T_cam_imu(to_T(T_c1_b.data()));
  f0->T_f_w_ = svoh::mul(f0->T_cam_imu(), to_T(T_b_w.data()));
  f1->T_f_w_ = svoh::mul(f1->T_cam_imu(), to_T(T_b_w.data()));
  f0->id_ = 3; f1->id_ = 4;
  // a few features that already carry landmarks (they count against triangulate_n_features and keep their slots)
  for (int k = 0; k < n_have; ++k) {
    const double px = 60.0 + 37.0 * k, py = 50.0 + 11.0 * k;
    f0->px_vec_.push_back(px); f0->px_vec_.push_back(py);
    const svoh::Vec3 b = svoh::back_project3(svoh::load_camera(cam), px, py);
    const double nn = sqrt(b.x * b.x + b.y * b.y + b.z * b.z);
    f0->f_vec_.push_back(b.x / nn); f0->f_vec_.push_back(b.y / nn); f0->f_vec_.push_back(b.z / nn);
    f0->grad_vec_.push_back(1.0); f0->grad_vec_.push_back(0.0);
    f0->score_vec_.push_back(50.0); f0->level_vec_.push_back(0); f0->type_vec_.push_back(SVOH_FT_CORNER);
    f0->landmark_vec_.push_back(std::make_shared<Point>()); f0->seed_ref_vec_.emplace_back(); f0->track_id_vec_.push_back(1000 + k);
    for (int j = 0; j < 4; ++j) f0->invmu_sigma2_a_b_vec_.push_back(0.0);
  }
  f0->num_features_ = (size_t)n_have;
  const size_t n_old = f0->num_features_;

  DetectorOptions dopt;
  dopt.detector_type = DetectorType::kFastGrad;
  dopt.cell_size = 32; dopt.max_level = 2; dopt.threshold_primary = 10.0; dopt.threshold_secondary = 100.0;
  auto det = std::make_shared<DetectorHip>(ctx, dopt, w, h);
  det->fillGridWithKeypoints(f0->px_vec_, f0->num_features_);   // FrameHandlerStereo marks the existing features
  StereoTriangulationOptions so;
  so.triangulate_n_features = (size_t)n_want;
  so.mean_depth_inv = dinv[0]; so.min_depth_inv = dinv[1]; so.max_depth_inv = dinv[2];
  StereoTriangulationHip stereo(ctx, so, det);
  // a reproducible order in place of rand(): reverse each part
  stereo.shuffle_ = [](std::vector<size_t>& idx, size_t n_corners) {
    std::reverse(idx.begin(), idx.begin() + (long)n_corners);
    std::reverse(idx.begin() + (long)n_corners, idx.end());
  };
  stereo.compute(f0, f1);
  const size_t n_new = f0->num_features_ - n_old;
  CHECK(n_new > 100);

  // ---- oracle: the same pair, the detector's output as the mirror stored it, the reference's loop ----
  std::vector<std::vector<uint8_t>> o0(n_levels), o1(n_levels);
  uint8_t* p0[SVOH_MAX_LEVELS]; uint8_t* p1[SVOH_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    o0[l].resize((size_t)(w >> l) * (h >> l)); o1[l].resize((size_t)(w >> l) * (h >> l));
    p0[l] = o0[l].data(); p1[l] = o1[l].data();
  }
  orc_create_img_pyramid(img0.data(), w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, p0);
  orc_create_img_pyramid(img1.data(), w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, p1);
  orc_frame_view v0, v1;
  memset(&v0, 0, sizeof v0); memset(&v1, 0, sizeof v1);
  v0.pyr.n_levels = v1.pyr.n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    v0.pyr.level[l] = orc_image{ o0[l].data(), w >> l, h >> l, w >> l, 0 };
    v1.pyr.level[l] = orc_image{ o1[l].data(), w >> l, h >> l, w >> l, 0 };
  }
  v0.cam = cam; v1.cam = cam;
  svoh::store_rigid(f0->T_f_w_, v0.T_f_w); svoh::store_rigid(f1->T_f_w_, v1.T_f_w);
  v0.id = 3; v1.id = 4;
  // the detector itself has its own parity tests (tests/test_detector_gpu.py); here its output is common input
  std::vector<int32_t> ridx(f0->num_features_, 0);
  orc_feature_batch fb;
  memset(&fb, 0, sizeof fb);
  fb.n = (int32_t)f0->num_features_; fb.ref_frame_idx = ridx.data(); fb.px = f0->px_vec_.data(); fb.f = f0->f_vec_.data();
  fb.grad = f0->grad_vec_.data(); fb.level = f0->level_vec_.data(); fb.type = f0->type_vec_.data();
  svoh_se3 T_f1f0;
  svoh::store_rigid(svoh::mul(f1->T_cam_imu(), f0->T_imu_cam()), T_f1f0);
  std::vector<int32_t> order(stereo.last_indices_.begin(), stereo.last_indices_.end());
  CHECK(order.size() == n_new);
  // corners come first and each part is reversed
  size_t n_corners = 0;
  for (size_t k = n_old; k < f0->num_features_; ++k) n_corners += f0->type_vec_[k] == SVOH_FT_CORNER;
  for (size_t k = 0; k < n_corners; ++k) CHECK(order[k] == (int32_t)(n_old + n_corners - 1 - k));
  std::vector<orc_stereo_match> om(n_new);
  std::vector<int32_t> ores(f0->num_features_);
  int o_failed = 0;
  const int n_desired = n_want - n_have;
  const int o_succ = orc_stereo_triangulate(&v0, &v1, &T_f1f0, &fb, (int)order.size(), order.data(), n_desired, dinv.data(), om.data(),
                                            ores.data(), &o_failed);
  printf("stereo: %zu new features, %d desired; mirror %zu ok / %zu failed, oracle %d ok / %d failed\n", n_new, n_desired,
         stereo.last_n_succeeded_, stereo.last_n_failed_, o_succ, o_failed);
  CHECK((int)stereo.last_n_succeeded_ == o_succ && (int)stereo.last_n_failed_ == o_failed);
  CHECK(o_succ > 20);
  for (size_t k = 0; k < n_new; ++k) CHECK(stereo.last_results_[k] == ores[n_old + k]);   // incl. -1 = not reached
  CHECK(f1->num_features_ == (size_t)o_succ);
  const Transformation T_w_c0 = svoh::inverse(f0->T_f_w_);
  double worst_px = 0, worst_pos = 0;
  for (int s = 0; s < o_succ; ++s) {
    const orc_stereo_match& m = om[s];
    const PointPtr& lm = f0->landmark_vec_[m.i_ref];
    CHECK(lm != nullptr && lm == f1->landmark_vec_[s] && lm->id() == s);
    CHECK(f0->track_id_vec_[m.i_ref] == s && f1->track_id_vec_[s] == s);
    CHECK(lm->obs_.size() == 2 && lm->obs_[0].keypoint_index_ == (size_t)m.i_ref && lm->obs_[1].keypoint_index_ == (size_t)s);
    CHECK(f1->type_vec_[s] == f0->type_vec_[m.i_ref] && f1->level_vec_[s] == f0->level_vec_[m.i_ref]);
    CHECK(f1->score_vec_[s] == f0->score_vec_[m.i_ref]);
    const svoh::Vec3 pc = { m.xyz_cam0[0], m.xyz_cam0[1], m.xyz_cam0[2] };
    const svoh::Vec3 pw = svoh::transform(T_w_c0, pc);
    worst_pos = fmax(worst_pos, fmax(fabs(pw.x - lm->pos_.x), fmax(fabs(pw.y - lm->pos_.y), fabs(pw.z - lm->pos_.z))) / m.depth);
    for (int j = 0; j < 2; ++j) worst_px = fmax(worst_px, fabs(f1->px_vec_[2 * s + j] - m.px[j]));
    for (int j = 0; j < 3; ++j) CHECK(fabs(f1->f_vec_[3 * s + j] - m.f[j]) < 1e-6);
    for (int j = 0; j < 2; ++j) CHECK(fabs(f1->grad_vec_[2 * s + j] - m.grad[j]) < 1e-9);
  }
  printf("worst matched-pixel difference %.2e px, worst landmark difference %.2e (relative to depth)\n", worst_px, worst_pos);
  CHECK(worst_px <= 1e-4 && worst_pos < 1e-9);
  // features that were not triangulated carry no landmark; the pre-existing ones are untouched
  size_t n_lm = 0;
  for (size_t k = n_old; k < f0->num_features_; ++k) n_lm += f0->landmark_vec_[k] != nullptr;
  CHECK(n_lm == (size_t)o_succ);
  for (int k = 0; k < n_have; ++k) CHECK(f0->track_id_vec_[k] == 1000 + k);
  // a second call has enough landmarks now when the target was reached: no effect
  if (o_succ == n_desired) {
    const size_t nf0 = f0->num_features_, nf1 = f1->num_features_;
    stereo.compute(f0, f1);
    CHECK(f0->num_features_ == nf0 && f1->num_features_ == nf1);
  }
  svoh_destroy(ctx);
  printf("PASS\n");
  return 0;
}
