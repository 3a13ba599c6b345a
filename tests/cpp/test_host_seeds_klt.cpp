// test_host_seeds_klt.cpp -- the C++ host mirrors DepthFilterHip::updateSeeds and
// feature_alignment::alignPyr2DVec against the CPU oracle, in the call shapes the
// reference uses (frame_handler_mono.cpp:125: depth_filter_->updateSeeds(overlap_kfs, lastFrame());
// feature_alignment.h:59-69).  Input: a dump written by tests/test_host_cpp_gpu.py.
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
  std::vector<int32_t> hdr = rd<int32_t>(f, 4);  // w, h, n_seeds, n_tracks
  const int w = hdr[0], h = hdr[1], ns = hdr[2], nt = hdr[3];
  std::vector<double> camv = rd<double>(f, 9), T_ref = rd<double>(f, 7), T_cur = rd<double>(f, 7), mu_range = rd<double>(f, 1);
  std::vector<double> px = rd<double>(f, 2 * (size_t)ns), fv = rd<double>(f, 3 * (size_t)ns), grad = rd<double>(f, 2 * (size_t)ns),
                      state = rd<double>(f, 4 * (size_t)ns);
  std::vector<int32_t> level = rd<int32_t>(f, ns);
  std::vector<uint8_t> type = rd<uint8_t>(f, ns);
  std::vector<int32_t> trk_ref = rd<int32_t>(f, 2 * (size_t)nt);
  std::vector<double> trk_cur = rd<double>(f, 2 * (size_t)nt);
  std::vector<uint8_t> img_ref = rd<uint8_t>(f, (size_t)w * h), img_cur = rd<uint8_t>(f, (size_t)w * h);
  fclose(f);

  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }
  svoh_camera cam{};
  cam.fx = camv[0]; cam.fy = camv[1]; cam.cx = camv[2]; cam.cy = camv[3];
  for (int i = 0; i < 4; ++i) cam.d[i] = camv[4 + i];
  cam.distortion = camv[8] != 0.0 ? SVOH_DISTORTION_RADTAN : SVOH_DISTORTION_NONE;
  cam.width = w; cam.height = h;

  const int n_levels = 5;
  FramePtr kf(new Frame), cur(new Frame);
  CHECK(svoh_build_pyramid(ctx, img_ref.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, nullptr, &kf->pyramid) == SVOH_OK);
  CHECK(svoh_build_pyramid(ctx, img_cur.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, nullptr, &cur->pyramid) == SVOH_OK);
  kf->cam = cam; cur->cam = cam;
  kf->T_f_w_ = to_T(T_ref.data()); cur->T_f_w_ = to_T(T_cur.data());
  kf->id_ = 7; cur->id_ = 8;
  kf->num_features_ = (size_t)ns;
  kf->px_vec_ = px; kf->f_vec_ = fv; kf->grad_vec_ = grad; kf->level_vec_ = level; kf->type_vec_ = type;
  kf->invmu_sigma2_a_b_vec_ = state; kf->seed_mu_range_ = mu_range[0];

  // ---- the reference's call: n = depth_filter_->updateSeeds({kf}, cur) ----
  DepthFilterOptions dfo;
  DepthFilterHip depth_filter(ctx, dfo);
  const size_t n_updated = depth_filter.updateSeeds({ kf }, cur);

  // ---- oracle ----
  std::vector<std::vector<uint8_t>> oref(n_levels), ocur(n_levels);
  uint8_t* rp[SVOH_MAX_LEVELS]; uint8_t* cp[SVOH_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    oref[l].resize((size_t)(w >> l) * (h >> l)); ocur[l].resize((size_t)(w >> l) * (h >> l));
    rp[l] = oref[l].data(); cp[l] = ocur[l].data();
  }
  orc_create_img_pyramid(img_ref.data(), w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, rp);
  orc_create_img_pyramid(img_cur.data(), w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, cp);
  orc_frame_view ov_r, ov_c;
  memset(&ov_r, 0, sizeof ov_r); memset(&ov_c, 0, sizeof ov_c);
  ov_r.pyr.n_levels = ov_c.pyr.n_levels = n_levels;
  for (int l = 0; l < n_levels; ++l) {
    ov_r.pyr.level[l] = orc_image{ oref[l].data(), w >> l, h >> l, w >> l, 0 };
    ov_c.pyr.level[l] = orc_image{ ocur[l].data(), w >> l, h >> l, w >> l, 0 };
  }
  ov_r.cam = cam; ov_c.cam = cam;
  svoh::store_rigid(kf->T_f_w_, ov_r.T_f_w); svoh::store_rigid(cur->T_f_w_, ov_c.T_f_w);
  ov_r.seed_mu_range = mu_range[0]; ov_r.id = 7; ov_c.id = 8;
  std::vector<int32_t> idx(ns, 0);
  std::vector<uint8_t> otype = type;
  std::vector<double> ostate = state;
  orc_feature_batch fb;
  memset(&fb, 0, sizeof fb);
  fb.n = ns; fb.ref_frame_idx = idx.data(); fb.px = px.data(); fb.f = fv.data(); fb.grad = grad.data();
  fb.level = level.data(); fb.type = otype.data();
  svoh_matcher_options mo = depth_filter.getMatcherOptions();
  svoh_depth_filter_options dopt{};
  dopt.seed_convergence_sigma2_thresh = 200; dopt.mappoint_convergence_sigma2_thresh = 500;
  dopt.px_error_angle = atan(1.0 / (2.0 * cam.fx)) + atan(1.0 / (2.0 * cam.fy));
  dopt.check_visibility = 1; dopt.use_vogiatzis_update = 1;
  std::vector<uint8_t> osucc(ns); std::vector<int32_t> omr(ns);
  const int on = orc_update_seeds_batch(&mo, &dopt, 1, &ov_r, &ov_c, &fb, ostate.data(), osucc.data(), omr.data());

  CHECK((int)n_updated == on);
  CHECK(kf->type_vec_ == otype);
  CHECK(depth_filter.lastMatchResults() == omr);
  double worst = 0;
  for (size_t i = 0; i < ostate.size(); ++i) {
    const double d = fabs(kf->invmu_sigma2_a_b_vec_[i] - ostate[i]) / fmax(1e-300, fabs(ostate[i]));
    if (d > worst) worst = d;
  }
  printf("seeds: %zu of %d updated, worst relative state difference %.3e\n", n_updated, ns, worst);
  CHECK(worst < 1e-9);

  // ---- round 4: the update queued BEFORE its frame's pose is final (DepthFilterHip::prepareUpdateSeeds, what the per-frame
  // chain does while the pose optimiser's kernel runs), the pose replaced just before the launch
  // (svoh_matcher_deferred_set_cur_frame): same states, types and result codes as the one-call update above ----
  {
    FramePtr kf2(new Frame);
    *kf2 = *kf;                               // (shares the pyramid handle; the seeds as they were before the update)
    kf2->invmu_sigma2_a_b_vec_ = state; kf2->type_vec_ = type;
    const Transformation T_final = cur->T_f_w_;
    cur->T_f_w_.t.x += 0.37; cur->T_f_w_.t.z -= 0.2;   // "not optimised yet": a pose the update must NOT be evaluated at
    DepthFilterHip df2(ctx, dfo);
    df2.prepareUpdateSeeds({ kf2 }, cur);
    cur->T_f_w_ = T_final;                    // the pose optimiser has written its result
    df2.updateSeedsAsync({ kf2 }, cur);
    const size_t n2 = df2.finishUpdateSeeds();
    CHECK(n2 == n_updated);
    CHECK(kf2->type_vec_ == kf->type_vec_);
    CHECK(kf2->invmu_sigma2_a_b_vec_ == kf->invmu_sigma2_a_b_vec_);   // bit for bit: the same kernel on the same inputs
    CHECK(df2.lastMatchResults() == depth_filter.lastMatchResults());
    // a prepared update that is never given its pose updates nothing
    FramePtr kf3(new Frame);
    *kf3 = *kf;
    kf3->invmu_sigma2_a_b_vec_ = state; kf3->type_vec_ = type;
    {
      DepthFilterHip df3(ctx, dfo);
      df3.prepareUpdateSeeds({ kf3 }, cur);
      CHECK(df3.finishUpdateSeeds() == 0);
    }
    CHECK(kf3->invmu_sigma2_a_b_vec_ == state && kf3->type_vec_ == type);
    // something else needs the context's deferred section between prepare and send-off (a reprojection on the same
    // context: finishPendingSeedUpdate): the prepared batch is dropped, and updateSeedsAsync then queues the update
    // afresh -- it used to throw "the previous update has not been finished", the frame's update lost
    FramePtr kf4(new Frame);
    *kf4 = *kf;
    kf4->invmu_sigma2_a_b_vec_ = state; kf4->type_vec_ = type;
    {
      DepthFilterHip df4(ctx, dfo);
      cur->T_f_w_.t.x += 0.37;
      df4.prepareUpdateSeeds({ kf4 }, cur);
      finishPendingSeedUpdate(ctx);
      CHECK(kf4->invmu_sigma2_a_b_vec_ == state && kf4->type_vec_ == type);   // nothing was written back
      cur->T_f_w_ = T_final;
      df4.updateSeedsAsync({ kf4 }, cur);
      CHECK(df4.finishUpdateSeeds() == n_updated);
      CHECK(kf4->invmu_sigma2_a_b_vec_ == kf->invmu_sigma2_a_b_vec_ && kf4->type_vec_ == kf->type_vec_);
    }
    kf2->pyramid = 0; kf3->pyramid = 0; kf4->pyramid = 0;       // the handle belongs to kf
    printf("seeds: the update queued ahead of its pose equals the one-call update\n");
  }

  // ---- alignPyr2DVec ----
  std::vector<Point2f> pr(nt), pc(nt);
  for (int i = 0; i < nt; ++i) { pr[i] = { (float)trk_ref[2 * i], (float)trk_ref[2 * i + 1] }; pc[i] = { (float)trk_cur[2 * i], (float)trk_cur[2 * i + 1] }; }
  std::vector<uint8_t> status;
  feature_alignment::alignPyr2DVec(ctx, kf->pyramid, cur->pyramid, 4, 0, { 16, 16, 16, 8, 8 }, 30, 0.001f, pr, pc, status);
  const int32_t ps[SVOH_MAX_LEVELS] = { 16, 16, 16, 8, 8, 8, 8, 8 };
  int n_ok = 0;
  for (int i = 0; i < nt; ++i) {
    // alignPyr2DVec: Keypoint px_cur_level_0(px_cur[i].x, px_cur[i].y) from floats; result narrowed to cv::Point2f
    double p[2] = { (double)(float)trk_cur[2 * i], (double)(float)trk_cur[2 * i + 1] };
    const int32_t r[2] = { trk_ref[2 * i], trk_ref[2 * i + 1] };
    const int ok = orc_align_pyr_2d(&ov_r.pyr, &ov_c.pyr, 4, 0, ps, 30, 0.001f, r, p);
    CHECK((status[i] != 0) == (ok != 0));
    CHECK(pc[i].x == (float)p[0] && pc[i].y == (float)p[1]);
    n_ok += ok;
  }
  printf("klt: %d of %d tracks converged, positions bit-identical\n", n_ok, nt);
  svoh_destroy(ctx);
  printf("PASS\n");
  return 0;
}
