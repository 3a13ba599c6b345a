// test_host_tracker.cpp -- FeatureTrackerHip::trackFrameBundle (one batched KLT launch for the whole bundle +
// the reference's track bookkeeping) against the per-track loop of FeatureTracker::trackFrameBundle
// (src/svo_tracker/src/feature_tracker.cpp:52-122) restated here with the oracle's alignPyr2D.
// Input: a dump written by tests/test_host_cpp_gpu.py (2 cameras x 3 time steps).
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

struct OrcPyr {
  std::vector<std::vector<uint8_t>> lv;
  orc_pyramid pyr;
  void build(const uint8_t* img, int w, int h, int n_levels)
  {
    lv.resize(n_levels);
    uint8_t* p[SVOH_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) { lv[l].resize((size_t)(w >> l) * (h >> l)); p[l] = lv[l].data(); }
    orc_create_img_pyramid(img, w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, p);
    memset(&pyr, 0, sizeof pyr);
    pyr.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) pyr.level[l] = orc_image{ lv[l].data(), w >> l, h >> l, w >> l, 0 };
  }
};

struct ExpTrack { int id; int first_frame_px[2]; double last_px[2]; int n_obs; };

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  std::vector<int32_t> hdr = rd<int32_t>(f, 4);  // w, h, n_tracks per camera, template_is_first_observation
  const int w = hdr[0], h = hdr[1], n = hdr[2];
  const bool first_obs = hdr[3] != 0;
  std::vector<double> camv = rd<double>(f, 9);
  std::vector<int32_t> px0 = rd<int32_t>(f, 2 * 2 * (size_t)n);         // camera-major: cam 0 tracks, cam 1 tracks
  std::vector<uint8_t> imgs = rd<uint8_t>(f, (size_t)6 * w * h);        // [t][cam]
  fclose(f);
  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }
  svoh_camera cam{};
  cam.fx = camv[0]; cam.fy = camv[1]; cam.cx = camv[2]; cam.cy = camv[3];
  for (int i = 0; i < 4; ++i) cam.d[i] = camv[4 + i];
  cam.distortion = camv[8] != 0.0 ? SVOH_DISTORTION_RADTAN : SVOH_DISTORTION_NONE;
  cam.width = w; cam.height = h;
  const int n_levels = 5;

  FrameBundle::Ptr bundles[3];
  OrcPyr opyr[3][2];
  for (int t = 0; t < 3; ++t) {
    bundles[t].reset(new FrameBundle);
    for (int c = 0; c < 2; ++c) {
      const uint8_t* img = imgs.data() + ((size_t)t * 2 + c) * w * h;
      FramePtr fr(new Frame);
      CHECK(svoh_build_pyramid(ctx, img, w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, nullptr, &fr->pyramid) == SVOH_OK);
      fr->cam = cam; fr->id_ = 10 * t + c;
      bundles[t]->frames_.push_back(fr);
      opyr[t][c].build(img, w, h, n_levels);
    }
  }
  // features of the first bundle (a detector's integer positions, feature_tracker.cpp:150-166)
  for (int c = 0; c < 2; ++c) {
    Frame& fr = *bundles[0]->frames_[c];
    fr.num_features_ = (size_t)n;
    for (int i = 0; i < n; ++i) {
      fr.px_vec_.push_back(px0[2 * ((size_t)c * n + i)]); fr.px_vec_.push_back(px0[2 * ((size_t)c * n + i) + 1]);
      fr.score_vec_.push_back(100.0 + i);
    }
  }
  FeatureTrackerOptions opt;
  opt.klt_template_is_first_observation = first_obs;
  FeatureTrackerHip tracker(ctx, opt, 2);
  CHECK(tracker.initializeNewTracks(bundles[0], { 0, 0 }) == (size_t)(2 * n));

  // expected state, advanced with the oracle's alignPyr2D track by track
  std::vector<std::vector<ExpTrack>> exp(2);
  std::vector<std::vector<int>> ref_t(2);   // time step of the template of every active track
  for (int c = 0; c < 2; ++c)
    for (int i = 0; i < n; ++i) {
      const int x = px0[2 * ((size_t)c * n + i)], y = px0[2 * ((size_t)c * n + i) + 1];
      exp[c].push_back(ExpTrack{ c * n + i, { x, y }, { (double)x, (double)y }, 1 });
    }
  const int32_t ps[SVOH_MAX_LEVELS] = { 16, 16, 16, 8, 8, 8, 8, 8 };
  size_t total_terminated = 0;
  for (int t = 1; t < 3; ++t) {
    const size_t n_active = tracker.trackFrameBundle(bundles[t]);
    size_t exp_active = 0;
    for (int c = 0; c < 2; ++c) {
      std::vector<ExpTrack> kept;
      std::vector<double> e_px, e_f;
      std::vector<int> e_ids;
      size_t n_term = 0;
      for (ExpTrack& tr : exp[c]) {
        // template: first observation (time 0, the detector position) or the last one (time t-1)
        int32_t r[2];
        const orc_pyramid* rp;
        if (first_obs) { r[0] = tr.first_frame_px[0]; r[1] = tr.first_frame_px[1]; rp = &opyr[0][c].pyr; }
        else { r[0] = (int32_t)tr.last_px[0]; r[1] = (int32_t)tr.last_px[1]; rp = &opyr[t - 1][c].pyr; }
        double p[2] = { tr.last_px[0], tr.last_px[1] };
        const int ok = orc_align_pyr_2d(rp, &opyr[t][c].pyr, 4, 0, ps, 30, 0.001f, r, p);
        if (ok) {
          tr.last_px[0] = p[0]; tr.last_px[1] = p[1]; ++tr.n_obs;
          e_px.push_back(p[0]); e_px.push_back(p[1]); e_ids.push_back(tr.id);
          double fv[3];
          orc_back_project3(&cam, p, fv);
          const double nn = sqrt(fv[0] * fv[0] + fv[1] * fv[1] + fv[2] * fv[2]);
          for (int j = 0; j < 3; ++j) e_f.push_back(fv[j] / nn);
          kept.push_back(tr);
        } else {
          ++n_term;
        }
      }
      exp[c].swap(kept);
      exp_active += exp[c].size();
      const Frame& fr = *bundles[t]->frames_[c];
      CHECK(fr.num_features_ == exp[c].size());
      CHECK(fr.px_vec_ == e_px);                         // bit-identical positions, same order
      CHECK(fr.track_id_vec_ == e_ids);
      for (size_t k = 0; k < e_f.size(); ++k) CHECK(fabs(fr.f_vec_[k] - e_f[k]) < 1e-14);
      CHECK(tracker.getTerminatedTracks(c).size() == n_term);
      total_terminated += n_term;
      const FeatureTracks& act = tracker.getActiveTracks(c);
      CHECK(act.size() == exp[c].size());
      for (size_t k = 0; k < act.size(); ++k) {
        CHECK(act[k].getTrackId() == exp[c][k].id && (int)act[k].size() == exp[c][k].n_obs);
        CHECK(act[k].back().getFrame() == bundles[t]->frames_[c] && act[k].back().getFeatureIndex() == k);
        CHECK(act[k].front().getFrame() == bundles[0]->frames_[c]);
      }
    }
    CHECK(n_active == exp_active);
    printf("tracker: step %d: %zu active tracks (bit-identical positions), %zu terminated so far\n", t, n_active, total_terminated);
  }
  CHECK(total_terminated > 0 && tracker.getTotalActiveTracks() > (size_t)n);

  // ---- AbstractDetector::detect(frame) on the last frame, cells of the tracked features marked occupied
  //      (FeatureTracker::initializeNewTracks, feature_tracker.cpp:135-166) ----
  {
    DetectorOptions dopt;
    dopt.detector_type = DetectorType::kFastGrad;
    DetectorHip detector(ctx, dopt, w, h);
    FramePtr fr = bundles[2]->frames_[0];
    const size_t n_old = fr->num_features_;
    std::vector<uint8_t> occ(detector.grid_.size(), 0);
    for (size_t i = 0; i < n_old; ++i) {   // grid_.fillWithKeypoints(frame->px_vec_)
      const size_t k = detector.grid_.getCellIndex((int)fr->px_vec_[2 * i], (int)fr->px_vec_[2 * i + 1], 1);
      detector.grid_.setOccupied(k);
      occ[k] = 1;
    }
    fr->grad_vec_.assign(2 * n_old, 0.0); fr->level_vec_.assign(n_old, 0); fr->type_vec_.assign(n_old, SVOH_FT_CORNER);
    detector.detect(fr);
    svoh_detector_options o{};
    o.cell_size = 30; o.max_level = 2; o.min_level = 0; o.border = 8; o.detect_edgelets = 1;
    o.threshold_primary = 10.0; o.threshold_secondary = 100.0;
    const size_t n_cells = detector.grid_.size();
    std::vector<double> opx(2 * n_cells), osc(n_cells), ogr(2 * n_cells);
    std::vector<int32_t> olv(n_cells);
    std::vector<uint8_t> oty(n_cells);
    const int on = orc_detect_features(&opyr[2][0].pyr, &o, occ.data(), nullptr, 0, (int)n_cells, opx.data(), osc.data(), olv.data(),
                                       ogr.data(), oty.data());
    CHECK(fr->num_features_ == n_old + (size_t)on && on > 20);
    for (int i = 0; i < on; ++i) {
      const size_t s = n_old + (size_t)i;
      CHECK(fr->px_vec_[2 * s] == opx[2 * i] && fr->px_vec_[2 * s + 1] == opx[2 * i + 1]);
      CHECK(fr->score_vec_[s] == osc[i] && fr->level_vec_[s] == olv[i] && fr->type_vec_[s] == oty[i]);
      CHECK(fabs(fr->f_vec_[3 * s] * fr->f_vec_[3 * s] + fr->f_vec_[3 * s + 1] * fr->f_vec_[3 * s + 1] + fr->f_vec_[3 * s + 2] * fr->f_vec_[3 * s + 2] - 1.0) < 1e-12);
    }
    CHECK(detector.grid_.numOccupied() == 0);   // resetGrid() at the end of detect
    printf("detector: %d new features next to %zu tracked ones, identical to the oracle\n", on, n_old);
  }
  svoh_destroy(ctx);
  printf("PASS\n");
  return 0;
}
