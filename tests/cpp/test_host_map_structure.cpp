// test_host_map_structure.cpp -- host mirrors next to the reprojector and the pose optimiser
// (SURVEY.md 8(f-3) second half, 8(f-4) second half):
//   optimizeStructure  (FrameHandlerBase::optimizeStructure, frame_handler_base.cpp:779-825) -> Point::optimize on
//                      the device, against orc_optimize_point on the same observations;
//   Frame::setKeyPoints (frame.cpp:171-227) and Map::getOverlapKeyframes / getClosestNKeyframesWithOverlap /
//                      getClosestKeyframe / removeKeyframe (map.cpp:29-177) against brute force.
// Self-contained: builds its own geometry.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <string>
#include <vector>

#include "../../oracle/svo_oracle.h"
#include "../../svo_pro_universal_amd/host/svo_hip_host.h"

using namespace svo_hip;

#define CHECK(cond)                                                            \
  do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); return 1; } } while (0)

static uint64_t g_state = 88172645463325252ull;
static double urand() { g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17; return (double)(g_state >> 11) / 9007199254740992.0; }
static double urand(double a, double b) { return a + (b - a) * urand(); }

static svoh_camera make_cam()
{
  svoh_camera cam{};
  cam.fx = 458.654; cam.fy = 457.296; cam.cx = 367.215; cam.cy = 248.375;
  cam.distortion = SVOH_DISTORTION_NONE;
  cam.width = 752; cam.height = 480;
  return cam;
}

static Transformation small_pose(double rot, double trans)
{
  double v[6] = { urand(-trans, trans), urand(-trans, trans), urand(-trans, trans), urand(-rot, rot), urand(-rot, rot), urand(-rot, rot) };
  return svoh::rigid_exp(v);
}

static int test_structure(svoh_ctx* ctx)
{
  const svoh_camera cam = make_cam();
  const int n_kf = 4, n_pts = 240;
  std::vector<FramePtr> kfs;
  for (int k = 0; k < n_kf; ++k) {
    FramePtr f(new Frame);
    f->cam = cam; f->id_ = 10 + k;
    f->T_f_w_ = small_pose(0.15, 0.5);
    kfs.push_back(f);
  }
  FramePtr cur(new Frame);
  cur->cam = cam; cur->id_ = 77; cur->T_f_w_ = small_pose(0.1, 0.3);
  std::vector<PointPtr> pts;
  std::vector<svoh::Vec3> truth;
  for (int i = 0; i < n_pts; ++i) {
    PointPtr p(new Point);
    p->id_ = i;
    const svoh::Vec3 gt{ urand(-2, 2), urand(-1.5, 1.5), urand(4, 9) };
    truth.push_back(gt);
    p->pos_ = svoh::Vec3{ gt.x * urand(0.9, 1.1) + 0.02, gt.y * urand(0.9, 1.1) - 0.01, gt.z * urand(0.9, 1.1) };
    const int n_obs = (i % 29 == 3) ? 1 : 2 + (int)(urand() * (n_kf - 1));   // a few landmarks with one observation only
    for (int o = 0; o < n_obs; ++o) {
      const FramePtr& kf = kfs[(i + o) % n_kf];
      const svoh::Vec3 pf = svoh::transform(kf->T_f_w_, gt);
      const double nrm = std::sqrt(pf.x * pf.x + pf.y * pf.y + pf.z * pf.z);
      const size_t idx = kf->num_features_++;
      kf->f_vec_.push_back(pf.x / nrm + urand(-1e-3, 1e-3)); kf->f_vec_.push_back(pf.y / nrm + urand(-1e-3, 1e-3)); kf->f_vec_.push_back(pf.z / nrm);
      kf->px_vec_.push_back(0); kf->px_vec_.push_back(0);
      kf->type_vec_.push_back(SVOH_FT_CORNER);
      kf->landmark_vec_.push_back(p);
      p->obs_.push_back(Point::Obs{ kf, idx });
    }
    pts.push_back(p);
    cur->num_features_++;
    cur->px_vec_.push_back(0); cur->px_vec_.push_back(0);
    cur->f_vec_.push_back(0); cur->f_vec_.push_back(0); cur->f_vec_.push_back(1);
    cur->type_vec_.push_back(i % 5 == 4 ? SVOH_FT_EDGELET : SVOH_FT_CORNER);
    cur->landmark_vec_.push_back(i % 17 == 8 ? nullptr : p);
  }
  FrameBundle::Ptr bundle(new FrameBundle);
  bundle->frames_.push_back(cur);

  // expected: orc_optimize_point on the same observations
  const int max_iter = 5;
  std::vector<svoh::Vec3> expect(n_pts), before(n_pts);
  std::vector<bool> handled(n_pts, false);
  size_t n_expected = 0;
  for (int i = 0; i < n_pts; ++i) {
    before[i] = pts[i]->pos_;
    expect[i] = pts[i]->pos_;
    if (cur->landmark_vec_[i] == nullptr || cur->type_vec_[i] == SVOH_FT_EDGELET) continue;
    handled[i] = true;
    ++n_expected;
    std::vector<svoh_se3> Ts(pts[i]->obs_.size());
    std::vector<const svoh_se3*> Tp;
    std::vector<double> fs;
    for (size_t o = 0; o < pts[i]->obs_.size(); ++o) {
      const FramePtr f = pts[i]->obs_[o].frame.lock();
      svoh::store_rigid(f->T_f_w_, Ts[o]);
      Tp.push_back(&Ts[o]);
      for (int c = 0; c < 3; ++c) fs.push_back(f->f_vec_[3 * pts[i]->obs_[o].keypoint_index_ + c]);
    }
    double pos[3] = { pts[i]->pos_.x, pts[i]->pos_.y, pts[i]->pos_.z };
    orc_optimize_point(max_iter, 0, (int)Tp.size(), Tp.data(), fs.data(), pos);
    expect[i] = svoh::Vec3{ pos[0], pos[1], pos[2] };
  }

  CHECK(optimizeStructure(ctx, bundle, 0, max_iter) == 0);          // max_n_pts == 0: nothing
  for (int i = 0; i < n_pts; ++i) CHECK(pts[i]->pos_.x == before[i].x && pts[i]->last_structure_optim_ == 0);
  // max_n_pts > 0 reorders the candidates but the reference's loop still visits all of them
  const size_t n_done = optimizeStructure(ctx, bundle, 10, max_iter);
  CHECK(n_done == n_expected);
  double worst = 0, err0 = 0, err1 = 0;
  int n_moved = 0;
  for (int i = 0; i < n_pts; ++i) {
    const svoh::Vec3& p = pts[i]->pos_;
    const double d = std::fmax(std::fabs(p.x - expect[i].x), std::fmax(std::fabs(p.y - expect[i].y), std::fabs(p.z - expect[i].z)));
    worst = std::fmax(worst, d);
    if (!handled[i]) {
      CHECK(p.x == before[i].x && p.y == before[i].y && p.z == before[i].z);   // edgelets and non-landmarks untouched
      CHECK(pts[i]->last_structure_optim_ == 0);
      continue;
    }
    CHECK(pts[i]->last_structure_optim_ == 77);
    if (pts[i]->obs_.size() < 2) { CHECK(p.x == before[i].x); continue; }      // "less than two observations"
    ++n_moved;
    err0 += std::sqrt(std::pow(before[i].x - truth[i].x, 2) + std::pow(before[i].y - truth[i].y, 2) + std::pow(before[i].z - truth[i].z, 2));
    err1 += std::sqrt(std::pow(p.x - truth[i].x, 2) + std::pow(p.y - truth[i].y, 2) + std::pow(p.z - truth[i].z, 2));
  }
  printf("optimizeStructure: %zu landmarks, %d moved, |gpu - oracle| max %.3e, mean error to truth %.4f -> %.4f m\n", n_done, n_moved,
         worst, err0 / n_moved, err1 / n_moved);
  CHECK(worst < 1e-6);            // see tests/test_point_optimize_gpu.py for the tolerance
  CHECK(err1 < 0.5 * err0);
  return 0;
}

static int test_map()
{
  const svoh_camera cam = make_cam();
  // --- setKeyPoints on a hand-made layout (cu = 376, cv = 240) ---
  FramePtr f(new Frame);
  f->cam = cam; f->id_ = 1;
  const double layout[][2] = { { 380, 244 },   // 0 near the centre
                               { 700, 460 },   // 1 bottom right, far out
                               { 600, 400 },   // 2 bottom right, less far
                               { 720, 20 },    // 3 top right: (u-cu)(v-cv) < 0, the largest product wins -> the LEAST extreme
                               { 500, 200 },   // 4 top right, product -4960 > 3's -75680
                               { 30, 30 },     // 5 top left
                               { 300, 100 },   // 6 u >= cv=240 -> NOT a left-hand candidate (the reference compares u with cv)
                               { 100, 400 },   // 7 bottom left
                               { 10, 470 } };  // 8 bottom left: product (10-376)(470-240) more negative than 7's
  const int n = 9;
  for (int i = 0; i < n; ++i) {
    f->px_vec_.push_back(layout[i][0]); f->px_vec_.push_back(layout[i][1]);
    f->type_vec_.push_back(SVOH_FT_CORNER);
    PointPtr p(new Point);
    p->pos_ = svoh::Vec3{ (double)i, 0, 5 };
    f->landmark_vec_.push_back(p);
  }
  f->num_features_ = n;
  f->type_vec_[1] = SVOH_FT_OUTLIER;   // outliers are skipped
  f->setKeyPoints();
  CHECK(f->key_pts_[0].first == 0);
  CHECK(f->key_pts_[1].first == 2);    // 1 is an outlier
  CHECK(f->key_pts_[2].first == 4);
  CHECK(f->key_pts_[3].first == 5);
  CHECK(f->key_pts_[4].first == 7);
  CHECK(f->key_pts_[4].second.x == 7.0);
  f->resetKeyPoints();
  for (const Frame::KeyPoint& k : f->key_pts_) CHECK(k.first == -1);

  // --- keyframes on a line, all looking down +z at landmarks around z = 6; one looks the other way ---
  Map map;
  std::vector<FramePtr> kfs;
  const int n_kf = 12;
  for (int k = 0; k < n_kf; ++k) {
    FramePtr kf(new Frame);
    kf->cam = cam; kf->id_ = 100 + 7 * k;
    Transformation T_w_f{ { 1, 0, 0, 0 }, { 0.35 * k * ((k % 2) ? 1 : -1), 0.05 * k, 0.0 } };
    if (k == 5) T_w_f.q = svoh::Quat{ 0, 0, 1, 0 };   // rotated by pi about y: looks at -z, sees landmarks behind the others
    kf->T_f_w_ = svoh::inverse(T_w_f);
    for (int i = 0; i < 5; ++i) {
      // landmarks in front of the keyframe's own camera
      const svoh::Vec3 pc{ urand(-1, 1), urand(-0.7, 0.7), 6.0 };
      PointPtr p(new Point);
      p->pos_ = svoh::transform(T_w_f, pc);
      double uv[2];
      CHECK(kf->isVisible(p->pos_, uv));
      kf->px_vec_.push_back(uv[0]); kf->px_vec_.push_back(uv[1]);
      kf->type_vec_.push_back(SVOH_FT_CORNER);
      kf->landmark_vec_.push_back(p);
      p->obs_.push_back(Point::Obs{ kf, (size_t)i });
    }
    kf->num_features_ = 5;
    kf->setKeyPoints();
    map.addKeyframe(kf, true);
    kfs.push_back(kf);
  }
  CHECK(map.size() == (size_t)n_kf && map.last_added_kf_id_ == 100 + 7 * (n_kf - 1));
  FramePtr cur(new Frame);
  cur->cam = cam; cur->id_ = 999;
  cur->T_f_w_ = svoh::inverse(Transformation{ { 1, 0, 0, 0 }, { 0.1, 0.0, 0.0 } });

  std::vector<std::pair<FramePtr, double>> overlap;
  map.getOverlapKeyframes(cur, &overlap);
  // brute force: a keyframe overlaps iff one of its key points is visible in cur
  std::set<int> expect_ids;
  for (const FramePtr& kf : kfs) {
    bool vis = false;
    for (const Frame::KeyPoint& kp : kf->key_pts_) vis = vis || (kp.first != -1 && cur->isVisible(kp.second, nullptr));
    if (vis) expect_ids.insert(kf->id_);
  }
  std::set<int> got_ids;
  for (const auto& p : overlap) {
    got_ids.insert(p.first->id_);
    const svoh::Vec3 a = cur->T_f_w_.t, b = p.first->T_f_w_.t;
    CHECK(std::fabs(p.second - std::sqrt(std::pow(a.x - b.x, 2) + std::pow(a.y - b.y, 2) + std::pow(a.z - b.z, 2))) < 1e-15);
  }
  CHECK(got_ids == expect_ids);
  CHECK(!expect_ids.count(kfs[5]->id_));            // the one looking away does not overlap
  CHECK(expect_ids.size() >= 6 && expect_ids.size() < (size_t)n_kf);

  for (size_t N : { (size_t)1, (size_t)5, (size_t)100 }) {
    std::vector<FramePtr> close;
    map.getClosestNKeyframesWithOverlap(cur, N, &close);
    CHECK(close.size() == std::min(N, overlap.size()));
    std::vector<double> all;
    for (const auto& p : overlap) all.push_back(p.second);
    std::sort(all.begin(), all.end());
    double far_in = 0;
    for (const FramePtr& c : close) {
      const svoh::Vec3 a = cur->T_f_w_.t, b = c->T_f_w_.t;
      far_in = std::fmax(far_in, std::sqrt(std::pow(a.x - b.x, 2) + std::pow(a.y - b.y, 2) + std::pow(a.z - b.z, 2)));
    }
    CHECK(std::fabs(far_in - all[close.size() - 1]) < 1e-15);   // exactly the N smallest distances
  }
  // getClosestKeyframe never returns the query frame itself
  const FramePtr closest = map.getClosestKeyframe(kfs[0]);
  CHECK(closest && closest != kfs[0]);
  CHECK(map.getKeyframeById(kfs[3]->id_) == kfs[3] && map.getKeyframeById(5) == nullptr);
  std::vector<FramePtr> sorted;
  map.getSortedKeyframes(sorted);
  for (size_t i = 1; i < sorted.size(); ++i) CHECK(sorted[i - 1]->id_ < sorted[i]->id_);
  CHECK(map.getFurthestKeyframe(cur->pos()) == kfs[n_kf - 1]);
  // removeKeyframe drops the frame and its observations
  const PointPtr lm = kfs[2]->landmark_vec_[0];
  CHECK(lm->obs_.size() == 1);
  map.removeKeyframe(kfs[2]->id_);
  CHECK(map.size() == (size_t)n_kf - 1 && lm->obs_.empty());
  map.removeKeyframe(12345);   // unknown id: nothing happens
  CHECK(map.size() == (size_t)n_kf - 1);
  // an empty map gives nothing
  map.reset();
  std::vector<FramePtr> none;
  map.getClosestNKeyframesWithOverlap(cur, 5, &none);
  CHECK(none.empty() && map.getClosestKeyframe(cur) == nullptr);
  printf("map: %zu of %d keyframes overlap the query frame\n", expect_ids.size(), n_kf);
  return 0;
}

int main(int argc, char** argv)
{
  if (test_map()) return 1;   // host logic only
  if (argc > 1 && std::string(argv[1]) == "--map-only") { printf("PASS\n"); return 0; }
  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }
  const int rc = test_structure(ctx);
  svoh_destroy(ctx);
  if (rc) return rc;
  printf("PASS\n");
  return 0;
}
