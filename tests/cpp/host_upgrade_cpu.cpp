// svo_hip::upgradeSeedsToFeatures (host half) and removeObservationsOf against what FrameHandlerBase::upgradeSeedsToFeatures
// (frame_handler_base.cpp:828-920) and Map::removeKeyframe do, on a hand-made keyframe / frame pair: points at the seeds' positions
// (T_world_cam * f * depth), types, observations, track ids, cleared seed references, the list of upgraded edgelets, a seed that two
// features hang on (one point), a feature that has a landmark already (one more observation).  No GPU call.  Prints "ok" or the first
// difference.
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>
#include "../../svo_pro_universal_amd/host/svo_hip_host.h"
using namespace svo_hip;
#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)
static FramePtr make_frame(size_t n, int id)
{
  auto f = std::make_shared<Frame>();
  f->num_features_ = n; f->id_ = id;
  f->px_vec_.assign(2 * n, 0.0); f->f_vec_.assign(3 * n, 0.0); f->grad_vec_.assign(2 * n, 0.0); f->level_vec_.assign(n, 0);
  f->type_vec_.assign(n, SVOH_FT_OUTLIER); f->invmu_sigma2_a_b_vec_.assign(4 * n, 0.0);
  f->landmark_vec_.assign(n, nullptr); f->seed_ref_vec_.assign(n, Frame::SeedRef()); f->track_id_vec_.assign(n, -1);
  return f;
}
int main()
{
  FramePtr kf = make_frame(5, 10), fr = make_frame(6, 17);
  kf->T_f_w_ = Transformation{ { 0.9238795325112867, 0.0, 0.3826834323650898, 0.0 }, { 0.1, -0.2, 0.3 } };   // 45 degrees about y
  const uint8_t kt[5] = { SVOH_FT_CORNER_SEED, SVOH_FT_EDGELET_SEED_CONVERGED, SVOH_FT_MAPPOINT_SEED, SVOH_FT_CORNER_SEED_CONVERGED, SVOH_FT_CORNER_SEED };
  for (size_t i = 0; i < 5; ++i) {
    kf->type_vec_[i] = kt[i];
    const double x = 0.1 * (double)i - 0.2, y = 0.05 * (double)i, nn = std::sqrt(x * x + y * y + 1.0);
    kf->f_vec_[3 * i] = x / nn; kf->f_vec_[3 * i + 1] = y / nn; kf->f_vec_[3 * i + 2] = 1.0 / nn;
    kf->invmu_sigma2_a_b_vec_[4 * i] = 1.0 / (2.0 + 0.5 * (double)i);   // inverse depth
  }
  // fr: 0 -> seed 0 (corner), 1 -> seed 1 (edgelet), 2 -> seed 2 (map point), 3 -> seed 0 again, 4: a landmark of its own, 5: nothing
  const int ref[6] = { 0, 1, 2, 0, -1, -1 };
  for (size_t i = 0; i < 6; ++i) {
    fr->type_vec_[i] = SVOH_FT_CORNER_SEED;
    if (ref[i] >= 0) { fr->seed_ref_vec_[i].keyframe = kf; fr->seed_ref_vec_[i].seed_id = ref[i]; }
  }
  auto old_pt = std::make_shared<Point>();
  old_pt->id_ = 99; old_pt->pos_ = svoh::Vec3{ 1, 2, 3 };
  fr->landmark_vec_[4] = old_pt; fr->type_vec_[4] = SVOH_FT_CORNER;
  int next_id = 7;
  std::vector<size_t> edgelets;
  const size_t n = upgradeSeedsToFeatures(fr, &next_id, &edgelets);
  CHECK(n == 4 && next_id == 10);                              // four features upgraded, three new points (seed 0 once)
  CHECK(edgelets.size() == 1 && edgelets[0] == 1);
  CHECK(fr->landmark_vec_[0] && fr->landmark_vec_[0] == fr->landmark_vec_[3] && fr->landmark_vec_[0] == kf->landmark_vec_[0]);
  CHECK(fr->landmark_vec_[1] == kf->landmark_vec_[1] && fr->landmark_vec_[2] == kf->landmark_vec_[2] && !kf->landmark_vec_[3] && !kf->landmark_vec_[4] && !fr->landmark_vec_[5]);
  CHECK(kf->type_vec_[0] == SVOH_FT_CORNER && fr->type_vec_[0] == SVOH_FT_CORNER && fr->type_vec_[3] == SVOH_FT_CORNER);
  CHECK(kf->type_vec_[1] == SVOH_FT_EDGELET && fr->type_vec_[1] == SVOH_FT_EDGELET && kf->type_vec_[2] == SVOH_FT_MAPPOINT && fr->type_vec_[2] == SVOH_FT_MAPPOINT);
  CHECK(kf->type_vec_[3] == SVOH_FT_CORNER_SEED_CONVERGED && kf->type_vec_[4] == SVOH_FT_CORNER_SEED && fr->type_vec_[5] == SVOH_FT_CORNER_SEED);
  for (size_t i = 0; i < 4; ++i) CHECK(!fr->seed_ref_vec_[i].keyframe && fr->seed_ref_vec_[i].seed_id == -1);
  CHECK(fr->track_id_vec_[0] == fr->landmark_vec_[0]->id() && kf->track_id_vec_[0] == fr->landmark_vec_[0]->id() && fr->landmark_vec_[0]->id() == 7);
  // positions: T_world_cam * (f * depth)
  for (int sid = 0; sid < 3; ++sid) {
    const double depth = 1.0 / kf->invmu_sigma2_a_b_vec_[4 * (size_t)sid];
    const svoh::Vec3 want = svoh::transform(svoh::inverse(kf->T_f_w_), svoh::Vec3{ kf->f_vec_[3 * sid] * depth, kf->f_vec_[3 * sid + 1] * depth, kf->f_vec_[3 * sid + 2] * depth });
    const svoh::Vec3 got = kf->landmark_vec_[(size_t)sid]->pos_;
    CHECK(got.x == want.x && got.y == want.y && got.z == want.z);
  }
  // observations: seed 0's point: keyframe, feature 0, feature 3; the old landmark: one more (this frame)
  const Point& p0 = *kf->landmark_vec_[0];
  CHECK(p0.obs_.size() == 3 && p0.obs_[0].frame.lock() == kf && p0.obs_[0].keypoint_index_ == 0 && p0.obs_[1].frame.lock() == fr && p0.obs_[1].keypoint_index_ == 0 &&
        p0.obs_[2].keypoint_index_ == 3);
  CHECK(old_pt->obs_.size() == 1 && old_pt->obs_[0].frame.lock() == fr && old_pt->obs_[0].keypoint_index_ == 4);
  // a second keyframe step on the same frame adds observations only (nothing hangs on a seed any more)
  std::vector<size_t> e2;
  CHECK(upgradeSeedsToFeatures(fr, &next_id, &e2) == 0 && e2.empty() && next_id == 10 && p0.obs_.size() == 5);
  // the keyframe leaves the map: its observations go, the others stay
  removeObservationsOf(*kf);
  CHECK(p0.obs_.size() == 4);
  for (const Point::Obs& o : p0.obs_) CHECK(o.frame.lock() == fr);
  printf("ok\n");
  return 0;
}
