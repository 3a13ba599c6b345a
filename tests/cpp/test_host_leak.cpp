// test_host_leak.cpp -- 500 frames through the host mirrors with the frame handles managed the way the adapter of
// INTEGRATION.md manages them (DeviceFrameCache: one device pyramid per live Frame object, released when the Frame
// is gone): the context's live-frame count and its device bytes must stay bounded (boundary hole:
// "as written a long sequence leaks one slab per frame").  The images are synthetic texture translated a little
// per frame; the point here is the lifetime bookkeeping, the numerics have their own tests.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <vector>

#include "../../svo_pro_universal_amd/host/svo_hip_host.h"

using namespace svo_hip;

#define CHECK(cond)                                                            \
  do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); return 1; } } while (0)

static void render(std::vector<uint8_t>& img, int w, int h, double shift)
{
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const double u = x + shift, v = y + 0.5 * shift;
      const double s = 128.0 + 50.0 * sin(0.11 * u) * cos(0.07 * v) + 40.0 * sin(0.023 * u + 0.031 * v) + 25.0 * cos(0.19 * v);
      img[(size_t)y * w + x] = (uint8_t)(s < 0 ? 0 : s > 255 ? 255 : s);
    }
}

int main(int argc, char** argv)
{
  const int n_frames = argc > 1 ? atoi(argv[1]) : 500;
  const int w = 640, h = 480, n_levels = 5, n_feat = 120, window = 6;   // a keyframe window like the map's
  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }
  svoh_camera cam{};
  cam.fx = cam.fy = 320.0; cam.cx = 320.0; cam.cy = 240.0; cam.distortion = SVOH_DISTORTION_NONE; cam.width = w; cam.height = h;
  const svoh::CamModel cm = svoh::load_camera(cam);

  {
    DeviceFrameCache cache(ctx);
    SparseImgAlignHip align(ctx, SparseImgAlignHip::getDefaultSolverOptions(), SparseImgAlignOptions());
    DepthFilterOptions dfo;
    DepthFilterHip depth_filter(ctx, dfo);
    std::deque<FramePtr> keyframes;     // the frames something still refers to
    FramePtr last;
    std::vector<uint8_t> img((size_t)w * h);
    svoh_context_stats_t st{}, st_warm{};
    int64_t max_live = 0;
    for (int k = 0; k < n_frames; ++k) {
      render(img, w, h, 0.4 * k);
      FramePtr f(new Frame);
      f->cam = cam; f->id_ = k;
      f->T_f_w_ = Transformation{ { 1, 0, 0, 0 }, { -0.0012 * k, -0.0006 * k, 0 } };
      // the adapter's deviceFrame(): pyramid built on first use, handle cached per Frame object
      f->pyramid = cache.get(f, [&](Frame&) {
        svoh_frame_t hdl = 0;
        if (svoh_build_pyramid(ctx, img.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, nullptr, &hdl) != SVOH_OK) {
          fprintf(stderr, "svoh_build_pyramid: %s\n", svoh_last_error_string(ctx)); exit(4);
        }
        return hdl;
      });
      // features with a landmark at 3 m (what the alignment needs) that double as seeds of the keyframes
      f->num_features_ = n_feat;
      for (int i = 0; i < n_feat; ++i) {
        const double px = 60.0 + (i % 12) * 45.0, py = 50.0 + (i / 12) * 40.0;
        f->px_vec_.push_back(px); f->px_vec_.push_back(py);
        svoh::Vec3 b = svoh::back_project3(cm, px, py);
        const double nn = sqrt(b.x * b.x + b.y * b.y + b.z * b.z);
        b.x /= nn; b.y /= nn; b.z /= nn;
        f->f_vec_.push_back(b.x); f->f_vec_.push_back(b.y); f->f_vec_.push_back(b.z);
        const svoh::Vec3 pw = svoh::transform(svoh::inverse(f->T_f_w_), svoh::Vec3{ 3.0 * b.x, 3.0 * b.y, 3.0 * b.z });
        f->pos_world_.push_back(pw.x); f->pos_world_.push_back(pw.y); f->pos_world_.push_back(pw.z);
        f->alignable_.push_back(1);
        f->grad_vec_.push_back(1.0); f->grad_vec_.push_back(0.0);
        f->level_vec_.push_back(i % 3); f->type_vec_.push_back(i % 4 == 0 ? SVOH_FT_EDGELET_SEED : SVOH_FT_CORNER_SEED);
        f->invmu_sigma2_a_b_vec_.push_back(1.0 / 3.2); f->invmu_sigma2_a_b_vec_.push_back(1.0 / 36.0);
        f->invmu_sigma2_a_b_vec_.push_back(10.0); f->invmu_sigma2_a_b_vec_.push_back(10.0);
      }
      f->seed_mu_range_ = 1.0;
      if (last) {
        FrameBundle::Ptr ref(new FrameBundle), cur(new FrameBundle);
        ref->frames_.push_back(last); cur->frames_.push_back(f);
        align.reset();
        CHECK(align.run(ref, cur) > 0);
        std::vector<FramePtr> kfs(keyframes.begin(), keyframes.end());
        if (!kfs.empty()) depth_filter.updateSeeds(kfs, f);
      }
      if (k % 5 == 0) { keyframes.push_back(f); if ((int)keyframes.size() > window) keyframes.pop_front(); }   // old keyframes die
      last = f;
      cache.sweep();
      CHECK(svoh_context_stats(ctx, &st) == SVOH_OK);
      if (st.live_frames > max_live) max_live = st.live_frames;
      if (k == 60) st_warm = st;
      if (k > 60) {
        CHECK(st.live_frames <= window + 2);                       // keyframe window + last + current
        CHECK(st.frame_bytes <= st_warm.frame_bytes + (1 << 20));   // no slab left behind
        CHECK(st.workspace_bytes <= st_warm.workspace_bytes + (4 << 20));
      }
    }
    printf("%d frames: at most %lld live device frames, %.1f MB of frame slabs and %.1f MB of workspaces at the end (warm: %.1f / %.1f MB), cache holds %zu\n",
           n_frames, (long long)max_live, st.frame_bytes / 1e6, st.workspace_bytes / 1e6, st_warm.frame_bytes / 1e6,
           st_warm.workspace_bytes / 1e6, cache.size());
    CHECK(max_live <= window + 2);
    keyframes.clear(); last.reset();
    CHECK(cache.sweep() >= 1);
    CHECK(svoh_context_stats(ctx, &st) == SVOH_OK);
    CHECK(st.live_frames == 0 && st.frame_bytes == 0);    // everything released once the frames are gone
  }
  svoh_destroy(ctx);
  printf("PASS\n");
  return 0;
}
