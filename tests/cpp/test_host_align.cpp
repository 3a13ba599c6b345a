// test_host_align.cpp -- the C++ host layer (SparseImgAlignHip, mirror of the
// reference's SparseImgAlign interface) against the CPU oracle, in the shape the
// reference's caller uses it (frame_handler_base.cpp:621-634):
//     reset(); [setWeightedPrior(...)]; setMaxNumFeaturesToAlign(n); run(last_frames, new_frames)
// Input: a scene dump written by tests/test_host_cpp_gpu.py.  Exit code 0 = pass.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "../../oracle/svo_oracle.h"
#include "../../svo_pro_universal_amd/host/svo_hip_host.h"

using namespace svo_hip;

static std::vector<double> read_doubles(FILE* f, size_t n)
{
  std::vector<double> v(n);
  if (fread(v.data(), sizeof(double), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
  return v;
}

static Transformation to_T(const double* v) { Transformation T{ { v[0], v[1], v[2], v[3] }, { v[4], v[5], v[6] } }; return T; }

#define CHECK(cond)                                                            \
  do { if (!(cond)) { fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv)
{
  if (argc < 2) { fprintf(stderr, "usage: %s scene.bin\n", argv[0]); return 2; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  int32_t hdr[4];  // width, height, n_features, use_prior
  CHECK(fread(hdr, sizeof(int32_t), 4, f) == 4);
  const int w = hdr[0], h = hdr[1], n = hdr[2];
  std::vector<double> camv = read_doubles(f, 9);  // fx fy cx cy k1 k2 p1 p2 has_dist
  std::vector<double> T_cam_imu = read_doubles(f, 7), T_ref_f_w = read_doubles(f, 7), T_cur_init_f_w = read_doubles(f, 7);
  std::vector<double> px = read_doubles(f, 2 * (size_t)n), fv = read_doubles(f, 3 * (size_t)n), pw = read_doubles(f, 3 * (size_t)n);
  std::vector<uint8_t> flags(n), img_ref((size_t)w * h), img_cur((size_t)w * h);
  CHECK(fread(flags.data(), 1, n, f) == (size_t)n);
  CHECK(fread(img_ref.data(), 1, img_ref.size(), f) == img_ref.size());
  CHECK(fread(img_cur.data(), 1, img_cur.size(), f) == img_cur.size());
  fclose(f);

  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }

  svoh_camera cam{};
  cam.fx = camv[0]; cam.fy = camv[1]; cam.cx = camv[2]; cam.cy = camv[3];
  for (int i = 0; i < 4; ++i) cam.d[i] = camv[4 + i];
  cam.distortion = camv[8] != 0.0 ? SVOH_DISTORTION_RADTAN : SVOH_DISTORTION_NONE;
  cam.width = w; cam.height = h;

  // frames: pyramid built on the device, host copy kept bit-identical (Seam 4 of SURVEY 8b)
  const int n_levels = 5;
  std::vector<std::vector<uint8_t>> ref_lv(n_levels), cur_lv(n_levels);
  uint8_t* ref_ptr[SVOH_MAX_LEVELS]; uint8_t* cur_ptr[SVOH_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    ref_lv[l].resize((size_t)(w >> l) * (h >> l)); cur_lv[l].resize((size_t)(w >> l) * (h >> l));
    ref_ptr[l] = ref_lv[l].data(); cur_ptr[l] = cur_lv[l].data();
  }
  FramePtr ref(new Frame), cur(new Frame);
  CHECK(svoh_build_pyramid(ctx, img_ref.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, ref_ptr, &ref->pyramid) == SVOH_OK);
  CHECK(svoh_build_pyramid(ctx, img_cur.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, cur_ptr, &cur->pyramid) == SVOH_OK);
  ref->cam = cam; cur->cam = cam;
  ref->set_T_cam_imu(to_T(T_cam_imu.data())); cur->set_T_cam_imu(to_T(T_cam_imu.data()));
  ref->T_f_w_ = to_T(T_ref_f_w.data());
  cur->T_f_w_ = to_T(T_cur_init_f_w.data());
  ref->num_features_ = (size_t)n;
  ref->px_vec_ = px; ref->f_vec_ = fv; ref->pos_world_ = pw; ref->alignable_ = flags;

  FrameBundle::Ptr last_frames(new FrameBundle), new_frames(new FrameBundle);
  last_frames->frames_.push_back(ref);
  new_frames->frames_.push_back(cur);

  // ---- the reference's calling sequence ----
  SparseImgAlignOptions img_align_options;
  img_align_options.max_level = 4;
  img_align_options.min_level = 2;  // svo_factory.cpp:137-138
  SparseImgAlignHip::Ptr sparse_img_align(new SparseImgAlignHip(ctx, SparseImgAlignHip::getDefaultSolverOptions(), img_align_options));
  sparse_img_align->reset();
  Transformation T_prior{ { 1, 0, 0, 0 }, { 0, 0, 0 } };
  if (hdr[3]) sparse_img_align->setWeightedPrior(T_prior, 0.0, 0.0, 0.5, 0.0, 0.0, 0.0);
  sparse_img_align->setMaxNumFeaturesToAlign(-1);
  const Transformation T_iref_world = ref->T_imu_world();
  const Transformation T_icur_iref_init = svoh::mul(cur->T_imu_world(), svoh::inverse(T_iref_world));
  const size_t img_align_n_tracked = sparse_img_align->run(last_frames, new_frames);

  // ---- oracle on the same inputs ----
  orc_align_problem pb;
  memset(&pb, 0, sizeof pb);
  pb.n_cams = 1;
  orc_align_camera& oc = pb.cams[0];
  oc.ref_pyr.n_levels = oc.cur_pyr.n_levels = n_levels;
  std::vector<std::vector<uint8_t>> oref(n_levels), ocur(n_levels);
  uint8_t* oref_p[SVOH_MAX_LEVELS]; uint8_t* ocur_p[SVOH_MAX_LEVELS];
  for (int l = 0; l < n_levels; ++l) {
    oref[l].resize(ref_lv[l].size()); ocur[l].resize(cur_lv[l].size());
    oref_p[l] = oref[l].data(); ocur_p[l] = ocur[l].data();
  }
  orc_create_img_pyramid(img_ref.data(), w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, oref_p);
  orc_create_img_pyramid(img_cur.data(), w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, ocur_p);
  for (int l = 0; l < n_levels; ++l) {
    CHECK(oref[l] == ref_lv[l]);  // host img_pyr_ copy is bit-identical to the oracle's pyramid
    CHECK(ocur[l] == cur_lv[l]);
    oc.ref_pyr.level[l] = orc_image{ oref[l].data(), w >> l, h >> l, w >> l, 0 };
    oc.cur_pyr.level[l] = orc_image{ ocur[l].data(), w >> l, h >> l, w >> l, 0 };
  }
  oc.cam = cam;
  svoh::store_rigid(ref->T_imu_cam(), oc.ref_T_imu_cam);
  svoh::store_rigid(ref->T_cam_imu(), oc.ref_T_cam_imu);
  svoh::store_rigid(cur->T_cam_imu(), oc.cur_T_cam_imu);
  const svoh::Vec3 rp = ref->pos();
  oc.ref_pos[0] = rp.x; oc.ref_pos[1] = rp.y; oc.ref_pos[2] = rp.z;
  oc.n_features = n; oc.px = px.data(); oc.f = fv.data(); oc.pos_world = pw.data(); oc.flags = flags.data();
  svoh::store_rigid(T_icur_iref_init, pb.T_icur_iref);
  if (hdr[3]) { pb.prior.have_prior = 1; svoh::store_rigid(T_prior, pb.prior.T_prior); pb.prior.lambda_rot = 0.5; }
  svoh_align_options opt{};
  opt.max_level = 4; opt.min_level = 2; opt.patch_size = 4; opt.max_iter = 10; opt.eps = 0.0005; opt.weight_scale = 10;
  svoh_align_result ores;
  const int on = orc_sparse_align_run(&opt, &pb, &ores, nullptr);

  CHECK((int)img_align_n_tracked == on);
  const Transformation T_exp = svoh::mul(svoh::mul(cur->T_cam_imu(), svoh::load_rigid(ores.T_icur_iref)), T_iref_world);
  const double d[7] = { cur->T_f_w_.q.w - T_exp.q.w, cur->T_f_w_.q.x - T_exp.q.x, cur->T_f_w_.q.y - T_exp.q.y,
                        cur->T_f_w_.q.z - T_exp.q.z, cur->T_f_w_.t.x - T_exp.t.x, cur->T_f_w_.t.y - T_exp.t.y,
                        cur->T_f_w_.t.z - T_exp.t.z };
  double m = 0;
  for (double v : d) m = fmax(m, fabs(v));
  printf("tracked %zu, |T_f_w(gpu) - T_f_w(oracle)| = %.3e, iters gpu/oracle L4 %d/%d L3 %d/%d L2 %d/%d\n",
         img_align_n_tracked, m, sparse_img_align->lastResult().iters[4], ores.iters[4],
         sparse_img_align->lastResult().iters[3], ores.iters[3], sparse_img_align->lastResult().iters[2], ores.iters[2]);
  CHECK(m < 1e-8);  // fp64 both sides; only summation order differs
  for (int l = 0; l < SVOH_MAX_LEVELS; ++l) CHECK(sparse_img_align->lastResult().iters[l] == ores.iters[l]);

  // ---- the patch-split form: one participant, the sum over participants is the identity ----
  {
    const Transformation T_run = cur->T_f_w_;
    const svoh_align_result whole = sparse_img_align->lastResult();
    cur->T_f_w_ = to_T(T_cur_init_f_w.data());
    if (hdr[3]) sparse_img_align->setWeightedPrior(T_prior, 0.0, 0.0, 0.5, 0.0, 0.0, 0.0);
    sparse_img_align->setMaxNumFeaturesToAlign(5);   // has no effect, as in the reference
    int n_sums = 0;
    bool args_ok = true;
    const size_t n_split = sparse_img_align->runSplit(last_frames, new_frames, 0, 1, [&](double* d_sums, size_t n) {
      args_ok = args_ok && d_sums != nullptr && n == SVOH_ALIGN_SUMS_DOUBLES;
      ++n_sums;
    });
    CHECK(args_ok);
    const svoh_align_result& sp = sparse_img_align->lastResult();
    int total_iters = 0;
    for (int l = 0; l < SVOH_MAX_LEVELS; ++l) { CHECK(sp.iters[l] == whole.iters[l]); CHECK(sp.n_meas[l] == whole.n_meas[l]); total_iters += sp.iters[l]; }
    CHECK(n_sums == total_iters);
    CHECK(n_split > 0 && n_split <= img_align_n_tracked);
    const double ds[7] = { cur->T_f_w_.q.w - T_run.q.w, cur->T_f_w_.q.x - T_run.q.x, cur->T_f_w_.q.y - T_run.q.y,
                           cur->T_f_w_.q.z - T_run.q.z, cur->T_f_w_.t.x - T_run.t.x, cur->T_f_w_.t.y - T_run.t.y,
                           cur->T_f_w_.t.z - T_run.t.z };
    double ms = 0;
    for (double v : ds) ms = fmax(ms, fabs(v));
    printf("runSplit: %zu visible patches, %d evaluations, |T_f_w(split) - T_f_w(run)| = %.3e\n", n_split, n_sums, ms);
    CHECK(ms < 1e-9);
    bool threw = false;
    try { sparse_img_align->runSplit(last_frames, new_frames, 2, 2, nullptr); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw);
  }
  svoh_destroy(ctx);
  printf("PASS\n");
  return 0;
}
