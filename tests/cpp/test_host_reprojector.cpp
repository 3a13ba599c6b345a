// test_host_reprojector.cpp -- the C++ host mirror reprojector_utils::matchCandidates (two speculative GPU
// batches + ordered host replay) against the oracle's sequential restatement of the reference loop
// (src/svo/src/reprojector.cpp:342-486), in the call shape of Reprojector::reprojectFrames (:104-131).
// Input: a dump written by tests/test_host_cpp_gpu.py.
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

struct OrcPyr {
  std::vector<std::vector<uint8_t>> lv;
  orc_frame_view view;
  void build(const std::vector<uint8_t>& img, int w, int h, int n_levels, const svoh_camera& cam, const Transformation& T,
             double mu_range, int id)
  {
    lv.resize(n_levels);
    uint8_t* p[SVOH_MAX_LEVELS];
    for (int l = 0; l < n_levels; ++l) { lv[l].resize((size_t)(w >> l) * (h >> l)); p[l] = lv[l].data(); }
    orc_create_img_pyramid(img.data(), w, h, w, n_levels, SVOH_HALFSAMPLE_REFERENCE, p);
    memset(&view, 0, sizeof view);
    view.pyr.n_levels = n_levels;
    for (int l = 0; l < n_levels; ++l) view.pyr.level[l] = orc_image{ lv[l].data(), w >> l, h >> l, w >> l, 0 };
    view.cam = cam;
    svoh::store_rigid(T, view.T_f_w);
    view.seed_mu_range = mu_range;
    view.id = id;
  }
};

int main(int argc, char** argv)
{
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror("open"); return 2; }
  std::vector<int32_t> hdr = rd<int32_t>(f, 4);  // w, h, n_features, max_n_features_per_frame
  const int w = hdr[0], h = hdr[1], n = hdr[2], max_n = hdr[3];
  std::vector<double> camv = rd<double>(f, 9), T_kf = rd<double>(f, 7), T_cur = rd<double>(f, 7), T_far = rd<double>(f, 7),
                      mu_range = rd<double>(f, 1);
  std::vector<double> px = rd<double>(f, 2 * (size_t)n), fv = rd<double>(f, 3 * (size_t)n), grad = rd<double>(f, 2 * (size_t)n),
                      state = rd<double>(f, 4 * (size_t)n), lm_pos = rd<double>(f, 3 * (size_t)n);
  std::vector<int32_t> level = rd<int32_t>(f, n);
  std::vector<uint8_t> type = rd<uint8_t>(f, n), lm_kind = rd<uint8_t>(f, n);  // 0 none, 1 landmark, 2 landmark seen only from afar
  std::vector<int32_t> order = rd<int32_t>(f, n);
  std::vector<double> cur_px = rd<double>(f, 2 * (size_t)n), score = rd<double>(f, n);
  std::vector<uint8_t> img_kf = rd<uint8_t>(f, (size_t)w * h), img_cur = rd<uint8_t>(f, (size_t)w * h);
  fclose(f);

  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) { fprintf(stderr, "svoh_create: %s\n", svoh_last_error_string(nullptr)); return 3; }
  svoh_camera cam{};
  cam.fx = camv[0]; cam.fy = camv[1]; cam.cx = camv[2]; cam.cy = camv[3];
  for (int i = 0; i < 4; ++i) cam.d[i] = camv[4 + i];
  cam.distortion = camv[8] != 0.0 ? SVOH_DISTORTION_RADTAN : SVOH_DISTORTION_NONE;
  cam.width = w; cam.height = h;
  const int n_levels = 5;

  auto make_frame = [&](const std::vector<uint8_t>& img, const double* T, int id) {
    FramePtr fr(new Frame);
    if (svoh_build_pyramid(ctx, img.data(), w, h, w, SVOH_MEM_HOST, n_levels, SVOH_HALFSAMPLE_REFERENCE, nullptr, &fr->pyramid) != SVOH_OK)
      { fprintf(stderr, "build_pyramid: %s\n", svoh_last_error_string(ctx)); exit(3); }
    fr->cam = cam; fr->T_f_w_ = to_T(T); fr->id_ = id;
    return fr;
  };
  FramePtr kf = make_frame(img_kf, T_kf.data(), 7), cur = make_frame(img_cur, T_cur.data(), 8), far = make_frame(img_kf, T_far.data(), 9);
  kf->num_features_ = (size_t)n;
  kf->px_vec_ = px; kf->f_vec_ = fv; kf->grad_vec_ = grad; kf->level_vec_ = level; kf->type_vec_ = type;
  kf->invmu_sigma2_a_b_vec_ = state; kf->seed_mu_range_ = mu_range[0];
  kf->landmark_vec_.resize(n);
  far->num_features_ = 1; far->px_vec_ = { 100, 100 }; far->f_vec_ = { 0, 0, 1 }; far->grad_vec_ = { 1, 0 };
  far->level_vec_ = { 0 }; far->type_vec_ = { SVOH_FT_CORNER }; far->invmu_sigma2_a_b_vec_ = { 1, 1, 10, 10 };
  std::vector<PointPtr> points(n);
  for (int i = 0; i < n; ++i) {
    if (!lm_kind[i]) continue;
    PointPtr p(new Point);
    p->pos_ = { lm_pos[3 * i], lm_pos[3 * i + 1], lm_pos[3 * i + 2] };
    p->id_ = 1000 + i;
    p->obs_.push_back(Point::Obs{ far, 0 });
    if (lm_kind[i] == 1) p->obs_.push_back(Point::Obs{ kf, (size_t)i });
    kf->landmark_vec_[i] = p;
    points[i] = p;
  }
  std::vector<reprojector::Candidate> candidates;
  for (int k = 0; k < n; ++k) {
    const int i = order[k];
    reprojector::Candidate c;
    c.ref_frame = kf; c.ref_index = (size_t)i; c.cur_px[0] = cur_px[2 * i]; c.cur_px[1] = cur_px[2 * i + 1];
    c.type = type[i]; c.score = score[i];
    candidates.push_back(c);
  }
  const int cell = 30;
  OccupandyGrid2D grid(cell, OccupandyGrid2D::getNCell(w, cell), OccupandyGrid2D::getNCell(h, cell));
  // a few cells are already taken (features matched from a previous keyframe)
  for (size_t k = 0; k < grid.size(); k += 17) grid.setOccupied(k);
  std::vector<uint8_t> occ0(grid.size());
  for (size_t k = 0; k < grid.size(); ++k) occ0[k] = grid.isOccupied(k);
  cur->num_features_ = 0;
  reprojector::Statistics stats;
  const double seed_sigma2_thresh = 200.0;

  // ---- oracle first (the mirror mutates kf's seeds) ----
  OrcPyr o_kf, o_cur;
  o_kf.build(img_kf, w, h, n_levels, cam, kf->T_f_w_, mu_range[0], 7);
  o_cur.build(img_cur, w, h, n_levels, cam, cur->T_f_w_, 0.0, 8);
  std::vector<orc_candidate> oc(n);
  for (int k = 0; k < n; ++k) {
    const int i = order[k];
    orc_candidate& c = oc[k];
    memset(&c, 0, sizeof c);
    c.ref_frame_idx = 0;
    const uint8_t t = type[i];
    if (lm_kind[i] == 2) c.kind = 3;
    else if (lm_kind[i] == 1) c.kind = 2;
    else c.kind = (t == SVOH_FT_CORNER_SEED_CONVERGED || t == SVOH_FT_EDGELET_SEED_CONVERGED || t == SVOH_FT_MAPPOINT_SEED_CONVERGED) ? 0 : 1;
    c.cur_px[0] = cur_px[2 * i]; c.cur_px[1] = cur_px[2 * i + 1];
    for (int j = 0; j < 2; ++j) { c.px[j] = px[2 * i + j]; c.grad[j] = grad[2 * i + j]; }
    for (int j = 0; j < 3; ++j) c.f[j] = fv[3 * i + j];
    for (int j = 0; j < 4; ++j) c.state[j] = state[4 * i + j];
    c.level = level[i]; c.type = t; c.ref_type = t; c.score = score[i];
    if (c.kind == 0) c.depth = 1.0 / state[4 * i];
    if (c.kind == 2) {
      const svoh::Vec3 p = kf->pos();
      c.depth = sqrt((p.x - lm_pos[3 * i]) * (p.x - lm_pos[3 * i]) + (p.y - lm_pos[3 * i + 1]) * (p.y - lm_pos[3 * i + 1]) +
                     (p.z - lm_pos[3 * i + 2]) * (p.z - lm_pos[3 * i + 2]));
    }
  }
  svoh_matcher_options mo{};
  mo.align_max_iter = 10; mo.max_epi_search_steps = 100; mo.subpix_refinement = 1; mo.epi_search_edgelet_filtering = 1;
  mo.scan_on_unit_sphere = 1; mo.affine_est_offset = 1; mo.affine_est_gain = 0;
  mo.epi_search_edgelet_max_angle = 0.7; mo.max_patch_diff_ratio = 2.0;
  svoh_depth_filter_options dopt{};
  dopt.seed_convergence_sigma2_thresh = seed_sigma2_thresh; dopt.mappoint_convergence_sigma2_thresh = seed_sigma2_thresh;
  dopt.px_error_angle = atan(1.0 / (2.0 * cam.fx)) + atan(1.0 / (2.0 * cam.fy));
  dopt.check_visibility = 0; dopt.check_convergence = 0; dopt.use_vogiatzis_update = 1;
  std::vector<uint8_t> oocc = occ0, ovis(n);
  std::vector<int32_t> ores(n);
  std::vector<orc_new_feature> onew(n);
  int o_nout = 0, o_trials = 0, o_matches = 0, o_failed = 0, o_succ = 0, o_numf = 0;
  const int o_consumed = orc_match_candidates(&mo, &dopt, 1, &o_kf.view, &o_cur.view, n, oc.data(), max_n, &o_numf, cell,
                                              grid.n_cols, grid.n_rows, oocc.data(), ovis.data(), ores.data(), onew.data(),
                                              &o_nout, &o_trials, &o_matches, &o_failed, &o_succ);

  // ---- the reference's call: reprojector_utils::matchCandidates(frame, max_n, aff_off, aff_gain, candidates, grid, stats, thresh) ----
  reprojector_utils::matchCandidates(ctx, cur, (size_t)max_n, true, false, candidates, grid, stats, seed_sigma2_thresh);
  const std::vector<int32_t>& res = reprojector_utils::lastMatchResults();

  CHECK((int)candidates.size() == n - o_consumed);
  CHECK((int)stats.n_trials == o_trials && (int)stats.n_matches == o_matches);
  CHECK((int)cur->num_features_ == o_numf && o_numf == o_nout);
  for (size_t k = 0; k < grid.size(); ++k) CHECK(grid.isOccupied(k) == (oocc[k] != 0));
  int n_kinds[4] = { 0, 0, 0, 0 }, n_visited = 0;
  for (int k = 0; k < n; ++k) {
    if (res[k] != ores[k])
      fprintf(stderr, "candidate %d (feature %d, kind %d, type %d, level %d): mirror result %d, oracle %d\n", k, order[k], oc[k].kind,
              (int)type[order[k]], level[order[k]], res[k], ores[k]);
    CHECK(res[k] == ores[k]);
    if (ovis[k]) { ++n_visited; ++n_kinds[oc[k].kind]; }
  }
  // side effects on the keyframe: seeds of visited unconverged candidates changed, all others untouched
  double worst_state = 0;
  for (int k = 0; k < n; ++k) {
    const int i = order[k];
    CHECK(kf->type_vec_[i] == oc[k].ref_type);
    for (int j = 0; j < 4; ++j) {
      const double a = kf->invmu_sigma2_a_b_vec_[4 * i + j], b = oc[k].state[j];
      const double d = fabs(a - b) / fmax(1e-300, fabs(b));
      if (d > worst_state) worst_state = d;
      if (!(ovis[k] && oc[k].kind == 1)) CHECK(a == state[4 * i + j]);
    }
  }
  CHECK(worst_state < 1e-9);
  int failed = 0, succ = 0;
  for (int i = 0; i < n; ++i) if (points[i]) { failed += points[i]->n_failed_reproj_; succ += points[i]->n_succeeded_reproj_; }
  CHECK(failed == o_failed && succ == o_succ);
  // the new features of the current frame, slot by slot
  double worst_px = 0, worst_f = 0, worst_g = 0;
  for (int s = 0; s < o_nout; ++s) {
    const orc_new_feature& o = onew[s];
    const int i = order[o.candidate];
    CHECK(cur->type_vec_[s] == o.type && cur->level_vec_[s] == o.level && cur->score_vec_[s] == o.score);
    for (int j = 0; j < 2; ++j) {
      worst_px = fmax(worst_px, fabs(cur->px_vec_[2 * s + j] - o.px[j]));
      worst_g = fmax(worst_g, fabs(cur->grad_vec_[2 * s + j] - o.grad[j]));
    }
    for (int j = 0; j < 3; ++j) worst_f = fmax(worst_f, fabs(cur->f_vec_[3 * s + j] - o.f[j]));
    for (int j = 0; j < 4; ++j)
      CHECK(fabs(cur->invmu_sigma2_a_b_vec_[4 * s + j] - o.state[j]) <= 1e-9 * fabs(o.state[j]));
    if (lm_kind[i] == 1) CHECK(cur->landmark_vec_[s] == points[i]);
    else CHECK(cur->seed_ref_vec_[s].keyframe == kf && cur->seed_ref_vec_[s].seed_id == i);
  }
  printf("reprojector: %d candidates, %d visited (converged seeds %d, seed updates %d, landmarks %d, no close view %d), "
         "%d matched, %d consumed; worst |dpx| %.2e |df| %.2e |dgrad| %.2e, seed state rel %.2e\n",
         n, n_visited, n_kinds[0], n_kinds[1], n_kinds[2], n_kinds[3], o_nout, o_consumed, worst_px, worst_f, worst_g, worst_state);
  CHECK(worst_px <= 1e-4 && worst_f <= 1e-6 && worst_g <= 1e-9);
  CHECK(n_kinds[0] > 0 && n_kinds[1] > 0 && n_kinds[2] > 0 && n_kinds[3] > 0 && o_nout > 10);
  if (max_n > 0) CHECK(o_consumed < n);   // the early break was exercised
  // ================= Reprojector::reprojectFrames (reprojector.cpp:27-306), fresh frames =================
  {
    FramePtr kf2 = make_frame(img_kf, T_kf.data(), 17), cur2 = make_frame(img_cur, T_cur.data(), 18), far2 = make_frame(img_kf, T_far.data(), 19);
    kf2->num_features_ = (size_t)n;
    kf2->px_vec_ = px; kf2->f_vec_ = fv; kf2->grad_vec_ = grad; kf2->level_vec_ = level; kf2->type_vec_ = type;
    kf2->invmu_sigma2_a_b_vec_ = state; kf2->seed_mu_range_ = mu_range[0]; kf2->score_vec_ = score;
    kf2->landmark_vec_.assign(n, nullptr);
    far2->num_features_ = 1; far2->px_vec_ = { 100, 100 }; far2->f_vec_ = { 0, 0, 1 }; far2->grad_vec_ = { 1, 0 };
    far2->level_vec_ = { 0 }; far2->type_vec_ = { SVOH_FT_CORNER }; far2->invmu_sigma2_a_b_vec_ = { 1, 1, 10, 10 };
    std::vector<PointPtr> pts(n);
    for (int i = 0; i < n; ++i) {
      if (!lm_kind[i]) continue;
      PointPtr p(new Point);
      p->pos_ = { lm_pos[3 * i], lm_pos[3 * i + 1], lm_pos[3 * i + 2] };
      p->obs_.push_back(Point::Obs{ far2, 0 });
      if (lm_kind[i] == 1) p->obs_.push_back(Point::Obs{ kf2, (size_t)i });
      p->n_succeeded_reproj_ = i % 5; p->n_failed_reproj_ = i % 3;
      kf2->landmark_vec_[i] = p;
      pts[i] = p;
    }
    ReprojectorOptions ro;
    ro.max_n_features_per_frame = (size_t)(max_n > 0 ? max_n : 220);
    ro.max_unconverged_seeds_ratio = 0.6;
    ReprojectorHip reprojector(ctx, ro, 0);
    std::vector<PointPtr> trash;

    // ---- expected: the same three passes, sequential, with the oracle's matcher ----
    const int n_cols = OccupandyGrid2D::getNCell(w, 30), n_rows = OccupandyGrid2D::getNCell(h, 30);
    std::vector<uint8_t> eocc((size_t)n_cols * n_rows, 0);
    struct Cand { int i; int n_reproj; double score; uint8_t type; double cur_px[2]; };
    auto project = [&](const double* xyz_w, double* pxo) {
      double xf[3], tl[3], z0[2] = { 0, 0 };
      orc_se3_transform(&o_cur.view.T_f_w, xyz_w, xf);
      orc_back_project3(&cam, z0, tl);
      const double min_cos = tl[2] / sqrt(tl[0] * tl[0] + tl[1] * tl[1] + tl[2] * tl[2]);
      if (xf[2] / sqrt(xf[0] * xf[0] + xf[1] * xf[1] + xf[2] * xf[2]) < min_cos) return false;
      orc_project3(&cam, xf, pxo, nullptr);
      if (!(pxo[0] >= 0 && pxo[1] >= 0 && pxo[0] < w && pxo[1] < h)) return false;
      const int a = (int)pxo[0], b = (int)pxo[1];
      return a >= 8 && b >= 8 && a < w - 8 && b < h - 8;
    };
    auto sort_c = [](std::vector<Cand>& v) {
      std::sort(v.begin(), v.end(), [](const Cand& l, const Cand& r) {
        return l.type > r.type || (l.type == r.type && l.n_reproj > r.n_reproj) || (l.type == r.type && l.n_reproj == r.n_reproj && l.score > r.score);
      });
    };
    std::vector<double> estate = state;
    std::vector<uint8_t> etype = type;
    std::vector<int> e_failed(n, 0), e_succ(n, 0);
    for (int i = 0; i < n; ++i) if (pts[i]) { e_failed[i] = pts[i]->n_failed_reproj_; e_succ[i] = pts[i]->n_succeeded_reproj_; }
    int e_numf = 0, e_trials = 0, e_matches = 0, e_trash = 0;
    std::vector<orc_new_feature> e_new;
    auto run_pass = [&](std::vector<Cand>& cs, int kind_of, int max_allowed) {
      sort_c(cs);
      std::vector<orc_candidate> ocs(cs.size());
      for (size_t k = 0; k < cs.size(); ++k) {
        orc_candidate& c = ocs[k];
        const int i = cs[k].i;
        memset(&c, 0, sizeof c);
        c.kind = kind_of; c.cur_px[0] = cs[k].cur_px[0]; c.cur_px[1] = cs[k].cur_px[1];
        for (int j = 0; j < 2; ++j) { c.px[j] = px[2 * i + j]; c.grad[j] = grad[2 * i + j]; }
        for (int j = 0; j < 3; ++j) c.f[j] = fv[3 * i + j];
        for (int j = 0; j < 4; ++j) c.state[j] = estate[4 * i + j];
        c.level = level[i]; c.type = cs[k].type; c.ref_type = etype[i]; c.score = cs[k].score;
        if (kind_of == 0) c.depth = 1.0 / estate[4 * i];
        if (kind_of == 2) {
          const svoh::Vec3 p = kf2->pos();
          c.depth = sqrt((p.x - lm_pos[3 * i]) * (p.x - lm_pos[3 * i]) + (p.y - lm_pos[3 * i + 1]) * (p.y - lm_pos[3 * i + 1]) +
                         (p.z - lm_pos[3 * i + 2]) * (p.z - lm_pos[3 * i + 2]));
        }
      }
      std::vector<uint8_t> vis(cs.size() + 1);
      std::vector<int32_t> res(cs.size() + 1);
      std::vector<orc_new_feature> nf(cs.size() + 1);
      int nout = 0, trials = 0, matches = 0, failed = 0, succ = 0;
      const int consumed = orc_match_candidates(&mo, &dopt, 1, &o_kf.view, &o_cur.view, (int)cs.size(), ocs.data(), max_allowed, &e_numf, 30, n_cols,
                                                n_rows, eocc.data(), vis.data(), res.data(), nf.data(), &nout, &trials, &matches, &failed, &succ);
      e_trials += trials; e_matches += matches;
      for (size_t k = 0; k < cs.size(); ++k) {
        const int i = cs[k].i;
        if (vis[k] && kind_of == 1) { for (int j = 0; j < 4; ++j) estate[4 * i + j] = ocs[k].state[j]; etype[i] = ocs[k].ref_type; }
        if (vis[k] && kind_of == 2) { if (res[k] == 0) ++e_succ[i]; else ++e_failed[i]; }
      }
      for (int k = 0; k < nout; ++k) { nf[k].candidate = cs[nf[k].candidate].i; e_new.push_back(nf[k]); }
      cs.erase(cs.begin(), cs.begin() + consumed);
    };
    auto enough = [&]() { return e_numf >= (int)ro.max_n_features_per_frame; };
    auto occupy = [&](const std::vector<Cand>& cs) {
      for (const Cand& c : cs) eocc[(size_t)(floor((double)(int)c.cur_px[1] / 30) * n_cols + floor((double)(int)c.cur_px[0] / 30))] = 1;
    };
    bool e_done = false;
    {
      std::vector<Cand> cs;
      for (int i = 0; i < n; ++i) {
        if (!lm_kind[i]) continue;
        if (lm_kind[i] == 2) { ++e_trash; continue; }   // one observation only: unconstrained
        Cand c{ i, e_succ[i] - e_failed[i], score[i], type[i], { 0, 0 } };
        if (project(&lm_pos[3 * i], c.cur_px)) cs.push_back(c);
      }
      run_pass(cs, 2, (int)ro.max_n_features_per_frame);
      if (enough()) occupy(cs);
    }
    auto seed_pass = [&](bool converged) {
      std::vector<Cand> cs;
      for (int i = 0; i < n; ++i) {
        if (lm_kind[i]) continue;
        const uint8_t t = etype[i];
        const bool conv = t == SVOH_FT_CORNER_SEED_CONVERGED || t == SVOH_FT_EDGELET_SEED_CONVERGED;
        const bool unconv = t == SVOH_FT_CORNER_SEED || t == SVOH_FT_EDGELET_SEED;
        if (!(converged ? conv : unconv)) continue;
        const double depth = 1.0 / estate[4 * i];
        const double in_f[3] = { fv[3 * i] * depth, fv[3 * i + 1] * depth, fv[3 * i + 2] * depth };
        svoh_se3 T_w_f;
        double xw[3];
        orc_se3_inverse(&o_kf.view.T_f_w, &T_w_f);
        orc_se3_transform(&T_w_f, in_f, xw);
        Cand c{ i, 0, score[i], t, { 0, 0 } };
        if (project(xw, c.cur_px)) cs.push_back(c);
      }
      return cs;
    };
    {
      std::vector<Cand> cs = seed_pass(true);
      if (enough()) { occupy(cs); e_done = true; }
      else { run_pass(cs, 0, (int)ro.max_n_features_per_frame); if (enough()) { occupy(cs); e_done = true; } }
    }
    if (!e_done) {
      std::vector<Cand> cs = seed_pass(false);
      size_t max_allowed_total = ro.max_n_features_per_frame;
      const size_t alt = (size_t)(e_numf / (1 - ro.max_unconverged_seeds_ratio));
      if (max_allowed_total > alt) max_allowed_total = alt;
      run_pass(cs, 1, (int)max_allowed_total);
      if (enough()) occupy(cs);
    }

    // ---- the mirror ----
    reprojector.reprojectFrames(cur2, { kf2 }, trash);
    CHECK((int)trash.size() == e_trash && e_trash > 0);
    CHECK((int)reprojector.stats_.n_trials == e_trials && (int)reprojector.stats_.n_matches == e_matches);
    CHECK((int)cur2->num_features_ == e_numf && e_numf == (int)e_new.size() && e_numf > 20);
    for (size_t k = 0; k < eocc.size(); ++k) CHECK(reprojector.grid_->isOccupied(k) == (eocc[k] != 0));
    int kinds[3] = { 0, 0, 0 };
    for (int s2 = 0; s2 < e_numf; ++s2) {
      const orc_new_feature& o = e_new[s2];
      const int i = o.candidate;
      CHECK(cur2->type_vec_[s2] == o.type && cur2->level_vec_[s2] == o.level);
      CHECK(fabs(cur2->px_vec_[2 * s2] - o.px[0]) <= 1e-4 && fabs(cur2->px_vec_[2 * s2 + 1] - o.px[1]) <= 1e-4);
      if (lm_kind[i]) { CHECK(cur2->landmark_vec_[s2] == pts[i]); ++kinds[0]; }
      else { CHECK(cur2->seed_ref_vec_[s2].keyframe == kf2 && cur2->seed_ref_vec_[s2].seed_id == i); ++kinds[(type[i] == 3 || type[i] == 4) ? 1 : 2]; }
    }
    for (int i = 0; i < n; ++i) {
      CHECK(kf2->type_vec_[i] == etype[i]);
      for (int j = 0; j < 4; ++j) CHECK(fabs(kf2->invmu_sigma2_a_b_vec_[4 * i + j] - estate[4 * i + j]) <= 1e-9 * fabs(estate[4 * i + j]));
      if (pts[i]) CHECK(pts[i]->n_failed_reproj_ == e_failed[i] && pts[i]->n_succeeded_reproj_ == e_succ[i]);
    }
    printf("reprojectFrames: %d features (%d landmarks, %d converged seeds, %d seed updates), %d trials, %d unconstrained points trashed\n",
           e_numf, kinds[0], kinds[1], kinds[2], e_trials, e_trash);
    CHECK(kinds[0] > 0 && (max_n > 0 || (kinds[1] > 0 && kinds[2] > 0)));   // the 220-feature run reaches all three passes
  }
  // ================= f-4 on the device: candidate projection queued on the context =================
  // The same reprojection twice on fresh frames: once with the candidates' projection computed by the host mirror
  // (reprojector_utils::getCandidate), once with the projection of every feature of the keyframe queued on the device
  // first (ReprojectorHip::enqueueCandidateProjection -> svoh_project_candidates_enqueue).  The device's pixel and
  // verdict of every feature must be the host's bit for bit, and so must everything reprojectFrames leaves behind.
  {
    struct Set { FramePtr kf, cur, far; std::vector<PointPtr> pts; };
    auto build = [&](int id0) {
      Set s;
      s.kf = make_frame(img_kf, T_kf.data(), id0); s.cur = make_frame(img_cur, T_cur.data(), id0 + 1); s.far = make_frame(img_kf, T_far.data(), id0 + 2);
      s.kf->num_features_ = (size_t)n;
      s.kf->px_vec_ = px; s.kf->f_vec_ = fv; s.kf->grad_vec_ = grad; s.kf->level_vec_ = level; s.kf->type_vec_ = type;
      s.kf->invmu_sigma2_a_b_vec_ = state; s.kf->seed_mu_range_ = mu_range[0]; s.kf->score_vec_ = score;
      s.kf->landmark_vec_.assign(n, nullptr);
      s.far->num_features_ = 1; s.far->px_vec_ = { 100, 100 }; s.far->f_vec_ = { 0, 0, 1 }; s.far->grad_vec_ = { 1, 0 };
      s.far->level_vec_ = { 0 }; s.far->type_vec_ = { SVOH_FT_CORNER }; s.far->invmu_sigma2_a_b_vec_ = { 1, 1, 10, 10 };
      s.pts.resize(n);
      for (int i = 0; i < n; ++i) {
        if (!lm_kind[i]) continue;
        PointPtr p(new Point);
        p->pos_ = { lm_pos[3 * i], lm_pos[3 * i + 1], lm_pos[3 * i + 2] };
        p->obs_.push_back(Point::Obs{ s.far, 0 });
        if (lm_kind[i] == 1) p->obs_.push_back(Point::Obs{ s.kf, (size_t)i });
        p->n_succeeded_reproj_ = i % 5; p->n_failed_reproj_ = i % 3;
        s.kf->landmark_vec_[i] = p;
        s.pts[i] = p;
      }
      return s;
    };
    ReprojectorOptions ro;
    ro.max_n_features_per_frame = (size_t)(max_n > 0 ? max_n : 220);
    ro.max_unconverged_seeds_ratio = 0.6;
    Set host = build(31), dev = build(41);
    // (a) the device's projection of every feature against the host's getCandidate
    {
      std::vector<uint8_t> kind(n), vis(n);
      std::vector<int32_t> kfi(n, 0);
      std::vector<double> v(3 * (size_t)n), mu(n), dpx(2 * (size_t)n);
      for (int i = 0; i < n; ++i) {
        kind[i] = lm_kind[i] ? 0 : 1;
        for (int j = 0; j < 3; ++j) v[3 * i + j] = lm_kind[i] ? lm_pos[3 * i + j] : fv[3 * i + j];
        mu[i] = state[4 * i];
      }
      svoh_se3 T_f_w, T_w_kf;
      svoh::store_rigid(dev.cur->T_f_w_, T_f_w);
      svoh::store_rigid(svoh::inverse(dev.kf->T_f_w_), T_w_kf);
      CHECK(svoh_project_candidates(ctx, &dev.cur->cam, &T_f_w, 1, &T_w_kf, n, kind.data(), kfi.data(), v.data(), mu.data(), dpx.data(), vis.data()) == SVOH_OK);
      int n_vis = 0;
      for (int i = 0; i < n; ++i) {
        reprojector::Candidate c;
        const bool ok = reprojector_utils::getCandidate(host.cur, host.kf, (size_t)i, c);
        CHECK(ok == (vis[i] != 0));
        if (ok) { CHECK(c.cur_px[0] == dpx[2 * i] && c.cur_px[1] == dpx[2 * i + 1]); ++n_vis; }
      }
      CHECK(n_vis > 20);
      // misuse: a second queued call before the first is collected, a collect with the wrong count
      CHECK(svoh_project_candidates_enqueue(ctx, &dev.cur->cam, &T_f_w, nullptr, -1, 1, &T_w_kf, n, kind.data(), kfi.data(), v.data(), mu.data()) == SVOH_OK);
      CHECK(svoh_project_candidates_enqueue(ctx, &dev.cur->cam, &T_f_w, nullptr, -1, 1, &T_w_kf, n, kind.data(), kfi.data(), v.data(), mu.data()) != SVOH_OK);
      CHECK(svoh_project_candidates_collect(ctx, n - 1, dpx.data(), vis.data()) != SVOH_OK);
      CHECK(svoh_project_candidates_collect(ctx, n, dpx.data(), vis.data()) == SVOH_OK);
      CHECK(svoh_project_candidates_enqueue(ctx, &dev.cur->cam, &T_f_w, nullptr, 0, 1, &T_w_kf, n, kind.data(), kfi.data(), v.data(), mu.data()) != SVOH_OK);   // pose from an alignment result without T_post
    }
    // (b) reprojectFrames with and without the queued projection
    ReprojectorHip r_host(ctx, ro, 0), r_dev(ctx, ro, 0);
    std::vector<PointPtr> trash_h, trash_d;
    r_host.reprojectFrames(host.cur, { host.kf }, trash_h);
    r_dev.enqueueCandidateProjection(dev.cur, { dev.kf }, nullptr, -1);
    r_dev.reprojectFrames(dev.cur, { dev.kf }, trash_d);
    CHECK(trash_h.size() == trash_d.size());
    CHECK(host.cur->num_features_ == dev.cur->num_features_ && host.cur->num_features_ > 20);
    CHECK(r_host.stats_.n_trials == r_dev.stats_.n_trials && r_host.stats_.n_matches == r_dev.stats_.n_matches);
    for (size_t k = 0; k < (size_t)r_host.grid_->occupancy_.size(); ++k) CHECK(r_host.grid_->isOccupied(k) == r_dev.grid_->isOccupied(k));
    for (size_t s2 = 0; s2 < host.cur->num_features_; ++s2) {
      CHECK(host.cur->px_vec_[2 * s2] == dev.cur->px_vec_[2 * s2] && host.cur->px_vec_[2 * s2 + 1] == dev.cur->px_vec_[2 * s2 + 1]);
      CHECK(host.cur->type_vec_[s2] == dev.cur->type_vec_[s2] && host.cur->level_vec_[s2] == dev.cur->level_vec_[s2]);
      for (int j = 0; j < 3; ++j) CHECK(host.cur->f_vec_[3 * s2 + j] == dev.cur->f_vec_[3 * s2 + j]);
    }
    for (int i = 0; i < n; ++i) {
      CHECK(host.kf->type_vec_[i] == dev.kf->type_vec_[i]);
      for (int j = 0; j < 4; ++j) CHECK(host.kf->invmu_sigma2_a_b_vec_[4 * i + j] == dev.kf->invmu_sigma2_a_b_vec_[4 * i + j]);
      if (host.pts[i]) CHECK(host.pts[i]->n_failed_reproj_ == dev.pts[i]->n_failed_reproj_ && host.pts[i]->n_succeeded_reproj_ == dev.pts[i]->n_succeeded_reproj_);
    }
    // a projection queued for another frame is not used (and does not block the context)
    Set other = build(51);
    r_dev.enqueueCandidateProjection(other.cur, { other.kf }, nullptr, -1);
    Set again = build(61);
    std::vector<PointPtr> trash_a;
    r_dev.reprojectFrames(again.cur, { again.kf }, trash_a);
    CHECK(again.cur->num_features_ == host.cur->num_features_);
    printf("device candidate projection: %zu features, identical to the host mirror\n", dev.cur->num_features_);

    // (c) a projection goes stale: a seed changes, and the keyframe's pose moves, between the queueing and the use.  The
    // reprojection must then be the one of a reprojector that never had a projection queued (the stale entries are
    // computed on the host instead of being taken from the snapshot).
    {
      Set a = build(71), b = build(71);
      ReprojectorHip ra(ctx, ro, 0), rb(ctx, ro, 0);
      ra.enqueueCandidateProjection(a.cur, { a.kf }, nullptr, -1);
      for (Set* s3 : { &a, &b }) {
        for (size_t i = 0; i < s3->kf->num_features_; i += 7)
          if (!s3->pts[i]) s3->kf->invmu_sigma2_a_b_vec_[4 * i] *= 1.02;   // every seventh seed has moved
      }
      std::vector<PointPtr> ta, tb;
      ra.reprojectFrames(a.cur, { a.kf }, ta);
      rb.reprojectFrames(b.cur, { b.kf }, tb);
      CHECK(a.cur->num_features_ == b.cur->num_features_ && ra.stats_.n_trials == rb.stats_.n_trials && ra.stats_.n_matches == rb.stats_.n_matches);
      for (size_t s2 = 0; s2 < a.cur->num_features_; ++s2)
        CHECK(a.cur->px_vec_[2 * s2] == b.cur->px_vec_[2 * s2] && a.cur->px_vec_[2 * s2 + 1] == b.cur->px_vec_[2 * s2 + 1]);
      Set c = build(81), d = build(81);
      ReprojectorHip rc(ctx, ro, 0), rd(ctx, ro, 0);
      rc.enqueueCandidateProjection(c.cur, { c.kf }, nullptr, -1);
      c.kf->T_f_w_.t.x += 0.01; d.kf->T_f_w_.t.x += 0.01;                      // the keyframe's pose has moved
      std::vector<PointPtr> tc, td;
      rc.reprojectFrames(c.cur, { c.kf }, tc);
      rd.reprojectFrames(d.cur, { d.kf }, td);
      CHECK(c.cur->num_features_ == d.cur->num_features_ && rc.stats_.n_trials == rd.stats_.n_trials && rc.stats_.n_matches == rd.stats_.n_matches);
      for (size_t s2 = 0; s2 < c.cur->num_features_; ++s2)
        CHECK(c.cur->px_vec_[2 * s2] == d.cur->px_vec_[2 * s2] && c.cur->px_vec_[2 * s2 + 1] == d.cur->px_vec_[2 * s2 + 1]);
      printf("stale projections (moved seeds, moved keyframe): recomputed on the host, same result\n");
    }

    // (d) a seed update sent off with updateSeedsAsync holds the context's one deferred section; a reprojection on the
    // same context in between finishes it instead of failing, and the update's results are those of the blocking call
    {
      Set e = build(91), f = build(91), g = build(92);
      DepthFilterOptions dfo;
      DepthFilterHip df_async(ctx, dfo), df_block(ctx, dfo);
      const size_t n_block = df_block.updateSeeds({ f.kf }, f.cur);
      df_async.updateSeedsAsync({ e.kf }, e.cur);
      CHECK(df_async.updatePending());
      ReprojectorHip rg(ctx, ro, 0);
      std::vector<PointPtr> tg;
      rg.reprojectFrames(g.cur, { g.kf }, tg);                                  // needs the deferred section
      CHECK(!df_async.updatePending());
      const size_t n_async = df_async.finishUpdateSeeds();
      CHECK(n_async == n_block && n_block > 0);
      for (size_t i = 0; i < e.kf->num_features_; ++i) {
        CHECK(e.kf->type_vec_[i] == f.kf->type_vec_[i]);
        for (int j = 0; j < 4; ++j) CHECK(e.kf->invmu_sigma2_a_b_vec_[4 * i + j] == f.kf->invmu_sigma2_a_b_vec_[4 * i + j]);
      }
      CHECK(g.cur->num_features_ > 20);
      printf("seed update in flight + reprojection on one context: finished early, %zu seeds updated as in the blocking call\n", n_async);
    }
  }
  svoh_destroy(ctx);
  printf("PASS\n");
  return 0;
}
