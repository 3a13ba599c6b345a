// reprojector_utils::getCandidate of the host mirror (svo_hip_host.cpp; reprojector.cpp:489-543) on inputs from a text file,
// for the CPU comparison with the NumPy restatement (tests/test_np_second_opinion_cpu.py).  No GPU call: the function is
// host arithmetic on Frame members only.
//   host_candidates_cpu <in.txt> <out.txt>
//   in : w h fx fy cx cy distortion k1 k2 p1 p2 / T_f_w_cur (qw qx qy qz tx ty tz) / T_f_w_ref / n / n lines: kind v0 v1 v2 mu
//        (kind 0: landmark at world position v; 1: seed with bearing vector v and inverse depth mu)
//   out: n lines: visible px0 px1
#include <cstdio>
#include <fstream>
#include <memory>
#include "../../svo_pro_universal_amd/host/svo_hip_host.h"
using namespace svo_hip;
int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  std::ifstream in(argv[1]);
  auto cur = std::make_shared<Frame>(), ref = std::make_shared<Frame>();
  svoh_camera cam{};
  in >> cam.width >> cam.height >> cam.fx >> cam.fy >> cam.cx >> cam.cy >> cam.distortion >> cam.d[0] >> cam.d[1] >> cam.d[2] >> cam.d[3];
  cur->cam = cam; ref->cam = cam;
  auto readT = [&](Transformation& T) { in >> T.q.w >> T.q.x >> T.q.y >> T.q.z >> T.t.x >> T.t.y >> T.t.z; };
  readT(cur->T_f_w_); readT(ref->T_f_w_);
  int n = 0;
  in >> n;
  ref->num_features_ = (size_t)n;
  ref->f_vec_.assign(3 * (size_t)n, 0.0); ref->invmu_sigma2_a_b_vec_.assign(4 * (size_t)n, 1.0);
  ref->type_vec_.assign((size_t)n, SVOH_FT_CORNER_SEED); ref->score_vec_.assign((size_t)n, 0.0);
  ref->landmark_vec_.assign((size_t)n, nullptr);
  for (int i = 0; i < n; ++i) {
    int kind; double v[3], mu;
    in >> kind >> v[0] >> v[1] >> v[2] >> mu;
    if (kind == 0) { PointPtr p(new Point); p->pos_ = { v[0], v[1], v[2] }; ref->landmark_vec_[i] = p; }
    else { for (int k = 0; k < 3; ++k) ref->f_vec_[3 * i + k] = v[k]; ref->invmu_sigma2_a_b_vec_[4 * i] = mu; }
  }
  if (!in) return 3;
  FILE* out = fopen(argv[2], "w");
  if (!out) return 4;
  for (int i = 0; i < n; ++i) {
    reprojector::Candidate c;
    const bool ok = reprojector_utils::getCandidate(cur, ref, (size_t)i, c);
    fprintf(out, "%d %.17g %.17g\n", ok ? 1 : 0, ok ? c.cur_px[0] : 0.0, ok ? c.cur_px[1] : 0.0);
  }
  fclose(out);
  return 0;
}
