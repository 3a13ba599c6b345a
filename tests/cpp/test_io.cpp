// test_io.cpp -- dumps what svo_hip::io parses (YAML subset, camera rig, front-end parameters, PNG, EuRoC
// folder) as "key value" lines; tests/test_io_cpu.py writes the inputs and checks the dump.  No GPU call.
#include <cstdio>
#include <cstring>
#include <string>

#include "../../svo_pro_universal_amd/host/svo_hip_io.h"

using namespace svo_hip;

int main(int argc, char** argv)
{
  if (argc < 3) return 2;
  const std::string what = argv[1];
  try {
    if (what == "rig") {
      for (const io::RigCamera& c : io::loadCameraRig(argv[2])) {
        printf("label %s\nsize %d %d\nintrinsics %.17g %.17g %.17g %.17g\ndistortion %d %.17g %.17g %.17g %.17g\n", c.label.c_str(),
               c.cam.width, c.cam.height, c.cam.fx, c.cam.fy, c.cam.cx, c.cam.cy, c.cam.distortion, c.cam.d[0], c.cam.d[1], c.cam.d[2], c.cam.d[3]);
        printf("T_B_C %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", c.T_B_C.q.w, c.T_B_C.q.x, c.T_B_C.q.y, c.T_B_C.q.z, c.T_B_C.t.x,
               c.T_B_C.t.y, c.T_B_C.t.z);
      }
    } else if (what == "params") {
      const io::FrontendParams p = std::string(argv[2]) == "-" ? io::frontendParamsFromYaml(io::YamlNode()) : io::loadFrontendParams(argv[2]);
      printf("img_align %d %d %d %d %d %d\n", p.img_align.max_level, p.img_align.min_level, (int)p.img_align.robustification,
             (int)p.img_align.use_distortion_jacobian, (int)p.img_align.estimate_illumination_gain, (int)p.img_align.estimate_illumination_offset);
      printf("prior %.17g %.17g\n", p.img_align_prior_lambda_rot, p.img_align_prior_lambda_trans);
      printf("reprojector %d %d %.17g %d %d\n", p.max_fts, p.grid_size, p.seed_sigma2_thresh, (int)p.reprojector_affine_est_offset,
             (int)p.reprojector_affine_est_gain);
      printf("depth_filter %d %.17g %.17g %d %d %d %d\n", (int)p.depth_filter.use_threaded_depthfilter, p.depth_filter.seed_convergence_sigma2_thresh,
             p.depth_filter.mappoint_convergence_sigma2_thresh, (int)p.depth_filter.scan_epi_unit_sphere, (int)p.depth_filter.affine_est_offset,
             (int)p.depth_filter.affine_est_gain, p.max_n_seeds_per_frame);
      printf("detector %zu %d %.17g %.17g %d\n", p.detector.cell_size, p.detector.max_level, p.detector.threshold_primary,
             p.detector.threshold_secondary, p.detector.detector_type == DetectorType::kFastGrad);
      printf("tracker %d %d pyr %d\n", p.tracker.klt_max_level, p.tracker.klt_min_level, p.n_pyr_levels_to_build);
    } else if (what == "png") {
      const io::GrayImage img = io::readPngGray(argv[2]);
      printf("size %d %d\n", img.width, img.height);
      unsigned long long sum = 0, wsum = 0;
      for (size_t i = 0; i < img.data.size(); ++i) { sum += img.data[i]; wsum += (unsigned long long)img.data[i] * (i % 251 + 1); }
      printf("sum %llu wsum %llu\n", sum, wsum);
      if (argc > 3) { FILE* f = fopen(argv[3], "wb"); fwrite(img.data.data(), 1, img.data.size(), f); fclose(f); }
    } else if (what == "euroc") {
      const io::EurocSequence s = io::openEuroc(argv[2]);
      printf("n %zu\n", s.size());
      for (size_t i = 0; i < s.size(); ++i) printf("frame %llu %s\n", (unsigned long long)s.cam_ts[i], s.cam0_files[i].c_str());
    } else if (what == "imu") {
      // imu <dataset_root> <t_old> <t_new> <max_dt>: the gyroscope prior between two camera times
      const std::vector<io::ImuMeasurement> imu = io::readEurocImu(argv[2]);
      const double bias[3] = { argc > 6 ? atof(argv[6]) : 0.0, argc > 7 ? atof(argv[7]) : 0.0, argc > 8 ? atof(argv[8]) : 0.0 };
      svoh::Quat q;
      const bool ok = io::relativeRotationPrior(imu, atof(argv[3]), atof(argv[4]), bias, 0.0, atof(argv[5]), &q);
      printf("n %zu\nok %d\nq %.17g %.17g %.17g %.17g\n", imu.size(), ok ? 1 : 0, q.w, q.x, q.y, q.z);
    } else if (what == "yaml") {
      const io::YamlNode n = io::loadYamlFile(argv[2]);
      printf("a.b.c %d\nlist %zu %.17g\nseq %zu %s %s\nstr %s\nkeyslash %d\nmissing %d\n", n["a"]["b"]["c"].asInt(-1), n["list"].size(),
             n["list"][2].asDouble(0), n["items"].size(), n["items"][0]["name"].asString("?").c_str(), n["items"][1]["name"].asString("?").c_str(),
             n["text"].asString("?").c_str(), n["T_world_imuinit/qw"].asInt(-1), n["nope"]["deeper"].asInt(-7));
    } else if (what == "traj") {
      io::TrajectoryWriter w(argv[2]);
      w.write(1403636579763555584ull, Transformation{ { 0.5, 0.5, -0.5, 0.5 }, { 1.0, -2.0, 3.25 } });
    } else return 2;
  } catch (const std::exception& e) {
    printf("error %s\n", e.what());
    return 1;
  }
  return 0;
}
