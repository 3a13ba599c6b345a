// svo_hip_io.h -- dataset / configuration plumbing for running the HIP front end on recorded
// sequences without OpenCV or yaml-cpp.  Counterparts in the reference:
//   examples/dataset/euroc.hpp:195-220      EuRoC folder layout, cam0/data.csv -> timestamps + image paths
//   examples/euroc_mono.cpp:11-62            the runner loop
//   src/svo/src/svo_factory.cpp:107-310      option keys and defaults read from the parameter YAML
//   examples/param/calib/*.yaml              NCamera calibration (vi::NCamera::loadFromYaml)
// Images: 8-bit PNG, grey or colour (colour is reduced like cv::imread + cv::cvtColor(BGR2GRAY),
// frame.cpp:72: fixed-point 0.299 R + 0.587 G + 0.114 B; identical to the input for grey files).
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "svo_hip_host.h"

namespace svo_hip {
namespace io {

// ---- YAML subset: block maps, block sequences ("- key: v"), flow sequences "[a, b]" (may span
// lines), scalars, comments.  Enough for the reference's parameter and calibration files. ----
struct YamlNode {
  enum Kind { kNull, kScalar, kMap, kSeq } kind = kNull;
  std::string scalar;
  std::vector<std::pair<std::string, YamlNode>> map;   // insertion order kept
  std::vector<YamlNode> seq;
  bool has(const std::string& key) const;
  const YamlNode& operator[](const std::string& key) const;   // null node if absent (yaml-cpp style)
  const YamlNode& operator[](size_t i) const { return seq.at(i); }
  size_t size() const { return kind == kSeq ? seq.size() : map.size(); }
  bool isNull() const { return kind == kNull; }
  // node["key"].as<T>(fallback) of yaml-cpp
  double asDouble(double fallback) const;
  int asInt(int fallback) const;
  bool asBool(bool fallback) const;
  std::string asString(const std::string& fallback) const;
  std::vector<double> asDoubles() const;   // flow / block sequence of numbers
};
YamlNode parseYaml(const std::string& text);
YamlNode loadYamlFile(const std::string& path);

// ---- camera rig ----
struct RigCamera {
  std::string label;
  svoh_camera cam{};
  Transformation T_B_C{ { 1, 0, 0, 0 }, { 0, 0, 0 } };   // body <- camera (T_imu_cam)
};
std::vector<RigCamera> loadCameraRig(const std::string& calib_yaml_path);
std::vector<RigCamera> cameraRigFromYaml(const YamlNode& root);

// ---- the front-end options of svo_factory.cpp, for the parts this library implements ----
struct FrontendParams {
  SparseImgAlignOptions img_align;          // img_align_* (svo_factory.cpp:137-146)
  double img_align_prior_lambda_rot = 0.0, img_align_prior_lambda_trans = 0.0;
  int max_fts = 160;                         // reprojector: max_fts (svo_factory.cpp:207)
  int grid_size = 35;                        // grid_size
  double seed_sigma2_thresh = 200.0;         // seed_convergence_sigma2_thresh
  bool reprojector_affine_est_offset = true, reprojector_affine_est_gain = false;
  DepthFilterOptions depth_filter;           // depth filter keys (svo_factory.cpp:254-263)
  int max_n_seeds_per_frame = 360;           // max_fts * max_seeds_ratio
  DetectorOptions detector;                  // grid_size, n_pyr_levels, detector_threshold_*, use_edgelets
  FeatureTrackerOptions tracker;             // klt_* (svo_factory.cpp:305-306)
  int n_pyr_levels_to_build = 5;             // img_align_max_level + 1 (frame_handler_base.cpp:186)
  int structure_optimization_max_pts = 20;   // structure_optimization_max_pts (svo_factory.cpp:122)
};
FrontendParams frontendParamsFromYaml(const YamlNode& node);
FrontendParams loadFrontendParams(const std::string& param_yaml_path);

// ---- images ----
struct GrayImage { int width = 0, height = 0; std::vector<uint8_t> data; };
GrayImage readPngGray(const std::string& path);                     // throws std::runtime_error
GrayImage decodePngGray(const uint8_t* bytes, size_t n_bytes);

// ---- EuRoC ASL folder (<root>/mav0/cam0/data.csv, <root>/mav0/cam0/data/<timestamp>.png) ----
struct EurocSequence {
  std::string mav_dir;
  std::vector<uint64_t> cam_ts;             // nanoseconds
  std::vector<std::string> cam0_files, cam1_files;
  size_t size() const { return cam_ts.size(); }
};
EurocSequence openEuroc(const std::string& dataset_root);

// ---- EuRoC IMU (<root>/mav0/imu0/data.csv: "timestamp [ns], w_x, w_y, w_z [rad/s], a_x, a_y, a_z [m/s^2]") ----
struct ImuMeasurement { double t = 0.0; double w[3] = { 0, 0, 0 }; double a[3] = { 0, 0, 0 }; };   // t in seconds
std::vector<ImuMeasurement> readEurocImu(const std::string& dataset_root);   // ascending time; empty when the file is absent

// ImuHandler::getRelativeRotationPrior (src/svo/src/imu_handler.cpp:270-297 over getMeasurements, :157-233): the
// gyroscope integrated between two camera timestamps, R_oldimu_newimu = prod exp((omega_j - bias) dt_j) over the
// measurements from the newest one at or before t_old (its time moved to t_old) to the newest one before t_new (which
// counts up to t_new); camera times are shifted by delay_imu_cam first.  False (and identity) where the reference
// returns false: no measurement at or before t_old, none before t_new, both the same, or the newest one older than
// max_imu_delta_t.  `imu` in ascending time.
bool relativeRotationPrior(const std::vector<ImuMeasurement>& imu, double t_old_cam, double t_new_cam, const double gyro_bias[3],
                           double delay_imu_cam, double max_imu_delta_t, svoh::Quat* R_oldimu_newimu);

// ---- trajectory output, TUM format: "timestamp tx ty tz qx qy qz qw" (seconds) ----
class TrajectoryWriter {
 public:
  explicit TrajectoryWriter(const std::string& path);
  ~TrajectoryWriter();
  void write(uint64_t timestamp_ns, const Transformation& T_world_body);
 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

}  // namespace io
}  // namespace svo_hip
