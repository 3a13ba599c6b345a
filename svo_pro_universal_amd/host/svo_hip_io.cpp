#include "svo_hip_io.h"

#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace svo_hip {
namespace io {

// ---------------------------------------------------------------------------
// YAML subset
// ---------------------------------------------------------------------------
namespace {
const YamlNode kNullNode;

struct Line { int indent; std::string text; };

std::string strip(const std::string& s)
{
  size_t a = 0, b = s.size();
  while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) ++a;
  while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) --b;
  return s.substr(a, b - a);
}

std::string strip_comment(const std::string& s)
{
  bool in_s = false, in_d = false;
  for (size_t i = 0; i < s.size(); ++i) {
    const char c = s[i];
    if (c == '\'' && !in_d) in_s = !in_s;
    else if (c == '"' && !in_s) in_d = !in_d;
    else if (c == '#' && !in_s && !in_d && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
  }
  return s;
}

std::string unquote(const std::string& s)
{
  if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
  return s;
}

int bracket_balance(const std::string& s)
{
  int b = 0;
  for (char c : s) { if (c == '[') ++b; else if (c == ']') --b; }
  return b;
}

YamlNode parse_scalar_or_flow(const std::string& raw)
{
  const std::string v = strip(raw);
  YamlNode n;
  if (v.empty() || v == "~" || v == "null") return n;
  if (v.front() == '[') {
    n.kind = YamlNode::kSeq;
    const std::string inner = v.substr(1, v.rfind(']') == std::string::npos ? std::string::npos : v.rfind(']') - 1);
    std::string item;
    int depth = 0;
    for (size_t i = 0; i <= inner.size(); ++i) {
      const char c = i < inner.size() ? inner[i] : ',';
      if (c == '[') ++depth;
      if (c == ']') --depth;
      if (c == ',' && depth == 0) {
        if (!strip(item).empty()) n.seq.push_back(parse_scalar_or_flow(item));
        item.clear();
      } else item.push_back(c);
    }
    return n;
  }
  n.kind = YamlNode::kScalar;
  n.scalar = unquote(v);
  return n;
}

// splits "key: value" at the first ": " (or trailing ':') outside quotes; returns false if the line is no mapping entry
bool split_key(const std::string& t, std::string& key, std::string& value)
{
  bool in_s = false, in_d = false;
  for (size_t i = 0; i < t.size(); ++i) {
    const char c = t[i];
    if (c == '\'' && !in_d) in_s = !in_s;
    else if (c == '"' && !in_s) in_d = !in_d;
    else if (c == ':' && !in_s && !in_d && (i + 1 == t.size() || t[i + 1] == ' ' || t[i + 1] == '\t')) {
      key = unquote(strip(t.substr(0, i)));
      value = strip(t.substr(i + 1));
      return !key.empty() && key.front() != '[';
    }
  }
  return false;
}

YamlNode parse_block(const std::vector<Line>& L, size_t& i, int indent);

YamlNode parse_value_after_key(const std::vector<Line>& L, size_t& i, int key_indent, const std::string& value)
{
  if (!value.empty()) return parse_scalar_or_flow(value);
  // nested block (deeper indent), or a block sequence at the same indent as the key ("key:\n- a")
  if (i < L.size() && (L[i].indent > key_indent || (L[i].indent == key_indent && L[i].text.compare(0, 1, "-") == 0 &&
                                                     (L[i].text.size() == 1 || L[i].text[1] == ' '))))
    return parse_block(L, i, L[i].indent);
  return YamlNode();
}

YamlNode parse_block(const std::vector<Line>& L, size_t& i, int indent)
{
  YamlNode node;
  const bool is_seq = L[i].text[0] == '-' && (L[i].text.size() == 1 || L[i].text[1] == ' ');
  node.kind = is_seq ? YamlNode::kSeq : YamlNode::kMap;
  while (i < L.size() && L[i].indent == indent) {
    const std::string& t = L[i].text;
    const bool dash = t[0] == '-' && (t.size() == 1 || t[1] == ' ');
    if (dash != is_seq) break;
    if (is_seq) {
      const std::string rest = strip(t.substr(1));
      const int inner_indent = indent + 1 + (int)(t.size() - 1 - strip(t.substr(1)).size() > 0 ? t.find_first_not_of(' ', 1) - 1 : 1);
      std::string key, value;
      ++i;
      if (rest.empty()) {
        node.seq.push_back(i < L.size() && L[i].indent > indent ? parse_block(L, i, L[i].indent) : YamlNode());
      } else if (split_key(rest, key, value)) {
        // "- key: value" opens a map whose further keys are indented to the column of `key`
        YamlNode m;
        m.kind = YamlNode::kMap;
        m.map.emplace_back(key, parse_value_after_key(L, i, inner_indent, value));
        while (i < L.size() && L[i].indent == inner_indent && !(L[i].text[0] == '-' && (L[i].text.size() == 1 || L[i].text[1] == ' '))) {
          std::string k2, v2;
          if (!split_key(L[i].text, k2, v2)) throw std::runtime_error("yaml: expected 'key: value' in '" + L[i].text + "'");
          ++i;
          m.map.emplace_back(k2, parse_value_after_key(L, i, inner_indent, v2));
        }
        node.seq.push_back(m);
      } else {
        node.seq.push_back(parse_scalar_or_flow(rest));
      }
    } else {
      std::string key, value;
      if (!split_key(t, key, value)) throw std::runtime_error("yaml: expected 'key: value' in '" + t + "'");
      ++i;
      node.map.emplace_back(key, parse_value_after_key(L, i, indent, value));
    }
  }
  return node;
}
}  // namespace

bool YamlNode::has(const std::string& key) const
{
  for (const auto& kv : map) if (kv.first == key) return true;
  return false;
}
const YamlNode& YamlNode::operator[](const std::string& key) const
{
  for (const auto& kv : map) if (kv.first == key) return kv.second;
  return kNullNode;
}
double YamlNode::asDouble(double fallback) const
{
  if (kind != kScalar) return fallback;
  char* end = nullptr;
  const double v = strtod(scalar.c_str(), &end);
  return (end && *end == 0 && end != scalar.c_str()) ? v : fallback;
}
int YamlNode::asInt(int fallback) const
{
  if (kind != kScalar) return fallback;
  char* end = nullptr;
  const long v = strtol(scalar.c_str(), &end, 10);
  if (end && *end == 0 && end != scalar.c_str()) return (int)v;
  const double d = asDouble(NAN);
  return d == d ? (int)d : fallback;
}
bool YamlNode::asBool(bool fallback) const
{
  if (kind != kScalar) return fallback;
  std::string s = scalar;
  for (char& c : s) c = (char)tolower(c);
  if (s == "true" || s == "yes" || s == "on" || s == "1") return true;
  if (s == "false" || s == "no" || s == "off" || s == "0") return false;
  return fallback;
}
std::string YamlNode::asString(const std::string& fallback) const { return kind == kScalar ? scalar : fallback; }
std::vector<double> YamlNode::asDoubles() const
{
  std::vector<double> v;
  for (const YamlNode& n : seq) v.push_back(n.asDouble(NAN));
  return v;
}

YamlNode parseYaml(const std::string& text)
{
  std::vector<Line> L;
  std::istringstream in(text);
  std::string raw, pending;
  int pending_indent = 0;
  while (std::getline(in, raw)) {
    std::string s = strip_comment(raw);
    if (strip(s).empty() || strip(s) == "---" || strip(s).compare(0, 1, "%") == 0) continue;
    if (!pending.empty()) {   // continuation of a flow sequence
      pending += " " + strip(s);
      if (bracket_balance(pending) <= 0) { L.push_back(Line{ pending_indent, pending }); pending.clear(); }
      continue;
    }
    int indent = 0;
    while (indent < (int)s.size() && s[indent] == ' ') ++indent;
    const std::string t = strip(s);
    if (bracket_balance(t) > 0) { pending = t; pending_indent = indent; continue; }
    L.push_back(Line{ indent, t });
  }
  if (!pending.empty()) throw std::runtime_error("yaml: unterminated '[' in '" + pending + "'");
  if (L.empty()) return YamlNode();
  size_t i = 0;
  YamlNode root = parse_block(L, i, L[0].indent);
  if (i != L.size()) throw std::runtime_error("yaml: unexpected indentation at '" + L[i].text + "'");
  return root;
}

static std::string read_file(const std::string& path)
{
  std::ifstream f(path, std::ios::binary);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::ostringstream ss;
  ss << f.rdbuf();
  return ss.str();
}

YamlNode loadYamlFile(const std::string& path) { return parseYaml(read_file(path)); }

// ---------------------------------------------------------------------------
// camera rig (vi::NCamera::loadFromYaml: cameras[i].camera.{image_width,image_height,type,intrinsics.data,
// distortion.{type,parameters.data}}, cameras[i].T_B_C.data row-major 4x4)
// ---------------------------------------------------------------------------
static svoh::Quat quat_from_R(const double R[9])
{
  // Eigen::Quaterniond(Matrix3d) (Shepperd)
  svoh::Quat q;
  const double t = R[0] + R[4] + R[8];
  if (t > 0.0) {
    double s = std::sqrt(t + 1.0);
    q.w = 0.5 * s; s = 0.5 / s;
    q.x = (R[7] - R[5]) * s; q.y = (R[2] - R[6]) * s; q.z = (R[3] - R[1]) * s;
  } else {
    int i = 0;
    if (R[4] > R[0]) i = 1;
    if (R[8] > R[i * 4]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
    double v[3];
    v[i] = 0.5 * s; s = 0.5 / s;
    q.w = (R[k * 3 + j] - R[j * 3 + k]) * s;
    v[j] = (R[j * 3 + i] + R[i * 3 + j]) * s;
    v[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    q.x = v[0]; q.y = v[1]; q.z = v[2];
  }
  return q;
}

std::vector<RigCamera> cameraRigFromYaml(const YamlNode& root)
{
  std::vector<RigCamera> rig;
  const YamlNode& cams = root["cameras"];
  if (cams.kind != YamlNode::kSeq || cams.seq.empty()) throw std::runtime_error("calibration: no 'cameras' sequence");
  for (const YamlNode& entry : cams.seq) {
    const YamlNode& c = entry["camera"];
    if (c.isNull()) throw std::runtime_error("calibration: entry without 'camera'");
    RigCamera rc;
    rc.label = c["label"].asString("cam");
    rc.cam.width = c["image_width"].asInt(0);
    rc.cam.height = c["image_height"].asInt(0);
    if (c["type"].asString("pinhole") != "pinhole") throw std::runtime_error("calibration: only pinhole cameras are supported");
    const std::vector<double> k = c["intrinsics"]["data"].asDoubles();
    if (k.size() != 4 || rc.cam.width <= 0 || rc.cam.height <= 0) throw std::runtime_error("calibration: bad intrinsics / image size");
    rc.cam.fx = k[0]; rc.cam.fy = k[1]; rc.cam.cx = k[2]; rc.cam.cy = k[3];
    const YamlNode& dist = c["distortion"];
    const std::string dtype = dist["type"].asString("none");
    if (dtype == "radial-tangential") {
      const std::vector<double> d = dist["parameters"]["data"].asDoubles();
      if (d.size() != 4) throw std::runtime_error("calibration: radial-tangential needs 4 parameters");
      for (int i = 0; i < 4; ++i) rc.cam.d[i] = d[i];
      rc.cam.distortion = SVOH_DISTORTION_RADTAN;
    } else if (dtype == "none" || dist.isNull()) {
      rc.cam.distortion = SVOH_DISTORTION_NONE;
    } else {
      throw std::runtime_error("calibration: unsupported distortion '" + dtype + "'");
    }
    const std::vector<double> T = entry["T_B_C"]["data"].asDoubles();
    if (T.size() == 16) {
      const double R[9] = { T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10] };
      rc.T_B_C.q = quat_from_R(R);
      rc.T_B_C.t = { T[3], T[7], T[11] };
    } else if (!entry["T_B_C"].isNull()) {
      throw std::runtime_error("calibration: T_B_C needs 16 numbers");
    }
    rig.push_back(rc);
  }
  return rig;
}
std::vector<RigCamera> loadCameraRig(const std::string& path) { return cameraRigFromYaml(loadYamlFile(path)); }

// ---------------------------------------------------------------------------
// svo_factory.cpp:107-310, the keys this library consumes
// ---------------------------------------------------------------------------
FrontendParams frontendParamsFromYaml(const YamlNode& node)
{
  FrontendParams p;
  p.img_align.max_level = node["img_align_max_level"].asInt(4);
  p.img_align.min_level = node["img_align_min_level"].asInt(2);
  p.img_align.robustification = node["img_align_robustification"].asBool(false);
  p.img_align.use_distortion_jacobian = node["img_align_use_distortion_jacobian"].asBool(false);
  p.img_align.estimate_illumination_gain = node["img_align_est_illumination_gain"].asBool(false);
  p.img_align.estimate_illumination_offset = node["img_align_est_illumination_offset"].asBool(false);
  p.img_align_prior_lambda_rot = node["img_align_prior_lambda_rot"].asDouble(0.0);
  p.img_align_prior_lambda_trans = node["img_align_prior_lambda_trans"].asDouble(0.0);
  p.n_pyr_levels_to_build = p.img_align.max_level + 1;
  p.max_fts = node["max_fts"].asInt(160);
  p.grid_size = node["grid_size"].asInt(35);
  p.seed_sigma2_thresh = node["seed_convergence_sigma2_thresh"].asDouble(200.0);
  p.reprojector_affine_est_offset = node["reprojector_affine_est_offset"].asBool(true);
  p.reprojector_affine_est_gain = node["reprojector_affine_est_gain"].asBool(false);
  p.depth_filter.use_threaded_depthfilter = node["use_threaded_depthfilter"].asBool(true);
  p.depth_filter.seed_convergence_sigma2_thresh = node["seed_convergence_sigma2_thresh"].asDouble(200.0);
  p.depth_filter.mappoint_convergence_sigma2_thresh = node["mappoint_convergence_sigma2_thresh"].asDouble(500.0);
  p.depth_filter.scan_epi_unit_sphere = node["scan_epi_unit_sphere"].asBool(false);
  p.depth_filter.affine_est_offset = node["depth_filter_affine_est_offset"].asBool(true);
  p.depth_filter.affine_est_gain = node["depth_filter_affine_est_gain"].asBool(false);
  p.max_n_seeds_per_frame = (int)((double)node["max_fts"].asInt(120) * node["max_seeds_ratio"].asDouble(3.0));
  p.detector.cell_size = (size_t)node["grid_size"].asInt(35);
  p.detector.max_level = node["n_pyr_levels"].asInt(3) - 1;
  p.detector.threshold_primary = node["detector_threshold_primary"].asInt(10);
  p.detector.threshold_secondary = node["detector_threshold_secondary"].asInt(200);
  p.detector.detector_type = node["use_edgelets"].asBool(true) ? DetectorType::kFastGrad : DetectorType::kFast;
  p.structure_optimization_max_pts = node["structure_optimization_max_pts"].asInt(20);
  p.tracker.klt_max_level = node["klt_max_level"].asInt(4);
  p.tracker.klt_min_level = node["klt_min_level"].asInt(0);
  return p;
}
FrontendParams loadFrontendParams(const std::string& path) { return frontendParamsFromYaml(loadYamlFile(path)); }

// ---------------------------------------------------------------------------
// PNG (8-bit, non-interlaced; grey, grey+alpha, RGB, RGBA)
// ---------------------------------------------------------------------------
static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

GrayImage decodePngGray(const uint8_t* b, size_t n)
{
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
  if (n < 8 || memcmp(b, sig, 8) != 0) throw std::runtime_error("png: bad signature");
  size_t pos = 8;
  int w = 0, h = 0, bit_depth = 0, color_type = -1, interlace = 0;
  std::vector<uint8_t> idat;
  bool end = false;
  while (!end && pos + 12 <= n) {
    const uint32_t len = be32(b + pos);
    const uint8_t* type = b + pos + 4;
    if (pos + 12 + (size_t)len > n) throw std::runtime_error("png: truncated chunk");
    const uint8_t* data = b + pos + 8;
    if (crc32(crc32(0L, Z_NULL, 0), type, len + 4) != be32(data + len)) throw std::runtime_error("png: chunk CRC mismatch");
    if (!memcmp(type, "IHDR", 4)) {
      if (len != 13) throw std::runtime_error("png: bad IHDR");
      w = (int)be32(data); h = (int)be32(data + 4); bit_depth = data[8]; color_type = data[9]; interlace = data[12];
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (w <= 0 || h <= 0 || !end) throw std::runtime_error("png: missing IHDR / IEND");
  if (bit_depth != 8 || interlace != 0) throw std::runtime_error("png: only 8-bit non-interlaced images are supported");
  int ch;
  switch (color_type) { case 0: ch = 1; break; case 2: ch = 3; break; case 4: ch = 2; break; case 6: ch = 4; break;
    default: throw std::runtime_error("png: palette images are not supported"); }
  // The header is not trusted with an allocation: a damaged (or hostile) width / height of 2^31 would ask for exabytes
  // before a single byte of pixel data has been looked at (found by tests/san/io_fuzz).  deflate expands by at most
  // ~1032 : 1, so the scanlines the header promises must fit what the IDAT chunks can possibly hold.
  if (w > (1 << 20) || h > (1 << 20)) throw std::runtime_error("png: image larger than 2^20 pixels a side");
  const size_t stride = (size_t)w * ch;
  if ((stride + 1) > ((size_t)idat.size() * 1032 + 1024) / (size_t)h + 1) throw std::runtime_error("png: the header promises more pixels than the image data can hold");
  std::vector<uint8_t> raw((stride + 1) * (size_t)h);
  uLongf out_len = (uLongf)raw.size();
  if (uncompress(raw.data(), &out_len, idat.data(), (uLong)idat.size()) != Z_OK || out_len != raw.size())
    throw std::runtime_error("png: inflate failed");
  // undo the scanline filters in place
  std::vector<uint8_t> prev(stride, 0), cur(stride);
  GrayImage img;
  img.width = w; img.height = h; img.data.resize((size_t)w * h);
  for (int y = 0; y < h; ++y) {
    const uint8_t* line = raw.data() + (size_t)y * (stride + 1);
    const int ft = line[0];
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= (size_t)ch ? cur[i - ch] : 0, bb = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
      int v = line[1 + i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += bb; break;
        case 3: v += (a + bb) >> 1; break;
        case 4: { const int p = a + bb - c, pa = abs(p - a), pb = abs(p - bb), pc = abs(p - c);
                  v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? bb : c); break; }
        default: throw std::runtime_error("png: bad filter type");
      }
      cur[i] = (uint8_t)v;
    }
    uint8_t* dst = img.data.data() + (size_t)y * w;
    if (ch <= 2) for (int x = 0; x < w; ++x) dst[x] = cur[(size_t)x * ch];
    else for (int x = 0; x < w; ++x) {   // cv::cvtColor BGR2GRAY, 8U: (R*4899 + G*9617 + B*1868 + 8192) >> 14
      const uint8_t* px = &cur[(size_t)x * ch];
      dst[x] = (uint8_t)((px[0] * 4899 + px[1] * 9617 + px[2] * 1868 + 8192) >> 14);
    }
    prev.swap(cur);
  }
  return img;
}

GrayImage readPngGray(const std::string& path)
{
  const std::string s = read_file(path);
  try { return decodePngGray(reinterpret_cast<const uint8_t*>(s.data()), s.size()); }
  catch (const std::exception& e) { throw std::runtime_error(path + ": " + e.what()); }
}

// ---------------------------------------------------------------------------
// EuRoC (examples/dataset/euroc.hpp:195-220)
// ---------------------------------------------------------------------------
EurocSequence openEuroc(const std::string& root)
{
  EurocSequence seq;
  seq.mav_dir = root + "/mav0";
  const std::string csv = seq.mav_dir + "/cam0/data.csv";
  std::ifstream f(csv);
  if (!f) throw std::runtime_error("cannot open " + csv);
  std::string s;
  while (std::getline(f, s)) {
    for (char& c : s) if (c == ',') c = ' ';
    s = strip(s);
    if (s.empty() || s[0] == '#') continue;
    std::istringstream ss(s);
    uint64_t ts = 0;
    if (!(ss >> ts)) throw std::runtime_error(csv + ": bad line '" + s + "'");
    seq.cam_ts.push_back(ts);
    seq.cam0_files.push_back(seq.mav_dir + "/cam0/data/" + std::to_string(ts) + ".png");
    seq.cam1_files.push_back(seq.mav_dir + "/cam1/data/" + std::to_string(ts) + ".png");
  }
  return seq;
}

// ---------------------------------------------------------------------------
std::vector<ImuMeasurement> readEurocImu(const std::string& root)
{
  std::vector<ImuMeasurement> out;
  const std::string csv = root + "/mav0/imu0/data.csv";
  std::ifstream f(csv);
  if (!f) return out;
  std::string s;
  while (std::getline(f, s)) {
    for (char& c : s) if (c == ',') c = ' ';
    s = strip(s);
    if (s.empty() || s[0] == '#') continue;
    std::istringstream ss(s);
    unsigned long long ts = 0;
    ImuMeasurement m;
    if (!(ss >> ts >> m.w[0] >> m.w[1] >> m.w[2] >> m.a[0] >> m.a[1] >> m.a[2])) throw std::runtime_error(csv + ": bad line '" + s + "'");
    m.t = (double)ts * 1e-9;
    if (!out.empty() && m.t < out.back().t) throw std::runtime_error(csv + ": timestamps not ascending");
    out.push_back(m);
  }
  return out;
}

bool relativeRotationPrior(const std::vector<ImuMeasurement>& imu, double t_old_cam, double t_new_cam, const double gyro_bias[3],
                           double delay_imu_cam, double max_imu_delta_t, svoh::Quat* R)
{
  *R = svoh::Quat{ 1.0, 0.0, 0.0, 0.0 };
  if (imu.empty() || !(t_new_cam > t_old_cam)) return false;
  const double t1 = t_old_cam - delay_imu_cam, t2 = t_new_cam - delay_imu_cam;
  // the reference walks its list from the newest measurement: it2 = the first with timestamp < t2, it1 = the first with
  // timestamp <= t1 (imu_handler.cpp:180-193)
  long i1 = -1, i2 = -1;
  for (long i = (long)imu.size() - 1; i >= 0; --i) {
    if (i2 < 0 && imu[(size_t)i].t < t2) i2 = i;
    if (imu[(size_t)i].t <= t1) { i1 = i; break; }
  }
  if (i1 < 0 || i2 < 0 || i1 == i2) return false;
  if (t2 - imu[(size_t)i2].t > max_imu_delta_t) return false;
  svoh::Quat q{ 1.0, 0.0, 0.0, 0.0 };
  for (long j = i1; j <= i2; ++j) {
    const double tj = j == i1 ? t1 : imu[(size_t)j].t;                        // "change timestamp of oldest measurement"
    const double dt = j == i2 ? t2 - imu[(size_t)j].t : imu[(size_t)j + 1].t - tj;   // the newest one counts up to the new frame
    const svoh::Vec3 w{ (imu[(size_t)j].w[0] - gyro_bias[0]) * dt, (imu[(size_t)j].w[1] - gyro_bias[1]) * dt, (imu[(size_t)j].w[2] - gyro_bias[2]) * dt };
    q = svoh::mul(q, svoh::quat_exp(w));
  }
  *R = q;
  return true;
}

// ---------------------------------------------------------------------------
struct TrajectoryWriter::Impl { FILE* f = nullptr; };
TrajectoryWriter::TrajectoryWriter(const std::string& path) : impl_(new Impl)
{
  impl_->f = fopen(path.c_str(), "w");
  if (!impl_->f) throw std::runtime_error("cannot write " + path);
  fprintf(impl_->f, "# timestamp tx ty tz qx qy qz qw\n");
}
TrajectoryWriter::~TrajectoryWriter() { if (impl_ && impl_->f) fclose(impl_->f); }
void TrajectoryWriter::write(uint64_t ts_ns, const Transformation& T)
{
  fprintf(impl_->f, "%llu.%09llu %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n", (unsigned long long)(ts_ns / 1000000000ull),
          (unsigned long long)(ts_ns % 1000000000ull), T.t.x, T.t.y, T.t.z, T.q.x, T.q.y, T.q.z, T.q.w);
}

}  // namespace io
}  // namespace svo_hip
