// svoh_mini_stereo -- the per-frame chain of FrameHandlerStereo::processFrame (src/svo/src/frame_handler_stereo.cpp:90-205,
// frame_handler_base.cpp:610-825) assembled from this library's mirrors on an EuRoC-layout STEREO sequence: BASELINE
// config 3 (stereo + IMU rotation prior, illumination gain and offset estimated) end to end.
//
//   first pair: stereo triangulation of new features                StereoTriangulationHip::compute  (StereoInit)
//   sparse image alignment of the bundle (2 cameras, 8 parameters:  SparseImgAlignHip::run, setWeightedPrior with the
//     pose + illumination gain / offset) with the rotation prior      IMU's relative rotation (frame_handler_base.cpp:619-631)
//   reprojection of the keyframes' landmarks / seeds, per camera    ReprojectorHip::reprojectFrames
//   pose optimisation of the rig on both cameras' matches           PoseOptimizerHip::run
//   depth-filter update of the keyframes' seeds, per camera         DepthFilterHip::updateSeeds
//   at keyframes: stereo triangulation again + new seeds            StereoTriangulationHip::compute, initializeSeeds
//
// A harness over the built mirrors, NOT FrameHandlerStereo: no map (every live keyframe counts as overlapping), keyframes by a
// fixed rule, the first rig pose given.  Round 6: a keyframe pair upgrades the seeds its frames hang on to landmarks
// (upgradeSeedsToFeatures) and every pair's landmarks go through optimizeStructure, as the frame handler does
// (SVOH_MINI_LANDMARKS=0: without).  The IMU prior is read
// from <dataset_root>/mav0/imu_prior.csv (one line per frame: qw qx qy qz of R_imu(k)_imu(k-1)), or, without that file,
// integrated from the raw gyroscope of an EuRoC folder (mav0/imu0/data.csv, io::relativeRotationPrior); with neither no
// prior is set.
//
//   svoh_mini_stereo <dataset_root> <calib.yaml (two cameras)> <params.yaml|-> <out_dir> <T_imu_world of frame 0: qw qx qy qz tx ty tz>
//                    [max_frames] [kf_every] [prior_lambda_rot] [n_streams] [n_workers] [n_groups]
// Writes <out>/trajectory.txt (TUM format, T_world_imu) and <out>/frontend.csv.
// n_streams given (round 6): that many stereo streams in LOCK STEP on one context (FrontendLockstepStereo, host/svo_hip_lockstep_stereo.h):
// one pair of every stream at a time, every per-pair stage one launch for all of them; stream k writes into <out>/stream<k>/ (k > 0).
// SVOH_MINI_STEREO_ROOTS=<root>:<root>:... gives the streams DIFFERENT sequences (stream k replays root k % n; a root's first pose is read
// from <root>/T0.txt, seven numbers, its priors from its own mav0/imu_prior.csv): every stream must write what the single-stream run of ITS
// root writes, byte for byte (tests/test_mini_stereo_gpu.py).
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "../svo_pro_universal_amd/host/svo_hip_io.h"
#include "../svo_pro_universal_amd/host/svo_hip_lockstep_stereo.h"

using namespace svo_hip;

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// the rotation priors of a dataset root: mav0/imu_prior.csv (one line per frame: qw qx qy qz of R_imu(k)_imu(k-1))
static void read_priors(const std::string& root, std::vector<svoh::Quat>* imu_prior, std::vector<bool>* have_prior)
{
  std::ifstream in(root + "/mav0/imu_prior.csv");
  std::string line;
  while (std::getline(in, line)) {
    if (line.empty() || line[0] == '#') continue;
    for (char& c : line) if (c == ',') c = ' ';
    std::istringstream ss(line);
    svoh::Quat q{ 1, 0, 0, 0 };
    const bool parsed = static_cast<bool>(ss >> q.w >> q.x >> q.y >> q.z);
    imu_prior->push_back(parsed ? q : svoh::Quat{ 1, 0, 0, 0 });
    have_prior->push_back(parsed);
  }
}

// n_streams stereo streams in lock step (FrontendLockstepStereo); stream k replays root k % n_roots
struct StereoRoot { io::EurocSequence seq; std::vector<io::GrayImage> left, right; std::vector<svoh::Quat> prior; std::vector<bool> have; Transformation T0; size_t n = 0;
                    std::vector<io::RigCamera> rig; /* <root>/calib.yaml when there is one: the root's own rig */ };
static std::vector<StereoRoot> load_roots(const std::vector<std::string>& roots, const Transformation& T0_default, size_t max_frames)
{
  std::vector<StereoRoot> data(roots.size());
  for (size_t r = 0; r < roots.size(); ++r) {
    StereoRoot& d = data[r];
    d.seq = io::openEuroc(roots[r]);
    if (d.seq.cam1_files.size() != d.seq.cam0_files.size()) throw std::runtime_error("two image folders are needed: " + roots[r]);
    d.n = std::min(d.seq.size(), max_frames);
    for (size_t k = 0; k < d.n; ++k) { d.left.push_back(io::readPngGray(d.seq.cam0_files[k])); d.right.push_back(io::readPngGray(d.seq.cam1_files[k])); }
    read_priors(roots[r], &d.prior, &d.have);
    d.T0 = T0_default;
    std::ifstream t0(roots[r] + "/T0.txt");
    double v[7];
    if (t0 >> v[0] >> v[1] >> v[2] >> v[3] >> v[4] >> v[5] >> v[6]) d.T0 = Transformation{ { v[0], v[1], v[2], v[3] }, { v[4], v[5], v[6] } };
    if (std::ifstream(roots[r] + "/calib.yaml").good()) d.rig = io::loadCameraRig(roots[r] + "/calib.yaml");
  }
  return data;
}

// one lock-step group: streams [s0, s0 + n_streams) of the run on a context and a thread of their own
struct StereoGroupResult { size_t pairs = 0; double ms = 0, round_ms = 0; int device_calls = 0; std::string error; };
static void run_lockstep(const std::vector<StereoRoot>& data, const std::vector<io::RigCamera>& rig, const io::FrontendParams& params, const std::string& out_dir,
                         size_t kf_every, double lambda_rot, int s0, int n_streams, int n_workers, std::atomic<int>* gate, int n_groups, StereoGroupResult* out)
try {
  svoh_ctx* ctx = nullptr;
  if (svoh_create(0, &ctx) != SVOH_OK) throw std::runtime_error(std::string("svoh_create: ") + svoh_last_error_string(nullptr));
  auto root_of = [&](int s) -> const StereoRoot& { return data[(size_t)(s0 + s) % data.size()]; };
  // the decoded pairs in page-locked memory, as a camera driver that feeds a GPU would deliver them: EVERY STREAM ITS OWN COPY (the device
  // reads the images in place; streams that shared a buffer would be served from its caches after the first)
  const size_t img_bytes = (size_t)data[0].left[0].width * data[0].left[0].height;
  std::vector<size_t> stream_off((size_t)n_streams + 1, 0);
  for (int s = 0; s < n_streams; ++s) stream_off[(size_t)s + 1] = stream_off[(size_t)s] + 2 * img_bytes * root_of(s).n;
  uint8_t* pinned = nullptr;
  if (svoh_host_alloc(ctx, stream_off[(size_t)n_streams], (void**)&pinned) != SVOH_OK) throw std::runtime_error(std::string("svoh_host_alloc: ") + svoh_last_error_string(ctx));
  for (int s = 0; s < n_streams; ++s) {
    const StereoRoot& d = root_of(s);
    for (size_t k = 0; k < d.n; ++k) {
      if (d.left[k].data.size() != img_bytes || d.right[k].data.size() != img_bytes) throw std::runtime_error("images of different sizes");
      memcpy(pinned + stream_off[(size_t)s] + (2 * k) * img_bytes, d.left[k].data.data(), img_bytes);
      memcpy(pinned + stream_off[(size_t)s] + (2 * k + 1) * img_bytes, d.right[k].data.data(), img_bytes);
    }
  }
  {
    StereoLockstepOptions lo;
    lo.params = params; lo.rig = rig; lo.kf_every = kf_every; lo.lambda_rot = lambda_rot; lo.n_workers = n_workers;
    lo.images_mem_space = SVOH_MEM_HOST_PINNED;
    {   // roots with a calibration of their own: a rig per stream (the command line's for the roots without)
      bool any = false;
      for (const StereoRoot& d : data) any = any || !d.rig.empty();
      if (any) for (int s = 0; s < n_streams; ++s) lo.per_stream_rig.push_back(root_of(s).rig.empty() ? rig : root_of(s).rig);
    }
    if (getenv("SVOH_MINI_LANDMARKS")) lo.landmarks = atoi(getenv("SVOH_MINI_LANDMARKS")) != 0;
    if (getenv("SVOH_LOCKSTEP_SPECULATE")) lo.speculate_all = std::string(getenv("SVOH_LOCKSTEP_SPECULATE")) == "all";
    if (getenv("SVOH_LOCKSTEP_RESIDENT")) lo.resident_features = atoi(getenv("SVOH_LOCKSTEP_RESIDENT")) != 0;   // (the test of the explicit-column batches)
    FrontendLockstepStereo fe(ctx, n_streams, lo);
    std::vector<std::unique_ptr<io::TrajectoryWriter>> traj;
    std::vector<FILE*> csv;
    for (int s = 0; s < n_streams; ++s) {
      const std::string dir = s0 + s == 0 ? out_dir : out_dir + "/stream" + std::to_string(s0 + s);
      traj.emplace_back(new io::TrajectoryWriter(dir + "/trajectory.txt"));
      FILE* fc = fopen((dir + "/frontend.csv").c_str(), "w");
      if (!fc) throw std::runtime_error("cannot write into " + dir);
      fprintf(fc, "frame,is_kf,n_aligned,n_reprojected,n_after_pose_opt,n_seeds_updated,n_landmarks,alpha,beta,ms_pyramid,ms_align,ms_reproject,ms_pose,ms_seeds,ms_kf,ms_pair\n");
      csv.push_back(fc);
    }
    auto write_rows = [&]() {
      for (int s = 0; s < n_streams; ++s)
        for (const FrontendLockstepStereo::PairRow& r : fe.completedRows(s))
          fprintf(csv[(size_t)s], "%zu,%d,%zu,%zu,%zu,%zu,%zu,%.6f,%.4f,0,0,0,0,0,0,0\n", r.k, (int)r.is_kf, r.n_aligned, r.n_reproj, r.n_pose, r.n_seed_upd, r.n_landmarks, r.alpha, r.beta);
    };
    size_t n_rounds = 0;
    for (int s = 0; s < n_streams; ++s) n_rounds = std::max(n_rounds, root_of(s).n);
    std::vector<const uint8_t*> left((size_t)n_streams), right((size_t)n_streams), next_left((size_t)n_streams), next_right((size_t)n_streams);
    const bool announce = !(getenv("SVOH_MINI_STEREO_PREFETCH") && atoi(getenv("SVOH_MINI_STEREO_PREFETCH")) == 0);   // (a replay knows its next pairs; =0: the test of the other path)
    std::vector<Transformation> T_first((size_t)n_streams);
    std::vector<const svoh::Quat*> prior((size_t)n_streams);
    double sum_ms = 0;
    size_t pairs = 0;
    std::vector<double> phase0((size_t)FrontendLockstepStereo::kNumPhases, 0.0), detail0((size_t)FrontendLockstepStereo::kNumDetails, 0.0);
    gate->fetch_add(1);   // all groups start their first round together
    while (gate->load() < n_groups) std::this_thread::yield();
    for (size_t k = 0; k < n_rounds; ++k) {
      size_t n_now = 0;
      for (int s = 0; s < n_streams; ++s) {
        const StereoRoot& d = root_of(s);
        const bool has = k < d.n;
        left[(size_t)s] = has ? pinned + stream_off[(size_t)s] + (2 * k) * img_bytes : nullptr;
        right[(size_t)s] = has ? pinned + stream_off[(size_t)s] + (2 * k + 1) * img_bytes : nullptr;
        T_first[(size_t)s] = d.T0;
        prior[(size_t)s] = has && k < d.prior.size() && d.have[k] ? &d.prior[k] : nullptr;
        n_now += has;
        const bool has_next = k + 1 < d.n;
        next_left[(size_t)s] = has_next ? pinned + stream_off[(size_t)s] + (2 * (k + 1)) * img_bytes : nullptr;
        next_right[(size_t)s] = has_next ? pinned + stream_off[(size_t)s] + (2 * (k + 1) + 1) * img_bytes : nullptr;
      }
      const double t0 = now_ms();
      const bool more = announce && k + 1 < n_rounds;
      fe.addPairs(left.data(), right.data(), data[0].left[0].width, T_first.data(), prior.data(), more ? next_left.data() : nullptr, more ? next_right.data() : nullptr);
      const double t1 = now_ms();
      if (k > 2) { sum_ms += t1 - t0; pairs += n_now; }   // (the first rounds pay the one-time costs)
      if (k == 2) {   // ... which the phase sums below leave out as well
        phase0.assign(fe.phaseTimes(), fe.phaseTimes() + FrontendLockstepStereo::kNumPhases);
        detail0.assign(fe.detailTimes(), fe.detailTimes() + FrontendLockstepStereo::kNumDetails);
      }
      for (int s = 0; s < n_streams; ++s) {
        const StereoRoot& d = root_of(s);
        if (k < d.n) traj[(size_t)s]->write(d.seq.cam_ts[k], svoh::inverse(fe.pose(s)));
      }
      write_rows();
    }
    fe.finish();
    write_rows();
    if (getenv("SVOH_LOCKSTEP_TIMING") && s0 == 0) {
      const double nr = n_rounds > 3 ? (double)(n_rounds - 3) : 1.0;   // (rounds 3 ..: the steady state the rate is quoted on)
      fprintf(stderr, "[lockstep stereo] mean ms per round:");
      for (int k = 0; k < FrontendLockstepStereo::kNumPhases; ++k) fprintf(stderr, " %s %.3f,", FrontendLockstepStereo::phaseName(k), (fe.phaseTimes()[k] - phase0[(size_t)k]) / nr);
      fprintf(stderr, " paused passes %zu\n", fe.pausedPasses());
      fprintf(stderr, "[lockstep stereo]   of which:");
      for (int k = 0; k < FrontendLockstepStereo::kNumDetails; ++k) fprintf(stderr, " %s %.3f,", FrontendLockstepStereo::detailName(k), (fe.detailTimes()[k] - detail0[(size_t)k]) / nr);
      fprintf(stderr, "\n");
    }
    for (FILE* f : csv) fclose(f);
    out->pairs = pairs; out->ms = sum_ms; out->round_ms = n_rounds > 3 ? sum_ms / (double)(n_rounds - 3) : 0.0; out->device_calls = fe.lastRoundDeviceCalls();
  }
  (void)svoh_host_free(ctx, pinned);
  svoh_destroy(ctx);
} catch (const std::exception& e) {
  out->error = e.what();
  gate->fetch_add(1);
}

int main(int argc, char** argv)
{
  if (argc < 12) {
    fprintf(stderr, "usage: %s <dataset_root> <calib.yaml> <params.yaml|-> <out_dir> qw qx qy qz tx ty tz [max_frames] [kf_every] [prior_lambda_rot]\n", argv[0]);
    return 2;
  }
  svoh_ctx* ctx = nullptr;
  try {
    const io::EurocSequence seq = io::openEuroc(argv[1]);
    const std::vector<io::RigCamera> rig = io::loadCameraRig(argv[2]);
    if (rig.size() != 2 || seq.cam1_files.size() != seq.cam0_files.size()) throw std::runtime_error("a stereo rig and two image folders are needed");
    // SVOH_MINI_STEREO_THRESHOLDS_OF=<calib.yaml>: the reference's process-wide static thresholds (pose_optimizer.cpp:211-212, depth_filter.cpp:383-384) are
    // taken from THAT rig's first camera -- the single-stream run of a stream that shares a lock-step engine with rigs of other focal lengths
    if (const char* e = getenv("SVOH_MINI_STEREO_THRESHOLDS_OF")) fixProcessWideThresholds(io::loadCameraRig(e).at(0).cam, 2.0);
    io::FrontendParams params = std::string(argv[3]) == "-" ? io::frontendParamsFromYaml(io::YamlNode()) : io::loadFrontendParams(argv[3]);
    const std::string out_dir = argv[4];
    const Transformation T_imu_world0{ { atof(argv[5]), atof(argv[6]), atof(argv[7]), atof(argv[8]) }, { atof(argv[9]), atof(argv[10]), atof(argv[11]) } };
    const size_t max_frames = argc > 12 ? (size_t)atol(argv[12]) : seq.size();
    const size_t kf_every = argc > 13 ? (size_t)atol(argv[13]) : 8;
    const double lambda_rot = argc > 14 ? atof(argv[14]) : 0.5;
    const size_t n_frames = std::min(seq.size(), max_frames);
    if (argc > 15) {   // lock step
      const int n_streams = atoi(argv[15]), n_workers = argc > 16 ? atoi(argv[16]) : 1, n_groups = argc > 17 ? atoi(argv[17]) : 1;
      if (n_streams < 1 || n_streams > 512 || n_workers < 1 || n_groups < 1 || n_groups > n_streams) throw std::runtime_error("n_streams / n_workers / n_groups out of range");
      std::vector<std::string> roots;
      if (const char* e = getenv("SVOH_MINI_STEREO_ROOTS")) { std::stringstream ss(e); std::string r; while (std::getline(ss, r, ':')) if (!r.empty()) roots.push_back(r); }
      if (roots.empty()) roots.push_back(argv[1]);
      for (int s = 1; s < n_streams; ++s) (void)mkdir((out_dir + "/stream" + std::to_string(s)).c_str(), 0755);
      params.depth_filter.use_threaded_depthfilter = false;
      const std::vector<StereoRoot> data = load_roots(roots, T_imu_world0, max_frames);
      // n_groups lock-step groups side by side (a context and a thread each, n_streams / n_groups streams): one group's host phases meet another's device waits
      std::vector<StereoGroupResult> res((size_t)n_groups);
      std::atomic<int> gate(0);
      std::vector<std::thread> threads;
      auto range = [&](int g, int* s0, int* n) { *s0 = (int)((long long)n_streams * g / n_groups); *n = (int)((long long)n_streams * (g + 1) / n_groups) - *s0; };
      for (int g = 1; g < n_groups; ++g) { int s0, n; range(g, &s0, &n); threads.emplace_back(run_lockstep, std::cref(data), std::cref(rig), std::cref(params), out_dir, kf_every, lambda_rot, s0, n, n_workers, &gate, n_groups, &res[(size_t)g]); }
      { int s0, n; range(0, &s0, &n); run_lockstep(data, rig, params, out_dir, kf_every, lambda_rot, s0, n, n_workers, &gate, n_groups, &res[0]); }
      for (std::thread& t : threads) t.join();
      double rate = 0;
      for (const StereoGroupResult& r : res) { if (!r.error.empty()) throw std::runtime_error(r.error); if (r.ms > 0) rate += 1e3 * (double)r.pairs / r.ms; }
      printf("svoh_mini_stereo lockstep: %d streams in %d group(s), %d host thread(s) per group: %.0f pairs/s in steady state, a round of group 0 %.3f ms, %d device calls per round\n",
             n_streams, n_groups, n_workers, rate, res[0].round_ms, res[0].device_calls);
      return 0;
    }

    // the IMU's relative rotations, if the dataset has them
    // have_prior[k]: a prior was really parsed / integrated for frame k.  The reference applies NO prior to a frame whose
    // ImuHandler::getRelativeRotationPrior fails (frame_handler_base.cpp:622: have_motion_prior_ stays false) -- a frame
    // without one must not be pulled towards zero rotation by an identity stand-in.
    std::vector<svoh::Quat> imu_prior;
    std::vector<bool> have_prior;
    {
      std::ifstream in(std::string(argv[1]) + "/mav0/imu_prior.csv");
      std::string line;
      while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        for (char& c : line) if (c == ',') c = ' ';
        std::istringstream ss(line);
        svoh::Quat q{ 1, 0, 0, 0 };
        const bool parsed = static_cast<bool>(ss >> q.w >> q.x >> q.y >> q.z);   // a short or garbled line: no prior for that frame
        imu_prior.push_back(parsed ? q : svoh::Quat{ 1, 0, 0, 0 });
        have_prior.push_back(parsed);
      }
    }
    // ... or the raw gyroscope of a real EuRoC folder (mav0/imu0/data.csv), integrated between the camera timestamps
    // as ImuHandler::getRelativeRotationPrior does (no bias estimate here: zero); T_newimu_lastimu_prior is the
    // inverse of R_lastimu_newimu (frame_handler_base.cpp:563, 1129)
    if (imu_prior.empty()) {
      const std::vector<io::ImuMeasurement> imu = io::readEurocImu(argv[1]);
      if (!imu.empty()) {
        const double bias[3] = { 0.0, 0.0, 0.0 };
        imu_prior.assign(n_frames, svoh::Quat{ 1, 0, 0, 0 });
        have_prior.assign(n_frames, false);
        size_t n_ok = 0;
        for (size_t k = 1; k < n_frames; ++k) {
          svoh::Quat R_old_new;
          if (io::relativeRotationPrior(imu, (double)seq.cam_ts[k - 1] * 1e-9, (double)seq.cam_ts[k] * 1e-9, bias, 0.0, 0.01, &R_old_new)) {
            imu_prior[k] = svoh::Quat{ R_old_new.w, -R_old_new.x, -R_old_new.y, -R_old_new.z };
            have_prior[k] = true;
            ++n_ok;
          }
        }
        fprintf(stderr, "svoh_mini_stereo: rotation priors from %zu gyroscope measurements (%zu of %zu frame intervals covered)\n", imu.size(), n_ok, n_frames - 1);
      }
    }

    if (svoh_create(0, &ctx) != SVOH_OK) throw std::runtime_error(std::string("svoh_create: ") + svoh_last_error_string(nullptr));
    params.depth_filter.use_threaded_depthfilter = false;
    // euroc_stereo_imu.yaml:30-31: img_align_est_illumination_gain / _offset
    params.img_align.estimate_illumination_gain = true;
    params.img_align.estimate_illumination_offset = true;
    SparseImgAlignHip img_align(ctx, SparseImgAlignHip::getDefaultSolverOptions(), params.img_align);
    ReprojectorOptions ropt;
    ropt.max_n_features_per_frame = (size_t)params.max_fts;
    ropt.cell_size = (size_t)params.grid_size;
    ropt.seed_sigma2_thresh = params.seed_sigma2_thresh;
    ropt.affine_est_offset = params.reprojector_affine_est_offset;
    ropt.affine_est_gain = true;   // the stereo-imu configuration estimates the gain in the matcher as well
    ReprojectorHip reprojector0(ctx, ropt, 0), reprojector1(ctx, ropt, 1);
    ReprojectorHip* reprojectors[2] = { &reprojector0, &reprojector1 };
    PoseOptimizerHip pose_optimizer(ctx);
    DepthFilterHip depth_filter(ctx, params.depth_filter);
    DetectorHip seed_detector(ctx, params.detector, rig[0].cam.width, rig[0].cam.height);
    std::shared_ptr<DetectorHip> tri_detector(new DetectorHip(ctx, params.detector, rig[0].cam.width, rig[0].cam.height));
    StereoTriangulationOptions sto;
    sto.triangulate_n_features = 120;   // svo_factory.cpp:240
    StereoTriangulationHip stereo(ctx, sto, tri_detector);
    unsigned shuffle_state = 12345u;    // reproducible order instead of rand()
    stereo.shuffle_ = [&](std::vector<size_t>& idx, size_t n_corners) {
      auto rnd = [&]() { shuffle_state = shuffle_state * 1664525u + 1013904223u; return shuffle_state >> 8; };
      auto shuf = [&](size_t a, size_t b) { for (size_t i = b; i > a + 1; --i) std::swap(idx[i - 1], idx[a + rnd() % (i - a)]); };
      shuf(0, std::min(n_corners, idx.size())); shuf(std::min(n_corners, idx.size()), idx.size());
    };

    io::TrajectoryWriter traj(out_dir + "/trajectory.txt");
    FILE* fc = fopen((out_dir + "/frontend.csv").c_str(), "w");
    if (!fc) throw std::runtime_error("cannot write into " + out_dir);
    fprintf(fc, "frame,is_kf,n_aligned,n_reprojected,n_after_pose_opt,n_seeds_updated,n_landmarks,alpha,beta,ms_pyramid,ms_align,ms_reproject,ms_pose,ms_seeds,ms_kf,ms_pair\n");

    std::deque<FramePtr> kfs;
    FrameBundle::Ptr last;
    auto num_landmarks = [](const Frame& f) { size_t n = 0; for (size_t i = 0; i < f.num_features_ && i < f.landmark_vec_.size(); ++i) n += f.landmark_vec_[i] != nullptr; return n; };
    auto scene_depth = [](const Frame& f, double& d_med, double& d_min) {   // frame_utils::getSceneDepth on the landmarks
      std::vector<double> d;
      for (size_t i = 0; i < f.num_features_ && i < f.landmark_vec_.size(); ++i)
        if (f.landmark_vec_[i]) { const svoh::Vec3 p = svoh::transform(f.T_f_w_, f.landmark_vec_[i]->pos()); d.push_back(sqrt(p.x * p.x + p.y * p.y + p.z * p.z)); }
      if (d.empty()) return false;
      std::sort(d.begin(), d.end());
      d_med = d[d.size() / 2]; d_min = d.front();
      return true;
    };
    auto add_keyframe = [&](const FramePtr& f) {
      kfs.push_back(f);
      while (kfs.size() > 2 * ropt.max_n_kfs) {
        for (auto& sr : kfs.front()->seed_ref_vec_) sr.keyframe.reset();
        removeObservationsOf(*kfs.front());   // (Map::removeKeyframe)
        kfs.pop_front();
      }
    };
    const bool kf_timing = getenv("SVOH_MINI_KF_TIMING") != nullptr;   // where a keyframe's time goes, to stderr
    const bool landmarks_on = getenv("SVOH_MINI_LANDMARKS") == nullptr || atoi(getenv("SVOH_MINI_LANDMARKS")) != 0;
    int next_point_id = 1 << 20;   // (the triangulation's points count from 0)
    auto make_keyframe = [&](const FrameBundle::Ptr& b, size_t kf_id) {
      // stereo triangulation of new features where the left frame has no feature yet (frame_handler_stereo.cpp:146-155); the frame that
      // is not the keyframe upgrades the seeds it hangs on first, as the reference's branch does (:149-154)
      const double tk0 = now_ms();
      if (landmarks_on) upgradeSeedsToFeatures(ctx, b->at(1 - kf_id), &next_point_id);
      tri_detector->resetGrid();
      tri_detector->fillGridWithKeypoints(b->at(0)->px_vec_, b->at(0)->num_features_);
      stereo.compute(b->at(0), b->at(1));
      const double tk1 = now_ms();
      // the keyframe's seeds become landmarks (upgradeSeedsToFeatures, :162), then new seeds in its free cells (depth_filter_->addKeyframe, :167-173)
      double d_med = 0, d_min = 0;
      const FramePtr& f = b->at(kf_id);
      if (landmarks_on) upgradeSeedsToFeatures(ctx, f, &next_point_id);
      if (scene_depth(*b->at(0), d_med, d_min)) {
        seed_detector.resetGrid();
        seed_detector.fillGridWithKeypoints(f->px_vec_, f->num_features_);
        const size_t n_old = f->num_features_;
        depth_filter_utils::initializeSeeds(f, seed_detector, (size_t)params.max_n_seeds_per_frame, (float)(0.5 * d_min), (float)(1.5 * d_med), (float)d_med);
        for (size_t i = n_old; i < f->num_features_; ++i) { f->seed_ref_vec_[i].keyframe = f; f->seed_ref_vec_[i].seed_id = (int)i; }
      }
      const double tk2 = now_ms();
      add_keyframe(b->at(0));
      add_keyframe(b->at(1));
      if (kf_timing) fprintf(stderr, "[keyframe] stereo triangulation %.3f ms, new seeds %.3f ms, keyframe window %.3f ms\n", tk1 - tk0, tk2 - tk1, now_ms() - tk2);
    };

    double sum_ms = 0;
    size_t n_done = 0;
    // Default flow: the second camera's seed update is sent off without waiting (DepthFilterHip::updateSeedsAsync) and
    // finished before anything reads the seeds again -- a keyframe of the same pair, or the next pair's alignment.  The
    // first camera's update cannot overlap it: both update the same seeds, one after the other (frame_handler_stereo.cpp:
    // 127-129).  SVOH_MINI_SYNC=1 waits in place as the reference does; both flows write the same files.
    const bool sync_flow = getenv("SVOH_MINI_SYNC") != nullptr;
    // a pair's CSV row is written once its seed updates have been finished
    struct Row { bool valid = false, finished = true; size_t n_seed_upd = 0; double ms[7] = { 0, 0, 0, 0, 0, 0, 0 }; size_t n_lm = 0; double alpha = 0, beta = 0; size_t k = 0; int is_kf = 0; size_t n_aligned = 0, n_reproj = 0, n_pose = 0; } row;
    auto finish_seeds = [&]() {
      if (row.valid && !row.finished) {
        const double tf0 = now_ms();
        row.n_seed_upd += depth_filter.finishUpdateSeeds();
        row.ms[4] += now_ms() - tf0;
        row.finished = true;
      }
    };
    auto write_row = [&]() {
      if (!row.valid) return;
      finish_seeds();
      fprintf(fc, "%zu,%d,%zu,%zu,%zu,%zu,%zu,%.6f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f\n", row.k, row.is_kf, row.n_aligned, row.n_reproj, row.n_pose, row.n_seed_upd,
              row.n_lm, row.alpha, row.beta, row.ms[0], row.ms[1], row.ms[2], row.ms[3], row.ms[4], row.ms[5], row.ms[6]);
      row.valid = false;
    };
    for (size_t k = 0; k < n_frames; ++k) {
      const double t0 = now_ms();
      const io::GrayImage img0 = io::readPngGray(seq.cam0_files[k]), img1 = io::readPngGray(seq.cam1_files[k]);
      const double t0b = now_ms();
      FrameBundle::Ptr bundle(new FrameBundle);
      const io::GrayImage* imgs[2] = { &img0, &img1 };
      for (int c = 0; c < 2; ++c) {
        FramePtr frame(new Frame, [ctx](Frame* f) { if (f->pyramid) svoh_release_frame(ctx, f->pyramid); delete f; });
        if (svoh_build_pyramid(ctx, imgs[c]->data.data(), imgs[c]->width, imgs[c]->height, imgs[c]->width, SVOH_MEM_HOST, params.n_pyr_levels_to_build,
                               SVOH_HALFSAMPLE_REFERENCE, nullptr, &frame->pyramid) != SVOH_OK)
          throw std::runtime_error(std::string("svoh_build_pyramid: ") + svoh_last_error_string(ctx));
        frame->cam = rig[(size_t)c].cam;
        frame->set_T_cam_imu(svoh::inverse(rig[(size_t)c].T_B_C));
        frame->id_ = (int)(2 * k + (size_t)c);
        bundle->frames_.push_back(frame);
      }
      // the pair before: its second seed update is needed from here on (alignment points, candidates); the wait is
      // booked on that pair's ms_seeds
      const double t0f = now_ms();
      write_row();
      const double t1 = now_ms();
      size_t n_aligned = 0, n_reproj = 0, n_pose = 0, n_seed_upd = 0;
      double t2 = t1, t3 = t1, t4 = t1, t5 = t1;
      bool is_kf = false, seeds_in_flight = false;
      if (k == 0) {
        for (const FramePtr& f : bundle->frames_) f->T_f_w_ = svoh::mul(f->T_cam_imu(), T_imu_world0);
        make_keyframe(bundle, 0);
        is_kf = true;
        t2 = t3 = t4 = t5 = now_ms();
      } else {
        // 1. sparse image alignment of the bundle, with the IMU's rotation prior (frame_handler_base.cpp:610-643)
        for (size_t c = 0; c < 2; ++c) { bundle->at(c)->T_f_w_ = last->at(c)->T_f_w_; resolveAlignmentPoints(*last->at(c)); }
        img_align.reset();
        if (k < imu_prior.size() && have_prior[k] && lambda_rot > 0) {
          Transformation T_prior{ imu_prior[k], { 0, 0, 0 } };   // T_newimu_lastimu_prior: the rotation is what the weights use
          img_align.setWeightedPrior(T_prior, 0.0, 0.0, lambda_rot, 0.0, 0.0, 0.0);
        }
        n_aligned = img_align.run(last, bundle);
        t2 = now_ms();
        // 2. reprojection, per camera (frame_handler_base.cpp:645-744).  (Queueing the first camera's candidate projection
        // behind the bundle alignment, as the mono harness does, was measured here and does not pay: 0.767 against 0.722 ms
        // per pair -- the hook's walk and three more stream operations cost the alignment stage what the reprojector saves.)
        std::vector<FramePtr> visible(kfs.begin(), kfs.end());
        for (size_t c = 0; c < 2; ++c) {
          std::vector<PointPtr> trash;
          reprojectors[c]->reprojectFrames(bundle->at(c), visible, trash);
          n_reproj += bundle->at(c)->num_features_;
        }
        t3 = now_ms();
        // 3. pose optimisation of the rig
        if (n_reproj >= 10) n_pose = pose_optimizer.run(bundle, 2.0);
        // 3b. structure optimisation of the pair's landmarks (frame_handler_stereo.cpp:114)
        if (landmarks_on) (void)optimizeStructure(ctx, bundle, params.structure_optimization_max_pts, 5);
        t4 = now_ms();
        // 4. depth filter, per camera (frame_handler_stereo.cpp:127-129); at a keyframe BEHIND the keyframe step, as makeKeyframe does
        // (:162-175: upgradeSeedsToFeatures, addKeyframe, then the two updateSeeds over the keyframes that were visible)
        const bool kf_next = k % kf_every == 0 || n_pose < 60;
        if (kf_next) { make_keyframe(bundle, (k / kf_every) % 2); is_kf = true; }
        n_seed_upd += depth_filter.updateSeeds(visible, bundle->at(0));
        if (sync_flow || kf_next) n_seed_upd += depth_filter.updateSeeds(visible, bundle->at(1));
        else { depth_filter.updateSeedsAsync(visible, bundle->at(1)); seeds_in_flight = true; }
        t5 = now_ms();
      }
      last = bundle;   // the pair before this one is dropped here unless it is a keyframe: its release is part of the pair's time
      const double t6 = now_ms();
      traj.write(seq.cam_ts[k], svoh::inverse(bundle->at(0)->T_imu_world()));
      row.valid = true; row.finished = !seeds_in_flight; row.k = k; row.is_kf = (int)is_kf; row.n_aligned = n_aligned; row.n_reproj = n_reproj; row.n_pose = n_pose;
      row.n_seed_upd = n_seed_upd; row.n_lm = num_landmarks(*bundle->at(0)) + num_landmarks(*bundle->at(1));
      row.alpha = img_align.lastResult().alpha; row.beta = img_align.lastResult().beta;
      row.ms[0] = t0f - t0b; row.ms[1] = t2 - t1; row.ms[2] = t3 - t2; row.ms[3] = t4 - t3; row.ms[4] = t5 - t4; row.ms[5] = t6 - t5;
      row.ms[6] = t6 - t0b;   // the pair on the caller's clock (image decoding excluded), the wait for the pair before included
      if (sync_flow) write_row();
      if (k > 0) { sum_ms += t6 - t0b; ++n_done; }
      (void)t0;
    }
    write_row();
    fclose(fc);
    printf("svoh_mini_stereo: %zu frame pairs, %.3f ms per pair on the GPU path (image decoding excluded), %zu keyframes alive\n", n_done + 1,
           n_done ? sum_ms / n_done : 0.0, kfs.size());
    for (const FramePtr& f : kfs) for (auto& sr : f->seed_ref_vec_) sr.keyframe.reset();
    if (last) for (const FramePtr& f : last->frames_) for (auto& sr : f->seed_ref_vec_) sr.keyframe.reset();
    kfs.clear(); last.reset();
    svoh_destroy(ctx);
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "svoh_mini_stereo: %s\n", e.what());
    return 1;
  }
}
