// svoh_mini_frontend -- the per-frame chain of FrameHandlerMono::processFrame (src/svo/src/frame_handler_mono.cpp,
// frame_handler_base.cpp:610-825) assembled from this library's mirrors, on an EuRoC-layout image sequence:
//
//   sparse image alignment (last frame -> new frame)            SparseImgAlignHip::run
//   reprojection of the keyframes' landmarks / seeds            ReprojectorHip::reprojectFrames
//   pose optimisation on the matched features                   PoseOptimizerHip::run
//   depth-filter update of the keyframes' seeds                 DepthFilterHip::updateSeeds
//   at keyframes: feature detection + seed initialisation       DetectorHip + depth_filter_utils::initializeSeeds
//
//   structure optimisation of the frame's landmarks              svo_hip::optimizeStructure (frame_handler_mono.cpp:157)
//   at keyframes: the seeds the frame hangs on become landmarks   svo_hip::upgradeSeedsToFeatures (frame_handler_base.cpp:828-920)
//
// It is an integration harness, NOT the reference's frame handler: there is no map, no initialiser (the first
// pose and a depth prior are given), no relocalisation, and keyframes are chosen by a fixed rule (every <kf_every>
// frames or when fewer than <min_tracked> features survive).  Round 6: landmarks -- a frame selected as keyframe upgrades
// the seeds its features hang on to points, as the frame handler does, so that the reprojector's landmark pass, the
// n_reproj ordering, landmark positions in the alignment and the pose optimiser, and the structure optimisation of every
// frame are part of the chain (SVOH_MINI_LANDMARKS=0: the seed-only chain of rounds 2 - 5, for comparison).
//
//   svoh_mini_frontend <dataset_root> <calib.yaml> <params.yaml|-> <out_dir> <T_f_w of frame 0: qw qx qy qz tx ty tz>
//                      <depth_min> <depth_mean> <depth_max> [max_frames] [kf_every] [n_streams] [threads|lockstep] [n_workers] [n_groups] [n_laps]
// Writes <out>/trajectory.txt (TUM format, T_world_cam) and <out>/frontend.csv (per-frame counters and timings).
// Environment: SVOH_MINI_SYNC=1 runs every stage as a blocking call, in the reference's order (round 2's flow); by
// default (a) the candidate projection of the reprojector (f-4) is queued on the device behind the alignment launch
// and comes back with the alignment's round trip, and (b) the depth filter's seed update is sent off without a wait
// and finished at the start of the next frame, before anything reads the seeds again.  Both flows give the same
// trajectory file byte for byte (tests/test_mini_frontend_gpu.py).
// n_streams > 1 (independent camera streams need no exchange): that many host threads, each
// with its own svoh_ctx (own HIP stream) and its own copy of the chain's state, run the same sequence side by side on
// ONE GPU; stream k > 0 writes into <out>/stream<k>/.  At EuRoC sizes a stream keeps the GPU busy for a small part
// of its frame time (every stage is a latency-bound round trip), so streams share a GPU almost for free until the
// host cores or the launch path saturate: the tool prints the aggregate frame rate.
// `lockstep` (round 5; host/svo_hip_lockstep.h): the n_streams streams take one frame each at a time and every STAGE of
// the chain is ONE launch for all of them (FrontendLockstep), their host work spread over n_workers threads; n_groups > 1
// runs that many lock-step groups side by side (own context, own threads, n_streams / n_groups streams each), so that one
// group's host phases overlap the other's device phases.  Every stream writes the trajectory and the counters of its
// single-stream run, byte for byte (tests/test_mini_frontend_gpu.py).  n_laps > 1 runs the sequence that many times back
// to back in the lock-step mode (the poses restart, the files hold the last lap): a longer steady state for the rate.
// SVOH_MINI_SPEC=<file> (round 6): streams that DIFFER.  One line per stream, key=value pairs:
//   start=<image> step=<+-n> frames=<n>   the stream's j-th frame is image pingpong(start + step * j) of the decoded sequence
//                                         (the walk turns round at both ends), j = 0 .. frames - 1
//   every=<n> phase=<n>                   lock-step only: the stream has a frame in round r iff r >= phase and (r - phase) % every == 0
//   kf_every=<n> min_tracked=<n>          its keyframe rule        params=<yaml>   its own parameter file (max_fts, ...)
//   T0=qw,qx,qy,qz,tx,ty,tz               T_f_w of its first frame
//   calib=<yaml>                          its own camera calibration (intrinsics, distortion, extrinsics; the image size of the run's)
// lockstep: n_streams = the number of lines (the command line's n_streams must agree); threads mode with n_streams = 1:
// SVOH_MINI_SPEC_LINE=<i> runs stream i of the file alone -- the run a lock-step stream must reproduce byte for byte.
// A frame's timestamp in the trajectory file is its image's.
#include <sys/stat.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <deque>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../svo_pro_universal_amd/host/svo_hip_io.h"
#include "../svo_pro_universal_amd/host/svo_hip_lockstep.h"

using namespace svo_hip;

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

namespace {
struct StreamResult { size_t n_done = 0, n_kfs = 0; double sum_ms = 0, wall_ms = 0; std::string error; };

// one line of SVOH_MINI_SPEC
struct StreamSpec {
  long start = 0, step = 1, every = 1, phase = 0, frames = -1, kf_every = -1, min_tracked = -1;
  std::string params, calib;
  bool has_T0 = false;
  Transformation T0{ { 1, 0, 0, 0 }, { 0, 0, 0 } };
  // image of the stream's j-th frame
  size_t image(long j, size_t n_images) const
  {
    if (n_images < 2) return 0;
    const long period = 2 * ((long)n_images - 1);
    long m = (start + step * j) % period;
    if (m < 0) m += period;
    return (size_t)(m < (long)n_images ? m : period - m);
  }
};
std::vector<StreamSpec> load_specs(const char* path)
{
  std::vector<StreamSpec> out;
  FILE* f = fopen(path, "r");
  if (!f) throw std::runtime_error(std::string("cannot read the stream spec ") + path);
  char line[4096];
  while (fgets(line, sizeof line, f)) {
    StreamSpec sp;
    bool any = false;
    for (char* tok = strtok(line, " \t\r\n"); tok; tok = strtok(nullptr, " \t\r\n")) {
      if (tok[0] == '#') break;
      char* eq = strchr(tok, '=');
      if (!eq) { fclose(f); throw std::runtime_error(std::string("stream spec: not key=value: ") + tok); }
      const std::string key(tok, eq), val(eq + 1);
      any = true;
      if (key == "start") sp.start = atol(val.c_str());
      else if (key == "step") sp.step = atol(val.c_str());
      else if (key == "every") sp.every = atol(val.c_str());
      else if (key == "phase") sp.phase = atol(val.c_str());
      else if (key == "frames") sp.frames = atol(val.c_str());
      else if (key == "kf_every") sp.kf_every = atol(val.c_str());
      else if (key == "min_tracked") sp.min_tracked = atol(val.c_str());
      else if (key == "params") sp.params = val;
      else if (key == "calib") sp.calib = val;
      else if (key == "T0") {
        double v[7];
        if (sscanf(val.c_str(), "%lf,%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6) != 7) { fclose(f); throw std::runtime_error("stream spec: T0 needs seven numbers"); }
        sp.T0 = Transformation{ { v[0], v[1], v[2], v[3] }, { v[4], v[5], v[6] } };
        sp.has_T0 = true;
      } else { fclose(f); throw std::runtime_error("stream spec: unknown key " + key); }
    }
    if (!any) continue;
    if (sp.every < 1 || sp.phase < 0 || sp.step == 0) { fclose(f); throw std::runtime_error("stream spec: every >= 1, phase >= 0, step != 0"); }
    out.push_back(sp);
  }
  fclose(f);
  if (out.empty()) throw std::runtime_error("stream spec: no streams");
  return out;
}

// one camera stream through the whole chain; images are decoded beforehand and shared read-only
void run_stream(const io::EurocSequence& seq, const std::vector<io::GrayImage>& images, const std::vector<io::RigCamera>& rig,
                io::FrontendParams params, const std::string& out_dir, const Transformation& T0, float depth_min, float depth_mean,
                float depth_max, size_t kf_every, std::atomic<int>* start_gate, int n_streams, StreamResult* out, const StreamSpec* spec = nullptr)
{
  try {
    size_t min_tracked = getenv("SVOH_MINI_MIN_TRACKED") ? (size_t)atol(getenv("SVOH_MINI_MIN_TRACKED")) : 60;   // (tests raise it to make the rule fire)
    // the images of the stream's frames, in its order (a spec line: its own walk over the decoded sequence)
    std::vector<size_t> order;
    if (spec) {
      const long n = spec->frames > 0 ? spec->frames : (long)images.size();
      for (long j = 0; j < n; ++j) order.push_back(spec->image(j, images.size()));
      if (spec->min_tracked >= 0) min_tracked = (size_t)spec->min_tracked;
    } else {
      for (size_t k = 0; k < images.size(); ++k) order.push_back(k);
    }
    const bool sync_flow = getenv("SVOH_MINI_SYNC") != nullptr && atoi(getenv("SVOH_MINI_SYNC")) != 0;
    // (opt-in: one stream's update is back before its next frame's alignment is set up -- 0.372 / 0.382 ms per frame with it against
    // 0.370 / 0.383 without; several streams in lock step gain 2 - 5 % from the same thing, FrontendLockstep)
    const bool align_ahead_on = getenv("SVOH_MINI_ALIGN_AHEAD") != nullptr && atoi(getenv("SVOH_MINI_ALIGN_AHEAD")) != 0;
    svoh_ctx* ctx = nullptr;
    if (svoh_create(0, &ctx) != SVOH_OK) throw std::runtime_error(std::string("svoh_create: ") + svoh_last_error_string(nullptr));
    // SVOH_MINI_ALIGN_SHARED_CLASSES=1: svoh_set_align_geometry_classes(ctx, 1) -- in both modes of the tool, so that a stream alone and in lock step agree
    if (getenv("SVOH_MINI_ALIGN_SHARED_CLASSES") && svoh_set_align_geometry_classes(ctx, atoi(getenv("SVOH_MINI_ALIGN_SHARED_CLASSES")) != 0) != SVOH_OK) throw std::runtime_error(svoh_last_error_string(ctx));
    const svoh_camera& cam = rig.at(0).cam;

    params.depth_filter.use_threaded_depthfilter = false;   // the synchronous path
    SparseImgAlignHip img_align(ctx, SparseImgAlignHip::getDefaultSolverOptions(), params.img_align);
    ReprojectorOptions ropt;
    ropt.max_n_features_per_frame = (size_t)params.max_fts;
    ropt.cell_size = (size_t)params.grid_size;
    ropt.seed_sigma2_thresh = params.seed_sigma2_thresh;
    ropt.affine_est_offset = params.reprojector_affine_est_offset;
    ropt.affine_est_gain = params.reprojector_affine_est_gain;
    ReprojectorHip reprojector(ctx, ropt, 0);
    PoseOptimizerHip pose_optimizer(ctx);
    DepthFilterHip depth_filter(ctx, params.depth_filter);
    DetectorHip detector(ctx, params.detector, cam.width, cam.height);

    io::TrajectoryWriter traj(out_dir + "/trajectory.txt");
    FILE* fc = fopen((out_dir + "/frontend.csv").c_str(), "w");
    if (!fc) throw std::runtime_error("cannot write into " + out_dir);
    fprintf(fc, "frame,is_kf,n_aligned,n_reprojected,n_after_pose_opt,n_seeds_updated,n_converged_seeds,ms_pyramid,ms_align,ms_reproject,ms_pose,ms_seeds,ms_kf,ms_frame,n_points_optimized,n_landmarks\n");

    std::deque<FramePtr> kfs;   // the last reprojector.max_n_kfs keyframes
    FramePtr last;
    const bool landmarks_on = getenv("SVOH_MINI_LANDMARKS") == nullptr || atoi(getenv("SVOH_MINI_LANDMARKS")) != 0;
    int next_point_id = 0;
    auto make_keyframe = [&](const FramePtr& f) {
      // the frame handler's step at a new keyframe (frame_handler_mono.cpp:186): the seeds this frame's features hang on become
      // landmarks (the seed update in flight, if any, leaves such a seed alone when it is written back: DepthFilterHip)
      if (landmarks_on) upgradeSeedsToFeatures(ctx, f, &next_point_id);
      detector.resetGrid();
      detector.fillGridWithKeypoints(f->px_vec_, f->num_features_);
      const size_t n_old = f->num_features_;
      depth_filter_utils::initializeSeeds(f, detector, (size_t)params.max_n_seeds_per_frame, depth_min, depth_max, depth_mean);
      // bootstrap stand-in for the initialiser's landmarks: a keyframe's own new seeds are usable for the
      // alignment of the next frame at their current depth estimate (self reference)
      for (size_t i = n_old; i < f->num_features_; ++i) { f->seed_ref_vec_[i].keyframe = f; f->seed_ref_vec_[i].seed_id = (int)i; }
      kfs.push_back(f);
      while (kfs.size() > ropt.max_n_kfs) {
        for (auto& sr : kfs.front()->seed_ref_vec_) sr.keyframe.reset();   // break the self references
        removeObservationsOf(*kfs.front());                                 // (Map::removeKeyframe)
        kfs.pop_front();
      }
    };
    // all streams start their first frame together
    start_gate->fetch_add(1);
    while (start_gate->load() < n_streams) std::this_thread::yield();
    // SVOH_MINI_PREPARE=1: the seed update is staged and uploaded while the pose kernel runs (PoseOptimizerHip::run's hook +
    // DepthFilterHip::prepareUpdateSeeds).  Off by default: worth 5 - 8 us to ONE stream, but eight streams on one GPU lose a
    // quarter of their total frame rate with it (measured: ~4 600 against ~6 300 frames/s; the hook's wait is an event wait).
    const bool no_prepare = getenv("SVOH_MINI_PREPARE") == nullptr;
    const double wall0 = now_ms();
    double sum_ms = 0;
    size_t n_done = 0;
    // a frame's CSV row is written once its seed update has been finished (at the start of the next frame in the
    // default flow): the counters of both flows are the same
    struct Row { bool valid = false, finished = false; size_t k = 0, n_seed_upd = 0; int is_kf = 0; size_t n_aligned = 0, n_reproj = 0, n_pose = 0, n_struct = 0, n_landmarks = 0; double ms[7] = {0, 0, 0, 0, 0, 0, 0}; } row;
    auto finish_row = [&]() {
      if (!row.valid) return;
      if (!row.finished) {
        const double tf0 = now_ms();
        row.n_seed_upd = depth_filter.finishUpdateSeeds();
        row.ms[4] += now_ms() - tf0;
      }
      const size_t n_seed_upd = row.n_seed_upd;
      size_t n_conv = 0;
      for (const FramePtr& f : kfs)
        for (size_t i = 0; i < f->num_features_; ++i)
          n_conv += f->type_vec_[i] == SVOH_FT_CORNER_SEED_CONVERGED || f->type_vec_[i] == SVOH_FT_EDGELET_SEED_CONVERGED;
      fprintf(fc, "%zu,%d,%zu,%zu,%zu,%zu,%zu,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%zu,%zu\n", row.k, row.is_kf, row.n_aligned, row.n_reproj, row.n_pose,
              n_seed_upd, n_conv, row.ms[0], row.ms[1], row.ms[2], row.ms[3], row.ms[4], row.ms[5], row.ms[6], row.n_struct, row.n_landmarks);
      row.valid = false;
    };
    StructureBatch structure;
    bool structure_in_flight = false;
    auto finish_structure = [&]() {
      if (!structure_in_flight) return;
      structure_in_flight = false;
      if (svoh_optimize_points_batch_collect(ctx, (int)structure.size(), structure.pos.data(), nullptr) != SVOH_OK)
        throw std::runtime_error(std::string("svoh_optimize_points_batch_collect: ") + svoh_last_error_string(ctx));
      structure.apply(structure.pos.data());
    };
    for (size_t k = 0; k < order.size(); ++k) {
      const io::GrayImage& img = images[order[k]];
      const double t0 = now_ms();
      FramePtr frame(new Frame, [ctx](Frame* f) { if (f->pyramid) svoh_release_frame(ctx, f->pyramid); delete f; });
      if (svoh_build_pyramid(ctx, img.data.data(), img.width, img.height, img.width, SVOH_MEM_HOST, params.n_pyr_levels_to_build,
                             SVOH_HALFSAMPLE_REFERENCE, nullptr, &frame->pyramid) != SVOH_OK)
        throw std::runtime_error(std::string("svoh_build_pyramid: ") + svoh_last_error_string(ctx));
      frame->cam = cam;
      frame->set_T_cam_imu(svoh::inverse(rig[0].T_B_C));
      frame->id_ = (int)k;
      // the previous frame's seed update: its results are needed from here on (alignment points, candidates); the wait
      // and the write-back are booked on that frame's ms_seeds (finish_row), not on this frame's ms_pyramid
      // (SVOH_MINI_ALIGN_AHEAD=1: the alignment is queued AHEAD of this wait -- its seed points take their
      // position on the device from the update's batch, svoh_align_camera::pos_seed_unit -- and the wait runs in its hook)
      const double t0f = now_ms();
      const bool align_ahead = !sync_flow && align_ahead_on && k > 0 && depth_filter.updateInFlight();
      finish_structure();   // the points optimised behind the frame before: their positions are read from here on
      if (!align_ahead) finish_row();
      const double t1 = now_ms();
      size_t n_aligned = 0, n_reproj = 0, n_pose = 0, n_seed_upd = 0, n_struct = 0;
      bool seeds_finished = false;
      double t2 = t1, t3 = t1, t4 = t1, t5 = t1;
      bool is_kf = false;
      if (k == 0) {
        frame->T_f_w_ = T0;
        make_keyframe(frame);
        is_kf = true;
        t2 = t3 = t4 = t5 = now_ms();
      } else {
        // 1. sparse image alignment against the last frame (frame_handler_base.cpp:610-643)
        frame->T_f_w_ = last->T_f_w_;
        if (align_ahead) resolveAlignmentPoints(*last, [&](const Frame& kf, size_t seed_id) { return depth_filter.unitOfPendingSeed(kf, seed_id); });
        else resolveAlignmentPoints(*last);
        FrameBundle::Ptr b_last(new FrameBundle), b_cur(new FrameBundle);
        b_last->frames_.push_back(last); b_cur->frames_.push_back(frame);
        img_align.reset();
        std::vector<FramePtr> visible(kfs.begin(), kfs.end());
        if (sync_flow) {
          n_aligned = img_align.run(b_last, b_cur);
        } else {
          // the candidates' projection is queued behind the alignment kernel; its pose is the alignment's result,
          // composed on the device (ReprojectorHip::enqueueCandidateProjection)
          n_aligned = img_align.run(b_last, b_cur, [&](const Transformation& T_iref_world) {
            if (align_ahead) finish_row();   // the previous frame's seed update: waited for and written back beside the alignment kernel
            reprojector.enqueueCandidateProjection(frame, visible, &T_iref_world, 0);
          });
          if (img_align.lastRunRepeated() || n_aligned == 0) reprojector.discardCandidateProjection();
        }
        t2 = now_ms();
        // 2. reprojection (frame_handler_base.cpp:645-744)
        std::vector<PointPtr> trash;
        reprojector.reprojectFrames(frame, visible, trash);
        n_reproj = frame->num_features_;
        t3 = now_ms();
        // 3. pose optimisation (frame_handler_base.cpp:746-790)
        // (default flow: while the pose kernel runs, the depth filter's update of this frame is staged and uploaded --
        // it needs the optimised pose only as the last word before its kernel goes off)
        if (frame->num_features_ >= 10) {
          if (sync_flow || no_prepare) n_pose = pose_optimizer.run(b_cur, 2.0);
          else n_pose = pose_optimizer.run(b_cur, 2.0, [&]() { depth_filter.prepareUpdateSeeds(visible, frame); });
        }
        // 3b. structure optimisation of the frame's landmarks (frame_handler_mono.cpp:157: optimizeStructure(new_frames_, max_pts, 5)).
        // Pipelined flow: queued here and collected at the next frame's start, before anything reads a point's position again (the
        // keyframe step in between makes points and adds observations; it moves none) -- the same bytes as the blocking call.
        if (landmarks_on && sync_flow) n_struct = optimizeStructure(ctx, b_cur, params.structure_optimization_max_pts, 5);
        else if (landmarks_on && params.structure_optimization_max_pts != 0) {
          structure.gather(*frame, params.structure_optimization_max_pts);
          n_struct = structure.size();
          if (n_struct) {
            if (svoh_optimize_points_batch_enqueue(ctx, 5, 0, (int)structure.views.size(), structure.views.data(), (int)structure.size(), structure.obs_begin.data(),
                                                   structure.obs_view.data(), structure.obs_f.data(), structure.pos.data()) != SVOH_OK)
              throw std::runtime_error(std::string("svoh_optimize_points_batch_enqueue: ") + svoh_last_error_string(ctx));
            structure_in_flight = true;
          }
        }
        t4 = now_ms();
        // 4. depth filter (frame_handler_mono.cpp:125: the keyframes that were visible from this frame are updated with it -- in the
        // reference at the start of the NEXT frame, i.e. after this frame may have become a keyframe and upgraded some of their seeds
        // to landmarks: a seed that is a feature by then is not updated.  The pipelined flow sends the update off here and drops, when it
        // writes the results back, those of seeds that were upgraded meanwhile; the blocking flow runs it after the keyframe step.)
        if (!sync_flow) depth_filter.updateSeedsAsync(visible, frame);
        t5 = now_ms();
        // 5. keyframe rule
        if (k % kf_every == 0 || frame->numTrackedFeatures() < min_tracked) { make_keyframe(frame); is_kf = true; }
        if (sync_flow) { depth_filter.updateSeedsAsync(visible, frame); n_seed_upd = depth_filter.finishUpdateSeeds(); seeds_finished = true; }   // wait here, as the reference does
      }
      // the frame before this one is dropped here unless it is a keyframe: its release (svoh_release_frame) is part of
      // the frame's time
      last = frame;
      const double t6 = now_ms();
      traj.write(seq.cam_ts[order[k]], svoh::inverse(frame->T_f_w_));
      row.valid = true; row.finished = seeds_finished || k == 0; row.n_seed_upd = n_seed_upd; row.k = k; row.is_kf = (int)is_kf; row.n_aligned = n_aligned; row.n_reproj = n_reproj; row.n_pose = n_pose;
      row.n_struct = n_struct; row.n_landmarks = 0;
      for (size_t i = 0; i < frame->num_features_ && i < frame->landmark_vec_.size(); ++i) row.n_landmarks += frame->landmark_vec_[i] != nullptr;
      row.ms[0] = t0f - t0; row.ms[1] = t2 - t1; row.ms[2] = t3 - t2; row.ms[3] = t4 - t3; row.ms[4] = t5 - t4; row.ms[5] = t6 - t5;
      row.ms[6] = t6 - t0;   // the frame as the caller's clock sees it, the previous frame's seed write-back included
      if (sync_flow) finish_row();
      if (k > 0) { sum_ms += t6 - t0; ++n_done; }
    }
    finish_structure();
    finish_row();
    out->wall_ms = now_ms() - wall0;
    fclose(fc);
    out->n_done = n_done; out->sum_ms = sum_ms; out->n_kfs = kfs.size();
    for (const FramePtr& f : kfs) for (auto& sr : f->seed_ref_vec_) sr.keyframe.reset();
    if (last) for (auto& sr : last->seed_ref_vec_) sr.keyframe.reset();
    kfs.clear(); last.reset();
    svoh_destroy(ctx);
  } catch (const std::exception& e) {
    out->error = e.what();
    start_gate->fetch_add(1);   // never leave the other streams waiting at the gate
  }
}

// one lock-step group: streams [s0, s0 + n) of the run, all fed the same decoded sequence
struct GroupResult { size_t frames = 0, steady_frames = 0; double wall_ms = 0, steady_ms = 0; size_t steady_rounds = 0; int device_calls = 0; FrontendLockstep::RoundTimes mean{}; std::string error; };

void run_lockstep_group(const io::EurocSequence& seq, const std::vector<io::GrayImage>& images, const std::vector<io::RigCamera>& rig, const io::FrontendParams& params,
                        const std::string& out_dir, const Transformation& T0, float depth_min, float depth_mean, float depth_max, size_t kf_every, int s0, int n,
                        int n_workers, int n_laps, std::atomic<int>* start_gate, int n_groups, GroupResult* out, const std::vector<StreamSpec>* specs = nullptr)
{
  try {
    svoh_ctx* ctx = nullptr;
    if (svoh_create(0, &ctx) != SVOH_OK) throw std::runtime_error(std::string("svoh_create: ") + svoh_last_error_string(nullptr));
    if (getenv("SVOH_MINI_ALIGN_SHARED_CLASSES") && svoh_set_align_geometry_classes(ctx, atoi(getenv("SVOH_MINI_ALIGN_SHARED_CLASSES")) != 0) != SVOH_OK) throw std::runtime_error(svoh_last_error_string(ctx));
    // the decoded sequence in page-locked memory, as a camera driver that feeds a GPU would deliver its images: EVERY STREAM
    // ITS OWN COPY (the device reads the images in place, and 32 streams reading one buffer would be served from its caches
    // after the first: every stream's image has to cross PCIe by itself, as the images of 32 different cameras do)
    const size_t img_bytes = (size_t)images[0].width * images[0].height;
    const size_t seq_bytes = img_bytes * images.size();
    uint8_t* pinned = nullptr;
    if (svoh_host_alloc(ctx, seq_bytes * (size_t)n, (void**)&pinned) != SVOH_OK) throw std::runtime_error(std::string("svoh_host_alloc: ") + svoh_last_error_string(ctx));
    for (size_t k = 0; k < images.size(); ++k) {
      if ((size_t)images[k].width * images[k].height != img_bytes) throw std::runtime_error("images of different sizes");
      for (int i = 0; i < n; ++i) memcpy(pinned + (size_t)i * seq_bytes + k * img_bytes, images[k].data.data(), img_bytes);
    }
    for (int lap = 0; lap < n_laps; ++lap) {
      LockstepOptions lo;
      lo.params = params; lo.cam = rig.at(0).cam; lo.T_B_C = rig[0].T_B_C;
      lo.depth_min = depth_min; lo.depth_mean = depth_mean; lo.depth_max = depth_max; lo.kf_every = kf_every; lo.n_workers = n_workers;
      if (getenv("SVOH_MINI_MIN_TRACKED")) lo.min_tracked = (size_t)atol(getenv("SVOH_MINI_MIN_TRACKED"));
      // streams that differ (SVOH_MINI_SPEC): every stream its own parameter file, keyframe rule, first pose and walk over the images
      std::vector<Transformation> T_first((size_t)n, T0);
      size_t n_rounds = images.size();
      if (specs) {
        n_rounds = 0;
        for (int i = 0; i < n; ++i) {
          const StreamSpec& sp = (*specs)[(size_t)(s0 + i)];
          LockstepStreamOptions so;
          so.params = sp.params.empty() ? params : io::loadFrontendParams(sp.params);
          so.depth_min = depth_min; so.depth_mean = depth_mean; so.depth_max = depth_max;
          so.kf_every = sp.kf_every > 0 ? (size_t)sp.kf_every : kf_every;
          so.min_tracked = sp.min_tracked >= 0 ? (size_t)sp.min_tracked : lo.min_tracked;
          if (!sp.calib.empty()) {
            const std::vector<io::RigCamera> own_rig = io::loadCameraRig(sp.calib);
            so.own_camera = true; so.cam = own_rig.at(0).cam; so.T_B_C = own_rig[0].T_B_C;
          }
          lo.per_stream.push_back(so);
          if (sp.has_T0) T_first[(size_t)i] = sp.T0;
          const long nf = sp.frames > 0 ? sp.frames : (long)images.size();
          n_rounds = std::max(n_rounds, (size_t)(sp.phase + sp.every * (nf - 1) + 1));
        }
      }
      // the image (or none) of stream i in round r, and the number of that frame in the stream's own count
      auto frame_of = [&](int i, size_t r, long* j) -> long {
        if (!specs) { *j = (long)r; return (long)r; }
        const StreamSpec& sp = (*specs)[(size_t)(s0 + i)];
        const long nf = sp.frames > 0 ? sp.frames : (long)images.size();
        if ((long)r < sp.phase || ((long)r - sp.phase) % sp.every != 0) return -1;
        *j = ((long)r - sp.phase) / sp.every;
        return *j < nf ? (long)sp.image(*j, images.size()) : -1;
      };
      lo.images_mem_space = SVOH_MEM_HOST_PINNED;
      // the two switches the tests use (the rest of round 5's A/B switches is gone: HISTORY round 5 has their numbers)
      if (const char* sp = getenv("SVOH_LOCKSTEP_SPECULATE")) lo.speculation = std::string(sp) == "all" ? LockstepOptions::kSpeculateAll : std::string(sp) == "never" ? LockstepOptions::kSpeculateNever : LockstepOptions::kSpeculateAsBefore;
      if (getenv("SVOH_LOCKSTEP_RESIDENT")) lo.resident_features = atoi(getenv("SVOH_LOCKSTEP_RESIDENT")) != 0;
      if (getenv("SVOH_MINI_LANDMARKS")) lo.landmarks = atoi(getenv("SVOH_MINI_LANDMARKS")) != 0;
      FrontendLockstep fe(ctx, n, lo);
      const bool last_lap = lap + 1 == n_laps;
      std::vector<std::unique_ptr<io::TrajectoryWriter>> traj;
      std::vector<FILE*> csv;
      if (last_lap)
        for (int i = 0; i < n; ++i) {
          const int s = s0 + i;
          const std::string dir = s == 0 ? out_dir : out_dir + "/stream" + std::to_string(s);
          traj.emplace_back(new io::TrajectoryWriter(dir + "/trajectory.txt"));
          FILE* fc = fopen((dir + "/frontend.csv").c_str(), "w");
          if (!fc) throw std::runtime_error("cannot write into " + dir);
          fprintf(fc, "frame,is_kf,n_aligned,n_reprojected,n_after_pose_opt,n_seeds_updated,n_converged_seeds,ms_pyramid,ms_align,ms_reproject,ms_pose,ms_seeds,ms_kf,ms_frame,n_points_optimized,n_landmarks\n");
          csv.push_back(fc);
        }
      std::vector<FrontendLockstep::RoundTimes> times;
      std::vector<std::vector<size_t>> round_of((size_t)n);   // the round in which a stream's j-th frame ran
      auto write_rows = [&]() {
        for (int i = 0; i < n; ++i)
          for (const FrontendLockstep::FrameRow& r : fe.completedRows(i)) {
            if (!last_lap) continue;
            const FrontendLockstep::RoundTimes& t = times.at(round_of[(size_t)i].at(r.k));
            fprintf(csv[(size_t)i], "%zu,%d,%zu,%zu,%zu,%zu,%zu,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%.4f,%zu,%zu\n", r.k, (int)r.is_kf, r.n_aligned, r.n_reproj, r.n_pose, r.n_seed_upd,
                    r.n_converged, t.pyramid, t.align, t.reproject, t.pose, t.seeds, t.keyframe, t.total, r.n_struct, r.n_landmarks);
          }
      };
      std::vector<const uint8_t*> ptrs((size_t)n), next((size_t)n);
      const bool prefetch = true;
      if (lap == 0) {   // all groups start their first frame together
        start_gate->fetch_add(1);
        while (start_gate->load() < n_groups) std::this_thread::yield();
      }
      const double wall0 = now_ms();
      for (size_t k = 0; k < n_rounds; ++k) {
        std::vector<long> img_now((size_t)n, -1);
        size_t n_now = 0;
        for (int i = 0; i < n; ++i) {
          long j = 0;
          const long f = frame_of(i, k, &j);
          img_now[(size_t)i] = f;
          ptrs[(size_t)i] = f >= 0 ? pinned + (size_t)i * seq_bytes + (size_t)f * img_bytes : nullptr;
          if (f >= 0) { round_of[(size_t)i].push_back(k); ++n_now; }
        }
        // the next images of a replay are there already: they go up while this round runs (SVOH_LOCKSTEP_PREFETCH=0: they do not)
        const bool announce = prefetch && k + 1 < n_rounds;
        if (announce)
          for (int i = 0; i < n; ++i) {
            long j = 0;
            const long f = frame_of(i, k + 1, &j);
            next[(size_t)i] = f >= 0 ? pinned + (size_t)i * seq_bytes + (size_t)f * img_bytes : nullptr;
          }
        const double tr0 = now_ms();
        fe.addImages(ptrs.data(), images[0].width, T_first.data(), announce ? next.data() : nullptr);
        const double tr1 = now_ms();
        FrontendLockstep::RoundTimes t = fe.lastRoundTimes();
        t.total = tr1 - tr0;
        times.push_back(t);
        if (last_lap) for (int i = 0; i < n; ++i) if (img_now[(size_t)i] >= 0) traj[(size_t)i]->write(seq.cam_ts[(size_t)img_now[(size_t)i]], svoh::inverse(fe.pose(i)));
        write_rows();
        if (k >= 3) {   // frames 1-2 pay the one-time costs (code objects, the scratch buffers' first allocation)
          out->steady_ms += tr1 - tr0; ++out->steady_rounds;
          out->mean.pyramid += t.pyramid; out->mean.align += t.align; out->mean.reproject += t.reproject; out->mean.pose += t.pose; out->mean.seeds += t.seeds; out->mean.keyframe += t.keyframe;
        }
        out->device_calls = fe.lastRoundDeviceCalls();
        out->frames += n_now;
        if (k >= 3) out->steady_frames += n_now;
      }
      fe.finish();
      write_rows();
      out->wall_ms += now_ms() - wall0;
      if (last_lap && s0 == 0 && getenv("SVOH_LOCKSTEP_TIMING")) {
        const double* ph = fe.phaseTimes();
        fprintf(stderr, "[lockstep] mean ms per round over %zu rounds:", n_rounds);
        for (int k = 0; k < FrontendLockstep::kNumPhases; ++k) if (ph[k] > 0) fprintf(stderr, " %s %.3f,", FrontendLockstep::phaseName(k), ph[k] / (double)n_rounds);
        fprintf(stderr, "\n");
      }
      for (FILE* f : csv) fclose(f);
    }
    (void)svoh_host_free(ctx, pinned);
    svoh_destroy(ctx);
  } catch (const std::exception& e) {
    out->error = e.what();
    start_gate->fetch_add(1);
  }
}
}  // namespace

int main(int argc, char** argv)
{
  if (argc < 15) {
    fprintf(stderr, "usage: %s <dataset_root> <calib.yaml> <params.yaml|-> <out_dir> qw qx qy qz tx ty tz depth_min depth_mean depth_max [max_frames] [kf_every] [n_streams]\n", argv[0]);
    return 2;
  }
  try {
    const io::EurocSequence seq = io::openEuroc(argv[1]);
    const std::vector<io::RigCamera> rig = io::loadCameraRig(argv[2]);
    const io::FrontendParams params = std::string(argv[3]) == "-" ? io::frontendParamsFromYaml(io::YamlNode()) : io::loadFrontendParams(argv[3]);
    const std::string out_dir = argv[4];
    const Transformation T0{ { atof(argv[5]), atof(argv[6]), atof(argv[7]), atof(argv[8]) }, { atof(argv[9]), atof(argv[10]), atof(argv[11]) } };
    const float depth_min = (float)atof(argv[12]), depth_mean = (float)atof(argv[13]), depth_max = (float)atof(argv[14]);
    const size_t max_frames = argc > 15 ? (size_t)atol(argv[15]) : seq.size();
    const size_t kf_every = argc > 16 ? (size_t)atol(argv[16]) : 8;
    int n_streams = argc > 17 ? atoi(argv[17]) : 1;
    std::vector<StreamSpec> specs;
    if (getenv("SVOH_MINI_SPEC")) specs = load_specs(getenv("SVOH_MINI_SPEC"));
    const bool lockstep = argc > 18 && std::string(argv[18]) == "lockstep";
    if (argc > 18 && !lockstep && std::string(argv[18]) != "threads") throw std::runtime_error("mode must be threads or lockstep");
    if (n_streams < 1 || n_streams > (lockstep ? 256 : 64)) throw std::runtime_error("n_streams out of range");
    std::vector<io::GrayImage> images;
    for (size_t k = 0; k < seq.size() && k < max_frames; ++k) images.push_back(io::readPngGray(seq.cam0_files[k]));
    if (lockstep) {
      if (!specs.empty() && (size_t)n_streams != specs.size()) throw std::runtime_error("n_streams must be the number of lines of SVOH_MINI_SPEC");
      const int n_workers = argc > 19 ? atoi(argv[19]) : 1;
      const int n_groups = argc > 20 ? atoi(argv[20]) : 1;
      const int n_laps = argc > 21 ? atoi(argv[21]) : 1;
      if (n_workers < 1 || n_workers > 256 || n_groups < 1 || n_groups > n_streams || n_laps < 1) throw std::runtime_error("n_workers / n_groups / n_laps out of range");
      if (images.empty()) throw std::runtime_error("no images");
      for (int s = 1; s < n_streams; ++s) (void)mkdir((out_dir + "/stream" + std::to_string(s)).c_str(), 0755);
      std::vector<GroupResult> res((size_t)n_groups);
      std::atomic<int> gate(0);
      std::vector<std::thread> threads;
      auto group_range = [&](int g, int* s0, int* n) { *s0 = (int)((long long)n_streams * g / n_groups); *n = (int)((long long)n_streams * (g + 1) / n_groups) - *s0; };
      for (int g = 1; g < n_groups; ++g) {
        int s0, n; group_range(g, &s0, &n);
        threads.emplace_back(run_lockstep_group, std::cref(seq), std::cref(images), std::cref(rig), std::cref(params), out_dir, std::cref(T0), depth_min, depth_mean, depth_max,
                             kf_every, s0, n, n_workers, n_laps, &gate, n_groups, &res[(size_t)g], specs.empty() ? nullptr : &specs);
      }
      { int s0, n; group_range(0, &s0, &n); run_lockstep_group(seq, images, rig, params, out_dir, T0, depth_min, depth_mean, depth_max, kf_every, s0, n, n_workers, n_laps, &gate, n_groups, &res[0], specs.empty() ? nullptr : &specs); }
      for (std::thread& t : threads) t.join();
      for (const GroupResult& r : res) if (!r.error.empty()) throw std::runtime_error(r.error);
      double wall = 0, steady_rate = 0;
      size_t frames = 0;
      for (int g = 0; g < n_groups; ++g) {
        const GroupResult& r = res[(size_t)g];
        int s0, n; group_range(g, &s0, &n);
        wall = r.wall_ms > wall ? r.wall_ms : wall; frames += r.frames;
        if (r.steady_rounds) steady_rate += 1e3 * (double)r.steady_frames / r.steady_ms;
      }
      const GroupResult& r0 = res[0];
      const double nr = r0.steady_rounds ? (double)r0.steady_rounds : 1.0;
      printf("svoh_mini_frontend lockstep: %d streams in %d group(s), %d host thread(s) per group: %zu frames in %.1f ms = %.0f frames/s overall, %.0f frames/s in steady state; "
             "a round of group 0: %.3f ms (pyramid %.3f align %.3f reproject %.3f pose %.3f seeds %.3f keyframe %.3f), %d device calls per round\n",
             n_streams, n_groups, n_workers, frames, wall, 1e3 * frames / wall, steady_rate, r0.steady_ms / nr, r0.mean.pyramid / nr, r0.mean.align / nr, r0.mean.reproject / nr,
             r0.mean.pose / nr, r0.mean.seeds / nr, r0.mean.keyframe / nr, r0.device_calls);
      return 0;
    }
    std::vector<StreamResult> results((size_t)n_streams);
    std::atomic<int> gate(0);
    std::vector<std::thread> threads;
    if (!specs.empty()) {   // ONE stream of the spec file alone, through the single-stream chain
      if (n_streams != 1 || !getenv("SVOH_MINI_SPEC_LINE")) throw std::runtime_error("threads mode with SVOH_MINI_SPEC: n_streams = 1 and SVOH_MINI_SPEC_LINE=<i>");
      const StreamSpec& sp = specs.at((size_t)atol(getenv("SVOH_MINI_SPEC_LINE")));
      const io::FrontendParams own = sp.params.empty() ? params : io::loadFrontendParams(sp.params);
      // the process-wide thresholds are those of the RUN's camera, as in the lock-step engine this stream is compared with (fixProcessWideThresholds)
      fixProcessWideThresholds(rig.at(0).cam, 2.0);
      const std::vector<io::RigCamera> own_rig = sp.calib.empty() ? rig : io::loadCameraRig(sp.calib);
      run_stream(seq, images, own_rig, own, out_dir, sp.has_T0 ? sp.T0 : T0, depth_min, depth_mean, depth_max, sp.kf_every > 0 ? (size_t)sp.kf_every : kf_every, &gate, 1, &results[0], &sp);
      if (!results[0].error.empty()) throw std::runtime_error(results[0].error);
      printf("svoh_mini_frontend: stream %s of the spec alone: %zu frames, %.3f ms/frame on the GPU path, %zu keyframes alive\n", getenv("SVOH_MINI_SPEC_LINE"), results[0].n_done + 1,
             results[0].n_done ? results[0].sum_ms / results[0].n_done : 0.0, results[0].n_kfs);
      return 0;
    }
    for (int s = 1; s < n_streams; ++s) {
      const std::string dir = out_dir + "/stream" + std::to_string(s);
      (void)mkdir(dir.c_str(), 0755);
      threads.emplace_back(run_stream, std::cref(seq), std::cref(images), std::cref(rig), params, dir, std::cref(T0), depth_min, depth_mean,
                           depth_max, kf_every, &gate, n_streams, &results[(size_t)s], (const StreamSpec*)nullptr);
    }
    run_stream(seq, images, rig, params, out_dir, T0, depth_min, depth_mean, depth_max, kf_every, &gate, n_streams, &results[0]);
    for (std::thread& t : threads) t.join();
    for (const StreamResult& r : results) if (!r.error.empty()) throw std::runtime_error(r.error);
    printf("svoh_mini_frontend: %zu frames, %.3f ms/frame on the GPU path, %zu keyframes alive\n", results[0].n_done + 1,
           results[0].n_done ? results[0].sum_ms / results[0].n_done : 0.0, results[0].n_kfs);
    if (n_streams > 1) {
      double wall = 0, per_frame = 0;
      size_t frames = 0;
      for (const StreamResult& r : results) { wall = r.wall_ms > wall ? r.wall_ms : wall; frames += r.n_done + 1; per_frame += r.n_done ? r.sum_ms / r.n_done : 0.0; }
      printf("svoh_mini_frontend: %d streams on one GPU: %zu frames in %.1f ms = %.0f frames/s in total, %.3f ms/frame per stream (mean)\n",
             n_streams, frames, wall, 1e3 * frames / wall, per_frame / n_streams);
    }
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "svoh_mini_frontend: %s\n", e.what());
    return 1;
  }
}
