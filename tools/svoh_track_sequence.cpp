// svoh_track_sequence -- runs the GPU feature-tracking front end over an EuRoC-layout image sequence:
// PNG -> 5-level pyramid -> FeatureTracker::trackAndDetect (batched KLT + FAST/edgelet detection when tracks run
// out), the loop of the reference's examples (examples/euroc_mono.cpp:30-58: read frame, hand it to the front end)
// for the stages this library implements.  Writes <out>/tracks.csv (frame, timestamp_ns, track_id, x, y) and
// <out>/timing.csv (frame, n_tracked, n_features, ms_read, ms_pyramid, ms_track_detect).
//
//   svoh_track_sequence <dataset_root> <calib.yaml> <params.yaml|-> <out_dir> [max_frames]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "../svo_pro_universal_amd/host/svo_hip_io.h"

using namespace svo_hip;

static double now_ms()
{
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
  if (argc < 5) {
    fprintf(stderr, "usage: %s <dataset_root> <calib.yaml> <params.yaml|-> <out_dir> [max_frames]\n", argv[0]);
    return 2;
  }
  try {
    const io::EurocSequence seq = io::openEuroc(argv[1]);
    const std::vector<io::RigCamera> rig = io::loadCameraRig(argv[2]);
    const io::FrontendParams params = std::string(argv[3]) == "-" ? io::frontendParamsFromYaml(io::YamlNode()) : io::loadFrontendParams(argv[3]);
    const std::string out_dir = argv[4];
    const size_t max_frames = argc > 5 ? (size_t)atol(argv[5]) : seq.size();
    svoh_ctx* ctx = nullptr;
    if (svoh_create(0, &ctx) != SVOH_OK) throw std::runtime_error(std::string("svoh_create: ") + svoh_last_error_string(nullptr));
    const svoh_camera& cam = rig.at(0).cam;
    FeatureTrackerHip tracker(ctx, params.tracker, 1);
    tracker.setDetectors({ std::make_shared<DetectorHip>(ctx, params.detector, cam.width, cam.height) });
    FILE* ft = fopen((out_dir + "/tracks.csv").c_str(), "w");
    FILE* fm = fopen((out_dir + "/timing.csv").c_str(), "w");
    if (!ft || !fm) throw std::runtime_error("cannot write into " + out_dir);
    fprintf(ft, "frame,timestamp_ns,track_id,x,y\n");
    fprintf(fm, "frame,n_tracked,n_features,ms_read,ms_pyramid,ms_track_detect\n");
    double sum_ms = 0.0;
    size_t n_done = 0;
    for (size_t k = 0; k < seq.size() && k < max_frames; ++k) {
      const double t0 = now_ms();
      const io::GrayImage img = io::readPngGray(seq.cam0_files[k]);
      if (img.width != cam.width || img.height != cam.height) throw std::runtime_error(seq.cam0_files[k] + ": size differs from the calibration");
      const double t1 = now_ms();
      FramePtr frame(new Frame, [ctx](Frame* f) { if (f->pyramid) svoh_release_frame(ctx, f->pyramid); delete f; });
      if (svoh_build_pyramid(ctx, img.data.data(), img.width, img.height, img.width, SVOH_MEM_HOST, params.n_pyr_levels_to_build,
                             SVOH_HALFSAMPLE_REFERENCE, nullptr, &frame->pyramid) != SVOH_OK)
        throw std::runtime_error(std::string("svoh_build_pyramid: ") + svoh_last_error_string(ctx));
      frame->cam = cam;
      frame->set_T_cam_imu(svoh::inverse(rig[0].T_B_C));
      frame->id_ = (int)k;
      FrameBundle::Ptr bundle(new FrameBundle);
      bundle->frames_.push_back(frame);
      const double t2 = now_ms();
      tracker.trackAndDetect(bundle);
      const double t3 = now_ms();
      const size_t n_tracked = tracker.getTotalActiveTracks();
      for (size_t i = 0; i < frame->num_features_; ++i)
        fprintf(ft, "%zu,%llu,%d,%.6f,%.6f\n", k, (unsigned long long)seq.cam_ts[k], frame->track_id_vec_[i], frame->px_vec_[2 * i],
                frame->px_vec_[2 * i + 1]);
      fprintf(fm, "%zu,%zu,%zu,%.4f,%.4f,%.4f\n", k, n_tracked, frame->num_features_, t1 - t0, t2 - t1, t3 - t2);
      sum_ms += (t3 - t1);
      ++n_done;
    }
    fclose(ft); fclose(fm);
    printf("svoh_track_sequence: %zu frames, %.3f ms/frame on the GPU path (pyramid + track/detect), %zu active tracks at the end\n",
           n_done, n_done ? sum_ms / n_done : 0.0, tracker.getTotalActiveTracks());
    tracker.reset();
    svoh_destroy(ctx);
    return 0;
  } catch (const std::exception& e) {
    fprintf(stderr, "svoh_track_sequence: %s\n", e.what());
    return 1;
  }
}
