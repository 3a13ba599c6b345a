// svo_hip_host_internal.h -- what svo_hip_host.cpp and svo_hip_lockstep.cpp share and no caller of the host layer sees:
// the speculative matcher batches of the reprojector (plan / replay), and the gathering of a depth-filter update.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "svo_hip_host.h"

namespace svo_hip {
// The host layer is compiled against include/svo_hip.h's struct layouts: a libsvo_hip.so of another SVOH_ABI_VERSION would
// read undersized / differently laid out structs.  Every mirror's constructor asks once.
inline void requireMatchingAbi()
{
  static const bool ok = [] {
    if (svoh_abi_version() != SVOH_ABI_VERSION)
      throw std::runtime_error("libsvo_hip.so has ABI version " + std::to_string(svoh_abi_version()) + ", the host layer was built for " + std::to_string(SVOH_ABI_VERSION));
    return true;
  }();
  (void)ok;
}
namespace reprojector_utils {
// sortCandidatesByReprojStats; order[k] = where the k-th candidate of the sorted list stood before (empty for lists of fewer than two)
void sortCandidatesWithOrder(std::vector<reprojector::Candidate>& candidates, std::vector<uint32_t>* order);
std::vector<int32_t>& g_last_results_ref();   // lastMatchResults(), writable
}  // namespace reprojector_utils
namespace detail {

// The matcher work of matchCandidate (reprojector.cpp:384-486) for every candidate, speculatively: it depends on no
// match result, only on the candidate.  plan() resolves what each candidate matches against and appends it to one of
// two batches (findMatchDirect / updateSeed); the batches of SEVERAL candidate lists can share one pair of launches
// (enqueue() / finish() on a context of their own, or -- FrontendLockstep -- copied into a launch that holds the batches
// of many streams); replay() is the reference's loop over one list (:356-381), reading the finished batches.
enum Kind { kConvergedSeed = 0, kUnconvergedSeed = 1, kLandmark = 2, kNoCloseView = 3 };
// ref / point: plain pointers.  The frame is kept alive by SpeculativeMatches::frames (every resolved frame has a slot
// there), the landmark by its keyframe's landmark_vec_ -- a shared_ptr per candidate here costs two atomic operations
// per candidate and list, on the frame's critical path.
struct Resolved { Kind kind; Frame* ref; size_t idx; Point* point; int frame_slot; int batch_pos; };

struct Batch {
  // inputs, one entry per unit, in the order plan() met them
  std::vector<int32_t> ref_idx, level;
  std::vector<double> px, f, grad, depth, state, px_cur;
  std::vector<uint8_t> type;
  // resident: a unit names its feature by (ref_idx, fidx) -- the reference frames' columns live on the device
  // (Frame::features, svoh_features_upload) -- and px / f / grad / level stay empty
  bool resident = false;
  std::vector<int32_t> fidx;
  // outputs of a batch that ran on the owner's own context
  std::vector<int32_t> result, search_level;
  std::vector<double> f_cur, A;
  std::vector<uint8_t> success;
  // where replay() reads the finished batch: the vectors above (useOwnOutputs) or a slice of a launch shared with
  // other streams (the driver sets these).  state / type / px_cur are in-out arrays of the kernels.
  struct Out {
    const int32_t* result = nullptr; const int32_t* search_level = nullptr;
    const double* px_cur = nullptr; const double* f_cur = nullptr; const double* A = nullptr; const double* state = nullptr;
    const uint8_t* type = nullptr; const uint8_t* success = nullptr;
  } out;
  void useOwnOutputs()
  {
    out.result = result.data(); out.search_level = search_level.data(); out.px_cur = px_cur.data(); out.f_cur = f_cur.data();
    out.A = A.data(); out.state = state.data(); out.type = type.data(); out.success = success.data();
  }
  void reserve_more(size_t n)
  {
    const size_t m = size() + n;
    ref_idx.reserve(m); type.reserve(m);
    if (resident) fidx.reserve(m);
    else { level.reserve(m); px.reserve(2 * m); f.reserve(3 * m); grad.reserve(2 * m); }
    depth.reserve(m); px_cur.reserve(2 * m); state.reserve(4 * m);
  }
  void push(const Frame& r, size_t i, int slot)
  {
    ref_idx.push_back(slot); type.push_back(r.type_vec_[i]);
    if (resident) { fidx.push_back(static_cast<int32_t>(i)); return; }
    level.push_back(r.level_vec_[i]);
    const double* p = &r.px_vec_[2 * i]; px.push_back(p[0]); px.push_back(p[1]);
    const double* q = &r.f_vec_[3 * i]; f.push_back(q[0]); f.push_back(q[1]); f.push_back(q[2]);
    const double* g = &r.grad_vec_[2 * i]; grad.push_back(g[0]); grad.push_back(g[1]);
  }
  size_t size() const { return type.size(); }
  void clear()
  {
    ref_idx.clear(); level.clear(); fidx.clear(); result.clear(); search_level.clear(); px.clear(); f.clear(); grad.clear(); depth.clear();
    state.clear(); px_cur.clear(); f_cur.clear(); A.clear(); type.clear(); success.clear();
    out = Out();
  }
};

struct SpeculativeMatches {
  std::vector<FramePtr> frames;   // distinct reference frames of all lists
  Batch direct, seeds;
  int last_slot = -1;
  bool in_flight = false;
  void clear() { frames.clear(); direct.clear(); seeds.clear(); last_slot = -1; }
  void setResident(bool on) { direct.resident = seeds.resident = on; }
  int slot_of(const FramePtr& f);
  // what each candidate of one list matches against
  std::vector<Resolved> plan(const FramePtr& frame, const std::vector<reprojector::Candidate>& candidates);
  // both batches, queued back to back and sent to the device (svoh_matcher_begin_deferred / flush): nothing is waited
  // for.  finish() is the one wait (svoh_matcher_collect); the caller may work in between -- the reprojector sorts its
  // candidate lists there, which the matcher work does not depend on.
  void enqueue(svoh_ctx* ctx, const FramePtr& frame, bool affine_est_offset, bool affine_est_gain, double seed_sigma2_thresh);
  void finish(svoh_ctx* ctx);
  void run(svoh_ctx* ctx, const FramePtr& frame, bool affine_est_offset, bool affine_est_gain, double seed_sigma2_thresh)
  {
    enqueue(ctx, frame, affine_est_offset, affine_est_gain, seed_sigma2_thresh);
    finish(ctx);
  }
  // the reference's loop over one candidate list, in candidate order (reprojector.cpp:356-381)
  // select_on: when set (and max_n_features_per_frame > 0), WHICH candidates the loop tries and where it stops is asked of the
  // device (svoh_select_matches_batch on the finished batches' success flags) instead of being found by walking the grid here
  void replay(const FramePtr& frame, size_t max_n_features_per_frame, std::vector<reprojector::Candidate>& candidates,
              const std::vector<Resolved>& rs, OccupandyGrid2D& grid, reprojector::Statistics& stats, svoh_ctx* select_on = nullptr);
};

// Matcher defaults (matcher.h:39-54) + the two affine flags, as reprojector_utils::matchCandidates sets them
// (reprojector.cpp:352-354), and updateSeed's options as matchCandidate calls it (:403-413)
svoh_matcher_options reprojectorMatcherOptions(bool affine_est_offset, bool affine_est_gain);
svoh_depth_filter_options reprojectorSeedOptions(const Frame& cur_frame, double seed_sigma2_thresh);
svoh_frame_view viewOf(const Frame& f);

}  // namespace detail
}  // namespace svo_hip
