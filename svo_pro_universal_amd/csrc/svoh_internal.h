// svoh_internal.h -- context, device frames and shared helpers of libsvo_hip.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <algorithm>
#include <vector>

#include "../../include/svo_hip.h"

namespace svoh {

// one pyramid level in device memory (row-major u8)
struct DevImage {
  const uint8_t* data;
  int32_t w, h, pitch, pad;
};

// Kernels read pixel rows in 8/16-byte pieces (sparse_align.hip, ImgView<false>::row): the last piece of the
// last row of the last level may reach up to 15 bytes past the image, so every slab ends with readable padding.
constexpr size_t kSlabTailPad = 64;

// Released frame slabs, kept for the next frame of the same size.  A per-frame front end makes one pyramid and drops
// one per image: hipMalloc + hipFree are 2 x 10-100 us of host time and hipFree waits for the whole device -- also for
// the depth filter's seed update that the caller deliberately left in flight.  A slab taken from here may still be read
// by kernels queued before its frame was released: whoever fills it again does so on the context's stream, behind them.
struct SlabPool {
  struct Entry { void* ptr; size_t alloc; };
  static constexpr size_t kMaxEntries = 8;
  static constexpr size_t kMaxBytes = (size_t)256 << 20;
  std::vector<Entry> free_list;
  size_t held = 0;
  void* take(size_t alloc)
  {
    for (size_t i = free_list.size(); i-- > 0;)
      if (free_list[i].alloc == alloc) {
        void* p = free_list[i].ptr;
        free_list.erase(free_list.begin() + (long)i);
        held -= alloc;
        return p;
      }
    return nullptr;
  }
  bool give(void* p, size_t alloc)
  {
    if (alloc > kMaxBytes) return false;
    // full: the OLDEST entry goes (not the newcomer).  A context whose frames change size -- another camera, another number of
    // streams per multi-frame call -- would otherwise keep eight slabs nobody asks for again and pay a hipMalloc + hipFree for
    // every frame from then on; this way the pool has turned over after eight releases.
    while (!free_list.empty() && (free_list.size() >= kMaxEntries || held + alloc > kMaxBytes)) {
      held -= free_list.front().alloc;
      (void)hipFree(free_list.front().ptr);
      free_list.erase(free_list.begin());
    }
    free_list.push_back({ p, alloc });
    held += alloc;
    return true;
  }
  ~SlabPool() { for (const Entry& e : free_list) (void)hipFree(e.ptr); }
};

// Released blocks of resident feature columns (svoh_features_upload), for the next upload: the smallest block that is large
// enough.  Same reason as SlabPool: hipFree waits for the whole device -- so blocks are never freed one by one: a call's block
// is as large as the call's new keyframes (one stream's, or sixty-four streams'), a pool that only took near-equal sizes back
// filled up with blocks nobody asked for again and then paid a hipFree per release (0.66 ms per round of a group that had
// served 64 streams before it served 32).  When the pool is full its smaller half is freed in one go.
struct BlockPool {
  struct Entry { void* ptr; size_t alloc; };
  static constexpr size_t kMaxEntries = 1024;
  static constexpr size_t kMaxBytes = (size_t)2 << 30;
  std::vector<Entry> free_list;
  size_t held = 0;
  void* take(size_t want, size_t* got)
  {
    size_t best = free_list.size();
    for (size_t i = 0; i < free_list.size(); ++i)
      if (free_list[i].alloc >= want && (best == free_list.size() || free_list[i].alloc < free_list[best].alloc)) best = i;
    if (best == free_list.size()) return nullptr;
    void* p = free_list[best].ptr;
    *got = free_list[best].alloc;
    held -= *got;
    free_list[best] = free_list.back();
    free_list.pop_back();
    return p;
  }
  bool give(void* p, size_t alloc)
  {
    if (free_list.size() >= kMaxEntries || held + alloc > kMaxBytes) {
      std::sort(free_list.begin(), free_list.end(), [](const Entry& a, const Entry& b) { return a.alloc < b.alloc; });
      const size_t drop = (free_list.size() + 1) / 2;
      for (size_t i = 0; i < drop; ++i) { held -= free_list[i].alloc; (void)hipFree(free_list[i].ptr); }
      free_list.erase(free_list.begin(), free_list.begin() + (long)drop);
      if (held + alloc > kMaxBytes) return false;
    }
    free_list.push_back({ p, alloc });
    held += alloc;
    return true;
  }
  ~BlockPool() { for (const Entry& e : free_list) (void)hipFree(e.ptr); }
};
struct FeatureBlock {   // one upload call's device memory, shared by its sets
  void* ptr = nullptr;
  size_t alloc = 0;
  std::shared_ptr<BlockPool> pool;
  ~FeatureBlock() { if (ptr && !(pool && pool->give(ptr, alloc))) (void)hipFree(ptr); }
};
struct FeatureSet {
  std::shared_ptr<FeatureBlock> block;
  int n = 0;
  const double* px = nullptr; const double* f = nullptr; const double* grad = nullptr; const int32_t* level = nullptr;   // device
  // the same columns once more in TILE order (128x32-pixel tiles of px, row-major: the order a large seed batch is processed in,
  // so that the seeds of a workgroup are neighbours in both images) and perm[q] = the feature that stands at place q of that
  // order.  A keyframe's pixels never change: the order is computed once, at upload, instead of a counting sort per frame.
  const double* spx = nullptr; const double* sf = nullptr; const double* sgrad = nullptr; const int32_t* slevel = nullptr;
  const int32_t* perm = nullptr;
};
// the tile of a pixel in that order (also the bins of the per-frame counting sort of batches that carry their own columns)
constexpr int kBinShiftX = 7, kBinShiftY = 5;

// device allocation shared by the frames carved out of it
struct Slab {
  void* ptr = nullptr;
  size_t bytes = 0;
  size_t alloc = 0;                  // what was asked of hipMalloc (bytes + tail padding)
  std::shared_ptr<SlabPool> pool;    // where the allocation goes when the last frame of the slab is released
  ~Slab() { if (ptr && !(pool && pool->give(ptr, alloc))) (void)hipFree(ptr); }   // hipFree waits for the device
};

struct Frame {
  std::shared_ptr<Slab> slab;
  int n_levels = 0;
  DevImage lv[SVOH_MAX_LEVELS];
};

// grow-only device / pinned-host scratch buffers (never reallocated inside a
// timed loop once warmed up).  A regrowth frees and allocates (0.3 - 1 ms, and hipFree waits for the device): a buffer
// starts at 1 MB -- nothing next to 288 GB, and what a per-frame call of an EuRoC-sized front end never outgrows --
// and grows by half.
constexpr size_t kMinScratchBytes = (size_t)1 << 20;
struct DevBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (ptr) { (void)hipFree(ptr); ptr = nullptr; cap = 0; }
    size_t want = bytes + bytes / 2 + 4096;
    if (want < kMinScratchBytes) want = kMinScratchBytes;
    hipError_t e = hipMalloc(&ptr, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  ~DevBuffer() { if (ptr) (void)hipFree(ptr); }
};

struct PinnedBuffer {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes)
  {
    if (bytes <= cap) return hipSuccess;
    if (ptr) { (void)hipHostFree(ptr); ptr = nullptr; cap = 0; }
    size_t want = bytes + bytes / 2 + 4096;
    if (want < kMinScratchBytes) want = kMinScratchBytes;
    hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  ~PinnedBuffer() { if (ptr) (void)hipHostFree(ptr); }
};

}  // namespace svoh

// Tuning and diagnostic knobs: read from the environment (SVOH_*) ONCE, when the context is made, and again only on
// svoh_reload_knobs -- no launch path calls getenv.  kKnobUnset = "let the launch code decide".
constexpr int kKnobUnset = -2147483647 - 1;
struct SvohKnobs {
  int klt_block = kKnobUnset;                 // SVOH_KLT_BLOCK: 64 / 128 / 256 threads per KLT workgroup
  int matcher_g8 = kKnobUnset;                // SVOH_MATCHER_G8: 0 one lane per unit, 1 eight lanes, 2 packed, 3 one wave per unit (epipolar scans)
  int seed_binning = kKnobUnset;              // SVOH_SEED_BINNING: 0 = no spatial binning of large seed batches
  int pose_threads = kKnobUnset;              // SVOH_POSE_THREADS: 64 / 256 / 512
  int align_cluster = kKnobUnset;             // SVOH_ALIGN_CLUSTER: workgroups per problem (0 = never)
#ifdef SVOH_TEST_HOOKS
  int align_cluster_test_absent = kKnobUnset; // SVOH_ALIGN_CLUSTER_TEST_ABSENT: a partner that never arrives (libsvo_hip_testhooks.so only)
#endif
  int align_threads = kKnobUnset;             // SVOH_ALIGN_THREADS: 256 / 512 (anything else = 256)
  int align_rows = kKnobUnset;                // SVOH_ALIGN_ROWS: lanes per patch of the alignment's 512-thread geometry: 2, 4 or 8 (<= patch size); anything else = a lane per patch
  int align_latency_build = kKnobUnset;       // SVOH_ALIGN_LATENCY_BUILD: 0 = small launches use the batch build of the 256-thread kernel too
  int align_lds = kKnobUnset;                 // SVOH_ALIGN_LDS: bytes of LDS for image levels (rounded down to a multiple of 16)
  int align_wg_per_cu = kKnobUnset;           // SVOH_ALIGN_WG_PER_CU
  int kernel_timing = kKnobUnset;             // SVOH_KERNEL_TIMING: 1 = bracket every kernel with an event pair (svoh_set_kernel_timing)
  int copy_kernel = kKnobUnset;               // SVOH_COPY_KERNEL / svoh_set_copy_policy: 0 = staged blocks through hipMemcpyAsync, 1 (default) = copy kernels for 16 KB .. 1 MB, 2 = copy kernels always
  static int or_default(int v, int dflt) { return v == kKnobUnset ? dflt : v; }
};
void load_knobs_from_env(SvohKnobs& k);

struct svoh_ctx {
  SvohKnobs knobs;
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cus = 0;
  size_t lds_per_block = 0;
  std::string err;
  std::shared_ptr<svoh::SlabPool> slab_pool = std::make_shared<svoh::SlabPool>();   // declared before `frames`: outlives them
  std::unordered_map<uint64_t, svoh::Frame> frames;
  uint64_t next_frame_id = 1;
  // bumped whenever a frame or a feature set is released: what was derived from handles before (the staged batches' view
  // tables, matcher.hip) must be looked up again
  uint64_t handle_generation = 0;
  // resident feature columns (svoh_features_upload)
  std::shared_ptr<svoh::BlockPool> feature_pool = std::make_shared<svoh::BlockPool>();   // declared before the sets: outlives them
  std::unordered_map<uint64_t, svoh::FeatureSet> feature_sets;
  uint64_t next_features_id = 1;
  svoh::PinnedBuffer h_features;        // the upload's staging block ...
  hipEvent_t ev_features = nullptr;     // ... and what says its last copy has run (made at first use)

  // sparse-align workspaces
  svoh::DevBuffer d_desc;      // problem + camera descriptors
  svoh::DevBuffer d_results;   // svoh_align_result[n]
  svoh::DevBuffer d_feat;      // per-feature workspace
  svoh::DevBuffer d_eval;      // evaluate() outputs
  svoh::DevBuffer d_xchg;      // cluster mode of the alignment: exchange slots + arrival counter
  svoh::DevBuffer d_split;     // svoh_sparse_align_split_buffers: a Gauss-Newton state + 74 sums
  svoh::PinnedBuffer h_desc;           // the pinned staging block of even launches ...
  svoh::PinnedBuffer h_desc_odd;       // ... and of odd ones (see align_launches_since_drain)
  svoh::PinnedBuffer h_results;
  int last_align_n = 0;
  // results of the launches queued since the last fetch, launch after launch in h_results
  static constexpr size_t kMaxQueuedResults = (size_t)1 << 18;
  size_t align_pending_results = 0, align_last_results_off = 0;
  // ... and on the device: every launch since the last fetch has its own block of d_results (align_pending_dev results are
  // taken), so that a candidate projection queued behind SEVERAL launches can read the result of any of them;
  // align_result_dev_index[k] = where result #k of the queue lives in d_results
  size_t align_pending_dev = 0;
  std::vector<uint32_t> align_result_dev_index;
  // a ring of event pairs, one per alignment launch: callers that queue launches back to back (enqueue without
  // fetch) can still read every launch's device time afterwards
  static constexpr int kAlignEventRing = 32;
  hipEvent_t ev_align_start[kAlignEventRing] = {}, ev_align_stop[kAlignEventRing] = {};
  unsigned long long align_launches = 0;
  // The pinned staging blocks are reused.  A launch fetched before the next one is queued (the per-frame use) has
  // drained the stream: nothing to protect, no event on the stream (an event record costs a launch 4.5 us of stream time,
  // tools/svoh_call_overhead).  Launches queued back to back alternate between the two blocks; from the second one on
  // each records ev_align_staged behind its upload, and the third and later wait for the event of the launch before them
  // -- which lies behind the upload that last read their block.
  hipEvent_t ev_align_staged = nullptr;
  bool align_staged_event_valid = false;      // ev_align_staged was recorded by the most recent launch
  unsigned align_launches_since_drain = 0;    // alignment launches queued since this file last waited for the stream
  unsigned align_desc_slot = 0;
  bool align_shared_classes = false;   // svoh_set_align_geometry_classes
  bool align_no_cluster = false;   // svoh_sparse_align_batch repeating a launch whose cluster gave up
  hipEvent_t ev_misc_start = nullptr, ev_misc_stop = nullptr;  // KLT / matcher / seeds
  // inside the hook of svoh_optimize_pose_batch_hook: where the launched batch's results will be on the device
  bool in_pose_hook = false; const void* d_pose_results = nullptr; int n_pose_results = 0;
  hipEvent_t ev_pose_done = nullptr;   // svoh_optimize_pose_batch_hook: behind the copy of the results (made at first use)
  // svoh_build_pyramid_multi_prefetch: the next frames' images come up on a stream of their own, beside the chain's work
  hipStream_t upload_stream = nullptr;
  hipEvent_t ev_upload = nullptr;
  bool upload_pending = false;           // ev_upload recorded and not yet waited for by the context's stream (svoh_prefetch_fence)
  bool misc_timed = false;     // the last KLT / matcher / seed / pose / detector launch was bracketed by the event pair
  bool misc_launched = false;  // ... has happened at all (its work counters exist)
  unsigned long long align_timed_launches = 0;   // alignment launches bracketed by events (ring slots in use)
  bool align_last_timed = false;                 // ... and whether the most recent alignment launch was one of them
  bool timing_on() const { return knobs.kernel_timing != kKnobUnset && knobs.kernel_timing != 0; }
  svoh::DevBuffer d_counters;  // 8 x uint64 work counters of the last KLT / matcher kernel
  svoh::DevBuffer d_unit_counts;  // 4 x uint32 per unit
  size_t unit_counts_pending = 0; // units of the last launch whose counts have not been added up yet

  // matcher calls between svoh_matcher_begin_deferred / svoh_matcher_collect: host-array batches are queued without
  // a synchronisation (the direct matches and the seed updates of one reprojection share ONE round trip); what has
  // to be copied to the caller's arrays once the stream has drained is remembered here
  bool matcher_deferred = false;
  bool matcher_deferred_used[2] = { false, false };   // [0] direct, [1] seeds: one batch of each kind per section
  struct PendingCopy { void* dst; const void* src; size_t bytes; };
  std::vector<PendingCopy> matcher_pending;
  struct PendingCount { int32_t* dst; const uint8_t* flags; int n; };
  std::vector<PendingCount> matcher_pending_counts;
  struct DeferredLaunch {              // a matcher launch waiting for svoh_matcher_collect (kernel arguments as bytes)
    std::vector<uint8_t> args;
    int n = 0, g8 = 0;
    int max_w = 0, max_h = 0;          // largest reference frame (the packed geometry's spatial bins)
    size_t out_off = 0, out_bytes = 0; // packed geometry: the batch's output area in its device block, zeroed ahead of the kernel
    void* d_block = nullptr;
    bool valid = false;
    void* d2h_dst = nullptr; const void* d2h_src = nullptr; size_t d2h_bytes = 0;
    // the frame views of the batch in its staging blocks (svoh_matcher_deferred_set_cur_frame): n_ref reference frames, then the current one(s)
    void* views_h = nullptr; void* views_d = nullptr; int n_ref = 0, n_cur = 0;
    uint64_t cur_frame_handle = 0;
    // current views whose pose is composed on the device from the pose batch in flight (svoh_frame_view::pose_result_index_plus1)
    const void* d_pose_results = nullptr; int n_pose_results = 0; bool pose_from_results = false;
    // features named by index (svoh_feature_batch::feature_index): gathered from the reference frames' resident columns ahead of the kernels
    const void* d_fidx = nullptr;
    bool seed_block_staged = false;    // a staged seed batch (its device block can serve svoh_align_camera::pos_seed_unit after the flush)
  } matcher_deferred_launch[2];
  // staging of the deferred batches, one pair per kind: nothing else stages through them, so any other call made
  // inside the section (a device-resident batch, an epipolar batch, the detector -- all on d_scratch1 / h_scratch1)
  // can neither overwrite a queued batch's inputs before its copy has read them nor make reserve() free them
  svoh::DevBuffer d_match_seeds, d_match_direct;
  svoh::PinnedBuffer h_match_seeds, h_match_direct;
  // svoh_matcher_stage: the layout of the block handed out for the section's direct [0] / seed [1] batch
  struct MatcherStage {
    bool valid = false, want_outputs = false, resident = false;
    int n = 0, max_views = 0;
    size_t o_fidx = 0;
    size_t o_views = 0, o_idx = 0, o_cidx = 0, o_px = 0, o_f = 0, o_grad = 0, o_level = 0, o_type = 0, o_depth = 0, o_pxcur = 0, o_state = 0,
           o_result = 0, o_success = 0, o_fcur = 0, o_slevel = 0, o_hinv = 0, o_A = 0, o_nsucc = 0, in_total = 0, back_from = 0, total = 0;
  } matcher_stage[2];

  // candidate projection of the reprojector (svoh_project_candidates_enqueue / _collect): its own staging pair -- the
  // call is queued behind an alignment launch whose own staging is still in flight
  svoh::DevBuffer d_cand;
  svoh::PinnedBuffer h_cand;
  int cand_pending_n = 0;              // points of the queued call whose results wait in h_cand (0: nothing queued)
  size_t cand_out_off = 0;
  // svoh_project_candidates_stage / _enqueue_staged / _wait: many jobs, staged in place (blocks of their own)
  svoh::DevBuffer d_cand_multi;
  svoh::PinnedBuffer h_cand_multi;
  struct CandStage { int n_jobs = 0, n_kf = 0, n_points = 0; size_t o_jobs = 0, o_kf = 0, o_job = 0, o_kind = 0, o_idx = 0, o_v = 0, o_mu = 0, in_total = 0, o_px = 0, o_vis = 0, total = 0, o_ranges = 0, o_dev_ranges = 0, in_total_without_v = 0, o_mu_unit = 0; bool ranges = false; int state = 0; } cand_stage;   // state: 0 none, 1 staged, 2 in flight

  svoh::DevBuffer d_seed_bin;          // packed seed update: histogram, ranks, sorted records (nothing else writes here)
  void* seed_hist_ptr = nullptr;       // the binning histogram at this address ...
  size_t seed_hist_clean_keys = 0;     // ... is known to be zero for this many keys (its last pass clears it)

  // the frame views of the last staged matcher batch as they were asked for (bytes of the svoh_frame_view arrays) and as they were
  // resolved: a direct batch and a seed batch of one reprojection name the same frames, the second one copies the table
  std::vector<uint8_t> staged_views_key; std::vector<uint8_t> staged_views_resolved; uint64_t staged_views_generation = ~0ull;
  int staged_views_ref_levels = 0, staged_views_max_w = 1, staged_views_max_h = 1;
  // the staged seed batch that was sent off last, as long as its device block stands (svoh_align_camera::pos_seed_unit reads
  // the seeds' positions from it): where its views, reference indices, bearing vectors and states lie
  struct SeedBlock { bool valid = false; const void* views = nullptr; int n_ref = 0; const int32_t* ref_idx = nullptr; const double* f = nullptr; const double* state = nullptr; int n = 0; } seed_block;
  hipEvent_t ev_matcher_done = nullptr;   // behind the copies of a flushed section's results: what svoh_matcher_collect waits for
  bool matcher_done_recorded = false;
  // svoh_detect_cells_batch_enqueue / _collect: the batch in flight, its blocks, the event behind its results
  struct DetectPending { bool in_flight = false, edgelets = false; int n_frames = 0, n_cells = 0; size_t cell_stride = 0, o_ck = 0, o_ek = 0, o_ang = 0; } detect_pending;
  svoh::DevBuffer d_detect;
  svoh::PinnedBuffer h_detect;
  hipEvent_t ev_detect = nullptr;

  // generic scratch for the other paths
  svoh::DevBuffer d_scratch0, d_scratch1, d_scratch2;
  svoh::PinnedBuffer h_scratch0, h_scratch1;
  // svoh_optimize_points_batch_enqueue / _collect: a points batch queued and not waited for (buffers of its own, an event behind its copy back)
  svoh::DevBuffer d_points_q;
  svoh::PinnedBuffer h_points_q;
  hipEvent_t ev_points = nullptr;
  int points_pending = 0;               // points of the queued batch, or 0
  size_t points_o_pos = 0, points_o_it = 0;
};

namespace svoh {

int set_error(svoh_ctx* ctx, int code, const char* fmt, ...);
// zeroes the context's 8 work counters on its stream and returns the device pointer
int reset_counters(svoh_ctx* ctx, unsigned long long** out);
// per-unit work counts (4 x uint32 per unit) written by a kernel with plain stores; reduce_unit_counts
// sums them into the context's 8 counters AFTER the timed region (atomics on a few shared words
// inside the kernel would serialise it)
int reserve_unit_counts(svoh_ctx* ctx, size_t n_units, unsigned int** out);
// A block of results back to PINNED host memory, queued on the context's stream.  The runtime's copy of 32 KB and more
// goes through the DMA engine, which costs a round trip 4 - 5 us more than a kernel's own stores over PCIe
// (tools/svoh_call_overhead, h2d_kernel_copykernel_sync against h2d_kernel_d2h_sync: 32 KB 20.6 / 24.2 us, 128 KB
// 23.4 / 28.4; 4 KB 18.4 / 16.5 the other way round): blocks of 16 KB .. 1 MB, 16-byte aligned, go through a copy
// kernel, everything else through hipMemcpyAsync.  SVOH_COPY_KERNEL=0: always hipMemcpyAsync.
hipError_t svoh_copy_to_host(svoh_ctx* ctx, void* dst_pinned, const void* src_device, size_t bytes);
// The other direction: a staged input block from PINNED host memory to the device.  Same finding (h2d_kernel_d2h_sync
// against copyin_kernel_d2h_sync with SVOH_OVERHEAD_N_IN): 32 KB 27.4 / 24.7 us, 64 KB 33.7 / 29.0, 128 KB 48.8 / 37.9; 16 KB
// 18.1 / 22.6 the other way round -- blocks of 32 KB .. 1 MB are read by a copy kernel, the rest goes through hipMemcpyAsync.
hipError_t svoh_copy_to_device(svoh_ctx* ctx, void* dst_device, const void* src_pinned, size_t bytes);
int reduce_unit_counts(svoh_ctx* ctx, size_t n_units);
void set_global_error(const char* msg);
const Frame* find_frame(const svoh_ctx* ctx, svoh_frame_t id);
// svoh_align_camera::pos_seed_unit (matcher.hip): for every job, pos[3 i ..] of the features with unit[i] >= 0 from the seed batch
// in flight.  `jobs_device`: n_jobs entries in device memory, queued on the context's stream
struct PosFromSeedsJob { double* pos; const int32_t* unit; int32_t n; int32_t pad_; };
int svoh_launch_pos_from_seed_batch(svoh_ctx* ctx, int n_jobs, int max_n, const PosFromSeedsJob* jobs_device);

#if defined(__HIPCC__)
// Wave64 all-lanes integer sum on the VALU's DPP network (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror, row_bcast:15, row_bcast:31, then v_readlane of lane
// 63): 6 dependent v_add_u32 instead of 6 ds_bpermute round trips through the LDS
// crossbar that __shfl_xor lowers to on gfx950.  Integer adds: exact in any order.
__device__ __forceinline__ int wave_sum_i32_dpp(int v)
{
  v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
  v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
  v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, false);  // row_half_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, false);  // row_mirror
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2,3
  return __builtin_amdgcn_readlane(v, 63);
}
#endif

// No exception crosses the C ABI (include/svo_hip.h): every entry point is a function-try-block ending in this.
#define SVOH_ABI_CATCH(ctxexpr)                                                                                  \
  catch (const std::bad_alloc&) { return svoh::set_error((ctxexpr), SVOH_ERR_OUT_OF_MEMORY, "out of host memory"); } \
  catch (const std::exception& e_) { return svoh::set_error((ctxexpr), SVOH_ERR_HIP, "exception at the ABI: %s", e_.what()); } \
  catch (...) { return svoh::set_error((ctxexpr), SVOH_ERR_HIP, "unknown exception at the ABI"); }

#define SVOH_HIP_TRY(ctx, expr)                                                          \
  do {                                                                                   \
    hipError_t svoh_e_ = (expr);                                                         \
    if (svoh_e_ != hipSuccess)                                                           \
      return svoh::set_error((ctx), SVOH_ERR_HIP, "%s failed: %s (%s:%d)", #expr,        \
                             hipGetErrorString(svoh_e_), __FILE__, __LINE__);            \
  } while (0)

#define SVOH_REQUIRE(ctx, cond, msg)                                                     \
  do {                                                                                   \
    if (!(cond)) return svoh::set_error((ctx), SVOH_ERR_INVALID_ARGUMENT, "%s", (msg));  \
  } while (0)

}  // namespace svoh
