// context.hip -- context lifetime, error strings, device frames and the image
// pyramid kernel (a-0).
//
// Pyramid: replaces frame_utils::createImgPyramid
// (src/svo_common/src/frame.cpp:372-386) -> vk::halfSample
// (src/vikit/vikit_common/src/vision.cpp:19-44 SSE2 rule, :73-111 dispatch and
// scalar rule).  Integer work: results are bit-identical to the reference's.
#include <cstdarg>
#include <algorithm>
#include <cstring>

#include "svoh_internal.h"
#include "svoh_math.h"

namespace svoh {

static thread_local std::string g_global_err = "no error";

int set_error(svoh_ctx* ctx, int code, const char* fmt, ...)
{
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf; else g_global_err = buf;
  return code;
}

void set_global_error(const char* msg) { g_global_err = msg; }

const Frame* find_frame(const svoh_ctx* ctx, svoh_frame_t id)
{
  auto it = ctx->frames.find(id);
  return it == ctx->frames.end() ? nullptr : &it->second;
}

__global__ __launch_bounds__(256) void svoh_copy_to_host_kernel(uint4* dst, const uint4* src, size_t n16, uint8_t* dst_tail, const uint8_t* src_tail, int n_tail)
{
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
  __threadfence_system();
}

// Policy 2 (svoh_set_copy_policy; what a driver of MANY contexts on one device sets): EVERY staged block travels through a copy
// kernel, whatever its size.  A hipMemcpyAsync between two kernels of a stream is a hand-over between the compute queue and the
// copy engine and back; one stream does not notice (policy 1 keeps the runtime's copy where it is the faster call), but the
// machine takes only ~160 k such mixed dispatches per second from ALL streams together (tools/svoh_dispatch_rate: 2, 4 and 8
// threads alike, against > 500 k copy-kernel dispatches and > 1 M plain launches) -- which is where four lock-step groups
// with a handful of copies per round each ran into a wall.
static bool copy_by_kernel(const svoh_ctx* ctx, const void* a, const void* b, size_t bytes, size_t min_bytes)
{
  const int policy = SvohKnobs::or_default(ctx->knobs.copy_kernel, 1);
  if (policy == 0 || ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15)) return false;
  if (policy >= 2) return bytes >= 16;
  return bytes >= min_bytes && bytes <= ((size_t)1 << 20);
}
static unsigned copy_blocks(size_t n16)
{
  // 128 KB in flight per 32 workgroups; a block of several MB gets more of them so that the link stays full
  size_t blocks = (n16 + 255) / 256;
  const size_t cap = n16 > ((size_t)1 << 16) ? 128 : 32;
  return (unsigned)(blocks > cap ? cap : (blocks ? blocks : 1));
}

hipError_t svoh_copy_to_host(svoh_ctx* ctx, void* dst_pinned, const void* src_device, size_t bytes)
{
  if (bytes == 0) return hipSuccess;
  const bool by_kernel = copy_by_kernel(ctx, dst_pinned, src_device, bytes, (size_t)16 << 10);
  if (!by_kernel) return hipMemcpyAsync(dst_pinned, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream);
  const size_t n16 = bytes / 16;
  const int n_tail = (int)(bytes - n16 * 16);
  const unsigned blocks = copy_blocks(n16);
  hipLaunchKernelGGL(svoh_copy_to_host_kernel, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<uint4*>(dst_pinned),
                     static_cast<const uint4*>(src_device), n16, static_cast<uint8_t*>(dst_pinned) + n16 * 16,
                     static_cast<const uint8_t*>(src_device) + n16 * 16, n_tail);
  return hipGetLastError();
}

hipError_t svoh_copy_to_device(svoh_ctx* ctx, void* dst_device, const void* src_pinned, size_t bytes)
{
  if (bytes == 0) return hipSuccess;
  const bool by_kernel = copy_by_kernel(ctx, dst_device, src_pinned, bytes, (size_t)32 << 10);
  if (!by_kernel) return hipMemcpyAsync(dst_device, src_pinned, bytes, hipMemcpyHostToDevice, ctx->stream);
  const size_t n16 = bytes / 16;
  const int n_tail = (int)(bytes - n16 * 16);
  const unsigned blocks = copy_blocks(n16);
  hipLaunchKernelGGL(svoh_copy_to_host_kernel, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<uint4*>(dst_device),
                     static_cast<const uint4*>(src_pinned), n16, static_cast<uint8_t*>(dst_device) + n16 * 16,
                     static_cast<const uint8_t*>(src_pinned) + n16 * 16, n_tail);
  return hipGetLastError();
}

int reset_counters(svoh_ctx* ctx, unsigned long long** out)
{
  SVOH_HIP_TRY(ctx, ctx->d_counters.reserve(8 * sizeof(unsigned long long)));
  *out = static_cast<unsigned long long*>(ctx->d_counters.ptr);   // zeroed when the totals are asked for
  return SVOH_OK;
}

__global__ __launch_bounds__(256) void reduce_unit_counts_kernel(const unsigned int* __restrict__ counts, size_t n,
                                                                  unsigned long long* __restrict__ out)
{
  // [0..3] sums of the four per-unit counters; [4..7] tails of the two loop counters (units with >= 5 / >= 10 of
  // the third counter, >= 20 / >= 50 of the second): how uneven the units of a wave are
  unsigned long long acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const uint4 c = reinterpret_cast<const uint4*>(counts)[i];
    acc[0] += c.x; acc[1] += c.y; acc[2] += c.z; acc[3] += c.w;
    acc[4] += c.z >= 5u; acc[5] += c.z >= 10u; acc[6] += c.y >= 20u; acc[7] += c.y >= 50u;
  }
  __shared__ unsigned long long s[4][8];
  for (int k = 0; k < 8; ++k) {
    unsigned long long v = acc[k];
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) atomicAdd(&out[threadIdx.x], s[0][threadIdx.x] + s[1][threadIdx.x] + s[2][threadIdx.x] + s[3][threadIdx.x]);
}

int reserve_unit_counts(svoh_ctx* ctx, size_t n_units, unsigned int** out)
{
  SVOH_HIP_TRY(ctx, ctx->d_unit_counts.reserve(n_units * 4 * sizeof(unsigned int)));   // every unit writes its own entry
  *out = static_cast<unsigned int*>(ctx->d_unit_counts.ptr);
  return SVOH_OK;
}

// The totals are a diagnostic: the per-unit counts of a launch are only added up when svoh_last_kernel_counters
// asks for them (a memset and a kernel less on every tracking / matching call).
int reduce_unit_counts(svoh_ctx* ctx, size_t n_units)
{
  ctx->unit_counts_pending = n_units;
  return SVOH_OK;
}

static int reduce_unit_counts_now(svoh_ctx* ctx, size_t n_units)
{
  SVOH_HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters.ptr, 0, 8 * sizeof(unsigned long long), ctx->stream));
  const int blocks = (int)((n_units + 255) / 256 > 64 ? 64 : (n_units + 255) / 256);
  hipLaunchKernelGGL(reduce_unit_counts_kernel, dim3(blocks), dim3(256), 0, ctx->stream,
                     static_cast<const unsigned int*>(ctx->d_unit_counts.ptr), n_units,
                     static_cast<unsigned long long*>(ctx->d_counters.ptr));
  SVOH_HIP_TRY(ctx, hipGetLastError());
  return SVOH_OK;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// layout of one frame inside a slab: levels tightly packed (pitch == width),
// each level starting on a 256-byte boundary
static size_t frame_layout(int w, int h, int n_levels, size_t* offs, int* ws, int* hs)
{
  size_t off = 0;
  for (int i = 0; i < n_levels; ++i) {
    offs[i] = off; ws[i] = w; hs[i] = h;
    off = align_up(off + (size_t)w * h, 256);
    w /= 2; h /= 2;
  }
  return off;
}

// ---------------------------------------------------------------------------
// halfSample kernel: one thread makes 4 horizontally adjacent output pixels
// from 2 rows x 8 input bytes.  Memory-bound: reads 4 B, writes 1 B per output
// pixel, all accesses coalesced (8-byte loads, 4-byte stores when aligned).
// grid.z indexes the image of a batch.
// ---------------------------------------------------------------------------
template <bool SSE2_RULE>
__device__ __forceinline__ unsigned half4(unsigned a, unsigned b, unsigned c, unsigned d)
{
  if (SSE2_RULE) {
    // _mm_avg_epu8(top,bottom) then _mm_avg_epu16(even,odd): round half up twice
    const unsigned v0 = (a + c + 1u) >> 1;
    const unsigned v1 = (b + d + 1u) >> 1;
    return (v0 + v1 + 1u) >> 1;
  }
  return (a + b + c + d) >> 2;  // (a+b+c+d)/4, truncating
}

template <bool SSE2_RULE>
__global__ __launch_bounds__(256) void half_sample_kernel(
    const uint8_t* __restrict__ in, size_t in_image_stride, int in_pitch,
    uint8_t* __restrict__ out, size_t out_image_stride, int out_pitch,
    int out_w_written, int out_h_written)
{
  const int x4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;  // first output column
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x4 >= out_w_written || y >= out_h_written) return;
  const uint8_t* top = in + (size_t)blockIdx.z * in_image_stride + (size_t)(2 * y) * in_pitch + 2 * x4;
  const uint8_t* bot = top + in_pitch;
  uint8_t* o = out + (size_t)blockIdx.z * out_image_stride + (size_t)y * out_pitch + x4;
  const int n = min(4, out_w_written - x4);
  if (n == 4 && ((reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(bot)) & 7) == 0 &&
      (reinterpret_cast<uintptr_t>(o) & 3) == 0) {
    const uint2 t = *reinterpret_cast<const uint2*>(top);
    const uint2 b = *reinterpret_cast<const uint2*>(bot);
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned tw = (k < 2) ? t.x : t.y;
      const unsigned bw = (k < 2) ? b.x : b.y;
      const int sh = (k & 1) * 16;
      const unsigned a = (tw >> sh) & 0xff, bb = (tw >> (sh + 8)) & 0xff;
      const unsigned c = (bw >> sh) & 0xff, d = (bw >> (sh + 8)) & 0xff;
      r |= half4<SSE2_RULE>(a, bb, c, d) << (8 * k);
    }
    *reinterpret_cast<unsigned*>(o) = r;
  } else {
    for (int k = 0; k < n; ++k)
      o[k] = (uint8_t)half4<SSE2_RULE>(top[2 * k], top[2 * k + 1], bot[2 * k], bot[2 * k + 1]);
  }
}

// One pyramid step for a batch of images living at a fixed stride.
static hipError_t launch_half_sample(hipStream_t s, const uint8_t* in, size_t in_stride, int in_w, int in_h,
                                     int in_pitch, uint8_t* out, size_t out_stride, int out_pitch,
                                     int n_images, int rounding)
{
  const int out_w = in_w / 2, out_h = in_h / 2;
  bool sse = false;
  if (rounding == SVOH_HALFSAMPLE_SSE2) sse = true;
  else if (rounding == SVOH_HALFSAMPLE_REFERENCE) sse = (in_w % 16 == 0) && (in_pitch == in_w);
  // the SSE2 routine writes (w>>4)*8 columns and h>>1 rows; the scalar one out_w x out_h
  const int ww = sse ? (in_w >> 4) * 8 : out_w;
  const int hw = sse ? (in_h >> 1) : out_h;
  if (ww <= 0 || hw <= 0 || n_images <= 0) return hipSuccess;
  dim3 block(64, 4, 1);
  dim3 grid((unsigned)((ww + 4 * 64 - 1) / (4 * 64)), (unsigned)((hw + 3) / 4), (unsigned)n_images);
  if (sse)
    hipLaunchKernelGGL(half_sample_kernel<true>, grid, block, 0, s, in, in_stride, in_pitch, out, out_stride,
                       out_pitch, ww, hw);
  else
    hipLaunchKernelGGL(half_sample_kernel<false>, grid, block, 0, s, in, in_stride, in_pitch, out, out_stride,
                       out_pitch, ww, hw);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Up to four halfSample steps in ONE launch: a workgroup takes a 64x64 tile of the input level, makes the 32x32 tile of
// the next level from global memory (as half_sample_kernel does: four outputs per thread from 2 x 8 bytes) and the
// 16x16, 8x8 and 4x4 tiles after it from LDS; every level is written to the slab.  64 is a multiple of 2^4, so a tile
// needs no neighbour's pixels.  Each step uses its own rounding rule (bit k of sse_mask), as the per-level launches
// do: same integers.  Three launches less per frame and levels 1..3 are not read back from memory.
// ---------------------------------------------------------------------------
struct PyramidArgs {
  uint8_t* base;           // level 0 of image 0
  size_t frame_stride;     // bytes between the images of a batch
  size_t off[5];           // byte offset of level k inside a frame
  int w[5], h[5];          // level sizes (pitch == width)
  int n_steps;             // 2..4
  unsigned sse_mask;       // bit k: step k -> k+1 rounds with the SSE2 rule
};

__device__ __forceinline__ unsigned half4_rule(bool sse, unsigned a, unsigned b, unsigned c, unsigned d)
{
  return sse ? half4<true>(a, b, c, d) : half4<false>(a, b, c, d);
}

__global__ __launch_bounds__(256) void pyramid_fused_kernel(const PyramidArgs a)
{
  __shared__ uint8_t s1[32 * 32], s2[16 * 16], s3[8 * 8];
  uint8_t* img = a.base + (size_t)blockIdx.z * a.frame_stride;
  const int t = (int)threadIdx.x;
  const int x0 = (int)blockIdx.x * 64, y0 = (int)blockIdx.y * 64;   // tile origin at the input level
  {
    // step 0: 32x32 outputs, thread t -> row t / 8, columns 4 * (t % 8) .. + 3
    const bool sse = (a.sse_mask & 1u) != 0;
    const int r = t >> 3, c4 = (t & 7) * 4;
    const int gx = x0 / 2 + c4, gy = y0 / 2 + r;
    const int n = gy < a.h[1] ? min(4, a.w[1] - gx) : 0;
    if (n > 0) {
      const uint8_t* top = img + a.off[0] + (size_t)(2 * gy) * a.w[0] + 2 * gx;
      const uint8_t* bot = top + a.w[0];
      uint8_t* o = img + a.off[1] + (size_t)gy * a.w[1] + gx;
      unsigned v[4] = { 0, 0, 0, 0 };
      if (n == 4 && ((reinterpret_cast<uintptr_t>(top) | reinterpret_cast<uintptr_t>(bot)) & 7) == 0) {
        const uint2 tt = *reinterpret_cast<const uint2*>(top);
        const uint2 bb = *reinterpret_cast<const uint2*>(bot);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned tw = (k < 2) ? tt.x : tt.y, bw = (k < 2) ? bb.x : bb.y;
          const int sh = (k & 1) * 16;
          v[k] = half4_rule(sse, (tw >> sh) & 0xff, (tw >> (sh + 8)) & 0xff, (bw >> sh) & 0xff, (bw >> (sh + 8)) & 0xff);
        }
      } else {
        for (int k = 0; k < n; ++k) v[k] = half4_rule(sse, top[2 * k], top[2 * k + 1], bot[2 * k], bot[2 * k + 1]);
      }
      if (n == 4 && (reinterpret_cast<uintptr_t>(o) & 3) == 0) *reinterpret_cast<unsigned*>(o) = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
      else for (int k = 0; k < n; ++k) o[k] = (uint8_t)v[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) s1[r * 32 + c4 + k] = (uint8_t)v[k];
    }
  }
  __syncthreads();
  // steps 1..3 from LDS: a pixel inside its level has both its source rows and columns inside the level below
  if (a.n_steps >= 2) {
    const int r = t >> 4, c = t & 15;
    const int gx = x0 / 4 + c, gy = y0 / 4 + r;
    if (gx < a.w[2] && gy < a.h[2]) {
      const uint8_t* p = s1 + (2 * r) * 32 + 2 * c;
      const unsigned v = half4_rule((a.sse_mask & 2u) != 0, p[0], p[1], p[32], p[33]);
      s2[r * 16 + c] = (uint8_t)v;
      img[a.off[2] + (size_t)gy * a.w[2] + gx] = (uint8_t)v;
    }
  }
  __syncthreads();
  if (a.n_steps >= 3 && t < 64) {
    const int r = t >> 3, c = t & 7;
    const int gx = x0 / 8 + c, gy = y0 / 8 + r;
    if (gx < a.w[3] && gy < a.h[3]) {
      const uint8_t* p = s2 + (2 * r) * 16 + 2 * c;
      const unsigned v = half4_rule((a.sse_mask & 4u) != 0, p[0], p[1], p[16], p[17]);
      s3[r * 8 + c] = (uint8_t)v;
      img[a.off[3] + (size_t)gy * a.w[3] + gx] = (uint8_t)v;
    }
  }
  __syncthreads();
  if (a.n_steps >= 4 && t < 16) {
    const int r = t >> 2, c = t & 3;
    const int gx = x0 / 16 + c, gy = y0 / 16 + r;
    if (gx < a.w[4] && gy < a.h[4]) {
      const uint8_t* p = s3 + (2 * r) * 8 + 2 * c;
      img[a.off[4] + (size_t)gy * a.w[4] + gx] = (uint8_t)half4_rule((a.sse_mask & 8u) != 0, p[0], p[1], p[8], p[9]);
    }
  }
}

// the rounding rule of one step, and whether it writes the whole next level (see launch_half_sample)
static bool step_rule(int rounding, int in_w, int in_h, bool* writes_all)
{
  bool sse = false;
  if (rounding == SVOH_HALFSAMPLE_SSE2) sse = true;
  else if (rounding == SVOH_HALFSAMPLE_REFERENCE) sse = (in_w % 16 == 0);
  *writes_all = !sse || ((in_w >> 4) * 8 == in_w / 2);
  return sse;
}

// Level 0 of up to 256 images that live at separate page-locked host addresses (one per camera stream) into the frames
// of one slab: ONE launch reads them all over PCIe, 16 bytes per lane, instead of one copy call per image.
struct GatherArgs {
  uint8_t* base;            // level 0 of frame 0
  size_t frame_stride;
  int width, height, pitch; // source geometry
  const uint8_t* src[256];
};
__global__ __launch_bounds__(256) void gather_images_kernel(const GatherArgs a)
{
  const uint8_t* src = a.src[blockIdx.y];
  uint8_t* dst = a.base + (size_t)blockIdx.y * a.frame_stride;
  const size_t n = (size_t)a.width * a.height;
  if (a.pitch == a.width && !(reinterpret_cast<uintptr_t>(src) & 15) && !(reinterpret_cast<uintptr_t>(dst) & 15)) {
    const size_t n16 = n / 16;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256)
      reinterpret_cast<uint4*>(dst)[i] = reinterpret_cast<const uint4*>(src)[i];
    for (size_t i = n16 * 16 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
  } else {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
      const size_t y = i / (size_t)a.width, x = i - y * (size_t)a.width;
      dst[i] = src[y * (size_t)a.pitch + x];
    }
  }
}

// a slab of `bytes` (+ tail padding): from the context's pool of released slabs when one of that size is there
static hipError_t make_slab(svoh_ctx* ctx, size_t bytes, std::shared_ptr<Slab>* out)
{
  auto slab = std::make_shared<Slab>();
  slab->alloc = bytes + kSlabTailPad;
  slab->ptr = ctx->slab_pool->take(slab->alloc);
  if (!slab->ptr) {
    hipError_t e = hipMalloc(&slab->ptr, slab->alloc);
    if (e != hipSuccess) { slab->ptr = nullptr; return e; }
  }
  slab->bytes = bytes;
  slab->pool = ctx->slab_pool;
  *out = std::move(slab);
  return hipSuccess;
}

static uint64_t register_frame(svoh_ctx* ctx, const std::shared_ptr<Slab>& slab, uint8_t* base, int w, int h,
                               int n_levels)
{
  size_t offs[SVOH_MAX_LEVELS];
  int ws[SVOH_MAX_LEVELS], hs[SVOH_MAX_LEVELS];
  frame_layout(w, h, n_levels, offs, ws, hs);
  Frame f;
  f.slab = slab;
  f.n_levels = n_levels;
  for (int i = 0; i < n_levels; ++i) {
    f.lv[i].data = base + offs[i];
    f.lv[i].w = ws[i]; f.lv[i].h = hs[i]; f.lv[i].pitch = ws[i]; f.lv[i].pad = 0;
  }
  const uint64_t id = ctx->next_frame_id++;
  ctx->frames.emplace(id, std::move(f));
  return id;
}

}  // namespace svoh

using namespace svoh;

// svoh_camera_maths: project3 (+ Jacobian) and backProject3 of svoh_math.h as the device compiles them
__global__ __launch_bounds__(64) void camera_maths_kernel(const svoh_camera cam, int n, const double* xyz, double* px, double* J, double* f_back)
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const CamModel cm = load_camera(cam);
  const Vec3 p = { xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2] };
  double u, v;
  project3(cm, p, u, v);
  px[2 * i] = u; px[2 * i + 1] = v;
  double Jl[6];
  project3_jacobian(cm, p, Jl);
  for (int k = 0; k < 6; ++k) J[6 * i + k] = Jl[k];
  const Vec3 f = back_project3(cm, u, v);
  f_back[3 * i] = f.x; f_back[3 * i + 1] = f.y; f_back[3 * i + 2] = f.z;
}

void load_knobs_from_env(SvohKnobs& k)
{
  auto get = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : kKnobUnset; };
  k.klt_block = get("SVOH_KLT_BLOCK");
  k.matcher_g8 = get("SVOH_MATCHER_G8");
  k.seed_binning = get("SVOH_SEED_BINNING");
  k.pose_threads = get("SVOH_POSE_THREADS");
  k.align_cluster = get("SVOH_ALIGN_CLUSTER");
#ifdef SVOH_TEST_HOOKS
  k.align_cluster_test_absent = get("SVOH_ALIGN_CLUSTER_TEST_ABSENT");
#endif
  k.align_threads = get("SVOH_ALIGN_THREADS");
  k.align_rows = get("SVOH_ALIGN_ROWS");
  k.align_latency_build = get("SVOH_ALIGN_LATENCY_BUILD");
  k.align_lds = get("SVOH_ALIGN_LDS");
  k.align_wg_per_cu = get("SVOH_ALIGN_WG_PER_CU");
  k.kernel_timing = get("SVOH_KERNEL_TIMING");
  k.copy_kernel = get("SVOH_COPY_KERNEL");
}

static int build_pyramid_levels(svoh_ctx* ctx, hipStream_t stream, const std::shared_ptr<Slab>& slab, uint8_t* base, size_t fbytes, const size_t* offs, const int* ws,
                                const int* hs, int n_images, int width, int height, int n_levels, int rounding, svoh_frame_t* out_frames);

extern "C" {

int svoh_abi_version(void) { return SVOH_ABI_VERSION; }

int svoh_set_kernel_timing(svoh_ctx* ctx, int enabled)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  ctx->knobs.kernel_timing = enabled ? 1 : 0;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_set_copy_policy(svoh_ctx* ctx, int policy)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, policy >= 0 && policy <= 2, "copy policy: 0 the runtime's copies, 1 copy kernels for 16 KB .. 1 MB, 2 copy kernels always");
  ctx->knobs.copy_kernel = policy;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_reload_knobs(svoh_ctx* ctx)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  const int timing = ctx->knobs.kernel_timing, copy_policy = ctx->knobs.copy_kernel;
  load_knobs_from_env(ctx->knobs);
  if (ctx->knobs.kernel_timing == kKnobUnset) ctx->knobs.kernel_timing = timing;   // set by svoh_set_kernel_timing, not by the environment
  if (ctx->knobs.copy_kernel == kKnobUnset) ctx->knobs.copy_kernel = copy_policy;  // ... by svoh_set_copy_policy
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_create(int device, svoh_ctx** out_ctx)
try {
  if (!out_ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "out_ctx is NULL");
  *out_ctx = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return set_error(nullptr, SVOH_ERR_NO_DEVICE, "no HIP device available (%s); libsvo_hip has no CPU path",
                     e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
  if (device < 0 || device >= n)
    return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "device %d out of range [0,%d)", device, n);
  svoh_ctx* ctx = new (std::nothrow) svoh_ctx();
  if (!ctx) return set_error(nullptr, SVOH_ERR_OUT_OF_MEMORY, "out of host memory");
  ctx->device = device;
  ctx->err = "no error";
  load_knobs_from_env(ctx->knobs);
  e = hipSetDevice(device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
  for (int k = 0; k < svoh_ctx::kAlignEventRing; ++k) {
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_align_start[k]);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev_align_stop[k]);
  }
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_align_staged, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev_misc_start);
  if (e == hipSuccess) e = hipEventCreate(&ctx->ev_misc_stop);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) {
    set_error(nullptr, SVOH_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
    delete ctx;
    return SVOH_ERR_HIP;
  }
  ctx->num_cus = prop.multiProcessorCount;
  ctx->lds_per_block = prop.sharedMemPerBlock;
  *out_ctx = ctx;
  return SVOH_OK;
} SVOH_ABI_CATCH(nullptr)

int svoh_destroy(svoh_ctx* ctx)
try {
  if (!ctx) return SVOH_OK;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); }
  if (ctx->upload_stream) { (void)hipStreamSynchronize(ctx->upload_stream); }
  ctx->frames.clear();
  for (int k = 0; k < svoh_ctx::kAlignEventRing; ++k) {
    if (ctx->ev_align_start[k]) (void)hipEventDestroy(ctx->ev_align_start[k]);
    if (ctx->ev_align_stop[k]) (void)hipEventDestroy(ctx->ev_align_stop[k]);
  }
  if (ctx->ev_align_staged) (void)hipEventDestroy(ctx->ev_align_staged);
  if (ctx->ev_misc_start) (void)hipEventDestroy(ctx->ev_misc_start);
  if (ctx->ev_misc_stop) (void)hipEventDestroy(ctx->ev_misc_stop);
  if (ctx->ev_pose_done) (void)hipEventDestroy(ctx->ev_pose_done);
  if (ctx->ev_features) (void)hipEventDestroy(ctx->ev_features);
  if (ctx->ev_points) (void)hipEventDestroy(ctx->ev_points);
  if (ctx->ev_detect) (void)hipEventDestroy(ctx->ev_detect);
  if (ctx->ev_matcher_done) (void)hipEventDestroy(ctx->ev_matcher_done);
  if (ctx->upload_stream) { (void)hipStreamSynchronize(ctx->upload_stream); (void)hipStreamDestroy(ctx->upload_stream); }
  if (ctx->ev_upload) (void)hipEventDestroy(ctx->ev_upload);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

const char* svoh_last_error_string(const svoh_ctx* ctx)
{
  return ctx ? ctx->err.c_str() : g_global_err.c_str();
}

int svoh_synchronize(svoh_ctx* ctx)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

void* svoh_stream(svoh_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int svoh_last_kernel_counters(svoh_ctx* ctx, uint64_t out[8])
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, out != nullptr && ctx->misc_launched && ctx->d_counters.ptr, "no counters yet");
  if (ctx->unit_counts_pending) {
    const int rc = reduce_unit_counts_now(ctx, ctx->unit_counts_pending);
    if (rc != SVOH_OK) return rc;
    ctx->unit_counts_pending = 0;
  }
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_counters.ptr, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_last_kernel_ms(svoh_ctx* ctx, float* ms)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, ms != nullptr && ctx->misc_timed, "the last KLT / matcher / seed / pose launch was not timed (svoh_set_kernel_timing), or there was none");
  SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_misc_stop));
  SVOH_HIP_TRY(ctx, hipEventElapsedTime(ms, ctx->ev_misc_start, ctx->ev_misc_stop));
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_upload_pyramid(svoh_ctx* ctx, int n_levels, const uint8_t* const* level_data, const int* width,
                        const int* height, const int* pitch, svoh_frame_t* out_frame)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, out_frame && level_data && width && height && pitch, "NULL argument");
  SVOH_REQUIRE(ctx, n_levels >= 1 && n_levels <= SVOH_MAX_LEVELS, "n_levels out of range");
  for (int i = 0; i < n_levels; ++i) {
    SVOH_REQUIRE(ctx, level_data[i] && width[i] > 0 && height[i] > 0 && pitch[i] >= width[i], "bad level");
    // the frame layout assumes the reference's rows/2 x cols/2 rule
    if (i > 0) SVOH_REQUIRE(ctx, width[i] == width[i - 1] / 2 && height[i] == height[i - 1] / 2,
                            "level sizes must halve (createImgPyramid, frame.cpp:381-384)");
  }
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t offs[SVOH_MAX_LEVELS]; int ws[SVOH_MAX_LEVELS], hs[SVOH_MAX_LEVELS];
  const size_t bytes = frame_layout(width[0], height[0], n_levels, offs, ws, hs);
  std::shared_ptr<Slab> slab;
  SVOH_HIP_TRY(ctx, make_slab(ctx, bytes, &slab));
  uint8_t* base = static_cast<uint8_t*>(slab->ptr);
  for (int i = 0; i < n_levels; ++i)
    SVOH_HIP_TRY(ctx, hipMemcpy2DAsync(base + offs[i], (size_t)ws[i], level_data[i], (size_t)pitch[i],
                                       (size_t)ws[i], (size_t)hs[i], hipMemcpyHostToDevice, ctx->stream));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *out_frame = register_frame(ctx, slab, base, width[0], height[0], n_levels);
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_build_pyramid_batch(svoh_ctx* ctx, const uint8_t* img, size_t image_stride, int n_images, int width,
                             int height, int pitch, int mem_space, int n_levels, int rounding,
                             svoh_frame_t* out_frames)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, img && out_frames, "NULL argument");
  SVOH_REQUIRE(ctx, n_images >= 1 && width > 0 && height > 0 && pitch >= width, "bad image geometry");
  SVOH_REQUIRE(ctx, n_levels >= 1 && n_levels <= SVOH_MAX_LEVELS, "n_levels out of range");
  SVOH_REQUIRE(ctx, rounding >= 0 && rounding <= 2, "bad rounding mode");
  SVOH_REQUIRE(ctx, (width >> (n_levels - 1)) > 0 && (height >> (n_levels - 1)) > 0, "too many levels");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t offs[SVOH_MAX_LEVELS]; int ws[SVOH_MAX_LEVELS], hs[SVOH_MAX_LEVELS];
  const size_t fbytes = frame_layout(width, height, n_levels, offs, ws, hs);
  std::shared_ptr<Slab> slab;
  SVOH_HIP_TRY(ctx, make_slab(ctx, fbytes * (size_t)n_images, &slab));
  uint8_t* base = static_cast<uint8_t*>(slab->ptr);
  // level 0: copy into the tightly packed layout
  const hipMemcpyKind kind = mem_space == SVOH_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (pitch == width && image_stride == (size_t)width * height && fbytes == image_stride) {
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(base, img, image_stride * n_images, kind, ctx->stream));
  } else {
    for (int i = 0; i < n_images; ++i)
      SVOH_HIP_TRY(ctx, hipMemcpy2DAsync(base + fbytes * i, (size_t)width, img + image_stride * i, (size_t)pitch,
                                         (size_t)width, (size_t)height, kind, ctx->stream));
  }
  const int rc_levels = build_pyramid_levels(ctx, ctx->stream, slab, base, fbytes, offs, ws, hs, n_images, width, height, n_levels, rounding, out_frames);
  return rc_levels;
} SVOH_ABI_CATCH(ctx)

// levels 1.. of n_images frames whose level 0 is in place (queued on `stream`), and the frames' handles
static int build_pyramid_levels(svoh_ctx* ctx, hipStream_t stream, const std::shared_ptr<Slab>& slab, uint8_t* base, size_t fbytes, const size_t* offs, const int* ws,
                                const int* hs, int n_images, int width, int height, int n_levels, int rounding, svoh_frame_t* out_frames)
{
  int first_single = 1;   // levels from here on are made one launch each
  {
    // the first 2..4 steps as one launch when every one of them writes its whole level
    const int n_steps = n_levels - 1 < 4 ? n_levels - 1 : 4;
    PyramidArgs pa{};
    bool ok = n_steps >= 2;
    for (int k = 0; ok && k < n_steps; ++k) {
      bool all = false;
      if (step_rule(rounding, ws[k], hs[k], &all)) pa.sse_mask |= 1u << k;
      ok = all;
    }
    if (ok) {
      pa.base = base; pa.frame_stride = fbytes; pa.n_steps = n_steps;
      for (int k = 0; k <= n_steps; ++k) { pa.off[k] = offs[k]; pa.w[k] = ws[k]; pa.h[k] = hs[k]; }
      const dim3 grid((unsigned)((width + 63) / 64), (unsigned)((height + 63) / 64), (unsigned)n_images);
      hipLaunchKernelGGL(pyramid_fused_kernel, grid, dim3(256), 0, stream, pa);
      SVOH_HIP_TRY(ctx, hipGetLastError());
      first_single = n_steps + 1;
    }
  }
  for (int l = first_single; l < n_levels; ++l) {
    // zero-fill is not needed: every byte of a level the reference would write is written
    SVOH_HIP_TRY(ctx, launch_half_sample(stream, base + offs[l - 1], fbytes, ws[l - 1], hs[l - 1], ws[l - 1],
                                         base + offs[l], fbytes, ws[l], n_images, rounding));
  }
  for (int i = 0; i < n_images; ++i)
    out_frames[i] = register_frame(ctx, slab, base + fbytes * i, width, height, n_levels);
  return SVOH_OK;
}

static int build_pyramid_multi_on(svoh_ctx* ctx, hipStream_t stream, const uint8_t* const* imgs, int n_images, int width, int height, int pitch,
                                  int mem_space, int n_levels, int rounding, svoh_frame_t* out_frames)
{
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, imgs && out_frames, "NULL argument");
  SVOH_REQUIRE(ctx, n_images >= 1 && width > 0 && height > 0 && pitch >= width, "bad image geometry");
  SVOH_REQUIRE(ctx, n_levels >= 1 && n_levels <= SVOH_MAX_LEVELS, "n_levels out of range");
  SVOH_REQUIRE(ctx, rounding >= 0 && rounding <= 2, "bad rounding mode");
  SVOH_REQUIRE(ctx, (width >> (n_levels - 1)) > 0 && (height >> (n_levels - 1)) > 0, "too many levels");
  SVOH_REQUIRE(ctx, mem_space == SVOH_MEM_HOST || mem_space == SVOH_MEM_DEVICE || mem_space == SVOH_MEM_HOST_PINNED, "bad mem_space");
  SVOH_REQUIRE(ctx, mem_space != SVOH_MEM_HOST_PINNED || n_images <= 256, "at most 256 page-locked images per call");
  for (int i = 0; i < n_images; ++i) SVOH_REQUIRE(ctx, imgs[i] != nullptr, "NULL image");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  size_t offs[SVOH_MAX_LEVELS]; int ws[SVOH_MAX_LEVELS], hs[SVOH_MAX_LEVELS];
  const size_t fbytes = frame_layout(width, height, n_levels, offs, ws, hs);
  std::shared_ptr<Slab> slab;
  SVOH_HIP_TRY(ctx, make_slab(ctx, fbytes * (size_t)n_images, &slab));
  uint8_t* base = static_cast<uint8_t*>(slab->ptr);
  if (mem_space == SVOH_MEM_HOST_PINNED) {
    GatherArgs ga;
    ga.base = base; ga.frame_stride = fbytes; ga.width = width; ga.height = height; ga.pitch = pitch;
    for (int i = 0; i < n_images; ++i) ga.src[i] = imgs[i];
    for (int i = n_images; i < 256; ++i) ga.src[i] = nullptr;
    // enough loads in flight to fill the link: 16 workgroups of 256 x 16 bytes per image
    hipLaunchKernelGGL(gather_images_kernel, dim3(16, (unsigned)n_images), dim3(256), 0, stream, ga);
    SVOH_HIP_TRY(ctx, hipGetLastError());
  } else {
    const hipMemcpyKind kind = mem_space == SVOH_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    for (int i = 0; i < n_images; ++i)
      SVOH_HIP_TRY(ctx, hipMemcpy2DAsync(base + fbytes * i, (size_t)width, imgs[i], (size_t)pitch, (size_t)width, (size_t)height, kind, stream));
  }
  return build_pyramid_levels(ctx, stream, slab, base, fbytes, offs, ws, hs, n_images, width, height, n_levels, rounding, out_frames);
}

int svoh_build_pyramid_multi(svoh_ctx* ctx, const uint8_t* const* imgs, int n_images, int width, int height, int pitch,
                             int mem_space, int n_levels, int rounding, svoh_frame_t* out_frames)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  return build_pyramid_multi_on(ctx, ctx->stream, imgs, n_images, width, height, pitch, mem_space, n_levels, rounding, out_frames);
} SVOH_ABI_CATCH(ctx)

int svoh_build_pyramid_multi_prefetch(svoh_ctx* ctx, const uint8_t* const* imgs, int n_images, int width, int height, int pitch,
                                      int mem_space, int n_levels, int rounding, svoh_frame_t* out_frames)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->upload_stream) {
    SVOH_HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    SVOH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_upload, hipEventDisableTiming));
  }
  // The slab the frames go into may come from the pool of released frames, whose last readers were queued on the context's
  // own stream: that stream has to be idle now (returns at once when it is, which is when a caller should prefetch --
  // right behind a call that waited), and whatever it is given from here on cannot know these slabs.
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  const int rc = build_pyramid_multi_on(ctx, ctx->upload_stream, imgs, n_images, width, height, pitch, mem_space, n_levels, rounding, out_frames);
  if (rc != SVOH_OK) return rc;
  SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_upload, ctx->upload_stream));
  ctx->upload_pending = true;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_prefetch_fence(svoh_ctx* ctx)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!ctx->upload_pending) return SVOH_OK;
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_upload, 0));
  ctx->upload_pending = false;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_host_alloc(svoh_ctx* ctx, size_t bytes, void** out)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, out && bytes > 0, "bad arguments");
  *out = nullptr;
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_HIP_TRY(ctx, hipHostMalloc(out, bytes, hipHostMallocDefault));
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_host_free(svoh_ctx* ctx, void* p)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  if (!p) return SVOH_OK;
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  SVOH_HIP_TRY(ctx, hipHostFree(p));   // (waits for work that may still read the block)
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_features_upload(svoh_ctx* ctx, int n_sets, const int32_t* n, const double* const* px, const double* const* f,
                         const double* const* grad, const int32_t* const* level, svoh_features_t* out)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, n_sets >= 0 && (n_sets == 0 || (n && px && f && grad && level && out)), "bad arguments");
  if (n_sets == 0) return SVOH_OK;
  // layout of the call's block: set after set, each [px 16n | f 24n | grad 16n | level 4n | the same four in tile order | perm 4n],
  // every array 64-byte aligned
  auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
  size_t total = 0;
  for (int k = 0; k < n_sets; ++k) {
    SVOH_REQUIRE(ctx, n[k] >= 0 && (n[k] == 0 || (px[k] && f[k] && grad[k] && level[k])), "a set's n is negative or one of its arrays is NULL");
    const size_t m = (size_t)n[k];
    total += 2 * (al(16 * m) + al(24 * m) + al(16 * m) + al(4 * m)) + al(4 * m);
    out[k] = 0;
  }
  if (total == 0) total = 64;
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the staging block may still be the source of the last upload's copy
  if (ctx->ev_features) SVOH_HIP_TRY(ctx, hipEventSynchronize(ctx->ev_features));
  else SVOH_HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_features, hipEventDisableTiming));
  SVOH_HIP_TRY(ctx, ctx->h_features.reserve(total));
  auto block = std::make_shared<svoh::FeatureBlock>();
  block->ptr = ctx->feature_pool->take(total, &block->alloc);
  if (!block->ptr) {
    const size_t want = (total + 0xffff) & ~(size_t)0xffff;   // 64 KB steps: blocks of nearly equal size are interchangeable
    hipError_t e = hipMalloc(&block->ptr, want);
    if (e != hipSuccess) { block->ptr = nullptr; SVOH_HIP_TRY(ctx, e); }
    block->alloc = want;
  }
  block->pool = ctx->feature_pool;
  uint8_t* h = static_cast<uint8_t*>(ctx->h_features.ptr);
  uint8_t* d = static_cast<uint8_t*>(block->ptr);
  size_t off = 0;
  std::vector<svoh::FeatureSet> sets((size_t)n_sets);
  for (int k = 0; k < n_sets; ++k) {
    const size_t m = (size_t)n[k];
    svoh::FeatureSet& fs = sets[(size_t)k];
    fs.block = block; fs.n = n[k];
    fs.px = reinterpret_cast<const double*>(d + off); if (m) memcpy(h + off, px[k], 16 * m); off += al(16 * m);
    fs.f = reinterpret_cast<const double*>(d + off); if (m) memcpy(h + off, f[k], 24 * m); off += al(24 * m);
    fs.grad = reinterpret_cast<const double*>(d + off); if (m) memcpy(h + off, grad[k], 16 * m); off += al(16 * m);
    fs.level = reinterpret_cast<const int32_t*>(d + off); if (m) memcpy(h + off, level[k], 4 * m); off += al(4 * m);
    // tile order: a stable sort of the features by (tile row, tile column) of their pixel
    std::vector<int32_t> perm(m);
    std::vector<uint32_t> key(m);
    for (size_t i = 0; i < m; ++i) {
      perm[i] = (int32_t)i;
      const double x = px[k][2 * i], y = px[k][2 * i + 1];
      const uint32_t tx = (x >= 0.0 && x < 1e9) ? (uint32_t)((int)x >> svoh::kBinShiftX) : 0u, ty = (y >= 0.0 && y < 1e9) ? (uint32_t)((int)y >> svoh::kBinShiftY) : 0u;
      key[i] = (ty << 12) | (tx & 0xfffu);
    }
    std::stable_sort(perm.begin(), perm.end(), [&](int32_t a, int32_t b) { return key[(size_t)a] < key[(size_t)b]; });
    double* spx = reinterpret_cast<double*>(h + off); fs.spx = reinterpret_cast<const double*>(d + off); off += al(16 * m);
    double* sf = reinterpret_cast<double*>(h + off); fs.sf = reinterpret_cast<const double*>(d + off); off += al(24 * m);
    double* sgrad = reinterpret_cast<double*>(h + off); fs.sgrad = reinterpret_cast<const double*>(d + off); off += al(16 * m);
    int32_t* slevel = reinterpret_cast<int32_t*>(h + off); fs.slevel = reinterpret_cast<const int32_t*>(d + off); off += al(4 * m);
    int32_t* sperm = reinterpret_cast<int32_t*>(h + off); fs.perm = reinterpret_cast<const int32_t*>(d + off); off += al(4 * m);
    for (size_t q = 0; q < m; ++q) {
      const size_t i = (size_t)perm[q];
      spx[2 * q] = px[k][2 * i]; spx[2 * q + 1] = px[k][2 * i + 1];
      sf[3 * q] = f[k][3 * i]; sf[3 * q + 1] = f[k][3 * i + 1]; sf[3 * q + 2] = f[k][3 * i + 2];
      sgrad[2 * q] = grad[k][2 * i]; sgrad[2 * q + 1] = grad[k][2 * i + 1];
      slevel[q] = level[k][i]; sperm[q] = perm[q];
    }
  }
  SVOH_HIP_TRY(ctx, svoh_copy_to_device(ctx, d, h, off ? off : 64));
  SVOH_HIP_TRY(ctx, hipEventRecord(ctx->ev_features, ctx->stream));
  for (int k = 0; k < n_sets; ++k) {
    const uint64_t id = ctx->next_features_id++;
    ctx->feature_sets.emplace(id, std::move(sets[(size_t)k]));
    out[k] = id;
  }
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_features_release(svoh_ctx* ctx, svoh_features_t features)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  auto it = ctx->feature_sets.find(features);
  if (it == ctx->feature_sets.end()) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown feature-set handle %llu", (unsigned long long)features);
  // (no wait: batches queued before this call may still read the block; whoever gets it from the pool writes it on the
  // context's stream, behind them -- or hipFree waits for the device by itself)
  ctx->feature_sets.erase(it);
  ++ctx->handle_generation;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_build_pyramid(svoh_ctx* ctx, const uint8_t* img, int width, int height, int pitch, int mem_space,
                       int n_levels, int rounding, uint8_t* const* host_levels_out, svoh_frame_t* out_frame)
try {
  int rc = svoh_build_pyramid_batch(ctx, img, (size_t)pitch * height, 1, width, height, pitch, mem_space, n_levels,
                                    rounding, out_frame);
  if (rc != SVOH_OK) return rc;
  if (host_levels_out) {
    const Frame* f = find_frame(ctx, *out_frame);
    for (int l = 0; l < n_levels; ++l) {
      if (!host_levels_out[l]) continue;
      SVOH_HIP_TRY(ctx, hipMemcpyAsync(host_levels_out[l], f->lv[l].data, (size_t)f->lv[l].w * f->lv[l].h,
                                       hipMemcpyDeviceToHost, ctx->stream));
    }
    SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_download_level(svoh_ctx* ctx, svoh_frame_t frame, int level, uint8_t* out, int* out_width, int* out_height)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  const Frame* f = find_frame(ctx, frame);
  if (!f) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown frame handle %llu", (unsigned long long)frame);
  SVOH_REQUIRE(ctx, level >= 0 && level < f->n_levels, "level out of range");
  if (out_width) *out_width = f->lv[level].w;
  if (out_height) *out_height = f->lv[level].h;
  if (out) {
    SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
    SVOH_HIP_TRY(ctx, hipMemcpyAsync(out, f->lv[level].data, (size_t)f->lv[level].w * f->lv[level].h,
                                     hipMemcpyDeviceToHost, ctx->stream));
    SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_frame_info(svoh_ctx* ctx, svoh_frame_t frame, int* n_levels, int* width0, int* height0)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  const Frame* f = find_frame(ctx, frame);
  if (!f) return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown frame handle %llu", (unsigned long long)frame);
  if (n_levels) *n_levels = f->n_levels;
  if (width0) *width0 = f->lv[0].w;
  if (height0) *height0 = f->lv[0].h;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_context_stats(svoh_ctx* ctx, svoh_context_stats_t* out)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, out != nullptr, "NULL argument");
  out->live_frames = (int64_t)ctx->frames.size();
  std::unordered_map<const void*, size_t> slabs;   // frames of a batch share one slab
  for (const auto& kv : ctx->frames)
    if (kv.second.slab) slabs[kv.second.slab.get()] = kv.second.slab->bytes;
  size_t fb = 0;
  for (const auto& kv : slabs) fb += kv.second;
  out->frame_bytes = (int64_t)fb;
  const svoh::DevBuffer* bufs[] = { &ctx->d_desc, &ctx->d_results, &ctx->d_feat, &ctx->d_eval, &ctx->d_xchg, &ctx->d_split,
                                    &ctx->d_counters, &ctx->d_unit_counts, &ctx->d_scratch0, &ctx->d_scratch1, &ctx->d_scratch2, &ctx->d_seed_bin, &ctx->d_match_seeds, &ctx->d_match_direct, &ctx->d_cand, &ctx->d_cand_multi };
  size_t wb = 0;
  for (const svoh::DevBuffer* b : bufs) wb += b->cap;
  out->workspace_bytes = (int64_t)wb;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_camera_maths(svoh_ctx* ctx, const svoh_camera* cam, int n, const double* xyz, double* px, double* J, double* f_back)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  SVOH_REQUIRE(ctx, cam && xyz && px && n >= 1 && n <= (1 << 20), "bad arguments");
  SVOH_REQUIRE(ctx, cam->distortion == SVOH_DISTORTION_NONE || cam->distortion == SVOH_DISTORTION_RADTAN, "unsupported distortion model");
  SVOH_HIP_TRY(ctx, hipSetDevice(ctx->device));
  // [xyz 3n | px 2n | J 6n | f 3n] doubles on the device
  const size_t nd = (size_t)n;
  SVOH_HIP_TRY(ctx, ctx->d_scratch2.reserve(14 * nd * sizeof(double)));
  double* d = static_cast<double*>(ctx->d_scratch2.ptr);
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(d, xyz, 3 * nd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(camera_maths_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, *cam, n, d, d + 3 * nd, d + 5 * nd,
                     d + 11 * nd);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(ctx, SVOH_ERR_HIP, "camera_maths launch failed: %s", hipGetErrorString(e));
  SVOH_HIP_TRY(ctx, hipMemcpyAsync(px, d + 3 * nd, 2 * nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (J) SVOH_HIP_TRY(ctx, hipMemcpyAsync(J, d + 5 * nd, 6 * nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (f_back) SVOH_HIP_TRY(ctx, hipMemcpyAsync(f_back, d + 11 * nd, 3 * nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SVOH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

int svoh_release_frame(svoh_ctx* ctx, svoh_frame_t frame)
try {
  if (!ctx) return set_error(nullptr, SVOH_ERR_INVALID_ARGUMENT, "ctx is NULL");
  auto it = ctx->frames.find(frame);
  if (it == ctx->frames.end())
    return set_error(ctx, SVOH_ERR_BAD_HANDLE, "unknown frame handle %llu", (unsigned long long)frame);
  // No wait here: kernels queued before this call may still read the frame.  Its slab either goes to the context's
  // pool -- the next frame is then written on the context's stream, behind those kernels -- or is freed by hipFree,
  // which waits for the device by itself.
  ctx->frames.erase(it);
  ++ctx->handle_generation;
  return SVOH_OK;
} SVOH_ABI_CATCH(ctx)

}  // extern "C"
