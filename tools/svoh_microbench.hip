// svoh_microbench -- two small calibrations that the roofline figures in bench.py / DESIGN.md lean on:
//   (1) the fp64 vector FMA rate of the device (the guide quotes no fp64 figure): a loop of independent
//       v_fma_f64 in every lane of every SIMD;
//   (2) what rocprofv3's FETCH_SIZE reports for the access widths this library uses (the guide calibrates only
//       16-B-per-lane streaming reads: x2 on gfx950): kernels that read a KNOWN number of bytes once from a
//       buffer far larger than the Infinity Cache, one per access pattern.  Run under
//       `rocprofv3 --pmc FETCH_SIZE` the per-kernel counter divided by the bytes printed here is the factor.
// Patterns: 16 B / 8 B / 4 B / 1 B per lane, consecutive lanes consecutive addresses (streaming); 16 B per lane
// by LDS-DMA (global_load_lds_dwordx4); 8 B per lane from a different 128-B line per lane (the seed kernel's
// row pieces); 12 unaligned bytes per lane from a different line per lane (the alignment's footprint rows).
// Usage: svoh_microbench [bytes_in_MiB=2048]      prints one JSON line.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void fma64_kernel(double* out, int iters, double seed)
{
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const double m = 1.0000001, c = 1e-9;
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_fma(a0, m, c); a1 = __builtin_fma(a1, m, c); a2 = __builtin_fma(a2, m, c); a3 = __builtin_fma(a3, m, c);
    a4 = __builtin_fma(a4, m, c); a5 = __builtin_fma(a5, m, c); a6 = __builtin_fma(a6, m, c); a7 = __builtin_fma(a7, m, c);
  }
  const double s = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (s == 12345.678) out[0] = s;   // never true: keeps the loop alive
}

// issue rate of v_cvt_f64_u32 against the two-instruction exact alternative (2^52 trick: or the integer into the
// mantissa of 2^52, subtract 2^52), 8 independent chains each
__global__ __launch_bounds__(256) void cvt64_kernel(double* out, int iters, unsigned seed)
{
  unsigned x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double s = 0.0;
  for (int i = 0; i < iters; ++i) {
    const double d0 = (double)(x0 & 255u), d1 = (double)(x1 & 255u), d2 = (double)(x2 & 255u), d3 = (double)(x3 & 255u);
    const double d4 = (double)(x4 & 255u), d5 = (double)(x5 & 255u), d6 = (double)(x6 & 255u), d7 = (double)(x7 & 255u);
    s += ((d0 + d1) + (d2 + d3)) + ((d4 + d5) + (d6 + d7));
    x0 += 3; x1 += 5; x2 += 7; x3 += 11; x4 += 13; x5 += 17; x6 += 19; x7 += 23;
  }
  if (s == 12345.678) out[0] = s;
}
__device__ __forceinline__ double u8_to_f64_magic(unsigned v)
{
  return __hiloint2double(0x43300000, (int)v) - 4503599627370496.0;   // exact for v < 2^32
}
__global__ __launch_bounds__(256) void cvt64_magic_kernel(double* out, int iters, unsigned seed)
{
  unsigned x0 = seed + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double s = 0.0;
  for (int i = 0; i < iters; ++i) {
    const double d0 = u8_to_f64_magic(x0 & 255u), d1 = u8_to_f64_magic(x1 & 255u), d2 = u8_to_f64_magic(x2 & 255u), d3 = u8_to_f64_magic(x3 & 255u);
    const double d4 = u8_to_f64_magic(x4 & 255u), d5 = u8_to_f64_magic(x5 & 255u), d6 = u8_to_f64_magic(x6 & 255u), d7 = u8_to_f64_magic(x7 & 255u);
    s += ((d0 + d1) + (d2 + d3)) + ((d4 + d5) + (d6 + d7));
    x0 += 3; x1 += 5; x2 += 7; x3 += 11; x4 += 13; x5 += 17; x6 += 19; x7 += 23;
  }
  if (s == 12345.678) out[0] = s;
}

template <typename T>
__global__ __launch_bounds__(256) void stream_read_kernel(const T* __restrict__ src, size_t n, unsigned* sink)
{
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const T v = src[i];
    if constexpr (sizeof(T) >= 4) {     // every 32-bit word is used, so the load keeps its full width
      unsigned w[sizeof(T) / 4];
      __builtin_memcpy(w, &v, sizeof(T));
      for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k] + k;
    } else {
      acc += (unsigned)v;
    }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// 16 B per lane straight into LDS (no VGPR): the form the alignment kernel uses for workspace rows / level images
__global__ __launch_bounds__(256) void lds_dma_read_kernel(const uint4* __restrict__ src, size_t n, unsigned* sink)
{
  __shared__ uint4 buf[256];
  unsigned acc = 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    typedef const __attribute__((address_space(1))) void* gptr;
    typedef __attribute__((address_space(3))) void* lptr;
    // per-lane global address, wave-uniform LDS base: lane k's 16 bytes land at base + 16 k
    __builtin_amdgcn_global_load_lds((gptr)(src + i), (lptr)(&buf[threadIdx.x & ~63u]), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    acc += buf[threadIdx.x].x;
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

// every lane reads BYTES bytes at byte offset `off` of its own 128-B line (lines visited once each)
template <int BYTES>
__global__ __launch_bounds__(256) void line_piece_kernel(const unsigned char* __restrict__ src, size_t n_lines, int off, unsigned* sink)
{
  unsigned acc = 0;
  for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < n_lines; l += (size_t)gridDim.x * blockDim.x) {
    // scatter the lines of neighbouring lanes: lane k takes line (l * 97) mod n_lines' neighbourhood
    const unsigned char* p = src + ((l * 40503ull) % n_lines) * 128 + off;
    if (BYTES == 8) {
      uint2 v; __builtin_memcpy(&v, p, 8); acc += v.x ^ v.y;
    } else if (BYTES == 12) {
      uint2 v; unsigned w; __builtin_memcpy(&v, p, 8); __builtin_memcpy(&w, p + 8, 4); acc += v.x ^ v.y ^ w;
    } else {
      acc += p[0];
    }
  }
  if (acc == 0xFFFFFFFFu) sink[0] = acc;
}

template <typename F>
static int time_it(const char* name, double units, const char* unit_name, F launch, bool last = false)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch();   // warm-up
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 3; ++r) {
    CHECK(hipEventRecord(a, 0));
    launch();
    CHECK(hipEventRecord(b, 0));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  CHECK(hipGetLastError());
  printf("\"%s\": {\"ms\": %.4f, \"%s\": %.6g, \"rate_per_s\": %.6g}%s", name, best, unit_name, units, units / (best * 1e-3), last ? "" : ", ");
  return 0;
}

int main(int argc, char** argv)
{
  const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 2048;
  const size_t bytes = mib << 20;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned char* buf; unsigned* sink; double* dout;
  CHECK(hipMalloc(&buf, bytes + 256)); CHECK(hipMalloc(&sink, 64)); CHECK(hipMalloc(&dout, 64));
  CHECK(hipMemset(buf, 1, bytes + 256));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"buffer_bytes\": %zu, ", prop.gcnArchName, cus, prop.clockRate / 1000, bytes);
  const int grid = cus * 8;
  // (1) fp64 FMA peak: cus*8 blocks * 256 lanes * 8 accumulators * iters FMAs, 2 flop each
  const int iters = 1 << 15;
  if (time_it("fma64", 2.0 * 8.0 * iters * 256.0 * grid, "flop", [&] { hipLaunchKernelGGL(fma64_kernel, dim3(grid), dim3(256), 0, 0, dout, iters, 0.5); })) return 1;
  if (time_it("cvt_f64_u32_x8_plus_adds", 8.0 * (1 << 13) * 256.0 * grid, "conversions", [&] { hipLaunchKernelGGL(cvt64_kernel, dim3(grid), dim3(256), 0, 0, dout, 1 << 13, 1u); })) return 1;
  if (time_it("magic_2p52_x8_plus_adds", 8.0 * (1 << 13) * 256.0 * grid, "conversions", [&] { hipLaunchKernelGGL(cvt64_magic_kernel, dim3(grid), dim3(256), 0, 0, dout, 1 << 13, 1u); })) return 1;
  // (2) known-bytes reads
  if (time_it("stream16", (double)bytes, "bytes", [&] { hipLaunchKernelGGL(stream_read_kernel<uint4>, dim3(grid * 4), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), bytes / 16, sink); })) return 1;
  if (time_it("stream8", (double)bytes, "bytes", [&] { hipLaunchKernelGGL(stream_read_kernel<uint2>, dim3(grid * 4), dim3(256), 0, 0, reinterpret_cast<const uint2*>(buf), bytes / 8, sink); })) return 1;
  if (time_it("stream4", (double)bytes / 2, "bytes", [&] { hipLaunchKernelGGL(stream_read_kernel<unsigned>, dim3(grid * 4), dim3(256), 0, 0, reinterpret_cast<const unsigned*>(buf), bytes / 8, sink); })) return 1;
  if (time_it("stream1", (double)bytes / 8, "bytes", [&] { hipLaunchKernelGGL(stream_read_kernel<unsigned char>, dim3(grid * 4), dim3(256), 0, 0, buf, bytes / 8, sink); })) return 1;
  if (time_it("lds_dma16", (double)bytes, "bytes", [&] { hipLaunchKernelGGL(lds_dma_read_kernel, dim3(grid * 4), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), bytes / 16, sink); })) return 1;
  const size_t n_lines = bytes / 128;
  if (time_it("line_piece8", (double)n_lines * 8, "bytes", [&] { hipLaunchKernelGGL(line_piece_kernel<8>, dim3(grid * 4), dim3(256), 0, 0, buf, n_lines, 40, sink); })) return 1;
  if (time_it("line_piece12_unaligned", (double)n_lines * 12, "bytes", [&] { hipLaunchKernelGGL(line_piece_kernel<12>, dim3(grid * 4), dim3(256), 0, 0, buf, n_lines, 57, sink); })) return 1;
  if (time_it("line_piece1", (double)n_lines, "bytes", [&] { hipLaunchKernelGGL(line_piece_kernel<1>, dim3(grid * 4), dim3(256), 0, 0, buf, n_lines, 3, sink); }, true)) return 1;
  printf("}\n");
  CHECK(hipFree(buf)); CHECK(hipFree(sink)); CHECK(hipFree(dout));
  return 0;
}
