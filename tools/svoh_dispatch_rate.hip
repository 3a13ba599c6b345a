// svoh_dispatch_rate -- how many small dispatches per second ONE MI355X takes from T host threads, each with a stream of its
// own: a tiny kernel, a 64 KB copy kernel from pinned memory, a 64 KB hipMemcpyAsync, with a hipStreamSynchronize every
// `chain` dispatches (the shape of the per-frame chain: a handful of dependent launches, then a wait).  The lock-step
// front end of many camera streams saturates near 25 - 28 k frames/s whatever the number of groups and threads; this
// tool asks whether the machine's dispatch path is what it runs into.
//   svoh_dispatch_rate [threads=4] [chain=6] [seconds=1.0]
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void tiny_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void copy_kernel(uint4* dst, const uint4* src, size_t n16) { for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i]; }

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv)
{
  const int T = argc > 1 ? atoi(argv[1]) : 4, chain = argc > 2 ? atoi(argv[2]) : 6;
  const double secs = argc > 3 ? atof(argv[3]) : 1.0;
  for (int mode = 0; mode < 4; ++mode) {
    std::atomic<long> total(0);
    std::atomic<int> go(0);
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t)
      th.emplace_back([&, t] {
        hipSetDevice(0);
        hipStream_t s;
        hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        int* d; hipMalloc(&d, 1 << 20);
        void* h; hipHostMalloc(&h, 1 << 20, hipHostMallocDefault);
        hipMemset(d, 0, 1 << 20);
        go.fetch_add(1);
        while (go.load() < T) {}
        long n = 0;
        const double t0 = now();
        while (now() - t0 < secs) {
          for (int k = 0; k < chain; ++k) {
            if (mode == 0) hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, s, d);
            else if (mode == 1) hipLaunchKernelGGL(copy_kernel, dim3(16), dim3(256), 0, s, (uint4*)d, (const uint4*)h, (size_t)4096);
            else if (mode == 2) hipMemcpyAsync(d, h, 65536, hipMemcpyHostToDevice, s);
            else { if (k % 3 == 0) hipLaunchKernelGGL(copy_kernel, dim3(16), dim3(256), 0, s, (uint4*)d, (const uint4*)h, (size_t)4096); else if (k % 3 == 1) hipLaunchKernelGGL(tiny_kernel, dim3(8), dim3(256), 0, s, d); else hipMemcpyAsync(h, d, 65536, hipMemcpyDeviceToHost, s); }
          }
          hipStreamSynchronize(s);
          n += chain;
        }
        total.fetch_add(n);
        hipStreamDestroy(s); hipFree(d); hipHostFree(h);
      });
    for (auto& x : th) x.join();
    const char* names[] = { "tiny kernels", "64 KB copy kernels (pinned -> device)", "64 KB hipMemcpyAsync (pinned -> device)", "mixed: copy kernel in, kernel, hipMemcpyAsync out" };
    printf("%d threads, a wait every %d dispatches, %s: %.0f dispatches/s in total (%.1f us per dispatch and thread)\n", T, chain, names[mode], total.load() / secs,
           1e6 * secs * T / total.load());
  }
  return 0;
}
