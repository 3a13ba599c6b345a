// What a blocking call costs around its kernel on this machine: staging copy, launch, copy back, wake-up -- and what
// the same call costs when the kernel reads its (small) inputs from pinned host memory and writes its results there.
// Diagnostic tool (per-frame chain); prints one JSON object.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// reads n_in doubles (every lane strides through them), spins `work` dependent FMAs, writes n_out doubles
__global__ void work_kernel(const double* in, double* out, int n_in, int n_out, int work)
{
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_in; i += blockDim.x) acc += in[i];
  for (int k = 0; k < work; ++k) acc = acc * 1.0000001 + 1e-9;
  for (int i = threadIdx.x; i < n_out; i += blockDim.x) out[i] = acc + i;
}

__global__ void zero_kernel(double* p, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.0; }

// copies n doubles from device memory into pinned host memory and then raises a flag there: what a D2H copy + the
// runtime's completion signal do, as one small kernel whose end the host sees by polling a word
__global__ void copy_signal_kernel(const double* src, double* dst_host, int n, unsigned* flag_host, unsigned seq)
{
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst_host[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// the work kernel writing its results into pinned host memory and raising the flag itself
__global__ void work_signal_kernel(const double* in, double* out_host, int n_in, int n_out, int work, unsigned* flag_host, unsigned seq)
{
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_in; i += blockDim.x) acc += in[i];
  for (int k = 0; k < work; ++k) acc = acc * 1.0000001 + 1e-9;
  for (int i = threadIdx.x; i < n_out; i += blockDim.x) out_host[i] = acc + i;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// results back to pinned host memory by a kernel's own stores (several workgroups), the wait left to the runtime
__global__ void copy_kernel(const double* src, double* dst_host, int n)
{
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst_host[i] = src[i];
  __threadfence_system();
}

// raises the flag only: queued behind an ordinary D2H copy, it tells the host that the copy has landed
__global__ void signal_kernel(unsigned* flag_host, unsigned seq)
{
  __hip_atomic_store(flag_host, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  const int n_in = getenv("SVOH_OVERHEAD_N_IN") ? atoi(getenv("SVOH_OVERHEAD_N_IN")) : 2048;   // doubles uploaded (16 KB: one frame's features; a seed batch: 16384)
  const int reps = 300;
  const int n_out = getenv("SVOH_OVERHEAD_N_OUT") ? atoi(getenv("SVOH_OVERHEAD_N_OUT")) : 512;   // doubles copied back (a matcher batch: 16384)
  // SVOH_OVERHEAD_SCHEDULE=spin|yield|block: the runtime's own wait policy (hipSetDeviceFlags) for the *_sync variants
  if (const char* sch = getenv("SVOH_OVERHEAD_SCHEDULE")) {
    const unsigned f = !strcmp(sch, "spin") ? hipDeviceScheduleSpin : !strcmp(sch, "yield") ? hipDeviceScheduleYield : hipDeviceScheduleBlockingSync;
    CK(hipSetDeviceFlags(f));
  }
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  double *h_in, *h_out, *h_in_nc, *h_out_nc, *d_in, *d_out;
  CK(hipHostMalloc(&h_in, n_in * 8, hipHostMallocDefault));
  CK(hipHostMalloc(&h_out, n_out * 8, hipHostMallocDefault));
  CK(hipHostMalloc(&h_in_nc, n_in * 8, hipHostMallocNonCoherent));
  CK(hipHostMalloc(&h_out_nc, n_out * 8, hipHostMallocNonCoherent));
  CK(hipMalloc(&d_in, n_in * 8));
  CK(hipMalloc(&d_out, n_out * 8));
  unsigned* h_flag;
  CK(hipHostMalloc(&h_flag, 64, hipHostMallocDefault));
  *h_flag = 0;
  unsigned seq = 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < n_in; ++i) h_in[i] = h_in_nc[i] = 1.0;
  printf("{");
  const char* names[] = { "kernel_sync", "h2d_kernel_sync", "h2d_kernel_d2h_sync", "h2d_memset_events_kernel_d2h_sync",
                          "zero_copy_coherent", "zero_copy_noncoherent", "zero_copy_in_noncoherent_out_device_d2h", "h2d_events_kernel_d2h_sync", "h2d_memset_kernel_d2h_sync", "h2d_fillkernel_kernel_d2h_sync",
                          "h2d_kernel_copysignal_poll", "h2d_worksignal_poll", "h2d_kernel_d2h_signal_poll", "h2d_kernel_copykernel_sync", "copyin_kernel_d2h_sync" };
  for (int work : { 0, 20000 }) {
    for (int v = 0; v < 15; ++v) {
      std::vector<double> t;
      for (int r = 0; r < reps + 20; ++r) {
        h_in[0] = h_in_nc[0] = (double)r;
        const double t0 = now_us();
        switch (v) {
          case 0: work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work); break;
          case 1: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work); break;
          case 2: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work);
                  CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); break;
          case 3: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); CK(hipMemsetAsync(d_out, 0, n_out * 8, s)); CK(hipEventRecord(e0, s));
                  work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work); CK(hipEventRecord(e1, s));
                  CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); break;
          case 4: work_kernel<<<1, 256, 0, s>>>(h_in, h_out, n_in, n_out, work); break;
          case 5: work_kernel<<<1, 256, 0, s>>>(h_in_nc, h_out_nc, n_in, n_out, work); break;
          case 7: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); CK(hipEventRecord(e0, s));
                  work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work); CK(hipEventRecord(e1, s));
                  CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); break;
          case 8: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); CK(hipMemsetAsync(d_out, 0, n_out * 8, s));
                  work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work);
                  CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); break;
          case 9: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); zero_kernel<<<1, 256, 0, s>>>(d_out, n_out);
                  work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work);
                  CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); break;
          case 10: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work);
                   copy_signal_kernel<<<1, 256, 0, s>>>(d_out, h_out, n_out, h_flag, ++seq); break;
          case 11: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s));
                   work_signal_kernel<<<1, 256, 0, s>>>(d_in, h_out, n_in, n_out, work, h_flag, ++seq); break;
          case 12: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work);
                   CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); signal_kernel<<<1, 1, 0, s>>>(h_flag, ++seq); break;
          case 13: CK(hipMemcpyAsync(d_in, h_in, n_in * 8, hipMemcpyHostToDevice, s)); work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work);
                   copy_kernel<<<(n_out + 1023) / 1024 > 32 ? 32 : (n_out + 1023) / 1024, 256, 0, s>>>(d_out, h_out, n_out); break;
          case 14: copy_kernel<<<(n_in + 1023) / 1024 > 32 ? 32 : (n_in + 1023) / 1024, 256, 0, s>>>(h_in, d_in, n_in);   // the upload by a kernel reading pinned host memory
                   work_kernel<<<1, 256, 0, s>>>(d_in, d_out, n_in, n_out, work);
                   CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); break;
          case 6: work_kernel<<<1, 256, 0, s>>>(h_in_nc, d_out, n_in, n_out, work); CK(hipMemcpyAsync(h_out, d_out, n_out * 8, hipMemcpyDeviceToHost, s)); break;
        }
        if (v >= 10 && v != 13 && v != 14) {   // the host polls the flag word (bounded: 20 ms, then the ordinary wait)
          const double dl = now_us() + 20000.0;
          while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != seq && now_us() < dl) __builtin_ia32_pause();
          if (h_out[n_out - 1] == -12345.0) return 2;   // keeps the results' read behind the flag's
        } else {
          CK(hipStreamSynchronize(s));
        }
        const double t1 = now_us();
        // SVOH_OVERHEAD_NO_DRAIN=1: the polled variants never call hipStreamSynchronize -- what the runtime's un-reaped
        // commands then cost the NEXT repetition's calls shows in that repetition's time
        if (v >= 10 && v != 13 && v != 14 && !getenv("SVOH_OVERHEAD_NO_DRAIN")) CK(hipStreamSynchronize(s));     // outside the clock: the stream is drained before the next repetition
        if (r >= 20) t.push_back(t1 - t0);
      }
      std::sort(t.begin(), t.end());
      printf("%s\"%s_work%d_us\": {\"median\": %.1f, \"p10\": %.1f}", (work == 0 && v == 0) ? "" : ", ", names[v], work, t[t.size() / 2], t[t.size() / 10]);
    }
  }
  printf("}\n");
  return 0;
}
