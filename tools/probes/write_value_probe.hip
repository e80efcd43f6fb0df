// Probe (not part of the build): how long after its predecessor on the stream does a word written by hipStreamWriteValue64 -- or by
// a one-thread kernel -- reach a kernel that is already running and polls it?  (The gated form of the multi-rank Euler step,
// tools/probes/gated_form.patch, waits that way for its ghost rows.)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/write_value_probe.hip -o /tmp/wvp2 && /tmp/wvp2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <thread>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void poll(const uint64_t *flag, uint64_t want, long long *stamp) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < want) {
      __builtin_amdgcn_s_sleep(4);
      if (wall_clock64() - t0 > 100000000ll) { stamp[2] = 1; break; }  // 1 s: give up
    }
    stamp[1] = wall_clock64();
  }
}
__global__ void mark(long long *stamp) { if (threadIdx.x == 0) stamp[0] = wall_clock64(); }
__global__ void set_word(uint64_t *flag, uint64_t v) { if (threadIdx.x == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

int main() {
  CK(hipSetDevice(0));
  int rate = 0;
  CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0));
  const double us = rate / 1000.0;
  uint64_t *plain = nullptr, *sig = nullptr;
  CK(hipMalloc((void **)&plain, 8));
  CK(hipMemset(plain, 0, 8));
  CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
  *sig = 0;
  long long *stamp = nullptr;
  CK(hipHostMalloc((void **)&stamp, 32, hipHostMallocDefault));
  hipStream_t st, cs;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  uint64_t n = 0;
  for (int mem = 0; mem < 2; ++mem) {
    uint64_t *flag = mem ? sig : plain;
    for (int how = 0; how < 2; ++how) {
      double sum = 0, worst = 0;
      int    gave_up = 0;
      const int K = 20;
      for (int i = 0; i < K; ++i) {
        ++n;
        stamp[0] = stamp[1] = stamp[2] = 0;
        poll<<<1, 64, 0, st>>>(flag, n, stamp);
        std::this_thread::sleep_for(std::chrono::microseconds(300));  // the polling kernel is running by now
        mark<<<1, 64, 0, cs>>>(stamp);                                // "the transfer": its end is the reference time
        if (how == 0) CK(hipStreamWriteValue64(cs, flag, n, 0));
        else set_word<<<1, 64, 0, cs>>>(flag, n);
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(cs));
        const double d = (stamp[1] - stamp[0]) / us;
        sum += d;
        if (d > worst) worst = d;
        gave_up += (int)stamp[2];
      }
      printf("%s memory, %s: word seen by the polling kernel %.2f us after the preceding kernel on that stream (mean of %d, worst %.2f, gave up %d)\n",
             mem ? "signal" : "plain device", how == 0 ? "hipStreamWriteValue64" : "one-thread kernel", sum / K, K, worst, gave_up);
    }
  }
  return 0;
}
