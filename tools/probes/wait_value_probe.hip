// Probe (not part of the build): can a stream on this device wait for a value a RUNNING kernel stores (hipStreamWaitValue64 on
// signal memory), and how long after the store does the waiting stream's next kernel start?  The multi-rank Euler step would use it
// to start the ghost transfer of step n + 1 as soon as step n's ghost-adjacent tiles have stored their send rows, while the same
// launch goes on with the interior tiles.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/wait_value_probe.hip -o /tmp/wait_value_probe && /tmp/wait_value_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void work(uint64_t *signal, uint64_t epoch, long long head_ticks, long long tail_ticks, long long *stamp) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < head_ticks) __builtin_amdgcn_s_sleep(8);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    __threadfence_system();
    stamp[0] = wall_clock64();
    __hip_atomic_store(signal, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  while (wall_clock64() - t0 < head_ticks + tail_ticks) __builtin_amdgcn_s_sleep(8);
}
__global__ void after(long long *stamp) { if (threadIdx.x == 0) stamp[1] = wall_clock64(); }

int main() {
  int can = 0, dev = 0;
  CK(hipSetDevice(dev));
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev));
  int rate = 0;
  CK(hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, dev));  // kHz
  printf("hipDeviceAttributeCanUseStreamWaitValue = %d, wall clock %d kHz\n", can, rate);
  if (!can) return 0;
  uint64_t *signal = nullptr;
  CK(hipExtMallocWithFlags((void **)&signal, 8, hipMallocSignalMemory));
  *signal = 0;  // signal memory is host-visible
  long long *stamp = nullptr;
  CK(hipHostMalloc((void **)&stamp, 16, hipHostMallocDefault));
  hipStream_t st, cs;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
  hipEvent_t evT, e0, e1;
  CK(hipEventCreateWithFlags(&evT, hipEventDisableTiming));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // a host watchdog: whatever happens, the waiting stream is released after 20 s
  std::atomic<bool> done{false};
  std::thread dog([&] {
    for (int i = 0; i < 200 && !done; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    if (!done) { *signal = ~0ull; printf("WATCHDOG released the signal\n"); }
  });
  const double us = rate / 1000.0;  // ticks per microsecond
  uint64_t epoch = 0;
  for (int mode = 0; mode < 3; ++mode) {
    // mode 0: the kernels alone; 1: + wait-value, kernel on the other stream, event back (signal at 30 % of the kernel);
    // 2: the same with the signal at the very end (nothing left to hide behind)
    for (double total_us : {20.0, 60.0}) {
      const double head = mode == 2 ? total_us : 0.3 * total_us, tail = total_us - head;
      const int K = 300;
      double lat_sum = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < K; ++i) {
          ++epoch;
          if (mode) {
            CK(hipStreamWaitEvent(st, evT, 0));  // the previous step's transfer (a no-op the first time)
          }
          work<<<256, 64, 0, st>>>(signal, epoch, (long long)(head * us), (long long)(tail * us), stamp);
          if (mode) {
            CK(hipStreamWaitValue64(cs, signal, epoch, hipStreamWaitValueGte, ~0ull));
            after<<<1, 64, 0, cs>>>(stamp);
            CK(hipEventRecord(evT, cs));
          }
        }
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        CK(hipStreamSynchronize(cs));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("mode %d kernel %.0f us (signal at %.0f us): %.2f us per step; last store -> next kernel on the waiting stream %.2f us\n", mode, total_us, head,
                        ms * 1e3 / K, mode ? (stamp[1] - stamp[0]) / us : 0.0);
      }
      (void)lat_sum;
    }
  }
  done = true;
  dog.join();
  printf("signal = %llu, epoch = %llu\n", (unsigned long long)*signal, (unsigned long long)epoch);
  return 0;
}
