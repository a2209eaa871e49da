// Tuning aid: what do the allocation calls of a context's start cost on this host?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipSetDevice(0);
  hipFree(nullptr);
  double t = now();
  auto lap = [&](const char *what, int n) {
    const double x = now();
    printf("%-50s %8.3f ms each\n", what, 1e3 * (x - t) / n);
    t = now();
  };
  void *p[64];
  for (int i = 0; i < 16; i++) hipMalloc(&p[i], 4096);
  lap("hipMalloc 4 KB", 16);
  for (int i = 0; i < 16; i++) hipMalloc(&p[i], 1 << 20);
  lap("hipMalloc 1 MB", 16);
  for (int i = 0; i < 8; i++) hipMalloc(&p[i], 64 << 20);
  lap("hipMalloc 64 MB", 8);
  hipMalloc(&p[0], (size_t)1 << 30);
  lap("hipMalloc 1 GB", 1);
  for (int i = 0; i < 8; i++) hipHostMalloc(&p[i], 4096, hipHostMallocDefault);
  lap("hipHostMalloc 4 KB", 8);
  for (int i = 0; i < 8; i++) hipHostMalloc(&p[i], 4 << 20, hipHostMallocDefault);
  lap("hipHostMalloc 4 MB", 8);
  hipHostMalloc(&p[0], 64 << 20, hipHostMallocDefault);
  lap("hipHostMalloc 64 MB", 1);
  hipHostMalloc(&p[0], 136 << 20, hipHostMallocDefault);
  lap("hipHostMalloc 136 MB", 1);
  hipStream_t s[8];
  for (int i = 0; i < 8; i++) hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
  lap("hipStreamCreateWithFlags", 8);
  hipEvent_t e[64];
  for (int i = 0; i < 64; i++) hipEventCreateWithFlags(&e[i], hipEventDisableTiming);
  lap("hipEventCreateWithFlags", 64);
  for (int i = 0; i < 8; i++) hipMemsetAsync(p[0], 0, 64, s[i]);
  lap("first hipMemsetAsync on a new stream", 8);
  for (int i = 0; i < 8; i++) hipStreamSynchronize(s[i]);
  lap("hipStreamSynchronize", 8);
  return 0;
}
