// Tuning aid: per-stream creation cost, sequential and from parallel threads
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k(int *p) { if (p) *p = 1; }
int main(int argc, char **argv) {
  const bool par = argc > 1;
  double t = now();
  hipSetDevice(0);
  hipFree(nullptr);
  printf("init %.2f ms\n", 1e3 * (now() - t));
  t = now();
  hipFuncAttributes fa;
  hipFuncGetAttributes(&fa, (const void *)k);
  printf("hipFuncGetAttributes (module load) %.2f ms\n", 1e3 * (now() - t));
  hipStream_t s[12];
  if (!par) {
    for (int i = 0; i < 12; i++) {
      t = now();
      hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking);
      const double a = now();
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s[i], (int *)nullptr);
      hipStreamSynchronize(s[i]);
      printf("stream %d: create %.2f ms, first launch + sync %.2f ms\n", i, 1e3 * (a - t), 1e3 * (now() - a));
    }
  } else {
    t = now();
    std::vector<std::thread> th;
    for (int i = 0; i < 8; i++) th.emplace_back([&, i] { hipSetDevice(0); hipStreamCreateWithFlags(&s[i], hipStreamNonBlocking); });
    for (auto &x : th) x.join();
    printf("8 streams from 8 threads: %.2f ms\n", 1e3 * (now() - t));
  }
  return 0;
}
