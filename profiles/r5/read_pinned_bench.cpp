// Tuning aid: how fast do N threads read 1.92 MB files (tmpfs) into pinned (hipHostMalloc) and into ordinary memory?
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
int main() {
  const int nf = 1024, nslot = 32;  // 2 GB of files: what the drop-in loop reads (not cache-resident)
  const size_t bytes = 1920000;
  std::vector<char> src(bytes, 1);
  const bool have = system("test -d /dev/shm/rpb") == 0;  // another process wrote the files (python numpy tofile, as bench.py does)
  system("mkdir -p /dev/shm/rpb");
  for (int i = 0; i < nf && !have; i++) {
    FILE *f = fopen(("/dev/shm/rpb/" + std::to_string(i) + ".bin").c_str(), "wb");
    fwrite(src.data(), 1, bytes, f);
    fclose(f);
  }
  // background activity next to the readers: 0 none, 1 a thread that keeps copying pinned buffers to the device, 2 a thread that
  // keeps allocating and freeing 169 KB blocks (mmap / munmap under glibc's threshold rules)
  for (int bg = 0; bg < 1; bg++)
  for (int pinned = 1; pinned < 2; pinned++) {
    std::vector<char *> buf(nslot);
    for (auto &b : buf) {
      if (pinned == 1) hipHostMalloc((void **)&b, getenv("RPB_POW2") ? 4194304 : 4000000, hipHostMallocDefault);
      else if (pinned == 2) hipHostMalloc((void **)&b, 4000000, hipHostMallocNonCoherent);
      else b = (char *)malloc(4000000);
    }
    std::atomic<bool> stop{false};
    std::thread bgt;
    if (bg == 1)
      bgt = std::thread([&] {
        char *d = nullptr;
        hipMalloc((void **)&d, 4000000 * 8);
        hipStream_t st;
        hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        while (!stop) {
          for (int k = 0; k < 8; k++) hipMemcpyAsync(d + 4000000 * k, buf[k], 1920000, hipMemcpyHostToDevice, st);
          hipStreamSynchronize(st);
        }
        hipFree(d);
      });
    if (bg == 2)
      bgt = std::thread([&] {
        while (!stop) {
          void *p = malloc(169048);
          memset(p, 0, 4096);
          free(p);
        }
      });
    for (int nt : {4, 1, 4}) {
      const auto t0 = std::chrono::steady_clock::now();
      for (int rep = 0; rep < 1; rep++) {
        std::vector<std::thread> th;
        for (int k = 0; k < nt; k++)
          th.emplace_back([&, k] {
            for (int i = k; i < nf; i += nt) {
              FILE *f = fopen(("/dev/shm/rpb/" + std::to_string(i) + ".bin").c_str(), "rb");
              fread(buf[i % nslot], 16, 250000, f);
              fclose(f);
            }
          });
        for (auto &t : th) t.join();
      }
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      printf("background %d %s threads %d: %.1f us per file (wall / files)\n", bg, pinned == 1 ? "pinned " : pinned == 2 ? "pinned-noncoherent" : "malloc ", nt, 1e6 * dt / nf);
    }
    stop = true;
    if (bgt.joinable()) bgt.join();
  }
  system("rm -rf /dev/shm/rpb");
  return 0;
}
