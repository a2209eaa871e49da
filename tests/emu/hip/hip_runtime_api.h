// TEST INFRASTRUCTURE: what a C++ host program on the C-ABI (hostcpp/examples/batch_replay_mgpu.cpp) uses of the HIP runtime
// API, for the CPU harness: "device memory" is host memory, everything is synchronous.  Only ever on the include path of
// harness builds (tests/); the product builds take the real header from /opt/rocm.
#pragma once
#include <cstdlib>
#include <cstring>
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
typedef void *hipStream_t;
typedef void *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
static inline const char *hipGetErrorString(hipError_t) { return "harness"; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
template <typename T>
static inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void *d, size_t dpitch, const void *s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
  for (size_t r = 0; r < height; r++) memcpy((char *)d + r * dpitch, (const char *)s + r * spitch, width);
  return hipSuccess;
}
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 1.f; return hipSuccess; }
