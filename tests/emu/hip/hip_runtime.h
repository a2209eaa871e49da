// TEST INFRASTRUCTURE: a tiny CPU execution harness for the HIP kernels of this repo.
//
// There is no GPU in the build container, so kernel *logic* is exercised on the CPU before a
// GPU slot is spent: the kernel headers under contour-context_amd/csrc/ are compiled unchanged
// with g++ against this stand-in for <hip/hip_runtime.h> (tests/emu is put first on the include
// path).  One OS thread per HIP thread, pthread barriers for __syncthreads(), per-wave barriers
// for the 64-lane cross-lane ops, __atomic builtins for atomics.  It is never part of the product:
// nothing under contour-context_amd/ includes or links it.
#pragma once
#define CC_EMU 1  // selects the shuffle-based forms of the 16-lane group collectives (csrc/cc_group.h)
#define CC_OPAQUE_I(x) asm volatile("" : "+r"(x))  // csrc/k_contours.h: the optimiser barrier, host constraint
#include <pthread.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

using std::isfinite;
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct float2 {
  float x, y;
};
struct float4 {
  float x, y, z, w;
};
struct int2 {
  int x, y;
};
struct uint2 {
  unsigned x, y;
};
struct int4 {
  int x, y, z, w;
};
static inline uint2 make_uint2(unsigned a, unsigned b) { return {a, b}; }
struct uint4 {
  unsigned x, y, z, w;
};
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return {a, b, c, d}; }
struct double2 {
  double x, y;
};
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return {a, b}; }

namespace emu {
struct WaveCtx {
  pthread_barrier_t bar;
  unsigned long long scratch64[2][64];  // ping-pong: one barrier per collective is enough
  pthread_barrier_t gbar[4];            // sub-wave collectives of width 16 (4 groups)
  unsigned long long gscratch[2][64];
};
struct BlockCtx {
  pthread_barrier_t bar;
  std::vector<WaveCtx> waves;
  char *dyn_smem = nullptr;
};
extern thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockCtx *t_block;
extern thread_local WaveCtx *t_wave;
extern thread_local int t_lane;
extern thread_local unsigned t_coll;  // per-thread count of wave collectives (selects the ping-pong buffer)
extern thread_local unsigned t_gcoll; // same for the width-16 collectives
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)emu::t_block->dyn_smem;

static inline void __syncthreads() { pthread_barrier_wait(&emu::t_block->bar); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int __lane_id() { return emu::t_lane; }
static inline long long wall_clock64() { return 0; }

// ---- cross-lane (wave = 64).  All 64 lanes of the wave must call these together. ----
// A lane can only reach its (n+2)-th collective (which reuses buffer n%2) after passing the barrier of collective
// n+1, i.e. after every lane has finished reading buffer n%2 -- so no second barrier is needed.
static inline unsigned long long __ballot(int pred) {
  emu::WaveCtx *w = emu::t_wave;
  unsigned long long *buf = w->scratch64[emu::t_coll++ & 1];
  buf[emu::t_lane] = pred ? 1ull : 0ull;
  pthread_barrier_wait(&w->bar);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) m |= (buf[i] & 1ull) << i;
  return m;
}
template <typename T>
static inline T emu_shfl_any(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  emu::WaveCtx *w = emu::t_wave;
  unsigned long long *buf = w->scratch64[emu::t_coll++ & 1];
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  buf[emu::t_lane] = raw;
  pthread_barrier_wait(&w->bar);
  unsigned long long r = buf[src & 63];
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
// width-16 variant: only the 16 lanes of the caller's group rendezvous (groups of one wave may diverge)
template <typename T>
static inline T emu_shfl_g16(T v, int src_in_group) {
  emu::WaveCtx *w = emu::t_wave;
  const int g = emu::t_lane >> 4;
  unsigned long long *buf = w->gscratch[emu::t_gcoll++ & 1];
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  buf[emu::t_lane] = raw;
  pthread_barrier_wait(&w->gbar[g]);
  unsigned long long r = buf[g * 16 + (src_in_group & 15)];
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
template <typename T>
static inline T emu_shfl_w(T v, int rel_src, int width) {  // rel_src: lane index within the width-sized group
  if (width == 64) return emu_shfl_any(v, rel_src);
  if (width == 16) return emu_shfl_g16(v, rel_src);
  fprintf(stderr, "emu: unsupported shuffle width %d\n", width);
  abort();
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
  return emu_shfl_w(v, src & (width - 1), width);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  const int rel = emu::t_lane & (width - 1);
  const int src = rel + (int)delta;
  return emu_shfl_w(v, src < width ? src : rel, width);
}
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  const int rel = emu::t_lane & (width - 1);
  const int src = rel - (int)delta;
  return emu_shfl_w(v, src >= 0 ? src : rel, width);
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  const int rel = emu::t_lane & (width - 1);
  return emu_shfl_w(v, (rel ^ mask) & (width - 1), width);
}
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
static inline int __builtin_amdgcn_readlane(int v, int lane) { return emu_shfl_any(v, lane); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu_shfl_any(v, 0); }
// v_mfma_f32_16x16x4_f32: lane l supplies A[row l & 15][k = l >> 4] and B[k = l >> 4][column l & 15] and receives
// D[row 4 (l >> 4) + r][column l & 15], r = 0..3: a k-ordered fmaf chain on top of C (cdna_hip_programming.md section 3)
typedef float emu_f32x4 __attribute__((__vector_size__(16)));
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  emu::WaveCtx *w = emu::t_wave;
  unsigned long long *buf = w->scratch64[emu::t_coll++ & 1];
  unsigned ua, ub_;
  std::memcpy(&ua, &a, 4);
  std::memcpy(&ub_, &b, 4);
  buf[emu::t_lane] = ((unsigned long long)ub_ << 32) | ua;
  pthread_barrier_wait(&w->bar);
  const int col = emu::t_lane & 15, rq = emu::t_lane >> 4;
  emu_f32x4 d = c;
  for (int r = 0; r < 4; r++) {
    const int row = 4 * rq + r;
    float acc = c[r];
    for (int k = 0; k < 4; k++) {
      const unsigned xa = (unsigned)(buf[row + 16 * k] & 0xFFFFFFFFull), xb = (unsigned)(buf[col + 16 * k] >> 32);
      float fa, fb;
      std::memcpy(&fa, &xa, 4);
      std::memcpy(&fb, &xb, 4);
      acc = std::fmaf(fa, fb, acc);
    }
    d[r] = acc;
  }
  return d;
}
static inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
static inline int __popc(unsigned x) { return __builtin_popcount(x); }
static inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
static inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }

// ---- atomics ----
template <typename T>
static inline T emu_atomic_minmax(T *p, T v, bool is_max) {
  T old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (true) {
    T want = is_max ? (old > v ? old : v) : (old < v ? old : v);
    if (want == old) return old;
    if (__atomic_compare_exchange_n(p, &old, want, false, __ATOMIC_SEQ_CST, __ATOMIC_RELAXED)) return old;
  }
}
static inline unsigned atomicMax(unsigned *p, unsigned v) { return emu_atomic_minmax(p, v, true); }
static inline unsigned atomicMin(unsigned *p, unsigned v) { return emu_atomic_minmax(p, v, false); }
static inline int atomicMax(int *p, int v) { return emu_atomic_minmax(p, v, true); }
static inline int atomicMin(int *p, int v) { return emu_atomic_minmax(p, v, false); }
static inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { return emu_atomic_minmax(p, v, true); }
static inline unsigned long long atomicMin(unsigned long long *p, unsigned long long v) { return emu_atomic_minmax(p, v, false); }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicOr(unsigned *p, unsigned v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned long long atomicOr(unsigned long long *p, unsigned long long v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicAnd(unsigned *p, unsigned v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicExch(unsigned *p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline int atomicExch(int *p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicCAS(unsigned *p, unsigned cmp, unsigned val) {
  __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
static inline int atomicCAS(int *p, int cmp, int val) {
  __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}
static inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long cmp, unsigned long long val) {
  __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
  return cmp;
}

static inline float __int_as_float(int x) {
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}
static inline int __float_as_int(float f) {
  int x;
  std::memcpy(&x, &f, 4);
  return x;
}
static inline unsigned __float_as_uint(float f) {
  unsigned x;
  std::memcpy(&x, &f, 4);
  return x;
}
static inline float __uint_as_float(unsigned x) {
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// ---- host runtime API stand-ins: "device memory" is host memory ----
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
typedef void *hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
template <typename T>
static inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
typedef void *hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
static const unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0;
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
// graphs: the harness runs every launch at once, so a "captured" chain has already run when the capture ends
typedef void *hipGraph_t;
typedef void *hipGraphExec_t;
typedef void *hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
enum hipGraphExecUpdateResult { hipGraphExecUpdateSuccess };
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone, hipStreamCaptureStatusActive };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = nullptr; return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t) { *e = (void *)1; return hipSuccess; }
static inline hipError_t hipGraphExecUpdate(hipGraphExec_t, hipGraph_t, hipGraphNode_t *, hipGraphExecUpdateResult *) { return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) emu::launch(kernel, (grid).x, (block).x, (size_t)(smem), __VA_ARGS__)

namespace emu {
// Run `kernel(args...)` over grid x block (1-D), one block at a time.
template <typename K, typename... Args>
void launch(K kernel, unsigned grid, unsigned block, size_t dyn_smem, Args... args);
}  // namespace emu
namespace emu {
// One set of `block` OS threads per launch, reused for every workgroup of the grid (workgroups run one after the other:
// `__shared__` variables are function-local statics shared by all threads).  A thread that returns early from the kernel
// simply waits at the end-of-workgroup barrier.
template <typename K, typename... Args>
void launch(K kernel, unsigned grid, unsigned block, size_t dyn_smem, Args... args) {
  if (block % 64 != 0) {
    fprintf(stderr, "emu: block size must be a multiple of 64\n");
    abort();
  }
  if (grid == 0) return;
  BlockCtx ctx;
  pthread_barrier_init(&ctx.bar, nullptr, block);
  ctx.waves.resize(block / 64);
  for (auto &w : ctx.waves) {
    pthread_barrier_init(&w.bar, nullptr, 64);
    for (auto &g : w.gbar) pthread_barrier_init(&g, nullptr, 16);
  }
  pthread_barrier_t next_bar;  // between workgroups (separate from ctx.bar: a kernel may leave threads at different phases)
  pthread_barrier_init(&next_bar, nullptr, block);
  std::vector<char> smem(dyn_smem + 64, 0x5a);  // poison: kernels must initialise their LDS
  ctx.dyn_smem = (char *)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
  std::vector<std::thread> th;
  th.reserve(block);
  for (unsigned t = 0; t < block; t++) {
    th.emplace_back([&, t]() {
      t_threadIdx = dim3(t);
      t_blockDim = dim3(block);
      t_gridDim = dim3(grid);
      t_block = &ctx;
      t_wave = &ctx.waves[t / 64];
      t_lane = t % 64;
      for (unsigned b = 0; b < grid; b++) {
        t_blockIdx = dim3(b);
        t_coll = 0;
        t_gcoll = 0;
        kernel(args...);
        pthread_barrier_wait(&next_bar);
        if (t == 0 && dyn_smem) memset(ctx.dyn_smem, 0x5a, dyn_smem);
        if (dyn_smem) pthread_barrier_wait(&next_bar);
      }
    });
  }
  for (auto &x : th) x.join();
  pthread_barrier_destroy(&ctx.bar);
  pthread_barrier_destroy(&next_bar);
  for (auto &w : ctx.waves) {
    pthread_barrier_destroy(&w.bar);
    for (auto &g : w.gbar) pthread_barrier_destroy(&g);
  }
}
}  // namespace emu
