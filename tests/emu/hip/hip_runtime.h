// TEST INFRASTRUCTURE: a tiny CPU execution harness for the HIP kernels of this repo.
//
// There is no GPU in the build container, so kernel *logic* is exercised on the CPU before a
// GPU slot is spent: the kernel headers under contour-context_amd/csrc/ are compiled unchanged
// with g++ against this stand-in for <hip/hip_runtime.h> (tests/emu is put first on the include
// path).  Every HIP thread of a workgroup is a FIBER (its own stack, a user-space context switch of six registers) of the
// OS thread that runs the workgroup; __syncthreads(), the 64-lane cross-lane operations and the 16-lane group operations are
// cooperative barriers (a fiber that has to wait hands over to the next one); the workgroups of a launch are spread over a
// few OS threads, `__shared__` being thread-local statics; __atomic builtins for atomics (workgroups on different OS threads
// really run side by side).  Round 3 used one OS thread per HIP thread and pthread barriers: the suite spent 40 of its 60
// CPU-minutes in futex calls.  It is never part of the product: nothing under contour-context_amd/ includes or links it.
#pragma once
#define CC_EMU 1  // selects the shuffle-based forms of the 16-lane group collectives (csrc/cc_group.h)
#define CC_OPAQUE_I(x) asm volatile("" : "+r"(x))  // csrc/k_contours.h: the optimiser barrier, host constraint
#include <sys/mman.h>
#include <atomic>
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
#define __shared__ static thread_local
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
static inline int4 make_int4(int a, int b, int c, int d) { return {a, b, c, d}; }
struct double2 {
  double x, y;
};
static inline double2 make_double2(double a, double b) { return {a, b}; }
static inline float2 make_float2(float a, float b) { return {a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return {a, b, c, d}; }
static inline int2 make_int2(int a, int b) { return {a, b}; }

namespace emu {
// A cooperative barrier: the last fiber to arrive releases the others; a waiting fiber hands over to the scheduler and
// looks again when it is resumed.
struct CoBarrier {
  int arrived = 0;
  unsigned gen = 0;
};
struct WaveCtx {
  CoBarrier bar;
  unsigned long long scratch64[2][64];  // ping-pong: one barrier per collective is enough
  CoBarrier gbar[4];                    // sub-wave collectives of width 16 (4 groups)
  unsigned long long gscratch[2][64];
  CoBarrier qbar[16];                   // ... of width 4 (16 quads)
  unsigned long long qscratch[2][64];
};
struct Fiber {
  void *sp = nullptr;  // saved stack pointer while the fiber is not running
  dim3 tidx;
  WaveCtx *wave = nullptr;
  int lane = 0;
  unsigned coll = 0, gcoll = 0, qcoll = 0;  // per-fiber count of wave / group / quad collectives (selects the ping-pong buffer)
  bool done = false;
};
struct BlockCtx {
  CoBarrier bar;
  std::vector<WaveCtx> waves;
  std::vector<Fiber> fibers;
  char *dyn_smem = nullptr;
  void (*run)(void *) = nullptr;  // the kernel call of this launch
  void *arg = nullptr;
  void *sched_sp = nullptr;       // the scheduler's saved stack pointer
};
extern thread_local dim3 t_blockIdx, t_blockDim, t_gridDim;
extern thread_local BlockCtx *t_block;
extern thread_local Fiber *t_cur;
extern "C" void emu_switch(void **save_sp, void *load_sp);  // emu_main.cpp: swap callee-saved registers and stacks

static inline void yield() { emu_switch(&t_cur->sp, t_block->sched_sp); }
static inline void co_wait(CoBarrier &b, int n) {
  const unsigned my = b.gen;
  if (++b.arrived == n) {
    b.arrived = 0;
    b.gen++;
    return;
  }
  while (b.gen == my) yield();
}
}  // namespace emu

#define threadIdx (emu::t_cur->tidx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)emu::t_block->dyn_smem;

static inline void __syncthreads() { emu::co_wait(emu::t_block->bar, (int)emu::t_blockDim.x); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int __lane_id() { return emu::t_cur->lane; }
static inline long long wall_clock64() { return 0; }

// ---- cross-lane (wave = 64).  All 64 lanes of the wave must call these together. ----
// A lane can only reach its (n+2)-th collective (which reuses buffer n%2) after passing the barrier of collective
// n+1, i.e. after every lane has finished reading buffer n%2 -- so no second barrier is needed.
static inline unsigned long long __ballot(int pred) {
  emu::Fiber *f = emu::t_cur;
  emu::WaveCtx *w = f->wave;
  unsigned long long *buf = w->scratch64[f->coll++ & 1];
  buf[f->lane] = pred ? 1ull : 0ull;
  emu::co_wait(w->bar, 64);
  unsigned long long m = 0;
  for (int i = 0; i < 64; i++) m |= (buf[i] & 1ull) << i;
  return m;
}
template <typename T>
static inline T emu_shfl_any(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl payload");
  emu::Fiber *f = emu::t_cur;
  emu::WaveCtx *w = f->wave;
  unsigned long long *buf = w->scratch64[f->coll++ & 1];
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  buf[f->lane] = raw;
  emu::co_wait(w->bar, 64);
  unsigned long long r = buf[src & 63];
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
// width-16 variant: only the 16 lanes of the caller's group rendezvous (groups of one wave may diverge)
template <typename T>
static inline T emu_shfl_g16(T v, int src_in_group) {
  emu::Fiber *f = emu::t_cur;
  emu::WaveCtx *w = f->wave;
  const int g = f->lane >> 4;
  unsigned long long *buf = w->gscratch[f->gcoll++ & 1];
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  buf[f->lane] = raw;
  emu::co_wait(w->gbar[g], 16);
  unsigned long long r = buf[g * 16 + (src_in_group & 15)];
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
// width-4 variant: the four lanes of the caller's quad rendezvous (quads of one wave may diverge)
template <typename T>
static inline T emu_shfl_g4(T v, int src_in_quad) {
  emu::Fiber *f = emu::t_cur;
  emu::WaveCtx *w = f->wave;
  const int g = f->lane >> 2;
  unsigned long long *buf = w->qscratch[f->qcoll++ & 1];
  unsigned long long raw = 0;
  std::memcpy(&raw, &v, sizeof(T));
  buf[f->lane] = raw;
  emu::co_wait(w->qbar[g], 4);
  unsigned long long r = buf[g * 4 + (src_in_quad & 3)];
  T out;
  std::memcpy(&out, &r, sizeof(T));
  return out;
}
template <typename T>
static inline T emu_shfl_w(T v, int rel_src, int width) {  // rel_src: lane index within the width-sized group
  if (width == 64) return emu_shfl_any(v, rel_src);
  if (width == 16) return emu_shfl_g16(v, rel_src);
  if (width == 4) return emu_shfl_g4(v, rel_src);
  fprintf(stderr, "emu: unsupported shuffle width %d\n", width);
  abort();
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
  return emu_shfl_w(v, src & (width - 1), width);
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
  const int rel = emu::t_cur->lane & (width - 1);
  const int src = rel + (int)delta;
  return emu_shfl_w(v, src < width ? src : rel, width);
}
template <typename T>
static inline T __shfl_up(T v, unsigned delta, int width = 64) {
  const int rel = emu::t_cur->lane & (width - 1);
  const int src = rel - (int)delta;
  return emu_shfl_w(v, src >= 0 ? src : rel, width);
}
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
  const int rel = emu::t_cur->lane & (width - 1);
  return emu_shfl_w(v, (rel ^ mask) & (width - 1), width);
}
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
static inline int __builtin_amdgcn_readlane(int v, int lane) { return emu_shfl_any(v, lane); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return emu_shfl_any(v, 0); }
// v_mfma_f32_16x16x4_f32: lane l supplies A[row l & 15][k = l >> 4] and B[k = l >> 4][column l & 15] and receives
// D[row 4 (l >> 4) + r][column l & 15], r = 0..3: a k-ordered fmaf chain on top of C (cdna_hip_programming.md section 3)
typedef float emu_f32x4 __attribute__((__vector_size__(16)));
static inline emu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, emu_f32x4 c, int, int, int) {
  emu::Fiber *f = emu::t_cur;
  emu::WaveCtx *w = f->wave;
  unsigned long long *buf = w->scratch64[f->coll++ & 1];
  unsigned ua, ub_;
  std::memcpy(&ua, &a, 4);
  std::memcpy(&ub_, &b, 4);
  buf[f->lane] = ((unsigned long long)ub_ << 32) | ua;
  emu::co_wait(w->bar, 64);
  const int col = f->lane & 15, rq = f->lane >> 4;
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
static inline int __mul24(int a, int b) { return (int)((unsigned)(((a << 8) >> 8) * (long long)((b << 8) >> 8))); }  // low 32 bits of the product of the low 24 bits, sign-extended
static inline unsigned __umul24(unsigned a, unsigned b) { return (unsigned)((unsigned long long)(a & 0xFFFFFFu) * (b & 0xFFFFFFu)); }
static inline int __clz(int x) { return x ? __builtin_clz((unsigned)x) : 32; }
static inline int __clzll(long long x) { return x ? __builtin_clzll((unsigned long long)x) : 64; }

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
static inline double __longlong_as_double(long long x) {
  double f;
  std::memcpy(&f, &x, 8);
  return f;
}
static inline long long __double_as_longlong(double f) {
  long long x;
  std::memcpy(&x, &f, 8);
  return x;
}

// ---- host runtime API stand-ins: "device memory" is host memory ----
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
typedef void *hipStream_t;
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyHostToHost };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
static inline const char *hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipGetDeviceCount(int *n) {  // CC_EMU_DEVICES: how many "devices" the harness shows (multi-rank tests: one per rank)
  const char *e = getenv("CC_EMU_DEVICES");
  *n = e && atoi(e) > 0 ? atoi(e) : 1;
  return hipSuccess;
}
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
template <typename T>
static inline hipError_t hipMalloc(T **p, size_t n) { *p = (T *)malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
typedef void *hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0; return hipSuccess; }
static const unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0;
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { p->multiProcessorCount = 0; return hipSuccess; }
static inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, unsigned, const unsigned *) { *s = nullptr; return hipSuccess; }
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
struct hipFuncAttributes { int numRegs = 0; size_t sharedSizeBytes = 0; };
static inline hipError_t hipFuncGetAttributes(hipFuncAttributes *, const void *) { return hipSuccess; }
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) emu::launch(kernel, (grid).x, (block).x, (size_t)(smem), __VA_ARGS__)

namespace emu {
extern "C" void emu_fiber_entry();  // emu_main.cpp
// Run `kernel(args...)` over grid x block (1-D).  The workgroups are dealt out to a few OS threads; an OS thread runs one
// workgroup at a time, its HIP threads as fibers on stacks of its own, scheduled round-robin: a fiber runs until it has to
// wait at a barrier / collective, or returns from the kernel.
template <typename K, typename... Args>
void launch(K kernel, unsigned grid, unsigned block, size_t dyn_smem, Args... args) {
  if (block % 64 != 0) {
    fprintf(stderr, "emu: block size must be a multiple of 64\n");
    abort();
  }
  if (grid == 0) return;
  auto call = [&]() { kernel(args...); };
  using Call = decltype(call);
  const size_t STK = 256 * 1024;
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 4;
  if (const char *e = getenv("CC_EMU_THREADS")) hw = (unsigned)atoi(e) > 0 ? (unsigned)atoi(e) : hw;
  const unsigned nworkers = grid < hw ? grid : hw;
  std::atomic<unsigned> next{0};
  auto worker = [&]() {
    BlockCtx ctx;
    ctx.waves.resize(block / 64);
    ctx.fibers.resize(block);
    ctx.run = [](void *p) { (*(Call *)p)(); };
    ctx.arg = (void *)&call;
    char *stacks = (char *)mmap(nullptr, STK * block, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char *)MAP_FAILED) {
      fprintf(stderr, "emu: cannot map fiber stacks\n");
      abort();
    }
    std::vector<char> smem(dyn_smem + 64, 0x5a);  // poison: kernels must initialise their LDS
    ctx.dyn_smem = (char *)(((uintptr_t)smem.data() + 15) & ~(uintptr_t)15);
    t_block = &ctx;
    t_blockDim = dim3(block);
    t_gridDim = dim3(grid);
    for (unsigned b = next.fetch_add(1); b < grid; b = next.fetch_add(1)) {
      t_blockIdx = dim3(b);
      if (dyn_smem) memset(ctx.dyn_smem, 0x5a, dyn_smem);
      ctx.bar = CoBarrier();
      for (auto &w : ctx.waves) {
        w.bar = CoBarrier();
        for (auto &g : w.gbar) g = CoBarrier();
      }
      for (unsigned t = 0; t < block; t++) {
        Fiber &f = ctx.fibers[t];
        f.tidx = dim3(t);
        f.wave = &ctx.waves[t / 64];
        f.lane = (int)(t % 64);
        f.coll = f.gcoll = f.qcoll = 0;
        f.done = false;
        // initial frame: six callee-saved registers (zero) and the entry point as the return address; at the entry the
        // stack pointer must be 8 modulo 16, as after a call
        uintptr_t top = ((uintptr_t)(stacks + STK * (t + 1)) & ~(uintptr_t)15) - 8;
        void **sp = (void **)top;
        *--sp = (void *)emu_fiber_entry;
        for (int r = 0; r < 6; r++) *--sp = nullptr;
        f.sp = (void *)sp;
      }
      unsigned live = block;
      while (live) {
        for (unsigned t = 0; t < block; t++) {
          Fiber &f = ctx.fibers[t];
          if (f.done) continue;
          t_cur = &f;
          emu_switch(&ctx.sched_sp, f.sp);
          if (f.done) live--;
        }
      }
    }
    t_cur = nullptr;
    t_block = nullptr;
    munmap(stacks, STK * block);
  };
  if (nworkers <= 1) {
    worker();
  } else {
    std::vector<std::thread> th;
    th.reserve(nworkers);
    for (unsigned i = 0; i < nworkers; i++) th.emplace_back(worker);
    for (auto &x : th) x.join();
  }
}
}  // namespace emu
