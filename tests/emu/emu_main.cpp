// TEST INFRASTRUCTURE: the product's HIP translation unit (kernels + C-ABI host code) compiled
// unchanged for the CPU against tests/emu/hip/hip_runtime.h, so the whole cc_* C-ABI -- kernel logic
// and host bookkeeping -- can be exercised against the oracle without a GPU.  Built by
// tests/emu/build.sh into tests/emu/libcc_emu.so.  Never loaded by the product.
#define CC_INGEST_BLOCK 256
#define CC_GMM_GRID 16  // grid-stride kernel: fewer workgroups = fewer OS threads to create, same results
#include "../../contour-context_amd/csrc/cont2_amd.hip"

namespace emu {
thread_local dim3 t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx *t_block;
thread_local Fiber *t_cur;
}  // namespace emu

// The context switch of the harness' fibers (x86-64 System V): push the callee-saved registers, swap the stack pointers,
// pop the other side's registers, return into it.
asm(R"(
.text
.globl emu_switch
.type emu_switch, @function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch, .-emu_switch
)");
// where a new fiber starts: run the launch's kernel call for this HIP thread, then hand over to the scheduler for good
extern "C" void emu_fiber_entry() {
  emu::BlockCtx *blk = emu::t_block;
  blk->run(blk->arg);
  emu::t_cur->done = true;
  emu::emu_switch(&emu::t_cur->sp, blk->sched_sp);
  abort();  // a finished fiber is never resumed
}

// the wave-parallel std::sort replay (cc_sort.h: std_sort_wave) on one array, as K2's size sort uses it
__global__ void emu_k_sort_wave(unsigned *arr, const unsigned *pristine, int n) {
  __shared__ unsigned a[4096];
  __shared__ unsigned short st[2 * 4096];
  __shared__ unsigned seg[CC_SORT_STACK];
  const int lane = threadIdx.x;
  for (int i = lane; i < n; i += 64) a[i] = pristine[i];
  ccsort::std_sort_wave(
      a, n, [](unsigned x) { return 0xFFFFu - (x >> 16); },
      [&]() {
        for (int i = lane; i < n; i += 64) a[i] = pristine[i];
      },
      lane, st, st + 4096, (unsigned *)st, seg);
  for (int i = lane; i < n; i += 64) arr[i] = a[i];
}

// the order kernel's workgroup sort (k_knn.h: cc_block_bitonic_u32), 1024 * R keys
template <int R>
__global__ void emu_k_block_bitonic(unsigned *arr) {
  __shared__ unsigned xch[1024 * R];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned v[R];
  for (int a = 0; a < R; a++) v[a] = arr[(wave * R + a) * 64 + lane];
  cc_block_bitonic_u32<R>(v, xch, tid);
  for (int a = 0; a < R; a++) arr[(wave * R + a) * 64 + lane] = v[a];
}
// and its in-place block scans (cc_block_scan), n a power of two <= 8192
__global__ void emu_k_block_scan(int *arr, int n, int is_max) {
  __shared__ int a[8192];
  __shared__ int wsum[16];
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += 1024) a[i] = arr[i];
  __syncthreads();
  if (is_max)
    cc_block_scan<true>(a, n, tid, wsum);
  else
    cc_block_scan<false>(a, n, tid, wsum);
  for (int i = tid; i < n; i += 1024) arr[i] = a[i];
}

extern "C" {
void emu_block_bitonic(unsigned *arr, int r) {
  if (r == 1) hipLaunchKernelGGL(emu_k_block_bitonic<1>, dim3(1), dim3(1024), 0, nullptr, arr);
  if (r == 4) hipLaunchKernelGGL(emu_k_block_bitonic<4>, dim3(1), dim3(1024), 0, nullptr, arr);
  if (r == 8) hipLaunchKernelGGL(emu_k_block_bitonic<8>, dim3(1), dim3(1024), 0, nullptr, arr);
}
void emu_block_scan(int *arr, int n, int is_max) { hipLaunchKernelGGL(emu_k_block_scan, dim3(1), dim3(1024), 0, nullptr, arr, n, is_max); }
void emu_sort_desc_wave(unsigned *arr, int n) {
  std::vector<unsigned> in(arr, arr + n);
  hipLaunchKernelGGL(emu_k_sort_wave, dim3(1), dim3(64), 0, nullptr, arr, (const unsigned *)in.data(), n);
}
// unit hooks for the std::sort replica and the 2x2 eigen solver
void emu_sort_desc(unsigned *arr, int n) {
  ccsort::std_sort(arr, n, [](unsigned x, unsigned y) { return (x >> 16) > (y >> 16); });
}
struct emu_fkey {
  float k;
  int idx;
};
void emu_sort_asc_f(emu_fkey *arr, int n) {
  ccsort::std_sort(arr, n, [](const emu_fkey &x, const emu_fkey &y) { return x.k < y.k; });
}
void emu_eigen2f(const float m[3], float ev[2], float vec[4]) { cc_eigen2f(m[0], m[1], m[2], ev, vec); }
// the atan2f replica on arrays (tests/test_atan2f_replica.py)
void emu_atan2f(const float *y, const float *x, float *out, long n) {
  for (long i = 0; i < n; i++) out[i] = cc_atan2f_fdlibm(y[i], x[i]);
}
// ... and the acosf replica
void emu_acosf(const float *x, float *out, long n) {
  for (long i = 0; i < n; i++) out[i] = cc_acosf_fdlibm(x[i]);
}
}
