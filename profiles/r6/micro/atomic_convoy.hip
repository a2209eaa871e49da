// Microbenchmark (round 6): what N workgroups pay for ONE returning atomicAdd each on the same address (the per-workgroup
// list compaction pattern: count, take a base from the list head, write), against one address per workgroup.
//   hipcc -O3 --offload-arch=gfx950 atomic_convoy.hip -o atomic_convoy && ./atomic_convoy
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_same(int *head, int *out, int work) {
  __shared__ int s;
  int acc = 0;
  for (int i = 0; i < work; i++) acc += (threadIdx.x * 31 + i) & 7;   // a little work in front, like counting
  if (threadIdx.x == 0) s = atomicAdd(head, 1 + (acc & 1));
  __syncthreads();
  if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = s;
}
__global__ void k_diff(int *heads, int *out, int work) {
  __shared__ int s;
  int acc = 0;
  for (int i = 0; i < work; i++) acc += (threadIdx.x * 31 + i) & 7;
  if (threadIdx.x == 0) s = atomicAdd(&heads[blockIdx.x * 32], 1 + (acc & 1));
  __syncthreads();
  if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = s;
}
__global__ void k_none(int *heads, int *out, int work) {
  __shared__ int s;
  int acc = 0;
  for (int i = 0; i < work; i++) acc += (threadIdx.x * 31 + i) & 7;
  if (threadIdx.x == 0) s = heads[blockIdx.x * 32] + (acc & 1);
  __syncthreads();
  if (threadIdx.x < 4) out[blockIdx.x * 4 + threadIdx.x] = s;
}
int main() {
  int *head, *out;
  hipMalloc(&head, 1 << 22);
  hipMalloc(&out, 1 << 22);
  hipMemset(head, 0, 1 << 22);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int n : {1024, 4608, 16384}) {
    for (int block : {64, 256}) {
      float ms[3];
      for (int v = 0; v < 3; v++) {
        for (int rep = 0; rep < 3; rep++) {
          hipEventRecord(e0);
          for (int i = 0; i < 20; i++) {
            if (v == 0) hipLaunchKernelGGL(k_same, dim3(n), dim3(block), 0, 0, head, out, 64);
            if (v == 1) hipLaunchKernelGGL(k_diff, dim3(n), dim3(block), 0, 0, head, out, 64);
            if (v == 2) hipLaunchKernelGGL(k_none, dim3(n), dim3(block), 0, 0, head, out, 64);
          }
          hipEventRecord(e1);
          hipEventSynchronize(e1);
          hipEventElapsedTime(&ms[v], e0, e1);
        }
      }
      printf("workgroups %6d x %3d threads: same address %7.1f us  own address %7.1f us  no atomic %7.1f us per launch\n", n, block,
             ms[0] * 50.f, ms[1] * 50.f, ms[2] * 50.f);
    }
  }
  return 0;
}
