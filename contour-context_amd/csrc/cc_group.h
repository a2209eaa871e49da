// 16-lane group collectives.  The query-side kernels give one candidate check / one correlation problem to a group of
// 16 lanes (a DPP "row"), four groups per wave; the groups of a wave may sit in different branches, so every
// collective below involves only the caller's own row:
//   * ballots: the wave-wide v_cmp mask, shifted to the row's 16 bits (inactive rows contribute zeros to a mask nobody
//     reads);
//   * prefix sums / reductions: DPP row_shr, quad_perm and row_mirror moves (register-to-register, no LDS crossbar);
//   * broadcast of one lane's value: ds_bpermute through __shfl(width 16).
// CC_EMU (defined only by the CPU test harness' stand-in for <hip/hip_runtime.h>) selects plain width-16 shuffles, which
// is what that harness can rendezvous on; the product build never sees it.
#pragma once
#include <hip/hip_runtime.h>

#define CC_G 16

// hand-off of LDS data between the lanes of ONE wave: LDS operations of a wave execute in issue order, only the compiler has
// to be kept from reordering them.  The CPU harness runs the lanes as OS threads and needs a real rendezvous.
__device__ __forceinline__ void cc_wave_sync() {
#ifndef CC_EMU
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  (void)__ballot(1);  // rendezvous of the wave's 64 OS threads (a workgroup may hold several waves)
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}


// bits of a wave-wide mask below the caller's lane: popcount(m & ((1 << lane) - 1)) as the two v_mbcnt instructions it is (the
// generic expression costs two ANDs and two bit counts; with the mask in scalar registers nothing else)
__device__ __forceinline__ int cc_mbcnt(unsigned long long m) {
#ifndef CC_EMU
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
#else
  return __builtin_popcountll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
#endif
}
// the wave's index inside its workgroup as a SCALAR (threadIdx.x >> 6 is a vector value to the compiler: loops over "my wave's
// chunks" then run on exec masks, and every ballot result they use sits in vector registers)
__device__ __forceinline__ int cc_wave_id() {
#ifndef CC_EMU
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
#else
  return (int)(threadIdx.x >> 6);
#endif
}

// (m << 1) | (sign bit of x): a comparison result appended to a bit string by ONE instruction (v_alignbit_b32) instead of a
// compare plus a conditional OR
__device__ __forceinline__ unsigned cc_push_sign(unsigned m, float x) {
#ifndef CC_EMU
  return __builtin_amdgcn_alignbit(m, __float_as_uint(x), 31);
#else
  return (m << 1) | (__float_as_uint(x) >> 31);
#endif
}
// two f32 values per lane and instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32)
#ifndef CC_EMU
typedef float cc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cc_f2 cc_pk_fma(cc_f2 a, cc_f2 b, cc_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ unsigned cc_brev(unsigned v) { return __brev(v); }
#else
struct cc_f2 {
  float x, y;
};
inline cc_f2 operator+(cc_f2 a, cc_f2 b) { return cc_f2{a.x + b.x, a.y + b.y}; }
inline cc_f2 operator-(cc_f2 a, cc_f2 b) { return cc_f2{a.x - b.x, a.y - b.y}; }
inline cc_f2 operator*(cc_f2 a, cc_f2 b) { return cc_f2{a.x * b.x, a.y * b.y}; }
inline cc_f2 operator-(cc_f2 a) { return cc_f2{-a.x, -a.y}; }
inline cc_f2 cc_pk_fma(cc_f2 a, cc_f2 b, cc_f2 c) { return cc_f2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }
inline unsigned cc_brev(unsigned v) {
  unsigned r = 0;
  for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
  return r;
}
#endif

// a value that is the same in every lane of the wave, handed to the compiler as a SCALAR (loaded through a vector load it is a
// vector value to the compiler, and every address built from it costs vector registers and vector arithmetic)
__device__ __forceinline__ int cc_uniform_i(int v) {
#ifndef CC_EMU
  return __builtin_amdgcn_readfirstlane(v);
#else
  return v;
#endif
}

template <typename T>
__device__ __forceinline__ T *cc_uniform_ptr(T *p) {
#ifndef CC_EMU
  const unsigned long long a = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
#else
  return p;
#endif
}

// Volatile reads of LDS words that other lanes are changing (the union-find forest of K2).  A plain `volatile T *` made from a
// generic pointer keeps the GENERIC address space -- the address-space inference leaves volatile accesses alone -- and is
// compiled to flat_load + s_waitcnt vmcnt(0) per access (round 4's labelling ran on those); spelled with the LDS address
// space the same read is a ds_read that is waited for where its value is used.
#ifndef CC_EMU
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
__device__ __forceinline__ unsigned cc_lds_vread16(const uint16_t *p) {
  return *(const volatile __attribute__((address_space(3))) uint16_t *)p;
}
__device__ __forceinline__ unsigned cc_lds_vread32(const unsigned *p) {
  return *(const volatile __attribute__((address_space(3))) unsigned *)p;
}
__device__ __forceinline__ void cc_lds_vwrite16(uint16_t *p, unsigned v) {
  *(volatile __attribute__((address_space(3))) uint16_t *)p = (uint16_t)v;
}
// Request the cache line of a global address without using the value: a volatile load is issued where it stands, nothing
// waits for it (its destination register is not reused before it has arrived: the compiler tracks that like any load).
__device__ __forceinline__ void cc_touch_global(const void *p) {
  (void)*(const volatile __attribute__((address_space(1))) unsigned *)p;
}
#pragma clang diagnostic pop
__device__ __forceinline__ double cc_rsq_seed(double x) { return __builtin_amdgcn_rsq(x); }  // v_rsq_f64
#else
__device__ __forceinline__ void cc_touch_global(const void *) {}
__device__ __forceinline__ unsigned cc_lds_vread16(const uint16_t *p) { return *(const volatile uint16_t *)p; }
__device__ __forceinline__ unsigned cc_lds_vread32(const unsigned *p) { return *(const volatile unsigned *)p; }
__device__ __forceinline__ void cc_lds_vwrite16(uint16_t *p, unsigned v) { *(volatile uint16_t *)p = (uint16_t)v; }
__device__ __forceinline__ double cc_rsq_seed(double x) { return 1.0 / sqrt(x); }
#endif

#ifndef CC_EMU
// LDS hand-off between the lanes of one group: a wave's lanes run in lockstep and its LDS operations are executed in
// issue order, so only the compiler has to be kept from moving LDS accesses across the hand-off (a wavefront-scope fence
// emits no wait by itself as long as the accesses are ds_* instructions).
__device__ __forceinline__ void cc_group_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

template <int CTRL>
__device__ __forceinline__ int cc_dpp_i(int v) {  // out-of-row sources read as 0
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double cc_dpp_d(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = cc_dpp_i<CTRL>((int)(b & 0xFFFFFFFFll)), hi = cc_dpp_i<CTRL>((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned long long)(unsigned)lo);
}
#define CC_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define CC_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define CC_DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i inside each half row
#define CC_DPP_MIRROR 0x140       // lane i <-> 15 - i inside the row

// bit i = pred of group lane i
__device__ __forceinline__ unsigned cc_group_ballot(bool pred) {
  const unsigned long long m = __ballot(pred);
  return (unsigned)(m >> (threadIdx.x & 48u)) & 0xFFFFu;
}
// inclusive prefix sum over the group's lanes
__device__ __forceinline__ int cc_group_scan_incl(int v) {
  v += cc_dpp_i<0x111>(v);  // row_shr:1
  v += cc_dpp_i<0x112>(v);
  v += cc_dpp_i<0x114>(v);
  v += cc_dpp_i<0x118>(v);
  return v;
}
// sum over the group, result in every lane
__device__ __forceinline__ int cc_group_sum_i(int v) {
  v += cc_dpp_i<CC_DPP_XOR1>(v);
  v += cc_dpp_i<CC_DPP_XOR2>(v);
  v += cc_dpp_i<CC_DPP_HALF_MIRROR>(v);
  v += cc_dpp_i<CC_DPP_MIRROR>(v);
  return v;
}
// bitwise OR over the group, result in every lane
__device__ __forceinline__ unsigned cc_group_or_u(unsigned v) {
  v |= (unsigned)cc_dpp_i<CC_DPP_XOR1>((int)v);
  v |= (unsigned)cc_dpp_i<CC_DPP_XOR2>((int)v);
  v |= (unsigned)cc_dpp_i<CC_DPP_HALF_MIRROR>((int)v);
  v |= (unsigned)cc_dpp_i<CC_DPP_MIRROR>((int)v);
  return v;
}
__device__ __forceinline__ double cc_group_sum_d(double v) {
  v += cc_dpp_d<CC_DPP_XOR1>(v);
  v += cc_dpp_d<CC_DPP_XOR2>(v);
  v += cc_dpp_d<CC_DPP_HALF_MIRROR>(v);
  v += cc_dpp_d<CC_DPP_MIRROR>(v);
  return v;
}
// lexicographic (larger a, then smaller b) over the group, result in every lane
__device__ __forceinline__ void cc_group_best(int &a, int &b) {
#define CC_GB_STEP(CTRL)                                   \
  {                                                        \
    const int oa = cc_dpp_i<CTRL>(a), ob = cc_dpp_i<CTRL>(b); \
    if (oa > a || (oa == a && ob < b)) {                   \
      a = oa;                                              \
      b = ob;                                              \
    }                                                      \
  }
  CC_GB_STEP(CC_DPP_XOR1)
  CC_GB_STEP(CC_DPP_XOR2)
  CC_GB_STEP(CC_DPP_HALF_MIRROR)
  CC_GB_STEP(CC_DPP_MIRROR)
#undef CC_GB_STEP
}
// value of group lane `src` (any lane-varying or uniform index 0..15)
template <typename T>
__device__ __forceinline__ T cc_group_bcast(T v, int src) {
  return __shfl(v, src, CC_G);
}
#else  // ---------------------------------------------------------------- CPU test harness
__device__ __forceinline__ void cc_group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  (void)__shfl(0, 0, CC_G);  // rendezvous of the group's 16 OS threads
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}
__device__ __forceinline__ unsigned cc_group_ballot(bool pred) {
  const int sl = threadIdx.x & 15;
  int v = pred ? (1 << sl) : 0;
  for (int o = 1; o < CC_G; o <<= 1) v |= __shfl_xor(v, o, CC_G);
  return (unsigned)v;
}
__device__ __forceinline__ int cc_group_scan_incl(int v) {
  const int sl = threadIdx.x & 15;
  for (int o = 1; o < CC_G; o <<= 1) {
    const int t = __shfl_up(v, o, CC_G);
    if (sl >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ int cc_group_sum_i(int v) {
  for (int o = 1; o < CC_G; o <<= 1) v += __shfl_xor(v, o, CC_G);
  return v;
}
__device__ __forceinline__ unsigned cc_group_or_u(unsigned v) {
  for (int o = 1; o < CC_G; o <<= 1) v |= (unsigned)__shfl_xor((int)v, o, CC_G);
  return v;
}
__device__ __forceinline__ double cc_group_sum_d(double v) {
  // same association as the DPP form: xor 1, xor 2, then the two mirror steps pair up the same partial sums
  v += __shfl_xor(v, 1, CC_G);
  v += __shfl_xor(v, 2, CC_G);
  const int sl = threadIdx.x & 15;
  v += __shfl(v, (sl & 8) | (7 - (sl & 7)), CC_G);
  v += __shfl(v, 15 - sl, CC_G);
  return v;
}
__device__ __forceinline__ void cc_group_best(int &a, int &b) {
  for (int o = 1; o < CC_G; o <<= 1) {
    const int oa = __shfl_xor(a, o, CC_G), ob = __shfl_xor(b, o, CC_G);
    if (oa > a || (oa == a && ob < b)) {
      a = oa;
      b = ob;
    }
  }
}
template <typename T>
__device__ __forceinline__ T cc_group_bcast(T v, int src) {
  return __shfl(v, src, CC_G);
}
#endif

// value of lane Q of this lane's quad (lanes 4 k .. 4 k + 3)
#ifndef CC_EMU
template <int Q>
__device__ __forceinline__ float cc_quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), Q | (Q << 2) | (Q << 4) | (Q << 6), 0xF, 0xF, true));  // quad_perm [Q,Q,Q,Q]
}
#else
template <int Q>
__device__ __forceinline__ float cc_quad_bcast(float v) { return __shfl(v, Q, 4); }
#endif

// this lane's bit of a wave-uniform 64-bit mask, as a condition (the mask stays in scalar registers: a select on it is one
// v_cndmask with the register pair as its condition)
#ifndef CC_EMU
__device__ __forceinline__ bool cc_mask_lane(unsigned long long m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
#else
__device__ __forceinline__ bool cc_mask_lane(unsigned long long m) { return (m >> (threadIdx.x & 63)) & 1ull; }
#endif

// value of the lane D places to the left / one place to the right inside the 16-lane row; 0 beyond the row's ends
#ifndef CC_EMU
template <int D>
__device__ __forceinline__ int cc_row_shr(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xF, 0xF, true);
}
__device__ __forceinline__ int cc_row_shl1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true); }
#else
template <int D>
__device__ __forceinline__ int cc_row_shr(int v) {
  const int sl = threadIdx.x & 15;
  const int o = __shfl(v, sl >= D ? sl - D : sl, 16);
  return sl >= D ? o : 0;
}
__device__ __forceinline__ int cc_row_shl1(int v) {
  const int sl = threadIdx.x & 15;
  const int o = __shfl(v, sl < 15 ? sl + 1 : sl, 16);
  return sl < 15 ? o : 0;
}
#endif

// inclusive prefix sum over the wave's 64 lanes: a Hillis-Steele scan inside each 16-lane row on DPP row shifts (register to
// register), then the three row totals added to the rows behind them -- ~12 instructions where six ds_bpermute round trips
// (__shfl_up) are six dependent LDS-crossbar latencies
__device__ __forceinline__ int cc_wave_scan_incl(int v) {
  v += cc_row_shr<1>(v);
  v += cc_row_shr<2>(v);
  v += cc_row_shr<4>(v);
  v += cc_row_shr<8>(v);
#ifndef CC_EMU
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
#else
  const int t0 = __shfl(v, 15), t1 = __shfl(v, 31), t2 = __shfl(v, 47);
#endif
  const int lane = (int)(threadIdx.x & 63);
  return v + (lane >= 16 ? t0 : 0) + (lane >= 32 ? t1 : 0) + (lane >= 48 ? t2 : 0);
}
// the whole wave's sum of v given its inclusive scan
__device__ __forceinline__ int cc_wave_scan_total(int incl) {
#ifndef CC_EMU
  return __builtin_amdgcn_readlane(incl, 63);
#else
  return __shfl(incl, 63);
#endif
}
