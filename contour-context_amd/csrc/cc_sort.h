// Device-side replica of libstdc++'s std::sort (introsort: median-of-3 quicksort down to 16
// elements, heapsort when the depth limit 2*floor(log2 n) is hit, final insertion sort).
//
// Why: the reference orders contours, BCI neighbours and constellation pairs with *unstable*
// std::sort calls (contour_mng.h:340, :596-599, :871; contour_db.h:616,630;
// src/cont2/contour_db.cpp:370), so which of two equal keys comes first is a property of the
// libstdc++ algorithm.  To produce bit-identical contour numbering on the GPU the same sequence of
// comparisons and moves has to be executed; this header restates bits/stl_algo.h / stl_heap.h
// (GCC 9-13, unchanged there) for a random-access array, one lane per array.
#pragma once
#include "cc_group.h"
#include <hip/hip_runtime.h>

namespace ccsort {

template <typename T, typename Less>
__device__ __forceinline__ void unguarded_linear_insert(T *a, int last, Less less) {
  T val = a[last];
  int next = last - 1;
  while (less(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <typename T, typename Less>
__device__ __forceinline__ void insertion_sort(T *a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      T val = a[i];
      for (int k = i; k > first; --k) a[k] = a[k - 1];  // move_backward(first, i, i+1)
      a[first] = val;
    } else {
      unguarded_linear_insert(a, i, less);
    }
  }
}

template <typename T, typename Less>
__device__ __forceinline__ void push_heap(T *a, int first, int holeIndex, int topIndex, T value, Less less) {
  int parent = (holeIndex - 1) / 2;
  while (holeIndex > topIndex && less(a[first + parent], value)) {
    a[first + holeIndex] = a[first + parent];
    holeIndex = parent;
    parent = (holeIndex - 1) / 2;
  }
  a[first + holeIndex] = value;
}

template <typename T, typename Less>
__device__ __forceinline__ void adjust_heap(T *a, int first, int holeIndex, int len, T value, Less less) {
  const int topIndex = holeIndex;
  int secondChild = holeIndex;
  while (secondChild < (len - 1) / 2) {
    secondChild = 2 * (secondChild + 1);
    if (less(a[first + secondChild], a[first + (secondChild - 1)])) secondChild--;
    a[first + holeIndex] = a[first + secondChild];
    holeIndex = secondChild;
  }
  if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
    secondChild = 2 * (secondChild + 1);
    a[first + holeIndex] = a[first + (secondChild - 1)];
    holeIndex = secondChild - 1;
  }
  push_heap(a, first, holeIndex, topIndex, value, less);
}

// __partial_sort(first, last, last) == make_heap + sort_heap
template <typename T, typename Less>
__device__ __forceinline__ void heap_sort(T *a, int first, int last, Less less) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    while (true) {
      T value = a[first + parent];
      adjust_heap(a, first, parent, len, value, less);
      if (parent == 0) break;
      parent--;
    }
  }
  int l = last;
  while (l - first > 1) {
    --l;
    T value = a[l];
    a[l] = a[first];
    adjust_heap(a, first, 0, l - first, value, less);
  }
}

template <typename T, typename Less>
__device__ __forceinline__ int unguarded_partition_pivot(T *a, int first, int last, Less less) {
  const int mid = first + (last - first) / 2;
  // __move_median_to_first(first, first+1, mid, last-1)
  const int ia = first + 1, ib = mid, ic = last - 1;
  int sel;
  if (less(a[ia], a[ib])) {
    if (less(a[ib], a[ic]))
      sel = ib;
    else if (less(a[ia], a[ic]))
      sel = ic;
    else
      sel = ia;
  } else if (less(a[ia], a[ic]))
    sel = ia;
  else if (less(a[ib], a[ic]))
    sel = ic;
  else
    sel = ib;
  {
    T t = a[first];
    a[first] = a[sel];
    a[sel] = t;
  }
  // __unguarded_partition(first+1, last, pivot = first)
  int f = first + 1, l = last;
  while (true) {
    while (less(a[f], a[first])) ++f;
    --l;
    while (less(a[first], a[l])) --l;
    if (!(f < l)) return f;
    T t = a[f];
    a[f] = a[l];
    a[l] = t;
    ++f;
  }
}

// std::sort(a, a + n, less).  n < 4096.  `stk`: CC_SORT_STACK words of caller-provided storage (LDS; one per sorting
// lane) for the pending (cut, last) halves of __introsort_loop -- at most 2*floor(log2 n) + 1 of them, packed as
// first | last << 12 | depth << 24 -- so that no lane-indexed array ends up in scratch memory.
#define CC_SORT_STACK 26
template <typename T, typename Less>
__device__ void std_sort(T *a, int n, Less less, unsigned *stk) {
  if (n <= 0) return;
  int sp = 0;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  stk[0] = 0u | ((unsigned)n << 12) | ((unsigned)(lg * 2) << 24);
  sp = 1;
  while (sp > 0) {
    --sp;
    const unsigned w_ = stk[sp];
    int first = (int)(w_ & 0xFFFu), last = (int)((w_ >> 12) & 0xFFFu), depth = (int)(w_ >> 24);
    // The reference recursion is: loop { cut = partition; introsort_loop(cut, last); last = cut; }
    // i.e. the RIGHT part is fully processed before the left part continues.  Disjoint ranges are
    // independent, so the order in which they are processed does not change the result.
    while (last - first > 16) {
      if (depth == 0) {
        heap_sort(a, first, last, less);
        break;
      }
      --depth;
      int cut = unguarded_partition_pivot(a, first, last, less);
      stk[sp] = (unsigned)cut | ((unsigned)last << 12) | ((unsigned)depth << 24);
      ++sp;
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    insertion_sort(a, 0, 16, less);
    for (int i = 16; i != n; ++i) unguarded_linear_insert(a, i, less);
  } else {
    insertion_sort(a, 0, n, less);
  }
}

// same, with the pending-halves stack in the lane's private memory (callers outside the hot kernels)
template <typename T, typename Less>
__device__ void std_sort(T *a, int n, Less less) {
  unsigned stk[CC_SORT_STACK];
  std_sort(a, n, less, stk);
}


// std::sort replayed by ONE WAVE on an array in LDS -- what a lane does alone above, for up to 4095 32-bit elements ordered
// by an unsigned key (`ukey(x) < ukey(y)` is the comparator's "x before y"):
//   (1) the median-of-3 Hoare partitions of __introsort_loop until every segment has <= 16 elements.  One partition is
//       data-parallel: with the positions of the elements !(x < pivot) in ascending order (l_k) and of the elements
//       !(pivot < x) in descending order (r_k), the sequential two-pointer loop swaps exactly the pairs (l_k, r_k) with
//       l_k < r_k -- a prefix k < K -- and returns min(l_K, r_(K-1)): neither pointer re-reads a swapped position before they
//       cross (the same argument as stage B1's replay, k_check.h);
//   (2) __final_insertion_sort is a STABLE sort of what (1) left: rank = #smaller keys + #equal keys at earlier positions.
// The heapsort branch (depth limit exhausted) is replayed serially by lane 0 on the input regenerated by `regen()`.
// lpos / rasc: n u16 each; tmp: n words (may alias lpos / rasc: used after the partitions); seg: CC_SORT_STACK words.
template <typename UKey, typename Regen>
__device__ __forceinline__ void std_sort_wave(unsigned *a, int n, UKey ukey, Regen regen, int lane, unsigned short *lpos, unsigned short *rasc,
                                              unsigned *tmp, unsigned *seg) {
  bool deep = false;
  cc_wave_sync();
  if (n > 16) {
    int lg = 0;
    for (int t = n; t > 1; t >>= 1) lg++;
    int nseg = 1;
    if (lane == 0) seg[0] = 0u | ((unsigned)n << 12) | ((unsigned)(lg * 2) << 24);
    cc_wave_sync();
    while (nseg > 0) {
      nseg--;
      const unsigned w_ = seg[nseg];
      const int first = (int)(w_ & 0xFFFu), last = (int)((w_ >> 12) & 0xFFFu);
      int depth = (int)(w_ >> 24);
      if (depth == 0) {
        deep = true;
        break;
      }
      depth--;
      const int mid = first + (last - first) / 2;
      const int ia = first + 1, ib = mid, ic = last - 1;
      const unsigned ka = ukey(a[ia]), kb = ukey(a[ib]), kc = ukey(a[ic]);
      int sel;  // __move_median_to_first(first, first+1, mid, last-1)
      if (ka < kb) {
        if (kb < kc)
          sel = ib;
        else if (ka < kc)
          sel = ic;
        else
          sel = ia;
      } else if (ka < kc)
        sel = ia;
      else if (kb < kc)
        sel = ic;
      else
        sel = ib;
      cc_wave_sync();
      if (lane == 0) {
        const unsigned t = a[first];
        a[first] = a[sel];
        a[sel] = t;
      }
      cc_wave_sync();
      const unsigned piv = ukey(a[first]);
      const unsigned long long lt = (1ull << lane) - 1ull;
      int nL = 0, nR = 0;
      for (int r0 = first + 1; r0 < last; r0 += 64) {
        const int i = r0 + lane;
        bool ls = false, rs = false;
        if (i < last) {
          const unsigned k = ukey(a[i]);
          ls = k >= piv;
          rs = k <= piv;
        }
        const unsigned long long mL = __ballot(ls), mR = __ballot(rs);
        if (ls) lpos[nL + __popcll(mL & lt)] = (unsigned short)i;
        if (rs) rasc[nR + __popcll(mR & lt)] = (unsigned short)i;
        nL += __popcll(mL);
        nR += __popcll(mR);
      }
      cc_wave_sync();
      const int nmin = nL < nR ? nL : nR;
      int K = 0;
      for (int k0 = 0; k0 < nmin; k0 += 64) {
        const int k = k0 + lane;
        K += __popcll(__ballot(k < nmin && lpos[k] < rasc[nR - 1 - k]));
      }
      for (int k = lane; k < K; k += 64) {
        const int x = lpos[k], y = rasc[nR - 1 - k];
        const unsigned t = a[x];
        a[x] = a[y];
        a[y] = t;
      }
      int cut = 0x7fff;
      if (K < nL) cut = lpos[K];
      if (K > 0 && (int)rasc[nR - K] < cut) cut = rasc[nR - K];
      cc_wave_sync();
      if (last - cut > 16) {
        if (lane == 0) seg[nseg] = (unsigned)cut | ((unsigned)last << 12) | ((unsigned)depth << 24);
        nseg++;
      }
      if (cut - first > 16) {
        if (lane == 0) seg[nseg] = (unsigned)first | ((unsigned)cut << 12) | ((unsigned)depth << 24);
        nseg++;
      }
      cc_wave_sync();
    }
  }
  if (deep) {  // rare: the serial replay on the pristine input
    cc_wave_sync();
    regen();
    cc_wave_sync();
    if (lane == 0) std_sort(a, n, [&](unsigned x, unsigned y) { return ukey(x) < ukey(y); }, seg);
    cc_wave_sync();
    return;
  }
  // stable rank of what the partitions left
  for (int i0 = 0; i0 < n; i0 += 64) {
    const int i = i0 + lane;
    if (i < n) {
      const unsigned v = a[i];
      const unsigned long long me = ((unsigned long long)ukey(v) << 32) | (unsigned)i;
      int rank = 0;
      for (int j = 0; j < n; j++) rank += ((((unsigned long long)ukey(a[j]) << 32) | (unsigned)j) < me) ? 1 : 0;
      tmp[rank] = v;
    }
  }
  cc_wave_sync();
  for (int i = lane; i < n; i += 64) a[i] = tmp[i];
  cc_wave_sync();
}

}  // namespace ccsort
