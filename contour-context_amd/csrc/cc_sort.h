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

}  // namespace ccsort
