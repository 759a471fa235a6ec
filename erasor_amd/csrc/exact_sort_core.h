// exact_sort_core.h — the order-defining pieces of libstdc++'s std::sort, written so that a
// partition step can be executed by many lanes at once and still leave the array in EXACTLY the
// state the sequential algorithm leaves it in.
//
// Why this exists: the reference sorts with an *unstable* std::sort in two places whose tie order
// feeds float32 running sums — pcl::VoxelGrid's index sort (reference erasor_utils.cpp:88-91 ->
// PCL voxel_grid.hpp) and R-GPF's z-sort (reference erasor.cpp:240).  Bit-exact centroids / plane
// fits therefore need the very permutation libstdc++ produces, not just "a" sorted order.
//
// libstdc++ std::sort(first,last,comp)  (bits/stl_algo.h):
//   __introsort_loop(first,last, 2*floor(log2(n)));  __final_insertion_sort(first,last);
//   __introsort_loop: while (last-first > 16) { if (depth==0) { heapsort(first,last); return; }
//       --depth; cut = __unguarded_partition_pivot(first,last); __introsort_loop(cut,last,depth); last = cut; }
//   __unguarded_partition_pivot: mid = first+(last-first)/2;
//       __move_median_to_first(first, first+1, mid, last-1); return __unguarded_partition(first+1,last,first);
//   __unguarded_partition(first,last,pivot): for(;;){ while(*first<*pivot)++first; --last;
//       while(*pivot<*last)--last; if(!(first<last)) return first; iter_swap(first,last); ++first; }
//   __final_insertion_sort == a stable insertion sort of what the loop left behind.
//
// Parallel formulation of one partition of [first,last), pivot p = A[first] after the median move,
// range [lo=first+1, hi=last):
//   L = positions i (ascending) with !(A[i] < p);  R = positions j (descending) with !(p < A[j]).
//   Both scanners only ever look at *original* elements (a swapped element is revisited only at the
//   crossing, which terminates), so:  swaps = pairs (L[k],R[k]) for k < m, m = #{k : L[k] < R[k]}
//   (a monotone predicate), and  cut = min( L[m] if it exists, R[m-1] if m > 0 ).
//   The median move guarantees L and R are non-empty.
// Because every element left of `cut` is <= p <= every element right of it, the final insertion
// sort equals an independent *stable* sort of every leaf segment (size <= 16).
//
// The same code compiles for the host (tests/cpp/esort_check.cpp pins it against the real
// std::sort) and for the device (exact_sort.hip.h drives it with wavefronts).
#ifndef ERASOR_EXACT_SORT_CORE_H
#define ERASOR_EXACT_SORT_CORE_H

#include <stdint.h>

#if defined(__HIPCC__)
#define ESORT_HD __host__ __device__ inline
#else
#define ESORT_HD inline
#endif

namespace esort {

static constexpr int kThreshold = 16;  // _S_threshold

ESORT_HD int lg2_floor(uint32_t n) {  // std::__lg
    int r = 0;
    while (n >>= 1) ++r;
    return r;
}

template <class KeyPtr, class ValPtr>
ESORT_HD void swap_kv(KeyPtr K, ValPtr V, uint32_t a, uint32_t b) {
    uint32_t tk = K[a];
    K[a] = K[b];
    K[b] = tk;
    uint32_t tv = V[a];
    V[a] = V[b];
    V[b] = tv;
}

// std::__move_median_to_first(result=first, a=first+1, b=mid, c=last-1) with operator< on keys
template <class KeyPtr, class ValPtr>
ESORT_HD void move_median_to_first(KeyPtr K, ValPtr V, uint32_t first, uint32_t last) {
    const uint32_t a = first + 1, b = first + (last - first) / 2, c = last - 1;
    const uint32_t ka = K[a], kb = K[b], kc = K[c];
    uint32_t m;
    if (ka < kb) {
        if (kb < kc)
            m = b;
        else if (ka < kc)
            m = c;
        else
            m = a;
    } else if (ka < kc)
        m = a;
    else if (kb < kc)
        m = c;
    else
        m = b;
    swap_kv(K, V, first, m);
}

// cut from the stop lists.  posL[k]: k-th left stop (ascending).  posRasc[k]: k-th right stop in
// ASCENDING order (so R[k] = posRasc[nR-1-k]).  m = number of swaps.
template <class PosPtr>
ESORT_HD uint32_t cut_from_lists(PosPtr posL, PosPtr posRasc, uint32_t nL, uint32_t nR, uint32_t m) {
    uint32_t cut = 0xFFFFFFFFu;
    if (m < nL) cut = posL[m];
    if (m > 0) {
        const uint32_t r = posRasc[nR - m];  // R[m-1]
        if (r < cut) cut = r;
    }
    return cut;
}

// ---- heapsort fallback (std::__partial_sort(first,last,last)), sequential, exact ----------------
template <class KeyPtr, class ValPtr>
ESORT_HD void push_heap_(KeyPtr K, ValPtr V, uint32_t first, int64_t hole, int64_t top, uint32_t vk, uint32_t vv) {
    int64_t parent = (hole - 1) / 2;
    while (hole > top && K[first + parent] < vk) {
        K[first + hole] = K[first + parent];
        V[first + hole] = V[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    K[first + hole] = vk;
    V[first + hole] = vv;
}
template <class KeyPtr, class ValPtr>
ESORT_HD void adjust_heap_(KeyPtr K, ValPtr V, uint32_t first, int64_t hole, int64_t len, uint32_t vk, uint32_t vv) {
    const int64_t top = hole;
    int64_t second = hole;
    while (second < (len - 1) / 2) {
        second = 2 * (second + 1);
        if (K[first + second] < K[first + (second - 1)]) second--;
        K[first + hole] = K[first + second];
        V[first + hole] = V[first + second];
        hole = second;
    }
    if ((len & 1) == 0 && second == (len - 2) / 2) {
        second = 2 * (second + 1);
        K[first + hole] = K[first + (second - 1)];
        V[first + hole] = V[first + (second - 1)];
        hole = second - 1;
    }
    push_heap_(K, V, first, hole, top, vk, vv);
}
template <class KeyPtr, class ValPtr>
ESORT_HD void heapsort_exact(KeyPtr K, ValPtr V, uint32_t first, uint32_t last) {
    const int64_t len = (int64_t)last - (int64_t)first;
    if (len < 2) return;
    // __make_heap
    int64_t parent = (len - 2) / 2;
    for (;;) {
        const uint32_t vk = K[first + parent], vv = V[first + parent];
        adjust_heap_(K, V, first, parent, len, vk, vv);
        if (parent == 0) break;
        parent--;
    }
    // __heap_select's loop over [middle,last) is empty (middle == last).  __sort_heap:
    int64_t l = len;
    while (l > 1) {
        --l;
        // __pop_heap(first, first+l, first+l)
        const uint32_t vk = K[first + l], vv = V[first + l];
        K[first + l] = K[first];
        V[first + l] = V[first];
        adjust_heap_(K, V, first, 0, l, vk, vv);
    }
}

// stable rank of element i inside its leaf [a,b): final position = a + rank
template <class KeyPtr>
ESORT_HD uint32_t leaf_rank(KeyPtr K, uint32_t a, uint32_t b, uint32_t i) {
    const uint32_t ki = K[i];
    uint32_t r = 0;
    for (uint32_t j = a; j < b; ++j) {
        const uint32_t kj = K[j];
        r += (kj < ki) || (kj == ki && j < i);
    }
    return r;
}

// order-preserving float -> uint32 key for `a.z < b.z` (reference erasor.cpp:200-202):
// -0.0f and +0.0f compare equal under operator<, so both map to the same key.
ESORT_HD uint32_t float_key(uint32_t bits) {
    if ((bits << 1) == 0u) bits = 0u;  // -0 -> +0
    return (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
}

}  // namespace esort
#endif
