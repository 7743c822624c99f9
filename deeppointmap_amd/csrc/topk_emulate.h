// Step-by-step replay of what torch.topk's CPU kernel does to a row (aten/src/ATen/native/cpu/TopKImpl.h): the row
// becomes (value, index) pairs in index order, and the first k after
//     std::partial_sort  (k * 64 <= n; libstdc++: __heap_select)          or
//     std::nth_element   (otherwise;   libstdc++: __introselect, with its heap-select fallback)
// are the answer -- with comp(a, b) = a.value > b.value for largest=True and a.value < b.value for largest=False.
// When values tie across the k-th place the chosen SET depends on these algorithms' data movement, so the kernels that
// must agree with the reference on tied rows (initial Kabsch inliers, neighbour queries) run this on one thread for
// exactly those rows.  Validated against torch.topk on 1429 tie-heavy arrays (tests/golden/make_golden_r2.py cases
// and scripts of round 2); inputs never hold NaNs here.
#pragma once

struct VI {
    float v;
    int i;
};

template <bool LARGEST>
__device__ __forceinline__ bool vi_before(const VI &a, const VI &b) {
    return LARGEST ? a.v > b.v : a.v < b.v;
}

template <bool LARGEST>
__device__ void vi_adjust_heap(VI *a, int hole, int len, VI value) {  // std::__adjust_heap + __push_heap
    const int top = hole;
    int sc = hole;
    while (sc < (len - 1) / 2) {
        sc = 2 * (sc + 1);
        if (vi_before<LARGEST>(a[sc], a[sc - 1])) --sc;
        a[hole] = a[sc], hole = sc;
    }
    if ((len & 1) == 0 && sc == (len - 2) / 2) {
        sc = 2 * (sc + 1);
        a[hole] = a[sc - 1], hole = sc - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && vi_before<LARGEST>(a[parent], value)) a[hole] = a[parent], hole = parent, parent = (hole - 1) / 2;
    a[hole] = value;
}
template <bool LARGEST>
__device__ void vi_heap_select(VI *a, int first, int middle, int last) {  // std::__heap_select
    const int len = middle - first;
    if (len >= 2)
        for (int parent = (len - 2) / 2;; --parent) {
            vi_adjust_heap<LARGEST>(a + first, parent, len, a[first + parent]);
            if (parent == 0) break;
        }
    for (int i = middle; i < last; ++i)
        if (vi_before<LARGEST>(a[i], a[first])) {
            const VI value = a[i];
            a[i] = a[first];
            vi_adjust_heap<LARGEST>(a + first, 0, len, value);
        }
}
template <bool LARGEST>
__device__ void vi_nth_element(VI *a, int n, int nth) {  // std::nth_element -> std::__introselect
    int first = 0, last = n;
    int depth = 2 * (31 - __builtin_clz(n));
    auto sw = [&](int x, int y) {
        const VI t = a[x];
        a[x] = a[y], a[y] = t;
    };
    while (last - first > 3) {
        if (depth == 0) {
            vi_heap_select<LARGEST>(a, first, nth + 1, last);
            sw(first, nth);
            return;
        }
        --depth;
        const int A = first + 1, B = first + (last - first) / 2, C = last - 1;  // __move_median_to_first
        if (vi_before<LARGEST>(a[A], a[B])) {
            if (vi_before<LARGEST>(a[B], a[C])) sw(first, B);
            else if (vi_before<LARGEST>(a[A], a[C])) sw(first, C);
            else sw(first, A);
        } else if (vi_before<LARGEST>(a[A], a[C])) sw(first, A);
        else if (vi_before<LARGEST>(a[B], a[C])) sw(first, C);
        else sw(first, B);
        const VI pivot = a[first];  // __unguarded_partition(first + 1, last, first)
        int lo = first + 1, hi = last;
        while (true) {
            while (vi_before<LARGEST>(a[lo], pivot)) ++lo;
            --hi;
            while (vi_before<LARGEST>(pivot, a[hi])) --hi;
            if (!(lo < hi)) break;
            sw(lo, hi);
            ++lo;
        }
        if (lo <= nth) first = lo;
        else last = lo;
    }
    for (int i = first + 1; i < last; ++i) {  // __insertion_sort
        const VI val = a[i];
        if (vi_before<LARGEST>(val, a[first])) {
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = val;
        } else {
            int j = i;
            while (vi_before<LARGEST>(val, a[j - 1])) a[j] = a[j - 1], --j;
            a[j] = val;
        }
    }
}

