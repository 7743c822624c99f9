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
__host__ __device__ __forceinline__ bool vi_before(const VI &a, const VI &b) {
    return LARGEST ? a.v > b.v : a.v < b.v;
}

template <bool LARGEST>
__host__ __device__ void vi_adjust_heap(VI *a, int hole, int len, VI value) {  // std::__adjust_heap + __push_heap
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
__host__ __device__ void vi_heap_select(VI *a, int first, int middle, int last) {  // std::__heap_select
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
__host__ __device__ void vi_nth_element(VI *a, int n, int nth) {  // std::nth_element -> std::__introselect
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


// ---- the same std::nth_element replay, executed by ONE WAVE over LDS.  The data movement is libstdc++'s, but a whole
// __unguarded_partition round runs in parallel: its up-scan stops at the positions p (ascending) with
// !before(a[p], pivot), its down-scan at the positions q (descending) with !before(pivot, a[q]); the k-th up-stop is
// swapped with the k-th down-stop as long as it lies left of it, neither scan ever revisits a swapped slot, and before
// the scans cross every up-stop is left of every down-stop -- so the stop lists follow from the array as it was when
// the round began, the swaps touch disjoint slots, and the cut is the first up-stop that is not left of its partner
// (or the slot the last swap moved an up-stop to, whichever comes first).
// Per round: one pass that compacts the two stop lists (ballot + popcount), one pass that swaps.  Control flow is
// wave-uniform.  `up` / `dn` are scratch for n 16-bit positions each (n <= 65535).
using LdsVI = __attribute__((address_space(3))) VI *;
using LdsU16 = __attribute__((address_space(3))) unsigned short *;

template <bool LARGEST>
__device__ void vi_nth_element_wave(LdsVI a, int n, int nth, LdsU16 up, LdsU16 dn) {
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    auto fence = [] { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); };
    auto ld = [&](int i) {  // uniform read
        VI x;
        x.v = a[i].v, x.i = a[i].i;
        return x;
    };
    auto sw = [&](int x, int y) {
        if (lane == 0) {
            const float tv = a[x].v;
            const int ti = a[x].i;
            a[x].v = a[y].v, a[x].i = a[y].i;
            a[y].v = tv, a[y].i = ti;
        }
        fence();
    };
    int first = 0, last = n;
    int depth = 2 * (31 - __builtin_clz(n));
    while (last - first > 3) {
        if (depth == 0) {  // introselect's fallback: sequential (adversarial inputs only)
            if (lane == 0) {
                VI *g = (VI *)a;
                vi_heap_select<LARGEST>(g, first, nth + 1, last);
                const VI t = g[first];
                g[first] = g[nth], g[nth] = t;
            }
            fence();
            return;
        }
        --depth;
        const int A = first + 1, B = first + (last - first) / 2, C = last - 1;  // __move_median_to_first
        const VI vA = ld(A), vB = ld(B), vC = ld(C);
        if (vi_before<LARGEST>(vA, vB)) {
            if (vi_before<LARGEST>(vB, vC)) sw(first, B);
            else if (vi_before<LARGEST>(vA, vC)) sw(first, C);
            else sw(first, A);
        } else if (vi_before<LARGEST>(vA, vC)) sw(first, A);
        else if (vi_before<LARGEST>(vB, vC)) sw(first, C);
        else sw(first, B);
        const VI pivot = ld(first);  // __unguarded_partition(first + 1, last, first)
        // stop lists of the range [first + 1, last): ascending up-stops, descending down-stops
        int nu = 0, nd = 0;
        for (int base = first + 1; base < last; base += 64) {
            const int i = base + lane, j = last - 1 - (base - (first + 1)) - lane;  // i walks up, j walks down
            VI x, y;
            x.v = a[min(i, last - 1)].v, y.v = a[max(j, first + 1)].v;
            const unsigned long long mu = __ballot(i < last && !vi_before<LARGEST>(x, pivot));
            const unsigned long long md = __ballot(j > first && !vi_before<LARGEST>(pivot, y));
            if ((mu >> lane) & 1ull) up[nu + __popcll(mu & below)] = (unsigned short)i;
            if ((md >> lane) & 1ull) dn[nd + __popcll(md & below)] = (unsigned short)j;
            nu += __popcll(mu), nd += __popcll(md);
        }
        fence();
        // pair k swaps while up[k] < dn[k]; the cut is up[k*] of the first pair that does not (it exists: the scans of the
        // sequential form always stop inside the range)
        int cut = -1;
        for (int k0 = 0; cut < 0; k0 += 64) {
            const int k = k0 + lane;
            const bool have = k < nu && k < nd;
            const int u = have ? up[k] : 0x7fffffff, d = have ? dn[k] : -1;
            const bool go = have && u < d;
            if (go) {
                const float tv = a[u].v;
                const int ti = a[u].i;
                a[u].v = a[d].v, a[u].i = a[d].i;
                a[d].v = tv, a[d].i = ti;
            }
            const unsigned long long stop = __ballot(!go);
            if (stop) {
                // the sequential up-scan would also stop on the up-stop that pair ks-1 just moved to dn[ks-1]
                const int ks = k0 + __builtin_ctzll(stop);
                const int c1 = ks < nu ? (int)up[ks] : 0x7fffffff, c2 = ks > 0 ? (int)dn[ks - 1] : 0x7fffffff;
                cut = min(c1, c2);
            }
        }
        fence();
        if (cut <= nth) first = cut;
        else last = cut;
    }
    if (lane == 0) {  // __insertion_sort of the last <= 3 elements
        for (int i = first + 1; i < last; ++i) {
            VI val;
            val.v = a[i].v, val.i = a[i].i;
            VI f0;
            f0.v = a[first].v;
            if (vi_before<LARGEST>(val, f0)) {
                for (int j = i; j > first; --j) a[j].v = a[j - 1].v, a[j].i = a[j - 1].i;
                a[first].v = val.v, a[first].i = val.i;
            } else {
                int j = i;
                while (true) {
                    VI p;
                    p.v = a[j - 1].v;
                    if (!vi_before<LARGEST>(val, p)) break;
                    a[j].v = a[j - 1].v, a[j].i = a[j - 1].i, --j;
                }
                a[j].v = val.v, a[j].i = val.i;
            }
        }
    }
    fence();
}


// ---- torch.topk(sorted=True) in full: not only WHICH k elements, but the ORDER they come out in (TopKImpl.h: after
// std::partial_sort the first k ARE sorted -- __heap_select + __sort_heap --; on the nth_element branch the first k - 1
// are sorted by std::sort -- __introsort_loop + __final_insertion_sort -- and the k-th stays where nth_element left it).
// Neither sort is stable, so among equal values the order is again a matter of libstdc++'s data movement.  The voxel
// sampler needs it: it returns points in the order of a topk over integer voxel populations, which tie all the time.
// One thread; the caller's VI row is permuted in place and its first k entries are the answer.
template <bool LARGEST>
__host__ __device__ void vi_sort_heap(VI *a, int first, int last) {  // std::__sort_heap (via __pop_heap)
    while (last - first > 1) {
        --last;
        const VI value = a[last];
        a[last] = a[first];
        vi_adjust_heap<LARGEST>(a + first, 0, last - first, value);
    }
}
template <bool LARGEST>
__host__ __device__ void vi_unguarded_linear_insert(VI *a, int last) {
    const VI val = a[last];
    int next = last - 1;
    while (vi_before<LARGEST>(val, a[next])) a[last] = a[next], last = next, --next;
    a[last] = val;
}
template <bool LARGEST>
__host__ __device__ void vi_insertion_sort(VI *a, int first, int last) {  // std::__insertion_sort
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (vi_before<LARGEST>(a[i], a[first])) {
            const VI val = a[i];
            for (int j = i; j > first; --j) a[j] = a[j - 1];
            a[first] = val;
        } else vi_unguarded_linear_insert<LARGEST>(a, i);
    }
}
template <bool LARGEST>
__host__ __device__ void vi_sort(VI *a, int first, int last) {  // std::sort -> std::__sort
    if (first == last) return;
    // __introsort_loop; its recursion on the right part becomes an explicit stack (depth <= 2 lg n <= 64)
    int stack_first[64], stack_last[64], stack_depth[64], sp = 0;
    int lo0 = first, hi0 = last, depth = 2 * (31 - __builtin_clz((unsigned)(last - first)));
    while (true) {
        while (hi0 - lo0 > 16) {
            if (depth == 0) {  // __partial_sort(first, last, last)
                vi_heap_select<LARGEST>(a, lo0, hi0, hi0);
                vi_sort_heap<LARGEST>(a, lo0, hi0);
                break;
            }
            --depth;
            auto sw = [&](int x, int y) {
                const VI t = a[x];
                a[x] = a[y], a[y] = t;
            };
            const int A = lo0 + 1, B = lo0 + (hi0 - lo0) / 2, C = hi0 - 1;  // __move_median_to_first(first, ...)
            if (vi_before<LARGEST>(a[A], a[B])) {
                if (vi_before<LARGEST>(a[B], a[C])) sw(lo0, B);
                else if (vi_before<LARGEST>(a[A], a[C])) sw(lo0, C);
                else sw(lo0, A);
            } else if (vi_before<LARGEST>(a[A], a[C])) sw(lo0, A);
            else if (vi_before<LARGEST>(a[B], a[C])) sw(lo0, C);
            else sw(lo0, B);
            const VI pivot = a[lo0];  // __unguarded_partition(first + 1, last, first)
            int lo = lo0 + 1, hi = hi0;
            while (true) {
                while (vi_before<LARGEST>(a[lo], pivot)) ++lo;
                --hi;
                while (vi_before<LARGEST>(pivot, a[hi])) --hi;
                if (!(lo < hi)) break;
                sw(lo, hi);
                ++lo;
            }
            // __introsort_loop(cut, last, depth) runs to completion BEFORE the left part continues: the order of the two
            // does not matter to the result (disjoint ranges), so the right part is parked on the stack
            stack_first[sp] = lo, stack_last[sp] = hi0, stack_depth[sp] = depth, ++sp;
            hi0 = lo;
        }
        if (sp == 0) break;
        --sp, lo0 = stack_first[sp], hi0 = stack_last[sp], depth = stack_depth[sp];
    }
    if (last - first > 16) {  // __final_insertion_sort
        vi_insertion_sort<LARGEST>(a, first, first + 16);
        for (int i = first + 16; i != last; ++i) vi_unguarded_linear_insert<LARGEST>(a, i);
    } else vi_insertion_sort<LARGEST>(a, first, last);
}
// the whole of topk_impl_loop for one row of n (value, index) pairs, sorted=True; 0 < k <= n
template <bool LARGEST>
__host__ __device__ void vi_topk_sorted(VI *a, int n, int k) {
    if ((long long)k * 64 <= n) {  // std::partial_sort(begin, begin + k, end)
        vi_heap_select<LARGEST>(a, 0, k, n);
        vi_sort_heap<LARGEST>(a, 0, k);
    } else {
        vi_nth_element<LARGEST>(a, n, k - 1);
        vi_sort<LARGEST>(a, 0, k - 1);
    }
}
