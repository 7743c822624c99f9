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
