// Benjamini-Hochberg FDR on the device, one segment per block (reference mustache/mustache.py:778:
// statsmodels.stats.multitest.multipletests(p, method='fdr_bh')).
//   sort p ascending (hipCUB segmented radix sort, keys = p, values = record index)
//   adj[i] = p_sorted[i] / ((i + 1) / m)            -- the same two IEEE divisions statsmodels / NumPy perform
//   q_sorted[i] = min(adj[i..m-1]), clipped at 1     -- min is exact, so the parallel suffix-min is bit-identical
//   q[record] = q_sorted[rank(record)]
// Ties in p receive equal q, so the (unstable) order among equal keys does not matter.
#include <hipcub/hipcub.hpp>
#include <cmath>
#include <vector>
#include "mst_common.h"

namespace {

constexpr int kBH = 1024;

__global__ void __launch_bounds__(256)
bh_setup_kernel(const uint32_t *__restrict__ count, uint32_t cap, int B, int *__restrict__ seg_begin,
                int *__restrict__ seg_end, uint32_t *__restrict__ idx) {
    const int b = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t m = count[b] < cap ? count[b] : cap;
        seg_begin[b] = (int)((size_t)b * cap);
        seg_end[b] = (int)((size_t)b * cap + m);
    }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += gridDim.x * blockDim.x)
        idx[(size_t)b * cap + i] = i;
}

__global__ void __launch_bounds__(kBH)
bh_adjust_kernel(const double *__restrict__ ps, const uint32_t *__restrict__ idx_sorted,
                 const uint32_t *__restrict__ count, uint32_t cap, double *__restrict__ q) {
    __shared__ double chunk_min[kBH];
    const int b = blockIdx.x, t = threadIdx.x;
    const uint32_t m = count[b] < cap ? count[b] : cap;
    if (m == 0) return;
    const double *p = ps + (size_t)b * cap;
    const uint32_t *id = idx_sorted + (size_t)b * cap;
    double *qb = q + (size_t)b * cap;
    const uint32_t L = (m + kBH - 1) / kBH;
    const uint32_t lo = t * L < m ? t * L : m, hi = (t + 1) * L < m ? (t + 1) * L : m;
    const double dm = (double)m;
    double mn = INFINITY;
    for (uint32_t i = lo; i < hi; ++i) {
        const double adj = p[i] / ((double)(i + 1) / dm);
        mn = adj < mn ? adj : mn;
    }
    chunk_min[t] = mn;
    __syncthreads();
    // suffix minimum over the chunks (inclusive), log-step scan
    for (int off = 1; off < kBH; off <<= 1) {
        const double other = (t + off < kBH) ? chunk_min[t + off] : INFINITY;
        __syncthreads();
        if (other < chunk_min[t]) chunk_min[t] = other;
        __syncthreads();
    }
    double run = (t + 1 < kBH) ? chunk_min[t + 1] : INFINITY;      // minimum over all later chunks
    for (uint32_t i = hi; i > lo; --i) {
        const double adj = p[i - 1] / ((double)i / dm);
        run = adj < run ? adj : run;
        qb[id[i - 1]] = run > 1.0 ? 1.0 : run;
    }
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t sort_temp_bytes(int B, uint32_t cap) {
    size_t bytes = 0;
    (void)hipcub::DeviceSegmentedRadixSort::SortPairs(nullptr, bytes, (const double *)nullptr, (double *)nullptr,
                                               (const uint32_t *)nullptr, (uint32_t *)nullptr, (int)((size_t)B * cap), B,
                                               (const int *)nullptr, (const int *)nullptr, 0, 64, (hipStream_t)0);
    return bytes;
}


// selection o < pt (mustache.py:789-797): the found records whose q-value is below the threshold, compacted per block.
// Only these (a few hundred per block) are ever looked at by the filters and the clustering -- a not-found pixel has
// o >= 1, and a found pixel with q >= pt can neither be a candidate nor the arg-min of a cluster that holds one.
__global__ void __launch_bounds__(256)
select_below_kernel(const mst_found *__restrict__ found, const double *__restrict__ q, const uint32_t *__restrict__ count,
                    uint32_t cap, double threshold, uint32_t out_cap, uint32_t *__restrict__ out_pixel,
                    uint32_t *__restrict__ out_level, double *__restrict__ out_q, uint32_t *__restrict__ out_count) {
    const int b = blockIdx.y;
    const uint32_t n = count[b] < cap ? count[b] : cap;
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {      // whole waves stay together
        const uint32_t i = i0 + threadIdx.x;
        const double qi = i < n ? q[(size_t)b * cap + i] : 2.0;
        const bool take = i < n && qi < threshold;
        const unsigned long long bal = __ballot(take);
        if (!bal) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(out_count + b, (uint32_t)__popcll(bal));             // one atomic per wave
        base = __shfl(base, 0, 64);
        if (take) {
            const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (slot < out_cap) {
                const mst_found r = found[(size_t)b * cap + i];
                out_pixel[(size_t)b * out_cap + slot] = r.pixel;
                out_level[(size_t)b * out_cap + slot] = r.level;
                out_q[(size_t)b * out_cap + slot] = qi;
            }
        }
    }
}

// ---- BH restricted to the records that can be selected (mst_bh_select) ------------------------------------------------
// q_(i) = min_{j >= i} p_(j) m / (j + 1) >= p_(i), so q < pt needs p < pt: only the k records with p < pt can be selected.
// They are the k smallest, so their ranks in the full sort are their ranks among themselves; and the part of the suffix
// minimum that comes from the other m - k records, T = min_{j >= k} p_(j) m / (j + 1), is >= pt (p_(j) >= pt, m/(j+1) >= 1).
// Hence for a record of the subset:  q = min(A, T) with A = the suffix minimum over the subset alone;  A < pt  =>  q = A
// exactly, and A >= pt  =>  q >= pt (not selected).  Sorting 1 % of the records instead of all of them gives the same
// selected set with bit-identical q-values (the global m is used in every division).
__global__ void __launch_bounds__(256)
compact_below_kernel(const double *__restrict__ pval, const uint32_t *__restrict__ count, uint32_t cap,
                     const double *__restrict__ threshold_of_block, double *__restrict__ keys, uint32_t *__restrict__ idx,
                     uint32_t *__restrict__ k_out) {
    const int b = blockIdx.y;
    const uint32_t n = count[b] < cap ? count[b] : cap;
    const double threshold = threshold_of_block[b];
    const int lane = threadIdx.x & 63;
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n; i0 += gridDim.x * blockDim.x) {      // whole waves stay together
        const uint32_t i = i0 + threadIdx.x;
        const double p = i < n ? pval[(size_t)b * cap + i] : 2.0;
        const bool take = i < n && p < threshold;
        const unsigned long long bal = __ballot(take);
        if (!bal) continue;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(k_out + b, (uint32_t)__popcll(bal));
        base = __shfl(base, 0, 64);
        if (take) {
            const uint32_t slot = base + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));   // < n <= cap: cannot overflow
            keys[(size_t)b * cap + slot] = p;
            idx[(size_t)b * cap + slot] = i;
        }
    }
}

__global__ void seg_bounds_kernel(const uint32_t *__restrict__ k, uint32_t cap, int B, int *__restrict__ seg_begin,
                                  int *__restrict__ seg_end) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        seg_begin[b] = (int)((size_t)b * cap);
        seg_end[b] = (int)((size_t)b * cap + k[b]);
    }
}

__global__ void __launch_bounds__(kBH)
bh_select_kernel(const double *__restrict__ ps, const uint32_t *__restrict__ idx_sorted, const uint32_t *__restrict__ k_sub,
                 const uint32_t *__restrict__ count, uint32_t cap, const mst_found *__restrict__ found, double threshold,
                 uint32_t out_cap, uint32_t *__restrict__ out_pixel, uint32_t *__restrict__ out_level,
                 double *__restrict__ out_q, uint32_t *__restrict__ out_count, uint32_t *__restrict__ out_index) {
    __shared__ double chunk_min[kBH];
    __shared__ uint32_t n_out;
    const int b = blockIdx.x, t = threadIdx.x;
    const uint32_t m = count[b] < cap ? count[b] : cap;        // the GLOBAL number of tests
    const uint32_t k = k_sub[b];                               // records with p < threshold, sorted ascending
    if (t == 0) n_out = 0;
    __syncthreads();
    if (k == 0) {
        if (t == 0) out_count[b] = 0;
        return;
    }
    const double *p = ps + (size_t)b * cap;
    const uint32_t *id = idx_sorted + (size_t)b * cap;
    const uint32_t L = (k + kBH - 1) / kBH;
    const uint32_t lo = t * L < k ? t * L : k, hi = (t + 1) * L < k ? (t + 1) * L : k;
    const double dm = (double)m;
    double mn = INFINITY;
    for (uint32_t i = lo; i < hi; ++i) {
        const double adj = p[i] / ((double)(i + 1) / dm);
        mn = adj < mn ? adj : mn;
    }
    chunk_min[t] = mn;
    __syncthreads();
    for (int off = 1; off < kBH; off <<= 1) {
        const double other = (t + off < kBH) ? chunk_min[t + off] : INFINITY;
        __syncthreads();
        if (other < chunk_min[t]) chunk_min[t] = other;
        __syncthreads();
    }
    double run = (t + 1 < kBH) ? chunk_min[t + 1] : INFINITY;
    for (uint32_t i = hi; i > lo; --i) {
        const double adj = p[i - 1] / ((double)i / dm);
        run = adj < run ? adj : run;
        const double q = run > 1.0 ? 1.0 : run;
        if (q < threshold) {
            const uint32_t slot = atomicAdd(&n_out, 1u);       // LDS atomic; the caller orders the records by pixel anyway
            if (slot < out_cap) {
                const mst_found r = found[(size_t)b * cap + id[i - 1]];
                out_pixel[(size_t)b * out_cap + slot] = r.pixel;
                out_level[(size_t)b * out_cap + slot] = r.level;
                out_q[(size_t)b * out_cap + slot] = q;
                if (out_index) out_index[(size_t)b * out_cap + slot] = id[i - 1];
            }
        }
    }
    __syncthreads();
    if (t == 0) out_count[b] = n_out;
}

// ---- how few records have to be sorted -----------------------------------------------------------------------------------
// The selected records are the ranks 1..j*, j* = the largest rank whose adjusted value p_(j) / (j / m) is below pt; they all
// have p < pt' j* / m (pt' = pt (1 + 1e-9) covers the rounding of the two divisions).  Any J >= j* therefore gives a
// threshold pt' J / m under which all of them lie, and c(t) = #{p < t} turns a threshold back into a bound:
//   J_0 = c(pt'),  J_{i+1} = c(e(pt' J_i / m)) >= j*      (e(t) = the next histogram edge >= t)
// decreases to a fixed point -- typically a few hundred records out of ~150 000, where p < pt alone keeps a third.  The
// histogram has 8 bins per binary order of magnitude down to 2^-40 (bin = top bits of the IEEE pattern, monotone for p >= 0),
// so c() is exact at the edges and one pass over the p-values suffices.  With S = {p < T}, |S| >= j*, the argument above
// (mst_bh_select) holds unchanged: ranks inside S are global ranks, and every rank beyond |S| has an adjusted value >= pt.
constexpr int kHistBase = (1023 - 40) << 3;                 // pattern >> 49 of 2^-40
constexpr int kHistTop = (1022 << 3) + 7;                   // pattern >> 49 of the largest double below 1.0
constexpr int kHistBins = kHistTop - kHistBase + 3;         // bin 0: p < 2^-40;  1..: one per pattern >> 49;  last: p >= 1.0

__device__ __forceinline__ int hist_bin(double p) {          // p >= 0
    const int code = (int)((unsigned long long)__double_as_longlong(p) >> 49);
    return code < kHistBase ? 0 : code > kHistTop ? kHistBins - 1 : code - kHistBase + 1;
}
__device__ __forceinline__ double hist_upper_edge(int bin) {   // smallest value NOT in bins 0..bin
    return bin >= kHistBins - 1 ? INFINITY : __longlong_as_double((long long)(kHistBase + bin) << 49);
}

__global__ void __launch_bounds__(256)
bh_hist_kernel(const double *__restrict__ pval, const uint32_t *__restrict__ count, uint32_t cap, double top,
               uint32_t *__restrict__ hist) {
    __shared__ uint32_t h[kHistBins];
    const int b = blockIdx.y;
    const uint32_t n = count[b] < cap ? count[b] : cap;
    for (int i = threadIdx.x; i < kHistBins; i += 256) h[i] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const double p = pval[(size_t)b * cap + i];
        if (p < top) atomicAdd(&h[hist_bin(p)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kHistBins; i += 256)
        if (h[i]) atomicAdd(hist + (size_t)b * kHistBins + i, h[i]);
}

__global__ void __launch_bounds__(64)
bh_threshold_kernel(uint32_t *__restrict__ hist, const uint32_t *__restrict__ count, uint32_t cap, int B, double top,
                    double *__restrict__ threshold_of_block) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    uint32_t *c = hist + (size_t)b * kHistBins;               // -> cumulative: c[i] = #{p < upper_edge(i), p < top}
    uint32_t run = 0;
    for (int i = 0; i < kHistBins; ++i) {
        run += c[i];
        c[i] = run;
    }
    const double dm = (double)(count[b] < cap ? count[b] : cap);
    double T = top;
    uint32_t J = run;                                         // c(top)
    for (int it = 0; it < kHistBins + 2 && J > 0; ++it) {
        const double t = top * ((double)J / dm);
        const int e = hist_bin(t);                            // upper_edge(e) > t
        const double edge = hist_upper_edge(e);
        if (!(edge < T)) break;                               // no tighter edge: fixed point
        T = edge;
        J = c[e];
    }
    threshold_of_block[b] = T;
}

// One workgroup sorts a block's subset in LDS (bitonic network on (p pattern, record index) -- a total order, so the result
// does not depend on the order the compaction happened to append in) and applies BH with the global m.  The LDS request is
// sized to the largest subset of the call (p_max = next power of two), normally 12-20 KB: in the pipeline this kernel runs
// while the NEXT group's fused kernel fills every CU with 2 x 77 KB, and a workgroup that needs more than the 83 KB one
// retiring fused workgroup frees is never placed until that kernel has drained (measured: a fixed 104 KB request cost the
// chromosome run its copy/compute overlap, 76 -> 92 ms).
constexpr int kSortMax = 4096;

__global__ void __launch_bounds__(kBH)
bh_select_lds_kernel(const double *__restrict__ keys, const uint32_t *__restrict__ idx, const uint32_t *__restrict__ k_sub,
                     const uint32_t *__restrict__ count, uint32_t cap, const mst_found *__restrict__ found, double threshold,
                     uint32_t out_cap, uint32_t *__restrict__ out_pixel, uint32_t *__restrict__ out_level,
                     double *__restrict__ out_q, uint32_t *__restrict__ out_count, uint32_t *__restrict__ out_index,
                     uint32_t p_max) {
    extern __shared__ unsigned long long lds_sort[];
    unsigned long long *sk = lds_sort;                                      // [p_max] p patterns
    double *chunk_min = reinterpret_cast<double *>(lds_sort + p_max);       // [kBH]
    uint32_t *si = reinterpret_cast<uint32_t *>(chunk_min + kBH);           // [p_max] record indices
    __shared__ uint32_t n_out;
    const int b = blockIdx.x, t = threadIdx.x;
    const uint32_t m = count[b] < cap ? count[b] : cap;
    const uint32_t k = k_sub[b];
    if (t == 0) n_out = 0;
    if (k == 0) {
        if (t == 0) out_count[b] = 0;
        return;
    }
    if (k > p_max) {                                     // only the no-wait form launches without knowing the subset sizes
        if (t == 0) out_count[b] = MST_BH_RETRY;
        return;
    }
    uint32_t P = 2;
    while (P < k) P <<= 1;
    for (uint32_t i = t; i < P; i += kBH) {
        sk[i] = i < k ? (unsigned long long)__double_as_longlong(keys[(size_t)b * cap + i]) : ~0ull;
        si[i] = i < k ? idx[(size_t)b * cap + i] : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= P; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = t; i < (P >> 1); i += kBH) {
                const uint32_t lo = 2 * i - (i & (stride - 1));             // partner pair (lo, lo + stride)
                const uint32_t hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned long long a = sk[lo], c = sk[hi];
                const uint32_t ia = si[lo], ic = si[hi];
                const bool gt = a > c || (a == c && ia > ic);
                if (gt == up) {
                    sk[lo] = c; sk[hi] = a;
                    si[lo] = ic; si[hi] = ia;
                }
            }
            __syncthreads();
        }
    }
    const uint32_t L = (k + kBH - 1) / kBH;
    const uint32_t lo = t * L < k ? t * L : k, hi = (t + 1) * L < k ? (t + 1) * L : k;
    const double dm = (double)m;
    double mn = INFINITY;
    for (uint32_t i = lo; i < hi; ++i) {
        const double adj = __longlong_as_double((long long)sk[i]) / ((double)(i + 1) / dm);
        mn = adj < mn ? adj : mn;
    }
    chunk_min[t] = mn;
    __syncthreads();
    for (int off = 1; off < kBH; off <<= 1) {
        const double other = (t + off < kBH) ? chunk_min[t + off] : INFINITY;
        __syncthreads();
        if (other < chunk_min[t]) chunk_min[t] = other;
        __syncthreads();
    }
    double run = (t + 1 < kBH) ? chunk_min[t + 1] : INFINITY;
    for (uint32_t i = hi; i > lo; --i) {
        const double adj = __longlong_as_double((long long)sk[i - 1]) / ((double)i / dm);
        run = adj < run ? adj : run;
        const double q = run > 1.0 ? 1.0 : run;
        if (q < threshold) {
            const uint32_t slot = atomicAdd(&n_out, 1u);
            if (slot < out_cap) {
                const mst_found r = found[(size_t)b * cap + si[i - 1]];
                out_pixel[(size_t)b * out_cap + slot] = r.pixel;
                out_level[(size_t)b * out_cap + slot] = r.level;
                out_q[(size_t)b * out_cap + slot] = q;
                if (out_index) out_index[(size_t)b * out_cap + slot] = si[i - 1];
            }
        }
    }
    __syncthreads();
    if (t == 0) out_count[b] = n_out;
}

constexpr size_t sort_lds_bytes(uint32_t p_max) {
    return (size_t)p_max * (sizeof(unsigned long long) + sizeof(uint32_t)) + kBH * sizeof(double);
}

}  // namespace

namespace {
int bh_select_impl(const mst_found *found, const double *pval, const uint32_t *count, int32_t B, uint32_t cap,
                   double threshold, uint32_t out_cap, uint32_t *out_pixel, uint32_t *out_level, double *out_q,
                   uint32_t *out_count, uint32_t *out_index, void *workspace, uint64_t workspace_bytes, void *stream,
                   uint32_t nowait_records = 0) {
    if (!found || !pval || !count || !out_pixel || !out_level || !out_q || !out_count || !workspace || B <= 0 ||
        B > 65535 || cap == 0 || out_cap == 0 || (size_t)B * cap > 0x7FFFFFFFull)
        return mst::fail(MST_E_ARG, "mst_bh_select: bad argument (B * cap must fit in int32)");
    if (workspace_bytes < mst_bh_workspace_bytes(B, cap))
        return mst::fail(MST_E_ARG, "mst_bh_select: workspace too small (mst_bh_workspace_bytes)");
    hipStream_t s = mst::as_stream(stream);
    const size_t n = (size_t)B * cap;
    char *w = reinterpret_cast<char *>(workspace);
    int *seg_begin = reinterpret_cast<int *>(w);
    int *seg_end = seg_begin + B;
    w += align_up(sizeof(int) * 2 * B, 256);
    uint32_t *idx_in = reinterpret_cast<uint32_t *>(w);
    w += align_up(sizeof(uint32_t) * n, 256);
    uint32_t *idx_out = reinterpret_cast<uint32_t *>(w);
    w += align_up(sizeof(uint32_t) * n, 256);
    double *keys_out = reinterpret_cast<double *>(w);
    w += align_up(sizeof(double) * n, 256);
    size_t temp = sort_temp_bytes(B, cap);
    char *tail = w + align_up(temp, 256);
    double *keys_in = reinterpret_cast<double *>(tail);
    tail += align_up(sizeof(double) * n, 256);
    uint32_t *k_sub = reinterpret_cast<uint32_t *>(tail);
    tail += align_up(sizeof(uint32_t) * (size_t)B, 256);
    uint32_t *hist = reinterpret_cast<uint32_t *>(tail);
    tail += align_up(sizeof(uint32_t) * (size_t)B * kHistBins, 256);
    double *thr = reinterpret_cast<double *>(tail);
    // 1. per block, the threshold under which every record that can be selected lies (histogram of p + fixed point)
    MST_HIP(hipMemsetAsync(k_sub, 0, (size_t)(reinterpret_cast<char *>(thr) - reinterpret_cast<char *>(k_sub)), s));
    const double top = threshold * (1.0 + 1e-9);
    const int gx = (int)((cap + 255) / 256 < 256 ? (cap + 255) / 256 : 256);
    bh_hist_kernel<<<dim3(gx, B), 256, 0, s>>>(pval, count, cap, top, hist);
    MST_LAUNCH_CHECK();
    bh_threshold_kernel<<<(B + 63) / 64, 64, 0, s>>>(hist, count, cap, B, top, thr);
    MST_LAUNCH_CHECK();
    // 2. those records, compacted (keys = p, values = record index)
    compact_below_kernel<<<dim3(gx, B), 256, 0, s>>>(pval, count, cap, thr, keys_in, idx_in, k_sub);
    MST_LAUNCH_CHECK();
    // 3. sort + BH + selection: in LDS when every block's subset fits (the normal case), else the segmented radix sort
    if (nowait_records) {
        // no look at the subset sizes: the in-LDS sort sized for the subset the CALLER expects (its LDS request decides whether
        // the kernel finds room next to a running fused kernel, see above); a block with a larger one reports MST_BH_RETRY
        bh_select_lds_kernel<<<B, kBH, sort_lds_bytes(nowait_records), s>>>(keys_in, idx_in, k_sub, count, cap, found, threshold,
                                                                           out_cap, out_pixel, out_level, out_q, out_count,
                                                                           out_index, nowait_records);
        MST_LAUNCH_CHECK();
        return MST_OK;
    }
    std::vector<uint32_t> k_host((size_t)B);
    MST_HIP(hipMemcpyAsync(k_host.data(), k_sub, sizeof(uint32_t) * (size_t)B, hipMemcpyDeviceToHost, s));
    MST_HIP(hipStreamSynchronize(s));
    uint32_t k_max = 0;
    for (uint32_t k : k_host) k_max = k > k_max ? k : k_max;
    if (k_max <= (uint32_t)kSortMax) {
        uint32_t p_max = 2;
        while (p_max < k_max) p_max <<= 1;
        static_assert(sort_lds_bytes(kSortMax) <= 64 * 1024, "stays under the default dynamic LDS limit");
        bh_select_lds_kernel<<<B, kBH, sort_lds_bytes(p_max), s>>>(keys_in, idx_in, k_sub, count, cap, found, threshold,
                                                                  out_cap, out_pixel, out_level, out_q, out_count, out_index,
                                                                  p_max);
        MST_LAUNCH_CHECK();
        return MST_OK;
    }
    seg_bounds_kernel<<<(B + 255) / 256, 256, 0, s>>>(k_sub, cap, B, seg_begin, seg_end);
    MST_LAUNCH_CHECK();
    MST_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(w, temp, keys_in, keys_out, idx_in, idx_out, (int)n, B, seg_begin,
                                                        seg_end, 0, 64, s));
    bh_select_kernel<<<B, kBH, 0, s>>>(keys_out, idx_out, k_sub, count, cap, found, threshold, out_cap, out_pixel,
                                       out_level, out_q, out_count, out_index);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
}  // namespace

extern "C" int mst_bh_select(const mst_found *found, const double *pval, const uint32_t *count, int32_t B, uint32_t cap,
                             double threshold, uint32_t out_cap, uint32_t *out_pixel, uint32_t *out_level, double *out_q,
                             uint32_t *out_count, void *workspace, uint64_t workspace_bytes, void *stream) {
    MST_RANGE("tail: mst_bh_select");
    return bh_select_impl(found, pval, count, B, cap, threshold, out_cap, out_pixel, out_level, out_q, out_count, nullptr,
                          workspace, workspace_bytes, stream);
}

extern "C" int mst_bh_select_records(const mst_found *found, const double *pval, const uint32_t *count, int32_t B,
                                     uint32_t cap, double threshold, uint32_t out_cap, uint32_t *out_pixel,
                                     uint32_t *out_level, double *out_q, uint32_t *out_index, uint32_t *out_count,
                                     void *workspace, uint64_t workspace_bytes, void *stream) {
    MST_RANGE("tail: mst_bh_select_records");
    if (!out_index) return mst::fail(MST_E_ARG, "mst_bh_select_records: bad argument");
    return bh_select_impl(found, pval, count, B, cap, threshold, out_cap, out_pixel, out_level, out_q, out_count, out_index,
                          workspace, workspace_bytes, stream);
}

extern "C" int mst_bh_select_nowait(const mst_found *found, const double *pval, const uint32_t *count, int32_t B,
                                    uint32_t cap, double threshold, uint32_t out_cap, uint32_t *out_pixel,
                                    uint32_t *out_level, double *out_q, uint32_t *out_index, uint32_t *out_count,
                                    uint32_t lds_records, void *workspace, uint64_t workspace_bytes, void *stream) {
    MST_RANGE("tail: mst_bh_select_nowait");
    if (lds_records < 2 || lds_records > (uint32_t)kSortMax || (lds_records & (lds_records - 1)))
        return mst::fail(MST_E_ARG, "mst_bh_select_nowait: lds_records must be a power of two in [2, %d]", kSortMax);
    return bh_select_impl(found, pval, count, B, cap, threshold, out_cap, out_pixel, out_level, out_q, out_count, out_index,
                          workspace, workspace_bytes, stream, lds_records);
}

extern "C" int mst_select_below(const mst_found *found, const double *q, const uint32_t *found_count, int32_t B,
                                uint32_t found_cap, double threshold, uint32_t out_cap, uint32_t *out_pixel,
                                uint32_t *out_level, double *out_q, uint32_t *out_count, void *stream) {
    MST_RANGE("tail: mst_select_below");
    if (!found || !q || !found_count || !out_pixel || !out_level || !out_q || !out_count || B <= 0 || B > 65535 ||
        found_cap == 0 || out_cap == 0)
        return mst::fail(MST_E_ARG, "mst_select_below: bad argument");
    hipStream_t s = mst::as_stream(stream);
    MST_HIP(hipMemsetAsync(out_count, 0, sizeof(uint32_t) * (size_t)B, s));
    const int gx = (int)((found_cap + 255) / 256 < 256 ? (found_cap + 255) / 256 : 256);
    select_below_kernel<<<dim3(gx, B), 256, 0, s>>>(found, q, found_count, found_cap, threshold, out_cap, out_pixel,
                                                   out_level, out_q, out_count);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" uint64_t mst_bh_workspace_bytes(int32_t B, uint32_t cap) {
    if (B <= 0 || cap == 0 || (size_t)B * cap > 0x7FFFFFFFull) return 0;
    const size_t n = (size_t)B * cap;
    // segment bounds, index in/out, sorted keys, the sort's temporary storage, and (mst_bh_select) compacted keys + subset sizes
    return align_up(sizeof(int) * 2 * B, 256) + align_up(sizeof(uint32_t) * n, 256) * 2 + align_up(sizeof(double) * n, 256) * 2 +
           align_up(sort_temp_bytes(B, cap), 256) + align_up(sizeof(uint32_t) * (size_t)B, 256) +
           align_up(sizeof(uint32_t) * (size_t)B * kHistBins, 256) + align_up(sizeof(double) * (size_t)B, 256);
}

extern "C" int mst_bh_fdr(const double *pval, const uint32_t *count, int32_t B, uint32_t cap, double *q, void *workspace,
                          uint64_t workspace_bytes, void *stream) {
    MST_RANGE("tail: mst_bh_fdr");
    if (!pval || !count || !q || !workspace || B <= 0 || B > 65535 || cap == 0 || (size_t)B * cap > 0x7FFFFFFFull)
        return mst::fail(MST_E_ARG, "mst_bh_fdr: bad argument (B * cap must fit in int32)");
    if (workspace_bytes < mst_bh_workspace_bytes(B, cap))
        return mst::fail(MST_E_ARG, "mst_bh_fdr: workspace too small");
    hipStream_t s = mst::as_stream(stream);
    const size_t n = (size_t)B * cap;
    char *w = reinterpret_cast<char *>(workspace);
    int *seg_begin = reinterpret_cast<int *>(w);
    int *seg_end = seg_begin + B;
    w += align_up(sizeof(int) * 2 * B, 256);
    uint32_t *idx_in = reinterpret_cast<uint32_t *>(w);
    w += align_up(sizeof(uint32_t) * n, 256);
    uint32_t *idx_out = reinterpret_cast<uint32_t *>(w);
    w += align_up(sizeof(uint32_t) * n, 256);
    double *keys_out = reinterpret_cast<double *>(w);
    w += align_up(sizeof(double) * n, 256);
    size_t temp = sort_temp_bytes(B, cap);
    const int gx = (int)((cap + 255) / 256 < 1024 ? (cap + 255) / 256 : 1024);
    bh_setup_kernel<<<dim3(gx, B), 256, 0, s>>>(count, cap, B, seg_begin, seg_end, idx_in);
    MST_LAUNCH_CHECK();
    MST_HIP(hipcub::DeviceSegmentedRadixSort::SortPairs(w, temp, pval, keys_out, idx_in, idx_out, (int)n, B, seg_begin,
                                                        seg_end, 0, 64, s));
    bh_adjust_kernel<<<B, kBH, 0, s>>>(keys_out, idx_out, count, cap, q);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
