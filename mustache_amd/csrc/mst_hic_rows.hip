// `.hic` rows -> band on the device (gfx950): the per-record half of the `.hic` read.
//
// The reference reaches a `.hic` file through hic-straw (mustache/mustache.py:328-333) and then walks the records in Python:
// bins (`// res`, :367-368), the distance filter and `counts > 0` (:385-389), and finally `cc[xc, yc] = vc` per block
// (:919-924).  Here the host (libmustache_io.so, mst_hic_rawstream_*) only inflates the zlib blocks and copies each row's
// record bytes, as the file stores them, into page-locked slabs with one 16-byte directory entry per row
// (include/mustache_hicrow.h); this kernel does everything per record:
//     column + count out of the payload (int16 / int32 column, float32 / int16 count, or a dense grid's bare counts with
//     NaN / -32768 holes) -> binX <= binY -> distance filter -> count / (norm[binX] * norm[binY]) in float64, rounded to
//     float32 as straw does -> NaN / non-positive dropped -> chromosome-size limit -> band[binY - binX][binX] = (double)value
// -- the arithmetic and the order of the tests of hic_reader.cpp's emit(), so the band is bit-identical to the host decode's.
// A workgroup per 8 rows, their records walked as one flat list (a row of a 1 kb map holds a few hundred records, some a
// thousand: 6-byte records read as 16-bit halves, contiguous across the lanes); the statistics (max binY + 1 kept, kept records,
// records the band cannot hold) are reduced per workgroup and leave with one atomic each.  HBM-bound in principle, 14 bytes per
// record with the band written 8 bytes at a time into 2002 different rows; in practice it runs under the host's inflate.
#include "mst_common.h"
#include "../../include/mustache_hicrow.h"

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ uint32_t ld16(const uint8_t *p) { return *reinterpret_cast<const uint16_t *>(p); }
__device__ __forceinline__ uint32_t ld32(const uint8_t *p) { return ld16(p) | (ld16(p + 2) << 16); }   // 2-byte aligned

__device__ __forceinline__ long long wave_max(long long v) {
    for (int o = 32; o > 0; o >>= 1) {
        const long long t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// VERIFY = false: scatter.  VERIFY = true: read-back -- counts the records whose pixel does not hold their value (a pixel
// two records with different values were written to shows up for one of them whichever store won: malformed input).
// A workgroup takes kRowsPerBlock consecutive rows and walks their records as ONE flat list (prefix of the row lengths in LDS,
// the row of a record found by a short search): every lane has work whatever the row lengths are -- with a wavefront per row
// (the first form) a slab took 78 us, its few 1000-record rows setting the pace and every wave paying a full reduction.
constexpr int kRowsPerBlock = 8;
template <bool VERIFY>
__global__ void __launch_bounds__(kThreads)
hic_rows_kernel(const uint8_t *__restrict__ payload, const mst_hic_row *__restrict__ rows, int n_rows,
                const double *__restrict__ norm, long long n_norm, long long max_dist, long long y_limit, long long n, int dpx,
                double *__restrict__ band, unsigned long long *__restrict__ stats) {
    __shared__ mst_hic_row srow[kRowsPerBlock];
    __shared__ int sbeg[kRowsPerBlock + 1];
    __shared__ unsigned long long sred[4][kThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63;
    long long ymax1 = 0;                                  // max binY + 1 over the records this lane kept
    unsigned long long kept = 0, beyond = 0, bad = 0;
    for (int r0 = blockIdx.x * kRowsPerBlock; r0 < n_rows; r0 += gridDim.x * kRowsPerBlock) {
        const int nr = n_rows - r0 < kRowsPerBlock ? n_rows - r0 : kRowsPerBlock;
        __syncthreads();                                  // the previous group's tables are no longer read
        if (tid < nr) srow[tid] = rows[r0 + tid];
        __syncthreads();
        if (tid == 0) {
            int run = 0;
            for (int i = 0; i < nr; ++i) {
                sbeg[i] = run;
                run += (int)(srow[i].count & MST_HIC_ROW_COUNT_MASK);
            }
            for (int i = nr; i <= kRowsPerBlock; ++i) sbeg[i] = run;
        }
        __syncthreads();
        const int total = sbeg[kRowsPerBlock];
        for (int i = tid; i < total; i += kThreads) {
            int q = 0;
#pragma unroll
            for (int t = 1; t < kRowsPerBlock; ++t) q += (i >= sbeg[t]) ? 1 : 0;       // the row this record belongs to
            const mst_hic_row e = srow[q];
            const int j = i - sbeg[q];
            const bool short_c = e.count & MST_HIC_ROW_SHORT_COUNTS, int_x = e.count & MST_HIC_ROW_INT_COLUMNS,
                       dense = e.count & MST_HIC_ROW_DENSE;
            const int rec = (dense ? 0 : (int_x ? 4 : 2)) + (short_c ? 2 : 4);
            const uint8_t *q8 = payload + e.off + (size_t)j * rec;
            int x = j;
            if (!dense) {
                x = int_x ? (int)ld32(q8) : (int)(int16_t)ld16(q8);
                q8 += int_x ? 4 : 2;
            }
            float val;
            if (short_c) {
                const int16_t sv = (int16_t)ld16(q8);
                if (dense && sv == -32768) continue;
                val = (float)sv;
            } else {
                val = __uint_as_float(ld32(q8));
                if (dense && val != val) continue;
            }
            long long bx = (long long)e.x_off + x, by = e.y;
            if (bx > by) {
                const long long t = bx;
                bx = by;
                by = t;
            }
            if (max_dist >= 0 && by - bx > max_dist) continue;
            float c = val;
            if (norm) {
                if (bx < 0 || by >= n_norm) continue;
                c = (float)((double)val / (norm[bx] * norm[by]));
            }
            if (c != c || !(c > 0.0f)) continue;
            if (by >= y_limit) continue;
            const long long d = by - bx;
            if (bx < 0 || by >= n || d > dpx + 1) {       // a record the band cannot hold: reported, never written
                ++beyond;
                continue;
            }
            if (VERIFY) {
                if (band[d * n + bx] != (double)c) ++bad;
            } else {
                band[d * n + bx] = (double)c;
                ++kept;
                ymax1 = by + 1 > ymax1 ? by + 1 : ymax1;
            }
        }
    }
    // one reduction per workgroup: waves with DPP-free shuffles, then the first lanes over the waves; four atomics at most
    ymax1 = wave_max(ymax1);
    kept = wave_sum(kept);
    beyond = wave_sum(beyond);
    bad = wave_sum(bad);
    if (lane == 0) {
        sred[0][tid >> 6] = (unsigned long long)ymax1;
        sred[1][tid >> 6] = kept;
        sred[2][tid >> 6] = beyond;
        sred[3][tid >> 6] = bad;
    }
    __syncthreads();
    if (tid == 0) {
        unsigned long long m = 0, k = 0, o = 0, v = 0;
        for (int w = 0; w < kThreads / 64; ++w) {
            m = sred[0][w] > m ? sred[0][w] : m;
            k += sred[1][w];
            o += sred[2][w];
            v += sred[3][w];
        }
        if (m) atomicMax(&stats[0], m);
        if (k) atomicAdd(&stats[1], k);
        if (o) atomicAdd(&stats[2], o);
        if (v) atomicAdd(&stats[3], v);
    }
}

}  // namespace

extern "C" int mst_band_scatter_hic_rows(const void *payload, const void *rows, int32_t n_rows, const double *norm,
                                         int64_t n_norm, int64_t max_dist, int64_t y_limit, int64_t n, int32_t dpx, double *band,
                                         uint64_t *stats, int32_t verify, void *stream) {
    MST_RANGE("read: mst_band_scatter_hic_rows");
    if (!band || !stats || n <= 0 || dpx < 0 || n_rows < 0 || (n_rows > 0 && (!payload || !rows)) || (norm && n_norm < 0) ||
        (reinterpret_cast<uintptr_t>(payload) & 1) || (reinterpret_cast<uintptr_t>(rows) & 3))
        return mst::fail(MST_E_ARG, "mst_band_scatter_hic_rows: bad argument (payload 2-byte, rows 4-byte aligned)");
    if (n_rows == 0) return MST_OK;
    hipStream_t s = mst::as_stream(stream);
    const int64_t want = ((int64_t)n_rows + kRowsPerBlock - 1) / kRowsPerBlock;
    const int g = (int)(want < 4096 ? want : 4096);
    auto *st = reinterpret_cast<unsigned long long *>(stats);
    auto *rw = static_cast<const mst_hic_row *>(rows);
    auto *pl = static_cast<const uint8_t *>(payload);
    if (y_limit <= 0) y_limit = INT64_MAX;
    if (verify)
        hic_rows_kernel<true><<<g, kThreads, 0, s>>>(pl, rw, n_rows, norm, n_norm, max_dist, y_limit, n, dpx, band, st);
    else
        hic_rows_kernel<false><<<g, kThreads, 0, s>>>(pl, rw, n_rows, norm, n_norm, max_dist, y_limit, n, dpx, band, st);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
