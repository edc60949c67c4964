// Small kernels after the sigma loop (gfx950):
//   mst_found_pvalues      <- reference mustache/mustache.py:755-756 (expon.fit + 1 - expon.cdf), found pixels only
//   mst_candidate_features <- reference mustache/mustache.py:800-811, :824 (window densities of nz, c[x, y])
//   mst_gather_diagonals   <- reference mustache/mustache.py:816-823 (diagonals for the diagonal-mean filter)
//   mst_diag_means         <- reference mustache/mustache.py:816-824 (np.mean of their non-zero entries, NumPy's order)
// All of them touch a few thousand pixels per block; they exist so the tail never pulls a dense block to the host.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mst_common.h"

namespace {

// One workgroup for the whole launch (B * n_tested fits are a few thousand doubles): the fits, the host's summary image and the
// flags word {1: a block's record count exceeds the capacity, 2: non-finite statistics in a block that tested pixels} -- the
// word is REDUCED inside the workgroup and stored once, so nothing has to be zeroed beforehand and no atomics are needed.
// (It used to be a zero-filled word that B workgroups OR-ed into.  Zeroing it with hipMemsetAsync put a memset node into the
// captured finish graph, and on this ROCm a replayed memset node sometimes writes stale host-heap bytes instead of its
// pattern: the flags word came back as 0x2, 0x20, 0x402, 0x22ee3402, half a heap pointer ... about once per 30 replays in a
// long sweep, with everything else in the summary correct -- LABBOOK.md R4.6.  No captured sequence of this library contains a
// memset any more.)
__global__ void __launch_bounds__(256)
fit_kernel(const double *__restrict__ level_stats, const uint32_t *__restrict__ nz_count, int n_tested,
           double *__restrict__ fit, int *__restrict__ flags, const uint32_t *__restrict__ found_count, uint32_t found_cap,
           char *__restrict__ summary, int B) {
    const size_t cw = 8 * (size_t)((B + 1) / 2);
    int bits = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        const uint32_t n = found_count[b];
        if (n > found_cap) bits |= 1;
        if (summary) {
            reinterpret_cast<uint32_t *>(summary + 16)[b] = n;
            reinterpret_cast<uint32_t *>(summary + 16 + cw)[b] = nz_count[b];
        }
    }
    double *sf = summary ? reinterpret_cast<double *>(summary + 16 + 2 * cw) : nullptr;
    for (int i = threadIdx.x; i < B * n_tested; i += blockDim.x) {
        const int b = i / n_tested, t = i - b * n_tested;
        const double mn = level_stats[((size_t)b * MST_MAX_TESTED + t) * 2];
        const double sm = level_stats[((size_t)b * MST_MAX_TESTED + t) * 2 + 1];
        const uint32_t nz = nz_count[b];
        const double cnt = (double)nz;
        const double loc = mn;
        const double scale = sm / cnt - loc;
        fit[((size_t)b * MST_MAX_TESTED + t) * 2] = loc;
        fit[((size_t)b * MST_MAX_TESTED + t) * 2 + 1] = scale;
        if (sf) {
            sf[((size_t)b * MST_MAX_TESTED + t) * 2] = loc;
            sf[((size_t)b * MST_MAX_TESTED + t) * 2 + 1] = scale;
        }
        if (nz > 0 && !(isfinite(mn) && isfinite(sm))) bits |= 2;
    }
    __shared__ int sbits[256];
    sbits[threadIdx.x] = bits;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sbits[threadIdx.x] |= sbits[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < 4) flags[threadIdx.x] = threadIdx.x == 0 ? sbits[0] : 0;      // the 16-byte header of the summary
}

// p = 1 - cdf, cdf = -expm1(-x) for x > 0 else 0 (scipy/stats/_distn_infrastructure.py:2127-2139,
// _continuous_distns.py:2087-2088).  SciPy's expm1 is exp(x) - 1 outside |x| <= 0.5; we form the same two
// roundings (E = exp(-x); E - 1; negate; 1 - .) so tiny p-values land on the same 2^-53 grid points.
__global__ void __launch_bounds__(256)
pvalue_kernel(const mst_found *__restrict__ found, uint32_t found_cap, const uint32_t *__restrict__ found_count,
              const double *__restrict__ fit, double *__restrict__ pval,
              int32_t *__restrict__ pix_out, uint8_t *__restrict__ lvl_out, double *__restrict__ pv_out, uint32_t pitch) {
    const int b = blockIdx.y;
    const uint32_t n = found_count[b];
    if (n > found_cap) return;          // fit_kernel has flagged the overflow: the caller re-runs with more room
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const mst_found rec = found[(size_t)b * found_cap + i];
        const int t = (int)rec.level - 1;
        const double loc = fit[((size_t)b * MST_MAX_TESTED + t) * 2];
        const double scale = fit[((size_t)b * MST_MAX_TESTED + t) * 2 + 1];
        const double x = (fabs(rec.value) - loc) / scale;
        double cdf;
        if (x != x || !(scale > 0.0)) {
            cdf = NAN;
        } else if (x > 0.5) {
            const double e = exp(-x);
            cdf = -(e - 1.0);
        } else if (x > 0.0) {
            cdf = -expm1(-x);
        } else {
            cdf = 0.0;
        }
        pval[(size_t)b * found_cap + i] = 1.0 - cdf;
        // the first `pitch` records of the block once more as three narrow, densely pitched arrays: what a caller that downloads
        // whole found sets copies -- 4 + 1 + 8 bytes per record in three CONTIGUOUS transfers, no element-wise unpacking passes
        if (pix_out && i < pitch) {
            pix_out[(size_t)b * pitch + i] = (int32_t)rec.pixel;
            lvl_out[(size_t)b * pitch + i] = (uint8_t)rec.level;
            pv_out[(size_t)b * pitch + i] = 1.0 - cdf;
        }
    }
}

// one wave per candidate: lanes stride over the (2h+1)^2 and (4h+1)^2 windows of the nz byte mask
__global__ void __launch_bounds__(256)
features_kernel(const double *__restrict__ c, const uint8_t *__restrict__ nz, int CH, int b,
                const uint32_t *__restrict__ pixel, const int32_t *__restrict__ half, int n,
                uint32_t *__restrict__ cnt1, uint32_t *__restrict__ cnt2, double *__restrict__ cval) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const uint8_t *nb = nz + (size_t)b * CH * CH;
    const int x = (int)(pixel[i] / (uint32_t)CH), y = (int)(pixel[i] % (uint32_t)CH);
    uint32_t out[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int s = half[i] * (pass + 1);
        uint32_t acc = 0;
        // nz[x-s : x+s+1, y-s : y+s+1] with Python slice semantics: a negative start wraps to CH + start, which
        // leaves the slice empty whenever CH >= 2s+1 -- the reference relies on that
        int xs = x - s, ys = y - s;
        if (xs < 0) xs = xs + CH > 0 ? xs + CH : 0;
        if (ys < 0) ys = ys + CH > 0 ? ys + CH : 0;
        const int x1 = x + s + 1 < CH ? x + s + 1 : CH, y1 = y + s + 1 < CH ? y + s + 1 : CH;
        const int w = y1 - ys, h = x1 - xs;
        if (w > 0 && h > 0) {
            for (int q = lane; q < w * h; q += 64) {
                const int dx = q / w, dy = q - dx * w;
                acc += nb[(size_t)(xs + dx) * CH + (ys + dy)];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        out[pass] = acc;
    }
    if (lane == 0) {
        cnt1[i] = out[0];
        cnt2[i] = out[1];
        cval[i] = c[(size_t)b * CH * CH + (size_t)x * CH + y];
    }
}

__global__ void __launch_bounds__(256)
diag_kernel(const double *__restrict__ c, int CH, int b, const int32_t *__restrict__ diag_k, double *__restrict__ out) {
    const int i = blockIdx.y;
    const int k = diag_k[i];
    const double *cb = c + (size_t)b * CH * CH;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < CH; r += gridDim.x * blockDim.x)
        out[(size_t)i * CH + r] = (k >= 0 && r + k < CH) ? cb[(size_t)r * CH + r + k] : 0.0;
}

// ---- the same two gathers when the block only exists as a window of the band (mst_scale_space_band) -----------------
// pixel (x, y) of the block that starts at `start`:  off = y - x,  raw = band[off][start + x] (0 outside 0..dpx+1 or past
// the chromosome end),  tested = raw != 0 && off >= 4,  value = 2 where off <= 4 or off >= dpx+1 else raw.
__device__ __forceinline__ double band_raw(const double *__restrict__ band, int64_t n, int dpx, int64_t start, int x,
                                           int y) {
    const int off = y - x;
    if (off < 0 || off > dpx + 1 || start + y >= n) return 0.0;
    return band[(int64_t)off * n + start + x];
}

// `starts` != nullptr: candidate i belongs to the block that starts at starts[i] (candidates of several blocks in one launch)
__global__ void __launch_bounds__(256)
features_band_kernel(const double *__restrict__ band, int64_t n, int dpx, int64_t start, const int64_t *__restrict__ starts,
                     int CH, const uint32_t *__restrict__ pixel, const int32_t *__restrict__ half, int ncand,
                     uint32_t *__restrict__ cnt1, uint32_t *__restrict__ cnt2, double *__restrict__ cval) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= ncand) return;
    if (starts) start = starts[i];
    const int x = (int)(pixel[i] / (uint32_t)CH), y = (int)(pixel[i] % (uint32_t)CH);
    uint32_t out[2];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int s = half[i] * (pass + 1);
        uint32_t acc = 0;
        int xs = x - s, ys = y - s;                              // Python slice semantics, see features_kernel
        if (xs < 0) xs = xs + CH > 0 ? xs + CH : 0;
        if (ys < 0) ys = ys + CH > 0 ? ys + CH : 0;
        const int x1 = x + s + 1 < CH ? x + s + 1 : CH, y1 = y + s + 1 < CH ? y + s + 1 : CH;
        const int w = y1 - ys, h = x1 - xs;
        if (w > 0 && h > 0) {
            for (int q = lane; q < w * h; q += 64) {
                const int dx = q / w, dy = q - dx * w;
                const int px = xs + dx, py = ys + dy;
                acc += (py - px >= 4 && band_raw(band, n, dpx, start, px, py) != 0.0) ? 1u : 0u;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
        out[pass] = acc;
    }
    if (lane == 0) {
        cnt1[i] = out[0];
        cnt2[i] = out[1];
        const int off = y - x;
        cval[i] = (off <= 4 || off >= dpx + 1) ? 2.0 : band_raw(band, n, dpx, start, x, y);
    }
}

__global__ void __launch_bounds__(256)
diag_band_kernel(const double *__restrict__ band, int64_t n, int dpx, int64_t start, int CH,
                 const int32_t *__restrict__ diag_k, double *__restrict__ out) {
    const int i = blockIdx.y;
    const int k = diag_k[i];
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < CH; r += gridDim.x * blockDim.x) {
        double v = 0.0;
        if (k >= 0 && r + k < CH) v = (k <= 4 || k >= dpx + 1) ? 2.0 : band_raw(band, n, dpx, start, r, r + k);
        out[(size_t)i * CH + r] = v;
    }
}

// ---- diagonal means for the diagonal-mean filter (mustache.py:816-824): mean of the NON-ZERO entries of diagonal k of the
// filled block, np.mean(dg[dg != 0]).  NumPy reduces a contiguous float64 array with its pairwise summation
// (numpy/_core/src/umath/loops_utils.h.src, @TYPE@_pairwise_sum: < 8 elements serially; <= 128 elements with eight strided
// accumulators combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and the n % 8 tail added serially; larger arrays split at
// n/2 rounded down to a multiple of 8, recursively).  The kernel follows that order exactly, on the compacted non-zero
// entries, so the mean -- and with it the filter's `c[x,y] > 2*mean` decision -- is bit-identical to the reference's.
__device__ double np_pairwise_leaf(const double *a, int n) {
    if (n < 8) {
        double res = -0.0;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = a[j];
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += a[i + j];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
}

// the recursion's stack lives in LDS (one lane runs this; as private arrays the compiler put it into scratch memory)
struct PairwiseStack {
    int off[24], len[24], state[24];
    double left[24];
};

__device__ double np_pairwise_sum(const double *a, int n, PairwiseStack &st) {
    int *off = st.off, *len = st.len, *state = st.state;
    double *left = st.left;
    int sp = 1;
    off[0] = 0;
    len[0] = n;
    state[0] = 0;
    double r = 0.0;
    bool have = false;                 // r holds the value of the node just popped
    while (sp > 0) {
        const int t = sp - 1;
        if (!have) {
            if (len[t] <= 128) {
                r = np_pairwise_leaf(a + off[t], len[t]);
                have = true;
                --sp;
            } else {                    // descend into the left half
                int n2 = len[t] / 2;
                n2 -= n2 % 8;
                state[t] = 0;
                off[sp] = off[t];
                len[sp] = n2;
                ++sp;
            }
        } else if (state[t] == 0) {    // left half done -> right half
            int n2 = len[t] / 2;
            n2 -= n2 % 8;
            left[t] = r;
            state[t] = 1;
            off[sp] = off[t] + n2;
            len[sp] = len[t] - n2;
            ++sp;
            have = false;
        } else {                        // both halves done
            r = left[t] + r;
            --sp;
        }
    }
    return r;
}

// one 64-lane workgroup per diagonal: ordered compaction of the non-zero entries into LDS, then lane 0 sums them
template <bool BAND>
__global__ void __launch_bounds__(64)
diag_mean_kernel(const double *__restrict__ src, int64_t n, int dpx, int64_t start, const int64_t *__restrict__ starts,
                 int CH, int b, const int32_t *__restrict__ diag_k, double *__restrict__ mean_out) {
    extern __shared__ double dm_buf[];
    __shared__ PairwiseStack pw_stack;
    const int i = blockIdx.x, lane = threadIdx.x;
    const int k = diag_k[i];
    if (BAND && starts) start = starts[i];             // one launch for the diagonals of many blocks
    const int L = (k >= 0 && k < CH) ? CH - k : 0;
    const double *cb = BAND ? src : src + (size_t)b * CH * CH;
    int base = 0;
    for (int r0 = 0; r0 < L; r0 += 64) {
        const int r = r0 + lane;
        double v = 0.0;
        if (r < L) {
            if (BAND) v = (k <= 4 || k >= dpx + 1) ? 2.0 : band_raw(src, n, dpx, start, r, r + k);
            else v = cb[(size_t)r * CH + r + k];
        }
        const bool keep = r < L && v != 0.0;           // NaN != 0 is true, as in NumPy
        const unsigned long long bal = __ballot(keep);
        if (keep) dm_buf[base + __popcll(bal & ((1ull << lane) - 1ull))] = v;
        base += __popcll(bal);
    }
    __syncthreads();
    if (lane == 0) mean_out[i] = np_pairwise_sum(dm_buf, base, pw_stack) / (double)base;
}

template <bool BAND>
int diag_means_launch(const double *src, int64_t n, int dpx, int64_t start, const int64_t *starts, int CH, int b,
                      const int32_t *diag_k, int nd, double *mean_out, hipStream_t s) {
    const size_t lds = (size_t)CH * sizeof(double);
    if (lds > 160 * 1024 - 1024) return mst::fail(MST_E_ARG, "diagonal means: CH %d does not fit the LDS", CH);
    if (lds > 64 * 1024)
        MST_HIP(hipFuncSetAttribute((const void *)diag_mean_kernel<BAND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds));
    diag_mean_kernel<BAND><<<nd, 64, lds, s>>>(src, n, dpx, start, starts, CH, b, diag_k, mean_out);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

}  // namespace

extern "C" int mst_diag_means(const double *c, int32_t CH, int32_t b, const int32_t *diag_k, int32_t nd, double *mean_out,
                              void *stream) {
    if (nd == 0) return MST_OK;
    if (!c || !diag_k || !mean_out || CH <= 0 || b < 0 || nd < 0)
        return mst::fail(MST_E_ARG, "mst_diag_means: bad argument");
    return diag_means_launch<false>(c, 0, 0, 0, nullptr, CH, b, diag_k, nd, mean_out, mst::as_stream(stream));
}

extern "C" int mst_diag_means_band(const double *band, int64_t n, int32_t dpx, int64_t start, int32_t CH,
                                   const int32_t *diag_k, int32_t nd, double *mean_out, void *stream) {
    MST_RANGE("tail: mst_diag_means_band");
    if (nd == 0) return MST_OK;
    if (!band || !diag_k || !mean_out || CH <= 0 || n <= 0 || dpx < 0 || nd < 0)
        return mst::fail(MST_E_ARG, "mst_diag_means_band: bad argument");
    return diag_means_launch<true>(band, n, dpx, start, nullptr, CH, 0, diag_k, nd, mean_out, mst::as_stream(stream));
}

extern "C" int mst_diag_means_band_multi(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t CH,
                                         const int32_t *diag_k, int32_t nd, double *mean_out, void *stream) {
    MST_RANGE("tail: mst_diag_means_band_multi");
    if (nd == 0) return MST_OK;
    if (!band || !starts || !diag_k || !mean_out || CH <= 0 || n <= 0 || dpx < 0 || nd < 0)
        return mst::fail(MST_E_ARG, "mst_diag_means_band_multi: bad argument");
    return diag_means_launch<true>(band, n, dpx, 0, starts, CH, 0, diag_k, nd, mean_out, mst::as_stream(stream));
}

extern "C" int mst_candidate_features_band(const double *band, int64_t n, int32_t dpx, int64_t start, int32_t CH,
                                           const uint32_t *pixel, const int32_t *half, int32_t ncand, uint32_t *cnt1,
                                           uint32_t *cnt2, double *cval, void *stream) {
    MST_RANGE("tail: mst_candidate_features_band");
    if (ncand == 0) return MST_OK;
    if (!band || !pixel || !half || !cnt1 || !cnt2 || !cval || CH <= 0 || n <= 0 || dpx < 0 || ncand < 0)
        return mst::fail(MST_E_ARG, "mst_candidate_features_band: bad argument");
    features_band_kernel<<<(ncand + 3) / 4, 256, 0, mst::as_stream(stream)>>>(band, n, dpx, start, nullptr, CH, pixel, half, ncand,
                                                                            cnt1, cnt2, cval);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_candidate_features_band_multi(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t CH,
                                                 const uint32_t *pixel, const int32_t *half, int32_t ncand, uint32_t *cnt1,
                                                 uint32_t *cnt2, double *cval, void *stream) {
    MST_RANGE("tail: mst_candidate_features_band_multi");
    if (ncand == 0) return MST_OK;
    if (!band || !starts || !pixel || !half || !cnt1 || !cnt2 || !cval || n <= 0 || dpx < 0 || CH <= 0 || ncand < 0)
        return mst::fail(MST_E_ARG, "mst_candidate_features_band_multi: bad argument");
    features_band_kernel<<<(ncand + 3) / 4, 256, 0, mst::as_stream(stream)>>>(band, n, dpx, 0, starts, CH, pixel, half, ncand,
                                                                              cnt1, cnt2, cval);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_gather_diagonals_band(const double *band, int64_t n, int32_t dpx, int64_t start, int32_t CH,
                                         const int32_t *diag_k, int32_t nd, double *out, void *stream) {
    if (nd == 0) return MST_OK;
    if (!band || !diag_k || !out || CH <= 0 || n <= 0 || dpx < 0 || nd < 0 || nd > 65535)
        return mst::fail(MST_E_ARG, "mst_gather_diagonals_band: bad argument");
    diag_band_kernel<<<dim3((CH + 255) / 256, nd), 256, 0, mst::as_stream(stream)>>>(band, n, dpx, start, CH, diag_k, out);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_found_pvalues(const mst_found *found, uint32_t found_cap, const uint32_t *found_count,
                                 const uint32_t *nz_count, const double *level_stats, int32_t B, int32_t n_tested,
                                 double *pval, double *fit, void *stream) {
    MST_RANGE("finish: mst_found_pvalues");
    if (!found || !found_count || !nz_count || !level_stats || !pval || !fit || B <= 0 || B > 65535 ||
        n_tested <= 0 || n_tested > MST_MAX_TESTED)
        return mst::fail(MST_E_ARG, "mst_found_pvalues: bad argument");
    hipStream_t s = mst::as_stream(stream);
    int *d_flags = nullptr;
    MST_HIP(hipMallocAsync((void **)&d_flags, 16, s));
    fit_kernel<<<1, 256, 0, s>>>(level_stats, nz_count, n_tested, fit, d_flags, found_count, found_cap, nullptr, B);
    MST_LAUNCH_CHECK();
    const int gx = (int)((found_cap + 255) / 256 < 256 ? (found_cap + 255) / 256 : 256);
    pvalue_kernel<<<dim3(gx > 0 ? gx : 1, B), 256, 0, s>>>(found, found_cap, found_count, fit, pval, nullptr, nullptr, nullptr, 0);
    MST_LAUNCH_CHECK();
    int flags = 0;
    MST_HIP(hipMemcpyAsync(&flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, s));
    MST_HIP(hipStreamSynchronize(s));
    MST_HIP(hipFreeAsync(d_flags, s));
    if (flags & 1)
        return mst::fail(MST_E_OVERFLOW, "found-pixel capacity %u exceeded in at least one block", found_cap);
#ifdef MST_PROFILE
    if (getenv("MST_IGNORE_NONFINITE")) flags &= ~2;        // PROFILE builds only: timing ablations produce garbage statistics
#endif
    if (flags & 2)
        return mst::fail(MST_E_NONFINITE, "non-finite DoG statistics (input block holds NaN/inf)");
    return MST_OK;
}

extern "C" int mst_found_summary_status(const void *summary_host, uint32_t found_cap) {
    if (!summary_host) return mst::fail(MST_E_ARG, "mst_found_summary_status: bad argument");
    int dflags = 0;
    memcpy(&dflags, summary_host, sizeof(int));
    if (dflags & 1) return mst::fail(MST_E_OVERFLOW, "found-pixel capacity %u exceeded in at least one block", found_cap);
    if (dflags & 2) return mst::fail(MST_E_NONFINITE, "non-finite DoG statistics (input block holds NaN/inf)");
    return MST_OK;
}

extern "C" uint64_t mst_found_summary_bytes(int32_t B) {
    if (B <= 0) return 0;
    return 16 + 8 * (uint64_t)((B + 1) / 2) * 2 + sizeof(double) * 2 * MST_MAX_TESTED * (uint64_t)B;
}

extern "C" int mst_found_finish(const mst_found *found, uint32_t found_cap, const uint32_t *found_count,
                                const uint32_t *nz_count, const double *level_stats, int32_t B, int32_t n_tested, double *pval,
                                double *fit, uint32_t pack_pitch, int32_t *pix_out, uint8_t *lvl_out, double *pv_out,
                                void *scratch_dev, void *summary_host, int32_t *pix_host, uint8_t *lvl_host, double *pv_host,
                                int32_t flags, void *stream) {
    MST_RANGE("finish: mst_found_finish");
    if (!found || !found_count || !nz_count || !level_stats || !pval || !fit || !scratch_dev || !summary_host || B <= 0 ||
        B > 65535 || n_tested <= 0 || n_tested > MST_MAX_TESTED)
        return mst::fail(MST_E_ARG, "mst_found_finish: bad argument");
    if (pack_pitch > 0 && (!pix_out || !lvl_out || !pv_out))
        return mst::fail(MST_E_ARG, "mst_found_finish: pack_pitch > 0 needs pix_out, lvl_out and pv_out");
    if (pix_host && (pack_pitch == 0 || !lvl_host || !pv_host))
        return mst::fail(MST_E_ARG, "mst_found_finish: the record prefetch needs pack_pitch > 0 and all three host arrays");
    hipStream_t s = mst::as_stream(stream);
    char *d_sum = static_cast<char *>(scratch_dev);           // device image of the summary (mst_found_summary_bytes(B))
    int *d_flags = reinterpret_cast<int *>(d_sum);
    auto enqueue = [&]() -> int {
        fit_kernel<<<1, 256, 0, s>>>(level_stats, nz_count, n_tested, fit, d_flags, found_count, found_cap, d_sum, B);
        MST_LAUNCH_CHECK();
        const int gx = (int)((found_cap + 255) / 256 < 256 ? (found_cap + 255) / 256 : 256);
        pvalue_kernel<<<dim3(gx > 0 ? gx : 1, B), 256, 0, s>>>(found, found_cap, found_count, fit, pval,
                                                               pack_pitch ? pix_out : nullptr, lvl_out, pv_out, pack_pitch);
        MST_LAUNCH_CHECK();
        // ONE copy for everything the host needs before it can size its downloads: flags, record counts, tested-pixel counts, fits
        MST_HIP(hipMemcpyAsync(summary_host, d_sum, mst_found_summary_bytes(B), hipMemcpyDeviceToHost, s));
        // ... and, speculatively, the packed records (the caller sized pack_pitch to its guess of the largest count: when the
        // counts in the summary confirm it, the records are on the host after this call's single synchronisation)
        if (pix_host) {
            const size_t m = (size_t)B * pack_pitch;
            MST_HIP(hipMemcpyAsync(pix_host, pix_out, m * 4, hipMemcpyDeviceToHost, s));
            MST_HIP(hipMemcpyAsync(lvl_host, lvl_out, m, hipMemcpyDeviceToHost, s));
            MST_HIP(hipMemcpyAsync(pv_host, pv_out, m * 8, hipMemcpyDeviceToHost, s));
        }
        return MST_OK;
    };
    // MST_FLAG_GRAPH: a call that repeats with every argument unchanged (a caller that keeps its buffers between launches: the
    // engine's single-launch path) is captured into a hipGraph the second time it is seen and replayed afterwards: the
    // stream operations above become one launch, and the graph's nodes follow each other without a dispatch gap -- they are what
    // stands between the fused kernel's end and the host's wake-up of a SMALL launch (chr21 @ 5 kb: 1.88 -> 1.85 ms per step).
    // Opt-in because it costs large pipelined launches: with the finish of one group replayed as a graph next to the fused
    // kernel of the next group, that kernel ran 3 % slower (measured A/B on one box: 17.76 -> 17.19 Gpix/s).  Per host thread;
    // not on the legacy default stream (it cannot be captured); not in PROFILE builds.
    bool done = false;
#ifndef MST_PROFILE
    struct FinishGraph {
        std::vector<int64_t> sig;
        hipGraphExec_t exec = nullptr;
        hipEvent_t ev = nullptr;
        int seen = 0;
        unsigned long long stamp = 0;
        void drop() {
            if (exec) {
                if (ev) (void)hipEventSynchronize(ev);
                (void)hipGraphExecDestroy(exec);
                exec = nullptr;
            }
        }
        ~FinishGraph() {
            drop();
            if (ev) (void)hipEventDestroy(ev);
        }
    };
    static thread_local FinishGraph fcache[6];
    static thread_local unsigned long long fstamp = 0;
    static const bool graphs_off = [] {
        const char *e = getenv("MUSTACHE_NO_GRAPHS");        // diagnostic switch: 1 = no graphs at all, "finish" = none here
        return e && *e && *e != '0' && *e != 'l';
    }();
    if ((flags & MST_FLAG_GRAPH) && s != nullptr && !graphs_off) {
        int dev = 0;
        MST_HIP(hipGetDevice(&dev));
        std::vector<int64_t> sig;
        for (const void *p : {(const void *)found, (const void *)found_count, (const void *)nz_count, (const void *)level_stats,
                              (const void *)pval, (const void *)fit, (const void *)pix_out, (const void *)lvl_out,
                              (const void *)pv_out, (const void *)scratch_dev, (const void *)summary_host, (const void *)pix_host,
                              (const void *)lvl_host, (const void *)pv_host})
            sig.push_back((int64_t)(intptr_t)p);
        for (int64_t v : {(int64_t)found_cap, (int64_t)B, (int64_t)n_tested, (int64_t)pack_pitch, (int64_t)dev})
            sig.push_back(v);
        FinishGraph *g = nullptr;
        for (FinishGraph &e : fcache)
            if (e.seen && e.sig == sig) g = &e;
        mst::note("found_finish graph B=%d cap=%u pitch=%u found=%p count=%p nz=%p stats=%p fit=%p scratch=%p host=%p -> %s", B, found_cap,
                  pack_pitch, (const void *)found, (const void *)found_count, (const void *)nz_count, (const void *)level_stats,
                  (const void *)fit, scratch_dev, summary_host, g && g->exec ? "REPLAY" : (g ? "CAPTURE" : "first sight"));
        if (g && g->exec) {
            g->stamp = ++fstamp;
            MST_HIP(hipGraphLaunch(g->exec, s));
            MST_HIP(hipEventRecord(g->ev, s));
            done = true;
        } else if (!g) {
            g = &fcache[0];
            for (FinishGraph &e : fcache)
                if (e.stamp < g->stamp) g = &e;
            g->drop();
            g->sig = sig;
            g->seen = 1;
            g->stamp = ++fstamp;
        } else {
            g->stamp = ++fstamp;
            if (!g->ev) MST_HIP(hipEventCreateWithFlags(&g->ev, hipEventDisableTiming));
            if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int erc = enqueue();
                hipGraph_t graph = nullptr;
                const hipError_t ee = hipStreamEndCapture(s, &graph);
                if (erc != MST_OK || ee != hipSuccess || !graph) {
                    if (graph) (void)hipGraphDestroy(graph);
                    g->seen = 0;
                    if (erc != MST_OK) return erc;
                    return mst::fail(MST_E_HIP, "mst_found_finish: graph capture failed: %s", hipGetErrorString(ee));
                }
                const hipError_t ie = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ie != hipSuccess) {
                    g->exec = nullptr;
                    g->seen = 0;
                    return mst::fail(MST_E_HIP, "mst_found_finish: graph instantiation failed: %s", hipGetErrorString(ie));
                }
                MST_HIP(hipGraphLaunch(g->exec, s));
                MST_HIP(hipEventRecord(g->ev, s));
                done = true;
            } else {
                (void)hipGetLastError();
                g->seen = 0;
            }
        }
    }
#endif
    if (!done) {
        if (!(flags & MST_FLAG_GRAPH))
            mst::note("found_finish plain B=%d cap=%u pitch=%u found=%p count=%p nz=%p stats=%p scratch=%p", B, found_cap, pack_pitch,
                      (const void *)found, (const void *)found_count, (const void *)nz_count, (const void *)level_stats, scratch_dev);
        const int erc = enqueue();
        if (erc != MST_OK) return erc;
    }
    if (flags & MST_FLAG_NO_WAIT) return MST_OK;             // the caller synchronises and asks mst_found_summary_status
    MST_HIP(hipStreamSynchronize(s));
    int dflags = 0;
    memcpy(&dflags, summary_host, sizeof(int));
    mst::note("found_finish summary flags=%d", dflags);
    if (dflags && getenv("MUSTACHE_GRAPH_DEBUG")) {
        std::vector<double> hs((size_t)B * MST_MAX_TESTED * 2);
        std::vector<uint32_t> hn((size_t)B);
        (void)hipMemcpy(hs.data(), level_stats, hs.size() * 8, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hn.data(), nz_count, hn.size() * 4, hipMemcpyDeviceToHost);
        unsigned devhdr[4] = {0, 0, 0, 0};
        (void)hipMemcpy(devhdr, d_sum, 16, hipMemcpyDeviceToHost);
        const unsigned *hh = static_cast<const unsigned *>(summary_host);
        fprintf(stderr, "[nonfinite] header host %#x %#x %#x %#x | device now %#x %#x %#x %#x\n", hh[0], hh[1], hh[2], hh[3], devhdr[0],
                devhdr[1], devhdr[2], devhdr[3]);
        const char *sh = static_cast<const char *>(summary_host);
        const size_t cw = 8 * (size_t)((B + 1) / 2);
        const uint32_t *s_cnt = reinterpret_cast<const uint32_t *>(sh + 16), *s_nz = reinterpret_cast<const uint32_t *>(sh + 16 + cw);
        const double *s_fit = reinterpret_cast<const double *>(sh + 16 + 2 * cw);
        for (int b = 0; b < B; ++b) {
            fprintf(stderr, "[nonfinite] block %d: summary count %u nz %u | memory nz %u\n", b, s_cnt[b], s_nz[b], hn[(size_t)b]);
            for (int t = 0; t < MST_MAX_TESTED; ++t)
                fprintf(stderr, "[nonfinite]   level %d: summary loc %g scale %g | memory min %g sum %g\n", t,
                        s_fit[((size_t)b * MST_MAX_TESTED + t) * 2], s_fit[((size_t)b * MST_MAX_TESTED + t) * 2 + 1],
                        hs[((size_t)b * MST_MAX_TESTED + t) * 2], hs[((size_t)b * MST_MAX_TESTED + t) * 2 + 1]);
        }
        mst::dump_notes();
    }
    if (dflags & 1)
        return mst::fail(MST_E_OVERFLOW, "found-pixel capacity %u exceeded in at least one block", found_cap);
#ifdef MST_PROFILE
    if (getenv("MST_IGNORE_NONFINITE")) dflags &= ~2;        // PROFILE builds only: timing ablations produce garbage statistics
#endif
    if (dflags & 2)
        return mst::fail(MST_E_NONFINITE, "non-finite DoG statistics (input block holds NaN/inf)");
    return MST_OK;
}

extern "C" int mst_candidate_features(const double *c, const uint8_t *nz, int32_t CH, int32_t b,
                                      const uint32_t *pixel, const int32_t *half, int32_t n, uint32_t *cnt1,
                                      uint32_t *cnt2, double *cval, void *stream) {
    if (n == 0) return MST_OK;
    if (!c || !nz || !pixel || !half || !cnt1 || !cnt2 || !cval || CH <= 0 || b < 0 || n < 0)
        return mst::fail(MST_E_ARG, "mst_candidate_features: bad argument");
    features_kernel<<<(n + 3) / 4, 256, 0, mst::as_stream(stream)>>>(c, nz, CH, b, pixel, half, n, cnt1, cnt2, cval);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_gather_diagonals(const double *c, int32_t CH, int32_t b, const int32_t *diag_k, int32_t n,
                                    double *out, void *stream) {
    if (n == 0) return MST_OK;
    if (!c || !diag_k || !out || CH <= 0 || b < 0 || n < 0 || n > 65535)
        return mst::fail(MST_E_ARG, "mst_gather_diagonals: bad argument");
    diag_kernel<<<dim3((CH + 255) / 256, n), 256, 0, mst::as_stream(stream)>>>(c, CH, b, diag_k, out);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

// ---- clustering of the surviving candidates (mustache.py:830-848) --------------------------------------------------------
// The reference paints every candidate and its 8 neighbours into a label matrix, labels the 8-connected components
// (scipy.ndimage.label, numbered in raster order of their first pixel) and reports, per component, the FIRST arg-min of o in
// raster order over all member pixels (o = q at found pixels, >= 1 elsewhere).  On the record lists that is:
//   * two candidates share a component iff a chain of candidates links them with Chebyshev steps <= 3 (their 3 x 3 halos
//     touch or overlap);
//   * the raster-first pixel of a component is the top-left halo pixel of its raster-first candidate, so the components
//     are numbered in the order of their first candidates;
//   * a member pixel can only be the arg-min if its o is below 1, i.e. if it is a SELECTED record (q < pt, candidate or
//     not: the sparsity / diagonal-mean filters drop candidates, not their q) within Chebyshev distance 1 of a candidate.
// One workgroup per block.  Inputs per block: its selected records sorted by pixel (sel_pix, sel_q) and the positions of the
// surviving candidates among them (cand_pos, ascending).  Scratch per candidate: label, min q (bit pattern), min pixel.
namespace {

constexpr int kClThreads = 256;

__device__ __forceinline__ int cl_lower_bound(const uint32_t *__restrict__ pix, const uint32_t *__restrict__ pos, int n,
                                               uint32_t key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pix[pos[mid]] < key) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// first candidate (index into the block's candidate list) within Chebyshev distance R of (x, y), or -1; with ALL != 0 the
// minimum label over all of them is returned instead (labels: lab[])
template <int R, bool MIN_LABEL>
__device__ __forceinline__ int cl_probe(const uint32_t *__restrict__ spix, const uint32_t *__restrict__ cpos, int nc, int CH,
                                        int x, int y, const uint32_t *__restrict__ lab, int init) {
    int best = init;
    const int y_lo = y - R > 0 ? y - R : 0, y_hi = y + R < CH - 1 ? y + R : CH - 1;
    for (int r = x - R; r <= x + R; ++r) {
        if (r < 0 || r >= CH) continue;
        const uint32_t k_lo = (uint32_t)r * (uint32_t)CH + (uint32_t)y_lo, k_hi = (uint32_t)r * (uint32_t)CH + (uint32_t)y_hi;
        for (int j = cl_lower_bound(spix, cpos, nc, k_lo); j < nc && spix[cpos[j]] <= k_hi; ++j) {
            if (!MIN_LABEL) return j;
            const int l = (int)lab[j];
            best = l < best ? l : best;
        }
    }
    return best;
}

__global__ void __launch_bounds__(kClThreads)
cluster_kernel(const uint32_t *__restrict__ sel_pix, const double *__restrict__ sel_q, const uint32_t *__restrict__ sel_off,
               const uint32_t *__restrict__ cand_pos, const uint32_t *__restrict__ cand_off, int CH, uint32_t *__restrict__ lab_all,
               unsigned long long *__restrict__ minq_all, uint32_t *__restrict__ minpix_all, uint32_t *__restrict__ rep_pos,
               uint32_t *__restrict__ rep_count) {
    __shared__ int changed;
    __shared__ uint32_t scan[kClThreads];
    const int b = blockIdx.x, tid = threadIdx.x;
    const uint32_t s0 = sel_off[b], c0 = cand_off[b];
    const int ns = (int)(sel_off[b + 1] - s0), nc = (int)(cand_off[b + 1] - c0);
    if (nc == 0) {
        if (tid == 0) rep_count[b] = 0;
        return;
    }
    const uint32_t *spix = sel_pix + s0;
    const double *sq = sel_q + s0;
    const uint32_t *cpos = cand_pos + c0;
    uint32_t *lab = lab_all + c0;
    unsigned long long *minq = minq_all + c0;
    uint32_t *minpix = minpix_all + c0;
    for (int i = tid; i < nc; i += kClThreads) {
        lab[i] = (uint32_t)i;
        minq[i] = ~0ull;
        minpix[i] = ~0u;
    }
    __syncthreads();
    // components: minimum-label propagation over the Chebyshev <= 3 links, with pointer jumping; a component's label
    // converges to the index of its raster-first candidate
    for (;;) {
        if (tid == 0) changed = 0;
        __syncthreads();
        for (int i = tid; i < nc; i += kClThreads) {
            const uint32_t p = spix[cpos[i]];
            const int cur = (int)lab[i];
            const int m = cl_probe<3, true>(spix, cpos, nc, CH, (int)(p / (uint32_t)CH), (int)(p % (uint32_t)CH), lab, cur);
            if (m < cur) {
                atomicMin(&lab[i], (uint32_t)m);
                atomicMin(&lab[cur], (uint32_t)m);          // hook the old root as well
                changed = 1;
            }
        }
        __syncthreads();
        for (int i = tid; i < nc; i += kClThreads) {         // pointer jumping
            uint32_t l = lab[i];
            while (lab[l] != l) l = lab[l];
            lab[i] = l;
        }
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    // per component: min q over the selected records inside its halo, then the raster-first record among the ties
    for (int s = tid; s < ns; s += kClThreads) {
        const uint32_t p = spix[s];
        const int j = cl_probe<1, false>(spix, cpos, nc, CH, (int)(p / (uint32_t)CH), (int)(p % (uint32_t)CH), nullptr, -1);
        if (j >= 0) atomicMin(&minq[lab[j]], (unsigned long long)__double_as_longlong(sq[s]));
    }
    __syncthreads();
    for (int s = tid; s < ns; s += kClThreads) {
        const uint32_t p = spix[s];
        const int j = cl_probe<1, false>(spix, cpos, nc, CH, (int)(p / (uint32_t)CH), (int)(p % (uint32_t)CH), nullptr, -1);
        if (j >= 0 && (unsigned long long)__double_as_longlong(sq[s]) == minq[lab[j]]) atomicMin(&minpix[lab[j]], p);
    }
    __syncthreads();
    // representatives in component order (= ascending root index): position of the winning record among the block's records
    const int per = (nc + kClThreads - 1) / kClThreads;
    const int i0 = tid * per, i1 = i0 + per < nc ? i0 + per : nc;
    uint32_t mine = 0;
    for (int i = i0; i < i1; ++i) mine += lab[i] == (uint32_t)i ? 1u : 0u;
    scan[tid] = mine;
    __syncthreads();
    for (int o = 1; o < kClThreads; o <<= 1) {
        const uint32_t v = tid >= o ? scan[tid - o] : 0u;
        __syncthreads();
        scan[tid] += v;
        __syncthreads();
    }
    uint32_t k = scan[tid] - mine;
    for (int i = i0; i < i1; ++i) {
        if (lab[i] != (uint32_t)i) continue;
        const uint32_t want = minpix[i];
        int lo = 0, hi = ns;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (spix[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        rep_pos[c0 + k] = (uint32_t)lo;
        ++k;
    }
    if (tid == kClThreads - 1) rep_count[b] = scan[tid];
}

}  // namespace

extern "C" uint64_t mst_cluster_workspace_bytes(uint32_t n_candidates) { return 16ull * (uint64_t)(n_candidates ? n_candidates : 1); }

extern "C" int mst_cluster_representatives(const uint32_t *sel_pix, const double *sel_q, const uint32_t *sel_off,
                                           const uint32_t *cand_pos, const uint32_t *cand_off, int32_t B, int32_t CH,
                                           uint32_t n_candidates, uint32_t *rep_pos, uint32_t *rep_count, void *workspace,
                                           uint64_t workspace_bytes, void *stream) {
    MST_RANGE("tail: mst_cluster_representatives");
    if (B <= 0) return MST_OK;
    if (!sel_pix || !sel_q || !sel_off || !cand_pos || !cand_off || !rep_pos || !rep_count || !workspace || CH <= 0 ||
        (int64_t)CH * CH > 0xFFFFFFFFLL)
        return mst::fail(MST_E_ARG, "mst_cluster_representatives: bad argument");
    if (workspace_bytes < mst_cluster_workspace_bytes(n_candidates))
        return mst::fail(MST_E_ARG, "mst_cluster_representatives: workspace too small");
    char *w = reinterpret_cast<char *>(workspace);
    unsigned long long *minq = reinterpret_cast<unsigned long long *>(w);
    uint32_t *lab = reinterpret_cast<uint32_t *>(w + 8ull * (n_candidates ? n_candidates : 1));
    uint32_t *minpix = lab + (n_candidates ? n_candidates : 1);
    cluster_kernel<<<B, kClThreads, 0, mst::as_stream(stream)>>>(sel_pix, sel_q, sel_off, cand_pos, cand_off, CH, lab, minq,
                                                                  minpix, rep_pos, rep_count);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
