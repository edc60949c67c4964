// Small kernels after the sigma loop (gfx950):
//   mst_found_pvalues      <- reference mustache/mustache.py:755-756 (expon.fit + 1 - expon.cdf), found pixels only
//   mst_candidate_features <- reference mustache/mustache.py:800-811, :824 (window densities of nz, c[x, y])
//   mst_gather_diagonals   <- reference mustache/mustache.py:816-823 (diagonals for the diagonal-mean filter)
// All of them touch a few thousand pixels per block; they exist so the tail never pulls a dense block to the host.
#include <cmath>
#include <cstdlib>
#include <vector>
#include "mst_common.h"

namespace {

// {loc, scale} per (block, tested level):  loc = min|D|, scale = mean|D| - loc   (scipy expon.fit,
// scipy/stats/_continuous_distns.py:2134-2147)
__global__ void fit_kernel(const double *__restrict__ level_stats, const uint32_t *__restrict__ nz_count,
                           int n_tested, double *__restrict__ fit, int *__restrict__ flags) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (t >= n_tested) return;
    const double mn = level_stats[((size_t)b * MST_MAX_TESTED + t) * 2];
    const double sm = level_stats[((size_t)b * MST_MAX_TESTED + t) * 2 + 1];
    const double cnt = (double)nz_count[b];
    const double loc = mn;
    const double scale = sm / cnt - loc;
    fit[((size_t)b * MST_MAX_TESTED + t) * 2] = loc;
    fit[((size_t)b * MST_MAX_TESTED + t) * 2 + 1] = scale;
    if (nz_count[b] > 0 && !(isfinite(mn) && isfinite(sm))) atomicOr(flags, 2);
}

// p = 1 - cdf, cdf = -expm1(-x) for x > 0 else 0 (scipy/stats/_distn_infrastructure.py:2127-2139,
// _continuous_distns.py:2087-2088).  SciPy's expm1 is exp(x) - 1 outside |x| <= 0.5; we form the same two
// roundings (E = exp(-x); E - 1; negate; 1 - .) so tiny p-values land on the same 2^-53 grid points.
__global__ void __launch_bounds__(256)
pvalue_kernel(const mst_found *__restrict__ found, uint32_t found_cap, const uint32_t *__restrict__ found_count,
              const double *__restrict__ fit, double *__restrict__ pval, int *__restrict__ flags) {
    const int b = blockIdx.y;
    const uint32_t n = found_count[b];
    if (n > found_cap) {
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicOr(flags, 1);
        return;
    }
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

__global__ void __launch_bounds__(256)
features_band_kernel(const double *__restrict__ band, int64_t n, int dpx, int64_t start, int CH,
                     const uint32_t *__restrict__ pixel, const int32_t *__restrict__ half, int ncand,
                     uint32_t *__restrict__ cnt1, uint32_t *__restrict__ cnt2, double *__restrict__ cval) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= ncand) return;
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

}  // namespace

extern "C" int mst_candidate_features_band(const double *band, int64_t n, int32_t dpx, int64_t start, int32_t CH,
                                           const uint32_t *pixel, const int32_t *half, int32_t ncand, uint32_t *cnt1,
                                           uint32_t *cnt2, double *cval, void *stream) {
    if (ncand == 0) return MST_OK;
    if (!band || !pixel || !half || !cnt1 || !cnt2 || !cval || CH <= 0 || n <= 0 || dpx < 0 || ncand < 0)
        return mst::fail(MST_E_ARG, "mst_candidate_features_band: bad argument");
    features_band_kernel<<<(ncand + 3) / 4, 256, 0, mst::as_stream(stream)>>>(band, n, dpx, start, CH, pixel, half, ncand,
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
    if (!found || !found_count || !nz_count || !level_stats || !pval || !fit || B <= 0 || B > 65535 ||
        n_tested <= 0 || n_tested > MST_MAX_TESTED)
        return mst::fail(MST_E_ARG, "mst_found_pvalues: bad argument");
    hipStream_t s = mst::as_stream(stream);
    int *d_flags = nullptr;
    MST_HIP(hipMallocAsync((void **)&d_flags, sizeof(int), s));
    MST_HIP(hipMemsetAsync(d_flags, 0, sizeof(int), s));
    fit_kernel<<<B, 64, 0, s>>>(level_stats, nz_count, n_tested, fit, d_flags);
    MST_LAUNCH_CHECK();
    const int gx = (int)((found_cap + 255) / 256 < 256 ? (found_cap + 255) / 256 : 256);
    pvalue_kernel<<<dim3(gx > 0 ? gx : 1, B), 256, 0, s>>>(found, found_cap, found_count, fit, pval, d_flags);
    MST_LAUNCH_CHECK();
    int flags = 0;
    MST_HIP(hipMemcpyAsync(&flags, d_flags, sizeof(int), hipMemcpyDeviceToHost, s));
    MST_HIP(hipStreamSynchronize(s));
    MST_HIP(hipFreeAsync(d_flags, s));
    if (flags & 1)
        return mst::fail(MST_E_OVERFLOW, "found-pixel capacity %u exceeded in at least one block", found_cap);
    if ((flags & 2) && !getenv("MST_IGNORE_NONFINITE"))     // (the env hook exists for timing ablations only)
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
