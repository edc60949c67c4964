// Two-sample (differential) additions, reference mustache/diff_mustache.py:260-569 (gfx950).
// The per-sample sigma loops are the single-sample fused kernel run on both blocks.  What is specific to the
// differential caller is small:
//   mst_diff_image      :262-276  nz = nz1 & nz2,  c = (c1 - c2) on nz else 0   (on the already filled blocks)
//   mst_masked_normfit  :371      norm.fit(Lc[nz]) = (mean, sqrt(mean((x-mean)^2))) of Lc = a - b over the mask
//   mst_pair_pvalues    :372-385  two-sided normal p-value of Lc at the found pixels
// Reference quirk kept on purpose: diff_mustache never advances the difference image's DoG inside its level loop
// (`Lc = Gc - Gn` is assigned once per octave at :336; the loop recomputes only `Ln`, :363), so every tested level of
// an octave scores the difference with the same image D_2 = G_2 - G_3.  The host therefore blurs the difference image
// at exactly two sigmas per octave (mst_gauss_blur) and passes those two images here.
#include <cmath>
#include "mst_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kParts = 256;     // partial sums per image: fixed -> deterministic reduction order

__global__ void __launch_bounds__(kThreads)
diff_image_kernel(const double *__restrict__ c1, const double *__restrict__ c2, const uint8_t *__restrict__ nz1,
                  const uint8_t *__restrict__ nz2, int64_t npx, double *__restrict__ cd, uint8_t *__restrict__ nzb,
                  uint32_t *__restrict__ count) {
    const int b = blockIdx.y;
    const int64_t base = (int64_t)b * npx;
    uint32_t local = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        const bool t = nz1[base + i] && nz2[base + i];
        cd[base + i] = t ? c1[base + i] - c2[base + i] : 0.0;
        nzb[base + i] = t ? 1 : 0;
        local += t ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count + b, local);
}

// MODE 0: sum of (a-b) over the mask; MODE 1: sum of ((a-b) - mean)^2.  One partial per workgroup.
template <int MODE>
__global__ void __launch_bounds__(kThreads)
masked_partial_kernel(const double *__restrict__ a, const double *__restrict__ bimg, const uint8_t *__restrict__ mask,
                      int64_t npx, const double *__restrict__ fit, double *__restrict__ partial) {
    __shared__ double sh[kThreads];
    const int b = blockIdx.y;
    const int64_t base = (int64_t)b * npx;
    const double mean = MODE == 1 ? fit[2 * b] : 0.0;
    double acc = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npx; i += stride) {
        if (mask[base + i]) {
            const double d = a[base + i] - bimg[base + i];
            if (MODE == 0) acc = acc + d;
            else { const double t = d - mean; acc = acc + t * t; }
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[(size_t)b * kParts + blockIdx.x] = sh[0];
}

template <int MODE>
__global__ void __launch_bounds__(kParts)
masked_finish_kernel(const double *__restrict__ partial, const uint32_t *__restrict__ count, double *__restrict__ fit) {
    __shared__ double sh[kParts];
    const int b = blockIdx.x;
    sh[threadIdx.x] = partial[(size_t)b * kParts + threadIdx.x];
    __syncthreads();
    for (int s = kParts / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = sh[threadIdx.x] + sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double n = (double)count[b];
        if (MODE == 0) fit[2 * b] = sh[0] / n;                 // loc = data.mean()
        else fit[2 * b + 1] = sqrt(sh[0] / n);                 // scale = sqrt(((data - loc)**2).mean())
    }
}

// scipy.special.ndtr (cephes): 0.5 + 0.5 erf(x/sqrt2) near 0, else 0.5 erfc(|x|/sqrt2), mirrored for x > 0
__device__ __forceinline__ double ndtr(double a) {
    const double x = a * 0.70710678118654752440;
    const double z = fabs(x);
    if (z < 0.70710678118654752440) return 0.5 + 0.5 * erf(x);
    double y = 0.5 * erfc(z);
    if (x > 0) y = 1.0 - y;
    return y;
}

__global__ void __launch_bounds__(kThreads)
pair_pvalue_kernel(const mst_found *__restrict__ found, uint32_t found_cap, const uint32_t *__restrict__ found_count,
                   const double *__restrict__ g2, const double *__restrict__ g3, const double *__restrict__ fit,
                   int B, int64_t npx, int tested_per_octave, int sample_offset, double *__restrict__ ppair) {
    const int b = blockIdx.y;                       // block pair index
    const int fb = b + sample_offset;               // where this sample's records live
    const uint32_t n = found_count[fb];
    if (n > found_cap) return;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const mst_found rec = found[(size_t)fb * found_cap + i];
        const int oct = ((int)rec.level - 1) / tested_per_octave;
        const size_t at = ((size_t)oct * B + b) * npx + rec.pixel;
        const double x = g2[at] - g3[at];
        const double loc = fit[2 * ((size_t)oct * B + b)], scale = fit[2 * ((size_t)oct * B + b) + 1];
        double cdf = ndtr((x - loc) / scale);
        if (!isfinite(cdf)) cdf = 1.0;                          // nan_to_num(..., nan=1, posinf=1, neginf=1)  (:380)
        if (cdf > 0.5) cdf = 1.0 - cdf;                         // (:381)
        ppair[(size_t)fb * found_cap + i] = cdf * 2.0;          // (:382)
    }
}

}  // namespace

extern "C" int mst_diff_image(const double *c1, const double *c2, const uint8_t *nz1, const uint8_t *nz2, int32_t B,
                              int32_t CH, double *cd, uint8_t *nzb, uint32_t *nzb_count, void *stream) {
    if (!c1 || !c2 || !nz1 || !nz2 || !cd || !nzb || !nzb_count || B <= 0 || B > 65535 || CH <= 0)
        return mst::fail(MST_E_ARG, "mst_diff_image: bad argument");
    hipStream_t s = mst::as_stream(stream);
    MST_HIP(hipMemsetAsync(nzb_count, 0, sizeof(uint32_t) * B, s));
    diff_image_kernel<<<dim3(1024, B), kThreads, 0, s>>>(c1, c2, nz1, nz2, (int64_t)CH * CH, cd, nzb, nzb_count);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_masked_normfit(const double *a, const double *b, const uint8_t *mask, const uint32_t *mask_count,
                                  int32_t B, int64_t npx, double *fit, void *workspace, uint64_t workspace_bytes,
                                  void *stream) {
    if (!a || !b || !mask || !mask_count || !fit || !workspace || B <= 0 || B > 65535 || npx <= 0 ||
        workspace_bytes < sizeof(double) * (size_t)B * kParts)
        return mst::fail(MST_E_ARG, "mst_masked_normfit: bad argument (workspace >= 2048 * B bytes)");
    hipStream_t s = mst::as_stream(stream);
    double *partial = reinterpret_cast<double *>(workspace);
    masked_partial_kernel<0><<<dim3(kParts, B), kThreads, 0, s>>>(a, b, mask, npx, fit, partial);
    MST_LAUNCH_CHECK();
    masked_finish_kernel<0><<<B, kParts, 0, s>>>(partial, mask_count, fit);
    MST_LAUNCH_CHECK();
    masked_partial_kernel<1><<<dim3(kParts, B), kThreads, 0, s>>>(a, b, mask, npx, fit, partial);
    MST_LAUNCH_CHECK();
    masked_finish_kernel<1><<<B, kParts, 0, s>>>(partial, mask_count, fit);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_pair_pvalues(const mst_found *found, uint32_t found_cap, const uint32_t *found_count,
                                const double *g2, const double *g3, const double *fit, int32_t B, int32_t CH,
                                int32_t n_octaves, int32_t tested_per_octave, int32_t sample_offset, double *ppair,
                                void *stream) {
    if (!found || !found_count || !g2 || !g3 || !fit || !ppair || B <= 0 || B > 65535 || CH <= 0 || n_octaves <= 0 ||
        tested_per_octave <= 0 || sample_offset < 0)
        return mst::fail(MST_E_ARG, "mst_pair_pvalues: bad argument");
    const int gx = (int)((found_cap + 255) / 256 < 256 ? (found_cap + 255) / 256 : 256);
    pair_pvalue_kernel<<<dim3(gx > 0 ? gx : 1, B), kThreads, 0, mst::as_stream(stream)>>>(
        found, found_cap, found_count, g2, g3, fit, B, (int64_t)CH * CH, tested_per_octave, sample_offset, ppair);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
