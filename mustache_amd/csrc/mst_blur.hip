// General separable Gaussian (any radius <= MST_MAX_RADIUS, any image size) -- bring-up / parity kernel.
//   mst_gauss_blur <- scipy.ndimage.gaussian_filter as the reference calls it (mustache/mustache.py:719, :725,
//   :734, :751): axis 0 pass, then axis 1 pass, mode='reflect', float64, and the tap order of SciPy's C
//   correlate1d on a symmetric kernel:  t = x[c]*w0;  for j = r..1: t += (x[c-j] + x[c+j]) * w[j]   (no FMA;
//   this file is compiled with -ffp-contract=off).
// One thread per output sample, taps served from L1/L2; the production path is the fused LDS-tiled kernel in
// mst_scale_space.hip -- this one exists so single levels can be checked against SciPy at any sigma.
#include "mst_common.h"

namespace {

struct Taps {
    double w[MST_MAX_RADIUS + 1];
};

__device__ __forceinline__ int reflect(int i, int n) {
    // half-sample symmetric, any number of folds:  d c b a | a b c d | d c b a
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

template <int AXIS>
__global__ void __launch_bounds__(256)
blur_axis_kernel(const double *__restrict__ in, double *__restrict__ out, int H, int W, Taps taps, int r) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= W) return;
    const int64_t base = (int64_t)blockIdx.z * H * W;
    const double *img = in + base;
    double t;
    if (AXIS == 0) {
        t = img[(int64_t)row * W + col] * taps.w[0];
        for (int j = r; j >= 1; --j) {
            const double a = img[(int64_t)reflect(row - j, H) * W + col];
            const double b = img[(int64_t)reflect(row + j, H) * W + col];
            t = t + (a + b) * taps.w[j];
        }
    } else {
        const double *line = img + (int64_t)row * W;
        t = line[col] * taps.w[0];
        for (int j = r; j >= 1; --j) {
            const double a = line[reflect(col - j, W)];
            const double b = line[reflect(col + j, W)];
            t = t + (a + b) * taps.w[j];
        }
    }
    out[base + (int64_t)row * W + col] = t;
}

}  // namespace

extern "C" int mst_gauss_blur(const double *in, double *out, double *tmp, int32_t B, int32_t H, int32_t W,
                              const double *taps, int32_t radius, void *stream) {
    if (!in || !out || !tmp || !taps || B <= 0 || H <= 0 || W <= 0 || radius < 0 || radius > MST_MAX_RADIUS ||
        H > 65535 || B > 65535)
        return mst::fail(MST_E_ARG, "mst_gauss_blur: bad argument (radius <= %d, H, B <= 65535)", MST_MAX_RADIUS);
    Taps t;
    for (int j = 0; j <= MST_MAX_RADIUS; ++j) t.w[j] = j <= radius ? taps[j] : 0.0;
    hipStream_t s = mst::as_stream(stream);
    dim3 grid((W + 255) / 256, H, B);
    blur_axis_kernel<0><<<grid, 256, 0, s>>>(in, tmp, H, W, t, radius);
    MST_LAUNCH_CHECK();
    blur_axis_kernel<1><<<grid, 256, 0, s>>>(tmp, out, H, W, t, radius);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
