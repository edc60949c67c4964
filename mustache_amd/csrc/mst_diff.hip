// Two-sample (differential) additions, reference mustache/diff_mustache.py:260-569 (gfx950).
// The per-sample sigma loops are the single-sample fused kernel run on both blocks.  What is specific to the
// differential caller is small:
//   mst_diff_image      :262-276  nz = nz1 & nz2,  c = (c1 - c2) on nz else 0   (on the already filled blocks)
//   mst_masked_normfit  :371      norm.fit(Lc[nz]) = (mean, sqrt(mean((x-mean)^2))) of Lc = a - b over the mask
//   mst_pair_pvalues    :372-385  two-sided normal p-value of Lc at the found pixels
// Reference quirk kept on purpose: diff_mustache never advances the difference image's DoG inside its level loop
// (`Lc = Gc - Gn` is assigned once per octave at :336; the loop recomputes only `Ln`, :363), so every tested level of
// an octave scores the difference with the same image D_2 = G_2 - G_3.
//
// Two forms.  Dense (the reference's own seam: diff_mustache(c1, c2, ...) receives dense blocks): mst_diff_image builds the
// difference image, the host blurs it at the two sigmas per octave with mst_gauss_blur, mst_masked_normfit +
// mst_pair_pvalues score it.  Band-direct (what the per-chromosome driver uses): mst_diff_dog_band cuts the two samples'
// tiles straight out of their bands, forms the difference image in LDS, runs the SAME LDS-tiled separable passes as the fused
// sigma-stack kernel (mst_fir.h: SciPy's tap order, no FMA -> G_2, G_3 bit-identical to gaussian_filter) and writes only
// D_2 per octave plus the masked sums of norm.fit: no dense blocks, no difference image, no G_2 / G_3 in HBM.
#include <cmath>
#include <cstring>
#include <vector>
#include "mst_common.h"
#include "mst_fir.h"

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


// ---- band-direct DoG of the difference image ---------------------------------------------------------------------------
struct DiffLevels {
    int n_octaves;
    int radius[16][2];                                  // per octave: radius of G_2 and of G_3
    double taps[16][2][MST_MAX_RADIUS + 1];
};

template <class T, int R>
__device__ __forceinline__ void diff_blur(const double *ct, double *vb, const double (&w)[T::RMAX + 1], int tid,
                                          const double *vsrc, double *vdst, const double *hsrc, double (&g)[T::K]) {
    vpass<T, R>(ct, vb, w, tid, vsrc, vdst, 0);
    __syncthreads();
    hpass<T, R>(hsrc, w, g);
    __syncthreads();                                    // the next V pass overwrites vb
}

template <class T>
__device__ __forceinline__ void diff_blur_dispatch(int r, const double *ct, double *vb, const double (&w)[T::RMAX + 1],
                                                   int tid, const double *vsrc, double *vdst, const double *hsrc,
                                                   double (&g)[T::K]) {
#define MST_CASE(R_)                                                                  \
    case R_:                                                                          \
        if constexpr (R_ <= T::RMAX) diff_blur<T, R_>(ct, vb, w, tid, vsrc, vdst, hsrc, g); \
        break;
    switch (r) {
        MST_CASE(1) MST_CASE(2) MST_CASE(3) MST_CASE(4) MST_CASE(5) MST_CASE(6) MST_CASE(7)
        MST_CASE(8) MST_CASE(9) MST_CASE(10) MST_CASE(11) MST_CASE(12) MST_CASE(13) MST_CASE(14)
        MST_CASE(15) MST_CASE(16) MST_CASE(17) MST_CASE(18) MST_CASE(19) MST_CASE(20) MST_CASE(21)
        MST_CASE(22) MST_CASE(23) MST_CASE(24) MST_CASE(25) MST_CASE(26) MST_CASE(27) MST_CASE(28)
        default: break;
    }
#undef MST_CASE
}

// One workgroup = one RGR x RGC tile of one block pair (no ring: nothing here looks at neighbours of the result).
//   difference image (diff_mustache.py:262-276):  tested_s = raw_s != 0 and off >= 4;  filled_s = 2 where off <= 4 or off >= dpx+1
//                                                 cd = filled_1 - filled_2 where tested_1 and tested_2, else 0
//   per octave:  D = gaussian_filter(cd, sigma_2) - gaussian_filter(cd, sigma_3)      (:315-336)
//   out: D [oct][b][CH][CH];  partial[b][tile][oct] = {sum of D, sum of D^2} over the pixels tested in both samples (norm.fit, :371)
template <class T>
__global__ void __launch_bounds__(T::NT)
diff_dog_kernel(const double *__restrict__ band1, const double *__restrict__ band2, int64_t n, int dpx,
                const int64_t *__restrict__ starts, int CH, int B, const DiffLevels *__restrict__ lv,
                double *__restrict__ dog, double *__restrict__ partial, uint32_t *__restrict__ mask_count, int tiles_x,
                int n_slots, const int32_t *__restrict__ tile_list) {
    constexpr int K = T::K, RGR = T::RGR, RGC = T::RGC, RMAX = T::RMAX;
    extern __shared__ __align__(16) double lds[];
    double *ct = lds;
    double *vb = ct + T::CT_ELEMS;
    double *red = vb + T::VB_ELEMS;                     // 2 * NT doubles: the masked sums' reduction
    const int tid = threadIdx.x, b = blockIdx.y;
    const int per_xcd = gridDim.x >> 3;                 // XCD-aware order, as in the fused kernel
    // a slot is an entry of the host's list of tiles whose pixels can reach the doubly tested band 4 <= col - row <= dpx + 1:
    // the others hold no pixel of either mask (their sums are zero) and no found pixel ever reads their DoG values
    const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (slot >= n_slots) return;
    const int tile = tile_list[slot];
    const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
    const int y0 = ty * RGR, x0 = tx * RGC;
    const int rr = tid % RGR, cg = tid / RGR;
    const int gy = y0 + rr;
    const int64_t start = starts[b];

    // ---- stage the difference image with its reflect halo, transposed (ct[col][row]); the mask of the region goes to vb
    uint8_t *nzb = reinterpret_cast<uint8_t *>(vb);
    const int Y0 = y0 - RMAX, X0 = x0 - RMAX;
    const bool inner = Y0 >= 0 && X0 >= 0 && Y0 + T::CTR <= CH && X0 + T::CTC <= CH;
    auto pixel = [&](double r1, double r2, int off, double &val, bool &both) {
        const bool fill = off <= 4 || off >= dpx + 1;
        both = r1 != 0.0 && r2 != 0.0 && off >= 4;
        val = both ? ((fill ? 2.0 : r1) - (fill ? 2.0 : r2)) : 0.0;
    };
    if (inner) {
        constexpr int ND = T::CTR + T::CTC - 1;
        constexpr int PER = (T::CTR + 63) / 64;
        constexpr int U = 10 / PER > 0 ? 10 / PER : 1;   // diagonals in flight per wave (two loads each)
        for (int q0 = tid >> 6; q0 < ND; q0 += T::NW * U) {
            double a1[U][PER], a2[U][PER];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * T::NW;
                const int dd = q - (T::CTR - 1);
                const int off = X0 - Y0 + dd;
                const int i_lo = dd < 0 ? -dd : 0;
                const int i_hi = T::CTC - dd < T::CTR ? T::CTC - dd : T::CTR;
                const bool in_band = q < ND && off >= 0 && off <= dpx + 1;
                const int64_t at = (int64_t)(in_band ? off : 0) * n + start + Y0;
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int i = i_lo + (tid & 63) + 64 * e;
                    const bool ok = in_band && i < i_hi && start + X0 + i + dd < n;
                    a1[u][e] = ok ? band1[at + i] : 0.0;
                    a2[u][e] = ok ? band2[at + i] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int q = q0 + u * T::NW;
                if (q >= ND) break;
                const int dd = q - (T::CTR - 1);
                const int off = X0 - Y0 + dd;
                const int i_lo = dd < 0 ? -dd : 0;
                const int i_hi = T::CTC - dd < T::CTR ? T::CTC - dd : T::CTR;
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int i = i_lo + (tid & 63) + 64 * e;
                    if (i >= i_hi) continue;
                    const int j = i + dd;
                    double val;
                    bool both;
                    pixel(a1[u][e], a2[u][e], off, val, both);
                    ct[j * T::CTP + i] = val;
                    const int ri = i - RMAX, rj = j - RMAX;
                    if (ri >= 0 && ri < RGR && rj >= 0 && rj < RGC) nzb[ri * RGC + rj] = both ? 1 : 0;
                }
            }
        }
    } else {
        for (int idx = tid; idx < T::CTR * T::CTC; idx += T::NT) {
            const int i = idx / T::CTC, j = idx - i * T::CTC;
            const int uy = Y0 + i, ux = X0 + j;
            const int by = reflect_idx(uy, CH), bx = reflect_idx(ux, CH);
            const int off = bx - by;
            double r1 = 0.0, r2 = 0.0;
            if (off >= 0 && off <= dpx + 1 && start + bx < n) {
                r1 = band1[(int64_t)off * n + start + by];
                r2 = band2[(int64_t)off * n + start + by];
            }
            double val;
            bool both;
            pixel(r1, r2, off, val, both);
            ct[j * T::CTP + i] = val;
            const int ri = i - RMAX, rj = j - RMAX;
            if (ri >= 0 && ri < RGR && rj >= 0 && rj < RGC) {
                const bool inside = uy >= 0 && uy < CH && ux >= 0 && ux < CH;
                nzb[ri * RGC + rj] = (inside && both) ? 1 : 0;
            }
        }
    }
    __syncthreads();
    uint32_t in_mask = 0, both_mask = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int gx = x0 + cg * K + k;
        if (gy < CH && gx < CH) {
            in_mask |= 1u << k;
            if (nzb[rr * RGC + cg * K + k]) both_mask |= 1u << k;
        }
    }
    uint32_t mine = __builtin_popcount(both_mask);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((tid & 63) == 0 && mine) atomicAdd(mask_count + b, mine);         // integer: exact in any order
    __syncthreads();                                                      // nzb is read; vb may be written

    constexpr int V_MAINC = T::NT / (T::RGR / K);
    const int v_rgp = tid / V_MAINC, v_col = tid - v_rgp * V_MAINC;
    const double *vsrc = ct + v_col * T::CTP + v_rgp * K;
    double *vdst = vb + (v_rgp * K) * T::VP + v_col;
    const double *hsrc = vb + rr * T::VP + cg * K;
    double *part = partial + (((size_t)b * n_slots + slot) * lv->n_octaves) * 2;
    for (int o = 0; o < lv->n_octaves; ++o) {
        double g2[K], g3[K];
        {
            double taps[RMAX + 1];
#pragma unroll
            for (int j = 0; j <= RMAX; ++j) taps[j] = lv->taps[o][0][j];
            diff_blur_dispatch<T>(lv->radius[o][0], ct, vb, taps, tid, vsrc, vdst, hsrc, g2);
        }
        {
            // the pointer is laundered so that this blur's taps (up to 29 doubles = 58 scalar registers) are fetched AFTER the
            // first blur has let go of its own -- hoisted above it, the two sets together spilled 64 scalar registers
            // (the default tile, radii <= 8, has registers to spare and is left as it was)
            const DiffLevels *lv3 = lv;
            if constexpr (RMAX > 8) asm volatile("" : "+s"(lv3));
            double taps[RMAX + 1];
#pragma unroll
            for (int j = 0; j <= RMAX; ++j) taps[j] = lv3->taps[o][1][j];
            diff_blur_dispatch<T>(lv3->radius[o][1], ct, vb, taps, tid, vsrc, vdst, hsrc, g3);
        }
        double s1 = 0.0, s2 = 0.0;
        double *out = dog + (((size_t)o * B + b) * CH + gy) * CH + x0 + cg * K;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double dk = g2[k] - g3[k];                               // Lc = Gc - Gn   (:336)
            if ((in_mask >> k) & 1u) out[k] = dk;
            if ((both_mask >> k) & 1u) {
                s1 = s1 + dk;
                s2 = s2 + dk * dk;
            }
        }
        red[tid] = s1;
        red[T::NT + tid] = s2;
        __syncthreads();
        for (int st = T::NT / 2; st > 0; st >>= 1) {                       // fixed tree: deterministic
            if (tid < st) {
                red[tid] = red[tid] + red[tid + st];
                red[T::NT + tid] = red[T::NT + tid] + red[T::NT + tid + st];
            }
            __syncthreads();
        }
        if (tid == 0) {
            part[2 * o] = red[0];
            part[2 * o + 1] = red[T::NT];
        }
        __syncthreads();
    }
}

// partial[b][tile][oct][2] -> fit[oct][b] = {loc, scale} of norm.fit over the doubly tested pixels (:371):
// loc = mean, scale = sqrt(mean(x^2) - loc^2).  (One pass: the DoG of a difference image has |mean| << std, so the
// subtraction costs no accuracy; the reference's two-pass value is reproduced to ~1e-15 relative.)
__global__ void __launch_bounds__(256)
diff_fit_kernel(const double *__restrict__ partial, int ntiles, int n_slots, const int32_t *__restrict__ slot_of_tile,
                int n_oct, int B, const uint32_t *__restrict__ mask_count, double *__restrict__ fit) {
    __shared__ double sa[256], sb[256];
    const int o = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    double a = 0.0, q = 0.0;
    // summation order fixed by the TILE numbering: a tile that was not launched would have contributed {0, 0}
    for (int i = tid; i < ntiles; i += 256) {
        const int sl = slot_of_tile[i];
        if (sl < 0) continue;
        a = a + partial[(((size_t)b * n_slots + sl) * n_oct + o) * 2];
        q = q + partial[(((size_t)b * n_slots + sl) * n_oct + o) * 2 + 1];
    }
    sa[tid] = a;
    sb[tid] = q;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) {
            sa[tid] = sa[tid] + sa[tid + st];
            sb[tid] = sb[tid] + sb[tid + st];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)mask_count[b];
        const double loc = sa[0] / cnt;
        double var = sb[0] / cnt - loc * loc;
        if (var < 0.0) var = 0.0;
        fit[2 * ((size_t)o * B + b)] = loc;
        fit[2 * ((size_t)o * B + b) + 1] = sqrt(var);
    }
}

__global__ void __launch_bounds__(kThreads)
pair_pvalue_dog_kernel(const mst_found *__restrict__ found, uint32_t found_cap, const uint32_t *__restrict__ found_count,
                       const double *__restrict__ dog, const double *__restrict__ fit, int B, int64_t npx,
                       int tested_per_octave, int sample_offset, double *__restrict__ ppair) {
    const int b = blockIdx.y;
    const int fb = b + sample_offset;
    const uint32_t nrec = found_count[fb];
    if (nrec > found_cap) return;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += gridDim.x * blockDim.x) {
        const mst_found rec = found[(size_t)fb * found_cap + i];
        const int oct = ((int)rec.level - 1) / tested_per_octave;
        const double x = dog[((size_t)oct * B + b) * npx + rec.pixel];
        const double loc = fit[2 * ((size_t)oct * B + b)], scale = fit[2 * ((size_t)oct * B + b) + 1];
        double cdf = ndtr((x - loc) / scale);
        if (!isfinite(cdf)) cdf = 1.0;                          // nan_to_num(..., nan=1, posinf=1, neginf=1)  (:380)
        if (cdf > 0.5) cdf = 1.0 - cdf;                         // (:381)
        ppair[(size_t)fb * found_cap + i] = cdf * 2.0;          // (:382)
    }
}

using DiffTile8 = Tile<32, 64, 8>;        // the reference's default octaves: G_2 / G_3 radii 4, 4 and 7, 8
using DiffTile14 = Tile<32, 64, 14>;
using DiffTile28 = Tile<32, 64, 28, 4, 1, false, true>;   // 512 threads x 4 pixels, tight pitches: the sigma loop's wide tile

template <class T>
size_t diff_lds_bytes() { return sizeof(double) * (size_t)(T::CT_ELEMS + T::VB_ELEMS + 2 * T::NT); }
template <class T>
int diff_tiles(int CH) { return ((CH + T::RGR - 1) / T::RGR) * ((CH + T::RGC - 1) / T::RGC); }

size_t diff_align(size_t v) { return (v + 255) / 256 * 256; }

int diff_levels(const mst_levels *lv, DiffLevels *out, int *max_radius) {
    if (!lv || lv->n_octaves < 1 || lv->n_octaves > 16 || lv->levels_per_octave < 3 ||
        lv->n_octaves * lv->levels_per_octave > MST_MAX_LEVELS)
        return mst::fail(MST_E_ARG, "mst_diff_dog_band: bad level table (1..16 octaves, >= 3 levels each)");
    memset(out, 0, sizeof(*out));
    out->n_octaves = lv->n_octaves;
    int mr = 0;
    for (int o = 0; o < lv->n_octaves; ++o)
        for (int q = 0; q < 2; ++q) {
            const int l = o * lv->levels_per_octave + 1 + q;          // sigma_2 and sigma_3 of the octave
            const int r = lv->radius[l];
            if (r < 1 || r > 28) return mst::fail(MST_E_ARG, "mst_diff_dog_band: blur radius %d outside [1, 28]", r);
            out->radius[o][q] = r;
            for (int j = 0; j <= r; ++j) out->taps[o][q][j] = lv->taps[l][j];
            mr = r > mr ? r : mr;
        }
    *max_radius = mr;
    return MST_OK;
}

// tiles (no ring here: a tile owns all its RGR x RGC pixels) that can reach the band, row-major, and the inverse map:
// list[0 .. m) = tile numbers, list[nt + t] = slot of tile t or -1.  Returns m.
template <class T>
int diff_tile_list(int CH, int dpx, int32_t *list) {
    const int tx = (CH + T::RGC - 1) / T::RGC, nt = diff_tiles<T>(CH);
    for (int i = 0; i < 2 * nt; ++i) list[i] = -1;
    int m = 0;
    for (int t = 0; t < nt; ++t) {
        const int y0 = (t / tx) * T::RGR, x0 = (t % tx) * T::RGC;
        const int r_hi = y0 + T::RGR - 1 < CH - 1 ? y0 + T::RGR - 1 : CH - 1;
        const int c_hi = x0 + T::RGC - 1 < CH - 1 ? x0 + T::RGC - 1 : CH - 1;
        if (c_hi - y0 >= 4 && x0 - r_hi <= dpx + 1) {
            list[(size_t)nt + t] = m;
            list[(size_t)m++] = t;
        }
    }
    return m;
}

template <class T>
int diff_dog_launch(const double *band1, const double *band2, int64_t n, int dpx, const int64_t *d_starts, int CH, int B,
                    const DiffLevels *d_lv, int n_oct, double *dog, double *partial, int32_t *d_tiles, int m,
                    uint32_t *mask_count, double *fit, hipStream_t s) {
    static unsigned long long lds_allowed = 0;
    MST_HIP(mst::allow_dynamic_lds(reinterpret_cast<const void *>(&diff_dog_kernel<T>), (int)diff_lds_bytes<T>(),
                                   &lds_allowed));
    const int tx = (CH + T::RGC - 1) / T::RGC, nt = diff_tiles<T>(CH);
    if (m > 0) {
        diff_dog_kernel<T><<<dim3((m + 7) / 8 * 8, B), T::NT, diff_lds_bytes<T>(), s>>>(band1, band2, n, dpx, d_starts, CH, B,
                                                                                      d_lv, dog, partial, mask_count, tx, m,
                                                                                      d_tiles);
        MST_LAUNCH_CHECK();
    }
    diff_fit_kernel<<<dim3(n_oct, B), 256, 0, s>>>(partial, nt, m, d_tiles + nt, n_oct, B, mask_count, fit);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

}  // namespace

extern "C" uint64_t mst_diff_dog_workspace_bytes(int32_t B, int32_t CH, const mst_levels *lv) {
    if (B <= 0 || CH <= 0 || !lv || lv->n_octaves < 1 || lv->n_octaves > 16) return 0;
    const int nt = diff_tiles<DiffTile28>(CH);          // the smallest tile has the most tiles
    return diff_align(sizeof(DiffLevels)) + diff_align(sizeof(int64_t) * (size_t)B) +
           diff_align(sizeof(int32_t) * 2 * (size_t)nt) + sizeof(double) * 2 * (size_t)B * nt * lv->n_octaves;
}

extern "C" int mst_diff_dog_band(const double *band1, const double *band2, int64_t n, int32_t dpx, const int64_t *starts,
                                 int32_t B, int32_t CH, const mst_levels *lv, double *dog, double *fit,
                                 uint32_t *mask_count, void *workspace, uint64_t workspace_bytes, void *stream) {
    MST_RANGE("launch: mst_diff_dog_band");
    if (!band1 || !band2 || !starts || !dog || !fit || !mask_count || !workspace || n <= 0 || dpx < 0 || B <= 0 ||
        B > 65535 || CH <= 0)
        return mst::fail(MST_E_ARG, "mst_diff_dog_band: bad argument");
    DiffLevels h;
    int mr = 0;
    int rc = diff_levels(lv, &h, &mr);
    if (rc != MST_OK) return rc;
    if (workspace_bytes < mst_diff_dog_workspace_bytes(B, CH, lv))
        return mst::fail(MST_E_ARG, "mst_diff_dog_band: workspace too small");
    hipStream_t s = mst::as_stream(stream);
    char *w = reinterpret_cast<char *>(workspace);
    DiffLevels *d_lv = reinterpret_cast<DiffLevels *>(w);
    w += diff_align(sizeof(DiffLevels));
    int64_t *d_starts = reinterpret_cast<int64_t *>(w);
    w += diff_align(sizeof(int64_t) * (size_t)B);
    int32_t *d_tiles = reinterpret_cast<int32_t *>(w);
    w += diff_align(sizeof(int32_t) * 2 * (size_t)diff_tiles<DiffTile28>(CH));
    double *partial = reinterpret_cast<double *>(w);
    // three uploads, each small enough for the runtime's in-queue blit path.  (Round 5 tried ONE 26 KB upload of the three tables:
    // 15 us less in front of the kernel of a six-pair call, but copies of that size go through the copy engine, where they
    // queue behind this stream's fused kernels and hold up every later small copy of the OTHER streams -- the pipelined
    // two-sample genome run lost its overlap, 0.077 -> 0.090 s; LABBOOK R5.5.)
    static thread_local std::vector<int32_t> tiles_h;
    tiles_h.resize(2 * (size_t)diff_tiles<DiffTile28>(CH));
    const int which = mr <= DiffTile8::RMAX ? 0 : (mr <= DiffTile14::RMAX ? 1 : 2);
    const int m = which == 0 ? diff_tile_list<DiffTile8>(CH, dpx, tiles_h.data())
                             : (which == 1 ? diff_tile_list<DiffTile14>(CH, dpx, tiles_h.data()) : diff_tile_list<DiffTile28>(CH, dpx, tiles_h.data()));
    MST_HIP(mst::upload_small(d_lv, &h, sizeof(h), s));
    MST_HIP(mst::upload_small(d_starts, starts, sizeof(int64_t) * B, s));
    MST_HIP(mst::upload_small(d_tiles, tiles_h.data(), sizeof(int32_t) * tiles_h.size(), s));
    MST_HIP(hipMemsetAsync(mask_count, 0, sizeof(uint32_t) * B, s));
    if (which == 0)
        return diff_dog_launch<DiffTile8>(band1, band2, n, dpx, d_starts, CH, B, d_lv, h.n_octaves, dog, partial, d_tiles, m,
                                          mask_count, fit, s);
    if (which == 1)
        return diff_dog_launch<DiffTile14>(band1, band2, n, dpx, d_starts, CH, B, d_lv, h.n_octaves, dog, partial, d_tiles, m,
                                           mask_count, fit, s);
    return diff_dog_launch<DiffTile28>(band1, band2, n, dpx, d_starts, CH, B, d_lv, h.n_octaves, dog, partial, d_tiles, m,
                                       mask_count, fit, s);
}

extern "C" int mst_pair_pvalues_dog(const mst_found *found, uint32_t found_cap, const uint32_t *found_count,
                                    const double *dog, const double *fit, int32_t B, int32_t CH, int32_t n_octaves,
                                    int32_t tested_per_octave, int32_t sample_offset, double *ppair, void *stream) {
    MST_RANGE("finish: mst_pair_pvalues_dog");
    if (!found || !found_count || !dog || !fit || !ppair || B <= 0 || B > 65535 || CH <= 0 || n_octaves <= 0 ||
        tested_per_octave <= 0 || sample_offset < 0)
        return mst::fail(MST_E_ARG, "mst_pair_pvalues_dog: bad argument");
    const int gx = (int)((found_cap + 255) / 256 < 256 ? (found_cap + 255) / 256 : 256);
    pair_pvalue_dog_kernel<<<dim3(gx > 0 ? gx : 1, B), kThreads, 0, mst::as_stream(stream)>>>(
        found, found_cap, found_count, dog, fit, B, (int64_t)CH * CH, tested_per_octave, sample_offset, ppair);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

namespace {

// The differential test of diff_mustache.py:567-568 needs, for a selected record of one sample, its pair p-value, its own
// winning DoG value and the OTHER sample's value at the same pixel (v = ones; v[nz] = vAll: the other sample's winning value
// if it found that pixel, else 0 on a tested pixel / 1 elsewhere -- the caller resolves "not found" with the nz mask).  One
// workgroup per selected record scans the partner block's found list (pixels are unique): a few dozen records per block
// against ~20 000, instead of sorting and downloading both found sets.
__global__ void __launch_bounds__(kThreads)
pair_gather_kernel(const mst_found *__restrict__ found, uint32_t found_cap, const uint32_t *__restrict__ found_count,
                   const double *__restrict__ ppair, int P, const uint32_t *__restrict__ sel_index,
                   const uint32_t *__restrict__ sel_pixel, const uint32_t *__restrict__ sel_count, uint32_t out_cap,
                   double *__restrict__ out_pair, double *__restrict__ out_value, double *__restrict__ out_other) {
    __shared__ double hit;
    const int fb = blockIdx.y, slot = blockIdx.x;
    const uint32_t sc = sel_count[fb];                  // MST_BH_RETRY: the selection of this block has not happened yet
    const uint32_t nsel = sc == MST_BH_RETRY ? 0u : (sc < out_cap ? sc : out_cap);
    if ((uint32_t)slot >= nsel) return;
    const size_t o = (size_t)fb * out_cap + slot;
    const uint32_t idx = sel_index[o], pixel = sel_pixel[o];
    if (threadIdx.x == 0) hit = __longlong_as_double(0x7FF8000000000000ll);     // NaN = the partner did not find this pixel
    __syncthreads();
    const int pb = fb < P ? fb + P : fb - P;
    const uint32_t np = found_count[pb] < found_cap ? found_count[pb] : found_cap;
    const mst_found *pf = found + (size_t)pb * found_cap;
    for (uint32_t i = threadIdx.x; i < np; i += kThreads)
        if (pf[i].pixel == pixel) hit = pf[i].value;
    __syncthreads();
    if (threadIdx.x == 0) {
        out_pair[o] = ppair[(size_t)fb * found_cap + idx];
        out_value[o] = found[(size_t)fb * found_cap + idx].value;
        out_other[o] = hit;
    }
}

// The same gather with ONE workgroup per block: the selected pixels go into a small open-addressing table in LDS (pixel ->
// slot), then the partner block's found list is walked once and every record looks its pixel up -- 12 workgroups reading
// their partner's list once instead of one workgroup per selected record reading it all (3 072 workgroups, 37 us on six block
// pairs: a third of what the whole tail of a two-sample call costs).  H = table size (a power of two >= 2 * out_cap).
constexpr int kGatherThreads = 1024;
__global__ void __launch_bounds__(kGatherThreads)
pair_gather_block_kernel(const mst_found *__restrict__ found, uint32_t found_cap, const uint32_t *__restrict__ found_count,
                         const double *__restrict__ ppair, int P, const uint32_t *__restrict__ sel_index,
                         const uint32_t *__restrict__ sel_pixel, const uint32_t *__restrict__ sel_count, uint32_t out_cap,
                         double *__restrict__ out_pair, double *__restrict__ out_value, double *__restrict__ out_other, uint32_t H,
                         uint32_t max_selected) {
    extern __shared__ uint32_t gather_tab[];
    uint32_t *keys = gather_tab, *slots = gather_tab + H;
    constexpr uint32_t kEmpty = 0xFFFFFFFFu;               // no pixel index: CH * CH - 1 < 2^32 - 1
    const int fb = blockIdx.x, tid = threadIdx.x;
    const uint32_t sc = sel_count[fb];                     // MST_BH_RETRY: the selection of this block has not happened yet
    const uint32_t lim = out_cap < max_selected ? out_cap : max_selected;      // (the table holds 2 * max_selected keys)
    const uint32_t nsel = sc == MST_BH_RETRY ? 0u : (sc < lim ? sc : lim);
    if (nsel == 0) return;
    for (uint32_t i = tid; i < H; i += kGatherThreads) keys[i] = kEmpty;
    __syncthreads();
    auto hash = [&](uint32_t px) { return (px * 2654435761u) & (H - 1); };
    for (uint32_t slot = tid; slot < nsel; slot += kGatherThreads) {
        const size_t o = (size_t)fb * out_cap + slot;
        const uint32_t idx = sel_index[o], pixel = sel_pixel[o];
        out_pair[o] = ppair[(size_t)fb * found_cap + idx];
        out_value[o] = found[(size_t)fb * found_cap + idx].value;
        out_other[o] = __longlong_as_double(0x7FF8000000000000ll);       // NaN = the partner did not find this pixel
        uint32_t h = hash(pixel);
        while (atomicCAS(&keys[h], kEmpty, pixel) != kEmpty) h = (h + 1) & (H - 1);   // pixels are unique inside a block
        slots[h] = slot;
    }
    __syncthreads();
    const int pb = fb < P ? fb + P : fb - P;
    const uint32_t np = found_count[pb] < found_cap ? found_count[pb] : found_cap;
    const mst_found *pf = found + (size_t)pb * found_cap;
    for (uint32_t i = tid; i < np; i += kGatherThreads) {
        const mst_found r = pf[i];
        for (uint32_t h = hash(r.pixel); keys[h] != kEmpty; h = (h + 1) & (H - 1))
            if (keys[h] == r.pixel) {
                out_other[(size_t)fb * out_cap + slots[h]] = r.value;
                break;
            }
    }
}

}  // namespace

extern "C" int mst_pair_gather(const mst_found *found, uint32_t found_cap, const uint32_t *found_count, const double *ppair,
                               int32_t P, const uint32_t *sel_index, const uint32_t *sel_pixel, const uint32_t *sel_count,
                               uint32_t out_cap, uint32_t max_selected, double *out_pair, double *out_value,
                               double *out_other, void *stream) {
    MST_RANGE("tail: mst_pair_gather");
    if (!found || !found_count || !ppair || !sel_index || !sel_pixel || !sel_count || !out_pair || !out_value ||
        !out_other || P <= 0 || 2 * P > 65535 || found_cap == 0 || out_cap == 0 || max_selected > out_cap)
        return mst::fail(MST_E_ARG, "mst_pair_gather: bad argument");
    if (max_selected == 0) return MST_OK;
    if (max_selected <= 4096) {
        // the table is sized by the largest selection, not by the capacity: in a pipelined run this kernel starts while the NEXT
        // group's fused kernel holds 2 x 77 KB of every CU's LDS, and a workgroup asking for more than the few KB left over
        // waits until a fused workgroup retires (a 64 KB request cost the two-sample genome run its overlap: 0.077 -> 0.094 s)
        uint32_t H = 64;
        while (H < 2 * max_selected) H <<= 1;
        pair_gather_block_kernel<<<2 * P, kGatherThreads, sizeof(uint32_t) * 2 * H, mst::as_stream(stream)>>>(
            found, found_cap, found_count, ppair, P, sel_index, sel_pixel, sel_count, out_cap, out_pair, out_value, out_other, H,
            max_selected);
        MST_LAUNCH_CHECK();
        return MST_OK;
    }
    pair_gather_kernel<<<dim3(max_selected, 2 * P), kThreads, 0, mst::as_stream(stream)>>>(
        found, found_cap, found_count, ppair, P, sel_index, sel_pixel, sel_count, out_cap, out_pair, out_value, out_other);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

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
