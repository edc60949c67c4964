// Fused sigma-stack kernel for gfx950 (MI355X):  reference mustache/mustache.py:714-772 in ONE launch.
//
// What the reference does per dense block (CH x CH float64), through SciPy:
//   for each octave o in {1.6, 3.2}:  G_k = gaussian_filter(c, sigma_k), k = 1..12   (:719, :725, :734, :751)
//                                     D_k = G_k - G_{k+1}, k = 1..11                 (:728, :738, :754)
//                                     M_k = maximum_filter(D_k, 3x3, zero padded)    (:740-743, :757-758)
//     for each tested level c = 2..10: on the nz pixels
//         loc/scale of |D_c| (expon.fit, :755), and the sieve (:760-768)
//         upd = D_c > best  &  D_c == M_c  &  (D_p == M_p | D_n == M_n)  &  D_c > M_p  &  D_c > M_n
// 24 full-image blurs, 22 DoG images, 22 max images, ~10 boolean gathers per level: ~600 B of HBM traffic per
// pixel if every level is materialised.
//
// What this kernel does instead: one workgroup owns a 62x62-pixel tile (64x64 with the 1-pixel ring the 3x3 max
// needs).  It loads the tile of c with a 14-pixel reflect halo into LDS ONCE (~12 B/pixel of HBM reads), then
// walks all 24 levels out of LDS:
//   V pass  (axis 0): each thread produces 8 vertically consecutive samples of one column from a register window
//                     of 8+2r taps (lanes run along columns -> conflict-free ds_read_b64), result -> LDS `vb`
//   H pass  (axis 1): each thread produces 8 horizontally consecutive samples of one row (lanes run along rows, odd
//                     LDS pitch -> conflict-free), result stays in registers
//   DoG, 3x3 zero-padded max through a small LDS tile `db`, the 5-term sieve and the running best/level per pixel
//   stay in registers across all levels and both octaves; per-level min / sum of |D_c| are reduced per workgroup.
// Only the found pixels (~0.5 % of the block) and 2 x 18 partial statistics per tile are written back.
// The kernel is therefore FP64-VALU bound (~1150 non-fusable flops per pixel for the blurs alone), not HBM bound.
//
// Bit-exactness: the tap order is SciPy's C correlate1d on a symmetric kernel,
//     t = x[c]*w0;  for j = r..1:  t += (x[c-j] + x[c+j]) * w[j]
// axis 0 first, float64 intermediate, mode='reflect'; compiled with -ffp-contract=off so no FMA is formed.
// The taps themselves come from the host (NumPy), see mustache_amd/levels.py.
#include <cmath>
#include <cstring>
#include "mst_common.h"

namespace {

struct DevLevels {
    int n_octaves;
    int levels_per_octave;
    int radius[MST_MAX_LEVELS];
    int pad_;
    double taps[MST_MAX_LEVELS][MST_MAX_RADIUS + 1];
};

template <int RG_, int RMAX_>
struct Tile {
    static constexpr int RG = RG_;                // region edge: interior + 1-pixel ring for the 3x3 max
    static constexpr int IT = RG_ - 2;            // interior (owned) pixels per edge
    static constexpr int RMAX = RMAX_;            // largest blur radius this instantiation supports
    static constexpr int K = 8;                   // samples per thread along the filter axis
    static constexpr int NT = RG * RG / K;        // threads per workgroup
    static constexpr int NW = NT / 64;            // waves per workgroup
    static constexpr int CT = RG + 2 * RMAX;      // edge of the c tile held in LDS
    static constexpr int CTP = CT | 1;            // odd pitches: conflict-free for lanes along rows or columns
    static constexpr int VP = (RG + 2 * RMAX) | 1;
    static constexpr int DP = RG | 1;
    static constexpr int CT_ELEMS = CT * CTP;
    static constexpr int VB_ELEMS = RG * VP;
    static constexpr int DB_ELEMS = RG * DP;
    static constexpr int ST_ELEMS = MST_MAX_TESTED * NW * 2;
    static constexpr size_t LDS_BYTES = sizeof(double) * (size_t)(CT_ELEMS + VB_ELEMS + DB_ELEMS + ST_ELEMS);
    static_assert(NT % 64 == 0, "whole waves");
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    const int p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

__device__ __forceinline__ double dmax(double a, double b) { return a > b ? a : b; }

// Axis-0 pass for radius R over the (RG rows) x (RG + 2R columns) strip the axis-1 pass will need.
template <class T, int R>
__device__ __forceinline__ void vpass(const double *__restrict__ ct, double *__restrict__ vb,
                                      const double *__restrict__ wg, int tid) {
    constexpr int K = T::K;
    constexpr int NC = T::RG + 2 * R;
    constexpr int NITEM = (T::RG / K) * NC;
    double w[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) w[j] = wg[j];
    for (int it = tid; it < NITEM; it += T::NT) {
        const int rgp = it / NC;
        const int col = it - rgp * NC;
        const int row0 = rgp * K;
        const double *p = ct + (row0 + T::RMAX - R) * T::CTP + (T::RMAX - R) + col;
        double win[K + 2 * R];
#pragma unroll
        for (int i = 0; i < K + 2 * R; ++i) win[i] = p[i * T::CTP];
        double *q = vb + row0 * T::VP + col;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double t = win[k + R] * w[0];
#pragma unroll
            for (int j = R; j >= 1; --j) t = t + (win[k + R - j] + win[k + R + j]) * w[j];
            q[k * T::VP] = t;
        }
    }
}

// Axis-1 pass: thread (row rr, column group cg) -> g[0..K) = G at region columns cg*K .. cg*K+K-1.
template <class T, int R>
__device__ __forceinline__ void hpass(const double *__restrict__ vb, const double *__restrict__ wg, int rr, int cg,
                                      double (&g)[T::K]) {
    constexpr int K = T::K;
    double w[R + 1];
#pragma unroll
    for (int j = 0; j <= R; ++j) w[j] = wg[j];
    const double *p = vb + rr * T::VP + cg * K;
    double win[K + 2 * R];
#pragma unroll
    for (int i = 0; i < K + 2 * R; ++i) win[i] = p[i];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double t = win[k + R] * w[0];
#pragma unroll
        for (int j = R; j >= 1; --j) t = t + (win[k + R - j] + win[k + R + j]) * w[j];
        g[k] = t;
    }
}

template <class T, int R>
__device__ __forceinline__ void blur_level(const double *ct, double *vb, const double *wg, int tid, int rr, int cg,
                                           double (&g)[T::K]) {
    vpass<T, R>(ct, vb, wg, tid);
    __syncthreads();
    hpass<T, R>(vb, wg, rr, cg, g);
}

template <class T>
__device__ __forceinline__ void blur_dispatch(int r, const double *ct, double *vb, const double *wg, int tid, int rr,
                                              int cg, double (&g)[T::K]) {
#define MST_CASE(R_)                                                  \
    case R_:                                                          \
        if constexpr (R_ <= T::RMAX) blur_level<T, R_>(ct, vb, wg, tid, rr, cg, g); \
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

template <class T>
__global__ void __launch_bounds__(T::NT)
scale_space_kernel(const double *__restrict__ c, const uint8_t *__restrict__ nz, int CH,
                   const DevLevels *__restrict__ lv, mst_found *__restrict__ found, uint32_t found_cap,
                   uint32_t *__restrict__ found_count, double *__restrict__ partial, int tiles_per_dim,
                   int n_tested, int skip_empty) {
    constexpr int K = T::K, RG = T::RG, IT = T::IT, RMAX = T::RMAX;
    extern __shared__ __align__(16) double lds[];
    double *ct = lds;
    double *vb = ct + T::CT_ELEMS;
    double *db = vb + T::VB_ELEMS;
    double *st = db + T::DB_ELEMS;

    const int tid = threadIdx.x;
    const int b = blockIdx.y;
    const int ntiles = tiles_per_dim * tiles_per_dim;
    // XCD-aware order: hardware places workgroup i on XCD i % 8 (gridDim.x is a multiple of 8), so give each XCD a
    // contiguous run of tiles -- neighbouring tiles share their halo rows through that XCD's L2.
    const int per_xcd = gridDim.x >> 3;
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int ty = tile / tiles_per_dim, tx = tile - ty * tiles_per_dim;
    const int y0 = ty * IT - 1, x0 = tx * IT - 1;  // block coordinates of region (0, 0)

    const int rr = tid % RG;  // region row owned in the H pass (lane index when RG == 64)
    const int cg = tid / RG;  // column group
    const int gy = y0 + rr;
    const bool row_in = gy >= 0 && gy < CH;
    const bool row_own = row_in && rr >= 1 && rr <= IT;
    const double *cb = c + (size_t)b * CH * CH;
    const uint8_t *nb = nz + (size_t)b * CH * CH;
    double *part = partial + ((size_t)b * ntiles + tile) * n_tested * 2;

    uint32_t in_mask = 0, nz_mask = 0;  // per-k bits: column inside the block / tested pixel owned by this thread
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int rc = cg * K + k;
        const int gx = x0 + rc;
        const bool col_in = gx >= 0 && gx < CH;
        if (row_in && col_in) in_mask |= 1u << k;
        if (row_own && col_in && rc >= 1 && rc <= IT && nb[(size_t)gy * CH + gx]) nz_mask |= 1u << k;
    }
    const int any_nz = __syncthreads_or(nz_mask != 0);
    if (!any_nz && skip_empty) {
        for (int t = tid; t < n_tested; t += T::NT) {
            part[2 * t] = INFINITY;
            part[2 * t + 1] = 0.0;
        }
        return;
    }

    // ---- stage the c tile (reflect halo) in LDS: the only bulk HBM read of the kernel
    for (int idx = tid; idx < T::CT * T::CT; idx += T::NT) {
        const int i = idx / T::CT, j = idx - i * T::CT;
        const int sy = reflect_idx(y0 - RMAX + i, CH);
        const int sx = reflect_idx(x0 - RMAX + j, CH);
        ct[i * T::CTP + j] = cb[(size_t)sy * CH + sx];
    }
    __syncthreads();

    double gprev[K], Mp[K], Dc[K], Mc[K], best[K];
    uint32_t lvl[K];
    uint32_t ep = 0, ec = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        gprev[k] = Mp[k] = Dc[k] = Mc[k] = best[k] = 0.0;
        lvl[k] = 0;
    }
    const int rr_m = rr > 0 ? rr - 1 : 0, rr_p = rr < RG - 1 ? rr + 1 : RG - 1;
    const int wave = tid >> 6, lane = tid & 63;

    const int n_oct = lv->n_octaves, lpo = lv->levels_per_octave;
    int tested = 0;
    for (int o = 0; o < n_oct; ++o) {
        for (int kl = 1; kl <= lpo; ++kl) {
            const int l = o * lpo + kl - 1;
            const int r = lv->radius[l];
            double g[K];
            blur_dispatch<T>(r, ct, vb, lv->taps[l], tid, rr, cg, g);
            double d[K];
            if (kl >= 2) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    d[k] = gprev[k] - g[k];
                    if (!((in_mask >> k) & 1u)) d[k] = 0.0;  // maximum_filter pads with zeros outside the block
                    db[rr * T::DP + cg * K + k] = d[k];
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) gprev[k] = g[k];
            __syncthreads();
            if (kl < 2) continue;

            // zero-padded 3x3 max at the owned pixels: column maxima over 3 rows, then over 3 columns
            double cm[K + 2];
#pragma unroll
            for (int j = 0; j < K + 2; ++j) {
                int col = cg * K + j - 1;
                col = col < 0 ? 0 : (col > RG - 1 ? RG - 1 : col);
                const double a = db[rr_m * T::DP + col];
                const double bb = db[rr * T::DP + col];
                const double cc = db[rr_p * T::DP + col];
                cm[j] = dmax(dmax(a, bb), cc);
            }
            uint32_t en = 0;
            double m[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                m[k] = dmax(dmax(cm[k], cm[k + 1]), cm[k + 2]);
                if (d[k] == m[k]) en |= 1u << k;
            }
            if (kl >= 4) {
                // tested level = D_{kl-2}: previous = D_{kl-3} (ep, Mp), current (Dc, Mc, ec), next = this one (d, m, en)
                double lmin = INFINITY, lsum = 0.0;
                const uint32_t code = (uint32_t)tested + 1u;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool tz = (nz_mask >> k) & 1u;
                    const bool upd = tz && (Dc[k] > best[k]) && ((ec >> k) & 1u) && (((ep | en) >> k) & 1u) &&
                                     (Dc[k] > Mp[k]) && (Dc[k] > m[k]);
                    if (upd) {
                        best[k] = Dc[k];
                        lvl[k] = code;
                    }
                    if (tz) {
                        const double a = fabs(Dc[k]);
                        lmin = a < lmin ? a : lmin;
                        lsum = lsum + a;
                    }
                }
                // fixed-order butterfly inside the wave, then one slot per (level, wave): deterministic
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double omin = __shfl_xor(lmin, off, 64);
                    const double osum = __shfl_xor(lsum, off, 64);
                    lmin = omin < lmin ? omin : lmin;
                    lsum = lsum + osum;
                }
                if (lane == 0) {
                    st[(tested * T::NW + wave) * 2] = lmin;
                    st[(tested * T::NW + wave) * 2 + 1] = lsum;
                }
                ++tested;
            }
            ep = ec;
            ec = en;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Mp[k] = Mc[k];
                Mc[k] = m[k];
                Dc[k] = d[k];
            }
        }
    }

    // ---- found pixels -> per-block record list (unordered; the host sorts by pixel index)
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (lvl[k]) {
            const uint32_t pos = atomicAdd(found_count + b, 1u);
            if (pos < found_cap) {
                mst_found rec;
                rec.pixel = (uint32_t)gy * (uint32_t)CH + (uint32_t)(x0 + cg * K + k);
                rec.level = lvl[k];
                rec.value = best[k];
                found[(size_t)b * found_cap + pos] = rec;
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < tested; t += T::NT) {
        double mn = st[(t * T::NW) * 2], sm = st[(t * T::NW) * 2 + 1];
        for (int w = 1; w < T::NW; ++w) {
            const double a = st[(t * T::NW + w) * 2];
            mn = a < mn ? a : mn;
            sm = sm + st[(t * T::NW + w) * 2 + 1];
        }
        part[2 * t] = mn;
        part[2 * t + 1] = sm;
    }
}

// partial[b][tile][t][2] -> level_stats[b][t][2], fixed summation order (tile-major, then a fixed tree)
__global__ void __launch_bounds__(256)
stats_reduce_kernel(const double *__restrict__ partial, int ntiles, int n_tested, double *__restrict__ level_stats) {
    __shared__ double smin[256], ssum[256];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const double *p = partial + (size_t)b * ntiles * n_tested * 2;
    double mn = INFINITY, sm = 0.0;
    for (int i = tid; i < ntiles; i += 256) {
        const double a = p[((size_t)i * n_tested + t) * 2];
        mn = a < mn ? a : mn;
        sm = sm + p[((size_t)i * n_tested + t) * 2 + 1];
    }
    smin[tid] = mn;
    ssum[tid] = sm;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            smin[tid] = smin[tid + s] < smin[tid] ? smin[tid + s] : smin[tid];
            ssum[tid] = ssum[tid] + ssum[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        level_stats[((size_t)b * MST_MAX_TESTED + t) * 2] = smin[0];
        level_stats[((size_t)b * MST_MAX_TESTED + t) * 2 + 1] = ssum[0];
    }
}

int check_levels(const mst_levels *lv, int *max_radius, int *n_tested) {
    if (!lv) return mst::fail(MST_E_ARG, "level table is null");
    if (lv->n_octaves < 1 || lv->levels_per_octave < 4 ||
        lv->n_octaves * lv->levels_per_octave > MST_MAX_LEVELS)
        return mst::fail(MST_E_ARG, "level table: need >= 1 octave, >= 4 levels per octave, <= %d levels",
                         MST_MAX_LEVELS);
    const int nt = lv->n_octaves * (lv->levels_per_octave - 3);
    if (nt > MST_MAX_TESTED) return mst::fail(MST_E_ARG, "level table: more than %d tested levels", MST_MAX_TESTED);
    int mr = 0;
    for (int l = 0; l < lv->n_octaves * lv->levels_per_octave; ++l) {
        if (lv->radius[l] < 1 || lv->radius[l] > 28)
            return mst::fail(MST_E_ARG, "level %d: blur radius %d outside the supported range [1, 28]", l,
                             lv->radius[l]);
        mr = lv->radius[l] > mr ? lv->radius[l] : mr;
    }
    *max_radius = mr;
    *n_tested = nt;
    return MST_OK;
}

using TileDefault = Tile<64, 14>;   // the reference's default octaves (radius <= 14): 155 KB LDS, 512 threads
using TileWide = Tile<32, 28>;      // -sz / -oc variants up to radius 28: smaller tile, same code

template <class T>
int tiles_per_dim(int CH) { return (CH + T::IT - 1) / T::IT; }

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" uint64_t mst_scale_space_workspace_bytes(int32_t B, int32_t CH, const mst_levels *lv) {
    int mr = 0, nt = 0;
    if (B <= 0 || CH <= 0 || check_levels(lv, &mr, &nt) != MST_OK) return 0;
    const int tpd = mr <= TileDefault::RMAX ? tiles_per_dim<TileDefault>(CH) : tiles_per_dim<TileWide>(CH);
    return align_up(sizeof(DevLevels), 256) + sizeof(double) * 2 * (size_t)B * tpd * tpd * nt;
}

template <class T>
static int launch_scale_space(const double *c, const uint8_t *nz, int B, int CH, const DevLevels *d_lv,
                              mst_found *found, uint32_t found_cap, uint32_t *found_count, double *partial,
                              int n_tested, int skip_empty, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        MST_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&scale_space_kernel<T>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)T::LDS_BYTES));
        attr_set = true;
    }
    const int tpd = tiles_per_dim<T>(CH);
    const int ntiles = tpd * tpd;
    const int gx = (ntiles + 7) / 8 * 8;
    scale_space_kernel<T><<<dim3(gx, B), T::NT, T::LDS_BYTES, s>>>(c, nz, CH, d_lv, found, found_cap, found_count,
                                                                 partial, tpd, n_tested, skip_empty);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_scale_space(const double *c, const uint8_t *nz, int32_t B, int32_t CH, const mst_levels *lv,
                               mst_found *found, uint32_t found_cap, uint32_t *found_count, double *level_stats,
                               int32_t skip_empty, void *workspace, uint64_t workspace_bytes, void *stream) {
    int mr = 0, nt = 0;
    int rc = check_levels(lv, &mr, &nt);
    if (rc != MST_OK) return rc;
    if (!c || !nz || !found || !found_count || !level_stats || !workspace || B <= 0 || B > 65535 || CH <= 0 ||
        (int64_t)CH * CH > 0xFFFFFFFFLL)
        return mst::fail(MST_E_ARG, "mst_scale_space: bad argument");
    const uint64_t need = mst_scale_space_workspace_bytes(B, CH, lv);
    if (workspace_bytes < need)
        return mst::fail(MST_E_ARG, "mst_scale_space: workspace too small (%llu < %llu bytes)",
                         (unsigned long long)workspace_bytes, (unsigned long long)need);
    hipStream_t s = mst::as_stream(stream);

    // level table -> device (pageable source: the runtime stages it before returning, so `h` may die)
    DevLevels h;
    memset(&h, 0, sizeof(h));
    h.n_octaves = lv->n_octaves;
    h.levels_per_octave = lv->levels_per_octave;
    for (int l = 0; l < lv->n_octaves * lv->levels_per_octave; ++l) {
        h.radius[l] = lv->radius[l];
        for (int j = 0; j <= lv->radius[l]; ++j) h.taps[l][j] = lv->taps[l][j];
    }
    DevLevels *d_lv = reinterpret_cast<DevLevels *>(workspace);
    double *partial = reinterpret_cast<double *>(reinterpret_cast<char *>(workspace) + align_up(sizeof(DevLevels), 256));
    MST_HIP(hipMemcpyAsync(d_lv, &h, sizeof(h), hipMemcpyHostToDevice, s));
    MST_HIP(hipMemsetAsync(found_count, 0, sizeof(uint32_t) * B, s));

    int ntiles;
    if (mr <= TileDefault::RMAX) {
        rc = launch_scale_space<TileDefault>(c, nz, B, CH, d_lv, found, found_cap, found_count, partial, nt,
                                             skip_empty, s);
        ntiles = tiles_per_dim<TileDefault>(CH) * tiles_per_dim<TileDefault>(CH);
    } else {
        rc = launch_scale_space<TileWide>(c, nz, B, CH, d_lv, found, found_cap, found_count, partial, nt,
                                          skip_empty, s);
        ntiles = tiles_per_dim<TileWide>(CH) * tiles_per_dim<TileWide>(CH);
    }
    if (rc != MST_OK) return rc;
    stats_reduce_kernel<<<dim3(nt, B), 256, 0, s>>>(partial, ntiles, nt, level_stats);
    MST_LAUNCH_CHECK();
    return MST_OK;
}
