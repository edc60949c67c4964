// Diagonal-major band layout and the diagonal-distance normalisation (gfx950).
//
// A chromosome's near-diagonal contacts live in HBM as   band[d * n + i] = value of pixel (i, i + d),
// d = 0 .. dpx+1, i = 0 .. n-1, 0.0 = no contact.  Every diagonal is a contiguous row, so the per-diagonal
// statistics and sliding windows of the reference's normalize_sparse (mustache/mustache.py:622-686) become
// coalesced streams, and cutting dense blocks out of it (mustache.py:919-924 + the prologue :699-706) is a
// tile transpose through LDS instead of a host-side boolean mask per block.
//
//   mst_band_from_coo     COO (x, y, v) -> band            (what `vals[x[indices]] = v[indices]` does per diagonal)
//   mst_normalize_band    mustache.py:628-685, both branches
//   mst_band_to_coo       gather the normalised values back into COO order (drop-in `v` in place semantics)
//   mst_blocks_from_band  mustache.py:919-924 + :699-706 fused: filled dense blocks + nz mask straight from the band
//
// Reproducibility note: the reference's window sums come from np.convolve -> BLAS ddot, whose
// accumulation order depends on the host's BLAS build, so bit-equality with "the" reference does not exist for this stage.
// Every window sum here is the sum of two partial sums of at most W terms each (the walking kernel below), accumulated in a
// fixed order: deterministic, and within 1.2e-12 (relative, z-scores) of the same formula evaluated in extended precision --
// the distance the reference's own float64 path has from it (scripts/norm_accuracy.py).  Tests hold 1e-11 against the oracle
// and require identical loop sets downstream.
#include <cmath>
#include <cstdlib>
#include "mst_common.h"

namespace {

constexpr int kThreads = 256;

__global__ void __launch_bounds__(kThreads)
band_scatter_kernel(const int64_t *__restrict__ x, const int64_t *__restrict__ y, const double *__restrict__ v,
                    int64_t nnz, int64_t n, int dpx, double *__restrict__ band) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
        const int64_t a = x[e], b = y[e];
        const int64_t lo = a < b ? a : b, d = a < b ? b - a : a - b;
        if (d <= dpx + 1 && lo >= 0 && lo + d < n) band[d * n + lo] = v[e];
    }
}

// the `.hic` reader's packed records (include/mustache_io.h, mst_hic_read_intra_packed): x = binX, dist = binY - binX >= 0,
// value = the float32 straw computes; 12 bytes per record
__global__ void __launch_bounds__(kThreads)
band_scatter_packed_kernel(const int32_t *__restrict__ x, const int32_t *__restrict__ dist, const float *__restrict__ v,
                           int64_t nnz, int64_t n, int dpx, double *__restrict__ band) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
        const int64_t lo = x[e], d = dist[e];
        if (d >= 0 && d <= dpx + 1 && lo >= 0 && lo + d < n) band[d * n + lo] = (double)v[e];
    }
}

// the same with the distance as uint16 (10 bytes per record on PCIe; the distance limit is at most 65 533 bins, as
// mst_normalize_band requires) -- slabs of a streaming read are scattered one by one, in any order
__global__ void __launch_bounds__(kThreads)
band_scatter_packed16_kernel(const int32_t *__restrict__ x, const uint16_t *__restrict__ dist, const float *__restrict__ v,
                             int64_t nnz, int64_t n, int dpx, double *__restrict__ band) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
        const int64_t lo = x[e], d = dist[e];
        if (d <= dpx + 1 && lo >= 0 && lo + d < n) band[d * n + lo] = (double)v[e];
    }
}

// read-back check of a packed scatter: counts the records whose pixel does not hold their value -- a pixel that two records
// with different values were written to shows up for one of them whichever store won the race (malformed input: a `.hic`
// matrix holds every pixel once)
template <class D>
__global__ void __launch_bounds__(kThreads)
band_verify_packed_kernel(const int32_t *__restrict__ x, const D *__restrict__ dist, const float *__restrict__ v, int64_t nnz,
                          int64_t n, int dpx, const double *__restrict__ band, unsigned long long *__restrict__ mismatches) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
        const int64_t lo = x[e], d = (int64_t)dist[e];
        if (d >= 0 && d <= dpx + 1 && lo >= 0 && lo + d < n && band[d * n + lo] != (double)v[e]) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

__global__ void __launch_bounds__(kThreads)
band_gather_kernel(const double *__restrict__ band, const int64_t *__restrict__ x, const int64_t *__restrict__ y,
                   int64_t nnz, int64_t n, int dpx, double *__restrict__ v) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
        const int64_t a = x[e], b = y[e];
        const int64_t lo = a < b ? a : b, d = a < b ? b - a : a - b;
        if (d <= dpx + 1 && lo >= 0 && lo + d < n) v[e] = band[d * n + lo];
    }
}

// fixed-order block reduction of (a, b): strided per-thread partials, then a binary tree in LDS
__device__ __forceinline__ void block_reduce2(double &a, double &b, double *sa, double *sb) {
    const int tid = threadIdx.x;
    sa[tid] = a;
    sb[tid] = b;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (tid < s) {
            sa[tid] = sa[tid] + sa[tid + s];
            sb[tid] = sb[tid] + sb[tid + s];
        }
        __syncthreads();
    }
    a = sa[0];
    b = sb[0];
    __syncthreads();
}

// One workgroup per diagonal: count, mean and population std of the raw entries (np.mean / np.std at
// mustache.py:638-643, :677-682; NaN -> mean 0, std 1), and the weight 1 + log30(1 + mean) (:667).
// diag_stats[d] = {mean, std, weight, count}
// The row is read ONCE: np.std's two passes (mean, then squared deviations) become shifted sums around a pivot K = the
// mean of the non-zero entries among 256 samples spread evenly over the row:
//     mean = K + sum(v - K) / n,      var = (sum((v - K)^2) - sum(v - K)^2 / n) / n.
// With K within a few standard deviations of the mean the cancellation in `var` costs a few ulp (relative error
// ~ eps * (1 + (mean - K)^2 / var)); the band is 4 GB for chr1 at 1 kb, so the second pass was as expensive as the first.
__global__ void __launch_bounds__(kThreads)
diag_stats_kernel(const double *__restrict__ band, int64_t n, double *__restrict__ diag_stats) {
    __shared__ double sa[kThreads], sb[kThreads];
    const int d = blockIdx.x;
    const int64_t L = n - d;
    const double *row = band + (int64_t)d * n;
    double hc = 0.0, hs = 0.0;
    if (L > 0) {
        // 256 samples spread evenly over the whole diagonal, not its head: acrocentric chromosomes and telomeric gaps begin
        // with megabases of empty bins, and a pivot of 0 would turn the formula into the naive sum(x^2) - sum(x)^2 / n
        const double v = row[(int64_t)threadIdx.x * L / kThreads];
        if (v != 0.0 && isfinite(v)) {
            hc = 1.0;
            hs = v;
        }
    }
    block_reduce2(hc, hs, sa, sb);
    const double K = hc > 0.0 ? hs / hc : 0.0;
    double cnt = 0.0, s1 = 0.0, s2 = 0.0, dummy = 0.0;
    {
        // four independent accumulator sets per thread (fixed assignment: sample i goes to set (i / kThreads) % 4), so four
        // loads are in flight per lane; folded in a fixed order
        constexpr int U = 4;
        double c[U] = {0.0, 0.0, 0.0, 0.0}, a[U] = {0.0, 0.0, 0.0, 0.0}, q[U] = {0.0, 0.0, 0.0, 0.0};
        int64_t i = threadIdx.x;
        for (; i + (U - 1) * kThreads < L; i += U * kThreads) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = row[i + u * kThreads];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (v[u] != 0.0) {
                    const double t = v[u] - K;
                    c[u] = c[u] + 1.0;
                    a[u] = a[u] + t;
                    q[u] = q[u] + t * t;
                }
        }
        for (int u = 0; i < L; i += kThreads, ++u) {
            const double v = row[i];
            if (v != 0.0) {
                const double t = v - K;
                c[u] = c[u] + 1.0;
                a[u] = a[u] + t;
                q[u] = q[u] + t * t;
            }
        }
        cnt = (c[0] + c[1]) + (c[2] + c[3]);
        s1 = (a[0] + a[1]) + (a[2] + a[3]);
        s2 = (q[0] + q[1]) + (q[2] + q[3]);
    }
    block_reduce2(cnt, s1, sa, sb);
    block_reduce2(s2, dummy, sa, sb);
    double mean = K + s1 / cnt;         // 0/0 -> NaN like np.mean of an empty selection
    double var = (s2 - s1 * s1 / cnt) / cnt;
    if (var < 0.0) var = 0.0;           // rounding of an (almost) constant diagonal; NaN (empty diagonal) passes through
    double sd = sqrt(var);
    if (mean != mean) mean = 0.0;       // math.isnan(mean) -> 0   (:640-641)
    if (sd != sd) sd = 1.0;             // math.isnan(std)  -> 1   (:642-643)
    if (threadIdx.x == 0) {
        diag_stats[4 * d + 0] = mean;
        diag_stats[4 * d + 1] = sd;
        diag_stats[4 * d + 2] = 1.0 + log(1.0 + mean) / log(30.0);
        diag_stats[4 * d + 3] = cnt;
    }
}

// Branch A (mustache.py:632-669): sliding-window z-score.  One workgroup = SEG consecutive positions of one
// diagonal; the SEG + W samples it needs (+0.001 shift applied, and their squares) are staged in LDS once, together
// with the sums of every 16-sample block aligned to absolute position (i mod 16 == 0).  A window is then summed as
// head (<= 15 samples) + whole aligned blocks + tail (<= 15 samples): ~150 additions instead of W = 2000, in an order
// that depends only on the window's absolute position (deterministic, independent of how the diagonal is segmented),
// and with the rounding behaviour of a two-level (blocked) summation.
// Two instantiations: <16, true, false> (windows up to ~8400 bins: samples AND their squares staged, 16-sample blocks -- the form
// every result so far was produced with) and <32, false, true> (windows up to 16384 bins, i.e. resolutions down to ~125 bp:
// samples only -- the square is formed where it is summed, the same product -- 32-sample blocks, so that 1024 + 16384 samples
// and their block sums fit the 160 KB of LDS, and the block sums turned into exclusive PREFIX sums over the tile's <= 545
// blocks, so that the whole blocks of a window cost one subtraction instead of up to 512 additions: head + tail <= 62 samples
// per output).  Both stay far below 128 VGPRs: no scratch (the walking kernel's <1024, 16> form, which served these windows
// until round 4, spilled 488 registers and was 1.6 x faster all the same: 9009-bin windows on 1.2e8 samples 11.8 ms there, 19.0 ms
// here -- the price of a kernel without scratch on a path for resolutions below 238 bp; LABBOOK R5.6).
constexpr int kSeg = 1024;
constexpr int kBlk = 16;
constexpr int kBlkWide = 32;

template <int BLK, bool STAGE_SQ, bool PREFIX>
__global__ void __launch_bounds__(kThreads)
normalize_local_kernel(const double *__restrict__ band_in, double *__restrict__ band_out, int64_t n, int W,
                       const double *__restrict__ diag_stats) {
    extern __shared__ __align__(16) double lds[];
    const int d = blockIdx.y;
    const int64_t L = n - d;
    const int64_t seg0 = (int64_t)blockIdx.x * kSeg;
    if (seg0 >= L) return;
    const int left = W / 2;                 // np.convolve(..., 'same'): window = [i - W/2, i - W/2 + W - 1]
    // tile element t <-> absolute position base + t; base is rounded DOWN to a multiple of BLK so LDS blocks are the
    // absolute aligned blocks
    const int64_t first = seg0 - left;
    const int64_t base = (first >= 0 ? first : first - (BLK - 1)) / BLK * BLK;      // floor to multiple of BLK
    const int tile = (int)(seg0 + kSeg - 1 - left + W - base) + 1;                     // covers the last window's end
    const int nblk = (tile + BLK - 1) / BLK;
    const int nb1 = nblk + (PREFIX ? 1 : 0);
    double *val = lds, *sqa = val + nblk * BLK, *b1 = sqa + (STAGE_SQ ? nblk * BLK : 0), *b2 = b1 + nb1;
    int *bc = reinterpret_cast<int *>(b2 + nb1);
    auto sq = [&](int t) { return STAGE_SQ ? sqa[t] : val[t] * val[t]; };       // vals ** 2  (:649)
    const double *row = band_in + (int64_t)d * n;
    for (int t = threadIdx.x; t < nblk * BLK; t += kThreads) {
        const int64_t i = base + t;
        double v = 0.0;
        if (i >= 0 && i < L) {
            const double r = row[i];
            if (r != 0.0) v = r + 0.001;    // vals[x] = v + 0.001   (:635)
        }
        val[t] = v;
        if (STAGE_SQ) sqa[t] = v * v;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nblk; q += kThreads) {
        double s1 = 0.0, s2 = 0.0;
        int c = 0;
#pragma unroll
        for (int u = 0; u < BLK; ++u) {
            const double a = val[q * BLK + u];
            c += (a != 0.0) ? 1 : 0;
            s1 = s1 + a;
            s2 = s2 + sq(q * BLK + u);
        }
        b1[q] = s1;
        b2[q] = s2;
        bc[q] = c;
    }
    __syncthreads();
    if (PREFIX) {
        // b1 / b2 / bc[q] -> sums of the blocks BEFORE q (nblk + 1 entries: the arrays are one longer in this form); three threads,
        // one array each, a fixed serial order
        if (threadIdx.x == 0) {
            double run = 0.0;
            for (int q = 0; q <= nblk; ++q) {
                const double v = q < nblk ? b1[q] : 0.0;
                b1[q] = run;
                run = run + v;
            }
        } else if (threadIdx.x == 64) {
            double run = 0.0;
            for (int q = 0; q <= nblk; ++q) {
                const double v = q < nblk ? b2[q] : 0.0;
                b2[q] = run;
                run = run + v;
            }
        } else if (threadIdx.x == 128) {
            int run = 0;
            for (int q = 0; q <= nblk; ++q) {
                const int v = q < nblk ? bc[q] : 0;
                bc[q] = run;
                run += v;
            }
        }
        __syncthreads();
    }
    const double mean = diag_stats[4 * d + 0], sd = diag_stats[4 * d + 1], wgt = diag_stats[4 * d + 2];
    const double std2 = sd * sd;
    double *orow = band_out + (int64_t)d * n;
    for (int k = threadIdx.x; k < kSeg; k += kThreads) {
        const int64_t i = seg0 + k;
        if (i >= L) break;
        const int ta = (int)(i - left - base);          // window = tile elements [ta, ta + W - 1]
        const double x = val[ta + left];
        double z = 0.0;
        if (x != 0.0) {
            const int tb = ta + W;                       // exclusive end
            const int A = (ta + BLK - 1) / BLK * BLK; // first aligned block start >= ta
            const int B = tb / BLK * BLK;              // last aligned block end <= tb
            double s1 = 0.0, s2 = 0.0;
            int c = 0;
            if (A <= B) {
                double h1 = 0.0, h2 = 0.0;
                for (int t = ta; t < A; ++t) {           // head
                    const double a = val[t];
                    c += (a != 0.0) ? 1 : 0;
                    h1 = h1 + a;
                    h2 = h2 + sq(t);
                }
                s1 = h1;
                s2 = h2;
                if (PREFIX) {                              // whole aligned blocks: one difference of prefix sums
                    c += bc[B / BLK] - bc[A / BLK];
                    s1 = s1 + (b1[B / BLK] - b1[A / BLK]);
                    s2 = s2 + (b2[B / BLK] - b2[A / BLK]);
                } else {
                    for (int q = A / BLK; q < B / BLK; ++q) {   // whole aligned blocks
                        c += bc[q];
                        s1 = s1 + b1[q];
                        s2 = s2 + b2[q];
                    }
                }
                double t1 = 0.0, t2 = 0.0;
                for (int t = B; t < tb; ++t) {           // tail
                    const double a = val[t];
                    c += (a != 0.0) ? 1 : 0;
                    t1 = t1 + a;
                    t2 = t2 + sq(t);
                }
                s1 = s1 + t1;
                s2 = s2 + t2;
            } else {                                      // window shorter than one aligned block
                for (int t = ta; t < tb; ++t) {
                    const double a = val[t];
                    c += (a != 0.0) ? 1 : 0;
                    s1 = s1 + a;
                    s2 = s2 + sq(t);
                }
            }
            const double cnt = (double)c;
            double var = (s2 - s1 * s1 / cnt) / (cnt - 1.0);        // (:650)
            if (!isfinite(var)) var = std2;                          // (:653-654)
            double mu = s1 / cnt;                                    // (:656)
            if (c < 30) {                                            // (:657-658)
                mu = mean;
                var = std2;
            }
            if (!isfinite(mu)) mu = mean;                            // (:660-661)
            z = (x - mu) / sqrt(var);                                // (:663-665)
            if (!isfinite(z)) z = 0.0;                               // (:666)
            z = z * wgt;                                             // (:667)
        }
        orow[i] = z;
    }
}

// Branch A, prefix-sum form (round 1's kernel; compiled in PROFILE builds only): one workgroup = kPSeg consecutive positions of one diagonal.  The kPSeg + W
// samples it needs are read ONCE and turned into three exclusive prefix arrays in LDS -- count of non-zero samples
// (exact, int32), sum of (v + 0.001), sum of squares -- so every window is two LDS reads per quantity instead of
// ~W/16 block sums: the kernel becomes a streaming pass over the band (8 B read + 8 B written per sample).
// Numerics: a window sum is P[end] - P[begin] with both prefixes accumulated from the tile start (at most kPSeg + W
// terms), i.e. a relative error of a few 1e-16 on the window sums -- the same order as the difference between BLAS
// builds of the reference's np.convolve; parity with the reference fixtures is held at 1e-9 (tests).  The summation
// order is fixed by (diagonal, segment), so results are deterministic.
#ifdef MST_PROFILE
constexpr int kPSeg = 1024;
#endif

// ---- wave-level inclusive scan with DPP moves (no LDS crossbar): Hillis-Steele inside the four 16-lane rows (row_shr 1, 2,
// 4, 8; lanes without a source read 0), then row_bcast15 / row_bcast31 add the totals of the rows below.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int x) {
    return __builtin_amdgcn_update_dpp(0, x, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_d(double x) {
    return __hiloint2double(dpp_i<CTRL, ROW_MASK>(__double2hiint(x)), dpp_i<CTRL, ROW_MASK>(__double2loint(x)));
}
__device__ __forceinline__ void wave_scan3(double &a, double &b, int &c) {
#define MST_SCAN_STEP(CTRL, MASK)           \
    {                                       \
        const double ua = dpp_d<CTRL, MASK>(a); \
        const double ub = dpp_d<CTRL, MASK>(b); \
        const int uc = dpp_i<CTRL, MASK>(c);    \
        a = a + ua;                         \
        b = b + ub;                         \
        c += uc;                            \
    }
    MST_SCAN_STEP(0x111, 0xf)   // row_shr:1
    MST_SCAN_STEP(0x112, 0xf)   // row_shr:2
    MST_SCAN_STEP(0x114, 0xf)   // row_shr:4
    MST_SCAN_STEP(0x118, 0xf)   // row_shr:8
    MST_SCAN_STEP(0x142, 0xa)   // row_bcast:15 -> rows 1 and 3
    MST_SCAN_STEP(0x143, 0xc)   // row_bcast:31 -> rows 2 and 3
#undef MST_SCAN_STEP
}
// inclusive scan value of the previous lane (0 for lane 0): wave_shr:1
__device__ __forceinline__ double lane_before(double x) { return dpp_d<0x138, 0xf>(x); }
__device__ __forceinline__ int lane_before_i(int x) { return dpp_i<0x138, 0xf>(x); }

#ifdef MST_PROFILE
constexpr int kPThreads = 256;                         // (1024-thread workgroups were measured: 20 % slower)
constexpr int kPMaxChunk = 17;                         // samples per thread a tile may need (W <= 3072); always odd
#endif

#ifdef MST_PROFILE   /* the round-1 segment / prefix-array form: PROFILE builds only (cross-check, scripts/norm_*.py) */
// raw samples of one item's tile into registers: r[u] = element t = tid + u * kPThreads (0 outside the diagonal)
template <int CHUNK>
__device__ __forceinline__ void normalize_prefix_fetch(const double *__restrict__ band_in, int64_t n, int W,
                                                       int nseg, int64_t item, double (&r)[CHUNK]) {
    const int d = (int)(item / nseg);
    const int64_t L = n - d;
    const int64_t base = (item - (int64_t)d * nseg) * kPSeg - W / 2;
    const int tile = kPSeg + W - 1;
    const double *row = band_in + (int64_t)d * n;
#pragma unroll
    for (int u = 0; u < CHUNK; ++u) {
        const int t = (int)threadIdx.x + u * kPThreads;
        const int64_t i = base + t;
        r[u] = (t < tile && i >= 0 && i < L) ? row[i] : 0.0;
    }
}

template <int CHUNK>
__device__ __forceinline__ void normalize_prefix_item(const double *__restrict__ band_in, double *__restrict__ band_out,
                                                      int64_t n, int W, const double *__restrict__ diag_stats,
                                                      int nseg, int64_t item, int64_t next_item, double (&r)[CHUNK],
                                                      double *lds, double *w1, double *w2, int *wc) {
    const int d = (int)(item / nseg);
    const int64_t L = n - d;
    const int64_t seg0 = (item - (int64_t)d * nseg) * kPSeg;
    double *orow = band_out + (int64_t)d * n;
    if (seg0 >= L) {                                   // past the end of this diagonal: the output band is zero there
        for (int k = threadIdx.x; k < kPSeg; k += kPThreads)
            if (seg0 + k < n) orow[seg0 + k] = 0.0;
        if (next_item >= 0) normalize_prefix_fetch<CHUNK>(band_in, n, W, nseg, next_item, r);
        return;
    }
    const int left = W / 2;                            // np.convolve(..., 'same'): window = [i - W/2, i - W/2 + W - 1]
    // tile element t <-> absolute position seg0 - left + t; the last window ends at tile element kPSeg + W - 2
    constexpr int cap = kPThreads * CHUNK;                  // >= tile + 1 slots per array
    double *P1 = lds, *P2 = P1 + cap;                  // exclusive prefixes: P[t] = sum of elements [0, t)
    int *Pc = reinterpret_cast<int *>(P2 + cap);
    double *X = reinterpret_cast<double *>(Pc + cap + (cap & 1));          // the segment's own (shifted) samples
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the tile's samples were fetched into registers while the previous item was being processed; they go to P1's slots
    // (each is overwritten by its prefix below), then the NEXT item's fetch is issued so its HBM latency hides under this
    // item's scans and outputs
#pragma unroll
    for (int u = 0; u < CHUNK; ++u) {
        const int t = tid + u * kPThreads;
        P1[t] = r[u] != 0.0 ? r[u] + 0.001 : 0.0;                          // vals[x] = v + 0.001   (:635)
    }
    if (next_item >= 0) normalize_prefix_fetch<CHUNK>(band_in, n, W, nseg, next_item, r);
    __syncthreads();
    // thread-serial chunk (odd length: conflict-free LDS stride; read in one batch), then an inclusive scan of the thread
    // totals across the wave with DPP moves and across the four waves through LDS
    const int t0 = tid * CHUNK;
    double vv[CHUNK];
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) vv[j] = P1[t0 + j];
    double a1 = 0.0, a2 = 0.0;
    int ac = 0;
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
        ac += (vv[j] != 0.0) ? 1 : 0;
        a1 = a1 + vv[j];
        a2 = a2 + vv[j] * vv[j];                       // vals ** 2             (:649)
    }
    double s1 = a1, s2 = a2;
    int sc = ac;
    wave_scan3(s1, s2, sc);
    if (lane == 63) {
        w1[wave] = s1;
        w2[wave] = s2;
        wc[wave] = sc;
    }
    __syncthreads();
    double o1 = 0.0, o2 = 0.0;
    int oc = 0;
    for (int w = 0; w < wave; ++w) {
        o1 = o1 + w1[w];
        o2 = o2 + w2[w];
        oc += wc[w];
    }
    // exclusive prefix at the start of this thread's chunk = waves before + the lanes before (inclusive minus own); every
    // sample slot is replaced by the prefix in front of it
    double p1 = o1 + lane_before(s1), p2 = o2 + lane_before(s2);
    int pc = oc + lane_before_i(sc);
#pragma unroll
    for (int j = 0; j < CHUNK; ++j) {
        const int t = t0 + j;
        P1[t] = p1;
        P2[t] = p2;
        Pc[t] = pc;
        if (t >= left && t < left + kPSeg) X[t - left] = vv[j];
        pc += (vv[j] != 0.0) ? 1 : 0;
        p1 = p1 + vv[j];
        p2 = p2 + vv[j] * vv[j];
    }
    __syncthreads();
    const double mean = diag_stats[4 * d + 0], sd = diag_stats[4 * d + 1], wgt = diag_stats[4 * d + 2];
    const double std2 = sd * sd;
    for (int k = tid; k < kPSeg; k += kPThreads) {
        const int64_t i = seg0 + k;
        if (i >= n) break;
        double z = 0.0;
        if (i < L) {
            const double x = X[k];
            if (x != 0.0) {
                const int ta = k, tb = k + W;           // window = tile elements [k, k + W)
                const int c = Pc[tb] - Pc[ta];
                const double s1w = P1[tb] - P1[ta], s2w = P2[tb] - P2[ta];
                // 1/cnt and 1/(cnt-1) from ONE division (FP64 divisions dominate this kernel's instruction count); a
                // window with c < 30 -- including c = 1, where the product form gives NaN -- takes the fallback below
                const double cnt = (double)c;
                const double rcc = 1.0 / (cnt * (cnt - 1.0));
                const double inv_c = rcc * (cnt - 1.0), inv_cm1 = rcc * cnt;
                double var = (s2w - s1w * s1w * inv_c) * inv_cm1;        // (:650)
                if (!isfinite(var)) var = std2;                          // (:653-654)
                double mu = s1w * inv_c;                                 // (:656)
                if (c < 30) {                                            // (:657-658)
                    mu = mean;
                    var = std2;
                }
                if (!isfinite(mu)) mu = mean;                            // (:660-661)
                z = (x - mu) * rsqrt(var);                               // (:663-665)
                if (!isfinite(z)) z = 0.0;                               // (:666)
                z = z * wgt;                                             // (:667)
            }
        }
        orow[i] = z;
    }
}

// Work item = (diagonal, segment), numbered diagonal-major.  Every workgroup walks a contiguous run of items, so the W
// samples two neighbouring segments share are re-read from its own CU's / XCD's caches, and the grid is a few thousand
// workgroups instead of one per item (488 k items for chr1 @ 1 kb would be workgroup-dispatch bound at ~18 ns each).
template <int CHUNK>
__global__ void __launch_bounds__(kPThreads)
normalize_prefix_kernel(const double *__restrict__ band_in, double *__restrict__ band_out, int64_t n, int W,
                        const double *__restrict__ diag_stats, int nseg, int nd, int items_per_wg) {
    extern __shared__ __align__(16) double lds[];
    __shared__ double w1[kPThreads / 64], w2[kPThreads / 64];
    __shared__ int wc[kPThreads / 64];
    const int64_t total = (int64_t)nseg * nd;
    const int64_t first = (int64_t)blockIdx.x * items_per_wg;
    const int64_t last = first + items_per_wg < total ? first + items_per_wg : total;
    double r[CHUNK];
    if (first < last) normalize_prefix_fetch<CHUNK>(band_in, n, W, nseg, first, r);
    for (int64_t item = first; item < last; ++item) {
        normalize_prefix_item<CHUNK>(band_in, band_out, n, W, diag_stats, nseg, item, item + 1 < last ? item + 1 : -1, r,
                              lds, w1, w2, wc);
        __syncthreads();                               // the LDS arrays are reused by the next item
    }
}
#endif  // MST_PROFILE

// ---- Branch A, walking form (the default) ------------------------------------------------------------------------------
// A workgroup walks along ONE diagonal in blocks of W samples, W = the window length.  With sample blocks
//     B_m = [m W - left, (m + 1) W - left),   left = W / 2   (np.convolve(..., 'same'): window of output i = [i - left, i - left + W)),
// the window of output i = m W + k is the tail of B_m from offset k plus the head of B_{m+1} up to offset k, so
//     window sum(i) = (T_m - P_m[k]) + P_{m+1}[k]          P = exclusive prefix inside a block, T = block total,
// and the thread that holds offset k of the scans needs nothing from any other thread: ONE 3-quantity scan (count, sum, sum of
// squares) per sample, no prefix arrays in LDS, no re-scan of the W - 1 samples neighbouring segments share (the segment
// form above scans every sample ~3 times at W = 2000 and parks 20 B per sample in LDS).  A thread owns C consecutive
// offsets: it loads them straight from the band (the C loads of a wave cover the same cache lines), scans serially, and
// joins the other threads through one DPP wave scan + one LDS exchange of the wave totals -- ONE barrier per block.  The
// centre sample x_i of an output sits W / 2 further along, in another thread's chunk: it is simply read again (cache hit).
// Numerics: every window sum is the sum of two partial sums of at most W terms each, accumulated in a fixed order
// (deterministic; relative error of a few 1e-16 on the sums, measured against extended-precision windows by
// scripts/norm_accuracy.py).  Counts are exact integers.
// Reciprocal and reciprocal square root for the walking kernel: the hardware estimate (v_rcp_f64 / v_rsq_f64) refined by two
// Newton steps in FMA arithmetic -- within an ulp of the correctly rounded value, a third of the instructions of the IEEE
// division / the library rsqrt (no scaling, no special-case selects).  Zero, negative and non-finite arguments produce
// inf / NaN, which the callers' nan_to_num steps turn into the reference's fall-backs exactly as a true division would.
__device__ __forceinline__ double rcp_newton(double a) {
    double y = __builtin_amdgcn_rcp(a);
    double e = __builtin_fma(-a, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-a, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double rsqrt_newton(double a) {
    double y = __builtin_amdgcn_rsq(a);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const double e = __builtin_fma(-(a * y), y, 1.0);      // 1 - a y^2
        y = __builtin_fma(0.5 * y, e, y);
    }
    return y;
}

template <int NT>
__device__ __forceinline__ void block_totals3(double &s1, double &s2, int &sc, double *w1, double *w2, int *wc,
                                               double &o1, double &o2, int &oc, double &t1, double &t2, int &tc) {
    // in: this thread's chunk totals.  out: (s*) inclusive scan across the wave, (o*) sum over the waves before this one,
    // (t*) block totals.  Fixed order: DPP scan inside the wave, waves added in index order.
    constexpr int NW = NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    wave_scan3(s1, s2, sc);
    if (lane == 63) {
        w1[wave] = s1;
        w2[wave] = s2;
        wc[wave] = sc;
    }
    __syncthreads();
    o1 = o2 = t1 = t2 = 0.0;
    oc = tc = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const double a = w1[w], b = w2[w];
        const int c = wc[w];
        if (w < wave) {
            o1 = o1 + a;
            o2 = o2 + b;
            oc += c;
        }
        t1 = t1 + a;
        t2 = t2 + b;
        tc += c;
    }
}

template <int NT, int C>
#ifndef MST_WALK_MINW
#define MST_WALK_MINW 4
#endif
__global__ void __launch_bounds__(NT, (C <= 4 ? MST_WALK_MINW : 2))
normalize_walk_kernel(const double *__restrict__ band_in, double *__restrict__ band_out, int64_t n, int W,
                      const double *__restrict__ diag_stats, int nb, int run, int runs_per_diag) {
    constexpr int NW = NT / 64;
    __shared__ double w1[2][NW], w2[2][NW];            // wave totals of the block being scanned, double-buffered: ONE barrier per block
    __shared__ int wc[2][NW];
    // The outputs' own samples (output i = m W + k is sample offset k + left of block m: another thread's chunk) are exchanged
    // through LDS: every thread parks the shifted samples it scans in a ring of three blocks, and the barrier of the scan
    // makes them visible -- no second read of the band (it was an L2 hit, but 4 strided loads with bounds arithmetic per
    // thread and block).  Geometries whose ring would not fit the 64 KB static limit keep the re-read.
    constexpr bool XLDS = NT * C <= 2048;
    constexpr int CAPW = NT * C;
    __shared__ __align__(16) double ring[XLDS ? 3 * CAPW : 2];
    const int tid = threadIdx.x;
    const int d = blockIdx.x / runs_per_diag;
    const int m_lo = (blockIdx.x - d * runs_per_diag) * run;
    const int m_hi = m_lo + run < nb ? m_lo + run : nb;
    const int64_t L = n - d;
    const int left = W / 2;
    const double *row = band_in + (int64_t)d * n;
    double *orow = band_out + (int64_t)d * n;
    const double mean = diag_stats[4 * d + 0], sd = diag_stats[4 * d + 1], wgt = diag_stats[4 * d + 2];
    const double std2 = sd * sd;
    const int k0 = tid * C;                             // this thread's offsets inside a block: k0 .. k0 + C - 1

    // RAW samples of this thread's offsets in sample block m; 0 outside the diagonal.  Each lane reads C consecutive doubles
    // (the scan is serial inside a thread); the C loads of a wave cover the same cache lines.  The shift vals[x] = v + 0.001
    // (:635) is applied where the values are consumed, one block later: a fetch must not touch what it loads, or the wave
    // waits for the memory right after issuing the loads and nothing is prefetched (that was the case until r02d: the kernel
    // ran at the speed of one memory round trip per block).
    auto fetch = [&](int m, double (&r)[C]) {
        const int64_t base = (int64_t)m * W - left + k0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int64_t q = base + j;
            r[j] = (k0 + j < W && q >= 0 && q < L) ? row[q] : 0.0;
        }
    };
    // the outputs' own samples: output i = m W + k is sample offset k + left of block m -- another thread's chunk, so it is
    // re-read from memory (the same workgroup loaded it one or two blocks ago: cache hits)
    auto fetch_x = [&](int m, double (&x)[C]) {
        const int64_t base = (int64_t)m * W + k0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int64_t q = base + j;
            x[j] = (k0 + j < W && q < L) ? row[q] : 0.0;
        }
    };
    // exclusive prefix of this thread's first offset (p*) and the block totals (t*) from the chunk totals (a*)
    auto scan = [&](int buf, double a1, double a2, int ac, double &p1, double &p2, int &pc, double &t1, double &t2, int &tc) {
        double s1 = a1, s2 = a2, o1, o2;
        int sc = ac, oc;
        block_totals3<NT>(s1, s2, sc, w1[buf], w2[buf], wc[buf], o1, o2, oc, t1, t2, tc);
        p1 = o1 + lane_before(s1);
        p2 = o2 + lane_before(s2);
        pc = oc + lane_before_i(sc);
    };

    auto shifted = [](double v) { return v != 0.0 ? v + 0.001 : 0.0; };
    double vv[C], nx[C], xv[C];
    double S1[C], S2[C];                                // tail sums of the previous block from each of my offsets: T - P[k]
    int Sc[C];
    // prime: sample block m_lo -> its tail sums
    fetch(m_lo, vv);
    fetch(m_lo + 1, nx);                                // in flight while block m_lo is scanned
    if constexpr (!XLDS) fetch_x(m_lo, xv);
#pragma unroll
    for (int j = 0; j < C; ++j) vv[j] = shifted(vv[j]);
    auto park = [&](int m, const double (&v)[C]) {      // shifted samples of block m -> its ring slot
        if constexpr (XLDS) {
            double *slot = ring + (m % 3) * CAPW + k0;
#pragma unroll
            for (int j = 0; j < C; ++j) slot[j] = v[j];
        }
    };
    park(m_lo, vv);
    {
        double a1 = 0.0, a2 = 0.0, p1, p2, t1, t2;
        int ac = 0, pc, tc;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            ac += (vv[j] != 0.0) ? 1 : 0;
            a1 = a1 + vv[j];
            a2 = a2 + vv[j] * vv[j];                    // vals ** 2  (:649)
        }
        scan(0, a1, a2, ac, p1, p2, pc, t1, t2, tc);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            S1[j] = t1 - p1;
            S2[j] = t2 - p2;
            Sc[j] = tc - pc;
            pc += (vv[j] != 0.0) ? 1 : 0;
            p1 = p1 + vv[j];
            p2 = p2 + vv[j] * vv[j];
        }
    }
    int buf = 1;
    for (int m = m_lo; m < m_hi; ++m, buf ^= 1) {
#pragma unroll
        for (int j = 0; j < C; ++j) vv[j] = shifted(nx[j]);      // sample block m + 1, loaded one block ago
        park(m + 1, vv);                                // visible to the other threads after the scan's barrier
        double x[C];
        if constexpr (!XLDS) {
#pragma unroll
            for (int j = 0; j < C; ++j) x[j] = shifted(xv[j]);
        }
        if (m + 1 < m_hi) {                             // next block's loads hide under this block's arithmetic
            fetch(m + 2, nx);
            if constexpr (!XLDS) fetch_x(m + 1, xv);
        }
        double a1 = 0.0, a2 = 0.0, p1, p2, t1, t2;
        int ac = 0, pc, tc;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            ac += (vv[j] != 0.0) ? 1 : 0;
            a1 = a1 + vv[j];
            a2 = a2 + vv[j] * vv[j];
        }
        scan(buf, a1, a2, ac, p1, p2, pc, t1, t2, tc);
        if constexpr (XLDS) {                           // sample offset k + left of block m, continuing into block m + 1
            const double *sm = ring + (m % 3) * CAPW, *sn = ring + ((m + 1) % 3) * CAPW;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int e = k0 + j + left;
                x[j] = k0 + j < W ? (e < W ? sm[e] : sn[e - W]) : 0.0;
            }
        }
        const int64_t i0 = (int64_t)m * W + k0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int c = Sc[j] + pc;
            const double s1w = S1[j] + p1, s2w = S2[j] + p2;
            double zz = 0.0;
            if (x[j] != 0.0) {
                // 1/cnt and 1/(cnt-1) from ONE division; a window with c < 30 -- including c = 1, where the product form
                // gives NaN -- takes the fallback below
                const double cnt = (double)c;
                const double rcc = rcp_newton(cnt * (cnt - 1.0));   // (a table of these quotients in LDS was measured: no gain)
                const double inv_c = rcc * (cnt - 1.0), inv_cm1 = rcc * cnt;
                double var = (s2w - s1w * s1w * inv_c) * inv_cm1;        // (:650)
                double mu = s1w * inv_c;                                 // (:656)
                // non-finite local variance -> global (:653-654); fewer than 30 samples -> global mean and variance (:657-658);
                // non-finite local mean -> global (:660-661): one select per quantity
                if (c < 30 || !isfinite(var)) var = std2;
                if (c < 30 || !isfinite(mu)) mu = mean;
                zz = (x[j] - mu) * rsqrt_newton(var);                    // (:663-665)
                if (!isfinite(zz)) zz = 0.0;                             // (:666)
                zz = zz * wgt;                                           // (:667)
            }
            if (k0 + j < W && i0 + j < n) orow[i0 + j] = zz;             // positions past the diagonal's end receive 0
            // this block's tail sums for the next output block
            S1[j] = t1 - p1;
            S2[j] = t2 - p2;
            Sc[j] = tc - pc;
            pc += (vv[j] != 0.0) ? 1 : 0;
            p1 = p1 + vv[j];
            p2 = p2 + vv[j] * vv[j];
        }
    }
}

// Branch B (mustache.py:671-685): plain per-diagonal z-score for d < min(dpx, n); other diagonals pass through
// (after the nan_to_num at :673).
__global__ void __launch_bounds__(kThreads)
normalize_global_kernel(const double *__restrict__ band_in, double *__restrict__ band_out, int64_t n, int dlimit,
                        const double *__restrict__ diag_stats) {
    const int d = blockIdx.y;
    const int64_t L = n - d;
    const double mean = diag_stats[4 * d + 0], sd = diag_stats[4 * d + 1];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        double v = i < L ? band_in[(int64_t)d * n + i] : 0.0;
        if (!isfinite(v)) v = 0.0;
        if (v != 0.0 && d < dlimit) {
            v = (v - mean) / sd;
            if (!isfinite(v)) v = 0.0;
        }
        band_out[(int64_t)d * n + i] = v;
    }
}

// Dense blocks from the band: one workgroup = a 16-row x 128-column strip of one block.  The strip's diagonals are
// staged through LDS (reads: 16 contiguous doubles per diagonal), then written out as full 1 KB row segments with
// 16-byte stores -- the kernel is a pure HBM writer (9 B per pixel) and wants long contiguous bursts.
constexpr int kTR = 16, kTC = 128;
constexpr int kND = kTR + kTC - 1;      // diagonals crossing a strip
constexpr int kTP = kTR + 1;            // LDS pitch

constexpr int kStrips = 8;             // 16-row strips per workgroup: 128 x 128 pixels, so the grid is not dispatch-bound

__global__ void __launch_bounds__(kThreads)
blocks_from_band_kernel(const double *__restrict__ band, int64_t n, int dpx, const int64_t *__restrict__ starts,
                        int CH, double *__restrict__ c, uint8_t *__restrict__ nz, uint32_t *__restrict__ nz_count) {
    __shared__ double tile[kND * kTP];
    const int b = blockIdx.z;
    const int c0 = blockIdx.x * kTC;
    const int64_t start = starts[b];
    double *cb = c + (size_t)b * CH * CH;
    uint8_t *nb = nz + (size_t)b * CH * CH;
    const int tid = threadIdx.x;
    const bool pairs_ok = (CH & 1) == 0;        // 16-byte stores need even row length (rows start 16-byte aligned)
    uint32_t local = 0;
    for (int sidx = 0; sidx < kStrips; ++sidx) {
        const int r0 = (blockIdx.y * kStrips + sidx) * kTR;
        if (r0 >= CH) break;
        const int off_lo = c0 - r0 - (kTR - 1), off_hi = c0 - r0 + (kTC - 1);   // range of col - row inside the strip
        const bool has_data = off_hi >= 0 && off_lo <= dpx + 1;                  // workgroup-uniform
        if (has_data) {
            __syncthreads();                     // previous strip's readers are done with the tile
            // tile[(d - off_lo) * kTP + ii] = band[d][start + r0 + ii]
            for (int idx = tid; idx < kND * kTR; idx += kThreads) {
                const int dd = idx / kTR, ii = idx - dd * kTR;
                const int d = off_lo + dd;
                const int64_t i = start + r0 + ii;
                double v = 0.0;
                if (d >= 0 && d <= dpx + 1 && i >= 0 && i + d < n && r0 + ii < CH) v = band[(int64_t)d * n + i];
                tile[dd * kTP + ii] = v;
            }
            __syncthreads();
        }
        for (int idx = tid; idx < kTR * (kTC / 2); idx += kThreads) {
            const int rr = idx / (kTC / 2), cp = idx - rr * (kTC / 2);
            const int row = r0 + rr, col = c0 + 2 * cp;
            if (row >= CH || col >= CH) continue;
            double val[2];
            uint32_t t[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int off = col + e - row;
                double raw = 0.0;
                if (has_data && off >= 0 && off <= dpx + 1) raw = tile[(off - off_lo) * kTP + rr];
                t[e] = (raw != 0.0 && off >= 4) ? 1u : 0u;                                   // (:699)
                val[e] = (off <= 4 || off >= dpx + 1) ? 2.0 : raw;                           // (:703-706)
            }
            const size_t at = (size_t)row * CH + col;
            if (pairs_ok && col + 1 < CH) {
                *reinterpret_cast<double2 *>(cb + at) = make_double2(val[0], val[1]);
                *reinterpret_cast<uint16_t *>(nb + at) = (uint16_t)(t[0] | (t[1] << 8));
                local += t[0] + t[1];
            } else {
                cb[at] = val[0];
                nb[at] = (uint8_t)t[0];
                local += t[0];
                if (col + 1 < CH) {
                    cb[at + 1] = val[1];
                    nb[at + 1] = (uint8_t)t[1];
                    local += t[1];
                }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
    if ((tid & 63) == 0 && local) atomicAdd(nz_count + b, local);
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

extern "C" int mst_band_from_coo(const int64_t *x, const int64_t *y, const double *v, int64_t nnz, int64_t n,
                                 int32_t dpx, double *band, void *stream) {
    MST_RANGE("read: mst_band_from_coo");
    if (!band || n <= 0 || dpx < 0 || nnz < 0 || (nnz > 0 && (!x || !y || !v)))
        return mst::fail(MST_E_ARG, "mst_band_from_coo: bad argument");
    hipStream_t s = mst::as_stream(stream);
    MST_HIP(hipMemsetAsync(band, 0, sizeof(double) * (size_t)(dpx + 2) * n, s));
    if (nnz == 0) return MST_OK;
    int64_t want = (nnz + kThreads - 1) / kThreads;
    band_scatter_kernel<<<(int)(want < 65536 ? want : 65536), kThreads, 0, s>>>(x, y, v, nnz, n, dpx, band);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_band_from_packed(const int32_t *x, const int32_t *dist, const float *v, int64_t nnz, int64_t n,
                                    int32_t dpx, double *band, void *stream) {
    MST_RANGE("read: mst_band_from_packed");
    if (!band || n <= 0 || dpx < 0 || nnz < 0 || (nnz > 0 && (!x || !dist || !v)))
        return mst::fail(MST_E_ARG, "mst_band_from_packed: bad argument");
    hipStream_t s = mst::as_stream(stream);
    MST_HIP(hipMemsetAsync(band, 0, sizeof(double) * (size_t)(dpx + 2) * n, s));
    if (nnz == 0) return MST_OK;
    int64_t want = (nnz + kThreads - 1) / kThreads;
    band_scatter_packed_kernel<<<(int)(want < 65536 ? want : 65536), kThreads, 0, s>>>(x, dist, v, nnz, n, dpx, band);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_band_scatter_packed(const int32_t *x, const void *dist, int32_t dist_bytes, const float *v, int64_t nnz,
                                       int64_t n, int32_t dpx, double *band, void *stream) {
    MST_RANGE("read: mst_band_scatter_packed");
    if (!band || n <= 0 || dpx < 0 || nnz < 0 || (nnz > 0 && (!x || !dist || !v)) || (dist_bytes != 2 && dist_bytes != 4) ||
        (dist_bytes == 2 && dpx + 1 > 65535))
        return mst::fail(MST_E_ARG, "mst_band_scatter_packed: bad argument (dist_bytes 2 or 4; 2 needs dpx + 1 <= 65535)");
    if (nnz == 0) return MST_OK;
    hipStream_t s = mst::as_stream(stream);
    int64_t want = (nnz + kThreads - 1) / kThreads;
    const int g = (int)(want < 65536 ? want : 65536);
    if (dist_bytes == 2)
        band_scatter_packed16_kernel<<<g, kThreads, 0, s>>>(x, (const uint16_t *)dist, v, nnz, n, dpx, band);
    else
        band_scatter_packed_kernel<<<g, kThreads, 0, s>>>(x, (const int32_t *)dist, v, nnz, n, dpx, band);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_band_verify_packed(const int32_t *x, const void *dist, int32_t dist_bytes, const float *v, int64_t nnz,
                                      int64_t n, int32_t dpx, const double *band, uint64_t *mismatches, void *stream) {
    if (!band || !mismatches || n <= 0 || dpx < 0 || nnz < 0 || (nnz > 0 && (!x || !dist || !v)) ||
        (dist_bytes != 2 && dist_bytes != 4))
        return mst::fail(MST_E_ARG, "mst_band_verify_packed: bad argument");
    if (nnz == 0) return MST_OK;
    hipStream_t s = mst::as_stream(stream);
    int64_t want = (nnz + kThreads - 1) / kThreads;
    const int g = (int)(want < 65536 ? want : 65536);
    unsigned long long *m = reinterpret_cast<unsigned long long *>(mismatches);
    if (dist_bytes == 2)
        band_verify_packed_kernel<uint16_t><<<g, kThreads, 0, s>>>(x, (const uint16_t *)dist, v, nnz, n, dpx, band, m);
    else
        band_verify_packed_kernel<int32_t><<<g, kThreads, 0, s>>>(x, (const int32_t *)dist, v, nnz, n, dpx, band, m);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_band_to_coo(const double *band, const int64_t *x, const int64_t *y, int64_t nnz, int64_t n,
                               int32_t dpx, double *v, void *stream) {
    if (!band || n <= 0 || dpx < 0 || nnz < 0 || (nnz > 0 && (!x || !y || !v)))
        return mst::fail(MST_E_ARG, "mst_band_to_coo: bad argument");
    if (nnz == 0) return MST_OK;
    int64_t want = (nnz + kThreads - 1) / kThreads;
    band_gather_kernel<<<(int)(want < 65536 ? want : 65536), kThreads, 0, mst::as_stream(stream)>>>(band, x, y, nnz,
                                                                                                   n, dpx, v);
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_normalize_band(const double *band_in, double *band_out, int64_t n, int32_t dpx, int32_t window,
                                  int32_t local, double *diag_stats, void *stream) {
    MST_RANGE("normalise: mst_normalize_band");
    if (!band_in || !band_out || !diag_stats || band_in == band_out || n <= 0 || dpx < 0 || dpx + 2 > 65535)
        return mst::fail(MST_E_ARG, "mst_normalize_band: bad argument (out of place, dpx + 2 <= 65535)");
#ifndef MST_PROFILE
    if (local != 0 && local != 1)
        return mst::fail(MST_E_ARG, "mst_normalize_band: local = %d selects a cross-check kernel that only PROFILE builds "
                         "carry (make PROFILE=1 -> libmustache_hip_profile.so); the product library takes 0 or 1", local);
#endif
    hipStream_t s = mst::as_stream(stream);
    const int nd = dpx + 2;
    diag_stats_kernel<<<nd, kThreads, 0, s>>>(band_in, n, diag_stats);
    MST_LAUNCH_CHECK();
    // walking kernel: windows up to 4096 bins at full speed; wider ones (resolutions below ~490 bp) go through the blocked-sum
    // kernel further down
    if (local == 1 && window >= 2 && window <= 512 * 8) {
        // default: the walking kernel -- one scan per sample, blocks of `window` samples along each diagonal
        const int nb = (int)((n + window - 1) / window);                  // output blocks per diagonal (covers [0, n))
        int run = (int)(((int64_t)nb * nd + 8191) / 8192);                // >= ~8 k workgroups when the band is large enough
        run = run < 4 ? (nb < 4 ? nb : 4) : (run > 32 ? 32 : run);        // priming costs one extra block scan per run
        const int rpd = (nb + run - 1) / run;
#define MST_WALK_CASE(NT_, C_)                                                                                          \
    {                                                                                                                  \
        normalize_walk_kernel<NT_, C_><<<(unsigned)(rpd * nd), NT_, 0, s>>>(band_in, band_out, n, window, diag_stats,   \
                                                                           nb, run, rpd);                            \
    }
        // 4 samples per thread keep the kernel at ~120 VGPRs (4 waves per SIMD); wider windows take more threads, not more
        // samples per thread (8 per thread need ~200 VGPRs)
        if (window <= 256 * 2) MST_WALK_CASE(256, 2)
        else if (window <= 256 * 4) MST_WALK_CASE(256, 4)
        // (measured at W = 2000: <512, 4> 3.4 ms, <256, 8> 4.0 ms, <1024, 2> 5.8 ms)
        else if (window <= 512 * 4) MST_WALK_CASE(512, 4)
        else MST_WALK_CASE(512, 8)
#undef MST_WALK_CASE
        MST_LAUNCH_CHECK();
        return MST_OK;
    }
#ifdef MST_PROFILE
    if (local && window >= 2) {
        // local == 3: the segment / prefix-array kernel (2 doubles + 1 int per sample in LDS), kept selectable for cross-checks
        const int tile = kPSeg + window - 1;
        const int chunk = ((tile + 1 + kPThreads - 1) / kPThreads) | 1;  // samples per thread, odd
        const size_t plds = (sizeof(double) * 2 + sizeof(int)) * (size_t)kPThreads * chunk + sizeof(double) * kPSeg + 16;
        if (plds <= 80 * 1024 && chunk <= kPMaxChunk && local == 3) {
            const int nseg = (int)((n + kPSeg - 1) / kPSeg);            // covers [0, n): the kernel also writes the zero tails
            const int64_t total = (int64_t)nseg * nd;
            const int64_t want_wgs = 256 * 2 * 8;                        // 8 waves of workgroups over 256 CUs x 2 resident
            const int ipw = (int)((total + want_wgs - 1) / want_wgs);
            const int64_t wgs = (total + ipw - 1) / ipw;
#define MST_PREFIX_CASE(C_)                                                                                            \
    case C_:                                                                                                           \
        MST_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&normalize_prefix_kernel<C_>),                      \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));                           \
        normalize_prefix_kernel<C_><<<(unsigned)wgs, kPThreads, plds, s>>>(band_in, band_out, n, window, diag_stats,  \
                                                                          nseg, nd, ipw);                            \
        break;
            switch (chunk) {
                MST_PREFIX_CASE(1) MST_PREFIX_CASE(3) MST_PREFIX_CASE(5) MST_PREFIX_CASE(7) MST_PREFIX_CASE(9)
                MST_PREFIX_CASE(11) MST_PREFIX_CASE(13) MST_PREFIX_CASE(15) MST_PREFIX_CASE(17)
                default: return mst::fail(MST_E_ARG, "mst_normalize_band: internal chunk %d", chunk);
            }
#undef MST_PREFIX_CASE
            MST_LAUNCH_CHECK();
            return MST_OK;
        }
    }
#endif
    if (local) {
        // wide windows: blocked-sum kernel.  LDS: 2 doubles per staged sample + 2 doubles and an int per 16-sample block, for up to
        // SEG + W + 2 * 16 samples; beyond ~8400 bins: 1 double per sample and 32-sample blocks (kBlkWide; see the kernel)
        auto lds_need = [&](size_t blk, size_t arrays) {
            const size_t nblk = (size_t)(kSeg + window + 2 * blk + blk - 1) / blk;
            return sizeof(double) * (arrays * nblk * blk + 2 * (nblk + 1)) + sizeof(int) * (nblk + 1) + 16;
        };
        const size_t lds = window < 2 ? 0 : lds_need(kBlk, 2), lds_wide = window < 2 ? 0 : lds_need(kBlkWide, 1);
        const bool wide = lds > 160 * 1024;
        if (window < 2 || (wide && lds_wide > 160 * 1024))
            return mst::fail(MST_E_ARG,
                             "mst_normalize_band: window of %d bins (= 2 Mb / resolution) is outside [2, 16384]: the sliding-window "
                             "normalisation supports resolutions down to ~125 bp",
                             window);
        // rows are written for i < n - d only; clear the tails so the output band is fully defined
        MST_HIP(hipMemsetAsync(band_out, 0, sizeof(double) * (size_t)nd * n, s));
        dim3 grid((unsigned)((n + kSeg - 1) / kSeg), nd);
        if (wide) {
            static unsigned long long lds_allowed_wide = 0;      // per device (mst_common.h)
            MST_HIP(mst::allow_dynamic_lds(reinterpret_cast<const void *>(&normalize_local_kernel<kBlkWide, false, true>), 160 * 1024,
                                           &lds_allowed_wide));
            normalize_local_kernel<kBlkWide, false, true><<<grid, kThreads, lds_wide, s>>>(band_in, band_out, n, window, diag_stats);
        } else {
            static unsigned long long lds_allowed = 0;
            MST_HIP(mst::allow_dynamic_lds(reinterpret_cast<const void *>(&normalize_local_kernel<kBlk, true, false>), 160 * 1024,
                                           &lds_allowed));
            normalize_local_kernel<kBlk, true, false><<<grid, kThreads, lds, s>>>(band_in, band_out, n, window, diag_stats);
        }
    } else {
        const int dlimit = (int64_t)dpx < n ? dpx : (int)n;     // range(min(distance_in_px, n))   (:674-675)
        int64_t want = (n + kThreads - 1) / kThreads;
        dim3 grid((unsigned)(want < 1024 ? want : 1024), nd);
        normalize_global_kernel<<<grid, kThreads, 0, s>>>(band_in, band_out, n, dlimit, diag_stats);
    }
    MST_LAUNCH_CHECK();
    return MST_OK;
}

extern "C" int mst_blocks_from_band(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t B,
                                    int32_t CH, double *c, uint8_t *nz, uint32_t *nz_count, void *stream) {
    if (!band || !starts || !c || !nz || !nz_count || n <= 0 || dpx < 0 || B <= 0 || B > 65535 || CH <= 0)
        return mst::fail(MST_E_ARG, "mst_blocks_from_band: bad argument");
    hipStream_t s = mst::as_stream(stream);
    int64_t *d_starts = nullptr;
    MST_HIP(hipMallocAsync((void **)&d_starts, sizeof(int64_t) * B, s));
    MST_HIP(mst::upload_small(d_starts, starts, sizeof(int64_t) * B, s));
    MST_HIP(hipMemsetAsync(nz_count, 0, sizeof(uint32_t) * B, s));
    blocks_from_band_kernel<<<dim3((CH + kTC - 1) / kTC, (CH + kTR * kStrips - 1) / (kTR * kStrips), B), kThreads, 0, s>>>(
        band, n, dpx, d_starts, CH, c, nz, nz_count);
    MST_LAUNCH_CHECK();
    MST_HIP(hipFreeAsync(d_starts, s));
    return MST_OK;
}
