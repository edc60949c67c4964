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
// What this kernel does instead: one 256-thread workgroup owns a 30 x 62 pixel tile (a 32 x 64 "region" with the
// 1-pixel ring the 3x3 max needs).  It loads the tile of c with a 14-pixel reflect halo into LDS ONCE (~15 B/pixel
// of HBM/L2 reads), then walks all 24 levels out of LDS:
//   V pass  (axis 0): a thread produces 8 vertically consecutive samples of one column from a register window of
//                     8+2r samples (c tile stored transposed -> the window is contiguous, read as ds_read_b128;
//                     lanes run along columns with a pitch of 30 mod 32 doubles -> conflict-free), result -> LDS vb
//   H pass  (axis 1): a thread produces 8 horizontally consecutive samples of one row (same access pattern on vb),
//                     result stays in registers
//   DoG in registers; zero-padded 3x3 max = horizontal 3-max in registers (2 edge samples through a 4 KB LDS strip)
//   then vertical 3-max across lanes with DPP wave shifts (lanes of a wave are consecutive rows); the 5-term sieve
//   and the running best/level per pixel live in registers across all levels and both octaves; per-level min / sum
//   of |D_c| are reduced per workgroup in a fixed order.
// Only the found pixels (~1 % of the block) and 2 x 18 partial statistics per tile are written back.
// 77 KB of LDS per workgroup -> two independent workgroups per CU, so one's LDS window loads overlap the other's
// FP64 work instead of every wave of the CU alternating between the two in barrier lock-step.
// The kernel is FP64-VALU bound (~1150 non-fusable flops per pixel for the blurs alone), not HBM bound.
// Tiles sit on a lattice anchored at the chromosome's origin, and a tile that lies inside two consecutive (overlapping) blocks of
// a launch with its whole blur halo is computed once for both (struct WorkItem, build_items): the reference's blocks overlap by
// up to half their edge (mustache.py:899-908) and recompute those pixels block by block.
//
// Bit-exactness: the tap order is SciPy's C correlate1d on a symmetric kernel,
//     t = x[c]*w0;  for j = r..1:  t += (x[c-j] + x[c+j]) * w[j]
// axis 0 first, float64 intermediate, mode='reflect'; compiled with -ffp-contract=off so no FMA is formed.
// The taps themselves come from the host (NumPy), see mustache_amd/levels.py.
// Build flags for this file add -fno-honor-nans -mno-amdgpu-ieee: inputs are finite (checked through the level
// statistics), and with IEEE mode off v_max_f64 needs no canonicalising pre-pass; no value-changing fast-math
// flag (reassociation, contraction, reciprocal) is enabled.
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "mst_common.h"

#include "mst_fir.h"

namespace {

__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }

// value held by the previous / next lane: DPP wave shifts, no LDS traffic.  bound_ctrl is set, so the lane without a
// source (0 for prev, 63 for next) reads 0.0 and no `old` operand has to be copied in first; those lanes hold ring
// rows whose maxima are never used.
__device__ __forceinline__ double lane_prev(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x138, 0xf, 0xf, true);   // wave_shr:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_next(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x130, 0xf, 0xf, true);   // wave_shl:1
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// 64-lane reductions with DPP moves only (no LDS round trips): inclusive scans inside the four 16-lane rows
// (row_shr 1, 2, 4, 8), then row_bcast15 / row_bcast31 carry the row totals upward; lane 63 ends up with the total.
// The combination order is fixed, so the result is deterministic.  ONLY lane 63 is meaningful afterwards: the lanes a
// shift or a row mask leaves without a source are not given an identity value (that costs four extra moves per step) --
// they read 0 / keep an unspecified register, and none of them lies on the path into lane 63 (lane 15 of a row combines
// lanes 14, 13, 11, 7 after steps 1, 2, 4, 8, each of which only ever read lanes with a valid source; the broadcasts
// read lanes 15 / 47 and 31).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_fetch(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, ROW_MASK, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ void wave_reduce_min_sum(double &mn, double &sm) {
#define MST_STEP(CTRL, MASK)                                           \
    {                                                                  \
        const double a_ = dpp_fetch<CTRL, MASK>(mn);                   \
        const double b_ = dpp_fetch<CTRL, MASK>(sm);                   \
        mn = a_ < mn ? a_ : mn;                                        \
        sm = sm + b_;                                                  \
    }
    MST_STEP(0x111, 0xf)   // row_shr:1
    MST_STEP(0x112, 0xf)   // row_shr:2
    MST_STEP(0x114, 0xf)   // row_shr:4
    MST_STEP(0x118, 0xf)   // row_shr:8
    MST_STEP(0x142, 0xa)   // row_bcast:15 -> rows 1 and 3
    MST_STEP(0x143, 0xc)   // row_bcast:31 -> rows 2 and 3
#undef MST_STEP
}

// K output samples from a register window of K + 2R input samples, SciPy's order per sample:
//     t = x[c]*w0;  for j = R..1:  t += (x[c-j] + x[c+j]) * w[j]
// The K accumulation chains are independent; the loops are written tap-major so the K adds / muls / adds of one
// tap are adjacent in program order and the FP64 pipe always has K independent instructions to issue (a
// sample-major order leaves one serial add->mul->add chain per sample and stalls on every instruction).
template <class T, int R>
__device__ __forceinline__ void blur_level(const double *ct, double *vb, const double (&w)[T::RMAX + 1], int tid,
                                           const double *vsrc, double *vdst, const double *hsrc, double (&g)[T::K],
                                           int variant, unsigned long long *tr MST_INJECT_ARG) {
    vpass<T, R>(ct, vb, w, tid, vsrc, vdst, variant MST_INJECT_PASS);
    MST_STAMP(tr, 1)
    __syncthreads();
    MST_STAMP(tr, 2)
    if (MST_VARIANT(32)) {              // [ablation 32] axis-0 pass only (with 24 = the other half: the "roles" experiment)
#pragma unroll
        for (int k = 0; k < T::K; ++k) g[k] = 0.0;
        return;
    }
    hpass<T, R>(hsrc, w, g);
    MST_STAMP(tr, 3)
}

template <class T>
__device__ __forceinline__ void blur_dispatch(int r, const double *ct, double *vb, const double (&wg)[T::RMAX + 1],
                                              int tid, const double *vsrc, double *vdst, const double *hsrc,
                                              double (&g)[T::K], int variant, unsigned long long *tr MST_INJECT_ARG) {
#define MST_CASE(R_)                                                  \
    case R_:                                                          \
        if constexpr (R_ <= T::RMAX) blur_level<T, R_>(ct, vb, wg, tid, vsrc, vdst, hsrc, g, variant, tr MST_INJECT_PASS); \
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

// Where the block pixels come from.  Dense: caller-built filled blocks + nz masks (the reference's `c`).  Band: straight from
// the chromosome's diagonal-major band -- the fills of mustache.py:703-706 and the nz rule of :699 are applied while the
// tile is staged, so dense blocks are never materialised (no 9 B/pixel write + re-read, no 18 GB for chr1 @ 1 kb).
struct BandSrc {
    const double *band;        // band[d * n + i] = pixel (i, i + d)
    int64_t n;
    int dpx;
    const int64_t *starts;     // dev [B]: block origins
    uint32_t *nz_count;        // dev [B]: out, number of tested pixels per block
    const double *band2;       // two-sample launches: blocks [split, B) are windows of this band (same n, dpx); else unused
    int split;                 // first block of band2 (INT_MAX: one band)
};

// One workgroup's job: the tile whose region (interior + ring) starts at (y0, x0) of block b.  Consecutive blocks of a
// chromosome overlap by half their edge (mustache.py:899-908), and a Gaussian level at a pixel depends on c within its blur
// radius only: a tile that lies inside BOTH blocks, halo included, sees the same c values, fills and tested pixels in either
// -- its DoG values, maxima, sieve decisions and statistics are the same bits.  Such a tile is computed ONCE (for block b) and
// its found records, tested-pixel count and level statistics are delivered to block b2 as well (pixel indices shifted by
// delta2 = start_b - start_b2 along both axes); b2 < 0: the tile belongs to one block only.  The host builds the list
// (build_items): tiles sit on a lattice anchored at chromosome coordinate 0, so that overlapping blocks cut the same tiles.
struct WorkItem {
    int32_t y0, x0, b, delta2;      // delta2 != 0: also delivered to block b + 1 (delta2 = start_b - start_{b+1} < 0)
};

// The kernel's argument block as the hardware lays it out (each argument at its natural alignment, in order): the wide-radius
// instantiation re-reads the arguments only its EPILOGUE needs from there, so that they do not occupy scalar registers through
// the level loop (29 taps of two registers each leave none to spare; the default tile is untouched).
struct KernArgs {
    const double *c;
    const uint8_t *nz;
    BandSrc src;
    int CH;
    const DevLevels *lv;
    mst_found *found;
    uint32_t found_cap;
    uint32_t *found_count;
    double *partial;
    const WorkItem *items;
    int n_items, n_tested, skip_empty, variant;
    unsigned long long *trace;
};

template <class T, bool BAND>
__global__ void __launch_bounds__(T::NT, T::MINW)
scale_space_kernel(const double *__restrict__ c, const uint8_t *__restrict__ nz, BandSrc src, int CH,
                   const DevLevels *__restrict__ lv, mst_found *__restrict__ found, uint32_t found_cap,
                   uint32_t *__restrict__ found_count, double *__restrict__ partial, const WorkItem *__restrict__ items,
                   int n_items, int n_tested, int skip_empty, int variant, unsigned long long *__restrict__ trace) {
    constexpr int K = T::K, RGR = T::RGR, RGC = T::RGC, RMAX = T::RMAX;
    extern __shared__ __align__(16) double lds[];
    double *ct = lds;
    double *vb = ct + T::CT_ELEMS;
    double *de = vb + T::VB_ELEMS;
    double *st = de + T::DE_ELEMS;

    const int tid = threadIdx.x;
    // XCD-aware order: hardware places workgroup i on XCD i % 8 (gridDim.x is a multiple of 8), so give each XCD a
    // contiguous run of the item list -- neighbouring tiles share their halo through that XCD's L2.  The host orders the list
    // so that every XCD's run holds the same mix of work (build_items).
    const int per_xcd = gridDim.x >> 3;
    const int slot = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (slot >= n_items) return;
    const WorkItem item = items[slot];
    const int b = item.b, b2 = item.delta2 != 0 ? item.b + 1 : -1;
    const int y0 = item.y0, x0 = item.x0;  // block coordinates of region (0, 0)

    const int rr = tid % RGR;  // region row owned in the H pass; consecutive lanes = consecutive rows
    const int cg = tid / RGR;  // column group
    const int gy = y0 + rr;
    const bool row_in = gy >= 0 && gy < CH;
    const bool row_own = row_in && rr >= 1 && rr <= T::ITR;
    double *part = partial + (size_t)slot * n_tested * 2;

    uint32_t in_mask = 0, nz_mask = 0;  // per-k bits: column inside the block / tested pixel owned by this thread
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int gx = x0 + cg * K + k;
        if (row_in && gx >= 0 && gx < CH) in_mask |= 1u << k;
    }

    if constexpr (!BAND) {
        const double *cb = c + (size_t)b * CH * CH;
        const uint8_t *nb = nz + (size_t)b * CH * CH;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int rc = cg * K + k;
            const int gx = x0 + rc;
            if (row_own && gx >= 0 && gx < CH && rc >= 1 && rc <= T::ITC && nb[(size_t)gy * CH + gx]) nz_mask |= 1u << k;
        }
        const int any_nz = __syncthreads_or(nz_mask != 0);
        if (!any_nz && skip_empty) {
            for (int t = tid; t < n_tested; t += T::NT) {
                part[2 * t] = INFINITY;
                part[2 * t + 1] = 0.0;
            }
            return;
        }
        // ---- stage the c tile (reflect halo) in LDS, transposed: the only bulk HBM/L2 read of the kernel.
        // A wave takes whole rows (row reflection is wave-uniform); each lane's column reflections are computed once.
        constexpr int CPL = (T::CTC + 63) / 64;          // columns per lane
        int sx[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) sx[q] = reflect_idx(x0 - RMAX + (tid & 63) + 64 * q, CH);
        for (int i = tid >> 6; i < T::CTR; i += T::NW) {
            const double *srow = cb + (size_t)reflect_idx(y0 - RMAX + i, CH) * CH;
#pragma unroll
            for (int q = 0; q < CPL; ++q) {
                const int j = (tid & 63) + 64 * q;
                if (j < T::CTC) ct[j * T::CTP + i] = srow[sx[q]];
            }
        }
        __syncthreads();
    } else {
        // ---- band source.  Pixel (by, bx) of the block: off = bx - by;
        //   raw    = band[off][start + by]            for 0 <= off <= dpx+1 and start + bx < n, else 0
        //   tested = raw != 0 and off >= 4            (mustache.py:699, before the fills)
        //   value  = 2 where off <= 4 or off >= dpx+1 (mustache.py:703-706), else raw
        const int64_t start = src.starts[b];
        const int64_t n = src.n;
        const int dpx = src.dpx;
        const double *const bandp = b >= src.split ? src.band2 : src.band;       // (workgroup-uniform)
        uint8_t *nzb = reinterpret_cast<uint8_t *>(vb);      // [RGR][RGC] tested flags of the region; vb is free until the V pass
        const int Y0 = y0 - RMAX, X0 = x0 - RMAX;
        const bool inner = Y0 >= 0 && X0 >= 0 && Y0 + T::CTR <= CH && X0 + T::CTC <= CH;   // no reflection in this tile
        // the tile incl. its halo lies entirely in one of the constant regions, col - row <= 3 or >= dpx + 2 (diagonals 4 and
        // dpx + 1 are filled too, but their pixels can be TESTED, mustache.py:699 vs :703-706): every sample is the fill
        // value and no pixel is tested -- the same LDS contents as the general walk below, written with wide stores
        const bool constant = inner && (X0 + T::CTC - 1 - Y0 <= 3 || X0 - (Y0 + T::CTR - 1) >= dpx + 2);
        if (constant) {
            double2 *ct2 = reinterpret_cast<double2 *>(ct);
            for (int i = tid; i < T::CT_ELEMS / 2; i += T::NT) ct2[i] = make_double2(2.0, 2.0);
            uint4 *nz4 = reinterpret_cast<uint4 *>(nzb);
            static_assert((RGR * RGC) % 16 == 0, "nzb is cleared in 16-byte pieces");
            for (int i = tid; i < RGR * RGC / 16; i += T::NT) nz4[i] = make_uint4(0, 0, 0, 0);
        } else if (inner) {
            // walk the tile by diagonals: pixels (Y0+i, X0+i+dd) of one diagonal are CONTIGUOUS in band row off = X0-Y0+dd,
            // so a wave reads one 480-byte run per diagonal; the fill decision is wave-uniform.  All loads of a wave's
            // share (U diagonals per round, two rounds) are issued before the first LDS store: the staging is latency
            // bound (each diagonal lives in a different band row), so memory-level parallelism is what shortens it.
            constexpr int ND = T::CTR + T::CTC - 1;
            constexpr int PER = (T::CTR + 63) / 64;      // elements of one diagonal per lane
            constexpr int U2 = (ND + 2 * T::NW - 1) / (2 * T::NW);
            constexpr int U = U2 * PER <= 20 ? U2 : 20 / PER;       // diagonals in flight per wave: loads first, then the stores
            for (int q0 = tid >> 6; q0 < ND; q0 += T::NW * U) {
                double rawv[U][PER];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int q = q0 + u * T::NW;
                    const int dd = q - (T::CTR - 1);
                    const int off = X0 - Y0 + dd;
                    const int i_lo = dd < 0 ? -dd : 0;
                    const int i_hi = T::CTC - dd < T::CTR ? T::CTC - dd : T::CTR;
                    const bool in_band = q < ND && off >= 0 && off <= dpx + 1;
                    const double *brow = bandp + (int64_t)(in_band ? off : 0) * n + start + Y0;
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        const int i = i_lo + (tid & 63) + 64 * e;
                        rawv[u][e] = (in_band && i < i_hi && start + X0 + i + dd < n) ? brow[i] : 0.0;
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
                    const bool filled = off <= 4 || off >= dpx + 1;
#pragma unroll
                    for (int e = 0; e < PER; ++e) {
                        const int i = i_lo + (tid & 63) + 64 * e;
                        if (i >= i_hi) continue;
                        const int j = i + dd;
                        const double raw = rawv[u][e];
                        ct[j * T::CTP + i] = filled ? 2.0 : raw;
                        const int ri = i - RMAX, rj = j - RMAX;
                        if (ri >= 0 && ri < RGR && rj >= 0 && rj < RGC)
                            nzb[ri * RGC + rj] = (raw != 0.0 && off >= 4) ? 1 : 0;
                    }
                }
            }
        } else {
            // border tiles: element-wise with reflection (few tiles; loads are served by L2)
            for (int idx = tid; idx < T::CTR * T::CTC; idx += T::NT) {
                const int i = idx / T::CTC, j = idx - i * T::CTC;
                const int uy = Y0 + i, ux = X0 + j;                       // unreflected block coordinates
                const int by = reflect_idx(uy, CH), bx = reflect_idx(ux, CH);
                const int off = bx - by;
                double raw = 0.0;
                if (off >= 0 && off <= dpx + 1 && start + bx < n) raw = bandp[(int64_t)off * n + start + by];
                ct[j * T::CTP + i] = (off <= 4 || off >= dpx + 1) ? 2.0 : raw;
                const int ri = i - RMAX, rj = j - RMAX;
                if (ri >= 0 && ri < RGR && rj >= 0 && rj < RGC) {
                    const bool inside = uy >= 0 && uy < CH && ux >= 0 && ux < CH;
                    nzb[ri * RGC + rj] = (inside && raw != 0.0 && off >= 4) ? 1 : 0;
                }
            }
        }
        __syncthreads();
        uint32_t mine = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int rc = cg * K + k;
            if (row_own && ((in_mask >> k) & 1u) && rc >= 1 && rc <= T::ITC && nzb[rr * RGC + rc]) nz_mask |= 1u << k;
        }
        mine = __builtin_popcount(nz_mask);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
        if ((tid & 63) == 0 && mine) {                                         // integer: exact in any order
            atomicAdd(src.nz_count + b, mine);
            if (b2 >= 0) atomicAdd(src.nz_count + b2, mine);
        }
        const int any_nz = __syncthreads_or(nz_mask != 0);                  // also fences nzb reads before vb is reused
        if (!any_nz && skip_empty) {
            for (int t = tid; t < n_tested; t += T::NT) {
                part[2 * t] = INFINITY;
                part[2 * t + 1] = 0.0;
            }
            return;
        }
    }

    // per-pixel rolling state across levels.  For the tested level c the sieve needs
    //   D_c == M_c (ec),  D_{c-1} == M_{c-1} (ep),  D_c > M_{c-1} (gp),  and against the incoming level D_c > M_{c+1};
    // the first three are decided when their level arrives and kept as one bit per pixel.
    double gprev[K], Dc[K], Mc[K], best[K];
    uint32_t lvl[K];
    uint32_t ep = 0, ec = 0, gp = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        gprev[k] = Dc[k] = Mc[k] = best[k] = 0.0;
        lvl[k] = 0;
    }
    const int wave = tid >> 6, lane = tid & 63;
    const bool wave_has_nz = __any(nz_mask != 0);
    const bool wave_all_in = __all(in_mask == (1u << K) - 1u);   // no pixel of this wave lies outside the block
    double *de_mine = de + (cg * 2) * RGR + rr;                                   // [cg][0 = left edge, 1 = right edge][rr]
    const double *de_left = de + ((cg > 0 ? cg - 1 : 0) * 2 + 1) * RGR + rr;      // left neighbour's right edge
    const double *de_right = de + ((cg < T::NCG - 1 ? cg + 1 : cg) * 2) * RGR + rr;  // right neighbour's left edge

    // thread-constant LDS addresses of the blur passes
    constexpr int V_MAINC = T::NT / (T::RGR / K);
    const int v_rgp = tid / V_MAINC, v_col = tid - v_rgp * V_MAINC;
    const double *vsrc = ct + v_col * T::CTP + v_rgp * K;
    double *vdst = vb + (v_rgp * K) * T::VP + v_col;
    const double *hsrc = vb + rr * T::VP + cg * K;

    const int n_oct = MST_VARIANT(4) ? 0 : lv->n_octaves, lpo = lv->levels_per_octave;   // [ablation 4: staging + epilogue only]
    const int prot = (int)(blockIdx.x >> 3) * 2 + (int)(blockIdx.x >> 11);
    int tested = 0;
#if defined(MST_INJECT)
    Inject inj;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        inj.q[k] = (double)(tid + k);
        inj.u[k] = (uint32_t)(tid * 7 + k);
    }
#endif
    unsigned long long *tr = nullptr;
#ifdef MST_PROFILE
    // timeline sample: MST_TRACE_WGS consecutive slots of block 0, starting at 3/8 of the grid (tiles inside the band)
    if (trace && slot >= 3 * (per_xcd >> 3) && slot < 3 * (per_xcd >> 3) + MST_TRACE_WGS) {
        tr = trace + ((size_t)(slot - 3 * (per_xcd >> 3)) * T::NW + (tid >> 6)) * (MST_MAX_LEVELS + 1) * MST_TRACE_STAMPS;
        MST_STAMP(tr, MST_MAX_LEVELS * MST_TRACE_STAMPS + 1)        // end of staging
        if ((tid & 63) == 0) tr[MST_MAX_LEVELS * MST_TRACE_STAMPS + 2] = (unsigned long long)slot | ((unsigned long long)(nz_mask != 0) << 40);
    }
#endif
    for (int o = 0; o < n_oct; ++o) {
        // With octaves a factor 2 apart and s = 10, sigma_11 and sigma_12 of one octave are bit-identical to sigma_1 and
        // sigma_2 of the next (checked on the host), so G_1, G_2 and D_1 of the new octave are exactly the G_11, G_12
        // and D_11 this thread already holds: the rolling state after the previous octave's last level IS the state
        // after this octave's second level, and the two blurs are skipped.
        for (int kl = lv->first_level[o]; kl <= lpo; ++kl) {
            const int l = o * lpo + kl - 1;
            const int r = lv->radius[l];
            MST_STAMP(tr, 0)
            // the level's taps, fetched ONCE (scalar loads, wave-uniform -> SGPRs) and shared by both passes
            double taps[RMAX + 1];
#pragma unroll
            for (int j = 0; j <= RMAX; ++j) taps[j] = lv->taps[l][j];
            double g[K];
            // the leftover V-pass pieces occupy the first ceil(R/4) waves of a rotated wave order, so that over the
            // levels (and between the workgroups sharing a CU) every SIMD carries the same share of them
            const int ptid = (tid + 64 * ((l + prot) & 3)) & (T::NT - 1);
            blur_dispatch<T>(r, ct, vb, taps, ptid, vsrc, vdst, hsrc, g, variant, tr MST_INJECT_PASS);
            double d[K];
            if (kl >= 2) {
#pragma unroll
                for (int k = 0; k < K; ++k) d[k] = gprev[k] - g[k];
                if (!wave_all_in) {                          // maximum_filter pads with zeros outside the block
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        if (!((in_mask >> k) & 1u)) d[k] = 0.0;
                }
                de_mine[0] = d[0];
                de_mine[RGR] = d[K - 1];
            }
#pragma unroll
            for (int k = 0; k < K; ++k) gprev[k] = g[k];
            MST_STAMP(tr, 4)
            __syncthreads();   // edge strip visible; every H-pass read of vb is done before the next V pass writes it
            MST_STAMP(tr, 5)
#ifdef MST_PROFILE
            unsigned long long *tr_lvl = tr;
            if (tr) tr += MST_TRACE_STAMPS;
#endif
#if defined(MST_INJECT) && MST_INJECT == 2
            inject_work<>(inj);
#endif
            if (kl < 2) continue;
            if (MST_VARIANT(1) || MST_VARIANT(32)) continue;     // [ablation 1] blur + DoG only

            // zero-padded 3x3 max at the owned pixels: 3-max along the row in registers, then across rows via lane shifts.
            // A pixel's maximum is only ever compared with DoG values of the SAME pixel (mustache.py:760-765, on the tested
            // pixels), the row shifts stay inside the wave and the neighbouring column groups read this wave's edge samples
            // from the strip written above -- so a wave that owns no tested pixel has no observable use for its maxima and
            // skips them together with the sieve (its Gaussian levels, DoG values and edge samples are produced as before).
            uint32_t en = 0, gn = 0;   // en: D_new == M_new;  gn: D_new > M_prev (M of the level before it)
            double m[K];
            if (wave_has_nz) {
                const double dl = de_left[0], dr = de_right[0];
                double hm[K];
#pragma unroll
                for (int j = 0; j < K / 2; ++j) {      // max(d[2j], d[2j+1]) serves both of its pixels: 12 v_max_f64, not 16
                    const double pj = dmax(d[2 * j], d[2 * j + 1]);
                    hm[2 * j] = dmax(j == 0 ? dl : d[2 * j - 1], pj);
                    hm[2 * j + 1] = dmax(pj, 2 * j + 2 == K ? dr : d[2 * j + 2]);
                }
#pragma unroll
                for (int k = 0; k < K; ++k) m[k] = dmax(dmax(lane_prev(hm[k]), hm[k]), lane_next(hm[k]));
                // the comparisons belong to the sieve, which the reference evaluates on the tested pixels only
                // (LocMaxC[nz] == Lc[nz] ..., mustache.py:760-765)
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (d[k] == m[k]) en |= 1u << k;
                    if (d[k] > Mc[k]) gn |= 1u << k;
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) m[k] = 0.0;
            }
            if (kl >= 4) {
                // tested level = D_{kl-2}: previous = D_{kl-3} (ep), current (Dc, ec, gp), next = this one (m, en)
                double lmin = INFINITY, lsum = 0.0;
                const uint32_t code = (uint32_t)tested + 1u;
                // the reference evaluates the sieve and expon.fit on the tested pixels only (Lc[nz], mustache.py:755-768);
                // a wave that owns none has nothing to do here
                if (wave_has_nz) {
                    // branch-free form: the five terms fold into ONE floating-point comparison per pixel (D_c > best and
                    // D_c > M_n <=> D_c > max(best, M_n): no NaN reaches here) and one bit of a mask that is combined for all
                    // K pixels at once; updates and statistics are selects, so the K pixels are independent instruction streams
                    // instead of K nested exec-mask branches in a row
                    const uint32_t flags = nz_mask & ec & (ep | en) & gp;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const bool bit = (flags & (1u << k)) != 0;
                        const bool upd = bool(int(Dc[k] > dmax(best[k], m[k])) & int(bit));
                        best[k] = upd ? Dc[k] : best[k];
                        lvl[k] = upd ? code : lvl[k];
                        const bool tz = (nz_mask & (1u << k)) != 0;
                        const double a = fabs(Dc[k]);
                        lmin = __builtin_fmin(lmin, tz ? a : INFINITY);      // a >= 0, lmin >= 0: the order of the operands is immaterial
                        lsum = lsum + (tz ? a : 0.0);                          // x + 0.0 == x for every x >= +0.0: the sum's bits are unchanged
                    }
                }
                // fixed-order DPP reduction inside the wave (total lands in lane 63), one slot per (level, wave); a wave without
                // a tested pixel holds the identities {inf, 0} in every lane already
                if (wave_has_nz && !MST_VARIANT(2)) wave_reduce_min_sum(lmin, lsum);
                if (lane == 63) {
                    st[(tested * T::NW + wave) * 2] = lmin;
                    st[(tested * T::NW + wave) * 2 + 1] = lsum;
                }
                ++tested;
            }
            ep = ec;
            ec = en;
            gp = gn;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                Mc[k] = m[k];
                Dc[k] = d[k];
            }
#ifdef MST_PROFILE
            MST_STAMP(tr_lvl, 6)
#endif
        }
    }

    // ---- found pixels -> per-block record list.  One atomicAdd per WORKGROUP reserves its slots (a returning atomic
    // per pixel serialises ~150k same-address operations per block at the L2); inside the reservation the order is
    // wave, then k, then lane.  The list is sorted by pixel index before it reaches the host.
    int e_b = b, e_b2 = b2, e_x0 = x0, e_delta2 = item.delta2, e_CH = CH;      // what the epilogue needs of the work item
    if constexpr (T::RMAX > 14) {
        typedef const KernArgs __attribute__((address_space(4))) *KernArgsPtr;       // constant address space: scalar loads
        KernArgsPtr ka = (KernArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
        found = ka->found;
        found_cap = ka->found_cap;
        found_count = ka->found_count;
        n_tested = ka->n_tested;
        part = ka->partial + (size_t)slot * n_tested * 2;
        const WorkItem again = ka->items[slot];
        e_b = again.b;
        e_b2 = again.delta2 != 0 ? again.b + 1 : -1;
        e_x0 = again.x0;
        e_delta2 = again.delta2;
        e_CH = ka->CH;
    }
#if defined(MST_INJECT)
    {
        double qs = 0.0;
        uint32_t us = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            qs += inj.q[k];
            us ^= inj.u[k];
        }
        if (qs == -1.0 && us == 0x12345u) part[0] = qs;      // never true: keeps the injected chains alive
    }
#endif
    uint32_t my_total = 0;
    uint32_t before[K];                 // records of this wave that precede mine for the same k
    uint32_t kbase[K];                  // records of this wave for smaller k
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const unsigned long long bal = __ballot(lvl[k] != 0);
        before[k] = __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        kbase[k] = my_total;
        my_total += (uint32_t)__builtin_popcountll(bal);      // wave-uniform
    }
    uint32_t *cnt_lds = reinterpret_cast<uint32_t *>(de);      // edge strip is dead now
    __syncthreads();
    if (lane == 0) cnt_lds[wave] = my_total;
    __syncthreads();
    if (tid == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < T::NW; ++w) {
            const uint32_t n = cnt_lds[w];
            cnt_lds[w] = tot;
            tot += n;
        }
        cnt_lds[T::NW] = tot ? atomicAdd(found_count + e_b, tot) : 0u;
        cnt_lds[T::NW + 1] = (tot && e_b2 >= 0) ? atomicAdd(found_count + e_b2, tot) : 0u;
    }
    __syncthreads();
    const uint32_t wave_base = cnt_lds[T::NW] + cnt_lds[wave];
    const uint32_t wave_base2 = cnt_lds[T::NW + 1] + cnt_lds[wave];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        if (lvl[k]) {
            const uint32_t off = kbase[k] + before[k];
            mst_found rec;
            rec.pixel = (uint32_t)gy * (uint32_t)e_CH + (uint32_t)(e_x0 + cg * K + k);
            rec.level = lvl[k];
            rec.value = best[k];
            if (wave_base + off < found_cap) found[(size_t)e_b * found_cap + wave_base + off] = rec;
            if (e_b2 >= 0 && wave_base2 + off < found_cap) {        // the same pixel in the neighbouring block's coordinates
                rec.pixel = (uint32_t)(gy + e_delta2) * (uint32_t)e_CH + (uint32_t)(e_x0 + cg * K + k + e_delta2);
                found[(size_t)e_b2 * found_cap + wave_base2 + off] = rec;
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

// partial[item][t][2] -> level_stats[b][t][2].  A block's tiles are numbered by their position in ITS tile grid (row-major);
// slot_of_pos[b][pos] is the work item that computed that tile -- one of the block's own, or a shared one that the previous
// block's list carries -- or -1.  The summation order is fixed by the POSITION numbering (thread i takes positions i, i + 256,
// ... in ascending order, then a fixed tree), whichever items were launched: a tile that is not in the list would have
// contributed {inf, 0}, which changes neither the minimum nor the sum, so a block's statistics do not depend on
// MST_FLAG_SKIP_EMPTY down to the last bit.
__global__ void __launch_bounds__(256)
stats_reduce_kernel(const double *__restrict__ partial, int npos, const int32_t *__restrict__ slot_of_pos, int n_tested,
                    double *__restrict__ level_stats) {
    __shared__ double smin[256], ssum[256];
    const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int32_t *sop = slot_of_pos + (size_t)b * npos;
    double mn = INFINITY, sm = 0.0;
    for (int i = tid; i < npos; i += 256) {
        const int slot = sop[i];
        if (slot < 0) continue;
        const double a = partial[((size_t)slot * n_tested + t) * 2];
        mn = a < mn ? a : mn;
        sm = sm + partial[((size_t)slot * n_tested + t) * 2 + 1];
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

#ifdef MST_EXP_DEFAULT_COLS
// Experiment (round 4, variant builds): the default radii on a K = 4 tile of MST_EXP_DEFAULT_COLS columns -- 96: 768 threads =
// 12 waves = THREE per SIMD in one workgroup per CU (<= 168 VGPRs, ~115 KB of LDS); 64: 512 threads, two per SIMD
#ifndef MST_EXP_DEFAULT_K
#define MST_EXP_DEFAULT_K 4
#endif
#ifndef MST_EXP_DEFAULT_ROWS
#define MST_EXP_DEFAULT_ROWS 32
#endif
#ifndef MST_EXP_DEFAULT_TIGHT
#define MST_EXP_DEFAULT_TIGHT true
#endif
using TileDefault = Tile<MST_EXP_DEFAULT_ROWS, MST_EXP_DEFAULT_COLS, 14, MST_EXP_DEFAULT_K, 1, false, MST_EXP_DEFAULT_TIGHT>;
#else
using TileDefault = Tile<32, 64, 14>;   // the reference's default octaves (radius <= 14)
#endif
// -sz / -oc variants up to radius 28: the default tile's 32 x 64 region with 512 threads x 4 pixels and tight LDS pitches
// (131 KB: ONE workgroup of 8 waves per CU, 155 VGPRs).  Round 4 measured it against the 32 x 32 / 256-thread tile of rounds 2-3
// (two workgroups per CU, 251 VGPRs, 1.26 x more executed blur flops): octaves (3.2, 6.4) 45.6 -> 25.4 ms per 12 blocks dense,
// 18.8 -> 10.2 ms with the tile list, identical found sets (LABBOOK.md R4.2)
using TileWide = Tile<32, 64, 28, 4, 1, false, true>;
using TileDefaultFma = Tile<32, 64, 14, 8, 1, true>;   // opt-in relaxed arithmetic (MST_FLAG_FMA), default radii only
#ifdef MST_EXP_TILE7
// Experiment (round 4, variant builds only: scripts/build_variant.sh ... -DMST_EXP_TILE7=<waves per SIMD>): a tile for level
// tables whose largest blur radius is 7 (octave 1.6 alone) -- 7-pixel halo, tight pitches: 53.3 KB of LDS, three workgroups
// per CU if the register allocation allows MST_EXP_TILE7 waves per SIMD.  Prices the "two launches by octave" design.
using TileOct1 = Tile<32, 64, 7, 8, MST_EXP_TILE7, false, true>;
#endif

// A block's tile grid has at most this many rows / columns of tiles, whatever the lattice phase of its origin
template <class T>
int grid_rows(int CH) { return (CH + T::ITR - 1) / T::ITR + 1; }
template <class T>
int grid_cols(int CH) { return (CH + T::ITC - 1) / T::ITC + 1; }
template <class T>
int grid_positions(int CH) { return grid_rows<T>(CH) * grid_cols<T>(CH); }

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
int positions_max(int CH) {
    const int a = grid_positions<TileDefault>(CH), b = grid_positions<TileWide>(CH);
    return a > b ? a : b;
}

int64_t floor_div(int64_t a, int64_t b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// The work list of one launch (see WorkItem).  Tiles sit on a lattice of ITR x ITC interiors; with `share` (band source: the
// blocks' origins in chromosome coordinates are known) the lattice is anchored at chromosome coordinate 0, so overlapping
// blocks cut the same tiles, and a tile that lies inside block b AND block b + 1 with its whole halo is listed once, for
// block b, with b2 = b + 1.  Without it (dense blocks from the caller, or MST_FLAG_NO_SHARE) every block has its own lattice
// anchored at its origin -- the grid of rounds 1 and 2.
//   band_only : list only the tiles whose owned pixels can reach the tested band 4 <= col - row <= dpx + 1 (skip_empty on the
//               band source: the others would return at once)
//   slot_of_pos[b * npos + pos] = index of the item that computes the tile at grid position pos of block b, or -1
// Order: a block's items row-major; without band_only the tile ROWS are dealt to eight buckets in turn (the kernel gives each
// XCD a contiguous eighth of the list): the upper rows of a block hold the wide part of the band, whose tiles cost more than
// the constant ones, and the dispatcher hands out workgroups in order, so an XCD that got only upper rows would set the pace.
//   cuts / stage_begin (staged launches, mst_scale_space_band_stage): stage i holds the items of blocks [cuts[i - 1], cuts[i]); the
//               list is ordered stage by stage (each stage dealt on its own) and stage_begin[i] is its first item.  The list
//               itself -- which tile is computed for which blocks -- is that of the whole launch: sharing crosses the cuts.
template <class T>
void build_items(const int64_t *starts, int B, int CH, int dpx, bool share, bool band_only, std::vector<WorkItem> &items,
                 std::vector<int32_t> &slot_of_pos, int *tiles_total_out, const int32_t *cuts = nullptr, int n_cuts = 0,
                 std::vector<int32_t> *stage_begin = nullptr) {
    const int npos = grid_positions<T>(CH), gcols = grid_cols<T>(CH);
    slot_of_pos.assign((size_t)B * npos, -1);
    std::vector<WorkItem> list;
    std::vector<int> row_of;                      // running tile-row counter of each item (for the dealing)
    std::vector<int32_t> pos_of, pos2_of;
    std::vector<char> given((size_t)npos, 0), given_next((size_t)npos, 0);   // tiles of block b that block b - 1 delivers
    int rows_seen = 0, tiles_total = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t s0 = share ? starts[b] : 0;
        const int64_t a_lo = floor_div(s0, T::ITR), c_lo = floor_div(s0, T::ITC);
        // lattice cell a owns chromosome rows [a ITR, a ITR + ITR); region row 0 is the ring row above it
        const int64_t a_hi = floor_div(s0 + CH - 1, T::ITR), c_hi = floor_div(s0 + CH - 1, T::ITC);
        const bool next_ok = share && b + 1 < B && starts[b + 1] > starts[b] && starts[b + 1] - starts[b] < CH;
        const int64_t s1 = next_ok ? starts[b + 1] : 0;
        std::fill(given_next.begin(), given_next.end(), 0);
        for (int64_t a = a_lo; a <= a_hi; ++a) {
            bool row_has = false;
            for (int64_t cc = c_lo; cc <= c_hi; ++cc) {
                const int y0 = (int)(a * T::ITR - s0) - 1, x0 = (int)(cc * T::ITC - s0) - 1;
                const int pos = (int)(a - a_lo) * gcols + (int)(cc - c_lo);
                ++tiles_total;
                // owned pixels inside the block
                const int r_lo = y0 + 1 > 0 ? y0 + 1 : 0, r_hi = y0 + T::ITR < CH - 1 ? y0 + T::ITR : CH - 1;
                const int q_lo = x0 + 1 > 0 ? x0 + 1 : 0, q_hi = x0 + T::ITC < CH - 1 ? x0 + T::ITC : CH - 1;
                if (r_lo > r_hi || q_lo > q_hi) continue;                       // the cell only grazes the block with its ring
                if (band_only && !((q_hi - r_lo >= 4) && (q_lo - r_hi <= dpx + 1))) continue;
                if (given[(size_t)pos]) continue;                               // block b - 1's list computes it for both
                WorkItem it;
                it.b = b;
                it.y0 = y0;
                it.x0 = x0;
                it.delta2 = 0;
                int pos2 = -1;
                if (next_ok) {
                    // staged window (region + blur halo) inside both blocks: no reflection, no zero padding, no pixel outside
                    const int Y0 = y0 - T::RMAX, X0 = x0 - T::RMAX;
                    const int d = (int)(s0 - s1);                               // block b + 1 coordinate = block b coordinate + d
                    const bool in_b = Y0 >= 0 && X0 >= 0 && Y0 + T::CTR <= CH && X0 + T::CTC <= CH;
                    const bool in_n = Y0 + d >= 0 && X0 + d >= 0 && Y0 + d + T::CTR <= CH && X0 + d + T::CTC <= CH;
                    if (in_b && in_n) {
                        const int64_t a1_lo = floor_div(s1, T::ITR), c1_lo = floor_div(s1, T::ITC);
                        pos2 = (int)(a - a1_lo) * gcols + (int)(cc - c1_lo);
                        it.delta2 = d;
                        given_next[(size_t)pos2] = 1;
                    }
                }
                list.push_back(it);
                row_of.push_back(rows_seen);
                pos_of.push_back(pos);
                pos2_of.push_back(pos2);
                row_has = true;
            }
            if (row_has) ++rows_seen;
        }
        given.swap(given_next);
    }
    // final order + the position maps
    const size_t n = list.size();
    items.clear();
    items.reserve(n);
    std::vector<size_t> order;
    order.reserve(n);
    if (stage_begin) stage_begin->clear();
    // `list` is in block order, so a stage is a contiguous run [lo, hi) of it
    size_t lo = 0;
    for (int sgi = 0; sgi <= n_cuts; ++sgi) {
        const int b_end = sgi < n_cuts ? cuts[sgi] : B;
        size_t hi = lo;
        while (hi < n && list[hi].b < b_end) ++hi;
        if (stage_begin) stage_begin->push_back((int32_t)order.size());
        if (band_only) {
            for (size_t i = lo; i < hi; ++i) order.push_back(i);
        } else {
            size_t at[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t i = lo; i < hi; ++i) ++at[row_of[i] % 8 + 1];
            for (int x = 0; x < 8; ++x) at[x + 1] += at[x];
            const size_t base = order.size();
            order.resize(base + (hi - lo));
            for (size_t i = lo; i < hi; ++i) order[base + at[row_of[i] % 8]++] = i;
        }
        lo = hi;
    }
    if (stage_begin) stage_begin->push_back((int32_t)order.size());
    for (size_t k = 0; k < n; ++k) {
        const size_t i = order[k];
        items.push_back(list[i]);
        slot_of_pos[(size_t)list[i].b * npos + pos_of[i]] = (int32_t)k;
        if (list[i].delta2 != 0) slot_of_pos[(size_t)(list[i].b + 1) * npos + pos2_of[i]] = (int32_t)k;
    }
    if (tiles_total_out) *tiles_total_out = tiles_total;
}

// the launch's counters to zero.  A kernel, not hipMemsetAsync: the launch can be captured into a hipGraph, and a replayed memset
// node is not reliable on this ROCm (mst_tail.hip, fit_kernel; LABBOOK.md R4.6)
__global__ void zero_counts_kernel(uint32_t *found_count, uint32_t *nz_count, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        found_count[i] = 0;
        if (nz_count) nz_count[i] = 0;
    }
}

// level_stats of blocks without any tested tile: {min, sum} = {inf, 0}, what the reduction of zero tiles yields
__global__ void fill_stats_kernel(double *level_stats, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        level_stats[2 * i] = INFINITY;
        level_stats[2 * i + 1] = 0.0;
    }
}

}  // namespace

extern "C" uint64_t mst_scale_space_workspace_bytes(int32_t B, int32_t CH, const mst_levels *lv) {
    int mr = 0, nt = 0;
    if (B <= 0 || CH <= 0 || check_levels(lv, &mr, &nt) != MST_OK) return 0;
    const size_t npos = (size_t)positions_max(CH);
    return align_up(sizeof(DevLevels), 256) + align_up(sizeof(int64_t) * (size_t)B, 256) +
           align_up(sizeof(WorkItem) * (size_t)B * npos, 256) + align_up(sizeof(int32_t) * (size_t)B * npos, 256) +
           sizeof(double) * 2 * (size_t)B * npos * nt;
}

template <class T, bool BAND>
static int launch_scale_space(const double *c, const uint8_t *nz, BandSrc src, int CH, const DevLevels *d_lv,
                              mst_found *found, uint32_t found_cap, uint32_t *found_count, double *partial,
                              int n_tested, int skip_empty, const WorkItem *d_items, int n_items, hipStream_t s) {
    static unsigned long long lds_allowed = 0;      // per device (mst_common.h)
    MST_HIP(mst::allow_dynamic_lds(reinterpret_cast<const void *>(&scale_space_kernel<T, BAND>), (int)T::LDS_BYTES,
                                   &lds_allowed));
    const int gx = (n_items + 7) / 8 * 8;
    int variant = 0;
    unsigned long long *trace = nullptr;
#ifdef MST_PROFILE
    const char *venv = getenv("MST_ABLATE");          // timing ablations (results invalid), PROFILE builds only
    variant = venv ? atoi(venv) : 0;
    const char *tpath = getenv("MST_TRACE");          // phase timeline of a sample of workgroups -> file
    const size_t tbytes = sizeof(unsigned long long) * MST_TRACE_WGS * T::NW * (MST_MAX_LEVELS + 1) * MST_TRACE_STAMPS;
    if (tpath) {
        MST_HIP(hipMalloc((void **)&trace, tbytes));
        MST_HIP(hipMemsetAsync(trace, 0, tbytes, s));
    }
#endif
    scale_space_kernel<T, BAND><<<gx, T::NT, T::LDS_BYTES, s>>>(c, nz, src, CH, d_lv, found, found_cap, found_count, partial,
                                                              d_items, n_items, n_tested, skip_empty, variant, trace);
    MST_LAUNCH_CHECK();
#ifdef MST_PROFILE
    if (trace) {
        std::vector<unsigned long long> host(tbytes / sizeof(unsigned long long));
        MST_HIP(hipStreamSynchronize(s));
        MST_HIP(hipMemcpy(host.data(), trace, tbytes, hipMemcpyDeviceToHost));
        MST_HIP(hipFree(trace));
        if (FILE *f = fopen(tpath, "wb")) {
            const unsigned long long hdr[4] = {MST_TRACE_WGS, (unsigned long long)T::NW, MST_MAX_LEVELS + 1, MST_TRACE_STAMPS};
            fwrite(hdr, sizeof(hdr), 1, f);
            fwrite(host.data(), 1, tbytes, f);
            fclose(f);
        }
    }
#endif
    return MST_OK;
}

extern "C" int mst_scale_space_band_tiles(int32_t CH, int32_t dpx, const mst_levels *lv, int32_t *tiles_total_out) {
    int mr = 0, nt = 0;
    if (CH <= 0 || dpx < 0 || check_levels(lv, &mr, &nt) != MST_OK) return -1;
    const bool wide = mr > TileDefault::RMAX;
    std::vector<WorkItem> items;
    std::vector<int32_t> sop;
    const int64_t start = 0;
    int total = 0;
    if (wide) build_items<TileWide>(&start, 1, CH, dpx, false, true, items, sop, &total);
    else build_items<TileDefault>(&start, 1, CH, dpx, false, true, items, sop, &total);
    if (tiles_total_out) *tiles_total_out = total;
    return (int)items.size();
}

extern "C" int mst_scale_space_band_items(const int64_t *starts, int32_t B, int32_t CH, int32_t dpx, const mst_levels *lv,
                                          int32_t flags, int64_t *tiles_out, int64_t *shared_out) {
    int mr = 0, nt = 0;
    if (!starts || B <= 0 || CH <= 0 || dpx < 0 || check_levels(lv, &mr, &nt) != MST_OK) return -1;
    const bool wide = mr > TileDefault::RMAX, share = !(flags & MST_FLAG_NO_SHARE), band_only = (flags & MST_FLAG_SKIP_EMPTY) != 0;
    std::vector<WorkItem> items;
    std::vector<int32_t> sop;
    int total = 0;
    if (wide) build_items<TileWide>(starts, B, CH, dpx, share, band_only, items, sop, &total);
    else build_items<TileDefault>(starts, B, CH, dpx, share, band_only, items, sop, &total);
    int64_t shared = 0;
    for (const WorkItem &it : items) shared += it.delta2 != 0;
    if (tiles_out) *tiles_out = (int64_t)items.size() + shared;      // tiles the blocks would run one by one
    if (shared_out) *shared_out = shared;
    return (int)items.size();
}

// MST_FLAG_GRAPH: a launch whose every argument repeats (same buffers, same blocks, same level table) is captured into a
// hipGraph the second time it is seen and REPLAYED from then on -- one hipGraphLaunch instead of ~16 runtime calls (four
// uploads, two memsets, two kernels and their bookkeeping) in front of the fused kernel.  That matters for small launches: six
// blocks of 2000 x 2000 are 1.75 ms of kernel and the enqueue was 0.07 ms during which the GPU waited.  The graph owns a
// page-locked image of what its copy nodes read (level table, block origins, work list, position map), so nothing it references
// can change or go away under it.  Entries live per host thread; an entry's graph is destroyed only after its last launch has
// completed.
struct GraphEntry {
    std::vector<int64_t> sig;                 // every scalar / pointer argument + the block origins
    mst_levels lv;                            // the level table the graph was captured with
    hipGraphExec_t exec = nullptr;
    hipEvent_t done = nullptr;                // behind the last launch
    char *image = nullptr;                    // page-locked sources of the graph's copy nodes
    size_t image_cap = 0, image_used = 0;
    int seen = 0;
    unsigned long long stamp = 0;
    void drop_graph() {
        if (exec) {
            if (done) (void)hipEventSynchronize(done);
            (void)hipGraphExecDestroy(exec);
            exec = nullptr;
        }
    }
    ~GraphEntry() {
        drop_graph();
        if (done) (void)hipEventDestroy(done);
        if (image) (void)hipHostFree(image);
    }
};

// shared body of mst_scale_space (dense blocks) and mst_scale_space_band (blocks cut out of the band on the fly)
template <bool BAND>
static int scale_space_impl(const double *c, const uint8_t *nz, BandSrc src, const int64_t *starts_host, int32_t B,
                            int32_t CH, const mst_levels *lv, mst_found *found, uint32_t found_cap,
                            uint32_t *found_count, double *level_stats, int32_t flags, void *workspace,
                            uint64_t workspace_bytes, void *stream, const char *who, const int32_t *cuts = nullptr,
                            int32_t n_cuts = 0, int32_t stage = -1) {
    // staged form (mst_scale_space_band_stage): stage >= 0 enqueues only the items of blocks [cuts[stage - 1], cuts[stage]) of the
    // launch's ONE work list and the level statistics of those blocks; stage 0 also uploads the tables and zeroes the counters
    const bool staged = stage >= 0;
    if (staged) flags &= ~MST_FLAG_GRAPH;
    const int skip_empty = (flags & MST_FLAG_SKIP_EMPTY) ? 1 : 0;
    const bool fma = (flags & MST_FLAG_FMA) != 0;
    int mr = 0, nt = 0;
    int rc = check_levels(lv, &mr, &nt);
    if (rc != MST_OK) return rc;
    if (!found || !found_count || !level_stats || !workspace || B <= 0 || B > 65535 || CH <= 0 ||
        (int64_t)CH * CH > 0xFFFFFFFFLL)
        return mst::fail(MST_E_ARG, "%s: bad argument", who);
    const uint64_t need = mst_scale_space_workspace_bytes(B, CH, lv);
    if (workspace_bytes < need)
        return mst::fail(MST_E_ARG, "%s: workspace too small (%llu < %llu bytes)", who,
                         (unsigned long long)workspace_bytes, (unsigned long long)need);
    hipStream_t s = mst::as_stream(stream);

    // level table -> device (through a pinned staging slot, mst::upload_small: `h` may die as soon as this returns)
    DevLevels h;
    memset(&h, 0, sizeof(h));
    h.n_octaves = lv->n_octaves;
    h.levels_per_octave = lv->levels_per_octave;
    for (int l = 0; l < lv->n_octaves * lv->levels_per_octave; ++l) {
        h.radius[l] = lv->radius[l];
        for (int j = 0; j <= lv->radius[l]; ++j) h.taps[l][j] = lv->taps[l][j];
    }
    if (lv->n_octaves > 16) return mst::fail(MST_E_ARG, "%s: more than 16 octaves", who);
    const int lpo = lv->levels_per_octave;
    for (int o = 0; o < lv->n_octaves; ++o) {
        h.first_level[o] = 1;
        if (o == 0) continue;
#ifdef MST_PROFILE
        if (getenv("MST_NO_LEVEL_REUSE")) continue;       // PROFILE builds only: time the 24-blur form
#endif
        bool same = true;
        for (int q = 0; q < 2 && same; ++q) {
            const int a = (o - 1) * lpo + lpo - 2 + q, b = o * lpo + q;      // (prev octave, k = lpo-1+q) vs (this, k = 1+q)
            same = lv->radius[a] == lv->radius[b] &&
                   memcmp(lv->taps[a], lv->taps[b], sizeof(double) * (lv->radius[a] + 1)) == 0;
        }
        if (same) h.first_level[o] = 3;
    }
    if (fma && mr > TileDefault::RMAX)
        return mst::fail(MST_E_ARG, "%s: MST_FLAG_FMA is only built for blur radii <= %d", who, TileDefault::RMAX);
    const bool wide = mr > TileDefault::RMAX;
    const size_t npos_max = (size_t)positions_max(CH);
    const int npos = wide ? grid_positions<TileWide>(CH) : grid_positions<TileDefault>(CH);

    char *w = reinterpret_cast<char *>(workspace);
    DevLevels *d_lv = reinterpret_cast<DevLevels *>(w);
    w += align_up(sizeof(DevLevels), 256);
    int64_t *d_starts = reinterpret_cast<int64_t *>(w);
    w += align_up(sizeof(int64_t) * (size_t)B, 256);
    WorkItem *d_items = reinterpret_cast<WorkItem *>(w);
    w += align_up(sizeof(WorkItem) * (size_t)B * npos_max, 256);
    int32_t *d_sop = reinterpret_cast<int32_t *>(w);
    w += align_up(sizeof(int32_t) * (size_t)B * npos_max, 256);
    double *partial = reinterpret_cast<double *>(w);

    // ---- graph replay (MST_FLAG_GRAPH, band source, a stream that can be captured: not the legacy default stream)
    GraphEntry *gent = nullptr;        // non-null: this call is being CAPTURED into gent
#ifndef MST_PROFILE
    static thread_local GraphEntry gcache[4];
    static thread_local unsigned long long gstamp = 0;
    static const bool graphs_off = [] {
        const char *e = getenv("MUSTACHE_NO_GRAPHS");        // diagnostic switch: 1 = no graphs at all, "launch" = none here
        return e && *e && *e != '0' && *e != 'f';
    }();
    if (BAND && (flags & MST_FLAG_GRAPH) && s != nullptr && !graphs_off) {
        int dev = 0;
        MST_HIP(hipGetDevice(&dev));
        std::vector<int64_t> sig;
        sig.reserve((size_t)B + 16);
        for (const void *p : {(const void *)src.band, (const void *)src.band2, (const void *)found, (const void *)found_count,
                              (const void *)level_stats, (const void *)src.nz_count, (const void *)workspace})
            sig.push_back((int64_t)(intptr_t)p);
        for (int64_t v : {(int64_t)src.n, (int64_t)src.dpx, (int64_t)src.split, (int64_t)B, (int64_t)CH, (int64_t)found_cap,
                          (int64_t)flags, (int64_t)workspace_bytes, (int64_t)dev})
            sig.push_back(v);
        sig.insert(sig.end(), starts_host, starts_host + B);
        GraphEntry *ge = nullptr;
        for (GraphEntry &e : gcache)
            if (e.seen && e.sig == sig && memcmp(&e.lv, lv, sizeof(mst_levels)) == 0) ge = &e;
        mst::note("scale_space graph B=%d CH=%d n=%lld dpx=%d cap=%u flags=%d band=%p found=%p count=%p stats=%p nz=%p ws=%p -> %s", B, CH,
                  (long long)src.n, (int)src.dpx, found_cap, flags, (const void *)src.band, (const void *)found, (const void *)found_count,
                  (const void *)level_stats, (const void *)src.nz_count, workspace, ge && ge->exec ? "REPLAY" : (ge ? "CAPTURE" : "first sight"));
        if (ge && ge->exec) {
            ge->stamp = ++gstamp;
            MST_HIP(hipGraphLaunch(ge->exec, s));
            MST_HIP(hipEventRecord(ge->done, s));
            return MST_OK;
        }
        if (!ge) {                     // first sight: remember the call, run it the ordinary way (one-off launches never pay a capture)
            ge = &gcache[0];
            for (GraphEntry &e : gcache)
                if (e.stamp < ge->stamp) ge = &e;
            ge->drop_graph();
            ge->sig = sig;
            memcpy(&ge->lv, lv, sizeof(mst_levels));
            ge->seen = 1;
            ge->stamp = ++gstamp;
        } else {
            ge->stamp = ++gstamp;
            gent = ge;                 // second sight: capture below
        }
    }
#endif

    // the launch's work list: band source -> tiles on the chromosome's lattice, tiles inside two consecutive blocks computed
    // once (MST_FLAG_NO_SHARE: every block on its own lattice, every tile once per block: the cross-check form)
    // (a few recent lists are kept per host thread: a caller that runs the same blocks again -- a benchmark loop, the second
    // sample of a two-sample run -- does not pay the ~1 ms of host time per 100 k items again)
    struct Cached {
        std::vector<int64_t> key;
        std::vector<WorkItem> items;
        std::vector<int32_t> sop;
        std::vector<int32_t> stage_begin;        // staged lists: first item of every stage, then the item count
        mst::PinnedList pin_items, pin_sop;      // page-locked copies the launches upload from (no host memcpy per launch)
        // ... and a DEVICE copy, uploaded once per list: a step that repeats its launch (a benchmark loop, the second sample, the
        // same chromosome again) does not send the list -- 17 MB for the 124 blocks of chr1 at 1 kb, 0.4 ms in front of the
        // kernel -- again.  The buffer belongs to the cache entry and only grows; `used` lies behind the last launch that read it.
        char *dbuf = nullptr;
        size_t dcap = 0;
        int ddev = -1;
        bool on_device = false;
        hipEvent_t up = nullptr, used = nullptr;
        bool used_pending = false;
        void wait_used() {
            if (used_pending && used) (void)hipEventSynchronize(used);
            used_pending = false;
        }
        ~Cached() {
            wait_used();
            if (up) (void)hipEventDestroy(up);
            if (used) (void)hipEventDestroy(used);
            if (dbuf) (void)hipFree(dbuf);
        }
    };
    static thread_local Cached cache[4];
    static thread_local unsigned cache_turn = 0;
    const bool share = BAND && !(flags & MST_FLAG_NO_SHARE);
    const bool band_only = BAND && skip_empty;
    std::vector<int64_t> key;
    key.reserve((size_t)B + 6);
    key.push_back(B);
    key.push_back(CH);
    key.push_back(BAND ? src.dpx : -1);
    key.push_back((share ? 1 : 0) | (band_only ? 2 : 0) | (wide ? 4 : 0));
    if (share) key.insert(key.end(), starts_host, starts_host + B);     // without sharing the list does not depend on the origins
    if (staged) {
        key.push_back(-7);
        key.insert(key.end(), cuts, cuts + n_cuts);
    }
    Cached *hit = nullptr;
    for (Cached &cd : cache)
        if (cd.key == key) hit = &cd;
    if (!hit) {
        hit = &cache[cache_turn++ % 4];
        hit->wait_used();                   // (a launch four lists ago: long done) its device copy is about to be replaced
        hit->on_device = false;
        hit->key = key;
        std::vector<int64_t> zeros;
        const int64_t *st = starts_host;
        if (!share) {
            zeros.assign((size_t)B, 0);
            st = zeros.data();
        }
        if (wide) build_items<TileWide>(st, B, CH, BAND ? src.dpx : 0, share, band_only, hit->items, hit->sop, nullptr,
                                        staged ? cuts : nullptr, staged ? n_cuts : 0, &hit->stage_begin);
        else build_items<TileDefault>(st, B, CH, BAND ? src.dpx : 0, share, band_only, hit->items, hit->sop, nullptr,
                                      staged ? cuts : nullptr, staged ? n_cuts : 0, &hit->stage_begin);
        hipError_t pe = hit->pin_items.assign(hit->items.data(), sizeof(WorkItem) * hit->items.size());
        if (pe == hipSuccess) pe = hit->pin_sop.assign(hit->sop.data(), sizeof(int32_t) * (size_t)B * npos);
        if (pe != hipSuccess) {
            hit->key.clear();
            MST_HIP(pe);
        }
    }
    const int n_items = (int)hit->items.size();
    const size_t items_bytes = sizeof(WorkItem) * (size_t)n_items, sop_bytes = sizeof(int32_t) * (size_t)B * npos;
    // this call's share of the list and of the blocks (everything, unless staged)
    const int it0 = staged ? hit->stage_begin[stage] : 0, it1 = staged ? hit->stage_begin[stage + 1] : n_items;
    const int blk0 = staged && stage > 0 ? cuts[stage - 1] : 0, blk1 = staged && stage < n_cuts ? cuts[stage] : B;
    const bool first = !staged || stage == 0;

    if (gent) {
        // everything the graph's copy nodes will read, in page-locked memory the entry owns; then the capture begins
        const size_t need_img = align_up(sizeof(DevLevels), 256) + align_up(sizeof(int64_t) * (size_t)B, 256) +
                                align_up(items_bytes, 256) + align_up(sop_bytes, 256);
        if (gent->image_cap < need_img) {
            if (gent->image) (void)hipHostFree(gent->image);
            gent->image = nullptr;
            gent->image_cap = 0;
            MST_HIP(hipHostMalloc((void **)&gent->image, need_img, hipHostMallocDefault));
            gent->image_cap = need_img;
        }
        gent->image_used = 0;
        if (!gent->done) MST_HIP(hipEventCreateWithFlags(&gent->done, hipEventDisableTiming));
        const hipError_t ce = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        if (ce != hipSuccess) {
            (void)hipGetLastError();
            gent = nullptr;            // this stream cannot be captured: ordinary launch
        }
    }
    // host -> device copies of small tables: through the staging ring, or (capturing) from the graph entry's own image
    auto up = [&](void *dst, const void *from, size_t bytes) -> hipError_t {
        if (!gent) return mst::upload_small(dst, from, bytes, s);
        char *p = gent->image + gent->image_used;
        memcpy(p, from, bytes);
        gent->image_used += align_up(bytes, 256);
        return hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, s);
    };
    // Captured launches (small, latency-bound): the four tables lie side by side in the workspace -- the tile positions packed
    // right behind the work list instead of at their worst-case offset -- in the order and alignment of the graph entry's
    // page-locked image, so ONE copy node uploads them all (four nodes of ~5 us each stood in front of a 0.5 ms kernel).
    if (gent) d_sop = reinterpret_cast<int32_t *>(reinterpret_cast<char *>(d_items) + align_up(items_bytes, 256));
    auto enqueue = [&]() -> int {
        if (gent) {
            char *img = gent->image;
            const size_t o1 = align_up(sizeof(DevLevels), 256), o2 = o1 + align_up(sizeof(int64_t) * (size_t)B, 256),
                         o3 = o2 + align_up(items_bytes, 256);
            memcpy(img, &h, sizeof(h));
            if (BAND) memcpy(img + o1, starts_host, sizeof(int64_t) * (size_t)B);
            if (n_items) {
                memcpy(img + o2, hit->items.data(), items_bytes);
                memcpy(img + o3, hit->sop.data(), sop_bytes);
            }
            gent->image_used = n_items ? o3 + sop_bytes : o2;
            MST_HIP(hipMemcpyAsync(d_lv, img, gent->image_used, hipMemcpyHostToDevice, s));
        } else if (first) {
            MST_HIP(up(d_lv, &h, sizeof(h)));
        }
        if (first) {
            zero_counts_kernel<<<(B + 255) / 256, 256, 0, s>>>(found_count, BAND ? src.nz_count : nullptr, B);
            MST_LAUNCH_CHECK();
        }
        if (BAND) {
            if (!gent && first) MST_HIP(up(d_starts, starts_host, sizeof(int64_t) * B));
            src.starts = d_starts;
        }
        if (n_items == 0) {          // no tile reaches the band: nothing is tested, nothing is found
            if (first) {
                fill_stats_kernel<<<(B * MST_MAX_TESTED + 255) / 256, 256, 0, s>>>(level_stats, B * MST_MAX_TESTED);
                MST_LAUNCH_CHECK();
            }
            return MST_OK;
        }
        if (!gent) {
            // the list's device copy (struct Cached): uploaded by the first launch that uses the list, read in place afterwards
            int dev = 0;
            MST_HIP(hipGetDevice(&dev));
            const size_t off_sop = align_up(items_bytes, 256), need_dev = off_sop + sop_bytes;
            if (hit->on_device && hit->ddev != dev) hit->on_device = false;
            if (!hit->on_device) {
                hit->wait_used();
                if (hit->dcap < need_dev || hit->ddev != dev) {
                    if (hit->dbuf) (void)hipFree(hit->dbuf);
                    hit->dbuf = nullptr;
                    hit->dcap = 0;
                    if (hit->up) (void)hipEventDestroy(hit->up);
                    if (hit->used) (void)hipEventDestroy(hit->used);
                    hit->up = hit->used = nullptr;
                    const size_t want = need_dev + need_dev / 4;
                    MST_HIP(hipMalloc((void **)&hit->dbuf, want));
                    hit->dcap = want;
                    hit->ddev = dev;
                    MST_HIP(hipEventCreateWithFlags(&hit->up, hipEventDisableTiming));
                    MST_HIP(hipEventCreateWithFlags(&hit->used, hipEventDisableTiming));
                }
                MST_HIP(hit->pin_items.upload(hit->dbuf, s));
                MST_HIP(hit->pin_sop.upload(hit->dbuf + off_sop, s));
                MST_HIP(hipEventRecord(hit->up, s));
                hit->on_device = true;
            } else {
                MST_HIP(hipStreamWaitEvent(s, hit->up, 0));          // uploaded on another stream, perhaps
            }
            d_items = reinterpret_cast<WorkItem *>(hit->dbuf);
            d_sop = reinterpret_cast<int32_t *>(hit->dbuf + off_sop);
        }
        // (staged: this stage's run of the list; its items' slots in `partial` keep their positions in the WHOLE list, which is
        //  what the position map refers to -- both pointers are simply advanced)
        const WorkItem *li = d_items + it0;
        double *lp = partial + (size_t)it0 * nt * 2;
        const int ln = it1 - it0;
        int lrc = MST_OK;
        if (ln == 0)
            ;
        else if (fma)
            lrc = launch_scale_space<TileDefaultFma, BAND>(c, nz, src, CH, d_lv, found, found_cap, found_count, lp, nt,
                                                           skip_empty, li, ln, s);
#ifdef MST_EXP_TILE7
        else if (mr <= 7 && getenv("MST_EXP_USE_TILE7"))
            lrc = launch_scale_space<TileOct1, BAND>(c, nz, src, CH, d_lv, found, found_cap, found_count, lp, nt,
                                                     skip_empty, li, ln, s);
#endif
        else if (!wide)
            lrc = launch_scale_space<TileDefault, BAND>(c, nz, src, CH, d_lv, found, found_cap, found_count, lp, nt,
                                                        skip_empty, li, ln, s);
        else
            lrc = launch_scale_space<TileWide, BAND>(c, nz, src, CH, d_lv, found, found_cap, found_count, lp, nt,
                                                     skip_empty, li, ln, s);
        if (lrc != MST_OK) return lrc;
        // a block's tiles are its own items and shared ones of the block before it: complete once its stage has run
        if (blk1 > blk0) {
            stats_reduce_kernel<<<dim3(nt, blk1 - blk0), 256, 0, s>>>(partial, npos, d_sop + (size_t)blk0 * npos, nt,
                                                                    level_stats + (size_t)blk0 * MST_MAX_TESTED * 2);
            MST_LAUNCH_CHECK();
        }
        if (!gent) {
            MST_HIP(hipEventRecord(hit->used, s));
            hit->used_pending = true;
        }
        return MST_OK;
    };
    if (!(flags & MST_FLAG_GRAPH))
        mst::note("scale_space plain B=%d CH=%d cap=%u flags=%d found=%p count=%p stats=%p ws=%p items=%d", B, CH, found_cap, flags,
                  (const void *)found, (const void *)found_count, (const void *)level_stats, workspace, n_items);
    rc = enqueue();
    if (gent) {
        hipGraph_t graph = nullptr;
        const hipError_t ee = hipStreamEndCapture(s, &graph);        // always: the stream must leave capture mode
        if (rc != MST_OK) {
            if (graph) (void)hipGraphDestroy(graph);
            gent->seen = 0;
            return rc;
        }
        if (ee != hipSuccess || !graph) {
            gent->seen = 0;
            return mst::fail(MST_E_HIP, "%s: graph capture failed: %s", who, hipGetErrorString(ee));
        }
        hipError_t ie = hipGraphInstantiate(&gent->exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ie != hipSuccess) {
            gent->exec = nullptr;
            gent->seen = 0;
            return mst::fail(MST_E_HIP, "%s: graph instantiation failed: %s", who, hipGetErrorString(ie));
        }
        MST_HIP(hipGraphLaunch(gent->exec, s));
        MST_HIP(hipEventRecord(gent->done, s));
    }
    return rc;
}

extern "C" int mst_scale_space(const double *c, const uint8_t *nz, int32_t B, int32_t CH, const mst_levels *lv,
                               mst_found *found, uint32_t found_cap, uint32_t *found_count, double *level_stats,
                               int32_t flags, void *workspace, uint64_t workspace_bytes, void *stream) {
    MST_RANGE("launch: mst_scale_space");
    if (!c || !nz) return mst::fail(MST_E_ARG, "mst_scale_space: bad argument");
    BandSrc none = {nullptr, 0, 0, nullptr, nullptr, nullptr, INT_MAX};
    return scale_space_impl<false>(c, nz, none, nullptr, B, CH, lv, found, found_cap, found_count, level_stats, flags,
                                   workspace, workspace_bytes, stream, "mst_scale_space");
}

extern "C" int mst_scale_space_band(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t B,
                                    int32_t CH, const mst_levels *lv, mst_found *found, uint32_t found_cap,
                                    uint32_t *found_count, double *level_stats, uint32_t *nz_count, int32_t flags,
                                    void *workspace, uint64_t workspace_bytes, void *stream) {
    MST_RANGE("launch: mst_scale_space_band");
    if (!band || !starts || !nz_count || n <= 0 || dpx < 0)
        return mst::fail(MST_E_ARG, "mst_scale_space_band: bad argument");
    BandSrc src = {band, n, dpx, nullptr, nz_count, nullptr, INT_MAX};
    return scale_space_impl<true>(nullptr, nullptr, src, starts, B, CH, lv, found, found_cap, found_count, level_stats,
                                  flags, workspace, workspace_bytes, stream, "mst_scale_space_band");
}

extern "C" int mst_scale_space_band_stage(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t B,
                                          int32_t CH, const mst_levels *lv, mst_found *found, uint32_t found_cap,
                                          uint32_t *found_count, double *level_stats, uint32_t *nz_count, int32_t flags,
                                          void *workspace, uint64_t workspace_bytes, const int32_t *cuts, int32_t n_cuts,
                                          int32_t stage, void *stream) {
    MST_RANGE("launch: mst_scale_space_band_stage");
    if (!band || !starts || !nz_count || n <= 0 || dpx < 0 || n_cuts < 0 || (n_cuts > 0 && !cuts) || stage < 0 || stage > n_cuts)
        return mst::fail(MST_E_ARG, "mst_scale_space_band_stage: bad argument");
    for (int i = 0; i < n_cuts; ++i)
        if (cuts[i] <= (i ? cuts[i - 1] : 0) || cuts[i] >= B)
            return mst::fail(MST_E_ARG, "mst_scale_space_band_stage: cuts must be ascending block indices in (0, B)");
    BandSrc src = {band, n, dpx, nullptr, nz_count, nullptr, INT_MAX};
    return scale_space_impl<true>(nullptr, nullptr, src, starts, B, CH, lv, found, found_cap, found_count, level_stats,
                                  flags, workspace, workspace_bytes, stream, "mst_scale_space_band_stage", cuts, n_cuts, stage);
}

extern "C" int mst_scale_space_band_pair(const double *band1, const double *band2, int32_t split, int64_t n, int32_t dpx,
                                         const int64_t *starts, int32_t B, int32_t CH, const mst_levels *lv, mst_found *found,
                                         uint32_t found_cap, uint32_t *found_count, double *level_stats, uint32_t *nz_count,
                                         int32_t flags, void *workspace, uint64_t workspace_bytes, void *stream) {
    MST_RANGE("launch: mst_scale_space_band_pair");
    if (!band1 || !band2 || !starts || !nz_count || n <= 0 || dpx < 0 || split < 1 || split >= B)
        return mst::fail(MST_E_ARG, "mst_scale_space_band_pair: bad argument (0 < split < B)");
    if (starts[split] > starts[split - 1] && starts[split] - starts[split - 1] < CH)
        return mst::fail(MST_E_ARG, "mst_scale_space_band_pair: block %d (the second band's first) must not continue block %d "
                         "(tiles are shared between CONSECUTIVE overlapping blocks of one band only)", split, split - 1);
    BandSrc src = {band1, n, dpx, nullptr, nz_count, band2, split};
    return scale_space_impl<true>(nullptr, nullptr, src, starts, B, CH, lv, found, found_cap, found_count, level_stats,
                                  flags, workspace, workspace_bytes, stream, "mst_scale_space_band_pair");
}
