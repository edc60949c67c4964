/*
 * mustache_hip.h -- C ABI of libmustache_hip.so, the MI355X (gfx950) implementation of the per-block
 * scale-space loop-calling hot path of ay-lab/mustache.
 *
 * The reference has no FFI / plugin interface (it is pure Python over SciPy); its narrowest stable seam is the
 * Python call  mustache(c, chromosome, chromosome2, res, pval_weights, start, end, mask_size, distance_in_px,
 * octave_values, st, pt)  (reference mustache/mustache.py:697-698), invoked per dense block by process_block
 * (:945-960) from regulator (:853-937).  This header is the set of entry points a binding for that seam needs;
 * each one names the reference code it replaces.  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   - every pointer marked "dev" is a device (HBM) pointer owned by the caller (e.g. a torch tensor's
 *     data_ptr()); "host" pointers are ordinary host memory.  Nothing is allocated or freed behind the ABI
 *     except small internal scratch that is cached per (device, stream).
 *   - `stream` is a hipStream_t passed as void*; NULL = the default stream.  All work is enqueued
 *     asynchronously on it; the functions do not synchronise unless stated.
 *   - return value: 0 = ok, <0 = error (MST_E_*); mst_last_error() returns a thread-local message.
 *     No C++ exception crosses the ABI.
 *   - all arithmetic is IEEE float64 without fused multiply-add, in the evaluation order of the SciPy
 *     kernels the reference calls, so DoG values are bit-identical to the reference's.
 */
#ifndef MUSTACHE_HIP_H
#define MUSTACHE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MST_ABI_VERSION 3

/* Stability of the entry points (annotations only: both expand to nothing and every symbol is exported).
 *   MST_STABLE   : the boundary SURVEY section 8(b) asks for -- scatter / prologue / blur / scale space / found records, their
 *                  finish (p-values, summary), the tail (BH, selection, features, diagonal means, clustering), the band-direct
 *                  forms of the same, the two-sample entry points.  Signature and meaning change only with MST_ABI_VERSION.
 *                  A binding written against INTEGRATION.md uses these only.
 *   MST_INTERNAL : fast paths of THIS package's own engine (no-wait / graph-replayed forms, batched "multi" forms, work-list
 *                  queries, packed-record and .hic-row scatters, the verifying scatter).  Exported because the Python host
 *                  binds them through ctypes, but they may change or disappear in any build without a version bump; every one
 *                  has a stable equivalent that returns the same results. */
#define MST_STABLE
#define MST_INTERNAL

#define MST_OK 0
#define MST_E_ARG (-1)      /* bad argument (null pointer, size, unsupported radius ...) */
#define MST_E_HIP (-2)      /* a HIP runtime call failed; message has hipGetErrorString */
#define MST_E_OVERFLOW (-3) /* an output buffer capacity was exceeded (caller re-runs with a larger one) */
#define MST_E_NONFINITE (-4)/* non-finite DoG statistics (the reference raises ValueError in expon.fit) */

/* mst_scale_space flags */
#define MST_FLAG_SKIP_EMPTY 1 /* tiles that contain no nz pixel are not computed (identical outputs, less work) */
#define MST_FLAG_NO_SHARE 4   /* band source only: compute every tile once PER BLOCK on the block's own tile lattice (the form
                               * of rounds 1 and 2).  Default: a tile that lies inside two consecutive blocks with its whole
                               * blur halo is computed once and delivered to both (identical records; see mst_scale_space_band) */
#define MST_FLAG_GRAPH 8      /* band source only: when a call repeats with EVERY argument unchanged (same device buffers, same
                               * block origins, same level table, same stream kind) it is captured into a hipGraph the second
                               * time and replayed afterwards -- one graph launch instead of ~16 runtime calls in front of the
                               * fused kernel (small launches are latency-bound).  Identical results; ignored on the legacy
                               * default stream (it cannot be captured) and by PROFILE builds. */
#define MST_FLAG_NO_WAIT 16   /* mst_found_finish only: enqueue and return -- the caller queues more work behind it (the two-sample
                               * path: pair p-values, BH, gathers), synchronises ONCE and then asks mst_found_summary_status */
#define MST_BH_RETRY 0xFFFFFFFFu /* mst_bh_select_nowait: out_count of a block whose candidate subset exceeds lds_records */
#define MST_FLAG_FMA 2        /* OPT-IN relaxed arithmetic: fuse the multiply-add of each tap pair.  DoG values then differ
                                 from the reference's by ~1e-16 relative (instead of being bit-identical); default off */

#define MST_MAX_LEVELS 64   /* octaves * (s + 2) */
#define MST_MAX_RADIUS 32
#define MST_MAX_TESTED 48   /* octaves * (s - 1) */

/* The Gaussian level table: what mustache.py:714-752 passes to scipy.ndimage.gaussian_filter, already reduced
 * by the host to integer radii and normalised taps (scipy/ndimage/_filters.py:226-236, :314-316).
 * Level index l = octave * levels_per_octave + (k - 1), k = 1 .. levels_per_octave. */
typedef struct mst_levels {
    int32_t n_octaves;
    int32_t levels_per_octave;                           /* s + 2 = 12 in the reference (s = 10, mustache.py:711) */
    int32_t radius[MST_MAX_LEVELS];
    int32_t _pad;
    double sigma[MST_MAX_LEVELS];
    double taps[MST_MAX_LEVELS][MST_MAX_RADIUS + 1];     /* taps[l][0] = centre, taps[l][j] = weight at +-j */
} mst_levels;

/* One "found" pixel = an nz pixel whose (x, y, sigma) sieve fired at least once (mustache.py:760-768). */
typedef struct mst_found {
    uint32_t pixel;     /* row * CH + col inside the block */
    uint32_t level;     /* 1 + octave * (s - 1) + (i - 3), i = the reference's loop index (mustache.py:744) */
    double value;       /* vAll: the winning DoG response */
} mst_found;

MST_STABLE int mst_abi_version(void);
MST_STABLE const char *mst_last_error(void);

/* mustache.py:919-924 (regulator's COO -> dense block scatter) for B blocks at once.
 * x, y, v: dev, upper-triangular COO in chromosome bin units; starts: host, B block origins.
 * c: dev [B][CH][CH] float64, fully overwritten (zero + scatter).  Entries with start <= x, y < start + CH go
 * into block b at (x - start, y - start), exactly like `cc[xc, yc] = vc`.  Duplicate (x, y) entries must not
 * occur (the reference takes the last one; readers never produce duplicates). */
MST_STABLE int mst_scatter_blocks(const int64_t *x, const int64_t *y, const double *v, int64_t nnz,
                       const int64_t *starts, int32_t B, int32_t CH, double *c, void *stream);

/* mustache.py:699-706 (prologue of mustache()): nz = (c != 0) & (col - row >= 4) taken before the fills,
 * then c[col-row <= 4] = 2 and, if intra != 0, c[col-row >= dpx+1] = 2.  In place on c.
 * nz: dev [B][CH][CH] uint8 (1 = tested pixel); nz_count: dev [B] uint32, overwritten with sum(nz). */
MST_STABLE int mst_block_prologue(double *c, uint8_t *nz, uint32_t *nz_count, int32_t B, int32_t CH, int32_t dpx,
                       int32_t intra, void *stream);

/* scipy.ndimage.gaussian_filter(c, sigma, truncate=t, order=0) as called at mustache.py:719/725/734/751, for a
 * batch of B images of H x W float64: axis 0 then axis 1, mode='reflect', symmetric pair-sum order.
 * taps: host, radius+1 doubles (centre first).  tmp: dev scratch, same size as `in`.  Bring-up / parity kernel
 * and the building block of the general (any radius) path. */
MST_STABLE int mst_gauss_blur(const double *in, double *out, double *tmp, int32_t B, int32_t H, int32_t W,
                   const double *taps, int32_t radius, void *stream);

/* mustache.py:714-772, the whole sigma loop of mustache() fused for B prologue'd blocks:
 * the 2 x 12 Gaussian blurs, the 11 DoG levels per octave, their zero-padded 3x3 maxima, the 5-term
 * (x, y, sigma) sieve with `best` carried across levels and octaves, and the per-level min / sum of |D| over nz
 * that scipy.stats.expon.fit needs (:755).
 *   c, nz       : dev, from mst_block_prologue
 *   lv          : host
 *   found       : dev [B][found_cap] records, unordered within a block
 *   found_count : dev [B] uint32, overwritten; a count > found_cap means records were dropped -> caller
 *                 must re-run with a larger capacity (mst_found_pvalues reports MST_E_OVERFLOW)
 *   level_stats : dev [B][MST_MAX_TESTED][2] float64, overwritten: {min |D_c| over nz, sum |D_c| over nz} per
 *                 tested level (deterministic reduction order)
 *   flags       : MST_FLAG_SKIP_EMPTY | MST_FLAG_FMA (see above); 0 = dense, exact
 *   workspace   : dev scratch of at least mst_scale_space_workspace_bytes(B, CH, lv) bytes
 */
MST_STABLE int mst_scale_space(const double *c, const uint8_t *nz, int32_t B, int32_t CH, const mst_levels *lv,
                    mst_found *found, uint32_t found_cap, uint32_t *found_count, double *level_stats,
                    int32_t flags, void *workspace, uint64_t workspace_bytes, void *stream);

/* Bytes of device scratch mst_scale_space needs for (B, CH, lv): the level table plus per-tile partial
 * statistics.  Returns 0 on bad arguments. */
MST_STABLE uint64_t mst_scale_space_workspace_bytes(int32_t B, int32_t CH, const mst_levels *lv);

/* mustache.py:755-756 for the found pixels only (deferred p-value; bit-for-bit the same quantity):
 * loc = min|D|, scale = mean|D| - loc per tested level, p = 1 - (-expm1(-(value - loc)/scale)).
 * pval: dev [B][found_cap] float64.  fit: dev [B][MST_MAX_TESTED][2] float64 out = {loc, scale}.
 * Synchronises the stream and returns MST_E_OVERFLOW / MST_E_NONFINITE when a block overflowed its record
 * capacity or produced non-finite statistics. */
MST_STABLE int mst_found_pvalues(const mst_found *found, uint32_t found_cap, const uint32_t *found_count,
                      const uint32_t *nz_count, const double *level_stats, int32_t B, int32_t n_tested,
                      double *pval, double *fit, void *stream);

/* mst_found_pvalues with everything a caller needs next delivered in the SAME round trip (one stream synchronisation per launch
 * instead of one per quantity).  summary_host (page-locked, mst_found_summary_bytes(B) bytes; scratch_dev: device memory of the
 * same size) receives {int32 flags, pad to 16 bytes | uint32 found_count[B] padded to an even count | uint32 nz_count[B]
 * likewise | double fit[B][MST_MAX_TESTED][2]} in one copy.  pack_pitch > 0: the first pack_pitch records of every block are
 * also written as three narrow, densely pitched device arrays pix_out int32 / lvl_out uint8 / pv_out float64 [B][pack_pitch]
 * (pixel index, 1-based tested level, p-value) -- what a caller that downloads whole found sets wants; with pix_host /
 * lvl_host / pv_host (page-locked, same shapes; all three or none) they are copied to the host before the synchronisation, so a
 * caller whose pack_pitch (its guess of the largest count) is confirmed by the summary needs no second round trip.
 * flags: 0, or MST_FLAG_GRAPH (a call that repeats with every argument unchanged is replayed as one hipGraph: for small,
 * latency-bound launches; it slows large pipelined ones), and / or MST_FLAG_NO_WAIT (no synchronisation, no status: the call
 * returns once everything is queued; after the caller's own synchronisation of `stream`, mst_found_summary_status(summary_host,
 * found_cap) gives the status this call would have returned).  Same error returns as mst_found_pvalues. */
MST_STABLE uint64_t mst_found_summary_bytes(int32_t B);
MST_STABLE int mst_found_summary_status(const void *summary_host, uint32_t found_cap);
MST_STABLE int mst_found_finish(const mst_found *found, uint32_t found_cap, const uint32_t *found_count, const uint32_t *nz_count,
                     const double *level_stats, int32_t B, int32_t n_tested, double *pval, double *fit, uint32_t pack_pitch,
                     int32_t *pix_out, uint8_t *lvl_out, double *pv_out, void *scratch_dev, void *summary_host,
                     int32_t *pix_host, uint8_t *lvl_host, double *pv_host, int32_t flags, void *stream);

/* mustache.py:778, multipletests(p, method='fdr_bh') per block, on the device: q[b][i] for the first count[b] records of
 * each block (sort ascending, p * m / rank with NumPy's operation order, suffix minimum, clip at 1, back to record order).
 * pval, q: dev [B][cap]; count: dev [B]; workspace: dev, >= mst_bh_workspace_bytes(B, cap). */
MST_STABLE uint64_t mst_bh_workspace_bytes(int32_t B, uint32_t cap);
MST_STABLE int mst_bh_fdr(const double *pval, const uint32_t *count, int32_t B, uint32_t cap, double *q, void *workspace,
               uint64_t workspace_bytes, void *stream);

/* BH + selection in one call, restricted to the records that can be selected (what the per-chromosome pipeline uses):
 * the records with q < threshold and their q-values, bit-identical to mst_bh_fdr + mst_select_below, but only a small
 * subset is sorted: the selected records are the BH ranks 1..j*, all below p = threshold * j* / m, and a histogram of the
 * p-values bounds j* from above (fixed point of J -> #{p < threshold * J / m}) -- typically a few hundred of ~150 000
 * records, sorted by one workgroup in LDS; the suffix minimum over the rest is >= threshold, and the global test count
 * m = found_count[b] enters every division.  A block whose subset exceeds 4096 records (threshold near 1) goes through the
 * segmented radix sort instead, same results.  Synchronises `stream` once (the subset sizes choose the route).  Outputs as
 * mst_select_below; workspace from mst_bh_workspace_bytes(B, found_cap).  (mustache.py:778-797) */
MST_STABLE int mst_bh_select(const mst_found *found, const double *pval, const uint32_t *found_count, int32_t B, uint32_t found_cap,
                  double threshold, uint32_t out_cap, uint32_t *out_pixel, uint32_t *out_level, double *out_q,
                  uint32_t *out_count, void *workspace, uint64_t workspace_bytes, void *stream);
/* The same, and also the selected records' positions in their block's found list (out_index: dev [B][out_cap] u32), so that
 * further per-record arrays (the pair p-values of the two-sample path) can be gathered for the selected records only. */
MST_STABLE int mst_bh_select_records(const mst_found *found, const double *pval, const uint32_t *found_count, int32_t B,
                          uint32_t found_cap, double threshold, uint32_t out_cap, uint32_t *out_pixel, uint32_t *out_level,
                          double *out_q, uint32_t *out_index, uint32_t *out_count, void *workspace,
                          uint64_t workspace_bytes, void *stream);

/* mst_bh_select_records (out_index may be NULL) WITHOUT the host synchronisation: the in-LDS sort is launched for every block,
 * sized for candidate subsets of up to `lds_records` records (a power of two <= 4096; 12 bytes of LDS each: the caller's guess,
 * e.g. twice what its last launch needed), and a block whose subset is larger gets out_count[b] = MST_BH_RETRY instead of a
 * count -- the caller, once it has synchronised for out_count anyway, then runs mst_bh_select / mst_bh_fdr + mst_select_below
 * for the launch.  Lets a latency-bound caller (the two-sample path on a few blocks) queue everything behind the fused
 * kernels and wait once. */
MST_INTERNAL int mst_bh_select_nowait(const mst_found *found, const double *pval, const uint32_t *found_count, int32_t B,
                         uint32_t found_cap, double threshold, uint32_t out_cap, uint32_t *out_pixel, uint32_t *out_level,
                         double *out_q, uint32_t *out_index, uint32_t *out_count, uint32_t lds_records, void *workspace,
                         uint64_t workspace_bytes, void *stream);

/* mustache.py:789-797 (selection of the pixels with o < pt) on the device: the found records of each block whose q-value
 * is below `threshold`, compacted into out_pixel / out_level / out_q [B][out_cap] (order within a block unspecified).
 * out_count: dev [B], overwritten; a count > out_cap means records were dropped -> re-run with a larger capacity.
 * Found pixels with q >= pt are never candidates and never a cluster's representative (mustache.py:843-848 takes the
 * arg-min of o, and every cluster holds a candidate), so the host tail needs nothing else. */
MST_STABLE int mst_select_below(const mst_found *found, const double *q, const uint32_t *found_count, int32_t B,
                     uint32_t found_cap, double threshold, uint32_t out_cap, uint32_t *out_pixel, uint32_t *out_level,
                     double *out_q, uint32_t *out_count, void *stream);

/* mustache.py:800-811 + :824 inputs for a list of candidate pixels of ONE block b:
 * cnt1[i] = sum nz[x-s:x+s+1, y-s:y+s+1], cnt2[i] = same with 2s (Python slice semantics: a window whose start
 * is negative is empty -> 0; windows are clipped at the far edge), cval[i] = c[x, y].
 * pixel, half: dev [n] (half = ceil(scale)); outputs dev [n]. */
MST_STABLE int mst_candidate_features(const double *c, const uint8_t *nz, int32_t CH, int32_t b, const uint32_t *pixel,
                           const int32_t *half, int32_t n, uint32_t *cnt1, uint32_t *cnt2, double *cval,
                           void *stream);

/* mustache.py:816-823: gather diagonals of block b for the diagonal-mean filter.
 * diag_k: dev [n] offsets k >= 0; out: dev [n][CH] float64, row i = c[r, r+k] for r < CH-k, zero padded. */
MST_STABLE int mst_gather_diagonals(const double *c, int32_t CH, int32_t b, const int32_t *diag_k, int32_t n, double *out,
                         void *stream);

/* mustache.py:816-824: mean_out[i] = np.mean(dg[dg != 0]) for dg = diagonal diag_k[i] of block b -- the non-zero entries
 * summed in NumPy's pairwise order (numpy/_core/src/umath/loops_utils.h.src), so the value is bit-identical to the
 * reference's and only nd doubles leave the device.  A diagonal without non-zero entries gives NaN (as np.mean does).
 * diag_k: dev [nd]; mean_out: dev [nd].  CH*8 bytes must fit the LDS (CH <= 20000). */
MST_STABLE int mst_diag_means(const double *c, int32_t CH, int32_t b, const int32_t *diag_k, int32_t nd, double *mean_out,
                   void *stream);

/* ---- diagonal-major band layout ---------------------------------------------------------------------------------
 * band[d * n + i] = value of pixel (i, i + d), d = 0 .. dpx+1, i = 0 .. n-1; 0.0 = no contact.  dev, float64. */

/* COO -> band: the per-diagonal `vals[x[indices]] = v[indices]` of mustache.py:633-635 for all diagonals at once.
 * band is zero-filled first; entries with |y - x| > dpx + 1 are ignored; x, y may come in either order. */
MST_STABLE int mst_band_from_coo(const int64_t *x, const int64_t *y, const double *v, int64_t nnz, int64_t n, int32_t dpx,
                      double *band, void *stream);

/* The same scatter from the packed records of the native `.hic` reader (include/mustache_io.h,
 * mst_hic_read_intra_packed: binX, binY - binX >= 0, float32 value; dev arrays): band[dist][x] = (double)v.  What
 * read_hic_file() -> `cc[xc, yc] = vc` amounts to (mustache.py:300-396, :921-924) without an int64 / float64 COO triple on
 * the host or across PCIe.  band is zero-filled first; records with dist > dpx + 1 are ignored. */
MST_STABLE int mst_band_from_packed(const int32_t *x, const int32_t *dist, const float *v, int64_t nnz, int64_t n, int32_t dpx,
                         double *band, void *stream);

/* Slab-wise form of the same scatter for a streaming read (the reader hands over page-locked slabs of records while later
 * `.hic` blocks are still being inflated; with one process per GPU every rank scatters the slabs of all ranks): NO clearing
 * -- the caller zero-fills `band` once -- and the distance as int32 (dist_bytes = 4) or uint16 (dist_bytes = 2: 10 bytes
 * per record across PCIe; needs dpx + 1 <= 65535).  Slabs may be scattered in any order: a matrix holds every pixel once. */
MST_INTERNAL int mst_band_scatter_packed(const int32_t *x, const void *dist, int32_t dist_bytes, const float *v, int64_t nnz, int64_t n,
                            int32_t dpx, double *band, void *stream);

/* The device half of the RAW `.hic` read (include/mustache_io.h, mst_hic_rawstream_*): the rows of inflated blocks, record
 * bytes exactly as the file stores them, decoded, normalised, filtered and scattered by one kernel -- what hic-straw's
 * readBlock() and the record loop of read_hic_file() do on the host (mustache.py:328-333, :340-389), followed by
 * `cc[xc, yc] = vc` (:919-924).  payload: dev bytes of one slab (2-byte aligned); rows: dev [n_rows] mst_hic_row
 * (include/mustache_hicrow.h; offsets relative to `payload`); norm: dev [n_norm] float64 normalisation vector or NULL (no
 * division); per record: binX <= binY ordered, dropped when binY - binX > max_dist (< 0: no limit), when either bin lies
 * outside the vector, when value = (float)(count / (norm[binX] * norm[binY])) is NaN or <= 0, or when binY >= y_limit
 * (<= 0: no limit); kept: band[(binY - binX) * n + binX] = (double)value.  NO clearing -- the caller zero-fills `band`
 * ([dpx + 2][n]) once and may scatter slabs in any order.  stats: dev uint64 [4], accumulated with atomics (the caller zeroes
 * them): [0] max binY + 1 over the kept records, [1] kept records, [2] records the band cannot hold (binY >= n or distance >
 * dpx + 1: never written), [3] verify mismatches.  verify != 0: nothing is written; every record that would be kept is read
 * back and counted in stats[3] when its pixel holds another value (two records sharing a pixel: malformed input). */
MST_INTERNAL int mst_band_scatter_hic_rows(const void *payload, const void *rows, int32_t n_rows, const double *norm, int64_t n_norm,
                              int64_t max_dist, int64_t y_limit, int64_t n, int32_t dpx, double *band, uint64_t *stats,
                              int32_t verify, void *stream);

/* Read-back check of packed scatters: *mismatches (dev uint64, zeroed by the caller) += the number of records whose pixel
 * does not hold their value afterwards -- a pixel written by two records with different values (malformed input; the
 * reference's scatter keeps the last one, mustache.py:921-924) shows up for one of them whichever store won the race. */
MST_INTERNAL int mst_band_verify_packed(const int32_t *x, const void *dist, int32_t dist_bytes, const float *v, int64_t nnz, int64_t n,
                           int32_t dpx, const double *band, uint64_t *mismatches, void *stream);

/* band -> COO order: v[e] = band[|y-x|][min(x, y)]  (the `v[indices] = vals[x[indices]]` write-back, :669). */
MST_STABLE int mst_band_to_coo(const double *band, const int64_t *x, const int64_t *y, int64_t nnz, int64_t n, int32_t dpx,
                    double *v, void *stream);

/* normalize_sparse (mustache.py:622-686) on the band, out of place (band_in != band_out).
 *   local != 0 : branch A (:628-669), taken by the caller when (n - dpx) * res > 2e6; `window` = int(2e6 / res).
 *                (local == 1: the library picks the kernel -- the walking kernel (blocks of `window` samples along each
 *                diagonal, one scan per sample) for windows up to 4096, blocked sums (16-sample blocks) up to ~8400 and
 *                (32-sample blocks, samples only in LDS) up to 16384, an error beyond.  PROFILE builds additionally accept
 *                local == 2 (blocked sums whatever the window) and local == 3 (round 1's segment kernel) as cross-checks;
 *                the product library refuses them.)
 *                Per diagonal d <= dpx+1: vals = v + 0.001; counts / sum / sum of squares over the zero-padded
 *                window [i - window/2, i - window/2 + window - 1] (np.convolve 'same'); local variance and mean
 *                with the global fallback below 30 samples or when non-finite; z = (vals - mean)/sqrt(var),
 *                non-finite -> 0, times 1 + log30(1 + global mean).
 *   local == 0 : branch B (:671-685): (v - mean)/std for d < min(dpx, n), other diagonals pass through.
 * diag_stats: dev [dpx+2][4] out = {global mean, global std, weight 1 + log30(1 + mean), entry count} per diagonal. */
MST_STABLE int mst_normalize_band(const double *band_in, double *band_out, int64_t n, int32_t dpx, int32_t window,
                       int32_t local, double *diag_stats, void *stream);

/* mustache.py:919-924 + :699-706 fused, for B blocks: dense filled blocks and the nz mask straight from the band
 * (intra-chromosomal).  starts: host [B]; c: dev [B][CH][CH]; nz: dev [B][CH][CH] uint8; nz_count: dev [B]. */
MST_STABLE int mst_blocks_from_band(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t B, int32_t CH,
                         double *c, uint8_t *nz, uint32_t *nz_count, void *stream);

/* ---- band-direct variants: the block is a window of the band, never materialised ---------------------------------------
 * Same results as mst_blocks_from_band + mst_scale_space / mst_candidate_features / mst_gather_diagonals, without the
 * [B][CH][CH] dense blocks (9 B per pixel written and read back, 18 GB for a 1 kb chr1). */

/* mustache.py:919-924 + :699-706 + :714-772 in one launch.  starts: host [B]; nz_count: dev [B] out = tested pixels per
 * block (what mst_found_pvalues needs).  Other arguments as mst_scale_space; workspace from mst_scale_space_workspace_bytes. */
MST_STABLE int mst_scale_space_band(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t B, int32_t CH,
                         const mst_levels *lv, mst_found *found, uint32_t found_cap, uint32_t *found_count,
                         double *level_stats, uint32_t *nz_count, int32_t flags, void *workspace,
                         uint64_t workspace_bytes, void *stream);
/* The same launch over blocks of TWO bands of one geometry (the two samples of diff_mustache.py:260-569, whose sigma loops
 * are independent): blocks [0, split) are windows of band1, blocks [split, B) of band2; starts: host [B] (the second band's block
 * origins usually repeat the first's; block `split` must not be a continuation of block split - 1: tiles are shared between
 * consecutive overlapping blocks of ONE band).  One launch instead of two: one set of uploads, one launch tail -- what a
 * latency-bound two-sample call on a few block pairs wants.  Everything else as mst_scale_space_band. */
MST_STABLE int mst_scale_space_band_pair(const double *band1, const double *band2, int32_t split, int64_t n, int32_t dpx,
                              const int64_t *starts, int32_t B, int32_t CH, const mst_levels *lv, mst_found *found,
                              uint32_t found_cap, uint32_t *found_count, double *level_stats, uint32_t *nz_count, int32_t flags,
                              void *workspace, uint64_t workspace_bytes, void *stream);
/* mst_scale_space_band (mustache.py:919-924 + :699-706 + :714-772) in STAGES (this package's engine; identical results): cuts[0 .. n_cuts) are ascending block indices in
 * (0, B); stage i (0 <= i <= n_cuts) enqueues the work items of blocks [cuts[i-1], cuts[i]) of the launch's ONE work list -- tile
 * sharing crosses the cuts, which separate launches over the same ranges would lose -- followed by the level statistics of
 * those blocks.  Call the stages in ascending order on one stream with identical arguments; stage 0 also uploads the tables
 * and zeroes the counters.  After stage i the outputs of the blocks below cuts[i] are final, so their mst_found_finish can
 * run on another stream (behind an event) while stage i + 1 executes.  MST_FLAG_GRAPH is ignored. */
MST_INTERNAL int mst_scale_space_band_stage(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t B, int32_t CH,
                                            const mst_levels *lv, mst_found *found, uint32_t found_cap, uint32_t *found_count,
                                            double *level_stats, uint32_t *nz_count, int32_t flags, void *workspace,
                                            uint64_t workspace_bytes, const int32_t *cuts, int32_t n_cuts, int32_t stage,
                                            void *stream);
/* How many of a block's tiles mst_scale_space_band launches with MST_FLAG_SKIP_EMPTY (those whose pixels can reach the tested
 * band 4 <= col - row <= dpx + 1), on the block's own tile lattice; *tiles_total = all tiles of the block.  Host only. */
MST_INTERNAL int mst_scale_space_band_tiles(int32_t CH, int32_t dpx, const mst_levels *lv, int32_t *tiles_total);
/* The work list mst_scale_space_band would build for these blocks and flags (host only): returns the number of workgroups it
 * launches; *tiles = the tiles the blocks would run one by one (workgroups + shared), *shared = tiles computed once for two
 * consecutive blocks.  Consecutive blocks of a chromosome overlap by half their edge (mustache.py:899-908); a tile that lies
 * inside both with its whole blur halo sees the same pixels in either, so its records and statistics are computed once and
 * delivered to both -- unless MST_FLAG_NO_SHARE is set. */
MST_INTERNAL int mst_scale_space_band_items(const int64_t *starts, int32_t B, int32_t CH, int32_t dpx, const mst_levels *lv, int32_t flags,
                               int64_t *tiles, int64_t *shared);

/* mst_candidate_features / mst_gather_diagonals / mst_diag_means for the block that starts at bin `start` of the band. */
MST_STABLE int mst_candidate_features_band(const double *band, int64_t n, int32_t dpx, int64_t start, int32_t CH,
                                const uint32_t *pixel, const int32_t *half, int32_t ncand, uint32_t *cnt1,
                                uint32_t *cnt2, double *cval, void *stream);
/* the same for candidates of several blocks in ONE launch: starts: dev [ncand], the block origin of each candidate */
MST_INTERNAL int mst_candidate_features_band_multi(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t CH,
                                      const uint32_t *pixel, const int32_t *half, int32_t ncand, uint32_t *cnt1,
                                      uint32_t *cnt2, double *cval, void *stream);
MST_STABLE int mst_gather_diagonals_band(const double *band, int64_t n, int32_t dpx, int64_t start, int32_t CH,
                              const int32_t *diag_k, int32_t nd, double *out, void *stream);
MST_STABLE int mst_diag_means_band(const double *band, int64_t n, int32_t dpx, int64_t start, int32_t CH, const int32_t *diag_k,
                        int32_t nd, double *mean_out, void *stream);
/* the same for diagonals of several blocks in ONE launch: starts: dev [nd], the block origin of each requested diagonal */
MST_INTERNAL int mst_diag_means_band_multi(const double *band, int64_t n, int32_t dpx, const int64_t *starts, int32_t CH,
                              const int32_t *diag_k, int32_t nd, double *mean_out, void *stream);

/* mustache.py:830-848 for B blocks in one launch: 8-connected clustering of every surviving candidate with its 3 x 3 halo
 * (scipy.ndimage.label numbering: components in raster order of their first pixel) and, per component, the FIRST arg-min of
 * o in raster order over its member pixels.  Inputs (dev): per block b the SELECTED records (q < pt) sorted by pixel --
 * sel_pix / sel_q [sel_off[b], sel_off[b+1]) -- and the positions of the candidates that survived the filters among them,
 * ascending -- cand_pos [cand_off[b], cand_off[b+1]).  Output: rep_pos[cand_off[b] + k] = position (in block b's records) of
 * the representative of its k-th component, rep_count[b] components.  workspace: mst_cluster_workspace_bytes(total
 * candidates).  Only selected records can be an arg-min (o >= 1 elsewhere), which is why they suffice. */
MST_STABLE uint64_t mst_cluster_workspace_bytes(uint32_t n_candidates);
MST_STABLE int mst_cluster_representatives(const uint32_t *sel_pix, const double *sel_q, const uint32_t *sel_off,
                                const uint32_t *cand_pos, const uint32_t *cand_off, int32_t B, int32_t CH,
                                uint32_t n_candidates, uint32_t *rep_pos, uint32_t *rep_count, void *workspace,
                                uint64_t workspace_bytes, void *stream);

/* ---- two-sample (differential) caller, reference mustache/diff_mustache.py:260-569 ------------------------------------
 * The per-sample sigma loops are mst_scale_space on both samples' blocks.  The entry points below add what
 * diff_mustache() computes on the difference image.  NB (reference behaviour, kept): the difference image's DoG is
 * assigned once per octave (diff_mustache.py:336) and never advanced inside the level loop, so every tested level of an
 * octave uses D_2 = G_2 - G_3 of that octave; the host produces G_2 and G_3 with mst_gauss_blur. */

/* diff_mustache.py:262-276 on already filled blocks: nzb = nz1 & nz2; cd = nzb ? c1 - c2 : 0; nzb_count[b] = sum. */
MST_STABLE int mst_diff_image(const double *c1, const double *c2, const uint8_t *nz1, const uint8_t *nz2, int32_t B, int32_t CH,
                   double *cd, uint8_t *nzb, uint32_t *nzb_count, void *stream);

/* scipy.stats.norm.fit((a - b)[mask]) per image (diff_mustache.py:371): fit[i] = {mean, sqrt(mean((x - mean)^2))},
 * i < B images of npx pixels; fixed summation order.  workspace: dev, >= 2048 * B bytes. */
MST_STABLE int mst_masked_normfit(const double *a, const double *b, const uint8_t *mask, const uint32_t *mask_count, int32_t B,
                       int64_t npx, double *fit, void *workspace, uint64_t workspace_bytes, void *stream);

/* diff_mustache.py:372-385 for the found pixels of one sample: ppair = 2 * min(cdf, 1 - cdf), cdf = ndtr((x-loc)/scale),
 * x = (g2 - g3)[octave of the record's level][pair b][pixel]; non-finite cdf -> 1.
 * found / found_count / ppair are the arrays of a 2B-block mst_scale_space launch (sample 1 = blocks [0, B), sample 2 =
 * [B, 2B)); sample_offset = 0 or B selects the sample.  g2, g3: dev [n_octaves][B][CH][CH]; fit: dev [n_octaves][B][2]. */
MST_STABLE int mst_pair_pvalues(const mst_found *found, uint32_t found_cap, const uint32_t *found_count, const double *g2,
                     const double *g3, const double *fit, int32_t B, int32_t CH, int32_t n_octaves,
                     int32_t tested_per_octave, int32_t sample_offset, double *ppair, void *stream);

/* ---- two-sample path, band-direct (what the per-chromosome driver uses; diff_mustache.py:262-276, :315-336, :371) -----------
 * For B block pairs cut out of the two samples' normalised bands (same n, dpx, starts): the difference image
 *     cd = filled_1 - filled_2 where both samples test the pixel (raw != 0, col - row >= 4), else 0
 * is formed tile by tile in LDS, blurred at sigma_2 and sigma_3 of every octave with the fused kernel's own separable passes
 * (SciPy tap order, no FMA) and only  dog[oct][b] = G_2 - G_3  (dev [n_octaves][B][CH][CH]) is written, together with
 * fit[oct][b] = {loc, scale} of norm.fit over the doubly tested pixels (dev [n_octaves][B][2]; scale from one pass,
 * sqrt(mean(x^2) - loc^2)) and mask_count[b] = their number.  No dense block, difference image or blurred level reaches HBM.
 * dog is written only in the tiles whose pixels can reach the tested band 4 <= col - row <= dpx + 1 (every found pixel lies
 * in one; the others hold no pixel of the fit either) -- the rest of the buffer is left as it was.
 * lv: the SAME level table as the sigma loop (levels 2 and 3 of each octave are used).  starts: host [B]. */
MST_STABLE uint64_t mst_diff_dog_workspace_bytes(int32_t B, int32_t CH, const mst_levels *lv);
MST_STABLE int mst_diff_dog_band(const double *band1, const double *band2, int64_t n, int32_t dpx, const int64_t *starts, int32_t B,
                      int32_t CH, const mst_levels *lv, double *dog, double *fit, uint32_t *mask_count, void *workspace,
                      uint64_t workspace_bytes, void *stream);
/* mst_pair_pvalues with x read from dog[octave][b][pixel] (the output of mst_diff_dog_band). */
MST_STABLE int mst_pair_pvalues_dog(const mst_found *found, uint32_t found_cap, const uint32_t *found_count, const double *dog,
                         const double *fit, int32_t B, int32_t CH, int32_t n_octaves, int32_t tested_per_octave,
                         int32_t sample_offset, double *ppair, void *stream);
/* The differential test's inputs for the SELECTED records only (diff_mustache.py:450-453, :567-568): for record `slot` of
 * block fb (2P blocks: sample 1 = [0, P), sample 2 = [P, 2P); sel_* = the outputs of mst_bh_select_records):
 * out_pair = ppair of the record, out_value = its winning DoG value, out_other = the partner block's (fb +- P) winning value
 * at the same pixel, NaN if the partner did not find that pixel.  out_*: dev [2P][out_cap] f64; max_selected = the largest
 * sel_count (host value: the grid's width). */
MST_STABLE int mst_pair_gather(const mst_found *found, uint32_t found_cap, const uint32_t *found_count, const double *ppair, int32_t P,
                    const uint32_t *sel_index, const uint32_t *sel_pixel, const uint32_t *sel_count, uint32_t out_cap,
                    uint32_t max_selected, double *out_pair, double *out_value, double *out_other, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MUSTACHE_HIP_H */
