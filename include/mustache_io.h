/*
 * mustache_io.h -- C ABI of libmustache_io.so: host-side (no GPU, no third-party module) readers feeding the loop caller's
 * COO input: Juicer `.hic` contact maps, versions 6-9, and the reference's text layouts (end of this file).
 *
 * It replaces what the reference does through the hic-straw Python module in read_hic_file()
 * (reference mustache/mustache.py:300-396): `hicstraw.HiCFile(f).getChromosomes()` (:308-312) and the windowed
 * `hicstraw.straw("observed", norm, f, "chr:start:end", "chr:start:end", "BP", res)` calls (:328-333) whose records the
 * reference de-duplicates, converts to bins (`// res`, :367-368) and filters to `|x - y| <= distance/res` and
 * `counts > 0` (:385-389).  The union of those windows is exactly "every record of the chromosome's intra matrix with
 * |binX - binY| <= distance/res", which is what mst_hic_read_intra returns in one pass over the zlib blocks near the
 * diagonal -- no Python record objects, no set differences.
 *
 * hic-straw is a third-party dependency that is absent from the reference tree and from the build image, and no `.hic`
 * file is available offline: the file layout below is restated from the published format (Juicer / straw,
 * github.com/aidenlab/straw, `straw.cpp`: readHeader, readFooter, readMatrixZoomData, readBlock, readNormalizationVector).
 * What pins it: the HEADER parse (magic, version, master-index offset, genome, attributes, chromosome table, resolutions) on
 * code the reference tree itself holds -- diff_mustache.py:182-249 readcstr()/read_header(), imported by
 * tests/golden/make_golden.py for version-8 files; the window / de-duplication / filter semantics on the reference's own
 * read_hic_file (tests/golden/readers_ref.npz); BLOCK decoding -- v6 plain records, v7-v9 row lists and dense grids, 16- and
 * 32-bit coordinates, short and float counts, the float32 / float64 norm-vector division, block selection by distance -- on TWO
 * independent readings of the format agreeing with this reader: the writer tests/hic_writer.py and the pure-Python reader
 * tests/hic_pyreader.py (written from the format description; no shared code), incl. hand-assembled corner cases
 * (tests/test_hic_two_readings.py).  hic-straw output on a real file remains unavailable offline; mustache_amd.readers
 * therefore still prefers hic-straw whenever that module is importable (MUSTACHE_HIC_BACKEND overrides).
 *
 * Conventions: int status (0 = ok, < 0 = MST_IO_E_*), mst_io_last_error() returns a thread-local message, no C++
 * exception crosses the ABI, arrays handed out are malloc'ed and released with mst_io_free().
 */
#ifndef MUSTACHE_IO_H
#define MUSTACHE_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MST_IO_ABI_VERSION 3
#define MST_IO_OK 0
#define MST_IO_E_ARG (-1)     /* bad argument */
#define MST_IO_E_FILE (-2)    /* cannot open / map the file */
#define MST_IO_E_FORMAT (-3)  /* not a .hic file, unsupported version, truncated or inconsistent structure */
#define MST_IO_E_NOTFOUND (-4)/* chromosome, resolution or normalisation vector not present in the file */
#define MST_IO_E_ZLIB (-5)    /* a block failed to inflate */

typedef struct mst_hic mst_hic;

int mst_io_abi_version(void);
const char *mst_io_last_error(void);
void mst_io_free(void *p);

/* The block reader's own zlib-stream decoder (csrc/mst_inflate.h: 64-bit bit buffer, 11-bit primary tables, Adler-32
 * verified) on one stream -- exported so that tests can hold it to zlib itself on arbitrary streams.  Returns the number of
 * bytes written to dst, MST_IO_E_ZLIB for an invalid stream, MST_IO_E_ARG when `capacity` (+ 266 bytes of working margin) is
 * too small.  MUSTACHE_HIC_ZLIB=1 makes the `.hic` readers inflate through zlib instead (cross-check). */
int64_t mst_io_inflate(const uint8_t *src, int64_t n, uint8_t *dst, int64_t capacity);

/* Opens the file (memory-mapped, read-only) and parses header + master index.  hicstraw.HiCFile(f)  (mustache.py:308). */
int mst_hic_open(const char *path, mst_hic **out);
void mst_hic_close(mst_hic *h);

int32_t mst_hic_version(const mst_hic *h);
/* Header fields as the reference's own (unused) header parser returns them, diff_mustache.py:201-249 read_header():
 * file offset of the master index (footer) and the genome id.  tests/golden/hic_header_v8.npz pins the header parse
 * on that function. */
int64_t mst_hic_master_offset(const mst_hic *h);
const char *mst_hic_genome(const mst_hic *h);
/* Chromosomes in file order, index 0 is usually the pseudo-chromosome "All" (the reference skips it, mustache.py:311). */
int32_t mst_hic_n_chromosomes(const mst_hic *h);
int mst_hic_chromosome(const mst_hic *h, int32_t i, const char **name, int64_t *length);
int32_t mst_hic_n_resolutions(const mst_hic *h);          /* base-pair resolutions */
int32_t mst_hic_resolution(const mst_hic *h, int32_t i);

/* Observed intra-chromosomal contacts of `chrom` (with or without a "chr" prefix, as straw matches names) at base-pair
 * resolution `resolution`, normalised by `norm` ("NONE", "KR", "VC", "VC_SQRT", "SCALE", ...):
 *     counts / (norm[binX] * norm[binY])   evaluated in double and rounded to float32, as straw does,
 * restricted to |binY - binX| <= max_dist_bins (a negative max_dist_bins keeps everything).  Records whose normalised
 * value is NaN are dropped and only values > 0 are kept (mustache.py:370-388).  Output: bin indices (binX <= binY) and
 * values as malloc'ed arrays; the return value is the number of records (>= 0) or an MST_IO_E_* code.
 * n_threads <= 0 picks the hardware concurrency.  Record order is file block order (deterministic). */
int64_t mst_hic_read_intra(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                           int32_t n_threads, int64_t **x, int64_t **y, double **v);

/* The same records in the form the GPU loader takes (mst_band_from_packed, include/mustache_hip.h): binX as int32, the
 * distance binY - binX (>= 0) as int32 and the value as the float32 straw computes -- 12 bytes per record instead of 24, no
 * int64 / float64 widening on the host.  chrom_size_bp > 0 additionally drops records with binY * resolution >= the size
 * (the end of straw's last window when the caller passes the chromosome size, mustache.py:320-333).  *n_bins receives
 * max(binY) + 1 over the returned records = the `n` of mustache.py:894.  Arrays are malloc'ed (mst_io_free). */
int64_t mst_hic_read_intra_packed(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                                  int64_t chrom_size_bp, int32_t n_threads, int32_t **x, int32_t **dist, float **v,
                                  int64_t *n_bins);

/* The two halves of mst_hic_read_intra_packed for callers that own the destination (a pinned staging buffer the GPU upload
 * reads at full PCIe rate): decode returns the record count and *n_bins and keeps the records in grow-only per-thread arenas
 * inside the handle (their capacity survives from chromosome to chromosome: no fresh pages after the largest one); fetch
 * copies them, in file block order, into x / dist / v with room for `capacity` records.  One decode at a time per handle. */
int64_t mst_hic_decode_intra_packed(mst_hic *h, const char *chrom, int32_t resolution, const char *norm,
                                    int64_t max_dist_bins, int64_t chrom_size_bp, int32_t n_threads, int64_t *n_bins);
int mst_hic_fetch_packed(mst_hic *h, int32_t *x, int32_t *dist, float *v, int64_t capacity, int32_t n_threads);
/* One process per GPU (the multi-GPU form of the per-chromosome run, reference mustache.py:913-937 forks per block and every
 * worker inherits the ONE parsed contact list): rank `part` of `n_parts` inflates and decodes only its share of the
 * chromosome's near-diagonal blocks -- a contiguous run in block index order holding 1 / n_parts of the compressed bytes --
 * so that N ranks on one host read the file ONCE between them; the ranks then exchange their packed records
 * (mustache_amd.sharding.all_gather_packed) and every rank scatters the same record set into its band.  *n_bins is this
 * part's max(binY) + 1 (the chromosome's is the maximum over the parts); *blocks_total / *blocks_mine (optional) count the
 * near-diagonal blocks of the chromosome and those this part decoded.  mst_hic_fetch_packed delivers the part's records. */
int64_t mst_hic_decode_intra_packed_part(mst_hic *h, const char *chrom, int32_t resolution, const char *norm,
                                         int64_t max_dist_bins, int64_t chrom_size_bp, int32_t n_threads, int32_t part,
                                         int32_t n_parts, int64_t *n_bins, int32_t *blocks_total, int32_t *blocks_mine);

/* ---- streaming form of the packed read ------------------------------------------------------------------------------------
 * The same records as mst_hic_decode_intra_packed_part, delivered in SLABS of caller-owned (page-locked) memory while later
 * blocks are still being inflated, so that the consumer's host-to-device copies overlap the decode: nothing is staged in the
 * handle, nothing is copied twice, the total count is not needed in advance.  `slab_memory` holds n_slabs slabs of
 * slab_records * (8 + dist_bytes) bytes each, laid out {binX int32 [slab_records], value float32 [slab_records], binY - binX
 * uint16 (dist_bytes = 2; needs 0 <= max_dist_bins <= 65535) or int32 (dist_bytes = 4) [slab_records]}; any EVEN slab_records
 * works (a block may hold more records than a slab; even, and slab_memory 4-byte aligned, so that every array is aligned).  Worker threads (n_threads <= 0: the default pool; never more than
 * n_slabs - 1) each fill one slab at a time and hand it over exactly when it is FULL -- in the middle of a block if need be --
 * and at the end of the work list: every delivered slab but each worker's last holds slab_records records, so a consumer can
 * move a full slab with one contiguous copy.
 *   mst_hic_stream_next    waits up to timeout_ms (< 0: for ever) for a filled slab: 1 = *slab / *count set (records [0, count)
 *                          of that slab), 2 = nothing ready yet, 0 = every record has been delivered, < 0 = error
 *   mst_hic_stream_release gives a slab back to the workers (after the consumer's copy out of it has completed)
 *   mst_hic_stream_close   joins the workers and frees the stream; *n_bins = max(binY) + 1 over the delivered records, *total
 *                          their number, *blocks_total / *blocks_mine as in mst_hic_decode_intra_packed_part
 * Slabs may be consumed in any order: a matrix holds every pixel once (mst_band_scatter_packed, include/mustache_hip.h).
 * The handle must stay open and must not serve another read until the stream is closed. */
typedef struct mst_hic_stream mst_hic_stream;
int mst_hic_stream_open(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                        int64_t chrom_size_bp, int32_t n_threads, int32_t part, int32_t n_parts, void *slab_memory,
                        int32_t n_slabs, int64_t slab_records, int32_t dist_bytes, mst_hic_stream **out);
int mst_hic_stream_next(mst_hic_stream *s, int32_t timeout_ms, int32_t *slab, int64_t *count);
int mst_hic_stream_release(mst_hic_stream *s, int32_t slab);
int mst_hic_stream_close(mst_hic_stream *s, int64_t *n_bins, int64_t *total, int32_t *blocks_total, int32_t *blocks_mine);

/* ---- streaming RAW read: the host only inflates, a kernel decodes the rows ------------------------------------------------
 * For versions 7-9.  Worker threads inflate the chromosome's near-diagonal blocks (share `part` of `n_parts`, the split of
 * mst_hic_decode_intra_packed_part) and copy the record bytes of every row AS THE FILE STORES THEM (6 bytes per record in the
 * usual float-count file; a decoded packed record is 10) into caller-owned (page-locked) slabs of `slab_bytes` bytes each:
 * payload from byte 0 upwards, one `mst_hic_row` entry per row (include/mustache_hicrow.h) from the slab's end downwards -- entry k at
 * slab + slab_bytes - 16 (k + 1), so the directory of a delivered slab is the `rows` entries ending at the slab's end.  No
 * host thread looks at a record: columns, counts, the division by the normalisation vector, the distance / NaN / sign /
 * chromosome-size filters and the scatter into the band are mst_band_scatter_hic_rows (include/mustache_hip.h).  A slab is
 * handed over when the next row would not fit (a block may continue in the next slab at any row) and at the end of the work
 * list.  slab_bytes: a multiple of 16 in [4096, 2^32]; slab_memory 16-byte aligned; a single row must fit a slab.
 *   mst_hic_rawstream_info     the chromosome's normalisation vector (host doubles, valid until close; *norm_values = NULL and
 *                              *norm_count = -1 for norm "NONE": the caller uploads it once for the kernel) and its length in
 *                              base pairs from the file's header (every bin lies below ceil(length / resolution))
 *   mst_hic_rawstream_next     1 = *slab delivered with *payload_bytes of records and *rows directory entries, 2 = nothing
 *                              ready within timeout_ms, 0 = everything delivered, < 0 = error
 *   mst_hic_rawstream_release  gives a slab back to the workers (after the consumer's copies out of it have completed)
 *   mst_hic_rawstream_close    joins the workers, frees the stream; totals of rows / payload bytes delivered, block counts
 * Version 6 files (plain records, no rows) are refused at open with MST_IO_E_FORMAT: use mst_hic_stream_open. */
#include "mustache_hicrow.h"
typedef struct mst_hic_rawstream mst_hic_rawstream;
int mst_hic_rawstream_open(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                           int32_t n_threads, int32_t part, int32_t n_parts, void *slab_memory, int32_t n_slabs,
                           int64_t slab_bytes, mst_hic_rawstream **out);
int mst_hic_rawstream_info(mst_hic_rawstream *s, const double **norm_values, int64_t *norm_count, int64_t *chrom_length_bp);
int mst_hic_rawstream_next(mst_hic_rawstream *s, int32_t timeout_ms, int32_t *slab, int64_t *payload_bytes, int32_t *rows);
int mst_hic_rawstream_release(mst_hic_rawstream *s, int32_t slab);
int mst_hic_rawstream_close(mst_hic_rawstream *s, int64_t *rows_total, int64_t *bytes_total, int32_t *blocks_total,
                            int32_t *blocks_mine);

/* ---- text contact maps ------------------------------------------------------------------------------------------------
 * The parse step of read_pd() (reference mustache/mustache.py:254-258): `pd.read_csv(f, sep=sep, header=None)` followed
 * by `df.dropna()`, for the two layouts the reference accepts -- 3 columns (pos1 pos2 count) or 5 columns (chr1 pos1 chr2
 * pos2 count) -- and, for 5 columns, the two `is_chr` row filters (:260-265).  pandas is a third-party dependency of the
 * reference (unpinned; 2.3.3 in the build image); its C parser's number conversion is restated here: the default
 * `float_precision` converter `precise_xstrtod` (pandas/_libs/src/parser/tokenizer.c: at most 17 significant digits are
 * accumulated, then ONE multiplication or division by an exact power of ten), the default NA strings (a row holding one
 * is dropped, as dropna() does), blank lines skipped, surrounding blanks of a field ignored.  PINNED: tests compare the
 * returned doubles bit for bit with pandas.read_csv on randomised files (tests/test_text_reader.py).
 *
 * Returns the number of rows (>= 0) with the three numeric columns as malloc'ed float64 arrays (integers below 2^53 are
 * exact, so pandas' int64-or-float64 column inference makes no difference downstream), n_cols = 3 or 5.
 * MST_IO_E_FORMAT is returned for anything this restatement does not cover (quotes, a line with more fields than the first,
 * a token that is neither a number nor an NA string): the caller then lets pandas itself read the file.
 * `chrom`: only used for 5-column files (rows are kept when both chromosome fields match it the way is_chr() does). */
int64_t mst_text_read_contacts(const char *path, char sep, const char *chrom, int32_t n_threads, int32_t *n_cols,
                               double **pos1, double **pos2, double **count);

/* The in-place fills the reference's mustache() applies to its caller's dense block (mustache.py:703-706: `c[col - row <= 4] = 2`,
 * and for chromosome == chromosome2 `c[col - row >= distance_in_px + 1] = 2`), written into a HOST block of n x n float64 whose rows
 * are `row_stride` doubles apart -- so that the drop-in mustache(c, ...) does not have to bring the filled block back over PCIe
 * (the device copy gets the same fills from mst_block_prologue; tests hold the two equal element for element). */
int mst_host_fill_block(double *c, int64_t n, int64_t row_stride, int32_t dpx, int32_t intra, int32_t n_threads);

#ifdef __cplusplus
}
#endif
#endif
