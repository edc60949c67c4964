/*
 * mustache_io.h -- C ABI of libmustache_io.so: a host-side (no GPU, no third-party module) reader for Juicer `.hic`
 * contact maps, versions 6-9, feeding the loop caller's COO input.
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
 * github.com/aidenlab/straw, `straw.cpp`: readHeader, readFooter, readMatrixZoomData, readBlock, readNormalizationVector)
 * and is exercised against files produced by tests/hic_writer.py, an independent writer of the same layout.  Parity
 * with hic-straw on a real file is UNPINNED here; mustache_amd.readers keeps a hic-straw backend for cross-checking
 * wherever that module is installed (MUSTACHE_HIC_BACKEND=hicstraw).
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

#define MST_IO_ABI_VERSION 1
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

/* Opens the file (memory-mapped, read-only) and parses header + master index.  hicstraw.HiCFile(f)  (mustache.py:308). */
int mst_hic_open(const char *path, mst_hic **out);
void mst_hic_close(mst_hic *h);

int32_t mst_hic_version(const mst_hic *h);
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

#ifdef __cplusplus
}
#endif
#endif
