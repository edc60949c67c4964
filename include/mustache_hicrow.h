/*
 * mustache_hicrow.h -- the one structure libmustache_io.so (host: mst_hic_rawstream_*) and libmustache_hip.so (device:
 * mst_band_scatter_hic_rows) share: the directory entry of one ROW of an inflated `.hic` block whose record bytes were
 * copied verbatim into a slab.  Replaces, for `.hic` versions 7-9, the per-record work of hic-straw's readBlock() that the
 * reference reaches through `hicstraw.straw(...)` (reference mustache/mustache.py:328-333) and the per-record Python of
 * read_hic_file (:340-389: bins, distance filter, `counts > 0`): rows are decoded on the GPU.
 *
 * Row payload at slab + off, `count` records:
 *   list-of-rows block (type 1): { column int16 (or int32 with MST_HIC_ROW_INT_COLUMNS), count float32 (or int16 with
 *                                  MST_HIC_ROW_SHORT_COUNTS) } ; binX = x_off + column
 *   dense block (type 2, MST_HIC_ROW_DENSE): { count float32 | int16 } only, binX = x_off + index; NaN / -32768 = no record
 * binY = y for every record of the row.  All fields little-endian, payload 2-byte aligned.
 */
#ifndef MUSTACHE_HICROW_H
#define MUSTACHE_HICROW_H

#include <stdint.h>

typedef struct mst_hic_row {
    uint32_t off;       /* byte offset of the row's first record from the start of the slab's payload */
    int32_t y;          /* binY (absolute: binYOffset + the row's number) */
    int32_t x_off;      /* the block's binXOffset */
    uint32_t count;     /* records in the row (bits 0-27) | MST_HIC_ROW_* flags */
} mst_hic_row;

#define MST_HIC_ROW_COUNT_MASK 0x0FFFFFFFu
#define MST_HIC_ROW_SHORT_COUNTS 0x10000000u
#define MST_HIC_ROW_INT_COLUMNS 0x20000000u
#define MST_HIC_ROW_DENSE 0x40000000u

#endif
