// libmustache_io.so -- native reader for Juicer `.hic` files (host only: g++ + zlib), see include/mustache_io.h.
//
// Replaces the hic-straw calls of reference mustache/mustache.py:300-396.  The file layout is restated from the
// published format as implemented by straw (github.com/aidenlab/straw, straw.cpp); all integers little-endian:
//   header  : "HIC\0", int32 version, int64 masterIndexPosition, genomeId\0, [v9: int64 nviPosition, int64 nviLength],
//             int32 nAttributes {key\0 value\0}, int32 nChrs {name\0, v9 int64 / v8 int32 length},
//             int32 nBpResolutions {int32}, int32 nFragResolutions {int32}
//   footer  : (at masterIndexPosition) v9 int64 / v8 int32 nBytes, int32 nEntries {key\0 "c1_c2", int64 position,
//             int32 size}; expected-value vectors; normalised expected-value vectors; then the normalisation-vector
//             index: int32 nEntries {type\0, int32 chrIdx, unit\0, int32 binSize, int64 position, v9 int64 / v8 int32 size}
//             (v9 files also give its position directly as nviPosition)
//   matrix  : int32 c1, int32 c2, int32 nResolutions, per resolution: unit\0, int32 zoomIndex, 4 x float32 statistics,
//             int32 binSize, int32 blockBinCount, int32 blockColumnCount, int32 nBlocks {int32 number, int64 position,
//             int32 size}
//   block   : zlib stream; int32 nRecords, int32 binXOffset, int32 binYOffset, byte useShort(0 = yes),
//             [v9: byte useShortBinX(0 = yes), byte useShortBinY(0 = yes)], byte type;
//             type 1 = list of rows {y, count {x, value}}, type 2 = dense w-wide grid with sentinels
//             (v6: plain {int32 x, int32 y, float32 value} records)
//   norm    : v9 int64 / v8 int32 nValues, then v9 float32 / v8 float64 values
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mustache_io.h"
#include "mst_inflate.h"

namespace {

thread_local char g_err[512] = "";

int vfail(int code, const char *fmt, va_list ap) {
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    return code;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfail(code, fmt, ap);
    va_end(ap);
    return code;
}

struct FormatError {
    const char *what;
};
}  // namespace

namespace mst_io {
int fail(int code, const char *fmt, ...) {          // the error buffer is shared with text_reader.cpp
    va_list ap;
    va_start(ap, fmt);
    vfail(code, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace mst_io

namespace {

// bounds-checked little-endian cursor over a byte range
struct Cursor {
    const uint8_t *p, *end;
    Cursor(const uint8_t *b, size_t n) : p(b), end(b + n) {}
    void need(size_t n) const {
        if ((size_t)(end - p) < n) throw FormatError{"truncated structure"};
    }
    template <class T>
    T get() {
        need(sizeof(T));
        T v;
        memcpy(&v, p, sizeof(T));
        p += sizeof(T);
        return v;
    }
    std::string str() {
        const void *z = memchr(p, 0, (size_t)(end - p));
        if (!z) throw FormatError{"unterminated string"};
        std::string s((const char *)p, (const char *)z);
        p = (const uint8_t *)z + 1;
        return s;
    }
    void skip(uint64_t n) {
        if ((uint64_t)(end - p) < n) throw FormatError{"truncated structure"};
        p += n;
    }
};

struct Chrom {
    std::string name;
    int64_t length;
};
struct BlockRef {
    int32_t number;
    int64_t pos;
    int32_t size;
};
struct NormRef {
    int64_t pos, size;
};

}  // namespace

struct mst_hic {
    int fd = -1;
    const uint8_t *map = nullptr;
    size_t size = 0;
    int32_t version = 0;
    int64_t master = 0, nvi_pos = 0, nvi_len = 0;
    std::string genome;
    std::vector<Chrom> chroms;
    std::vector<int32_t> bp_res;
    std::map<std::string, std::pair<int64_t, int32_t>> matrices;     // "c1_c2" -> (position, size)
    size_t after_master = 0;                                          // file offset right behind the master index entries
    bool norm_index_read = false;
    std::map<std::string, NormRef> norm_index;                        // "TYPE|chrIdx|UNIT|binSize"
    // records of the last mst_hic_decode_intra_packed call: one grow-only arena per worker thread (capacity survives
    // from call to call, so a whole-genome run stops paying for fresh pages after its largest chromosome), and per
    // block which arena holds its records and where
    struct PackedArena {
        std::vector<int32_t> x, d;
        std::vector<float> v;
    };
    struct PackedSpan {
        int arena;
        size_t begin, count;
    };
    std::vector<PackedArena> arenas;
    std::vector<std::vector<uint8_t>> inflate_bufs;                   // per worker thread, grow-only like the arenas
    std::vector<PackedSpan> spans;
    int64_t packed_total = -1;

    Cursor at(int64_t pos) const {
        if (pos < 0 || (uint64_t)pos > size) throw FormatError{"file position outside the file"};
        return Cursor(map + pos, size - (size_t)pos);
    }
};

namespace {

void parse_header(mst_hic *h) {
    Cursor c = h->at(0);
    if (c.str() != "HIC") throw FormatError{"missing HIC magic"};
    h->version = c.get<int32_t>();
    if (h->version < 6 || h->version > 9) throw FormatError{"unsupported .hic version (6-9 are handled)"};
    h->master = c.get<int64_t>();
    h->genome = c.str();
    if (h->version > 8) {
        h->nvi_pos = c.get<int64_t>();
        h->nvi_len = c.get<int64_t>();
    }
    const int32_t n_attr = c.get<int32_t>();
    if (n_attr < 0) throw FormatError{"negative attribute count"};
    for (int32_t i = 0; i < n_attr; ++i) {
        c.str();
        c.str();
    }
    const int32_t n_chr = c.get<int32_t>();
    if (n_chr < 0 || n_chr > 1000000) throw FormatError{"implausible chromosome count"};
    for (int32_t i = 0; i < n_chr; ++i) {
        Chrom ch;
        ch.name = c.str();
        ch.length = h->version > 8 ? c.get<int64_t>() : (int64_t)c.get<int32_t>();
        h->chroms.push_back(ch);
    }
    const int32_t n_res = c.get<int32_t>();
    if (n_res < 0 || n_res > 10000) throw FormatError{"implausible resolution count"};
    for (int32_t i = 0; i < n_res; ++i) h->bp_res.push_back(c.get<int32_t>());
    // fragment resolutions and restriction sites follow; nothing on this path needs them
}

void parse_master_index(mst_hic *h) {
    Cursor c = h->at(h->master);
    if (h->version > 8) c.get<int64_t>(); else c.get<int32_t>();          // nBytes
    const int32_t n = c.get<int32_t>();
    if (n < 0) throw FormatError{"negative master index size"};
    for (int32_t i = 0; i < n; ++i) {
        std::string key = c.str();
        const int64_t pos = c.get<int64_t>();
        const int32_t size = c.get<int32_t>();
        h->matrices[key] = std::make_pair(pos, size);
    }
    h->after_master = (size_t)(c.p - h->map);
}

void skip_expected(mst_hic *h, Cursor &c, bool normalised) {
    const int32_t n = c.get<int32_t>();
    if (n < 0) throw FormatError{"negative expected-value count"};
    for (int32_t i = 0; i < n; ++i) {
        if (normalised) c.str();                                           // normalisation type
        c.str();                                                           // unit
        c.get<int32_t>();                                                  // bin size
        const int64_t nv = h->version > 8 ? c.get<int64_t>() : (int64_t)c.get<int32_t>();
        if (nv < 0) throw FormatError{"negative expected-value length"};
        c.skip((uint64_t)nv * (h->version > 8 ? 4u : 8u));
        const int32_t nf = c.get<int32_t>();
        if (nf < 0) throw FormatError{"negative normalisation-factor count"};
        c.skip((uint64_t)nf * (4u + (h->version > 8 ? 4u : 8u)));
    }
}

std::string norm_key(const std::string &type, int32_t chr_idx, const std::string &unit, int32_t bin) {
    return type + "|" + std::to_string(chr_idx) + "|" + unit + "|" + std::to_string(bin);
}

void read_norm_index(mst_hic *h) {
    if (h->norm_index_read) return;
    Cursor c = h->at(0);
    if (h->version > 8 && h->nvi_pos > 0) {
        c = h->at(h->nvi_pos);
    } else {
        c = h->at((int64_t)h->after_master);
        skip_expected(h, c, false);
        skip_expected(h, c, true);
    }
    const int32_t n = c.get<int32_t>();
    if (n < 0) throw FormatError{"negative normalisation-vector count"};
    for (int32_t i = 0; i < n; ++i) {
        std::string type = c.str();
        const int32_t chr_idx = c.get<int32_t>();
        std::string unit = c.str();
        const int32_t bin = c.get<int32_t>();
        NormRef r;
        r.pos = c.get<int64_t>();
        r.size = h->version > 8 ? c.get<int64_t>() : (int64_t)c.get<int32_t>();
        h->norm_index[norm_key(type, chr_idx, unit, bin)] = r;
    }
    h->norm_index_read = true;
}

std::vector<double> read_norm_vector(mst_hic *h, const NormRef &r) {
    Cursor c = h->at(r.pos);
    const int64_t n = h->version > 8 ? c.get<int64_t>() : (int64_t)c.get<int32_t>();
    if (n < 0 || n > (int64_t)1 << 40) throw FormatError{"implausible normalisation-vector length"};
    std::vector<double> v((size_t)n);
    if (h->version > 8) {
        for (int64_t i = 0; i < n; ++i) v[(size_t)i] = (double)c.get<float>();
    } else {
        for (int64_t i = 0; i < n; ++i) v[(size_t)i] = c.get<double>();
    }
    return v;
}

struct ZoomData {
    int32_t bin_size = 0, block_bin_count = 0, block_column_count = 0;
    std::vector<BlockRef> blocks;
    bool found = false;
};

ZoomData read_zoom(mst_hic *h, int64_t matrix_pos, int32_t resolution) {
    Cursor c = h->at(matrix_pos);
    c.get<int32_t>();                                                     // c1
    c.get<int32_t>();                                                     // c2
    const int32_t n_res = c.get<int32_t>();
    if (n_res < 0 || n_res > 10000) throw FormatError{"implausible zoom count"};
    ZoomData z;
    for (int32_t i = 0; i < n_res; ++i) {
        const std::string unit = c.str();
        c.get<int32_t>();                                                 // zoom index
        c.skip(16);                                                       // sumCounts, occupiedCellCount, stdDev, percent95
        const int32_t bin = c.get<int32_t>();
        const int32_t bbc = c.get<int32_t>();
        const int32_t bcc = c.get<int32_t>();
        const int32_t nb = c.get<int32_t>();
        if (nb < 0) throw FormatError{"negative block count"};
        if (unit == "BP" && bin == resolution) {
            z.bin_size = bin;
            z.block_bin_count = bbc;
            z.block_column_count = bcc;
            z.blocks.resize((size_t)nb);
            for (int32_t b = 0; b < nb; ++b) {
                z.blocks[(size_t)b].number = c.get<int32_t>();
                z.blocks[(size_t)b].pos = c.get<int64_t>();
                z.blocks[(size_t)b].size = c.get<int32_t>();
            }
            z.found = true;
            return z;
        }
        c.skip((uint64_t)nb * 16u);
    }
    return z;
}

// Can block `number` hold a record with |binX - binY| <= max_dist?  Errs on the side of reading the block.
bool block_near_diagonal(int32_t version, int32_t number, int32_t bbc, int32_t bcc, int64_t max_dist) {
    if (max_dist < 0 || bbc <= 0 || bcc <= 0) return true;
    if (version > 8) {
        // v9 intra-chromosomal blocks are indexed (depth, position along the diagonal):
        //   depth = int(log2(1 + |binX - binY| / sqrt(2) / blockBinCount)),  number = depth * blockColumnCount + pad
        const int64_t depth = number / bcc;
        const int64_t far = (int64_t)std::log2(1.0 + (double)max_dist / std::sqrt(2.0) / (double)bbc) + 1;
        return depth <= far;
    }
    const int64_t r = number / bcc, col = number % bcc;                    // number = row * blockColumnCount + column
    const int64_t d = r > col ? r - col : col - r;
    // bins of block row r / column col differ by at least (d - 1) * blockBinCount + 1 when d >= 1 -- exact, so the blocks just
    // beyond the distance limit (a few per cent of a 1 kb chromosome's compressed bytes) are not inflated for nothing
    return d == 0 || (d - 1) * (int64_t)bbc + 1 <= max_dist;
}

struct Records {
    std::vector<int64_t> x, y;
    std::vector<double> v;
};

// packed sink: binX, binY - binX, float32 value appended to a worker's arena; y_limit = first bin past the caller's size
struct PackedSink {
    mst_hic::PackedArena *a;
    int64_t y_limit, ymax;
    void push(int64_t bx, int64_t by, float c) {
        if (by >= y_limit) return;
        a->x.push_back((int32_t)bx);
        a->d.push_back((int32_t)(by - bx));
        a->v.push_back(c);
        ymax = by > ymax ? by : ymax;
    }
    void reserve(size_t) {}
};

inline void emit(PackedSink &out, int64_t bx, int64_t by, float counts, const std::vector<double> *norm, int64_t max_dist) {
    if (bx > by) {
        const int64_t t = bx;
        bx = by;
        by = t;
    }
    if (max_dist >= 0 && by - bx > max_dist) return;
    float c = counts;
    if (norm) {
        if (bx < 0 || (size_t)by >= norm->size()) return;
        c = (float)((double)counts / ((*norm)[(size_t)bx] * (*norm)[(size_t)by]));
    }
    if (std::isnan(c) || !(c > 0.0f)) return;
    out.push(bx, by, c);
}

inline void emit(Records &out, int64_t bx, int64_t by, float counts, const std::vector<double> *norm, int64_t max_dist) {
    if (bx > by) {                                                         // intra blocks store binX <= binY; be lenient
        const int64_t t = bx;
        bx = by;
        by = t;
    }
    if (max_dist >= 0 && by - bx > max_dist) return;
    float c = counts;
    if (norm) {
        if (bx < 0 || (size_t)by >= norm->size()) return;                  // straw would index out of range; drop
        c = (float)((double)counts / ((*norm)[(size_t)bx] * (*norm)[(size_t)by]));
    }
    if (std::isnan(c) || !(c > 0.0f)) return;                              // mustache.py:370-373, :385-388
    out.x.push_back(bx);
    out.y.push_back(by);
    out.v.push_back((double)c);
}

inline void sink_reserve(Records &out, size_t room) {
    out.x.reserve(room);
    out.y.reserve(room);
    out.v.reserve(room);
}
inline void sink_reserve(PackedSink &, size_t) {}

// MUSTACHE_HIC_ZLIB=1: inflate through zlib instead of mst_inflate.h (the cross-check the tests use)
bool use_zlib() {
    static const bool z = [] {
        const char *e = getenv("MUSTACHE_HIC_ZLIB");
        return e && *e && *e != '0';
    }();
    return z;
}

// one zlib stream -> buf (grown until it fits: the uncompressed size is not stored); returns the byte count.
// `readable_past`: bytes known to be readable behind comp + comp_size (the decoder prefetches up to mst_inflate::kSlack).
size_t inflate_block(const uint8_t *comp, size_t comp_size, size_t readable_past, std::vector<uint8_t> &buf,
                     std::vector<uint8_t> &pad) {
    if (buf.size() < comp_size * 4 + 4096) buf.resize(comp_size * 4 + 4096);
    if (!use_zlib()) {
        const uint8_t *src = comp;
        if (readable_past < mst_inflate::kSlack) {          // the last block(s) of the file: decode from a padded copy
            pad.assign(comp_size + mst_inflate::kSlack, 0);
            memcpy(pad.data(), comp, comp_size);
            src = pad.data();
        }
        for (;;) {
            size_t n_out = 0;
            const int rc = mst_inflate::inflate_zlib(src, comp_size, buf.data(), buf.size() - mst_inflate::kSlack, &n_out);
            if (rc == mst_inflate::kOk) return n_out;
            if (rc == mst_inflate::kOutputFull && buf.size() < ((size_t)1 << 31)) {
                buf.resize(buf.size() * 2);
                continue;
            }
            throw FormatError{"zlib inflate failed"};
        }
    }
    size_t n_out = 0;
    for (;;) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (inflateInit(&zs) != Z_OK) throw FormatError{"zlib init failed"};
        zs.next_in = const_cast<Bytef *>(comp);
        zs.avail_in = (uInt)comp_size;
        zs.next_out = buf.data();
        zs.avail_out = (uInt)buf.size();
        const int rc = inflate(&zs, Z_FINISH);
        n_out = zs.total_out;
        inflateEnd(&zs);
        if (rc == Z_STREAM_END) break;
        if ((rc == Z_BUF_ERROR || rc == Z_OK) && buf.size() < ((size_t)1 << 31)) {
            buf.resize(buf.size() * 2);
            continue;
        }
        throw FormatError{"zlib inflate failed"};
    }
    return n_out;
}

template <class Sink>
void decode_records(int32_t version, const uint8_t *data, size_t n_out, Sink &out, const std::vector<double> *norm,
                    int64_t max_dist);

template <class Sink>
void decode_block(int32_t version, const uint8_t *comp, size_t comp_size, size_t readable_past, std::vector<uint8_t> &buf,
                  Sink &out, const std::vector<double> *norm, int64_t max_dist) {
    static thread_local std::vector<uint8_t> pad;
    const size_t n_out = inflate_block(comp, comp_size, readable_past, buf, pad);
    decode_records(version, buf.data(), n_out, out, norm, max_dist);
}

template <class Sink>
void decode_records(int32_t version, const uint8_t *data, size_t n_out, Sink &out, const std::vector<double> *norm,
                    int64_t max_dist) {
    Cursor c(data, n_out);
    const int32_t n_rec = c.get<int32_t>();
    if (n_rec < 0) throw FormatError{"negative record count in a block"};
    const size_t room = (size_t)n_rec < n_out ? (size_t)n_rec : n_out;      // a record takes at least one byte
    sink_reserve(out, room);
    if (version < 7) {
        for (int32_t i = 0; i < n_rec; ++i) {
            const int32_t bx = c.get<int32_t>();
            const int32_t by = c.get<int32_t>();
            const float v = c.get<float>();
            emit(out, bx, by, v, norm, max_dist);
        }
        return;
    }
    const int32_t x_off = c.get<int32_t>();
    const int32_t y_off = c.get<int32_t>();
    const bool short_counts = c.get<uint8_t>() == 0;                      // 0 means "yes" in this format
    bool short_x = true, short_y = true;
    if (version > 8) {
        short_x = c.get<uint8_t>() == 0;
        short_y = c.get<uint8_t>() == 0;
    }
    const uint8_t type = c.get<uint8_t>();
    if (type == 1) {
        const int32_t rows = short_y ? (int32_t)c.get<int16_t>() : c.get<int32_t>();
        for (int32_t r = 0; r < rows; ++r) {
            const int32_t y = short_y ? (int32_t)c.get<int16_t>() : c.get<int32_t>();
            const int32_t cols = short_x ? (int32_t)c.get<int16_t>() : c.get<int32_t>();
            for (int32_t j = 0; j < cols; ++j) {
                const int32_t x = short_x ? (int32_t)c.get<int16_t>() : c.get<int32_t>();
                const float v = short_counts ? (float)c.get<int16_t>() : c.get<float>();
                emit(out, (int64_t)x_off + x, (int64_t)y_off + y, v, norm, max_dist);
            }
        }
    } else if (type == 2) {
        const int32_t n_pts = c.get<int32_t>();
        const int32_t w = (int32_t)c.get<int16_t>();
        if (w <= 0) throw FormatError{"dense block of width 0"};
        for (int32_t i = 0; i < n_pts; ++i) {
            const int32_t row = i / w, col = i - row * w;
            if (short_counts) {
                const int16_t s = c.get<int16_t>();
                if (s != -32768) emit(out, (int64_t)x_off + col, (int64_t)y_off + row, (float)s, norm, max_dist);
            } else {
                const float v = c.get<float>();
                if (!std::isnan(v)) emit(out, (int64_t)x_off + col, (int64_t)y_off + row, v, norm, max_dist);
            }
        }
    } else {
        throw FormatError{"unknown block type"};
    }
}

// ---- streaming read: records land in caller-owned slabs {x int32 [cap], v float32 [cap], dist uint16 | int32 [cap]} ----------
struct SlabSink {
    int32_t *x;
    float *v;
    void *d;
    int dist_bytes;
    int64_t count, cap, y_limit, ymax;
    void (*on_full)(SlabSink &, void *) = nullptr;        // streamed read: hand the full slab over and continue in a fresh one
    void *ctx = nullptr;
    void push(int64_t bx, int64_t by, float c) {
        if (by >= y_limit) return;
        if (count >= cap) {
            if (!on_full) throw FormatError{"a block holds more records than its header says"};
            on_full(*this, ctx);
        }
        x[count] = (int32_t)bx;
        v[count] = c;
        if (dist_bytes == 2) ((uint16_t *)d)[count] = (uint16_t)(by - bx);
        else ((int32_t *)d)[count] = (int32_t)(by - bx);
        ++count;
        ymax = by > ymax ? by : ymax;
    }
};

inline void emit(SlabSink &out, int64_t bx, int64_t by, float counts, const std::vector<double> *norm, int64_t max_dist) {
    if (bx > by) {
        const int64_t t = bx;
        bx = by;
        by = t;
    }
    if (max_dist >= 0 && by - bx > max_dist) return;
    float c = counts;
    if (norm) {
        if (bx < 0 || (size_t)by >= norm->size()) return;
        c = (float)((double)counts / ((*norm)[(size_t)bx] * (*norm)[(size_t)by]));
    }
    if (std::isnan(c) || !(c > 0.0f)) return;
    out.push(bx, by, c);
}
inline void sink_reserve(SlabSink &, size_t) {}

// The layout almost every block of a real file has -- version 7-9, list of rows -- decoded with raw pointers: the row's byte
// extent is checked once, the records go through the same emit() as the general decoder (same filters, same arithmetic).
// Returns false when the block is of another kind (the caller then runs decode_records).
template <bool SHORT_X, bool SHORT_C>
void decode_rows(const uint8_t *p, const uint8_t *end, int32_t rows, bool short_y, int32_t x_off, int32_t y_off, SlabSink &out,
                 const std::vector<double> *norm, int64_t max_dist) {
    constexpr size_t rec = (SHORT_X ? 2 : 4) + (SHORT_C ? 2 : 4);
    for (int32_t r = 0; r < rows; ++r) {
        const size_t head = (short_y ? 2 : 4) + (SHORT_X ? 2 : 4);
        if ((size_t)(end - p) < head) throw FormatError{"truncated structure"};
        int32_t y, cols;
        if (short_y) {
            int16_t t;
            memcpy(&t, p, 2);
            y = t;
            p += 2;
        } else {
            memcpy(&y, p, 4);
            p += 4;
        }
        if (SHORT_X) {
            int16_t t;
            memcpy(&t, p, 2);
            cols = t;
            p += 2;
        } else {
            memcpy(&cols, p, 4);
            p += 4;
        }
        if (cols < 0) cols = 0;                                   // the general decoder's loop runs zero times as well
        if ((size_t)(end - p) < (size_t)cols * rec) throw FormatError{"truncated structure"};
        const int64_t by = (int64_t)y_off + y;
        for (int32_t j = 0; j < cols; ++j) {
            int32_t x;
            float val;
            if (SHORT_X) {
                int16_t t;
                memcpy(&t, p, 2);
                x = t;
                p += 2;
            } else {
                memcpy(&x, p, 4);
                p += 4;
            }
            if (SHORT_C) {
                int16_t t;
                memcpy(&t, p, 2);
                val = (float)t;
                p += 2;
            } else {
                memcpy(&val, p, 4);
                p += 4;
            }
            emit(out, (int64_t)x_off + x, by, val, norm, max_dist);
        }
    }
}

bool decode_rows_fast(int32_t version, const uint8_t *data, size_t n, SlabSink &out, const std::vector<double> *norm,
                      int64_t max_dist) {
    if (version < 7) return false;
    const size_t hdr = 4 + 4 + 4 + 1 + (version > 8 ? 2 : 0) + 1;
    if (n < hdr) return false;
    const uint8_t *p = data + 4;
    int32_t x_off, y_off;
    memcpy(&x_off, p, 4);
    memcpy(&y_off, p + 4, 4);
    p += 8;
    const bool short_c = *p++ == 0;
    bool short_x = true, short_y = true;
    if (version > 8) {
        short_x = *p++ == 0;
        short_y = *p++ == 0;
    }
    if (*p++ != 1) return false;
    const uint8_t *end = data + n;
    int32_t rows;
    if (short_y) {
        if (end - p < 2) return false;
        int16_t t;
        memcpy(&t, p, 2);
        rows = t;
        p += 2;
    } else {
        if (end - p < 4) return false;
        memcpy(&rows, p, 4);
        p += 4;
    }
    if (short_x && short_c) decode_rows<true, true>(p, end, rows, short_y, x_off, y_off, out, norm, max_dist);
    else if (short_x) decode_rows<true, false>(p, end, rows, short_y, x_off, y_off, out, norm, max_dist);
    else if (short_c) decode_rows<false, true>(p, end, rows, short_y, x_off, y_off, out, norm, max_dist);
    else decode_rows<false, false>(p, end, rows, short_y, x_off, y_off, out, norm, max_dist);
    return true;
}

int find_chromosome(const mst_hic *h, const char *name) {
    const std::string want(name);
    const std::string bare = want.rfind("chr", 0) == 0 ? want.substr(3) : want;
    for (int pass = 0; pass < 2; ++pass)
        for (size_t i = 0; i < h->chroms.size(); ++i) {
            const std::string &n = h->chroms[i].name;
            if (pass == 0 ? n == want : (n == bare || n == "chr" + bare)) return (int)i;
        }
    return -1;
}

}  // namespace

extern "C" int mst_io_abi_version(void) { return MST_IO_ABI_VERSION; }
extern "C" const char *mst_io_last_error(void) { return g_err; }
extern "C" void mst_io_free(void *p) { free(p); }

extern "C" int64_t mst_io_inflate(const uint8_t *src, int64_t n, uint8_t *dst, int64_t capacity) {
    if (!src || !dst || n < 0 || capacity < 0) return fail(MST_IO_E_ARG, "mst_io_inflate: bad argument");
    std::vector<uint8_t> in((size_t)n + mst_inflate::kSlack, 0), out((size_t)capacity + mst_inflate::kSlack);
    memcpy(in.data(), src, (size_t)n);
    size_t produced = 0;
    const int rc = mst_inflate::inflate_zlib(in.data(), (size_t)n, out.data(), (size_t)capacity, &produced);
    if (rc == mst_inflate::kOutputFull) return fail(MST_IO_E_ARG, "mst_io_inflate: output does not fit %lld bytes", (long long)capacity);
    if (rc != mst_inflate::kOk) return fail(MST_IO_E_ZLIB, "mst_io_inflate: not a valid zlib stream");
    memcpy(dst, out.data(), produced);
    return (int64_t)produced;
}

extern "C" int mst_hic_open(const char *path, mst_hic **out) {
    if (!path || !out) return fail(MST_IO_E_ARG, "mst_hic_open: null argument");
    *out = nullptr;
    mst_hic *h = new (std::nothrow) mst_hic();
    if (!h) return fail(MST_IO_E_FILE, "out of memory");
    h->fd = open(path, O_RDONLY);
    struct stat st;
    if (h->fd < 0 || fstat(h->fd, &st) != 0 || st.st_size <= 0) {
        mst_hic_close(h);
        return fail(MST_IO_E_FILE, "cannot open %s", path);
    }
    h->size = (size_t)st.st_size;
    void *m = mmap(nullptr, h->size, PROT_READ, MAP_PRIVATE, h->fd, 0);
    if (m == MAP_FAILED) {
        mst_hic_close(h);
        return fail(MST_IO_E_FILE, "cannot map %s", path);
    }
    h->map = (const uint8_t *)m;
    try {
        parse_header(h);
        parse_master_index(h);
    } catch (const FormatError &e) {
        mst_hic_close(h);
        return fail(MST_IO_E_FORMAT, "%s: %s", path, e.what);
    } catch (...) {
        mst_hic_close(h);
        return fail(MST_IO_E_FORMAT, "%s: unreadable", path);
    }
    *out = h;
    return MST_IO_OK;
}

extern "C" void mst_hic_close(mst_hic *h) {
    if (!h) return;
    if (h->map) munmap((void *)h->map, h->size);
    if (h->fd >= 0) close(h->fd);
    delete h;
}

extern "C" int32_t mst_hic_version(const mst_hic *h) { return h ? h->version : 0; }
extern "C" int64_t mst_hic_master_offset(const mst_hic *h) { return h ? h->master : 0; }
extern "C" const char *mst_hic_genome(const mst_hic *h) { return h ? h->genome.c_str() : ""; }
extern "C" int32_t mst_hic_n_chromosomes(const mst_hic *h) { return h ? (int32_t)h->chroms.size() : 0; }
extern "C" int mst_hic_chromosome(const mst_hic *h, int32_t i, const char **name, int64_t *length) {
    if (!h || i < 0 || (size_t)i >= h->chroms.size()) return fail(MST_IO_E_ARG, "mst_hic_chromosome: bad index");
    if (name) *name = h->chroms[(size_t)i].name.c_str();
    if (length) *length = h->chroms[(size_t)i].length;
    return MST_IO_OK;
}
extern "C" int32_t mst_hic_n_resolutions(const mst_hic *h) { return h ? (int32_t)h->bp_res.size() : 0; }
extern "C" int32_t mst_hic_resolution(const mst_hic *h, int32_t i) {
    return (h && i >= 0 && (size_t)i < h->bp_res.size()) ? h->bp_res[(size_t)i] : 0;
}

// Worker threads when the caller passes n_threads <= 0: the hardware concurrency, capped at twice the container's CPU
// quota when there is one (cgroup v2 cpu.max / v1 cpu.cfs_quota_us): inflate is compute bound, and 256 threads on a 16-CPU
// quota run slower than 32 (measured on the GPU box: 0.20 s against 0.12 s for 125 M records).
static int default_threads() {
    int hw = (int)std::thread::hardware_concurrency();
    if (hw < 1) hw = 1;
    double quota = 0.0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = {0};
        long long period = 0;
        if (fscanf(f, "%63s %lld", a, &period) == 2 && strcmp(a, "max") != 0 && period > 0) quota = atof(a) / (double)period;
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        long long q = -1, period = 0;
        if (fscanf(g, "%lld", &q) != 1) q = -1;
        fclose(g);
        if (FILE *p = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(p, "%lld", &period) != 1) period = 0;
            fclose(p);
        }
        if (q > 0 && period > 0) quota = (double)q / (double)period;
    }
    if (quota > 0.0) {
        const int cap = (int)(2.0 * quota + 0.5);
        if (cap >= 1 && cap < hw) hw = cap;
    }
    return hw;
}

// Shared body of the two record readers: every near-diagonal block of the chromosome's intra matrix inflated and decoded on
// the worker threads, one Records per block in file block order.  Returns 0 or an MST_IO_E_* code (message set).
static int read_intra_parts(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                            int32_t n_threads, std::vector<Records> &part, int *threads_used) {
    const int ci = find_chromosome(h, chrom);
    if (ci < 0) return fail(MST_IO_E_NOTFOUND, "chromosome %s is not in the file", chrom);
    const std::string key = std::to_string(ci) + "_" + std::to_string(ci);
    auto it = h->matrices.find(key);
    if (it == h->matrices.end()) return fail(MST_IO_E_NOTFOUND, "no intra-chromosomal matrix for %s", chrom);
    ZoomData z = read_zoom(h, it->second.first, resolution);
    if (!z.found) return fail(MST_IO_E_NOTFOUND, "resolution %d is not in the file", resolution);

    std::vector<double> norm_vec;
    const bool use_norm = norm && *norm && strcmp(norm, "NONE") != 0;
    if (use_norm) {
        read_norm_index(h);
        auto nit = h->norm_index.find(norm_key(norm, ci, "BP", resolution));
        if (nit == h->norm_index.end())
            return fail(MST_IO_E_NOTFOUND, "no %s normalisation vector for %s at %d bp", norm, chrom, resolution);
        norm_vec = read_norm_vector(h, nit->second);
    }

    std::vector<const BlockRef *> todo;
    for (const BlockRef &b : z.blocks) {
        if (b.size <= 0) continue;
        if (b.pos < 0 || (uint64_t)b.pos + (uint64_t)b.size > h->size) throw FormatError{"block outside the file"};
        if (block_near_diagonal(h->version, b.number, z.block_bin_count, z.block_column_count, max_dist_bins))
            todo.push_back(&b);
    }
    int nt = n_threads > 0 ? n_threads : default_threads();
    if (nt < 1) nt = 1;
    if ((size_t)nt > todo.size()) nt = todo.empty() ? 1 : (int)todo.size();
    part.assign(todo.size(), Records());
    std::atomic<size_t> next(0);
    std::atomic<int> bad(0);
    const char *bad_what = nullptr;
    auto work = [&]() {
        std::vector<uint8_t> buf;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= todo.size() || bad.load()) return;
            try {
                decode_block(h->version, h->map + todo[i]->pos, (size_t)todo[i]->size,
                             h->size - (size_t)todo[i]->pos - (size_t)todo[i]->size, buf, part[i],
                             use_norm ? &norm_vec : nullptr, max_dist_bins);
            } catch (const FormatError &e) {
                bad_what = e.what;
                bad.store(1);
                return;
            } catch (...) {
                bad_what = "out of memory";
                bad.store(1);
                return;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    if (bad.load()) return fail(MST_IO_E_ZLIB, "block decode failed: %s", bad_what ? bad_what : "?");
    *threads_used = nt;
    return MST_IO_OK;
}

// run `fn(i)` for i in [0, n) on nt threads (dynamic dealing)
template <class F>
static void parallel_for(size_t n, int nt, F fn) {
    std::atomic<size_t> next(0);
    auto body = [&]() {
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n) return;
            fn(i);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(body);
    body();
    for (auto &t : pool) t.join();
}

extern "C" int64_t mst_hic_read_intra(mst_hic *h, const char *chrom, int32_t resolution, const char *norm,
                                      int64_t max_dist_bins, int32_t n_threads, int64_t **x, int64_t **y, double **v) {
    if (!h || !chrom || !x || !y || !v || resolution <= 0) return fail(MST_IO_E_ARG, "mst_hic_read_intra: bad argument");
    *x = *y = nullptr;
    *v = nullptr;
    try {
        std::vector<Records> part;
        int nt = 1;
        const int rc = read_intra_parts(h, chrom, resolution, norm, max_dist_bins, n_threads, part, &nt);
        if (rc != MST_IO_OK) return rc;
        std::vector<size_t> offs(part.size() + 1, 0);
        for (size_t i = 0; i < part.size(); ++i) offs[i + 1] = offs[i] + part[i].v.size();
        const size_t total = offs[part.size()];
        int64_t *ox = (int64_t *)malloc((total ? total : 1) * sizeof(int64_t));
        int64_t *oy = (int64_t *)malloc((total ? total : 1) * sizeof(int64_t));
        double *ov = (double *)malloc((total ? total : 1) * sizeof(double));
        if (!ox || !oy || !ov) {
            free(ox);
            free(oy);
            free(ov);
            return fail(MST_IO_E_FILE, "out of memory for %zu records", total);
        }
        // concatenate in file block order (deterministic), the copies spread over the same worker threads
        parallel_for(part.size(), nt, [&](size_t i) {
            Records &r = part[i];
            if (r.v.empty()) return;
            memcpy(ox + offs[i], r.x.data(), r.v.size() * sizeof(int64_t));
            memcpy(oy + offs[i], r.y.data(), r.v.size() * sizeof(int64_t));
            memcpy(ov + offs[i], r.v.data(), r.v.size() * sizeof(double));
            std::vector<int64_t>().swap(r.x);                      // release the block's buffers as we go
            std::vector<int64_t>().swap(r.y);
            std::vector<double>().swap(r.v);
        });
        *x = ox;
        *y = oy;
        *v = ov;
        return (int64_t)total;
    } catch (const FormatError &e) {
        return fail(MST_IO_E_FORMAT, "%s", e.what);
    } catch (...) {
        return fail(MST_IO_E_FORMAT, "unreadable file (out of memory?)");
    }
}

// block list + normalisation vector of one chromosome's intra matrix (shared by the record readers)
static int intra_todo(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                      std::vector<const BlockRef *> &todo, ZoomData &z, std::vector<double> &norm_vec, bool *use_norm) {
    const int ci = find_chromosome(h, chrom);
    if (ci < 0) return fail(MST_IO_E_NOTFOUND, "chromosome %s is not in the file", chrom);
    const std::string key = std::to_string(ci) + "_" + std::to_string(ci);
    auto it = h->matrices.find(key);
    if (it == h->matrices.end()) return fail(MST_IO_E_NOTFOUND, "no intra-chromosomal matrix for %s", chrom);
    z = read_zoom(h, it->second.first, resolution);
    if (!z.found) return fail(MST_IO_E_NOTFOUND, "resolution %d is not in the file", resolution);
    *use_norm = norm && *norm && strcmp(norm, "NONE") != 0;
    if (*use_norm) {
        read_norm_index(h);
        auto nit = h->norm_index.find(norm_key(norm, ci, "BP", resolution));
        if (nit == h->norm_index.end())
            return fail(MST_IO_E_NOTFOUND, "no %s normalisation vector for %s at %d bp", norm, chrom, resolution);
        norm_vec = read_norm_vector(h, nit->second);
    }
    for (const BlockRef &b : z.blocks) {
        if (b.size <= 0) continue;
        if (b.pos < 0 || (uint64_t)b.pos + (uint64_t)b.size > h->size) throw FormatError{"block outside the file"};
        if (block_near_diagonal(h->version, b.number, z.block_bin_count, z.block_column_count, max_dist_bins))
            todo.push_back(&b);
    }
    return MST_IO_OK;
}

// Part `part` of `n_parts` of a chromosome's block list (one process per GPU, each rank inflates its share): a contiguous run
// of the near-diagonal blocks in file index order, cut so that the parts hold equal shares of the COMPRESSED bytes (block i
// belongs to the part its byte midpoint falls into) -- every block belongs to exactly one part, whatever n_parts is.  The ONE
// definition every reader uses: ranks that disagreed on it would drop or double-decode blocks without any error.
static void split_todo(std::vector<const BlockRef *> &todo, int32_t part, int32_t n_parts) {
    if (n_parts <= 1) return;
    double total = 0.0, run = 0.0;
    for (const BlockRef *b : todo) total += (double)b->size;
    std::vector<const BlockRef *> own;
    for (const BlockRef *b : todo) {
        const double mid = run + 0.5 * (double)b->size;
        run += (double)b->size;
        int p = total > 0.0 ? (int)(mid / total * (double)n_parts) : 0;
        p = p < 0 ? 0 : (p >= n_parts ? n_parts - 1 : p);
        if (p == part) own.push_back(b);
    }
    todo.swap(own);
}

extern "C" int64_t mst_hic_decode_intra_packed(mst_hic *h, const char *chrom, int32_t resolution, const char *norm,
                                               int64_t max_dist_bins, int64_t chrom_size_bp, int32_t n_threads,
                                               int64_t *n_bins) {
    return mst_hic_decode_intra_packed_part(h, chrom, resolution, norm, max_dist_bins, chrom_size_bp, n_threads, 0, 1, n_bins,
                                            nullptr, nullptr);
}

extern "C" int64_t mst_hic_decode_intra_packed_part(mst_hic *h, const char *chrom, int32_t resolution, const char *norm,
                                                    int64_t max_dist_bins, int64_t chrom_size_bp, int32_t n_threads,
                                                    int32_t part, int32_t n_parts, int64_t *n_bins, int32_t *blocks_total,
                                                    int32_t *blocks_mine) {
    if (!h || !chrom || !n_bins || resolution <= 0 || n_parts < 1 || part < 0 || part >= n_parts)
        return fail(MST_IO_E_ARG, "mst_hic_decode_intra_packed: bad argument");
    *n_bins = 0;
    h->packed_total = -1;
    try {
        std::vector<const BlockRef *> todo;
        ZoomData z;
        std::vector<double> norm_vec;
        bool use_norm = false;
        const int rc = intra_todo(h, chrom, resolution, norm, max_dist_bins, todo, z, norm_vec, &use_norm);
        if (rc != MST_IO_OK) return rc;
        if (blocks_total) *blocks_total = (int32_t)todo.size();
        split_todo(todo, part, n_parts);
        if (blocks_mine) *blocks_mine = (int32_t)todo.size();
        int nt = n_threads > 0 ? n_threads : default_threads();
        if (nt < 1) nt = 1;
        if ((size_t)nt > todo.size()) nt = todo.empty() ? 1 : (int)todo.size();
        if (h->arenas.size() < (size_t)nt) h->arenas.resize((size_t)nt);
        if (h->inflate_bufs.size() < (size_t)nt) h->inflate_bufs.resize((size_t)nt);
        for (auto &a : h->arenas) {
            a.x.clear();
            a.d.clear();
            a.v.clear();
        }
        h->spans.assign(todo.size(), mst_hic::PackedSpan{0, 0, 0});
        // straw's window end (mustache.py:320-333): no position at or past the chromosome size the caller gave
        const int64_t y_limit = chrom_size_bp > 0 ? (chrom_size_bp + resolution - 1) / resolution : INT64_MAX;
        std::atomic<size_t> next(0);
        std::atomic<int> bad(0);
        const char *bad_what = nullptr;
        std::vector<int64_t> ymax((size_t)nt, -1);
        auto work = [&](int t) {
            std::vector<uint8_t> &buf = h->inflate_bufs[(size_t)t];
            PackedSink sink{&h->arenas[(size_t)t], y_limit, -1};
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= todo.size() || bad.load()) break;
                const size_t before = sink.a->v.size();
                try {
                    decode_block(h->version, h->map + todo[i]->pos, (size_t)todo[i]->size,
                                 h->size - (size_t)todo[i]->pos - (size_t)todo[i]->size, buf, sink,
                                 use_norm ? &norm_vec : nullptr, max_dist_bins);
                } catch (const FormatError &e) {
                    bad_what = e.what;
                    bad.store(1);
                    break;
                } catch (...) {
                    bad_what = "out of memory";
                    bad.store(1);
                    break;
                }
                h->spans[i] = mst_hic::PackedSpan{t, before, sink.a->v.size() - before};
            }
            ymax[(size_t)t] = sink.ymax;
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto &t : pool) t.join();
        if (bad.load()) return fail(MST_IO_E_ZLIB, "block decode failed: %s", bad_what ? bad_what : "?");
        int64_t total = 0, top = -1;
        for (const auto &sp : h->spans) total += (int64_t)sp.count;
        for (int64_t m : ymax) top = m > top ? m : top;
        if (top >= INT32_MAX) return fail(MST_IO_E_FORMAT, "bin index %lld does not fit 32 bits", (long long)top);
        h->packed_total = total;
        *n_bins = top + 1;
        return total;
    } catch (const FormatError &e) {
        return fail(MST_IO_E_FORMAT, "%s", e.what);
    } catch (...) {
        return fail(MST_IO_E_FORMAT, "unreadable file (out of memory?)");
    }
}

extern "C" int mst_hic_fetch_packed(mst_hic *h, int32_t *x, int32_t *dist, float *v, int64_t capacity, int32_t n_threads) {
    if (!h || h->packed_total < 0) return fail(MST_IO_E_ARG, "mst_hic_fetch_packed: no decoded records (call mst_hic_decode_intra_packed)");
    if (capacity < h->packed_total || (h->packed_total > 0 && (!x || !dist || !v)))
        return fail(MST_IO_E_ARG, "mst_hic_fetch_packed: room for %lld records, %lld decoded", (long long)capacity,
                    (long long)h->packed_total);
    std::vector<size_t> offs(h->spans.size() + 1, 0);
    for (size_t i = 0; i < h->spans.size(); ++i) offs[i + 1] = offs[i] + h->spans[i].count;
    int nt = n_threads > 0 ? n_threads : default_threads();
    if (nt < 1) nt = 1;
    if ((size_t)nt > h->spans.size()) nt = h->spans.empty() ? 1 : (int)h->spans.size();
    // file block order (deterministic whatever thread decoded a block), the copies spread over the worker threads
    parallel_for(h->spans.size(), nt, [&](size_t i) {
        const mst_hic::PackedSpan &sp = h->spans[i];
        if (!sp.count) return;
        const mst_hic::PackedArena &a = h->arenas[(size_t)sp.arena];
        memcpy(x + offs[i], a.x.data() + sp.begin, sp.count * sizeof(int32_t));
        memcpy(dist + offs[i], a.d.data() + sp.begin, sp.count * sizeof(int32_t));
        memcpy(v + offs[i], a.v.data() + sp.begin, sp.count * sizeof(float));
    });
    return MST_IO_OK;
}

extern "C" int64_t mst_hic_read_intra_packed(mst_hic *h, const char *chrom, int32_t resolution, const char *norm,
                                             int64_t max_dist_bins, int64_t chrom_size_bp, int32_t n_threads,
                                             int32_t **x, int32_t **dist, float **v, int64_t *n_bins) {
    if (!x || !dist || !v) return fail(MST_IO_E_ARG, "mst_hic_read_intra_packed: bad argument");
    *x = *dist = nullptr;
    *v = nullptr;
    const int64_t total = mst_hic_decode_intra_packed(h, chrom, resolution, norm, max_dist_bins, chrom_size_bp, n_threads,
                                                      n_bins);
    if (total < 0) return total;
    int32_t *ox = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    int32_t *od = (int32_t *)malloc((size_t)(total ? total : 1) * sizeof(int32_t));
    float *ov = (float *)malloc((size_t)(total ? total : 1) * sizeof(float));
    if (!ox || !od || !ov) {
        free(ox);
        free(od);
        free(ov);
        return fail(MST_IO_E_FILE, "out of memory for %lld records", (long long)total);
    }
    const int rc = mst_hic_fetch_packed(h, ox, od, ov, total, n_threads);
    if (rc != MST_IO_OK) {
        free(ox);
        free(od);
        free(ov);
        return rc;
    }
    *x = ox;
    *dist = od;
    *v = ov;
    return total;
}

// ---- streaming packed read -------------------------------------------------------------------------------------------------
// Worker threads inflate and decode the chromosome's near-diagonal blocks (this part's share of them) straight into
// caller-owned slabs; a slab is handed to the consumer as soon as the next block would not fit, so the consumer's H2D copies
// run while later blocks are still being inflated.  No arena, no second copy, no total count needed in advance.
#include <condition_variable>
#include <deque>
#include <mutex>

struct mst_hic_stream {
    mst_hic *h = nullptr;
    std::vector<const BlockRef *> todo;
    ZoomData zoom;
    std::vector<double> norm_vec;
    bool use_norm = false;
    int64_t max_dist = -1, y_limit = INT64_MAX;
    uint8_t *base = nullptr;
    int32_t n_slabs = 0, dist_bytes = 2;
    int64_t cap = 0;
    std::vector<int64_t> counts;
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    std::deque<int32_t> free_q, ready_q;
    std::vector<std::thread> workers;
    std::atomic<size_t> next{0};
    int active = 0;
    bool failed = false, cancelled = false;
    std::string error;
    int64_t ymax = -1, total = 0;
    int32_t blocks_total = 0;

    uint8_t *slab(int32_t i) const { return base + (size_t)i * (size_t)cap * (size_t)(8 + dist_bytes); }

    void fail_with(const char *what) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) error = what;
        failed = true;
        cv_free.notify_all();
        cv_ready.notify_all();
    }

    struct Cancelled {};

    // One worker: blocks are decoded straight into the slab it holds; a slab is handed over exactly when it is FULL (in the
    // middle of a block if need be: a block may hold more records than a slab) and at the end of the work list, so the
    // consumer sees a steady flow of full slabs however many workers there are.
    struct Worker {
        mst_hic_stream *s;
        int32_t cur = -1;
        SlabSink sink;
        void publish() {
            if (cur < 0) return;
            std::lock_guard<std::mutex> lk(s->mu);
            s->counts[(size_t)cur] = sink.count;
            s->total += sink.count;
            if (sink.count > 0) {
                s->ready_q.push_back(cur);
                s->cv_ready.notify_one();
            } else {
                s->free_q.push_back(cur);
                s->cv_free.notify_one();
            }
            cur = -1;
        }
        void next_slab() {
            publish();
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_free.wait(lk, [&] { return !s->free_q.empty() || s->failed || s->cancelled; });
            if (s->failed || s->cancelled) throw Cancelled{};
            cur = s->free_q.front();
            s->free_q.pop_front();
            lk.unlock();
            uint8_t *m = s->slab(cur);
            sink.x = reinterpret_cast<int32_t *>(m);
            sink.v = reinterpret_cast<float *>(m + (size_t)s->cap * 4);
            sink.d = m + (size_t)s->cap * 8;
            sink.count = 0;
            sink.cap = s->cap;
        }
        static void full(SlabSink &, void *self) { static_cast<Worker *>(self)->next_slab(); }
    };

    void work() {
        std::vector<uint8_t> buf, pad;
        Worker w{this, -1, SlabSink{nullptr, nullptr, nullptr, dist_bytes, 0, 0, y_limit, -1}};
        w.sink.on_full = &Worker::full;
        w.sink.ctx = &w;
        try {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= todo.size()) break;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (failed || cancelled) break;
                }
                const BlockRef *b = todo[i];
                const size_t n_out = inflate_block(h->map + b->pos, (size_t)b->size, h->size - (size_t)b->pos - (size_t)b->size,
                                                   buf, pad);
                if (n_out < 4) throw FormatError{"truncated structure"};
                int32_t n_rec;
                memcpy(&n_rec, buf.data(), 4);
                if (n_rec < 0) throw FormatError{"negative record count in a block"};
                if (w.cur < 0) w.next_slab();
                if (!decode_rows_fast(h->version, buf.data(), n_out, w.sink, use_norm ? &norm_vec : nullptr, max_dist))
                    decode_records(h->version, buf.data(), n_out, w.sink, use_norm ? &norm_vec : nullptr, max_dist);
            }
            w.publish();
        } catch (const Cancelled &) {
        } catch (const FormatError &e) {
            fail_with(e.what);
        } catch (...) {
            fail_with("out of memory");
        }
        std::lock_guard<std::mutex> lk(mu);
        ymax = w.sink.ymax > ymax ? w.sink.ymax : ymax;
        if (--active == 0) cv_ready.notify_all();
    }
};

extern "C" int mst_hic_stream_open(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                                   int64_t chrom_size_bp, int32_t n_threads, int32_t part, int32_t n_parts, void *slab_memory,
                                   int32_t n_slabs, int64_t slab_records, int32_t dist_bytes, mst_hic_stream **out) {
    if (!h || !chrom || !out || resolution <= 0 || n_parts < 1 || part < 0 || part >= n_parts || !slab_memory || n_slabs < 2 ||
        slab_records < 2 || (slab_records & 1) || (dist_bytes != 2 && dist_bytes != 4) ||
        (reinterpret_cast<uintptr_t>(slab_memory) & 3))
        return fail(MST_IO_E_ARG, "mst_hic_stream_open: bad argument (slab_records must be even, slab_memory 4-byte aligned)");
    if (dist_bytes == 2 && (max_dist_bins < 0 || max_dist_bins > 65535))
        return fail(MST_IO_E_ARG, "mst_hic_stream_open: 16-bit distances need 0 <= max_dist_bins <= 65535");
    *out = nullptr;
    mst_hic_stream *s = nullptr;
    try {
        s = new mst_hic_stream();
        s->h = h;
        std::vector<const BlockRef *> todo;
        const int rc = intra_todo(h, chrom, resolution, norm, max_dist_bins, todo, s->zoom, s->norm_vec, &s->use_norm);
        if (rc != MST_IO_OK) {
            delete s;
            return rc;
        }
        // todo points into `zoom.blocks`, which intra_todo filled inside s->zoom: the pointers stay valid with the stream
        s->blocks_total = (int32_t)todo.size();
        split_todo(todo, part, n_parts);
        s->todo.swap(todo);
        s->max_dist = max_dist_bins;
        s->y_limit = chrom_size_bp > 0 ? (chrom_size_bp + resolution - 1) / resolution : INT64_MAX;
        s->base = static_cast<uint8_t *>(slab_memory);
        s->n_slabs = n_slabs;
        s->cap = slab_records;
        s->dist_bytes = dist_bytes;
        s->counts.assign((size_t)n_slabs, 0);
        for (int32_t i = 0; i < n_slabs; ++i) s->free_q.push_back(i);
        int nt = n_threads > 0 ? n_threads : default_threads();
        if (nt < 1) nt = 1;
        if ((size_t)nt > s->todo.size()) nt = s->todo.empty() ? 1 : (int)s->todo.size();
        if (nt > n_slabs - 1) nt = n_slabs - 1;                 // every worker holds a slab; one more keeps the consumer fed
        s->active = nt;
        for (int t = 0; t < nt; ++t) s->workers.emplace_back([s] { s->work(); });
        *out = s;
        return MST_IO_OK;
    } catch (const FormatError &e) {
        delete s;
        return fail(MST_IO_E_FORMAT, "%s", e.what);
    } catch (...) {
        delete s;
        return fail(MST_IO_E_FORMAT, "unreadable file (out of memory?)");
    }
}

extern "C" int mst_hic_stream_next(mst_hic_stream *s, int32_t timeout_ms, int32_t *slab, int64_t *count) {
    if (!s || !slab || !count) return fail(MST_IO_E_ARG, "mst_hic_stream_next: bad argument");
    std::unique_lock<std::mutex> lk(s->mu);
    auto ready = [&] { return !s->ready_q.empty() || s->failed || s->active == 0; };
    if (timeout_ms < 0) s->cv_ready.wait(lk, ready);
    else if (!s->cv_ready.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) return 2;      // nothing yet
    if (s->failed) return fail(MST_IO_E_ZLIB, "block decode failed: %s", s->error.c_str());
    if (!s->ready_q.empty()) {
        *slab = s->ready_q.front();
        s->ready_q.pop_front();
        *count = s->counts[(size_t)*slab];
        return 1;
    }
    return 0;                                                                                         // all delivered
}

extern "C" int mst_hic_stream_release(mst_hic_stream *s, int32_t slab) {
    if (!s || slab < 0 || slab >= s->n_slabs) return fail(MST_IO_E_ARG, "mst_hic_stream_release: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    s->free_q.push_back(slab);
    s->cv_free.notify_one();
    return MST_IO_OK;
}

extern "C" int mst_hic_stream_close(mst_hic_stream *s, int64_t *n_bins, int64_t *total, int32_t *blocks_total,
                                    int32_t *blocks_mine) {
    if (!s) return MST_IO_OK;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->cancelled = true;
        s->cv_free.notify_all();
    }
    for (auto &t : s->workers) t.join();
    int rc = MST_IO_OK;
    if (s->failed) rc = fail(MST_IO_E_ZLIB, "block decode failed: %s", s->error.c_str());
    else if (s->ymax >= INT32_MAX) rc = fail(MST_IO_E_FORMAT, "bin index %lld does not fit 32 bits", (long long)s->ymax);
    if (n_bins) *n_bins = s->ymax + 1;
    if (total) *total = s->total;
    if (blocks_total) *blocks_total = s->blocks_total;
    if (blocks_mine) *blocks_mine = (int32_t)s->todo.size();
    delete s;
    return rc;
}

// ---- streaming RAW read: the host only inflates ----------------------------------------------------------------------------
// Worker threads inflate the chromosome's near-diagonal blocks (this part's share of them) and copy the RECORD BYTES of every
// row, exactly as the file stores them (2 or 4 bytes of column + 2 or 4 bytes of count: 6 bytes per record in the usual
// float-count file, against the 10 of a decoded packed record), into caller-owned slabs, with one 16-byte directory entry per
// row {byte offset, binY, binXOffset, record count | layout flags}.  The rows are decoded -- columns, counts, normalisation
// vector, distance / NaN / sign filters, scatter into the band -- by a kernel (mst_band_scatter_hic_rows in
// libmustache_hip.so), so no host thread touches a record.  Slab layout: payload from byte 0 upwards, directory entries from
// the slab's end downwards (entry k at slab_bytes - 16 (k + 1)); a slab is handed over when the next row would not fit.
// A block may be split between slabs at any row.  v6 blocks (plain records, no rows) are not served: MST_IO_E_FORMAT at open.
struct mst_hic_rawstream {
    mst_hic *h = nullptr;
    std::vector<const BlockRef *> todo;
    ZoomData zoom;
    std::vector<double> norm_vec;
    bool use_norm = false;
    uint8_t *base = nullptr;
    int32_t n_slabs = 0;
    int64_t slab_bytes = 0;
    std::vector<int64_t> pay_bytes;
    std::vector<int32_t> row_count;
    std::mutex mu;
    std::condition_variable cv_free, cv_ready;
    std::deque<int32_t> free_q, ready_q;
    std::vector<std::thread> workers;
    std::atomic<size_t> next{0};
    int active = 0;
    bool failed = false, cancelled = false;
    std::string error;
    int64_t rows_total = 0, bytes_total = 0, chrom_length = 0;
    int32_t blocks_total = 0;

    struct Cancelled {};

    void fail_with(const char *what) {
        std::lock_guard<std::mutex> lk(mu);
        if (!failed) error = what;
        failed = true;
        cv_free.notify_all();
        cv_ready.notify_all();
    }

    struct Worker {
        mst_hic_rawstream *s;
        int32_t cur = -1;
        uint8_t *m = nullptr;
        int64_t pay = 0;
        int32_t rows = 0;
        void publish() {
            if (cur < 0) return;
            std::lock_guard<std::mutex> lk(s->mu);
            s->pay_bytes[(size_t)cur] = pay;
            s->row_count[(size_t)cur] = rows;
            s->rows_total += rows;
            s->bytes_total += pay;
            if (rows > 0) {
                s->ready_q.push_back(cur);
                s->cv_ready.notify_one();
            } else {
                s->free_q.push_back(cur);
                s->cv_free.notify_one();
            }
            cur = -1;
        }
        void next_slab() {
            publish();
            std::unique_lock<std::mutex> lk(s->mu);
            s->cv_free.wait(lk, [&] { return !s->free_q.empty() || s->failed || s->cancelled; });
            if (s->failed || s->cancelled) throw Cancelled{};
            cur = s->free_q.front();
            s->free_q.pop_front();
            lk.unlock();
            m = s->base + (size_t)cur * (size_t)s->slab_bytes;
            pay = 0;
            rows = 0;
        }
        // directory entry of a row whose records already lie in this slab at byte `off`
        void list_row(int64_t off, int64_t count, int64_t y, int32_t x_off, uint32_t flags) {
            if (y < INT32_MIN || y > INT32_MAX) throw FormatError{"bin index does not fit 32 bits"};
            if (count >= ((int64_t)1 << 28)) throw FormatError{"a row holds more records than a slab"};
            mst_hic_row e{(uint32_t)off, (int32_t)y, x_off, (uint32_t)count | flags};
            memcpy(m + s->slab_bytes - 16 * ((int64_t)rows + 1), &e, 16);
            ++rows;
        }
        // one row: `count` records of `rec` bytes at p (columns + counts, or counts alone for a dense grid's row), copied in
        void put_row(const uint8_t *p, int64_t count, int64_t rec, int64_t y, int32_t x_off, uint32_t flags) {
            if (count <= 0) return;
            if (y < INT32_MIN || y > INT32_MAX) throw FormatError{"bin index does not fit 32 bits"};
            if (count >= ((int64_t)1 << 28)) throw FormatError{"a row holds more records than a slab"};
            const int64_t need = count * rec;
            if (cur < 0 || pay + need + 16 * ((int64_t)rows + 1) > s->slab_bytes) {
                if (need + 16 > s->slab_bytes) throw FormatError{"a row holds more records than a slab"};
                next_slab();
            }
            memcpy(m + pay, p, (size_t)need);
            mst_hic_row e{(uint32_t)pay, (int32_t)y, x_off, (uint32_t)count | flags};
            memcpy(m + s->slab_bytes - 16 * ((int64_t)rows + 1), &e, 16);
            pay += need + (need & 1);                 // (rec is even for every layout; kept even for the 16-bit loads)
            ++rows;
        }
    };

    // header + rows of one inflated block (the walk of decode_records, without touching a record): row(p, count, rec, y, x_off, flags)
    template <class Row>
    void walk_block(const uint8_t *data, size_t n, Row row) {
        Cursor c(data, n);
        if (c.get<int32_t>() < 0) throw FormatError{"negative record count in a block"};
        const int32_t x_off = c.get<int32_t>();
        const int32_t y_off = c.get<int32_t>();
        const bool short_counts = c.get<uint8_t>() == 0;
        bool short_x = true, short_y = true;
        if (h->version > 8) {
            short_x = c.get<uint8_t>() == 0;
            short_y = c.get<uint8_t>() == 0;
        }
        const uint8_t type = c.get<uint8_t>();
        const uint32_t fc = short_counts ? MST_HIC_ROW_SHORT_COUNTS : 0u;
        if (type == 1) {
            const int64_t rec = (short_x ? 2 : 4) + (short_counts ? 2 : 4);
            const int32_t rows = short_y ? (int32_t)c.get<int16_t>() : c.get<int32_t>();
            for (int32_t r = 0; r < rows; ++r) {
                const int32_t y = short_y ? (int32_t)c.get<int16_t>() : c.get<int32_t>();
                int32_t cols = short_x ? (int32_t)c.get<int16_t>() : c.get<int32_t>();
                if (cols < 0) cols = 0;                           // the decoder's loop runs zero times as well
                const uint8_t *p = c.p;
                c.skip((uint64_t)cols * (uint64_t)rec);
                if (cols > 0) row(p, (int64_t)cols, rec, (int64_t)y_off + y, x_off, fc | (short_x ? 0u : MST_HIC_ROW_INT_COLUMNS));
            }
        } else if (type == 2) {
            const int32_t n_pts = c.get<int32_t>();
            const int32_t wd = (int32_t)c.get<int16_t>();
            if (wd <= 0) throw FormatError{"dense block of width 0"};
            const int64_t rec = short_counts ? 2 : 4;
            for (int64_t i = 0; i < n_pts; i += wd) {
                const int64_t cnt = n_pts - i < wd ? n_pts - i : wd;
                const uint8_t *p = c.p;
                c.skip((uint64_t)cnt * (uint64_t)rec);
                row(p, cnt, rec, (int64_t)y_off + i / wd, x_off, fc | MST_HIC_ROW_DENSE);
            }
        } else {
            throw FormatError{"unknown block type"};
        }
    }

    // One block.  The usual case: the zlib stream is inflated STRAIGHT INTO the worker's slab (no staging buffer, no copy) when
    // the room left is at least `ratio` times the compressed size -- the largest expansion this worker has seen so far, with a
    // margin -- and its rows are then listed where they lie (the block's 14-18 header bytes and 4-8 bytes per row travel along
    // unused).  A block that does not fit what is left opens the next slab; one that does not fit an empty slab goes through the
    // staging buffer and is cut at row boundaries.
    void one_block(Worker &w, const BlockRef *b, std::vector<uint8_t> &buf, std::vector<uint8_t> &pad, double &ratio) {
        const uint8_t *comp = h->map + b->pos;
        const size_t comp_size = (size_t)b->size, past = h->size - (size_t)b->pos - comp_size;
        if (!use_zlib() && past >= mst_inflate::kSlack) {
            for (int attempt = 0; attempt < 2; ++attempt) {
                if (w.cur < 0) w.next_slab();
                const int64_t at = (w.pay + 15) / 16 * 16;
                // the directory grows down from the slab's end: room for this block's rows is kept free (checked after the walk)
                const int64_t room = slab_bytes - at - 16 * ((int64_t)w.rows + 1) - (int64_t)mst_inflate::kSlack;
                if (room >= (int64_t)((double)comp_size * ratio) + 64) {
                    size_t n_out = 0;
                    const int rc = mst_inflate::inflate_zlib(comp, comp_size, w.m + at, (size_t)room, &n_out);
                    if (rc == mst_inflate::kOk) {
                        const double seen = (double)n_out / (double)(comp_size ? comp_size : 1) * 1.05;
                        if (seen > ratio) ratio = seen;
                        int64_t nrows = 0;
                        walk_block(w.m + at, n_out, [&](const uint8_t *, int64_t, int64_t, int64_t, int32_t, uint32_t) { ++nrows; });
                        if (at + (int64_t)n_out + 16 * ((int64_t)w.rows + nrows) <= slab_bytes) {
                            walk_block(w.m + at, n_out, [&](const uint8_t *p, int64_t count, int64_t, int64_t y, int32_t x_off, uint32_t flags) {
                                w.list_row((int64_t)(p - w.m), count, y, x_off, flags);
                            });
                            w.pay = at + (int64_t)n_out;
                            w.pay += w.pay & 1;
                            return;
                        }
                    } else if (rc != mst_inflate::kOutputFull) {
                        throw FormatError{"zlib inflate failed"};
                    } else if (ratio < 64.0) {
                        ratio *= 2.0;                           // the estimate was too small: remember, and take a fresh slab
                    }
                }
                if (w.rows == 0 && w.pay == 0) break;           // does not fit an EMPTY slab: staged below
                w.next_slab();
            }
        }
        const size_t n_out = inflate_block(comp, comp_size, past, buf, pad);
        walk_block(buf.data(), n_out, [&](const uint8_t *p, int64_t count, int64_t rec, int64_t y, int32_t x_off, uint32_t flags) {
            w.put_row(p, count, rec, y, x_off, flags);
        });
    }

    void work() {
        std::vector<uint8_t> buf, pad;
        Worker w{this};
        double ratio = 2.0;
        try {
            for (;;) {
                const size_t i = next.fetch_add(1);
                if (i >= todo.size()) break;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (failed || cancelled) break;
                }
                one_block(w, todo[i], buf, pad, ratio);
            }
            w.publish();
        } catch (const Cancelled &) {
        } catch (const FormatError &e) {
            fail_with(e.what);
        } catch (...) {
            fail_with("out of memory");
        }
        std::lock_guard<std::mutex> lk(mu);
        if (--active == 0) cv_ready.notify_all();
    }
};

extern "C" int mst_hic_rawstream_open(mst_hic *h, const char *chrom, int32_t resolution, const char *norm, int64_t max_dist_bins,
                                      int32_t n_threads, int32_t part, int32_t n_parts, void *slab_memory, int32_t n_slabs,
                                      int64_t slab_bytes, mst_hic_rawstream **out) {
    if (!h || !chrom || !out || resolution <= 0 || n_parts < 1 || part < 0 || part >= n_parts || !slab_memory || n_slabs < 2 ||
        slab_bytes < 4096 || (slab_bytes & 15) || slab_bytes > ((int64_t)1 << 32) || (reinterpret_cast<uintptr_t>(slab_memory) & 15))
        return fail(MST_IO_E_ARG, "mst_hic_rawstream_open: bad argument (slab_bytes: a multiple of 16 in [4096, 2^32]; "
                                  "slab_memory 16-byte aligned)");
    *out = nullptr;
    if (h->version < 7)
        return fail(MST_IO_E_FORMAT, "mst_hic_rawstream_open: version %d blocks are plain records, not rows: use mst_hic_stream_open",
                    h->version);
    mst_hic_rawstream *s = nullptr;
    try {
        s = new mst_hic_rawstream();
        s->h = h;
        std::vector<const BlockRef *> todo;
        const int rc = intra_todo(h, chrom, resolution, norm, max_dist_bins, todo, s->zoom, s->norm_vec, &s->use_norm);
        if (rc != MST_IO_OK) {
            delete s;
            return rc;
        }
        s->blocks_total = (int32_t)todo.size();
        s->chrom_length = h->chroms[(size_t)find_chromosome(h, chrom)].length;
        split_todo(todo, part, n_parts);
        s->todo.swap(todo);
        s->base = static_cast<uint8_t *>(slab_memory);
        s->n_slabs = n_slabs;
        s->slab_bytes = slab_bytes;
        s->pay_bytes.assign((size_t)n_slabs, 0);
        s->row_count.assign((size_t)n_slabs, 0);
        for (int32_t i = 0; i < n_slabs; ++i) s->free_q.push_back(i);
        int nt = n_threads > 0 ? n_threads : default_threads();
        if (nt < 1) nt = 1;
        if ((size_t)nt > s->todo.size()) nt = s->todo.empty() ? 1 : (int)s->todo.size();
        if (nt > n_slabs - 1) nt = n_slabs - 1;                 // every worker holds a slab; one more keeps the consumer fed
        s->active = nt;
        for (int t = 0; t < nt; ++t) s->workers.emplace_back([s] { s->work(); });
        *out = s;
        return MST_IO_OK;
    } catch (const FormatError &e) {
        delete s;
        return fail(MST_IO_E_FORMAT, "%s", e.what);
    } catch (...) {
        delete s;
        return fail(MST_IO_E_FORMAT, "unreadable file (out of memory?)");
    }
}

extern "C" int mst_hic_rawstream_info(mst_hic_rawstream *s, const double **norm_values, int64_t *norm_count,
                                      int64_t *chrom_length_bp) {
    if (!s || !norm_values || !norm_count || !chrom_length_bp) return fail(MST_IO_E_ARG, "mst_hic_rawstream_info: bad argument");
    *norm_values = s->use_norm ? s->norm_vec.data() : nullptr;
    *norm_count = s->use_norm ? (int64_t)s->norm_vec.size() : -1;
    *chrom_length_bp = s->chrom_length;
    return MST_IO_OK;
}

extern "C" int mst_hic_rawstream_next(mst_hic_rawstream *s, int32_t timeout_ms, int32_t *slab, int64_t *payload_bytes,
                                      int32_t *rows) {
    if (!s || !slab || !payload_bytes || !rows) return fail(MST_IO_E_ARG, "mst_hic_rawstream_next: bad argument");
    std::unique_lock<std::mutex> lk(s->mu);
    auto ready = [&] { return !s->ready_q.empty() || s->failed || s->active == 0; };
    if (timeout_ms < 0) s->cv_ready.wait(lk, ready);
    else if (!s->cv_ready.wait_for(lk, std::chrono::milliseconds(timeout_ms), ready)) return 2;      // nothing yet
    if (s->failed) return fail(MST_IO_E_ZLIB, "block decode failed: %s", s->error.c_str());
    if (!s->ready_q.empty()) {
        *slab = s->ready_q.front();
        s->ready_q.pop_front();
        *payload_bytes = s->pay_bytes[(size_t)*slab];
        *rows = s->row_count[(size_t)*slab];
        return 1;
    }
    return 0;                                                                                         // all delivered
}

extern "C" int mst_hic_rawstream_release(mst_hic_rawstream *s, int32_t slab) {
    if (!s || slab < 0 || slab >= s->n_slabs) return fail(MST_IO_E_ARG, "mst_hic_rawstream_release: bad argument");
    std::lock_guard<std::mutex> lk(s->mu);
    s->free_q.push_back(slab);
    s->cv_free.notify_one();
    return MST_IO_OK;
}

extern "C" int mst_hic_rawstream_close(mst_hic_rawstream *s, int64_t *rows_total, int64_t *bytes_total, int32_t *blocks_total,
                                       int32_t *blocks_mine) {
    if (!s) return MST_IO_OK;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->cancelled = true;
        s->cv_free.notify_all();
    }
    for (auto &t : s->workers) t.join();
    int rc = MST_IO_OK;
    if (s->failed) rc = fail(MST_IO_E_ZLIB, "block decode failed: %s", s->error.c_str());
    if (rows_total) *rows_total = s->rows_total;
    if (bytes_total) *bytes_total = s->bytes_total;
    if (blocks_total) *blocks_total = s->blocks_total;
    if (blocks_mine) *blocks_mine = (int32_t)s->todo.size();
    delete s;
    return rc;
}
