// libmustache_io.so -- text contact maps (3 or 5 columns), see include/mustache_io.h.
//
// Restates what `pd.read_csv(f, sep=sep, header=None); df.dropna()` yields for the reference's read_pd()
// (reference mustache/mustache.py:254-265) on plain numeric files: pandas' C tokenizer + its default number converter
// precise_xstrtod (pandas/_libs/src/parser/tokenizer.c).  The file is memory-mapped, cut into byte ranges at line
// boundaries and parsed by a pool of threads; rows come back in file order.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mustache_io.h"

namespace mst_io {
int fail(int code, const char *fmt, ...);      // hic_reader.cpp (shared thread-local error buffer)
}

namespace {

const double kPow10[309] = {
#define E10(a) 1e##a
    1e0,   1e1,   1e2,   1e3,   1e4,   1e5,   1e6,   1e7,   1e8,   1e9,   1e10,  1e11,  1e12,  1e13,  1e14,  1e15,  1e16,
    1e17,  1e18,  1e19,  1e20,  1e21,  1e22,  1e23,  1e24,  1e25,  1e26,  1e27,  1e28,  1e29,  1e30,  1e31,  1e32,  1e33,
    1e34,  1e35,  1e36,  1e37,  1e38,  1e39,  1e40,  1e41,  1e42,  1e43,  1e44,  1e45,  1e46,  1e47,  1e48,  1e49,  1e50,
    1e51,  1e52,  1e53,  1e54,  1e55,  1e56,  1e57,  1e58,  1e59,  1e60,  1e61,  1e62,  1e63,  1e64,  1e65,  1e66,  1e67,
    1e68,  1e69,  1e70,  1e71,  1e72,  1e73,  1e74,  1e75,  1e76,  1e77,  1e78,  1e79,  1e80,  1e81,  1e82,  1e83,  1e84,
    1e85,  1e86,  1e87,  1e88,  1e89,  1e90,  1e91,  1e92,  1e93,  1e94,  1e95,  1e96,  1e97,  1e98,  1e99,  1e100, 1e101,
    1e102, 1e103, 1e104, 1e105, 1e106, 1e107, 1e108, 1e109, 1e110, 1e111, 1e112, 1e113, 1e114, 1e115, 1e116, 1e117, 1e118,
    1e119, 1e120, 1e121, 1e122, 1e123, 1e124, 1e125, 1e126, 1e127, 1e128, 1e129, 1e130, 1e131, 1e132, 1e133, 1e134, 1e135,
    1e136, 1e137, 1e138, 1e139, 1e140, 1e141, 1e142, 1e143, 1e144, 1e145, 1e146, 1e147, 1e148, 1e149, 1e150, 1e151, 1e152,
    1e153, 1e154, 1e155, 1e156, 1e157, 1e158, 1e159, 1e160, 1e161, 1e162, 1e163, 1e164, 1e165, 1e166, 1e167, 1e168, 1e169,
    1e170, 1e171, 1e172, 1e173, 1e174, 1e175, 1e176, 1e177, 1e178, 1e179, 1e180, 1e181, 1e182, 1e183, 1e184, 1e185, 1e186,
    1e187, 1e188, 1e189, 1e190, 1e191, 1e192, 1e193, 1e194, 1e195, 1e196, 1e197, 1e198, 1e199, 1e200, 1e201, 1e202, 1e203,
    1e204, 1e205, 1e206, 1e207, 1e208, 1e209, 1e210, 1e211, 1e212, 1e213, 1e214, 1e215, 1e216, 1e217, 1e218, 1e219, 1e220,
    1e221, 1e222, 1e223, 1e224, 1e225, 1e226, 1e227, 1e228, 1e229, 1e230, 1e231, 1e232, 1e233, 1e234, 1e235, 1e236, 1e237,
    1e238, 1e239, 1e240, 1e241, 1e242, 1e243, 1e244, 1e245, 1e246, 1e247, 1e248, 1e249, 1e250, 1e251, 1e252, 1e253, 1e254,
    1e255, 1e256, 1e257, 1e258, 1e259, 1e260, 1e261, 1e262, 1e263, 1e264, 1e265, 1e266, 1e267, 1e268, 1e269, 1e270, 1e271,
    1e272, 1e273, 1e274, 1e275, 1e276, 1e277, 1e278, 1e279, 1e280, 1e281, 1e282, 1e283, 1e284, 1e285, 1e286, 1e287, 1e288,
    1e289, 1e290, 1e291, 1e292, 1e293, 1e294, 1e295, 1e296, 1e297, 1e298, 1e299, 1e300, 1e301, 1e302, 1e303, 1e304, 1e305,
    1e306, 1e307, 1e308};
#undef E10

inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// pandas precise_xstrtod on the token [p, e) (blanks already trimmed).  Returns false when the token is not a number.
bool precise_xstrtod(const char *p, const char *e, double *out) {
    bool negative = false;
    if (p < e && (*p == '-' || *p == '+')) {
        negative = *p == '-';
        ++p;
    }
    double number = 0.0;
    int exponent = 0, num_digits = 0, num_decimals = 0;
    const int max_digits = 17;
    while (p < e && is_digit(*p)) {
        if (num_digits < max_digits) {
            number = number * 10. + (*p - '0');
            ++num_digits;
        } else {
            ++exponent;
        }
        ++p;
    }
    if (p < e && *p == '.') {
        ++p;
        while (num_digits < max_digits && p < e && is_digit(*p)) {
            number = number * 10. + (*p - '0');
            ++p;
            ++num_digits;
            ++num_decimals;
        }
        if (num_digits >= max_digits)
            while (p < e && is_digit(*p)) ++p;
        exponent -= num_decimals;
    }
    if (num_digits == 0) return false;
    if (negative) number = -number;
    if (p < e && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < e && (*p == '-' || *p == '+')) {
            eneg = *p == '-';
            ++p;
        }
        int n = 0, nd = 0;
        while (p < e && is_digit(*p)) {
            if (n < 100000) n = n * 10 + (*p - '0');
            ++p;
            ++nd;
        }
        if (nd == 0) return false;
        exponent += eneg ? -n : n;
    }
    if (p != e) return false;
    if (exponent > 308) {
        number = negative ? -HUGE_VAL : HUGE_VAL;
    } else if (exponent > 0) {
        number *= kPow10[exponent];
    } else if (exponent < -308) {
        if (exponent < -616) {
            number = 0.;
        } else {
            number /= kPow10[-308 - exponent];
            number /= kPow10[308];
        }
    } else {
        number /= kPow10[-exponent];
    }
    *out = number;
    return true;
}

// pandas' default NA strings (pandas/_libs/parsers.pyx STR_NA_VALUES)
bool is_na(const char *p, const char *e) {
    static const char *const na[] = {"",      "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN",
                                     "<NA>", "N/A",  "NA",       "NULL", "NaN",    "None",     "n/a",  "nan",  "null"};
    const size_t n = (size_t)(e - p);
    for (const char *s : na)
        if (strlen(s) == n && memcmp(s, p, n) == 0) return true;
    return false;
}

bool is_inf(const char *p, const char *e, double *out) {
    bool neg = false;
    if (p < e && (*p == '-' || *p == '+')) {
        neg = *p == '-';
        ++p;
    }
    const size_t n = (size_t)(e - p);
    auto ieq = [&](const char *s) {
        if (strlen(s) != n) return false;
        for (size_t i = 0; i < n; ++i)
            if ((p[i] | 0x20) != s[i]) return false;
        return true;
    };
    if (ieq("inf") || ieq("infinity")) {
        *out = neg ? -HUGE_VAL : HUGE_VAL;
        return true;
    }
    return false;
}

// str(c).replace('chr', '') == str(s).replace('chr', '')   (is_chr, mustache.py:192-196)
std::string strip_chr(const char *p, const char *e) {
    std::string s(p, e), out;
    size_t i = 0;
    while (i < s.size()) {
        if (s.compare(i, 3, "chr") == 0) i += 3;
        else out.push_back(s[i++]);
    }
    return out;
}

struct Rows {
    std::vector<double> a, b, c;
};

enum { FIELD_OK = 0, FIELD_NA = 1, FIELD_BAD = 2 };

int parse_number(const char *p, const char *e, double *out) {
    while (p < e && (*p == ' ' || *p == '\t')) ++p;
    while (e > p && (e[-1] == ' ' || e[-1] == '\t' || e[-1] == '\r')) --e;
    if (precise_xstrtod(p, e, out)) return FIELD_OK;
    if (is_na(p, e)) return FIELD_NA;
    if (is_inf(p, e, out)) return FIELD_OK;
    return FIELD_BAD;
}

// parse the lines of [p, end); returns false on a construct the restatement does not cover
bool parse_range(const char *p, const char *end, char sep, int ncols, const std::string &chrom, Rows &out) {
    const char *f0[8], *f1[8];
    while (p < end) {
        const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = eol ? eol : end;
        const char *next = eol ? eol + 1 : end;
        const char *q = le;
        if (q > p && q[-1] == '\r') --q;
        if (q == p) {                       // blank line (skip_blank_lines=True)
            p = next;
            continue;
        }
        int nf = 0;
        const char *s = p;
        for (const char *t = p;; ++t) {
            if (t == q || *t == sep) {
                if (nf >= ncols) return false;             // more fields than the first line: pandas raises
                f0[nf] = s;
                f1[nf] = t;
                ++nf;
                s = t + 1;
                if (t == q) break;
            } else if (*t == '"') {
                return false;
            }
        }
        p = next;
        if (nf < ncols) continue;                          // short line: missing fields are NaN -> dropna() drops the row
        double v[3];
        const int num_idx3[3] = {0, 1, 2}, num_idx5[3] = {1, 3, 4};
        const int *idx = ncols == 3 ? num_idx3 : num_idx5;
        bool drop = false;
        for (int k = 0; k < 3 && !drop; ++k) {
            const int r = parse_number(f0[idx[k]], f1[idx[k]], &v[k]);
            if (r == FIELD_BAD) return false;
            if (r == FIELD_NA || std::isnan(v[k])) drop = true;
        }
        if (drop) continue;
        if (ncols == 5) {
            bool na_name = false, match = true;
            for (int c : {0, 2}) {
                const char *a = f0[c], *b = f1[c];
                while (a < b && (*a == ' ' || *a == '\t')) ++a;
                while (b > a && (b[-1] == ' ' || b[-1] == '\t')) --b;
                if (is_na(a, b)) na_name = true;
                else if (strip_chr(a, b) != chrom) match = false;
            }
            if (na_name || !match) continue;
        }
        out.a.push_back(v[0]);
        out.b.push_back(v[1]);
        out.c.push_back(v[2]);
    }
    return true;
}

}  // namespace

extern "C" int64_t mst_text_read_contacts(const char *path, char sep, const char *chrom, int32_t n_threads, int32_t *n_cols,
                                          double **pos1, double **pos2, double **count) {
    if (!path || !n_cols || !pos1 || !pos2 || !count || sep == '\0' || sep == '\n' || sep == '"')
        return mst_io::fail(MST_IO_E_ARG, "mst_text_read_contacts: bad argument");
    *pos1 = *pos2 = *count = nullptr;
    *n_cols = 0;
    const int fd = open(path, O_RDONLY);
    struct stat st;
    if (fd < 0 || fstat(fd, &st) != 0) {
        if (fd >= 0) close(fd);
        return mst_io::fail(MST_IO_E_FILE, "cannot open %s", path);
    }
    const size_t size = (size_t)st.st_size;
    const char *base = nullptr;
    if (size) {
        void *m = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            close(fd);
            return mst_io::fail(MST_IO_E_FILE, "cannot map %s", path);
        }
        base = (const char *)m;
    }
    int64_t result = 0;
    try {
        // number of columns = fields of the first non-blank line
        const char *p = base, *end = base + size;
        int ncols = 0;
        while (p < end && ncols == 0) {
            const char *eol = (const char *)memchr(p, '\n', (size_t)(end - p));
            const char *q = eol ? eol : end;
            if (q > p && q[-1] == '\r') --q;
            if (q > p) {
                ncols = 1;
                for (const char *t = p; t < q; ++t) ncols += (*t == sep);
            }
            p = eol ? eol + 1 : end;
        }
        if (ncols != 3 && ncols != 5) {
            result = ncols == 0 ? mst_io::fail(MST_IO_E_FORMAT, "%s: empty file", path)
                                : mst_io::fail(MST_IO_E_FORMAT, "%s: %d columns (3 or 5 expected)", path, ncols);
        } else {
            const std::string want = chrom ? strip_chr(chrom, chrom + strlen(chrom)) : std::string();
            int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
            if (nt < 1) nt = 1;
            const size_t kMinChunk = (size_t)4 << 20;
            size_t nchunks = size / kMinChunk + 1;
            if (nchunks > 4096) nchunks = 4096;
            if ((size_t)nt > nchunks) nt = (int)nchunks;
            std::vector<const char *> cut(nchunks + 1);
            cut[0] = base;
            cut[nchunks] = end;
            for (size_t i = 1; i < nchunks; ++i) {          // advance each cut to the next line start
                const char *c = base + size / nchunks * i;
                if (c < cut[i - 1]) c = cut[i - 1];
                const char *nl = c < end ? (const char *)memchr(c, '\n', (size_t)(end - c)) : nullptr;
                cut[i] = nl ? nl + 1 : end;
            }
            std::vector<Rows> part(nchunks);
            std::atomic<size_t> next(0);
            std::atomic<int> bad(0);
            auto work = [&]() {
                for (;;) {
                    const size_t i = next.fetch_add(1);
                    if (i >= nchunks || bad.load()) return;
                    try {
                        if (!parse_range(cut[i], cut[i + 1], sep, ncols, want, part[i])) bad.store(1);
                    } catch (...) {
                        bad.store(2);
                    }
                }
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; ++t) pool.emplace_back(work);
            work();
            for (auto &t : pool) t.join();
            if (bad.load() == 1) {
                result = mst_io::fail(MST_IO_E_FORMAT, "%s: not a plain numeric %d-column file", path, ncols);
            } else if (bad.load()) {
                result = mst_io::fail(MST_IO_E_FILE, "out of memory while parsing %s", path);
            } else {
                size_t total = 0;
                for (const Rows &r : part) total += r.c.size();
                double *a = (double *)malloc((total ? total : 1) * sizeof(double));
                double *b = (double *)malloc((total ? total : 1) * sizeof(double));
                double *c = (double *)malloc((total ? total : 1) * sizeof(double));
                if (!a || !b || !c) {
                    free(a);
                    free(b);
                    free(c);
                    result = mst_io::fail(MST_IO_E_FILE, "out of memory for %zu rows", total);
                } else {
                    size_t off = 0;
                    for (const Rows &r : part) {
                        if (r.c.empty()) continue;
                        memcpy(a + off, r.a.data(), r.c.size() * sizeof(double));
                        memcpy(b + off, r.b.data(), r.c.size() * sizeof(double));
                        memcpy(c + off, r.c.data(), r.c.size() * sizeof(double));
                        off += r.c.size();
                    }
                    *pos1 = a;
                    *pos2 = b;
                    *count = c;
                    *n_cols = ncols;
                    result = (int64_t)total;
                }
            }
        }
    } catch (...) {
        result = mst_io::fail(MST_IO_E_FILE, "out of memory while parsing %s", path);
    }
    if (base) munmap((void *)base, size);
    close(fd);
    return result;
}

// The in-place fills of the reference's mustache() on a caller's dense host block (mustache.py:703-706): 2 on and below
// diagonal 4, and (within a chromosome) from diagonal dpx + 1 outwards.  Rows are `row_stride` doubles apart.
extern "C" int mst_host_fill_block(double *c, int64_t n, int64_t row_stride, int32_t dpx, int32_t intra, int32_t n_threads) {
    if (!c || n <= 0 || row_stride < n || dpx < 0) return mst_io::fail(MST_IO_E_ARG, "mst_host_fill_block: bad argument");
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(n_threads > 0 ? n_threads : 8, n / 256 + 1));
    auto work = [&](int t) {
        for (int64_t r = t; r < n; r += nt) {           // interleaved rows: the filled length grows with r
            double *row = c + r * row_stride;
            const int64_t lo = std::min<int64_t>(n, r + 5);
            for (int64_t j = 0; j < lo; ++j) row[j] = 2.0;
            if (intra)
                for (int64_t j = r + (int64_t)dpx + 1; j < n; ++j) row[j] = 2.0;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    return MST_IO_OK;
}

