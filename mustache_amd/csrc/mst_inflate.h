// zlib-stream (RFC 1950 / 1951) decoder for the `.hic` block reader -- host only, header only, no third-party code.
//
// Why not zlib's inflate(): a chromosome at 1 kb is ~900 independent zlib streams of ~1 MB each, float32 counts barely
// compress (ratio ~0.86), so the streams are almost all literals -- and zlib 1.2.11 decodes a literal per ~10 cycles through
// its 9-bit table and 32-bit bit buffer.  This decoder is the textbook table-driven form for 64-bit machines:
//   * a 64-bit bit buffer refilled with ONE unaligned 8-byte load (branch-free: `in += (63 - bitcnt) >> 3; bitcnt |= 56`);
//     after a refill >= 56 bits are available, enough for three literal/length codes (<= 15 bits each) or for a complete
//     length + distance pair (15 + 5 + 15 + 13 = 48 bits) without another load;
//   * an 11-bit primary table for the literal/length alphabet and an 8-bit one for distances, each entry a packed uint32
//     {bits to consume, kind, base value, extra-bit count}; longer codes go through second-level tables;
//   * up to three literals per refill; matches are copied 8 bytes at a time (distance >= 8) with the output buffer's slack
//     absorbing the overshoot.
// The Adler-32 of the output is verified (AVX2 when the CPU has it), as zlib does: a corrupted block is an error, never a
// silently different record set.
// Contract: `src` must be READABLE for kSlack bytes past `src + n` (the reader copies the last blocks of a file into a padded
// buffer; all others lie inside the memory-mapped file), `dst` must have kSlack bytes of room past `cap`.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace mst_inflate {

constexpr int kOk = 0, kCorrupt = -1, kOutputFull = -2;
// readable bytes past the input, writable bytes past the output capacity.  Input: the overrun test lets `in` stand at most 8
// bytes past the end, at most two refills (7 bytes each) follow before the next test, and a refill loads 8 bytes: 8 + 14 + 8 = 30.
constexpr size_t kSlack = 32;
constexpr size_t kOutMargin = 258 + 8;   // the fast loop wants room for one maximal match + copy overshoot

namespace detail {

constexpr int kLitlenRoot = 11, kDistRoot = 8;
constexpr int kLitlenCap = 2048 + 1024, kDistCap = 256 + 512;      // primary + second-level entries (libdeflate: 2342 / 402)
// entry: bits 0-4 code bits to consume; bits 5-7 kind; bits 8-23 value; bits 24-27 extra-bit count / second-level bits
enum Kind : uint32_t { kLiteral = 0, kLength = 1, kEnd = 2, kSub = 3, kBad = 4, kDistance = 5 };
inline uint32_t make(uint32_t nbits, uint32_t kind, uint32_t value, uint32_t extra) {
    return nbits | (kind << 5) | (value << 8) | (extra << 24);
}
inline uint32_t e_bits(uint32_t e) { return e & 31u; }
inline uint32_t e_kind(uint32_t e) { return (e >> 5) & 7u; }
inline uint32_t e_value(uint32_t e) { return (e >> 8) & 0xFFFFu; }
inline uint32_t e_extra(uint32_t e) { return (e >> 24) & 15u; }

inline uint32_t reverse_bits(uint32_t code, int len) {          // len <= 15: swap the bits of a 16-bit word, keep the top `len`
    uint32_t v = code & 0xFFFFu;
    v = ((v & 0x5555u) << 1) | ((v >> 1) & 0x5555u);
    v = ((v & 0x3333u) << 2) | ((v >> 2) & 0x3333u);
    v = ((v & 0x0F0Fu) << 4) | ((v >> 4) & 0x0F0Fu);
    v = ((v & 0x00FFu) << 8) | ((v >> 8) & 0x00FFu);
    return v >> (16 - len);
}

static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115,
                                      131, 163, 195, 227, 258};
static const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537,
                                       2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12,
                                       13, 13};

// payload of a symbol, without its code length
inline uint32_t litlen_payload(int sym) {
    if (sym < 256) return make(0, kLiteral, (uint32_t)sym, 0);
    if (sym == 256) return make(0, kEnd, 0, 0);
    if (sym <= 285) return make(0, kLength, kLenBase[sym - 257], kLenExtra[sym - 257]);
    return make(0, kBad, 0, 0);
}
inline uint32_t dist_payload(int sym) {
    if (sym < 30) return make(0, kDistance, kDistBase[sym], kDistExtra[sym]);
    return make(0, kBad, 0, 0);
}

// Canonical Huffman code (RFC 1951 3.2.2) -> decode table indexed by the next `root` input bits (LSB first).  Codes longer
// than `root` share a primary entry {kind kSub, value = start of their second-level table, extra = its index bits}.
// Returns false for an over-subscribed code, or an incomplete one -- except, as zlib's inftrees.c has it, `allow_incomplete`
// alphabets (literal/length and distance, not the code-length code) whose LONGEST code is one bit long (a block with a single
// literal/length or distance symbol), and a distance alphabet without any code (a block of literals only).
template <class Payload>
inline bool build_table(const uint8_t *lens, int nsyms, int root, uint32_t *table, int cap, bool allow_incomplete,
                        Payload payload) {
    int count[16] = {0};
    for (int s = 0; s < nsyms; ++s) ++count[lens[s]];
    count[0] = 0;
    int left = 1, used = 0, maxlen = 0;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;
        used += count[l];
        if (count[l]) maxlen = l;
    }
    if (left > 0 && !(allow_incomplete && (maxlen == 1 || used == 0))) return false;
    uint32_t next_code[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; ++l) {
        code = (code + (uint32_t)count[l - 1]) << 1;
        next_code[l] = code;
    }
    const int nroot = 1 << root;
    const uint32_t bad = make(1, kBad, 0, 0);
    for (int i = 0; i < nroot; ++i) table[i] = bad;
    // longest code per primary prefix among the codes that do not fit the primary table
    uint8_t longest[1 << kLitlenRoot];
    bool any_long = false;
    uint32_t rev[288];
    for (int s = 0; s < nsyms; ++s) {
        const int l = lens[s];
        if (!l) continue;
        rev[s] = reverse_bits(next_code[l]++, l);
        if (l > root) any_long = true;
    }
    if (any_long) {
        memset(longest, 0, (size_t)nroot);
        for (int s = 0; s < nsyms; ++s)
            if (lens[s] > root) {
                uint8_t &m = longest[rev[s] & (uint32_t)(nroot - 1)];
                if (lens[s] > m) m = lens[s];
            }
    }
    int next_free = nroot;
    for (int s = 0; s < nsyms; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t pay = payload(s);
        if (l <= root) {
            const uint32_t e = pay | (uint32_t)l;
            for (uint32_t i = rev[s]; i < (uint32_t)nroot; i += 1u << l) table[i] = e;
            continue;
        }
        const uint32_t p = rev[s] & (uint32_t)(nroot - 1);
        uint32_t head = table[p];
        if (e_kind(head) != kSub) {
            const int sub_bits = longest[p] - root;
            if (next_free + (1 << sub_bits) > cap) return false;
            head = make((uint32_t)root, kSub, (uint32_t)next_free, (uint32_t)sub_bits);
            table[p] = head;
            for (int i = 0; i < (1 << sub_bits); ++i) table[next_free + i] = bad;
            next_free += 1 << sub_bits;
        }
        const uint32_t start = e_value(head), sub_bits = e_extra(head);
        const uint32_t e = pay | (uint32_t)(l - root);
        for (uint32_t i = rev[s] >> root; i < (1u << sub_bits); i += 1u << (l - root)) table[start + i] = e;
    }
    return true;
}

inline uint64_t load64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8);
    return v;          // little-endian hosts only (x86-64, aarch64 LE): checked by a static_assert below
}
#if defined(__BYTE_ORDER__) && defined(__ORDER_LITTLE_ENDIAN__)
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "mst_inflate.h assumes a little-endian host");
#endif

inline uint32_t adler32_scalar(uint32_t adler, const uint8_t *p, size_t n) {
    uint32_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        while (k >= 8) {
            s1 += p[0]; s2 += s1; s1 += p[1]; s2 += s1; s1 += p[2]; s2 += s1; s1 += p[3]; s2 += s1;
            s1 += p[4]; s2 += s1; s1 += p[5]; s2 += s1; s1 += p[6]; s2 += s1; s1 += p[7]; s2 += s1;
            p += 8;
            k -= 8;
        }
        while (k--) {
            s1 += *p++;
            s2 += s1;
        }
        s1 %= 65521u;
        s2 %= 65521u;
    }
    return (s2 << 16) | s1;
}

#if defined(__x86_64__)
// 32 bytes per step: s1 += sum(b), s2 += 32 * s1_before + sum((32 - i) * b[i]); the lane sums stay below 2^32 for a
// 5536-byte chunk (173 steps), the final combination is done in 64 bits
__attribute__((target("avx2"))) inline uint32_t adler32_avx2(uint32_t adler, const uint8_t *p, size_t n) {
    uint64_t s1 = adler & 0xFFFF, s2 = adler >> 16;
    const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12,
                                             11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m256i ones16 = _mm256_set1_epi16(1), zero = _mm256_setzero_si256();
    while (n >= 32) {
        size_t steps = n / 32;
        if (steps > 173) steps = 173;
        n -= steps * 32;
        __m256i v_s1 = zero, v_ps = zero, v_s2 = zero;        // byte sums (4 x u64), sums of the previous v_s1, weighted sums (8 x u32)
        for (size_t j = 0; j < steps; ++j) {
            const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p));
            p += 32;
            v_ps = _mm256_add_epi64(v_ps, v_s1);
            v_s1 = _mm256_add_epi64(v_s1, _mm256_sad_epu8(b, zero));
            v_s2 = _mm256_add_epi32(v_s2, _mm256_madd_epi16(_mm256_maddubs_epi16(b, weights), ones16));
        }
        uint64_t a1[4], aps[4];
        uint32_t a2[8];
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(a1), v_s1);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(aps), v_ps);
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(a2), v_s2);
        const uint64_t sum1 = a1[0] + a1[1] + a1[2] + a1[3], sumps = aps[0] + aps[1] + aps[2] + aps[3];
        uint64_t sum2 = 0;
        for (int i = 0; i < 8; ++i) sum2 += a2[i];
        s2 = (s2 + 32 * (steps * s1 + sumps) + sum2) % 65521u;
        s1 = (s1 + sum1) % 65521u;
    }
    return adler32_scalar((uint32_t)((s2 << 16) | s1), p, n);
}
#endif

inline uint32_t adler32(uint32_t adler, const uint8_t *p, size_t n) {
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) return adler32_avx2(adler, p, n);
#endif
    return adler32_scalar(adler, p, n);
}

struct Tables {
    uint32_t litlen[kLitlenCap];
    uint32_t dist[kDistCap];
};

}  // namespace detail

// Raw deflate stream (RFC 1951) src[0, n) -> dst; *out_n = bytes produced, *in_used = bytes of src consumed.
inline int inflate_raw(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_n, size_t *in_used) {
    using namespace detail;
    const uint8_t *in = src, *const in_end = src + n;
    uint8_t *out = dst, *const out_end = dst + cap;
    uint64_t bitbuf = 0;
    uint32_t bitcnt = 0;
    Tables t;
    static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

#define MST_REFILL()                                              \
    do {                                                          \
        bitbuf |= load64(in) << (bitcnt & 63);                    \
        in += (63 - bitcnt) >> 3;                                 \
        bitcnt |= 56;                                             \
    } while (0)
#define MST_TAKE(nb_) (tmp_ = (uint32_t)(bitbuf & ((1ull << (nb_)) - 1)), bitbuf >>= (nb_), bitcnt -= (nb_), tmp_)
#define MST_OVERRUN() (in > in_end + 8)          /* a valid stream never has more than 7 prefetched bytes past its end */

    uint32_t tmp_;
    bool final = false;
    while (!final) {
        if (MST_OVERRUN()) return kCorrupt;
        MST_REFILL();
        final = MST_TAKE(1) != 0;
        const uint32_t type = MST_TAKE(2);
        if (type == 0) {
            // stored: drop the rest of the byte, give the whole buffered bytes back to the input
            const uint32_t drop = bitcnt & 7;
            bitbuf >>= drop;
            bitcnt -= drop;
            in -= bitcnt >> 3;
            bitbuf = 0;
            bitcnt = 0;
            if (in + 4 > in_end) return kCorrupt;
            const uint32_t len = in[0] | (in[1] << 8), nlen = in[2] | (in[3] << 8);
            in += 4;
            if ((len ^ 0xFFFFu) != nlen || (size_t)(in_end - in) < len) return kCorrupt;
            if ((size_t)(out_end - out) < len) return kOutputFull;
            memcpy(out, in, len);
            out += len;
            in += len;
            continue;
        }
        if (type == 3) return kCorrupt;
        uint8_t lens[288 + 32];
        int nlit, ndist;
        if (type == 1) {
            nlit = 288;
            ndist = 32;
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
        } else {
            nlit = (int)MST_TAKE(5) + 257;
            ndist = (int)MST_TAKE(5) + 1;
            const int nclen = (int)MST_TAKE(4) + 4;
            if (nlit > 286 || ndist > 30) return kCorrupt;
            uint8_t cl[19] = {0};
            for (int i = 0; i < nclen; ++i) {
                if (bitcnt < 3) {
                    if (MST_OVERRUN()) return kCorrupt;
                    MST_REFILL();
                }
                cl[kOrder[i]] = (uint8_t)MST_TAKE(3);
            }
            uint32_t cltab[128 + 8];
            if (!build_table(cl, 19, 7, cltab, 128, false, [](int s) { return make(0, kLiteral, (uint32_t)s, 0); }))
                return kCorrupt;
            int i = 0;
            while (i < nlit + ndist) {
                if (MST_OVERRUN()) return kCorrupt;
                MST_REFILL();
                const uint32_t e = cltab[bitbuf & 127];
                if (e_kind(e) != kLiteral) return kCorrupt;
                (void)MST_TAKE(e_bits(e));
                const uint32_t sym = e_value(e);
                if (sym < 16) {
                    lens[i++] = (uint8_t)sym;
                    continue;
                }
                uint32_t rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return kCorrupt;
                    val = lens[i - 1];
                    rep = 3 + MST_TAKE(2);
                } else if (sym == 17) {
                    rep = 3 + MST_TAKE(3);
                } else {
                    rep = 11 + MST_TAKE(7);
                }
                if (i + (int)rep > nlit + ndist) return kCorrupt;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) return kCorrupt;           // no end-of-block code
            if (nlit < 288) memmove(lens + 288, lens + nlit, (size_t)ndist);
            for (int k = nlit; k < 288; ++k) lens[k] = 0;
        }
        if (!build_table(lens, 288, kLitlenRoot, t.litlen, kLitlenCap, true, litlen_payload)) return kCorrupt;
        if (!build_table(lens + 288, ndist, kDistRoot, t.dist, kDistCap, true, dist_payload)) return kCorrupt;

        // ---- symbols of the block.  (A fully branch-free form -- distance table looked up for every symbol, bits consumed
        // under a mask, one 8-byte copy per symbol -- was built and measured: it removes the literal-or-match mispredictions
        // but chains both table look-ups into the bit buffer's dependency, 21 cycles per symbol against 12 for this form.)
        // software-pipelined: the table entry of the NEXT symbol is fetched before the current symbol's output work (the copy of
        // a match, the third literal's store), so its L1 latency is off the bit buffer's dependency chain
#define MST_RESOLVE(tab_)                                                                   \
    if (__builtin_expect(e_kind(e) == kSub, 0)) {                                           \
        bitbuf >>= e_bits(e);                                                               \
        bitcnt -= e_bits(e);                                                                \
        e = (tab_)[e_value(e) + (uint32_t)(bitbuf & ((1u << e_extra(e)) - 1))];             \
    }                                                                                       \
    bitbuf >>= e_bits(e);                                                                   \
    bitcnt -= e_bits(e);
        MST_REFILL();
        uint32_t e = t.litlen[bitbuf & ((1u << kLitlenRoot) - 1)];          // fetched, not yet consumed
        for (;;) {
            if (MST_OVERRUN()) return kCorrupt;
            if ((size_t)(out_end - out) < kOutMargin) return kOutputFull;
            MST_RESOLVE(t.litlen)
            if (e_kind(e) == kLiteral) {
                *out++ = (uint8_t)e_value(e);
                e = t.litlen[bitbuf & ((1u << kLitlenRoot) - 1)];
                MST_RESOLVE(t.litlen)
                if (e_kind(e) == kLiteral) {
                    *out++ = (uint8_t)e_value(e);
                    e = t.litlen[bitbuf & ((1u << kLitlenRoot) - 1)];
                    MST_RESOLVE(t.litlen)
                    if (e_kind(e) == kLiteral) {
                        const uint8_t lit3 = (uint8_t)e_value(e);
                        MST_REFILL();
                        e = t.litlen[bitbuf & ((1u << kLitlenRoot) - 1)];
                        *out++ = lit3;
                        continue;
                    }
                }
                MST_REFILL();                      // a length's extra bits + a whole distance: up to 33 more bits
            }
            if (__builtin_expect(e_kind(e) != kLength, 0)) {
                if (e_kind(e) == kEnd) break;
                return kCorrupt;
            }
            const uint32_t len = e_value(e) + MST_TAKE(e_extra(e));
            e = t.dist[bitbuf & ((1u << kDistRoot) - 1)];
            MST_RESOLVE(t.dist)
            if (__builtin_expect(e_kind(e) != kDistance, 0)) return kCorrupt;
            const uint32_t dist = e_value(e) + MST_TAKE(e_extra(e));
            if (__builtin_expect(dist > (size_t)(out - dst), 0)) return kCorrupt;
            MST_REFILL();
            e = t.litlen[bitbuf & ((1u << kLitlenRoot) - 1)];               // the symbol after the match, before its copy
            const uint8_t *from = out - dist;
            uint8_t *const stop = out + len;
            if (len <= 8 && dist >= len) {         // the common case in row lists: 3-4 bytes from a few records back
                const uint64_t w8 = load64(from);  // load, then store: source and destination may overlap beyond `len`
                memcpy(out, &w8, 8);
            } else if (dist >= 8) {
                do {
                    memcpy(out, from, 8);
                    out += 8;
                    from += 8;
                } while (out < stop);
            } else if (dist == 1) {
                memset(out, from[0], len);
            } else {
                while (out < stop) *out++ = *from++;
            }
            out = stop;
        }
#undef MST_RESOLVE
    }
    // give back the whole bytes still buffered
    in -= bitcnt >> 3;
    if (in > in_end) return kCorrupt;
    *out_n = (size_t)(out - dst);
    *in_used = (size_t)(in - src);
    return kOk;
#undef MST_REFILL
#undef MST_TAKE
#undef MST_OVERRUN
}

// zlib stream (RFC 1950): 2-byte header, deflate data, Adler-32 of the output (verified).
inline int inflate_zlib(const uint8_t *src, size_t n, uint8_t *dst, size_t cap, size_t *out_n) {
    if (n < 6) return kCorrupt;
    const uint32_t cmf = src[0], flg = src[1];
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return kCorrupt;   // deflate, <= 32 KB window, no preset dictionary
    size_t produced = 0, used = 0;
    const int rc = inflate_raw(src + 2, n - 2, dst, cap, &produced, &used);
    if (rc != kOk) return rc;
    if (used + 4 > n - 2) return kCorrupt;
    const uint8_t *a = src + 2 + used;
    const uint32_t want = ((uint32_t)a[0] << 24) | ((uint32_t)a[1] << 16) | ((uint32_t)a[2] << 8) | (uint32_t)a[3];
    if (detail::adler32(1, dst, produced) != want) return kCorrupt;
    *out_n = produced;
    return kOk;
}

}  // namespace mst_inflate
