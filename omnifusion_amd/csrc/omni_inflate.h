// omni_inflate.h — zlib-stream (RFC 1950) / DEFLATE (RFC 1951) decoder and the two checksums of a PNG file, host code only.
//
// Why not libz (rounds 4-5 used it): the decode pool is what bounds a PNG-fed evaluation (dataset_loader_stanford.py:85,96 -> test.py:90-97) — the GPU
// boxes grant 16 CPUs and the forward consumes ~4000 panoramas/s; libz's inflate is 70 % of a panorama's 7-14 ms, its table-driven crc32 another 7 %.
// A photograph's stream is mostly literals, which libz decodes one per loop trip with a 32-bit bit buffer.  Here: a 64-bit bit buffer refilled by one
// unaligned load, an 11-bit first-level table for literals / lengths (up to three literals per refill), matches copied 16 bytes at a time, Adler-32
// with SSSE3 and CRC-32 by carry-less multiplication where the CPU has them (sliced by 8 otherwise).
//
// The input is untrusted: every read is inside [in, in_end) and every write inside [out, out_end); the fast loop runs only while 16 input bytes and
// 6 + 258 + 16 output bytes remain, the careful loop finishes.  A code-length set is accepted exactly where RFC 1951 / libz accept it (over-subscribed:
// refused; incomplete: refused unless it is a single 1-bit code; no end-of-block code: refused); distances beyond the produced output: refused.
// tests/test_png.py decodes the same streams with Python's zlib (valid, truncated at every length class, bit-flipped) and compares result and refusal.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <immintrin.h>

namespace omni_inflate {

enum { INF_OK = 0, INF_CORRUPT = -1, INF_TRUNCATED = -2, INF_OUTPUT_FULL = -3 };

// ------------------------------------------------------------------ checksums
inline const uint32_t (*crc_tables())[256]
{
    static uint32_t t[8][256];
    static const bool once = [] {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
        return true;
    }();
    (void)once;
    return t;
}

// CRC-32 (ISO 3309, the PNG chunk check): crc32(0, p, n) of an empty prefix is 0, like libz's
inline uint32_t crc32(uint32_t crc, const unsigned char* p, size_t n)
{
    const uint32_t (*t)[256] = crc_tables();
    uint32_t c = ~crc;
    while (n && ((uintptr_t)p & 7)) { c = (c >> 8) ^ t[0][(c ^ *p++) & 0xff]; --n; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = t[7][v & 0xff] ^ t[6][(v >> 8) & 0xff] ^ t[5][(v >> 16) & 0xff] ^ t[4][(v >> 24) & 0xff] ^
            t[3][(v >> 32) & 0xff] ^ t[2][(v >> 40) & 0xff] ^ t[1][(v >> 48) & 0xff] ^ t[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ t[0][(c ^ *p++) & 0xff];
    return ~c;
}

// the same CRC by carry-less multiplication (PCLMULQDQ): four 128-bit lanes folded across 64 bytes per step, X -> X.lo * K(D + 32) ^ X.hi * K(D - 32) ^ next
// with K(n) = bit-reflect32(x^n mod P) << 1, P = 0x104C11DB7 (D = 512: 0x154442bd4, 0x1c6e41596; D = 128: 0x1751997d0, 0xccaa009e — computed from that
// formula and checked against the table form, tests/test_png.py).  The last 128-bit lane is finished by the table walk from a zero state (the CRC is linear:
// that lane IS a 16-byte message with the CRC state of everything before it folded in): no Barrett reduction to get wrong.
__attribute__((target("pclmul,sse4.1"))) inline uint32_t crc32_pclmul(uint32_t crc, const unsigned char* p, size_t n)
{
    if (n < 64) return crc32(crc, p, n);
    const __m128i k12 = _mm_set_epi64x(0x1c6e41596ll, 0x154442bd4ll), k34 = _mm_set_epi64x(0xccaa009ell, 0x1751997d0ll);
#define ld(q) _mm_loadu_si128((const __m128i*)(q))                   // (macros: a lambda does not inherit the target attribute)
#define fold(x, k) _mm_xor_si128(_mm_clmulepi64_si128((x), (k), 0x00), _mm_clmulepi64_si128((x), (k), 0x11))
    __m128i x0 = _mm_xor_si128(ld(p), _mm_cvtsi32_si128((int)~crc)), x1 = ld(p + 16), x2 = ld(p + 32), x3 = ld(p + 48);
    p += 64; n -= 64;
    while (n >= 64) {
        x0 = _mm_xor_si128(fold(x0, k12), ld(p));
        x1 = _mm_xor_si128(fold(x1, k12), ld(p + 16));
        x2 = _mm_xor_si128(fold(x2, k12), ld(p + 32));
        x3 = _mm_xor_si128(fold(x3, k12), ld(p + 48));
        p += 64; n -= 64;
    }
    x0 = _mm_xor_si128(fold(x0, k34), x1);
    x0 = _mm_xor_si128(fold(x0, k34), x2);
    x0 = _mm_xor_si128(fold(x0, k34), x3);
    while (n >= 16) { x0 = _mm_xor_si128(fold(x0, k34), ld(p)); p += 16; n -= 16; }
    unsigned char lane[16];
    _mm_storeu_si128((__m128i*)lane, x0);
    const uint32_t (*t)[256] = crc_tables();
    uint32_t c = 0;
    for (int i = 0; i < 16; ++i) c = (c >> 8) ^ t[0][(c ^ lane[i]) & 0xff];
    while (n--) c = (c >> 8) ^ t[0][(c ^ *p++) & 0xff];
    return ~c;
#undef ld
#undef fold
}

inline uint32_t crc32_fast(uint32_t crc, const unsigned char* p, size_t n)
{
    static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    return clmul ? crc32_pclmul(crc, p, n) : crc32(crc, p, n);
}

inline uint32_t adler32_scalar(uint32_t adler, const unsigned char* p, size_t n)
{
    uint32_t s1 = adler & 0xffff, s2 = adler >> 16;
    while (n) {
        size_t k = n < 5552 ? n : 5552;                              // the largest run for which s2 cannot overflow 32 bits
        n -= k;
        while (k--) { s1 += *p++; s2 += s1; }
        s1 %= 65521u; s2 %= 65521u;
    }
    return (s2 << 16) | s1;
}

// 16 bytes per step: s1 += sum(b), s2 += 16 * s1_before + sum((16 - i) * b_i)
__attribute__((target("ssse3"))) inline uint32_t adler32_ssse3(uint32_t adler, const unsigned char* p, size_t n)
{
    uint32_t s1 = adler & 0xffff, s2 = adler >> 16;
    const __m128i weights = _mm_setr_epi8(16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1);
    const __m128i ones = _mm_set1_epi16(1), zero = _mm_setzero_si128();
    while (n >= 16) {
        size_t blocks = n / 16;
        if (blocks > 5552 / 16) blocks = 5552 / 16;
        n -= blocks * 16;
        __m128i v_s1 = zero, v_s2 = zero, v_ps = zero;               // v_ps: sum over the blocks of (s1 before the block - s1 at the start of the run)
        const uint64_t run = blocks;
        for (; blocks; --blocks, p += 16) {
            const __m128i b = _mm_loadu_si128((const __m128i*)p);
            v_ps = _mm_add_epi32(v_ps, v_s1);
            v_s1 = _mm_add_epi32(v_s1, _mm_sad_epu8(b, zero));
            v_s2 = _mm_add_epi32(v_s2, _mm_madd_epi16(_mm_maddubs_epi16(b, weights), ones));
        }
        auto hsum = [](__m128i v) -> uint64_t {
            uint32_t l[4];
            _mm_storeu_si128((__m128i*)l, v);
            return (uint64_t)l[0] + l[1] + l[2] + l[3];
        };
        const uint64_t t2 = (uint64_t)s2 + 16ull * (hsum(v_ps) + run * s1) + hsum(v_s2);
        s1 = (uint32_t)(((uint64_t)s1 + hsum(v_s1)) % 65521u);
        s2 = (uint32_t)(t2 % 65521u);
    }
    return adler32_scalar((s2 << 16) | s1, p, n);
}

inline uint32_t adler32(uint32_t adler, const unsigned char* p, size_t n)
{
    static const bool fast = __builtin_cpu_supports("ssse3");
    return fast ? adler32_ssse3(adler, p, n) : adler32_scalar(adler, p, n);
}

// ------------------------------------------------------------------ Huffman tables
// entry: bits 0-7 bits to consume (code + extra bits: ONE shift per symbol on the serial chain, the extra bits are read from the saved buffer) | bits 8-12 extra bits (lengths, distances) or index bits (second-level pointer) | bit 13 second-level pointer
//        | bit 14 end of block | bit 15 literal | bits 16-31 literal / base value / offset of the second-level table.  Bits 14 and 15 both: no such code.
constexpr uint32_t E_SUB = 1u << 13, E_EOB = 1u << 14, E_LIT = 1u << 15, E_INVALID = E_EOB | E_LIT, E_PAIR = 1u << 8;   // (E_PAIR: literal entries only)
constexpr int LIT_TB = 11, DIST_TB = 8;
constexpr int LIT_ENTRIES = (1 << LIT_TB) + 288 * 16, DIST_ENTRIES = (1 << DIST_TB) + 32 * 128, PRE_ENTRIES = 1 << 7;

inline uint32_t bit_reverse(uint32_t code, int len)
{
    uint32_t r = 0;
    for (int i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// kind 0: code-length alphabet (symbol = value), 1: literals / lengths, 2: distances.  lens[n] in 0..15.  Returns false for a set RFC 1951 / libz refuse.
inline bool build_table(const unsigned char* lens, int n, int kind, uint32_t* table, int tb)
{
    static const unsigned short len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const unsigned char len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const unsigned short dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const unsigned char dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    int count[16] = {0};
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    int maxlen = 15;
    while (maxlen > 0 && count[maxlen] == 0) --maxlen;
    const int primary = 1 << tb;
    for (int i = 0; i < primary; ++i) table[i] = E_INVALID;
    if (maxlen == 0) return kind == 2;                                // no code at all: legal for distances (a block of literals only), its use is an error
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;                                   // over-subscribed
    }
    if (left > 0 && (kind == 0 || maxlen != 1)) return false;         // incomplete (a single 1-bit code is the one exception)
    unsigned short order[320];
    int offs[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = offs[l] + count[l];
    for (int i = 0; i < n; ++i) if (lens[i]) order[offs[lens[i]]++] = (unsigned short)i;
    auto entry = [&](int sym, int nbits) -> uint32_t {
        if (kind == 0) return ((uint32_t)sym << 16) | E_LIT | (uint32_t)nbits;
        if (kind == 1) {
            if (sym < 256) return ((uint32_t)sym << 16) | E_LIT | (uint32_t)nbits;
            if (sym == 256) return E_EOB | (uint32_t)nbits;
            if (sym > 285) return E_INVALID | (uint32_t)nbits;         // 286, 287: in the fixed code, never legal in data
            return ((uint32_t)len_base[sym - 257] << 16) | ((uint32_t)len_extra[sym - 257] << 8) | (uint32_t)(nbits + len_extra[sym - 257]);
        }
        if (sym > 29) return E_INVALID | (uint32_t)nbits;
        return ((uint32_t)dist_base[sym] << 16) | ((uint32_t)dist_extra[sym] << 8) | (uint32_t)(nbits + dist_extra[sym]);
    };
    // first pass over the long codes: the longest code behind each first-level prefix
    unsigned char sub_bits[1 << LIT_TB];
    bool any_long = maxlen > tb;
    if (any_long) memset(sub_bits, 0, (size_t)primary);
    uint32_t code = 0;
    int k = 0;
    uint32_t first_code[16];
    for (int l = 1; l <= 15; ++l) { first_code[l] = code; code = (code + (uint32_t)count[l]) << 1; }
    if (any_long) {
        for (int l = tb + 1; l <= maxlen; ++l)
            for (int j = 0; j < count[l]; ++j) {
                const uint32_t c = first_code[l] + (uint32_t)j;
                const uint32_t prefix = bit_reverse(c >> (l - tb), tb);
                if (sub_bits[prefix] < l - tb) sub_bits[prefix] = (unsigned char)(l - tb);
            }
    }
    int next_sub = primary;
    for (int l = 1; l <= maxlen; ++l) {
        for (int j = 0; j < count[l]; ++j, ++k) {
            const uint32_t c = first_code[l] + (uint32_t)j;
            const int sym = order[k];
            if (l <= tb) {
                const uint32_t r = bit_reverse(c, l), e = entry(sym, l);
                for (uint32_t i = r; i < (uint32_t)primary; i += 1u << l) table[i] = e;
            } else {
                const uint32_t prefix = bit_reverse(c >> (l - tb), tb);
                const int sb = sub_bits[prefix];
                if (!(table[prefix] & E_SUB) || (table[prefix] & E_INVALID) == E_INVALID) {
                    table[prefix] = ((uint32_t)next_sub << 16) | E_SUB | ((uint32_t)sb << 8) | (uint32_t)tb;
                    for (int i = 0; i < (1 << sb); ++i) table[next_sub + i] = E_INVALID;
                    next_sub += 1 << sb;
                }
                const uint32_t base = table[prefix] >> 16;
                const uint32_t r = bit_reverse(c & ((1u << (l - tb)) - 1u), l - tb), e = entry(sym, l - tb);
                for (uint32_t i = r; i < (1u << sb); i += 1u << (l - tb)) table[base + i] = e;
            }
        }
    }
    if (kind == 1) {
        // two literals per look-up where both codes fit the first-level index (a photograph's stream is literals of 4-6 bits: the serial chain
        // look-up -> code length -> shift is what bounds the decoder, and a pair entry halves it).  Pair entry: E_PAIR, bits 16-23 the first
        // literal, bits 24-31 the second, bits 0-7 the bits of both codes.
        uint32_t single[1 << LIT_TB];
        memcpy(single, table, sizeof(single));
        for (uint32_t i = 0; i < (uint32_t)primary; ++i) {
            const uint32_t e1 = single[i];
            if ((e1 & (E_LIT | E_EOB | E_SUB)) != E_LIT) continue;
            const int l1 = (int)(e1 & 0xff);
            if (l1 >= tb) continue;
            const uint32_t e2 = single[i >> l1];                       // (the bits above tb - l1 of this index are unknown: e2 counts only if it does not read them)
            if ((e2 & (E_LIT | E_EOB | E_SUB)) != E_LIT || (int)(e2 & 0xff) > tb - l1) continue;
            table[i] = ((e2 >> 16) << 24) | (e1 & 0x00ff0000u) | E_LIT | E_PAIR | (uint32_t)(l1 + (int)(e2 & 0xff));
        }
    }
    return true;
}

struct Tables {
    uint32_t lit[LIT_ENTRIES];
    uint32_t dist[DIST_ENTRIES];
};

// the fixed code of RFC 1951 3.2.6, built once
inline const Tables& fixed_tables()
{
    static Tables t;
    static const bool once = [] {
        unsigned char l[288];
        for (int i = 0; i < 144; ++i) l[i] = 8;
        for (int i = 144; i < 256; ++i) l[i] = 9;
        for (int i = 256; i < 280; ++i) l[i] = 7;
        for (int i = 280; i < 288; ++i) l[i] = 8;
        build_table(l, 288, 1, t.lit, LIT_TB);
        unsigned char d[32];
        for (int i = 0; i < 32; ++i) d[i] = 5;
        build_table(d, 32, 2, t.dist, DIST_TB);
        return true;
    }();
    (void)once;
    return t;
}

// ------------------------------------------------------------------ the decoder
struct BitReader {
    const unsigned char* in;
    const unsigned char* in_end;
    uint64_t buf = 0;
    int cnt = 0;                                                       // valid bits in buf
    // careful refill: whole bytes while they exist
    inline void fill_careful() { while (cnt <= 56 && in < in_end) { buf |= (uint64_t)*in++ << cnt; cnt += 8; } }
    // fast refill (needs in + 8 <= in_end): at least 56 valid bits afterwards
    inline void fill_fast()
    {
        uint64_t v;
        memcpy(&v, in, 8);
        buf |= v << cnt;
        in += (63 - cnt) >> 3;
        cnt |= 56;
    }
    inline void drop(int n) { buf >>= n; cnt -= n; }
};

// One zlib stream in [in, in + n_in) -> at most n_out bytes at out.  *out_len = bytes produced, *in_used = bytes of the stream (header .. Adler-32).
// INF_OUTPUT_FULL: the stream holds more than n_out bytes; INF_TRUNCATED: the input ends inside the stream; INF_CORRUPT: everything else.
__attribute__((always_inline)) inline int zlib_core(const unsigned char* in, size_t n_in, unsigned char* out, size_t n_out, const unsigned char*& in_pos, unsigned char*& out_pos)
{
    in_pos = in; out_pos = out;
    if (n_in < 2) return INF_TRUNCATED;
    const unsigned cmf = in[0], flg = in[1];
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) return INF_CORRUPT;   // deflate, window <= 32 KiB, header check, no preset dictionary
    BitReader br;
    br.in = in + 2; br.in_end = in + n_in;
    unsigned char* const out0 = out;
    unsigned char* o = out;                                            // (a local: a byte store through a reference to the caller's pointer would have to re-load it)
#define OMNI_INF_RET(x) do { out_pos = o; return (x); } while (0)     // the caller sees how far the output got on every return path
    unsigned char* const out_end = out + n_out;
    Tables dyn;                                                        // ~45 KB: the decoder threads have megabytes of stack
    static const unsigned char pre_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    bool last = false;
    while (!last) {
        br.fill_careful();
        if (br.cnt < 3) OMNI_INF_RET(INF_TRUNCATED);
        last = br.buf & 1;
        const int type = (int)((br.buf >> 1) & 3);
        br.drop(3);
        const Tables* T;
        if (type == 0) {                                               // stored: LEN, ~LEN at the next byte boundary, then LEN bytes
            br.drop(br.cnt & 7);
            br.fill_careful();
            if (br.cnt < 32) OMNI_INF_RET(INF_TRUNCATED);
            const unsigned len = (unsigned)(br.buf & 0xffff), nlen = (unsigned)((br.buf >> 16) & 0xffff);
            br.drop(32);
            if ((len ^ nlen) != 0xffff) OMNI_INF_RET(INF_CORRUPT);
            // the bit buffer holds whole bytes here: hand them back
            br.in -= br.cnt >> 3; br.buf = 0; br.cnt = 0;
            if ((size_t)(br.in_end - br.in) < len) OMNI_INF_RET(INF_TRUNCATED);
            if ((size_t)(out_end - o) < len) OMNI_INF_RET(INF_OUTPUT_FULL);
            memcpy(o, br.in, len);
            o += len; br.in += len;
            continue;
        } else if (type == 1) {
            T = &fixed_tables();
        } else if (type == 2) {
            br.fill_careful();
            if (br.cnt < 14) OMNI_INF_RET(INF_TRUNCATED);
            const int hlit = (int)(br.buf & 31) + 257, hdist = (int)((br.buf >> 5) & 31) + 1, hclen = (int)((br.buf >> 10) & 15) + 4;
            br.drop(14);
            if (hlit > 286 || hdist > 30) OMNI_INF_RET(INF_CORRUPT);
            unsigned char pl[19] = {0};
            for (int i = 0; i < hclen; ++i) {
                br.fill_careful();
                if (br.cnt < 3) OMNI_INF_RET(INF_TRUNCATED);
                pl[pre_order[i]] = (unsigned char)(br.buf & 7);
                br.drop(3);
            }
            uint32_t pre[PRE_ENTRIES];
            if (!build_table(pl, 19, 0, pre, 7)) OMNI_INF_RET(INF_CORRUPT);
            unsigned char lens[288 + 32];
            int i = 0;
            while (i < hlit + hdist) {
                br.fill_careful();
                const uint32_t e = pre[br.buf & 127];
                if ((e & E_INVALID) == E_INVALID) OMNI_INF_RET(br.cnt < 7 ? INF_TRUNCATED : INF_CORRUPT);
                const int nb = (int)(e & 0xff), sym = (int)(e >> 16);
                if (br.cnt < nb) OMNI_INF_RET(INF_TRUNCATED);
                br.drop(nb);
                if (sym < 16) { lens[i++] = (unsigned char)sym; continue; }
                int rep, val = 0, xb;
                if (sym == 16) { if (i == 0) OMNI_INF_RET(INF_CORRUPT); val = lens[i - 1]; xb = 2; rep = 3; }
                else if (sym == 17) { xb = 3; rep = 3; }
                else { xb = 7; rep = 11; }
                if (br.cnt < xb) { br.fill_careful(); if (br.cnt < xb) OMNI_INF_RET(INF_TRUNCATED); }
                rep += (int)(br.buf & ((1u << xb) - 1u));
                br.drop(xb);
                if (i + rep > hlit + hdist) OMNI_INF_RET(INF_CORRUPT);
                while (rep--) lens[i++] = (unsigned char)val;
            }
            if (lens[256] == 0) OMNI_INF_RET(INF_CORRUPT);                    // no end-of-block code
            if (!build_table(lens, hlit, 1, dyn.lit, LIT_TB)) OMNI_INF_RET(INF_CORRUPT);
            if (!build_table(lens + hlit, hdist, 2, dyn.dist, DIST_TB)) OMNI_INF_RET(INF_CORRUPT);
            T = &dyn;
        } else OMNI_INF_RET(INF_CORRUPT);

        const uint32_t* const lt = T->lit;
        const uint32_t* const dt = T->dist;
        constexpr uint32_t LMASK = (1u << LIT_TB) - 1u, DMASK = (1u << DIST_TB) - 1u;
        bool eob = false;
        // ---- fast loop: 16 input bytes and 6 literals + a 258-byte match + 16 bytes of copy overshoot always available
        while ((size_t)(br.in_end - br.in) >= 16 && (size_t)(out_end - o) >= 6 + 258 + 16) {
            br.fill_fast();
            uint32_t e = lt[br.buf & LMASK];
            // (a literal entry stores two bytes whatever it holds — the second is overwritten by the next symbol unless E_PAIR advances past it)
#define OMNI_INF_EMIT(e) do { const uint16_t two = (uint16_t)((e) >> 16); memcpy(o, &two, 2); o += 1 + (((e) >> 8) & 1); br.drop((int)((e) & 0xff)); } while (0)
            if ((e & (E_LIT | E_EOB)) == E_LIT) {
                OMNI_INF_EMIT(e);
                e = lt[br.buf & LMASK];
                if ((e & (E_LIT | E_EOB)) == E_LIT) {
                    OMNI_INF_EMIT(e);
                    e = lt[br.buf & LMASK];
                    if ((e & (E_LIT | E_EOB)) == E_LIT) {
                        OMNI_INF_EMIT(e);
                        continue;
                    }
                }
                br.fill_fast();
            }
            if (e & E_SUB) {
                br.drop(LIT_TB);
                e = lt[(e >> 16) + (uint32_t)(br.buf & ((1u << ((e >> 8) & 31)) - 1u))];
            }
            if (e & E_LIT) {
                if (e & E_EOB) OMNI_INF_RET(INF_CORRUPT);
                OMNI_INF_EMIT(e);                                      // (second-level literal entries are never pairs: one byte kept)
                continue;
            }
#undef OMNI_INF_EMIT
            if (e & E_EOB) { br.drop((int)(e & 0xff)); eob = true; break; }
            const uint64_t sv = br.buf;
            const int tot = (int)(e & 0xff), xl = (int)((e >> 8) & 31);
            br.drop(tot);
            uint32_t d = dt[br.buf & DMASK];
            const unsigned len = (e >> 16) + (unsigned)((sv >> (tot - xl)) & ((1u << xl) - 1u));
            if (d & E_SUB) {
                br.drop(DIST_TB);
                d = dt[(d >> 16) + (uint32_t)(br.buf & ((1u << ((d >> 8) & 31)) - 1u))];
            }
            if (d & E_INVALID) OMNI_INF_RET(INF_CORRUPT);
            const uint64_t sv2 = br.buf;
            const int tot2 = (int)(d & 0xff), xd = (int)((d >> 8) & 31);
            br.drop(tot2);
            const size_t dist = (d >> 16) + (size_t)((sv2 >> (tot2 - xd)) & ((1u << xd) - 1u));
            if (dist > (size_t)(o - out0)) OMNI_INF_RET(INF_CORRUPT);
            const unsigned char* s = o - dist;
            unsigned char* const oe = o + len;
            if (dist >= 16) {
                do { __m128i v = _mm_loadu_si128((const __m128i*)s); _mm_storeu_si128((__m128i*)o, v); s += 16; o += 16; } while (o < oe);
            } else if (dist == 1) {
                memset(o, *s, len);
            } else if (dist >= 8) {
                do { uint64_t v; memcpy(&v, s, 8); memcpy(o, &v, 8); s += 8; o += 8; } while (o < oe);
            } else {
                while (o < oe) *o++ = *s++;
            }
            o = oe;
        }
        // ---- careful loop: one symbol per trip, every bit and byte counted
        while (!eob) {
            br.fill_careful();
            uint64_t b = br.buf;
            uint32_t e = lt[b & LMASK];
            int used = 0;
            if (e & E_SUB) {
                used = LIT_TB;
                e = lt[(e >> 16) + (uint32_t)((b >> LIT_TB) & ((1u << ((e >> 8) & 31)) - 1u))];
            }
            if ((e & E_INVALID) == E_INVALID) OMNI_INF_RET(br.cnt < 15 ? INF_TRUNCATED : INF_CORRUPT);
            used += (int)(e & 0xff);
            if (used > br.cnt) OMNI_INF_RET(INF_TRUNCATED);
            br.drop(used);
            if (e & E_LIT) {
                if (o >= out_end) OMNI_INF_RET(INF_OUTPUT_FULL);
                *o++ = (unsigned char)(e >> 16);
                if (e & E_PAIR) {
                    if (o >= out_end) OMNI_INF_RET(INF_OUTPUT_FULL);
                    *o++ = (unsigned char)(e >> 24);
                }
                continue;
            }
            if (e & E_EOB) { eob = true; break; }
            const int xl = (int)((e >> 8) & 31);
            const unsigned len = (e >> 16) + (unsigned)((b >> (used - xl)) & ((1u << xl) - 1u));
            br.fill_careful();
            b = br.buf;
            uint32_t d = dt[b & DMASK];
            used = 0;
            if (d & E_SUB) {
                used = DIST_TB;
                d = dt[(d >> 16) + (uint32_t)((b >> DIST_TB) & ((1u << ((d >> 8) & 31)) - 1u))];
            }
            if (d & E_INVALID) OMNI_INF_RET(br.cnt < 15 ? INF_TRUNCATED : INF_CORRUPT);
            used += (int)(d & 0xff);
            if (used > br.cnt) OMNI_INF_RET(INF_TRUNCATED);
            br.drop(used);
            const int xd = (int)((d >> 8) & 31);
            const size_t dist = (d >> 16) + (size_t)((b >> (used - xd)) & ((1u << xd) - 1u));
            if (dist > (size_t)(o - out0)) OMNI_INF_RET(INF_CORRUPT);
            if ((size_t)(out_end - o) < len) OMNI_INF_RET(INF_OUTPUT_FULL);
            const unsigned char* s = o - dist;
            for (unsigned i = 0; i < len; ++i) o[i] = s[i];
            o += len;
        }
    }
    // Adler-32 of the output, big-endian, at the next byte boundary
    br.drop(br.cnt & 7);
    br.in -= br.cnt >> 3; br.buf = 0; br.cnt = 0;                      // (whole bytes only: hand them back)
    in_pos = br.in;
    if ((size_t)(br.in_end - br.in) < 4) OMNI_INF_RET(INF_TRUNCATED);
    const uint32_t want = ((uint32_t)br.in[0] << 24) | ((uint32_t)br.in[1] << 16) | ((uint32_t)br.in[2] << 8) | (uint32_t)br.in[3];
    in_pos = br.in + 4;
    if (adler32(1u, out0, (size_t)(o - out0)) != want) OMNI_INF_RET(INF_CORRUPT);
    OMNI_INF_RET(INF_OK);
}

// the same decoder compiled twice: with BMI2 the variable shifts of the serial chain are single-uop shrx / shlx (5 % on a photograph's stream)
inline int zlib_core_plain(const unsigned char* in, size_t n_in, unsigned char* out, size_t n_out, const unsigned char*& in_pos, unsigned char*& out_pos)
{ return zlib_core(in, n_in, out, n_out, in_pos, out_pos); }
__attribute__((target("bmi2"))) inline int zlib_core_bmi2(const unsigned char* in, size_t n_in, unsigned char* out, size_t n_out, const unsigned char*& in_pos, unsigned char*& out_pos)
{ return zlib_core(in, n_in, out, n_out, in_pos, out_pos); }

inline int zlib_decompress(const unsigned char* in, size_t n_in, unsigned char* out, size_t n_out, size_t* in_used, size_t* out_len)
{
    const unsigned char* ip = in;
    unsigned char* op = out;
    static const bool bmi2 = __builtin_cpu_supports("bmi2");
    const int rc = bmi2 ? zlib_core_bmi2(in, n_in, out, n_out, ip, op) : zlib_core_plain(in, n_in, out, n_out, ip, op);
    if (in_used) *in_used = (size_t)(ip - in);
    if (out_len) *out_len = (size_t)(op - out);
    return rc;
}

}  // namespace omni_inflate
