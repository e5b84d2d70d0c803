// sj_bigdec.h -- the last step of decimal -> binary64 for literals of more than 19 significant digits whose two 19-digit
// neighbours round to DIFFERENT (adjacent) doubles (sj_number.h): which of the two is the literal's correctly rounded value?
//
// What DoubleParser.slowlyParseDouble decides digit by digit (/root/reference/src/main/java/org/simdjson/DoubleParser.java:
// 216-330, Nigel Tao's "simple decimal" conversion over at most 800 digits + a truncation flag).  Here: an exact comparison.
// The candidates are adjacent doubles lo < hi, so the literal x rounds to lo iff x < m, to hi iff x > m, and to the one with
// the even significand iff x == m, where m = (lo + hi) / 2 = (2 mant + 1) * 2^(e - 1) is a dyadic rational.  With
// x = D * 10^k (D = up to 800 significant digits as an integer, the digits beyond only as "and something more"), the
// comparison D * 10^k ? M * 2^E is one of two big integers after moving every negative power to the other side:
//   D * 5^max(k,0) * 2^a   ?   M * 5^max(-k,0) * 2^b        (a, b >= 0, their common part cancelled)
// At most 84 + 82 words of 32 bits on either side.  Plain loops over word arrays the caller provides (the device keeps
// them in LDS: one lane of a wave works, this is the rarest path of the engine; the host walker keeps them on its stack).
// Compiles as plain C++ and as HIP device code.
#pragma once
#include <stdint.h>

#include "sj_number.h"

namespace sjmi {

constexpr int SJ_BIG_WORDS = 200;
constexpr uint32_t SJ_BIG_MAX_DIGITS = 800;  // DoubleParser.java: SLOW_PATH_MAX_DIGIT_COUNT

struct SjBig {
    uint32_t* w;  // little endian
    int n;        // words in use (no leading zero word; 0 = the value 0)
};

SJN_DEV inline void sj_big_mul_add(SjBig& x, uint32_t m, uint32_t add) {  // x = x * m + add
    unsigned long long carry = add;
    for (int i = 0; i < x.n; ++i) {
        const unsigned long long t = (unsigned long long)x.w[i] * m + carry;
        x.w[i] = (uint32_t)t;
        carry = t >> 32;
    }
    if (carry && x.n < SJ_BIG_WORDS) x.w[x.n++] = (uint32_t)carry;
}
SJN_DEV inline void sj_big_mul_pow5(SjBig& x, uint32_t e) {
    while (e >= 13) {
        sj_big_mul_add(x, 1220703125u, 0);  // 5^13
        e -= 13;
    }
    uint32_t m = 1;
    for (uint32_t i = 0; i < e; ++i) m *= 5;
    if (m > 1) sj_big_mul_add(x, m, 0);
}
SJN_DEV inline void sj_big_shl(SjBig& x, uint32_t bits) {
    if (x.n == 0 || bits == 0) return;
    const int ws = (int)(bits >> 5);
    const uint32_t bs = bits & 31u;
    int n = x.n + ws + 1;
    if (n > SJ_BIG_WORDS) n = SJ_BIG_WORDS;
    for (int i = n - 1; i >= 0; --i) {
        const int s = i - ws;
        const uint32_t lo = (s >= 0 && s < x.n) ? x.w[s] : 0u, below = (s - 1 >= 0 && s - 1 < x.n) ? x.w[s - 1] : 0u;
        x.w[i] = bs ? (lo << bs) | (below >> (32u - bs)) : lo;
    }
    while (n > 0 && x.w[n - 1] == 0) --n;
    x.n = n;
}
SJN_DEV inline int sj_big_cmp(const SjBig& a, const SjBig& b) {
    if (a.n != b.n) return a.n < b.n ? -1 : 1;
    for (int i = a.n - 1; i >= 0; --i)
        if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
    return 0;
}

// B(q) = the byte at q; the literal (sign excluded) starts at p and was accepted by sj_scan_number.  lo_mag = the bits of the
// LOWER candidate's magnitude (sign bit clear); the upper one is lo_mag + 1 (also across a power of two, into the
// subnormals' neighbour and into infinity).  wa / wb: SJ_BIG_WORDS words each.  -> the bits of the magnitude.
template <class ByteAt>
SJN_DEV inline unsigned long long sj_decide_double(ByteAt&& B, uint32_t p, unsigned long long lo_mag, uint32_t* wa, uint32_t* wb) {
    SjBig lhs{wa, 0}, rhs{wb, 0};
    // ---- the literal as D * 10^k ----
    long long k = 0;
    uint32_t nd = 0, chunk = 0, cm = 1;
    bool seen = false, more = false;
    auto digit = [&](uint32_t d, bool frac) {
        if (!seen && d == 0) {
            k -= frac ? 1 : 0;
            return;
        }
        seen = true;
        if (nd < SJ_BIG_MAX_DIGITS) {
            chunk = chunk * 10 + d;
            cm *= 10;
            ++nd;
            k -= frac ? 1 : 0;
            if (cm == 1000000000u) {
                sj_big_mul_add(lhs, cm, chunk);
                chunk = 0;
                cm = 1;
            }
        } else {
            more |= d != 0;
            k += frac ? 0 : 1;
        }
    };
    uint32_t c = B(p);
    while (c - '0' <= 9u) {
        digit(c - '0', false);
        c = B(++p);
    }
    if (c == '.') {
        c = B(++p);
        while (c - '0' <= 9u) {
            digit(c - '0', true);
            c = B(++p);
        }
    }
    if (cm > 1) sj_big_mul_add(lhs, cm, chunk);
    if (c == 'e' || c == 'E') {
        c = B(++p);
        const bool eneg = c == '-';
        if (c == '-' || c == '+') c = B(++p);
        long long e = 0;
        while (c - '0' <= 9u) {
            if (e < 100000000000000000LL) e = e * 10 + (long long)(c - '0');  // (the same 18-digit saturation as sj_scan_number: the two must agree on the decimal exponent)
            c = B(++p);
        }
        k += eneg ? -e : e;
    }
    // (the candidates are finite or infinity's neighbour: the literal is within [10^-343, 10^310), so |k| is bounded; a value
    //  that is not could not have produced two different candidates)
    if (k > 400) k = 400;
    if (k < -1300) k = -1300;
    // ---- the midpoint M * 2^E between lo and lo's successor ----
    const unsigned long long frac = lo_mag & ((1ull << 52) - 1ull);
    const int ef = (int)(lo_mag >> 52) & 0x7FF;
    const unsigned long long mant = ef ? (frac | (1ull << 52)) : frac;
    const long long E = (ef ? (long long)ef - 1075 : -1074) - 1;
    const unsigned long long M = 2 * mant + 1;
    rhs.w[0] = (uint32_t)M;
    rhs.w[1] = (uint32_t)(M >> 32);
    rhs.n = rhs.w[1] ? 2 : 1;
    long long a2 = 0, b2 = 0;
    if (k >= 0) {
        sj_big_mul_pow5(lhs, (uint32_t)k);
        a2 += k;
    } else {
        sj_big_mul_pow5(rhs, (uint32_t)(-k));
        b2 += -k;
    }
    if (E >= 0) b2 += E;
    else a2 += -E;
    const long long common = a2 < b2 ? a2 : b2;
    sj_big_shl(lhs, (uint32_t)(a2 - common));
    sj_big_shl(rhs, (uint32_t)(b2 - common));
    int cmp = sj_big_cmp(lhs, rhs);
    if (cmp == 0 && more) cmp = 1;  // the digits that were left out: "and something more"
    if (cmp == 0) cmp = (mant & 1ull) ? 1 : -1;  // exactly half way: to the even significand
    return cmp > 0 ? lo_mag + 1 : lo_mag;
}

}  // namespace sjmi
