// Rational reconstruction for the exact certificate (certify.hip), host only: given a (mod M), the fraction n/d with
// |n|, d <= sqrt(M/2) (Wang's bound) and n = a d (mod M), if there is one.  It is the extended Euclidean sequence of (M, a),
// stopped at the first remainder inside the bound.  On the 4000-bit moduli of a certificate the plain sequence is ~1200 big
// divisions (0.4 ms, most of the host time of a lifting once the rest was trimmed); Lehmer's device does the same sequence in
// batches: the quotients are simulated on the leading 126 bits (exactly, by Knuth's two-sided quotient test, TAOCP 4.5.2 Alg. L)
// and ~60 bits of progress are applied to the big numbers with one 2 x 2 integer matrix.  Every remainder produced is a member of
// the plain sequence, and a batch that would step over the stopping point is discarded for a plain step, so the result is
// IDENTICAL to the plain algorithm's (tests/test_bigint.py compares the two and a Python restatement).
#pragma once
#include "bigint.hpp"

namespace relp {

// |v| <= sqrt(M / 2), decided by bit lengths: v < 2^k with 2k <= bits(M) - 2 gives v^2 < 2^(bits(M) - 2) <= M / 2.  (Sufficient,
// not necessary: at most two bits stricter than Wang's bound, and free -- the exact test is a big multiplication per call.)
inline bool within_wang_bound(const BigInt& v, const BigInt& M) { return 2 * v.bits() + 2 <= M.bits(); }

namespace detail {
// the 126 bits of |v| from bit position `shift` upwards (v has at most shift + 126 bits)
inline unsigned __int128 window_bits(const BigInt& v, size_t shift) {
    unsigned __int128 w = 0;
    const size_t word = shift / 32, bit = shift % 32;
    for (int k = 4; k >= 0; --k) {  // five limbs cover 126 bits at any alignment
        const size_t idx = word + (size_t)k;
        if (idx >= v.mag.size()) continue;
        const unsigned __int128 limb = v.mag[idx];
        const int up = 32 * k - (int)bit;
        if (up >= 0) {
            if (up < 128) w |= limb << up;
        } else {
            w |= limb >> (-up);
        }
    }
    return w;
}
inline BigInt small_times(__int128 c, const BigInt& v) { return BigInt::from_i128(c) * v; }
}  // namespace detail

// `prime`: non-zero when M is a power of this prime (the p-adic liftings).  gcd(remainder, cofactor) divides M at every step of the
// sequence (r = s M + t a with gcd(s, t) = 1), so with M = p^K the fraction is in lowest terms unless p divides the cofactor -- and
// then there is no fraction with an invertible denominator: one division by a word instead of a 1000-step big gcd.
inline bool rational_reconstruct(const BigInt& a, const BigInt& M, BigInt& n, BigInt& d, bool lehmer = true, uint32_t prime = 0) {
    BigInt r0 = M, r1 = a % M;
    if (r1.sign() < 0) r1 = r1 + M;
    BigInt t0(0), t1(1);
    auto too_big = [&](const BigInt& r) { return !within_wang_bound(r, M); };
    const __int128 cap = (__int128)1 << 62;
    while (too_big(r1)) {
        bool stepped = false;
        if (lehmer && r0.bits() > 130 && r1.bits() + 64 > r0.bits()) {
            const size_t shift = r0.bits() - 126;
            __int128 x = (__int128)detail::window_bits(r0, shift), y = (__int128)detail::window_bits(r1, shift);
            __int128 A = 1, B = 0, C = 0, D = 1;
            int steps = 0;
            while (true) {
                if (y + C == 0 || y + D == 0) break;
                const __int128 q = (x + A) / (y + C);
                if (q != (x + B) / (y + D) || q >= cap) break;
                const __int128 nC = A - q * C, nD = B - q * D;
                if (nC >= cap || nC <= -cap || nD >= cap || nD <= -cap) break;
                A = C;
                C = nC;
                B = D;
                D = nD;
                const __int128 ny = x - q * y;
                x = y;
                y = ny;
                ++steps;
            }
            if (steps > 0) {
                BigInt nr0 = detail::small_times(A, r0) + detail::small_times(B, r1);
                if (too_big(nr0)) {  // (else the batch steps over the first remainder inside the bound: plain steps from here)
                    BigInt nr1 = detail::small_times(C, r0) + detail::small_times(D, r1);
                    BigInt nt0 = detail::small_times(A, t0) + detail::small_times(B, t1);
                    BigInt nt1 = detail::small_times(C, t0) + detail::small_times(D, t1);
                    r0 = std::move(nr0);
                    r1 = std::move(nr1);
                    t0 = std::move(nt0);
                    t1 = std::move(nt1);
                    stepped = true;
                }
            }
        }
        if (!stepped) {
            BigInt q, rem;
            BigInt::divmod(r0, r1, q, rem);
            BigInt t2 = t0 - q * t1;
            r0 = std::move(r1);
            r1 = std::move(rem);
            t0 = std::move(t1);
            t1 = std::move(t2);
        }
    }
    if (t1.is_zero() || too_big(t1.abs())) return false;
    n = t1.sign() < 0 ? -r1 : r1;
    d = t1.abs();
    if (prime != 0) {
        BigInt copy = d;
        return copy.div_small(prime) != 0;
    }
    BigInt g = BigInt::gcd(n, d);
    if (!(g == BigInt(1))) {
        if (g.is_zero()) return false;
        n = n / g;
        d = d / g;
    }
    return true;
}

}  // namespace relp
