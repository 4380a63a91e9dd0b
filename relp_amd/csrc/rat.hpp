// Exact rational over 128-bit integers for the HOST side of the drop-in boundary.
//
// The reference reads MPS numbers into `Rational64` (relp-num; /root/reference/src/io/mps/number/parse.rs:46-65)
// and standardises the program with exact arithmetic (general_form/mod.rs:325-332).  The product keeps the
// same exactness up to the device upload: every input coefficient is a normalised `Rat` (num/den, den > 0);
// an operation whose result does not fit 128 bits throws `RatOverflow` (surfaced as RELP_STATUS_OVERFLOW).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>

namespace relp {

using i128 = __int128;
using u128 = unsigned __int128;

struct RatOverflow : std::runtime_error {
    RatOverflow() : std::runtime_error("128-bit rational overflow") {}
};

inline i128 iabs128(i128 x) { return x < 0 ? -x : x; }

inline i128 gcd128(i128 a, i128 b) {
    a = iabs128(a);
    b = iabs128(b);
    while (b != 0) {
        i128 t = a % b;
        a = b;
        b = t;
    }
    return a;
}

inline i128 mul_checked(i128 a, i128 b) {
    i128 r;
    if (__builtin_mul_overflow(a, b, &r)) throw RatOverflow();
    return r;
}
inline i128 add_checked(i128 a, i128 b) {
    i128 r;
    if (__builtin_add_overflow(a, b, &r)) throw RatOverflow();
    return r;
}

struct Rat {
    i128 n = 0;
    i128 d = 1;

    Rat() = default;
    Rat(long long v) : n(v), d(1) {}
    Rat(i128 num, i128 den) : n(num), d(den) { normalise(); }

    void normalise() {
        if (d == 0) throw std::runtime_error("zero denominator");
        if (d < 0) { n = -n; d = -d; }
        i128 g = gcd128(n, d);
        if (g > 1) { n /= g; d /= g; }
        if (n == 0) d = 1;
    }
    bool is_zero() const { return n == 0; }
    int sign() const { return n > 0 ? 1 : (n < 0 ? -1 : 0); }
    double to_double() const { return (double)n / (double)d; }
    Rat operator-() const { Rat r; r.n = -n; r.d = d; return r; }
};

inline Rat operator+(const Rat& a, const Rat& b) {
    i128 g = gcd128(a.d, b.d);
    i128 bd = b.d / g;
    return Rat(add_checked(mul_checked(a.n, bd), mul_checked(b.n, a.d / g)), mul_checked(a.d, bd));
}
inline Rat operator-(const Rat& a, const Rat& b) { return a + (-b); }
inline Rat operator*(const Rat& a, const Rat& b) {
    i128 g1 = gcd128(a.n, b.d), g2 = gcd128(b.n, a.d);
    if (g1 == 0) g1 = 1;
    if (g2 == 0) g2 = 1;
    return Rat(mul_checked(a.n / g1, b.n / g2), mul_checked(a.d / g2, b.d / g1));
}
inline Rat operator/(const Rat& a, const Rat& b) {
    if (b.n == 0) throw std::runtime_error("division by zero");
    Rat inv;
    inv.n = b.n < 0 ? -b.d : b.d;
    inv.d = iabs128(b.n);
    return a * inv;
}
inline int cmp(const Rat& a, const Rat& b) { return (a - b).sign(); }
inline bool operator==(const Rat& a, const Rat& b) { return a.n == b.n && a.d == b.d; }
inline bool operator!=(const Rat& a, const Rat& b) { return !(a == b); }
inline bool operator<(const Rat& a, const Rat& b) { return cmp(a, b) < 0; }
inline bool operator>(const Rat& a, const Rat& b) { return cmp(a, b) > 0; }
inline bool operator<=(const Rat& a, const Rat& b) { return cmp(a, b) <= 0; }
inline bool operator>=(const Rat& a, const Rat& b) { return cmp(a, b) >= 0; }

inline std::string to_string128(i128 v) {
    if (v == 0) return "0";
    bool neg = v < 0;
    u128 u = neg ? (u128)(-(v + 1)) + 1 : (u128)v;
    std::string s;
    while (u != 0) {
        s.push_back((char)('0' + (int)(u % 10)));
        u /= 10;
    }
    if (neg) s.push_back('-');
    return std::string(s.rbegin(), s.rend());
}
inline std::string to_string(const Rat& r) { return to_string128(r.n) + "/" + to_string128(r.d); }

}  // namespace relp
