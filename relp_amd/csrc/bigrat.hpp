// Arbitrary-precision rationals on the HOST, for the presolve only: domain propagation multiplies and divides bounds
// repeatedly and leaves the 128-bit range of `Rat` on problems such as 25FV47 (numerators of thousands of bits).
// Always normalised (gcd 1, positive denominator).  Values whose parts fit 62 bits stay inline and use __int128
// intermediates; the rest uses `Nat`: 64-bit limbs, schoolbook multiply, Knuth D division, Lehmer gcd, and the
// gcd-saving forms of Knuth 4.5.1 for + and *.  (The CPU oracle carries its own copy of this arithmetic in
// oracle/cpp/rational.hpp; the product does not depend on anything under oracle/.)  tests/test_bigint.py fuzzes it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rat.hpp"

namespace relp {



inline uint64_t gcd64(uint64_t a, uint64_t b) {
    while (b != 0) {
        uint64_t t = a % b;
        a = b;
        b = t;
    }
    return a;
}
inline u128 gcd128u(u128 a, u128 b) {
    while (b != 0) {
        if ((a >> 64) == 0 && (b >> 64) == 0) return gcd64((uint64_t)a, (uint64_t)b);
        u128 t = a % b;
        a = b;
        b = t;
    }
    return a;
}
// (hi:lo) / d with hi < d
inline uint64_t div128(uint64_t hi, uint64_t lo, uint64_t d, uint64_t& rem) {
    uint64_t q;
    __asm__("divq %4" : "=a"(q), "=d"(rem) : "a"(lo), "d"(hi), "r"(d) : "cc");
    return q;
}

// ---- natural numbers -------------------------------------------------------------------------------------------------
struct Nat {
    std::vector<uint64_t> w;  // little endian, no leading zero limbs; empty = 0

    Nat() {}
    explicit Nat(u128 v) {
        if ((uint64_t)v || (v >> 64)) w.push_back((uint64_t)v);
        if (v >> 64) w.push_back((uint64_t)(v >> 64));
    }
    bool zero() const { return w.empty(); }
    bool is_one() const { return w.size() == 1 && w[0] == 1; }
    size_t size() const { return w.size(); }
    void trim() {
        while (!w.empty() && w.back() == 0) w.pop_back();
    }
    size_t bits() const { return w.empty() ? 0 : (w.size() - 1) * 64 + (64 - __builtin_clzll(w.back())); }
    bool fits62() const { return w.empty() || (w.size() == 1 && (w[0] >> 62) == 0); }
    uint64_t low() const { return w.empty() ? 0 : w[0]; }

    static int cmp(const Nat& a, const Nat& b) {
        if (a.w.size() != b.w.size()) return a.w.size() < b.w.size() ? -1 : 1;
        for (size_t i = a.w.size(); i-- > 0;)
            if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1;
        return 0;
    }
    static Nat add(const Nat& a, const Nat& b) {
        const Nat& x = a.size() >= b.size() ? a : b;
        const Nat& y = a.size() >= b.size() ? b : a;
        Nat r;
        r.w.resize(x.size() + 1);
        uint64_t carry = 0;
        for (size_t i = 0; i < x.size(); ++i) {
            u128 s = (u128)x.w[i] + (i < y.size() ? y.w[i] : 0) + carry;
            r.w[i] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
        r.w[x.size()] = carry;
        r.trim();
        return r;
    }
    static Nat sub(const Nat& a, const Nat& b) {  // a >= b
        Nat r;
        r.w.resize(a.size());
        uint64_t borrow = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            uint64_t bi = i < b.size() ? b.w[i] : 0;
            u128 d = (u128)a.w[i] - bi - borrow;
            r.w[i] = (uint64_t)d;
            borrow = (uint64_t)(d >> 64) & 1;
        }
        r.trim();
        return r;
    }
    static Nat mul(const Nat& a, const Nat& b) {
        Nat r;
        if (a.zero() || b.zero()) return r;
        if (b.is_one()) return a;
        if (a.is_one()) return b;
        r.w.assign(a.size() + b.size(), 0);
        for (size_t i = 0; i < a.size(); ++i) {
            uint64_t carry = 0;
            const uint64_t ai = a.w[i];
            for (size_t j = 0; j < b.size(); ++j) {
                u128 t = (u128)ai * b.w[j] + r.w[i + j] + carry;
                r.w[i + j] = (uint64_t)t;
                carry = (uint64_t)(t >> 64);
            }
            r.w[i + b.size()] = carry;
        }
        r.trim();
        return r;
    }
    void mul_add_small(uint64_t m, uint64_t a) {  // this = this * m + a
        uint64_t carry = a;
        for (size_t i = 0; i < w.size(); ++i) {
            u128 t = (u128)w[i] * m + carry;
            w[i] = (uint64_t)t;
            carry = (uint64_t)(t >> 64);
        }
        if (carry) w.push_back(carry);
    }
    uint64_t div_small(uint64_t d) {  // this /= d, returns the remainder
        uint64_t rem = 0;
        for (size_t i = w.size(); i-- > 0;) w[i] = div128(rem, w[i], d, rem);
        trim();
        return rem;
    }
    uint64_t mod_small(uint64_t d) const {
        uint64_t rem = 0, q;
        for (size_t i = w.size(); i-- > 0;) {
            q = div128(rem, w[i], d, rem);
            (void)q;
        }
        return rem;
    }
    // Knuth 4.3.1 algorithm D; q or r may be null
    static void divmod(const Nat& a, const Nat& b, Nat* q, Nat* r) {
        if (b.zero()) throw std::runtime_error("Nat division by zero");
        if (cmp(a, b) < 0) {
            if (q) *q = Nat();
            if (r) *r = a;
            return;
        }
        if (b.size() == 1) {
            Nat t = a;
            uint64_t rem = t.div_small(b.w[0]);
            if (q) *q = t;
            if (r) *r = Nat((u128)rem);
            return;
        }
        const int s = __builtin_clzll(b.w.back());
        const size_t n = b.size(), m = a.size() - n;
        std::vector<uint64_t> v(n), u(a.size() + 1);
        for (size_t i = n; i-- > 0;) v[i] = s ? (b.w[i] << s) | (i ? b.w[i - 1] >> (64 - s) : 0) : b.w[i];
        u[a.size()] = s ? a.w.back() >> (64 - s) : 0;
        for (size_t i = a.size(); i-- > 0;) u[i] = s ? (a.w[i] << s) | (i ? a.w[i - 1] >> (64 - s) : 0) : a.w[i];
        std::vector<uint64_t> quotient(m + 1, 0);
        for (size_t j = m + 1; j-- > 0;) {
            uint64_t qhat;
            u128 rhat;
            if (u[j + n] >= v[n - 1]) {
                qhat = ~(uint64_t)0;
                rhat = (u128)u[j + n - 1] + v[n - 1] + (((u128)(u[j + n] - v[n - 1])) << 64);
            } else {
                uint64_t rem;
                qhat = div128(u[j + n], u[j + n - 1], v[n - 1], rem);
                rhat = rem;
            }
            while ((rhat >> 64) == 0 && (u128)qhat * v[n - 2] > ((rhat << 64) | u[j + n - 2])) {
                --qhat;
                rhat += v[n - 1];
            }
            uint64_t borrow = 0, carry = 0;
            for (size_t i = 0; i < n; ++i) {
                u128 p = (u128)qhat * v[i] + carry;
                carry = (uint64_t)(p >> 64);
                u128 d = (u128)u[i + j] - (uint64_t)p - borrow;
                u[i + j] = (uint64_t)d;
                borrow = (uint64_t)(d >> 64) & 1;
            }
            u128 d = (u128)u[j + n] - carry - borrow;
            u[j + n] = (uint64_t)d;
            if ((uint64_t)(d >> 64) & 1) {  // qhat was one too large: add the divisor back
                --qhat;
                uint64_t c = 0;
                for (size_t i = 0; i < n; ++i) {
                    u128 t = (u128)u[i + j] + v[i] + c;
                    u[i + j] = (uint64_t)t;
                    c = (uint64_t)(t >> 64);
                }
                u[j + n] += c;
            }
            quotient[j] = qhat;
        }
        if (q) {
            q->w = quotient;
            q->trim();
        }
        if (r) {
            r->w.assign(n, 0);
            for (size_t i = 0; i < n; ++i) r->w[i] = s ? (u[i] >> s) | (u[i + 1] << (64 - s)) : u[i];
            r->trim();
        }
    }
    static Nat div(const Nat& a, const Nat& b) {
        if (b.is_one()) return a;
        Nat q;
        divmod(a, b, &q, nullptr);
        return q;
    }
    // Lehmer's gcd (Cohen, A Course in Computational Algebraic Number Theory, algorithm 1.3.7) on 62-bit leading digits
    static Nat gcd(Nat a, Nat b) {
        if (cmp(a, b) < 0) std::swap(a, b);
        std::vector<uint64_t> na, nb;  // reused across steps
        while (b.size() > 1) {
            const size_t n = a.size();
            const int s = __builtin_clzll(a.w[n - 1]);
            auto limb = [](const Nat& x, size_t i) { return i < x.size() ? x.w[i] : (uint64_t)0; };
            uint64_t at = s ? (a.w[n - 1] << s) | (a.w[n - 2] >> (64 - s)) : a.w[n - 1];
            uint64_t bt = s ? (limb(b, n - 1) << s) | (limb(b, n - 2) >> (64 - s)) : limb(b, n - 1);
            int64_t ah = (int64_t)(at >> 2), bh = (int64_t)(bt >> 2);  // cofactors stay below the leading digit
            int64_t A = 1, B = 0, C = 0, D = 1;
            for (;;) {
                if (bh + C <= 0 || bh + D <= 0) break;
                const int64_t q1 = (ah + A) / (bh + C), q2 = (ah + B) / (bh + D);
                if (q1 != q2) break;
                int64_t T = A - q1 * C;
                A = C;
                C = T;
                T = B - q1 * D;
                B = D;
                D = T;
                T = ah - q1 * bh;
                ah = bh;
                bh = T;
            }
            if (B == 0) {
                Nat r;
                divmod(a, b, nullptr, &r);
                a.w.swap(b.w);
                b.w.swap(r.w);
            } else {
                // (a, b) <- (A a + B b, C a + D b); each pair of cofactors has opposite signs and both results are >= 0
                const size_t len = a.size();
                na.resize(len);
                nb.resize(len);
                i128 ca = 0, cb = 0;
                const i128 a64 = A, b64 = B, c64 = C, d64 = D;
                for (size_t i = 0; i < len; ++i) {
                    const uint64_t ai = a.w[i], bi = limb(b, i);
                    // split the products to keep every intermediate inside 128 bits
                    i128 ta = ca + (i128)((u128)(uint64_t)(a64 < 0 ? -a64 : a64) * ai) * (a64 < 0 ? -1 : 1)
                                 + (i128)((u128)(uint64_t)(b64 < 0 ? -b64 : b64) * bi) * (b64 < 0 ? -1 : 1);
                    i128 tb = cb + (i128)((u128)(uint64_t)(c64 < 0 ? -c64 : c64) * ai) * (c64 < 0 ? -1 : 1)
                                 + (i128)((u128)(uint64_t)(d64 < 0 ? -d64 : d64) * bi) * (d64 < 0 ? -1 : 1);
                    na[i] = (uint64_t)ta;
                    nb[i] = (uint64_t)tb;
                    ca = ta >> 64;
                    cb = tb >> 64;
                }
                a.w.swap(na);
                b.w.swap(nb);
                a.trim();
                b.trim();
                if (cmp(a, b) < 0) std::swap(a, b);
            }
        }
        if (b.zero()) return a;
        uint64_t r = a.mod_small(b.w[0]);
        return Nat((u128)gcd64(b.w[0], r));
    }
    std::string to_string() const {
        if (w.empty()) return "0";
        Nat t = *this;
        std::string out;
        while (!t.w.empty()) {
            uint64_t rem = t.div_small(1000000000000000000ull);
            for (int k = 0; k < 18; ++k) {
                out.push_back((char)('0' + rem % 10));
                rem /= 10;
                if (t.w.empty() && rem == 0) break;
            }
        }
        while (out.size() > 1 && out.back() == '0') out.pop_back();
        return std::string(out.rbegin(), out.rend());
    }
    double to_double_scaled(long& exponent) const {  // value = result * 2^exponent
        double x = 0;
        const size_t limbs = w.size(), take = limbs < 2 ? limbs : 2;
        for (size_t i = 0; i < take; ++i) x = x * 18446744073709551616.0 + (double)w[limbs - 1 - i];
        exponent = 64L * (long)(limbs - take);
        return x;
    }
};

// ---- rationals -------------------------------------------------------------------------------------------------------
class BigRat {
    struct Big {
        Nat n, d;
        bool neg;
    };
    int64_t n_ = 0, d_ = 1;           // valid when big_ is null; |n_| < 2^62, 0 < d_ < 2^62
    std::shared_ptr<const Big> big_;  // immutable, so copies (the reference clones values freely) stay cheap

    static const i128 LIMIT = ((i128)1 << 62) - 1;

    static BigRat from_i128(i128 n, i128 d) {  // d > 0, not yet reduced
        if (n == 0) return BigRat();
        u128 g = gcd128u(n < 0 ? (u128)(-n) : (u128)n, (u128)d);
        if (g > 1) {
            n /= (i128)g;
            d /= (i128)g;
        }
        BigRat r;
        if (n >= -LIMIT && n <= LIMIT && d <= LIMIT) {
            r.n_ = (int64_t)n;
            r.d_ = (int64_t)d;
        } else {
            auto b = std::make_shared<Big>();
            b->neg = n < 0;
            b->n = Nat(n < 0 ? (u128)(-n) : (u128)n);
            b->d = Nat((u128)d);
            r.big_ = b;
        }
        return r;
    }
    // n/d already in lowest terms, d != 0
    static BigRat from_reduced(bool neg, Nat n, Nat d) {
        BigRat r;
        if (n.zero()) return r;
        if (n.fits62() && d.fits62()) {
            r.n_ = neg ? -(int64_t)n.low() : (int64_t)n.low();
            r.d_ = (int64_t)d.low();
        } else {
            auto b = std::make_shared<Big>();
            b->neg = neg;
            b->n.w.swap(n.w);
            b->d.w.swap(d.w);
            r.big_ = b;
        }
        return r;
    }
    static BigRat from_unreduced(bool neg, Nat n, Nat d) {
        if (n.zero()) return BigRat();
        Nat g = Nat::gcd(n, d);
        if (!g.is_one()) {
            n = Nat::div(n, g);
            d = Nat::div(d, g);
        }
        return from_reduced(neg, n, d);
    }
    bool neg() const { return big_ ? big_->neg : n_ < 0; }
    Nat nat_n() const { return big_ ? big_->n : Nat((u128)(n_ < 0 ? -n_ : n_)); }
    Nat nat_d() const { return big_ ? big_->d : Nat((u128)d_); }

    // Knuth 4.5.1: a/b + c/d with one gcd of the denominators (and a second, small one only when they share a factor)
    static BigRat add_big(const BigRat& x, const BigRat& y, bool negate_y) {
        if (y.is_zero()) return x;
        if (x.is_zero()) return negate_y ? -y : y;
        const Nat a = x.nat_n(), b = x.nat_d(), c = y.nat_n(), d = y.nat_d();
        const bool xs = x.neg(), ys = y.neg() != negate_y;
        Nat g = Nat::gcd(b, d);
        Nat bp = Nat::div(b, g), dp = Nat::div(d, g);
        Nat l = Nat::mul(a, dp), r = Nat::mul(c, bp), t;
        bool ts;
        if (xs == ys) {
            t = Nat::add(l, r);
            ts = xs;
        } else {
            int cmp = Nat::cmp(l, r);
            if (cmp == 0) return BigRat();
            t = cmp > 0 ? Nat::sub(l, r) : Nat::sub(r, l);
            ts = cmp > 0 ? xs : ys;
        }
        if (g.is_one()) return from_reduced(ts, t, Nat::mul(b, d));
        Nat g2 = Nat::gcd(t, g);
        if (g2.is_one()) return from_reduced(ts, t, Nat::mul(bp, d));
        return from_reduced(ts, Nat::div(t, g2), Nat::mul(bp, Nat::div(d, g2)));
    }
    // (a/b) * (c/d) with the cross gcds
    static BigRat mul_parts(bool neg, const Nat& a, const Nat& b, const Nat& c, const Nat& d) {
        if (a.zero() || c.zero()) return BigRat();
        Nat g1 = Nat::gcd(a, d), g2 = Nat::gcd(c, b);
        return from_reduced(neg, Nat::mul(Nat::div(a, g1), Nat::div(c, g2)), Nat::mul(Nat::div(b, g2), Nat::div(d, g1)));
    }

public:
    BigRat() = default;
    BigRat(long long v) {
        if (v > -LIMIT && v < LIMIT) n_ = v;
        else *this = from_i128(v, 1);
    }
    explicit BigRat(const Rat& r) { *this = from_i128(r.n, r.d); }
    // back to the fixed-width type of the host model; RatOverflow when it does not fit
    Rat to_rat() const {
        if (!big_) return Rat((i128)n_, (i128)d_);
        if (big_->n.bits() > 126 || big_->d.bits() > 126) throw RatOverflow();
        auto to128 = [](const Nat& v) {
            u128 out = 0;
            for (size_t i = v.w.size(); i-- > 0;) out = (out << 64) | v.w[i];
            return (i128)out;
        };
        Rat r;
        r.n = big_->neg ? -to128(big_->n) : to128(big_->n);
        r.d = to128(big_->d);
        return r;
    }
    static BigRat make(long long n, long long d) {
        if (d == 0) throw std::runtime_error("zero denominator");
        return d < 0 ? from_i128(-(i128)n, -(i128)d) : from_i128(n, d);
    }
    // "num/den" or "num" in decimal
    static BigRat parse(const std::string& text) {
        size_t slash = text.find('/');
        bool neg = false;
        auto to_nat = [&neg](const std::string& s) {
            Nat v;
            size_t i = 0;
            if (i < s.size() && (s[i] == '-' || s[i] == '+')) {
                if (s[i] == '-') neg = !neg;
                ++i;
            }
            if (i == s.size()) throw std::runtime_error("bad integer: " + s);
            for (; i < s.size(); ++i) {
                if (s[i] < '0' || s[i] > '9') throw std::runtime_error("bad integer: " + s);
                v.mul_add_small(10u, (uint64_t)(s[i] - '0'));
            }
            v.trim();
            return v;
        };
        Nat n = to_nat(slash == std::string::npos ? text : text.substr(0, slash));
        Nat d = slash == std::string::npos ? Nat((u128)1) : to_nat(text.substr(slash + 1));
        if (d.zero()) throw std::runtime_error("zero denominator");
        return from_unreduced(neg, n, d);
    }

    bool is_zero() const { return !big_ && n_ == 0; }
    int sign() const { return big_ ? (big_->neg ? -1 : 1) : (n_ > 0) - (n_ < 0); }
    bool is_big() const { return (bool)big_; }
    size_t bits() const { return std::max(nat_n().bits(), nat_d().bits()); }
    std::string to_string() const { return (neg() ? "-" : "") + nat_n().to_string() + "/" + nat_d().to_string(); }
    double to_double() const {
        if (!big_) return (double)n_ / (double)d_;
        long en, ed;
        double xn = big_->n.to_double_scaled(en), xd = big_->d.to_double_scaled(ed);
        double v = std::ldexp(xn / xd, (int)(en - ed));
        return big_->neg ? -v : v;
    }

    BigRat operator-() const {
        if (!big_) {
            BigRat r;
            r.n_ = -n_;
            r.d_ = d_;
            return r;
        }
        auto b = std::make_shared<Big>(*big_);
        b->neg = !b->neg;
        BigRat r;
        r.big_ = b;
        return r;
    }
    friend BigRat operator+(const BigRat& a, const BigRat& b) {
        if (!a.big_ && !b.big_) {
            if (a.d_ == b.d_) return from_i128((i128)a.n_ + b.n_, a.d_);
            return from_i128((i128)a.n_ * b.d_ + (i128)b.n_ * a.d_, (i128)a.d_ * b.d_);
        }
        return add_big(a, b, false);
    }
    friend BigRat operator-(const BigRat& a, const BigRat& b) {
        if (!a.big_ && !b.big_) {
            if (a.d_ == b.d_) return from_i128((i128)a.n_ - b.n_, a.d_);
            return from_i128((i128)a.n_ * b.d_ - (i128)b.n_ * a.d_, (i128)a.d_ * b.d_);
        }
        return add_big(a, b, true);
    }
    friend BigRat operator*(const BigRat& a, const BigRat& b) {
        if (!a.big_ && !b.big_) return from_i128((i128)a.n_ * b.n_, (i128)a.d_ * b.d_);
        return mul_parts(a.neg() != b.neg(), a.nat_n(), a.nat_d(), b.nat_n(), b.nat_d());
    }
    friend BigRat operator/(const BigRat& a, const BigRat& b) {
        if (b.is_zero()) throw std::runtime_error("division by zero");
        if (!a.big_ && !b.big_) {
            i128 n = (i128)a.n_ * b.d_, d = (i128)a.d_ * b.n_;
            return d < 0 ? from_i128(-n, -d) : from_i128(n, d);
        }
        return mul_parts(a.neg() != b.neg(), a.nat_n(), a.nat_d(), b.nat_d(), b.nat_n());
    }
    BigRat& operator+=(const BigRat& o) { return *this = *this + o; }
    BigRat& operator-=(const BigRat& o) { return *this = *this - o; }
    BigRat& operator*=(const BigRat& o) { return *this = *this * o; }
    BigRat& operator/=(const BigRat& o) { return *this = *this / o; }

    friend int compare(const BigRat& a, const BigRat& b) {
        if (!a.big_ && !b.big_) {
            i128 l = (i128)a.n_ * b.d_, r = (i128)b.n_ * a.d_;
            return (l > r) - (l < r);
        }
        int sa = a.sign(), sb = b.sign();
        if (sa != sb) return sa < sb ? -1 : 1;
        if (sa == 0) return 0;
        int c = Nat::cmp(Nat::mul(a.nat_n(), b.nat_d()), Nat::mul(b.nat_n(), a.nat_d()));
        return sa < 0 ? -c : c;
    }
    friend bool operator==(const BigRat& a, const BigRat& b) {
        if (!a.big_ && !b.big_) return a.n_ == b.n_ && a.d_ == b.d_;
        if (!a.big_ || !b.big_) return false;  // normalised: a value that fits the inline form is never stored big
        return a.big_->neg == b.big_->neg && Nat::cmp(a.big_->n, b.big_->n) == 0 && Nat::cmp(a.big_->d, b.big_->d) == 0;
    }
    friend bool operator!=(const BigRat& a, const BigRat& b) { return !(a == b); }
    friend bool operator<(const BigRat& a, const BigRat& b) { return compare(a, b) < 0; }
    friend bool operator>(const BigRat& a, const BigRat& b) { return compare(a, b) > 0; }
    friend bool operator<=(const BigRat& a, const BigRat& b) { return compare(a, b) <= 0; }
    friend bool operator>=(const BigRat& a, const BigRat& b) { return compare(a, b) >= 0; }
};


inline int cmp(const BigRat& a, const BigRat& b) { return compare(a, b); }

}  // namespace relp
