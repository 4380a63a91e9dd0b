// Minimal arbitrary-precision signed integer for the HOST side of the exact certificate (certify.hip):
// Horner assembly of p-adic digits, rational reconstruction, and sign checks.  No GMP headers exist in the image.
// 32-bit limbs, little endian, sign-magnitude.  Only what the certificate needs; O(n^2) algorithms (n <= a few
// hundred limbs there).
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace relp {

class BigInt {
public:
    std::vector<uint32_t> mag;  // magnitude, no leading zero limbs
    bool neg = false;

    BigInt() = default;
    BigInt(long long v) { assign((__int128)v); }
    static BigInt from_i128(__int128 v) {
        BigInt b;
        b.assign(v);
        return b;
    }
    void assign(__int128 v) {
        mag.clear();
        neg = v < 0;
        unsigned __int128 u = neg ? (unsigned __int128)(-(v + 1)) + 1 : (unsigned __int128)v;
        while (u != 0) {
            mag.push_back((uint32_t)u);
            u >>= 32;
        }
    }
    bool is_zero() const { return mag.empty(); }
    int sign() const { return mag.empty() ? 0 : (neg ? -1 : 1); }
    size_t bits() const {
        if (mag.empty()) return 0;
        return (mag.size() - 1) * 32 + (32 - __builtin_clz(mag.back()));
    }
    void trim() {
        while (!mag.empty() && mag.back() == 0) mag.pop_back();
        if (mag.empty()) neg = false;
    }
    BigInt operator-() const {
        BigInt r = *this;
        if (!r.mag.empty()) r.neg = !r.neg;
        return r;
    }
    BigInt abs() const {
        BigInt r = *this;
        r.neg = false;
        return r;
    }

    static int cmp_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
        if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
        for (size_t i = a.size(); i-- > 0;)
            if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
        return 0;
    }
    static std::vector<uint32_t> add_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
        const std::vector<uint32_t>& x = a.size() >= b.size() ? a : b;
        const std::vector<uint32_t>& y = a.size() >= b.size() ? b : a;
        std::vector<uint32_t> r(x.size() + 1);
        uint64_t carry = 0;
        for (size_t i = 0; i < x.size(); ++i) {
            uint64_t s = (uint64_t)x[i] + (i < y.size() ? y[i] : 0) + carry;
            r[i] = (uint32_t)s;
            carry = s >> 32;
        }
        r[x.size()] = (uint32_t)carry;
        while (!r.empty() && r.back() == 0) r.pop_back();
        return r;
    }
    // a >= b
    static std::vector<uint32_t> sub_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
        std::vector<uint32_t> r(a.size());
        int64_t borrow = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            int64_t d = (int64_t)a[i] - (i < b.size() ? b[i] : 0) - borrow;
            borrow = d < 0;
            if (d < 0) d += ((int64_t)1 << 32);
            r[i] = (uint32_t)d;
        }
        while (!r.empty() && r.back() == 0) r.pop_back();
        return r;
    }
    friend BigInt operator+(const BigInt& a, const BigInt& b) {
        BigInt r;
        if (a.neg == b.neg) {
            r.mag = add_mag(a.mag, b.mag);
            r.neg = a.neg;
        } else {
            int c = cmp_mag(a.mag, b.mag);
            if (c == 0) return r;
            if (c > 0) { r.mag = sub_mag(a.mag, b.mag); r.neg = a.neg; }
            else { r.mag = sub_mag(b.mag, a.mag); r.neg = b.neg; }
        }
        r.trim();
        return r;
    }
    friend BigInt operator-(const BigInt& a, const BigInt& b) { return a + (-b); }
    friend BigInt operator*(const BigInt& a, const BigInt& b) {
        BigInt r;
        if (a.mag.empty() || b.mag.empty()) return r;
        r.mag.assign(a.mag.size() + b.mag.size(), 0);
        for (size_t i = 0; i < a.mag.size(); ++i) {
            uint64_t carry = 0;
            const uint64_t ai = a.mag[i];
            for (size_t j = 0; j < b.mag.size(); ++j) {
                uint64_t t = ai * b.mag[j] + r.mag[i + j] + carry;
                r.mag[i + j] = (uint32_t)t;
                carry = t >> 32;
            }
            r.mag[i + b.mag.size()] += (uint32_t)carry;
        }
        r.neg = a.neg != b.neg;
        r.trim();
        return r;
    }
    friend int cmp(const BigInt& a, const BigInt& b) {
        if (a.sign() != b.sign()) return a.sign() < b.sign() ? -1 : 1;
        int c = cmp_mag(a.mag, b.mag);
        return a.neg ? -c : c;
    }
    friend bool operator==(const BigInt& a, const BigInt& b) { return cmp(a, b) == 0; }
    friend bool operator!=(const BigInt& a, const BigInt& b) { return cmp(a, b) != 0; }
    friend bool operator<(const BigInt& a, const BigInt& b) { return cmp(a, b) < 0; }

    // in-place: this = this * m + a   (m, a unsigned 32-bit; this >= 0)
    void mul_add_small(uint32_t m, uint32_t a) {
        uint64_t carry = a;
        for (size_t i = 0; i < mag.size(); ++i) {
            uint64_t t = (uint64_t)mag[i] * m + carry;
            mag[i] = (uint32_t)t;
            carry = t >> 32;
        }
        if (carry) mag.push_back((uint32_t)carry);
    }
    // in-place magnitude division by a small value, returns the remainder
    uint32_t div_small(uint32_t d) {
        uint64_t rem = 0;
        for (size_t i = mag.size(); i-- > 0;) {
            uint64_t cur = (rem << 32) | mag[i];
            mag[i] = (uint32_t)(cur / d);
            rem = cur % d;
        }
        trim();
        return (uint32_t)rem;
    }

    // Truncated division (quotient toward zero, remainder has the sign of the dividend).  Knuth algorithm D.
    static void divmod(const BigInt& a, const BigInt& b, BigInt& q, BigInt& r) {
        q = BigInt();
        r = BigInt();
        if (b.mag.empty()) throw std::runtime_error("BigInt division by zero");
        if (cmp_mag(a.mag, b.mag) < 0) {
            r = a;
            return;
        }
        if (b.mag.size() == 1) {
            q = a;
            q.neg = false;
            uint32_t rem = q.div_small(b.mag[0]);
            r = BigInt((long long)rem);
        } else {
            const int s = __builtin_clz(b.mag.back());
            std::vector<uint32_t> v = shl_mag(b.mag, s);
            std::vector<uint32_t> u = shl_mag(a.mag, s);
            if (u.size() == a.mag.size()) u.push_back(0);
            const size_t n = v.size(), mlen = u.size() - n;
            std::vector<uint32_t> qm(mlen, 0);
            for (size_t j = mlen; j-- > 0;) {
                uint64_t num = ((uint64_t)u[j + n] << 32) | u[j + n - 1];
                uint64_t qhat = num / v[n - 1];
                uint64_t rhat = num % v[n - 1];
                while (qhat >= ((uint64_t)1 << 32) || qhat * v[n - 2] > ((rhat << 32) | u[j + n - 2])) {
                    --qhat;
                    rhat += v[n - 1];
                    if (rhat >= ((uint64_t)1 << 32)) break;
                }
                int64_t borrow = 0;
                uint64_t carry = 0;
                for (size_t i = 0; i < n; ++i) {
                    uint64_t pr = qhat * v[i] + carry;
                    carry = pr >> 32;
                    int64_t t = (int64_t)u[i + j] - borrow - (int64_t)(uint32_t)pr;
                    borrow = t < 0;
                    u[i + j] = (uint32_t)t;
                }
                int64_t t = (int64_t)u[j + n] - borrow - (int64_t)carry;
                borrow = t < 0;
                u[j + n] = (uint32_t)t;
                if (borrow) {
                    --qhat;
                    uint64_t c2 = 0;
                    for (size_t i = 0; i < n; ++i) {
                        uint64_t sum = (uint64_t)u[i + j] + v[i] + c2;
                        u[i + j] = (uint32_t)sum;
                        c2 = sum >> 32;
                    }
                    u[j + n] += (uint32_t)c2;
                }
                qm[j] = (uint32_t)qhat;
            }
            q.mag = qm;
            q.trim();
            u.resize(n);
            r.mag = shr_mag(u, s);
            r.trim();
        }
        q.neg = !q.mag.empty() && (a.neg != b.neg);
        r.neg = !r.mag.empty() && a.neg;
    }
    static std::vector<uint32_t> shl_mag(const std::vector<uint32_t>& a, int s) {
        if (s == 0) return a;
        std::vector<uint32_t> r(a.size() + 1, 0);
        for (size_t i = 0; i < a.size(); ++i) {
            r[i] |= a[i] << s;
            r[i + 1] = a[i] >> (32 - s);
        }
        while (!r.empty() && r.back() == 0) r.pop_back();
        return r;
    }
    static std::vector<uint32_t> shr_mag(const std::vector<uint32_t>& a, int s) {
        if (s == 0) return a;
        std::vector<uint32_t> r(a.size(), 0);
        for (size_t i = 0; i < a.size(); ++i) {
            r[i] = a[i] >> s;
            if (i + 1 < a.size()) r[i] |= a[i + 1] << (32 - s);
        }
        while (!r.empty() && r.back() == 0) r.pop_back();
        return r;
    }
    friend BigInt operator/(const BigInt& a, const BigInt& b) {
        BigInt q, r;
        divmod(a, b, q, r);
        return q;
    }
    friend BigInt operator%(const BigInt& a, const BigInt& b) {
        BigInt q, r;
        divmod(a, b, q, r);
        return r;
    }
    static BigInt gcd(BigInt a, BigInt b) {
        a.neg = b.neg = false;
        while (!b.is_zero()) {
            BigInt r = a % b;
            a = b;
            b = r;
        }
        return a;
    }
    std::string to_string() const {
        if (mag.empty()) return "0";
        BigInt t = *this;
        t.neg = false;
        std::string out;
        while (!t.mag.empty()) {
            uint32_t rem = t.div_small(1000000000u);
            for (int k = 0; k < 9; ++k) {
                out.push_back((char)('0' + rem % 10));
                rem /= 10;
                if (t.mag.empty() && rem == 0) break;
            }
        }
        while (out.size() > 1 && out.back() == '0') out.pop_back();
        if (neg) out.push_back('-');
        return std::string(out.rbegin(), out.rend());
    }
};

}  // namespace relp
