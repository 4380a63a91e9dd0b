// The revised simplex loop in EXACT fixed-width integer arithmetic on the device (BASELINE.json north_star: "fixed-width
// int128/int256 rational arithmetic replaces arbitrary-precision on device").
//
// Every rational of the reference's `Carry<RationalBig, _>` at a basis B is held over ONE common denominator -- Edmonds'
// integer-preserving pivoting: with the rows of the LP scaled to integers,
//     N = D * B^-1,   D = |det B|,        (all entries of N are integers)
//     alpha~_q = N a_q,  x~_B = N b,  c~_j = D c_j - c_B' N a_j,  gamma~_j = D^2 + |N a_j|^2       (numerators over D, D, D, D^2)
// and a pivot on row p is   D' = alpha~_p,   N'_i = (alpha~_p N_i - alpha~_i N_p) / D  (i != p),  N'_p = N_p,
// where the division is EXACT.  The rows are scaled to integers (r_i = lcm of the row's denominators); the unit columns -- slacks and
// artificials -- are scaled BACK by 1 / r_i, so that the first basis is the identity and D starts at 1 instead of at prod r_i
// (14 000 bits on 25FV47).  Column scaling is invisible to the ratio test; the pricing rule sees it and is corrected exactly:
// with sigma_j the column factors and W = lcm(r)^2,  key_j = c~_j^2 / (w_j D^2 + sum_i w_{B_i} (N a_j)_i^2),  w_j = W sigma_j^2.  Integers are LIMBS x 64-bit two's complement words, LIMBS in {2, 4, 8, 16, 32} (int128 and
// int256 are the two smallest instantiations).  No gcd and no long division ever runs on the device: an exact quotient is one
// truncated multiplication with the inverse of D modulo 2^(64 LIMBS) (Newton iteration, once per pivot).  Every result is
// guarded by a floating-point magnitude bound; when a value might not fit, the solve stops with status OVERFLOW and the host
// restarts it with twice the limbs.
//
// The decisions are the reference's own, exactly: steepest-edge pricing with the LAST maximum on ties
// (strategy/pivot_rule.rs:221-241), the minimum ratio with Bland's rule on ties (tableau/mod.rs:287-313), zero-level pivots
// (phase_one.rs:232-278).  Candidates are ranked with double-precision estimates of the exact quantities and every near-tie is
// decided by exact cross-multiplication, so the pivot sequence is the oracle's (tests/test_gpu_exact.py: whole golden traces).
// One workgroup owns one LP (these are the small LPs; the whole solve is one kernel, no host round trip per pivot).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "bigint.hpp"
#include "solver.hpp"
#include <hip/hip_cooperative_groups.h>

#include "grid_barrier.hpp"
#include "wave_ops.hpp"

namespace relp {

namespace {

using u64 = unsigned long long;
using i64 = long long;
using u128 = unsigned __int128;

constexpr int EX_THREADS = 256;
enum : int { EX_RUNNING = 0, EX_OPTIMAL = 1, EX_INFEASIBLE = 2, EX_UNBOUNDED = 3, EX_OVERFLOW = 4, EX_PIVOT_LIMIT = 5, EX_REDUNDANT_ROWS = 6 };

// ---------------------------------------------------------------------------------------------------
// LIMBS x 64-bit two's complement integers
// ---------------------------------------------------------------------------------------------------
template <int L>
struct Big {
    u64 w[L];
};

template <int L>
__device__ __forceinline__ Big<L> big_from(i64 v) {
    Big<L> r;
    r.w[0] = (u64)v;
#pragma unroll L <= 8 ? L : 1
    for (int k = 1; k < L; ++k) r.w[k] = v < 0 ? ~0ull : 0ull;
    return r;
}
template <int L>
__device__ __forceinline__ bool big_neg(const Big<L>& a) { return (i64)a.w[L - 1] < 0; }
template <int L>
__device__ __forceinline__ bool big_zero(const Big<L>& a) {
    u64 acc = 0;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) acc |= a.w[k];
    return acc == 0;
}
template <int L>
__device__ __forceinline__ Big<L> big_add(const Big<L>& a, const Big<L>& b) {
    Big<L> r;
    u64 carry = 0;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) {
        const u128 s = (u128)a.w[k] + b.w[k] + carry;
        r.w[k] = (u64)s;
        carry = (u64)(s >> 64);
    }
    return r;
}
template <int L>
__device__ __forceinline__ Big<L> big_negate(const Big<L>& a) {
    Big<L> r;
    u64 carry = 1;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) {
        const u128 s = (u128)(~a.w[k]) + carry;
        r.w[k] = (u64)s;
        carry = (u64)(s >> 64);
    }
    return r;
}
template <int L>
__device__ __forceinline__ Big<L> big_sub(const Big<L>& a, const Big<L>& b) { return big_add(a, big_negate(b)); }
// a * b mod 2^(64 L), b a signed 64-bit value
template <int L>
__device__ __forceinline__ Big<L> big_mul_small(const Big<L>& a, i64 b) {
    const u64 mag = b < 0 ? (u64)(-(b + 1)) + 1 : (u64)b;
    Big<L> r;
    u64 carry = 0;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) {
        const u128 t = (u128)a.w[k] * mag + carry;
        r.w[k] = (u64)t;
        carry = (u64)(t >> 64);
    }
    return b < 0 ? big_negate(r) : r;
}
// The same product for the wide types (16 limbs and more), whose operands live in scratch memory.  The loop below reads b.w[j] and
// r.w[i + j] and writes r.w[i + j] for every word product: three scratch accesses per product, and at 128 limbs the update of N
// ran at 4 % of the multiplier's rate (25FV47: 132 of 252 s).  Here the product is formed four output words at a time
// (product scanning by 4 x 4 blocks): a block pair costs eight scratch reads for sixteen word products, the running sum is a
// window of nine words in registers, and nothing is written but the result.  Same value: the product modulo 2^(64 L).
// `blocks` < L / 4: only the low 4 * blocks words of the product are formed (the words above are left zero) -- the caller knows
// that the value it is after fits there (see the update of N).
template <int L>
__device__ __forceinline__ Big<L> big_mul_lo_blocked(const Big<L>& a, const Big<L>& b, int blocks = L / 4) {
    static_assert(L % 4 == 0, "four words per block");
    const int NB = blocks;
    Big<L> r;
    u64 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0;
    for (int K = 0; K < NB; ++K) {
        for (int I = 0; I <= K; ++I) {
            const int J = K - I;
            u64 a4[4], b4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a4[t] = a.w[4 * I + t];
                b4[t] = b.w[4 * J + t];
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                u64 carry = 0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const u128 t = (u128)a4[ii] * b4[jj] + acc[ii + jj] + carry;
                    acc[ii + jj] = (u64)t;
                    carry = (u64)(t >> 64);
                }
#pragma unroll
                for (int k = ii + 4; k < 9; ++k) {
                    const u128 t = (u128)acc[k] + carry;
                    acc[k] = (u64)t;
                    carry = (u64)(t >> 64);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) r.w[4 * K + t] = acc[t];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = acc[t + 4];
#pragma unroll
        for (int t = 5; t < 9; ++t) acc[t] = 0;
    }
#pragma unroll L <= 8 ? L : 1
    for (int k = 4 * NB; k < L; ++k) r.w[k] = 0;
    return r;
}
// (the same with a read through a pointer: the workgroup's copy of alpha~_p u in LDS)
template <int L>
__device__ __forceinline__ Big<L> big_mul_lo_blocked_p(const u64* a, const Big<L>& b, int blocks) {
    static_assert(L % 4 == 0, "four words per block");
    const int NB = blocks;
    Big<L> r;
    u64 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0;
    for (int K = 0; K < NB; ++K) {
        for (int I = 0; I <= K; ++I) {
            const int J = K - I;
            u64 a4[4], b4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a4[t] = a[4 * I + t];
                b4[t] = b.w[4 * J + t];
            }
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                u64 carry = 0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const u128 t = (u128)a4[ii] * b4[jj] + acc[ii + jj] + carry;
                    acc[ii + jj] = (u64)t;
                    carry = (u64)(t >> 64);
                }
#pragma unroll
                for (int k = ii + 4; k < 9; ++k) {
                    const u128 t = (u128)acc[k] + carry;
                    acc[k] = (u64)t;
                    carry = (u64)(t >> 64);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) r.w[4 * K + t] = acc[t];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = acc[t + 4];
#pragma unroll
        for (int t = 5; t < 9; ++t) acc[t] = 0;
    }
#pragma unroll L <= 8 ? L : 1
    for (int k = 4 * NB; k < L; ++k) r.w[k] = 0;
    return r;
}
// a * b + c * d modulo 2^(64 * 4 * blocks) in ONE pass over the blocks: both block products of a pair (I, J) go into the same window (the
// update of N: alpha~_p u N_ik + (-alpha~_i u) N_pk, the second factor stored negated so that the difference is a sum).  a is read
// through a pointer -- the caller passes the workgroup's copy in LDS, the same words for every thread -- b, c, d are the thread's own.
template <int L>
__device__ __forceinline__ Big<L> big_mul_add_lo_blocked(const u64* a, const Big<L>& b, const Big<L>& c, const Big<L>& d, int blocks) {
    static_assert(L % 4 == 0, "four words per block");
    const int NB = blocks;
    Big<L> r;
    u64 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0;
    for (int K = 0; K < NB; ++K) {
        for (int I = 0; I <= K; ++I) {
            const int J = K - I;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                u64 a4[4], b4[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a4[t] = which == 0 ? a[4 * I + t] : c.w[4 * I + t];
                    b4[t] = which == 0 ? b.w[4 * J + t] : d.w[4 * J + t];
                }
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    u64 carry = 0;
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        const u128 t = (u128)a4[ii] * b4[jj] + acc[ii + jj] + carry;
                        acc[ii + jj] = (u64)t;
                        carry = (u64)(t >> 64);
                    }
#pragma unroll
                    for (int k = ii + 4; k < 9; ++k) {
                        const u128 t = (u128)acc[k] + carry;
                        acc[k] = (u64)t;
                        carry = (u64)(t >> 64);
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) r.w[4 * K + t] = acc[t];
#pragma unroll
        for (int t = 0; t < 5; ++t) acc[t] = acc[t + 4];
#pragma unroll
        for (int t = 5; t < 9; ++t) acc[t] = 0;
    }
#pragma unroll L <= 8 ? L : 1
    for (int k = 4 * NB; k < L; ++k) r.w[k] = 0;
    return r;
}
// -(a * b) (negate) or a * b modulo 2^(64 L) by one WAVE, stored word-major at out[k * stride]: lane K < L / 4 forms the block products that land in
// output block K (the same 4 x 4 blocks, a window of nine words with nothing carried in), then the windows are chained in
// order -- every lane alike, from the other lanes' registers -- and lane 0 stores the words.  a in memory (L words side by side),
// b in LDS.  (The rows' factors alpha~_i u of the update: a product per row by one thread each was 1.3 ms of every pivot at 128 limbs.)
template <int L>
__device__ __noinline__ void wave_mul_lo_store(const u64* a, const u64* b, u64* out, size_t stride, int lane, bool negate = true) {
    static_assert(L % 4 == 0 && L / 4 <= WAVE, "a lane per block of four words");
    constexpr int NB = L / 4;
    u64 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0;
    // (block I of a is read ONCE, by lane I, and handed to everybody with readlane when its turn comes: read from memory inside the
    //  loop, every turn was a round trip -- 32 of them at 128 limbs, 40 us for a product that is 3 us of arithmetic)
    u64 a_mine[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) a_mine[t] = lane < NB ? a[4 * lane + t] : 0ull;
    {
#pragma unroll 1
        for (int I = 0; I < NB; ++I) {
            const int J = lane - I;
            u64 a4[4], b4[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)a_mine[t], I), hi = (unsigned)__builtin_amdgcn_readlane((int)(a_mine[t] >> 32), I);
                a4[t] = ((u64)hi << 32) | lo;
                b4[t] = (J >= 0 && lane < NB) ? b[4 * J + t] : 0ull;
            }
            if (J < 0 || lane >= NB) continue;
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                u64 carry = 0;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const u128 t = (u128)a4[ii] * b4[jj] + acc[ii + jj] + carry;
                    acc[ii + jj] = (u64)t;
                    carry = (u64)(t >> 64);
                }
#pragma unroll
                for (int k = ii + 4; k < 9; ++k) {
                    const u128 t = (u128)acc[k] + carry;
                    acc[k] = (u64)t;
                    carry = (u64)(t >> 64);
                }
            }
        }
    }
    u64 over[5] = {0, 0, 0, 0, 0};  // what the blocks below carry into the current one
    bool negation_carry = true;
#pragma unroll 1
    for (int K = 0; K < NB; ++K) {
        u64 window[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)acc[k], K), hi = (unsigned)__builtin_amdgcn_readlane((int)(acc[k] >> 32), K);
            window[k] = ((u64)hi << 32) | lo;
        }
        u64 carry = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const u128 t = (u128)window[k] + (k < 5 ? over[k] : 0ull) + carry;
            window[k] = (u64)t;
            carry = (u64)(t >> 64);
        }
        if (lane == 0) {  // (stored NEGATED where asked, modulo 2^(64 L): ~w + 1 with the carry running while the words are zero)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                out[(size_t)(4 * K + t) * stride] = negate ? ~window[t] + (negation_carry ? 1ull : 0ull) : window[t];
                negation_carry = negation_carry && window[t] == 0;
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) over[k] = window[k + 4];
    }
}
// the value of the low `words` words of a, as a signed number of that length, in all L words
template <int L>
__device__ __forceinline__ Big<L> big_sign_extend(const Big<L>& a, int words) {
    Big<L> r = a;
    const u64 fill = (i64)a.w[words - 1] < 0 ? ~0ull : 0ull;
#pragma unroll L <= 8 ? L : 1
    for (int k = words; k < L; ++k) r.w[k] = fill;
    return r;
}
// a * b mod 2^(64 L)  (the low half of the product: exact whenever the true product fits)
template <int L>
__device__ __forceinline__ Big<L> big_mul_lo(const Big<L>& a, const Big<L>& b) {
    if constexpr (L >= 16) return big_mul_lo_blocked(a, b);
    Big<L> r;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) r.w[k] = 0;
#pragma unroll L <= 8 ? L : 1
    for (int i = 0; i < L; ++i) {
        u64 carry = 0;
#pragma unroll L <= 8 ? L : 1
        for (int j = 0; j < L - i; ++j) {
            const u128 t = (u128)a.w[i] * b.w[j] + r.w[i + j] + carry;
            r.w[i + j] = (u64)t;
            carry = (u64)(t >> 64);
        }
    }
    return r;
}
// bit length of |a| (0 for zero): the magnitude bounds that guard every operation are sums of these
template <int L>
__device__ __forceinline__ int big_bits(const Big<L>& a) {
    const Big<L> m = big_neg(a) ? big_negate(a) : a;
    int top = -1;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k)
        if (m.w[k] != 0) top = k;
    if (top < 0) return 0;
    return 64 * top + (64 - __clzll((long long)m.w[top]));
}
__device__ __forceinline__ int small_bits(i64 v) {
    const u64 mag = v < 0 ? (u64)(-(v + 1)) + 1 : (u64)v;
    return mag == 0 ? 0 : 64 - __clzll((long long)mag);
}
// a / b as a double for values that are themselves far outside the range of a double (2048-bit integers): mantissas from the two
// leading limbs, exponents apart
template <int L>
__device__ __forceinline__ double big_ratio(const Big<L>& a, const Big<L>& b) {
    auto split = [](const Big<L>& v, int* exponent) {
        const bool neg = big_neg(v);
        const Big<L> m = neg ? big_negate(v) : v;
        int top = -1;
#pragma unroll L <= 8 ? L : 1
        for (int k = 0; k < L; ++k)
            if (m.w[k] != 0) top = k;
        if (top < 0) { *exponent = 0; return 0.0; }
        double x = (double)m.w[top];
        if (top > 0) x = x * 18446744073709551616.0 + (double)m.w[top - 1];
        *exponent = 64 * (top > 0 ? top - 1 : 0);
        return neg ? -x : x;
    };
    int ea = 0, eb = 0;
    const double ma = split(a, &ea), mb = split(b, &eb);
    return ldexp(ma / mb, ea - eb);
}
// The two leading words of |v| as a double and the exponent that goes with them (the `split` of big_ratio), for a value whose
// words arrive one at a time, least significant first, and are never held together: both readings are carried along -- the value
// itself and its two's complement negation, whose carry runs while the words are zero -- and the sign (the last word) picks one.
struct LeadingWords {
    int top_p = -1, top_n = -1;
    u64 p_top = 0, p_below = 0, n_top = 0, n_below = 0, prev = 0, prev_n = 0;
    bool carry = true;
    __device__ __forceinline__ void feed(int k, u64 word) {
        if (word != 0) { top_p = k; p_top = word; p_below = prev; }
        prev = word;
        const u64 negated = ~word + (carry ? 1ull : 0ull);
        carry = carry && word == 0;
        if (negated != 0) { top_n = k; n_top = negated; n_below = prev_n; }
        prev_n = negated;
    }
    // (after the last word)
    __device__ __forceinline__ double mantissa(int* exponent) const {
        const bool neg = (i64)prev < 0;
        const int top = neg ? top_n : top_p;
        if (top < 0) { *exponent = 0; return 0.0; }
        double x = (double)(neg ? n_top : p_top);
        if (top > 0) x = x * 18446744073709551616.0 + (double)(neg ? n_below : p_below);
        *exponent = 64 * (top > 0 ? top - 1 : 0);
        return neg ? -x : x;
    }
};
template <int L>
__device__ __forceinline__ double big_mantissa(const Big<L>& v, int* exponent) {  // (the same numbers from a value held whole)
    LeadingWords lead;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) lead.feed(k, v.w[k]);
    return lead.mantissa(exponent);
}
template <int L>
__device__ __forceinline__ int big_ctz(const Big<L>& a) {  // a != 0
    int bits = 0;
    bool done = false;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) {
        if (!done) {
            if (a.w[k] == 0) bits += 64;
            else { bits += __ffsll((long long)a.w[k]) - 1; done = true; }
        }
    }
    return bits;
}
template <int L>
__device__ __forceinline__ Big<L> big_sar(const Big<L>& a, int bits) {  // arithmetic shift right
    const int words = bits >> 6, rem = bits & 63;
    const u64 fill = big_neg(a) ? ~0ull : 0ull;
    Big<L> r;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) {
        const int lo = k + words, hi = k + words + 1;
        const u64 wl = lo < L ? a.w[lo < L ? lo : 0] : fill;
        const u64 wh = hi < L ? a.w[hi < L ? hi : 0] : fill;
        r.w[k] = rem ? (wl >> rem) | (wh << (64 - rem)) : wl;
    }
    return r;
}
// 1 / d modulo 2^(64 L) and truncated products on word arrays in LDS, by the whole workgroup (round 3 had ONE thread do them, once per
// pivot, with the precision doubling word by word: at 128 limbs that Newton iteration was 1.5 ms of every pivot).  A
// product is formed by columns: thread k sums the word products of output word k in three words, one thread then runs the carries
// through the columns -- two barriers a product, about 3 k cycles at 128 limbs instead of 100 k.  Called by every thread of the
// workgroup (the arrays are in LDS); `part` holds 3 words per output word.
__device__ __forceinline__ void block_mul_lo(const u64* a, int la, const u64* b, int lb, u64* out, int lo, u64* part) {  // out may not alias a, b
    const int k = threadIdx.x;
    if (k < lo) {
        u64 c0 = 0, c1 = 0, c2 = 0;
        const int j0 = k - la + 1 > 0 ? k - la + 1 : 0, j1 = k < lb - 1 ? k : lb - 1;
        for (int j = j0; j <= j1; ++j) {
            const u128 prod = (u128)a[k - j] * b[j];
            const u128 low = (u128)c0 + (u64)prod;
            c0 = (u64)low;
            const u128 mid = (u128)c1 + (u64)(prod >> 64) + (u64)(low >> 64);
            c1 = (u64)mid;
            c2 += (u64)(mid >> 64);
        }
        part[3 * k] = c0;
        part[3 * k + 1] = c1;
        part[3 * k + 2] = c2;
    }
    __syncthreads();
    if (k == 0) {
        u64 r1 = 0, r2 = 0;  // what the lower columns carry into this one (two words)
        for (int c = 0; c < lo; ++c) {
            const u128 low = (u128)part[3 * c] + r1;
            out[c] = (u64)low;
            const u128 mid = (u128)part[3 * c + 1] + r2 + (u64)(low >> 64);
            r1 = (u64)mid;
            r2 = part[3 * c + 2] + (u64)(mid >> 64);
        }
    }
    __syncthreads();
}
// a * b modulo 2^(256 nb) -- or 2 minus that -- for word arrays in LDS (a: 4 nb words, b: 4 nb_b words, out: 4 nb words, distinct from
// both), by ONE wave: lane K < nb forms the block products that land in output block K, then the windows are chained in order by every
// lane alike (as wave_mul_lo_store).  The caller puts workgroup barriers around it.
__device__ __forceinline__ void wave_mul_lo_lds(const u64* a, const u64* b, int nb_b, u64* out, int nb, int lane, bool two_minus) {
    u64 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0;
#pragma unroll 1
    for (int J = 0; J < nb_b; ++J) {  // (J the same for every lane: b's block is one broadcast read)
        const int I = lane - J;
        if (I < 0 || lane >= nb) continue;
        u64 a4[4], b4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a4[t] = a[4 * I + t];
            b4[t] = b[4 * J + t];
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            u64 carry = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const u128 t = (u128)a4[ii] * b4[jj] + acc[ii + jj] + carry;
                acc[ii + jj] = (u64)t;
                carry = (u64)(t >> 64);
            }
#pragma unroll
            for (int k = ii + 4; k < 9; ++k) {
                const u128 t = (u128)acc[k] + carry;
                acc[k] = (u64)t;
                carry = (u64)(t >> 64);
            }
        }
    }
    u64 over[5] = {0, 0, 0, 0, 0};  // what the blocks below carry into the current one
    u64 complement_carry = 3;        // 2 - v = ~v + 3
#pragma unroll 1
    for (int K = 0; K < nb; ++K) {
        u64 window[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)acc[k], K), hi = (unsigned)__builtin_amdgcn_readlane((int)(acc[k] >> 32), K);
            window[k] = ((u64)hi << 32) | lo;
        }
        u64 carry = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const u128 t = (u128)window[k] + (k < 5 ? over[k] : 0ull) + carry;
            window[k] = (u64)t;
            carry = (u64)(t >> 64);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            u64 word = window[t];
            if (two_minus) {
                const u128 sum = (u128)(~word) + complement_carry;
                word = (u64)sum;
                complement_carry = (u64)(sum >> 64);
            }
            if (lane == 0) out[4 * K + t] = word;
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) over[k] = window[k + 4];
    }
}
// 1 / d modulo 2^(64 L) for the wide types (L >= 16), d odd, all arrays in LDS: the first four words by one thread (Newton's doubling on
// 64-bit and 128-bit arithmetic), then every doubling is two truncated products by the FIRST WAVE of the workgroup (wave_mul_lo_lds).
// (block_inverse_odd's products -- a thread per column of the product, its operands read from LDS term by term, then one thread
//  running the carries through 128 columns -- made 105 us of every pivot at 128 limbs; this is 15.)
template <int L>
__device__ void wave_inverse_odd(const u64* d, u64* x, u64* t, u64* x2) {
    static_assert(L >= 16 && L % 4 == 0, "blocks of four words");
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    if (tid == 0) {
        auto mul_lo = [](const u64* p, const u64* q, u64* r, int n) {  // r = p q modulo 2^(64 n), n <= 4
            u64 w[4] = {0, 0, 0, 0};
            for (int i = 0; i < n; ++i) {
                u64 carry = 0;
                for (int j = 0; i + j < n; ++j) {
                    const u128 pr = (u128)p[i] * q[j] + w[i + j] + carry;
                    w[i + j] = (u64)pr;
                    carry = (u64)(pr >> 64);
                }
            }
            for (int i = 0; i < n; ++i) r[i] = w[i];
        };
        u64 d4[4] = {d[0], d[1], d[2], d[3]}, x4[4] = {0, 0, 0, 0};
        u64 inv = d4[0];  // d * d = 1 (mod 8): three correct bits, doubled five times
        for (int k = 0; k < 5; ++k) inv *= 2 - d4[0] * inv;
        x4[0] = inv;
        for (int want = 2; want <= 4; want *= 2) {
            u64 t4[4], y4[4];
            mul_lo(d4, x4, t4, want);
            u64 carry = 3;  // t <- 2 - t = ~t + 3
            for (int k = 0; k < want; ++k) {
                const u128 sum = (u128)(~t4[k]) + carry;
                t4[k] = (u64)sum;
                carry = (u64)(sum >> 64);
            }
            mul_lo(t4, x4, y4, want);
            for (int k = 0; k < want; ++k) x4[k] = y4[k];
        }
        for (int k = 0; k < 4; ++k) x[k] = x4[k];
    }
    __syncthreads();
    for (int have = 4; have < L; have *= 2) {
        const int want = 2 * have < L ? 2 * have : L;
        if (tid < WAVE) wave_mul_lo_lds(d, x, have / 4, t, want / 4, lane, true);  // t = 2 - d x
        __syncthreads();
        if (tid < WAVE) wave_mul_lo_lds(t, x, have / 4, x2, want / 4, lane, false);
        __syncthreads();
        if (tid < want) x[tid] = x2[tid];
        __syncthreads();
    }
}
// odd = v >> ctz(v) for a positive integer of L words in LDS (v != 0), by the whole workgroup: a thread per word, the lowest non-zero
// word from the waves' ballots (`scratch`: a word per wave, LDS); the caller's barrier before, one after
template <int L>
__device__ __forceinline__ void block_odd_part(const u64* v, u64* odd, u64* scratch) {
    const int tid = threadIdx.x;
    const u64 w = tid < L ? v[tid] : 0ull;
    const unsigned long long nonzero = __ballot(w != 0);
    if ((tid & (WAVE - 1)) == 0) scratch[tid / WAVE] = nonzero;
    __syncthreads();
    int low = 0;
    for (int wv = (int)blockDim.x / WAVE - 1; wv >= 0; --wv)
        if (scratch[wv] != 0) low = WAVE * wv + __ffsll((long long)scratch[wv]) - 1;
    const int shift_bits = 64 * low + __ffsll((long long)v[low]) - 1;
    if (tid < L) {
        const int ws = shift_bits >> 6, bs = shift_bits & 63;
        const u64 lo = tid + ws < L ? v[tid + ws] : 0ull, hi = tid + ws + 1 < L ? v[tid + ws + 1] : 0ull;
        odd[tid] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
    }
    __syncthreads();
}
template <int L>
__device__ void block_inverse_odd(const u64* d, u64* x, u64* t, u64* x2, u64* part) {  // x = 1 / d modulo 2^(64 L), d odd; every thread of the workgroup
    if (threadIdx.x == 0) {
        u64 inv = d[0];  // d * d = 1 (mod 8): three correct bits, doubled five times
        for (int k = 0; k < 5; ++k) inv *= 2 - d[0] * inv;
        x[0] = inv;
    }
    __syncthreads();
    for (int have = 1; have < L; have *= 2) {
        const int want = 2 * have < L ? 2 * have : L;
        block_mul_lo(d, want, x, have, t, want, part);
        if (threadIdx.x == 0) {
            u64 carry = 3;  // t <- 2 - t = ~t + 3
            for (int k = 0; k < want; ++k) {
                const u128 sum = (u128)(~t[k]) + carry;
                t[k] = (u64)sum;
                carry = (u64)(sum >> 64);
            }
        }
        __syncthreads();
        block_mul_lo(t, want, x, have, x2, want, part);
        if ((int)threadIdx.x < want) x[threadIdx.x] = x2[threadIdx.x];
        __syncthreads();
    }
}
template <int L>
__device__ __forceinline__ Big<L> big_load(const u64* p) {
    Big<L> r;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) r.w[k] = p[k];
    return r;
}
template <int L>
__device__ __forceinline__ void big_store(u64* p, const Big<L>& a) {
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) p[k] = a.w[k];
}
// The m x m matrix N, the pricing pass's products N a_j and the rows' update factors are stored WORD-MAJOR: word k of entry e at
// base[k * entries + e], with the row index running fastest through the entries.  The threads of a wave work on neighbouring
// entries, so a load of word k is one coalesced 512-byte access; with an integer's words side by side (round 3: 1 KB apart from
// lane to lane at 128 limbs) every lane pulled its own cache lines and the passes over N ran at a tenth of the memory's rate.
template <int L>
__device__ __forceinline__ Big<L> big_load_s(const u64* p, size_t stride) {
    Big<L> r;
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) r.w[k] = p[(size_t)k * stride];
    return r;
}
template <int L>
__device__ __forceinline__ void big_store_s(u64* p, size_t stride, const Big<L>& a) {
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < L; ++k) p[(size_t)k * stride] = a.w[k];
}
// sign of a * b - c * d, exactly (2 L limbs): the tie breaker of the ratio test and of the pricing rule
template <int L>
__device__ int sign_of_difference(const Big<L>& a, const Big<L>& b, const Big<L>& c, const Big<L>& d) {
    // magnitudes and signs apart: |a||b| and |c||d| as 2 L-limb unsigned numbers
    auto product = [](const Big<L>& x, const Big<L>& y, u64* out, bool* negative) {
        const bool nx = big_neg(x), ny = big_neg(y);
        const Big<L> mx = nx ? big_negate(x) : x, my = ny ? big_negate(y) : y;
        for (int k = 0; k < 2 * L; ++k) out[k] = 0;
        if (big_zero(x) || big_zero(y)) {  // (the ties of a degenerate pivot are x~_i = 0 against x~_j = 0: no L^2 word products for those)
            *negative = false;
            return true;
        }
        int lx = 0, ly = 0;  // words in use: the integers of a run rarely fill the width its largest one forced
        for (int k = 0; k < L; ++k) {
            if (mx.w[k] != 0) lx = k + 1;
            if (my.w[k] != 0) ly = k + 1;
        }
        for (int i = 0; i < lx; ++i) {
            u64 carry = 0;
            for (int j = 0; j < ly; ++j) {
                const u128 t = (u128)mx.w[i] * my.w[j] + out[i + j] + carry;
                out[i + j] = (u64)t;
                carry = (u64)(t >> 64);
            }
            out[i + ly] += carry;
        }
        bool zero = true;
        for (int k = 0; k < 2 * L; ++k) zero = zero && out[k] == 0;
        *negative = !zero && (nx != ny);
        return zero;
    };
    u64 p1[2 * L], p2[2 * L];
    bool n1 = false, n2 = false;
    const bool z1 = product(a, b, p1, &n1), z2 = product(c, d, p2, &n2);
    const int s1 = z1 ? 0 : (n1 ? -1 : 1), s2 = z2 ? 0 : (n2 ? -1 : 1);
    if (s1 != s2) return s1 > s2 ? 1 : -1;
    if (s1 == 0) return 0;
    int cmp = 0;
    for (int k = 2 * L - 1; k >= 0 && cmp == 0; --k)
        if (p1[k] != p2[k]) cmp = p1[k] > p2[k] ? 1 : -1;
    return s1 > 0 ? cmp : -cmp;
}

// sign of a b - c d for NON-NEGATIVE integers of la, lb, lc, ld words, by one wave: a lane forms the column sums of the words
// w = lane, lane + 64, ... of both products (three words each), then the borrow of the difference runs through the words in order,
// every lane the same chain with the sums fetched from their lanes.  (The exact tie-breaks of the ratio test: one thread forming two
// whole products in scratch memory was two milliseconds per tied pivot at 128 limbs.)
template <int L>
__device__ __forceinline__ int wave_sign_of_difference(const u64* a, int la, const u64* b, int lb, const u64* c, int lc, const u64* d, int ld, int lane) {
    constexpr int SLOTS = (2 * L + WAVE - 1) / WAVE;
    u64 sum[2][SLOTS][3];
    auto column_sums = [&](const u64* x, int lx, const u64* y, int ly, u64 (*out)[3]) {
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            const int w = lane + t * WAVE;
            u64 s0 = 0, s1 = 0, s2 = 0;
            if (lx > 0 && ly > 0 && w < lx + ly) {
                const int j0 = max(0, w - lx + 1), j1 = min(w, ly - 1);
                for (int j = j0; j <= j1; ++j) {
                    const u128 prod = (u128)x[w - j] * y[j];
                    const u128 low = (u128)s0 + (u64)prod;
                    s0 = (u64)low;
                    const u128 mid = (u128)s1 + (u64)(prod >> 64) + (u64)(low >> 64);
                    s1 = (u64)mid;
                    s2 += (u64)(mid >> 64);
                }
            }
            out[t][0] = s0;
            out[t][1] = s1;
            out[t][2] = s2;
        }
    };
    column_sums(a, la, b, lb, sum[0]);
    column_sums(c, lc, d, ld, sum[1]);
    const int words = max(la + lb, lc + ld);
    u128 run[2] = {0, 0};
    u64 run_top[2] = {0, 0};
    u64 borrow = 0;
    bool nonzero = false;
    for (int w = 0; w < words; ++w) {
        const int owner = w & (WAVE - 1), slot = w / WAVE;
        u64 word[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            u64 part[3] = {0, 0, 0};
#pragma unroll
            for (int t = 0; t < SLOTS; ++t)
                if (t == slot) { part[0] = sum[k][t][0]; part[1] = sum[k][t][1]; part[2] = sum[k][t][2]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) part[q] = __shfl(part[q], owner);
            const u128 add = (u128)part[0] | ((u128)part[1] << 64);
            run[k] += add;
            run_top[k] += part[2] + (run[k] < add ? 1 : 0);
            word[k] = (u64)run[k];
            run[k] = (run[k] >> 64) | ((u128)run_top[k] << 64);
            run_top[k] = 0;
        }
        const u64 t = word[0] - word[1];
        const u64 diff = t - borrow;
        borrow = ((word[0] < word[1]) || (t < borrow)) ? 1 : 0;
        nonzero = nonzero || diff != 0;
    }
    return borrow ? -1 : (nonzero ? 1 : 0);
}

// The grid barrier of the cooperative launch: grid_barrier.hpp (per-die counting, one release fence per XCD, a watchdog that ends the
// launch instead of hanging the device when the workgroups' barrier counts ever differ).
// One LDS arena for the steps of a pivot that follow each other behind barriers -- the tournament's keys (41 KB at 128 limbs), the exact
// weights' magnitudes, the entering column's rows, the update's operand images (44 KB since round 6's sixteen copies): as objects of
// their own they added up to 99 KB a workgroup, ONE workgroup per CU instead of two, and the tiles ran at half the waves.
template <int L>
constexpr size_t exact_arena_bytes() { return (size_t)328 * L + 2064 + 64; }  // sizeof(UpdateLds<L>) (asserted where it is used) >= the others
template <int L>
__device__ __forceinline__ unsigned char* exact_arena() {
    __shared__ __attribute__((aligned(16))) unsigned char s_arena[exact_arena_bytes<L>()];
    return s_arena;
}
constexpr int EX_GAMMA_BATCH = 64;  // tied candidates whose exact weights are formed at a time (gamma_terms)
constexpr int EX_PRODUCT_SLOTS = EX_GAMMA_BATCH;  // columns whose exact products N a_j price_a holds at a time

// ---------------------------------------------------------------------------------------------------
// the LP on the device: integer columns (rows scaled), [artificials | provider columns] as in solver.hip
// ---------------------------------------------------------------------------------------------------
struct ExactLP {
    int m, n, n_art, limbs;
    const int* col_start;
    const int* row_index;
    const i64* value;
    const i64* cost2;     // phase-two costs (scaled to integers)
    const i64* cost1;     // phase-one costs: lcm(r over the artificial rows) / r_i on the artificial of row i, 0 elsewhere
    const u64* weight;    // w_j = W sigma_j^2 (see the header): W for a structural column, (lcm(r) / r_i)^2 for a unit column of row i -- two words each,
                          //   low first (round 6: rows that need ten and more decimal digits -- CAPRI, ETAMACRO, FINNIS, STAIR -- make W = lcm(r)^2 67 to 80 bits)
    const i64* rhs;       // scaled right-hand side
    int* basis;           // [m]
    int* pos;             // [n]
    u64* N;               // [limbs][m columns][m rows]: word-major, entry (row i, column c) = c * m + i (see big_load_s)
    u64* D;               // Big (followed by its odd part's inverse, and the scratch of the tie breakers)
    u64* xt;              // [m] Big: x~_B
    u64* alpha;           // [m] Big: alpha~_q
    u64* ctil;            // [n] Big: c~_j
    double* key;          // [n] estimate of c~_j^2 / gamma~_j (0: not a candidate)
    int* trace;           // [4 * trace_capacity]: phase, q, p, leaving
    int trace_capacity;
    long long max_pivots;
    int* out;             // [16]: status, pivots phase one, pivots phase two, limbs, trace entries, redundant rows; after an overflow
                          //       also [6..9] = phase, trace count, drive row, removed rows at the START of the pivot that did not fit
    const int* resume;    // [8]: [0] != 0: continue a run that overflowed at a narrower width (N, D, basis, pos, removed are its state
                          //      before the pivot that did not fit); [1..6] = phase, pivots one, pivots two, trace count, drive row, removed
    int* removed;         // [m] 1: the row is redundant -- its artificial cannot be pivoted out (`RemoveRows` of the reference); [m + i]: such rows above row i
    int* shared_words;    // [16] grid-wide overflow flags, decisions of workgroup 0's thread 0, counters of the candidate lists
    double* part_key;     // [2][grid] per-workgroup partials of the grid arg-max reductions
    unsigned long long* part_rank;
    unsigned long long* prof;  // [EX_PROF_WORDS] the leader's time per step of the loop in ticks of the 100 MHz wall clock [0..9], candidate counts [12],
                               // and the word products (64 x 64 -> 128 bit) of the update of N: [16] those the entries need, [17] those the waves issue
    u64* price_a;         // [limbs][EX_PRODUCT_SLOTS][m]: exact products (N a_j)_i, word-major, of the columns of a list (price_products); scratch of the entering column's chunks
    double* price_err;    // ... and a bound on that share's error (price_estimates: the products are formed from leading words)
    double* price_term;   // ... its share of the steepest-edge estimate
    int* bracket;         // [max(n, m) + 1] the tournament brackets over the candidates
    int* cand;            // [max(n, m) + 1] columns whose key estimate is within 1e-9 of the best (pricing); near-tied rows (ratio test)
    u64* gamma;           // [n][2 limbs + 2] their exact weights
    u64* gamma_terms;     // [EX_GAMMA_BATCH candidates][m + 1][2 limbs + 2] the terms of those sums
    u64* x_part;          // [m][ceil(m / 32)] Big: partial sums of x~_B = N b (first turn of a run); afterwards [limbs][m]: -alpha~_i / D_odd (negated: the update adds), word-major
    int* x_bits;          // ... their bit bounds
    i64* cb_row;          // [m] cost of the basic column of each row in the current phase (pricing pass B)
    int* row_list;        // [2 m] the rows with alpha~_i != 0, in order, and from [m] on the others (the update of N)
    int* col_heavy;       // [m] the columns with N(p, k) != 0, in order, and col_light [m] the others (the update of N).  (Lists of their own:
    int* col_light;       //  in `bracket` / `cand` a workgroup late out of the ratio test's last barrier could have read the winner's slot overwritten)
    int* N_bits;          // [m columns][m rows] bit length of |N(i, c)|, kept by whoever writes an entry (the bounds of the passes over N read 4 bytes instead of the integer)
    // the update of N on the matrix cores (limbs >= 16, see mfma_update_tile): the numerators 2^s N'_ik as the tiles leave them --
    u64* T;               // [limbs][m columns][m rows] words (each 128-bit pair of an entry summed by one lane) ...
    int* T_carry;         // [limbs / 2][m columns][m rows] ... and what a pair carries into the next one
    int* T_words;         // [m columns][m rows] words of the entry that T holds (0: the entry was not touched; reset by the pass that reads T)
    // round 5, pricing through y = c_B' N: c~_j = c_j D - y a_j for every column, the products N a_j only for the columns with c~_j < 0
    u64* y;               // [m][limbs] y_k = sum_i c_B(i) N(i, k), entry-major
    int* y_bits;          // [m] bound on the bit length of y_k
    double* cd;           // [n] c~_j / D as a double (the key's numerator) for the columns with c~_j < 0
    int* neg_list;        // [n] those columns, in no particular order; neg_list[n] = their count
    u64* Tx;              // [limbs][2 m], Tx_carry [limbs / 2][2 m], Tx_words [2 m]: the same for x~_B, the column after the last (entries 0 .. m - 1),
    int* Tx_carry;        //   and for y, one more ROW of N (entries m .. 2 m - 1: y'_k = (alpha~_p y_k + c~_q N(p, k)) / D)
    int* Tx_words;
    u64* y_part;          // [limbs] c~_q / D_odd modulo 2^(64 limbs): y's factor of the update, as x_part holds the rows'
    u64* next_dinv;       // [limbs] 1 / D'_odd for the NEXT pivot, formed by the last workgroup while the others update N (update_on_matrix_cores)
    int* xt_bits;         // [m] bit length of |x~_i|, kept by whoever writes an entry
    int mfma_update;      // 1: the update runs on the matrix cores
    // round 6: the second pass of the update inside the tiles.  N is double-buffered: a pivot's tiles read `N` and write the finished
    // entries -- carries run through, shifted, sign-extended, bit lengths -- into `N_alt`; the kernel then swaps the two (and the bit lengths)
    u64* N_alt;           // [limbs][m columns][m rows] the other buffer (zero at the launch)
    int* N_bits_alt;      // [m columns][m rows] bit lengths of what N_alt holds
    int fused_update;     // 1: the tiles finish their entries themselves (a pivot on a negative element -- zero-level pivots only -- takes the two passes)
    u64* xt_alt;          // [m] Big, xt_bits_alt [m]: the other buffer of x~_B, likewise (a thread walking the 128 words of ONE entry of x~_B behind the
    int* xt_bits_alt;     //   barrier was 0.2 ms of every pivot -- as long as the whole second pass over N, whose latency it had hidden in)
    u64* y_alt;           // [m][limbs], y_bits_alt [m]: and of y = c_B' N
    int* y_bits_alt;
    int price_exactly;    // 1: every column that can enter has its products formed exactly (test hook: the path of a column whose estimate is not good enough)
    unsigned* barrier;    // [EX_BARRIER_WORDS] the grid barrier's counters (grid_barrier), zero at the launch
};

// out = w * v^2, unsigned, 2 L + 2 limbs: one term of the exact weight gamma~_j = w_j D^2 + sum_i w_i (N a_j)_i^2 of a tied candidate
__device__ __forceinline__ double weight_as_double(const ExactLP& lp, int j) {
    return (double)lp.weight[2 * j + 1] * 18446744073709551616.0 + (double)lp.weight[2 * j];
}
template <int L>
__device__ void weighted_square(const Big<L>& v, u64 w, u64 w_high, u64* out) {
    const Big<L> mag = big_neg(v) ? big_negate(v) : v;
    u64 sq[2 * L];
    for (int k = 0; k < 2 * L; ++k) sq[k] = 0;
    for (int i = 0; i < L; ++i) {
        u64 carry = 0;
        for (int t = 0; t < L; ++t) {
            const u128 prod = (u128)mag.w[i] * mag.w[t] + sq[i + t] + carry;
            sq[i + t] = (u64)prod;
            carry = (u64)(prod >> 64);
        }
        sq[i + L] += carry;
    }
    u128 carry = 0;  // (a two-word weight: the product has 2 L + 2 words)
    for (int k = 0; k < 2 * L; ++k) {
        const u128 low = (u128)sq[k] * w + (u64)carry;
        out[k] = (u64)low;
        carry = (u128)sq[k] * w_high + (carry >> 64) + (low >> 64);
    }
    out[2 * L] = (u64)carry;
    out[2 * L + 1] = (u64)(carry >> 64);
}
// |v| of an L-word two's complement integer (word k at v[k * stride]) into out[0 .. L) (LDS), a lane per word: the two's complement of a
// negative value is zero up to its lowest non-zero word, that word's complement plus one, the complements above.  Returns v != 0.
template <int L>
__device__ __forceinline__ bool wave_magnitude(const u64* v, size_t stride, u64* out, int lane) {
    constexpr int SLOTS = (L + WAVE - 1) / WAVE;
    u64 w[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) w[t] = lane + t * WAVE < L ? v[(size_t)(lane + t * WAVE) * stride] : 0ull;
    const u64 top_word = __shfl(w[(L - 1) / WAVE], (L - 1) & (WAVE - 1));
    const bool negative = (i64)top_word < 0;
    int lowest = L;
#pragma unroll
    for (int t = SLOTS - 1; t >= 0; --t) {
        const unsigned long long nonzero = __ballot(lane + t * WAVE < L && w[t] != 0);
        if (nonzero != 0) lowest = t * WAVE + __ffsll((long long)nonzero) - 1;
    }
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
        const int k = lane + t * WAVE;
        if (k < L) out[k] = !negative ? w[t] : k < lowest ? 0ull : k == lowest ? ~w[t] + 1ull : ~w[t];
    }
    return lowest < L;
}
// x^2 for an unsigned integer of L words in LDS, by one wave: a lane per word of the result forms its column sum (three words), one chain
// runs the carries through; emit(w, word) is called by every lane alike, w = 0 .. 2 L - 1 in order
template <int L, class Emit>
__device__ __forceinline__ void wave_square(const u64* x, int lane, Emit emit) {
    constexpr int SLOTS2 = (2 * L + WAVE - 1) / WAVE;
    u64 sum[SLOTS2][3];
#pragma unroll
    for (int t = 0; t < SLOTS2; ++t) {
        const int w = lane + t * WAVE;
        u64 s0 = 0, s1 = 0, s2 = 0;
        if (w < 2 * L) {
            const int j0 = max(0, w - L + 1), j1 = min(w, L - 1);
            for (int j = j0; j <= j1; ++j) {
                const u128 prod = (u128)x[w - j] * x[j];
                const u128 low = (u128)s0 + (u64)prod;
                s0 = (u64)low;
                const u128 mid = (u128)s1 + (u64)(prod >> 64) + (u64)(low >> 64);
                s1 = (u64)mid;
                s2 += (u64)(mid >> 64);
            }
        }
        sum[t][0] = s0;
        sum[t][1] = s1;
        sum[t][2] = s2;
    }
    u128 run = 0;
    u64 run_top = 0;
    for (int w = 0; w < 2 * L; ++w) {
        const int owner = w & (WAVE - 1), slot = w / WAVE;
        u64 part[3] = {0, 0, 0};
#pragma unroll
        for (int t = 0; t < SLOTS2; ++t)
            if (t == slot) { part[0] = sum[t][0]; part[1] = sum[t][1]; part[2] = sum[t][2]; }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)part[q], owner), hi = (unsigned)__builtin_amdgcn_readlane((int)(part[q] >> 32), owner);
            part[q] = ((u64)hi << 32) | lo;
        }
        const u128 add = (u128)part[0] | ((u128)part[1] << 64);
        run += add;
        run_top += part[2] + (run < add ? 1 : 0);
        emit(w, (u64)run);
        run = (run >> 64) | ((u128)run_top << 64);
        run_top = 0;
    }
}
// The exact comparison of two tied candidates' keys c~^2 / gamma~ by ONE WAVE: + 1 when column a has the larger key, as compare_keys (one
// thread forming two squares and two (2 L) x (2 L + 2)-word products in scratch memory: 0.4 ms a comparison at 32 limbs, 60 % of
// BANDM's solve, whose pivots tie two or three candidates as a rule -- and 12 KB of scratch per lane of the whole kernel at 128 limbs).
// The magnitudes |c~| and the weights go to the wave's words in LDS (`room`: 10 L + 4 words); |c~_a| = |c~_b| decides on the weights
// alone; otherwise the squares are formed a lane per word (column sums, the carries in one chain) and the sign of c~_a^2 gamma~_b -
// c~_b^2 gamma~_a by wave_sign_of_difference.
template <int L>
__device__ __forceinline__ int wave_compare_keys(const u64* ca, const u64* gamma_a, const u64* cb, const u64* gamma_b, u64* room, int lane) {
    constexpr int GW = 2 * L + 2, SLOTS = (L + WAVE - 1) / WAVE, SLOTSG = (GW + WAVE - 1) / WAVE;
    u64 *mag_a = room, *mag_b = room + L, *sq_a = room + 2 * L, *sq_b = room + 4 * L, *ga = room + 6 * L, *gb = room + 8 * L + 2;
    const bool a_nonzero = wave_magnitude<L>(ca, 1, mag_a, lane);
    wave_magnitude<L>(cb, 1, mag_b, lane);
#pragma unroll
    for (int t = 0; t < SLOTSG; ++t) {
        const int k = lane + t * WAVE;
        if (k < GW) {
            ga[k] = gamma_a[k];
            gb[k] = gamma_b[k];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool same = true;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) same = same && __ballot(lane + t * WAVE < L && mag_a[lane + t * WAVE < L ? lane + t * WAVE : 0] != mag_b[lane + t * WAVE < L ? lane + t * WAVE : 0]) == 0;
    if (same) {  // (the tied candidates of a degenerate pivot, as a rule): the larger key has the smaller gamma -- no products
        if (!a_nonzero) return 0;
        int result = 0;
#pragma unroll
        for (int t = SLOTSG - 1; t >= 0; --t) {
            const int k = lane + t * WAVE;
            const u64 wa = k < GW ? ga[k] : 0ull, wb = k < GW ? gb[k] : 0ull;
            const unsigned long long differs = __ballot(wa != wb);
            if (result == 0 && differs != 0) {
                const int owner = 63 - __clzll((long long)differs);
                const u64 xa = __shfl(wa, owner), xb = __shfl(wb, owner);
                result = xb > xa ? 1 : -1;
            }
        }
        return result;
    }
    wave_square<L>(mag_a, lane, [&](int w, u64 word) { if (lane == 0) sq_a[w] = word; });
    wave_square<L>(mag_b, lane, [&](int w, u64 word) { if (lane == 0) sq_b[w] = word; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return wave_sign_of_difference<2 * L + 1>(sq_a, 2 * L, gb, GW, sq_b, 2 * L, ga, GW, lane);
}
// The terms w_i (N a_j)_i^2 (and w_j D^2, "row" m) of the exact weights of `batch` tied candidates, a WAVE per term: the magnitude a
// lane per word into the wave's LDS, the square a lane per word of the result (wave_square), the small weight multiplied in as the
// words leave the chain.  (A thread per term with its square in scratch memory: L^2 dependent multiply-adds, a millisecond at 128
// limbs, and 2 KB of the kernel's scratch per lane.)
template <int L>
__device__ __noinline__ void exact_weight_terms(const ExactLP& lp, int c0, int batch) {
    constexpr int GW = 2 * L + 2;
    static_assert(sizeof(u64) * (EX_THREADS / WAVE) * L <= exact_arena_bytes<L>(), "arena");
    u64 (*s_magnitude)[L] = reinterpret_cast<u64 (*)[L]>(exact_arena<L>());
    const int m = lp.m, lane = threadIdx.x & (WAVE - 1);
    const long long wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    u64* magnitude = s_magnitude[threadIdx.x / WAVE];
    for (long long item = wave_of_grid; item < (long long)batch * (m + 1); item += waves_of_grid) {
        const int local = (int)(item / (m + 1)), i = (int)(item - (long long)local * (m + 1));
        const int j = lp.cand[c0 + local];
        u64* out = lp.gamma_terms + ((size_t)local * (m + 1) + i) * GW;
        const int weighted = i == m ? j : lp.basis[i];
        const u64 weight = lp.weight[2 * weighted], weight_high = lp.weight[2 * weighted + 1];
        if (i == m) wave_magnitude<L>(lp.D, 1, magnitude, lane);
        else wave_magnitude<L>(lp.price_a + (size_t)local * m + i, (size_t)EX_PRODUCT_SLOTS * m, magnitude, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u128 carry = 0;
        wave_square<L>(magnitude, lane, [&](int w, u64 word) {
            const u128 low = (u128)word * weight + (u64)carry;
            if (lane == 0) out[w] = (u64)low;
            carry = (u128)word * weight_high + (carry >> 64) + (low >> 64);
        });
        if (lane == 0) {
            out[2 * L] = (u64)carry;
            out[2 * L + 1] = (u64)(carry >> 64);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// one round of the tournament over the tied candidates (the brackets `stride` apart), a wave per comparison; a function of its own for
// its registers' sake (the sums of a 4 L-word product a lane per word)
template <int L>
__device__ __noinline__ void tournament_round(const ExactLP& lp, int n_cand, int stride) {
    static_assert(sizeof(u64) * (EX_THREADS / WAVE) * (10 * L + 4) <= exact_arena_bytes<L>(), "arena");
    u64 (*s_keys)[10 * L + 4] = reinterpret_cast<u64 (*)[10 * L + 4]>(exact_arena<L>());
    const int lane = threadIdx.x & (WAVE - 1);
    const long long wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    for (long long c = wave_of_grid * 2 * stride; c + stride < n_cand; c += waves_of_grid * 2 * stride) {
        const int ca = lp.bracket[c], cb2 = lp.bracket[c + stride];
        const int ja = lp.cand[ca], jb = lp.cand[cb2];
        const int cmp = wave_compare_keys<L>(lp.ctil + (size_t)jb * L, lp.gamma + (size_t)cb2 * (2 * L + 2), lp.ctil + (size_t)ja * L, lp.gamma + (size_t)ca * (2 * L + 2),
                                             s_keys[threadIdx.x / WAVE], lane);
        if (lane == 0 && (cmp > 0 || (cmp == 0 && jb > ja))) lp.bracket[c] = cb2;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}
// c_a^2 * gamma_b  vs  c_b^2 * gamma_a  (unsigned, (4 L + 1) limbs): +1 when column a has the larger key
template <int L>
__device__ int compare_keys(const Big<L>& ca, const u64* gamma_a, const Big<L>& cb, const u64* gamma_b) {
    {   // |c_a| = |c_b| (the tied candidates of a degenerate pivot, as a rule): the larger key has the smaller gamma -- no products
        const Big<L> ma = big_neg(ca) ? big_negate(ca) : ca, mb = big_neg(cb) ? big_negate(cb) : cb;
        bool same = true;
        for (int k = 0; k < L; ++k) same = same && ma.w[k] == mb.w[k];
        if (same) {
            if (big_zero(ma)) return 0;
            for (int k = 2 * L + 1; k >= 0; --k)
                if (gamma_a[k] != gamma_b[k]) return gamma_b[k] > gamma_a[k] ? 1 : -1;
            return 0;
        }
    }
    auto square = [](const Big<L>& v, u64* sq) {
        const Big<L> mag = big_neg(v) ? big_negate(v) : v;
        for (int k = 0; k < 2 * L; ++k) sq[k] = 0;
        for (int i = 0; i < L; ++i) {
            u64 carry = 0;
            for (int t = 0; t < L; ++t) {
                const u128 prod = (u128)mag.w[i] * mag.w[t] + sq[i + t] + carry;
                sq[i + t] = (u64)prod;
                carry = (u64)(prod >> 64);
            }
            sq[i + L] += carry;
        }
    };
    u64 sa[2 * L], sb[2 * L], left[4 * L + 2], right[4 * L + 2];
    square(ca, sa);
    square(cb, sb);
    auto multiply = [](const u64* x, const u64* g, u64* out) {  // x: 2L limbs, g: 2L + 2 limbs
        for (int k = 0; k < 4 * L + 2; ++k) out[k] = 0;
        for (int i = 0; i < 2 * L; ++i) {
            u64 carry = 0;
            for (int t = 0; t < 2 * L + 2; ++t) {
                const u128 prod = (u128)x[i] * g[t] + out[i + t] + carry;
                out[i + t] = (u64)prod;
                carry = (u64)(prod >> 64);
            }
        }
    };
    multiply(sa, gamma_b, left);
    multiply(sb, gamma_a, right);
    for (int k = 4 * L + 1; k >= 0; --k)
        if (left[k] != right[k]) return left[k] > right[k] ? 1 : -1;
    return 0;
}

// sum_e v_e N(i, r_e) over the entries [e0, e1) of one column for 64 neighbouring rows, by one wave, word by word (the first pricing
// pass for every column, the entering column once more): word k of row i's sum goes to out[k * out_stride] (out is this lane's), its
// leading words to `lead`, and the bit bound of the operands (the fit test's) is returned.
// (the bit bound of row i's sum -- the operands' bit lengths, no word of N is read -- and the words of the result that the wave forms:
//  the sum fits `awide` bits, the words above that many are its sign, not worth their operands' loads)
template <int L>
__device__ __forceinline__ int column_products_bound(const ExactLP& lp, int e0, int e1, int i, bool active, int* words_of_wave) {
    constexpr int KU = L >= 8 ? 8 : (L >= 4 ? 4 : (L >= 2 ? 2 : 1));  // words of the result per turn (stream_column_products)
    const int m = lp.m;
    int awide = 0;
    for (int e = e0; e < e1; ++e)
        if (active) awide = max(awide, lp.N_bits[(size_t)lp.row_index[e] * m + i] + small_bits(lp.value[e]));
    awide += 32 - __clz(e1 - e0 > 1 ? e1 - e0 - 1 : 1) + 1;  // (log2_ceil of the kernel)
    int words = active ? min(L, (awide + 2 + 63) / 64) : 1;
    for (int d = 1; d < WAVE; d *= 2) words = max(words, __shfl_xor(words, d));
    *words_of_wave = min(L, (words + KU - 1) / KU * KU);
    return awide;
}
// 192 bits, two's complement: the running sum of the multiples that reach a word of the result and the words above it
struct SignedSum {
    u128 acc = 0;
    i64 top = 0;
    __device__ __forceinline__ void add(u128 v) {
        acc += v;
        top += acc < v ? 1 : 0;
    }
    __device__ __forceinline__ void sub(u128 v) {
        const bool borrows = acc < v;
        acc -= v;
        top -= borrows ? 1 : 0;
    }
    __device__ __forceinline__ void add(const SignedSum& other) {
        acc += other.acc;
        top += other.top + (acc < other.acc ? 1 : 0);
    }
    __device__ __forceinline__ u64 pop() {  // the lowest word leaves, the rest moves down (the sign stays)
        const u64 word = (u64)acc;
        acc = (acc >> 64) | ((u128)(u64)top << 64);
        top >>= 63;
        return word;
    }
};
// CHUNK: only the words [k_begin, k_limit) of the sum are formed, from those words of the operands alone -- word k - k_begin goes to
// out[(k - k_begin) * out_stride] and the three words that the chunk carries into the words above it to the places (k_limit - k_begin)
// + 0, 1, 2 (entering_column_rows adds the chunks up); `lead` is not fed.
template <int L, bool CHUNK = false>
__device__ __forceinline__ int stream_column_products(const ExactLP& lp, int e0, int e1, int i, bool active, int lane, size_t MM, u64* out, size_t out_stride,
                                                       LeadingWords& lead, int k_begin = 0, int k_limit = L) {
    const int m = lp.m;
    int words = 0;
    const int awide = column_products_bound<L>(lp, e0, e1, i, active, &words);
    // The column's entries are read once, one per lane, and handed round with readlane (columns of more than 64 entries read
    // them from memory at every use); two words of the result are formed per turn, the operands of both in flight
    // together -- with one wave per SIMD on the grid the pass waits on memory, not on arithmetic.
    const int len = e1 - e0;
    const bool in_lanes = len <= WAVE;
    int my_offset = 0;  // row_index * m of entry e0 + lane
    i64 my_value = 0;
    if (in_lanes && lane < len) {
        my_offset = lp.row_index[e0 + lane] * m;
        my_value = lp.value[e0 + lane];
    }
    auto entry = [&](int e, int* offset, i64* value) {  // e - e0 uniform over the wave
        if (in_lanes) {
            *offset = __builtin_amdgcn_readlane(my_offset, e);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u64)my_value, e);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((u64)my_value >> 32), e);
            *value = (i64)(((u64)hi << 32) | lo);
        } else {
            *offset = lp.row_index[e0 + e] * m;
            *value = lp.value[e0 + e];
        }
    };
    // (Round 5: ONE signed running sum per word of the turn -- 192 bits, two's complement, the multiples of the negative entries
    //  subtracted -- where there were two unsigned ones and a borrow between them: half the registers, which pay for eight words of
    //  the result per turn instead of four.  The pass waits for memory -- a turn is a round trip per four entries of the column --
    //  and now makes half as many turns.)
    using Sum = SignedSum;
    Sum running;
    const int k_first = CHUNK ? k_begin : 0, k_end = CHUNK ? min(words, k_limit) : words;
    auto emit = [&](int k) {
        const u64 word = running.pop();
        if (active) out[(size_t)(k - k_first) * out_stride] = word;
        if (!CHUNK) lead.feed(k, word);
    };
    constexpr int KU = L >= 8 ? 8 : (L >= 4 ? 4 : (L >= 2 ? 2 : 1));  // words of the result per turn
    for (int k = k_first; k < k_end; k += KU) {
        const u64* word_k = lp.N + (size_t)k * MM + (active ? i : 0);
        Sum next[KU > 1 ? KU - 1 : 1];  // the multiples of the words k + 1 ...
        for (int e = 0; e < len; e += 4) {  // four operands (of every word of the turn) in flight
            u64 w[KU][4];
            i64 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int offset;
                entry(e + u < len ? e + u : len - 1, &offset, &v[u]);
                if (e + u >= len) v[u] = 0;
#pragma unroll
                for (int t = 0; t < KU; ++t) w[t][u] = word_k[(size_t)t * MM + offset];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u64 mag = v[u] < 0 ? (u64)(-(v[u] + 1)) + 1 : (u64)v[u];
#pragma unroll
                for (int t = 0; t < KU; ++t) {
                    Sum& target = t == 0 ? running : next[t > 0 ? t - 1 : 0];
                    if (v[u] >= 0) target.add((u128)w[t][u] * mag);
                    else target.sub((u128)w[t][u] * mag);
                }
            }
        }
        emit(k);
#pragma unroll
        for (int t = 1; t < KU; ++t) {
            running.add(next[t - 1]);
            emit(k + t);
        }
    }
    if (CHUNK) {
        if (active && k_end > k_first) {
            u64* carried = out + (size_t)(k_limit - k_first) * out_stride;
            carried[0] = (u64)running.acc;
            carried[out_stride] = (u64)(running.acc >> 64);
            carried[2 * out_stride] = (u64)running.top;
        }
    } else {
        const u64 fill = (i64)lead.prev < 0 ? ~0ull : 0ull;
#pragma unroll L <= 8 ? L : 1
        for (int k = words; k < L; ++k) {
            if (active) out[(size_t)k * out_stride] = fill;
            lead.feed(k, fill);
        }
    }
    return awide;
}

// Bit length of |v| for an integer of L words side by side in memory, by one wave (a lane per word, two at 128 limbs): the highest word
// that differs from the sign decides.  (One thread scanning its own copy from the top is a chain of L dependent scratch reads: 0.1 ms
// at 128 limbs.)  The same value in every lane.
template <int L>
__device__ __forceinline__ int wave_bit_length_words(const u64 (&w)[(L + WAVE - 1) / WAVE], int lane);
template <int L>
__device__ __forceinline__ int wave_bit_length(const u64* v, int lane) {
    constexpr int SLOTS = (L + WAVE - 1) / WAVE;
    u64 w[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) w[t] = lane + t * WAVE < L ? v[lane + t * WAVE] : 0ull;
    return wave_bit_length_words<L>(w, lane);
}
template <int L>
__device__ __forceinline__ int wave_bit_length_words(const u64 (&w)[(L + WAVE - 1) / WAVE], int lane) {  // (word lane + 64 t in slot t; zero beyond L)
    constexpr int SLOTS = (L + WAVE - 1) / WAVE;
    const u64 top_word = __shfl(w[(L - 1) / WAVE], (L - 1) & (WAVE - 1));
    const bool negative = (i64)top_word < 0;
    const u64 sign = negative ? ~0ull : 0ull;
    int top = -1;  // the highest word that differs from the sign
    u64 at_top = 0;
#pragma unroll
    for (int t = SLOTS - 1; t >= 0; --t) {
        const unsigned long long differs = __ballot(lane + t * WAVE < L && w[t] != sign);
        if (top < 0 && differs != 0) {
            const int owner = 63 - __clzll((long long)differs);
            top = owner + t * WAVE;
            at_top = __shfl(w[t], owner);
        }
    }
    if (!negative) return top < 0 ? 0 : 64 * top + (64 - __clzll((long long)at_top));
    if (top < 0) return 1;  // -1
    // |v| = ~v + 1: the bits of ~v, one more when the + 1 carries into a new bit (v = -(2^k): zeros below, the top word of ~v = 2^j - 1)
    bool zeros_below = true;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) zeros_below = zeros_below && __ballot(lane + t * WAVE < top && w[t] != 0) == 0;
    const u64 inverted = ~at_top;
    int bits = 64 * top + (64 - __clzll((long long)inverted));
    if (zeros_below && (inverted & (inverted + 1)) == 0) bits += 1;
    return bits;
}

// sum_i cb(i) X_i for one word-major column of m integers (word k of row i at column[k * word_stride + i]), by one wave: every lane
// streams the words of its rows through its own carry-save accumulators, the 64 partial words are added across the wave as two sums of
// 32-bit halves; emit(k, word) is called by every lane with word k of the sum (two's complement, modulo 2^(64 L)).
template <int L, class Emit>
__device__ __forceinline__ void wave_cost_dot(const u64* column, size_t word_stride, const i64* cb_row, int m, int lane, int words, Emit emit) {
    // (`words`: the sum fits that many words -- the caller's bit bound; the words above are its sign.  Sixteen rows of a word are
    //  requested together: with four, a word of an 821-row column was four round trips one after the other, 1.3 ms per column.)
    u128 acc_p = 0, acc_q = 0;
    u64 top_p = 0, top_q = 0, borrow_lane = 0;
    u128 carry_sum = 0;
    u64 last = 0;
    words = min(words, L);
#pragma unroll L <= 8 ? L : 1
    for (int k = 0; k < words; ++k) {
        const u64* word_k = column + (size_t)k * word_stride;
        for (int i0 = lane; i0 < m; i0 += 16 * WAVE) {
            u64 w[16];
            i64 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + u * WAVE;
                v[u] = i < m ? cb_row[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = i0 + u * WAVE;
                w[u] = (i < m && v[u] != 0) ? word_k[i] : 0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const u64 mag = v[u] < 0 ? (u64)(-(v[u] + 1)) + 1 : (u64)v[u];
                const u128 prod = (u128)w[u] * mag;
                if (v[u] >= 0) {
                    acc_p += prod;
                    top_p += acc_p < prod ? 1 : 0;
                } else {
                    acc_q += prod;
                    top_q += acc_q < prod ? 1 : 0;
                }
            }
        }
        const u64 pk = (u64)acc_p, qk = (u64)acc_q;
        acc_p = (acc_p >> 64) | ((u128)top_p << 64);
        acc_q = (acc_q >> 64) | ((u128)top_q << 64);
        top_p = top_q = 0;
        const u64 t = pk - qk;
        const u64 mine = t - borrow_lane;  // word k of this lane's rows (two's complement, modulo 2^(64 L))
        borrow_lane = ((pk < qk) || (t < borrow_lane)) ? 1 : 0;
        u64 lo = mine & 0xffffffffull, hi = mine >> 32;
        for (int d = 1; d < WAVE; d *= 2) {
            lo += __shfl_xor(lo, d);
            hi += __shfl_xor(hi, d);
        }
        const u128 total = carry_sum + lo + ((u128)hi << 32);
        carry_sum = total >> 64;
        last = (u64)total;
        emit(k, last);
    }
    const u64 fill = (i64)last < 0 ? ~0ull : 0ull;
    for (int k = words; k < L; ++k) emit(k, fill);
}

// c~_j = c_j D - sum_e v_e y[r_e] for one column, by one wave, a lane per word of the sum (two words per lane at 128 limbs): the
// lane adds the multiples of ITS word of the y's in carry-save form, then the carries, the multiple of D and the difference run
// through the words in order (every lane the same scalar chain, the partial words fetched from their lanes).  Stores c~_j; returns
// c~_j / D as a double when c~_j < 0 (else 0) and the bit bound of the operands.
template <int L>
__device__ __forceinline__ double reduced_cost_wave(const ExactLP& lp, const u64* D, int j, int phase, int lane, double mD, int eD, int D_bits, int* bits_bound) {
    constexpr int SLOTS = (L + WAVE - 1) / WAVE;  // words per lane
    const i64 cj = phase == 1 ? lp.cost1[j] : lp.cost2[j];
    const u64 cj_mag = cj < 0 ? (u64)(-(cj + 1)) + 1 : (u64)cj;
    const int e0 = lp.col_start[j], e1 = lp.col_start[j + 1];
    u128 acc_p[SLOTS], acc_q[SLOTS];
    u64 top_p[SLOTS], top_q[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) { acc_p[t] = acc_q[t] = 0; top_p[t] = top_q[t] = 0; }
    // (The column's entries are read once, one per lane -- row, value and the bit length of that y -- and handed round with readlane;
    //  the words of four y's are requested together.  Entry after entry, each with its row index, then y's bit length, then y's words
    //  one after the other, a column of seven entries was fourteen round trips: a third of this pass.)
    const int len = e1 - e0;
    const bool in_lanes = len <= WAVE;
    int my_row = 0, my_bits = 0;
    i64 my_value = 0;
    if (in_lanes && lane < len) {
        my_row = lp.row_index[e0 + lane];
        my_value = lp.value[e0 + lane];
        my_bits = lp.y_bits[my_row] + small_bits(my_value);
    }
    int widest = my_bits;
    if (in_lanes) {
        for (int d = 1; d < WAVE; d *= 2) widest = max(widest, __shfl_xor(widest, d));
    } else {
        for (int e = e0; e < e1; ++e) widest = max(widest, lp.y_bits[lp.row_index[e]] + small_bits(lp.value[e]));
    }
    auto entry = [&](int e, int* row, i64* value) {  // e uniform over the wave
        if (in_lanes) {
            *row = __builtin_amdgcn_readlane(my_row, e);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u64)my_value, e);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((u64)my_value >> 32), e);
            *value = (i64)(((u64)hi << 32) | lo);
        } else {
            *row = lp.row_index[e0 + e];
            *value = lp.value[e0 + e];
        }
    };
    for (int e = 0; e < len; e += 4) {
        u64 w[4][SLOTS];
        i64 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int r;
            entry(e + u < len ? e + u : len - 1, &r, &v[u]);
            if (e + u >= len) v[u] = 0;
#pragma unroll
            for (int t = 0; t < SLOTS; ++t) {
                const int k = lane + t * WAVE;
                w[u][t] = k < L ? lp.y[(size_t)r * L + k] : 0ull;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u64 mag = v[u] < 0 ? (u64)(-(v[u] + 1)) + 1 : (u64)v[u];
#pragma unroll
            for (int t = 0; t < SLOTS; ++t) {
                const u128 prod = (u128)w[u][t] * mag;
                if (v[u] >= 0) {
                    acc_p[t] += prod;
                    top_p[t] += acc_p[t] < prod ? 1 : 0;
                } else {
                    acc_q[t] += prod;
                    top_q[t] += acc_q[t] < prod ? 1 : 0;
                }
            }
        }
    }
    widest += 32 - __clz(e1 - e0 > 1 ? e1 - e0 - 1 : 1) + 1;
    widest = max(widest, D_bits + small_bits(cj));
    *bits_bound = widest;
    // c~_j = c_j D - sum, a lane per word (two at 128 limbs).  Lane k's share is the signed four-word value V_k = c_j D_k - P_k + Q_k
    // (P, Q: its carry-save sums of the positive and of the negative multiples); word k of the result is the sum of word 0 of V_k,
    // word 1 of V_(k-1), word 2 of V_(k-2), word 3 of V_(k-3), less one for a negative V_(k-4) -- its neighbours' registers -- and what is
    // carried on, small and of either sign, goes from lane to lane until none is left (a round or two; through a run of 0x00.. / 0xFF..
    // words one more per word).  (Rounds 4-5 ran ONE chain through the words, every lane the same scalar arithmetic: 150 instructions a
    // word, 60 to 120 us for a column at 128 limbs -- most of the reduced-cost pass.)
    u64 value[SLOTS][4];
    bool value_negative[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
        const int k = lane + t * WAVE;
        const u64 d_word = k < L ? D[k] : 0ull;
        const u128 multiple = (u128)d_word * cj_mag;
        u64 v[4] = {0, 0, 0, 0};
        auto add_words = [&](u64 a0, u64 a1, u64 a2) {  // v += a0 + 2^64 a1 + 2^128 a2
            u128 sum = (u128)v[0] + a0;
            v[0] = (u64)sum;
            sum = (u128)v[1] + a1 + (u64)(sum >> 64);
            v[1] = (u64)sum;
            sum = (u128)v[2] + a2 + (u64)(sum >> 64);
            v[2] = (u64)sum;
            v[3] += (u64)(sum >> 64);
        };
        auto sub_words = [&](u64 a0, u64 a1, u64 a2) {  // v -= a0 + 2^64 a1 + 2^128 a2
            u128 diff = (u128)v[0] - a0;
            v[0] = (u64)diff;
            diff = (u128)v[1] - a1 - ((u64)(diff >> 64) & 1ull);
            v[1] = (u64)diff;
            diff = (u128)v[2] - a2 - ((u64)(diff >> 64) & 1ull);
            v[2] = (u64)diff;
            v[3] -= (u64)(diff >> 64) & 1ull;
        };
        if (cj >= 0) add_words((u64)multiple, (u64)(multiple >> 64), 0ull);
        else sub_words((u64)multiple, (u64)(multiple >> 64), 0ull);
        sub_words((u64)acc_p[t], (u64)(acc_p[t] >> 64), top_p[t]);
        add_words((u64)acc_q[t], (u64)(acc_q[t] >> 64), top_q[t]);
#pragma unroll
        for (int c = 0; c < 4; ++c) value[t][c] = v[c];
        value_negative[t] = (i64)v[3] < 0;
    }
    // word k - d of a per-word quantity, in the lane of word k (zero below word 0): slot t's comes from the lane d below, or -- in the
    // first d lanes -- from the top lanes of the slot below
    auto from_below = [&](const u64 (&x)[SLOTS], int d, u64 (&out)[SLOTS]) {
        u64 wrapped[SLOTS];
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) wrapped[t] = __shfl(x[t], (lane - d) & (WAVE - 1));
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) out[t] = lane >= d ? wrapped[t] : (t > 0 ? wrapped[t > 0 ? t - 1 : 0] : 0ull);
    };
    u64 word[SLOTS];
    i64 carried[SLOTS];  // what word k hands to word k + 1
    {
        u64 part[4][SLOTS], neg[SLOTS], shifted[SLOTS];
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            part[0][t] = value[t][0];
            neg[t] = value_negative[t] ? 1ull : 0ull;
        }
#pragma unroll
        for (int c = 1; c < 4; ++c) {
#pragma unroll
            for (int t = 0; t < SLOTS; ++t) shifted[t] = value[t][c];
            from_below(shifted, c, part[c]);
        }
        u64 neg_below[SLOTS];
        from_below(neg, 4, neg_below);
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            const __int128 sum = (__int128)((u128)part[0][t] + part[1][t] + part[2][t] + part[3][t]) - (__int128)neg_below[t];
            word[t] = (u64)sum;
            carried[t] = (i64)(sum >> 64);
        }
    }
    for (;;) {
        u64 out_going[SLOTS], incoming[SLOTS];
        bool any = false;
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            out_going[t] = (u64)carried[t];
            any = any || (carried[t] != 0 && lane + t * WAVE + 1 < L);  // (what leaves the last word is dropped: modulo 2^(64 L))
        }
        if (__ballot(any) == 0) break;
        from_below(out_going, 1, incoming);
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            const __int128 sum = (__int128)(u128)word[t] + (__int128)(i64)incoming[t];
            word[t] = (u64)sum;
            carried[t] = (i64)(sum >> 64);
        }
    }
#pragma unroll
    for (int t = 0; t < SLOTS; ++t)
        if (lane + t * WAVE < L) lp.ctil[(size_t)j * L + lane + t * WAVE] = word[t];
    // D > 0: the sign of c~_j is the sign of the relative cost; a negative one as a double from the two leading words of its magnitude
    // (~w + 1: zero up to the lowest non-zero word, that word's two's complement, the others' complements)
    const u64 top_word = __shfl(word[(L - 1) / WAVE], (L - 1) & (WAVE - 1));
    if ((i64)top_word >= 0) return 0.0;
    int lowest = L;
#pragma unroll
    for (int t = SLOTS - 1; t >= 0; --t) {
        const unsigned long long nonzero = __ballot(lane + t * WAVE < L && word[t] != 0);
        if (nonzero != 0) lowest = t * WAVE + __ffsll((long long)nonzero) - 1;
    }
    u64 magnitude[SLOTS];
    int top = -1;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
        const int k = lane + t * WAVE;
        magnitude[t] = k >= L || k < lowest ? 0ull : k == lowest ? ~word[t] + 1ull : ~word[t];
    }
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
        const unsigned long long nonzero = __ballot(magnitude[t] != 0);
        if (nonzero != 0) top = t * WAVE + 63 - __clzll((long long)nonzero);
    }
    u64 m_top = 0, m_below = 0;
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) {
        const u64 a = __shfl(magnitude[t], top & (WAVE - 1)), b = __shfl(magnitude[t], (top - 1) & (WAVE - 1));
        if (t == top / WAVE) m_top = a;
        if (top > 0 && t == (top - 1) / WAVE) m_below = b;
    }
    double x = (double)m_top;  // (LeadingWords::mantissa's numbers)
    if (top > 0) x = x * 18446744073709551616.0 + (double)m_below;
    return ldexp(-x / mD, 64 * (top > 0 ? top - 1 : 0) - eD);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 5: the update of N on the matrix cores.  2^s N'_ik = (alpha~_p u) N_ik + (-alpha~_i u) N_pk modulo 2^(64 L) is, in the bytes of
// the integers, a product with Toeplitz matrices: digit d of x * y is sum_k x_k y_(d-k).  For one column k and 16 rows a wave forms
//     C[digit d][entry e] = sum_k T_A[d][k] E1[k][e] + sum_k T_K[d][k] E2[k][e]
// with v_mfma_i32_16x16x64_i8: E1 = the bytes of the entries N_ik, E2 = the bytes of the rows' factors -alpha~_i u, T_A[d][k] = byte
// d - k of alpha~_p u (zero for d < k), T_K the same of N_pk -- the same matrix for every entry (T_A) or for every entry of a column
// (T_K), so its fragments come out of a few KB of LDS (four copies of the reversed byte string, one per alignment modulo 4).  The
// instruction multiplies SIGNED bytes: every byte travels as b ^ 0x80 = b - 128 and the identity
//     sum_k u_k t_k = sum_k (u_k - 128)(t_k - 128) + 128 sum_k (u_k - 128) + 128 sum_k t_k
// gives the unsigned sum back: the middle term is one more MFMA per block of k against a matrix of ones, the last one a prefix sum of
// the bytes of alpha~_p u / N_pk (per digit, kept in LDS).  The 16 rows of a tile are mapped to digits so that a lane ends up with 16
// neighbouring digits of one entry (row 4 G + r of tile t' <-> digit 64 B + 16 G + 4 t' + r): it adds them into two 64-bit words and
// what they carry on, and a second pass (a thread per entry) runs the carries through, shifts, negates and stores.  All integer,
// all exact: the sums stay below 2^28 in the 32-bit accumulators.  2.39 P MAC/s of i8 are 37 T word products/s; the multiplier of
// the vector unit (v_mad_u64_u32, full rate) gives 2.1 T/s to the compiled block products above and the update ran at 0.34
// (tools/micro/intmul_rates.hip, profiles/r5_micro_intmul_rates.txt).
typedef int v4i __attribute__((ext_vector_type(4)));
// -DRELP_TILE_VARIANT=n (tools/tile_bench.py: the tile by itself with one of its resources taken out -- results are WRONG, only the time
// means something): 1 no stores, 2 no Toeplitz fragments from LDS, 3 no entries from memory, 4 no MFMAs, 5 no epilogue
#ifndef RELP_TILE_VARIANT
#define RELP_TILE_VARIANT 0
#endif
#ifndef RELP_UPDATE_PASS_BLOCKS
#define RELP_UPDATE_PASS_BLOCKS 4
#endif
constexpr int UPDATE_PASS_BLOCKS = RELP_UPDATE_PASS_BLOCKS;  // (build-time A/B: 4 halves the accumulators of a tile and doubles its passes)

template <int L>
struct alignas(16) UpdateLds {
    static constexpr int WB = 8 * L;        // bytes of an integer
    static constexpr int STRIDE = WB + 64;  // of one copy: index y <-> byte WB - 1 - y of the integer, zeros above
    int prefix[2][WB];                      // [operand] 128 * sum_{x <= d} byte x      (operand 0: alpha~_p u, 1: N(p, k))
    // Round 6: SIXTEEN copies, one per byte shift, so that every fragment is one 16-BYTE-ALIGNED ds_read_b128.  Rounds 5 kept four copies
    // (shifts 0..3) and read at 4-byte-aligned addresses: the LDS pipe takes such a read apart -- 6.7-7.4 ns per fragment per CU against 2.2
    // aligned with the tile's own lane pattern (tools/micro/lds_fragment_bench.hip, profiles/r6_micro_lds_fragment.txt), 260 us of LDS time
    // in every pivot of 25FV47 at 128 limbs, half of the tile phase.  35 KB at 128 limbs (two workgroups per CU: 160 KB of LDS).
    unsigned toeplitz[2][16][STRIDE / 4];   // [operand][shift s][..]: byte y of copy s = (byte WB - 1 - (y + s) of the integer) ^ 0x80
    u64 words[L];                           // N(p, k) on its way in
    int scan[EX_THREADS / WAVE];
};

// the LDS image of one Toeplitz operand from its words (in LDS), by the whole workgroup; the caller puts barriers around it
template <int L>
__device__ __forceinline__ void build_toeplitz(UpdateLds<L>& lds, int op, const u64* xw) {
    constexpr int WB = UpdateLds<L>::WB, STRIDE = UpdateLds<L>::STRIDE;
    const int tid = threadIdx.x, T = blockDim.x;
    const unsigned char* xb = (const unsigned char*)xw;
    for (int q = tid; q < 4 * STRIDE; q += T) {  // (16 copies of STRIDE / 4 dwords)
        const int shift = q / (STRIDE / 4), y = 4 * (q - shift * (STRIDE / 4));
        unsigned v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int index = y + shift + j;
            const unsigned byte = index < WB ? xb[WB - 1 - index] : 0u;
            v |= (byte ^ 0x80u) << (8 * j);
        }
        lds.toeplitz[op][shift][y / 4] = v;
    }
    // 128 * inclusive prefix sums of the bytes: four bytes per thread, a scan over the threads
    const bool mine = 4 * tid < WB;
    int s[4] = {0, 0, 0, 0};
    if (mine) {
        const unsigned v = ((const unsigned*)xw)[tid];
        s[0] = v & 255;
        s[1] = s[0] + ((v >> 8) & 255);
        s[2] = s[1] + ((v >> 16) & 255);
        s[3] = s[2] + (v >> 24);
    }
    int inclusive = s[3];
    const int lane = tid & (WAVE - 1), wave = tid / WAVE;
    for (int d = 1; d < WAVE; d *= 2) {
        const int other = __shfl_up(inclusive, d);
        if (lane >= d) inclusive += other;
    }
    if (lane == WAVE - 1) lds.scan[wave] = inclusive;
    __syncthreads();
    int before = inclusive - s[3];
    for (int wv = 0; wv < wave; ++wv) before += lds.scan[wv];
    if (mine) {
#pragma unroll
        for (int j = 0; j < 4; ++j) lds.prefix[op][4 * tid + j] = 128 * (before + s[j]);
    }
}

// One tile: column k, the rows `row` of the 16 entries (lane & 15 selects the entry; row < 0: none), `terms` = 2 where N(p, k) != 0
// and the rows have alpha~_i != 0, else 1 (the entry is only rescaled).  `nb64`: 64-byte blocks of the result that are formed (the
// largest need among the 16 entries).  Writes the words and carries of the numerators to lp.T / lp.T_carry; returns the MFMAs issued.
// (A function of its own, not inlined: its 128 accumulator registers are allocated apart from the loop's state, which the compiler
//  otherwise spills around every MFMA; the LDS operands arrive as address-space pointers so that their loads stay ds_read.)
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef __attribute__((address_space(3))) v4i lds_v4i;
struct UpdateTileArgs {
    const u64* x_part;  // the rows' factors -alpha~_i u, word-major with stride m
    int m;
    unsigned long long* stamps;  // diagnostic build (-DRELP_TILE_STAMPS): cycles of one wave inside its tiles, else unused
};
#ifdef RELP_TILE_STAMPS
#define TILE_STAMP(k) do { if (lane == 0 && blockIdx.x == 0 && threadIdx.x < 64) { const unsigned long long t__ = clock64(); lp.stamps[k] += t__ - t_tile; t_tile = t__; } } while (0)
#else
#define TILE_STAMP(k) do {} while (0)
#endif
// what a lane reads and writes for its entry: word w of the entry at entry[w * entry_stride], of the numerator at numerator[w * numerator_stride]
// (pointers into device memory, and typed so: through generic pointers every load of the ring below was a flat_load that the compiler
//  could only wait for with vmcnt(0) -- everything in flight, the prefetched steps included)
typedef __attribute__((address_space(1))) u64 global_u64;
typedef __attribute__((address_space(1))) const u64 global_cu64;
typedef __attribute__((address_space(1))) int global_i32;
struct UpdateTileEntry {
    global_cu64* entry;
    size_t entry_stride;
    global_cu64* second;  // the factor of the second term (two-term tiles): the row's -alpha~_i u of x_part; N(p, k) for an entry of y
    size_t second_stride;
    global_u64* numerator;
    global_i32* carry;
    global_i32* words;
    size_t numerator_stride;
    __device__ UpdateTileEntry(const u64* entry_, size_t entry_stride_, const u64* second_, size_t second_stride_, u64* numerator_, int* carry_, int* words_, size_t numerator_stride_)
        : entry((global_cu64*)entry_), entry_stride(entry_stride_), second((global_cu64*)second_), second_stride(second_stride_), numerator((global_u64*)numerator_),
          carry((global_i32*)carry_), words((global_i32*)words_), numerator_stride(numerator_stride_) {}
};
// (the fields of UpdateTileEntry one by one: as a struct the argument travelled through the stack, and every tile began by waiting for
//  four scratch loads before its first request could go out -- a third of the time of the "requests" section, tools: -DRELP_TILE_STAMPS)
// FUSED (round 6): the tile finishes its entries itself -- `numerator_of` is then word 0 of the entry in the OTHER buffer of N (stride
// `numerator_stride`), `words_of` its bit length there, `shift` = ctz(D) -- see the epilogue below.
template <int L, bool FUSED>
__device__ __noinline__ int mfma_update_tile_fields(unsigned long long* stamps_of_lp, global_cu64* entry_of, size_t entry_stride, global_cu64* second_of, size_t second_stride,
                                                    global_u64* numerator_of, global_i32* carry_of, global_i32* words_of, size_t numerator_stride, const lds_u32* toeplitz_lds,
                                                    const lds_i32* prefix_lds, int row, bool store, int terms, int nb64, int lane, int shift) {
    struct {
        unsigned long long* stamps;
    } lp{stamps_of_lp};
    struct {
        global_cu64* entry;
        size_t entry_stride;
        global_cu64* second;
        size_t second_stride;
        global_u64* numerator;
        global_i32* carry;
        global_i32* words;
        size_t numerator_stride;
    } at_entry{entry_of, entry_stride, second_of, second_stride, numerator_of, carry_of, words_of, numerator_stride};
    (void)lp;
    constexpr int WB = UpdateLds<L>::WB, STRIDE = UpdateLds<L>::STRIDE;
    // (the same for every lane -- and said so: left as vector values, every "is this block wanted" below became an exec mask with a
    //  full wait on LDS behind it instead of a scalar branch)
    nb64 = __builtin_amdgcn_readfirstlane(nb64);
    terms = __builtin_amdgcn_readfirstlane(terms);
#ifdef RELP_TILE_STAMPS
    unsigned long long t_tile = clock64();
#endif
    const int g = lane >> 4;                         // k-group of the operands, digit group of the results
    const int gq = (lane & 15) >> 2, rq = lane & 3;  // this lane's row of the Toeplitz tile: 4 gq + rq
    const bool have = row >= 0;
    const v4i ones = {0x01010101, 0x01010101, 0x01010101, 0x01010101};
    int issued = 0;
    constexpr int PB = UPDATE_PASS_BLOCKS;  // blocks of 64 digits per pass: 4 PB accumulator tiles
    // ---- FUSED: what the epilogue carries from pass to pass (see below) ----
    int k_below = 0;           // (lanes g == 0) what the pair below this pass's first carries on, as the tiles formed it
    unsigned long long carry_below_mask = 0;  // ... and the bits that the additions below hand on (bits 0..15: entry e; the same in every lane)
    u64 pending = 0;           // (lanes g == 3) the pass's top word, not yet shifted: its upper neighbour belongs to the next pass
    int top_nonzero = -1, top_not_ones = -1, lowest_nonzero = 4 * L;  // of the result's words this lane has formed
    u64 word_nonzero = 0, word_not_ones = 0;
    // (what the other buffer holds at this entry's place, asked for now: at the end of the tile the two loads were a round trip nothing hid)
    int stale_bits = 0;
    u64 stale_top = 0;
    if constexpr (FUSED) {
        if (store) {
            stale_bits = *at_entry.words;
            stale_top = at_entry.numerator[(size_t)(L - 1) * at_entry.numerator_stride];
        }
    }
    const int bs = shift & 63, ws = shift >> 6;  // (D = 2^s D_odd: decimal data scaled to integers leave hundreds of factors 2 in it)
    auto shifted = [&](u64 w, u64 above) { return bs ? (w >> bs) | (above << (64 - bs)) : w; };
    auto emit = [&](int j, u64 v) {  // word j + ws of the numerator, shifted = word j of the new entry: tracked for the bit length, stored
        if (j < 0) return;  // (the numerator's low words: zero, the division is exact)
        if (v != 0) { top_nonzero = j; word_nonzero = v; lowest_nonzero = min(lowest_nonzero, j); }
        if (v != ~0ull) { top_not_ones = j; word_not_ones = v; }
        if (store && RELP_TILE_VARIANT != 1) at_entry.numerator[(size_t)j * at_entry.numerator_stride] = v;
    };
    for (int bp = 0; bp < nb64; bp += PB) {
        v4i acc[PB][4];
#pragma unroll
        for (int bl = 0; bl < PB; ++bl)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq) {
                if constexpr (FUSED) {
                    // (the accumulators START from the offset-binary corrections 128 * sum of the operands' bytes -- read here, behind the
                    //  pass's first requests, instead of in the epilogue; sums stay below 2^29)
                    const lds_v4i* pa = (const lds_v4i*)(prefix_lds + 64 * min(bp + bl, L / 8 - 1) + 16 * g + 4 * tq);
                    v4i start = pa[0];
                    if (terms == 2) start += pa[WB / 4];
                    acc[bl][tq] = start;
                } else {
                    acc[bl][tq] = v4i{0, 0, 0, 0};
                }
            }
        const int kb_end = min(nb64, bp + PB);
        // The steps of the pass are (term, block of 64 bytes of the entries' operand) in order; eight steps' words are in flight at any
        // time (a ring of registers): with one block requested per step the tile waited for memory at every step.
        const int steps = terms * kb_end;
        auto fetch = [&](int step, u64& w0, u64& w1) {
            const int term = step >= kb_end ? 1 : 0, kb = step - term * kb_end;
            global_cu64* src = term == 0 ? at_entry.entry : at_entry.second;
            const size_t stride = term == 0 ? at_entry.entry_stride : at_entry.second_stride;
            const bool wanted = have && step < steps && RELP_TILE_VARIANT != 3;
            w0 = wanted ? src[(size_t)(8 * kb + 2 * g) * stride] : (u64)step;
            w1 = wanted ? src[(size_t)(8 * kb + 2 * g + 1) * stride] : (u64)lane;
        };
        u64 ring[8][2];
#pragma unroll
        for (int j = 0; j < 8; ++j) fetch(j, ring[j][0], ring[j][1]);
        v4i ones_acc = {0, 0, 0, 0};
        TILE_STAMP(26);  // the requests of the first eight steps are out
        for (int step0 = 0; step0 < steps; step0 += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int step = step0 + j;
                if (step < steps) {
                    const int term = step >= kb_end ? 1 : 0, kb = step - term * kb_end;
                    const lds_u32* image = toeplitz_lds + (term * 16 + (3 - rq)) * (STRIDE / 4);  // copies 4 (3 - tq) + 3 - rq of this term's operand
                    if (kb == 0) ones_acc = v4i{0, 0, 0, 0};
                    v4i entries;  // bytes 64 kb + 16 g .. + 15 of this lane's entry, each minus 128
                    entries[0] = (int)((unsigned)ring[j][0] ^ 0x80808080u);
                    entries[1] = (int)((unsigned)(ring[j][0] >> 32) ^ 0x80808080u);
                    entries[2] = (int)((unsigned)ring[j][1] ^ 0x80808080u);
                    entries[3] = (int)((unsigned)(ring[j][1] >> 32) ^ 0x80808080u);
                    fetch(step + 8, ring[j][0], ring[j][1]);
                    ones_acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(ones, entries, ones_acc, 0, 0, 0);
                    ++issued;
                    // row 4 gq + rq of tile tq of block b is digit d = 64 b + 16 gq + 4 tq + rq; its k-th byte is byte d - (64 kb + 16 g + j)
                    // of the integer = index WB - 1 - d + 64 kb + 16 g + j of the reversed string: a 16-aligned offset in copy 4 (3 - tq) + 3 - rq.
                    // The four fragments of block b + 1 are requested before the MFMAs of block b are issued (two sets of registers,
                    // by the parity of the block): with a block's loads and its MFMAs back to back every block waited out the LDS
                    // latency, 13 of the 27 us of a tile.
                    v4i toeplitz[2][4];
                    auto request = [&](int bl, v4i* into) {
                        const int base = WB - 16 - 64 * (bp + bl - kb) - 16 * (gq - g);
#pragma unroll
                        for (int tq = 0; tq < 4; ++tq) {  // (the byte offset 4 (3 - tq) + 3 - rq is the copy's shift: the address is 16-byte aligned)
                            const lds_v4i* at = (const lds_v4i*)(image + 4 * (3 - tq) * (STRIDE / 4) + base / 4);
                            if (RELP_TILE_VARIANT == 2) into[tq] = v4i{base, tq, bl, lane};
                            else into[tq] = *at;
                        }
                    };
                    auto wanted = [&](int bl) { return bp + bl >= kb && bp + bl < nb64; };
                    if (wanted(0)) request(0, toeplitz[0]);
#pragma unroll
                    for (int bl = 0; bl < PB; ++bl) {
                        if (bl + 1 < PB && wanted(bl + 1)) request(bl + 1, toeplitz[(bl + 1) & 1]);
                        if (wanted(bl)) {
#pragma unroll
                            for (int tq = 0; tq < 4; ++tq) {
                                if (RELP_TILE_VARIANT == 4) acc[bl][tq] += toeplitz[bl & 1][tq] ^ entries;
                                else acc[bl][tq] = __builtin_amdgcn_mfma_i32_16x16x64_i8(toeplitz[bl & 1][tq], entries, acc[bl][tq], 0, 0, 0);
                            }
                            issued += 4;
                            if (bp + bl == kb) {  // the block is complete for this term: 128 * (sum of the entry's bytes - 128 each, so far)
#pragma unroll
                                for (int tq = 0; tq < 4; ++tq)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) acc[bl][tq][r] += 128 * ones_acc[r];
                            }
                        }
                    }
                }
            }
        }
        TILE_STAMP(27);  // the steps of the pass
        if (RELP_TILE_VARIANT == 5) {  // (diagnostic: the accumulators are used, nothing else happens)
            int any = 0;
#pragma unroll
            for (int bl = 0; bl < PB; ++bl)
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) any |= acc[bl][tq][0] ^ acc[bl][tq][1] ^ acc[bl][tq][2] ^ acc[bl][tq][3];
            if (any == 0x7fffffff && store) at_entry.numerator[0] = 1;
        } else if constexpr (FUSED) {
            // Round 6: the second pass in the tile's own registers.  Lane (entry e = lane & 15, group g = lane >> 4) holds, for block b of the
            // pass, the 128-bit pair P = 4 b + g of its entry (digits 64 b + 16 g ..) and what it carries on, k_P < 2^22.  The finished pair is
            // V_P + k_(P-1) + c_P where c_P is the carry bit of the additions below -- pair P - 1 sits in the lane 16 below (g > 0) or, for
            // g = 0, in lane e + 48 of the block below (the previous pass's last block: k_below, carry_below_mask).  The bits c_P are a
            // carry-lookahead over (generate, propagate) flags gathered with ballots: no chain through the pairs, no numerator in memory.
            // Then the arithmetic shift by s = ctz(D) bits (whole words by the index, the rest with a word's upper neighbour from the lane 16 above; a pass's top word waits
            // for the next pass), the words straight into the OTHER buffer of N, the sign's fill above them at the end.
            const int blocks_here = min(PB, nb64 - bp);  // (the same for every lane)
            u64 lo[PB], hi[PB];
            int kout[PB];
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) {
                // sixteen digits d_0 .. d_15 (each below 2^29, non-negative) -> sum d_i 2^(8 i): four quads by multiply-adds (v_mad_u64_u32 is
                // full rate), two words and what they carry on
                u64 quad[4];
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const v4i digit = acc[bl][tq];
                    u64 q = (u64)(unsigned)digit[3] * 256u + (unsigned)digit[2];
                    q = q * 256u + (unsigned)digit[1];   // (q < 2^45: the products stay in 64 bits)
                    q = q * 256u + (unsigned)digit[0];
                    quad[tq] = q;                        // < 2^53
                }
                const u64 low0 = quad[0] + (quad[1] << 32);
                const u64 high0 = (quad[1] >> 32) + (low0 < quad[0] ? 1ull : 0ull);
                const u64 mid = quad[2] + high0;         // (no carry: both below 2^54)
                const u64 low1 = mid + (quad[3] << 32);
                const u64 high1 = (quad[3] >> 32) + (low1 < mid ? 1ull : 0ull);
                const bool valid = bl < blocks_here;
                lo[bl] = valid ? low0 : 0ull;
                hi[bl] = valid ? low1 : 0ull;
                kout[bl] = valid ? (int)high1 : 0;
            }
            // what pair P - 1 carries on, as formed
            int from_below[PB];
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) from_below[bl] = __shfl(kout[bl], (lane + 48) & (WAVE - 1));  // lane - 16; for g = 0: lane e + 48, same block
            unsigned long long generate_mask[PB], propagate_mask[PB];
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) {
                const int kin = g > 0 ? from_below[bl] : (bl > 0 ? from_below[bl > 0 ? bl - 1 : 0] : k_below);
                const u64 l = lo[bl] + (u64)(unsigned)kin;
                const u64 c = l < lo[bl] ? 1ull : 0ull;
                const u64 h = hi[bl] + c;
                const bool valid = bl < blocks_here;
                generate_mask[bl] = __ballot(valid && c != 0 && h == 0);
                propagate_mask[bl] = __ballot(valid && l == ~0ull && h == ~0ull);
                lo[bl] = l;
                hi[bl] = h;
            }
            {   // (lanes g = 0 of the next pass: lane e + 48's last block of this one)
                int k_next = from_below[0];
#pragma unroll
                for (int bl = 1; bl < PB; ++bl) k_next = bl < blocks_here ? from_below[bl] : k_next;
                k_below = k_next;
            }
            // The carry-lookahead on the SCALAR unit: a ballot has lane e + 16 g at bit e + 16 g, so the chain g = 0 .. 3 of all sixteen
            // entries at once is three steps "carry << 16" on the 64-bit masks, and a block hands on to the next from its bits 48..63 to the
            // next one's bits 0..15 -- once per wave, not once per lane (the flags gathered per lane and one addition: ~90 vector instructions a pass).
            unsigned long long carry_into[PB];  // bit e + 16 g: the carry into pair 4 bl + g of entry e
            {
                unsigned long long incoming = carry_below_mask;  // bits 0..15: into the pass's first pairs (g = 0 of block 0)
#pragma unroll
                for (int bl = 0; bl < PB; ++bl) {
                    const unsigned long long gm = generate_mask[bl], pm = propagate_mask[bl];
                    unsigned long long c = incoming & 0xffffull;  // into g = 0
                    unsigned long long all = c;
#pragma unroll
                    for (int step = 0; step < 3; ++step) {  // into g = 1, 2, 3: out of the group below, whose flags sit in its sixteen bits
                        c = ((gm | (pm & c)) & (0xffffull << (16 * step))) << 16;
                        all |= c;
                    }
                    carry_into[bl] = all;
                    const unsigned long long out_of_top = (gm | (pm & all)) >> 48;  // out of g = 3: into the next block's g = 0
                    incoming = bl < blocks_here ? out_of_top : incoming;
                }
                carry_below_mask = incoming;
            }
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) {
                const u64 cbit = (carry_into[bl] >> lane) & 1ull;
                const u64 l = lo[bl] + cbit;
                hi[bl] += l < cbit ? 1ull : 0ull;
                lo[bl] = l;
            }
            // shift and store: word 2 P of the entry is lo, 2 P + 1 is hi; the word above hi is the lo of pair P + 1
            u64 above[PB];
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) above[bl] = __shfl(lo[bl], (lane + 16) & (WAVE - 1));  // for g = 3: lane e, same block
            if (g == 3 && bp > 0) emit(8 * bp - 1 - ws, shifted(pending, above[0]));  // (the previous pass's top word: its neighbour is this pass's first)
#pragma unroll
            for (int bl = 0; bl < PB; ++bl) {
                if (bl < blocks_here) {
                    const int j = 8 * (bp + bl) + 2 * g - ws;
                    emit(j, shifted(lo[bl], hi[bl]));
                    const bool top_of_pass = g == 3 && bl == blocks_here - 1;
                    const u64 next = g < 3 ? above[bl] : above[bl + 1 < PB ? bl + 1 : bl];
                    if (top_of_pass) pending = hi[bl];
                    else emit(j + 1, shifted(hi[bl], next));
                }
            }
        } else {
        // this lane holds digits 64 b + 16 g + (4 tq + r) of entry (lane & 15): two words and a carry per block
#pragma unroll
        for (int bl = 0; bl < PB; ++bl) {
            const int b = bp + bl;
            if (b < nb64) {
                i64 quad[4];
#pragma unroll
                for (int tq = 0; tq < 4; ++tq) {
                    const lds_i32* pa = prefix_lds + 64 * b + 16 * g + 4 * tq;
                    v4i digit = acc[bl][tq] + v4i{pa[0], pa[1], pa[2], pa[3]};
                    if (terms == 2) digit += v4i{pa[WB], pa[WB + 1], pa[WB + 2], pa[WB + 3]};
                    quad[tq] = (i64)digit[0] + ((i64)digit[1] << 8) + ((i64)digit[2] << 16) + ((i64)digit[3] << 24);
                }
                // words: V0 = quad0 + 2^32 quad1, V1 = quad2 + 2^32 quad3 (each below 2^86 in magnitude)
                const __int128 v0 = (__int128)quad[0] + ((__int128)quad[1] << 32);
                const __int128 v1 = (__int128)quad[2] + ((__int128)quad[3] << 32) + (v0 >> 64);
                if (store) {
                    at_entry.numerator[(size_t)(8 * b + 2 * g) * at_entry.numerator_stride] = (u64)v0;
                    at_entry.numerator[(size_t)(8 * b + 2 * g + 1) * at_entry.numerator_stride] = (u64)v1;
                    at_entry.carry[(size_t)(4 * b + g) * at_entry.numerator_stride] = (int)(v1 >> 64);
                }
            }
        }
        }
    }
    TILE_STAMP(28);  // words and carries out
    if constexpr (FUSED) {
        // the top word (lanes g = 3 hold it unshifted), the sign's fill above the words formed, the bit length of the new entry
        const int formed = 8 * nb64 - ws;  // words of the numerator, less those the shift drops (formed > 0: the numerator's bound counts the shift)
        const int e16 = lane & 15;
        const u64 fill = __shfl((i64)pending < 0 ? ~0ull : 0ull, e16 + 48);
        // (what the other buffer holds at this place: an entry of bit length `stale_bits`, sign-extended through all L words -- where the
        //  sign stays, the fill above its words and the new value's is already there, as in the second pass of round 5)
        const int stale_words = min(L, (stale_bits + 1 + 63) >> 6);
        const bool same_sign = ((i64)stale_top < 0) == (fill != 0);
        const int fill_end = same_sign ? max(formed, stale_words) : L;
        if (g == 3) emit(formed - 1, shifted(pending, fill));
        if (store)
            for (int j = formed + g; j < fill_end; j += 4) at_entry.numerator[(size_t)j * at_entry.numerator_stride] = fill;
        // the bit length of the magnitude (finish_update_entry's rules), over the four lanes of the entry
        int top = fill ? top_not_ones : top_nonzero;
        u64 at_top = fill ? ~word_not_ones : word_nonzero;
        int lowest = lowest_nonzero;
#pragma unroll
        for (int d = 16; d < WAVE; d *= 2) {
            const int other_top = __shfl_xor(top, d);
            const u64 other_word = __shfl_xor(at_top, d);
            lowest = min(lowest, __shfl_xor(lowest, d));
            if (other_top > top) { top = other_top; at_top = other_word; }
        }
        int bits;
        if (fill == 0) bits = top < 0 ? 0 : 64 * top + (64 - __clzll((long long)at_top));
        else if (top < 0) bits = 1;  // -1
        else {
            bits = 64 * top + (64 - __clzll((long long)at_top));
            if (lowest >= top && (at_top & (at_top + 1)) == 0) bits += 1;  // -(2^k): ~v + 1 carries into a new bit
        }
        if (store && g == 0) *at_entry.words = bits;
    } else {
        if (store && g == 0) *at_entry.words = 8 * nb64;
    }
    return issued;
}
template <int L, bool FUSED = false>
__device__ __forceinline__ int mfma_update_tile(const UpdateTileArgs lp, const UpdateTileEntry at_entry, const lds_u32* toeplitz_lds, const lds_i32* prefix_lds, int row, bool store,
                                                int terms, int nb64, int lane, int shift = 0) {
    return mfma_update_tile_fields<L, FUSED>(lp.stamps, at_entry.entry, at_entry.entry_stride, at_entry.second, at_entry.second_stride, at_entry.numerator, at_entry.carry,
                                             at_entry.words, at_entry.numerator_stride, toeplitz_lds, prefix_lds, row, store, terms, nb64, lane, shift);
}

// The numerator of one entry as the tiles left it (word w at numerator[w * numerator_stride], the carry of pair P at carries[P *
// numerator_stride]): carries run through the pairs, shifted right by `shift` (sign-extended from its 64 * words bits), negated where
// `flip`; the L words of the new entry stored at result[w * result_stride], its bit length returned.
template <int L>
__device__ __forceinline__ int finish_update_entry(const u64* numerator, const int* carries, size_t numerator_stride, u64* result, size_t result_stride, int words, int shift,
                                                   bool flip, int old_bits = -1) {
    const int ws = min(shift >> 6, L), bs = shift & 63;
    // (old_bits >= 0: `result` holds an integer of that bit length -- the entry before this pivot.  Above its words and the new value's,
    //  memory already holds the sign's fill wherever the sign stays: those words are not written again, a fifth of the pass's stores
    //  at 128 limbs.  Not on a flipped pivot.)
    const int old_words = old_bits < 0 ? L : min(L, (old_bits + 1 + 63) >> 6);
    const bool old_negative = old_bits >= 0 && !flip ? (i64)result[(size_t)(L - 1) * result_stride] < 0 : false;
    const bool may_skip = old_bits >= 0 && !flip;
    i64 carry = 0;
    u64 fill = 0;
    bool negation_carry = true;
    int top_nonzero = -1, top_not_ones = -1;
    u64 word_nonzero = 0, word_not_ones = 0;
    bool zeros_so_far = true, zeros_below_not_ones = true;
    auto emit = [&](int j, u64 v, bool pure_fill = false) {  // word j of the shifted numerator: tracked for the bit length, negated where asked, stored
        if (v != 0) { top_nonzero = j; word_nonzero = v; }
        if (v != ~0ull) { top_not_ones = j; word_not_ones = ~v; zeros_below_not_ones = zeros_so_far; }
        zeros_so_far = zeros_so_far && v == 0;
        u64 stored = v;
        if (flip) {
            stored = ~v + (negation_carry ? 1ull : 0ull);
            negation_carry = negation_carry && v == 0;
        }
        if (pure_fill && may_skip && j >= old_words && (fill != 0) == old_negative) return;  // (already there)
        result[(size_t)j * result_stride] = stored;
    };
    // eight words and four carries are requested together, then the carries run; word j of the result is word j + ws of the
    // numerator shifted down by bs bits with the word above it (the sign above the numerator's top)
    u64 current = 0;
#pragma unroll 1
    for (int group = 0; group < L / 8; ++group) {
        u64 r[8];
        const bool beyond = 8 * group >= words;  // (this group's words are the sign's fill)
        if (8 * group < words) {
            u64 w[8];
            i64 out[4];
#pragma unroll
            for (int t = 0; t < 8; ++t) w[t] = numerator[(size_t)(8 * group + t) * numerator_stride];
#pragma unroll
            for (int t = 0; t < 4; ++t) out[t] = carries[(size_t)(4 * group + t) * numerator_stride];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const u64 r0 = w[2 * t] + (u64)carry;
                const i64 k0 = carry >= 0 ? (r0 < w[2 * t] ? 1 : 0) : (r0 > w[2 * t] ? -1 : 0);
                const u64 r1 = w[2 * t + 1] + (u64)k0;
                const i64 k1 = k0 >= 0 ? (r1 < w[2 * t + 1] ? 1 : 0) : (r1 > w[2 * t + 1] ? -1 : 0);
                carry = out[t] + k1;
                r[2 * t] = r0;
                r[2 * t + 1] = r1;
            }
            if (8 * group + 8 >= words) fill = (i64)r[7] < 0 ? ~0ull : 0ull;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t) r[t] = fill;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = 8 * group + t - 1 - ws;
            if (j >= 0) emit(j, bs ? (current >> bs) | (r[t] << (64 - bs)) : current, beyond && t > 0);
            current = r[t];
        }
    }
    for (int j = max(0, L - 1 - ws); j < L; ++j) {
        emit(j, bs ? (current >> bs) | (fill << (64 - bs)) : current, j > max(0, L - 1 - ws) || words < L);
        current = fill;
    }
    // bit length of the magnitude (the same for the value and its negation)
    if (fill == 0) return top_nonzero < 0 ? 0 : 64 * top_nonzero + (64 - __clzll((long long)word_nonzero));
    if (top_not_ones < 0) return 1;  // -1
    int bits = 64 * top_not_ones + (64 - __clzll((long long)word_not_ones));
    if (zeros_below_not_ones && (word_not_ones & (word_not_ones + 1)) == 0) bits += 1;  // -(2^k): ~v + 1 carries into a new bit
    return bits;
}

// The entering column alpha~_q = N a_q, exactly, at the wide types.  One wave per 64 rows streaming the column's operands from word 0 to the
// last is 13 waves on 25FV47 and 32 round trips one after the other, 0.3 ms a pivot with the rest of the grid waiting.  So the words
// are cut into chunks of ENTER_CHUNK: a wave forms the words of one chunk for 64 rows from those words of the operands alone
// (entering_column_chunks: eight times the waves at 128 limbs, an eighth of the turns each), and after a barrier a wave per ROW adds
// the chunks up -- each carries three words into the next --, stores alpha~_i, its bit length and the row's factor (entering_column_rows).  The
// chunks' words lie in price_a, which nobody needs at this point of a pivot ((ENTER_CHUNK + 3) words per chunk and row, word-major).
constexpr int ENTER_CHUNK = 16;
template <int L>
__device__ __noinline__ void entering_column_chunks(const ExactLP& lp, int q) {
    constexpr int CHUNKS = L / ENTER_CHUNK;
    const int m = lp.m, lane = threadIdx.x & (WAVE - 1);
    const int wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    const size_t MM = (size_t)m * m;
    const int row_blocks = (m + WAVE - 1) / WAVE;
    const int e0 = __builtin_amdgcn_readfirstlane(lp.col_start[q]), e1 = __builtin_amdgcn_readfirstlane(lp.col_start[q + 1]);
    for (int item = wave_of_grid; item < row_blocks * CHUNKS; item += waves_of_grid) {
        const int chunk = item / row_blocks, i = (item - chunk * row_blocks) * WAVE + lane;
        const bool active = i < m;
        LeadingWords unused;
        stream_column_products<L, true>(lp, e0, e1, i, active, lane, MM, lp.price_a + (size_t)chunk * (ENTER_CHUNK + 3) * m + (active ? i : 0), (size_t)m, unused,
                                        chunk * ENTER_CHUNK, (chunk + 1) * ENTER_CHUNK);
    }
}
// word k - d of a per-word quantity held a lane per word (slot t of lane l: word l + 64 t), in the lane of word k; zero below word 0
template <int SLOTS>
__device__ __forceinline__ void words_from_below(const u64 (&x)[SLOTS], int d, u64 (&out)[SLOTS], int lane) {
    u64 wrapped[SLOTS];
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) wrapped[t] = __shfl(x[t], (lane - d) & (WAVE - 1));
#pragma unroll
    for (int t = 0; t < SLOTS; ++t) out[t] = lane >= d ? wrapped[t] : (t > 0 ? wrapped[t > 0 ? t - 1 : 0] : 0ull);
}
// -(a * b) modulo 2^(64 L), a and b in LDS (a: this wave's), stored word-major at out[k * stride], by ONE wave with every lane at work:
// WAVE / (L / 4) lanes share the block products of an output block (wave_mul_lo_store gives a block to one lane: half the lanes idle,
// the others in step with the longest sum), their nine-word windows are added across the lanes, the windows are laid over each other
// a lane per block -- what a block carries on goes from lane to lane until none is left --, and the two's complement is taken by the
// lanes together (zero up to the lowest non-zero word).  (The row factors -alpha~_i u of the update: 48 -> 15 us a pivot at 128 limbs.)
template <int L, bool NEGATE = true>
__device__ __forceinline__ void wave_mul_lo_negated(const u64* a, const u64* b, u64* out, size_t stride, int lane) {
    static_assert(L % 4 == 0 && L / 4 <= WAVE && WAVE % (L / 4) == 0, "a lane per block of four words, or several");
    constexpr int NB = L / 4, SHARE = WAVE / NB;  // lanes per output block
    const int K = lane % NB, part = lane / NB;
    u64 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0;
#pragma unroll 1
    for (int I = part; I <= K; I += SHARE) {
        const int J = K - I;
        u64 a4[4], b4[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            a4[t] = a[4 * I + t];
            b4[t] = b[4 * J + t];
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            u64 carry = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const u128 t = (u128)a4[ii] * b4[jj] + acc[ii + jj] + carry;
                acc[ii + jj] = (u64)t;
                carry = (u64)(t >> 64);
            }
#pragma unroll
            for (int k = ii + 4; k < 9; ++k) {
                const u128 t = (u128)acc[k] + carry;
                acc[k] = (u64)t;
                carry = (u64)(t >> 64);
            }
        }
    }
    for (int d = NB; d < WAVE; d *= 2) {  // the windows of the lanes that share a block, added up (the same sum in all of them)
        u64 carry = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const u128 t = (u128)acc[k] + __shfl_xor(acc[k], d) + carry;
            acc[k] = (u64)t;
            carry = (u64)(t >> 64);
        }
    }
    // block K's four words: its window's first four, the next four of block K - 1's, the ninth of block K - 2's (lanes 0 .. NB - 1)
    u64 word[4];
    u64 carried;  // into block K + 1
    {
        u64 below[4], below2;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u64 v = __shfl_up(acc[4 + r], 1);
            below[r] = K >= 1 ? v : 0ull;
        }
        {
            const u64 v = __shfl_up(acc[8], 2);
            below2 = K >= 2 ? v : 0ull;
        }
        u64 carry = 0;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u128 t = (u128)acc[r] + below[r] + (r == 0 ? below2 : 0ull) + carry;
            word[r] = (u64)t;
            carry = (u64)(t >> 64);
        }
        carried = carry;
    }
    for (;;) {
        const bool hands_on = carried != 0 && lane < NB - 1;  // (what leaves the last block is dropped: modulo 2^(64 L))
        if (__ballot(hands_on && lane < NB) == 0) break;
        const u64 v = __shfl_up(carried, 1);
        u64 carry = lane >= 1 && lane < NB ? v : 0ull;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const u128 t = (u128)word[r] + carry;
            word[r] = (u64)t;
            carry = (u64)(t >> 64);
        }
        carried = carry;
    }
    if (!NEGATE) {  // (a * b itself)
        if (lane < NB) {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(4 * lane + r) * stride] = word[r];
        }
        return;
    }
    // the two's complement: zero up to the lowest non-zero word, that word's complement plus one, the complements above
    bool nonzero = false;
    int first = 4;
#pragma unroll
    for (int r = 3; r >= 0; --r)
        if (word[r] != 0) { nonzero = true; first = r; }
    const unsigned long long lanes_nonzero = __ballot(nonzero && lane < NB);
    const int lowest_lane = lanes_nonzero ? __ffsll((long long)lanes_nonzero) - 1 : NB;
    if (lane < NB) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            u64 v = word[r];
            if (lane > lowest_lane || (lane == lowest_lane && r > first)) v = ~v;
            else if (lane == lowest_lane && r == first) v = ~v + 1ull;
            out[(size_t)(4 * lane + r) * stride] = v;
        }
    }
}
// The second half of the entering column (after entering_column_chunks and a barrier), a WAVE PER ROW: the chunks of row i are added up
// a lane per word -- a word, what the chunk below carries into it, less one where that is negative; the small signed carries from lane
// to lane until none is left --, alpha~_i and its bit length are stored, and the row's factor of the update -alpha~_i u follows at
// once from the words the wave holds (through LDS).  (A wave per 64 rows adding the chunks up word after word, a barrier, and a
// wave per row multiplying were three steps: 36 + 6 + 48 us of every pivot at 128 limbs.)
template <int L>
__device__ __noinline__ void entering_column_rows(const ExactLP& lp, int q, const u64* dinv, int limit_bits, int* overflow, bool with_y, int* cq_bits_out) {
    constexpr int SLOTS = (L + WAVE - 1) / WAVE;
    u64 (*s_row)[L] = reinterpret_cast<u64 (*)[L]>(exact_arena<L>());
    const int m = lp.m, lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
    const int wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    const int e0 = __builtin_amdgcn_readfirstlane(lp.col_start[q]), e1 = __builtin_amdgcn_readfirstlane(lp.col_start[q + 1]);
    for (int row = wave_of_grid; row < m; row += waves_of_grid) {
        // (the words the chunks were formed to: the bound of the row's block of 64, as entering_column_chunks found it)
        const int base = row & ~(WAVE - 1);
        int words = 0;
        const int awide_lane = column_products_bound<L>(lp, e0, e1, base + lane, base + lane < m, &words);
        const int awide = __shfl(awide_lane, row - base);
        u64 word[SLOTS];
        i64 carried[SLOTS];
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            const int k = lane + t * WAVE, chunk = k / ENTER_CHUNK, at = k % ENTER_CHUNK;
            const bool valid = k < words;
            const u64* part = lp.price_a + (size_t)chunk * (ENTER_CHUNK + 3) * m + row;
            const u64* below = part - (size_t)(ENTER_CHUNK + 3) * m + (size_t)ENTER_CHUNK * m;  // the three words the chunk below carries on
            const u64 own = valid ? part[(size_t)at * m] : 0ull;
            const u64 from_below = valid && chunk > 0 && at < 3 ? below[(size_t)at * m] : 0ull;
            const u64 sign_below = valid && chunk > 0 && at == 3 ? below[(size_t)2 * m] >> 63 : 0ull;
            const __int128 sum = (__int128)((u128)own + from_below) - (__int128)sign_below;
            word[t] = (u64)sum;
            carried[t] = (i64)(sum >> 64);
        }
        for (;;) {
            u64 out_going[SLOTS], incoming[SLOTS];
            bool any = false;
#pragma unroll
            for (int t = 0; t < SLOTS; ++t) {
                out_going[t] = (u64)carried[t];
                any = any || (carried[t] != 0 && lane + t * WAVE + 1 < words);
            }
            if (__ballot(any) == 0) break;
            words_from_below<SLOTS>(out_going, 1, incoming, lane);
#pragma unroll
            for (int t = 0; t < SLOTS; ++t) {
                const __int128 sum = (__int128)(u128)word[t] + (__int128)(i64)incoming[t];
                word[t] = lane + t * WAVE < words ? (u64)sum : 0ull;
                carried[t] = lane + t * WAVE < words ? (i64)(sum >> 64) : 0;
            }
        }
        // the words above the bound are the sign's
        u64 top_word = 0;
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            const u64 v = __shfl(word[t], (words - 1) & (WAVE - 1));
            if (t == (words - 1) / WAVE) top_word = v;
        }
        const u64 fill = (i64)top_word < 0 ? ~0ull : 0ull;
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            const int k = lane + t * WAVE;
            if (k >= words) word[t] = fill;
            if (k < L) {
                lp.alpha[(size_t)row * L + k] = word[t];
                s_row[wave][k] = word[t];
            }
        }
        const int bits = wave_bit_length_words<L>(word, lane);
        if (lane == 0) {
            lp.x_bits[row] = bits;  // (the fit test of the update wants it once per ENTRY of N)
            if (awide >= limit_bits) *overflow = 1;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wave_mul_lo_negated<L>(s_row[wave], dinv, lp.x_part + row, (size_t)m, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // ... and y's factor c~_q u with the bit length of c~_q, by the last wave of the grid (y rides along with the update of N, see
    // update_on_matrix_cores; on 25FV47 that wave has no row)
    if (with_y && wave_of_grid == waves_of_grid - 1) {
        u64 word[SLOTS];
#pragma unroll
        for (int t = 0; t < SLOTS; ++t) {
            const int k = lane + t * WAVE;
            word[t] = k < L ? lp.ctil[(size_t)q * L + k] : 0ull;
            if (k < L) s_row[wave][k] = word[t];
        }
        const int bits = wave_bit_length_words<L>(word, lane);
        if (lane == 0) *cq_bits_out = bits;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        wave_mul_lo_negated<L, false>(s_row[wave], dinv, lp.y_part, 1, lane);
    }
}

// The three passes of the pricing step, each a function of its own -- not inlined, so that the registers of its loop are allocated apart
// from the state of the pivot loop (as for the update below: with everything in one body, a new variable ANYWHERE in the loop moved the
// spills of a kernel that is held to 256 registers into these loops -- the pass over N went from 0.8 to 2.1 s of 25FV47's solve and the
// products from 1.2 to 2.2 when the loop learnt to carry y along).  Every workgroup of the grid calls them; `overflow` is the
// workgroup's flag (LDS) for a value whose bound reaches `limit_bits`.
// y_k = sum_i c_B(i) N(i, k) and the bit length of |y_k|: a wave per column of N
template <int L>
__device__ __noinline__ void price_form_y(const ExactLP& lp, int limit_bits, int* overflow) {
    const int m = lp.m, lane = threadIdx.x & (WAVE - 1);
    const int wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    const size_t MM = (size_t)m * m;
    const int log_m = 32 - __clz(m > 1 ? m - 1 : 1) + 1;
    for (int k = wave_of_grid; k < m; k += waves_of_grid) {
        int widest = 0;
        for (int i = lane; i < m; i += WAVE) {
            const i64 cb = lp.cb_row[i];
            if (cb != 0) widest = max(widest, lp.N_bits[(size_t)k * m + i] + small_bits(cb));
        }
        for (int d = 1; d < WAVE; d *= 2) widest = max(widest, __shfl_xor(widest, d));
        widest += log_m;
        if (lane == 0 && widest >= limit_bits) *overflow = 1;
        u64* yk = lp.y + (size_t)k * L;
        // (the bit length of |y_k| from the words as they pass, as finish_update_entry finds an entry's: the update's fit test and the
        //  bounds of the next pass read it, and a pivot on the matrix cores leaves the exact length there as well)
        int top_nonzero = -1, top_not_ones = -1;
        u64 word_nonzero = 0, word_not_ones = 0, last = 0;
        bool zeros_below_not_ones = true, zeros_so_far = true;
        wave_cost_dot<L>(lp.N + (size_t)k * m, MM, lp.cb_row, m, lane, (widest + 2 + 63) / 64, [&](int word, u64 value) {
            if (lane == 0) yk[word] = value;
            if (value != 0) { top_nonzero = word; word_nonzero = value; }
            if (value != ~0ull) { top_not_ones = word; word_not_ones = ~value; zeros_below_not_ones = zeros_so_far; }
            zeros_so_far = zeros_so_far && value == 0;
            last = value;
        });
        if (lane == 0) {
            int bits;
            if ((i64)last >= 0) bits = top_nonzero < 0 ? 0 : 64 * top_nonzero + (64 - __clzll((long long)word_nonzero));
            else if (top_not_ones < 0) bits = 1;
            else {
                bits = 64 * top_not_ones + (64 - __clzll((long long)word_not_ones));
                if (zeros_below_not_ones && (word_not_ones & (word_not_ones + 1)) == 0) bits += 1;
            }
            lp.y_bits[k] = bits;
        }
    }
}
// c~_j for every non-basic, non-artificial column (a wave per column); the columns with c~_j < 0 into neg_list
template <int L>
__device__ __noinline__ void price_reduced_costs(const ExactLP& lp, int phase, double mD, int eD, int D_bits, int limit_bits, int* overflow) {
    const int n = lp.n, lane = threadIdx.x & (WAVE - 1);
    const int wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    const int n_priced = n - lp.n_art;
    for (int item = wave_of_grid; item < n_priced; item += waves_of_grid) {
        const int j = lp.n_art + item;
        double cd = 0.0;
        if (lp.pos[j] < 0) {  // (the whole wave)
            int widest = 0;
            cd = reduced_cost_wave<L>(lp, lp.D, j, phase, lane, mD, eD, D_bits, &widest);
            if (lane == 0 && widest + 1 >= limit_bits) *overflow = 1;
        }
        if (lane == 0) {
            lp.key[j] = 0.0;
            lp.cd[j] = cd;
            if (cd != 0.0) lp.neg_list[atomicAdd(&lp.neg_list[n], 1)] = j;
        }
    }
}
// The EXACT products (N a_j)_i for the columns of `list` into price_a: a wave takes a column and 64 neighbouring rows and forms the sum
// word by word, least significant first (stream_column_products); its share of the weight estimate from the two leading words, with no
// error.  Rounds 2 to 5 ran this for every column that could enter -- each column of N read 3.6 times per pass on 25FV47, 2 GB a pivot,
// 1.2 of 4.4 s; now it runs for the columns whose estimate could not be bounded (none, so far) and for tied candidates, whose exact
// weights are sums of these squares.
template <int L>
__device__ __noinline__ void price_products(const ExactLP& lp, const int* list, int count, double mD, int eD, int limit_bits, int* overflow) {
    const int m = lp.m, lane = threadIdx.x & (WAVE - 1);
    const int wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    const size_t MM = (size_t)m * m;
    const size_t PP = (size_t)EX_PRODUCT_SLOTS * m;  // (price_a: column c of the list in slot c modulo EX_PRODUCT_SLOTS -- a caller that reads the products
    const int row_blocks = (m + WAVE - 1) / WAVE;    //  passes no more columns than that at a time)
    for (long long item = wave_of_grid; item < (long long)count * row_blocks; item += waves_of_grid) {
        const int c = (int)(item / row_blocks), i = (int)(item - (long long)c * row_blocks) * WAVE + lane;
        const int j = list[c], jj = j - lp.n_art;
        const int e0 = __builtin_amdgcn_readfirstlane(lp.col_start[j]), e1 = __builtin_amdgcn_readfirstlane(lp.col_start[j + 1]);
        const bool active = i < m;
        const size_t pair = (size_t)jj * m + (active ? i : 0);
        LeadingWords lead;
        const int awide = stream_column_products<L>(lp, e0, e1, i, active, lane, MM, lp.price_a + (size_t)(c % EX_PRODUCT_SLOTS) * m + (active ? i : 0), PP, lead);
        if (active) {
            if (awide >= limit_bits) *overflow = 1;
            int ea = 0;
            const double ma = lead.mantissa(&ea);
            const double ad = ldexp(ma / mD, ea - eD);
            lp.price_term[pair] = ad * ad * weight_as_double(lp, lp.basis[i]);
            lp.price_err[pair] = 0.0;
        }
    }
}
// The weight ESTIMATES of the columns of neg_list from the leading words of the operands only.  key_j = (c~_j / D)^2 / (w_j + sum_i
// w_i ((N a_j)_i / D)^2) is compared as a double -- every column within 1e-9 of the best goes on to an exact comparison -- so all it
// needs of (N a_j)_i = sum_e v_e N(i, r_e) is a dozen digits.  A wave takes a column and 64 neighbouring rows; with B the largest bit
// length among its operands' products (N_bits: no word of N is read for that) it reads the KW = 4 words of every operand that end at B's
// word -- one round trip, 2 KB per entry of the column where the exact sum reads 64 KB at 128 limbs -- and adds the multiples of those
// windows exactly (two's complement, KW + 2 words).  What is cut off below the window is less than one unit per operand: the sum is
// (N a_j)_i / 2^(64 k0) up to sum_e |v_e|, at least 180 - bits(v) bits below the largest product.  That bound is not taken on trust: every
// term comes with the error it may carry ((2 |a| eps + eps^2) w_i, stored beside it), the pass over the terms adds both up, and a
// column whose error could reach 1e-11 of its weight is formed exactly (price_products) before anything is decided.  When the window
// reaches down to word 0 the estimate IS the exact sum's (same leading words, same double).
template <int L>
__device__ __noinline__ void price_estimates(const ExactLP& lp, double mD, int eD) {
    constexpr int KW = L < 4 ? L : 4;  // words of every operand that are read
    constexpr int SW = KW + 2;         // words of the sum
    const int m = lp.m, n = lp.n, lane = threadIdx.x & (WAVE - 1);
    const int wave_of_grid = (blockIdx.x * blockDim.x + threadIdx.x) / WAVE, waves_of_grid = gridDim.x * blockDim.x / WAVE;
    const size_t MM = (size_t)m * m;
    const int row_blocks = (m + WAVE - 1) / WAVE;
    const int n_negative = lp.neg_list[n];
    for (long long item = wave_of_grid; item < (long long)n_negative * row_blocks; item += waves_of_grid) {
        const int c = (int)(item / row_blocks), i = (int)(item - (long long)c * row_blocks) * WAVE + lane;
        const int j = lp.neg_list[c], jj = j - lp.n_art;
        const int e0 = __builtin_amdgcn_readfirstlane(lp.col_start[j]), e1 = __builtin_amdgcn_readfirstlane(lp.col_start[j + 1]);
        const bool active = i < m;
        const int row = active ? i : 0;
        const size_t pair = (size_t)jj * m + row;
        // the column's entries, one per lane, handed round with readlane (longer columns read them from memory at every use)
        const int len = e1 - e0;
        const bool in_lanes = len <= WAVE;
        int my_offset = 0;  // row_index * m of entry e0 + lane
        i64 my_value = 0;
        if (in_lanes && lane < len) {
            my_offset = lp.row_index[e0 + lane] * m;
            my_value = lp.value[e0 + lane];
        }
        auto entry = [&](int e, int* offset, i64* value) {  // e uniform over the wave
            if (in_lanes) {
                *offset = __builtin_amdgcn_readlane(my_offset, e);
                const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(u64)my_value, e);
                const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)((u64)my_value >> 32), e);
                *value = (i64)(((u64)hi << 32) | lo);
            } else {
                *offset = lp.row_index[e0 + e] * m;
                *value = lp.value[e0 + e];
            }
        };
        int B = 0;  // the largest bit length of a product v_e N(i, r_e) over the wave's rows
        double v_sum = 0.0;  // sum_e |v_e|
        for (int e = 0; e < len; e += 4) {  // (four bit lengths in flight)
            int bits[4];
            i64 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int offset;
                entry(e + u < len ? e + u : len - 1, &offset, &v[u]);
                if (e + u >= len) v[u] = 0;
                bits[u] = active ? lp.N_bits[(size_t)offset + row] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (bits[u] != 0 && v[u] != 0) B = max(B, bits[u] + small_bits(v[u]));
                v_sum += fabs((double)v[u]);
            }
        }
        for (int d = 1; d < WAVE; d *= 2) B = max(B, __shfl_xor(B, d));
        B = __builtin_amdgcn_readfirstlane(B);
        // the window: words k0 .. k0 + KW - 1, the last one holding the sign of every operand (B + 8 bits fit below its top, or it is the
        // last word of the integers)
        const int k_top = min(L - 1, max(KW - 1, (B + 8) / 64)), k0 = k_top - (KW - 1);
        u64 S[SW];
#pragma unroll
        for (int t = 0; t < SW; ++t) S[t] = 0;
        const u64* window = lp.N + (size_t)k0 * MM + row;
        for (int e = 0; e < len; e += 4) {  // four operands in flight
            u64 w[4][KW];
            i64 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int offset;
                entry(e + u < len ? e + u : len - 1, &offset, &v[u]);
                if (e + u >= len) v[u] = 0;
#pragma unroll
                for (int t = 0; t < KW; ++t) w[u][t] = window[(size_t)t * MM + offset];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const u64 mag = v[u] < 0 ? (u64)(-(v[u] + 1)) + 1 : (u64)v[u];
                const bool x_negative = (i64)w[u][KW - 1] < 0;
                // |v| X as SW words, two's complement: |v| U - [X < 0] |v| 2^(64 KW) for the unsigned reading U of the window
                u64 prod[SW];
                u64 carry = 0;
#pragma unroll
                for (int t = 0; t < KW; ++t) {
                    const u128 pr = (u128)w[u][t] * mag + carry;
                    prod[t] = (u64)pr;
                    carry = (u64)(pr >> 64);
                }
                prod[KW] = carry - (x_negative ? mag : 0ull);
                prod[KW + 1] = x_negative && mag != 0 ? ~0ull : 0ull;
                if (v[u] >= 0) {
                    u64 cy = 0;
#pragma unroll
                    for (int t = 0; t < SW; ++t) {
                        const u128 sum = (u128)S[t] + prod[t] + cy;
                        S[t] = (u64)sum;
                        cy = (u64)(sum >> 64);
                    }
                } else {
                    u64 borrow = 0;
#pragma unroll
                    for (int t = 0; t < SW; ++t) {
                        const u128 diff = (u128)S[t] - prod[t] - borrow;
                        S[t] = (u64)diff;
                        borrow = (u64)(diff >> 64) & 1ull;
                    }
                }
            }
        }
        if (active) {
            LeadingWords lead;
#pragma unroll
            for (int t = 0; t < SW; ++t) lead.feed(t, S[t]);
            int ea = 0;
            const double ma = lead.mantissa(&ea);
            const double ad = ldexp(ma / mD, ea + 64 * k0 - eD);
            const double eps = k0 > 0 ? ldexp(v_sum / mD, 64 * k0 - eD) : 0.0;  // what the cut-off words could add to |a_i / D|
            const double w = weight_as_double(lp, lp.basis[i]);
            lp.price_term[pair] = ad * ad * w;
            lp.price_err[pair] = (2.0 * fabs(ad) * eps + eps * eps) * w;
        }
    }
}

// The whole update of a pivot on the matrix cores (every workgroup of the cooperative grid calls it): a function of its own so that its
// registers -- and those its tiles force the caller to save -- are allocated apart from the other steps of the loop (inlined, the
// pricing passes lost a quarter of their speed to spills).
struct UpdateScalars {
    int p, shift, flip, ap_bits, D_bits, xp_bits, n_heavy, n_rows_alpha;
    int with_y, cq_bits;  // y = c_B' N rides along as one more row of N (its factor c~_q u in lp.y_part), |c~_q| has that many bits
    int fused;            // the tiles finish the entries of N themselves, into lp.N_alt (x~_B and y keep the two passes)
};
template <int L>
__device__ __noinline__ void update_on_matrix_cores(const ExactLP& lp, const UpdateScalars sc, const u64* s_c1, const u64* s_ap, unsigned long long& products_needed,
                                                    unsigned long long& products_issued, unsigned& barrier_epoch, const BarrierPlace barrier_place) {
    struct {
        unsigned* words;
        unsigned& epoch;
        BarrierPlace place;
        __device__ void sync() { grid_barrier(words, epoch, place); }
    } grid{lp.barrier, barrier_epoch, barrier_place};
    const int tid = threadIdx.x, T = blockDim.x;
    const int G = gridDim.x, block = blockIdx.x;
    const int gtid = block * T + tid, GT = G * T;
    const bool leader = block == 0 && tid == 0;
    const int m = lp.m;
    const size_t MM = (size_t)m * m;
    auto N_at = [&](int i, int c) { return lp.N + (size_t)c * m + i; };
    const int p = sc.p, shift = sc.shift, ap_bits = sc.ap_bits, D_bits = sc.D_bits, xp_bits = sc.xp_bits, n_heavy = sc.n_heavy, n_rows_alpha = sc.n_rows_alpha;
    const bool flip = sc.flip != 0;
    const bool fused = sc.fused != 0;
        // ---- the update on the matrix cores (see mfma_update_tile): tiles of 16 entries of a column, a wave each ----
        static_assert(sizeof(UpdateLds<L>) <= exact_arena_bytes<L>(), "arena");
        UpdateLds<L>& s_update = *reinterpret_cast<UpdateLds<L>*>(exact_arena<L>());
        const int lane = tid & (WAVE - 1), wave = tid / WAVE, waves = T / WAVE;
        const int e16 = lane & 15;
        int issued = 0;
        const UpdateTileArgs tile_args{lp.x_part, m, lp.prof};
        const lds_u32* toeplitz_lds = (const lds_u32*)&s_update.toeplitz[0][0][0];
        const lds_i32* prefix_lds = (const lds_i32*)&s_update.prefix[0][0];
        // The LAST workgroup takes no tiles: it forms 1 / D'_odd (D' = |alpha~_p|) for the next pivot meanwhile -- five doublings of two
        // truncated products each, 0.1 ms at 128 limbs that every workgroup spent at the top of every pivot with nothing beside it.
        const bool inverse_duty = G > 1 && block == G - 1;
        const int tile_blocks = G > 1 ? G - 1 : G;  // workgroups that take tiles
        __syncthreads();
        if (inverse_duty) {
            u64* buffer = (u64*)&s_update.prefix[0][0];  // (8 L words: this workgroup builds no operand images)
            u64 *d_odd = buffer, *x = buffer + L, *t = buffer + 2 * L, *x2 = buffer + 3 * L, *scratch = buffer + 4 * L;
            if (tid < L) x2[tid] = s_ap[tid];
            __syncthreads();
            if (flip && tid == 0) {  // (a zero-level pivot on a negative element: D' = -alpha~_p)
                bool carry = true;
                for (int k = 0; k < L; ++k) {
                    const u64 w = ~x2[k] + (carry ? 1ull : 0ull);
                    carry = carry && x2[k] == 0;
                    x2[k] = w;
                }
            }
            __syncthreads();
            block_odd_part<L>(x2, d_odd, scratch);
            wave_inverse_odd<L>(d_odd, x, t, x2);
            if (tid < L) lp.next_dinv[tid] = x[tid];
        } else {
            build_toeplitz<L>(s_update, 0, s_c1);  // alpha~_p u: the same operand for every entry of N
        }
        __syncthreads();
        unsigned long long t_sub = wall_clock64();
#ifdef RELP_STAMPS  // (diagnostic build, `make stamps`: how evenly the tiles are spread -- every workgroup's time in (a) + (b): sum in prof[33], the pivot's
                    //  longest in prof[35] -> [34]; (a) alone in prof[36], [38] -> [37]; [39] pivots whose longest workgroup was one of the last eight)
        const unsigned long long t_tiles0 = t_sub;
#endif
        auto substamp = [&](int slot) {  // (diagnostic: the leader's time inside the update: [20] both-term tiles, [21] rescaled tiles, [22] barrier, [23] second pass)
            if (leader) {
                const unsigned long long t = wall_clock64();
                lp.prof[slot] += t - t_sub;
                t_sub = t;
            }
        };
        // the need of an entry, in 64-byte blocks of the numerator (the fit test's bound, as the vector path's `blocks`)
        auto blocks64 = [&](int needed_bits) { return min(L / 8, max(1, (needed_bits + 511) / 512)); };
        auto wave_max = [&](int v) {
            for (int d = 1; d < WAVE; d *= 2) v = max(v, __shfl_xor(v, d));
            return v;
        };
        // (a) N(p, k) != 0 and alpha~_i != 0: both terms.  A workgroup takes a run of tiles, column after column, and keeps the
        //     column's N(p, k) as the second Toeplitz operand in LDS while its four waves work through the column's tiles.
        {
            // (x~_B is one more column of N -- x~'_i = (alpha~_p x~_i - alpha~_i x~_p) / D, every row but p -- and is taken here as the
            //  column after the last: a thread per row with two whole products of its own was 2 ms at the end of every pivot at 128 limbs)
            // (... and y = c_B' N is one more ROW of it: y'_k = (alpha~_p y_k + c~_q N(p, k)) / D for every column k -- the roles turned
            //  round, c~_q u is the Toeplitz operand and N(p, k) the entry's factor.  Carried along like this the pricing pass no
            //  longer reads the whole of N once per pivot to form it: 0.8 of 25FV47's 5.4 s.)
            const int tiles_per_column = (n_rows_alpha + 15) / 16, tiles_of_x = (m + 15) / 16, tiles_of_y = sc.with_y ? (m + 15) / 16 : 0;
            const long long total_N = (long long)n_heavy * tiles_per_column, total = total_N + tiles_of_x + tiles_of_y;
            // (a tile of x~_B or y counts for three when the runs are measured out: their entries lie limb after limb -- sixteen cache lines
            //  per load of a tile where sixteen entries of a column of N share one -- and with tiles counted alike the last workgroups, whose
            //  runs they are, were the pivot's longest eight times as often as their number says)
            constexpr int VECTOR_TILE_WEIGHT = 3;
            const long long weighted = total_N + (long long)VECTOR_TILE_WEIGHT * (tiles_of_x + tiles_of_y);
            const long long per_block = (weighted + tile_blocks - 1) / tile_blocks;
            auto tile_at = [&](long long v) { return v <= total_N ? v : min(total, total_N + (v - total_N + VECTOR_TILE_WEIGHT - 1) / VECTOR_TILE_WEIGHT); };
            long long u = inverse_duty ? total : tile_at(min(weighted, (long long)block * per_block));
            const long long u_end = inverse_duty ? total : tile_at(min(weighted, (long long)(block + 1) * per_block));
            const size_t M2 = 2 * (size_t)m;  // stride of the numerators of x~_B and y
            while (u < u_end) {
                const int kind = u < total_N ? 0 : u < total_N + tiles_of_x ? 1 : 2;  // a column of N, x~_B, y
                const int kk = kind == 0 ? (int)(u / tiles_per_column) : n_heavy;
                const int k = kind == 0 ? lp.col_heavy[kk] : -1;
                const long long column_first = kind == 0 ? (long long)kk * tiles_per_column : kind == 1 ? total_N : total_N + tiles_of_x;
                const long long column_end = min(u_end, column_first + (kind == 0 ? tiles_per_column : kind == 1 ? tiles_of_x : tiles_of_y));
                __syncthreads();  // (the previous column's tiles are done with the image)
                for (int w = tid; w < L; w += T) s_update.words[w] = kind == 0 ? N_at(p, k)[(size_t)w * MM] : kind == 1 ? lp.xt[(size_t)p * L + w] : lp.y_part[w];
                __syncthreads();
                build_toeplitz<L>(s_update, 1, s_update.words);
                __syncthreads();
                const int operand_bits = kind == 0 ? lp.N_bits[(size_t)k * m + p] : kind == 1 ? xp_bits : sc.cq_bits;
                for (long long t = u + wave; t < column_end; t += waves) {
                    const int first = 16 * (int)(t - column_first);
                    int row;  // (for y: the column of N)
                    if (kind == 0) row = first + e16 < n_rows_alpha ? lp.row_list[first + e16] : -1;
                    else row = first + e16 < m ? first + e16 : -1;
                    const bool store = row >= 0 && (kind == 2 || row != p);
                    int nb = 1;
                    UpdateTileEntry at_entry{lp.N, MM, lp.x_part, (size_t)m, lp.T, lp.T_carry, lp.T_words, MM};
                    if (store) {
                        const size_t idx = kind == 0 ? (size_t)k * m + row : (size_t)row;
                        const int entry_bits = kind == 0 ? lp.N_bits[idx] : kind == 1 ? lp.xt_bits[row] : lp.y_bits[row];
                        const int factor_bits = kind == 2 ? lp.N_bits[(size_t)row * m + p] : lp.x_bits[row];
                        const int needed = max(ap_bits + entry_bits, factor_bits + operand_bits) + 1 - (D_bits - 1) + shift + 2;
                        nb = blocks64(needed);
                        if (lane < 16) {
                            const int blocks = min(L / 4, max(1, (needed + 255) / 256));
                            products_needed += 16ull * blocks * (blocks + 1);
                        }
                        if (kind == 0 && fused) at_entry = UpdateTileEntry{lp.N + idx, MM, lp.x_part + row, (size_t)m, lp.N_alt + idx, nullptr, lp.N_bits_alt + idx, MM};
                        else if (kind == 0) at_entry = UpdateTileEntry{lp.N + idx, MM, lp.x_part + row, (size_t)m, lp.T + idx, lp.T_carry + idx, lp.T_words + idx, MM};
                        else if (kind == 1 && fused) at_entry = UpdateTileEntry{lp.xt + idx * L, 1, lp.x_part + row, (size_t)m, lp.xt_alt + idx * L, nullptr, lp.xt_bits_alt + idx, 1};
                        else if (kind == 1) at_entry = UpdateTileEntry{lp.xt + idx * L, 1, lp.x_part + row, (size_t)m, lp.Tx + idx, lp.Tx_carry + idx, lp.Tx_words + idx, M2};
                        else if (fused) at_entry = UpdateTileEntry{lp.y + idx * L, 1, N_at(p, row), MM, lp.y_alt + idx * L, nullptr, lp.y_bits_alt + idx, 1};
                        else at_entry = UpdateTileEntry{lp.y + idx * L, 1, N_at(p, row), MM, lp.Tx + m + idx, lp.Tx_carry + m + idx, lp.Tx_words + m + idx, M2};
                    }
                    nb = wave_max(nb);
                    if (fused) issued += mfma_update_tile<L, true>(tile_args, at_entry, toeplitz_lds, prefix_lds, row, store, 2, nb, lane, shift);
                    else issued += mfma_update_tile<L>(tile_args, at_entry, toeplitz_lds, prefix_lds, row, store, 2, nb, lane);
                }
                u = column_end;
            }
        }
        substamp(20);
#ifdef RELP_STAMPS
        __syncthreads();  // (the workgroup's slowest wave counts; the diagnostic build pays for it)
        if (tid == 0 && !inverse_duty) {
            const unsigned long long dt = wall_clock64() - t_tiles0;
            atomicAdd(&lp.prof[36], dt);
            atomicMax(&lp.prof[38], dt);
        }
#endif
        // (b) the entries that are only rescaled: the rows with alpha~_i = 0 of those columns, then every row of the other columns
        {
            const int rest_rows = m - n_rows_alpha;
            const int tiles_a = (rest_rows + 15) / 16, tiles_b = (m + 15) / 16;
            const long long total_a = (long long)n_heavy * tiles_a, total_b = (long long)(m - n_heavy) * tiles_b;
            for (long long t = inverse_duty ? total_a + total_b : (long long)block * waves + wave; t < total_a + total_b; t += (long long)tile_blocks * waves) {
                int k, row;
                if (t < total_a) {
                    const int kk = (int)(t / tiles_a), first = 16 * (int)(t - (long long)kk * tiles_a);
                    k = lp.col_heavy[kk];
                    row = first + e16 < rest_rows ? lp.row_list[m + first + e16] : -1;
                } else {
                    const long long rest = t - total_a;
                    const int kk = (int)(rest / tiles_b), first = 16 * (int)(rest - (long long)kk * tiles_b);
                    k = lp.col_light[kk];
                    row = first + e16 < m ? first + e16 : -1;
                }
                const int entry_bits = row >= 0 ? lp.N_bits[(size_t)k * m + row] : 0;
                const bool store = row >= 0 && row != p && entry_bits != 0;  // (a zero stays a zero)
                int nb = 0;
                if (store) {
                    const int needed = ap_bits + entry_bits + 1 - (D_bits - 1) + shift + 2;
                    nb = blocks64(needed);
                    if (lane < 16) {
                        const int blocks = min(L / 4, max(1, (needed + 255) / 256));
                        products_needed += 8ull * blocks * (blocks + 1);
                    }
                }
                nb = wave_max(nb);
                const size_t idx = (size_t)k * m + (row >= 0 ? row : 0);
                if (fused && lane < 16 && row >= 0 && row != p && entry_bits == 0 && lp.N_bits_alt[idx] != 0) {
                    // (a zero stays a zero -- but the other buffer holds the entry of two pivots ago at this place)
                    for (int w = 0; w < L; ++w) lp.N_alt[(size_t)w * MM + idx] = 0ull;
                    lp.N_bits_alt[idx] = 0;
                }
                if (nb == 0) continue;  // sixteen zeros
                if (fused) {
                    const UpdateTileEntry at_entry{lp.N + idx, MM, lp.x_part, (size_t)m, lp.N_alt + idx, nullptr, lp.N_bits_alt + idx, MM};
                    issued += mfma_update_tile<L, true>(tile_args, at_entry, toeplitz_lds, prefix_lds, row, store, 1, nb, lane, shift);
                } else {
                    const UpdateTileEntry at_entry{lp.N + idx, MM, lp.x_part, (size_t)m, lp.T + idx, lp.T_carry + idx, lp.T_words + idx, MM};
                    issued += mfma_update_tile<L>(tile_args, at_entry, toeplitz_lds, prefix_lds, row, store, 1, nb, lane);
                }
            }
        }
        if (lane == 0) products_issued += 256ull * issued;  // an MFMA is 16 x 16 x 64 byte products = 256 word products
#ifdef RELP_STAMPS
        __syncthreads();
        if (tid == 0 && !inverse_duty) {
            const unsigned long long dt = wall_clock64() - t_tiles0;
            atomicAdd(&lp.prof[33], dt);
            atomicMax(&lp.prof[35], dt * 1024ull + (unsigned long long)block);  // (... and which workgroup it was)
        }
#endif
        if (fused) {  // row p stays as it is (D' = alpha~_p): into the other buffers with it
            for (long long t = gtid; t < (long long)m * L; t += GT) {
                const int w = (int)(t / m), k = (int)(t - (long long)w * m);
                const size_t idx = (size_t)k * m + p;
                lp.N_alt[(size_t)w * MM + idx] = lp.N[(size_t)w * MM + idx];
                if (w == 0) lp.N_bits_alt[idx] = lp.N_bits[idx];
            }
            if (gtid < L) lp.xt_alt[(size_t)p * L + gtid] = lp.xt[(size_t)p * L + gtid];
            if (gtid == 0) lp.xt_bits_alt[p] = lp.xt_bits[p];
            substamp(21);
            return;  // (nothing is left for a second pass, and nobody reads the new values before the barrier that ends the pivot)
        }
        substamp(21);
        grid.sync();  // every numerator is in lp.T
        substamp(22);
        for (size_t idx = gtid; idx < MM; idx += GT) {
            const int words = lp.T_words[idx];
            if (words == 0) continue;
            lp.T_words[idx] = 0;
            lp.N_bits[idx] = finish_update_entry<L>(lp.T + idx, lp.T_carry + idx, MM, lp.N + idx, MM, words, shift, flip, lp.N_bits[idx]);
        }
        for (int i = gtid; i < 2 * m; i += GT) {  // x~_B, then y
            const int words = lp.Tx_words[i];
            if (words == 0) continue;
            lp.Tx_words[i] = 0;
            const int bits = finish_update_entry<L>(lp.Tx + i, lp.Tx_carry + i, 2 * (size_t)m, (i < m ? lp.xt + (size_t)i * L : lp.y + (size_t)(i - m) * L), 1, words, shift, flip);
            if (i < m) lp.xt_bits[i] = bits;
            else lp.y_bits[i - m] = bits;
        }
        substamp(23);
}

// The loop on the WHOLE grid (round 3; round 2 ran it in one workgroup: E226 took 100 s for 342 pivots on 2048-bit integers while
// 255 CUs idled).  The two heavy steps of a pivot -- the pricing pass (every non-basic column against every row of N) and the
// integer-preserving update of the m x m matrix N -- are independent per column / per entry and are spread over all workgroups of
// a COOPERATIVE launch; between steps the workgroups meet at a grid barrier (cooperative groups: ~2 us at 8 workgroups, 7.5 at 64,
// 25 at 256, tools/micro/grid_sync_bench.hip -- the host picks the grid by the work of a pivot).  Every workgroup runs the same
// control flow on the same decisions: arg-max / arg-min reductions go through per-workgroup partials in global memory that every
// workgroup reduces again in the same order; what one thread decides (exact tie breaks, bookkeeping) is decided by thread 0 of
// workgroup 0 and read by everybody after the barrier.  The decisions, and therefore the pivot sequence, are those of the
// one-workgroup kernel bit for bit (tests/test_gpu_exact.py: whole golden traces).
// (amdgpu_waves_per_eu(2, 2): at most 256 registers per lane.  Left to itself the 64-limb instance took 256 VGPRs and 114 AGPRs,
//  one wave per SIMD, and faulted on its first pricing pass on gfx950 / ROCm 7.2 while the 32- and 128-limb instances -- no AGPRs --
//  ran; held to two waves it spills 141 registers to scratch memory instead and runs.)
template <int L>
__global__ void __launch_bounds__(EX_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) exact_simplex_kernel(ExactLP lp) {
    unsigned barrier_epoch = 0;  // grid barriers made so far (grid_barrier: the same count in every workgroup)
    const BarrierPlace barrier_place = grid_barrier_place(lp.barrier);
    struct {
        unsigned* words;
        unsigned& epoch;
        BarrierPlace place;
        __device__ void sync() { grid_barrier(words, epoch, place); }
    } grid{lp.barrier, barrier_epoch, barrier_place};
    __shared__ double s_key[EX_THREADS / WAVE];
    __shared__ unsigned long long s_rank[EX_THREADS / WAVE];
    __shared__ int s_overflow;
    __shared__ u64 s_dinv[L];
    __shared__ u64 s_c1[L];
    __shared__ u64 s_words[3][L];  // operands and scratch of the workgroup's word-array arithmetic
    __shared__ u64 s_part[3 * L];  // ... the columns of its products (block_mul_lo)
    static_assert(L <= EX_THREADS, "a thread per output word");
    __shared__ int s_shift, s_D_bits, s_eD, s_ap_bits, s_flip;
    __shared__ double s_mD;
    const int tid = threadIdx.x, T = blockDim.x;
    const int G = gridDim.x, block = blockIdx.x;
    const int gtid = block * T + tid, GT = G * T;
    const bool leader = block == 0 && tid == 0;
    const int m = lp.m, n = lp.n;
    const size_t MM = (size_t)m * m;                                      // entries of N = its word stride
    auto N_at = [&](int i, int c) { return lp.N + (size_t)c * m + i; };   // word 0 of N(row i, column c)
    const int LIMIT_BITS = 64 * L - 3;  // a value whose magnitude bound reaches this many bits might not fit
    u64* gD = lp.D;
    int* word = lp.shared_words;  // [10], [11] overflow flags of the grid (sync_overflow), [1..9] what the leader decides and the counters of the lists
    int phase = lp.n_art > 0 ? 1 : 2;
    long long pivots[2] = {0, 0};
    int trace_count = 0;
    int status = EX_RUNNING;
    unsigned long long t_last = wall_clock64();
    auto stamp = [&](int k) {  // (one thread, ten reads of the constant 100 MHz counter per pivot)
        if (leader) {
            const unsigned long long t = wall_clock64();
            lp.prof[k] += t - t_last;
            t_last = t;
        }
    };
    unsigned long long t_mark = 0;
    auto mark = [&]() { if (leader) t_mark = wall_clock64(); };  // (diagnostic: parts of a step, prof[10, 11, 14, 15, 18, 19])
    auto lap = [&](int k) {
        if (leader) {
            const unsigned long long t = wall_clock64();
            lp.prof[k] += t - t_mark;
            t_mark = t;
        }
    };
    unsigned long long products_needed = 0, products_issued = 0;  // this thread's word products in the update of N, whole run
    int n_swaps = 0;  // pivots whose fused update left N in the other buffer (lp.N and lp.N_alt swapped)
    u64* const xt_of_host = lp.xt;  // (x~_B swaps as well; the final x~_B goes where the host reads it)
    bool on_matrix_cores = false;  // the update of N by mfma_update_tile
    if constexpr (L >= 16) on_matrix_cores = lp.mfma_update != 0;
    int parity = 0;  // the partial arrays of the grid reductions alternate, so that a fast workgroup never overwrites what a slow one still reads
    if (tid == 0) s_overflow = 0;
    __syncthreads();
    auto flag_overflow = [&](int bits) {
        if (bits >= LIMIT_BITS) s_overflow = 1;
    };
    auto log2_ceil = [](int count) { return 32 - __clz(count > 1 ? count - 1 : 1) + 1; };
    // grid barrier that also tells every workgroup whether any of them saw a value that might not fit
    // (Two flags, taken in turn: with one, a workgroup that left call k early and raised the flag on its way to call k + 1 -- the zero-level
    //  pivots make two calls with no other barrier between them -- could be seen by a slow workgroup still reading the flag of call k; that one
    //  stopped a barrier earlier than the rest and the launch hung.  A raised flag ends the run at the call it was raised for, in every
    //  workgroup, so nothing is ever cleared.)
    int overflow_calls = 0;
    auto sync_overflow = [&]() {
        int* flag = word + 10 + (overflow_calls & 1);
        ++overflow_calls;
        __syncthreads();
        if (tid == 0 && s_overflow) atomicOr(flag, 1);
        grid.sync();
        return *(volatile int*)flag != 0;
    };
    // arg-max of (key, smallest rank) over the grid: the winner in every thread of every workgroup
    auto grid_argbest = [&](double& key, unsigned long long& rank) {
        block_argbest(key, rank, s_key, s_rank);
        if (tid == 0) {
            lp.part_key[parity * G + block] = key;
            lp.part_rank[parity * G + block] = rank;
        }
        grid.sync();
        double bk = 0.0;
        unsigned long long br = RANK_NONE;
        for (int b = tid; b < G; b += T) {
            const double k = lp.part_key[parity * G + b];
            const unsigned long long r = lp.part_rank[parity * G + b];
            if (r != RANK_NONE && (br == RANK_NONE || k > bk || (k == bk && r < br))) { bk = k; br = r; }
        }
        block_argbest(bk, br, s_key, s_rank);
        key = bk;
        rank = br;
        parity ^= 1;
    };
    // a value decided by the leader, for everybody (each call site has its own slot; a slot is rewritten one pivot later at the earliest)
    auto broadcast = [&](int slot, int value_of_leader) {
        if (leader) word[slot] = value_of_leader;
        grid.sync();
        return word[slot];
    };
    // drive_row >= 0: the zero-level pivots of phase_one.rs:232-278 are under way, this is the next row to look at
    int drive_row = -1;
    int n_removed = 0;  // redundant rows found there
    if (lp.resume[0] != 0) {  // the loop-carried state of the run this one continues (see the host driver)
        phase = lp.resume[1];
        pivots[0] = lp.resume[2];
        pivots[1] = lp.resume[3];
        trace_count = lp.resume[4];
        drive_row = lp.resume[5];
        n_removed = lp.resume[6];
    }
    for (size_t e = gtid; e < MM; e += GT) lp.N_bits[e] = big_bits(big_load_s<L>(lp.N + e, MM));
    if (leader) {
        lp.neg_list[n] = 0;
        word[9] = 0;
    }
    grid.sync();
    int at_phase = phase, at_drive_row = drive_row, at_removed = n_removed;  // ... at the start of the current turn of the loop
    bool have_xb = false;  // x~_B belongs to the current basis
    bool dinv_ready = false;  // lp.next_dinv holds 1 / D_odd of the current D
    int y_phase = 0;       // lp.y = c_B' N belongs to the current basis and to this phase's costs (0: to neither); on the matrix cores a pivot
                           // carries it along as one more row of N, otherwise every pricing pass forms it
    while (status == EX_RUNNING) {
        if (pivots[0] + pivots[1] >= lp.max_pivots) { status = EX_PIVOT_LIMIT; break; }
        at_phase = phase;
        at_drive_row = drive_row;
        at_removed = n_removed;
        // (From 16 limbs on no thread holds a copy of D, of 1 / D_odd, of alpha~_p or of alpha~_p u: an integer is 128 to 1024 bytes of
        //  scratch memory per thread, 134 MB a copy over the grid at 128 limbs -- four of them cost 25FV47 0.4 ms of every pivot.  One
        //  thread of the workgroup reads what is needed of them into LDS; the words of D are read from memory where they are used.)
        Big<L> D;
        int D_bits = 0;
        if constexpr (L < 16) {
            D = big_load<L>(gD);
            D_bits = big_bits(D);
        }
        // 1 / D_odd modulo 2^(64 L) (D = 2^shift D_odd): every exact quotient of this turn is a truncated product with it.  Every
        // workgroup for itself: cheap, and no exchange is needed this way.
        if constexpr (L < 16) {
            if (tid == 0) {
                const Big<L> D0 = big_load<L>(gD);
                const int shift = big_ctz(D0);
                const Big<L> odd = big_sar(D0, shift);
#pragma unroll L <= 8 ? L : 1
                for (int k = 0; k < L; ++k) s_words[0][k] = odd.w[k];
                s_shift = shift;
                s_D_bits = big_bits(D0);
                int exponent = 0;
                s_mD = big_mantissa(D0, &exponent);
                s_eD = exponent;
            }
            __syncthreads();
        } else {
            // a thread per word of D (D > 0): its lowest and its highest non-zero word from the waves' ballots, the odd part shifted down
            // by every thread for its own word (one thread with the integer in scratch memory took a tenth of a millisecond per pivot)
            __shared__ unsigned long long s_nonzero[EX_THREADS / WAVE];
            const u64 w = tid < L ? gD[tid] : 0ull;
            const unsigned long long nonzero = __ballot(w != 0);
            if ((tid & (WAVE - 1)) == 0) s_nonzero[tid / WAVE] = nonzero;
            if (tid < L) s_words[1][tid] = w;
            __syncthreads();
            int low = 0, top = 0;
            for (int wv = EX_THREADS / WAVE - 1; wv >= 0; --wv)
                if (s_nonzero[wv] != 0) low = WAVE * wv + __ffsll((long long)s_nonzero[wv]) - 1;
            for (int wv = 0; wv < EX_THREADS / WAVE; ++wv)
                if (s_nonzero[wv] != 0) top = WAVE * wv + 63 - __clzll((long long)s_nonzero[wv]);
            const u64 w_low = s_words[1][low], w_top = s_words[1][top];
            const int shift_bits = 64 * low + __ffsll((long long)w_low) - 1;
            if (tid < L) {
                const int ws = shift_bits >> 6, bs = shift_bits & 63;
                const u64 lo = tid + ws < L ? s_words[1][tid + ws] : 0ull, hi = tid + ws + 1 < L ? s_words[1][tid + ws + 1] : 0ull;
                s_words[0][tid] = bs ? (lo >> bs) | (hi << (64 - bs)) : lo;
            }
            if (tid == 0) {
                s_shift = shift_bits;
                s_D_bits = 64 * top + (64 - __clzll((long long)w_top));
                double x = (double)w_top;  // (big_mantissa's numbers: the two leading words, the exponent of the lower one)
                if (top > 0) x = x * 18446744073709551616.0 + (double)s_words[1][top - 1];
                s_mD = x;
                s_eD = 64 * (top > 0 ? top - 1 : 0);
            }
            __syncthreads();
        }
        if constexpr (L >= 16) D_bits = s_D_bits;
        mark();
        if (dinv_ready) {  // (formed beside the previous pivot's update)
            if (tid < L) s_dinv[tid] = lp.next_dinv[tid];
            __syncthreads();
        } else if constexpr (L >= 16) wave_inverse_odd<L>(s_words[0], s_dinv, s_words[1], s_words[2]);
        else block_inverse_odd<L>(s_words[0], s_dinv, s_words[1], s_words[2], s_part);
        lap(14);
        const int shift = s_shift;
        Big<L> Dinv;
        if constexpr (L < 16) {
#pragma unroll L <= 8 ? L : 1
            for (int k = 0; k < L; ++k) Dinv.w[k] = s_dinv[k];
        }
        // ---- x~_B = N b: a thread per (row, chunk of 32 columns), then a thread per row over its chunks (same bounds as the serial loop).
        //      Only on the first turn of a run: a pivot updates x~_B like one more column of N (below) -- recomputing it was 15 % of E226. ----
        if (!have_xb) {
            have_xb = true;
            constexpr int XC = 32;
            const int chunks = (m + XC - 1) / XC;
            for (long long pair = gtid; pair < (long long)m * chunks; pair += GT) {
                const int i = (int)(pair / chunks), c = (int)(pair - (long long)i * chunks);
                Big<L> acc = big_from<L>(0);
                int widest = 0;
                for (int k = c * XC; k < min(m, (c + 1) * XC); ++k) {
                    const i64 b = lp.rhs[k];
                    if (b == 0) continue;
                    const Big<L> nik = big_load_s<L>(N_at(i, k), MM);
                    acc = big_add(acc, big_mul_small(nik, b));
                    widest = max(widest, big_bits(nik) + small_bits(b));
                }
                big_store(lp.x_part + (size_t)pair * L, acc);
                lp.x_bits[pair] = widest;
            }
            grid.sync();
            for (int i = gtid; i < m; i += GT) {
                Big<L> acc = big_from<L>(0);
                int widest = 0;
                for (int c = 0; c < chunks; ++c) {
                    acc = big_add(acc, big_load<L>(lp.x_part + ((size_t)i * chunks + c) * L));
                    widest = max(widest, lp.x_bits[(size_t)i * chunks + c]);
                }
                flag_overflow(widest + log2_ceil(m));
                big_store(lp.xt + (size_t)i * L, acc);
                lp.xt_bits[i] = big_bits(acc);
            }
        }
        int q = -1, p = -1;
        stamp(0);
        if (drive_row < 0) {
            // ---- pricing: c~_j and the key estimate for every non-basic, non-artificial column (pivot_rule.rs:221-241) --------
            // Pass A, a thread per (column j, row i): a_ij = (N a_j)_i exactly, its share of the weight estimate as a double, its bit
            // bound.  (One thread per column walking all m rows -- the one-workgroup form -- was 80 % of the run on the grid: 126 ms
            // per pass on E226, a few hundred busy threads.)  Pass B, a thread per column: c~_j and the weight from the stored terms,
            // in the order and with the bounds of the serial loop -- the estimates are the same bits as before.
            // (Round 4.  A thread per pair that held N(i, r), its multiple and the running sum as three integers in scratch memory
            //  made ~400 scratch accesses for every 32 words it read from N: 72 of 25FV47's 147 s at 128 limbs.  Now a WAVE takes a
            //  column and 64 neighbouring rows and the sum is formed word by word, least significant first: the words of the
            //  N(i, r_e) come straight from memory (word-major: one coalesced access per operand), the positive and the negative
            //  multiples run in two carry-save accumulators, their difference is stored as it appears, and nothing is held but the
            //  carries.  The bit bound comes from N_bits, the double from the two leading words gathered on the way: same numbers.)
            // Round 5: c~_j = c_j D - (c_B' N) a_j.  The row vector y = c_B' N is ONE pass over N (a wave per column); with it every
            // c~_j is a handful of small multiples of y's entries, and the products N a_j -- what the weights need, the expensive pass: every
            // column of N once per non-zero of its row of A -- are formed only for the columns with c~_j < 0, the only ones that can
            // enter.  (Rounds 2-4 formed N a_j for EVERY non-basic column and c~_j from those: 3.0 + 1.5 of 25FV47's 9.7 s.)  The
            // same integers c~_j, the same doubles for the keys of the candidates, so the same pivots.
#ifdef RELP_Y_ALWAYS
            const bool form_y = true;  // (diagnostic build: the pass over N every pivot, as before round 5)
#else
            const bool form_y = y_phase != phase;  // (the same in every workgroup)
#endif
            if (form_y)
                for (int i = gtid; i < m; i += GT) lp.cb_row[i] = phase == 1 ? lp.cost1[lp.basis[i]] : lp.cost2[lp.basis[i]];
            // (the counters of neg_list and of the columns to be formed exactly are zero here: reset at the launch and by the bookkeeping of
            //  every pivot, behind a barrier -- a barrier of its own for that was one of twenty a pivot, 11 us each on 512 workgroups)
            const int eD = s_eD;
            const double mD = s_mD;
            const int lane = tid & (WAVE - 1);
            mark();
            if (form_y) grid.sync();  // (cb_row)
            lap(26);
            if (form_y) {
                price_form_y<L>(lp, LIMIT_BITS, &s_overflow);
                grid.sync();
                y_phase = phase;
            }
            mark();
#ifdef RELP_PRICE_BLOCK_TIMES
            const unsigned long long t_block0 = wall_clock64();
#endif
            price_reduced_costs<L>(lp, phase, mD, eD, D_bits, LIMIT_BITS, &s_overflow);
#ifdef RELP_PRICE_BLOCK_TIMES
            if (L == 128 && (tid & 63) == 0) {  // (diagnostic build: histogram of the waves' times in the pass, in prof[33 .. 37] of the 128-limb run)
                const unsigned long long dt = wall_clock64() - t_block0;  // ticks of 10 ns
                const int slots[5] = {33, 34, 35, 36, 37};  // (free outside the `make stamps` build, whose tile timers they are)
                const int bin = dt < 1500 ? 0 : dt < 3000 ? 1 : dt < 6000 ? 2 : dt < 12000 ? 3 : 4;
                atomicAdd(&lp.prof[slots[bin]], 1ull);
            }
#endif
            lap(27);
            grid.sync();
            lap(28);
            stamp(1);
            const int n_negative = lp.neg_list[n];
            price_estimates<L>(lp, mD, eD);  // the terms of the weight estimates of the columns that can enter, from the leading words of N
            lap(29);
            grid.sync();
            lap(15);
            // ... and their keys: the weight estimate is the sum of the stored terms, key = (c~_j / D)^2
            // / that; a column whose terms may be off by more than 1e-11 of the sum in all goes on the list of those to be formed exactly
            const double error_allowed = lp.price_exactly ? -1.0 : 1e-11;
            auto form_keys = [&](const int* list, int count, bool list_inexact) {
                for (int c = gtid / WAVE; c < count; c += GT / WAVE) {
                    const int j = list[c];
                    const size_t base = (size_t)(j - lp.n_art) * m;
                    // (a lane adds every 64th row, the lanes' sums are added as a tree -- the same order on any grid; the sum in the order of
                    //  the rows, every term handed round the wave, was a chain of m dependent additions: 30 us of a pivot on 25FV47)
                    double terms = 0.0, errors = 0.0;
                    for (int i = lane; i < m; i += WAVE) {
                        terms += lp.price_term[base + i];
                        errors += lp.price_err[base + i];
                    }
                    for (int d = 1; d < WAVE; d *= 2) {
                        terms += __shfl_xor(terms, d);
                        errors += __shfl_xor(errors, d);
                    }
                    const double sumsq = weight_as_double(lp, j) + terms;
                    const double cd = lp.cd[j];
                    if (lane == 0) {
                        lp.key[j] = cd * cd / sumsq;
                        if (list_inexact && !(errors <= error_allowed * sumsq)) lp.bracket[atomicAdd(&word[9], 1)] = j;
                    }
                }
            };
            form_keys(lp.neg_list, n_negative, true);
            if (leader) {  // (diagnostic: how many of the priced columns have a negative reduced cost)
                lp.prof[30] += n_negative;
                lp.prof[31] += 1;
            }
            if (sync_overflow()) { status = EX_OVERFLOW; break; }  // (before any decision is taken on values that may not have fit)
            const int n_inexact = word[9];
            if (n_inexact > 0) {  // the columns whose estimate is not good enough: their products exactly, their keys from those
                price_products<L>(lp, lp.bracket, n_inexact, mD, eD, LIMIT_BITS, &s_overflow);
                grid.sync();
                if (leader) lp.prof[13] += n_inexact;
                form_keys(lp.bracket, n_inexact, false);  // (exact terms carry no error; the list is this call's input: nothing is listed again, its counter is reset by the pivot's bookkeeping)
                if (sync_overflow()) { status = EX_OVERFLOW; break; }
            }
            stamp(9);
            // the largest estimate; ties to the larger index ("last maximum", pivot_rule.rs:230-240)
            double best = 0.0;
            unsigned long long rank = RANK_NONE;
            for (int j = lp.n_art + gtid; j < n; j += GT) {
                const double k = lp.key[j];
                if (k > 0.0) {
                    const unsigned long long r = (unsigned long long)(0x7fffffff - j);
                    if (rank == RANK_NONE || k > best || (k == best && r < rank)) { best = k; rank = r; }
                }
            }
            grid_argbest(best, rank);
            if (rank != RANK_NONE) {
                q = 0x7fffffff - (int)rank;
                // Every column whose estimate is within 1e-9 of the best is compared exactly (c~^2 gamma~ cross products).  The exact
                // weight of a candidate is m squarings of L-limb integers -- a millisecond for one thread at 32 limbs, and degenerate
                // LPs have a hundred candidates per pivot (round 2 did them one after the other: most of E226's 100 s) -- so every
                // candidate gets a thread of its own, anywhere on the grid; the tournament over the finished weights is the leader's.
                for (int j = lp.n_art + gtid; j < n; j += GT)
                    if (lp.key[j] >= best * (1.0 - 1e-9)) lp.cand[atomicAdd(&word[4], 1)] = j;
                grid.sync();
                stamp(2);
                const int n_cand = word[4];
                if (leader) lp.prof[12] += n_cand;
                int winner = q;
                if (n_cand > 1) {
                    // gamma~_j = w_j D^2 + sum_i w_i (N a_j)_i^2 exactly ((2 L + 2)-limb sums of squares).  The (N a_j)_i are the ones
                    // the pricing pass stored; a thread per (candidate, row) squares one of them, a thread per candidate adds them up.
                    constexpr int GW = 2 * L + 2;
                    // (the terms of GAMMA_BATCH candidates at a time: room for the terms of EVERY column that could tie was 2.6 GB at 128
                    //  limbs on 25FV47, allocated and freed at every width for the handful of ties a pivot has)
                    for (int c0 = 0; c0 < n_cand; c0 += EX_GAMMA_BATCH) {
                    const int batch = min(EX_GAMMA_BATCH, n_cand - c0);
                    price_products<L>(lp, lp.cand + c0, batch, mD, eD, LIMIT_BITS, &s_overflow);  // (the estimates left nothing in price_a)
                    grid.sync();
                    if constexpr (L >= 16) exact_weight_terms<L>(lp, c0, batch);  // a wave per term
                    else
                    for (long long pair = gtid; pair < (long long)batch * (m + 1); pair += GT) {
                        const int c = c0 + (int)(pair / (m + 1)), i = (int)(pair - (long long)(c - c0) * (m + 1));
                        const int j = lp.cand[c];
                        u64* out = lp.gamma_terms + ((size_t)(c - c0) * (m + 1) + i) * GW;
                        if (i == m) weighted_square<L>(big_load<L>(gD), lp.weight[2 * j], lp.weight[2 * j + 1], out);
                        else weighted_square<L>(big_load_s<L>(lp.price_a + (size_t)(c - c0) * m + i, (size_t)EX_PRODUCT_SLOTS * m), lp.weight[2 * lp.basis[i]], lp.weight[2 * lp.basis[i] + 1], out);
                    }
                    grid.sync();
                    // (a wave per candidate, a lane per word of the sum: the terms are read a whole row of words at a time, the carries
                    //  run through the words once at the end.  One thread per candidate walked (m + 1) (2 L + 2) dependent additions:
                    //  a third of SCORPION's solve, whose degenerate pivots tie a hundred candidates at a time.)
                    constexpr int WPL = (GW + WAVE - 1) / WAVE;  // words per lane
                    for (int c = c0 + gtid / WAVE; c < c0 + batch; c += GT / WAVE) {
                        const int ln = tid & (WAVE - 1);
                        const u64* terms = lp.gamma_terms + (size_t)(c - c0) * (m + 1) * GW;
                        u128 total[WPL];
#pragma unroll
                        for (int u = 0; u < WPL; ++u) total[u] = 0;
                        for (int i = 0; i <= m; ++i) {
#pragma unroll
                            for (int u = 0; u < WPL; ++u) {
                                const int k = ln + u * WAVE;
                                if (k < GW) total[u] += terms[(size_t)i * GW + k];
                            }
                        }
                        u64* g = lp.gamma + (size_t)c * GW;
                        u128 carry = 0;
#pragma unroll
                        for (int u = 0; u < WPL; ++u) {
                            const int count = GW - u * WAVE < WAVE ? GW - u * WAVE : WAVE;
                            for (int src = 0; src < count; ++src) {
                                const u64 lo = (u64)total[u], hi = (u64)(total[u] >> 64);
                                const u64 word_lo = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(lo >> 32), src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, src);
                                const u64 word_hi = ((u64)(unsigned)__builtin_amdgcn_readlane((int)(hi >> 32), src) << 32) | (unsigned)__builtin_amdgcn_readlane((int)hi, src);
                                const u128 t = (((u128)word_hi << 64) | word_lo) + carry;
                                if (ln == 0) g[u * WAVE + src] = (u64)t;
                                carry = t >> 64;
                            }
                        }
                    }
                    grid.sync();
                    }
                    stamp(3);
                    // the tournament as a tree over the grid: the order "larger exact key, then larger column" is total, so any bracket
                    // gives the winner of the serial scan (which was 0.1 ms per comparison at 32 limbs, hundreds of candidates on SCORPION)
                    for (int c = gtid; c < n_cand; c += GT) lp.bracket[c] = c;
                    grid.sync();
                    for (int stride = 1; stride < n_cand; stride *= 2) {
                        if constexpr (L >= 16) {  // a WAVE per comparison
                            tournament_round<L>(lp, n_cand, stride);
                        } else {
                        for (long long c = (long long)gtid * 2 * stride; c + stride < n_cand; c += (long long)GT * 2 * stride) {
                            const int ca = lp.bracket[c], cb2 = lp.bracket[c + stride];
                            const int ja = lp.cand[ca], jb = lp.cand[cb2];
                            const int cmp = compare_keys<L>(big_load<L>(lp.ctil + (size_t)jb * L), lp.gamma + (size_t)cb2 * (2 * L + 2),
                                                            big_load<L>(lp.ctil + (size_t)ja * L), lp.gamma + (size_t)ca * (2 * L + 2));
                            if (cmp > 0 || (cmp == 0 && jb > ja)) lp.bracket[c] = cb2;
                        }
                        }
                        grid.sync();
                    }
                    winner = lp.cand[lp.bracket[0]];
                }
                q = winner;  // (every workgroup read the same bracket behind the tournament's last barrier)
                stamp(4);
            }
            if (q < 0) {  // no candidate: the end of this phase
                if (phase == 2) { status = EX_OPTIMAL; break; }
                // phase one is over: feasible iff the artificial variables sum to zero (phase_one.rs:160-176)
                int verdict = 0;  // bit 0: an artificial is positive; bits 1..: the number of basic artificials
                if (leader) {
                    int positive = 0, basic_artificials = 0;
                    for (int i = 0; i < m; ++i)
                        if (lp.basis[i] < lp.n_art) {
                            ++basic_artificials;
                            if (!big_zero(big_load<L>(lp.xt + (size_t)i * L))) positive = 1;
                        }
                    verdict = positive | (basic_artificials << 1);
                }
                verdict = broadcast(2, verdict);
                if (verdict & 1) { status = EX_INFEASIBLE; break; }
                if ((verdict >> 1) > 0) { drive_row = 0; continue; }
                phase = 2;
                continue;
            }
        } else {
            // ---- zero-level pivots: the next row whose basic variable is artificial, the first non-basic column with a
            //      non-zero entry in that row of the tableau (phase_one.rs:232-278) ----------------------------------------
            int r = drive_row;  // (the basis is the same for everybody: the previous pivot's bookkeeping is behind a barrier)
            while (r < m && lp.basis[r] >= lp.n_art) ++r;
            if (r >= m) {
                drive_row = -1;
                phase = 2;
                grid.sync();  // (x~_B of this turn is complete before the next turn rewrites it)
                continue;
            }
            // (N a_j)_r for every candidate column; the first non-zero wins
            unsigned long long first = RANK_NONE;
            double dummy = 0.0;
            for (int j = lp.n_art + gtid; j < n; j += GT) {
                if (lp.pos[j] >= 0) continue;
                Big<L> a = big_from<L>(0);
                for (int e = lp.col_start[j]; e < lp.col_start[j + 1]; ++e)
                    a = big_add(a, big_mul_small(big_load_s<L>(N_at(r, lp.row_index[e]), MM), lp.value[e]));
                if (!big_zero(a) && (first == RANK_NONE || (unsigned long long)j < first)) { first = (unsigned long long)j; dummy = 1.0; }
            }
            grid_argbest(dummy, first);
            if (first == RANK_NONE) {
                // The row of the tableau is zero on every column that could enter: the constraint is redundant and the reference
                // REMOVES it (phase_one.rs:232-278 collects such rows, `RemoveRows` re-indexes the rest,
                // filter/generic_wrapper.rs:98-205).  Here the row stays, with its zero-level artificial basic: alpha_r = 0 for
                // every entering column from now on, so it never takes part in a ratio test and adds nothing to a steepest-edge
                // weight; only the ROW INDICES reported for phase two are shifted as the removal shifts them.
                if (leader) {
                    lp.removed[r] = 1;
                    for (int i = r + 1; i < m; ++i) lp.removed[m + i] += 1;  // (the count the trace of phase two subtracts: kept here, once per
                }                                                            //  removed row, not summed over the rows above p at every pivot)
                ++n_removed;
                drive_row = r + 1;
                continue;
            }
            q = (int)first;
            p = r;
            drive_row = r + 1;
        }
        // y = c_B' N is carried through this pivot when the update runs on the matrix cores and q came out of the pricing pass (whose
        // c~_q is the factor; a zero-level pivot of phase one leaves y to the next pricing pass)
#ifdef RELP_NO_Y_RIDE
        const bool y_rides = false;  // (diagnostic build)
#else
        const bool y_rides = on_matrix_cores && p < 0 && y_phase == phase;
#endif
        // ---- alpha~_q = N a_q (tableau/mod.rs:126-130) -----------------------------------------------------------------------
        if constexpr (L >= 16) {  // the wide types: streamed like the pricing pass in chunks of words (a thread per row with its integers in scratch: 1.2 ms a pivot at 128 limbs)
            // (rounds 4-5 copied a priced column's products out of price_a; the pricing pass now forms estimates only.  price_a holds the
            //  chunks: (L / 16) 19 m words -- one chunk at 16 limbs, which took a path of its own until round 6)
            mark();
            entering_column_chunks<L>(lp, q);
            grid.sync();
            lap(10);
            entering_column_rows<L>(lp, q, s_dinv, LIMIT_BITS, &s_overflow, y_rides, word + 8);  // ... added up, and the rows' factors -alpha~_i u
            lap(11);
        } else {
            for (int i = gtid; i < m; i += GT) {
                Big<L> a = big_from<L>(0);
                int awide = 0;
                for (int e = lp.col_start[q]; e < lp.col_start[q + 1]; ++e) {
                    const Big<L> nir = big_load_s<L>(N_at(i, lp.row_index[e]), MM);
                    a = big_add(a, big_mul_small(nir, lp.value[e]));
                    awide = max(awide, big_bits(nir) + small_bits(lp.value[e]));
                }
                flag_overflow(awide + log2_ceil(lp.col_start[q + 1] - lp.col_start[q]));
                big_store(lp.alpha + (size_t)i * L, a);
                lp.x_bits[i] = big_bits(a);  // (the fit test of the update below wants it once per ENTRY of N)
                if constexpr (L < 16) big_store_s(lp.x_part + i, (size_t)m, big_negate(big_mul_lo(a, Dinv)));  // -alpha~_i / D_odd: the row's factor of the update below
            }
        }
        if (sync_overflow()) { status = EX_OVERFLOW; break; }
        stamp(5);
        if (p < 0) {
            // ---- ratio test: min x~_i / alpha~_i over alpha~_i > 0, ties to the lowest basic column (tableau/mod.rs:287-313) --
            double best = 0.0;
            unsigned long long rank = RANK_NONE;
            // The estimate x~_i / alpha~_i from the two leading words of each: their places are known from the bit lengths kept beside the
            // integers (x_bits for alpha~, xt_bits for x~_B) -- four loads per row instead of two integers scanned in scratch memory
            // twice per pivot.  Same double as big_ratio for non-negative x~_i (anything else takes the long way).
            auto ratio_estimate = [&](int i, double* ratio) {  // false: alpha~_i <= 0, the row does not take part
                const int a_bits = lp.x_bits[i];
                const u64* a = lp.alpha + (size_t)i * L;
                if (a_bits == 0 || (i64)a[L - 1] < 0) return false;
                const u64* x = lp.xt + (size_t)i * L;
                if ((i64)x[L - 1] < 0) {
                    *ratio = big_ratio(big_load<L>(x), big_load<L>(a));
                    return true;
                }
                auto leading = [](const u64* v, int bits, int* exponent) {
                    if (bits == 0) { *exponent = 0; return 0.0; }
                    const int top = (bits - 1) >> 6;
                    double value = (double)v[top];
                    if (top > 0) value = value * 18446744073709551616.0 + (double)v[top - 1];
                    *exponent = 64 * (top > 0 ? top - 1 : 0);
                    return value;
                };
                int ex = 0, ea = 0;
                const double mx = leading(x, lp.xt_bits[i], &ex), ma = leading(a, a_bits, &ea);
                *ratio = ldexp(mx / ma, ex - ea);
                return true;
            };
            for (int i = gtid; i < m; i += GT) {
                double ratio;
                if (!ratio_estimate(i, &ratio)) continue;
                const unsigned long long r = ((unsigned long long)(unsigned)lp.basis[i] << 32) | (unsigned)i;
                const double k = -ratio;  // block_argbest maximises
                if (rank == RANK_NONE || k > best || (k == best && r < rank)) { best = k; rank = r; }
            }
            grid_argbest(best, rank);
            if (rank == RANK_NONE) { status = EX_UNBOUNDED; break; }
            p = (int)(rank & 0xffffffffu);
            // the rows whose ratio estimate is within 1e-9 of the smallest: found by everybody, compared exactly by the leader
            {
                const double ratio_p = -best;
                for (int i = gtid; i < m; i += GT) {
                    if (i == p) continue;
                    double ratio;
                    if (!ratio_estimate(i, &ratio)) continue;
                    if (!(ratio <= ratio_p + 1e-9 * fabs(ratio_p) + 1e-300)) continue;
                    lp.cand[atomicAdd(&word[5], 1)] = i;
                }
            }
            grid.sync();
            // the exact minimum ratio among the near-tied rows (and p itself), ties to the lowest basic column: a total order again,
            // decided by a bracket over the grid
            const int n_near = word[5];
            int winner = p;
            if (n_near > 0) {
                if (leader) {
                    lp.cand[n_near] = p;
                    lp.prof[24] += 1;         // (diagnostic: pivots with near-tied rows, and how many of those rows)
                    lp.prof[25] += n_near;
                }
                for (int c = gtid; c <= n_near; c += GT) lp.bracket[c] = c;
                grid.sync();
                for (int stride = 1; stride <= n_near; stride *= 2) {
                    if constexpr (L >= 16) {  // a WAVE per comparison (wave_sign_of_difference); anything negative goes the one-thread way
                        const int lane_r = tid & (WAVE - 1);
                        for (long long c = (long long)(gtid / WAVE) * 2 * stride; c + stride <= n_near; c += (long long)(GT / WAVE) * 2 * stride) {
                            const int ia = lp.cand[lp.bracket[c]], ib = lp.cand[lp.bracket[c + stride]];
                            const u64 *xa = lp.xt + (size_t)ia * L, *xb = lp.xt + (size_t)ib * L, *aa = lp.alpha + (size_t)ia * L, *ab = lp.alpha + (size_t)ib * L;
                            int cmp;
                            if ((i64)xa[L - 1] < 0 || (i64)xb[L - 1] < 0 || (i64)aa[L - 1] < 0 || (i64)ab[L - 1] < 0) {
                                cmp = 0;
                                if (lane_r == 0) cmp = sign_of_difference<L>(big_load<L>(xb), big_load<L>(aa), big_load<L>(xa), big_load<L>(ab));
                                cmp = __shfl(cmp, 0);
                            } else {
                                auto words_of = [](int bits) { return (bits + 63) >> 6; };
                                // x_b / a_b  vs  x_a / a_a   <=>   x_b a_a  vs  x_a a_b   (both alpha > 0)
                                cmp = wave_sign_of_difference<L>(xb, words_of(lp.xt_bits[ib]), aa, words_of(lp.x_bits[ia]), xa, words_of(lp.xt_bits[ia]), ab,
                                                                 words_of(lp.x_bits[ib]), lane_r);
                            }
                            if (lane_r == 0 && (cmp < 0 || (cmp == 0 && lp.basis[ib] < lp.basis[ia]))) lp.bracket[c] = lp.bracket[c + stride];
                        }
                    } else {
                        for (long long c = (long long)gtid * 2 * stride; c + stride <= n_near; c += (long long)GT * 2 * stride) {
                            const int ia = lp.cand[lp.bracket[c]], ib = lp.cand[lp.bracket[c + stride]];
                            // x_b / a_b  vs  x_a / a_a   <=>   x_b a_a  vs  x_a a_b   (both alpha > 0)
                            const int cmp = sign_of_difference<L>(big_load<L>(lp.xt + (size_t)ib * L), big_load<L>(lp.alpha + (size_t)ia * L),
                                                                  big_load<L>(lp.xt + (size_t)ia * L), big_load<L>(lp.alpha + (size_t)ib * L));
                            if (cmp < 0 || (cmp == 0 && lp.basis[ib] < lp.basis[ia])) lp.bracket[c] = lp.bracket[c + stride];
                        }
                    }
                    grid.sync();
                }
                winner = lp.cand[lp.bracket[0]];
                if (leader) word[5] = 0;
            }
            p = winner;  // (as for q)
        }
        stamp(6);
        // ---- the pivot: D' = alpha~_p, N'_i = (alpha~_p N_i - alpha~_i N_p) / D  (exact), row p stays ---------------------------
        if constexpr (L < 16) {
            if (tid == 0) {
                const Big<L> ap = big_load<L>(lp.alpha + (size_t)p * L);
#pragma unroll L <= 8 ? L : 1
                for (int k = 0; k < L; ++k) s_words[0][k] = ap.w[k];
                s_ap_bits = big_bits(ap);
                s_flip = big_neg(ap) ? 1 : 0;
            }
            __syncthreads();
        } else {  // (a thread per word, as for D above; a negative pivot element -- zero-level pivots only -- goes the one-thread way)
            __shared__ unsigned long long s_ap_nonzero[EX_THREADS / WAVE];
            const u64 w = tid < L ? lp.alpha[(size_t)p * L + tid] : 0ull;
            const unsigned long long nonzero = __ballot(w != 0);
            if ((tid & (WAVE - 1)) == 0) s_ap_nonzero[tid / WAVE] = nonzero;
            if (tid < L) s_words[0][tid] = w;
            __syncthreads();
            if (tid == 0) {
                const bool negative = (i64)s_words[0][L - 1] < 0;
                s_flip = negative ? 1 : 0;
                if (negative) {
                    s_ap_bits = big_bits(big_load<L>(lp.alpha + (size_t)p * L));
                } else {
                    int top = -1;
                    for (int wv = 0; wv < EX_THREADS / WAVE; ++wv)
                        if (s_ap_nonzero[wv] != 0) top = WAVE * wv + 63 - __clzll((long long)s_ap_nonzero[wv]);
                    s_ap_bits = top < 0 ? 0 : 64 * top + (64 - __clzll((long long)s_words[0][top]));
                }
            }
            __syncthreads();
        }
        const bool flip = s_flip != 0;  // (only a zero-level pivot can have a negative pivot element): keep D > 0
        const int ap_bits = s_ap_bits;
        // With D = 2^s D_odd and u = 1 / D_odd modulo 2^(64 L):  (alpha~_p u) N_ik - (alpha~_i u) N_pk = 2^s N'_ik modulo 2^(64 L), so the
        // new entry is that value shifted right by s -- known modulo 2^(64 L - s), sign-extended from there, and it must fit there.
        // TWO truncated products per entry (the numerator first and then its product with u were three); alpha~_p u once per
        // workgroup, alpha~_i u once per row (the alpha step).
        if constexpr (L >= ENTER_CHUNK) {
            // (alpha~_p u is the negative of row p's factor, which entering_column_rows has just stored: a thread per word takes the two's
            //  complement -- zero up to the lowest non-zero word, that word's complement plus one, the complements above -- instead of
            //  one more truncated product by every workgroup, 8 us of every pivot at 128 limbs)
            __shared__ unsigned long long s_c1_nonzero[EX_THREADS / WAVE];
            const u64 w = tid < L ? lp.x_part[(size_t)tid * m + p] : 0ull;
            const unsigned long long nonzero = __ballot(w != 0);
            if ((tid & (WAVE - 1)) == 0) s_c1_nonzero[tid / WAVE] = nonzero;
            __syncthreads();
            int lowest = L;
            for (int wv = EX_THREADS / WAVE - 1; wv >= 0; --wv)
                if (s_c1_nonzero[wv] != 0) lowest = WAVE * wv + __ffsll((long long)s_c1_nonzero[wv]) - 1;
            if (tid < L) s_c1[tid] = tid < lowest ? 0ull : tid == lowest ? ~w + 1ull : ~w;
            __syncthreads();
        } else {
            block_mul_lo(s_words[0], L, s_dinv, L, s_c1, L, s_part);
        }
        Big<L> c1;
        if (!on_matrix_cores) {
#pragma unroll L <= 8 ? L : 1
            for (int k = 0; k < L; ++k) c1.w[k] = s_c1[k];
        }
        // Will every new entry fit?  Decided from the bit lengths of the operands BEFORE anything is written (one more read of N:
        // microseconds beside the multiplications below), so that a run that does not fit stops with N, D and the basis as they
        // were at the start of this pivot -- the state the next width continues from (host driver).
        for (int idx = gtid; idx < m * m; idx += GT) {  // (entry idx = column k, row i: the rows run through a wave, N(p, k) is the same word for all of it)
            const int k = idx / m, i = idx - k * m;
            if (i == p) continue;
            const int estimate = max(ap_bits + lp.N_bits[idx], lp.x_bits[i] + lp.N_bits[(size_t)k * m + p]) + 1 - (D_bits - 1);
            if (estimate >= LIMIT_BITS - shift) s_overflow = 1;
        }
        const int xp_bits = lp.xt_bits[p];  // (bit lengths of x~_B: kept by whoever writes an entry, like N_bits)
        for (int i = gtid; i < m; i += GT) {
            if (i == p) continue;
            const int estimate = max(ap_bits + lp.xt_bits[i], lp.x_bits[i] + xp_bits) + 1 - (D_bits - 1);
            if (estimate >= LIMIT_BITS - shift) s_overflow = 1;
        }
        const int cq_bits = y_rides ? word[8] : 0;
        if (y_rides)
            for (int k = gtid; k < m; k += GT) {
                const int estimate = max(ap_bits + lp.y_bits[k], cq_bits + lp.N_bits[(size_t)k * m + p]) + 1 - (D_bits - 1);
                if (estimate >= LIMIT_BITS - shift) s_overflow = 1;
            }
        // The entries of N by what they cost below.  N'_ik = (alpha~_p N_ik - alpha~_i N_pk) / D: two products where N(p, k) != 0 AND
        // alpha~_i != 0, one (the entry is only rescaled) where either is zero, nothing where N_ik is zero as well.  Columns and rows
        // are each split into the two kinds (workgroup 0, ordered lists) and every class of entries is spread over the whole grid by
        // itself: with the entries taken as they come a wave drew ten columns and the slowest wave's draw set the pace, and a row
        // with alpha~_i = 0 inside a wave of rows with alpha~_i != 0 saved nothing.
        // (two workgroups, a list each: one workgroup building both while the grid waited at the barrier below was 20 us of every pivot)
        const int column_splitter = G >= 3 ? 1 : 0, row_splitter = G >= 3 ? 2 : 0;
        if (block == column_splitter || block == row_splitter) {
            __shared__ int s_class_count[EX_THREADS / WAVE][2];
            auto split = [&](auto&& is_first, int* first_list, int* second_list) {  // indices 0 .. m - 1 in order into the two lists; returns the first's length
                int done[2] = {0, 0};
                for (int base = 0; base < m; base += T) {
                    const int k = base + tid;
                    const bool valid = k < m;
                    const bool first = valid && is_first(k);
                    const unsigned long long first_mask = __ballot(first), second_mask = __ballot(valid && !first);
                    const int wave = tid / WAVE, ln = tid & (WAVE - 1);
                    __syncthreads();
                    if (ln == 0) {
                        s_class_count[wave][0] = __popcll(first_mask);
                        s_class_count[wave][1] = __popcll(second_mask);
                    }
                    __syncthreads();
                    int before[2] = {done[0], done[1]}, all[2] = {0, 0};
                    for (int wv = 0; wv < T / WAVE; ++wv)
                        for (int c = 0; c < 2; ++c) {
                            if (wv < wave) before[c] += s_class_count[wv][c];
                            all[c] += s_class_count[wv][c];
                        }
                    const unsigned long long below = (1ull << ln) - 1ull;
                    if (first) first_list[before[0] + __popcll(first_mask & below)] = k;
                    else if (valid) second_list[before[1] + __popcll(second_mask & below)] = k;
                    done[0] += all[0];
                    done[1] += all[1];
                }
                return done[0];
            };
            if (block == column_splitter) {
                const int heavy_columns = split([&](int k) { return lp.N_bits[(size_t)k * m + p] != 0; }, lp.col_heavy, lp.col_light);
                if (tid == 0) word[7] = heavy_columns;
            }
            if (block == row_splitter) {
                const int rows_with_alpha = split([&](int i) { return lp.x_bits[i] != 0; }, lp.row_list, lp.row_list + m);
                if (tid == 0) word[6] = rows_with_alpha;
            }
        }
        if (sync_overflow()) { status = EX_OVERFLOW; break; }
        const int n_heavy = word[7], n_rows_alpha = word[6];
        bool finished_in_tiles = false;
        if constexpr (L >= 16) if (on_matrix_cores) {
            // (the fused epilogue does not negate: a negative pivot element -- zero-level pivots only -- takes the two passes of round 5, in place)
            const bool fused = lp.fused_update != 0 && !flip;
            const UpdateScalars scalars{p, shift, flip ? 1 : 0, ap_bits, D_bits, xp_bits, n_heavy, n_rows_alpha, y_rides ? 1 : 0, cq_bits, fused ? 1 : 0};
            update_on_matrix_cores<L>(lp, scalars, s_c1, s_words[0], products_needed, products_issued, barrier_epoch, barrier_place);
            dinv_ready = G > 1;  // (the last workgroup left 1 / D'_odd in lp.next_dinv)
            finished_in_tiles = fused;
            if (fused) {  // the other buffer holds N now (every workgroup alike; nobody reads N again before the barrier that ends the pivot)
                u64* const was = lp.N;
                lp.N = lp.N_alt;
                lp.N_alt = was;
                int* const was_bits = lp.N_bits;
                lp.N_bits = lp.N_bits_alt;
                lp.N_bits_alt = was_bits;
                ++n_swaps;
                {   // x~_B likewise
                    u64* const x_was = lp.xt;
                    lp.xt = lp.xt_alt;
                    lp.xt_alt = x_was;
                    int* const x_was_bits = lp.xt_bits;
                    lp.xt_bits = lp.xt_bits_alt;
                    lp.xt_bits_alt = x_was_bits;
                }
                if (y_rides) {  // ... and y where it rode along
                    u64* const y_was = lp.y;
                    lp.y = lp.y_alt;
                    lp.y_alt = y_was;
                    int* const y_was_bits = lp.y_bits;
                    lp.y_bits = lp.y_bits_alt;
                    lp.y_bits_alt = y_was_bits;
                }
            }
        }
        if (!on_matrix_cores)
        for (long long unit = gtid; unit < (long long)n_heavy * n_rows_alpha; unit += GT) {  // N(p, k) != 0 and alpha~_i != 0: two products
            const int kk = (int)(unit / n_rows_alpha), i = lp.row_list[(int)(unit - (long long)kk * n_rows_alpha)];
            if (i == p) continue;
            const int k = lp.col_heavy[kk];
            const size_t idx = (size_t)k * m + i;
            const Big<L> ri = big_load_s<L>(lp.x_part + i, (size_t)m);
            const Big<L> nik = big_load_s<L>(lp.N + idx, MM);
            const Big<L> npk = big_load_s<L>(N_at(p, k), MM);
            Big<L> quotient;
            if constexpr (L >= 16) {
                // 2^shift N'_ik is known to fit `needed` bits (the bound of the fit test above) and the products are exact modulo any
                // power of two: only the 4-word blocks that hold it are formed, the largest count of the wave for all of it (the
                // integers of a run rarely fill the width its largest one forced: 16.5 -> 11.8 s of 25FV47's update at 128 limbs).
                const int needed = max(ap_bits + lp.N_bits[idx], lp.x_bits[i] + lp.N_bits[(size_t)k * m + p]) + 1 - (D_bits - 1) + shift + 2;
                int blocks = min(L / 4, max(1, (needed + 255) / 256));
                products_needed += 16ull * blocks * (blocks + 1);  // two truncated products of 4 x 4-word blocks
                for (int d = 1; d < WAVE; d *= 2) blocks = max(blocks, __shfl_xor(blocks, d));
                blocks = min(L / 4, blocks);  // (a lane that sits this turn out contributes whatever its register holds)
                products_issued += 16ull * blocks * (blocks + 1);
                const Big<L> numerator = big_mul_add_lo_blocked<L>(s_c1, nik, ri, npk, blocks);  // (ri is stored negated)
                quotient = big_sar(big_sign_extend(numerator, 4 * blocks), shift);
            } else {
                products_needed += (unsigned long long)L * (L + 1);
                products_issued += (unsigned long long)L * (L + 1);
                quotient = big_sar(big_add(big_mul_lo(c1, nik), big_mul_lo(ri, npk)), shift);
            }
            if (flip) quotient = big_negate(quotient);
            big_store_s(lp.N + idx, MM, quotient);
            lp.N_bits[idx] = big_bits(quotient);
        }
        // ... and the entries that are only rescaled: the rows with alpha~_i = 0 of those columns, then every row of the other columns
        const long long rescaled_a = (long long)n_heavy * (m - n_rows_alpha), rescaled_b = (long long)(m - n_heavy) * m;
        if (!on_matrix_cores)
        for (long long unit = gtid; unit < rescaled_a + rescaled_b; unit += GT) {
            int k, i;
            if (unit < rescaled_a) {
                const int kk = (int)(unit / (m - n_rows_alpha));
                k = lp.col_heavy[kk];
                i = lp.row_list[m + (int)(unit - (long long)kk * (m - n_rows_alpha))];
            } else {
                const long long rest = unit - rescaled_a;
                const int kk = (int)(rest / m);
                k = lp.col_light[kk];
                i = (int)(rest - (long long)kk * m);
            }
            if (i == p) continue;
            const size_t idx = (size_t)k * m + i;
            const bool zero = lp.N_bits[idx] == 0;  // (a zero stays a zero)
            Big<L> quotient;
            if constexpr (L >= 16) {
                const int needed = zero ? 0 : ap_bits + lp.N_bits[idx] + 1 - (D_bits - 1) + shift + 2;
                int blocks = min(L / 4, max(1, (needed + 255) / 256));
                const int own_blocks = blocks;
                for (int d = 1; d < WAVE; d *= 2) blocks = max(blocks, __shfl_xor(blocks, d));
                blocks = min(L / 4, blocks);
                if (zero) continue;
                products_needed += 8ull * own_blocks * (own_blocks + 1);
                products_issued += 8ull * blocks * (blocks + 1);
                quotient = big_sar(big_sign_extend(big_mul_lo_blocked_p<L>(s_c1, big_load_s<L>(lp.N + idx, MM), blocks), 4 * blocks), shift);
            } else {
                if (zero) continue;
                products_needed += (unsigned long long)L * (L + 1) / 2;
                products_issued += (unsigned long long)L * (L + 1) / 2;
                quotient = big_sar(big_mul_lo(c1, big_load_s<L>(lp.N + idx, MM)), shift);
            }
            if (flip) quotient = big_negate(quotient);
            big_store_s(lp.N + idx, MM, quotient);
            lp.N_bits[idx] = big_bits(quotient);
        }
        if (!on_matrix_cores) {   // x~_B = N b is one more column of N: x~'_i = (alpha~_p x~_i - alpha~_i x~_p) / D, row p stays
            const Big<L> xp = big_load<L>(lp.xt + (size_t)p * L);
            for (int i = gtid; i < m; i += GT) {
                if (i == p) continue;
                const Big<L> ri = big_load_s<L>(lp.x_part + i, (size_t)m);
                Big<L> quotient = big_sar(big_add(big_mul_lo(c1, big_load<L>(lp.xt + (size_t)i * L)), big_mul_lo(ri, xp)), shift);  // (ri is stored negated)
                if (flip) quotient = big_negate(quotient);
                big_store(lp.xt + (size_t)i * L, quotient);
                lp.xt_bits[i] = big_bits(quotient);
            }
        }
        stamp(7);
        // (row p is an operand of every other row above -- nobody may still be reading it; on the matrix cores the barrier between the tiles
        //  and the second pass has seen to that, and the second pass reads and writes nothing that is touched below)
        if (!on_matrix_cores) grid.sync();
        if (flip) {
            for (int k = gtid; k < m; k += GT) big_store_s(N_at(p, k), MM, big_negate(big_load_s<L>(N_at(p, k), MM)));
            if (gtid == 0) big_store(lp.xt + (size_t)p * L, big_negate(big_load<L>(lp.xt + (size_t)p * L)));
        }
        if constexpr (L >= 16)
            if (block == 0 && !flip)  // (one thread with the integer in scratch memory took 0.1 ms of every pivot at 128 limbs)
                for (int k = tid; k < L; k += T) gD[k] = lp.alpha[(size_t)p * L + k];
        if (leader) {
            word[4] = 0;  // (the candidate counter of the next pricing pass)
            word[9] = 0;  // (... of the columns whose estimated weight has to be formed exactly)
            lp.neg_list[n] = 0;
            const int leaving = lp.basis[p];
            if (L < 16 || flip) {  // (the usual case at the wide types -- D' = alpha~_p as it is -- is copied a thread per word below)
                const Big<L> ap = big_load<L>(lp.alpha + (size_t)p * L);
                big_store(gD, flip ? big_negate(ap) : ap);
            }
            lp.basis[p] = q;
            lp.pos[q] = p;
            lp.pos[leaving] = -1;
            if (trace_count < lp.trace_capacity) {
                int p_reported = p;
                if (phase == 2 && n_removed > 0)  // the reference's phase two counts the rows that are left
                    p_reported -= lp.removed[m + p];  // (the leader walking up to m flags was 50 us of every pivot of phase two on 25FV47)
                lp.trace[4 * trace_count] = phase;
                lp.trace[4 * trace_count + 1] = q;
                lp.trace[4 * trace_count + 2] = p_reported;
                lp.trace[4 * trace_count + 3] = leaving;
            }
        }
        if (!y_rides) y_phase = 0;  // y belongs to the basis that was
        ++trace_count;
        pivots[phase - 1]++;
        // (step timers: a fused update has no barrier of its own -- the leader's wait for the last tiles of the grid is this barrier's, and
        //  is counted as the update's, where it was before round 6)
        if (finished_in_tiles) stamp(8);
        grid.sync();  // the new basis, D and (flip) row p for everybody
        stamp(finished_in_tiles ? 7 : 8);
#ifdef RELP_STAMPS
        if (leader) {  // (the pivot's longest workgroup: see update_on_matrix_cores)
            lp.prof[34] += lp.prof[35] / 1024ull;
            if ((int)(lp.prof[35] % 1024ull) >= (int)gridDim.x - 9) lp.prof[39] += 1;
            lp.prof[35] = 0;
            lp.prof[37] += lp.prof[38];
            lp.prof[38] = 0;
        }
#endif
    }
    grid.sync();
    // the final x~_B belongs to the final basis: recompute it (the loop computes it at the top of an iteration)
    for (int i = gtid; i < m; i += GT) {
        Big<L> acc = big_from<L>(0);
        for (int k = 0; k < m; ++k) {
            const i64 b = lp.rhs[k];
            if (b != 0) acc = big_add(acc, big_mul_small(big_load_s<L>(N_at(i, k), MM), b));
        }
        big_store(xt_of_host + (size_t)i * L, acc);
    }
    {   // the word products of the run: one sum per workgroup, one atomic per workgroup
        __shared__ unsigned long long s_products[2];
        if (tid == 0) s_products[0] = s_products[1] = 0;
        __syncthreads();
        for (int d = WAVE / 2; d > 0; d /= 2) {
            products_needed += __shfl_xor(products_needed, d);
            products_issued += __shfl_xor(products_issued, d);
        }
        if ((tid & (WAVE - 1)) == 0) {
            atomicAdd(&s_products[0], products_needed);
            atomicAdd(&s_products[1], products_issued);
        }
        __syncthreads();
        if (tid == 0) {
            atomicAdd(&lp.prof[16], s_products[0]);
            atomicAdd(&lp.prof[17], s_products[1]);
        }
    }
    if (leader) {
        lp.out[0] = status;
        lp.out[1] = (int)pivots[0];
        lp.out[2] = (int)pivots[1];
        lp.out[3] = L;
        lp.out[4] = trace_count < lp.trace_capacity ? trace_count : lp.trace_capacity;
        lp.out[5] = n_removed;
        lp.out[6] = at_phase;
        lp.out[7] = trace_count;
        lp.out[8] = at_drive_row;
        lp.out[9] = at_removed;
        lp.out[10] = n_swaps & 1;  // which buffer holds N now (the fused update swaps N and N_alt pivot by pivot)
        lp.prof[32] = (unsigned long long)n_swaps;  // (diagnostic: pivots whose tiles finished their entries themselves)
    }
}

// Sign extension of `count` integers from `from` to `to` words each (a run that overflowed continues at the next width).
__global__ void __launch_bounds__(256) exact_widen_kernel(const u64* src, u64* dst, long long count, int from, int to) {
    const long long total = count * to;  // (word-major on both sides: word k of integer v at [k * count + v])
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(idx / count);
        const long long v = idx - (long long)k * count;
        dst[idx] = k < from ? src[idx] : (u64)((i64)src[(long long)(from - 1) * count + v] >> 63);
    }
}

template <class T>
T* dalloc(size_t count, std::vector<void*>& owned) {
    void* p = nullptr;
    RELP_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
    owned.push_back(p);
    return reinterpret_cast<T*>(p);
}

BigInt big_from_words(const u64* w, int limbs) {  // two's complement words -> sign-magnitude
    const bool neg = (i64)w[limbs - 1] < 0;
    std::vector<u64> mag(w, w + limbs);
    if (neg) {
        u64 carry = 1;
        for (int k = 0; k < limbs; ++k) {
            const u128 s = (u128)(~mag[k]) + carry;
            mag[k] = (u64)s;
            carry = (u64)(s >> 64);
        }
    }
    BigInt r;
    for (int k = 0; k < limbs; ++k) {
        r.mag.push_back((uint32_t)mag[k]);
        r.mag.push_back((uint32_t)(mag[k] >> 32));
    }
    r.neg = neg;
    r.trim();
    return r;
}

// finish_update_entry on `count` entries given as they would leave the tiles (tests: carries, shifts of 64 bits and more, negation, bit lengths)
template <int L>
__global__ void __launch_bounds__(256) exact_finish_test_kernel(ExactLP lp, int count, const int* words, int shift, int flip, int* bits) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < count) bits[e] = finish_update_entry<L>(lp.T + e, lp.T_carry + e, (size_t)count, lp.N + e, (size_t)count, words[e], shift, flip != 0);
}
}  // namespace

// Host driver: scales the LP to integers, runs the kernel with 2, 4, ... limbs until it does not overflow.
//   status: 1 optimal | 2 infeasible | 3 unbounded | 4 overflow at the largest width | 5 pivot limit   (6 is no longer produced)
void exact_simplex(const StandardForm& form, int device, hipStream_t stream, int first_limbs, int max_limbs, long long max_pivots,
                   int trace_capacity, int* status, int* limbs_used, long long* pivots_phase_one, long long* pivots_phase_two,
                   std::vector<int>* trace, std::string* objective, std::vector<int>* final_basis,
                   std::vector<std::pair<int, long long>>* pivots_survived, int* redundant_rows, std::vector<ExactWidthRecord>* counters, int update_mode, int forced_grid) {
    RELP_HIP(hipSetDevice(device));
    if (counters) counters->clear();
    if (redundant_rows) *redundant_rows = 0;
    const MatrixData& md = form.data;
    const int m = md.nr_rows(), n_p = md.nr_columns();
    std::vector<SparseColumn> columns(n_p);
    for (int j = 0; j < n_p; ++j) columns[j] = md.column(j);
    std::vector<Rat> rhs = md.right_hand_side();
    // row multipliers: lcm of the denominators of the row's coefficients and of its right-hand side
    std::vector<i128> row_mult(m, 1);
    auto lcm = [](i128 a, i128 b) { return mul_checked(a / gcd128(a, b), b); };
    for (int j = 0; j < n_p; ++j)
        for (size_t e = 0; e < columns[j].nnz(); ++e) row_mult[columns[j].index[e]] = lcm(row_mult[columns[j].index[e]], columns[j].value[e].d);
    for (int i = 0; i < m; ++i) row_mult[i] = lcm(row_mult[i], rhs[i].d);
    i128 cost_mult = 1;
    for (int j = 0; j < n_p; ++j) cost_mult = lcm(cost_mult, md.cost_value(j).d);
    auto small = [](i128 v) {
        if (v >= ((i128)1 << 62) || v <= -((i128)1 << 62)) throw RatOverflow();
        return (i64)v;
    };
    // index space of the device loop: artificials first (one per row without an initial pivot), then the provider columns
    auto pivots = md.pivot_element_indices();
    std::vector<int> real_column_of_row(m, -1);
    for (auto& [row, column] : pivots) real_column_of_row[row] = column;
    std::vector<int> artificial_rows;
    for (int i = 0; i < m; ++i)
        if (real_column_of_row[i] < 0) artificial_rows.push_back(i);
    const int n_art = (int)artificial_rows.size(), n = n_art + n_p;
    // lcm of the row multipliers (W = lcm^2 weights the squared norms) and of those of the artificial rows (phase-one costs)
    i128 lcm_all = 1, lcm_art = 1;
    for (int i = 0; i < m; ++i) lcm_all = lcm(lcm_all, row_mult[i]);
    for (int i : artificial_rows) lcm_art = lcm(lcm_art, row_mult[i]);
    std::vector<int> col_start(n + 1, 0), row_index;
    std::vector<i64> value, cost2(n, 0), cost1(n, 0), rhs_scaled(m);
    std::vector<u64> weight(2 * (size_t)n, 0);  // two words per column, low first
    auto set_weight = [&](int j, i128 ratio) {  // w_j = ratio^2: below 2^124 (the sums of m + 1 weighted squares keep their 2 L + 2 words)
        if (ratio < 0 || ratio >= ((i128)1 << 54)) throw RatOverflow();  // (w < 2^108: with values below 2^(64 L - 3) and m + 1 < 2^17 terms the sums stay below 2^(128 L + 128))
        const u128 w = (u128)ratio * (u128)ratio;
        weight[2 * (size_t)j] = (u64)w;
        weight[2 * (size_t)j + 1] = (u64)(w >> 64);
    };
    for (int k = 0; k < n_art; ++k) {  // artificial of row i: the unit column, scaled by sigma = 1 / r_i => entry 1
        const int i = artificial_rows[k];
        row_index.push_back(i);
        value.push_back(1);
        col_start[k + 1] = (int)row_index.size();
        cost1[k] = small(lcm_art / row_mult[i]);
        set_weight(k, lcm_all / row_mult[i]);
    }
    for (int j = 0; j < n_p; ++j) {
        const Rat c = md.cost_value(j);
        // a column with a single entry +-1 and no cost (a slack) is scaled by sigma = 1 / r_i like the artificial columns
        const bool unit = columns[j].nnz() == 1 && columns[j].value[0].d == 1 && (columns[j].value[0].n == 1 || columns[j].value[0].n == -1) && c.is_zero();
        for (size_t e = 0; e < columns[j].nnz(); ++e) {
            const int i = columns[j].index[e];
            row_index.push_back(i);
            value.push_back(unit ? (i64)columns[j].value[e].n : small(mul_checked(columns[j].value[e].n, row_mult[i] / columns[j].value[e].d)));
        }
        col_start[n_art + j + 1] = (int)row_index.size();
        cost2[n_art + j] = small(mul_checked(c.n, cost_mult / c.d));
        const i128 ratio = unit ? lcm_all / row_mult[columns[j].index[0]] : lcm_all;
        set_weight(n_art + j, ratio);
    }
    for (int i = 0; i < m; ++i) rhs_scaled[i] = small(mul_checked(rhs[i].n, row_mult[i] / rhs[i].d));
    std::vector<int> basis0(m), pos0(n, -1);
    {
        int k = 0;
        for (int i = 0; i < m; ++i) {
            basis0[i] = real_column_of_row[i] < 0 ? k++ : n_art + real_column_of_row[i];
            pos0[basis0[i]] = i;
        }
    }
    // B_0 = diag(d_i) with d_i the unit entry of row i's initial basic column (1 after the scaling above): D_0 = prod d_i
    std::vector<i64> diag0(m);
    for (int i = 0; i < m; ++i) {
        const int c = basis0[i];
        if (col_start[c + 1] - col_start[c] != 1 || row_index[col_start[c]] != i || value[col_start[c]] <= 0)
            throw std::runtime_error("exact simplex: the initial basis is not a positive diagonal");
        diag0[i] = value[col_start[c]];
    }
    BigInt D0(1);
    for (int i = 0; i < m; ++i) D0 = D0 * BigInt(diag0[i]);

    std::vector<void*> owned;
    struct Free {
        std::vector<void*>& p;
        ~Free() { for (void* q : p) (void)hipFree(q); }
    } free_all{owned};
    int* d_col_start = dalloc<int>(n + 1, owned);
    int* d_row_index = dalloc<int>(row_index.size(), owned);
    i64* d_value = dalloc<i64>(value.size(), owned);
    i64* d_cost2 = dalloc<i64>(n, owned);
    i64* d_cost1 = dalloc<i64>(n, owned);
    u64* d_weight = dalloc<u64>(2 * (size_t)n, owned);
    i64* d_rhs = dalloc<i64>(m, owned);
    int* d_basis = dalloc<int>(m, owned);
    int* d_pos = dalloc<int>(n, owned);
    double* d_key = dalloc<double>(n, owned);
    int* d_trace = dalloc<int>((size_t)4 * trace_capacity, owned);
    int* d_out = dalloc<int>(16, owned);
    int* d_resume = dalloc<int>(8, owned);
    int* d_removed = dalloc<int>(2 * (size_t)m, owned);  // [m] the flags, [m] how many removed rows lie above each row
    int* d_words = dalloc<int>(16, owned);
    constexpr int EX_MAX_GRID = 1024;
    double* d_part_key = dalloc<double>(2 * EX_MAX_GRID, owned);
    unsigned long long* d_part_rank = dalloc<unsigned long long>(2 * EX_MAX_GRID, owned);
    unsigned long long* d_prof = dalloc<unsigned long long>(EX_PROF_WORDS, owned);
    unsigned* d_barrier = dalloc<unsigned>(EX_BARRIER_WORDS, owned);
    const bool print_profile = diagnostic("RELP_EXACT_PROFILE");
    const size_t pairs = (size_t)std::max(1, n - n_art) * m;
    u64* d_price_a = nullptr;   // (sized per limb count below)
    double* d_price_err = dalloc<double>(pairs, owned);
    double* d_price_term = dalloc<double>(pairs, owned);
    int* d_bracket = dalloc<int>(std::max(n, m) + 1, owned);
    int* d_cand = dalloc<int>(std::max(n, m) + 1, owned);
    int* d_N_bits = dalloc<int>((size_t)m * m, owned);
    int* d_y_bits = dalloc<int>((size_t)m, owned);
    double* d_cd = dalloc<double>((size_t)n, owned);
    int* d_neg_list = dalloc<int>((size_t)n + 1, owned);
    i64* d_cb_row = dalloc<i64>(m, owned);
    int* d_row_list = dalloc<int>((size_t)2 * m, owned);
    int* d_col_heavy = dalloc<int>((size_t)m + 1, owned);
    int* d_col_light = dalloc<int>((size_t)m + 1, owned);
#ifdef RELP_REPRO_R5_LIST_RACE
    // (diagnostic build, tools/repro_16_limb_hang.sh: the column classes of the update in the tournaments' arrays again, as they were when the
    //  16-limb matrix-core run hung on ISRAEL in round 5 -- the splitter workgroup overwrites bracket[0] / cand[..] while a workgroup late
    //  out of the ratio test's last barrier still reads the winner through them)
    d_col_heavy = d_bracket;
    d_col_light = d_cand;
#endif
    RELP_HIP(hipMemcpyAsync(d_col_start, col_start.data(), (n + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemcpyAsync(d_row_index, row_index.data(), row_index.size() * sizeof(int), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemcpyAsync(d_value, value.data(), value.size() * sizeof(i64), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemcpyAsync(d_cost2, cost2.data(), n * sizeof(i64), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemcpyAsync(d_cost1, cost1.data(), n * sizeof(i64), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemcpyAsync(d_weight, weight.data(), weight.size() * sizeof(u64), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemcpyAsync(d_rhs, rhs_scaled.data(), m * sizeof(i64), hipMemcpyHostToDevice, stream));

    *status = EX_OVERFLOW;
    *limbs_used = 0;
    if (pivots_survived) pivots_survived->clear();
    // A width that overflows hands its state to the next one: the kernel decides whether a pivot fits BEFORE it writes anything, so
    // N, D, the basis and the bookkeeping are those at the start of that pivot; they are sign-extended to twice the words and the
    // loop continues there.  (Rounds 2 and 3a started every width from the initial basis: SCORPION made 45 + 74 + 91 + 142 + 253 +
    // 366 pivots for its 366.)  The arithmetic is exact, so the pivot sequence does not depend on where the widths change.
    u64* previous_N = nullptr;
    u64* previous_D = nullptr;
    int previous_limbs = 0;
    int resume_state[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RELP_HIP(hipMemcpyAsync(d_basis, basis0.data(), m * sizeof(int), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemcpyAsync(d_pos, pos0.data(), n * sizeof(int), hipMemcpyHostToDevice, stream));
    RELP_HIP(hipMemsetAsync(d_removed, 0, 2 * (size_t)m * sizeof(int), stream));
    std::vector<void*> width_owned;  // the big-integer buffers of the current width (the previous width's are freed once widened)
    struct FreeWidth {
        std::vector<void*>& p;
        ~FreeWidth() { for (void* q : p) (void)hipFree(q); }
    } free_width{width_owned};
    for (int limbs = std::max(1, first_limbs); limbs <= max_limbs; limbs *= 2) {
        const size_t big = (size_t)limbs;
        std::vector<void*> fresh;
        u64* d_N = dalloc<u64>((size_t)m * m * big, fresh);
        u64* d_D = dalloc<u64>(2 * big + 2 * (2 * big + 2) + 8, fresh);
        u64* d_xt = dalloc<u64>((size_t)m * big, fresh);
        u64* d_alpha = dalloc<u64>((size_t)m * big, fresh);
        u64* d_ctil = dalloc<u64>((size_t)n * big, fresh);
        d_price_a = dalloc<u64>((size_t)EX_PRODUCT_SLOTS * m * big, fresh);  // (also the chunks of the entering column: (big / 16) 19 m words)
        u64* d_gamma = dalloc<u64>((size_t)n * (2 * big + 2), fresh);  // (round 3 sized it for 32 limbs: a wider run with many tied candidates wrote past it)
        u64* d_gamma_terms = dalloc<u64>((size_t)EX_GAMMA_BATCH * (m + 1) * (2 * big + 2), fresh);
        u64* d_x_part = dalloc<u64>((size_t)m * ((m + 31) / 32) * big, fresh);
        int* d_x_bits = dalloc<int>((size_t)m * ((m + 31) / 32), fresh);
        // the update of N on the matrix cores (mfma_update_tile): from 16 limbs on (update_mode bit 0: never).  Round 6: the tiles finish
        // their entries themselves into a second buffer of N (update_mode bit 2: the two passes of round 5 -- numerators to T, a thread per
        // entry behind a barrier -- which remain for the pivots the fused epilogue does not take: a negative pivot element, zero-level pivots only).
        const bool mfma_update = limbs >= 16 && (update_mode & 1) == 0;
        const bool fused_update = mfma_update && (update_mode & 4) == 0;
        u64* d_N_alt = fused_update ? dalloc<u64>((size_t)m * m * big, fresh) : nullptr;
        int* d_N_bits_alt = fused_update ? dalloc<int>((size_t)m * m, fresh) : nullptr;
        u64* d_xt_alt = fused_update ? dalloc<u64>((size_t)m * big, fresh) : nullptr;
        int* d_xt_bits_alt = fused_update ? dalloc<int>((size_t)m, fresh) : nullptr;
        u64* d_y_alt = fused_update ? dalloc<u64>((size_t)m * big, fresh) : nullptr;
        int* d_y_bits_alt = fused_update ? dalloc<int>((size_t)m, fresh) : nullptr;
        if (fused_update) {  // (every entry of a buffer is a whole sign-extended integer whose bit length its array holds: zeros to begin with)
            RELP_HIP(hipMemsetAsync(d_N_alt, 0, (size_t)m * m * big * sizeof(u64), stream));
            RELP_HIP(hipMemsetAsync(d_N_bits_alt, 0, (size_t)m * m * sizeof(int), stream));
            RELP_HIP(hipMemsetAsync(d_xt_alt, 0, (size_t)m * big * sizeof(u64), stream));
            RELP_HIP(hipMemsetAsync(d_xt_bits_alt, 0, (size_t)m * sizeof(int), stream));
            RELP_HIP(hipMemsetAsync(d_y_alt, 0, (size_t)m * big * sizeof(u64), stream));
            RELP_HIP(hipMemsetAsync(d_y_bits_alt, 0, (size_t)m * sizeof(int), stream));
        }
        u64* d_T = mfma_update ? dalloc<u64>((size_t)m * m * big, fresh) : nullptr;
        int* d_T_carry = mfma_update ? dalloc<int>((size_t)m * m * (big / 2), fresh) : nullptr;
        int* d_T_words = mfma_update ? dalloc<int>((size_t)m * m, fresh) : nullptr;
        u64* d_y = dalloc<u64>((size_t)m * big, fresh);
        u64* d_Tx = mfma_update ? dalloc<u64>(2 * (size_t)m * big, fresh) : nullptr;
        int* d_Tx_carry = mfma_update ? dalloc<int>(2 * (size_t)m * (big / 2), fresh) : nullptr;
        int* d_Tx_words = mfma_update ? dalloc<int>(2 * (size_t)m, fresh) : nullptr;
        u64* d_y_part = dalloc<u64>(big, fresh);
        u64* d_next_dinv = dalloc<u64>(big, fresh);
        int* d_xt_bits = dalloc<int>((size_t)m, fresh);
        if (mfma_update) {
            RELP_HIP(hipMemsetAsync(d_T_words, 0, (size_t)m * m * sizeof(int), stream));
            RELP_HIP(hipMemsetAsync(d_Tx_words, 0, 2 * (size_t)m * sizeof(int), stream));
        }
        auto adopt = [&]() {  // the new width's buffers replace the previous width's
            RELP_HIP(hipStreamSynchronize(stream));
            for (void* q : width_owned) (void)hipFree(q);
            width_owned = fresh;
        };
        // N_0 and D_0 as two's complement words (positive values)
        auto words = [&](const BigInt& v, u64* out) {
            for (int k = 0; k < limbs; ++k) {
                const uint64_t lo = 2 * k < (int)v.mag.size() ? v.mag[2 * k] : 0, hi = 2 * k + 1 < (int)v.mag.size() ? v.mag[2 * k + 1] : 0;
                out[k] = lo | (hi << 32);
            }
        };
        if (!previous_N && D0.bits() + 3 > (size_t)64 * limbs) {  // does not even hold the first determinant
            if (pivots_survived) pivots_survived->push_back({limbs, 0});
            for (void* q : fresh) (void)hipFree(q);
            continue;
        }
        if (previous_N) {
            hipLaunchKernelGGL(exact_widen_kernel, dim3(1024), dim3(256), 0, stream, previous_N, d_N, (long long)m * m, previous_limbs, limbs);
            hipLaunchKernelGGL(exact_widen_kernel, dim3(1), dim3(64), 0, stream, previous_D, d_D, 1LL, previous_limbs, limbs);
            resume_state[0] = 1;
        } else {
            std::vector<u64> hN((size_t)m * m * big, 0), hD(big);
            words(D0, hD.data());
            for (int i = 0; i < m; ++i) {
                BigInt q, r;
                BigInt::divmod(D0, BigInt(diag0[i]), q, r);
                std::vector<u64> w(big);
                words(q, w.data());
                for (int k = 0; k < limbs; ++k) hN[(size_t)k * m * m + (size_t)i * m + i] = w[k];  // word-major (big_load_s)
            }
            RELP_HIP(hipMemcpyAsync(d_N, hN.data(), hN.size() * sizeof(u64), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(d_D, hD.data(), big * sizeof(u64), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipStreamSynchronize(stream));  // (hN and hD leave scope)
        }
        adopt();
        RELP_HIP(hipMemcpyAsync(d_resume, resume_state, sizeof(resume_state), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemsetAsync(d_words, 0, 16 * sizeof(int), stream));
        RELP_HIP(hipMemsetAsync(d_prof, 0, EX_PROF_WORDS * sizeof(unsigned long long), stream));
        RELP_HIP(hipMemsetAsync(d_barrier, 0, EX_BARRIER_WORDS * sizeof(unsigned), stream));
        const auto width_start = std::chrono::steady_clock::now();
        ExactLP lp{m, n, n_art, limbs, d_col_start, d_row_index, d_value, d_cost2, d_cost1, d_weight, d_rhs, d_basis, d_pos, d_N, d_D, d_xt, d_alpha,
                   d_ctil, d_key, d_trace, trace_capacity, max_pivots, d_out, d_resume, d_removed, d_words, d_part_key, d_part_rank, d_prof, d_price_a, d_price_err, d_price_term, d_bracket, d_cand, d_gamma, d_gamma_terms, d_x_part, d_x_bits, d_cb_row, d_row_list, d_col_heavy, d_col_light, d_N_bits,
                   d_T, d_T_carry, d_T_words, d_y, d_y_bits, d_cd, d_neg_list, d_Tx, d_Tx_carry, d_Tx_words, d_y_part, d_next_dinv, d_xt_bits, mfma_update ? 1 : 0, d_N_alt, d_N_bits_alt, fused_update ? 1 : 0, d_xt_alt, d_xt_bits_alt, d_y_alt, d_y_bits_alt, (update_mode & 2) ? 1 : 0, d_barrier};
        // The grid by the work of a pivot (m^2 entries of `limbs`^2 word products each, and as much again for pricing): one workgroup
        // for the smallest LPs -- a grid barrier costs 2 us at 8 workgroups, 25 at 256 -- up to one per CU.  relp_options.exact_grid: A/B hook.
        int grid = (int)std::min<long long>(256, std::max<long long>(1, (long long)m * m * limbs / 4096));
        // ... and two per CU where a pivot is milliseconds of arithmetic (25FV47 from 64 limbs on: 50 -> 39 s; at one wave per SIMD the
        // passes wait on memory and on scratch) -- not below: the barriers of 512 workgroups cost SCORPION and E226 a tenth of a second.
        if ((double)m * m * limbs * limbs >= 1e9) grid = 512;
        // ... and at most one per two CUs for the mid-size LPs, whose pivot is a fraction of a millisecond since the round-4 rework: the
        // barriers of 256 workgroups (25 us each, fifteen a pivot) were a third of it (E226 0.38 -> 0.33 s, BRANDY 0.52 -> 0.41 s).
        else if ((double)m * m * limbs < 8e6) grid = std::min(grid, 128);
        if (forced_grid > 0) grid = std::max(1, std::min(EX_MAX_GRID, forced_grid));
        void* kernel = nullptr;
        switch (limbs) {
            case 1: kernel = (void*)exact_simplex_kernel<1>; break;
            case 2: kernel = (void*)exact_simplex_kernel<2>; break;
            case 4: kernel = (void*)exact_simplex_kernel<4>; break;
            case 8: kernel = (void*)exact_simplex_kernel<8>; break;
            case 16: kernel = (void*)exact_simplex_kernel<16>; break;
            case 32: kernel = (void*)exact_simplex_kernel<32>; break;
            case 64: kernel = (void*)exact_simplex_kernel<64>; break;
            case 128: kernel = (void*)exact_simplex_kernel<128>; break;
            default: throw std::invalid_argument("limbs must be a power of two between 1 and 128");
        }
        {
            int per_cu = 0, cus = 0;
            RELP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, EX_THREADS, 0));
            RELP_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
            grid = std::max(1, std::min(grid, per_cu * cus));  // (a cooperative launch needs every workgroup resident)
        }
        void* args[] = {(void*)&lp};
        RELP_HIP(hipLaunchCooperativeKernel(kernel, dim3(grid), dim3(EX_THREADS), args, 0, stream));
        RELP_HIP(hipGetLastError());
        int out[16];
        RELP_HIP(hipMemcpyAsync(out, d_out, sizeof(out), hipMemcpyDeviceToHost, stream));
        RELP_HIP(hipStreamSynchronize(stream));
        {   // the barrier's watchdog (grid_barrier.hpp): a launch whose workgroups made different numbers of barriers ends itself
            std::vector<unsigned> tail(EX_BARRIER_WORDS - EX_BARRIER_ABORT);
            RELP_HIP(hipMemcpy(tail.data(), d_barrier + EX_BARRIER_ABORT, tail.size() * sizeof(unsigned), hipMemcpyDeviceToHost));
            if (tail[0] != 0) {
                unsigned lowest = ~0u, highest = 0;
                int waiting = 0;
                for (int g = 0; g < grid; ++g) {
                    const unsigned at = tail[EX_BARRIER_STUCK - EX_BARRIER_ABORT + g];
                    if (at == 0) continue;
                    ++waiting;
                    lowest = std::min(lowest, at);
                    highest = std::max(highest, at);
                }
                throw std::runtime_error("exact simplex at " + std::to_string(limbs) + " limbs: the grid barrier's watchdog ended the launch (barrier " + std::to_string(tail[0]) + ", " +
                                         std::to_string(waiting) + " of " + std::to_string(grid) + " workgroups found waiting, at barriers " + std::to_string(lowest) + " to " +
                                         std::to_string(highest) + "): the workgroups' barrier counts differ");
            }
        }
        {
            unsigned long long prof[EX_PROF_WORDS];
            RELP_HIP(hipMemcpy(prof, d_prof, sizeof(prof), hipMemcpyDeviceToHost));
            static const char* names[] = {"x_B", "reduced costs (y = c_B' N)", "arg-max + candidates", "exact weights", "tournament", "alpha", "ratio test", "update", "bookkeeping", "products + keys of the columns with c~ < 0"};
            if (print_profile) {
                fprintf(stderr, "[exact] %d limbs, grid %d, %d pivots, candidates %llu, update word products %.3e needed / %.3e issued:", limbs, grid, out[1] + out[2],
                        prof[12], (double)prof[16], (double)prof[17]);
                for (int k = 0; k < 10; ++k) fprintf(stderr, " %s %.1f ms", names[k], prof[k] / 1e5);
                fprintf(stderr, " | ratio test: %llu pivots with near-tied rows (%llu rows in all)", prof[24], prof[25]);
                fprintf(stderr, " | columns with a negative reduced cost per pivot: %.1f", prof[31] ? (double)prof[30] / prof[31] : 0.0);
#ifdef RELP_TILE_STAMPS
                fprintf(stderr, " | one wave's tiles, M cycles: requests %.1f, steps %.1f, epilogue %.1f, tiles %llu", prof[26] / 1e6, prof[27] / 1e6, prof[28] / 1e6, prof[29]);
#endif
#ifndef RELP_TILE_STAMPS
                fprintf(stderr, " | pricing: first barrier %.1f ms, c~ pass %.1f, barrier %.1f, estimates %.1f, barrier %.1f", prof[26] / 1e5, prof[27] / 1e5, prof[28] / 1e5, prof[29] / 1e5, prof[15] / 1e5);
#endif
                fprintf(stderr, " | 1 / D_odd %.1f ms | entering column: chunks + barrier %.1f, adding up %.1f, barrier %.1f, row factors %.1f", prof[14] / 1e5, prof[10] / 1e5, prof[11] / 1e5,
                        prof[18] / 1e5, prof[19] / 1e5);
                fprintf(stderr, " | inside the update: both-term tiles %.1f ms, rescaled tiles %.1f, barrier %.1f, second pass %.1f (%llu pivots finished inside their tiles)\n", prof[20] / 1e5, prof[21] / 1e5,
                        prof[22] / 1e5, prof[23] / 1e5, prof[32]);
#ifdef RELP_PRICE_BLOCK_TIMES
                fprintf(stderr, "[exact] waves of the c~ pass by their time in it: < 15 us %llu, < 30 us %llu, < 60 us %llu, < 120 us %llu, longer %llu\n", prof[33], prof[34], prof[35], prof[36], prof[37]);
#endif
#ifdef RELP_STAMPS
                fprintf(stderr, "[exact] tiles of the update by workgroup: mean %.1f ms, the pivots' longest %.1f ms (sums over the pivots)\n", prof[33] / 1e5 / std::max(1, grid - 1), prof[34] / 1e5);
                fprintf(stderr, "[exact] ... the both-term tiles alone: mean %.1f ms, the pivots' longest %.1f ms; the longest workgroup was one of the last eight in %llu pivots\n",
                        prof[36] / 1e5 / std::max(1, grid - 1), prof[37] / 1e5, prof[39]);
#endif
            }
            if (counters) {
                ExactWidthRecord record;
                record.limbs = limbs;
                record.grid = grid;
                record.pivots_total_at_end = (long long)out[1] + out[2];
                record.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - width_start).count();
                for (int k = 0; k < 10; ++k) record.step_seconds[k] = prof[k] * 1e-8;  // ticks of 10 ns
                record.update_products_needed = (long long)prof[16];
                record.update_products_issued = (long long)prof[17];
                counters->push_back(record);
            }
        }
        *status = out[0];
        *limbs_used = limbs;
        *pivots_phase_one = out[1];
        *pivots_phase_two = out[2];
        if (pivots_survived) pivots_survived->push_back({limbs, (long long)out[1] + out[2]});
        if (out[0] == EX_OVERFLOW) {
            previous_N = out[10] ? d_N_alt : d_N;  // (the buffer that held N when the kernel stopped: the fused update swaps them pivot by pivot)
            previous_D = d_D;
            previous_limbs = limbs;
            resume_state[1] = out[6];
            resume_state[2] = out[1];
            resume_state[3] = out[2];
            resume_state[4] = out[7];
            resume_state[5] = out[8];
            resume_state[6] = out[9];
            continue;
        }
        if (redundant_rows) *redundant_rows = out[5];
        // results of the run that did not overflow
        if (trace) {
            trace->assign((size_t)4 * out[4], 0);
            if (out[4] > 0) RELP_HIP(hipMemcpy(trace->data(), d_trace, trace->size() * sizeof(int), hipMemcpyDeviceToHost));
        }
        std::vector<int> basis(m);
        RELP_HIP(hipMemcpy(basis.data(), d_basis, m * sizeof(int), hipMemcpyDeviceToHost));
        if (final_basis) {
            final_basis->resize(m);
            for (int i = 0; i < m; ++i) (*final_basis)[i] = basis[i] >= n_art ? basis[i] - n_art : -1 - basis[i];
        }
        if (objective && out[0] == EX_OPTIMAL) {
            // objective = sum_i c_{B_i} x~_i / (D cost_mult) + fixed cost   (general_form/mod.rs:840-851)
            std::vector<u64> hx((size_t)m * big), hd(big);
            RELP_HIP(hipMemcpy(hx.data(), d_xt, hx.size() * sizeof(u64), hipMemcpyDeviceToHost));
            RELP_HIP(hipMemcpy(hd.data(), d_D, big * sizeof(u64), hipMemcpyDeviceToHost));
            BigInt num(0);
            for (int i = 0; i < m; ++i)
                if (basis[i] >= n_art && cost2[basis[i]] != 0) num = num + BigInt(cost2[basis[i]]) * big_from_words(hx.data() + (size_t)i * big, limbs);
            BigInt den = big_from_words(hd.data(), limbs) * BigInt::from_i128(cost_mult);
            const Rat& fixed = form.fixed_cost;
            num = num * BigInt::from_i128(fixed.d) + BigInt::from_i128(fixed.n) * den;
            den = den * BigInt::from_i128(fixed.d);
            if (den.sign() < 0) { num = -num; den = -den; }
            BigInt g = BigInt::gcd(num, den);
            if (!g.is_zero() && !(g == BigInt(1))) {
                num = num / g;
                den = den / g;
            }
            *objective = num.to_string() + "/" + den.to_string();
        }
        return;
    }
}

// `finish_update_entry` by itself: count entries, word-major operands as lp.T / lp.T_carry hold them (entry stride = count)
// the word arithmetic of the pivot's scalars on `count` operand pairs, a workgroup each (tests): mode 0: 1 / a modulo 2^(64 L), a odd
// (wave_inverse_odd); 1: -(a b) modulo 2^(64 L) (wave_mul_lo_negated, the rows' factors); 2: a b (y's factor)
template <int L>
__global__ void __launch_bounds__(EX_THREADS) exact_words_test_kernel(int mode, const u64* a, const u64* b, u64* out) {
    __shared__ u64 s_a[L], s_b[L], s_x[L], s_t[L], s_x2[L];
    const int tid = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * L;
    if (tid < L) {
        s_a[tid] = a[base + tid];
        s_b[tid] = b[base + tid];
    }
    __syncthreads();
    if (mode == 0) {
        wave_inverse_odd<L>(s_a, s_x, s_t, s_x2);
        if (tid < L) out[base + tid] = s_x[tid];
    } else if (tid < WAVE) {
        if (mode == 1) wave_mul_lo_negated<L, true>(s_a, s_b, out + base, 1, tid);
        else wave_mul_lo_negated<L, false>(s_a, s_b, out + base, 1, tid);
    }
}
void exact_words_test(int device, int limbs, int mode, int count, const unsigned long long* a, const unsigned long long* b, unsigned long long* out) {
    RELP_HIP(hipSetDevice(device));
    std::vector<void*> owned;
    struct Free {
        std::vector<void*>& p;
        ~Free() { for (void* q : p) (void)hipFree(q); }
    } free_all{owned};
    const size_t bytes = (size_t)count * limbs * sizeof(u64);
    u64 *d_a = nullptr, *d_b = nullptr, *d_out = nullptr;
    RELP_HIP(hipMalloc((void**)&d_a, bytes)); owned.push_back(d_a);
    RELP_HIP(hipMalloc((void**)&d_b, bytes)); owned.push_back(d_b);
    RELP_HIP(hipMalloc((void**)&d_out, bytes)); owned.push_back(d_out);
    RELP_HIP(hipMemcpy(d_a, a, bytes, hipMemcpyHostToDevice));
    RELP_HIP(hipMemcpy(d_b, b, bytes, hipMemcpyHostToDevice));
    const dim3 grid(count), block(EX_THREADS);
    switch (limbs) {
        case 16: hipLaunchKernelGGL(exact_words_test_kernel<16>, grid, block, 0, 0, mode, d_a, d_b, d_out); break;
        case 32: hipLaunchKernelGGL(exact_words_test_kernel<32>, grid, block, 0, 0, mode, d_a, d_b, d_out); break;
        case 64: hipLaunchKernelGGL(exact_words_test_kernel<64>, grid, block, 0, 0, mode, d_a, d_b, d_out); break;
        case 128: hipLaunchKernelGGL(exact_words_test_kernel<128>, grid, block, 0, 0, mode, d_a, d_b, d_out); break;
        default: throw std::invalid_argument("limbs must be 16, 32, 64 or 128");
    }
    RELP_HIP(hipGetLastError());
    RELP_HIP(hipDeviceSynchronize());
    RELP_HIP(hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost));
}
// The update tile by itself (tools/tile_bench.py): every wave of a full grid runs `tiles` fused tiles of `nb64` blocks and `terms` terms on
// synthetic operands; returns the seconds of the launch.  With -DRELP_TILE_VARIANT the same with one resource taken out.
template <int L>
__global__ void __launch_bounds__(EX_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) exact_tile_bench_kernel(const u64* N, u64* N_alt, int* bits_alt, const u64* x_part, int m,
                                                                                                               int tiles, int nb64, int terms, int shift, const u64* operand) {
    UpdateLds<L>& s_update = *reinterpret_cast<UpdateLds<L>*>(exact_arena<L>());
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const size_t MM = (size_t)m * m;
    for (int w = tid; w < L; w += blockDim.x) s_update.words[w] = operand[w];
    __syncthreads();
    build_toeplitz<L>(s_update, 0, s_update.words);
    __syncthreads();
    build_toeplitz<L>(s_update, 1, s_update.words);
    __syncthreads();
    const lds_u32* toeplitz_lds = (const lds_u32*)&s_update.toeplitz[0][0][0];
    const lds_i32* prefix_lds = (const lds_i32*)&s_update.prefix[0][0];
    const UpdateTileArgs tile_args{x_part, m, nullptr};
    int issued = 0;
    for (int t = 0; t < tiles; ++t) {
        const size_t idx = (((size_t)blockIdx.x * (EX_THREADS / WAVE) + wave) * tiles + t) * 16 + (lane & 15);
        const int row = (int)(idx % m);
        const UpdateTileEntry at_entry{N + idx, MM, x_part + row, (size_t)m, N_alt + idx, nullptr, bits_alt + idx, MM};
        issued += mfma_update_tile<L, true>(tile_args, at_entry, toeplitz_lds, prefix_lds, row, idx < MM, terms, nb64, lane, shift);
    }
    if (issued == -1) bits_alt[0] = 0;
}
double exact_tile_bench(int device, int limbs, int tiles, int nb64, int terms, int shift) {
    RELP_HIP(hipSetDevice(device));
    if (limbs != 128 && limbs != 64 && limbs != 32 && limbs != 16) throw std::invalid_argument("limbs must be 16, 32, 64 or 128");
    const int grid = 512, waves = EX_THREADS / WAVE;
    int m = 64;
    while ((size_t)m * m < (size_t)grid * waves * tiles * 16) m += 64;
    const size_t MM = (size_t)m * m;
    std::vector<void*> owned;
    struct Free {
        std::vector<void*>& p;
        ~Free() { for (void* q : p) (void)hipFree(q); }
    } free_all{owned};
    auto bytes = [&](size_t n) {
        void* p = nullptr;
        RELP_HIP(hipMalloc(&p, n));
        owned.push_back(p);
        return p;
    };
    u64* d_N = (u64*)bytes(MM * limbs * sizeof(u64));
    u64* d_alt = (u64*)bytes(MM * limbs * sizeof(u64));
    int* d_bits = (int*)bytes(MM * sizeof(int));
    u64* d_x = (u64*)bytes((size_t)m * limbs * sizeof(u64));
    u64* d_op = (u64*)bytes(limbs * sizeof(u64));
    RELP_HIP(hipMemset(d_N, 0x5a, MM * limbs * sizeof(u64)));
    RELP_HIP(hipMemset(d_alt, 0, MM * limbs * sizeof(u64)));
    RELP_HIP(hipMemset(d_bits, 0, MM * sizeof(int)));
    RELP_HIP(hipMemset(d_x, 0x3c, (size_t)m * limbs * sizeof(u64)));
    RELP_HIP(hipMemset(d_op, 0x71, limbs * sizeof(u64)));
    hipEvent_t start, stop;
    RELP_HIP(hipEventCreate(&start));
    RELP_HIP(hipEventCreate(&stop));
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {  // (the second launch is the one timed)
        RELP_HIP(hipEventRecord(start, 0));
        switch (limbs) {
            case 16: hipLaunchKernelGGL(exact_tile_bench_kernel<16>, dim3(grid), dim3(EX_THREADS), 0, 0, d_N, d_alt, d_bits, d_x, m, tiles, nb64, terms, shift, d_op); break;
            case 32: hipLaunchKernelGGL(exact_tile_bench_kernel<32>, dim3(grid), dim3(EX_THREADS), 0, 0, d_N, d_alt, d_bits, d_x, m, tiles, nb64, terms, shift, d_op); break;
            case 64: hipLaunchKernelGGL(exact_tile_bench_kernel<64>, dim3(grid), dim3(EX_THREADS), 0, 0, d_N, d_alt, d_bits, d_x, m, tiles, nb64, terms, shift, d_op); break;
            default: hipLaunchKernelGGL(exact_tile_bench_kernel<128>, dim3(grid), dim3(EX_THREADS), 0, 0, d_N, d_alt, d_bits, d_x, m, tiles, nb64, terms, shift, d_op); break;
        }
        RELP_HIP(hipEventRecord(stop, 0));
        RELP_HIP(hipEventSynchronize(stop));
        RELP_HIP(hipGetLastError());
        RELP_HIP(hipEventElapsedTime(&ms, start, stop));
    }
    (void)hipEventDestroy(start);
    (void)hipEventDestroy(stop);
    return ms * 1e-3;
}
void exact_finish_entries(int device, int limbs, int count, const unsigned long long* T, const int* carry, const int* words, int shift, int flip,
                          unsigned long long* N_out, int* bits_out) {
    RELP_HIP(hipSetDevice(device));
    std::vector<void*> owned;
    struct Free {
        std::vector<void*>& p;
        ~Free() { for (void* q : p) (void)hipFree(q); }
    } free_all{owned};
    auto dalloc_bytes = [&](size_t bytes) {
        void* p = nullptr;
        RELP_HIP(hipMalloc(&p, std::max<size_t>(bytes, 8)));
        owned.push_back(p);
        return p;
    };
    const size_t entries = (size_t)count;
    u64* d_T = (u64*)dalloc_bytes(entries * limbs * sizeof(u64));
    int* d_carry = (int*)dalloc_bytes(entries * (limbs / 2) * sizeof(int));
    int* d_words = (int*)dalloc_bytes(entries * sizeof(int));
    u64* d_N = (u64*)dalloc_bytes(entries * limbs * sizeof(u64));
    int* d_bits = (int*)dalloc_bytes(entries * sizeof(int));
    RELP_HIP(hipMemcpy(d_T, T, entries * limbs * sizeof(u64), hipMemcpyHostToDevice));
    RELP_HIP(hipMemcpy(d_carry, carry, entries * (limbs / 2) * sizeof(int), hipMemcpyHostToDevice));
    RELP_HIP(hipMemcpy(d_words, words, entries * sizeof(int), hipMemcpyHostToDevice));
    ExactLP lp{};
    lp.N = d_N;
    lp.T = d_T;
    lp.T_carry = d_carry;
    const dim3 grid((count + 255) / 256), block(256);
    switch (limbs) {
        case 16: hipLaunchKernelGGL(exact_finish_test_kernel<16>, grid, block, 0, 0, lp, count, d_words, shift, flip, d_bits); break;
        case 32: hipLaunchKernelGGL(exact_finish_test_kernel<32>, grid, block, 0, 0, lp, count, d_words, shift, flip, d_bits); break;
        case 64: hipLaunchKernelGGL(exact_finish_test_kernel<64>, grid, block, 0, 0, lp, count, d_words, shift, flip, d_bits); break;
        case 128: hipLaunchKernelGGL(exact_finish_test_kernel<128>, grid, block, 0, 0, lp, count, d_words, shift, flip, d_bits); break;
        default: throw std::invalid_argument("limbs must be 16, 32, 64 or 128");
    }
    RELP_HIP(hipGetLastError());
    RELP_HIP(hipDeviceSynchronize());
    RELP_HIP(hipMemcpy(N_out, d_N, entries * limbs * sizeof(u64), hipMemcpyDeviceToHost));
    RELP_HIP(hipMemcpy(bits_out, d_bits, entries * sizeof(int), hipMemcpyDeviceToHost));
}

}  // namespace relp
