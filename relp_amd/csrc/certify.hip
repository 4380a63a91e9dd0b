// Exact certificate for the basis the f64 device simplex ends on: bit-exact rational optimum.
//
// The optimal objective of an LP is a property of the LP, not of the pivot path, so the reference's
// `RationalBig` optimum (tests/netlib/mod.rs:62-70: Carry<RationalBig, LUDecomposition<_>>) can be reproduced by
// proving the final basis B optimal in exact arithmetic:
//     B x_B = b,  x_B >= 0            (primal feasibility; Carry::b, carry/mod.rs:46-66)
//     B' y  = c_B, c_j - a_j'y >= 0   (dual feasibility; Tableau::relative_cost, tableau/mod.rs:106-112)
//     objective = c_B' x_B + fixed_cost  (general_form/mod.rs:840-851)
// Fixed-width integer arithmetic on the device replaces arbitrary precision there (north_star): the two linear
// systems are solved by Dixon p-adic lifting -- one modular inverse C = B^-1 mod p (Gauss-Jordan over Z_p, p < 2^31,
// 64-bit products) and then, per p-adic digit, a modular mat-vec and an exact integer residual update carried in
// 128-bit accumulators.  Only the assembly of the digits (Horner), the rational reconstruction and the sign checks use
// host big integers (bigint.hpp).  Every reconstructed vector is VERIFIED by exact substitution before it is used.
#include <algorithm>
#include <numeric>

#include "bigint.hpp"
#include "solver.hpp"

namespace relp {

namespace {

using u32 = uint32_t;
using u64 = uint64_t;
using i64 = long long;

// ---------------------------------------------------------------------------------------------------
// device: Z_p kernels
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 mod_inverse(u32 a, u32 p) {  // extended Euclid, a in [1, p)
    i64 t = 0, nt = 1, r = p, nr = a;
    while (nr != 0) {
        i64 q = r / nr;
        i64 tmp = t - q * nt; t = nt; nt = tmp;
        tmp = r - q * nr; r = nr; nr = tmp;
    }
    if (t < 0) t += p;
    return (u32)t;
}

// Gauss-Jordan step k on the augmented matrix M = [B | I] (m x ld, ld = 2m): pivot search, row swap, row scale, and a
// copy of column k (so that the elimination kernel can overwrite it).  One workgroup.
__global__ void __launch_bounds__(256) gj_pivot_kernel(u32* M, int m, int ld, int k, u32 p, u32* colk, int* info) {
    __shared__ int s_row;
    __shared__ u32 s_inv;
    if (info[0]) return;
    if (threadIdx.x == 0) s_row = 0x7fffffff;
    __syncthreads();
    for (int r = k + threadIdx.x; r < m; r += blockDim.x)
        if (M[(size_t)r * ld + k] != 0) atomicMin(&s_row, r);
    __syncthreads();
    const int r = s_row;
    if (r == 0x7fffffff) {
        if (threadIdx.x == 0) info[0] = 1;  // singular modulo p
        return;
    }
    if (r != k) {
        for (int j = threadIdx.x; j < ld; j += blockDim.x) {
            const u32 a = M[(size_t)r * ld + j];
            M[(size_t)r * ld + j] = M[(size_t)k * ld + j];
            M[(size_t)k * ld + j] = a;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_inv = mod_inverse(M[(size_t)k * ld + k], p);
    __syncthreads();
    const u32 inv = s_inv;
    for (int j = threadIdx.x; j < ld; j += blockDim.x) M[(size_t)k * ld + j] = (u32)(((u64)M[(size_t)k * ld + j] * inv) % p);
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += blockDim.x) colk[i] = (i == k) ? 0u : M[(size_t)i * ld + k];
}

// rows i != k: M[i][:] -= M[i][k] * M[k][:]  (mod p)
__global__ void __launch_bounds__(256) gj_eliminate_kernel(u32* M, int m, int ld, int k, u32 p, const u32* colk, const int* info) {
    if (info[0]) return;
    const int i = blockIdx.y;
    const u32 f = colk[i];
    if (f == 0) return;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld) return;
    const u64 neg = p - f;
    M[(size_t)i * ld + j] = (u32)((M[(size_t)i * ld + j] + neg * M[(size_t)k * ld + j]) % p);
}

// [B | I] from the CSC of the (row-scaled, integer) basis columns.
__global__ void build_augmented_kernel(u32* M, int m, int ld, const int* col_start, const int* row_index, const i64* value, u32 p) {
    const int k = blockIdx.x;  // basis position = matrix column
    for (int i = threadIdx.x; i < m; i += blockDim.x) {
        M[(size_t)i * ld + k] = 0;
        M[(size_t)i * ld + m + k] = (i == k) ? 1u : 0u;
    }
    __syncthreads();
    for (int e = col_start[k] + threadIdx.x; e < col_start[k + 1]; e += blockDim.x) {
        i64 v = value[e] % (i64)p;
        if (v < 0) v += p;
        M[(size_t)row_index[e] * ld + k] = (u32)v;
    }
}

// Dixon digit: x = C (r mod p) mod p   (transpose = 0)   or   x = C' (r mod p) mod p   (transpose = 1),
// C = right half of M.  One wave per output entry for the row-wise product, one thread per entry for the transposed one.
__global__ void __launch_bounds__(256) dixon_digit_kernel(const u32* M, int m, int ld, u32 p, const i64* r, u32* x, int transpose) {
    extern __shared__ u32 s_r[];
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        i64 v = r[j] % (i64)p;
        if (v < 0) v += p;
        s_r[j] = (u32)v;
    }
    __syncthreads();
    const u32* C = M + m;
    if (!transpose) {
        const int lane = threadIdx.x & 63;
        const int i = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
        if (i >= m) return;
        u64 acc = 0;
        for (int j = lane; j < m; j += 64) acc += ((u64)C[(size_t)i * ld + j] * s_r[j]) % p;
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
        if (lane == 0) x[i] = (u32)(acc % p);
    } else {
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        if (i >= m) return;
        u64 acc = 0;
        for (int j = 0; j < m; ++j) acc += ((u64)C[(size_t)j * ld + i] * s_r[j]) % p;
        x[i] = (u32)(acc % p);
    }
}

// r <- (r - A x) / p exactly, A given by rows (CSR of B for B x = b; CSR of B' = CSC of B for B' y = c).
// 128-bit accumulation: |A_ij| < 2^63 and x_j < 2^31.
__global__ void __launch_bounds__(256) dixon_residual_kernel(int m, const int* row_start, const int* col_index, const i64* value,
                                                           const u32* x, i64* r, u32 p, int* info) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    __int128 acc = r[i];
    for (int e = row_start[i]; e < row_start[i + 1]; ++e) acc -= (__int128)value[e] * (i64)x[col_index[e]];
    // exact division of a 128-bit value by p < 2^31 with 64-bit operations (no 128-bit divide on the device)
    const bool negative = acc < 0;
    unsigned __int128 mag = negative ? (unsigned __int128)(-acc) : (unsigned __int128)acc;
    u64 rem = 0;
    unsigned __int128 quotient = 0;
#pragma unroll
    for (int part = 3; part >= 0; --part) {
        const u64 cur = (rem << 32) | (u64)(u32)(mag >> (32 * part));
        quotient = (quotient << 32) | (cur / p);
        rem = cur % p;
    }
    if (rem != 0) info[1] = 1;  // cannot happen when C is the inverse of B modulo p
    if (quotient > (unsigned __int128)0x3fffffffffffffffULL) info[2] = 1;  // overflow guard
    acc = negative ? -(__int128)quotient : (__int128)quotient;
    r[i] = (i64)acc;
}

struct DeviceBuffers {
    std::vector<void*> ptrs;
    template <class T>
    T* alloc(size_t count) {
        void* p = nullptr;
        RELP_HIP(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)));
        ptrs.push_back(p);
        return reinterpret_cast<T*>(p);
    }
    ~DeviceBuffers() {
        for (void* p : ptrs) (void)hipFree(p);
    }
};

// ---------------------------------------------------------------------------------------------------
// host: exact helpers
// ---------------------------------------------------------------------------------------------------
BigInt big_from_i128(i128 v) { return BigInt::from_i128(v); }

i128 lcm128(i128 a, i128 b) {
    i128 g = gcd128(a, b);
    return mul_checked(a / g, b);
}

// Rational reconstruction of a (mod M): n/d with |n|, d <= sqrt(M/2) (Wang's bound); returns false if none.
bool rational_reconstruct(const BigInt& a, const BigInt& M, BigInt& n, BigInt& d) {
    BigInt r0 = M, r1 = a % M;
    if (r1.sign() < 0) r1 = r1 + M;
    BigInt t0(0), t1(1);
    auto too_big = [&](const BigInt& r) { return cmp(r * r * BigInt(2), M) > 0; };
    while (too_big(r1)) {
        BigInt q, rem;
        BigInt::divmod(r0, r1, q, rem);
        BigInt t2 = t0 - q * t1;
        r0 = r1;
        r1 = rem;
        t0 = t1;
        t1 = t2;
    }
    if (t1.is_zero() || too_big(t1.abs())) return false;
    n = t1.sign() < 0 ? -r1 : r1;
    d = t1.abs();
    BigInt g = BigInt::gcd(n, d);
    if (!(g == BigInt(1))) {
        if (g.is_zero()) return false;
        n = n / g;
        d = d / g;
    }
    return true;
}

// bits [lo, lo + count) of |v|, count <= 60
u64 extract_bits(const BigInt& v, size_t lo, int count) {
    unsigned __int128 window = 0;
    const size_t word = lo / 32;
    for (int k = 0; k < 4; ++k)
        if (word + k < v.mag.size()) window |= (unsigned __int128)v.mag[word + k] << (32 * k);
    window >>= (lo % 32);
    return (u64)(window & (((unsigned __int128)1 << count) - 1));
}

struct IntegerBasis {          // row-scaled integer basis, both orientations
    int m = 0;
    std::vector<int> col_start, row_index;   // CSC (columns = basis positions)
    std::vector<i64> value;
    std::vector<int> row_start, col_index;   // CSR
    std::vector<i64> row_value;
};

struct ExactVector {           // numer[i] / denom
    std::vector<BigInt> numer;
    BigInt denom = BigInt(1);
};

// Solve  A z = rhs  (transpose = 0: A = B;  transpose = 1: A = B')  by Dixon lifting; the result is verified exactly.
bool dixon_solve(const IntegerBasis& B, const std::vector<i64>& rhs, int transpose, u32 p, u32* dM, int ld,
                 DeviceBuffers& buf, const int* d_row_start, const int* d_col_index, const i64* d_row_value,
                 hipStream_t stream, ExactVector* out, std::string* message) {
    const int m = B.m;
    i64* d_r = buf.alloc<i64>(m);
    int* d_info = buf.alloc<int>(4);
    RELP_HIP(hipMemsetAsync(d_info, 0, 4 * sizeof(int), stream));
    RELP_HIP(hipMemcpyAsync(d_r, rhs.data(), m * sizeof(i64), hipMemcpyHostToDevice, stream));
    bool all_zero = std::all_of(rhs.begin(), rhs.end(), [](i64 v) { return v == 0; });
    out->numer.assign(m, BigInt(0));
    out->denom = BigInt(1);
    if (all_zero) return true;

    std::vector<std::vector<u32>> digits;  // digits[step][i]
    int steps_done = 0;
    int target = 32;
    const int max_steps = 1 << 15;
    u32* d_digits = nullptr;
    int digits_capacity = 0;
    while (true) {
        if (target > digits_capacity) {
            u32* nd = buf.alloc<u32>((size_t)target * m);
            if (d_digits && steps_done > 0)
                RELP_HIP(hipMemcpyAsync(nd, d_digits, (size_t)steps_done * m * sizeof(u32), hipMemcpyDeviceToDevice, stream));
            d_digits = nd;
            digits_capacity = target;
        }
        for (int s = steps_done; s < target; ++s) {
            u32* xs = d_digits + (size_t)s * m;
            const int blocks = transpose ? (m + 255) / 256 : (m + 3) / 4;
            hipLaunchKernelGGL(dixon_digit_kernel, dim3(blocks), dim3(256), m * sizeof(u32), stream, dM, m, ld, p, d_r, xs, transpose);
            hipLaunchKernelGGL(dixon_residual_kernel, dim3((m + 255) / 256), dim3(256), 0, stream, m, d_row_start, d_col_index,
                               d_row_value, xs, d_r, p, d_info);
        }
        std::vector<u32> flat((size_t)(target - steps_done) * m);
        int info[4];
        RELP_HIP(hipMemcpyAsync(flat.data(), d_digits + (size_t)steps_done * m, flat.size() * sizeof(u32), hipMemcpyDeviceToHost, stream));
        RELP_HIP(hipMemcpyAsync(info, d_info, sizeof(info), hipMemcpyDeviceToHost, stream));
        RELP_HIP(hipStreamSynchronize(stream));
        if (info[1] || info[2]) {
            *message = info[2] ? "Dixon residual overflow (coefficients too large for the 128-bit path)" : "Dixon residual not divisible by p";
            return false;
        }
        for (int s = steps_done; s < target; ++s)
            digits.emplace_back(flat.begin() + (size_t)(s - steps_done) * m, flat.begin() + (size_t)(s - steps_done + 1) * m);
        steps_done = target;

        // ---- assemble, reconstruct with a common denominator, verify ---------------------------------
        BigInt modulus(1);
        for (int s = 0; s < steps_done; ++s) modulus.mul_add_small(p, 0);
        std::vector<BigInt> residue(m);
        for (int i = 0; i < m; ++i) {
            BigInt acc(0);
            for (int s = steps_done; s-- > 0;) acc.mul_add_small(p, digits[s][i]);
            acc.trim();
            residue[i] = acc;
        }
        bool ok = true;
        BigInt denom(1);
        std::vector<BigInt> numer(m);
        const BigInt half = modulus / BigInt(2);
        for (int i = 0; i < m && ok; ++i) {
            BigInt t = (residue[i] * denom) % modulus;
            if (cmp(t, half) > 0) t = t - modulus;
            // accept t as the numerator when it is "small": |t| * 2^(32) < modulus / denom-size proxy; otherwise reconstruct
            BigInt n, d;
            if (cmp(t.abs() * t.abs() * BigInt(2), modulus) <= 0) {
                numer[i] = t;
                continue;
            }
            if (!rational_reconstruct(t, modulus, n, d)) { ok = false; break; }
            // new common denominator: denom * d; earlier numerators scale by d
            for (int k = 0; k < i; ++k) numer[k] = numer[k] * d;
            denom = denom * d;
            numer[i] = n;
            if (cmp(denom * denom * BigInt(2), modulus) > 0) { ok = false; break; }
        }
        if (ok) {
            // exact verification: A numer == denom * rhs
            for (int i = 0; i < m && ok; ++i) {
                BigInt acc(0);
                if (!transpose)
                    for (int e = B.row_start[i]; e < B.row_start[i + 1]; ++e) acc = acc + BigInt(B.row_value[e]) * numer[B.col_index[e]];
                else
                    for (int e = B.col_start[i]; e < B.col_start[i + 1]; ++e) acc = acc + BigInt(B.value[e]) * numer[B.row_index[e]];
                if (!(acc == denom * BigInt(rhs[i]))) ok = false;
            }
        }
        if (ok) {
            // normalise by the gcd of everything
            BigInt g = denom;
            for (int i = 0; i < m && !(g == BigInt(1)); ++i)
                if (!numer[i].is_zero()) g = BigInt::gcd(g, numer[i]);
            if (!(g == BigInt(1)) && !g.is_zero()) {
                for (auto& v : numer) v = v / g;
                denom = denom / g;
            }
            out->numer = std::move(numer);
            out->denom = denom;
            return true;
        }
        if (steps_done >= max_steps) {
            *message = "Dixon lifting did not converge";
            return false;
        }
        target = steps_done * 2;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// entry point
// ---------------------------------------------------------------------------------------------------
void certify_basis(const StandardForm& form, const std::vector<int>& basis_columns, int device, hipStream_t stream,
                   std::string* objective, bool* certified, long long* repair_pivots, std::string* message) {
    objective->clear();
    *certified = false;
    *repair_pivots = 0;
    message->clear();
    RELP_HIP(hipSetDevice(device));
    const MatrixData& md = form.data;
    const int m = md.nr_rows();
    const int n_p = md.nr_columns();

    // ---- integer scaling: row multipliers (lcm of the denominators of the row's coefficients), cost multiplier; the
    //      right-hand side keeps its own common denominator and any width (presolve leaves ~100-bit values there) ----
    std::vector<SparseColumn> columns(n_p);
    for (int j = 0; j < n_p; ++j) columns[j] = md.column(j);
    std::vector<Rat> rhs = md.right_hand_side();
    std::vector<i128> row_mult(m, 1);
    try {
        for (int j = 0; j < n_p; ++j)
            for (size_t e = 0; e < columns[j].nnz(); ++e) row_mult[columns[j].index[e]] = lcm128(row_mult[columns[j].index[e]], columns[j].value[e].d);
    } catch (const RatOverflow&) {
        *message = "row scaling overflows 128 bits";
        return;
    }
    i128 cost_mult = 1;
    for (int j = 0; j < n_p; ++j) cost_mult = lcm128(cost_mult, md.cost_value(j).d);
    auto scaled = [&](const Rat& v, i128 mult) { return mul_checked(v.n, mult / v.d); };
    auto fits = [](i128 v) { return v < ((i128)1 << 62) && v > -((i128)1 << 62); };

    // basis columns: provider column c >= 0, or artificial -1-k (unit column on its row, cost 0; redundant rows)
    std::vector<int> artificial_rows;
    {
        auto pivots = md.pivot_element_indices();
        std::vector<char> has(m, 0);
        for (auto& [row, column] : pivots) has[row] = 1;
        for (int i = 0; i < m; ++i)
            if (!has[i]) artificial_rows.push_back(i);
    }
    // b_i * row_mult_i = rhs_big[i] / rhs_den  (exact, arbitrary width)
    std::vector<BigInt> rhs_big(m);
    BigInt rhs_den(1);
    {
        std::vector<BigInt> numer(m);
        std::vector<i128> denom(m);
        for (int i = 0; i < m; ++i) {
            const i128 g = gcd128(row_mult[i], rhs[i].d);
            numer[i] = big_from_i128(rhs[i].n) * big_from_i128(row_mult[i] / g);
            denom[i] = rhs[i].d / g;
            const BigInt d = big_from_i128(denom[i]);
            rhs_den = rhs_den / BigInt::gcd(rhs_den, d) * d;
        }
        for (int i = 0; i < m; ++i) rhs_big[i] = numer[i] * (rhs_den / big_from_i128(denom[i]));
    }
    std::vector<int> basis = basis_columns;  // repaired in place by exact pivots when a check fails
    const int max_repairs = 200;
    const u32 primes[] = {2147483647u, 2147483629u, 2147483587u, 2147483579u};

    for (int round = 0; round <= max_repairs; ++round) {
        // ---- integer basis (CSC + CSR) ------------------------------------------------------------------
        IntegerBasis B;
        B.m = m;
        B.col_start.assign(m + 1, 0);
        std::vector<i64> cost_basis(m, 0);
        std::vector<char> in_basis(n_p, 0);
        for (int k = 0; k < m; ++k) {
            int c = basis[k];
            if (c >= 0) {
                in_basis[c] = 1;
                for (size_t e = 0; e < columns[c].nnz(); ++e) {
                    i128 v = scaled(columns[c].value[e], row_mult[columns[c].index[e]]);
                    if (!fits(v)) { *message = "scaled coefficient does not fit 62 bits"; return; }
                    B.row_index.push_back(columns[c].index[e]);
                    B.value.push_back((i64)v);
                }
                i128 cv = scaled(md.cost_value(c), cost_mult);
                if (!fits(cv)) { *message = "scaled cost does not fit 62 bits"; return; }
                cost_basis[k] = (i64)cv;
            } else {
                int row = artificial_rows.at(-1 - c);
                if (!fits(row_mult[row])) { *message = "row multiplier does not fit 62 bits"; return; }
                B.row_index.push_back(row);
                B.value.push_back((i64)row_mult[row]);
            }
            B.col_start[k + 1] = (int)B.row_index.size();
        }
        const size_t nnz = B.row_index.size();
        B.row_start.assign(m + 1, 0);
        B.col_index.resize(nnz);
        B.row_value.resize(nnz);
        for (size_t e = 0; e < nnz; ++e) B.row_start[B.row_index[e] + 1]++;
        for (int i = 0; i < m; ++i) B.row_start[i + 1] += B.row_start[i];
        {
            std::vector<int> fill(B.row_start.begin(), B.row_start.end() - 1);
            for (int k = 0; k < m; ++k)
                for (int e = B.col_start[k]; e < B.col_start[k + 1]; ++e) {
                    int dst = fill[B.row_index[e]]++;
                    B.col_index[dst] = k;
                    B.row_value[dst] = B.value[e];
                }
        }

        // ---- device: C = B^-1 mod p ---------------------------------------------------------------------
        DeviceBuffers buf;
        const int ld = 2 * m;
        u32* dM = buf.alloc<u32>((size_t)m * ld);
        u32* d_colk = buf.alloc<u32>(m);
        int* d_info = buf.alloc<int>(4);
        int* d_col_start = buf.alloc<int>(m + 1);
        int* d_row_index = buf.alloc<int>(nnz);
        i64* d_value = buf.alloc<i64>(nnz);
        int* d_row_start = buf.alloc<int>(m + 1);
        int* d_col_index = buf.alloc<int>(nnz);
        i64* d_row_value = buf.alloc<i64>(nnz);
        RELP_HIP(hipMemcpyAsync(d_col_start, B.col_start.data(), (m + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(d_row_index, B.row_index.data(), nnz * sizeof(int), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(d_value, B.value.data(), nnz * sizeof(i64), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(d_row_start, B.row_start.data(), (m + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(d_col_index, B.col_index.data(), nnz * sizeof(int), hipMemcpyHostToDevice, stream));
        RELP_HIP(hipMemcpyAsync(d_row_value, B.row_value.data(), nnz * sizeof(i64), hipMemcpyHostToDevice, stream));
        u32 p = 0;
        for (u32 candidate : primes) {
            RELP_HIP(hipMemsetAsync(d_info, 0, 4 * sizeof(int), stream));
            hipLaunchKernelGGL(build_augmented_kernel, dim3(m), dim3(64), 0, stream, dM, m, ld, d_col_start, d_row_index, d_value, candidate);
            for (int k = 0; k < m; ++k) {
                hipLaunchKernelGGL(gj_pivot_kernel, dim3(1), dim3(256), 0, stream, dM, m, ld, k, candidate, d_colk, d_info);
                hipLaunchKernelGGL(gj_eliminate_kernel, dim3((ld + 255) / 256, m), dim3(256), 0, stream, dM, m, ld, k, candidate, d_colk, d_info);
            }
            int info[4];
            RELP_HIP(hipMemcpyAsync(info, d_info, sizeof(info), hipMemcpyDeviceToHost, stream));
            RELP_HIP(hipStreamSynchronize(stream));
            if (!info[0]) { p = candidate; break; }
        }
        if (p == 0) { *message = "basis singular modulo every trial prime (singular basis?)"; return; }
        auto solve = [&](const std::vector<i64>& r, int transpose, ExactVector* out) {
            return transpose ? dixon_solve(B, r, 1, p, dM, ld, buf, d_col_start, d_row_index, d_value, stream, out, message)
                             : dixon_solve(B, r, 0, p, dM, ld, buf, d_row_start, d_col_index, d_row_value, stream, out, message);
        };

        // ---- exact primal and dual solutions ------------------------------------------------------------
        // B x = rhs_big: one lifting when every entry fits the 62-bit residual path, else one per 60-bit limb of the
        // right-hand side (the solve is linear; every partial solution is verified exactly, and so is their sum)
        auto solve_wide = [&](const std::vector<BigInt>& r, ExactVector* out) {
            size_t widest = 0;
            for (const BigInt& v : r) widest = std::max(widest, v.bits());
            const int limbs = std::max<int>(1, (int)((widest + 59) / 60));
            out->numer.assign(m, BigInt(0));
            out->denom = BigInt(1);
            BigInt scale(1);
            const BigInt step = big_from_i128((i128)1 << 60);
            for (int k = 0; k < limbs; ++k) {
                std::vector<i64> limb(m);
                for (int i = 0; i < m; ++i) {
                    const i64 v = (i64)extract_bits(r[i], (size_t)60 * k, 60);
                    limb[i] = r[i].sign() < 0 ? -v : v;
                }
                ExactVector part;
                if (!solve(limb, 0, &part)) return false;
                const BigInt g = BigInt::gcd(out->denom, part.denom);
                const BigInt grow = part.denom / g, part_factor = (out->denom / g) * scale;
                for (int i = 0; i < m; ++i) out->numer[i] = out->numer[i] * grow + part.numer[i] * part_factor;
                out->denom = out->denom * grow;
                scale = scale * step;
            }
            return true;
        };
        ExactVector x, y;
        if (!solve_wide(rhs_big, &x)) return;
        x.denom = x.denom * rhs_den;  // x_B = numer / (denom * rhs_den); all sign checks below only need denom > 0
        if (!solve(cost_basis, 1, &y)) return;  // B' y = c_B: the rows of B' are the columns of B

        // ---- checks ---------------------------------------------------------------------------------------
        int worst_row = -1;  // most negative x_B (all share the positive denominator)
        for (int k = 0; k < m; ++k) {
            if (basis[k] < 0 && x.numer[k].sign() > 0) { *message = "artificial variable positive in exact arithmetic"; return; }
            if (x.numer[k].sign() < 0 && (worst_row < 0 || cmp(x.numer[k], x.numer[worst_row]) < 0)) worst_row = k;
        }
        // reduced costs (common positive denominator cost_mult * Dy):  c_j*cost_mult*Dy - sum_i a_ij*row_mult_i*Y_i
        std::vector<BigInt> dhat(n_p);
        int worst_col = -1;
        for (int j = 0; j < n_p; ++j) {
            if (in_basis[j]) continue;
            BigInt acc = big_from_i128(scaled(md.cost_value(j), cost_mult)) * y.denom;
            for (size_t e = 0; e < columns[j].nnz(); ++e)
                acc = acc - big_from_i128(scaled(columns[j].value[e], row_mult[columns[j].index[e]])) * y.numer[columns[j].index[e]];
            dhat[j] = acc;
            if (acc.sign() < 0 && (worst_col < 0 || cmp(acc, dhat[worst_col]) < 0)) worst_col = j;
        }
        if (worst_row < 0 && worst_col < 0) {
            // ---- optimal: objective = (sum_k cost_basis[k] X_k) / (cost_mult * Dx) + fixed ----------------------
            BigInt num(0);
            for (int k = 0; k < m; ++k)
                if (cost_basis[k] != 0) num = num + BigInt(cost_basis[k]) * x.numer[k];
            BigInt den = big_from_i128(cost_mult) * x.denom;
            const Rat& fixed = form.fixed_cost;
            num = num * big_from_i128(fixed.d) + big_from_i128(fixed.n) * den;
            den = den * big_from_i128(fixed.d);
            BigInt g = BigInt::gcd(num, den);
            if (!g.is_zero() && !(g == BigInt(1))) {
                num = num / g;
                den = den / g;
            }
            *objective = num.to_string() + "/" + den.to_string();
            *certified = true;
            *repair_pivots = round;
            return;
        }
        if (round == max_repairs) break;
        auto scaled_column = [&](int j) {
            std::vector<i64> r(m, 0);
            for (size_t e = 0; e < columns[j].nnz(); ++e) r[columns[j].index[e]] = (i64)scaled(columns[j].value[e], row_mult[columns[j].index[e]]);
            return r;
        };
        if (worst_row < 0) {
            // ---- exact primal simplex pivot: entering = most negative reduced cost, ratio test with Bland ties
            //      (tableau/mod.rs:287-313) -------------------------------------------------------------------
            const int q = worst_col;
            ExactVector alpha;
            if (!solve(scaled_column(q), 0, &alpha)) return;
            int leave = -1;
            for (int k = 0; k < m; ++k) {
                if (alpha.numer[k].sign() <= 0) continue;
                if (leave < 0) { leave = k; continue; }
                // x_k/alpha_k < x_l/alpha_l  <=>  X_k * A_l < X_l * A_k  (A > 0, common denominators cancel)
                BigInt lhs = x.numer[k] * alpha.numer[leave], rhs2 = x.numer[leave] * alpha.numer[k];
                int c = cmp(lhs, rhs2);
                if (c < 0 || (c == 0 && basis[k] < basis[leave])) leave = k;
            }
            if (leave < 0) { *message = "exact repair: entering column is unbounded"; return; }
            basis[leave] = q;
        } else if (worst_col < 0) {
            // ---- exact dual simplex pivot on the most infeasible row ---------------------------------------------
            std::vector<i64> unit(m, 0);
            unit[worst_row] = 1;
            ExactVector rho;
            if (!solve(unit, 1, &rho)) return;  // rho = e_p' B^-1 (scaled rows)
            int enter = -1;
            BigInt best_d, best_a;
            for (int j = 0; j < n_p; ++j) {
                if (in_basis[j]) continue;
                BigInt a_pj(0);
                for (size_t e = 0; e < columns[j].nnz(); ++e)
                    a_pj = a_pj + big_from_i128(scaled(columns[j].value[e], row_mult[columns[j].index[e]])) * rho.numer[columns[j].index[e]];
                if (a_pj.sign() >= 0) continue;
                BigInt neg_a = -a_pj;
                // d_j / (-a_pj) minimal: d_j * best_a < best_d * neg_a
                if (enter < 0 || cmp(dhat[j] * best_a, best_d * neg_a) < 0) {
                    enter = j;
                    best_d = dhat[j];
                    best_a = neg_a;
                }
            }
            if (enter < 0) { *message = "exact repair: the LP is infeasible (dual ray)"; return; }
            basis[worst_row] = enter;
        } else {
            *message = "basis neither primal nor dual feasible in exact arithmetic";
            return;
        }
    }
    *message = "exact repair did not converge";
}

}  // namespace relp
