// Exact certificate for the basis the f64 device simplex ends on: bit-exact rational optimum.
//
// The optimal objective of an LP is a property of the LP, not of the pivot path, so the reference's
// `RationalBig` optimum (tests/netlib/mod.rs:62-70: Carry<RationalBig, LUDecomposition<_>>) can be reproduced by
// proving the final basis B optimal in exact arithmetic:
//     B x_B = b,  x_B >= 0            (primal feasibility; Carry::b, carry/mod.rs:46-66)
//     B' y  = c_B, c_j - a_j'y >= 0   (dual feasibility; Tableau::relative_cost, tableau/mod.rs:106-112)
//     objective = c_B' x_B + fixed_cost  (general_form/mod.rs:840-851)
// Fixed-width integer arithmetic on the device replaces arbitrary precision there (north_star): the two linear
// systems are solved by Dixon p-adic lifting -- one modular inverse C = B^-1 mod p (p < 2^31, 64-bit products: the sparse LU
// factors of B mod p from the host's Markowitz elimination, then all m columns of the inverse at once on the device, one
// thread per column walking the same factor entries) and then, per p-adic digit, a modular mat-vec and an exact integer
// residual update carried in 128-bit accumulators.  Only the assembly of the digits (Horner), the rational reconstruction and the sign checks use
// host big integers (bigint.hpp).  Every reconstructed vector is VERIFIED by exact substitution before it is used.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <numeric>

#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "bigint.hpp"
#include "rational_reconstruct.hpp"
#include "lu_host.hpp"
#include "solver.hpp"

namespace relp {

namespace {

using u32 = uint32_t;
using u64 = uint64_t;
using i64 = long long;

// ---------------------------------------------------------------------------------------------------
// device: Z_p kernels
// ---------------------------------------------------------------------------------------------------
// v mod p for v < 2^64; p = 2^31 - 1 (the first trial prime) folds instead of dividing
__device__ __forceinline__ u32 reduce64(u64 v, u32 p) {
    if (p == 0x7fffffffu) {
        v = (v & 0x7fffffffu) + (v >> 31);   // < 2^34
        v = (v & 0x7fffffffu) + (v >> 31);   // < 2^31 + 8
        u32 r = (u32)v;
        return r >= p ? r - p : r;
    }
    return (u32)(v % p);
}

// All m columns of C = B^-1 mod p from the sparse factors P B Q = L U mod p: column j is the solve of e_j, one thread per
// column; every thread walks the SAME factor entries in the same order (no divergence) and owns column j of the work matrix.
//   l_* / u_*: strictly triangular parts by rows of the position space; dinv = 1 / diag mod p.
// in_lds: the `columns` work columns of a workgroup live in LDS (xs[position][column], side by side) -- a solve is a chain of m
// dependent rows, and a row costs an LDS round trip there instead of two trips to L2; staged: the FACTORS are copied to LDS as
// well by all 256 threads of the workgroup first -- read from global memory every entry was a dependent uniform load of its
// own (index -> operand address), 3 ms at m = 821 for 4.5 k entries; out of LDS the same walk takes 0.2 ms.
// Writes C (row-major: C[slot][j]) and its transpose CT.
__global__ void __launch_bounds__(256) modular_inverse_kernel(int m, u32 p, const int* rowpos, const int* colpos, const int* l_start,
                                                               const int* l_col, const u32* l_val, const int* u_start, const int* u_col,
                                                               const u32* u_val, const u32* dinv, u32* X, u32* C, u32* CT, int in_lds,
                                                               int columns, int staged) {
    extern __shared__ u32 xs[];
    if (staged) {
        const int nl = l_start[m], nu = u_start[m];
        int* s_ls = reinterpret_cast<int*>(xs + (size_t)columns * m);
        int* s_us = s_ls + (m + 1);
        int* s_lc = s_us + (m + 1);
        u32* s_lv = reinterpret_cast<u32*>(s_lc + nl);
        int* s_uc = reinterpret_cast<int*>(s_lv + nl);
        u32* s_uv = reinterpret_cast<u32*>(s_uc + nu);
        u32* s_dinv = s_uv + nu;
        for (int i = threadIdx.x; i <= m; i += blockDim.x) {
            s_ls[i] = l_start[i];
            s_us[i] = u_start[i];
        }
        for (int e = threadIdx.x; e < nl; e += blockDim.x) {
            s_lc[e] = l_col[e];
            s_lv[e] = l_val[e];
        }
        for (int e = threadIdx.x; e < nu; e += blockDim.x) {
            s_uc[e] = u_col[e];
            s_uv[e] = u_val[e];
        }
        for (int i = threadIdx.x; i < m; i += blockDim.x) s_dinv[i] = dinv[i];
        __syncthreads();
        l_start = s_ls;
        u_start = s_us;
        l_col = s_lc;
        l_val = s_lv;
        u_col = s_uc;
        u_val = s_uv;
        dinv = s_dinv;
    }
    if ((int)threadIdx.x >= columns) return;
    const int j = blockIdx.x * columns + threadIdx.x;
    if (j >= m) return;
    const int width = in_lds ? columns : m;
    u32* x = in_lds ? xs + threadIdx.x : X + j;  // element i at x[i * width]
    const int start = rowpos[j];  // e_j in position space
    for (int i = 0; i < m; ++i) x[(size_t)i * width] = i == start ? 1u : 0u;
    const u64 two32 = (1ull << 32) % p;
    // L x = e (unit diagonal), rows ascending; rows before `start` stay zero
    for (int i = start + 1; i < m; ++i) {
        const int a = l_start[i], b = l_start[i + 1];
        if (a == b) continue;
        u64 lo = x[(size_t)i * width], hi = 0;
        for (int e = a; e < b; ++e) {
            const u64 prod = (u64)(p - l_val[e]) * x[(size_t)l_col[e] * width];  // -l x  (mod p)
            lo += prod & 0xffffffffu;
            hi += prod >> 32;
        }
        x[(size_t)i * width] = reduce64((u64)reduce64(hi, p) * two32 + reduce64(lo, p), p);
    }
    // U x = y, rows descending
    for (int i = m - 1; i >= 0; --i) {
        const int a = u_start[i], b = u_start[i + 1];
        u64 lo = x[(size_t)i * width], hi = 0;
        for (int e = a; e < b; ++e) {
            const u64 prod = (u64)(p - u_val[e]) * x[(size_t)u_col[e] * width];
            lo += prod & 0xffffffffu;
            hi += prod >> 32;
        }
        const u32 v = reduce64((u64)reduce64(hi, p) * two32 + reduce64(lo, p), p);
        x[(size_t)i * width] = reduce64((u64)v * dinv[i], p);
    }
    for (int s = 0; s < m; ++s) {
        const u32 v = x[(size_t)colpos[s] * width];
        C[(size_t)s * m + j] = v;
        CT[(size_t)j * m + s] = v;
    }
}

// The same inverse, level scheduled: ONE WAVE per column of C.  The solves of different columns are independent, but 821
// columns are 13 waves of the 1024 the chip holds when a thread walks a whole solve; the rows of one LEVEL of a factor are
// independent too (lu_host.hpp `lu_schedules`: 25-40 levels per triangle on a basis of 25FV47, the first holding half of the
// rows), so the 64 lanes of a wave take the rows of a level side by side and a column costs ~60 short steps instead of 4.6 k
// dependent entries.  Factors and schedules are staged in LDS once per workgroup (4 waves = 4 columns); each wave keeps its
// column there.  Writes CT (row j of CT = column j of C, coalesced); C is its transpose (transpose_kernel).
struct ModularFactors {
    const int* rowpos; const int* colpos;
    const int* l_start; const int* l_col; const u32* l_val;
    const int* u_start; const int* u_col; const u32* u_val;
    const u32* dinv;
    const int* lev_start_l; const int* lev_row_l; int levels_l;
    const int* lev_start_u; const int* lev_row_u; int levels_u;
};
__global__ void __launch_bounds__(256) modular_inverse_levels_kernel(int m, u32 p, ModularFactors f, u32* CT) {
    extern __shared__ u32 smem[];
    const int nl = f.l_start[m], nu = f.u_start[m];
    int* s_ls = reinterpret_cast<int*>(smem);
    int* s_us = s_ls + (m + 1);
    int* s_lc = s_us + (m + 1);
    u32* s_lv = reinterpret_cast<u32*>(s_lc + nl);
    int* s_uc = reinterpret_cast<int*>(s_lv + nl);
    u32* s_uv = reinterpret_cast<u32*>(s_uc + nu);
    u32* s_dinv = s_uv + nu;
    int* s_levl = reinterpret_cast<int*>(s_dinv + m);   // levels_l + 1
    int* s_rowl = s_levl + (f.levels_l + 1);            // m
    int* s_levu = s_rowl + m;                            // levels_u + 1
    int* s_rowu = s_levu + (f.levels_u + 1);             // m
    int* s_colpos = s_rowu + m;                          // m
    u32* s_x = reinterpret_cast<u32*>(s_colpos + m);     // 4 columns of m
    for (int i = threadIdx.x; i <= m; i += 256) {
        s_ls[i] = f.l_start[i];
        s_us[i] = f.u_start[i];
    }
    for (int e = threadIdx.x; e < nl; e += 256) {
        s_lc[e] = f.l_col[e];
        s_lv[e] = f.l_val[e];
    }
    for (int e = threadIdx.x; e < nu; e += 256) {
        s_uc[e] = f.u_col[e];
        s_uv[e] = f.u_val[e];
    }
    for (int i = threadIdx.x; i < m; i += 256) {
        s_dinv[i] = f.dinv[i];
        s_rowl[i] = f.lev_row_l[i];
        s_rowu[i] = f.lev_row_u[i];
        s_colpos[i] = f.colpos[i];
    }
    for (int l = threadIdx.x; l <= f.levels_l; l += 256) s_levl[l] = f.lev_start_l[l];
    for (int l = threadIdx.x; l <= f.levels_u; l += 256) s_levu[l] = f.lev_start_u[l];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 4 + wave;
    if (j >= m) return;
    u32* x = s_x + (size_t)wave * m;
    const int start = f.rowpos[j];  // e_j in position space
    for (int i = lane; i < m; i += 64) x[i] = i == start ? 1u : 0u;
    const u64 two32 = (1ull << 32) % p;
    __builtin_amdgcn_wave_barrier();
    // L x = e (unit diagonal): level 0 has no entries
    for (int l = 1; l < f.levels_l; ++l) {
        for (int r = s_levl[l] + lane; r < s_levl[l + 1]; r += 64) {
            const int i = s_rowl[r];
            u64 lo = x[i], hi = 0;
            for (int e = s_ls[i]; e < s_ls[i + 1]; ++e) {
                const u64 prod = (u64)(p - s_lv[e]) * x[s_lc[e]];  // -l x  (mod p)
                lo += prod & 0xffffffffu;
                hi += prod >> 32;
            }
            x[i] = reduce64((u64)reduce64(hi, p) * two32 + reduce64(lo, p), p);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // U x = y
    for (int l = 0; l < f.levels_u; ++l) {
        for (int r = s_levu[l] + lane; r < s_levu[l + 1]; r += 64) {
            const int i = s_rowu[r];
            u64 lo = x[i], hi = 0;
            for (int e = s_us[i]; e < s_us[i + 1]; ++e) {
                const u64 prod = (u64)(p - s_uv[e]) * x[s_uc[e]];
                lo += prod & 0xffffffffu;
                hi += prod >> 32;
            }
            const u32 v = reduce64((u64)reduce64(hi, p) * two32 + reduce64(lo, p), p);
            x[i] = reduce64((u64)v * s_dinv[i], p);
        }
        __builtin_amdgcn_wave_barrier();
    }
    for (int sl = lane; sl < m; sl += 64) CT[(size_t)j * m + sl] = x[s_colpos[sl]];
}
__global__ void __launch_bounds__(256) transpose_u32_kernel(int m, const u32* in, u32* out) {
    __shared__ u32 tile[32][33];
    const int bx = blockIdx.x * 32, by = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int k = ty; k < 32; k += 8)
        if (by + k < m && bx + tx < m) tile[k][tx] = in[(size_t)(by + k) * m + bx + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (bx + k < m && by + tx < m) out[(size_t)(bx + k) * m + by + tx] = tile[tx][k];
}

// Dixon digit: x = A (r mod p) mod p for a row-major m x m matrix A (C for B x = b, C' for B' y = c).  One wave per output
// entry; the products are accumulated in two 64-bit halves and reduced once (a `% p` per term was 25 x slower).
__global__ void __launch_bounds__(256) dixon_digit_kernel(const u32* A, int m, u32 p, const i64* r, u32* x) {
    extern __shared__ u32 s_r[];
    for (int j = threadIdx.x; j < m; j += blockDim.x) {
        i64 v = r[j] % (i64)p;
        if (v < 0) v += p;
        s_r[j] = (u32)v;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (i >= m) return;
    const u32* row = A + (size_t)i * m;
    u64 lo = 0, hi = 0;
    for (int j = lane; j < m; j += 64) {
        const u64 prod = (u64)row[j] * s_r[j];
        lo += prod & 0xffffffffu;
        hi += prod >> 32;
    }
    for (int off = 32; off > 0; off >>= 1) {
        lo += __shfl_down(lo, off);
        hi += __shfl_down(hi, off);
    }
    if (lane == 0) {
        const u64 two32 = (1ull << 32) % p;
        x[i] = reduce64((u64)reduce64(hi, p) * two32 + reduce64(lo, p), p);
    }
}

// r <- (r - A x) / p exactly, A given by rows (CSR of B for B x = b; CSR of B' = CSC of B for B' y = c).
// 128-bit accumulation: |A_ij| < 2^63 and x_j < 2^31.  One WAVE per row: the lanes share the row's entries (a thread per row
// made the longest row -- hundreds of entries -- the length of the kernel: 12 us per step against 7 for the digit mat-vec).
__global__ void __launch_bounds__(256) dixon_residual_kernel(int m, const int* row_start, const int* col_index, const i64* value,
                                                           const u32* x, i64* r, u32 p, int* info) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (i >= m) return;
    __int128 acc = 0;
    for (int e = row_start[i] + lane; e < row_start[i + 1]; e += 64) acc -= (__int128)value[e] * (i64)x[col_index[e]];
    for (int off = 32; off > 0; off >>= 1) {  // 128-bit tree sum over the wave (two 64-bit halves per shuffle)
        const u64 lo = __shfl_down((u64)(unsigned __int128)acc, off);
        const u64 hi = __shfl_down((u64)((unsigned __int128)acc >> 64), off);
        acc += (__int128)(((unsigned __int128)hi << 64) | lo);
    }
    if (lane != 0) return;
    acc += r[i];
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

// diagnostic timeline (RELP_TIME_CERTIFY=1): where the certificate's wall time goes
struct CertifyTimes {
    double device_digits = 0.0, host_assemble = 0.0, inverse = 0.0, setup = 0.0, checks = 0.0, reconstruct = 0.0, parallel = 0.0;
    double unpack = 0.0, horner = 0.0, combine = 0.0, numerators = 0.0, verify = 0.0, normalise = 0.0;  // parts of host_assemble
    int digit_launches = 0, solves = 0, reconstructs = 0;
};
thread_local CertifyTimes g_times;  // (per host thread: certificates of a batch run concurrently on its worker threads)
double wall_now() {
    using clock = std::chrono::steady_clock;
    return std::chrono::duration<double>(clock::now().time_since_epoch()).count();
}

// Host threads for the big-integer part (Horner assembly of the p-adic digits, products with the common denominator, the
// exact substitution check): m independent entries each.  A small persistent pool; the calling thread works too.
class WorkerPool {
public:
    // Two pools: the primal and the dual lifting of a certificate assemble their digits at the same time (transpose = which).
    static WorkerPool& get(int which = 0) {
        static WorkerPool pools[2];
        return pools[which & 1];
    }
    template <class F>
    void run(int n, F&& fn) {
        if (n <= 0) return;
        std::lock_guard<std::mutex> one_caller(run_mutex_);  // (the two Dixon solves of a certificate run on two host threads)
        if (threads_.empty() || n < 8) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        std::function<void(int)> job = std::forward<F>(fn);
        {
            std::lock_guard<std::mutex> lock(mutex_);
            job_ = &job;
            total_ = n;
            next_.store(0);
            done_.store(0);
            ++generation_;
        }
        wake_.notify_all();
        work();
        std::unique_lock<std::mutex> lock(mutex_);
        finished_.wait(lock, [&] { return done_.load() >= total_ && active_ == 0; });
        job_ = nullptr;
    }

private:
    std::mutex run_mutex_;
    WorkerPool() {
        unsigned count = std::thread::hardware_concurrency();
        if (thread_tuning().certify_threads > 0) count = (unsigned)thread_tuning().certify_threads;  // (the pool is made once: the first certificate of the process decides)
        count = std::min(count, 32u) / 2;  // (per pool)
        for (unsigned t = 1; t < count; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        {
            std::lock_guard<std::mutex> lock(mutex_);
            stop_ = true;
        }
        wake_.notify_all();
        for (auto& t : threads_) t.join();
    }
    void work() {
        while (true) {
            const int i = next_.fetch_add(1);
            if (i >= total_) break;
            (*job_)(i);
            done_.fetch_add(1);
        }
    }
    void loop() {
        unsigned long long seen = 0;
        while (true) {
            {
                std::unique_lock<std::mutex> lock(mutex_);
                wake_.wait(lock, [&] { return stop_ || generation_ != seen; });
                if (stop_) return;
                seen = generation_;
                ++active_;
            }
            work();
            {
                std::lock_guard<std::mutex> lock(mutex_);
                --active_;
            }
            finished_.notify_all();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex mutex_;
    std::condition_variable wake_, finished_;
    std::function<void(int)>* job_ = nullptr;
    int total_ = 0, active_ = 0;
    std::atomic<int> next_{0}, done_{0};
    unsigned long long generation_ = 0;
    bool stop_ = false;
};

struct DeviceBuffers {  // device memory of one certificate, drawn from and returned to the handle's scratch (no hipMalloc / hipFree per call)
    CertifyScratch* scratch = nullptr;
    std::mutex* guard = nullptr;  // (the dual lifting allocates from a second host thread)
    std::vector<void*> ptrs;
    template <class T>
    T* alloc(size_t count) {
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        void* p = nullptr;
        if (scratch) {
            std::lock_guard<std::mutex> lock(*guard);
            p = scratch->take(bytes);
        } else {
            RELP_HIP(hipMalloc(&p, bytes));
        }
        ptrs.push_back(p);
        return reinterpret_cast<T*>(p);
    }
    ~DeviceBuffers() {
        if (scratch) {
            std::lock_guard<std::mutex> lock(*guard);
            for (void* p : ptrs) scratch->give_back(p);
        } else {
            for (void* p : ptrs) (void)hipFree(p);
        }
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
bool dixon_solve(const IntegerBasis& B, const std::vector<i64>& rhs, int transpose, u32 p, const u32* dA,
                 DeviceBuffers& buf, const int* d_row_start, const int* d_col_index, const i64* d_row_value,
                 hipStream_t stream, ExactVector* out, std::string* message, CertifyTimes& times, int first_target = 32,
                 int* digits_used = nullptr) {
    const int m = B.m;
    i64* d_r = buf.alloc<i64>(m);
    int* d_info = buf.alloc<int>(4);
    RELP_HIP(hipMemsetAsync(d_info, 0, 4 * sizeof(int), stream));
    RELP_HIP(hipMemcpyAsync(d_r, rhs.data(), m * sizeof(i64), hipMemcpyHostToDevice, stream));
    bool all_zero = std::all_of(rhs.begin(), rhs.end(), [](i64 v) { return v == 0; });
    out->numer.assign(m, BigInt(0));
    out->denom = BigInt(1);
    if (all_zero) return true;

    std::vector<const u32*> digits;             // digits[step][i]: rows of the downloaded blocks below (no copy)
    std::vector<std::vector<u32>> blocks;       // one block of digit steps per round of the doubling
    int steps_done = 0;
    int target = std::max(8, first_target);
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
        const double t_digits = wall_now();
        times.digit_launches += target - steps_done;
        for (int s = steps_done; s < target; ++s) {
            u32* xs = d_digits + (size_t)s * m;
            hipLaunchKernelGGL(dixon_digit_kernel, dim3((m + 3) / 4), dim3(256), m * sizeof(u32), stream, dA, m, p, d_r, xs);
            hipLaunchKernelGGL(dixon_residual_kernel, dim3((m + 3) / 4), dim3(256), 0, stream, m, d_row_start, d_col_index,
                               d_row_value, xs, d_r, p, d_info);
        }
        blocks.emplace_back((size_t)(target - steps_done) * m);
        std::vector<u32>& flat = blocks.back();
        int info[4];
        RELP_HIP(hipMemcpyAsync(flat.data(), d_digits + (size_t)steps_done * m, flat.size() * sizeof(u32), hipMemcpyDeviceToHost, stream));
        RELP_HIP(hipMemcpyAsync(info, d_info, sizeof(info), hipMemcpyDeviceToHost, stream));
        RELP_HIP(hipStreamSynchronize(stream));
        times.device_digits += wall_now() - t_digits;
        const double t_host = wall_now();
        struct HostTimer {
            double t0;
            double* sum;
            ~HostTimer() { *sum += wall_now() - t0; }
        } host_timer{t_host, &times.host_assemble};
        if (info[1] || info[2]) {
            *message = info[2] ? "Dixon residual overflow (coefficients too large for the 128-bit path)" : "Dixon residual not divisible by p";
            return false;
        }
        for (int s = steps_done; s < target; ++s) digits.push_back(flat.data() + (size_t)(s - steps_done) * m);
        steps_done = target;

        // ---- assemble, reconstruct with a common denominator, verify ---------------------------------
        WorkerPool& pool = WorkerPool::get(transpose);
        double t_part = t_host;
        auto part = [&](double& sum) {
            const double now = wall_now();
            sum += now - t_part;
            t_part = now;
        };
        part(times.unpack);
        BigInt modulus(1);
        for (int s = 0; s < steps_done; ++s) modulus.mul_add_small(p, 0);
        part(times.horner);  // (no Horner pass any more: the numerators are formed from the digits directly, below)
        bool ok = true;
        BigInt denom(1);
        std::vector<BigInt> numer(m);
        const BigInt half = modulus / BigInt(2);
        // numer_i = centred(residue_i * denom mod modulus) is the numerator as soon as denom is the common denominator: all
        // entries in parallel; every entry that is still "large" contributes its own denominator (rational reconstruction,
        // sequential: there are few), then the pass is repeated with the grown denominator.
        std::vector<char> small(m);
        {
            // A random integer combination of the entries has, almost surely, the common denominator of all of them: ONE
            // rational reconstruction instead of one per entry that brings a new factor (46 of them on 25FV47).  Whatever it
            // misses is found by the loop below; every result is verified exactly anyway.
            // (Formed on the digits: column sums of weight x digit stay below 2^57, one carry sweep in base p, one Horner pass --
            // no big-integer arithmetic per entry.)
            std::vector<unsigned long long> column(steps_done, 0);
            std::vector<uint32_t> weight(m);
            unsigned long long state = 0x9E3779B97F4A7C15ull;
            for (int i = 0; i < m; ++i) {
                state = state * 6364136223846793005ull + 1442695040888963407ull;
                weight[i] = (uint32_t)(1 + ((state >> 33) & 0xffff));
            }
            for (int s = 0; s < steps_done; ++s) {
                const u32* row = digits[s];
                unsigned long long sum = 0;
                for (int i = 0; i < m; ++i) sum += (unsigned long long)weight[i] * row[i];
                column[s] = sum;
            }
            unsigned long long carry = 0;
            for (int s = 0; s < steps_done; ++s) {  // (what is carried out of the top digit is a multiple of the modulus)
                const unsigned long long v = column[s] + carry;
                column[s] = v % p;
                carry = v / p;
            }
            BigInt combo(0);
            for (int s = steps_done; s-- > 0;) combo.mul_add_small(p, (uint32_t)column[s]);
            combo.trim();
            BigInt n, d;
            const double t_rr = wall_now();
            const bool found = rational_reconstruct(combo, modulus, n, d, true, p);
            times.reconstruct += wall_now() - t_rr;
            times.reconstructs++;
            if (found && within_wang_bound(d, modulus)) denom = d;
            else ok = false;  // not enough digits yet
        }
        part(times.combine);
        // First pass: every entry against the denominator of the combination, in parallel.  What is still "large" then has a
        // factor the combination lost (a small prime that happened to divide its numerator): one reconstruction finds it, the
        // denominator grows by it, and the entries are multiplied by the factor -- the small ones stay small (no reduction), the
        // others are reduced again.  denom is the lcm of reduced denominators at every point, so the result is in lowest terms
        // (gcd(denom, all numerators) = 1: a prime power q^e || denom divides exactly the denominator of some entry, whose
        // numerator q does not divide) and no gcd pass is needed afterwards.
        // numer_i = (sum_s digit_s[i] p^s) denom  mod p^K  =  sum_s digit_s[i] W_s  mod p^K   with  W_s = denom p^s mod p^K:  the K
        // multipliers are made once (each from the last by one small multiplication and a one-word quotient), and an entry is K
        // multiply-adds of a 31-bit digit into 128-bit accumulators per 64-bit word, one carry sweep and one reduction by a two-word
        // quotient -- instead of a Horner pass, a 4000 x 2000-bit product and a 6000 / 4000-bit division per entry on 32-bit limbs.
        if (ok) {
            typedef unsigned __int128 u128;
            const int words = (int)((modulus.bits() + 63) / 64);
            std::vector<u64> table((size_t)steps_done * words, 0);
            {
                BigInt w = denom % modulus;
                for (int s = 0; s < steps_done; ++s) {
                    u64* row = table.data() + (size_t)s * words;
                    for (size_t l = 0; l < w.mag.size(); ++l) row[l / 2] |= (u64)w.mag[l] << (32 * (l % 2));
                    if (s + 1 < steps_done) {
                        w.mul_add_small(p, 0);
                        w = w % modulus;
                    }
                }
            }
            pool.run(m, [&](int i) {
                std::vector<u128> acc(words, 0);
                for (int s = 0; s < steps_done; ++s) {
                    const u64 d = digits[s][i];
                    if (d == 0) continue;
                    const u64* row = table.data() + (size_t)s * words;
                    for (int l = 0; l < words; ++l) acc[l] += (u128)d * row[l];
                }
                BigInt t;
                t.mag.reserve(2 * words + 4);
                u128 carry = 0;
                for (int l = 0; l < words; ++l) {
                    const u128 v = acc[l] + carry;  // (acc < 2^102, carry < 2^64)
                    t.mag.push_back((uint32_t)(u64)v);
                    t.mag.push_back((uint32_t)((u64)v >> 32));
                    carry = v >> 64;
                }
                while (carry != 0) {
                    t.mag.push_back((uint32_t)(u64)carry);
                    carry >>= 32;
                }
                t.trim();
                t = t % modulus;
                if (cmp(t, half) > 0) t = t - modulus;
                small[i] = within_wang_bound(t, modulus) ? 1 : 0;
                numer[i] = t;
            });
        }
        if (ok) {
            BigInt factor(1);  // product of the factors the combination lost
            auto centred = [&](BigInt t) {
                t = t % modulus;
                if (cmp(t, half) > 0) t = t - modulus;
                else if (t.sign() < 0 && cmp(t.abs(), half) > 0) t = t + modulus;
                return t;
            };
            for (int i = 0; i < m && ok; ++i) {
                if (small[i]) continue;
                const BigInt t = centred(numer[i] * factor);
                if (within_wang_bound(t, modulus)) continue;  // the factors found so far cover it
                BigInt n, d;
                const double t_rr = wall_now();
                const bool found = rational_reconstruct(t, modulus, n, d, true, p);
                times.reconstruct += wall_now() - t_rr;
                times.reconstructs++;
                if (!found || d == BigInt(1)) { ok = false; break; }  // (d = 1 with a large numerator: not enough digits)
                factor = factor * d;
                denom = denom * d;
                if (!within_wang_bound(denom, modulus)) ok = false;
            }
            if (ok && !(factor == BigInt(1))) {
                std::atomic<int> large{0};
                pool.run(m, [&](int i) {
                    if (numer[i].is_zero()) return;
                    BigInt t = numer[i] * factor;
                    if (!small[i] || !within_wang_bound(t, modulus)) {
                        t = centred(t);
                        if (!within_wang_bound(t, modulus)) large.fetch_add(1);
                    }
                    numer[i] = t;
                });
                if (large.load() != 0) ok = false;
            }
        }
        part(times.numerators);
        if (ok) {
            // exact verification: A numer == denom * rhs
            std::atomic<int> bad{0};
            pool.run(m, [&](int i) {
                BigInt acc(0);
                if (!transpose)
                    for (int e = B.row_start[i]; e < B.row_start[i + 1]; ++e) acc = acc + BigInt(B.row_value[e]) * numer[B.col_index[e]];
                else
                    for (int e = B.col_start[i]; e < B.col_start[i + 1]; ++e) acc = acc + BigInt(B.value[e]) * numer[B.row_index[e]];
                if (!(acc == denom * BigInt(rhs[i]))) bad.fetch_add(1);
            });
            if (bad.load() != 0) ok = false;
        }
        part(times.verify);
        if (ok) {
            out->numer = std::move(numer);
            out->denom = denom;
            if (digits_used) {
                // What the NEXT lifting of the same system should start with: the digits this solution needs by the bound used above
                // (2 bits(v) + 2 <= bits(p^K) for its largest numerator and the denominator), plus a margin of four -- not the power
                // of two the doubling happened to stop at (128 where 25FV47 needs 117: a tenth of the device steps and a fifth of the
                // host's quadratic work).
                size_t widest = denom.bits();
                for (const BigInt& v : out->numer) widest = std::max(widest, v.bits());
                const int needed = (int)((2 * widest + 2 + 30) / 31) + 1;  // p > 2^30.99: K digits hold more than 30.99 K bits
                *digits_used = std::min(steps_done, needed + 4);
            }
            part(times.normalise);
            return true;
        }
        if (steps_done >= max_steps) {
            *message = "Dixon lifting did not converge";
            return false;
        }
        // not enough digits: double from a cold start; a hinted start that fell short (another basis of the same LP) grows by a quarter
        target = (first_target > 32 && steps_done < 2 * first_target) ? steps_done + std::max(8, first_target / 4) : steps_done * 2;
    }
}

}  // namespace

void* CertifyScratch::take(size_t bytes) {
    Block* best = nullptr;
    for (Block& b : blocks)
        if (!b.busy && b.bytes >= bytes && (!best || b.bytes < best->bytes)) best = &b;
    if (best && best->bytes <= 4 * bytes + 4096) {  // (a far larger block stays free for a request of its own size)
        best->busy = true;
        return best->ptr;
    }
    void* p = nullptr;
    RELP_HIP(hipMalloc(&p, bytes));
    blocks.push_back(Block{p, bytes, true});
    return p;
}
void CertifyScratch::give_back(void* ptr) {
    for (Block& b : blocks)
        if (b.ptr == ptr) b.busy = false;
}
void CertifyScratch::release() {
    for (Block& b : blocks) (void)hipFree(b.ptr);
    blocks.clear();
    if (second) (void)hipStreamDestroy(second);
    second = nullptr;
}

// ---------------------------------------------------------------------------------------------------
// entry point
// ---------------------------------------------------------------------------------------------------
// What depends on the loaded LP only (kept by the handle between certificates, CertifyScratch::statics).
struct CertifyStatic {
    std::vector<SparseColumn> columns;
    std::vector<Rat> rhs;
    std::vector<i128> row_mult;
    i128 cost_mult = 1;
    std::vector<int> artificial_rows;
    std::vector<BigInt> rhs_big;
    BigInt rhs_den = BigInt(1);
};

// mode 0: the basis is optimal (x_B >= 0, zero artificials, every reduced cost >= 0) -> exact objective.
// mode 1: the LP is INFEASIBLE: the same checks for the phase-one costs (1 on the artificial columns, 0 elsewhere;
//         phase_one.rs:123-179) with a POSITIVE optimum -- the dual solution y is a Farkas certificate (y'A <= 0, y'b > 0).
// mode 2: the LP is UNBOUNDED along provider column `entering`: x_B >= 0, cbar_q < 0 and B^-1 a_q <= 0 exactly
//         (phase_two.rs:53; zero on the rows whose basic variable is a zero-level artificial).
struct ExactPrimal {  // (declared in solver.hpp)
    std::vector<int> basis;     // provider column per row (-1-k: artificial k, value 0)
    std::vector<BigInt> numer;  // x_B[k] = numer[k] / denom
    BigInt denom;
};
std::vector<std::pair<int, std::string>> exact_primal_values(const ExactPrimal& primal) {
    std::vector<std::pair<int, std::string>> out;
    for (size_t k = 0; k < primal.basis.size(); ++k) {
        if (primal.basis[k] < 0 || primal.numer[k].sign() == 0) continue;
        BigInt n = primal.numer[k], d = primal.denom;
        if (d.sign() < 0) { n = -n; d = -d; }
        const BigInt g = BigInt::gcd(n, d);
        if (!g.is_zero() && !(g == BigInt(1))) { n = n / g; d = d / g; }
        out.push_back({primal.basis[k], n.to_string() + "/" + d.to_string()});
    }
    std::sort(out.begin(), out.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    return out;
}

void certify_basis(const StandardForm& form, const std::vector<int>& basis_columns, int device, hipStream_t stream,
                   std::string* objective, bool* certified, long long* repair_pivots, std::string* message, int mode, int entering,
                   std::shared_ptr<const ExactPrimal>* primal, CertifyScratch* scratch) {
    // digit_hints[0 / 1]: p-adic digits the primal / dual solve of this LP needed last time (0: unknown).  The number of digits is
    // found by doubling from 32 (Cramer's bound over-estimates it three-fold); a handle that solves the same LP again -- a warm
    // start, a batch pass, a re-solve after a bound change -- starts where the last certificate ended instead of paying for the
    // two failed reconstructions on the way up.  A wrong hint costs time only: every result is verified by exact substitution.
    int no_hints[2] = {0, 0};
    int* digit_hints = scratch && mode == 0 ? scratch->digit_hints : no_hints;
    std::mutex scratch_guard;
    if (primal) primal->reset();
    objective->clear();
    *certified = false;
    *repair_pivots = 0;
    message->clear();
    RELP_HIP(hipSetDevice(device));
    const MatrixData& md = form.data;
    const int m = md.nr_rows();
    const int n_p = md.nr_columns();
    g_times = CertifyTimes{};
    const double t_begin = wall_now();
    struct Report {
        double t0;
        ~Report() {
            if (!diagnostic("RELP_TIME_CERTIFY")) return;
            fprintf(stderr, "[certify] total %.2f ms: setup %.2f, inverse mod p %.2f, Dixon device %.2f (%d digit steps, %d solves), "
                            "host assemble/reconstruct/verify %.2f (of which %d rational reconstructions %.2f), checks %.2f\n",
                    (wall_now() - t0) * 1e3, g_times.setup * 1e3, g_times.inverse * 1e3, g_times.device_digits * 1e3, g_times.digit_launches,
                    g_times.solves, g_times.host_assemble * 1e3, g_times.reconstructs, g_times.reconstruct * 1e3, g_times.checks * 1e3);
            fprintf(stderr, "[certify]   host parts (summed over the solves): Horner %.2f, combined unknown + its reconstruction %.2f, numerators %.2f, "
                            "exact substitution %.2f, gcd %.2f ms\n", g_times.horner * 1e3, g_times.combine * 1e3, g_times.numerators * 1e3,
                    g_times.verify * 1e3, g_times.normalise * 1e3);
        }
    } report{t_begin};
    const bool timeline = diagnostic("RELP_TIME_CERTIFY");
    auto stamp = [&](const char* what) {
        if (timeline) fprintf(stderr, "[certify]   +%.2f ms %s\n", (wall_now() - t_begin) * 1e3, what);
    };

    // ---- integer scaling: row multipliers (lcm of the denominators of the row's coefficients), cost multiplier; the
    //      right-hand side keeps its own common denominator and any width (presolve leaves ~100-bit values there).  It depends
    //      on the LP only: built once per loaded LP and kept by the handle (0.5 ms of every certificate of 25FV47 otherwise) ----
    std::shared_ptr<const CertifyStatic> statics = scratch ? std::static_pointer_cast<const CertifyStatic>(scratch->statics) : nullptr;
    if (!statics) {
        auto built = std::make_shared<CertifyStatic>();
        built->columns.resize(n_p);
        for (int j = 0; j < n_p; ++j) built->columns[j] = md.column(j);
        built->rhs = md.right_hand_side();
        built->row_mult.assign(m, 1);
        try {
            for (int j = 0; j < n_p; ++j)
                for (size_t e = 0; e < built->columns[j].nnz(); ++e)
                    built->row_mult[built->columns[j].index[e]] = lcm128(built->row_mult[built->columns[j].index[e]], built->columns[j].value[e].d);
        } catch (const RatOverflow&) {
            *message = "row scaling overflows 128 bits";
            return;
        }
        try {
            for (int j = 0; j < n_p; ++j) built->cost_mult = lcm128(built->cost_mult, md.cost_value(j).d);
        } catch (const RatOverflow&) {
            built->cost_mult = 0;  // (only the phase-one certificate, which has its own costs, can do without)
        }
        // basis columns: provider column c >= 0, or artificial -1-k (unit column on its row, cost 0; redundant rows)
        {
            auto pivots = md.pivot_element_indices();
            std::vector<char> has(m, 0);
            for (auto& [row, column] : pivots) has[row] = 1;
            for (int i = 0; i < m; ++i)
                if (!has[i]) built->artificial_rows.push_back(i);
        }
        // b_i * row_mult_i = rhs_big[i] / rhs_den  (exact, arbitrary width)
        built->rhs_big.resize(m);
        {
            std::vector<BigInt> numer(m);
            std::vector<i128> denom(m);
            for (int i = 0; i < m; ++i) {
                const i128 g = gcd128(built->row_mult[i], built->rhs[i].d);
                numer[i] = big_from_i128(built->rhs[i].n) * big_from_i128(built->row_mult[i] / g);
                denom[i] = built->rhs[i].d / g;
                const BigInt d = big_from_i128(denom[i]);
                built->rhs_den = built->rhs_den / BigInt::gcd(built->rhs_den, d) * d;
            }
            for (int i = 0; i < m; ++i) built->rhs_big[i] = numer[i] * (built->rhs_den / big_from_i128(denom[i]));
        }
        statics = built;
        if (scratch) scratch->statics = statics;
    }
    const std::vector<SparseColumn>& columns = statics->columns;
    const std::vector<i128>& row_mult = statics->row_mult;
    if (mode != 1 && statics->cost_mult == 0) {
        *message = "cost scaling overflows 128 bits";
        return;
    }
    const i128 cost_mult = mode == 1 ? (i128)1 : statics->cost_mult;
    const std::vector<int>& artificial_rows = statics->artificial_rows;
    const std::vector<BigInt>& rhs_big = statics->rhs_big;
    const BigInt& rhs_den = statics->rhs_den;
    auto scaled = [&](const Rat& v, i128 mult) { return mul_checked(v.n, mult / v.d); };
    auto scaled_cost = [&](int j) -> i128 { return mode == 1 ? (i128)0 : scaled(md.cost_value(j), cost_mult); };
    auto fits = [](i128 v) { return v < ((i128)1 << 62) && v > -((i128)1 << 62); };
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
                i128 cv = scaled_cost(c);
                if (!fits(cv)) { *message = "scaled cost does not fit 62 bits"; return; }
                cost_basis[k] = (i64)cv;
            } else {
                int row = artificial_rows.at(-1 - c);
                if (!fits(row_mult[row])) { *message = "row multiplier does not fit 62 bits"; return; }
                B.row_index.push_back(row);
                B.value.push_back((i64)row_mult[row]);
                cost_basis[k] = mode == 1 ? 1 : 0;  // artificial::Cost::One in phase one (kind/artificial/partially.rs:42-50)
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

        g_times.setup += wall_now() - t_begin - g_times.setup - g_times.inverse - g_times.device_digits - g_times.host_assemble - g_times.checks;
        const double t_inverse = wall_now();
        // ---- device: C = B^-1 mod p ---------------------------------------------------------------------
        stamp("integer basis built");
        DeviceBuffers buf;
        buf.scratch = scratch;
        buf.guard = &scratch_guard;
        u32* dC = buf.alloc<u32>((size_t)m * m);
        u32* dCT = buf.alloc<u32>((size_t)m * m);
        u32* dX = buf.alloc<u32>((size_t)m * m);
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
        stamp("device buffers allocated, uploads enqueued");
        u32 p = 0;
        for (u32 candidate : primes) {
            // sparse LU of B mod p on the host (Markowitz order; any non-zero pivot is exact in Z_p): a few 10^4 operations
            std::vector<u32> value_mod(nnz);
            for (size_t e = 0; e < nnz; ++e) {
                i64 v = B.value[e] % (i64)candidate;
                if (v < 0) v += candidate;
                value_mod[e] = (u32)v;
            }
            LuOptions lo;
            lo.threshold = 0.0;
            const LuModOps ops{candidate};
            const double t_lu = wall_now();
            const HostLUT<u32> f = lu_factor_t<LuModOps>(m, B.col_start.data(), B.row_index.data(), value_mod.data(), lo, ops);
            if (timeline) fprintf(stderr, "[certify]   modular LU on the host %.2f ms (L %zu, U %zu entries)\n", (wall_now() - t_lu) * 1e3, f.l_col.size(), f.u_col.size());
            if (f.singular) continue;  // singular modulo this prime (or singular): try the next one
            std::vector<u32> dinv(m);
            for (int i = 0; i < m; ++i) dinv[i] = ops.inverse(f.diag[i]);
            const size_t nl = f.l_col.size(), nu = f.u_col.size();
            int* d_rowpos = buf.alloc<int>(m);
            int* d_colpos = buf.alloc<int>(m);
            int* d_ls = buf.alloc<int>(m + 1);
            int* d_us = buf.alloc<int>(m + 1);
            int* d_lc = buf.alloc<int>(nl);
            int* d_uc = buf.alloc<int>(nu);
            u32* d_lv = buf.alloc<u32>(nl);
            u32* d_uv = buf.alloc<u32>(nu);
            u32* d_dinv = buf.alloc<u32>(m);
            RELP_HIP(hipMemcpyAsync(d_rowpos, f.rowpos.data(), m * sizeof(int), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(d_colpos, f.colpos.data(), m * sizeof(int), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(d_ls, f.l_start.data(), (m + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(d_us, f.u_start.data(), (m + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
            if (nl) RELP_HIP(hipMemcpyAsync(d_lc, f.l_col.data(), nl * sizeof(int), hipMemcpyHostToDevice, stream));
            if (nl) RELP_HIP(hipMemcpyAsync(d_lv, f.l_val.data(), nl * sizeof(u32), hipMemcpyHostToDevice, stream));
            if (nu) RELP_HIP(hipMemcpyAsync(d_uc, f.u_col.data(), nu * sizeof(int), hipMemcpyHostToDevice, stream));
            if (nu) RELP_HIP(hipMemcpyAsync(d_uv, f.u_val.data(), nu * sizeof(u32), hipMemcpyHostToDevice, stream));
            RELP_HIP(hipMemcpyAsync(d_dinv, dinv.data(), m * sizeof(u32), hipMemcpyHostToDevice, stream));
            HostLUT<u32> fs = f;
            lu_schedules(fs);
            const int levels_l = (int)fs.lev_start[0].size() - 1, levels_u = (int)fs.lev_start[1].size() - 1;
            const size_t level_lds = ((size_t)2 * (m + 1) + 2 * nl + 2 * nu + m + (levels_l + 1) + m + (levels_u + 1) + m + m + (size_t)4 * m) * sizeof(u32);
            if (level_lds <= 150 * 1024 && !thread_tuning().has(RELP_SW_CERTIFY_NO_LEVELS)) {
                // one wave per column, level by level (modular_inverse_levels_kernel)
                int* d_levl = buf.alloc<int>(levels_l + 1);
                int* d_rowl = buf.alloc<int>(m);
                int* d_levu = buf.alloc<int>(levels_u + 1);
                int* d_rowu = buf.alloc<int>(m);
                RELP_HIP(hipMemcpyAsync(d_levl, fs.lev_start[0].data(), (levels_l + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
                RELP_HIP(hipMemcpyAsync(d_rowl, fs.lev_row[0].data(), m * sizeof(int), hipMemcpyHostToDevice, stream));
                RELP_HIP(hipMemcpyAsync(d_levu, fs.lev_start[1].data(), (levels_u + 1) * sizeof(int), hipMemcpyHostToDevice, stream));
                RELP_HIP(hipMemcpyAsync(d_rowu, fs.lev_row[1].data(), m * sizeof(int), hipMemcpyHostToDevice, stream));
                static PerDeviceOnce configured_levels;  // (certificates run from the worker threads of a batch, possibly on several devices)
                configured_levels.run([] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modular_inverse_levels_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                });
                ModularFactors mf{d_rowpos, d_colpos, d_ls, d_lc, d_lv, d_us, d_uc, d_uv, d_dinv, d_levl, d_rowl, levels_l, d_levu, d_rowu, levels_u};
                hipLaunchKernelGGL(modular_inverse_levels_kernel, dim3((m + 3) / 4), dim3(256), level_lds, stream, m, candidate, mf, dCT);
                hipLaunchKernelGGL(transpose_u32_kernel, dim3((m + 31) / 32, (m + 31) / 32), dim3(256), 0, stream, m, dCT, dC);
            } else {
                // columns per workgroup: as many (a power of two, at most 32) as fit the LDS next to each other -- beside the
                // factors themselves when those fit too
                const size_t factor_bytes = ((size_t)2 * (m + 1) + 2 * nl + 2 * nu + m) * sizeof(u32);
                const size_t lds_cap = 150 * 1024;
                int columns = 32;
                bool staged = factor_bytes + (size_t)8 * m * sizeof(u32) <= lds_cap;  // at least 8 columns beside the factors
                const size_t room = staged ? lds_cap - factor_bytes : lds_cap;
                while (columns > 1 && (size_t)columns * m * sizeof(u32) > room) columns /= 2;
                const bool in_lds = (size_t)columns * m * sizeof(u32) <= room;
                if (!in_lds) staged = false;
                static PerDeviceOnce configured;
                configured.run([] {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&modular_inverse_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                });
                if (!in_lds) columns = 64;
                const size_t lds = in_lds ? (size_t)columns * m * sizeof(u32) + (staged ? factor_bytes : 0) : 0;
                hipLaunchKernelGGL(modular_inverse_kernel, dim3((m + columns - 1) / columns), dim3(staged ? 256 : columns), lds, stream, m, candidate,
                                   d_rowpos, d_colpos, d_ls, d_lc, d_lv, d_us, d_uc, d_uv, d_dinv, dX, dC, dCT, in_lds ? 1 : 0, columns, staged ? 1 : 0);
            }
            const double t_sync = wall_now();
            RELP_HIP(hipStreamSynchronize(stream));  // (the staging vectors above go out of scope)
            if (timeline) fprintf(stderr, "[certify]   waited %.2f ms for the uploads and the inverse kernel\n", (wall_now() - t_sync) * 1e3);
            p = candidate;
            break;
        }
        if (p == 0) { *message = "basis singular modulo every trial prime (singular basis?)"; return; }
        // (The number of p-adic digits is found by doubling: Cramer's bound through |det B| of the row-scaled integer matrix
        //  over-estimates it three-fold on 25FV47, and the cost of the reconstruction grows with the square of it.)
        g_times.inverse += wall_now() - t_inverse;
        stamp("inverse mod p ready");
        int primal_digits = 0;
        auto solve = [&](const std::vector<i64>& r, int transpose, ExactVector* out) {
            g_times.solves++;
            const int first = (!transpose && digit_hints[0] > 0) ? digit_hints[0] : 32;
            int used = 0;
            const bool solved = transpose ? dixon_solve(B, r, 1, p, dCT, buf, d_col_start, d_row_index, d_value, stream, out, message, g_times, 32)
                                          : dixon_solve(B, r, 0, p, dC, buf, d_row_start, d_col_index, d_row_value, stream, out, message, g_times, first, &used);
            if (solved && !transpose && round == 0) primal_digits = std::max(primal_digits, used);
            return solved;
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
                stamp("primal lifting returned");
                const BigInt g = BigInt::gcd(out->denom, part.denom);
                const BigInt grow = part.denom / g, part_factor = (out->denom / g) * scale;
                for (int i = 0; i < m; ++i) out->numer[i] = out->numer[i] * grow + part.numer[i] * part_factor;
                out->denom = out->denom * grow;
                scale = scale * step;
            }
            return true;
        };
        ExactVector x, y;
        {
            // The dual solve B' y = c_B (the rows of B' are the columns of B) runs on a second host thread and stream beside the
            // primal one: the liftings are chains of small kernels that leave most of the GPU idle, and while one solve
            // assembles its digits on the host the other one's kernels run.
            bool dual_ok = false;
            std::string dual_message;
            std::exception_ptr dual_error;
            CertifyTimes dual_times;
            std::thread dual([&] {
                try {
                    RELP_HIP(hipSetDevice(device));
                    hipStream_t second = scratch ? scratch->second : nullptr;
                    if (!second) RELP_HIP(hipStreamCreateWithFlags(&second, hipStreamNonBlocking));
                    if (scratch) scratch->second = second;  // kept by the handle: the next certificate finds it
                    stamp("dual thread has its stream");
                    try {
                        DeviceBuffers dual_buffers;
                        dual_buffers.scratch = scratch;
                        dual_buffers.guard = &scratch_guard;
                        int used = 0;
                        dual_ok = dixon_solve(B, cost_basis, 1, p, dCT, dual_buffers, d_col_start, d_row_index, d_value, second, &y,
                                              &dual_message, dual_times, digit_hints[1] > 0 ? digit_hints[1] : 32, &used);
                        if (dual_ok && round == 0) digit_hints[1] = used;
                        stamp("dual solve done");
                        RELP_HIP(hipStreamSynchronize(second));
                    } catch (...) {
                        if (!scratch) (void)hipStreamDestroy(second);
                        throw;
                    }
                    if (!scratch) (void)hipStreamDestroy(second);
                } catch (...) {
                    dual_error = std::current_exception();
                }
            });
            bool primal_ok = false;
            std::exception_ptr primal_error;
            try {
                primal_ok = solve_wide(rhs_big, &x);
            } catch (...) {
                primal_error = std::current_exception();
            }
            stamp("primal solve done");
            dual.join();
            stamp("dual solve joined");
            if (timeline)
                fprintf(stderr, "[certify]   unpack %.2f / %.2f; primal: device %.2f host %.2f (Horner %.2f combine %.2f numerators %.2f verify %.2f gcd %.2f, %d reconstructions %.2f); "
                                "dual: device %.2f host %.2f (Horner %.2f combine %.2f numerators %.2f verify %.2f gcd %.2f, %d reconstructions %.2f) ms\n",
                        g_times.unpack * 1e3, dual_times.unpack * 1e3, g_times.device_digits * 1e3, g_times.host_assemble * 1e3, g_times.horner * 1e3, g_times.combine * 1e3, g_times.numerators * 1e3,
                        g_times.verify * 1e3, g_times.normalise * 1e3, g_times.reconstructs, g_times.reconstruct * 1e3, dual_times.device_digits * 1e3,
                        dual_times.host_assemble * 1e3, dual_times.horner * 1e3, dual_times.combine * 1e3, dual_times.numerators * 1e3, dual_times.verify * 1e3,
                        dual_times.normalise * 1e3, dual_times.reconstructs, dual_times.reconstruct * 1e3);
            g_times.solves++;
            g_times.device_digits += dual_times.device_digits;
            g_times.host_assemble += dual_times.host_assemble;
            g_times.digit_launches += dual_times.digit_launches;
            g_times.reconstruct += dual_times.reconstruct;
            g_times.reconstructs += dual_times.reconstructs;
            g_times.horner += dual_times.horner;
            g_times.combine += dual_times.combine;
            g_times.numerators += dual_times.numerators;
            g_times.verify += dual_times.verify;
            g_times.normalise += dual_times.normalise;
            if (primal_ok && round == 0 && primal_digits > 0) digit_hints[0] = primal_digits;
            if (primal_error) std::rethrow_exception(primal_error);
            if (dual_error) std::rethrow_exception(dual_error);
            if (!primal_ok) return;
            if (!dual_ok) {
                *message = dual_message;
                return;
            }
        }
        x.denom = x.denom * rhs_den;  // x_B = numer / (denom * rhs_den); all sign checks below only need denom > 0

        // ---- checks ---------------------------------------------------------------------------------------
        const double t_checks = wall_now();
        struct ChecksTimer {
            double t0;
            ~ChecksTimer() { g_times.checks += wall_now() - t0; }
        } checks_timer{t_checks};
        int worst_row = -1;  // most negative x_B (all share the positive denominator)
        for (int k = 0; k < m; ++k) {
            if (mode != 1 && basis[k] < 0 && x.numer[k].sign() > 0) { *message = "artificial variable positive in exact arithmetic"; return; }
            if (x.numer[k].sign() < 0 && (worst_row < 0 || cmp(x.numer[k], x.numer[worst_row]) < 0)) worst_row = k;
        }
        // reduced costs (common positive denominator cost_mult * Dy):  c_j*cost_mult*Dy - sum_i a_ij*row_mult_i*Y_i
        std::vector<BigInt> dhat(n_p);
        int worst_col = -1;
        WorkerPool::get().run(n_p, [&](int j) {  // (independent columns; the most negative one is picked in order below)
            if (in_basis[j]) return;
            BigInt acc = big_from_i128(scaled_cost(j)) * y.denom;
            for (size_t e = 0; e < columns[j].nnz(); ++e)
                acc = acc - big_from_i128(scaled(columns[j].value[e], row_mult[columns[j].index[e]])) * y.numer[columns[j].index[e]];
            dhat[j] = acc;
        });
        for (int j = 0; j < n_p; ++j)
            if (!in_basis[j] && dhat[j].sign() < 0 && (worst_col < 0 || cmp(dhat[j], dhat[worst_col]) < 0)) worst_col = j;
        if (mode == 2) {
            // ---- unbounded ray: x_B >= 0, cbar_q < 0, alpha = B^-1 a_q <= 0 (zero where an artificial is basic) ------------
            if (worst_row >= 0) { *message = "unbounded: the basis is not primal feasible in exact arithmetic"; return; }
            if (entering < 0 || entering >= n_p || in_basis[entering]) { *message = "unbounded: no entering column"; return; }
            if (dhat[entering].sign() >= 0) { *message = "unbounded: the entering column's reduced cost is not negative in exact arithmetic"; return; }
            std::vector<i64> aq(m, 0);
            for (size_t e = 0; e < columns[entering].nnz(); ++e) {
                const i128 v = scaled(columns[entering].value[e], row_mult[columns[entering].index[e]]);
                if (!fits(v)) { *message = "scaled coefficient does not fit 62 bits"; return; }
                aq[columns[entering].index[e]] = (i64)v;
            }
            ExactVector alpha;
            if (!solve(aq, 0, &alpha)) return;
            for (int k = 0; k < m; ++k) {
                const int sgn = alpha.numer[k].sign();
                if (sgn > 0 || (basis[k] < 0 && sgn != 0)) { *message = "unbounded: the ray leaves the feasible region in exact arithmetic"; return; }
            }
            *objective = "-inf";
            *certified = true;
            return;
        }
        if (mode == 1 && worst_row < 0 && worst_col < 0) {
            // ---- infeasible: the phase-one optimum (sum of the artificial variables) is positive -------------------------
            BigInt num(0);
            for (int k = 0; k < m; ++k)
                if (cost_basis[k] != 0) num = num + x.numer[k];
            if (num.sign() <= 0) { *message = "infeasible: the phase-one optimum is zero in exact arithmetic (the LP is feasible)"; return; }
            BigInt den = x.denom;
            BigInt g = BigInt::gcd(num, den);
            if (!g.is_zero() && !(g == BigInt(1))) {
                num = num / g;
                den = den / g;
            }
            *objective = num.to_string() + "/" + den.to_string();  // the exact phase-one optimum: the certified infeasibility
            *certified = true;
            return;
        }
        if (mode == 1) { *message = "infeasible: the final phase-one basis is not optimal in exact arithmetic"; return; }
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
            stamp("checks done, objective reduced");
            *objective = num.to_string() + "/" + den.to_string();
            stamp("objective as decimal text");
            *certified = true;
            *repair_pivots = round;
            if (primal) {  // OptimizationResult::FiniteOptimum(x) in exact form (algorithm/mod.rs:43-47), kept as integers over one denominator
                auto kept = std::make_shared<ExactPrimal>();
                kept->basis = basis;
                kept->numer = std::move(x.numer);
                kept->denom = x.denom;
                *primal = kept;
            }
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
