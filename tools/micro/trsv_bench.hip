// Micro-benchmark (not part of the product): what the LU route of the north star would cost per FTRAN on gfx950.
// Dense unit-lower / upper triangular solves L y = b, U x = y at m = 821 (25FV47) and other sizes, written the way the
// north star sketches them: one workgroup, diagonal blocks staged in LDS (64 x 64), panel updates as coalesced GEMVs.
// The explicit-inverse FTRAN of the product (a few coalesced column reads, no dependency chain) is timed beside it.
// hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int NB = 64;
constexpr int THREADS = 1024;

// Column-major m x m matrix F with L strictly below the diagonal (unit diagonal implied) and U on and above it.
// In place on x (LDS-resident vector): forward substitution with L, then backward substitution with U.
__global__ void __launch_bounds__(THREADS) lu_solve_kernel(const double* __restrict__ F, const double* __restrict__ b,
                                                        double* __restrict__ out, int m) {
    extern __shared__ double smem[];
    double* x = smem;              // m
    double* blk = smem + m;        // NB x NB diagonal block, column-major
    for (int i = threadIdx.x; i < m; i += THREADS) x[i] = b[i];
    __syncthreads();
    // ---- L y = b ----------------------------------------------------------------------------------
    for (int k0 = 0; k0 < m; k0 += NB) {
        const int kb = min(NB, m - k0);
        for (int e = threadIdx.x; e < kb * kb; e += THREADS) blk[e] = F[(size_t)(k0 + e / kb) * m + k0 + e % kb];
        __syncthreads();
        if (threadIdx.x < 64) {  // one wave solves the diagonal block: kb dependent steps
            const int i = threadIdx.x;
            double xi = i < kb ? x[k0 + i] : 0.0;
            for (int c = 0; c < kb; ++c) {
                const double xc = __shfl(xi, c);
                if (i > c && i < kb) xi -= blk[c * kb + i] * xc;
            }
            if (i < kb) x[k0 + i] = xi;
        }
        __syncthreads();
        // panel update: x[k0+kb:] -= L[k0+kb:, k0:k0+kb] * x[k0:k0+kb]  (coalesced over rows)
        for (int i = k0 + kb + threadIdx.x; i < m; i += THREADS) {
            double acc = 0.0;
            for (int c = 0; c < kb; ++c) acc += F[(size_t)(k0 + c) * m + i] * x[k0 + c];
            x[i] -= acc;
        }
        __syncthreads();
    }
    // ---- U x = y ----------------------------------------------------------------------------------
    for (int k1 = m; k1 > 0; k1 -= NB) {
        const int k0 = max(0, k1 - NB), kb = k1 - k0;
        for (int e = threadIdx.x; e < kb * kb; e += THREADS) blk[e] = F[(size_t)(k0 + e / kb) * m + k0 + e % kb];
        __syncthreads();
        if (threadIdx.x < 64) {
            const int i = threadIdx.x;
            double xi = i < kb ? x[k0 + i] : 0.0;
            for (int c = kb - 1; c >= 0; --c) {
                double xc = __shfl(xi, c);
                xc /= blk[c * kb + c];
                if (i == c) xi = xc;
                if (i < c) xi -= blk[c * kb + i] * xc;
            }
            if (i < kb) x[k0 + i] = xi;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < k0; i += THREADS) {
            double acc = 0.0;
            for (int c = 0; c < kb; ++c) acc += F[(size_t)(k0 + c) * m + i] * x[k0 + c];
            x[i] -= acc;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < m; i += THREADS) out[i] = x[i];
}

// explicit inverse: alpha = sum_e v_e T[:, r_e] for a sparse column of nnz entries (what the product's K2 does)
__global__ void __launch_bounds__(THREADS) inverse_ftran_kernel(const double* __restrict__ T, const int* rows, const double* vals, int nnz,
                                                             double* __restrict__ out, int m) {
    for (int i = threadIdx.x; i < m; i += THREADS) {
        double acc = 0.0;
        for (int e = 0; e < nnz; ++e) acc += T[(size_t)rows[e] * m + i] * vals[e];
        out[i] = acc;
    }
}

int main() {
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int m : {821, 2048, 4096}) {
        std::vector<double> F((size_t)m * m), b(m);
        srand(7);
        for (int j = 0; j < m; ++j)
            for (int i = 0; i < m; ++i) F[(size_t)j * m + i] = (i == j) ? 4.0 + (rand() % 5) : ((rand() % 16) == 0 ? (rand() % 7 - 3) * 0.125 : 0.0) / 8.0;
        for (int i = 0; i < m; ++i) b[i] = (rand() % 9) - 4;
        double *dF, *db, *dout, *dvals; int* drows;
        CHECK(hipMalloc(&dF, F.size() * 8)); CHECK(hipMalloc(&db, m * 8)); CHECK(hipMalloc(&dout, m * 8));
        CHECK(hipMalloc(&dvals, 64 * 8)); CHECK(hipMalloc(&drows, 64 * 4));
        CHECK(hipMemcpy(dF, F.data(), F.size() * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(db, b.data(), m * 8, hipMemcpyHostToDevice));
        std::vector<int> rows(8); std::vector<double> vals(8, 1.0);
        for (int e = 0; e < 8; ++e) rows[e] = (e * 97) % m;
        CHECK(hipMemcpy(drows, rows.data(), 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dvals, vals.data(), 64, hipMemcpyHostToDevice));
        const size_t lds = (size_t)(m + NB * NB) * 8;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lu_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float best_lu = 1e9f, best_inv = 1e9f;
        for (int rep = 0; rep < 20; ++rep) {
            hipExtLaunchKernelGGL(lu_solve_kernel, dim3(1), dim3(THREADS), lds, s, e0, e1, 0, dF, db, dout, m);
            CHECK(hipStreamSynchronize(s));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 2 && ms < best_lu) best_lu = ms;
            hipExtLaunchKernelGGL(inverse_ftran_kernel, dim3(1), dim3(THREADS), 0, s, e0, e1, 0, dF, drows, dvals, 8, dout, m);
            CHECK(hipStreamSynchronize(s));
            CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 2 && ms < best_inv) best_inv = ms;
        }
        // residual check of the LU solve
        std::vector<double> x(m);
        CHECK(hipMemcpy(db, b.data(), m * 8, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(lu_solve_kernel, dim3(1), dim3(THREADS), lds, s, dF, db, dout, m);
        CHECK(hipStreamSynchronize(s));
        CHECK(hipMemcpy(x.data(), dout, m * 8, hipMemcpyDeviceToHost));
        // y = U x, then L y should equal b
        std::vector<double> y(m, 0.0), r(m, 0.0);
        for (int j = 0; j < m; ++j) for (int i = 0; i <= j; ++i) y[i] += F[(size_t)j * m + i] * x[j];
        for (int j = 0; j < m; ++j) { r[j] += y[j]; for (int i = j + 1; i < m; ++i) r[i] += F[(size_t)j * m + i] * y[j]; }
        double err = 0; for (int i = 0; i < m; ++i) err = std::fmax(err, std::fabs(r[i] - b[i]));
        printf("m=%4d: dense LU solve (L then U, one workgroup, 64x64 LDS blocks) %.1f us; explicit-inverse FTRAN (8-entry column) %.1f us; LU residual %.1e\n",
               m, best_lu * 1e3, best_inv * 1e3, err);
    }
    return 0;
}
