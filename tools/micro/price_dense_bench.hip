// Micro-benchmark (not part of the product): variants of the dense pricing pass, timed in a loop that mimics the solver's
// pivot sequence (a 134 MB read-modify-write "update" precedes every pricing launch).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int WAVE = 64;
constexpr int THREADS = 1024;

__device__ inline double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct Args {
    const double* A;    // n columns of mp doubles
    const int* pos;     // >= 0: basic (skip)
    const double* pi; const double* rho; const double* w;
    double* out;        // 3 per column
    int n, m, mp;
};

// A: the current kernel's structure
__global__ void __launch_bounds__(THREADS) variant_a(Args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_pi = smem; double* s_rho = smem + a.mp; double* s_w = smem + 2 * a.mp;
    for (int i = threadIdx.x; i < a.mp; i += THREADS) { s_pi[i] = a.pi[i]; s_rho[i] = a.rho[i]; s_w[i] = a.w[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const int half = a.mp / 2;
    for (int jd = blockIdx.x * (THREADS / WAVE) + wave; jd < a.n; jd += waves_total) {
        if (a.pos[jd] >= 0) continue;
        const double2* col = reinterpret_cast<const double2*>(a.A + (size_t)jd * a.mp);
        const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
        const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
        const double2* w2 = reinterpret_cast<const double2*>(s_w);
        double d0 = 0, d1 = 0, d2 = 0;
        for (int k0 = lane; k0 < half; k0 += 8 * WAVE) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int k = k0 + u * WAVE; v[u] = k < half ? col[k] : make_double2(0, 0); }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * WAVE;
                if (k < half) {
                    const double2 x = pi2[k], y = rho2[k], z = w2[k];
                    d0 += v[u].x * x.x + v[u].y * x.y; d1 += v[u].x * y.x + v[u].y * y.y; d2 += v[u].x * z.x + v[u].y * z.y;
                }
            }
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
        if (lane == 0) { a.out[3 * jd] = d0; a.out[3 * jd + 1] = d1; a.out[3 * jd + 2] = d2; }
    }
}

// B: software pipelined (next chunk's loads in flight while this chunk is consumed), pos prefetched one column ahead,
// first loads issued before the LDS fill.  Requires half % (8*64) == 0.
template <int LOADS>
__global__ void __launch_bounds__(THREADS) variant_b(Args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_pi = smem; double* s_rho = smem + a.mp; double* s_w = smem + 2 * a.mp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const int half = a.mp / 2;
    const int chunks = half / (LOADS * WAVE);
    const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
    const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
    const double2* w2 = reinterpret_cast<const double2*>(s_w);
    // find the first non-basic column of this wave
    int jd = blockIdx.x * (THREADS / WAVE) + wave;
    while (jd < a.n && a.pos[jd] >= 0) jd += waves_total;
    double2 v[LOADS], nx[LOADS];
    int chunk = 0;
    if (jd < a.n) {
        const double2* col = reinterpret_cast<const double2*>(a.A + (size_t)jd * a.mp);
#pragma unroll
        for (int u = 0; u < LOADS; ++u) v[u] = col[lane + u * WAVE];
    }
    for (int i = threadIdx.x; i < a.mp; i += THREADS) { s_pi[i] = a.pi[i]; s_rho[i] = a.rho[i]; s_w[i] = a.w[i]; }
    __syncthreads();
    double d0 = 0, d1 = 0, d2 = 0;
    while (jd < a.n) {
        // advance to the next (column, chunk)
        int jn = jd, cn = chunk + 1;
        if (cn == chunks) {
            cn = 0;
            jn = jd + waves_total;
            while (jn < a.n && a.pos[jn] >= 0) jn += waves_total;
        }
        if (jn < a.n) {
            const double2* col = reinterpret_cast<const double2*>(a.A + (size_t)jn * a.mp) + cn * (LOADS * WAVE);
#pragma unroll
            for (int u = 0; u < LOADS; ++u) nx[u] = col[lane + u * WAVE];
        }
        const int base = chunk * (LOADS * WAVE) + lane;
#pragma unroll
        for (int u = 0; u < LOADS; ++u) {
            const int k = base + u * WAVE;
            const double2 x = pi2[k], y = rho2[k], z = w2[k];
            d0 += v[u].x * x.x + v[u].y * x.y; d1 += v[u].x * y.x + v[u].y * y.y; d2 += v[u].x * z.x + v[u].y * z.y;
        }
        if (cn == 0) {
            d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
            if (lane == 0) { a.out[3 * jd] = d0; a.out[3 * jd + 1] = d1; a.out[3 * jd + 2] = d2; }
            d0 = d1 = d2 = 0;
        }
#pragma unroll
        for (int u = 0; u < LOADS; ++u) v[u] = nx[u];
        jd = jn; chunk = cn;
    }
}

// D: pure streaming read of the same columns, no LDS vectors (upper bound for a wave-per-column read)
__global__ void __launch_bounds__(THREADS) variant_d(Args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const int half = a.mp / 2;
    for (int jd = blockIdx.x * (THREADS / WAVE) + wave; jd < a.n; jd += waves_total) {
        if (a.pos[jd] >= 0) continue;
        const double2* col = reinterpret_cast<const double2*>(a.A + (size_t)jd * a.mp);
        double d0 = 0;
        for (int k0 = lane; k0 < half; k0 += 8 * WAVE) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[k0 + u * WAVE];
#pragma unroll
            for (int u = 0; u < 8; ++u) d0 += v[u].x + v[u].y;
        }
        d0 = wave_sum(d0);
        if (lane == 0) a.out[3 * jd] = d0;
    }
}

// E: flat grid-stride streaming read of the whole array (peak calibration)
__global__ void __launch_bounds__(256) variant_e(const double2* p, size_t count, double* out) {
    double acc = 0;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i + 3 * stride < count; i += 4 * stride) {
        double2 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    for (; i < count; i += stride) acc += p[i].x + p[i].y;
    if (acc == 1.2345e-300) out[0] = acc;
}

// F: the whole workgroup streams one column at a time (adjacent waves read adjacent KiB), one LDS reduction per column
__global__ void __launch_bounds__(THREADS) variant_f(Args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double s_part[2][3][THREADS / WAVE];
    double* s_pi = smem; double* s_rho = smem + a.mp; double* s_w = smem + 2 * a.mp;
    for (int i = threadIdx.x; i < a.mp; i += THREADS) { s_pi[i] = a.pi[i]; s_rho[i] = a.rho[i]; s_w[i] = a.w[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int half = a.mp / 2;   // 2048 double2 = 2 per thread
    const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
    const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
    const double2* w2 = reinterpret_cast<const double2*>(s_w);
    // columns of this block: contiguous range
    const int per = (a.n + gridDim.x - 1) / gridDim.x;
    const int j0 = blockIdx.x * per, j1 = min(a.n, j0 + per);
    int buf = 0;
    int jprev = -1;
    for (int jd = j0; jd < j1; ++jd) {
        if (a.pos[jd] >= 0) continue;
        const double2* col = reinterpret_cast<const double2*>(a.A + (size_t)jd * a.mp);
        const int k0 = threadIdx.x, k1 = threadIdx.x + THREADS;
        const double2 v0 = col[k0], v1 = k1 < half ? col[k1] : make_double2(0, 0);
        // finish the previous column while these loads fly
        if (jprev >= 0 && wave == 0) {
            double t0 = lane < THREADS / WAVE ? s_part[buf ^ 1][0][lane] : 0, t1 = lane < THREADS / WAVE ? s_part[buf ^ 1][1][lane] : 0, t2 = lane < THREADS / WAVE ? s_part[buf ^ 1][2][lane] : 0;
            t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2);
            if (lane == 0) { a.out[3 * jprev] = t0; a.out[3 * jprev + 1] = t1; a.out[3 * jprev + 2] = t2; }
        }
        double d0, d1, d2;
        {
            const double2 x = pi2[k0], y = rho2[k0], z = w2[k0];
            d0 = v0.x * x.x + v0.y * x.y; d1 = v0.x * y.x + v0.y * y.y; d2 = v0.x * z.x + v0.y * z.y;
        }
        if (k1 < half) {
            const double2 x = pi2[k1], y = rho2[k1], z = w2[k1];
            d0 += v1.x * x.x + v1.y * x.y; d1 += v1.x * y.x + v1.y * y.y; d2 += v1.x * z.x + v1.y * z.y;
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
        if (lane == 0) { s_part[buf][0][wave] = d0; s_part[buf][1][wave] = d1; s_part[buf][2][wave] = d2; }
        __syncthreads();
        jprev = jd;
        buf ^= 1;
    }
    if (jprev >= 0 && wave == 0) {
        double t0 = lane < THREADS / WAVE ? s_part[buf ^ 1][0][lane] : 0, t1 = lane < THREADS / WAVE ? s_part[buf ^ 1][1][lane] : 0, t2 = lane < THREADS / WAVE ? s_part[buf ^ 1][2][lane] : 0;
        t0 = wave_sum(t0); t1 = wave_sum(t1); t2 = wave_sum(t2);
        if (lane == 0) { a.out[3 * jprev] = t0; a.out[3 * jprev + 1] = t1; a.out[3 * jprev + 2] = t2; }
    }
}

// G: variant A with non-temporal column loads
__global__ void __launch_bounds__(THREADS) variant_g(Args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_pi = smem; double* s_rho = smem + a.mp; double* s_w = smem + 2 * a.mp;
    for (int i = threadIdx.x; i < a.mp; i += THREADS) { s_pi[i] = a.pi[i]; s_rho[i] = a.rho[i]; s_w[i] = a.w[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const int half = a.mp / 2;
    for (int jd = blockIdx.x * (THREADS / WAVE) + wave; jd < a.n; jd += waves_total) {
        if (a.pos[jd] >= 0) continue;
        const double* colp = a.A + (size_t)jd * a.mp;
        const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
        const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
        const double2* w2 = reinterpret_cast<const double2*>(s_w);
        double d0 = 0, d1 = 0, d2 = 0;
        for (int k0 = lane; k0 < half; k0 += 8 * WAVE) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * WAVE;
                v[u].x = __builtin_nontemporal_load(colp + 2 * k);
                v[u].y = __builtin_nontemporal_load(colp + 2 * k + 1);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * WAVE;
                const double2 x = pi2[k], y = rho2[k], z = w2[k];
                d0 += v[u].x * x.x + v[u].y * x.y; d1 += v[u].x * y.x + v[u].y * y.y; d2 += v[u].x * z.x + v[u].y * z.y;
            }
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
        if (lane == 0) { a.out[3 * jd] = d0; a.out[3 * jd + 1] = d1; a.out[3 * jd + 2] = d2; }
    }
}

// H: two vectors only (64 KB LDS, two workgroups per CU), the shape of the BTRAN pass
__global__ void __launch_bounds__(THREADS) variant_h(Args a) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_pi = smem; double* s_rho = smem + a.mp;
    for (int i = threadIdx.x; i < a.mp; i += THREADS) { s_pi[i] = a.pi[i]; s_rho[i] = a.rho[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const int half = a.mp / 2;
    for (int jd = blockIdx.x * (THREADS / WAVE) + wave; jd < a.n; jd += waves_total) {
        if (a.pos[jd] >= 0) continue;
        const double2* col = reinterpret_cast<const double2*>(a.A + (size_t)jd * a.mp);
        const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
        const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
        double d0 = 0, d1 = 0;
        for (int k0 = lane; k0 < half; k0 += 8 * WAVE) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[k0 + u * WAVE];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * WAVE;
                const double2 x = pi2[k], y = rho2[k];
                d0 += v[u].x * x.x + v[u].y * x.y; d1 += v[u].x * y.x + v[u].y * y.y;
            }
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1);
        if (lane == 0) { a.out[3 * jd] = d0; a.out[3 * jd + 1] = d1; }
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// I: float storage, one column per wave (the product's price_dense_kernel<true>)
__global__ void __launch_bounds__(THREADS) variant_i(Args a, const float* A32) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_pi = smem; double* s_rho = smem + a.mp; double* s_w = smem + 2 * a.mp;
    for (int i = threadIdx.x; i < a.mp; i += THREADS) { s_pi[i] = a.pi[i]; s_rho[i] = a.rho[i]; s_w[i] = a.w[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const int quarter = a.mp / 4;
    const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
    const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
    const double2* w2 = reinterpret_cast<const double2*>(s_w);
    for (int jd = blockIdx.x * (THREADS / WAVE) + wave; jd < a.n; jd += waves_total) {
        if (a.pos[jd] >= 0) continue;
        const f32x4* col = reinterpret_cast<const f32x4*>(A32 + (size_t)jd * a.mp);
        double d0 = 0, d1 = 0, d2 = 0;
        for (int k0 = lane; k0 < quarter; k0 += 8 * WAVE) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(col + k0 + u * WAVE);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = k0 + u * WAVE;
                const double x0 = v[u].x, x1 = v[u].y, x2 = v[u].z, x3 = v[u].w;
                const double2 a0 = pi2[2 * k], a1 = pi2[2 * k + 1], b0 = rho2[2 * k], b1 = rho2[2 * k + 1], c0 = w2[2 * k], c1 = w2[2 * k + 1];
                d0 += (x0 * a0.x + x1 * a0.y) + (x2 * a1.x + x3 * a1.y);
                d1 += (x0 * b0.x + x1 * b0.y) + (x2 * b1.x + x3 * b1.y);
                d2 += (x0 * c0.x + x1 * c0.y) + (x2 * c1.x + x3 * c1.y);
            }
        }
        d0 = wave_sum(d0); d1 = wave_sum(d1); d2 = wave_sum(d2);
        if (lane == 0) { a.out[3 * jd] = d0; a.out[3 * jd + 1] = d1; a.out[3 * jd + 2] = d2; }
    }
}
// J: float storage, two columns per wave at a time sharing the LDS reads
__global__ void __launch_bounds__(THREADS) variant_j(Args a, const float* A32) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* s_pi = smem; double* s_rho = smem + a.mp; double* s_w = smem + 2 * a.mp;
    for (int i = threadIdx.x; i < a.mp; i += THREADS) { s_pi[i] = a.pi[i]; s_rho[i] = a.rho[i]; s_w[i] = a.w[i]; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x / WAVE;
    const int waves_total = gridDim.x * (THREADS / WAVE);
    const int quarter = a.mp / 4;
    const double2* pi2 = reinterpret_cast<const double2*>(s_pi);
    const double2* rho2 = reinterpret_cast<const double2*>(s_rho);
    const double2* w2 = reinterpret_cast<const double2*>(s_w);
    int jd = blockIdx.x * (THREADS / WAVE) + wave;
    while (jd < a.n) {
        // next two non-basic columns of this wave
        int ja = jd;
        while (ja < a.n && a.pos[ja] >= 0) ja += waves_total;
        int jb = ja + waves_total;
        while (jb < a.n && a.pos[jb] >= 0) jb += waves_total;
        jd = jb + waves_total;
        if (ja >= a.n) break;
        const bool two = jb < a.n;
        const f32x4* ca = reinterpret_cast<const f32x4*>(A32 + (size_t)ja * a.mp);
        const f32x4* cb = reinterpret_cast<const f32x4*>(A32 + (size_t)(two ? jb : ja) * a.mp);
        double p0 = 0, p1 = 0, p2 = 0, q0 = 0, q1 = 0, q2 = 0;
        for (int k0 = lane; k0 < quarter; k0 += 4 * WAVE) {
            f32x4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { va[u] = __builtin_nontemporal_load(ca + k0 + u * WAVE); vb[u] = __builtin_nontemporal_load(cb + k0 + u * WAVE); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * WAVE;
                const double2 a0 = pi2[2 * k], a1 = pi2[2 * k + 1], b0 = rho2[2 * k], b1 = rho2[2 * k + 1], c0 = w2[2 * k], c1 = w2[2 * k + 1];
                {
                    const double x0 = va[u].x, x1 = va[u].y, x2 = va[u].z, x3 = va[u].w;
                    p0 += (x0 * a0.x + x1 * a0.y) + (x2 * a1.x + x3 * a1.y);
                    p1 += (x0 * b0.x + x1 * b0.y) + (x2 * b1.x + x3 * b1.y);
                    p2 += (x0 * c0.x + x1 * c0.y) + (x2 * c1.x + x3 * c1.y);
                }
                {
                    const double x0 = vb[u].x, x1 = vb[u].y, x2 = vb[u].z, x3 = vb[u].w;
                    q0 += (x0 * a0.x + x1 * a0.y) + (x2 * a1.x + x3 * a1.y);
                    q1 += (x0 * b0.x + x1 * b0.y) + (x2 * b1.x + x3 * b1.y);
                    q2 += (x0 * c0.x + x1 * c0.y) + (x2 * c1.x + x3 * c1.y);
                }
            }
        }
        p0 = wave_sum(p0); p1 = wave_sum(p1); p2 = wave_sum(p2);
        q0 = wave_sum(q0); q1 = wave_sum(q1); q2 = wave_sum(q2);
        if (lane == 0) {
            a.out[3 * ja] = p0; a.out[3 * ja + 1] = p1; a.out[3 * ja + 2] = p2;
            if (two) { a.out[3 * jb] = q0; a.out[3 * jb + 1] = q1; a.out[3 * jb + 2] = q2; }
        }
    }
}

__global__ void dirty_kernel(double2* p, size_t count) {  // stands in for the inverse update: read-modify-write
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < count; i += stride) { double2 v = p[i]; v.x += 1.0; v.y -= 1.0; p[i] = v; }
}

int main(int argc, char** argv) {
    const int n = 8192, m = 4096, mp = 4096;
    const double basic_fraction = argc > 1 ? atof(argv[1]) : 0.25;
    const int reps = 30;
    std::vector<double> hA((size_t)n * mp);
    for (size_t i = 0; i < hA.size(); ++i) hA[i] = 1.0 + (double)((i * 2654435761u) % 100);
    std::vector<int> hpos(n);
    int nonbasic = 0;
    srand(1);
    for (int j = 0; j < n; ++j) { hpos[j] = (rand() / (double)RAND_MAX) < basic_fraction ? 1 : -1; nonbasic += hpos[j] < 0; }
    std::vector<double> hv(mp);
    for (int i = 0; i < mp; ++i) hv[i] = 1.0 / (1 + i % 7);
    double *A, *pi, *rho, *w, *out, *inv;
    int* pos;
    CHECK(hipMalloc(&A, hA.size() * 8)); CHECK(hipMalloc(&pi, mp * 8)); CHECK(hipMalloc(&rho, mp * 8)); CHECK(hipMalloc(&w, mp * 8));
    CHECK(hipMalloc(&out, (size_t)3 * n * 8)); CHECK(hipMalloc(&pos, n * 4)); CHECK(hipMalloc(&inv, (size_t)m * m * 8));
    CHECK(hipMemcpy(A, hA.data(), hA.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(pi, hv.data(), mp * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(rho, hv.data(), mp * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(w, hv.data(), mp * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(pos, hpos.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(inv, 0, (size_t)m * m * 8));
    std::vector<float> hA32(hA.begin(), hA.end());
    float* A32;
    CHECK(hipMalloc(&A32, hA32.size() * 4));
    CHECK(hipMemcpy(A32, hA32.data(), hA32.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_i), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)3 * mp * 8)));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_j), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)3 * mp * 8)));
    Args a = {A, pos, pi, rho, w, out, n, m, mp};
    const size_t lds = (size_t)3 * mp * 8;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_a), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_b<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_b<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_f), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_g), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&variant_h), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipStream_t s;
    CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const double bytes = (double)nonbasic * mp * 8;
    std::vector<double> ref(3 * n), got(3 * n);
    for (int dirty = 1; dirty < 3; ++dirty) {
        for (int variant = 0; variant < 12; ++variant) {
            for (int blocks : {256, 512}) {
                if ((variant == 5 || variant == 6) && blocks != 256) continue;
                std::vector<float> times;
                CHECK(hipMemset(out, 0, (size_t)3 * n * 8));
                for (int r = 0; r < reps; ++r) {
                    if (dirty == 1) dirty_kernel<<<2048, 256, 0, s>>>(reinterpret_cast<double2*>(inv), (size_t)m * m / 2);
                    if (dirty == 2) { variant_e<<<8192, 256, 0, s>>>(reinterpret_cast<const double2*>(inv), (size_t)m * m / 2, out); variant_e<<<8192, 256, 0, s>>>(reinterpret_cast<const double2*>(inv), (size_t)m * m / 2, out); }
                    switch (variant) {
                        case 0: hipExtLaunchKernelGGL(variant_a, dim3(blocks), dim3(THREADS), lds, s, e0, e1, 0, a); break;
                        case 1: hipExtLaunchKernelGGL(variant_b<8>, dim3(blocks), dim3(THREADS), lds, s, e0, e1, 0, a); break;
                        case 2: hipExtLaunchKernelGGL(variant_b<4>, dim3(blocks), dim3(THREADS), lds, s, e0, e1, 0, a); break;
                        case 3: hipExtLaunchKernelGGL(variant_d, dim3(blocks), dim3(THREADS), 0, s, e0, e1, 0, a); break;
                        case 4: hipExtLaunchKernelGGL(variant_e, dim3(blocks * 8), dim3(256), 0, s, e0, e1, 0, reinterpret_cast<const double2*>(A), (size_t)n * mp / 2, out); break;
                        case 5: hipExtLaunchKernelGGL(variant_e, dim3(8192), dim3(256), 0, s, e0, e1, 0, reinterpret_cast<const double2*>(A), (size_t)n * mp / 2, out); break;
                        case 7: hipExtLaunchKernelGGL(variant_f, dim3(blocks), dim3(THREADS), lds, s, e0, e1, 0, a); break;
                        case 8: hipExtLaunchKernelGGL(variant_g, dim3(blocks), dim3(THREADS), lds, s, e0, e1, 0, a); break;
                        case 9: hipExtLaunchKernelGGL(variant_h, dim3(blocks), dim3(THREADS), lds * 2 / 3, s, e0, e1, 0, a); break;
                        case 10: hipExtLaunchKernelGGL(variant_i, dim3(blocks), dim3(THREADS), lds, s, e0, e1, 0, a, A32); break;
                        case 11: hipExtLaunchKernelGGL(variant_j, dim3(blocks), dim3(THREADS), lds, s, e0, e1, 0, a, A32); break;
                        case 6: hipExtLaunchKernelGGL(variant_e, dim3(32768), dim3(256), 0, s, e0, e1, 0, reinterpret_cast<const double2*>(A), (size_t)n * mp / 2, out); break;
                    }
                    CHECK(hipStreamSynchronize(s));
                    float ms;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 5) times.push_back(ms);
                }
                std::sort(times.begin(), times.end());
                const double med = times[times.size() / 2] * 1e-3;
                const double b = (variant >= 4 && variant <= 6) ? (double)n * mp * 8 : (variant >= 10 ? bytes / 2 : bytes);
                const char* names[] = {"A current", "B pipelined x8", "B pipelined x4", "D stream/col", "E flat", "E flat 8192", "E flat 32768", "F block/col", "G nontemporal", "H 2 vectors", "I float", "J float x2 cols"};
                if (variant <= 2 || variant == 7 || variant == 8 || variant >= 10) {
                    CHECK(hipMemcpy(got.data(), out, (size_t)3 * n * 8, hipMemcpyDeviceToHost));
                    if (variant == 0 && blocks == 256 && dirty == 1) ref = got;
                    double worst = 0;
                    for (int j = 0; j < 3 * n; ++j) worst = std::max(worst, std::abs(got[j] - ref[j]) / (1e-300 + std::abs(ref[j])));
                    printf("dirty=%d %-16s blocks=%4d  %7.1f us  %6.0f GB/s  (max rel diff vs A %.1e)\n", dirty, names[variant], blocks, med * 1e6, b / med / 1e9, worst);
                } else {
                    printf("dirty=%d %-16s blocks=%4d  %7.1f us  %6.0f GB/s\n", dirty, names[variant], blocks, med * 1e6, b / med / 1e9);
                }
            }
        }
    }
    return 0;
}
