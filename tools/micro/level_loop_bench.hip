// Micro-benchmark: the level loop of lu_solve_tasks (relp_amd/csrc/lu.hip) on synthetic slots, to see what a level costs and why.
// 1024 threads; `per_level` consecutive slots per level (one wave holds 64 / per_level consecutive levels); TE = 4.
// Variants: 0 = loop as in the kernel; 1 = without the DPP group sum; 2 = without the own-component read-modify-write (plain store);
//           3 = scalar bookkeeping only (no LDS traffic at all).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) double lds_f64;
constexpr int WAVE = 64, TE = 4, NONE = 0x7fffffff;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double group_sum_by(double v, const int g, const unsigned gbits) {
    if (gbits == 0) return v;
    double s1 = v + dpp_f64<0xB1, 0xF>(0.0, v);
    double out = g >= 1 ? s1 : v;
    if (gbits & 2u) {
        const double s2 = s1 + dpp_f64<0x4E, 0xF>(0.0, s1);
        out = g >= 2 ? s2 : out;
    }
    return out;
}
template <int VARIANT>
__global__ void __launch_bounds__(1024) loop(double* out, unsigned long long* ticks, int n_levels, int per_level, int gmax) {
    extern __shared__ double smem[];
    volatile lds_f64* x0 = (volatile lds_f64*)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 4096; i += blockDim.x) x0[i] = 1.0 + i * 1e-6;
    int col[TE];
    double val[TE];
    for (int e = 0; e < TE; ++e) {
        col[e] = (tid * (7 + 6 * e) + 11 * e) & 4095;
        val[e] = 0.001 * (e + 1);
    }
    const int pos = (tid * 5 + 3) & 4095;
    const int lev = 1 + tid / per_level;
    const int flags = ((lane % 4 == 3 || gmax == 0) ? 1 << 8 : 0) | (gmax ? 2 : 0);
    const int g = flags & 0xff;
    const unsigned gbits = (__any(g > 0) ? 1u : 0u) | (__any(g > 1) ? 2u : 0u);
    const double dinv = 0.999;
    __syncthreads();
    int first_lane = 0;
    int wave_next = __builtin_amdgcn_readfirstlane(lev);
    const unsigned long long t0 = clock64();
    for (int l = 1; l < n_levels; ++l) {
        while (wave_next == l) {
            const bool active = lane >= first_lane && lev == l;
            double s0 = 0.0, own0 = 0.0;
            if (VARIANT < 3) {
                double xv[TE];
#pragma unroll
                for (int e = 0; e < TE; ++e) xv[e] = x0[col[e]];
                if (VARIANT < 2) own0 = x0[pos];
#pragma unroll
                for (int e = 0; e < TE; ++e) s0 += val[e] * xv[e];
                if (VARIANT < 1) s0 = group_sum_by(s0, g, gbits);
                if (active && ((flags >> 8) & 1)) x0[pos] = (own0 - s0) * dinv;
            }
            first_lane += __popcll(__ballot(active));
            wave_next = first_lane < WAVE ? __builtin_amdgcn_readlane(lev, first_lane < WAVE ? first_lane : 0) : NONE;
        }
        lds_barrier();
    }
    const unsigned long long t1 = clock64();
    if (tid == 0) ticks[0] = t1 - t0;
    out[tid] = x0[pos] + first_lane;
}
int main() {
    double* out;
    unsigned long long* ticks;
    (void)hipMalloc(&out, 1024 * 8);
    (void)hipMalloc(&ticks, 8);
    for (int per_level : {8, 32, 64}) {
        const int n_levels = 1024 / per_level + 1;
        for (int gmax : {0, 2}) {
            for (int variant = 0; variant < 4; ++variant) {
                for (int rep = 0; rep < 3; ++rep) {
                    if (variant == 0) hipLaunchKernelGGL(loop<0>, dim3(1), dim3(1024), 4096 * 8, 0, out, ticks, n_levels, per_level, gmax);
                    if (variant == 1) hipLaunchKernelGGL(loop<1>, dim3(1), dim3(1024), 4096 * 8, 0, out, ticks, n_levels, per_level, gmax);
                    if (variant == 2) hipLaunchKernelGGL(loop<2>, dim3(1), dim3(1024), 4096 * 8, 0, out, ticks, n_levels, per_level, gmax);
                    if (variant == 3) hipLaunchKernelGGL(loop<3>, dim3(1), dim3(1024), 4096 * 8, 0, out, ticks, n_levels, per_level, gmax);
                    (void)hipDeviceSynchronize();
                }
                unsigned long long t = 0;
                (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
                printf("slots per level %2d  multi-lane rows %d  variant %d: %.0f ticks per level\n", per_level, gmax ? 1 : 0, variant, (double)t / (n_levels - 1));
            }
        }
    }
    return 0;
}
