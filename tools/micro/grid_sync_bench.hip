// Micro-benchmark (not part of the product): cost of a cooperative-groups grid barrier on gfx950 for several grid sizes,
// with a small cross-workgroup data exchange per phase (each workgroup publishes a value, everyone reads all of them).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
namespace cg = cooperative_groups;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(512) sync_kernel(double* slots, double* out, int iters, int exchange) {
    cg::grid_group grid = cg::this_grid();
    double acc = 0.0;
    for (int it = 0; it < iters; ++it) {
        if (exchange) {
            if (threadIdx.x == 0) slots[blockIdx.x] = (double)(it + blockIdx.x);
        }
        grid.sync();
        if (exchange) {
            double v = 0.0;
            for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) v += slots[b];
            acc += v;
            grid.sync();
        }
    }
    if (acc == -1.0) out[0] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = acc;
}

int main() {
    double *slots, *out;
    CHECK(hipMalloc(&slots, 4096 * 8)); CHECK(hipMalloc(&out, 64));
    hipStream_t s; CHECK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int exchange = 0; exchange < 2; ++exchange)
        for (int blocks : {8, 32, 64, 128, 256}) {
            int iters = 2000;
            void* args[] = {&slots, &out, &iters, &exchange};
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0, s));
                CHECK(hipLaunchCooperativeKernel((void*)sync_kernel, dim3(blocks), dim3(512), args, 0, s));
                CHECK(hipEventRecord(e1, s));
                CHECK(hipStreamSynchronize(s));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep == 2) printf("exchange=%d blocks=%3d: %.2f us per iteration (%d grid syncs each)\n", exchange, blocks, ms * 1e3 / iters, exchange ? 2 : 1);
            }
        }
    return 0;
}
