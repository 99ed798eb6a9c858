// Launch floor of the GEMM grids: empty kernels with the same workgroup size / LDS / register footprint.
// hipcc --offload-arch=gfx950 -O3 tools/empty_launch.hip -o /tmp/empty_launch && /tmp/empty_launch
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OCC>
__global__ __launch_bounds__(512, OCC * 2) void k_empty(float* out) {
    extern __shared__ unsigned char smem[];
    if (threadIdx.x == 9999) out[0] = smem[threadIdx.x];
}
template <int OCC>
static void run(const char* name, int wgs, int lds) {
    float* d; hipMalloc(&d, 4);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_empty<OCC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 10; ++i) k_empty<OCC><<<wgs, 512, lds>>>(d);
    hipEventRecord(a);
    const int N = 200;
    for (int i = 0; i < N; ++i) k_empty<OCC><<<wgs, 512, lds>>>(d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-40s wgs=%5d lds=%6d : %.2f us per launch (back to back)\n", name, wgs, lds, ms * 1e3 / N);
}
int main() {
    run<1>("empty 512 thr, 1 wg/CU regs", 240, 120 * 1024);
    run<1>("empty 512 thr, 1 wg/CU regs", 960, 120 * 1024);
    run<2>("empty 512 thr, 2 wg/CU regs", 960, 80 * 1024);
    run<2>("empty 512 thr, 2 wg/CU regs", 768, 80 * 1024);
    run<2>("empty 512 thr, 2 wg/CU regs", 512, 80 * 1024);
    run<2>("empty 512 thr, 2 wg/CU regs, no lds", 960, 0);
    run<1>("empty 512 thr, 1 wg/CU, 256", 256, 120 * 1024);
    return 0;
}
