// Probe of ds_read_b64_tr_b16 (gfx950): which LDS elements land in which lane/slot.
// hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip -o tools/tr_probe.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__global__ void probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    unsigned addr;
    if (mode == 0) addr = l * 8;                       // lane-contiguous 8-byte pieces
    else if (mode == 1) addr = (l & 15) * 64 + (l >> 4) * 8;   // 16 rows of 32 elements (64 B), lane group picks 4-column block
    else addr = (l & 15) * 128 + (l >> 4) * 8;         // row pitch 128 B
    addr += (unsigned)(uintptr_t)lds;                  // generic->LDS offset (low 32 bits are the LDS address)
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[l * 4 + 0] = v[0] & 0xffff; out[l * 4 + 1] = v[0] >> 16;
    out[l * 4 + 2] = v[1] & 0xffff; out[l * 4 + 3] = v[1] >> 16;
}

int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 3; ++mode) {
        probe<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (element indices; each lane's own address covers elements a..a+3)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("  lane %2d: %4d %4d %4d %4d", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
            if (l % 4 == 3) printf("\n");
        }
    }
    return 0;
}
