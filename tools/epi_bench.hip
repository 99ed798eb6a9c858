// Store-path micro-benchmark for the GEMM epilogues (GPU box): how long does it take 256 CUs to write the output
// tiles of the hot-path GEMMs, by access pattern?   hipcc --offload-arch=gfx950 -O3 tools/epi_bench.hip -o tools/epi_bench.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// tile 128 x 192 per workgroup of 8 waves (4 x 2), wave sub-tile 32 x 96; tiles row-major over (M/128) x (N/192)
// MODE 0: bf16 out, 8 B per lane, rows of the sub-tile contiguous (the slab epilogue's pattern: 24 lanes per row)
// MODE 1: bf16 out, 16 B per lane, 12 lanes per row
// MODE 2: bf16 out, 16 B per lane, one row per lane pair (the direct epilogue's pattern)
// MODE 3: fp32 out, 16 B per lane, 24 lanes per row (slab pattern)
// MODE 4: fp32 out, 16 B per lane, row per lane (direct pattern)
// MODE 5: fp32 in (residual) + fp32 out + bf16 out, slab pattern (LN-producer epilogue traffic)
// MODE 6: nothing (launch + drain baseline)
// MODE 7: MODE 0 but with a 12 x (v_rcp chain) of VALU work per store (GELU-like)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_store(void* out, void* out2, const float* resid, int N, int tn, float seed) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, wm = wave >> 1, wn = wave & 1;
    const int tm_i = blockIdx.x / tn, tn_i = blockIdx.x % tn;
    const int row0 = tm_i * 128 + wm * 32, col0 = tn_i * 192 + wn * 96;
    if (MODE == 6) return;
    if (MODE == 0 || MODE == 7) {
        unsigned short* o = (unsigned short*)out;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int idx = it * 64 + lane, rr = idx / 24, ch = idx % 24;
                float v = seed + idx;
                if (MODE == 7) {
#pragma unroll
                    for (int k = 0; k < 12; ++k) v = __builtin_amdgcn_rcpf(v + 1.0f) * v + 0.5f;
                }
                u32x2 w = {__float_as_uint(v), __float_as_uint(v * 2.f)};
                *(u32x2*)(o + (size_t)(row0 + sl * 16 + rr) * N + col0 + ch * 4) = w;
            }
    } else if (MODE == 1) {
        unsigned short* o = (unsigned short*)out;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int it = 0; it < 3; ++it) {
                const int idx = it * 64 + lane, rr = idx / 12, ch = idx % 12;
                const unsigned u = __float_as_uint(seed + idx);
                u32x4 w = {u, u + 1, u + 2, u + 3};
                *(u32x4*)(o + (size_t)(row0 + sl * 16 + rr) * N + col0 + ch * 8) = w;
            }
    } else if (MODE == 2) {
        unsigned short* o = (unsigned short*)out;
        const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const unsigned u = __float_as_uint(seed + j);
                u32x4 w = {u, u + 1, u + 2, u + 3};
                *(u32x4*)(o + (size_t)(row0 + fr) * N + col0 + j * 32 + 16 * gp + 8 * fh) = w;
            }
    } else if (MODE == 3) {
        float* o = (float*)out;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int idx = it * 64 + lane, rr = idx / 24, ch = idx % 24;
                f32x4 w = {seed, seed + 1, seed + 2, seed + 3};
                *(f32x4*)(o + (size_t)(row0 + sl * 16 + rr) * N + col0 + ch * 4) = w;
            }
    } else if (MODE == 4) {
        float* o = (float*)out;
        const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 w = {seed, seed + 1, seed + 2, seed + 3};
                *(f32x4*)(o + (size_t)(row0 + fr) * N + col0 + j * 32 + 8 * g + 4 * fh) = w;
            }
    } else if (MODE == 5) {
        float* o = (float*)out;
        unsigned short* o2 = (unsigned short*)out2;
        f32x4 r[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int idx = (k % 6) * 64 + lane, rr = idx / 24, ch = idx % 24;
            r[k] = *(const f32x4*)(resid + (size_t)(row0 + (k / 6) * 16 + rr) * N + col0 + ch * 4);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int idx = (k % 6) * 64 + lane, rr = idx / 24, ch = idx % 24;
            f32x4 w = r[k] + seed;
            const size_t off = (size_t)(row0 + (k / 6) * 16 + rr) * N + col0 + ch * 4;
            *(f32x4*)(o + off) = w;
            u32x2 p = {__float_as_uint(w[0]) >> 16 | (__float_as_uint(w[1]) & 0xffff0000u), __float_as_uint(w[2]) >> 16 | (__float_as_uint(w[3]) & 0xffff0000u)};
            *(u32x2*)(o2 + off) = p;
        }
    }
}

template <int MODE>
static void bench(const char* name, int M, int N, void* a, void* b, const float* r, double bytes) {
    const int tn = N / 192, nwg = (M / 128) * tn;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) k_store<MODE><<<nwg, 512>>>(a, b, r, N, tn, 1.0f);
    const int reps = 40;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) k_store<MODE><<<nwg, 512>>>(a, b, r, N, tn, (float)i);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-64s %4d WGs  %7.2f us/launch  %7.2f MB  %6.2f TB/s\n", name, nwg, us, bytes / 1e6, bytes / us / 1e6);
}

int main() {
    const int M = 7680;
    void *a, *b; float* r;
    hipMalloc(&a, (size_t)M * 3072 * 4); hipMalloc(&b, (size_t)M * 3072 * 2); hipMalloc((void**)&r, (size_t)M * 3072 * 4);
    hipMemset(a, 0, (size_t)M * 3072 * 4); hipMemset(b, 0, (size_t)M * 3072 * 2); hipMemset(r, 0, (size_t)M * 3072 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        bench<6>("empty kernel, 960 WGs", M, 3072, a, b, r, 0);
        bench<6>("empty kernel, 240 WGs", M, 768, a, b, r, 0);
        bench<0>("FFN-up out: bf16, 8 B/lane, slab pattern", M, 3072, a, b, r, (double)M * 3072 * 2);
        bench<7>("FFN-up out: same + VALU chain", M, 3072, a, b, r, (double)M * 3072 * 2);
        bench<1>("FFN-up out: bf16, 16 B/lane, 12 lanes/row", M, 3072, a, b, r, (double)M * 3072 * 2);
        bench<2>("FFN-up out: bf16, 16 B/lane, row per lane (direct)", M, 3072, a, b, r, (double)M * 3072 * 2);
        bench<0>("N=768 bf16, 8 B/lane slab", M, 768, a, b, r, (double)M * 768 * 2);
        bench<3>("N=768 fp32, 16 B/lane slab", M, 768, a, b, r, (double)M * 768 * 4);
        bench<4>("N=768 fp32, 16 B/lane row per lane (direct)", M, 768, a, b, r, (double)M * 768 * 4);
        bench<5>("N=768 LN-producer: fp32 in + fp32 out + bf16 out (slab)", M, 768, a, b, r, (double)M * 768 * 10);
        bench<3>("N=3072 fp32, 16 B/lane slab (94 MB)", M, 3072, a, b, r, (double)M * 3072 * 4);
    }
    return 0;
}
