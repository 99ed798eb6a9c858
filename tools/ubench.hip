// Micro-benchmarks that calibrate the numbers used in DESIGN.md: s_memtime tick rate, s_barrier cost with
// 8 waves, MFMA 32x32x16 bf16 issue rate with 1 and 2 waves per SIMD under full-chip load.
// hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k_barrier(long long* out, int iters) {
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_barrier();
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(512) void k_mfma(long long* out, float* sink, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x & 7); y[j] = (__bf16)1.0f; }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(512) void k_mfma_bar(long long* out, float* sink, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(float)(threadIdx.x & 7); y[j] = (__bf16)1.0f; }
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
        __builtin_amdgcn_s_barrier();
    }
    long long t1 = clock64();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// the GEMM's inner loop without global traffic: per k-step 4 ds_read_b128 (1 A + 3 B fragments, conflict-free
// swizzled addresses) issued two k-steps ahead of the 3 MFMAs that consume them, one barrier per 4 k-steps
__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <int DEPTH>
__global__ __launch_bounds__(512) void k_gemm_loop(long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = threadIdx.x; i < 40960 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    bf16x8 fa[4], fb[4][3];
    auto ld = [&](int ks) {
        fa[ks] = *reinterpret_cast<const bf16x8*>(smem + lds_off(wm * 32 + fr, ks * 2 + fh));
        for (int j = 0; j < 3; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(smem + 128 * 128 + lds_off(wn * 96 + j * 32 + fr, ks * 2 + fh));
    };
    auto touch = [&](int ks) {
        asm volatile("" : "+v"(fa[ks]));
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(fb[ks][j]));
    };
    auto mma = [&](int ks) {
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks][j], acc[j], 0, 0, 0);
    };
    for (int k = 0; k < DEPTH; ++k) ld(k);
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            touch(ks); __builtin_amdgcn_sched_barrier(0);
            if (DEPTH == 2) ld((ks + 2) & 3);
            __builtin_amdgcn_sched_barrier(0);
            mma(ks); __builtin_amdgcn_sched_barrier(0);
            if (DEPTH == 0) ld(ks);            // reload in place after use: reads are NOT ahead
            if (ks == 1) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
        }
    }
    long long t1 = clock64();
    float s = acc[0][0] + acc[1][0] + acc[2][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// same loop + the operand stream: 5 x 1 KiB LDS-DMA pieces per wave per tile into a 3-slot ring, counted vmcnt
template <int MODE, int PANELS = 60, bool LINEAR = false>   // 0: LDS-DMA (buffer_load..lds)  1: global_load to VGPRs + ds_write_b128  2: issue DMA but never wait (overwrites allowed)
__global__ __launch_bounds__(512) void k_gemm_loop_dma(long long* out, float* sink, const unsigned char* __restrict__ src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    for (int i = threadIdx.x; i < 122880 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    bf16x8 fa[4], fb[4][3];
    auto ld = [&](int slot, int ks) {
        const unsigned char* b = smem + slot * 40960;
        fa[ks] = *reinterpret_cast<const bf16x8*>(b + lds_off(wm * 32 + fr, ks * 2 + fh));
        for (int j = 0; j < 3; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(b + 128 * 128 + lds_off(wn * 96 + j * 32 + fr, ks * 2 + fh));
    };
    auto touch = [&](int ks) {
        asm volatile("" : "+v"(fa[ks]));
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(fb[ks][j]));
    };
    auto mma = [&](int ks) {
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks][j], acc[j], 0, 0, 0);
    };
    // each workgroup streams its own 320-row x (iters*128 B) panel; row stride 1536 B like a K=768 bf16 matrix
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    unsigned voff[5];
    for (int i = 0; i < 5; ++i) {
        const int g = i * 8 + wave, r = g * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        voff[i] = LINEAR ? (unsigned)((blockIdx.x % PANELS) * 491520 + g * 1024 + lane * 16)     // 1 KiB contiguous per piece
                         : (unsigned)(((blockIdx.x % PANELS) * 320 + r) * 1536 + c * 16);
    }
    uint4 regs[5];
    auto piece = [&](int slot, int t, int i) {
        const int soff = LINEAR ? (t % 12) * 40960 : (t % 12) * 128;
        const int g = i * 8 + wave;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + slot * 40960 + g * 1024), 16, voff[i], soff, 0, 0);
    };
    auto stage = [&](int slot, int t) {
        const int soff = LINEAR ? (t % 12) * 40960 : (t % 12) * 128;
        for (int i = 0; i < 5; ++i) {
            const int g = i * 8 + wave;
            if (MODE == 1) regs[i] = *reinterpret_cast<const uint4*>(src + voff[i] + soff);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + slot * 40960 + g * 1024), 16, voff[i], soff, 0, 0);
        }
    };
    auto commit = [&](int slot) {     // MODE 1: registers -> LDS
        for (int i = 0; i < 5; ++i) {
            const int g = i * 8 + wave;
            *reinterpret_cast<uint4*>(smem + slot * 40960 + g * 1024 + lane * 16) = regs[i];
        }
    };
    if (MODE != 1) { stage(0, 0); stage(1, 1); asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    ld(0, 0); ld(0, 1);
    int slot = 0;
    long long t0 = clock64();
    for (int t = 0; t < iters; ++t) {
        int nslot = slot + 1 == 3 ? 0 : slot + 1;
        int ns2 = slot + 2 >= 3 ? slot - 1 : slot + 2;
        if (MODE != 3) stage(ns2, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        touch(0); __builtin_amdgcn_sched_barrier(0); ld(slot, 2); __builtin_amdgcn_sched_barrier(0); mma(0); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 0); piece(ns2, t + 2, 4); __builtin_amdgcn_sched_barrier(0); }
        touch(1); __builtin_amdgcn_sched_barrier(0); ld(slot, 3); __builtin_amdgcn_sched_barrier(0); mma(1); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 1); __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        touch(2); __builtin_amdgcn_sched_barrier(0); ld(nslot, 0); __builtin_amdgcn_sched_barrier(0); mma(2); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 2); __builtin_amdgcn_sched_barrier(0); }
        touch(3); __builtin_amdgcn_sched_barrier(0); ld(nslot, 1); __builtin_amdgcn_sched_barrier(0); mma(3); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 3); __builtin_amdgcn_sched_barrier(0); }
        if (MODE == 1) { commit(ns2); __builtin_amdgcn_sched_barrier(0); }     // compiler waits vmcnt for regs[] here
        slot = nslot;
    }
    long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = acc[0][0] + acc[1][0] + acc[2][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// same loop + operand stream + VALU payload per k-step (does epilogue math of ANOTHER tile hide under the MFMAs of this wave?)
// same loop + the operand stream: 5 x 1 KiB LDS-DMA pieces per wave per tile into a 3-slot ring, counted vmcnt
template <int MODE, int PANELS, bool LINEAR, int VALU>   // 0: LDS-DMA (buffer_load..lds)  1: global_load to VGPRs + ds_write_b128  2: issue DMA but never wait (overwrites allowed)
__global__ __launch_bounds__(512) void k_gemm_loop_valu(long long* out, float* sink, const unsigned char* __restrict__ src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    for (int i = threadIdx.x; i < 122880 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    bf16x8 fa[4], fb[4][3];
    float pv[8];
    for (int e = 0; e < 8; ++e) pv[e] = (float)(lane + e);
    auto payload = [&]() {
#pragma unroll
        for (int q = 0; q < VALU; ++q) { pv[q & 7] = __builtin_fmaf(pv[q & 7], 1.0001f, 0.5f); asm volatile("" : "+v"(pv[q & 7])); }
    };
    auto ld = [&](int slot, int ks) {
        const unsigned char* b = smem + slot * 40960;
        fa[ks] = *reinterpret_cast<const bf16x8*>(b + lds_off(wm * 32 + fr, ks * 2 + fh));
        for (int j = 0; j < 3; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(b + 128 * 128 + lds_off(wn * 96 + j * 32 + fr, ks * 2 + fh));
    };
    auto touch = [&](int ks) {
        asm volatile("" : "+v"(fa[ks]));
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(fb[ks][j]));
    };
    auto mma = [&](int ks) {
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks][j], acc[j], 0, 0, 0);
    };
    // each workgroup streams its own 320-row x (iters*128 B) panel; row stride 1536 B like a K=768 bf16 matrix
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
    unsigned voff[5];
    for (int i = 0; i < 5; ++i) {
        const int g = i * 8 + wave, r = g * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
        voff[i] = LINEAR ? (unsigned)((blockIdx.x % PANELS) * 491520 + g * 1024 + lane * 16)     // 1 KiB contiguous per piece
                         : (unsigned)(((blockIdx.x % PANELS) * 320 + r) * 1536 + c * 16);
    }
    uint4 regs[5];
    auto piece = [&](int slot, int t, int i) {
        const int soff = LINEAR ? (t % 12) * 40960 : (t % 12) * 128;
        const int g = i * 8 + wave;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + slot * 40960 + g * 1024), 16, voff[i], soff, 0, 0);
    };
    auto stage = [&](int slot, int t) {
        const int soff = LINEAR ? (t % 12) * 40960 : (t % 12) * 128;
        for (int i = 0; i < 5; ++i) {
            const int g = i * 8 + wave;
            if (MODE == 1) regs[i] = *reinterpret_cast<const uint4*>(src + voff[i] + soff);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + slot * 40960 + g * 1024), 16, voff[i], soff, 0, 0);
        }
    };
    auto commit = [&](int slot) {     // MODE 1: registers -> LDS
        for (int i = 0; i < 5; ++i) {
            const int g = i * 8 + wave;
            *reinterpret_cast<uint4*>(smem + slot * 40960 + g * 1024 + lane * 16) = regs[i];
        }
    };
    if (MODE != 1) { stage(0, 0); stage(1, 1); asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    ld(0, 0); ld(0, 1);
    int slot = 0;
    long long t0 = clock64();
    for (int t = 0; t < iters; ++t) {
        int nslot = slot + 1 == 3 ? 0 : slot + 1;
        int ns2 = slot + 2 >= 3 ? slot - 1 : slot + 2;
        if (MODE != 3) stage(ns2, t + 2);
        __builtin_amdgcn_sched_barrier(0);
        touch(0); __builtin_amdgcn_sched_barrier(0); ld(slot, 2); __builtin_amdgcn_sched_barrier(0); mma(0); payload(); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 0); piece(ns2, t + 2, 4); __builtin_amdgcn_sched_barrier(0); }
        touch(1); __builtin_amdgcn_sched_barrier(0); ld(slot, 3); __builtin_amdgcn_sched_barrier(0); mma(1); payload(); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 1); __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
        if (MODE == 0) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        touch(2); __builtin_amdgcn_sched_barrier(0); ld(nslot, 0); __builtin_amdgcn_sched_barrier(0); mma(2); payload(); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 2); __builtin_amdgcn_sched_barrier(0); }
        touch(3); __builtin_amdgcn_sched_barrier(0); ld(nslot, 1); __builtin_amdgcn_sched_barrier(0); mma(3); payload(); __builtin_amdgcn_sched_barrier(0);
        if (MODE == 3) { piece(ns2, t + 2, 3); __builtin_amdgcn_sched_barrier(0); }
        if (MODE == 1) { commit(ns2); __builtin_amdgcn_sched_barrier(0); }     // compiler waits vmcnt for regs[] here
        slot = nslot;
    }
    long long t1 = clock64();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = acc[0][0] + acc[1][0] + acc[2][0] + pv[0] + pv[1] + pv[2] + pv[3] + pv[4] + pv[5] + pv[6] + pv[7];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// register-staged operand stream, software-pipelined: the global loads of tile t+2 are issued at the top of
// iteration t into one of two register sets and written to LDS (ds_write_b128) at the end of iteration t+1,
// so a full iteration of MFMA work covers their latency.  Tests whether ds_write_b128 is cheaper for the LDS
// port than the LDS-DMA path.
template <int PANELS>
__global__ __launch_bounds__(512, 2) void k_gemm_loop_reg(long long* out, float* sink, const unsigned char* __restrict__ src, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wm = wave >> 1, wn = wave & 1;
    for (int i = threadIdx.x; i < 122880 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    f32x16 acc[3];
    for (int a = 0; a < 3; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int fr = lane & 31, fh = lane >> 5;
    bf16x8 fa[4], fb[4][3];
    auto ld = [&](int slot, int ks) {
        const unsigned char* b = smem + slot * 40960;
        fa[ks] = *reinterpret_cast<const bf16x8*>(b + lds_off(wm * 32 + fr, ks * 2 + fh));
        for (int j = 0; j < 3; ++j) fb[ks][j] = *reinterpret_cast<const bf16x8*>(b + 128 * 128 + lds_off(wn * 96 + j * 32 + fr, ks * 2 + fh));
    };
    auto touch = [&](int ks) {
        asm volatile("" : "+v"(fa[ks]));
        for (int j = 0; j < 3; ++j) asm volatile("" : "+v"(fb[ks][j]));
    };
    auto mma = [&](int ks) {
        for (int j = 0; j < 3; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks], fb[ks][j], acc[j], 0, 0, 0);
    };
    unsigned voff[5];
    for (int i = 0; i < 5; ++i) {
        const int g = i * 8 + wave, r = g * 8 + (lane >> 3), c = (lane & 7);
        voff[i] = (unsigned)(((blockIdx.x % PANELS) * 320 + r) * 1536 + c * 16);
    }
    uint4 a0, a1, a2, a3, a4, b0, b1, b2, b3, b4;      // two register sets (named scalars: arrays behind lambdas end up in scratch)
#define UB_LOAD(i, t) (*reinterpret_cast<const uint4*>(src + voff[i] + ((t) % 12) * 128))
#define UB_STORE(i, slot, v) (*reinterpret_cast<uint4*>(smem + (slot) * 40960 + lds_off(((i) * 8 + wave) * 8 + (lane >> 3), lane & 7)) = (v))
#define UB_SB() __builtin_amdgcn_sched_barrier(0)
#define UB_ITER(t, n0, n1, n2, n3, n4, o0, o1, o2, o3, o4)                                                  \
    {                                                                                                       \
        const int nslot = slot + 1 == 3 ? 0 : slot + 1, wslot = slot + 2 >= 3 ? slot - 1 : slot + 2;        \
        n0 = UB_LOAD(0, (t) + 2); n1 = UB_LOAD(1, (t) + 2); n2 = UB_LOAD(2, (t) + 2); n3 = UB_LOAD(3, (t) + 2); n4 = UB_LOAD(4, (t) + 2); \
        UB_SB(); touch(0); UB_SB(); ld(slot, 2); UB_SB(); mma(0); UB_SB();                                  \
        touch(1); UB_SB(); ld(slot, 3); UB_SB(); mma(1); UB_SB();                                           \
        __builtin_amdgcn_s_barrier(); UB_SB();                                                              \
        touch(2); UB_SB(); ld(nslot, 0); UB_SB(); mma(2); UB_SB();                                          \
        UB_STORE(0, wslot, o0); UB_STORE(1, wslot, o1); UB_STORE(2, wslot, o2); UB_STORE(3, wslot, o3); UB_STORE(4, wslot, o4); \
        UB_SB(); touch(3); UB_SB(); ld(nslot, 1); UB_SB(); mma(3); UB_SB();                                 \
        slot = nslot;                                                                                       \
    }
    b0 = UB_LOAD(0, 1); b1 = UB_LOAD(1, 1); b2 = UB_LOAD(2, 1); b3 = UB_LOAD(3, 1); b4 = UB_LOAD(4, 1);
    __builtin_amdgcn_s_barrier();
    ld(0, 0); ld(0, 1);
    int slot = 0;
    long long t0 = clock64();
    for (int t = 0; t < iters; t += 2) {
        UB_ITER(t, a0, a1, a2, a3, a4, b0, b1, b2, b3, b4)
        UB_ITER(t + 1, b0, b1, b2, b3, b4, a0, a1, a2, a3, a4)
    }
    long long t1 = clock64();
    float s = acc[0][0] + acc[1][0] + acc[2][0];
    if (s == 12345.f) sink[0] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <typename F>
static void run(const char* name, F launch, long long* d, int nblk, double work_per_blk_iter, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[1024];
    hipMemcpy(h, d, sizeof(long long) * nblk, hipMemcpyDeviceToHost);
    double mean = 0; for (int i = 0; i < nblk; ++i) mean += h[i]; mean /= nblk;
    printf("%-34s wall %8.2f us  ticks/blk %10.0f  ticks/iter %8.1f  ticks/ns %.3f  %s\n", name, ms * 1e3, mean, mean / iters,
           mean / (ms * 1e6), "");
    if (work_per_blk_iter > 0) printf("%-34s   -> %.1f TFLOP/s\n", "", work_per_blk_iter * iters * nblk / (ms * 1e-3) / 1e12);
}

int main() {
    long long* d; float* sink;
    hipMalloc(&d, 1024 * sizeof(long long)); hipMalloc(&sink, 16);
    const int iters = 2000;
    run("barrier x8 waves, 256 WGs", [&] { k_barrier<<<256, 512>>>(d, iters); }, d, 256, 0, iters);
    run("barrier x4 waves, 256 WGs", [&] { k_barrier<<<256, 256>>>(d, iters); }, d, 256, 0, iters);
    const double fl = 2.0 * 32 * 32 * 16;
    run("mfma 3 acc, 8 waves/WG (2/SIMD)", [&] { k_mfma<3><<<256, 512>>>(d, sink, iters); }, d, 256, fl * 3 * 8, iters);
    run("mfma 3 acc, 4 waves/WG (1/SIMD)", [&] { k_mfma<3><<<256, 256>>>(d, sink, iters); }, d, 256, fl * 3 * 4, iters);
    run("mfma 12 acc, 4 waves/WG", [&] { k_mfma<12><<<256, 256>>>(d, sink, iters); }, d, 256, fl * 12 * 4, iters);
    run("mfma 12 + barrier, 8 waves/WG", [&] { k_mfma_bar<12><<<256, 512>>>(d, sink, iters); }, d, 256, fl * 12 * 8, iters);
    run("mfma 12 + barrier, 4 waves/WG", [&] { k_mfma_bar<12><<<256, 256>>>(d, sink, iters); }, d, 256, fl * 12 * 4, iters);
    run("mfma 3 + barrier, 8 waves/WG", [&] { k_mfma_bar<3><<<256, 512>>>(d, sink, iters); }, d, 256, fl * 3 * 8, iters);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    unsigned char* src; hipMalloc(&src, 64u << 20); hipMemset(src, 0, 64u << 20);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    run("gemm loop + LDS-DMA stream, 8w", [&] { k_gemm_loop_dma<0><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    run("gemm loop + LDS-DMA, 2 hot panels", [&] { k_gemm_loop_dma<0, 2><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<0, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    run("gemm loop + LDS-DMA, 8 hot panels", [&] { k_gemm_loop_dma<0, 8><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<0, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    run("LDS-DMA, 8 hot panels, LINEAR src", [&] { k_gemm_loop_dma<0, 8, true><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_dma<0, 60, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    run("LDS-DMA, 60 panels, LINEAR src", [&] { k_gemm_loop_dma<0, 60, true><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("gemm loop + LDS-DMA spread, 8w", [&] { k_gemm_loop_dma<3><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("gemm loop + reg-staged stream, 8w", [&] { k_gemm_loop_dma<1><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_reg<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_reg<60>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    run("reg-staged PIPELINED, 2 hot panels", [&] { k_gemm_loop_reg<2><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("reg-staged PIPELINED, 60 panels", [&] { k_gemm_loop_reg<60><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_valu<0, 2, false, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_valu<0, 2, false, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_valu<0, 2, false, 24>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_gemm_loop_valu<0, 2, false, 48>), hipFuncAttributeMaxDynamicSharedMemorySize, 122880);
    run("DMA loop + 0 VALU/k-step", [&] { k_gemm_loop_valu<0, 2, false, 0><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("DMA loop + 12 VALU/k-step", [&] { k_gemm_loop_valu<0, 2, false, 12><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("DMA loop + 24 VALU/k-step", [&] { k_gemm_loop_valu<0, 2, false, 24><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("DMA loop + 48 VALU/k-step", [&] { k_gemm_loop_valu<0, 2, false, 48><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("gemm loop + LDS-DMA no wait, 8w", [&] { k_gemm_loop_dma<2><<<256, 512, 122880>>>(d, sink, src, iters); }, d, 256, fl * 12 * 8, iters);
    run("gemm loop (LDS reads 2 ahead), 8w", [&] { k_gemm_loop<2><<<256, 512, 122880>>>(d, sink, iters); }, d, 256, fl * 12 * 8, iters);
    run("gemm loop (reads after use), 8w", [&] { k_gemm_loop<0><<<256, 512, 122880>>>(d, sink, iters); }, d, 256, fl * 12 * 8, iters);
    return 0;
}
