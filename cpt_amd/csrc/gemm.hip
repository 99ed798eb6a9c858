// MFMA GEMM for the dense projections of the CPT hot path:
//   out[M][N] = epilogue( A[M][K] . W[N][K]^T + bias[N] (+ resid[M][N]) )
// W is in nn.Linear layout (out_features x in_features, row-major), so both operands are
// K-contiguous and feed MFMA fragments with 16-byte reads.  Replaces the torch Linear calls at
// /root/reference/Oscar/oscar/modeling/modeling_bert.py:38-40 (Q,K,V fused into N=3H), :85
// (BertSelfOutput.dense), :144 (BertIntermediate.dense + gelu), :145 (BertOutput.dense), :261
// (img_embedding) and the BertLMPredictionHead / BertPooler denses (modeling_rec.py:143,
// modeling_bert.py:275).
//
// gfx950 design: 128x128 tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave = 2x2
// MFMA 32x32 blocks), K-tile of 128 bytes per row (64 bf16 / 32 f32) staged through LDS with a
// 16-row XOR swizzle so ds_read_b128 fragment reads are bank-conflict free, register-staged
// double buffering (next tile's global loads are issued before the current tile's MFMAs and
// written to the other LDS buffer afterwards: one barrier per K-tile).
//   T = bf16 : v_mfma_f32_32x32x16_bf16, fp32 accumulate
//   T = f32  : v_mfma_f32_32x32x2_f32 (exact fp32; parity mode)
#include "common.h"
#include "kernels.h"

namespace cpt {

constexpr int BM = 128, BN = 128, ROWB = 128;  // ROWB = bytes per tile row
constexpr int GEMM_THREADS = 256;

__device__ __forceinline__ int lds_off(int row, int chunk) {
    // 128-byte rows, two rows per 256-byte bank row: conflict-free when the 16 lanes of a
    // ds_read_b128 group touch rows that are distinct mod 16.
    return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4);
}

template <typename T, int EPI, typename OT>
__global__ __launch_bounds__(GEMM_THREADS) void gemm_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
    const float* __restrict__ bias, const float* __restrict__ resid, int ldr,
    OT* __restrict__ out, int ldo, int M, int N, int K) {
    typedef typename FragOf<T>::type frag_t;
    constexpr int CE = Chunk<T>::N;          // elements per 16-byte chunk
    constexpr int BK = ROWB / (int)sizeof(T);  // elements per K-tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][BM * ROWB];  // [buf][A|W]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    // staging map: 1024 chunks per operand tile, 4 per thread
    const int sc = tid & 7;       // chunk within the row
    const int sr = tid >> 3;      // row 0..31 (+32*i)
    uint4 ra[4], rw[4];

    auto gload = [&](int k0) {
        const int kc = k0 + sc * CE;
        const bool kok = kc < K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = sr + 32 * i;
            const int gm = m0 + r, gn = n0 + r;
            ra[i] = (kok && gm < M) ? *reinterpret_cast<const uint4*>(A + (size_t)gm * lda + kc)
                                    : make_uint4(0, 0, 0, 0);
            rw[i] = (kok && gn < N) ? *reinterpret_cast<const uint4*>(W + (size_t)gn * ldw + kc)
                                    : make_uint4(0, 0, 0, 0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = sr + 32 * i;
            *reinterpret_cast<uint4*>(&smem[buf][0][lds_off(r, sc)]) = ra[i];
            *reinterpret_cast<uint4*>(&smem[buf][1][lds_off(r, sc)]) = rw[i];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = (K + BK - 1) / BK;
    gload(0);
    lstore(0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) gload((t + 1) * BK);
        const unsigned char* sa = smem[buf][0];
        const unsigned char* sw = smem[buf][1];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const frag_t*>(sa + lds_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
                fb[i] = *reinterpret_cast<const frag_t*>(sw + lds_off(wn * 64 + i * 32 + fr, ks * 2 + fh));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk(acc[i][j], fa[i], fb[j]);
        }
        if (t + 1 < nt) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + acc_col(lane);
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (row >= M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == CPT_EPI_GELU) v = gelu_erf(v);
                if (EPI == CPT_EPI_TANH) v = tanhf(v);
                if (EPI == CPT_EPI_RESID) v += resid[(size_t)row * ldr + col];
                out[(size_t)row * ldo + col] = from_f32<OT>(v);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Fast path (K a multiple of the K-tile): operands go HBM/L2 -> LDS directly with
// global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass); the XOR swizzle is applied to
// the per-lane SOURCE address because the LDS destination of an LDS-DMA is lane-linear.
// Out-of-range rows are clamped to a valid row (their results are never stored), so the loop has
// no exec-mask branches.  Workgroups are numbered XCD-first and then in groups of GROUP_M row
// tiles so that the ~64 workgroups resident on one XCD share A/W panels through that XCD's L2.
// ---------------------------------------------------------------------------------------------
constexpr int GROUP_M = 8;

template <int TBM>
__device__ __forceinline__ void tile_of_block(int M, int N, int& m0, int& n0) {
    const int tm = (M + TBM - 1) / TBM, tn = (N + BN - 1) / BN;
    const int nwg = tm * tn, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);   // bijective
    const int per_group = GROUP_M * tn;
    const int g = lid / per_group, first_m = g * GROUP_M;
    const int gsz = min(tm - first_m, GROUP_M);
    const int in_g = lid - g * per_group;
    m0 = (first_m + in_g % gsz) * TBM;
    n0 = (in_g / gsz) * BN;
}

template <typename T, int EPI, typename OT, int TBM>
__global__ __launch_bounds__(TBM * 2) void gemm_glds_kernel(
    const T* __restrict__ A, int lda, const T* __restrict__ W, int ldw,
    const float* __restrict__ bias, const float* __restrict__ resid, int ldr,
    OT* __restrict__ out, int ldo, int M, int N, int K) {
    typedef typename FragOf<T>::type frag_t;
    constexpr int CE = Chunk<T>::N;
    constexpr int BK = ROWB / (int)sizeof(T);
    constexpr int NW = TBM / 32;                 // waves: (TBM/64) x 2, 64x64 outputs each
    constexpr int ROWS = TBM + BN;               // A rows then W rows in one LDS image
    constexpr int GROUPS = ROWS / 8 / NW;        // 1 KiB LDS-DMA pieces per wave per K-tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][ROWS * ROWB];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int m0, n0;
    tile_of_block<TBM>(M, N, m0, n0);

    // per-lane source pointers for this wave's pieces (advance by BK elements per K-tile)
    const T* src[GROUPS];
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
        const int g = i * NW + wave;
        const int r = g * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);         // logical chunk stored at this lane's slot
        if (g * 8 < TBM) src[i] = A + (size_t)min(m0 + r, M - 1) * lda + c * CE;
        else             src[i] = W + (size_t)min(n0 + r - TBM, N - 1) * ldw + c * CE;
    }
    auto stage = [&](int buf, int k0) {
#pragma unroll
        for (int i = 0; i < GROUPS; ++i) {
            const int g = i * NW + wave;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void*)(src[i] + k0),
                (__attribute__((address_space(3))) void*)(&smem[buf][g * 1024]), 16, 0, 0);
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nt = K / BK;
    stage(0, 0);
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    for (int t = 0; t < nt; ++t) {
        const int buf = t & 1;
        if (t + 1 < nt) stage(buf ^ 1, (t + 1) * BK);
        const unsigned char* sa = smem[buf];
        const unsigned char* sw = smem[buf] + TBM * ROWB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            frag_t fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                fa[i] = *reinterpret_cast<const frag_t*>(sa + lds_off(wm * 64 + i * 32 + fr, ks * 2 + fh));
                fb[i] = *reinterpret_cast<const frag_t*>(sw + lds_off(wn * 64 + i * 32 + fr, ks * 2 + fh));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) mfma_chunk(acc[i][j], fa[i], fb[j]);
        }
        __syncthreads();     // drains this wave's LDS-DMA (vmcnt(0)) and fences the buffer swap
    }

#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = n0 + wn * 64 + j * 32 + acc_col(lane);
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + i * 32 + acc_row(r, lane);
                if (row >= M) continue;
                float v = acc[i][j][r] + bv;
                if (EPI == CPT_EPI_GELU) v = gelu_erf(v);
                if (EPI == CPT_EPI_TANH) v = tanhf(v);
                if (EPI == CPT_EPI_RESID) v += resid[(size_t)row * ldr + col];
                out[(size_t)row * ldo + col] = from_f32<OT>(v);
            }
        }
    }
}

int g_gemm_variant = 1;      // 0: register-staged generic kernel only; 1: LDS-DMA 128x128; 2: LDS-DMA 256x128

template <typename T, int EPI, typename OT>
static void launch_fast(int variant, const T* A, int lda, const T* W, int ldw, const float* bias,
                        const float* resid, int ldr, OT* out, int ldo, int M, int N, int K, hipStream_t s) {
    if (variant == 2 && M >= 1024) {
        const int nwg = ((M + 255) / 256) * ((N + BN - 1) / BN);
        gemm_glds_kernel<T, EPI, OT, 256><<<dim3(nwg), dim3(512), 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
    } else {
        const int nwg = ((M + 127) / 128) * ((N + BN - 1) / BN);
        gemm_glds_kernel<T, EPI, OT, 128><<<dim3(nwg), dim3(256), 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
    }
}

template <typename T, typename OT>
static int launch_epi(int epi, const T* A, int lda, const T* W, int ldw, const float* bias,
                      const float* resid, int ldr, OT* out, int ldo, int M, int N, int K,
                      hipStream_t s) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(GEMM_THREADS);
    constexpr int BKE = ROWB / (int)sizeof(T);
    if (g_gemm_variant > 0 && K % BKE == 0) {
        switch (epi) {
            case CPT_EPI_NONE: launch_fast<T, CPT_EPI_NONE, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            case CPT_EPI_GELU: launch_fast<T, CPT_EPI_GELU, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            case CPT_EPI_TANH: launch_fast<T, CPT_EPI_TANH, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            case CPT_EPI_RESID:
                if (!resid) return CPT_ERR_SHAPE;
                launch_fast<T, CPT_EPI_RESID, OT>(g_gemm_variant, A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K, s); return CPT_OK;
            default: return CPT_ERR_SHAPE;
        }
    }
    switch (epi) {
        case CPT_EPI_NONE:
            gemm_kernel<T, CPT_EPI_NONE, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        case CPT_EPI_GELU:
            gemm_kernel<T, CPT_EPI_GELU, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        case CPT_EPI_TANH:
            gemm_kernel<T, CPT_EPI_TANH, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        case CPT_EPI_RESID:
            if (!resid) return CPT_ERR_SHAPE;
            gemm_kernel<T, CPT_EPI_RESID, OT><<<grid, block, 0, s>>>(A, lda, W, ldw, bias, resid, ldr, out, ldo, M, N, K);
            break;
        default:
            return CPT_ERR_SHAPE;
    }
    return CPT_OK;
}

int gemm(int dtype, int epi, const void* A, int lda, const void* W, int ldw, const float* bias,
         const float* resid, int ldr, void* out, int out_dtype, int ldo, int M, int N, int K,
         hipStream_t s) {
    if (M <= 0 || N <= 0 || K <= 0) return CPT_ERR_SHAPE;
    const int ce = dtype == CPT_BF16 ? 8 : 4;
    if (K % ce || lda % ce || ldw % ce) return CPT_ERR_ALIGN;
    if (((uintptr_t)A | (uintptr_t)W) & 15) return CPT_ERR_ALIGN;
    if (dtype == CPT_BF16) {
        if (out_dtype == CPT_BF16)
            return launch_epi<bf16, bf16>(epi, (const bf16*)A, lda, (const bf16*)W, ldw, bias, resid, ldr, (bf16*)out, ldo, M, N, K, s);
        return launch_epi<bf16, float>(epi, (const bf16*)A, lda, (const bf16*)W, ldw, bias, resid, ldr, (float*)out, ldo, M, N, K, s);
    }
    if (dtype == CPT_F32) {
        if (out_dtype != CPT_F32) return CPT_ERR_DTYPE;
        return launch_epi<float, float>(epi, (const float*)A, lda, (const float*)W, ldw, bias, resid, ldr, (float*)out, ldo, M, N, K, s);
    }
    return CPT_ERR_DTYPE;
}

void set_gemm_variant(int v) { g_gemm_variant = v; }

}  // namespace cpt
