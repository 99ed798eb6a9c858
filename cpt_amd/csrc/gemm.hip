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

template <typename T, typename OT>
static int launch_epi(int epi, const T* A, int lda, const T* W, int ldw, const float* bias,
                      const float* resid, int ldr, OT* out, int ldo, int M, int N, int K,
                      hipStream_t s) {
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(GEMM_THREADS);
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

}  // namespace cpt
