// Fused self-attention core for head_dim 64 (Oscar-base 12x64, Oscar-large 16x64):
//   ctx = softmax(Q K^T / 8 + (1 - mask) * -10000) V          per (sequence, head)
// Replaces /root/reference/Oscar/oscar/modeling/modeling_bert.py:42-67 (transpose_for_scores,
// two batched matmuls, softmax, permute/merge) -- the (B,heads,L,L) score tensor never leaves
// the chip.  Dropout on the probabilities (:57) is identity in eval; the training path of this
// build runs with dropout disabled.
//
// gfx950 design: one 256-thread workgroup per (sequence, head, 128-query tile); the head's K
// tile [L][64] (XOR-swizzled) and V tile (bf16: row-major, read back transposed by ds_read_b64_tr_b16 in
// attn_core.h; fp32: V^T [64][L] + pad) are staged once in LDS; each of the 4
// waves owns 32 query rows and keeps its whole score strip S^T = K.Q^T in MFMA accumulators
// (sequence length <= 288, so no online softmax is needed).  Computing the TRANSPOSED scores puts
// one query row per lane: the softmax row reduction is lane-local plus one cross-half shuffle,
// and the probabilities already sit in the A-operand layout of the P.V MFMA if V's key order is
// permuted the same way -- no LDS round trip for P.
#include "common.h"
#include "attn_core.h"
#include "kernels.h"

namespace cpt {

constexpr int HD = 64;           // head dim
constexpr int ATT_THREADS = 256;
constexpr int VP16 = ATT_VP16;
constexpr float LOG2E = ATT_LOG2E;

template <typename T> __device__ __forceinline__ int k_off(int row, int chunk);
// K tile rows are 64 elements: 128 B (bf16, 8 chunks) or 256 B (f32, 16 chunks)
template <> __device__ __forceinline__ int k_off<bf16>(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
template <> __device__ __forceinline__ int k_off<float>(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }

// key index held by accumulator register r of half-wave h inside a 32-key block
__device__ __forceinline__ int key_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// LEAN (round 4): the inference instantiation -- no dropout, 2-D mask, no saved probabilities.  One kernel for everything carried the
// Philox mask code and the per-score 3-D mask path behind run-time branches and spilled 72 / 168 / 234 SGPRs at NKB = 4 / 7 / 9.
template <typename T, int NKB, int LEAN>      // 1: inference; 2: training without the rare extras (dropout on, 2-D mask, no saved probabilities); 0: everything
__global__ __launch_bounds__(ATT_THREADS, NKB >= 7 ? 2 : 1) void attention_kernel(
    const T* __restrict__ qkv, const int64_t* __restrict__ attn_mask, T* __restrict__ ctx,
    T* __restrict__ probs_arg, int B, int L, int heads, DropSpec dr_arg, int mask3_arg, int ctx_panel, int remap, float* __restrict__ stats_arg) {
    const DropSpec dr = LEAN == 1 ? DropSpec{} : dr_arg;
    const int mask3 = LEAN ? 0 : mask3_arg;
    T* __restrict__ probs = LEAN ? nullptr : probs_arg;
    float* __restrict__ stats = LEAN == 1 ? nullptr : stats_arg;      // training forward (bf16): [B * heads][L][2] softmax statistics of every query row
    // remap (round 3, sequences of several 128-query tiles): a 1-D grid whose workgroups id, id + 8, .. (same XCD -- workgroups go to
    // the XCDs round robin -- dispatched back to back) are the query tiles of ONE (sequence, head), so the K / V rows the tiles share are
    // fetched from memory once and hit that XCD's L2 for the other tiles.  With the (pair, tile) grid the tiles of a pair ran 3072
    // workgroups apart: at B = 256, L = 210 the launch read K / V twice from HBM (the 248 MB QKV tensor does not stay in the Infinity
    // Cache) and ran at the rate of that traffic, 413 MB in 157 us.
    // ctx_panel (bf16 inference, round 3): ctx leaves in the fragment-major panel layout of the attn-out producer (gemm_prod.hip)
    // mask3: attn_mask is [B][L][L] (one row per query, modeling_bert.py:215-216) instead of [B][L]: the per-key LDS vector
    // then only marks the padding keys and every lane adds its own query's row from global memory
    typedef typename FragOf<T>::type frag_t;
    constexpr int CE = Chunk<T>::N;
    constexpr int NC = HD / CE;                       // chunks per K row
    constexpr int LP = NKB * 32;                      // padded key count
    constexpr bool LPT = sizeof(T) == 2;              // bf16 path: V stays row-major, read with the hardware transpose
    constexpr int VPAD = 16;                          // f32: bytes, makes V^T column reads conflict-free
    constexpr int VROW = LP * (int)sizeof(T) + VPAD;  // f32: bytes per V^T row
    constexpr int KS = NC / 2;                        // MFMA chunk steps over head dim
    constexpr bool VSWZ = LPT && NKB > 7;             // L > 224: swizzled 128-byte V rows (73.7 + 1.2 KB: two workgroups per CU) instead of the padded pitch
    constexpr int V_BYTES = LPT ? LP * (VSWZ ? 128 : VP16) : HD * VROW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sK = smem;                              // LP rows
    unsigned char* sV = smem + LP * HD * sizeof(T);        // bf16: LP rows of VP16 bytes; f32: 64 rows of VROW bytes (V^T)
    float* sMask = reinterpret_cast<float*>(sV + V_BYTES); // LP floats

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bh = blockIdx.x, qt = blockIdx.y;
    if (remap) {
        const int nqt = (L + 127) / 128, id = blockIdx.x, slot = id >> 3;
        qt = slot % nqt;
        bh = (slot / nqt) * 8 + (id & 7);
        if (bh >= B * heads) return;          // (pairs are padded to a multiple of 8; uniform per workgroup, ahead of the barrier)
    }
    const int b = bh / heads, h = bh % heads;
    const int q0 = qt * 128 + wave * 32;
    const int H = heads * HD;
    const size_t ldq = (size_t)3 * H;
    const T* base = qkv + (size_t)b * L * ldq + h * HD;

    // ---- all global loads first (Q fragments, K/V chunks, mask), then the LDS writes: one memory round trip ----
    // Q fragments straight from global: lane = query row (l&31), chunks 2*ks + (l>>5)
    const int fr = lane & 31, fh = lane >> 5;
    const int q = q0 + fr;
    frag_t fq[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        uint4 t = make_uint4(0, 0, 0, 0);
        if (q < L) t = *reinterpret_cast<const uint4*>(base + (size_t)q * ldq + (2 * ks + fh) * CE);
        fq[ks] = *reinterpret_cast<frag_t*>(&t);
    }
    constexpr int NLD = (LP * NC + ATT_THREADS - 1) / ATT_THREADS;
    uint4 kreg[NLD], vreg[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * ATT_THREADS, key = idx / NC, c = idx % NC;
        kreg[i] = make_uint4(0, 0, 0, 0);
        vreg[i] = make_uint4(0, 0, 0, 0);
        if (idx < LP * NC && key < L) {
            kreg[i] = *reinterpret_cast<const uint4*>(base + (size_t)key * ldq + H + c * CE);
            vreg[i] = *reinterpret_cast<const uint4*>(base + (size_t)key * ldq + 2 * H + c * CE);
        }
    }
    for (int key = tid; key < LP; key += ATT_THREADS) {
        float mv = -INFINITY;                          // padding keys beyond L: excluded outright
        if (key < L) mv = (attn_mask && !mask3) ? (1.0f - (float)attn_mask[(size_t)b * L + key]) * -10000.0f : 0.f;
        sMask[key] = LPT ? mv * LOG2E : mv;            // bf16 path: softmax in base 2 (one v_exp_f32 per score)
    }
    // K rows XOR-swizzled; V row-major (bf16, transpose-read later) or transposed element-wise (f32)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + i * ATT_THREADS, key = idx / NC, c = idx % NC;
        if (idx < LP * NC) {
            *reinterpret_cast<uint4*>(sK + k_off<T>(key, c)) = kreg[i];
            if constexpr (LPT) {
                if constexpr (VSWZ) *reinterpret_cast<uint4*>(sV + att_voff_swz(key, c)) = vreg[i];
                else *reinterpret_cast<uint4*>(sV + key * VP16 + c * 16) = vreg[i];
            } else {
                const T* ve = reinterpret_cast<const T*>(&vreg[i]);
#pragma unroll
                for (int j = 0; j < CE; ++j)
                    *reinterpret_cast<T*>(sV + (c * CE + j) * VROW + key * sizeof(T)) = ve[j];
            }
        }
    }
    __syncthreads();
    if (q0 >= L) return;   // whole wave has no query rows (uniform per wave)
    const int64_t* mrow = (mask3 && attn_mask) ? attn_mask + ((size_t)b * L + min(q, L - 1)) * L : nullptr;
    if constexpr (LPT) {     // bf16: the shared core (attn_core.h); everything below is the fp32 parity path
        T* crow = ctx + ((size_t)b * L + min(q, L - 1)) * H + h * HD;
        T* prow = probs ? probs + (((size_t)b * heads + h) * L + min(q, L - 1)) * L : nullptr;
        if (ctx_panel)
            attn_core_bf16<NKB, VSWZ, true>(fq, sK, sV, sMask, lane, q < L, nullptr, nullptr, L, dr, (uint32_t)bh, min(q, L - 1), mrow,
                                            (void*)ctx, (int)min((size_t)(((size_t)B * L + 31) & ~(size_t)31) * H * 2, (size_t)0x7fffffff), b * L + min(q, L - 1), h * 8, H >> 4);
        else
        attn_core_bf16<NKB, VSWZ>(fq, sK, sV, sMask, lane, q < L, crow, prow, L, dr, (uint32_t)bh, min(q, L - 1), mrow, nullptr, 0, 0, 0, 0,
                                  stats ? reinterpret_cast<float2*>(stats) + (size_t)bh * L + min(q, L - 1) : nullptr);
        return;
    }

    // ---- fp32: S^T = K . Q^T : accumulator rows = keys, column (lane&31) = query ----
    f32x16 st[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const frag_t fk = *reinterpret_cast<const frag_t*>(sK + k_off<T>(kb * 32 + fr, 2 * ks + fh));
            mfma_chunk(st[kb], fk, fq[ks]);
        }
    }

    // ---- softmax over keys: lane-local + one exchange with the other half-wave ----
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = st[kb][r] * 0.125f + sMask[kb * 32 + key_of(r, fh)];
            if (mrow) { const int key = kb * 32 + key_of(r, fh); if (key < L) s += (1.0f - (float)mrow[key]) * -10000.0f; }
            st[kb][r] = s;
            mx = fmaxf(mx, s);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = expf(st[kb][r] - mx);
            st[kb][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] *= inv;
    if (dr.thresh != 0) {   // training: dropout on the probabilities (modeling_bert.py:57), same mask function as the bf16 core
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bool keep[4];
                drop_attn_row4(dr, (uint32_t)bh, min(q, L - 1), kb * 8 + 2 * g + fh, keep);
#pragma unroll
                for (int j = 0; j < 4; ++j) st[kb][4 * g + j] = keep[j] ? st[kb][4 * g + j] * dr.scale : 0.f;
            }
    }

    if (probs && q < L) {   // [B][heads][L][L], saved for the backward pass only
        T* pr = probs + (((size_t)b * heads + h) * L + q) * L;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + key_of(r, fh);
                if (key < L) pr[key] = from_f32<T>(st[kb][r]);
            }
    }

    // ---- O = P . V : A operand = P (this lane's registers), B operand = V^T rows from LDS ----
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;

#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        {
            // one v_mfma_f32_32x32x2_f32 per accumulator register: keys key_of(r,0) / key_of(r,1)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const unsigned char* vr = sV + (db * 32 + fr) * VROW + (kb * 32 + 8 * g + 4 * fh) * 4;
                    const f32x4 vv = *reinterpret_cast<const f32x4*>(vr);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[kb][4 * g + j], vv[j], o[db], 0, 0, 0);
                }
            }
        }
    }

    // ---- store context rows (merge heads: column h*64 + d) ----
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int qq = q0 + acc_row(r, lane);
            if (qq < L) ctx[((size_t)b * L + qq) * H + h * HD + db * 32 + acc_col(lane)] = from_f32<T>(o[db][r]);
        }
}

CPT_SWITCH(int g_attn_remap, 1);       // cpt_set_tuning(21, v): 0 = the (pair, tile) grid of rounds 1-2 (A/B)
void set_attn_qt_all(int v) { CPT_SWITCH_SET(g_attn_remap = v); (void)v; }

template <typename T, int NKB>
static size_t att_lds_bytes() {
    constexpr int LP = NKB * 32;
    const size_t v_bytes = sizeof(T) == 2 ? (size_t)LP * (NKB > 7 ? 128 : VP16) : (size_t)HD * (LP * sizeof(T) + 16);
    return (size_t)LP * HD * sizeof(T) + v_bytes + (size_t)LP * sizeof(float);
}

template <typename T, int NKB>
static int att_launch(const void* qkv, const int64_t* mask, void* ctx, void* probs, int B, int L, int heads, const DropSpec& dr, hipStream_t s, int mask3, int ctx_panel, float* stats) {
    const size_t lds = att_lds_bytes<T, NKB>();
    const int lean = (!mask3 && !probs) ? ((dr.thresh == 0 && !stats) ? 1 : 2) : 0;
    auto kern = lean == 1 ? attention_kernel<T, NKB, 1> : (lean == 2 ? attention_kernel<T, NKB, 2> : attention_kernel<T, NKB, 0>);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
    }
    const int nqt = (L + 127) / 128;
    const int remap = (nqt > 1 && g_attn_remap) ? 1 : 0;
    dim3 grid(remap ? ((B * heads + 7) / 8 * 8) * nqt : B * heads, remap ? 1 : nqt), block(ATT_THREADS);
    kern<<<grid, block, lds, s>>>((const T*)qkv, mask, (T*)ctx, (T*)probs, B, L, heads, dr, mask3, ctx_panel, remap, stats);
    return CPT_OK;
}

// ---- sequences beyond 288 positions (round 5; inference) -----------------------------------------------------------------------------
// The reference takes whatever fits its position table plus the region slots (modeling_bert.py:244-269: up to 512 text positions + regions); no CPT
// configuration goes beyond L = 265, so this is a coverage kernel, not a fast one: one wave per query, no MFMA, nothing tiled.
//   pass 1: lane = key (64 keys per step): s = q . k / 8 + mask from the lane's own K row (vector loads of the whole row), scores parked in LDS, running maximum;
//   pass 2: p = exp(s - max), row sum;   pass 3: lane = head-dim column: ctx[d] = sum_j p_j v[j][d] / sum, V rows read coalesced, p_j broadcast from LDS.
// fp32 everywhere (libm expf), inputs / outputs in T.  2-D and 3-D masks; no dropout, no saved probabilities, row-major ctx.
constexpr int ATT_LONG_MAX = 1024;
template <typename T>
__global__ __launch_bounds__(ATT_THREADS) void attention_long_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ attn_mask, T* __restrict__ ctx,
                                                                      int B, int L, int heads, int mask3) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* sS = reinterpret_cast<float*>(smem) + (size_t)wave * (L + 64);          // this wave's scores, then probabilities
    float* sQ = sS + L;                                                           // its query row (64 floats)
    const int bh = blockIdx.x, b = bh / heads, h = bh % heads;
    const int q = blockIdx.y * 4 + wave;
    if (q >= L) return;
    const int H = heads * HD;
    const size_t ldq = (size_t)3 * H;
    const T* base = qkv + (size_t)b * L * ldq + h * HD;
    sQ[lane] = to_f32(base[(size_t)q * ldq + lane]);
    float mx = -INFINITY;
    for (int j0 = 0; j0 < L; j0 += 64) {
        const int j = j0 + lane;
        float sc = -INFINITY;
        if (j < L) {
            const T* kr = base + (size_t)j * ldq + H;
            float acc = 0.f;
            constexpr int CE = Chunk<T>::N;
#pragma unroll
            for (int c = 0; c < HD / CE; ++c) {
                const uint4 t = *reinterpret_cast<const uint4*>(kr + c * CE);
                const T* e = reinterpret_cast<const T*>(&t);
#pragma unroll
                for (int i = 0; i < CE; ++i) acc = fmaf(sQ[c * CE + i], to_f32(e[i]), acc);
            }
            float mv = 0.f;
            if (attn_mask) mv = (1.0f - (float)(mask3 ? attn_mask[((size_t)b * L + q) * L + j] : attn_mask[(size_t)b * L + j])) * -10000.0f;
            sc = acc * 0.125f + mv;
            sS[j] = sc;
        }
        mx = fmaxf(mx, sc);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) {
        const float pj = expf(sS[j] - mx);
        sS[j] = pj;
        sum += pj;
    }
    sum = wave_sum(sum);
    // (the wave's own LDS writes above are visible to its reads below: LDS operations of one wave execute in order)
    float o = 0.f;
    const T* vcol = base + 2 * H + lane;
    for (int j = 0; j < L; ++j) o = fmaf(sS[j], to_f32(vcol[(size_t)j * ldq]), o);
    ctx[((size_t)b * L + q) * H + h * HD + lane] = from_f32<T>(o / sum);
}

template <typename T>
static int att_long(const void* qkv, const int64_t* mask, void* ctx, int B, int L, int heads, hipStream_t s, int mask3) {
    const size_t lds = (size_t)4 * (L + 64) * sizeof(float);
    attention_long_kernel<T><<<dim3(B * heads, (L + 3) / 4), dim3(ATT_THREADS), lds, s>>>((const T*)qkv, mask, (T*)ctx, B, L, heads, mask3);
    return CPT_OK;
}

int attention_max_len(int inference) { return inference ? ATT_LONG_MAX : 288; }

template <typename T>
static int att_dispatch(const void* qkv, const int64_t* mask, void* ctx, void* probs, int B, int L, int heads, const DropSpec& dr, hipStream_t s, int mask3, int ctx_panel, float* stats) {
    if (L > 288) {      // beyond the register-resident score strip: the coverage kernel (inference only: no dropout, no saved probabilities, row-major ctx)
        if (L > ATT_LONG_MAX || probs || dr.thresh != 0 || ctx_panel || stats) return CPT_ERR_SHAPE;
        return att_long<T>(qkv, mask, ctx, B, L, heads, s, mask3);
    }
    if (L <= 32) return att_launch<T, 1>(qkv, mask, ctx, probs, B, L, heads, dr, s, mask3, ctx_panel, stats);
    if (L <= 128) return att_launch<T, 4>(qkv, mask, ctx, probs, B, L, heads, dr, s, mask3, ctx_panel, stats);
    if (L <= 224) return att_launch<T, 7>(qkv, mask, ctx, probs, B, L, heads, dr, s, mask3, ctx_panel, stats);
    return att_launch<T, 9>(qkv, mask, ctx, probs, B, L, heads, dr, s, mask3, ctx_panel, stats);
}

int attention(int dtype, const void* qkv, const int64_t* attn_mask, void* ctx, void* probs, int B, int L, int heads, hipStream_t s,
              const DropSpec* drop, int mask_3d, int ctx_panel, float* stats) {
    if (B <= 0 || L <= 0 || heads <= 0) return CPT_ERR_SHAPE;
    if (!qkv || !ctx) return CPT_ERR_NULL;
    if (stats && (dtype != CPT_BF16 || mask_3d || probs)) return CPT_ERR_SHAPE;      // statistics export: the bf16 training forward with a per-key mask
    if (ctx_panel && (dtype != CPT_BF16 || probs || ((uintptr_t)ctx & 15) || ((size_t)B * L) % 32)) return CPT_ERR_SHAPE;     // panel output: bf16 inference, whole 32-row blocks
    const DropSpec dr = drop ? *drop : DropSpec{};
    if (dtype == CPT_BF16) return att_dispatch<bf16>(qkv, attn_mask, ctx, probs, B, L, heads, dr, s, mask_3d, ctx_panel, stats);
    if (dtype == CPT_F32) return att_dispatch<float>(qkv, attn_mask, ctx, probs, B, L, heads, dr, s, mask_3d, 0, nullptr);
    return CPT_ERR_DTYPE;
}

// ---- bf16x3 parity mode (round 4): attention on MFMA with split operands --------------------------------------------------------------
// The parity mode ran its attention on the fp32 MFMA (1/16 of the bf16 rate: 52 us per layer at B = 64, 9 % of its step).  Here every
// operand is split like the mode's GEMM operands, x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and each product is the three-term
// sum hi.hi + hi.lo + lo.hi in fp32 accumulators (error ~2^-16 relative per product, the dropped lo.lo term):
//   S^T = K Q^T      from K hi / lo tiles in LDS and Q hi / lo fragments in registers (fp32 qkv rows split on load),
//   softmax in fp32  (base 2, as the bf16 core), probabilities split in registers,
//   O^T = V^T P^T    from V hi / lo tiles (row-major, transpose-read).
// Output: ctx as fp32 [M][H], or (out_split) directly as the split copy [M][hi | hi | lo] (ld 3H) that the attention-output GEMM of the
// mode reads -- the stand-alone cpt_split3 pass over ctx is gone.  2-D masks, L <= 224 (K + V hi / lo tiles: 112 KB at L = 224).
__device__ __forceinline__ void split_chunk8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi[e] = (bf16)a[e]; lo[e] = (bf16)(a[e] - (float)hi[e]);
        hi[4 + e] = (bf16)b[e]; lo[4 + e] = (bf16)(b[e] - (float)hi[4 + e]);
    }
}

template <int NKB>
__global__ __launch_bounds__(ATT_THREADS, NKB <= 4 ? 2 : 1) void attention_x3_kernel(
    const float* __restrict__ qkv, const int64_t* __restrict__ attn_mask, float* __restrict__ ctx, bf16* __restrict__ ctx_split,
    int B, int L, int heads, DropSpec dr) {
    constexpr int LP = NKB * 32;
    constexpr int K_BYTES = LP * 128, V_BYTES = LP * 128;       // V rows swizzled (att_voff_swz): 64.5 KB at L <= 128 = two workgroups per CU
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sKh = smem;
    unsigned char* sKl = sKh + K_BYTES;
    unsigned char* sVh = sKl + K_BYTES;
    unsigned char* sVl = sVh + V_BYTES;
    float* sMask = reinterpret_cast<float*>(sVl + V_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bh = blockIdx.x, qt = blockIdx.y;
    const int b = bh / heads, h = bh % heads;
    const int q0 = qt * 128 + wave * 32;
    const int H = heads * HD;
    const size_t ldq = (size_t)3 * H;
    const float* base = qkv + (size_t)b * L * ldq + h * HD;
    const int fr = lane & 31, fh = lane >> 5;
    const int q = q0 + fr;

    // Q fragments: lane = query row, chunk 2 ks + fh of its 64 head-dim columns (8 floats = two 16-byte loads), split in registers
    bf16x8 fqh[4], fql[4];
    {
        f32x4 t[4][2];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            t[ks][0] = f32x4{0.f, 0.f, 0.f, 0.f}; t[ks][1] = t[ks][0];
            if (q < L) {
                const float* src = base + (size_t)q * ldq + (2 * ks + fh) * 8;
                t[ks][0] = *reinterpret_cast<const f32x4*>(src);
                t[ks][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) split_chunk8(t[ks][0], t[ks][1], fqh[ks], fql[ks]);
    }
    // K / V rows: 8 chunks of 8 floats per key; all loads of a thread first, then the split + LDS writes
    constexpr int NLD = (LP * 8 + ATT_THREADS - 1) / ATT_THREADS;
#pragma unroll
    for (int half = 0; half < 2; ++half) {            // K, then V (keeps the register footprint at one operand's loads)
        f32x4 r[NLD][2];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * ATT_THREADS, key = idx >> 3, c = idx & 7;
            r[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; r[i][1] = r[i][0];
            if (idx < LP * 8 && key < L) {
                const float* src = base + (size_t)key * ldq + (half + 1) * H + c * 8;
                r[i][0] = *reinterpret_cast<const f32x4*>(src);
                r[i][1] = *reinterpret_cast<const f32x4*>(src + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = tid + i * ATT_THREADS, key = idx >> 3, c = idx & 7;
            if (idx < LP * 8) {
                bf16x8 hi, lo;
                split_chunk8(r[i][0], r[i][1], hi, lo);
                if (half == 0) {
                    *reinterpret_cast<bf16x8*>(sKh + k_off<bf16>(key, c)) = hi;
                    *reinterpret_cast<bf16x8*>(sKl + k_off<bf16>(key, c)) = lo;
                } else {
                    *reinterpret_cast<bf16x8*>(sVh + att_voff_swz(key, c)) = hi;
                    *reinterpret_cast<bf16x8*>(sVl + att_voff_swz(key, c)) = lo;
                }
            }
        }
    }
    for (int key = tid; key < LP; key += ATT_THREADS) {
        float mv = -INFINITY;
        if (key < L) mv = attn_mask ? (1.0f - (float)attn_mask[(size_t)b * L + key]) * -10000.0f : 0.f;
        sMask[key] = mv * LOG2E;
    }
    __syncthreads();
    if (q0 >= L) return;

    // S^T = K . Q^T, three terms (small ones first)
    f32x16 st[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 fkh = *reinterpret_cast<const bf16x8*>(sKh + k_off<bf16>(kb * 32 + fr, 2 * ks + fh));
            const bf16x8 fkl = *reinterpret_cast<const bf16x8*>(sKl + k_off<bf16>(kb * 32 + fr, 2 * ks + fh));
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fkl, fqh[ks], st[kb], 0, 0, 0);
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fkh, fql[ks], st[kb], 0, 0, 0);
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fkh, fqh[ks], st[kb], 0, 0, 0);
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sc = st[kb][r] * (0.125f * LOG2E) + sMask[kb * 32 + key_of(r, fh)];
            st[kb][r] = sc;
            mx = fmaxf(mx, sc);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(st[kb][r] - mx);      // v_exp_f32: 1 ulp, three decimal orders below the 2^-16 of the split products
            st[kb][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] *= inv;
    // training (bf16x3 mode's forward): dropout on the probabilities, the same stream the backward kernels regenerate (attn_core.h)
    if (dr.thresh != 0) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bool keep[4];
                drop_attn_row4(dr, (uint32_t)bh, q, kb * 8 + 2 * g + fh, keep);
#pragma unroll
                for (int j = 0; j < 4; ++j) st[kb][4 * g + j] = keep[j] ? st[kb][4 * g + j] * dr.scale : 0.f;
            }
    }

    // O^T = V^T . P^T, three terms
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 ph, pl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float p = st[kb][8 * s2 + j]; ph[j] = (bf16)p; pl[j] = (bf16)(p - (float)ph[j]); }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int vrow = kb * 32 + 16 * s2 + 4 * fh + ((lane & 15) >> 2), vcol = db * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                const int off = att_voff_swz(vrow, vcol >> 3) + (vcol & 7) * 2, off8 = att_voff_swz(vrow + 8, vcol >> 3) + (vcol & 7) * 2;
                bf16x8 vh, vl;
                {
                    const bf16x4 a0 = lds_read_tr16(sVh + off), a1 = lds_read_tr16(sVh + off8);
                    const bf16x4 b0 = lds_read_tr16(sVl + off), b1 = lds_read_tr16(sVl + off8);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { vh[j] = a0[j]; vh[4 + j] = a1[j]; vl[j] = b0[j]; vl[4 + j] = b1[j]; }
                }
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vl, ph, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, pl, o[db], 0, 0, 0);
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o[db], 0, 0, 0);
            }
        }
    }
    // O^T accumulators: register r <-> head-dim column db*32 + 8*(r>>2) + 4*fh + (r&3), lane&31 <-> query
    if (q < L) {
        const size_t row = (size_t)b * L + q;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = h * HD + 4 * fh + db * 32 + 8 * g;
                const f32x4 v = {o[db][4 * g], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]};
                if (ctx_split) {
                    bf16x4 hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { hi[e] = (bf16)v[e]; lo[e] = (bf16)(v[e] - (float)hi[e]); }
                    bf16* dst = ctx_split + row * 3 * H + col;
                    *reinterpret_cast<bf16x4*>(dst) = hi;
                    *reinterpret_cast<bf16x4*>(dst + H) = hi;
                    *reinterpret_cast<bf16x4*>(dst + 2 * H) = lo;
                } else {
                    *reinterpret_cast<f32x4*>(ctx + row * H + col) = v;
                }
            }
    }
}

template <int NKB>
static int att_x3_launch(const float* qkv, const int64_t* mask, float* ctx, bf16* ctx_split, int B, int L, int heads, hipStream_t s, const DropSpec& dr) {
    constexpr int LP = NKB * 32;
    const size_t lds = (size_t)4 * LP * 128 + (size_t)LP * sizeof(float);
    auto kern = attention_x3_kernel<NKB>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
    }
    kern<<<dim3(B * heads, (L + 127) / 128), dim3(ATT_THREADS), lds, s>>>(qkv, mask, ctx, ctx_split, B, L, heads, dr);
    return CPT_OK;
}

int attention_x3_supported(int L) { return L > 0 && L <= 288; }

int attention_x3(const float* qkv, const int64_t* attn_mask, float* ctx, void* ctx_split, int B, int L, int heads, hipStream_t s, const DropSpec* drop) {
    const DropSpec dr = drop ? *drop : DropSpec{};
    if (B <= 0 || heads <= 0 || !attention_x3_supported(L)) return CPT_ERR_SHAPE;
    if (!qkv || (!ctx && !ctx_split)) return CPT_ERR_NULL;
    if (((uintptr_t)qkv | (uintptr_t)ctx) & 15 || ((uintptr_t)ctx_split & 7)) return CPT_ERR_ALIGN;
    if (L <= 32) return att_x3_launch<1>(qkv, attn_mask, ctx, (bf16*)ctx_split, B, L, heads, s, dr);
    if (L <= 128) return att_x3_launch<4>(qkv, attn_mask, ctx, (bf16*)ctx_split, B, L, heads, s, dr);
    if (L <= 224) return att_x3_launch<7>(qkv, attn_mask, ctx, (bf16*)ctx_split, B, L, heads, s, dr);
    return att_x3_launch<9>(qkv, attn_mask, ctx, (bf16*)ctx_split, B, L, heads, s, dr);      // Oscar-large VCR, L = 265: 148.6 KB of LDS
}

}  // namespace cpt
