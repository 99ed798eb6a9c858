// bf16 self-attention core shared by the stand-alone attention kernel (attention.hip) and the fused
// QKV-projection + attention kernel (gemm.hip, CPT_EPI_ATTN*): one wave, 32 queries, K / V / mask tiles in LDS.
//   ctx[q][:] = softmax(Q K^T / 8 + mask) V        (modeling_bert.py:42-67 of the reference, per sequence and head)
#pragma once
#include "common.h"
#include "dropout.h"

namespace cpt {

constexpr int ATT_HD = 64;        // head dim
// V row pitch in LDS: 64 d x 2 B + 64 B pad: the four key rows one ds_read_b64_tr_b16 lane group touches
// (32 B each, two groups per half-wave) land in four distinct 64-B bank quarters.
constexpr int ATT_VP16 = 192;
constexpr float ATT_LOG2E = 1.44269504088896340736f;

// K / Q tile rows are 64 bf16 = 128 B = 8 chunks of 16 B, XOR-swizzled so that fragment reads are conflict-free
__device__ __forceinline__ int att_koff16(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// V tile, swizzled form (VSWZ, used where the padded pitch does not leave room for two workgroups per CU: L > 224): 128-byte rows
// whose 16-byte chunks are XOR-ed with the row-pair index rotated by one bit -- rows r, r+1 sit in the two halves of a 256-byte
// bank row, rows r+2, r+3 in the other 64-byte half of each, so the four rows of a transpose read use four different quarters.
__device__ __forceinline__ int att_voff_swz(int row, int chunk) {
    const int p = (row >> 1) & 7;
    return row * 128 + ((chunk ^ (((p & 1) << 2) | (p >> 1))) << 4);
}

// ds_read_b64_tr_b16 (gfx950 LDS transpose read; lane mapping measured with tools/tr_probe.hip): within a 16-lane
// group lane i, slot j receives element (i & 3) of the 8 bytes addressed by lane 4*j + (i >> 2).  With lane s
// pointing at V[key0 + (s >> 2)][d0 + 4*(s & 3) ...] the group reads a row-major [4 keys][16 d] block and lane i
// gets V[key0 + 0..3][d0 + i]: four consecutive keys of ONE head-dim column, i.e. half an MFMA operand of V^T.
__device__ __forceinline__ bf16x4 lds_read_tr16(const unsigned char* p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    return *reinterpret_cast<const bf16x4*>(&v);
}

// key index held by accumulator register r of half-wave h inside a 32-key block
__device__ __forceinline__ int att_key_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// One wave: queries fq (lane = query lane&31, chunks 2*ks + (lane>>5) of its Q row), keys/values/mask in LDS:
//   sK   [NKB*32][64] bf16, rows swizzled with att_koff16
//   sV   [NKB*32] rows of ATT_VP16 bytes, row-major
//   sMask[NKB*32] additive mask already multiplied by log2(e); -inf for padding keys
// Writes ctx_row[0..63] (this lane's query, head slice) if `valid`; optionally the probabilities (training).
// PANEL (round 3): the context goes out in the fragment-major panel layout of the attn-out GEMM's A operand (common.h panel_unit,
// gemm_prod.hip) as 16-byte write-through stores: ctx_row is ignored, the destination is (pbase / pbytes = the whole ctx
// buffer, prow = global row of this lane's query, pc8 = first 8-column chunk of the head's slice, pk16 = hidden / 16).
template <int NKB, bool VSWZ = false, bool PANEL = false>
__device__ __forceinline__ void attn_core_bf16(const bf16x8 (&fq)[4], const unsigned char* sK, const unsigned char* sV,
                                               const float* sMask, int lane, bool valid, bf16* ctx_row,
                                               bf16* probs_row, int L, const DropSpec& dr = DropSpec{}, uint32_t bh = 0, int q = 0,
                                               const int64_t* mrow = nullptr,   // mrow: this lane's row of a 3-D attention mask (modeling_bert.py:215-216)
                                               void* pbase = nullptr, int pbytes = 0, int prow = 0, int pc8 = 0, int pk16 = 0,
                                               float2* stat_row = nullptr) {     // stat_row (training forward, round 6): this query's (row max in base 2, 1 / row sum) for the backward kernel
    const int fr = lane & 31, fh = lane >> 5;
    // S^T = K . Q^T : accumulator rows = keys, column (lane&31) = query
    f32x16 st[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 fk = *reinterpret_cast<const bf16x8*>(sK + att_koff16(kb * 32 + fr, 2 * ks + fh));
            st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fk, fq[ks], st[kb], 0, 0, 0);
        }
    }
    // softmax over keys in base 2: lane-local + one exchange with the other half-wave
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float s = st[kb][r] * (0.125f * ATT_LOG2E) + sMask[kb * 32 + att_key_of(r, fh)];
            if (mrow) { const int key = kb * 32 + att_key_of(r, fh); if (key < L) s += (1.0f - (float)mrow[key]) * (-10000.0f * ATT_LOG2E); }
            st[kb][r] = s;
            mx = fmaxf(mx, s);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(st[kb][r] - mx);
            st[kb][r] = p;
            sum += p;
        }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (stat_row && valid && fh == 0) *stat_row = float2{mx, inv};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[kb][r] *= inv;
    // training: dropout on the probabilities (modeling_bert.py:57); bh = sequence * heads + head, q = this lane's query
    // One Philox call yields the uniforms of TWO queries (2 x 4 block, dropout.h): the neighbour lanes q and q ^ 1 (same half-wave, same key
    // quads) each run every other call and hand the partner its half through a quad-permute (round 6: 8 calls per lane instead of 16).
    if (dr.thresh != 0) {
        const bool odd = lane & 1;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                uint32_t u[4];
                drop_attn_call(dr, bh, (uint32_t)q >> 2, (uint32_t)(kb * 8 + 2 * (2 * gp + (odd ? 1 : 0)) + fh), ((uint32_t)q >> 1) & 1, u);
                const uint32_t m0 = odd ? u[2] : u[0], m1 = odd ? u[3] : u[1];      // this query's row of this lane's call
                const uint32_t r0 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? u[0] : u[2]), 0xb1, 0xf, 0xf, true);   // the partner's call, this query's row
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_mov_dpp((int)(odd ? u[1] : u[3]), 0xb1, 0xf, 0xf, true);
                const uint32_t e0 = odd ? r0 : m0, e1 = odd ? r1 : m1;              // key quad g = 2 gp
                const uint32_t o0 = odd ? m0 : r0, o1 = odd ? m1 : r1;              // key quad g = 2 gp + 1
                const uint32_t w[2][2] = {{e0, e1}, {o0, o1}};
#pragma unroll
                for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool keep = ((w[gg][j >> 1] >> ((j & 1) * 16)) & 0xffffu) >= dr.thresh;
                        st[kb][4 * (2 * gp + gg) + j] = keep ? st[kb][4 * (2 * gp + gg) + j] * dr.scale : 0.f;
                    }
            }
    }

    if (probs_row && valid) {   // [B][heads][L][L], saved for the backward pass only
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + att_key_of(r, fh);
                if (key < L) probs_row[key] = (bf16)st[kb][r];
            }
    }

    // O^T = V^T . P^T: two MFMA k-steps of 16 keys per 32-key block; operand slot j of half h <-> key
    // 16*s + 4*h + (j&3) + 8*(j>>2).  A operand = V^T (row = head-dim column db*32 + fr) by two transpose reads of
    // 4 consecutive keys each, B operand = P^T (column = this lane's query): the probabilities already sit there.
    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 pa;
#pragma unroll
            for (int j = 0; j < 8; ++j) pa[j] = (bf16)st[kb][8 * s2 + j];
#pragma unroll
            for (int db = 0; db < 2; ++db) {
                const int vrow = kb * 32 + 16 * s2 + 4 * fh + ((lane & 15) >> 2), vcol = db * 32 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
                const unsigned char* vr = VSWZ ? sV + att_voff_swz(vrow, vcol >> 3) + (vcol & 7) * 2 : sV + vrow * ATT_VP16 + vcol * 2;
                const unsigned char* vr8 = VSWZ ? sV + att_voff_swz(vrow + 8, vcol >> 3) + (vcol & 7) * 2 : vr + 8 * ATT_VP16;
                const bf16x4 lo = lds_read_tr16(vr);
                const bf16x4 hi = lds_read_tr16(vr8);
                bf16x8 vb;
#pragma unroll
                for (int j = 0; j < 4; ++j) { vb[j] = lo[j]; vb[4 + j] = hi[j]; }
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vb, pa, o[db], 0, 0, 0);
            }
        }
    }
    // O^T accumulators: register r <-> head-dim column db*32 + 8*(r>>2) + 4*fh + (r&3), lane&31 <-> query:
    // four consecutive columns per register quad -> one 8-byte store
    if constexpr (PANEL) {
        // quads 2 gp and 2 gp + 1 of the two half-waves paired by v_permlane32_swap: lanes 0-31 hold columns [16 gp, 16 gp + 8) of
        // block db, lanes 32-63 the next 8 (gemm.hip "direct epilogue"); both halves of a lane pair are the same query
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const auto prs = __builtin_amdgcn_make_buffer_rsrc(pbase, 0, pbytes, 0x00020000);
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                u32x2 pk[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bf16x4 p4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) p4[e] = (bf16)o[db][4 * (2 * gp + h) + e];
                    pk[h] = __builtin_bit_cast(u32x2, p4);
                }
                const u32x2 s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                const u32x4_t w = {s0[0], s1[0], s0[1], s1[1]};
                if (valid) __builtin_amdgcn_raw_buffer_store_b128(w, prs, panel_unit(prow, pc8 + db * 4 + 2 * gp + fh, pk16) * 16u, 0, CPT_ST_AUX);
            }
    } else
    if (valid) {
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 pk;
#pragma unroll
                for (int e = 0; e < 4; ++e) pk[e] = (bf16)o[db][4 * g + e];
                *reinterpret_cast<bf16x4*>(ctx_row + 4 * fh + db * 32 + 8 * g) = pk;
            }
    }
}

}  // namespace cpt
