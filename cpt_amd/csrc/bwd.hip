// Backward-pass and optimizer kernels of the CPT few-shot step:
// what autograd runs for `loss.backward()` at /root/reference/Oscar/oscar/fewshot/refcoco_cpt.py:248
// and `optimizer.step()` (torch.optim.AdamW, :249,:343), restated as explicit kernels.
// The dense parts (dgrad / wgrad) reuse the MFMA GEMM of gemm.hip on transposed, zero-padded
// operands produced here; everything in this file is HBM-bound row / element work in fp32.
#include "common.h"
#include "attn_core.h"
#include "kernels.h"

namespace cpt {

// ---- transpose + cast: in[R][C] (ld ldi) -> out[C][ldo], columns R..ldo-1 zero -----------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void transpose_kernel(const TI* __restrict__ in, int ldi, TO* __restrict__ out,
                                                        int ldo, int R, int C) {
    __shared__ float tile[64][65];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;   // 64 x 4
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < R && c < C) ? to_f32(in[(size_t)r * ldi + c]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;      // out row = c, out col = r
        if (c < C && r < ldo) out[(size_t)c * ldo + r] = from_f32<TO>(tile[tx][ty + 4 * i]);
    }
}

// bf16 -> bf16 fast path: a 64 x 64 tile goes to LDS row-major with 16-byte accesses (pitch 192 B, the conflict-free
// pitch of the transpose read measured for attention's V), and comes back through ds_read_b64_tr_b16: two transpose
// reads give a lane 8 consecutive input rows of ONE column = 16 contiguous bytes of an output row.  The four 16-lane
// groups of a wave take adjacent row chunks of the same 16 output rows, so a store instruction writes 64-byte runs.
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16* __restrict__ in, int ldi, bf16* __restrict__ out,
                                                             int ldo, int R, int C) {
    constexpr int PITCH = 192;
    __shared__ __attribute__((aligned(16))) unsigned char tile[64 * PITCH];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256, rr = idx >> 3, ch = idx & 7;      // 64 rows x 8 chunks of 8 columns
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + rr < R && c0 + ch * 8 < C) v = *reinterpret_cast<const uint4*>(in + (size_t)(r0 + rr) * ldi + c0 + ch * 8);
        *reinterpret_cast<uint4*>(tile + rr * PITCH + ch * 16) = v;
    }
    __syncthreads();
    const int grp = lane >> 4, i16 = lane & 15;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int c16 = wave, rchunk = it * 4 + grp;                      // 16 output rows (columns c16*16 ..), rows 8*rchunk ..
        const unsigned char* p = tile + (rchunk * 8 + (i16 >> 2)) * PITCH + (c16 * 16 + 4 * (i16 & 3)) * 2;
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * PITCH));
        const int oc = c0 + c16 * 16 + i16, orow0 = r0 + rchunk * 8;      // output row oc, columns orow0 .. orow0+7
        if (oc < C && orow0 < ldo) {
            uint4 o;
            o.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
            o.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
            o.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
            o.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
            *reinterpret_cast<uint4*>(out + (size_t)oc * ldo + orow0) = o;
        }
    }
}

int transpose_cast(const void* in, int in_dtype, int ldi, void* out, int out_dtype, int ldo, int R, int C, hipStream_t s) {
    if (R <= 0 || C <= 0 || ldo < R || ldi < C) return CPT_ERR_SHAPE;
    if (in_dtype == CPT_BF16 && out_dtype == CPT_BF16 && ldi % 8 == 0 && ldo % 8 == 0 && C % 8 == 0 &&
        ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0) {
        dim3 grid((C + 63) / 64, (ldo + 63) / 64), block(256);
        transpose_bf16_kernel<<<grid, block, 0, s>>>((const bf16*)in, ldi, (bf16*)out, ldo, R, C);
        return CPT_OK;
    }
    dim3 grid((C + 63) / 64, (ldo + 63) / 64), block(256);
    if (in_dtype == CPT_F32 && out_dtype == CPT_F32) transpose_kernel<float, float><<<grid, block, 0, s>>>((const float*)in, ldi, (float*)out, ldo, R, C);
    else if (in_dtype == CPT_F32 && out_dtype == CPT_BF16) transpose_kernel<float, bf16><<<grid, block, 0, s>>>((const float*)in, ldi, (bf16*)out, ldo, R, C);
    else if (in_dtype == CPT_BF16 && out_dtype == CPT_BF16) transpose_kernel<bf16, bf16><<<grid, block, 0, s>>>((const bf16*)in, ldi, (bf16*)out, ldo, R, C);
    else return CPT_ERR_DTYPE;
    return CPT_OK;
}

// ---- column sums (bias gradients): out[c] += sum_r x[r][c] ----------------------------------------
// 256 threads = 32 column groups (16 bytes each: 8 bf16 / 4 f32) x 8 row lanes; a block covers CS_ROWS rows of its
// column span, reduces the 8 row lanes through LDS and issues one atomicAdd per column.
constexpr int CS_ROWS = 128;
template <typename TI>
__global__ __launch_bounds__(256) void colsum_kernel(const TI* __restrict__ x, int ld, float* __restrict__ out, int R, int C) {
    constexpr int VE = 16 / (int)sizeof(TI);              // elements per 16-byte load
    __shared__ float red[8][32 * VE + 1];
    const int cg = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int c0 = (blockIdx.x * 32 + cg) * VE;
    const int r0 = blockIdx.y * CS_ROWS, r1 = min(R, r0 + CS_ROWS);
    float acc[VE];
#pragma unroll
    for (int e = 0; e < VE; ++e) acc[e] = 0.f;
    if (c0 + VE <= C) {
        for (int r = r0 + ry; r < r1; r += 8) {
            const uint4 raw = *reinterpret_cast<const uint4*>(x + (size_t)r * ld + c0);
            const TI* v = reinterpret_cast<const TI*>(&raw);
#pragma unroll
            for (int e = 0; e < VE; ++e) acc[e] += to_f32(v[e]);
        }
    } else {
        for (int r = r0 + ry; r < r1; r += 8)
#pragma unroll
            for (int e = 0; e < VE; ++e)
                if (c0 + e < C) acc[e] += to_f32(x[(size_t)r * ld + c0 + e]);
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) red[ry][cg * VE + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 32 * VE; c += 256) {
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < 8; ++y) t += red[y][c];
        const int col = blockIdx.x * 32 * VE + c;
        if (col < C) atomicAdd(&out[col], t);
    }
}

// generic (rows not 16-byte aligned): one thread per column
template <typename TI>
__global__ __launch_bounds__(256) void colsum_scalar_kernel(const TI* __restrict__ x, int ld, float* __restrict__ out, int R, int C, int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(R, r0 + rows_per_block);
    float acc = 0.f;
    for (int r = r0; r < r1; ++r) acc += to_f32(x[(size_t)r * ld + c]);
    atomicAdd(&out[c], acc);
}

int colsum(const void* x, int dtype, int ld, float* out, int R, int C, hipStream_t s) {
    if (R <= 0 || C <= 0) return CPT_ERR_SHAPE;
    if (dtype != CPT_BF16 && dtype != CPT_F32) return CPT_ERR_DTYPE;
    const int es = dtype == CPT_BF16 ? 2 : 4, ve = 16 / es;
    if (((size_t)ld * es) % 16 == 0 && ((uintptr_t)x % 16) == 0) {
        dim3 grid((C + 32 * ve - 1) / (32 * ve), (R + CS_ROWS - 1) / CS_ROWS), block(256);
        if (dtype == CPT_BF16) colsum_kernel<bf16><<<grid, block, 0, s>>>((const bf16*)x, ld, out, R, C);
        else colsum_kernel<float><<<grid, block, 0, s>>>((const float*)x, ld, out, R, C);
        return CPT_OK;
    }
    const int rpb = 64;
    dim3 grid((C + 255) / 256, (R + rpb - 1) / rpb), block(256);
    if (dtype == CPT_BF16) colsum_scalar_kernel<bf16><<<grid, block, 0, s>>>((const bf16*)x, ld, out, R, C, rpb);
    else colsum_scalar_kernel<float><<<grid, block, 0, s>>>((const float*)x, ld, out, R, C, rpb);
    return CPT_OK;
}

// ---- GELU forward / backward on [n] elements: 16 bytes per thread --------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T* __restrict__ u, T* __restrict__ h, size_t n) {
    constexpr int VE = 16 / (int)sizeof(T);
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VE;
    if (i >= n) return;
    if (i + VE <= n) {
        const uint4 raw = *reinterpret_cast<const uint4*>(u + i);
        const T* v = reinterpret_cast<const T*>(&raw);
        uint4 o;
        T* w = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int e = 0; e < VE; ++e) w[e] = from_f32<T>(gelu_for<T>(to_f32(v[e])));
        *reinterpret_cast<uint4*>(h + i) = o;
    } else {
        for (size_t e = i; e < n; ++e) h[e] = from_f32<T>(gelu_for<T>(to_f32(u[e])));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dh, const T* __restrict__ u, T* __restrict__ du, size_t n) {
    constexpr int VE = 16 / (int)sizeof(T);
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * VE;
    if (i >= n) return;
    if (i + VE <= n) {
        const uint4 ra = *reinterpret_cast<const uint4*>(dh + i), rb = *reinterpret_cast<const uint4*>(u + i);
        const T* a = reinterpret_cast<const T*>(&ra);
        const T* b = reinterpret_cast<const T*>(&rb);
        uint4 o;
        T* w = reinterpret_cast<T*>(&o);
#pragma unroll
        for (int e = 0; e < VE; ++e) w[e] = from_f32<T>(to_f32(a[e]) * gelu_grad_for<T>(to_f32(b[e])));
        *reinterpret_cast<uint4*>(du + i) = o;
    } else {
        for (size_t e = i; e < n; ++e) du[e] = from_f32<T>(to_f32(dh[e]) * gelu_grad_for<T>(to_f32(u[e])));
    }
}
int gelu_fwd(const void* u, void* h, int dtype, size_t n, hipStream_t s) {
    const size_t ve = dtype == CPT_BF16 ? 8 : 4;
    dim3 grid((unsigned)(((n + ve - 1) / ve + 255) / 256)), block(256);
    if (dtype == CPT_BF16) gelu_fwd_kernel<bf16><<<grid, block, 0, s>>>((const bf16*)u, (bf16*)h, n);
    else gelu_fwd_kernel<float><<<grid, block, 0, s>>>((const float*)u, (float*)h, n);
    return CPT_OK;
}
int gelu_bwd(const void* dh, const void* u, void* du, int dtype, size_t n, hipStream_t s) {
    const size_t ve = dtype == CPT_BF16 ? 8 : 4;
    dim3 grid((unsigned)(((n + ve - 1) / ve + 255) / 256)), block(256);
    if (dtype == CPT_BF16) gelu_bwd_kernel<bf16><<<grid, block, 0, s>>>((const bf16*)dh, (const bf16*)u, (bf16*)du, n);
    else gelu_bwd_kernel<float><<<grid, block, 0, s>>>((const float*)dh, (const float*)u, (float*)du, n);
    return CPT_OK;
}

// ---- LayerNorm backward --------------------------------------------------------------------------
// y = (x - mean) * rstd * g + b over rows of x[R][H] (x = gelu(u) when GELU_IN).
// dy row r is read at (r / grp) * grp_stride + grp_off + r % grp (the residual-stream placement used
// in the forward).  dx (fp32) and dx_lp (optional) are written compact at row r; dg/db accumulated
// with one atomicAdd per column per workgroup.
constexpr int LNB_MAXV = 4;

// Column-sum jobs that ride at the end of another launch's grid (round 6): dst[c / seg][c % seg] += sum over `rows` of src[r * row_stride + c].
// The training backward leaves per-workgroup partial rows (LayerNorm gamma / beta / dense-bias sums, the FFN-up bias sums of the GELU-gradient
// GEMM) and lets the NEXT row pass add them up in spare workgroups instead of paying a launch per reduction (24 launches of 4.9 us per step).
__device__ __forceinline__ void col_job_run(const ColJob& jb, int j) {
    const int gx = (jb.cols + 255) / 256;
    const int c = (j % gx) * 256 + threadIdx.x, r0 = (j / gx) * 16;
    if (c >= jb.cols) return;
    float v[16];                                        // all loads in flight before the first add
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (r0 + k < jb.rows) ? jb.src[(size_t)(r0 + k) * jb.row_stride + c] : 0.f;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += v[k];
    const int sg = c / jb.seg;
    atomicAdd((sg == 0 ? jb.dst0 : (sg == 1 ? jb.dst1 : jb.dst2)) + (c - sg * jb.seg), a);
}
__device__ __forceinline__ void col_jobs_run(const ColJobs& jobs, int j) {
    for (int k = 0; k < jobs.n; ++k) {
        const int nj = ((jobs.j[k].cols + 255) / 256) * ((jobs.j[k].rows + 15) / 16);
        if (j < nj) { col_job_run(jobs.j[k], j); return; }
        j -= nj;
    }
}
int col_jobs_blocks(const ColJobs* jobs) {
    int n = 0;
    if (jobs) for (int k = 0; k < jobs->n; ++k) n += ((jobs->j[k].cols + 255) / 256) * ((jobs->j[k].rows + 15) / 16);
    return n;
}
__global__ __launch_bounds__(256) void col_jobs_kernel(ColJobs jobs) { col_jobs_run(jobs, blockIdx.x); }
int col_jobs_flush(const ColJobs& jobs, hipStream_t s) {
    const int n = col_jobs_blocks(&jobs);
    if (n > 0) col_jobs_kernel<<<dim3(n), dim3(256), 0, s>>>(jobs);
    return CPT_OK;
}

template <typename LP, bool GELU_IN, int NA = 4>      // NA: float4 per lane per row array (4: any H <= 1024; 3: the H = 768 instantiation -- a quarter fewer registers, 36 instead of 48 KB of staging: four workgroups per CU)
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ g, float eps, float* dx,      // (dx may be dy_resid's buffer: every row is read before it is written)
                                                     LP* __restrict__ dx_lp, float* __restrict__ dg, float* __restrict__ db,
                                                     int R, int H, int grp, int grp_stride, int grp_off, int rows_per_block,
                                                     float* __restrict__ part, DropSpec dr, float* __restrict__ dbias,
                                                     const float* __restrict__ stats, int dy_parts, size_t dy_stride, const float* dy_resid, int dy_bf16,
                                                     int nb_main, ColJobs jobs, RowMap drows, const unsigned short* __restrict__ keep_bits, DropSpec idr) {      // idr (round 6): dropout on the incoming gradient rows (LnBwdExtra::in_drop)
    // dr / dbias (training backward of LN(dropout(dense) + residual), round 2): dx_lp receives the gradient that enters the dense
    // layer -- dx through the dropout mask of the forward (regenerated, dropout.h; thresh 0: identity) -- and dbias its column
    // sums = the gradient of the dense bias; dx itself stays unmasked (it feeds the residual path).  Replaces a dropout_rows and
    // a colsum launch per LayerNorm.
    // Round 6: stats = the forward's (mean, rstd) per row (two of the three dependent wave reductions per row go away);
    // dy_parts / dy_stride / dy_resid = the incoming gradient as split-K partial matrices of the data-gradient GEMM in front (+ its residual),
    // added here in split order instead of by a reduction launch; workgroups from nb_main on run the column-sum jobs of EARLIER launches.
    constexpr int LNB_MAXV = NA;
    if ((int)blockIdx.x >= nb_main) { col_jobs_run(jobs, (int)blockIdx.x - nb_main); return; }
    __shared__ float red[3][4][256 * LNB_MAXV];     // [dg|db|dbias][wave][column]  (48 KB; 36 KB at NA = 3)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = (H + 255) / 256;
    f32x4 gsum[LNB_MAXV], bsum[LNB_MAXV], gg[LNB_MAXV], xsum[LNB_MAXV];
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        xsum[i] = f32x4{0, 0, 0, 0};
        gsum[i] = f32x4{0, 0, 0, 0};
        bsum[i] = f32x4{0, 0, 0, 0};
        const int c = (lane + 64 * i) * 4;
        gg[i] = (g && i < nv && c < H) ? *reinterpret_cast<const f32x4*>(g + c) : f32x4{0, 0, 0, 0};
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    if (!g) {
        // no LayerNorm on this path (use_img_layernorm = 0, modeling_bert.py:263-264): the backward is the row gather alone
        for (int r = r0 + wave; r < r1; r += 4) {
            const size_t yrow = (size_t)(r / grp) * grp_stride + grp_off + (r % grp);
#pragma unroll
            for (int i = 0; i < LNB_MAXV; ++i) {
                const int c = (lane + 64 * i) * 4;
                if (i < nv && c < H) {
                    f32x4 o = *reinterpret_cast<const f32x4*>(dy + yrow * H + c);
                    if (idr.thresh != 0) {
                        bool keep[4];
                        drop_hidden4(idr, ((uint64_t)yrow * H + c) >> 2, keep);
#pragma unroll
                        for (int j = 0; j < 4; ++j) o[j] = keep[j] ? o[j] * idr.scale : 0.f;
                    }
                    if (dx) *reinterpret_cast<f32x4*>(dx + (size_t)r * H + c) = o;
                    if (dx_lp) {
                        if constexpr (sizeof(LP) == 2) {
                            bf16x4 p;
#pragma unroll
                            for (int j = 0; j < 4; ++j) p[j] = (bf16)o[j];
                            *reinterpret_cast<bf16x4*>(dx_lp + (size_t)r * H + c) = p;
                        } else {
                            *reinterpret_cast<f32x4*>(dx_lp + (size_t)r * H + c) = o;
                        }
                    }
                }
            }
        }
        return;
    }
    // round 3: the NEXT row's x and dy are requested before this row is reduced (a wave walks two rows at eight rows per workgroup: the second
    // row's memory round trip used to start only after the first row's four wave reductions and stores)
    f32x4 nx[LNB_MAXV], nd[LNB_MAXV];
    float2 nst = {0.f, 1.f};
    unsigned nkb = 0;      // the row's keep bits (keep_bits), fetched with the row
    auto fetch = [&](int r) {
        const size_t yr = (size_t)(r / grp) * grp_stride + grp_off + (r % grp);
        if (stats) nst = *reinterpret_cast<const float2*>(stats + 2 * (size_t)__builtin_amdgcn_readfirstlane(r));
        if (keep_bits) nkb = keep_bits[(size_t)r * 64 + lane];
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
                nx[i] = *reinterpret_cast<const f32x4*>(x + (size_t)r * H + c);
                if (dy_bf16) {      // two bf16 partial matrices of the data-gradient GEMM (gemm_nn split2_bf16): bf16 -> fp32 is exact, the sum and the residual add are fp32
                    const bf16* dyh = reinterpret_cast<const bf16*>(dy);
                    const bf16x4 p0 = *reinterpret_cast<const bf16x4*>(dyh + yr * H + c), p1 = *reinterpret_cast<const bf16x4*>(dyh + dy_stride + yr * H + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) nd[i][j] = (float)p0[j] + (float)p1[j];
                } else {
                nd[i] = *reinterpret_cast<const f32x4*>(dy + yr * H + c);
                for (int k0 = 1; k0 < dy_parts; k0 += 8) {      // eight partial matrices' loads in flight at a time, added in split order
                    f32x4 t[8];
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
                        if (k0 + kk < dy_parts) t[kk] = *reinterpret_cast<const f32x4*>(dy + (size_t)(k0 + kk) * dy_stride + yr * H + c);
#pragma unroll
                    for (int kk = 0; kk < 8; ++kk)
                        if (k0 + kk < dy_parts) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) nd[i][j] += t[kk][j];
                        }
                }
                }
                if (dy_resid) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(dy_resid + yr * H + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) nd[i][j] += t[j];
                }
                if (idr.thresh != 0) {
                    bool keep[4];
                    drop_hidden4(idr, ((uint64_t)yr * H + c) >> 2, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) nd[i][j] = keep[j] ? nd[i][j] * idr.scale : 0.f;
                }
            }
        }
    };
    if (r0 + wave < r1) fetch(r0 + wave);
    for (int r = r0 + wave; r < r1; r += 4) {
        f32x4 xv[LNB_MAXV], dv[LNB_MAXV], uv[LNB_MAXV];
        float s = 0.f;
        const float2 st = nst;
        const unsigned kb_row = nkb;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i) { xv[i] = nx[i]; dv[i] = nd[i]; }
        if (r + 4 < r1) fetch(r + 4);
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
                if (GELU_IN) {
                    uv[i] = xv[i];
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[i][j] = gelu_erf(uv[i][j]);
                }
                s += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
            }
        }
        float mean, rstd;
        if (stats) { mean = st.x; rstd = st.y; }
        else {
        mean = wave_sum(s) / (float)H;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i)
            if (i < nv && (lane + 64 * i) * 4 < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = xv[i][j] - mean; q += d * d; }
            }
        rstd = 1.0f / sqrtf(wave_sum(q) / (float)H + eps);
        }
        float s1 = 0.f, s2 = 0.f;     // sum(dy*g), sum(dy*g*xhat)
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i)
            if (i < nv && (lane + 64 * i) * 4 < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[i][j] - mean) * rstd;
                    const float gy = dv[i][j] * gg[i][j];
                    s1 += gy;
                    s2 += gy * xh;
                    gsum[i][j] += dv[i][j] * xh;
                    bsum[i][j] += dv[i][j];
                }
            }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[i][j] - mean) * rstd;
                    float d = rstd * (dv[i][j] * gg[i][j] - s1 - xh * s2);
                    if (GELU_IN) d *= gelu_erf_grad(uv[i][j]);
                    o[j] = d;
                }
                if (dx) *reinterpret_cast<f32x4*>(dx + (size_t)r * H + c) = o;
                if (dr.thresh != 0) {
                    bool keep[4];
                    if (keep_bits) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) keep[j] = (kb_row >> (4 * i + j)) & 1u;
                    } else
                    drop_hidden4(dr, ((uint64_t)rowmap_row(drows, r) * H + c) >> 2, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = keep[j] ? o[j] * dr.scale : 0.f;
                }
                if (dbias) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) xsum[i][j] += o[j];
                }
                if (dx_lp) {
                    if constexpr (sizeof(LP) == 2) {
                        bf16x4 p;
#pragma unroll
                        for (int j = 0; j < 4; ++j) p[j] = (bf16)o[j];
                        *reinterpret_cast<bf16x4*>(dx_lp + (size_t)r * H + c) = p;
                    } else {
                        *reinterpret_cast<f32x4*>(dx_lp + (size_t)r * H + c) = o;
                    }
                }
            }
        }
    }
    // reduce dg/db over the 4 waves, one atomic per column per workgroup  (one pass over a [3][4][H] staging array: the 16 KB
    // one-sum-at-a-time form that lets eight workgroups share a CU was measured SLOWER, 19.5 vs 16.8 us at 3840 rows -- the kernel
    // is bound by its per-row dependency chain, not by occupancy)
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][wave][(lane + 64 * i) * 4 + j] = gsum[i][j];
            red[1][wave][(lane + 64 * i) * 4 + j] = bsum[i][j];
            red[2][wave][(lane + 64 * i) * 4 + j] = xsum[i][j];
        }
    __syncthreads();
    const int nsum = dbias ? 3 : 2;
    if (part) {     // two-stage column sums: this block's partial rows [nsum][H], added up by ln_bwd_reduce_kernel or a later launch's column-sum job
        for (int c = threadIdx.x; c < H; c += 256)
            for (int k = 0; k < nsum; ++k)
                part[((size_t)blockIdx.x * nsum + k) * H + c] = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
        return;
    }
    for (int c = threadIdx.x; c < H; c += 256) {
        atomicAdd(&dg[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(&db[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        if (dbias) atomicAdd(&dbias[c], red[2][0][c] + red[2][1][c] + red[2][2][c] + red[2][3][c]);
    }
}

// second stage: column c of [nb][nsum][H] partials; blockIdx.y takes a slice of the nb blocks, one atomic per (slice, column)
__global__ __launch_bounds__(256) void ln_bwd_reduce_kernel(const float* __restrict__ part, float* __restrict__ dg, float* __restrict__ db,
                                                            float* __restrict__ dbias, int nb, int H, int per) {
    const int nsum = dbias ? 3 : 2;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= nsum * H) return;
    const int b0 = blockIdx.y * per;
    float v[16];                                        // per == 16: all loads in flight before the first add
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = (b0 + k < nb) ? part[(size_t)(b0 + k) * nsum * H + c] : 0.f;
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) a += v[k];
    atomicAdd(c < H ? &dg[c] : (c < 2 * H ? &db[c - H] : &dbias[c - 2 * H]), a);
}

CPT_SWITCH(int g_lnb_rpb, 0);       // cpt_set_tuning(17, rows): rows per workgroup of the two-stage form (0: 8 from 2048 rows on, else 4 = one row per wave)
void set_lnb_rpb(int v) { CPT_SWITCH_SET(g_lnb_rpb = v > 0 ? v : 0); (void)v; }

// rows per workgroup / partial rows of the two-stage form for R rows (the training step sizes its partial buffers and chains its reductions with these)
static int ln_bwd_rpb_two_stage(int R) { return g_lnb_rpb > 0 ? g_lnb_rpb : (R >= 2048 ? 8 : 4); }
int ln_bwd_part_rows(int R) { const int rpb = ln_bwd_rpb_two_stage(R); return (R + rpb - 1) / rpb; }

int ln_bwd(const float* dy, const float* x, const float* g, float eps, float* dx, void* dx_lp, int lp_dtype,
           float* dg, float* db, int R, int H, int grp, int grp_stride, int grp_off, int gelu_in, hipStream_t s,
           float* part, size_t part_bytes, const DropSpec* drop, float* dbias, const LnBwdExtra* ext) {
    const DropSpec dr = drop ? *drop : DropSpec{};
    if ((dr.thresh != 0 && (!g || grp != R)) || (dbias && !g)) return CPT_ERR_SHAPE;      // mask: compact rows only (its element index is the compact row's); bias sums: any row placement, but not the plain row gather
    const int nsum = dbias ? 3 : 2;
    if (R <= 0 || H % 4 || H > 256 * LNB_MAXV || grp <= 0) return CPT_ERR_SHAPE;
    if (!dy || (g && (!x || !dg || !db))) return CPT_ERR_NULL;      // g == NULL: identity (row gather only)
    const LnBwdExtra e0 = {};
    const LnBwdExtra& ex = ext ? *ext : e0;
    if (ex.dy_parts > 1 && (grp != R || !g)) return CPT_ERR_SHAPE;
    if (ex.dy_parts_bf16 && ex.dy_parts != 2) return CPT_ERR_SHAPE;
    if (ex.defer_reduce && !(g && part && (size_t)ln_bwd_part_rows(R) * nsum * H * 4 <= part_bytes)) return CPT_ERR_WORKSPACE;
    // rows per block: with atomics fewer blocks = fewer dgamma/dbeta atomics (2*H per block); with a partial-sum buffer (two-stage
    // column sums) one row per wave keeps 4x the rows in flight: the kernel is latency-bound otherwise (22 -> ~10 us at 3840 rows)
    int rpb = R >= 2048 ? 16 : (R > 256 ? 8 : 4);      // (a few rows -- the head's 4 .. 64 -- one per wave: each wave's row chain once instead of twice in sequence)
    if (g && part && (R >= 1024 || ex.defer_reduce) && (size_t)ln_bwd_part_rows(R) * nsum * H * 4 <= part_bytes) rpb = ln_bwd_rpb_two_stage(R); else part = nullptr;      // 8: two rows per wave, half the partial rows (3840 rows: 5.83 vs 5.89 ms per step; 960 blocks of 4 rows are 1.25 rounds of the 3 blocks per CU the 48 KB staging array allows)
    const int nb = (R + rpb - 1) / rpb;
    const ColJobs j0 = {};
    const ColJobs& jobs = ex.jobs ? *ex.jobs : j0;
    dim3 grid(nb + col_jobs_blocks(&jobs)), block(256);
    const bool lp16 = dx_lp && lp_dtype == CPT_BF16;
    const int dyp = ex.dy_parts > 1 ? ex.dy_parts : 1;
#define LNB(LPT, GI) do { if (H == 768) ln_bwd_kernel<LPT, GI, 3><<<grid, block, 0, s>>>(dy, x, g, eps, dx, (LPT*)dx_lp, dg, db, R, H, grp, grp_stride, grp_off, rpb, part, dr, dbias, ex.stats, dyp, ex.dy_stride, ex.dy_resid, (ex.dy_parts_bf16 && dyp == 2) ? 1 : 0, nb, jobs, ex.drop_rows, (dr.thresh != 0 && H <= 1024) ? ex.keep_bits : nullptr, ex.in_drop); \
                          else ln_bwd_kernel<LPT, GI, 4><<<grid, block, 0, s>>>(dy, x, g, eps, dx, (LPT*)dx_lp, dg, db, R, H, grp, grp_stride, grp_off, rpb, part, dr, dbias, ex.stats, dyp, ex.dy_stride, ex.dy_resid, (ex.dy_parts_bf16 && dyp == 2) ? 1 : 0, nb, jobs, ex.drop_rows, (dr.thresh != 0 && H <= 1024) ? ex.keep_bits : nullptr, ex.in_drop); } while (0)
    if (lp16) { if (gelu_in) LNB(bf16, true); else LNB(bf16, false); }
    else      { if (gelu_in) LNB(float, true); else LNB(float, false); }
#undef LNB
    if (part && !ex.defer_reduce) {
        const int per = 16;
        ln_bwd_reduce_kernel<<<dim3((nsum * H + 255) / 256, (nb + per - 1) / per), dim3(256), 0, s>>>(part, dg, db, dbias, nb, H, per);
    }
    return CPT_OK;
}

// ---- embedding LayerNorm backward + scatter-add into the three tables -----------------------------
__global__ __launch_bounds__(256) void embed_bwd_kernel(
    const float* __restrict__ dy, const int64_t* __restrict__ ids, const int64_t* __restrict__ tt, const int64_t* __restrict__ pos,
    const float* __restrict__ word, const float* __restrict__ posw, const float* __restrict__ typew, const float* __restrict__ g,
    float eps, float* __restrict__ dword, float* __restrict__ dposw, float* __restrict__ dtypew, float* __restrict__ dg,
    float* __restrict__ db, int B, int Lt, int L, int H, int vocab, int max_pos, int type_vocab, int rows_per_block, DropSpec idr) {
    __shared__ float red[2][4][256 * LNB_MAXV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = (H + 255) / 256;
    const int R = B * Lt;
    const bool two_types = type_vocab <= 2;
    f32x4 gsum[LNB_MAXV], bsum[LNB_MAXV], gg[LNB_MAXV], t0[LNB_MAXV], t1[LNB_MAXV];
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) {
        t0[i] = f32x4{0, 0, 0, 0};
        t1[i] = f32x4{0, 0, 0, 0};
        gsum[i] = f32x4{0, 0, 0, 0};
        bsum[i] = f32x4{0, 0, 0, 0};
        const int c = (lane + 64 * i) * 4;
        gg[i] = (i < nv && c < H) ? *reinterpret_cast<const f32x4*>(g + c) : f32x4{0, 0, 0, 0};
    }
    // round 6: rows are walked POSITION-major (r' = t * B + b), so the rows of a workgroup share their position-table row (B >= rows per
    // workgroup): its gradient rides in registers and leaves as one atomic per column per run of equal position ids instead of one per
    // row -- the 32 sequences of the step used to send 32-way contended atomics at the same 70 rows (53 us)
    f32x4 pacc[LNB_MAXV];
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i) pacc[i] = f32x4{0, 0, 0, 0};
    long cur_pid = -1;
    auto flush_pos = [&]() {
        if (cur_pid < 0) return;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(&dposw[(size_t)cur_pid * H + c + j], pacc[i][j]);
            }
            pacc[i] = f32x4{0, 0, 0, 0};
        }
    };
    const int r0 = blockIdx.x * rows_per_block, r1 = min(R, r0 + rows_per_block);
    for (int rp = r0 + wave; rp < r1; rp += 4) {
        const int t = rp / B, b = rp % B;
        const int r = b * Lt + t;
        long wid = ids[r], pid = pos ? pos[r] : t, tid = tt ? tt[r] : 0;
        wid = wid < 0 ? 0 : (wid >= vocab ? vocab - 1 : wid);
        pid = pid < 0 ? 0 : (pid >= max_pos ? max_pos - 1 : pid);
        tid = tid < 0 ? 0 : (tid >= type_vocab ? type_vocab - 1 : tid);
        if (pid != cur_pid) { flush_pos(); cur_pid = pid; }
        f32x4 xv[LNB_MAXV], dv[LNB_MAXV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(word + (size_t)wid * H + c);
                const f32x4 p = *reinterpret_cast<const f32x4*>(posw + (size_t)pid * H + c);
                const f32x4 q = *reinterpret_cast<const f32x4*>(typew + (size_t)tid * H + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) xv[i][j] = a[j] + p[j] + q[j];
                dv[i] = *reinterpret_cast<const f32x4*>(dy + ((size_t)b * L + t) * H + c);
                if (idr.thresh != 0) {      // BertEmbeddings' dropout, backward: the mask of the forward at the same element index
                    bool keep[4];
                    drop_hidden4(idr, (((uint64_t)b * L + t) * H + c) >> 2, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dv[i][j] = keep[j] ? dv[i][j] * idr.scale : 0.f;
                }
                s += xv[i][0] + xv[i][1] + xv[i][2] + xv[i][3];
            }
        }
        const float mean = wave_sum(s) / (float)H;
        float q2 = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i)
            if (i < nv && (lane + 64 * i) * 4 < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const float d = xv[i][j] - mean; q2 += d * d; }
            }
        const float rstd = 1.0f / sqrtf(wave_sum(q2) / (float)H + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i)
            if (i < nv && (lane + 64 * i) * 4 < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[i][j] - mean) * rstd;
                    const float gy = dv[i][j] * gg[i][j];
                    s1 += gy;
                    s2 += gy * xh;
                    gsum[i][j] += dv[i][j] * xh;
                    bsum[i][j] += dv[i][j];
                }
            }
        s1 = wave_sum(s1) / (float)H;
        s2 = wave_sum(s2) / (float)H;
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i) {
            const int c = (lane + 64 * i) * 4;
            if (i < nv && c < H) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (xv[i][j] - mean) * rstd;
                    const float d = rstd * (dv[i][j] * gg[i][j] - s1 - xh * s2);
                    // nn.Embedding(padding_idx=0): the lookup contributes no gradient to row 0
                    if (wid != 0) atomicAdd(&dword[(size_t)wid * H + c + j], d);
                    pacc[i][j] += d;
                    // token_type table: two rows shared by every token -- per-element atomics were 1100-way contended (226 us);
                    // the block keeps one partial row per type and adds it once
                    if (two_types) { if (tid == 0) t0[i][j] += d; else t1[i][j] += d; }
                    else atomicAdd(&dtypew[(size_t)tid * H + c + j], d);
                }
            }
        }
    }
    // the waves' position-row sums: combined over runs of waves that ended on the same position id (all four, normally), one atomic per column per run
    __shared__ long spid[4];
    if (lane == 0) spid[wave] = cur_pid;
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) red[0][wave][(lane + 64 * i) * 4 + j] = pacc[i][j];
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        float acc = 0.f;
        long p = -1;
        for (int wv = 0; wv < 4; ++wv) {
            const long pw = spid[wv];
            if (pw < 0) continue;
            if (pw != p) { if (p >= 0) atomicAdd(&dposw[(size_t)p * H + c], acc); acc = 0.f; p = pw; }
            acc += red[0][wv][c];
        }
        if (p >= 0) atomicAdd(&dposw[(size_t)p * H + c], acc);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < LNB_MAXV; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][wave][(lane + 64 * i) * 4 + j] = gsum[i][j];
            red[1][wave][(lane + 64 * i) * 4 + j] = bsum[i][j];
        }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256) {
        atomicAdd(&dg[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(&db[c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
    if (two_types) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < LNB_MAXV; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                red[0][wave][(lane + 64 * i) * 4 + j] = t0[i][j];
                red[1][wave][(lane + 64 * i) * 4 + j] = t1[i][j];
            }
        __syncthreads();
        for (int c = threadIdx.x; c < H; c += 256) {
            atomicAdd(&dtypew[c], red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
            if (type_vocab > 1) atomicAdd(&dtypew[H + c], red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
        }
    }
}

int embed_bwd(const float* dy, const int64_t* ids, const int64_t* tt, const int64_t* pos, const float* word,
              const float* posw, const float* typew, const float* g, float eps, float* dword, float* dposw,
              float* dtypew, float* dg, float* db, int B, int Lt, int L, int H, int vocab, int max_pos,
              int type_vocab, hipStream_t s, const DropSpec* in_drop) {
    if (B <= 0 || Lt <= 0 || H % 4 || H > 256 * LNB_MAXV) return CPT_ERR_SHAPE;
    // rows per workgroup (round 6): 16 rows leave 18 workgroups at 4 sequences, each wave walking four rows' load -> reduce -> atomics chains in sequence
    // (27.2 us; one row per wave: 11.7 us).  From 1024 rows on 16 stay (2240 rows: 35.0 us against 39.4 us at 8 -- twice the position-table and gain / shift atomics).
    const int rpb = B * Lt >= 1024 ? 16 : 4;
    dim3 grid((B * Lt + rpb - 1) / rpb), block(256);
    embed_bwd_kernel<<<grid, block, 0, s>>>(dy, ids, tt, pos, word, posw, typew, g, eps, dword, dposw, dtypew, dg, db,
                                           B, Lt, L, H, vocab, max_pos, type_vocab, rpb, in_drop ? *in_drop : DropSpec{});
    return CPT_OK;
}

// ---- small helpers ------------------------------------------------------------------------------------
// dst[b*L + pos[b]][:] += src[b][:]   (gradient of the [MASK]-row gather)
__global__ __launch_bounds__(256) void scatter_rows_add_kernel(const float* __restrict__ src, const int64_t* __restrict__ pos,
                                                               float* __restrict__ dst, int L, int H, const int64_t* __restrict__ seq, int n_seq) {
    const int b = blockIdx.x;
    long p = pos ? pos[b] : 0;
    p = p < 0 ? 0 : (p >= L ? L - 1 : p);
    long q = seq ? seq[b] : b;                    // seq: source row b belongs to position pos[b] of sequence seq[b] (label grids)
    q = q < 0 ? 0 : (q >= n_seq ? n_seq - 1 : q);
    // (atomic: a clamped or repeated (sequence, position) pair must not lose an update)
    for (int c = threadIdx.x; c < H; c += 256) atomicAdd(&dst[((size_t)q * L + p) * H + c], src[(size_t)b * H + c]);
}
int scatter_rows_add(const float* src, const int64_t* pos, float* dst, int B, int L, int H, hipStream_t s, const int64_t* seq, int n_seq) {
    scatter_rows_add_kernel<<<dim3(B), dim3(256), 0, s>>>(src, pos, dst, L, H, seq, seq ? n_seq : B);
    return CPT_OK;
}

// dx = dy * (1 - y^2), y = tanh(.) as saved by the forward (BertPooler, third-party); optional copy in lp_dtype
template <typename TL>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                       TL* __restrict__ dx_lp, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = dy[i] * (1.0f - y[i] * y[i]);
    dx[i] = v;
    if (dx_lp) dx_lp[i] = from_f32<TL>(v);
}
int tanh_bwd(const float* dy, const float* y, float* dx, void* dx_lp, int lp_dtype, size_t n, hipStream_t s) {
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (lp_dtype == CPT_BF16) tanh_bwd_kernel<bf16><<<grid, block, 0, s>>>(dy, y, dx, (bf16*)dx_lp, n);
    else tanh_bwd_kernel<float><<<grid, block, 0, s>>>(dy, y, dx, (float*)dx_lp, n);
    return CPT_OK;
}

// dst[R][K] = src[R][Kp]  (drop the zero padding of the img weight gradient; like every Linear weight gradient it is WRITTEN, not added)
__global__ __launch_bounds__(256) void unpad_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int K, int Kp) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)R * K) return;
    const int r = (int)(idx / K), c = (int)(idx % K);
    dst[idx] = src[(size_t)r * Kp + c];
}
int unpad_add(const float* src, float* dst, int R, int K, int Kp, hipStream_t s) {
    const size_t n = (size_t)R * K;
    unpad_add_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(src, dst, R, K, Kp);
    return CPT_OK;
}

// One launch that clears up to ZS_MAX small tensors (the gradient vectors the backward adds into with atomics): the table rides in the
// kernel arguments, blockIdx.x = segment, blockIdx.y strides over it.
__global__ __launch_bounds__(256) void zero_segments_kernel(ZeroSegs z) {
    float* p = z.p[blockIdx.x];
    const unsigned n = z.n[blockIdx.x];
    for (unsigned i = blockIdx.y * 256 + threadIdx.x; i < n; i += gridDim.y * 256) p[i] = 0.f;
}
int zero_segments(const ZeroSegs& z, hipStream_t s) {
    if (z.count <= 0) return CPT_OK;
    if (z.count > ZS_MAX) return CPT_ERR_SHAPE;
    zero_segments_kernel<<<dim3((unsigned)z.count, 64), dim3(256), 0, s>>>(z);      // (64 slices: the 512 x 768 position table is the long segment; 8 slices left its 393 k stores to 2048 threads, 12 us)
    return CPT_OK;
}

// out[r][c] = x[r][c] * scale / max(count,1) for c < C, 0 for C <= c < ldo; count read from device
template <typename TO>
__global__ __launch_bounds__(256) void scale_cast_kernel(const float* __restrict__ x, const float* __restrict__ loss_acc,
                                                         float scale, const float* __restrict__ dscale, TO* __restrict__ out,
                                                         int R, int C, int ldo) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)R * ldo) return;
    const int r = (int)(idx / ldo), c = (int)(idx % ldo);
    const float k = scale * (dscale ? dscale[0] : 1.f) / fmaxf(loss_acc ? loss_acc[1] : 1.f, 1.f);
    out[idx] = from_f32<TO>(c < C ? x[(size_t)r * C + c] * k : 0.f);
}
// ... with the column sums of the ROUNDED output added into colsum[c] (round 6: the decoder / relation bias gradient -- one launch instead of scale_cast +
// colsum): one thread per column, the R <= 64 head rows in a loop, eight loads in flight
template <typename TO>
__global__ __launch_bounds__(256) void scale_cast_colsum_kernel(const float* __restrict__ x, const float* __restrict__ loss_acc, float scale,
                                                                const float* __restrict__ dscale, TO* __restrict__ out, int R, int C, int ldo,
                                                                float* __restrict__ colsum) {
    // 64 columns per workgroup; wave w takes rows w, w + 4, ... (R <= 64: at most 16 per thread, all loads in flight), the four partial sums meet in LDS.
    // (One thread per column over all rows was 11.9 us at the 32 head rows of the 32-sequence step: four, then one long, batch of strided loads per thread.)
    __shared__ float part[4][64];
    const int cx = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cx;
    const float k = scale * (dscale ? dscale[0] : 1.f) / fmaxf(loss_acc ? loss_acc[1] : 1.f, 1.f);
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = rg + 4 * u;
        v[u] = (c < C && r < R) ? x[(size_t)r * C + c] * k : 0.f;
    }
    float acc = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int r = rg + 4 * u;
        if (c < ldo && r < R) {
            const TO o = from_f32<TO>(v[u]);
            out[(size_t)r * ldo + c] = o;
            acc += (float)o;
        }
    }
    part[rg][cx] = acc;
    __syncthreads();
    if (rg == 0 && c < C) atomicAdd(&colsum[c], part[0][cx] + part[1][cx] + part[2][cx] + part[3][cx]);
}
int scale_cast(const float* x, const float* loss_acc, float scale, const float* dscale, void* out, int out_dtype, int R, int C, int ldo,
               hipStream_t s, float* colsum_out) {
    if (colsum_out && R <= 64) {
        dim3 grid((unsigned)((ldo + 63) / 64)), block(256);
        if (out_dtype == CPT_BF16) scale_cast_colsum_kernel<bf16><<<grid, block, 0, s>>>(x, loss_acc, scale, dscale, (bf16*)out, R, C, ldo, colsum_out);
        else scale_cast_colsum_kernel<float><<<grid, block, 0, s>>>(x, loss_acc, scale, dscale, (float*)out, R, C, ldo, colsum_out);
        return CPT_OK;
    }
    const size_t n = (size_t)R * ldo;
    dim3 grid((unsigned)((n + 255) / 256)), block(256);
    if (out_dtype == CPT_BF16) scale_cast_kernel<bf16><<<grid, block, 0, s>>>(x, loss_acc, scale, dscale, (bf16*)out, R, C, ldo);
    else scale_cast_kernel<float><<<grid, block, 0, s>>>(x, loss_acc, scale, dscale, (float*)out, R, C, ldo);
    if (colsum_out) return colsum(out, out_dtype, ldo, colsum_out, R, C, s);      // (many rows: the tiled column-sum launch)
    return CPT_OK;
}

// ---- hidden dropout as a row pass: y = dropout(x) (+ resid), optional low-precision copy ------------------------------
// Forward of the reference's hidden dropouts (BertEmbeddings / modeling_bert.py:266 / BertSelfOutput / BertOutput) and,
// with resid = NULL, their backward.  4 elements per thread = one Philox call (dropout.h).
template <typename TL>
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* x, const float* resid, float* y,      // x == y in the in-place calls of the training step: no __restrict__
                                                           TL* __restrict__ y_lp, size_t n4, DropSpec d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
    bool keep[4];
    drop_hidden4(d, i, keep);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * d.scale : 0.f;
    if (resid) {
        const f32x4 r = *reinterpret_cast<const f32x4*>(resid + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += r[e];
    }
    if (y) *reinterpret_cast<f32x4*>(y + i * 4) = v;
    if (y_lp) {
        if constexpr (sizeof(TL) == 2) {
            bf16x4 pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (bf16)v[e];
            *reinterpret_cast<bf16x4*>(y_lp + i * 4) = pk;
        } else {
            *reinterpret_cast<f32x4*>(y_lp + i * 4) = v;
        }
    }
}
int dropout_rows(const float* x, const float* resid, float* y, void* y_lp, int lp_dtype, int R, int H, const DropSpec& d, hipStream_t s) {
    if (R <= 0 || H <= 0 || H % 4) return CPT_ERR_SHAPE;
    if (!x || (!y && !y_lp)) return CPT_ERR_NULL;
    const size_t n4 = (size_t)R * H / 4;
    dim3 grid((unsigned)((n4 + 255) / 256)), block(256);
    if (y_lp && lp_dtype == CPT_BF16) dropout_rows_kernel<bf16><<<grid, block, 0, s>>>(x, resid, y, (bf16*)y_lp, n4, d);
    else dropout_rows_kernel<float><<<grid, block, 0, s>>>(x, resid, y, (float*)y_lp, n4, d);
    return CPT_OK;
}

// keep-mask export for the tests / the oracle: kind 0 = hidden site [n0 rows][n1 columns]; kind 1 = attention site
// [n0 = sequences * heads][n1 = queries][n2 = keys].  out[...] = 1 keep, 0 drop.
__global__ __launch_bounds__(256) void dropout_mask_kernel(int kind, unsigned char* __restrict__ out, size_t n, int n1, int n2, DropSpec d) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    if (kind == 0) {
        bool keep[4];
        drop_hidden4(d, i >> 2, keep);
        out[i] = keep[i & 3] ? 1 : 0;
    } else {
        const int k = (int)(i % n2), q = (int)((i / n2) % n1);
        const uint32_t bh = (uint32_t)(i / ((size_t)n1 * n2));
        out[i] = drop_attn_one(d, bh, q, k) ? 1 : 0;
    }
}
int dropout_mask(int kind, unsigned char* out, int n0, int n1, int n2, const DropSpec& d, hipStream_t s) {
    if (!out) return CPT_ERR_NULL;
    if (n0 <= 0 || n1 <= 0 || (kind == 1 && n2 <= 0) || (kind != 0 && kind != 1)) return CPT_ERR_SHAPE;
    const size_t n = kind == 0 ? (size_t)n0 * n1 : (size_t)n0 * n1 * n2;
    dropout_mask_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(kind, out, n, n1, n2, d);
    return CPT_OK;
}

// ---- attention backward (generic, fp32 math on LDS tiles; one workgroup per (sequence, head)) ----
// Recomputes P = softmax(QK^T/8 + mask) per 32-query block, then
//   dV += P^T dO,  dP = dO V^T,  dS = P (dP - rowsum(dP P)),  dQ = dS K / 8,  dK += dS^T Q / 8.
// QB = query rows per block; VG (round 3, 176 < L <= 288 in the fp32 parity mode: the GQA / VCR few-shot lengths 210 / 265): V is not
// held in LDS but read from global memory (74 KB per head, L2-resident) and the query block shrinks to 16 rows -- 140 KB at
// L = 288 with dropout instead of the 240 KB the QB = 32 / V-in-LDS form would need.
constexpr int AB_QB = 32, AB_D = 64;
template <typename T, int MAXE, int QB = AB_QB, bool VG = false>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ attn_mask,
                                                       const T* __restrict__ dctx, T* __restrict__ dqkv, int B, int L, int heads,
                                                       DropSpec dr, int mask3d) {      // mask3d: attn_mask is [B][L][L], one row per query (modeling_bert.py:215-216)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int LP1 = L + 1;
    float* sK = reinterpret_cast<float*>(smem);          // [L][65]
    float* sV = sK + L * 65;                              // [L][65]  (VG: absent)
    float* sQ = sV + (VG ? 0 : L * 65);                   // [QB][65]
    float* sO = sQ + QB * 65;                          // [32][65]  (dO block)
    float* sP = sO + QB * 65;                          // [32][L+1]
    float* sS = sP + QB * LP1;                         // [32][L+1]  (dP then dS)
    float* sM = sS + QB * LP1;                         // [L] additive mask
    float* sW = sM + L;                                   // [32][L+1] dropout multipliers (0 or 1/(1-p)); only with dropout
    const bool drop = dr.thresh != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int H = heads * AB_D;
    const size_t ldq = (size_t)3 * H;
    const T* base = qkv + (size_t)b * L * ldq + h * AB_D;
    T* dbase = dqkv + (size_t)b * L * ldq + h * AB_D;
    for (int idx = tid; idx < L * AB_D; idx += 256) {
        const int r = idx / AB_D, c = idx % AB_D;
        sK[r * 65 + c] = to_f32(base[(size_t)r * ldq + H + c]);
        if (!VG) sV[r * 65 + c] = to_f32(base[(size_t)r * ldq + 2 * H + c]);
    }
    for (int j = tid; j < L; j += 256) sM[j] = (attn_mask && !mask3d) ? (1.0f - (float)attn_mask[(size_t)b * L + j]) * -10000.0f : 0.f;
    const int64_t* m3 = (attn_mask && mask3d) ? attn_mask + (size_t)b * L * L : nullptr;

    // this thread's share of the dK / dV accumulators: elements e = tid + 256*k of the [L][64] tiles
    float accK[MAXE], accV[MAXE];                          // MAXE >= L*64/256
    const int ne = (L * AB_D + 255) / 256;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) { accK[k] = 0.f; accV[k] = 0.f; }

    for (int q0 = 0; q0 < L; q0 += QB) {
        const int nq = min(QB, L - q0);
        __syncthreads();
        for (int idx = tid; idx < QB * AB_D; idx += 256) {
            const int r = idx / AB_D, c = idx % AB_D;
            const bool ok = r < nq;
            sQ[r * 65 + c] = ok ? to_f32(base[(size_t)(q0 + r) * ldq + c]) : 0.f;
            sO[r * 65 + c] = ok ? to_f32(dctx[((size_t)b * L + q0 + r) * H + h * AB_D + c]) : 0.f;
        }
        __syncthreads();
        // scores and dP for the block
        for (int idx = tid; idx < QB * L; idx += 256) {
            const int i = idx / L, j = idx % L;
            float s = 0.f, dp = 0.f;
#pragma unroll 8
            for (int d = 0; d < AB_D; ++d) {
                s += sQ[i * 65 + d] * sK[j * 65 + d];
                dp += sO[i * 65 + d] * (VG ? to_f32(base[(size_t)j * ldq + 2 * H + d]) : sV[j * 65 + d]);
            }
            sP[i * LP1 + j] = s * 0.125f + (m3 ? (1.0f - (float)m3[(size_t)min(q0 + i, L - 1) * L + j]) * -10000.0f : sM[j]);
            if (drop) {      // dp arrives as the gradient of the DROPPED probabilities: d/dP = mask / (1-p) times it
                const float wgt = drop_attn_one(dr, (uint32_t)blockIdx.x, min(q0 + i, L - 1), j) ? dr.scale : 0.f;
                sW[i * LP1 + j] = wgt;
                dp *= wgt;
            }
            sS[i * LP1 + j] = dp;
        }
        __syncthreads();
        // softmax rows + dS: wave w owns rows w, w+4, ...
        for (int i = wave; i < QB; i += 4) {
            float m = -INFINITY;
            for (int j = lane; j < L; j += 64) m = fmaxf(m, sP[i * LP1 + j]);
            m = wave_max(m);
            float sum = 0.f;
            for (int j = lane; j < L; j += 64) { const float p = expf(sP[i * LP1 + j] - m); sP[i * LP1 + j] = p; sum += p; }
            sum = wave_sum(sum);
            const float inv = 1.0f / sum;
            float dd = 0.f;
            for (int j = lane; j < L; j += 64) { const float p = sP[i * LP1 + j] * inv; sP[i * LP1 + j] = p; dd += p * sS[i * LP1 + j]; }
            dd = wave_sum(dd);
            for (int j = lane; j < L; j += 64) sS[i * LP1 + j] = sP[i * LP1 + j] * (sS[i * LP1 + j] - dd);
        }
        __syncthreads();
        // dQ block
        for (int idx = tid; idx < QB * AB_D; idx += 256) {
            const int i = idx / AB_D, d = idx % AB_D;
            if (i < nq) {
                float a = 0.f;
                for (int j = 0; j < L; ++j) a += sS[i * LP1 + j] * sK[j * 65 + d];
                dbase[(size_t)(q0 + i) * ldq + d] = from_f32<T>(a * 0.125f);
            }
        }
        // dK, dV accumulation
#pragma unroll
        for (int k = 0; k < MAXE; ++k) {
            if (k < ne) {
                const int e = tid + 256 * k;
                if (e < L * AB_D) {
                    const int j = e / AB_D, d = e % AB_D;
                    float ak = 0.f, av = 0.f;
                    for (int i = 0; i < nq; ++i) {
                        ak += sS[i * LP1 + j] * sQ[i * 65 + d];
                        av += (drop ? sP[i * LP1 + j] * sW[i * LP1 + j] : sP[i * LP1 + j]) * sO[i * 65 + d];
                    }
                    accK[k] += ak;
                    accV[k] += av;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
        if (k < ne) {
            const int e = tid + 256 * k;
            if (e < L * AB_D) {
                const int j = e / AB_D, d = e % AB_D;
                dbase[(size_t)j * ldq + H + d] = from_f32<T>(accK[k] * 0.125f);
                dbase[(size_t)j * ldq + 2 * H + d] = from_f32<T>(accV[k]);
            }
        }
    }
}

// ---- attention backward on MFMA (bf16, L <= 128): same math, scores recomputed on the matrix cores ----
// One workgroup per (sequence, head).  LDS holds Q, K, V, dO as swizzled row tiles (A/B operands
// of the score-shaped products) and Q^T, K^T, dO^T as padded transposed tiles (B operands of the
// products that contract over rows).  Phase A (wave <-> 32-query block, transposed scores: one query
// per lane) yields the softmax statistics, D = rowsum(dP.P) and dQ; phase B (wave <-> 32-key block,
// one key per lane) recomputes P/dS blockwise and accumulates dK, dV over the query blocks.
// As in the forward kernel, probabilities sit in the A-operand layout of the next MFMA once the
// contraction index of the B operand is permuted identically, so nothing round-trips through LDS.
__device__ __forceinline__ int kq_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// TR = true (round 2, 128 < L <= 288: the GQA / VCR few-shot shapes L = 165 + 45 and 165 + 100): the transposed copies of
// Q, K, dO (3 x 64 x (2 LP + 8) bytes) do not fit beside four 288-row tiles, so the B operands of the products that
// contract over rows are read from the ROW tiles with the gfx950 LDS transpose read ds_read_b64_tr_b16 (as the forward
// kernel reads V, attn_core.h).  The row swizzle of this variant rotates the 3-bit row-pair index ((p & 1) << 2 | p >> 1):
// still a bijection over the 8 row pairs a 16-byte fragment read touches (conflict-free), and the four consecutive rows of
// one transpose read land in four different 32-byte bank ranges.
__device__ __forceinline__ int kq_off_tr(int row, int chunk) {
    const int p = (row >> 1) & 7;
    return row * 128 + ((chunk ^ (((p & 1) << 2) | (p >> 1))) << 4);
}

template <int NKB, bool TR = false, bool HAVE = false>      // HAVE (TR only): ctx + stats given
__global__ __launch_bounds__(256, (TR && NKB <= 4) ? 2 : 1) void attn_bwd_mfma_kernel(const bf16* __restrict__ qkv, const int64_t* __restrict__ attn_mask,
                                                            const bf16* __restrict__ dctx, bf16* __restrict__ dqkv, int B, int L, int heads,
                                                            DropSpec dr, float* __restrict__ dbias, const bf16* __restrict__ ctx, const float* __restrict__ stats, int split) {
    // split (round 6, HAVE only): TWO workgroups per (sequence, head) -- with the forward's statistics the two phases below share nothing but the tiles, so
    // workgroup 2 p runs phase A (dQ) and 2 p + 1 phase B (dK, dV) of pair p side by side.  For launches that leave most of the chip idle (4 sequences:
    // 48 pairs on 256 CUs), where the launch lasts as long as ONE workgroup's load -> phase A -> phase B -> write-back chain.
    // ctx + stats (round 6, TR variants): the forward's context rows O and per-query softmax statistics (row max in base 2, 1 / row sum; attn_core.h
    // stat_row).  Phase A then needs neither the row reductions nor a second evaluation of the dP blocks: D = rowsum(dP . P) = rowsum(dO . O) (also
    // under dropout: O = P~ V) comes from the tile loads, and every key block is finished in one go (scores, dP, dS, dQ) instead of held.
    // dbias (optional, [3 * heads * 64]): += column sums of dqkv over this (sequence, head)'s rows = the gradient of the stacked
    // Q|K|V bias (replaces a colsum launch over the M x 3H tensor): every lane owns one column of its 32 x 32 block, the two
    // half-waves hold the two row halves
    constexpr int LP = NKB * 32;
    constexpr int TROW = LP * 2 + 8;                  // bytes per transposed-tile row (pad: conflict-free b64 reads)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sQ = smem;                          // [LP][64] bf16 swizzled rows
    unsigned char* sK = sQ + LP * 128;
    unsigned char* sV = sK + LP * 128;
    unsigned char* sO = sV + LP * 128;                 // dO rows
    unsigned char* tQ = sO + LP * 128;                 // [64][LP] transposed (+pad); TR: the row tiles themselves
    unsigned char* tK = tQ + 64 * TROW;
    unsigned char* tO = tK + 64 * TROW;
    float* sMask = reinterpret_cast<float*>(tO + 64 * TROW);   // [LP]
    if constexpr (TR) { tQ = sQ; tK = sK; tO = sO; sMask = reinterpret_cast<float*>(sO + LP * 128); }
    auto roff = [](int row, int chunk) { return TR ? kq_off_tr(row, chunk) : kq_off(row, chunk); };
    float* sM = sMask + LP;                            // row max (base 2, like the forward kernel's softmax)
    float* sLi = sM + LP;                              // 1 / row sum
    float* sD = sLi + LP;                              // rowsum(dP * P)
    uint32_t* sBits = reinterpret_cast<uint32_t*>(sD + LP);   // [NKB][LP] keep bits of the attention dropout: bit (key & 31) of word [key >> 5][query]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pair = (TR && HAVE && split) ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
    const int phase = (TR && HAVE && split) ? (int)(blockIdx.x & 1) : 2;      // 0: phase A only, 1: phase B only, 2: both
    const int b = pair / heads, h = pair % heads;
    const int H = heads * 64;
    const size_t ldq = (size_t)3 * H;
    const bf16* base = qkv + (size_t)b * L * ldq + h * 64;
    bf16* dbase = dqkv + (size_t)b * L * ldq + h * 64;

    // L <= 128: all global loads first (LP * 8 sixteen-byte chunks per tile = NKB per thread), the dropout bit plane under their latency, then the
    // LDS writes.  Longer sequences (NKB >= 5: 144 staging registers at NKB = 9 would spill) fill the bit plane first and load tile rows in a loop.
    constexpr int NST = NKB <= 4 ? NKB : 1;
    uint4 rq[NST], rk[NST], rv[NST], ro[NST];
    auto ld4 = [&](int idx, uint4& q4, uint4& k4, uint4& v4, uint4& o4) {
        const int r = idx >> 3, c = idx & 7;
        q4 = make_uint4(0, 0, 0, 0); k4 = q4; v4 = q4; o4 = q4;
        if (r < L) {
            q4 = *reinterpret_cast<const uint4*>(base + (size_t)r * ldq + c * 8);
            k4 = *reinterpret_cast<const uint4*>(base + (size_t)r * ldq + H + c * 8);
            v4 = *reinterpret_cast<const uint4*>(base + (size_t)r * ldq + 2 * H + c * 8);
            o4 = *reinterpret_cast<const uint4*>(dctx + ((size_t)b * L + r) * H + h * 64 + c * 8);
        }
    };
    constexpr bool have = TR && HAVE;
    auto st4 = [&](int idx, const uint4& q4, const uint4& k4, const uint4& v4, const uint4& o4) {
        const int r = idx >> 3, c = idx & 7;
        if constexpr (have) {      // D = rowsum(dO . O): the eight lanes that hold the row's chunks
            uint4 c4 = make_uint4(0, 0, 0, 0);
            if (r < L) c4 = *reinterpret_cast<const uint4*>(ctx + ((size_t)b * L + r) * H + h * 64 + c * 8);
            const bf16* oe = reinterpret_cast<const bf16*>(&o4);
            const bf16* ce = reinterpret_cast<const bf16*>(&c4);
            float dsum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) dsum += (float)oe[j] * (float)ce[j];
            dsum += lane_xor1(dsum); dsum += lane_xor2(dsum); dsum += lane_xor4(dsum);
            if (c == 0) sD[r] = dsum;
        }
        *reinterpret_cast<uint4*>(sQ + roff(r, c)) = q4;
        *reinterpret_cast<uint4*>(sK + roff(r, c)) = k4;
        *reinterpret_cast<uint4*>(sV + roff(r, c)) = v4;
        *reinterpret_cast<uint4*>(sO + roff(r, c)) = o4;
        if constexpr (!TR) {
        const bf16* qe = reinterpret_cast<const bf16*>(&q4);
        const bf16* ke = reinterpret_cast<const bf16*>(&k4);
        const bf16* oe = reinterpret_cast<const bf16*>(&o4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            *reinterpret_cast<bf16*>(tQ + (c * 8 + j) * TROW + r * 2) = qe[j];
            *reinterpret_cast<bf16*>(tK + (c * 8 + j) * TROW + r * 2) = ke[j];
            *reinterpret_cast<bf16*>(tO + (c * 8 + j) * TROW + r * 2) = oe[j];
        }
        }
    };
    if constexpr (NKB <= 4) {
#pragma unroll
        for (int it = 0; it < NKB; ++it) ld4(tid + it * 256, rq[it], rk[it], rv[it], ro[it]);
    }
    float mreg[(LP + 255) / 256];
#pragma unroll
    for (int i = 0; i < (LP + 255) / 256; ++i) {
        const int key = tid + i * 256;
        float mv = -INFINITY;
        if (key < L) mv = attn_mask ? (1.0f - (float)attn_mask[(size_t)b * L + key]) * -10000.0f : 0.f;
        mreg[i] = mv * ATT_LOG2E;
    }
    if (dr.thresh != 0) {
        // units of (query pair, 32-key block): eight Philox calls each, LP / 2 * NKB units over 256 threads (NKB = 4: one per thread)
        for (int u = tid; u < (LP / 2) * NKB; u += 256) {
            const int rp = u / NKB, kb = u % NKB;
            uint32_t w0 = 0, w1 = 0;
            if (2 * rp < L && kb * 32 < L) drop_attn_bits2x32(dr, (uint32_t)pair, (uint32_t)rp, (uint32_t)kb, w0, w1);
            *reinterpret_cast<uint2*>(&sBits[kb * LP + 2 * rp]) = make_uint2(w0, w1);
        }
    }
    if constexpr (NKB <= 4) {
#pragma unroll
        for (int it = 0; it < NKB; ++it) st4(tid + it * 256, rq[it], rk[it], rv[it], ro[it]);
    } else {
        for (int idx = tid; idx < LP * 8; idx += 256) {
            ld4(idx, rq[0], rk[0], rv[0], ro[0]);
            st4(idx, rq[0], rk[0], rv[0], ro[0]);
        }
    }
#pragma unroll
    for (int i = 0; i < (LP + 255) / 256; ++i) {
        const int key = tid + i * 256;
        if (key < LP) {
            sMask[key] = mreg[i];           // additive mask times log2(e): the softmax runs in base 2 (one v_exp_f32 per score), as in the forward kernel
            if constexpr (have) {                     // (rows beyond L: 1 / sum = 0 -> their probabilities are 0)
                const float2 sv = key < L ? *reinterpret_cast<const float2*>(stats + 2 * ((size_t)pair * L + key)) : float2{0.f, 0.f};
                sM[key] = sv.x; sLi[key] = sv.y;
            }
        }
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    auto rowfrag = [&](const unsigned char* tile, int row, int ks) {
        return *reinterpret_cast<const bf16x8*>(tile + roff(row, 2 * ks + fh));
    };
    auto pack8 = [&](const f32x16& x, int s2) {
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)x[8 * s2 + j];
        return o;
    };
    // B operand of a product contracting over tile rows: lane (lane & 31) = head-dim column `row`, slots = tile rows
    // e0 + [0, 4) and e0 + 8 + [0, 4)  (e0 = 32 blk + 16 s2 + 4 fh: the row order the accumulator-derived A operand uses)
    auto ldT = [&](const unsigned char* tile, int row, int e0) {
        if constexpr (TR) {
            // one transpose read = a [4 rows][16 columns] block per 16-lane group: lane s points at row e0 + ((s & 15) >> 2),
            // columns 16 ((s >> 4) & 1) + 4 (s & 3) of this 32-column block, and receives its own column (lane & 31) of the
            // four rows.  (`row` = db * 32 + fr: only the block index db enters the address.)
            const int r0 = e0 + ((lane & 15) >> 2);
            const int dcol = (row & ~31) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
            const bf16x4 lo = lds_read_tr16(tile + kq_off_tr(r0, dcol >> 3) + (dcol & 7) * 2);
            const bf16x4 hi = lds_read_tr16(tile + kq_off_tr(r0 + 8, dcol >> 3) + (dcol & 7) * 2);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
            return o;
        }
        const unsigned char* p = tile + row * TROW + e0 * 2;
        const bf16x4 lo = *reinterpret_cast<const bf16x4*>(p);
        const bf16x4 hi = *reinterpret_cast<const bf16x4*>(p + 16);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
        return o;
    };

    // ================= phase A: query blocks (lane = query) =================
    if constexpr (have) {
    if (phase != 1)
    for (int qb = wave; qb < NKB; qb += 4) {
        const float mq = sM[qb * 32 + fr], lq = sLi[qb * 32 + fr], dq = sD[qb * 32 + fr];
        bf16x8 fo[4], fqv[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { fo[ks] = rowfrag(sO, qb * 32 + fr, ks); fqv[ks] = rowfrag(sQ, qb * 32 + fr, ks); }
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll(NKB <= 4 ? NKB : 1)      // (longer sequences: one key block at a time -- unrolled, the scheduler hoists every block's operand reads and spills)
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 st, d;
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[r] = 0.f; d[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sK, kb * 32 + fr, ks), fqv[ks], st, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sV, kb * 32 + fr, ks), fo[ks], d, 0, 0, 0);
            }
            uint32_t wbits = 0xffffffffu;
            float sc = 1.f;
            if (dr.thresh != 0) { wbits = sBits[kb * LP + qb * 32 + fr] >> (4 * fh); sc = dr.scale; }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(st[r] * (0.125f * ATT_LOG2E) + sMask[kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh] - mq) * lq;
                const float dpr = ((wbits >> (8 * (r >> 2) + (r & 3))) & 1u) ? d[r] * sc : 0.f;
                d[r] = p * (dpr - dq);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pa = pack8(d, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, ldT(tK, db * 32 + fr, kb * 32 + 16 * s2 + 4 * fh), o[db], 0, 0, 0);
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + acc_row(r, lane);
                if (q < L) { dbase[(size_t)q * ldq + db * 32 + acc_col(lane)] = (bf16)(o[db][r] * 0.125f); cs += o[db][r] * 0.125f; }
            }
            if (dbias) {
                cs += __shfl_xor(cs, 32, 64);
                if (lane < 32) atomicAdd(&dbias[h * 64 + db * 32 + lane], cs);
            }
        }
    }
    } else
    if constexpr (TR) {
    // Long sequences: only the scores of all key blocks stay in registers (16 NKB); the dP blocks are computed twice --
    // once for D = rowsum(dP . P), once for dS -- instead of being held (another 16 NKB registers: spills at NKB >= 7).
    for (int qb = wave; qb < NKB; qb += 4) {
        f32x16 st[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sK, kb * 32 + fr, ks), rowfrag(sQ, qb * 32 + fr, ks), st[kb], 0, 0, 0);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = st[kb][r] * (0.125f * ATT_LOG2E) + sMask[kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];
                st[kb][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(st[kb][r] - mx); st[kb][r] = e; sum += e; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        bf16x8 fo[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fo[ks] = rowfrag(sO, qb * 32 + fr, ks);
        auto dp_block = [&](int kb) {      // gradient of the (dropped) probabilities of key block kb, transposed like st
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sV, kb * 32 + fr, ks), fo[ks], d, 0, 0, 0);
            if (dr.thresh != 0) {      // keep bit of (this lane's query, key kb * 32 + 8 g + 4 fh + j) = bit 8 g + 4 fh + j of the plane's word
                const uint32_t wbits = sBits[kb * LP + qb * 32 + fr] >> (4 * fh);
#pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = ((wbits >> (8 * (r >> 2) + (r & 3))) & 1u) ? d[r] * dr.scale : 0.f;
            }
            return d;
        };
        float dd = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f32x16 d = dp_block(kb);
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] *= inv; dd += st[kb][r] * d[r]; }
        }
        dd += __shfl_xor(dd, 32, 64);
        if (fh == 0) { sM[qb * 32 + fr] = mx; sLi[qb * 32 + fr] = inv; sD[qb * 32 + fr] = dd; }
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 d = dp_block(kb);
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = st[kb][r] * (d[r] - dd);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pa = pack8(d, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, ldT(tK, db * 32 + fr, kb * 32 + 16 * s2 + 4 * fh), o[db], 0, 0, 0);
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + acc_row(r, lane);
                if (q < L) { dbase[(size_t)q * ldq + db * 32 + acc_col(lane)] = (bf16)(o[db][r] * 0.125f); cs += o[db][r] * 0.125f; }
            }
            if (dbias) {
                cs += __shfl_xor(cs, 32, 64);
                if (lane < 32) atomicAdd(&dbias[h * 64 + db * 32 + lane], cs);
            }
        }
    }
    } else {
    for (int qb = wave; qb < NKB; qb += 4) {
        f32x16 st[NKB], dp[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] = 0.f; dp[kb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                st[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sK, kb * 32 + fr, ks), rowfrag(sQ, qb * 32 + fr, ks), st[kb], 0, 0, 0);
                dp[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sV, kb * 32 + fr, ks), rowfrag(sO, qb * 32 + fr, ks), dp[kb], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = st[kb][r] * (0.125f * ATT_LOG2E) + sMask[kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];
                st[kb][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = __builtin_amdgcn_exp2f(st[kb][r] - mx); st[kb][r] = e; sum += e; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (dr.thresh != 0) {       // dp is the gradient of the DROPPED probabilities: times mask / (1-p) (same mask as forward)
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                const uint32_t wbits = sBits[kb * LP + qb * 32 + fr] >> (4 * fh);
#pragma unroll
                for (int r = 0; r < 16; ++r) dp[kb][r] = ((wbits >> (8 * (r >> 2) + (r & 3))) & 1u) ? dp[kb][r] * dr.scale : 0.f;
            }
        }
        float dd = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] *= inv; dd += st[kb][r] * dp[kb][r]; }
        dd += __shfl_xor(dd, 32, 64);
        if (fh == 0) { sM[qb * 32 + fr] = mx; sLi[qb * 32 + fr] = inv; sD[qb * 32 + fr] = dd; }
        // dS^T in place of dp, then dQ = dS K / 8
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[kb][r] = st[kb][r] * (dp[kb][r] - dd);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pa = pack8(dp[kb], s2);
#pragma unroll
                for (int db = 0; db < 2; ++db)
                    o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, ldT(tK, db * 32 + fr, kb * 32 + 16 * s2 + 4 * fh), o[db], 0, 0, 0);
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + acc_row(r, lane);
                if (q < L) { dbase[(size_t)q * ldq + db * 32 + acc_col(lane)] = (bf16)(o[db][r] * 0.125f); cs += o[db][r] * 0.125f; }
            }
            if (dbias) {
                cs += __shfl_xor(cs, 32, 64);
                if (lane < 32) atomicAdd(&dbias[h * 64 + db * 32 + lane], cs);
            }
        }
    }
    }
    __syncthreads();

    // ================= phase B: key blocks (lane = key) =================
    if (phase != 0)
    for (int kb = wave; kb < NKB; kb += 4) {
        f32x16 aK[2], aV[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { aK[db][r] = 0.f; aV[db][r] = 0.f; }
        const float mk = sMask[kb * 32 + fr];
#pragma unroll 1
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 sb, db_;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sb[r] = 0.f; db_[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sb = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sQ, qb * 32 + fr, ks), rowfrag(sK, kb * 32 + fr, ks), sb, 0, 0, 0);
                db_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(rowfrag(sO, qb * 32 + fr, ks), rowfrag(sV, kb * 32 + fr, ks), db_, 0, 0, 0);
            }
            // registers 4 g .. 4 g + 3 = queries qb * 32 + 8 g + 4 fh + (0..3): their statistics (and dropout words) come as 16-byte LDS reads.
            // Dropout: lane = key, so the keep bit is bit fr of the word (this key block, query) -- one word for the 32 lanes of a half-wave (broadcast).
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int q0 = qb * 32 + 8 * g + 4 * fh;
                const float4 m4 = *reinterpret_cast<const float4*>(&sM[q0]), l4 = *reinterpret_cast<const float4*>(&sLi[q0]), d4 = *reinterpret_cast<const float4*>(&sD[q0]);
                uint4 b4 = make_uint4(0, 0, 0, 0);
                if (dr.thresh != 0) b4 = *reinterpret_cast<const uint4*>(&sBits[kb * LP + q0]);
                const float mq[4] = {m4.x, m4.y, m4.z, m4.w}, lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
                const uint32_t bq[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g + j;
                    const float p = __builtin_amdgcn_exp2f(sb[r] * (0.125f * ATT_LOG2E) + mk - mq[j]) * lq[j];
                    float wgt = 1.f;
                    if (dr.thresh != 0) wgt = ((bq[j] >> fr) & 1u) ? dr.scale : 0.f;
                    sb[r] = p * wgt;                                      // dropped probabilities -> dV
                    db_[r] = p * (db_[r] * wgt - dq[j]);                  // dS -> dK
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pp = pack8(sb, s2), pd = pack8(db_, s2);
                const int e0 = qb * 32 + 16 * s2 + 4 * fh;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    aV[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pp, ldT(tO, db * 32 + fr, e0), aV[db], 0, 0, 0);
                    aK[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pd, ldT(tQ, db * 32 + fr, e0), aK[db], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float ck = 0.f, cv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + acc_row(r, lane);
                if (key < L) {
                    dbase[(size_t)key * ldq + H + db * 32 + acc_col(lane)] = (bf16)(aK[db][r] * 0.125f);
                    dbase[(size_t)key * ldq + 2 * H + db * 32 + acc_col(lane)] = (bf16)aV[db][r];
                    ck += aK[db][r] * 0.125f; cv += aV[db][r];
                }
            }
            if (dbias) {
                ck += __shfl_xor(ck, 32, 64); cv += __shfl_xor(cv, 32, 64);
                if (lane < 32) {
                    atomicAdd(&dbias[H + h * 64 + db * 32 + lane], ck);
                    atomicAdd(&dbias[2 * H + h * 64 + db * 32 + lane], cv);
                }
            }
        }
    }
}

CPT_SWITCH(int g_attn_bwd_split, 1);       // cpt_set_tuning(38, v): two workgroups (one per phase) per (sequence, head) in small launches
void set_attn_bwd_split(int v) { CPT_SWITCH_SET(g_attn_bwd_split = v); (void)v; }
template <int NKB, bool TR = false>
static int attn_bwd_mfma_launch(const void* qkv, const int64_t* mask, const void* dctx, void* dqkv, int B, int L, int heads, const DropSpec& dr,
                                hipStream_t s, float* dbias, const void* ctx = nullptr, const float* stats = nullptr) {
    constexpr int LP = NKB * 32;
    const size_t lds = (size_t)4 * LP * 128 + (TR ? 0 : (size_t)3 * 64 * (LP * 2 + 8)) + (size_t)4 * LP * sizeof(float) + (size_t)LP * NKB * 4;      // + the dropout bit plane
    const bool have = TR && ctx != nullptr && stats != nullptr;
    auto k = have ? attn_bwd_mfma_kernel<NKB, TR, TR> : attn_bwd_mfma_kernel<NKB, TR, false>;
    static bool done[2][CPT_MAX_DEV] = {};
    const int dev = current_device_slot();
    if (lds > 64 * 1024 && !done[have ? 1 : 0][dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        done[have ? 1 : 0][dev] = true;
    }
    // two workgroups per (sequence, head), one per phase, where even those leave CUs idle (the phases run side by side instead of one after the other)
    const int split = (have && NKB <= 4 && g_attn_bwd_split && 2 * B * heads <= 256) ? 1 : 0;
    k<<<dim3(B * heads * (1 + split)), dim3(256), lds, s>>>((const bf16*)qkv, mask, (const bf16*)dctx, (bf16*)dqkv, B, L, heads, dr, dbias, (const bf16*)ctx, stats, split);
    return CPT_OK;
}

// ---- bf16x3 training (round 4): the same two-phase MFMA backward on SPLIT fp32 operands ----------------------------------------------
// qkv, dctx and dqkv are fp32.  Every tile lives in LDS twice -- bf16 hi and bf16 lo = bf16(x - hi) -- and every product a.b runs as
// hi.hi + hi.lo + lo.hi (three bf16 MFMAs, fp32 accumulate: ~2^-16 per product); probabilities and dS, which exist only in registers,
// are split there.  Row tiles only (8 x LP x 128 bytes), the operands of the products that contract over rows come through the LDS
// transpose read as in the TR variant above; scores AND dP blocks are held in registers (one workgroup per CU: 512 registers per lane).
// L <= 128.  Without it the mode's backward spent 1.2 ms per layer in the scalar fp32 kernel (44 % of the step).
struct Frag2 { bf16x8 h, l; };
__device__ __forceinline__ f32x16 mma3(const Frag2& a, const Frag2& b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.h, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.h, b.l, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.l, b.h, c, 0, 0, 0);
    return c;
}
template <int NKB>
__global__ __launch_bounds__(256, 1) void attn_bwd_x3_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ attn_mask,
                                                             const float* __restrict__ dctx, float* __restrict__ dqkv, int B, int L, int heads,
                                                             DropSpec dr, float* __restrict__ dbias) {
    constexpr int LP = NKB * 32;
    constexpr int LO = 4 * LP * 128;                   // byte distance from a hi tile to its lo tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sQ = smem;                          // [LP][64] bf16, swizzled rows (kq_off_tr); hi tiles Q K V dO, then the four lo tiles
    unsigned char* sK = sQ + LP * 128;
    unsigned char* sV = sK + LP * 128;
    unsigned char* sO = sV + LP * 128;
    float* sMask = reinterpret_cast<float*>(smem + 2 * LO);
    float* sM = sMask + LP;
    float* sLi = sM + LP;
    float* sD = sLi + LP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int H = heads * 64;
    const size_t ldq = (size_t)3 * H;
    const float* base = qkv + (size_t)b * L * ldq + h * 64;
    float* dbase = dqkv + (size_t)b * L * ldq + h * 64;

    auto put = [&](unsigned char* tile, int r, int c, const float* src) {      // 8 fp32 -> hi chunk + lo chunk
        bf16x8 hi, lo;
        if (src) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = j < 4 ? x0[j] : x1[j - 4];
                const bf16 hh = (bf16)x;
                hi[j] = hh; lo[j] = (bf16)(x - (float)hh);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = (bf16)0.f; lo[j] = (bf16)0.f; }
        }
        *reinterpret_cast<bf16x8*>(tile + kq_off_tr(r, c)) = hi;
        *reinterpret_cast<bf16x8*>(tile + LO + kq_off_tr(r, c)) = lo;
    };
    for (int idx = tid; idx < LP * 8; idx += 256) {
        const int r = idx >> 3, c = idx & 7;
        const bool in = r < L;
        put(sQ, r, c, in ? base + (size_t)r * ldq + c * 8 : nullptr);
        put(sK, r, c, in ? base + (size_t)r * ldq + H + c * 8 : nullptr);
        put(sV, r, c, in ? base + (size_t)r * ldq + 2 * H + c * 8 : nullptr);
        put(sO, r, c, in ? dctx + ((size_t)b * L + r) * H + h * 64 + c * 8 : nullptr);
    }
    for (int key = tid; key < LP; key += 256) {
        float mv = -INFINITY;
        if (key < L) mv = attn_mask ? (1.0f - (float)attn_mask[(size_t)b * L + key]) * -10000.0f : 0.f;
        sMask[key] = mv;
    }
    __syncthreads();

    const int fr = lane & 31, fh = lane >> 5;
    auto rowfrag = [&](const unsigned char* tile, int row, int ks) {
        Frag2 f;
        f.h = *reinterpret_cast<const bf16x8*>(tile + kq_off_tr(row, 2 * ks + fh));
        f.l = *reinterpret_cast<const bf16x8*>(tile + LO + kq_off_tr(row, 2 * ks + fh));
        return f;
    };
    auto pack8 = [&](const f32x16& x, int s2) {
        Frag2 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[8 * s2 + j];
            const bf16 hh = (bf16)v;
            f.h[j] = hh; f.l[j] = (bf16)(v - (float)hh);
        }
        return f;
    };
    auto ldT1 = [&](const unsigned char* tile, int row, int e0) {
        const int r0 = e0 + ((lane & 15) >> 2);
        const int dcol = (row & ~31) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const bf16x4 lo = lds_read_tr16(tile + kq_off_tr(r0, dcol >> 3) + (dcol & 7) * 2);
        const bf16x4 hi = lds_read_tr16(tile + kq_off_tr(r0 + 8, dcol >> 3) + (dcol & 7) * 2);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
        return o;
    };
    auto ldT = [&](const unsigned char* tile, int row, int e0) {
        Frag2 f;
        f.h = ldT1(tile, row, e0);
        f.l = ldT1(tile + LO, row, e0);
        return f;
    };

    // ================= phase A: query blocks (lane = query) =================
    for (int qb = wave; qb < NKB; qb += 4) {
        f32x16 st[NKB], dp[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] = 0.f; dp[kb][r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                st[kb] = mma3(rowfrag(sK, kb * 32 + fr, ks), rowfrag(sQ, qb * 32 + fr, ks), st[kb]);
                dp[kb] = mma3(rowfrag(sV, kb * 32 + fr, ks), rowfrag(sO, qb * 32 + fr, ks), dp[kb]);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = st[kb][r] * 0.125f + sMask[kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];
                st[kb][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = expf(st[kb][r] - mx); st[kb][r] = e; sum += e; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        if (dr.thresh != 0) {
            const int qd = min(qb * 32 + fr, L - 1);
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bool keep[4];
                    drop_attn_row4(dr, (uint32_t)blockIdx.x, qd, kb * 8 + 2 * g + fh, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dp[kb][4 * g + j] = keep[j] ? dp[kb][4 * g + j] * dr.scale : 0.f;
                }
        }
        float dd = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] *= inv; dd += st[kb][r] * dp[kb][r]; }
        dd += __shfl_xor(dd, 32, 64);
        if (fh == 0) { sM[qb * 32 + fr] = mx; sLi[qb * 32 + fr] = inv; sD[qb * 32 + fr] = dd; }
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dp[kb][r] = st[kb][r] * (dp[kb][r] - dd);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const Frag2 pa = pack8(dp[kb], s2);
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = mma3(pa, ldT(sK, db * 32 + fr, kb * 32 + 16 * s2 + 4 * fh), o[db]);
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + acc_row(r, lane);
                if (q < L) { dbase[(size_t)q * ldq + db * 32 + acc_col(lane)] = o[db][r] * 0.125f; cs += o[db][r] * 0.125f; }
            }
            if (dbias) {
                cs += __shfl_xor(cs, 32, 64);
                if (lane < 32) atomicAdd(&dbias[h * 64 + db * 32 + lane], cs);
            }
        }
    }
    __syncthreads();

    // ================= phase B: key blocks (lane = key) =================
    for (int kb = wave; kb < NKB; kb += 4) {
        f32x16 aK[2], aV[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { aK[db][r] = 0.f; aV[db][r] = 0.f; }
        const float mk = sMask[kb * 32 + fr];
#pragma unroll 1
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 sb, db_;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sb[r] = 0.f; db_[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sb = mma3(rowfrag(sQ, qb * 32 + fr, ks), rowfrag(sK, kb * 32 + fr, ks), sb);
                db_ = mma3(rowfrag(sO, qb * 32 + fr, ks), rowfrag(sV, kb * 32 + fr, ks), db_);
            }
            float wgt[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) wgt[r] = 1.f;
            if (dr.thresh != 0) {
                const int kd = min(kb * 32 + fr, L - 1);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bool keep[4];
                    drop_attn_col4(dr, (uint32_t)blockIdx.x, qb * 8 + 2 * g + fh, kd, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wgt[4 * g + j] = keep[j] ? dr.scale : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float p = expf(sb[r] * 0.125f + mk - sM[q]) * sLi[q];
                sb[r] = p * wgt[r];
                db_[r] = p * (db_[r] * wgt[r] - sD[q]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const Frag2 pp = pack8(sb, s2), pd = pack8(db_, s2);
                const int e0 = qb * 32 + 16 * s2 + 4 * fh;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    aV[db] = mma3(pp, ldT(sO, db * 32 + fr, e0), aV[db]);
                    aK[db] = mma3(pd, ldT(sQ, db * 32 + fr, e0), aK[db]);
                }
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float ck = 0.f, cv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + acc_row(r, lane);
                if (key < L) {
                    dbase[(size_t)key * ldq + H + db * 32 + acc_col(lane)] = aK[db][r] * 0.125f;
                    dbase[(size_t)key * ldq + 2 * H + db * 32 + acc_col(lane)] = aV[db][r];
                    ck += aK[db][r] * 0.125f; cv += aV[db][r];
                }
            }
            if (dbias) {
                ck += __shfl_xor(ck, 32, 64); cv += __shfl_xor(cv, 32, 64);
                if (lane < 32) {
                    atomicAdd(&dbias[H + h * 64 + db * 32 + lane], ck);
                    atomicAdd(&dbias[2 * H + h * 64 + db * 32 + lane], cv);
                }
            }
        }
    }
}

// Long sequences (128 < L <= 288: the GQA / VCR few-shot lengths): eight split tiles do not fit, so each phase keeps only the two tensors it
// reads ACROSS blocks in LDS (hi + lo: LP x 512 bytes = 147 KB at 288) -- K, V in phase A, then Q, dO in phase B -- and takes its OWN 32-row block of
// the other two straight from global memory into registers (a row fragment is 32 contiguous bytes of fp32 per lane and k-step).  dP blocks are
// computed twice as in the bf16 transpose-read variant (scores of all key blocks stay in registers: 16 NKB).
template <int NKB>
__global__ __launch_bounds__(256, 1) void attn_bwd_x3_long_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ attn_mask,
                                                                  const float* __restrict__ dctx, float* __restrict__ dqkv, int B, int L, int heads,
                                                                  DropSpec dr, float* __restrict__ dbias) {
    constexpr int LP = NKB * 32;
    constexpr int LO = 2 * LP * 128;                   // byte distance from a hi tile to its lo tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                          // phase A: K, phase B: Q    ([LP][64] bf16 rows, kq_off_tr swizzle; lo copy at + LO)
    unsigned char* sY = sX + LP * 128;                 // phase A: V, phase B: dO
    float* sMask = reinterpret_cast<float*>(smem + 2 * LO);
    float* sM = sMask + LP;
    float* sLi = sM + LP;
    float* sD = sLi + LP;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / heads, h = blockIdx.x % heads;
    const int H = heads * 64;
    const size_t ldq = (size_t)3 * H;
    const float* base = qkv + (size_t)b * L * ldq + h * 64;
    const float* obase = dctx + (size_t)b * L * H + h * 64;
    float* dbase = dqkv + (size_t)b * L * ldq + h * 64;
    const int fr = lane & 31, fh = lane >> 5;

    auto split8 = [&](const float* src, bf16x8& hi, bf16x8& lo) {          // src == nullptr: zeros (rows beyond L)
        if (src) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = j < 4 ? x0[j] : x1[j - 4];
                const bf16 hh = (bf16)x;
                hi[j] = hh; lo[j] = (bf16)(x - (float)hh);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = (bf16)0.f; lo[j] = (bf16)0.f; }
        }
    };
    auto fill = [&](const float* x, size_t ldx, const float* y, size_t ldy) {    // rows of two tensors -> sX / sY (hi + lo)
        for (int idx = tid; idx < LP * 8; idx += 256) {
            const int r = idx >> 3, c = idx & 7;
            bf16x8 hi, lo;
            split8(r < L ? x + (size_t)r * ldx + c * 8 : nullptr, hi, lo);
            *reinterpret_cast<bf16x8*>(sX + kq_off_tr(r, c)) = hi;
            *reinterpret_cast<bf16x8*>(sX + LO + kq_off_tr(r, c)) = lo;
            split8(r < L ? y + (size_t)r * ldy + c * 8 : nullptr, hi, lo);
            *reinterpret_cast<bf16x8*>(sY + kq_off_tr(r, c)) = hi;
            *reinterpret_cast<bf16x8*>(sY + LO + kq_off_tr(r, c)) = lo;
        }
    };
    // this lane's fragments of row `row` of a global tensor: what rowfrag would read from a tile holding it
    auto ownfrag = [&](const float* x, size_t ldx, int row, Frag2 (&f)[4]) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) split8(row < L ? x + (size_t)row * ldx + (2 * ks + fh) * 8 : nullptr, f[ks].h, f[ks].l);
    };
    auto rowfrag = [&](const unsigned char* tile, int row, int ks) {
        Frag2 f;
        f.h = *reinterpret_cast<const bf16x8*>(tile + kq_off_tr(row, 2 * ks + fh));
        f.l = *reinterpret_cast<const bf16x8*>(tile + LO + kq_off_tr(row, 2 * ks + fh));
        return f;
    };
    auto pack8 = [&](const f32x16& x, int s2) {
        Frag2 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[8 * s2 + j];
            const bf16 hh = (bf16)v;
            f.h[j] = hh; f.l[j] = (bf16)(v - (float)hh);
        }
        return f;
    };
    auto ldT1 = [&](const unsigned char* tile, int row, int e0) {
        const int r0 = e0 + ((lane & 15) >> 2);
        const int dcol = (row & ~31) + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const bf16x4 lo = lds_read_tr16(tile + kq_off_tr(r0, dcol >> 3) + (dcol & 7) * 2);
        const bf16x4 hi = lds_read_tr16(tile + kq_off_tr(r0 + 8, dcol >> 3) + (dcol & 7) * 2);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = lo[j]; o[4 + j] = hi[j]; }
        return o;
    };
    auto ldT = [&](const unsigned char* tile, int row, int e0) {
        Frag2 f;
        f.h = ldT1(tile, row, e0);
        f.l = ldT1(tile + LO, row, e0);
        return f;
    };

    // ================= phase A: K, V in LDS; query blocks (lane = query), Q and dO rows from global =================
    fill(base + H, ldq, base + 2 * H, ldq);
    for (int key = tid; key < LP; key += 256) {
        float mv = -INFINITY;
        if (key < L) mv = attn_mask ? (1.0f - (float)attn_mask[(size_t)b * L + key]) * -10000.0f : 0.f;
        sMask[key] = mv;
    }
    __syncthreads();
    for (int qb = wave; qb < NKB; qb += 4) {
        Frag2 fq[4], fo[4];
        ownfrag(base, ldq, qb * 32 + fr, fq);
        ownfrag(obase, (size_t)H, qb * 32 + fr, fo);
        f32x16 st[NKB];
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) st[kb] = mma3(rowfrag(sX, kb * 32 + fr, ks), fq[ks], st[kb]);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = st[kb][r] * 0.125f + sMask[kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh];
                st[kb][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float e = expf(st[kb][r] - mx); st[kb][r] = e; sum += e; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const int qd = min(qb * 32 + fr, L - 1);
        auto dp_block = [&](int kb) {
            f32x16 d;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) d = mma3(rowfrag(sY, kb * 32 + fr, ks), fo[ks], d);
            if (dr.thresh != 0) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bool keep[4];
                    drop_attn_row4(dr, (uint32_t)blockIdx.x, qd, kb * 8 + 2 * g + fh, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[4 * g + j] = keep[j] ? d[4 * g + j] * dr.scale : 0.f;
                }
            }
            return d;
        };
        float dd = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            const f32x16 d = dp_block(kb);
#pragma unroll
            for (int r = 0; r < 16; ++r) { st[kb][r] *= inv; dd += st[kb][r] * d[r]; }
        }
        dd += __shfl_xor(dd, 32, 64);
        if (fh == 0) { sM[qb * 32 + fr] = mx; sLi[qb * 32 + fr] = inv; sD[qb * 32 + fr] = dd; }
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            f32x16 d = dp_block(kb);
#pragma unroll
            for (int r = 0; r < 16; ++r) d[r] = st[kb][r] * (d[r] - dd);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const Frag2 pa = pack8(d, s2);
#pragma unroll
                for (int db = 0; db < 2; ++db) o[db] = mma3(pa, ldT(sX, db * 32 + fr, kb * 32 + 16 * s2 + 4 * fh), o[db]);
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float cs = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + acc_row(r, lane);
                if (q < L) { dbase[(size_t)q * ldq + db * 32 + acc_col(lane)] = o[db][r] * 0.125f; cs += o[db][r] * 0.125f; }
            }
            if (dbias) {
                cs += __shfl_xor(cs, 32, 64);
                if (lane < 32) atomicAdd(&dbias[h * 64 + db * 32 + lane], cs);
            }
        }
    }
    __syncthreads();

    // ================= phase B: Q, dO in LDS; key blocks (lane = key), K and V rows from global =================
    fill(base, ldq, obase, (size_t)H);
    __syncthreads();
    for (int kb = wave; kb < NKB; kb += 4) {
        Frag2 fk[4], fv[4];
        ownfrag(base + H, ldq, kb * 32 + fr, fk);
        ownfrag(base + 2 * H, ldq, kb * 32 + fr, fv);
        f32x16 aK[2], aV[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) { aK[db][r] = 0.f; aV[db][r] = 0.f; }
        const float mk = sMask[kb * 32 + fr];
#pragma unroll 1
        for (int qb = 0; qb < NKB; ++qb) {
            f32x16 sb, db_;
#pragma unroll
            for (int r = 0; r < 16; ++r) { sb[r] = 0.f; db_[r] = 0.f; }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                sb = mma3(rowfrag(sX, qb * 32 + fr, ks), fk[ks], sb);
                db_ = mma3(rowfrag(sY, qb * 32 + fr, ks), fv[ks], db_);
            }
            float wgt[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) wgt[r] = 1.f;
            if (dr.thresh != 0) {
                const int kd = min(kb * 32 + fr, L - 1);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bool keep[4];
                    drop_attn_col4(dr, (uint32_t)blockIdx.x, qb * 8 + 2 * g + fh, kd, keep);
#pragma unroll
                    for (int j = 0; j < 4; ++j) wgt[4 * g + j] = keep[j] ? dr.scale : 0.f;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qb * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float p = expf(sb[r] * 0.125f + mk - sM[q]) * sLi[q];
                sb[r] = p * wgt[r];
                db_[r] = p * (db_[r] * wgt[r] - sD[q]);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const Frag2 pp = pack8(sb, s2), pd = pack8(db_, s2);
                const int e0 = qb * 32 + 16 * s2 + 4 * fh;
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    aV[db] = mma3(pp, ldT(sY, db * 32 + fr, e0), aV[db]);
                    aK[db] = mma3(pd, ldT(sX, db * 32 + fr, e0), aK[db]);
                }
            }
        }
#pragma unroll
        for (int db = 0; db < 2; ++db) {
            float ck = 0.f, cv = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = kb * 32 + acc_row(r, lane);
                if (key < L) {
                    dbase[(size_t)key * ldq + H + db * 32 + acc_col(lane)] = aK[db][r] * 0.125f;
                    dbase[(size_t)key * ldq + 2 * H + db * 32 + acc_col(lane)] = aV[db][r];
                    ck += aK[db][r] * 0.125f; cv += aV[db][r];
                }
            }
            if (dbias) {
                ck += __shfl_xor(ck, 32, 64); cv += __shfl_xor(cv, 32, 64);
                if (lane < 32) {
                    atomicAdd(&dbias[H + h * 64 + db * 32 + lane], ck);
                    atomicAdd(&dbias[2 * H + h * 64 + db * 32 + lane], cv);
                }
            }
        }
    }
}

template <int NKB>
static int attn_bwd_x3_long_launch(const float* qkv, const int64_t* mask, const float* dctx, float* dqkv, int B, int L, int heads, const DropSpec& dr,
                                   hipStream_t s, float* dbias) {
    constexpr int LP = NKB * 32;
    const size_t lds = (size_t)4 * LP * 128 + (size_t)4 * LP * sizeof(float);
    auto k = attn_bwd_x3_long_kernel<NKB>;
    static bool done = false;
    if (lds > 64 * 1024 && !done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        done = true;
    }
    k<<<dim3(B * heads), dim3(256), lds, s>>>(qkv, mask, dctx, dqkv, B, L, heads, dr, dbias);
    return CPT_OK;
}

template <int NKB>
static int attn_bwd_x3_launch(const float* qkv, const int64_t* mask, const float* dctx, float* dqkv, int B, int L, int heads, const DropSpec& dr,
                              hipStream_t s, float* dbias) {
    constexpr int LP = NKB * 32;
    const size_t lds = (size_t)8 * LP * 128 + (size_t)4 * LP * sizeof(float);
    auto k = attn_bwd_x3_kernel<NKB>;
    static bool done = false;
    if (lds > 64 * 1024 && !done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return CPT_ERR_HIP - (int)e;
        done = true;
    }
    k<<<dim3(B * heads), dim3(256), lds, s>>>(qkv, mask, dctx, dqkv, B, L, heads, dr, dbias);
    return CPT_OK;
}
int attention_bwd_x3_supported(int L, int mask_3d) { return !mask_3d && L > 0 && L <= 288; }
int attention_bwd_x3(const float* qkv, const int64_t* attn_mask, const float* dctx, float* dqkv, int B, int L, int heads, hipStream_t s,
                     const DropSpec* drop, float* dbias) {
    if (B <= 0 || heads <= 0 || !attention_bwd_x3_supported(L, 0)) return CPT_ERR_SHAPE;
    if (!qkv || !dctx || !dqkv) return CPT_ERR_NULL;
    if ((((uintptr_t)qkv | (uintptr_t)dctx) & 15)) return CPT_ERR_ALIGN;
    const DropSpec dr = drop ? *drop : DropSpec{};
    if (L <= 32) return attn_bwd_x3_launch<1>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
    if (L <= 64) return attn_bwd_x3_launch<2>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
    if (L <= 96) return attn_bwd_x3_launch<3>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
    if (L <= 128) return attn_bwd_x3_launch<4>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
    if (L <= 160) return attn_bwd_x3_long_launch<5>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
    if (L <= 224) return attn_bwd_x3_long_launch<7>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
    return attn_bwd_x3_long_launch<9>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
}

CPT_SWITCH(int g_attn_bwd_variant, 1);      // 1: transpose-read MFMA kernels for bf16 (L <= 288); 0: generic kernel always; 2: L <= 128 through the older MFMA kernel with transposed tile copies
void set_attn_bwd_variant(int v) { CPT_SWITCH_SET(g_attn_bwd_variant = v); (void)v; }

static size_t attn_bwd_vg_lds(int L, int has_drop) { return ((size_t)L * 65 + 2 * 16 * 65 + (has_drop ? 3 : 2) * 16 * (L + 1) + L) * sizeof(float); }
// Whether attention_bwd has a kernel for (dtype, L, dropout on the probabilities): asked by cpt_train_fwd so that an unsupported
// combination is rejected BEFORE the forward runs, not after it
int attention_bwd_supported(int dtype, int L, int has_drop, int mask_3d) {
    if (L <= 0) return 0;
    if (!mask_3d && dtype == CPT_BF16 && g_attn_bwd_variant != 0 && L <= 288) return 1;
    if (dtype != CPT_BF16 && dtype != CPT_F32) return 0;
    const size_t lds = ((size_t)2 * L * 65 + 2 * AB_QB * 65 + (has_drop ? 3 : 2) * AB_QB * (L + 1) + L) * sizeof(float);
    if (lds <= 160 * 1024) return 1;
    return L <= 288 && attn_bwd_vg_lds(L, has_drop) <= 160 * 1024;      // V from global memory, 16-query blocks
}

int attention_bwd(int dtype, const void* qkv, const int64_t* attn_mask, const void* dctx, void* dqkv, int B, int L, int heads, hipStream_t s,
                  const DropSpec* drop, float* dbias, int mask_3d, const void* ctx, const float* stats) {
    if (B <= 0 || L <= 0 || heads <= 0) return CPT_ERR_SHAPE;
    const DropSpec dr = drop ? *drop : DropSpec{};
    // a [B][L][L] mask (one row per query) goes through the generic kernels, which read it per score; the MFMA kernels keep one
    // additive mask value per key in LDS
    const bool mfma_ok = !(mask_3d && attn_mask);
    // L <= 128: the transpose-read kernel needs 68 KB of LDS and 209 registers -> two workgroups per CU (B * heads = 384 workgroups
    // in one round instead of two): 43.6 vs 51.2 us at B = 32 (rocprofv3)
    if (mfma_ok && dtype == CPT_BF16 && g_attn_bwd_variant == 1 && L > 64 && L <= 128) return attn_bwd_mfma_launch<4, true>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias, ctx, stats);
    if (mfma_ok && dtype == CPT_BF16 && g_attn_bwd_variant != 0 && L <= 128) {
        if (L <= 32) return attn_bwd_mfma_launch<1>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
        if (L <= 64) return attn_bwd_mfma_launch<2>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
        return attn_bwd_mfma_launch<4>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias);
    }
    if (mfma_ok && dtype == CPT_BF16 && g_attn_bwd_variant != 0 && L <= 288) {      // GQA / VCR shapes: transpose-read variant
        if (L <= 160) return attn_bwd_mfma_launch<5, true>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias, ctx, stats);
        if (L <= 224) return attn_bwd_mfma_launch<7, true>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias, ctx, stats);
        return attn_bwd_mfma_launch<9, true>(qkv, attn_mask, dctx, dqkv, B, L, heads, dr, s, dbias, ctx, stats);
    }
    size_t lds = ((size_t)2 * L * 65 + 2 * AB_QB * 65 + (dr.thresh ? 3 : 2) * AB_QB * (L + 1) + L) * sizeof(float);
    const bool vg = lds > 160 * 1024;                      // beyond L ~ 176: the form that reads V from global memory (L <= 288)
    if (vg) {
        lds = attn_bwd_vg_lds(L, dr.thresh ? 1 : 0);
        if (L > 288 || lds > 160 * 1024) return CPT_ERR_SHAPE;
    }
    dim3 grid(B * heads), block(256);
    hipError_t e;
#define ABKV(TT)                                                                                                      \
    do {                                                                                                              \
        auto k = attn_bwd_kernel<TT, 72, 16, true>;                                                                   \
        if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)             \
            return CPT_ERR_HIP - (int)e;                                                                              \
        k<<<grid, block, lds, s>>>((const TT*)qkv, attn_mask, (const TT*)dctx, (TT*)dqkv, B, L, heads, dr, mask_3d);  \
    } while (0)
    if (vg) {
        if (dtype == CPT_BF16) ABKV(bf16); else if (dtype == CPT_F32) ABKV(float); else return CPT_ERR_DTYPE;
        if (dbias) return colsum(dqkv, dtype, 3 * heads * 64, dbias, B * L, 3 * heads * 64, s);
        return CPT_OK;
    }
#define ABK(TT, ME)                                                                                                   \
    do {                                                                                                              \
        auto k = attn_bwd_kernel<TT, ME>;                                                                             \
        if (lds > 64 * 1024 && (e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),                             \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)             \
            return CPT_ERR_HIP - (int)e;                                                                              \
        k<<<grid, block, lds, s>>>((const TT*)qkv, attn_mask, (const TT*)dctx, (TT*)dqkv, B, L, heads, dr, mask_3d);  \
    } while (0)
    if (dtype == CPT_BF16) { if (L <= 128) ABK(bf16, 32); else ABK(bf16, 44); }
    else if (dtype == CPT_F32) { if (L <= 128) ABK(float, 32); else ABK(float, 44); }
    else return CPT_ERR_DTYPE;
#undef ABK
#undef ABKV
    if (dbias) return colsum(dqkv, dtype, 3 * heads * 64, dbias, B * L, 3 * heads * 64, s);      // the generic kernels leave the bias sums to a pass of their own
    return CPT_OK;
}

// ---- fused AdamW over the flat parameter buffer (torch.optim.AdamW single-tensor update) ---------
// code[i]: 0 = parameter has no gradient (skipped, as torch skips grad=None), 1 = weight decay, 2 = no decay
// HF (round 5): the arithmetic of pytorch_transformers.AdamW (transformers@067923d optimization.py, the optimizer of the GQA / VCR few-shot drivers,
// fewshot/vcr_nsp_cpt.py:385, gqa_cpt.py:342) instead of torch.optim.AdamW's: eps is added to sqrt(v) BEFORE the bias corrections scale the step
// (step = lr sqrt(bc2) / bc1, or lr without correct_bias: bc1 = bc2_sqrt = 1 then), and the decoupled decay p -= lr wd p acts on the UPDATED parameter.
// Scalars (round 6, ABI 7): everything derived from lr / betas / eps / weight decay is formed in DOUBLE on the host and rounded to fp32 once, exactly where torch
// rounds the Python-float scalars it multiplies into fp32 tensors (1 - beta2 in fp32 from a fp32 beta2 was 1.3e-5 off):
//   torch.optim.AdamW (single tensor):  p *= decay = 1 - lr wd;  m = lerp(m, g, om1);  v = v b2 + (om2 g) g;  p -= step (m / (sqrt(v) / bc2s + eps)),  step = lr / bc1
//   pytorch_transformers.AdamW:         m = m b1 + om1 g;  v = v b2 + (om2 g) g;  p -= step (m / (sqrt(v) + eps)),  step = lr sqrt(bc2) / bc1;  p += nlw p,  nlw = -lr wd
struct AdamScalars { float b1, b2, om1, om2, eps, step, bc2s, decay, nlw; };
template <bool SHADOW, bool HF = false>
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, const unsigned char* __restrict__ code,
                                                    bf16* __restrict__ shadow, size_t n, AdamScalars a, float grad_scale) {
    const size_t i0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= n) return;
    const f32x4 pv = *reinterpret_cast<const f32x4*>(p + i0);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i0);
    f32x4 mv = *reinterpret_cast<const f32x4*>(m + i0);
    f32x4 vv = *reinterpret_cast<const f32x4*>(v + i0);
    const uchar4 cv = *reinterpret_cast<const uchar4*>(code + i0);
    const unsigned char cc[4] = {cv.x, cv.y, cv.z, cv.w};
    f32x4 po = pv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (cc[e] == 0) continue;
        const float gr = gv[e] * grad_scale;
        if constexpr (HF) {
            mv[e] = mv[e] * a.b1 + a.om1 * gr;                             // exp_avg.mul_(beta1).add_(1 - beta1, grad)
            vv[e] = vv[e] * a.b2 + (a.om2 * gr) * gr;                      // exp_avg_sq.mul_(beta2).addcmul_(1 - beta2, grad, grad)
            const float denom = sqrtf(vv[e]) + a.eps;
            float pe = pv[e] - a.step * (mv[e] / denom);                   // p.addcdiv_(-step_size, exp_avg, denom)
            if (cc[e] == 1) pe = pe + a.nlw * pe;                          // p.add_(-lr * weight_decay, p)
            po[e] = pe;
            continue;
        }
        float pe = cc[e] == 1 ? pv[e] * a.decay : pv[e];                   // param.mul_(1 - lr * weight_decay)
        mv[e] = mv[e] + (gr - mv[e]) * a.om1;                              // exp_avg.lerp_(grad, 1 - beta1)
        vv[e] = vv[e] * a.b2 + (a.om2 * gr) * gr;                          // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        const float denom = sqrtf(vv[e]) / a.bc2s + a.eps;                 // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
        pe = pe - a.step * (mv[e] / denom);                                // param.addcdiv_(exp_avg, denom, value=-step_size)
        po[e] = pe;
    }
    *reinterpret_cast<f32x4*>(p + i0) = po;
    *reinterpret_cast<f32x4*>(m + i0) = mv;
    *reinterpret_cast<f32x4*>(v + i0) = vv;
    if (SHADOW) {
        bf16x4 s4;
#pragma unroll
        for (int e = 0; e < 4; ++e) s4[e] = (bf16)po[e];
        *reinterpret_cast<bf16x4*>(shadow + i0) = s4;
    }
}

int adamw_flat(float* p, const float* g, float* m, float* v, const unsigned char* code, void* shadow_bf16, size_t n,
               double lr, double beta1, double beta2, double eps, double wd, int step, float grad_scale, hipStream_t s, int flags) {
    if (!p || !g || !m || !v || !code) return CPT_ERR_NULL;
    if (n % 4 || step < 1 || (flags & ~3)) return CPT_ERR_SHAPE;
    const bool hf = flags & 1, no_bc = flags & 2;
    const double bc1 = no_bc ? 1.0 : 1.0 - pow(beta1, (double)step);
    const double bc2 = no_bc ? 1.0 : 1.0 - pow(beta2, (double)step);
    AdamScalars a;
    a.b1 = (float)beta1; a.b2 = (float)beta2; a.om1 = (float)(1.0 - beta1); a.om2 = (float)(1.0 - beta2); a.eps = (float)eps;
    a.bc2s = (float)sqrt(bc2);
    a.step = hf ? (float)(lr * sqrt(bc2) / bc1) : (float)(lr / bc1);
    a.decay = (float)(1.0 - lr * wd);
    a.nlw = (float)(-lr * wd);
    dim3 grid((unsigned)((n / 4 + 255) / 256)), block(256);
    if (hf) {
        if (shadow_bf16) adamw_kernel<true, true><<<grid, block, 0, s>>>(p, g, m, v, code, (bf16*)shadow_bf16, n, a, grad_scale);
        else adamw_kernel<false, true><<<grid, block, 0, s>>>(p, g, m, v, code, nullptr, n, a, grad_scale);
        return CPT_OK;
    }
    if (shadow_bf16) adamw_kernel<true><<<grid, block, 0, s>>>(p, g, m, v, code, (bf16*)shadow_bf16, n, a, grad_scale);
    else adamw_kernel<false><<<grid, block, 0, s>>>(p, g, m, v, code, nullptr, n, a, grad_scale);
    return CPT_OK;
}

}  // namespace cpt
