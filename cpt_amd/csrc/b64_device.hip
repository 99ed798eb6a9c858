// Device side of the region-feature wire format (include/cpt_io.h; round 5): the base64 TEXT of a batch's regions
// travels to the GPU as it stands in the predictions file and is decoded there -- byte work at the memory system's rate instead of on the host's
// cores, which then only locate the strings (memchr) and copy them into the pinned ring (cpt_pack_tsv_rows, io_decode.hip).
// Replaces np.frombuffer(base64.b64decode(s), np.float32) + np.stack + zero padding (Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:161-180, :119-120);
// the same strictness and the same bytes as the host decoder (cpt_b64_decode_f32).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cpt_hip.h"
#include "cpt_io.h"

namespace cpt { int abi_fail(int code, const char* fmt, ...); }

namespace cpt {

// One workgroup per region slot (sequence s, region i).  text: [n_seq][max_regions][chars] (chars = 4 ceil(4 dim / 3), a multiple of 4: every
// slot starts on a 4-byte boundary), mask: [n_seq][max_regions] int64 (0: no region -> the slot's row of out is zero), out: [n_seq][max_regions][dim].
// A thread turns 16 characters into 12 bytes = three floats per step; thread 0 finishes the slot's last one to three groups and its padding.
// First error (lowest slot, then lowest character): err = ~((slot << 32) | character) by atomicMax on a word that holds 0 while there is none.
__global__ __launch_bounds__(256) void b64_regions_kernel(const unsigned* __restrict__ text, const int64_t* __restrict__ mask, unsigned* __restrict__ out,
                                                          int dim, int words, unsigned long long* __restrict__ err) {
    __shared__ unsigned char lut[256];      // 0..63: value; 64: '='; 0xff: outside the alphabet
    {
        const int c = threadIdx.x;
        unsigned char v = 0xff;
        if (c >= 'A' && c <= 'Z') v = (unsigned char)(c - 'A');
        else if (c >= 'a' && c <= 'z') v = (unsigned char)(c - 'a' + 26);
        else if (c >= '0' && c <= '9') v = (unsigned char)(c - '0' + 52);
        else if (c == '+') v = 62;
        else if (c == '/') v = 63;
        else if (c == '=') v = 64;
        lut[c] = v;
    }
    __syncthreads();
    const size_t slot = blockIdx.x;
    unsigned* o = out + slot * (size_t)dim;
    if (mask[slot] == 0) {
        for (int c = threadIdx.x; c < dim; c += 256) o[c] = 0u;
        return;
    }
    const unsigned* t = text + slot * (size_t)words;
    const int nbytes = dim * 4, full_groups = nbytes / 3, rem = nbytes - full_groups * 3;
    const int chunks = full_groups >> 2;
    int bad_at = -1;                        // first character of this thread's first group with a character outside the alphabet
    int pad_bad = -1;                       // (thread 0) the character of the padded group that breaks its form
    auto group = [&](unsigned w, int at) -> unsigned {
        const unsigned a = lut[w & 255u], b = lut[(w >> 8) & 255u], c = lut[(w >> 16) & 255u], d = lut[w >> 24];
        if (((a | b | c | d) & 0xc0u) && bad_at < 0) bad_at = at;
        return (a << 18) | (b << 12) | (c << 6) | d;
    };
    // (slots start on 4-byte boundaries only: one 16-byte load / one 12-byte store from a 4-byte-aligned address each -- legal, the HSA target runs in
    // unaligned-access mode -- instead of four and three 4-byte ones)
    typedef unsigned u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef unsigned u32x3_a4 __attribute__((ext_vector_type(3), aligned(4)));
    // (round 6: a thread's chunks three at a time with all three 16-byte loads in flight before the first table lookup -- float32[2054] is 684 chunks,
    // 2.7 per thread: the one-at-a-time loop ran each thread's load -> lookup -> store chain three times in sequence)
    for (int c0 = threadIdx.x; c0 < chunks; c0 += 3 * 256) {
        u32x4_a4 wv[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = c0 + u * 256;
            if (c < chunks) wv[u] = *reinterpret_cast<const u32x4_a4*>(t + 4 * c);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int c = c0 + u * 256;
            if (c >= chunks) continue;
            const u32x4_a4 w = wv[u];
            const unsigned t0 = group(w[0], 16 * c), t1 = group(w[1], 16 * c + 4), t2 = group(w[2], 16 * c + 8), t3 = group(w[3], 16 * c + 12);
            // bytes in memory order: t0[23:16] t0[15:8] t0[7:0] t1[23:16] ...
            u32x3_a4 ov;
            ov[0] = ((t0 >> 16) & 255u) | (((t0 >> 8) & 255u) << 8) | ((t0 & 255u) << 16) | (((t1 >> 16) & 255u) << 24);
            ov[1] = ((t1 >> 8) & 255u) | ((t1 & 255u) << 8) | (((t2 >> 16) & 255u) << 16) | (((t2 >> 8) & 255u) << 24);
            ov[2] = (t2 & 255u) | (((t3 >> 16) & 255u) << 8) | (((t3 >> 8) & 255u) << 16) | ((t3 & 255u) << 24);
            *reinterpret_cast<u32x3_a4*>(o + 3 * c) = ov;
        }
    }
    if (threadIdx.x == 0) {
        // the groups behind the last whole chunk: up to three full ones, then the padded one (rem = 1: "xx==", rem = 2: "xxx=")
        unsigned char by[12];
        int nb = 0;
        for (int g = chunks * 4; g < full_groups; ++g) {
            const unsigned v = group(t[g], 4 * g);
            by[nb++] = (unsigned char)(v >> 16); by[nb++] = (unsigned char)(v >> 8); by[nb++] = (unsigned char)v;
        }
        if (rem) {
            const unsigned w = t[full_groups];
            const unsigned a = lut[w & 255u], b = lut[(w >> 8) & 255u], c = lut[(w >> 16) & 255u], d = lut[w >> 24];
            // the host decoder's rule: the padding is what the trailing '=' say, and it must leave exactly 4 dim bytes
            const int wrong = a >= 64 ? 0 : (b >= 64 ? 1 : ((rem == 2 ? c >= 64 : c != 64) ? 2 : (d != 64 ? 3 : -1)));
            if (wrong >= 0 && bad_at < 0) pad_bad = 4 * full_groups + wrong;
            const unsigned v = (a << 18) | (b << 12) | ((rem == 2 ? c : 0u) << 6);
            by[nb++] = (unsigned char)(v >> 16);
            if (rem == 2) by[nb++] = (unsigned char)(v >> 8);
        }
        for (int k = 0; k + 3 < nb; k += 4)
            o[3 * chunks + (k >> 2)] = (unsigned)by[k] | ((unsigned)by[k + 1] << 8) | ((unsigned)by[k + 2] << 16) | ((unsigned)by[k + 3] << 24);
    }
    if (bad_at >= 0) {
        // the exact character inside the group
        const unsigned w = t[bad_at >> 2];
        int ch = bad_at;
        for (int k = 0; k < 4; ++k)
            if (lut[(w >> (8 * k)) & 255u] & 0xc0u) { ch = bad_at + k; break; }
        atomicMax(err, ~(((unsigned long long)slot << 32) | (unsigned)ch));
    } else if (pad_bad >= 0) {
        atomicMax(err, ~(((unsigned long long)slot << 32) | (unsigned)pad_bad));
    }
}

}  // namespace cpt

extern "C" {

size_t cpt_b64_chars(int dim) {
    if (dim <= 0) return 0;
    return 4 * (((size_t)dim * 4 + 2) / 3);
}

int cpt_b64_decode_regions_device(const void* text_dev, const int64_t* mask_img_dev, int n_seq, int dim, int max_regions, float* out_dev,
                                  unsigned long long* err_dev, void* stream) {
    if (!text_dev || !mask_img_dev || !out_dev || !err_dev) return cpt::abi_fail(CPT_ERR_NULL, "cpt_b64_decode_regions_device: null argument");
    if (n_seq < 0 || dim <= 0 || max_regions <= 0) return cpt::abi_fail(CPT_ERR_SHAPE, "cpt_b64_decode_regions_device: bad sizes");
    if (((uintptr_t)text_dev | (uintptr_t)out_dev) & 3 || ((uintptr_t)err_dev & 7) || ((uintptr_t)mask_img_dev & 7))
        return cpt::abi_fail(CPT_ERR_ALIGN, "cpt_b64_decode_regions_device: text / out must be 4-byte, mask / err 8-byte aligned");
    if (n_seq == 0) return CPT_OK;
    const size_t slots = (size_t)n_seq * max_regions;
    if (slots > (size_t)0x7fffffff) return cpt::abi_fail(CPT_ERR_SHAPE, "cpt_b64_decode_regions_device: %zu region slots", slots);
    cpt::b64_regions_kernel<<<dim3((unsigned)slots), dim3(256), 0, (hipStream_t)stream>>>((const unsigned*)text_dev, mask_img_dev, (unsigned*)out_dev, dim,
                                                                                          (int)(cpt_b64_chars(dim) / 4), err_dev);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cpt::abi_fail(CPT_ERR_HIP - (int)e, "cpt_b64_decode_regions_device: %s", hipGetErrorString(e));
    return CPT_OK;
}

}  // extern "C"
