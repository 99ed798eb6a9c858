// AVX2 inner loop of the base64 decoder of the region-feature wire format (io_decode.hip; include/cpt_io.h).
// The reference decodes every region's float32[2054] with base64.b64decode, one Python call per box
// (/root/reference/Oscar/oscar/datasets/refcoco_zsl_cpt_dataset.py:161-180).  At the GPU's forward rate the host has to turn 35 MB of
// base64 text per 64-sequence step into floats, and the table-driven scalar loop (3-4 GB/s per core) left the decode workers as the limit of
// the measured input pipeline (bench.py io_pipeline_measured: 0.88 of the forward-only rate with 16 threads).  This is the vectorised
// lookup + pack of Mula & Lemire, "Faster Base64 Encoding and Decoding using AVX2 Instructions" (ACM TWEB 2018): 32 characters -> 24 bytes
// per iteration; validity comes out of the same two nibble lookups.  Compiled as its own translation unit with -mavx2 and only ever
// called behind a run-time CPU check (io_decode.hip); an invalid block is left to the scalar loop, which reports the offending character.
#include <immintrin.h>
#include <stddef.h>

namespace cpt {

// run-time CPU check (a host builtin: this file is compiled for the host only)
int b64_cpu_has_avx2() {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") ? 1 : 0;
}

// Decodes up to `blocks` consecutive 32-character groups of s into 24 bytes each; returns the number of groups decoded (stops in front of the
// first group that holds a character outside the alphabet -- '=' padding included).  Writes exactly 24 bytes per decoded group.
size_t b64_avx2_blocks(const unsigned char* s, size_t blocks, unsigned char* out) {
    const __m256i lut_lo = _mm256_setr_epi8(0x15, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x13, 0x1A, 0x1B, 0x1B, 0x1B, 0x1A,
                                            0x15, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x11, 0x13, 0x1A, 0x1B, 0x1B, 0x1B, 0x1A);
    const __m256i lut_hi = _mm256_setr_epi8(0x10, 0x10, 0x01, 0x02, 0x04, 0x08, 0x04, 0x08, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10,
                                            0x10, 0x10, 0x01, 0x02, 0x04, 0x08, 0x04, 0x08, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10, 0x10);
    const __m256i lut_roll = _mm256_setr_epi8(0, 16, 19, 4, -65, -65, -71, -71, 0, 0, 0, 0, 0, 0, 0, 0,
                                              0, 16, 19, 4, -65, -65, -71, -71, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i mask_2f = _mm256_set1_epi8(0x2f);
    const __m256i pack_bytes = _mm256_setr_epi8(2, 1, 0, 6, 5, 4, 10, 9, 8, 14, 13, 12, -1, -1, -1, -1,
                                                2, 1, 0, 6, 5, 4, 10, 9, 8, 14, 13, 12, -1, -1, -1, -1);
    const __m256i pack_lanes = _mm256_setr_epi32(0, 1, 2, 4, 5, 6, -1, -1);
    size_t b = 0;
    for (; b < blocks; ++b, s += 32, out += 24) {
        __m256i str = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(s));
        // character class by high nibble x low nibble: (lo & hi) != 0 for every byte outside A-Z a-z 0-9 + /
        const __m256i hi_nib = _mm256_and_si256(_mm256_srli_epi32(str, 4), mask_2f);
        const __m256i lo_nib = _mm256_and_si256(str, mask_2f);
        const __m256i hi = _mm256_shuffle_epi8(lut_hi, hi_nib);
        const __m256i lo = _mm256_shuffle_epi8(lut_lo, lo_nib);
        if (!_mm256_testz_si256(lo, hi)) break;
        // character -> 6-bit value: add an offset chosen by the high nibble ('/' = 0x2f shares a nibble with '+': one step back in the table)
        const __m256i eq_2f = _mm256_cmpeq_epi8(str, mask_2f);
        const __m256i roll = _mm256_shuffle_epi8(lut_roll, _mm256_add_epi8(eq_2f, hi_nib));
        str = _mm256_add_epi8(str, roll);
        // four 6-bit values -> three bytes: (a << 6 | b), (c << 6 | d) as 16-bit, then both as one 24-bit group per 32-bit lane; bytes into order
        const __m256i ab_cd = _mm256_maddubs_epi16(str, _mm256_set1_epi32(0x01400140));
        __m256i v = _mm256_madd_epi16(ab_cd, _mm256_set1_epi32(0x00011000));
        v = _mm256_shuffle_epi8(v, pack_bytes);
        v = _mm256_permutevar8x32_epi32(v, pack_lanes);
        _mm_storeu_si128(reinterpret_cast<__m128i*>(out), _mm256_castsi256_si128(v));
        _mm_storel_epi64(reinterpret_cast<__m128i*>(out + 16), _mm256_extracti128_si256(v, 1));
    }
    return b;
}

}  // namespace cpt
