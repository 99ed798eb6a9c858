// Ceiling yardstick for the bench shapes ("bf16 peak ... measured by a hipBLASLt / own-kernel peak probe on the
// box ... re-measure, do not trust").  TOOLS ONLY: never linked into libcpt_hip.so, never on the product path.
//   1. hipBLASLt on the four encoder GEMMs of BASELINE configs[1] (M = 64 x 120 = 7680; N x K = 2304 x 768, 768 x 768,
//      3072 x 768, 768 x 3072), bf16 in / bf16 out / fp32 accumulate, operands in the layouts the path holds them
//      (A [M][K] row-major, W [N][K] row-major = nn.Linear), every algorithm the heuristic returns timed, best reported.
//      Random [-1, 1) operands (the guide's rule 25: zero-filled operands clock higher).  Plain GEMM: no bias / GELU / residual /
//      LayerNorm / attention, so this is a ceiling for the MATRIX part of each launch, not for the fused launch.
//   2. HBM streaming: float4 copy and read-only sum over buffers far beyond the 256 MB Infinity Cache (1 GiB each), plus the
//      same kernels on a 48 MB working set (what the step's row kernels actually see: MALL-resident).
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/yardstick.hip -lhipblaslt -o tools/yardstick.bin
//                         tools/yardstick.bin > gpurun_out/r03_yardstick.json
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <dlfcn.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define CB(x) do { hipblasStatus_t s_ = (x); if (s_ != HIPBLAS_STATUS_SUCCESS) { fprintf(stderr, "hipBLASLt error %d at %s:%d\n", (int)s_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        const float f = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;          // uniform [-1, 1)
        p[i] = (unsigned short)(__float_as_uint(f) >> 16);
    }
}
__global__ void fill_f32(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (float)(i & 1023) * 0.001f;
}
__global__ __launch_bounds__(256) void copy_f4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void read_f4(const float4* __restrict__ a, float* __restrict__ out, size_t n4) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;      // keeps the loads alive
}

template <typename F> static double time_us(F&& fn, int warm, int iters) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < warm; ++i) fn();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return (double)ms * 1e3 / iters;
}

struct Shape { const char* name; int M, N, K; };

// --loop <shape 0..3> <iters> [seconds]: find the best algorithm of that shape, then launch it `iters` times back to back (for
// rocprofv3 --pmc / --kernel-trace passes and power sampling: tools/kloop_diag.sh); with `seconds`, keep looping that long.
static int loop_mode(int si, int iters, double seconds);
// --chain <steps> [path of libcpt_hip.so]: the composite VERDICT r3 priced from stand-alone numbers ("vendor GEMM + row kernels"), measured as ONE
// back-to-back chain under the same package power cap as the fused encoder: per layer hipBLASLt QKV (+bias) -> cpt_attention -> hipBLASLt
// attention-output (+bias, + residual through beta = 1, fp32 out) -> cpt_layernorm_rows -> hipBLASLt FFN-up (+bias + GELU epilogue) ->
// hipBLASLt FFN-down (+bias + residual, fp32 out) -> cpt_layernorm_rows; 12 layers = one encoder pass at BASELINE configs[1] (64 x 120 rows).
static int chain_mode(int steps, const char* lib);

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "--loop")) return loop_mode(atoi(argv[2]), atoi(argv[3]), argc >= 5 ? atof(argv[4]) : 0.0);
    if (argc >= 3 && !strcmp(argv[1], "--chain")) return chain_mode(atoi(argv[2]), argc >= 4 ? argv[3] : "cpt_amd/libcpt_hip.so");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"note\": \"tools-only ceiling probe; random [-1,1) bf16 operands; HIP-event time over back-to-back launches\",\n",
           prop.gcnArchName, prop.multiProcessorCount);

    // ---- 1. hipBLASLt ----------------------------------------------------------------------------------------------
    hipblasLtHandle_t h;
    CB(hipblasLtCreate(&h));
    const size_t ws_bytes = 256u << 20;
    void* ws;
    CK(hipMalloc(&ws, ws_bytes));
    const Shape shapes[4] = {{"gemm_qkv (7680 x 2304 x 768)", 7680, 2304, 768}, {"gemm_attn_out (7680 x 768 x 768)", 7680, 768, 768},
                             {"gemm_ffn_up (7680 x 3072 x 768)", 7680, 3072, 768}, {"gemm_ffn_down (7680 x 768 x 3072)", 7680, 768, 3072}};
    printf(" \"hipblaslt_bf16\": {\n");
    for (int si = 0; si < 4; ++si) {
        const Shape& s = shapes[si];
        unsigned short *A, *W, *C;
        CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&W, (size_t)s.N * s.K * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
        fill_bf16<<<1024, 256>>>(A, (size_t)s.M * s.K, 1u);
        fill_bf16<<<1024, 256>>>(W, (size_t)s.N * s.K, 2u);
        // row-major out[M][N] = A[M][K] . W[N][K]^T   ==   column-major C(N x M) = op_T(W as K x N, ld K) . (A as K x M, ld K)
        hipblasLtMatmulDesc_t desc;
        CB(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
        CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
        hipblasLtMatrixLayout_t la, lb, lc;
        CB(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.K, s.N, s.K));
        CB(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.K, s.M, s.K));
        CB(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, s.N, s.M, s.N));
        hipblasLtMatmulPreference_t pref;
        CB(hipblasLtMatmulPreferenceCreate(&pref));
        CB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        const int want = 32;
        std::vector<hipblasLtMatmulHeuristicResult_t> res(want);
        int got = 0;
        CB(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, want, res.data(), &got));
        const float alpha = 1.f, beta = 0.f;
        double best = 1e30, first = -1;
        int best_i = -1;
        for (int i = 0; i < got; ++i) {
            if (res[i].state != HIPBLAS_STATUS_SUCCESS) continue;
            auto run = [&]() { CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[i].algo, ws, ws_bytes, 0)); };
            const double us = time_us(run, 3, 20);
            if (first < 0) first = us;
            if (us < best) { best = us; best_i = i; }
        }
        // best algorithm again with more iterations
        if (best_i >= 0) {
            auto run = [&]() { CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[best_i].algo, ws, ws_bytes, 0)); };
            best = std::min(best, time_us(run, 5, 100));
        }
        const double flop = 2.0 * s.M * s.N * s.K;
        printf("  \"%s\": {\"algos_tried\": %d, \"heuristic_first_us\": %.2f, \"best_us\": %.2f, \"best_TFLOPs\": %.1f, \"frac_of_2.5PF\": %.4f}%s\n",
               s.name, got, first, best, flop / best * 1e-6, flop / best * 1e-6 / 2500.0, si < 3 ? "," : "");
        CB(hipblasLtMatmulPreferenceDestroy(pref));
        CB(hipblasLtMatrixLayoutDestroy(la)); CB(hipblasLtMatrixLayoutDestroy(lb)); CB(hipblasLtMatrixLayoutDestroy(lc));
        CB(hipblasLtMatmulDescDestroy(desc));
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C));
    }
    printf(" },\n");
    CK(hipFree(ws));
    CB(hipblasLtDestroy(h));

    // ---- 2. HBM streaming ------------------------------------------------------------------------------------------
    printf(" \"hbm_stream\": {\n");
    const size_t sizes[2] = {(size_t)1 << 30, (size_t)24 << 20};       // 1 GiB per buffer (>> 256 MB MALL); 24 MB per buffer (MALL-resident)
    const char* tags[2] = {"1GiB_per_buffer", "24MiB_per_buffer_MALL_resident"};
    for (int k = 0; k < 2; ++k) {
        float *a, *b, *o;
        CK(hipMalloc(&a, sizes[k])); CK(hipMalloc(&b, sizes[k])); CK(hipMalloc(&o, 256));
        fill_f32<<<2048, 256>>>(a, sizes[k] / 4);
        const size_t n4 = sizes[k] / 16;
        double best_c = 1e30, best_r = 1e30;
        int gc = 0, gr = 0;
        const int grids[4] = {1024, 2048, 4096, 8192};
        for (int g : grids) {
            const double c = time_us([&]() { copy_f4<<<g, 256>>>((const float4*)a, (float4*)b, n4); }, 2, k ? 50 : 10);
            const double r = time_us([&]() { read_f4<<<g, 256>>>((const float4*)a, o, n4); }, 2, k ? 50 : 10);
            if (c < best_c) { best_c = c; gc = g; }
            if (r < best_r) { best_r = r; gr = g; }
        }
        printf("  \"%s\": {\"copy_us\": %.1f, \"copy_GBs_read_plus_write\": %.0f, \"copy_grid\": %d, \"read_us\": %.1f, \"read_GBs\": %.0f, \"read_grid\": %d}%s\n",
               tags[k], best_c, 2.0 * sizes[k] / best_c * 1e-3, gc, best_r, (double)sizes[k] / best_r * 1e-3, gr, k == 0 ? "," : "");
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(o));
    }
    printf(" }\n}\n");
    return 0;
}

static int loop_mode(int si, int iters, double seconds) {
    const Shape shapes[4] = {{"gemm_qkv", 7680, 2304, 768}, {"gemm_attn_out", 7680, 768, 768}, {"gemm_ffn_up", 7680, 3072, 768}, {"gemm_ffn_down", 7680, 768, 3072}};
    if (si < 0 || si > 3) return 2;
    const Shape& s = shapes[si];
    hipblasLtHandle_t h;
    CB(hipblasLtCreate(&h));
    const size_t ws_bytes = 256u << 20;
    void* ws;
    CK(hipMalloc(&ws, ws_bytes));
    unsigned short *A, *W, *C;
    CK(hipMalloc(&A, (size_t)s.M * s.K * 2)); CK(hipMalloc(&W, (size_t)s.N * s.K * 2)); CK(hipMalloc(&C, (size_t)s.M * s.N * 2));
    fill_bf16<<<1024, 256>>>(A, (size_t)s.M * s.K, 1u);
    fill_bf16<<<1024, 256>>>(W, (size_t)s.N * s.K, 2u);
    hipblasLtMatmulDesc_t desc;
    CB(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CB(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtMatrixLayout_t la, lb, lc;
    CB(hipblasLtMatrixLayoutCreate(&la, HIP_R_16BF, s.K, s.N, s.K));
    CB(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16BF, s.K, s.M, s.K));
    CB(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16BF, s.N, s.M, s.N));
    hipblasLtMatmulPreference_t pref;
    CB(hipblasLtMatmulPreferenceCreate(&pref));
    CB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(32);
    int got = 0;
    CB(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 32, res.data(), &got));
    const float alpha = 1.f, beta = 0.f;
    double best = 1e30;
    int best_i = -1;
    for (int i = 0; i < got; ++i) {
        if (res[i].state != HIPBLAS_STATUS_SUCCESS) continue;
        auto run = [&]() { CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[i].algo, ws, ws_bytes, 0)); };
        const double us = time_us(run, 3, 20);
        if (us < best) { best = us; best_i = i; }
    }
    if (best_i < 0) return 3;
    auto run = [&]() { CB(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &res[best_i].algo, ws, ws_bytes, 0)); };
    CK(hipDeviceSynchronize());
    // marker launches (a distinct kernel name) fence the loop in a kernel trace: everything between the two read_f4 markers is the winner
    float* o; CK(hipMalloc(&o, 256));
    read_f4<<<1, 256>>>((const float4*)A, o, 256);
    const double us = time_us(run, 5, iters);
    read_f4<<<1, 256>>>((const float4*)A, o, 256);
    CK(hipDeviceSynchronize());
    long spins = 0;
    if (seconds > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) { for (int i = 0; i < 200; ++i) run(); CK(hipDeviceSynchronize()); spins += 200; }
    }
    printf("{\"shape\": \"%s\", \"M\": %d, \"N\": %d, \"K\": %d, \"algos\": %d, \"best_algo_index\": %d, \"loop_us\": %.2f, \"TFLOPs\": %.1f, \"sustained_launches\": %ld}\n",
           s.name, s.M, s.N, s.K, got, best_i, us, 2.0 * s.M * s.N * s.K / us * 1e-6, spins);
    return 0;
}

// ---- --chain --------------------------------------------------------------------------------------------------------------------
namespace {
struct LtGemm {
    hipblasLtMatmulDesc_t desc; hipblasLtMatrixLayout_t la, lb, lc; hipblasLtMatmulHeuristicResult_t algo; float beta; int M, N, K; double us; int algos;
};
// row-major D[M][N] (dtype dout) = A[M][K] . W[N][K]^T + bias[N] (+ C[M][N] when resid) with an optional GELU; bf16 operands
static bool make_gemm(hipblasLtHandle_t h, LtGemm& g, int M, int N, int K, hipDataType dout, bool resid, bool gelu, void* ws, size_t ws_bytes,
                      const void* A, const void* W, const void* bias, void* D) {
    g.M = M; g.N = N; g.K = K; g.beta = resid ? 1.f : 0.f;
    CB(hipblasLtMatmulDescCreate(&g.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CB(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CB(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtEpilogue_t epi = gelu ? HIPBLASLT_EPILOGUE_GELU_BIAS : HIPBLASLT_EPILOGUE_BIAS;
    CB(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
    CB(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)));
    hipDataType bt = HIP_R_32F;
    CB(hipblasLtMatmulDescSetAttribute(g.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    CB(hipblasLtMatrixLayoutCreate(&g.la, HIP_R_16BF, K, N, K));
    CB(hipblasLtMatrixLayoutCreate(&g.lb, HIP_R_16BF, K, M, K));
    CB(hipblasLtMatrixLayoutCreate(&g.lc, dout, N, M, N));
    hipblasLtMatmulPreference_t pref;
    CB(hipblasLtMatmulPreferenceCreate(&pref));
    CB(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
    std::vector<hipblasLtMatmulHeuristicResult_t> res(32);
    int got = 0;
    if (hipblasLtMatmulAlgoGetHeuristic(h, g.desc, g.la, g.lb, g.lc, g.lc, pref, 32, res.data(), &got) != HIPBLAS_STATUS_SUCCESS || got == 0) return false;
    const float alpha = 1.f;
    double best = 1e30; int bi = -1;
    for (int i = 0; i < got; ++i) {
        if (res[i].state != HIPBLAS_STATUS_SUCCESS) continue;
        bool ok = true;
        auto run = [&]() { if (hipblasLtMatmul(h, g.desc, &alpha, W, g.la, A, g.lb, &g.beta, D, g.lc, D, g.lc, &res[i].algo, ws, ws_bytes, 0) != HIPBLAS_STATUS_SUCCESS) ok = false; };
        const double us = time_us(run, 3, 20);
        if (ok && us < best) { best = us; bi = i; }
    }
    CB(hipblasLtMatmulPreferenceDestroy(pref));
    if (bi < 0) return false;
    g.algo = res[bi]; g.us = best; g.algos = got;
    return true;
}
typedef int (*attn_fn)(int, const void*, const int64_t*, void*, void*, int, int, int, void*);
typedef int (*ln_fn)(const float*, const float*, const float*, float, float*, void*, int, int, int, int, int, int, void*);
__global__ void fill_i64(int64_t* p, size_t n, int64_t v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
__global__ void fill_const(float* p, size_t n, float v) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v; }
}  // namespace

static int chain_mode(int steps, const char* libpath) {
    void* lib = dlopen(libpath, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { fprintf(stderr, "dlopen %s: %s\n", libpath, dlerror()); return 2; }
    attn_fn attention = (attn_fn)dlsym(lib, "cpt_attention");
    ln_fn layernorm = (ln_fn)dlsym(lib, "cpt_layernorm_rows");
    if (!attention || !layernorm) { fprintf(stderr, "cpt_attention / cpt_layernorm_rows not exported\n"); return 2; }
    const int B = 64, L = 120, M = B * L, H = 768, I = 3072, heads = 12, layers = 12;
    hipblasLtHandle_t h;
    CB(hipblasLtCreate(&h));
    const size_t ws_bytes = 256u << 20;
    void* ws; CK(hipMalloc(&ws, ws_bytes));
    // one layer's weights x 12 (distinct buffers, as in the model: 14 MB per layer streams from HBM / the Infinity Cache)
    unsigned short *x_lp, *a_lp, *qkv, *ctx, *ffn;
    float *x_f32, *a_f32, *pre, *bias, *gam, *bet;
    int64_t* mask;
    CK(hipMalloc(&x_lp, (size_t)M * H * 2)); CK(hipMalloc(&a_lp, (size_t)M * H * 2)); CK(hipMalloc(&qkv, (size_t)M * 3 * H * 2));
    CK(hipMalloc(&ctx, (size_t)M * H * 2)); CK(hipMalloc(&ffn, (size_t)M * I * 2));
    CK(hipMalloc(&x_f32, (size_t)M * H * 4)); CK(hipMalloc(&a_f32, (size_t)M * H * 4)); CK(hipMalloc(&pre, (size_t)M * H * 4));
    CK(hipMalloc(&bias, (size_t)I * 4)); CK(hipMalloc(&gam, (size_t)H * 4)); CK(hipMalloc(&bet, (size_t)H * 4));
    CK(hipMalloc(&mask, (size_t)B * L * 8));
    fill_bf16<<<1024, 256>>>(x_lp, (size_t)M * H, 7u);
    fill_const<<<64, 256>>>(bias, I, 0.01f); fill_const<<<8, 256>>>(gam, H, 1.0f); fill_const<<<8, 256>>>(bet, H, 0.0f);
    fill_i64<<<64, 256>>>(mask, (size_t)B * L, 1);
    fill_f32<<<1024, 256>>>(x_f32, (size_t)M * H);
    std::vector<unsigned short*> wq(layers), wo(layers), wi(layers), wd(layers);
    for (int l = 0; l < layers; ++l) {
        CK(hipMalloc(&wq[l], (size_t)3 * H * H * 2)); CK(hipMalloc(&wo[l], (size_t)H * H * 2)); CK(hipMalloc(&wi[l], (size_t)I * H * 2)); CK(hipMalloc(&wd[l], (size_t)H * I * 2));
        // (the LayerNorm behind every dense pair keeps the chain's activations finite whatever the weight scale)
        fill_bf16<<<1024, 256>>>(wq[l], (size_t)3 * H * H, 11u + l); fill_bf16<<<1024, 256>>>(wo[l], (size_t)H * H, 31u + l);
        fill_bf16<<<1024, 256>>>(wi[l], (size_t)I * H, 51u + l); fill_bf16<<<1024, 256>>>(wd[l], (size_t)H * I, 71u + l);
    }
    CK(hipDeviceSynchronize());
    LtGemm gq, go, gi, gd;
    hipDataType sum_dt = HIP_R_32F;
    if (!make_gemm(h, gq, M, 3 * H, H, HIP_R_16BF, false, false, ws, ws_bytes, x_lp, wq[0], bias, qkv)) { fprintf(stderr, "no algorithm: qkv\n"); return 3; }
    if (!make_gemm(h, go, M, H, H, sum_dt, true, false, ws, ws_bytes, ctx, wo[0], bias, pre)) { fprintf(stderr, "no algorithm: attn-out with fp32 C/D\n"); return 3; }
    if (!make_gemm(h, gi, M, I, H, HIP_R_16BF, false, true, ws, ws_bytes, a_lp, wi[0], bias, ffn)) { fprintf(stderr, "no algorithm: ffn-up\n"); return 3; }
    if (!make_gemm(h, gd, M, H, I, sum_dt, true, false, ws, ws_bytes, ffn, wd[0], bias, pre)) { fprintf(stderr, "no algorithm: ffn-down with fp32 C/D\n"); return 3; }
    const float alpha = 1.f;
    auto mm = [&](LtGemm& g, const void* A, const void* W, const void* C, void* D) {
        CB(hipblasLtMatmul(h, g.desc, &alpha, W, g.la, A, g.lb, &g.beta, C, g.lc, D, g.lc, &g.algo.algo, ws, ws_bytes, 0));
    };
    int rc = 0;
    auto encoder = [&]() {
        for (int l = 0; l < layers; ++l) {
            mm(gq, x_lp, wq[l], qkv, qkv);
            rc |= attention(1 /* CPT_BF16 */, qkv, mask, ctx, nullptr, B, L, heads, nullptr);
            mm(go, ctx, wo[l], x_f32, pre);                                     // pre = ctx.Wo^T + b + x
            rc |= layernorm(pre, gam, bet, 1e-12f, a_f32, a_lp, 1, M, H, M, 0, 0, nullptr);
            mm(gi, a_lp, wi[l], ffn, ffn);
            mm(gd, ffn, wd[l], a_f32, pre);                                     // pre = h.Wd^T + b + a
            rc |= layernorm(pre, gam, bet, 1e-12f, x_f32, x_lp, 1, M, H, M, 0, 0, nullptr);
        }
    };
    // per-launch brackets of one warm pass (events between launches), then the chain timed as a whole
    const double us_pass = time_us(encoder, 3, steps);
    if (rc) { fprintf(stderr, "cpt_* returned %d\n", rc); return 4; }
    double us_attn = time_us([&]() { attention(1, qkv, mask, ctx, nullptr, B, L, heads, nullptr); }, 3, 50);
    double us_ln = time_us([&]() { layernorm(pre, gam, bet, 1e-12f, a_f32, a_lp, 1, M, H, M, 0, 0, nullptr); }, 3, 50);
    // sustained: >= 2 s of back-to-back passes
    const auto t0 = std::chrono::steady_clock::now();
    long passes = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 2.0) { for (int i = 0; i < 50; ++i) encoder(); CK(hipDeviceSynchronize()); passes += 50; }
    const double sus_ms = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3 / passes;
    printf("{\"what\": \"vendor GEMM + row kernels, one encoder pass (12 layers, 64 x 120 rows, bf16) as one back-to-back chain\",\n"
           " \"standalone_best_us\": {\"qkv\": %.2f, \"attn_out_bias_resid_f32out\": %.2f, \"ffn_up_bias_gelu\": %.2f, \"ffn_down_bias_resid_f32out\": %.2f, \"cpt_attention\": %.2f, \"cpt_layernorm_rows\": %.2f},\n"
           " \"standalone_sum_us_per_layer\": %.2f,\n \"chain_ms_per_encoder_pass\": %.4f, \"chain_us_per_layer\": %.2f, \"steps\": %d,\n"
           " \"sustained_2s_ms_per_encoder_pass\": %.4f, \"sustained_passes\": %ld}\n",
           gq.us, go.us, gi.us, gd.us, us_attn, us_ln, gq.us + go.us + gi.us + gd.us + us_attn + 2 * us_ln,
           us_pass * 1e-3, us_pass / layers, steps, sus_ms, passes);
    return 0;
}
