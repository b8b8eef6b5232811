// Standalone micro-laboratory for the cached-decode kernels (no torch): the PRODUCT kernels are compiled in from csrc/ with CXR_STAMP defined,
// so every wave records s_memrealtime (100 MHz) at the marked points; chains of dependent launches are replayed from a hipGraph exactly as the
// decode loop replays them.   Build: scripts/lab/build.sh    Run on the GPU box: scripts/lab/decode_lab [filter]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <string>
#include <algorithm>
#include <functional>

__device__ unsigned long long* g_stamp_buf = nullptr;
#define CXR_STAMP(i)                                                                                                         \
    do {                                                                                                                     \
        if ((threadIdx.x & 63) == 0 && g_stamp_buf && blockIdx.x < 2048)                                                     \
            g_stamp_buf[((long)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();          \
        asm volatile("" ::: "memory");                                                                                       \
    } while (0)

extern "C" { int g_cxr_last_hip_error = 0; }
#include "../../cxrmate_amd/csrc/decode.hip"
#include "../../cxrmate_amd/csrc/decode_gemm.hip"
#include "../../cxrmate_amd/csrc/loss.hip"

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
#define RC(x) do { int r_ = (x); if (r_ != 0) { printf("cxr error %d at %s:%d\n", r_, __FILE__, __LINE__); exit(1); } } while (0)

__global__ void clock_probe_kernel(unsigned long long* out, int spins) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < spins; ++i) x = x * 1.0001f + 0.5f;
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)x; }
}
template <int N>
__global__ void sled_kernel(float* out) {
    CXR_STAMP(0);
    float x = threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("v_add_f32 %0, %0, 1.0" : "+v"(x));
    CXR_STAMP(1);
    if (x == -1.f) out[0] = x;
}
__global__ void kernarg_probe_kernel(const DecArgs g, float* out) {
    CXR_STAMP(0);
    asm volatile("" :: "s"(g.A), "s"(g.residual), "s"(g.stats), "s"(g.ldr), "s"(g.M), "s"(g.K));
    const DecProb P = g.p[blockIdx.y];
    asm volatile("" :: "s"(P.Wp), "s"(P.bc), "s"(P.ldc), "s"(P.N));
    CXR_STAMP(1);
    if (g.M == -1) out[0] = (float)P.N;
}
__global__ void empty_kernel(int* p) { if (threadIdx.x == 0 && blockIdx.x == 0 && p) p[0] += 1; }

static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint32_t rng_state = 12345;
static float frand() { rng_state = rng_state * 1664525u + 1013904223u; return ((rng_state >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

template <class T> T* dalloc(size_t n) { T* p; HC(hipMalloc(&p, n * sizeof(T))); HC(hipMemset(p, 0, n * sizeof(T))); return p; }
static uint16_t* dbf16(size_t n, float scale, std::vector<uint16_t>* keep = nullptr) {
    std::vector<uint16_t> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2b(frand() * 2.f * scale);
    uint16_t* p = dalloc<uint16_t>(n);
    HC(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice));
    if (keep) *keep = h;
    return p;
}
static float* df32(size_t n, float scale, float offs, std::vector<float>* keep = nullptr) {
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = offs + frand() * 2.f * scale;
    float* p = dalloc<float>(n);
    HC(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    if (keep) *keep = h;
    return p;
}

static unsigned long long* d_stamps;
static const char* g_filter = nullptr;

// time a chain of `reps` dependent launches replayed from a graph; print us / launch and the stamp profile of the LAST launch
static const int g_order_default[8] = {0, 1, 2, 3, 4, 5, -1, -1};
static const int g_order_gemm[8] = {0, 6, 7, 1, 2, 3, 4, 5};
static const int g_order_sample[8] = {0, 1, 2, 3, 4, 5, 6, 7};
static void bench(const char* name, int reps, const std::function<void(int, hipStream_t)>& fn, int nblocks_for_stamps = 0, int nwaves = 4, int nstamps = 6, const int* order = g_order_default) {
    if (g_filter && !strstr(name, g_filter)) return;
    hipStream_t s; HC(hipStreamCreate(&s));
    for (int i = 0; i < 3; ++i) fn(i, s);
    HC(hipStreamSynchronize(s));
    hipGraph_t graph; hipGraphExec_t exec;
    HC(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < reps; ++i) fn(i, s);
    HC(hipStreamEndCapture(s, &graph));
    HC(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    float best = 1e30f, med[9];
    for (int r = 0; r < 9; ++r) {
        HC(hipEventRecord(e0, s)); HC(hipGraphLaunch(exec, s)); HC(hipEventRecord(e1, s)); HC(hipStreamSynchronize(s));
        float ms; HC(hipEventElapsedTime(&ms, e0, e1)); med[r] = ms * 1e3f / reps; best = std::min(best, med[r]);
    }
    std::sort(med, med + 9);
    printf("%-58s %7.2f us/launch (min %6.2f)", name, med[4], best);
    if (nblocks_for_stamps > 0) {
        // stamps of the last launch. Per wave: d_i = t_i - t_{i-1} (d_0 = start relative to the launch's earliest wave); median / max over waves
        std::vector<unsigned long long> h(2048 * 16 * 8);
        HC(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, tlast = 0;
        int nb = std::min(nblocks_for_stamps, 2048);
        for (int b = 0; b < nb; ++b) for (int w = 0; w < nwaves; ++w) { unsigned long long v = h[((long)b * 16 + w) * 8]; if (v) t0 = std::min(t0, v); }
        printf("  | start");
        for (int ii = 0; ii < nstamps; ++ii) {
            const int i = order[ii], ip = ii ? order[ii - 1] : -1;
            if (i < 0) break;
            std::vector<double> d;
            for (int b = 0; b < nb; ++b) for (int w = 0; w < nwaves; ++w) {
                const unsigned long long* r = &h[((long)b * 16 + w) * 8];
                if (!r[i] || r[i] < t0) continue;
                tlast = std::max(tlast, r[i]);
                if (ii == 0) d.push_back((r[0] - t0) * 0.01);
                else if (r[ip] && r[i] >= r[ip]) d.push_back((r[i] - r[ip]) * 0.01);
            }
            if (d.empty()) { printf(" [%d] -", i); continue; }
            std::sort(d.begin(), d.end());
            printf(" %s%.2f/%.2f", ii ? "+" : "", d[d.size() / 2], d.back());
        }
        printf("  span %.2f us", (tlast - t0) * 0.01);
    }
    printf("\n");
    fflush(stdout);
    HC(hipGraphExecDestroy(exec)); HC(hipGraphDestroy(graph)); HC(hipStreamDestroy(s));
    HC(hipMemset(d_stamps, 0, 2048 * 16 * 8 * 8));
}

int main(int argc, char** argv) {
    if (argc > 1 && strcmp(argv[1], "all")) g_filter = argv[1];
    printf("decode_lab filter=%s nset=%s\n", argc > 1 ? argv[1] : "all", argc > 2 ? argv[2] : "24");
    HC(hipSetDevice(0));
    d_stamps = dalloc<unsigned long long>(2048 * 16 * 8);
    HC(hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_buf), &d_stamps, sizeof(d_stamps)));
    const int M = 32, D = 768, F = 3072, V = 30000;
    constexpr int NSETMAX = 24;
    const int NSET = (argc > 2 && atoi(argv[2]) > 0 && atoi(argv[2]) <= NSETMAX) ? atoi(argv[2]) : NSETMAX;      // distinct weight sets cycled through a chain
    std::vector<uint16_t> hA, hW0, hA3;
    uint16_t* A = dbf16((size_t)M * D, 1.0f, &hA);
    uint16_t* A3 = dbf16((size_t)M * F, 0.5f, &hA3);
    uint16_t* Rsd = dbf16((size_t)M * D, 1.0f);
    uint16_t* W768[NSETMAX]; uint16_t* Wup[NSETMAX]; uint16_t* Wdn[NSETMAX];
    for (int i = 0; i < NSET; ++i) { W768[i] = dbf16((size_t)3 * D * D, 0.04f, i == 0 ? &hW0 : nullptr); Wup[i] = dbf16((size_t)F * D, 0.04f); Wdn[i] = dbf16((size_t)D * F, 0.04f); }
    uint16_t* Wv = dbf16((size_t)V * D, 0.04f);
    std::vector<float> hg, hb, hbias;
    float* gam = df32(D, 0.2f, 1.0f, &hg); float* bet = df32(D, 0.1f, 0.0f, &hb);
    float* bias = df32(F, 0.1f, 0.0f, &hbias); float* biasv = df32(V, 0.1f, 0.0f);
    float* lnstats = dalloc<float>(M * 2);
    uint16_t* C = dalloc<uint16_t>((size_t)M * F); uint16_t* C1 = dalloc<uint16_t>((size_t)M * D); uint16_t* C2 = dalloc<uint16_t>((size_t)M * D);
    float* Cf = dalloc<float>((size_t)M * V);
    unsigned int* seed = dalloc<unsigned int>(1);
    int* cnt = dalloc<int>(4);
    std::vector<uint16_t> hlA, hlB;
    uint16_t* lrA = dbf16(8 * D, 0.05f, &hlA); uint16_t* lrB = dbf16((size_t)D * 8, 0.05f, &hlB);
    uint16_t* lrA2 = dalloc<uint16_t>(16 * D);
    RC(cxr_dec_pack_lora_bf16(lrA, gam, bet, lrA2, D, 0));
    // packed weights: LN-folded (f*) and plain (p*)
    uint16_t* Wf768[NSETMAX]; float* bcf768[NSETMAX]; uint16_t* Wp768[NSETMAX]; float* bcp768[NSETMAX];
    uint16_t* Wfup[NSETMAX]; float* bcfup[NSETMAX]; uint16_t* Wpdn[NSETMAX]; float* bcpdn[NSETMAX];
    for (int i = 0; i < NSET; ++i) {
        Wf768[i] = dalloc<uint16_t>((size_t)3 * D * D); bcf768[i] = dalloc<float>(2 * 3 * D);
        RC(cxr_dec_pack_weight_bf16(W768[i], D, gam, bet, bias, Wf768[i], bcf768[i], 3 * D, D, 0));
        Wp768[i] = dalloc<uint16_t>((size_t)3 * D * D); bcp768[i] = dalloc<float>(2 * 3 * D);
        RC(cxr_dec_pack_weight_bf16(W768[i], D, nullptr, nullptr, bias, Wp768[i], bcp768[i], 3 * D, D, 0));
        Wfup[i] = dalloc<uint16_t>((size_t)F * D); bcfup[i] = dalloc<float>(2 * F);
        RC(cxr_dec_pack_weight_bf16(Wup[i], D, gam, bet, bias, Wfup[i], bcfup[i], F, D, 0));
        Wpdn[i] = dalloc<uint16_t>((size_t)D * F); bcpdn[i] = dalloc<float>(2 * D);
        RC(cxr_dec_pack_weight_bf16(Wdn[i], F, nullptr, nullptr, bias, Wpdn[i], bcpdn[i], D, F, 0));
    }
    uint16_t* Wfv = dalloc<uint16_t>((size_t)V * D); float* bcv = dalloc<float>(2 * (size_t)V);
    RC(cxr_dec_pack_weight_bf16(Wv, D, gam, bet, biasv, Wfv, bcv, V, D, 0));
    float* stA = dalloc<float>(48 * M * 2); float* stO = dalloc<float>(48 * M * 2);
    uint16_t* Ad = dalloc<uint16_t>((size_t)M * D); uint16_t* A3d = dalloc<uint16_t>((size_t)M * F); uint16_t* Rd = dalloc<uint16_t>((size_t)M * D);
    uint16_t* Cd = dalloc<uint16_t>((size_t)M * F);
    RC(cxr_dec_to_dal_bf16(A, D, M, D, Ad, stA, 0));
    RC(cxr_dec_to_dal_bf16(A3, F, M, F, A3d, nullptr, 0));
    RC(cxr_dec_to_dal_bf16(Rsd, D, M, D, Rd, nullptr, 0));
    std::vector<float> hrgb(2 * D);
    for (int k = 0; k < D; ++k) { hrgb[2 * k] = hg[k]; hrgb[2 * k + 1] = hb[k]; }
    float* rgb = dalloc<float>(2 * D); HC(hipMemcpy(rgb, hrgb.data(), hrgb.size() * 4, hipMemcpyHostToDevice));
    HC(hipDeviceSynchronize());

    // ---------------------------------------------------------------- correctness of the folded GEMM against a host reference (set 0)
    if (!g_filter || strstr("check", g_filter)) {
        cxr_dec_gemm_desc d; memset(&d, 0, sizeof(d));
        d.A = Ad; d.M = M; d.K = D; d.nprob = 1; d.eps = 1e-12f;
        d.p[0].Wp = Wf768[0]; d.p[0].bc = bcf768[0]; d.p[0].C = Cd; d.p[0].c_dal = 1; d.p[0].N = D; d.p[0].fold = 1;
        d.stats = stA; d.stats_tiles = 48; d.out_stats = stO;
        auto reference = [&](bool with_lora, const std::vector<uint16_t>& hc, const std::vector<float>& hs, const char* what) {
            double err = 0, ref2 = 0, serr = 0;
            for (int m = 0; m < M; ++m) {
                double mean = 0, var = 0;
                for (int k = 0; k < D; ++k) mean += b2f(hA[m * D + k]);
                mean /= D;
                for (int k = 0; k < D; ++k) { double dd = b2f(hA[m * D + k]) - mean; var += dd * dd; }
                const double rstd = 1.0 / sqrt(var / D + 1e-12);
                double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                if (with_lora) for (int r = 0; r < 8; ++r) { for (int k = 0; k < D; ++k) t[r] += ((b2f(hA[m * D + k]) - mean) * rstd * hg[k] + hb[k]) * b2f(hlA[r * D + k]); t[r] *= 4.0; }
                for (int n = 0; n < D; n += 7) {
                    double acc = hbias[n];
                    for (int k = 0; k < D; ++k) acc += ((b2f(hA[m * D + k]) - mean) * rstd * hg[k] + hb[k]) * b2f(hW0[(size_t)n * D + k]);
                    for (int r = 0; r < 8; ++r) acc += t[r] * b2f(hlB[n * 8 + r]);
                    const double got = b2f(hc[m * D + n]);
                    err += (got - acc) * (got - acc); ref2 += acc * acc;
                }
                double S = 0; for (int tt = 0; tt < 48; ++tt) S += hs[(tt * M + m) * 2];
                double S2 = 0; for (int n = 0; n < D; ++n) S2 += b2f(hc[m * D + n]);
                serr = std::max(serr, fabs(S - S2));
            }
            printf("check %s: rel-rms vs fp64 LN->GEMM reference %.5f ; out_stats sum err %.5f\n", what, sqrt(err / ref2), serr);
        };
        std::vector<uint16_t> hc((size_t)M * D); std::vector<float> hs(48 * M * 2);
        RC(cxr_dec_gemm_bf16(&d, 0));
        RC(cxr_dec_from_dal_bf16(Cd, M, D, C1, D, 0));
        HC(hipMemcpy(hc.data(), C1, hc.size() * 2, hipMemcpyDeviceToHost)); HC(hipMemcpy(hs.data(), stO, hs.size() * 4, hipMemcpyDeviceToHost));
        reference(false, hc, hs, "LN-fold, packed weights, DAL in/out");
        d.p[0].lr_Ap = lrA2; d.p[0].lr_B = lrB; d.lr_scale = 4.0f;
        RC(cxr_dec_gemm_bf16(&d, 0));
        RC(cxr_dec_from_dal_bf16(Cd, M, D, C1, D, 0));
        HC(hipMemcpy(hc.data(), C1, hc.size() * 2, hipMemcpyDeviceToHost)); HC(hipMemcpy(hs.data(), stO, hs.size() * 4, hipMemcpyDeviceToHost));
        reference(true, hc, hs, "LN-fold + LoRA (no dropout)");
        // plain weights, row-major fp32 output, residual with LayerNorm: y = A.W^T + b + LN(R)
        d.p[0].lr_Ap = nullptr; d.p[0].lr_B = nullptr; d.p[0].Wp = Wp768[0]; d.p[0].bc = bcp768[0]; d.p[0].fold = 0; d.p[0].C = Cf; d.p[0].c_dal = 0; d.p[0].ldc = D;
        d.out_f32 = 1; d.residual = Ad; d.ldr = 0; d.rgb = rgb; d.out_stats = nullptr;
        RC(cxr_dec_gemm_bf16(&d, 0));
        std::vector<float> hf((size_t)M * D); HC(hipMemcpy(hf.data(), Cf, hf.size() * 4, hipMemcpyDeviceToHost));
        double err = 0, ref2 = 0;
        for (int m = 0; m < M; m += 3) {
            double mean = 0, var = 0;
            for (int k = 0; k < D; ++k) mean += b2f(hA[m * D + k]);
            mean /= D;
            for (int k = 0; k < D; ++k) { double dd = b2f(hA[m * D + k]) - mean; var += dd * dd; }
            const double rstd = 1.0 / sqrt(var / D + 1e-12);
            for (int n = 0; n < D; n += 5) {
                double acc = hbias[n] + (b2f(hA[m * D + n]) - mean) * rstd * hg[n] + hb[n];
                for (int k = 0; k < D; ++k) acc += b2f(hA[m * D + k]) * b2f(hW0[(size_t)n * D + k]);
                err += (hf[m * D + n] - acc) * (hf[m * D + n] - acc); ref2 += acc * acc;
            }
        }
        printf("check plain + LN(residual), fp32 row-major out: rel-rms %.6f\n", sqrt(err / ref2));
    }

    // ---------------------------------------------------------------- what a kernel start costs
    {
        float* dump = dalloc<float>(4);
        bench("sled 64 instructions", 48, [&](int i, hipStream_t s) { hipLaunchKernelGGL(sled_kernel<64>, dim3(48), dim3(256), 0, s, dump); }, 48, 4, 2);
        bench("sled 256 instructions", 48, [&](int i, hipStream_t s) { hipLaunchKernelGGL(sled_kernel<256>, dim3(48), dim3(256), 0, s, dump); }, 48, 4, 2);
        bench("sled 1024 instructions", 48, [&](int i, hipStream_t s) { hipLaunchKernelGGL(sled_kernel<1024>, dim3(48), dim3(256), 0, s, dump); }, 48, 4, 2);
        bench("sled 1024 instr alternating with a weight-streaming GEMM", 48, [&](int i, hipStream_t s) {
            if (i & 1) hipLaunchKernelGGL(sled_kernel<1024>, dim3(48), dim3(256), 0, s, dump);
            else RC(cxr_gemm_skinny_bf16(A, D, Wup[i % NSET], D, C, F, bias, nullptr, 0, M, F, D, 0, 0, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, 0, s)); }, 48, 4, 2);
        DecArgs ka; memset(&ka, 0, sizeof(ka)); ka.M = 32;
        bench("kernarg probe (352-byte argument struct)", 48, [&](int i, hipStream_t s) { hipLaunchKernelGGL(kernarg_probe_kernel, dim3(48, 1), dim3(256), 0, s, ka, dump); }, 48, 4, 2);
    }
    // ---------------------------------------------------------------- chains
    bench("empty kernel", 64, [&](int i, hipStream_t s) { hipLaunchKernelGGL(empty_kernel, dim3(48), dim3(256), 0, s, cnt); });
    bench("old skinny 768x768 plain", 48, [&](int i, hipStream_t s) {
        RC(cxr_gemm_skinny_bf16(A, D, W768[i % NSET], D, C1, D, bias, nullptr, 0, M, D, D, 0, 0, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, 0, s)); }, 48);
    bench("old skinny 768x768 lnA", 48, [&](int i, hipStream_t s) {
        RC(cxr_gemm_skinny_bf16(A, D, W768[i % NSET], D, C1, D, bias, nullptr, 0, M, D, D, 0, 0, gam, bet, 1e-12f, lnstats, nullptr, nullptr, nullptr, 0.f, nullptr, 0, 0, s)); }, 48);
    bench("old skinny 768x768 lnR+res+drop", 48, [&](int i, hipStream_t s) {
        RC(cxr_gemm_skinny_bf16(A, D, W768[i % NSET], D, C1, D, bias, Rsd, D, M, D, D, 0, 0, nullptr, nullptr, 0.f, nullptr, lnstats, gam, bet, 0.1f, seed, 5, 9, s)); }, 48);
    bench("old qkv lnA + LoRA in-kernel (p=0.1)", 48, [&](int i, hipStream_t s) {
        const uint16_t* w = W768[i % NSET];
        RC(cxr_gemm_skinny3_bf16(A, D, w, bias, C, D, w + D * D, bias, C1, D, w + 2 * D * D, bias, C2, D, D, M, D, D, gam, bet, 1e-12f, lnstats,
                                 nullptr, lrB, nullptr, lrB, lrA, lrA, 0.1f, seed, 21, 22, 9, 4.0f, s)); }, 144);
    bench("old ffn-up 3072x768 lnA gelu", 48, [&](int i, hipStream_t s) {
        RC(cxr_gemm_skinny_bf16(A, D, Wup[i % NSET], D, C, F, bias, nullptr, 0, M, F, D, 1, 0, gam, bet, 1e-12f, lnstats, nullptr, nullptr, nullptr, 0.f, nullptr, 0, 0, s)); }, 192);
    bench("old ffn-down 768x3072 lnR+res+drop", 48, [&](int i, hipStream_t s) {
        RC(cxr_gemm_skinny_bf16(A3, F, Wdn[i % NSET], F, C1, D, bias, Rsd, D, M, D, F, 0, 0, nullptr, nullptr, 0.f, nullptr, lnstats, gam, bet, 0.1f, seed, 5, 9, s)); }, 48, 8);
    bench("old LM head 30000x768 f32", 8, [&](int i, hipStream_t s) {
        RC(cxr_gemm_skinny_bf16(A, D, Wv, D, Cf, V, biasv, nullptr, 0, M, V, D, 0, 1, nullptr, nullptr, 0.f, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, 0, 0, s)); }, 1875);

    auto newgemm = [&](const char* name, int reps, int waves, std::function<void(int, cxr_dec_gemm_desc&)> fill, int blocks) {
        bench(name, reps, [&](int i, hipStream_t s) {
            cxr_dec_gemm_desc d; memset(&d, 0, sizeof(d));
            d.A = Ad; d.M = M; d.K = D; d.nprob = 1; d.eps = 1e-12f;
            fill(i, d);
            RC(cxr_dec_gemm_bf16(&d, s)); }, blocks, waves, 8, g_order_gemm);
    };
    auto prob = [&](const void* Wp, const float* bc, void* Cc, int N, int fold, int dal, long ldc) {
        cxr_dec_gemm_prob q; memset(&q, 0, sizeof(q)); q.Wp = Wp; q.bc = bc; q.C = Cc; q.N = N; q.fold = fold; q.c_dal = dal; q.ldc = ldc; return q; };
    newgemm("new 768x768 LN-fold, all rows per WG", 48, 8, [&](int i, cxr_dec_gemm_desc& d) { d.p[0] = prob(Wf768[i % NSET], bcf768[i % NSET], Cd, D, 1, 1, 0); d.stats = stA; d.stats_tiles = 48; d.mt_hint = 2; }, 48);
    newgemm("new 768x768 plain", 48, 8, [&](int i, cxr_dec_gemm_desc& d) { d.p[0] = prob(Wp768[i % NSET], bcp768[i % NSET], Cd, D, 0, 1, 0); }, 48);
    newgemm("new 768x768 LN-fold", 48, 8, [&](int i, cxr_dec_gemm_desc& d) { d.p[0] = prob(Wf768[i % NSET], bcf768[i % NSET], Cd, D, 1, 1, 0); d.stats = stA; d.stats_tiles = 48; }, 48);
    newgemm("new 768x768 res-LN + drop + out_stats", 48, 8, [&](int i, cxr_dec_gemm_desc& d) {
        d.p[0] = prob(Wp768[i % NSET], bcp768[i % NSET], Cd, D, 0, 1, 0); d.stats = stA; d.stats_tiles = 48;
        d.residual = Rd; d.rgb = rgb; d.out_stats = stO; d.drop_p = 0.1f; d.drop_seed = seed; d.drop_site = 5; d.drop_t = 9; }, 48);
    newgemm("new qkv LN-fold + LoRA (p=0.1)", 48, 8, [&](int i, cxr_dec_gemm_desc& d) {
        const int j = i % NSET; d.nprob = 3; d.stats = stA; d.stats_tiles = 48;
        d.p[0] = prob(Wf768[j], bcf768[j], C, D, 1, 0, D); d.p[0].lr_Ap = lrA2; d.p[0].lr_B = lrB; d.p[0].lr_site = 21;
        d.p[1] = prob(Wf768[j] + D * D, bcf768[j] + 2 * D, C1, D, 1, 0, D); d.p[1].lr_Ap = lrA2; d.p[1].lr_B = lrB; d.p[1].lr_site = 22;
        d.p[2] = prob(Wf768[j] + 2 * D * D, bcf768[j] + 4 * D, C2, D, 1, 0, D);
        d.lr_p = 0.1f; d.lr_seed = seed; d.lr_scale = 4.0f; d.lr_t = 9; }, 48);
    newgemm("new qkv LN-fold no LoRA", 48, 8, [&](int i, cxr_dec_gemm_desc& d) {
        const int j = i % NSET; d.nprob = 3; d.stats = stA; d.stats_tiles = 48;
        d.p[0] = prob(Wf768[j], bcf768[j], C, D, 1, 0, D);
        d.p[1] = prob(Wf768[j] + D * D, bcf768[j] + 2 * D, C1, D, 1, 0, D);
        d.p[2] = prob(Wf768[j] + 2 * D * D, bcf768[j] + 4 * D, C2, D, 1, 0, D); }, 48);
    newgemm("new ffn-up 3072x768 LN-fold gelu", 48, 8, [&](int i, cxr_dec_gemm_desc& d) {
        const int j = i % NSET; d.p[0] = prob(Wfup[j], bcfup[j], Cd, F, 1, 1, 0); d.stats = stA; d.stats_tiles = 48; d.act = 1; }, 192);
    for (int nc : {1, 4}) {
        char nm[96];
        snprintf(nm, 96, "new LM head 30000x768 LN-fold f32 NC=%d", nc);
        newgemm(nm, 8, 8, [&](int i, cxr_dec_gemm_desc& d) { d.p[0] = prob(Wfv, bcv, Cf, V, 1, 0, V); d.stats = stA; d.stats_tiles = 48; d.out_f32 = 1; d.nc_hint = nc; }, nc == 1 ? 1875 : 469);
    }
    bench("new ffn-down 768x3072 res-LN + drop + out_stats", 48, [&](int i, hipStream_t s) {
        cxr_dec_gemm_desc d; memset(&d, 0, sizeof(d));
        d.A = A3d; d.M = M; d.K = F; d.nprob = 1; d.eps = 1e-12f;
        d.p[0] = prob(Wpdn[i % NSET], bcpdn[i % NSET], Cd, D, 0, 1, 0); d.stats = stA; d.stats_tiles = 48;
        d.residual = Rd; d.rgb = rgb; d.out_stats = stO; d.drop_p = 0.1f; d.drop_seed = seed; d.drop_site = 5; d.drop_t = 9;
        RC(cxr_dec_gemm_bf16(&d, s)); }, 48, 16, 8, g_order_gemm);

    // ---------------------------------------------------------------- attention decode
    {
        const int B = 32, Bkv = 16, H = 12, S = 1152;
        uint16_t* q = dbf16((size_t)B * D, 1.0f);
        const int NKV = 6;                                  // one K/V pair per decoder layer (340 MB per token in the real step)
        uint16_t* K[NKV]; uint16_t* Vv[NKV];
        for (int i = 0; i < NKV; ++i) { K[i] = dbf16((size_t)Bkv * S * D, 1.0f); Vv[i] = dbf16((size_t)Bkv * S * D, 1.0f); }
        uint16_t* o = dalloc<uint16_t>((size_t)B * D); uint16_t* o2 = dalloc<uint16_t>((size_t)B * D);
        float* ws = dalloc<float>((size_t)B * H * 8 * 66);
        unsigned int* wc = dalloc<unsigned int>(B * H);
        unsigned char* kpm = dalloc<unsigned char>((size_t)Bkv * S); HC(hipMemset(kpm, 1, (size_t)Bkv * S));
        unsigned int* kbits = dalloc<unsigned int>((size_t)Bkv * 36);
        RC(cxr_pack_mask_bits(kpm, S, Bkv, S, kbits, 36, 0));
        struct { const char* name; int wg; int blocks; int waves; int bits; } geo[] = {
            {"cross-attn 256 keys/WG split 5 + merge, byte mask", 256, 960, 4, 0}, {"cross-attn 288 keys/WG split 4 + merge, bit mask", 288, 768, 4, 1},
            {"cross-attn 576 keys/WG split 2 + merge, bit mask", 576, 384, 8, 1}, {"cross-attn 576 keys/pass looping, bit mask", -576, 192, 8, 1},
            {"cross-attn 576 keys/pass looping, byte mask", -576, 192, 8, 0},
            {"cross-attn 1152 keys in ONE pass (1024 threads), bit mask", 1152, 192, 16, 1}, {"cross-attn 1152 keys in ONE pass, byte mask", 1152, 192, 16, 0}};
        std::vector<uint16_t> ref((size_t)B * D), got((size_t)B * D);
        for (auto& gm : geo) {
            const void* mk = gm.bits ? (const void*)kbits : (const void*)kpm; const long mbs = gm.bits ? 36 * 4 : S;
            bench(gm.name, 24, [&](int i, hipStream_t s) {
                RC(cxr_attn_decode_bf16(q, K[i % NKV], Vv[i % NKV], o, mk, D, (long)S * D, D, (long)S * D, D, D, mbs, B, H, S, 0.125f, 2, ws, 64, 0.1f, seed, 3, 9, gm.wg, 0, gm.bits, s)); }, gm.blocks, gm.waves, 5);
            RC(cxr_attn_decode_bf16(q, K[0], Vv[0], o2, mk, D, (long)S * D, D, (long)S * D, D, D, mbs, B, H, S, 0.125f, 2, ws, 64, 0.1f, seed, 3, 9, gm.wg, 0, gm.bits, 0));
            HC(hipDeviceSynchronize());
            HC(hipMemcpy(got.data(), o2, got.size() * 2, hipMemcpyDeviceToHost));
            if (gm.wg == 256) ref = got;
            double e = 0, r2 = 0; for (size_t i = 0; i < got.size(); ++i) { double a_ = b2f(got[i]), b_ = b2f(ref[i]); e += (a_ - b_) * (a_ - b_); r2 += b_ * b_; }
            if (!g_filter || strstr(gm.name, g_filter)) printf("    rel-rms vs the 256-key geometry: %.6f\n", sqrt(e / (r2 + 1e-30)));
        }
        // self-attention over a 128-token cache, 32 rows
        const int T = 128, Tmax = 264;
        uint16_t* Ks = dbf16((size_t)B * Tmax * D, 1.0f); uint16_t* Vs = dbf16((size_t)B * Tmax * D, 1.0f);
        unsigned char* am = dalloc<unsigned char>((size_t)B * Tmax); HC(hipMemset(am, 1, (size_t)B * Tmax));
        bench("self-attn T=128 (32 rows x 12 heads)", 48, [&](int i, hipStream_t s) {
            RC(cxr_attn_decode_bf16(q, Ks, Vs, o, am, D, (long)Tmax * D, D, (long)Tmax * D, D, D, Tmax, B, H, T, 0.125f, 1, ws, 64, 0.1f, seed, 3, 9, 256, 0, 0, s)); }, 384, 4, 5);
        bench("self-attn T=250", 48, [&](int i, hipStream_t s) {
            RC(cxr_attn_decode_bf16(q, Ks, Vs, o, am, D, (long)Tmax * D, D, (long)Tmax * D, D, D, Tmax, B, H, 250, 0.125f, 1, ws, 64, 0.1f, seed, 3, 9, 256, 0, 0, s)); }, 384, 4, 5);
        bench("self-attn T=6 (start of a decode)", 48, [&](int i, hipStream_t s) {
            RC(cxr_attn_decode_bf16(q, Ks, Vs, o, am, D, (long)Tmax * D, D, (long)Tmax * D, D, D, Tmax, B, H, 6, 0.125f, 1, ws, 64, 0.1f, seed, 3, 9, 256, 0, 0, s)); }, 384, 4, 5);
    }
    // ---------------------------------------------------------------- token selection (16 sampled rows top-k 50 + 16 greedy rows)
    {
        float* logits = df32((size_t)M * V, 3.0f, 0.f);
        float* u = df32(M, 0.49f, 0.5f);
        long* nxt = dalloc<long>(M);
        bench("select_token: 16 sampled (top-k 50) + 16 argmax rows", 32, [&](int i, hipStream_t s) {
            RC(cxr_select_token(logits, V, M, V, 1, 1.0f, 50, 1.0f, u, nxt, 1, nullptr, -1, 0, nullptr, 16, s)); }, 16, 16, 8, g_order_sample);
        bench("select_token: 32 argmax rows", 32, [&](int i, hipStream_t s) {
            RC(cxr_select_token(logits, V, M, V, 0, 1.0f, 0, 1.0f, nullptr, nxt, 1, nullptr, -1, 0, nullptr, -1, s)); });
    }
    printf("done\n");
    return 0;
}
