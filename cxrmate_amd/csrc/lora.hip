// Rank-8 LoRA branch with dropout on its input -- peft's Linear in train mode: y = base(x) + (alpha/r) * B(A(dropout(x)))
// (reference modules/transformers/longitudinal_model/modelling_longitudinal.py:163-170: r = 8, alpha = 32, lora_dropout = 0.1 on
// self-attention query / key). In eval mode the branch is merged into the weight (decoder.py); under model.train() the dropout makes
// that impossible, so the three rank-8 contractions run in these small kernels next to the base GEMM:
//   down : t[m, r]   = sum_k f(m,k) * x[m,k] * W(r,k)                  (f = dropout factor of (seed, site, b, t, k); optional LayerNorm of x)
//   up   : y[m, n]  += s * f(m,n)? * sum_r t[m,r] * W(r,n)             (forward: W = B; backward dx: W = A with the forward mask)
//   outer: G(k, r)  += s * sum_m f(m,k)? * a[m,k] * t[m,r]             (dB from dy and t; dA from dropout(x) and dt)
// Masks come from the same counter-based hash as every other dropout of the path (common.h), so nothing is stored.
#include "common.h"

constexpr int LR = 8;      // rank

struct LoraDrop { const uint32_t* seed; uint32_t site, thr16; float inv; int rows_per_b, t0; };

__device__ __forceinline__ float lora_factor(const LoraDrop& d, uint32_t seedv, long m, int k) {
    if (!d.thr16) return 1.0f;
    const uint32_t key = dropout_row_key(seedv, d.site, (uint32_t)(m / d.rows_per_b), (uint32_t)(d.t0 + (int)(m % d.rows_per_b)));
    return dropout_keep(key, (uint32_t)k, d.thr16) ? d.inv : 0.f;
}

// one wave per row; up to two problems per launch (blockIdx.y): query and key share x but not W / site / output
struct LoraDownProb { const bf16_t* W; long w_rs, w_cs; float* t; LoraDrop drop; };
__global__ __launch_bounds__(256) void lora_down_kernel(const bf16_t* __restrict__ x, long ldx, long M, int K, LoraDownProb p0, LoraDownProb p1,
                                                        const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps, float scale) {
    const LoraDownProb P = blockIdx.y == 0 ? p0 : p1;
    const int lane = threadIdx.x & 63;
    const long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const uint32_t seedv = P.drop.thr16 ? *P.drop.seed : 0u;
    // K <= 64*16 elements per row: each lane keeps its strided elements in registers (K = 768 -> 12)
    float xv[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + 64 * i;
        xv[i] = k < K ? bf2f(x[m * ldx + k]) : 0.f;
        s += xv[i];
    }
    if (ln_g) {                                                      // x = LayerNorm(raw row), two-pass statistics in registers
        const float mean = group_sum<64>(s) / K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) if (lane + 64 * i < K) q += (xv[i] - mean) * (xv[i] - mean);
        const float rstd = rsqrtf(group_sum<64>(q) / K + ln_eps);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = lane + 64 * i;
            if (k < K) xv[i] = bf2f(f2bf((xv[i] - mean) * rstd * ln_g[k] + ln_b[k]));      // what the separate LayerNorm kernel would hand to the GEMM
        }
    }
    float acc[LR];
#pragma unroll
    for (int r = 0; r < LR; ++r) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = lane + 64 * i;
        if (k < K) {
            const float v = xv[i] * lora_factor(P.drop, seedv, m, k);
#pragma unroll
            for (int r = 0; r < LR; ++r) acc[r] += v * bf2f(P.W[r * P.w_rs + k * P.w_cs]);
        }
    }
#pragma unroll
    for (int r = 0; r < LR; ++r) acc[r] = group_sum<64>(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < LR; ++r) P.t[m * LR + r] = acc[r] * scale;
    }
}

// Few rows (cached decode: M = sequences, one token each): a whole 256-thread workgroup per (row, problem) -- the one-wave-per-row kernel
// above is a 17-us latency chain there (12 strided elements and 96 weight loads per lane, in sequence).
__global__ __launch_bounds__(256) void lora_down_row_kernel(const bf16_t* __restrict__ x, long ldx, long M, int K, LoraDownProb p0, LoraDownProb p1,
                                                            const float* __restrict__ ln_g, const float* __restrict__ ln_b, float ln_eps, float scale) {
    __shared__ float sh[4][LR];
    const LoraDownProb P = blockIdx.y == 0 ? p0 : p1;
    const long m = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t seedv = P.drop.thr16 ? *P.drop.seed : 0u;
    float xv[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        xv[i] = k < K ? bf2f(x[m * ldx + k]) : 0.f;
        s += xv[i];
    }
    if (ln_g) {
        s = group_sum<64>(s);
        if (lane == 0) sh[wave][0] = s;
        __syncthreads();
        const float mean = ((sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0])) / K;
        __syncthreads();
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) if (tid + 256 * i < K) q += (xv[i] - mean) * (xv[i] - mean);
        q = group_sum<64>(q);
        if (lane == 0) sh[wave][0] = q;
        __syncthreads();
        const float rstd = rsqrtf(((sh[0][0] + sh[1][0]) + (sh[2][0] + sh[3][0])) / K + ln_eps);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = tid + 256 * i;
            if (k < K) xv[i] = bf2f(f2bf((xv[i] - mean) * rstd * ln_g[k] + ln_b[k]));
        }
    }
    float acc[LR];
#pragma unroll
    for (int r = 0; r < LR; ++r) acc[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = tid + 256 * i;
        if (k < K) {
            const float v = xv[i] * lora_factor(P.drop, seedv, m, k);
#pragma unroll
            for (int r = 0; r < LR; ++r) acc[r] += v * bf2f(P.W[r * P.w_rs + k * P.w_cs]);
        }
    }
#pragma unroll
    for (int r = 0; r < LR; ++r) acc[r] = group_sum<64>(acc[r]);
    if (lane == 0) {
#pragma unroll
        for (int r = 0; r < LR; ++r) sh[wave][r] = acc[r];
    }
    __syncthreads();
    if (tid < LR) P.t[m * LR + tid] = ((sh[0][tid] + sh[1][tid]) + (sh[2][tid] + sh[3][tid])) * scale;
}

extern "C" int cxr_lora_down_bf16(const void* x, long ldx, long M, int K, const void* W0, long w0_rs, long w0_cs, float* t0, float p0,
                                  unsigned int site0, const void* W1, long w1_rs, long w1_cs, float* t1, float p1, unsigned int site1,
                                  const unsigned int* seed, int rows_per_b, int tpos0, const float* ln_gamma, const float* ln_beta, float ln_eps,
                                  float scale, hipStream_t stream) {
    if (M <= 0 || K <= 0 || K > 1024 || rows_per_b <= 0 || !W0 || !t0 || ((p0 > 0.f || p1 > 0.f) && !seed)) return CXR_ERR_ARG;
    auto mk = [&](const void* W, long rs, long cs, float* t, float p, unsigned int site) {
        LoraDownProb q; q.W = (const bf16_t*)W; q.w_rs = rs; q.w_cs = cs; q.t = t;
        q.drop.seed = seed; q.drop.site = site; q.drop.thr16 = p > 0.f ? dropout_thr16(p) : 0u; q.drop.inv = 1.0f / (1.0f - p);
        q.drop.rows_per_b = rows_per_b; q.drop.t0 = tpos0; return q;
    };
    if (M <= 256)
        CXR_LAUNCH(lora_down_row_kernel, dim3((unsigned)M, W1 ? 2 : 1), dim3(256), 0, stream, (const bf16_t*)x, ldx, M, K, mk(W0, w0_rs, w0_cs, t0, p0, site0),
                           mk(W1 ? W1 : W0, w1_rs, w1_cs, t1 ? t1 : t0, p1, site1), ln_gamma, ln_beta, ln_eps, scale);
    else
        CXR_LAUNCH(lora_down_kernel, dim3(cdiv(M, 4), W1 ? 2 : 1), dim3(256), 0, stream, (const bf16_t*)x, ldx, M, K, mk(W0, w0_rs, w0_cs, t0, p0, site0),
                           mk(W1 ? W1 : W0, w1_rs, w1_cs, t1 ? t1 : t0, p1, site1), ln_gamma, ln_beta, ln_eps, scale);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// y[m, n] += f(m,n) * sum_r t[m,r] * W(r,n)   (bf16 in/out, 8 columns per thread)
__global__ __launch_bounds__(256) void lora_up_add_kernel(bf16_t* __restrict__ y, long ldy, long M, int N, const float* __restrict__ t,
                                                          const bf16_t* __restrict__ W, long w_rs, long w_cs, LoraDrop drop) {
    const int nch = N / 8;
    const long total = M * nch;
    const uint32_t seedv = drop.thr16 ? *drop.seed : 0u;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long m = idx / nch;
        const int n8 = (int)(idx % nch) * 8;
        float tv[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) tv[r] = t[m * LR + r];
        float o[8];
        unpack8(*reinterpret_cast<const uint4*>(y + m * ldy + n8), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < LR; ++r) a += tv[r] * bf2f(W[r * w_rs + (long)(n8 + j) * w_cs]);
            o[j] += a * lora_factor(drop, seedv, m, n8 + j);
        }
        *reinterpret_cast<uint4*>(y + m * ldy + n8) = pack8(o);
    }
}

extern "C" int cxr_lora_up_add_bf16(void* y, long ldy, long M, int N, const float* t, const void* W, long w_rs, long w_cs, float p,
                                    const unsigned int* seed, unsigned int site, int rows_per_b, int tpos0, hipStream_t stream) {
    if (M <= 0 || N <= 0 || (N % 8) || (ldy % 8) || rows_per_b <= 0 || (p > 0.f && !seed)) return CXR_ERR_ARG;
    LoraDrop d; d.seed = seed; d.site = site; d.thr16 = p > 0.f ? dropout_thr16(p) : 0u; d.inv = 1.0f / (1.0f - p); d.rows_per_b = rows_per_b; d.t0 = tpos0;
    const long total = M * (N / 8);
    CXR_LAUNCH(lora_up_add_kernel, dim3((unsigned)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096)), dim3(256), 0, stream, (bf16_t*)y, ldy, M, N, t,
                       (const bf16_t*)W, w_rs, w_cs, d);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// G[k*g_ks + r*g_rs] += scale * sum_m f(m,k) * a[m,k] * t[m,r]     (fp32 atomics; a row chunk per workgroup, one column per thread)
__global__ __launch_bounds__(256) void lora_outer_kernel(const bf16_t* __restrict__ a, long lda, long M, int K, const float* __restrict__ t,
                                                         float* __restrict__ G, long g_ks, long g_rs, float scale, LoraDrop drop, int rows_per_block) {
    const uint32_t seedv = drop.thr16 ? *drop.seed : 0u;
    const long m0 = (long)blockIdx.x * rows_per_block, m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
    for (int k = threadIdx.x + blockIdx.y * 256; k < K; k += 256 * gridDim.y) {
        float acc[LR];
#pragma unroll
        for (int r = 0; r < LR; ++r) acc[r] = 0.f;
        for (long m = m0; m < m1; ++m) {
            const float v = bf2f(a[m * lda + k]) * lora_factor(drop, seedv, m, k);
#pragma unroll
            for (int r = 0; r < LR; ++r) acc[r] += v * t[m * LR + r];        // t row: wave-uniform address -> scalar/broadcast loads
        }
#pragma unroll
        for (int r = 0; r < LR; ++r) atomicAdd(G + k * g_ks + r * g_rs, acc[r] * scale);
    }
}

extern "C" int cxr_lora_outer_bf16(const void* a, long lda, long M, int K, const float* t, float* G, long g_ks, long g_rs, float scale, float p,
                                   const unsigned int* seed, unsigned int site, int rows_per_b, int tpos0, hipStream_t stream) {
    if (M <= 0 || K <= 0 || rows_per_b <= 0 || (p > 0.f && !seed)) return CXR_ERR_ARG;
    LoraDrop d; d.seed = seed; d.site = site; d.thr16 = p > 0.f ? dropout_thr16(p) : 0u; d.inv = 1.0f / (1.0f - p); d.rows_per_b = rows_per_b; d.t0 = tpos0;
    const int rpb = (int)(cdiv(M, 256) < 32 ? 32 : cdiv(M, 256));
    CXR_LAUNCH(lora_outer_kernel, dim3(cdiv(M, rpb), cdiv(K, 256)), dim3(256), 0, stream, (const bf16_t*)a, lda, M, K, t, G, g_ks, g_rs, scale, d, rpb);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
