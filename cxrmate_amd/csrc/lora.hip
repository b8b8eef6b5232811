// Rank-8 LoRA branch with dropout on its input -- peft's Linear in train mode: y = base(x) + (alpha/r) * B(A(dropout(x)))
// (reference modules/transformers/longitudinal_model/modelling_longitudinal.py:163-170: r = 8, alpha = 32, lora_dropout = 0.1 on
// self-attention query / key). In eval mode the branch is merged into the weight (decoder.py); under model.train() the dropout makes
// that impossible, so the three rank-8 contractions run in these small kernels next to the base GEMM:
//   down : t[m, r]   = sum_k f(m,k) * x[m,k] * W(r,k)                  (f = dropout factor of (seed, site, b, t, k); optional LayerNorm of x)
//   up   : y[m, n]  += s * f(m,n)? * sum_r t[m,r] * W(r,n)             (forward: W = B; backward dx: W = A with the forward mask)
//   outer: G(k, r)  += s * sum_m f(m,k)? * a[m,k] * t[m,r]             (dB from dy and t; dA from dropout(x) and dt)
// Masks come from the same counter-based hash as every other dropout of the path (common.h), so nothing is stored.
#include "common.h"
#include "../../include/cxrmate_hip.h"

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

// ------------------------------------------------------------------------------------------------------------------ teacher-forced pass (many rows)
// The one-wave-per-row / one-thread-per-8-columns kernels above cost 30-39 us per launch at the 4080 rows of an SCST re-scoring pass (2-byte strided
// loads, 64-96 weight loads per row, up to 128 same-address atomics) -- 66 launches, 2.3 ms of a 12-ms pass. Below: several problems per launch, the down
// projection on the matrix cores, weights of the up / outer kernels in registers across the row loop.

__device__ __forceinline__ uint4 lora_mask8(uint4 v, uint32_t key, uint32_t pair0, uint32_t thr16) {        // zero the dropped elements of 8 consecutive columns
    uint32_t* d = reinterpret_cast<uint32_t*>(&v);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t bits = dropout_pair_bits(key, pair0 + j);
        d[j] &= ((bits & 0xffffu) >= thr16 ? 0x0000ffffu : 0u) | ((bits >> 16) >= thr16 ? 0xffff0000u : 0u);
    }
    return v;
}

struct LoraDownMP { const bf16_t* x; long ldx; const bf16_t* W; long w_rs, w_cs; float* t; LoraDrop drop; };
struct LoraDownMArgs { LoraDownMP p[2]; long M; int K; float scale; };
constexpr int LDM_PITCH = 1024 + 8;

// t[m, r] = scale * sum_k f(m,k) x[m,k] W(r,k): one workgroup = 16 rows of one problem, its 4 waves split the K / 32 steps of
// v_mfma_f32_16x16x32_bf16 (A = masked x rows, B = W as [k][r] with columns 8..15 zero); W staged once in LDS as [r][k] whatever its strides.
__global__ __launch_bounds__(256) void lora_down_mfma_kernel(const LoraDownMArgs g) {
    __shared__ __attribute__((aligned(16))) bf16_t wl[8 * LDM_PITCH];
    __shared__ float red[4][16][8];
    const LoraDownMP& P = g.p[blockIdx.y];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, fg = lane >> 4;
    for (int i = tid; i < 8 * g.K; i += 256) {
        const int r = i / g.K, k = i - r * g.K;
        wl[r * LDM_PITCH + k] = P.W[r * P.w_rs + k * P.w_cs];
    }
    const long m0 = (long)blockIdx.x * 16;
    const long mc = m0 + n < g.M ? m0 + n : g.M - 1;
    const uint32_t thr = P.drop.thr16;
    const uint32_t mq = (uint32_t)mc / (uint32_t)P.drop.rows_per_b;        // (32-bit division: rows < 2^31)
    const uint32_t key = thr ? dropout_row_key(*P.drop.seed, P.drop.site, mq, (uint32_t)(P.drop.t0 + (int)((uint32_t)mc - mq * (uint32_t)P.drop.rows_per_b))) : 0u;
    const int nks = g.K >> 5;
    uint4 xr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                          // all of this wave's row pieces in flight before the first MFMA
        int ks = wave + 4 * j; ks = ks < nks ? ks : nks - 1;
        xr[j] = *reinterpret_cast<const uint4*>(P.x + mc * P.ldx + ks * 32 + fg * 8);
    }
    __syncthreads();
    f32x4_t acc = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ks = wave + 4 * j;
        if (ks < nks) {                                                    // (wave-uniform)
            uint4 xv = xr[j];
            if (thr) xv = lora_mask8(xv, key, (uint32_t)(ks * 16 + fg * 4), thr);
            uint4 wv = make_uint4(0, 0, 0, 0);
            if (n < 8) wv = *reinterpret_cast<const uint4*>(wl + n * LDM_PITCH + ks * 32 + fg * 8);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, xv), __builtin_bit_cast(bf16x8_t, wv), acc, 0, 0, 0);
        }
    }
    if (n < 8) {                                                           // D: lane (column n, group fg) holds rows 4 fg + i
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wave][4 * fg + i][n] = acc[i];
    }
    __syncthreads();
    if (tid < 128) {
        const int row = tid >> 3, r = tid & 7;
        if (m0 + row < g.M)
            P.t[(m0 + row) * LR + r] = ((red[0][row][r] + red[1][row][r]) + (red[2][row][r] + red[3][row][r])) * g.scale * (thr ? P.drop.inv : 1.0f);
    }
}

extern "C" int cxr_lora_down_multi_bf16(const cxr_lora_down_desc* probs, int nprob, long M, int K, const unsigned int* seed, int rows_per_b, int tpos0,
                                        float scale, hipStream_t stream) {
    if (!probs || nprob < 1 || nprob > 2 || M <= 0 || M > 0x7fffffffL || K <= 0 || K > 1024 || (K % 32) || rows_per_b <= 0) return CXR_ERR_ARG;
    LoraDownMArgs g;
    g.M = M; g.K = K; g.scale = scale;
    for (int q = 0; q < 2; ++q) {
        const cxr_lora_down_desc& s = probs[q < nprob ? q : 0];
        if (!s.x || !s.W || !s.t || (s.ldx % 8) || (((size_t)s.x) % 16) || (s.p > 0.f && !seed) || s.p >= 1.f) return CXR_ERR_ARG;
        LoraDownMP& d = g.p[q];
        d.x = (const bf16_t*)s.x; d.ldx = s.ldx; d.W = (const bf16_t*)s.W; d.w_rs = s.w_rs; d.w_cs = s.w_cs; d.t = s.t;
        d.drop.seed = seed; d.drop.site = s.site; d.drop.thr16 = s.p > 0.f ? dropout_thr16(s.p) : 0u; d.drop.inv = 1.0f / (1.0f - s.p);
        d.drop.rows_per_b = rows_per_b; d.drop.t0 = tpos0;
    }
    CXR_LAUNCH(lora_down_mfma_kernel, dim3((unsigned)cdiv(M, 16), nprob), dim3(256), 0, stream, g);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// thread = (8-column chunk c, row lane rl): the chunk's 64 weights per problem stay in registers across the row loop
struct LoraUpMP { bf16_t* y; long ldy; const float* t; const bf16_t* W; long w_rs, w_cs; LoraDrop drop; };
struct LoraUpMArgs { LoraUpMP p[2]; long M; int N, rows_per_block; };

constexpr int LUP_RPT = 4;                                                  // rows per thread: their y / t loads are all issued before the first FMA

template <int NQ>
__global__ __launch_bounds__(256) void lora_up_multi_kernel(const LoraUpMArgs g) {
    const int nch = g.N >> 3, RL = 256 / nch;
    const int c = threadIdx.x % nch, rl = threadIdx.x / nch;
    if (rl >= RL) return;
    const int q0 = NQ == 2 ? 0 : blockIdx.y;
    bf16_t* y = g.p[q0].y;
    const long ldy = g.p[q0].ldy;
    const long mb = (long)blockIdx.x * (RL * LUP_RPT) + rl;
    uint4 yr[LUP_RPT];
    float4 tr[NQ][LUP_RPT][2];
#pragma unroll
    for (int i = 0; i < LUP_RPT; ++i) {
        long m = mb + (long)i * RL; m = m < g.M ? m : g.M - 1;
        yr[i] = *reinterpret_cast<const uint4*>(y + m * ldy + c * 8);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            tr[q][i][0] = *reinterpret_cast<const float4*>(g.p[q0 + q].t + m * LR);
            tr[q][i][1] = *reinterpret_cast<const float4*>(g.p[q0 + q].t + m * LR + 4);
        }
    }
    float w[NQ][LR][8];
    uint32_t seedv[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const LoraUpMP& P = g.p[q0 + q];
        seedv[q] = P.drop.thr16 ? *P.drop.seed : 0u;
        // the chunk's 64 weights as eight 16-byte loads in the two layouts that occur (one 2-byte load per weight touches 64 cache lines per wave
        // instruction: 64 such instructions per thread were most of this kernel's 30 us)
        if (P.w_cs == 1 && (P.w_rs % 8) == 0 && (((size_t)P.W) % 16) == 0) {            // A [8][N]: row r, columns 8c .. 8c+7
#pragma unroll
            for (int r = 0; r < LR; ++r) unpack8(*reinterpret_cast<const uint4*>(P.W + r * P.w_rs + c * 8), w[q][r]);
        } else if (P.w_rs == 1 && P.w_cs == LR && (((size_t)P.W) % 16) == 0) {          // B [N][8]: the chunk's 8 rows are 128 consecutive bytes
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float row[LR];
                unpack8(*reinterpret_cast<const uint4*>(P.W + (long)(c * 8 + j) * LR), row);
#pragma unroll
                for (int r = 0; r < LR; ++r) w[q][r][j] = row[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < LR; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) w[q][r][j] = bf2f(P.W[r * P.w_rs + (long)(c * 8 + j) * P.w_cs]);
        }
    }
#pragma unroll
    for (int i = 0; i < LUP_RPT; ++i) {
        const long m = mb + (long)i * RL;
        if (m >= g.M) break;
        float o[8];
        unpack8(yr[i], o);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const LoraUpMP& P = g.p[q0 + q];
            const float4 ta = tr[q][i][0], tb = tr[q][i][1];
            const float tv[LR] = {ta.x, ta.y, ta.z, ta.w, tb.x, tb.y, tb.z, tb.w};
            float a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = 0.f;
#pragma unroll
            for (int r = 0; r < LR; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] = fmaf(tv[r], w[q][r][j], a[j]);
            if (P.drop.thr16) {
                const uint32_t mq = (uint32_t)m / (uint32_t)P.drop.rows_per_b;            // (32-bit: rows < 2^31; a 64-bit division is ~150 VALU instructions)
                const uint32_t key = dropout_row_key(seedv[q], P.drop.site, mq, (uint32_t)(P.drop.t0 + (int)((uint32_t)m - mq * (uint32_t)P.drop.rows_per_b)));
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const uint32_t bits = dropout_pair_bits(key, (uint32_t)(c * 4 + (j >> 1)));
                    a[j] = (bits & 0xffffu) >= P.drop.thr16 ? a[j] * P.drop.inv : 0.f;
                    a[j + 1] = (bits >> 16) >= P.drop.thr16 ? a[j + 1] * P.drop.inv : 0.f;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] += a[j];
        }
        *reinterpret_cast<uint4*>(y + m * ldy + c * 8) = pack8(o);
    }
}

extern "C" int cxr_lora_up_add_multi_bf16(const cxr_lora_up_desc* probs, int nprob, long M, int N, const unsigned int* seed, int rows_per_b, int tpos0,
                                          hipStream_t stream) {
    if (!probs || nprob < 1 || nprob > 2 || M <= 0 || M > 0x7fffffffL || N <= 0 || (N % 8) || N > 2048 || rows_per_b <= 0) return CXR_ERR_ARG;
    LoraUpMArgs g;
    g.M = M; g.N = N;
    for (int q = 0; q < 2; ++q) {
        const cxr_lora_up_desc& s = probs[q < nprob ? q : 0];
        if (!s.y || !s.t || !s.W || (s.ldy % 8) || (((size_t)s.y) % 16) || (((size_t)s.t) % 16) || (s.p > 0.f && !seed) || s.p >= 1.f) return CXR_ERR_ARG;
        LoraUpMP& d = g.p[q];
        d.y = (bf16_t*)s.y; d.ldy = s.ldy; d.t = s.t; d.W = (const bf16_t*)s.W; d.w_rs = s.w_rs; d.w_cs = s.w_cs;
        d.drop.seed = seed; d.drop.site = s.site; d.drop.thr16 = s.p > 0.f ? dropout_thr16(s.p) : 0u; d.drop.inv = 1.0f / (1.0f - s.p);
        d.drop.rows_per_b = rows_per_b; d.drop.t0 = tpos0;
    }
    const bool same_y = nprob == 2 && probs[0].y == probs[1].y;
    if (same_y && probs[0].ldy != probs[1].ldy) return CXR_ERR_ARG;
    const int RL = 256 / (N / 8);
    const long rpb = (long)RL * LUP_RPT;
    g.rows_per_block = (int)rpb;
    if (same_y) CXR_LAUNCH(lora_up_multi_kernel<2>, dim3((unsigned)cdiv(M, rpb), 1), dim3(256), 0, stream, g);
    else CXR_LAUNCH(lora_up_multi_kernel<1>, dim3((unsigned)cdiv(M, rpb), nprob), dim3(256), 0, stream, g);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// G[k, r] += scale * sum_m f(m,k) a[m,k] t[m,r]: thread = (8-column chunk of a, row lane), 64 accumulators. One workgroup = (run of rows, block of
// <= 32 chunks = 256 columns, problem); its 256 / chunks row lanes meet in LDS, then ONE atomic per (column, r) and workgroup. The atomics are what
// such a kernel costs (device-scope fp32 adds on 24 KB of gradients: ~15 per ns): 16 row runs per problem (the kernel above: one per 32 rows; a first
// version of this one with 64 runs of all columns: 100 us for four problems)
struct LoraOuterMP { const bf16_t* a; long lda; const float* t; float* G; long g_ks, g_rs; LoraDrop drop; };
struct LoraOuterMArgs { LoraOuterMP p[4]; long M; int K, rows_per_block, nchb, ncb; float scale; };

__global__ __launch_bounds__(256) void lora_outer_multi_kernel(const LoraOuterMArgs g) {
    extern __shared__ float lom_red[];                                     // [RL - 1][64][nchb]
    const LoraOuterMP& P = g.p[blockIdx.y / g.ncb];
    const int cb = blockIdx.y % g.ncb;
    const int nch = g.K >> 3, nchb = g.nchb, RL = 256 / nchb;
    const int cl = threadIdx.x % nchb, rl = threadIdx.x / nchb, c = cb * nchb + cl;
    const bool on = rl < RL && c < nch;
    const uint32_t thr = P.drop.thr16;
    const uint32_t seedv = thr ? *P.drop.seed : 0u;
    float acc[LR][8];
#pragma unroll
    for (int r = 0; r < LR; ++r)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[r][j] = 0.f;
    const long mb = (long)blockIdx.x * g.rows_per_block, me = mb + g.rows_per_block < g.M ? mb + g.rows_per_block : g.M;
    if (on) {
        for (long m0 = mb + rl; m0 < me; m0 += 4L * RL) {                  // four rows per trip: their loads are in flight together
            uint4 raw[4];
            float4 ta[4], tb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                long m = m0 + (long)i * RL; m = m < me ? m : me - 1;
                raw[i] = *reinterpret_cast<const uint4*>(P.a + m * P.lda + c * 8);
                ta[i] = *reinterpret_cast<const float4*>(P.t + m * LR);
                tb[i] = *reinterpret_cast<const float4*>(P.t + m * LR + 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long m = m0 + (long)i * RL;
                if (m >= me) break;
                uint4 rv = raw[i];
                if (thr) {
                    const uint32_t mq = (uint32_t)m / (uint32_t)P.drop.rows_per_b;
                    const uint32_t key = dropout_row_key(seedv, P.drop.site, mq, (uint32_t)(P.drop.t0 + (int)((uint32_t)m - mq * (uint32_t)P.drop.rows_per_b)));
                    rv = lora_mask8(rv, key, (uint32_t)(c * 4), thr);
                }
                float av[8];
                unpack8(rv, av);
                const float tv[LR] = {ta[i].x, ta[i].y, ta[i].z, ta[i].w, tb[i].x, tb[i].y, tb[i].z, tb[i].w};
#pragma unroll
                for (int r = 0; r < LR; ++r)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[r][j] = fmaf(tv[r], av[j], acc[r][j]);
            }
        }
    }
    // row lanes 1 .. RL-1 -> LDS (value-major: neighbouring threads, neighbouring words), row lane 0 sums
    if (rl > 0 && rl < RL) {
        float* dst = lom_red + (long)(rl - 1) * 64 * nchb + cl;
#pragma unroll
        for (int r = 0; r < LR; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[(r * 8 + j) * nchb] = acc[r][j];
    }
    __syncthreads();
    if (on && rl == 0) {
        const float s = g.scale * (thr ? P.drop.inv : 1.0f);
        for (int o = 1; o < RL; ++o) {
            const float* src = lom_red + (long)(o - 1) * 64 * nchb + cl;
#pragma unroll
            for (int r = 0; r < LR; ++r)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[r][j] += src[(r * 8 + j) * nchb];
        }
#pragma unroll
        for (int r = 0; r < LR; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(P.G + (long)(c * 8 + j) * P.g_ks + r * P.g_rs, acc[r][j] * s);
    }
}

extern "C" int cxr_lora_outer_multi_bf16(const cxr_lora_outer_desc* probs, int nprob, long M, int K, float scale, const unsigned int* seed, int rows_per_b,
                                         int tpos0, hipStream_t stream) {
    if (!probs || nprob < 1 || nprob > 4 || M <= 0 || M > 0x7fffffffL || K <= 0 || (K % 8) || K > 2048 || rows_per_b <= 0) return CXR_ERR_ARG;
    LoraOuterMArgs g;
    g.M = M; g.K = K; g.scale = scale;
    for (int q = 0; q < 4; ++q) {
        const cxr_lora_outer_desc& s = probs[q < nprob ? q : 0];
        if (!s.a || !s.t || !s.G || (s.lda % 8) || (((size_t)s.a) % 16) || (((size_t)s.t) % 16) || (s.p > 0.f && !seed) || s.p >= 1.f) return CXR_ERR_ARG;
        LoraOuterMP& d = g.p[q];
        d.a = (const bf16_t*)s.a; d.lda = s.lda; d.t = s.t; d.G = s.G; d.g_ks = s.g_ks; d.g_rs = s.g_rs;
        d.drop.seed = seed; d.drop.site = s.site; d.drop.thr16 = s.p > 0.f ? dropout_thr16(s.p) : 0u; d.drop.inv = 1.0f / (1.0f - s.p);
        d.drop.rows_per_b = rows_per_b; d.drop.t0 = tpos0;
    }
    const int nch = K / 8;
    g.nchb = nch < 32 ? nch : 32;                                            // chunks per column block (8 row lanes at 32)
    g.ncb = cdiv(nch, g.nchb);
    const int RL = 256 / g.nchb;
    long rpb = cdiv(M, 16); rpb = rpb < 4L * RL ? 4L * RL : rpb;
    g.rows_per_block = (int)rpb;
    const size_t lds = (size_t)(RL - 1) * g.nchb * 64 * sizeof(float);        // <= 63.5 KB ((RL - 1) * nchb < 256)
    CXR_LAUNCH(lora_outer_multi_kernel, dim3((unsigned)cdiv(M, rpb), nprob * g.ncb), dim3(256), lds, stream, g);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
