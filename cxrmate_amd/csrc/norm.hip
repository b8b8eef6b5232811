// LayerNorm forward / backward over the channel dimension of token-major bf16 activations [rows, C]
// (CvT conv-embedding LN + layernorm_before/after, projection-head LN, every BERT LayerNorm).
// HBM-bound: 16-byte vector loads, a row is owned by a group of LPR lanes of one wave (wave-shuffle reductions),
// several rows per wave when C is small (C=64 -> 8 rows per wave) so that every lane issues a full 16-byte access.
#include "common.h"

template <int C> struct LNCfg {
    static constexpr int CH = C / 8;                                    // 16-byte chunks per row
    static constexpr int LPR = CH >= 64 ? 64 : (CH > 32 ? 64 : (CH > 16 ? 32 : (CH > 8 ? 16 : (CH > 4 ? 8 : 4))));
    static constexpr int CPL = (CH + LPR - 1) / LPR;                    // chunks per lane
    static constexpr int RPW = 64 / LPR;                                // rows per wave
};

template <int C>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y, long ldy,
                                                            float* __restrict__ stats, long rows, float eps) {
    using L = LNCfg<C>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % L::LPR, grp = lane / L::LPR;
    const long rows_per_block = 4 * L::RPW;
    for (long base = (long)blockIdx.x * rows_per_block; base < rows; base += (long)gridDim.x * rows_per_block) {
        const long row = base + wave * L::RPW + grp;
        const bool live = row < rows;
        float v[L::CPL][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < L::CPL; ++i) {
            const int ch = sub + i * L::LPR;
            if (live && ch < L::CH) {
                const uint4 u = *reinterpret_cast<const uint4*>(x + row * ldx + ch * 8);
                unpack8(u, v[i]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[i][j];
        }
        const float mean = group_sum<L::LPR>(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < L::CPL; ++i) {
            const int ch = sub + i * L::LPR;
            if (ch < L::CH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        }
        const float rstd = rsqrtf(group_sum<L::LPR>(q) * (1.0f / C) + eps);
        if (!live) continue;
#pragma unroll
        for (int i = 0; i < L::CPL; ++i) {
            const int ch = sub + i * L::LPR;
            if (ch < L::CH) {
                const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(beta + ch * 8), b1 = *reinterpret_cast<const float4*>(beta + ch * 8 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
                *reinterpret_cast<uint4*>(y + row * ldy + ch * 8) = pack8(o);
            }
        }
        if (stats && sub == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ;  dgamma += sum dy*xhat ; dbeta += sum dy
template <int C>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                            const float* __restrict__ gamma, const float* __restrict__ stats,
                                                            const bf16_t* __restrict__ add, long ldadd,
                                                            bf16_t* __restrict__ dx, long lddx, float* __restrict__ partial /*[grid][2][C]*/,
                                                            long rows) {
    using L = LNCfg<C>;
    __shared__ float red[2][C];
    for (int i = threadIdx.x; i < 2 * C; i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % L::LPR, grp = lane / L::LPR;
    const long rows_per_block = 4 * L::RPW;
    float ag[L::CPL][8], ab[L::CPL][8];
#pragma unroll
    for (int i = 0; i < L::CPL; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; }
    for (long base = (long)blockIdx.x * rows_per_block; base < rows; base += (long)gridDim.x * rows_per_block) {
        const long row = base + wave * L::RPW + grp;
        const bool live = row < rows;
        float mean = 0.f, rstd = 0.f;
        if (live) { mean = stats[2 * row]; rstd = stats[2 * row + 1]; }
        float xh[L::CPL][8], g[L::CPL][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < L::CPL; ++i) {
            const int ch = sub + i * L::LPR;
            if (live && ch < L::CH) {
                float xv[8], dv[8];
                unpack8(*reinterpret_cast<const uint4*>(x + row * ldx + ch * 8), xv);
                unpack8(*reinterpret_cast<const uint4*>(dy + row * lddy + ch * 8), dv);
                const float4 g0 = *reinterpret_cast<const float4*>(gamma + ch * 8), g1 = *reinterpret_cast<const float4*>(gamma + ch * 8 + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[i][j] = (xv[j] - mean) * rstd;
                    g[i][j] = dv[j] * gg[j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                    ag[i][j] += dv[j] * xh[i][j];
                    ab[i][j] += dv[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { xh[i][j] = 0.f; g[i][j] = 0.f; }
            }
        }
        s1 = group_sum<L::LPR>(s1) * (1.0f / C);
        s2 = group_sum<L::LPR>(s2) * (1.0f / C);
        if (!live) continue;
#pragma unroll
        for (int i = 0; i < L::CPL; ++i) {
            const int ch = sub + i * L::LPR;
            if (ch < L::CH) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = rstd * (g[i][j] - s1 - xh[i][j] * s2);
                if (add) {
                    float av[8];
                    unpack8(*reinterpret_cast<const uint4*>(add + row * ldadd + ch * 8), av);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] += av[j];
                }
                *reinterpret_cast<uint4*>(dx + row * lddx + ch * 8) = pack8(o);
            }
        }
    }
    if (partial) {
#pragma unroll
        for (int i = 0; i < L::CPL; ++i) {
            const int ch = sub + i * L::LPR;
            if (ch < L::CH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { atomicAdd(&red[0][ch * 8 + j], ag[i][j]); atomicAdd(&red[1][ch * 8 + j], ab[i][j]); }
            }
        }
        __syncthreads();
        float* out = partial + (long)blockIdx.x * 2 * C;                  // plain stores: no cross-workgroup contention
        for (int i = threadIdx.x; i < 2 * C; i += 256) out[i] = (&red[0][0])[i];
    }
}

// dgamma[c] += sum_b partial[b][0][c]; dbeta[c] += sum_b partial[b][1][c].  Block = 32 columns x 8 partial-row lanes.
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;                      // index into the 2*C concatenated [gamma | beta] sums
    float s = 0.f;
    if (i < 2 * C)
        for (int b = rl; b < nblocks; b += 8) s += partial[(long)b * 2 * C + i];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][cl];
        if (i < C) dgamma[i] += t; else dbeta[i - C] += t;
    }
}

#define LN_DISPATCH(C_, KERNEL, ...)                                                             \
    switch (C_) {                                                                                \
        case 64: CXR_LAUNCH((KERNEL<64>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;    \
        case 128: CXR_LAUNCH((KERNEL<128>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;  \
        case 192: CXR_LAUNCH((KERNEL<192>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;  \
        case 384: CXR_LAUNCH((KERNEL<384>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;  \
        case 768: CXR_LAUNCH((KERNEL<768>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;  \
        default: return CXR_ERR_ARG;                                                              \
    }

static inline int ln_grid(long rows, int C) {
    const int lpr = C / 8 > 32 ? 64 : (C / 8 > 16 ? 32 : (C / 8 > 8 ? 16 : (C / 8 > 4 ? 8 : 4)));
    const long rpb = 4 * (64 / lpr);
    long g = (rows + rpb - 1) / rpb;
    return (int)(g < 4096 ? g : 4096);
}

extern "C" int cxr_layernorm_fwd_bf16(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                      float* stats, long rows, int C, float eps, hipStream_t stream) {
    if (rows <= 0 || (ldx % 8) || (ldy % 8)) return CXR_ERR_ARG;
    const int grid = ln_grid(rows, C);
    LN_DISPATCH(C, layernorm_fwd_kernel, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, stats, rows, eps);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// workspace: fp32 [cxr_layernorm_bwd_grid(rows, C)][2][C] (may be null when dgamma/dbeta are not wanted)
extern "C" int cxr_layernorm_bwd_grid(long rows, int C) {
    int grid = ln_grid(rows, C);
    return grid < 512 ? grid : 512;
}

extern "C" int cxr_layernorm_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* gamma, const float* stats,
                                      const void* add, long ldadd, void* dx, long lddx, float* dgamma, float* dbeta, float* workspace,
                                      long rows, int C, hipStream_t stream) {
    if (rows <= 0 || (ldx % 8) || (lddy % 8) || (lddx % 8) || (add && (ldadd % 8)) || (dgamma && !workspace)) return CXR_ERR_ARG;
    const int grid = cxr_layernorm_bwd_grid(rows, C);
    float* partial = dgamma ? workspace : nullptr;
    LN_DISPATCH(C, layernorm_bwd_kernel, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, gamma, stats, (const bf16_t*)add, ldadd,
                (bf16_t*)dx, lddx, partial, rows);
    if (dgamma) CXR_LAUNCH(layernorm_bwd_reduce_kernel, dim3(cdiv(2 * C, 32)), dim3(256), 0, stream, partial, grid, C, dgamma, dbeta);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
