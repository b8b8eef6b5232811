// LayerNorm forward / backward over the channel dimension of token-major bf16 activations [rows, C]
// (CvT conv-embedding LN + layernorm_before/after, projection-head LN, every BERT LayerNorm).
// HBM-bound: 16-byte vector loads, a row is owned by a group of LPR lanes of one wave (wave-shuffle reductions),
// several rows per wave when C is small (C=64 -> 8 rows per wave) so that every lane issues a full 16-byte access.
#include "common.h"
#include <stdlib.h>

template <int C, bool LN_TRI = false> struct LNCfg {
    static constexpr int CH = C / 8;                                    // 16-byte chunks per row
    // CH = 3 * 2^k (C = 192 / 384 / 768), forward only: 2^k lanes x 3 chunks keeps every lane busy (a power-of-two group of >= CH lanes idles
    // 25 %); measured 6-20 % faster forward, but 1.5x SLOWER backward (three tensors x three chunks per lane leave no room for unrolling rows)
    static constexpr bool TRI = LN_TRI && (CH % 3 == 0) && ((CH / 3) & (CH / 3 - 1)) == 0 && CH / 3 >= 4;
    static constexpr int LPR = TRI ? CH / 3 : (CH >= 64 ? 64 : (CH > 32 ? 64 : (CH > 16 ? 32 : (CH > 8 ? 16 : (CH > 4 ? 8 : 4)))));
    static constexpr int CPL = (CH + LPR - 1) / LPR;                    // chunks per lane
    static constexpr int RPW = 64 / LPR;                                // rows per wave
};

// U rows per row-group per iteration: all loads of the U rows are issued before the first reduction, so every lane keeps U (forward)
// or 2-3 U (backward) 16-byte loads in flight -- at 8 waves per CU a single row per iteration leaves the kernel latency-bound (~2 TB/s).
// Q8: the output goes out as e4m3 (y8 = value * inv8, saturating) for a consumer that is an e4m3 GEMM; y (bf16) is then optional
template <int C, int U, bool Q8 = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ y, long ldy,
                                                            float* __restrict__ stats, long rows, float eps,
                                                            unsigned char* __restrict__ y8 = nullptr, long ldy8 = 0, float inv8 = 0.f) {
    CXR_PRIO_MAIN();
    using L = LNCfg<C, true>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % L::LPR, grp = lane / L::LPR;
    const long rows_per_block = 4 * L::RPW * U;
    // gamma / beta of this lane's chunks: once per kernel (they were re-loaded per row inside the loop)
    float gg[L::CPL][8], bb[L::CPL][8];
#pragma unroll
    for (int i = 0; i < L::CPL; ++i) {
        const int ch = sub + i * L::LPR, cc = ch < L::CH ? ch : 0;
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + cc * 8), g1 = *reinterpret_cast<const float4*>(gamma + cc * 8 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + cc * 8), b1 = *reinterpret_cast<const float4*>(beta + cc * 8 + 4);
        gg[i][0] = g0.x; gg[i][1] = g0.y; gg[i][2] = g0.z; gg[i][3] = g0.w; gg[i][4] = g1.x; gg[i][5] = g1.y; gg[i][6] = g1.z; gg[i][7] = g1.w;
        bb[i][0] = b0.x; bb[i][1] = b0.y; bb[i][2] = b0.z; bb[i][3] = b0.w; bb[i][4] = b1.x; bb[i][5] = b1.y; bb[i][6] = b1.z; bb[i][7] = b1.w;
    }
    for (long base = (long)blockIdx.x * rows_per_block; base < rows; base += (long)gridDim.x * rows_per_block) {
        float v[U][L::CPL][8];
        long row[U];
        // UNCONDITIONAL loads (row / chunk clamped, dead lanes zeroed by a select): behind the per-lane `row < rows && ch < CH` guard hipcc
        // branched around every load and waited vmcnt(0) after it -- U * CPL dependent round trips instead of U * CPL loads in flight
        uint4 raw[U][L::CPL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            row[u] = base + u * 4 * L::RPW + wave * L::RPW + grp;
            const long rc = row[u] < rows ? row[u] : rows - 1;
#pragma unroll
            for (int i = 0; i < L::CPL; ++i) {
                const int ch = sub + i * L::LPR, cc = ch < L::CH ? ch : 0;
                raw[u][i] = *reinterpret_cast<const uint4*>(x + rc * ldx + cc * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < L::CPL; ++i) {
                unpack8(raw[u][i], v[u][i]);
                const bool on = row[u] < rows && sub + i * L::LPR < L::CH;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[u][i][j] = on ? v[u][i][j] : 0.f;
            }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < L::CPL; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) s += v[u][i][j];
            const float mean = group_sum<L::LPR>(s) * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int i = 0; i < L::CPL; ++i) {
                if (sub + i * L::LPR < L::CH) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float d = v[u][i][j] - mean; q += d * d; }
                }
            }
            const float rstd = rsqrtf(group_sum<L::LPR>(q) * (1.0f / C) + eps);
            if (row[u] >= rows) continue;
#pragma unroll
            for (int i = 0; i < L::CPL; ++i) {
                const int ch = sub + i * L::LPR;
                if (ch < L::CH) {
                    float o[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (v[u][i][j] - mean) * rstd * gg[i][j] + bb[i][j];
                    if (!Q8 || y) *reinterpret_cast<uint4*>(y + row[u] * ldy + ch * 8) = pack8(o);
                    if (Q8) *reinterpret_cast<uint2*>(y8 + row[u] * ldy8 + ch * 8) = pack8_fp8(o, inv8);
                }
            }
            if (stats && sub == 0) { stats[2 * row[u]] = mean; stats[2 * row[u] + 1] = rstd; }
        }
    }
}

// Optional second output dx2 = f * dx: the gradient of a `dense -> dropout -> + residual` (or DropPath) branch that consumes this LayerNorm's
// input -- the forward mask re-applied (element hash, or a per-image factor row_scale[row / rows_per_b]) while dx is still in registers.
struct LnBwdDrop { bf16_t* dx2; long lddx2; const uint32_t* seed; uint32_t site, thr16; float inv; int rows_per_b, t0; const float* row_scale; };

// dx = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma ;  dgamma += sum dy*xhat ; dbeta += sum dy
// PF (round 6): the loads of the NEXT row block are issued before the current block's arithmetic and stores (two register stages, the loop unrolled
// by two so that both are statically indexed). Without it a wave alternates between "all loads in flight" and "no load in flight" (reductions,
// stores): at 18 rows per wave (36928 x 384 over 512 workgroups) the kernel is a chain of ~9 dependent HBM round trips, 2.9 TB/s alone.
template <int C, int U, bool PF = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ x, long ldx, const bf16_t* __restrict__ dy, long lddy,
                                                            const float* __restrict__ gamma, const float* __restrict__ stats,
                                                            const bf16_t* __restrict__ add, long ldadd,
                                                            bf16_t* __restrict__ dx, long lddx, float* __restrict__ partial /*[grid][2][C]*/,
                                                            long rows, LnBwdDrop dd) {
    CXR_PRIO_MAIN();
    using L = LNCfg<C>;
    __shared__ float red[4][2][C];
    const uint32_t dseed = dd.thr16 ? *dd.seed : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % L::LPR, grp = lane / L::LPR;
    const long rows_per_block = 4 * L::RPW * U;
    float ag[L::CPL][8], ab[L::CPL][8], gam[L::CPL][8];
#pragma unroll
    for (int i = 0; i < L::CPL; ++i) {
        const int ch = sub + i * L::LPR;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ag[i][j] = 0.f; ab[i][j] = 0.f; gam[i][j] = ch < L::CH ? gamma[ch * 8 + j] : 0.f; }
    }
    struct Stage { uint4 xr[U][L::CPL], dr[U][L::CPL], ar[U][L::CPL]; long row[U]; float mean[U], rstd[U], rsf[U]; };
    auto load = [&](Stage& s, const long base) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            s.row[u] = base + u * 4 * L::RPW + wave * L::RPW + grp;
            const bool live = s.row[u] < rows;
            const long rc = live ? s.row[u] : rows - 1;
            s.mean[u] = stats[2 * rc]; s.rstd[u] = stats[2 * rc + 1];
            // the DropPath factor of the row, fetched WITH the row (unconditional, stand-in address when unused): loaded where it is used, after
            // the dx store, it was a dependent round trip (load + s_waitcnt vmcnt(0)) per row in the middle of the loop
            s.rsf[u] = (dd.row_scale ? dd.row_scale : stats)[dd.row_scale ? (uint32_t)rc / (uint32_t)dd.rows_per_b : 0u];      // (32-bit division: rows < 2^31)
#pragma unroll
            for (int i = 0; i < L::CPL; ++i) {
                const int ch = sub + i * L::LPR;
                const int cc = ch < L::CH ? ch : 0;
                s.xr[u][i] = ld_stream16(x + rc * ldx + cc * 8);
                s.dr[u][i] = ld_stream16(dy + rc * lddy + cc * 8);
                if (add) s.ar[u][i] = ld_stream16(add + rc * ldadd + cc * 8);
            }
        }
    };
    auto compute = [&](Stage& s) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = s.row[u] < rows;
            float xh[L::CPL][8], g[L::CPL][8];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < L::CPL; ++i) {
                const int ch = sub + i * L::LPR;
                float xv[8], dv[8];
                unpack8(s.xr[u][i], xv);
                unpack8(s.dr[u][i], dv);
                const bool on = live && ch < L::CH;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    xh[i][j] = on ? (xv[j] - s.mean[u]) * s.rstd[u] : 0.f;
                    const float d = on ? dv[j] : 0.f;
                    g[i][j] = d * gam[i][j];
                    s1 += g[i][j];
                    s2 += g[i][j] * xh[i][j];
                    ag[i][j] += d * xh[i][j];
                    ab[i][j] += d;
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
                    for (int j = 0; j < 8; ++j) o[j] = s.rstd[u] * (g[i][j] - s1 - xh[i][j] * s2);
                    if (add) {
                        float av[8];
                        unpack8(s.ar[u][i], av);
#pragma unroll
                        for (int j = 0; j < 8; ++j) o[j] += av[j];
                    }
                    if (dx) *reinterpret_cast<uint4*>(dx + s.row[u] * lddx + ch * 8) = pack8(o);      // (dx may be null when only the scaled copy dx2 is wanted)
                    if (dd.dx2) {
                        if (dd.row_scale) {
                            const float f = s.rsf[u];
#pragma unroll
                            for (int j = 0; j < 8; ++j) o[j] *= f;
                        } else {
                            const uint32_t rq = (uint32_t)s.row[u] / (uint32_t)dd.rows_per_b;          // (32-bit: a 64-bit division is ~150 VALU instructions)
                            const uint32_t key = dropout_row_key(dseed, dd.site, rq, (uint32_t)(dd.t0 + (int)((uint32_t)s.row[u] - rq * (uint32_t)dd.rows_per_b)));
#pragma unroll
                            for (int j = 0; j < 8; j += 2) {
                                const uint32_t bits = dropout_pair_bits(key, (uint32_t)(ch * 8 + j) >> 1);
                                o[j] = (bits & 0xffffu) >= dd.thr16 ? o[j] * dd.inv : 0.f;
                                o[j + 1] = (bits >> 16) >= dd.thr16 ? o[j + 1] * dd.inv : 0.f;
                            }
                        }
                        *reinterpret_cast<uint4*>(dd.dx2 + s.row[u] * dd.lddx2 + ch * 8) = pack8(o);
                    }
                }
            }
        }
    };
    const long step = (long)gridDim.x * rows_per_block;
    if constexpr (PF) {
        Stage s0, s1;
        long base = (long)blockIdx.x * rows_per_block;
        if (base < rows) load(s0, base);
        while (base < rows) {
            const bool more1 = base + step < rows;
            if (more1) load(s1, base + step);
            compute(s0);
            if (!more1) break;
            base += step;
            const bool more0 = base + step < rows;
            if (more0) load(s0, base + step);
            compute(s1);
            if (!more0) break;
            base += step;
        }
    } else {
        for (long base = (long)blockIdx.x * rows_per_block; base < rows; base += step) {
            Stage s0;
            load(s0, base);
            compute(s0);
        }
    }
    if (partial) {
        // workgroup sum of the per-lane (dgamma, dbeta) partials without LDS atomics (those serialised 4-8 ways per address and cost as much as
        // the streaming part of the kernel): row groups inside a wave by butterfly, then one plain store per wave and a 4-way sum
#pragma unroll
        for (int i = 0; i < L::CPL; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int o = L::LPR; o < 64; o <<= 1) { ag[i][j] += __shfl_xor(ag[i][j], o, 64); ab[i][j] += __shfl_xor(ab[i][j], o, 64); }
            }
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < L::CPL; ++i) {
                const int ch = sub + i * L::LPR;
                if (ch < L::CH) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { red[wave][0][ch * 8 + j] = ag[i][j]; red[wave][1][ch * 8 + j] = ab[i][j]; }
                }
            }
        }
        __syncthreads();
        float* out = partial + (long)blockIdx.x * 2 * C;                  // plain stores: no cross-workgroup contention
        for (int i = threadIdx.x; i < 2 * C; i += 256)
            out[i] = ((&red[0][0][0])[i] + (&red[1][0][0])[i]) + ((&red[2][0][0])[i] + (&red[3][0][0])[i]);
    }
}

// dgamma[c] += sum_b partial[b][0][c]; dbeta[c] += sum_b partial[b][1][c].  Block = 32 columns x 64 partial rows (8 row lanes x 8 rows each);
// grid.y covers the partial rows, so the 6 MB of partials of a 2048-workgroup launch are summed by ~1500 workgroups, then ONE atomic per
// (column, 64-row chunk) -- 32x fewer contended atomics than adding from the main kernel.
__global__ __launch_bounds__(256) void layernorm_bwd_reduce_kernel(const float* __restrict__ partial, int nblocks, int C,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;                      // index into the 2*C concatenated [gamma | beta] sums
    const int b0 = blockIdx.y * 64 + rl * 8;
    float s = 0.f;
    if (i < 2 * C) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (b0 + u < nblocks) ? partial[(long)(b0 + u) * 2 * C + i] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && i < 2 * C) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) t += red[r][cl];
        atomicAdd(i < C ? dgamma + i : dbeta + (i - C), t);
    }
}

#define LN_DISPATCH(C_, KERNEL, U64, U128, U192, U384, U768, ...)                                                     \
    switch (C_) {                                                                                                        \
        case 64: CXR_LAUNCH((KERNEL<64, U64>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;                    \
        case 128: CXR_LAUNCH((KERNEL<128, U128>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;                 \
        case 192: CXR_LAUNCH((KERNEL<192, U192>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;                 \
        case 384: CXR_LAUNCH((KERNEL<384, U384>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;                 \
        case 768: CXR_LAUNCH((KERNEL<768, U768>), dim3(grid), dim3(256), 0, stream, __VA_ARGS__); break;                 \
        default: return CXR_ERR_ARG;                                                                                     \
    }

static inline int ln_grid(long rows, int C, int U, bool fwd) {
    const int ch = C / 8;
    const bool tri = fwd && (ch % 3 == 0) && ((ch / 3) & (ch / 3 - 1)) == 0 && ch / 3 >= 4;               // must mirror LNCfg<C, fwd>
    const int lpr = tri ? ch / 3 : (ch > 32 ? 64 : (ch > 16 ? 32 : (ch > 8 ? 16 : (ch > 4 ? 8 : 4))));
    const long rpb = 4 * (64 / lpr) * U;
    long g = (rows + rpb - 1) / rpb;
    return (int)(g < 4096 ? g : 4096);
}

extern "C" int cxr_layernorm_fwd_bf16(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
                                      float* stats, long rows, int C, float eps, hipStream_t stream) {
    if (rows <= 0 || (ldx % 8) || (ldy % 8)) return CXR_ERR_ARG;
    const int grid = ln_grid(rows, C, (C == 64 || C == 128) ? 4 : 2, true);
    LN_DISPATCH(C, layernorm_fwd_kernel, 4, 4, 2, 2, 2, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, stats, rows, eps);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// LayerNorm whose consumer is an e4m3 GEMM: y8[rows, C] = e4m3(LN(x) * inv_scale); y (bf16) may be null
extern "C" int cxr_layernorm_q8_bf16(const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy, void* y8, long ldy8,
                                     float inv_scale, float* stats, long rows, int C, float eps, hipStream_t stream) {
    if (rows <= 0 || (ldx % 8) || (y && (ldy % 8)) || !y8 || (ldy8 % 8) || (((size_t)y8) % 8)) return CXR_ERR_ARG;
    const int grid = ln_grid(rows, C, (C == 64 || C == 128) ? 4 : 2, true);
#define LN_Q8(C_, U_) CXR_LAUNCH((layernorm_fwd_kernel<C_, U_, true>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, gamma, beta, (bf16_t*)y, ldy, \
                                 stats, rows, eps, (unsigned char*)y8, ldy8, inv_scale)
    switch (C) {
        case 64: LN_Q8(64, 4); break;
        case 128: LN_Q8(128, 4); break;
        case 192: LN_Q8(192, 2); break;
        case 384: LN_Q8(384, 2); break;
        case 768: LN_Q8(768, 2); break;
        default: return CXR_ERR_ARG;
    }
#undef LN_Q8
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// workspace: fp32 [cxr_layernorm_bwd_grid(rows, C)][2][C] (may be null when dgamma/dbeta are not wanted)
extern "C" int cxr_layernorm_bwd_grid(long rows, int C) {
    int grid = ln_grid(rows, C, C == 192 ? 4 : (C == 768 ? 1 : 2), false);
    static int cap = -1;                                   // CXR_LN_BWD_GRID: workgroup cap of the grid-stride loop (lab switch)
    if (cap < 0) { const char* e = getenv("CXR_LN_BWD_GRID"); cap = e ? atoi(e) : 512; if (cap < 64) cap = 512; }
    return grid < cap ? grid : cap;
}

extern "C" int cxr_layernorm_bwd_bf16(const void* x, long ldx, const void* dy, long lddy, const float* gamma, const float* stats,
                                      const void* add, long ldadd, void* dx, long lddx, float* dgamma, float* dbeta, float* workspace,
                                      long rows, int C, void* dx2, long lddx2, float drop_p, const unsigned int* drop_seed, unsigned int drop_site,
                                      int drop_rows_per_b, int drop_t0, const float* row_scale, hipStream_t stream) {
    if (rows <= 0 || (ldx % 8) || (lddy % 8) || (lddx % 8) || (add && (ldadd % 8)) || (dgamma && (!workspace || !dbeta)) || (!dx && !dx2)) return CXR_ERR_ARG;
    if (dx2 && ((lddx2 % 8) || drop_rows_per_b <= 0 || (!row_scale && (drop_p <= 0.f || drop_p >= 1.f || !drop_seed)))) return CXR_ERR_ARG;
    LnBwdDrop dd;
    dd.dx2 = (bf16_t*)dx2; dd.lddx2 = lddx2; dd.seed = drop_seed; dd.site = drop_site; dd.thr16 = (dx2 && !row_scale) ? dropout_thr16(drop_p) : 0u;
    dd.inv = drop_p < 1.f ? 1.0f / (1.0f - drop_p) : 1.f; dd.rows_per_b = drop_rows_per_b > 0 ? drop_rows_per_b : 1; dd.t0 = drop_t0;
    dd.row_scale = row_scale;
    const int grid = cxr_layernorm_bwd_grid(rows, C);
    float* partial = workspace;                                   // partial rows are written whenever a workspace is given; the caller may run
                                                                  // cxr_layernorm_bwd_reduce on another stream (dgamma == NULL here)
    static int pf = -1;                                           // CXR_LN_BWD_PF=0: the un-pipelined loop of rounds 1-5 (A/B)
    if (pf < 0) { const char* e = getenv("CXR_LN_BWD_PF"); pf = (e && e[0] == '0') ? 0 : 1; }
#define LN_BWD(C_, U_) do { if (pf) CXR_LAUNCH((layernorm_bwd_kernel<C_, U_, true>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, gamma, stats, \
                                               (const bf16_t*)add, ldadd, (bf16_t*)dx, lddx, partial, rows, dd);                                                              \
                            else CXR_LAUNCH((layernorm_bwd_kernel<C_, U_, false>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, gamma, stats, \
                                            (const bf16_t*)add, ldadd, (bf16_t*)dx, lddx, partial, rows, dd); } while (0)
    switch (C) {
        case 64: LN_BWD(64, 2); break;
        case 128: LN_BWD(128, 2); break;
        // (C = 192 at U = 4 needs 256 registers with the second stage: measured 49.9 -> 64.5 us at 147456 rows; it keeps the old loop.
        //  384: 30.7 -> 28.2 us at 36928 rows, 64: 57.1 -> 54.1 at 589824, 768: 16.6 -> 16.4 at 8192 -- gpurun_out r6 call 3, profiles/r06_ln_bwd.txt)
        case 192: CXR_LAUNCH((layernorm_bwd_kernel<192, 4, false>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, gamma, stats,
                             (const bf16_t*)add, ldadd, (bf16_t*)dx, lddx, partial, rows, dd); break;
        case 384: LN_BWD(384, 2); break;
        case 768: LN_BWD(768, 1); break;
        default: return CXR_ERR_ARG;
    }
#undef LN_BWD
    if (dgamma) CXR_LAUNCH(layernorm_bwd_reduce_kernel, dim3(cdiv(2 * C, 32), cdiv(grid, 64)), dim3(256), 0, stream, partial, grid, C, dgamma, dbeta);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// second half of cxr_layernorm_bwd_bf16 on its own (parameter gradients only feed the optimiser: the caller may put this on a side stream)
extern "C" int cxr_layernorm_bwd_reduce(const float* workspace, long rows, int C, float* dgamma, float* dbeta, hipStream_t stream) {
    if (!workspace || !dgamma || !dbeta || rows <= 0) return CXR_ERR_ARG;
    const int grid = cxr_layernorm_bwd_grid(rows, C);
    CXR_LAUNCH(layernorm_bwd_reduce_kernel, dim3(cdiv(2 * C, 32), cdiv(grid, 64)), dim3(256), 0, stream, workspace, grid, C, dgamma, dbeta);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
