// Decode-step linear layers (SURVEY.md 2.3 K9-K12 at query length 1; TF5:bert:164-203,289-351,466-496 as driven by the cached decode of
// TF5:gen:2783-2973), built for what actually bounds a 2-5 us kernel on gfx950 (measured with per-wave s_memrealtime stamps,
// scripts/lab/decode_lab.hip):
//   * a wave's critical path is counted in INSTRUCTIONS (~3.3 ns per dependent VALU instruction at one wave per SIMD) and in VECTOR-MEMORY
//     INSTRUCTIONS PER CU: the CU's address / data path retires one 64-lane load or store per ~45-55 cycles whatever its width or locality
//     (72 fragment loads of 1 KB = 1.7 us; 40 dword loads = 1.0 us). So: few, wide (16 B per lane) memory instructions, none of them guarded by a
//     branch (hipcc parks `s_waitcnt vmcnt(0)` behind a conditional load: the round-1 kernel thereby made ~10 dependent round trips before its
//     first weight load), kernel arguments fetched by ONE scalar batch, epilogue operands packed per column.
//   * LayerNorm FOLDED INTO THE WEIGHTS:  LN(x) . W^T + b = rstd * (x . W'^T - mean * s) + b',  W' = W diag(gamma), s_n = sum_k W'[n,k],
//     b' = b + W beta: the GEMM reads the RAW rows, the normalisation is two FMAs per OUTPUT element (round 1 normalised 96 elements per lane in
//     every consumer: two-pass statistics, two barriers, 24 gamma / beta loads per lane). Row statistics come from the PRODUCER of x: each of its
//     workgroups publishes the (sum, M2 about the tile mean) of its 16 output columns per row; a consumer combines the N/16 partials (Chan's
//     parallel-variance update: deterministic, no atomics, no cancellation) while its weights are in flight.
//   * OPERANDS IN MFMA-FRAGMENT ORDER. Weights are re-laid out once per weight version (cxr_dec_pack_weight_bf16): the 16 x 32 block a wave feeds
//     to one v_mfma_f32_16x16x32_bf16 is 1 KB of consecutive bytes in lane order, so a weight tile is ONE contiguous stream. Activations between the
//     kernels of a decode step live in the same order ("decode activation layout": element (m, k) of an [Mpad, K] matrix at
//     ((k/32)*(Mpad/16) + m/16)*512 + ((k%32)/8*16 + m%16)*8 + k%8): producers scatter 4-byte pairs, consumers read whole fragments.
//
// Layout of a workgroup: NW waves split K; a workgroup owns 16*NC output columns of ONE problem of a grouped launch (q / k / v; blockIdx.y) for
// all M <= 64 rows; weights go straight from HBM into MFMA B fragments (read once chip-wide: no LDS round trip); partial accumulators meet in
// LDS; the epilogue handles two adjacent columns per thread.
//
// Train-mode LoRA on the query / key projections (REF:modelling_longitudinal.py:162-171, peft Linear: base(x) + (alpha/r) B(A(dropout(x)))):
// with x = LN(raw) and an element-wise mask m in {0, 1}: dropout(LN(raw)) . A^T = 1/(1-p) * ( rstd * ((m*raw) . A'^T - mean * (m . A'^T)) + m . A''^T ),
// A' = A diag(gamma), A'' = A diag(beta). [A'; A''] is ONE 16-row B operand: two extra MFMAs per k-step (A operand m*raw, then m) give all three
// products (columns 0-7 / 0-7 / 8-15).
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include "../../include/cxrmate_hip.h"

#ifndef CXR_STAMP
#define CXR_STAMP(i)
#endif

namespace {

// Every pointer below is a VALID device address: the host entry substitutes a harmless readable address (problem 0's weights) for absent
// operands and says so in `flags`, so that the kernel's load phase has no branch.
struct DecProb { const bf16_t* Wp; const float2* bc; void* C; const bf16_t* lr_Ap; const bf16_t* lr_B; long ldc;
                 int N; uint32_t lr_site, flags, c_dal; };                        // flags: 1 bias, 2 colsum (LN folded), 4 LoRA; c_dal: C in decode activation layout
struct DecArgs {
    const bf16_t* A; const bf16_t* residual; const float* stats; const float2* rgb; float* out_stats;
    const uint32_t* drop_seed; const uint32_t* lr_seed;
    long ldr;                                                                    // 0: residual in decode activation layout
    int M, K, act, out_f32, stats_tiles; uint32_t flags;                          // flags: 1 residual, 2 residual is LayerNorm'ed, 4 out_stats, 8 stats given
    float eps; uint32_t drop_site, drop_thr16; float drop_inv; int drop_t; uint32_t lr_thr16; float lr_inv, lr_scale; int lr_t;
    int mtl;                                                                     // 16-row tiles of the activation LAYOUT (>= MT of one workgroup)
    DecProb p[3];
};

// MT = 16-row tiles of A per workgroup (blockIdx.z selects the tile group: with MT = 1 a 32-row step runs as two workgroups per column tile,
// each pulling half of the activations through its CU); NW = waves per workgroup, each owning KB k-steps of 32 (K = NW*KB*32); NC = 16-column tiles per
// workgroup sharing the A fragments (4 for the vocabulary projection); LORA = the launch carries a LoRA branch.
template <int MT, int NW, int KB, int NC, bool LORA>
__global__ __launch_bounds__(NW * 64) void dec_gemm_kernel(const DecArgs g) {
    constexpr int NT = NW * 64;
    constexpr int ROWS = MT * 16;
    constexpr int TW = NC * 16;
    constexpr int KS = NW * KB;                          // k-steps of the whole reduction
    constexpr int TPR = NT / ROWS;                      // threads per row in the statistics combine
    constexpr int MAXP = (48 + TPR - 1) / TPR;          // partials per thread: N = 768 -> 48 tiles (host checks stats_tiles <= MAXP * TPR)
    constexpr int PAIRS = ROWS * TW / 2;                // the epilogue handles two adjacent columns per thread
    constexpr int EPT = (PAIRS + NT - 1) / NT;
    __shared__ float red[NW][MT][NC][64][4];
    __shared__ float redl[LORA ? 2 : 1][LORA ? NW : 1][LORA ? MT : 1][64][4];     // LoRA products (tile 0: (m*x).[A';A''], tile 1: m.[A';A''])
    __shared__ float s_mean[ROWS], s_rstd[ROWS];
    __shared__ float s_t[LORA ? ROWS : 1][8];           // LoRA down-projection t[m][r]
    CXR_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const DecProb P = g.p[blockIdx.y];
    // every kernel argument the load phase needs is pulled into SGPRs by ONE batch of scalar loads (hipcc otherwise fetches them where first
    // used: three dependent kernarg round trips of 0.3 us each before the first weight load)
    asm volatile("" :: "s"(g.A), "s"(g.residual), "s"(g.stats), "s"(g.rgb), "s"(g.drop_seed), "s"(g.lr_seed), "s"(g.ldr), "s"(g.M), "s"(g.stats_tiles));
    asm volatile("" :: "s"(P.Wp), "s"(P.bc), "s"(P.lr_Ap), "s"(P.lr_B), "s"(P.N));
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));      // a zero hipcc cannot see through: the seed reads stay plain per-lane loads (a uniform
    CXR_STAMP(6);                                       // address makes it wait for the value and v_readfirstlane it in the middle of the load phase)
    const int tile0 = blockIdx.x * NC;                  // (tiles beyond a narrower grouped problem compute on the last tile and store nothing)
    const int row0 = blockIdx.z * ROWS, MTL = g.mtl;
    const int ntiles = (P.N + 15) >> 4;                 // (the packed weights are zero-padded to whole 16-column tiles)
    // ---- the weight stream and the activation rows first: every load of the wave is in flight before anything waits
    bf16x8_t wf[NC][KB], af[MT][KB], lf[LORA ? KB : 1];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        int tl = tile0 + c; tl = tl < ntiles ? tl : ntiles - 1;
        const bf16_t* wp = P.Wp + ((long)(tl * KS + wave * KB) * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < KB; ++s) wf[c][s] = *reinterpret_cast<const bf16x8_t*>(wp + s * 512);
    }
    {
        const bf16_t* ap = g.A + ((long)(wave * KB * MTL + blockIdx.z * MT) * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < KB; ++s)
#pragma unroll
            for (int t = 0; t < MT; ++t) af[t][s] = *reinterpret_cast<const bf16x8_t*>(ap + (s * MTL + t) * 512);
    }
    if (LORA) {
        const bf16_t* lp = P.lr_Ap + ((long)(wave * KB) * 64 + lane) * 8;
#pragma unroll
        for (int s = 0; s < KB; ++s) lf[s] = *reinterpret_cast<const bf16x8_t*>(lp + s * 512);
    }
    CXR_STAMP(7);
    // ---- epilogue operands of this thread's column pairs (raw: converted after the MFMAs) and the statistics partials
    float4 e_bc[EPT], e_rgb[EPT];
    uint32_t e_res[EPT];
    uint4 e_lb[LORA ? EPT : 1][2];
    const int n0 = tile0 * 16;
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;                                               // pair index: row = e / (TW/2), columns 2*(e % (TW/2)), +1
        int row = row0 + e / (TW / 2), n = n0 + 2 * (e % (TW / 2));
        row = row < g.M ? row : g.M - 1; n = n < P.N ? n : P.N - 2;
        e_bc[i] = *reinterpret_cast<const float4*>(P.bc + n);
        e_rgb[i] = *reinterpret_cast<const float4*>(g.rgb + n);
        e_res[i] = *reinterpret_cast<const uint32_t*>(g.residual + (g.ldr ? (long)row * g.ldr + n : dal_off(row, n, MTL)));
        if (LORA) {
            e_lb[i][0] = *reinterpret_cast<const uint4*>(P.lr_B + (long)n * 8);
            e_lb[i][1] = *reinterpret_cast<const uint4*>(P.lr_B + (long)n * 8 + 8);
        }
    }
    const int srow = tid / TPR, spart = tid % TPR;
    float2 pst[MAXP];
    {
        const int r_ = row0 + srow < g.M ? row0 + srow : g.M - 1;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            int tile = spart + j * TPR; tile = tile < g.stats_tiles ? tile : 0;
            pst[j] = *reinterpret_cast<const float2*>(g.stats + ((long)tile * g.M + r_) * 2);
        }
    }
    const uint32_t dseed = g.drop_seed[vzero], lseed = g.lr_seed[vzero];
    __builtin_amdgcn_sched_barrier(0);                  // every load above is ISSUED before the first MFMA waits (hipcc otherwise interleaves them)
    CXR_STAMP(1);
    const bool lora = LORA && (P.flags & 4u);
    const bool fold = (P.flags & 2u) != 0;
    f32x4_t acc[MT][NC], al0[MT], al1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[t][c] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        al0[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; al1[t] = al0[t];
    }
#pragma unroll
    for (int s = 0; s < KB; ++s)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][s], wf[c][s], acc[t][c], 0, 0, 0);
    if (LORA) {
        if (lora) {
            // LoRA branch input: dropout mask of (row, position lr_t, column k) -- the hash of lora.hip / the teacher-forced re-scoring pass
            const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
            for (int s = 0; s < KB; ++s) {
                const int kcol = (wave * KB + s) * 32 + fq * 8;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    int m = row0 + t * 16 + fr; m = m < g.M ? m : g.M - 1;
                    s16x8_t xm = __builtin_bit_cast(s16x8_t, af[t][s]), mk;
                    const uint32_t key = dropout_row_key(lseed, P.lr_site, (uint32_t)m, (uint32_t)g.lr_t);
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const uint32_t bits = dropout_pair_bits(key, (uint32_t)(kcol + j) >> 1);
                        const bool k0_ = (bits & 0xffffu) >= g.lr_thr16, k1_ = (bits >> 16) >= g.lr_thr16;      // lr_thr16 == 0: everything kept
                        xm[j] = k0_ ? xm[j] : (short)0; xm[j + 1] = k1_ ? xm[j + 1] : (short)0;
                        mk[j] = k0_ ? (short)0x3F80 : (short)0; mk[j + 1] = k1_ ? (short)0x3F80 : (short)0;
                    }
                    al0[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, xm), lf[s], al0[t], 0, 0, 0);
                    al1[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, mk), lf[s], al1[t], 0, 0, 0);
                }
            }
        }
    }
    CXR_STAMP(2);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
#pragma unroll
        for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4_t*>(&red[wave][t][c][lane][0]) = acc[t][c];
        if (LORA) {
            *reinterpret_cast<f32x4_t*>(&redl[0][wave][t][lane][0]) = al0[t];
            *reinterpret_cast<f32x4_t*>(&redl[1][wave][t][lane][0]) = al1[t];
        }
    }
    {
        // Chan et al.: n_i = 16 per tile. mean = sum S_i / n ; M2 = sum M2_i + 16 * (S_i/16 - mean)^2   (stats_tiles == 0: unused values)
        float S = 0.f;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) S += (spart + j * TPR < g.stats_tiles) ? pst[j].x : 0.f;
        S = group_sum<TPR>(S);
        const float ntot = 16.0f * (float)(g.stats_tiles > 0 ? g.stats_tiles : 1);
        const float mean = S / ntot;
        float Q = 0.f;
#pragma unroll
        for (int j = 0; j < MAXP; ++j) {
            const float d = pst[j].x * (1.0f / 16.0f) - mean;
            Q += (spart + j * TPR < g.stats_tiles) ? pst[j].y + 16.0f * d * d : 0.f;
        }
        Q = group_sum<TPR>(Q);
        if (spart == 0) { s_mean[srow] = mean; s_rstd[srow] = rsqrtf(Q / ntot + g.eps); }
    }
    CXR_STAMP(3);
    __syncthreads();
    CXR_STAMP(4);
    if (LORA) {
        if (lora) {
            // t[m][r] = lr_scale/(1-p) * ( rstd_m * (P1[m][r] - mean_m * P2[m][r]) + P3[m][r] ); element (m, c) of a 16x16 D tile sits in
            // lane (m%16/4)*16 + c, register m%4
            for (int e = tid; e < ROWS * 8; e += NT) {
                const int row = e >> 3, r8 = e & 7;
                const int t = row >> 4, l1 = ((row & 15) >> 2) * 16 + r8, rr = row & 3;
                float p1 = 0.f, p2 = 0.f, p3 = 0.f;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    p1 += redl[0][w][t][l1][rr];
                    p2 += redl[1][w][t][l1][rr];
                    p3 += redl[1][w][t][l1 + 8][rr];
                }
                const float tv = fold ? s_rstd[row] * (p1 - s_mean[row] * p2) + p3 : p1;
                s_t[row][r8] = tv * g.lr_scale * g.lr_inv;
            }
        }
        __syncthreads();
    }
    const bool has_stats = (g.flags & 8u) != 0, r_ln = (g.flags & 2u) != 0 && blockIdx.y == 0, has_res = (g.flags & 1u) != 0 && blockIdx.y == 0;
    float o_val[EPT][2];
#pragma unroll
    for (int i = 0; i < EPT; ++i) {
        const int e = tid + i * NT;
        const int lrow = e / (TW / 2), row = row0 + lrow, col = 2 * (e % (TW / 2)), n = n0 + col;       // lrow: row inside this workgroup's tiles
        const int t = lrow >> 4, l1 = ((lrow & 15) >> 2) * 16 + (col & 15), rr = lrow & 3, ct = col >> 4;
        const bool in_tile = e < PAIRS;
        float v0 = 0.f, v1 = 0.f;
        if (in_tile) {
#pragma unroll
            for (int w = 0; w < NW; ++w) { v0 += red[w][t][ct][l1][rr]; v1 += red[w][t][ct][l1 + 1][rr]; }
        }
        const float mean = (has_stats && in_tile) ? s_mean[lrow] : 0.f, rstd = (has_stats && in_tile) ? s_rstd[lrow] : 1.f;
        if (fold) { v0 = rstd * (v0 - mean * e_bc[i].y); v1 = rstd * (v1 - mean * e_bc[i].w); }
        if (P.flags & 1u) { v0 += e_bc[i].x; v1 += e_bc[i].z; }
        if (LORA) {
            if (lora && in_tile) {
                float lb0[8], lb1[8];
                unpack8(e_lb[i][0], lb0); unpack8(e_lb[i][1], lb1);
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) { const float tv = s_t[lrow][r8]; v0 += tv * lb0[r8]; v1 += tv * lb1[r8]; }
            }
        }
        if (g.act == 1) { v0 = gelu_f(v0); v1 = gelu_f(v1); }
        if (g.drop_thr16) {                                                        // columns n, n+1 (n even) share one hash word
            const uint32_t bits = dropout_pair_bits(dropout_row_key(dseed, g.drop_site, (uint32_t)row, (uint32_t)g.drop_t), (uint32_t)n >> 1);
            v0 = (bits & 0xffffu) >= g.drop_thr16 ? v0 * g.drop_inv : 0.f;
            v1 = (bits >> 16) >= g.drop_thr16 ? v1 * g.drop_inv : 0.f;
        }
        float r0 = has_res ? __uint_as_float(e_res[i] << 16) : 0.f, r1 = has_res ? __uint_as_float(e_res[i] & 0xffff0000u) : 0.f;
        if (r_ln) { r0 = (r0 - mean) * rstd * e_rgb[i].x + e_rgb[i].y; r1 = (r1 - mean) * rstd * e_rgb[i].z + e_rgb[i].w; }
        v0 += r0; v1 += r1;
        const bool ok = in_tile && row < g.M && n < P.N;
        if (g.out_f32) { if (ok) *reinterpret_cast<float2*>(reinterpret_cast<float*>(P.C) + (long)row * P.ldc + n) = make_float2(v0, v1); }
        else {
            const uint32_t pk = pack2bf(v0, v1);
            if (ok) *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(P.C) + (P.c_dal ? dal_off(row, n, MTL) : (long)row * P.ldc + n)) = pk;
            v0 = __uint_as_float(pk << 16); v1 = __uint_as_float(pk & 0xffff0000u);    // the statistics describe what the consumers will read
        }
        o_val[i][0] = ok ? v0 : 0.f; o_val[i][1] = ok ? v1 : 0.f;
    }
    if (NC == 1) {
        if ((g.flags & 4u) && blockIdx.y == 0) {
            // partial statistics of this tile's 16 columns per row: 8 consecutive lanes hold one row
#pragma unroll
            for (int i = 0; i < EPT; ++i) {
                const int e = tid + i * NT;
                const int row = row0 + (e >> 3);
                const float S = group_sum<8>(o_val[i][0] + o_val[i][1]);
                const float d0 = o_val[i][0] - S * (1.0f / 16.0f), d1 = o_val[i][1] - S * (1.0f / 16.0f);
                const float M2 = group_sum<8>(d0 * d0 + d1 * d1);
                if ((e & 7) == 0 && e < PAIRS && row < g.M && n0 < P.N)
                    *reinterpret_cast<float2*>(g.out_stats + ((long)blockIdx.x * g.M + row) * 2) = make_float2(S, M2);
            }
        }
    }
    CXR_STAMP(5);
}

// One pass per weight version: W [N, K] row-major bf16 -> Wp in MFMA-fragment order (block (tile j, k-step s) = 1 KB: lane l holds
// W'[16 j + l%16][32 s + 8 (l/16) .. +8]), W' = W diag(gamma) (gamma == NULL: W' = W), and bc[n] = (bias_n + sum_k W[n,k] beta_k, sum_k W'[n,k]).
// colsum sums the ROUNDED values: the epilogue's mean * colsum must cancel what the MFMAs summed. One wave per output row.
__global__ __launch_bounds__(256) void dec_pack_weight_kernel(const bf16_t* __restrict__ W, long ldw, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ bias, bf16_t* __restrict__ Wp,
                                                              float2* __restrict__ bc, int N, int K) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= ((N + 15) & ~15)) return;
    const int KS = K >> 5;
    float cs = 0.f, bs = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        float w[8], o[8];
        unpack8(n < N ? *reinterpret_cast<const uint4*>(W + (long)n * ldw + k) : make_uint4(0, 0, 0, 0), w);      // rows past N: zero padding of the last tile
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = bf2f(f2bf(gamma ? w[j] * gamma[k + j] : w[j]));
            cs += o[j];
            bs += beta ? w[j] * beta[k + j] : 0.f;
        }
        const long blk = (long)(n >> 4) * KS + (k >> 5);
        *reinterpret_cast<uint4*>(Wp + (blk * 64 + (((k & 31) >> 3) << 4) + (n & 15)) * 8) = pack8(o);
    }
    cs = group_sum<64>(cs); bs = group_sum<64>(bs);
    if (lane == 0 && n < N) bc[n] = make_float2((bias ? bias[n] : 0.f) + bs, cs);
}

// LoRA: A bf16 [8][K] -> fragment-ordered [K/32][64 lanes][8]: B-operand rows 0-7 = A diag(gamma), rows 8-15 = A diag(beta)
// (gamma == NULL: rows 0-7 = A, rows 8-15 = 0)
__global__ __launch_bounds__(256) void dec_pack_lora_kernel(const bf16_t* __restrict__ A, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            bf16_t* __restrict__ out, int K) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 8 * K) return;
    const int r = i / K, k = i % K;
    const float a = bf2f(A[i]);
    const long base = ((long)(k >> 5) * 64 + (((k & 31) >> 3) << 4)) * 8 + (k & 7);
    out[base + r * 8] = f2bf(gamma ? a * gamma[k] : a);
    out[base + (r + 8) * 8] = f2bf(gamma ? a * beta[k] : 0.f);
}

// row-major [M, K] bf16 -> decode activation layout [16*MT, K], and the per-row (sum, M2) partials over 16-column tiles a dec_gemm epilogue
// would have published (stats == NULL: layout only). One thread per (row, 8-column chunk).
__global__ __launch_bounds__(256) void dec_to_dal_kernel(const bf16_t* __restrict__ x, long ldx, int M, int K, int MT, bf16_t* __restrict__ out,
                                                         float* __restrict__ stats) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int chunks = K / 8;
    if (i >= M * chunks) return;
    const int row = i / chunks, k = (i % chunks) * 8;
    const uint4 raw = *reinterpret_cast<const uint4*>(x + (long)row * ldx + k);
    if (out) *reinterpret_cast<uint4*>(out + dal_off(row, k, MT)) = raw;
    if (stats) {
        float v[8];
        unpack8(raw, v);
        float S = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) S += v[j];
        const float S16 = S + __shfl_xor(S, 1, 64);                   // the two chunks of a 16-column tile sit in adjacent lanes
        float M2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - S16 * (1.0f / 16.0f); M2 += d * d; }
        M2 += __shfl_xor(M2, 1, 64);
        if (!(i & 1)) *reinterpret_cast<float2*>(stats + ((long)(k >> 4) * M + row) * 2) = make_float2(S16, M2);
    }
}

// decode activation layout -> row-major (tests, and consumers outside the decode chain)
__global__ __launch_bounds__(256) void dec_from_dal_kernel(const bf16_t* __restrict__ x, int M, int K, int MT, bf16_t* __restrict__ out, long ldo) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int chunks = K / 8;
    if (i >= M * chunks) return;
    const int row = i / chunks, k = (i % chunks) * 8;
    *reinterpret_cast<uint4*>(out + (long)row * ldo + k) = *reinterpret_cast<const uint4*>(x + dal_off(row, k, MT));
}

}  // namespace

extern "C" int cxr_dec_gemm_bf16(const cxr_dec_gemm_desc* d, hipStream_t stream) {
    if (!d) return CXR_ERR_ARG;
    if (d->M <= 0 || d->M > 64 || d->nprob < 1 || d->nprob > 3 || !d->A) return CXR_ERR_ARG;
    if (d->drop_p < 0.f || d->drop_p >= 1.f || (d->drop_p > 0.f && !d->drop_seed) || d->lr_p < 0.f || d->lr_p >= 1.f) return CXR_ERR_ARG;
    const void* dummy = d->p[0].Wp;                    // readable stand-in for absent operands (see DecProb)
    if (!dummy) return CXR_ERR_ARG;
    DecArgs g;
    memset(&g, 0, sizeof(g));
    g.A = (const bf16_t*)d->A; g.M = d->M; g.K = d->K; g.act = d->act; g.out_f32 = d->out_f32;
    int nmax = 0;
    bool any_lora = false;
    for (int i = 0; i < 3; ++i) {
        const cxr_dec_gemm_prob& q = d->p[i < d->nprob ? i : 0];
        if (q.N < 16 || (q.N % 2) || !q.Wp || !q.bc || !q.C || (q.lr_Ap && !q.lr_B) || (q.fold && !d->stats)) return CXR_ERR_ARG;
        if (!q.c_dal && (q.ldc % 2)) return CXR_ERR_ARG;
        if (q.c_dal && d->out_f32) return CXR_ERR_ARG;
        DecProb& P = g.p[i];
        P.Wp = (const bf16_t*)q.Wp; P.bc = (const float2*)q.bc; P.C = q.C; P.ldc = q.ldc; P.N = q.N; P.lr_site = q.lr_site; P.c_dal = q.c_dal ? 1u : 0u;
        P.lr_Ap = q.lr_Ap ? (const bf16_t*)q.lr_Ap : (const bf16_t*)dummy; P.lr_B = q.lr_B ? (const bf16_t*)q.lr_B : (const bf16_t*)dummy;
        P.flags = (q.no_bias ? 0u : 1u) | (q.fold ? 2u : 0u) | (q.lr_Ap ? 4u : 0u);
        if (i < d->nprob) { nmax = q.N > nmax ? q.N : nmax; any_lora |= q.lr_Ap != nullptr; }
    }
    if (d->stats && (d->stats_tiles <= 0 || d->stats_tiles > 48)) return CXR_ERR_ARG;
    if (d->rgb && (!d->residual || !d->stats)) return CXR_ERR_ARG;
    if (any_lora && d->lr_p > 0.f && !d->lr_seed) return CXR_ERR_ARG;
    if ((d->ldr % 2) || (d->out_stats && (d->p[0].N % 16))) return CXR_ERR_ARG;
    g.stats = d->stats ? d->stats : (const float*)dummy; g.stats_tiles = d->stats ? d->stats_tiles : 0; g.eps = d->eps;
    g.residual = d->residual ? (const bf16_t*)d->residual : (const bf16_t*)dummy; g.ldr = d->residual ? d->ldr : 2;
    g.rgb = d->rgb ? (const float2*)d->rgb : (const float2*)dummy;
    g.out_stats = d->out_stats;
    g.flags = (d->residual ? 1u : 0u) | (d->rgb ? 2u : 0u) | (d->out_stats ? 4u : 0u) | (d->stats ? 8u : 0u);
    g.drop_seed = d->drop_seed ? d->drop_seed : (const uint32_t*)dummy; g.drop_site = d->drop_site;
    g.drop_thr16 = d->drop_p > 0.f ? dropout_thr16(d->drop_p) : 0u; g.drop_inv = 1.0f / (1.0f - d->drop_p); g.drop_t = d->drop_t;
    g.lr_seed = d->lr_seed ? d->lr_seed : (const uint32_t*)dummy; g.lr_thr16 = d->lr_p > 0.f ? dropout_thr16(d->lr_p) : 0u;
    g.lr_inv = 1.0f / (1.0f - d->lr_p); g.lr_scale = d->lr_scale; g.lr_t = d->lr_t;
    const int mtl = cdiv(g.M, 16) == 3 ? 4 : cdiv(g.M, 16);
    g.mtl = mtl;
    // geometry: K = 768 -> 8 waves x 3 k-steps; K = 3072 -> 16 waves x 6 k-steps; vocabulary-sized problems take 64 columns per workgroup.
    // Row tiles: launches that still fit one workgroup per CU afterwards (the 768-wide single problems: 48 column tiles) run ONE 16-row tile
    // per workgroup (grid.z = tiles): a CU retires ~46 GB/s of loads whatever the mix, so halving the activation rows per workgroup shortens
    // the load phase (768x768: 6.2 -> 5.7 us, 768x3072: 9.1 -> 8.2); wider launches lose more to the second wave of workgroups than they gain
    int nc = (d->nprob == 1 && nmax >= 8192 && !d->out_stats && !any_lora) ? 4 : 1;
    if (d->nc_hint == 1 || (d->nc_hint == 4 && !d->out_stats && !any_lora)) nc = d->nc_hint;
    if (g.K != 768 && g.K != 3072) return CXR_ERR_ARG;            // instantiated reductions: BERT-base hidden / intermediate size
    if (g.K == 3072 && (any_lora || nc != 1)) return CXR_ERR_ARG;
    const bool fits = (long)cdiv(nmax, 16 * nc) * d->nprob * mtl <= 256;                          // one workgroup per CU after the split
    int mt = (d->mt_hint > 0 ? d->mt_hint == 1 : (fits && mtl > 1)) ? 1 : mtl;                     // tiles per workgroup
    if (g.K == 3072 && mt == 4) mt = 2;                            // (16 waves x 4 row tiles x 6 k-steps need 96 registers of A fragments alone: 128-register cap, spills)
    const dim3 grid(cdiv(nmax, 16 * nc), d->nprob, mtl / mt);
#define DG(MT_, NW_, KB_, NC_, L_) CXR_LAUNCH((dec_gemm_kernel<MT_, NW_, KB_, NC_, L_>), grid, dim3(NW_ * 64), 0, stream, g)
#define DGM(MT_, MT3072_) do {                                                                                     \
        if (g.K == 3072) DG(MT3072_, 16, 6, 1, false);                                                              \
        else if (any_lora) DG(MT_, 8, 3, 1, true); else if (nc == 4) DG(MT_, 8, 3, 4, false); else DG(MT_, 8, 3, 1, false); \
    } while (0)
    if (mt == 1) DGM(1, 1); else if (mt == 2) DGM(2, 2); else DGM(4, 2);
#undef DGM
#undef DG
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dec_pack_weight_bf16(const void* W, long ldw, const float* gamma, const float* beta, const float* bias, void* Wp, float* bc,
                                        int N, int K, hipStream_t stream) {
    if (N <= 0 || K <= 0 || (K % 32) || (ldw % 8) || (gamma && !beta) || !Wp || !bc) return CXR_ERR_ARG;
    CXR_LAUNCH(dec_pack_weight_kernel, dim3(cdiv((N + 15) & ~15, 4)), dim3(256), 0, stream, (const bf16_t*)W, ldw, gamma, beta, bias, (bf16_t*)Wp, (float2*)bc, N, K);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dec_pack_lora_bf16(const void* A, const float* gamma, const float* beta, void* out, int K, hipStream_t stream) {
    if (K <= 0 || (K % 32) || (gamma && !beta)) return CXR_ERR_ARG;
    CXR_LAUNCH(dec_pack_lora_kernel, dim3(cdiv(8L * K, 256)), dim3(256), 0, stream, (const bf16_t*)A, gamma, beta, (bf16_t*)out, K);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dec_to_dal_bf16(const void* x, long ldx, int M, int K, void* out, float* stats, hipStream_t stream) {
    if (M <= 0 || M > 64 || K <= 0 || (K % 32) || (ldx % 8) || (!out && !stats)) return CXR_ERR_ARG;
    const int mt = cdiv(M, 16) == 3 ? 4 : cdiv(M, 16);
    CXR_LAUNCH(dec_to_dal_kernel, dim3(cdiv((long)M * (K / 8), 256)), dim3(256), 0, stream, (const bf16_t*)x, ldx, M, K, mt, (bf16_t*)out, stats);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dec_from_dal_bf16(const void* x, int M, int K, void* out, long ldo, hipStream_t stream) {
    if (M <= 0 || M > 64 || K <= 0 || (K % 32) || (ldo % 8)) return CXR_ERR_ARG;
    const int mt = cdiv(M, 16) == 3 ? 4 : cdiv(M, 16);
    CXR_LAUNCH(dec_from_dal_kernel, dim3(cdiv((long)M * (K / 8), 256)), dim3(256), 0, stream, (const bf16_t*)x, M, K, mt, (bf16_t*)out, ldo);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
