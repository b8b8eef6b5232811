// Fused scaled-dot-product attention, head_dim 64, bf16 in / fp32 softmax, flash-style (no score matrix in HBM).
// Covers every attention on the CXRMate hot path:
//   CvT stages (no mask, scale = embed_dim^-0.5, quirk Q1), BERT decoder self-attention (causal + key padding),
//   decoder cross-attention over N*576 encoder tokens (key padding from zero images, quirk Q3), CXR-BERT (key padding).
//
// gfx950 structure: workgroup = 4 waves, each wave owns 32 query rows; K/V tiles of 64 keys are register-staged
// into padded LDS images (K rows 144 B -> conflict-free ds_read_b128; V rows 192 B -> conflict-free
// ds_read_b64_tr_b16). S^T = K.Q^T is computed "swapped" with v_mfma_f32_32x32x16_bf16 so that a query row
// lives on ONE lane pair: softmax needs a single cross-lane exchange, and the exponentiated tile is already the
// B operand of O^T += V^T.P^T (accumulator-as-operand, k-permutation per cdna_hip_programming.md section 3).
#include "common.h"
#include <stdlib.h>

#define ATT_NEG (-1.0e30f)      // "masked" sentinel (finite: masked-only rows become uniform, like finfo.min in the reference)

struct AttnArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; float* LSE;
    const unsigned char* kpm;          // [B, Tk] 1 = attend, 0 = masked; may be null
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, kpm_bs;
    int B, H, Tq, Tk;
    float scale_log2e;                 // softmax scale * log2(e)
    int causal, causal_shift;          // key j visible to query i iff j <= i + causal_shift
    // dropout on the attention probabilities (TF5 modeling_bert.py:131, train mode): P*keep/(1-p) feeds P.V, the softmax sums do not change
    const uint32_t* drop_seed; uint32_t drop_site, drop_thr16; float drop_inv; int drop_t0;      // drop_thr16 == 0: off
    unsigned char* O8; long o8_bs, o8_rs; float o8_inv;      // attn_fwd2_kernel only: e4m3 output (value * o8_inv) for a consumer that is an e4m3 GEMM; O may be null then
};

constexpr int KS_STRIDE = 72;          // bf16 elements per K row in LDS (144 B)
constexpr int VS_STRIDE = 96;          // bf16 elements per V row in LDS (192 B)

__device__ __forceinline__ bf16x8_t tr_pair(const bf16_t* p0, const bf16_t* p1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(const AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KS_STRIDE];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * VS_STRIDE];
    __shared__ unsigned char Ms[64];
    __shared__ __attribute__((aligned(16))) bf16_t Os[4][32 * 64];      // per-wave output tile for the row-contiguous store

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int head = blockIdx.y, b = blockIdx.z;
    const int qb0 = blockIdx.x * 128;
    const int ql = lane & 31, hh = lane >> 5;
    const int qrow = qb0 + wave * 32 + ql;                 // this lane's query row
    const int qclamped = qrow < a.Tq ? qrow : a.Tq - 1;

    // Q fragments (B operand of S^T = K.Q^T): Q[q][16s + 8h + j]
    bf16x8_t qf[4];
    {
        const bf16_t* qp = a.Q + (long)b * a.q_bs + (long)qclamped * a.q_rs + head * 64 + hh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16);
    }

    int ntiles = (a.Tk + 63) >> 6;
    if (a.causal) {
        const int last = qb0 + 127 + a.causal_shift;
        const int lim = last < 0 ? 0 : (last >> 6) + 1;
        ntiles = lim < ntiles ? lim : ntiles;
    }

    const bf16_t* kbase = a.K + (long)b * a.k_bs + head * 64;
    const bf16_t* vbase = a.V + (long)b * a.v_bs + head * 64;
    uint4 kreg0, kreg1, vreg0, vreg1;
    unsigned char mbyte = 0;
    const unsigned char* mrow = a.kpm ? a.kpm + (long)b * a.kpm_bs : reinterpret_cast<const unsigned char*>(kbase);       // stand-in: any valid address
    const int srow0 = tid >> 3, srow1 = (256 + tid) >> 3, sc = tid & 7;     // staging slots of this thread
#define ATT_GLOAD(tile)                                                                              \
    do {                                                                                             \
        int key0_ = (tile) * 64 + srow0; key0_ = key0_ < a.Tk ? key0_ : a.Tk - 1;                    \
        int key1_ = (tile) * 64 + srow1; key1_ = key1_ < a.Tk ? key1_ : a.Tk - 1;                    \
        kreg0 = *reinterpret_cast<const uint4*>(kbase + (long)key0_ * a.k_rs + sc * 8);              \
        kreg1 = *reinterpret_cast<const uint4*>(kbase + (long)key1_ * a.k_rs + sc * 8);              \
        vreg0 = *reinterpret_cast<const uint4*>(vbase + (long)key0_ * a.v_rs + sc * 8);              \
        vreg1 = *reinterpret_cast<const uint4*>(vbase + (long)key1_ * a.v_rs + sc * 8);              \
        {   /* key-padding byte of this thread's key of the tile, prefetched WITH the tile. Unconditional (stand-in address without a mask): */ \
            /* guarded, hipcc branches around the load and waits vmcnt(0) behind it -- one exposed L2 round trip per tile of a masked call    */ \
            const int kk_ = (tile) * 64 + (tid & 63);                                                \
            mbyte = mrow[a.kpm ? (kk_ < a.Tk ? kk_ : a.Tk - 1) : 0];                                 \
        }                                                                                            \
    } while (0)

    f32x16_t o[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { o[0][i] = 0.f; o[1][i] = 0.f; }
    float m_run = ATT_NEG, l_run = 0.f;
    const uint32_t drop_key = a.drop_thr16 ? dropout_row_key(*a.drop_seed, a.drop_site, (uint32_t)(b * a.H + head), (uint32_t)(qrow + a.drop_t0)) : 0u;

    if (ntiles > 0) ATT_GLOAD(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();                                   // previous tile fully consumed
        *reinterpret_cast<uint4*>(Ks + srow0 * KS_STRIDE + sc * 8) = kreg0;
        *reinterpret_cast<uint4*>(Ks + srow1 * KS_STRIDE + sc * 8) = kreg1;
        *reinterpret_cast<uint4*>(Vs + srow0 * VS_STRIDE + sc * 8) = vreg0;
        *reinterpret_cast<uint4*>(Vs + srow1 * VS_STRIDE + sc * 8) = vreg1;
        if (tid < 64) {
            const int key = tile * 64 + tid;
            Ms[tid] = key < a.Tk ? ((a.kpm == nullptr || mbyte) ? 2 : 1) : 0;      // 0 = beyond Tk, 1 = masked, 2 = attend
        }
        __syncthreads();
        if (tile + 1 < ntiles) ATT_GLOAD(tile + 1);            // in flight during this tile's math (T14 split)

        // ---- S^T = K . Q^T  (two 32-key sub-tiles)
        f32x16_t st[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) st[kt][i] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Ks + (kt * 32 + ql) * KS_STRIDE + s * 16 + hh * 8);
                st[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], st[kt], 0, 0, 0);
            }
        }
        // ---- scale + mask + online softmax (row = lane pair {l, l^32})
        const int kv0 = tile * 64;
        const bool fast = (a.kpm == nullptr) && !a.causal && (kv0 + 64 <= a.Tk);     // block-uniform: no masking work at all
        float mloc = ATT_NEG;
        unsigned deadmask = 0u;
        float m_new, alpha, psum = 0.f;
        if (fast) {
            // raw scores stay in the accumulators; the softmax scale is folded into the exp2 argument (one FMA per score)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[kt][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            m_new = fmaxf(m_run, mloc * a.scale_log2e);            // scale > 0
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(st[kt][r], a.scale_log2e, -m_new));
                    st[kt][r] = p;
                    psum += p;
                }
        } else {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const unsigned char code = Ms[kl];
                    bool ok = code == 2;
                    if (a.causal) ok = ok && (kv0 + kl <= qrow + a.causal_shift);
                    deadmask |= (code == 0 ? 1u : 0u) << (kt * 16 + r);
                    const float v = ok ? st[kt][r] * a.scale_log2e : ATT_NEG;
                    st[kt][r] = v;
                    mloc = fmaxf(mloc, v);
                }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            m_new = fmaxf(m_run, mloc);
            alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float p = __builtin_amdgcn_exp2f(st[kt][r] - m_new);
                    if ((deadmask >> (kt * 16 + r)) & 1u) p = 0.f;
                    st[kt][r] = p;
                    psum += p;
                }
        }
        m_run = m_new;
        l_run = l_run * alpha + psum;
        if (!__all(alpha == 1.0f)) {                               // the running max settles after a few tiles: skip the O rescale then
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[0][i] *= alpha; o[1][i] *= alpha; }
        }

        if (a.drop_thr16) {                                        // launch-uniform; adjacent registers (r, r+1) hold adjacent keys
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int kl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const uint32_t bits = dropout_pair_bits(drop_key, (uint32_t)(kv0 + kl) >> 1);
                    st[kt][r] = (bits & 0xffffu) >= a.drop_thr16 ? st[kt][r] * a.drop_inv : 0.f;
                    st[kt][r + 1] = (bits >> 16) >= a.drop_thr16 ? st[kt][r + 1] * a.drop_inv : 0.f;
                }
        }
        // ---- O^T += V^T . P^T   (P accumulator -> bf16 B operand; V through transposed LDS reads)
        const int g = lane >> 4, li = lane & 15;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                s16x8_t pv;
#pragma unroll
                for (int j = 0; j < 8; ++j) pv[j] = (short)f2bf(st[kt][8 * s + j]);
                const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv);
                const int kb = kt * 32 + s * 16 + 4 * hh + (li >> 2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int col = dt * 32 + 16 * (g & 1) + 4 * (li & 3);
                    const bf16x8_t vf = tr_pair(Vs + kb * VS_STRIDE + col, Vs + (kb + 8) * VS_STRIDE + col);
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                }
            }
    }

    // ---- finalize
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    {
        const int row0 = qb0 + wave * 32;                          // first query row of this wave
        const int valid = a.Tq - row0 < 32 ? a.Tq - row0 : 32;
        // output rows through the wave's LDS tile: 16 bytes per lane, one full 128-byte head row per 8 lanes (common.h)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 pk;
                pk.x = pack2bf(o[dt][4 * rg + 0] * inv, o[dt][4 * rg + 1] * inv);
                pk.y = pack2bf(o[dt][4 * rg + 2] * inv, o[dt][4 * rg + 3] * inv);
                TILE_PUT(Os[wave], lane, dt, rg, pk);
            }
        tile_rows_store(Os[wave], lane, a.O + (long)b * a.o_bs + (long)row0 * a.o_rs + head * 64, a.o_rs, valid);
    }
    if (qrow < a.Tq) {
        if (a.LSE && hh == 0)
            a.LSE[((long)b * a.H + head) * a.Tq + qrow] = (m_run + log2f(l_tot)) * 0.69314718055994531f;
    }
}


// ---- round 3: 64 query rows per wave, double-buffered K/V tiles, ONE barrier per tile -------------------------------------------------------
// What bounded attn_fwd_kernel above was latency, not a pipe: one 32-row chain per wave (LDS read -> 4 dependent MFMAs -> serial max / exp / sum
// -> pack -> MFMAs), two barriers per 64-key tile and 2 waves per SIMD to hide all of it (224 registers). Here a wave owns TWO 32-row query
// blocks: every K / V fragment read from LDS feeds two MFMAs (LDS bytes per FLOP halve: with head_dim 64 the 32-row form needs as many LDS
// cycles as MFMA cycles), the two softmax chains are independent instruction streams the scheduler interleaves, and tile t+1 is written to the
// other LDS buffer while tile t is being used, so the only barrier of an iteration is the one at its end. The online softmax advances in
// 32-key steps (64 live score registers instead of 128: the kernel fits 2 waves per SIMD without spilling).
constexpr int A2_KB = 64 * KS_STRIDE * 2, A2_VB = 64 * VS_STRIDE * 2, A2_BUF = A2_KB + A2_VB;

// one 32-key step of both query blocks of a wave. MASKED = false: straight-line code (no key mask, no causal mask, all 32 keys exist)
template <bool MASKED, bool DROP>
__device__ __forceinline__ void att2_step(const AttnArgs& a, const bf16_t* Kc, const bf16_t* Vc, const unsigned char* Mc, const int kt, const int kv0,
                                          const bf16x8_t (&qf)[2][4], f32x16_t (&o)[2][2], float (&m_run)[2], float (&l_run)[2],
                                          const uint32_t (&drop_key)[2], const int row0, const int lane) {
    const int ql = lane & 31, hh = lane >> 5, g = lane >> 4, li = lane & 15;
    f32x16_t st[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { st[0][i] = 0.f; st[1][i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {                                   // S^T = K . Q^T: one K fragment read, two MFMAs
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(Kc + (kt * 32 + ql) * KS_STRIDE + s * 16 + hh * 8);
        st[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[0][s], st[0], 0, 0, 0);
        st[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[1][s], st[1], 0, 0, 0);
    }
    float alpha[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        float mloc = ATT_NEG, m_new, psum = 0.f;
        if (!MASKED) {
            // raw scores stay in the accumulators; the softmax scale is folded into the exp2 argument (one FMA per score)
#pragma unroll
            for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[x][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            m_new = fmaxf(m_run[x], mloc * a.scale_log2e);            // scale > 0
            alpha[x] = __builtin_amdgcn_exp2f(m_run[x] - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[x][r], a.scale_log2e, -m_new));
                st[x][r] = p;
                psum += p;
            }
        } else {
            const int qrow = row0 + x * 32 + ql;
            unsigned deadmask = 0u;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const unsigned char code = Mc[kl];
                bool ok = code == 2;
                if (a.causal) ok = ok && (kv0 + kl <= qrow + a.causal_shift);
                deadmask |= (code == 0 ? 1u : 0u) << r;
                const float v = ok ? st[x][r] * a.scale_log2e : ATT_NEG;
                st[x][r] = v;
                mloc = fmaxf(mloc, v);
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
            m_new = fmaxf(m_run[x], mloc);
            alpha[x] = __builtin_amdgcn_exp2f(m_run[x] - m_new);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = __builtin_amdgcn_exp2f(st[x][r] - m_new);
                if ((deadmask >> r) & 1u) p = 0.f;
                st[x][r] = p;
                psum += p;
            }
        }
        m_run[x] = m_new;
        l_run[x] = l_run[x] * alpha[x] + psum;
        if (DROP) {                                                 // adjacent registers (r, r+1) hold adjacent keys
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int kl = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const uint32_t bits = dropout_pair_bits(drop_key[x], (uint32_t)(kv0 + kl) >> 1);
                st[x][r] = (bits & 0xffffu) >= a.drop_thr16 ? st[x][r] * a.drop_inv : 0.f;
                st[x][r + 1] = (bits >> 16) >= a.drop_thr16 ? st[x][r + 1] * a.drop_inv : 0.f;
            }
        }
    }
    if (!__all(alpha[0] == 1.0f && alpha[1] == 1.0f)) {             // the running max settles after a few tiles: skip the O rescale then
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[0][0][i] *= alpha[0]; o[0][1][i] *= alpha[0]; o[1][0][i] *= alpha[1]; o[1][1][i] *= alpha[1]; }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {                                   // O^T += V^T . P^T: one V fragment (two transposed reads), two MFMAs
        s16x8_t pv0, pv1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pv0[j] = (short)f2bf(st[0][8 * s + j]); pv1[j] = (short)f2bf(st[1][8 * s + j]); }
        const bf16x8_t pf0 = __builtin_bit_cast(bf16x8_t, pv0), pf1 = __builtin_bit_cast(bf16x8_t, pv1);
        const int kb = kt * 32 + s * 16 + 4 * hh + (li >> 2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + 16 * (g & 1) + 4 * (li & 3);
            const bf16x8_t vf = tr_pair(Vc + kb * VS_STRIDE + col, Vc + (kb + 8) * VS_STRIDE + col);
            o[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf0, o[0][dt], 0, 0, 0);
            o[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf1, o[1][dt], 0, 0, 0);
        }
    }
}

// MODE 0: no key-padding mask, not causal, no dropout (the CvT stages): full tiles run the straight-line step, a ragged last tile the masked one
// MODE 1: key-padding and / or causal mask;  MODE 2: MODE 1 + dropout on the probabilities (decoder, train mode)
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd2_kernel(const AttnArgs a) {
    CXR_PRIO_MAIN();
    constexpr int NT = NW * 64, CH = 512 / NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * A2_BUF + 128 + 16];
    unsigned char* Ms = smem + 2 * A2_BUF;
    int* Mf = reinterpret_cast<int*>(smem + 2 * A2_BUF + 128);     // per buffer: bit 0 = every key of the tile attends, bit 1 = none does

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, head, b;
    xcd_block_remap(bx, head, b);                                  // the query blocks of one (image, head) run on ONE XCD: its K / V stay in that L2
    const int qb0 = bx * (NW * 64);
    const int ql = lane & 31, hh = lane >> 5;
    const int row0 = qb0 + wave * 64;                              // first query row of this wave
    const bool wave_on = row0 < a.Tq;                              // wave-uniform: a wave without queries only stages tiles and meets the barriers

    bf16x8_t qf[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int qr = row0 + x * 32 + ql;
        const bf16_t* qp = a.Q + (long)b * a.q_bs + (long)(qr < a.Tq ? qr : a.Tq - 1) * a.q_rs + head * 64 + hh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[x][s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16);
    }

    int ntiles = (a.Tk + 63) >> 6;
    if (MODE != 0 && a.causal) {
        const int last = qb0 + NW * 64 - 1 + a.causal_shift;
        const int lim = last < 0 ? 0 : (last >> 6) + 1;
        ntiles = lim < ntiles ? lim : ntiles;
    }
    const int nloop = MODE == 0 ? (a.Tk >> 6) : ntiles;            // MODE 0: the full tiles; the ragged one (if any) follows the loop
    const bf16_t* kbase = a.K + (long)b * a.k_bs + head * 64;
    const bf16_t* vbase = a.V + (long)b * a.v_bs + head * 64;
    const unsigned char* mrow = (MODE != 0 && a.kpm) ? a.kpm + (long)b * a.kpm_bs : reinterpret_cast<const unsigned char*>(kbase);
    uint4 kreg_0, kreg_1, kreg_2, kreg_3, vreg_0, vreg_1, vreg_2, vreg_3;      // (named scalars: as arrays hipcc keeps them in scratch memory)
    unsigned char mbyte = 0;
#define A2_GL1(c, tile)                                                                              \
    if (c < CH) {                                                                                    \
        const int i_ = c * NT + tid;                                                                 \
        int key_ = (tile) * 64 + (i_ >> 3); key_ = key_ < a.Tk ? key_ : a.Tk - 1;                    \
        kreg_##c = *reinterpret_cast<const uint4*>(kbase + (long)key_ * a.k_rs + (i_ & 7) * 8);      \
        vreg_##c = *reinterpret_cast<const uint4*>(vbase + (long)key_ * a.v_rs + (i_ & 7) * 8);      \
    }
#define A2_GLOAD(tile)                                                                               \
    do {                                                                                             \
        A2_GL1(0, tile) A2_GL1(1, tile) A2_GL1(2, tile) A2_GL1(3, tile)                              \
        if (MODE != 0) {   /* key-padding byte of this thread's key, prefetched WITH the tile; unconditional (stand-in address without a mask) */ \
            const int kk_ = (tile) * 64 + (tid & 63);                                                \
            mbyte = mrow[a.kpm ? (kk_ < a.Tk ? kk_ : a.Tk - 1) : 0];                                 \
        }                                                                                            \
    } while (0)
#define A2_LW1(c, ks_, vs_)                                                                          \
    if (c < CH) {                                                                                    \
        const int i_ = c * NT + tid;                                                                 \
        *reinterpret_cast<uint4*>(ks_ + (i_ >> 3) * KS_STRIDE + (i_ & 7) * 8) = kreg_##c;            \
        *reinterpret_cast<uint4*>(vs_ + (i_ >> 3) * VS_STRIDE + (i_ & 7) * 8) = vreg_##c;            \
    }
#define A2_LWRITE(tile, buf)                                                                         \
    do {                                                                                             \
        bf16_t* ks_ = reinterpret_cast<bf16_t*>(smem + (buf) * A2_BUF);                              \
        bf16_t* vs_ = reinterpret_cast<bf16_t*>(smem + (buf) * A2_BUF + A2_KB);                      \
        A2_LW1(0, ks_, vs_) A2_LW1(1, ks_, vs_) A2_LW1(2, ks_, vs_) A2_LW1(3, ks_, vs_)              \
        if (tid < 64) {                                                                              \
            const int key_ = (tile) * 64 + tid;        /* 0 = beyond Tk, 1 = masked, 2 = attend */   \
            const int code_ = key_ < a.Tk ? ((MODE == 0 || a.kpm == nullptr || mbyte) ? 2 : 1) : 0;  \
            Ms[(buf) * 64 + tid] = code_;                                                            \
            if (MODE != 0) {                           /* (tid < 64 is the whole wave 0) */          \
                const unsigned long long live_ = __ballot(code_ == 2);                               \
                if (tid == 0) Mf[buf] = (live_ == ~0ull ? 1 : 0) | (live_ == 0ull ? 2 : 0);          \
            }                                                                                        \
        }                                                                                            \
    } while (0)

    f32x16_t o[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 16; ++i) { o[x][0][i] = 0.f; o[x][1][i] = 0.f; }
    float m_run[2] = {ATT_NEG, ATT_NEG}, l_run[2] = {0.f, 0.f};
    uint32_t drop_key[2] = {0u, 0u};
    if (MODE == 2) {
        drop_key[0] = dropout_row_key(*a.drop_seed, a.drop_site, (uint32_t)(b * a.H + head), (uint32_t)(row0 + ql + a.drop_t0));
        drop_key[1] = dropout_row_key(*a.drop_seed, a.drop_site, (uint32_t)(b * a.H + head), (uint32_t)(row0 + 32 + ql + a.drop_t0));
    }

    if (ntiles > 0) { A2_GLOAD(0); A2_LWRITE(0, 0); }
    if (ntiles > 1) A2_GLOAD(1);
    __syncthreads();
    // MODE != 0, per wave and tile: every key attends and is causally visible to all 64 rows -> the straight-line step; no key attends or
    // none is visible -> no math at all (exact: with key 0 live and visible to every row, m_run is finite after tile 0 and a masked key's
    // probability is exp2(-1e30 - m) = 0); otherwise the masked step
    const bool first_live = MODE != 0 && ntiles > 0 && Ms[0] == 2 && (!a.causal || a.causal_shift >= 0);
    for (int tile = 0; tile < nloop; ++tile) {
        const int cur = tile & 1;
        const bf16_t* Kc = reinterpret_cast<const bf16_t*>(smem + cur * A2_BUF);
        const bf16_t* Vc = reinterpret_cast<const bf16_t*>(smem + cur * A2_BUF + A2_KB);
        int how = wave_on ? (MODE == 0 ? 1 : 2) : 0;               // 0 = skip, 1 = straight-line step, 2 = masked step (wave-uniform)
        if (MODE != 0 && wave_on) {
            const int fl = __builtin_amdgcn_readfirstlane(Mf[cur]);
            const int kv0 = tile * 64;
            if (((fl & 2) && tile > 0 && first_live) || (a.causal && kv0 > row0 + 63 + a.causal_shift)) how = 0;
            else if ((fl & 1) && (!a.causal || kv0 + 63 <= row0 + a.causal_shift)) how = 1;
        }
        if (how == 1) att2_step<false, MODE == 2>(a, Kc, Vc, Ms + cur * 64, 0, tile * 64, qf, o, m_run, l_run, drop_key, row0, lane);
        else if (MODE != 0 && how == 2) att2_step<true, MODE == 2>(a, Kc, Vc, Ms + cur * 64, 0, tile * 64, qf, o, m_run, l_run, drop_key, row0, lane);
        // tile + 1 (in registers since the previous iteration) -> the other buffer; tile + 2 -> registers, in flight until the next iteration
        if (tile + 1 < ntiles) {
            A2_LWRITE(tile + 1, cur ^ 1);
            if (tile + 2 < ntiles) A2_GLOAD(tile + 2);
        }
        if (how == 1) att2_step<false, MODE == 2>(a, Kc, Vc, Ms + cur * 64, 1, tile * 64, qf, o, m_run, l_run, drop_key, row0, lane);
        else if (MODE != 0 && how == 2) att2_step<true, MODE == 2>(a, Kc, Vc, Ms + cur * 64, 1, tile * 64, qf, o, m_run, l_run, drop_key, row0, lane);
        __syncthreads();                                           // tile + 1 is visible; nobody still reads buffer cur
    }
    if (MODE == 0 && nloop < ntiles) {                             // ragged last tile (Tk = 145: 17 keys): keys beyond Tk are dead
        const int cur = nloop & 1;
        const bf16_t* Kc = reinterpret_cast<const bf16_t*>(smem + cur * A2_BUF);
        const bf16_t* Vc = reinterpret_cast<const bf16_t*>(smem + cur * A2_BUF + A2_KB);
        if (wave_on) {
            att2_step<true, false>(a, Kc, Vc, Ms + cur * 64, 0, nloop * 64, qf, o, m_run, l_run, drop_key, row0, lane);
            if (nloop * 64 + 32 < a.Tk) att2_step<true, false>(a, Kc, Vc, Ms + cur * 64, 1, nloop * 64, qf, o, m_run, l_run, drop_key, row0, lane);
        }
        __syncthreads();
    }
#undef A2_GL1
#undef A2_GLOAD
#undef A2_LW1
#undef A2_LWRITE

    // ---- finalize (the K / V buffers are free after the last barrier: the per-wave output tiles live there)
    bf16_t* Ot = reinterpret_cast<bf16_t*>(smem) + wave * (32 * 64);
    if (wave_on) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const float l_tot = l_run[x] + __shfl_xor(l_run[x], 32, 64);
            const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
            const int r0 = row0 + x * 32;
            const int valid = a.Tq - r0 < 32 ? a.Tq - r0 : 32;
            if (valid > 0) {                                       // wave-uniform
                if (x) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        uint2 pk;
                        pk.x = pack2bf(o[x][dt][4 * rg + 0] * inv, o[x][dt][4 * rg + 1] * inv);
                        pk.y = pack2bf(o[x][dt][4 * rg + 2] * inv, o[x][dt][4 * rg + 3] * inv);
                        TILE_PUT(Ot, lane, dt, rg, pk);
                    }
                if (a.O) tile_rows_store(Ot, lane, a.O + (long)b * a.o_bs + (long)r0 * a.o_rs + head * 64, a.o_rs, valid);
                if (a.O8) tile_rows_store_q8(Ot, lane, a.O8 + (long)b * a.o8_bs + (long)r0 * a.o8_rs + head * 64, a.o8_rs, valid, a.o8_inv);
            }
            const int qrow = r0 + ql;
            if (a.LSE && hh == 0 && qrow < a.Tq)
                a.LSE[((long)b * a.H + head) * a.Tq + qrow] = (m_run[x] + log2f(l_tot)) * 0.69314718055994531f;
        }
    }
}

// (Measured and removed in round 4: a K / V-RESIDENT forward for short key sets -- CvT stage 3, 577 queries x 145 keys per (image, head): one workgroup
// per (image, head) stages all keys once and its four waves walk the ten 64-query blocks without another barrier, instead of three workgroups that each
// stage the same three key tiles. Bit-identical (same per-wave step) and SLOWER: 32.2 -> 34.9 us per call, no difference in the step. These calls are
// not bound by the K / V staging but by each wave's serial MFMA -> softmax -> MFMA chain: 384 workgroups of 4 waves put 1.5 waves on a SIMD where
// 1152 put 4.5 rounds of 2.)
template <int NW>
static void attn_fwd2_launch(const AttnArgs& a, hipStream_t stream) {
    const dim3 grid(cdiv(a.Tq, NW * 64), a.H, a.B), block(NW * 64);
    if (a.drop_thr16) { CXR_LAUNCH((attn_fwd2_kernel<NW, 2>), grid, block, 0, stream, a); }
    else if (a.kpm || a.causal) { CXR_LAUNCH((attn_fwd2_kernel<NW, 1>), grid, block, 0, stream, a); }
    else { CXR_LAUNCH((attn_fwd2_kernel<NW, 0>), grid, block, 0, stream, a); }
}

static int env_version(const char* name) { const char* e = getenv(name); return (e && e[0] == '1') ? 1 : 2; }
static int g_attn_fwd_version = env_version("CXR_ATT_FWD");        // lab switches: CXR_ATT_FWD=1 / CXR_ATT_BWD=1 start with the round-2 kernels
int g_attn_bwd_version = env_version("CXR_ATT_BWD");
// A/B switch (tests, micro-benchmarks): forward 1 = attn_fwd_kernel (32 rows per wave), 2 = attn_fwd2_kernel; backward likewise (attention_bwd.hip)
extern "C" int cxr_attn_config(int fwd_version, int bwd_version) {
    if (fwd_version == 1 || fwd_version == 2) g_attn_fwd_version = fwd_version;
    if (bwd_version == 1 || bwd_version == 2) g_attn_bwd_version = bwd_version;
    return CXR_OK;
}

static int attn_fwd_launch(const void* Q, const void* K, const void* V, void* O, float* LSE, const void* kpm,
                           long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                           long kpm_bs, int B, int H, int Tq, int Tk, float scale, int causal, int causal_shift,
                           float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t0,
                           void* O8, long o8_bs, long o8_rs, float o8_inv, hipStream_t stream) {
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed) || (!O && !O8)) return CXR_ERR_ARG;
    if ((q_rs % 8) || (k_rs % 8) || (v_rs % 8) || (q_bs % 8) || (k_bs % 8) || (v_bs % 8)) return CXR_ERR_ARG;
    if (O && ((o_rs % 8) || (o_bs % 8) || (((size_t)O) % 16))) return CXR_ERR_ARG;
    if (O8 && ((o8_rs % 8) || (o8_bs % 8) || (((size_t)O8) % 8) || !(o8_inv > 0.f))) return CXR_ERR_ARG;
    AttnArgs a;
    a.O8 = (unsigned char*)O8; a.o8_bs = o8_bs; a.o8_rs = o8_rs; a.o8_inv = o8_inv;
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O; a.LSE = LSE;
    a.kpm = (const unsigned char*)kpm;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_rs = o_rs;
    a.kpm_bs = kpm_bs; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk;
    a.scale_log2e = scale * 1.4426950408889634f; a.causal = causal; a.causal_shift = causal_shift;
    a.drop_seed = drop_seed; a.drop_site = drop_site; a.drop_thr16 = drop_p > 0.f ? dropout_thr16(drop_p) : 0u;
    a.drop_inv = 1.0f / (1.0f - drop_p); a.drop_t0 = drop_t0;
    if (g_attn_fwd_version >= 2 || O8) {
        static const int force_nw = getenv("CXR_ATT_NW") ? atoi(getenv("CXR_ATT_NW")) : 0;      // lab switch
        if (force_nw ? force_nw == 4 : Tq > 128) attn_fwd2_launch<4>(a, stream);
        else attn_fwd2_launch<2>(a, stream);
    } else {
        CXR_LAUNCH(attn_fwd_kernel, dim3(cdiv(Tq, 128), H, B), dim3(256), 0, stream, a);
    }
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_attn_fwd_bf16(const void* Q, const void* K, const void* V, void* O, float* LSE, const void* kpm,
                                 long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                                 long kpm_bs, int B, int H, int Tq, int Tk, float scale, int causal, int causal_shift,
                                 float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t0, hipStream_t stream) {
    if (!O) return CXR_ERR_ARG;
    return attn_fwd_launch(Q, K, V, O, LSE, kpm, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, kpm_bs, B, H, Tq, Tk, scale, causal, causal_shift,
                           drop_p, drop_seed, drop_site, drop_t0, nullptr, 0, 0, 0.f, stream);
}

// the same attention with the context written as e4m3 (value * inv_scale, saturating; element (b,t,h,d) at O8 + b*o8_bs + t*o8_rs + h*64 + d bytes)
// for a consumer that is an e4m3 GEMM: the bf16-rounded context is what gets quantised, exactly as a separate pass over a bf16 output would
extern "C" int cxr_attn_fwd_q8_bf16(const void* Q, const void* K, const void* V, void* O8, long o8_bs, long o8_rs, float inv_scale, const void* kpm,
                                    long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long kpm_bs, int B, int H, int Tq, int Tk,
                                    float scale, int causal, int causal_shift, hipStream_t stream) {
    return attn_fwd_launch(Q, K, V, nullptr, nullptr, kpm, q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, 0, 0, kpm_bs, B, H, Tq, Tk, scale, causal, causal_shift,
                           0.f, nullptr, 0, 0, O8, o8_bs, o8_rs, inv_scale, stream);
}
