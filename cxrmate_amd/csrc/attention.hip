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

extern "C" int cxr_attn_fwd_bf16(const void* Q, const void* K, const void* V, void* O, float* LSE, const void* kpm,
                                 long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs, long o_bs, long o_rs,
                                 long kpm_bs, int B, int H, int Tq, int Tk, float scale, int causal, int causal_shift,
                                 float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t0, hipStream_t stream) {
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed)) return CXR_ERR_ARG;
    if ((q_rs % 8) || (k_rs % 8) || (v_rs % 8) || (o_rs % 8) || (q_bs % 8) || (k_bs % 8) || (v_bs % 8) || (o_bs % 8) || (((size_t)O) % 16)) return CXR_ERR_ARG;
    AttnArgs a;
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O; a.LSE = LSE;
    a.kpm = (const unsigned char*)kpm;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_rs = o_rs;
    a.kpm_bs = kpm_bs; a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk;
    a.scale_log2e = scale * 1.4426950408889634f; a.causal = causal; a.causal_shift = causal_shift;
    a.drop_seed = drop_seed; a.drop_site = drop_site; a.drop_thr16 = drop_p > 0.f ? dropout_thr16(drop_p) : 0u;
    a.drop_inv = 1.0f / (1.0f - drop_p); a.drop_t0 = drop_t0;
    dim3 grid(cdiv(Tq, 128), H, B);
    CXR_LAUNCH(attn_fwd_kernel, grid, dim3(256), 0, stream, a);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
