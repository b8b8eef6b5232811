// Backward of the fused attention (head_dim 64, bf16, fp32 accumulate): dQ, dK, dV with P recomputed from (Q, K, LSE).
// Two deterministic kernels (no atomics):
//   dkdv: workgroup = 128 keys (32 per wave, key on the MFMA lane) sweeping 64-query tiles; S and dP accumulators are
//         already the B operands of dV^T += dO^T.P and dK^T += Q^T.dS (accumulator-as-operand); dK/dV never leave registers.
//   dq  : workgroup = 128 queries (32 per wave, query on the lane) sweeping 64-key tiles; dQ^T += K^T.dS^T.
// Tiles that are consumed both row-wise (ds_read_b128) and column-wise (ds_read_b64_tr_b16) are kept as two LDS images
// (144-B rows / 192-B rows) so both read patterns are bank-conflict free.
#include "common.h"
#include <stdlib.h>

struct AttnBwdArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; const bf16_t* dO; const bf16_t* O;
    const float* LSE; float* delta;          // delta[B,H,Tq] = rowsum(dO*O): written by the dQ kernel, read by the dK/dV kernel
    bf16_t* dQ; bf16_t* dK; bf16_t* dV;
    const unsigned char* kpm;
    long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs, kpm_bs;
    long dq_bs, dq_rs, dk_bs, dk_rs;         // dQ has Q's logical shape, dK/dV have K's (contiguous outputs)
    int B, H, Tq, Tk;
    float scale, scale_log2e;
    int causal, causal_shift;
    const uint32_t* drop_seed; uint32_t drop_site, drop_thr16; float drop_inv; int drop_t0;      // as in AttnArgs (attention.hip)
};

constexpr int RS = 72;     // row-image stride (bf16 elements)
constexpr int TS = 96;     // transposed-read image stride

__device__ __forceinline__ bf16x8_t trp(const bf16_t* p0, const bf16_t* p1) {
    const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
    const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}

__device__ __forceinline__ bf16x8_t pack_frag(const f32x16_t& x, int s) {
    s16x8_t pv;
#pragma unroll
    for (int j = 0; j < 8; ++j) pv[j] = (short)f2bf(x[8 * s + j]);
    return __builtin_bit_cast(bf16x8_t, pv);
}

// (delta[b,h,q] = sum_d dO[b,q,h,d] * O[b,q,h,d] is produced by attn_bwd_dq_kernel, which runs first)
// NW = waves (32 keys each) per workgroup: 4, or 5 when a 160-key block covers ALL keys of an (image, head) -- CvT stage 3 has 145: as two 128-key
// workgroups the second one staged every Q / dO tile again for 17 keys (one live wave out of four).
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dkdv_kernel(const AttnBwdArgs a) {
    CXR_PRIO_MAIN();
    constexpr int NT = NW * 64;
    __shared__ __attribute__((aligned(16))) bf16_t Qr[64 * RS];
    __shared__ __attribute__((aligned(16))) bf16_t Qt[64 * TS];
    __shared__ __attribute__((aligned(16))) bf16_t Dr[64 * RS];
    __shared__ __attribute__((aligned(16))) bf16_t Dt[64 * TS];
    __shared__ __attribute__((aligned(16))) float Ls[64];
    __shared__ __attribute__((aligned(16))) float Es[64];
    __shared__ __attribute__((aligned(16))) uint32_t Rk[64];          // dropout row keys of the tile's queries
    __shared__ __attribute__((aligned(16))) bf16_t Os[NW][32 * 64];   // per-wave tile for the row-contiguous dK / dV stores

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, head, b;
    xcd_block_remap(bx, head, b);                                  // the key blocks of one (image, head) run on ONE XCD: its Q / dO stream stays in that L2
    const int kb0 = bx * (NW * 32);
    const uint32_t drop_seed = a.drop_thr16 ? *a.drop_seed : 0u;
    const int kl = lane & 31, hh = lane >> 5;
    const int key = kb0 + wave * 32 + kl;
    const int keyc = key < a.Tk ? key : a.Tk - 1;
    int code = 0;
    if (key < a.Tk) code = (a.kpm == nullptr || a.kpm[(long)b * a.kpm_bs + key]) ? 2 : 1;

    bf16x8_t kf[4], vf[4];
    {
        const bf16_t* kp = a.K + (long)b * a.k_bs + (long)keyc * a.k_rs + head * 64 + hh * 8;
        const bf16_t* vp = a.V + (long)b * a.v_bs + (long)keyc * a.v_rs + head * 64 + hh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) { kf[s] = *reinterpret_cast<const bf16x8_t*>(kp + s * 16); vf[s] = *reinterpret_cast<const bf16x8_t*>(vp + s * 16); }
    }

    const int ntiles = (a.Tq + 63) >> 6;
    int tile0 = 0;
    if (a.causal) { const int first = kb0 - a.causal_shift; tile0 = first > 0 ? (first >> 6) : 0; }

    const bf16_t* qbase = a.Q + (long)b * a.q_bs + head * 64;
    const bf16_t* dbase = a.dO + (long)b * a.o_bs + head * 64;
    const float* lbase = a.LSE + ((long)b * a.H + head) * a.Tq;
    const float* ebase = a.delta + ((long)b * a.H + head) * a.Tq;
    // 512 16-byte pieces per 64 x 64 tile: thread t stages piece t and piece NT + t (the second one exists for t < 512 - NT; its load is
    // unconditional on a clamped index, only the LDS write is predicated)
    const int pc1 = NT + tid < 512 ? NT + tid : 511;
    const bool two = NT + tid < 512;
    const int srow0 = tid >> 3, srow1 = pc1 >> 3, sc = tid & 7, sc1 = pc1 & 7;
    uint4 q0, q1, d0, d1;
    float lsev = 0.f;
#define BWD_GLOAD_Q(tile)                                                                                   \
    do {                                                                                                    \
        int r0_ = (tile) * 64 + srow0; r0_ = r0_ < a.Tq ? r0_ : a.Tq - 1;                                   \
        int r1_ = (tile) * 64 + srow1; r1_ = r1_ < a.Tq ? r1_ : a.Tq - 1;                                   \
        q0 = *reinterpret_cast<const uint4*>(qbase + (long)r0_ * a.q_rs + sc * 8);                          \
        q1 = *reinterpret_cast<const uint4*>(qbase + (long)r1_ * a.q_rs + sc1 * 8);                         \
        d0 = *reinterpret_cast<const uint4*>(dbase + (long)r0_ * a.o_rs + sc * 8);                          \
        d1 = *reinterpret_cast<const uint4*>(dbase + (long)r1_ * a.o_rs + sc1 * 8);                         \
        {   /* UNCONDITIONAL load (pointer select + clamped index): a load guarded by a per-lane condition makes hipcc branch around it and  */ \
            /* park s_waitcnt vmcnt(0) behind the branch -- which also waited for the tile prefetch just issued, in every iteration           */ \
            /* the value is only TOUCHED at the top of the next iteration: arithmetic on it here makes the wave wait for the prefetch now */ \
            const int qq_ = (tile) * 64 + (tid & 63);                                                       \
            lsev = (tid < 64 ? lbase : ebase)[qq_ < a.Tq ? qq_ : a.Tq - 1];                                 \
        }                                                                                                   \
    } while (0)

    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] = 0.f; dk[1][i] = 0.f; dv[0][i] = 0.f; dv[1][i] = 0.f; }

    if (tile0 < ntiles) BWD_GLOAD_Q(tile0);
    // the K / V fragments (loaded above, loop-invariant) are "used" here so that their wait sits BEFORE the loop: hipcc's wait-count pass otherwise
    // keeps an s_waitcnt vmcnt(0) in front of their first use INSIDE the loop (the back edge merges to "possibly pending"), where it also waits
    // for the tile prefetch issued a few instructions earlier -- every iteration exposed the whole global-load latency
#pragma unroll
    for (int s = 0; s < 4; ++s) asm volatile("" :: "v"(kf[s]), "v"(vf[s]));
    for (int tile = tile0; tile < ntiles; ++tile) {
        __syncthreads();
        *reinterpret_cast<uint4*>(Qr + srow0 * RS + sc * 8) = q0; *reinterpret_cast<uint4*>(Qt + srow0 * TS + sc * 8) = q0;
        *reinterpret_cast<uint4*>(Dr + srow0 * RS + sc * 8) = d0; *reinterpret_cast<uint4*>(Dt + srow0 * TS + sc * 8) = d0;
        if (NT == 256 || two) {
            *reinterpret_cast<uint4*>(Qr + srow1 * RS + sc1 * 8) = q1; *reinterpret_cast<uint4*>(Qt + srow1 * TS + sc1 * 8) = q1;
            *reinterpret_cast<uint4*>(Dr + srow1 * RS + sc1 * 8) = d1; *reinterpret_cast<uint4*>(Dt + srow1 * TS + sc1 * 8) = d1;
        }
        const bool inq = tile * 64 + (tid & 63) < a.Tq;
        if (tid < 64) Ls[tid] = inq ? lsev * 1.4426950408889634f : INFINITY;
        else if (tid < 128) Es[tid - 64] = inq ? lsev : 0.f;
        else if (a.drop_thr16 && tid < 192)
            Rk[tid - 128] = dropout_row_key(drop_seed, a.drop_site, (uint32_t)(b * a.H + head), (uint32_t)(tile * 64 + tid - 128 + a.drop_t0));
        __syncthreads();
        if (tile + 1 < ntiles) BWD_GLOAD_Q(tile + 1);

        const int g = lane >> 4, li = lane & 15;
        // wave-uniform skips (the staging and the barriers above are still shared): a wave whose 32 keys lie past Tk (Tk = 145: three of the eight
        // waves of a (image, head)) and the second 32-query half of a tile that lies past Tq (Tq = 577: half of the tenth tile) do no math
        if (kb0 + wave * 32 >= a.Tk) continue;
#pragma unroll
        for (int qs = 0; qs < 2; ++qs) {
            if (qs == 1 && tile * 64 + 32 >= a.Tq) continue;
            f32x16_t S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t qa = *reinterpret_cast<const bf16x8_t*>(Qr + (qs * 32 + kl) * RS + s * 16 + hh * 8);
                const bf16x8_t da = *reinterpret_cast<const bf16x8_t*>(Dr + (qs * 32 + kl) * RS + s * 16 + hh * 8);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa, kf[s], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, vf[s], dP, 0, 0, 0);
            }
            // rows of this accumulator: q = 32*qs + 8*(r>>2) + 4*hh + (r&3): four consecutive queries per register quad -> float4 LDS reads
            const bool fast = (a.kpm == nullptr) && !a.causal && (kb0 + NW * 32 <= a.Tk);       // block-uniform
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int qb4 = qs * 32 + 8 * g4 + 4 * hh;
                const float4 L4 = *reinterpret_cast<const float4*>(&Ls[qb4]);
                const float4 E4 = *reinterpret_cast<const float4*>(&Es[qb4]);
                const float Lq[4] = {L4.x, L4.y, L4.z, L4.w}, Eq[4] = {E4.x, E4.y, E4.z, E4.w};
                float fdrop[4] = {1.0f, 1.0f, 1.0f, 1.0f};
                if (a.drop_thr16) {
                    // one hash serves the key pair (key, key^1) = lanes (l, l^1): each lane hashes the queries of its own parity and takes
                    // the others from its neighbour (one DPP exchange instead of a second hash)
                    const int par = key & 1;
#pragma unroll
                    for (int j0 = 0; j0 < 4; j0 += 2) {
                        const uint32_t mine = dropout_pair_bits(Rk[qb4 + j0 + par], (uint32_t)key >> 1);
                        const uint32_t other = (uint32_t)__shfl_xor((int)mine, 1, 64);
                        const uint32_t b0 = par ? other : mine, b1 = par ? mine : other;            // bits of queries j0, j0 + 1
                        fdrop[j0] = (par ? (b0 >> 16) : (b0 & 0xffffu)) >= a.drop_thr16 ? a.drop_inv : 0.f;
                        fdrop[j0 + 1] = (par ? (b1 >> 16) : (b1 & 0xffffu)) >= a.drop_thr16 ? a.drop_inv : 0.f;
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * g4 + j;
                    float p = __builtin_amdgcn_exp2f(fmaf(S[r], a.scale_log2e, -Lq[j]));          // Ls = +inf beyond Tq -> 0
                    if (!fast) {
                        bool ok = code == 2;
                        if (a.causal) ok = ok && (key <= tile * 64 + qb4 + j + a.causal_shift);
                        p = ok ? p : 0.f;
                    }
                    const float f = fdrop[j];
                    S[r] = p * f;                                  // dropped probabilities feed dV
                    dP[r] = p * (f * dP[r] - Eq[j]);               // the softmax scale is applied once to the dK accumulators
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t pf = pack_frag(S, s2), dsf = pack_frag(dP, s2);
                const int qb = qs * 32 + s2 * 16 + 4 * hh + (li >> 2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int col = dt * 32 + 16 * (g & 1) + 4 * (li & 3);
                    const bf16x8_t dot = trp(Dt + qb * TS + col, Dt + (qb + 8) * TS + col);
                    dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dot, pf, dv[dt], 0, 0, 0);
                    const bf16x8_t qt = trp(Qt + qb * TS + col, Qt + (qb + 8) * TS + col);
                    dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qt, dsf, dk[dt], 0, 0, 0);
                }
            }
        }
    }
#undef BWD_GLOAD_Q
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk[0][i] *= a.scale; dk[1][i] *= a.scale; }
    {
        // dK / dV rows of this wave: 32 keys x 64 columns each, written as full 128-byte rows through the wave's LDS tile (common.h)
        const int row0 = kb0 + wave * 32;
        const int valid = a.Tk - row0 < 32 ? a.Tk - row0 : 32;
        bf16_t* kp = a.dK + (long)b * a.dk_bs + (long)row0 * a.dk_rs + head * 64;
        bf16_t* vp = a.dV + (long)b * a.dk_bs + (long)row0 * a.dk_rs + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 pk;
                pk.x = pack2bf(dk[dt][4 * rg], dk[dt][4 * rg + 1]); pk.y = pack2bf(dk[dt][4 * rg + 2], dk[dt][4 * rg + 3]);
                TILE_PUT(Os[wave], lane, dt, rg, pk);
            }
        tile_rows_store(Os[wave], lane, kp, a.dk_rs, valid);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 pk;
                pk.x = pack2bf(dv[dt][4 * rg], dv[dt][4 * rg + 1]); pk.y = pack2bf(dv[dt][4 * rg + 2], dv[dt][4 * rg + 3]);
                TILE_PUT(Os[wave], lane, dt, rg, pk);
            }
        tile_rows_store(Os[wave], lane, vp, a.dk_rs, valid);
    }
}

__global__ __launch_bounds__(256, 3) void attn_bwd_dq_kernel(const AttnBwdArgs a) {
    CXR_PRIO_MAIN();
    __shared__ __attribute__((aligned(16))) bf16_t Kr[64 * RS];
    __shared__ __attribute__((aligned(16))) bf16_t Kt[64 * TS];
    __shared__ __attribute__((aligned(16))) bf16_t Vr[64 * RS];
    __shared__ unsigned char Ms[64];
    __shared__ __attribute__((aligned(16))) bf16_t Os[4][32 * 64];    // per-wave tile for the row-contiguous dQ store

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, head, b;
    xcd_block_remap(bx, head, b);
    const int qb0 = bx * 128;
    const int ql = lane & 31, hh = lane >> 5;
    const int qrow = qb0 + wave * 32 + ql;
    const int qc = qrow < a.Tq ? qrow : a.Tq - 1;

    bf16x8_t qf[4], dof[4];
    {
        const bf16_t* qp = a.Q + (long)b * a.q_bs + (long)qc * a.q_rs + head * 64 + hh * 8;
        const bf16_t* dp = a.dO + (long)b * a.o_bs + (long)qc * a.o_rs + head * 64 + hh * 8;
#pragma unroll
        for (int s = 0; s < 4; ++s) { qf[s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16); dof[s] = *reinterpret_cast<const bf16x8_t*>(dp + s * 16); }
    }
    const float lse2 = a.LSE[((long)b * a.H + head) * a.Tq + qc] * 1.4426950408889634f;
    // delta = rowsum(dO * O) of this lane's query row: the lane pair (hh = 0 / 1) holds the row's 64 dO values between them, O is read the same
    // way. Computed here instead of a separate pass over O and dO (33 launches per training step), and left in a.delta for the dK/dV kernel.
    float dl;
    {
        const bf16_t* op = a.O + (long)b * a.o_bs + (long)qc * a.o_rs + head * 64 + hh * 8;
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float ov[8], dv[8];
            unpack8(*reinterpret_cast<const uint4*>(op + s * 16), ov);
            unpack8(__builtin_bit_cast(uint4, dof[s]), dv);
#pragma unroll
            for (int j = 0; j < 8; ++j) part = fmaf(ov[j], dv[j], part);
        }
        dl = part + __shfl_xor(part, 32, 64);
        if (hh == 0 && qrow < a.Tq) a.delta[((long)b * a.H + head) * a.Tq + qrow] = dl;
    }
    const uint32_t drop_key = a.drop_thr16 ? dropout_row_key(*a.drop_seed, a.drop_site, (uint32_t)(b * a.H + head), (uint32_t)(qrow + a.drop_t0)) : 0u;

    int ntiles = (a.Tk + 63) >> 6;
    if (a.causal) {
        const int last = qb0 + 127 + a.causal_shift;
        const int lim = last < 0 ? 0 : (last >> 6) + 1;
        ntiles = lim < ntiles ? lim : ntiles;
    }
    const bf16_t* kbase = a.K + (long)b * a.k_bs + head * 64;
    const bf16_t* vbase = a.V + (long)b * a.v_bs + head * 64;
    const int srow0 = tid >> 3, srow1 = (256 + tid) >> 3, sc = tid & 7;
    uint4 k0, k1, v0, v1;
    unsigned char mbyte = 0;
    const unsigned char* mrow = a.kpm ? a.kpm + (long)b * a.kpm_bs : reinterpret_cast<const unsigned char*>(kbase);       // stand-in: any valid address
#define BWD_GLOAD_KV(tile)                                                                           \
    do {                                                                                             \
        int r0_ = (tile) * 64 + srow0; r0_ = r0_ < a.Tk ? r0_ : a.Tk - 1;                            \
        int r1_ = (tile) * 64 + srow1; r1_ = r1_ < a.Tk ? r1_ : a.Tk - 1;                            \
        k0 = *reinterpret_cast<const uint4*>(kbase + (long)r0_ * a.k_rs + sc * 8);                   \
        k1 = *reinterpret_cast<const uint4*>(kbase + (long)r1_ * a.k_rs + sc * 8);                   \
        v0 = *reinterpret_cast<const uint4*>(vbase + (long)r0_ * a.v_rs + sc * 8);                   \
        v1 = *reinterpret_cast<const uint4*>(vbase + (long)r1_ * a.v_rs + sc * 8);                   \
        {   /* key-padding byte of this thread's key of the tile, prefetched WITH the tile (unconditional: stand-in address without a mask) */ \
            const int kk_ = (tile) * 64 + (tid & 63);                                                \
            mbyte = mrow[a.kpm ? (kk_ < a.Tk ? kk_ : a.Tk - 1) : 0];                                 \
        }                                                                                            \
    } while (0)

    f32x16_t dq[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] = 0.f; dq[1][i] = 0.f; }

    if (ntiles > 0) BWD_GLOAD_KV(0);
    for (int tile = 0; tile < ntiles; ++tile) {
        __syncthreads();
        *reinterpret_cast<uint4*>(Kr + srow0 * RS + sc * 8) = k0; *reinterpret_cast<uint4*>(Kr + srow1 * RS + sc * 8) = k1;
        *reinterpret_cast<uint4*>(Kt + srow0 * TS + sc * 8) = k0; *reinterpret_cast<uint4*>(Kt + srow1 * TS + sc * 8) = k1;
        *reinterpret_cast<uint4*>(Vr + srow0 * RS + sc * 8) = v0; *reinterpret_cast<uint4*>(Vr + srow1 * RS + sc * 8) = v1;
        if (tid < 64) {
            const int kk = tile * 64 + tid;
            Ms[tid] = kk < a.Tk ? ((a.kpm == nullptr || mbyte) ? 2 : 1) : 0;      // 0 = beyond Tk, 1 = masked, 2 = attend
        }
        __syncthreads();
        if (tile + 1 < ntiles) BWD_GLOAD_KV(tile + 1);

        const int g = lane >> 4, li = lane & 15;
        if (qb0 + wave * 32 >= a.Tq) continue;                     // (wave-uniform: no query of this wave exists; staging and barriers above are shared)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt == 1 && tile * 64 + 32 >= a.Tk) continue;        // (the second 32-key half of the last tile lies past Tk)
            f32x16_t S, dP;
#pragma unroll
            for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t ka = *reinterpret_cast<const bf16x8_t*>(Kr + (kt * 32 + ql) * RS + s * 16 + hh * 8);
                const bf16x8_t va = *reinterpret_cast<const bf16x8_t*>(Vr + (kt * 32 + ql) * RS + s * 16 + hh * 8);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[s], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[s], dP, 0, 0, 0);
            }
            const bool fast = (a.kpm == nullptr) && !a.causal && (tile * 64 + 64 <= a.Tk);
            if (a.drop_thr16) {                                        // dP <- keep/(1-p) * dP before the softmax backward
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int kloc = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    const uint32_t bits = dropout_pair_bits(drop_key, (uint32_t)(tile * 64 + kloc) >> 1);
                    dP[r] = (bits & 0xffffu) >= a.drop_thr16 ? dP[r] * a.drop_inv : 0.f;
                    dP[r + 1] = (bits >> 16) >= a.drop_thr16 ? dP[r + 1] * a.drop_inv : 0.f;
                }
            }
            if (fast) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(S[r], a.scale_log2e, -lse2));
                    dP[r] = p * (dP[r] - dl);                          // softmax scale applied once to the dQ accumulators
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kloc = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    bool ok = Ms[kloc] == 2;
                    if (a.causal) ok = ok && (tile * 64 + kloc <= qrow + a.causal_shift);
                    const float p = ok ? __builtin_amdgcn_exp2f(fmaf(S[r], a.scale_log2e, -lse2)) : 0.f;
                    dP[r] = p * (dP[r] - dl);
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8_t dsf = pack_frag(dP, s2);
                const int kb = kt * 32 + s2 * 16 + 4 * hh + (li >> 2);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const int col = dt * 32 + 16 * (g & 1) + 4 * (li & 3);
                    const bf16x8_t ktf = trp(Kt + kb * TS + col, Kt + (kb + 8) * TS + col);
                    dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[dt], 0, 0, 0);
                }
            }
        }
    }
#undef BWD_GLOAD_KV
#pragma unroll
    for (int i = 0; i < 16; ++i) { dq[0][i] *= a.scale; dq[1][i] *= a.scale; }
    {
        const int row0 = qb0 + wave * 32;
        const int valid = a.Tq - row0 < 32 ? a.Tq - row0 : 32;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 pk;
                pk.x = pack2bf(dq[dt][4 * rg], dq[dt][4 * rg + 1]); pk.y = pack2bf(dq[dt][4 * rg + 2], dq[dt][4 * rg + 3]);
                TILE_PUT(Os[wave], lane, dt, rg, pk);
            }
        tile_rows_store(Os[wave], lane, a.dQ + (long)b * a.dq_bs + (long)row0 * a.dq_rs + head * 64, a.dq_rs, valid);
    }
}

// ---- round 3: the dQ kernel in the structure of attn_fwd2_kernel (attention.hip): 64 query rows per wave (every K / V fragment read from LDS
// feeds two MFMAs, two independent softmax-backward chains per wave), K / V tiles double-buffered with ONE barrier per tile, straight-line
// 32-key step for tiles without masking, no math for tiles whose keys are all masked or causally invisible to the wave (exact: p = 0 there).
constexpr int Q2_KR = 64 * RS * 2, Q2_KT = 64 * TS * 2, Q2_BUF = 2 * Q2_KR + Q2_KT;        // row image of K, transposed-read image of K, row image of V

template <bool MASKED, bool DROP>
__device__ __forceinline__ void dq2_step(const AttnBwdArgs& a, const bf16_t* Kr, const bf16_t* Kt, const bf16_t* Vr, const unsigned char* Mc, const int kt,
                                         const int kv0, const bf16x8_t (&qf)[2][4], const bf16x8_t (&dof)[2][4], f32x16_t (&dq)[2][2],
                                         const float (&lse2)[2], const float (&dl)[2], const uint32_t (&drop_key)[2], const int row0, const int lane) {
    const int ql = lane & 31, hh = lane >> 5, g = lane >> 4, li = lane & 15;
    f32x16_t S[2], dP[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[0][i] = 0.f; S[1][i] = 0.f; dP[0][i] = 0.f; dP[1][i] = 0.f; }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const bf16x8_t ka = *reinterpret_cast<const bf16x8_t*>(Kr + (kt * 32 + ql) * RS + s * 16 + hh * 8);
        const bf16x8_t va = *reinterpret_cast<const bf16x8_t*>(Vr + (kt * 32 + ql) * RS + s * 16 + hh * 8);
        S[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[0][s], S[0], 0, 0, 0);
        S[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka, qf[1][s], S[1], 0, 0, 0);
        dP[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[0][s], dP[0], 0, 0, 0);
        dP[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dof[1][s], dP[1], 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        if (DROP) {                                                 // dP <- keep/(1-p) * dP before the softmax backward
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const int kloc = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                const uint32_t bits = dropout_pair_bits(drop_key[x], (uint32_t)(kv0 + kloc) >> 1);
                dP[x][r] = (bits & 0xffffu) >= a.drop_thr16 ? dP[x][r] * a.drop_inv : 0.f;
                dP[x][r + 1] = (bits >> 16) >= a.drop_thr16 ? dP[x][r + 1] * a.drop_inv : 0.f;
            }
        }
        if (!MASKED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(S[x][r], a.scale_log2e, -lse2[x]));
                dP[x][r] = p * (dP[x][r] - dl[x]);                  // softmax scale applied once to the dQ accumulators
            }
        } else {
            const int qrow = row0 + x * 32 + ql;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kloc = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                bool ok = Mc[kloc] == 2;
                if (a.causal) ok = ok && (kv0 + kloc <= qrow + a.causal_shift);
                const float p = ok ? __builtin_amdgcn_exp2f(fmaf(S[x][r], a.scale_log2e, -lse2[x])) : 0.f;
                dP[x][r] = p * (dP[x][r] - dl[x]);
            }
        }
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const bf16x8_t ds0 = pack_frag(dP[0], s2), ds1 = pack_frag(dP[1], s2);
        const int kb = kt * 32 + s2 * 16 + 4 * hh + (li >> 2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + 16 * (g & 1) + 4 * (li & 3);
            const bf16x8_t ktf = trp(Kt + kb * TS + col, Kt + (kb + 8) * TS + col);
            dq[0][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, ds0, dq[0][dt], 0, 0, 0);
            dq[1][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, ds1, dq[1][dt], 0, 0, 0);
        }
    }
}

// MODE 0: no key-padding mask, not causal, no dropout; 1: masks; 2: masks + dropout (as attn_fwd2_kernel)
template <int NW, int MODE>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq2_kernel(const AttnBwdArgs a) {
    CXR_PRIO_MAIN();
    constexpr int NT = NW * 64, CH = 512 / NT;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * Q2_BUF + 128 + 16];
    unsigned char* Ms = smem + 2 * Q2_BUF;
    int* Mf = reinterpret_cast<int*>(smem + 2 * Q2_BUF + 128);     // per buffer: bit 0 = every key of the tile attends, bit 1 = none does

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, head, b;
    xcd_block_remap(bx, head, b);
    const int qb0 = bx * (NW * 64);
    const int ql = lane & 31, hh = lane >> 5;
    const int row0 = qb0 + wave * 64;
    const bool wave_on = row0 < a.Tq;

    bf16x8_t qf[2][4], dof[2][4];
    float lse2[2], dl[2];
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int qrow = row0 + x * 32 + ql;
        const int qc = qrow < a.Tq ? qrow : a.Tq - 1;
        const bf16_t* qp = a.Q + (long)b * a.q_bs + (long)qc * a.q_rs + head * 64 + hh * 8;
        const bf16_t* dp = a.dO + (long)b * a.o_bs + (long)qc * a.o_rs + head * 64 + hh * 8;
        const bf16_t* op = a.O + (long)b * a.o_bs + (long)qc * a.o_rs + head * 64 + hh * 8;
        uint4 ov[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qf[x][s] = *reinterpret_cast<const bf16x8_t*>(qp + s * 16);
            dof[x][s] = *reinterpret_cast<const bf16x8_t*>(dp + s * 16);
            ov[s] = *reinterpret_cast<const uint4*>(op + s * 16);
        }
        lse2[x] = a.LSE[((long)b * a.H + head) * a.Tq + qc] * 1.4426950408889634f;
        // delta = rowsum(dO * O) of this lane's query row (the lane pair hh = 0 / 1 holds the row between them); left in a.delta for the dK/dV kernel
        float part = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            float of[8], df[8];
            unpack8(ov[s], of);
            unpack8(__builtin_bit_cast(uint4, dof[x][s]), df);
#pragma unroll
            for (int j = 0; j < 8; ++j) part = fmaf(of[j], df[j], part);
        }
        dl[x] = part + __shfl_xor(part, 32, 64);
        if (hh == 0 && qrow < a.Tq) a.delta[((long)b * a.H + head) * a.Tq + qrow] = dl[x];
    }
    uint32_t drop_key[2] = {0u, 0u};
    if (MODE == 2) {
        drop_key[0] = dropout_row_key(*a.drop_seed, a.drop_site, (uint32_t)(b * a.H + head), (uint32_t)(row0 + ql + a.drop_t0));
        drop_key[1] = dropout_row_key(*a.drop_seed, a.drop_site, (uint32_t)(b * a.H + head), (uint32_t)(row0 + 32 + ql + a.drop_t0));
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
#define Q2_GL1(c, tile)                                                                              \
    if (c < CH) {                                                                                    \
        const int i_ = c * NT + tid;                                                                 \
        int key_ = (tile) * 64 + (i_ >> 3); key_ = key_ < a.Tk ? key_ : a.Tk - 1;                    \
        kreg_##c = *reinterpret_cast<const uint4*>(kbase + (long)key_ * a.k_rs + (i_ & 7) * 8);      \
        vreg_##c = *reinterpret_cast<const uint4*>(vbase + (long)key_ * a.v_rs + (i_ & 7) * 8);      \
    }
#define Q2_GLOAD(tile)                                                                               \
    do {                                                                                             \
        Q2_GL1(0, tile) Q2_GL1(1, tile) Q2_GL1(2, tile) Q2_GL1(3, tile)                              \
        if (MODE != 0) {                                                                             \
            const int kk_ = (tile) * 64 + (tid & 63);                                                \
            mbyte = mrow[a.kpm ? (kk_ < a.Tk ? kk_ : a.Tk - 1) : 0];                                 \
        }                                                                                            \
    } while (0)
#define Q2_LW1(c, kr_, kt_, vr_)                                                                     \
    if (c < CH) {                                                                                    \
        const int i_ = c * NT + tid;                                                                 \
        *reinterpret_cast<uint4*>(kr_ + (i_ >> 3) * RS + (i_ & 7) * 8) = kreg_##c;                   \
        *reinterpret_cast<uint4*>(kt_ + (i_ >> 3) * TS + (i_ & 7) * 8) = kreg_##c;                   \
        *reinterpret_cast<uint4*>(vr_ + (i_ >> 3) * RS + (i_ & 7) * 8) = vreg_##c;                   \
    }
#define Q2_LWRITE(tile, buf)                                                                         \
    do {                                                                                             \
        bf16_t* kr_ = reinterpret_cast<bf16_t*>(smem + (buf) * Q2_BUF);                              \
        bf16_t* kt_ = reinterpret_cast<bf16_t*>(smem + (buf) * Q2_BUF + Q2_KR);                      \
        bf16_t* vr_ = reinterpret_cast<bf16_t*>(smem + (buf) * Q2_BUF + Q2_KR + Q2_KT);              \
        Q2_LW1(0, kr_, kt_, vr_) Q2_LW1(1, kr_, kt_, vr_) Q2_LW1(2, kr_, kt_, vr_) Q2_LW1(3, kr_, kt_, vr_) \
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

    f32x16_t dq[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int i = 0; i < 16; ++i) { dq[x][0][i] = 0.f; dq[x][1][i] = 0.f; }

    if (ntiles > 0) { Q2_GLOAD(0); Q2_LWRITE(0, 0); }
    if (ntiles > 1) Q2_GLOAD(1);
    __syncthreads();
    for (int tile = 0; tile < nloop; ++tile) {
        const int cur = tile & 1;
        const bf16_t* Kr = reinterpret_cast<const bf16_t*>(smem + cur * Q2_BUF);
        const bf16_t* Kt = reinterpret_cast<const bf16_t*>(smem + cur * Q2_BUF + Q2_KR);
        const bf16_t* Vr = reinterpret_cast<const bf16_t*>(smem + cur * Q2_BUF + Q2_KR + Q2_KT);
        int how = wave_on ? (MODE == 0 ? 1 : 2) : 0;               // 0 = skip, 1 = straight-line step, 2 = masked step (wave-uniform)
        if (MODE != 0 && wave_on) {
            const int fl = __builtin_amdgcn_readfirstlane(Mf[cur]);
            const int kv0 = tile * 64;
            if ((fl & 2) || (a.causal && kv0 > row0 + 63 + a.causal_shift)) how = 0;
            else if ((fl & 1) && (!a.causal || kv0 + 63 <= row0 + a.causal_shift)) how = 1;
        }
        if (how == 1) dq2_step<false, MODE == 2>(a, Kr, Kt, Vr, Ms + cur * 64, 0, tile * 64, qf, dof, dq, lse2, dl, drop_key, row0, lane);
        else if (MODE != 0 && how == 2) dq2_step<true, MODE == 2>(a, Kr, Kt, Vr, Ms + cur * 64, 0, tile * 64, qf, dof, dq, lse2, dl, drop_key, row0, lane);
        if (tile + 1 < ntiles) {
            Q2_LWRITE(tile + 1, cur ^ 1);
            if (tile + 2 < ntiles) Q2_GLOAD(tile + 2);
        }
        if (how == 1) dq2_step<false, MODE == 2>(a, Kr, Kt, Vr, Ms + cur * 64, 1, tile * 64, qf, dof, dq, lse2, dl, drop_key, row0, lane);
        else if (MODE != 0 && how == 2) dq2_step<true, MODE == 2>(a, Kr, Kt, Vr, Ms + cur * 64, 1, tile * 64, qf, dof, dq, lse2, dl, drop_key, row0, lane);
        __syncthreads();
    }
    if (MODE == 0 && nloop < ntiles) {                             // ragged last tile: keys beyond Tk are dead
        const int cur = nloop & 1;
        const bf16_t* Kr = reinterpret_cast<const bf16_t*>(smem + cur * Q2_BUF);
        const bf16_t* Kt = reinterpret_cast<const bf16_t*>(smem + cur * Q2_BUF + Q2_KR);
        const bf16_t* Vr = reinterpret_cast<const bf16_t*>(smem + cur * Q2_BUF + Q2_KR + Q2_KT);
        if (wave_on) {
            dq2_step<true, false>(a, Kr, Kt, Vr, Ms + cur * 64, 0, nloop * 64, qf, dof, dq, lse2, dl, drop_key, row0, lane);
            if (nloop * 64 + 32 < a.Tk) dq2_step<true, false>(a, Kr, Kt, Vr, Ms + cur * 64, 1, nloop * 64, qf, dof, dq, lse2, dl, drop_key, row0, lane);
        }
        __syncthreads();
    }
#undef Q2_GL1
#undef Q2_GLOAD
#undef Q2_LW1
#undef Q2_LWRITE

    bf16_t* Ot = reinterpret_cast<bf16_t*>(smem) + wave * (32 * 64);
    if (wave_on) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int r0 = row0 + x * 32;
            const int valid = a.Tq - r0 < 32 ? a.Tq - r0 : 32;
            if (valid > 0) {                                       // wave-uniform
                if (x) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        uint2 pk;
                        pk.x = pack2bf(dq[x][dt][4 * rg] * a.scale, dq[x][dt][4 * rg + 1] * a.scale);
                        pk.y = pack2bf(dq[x][dt][4 * rg + 2] * a.scale, dq[x][dt][4 * rg + 3] * a.scale);
                        TILE_PUT(Ot, lane, dt, rg, pk);
                    }
                tile_rows_store(Ot, lane, a.dQ + (long)b * a.dq_bs + (long)r0 * a.dq_rs + head * 64, a.dq_rs, valid);
            }
        }
    }
}

// (Measured and removed in round 4: the dK / dV kernel with 5 / 6 / 8 waves of 32 keys sharing ONE double-buffered Q / dO tile stream -- half the LDS
// staging writes per wave, one barrier per tile instead of two, no nearly empty second workgroup at Tk = 145; 88 KB of LDS = one workgroup per CU.
// Bit-identical and SLOWER on every shape: stage 1 backward 1478 -> 1646 us (dQ + dK/dV), stage 2 364 -> 436, stage 3 94 -> 106, decoder self 59 -> 67,
// cross 216 -> 256; TF step +1.0 ms on the same box. Two independent 4-wave workgroups per CU overlap one another's staging and math phases; eight
// waves behind one barrier stage together and compute together, and the matrix pipe idles through every staging phase.)
// (Measured and removed in round 3: a dK / dV kernel with 64 keys per wave -- every Q / dO fragment feeding two MFMAs, dK / dV of both key blocks in
// 128 accumulator registers, double-buffered tiles, one barrier per tile, ~480 registers = one wave per SIMD. Bit-identical to attn_bwd_dkdv_kernel and
// 8-30 % SLOWER (stage 1 backward 1479 -> 1843 us, stage 2 364 -> 491, stage 3 94 -> 103): with a single wave per SIMD nothing runs under the
// S / dP -> exp -> pack -> dV / dK dependency chain; two waves per SIMD need <= 256 registers, which 64 keys per wave cannot meet.)
// Only the unmasked form is launched: with masks / dropout the kernel needs more than 256 registers (100 - 140 bytes of scratch per lane) and
// measured 4 - 17 % SLOWER than attn_bwd_dq_kernel on the decoder's shapes, 3 - 8 % faster on the CvT stages (scripts/attn_micro.py).
template <int NW>
static void attn_bwd_dq2_launch(const AttnBwdArgs& a, hipStream_t stream) {
    const dim3 grid(cdiv(a.Tq, NW * 64), a.H, a.B), block(NW * 64);
    CXR_LAUNCH((attn_bwd_dq2_kernel<NW, 0>), grid, block, 0, stream, a);
}

extern int g_attn_bwd_version;      // attention.hip (cxr_attn_config)

// dQ [B,Tq,H*64] and dK / dV [B,Tk,H*64] are written contiguously (dq_rs == 0, dkv_rs == 0) or with batch / row strides (the self-attention
// gradients of a layer as the three column blocks of one [B*T, 3*D] matrix; the
// decoder writes the cross-attention dK / dV of all layers into one [B*S, layers*2*D] matrix: one dX and one dW GEMM for all of them).
extern "C" int cxr_attn_bwd_bf16(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* LSE, float* delta,
                                 void* dQ, void* dK, void* dV, const void* kpm, long q_bs, long q_rs, long k_bs, long k_rs, long v_bs, long v_rs,
                                 long o_bs, long o_rs, long kpm_bs, int B, int H, int Tq, int Tk, float scale, int causal, int causal_shift,
                                 float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t0, long dkv_bs, long dkv_rs,
                                 long dq_bs, long dq_rs, hipStream_t stream) {
    if (B <= 0 || H <= 0 || Tq <= 0 || Tk <= 0 || !LSE || !delta || !O) return CXR_ERR_ARG;
    if ((dkv_rs % 8) || (dkv_bs % 8) || ((dkv_rs != 0) != (dkv_bs != 0)) || (dkv_rs && ((((size_t)dK) % 16) || (((size_t)dV) % 16)))) return CXR_ERR_ARG;
    if ((dq_rs % 8) || (dq_bs % 8) || ((dq_rs != 0) != (dq_bs != 0)) || (dq_rs && (((size_t)dQ) % 16))) return CXR_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed)) return CXR_ERR_ARG;
    if ((q_rs % 8) || (k_rs % 8) || (v_rs % 8) || (o_rs % 8) || (q_bs % 8) || (k_bs % 8) || (v_bs % 8) || (o_bs % 8)) return CXR_ERR_ARG;
    AttnBwdArgs a;
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.dO = (const bf16_t*)dO; a.O = (const bf16_t*)O; a.LSE = LSE; a.delta = delta;
    a.dQ = (bf16_t*)dQ; a.dK = (bf16_t*)dK; a.dV = (bf16_t*)dV; a.kpm = (const unsigned char*)kpm;
    a.q_bs = q_bs; a.q_rs = q_rs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.o_rs = o_rs; a.kpm_bs = kpm_bs;
    a.dq_rs = dq_rs ? dq_rs : (long)H * 64; a.dq_bs = dq_rs ? dq_bs : (long)Tq * H * 64; a.dk_rs = dkv_rs ? dkv_rs : (long)H * 64; a.dk_bs = dkv_rs ? dkv_bs : (long)Tk * H * 64;
    a.B = B; a.H = H; a.Tq = Tq; a.Tk = Tk; a.scale = scale; a.scale_log2e = scale * 1.4426950408889634f;
    a.causal = causal; a.causal_shift = causal_shift;
    a.drop_seed = drop_seed; a.drop_site = drop_site; a.drop_thr16 = drop_p > 0.f ? dropout_thr16(drop_p) : 0u;
    a.drop_inv = 1.0f / (1.0f - drop_p); a.drop_t0 = drop_t0;
    // the dQ kernel also writes delta = rowsum(dO * O)
    if (g_attn_bwd_version == 2 && !a.kpm && !a.causal && !a.drop_thr16 && Tq > 1024) {      // (Tq 577: equal within noise)
        attn_bwd_dq2_launch<4>(a, stream);
    } else {
        CXR_LAUNCH(attn_bwd_dq_kernel, dim3(cdiv(Tq, 128), H, B), dim3(256), 0, stream, a);
    }
    // (NW = 5, one 160-key workgroup per (image, head) at Tk = 145: measured SLOWER in round 4, 95 -> 104 us per CvT stage-3 call -- 384 workgroups of 5
    // waves fill the chip worse than 768 of 4, of which every second one retires almost at once; the switch was removed in round 6,
    // profiles/r04_attention_dkdv_variants.txt)
    CXR_LAUNCH(attn_bwd_dkdv_kernel<4>, dim3(cdiv(Tk, 128), H, B), dim3(256), 0, stream, a);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
