// bf16 MFMA GEMM for every dense contraction on the CXRMate hot path.
//
//   C[M,N] = epilogue( alpha * A[M,K] . W[N,K]^T )          ("NT": both operands K-contiguous = nn.Linear layout)
//
// gfx950 design: 128x128 block tile, 4 waves (2x2), each wave 64x64 as 4x4 v_mfma_f32_16x16x32_bf16 tiles;
// operands staged HBM->LDS with 16-byte LDS-DMA (global_load_lds_dwordx4) into a double-buffered, XOR-swizzled
// image (swizzle applied on the per-lane SOURCE address; ds_read_b128 fragment reads are bank-conflict free);
// the MFMA is issued as D^T = W.A^T so that each lane owns 4 CONSECUTIVE output columns -> 8/16-byte epilogue
// accesses for bias / residual / store. 1-D grid with an XCD-aware (bijective) tile remap so the 8 private L2s
// each see a contiguous run of N-tiles sharing one A panel.
#include "common.h"
#include "gemm_args.h"
#include "../../include/cxrmate_hip.h"
#include <stdlib.h>
#include <string.h>


// NST = number of LDS stages. NST == 2: one tile in flight behind the math (vmcnt(0) + __syncthreads per K step).
// NST >= 3 (LDS-DMA only): NST-1 tiles in flight across raw s_barriers with a COUNTED vmcnt, so that the HBM/L2 latency of the
// short K loops of this model (K = 384..768 -> 6..24 steps) is covered by more than one step of MFMA work.
// BN = 128: 128x128 tile (wave tile 64x64). BN = 64: 128x64 tile (wave tile 64x32) for outputs whose 128x128 tiling would leave CUs idle
// or waste half a tile (N = 64 / 192, or fewer than ~1.5 tiles per CU).
__device__ uint4 g_tn_zero_row[16];       // 256 zero bytes: LDS-DMA source for rows that do not exist (token rows beyond R of the weight-gradient GEMM,
                                          // taps outside the image of the implicit-GEMM convolution); a zero-initialised device global

// CONV (round 4): the A operand is NOT a matrix in memory but the im2col view of a token-major activation x [Bn, Hin*Win, Cin] under a ksz x ksz
// convolution (CvT stage-2 / stage-3 patch embeddings, TF5 modeling_cvt.py:77-90: 3 x 3, stride 2, padding 1): row m = output pixel (b, oy, ox),
// column k = (ky, kx, c) -- the K order the embedding weights are re-laid out in (encoder.prepare). With Cin a multiple of 64 a 64-wide k-step is
// ONE tap and 64 consecutive channels, i.e. one contiguous 128-byte piece of x per output row: the LDS-DMA source address is per lane anyway, so
// the gather costs two compares and a select per staged piece and the [M, 9 Cin] im2col matrix (170 / 127 MB per step for stages 2 / 3) is never
// written or read. Taps outside the image read the zero row.
struct ConvA { int Hin, Win, Cin, ksz, Ho, Wo, stride, pad; long x_bs, x_rs; };

template <int BK, bool GLDS, int NST, int BN, bool CONV = false>
__device__ __forceinline__ void gemm_nt_body(const GemmArgs& g, const int bid, const int nwg, const ConvA* cvp = nullptr) {
    static_assert(!CONV || (BK == 64 && GLDS), "the implicit-GEMM A gather is built on the 64-wide LDS-DMA staging");
    CXR_PRIO_MAIN();
    constexpr int BM = 128;
    constexpr int NTL = BN / 32;                // 16-column MFMA tiles per wave along N
    constexpr int CPR = BK / 8;                 // 16-byte chunks per tile row
    constexpr int SLOTS = BM * CPR;             // 16-byte slots of the A tile
    constexpr int PASSES = SLOTS / 256;
    constexpr int WPASSES = BN * CPR / 256;
    constexpr int TILE_BYTES = SLOTS * 16;
    constexpr int LPS = PASSES + WPASSES;       // LDS-DMA instructions per thread per stage
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * 2 * TILE_BYTES];   // [stage][A|W]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;

    const int tiles_n = (g.N + BN - 1) / BN;
    int swz;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tm = swz / tiles_n, tn = swz % tiles_n;

    // per-thread staging sources (row clamp keeps every load in bounds; out-of-range rows are never stored)
    const bf16_t* srcA[PASSES];
    const bf16_t* srcW[WPASSES];
    const bf16_t* srcZ[CONV ? PASSES : 1];
    int civ[CONV ? PASSES : 1], cix[CONV ? PASSES : 1];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) {
        const int s = p * 256 + tid;
        const int row = s / CPR, cp = s % CPR;
        const int c = cp ^ Swz<BK>::f(row);
        int ra = tm * BM + row; ra = ra < g.M ? ra : g.M - 1;
        if constexpr (CONV) {
            const ConvA& cv = *cvp;
            const int hw = cv.Ho * cv.Wo;
            const int b_ = ra / hw, r_ = ra - b_ * hw, oy = r_ / cv.Wo, ox = r_ - oy * cv.Wo;
            civ[p] = oy * cv.stride - cv.pad; cix[p] = ox * cv.stride - cv.pad;
            // (window origin: may lie one row / column outside the image -- only dereferenced for taps inside it)
            srcA[p] = g.A + (long)b_ * cv.x_bs + ((long)civ[p] * cv.Win + cix[p]) * cv.x_rs + c * 8;
            srcZ[p] = reinterpret_cast<const bf16_t*>(g_tn_zero_row) + c * 8;
        } else
        srcA[p] = g.A + (long)ra * g.lda + c * 8;
        if (p < WPASSES) {
            int rw = tn * BN + row; rw = rw < g.N ? rw : g.N - 1;
            srcW[p] = g.W + (long)rw * g.ldw + c * 8;
        }
    }

    auto stage = [&](int buf, int kt) {
        unsigned char* la = lds + (buf * 2 + 0) * TILE_BYTES;
        unsigned char* lw = lds + (buf * 2 + 1) * TILE_BYTES;
        const long k0 = (long)kt * BK;
        int cky = 0, ckx = 0; long coff = 0;                       // CONV: tap and channel block of this k-step (uniform)
        if constexpr (CONV) {
            const ConvA& cv = *cvp;
            const int tap = (int)k0 / cv.Cin, c0 = (int)k0 - tap * cv.Cin;
            cky = tap / cv.ksz; ckx = tap - cky * cv.ksz;
            coff = ((long)cky * cv.Win + ckx) * cv.x_rs + c0;
        }
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if constexpr (GLDS) {
                const int wbase = (p * 256 + wave * 64) * 16;      // wave-uniform; hardware adds lane*16
                const bf16_t* sa;
                if constexpr (CONV) {
                    const bool in = (unsigned)(civ[p] + cky) < (unsigned)cvp->Hin && (unsigned)(cix[p] + ckx) < (unsigned)cvp->Win;
                    sa = in ? srcA[p] + coff : srcZ[p];
                } else sa = srcA[p] + k0;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                                 (__attribute__((address_space(3))) void*)(la + wbase), 16, 0, 0);
                if (p < WPASSES)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[p] + k0),
                                                     (__attribute__((address_space(3))) void*)(lw + wbase), 16, 0, 0);
            } else {
                const uint4 va = *reinterpret_cast<const uint4*>(srcA[p] + k0);
                *reinterpret_cast<uint4*>(la + (p * 256 + tid) * 16) = va;
                if (p < WPASSES) {
                    const uint4 vw = *reinterpret_cast<const uint4*>(srcW[p] + k0);
                    *reinterpret_cast<uint4*>(lw + (p * 256 + tid) * 16) = vw;
                }
            }
        }
    };

    f32x4_t acc[NTL][4];
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int nk = g.K / BK;
    if constexpr (NST == 2) {
        stage(0, 0);
    } else {
#pragma unroll
        for (int s = 0; s < NST - 1; ++s)
            if (s < nk) stage(s, s);
    }
    for (int kt = 0; kt < nk; ++kt) {
        int cur;
        if constexpr (NST == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
            cur = kt & 1;
        } else {
            // tiles kt+1 .. kt+NST-2 may stay in flight; tile kt must have landed (vmcnt counts in issue order)
            const int ahead = nk - 1 - kt;
            if (ahead >= NST - 2) wait_vmcnt<LPS * (NST - 2)>();
            else if (NST > 3 && ahead == 1) wait_vmcnt<LPS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if (kt + NST - 1 < nk) stage((kt + NST - 1) % NST, kt + NST - 1);     // refills the stage consumed in step kt-1
            cur = kt % NST;
        }
        const unsigned char* la = lds + (cur * 2 + 0) * TILE_BYTES;
        const unsigned char* lw = lds + (cur * 2 + 1) * TILE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8_t fa[4], fw[NTL];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ra = wm * 64 + t * 16 + fr;
                const int c = kk * 4 + fq;
                fa[t] = *reinterpret_cast<const bf16x8_t*>(la + (ra * CPR + (c ^ Swz<BK>::f(ra))) * 16);
                if (t < NTL) {
                    const int rw = wn * (BN / 2) + t * 16 + fr;
                    fw[t] = *reinterpret_cast<const bf16x8_t*>(lw + (rw * CPR + (c ^ Swz<BK>::f(rw))) * 16);
                }
            }
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
        }
    }

    const uint32_t dseed = g.drop_thr16 ? *g.drop_seed : 0u;
    // ---- row-contiguous epilogue. In the MFMA layout a lane owns 4 consecutive columns of 16 different rows, so a wave's store instruction
    // touches 16 rows x 32 bytes (bf16): quarter-line writes, and the same for the residual / saved pre-activation streams -- for this model's
    // short K loops (6-12 steps) that was up to 40 % of a GEMM's time. Each wave parks its 64 x (BN/2) fp32 sub-tile (bias already added) in
    // the now idle staging LDS (XOR-swizzled 16-byte slots: conflict-free both ways) and reads it back 8 columns per lane, 8 lanes per row:
    // every global access of the epilogue is 16 bytes per lane and 128-256 contiguous bytes per row.
    constexpr int WCOLS = BN / 2;
    constexpr bool LDS_EPI_FITS = NST * 2 * TILE_BYTES >= 4 * 64 * WCOLS * 4;
    if (LDS_EPI_FITS && g.lds_epilogue) {
        __syncthreads();                                         // every wave has finished reading the last K step's fragments
        float* stg = reinterpret_cast<float*>(lds) + wave * (64 * WCOLS);
        // the lane's bias values: NTL unconditional loads (clamped column) issued together -- per (mt, nt) and guarded by `n0 < N` they were
        // 4 * NTL branch + load + s_waitcnt vmcnt(0) sequences in front of the LDS writes
        float4 bv[NTL];
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int n0 = tn * BN + wn * WCOLS + (nt * 4 + fq) * 4;
            bv[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.bias) bv[nt] = *reinterpret_cast<const float4*>(g.bias + (n0 < g.N ? n0 : g.N - 4));      // (uniform branch; N % 4 == 0)
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTL; ++nt) {
                const int row = mt * 16 + fr, c4 = nt * 4 + fq;
                float4 v = make_float4(acc[nt][mt][0] * g.alpha + bv[nt].x, acc[nt][mt][1] * g.alpha + bv[nt].y, acc[nt][mt][2] * g.alpha + bv[nt].z,
                                       acc[nt][mt][3] * g.alpha + bv[nt].w);
                *reinterpret_cast<float4*>(stg + row * WCOLS + ((c4 ^ (row & (WCOLS / 4 - 1))) << 2)) = v;
            }
        __syncthreads();
        constexpr int CPRW = WCOLS / 8;                          // 8-column chunks per sub-tile row
        if ((g.N & 7) == 0 && !(g.act == 2 && g.residual)) {
            // Every global READ of the epilogue (residual, saved pre-activation, DropPath factor) is issued up front for all CPRW passes,
            // unconditionally (clamped rows / columns), and only the stores are guarded: with the loads inside the per-lane `m < M && n < N`
            // guard hipcc branches around each and waits vmcnt(0) behind it -- CPRW dependent memory round trips per tile.
            uint4 rop[CPRW];                                     // residual OR saved pre-activation (never both on this path: register budget)
            float rsc[CPRW];
            const bool rd_aux = g.act == 2;
            const bf16_t* opp = rd_aux ? g.aux : g.residual;
            const long ldop = rd_aux ? g.ldaux : g.ldr;
#pragma unroll
            for (int it = 0; it < CPRW; ++it) {
                const int c = it * 64 + lane;
                const int row = c / CPRW, cc = c % CPRW;
                int m = tm * BM + wm * 64 + row, n = tn * BN + wn * WCOLS + cc * 8;
                m = m < g.M ? m : g.M - 1; n = n < g.N ? n : g.N - 8;
                if (opp) rop[it] = *reinterpret_cast<const uint4*>(opp + (long)m * ldop + n);
                if (g.row_scale) rsc[it] = g.row_scale[(unsigned)m / (unsigned)g.rs_rows];
            }
#pragma unroll
            for (int it = 0; it < CPRW; ++it) {
                const int c = it * 64 + lane;
                const int row = c / CPRW, cc = c % CPRW;
                const int m = tm * BM + wm * 64 + row;
                const int n = tn * BN + wn * WCOLS + cc * 8;
                const bool ok = m < g.M && n < g.N;
                const int sw = row & (WCOLS / 4 - 1);
                const float4 lo = *reinterpret_cast<const float4*>(stg + row * WCOLS + (((2 * cc) ^ sw) << 2));
                const float4 hi = *reinterpret_cast<const float4*>(stg + row * WCOLS + (((2 * cc + 1) ^ sw) << 2));
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (g.act == 1) {
                    if (g.aux && ok) *reinterpret_cast<uint4*>(g.aux + (long)m * g.ldaux + n) = pack8(v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
                } else if (rd_aux) {
                    float a8[8];
                    unpack8(rop[it], a8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a8[j]);
                }
                if (g.drop_thr16) {
                    const uint32_t dkey = dropout_row_key(dseed, g.drop_site, (uint32_t)(m / g.drop_rows_per_b), (uint32_t)(g.drop_t0 + m % g.drop_rows_per_b));
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const uint32_t bits = dropout_pair_bits(dkey, (uint32_t)(n + j) >> 1);
                        v[j] = (bits & 0xffffu) >= g.drop_thr16 ? v[j] * g.drop_inv : 0.f;
                        v[j + 1] = (bits >> 16) >= g.drop_thr16 ? v[j + 1] * g.drop_inv : 0.f;
                    }
                }
                const float rscale = g.row_scale ? rsc[it] : 1.0f;
                if (g.row_scale && !g.rs_after) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= rscale;
                }
                if (g.residual) {
                    float a8[8];
                    unpack8(rop[it], a8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] += a8[j];
                }
                if (g.row_scale && g.rs_after) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] *= rscale;
                }
                if (g.out_f32) {
                    float* cp = reinterpret_cast<float*>(g.C) + (long)(ok ? m : 0) * g.ldc + (ok ? n : 0);
                    float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                    if (g.accumulate && ok) {
                        const float4 p0 = *reinterpret_cast<const float4*>(cp), p1 = *reinterpret_cast<const float4*>(cp + 4);
                        o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w; o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
                    }
                    if (ok) { *reinterpret_cast<float4*>(cp) = o0; *reinterpret_cast<float4*>(cp + 4) = o1; }
                } else if (ok) {
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = pack8(v);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < CPRW; ++it) {
            const int c = it * 64 + lane;
            const int row = c / CPRW, cc = c % CPRW;
            const int m = tm * BM + wm * 64 + row;
            const int n = tn * BN + wn * WCOLS + cc * 8;
            if (m >= g.M || n >= g.N) continue;
            const bool full = n + 8 <= g.N;                       // N % 4 == 0: otherwise exactly 4 valid columns
            const int sw = row & (WCOLS / 4 - 1);
            const float4 lo = *reinterpret_cast<const float4*>(stg + row * WCOLS + (((2 * cc) ^ sw) << 2));
            const float4 hi = *reinterpret_cast<const float4*>(stg + row * WCOLS + (((2 * cc + 1) ^ sw) << 2));
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (g.act == 1) {
                if (g.aux) {
                    bf16_t* ap = g.aux + (long)m * g.ldaux + n;
                    if (full) *reinterpret_cast<uint4*>(ap) = pack8(v);
                    else { uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); *reinterpret_cast<uint2*>(ap) = pk; }
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = gelu_f(v[j]);
            } else if (g.act == 2) {
                float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const bf16_t* ap = g.aux + (long)m * g.ldaux + n;
                if (full) unpack8(*reinterpret_cast<const uint4*>(ap), a);
                else { const uint2 pk = *reinterpret_cast<const uint2*>(ap); a[0] = __uint_as_float(pk.x << 16); a[1] = __uint_as_float(pk.x & 0xffff0000u);
                       a[2] = __uint_as_float(pk.y << 16); a[3] = __uint_as_float(pk.y & 0xffff0000u); }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a[j]);
            }
            if (g.drop_thr16) {
                const uint32_t dkey = dropout_row_key(dseed, g.drop_site, (uint32_t)(m / g.drop_rows_per_b), (uint32_t)(g.drop_t0 + m % g.drop_rows_per_b));
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const uint32_t bits = dropout_pair_bits(dkey, (uint32_t)(n + j) >> 1);
                    v[j] = (bits & 0xffffu) >= g.drop_thr16 ? v[j] * g.drop_inv : 0.f;
                    v[j + 1] = (bits >> 16) >= g.drop_thr16 ? v[j + 1] * g.drop_inv : 0.f;
                }
            }
            const float rscale = g.row_scale ? g.row_scale[m / g.rs_rows] : 1.0f;
            if (g.row_scale && !g.rs_after) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= rscale;
            }
            if (g.residual) {
                float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                const bf16_t* rp = g.residual + (long)m * g.ldr + n;
                if (full) unpack8(*reinterpret_cast<const uint4*>(rp), a);
                else { const uint2 pk = *reinterpret_cast<const uint2*>(rp); a[0] = __uint_as_float(pk.x << 16); a[1] = __uint_as_float(pk.x & 0xffff0000u);
                       a[2] = __uint_as_float(pk.y << 16); a[3] = __uint_as_float(pk.y & 0xffff0000u); }
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += a[j];
            }
            if (g.row_scale && g.rs_after) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= rscale;
            }
            if (g.out_f32) {
                float* cp = reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n;
                float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                if (g.accumulate) {
                    const float4 p0 = *reinterpret_cast<const float4*>(cp);
                    o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w;
                    if (full) { const float4 p1 = *reinterpret_cast<const float4*>(cp + 4); o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w; }
                }
                *reinterpret_cast<float4*>(cp) = o0;
                if (full) *reinterpret_cast<float4*>(cp + 4) = o1;
            } else {
                bf16_t* cp = reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n;
                if (full) *reinterpret_cast<uint4*>(cp) = pack8(v);
                else { uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]); *reinterpret_cast<uint2*>(cp) = pk; }
            }
        }
        return;
    }
    // epilogue (fallback: BK = 32 tiles or operands without 16-byte alignment): lane owns C[m][n0..n0+3], m = .. + (lane&15), n0 = .. + (lane>>4)*4
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = tm * BM + wm * 64 + mt * 16 + fr;
        if (m >= g.M) continue;
        const uint32_t dkey = g.drop_thr16 ? dropout_row_key(dseed, g.drop_site, (uint32_t)(m / g.drop_rows_per_b),
                                                             (uint32_t)(g.drop_t0 + m % g.drop_rows_per_b)) : 0u;
        const float rscale = g.row_scale ? g.row_scale[m / g.rs_rows] : 1.0f;
#pragma unroll
        for (int nt = 0; nt < NTL; ++nt) {
            const int n0 = tn * BN + wn * (BN / 2) + nt * 16 + fq * 4;
            if (n0 >= g.N) continue;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[nt][mt][r] * g.alpha;
            if (g.bias) {
                const float4 b = *reinterpret_cast<const float4*>(g.bias + n0);
                v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
            }
            if (g.act == 1) {
                if (g.aux) {
                    uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>(g.aux + (long)m * g.ldaux + n0) = pk;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
            } else if (g.act == 2) {
                const uint2 pk = *reinterpret_cast<const uint2*>(g.aux + (long)m * g.ldaux + n0);
                v[0] *= gelu_grad_f(__uint_as_float(pk.x << 16)); v[1] *= gelu_grad_f(__uint_as_float(pk.x & 0xffff0000u));
                v[2] *= gelu_grad_f(__uint_as_float(pk.y << 16)); v[3] *= gelu_grad_f(__uint_as_float(pk.y & 0xffff0000u));
            }
            if (g.drop_thr16) {                                 // n0 is a multiple of 4: two hashes cover the lane's four columns
                const uint32_t b0 = dropout_pair_bits(dkey, (uint32_t)n0 >> 1), b1 = dropout_pair_bits(dkey, ((uint32_t)n0 >> 1) + 1);
                v[0] = (b0 & 0xffffu) >= g.drop_thr16 ? v[0] * g.drop_inv : 0.f; v[1] = (b0 >> 16) >= g.drop_thr16 ? v[1] * g.drop_inv : 0.f;
                v[2] = (b1 & 0xffffu) >= g.drop_thr16 ? v[2] * g.drop_inv : 0.f; v[3] = (b1 >> 16) >= g.drop_thr16 ? v[3] * g.drop_inv : 0.f;
            }
            if (g.row_scale && !g.rs_after) { v[0] *= rscale; v[1] *= rscale; v[2] *= rscale; v[3] *= rscale; }
            if (g.residual) {
                const uint2 pk = *reinterpret_cast<const uint2*>(g.residual + (long)m * g.ldr + n0);
                v[0] += __uint_as_float(pk.x << 16); v[1] += __uint_as_float(pk.x & 0xffff0000u);
                v[2] += __uint_as_float(pk.y << 16); v[3] += __uint_as_float(pk.y & 0xffff0000u);
            }
            if (g.row_scale && g.rs_after) { v[0] *= rscale; v[1] *= rscale; v[2] *= rscale; v[3] *= rscale; }
            if (g.out_f32) {
                float* c = reinterpret_cast<float*>(g.C) + (long)m * g.ldc + n0;
                float4 o = make_float4(v[0], v[1], v[2], v[3]);
                if (g.accumulate) {
                    const float4 old = *reinterpret_cast<const float4*>(c);
                    o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                }
                *reinterpret_cast<float4*>(c) = o;
            } else {
                uint2 pk; pk.x = pack2bf(v[0], v[1]); pk.y = pack2bf(v[2], v[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n0) = pk;
            }
        }
    }
}

template <int BK, bool GLDS, int NST, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const GemmArgs g) {
    gemm_nt_body<BK, GLDS, NST, BN>(g, blockIdx.x, gridDim.x);
}

struct GemmConvArgs { GemmArgs g; ConvA cv; };
template <int BN>
__global__ __launch_bounds__(256) void gemm_nt_conv_kernel(const GemmConvArgs a) {
    gemm_nt_body<64, true, 2, BN, true>(a.g, blockIdx.x, gridDim.x, &a.cv);
}

// Up to three independent problems with the same N, K and epilogue class in ONE launch (workgroups [start[p], start[p+1]) serve problem p):
// the query / key / value projections of a CvT layer have 145..577 rows per image -- the key and value GEMMs alone fill under half of the
// 256 CUs (111 tiles at batch 32) and cost a launch latency each; grouped with the query GEMM they ride in its tail.
struct GemmGroupArgs { GemmArgs g[3]; int start[3]; int n; };
template <int BK, bool GLDS, int NST, int BN>
__global__ __launch_bounds__(256) void gemm_nt_group_kernel(const GemmGroupArgs gg) {
    const int b = blockIdx.x;
    const int p = (gg.n > 2 && b >= gg.start[2]) ? 2 : ((gg.n > 1 && b >= gg.start[1]) ? 1 : 0);
    const int end = p + 1 < gg.n ? gg.start[p + 1] : (int)gridDim.x;
    if (p == 0)      gemm_nt_body<BK, GLDS, NST, BN>(gg.g[0], b, end);
    else if (p == 1) gemm_nt_body<BK, GLDS, NST, BN>(gg.g[1], b - gg.start[1], end - gg.start[1]);
    else             gemm_nt_body<BK, GLDS, NST, BN>(gg.g[2], b - gg.start[2], end - gg.start[2]);
}

static int g_gemm_regstage = 0;     // debugging aid: 1 = stage through registers instead of LDS-DMA
static int g_gemm_exclusive = 1;    // 1: NT GEMM launches may use the persistent kernels (gemm_ws.hip / gemm_pk.hip)

extern "C" int cxr_gemm_set_exclusive(int on) { g_gemm_exclusive = on != 0; return CXR_OK; }

extern "C" int cxr_gemm_set_regstage(int on) { g_gemm_regstage = on; return CXR_OK; }

static int gemm_nt_fill(GemmArgs& g, const cxr_gemm_nt_desc& d) {
    if (d.M <= 0 || d.N <= 0 || d.K <= 0) return CXR_ERR_ARG;
    if (d.drop_p < 0.f || d.drop_p >= 1.f || (d.drop_p > 0.f && (!d.drop_seed || d.drop_rows_per_b <= 0)) || (d.row_scale && d.rs_rows <= 0)) return CXR_ERR_ARG;
    if ((d.K % 32) || (d.N % 4) || (d.lda % 8) || (d.ldw % 8) || (d.ldc % 4)) return CXR_ERR_ARG;
    if (d.residual && (d.ldr % 4)) return CXR_ERR_ARG;
    if (d.act == 2 && !d.aux) return CXR_ERR_ARG;
    if (d.aux && (d.ldaux % 4)) return CXR_ERR_ARG;
    if (d.accumulate && !d.out_f32) return CXR_ERR_ARG;
    g.A = (const bf16_t*)d.A; g.lda = d.lda; g.W = (const bf16_t*)d.W; g.ldw = d.ldw; g.C = d.C; g.ldc = d.ldc;
    g.bias = d.bias; g.residual = (const bf16_t*)d.residual; g.ldr = d.ldr; g.aux = (bf16_t*)d.aux; g.ldaux = d.ldaux;
    g.M = d.M; g.N = d.N; g.K = d.K; g.alpha = d.alpha; g.act = d.act; g.out_f32 = d.out_f32; g.accumulate = d.accumulate;
    g.drop_seed = d.drop_seed; g.drop_site = d.drop_site; g.drop_thr16 = d.drop_p > 0.f ? dropout_thr16(d.drop_p) : 0u; g.drop_inv = 1.0f / (1.0f - d.drop_p);
    g.drop_rows_per_b = d.drop_rows_per_b > 0 ? d.drop_rows_per_b : 1; g.drop_t0 = d.drop_t0;
    g.row_scale = d.row_scale; g.rs_rows = d.rs_rows > 0 ? d.rs_rows : 1; g.rs_after = d.rs_after;
    static int lds_epi = -1;          // CXR_GEMM_LDS_EPILOGUE=0 falls back to the per-lane 8-byte epilogue (A/B aid)
    if (lds_epi < 0) { const char* e = getenv("CXR_GEMM_LDS_EPILOGUE"); lds_epi = e ? atoi(e) : 1; }
    auto al16 = [](const void* p, long ld, int esz) { return p == nullptr || ((((size_t)p) % 16) == 0 && ((ld * esz) % 16) == 0); };
    g.lds_epilogue = lds_epi && (d.N % 4) == 0 && al16(d.C, d.ldc, d.out_f32 ? 4 : 2) && al16(d.residual, d.ldr, 2) && al16(d.aux, d.ldaux, 2) &&
                     (d.bias == nullptr || (((size_t)d.bias) % 16) == 0);
    return CXR_OK;
}

// tile shape: narrow (128x64) when the last N tile would be mostly empty (N = 64, 192) or when 128x128 tiling gives at most one tile per CU
static bool gemm_nt_narrow(int N, long tiles128) {
    static int force_bn = -1;         // tuning aid: CXR_GEMM_BN=64|128
    if (force_bn < 0) { const char* e = getenv("CXR_GEMM_BN"); force_bn = e ? atoi(e) : 0; }
    bool bn64 = (N % 128) != 0 && (N % 128) <= 64 && N < 1024;      // (a ragged last tile of a wide N -- the 30000-column LM head -- is 1 tile in 235)
    if (tiles128 <= 256) bn64 = true;           // at most one 128-wide tile per CU: twice as many half-width workgroups finish ~15 % sooner (scripts/gemm_bn_threshold.py)
    if (force_bn == 64) bn64 = true; else if (force_bn == 128) bn64 = false;
    return bn64;
}

extern "C" int cxr_gemm_nt_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                                const float* bias, const void* residual, long ldr, void* aux, long ldaux,
                                int M, int N, int K, float alpha, int act, int out_f32, int accumulate,
                                float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_rows_per_b, int drop_t0,
                                const float* row_scale, int rs_rows, int rs_after, hipStream_t stream) {
    const cxr_gemm_nt_desc d{A, lda, W, ldw, C, ldc, bias, residual, ldr, aux, ldaux, M, N, K, alpha, act, out_f32, accumulate,
                             drop_p, drop_seed, drop_site, drop_rows_per_b, drop_t0, row_scale, rs_rows, rs_after};
    GemmArgs g;
    const int rc = gemm_nt_fill(g, d);
    if (rc) return rc;
    // Persistent one-workgroup-per-CU kernels (144 KB of LDS each) for the shapes they win on -- only while the launching stream has the GPU to
    // itself (cxr_gemm_set_exclusive): beside the weight-gradient stream's kernels their workgroups cannot be co-resident and the static work
    // partition waits for the last CU to free up.
    if (gemm_strip_launch(g, stream)) { CXR_LAUNCH_CHECK(); return CXR_OK; }   // M x 384 x K: one strip of rows x all 384 columns per workgroup (gemm_strip.hip)
    if (g_gemm_exclusive) {
        if (gemm_ws_launch(g, stream)) { CXR_LAUNCH_CHECK(); return CXR_OK; }   // K = 384, N >= 768: W resident in registers (gemm_ws.hip)
        if (gemm_pk_launch(g, stream)) { CXR_LAUNCH_CHECK(); return CXR_OK; }   // tall / very wide problems: persistent 256-row tiles (gemm_pk.hip)
    }
    static int force_bk = -1, stages = -1;     // tuning aids: CXR_GEMM_BK=32|64, CXR_GEMM_STAGES=2|3|4
    if (force_bk < 0) { const char* e = getenv("CXR_GEMM_BK"); force_bk = e ? atoi(e) : 0; }
    if (stages < 0) { const char* e = getenv("CXR_GEMM_STAGES"); stages = e ? atoi(e) : 2; }
    const bool bk64 = (K % 64) == 0 && force_bk != 32;
    const bool bn64 = gemm_nt_narrow(N, (long)cdiv(M, 128) * cdiv(N, 128));
    const int grid = cdiv(M, 128) * cdiv(N, bn64 ? 64 : 128);
    if (g_gemm_regstage) {
        if (bk64) CXR_LAUNCH((gemm_nt_kernel<64, false, 2, 128>), dim3(cdiv(M, 128) * cdiv(N, 128)), dim3(256), 0, stream, g);
        else      CXR_LAUNCH((gemm_nt_kernel<32, false, 2, 128>), dim3(cdiv(M, 128) * cdiv(N, 128)), dim3(256), 0, stream, g);
    } else if (stages >= 4 && force_bk != 64) {
        CXR_LAUNCH((gemm_nt_kernel<32, true, 4, 128>), dim3(cdiv(M, 128) * cdiv(N, 128)), dim3(256), 0, stream, g);
    } else if (stages == 3 && force_bk != 64) {
        CXR_LAUNCH((gemm_nt_kernel<32, true, 3, 128>), dim3(cdiv(M, 128) * cdiv(N, 128)), dim3(256), 0, stream, g);
    } else if (bn64) {
        if (bk64) CXR_LAUNCH((gemm_nt_kernel<64, true, 2, 64>), dim3(grid), dim3(256), 0, stream, g);
        else      CXR_LAUNCH((gemm_nt_kernel<32, true, 2, 64>), dim3(grid), dim3(256), 0, stream, g);
    } else {
        if (bk64) CXR_LAUNCH((gemm_nt_kernel<64, true, 2, 128>), dim3(grid), dim3(256), 0, stream, g);
        else      CXR_LAUNCH((gemm_nt_kernel<32, true, 2, 128>), dim3(grid), dim3(256), 0, stream, g);
    }
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// C[Bn*Ho*Wo, N] = conv_ksz,stride,pad(x) as an implicit GEMM + bias (see ConvA). x [Bn, Hin*Win, Cin] token-major bf16, W [N, ksz*ksz*Cin] in
// (ky, kx, c) order. Replaces im2col + GEMM for nn.Conv2d(Cin, N, 3, 2, 1) of the CvT stage embeddings (TF5 modeling_cvt.py:77-90).
extern "C" int cxr_gemm_nt_conv_bf16(const void* x, long x_bs, long x_rs, int Bn, int Hin, int Win, int Cin, int ksz, int stride, int pad,
                                     const void* W, long ldw, void* C, long ldc, const float* bias, int N, hipStream_t stream) {
    if (!x || !W || !C || Bn <= 0 || Hin <= 0 || Win <= 0 || Cin <= 0 || (Cin % 64) || ksz < 1 || stride < 1 || pad < 0 || (x_bs % 8) || (x_rs % 8))
        return CXR_ERR_ARG;
    const int Ho = (Hin + 2 * pad - ksz) / stride + 1, Wo = (Win + 2 * pad - ksz) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return CXR_ERR_ARG;
    const long M = (long)Bn * Ho * Wo;
    if (M > 0x7fffffffL) return CXR_ERR_ARG;
    const cxr_gemm_nt_desc d{x, 8, W, ldw, C, ldc, bias, nullptr, 0, nullptr, 0, (int)M, N, ksz * ksz * Cin, 1.0f, 0, 0, 0, 0.f, nullptr, 0u, 1, 0, nullptr, 1, 0};
    GemmConvArgs a;
    const int rc = gemm_nt_fill(a.g, d);
    if (rc) return rc;
    a.cv = ConvA{Hin, Win, Cin, ksz, Ho, Wo, stride, pad, x_bs, x_rs};
    const bool bn64 = gemm_nt_narrow(N, (long)cdiv((int)M, 128) * cdiv(N, 128));
    if (bn64) CXR_LAUNCH((gemm_nt_conv_kernel<64>), dim3(cdiv((int)M, 128) * cdiv(N, 64)), dim3(256), 0, stream, a);
    else      CXR_LAUNCH((gemm_nt_conv_kernel<128>), dim3(cdiv((int)M, 128) * cdiv(N, 128)), dim3(256), 0, stream, a);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// n <= 3 problems with equal N and K in one launch (see gemm_nt_group_kernel); epilogues may differ per problem
extern "C" int cxr_gemm_nt_group_bf16(const cxr_gemm_nt_desc* d, int n, hipStream_t stream) {
    if (!d || n < 1 || n > 3) return CXR_ERR_ARG;
    GemmGroupArgs gg;
    long tiles128 = 0;
    int m = 0;                                                     // problems that stay in the grouped launch
    for (int i = 0; i < n; ++i) {
        if (d[i].N != d[0].N || d[i].K != d[0].K) return CXR_ERR_ARG;
        const int rc = gemm_nt_fill(gg.g[m], d[i]);
        if (rc) return rc;
        // (round 6, measured and not kept: handing the 36928-row member of a q / k / v group to the row-strip kernel as its own launch takes back 0.7 of the
        //  0.8 ms the strip kernel gains elsewhere -- the two 9280-row members then run alone on a third of the chip; scripts/r6/call17.sh)
        tiles128 += (long)cdiv(d[i].M, 128) * cdiv(d[i].N, 128);
        ++m;
    }
    if (m == 0) { CXR_LAUNCH_CHECK(); return CXR_OK; }
    n = m;
    if (gemm_strip_group_launch(gg.g, n, stream)) { CXR_LAUNCH_CHECK(); return CXR_OK; }       // round 6: all members as strips of ONE launch (gemm_strip.hip)
    for (int i = n; i < 3; ++i) gg.g[i] = gg.g[0];
    const int N = gg.g[0].N, K = gg.g[0].K;
    const bool bk64 = (K % 64) == 0;
    const bool bn64 = gemm_nt_narrow(N, tiles128);
    int grid = 0;
    for (int i = 0; i < 3; ++i) {
        gg.start[i] = grid;
        if (i < n) grid += cdiv(gg.g[i].M, 128) * cdiv(N, bn64 ? 64 : 128);
    }
    gg.n = n;
    if (bn64) {
        if (bk64) CXR_LAUNCH((gemm_nt_group_kernel<64, true, 2, 64>), dim3(grid), dim3(256), 0, stream, gg);
        else      CXR_LAUNCH((gemm_nt_group_kernel<32, true, 2, 64>), dim3(grid), dim3(256), 0, stream, gg);
    } else {
        if (bk64) CXR_LAUNCH((gemm_nt_group_kernel<64, true, 2, 128>), dim3(grid), dim3(256), 0, stream, gg);
        else      CXR_LAUNCH((gemm_nt_group_kernel<32, true, 2, 128>), dim3(grid), dim3(256), 0, stream, gg);
    }
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM ("TN", split over the token dimension):
//     C[I,J] += alpha * sum_r P[r,I] * Q[r,J]        (dW[N,K] += dY[M,N]^T . X[M,K]),   dbias[I] += sum_r P[r,I]
// Both operands are read in their natural row-major (token-major) layout: a 32-row x 128-col tile of each is register-staged
// into LDS (rows padded to 288 B and permuted so that ds_read_b64_tr_b16 is bank-conflict free) and the MFMA fragments
// (8 consecutive tokens of one column) come from the hardware transposing read. The reduction dimension (tokens, 8k..295k)
// is split across workgroups so that the small I x J outputs (64x64 .. 1536x384) still fill 256 CUs; partial products are
// combined with fp32 atomics into the (already fp32, already accumulating) gradient buffer.
struct GemmTnArgs {
    const bf16_t* P; long ldp;
    const bf16_t* Q; long ldq;
    float* C; long ldc;
    float* dbias;
    int R, I, J;
    int splits, tiles_j, rt_per_split;
    float alpha;
    float* ws;          // deterministic split-K: split s leaves its partial tile in ws[s][I_pad][J_pad] (plain stores) and partial bias sums in
    float* wsb;         // wsb[s][I_pad]; gemm_tn_reduce_kernel adds them to C / dbias in split order. NULL: fp32 atomics straight into C / dbias
    int mode;           // 0 atomics, 1 one split: this workgroup owns its tile -> plain read-modify-write of C, 2 partials to ws
    int debug;          // tuning aid (CXR_TN_DEBUG): 1 skip epilogue atomics, 2 skip MFMA, 4 skip LDS-DMA refills, 8 skip fragment reads
};

__device__ __forceinline__ int tn_gsw(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

// MFMA fragment (8 consecutive tokens of one column) through the transposing LDS read. Tile image: [32 token rows][256 B], 16-byte chunk
// index XOR-swizzled by (gsw(row) << 1): the 8 rows one half-wave reads together ({r..r+3} and {r+8..r+11}) land on 8 distinct 32-byte
// bank groups -> conflict-free, while each row stays inside its own 256 B so the image can be filled by lane-linear LDS-DMA.
// The reads are issued as inline asm: in front of the __builtin_amdgcn_ds_read_tr16_b64 builtin hipcc (ROCm 7.2) emits s_waitcnt vmcnt(0)
// whenever an LDS-DMA is outstanding, which would drain the 3-tile-deep pipeline every step. The caller waits lgkmcnt(0) once after
// issuing all 16 reads of a tile and fences the MFMAs behind it with sched_barrier (cdna_hip_programming.md 5.4 rule 18).
__device__ __forceinline__ void tn_frag_issue(const unsigned char* tile, int colbase, int lane, s16x4_t& lo, s16x4_t& hi) {
    const int g4 = lane >> 4, i = lane & 15;
    const int q = i >> 2, p = i & 3;
    const int r0 = 8 * g4 + q, r1 = r0 + 4;
    const int ch = (colbase >> 3) + (p >> 1);                  // 16-byte chunk index of this lane's 4 columns
    const unsigned a0 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(tile + r0 * 256 + ((ch ^ (tn_gsw(r0) << 1)) << 4) + (p & 1) * 8);
    const unsigned a1 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(tile + r1 * 256 + ((ch ^ (tn_gsw(r1) << 1)) << 4) + (p & 1) * 8);
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(a1));
}
__device__ __forceinline__ bf16x8_t tn_frag_join(const s16x4_t& lo, const s16x4_t& hi) {
    s16x8_t v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int NST>
__global__ __launch_bounds__(256) void gemm_tn_kernel(const GemmTnArgs g) {
    constexpr int BR = 32, TILE = BR * 256, LPS = 4;               // 4 LDS-DMA instructions per thread per stage (2 per operand)
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * 2 * TILE];      // [stage][P|Q]: 64 KB at 4 stages, 32 KB at 2
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 1, wj = wave >> 1;
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (bid % 8), each with a private 4 MB L2. All output tiles of one
    // token split read the SAME rows of P and Q, so the split-major work list is cut into 8 contiguous runs, one per XCD (bijective
    // remap): a split's tiles run back-to-back on one XCD and its ~1-3 MB row chunk is fetched from HBM once instead of once per
    // tile (the kernel is otherwise bound by those re-reads: 64 FLOP per byte at a 128x128x32 step).
    int swz;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int tiles = gridDim.x / g.splits;
    const int split = swz / tiles, tile = swz % tiles;
    const int ti = tile / g.tiles_j, tj = tile % g.tiles_j;
    const int nrt = (g.R + BR - 1) / BR;
    const int rt0 = split * g.rt_per_split;
    const int rt1 = min(nrt, rt0 + g.rt_per_split);
    const int nt = rt1 - rt0;
    if (nt <= 0) return;

    // staging slots: slot s = pass*256 + tid -> row = s >> 4, physical chunk = s & 15, logical chunk = physical ^ (gsw(row) << 1)
    int srow[2], scol_p[2], scol_q[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int sl = p * 256 + tid;
        srow[p] = sl >> 4;
        const int c = (sl & 15) ^ (tn_gsw(srow[p]) << 1);
        int cp = ti * 128 + c * 8; if (cp >= g.I) cp = 0;
        int cq = tj * 128 + c * 8; if (cq >= g.J) cq = 0;
        scol_p[p] = cp; scol_q[p] = cq;
    }
    const bf16_t* zrow = reinterpret_cast<const bf16_t*>(g_tn_zero_row);
    auto stage = [&](int st, int rt) {
        unsigned char* lp = lds + (st * 2 + 0) * TILE;
        unsigned char* lq = lp + TILE;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const long r = (long)rt * BR + srow[p];
            const bool ok = r < g.R;
            const bf16_t* sp = ok ? g.P + r * g.ldp + scol_p[p] : zrow + ((p * 256 + tid) & 15) * 8;
            const bf16_t* sq = ok ? g.Q + r * g.ldq + scol_q[p] : zrow + ((p * 256 + tid) & 15) * 8;
            const int wbase = (p * 256 + wave * 64) * 16;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                             (__attribute__((address_space(3))) void*)(lp + wbase), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq,
                                             (__attribute__((address_space(3))) void*)(lq + wbase), 16, 0, 0);
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of P = P^T . 1: four extra MFMAs against an all-ones B fragment in the tj == 0 workgroups
    // (an ordinary ds_read of the staging array here would make hipcc drain the LDS-DMA pipeline with vmcnt(0))
    const bool do_bias = g.dbias != nullptr && tj == 0 && wj == 0;
    f32x4_t accb[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) accb[a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    s16x8_t ones_v;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones_v[j] = (short)0x3F80;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_v);

#pragma unroll
    for (int s2 = 0; s2 < NST - 1; ++s2)
        if (s2 < nt) stage(s2, rt0 + s2);
    for (int kt = 0; kt < nt; ++kt) {
        const int ahead = nt - 1 - kt;
        if (ahead >= NST - 2) wait_vmcnt<LPS * (NST - 2)>();
        else if (ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nt && !(g.debug & 4)) stage((kt + NST - 1) % NST, rt0 + kt + NST - 1);
        const unsigned char* tp = lds + ((kt % NST) * 2 + 0) * TILE;
        const unsigned char* tq = tp + TILE;
        s16x4_t alo[4], ahi[4], blo[4], bhi[4];
        if (!(g.debug & 8) || kt == 0) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                tn_frag_issue(tp, wi * 64 + t * 16, lane, alo[t], ahi[t]);
                tn_frag_issue(tq, wj * 64 + t * 16, lane, blo[t], bhi[t]);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        bf16x8_t fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { fa[t] = tn_frag_join(alo[t], ahi[t]); fb[t] = tn_frag_join(blo[t], bhi[t]); }
        if (do_bias) {
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2) accb[i2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i2], ones, accb[i2], 0, 0, 0);
        }
        if (!(g.debug & 2)) {
#pragma unroll
            for (int i2 = 0; i2 < 4; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < 4; ++j2)
                    acc[i2][j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i2], fb[j2], acc[i2][j2], 0, 0, 0);
        } else {
            acc[0][0][0] += (float)fa[0][0] + (float)fb[3][7];
        }
    }
    if (g.debug & 1) { if (acc[0][0][0] == 123.456f) g.C[0] = 1.f; return; }
    // epilogue: each wave parks its 64x64 fp32 sub-tile in its own 16 KB of the (now idle) staging LDS, then adds it to the gradient
    // buffer one ROW per wave-instruction: 64 lanes x 4 B = 256 contiguous bytes, the shape at which global float atomics run at full rate
    const int fr = lane & 15, fq = lane >> 4;
    __syncthreads();                                           // every wave is done reading the staging tiles
    // the staging LDS holds HALVES x (rows per pass) of every wave's 64-row sub-tile: all 64 rows at 4 stages (64 KB), 32 rows per pass at 2
    constexpr int PASS_ROWS = (NST * 2 * TILE) / (4 * 64 * 4) >= 64 ? 64 : 32;
    float* wtile = reinterpret_cast<float*>(lds) + wave * (PASS_ROWS * 64);
#pragma unroll
    for (int half = 0; half < 64 / PASS_ROWS; ++half) {
#pragma unroll
        for (int it = 0; it < PASS_ROWS / 16; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    wtile[(it * 16 + fq * 4 + r) * 64 + jt * 16 + fr] = acc[half * (PASS_ROWS / 16) + it][jt][r] * g.alpha;
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0): own writes landed (each wave reads back only its own tile)
        const int j = tj * 128 + wj * 64 + lane;
        const int ibase = ti * 128 + wi * 64 + half * PASS_ROWS;
        if (g.mode == 2) {                                         // partial tile of this split: whole 256-B rows, plain stores
            const int Jp = g.tiles_j * 128, Ip = (tiles / g.tiles_j) * 128;
            float* wrow = g.ws + ((long)split * Ip + ibase) * Jp + tj * 128 + wj * 64 + lane;
#pragma unroll 8
            for (int row = 0; row < PASS_ROWS; ++row) wrow[(long)row * Jp] = wtile[row * 64 + lane];
        } else if (j < g.J) {
            if (g.mode == 1) {                                     // the only split: this workgroup owns the tile
#pragma unroll 8
                for (int row = 0; row < PASS_ROWS; ++row) {
                    const int i = ibase + row;
                    if (i < g.I) g.C[(long)i * g.ldc + j] += wtile[row * 64 + lane];
                }
            } else {
#pragma unroll 8
                for (int row = 0; row < PASS_ROWS; ++row) {
                    const int i = ibase + row;
                    if (i < g.I) atomicAdd(g.C + (long)i * g.ldc + j, wtile[row * 64 + lane]);
                }
            }
        }
        if (half + 1 < 64 / PASS_ROWS) __builtin_amdgcn_s_waitcnt(0xC07F);          // the next pass overwrites the rows just read
    }
    if (do_bias && fr == 0) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = ti * 128 + wi * 64 + it * 16 + fq * 4 + r;
                if (g.mode == 2) g.wsb[(long)split * ((tiles / g.tiles_j) * 128) + i] = accb[it][r];
                else if (i < g.I) {
                    if (g.mode == 1) g.dbias[i] += accb[it][r];
                    else atomicAdd(g.dbias + i, accb[it][r]);
                }
            }
    }
}

// ---- round 3: 256 x 256 output blocks (weight gradients with both dimensions > 128) -------------------------------------------------------------
// gemm_tn_kernel moves 16 KB through LDS per 1 MFLOP step (64 FLOP / byte) and keeps at most 48 KB in flight per CU against ~2 us of loaded
// memory latency: its step takes ~2000 cycles for 256 cycles of MFMA work (8 % of peak, in-step profile). Here a workgroup owns up to 256 x 256
// outputs (eight waves x 128 x 64: 128 accumulator registers per lane, two waves per SIMD; four waves x 128 x 128 need all 256 accumulator
// registers and hipcc then spills accumulators around the loop's back edge), i.e. 128 FLOP per staged byte and half as many partial tiles per launch; a step stages (NI + NJ) sub-images of [32 tokens][128 columns] in the layout of gemm_tn_kernel (same swizzle, same
// transposing reads), three stages deep = 96 KB, which still leaves a co-resident main-stream workgroup its 64 KB. Blocks at the edge of an output
// whose width is 128 (mod 256) are 128 wide (NI / NJ = 1): 384 x 384 is covered by a 256x256, a 256x128, a 128x256 and a 128x128 block.
// tn_frag_issue with the address split into a per-lane constant and the 16-column MFMA tile index t16 (0..7) of the sub-image: the swizzled
// chunk index (2*t16 + (p >> 1)) ^ (gsw(row) << 1) is lane_off's chunk bits XOR (t16 << 1) (gsw(row + 4) == gsw(row): the second read is the
// first + 4 rows = offset:1024) -- one v_xad per fragment instead of two precomputed addresses per fragment and stage (96 registers)
__device__ __forceinline__ void tn2_frag_issue(const unsigned lane_off, const unsigned tile_base, const int t16, s16x4_t& lo, s16x4_t& hi) {
    const unsigned a = (lane_off ^ ((unsigned)t16 << 5)) + tile_base;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(a));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(hi) : "v"(a));
}

// WI = waves along I (8 / WI along J); i0 / j0 = first output row / column of the block. (NI, NJ, WI) = (3, 1, 4) is round 4's 384 x 128 block: the
// CvT stage-3 gradients (384 x 384, 384 x 1536, 1536 x 384) are covered by EQUAL blocks -- as 256-blocks they were a mix of 256 x 256, 256 x 128
// and 128 x 128 blocks with 4 : 2 : 1 work per workgroup, and the launch waited for the big ones.
// NW = waves of the workgroup (8, or 4: round 5's co-resident form, gemm_tn4_kernel), NST = staging depth
// AUX = cache policy of the staging loads (0 default, 2 = non-temporal: lab switch CXR_TN_NT=1, round 5)
// JT = 16-column tiles of the block along J that hold data (0: all NJ * 8). Round 6: (NI, NJ, WI, JT) = (3, 2, 4, 12) is a 384 x 192 block -- the last
// Q sub-image is staged for its first 64 columns only (the other lanes of its LDS-DMA pieces fetch the cached zero row) and no wave reads the rest.
template <int NI, int NJ, int WI = 2, int NW = 8, int NST = 3, int AUX = 0, int JT = 0>
__device__ __forceinline__ void gemm_tn2_body(const GemmTnArgs& g, unsigned char* lds, const int i0, const int j0, const bool bias_block, const int split) {
    constexpr int BR = 32, TILE = BR * 256, SP = 8 / NW, LPS = (NI + NJ) * SP, STAGE = (NI + NJ) * TILE;      // SP = staging passes per sub-image
    constexpr int WJ = NW / WI;
    constexpr int JT16 = JT ? JT : NJ * 8;                         // live 16-column tiles along J
    constexpr int MI = NI * 8 / WI, MJ = JT16 / WJ;                // 16 x 16 MFMA tiles of a wave: (128 NI / WI) x (16 JT16 / WJ) outputs
    static_assert((NW == 8 || NW == 4) && (NI * 8) % WI == 0 && JT16 % WJ == 0 && JT16 <= NJ * 8 && MI % 2 == 0 && MJ >= 1 && MJ <= 6, "wave grid");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // NW waves: WI (I) x WJ (J)
    const int wi = wave % WI, wj = wave / WI;
    const int nrt = (g.R + BR - 1) / BR;
    const int rt0 = split * g.rt_per_split;
    const int rt1 = min(nrt, rt0 + g.rt_per_split);
    const int nt = rt1 - rt0;
    if (nt <= 0) return;

    // staging slot of a sub-image: slot = pass * (64 NW) + tid -> row = slot >> 4, physical chunk = tid & 15, logical chunk = physical ^ (gsw(row) << 1)
    // (gsw(row) only depends on row bits 0, 1, 3: with 4 waves the second pass is 16 rows further down and flips bit 4 only -- same column)
    const int srow = tid >> 4;
    const int scol = ((tid & 15) ^ (tn_gsw(srow) << 1)) * 8;
    const bf16_t* zr = reinterpret_cast<const bf16_t*>(g_tn_zero_row) + (tid & 15) * 8;
    auto stage = [&](int st, int rt) {
#pragma unroll
        for (int ps = 0; ps < SP; ++ps) {
            unsigned char* base = lds + st * STAGE + (ps * NW + wave) * 1024;
            const long r = (long)rt * BR + srow + ps * (NW * 4);
            const bool ok = r < g.R;
#pragma unroll
            for (int u = 0; u < NI; ++u) {
                int c = i0 + u * 128 + scol; if (c >= g.I) c = 0;
                const bf16_t* sp = ok ? g.P + r * g.ldp + c : zr;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sp,
                                                 (__attribute__((address_space(3))) void*)(base + u * TILE), 16, 0, AUX);
            }
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                int c = j0 + u * 128 + scol; if (c >= g.J) c = 0;
                const bool live = JT == 0 || u * 128 + scol < JT16 * 16;       // (compile-time true for full blocks)
                const bf16_t* sq = (ok && live) ? g.Q + r * g.ldq + c : zr;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sq,
                                                 (__attribute__((address_space(3))) void*)(base + (NI + u) * TILE), 16, 0, AUX);
            }
        }
    };

    f32x4_t acc[MI][MJ];
#pragma unroll
    for (int a = 0; a < MI; ++a)
#pragma unroll
        for (int b = 0; b < MJ; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of P: the 8 tokens a lane holds of its column, added with v_dot2c_f32_bf16 against (1, 1) -- one fp32 register
    // per 16-column tile (the all-ones MFMA of gemm_tn_kernel needs four)
    const bool do_bias = g.dbias != nullptr && bias_block && wj == 0;
    float accb[MI];
#pragma unroll
    for (int a = 0; a < MI; ++a) accb[a] = 0.f;
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const bf16x2_t one2 = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
    // this wave's first 16-column MFMA tile inside the block (tile r16 lives in sub-image r16 >> 3, at 16-column index r16 & 7: wave-uniform)
    const int ri0 = wi * MI, rj0 = wj * MJ;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned lane_off;                                              // this lane's part of every fragment address (tn2_frag_issue)
    {
        const int g4 = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        const int r0 = 8 * g4 + q;
        lane_off = r0 * 256 + ((((tn_gsw(r0) << 1) | (p >> 1)) << 4)) + (p & 1) * 8;
    }

#pragma unroll
    for (int s2 = 0; s2 < NST - 1; ++s2)
        if (s2 < nt) stage(s2, rt0 + s2);
    for (int kt = 0; kt < nt; ++kt) {
        const int ahead = nt - 1 - kt;
        if (ahead >= NST - 2) wait_vmcnt<LPS * (NST - 2)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nt) stage((kt + NST - 1) % NST, rt0 + kt + NST - 1);
        const unsigned tp = lds_base + (kt % NST) * STAGE;                      // wave-uniform
        const unsigned tq = tp + NI * TILE;
        s16x4_t alo[MI], ahi[MI], blo[MJ], bhi[MJ];
#pragma unroll
        for (int t = 0; t < MI; ++t) tn2_frag_issue(lane_off, tp + ((ri0 + t) >> 3) * TILE, (ri0 + t) & 7, alo[t], ahi[t]);
#pragma unroll
        for (int t = 0; t < MJ; ++t) tn2_frag_issue(lane_off, tq + ((rj0 + t) >> 3) * TILE, (rj0 + t) & 7, blo[t], bhi[t]);
        // the B fragments arrive in issue order (LDS returns in order): column j of the MFMA grid starts as soon as its fragment is there,
        // the reads of the later columns land behind the MFMAs of the earlier ones
        bf16x8_t fa[MI];
#pragma unroll
        for (int j2 = 0; j2 < MJ; ++j2) {
            if (MJ - 1 - j2 == 5) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
            else if (MJ - 1 - j2 == 4) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            else if (MJ - 1 - j2 == 3) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
            else if (MJ - 1 - j2 == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            else if (MJ - 1 - j2 == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (j2 == 0) {
#pragma unroll
                for (int t = 0; t < MI; ++t) fa[t] = tn_frag_join(alo[t], ahi[t]);
                if (do_bias) {
#pragma unroll
                    for (int i2 = 0; i2 < MI; ++i2) {
                        const uint4 w = __builtin_bit_cast(uint4, fa[i2]);
                        accb[i2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.x), one2, accb[i2], false);
                        accb[i2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.y), one2, accb[i2], false);
                        accb[i2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.z), one2, accb[i2], false);
                        accb[i2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, w.w), one2, accb[i2], false);
                    }
                }
            }
            const bf16x8_t fb = tn_frag_join(blo[j2], bhi[j2]);
#pragma unroll
            for (int i2 = 0; i2 < MI; ++i2) acc[i2][j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i2], fb, acc[i2][j2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // epilogue: the waves park 32 rows of their sub-tile at a time in the (idle) staging LDS and hand them out as runs of whole 128 / 256-byte
    // row pieces per wave-instruction, as gemm_tn_kernel does
    const int fr = lane & 15, fq = lane >> 4;
    __syncthreads();
    constexpr int WCOLS = 16 * MJ;                                 // columns of a wave's sub-tile (64 or 32; 96 for the 384 x 192 block)
    if constexpr (WCOLS == 96) {
        // 96-column wave tiles: 32 rows x 96 floats parked per pass; handed out as 64 columns of one row per instruction, then the remaining 32
        // columns of two rows per instruction (whole 256- / 128-byte row pieces, as below)
        float* wt = reinterpret_cast<float*>(lds) + wave * (32 * 96);
        const int Jp_ = g.tiles_j * 128;
        const int Ipad_ = ((g.I + 127) / 128) * 128;
        const int iw = i0 + ri0 * 16, jw = j0 + rj0 * 16;
        const int fr_ = lane & 15, fq_ = lane >> 4;
#pragma unroll
        for (int pass = 0; pass < MI / 2; ++pass) {
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int jt = 0; jt < MJ; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        wt[(it * 16 + fq_ * 4 + r) * 96 + jt * 16 + fr_] = acc[pass * 2 + it][jt][r] * g.alpha;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            const int ib = iw + pass * 32;
            const int jA = jw + lane, jB = jw + 64 + (lane & 31), rB = lane >> 5;
            if (g.mode == 2) {
                float* wa = g.ws + ((long)split * Ipad_ + ib) * Jp_ + jA;
                float* wb = g.ws + ((long)split * Ipad_ + ib + rB) * Jp_ + jB;
#pragma unroll 8
                for (int row = 0; row < 32; ++row) wa[(long)row * Jp_] = wt[row * 96 + lane];
#pragma unroll 8
                for (int row = 0; row < 32; row += 2) wb[(long)row * Jp_] = wt[(row + rB) * 96 + 64 + (lane & 31)];
            } else {
#pragma unroll 8
                for (int row = 0; row < 32; ++row) {
                    const int i = ib + row;
                    if (i < g.I && jA < g.J) { if (g.mode == 1) g.C[(long)i * g.ldc + jA] += wt[row * 96 + lane]; else atomicAdd(g.C + (long)i * g.ldc + jA, wt[row * 96 + lane]); }
                }
#pragma unroll 8
                for (int row = 0; row < 32; row += 2) {
                    const int i = ib + row + rB;
                    const float v = wt[(row + rB) * 96 + 64 + (lane & 31)];
                    if (i < g.I && jB < g.J) { if (g.mode == 1) g.C[(long)i * g.ldc + jB] += v; else atomicAdd(g.C + (long)i * g.ldc + jB, v); }
                }
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (do_bias) {
#pragma unroll
            for (int it = 0; it < MI; ++it) {
                float v = accb[it];
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
                const int i = iw + it * 16 + fr_;
                if (fq_ == 0) {
                    if (g.mode == 2) g.wsb[(long)split * Ipad_ + i] = v;
                    else if (i < g.I) { if (g.mode == 1) g.dbias[i] += v; else atomicAdd(g.dbias + i, v); }
                }
            }
        }
        return;
    }
    constexpr int RPI = WCOLS == 96 ? 1 : 64 / WCOLS;              // rows per store instruction (1 or 2)
    float* wtile = reinterpret_cast<float*>(lds) + wave * (32 * WCOLS);
    const int Jp = g.tiles_j * 128;
    const int Ipad = ((g.I + 127) / 128) * 128;
    const int i_wave = i0 + ri0 * 16;
    const int j_wave = j0 + rj0 * 16;
    const int lrow = lane / WCOLS, lcol = lane % WCOLS;
    const int j = j_wave + lcol;
#pragma unroll
    for (int pass = 0; pass < MI / 2; ++pass) {                     // 32 rows = two 16-row MFMA tiles per pass
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < MJ; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    wtile[(it * 16 + fq * 4 + r) * WCOLS + jt * 16 + fr] = acc[pass * 2 + it][jt][r] * g.alpha;
        __builtin_amdgcn_s_waitcnt(0xC07F);                        // lgkmcnt(0): own writes landed (each wave reads back only its own tile)
        const int ibase = i_wave + pass * 32 + lrow;
        if (g.mode == 2) {                                         // partial tile of this split: plain stores (padded workspace: no bounds)
            float* wrow = g.ws + ((long)split * Ipad + ibase) * Jp + j;
#pragma unroll 8
            for (int row = 0; row < 32; row += RPI) wrow[(long)row * Jp] = wtile[row * WCOLS + lane];
        } else if (j < g.J) {
            if (g.mode == 1) {                                     // the only split: this workgroup owns the tile
#pragma unroll 8
                for (int row = 0; row < 32; row += RPI) {
                    const int i = ibase + row;
                    if (i < g.I) g.C[(long)i * g.ldc + j] += wtile[row * WCOLS + lane];
                }
            } else {
#pragma unroll 8
                for (int row = 0; row < 32; row += RPI) {
                    const int i = ibase + row;
                    if (i < g.I) atomicAdd(g.C + (long)i * g.ldc + j, wtile[row * WCOLS + lane]);
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                        // the next pass overwrites the rows just read
        __builtin_amdgcn_sched_barrier(0);
    }
    if (do_bias) {                                                  // (wave-uniform) lane l of the A fragment holds column l & 15, tokens 8 * (l >> 4) ..
#pragma unroll
        for (int it = 0; it < MI; ++it) {
            float v = accb[it];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            const int i = i_wave + it * 16 + fr;
            if (fq == 0) {
                if (g.mode == 2) g.wsb[(long)split * Ipad + i] = v;
                else if (i < g.I) {
                    if (g.mode == 1) g.dbias[i] += v;
                    else atomicAdd(g.dbias + i, v);
                }
            }
        }
    }
}

__global__ __launch_bounds__(512, 2) void gemm_tn2_kernel(const GemmTnArgs g, const int blocks_j) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[3 * 4 * 32 * 256];      // 3 stages x (2 + 2) sub-images = 96 KB
    int swz;
    {   // split-major work list cut into 8 contiguous runs, one per XCD (as gemm_tn_kernel)
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int blocks = gridDim.x / g.splits;
    const int split = swz / blocks, blk = swz % blocks;
    const int bi = blk / blocks_j, bj = blk % blocks_j;
    const int ni = g.I - bi * 256 > 128 ? 2 : 1, nj = g.J - bj * 256 > 128 ? 2 : 1;      // block-uniform
    if (ni == 2 && nj == 2) gemm_tn2_body<2, 2>(g, lds, bi * 256, bj * 256, bj == 0, split);
    else if (ni == 2) gemm_tn2_body<2, 1>(g, lds, bi * 256, bj * 256, bj == 0, split);
    else if (nj == 2) gemm_tn2_body<1, 2>(g, lds, bi * 256, bj * 256, bj == 0, split);
    else gemm_tn2_body<1, 1>(g, lds, bi * 256, bj * 256, bj == 0, split);
}

// ---- round 6: 384 x 192 blocks for outputs whose dimensions are multiples of 384 / 192 but not both of 256 (CvT stage 3: 384 x 384, 1536 x 384,
// 384 x 1536). As 256-blocks such an output is a mix of 256 x 256, 256 x 128 and 128 x 128 blocks: per 32 tokens a 384 x 384 output stages 1536
// columns for 147456 products (96 per column), equal 384 x 192 blocks stage 1152 (128 per column, what a 256 x 256 block gets) -- the operand bytes a
// launch pulls through its CUs (and from HBM: the blocks of a split run on one XCD but drift apart) drop by a quarter (1536 x 384: by a seventh).
// Eight waves as 4 (I) x 2 (J): 96 x 96 outputs = 6 x 6 MFMA tiles = 144 accumulator registers per lane; three stages of 3 + 2 sub-images = 120 KB.
template <int NST>
__global__ __launch_bounds__(512, 2) void gemm_tn5_kernel(const GemmTnArgs g, const int blocks_j) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[(NST * 5 * 32 * 256 > 8 * 32 * 96 * 4) ? NST * 5 * 32 * 256 : 8 * 32 * 96 * 4];      // NST stages x (3 + 2) sub-images (120 KB at 3); epilogue: 8 x 12 KB
    int swz;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        swz = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int blocks = gridDim.x / g.splits;
    const int split = swz / blocks, blk = swz % blocks;
    const int bi = blk / blocks_j, bj = blk % blocks_j;
    gemm_tn2_body<3, 2, 4, 8, NST, 0, 12>(g, lds, bi * 384, bj * 192, bj == 0, split);
}

// C[i][j] += sum of the splits' partial tiles, dbias likewise, in a FIXED order: SL lanes share one group of 4 columns, lane l sums the splits
// congruent to l (ascending), a butterfly over the SL lanes finishes (small outputs have up to 176 splits: one thread walking them serially is a
// chain of 176 dependent loads).
template <int SL>
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ wsb, float* __restrict__ C, long ldc,
                                                             float* __restrict__ dbias, int I, int J, int Ip, int Jp, int splits) {
    const int j4 = J / 4;
    const long total = (long)I * j4;
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) / SL;
    const int sl = threadIdx.x % SL;
    if (e < total) {
        const int i = (int)(e / j4), j = (int)(e % j4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // up to 12 partials per lane, ALL loaded before the first add (split index clamped, surplus dropped by a select): a loop of dependent
        // load -> add steps made this 2.4 MB reduction take 12 us
        float4 v[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int s2 = sl + k * SL;
            v[k] = *reinterpret_cast<const float4*>(ws + ((long)(s2 < splits ? s2 : splits - 1) * Ip + i) * Jp + j);
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const bool on = sl + k * SL < splits;
            acc.x += on ? v[k].x : 0.f; acc.y += on ? v[k].y : 0.f; acc.z += on ? v[k].z : 0.f; acc.w += on ? v[k].w : 0.f;
        }
#pragma unroll
        for (int o = 1; o < SL; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64); acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (sl == 0) {
            float* c = C + (long)i * ldc + j;
            if ((ldc & 3) == 0 && ((size_t)C & 15) == 0) {
                float4 o = *reinterpret_cast<float4*>(c);
                o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
                *reinterpret_cast<float4*>(c) = o;
            } else { c[0] += acc.x; c[1] += acc.y; c[2] += acc.z; c[3] += acc.w; }
        }
    } else if (dbias && e - total < I) {
        const int i = (int)(e - total);
        float acc = 0.f;
        for (int s2 = sl; s2 < splits; s2 += SL) acc += wsb[(long)s2 * Ip + i];
#pragma unroll
        for (int o = 1; o < SL; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (sl == 0) dbias[i] += acc;
    }
}

// ---- round 4: ONE reduce launch for many weight gradients ------------------------------------------------------------------------------------------
// A training step has ~170 weight-gradient GEMMs with several token splits; each used to be followed by its own gemm_tn_reduce_kernel launch (3.7 ms
// of weight-gradient-stream time per step: 20-30 us apiece for 0.6-20 MB of partial tiles, because a small launch in the shadow of the main stream's
// kernels gets few CUs and the stream then waits for it). cxr_gemm_tn_partial_bf16 only leaves the partial tiles behind (each GEMM in scratch of its
// own) and describes the pending sum; cxr_gemm_tn_reduce_batch adds up to TN_BATCH of them per launch, every output element still summed over its
// splits in the same fixed order by the same lane grouping as gemm_tn_reduce_kernel (bit-identical results).
constexpr int TN_BATCH = 40;
struct TnRedItem { const float* ws; const float* wsb; float* C; float* dbias; long ldc; int I, J, Ip, Jp, splits, sl; unsigned block0, pad; };
struct TnRedBatch { TnRedItem d[TN_BATCH]; int n; };

__global__ __launch_bounds__(256) void gemm_tn_reduce_batch_kernel(const TnRedBatch bt) {
    int k = 0;                                                     // (block-uniform) the last item whose first block is <= this block
    for (int i = 1; i < bt.n; ++i) k = blockIdx.x >= bt.d[i].block0 ? i : k;
    const TnRedItem& it = bt.d[k];
    const float* __restrict__ ws = it.ws; const float* __restrict__ wsb = it.wsb; float* __restrict__ C = it.C; float* __restrict__ dbias = it.dbias;
    const long ldc = it.ldc;
    const int I = it.I, J = it.J, Ip = it.Ip, Jp = it.Jp, splits = it.splits, SL = it.sl;
    const int j4 = J / 4;
    const long total = (long)I * j4;
    const long e = ((long)(blockIdx.x - it.block0) * 256 + threadIdx.x) / SL;
    const int sl = threadIdx.x % SL;
    if (e < total) {
        const int i = (int)(e / j4), j = (int)(e % j4) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 v[12];                                              // ALL partials of the lane loaded before the first add (as gemm_tn_reduce_kernel)
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const int s2 = sl + q * SL;
            v[q] = *reinterpret_cast<const float4*>(ws + ((long)(s2 < splits ? s2 : splits - 1) * Ip + i) * Jp + j);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            const bool on = sl + q * SL < splits;
            acc.x += on ? v[q].x : 0.f; acc.y += on ? v[q].y : 0.f; acc.z += on ? v[q].z : 0.f; acc.w += on ? v[q].w : 0.f;
        }
        for (int o = 1; o < SL; o <<= 1) {
            acc.x += __shfl_xor(acc.x, o, 64); acc.y += __shfl_xor(acc.y, o, 64); acc.z += __shfl_xor(acc.z, o, 64); acc.w += __shfl_xor(acc.w, o, 64);
        }
        if (sl == 0) {
            float* c = C + (long)i * ldc + j;
            if ((ldc & 3) == 0 && ((size_t)C & 15) == 0) {
                float4 o = *reinterpret_cast<float4*>(c);
                o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
                *reinterpret_cast<float4*>(c) = o;
            } else { c[0] += acc.x; c[1] += acc.y; c[2] += acc.z; c[3] += acc.w; }
        }
    } else if (dbias && e - total < I) {
        const int i = (int)(e - total);
        float acc = 0.f;
        for (int s2 = sl; s2 < splits; s2 += SL) acc += wsb[(long)s2 * Ip + i];
        for (int o = 1; o < SL; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (sl == 0) dbias[i] += acc;
    }
}

// Launch plan of a weight-gradient GEMM (shared by cxr_gemm_tn_bf16 and cxr_gemm_tn_plan)
struct TnPlan { bool big; bool b5; int tiles_i, tiles_j, blocks_j, wgs_per_split, splits, rt_per_split; long need; };
static TnPlan tn_plan(int R, int I, int J) {
    TnPlan p;
    p.tiles_i = cdiv(I, 128); p.tiles_j = cdiv(J, 128);
    const int nrt = cdiv(R, 32);
    static int tn2 = -1, target_wgs = -1, target_big = -1;
    if (tn2 < 0) { const char* e = getenv("CXR_TN2"); tn2 = (e && e[0] == '0') ? 0 : 1; }       // CXR_TN2=0: every shape through gemm_tn_kernel (A/B)
    // Workgroups a launch aims for. These kernels run on the weight-gradient stream BESIDE the critical dX / attention kernels and every
    // workgroup holds 64 KB of LDS (half a CU's co-residency): ~0.7 workgroups per CU leaves the main stream its slots and halves the
    // atomic traffic (one 64 KB fp32 tile per split). Measured on the training step: 448 -> 32.5 ms, 320 -> 31.7, 224 -> 31.3, 176 -> 31.0,
    // 128 -> 31.5 (same box); alone, a single launch is up to 1.4x faster at 320. CXR_TN_WGS overrides.
    if (target_wgs < 0) { const char* e = getenv("CXR_TN_WGS"); target_wgs = e ? atoi(e) : 176; if (target_wgs < 1) target_wgs = 176; }
    if (target_big < 0) { const char* e = getenv("CXR_TN2_WGS"); target_big = e ? atoi(e) : 96; if (target_big < 1) target_big = 96; }
    // (gemm_tn2_kernel aims for 96 workgroups: same-box alternation on the training step, 80 / 96 / 112 / 128 / 160 -> 42.2 / 41.85 / 42.3 / 42.4 / 42.8 ms)
    // 256 x 256 blocks (gemm_tn2_kernel) when both dimensions exceed 128 AND a workgroup gets a long enough token run: its prologue and its
    // 256-KB partial tile only pay off from ~24 steps of 32 tokens per workgroup (scripts/tn_micro.py: 8192 x 768 x 768 is 1.35x SLOWER with it,
    // 36864 x 768 x 768 1.5x faster)
    p.blocks_j = cdiv(J, 256);
    const int blocks = cdiv(I, 256) * p.blocks_j;
    static long min_steps = -1;                            // CXR_TN2_MIN: blocks x 32-token steps from which the 256 x 256 blocks are used
    if (min_steps < 0) { const char* e = getenv("CXR_TN2_MIN"); min_steps = e ? atol(e) : 4096; }
    p.big = tn2 && I > 128 && J > 128 && (long)blocks * nrt >= min_steps;
    // (round 4's 384 x 128 blocks and round 5's co-resident 4-wave form were faster alone and 0.2 - 1.0 ms SLOWER in the step; both were removed in
    //  round 6, their measurements stay in profiles/r04_tn_micro_blocks384.txt, r04_ab_wgrad_stream.txt, r05_tn4_coresident.txt)
    // round 6: equal 384 x 192 blocks (gemm_tn5_kernel) where 256-blocks would be of unequal size. CXR_TN5=0: the 256-blocks (A/B)
    static int tn5 = -1;
    if (tn5 < 0) { const char* e = getenv("CXR_TN5"); tn5 = (e && e[0] == '0') ? 0 : 1; }
    static long min5 = -1;                                 // CXR_TN5_MIN: blocks x 32-token steps from which the 384 x 192 blocks are used
    if (min5 < 0) { const char* e = getenv("CXR_TN5_MIN"); min5 = e ? atol(e) : 4096; }
    p.b5 = tn2 && tn5 && (I % 384) == 0 && (J % 192) == 0 && ((I % 256) != 0 || (J % 256) != 0) && (long)(I / 384) * (J / 192) * nrt >= min5;
    if (p.b5) p.big = true;
    if (p.b5) p.blocks_j = J / 192;
    p.wgs_per_split = p.b5 ? (I / 384) * p.blocks_j : (p.big ? blocks : p.tiles_i * p.tiles_j);
    static int target_b5 = -1;                             // CXR_TN5_WGS: workgroups a 384 x 192-block launch aims for
    if (target_b5 < 0) { const char* e = getenv("CXR_TN5_WGS"); target_b5 = e ? atoi(e) : 64; if (target_b5 < 1) target_b5 = 64; }
    int splits = cdiv(p.b5 ? target_b5 : (p.big ? target_big : target_wgs), p.wgs_per_split);
    const int max_splits = nrt / 8 > 0 ? nrt / 8 : 1;      // at least 256 tokens per split
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.rt_per_split = cdiv(nrt, splits);
    p.splits = cdiv(nrt, p.rt_per_split);
    const long Ip = p.tiles_i * 128, Jp = p.tiles_j * 128;
    p.need = (long)p.splits * Ip * Jp + (long)p.splits * Ip;
    return p;
}

// what a launch on this shape will do: *splits token splits, *ws_floats floats of scratch for the deterministic (partial tiles + ordered sum) path
extern "C" int cxr_gemm_tn_plan(int R, int I, int J, int* splits, long* ws_floats) {
    if (R <= 0 || I <= 0 || J <= 0) return CXR_ERR_ARG;
    const TnPlan p = tn_plan(R, I, J);
    if (splits) *splits = p.splits;
    if (ws_floats) *ws_floats = p.splits > 1 ? p.need : 0;
    return CXR_OK;
}

static int tn_sl(int splits) { return splits <= 12 ? 1 : (splits <= 48 ? 4 : 16); }      // 12 partials per lane at most (splits <= 176 + 1)

// pending != null: the partial tiles are left in `ws` and *pending describes the sum still to be added to C / dbias (splits == 0: nothing is pending --
// one split, or no usable scratch: the launch accumulated by itself)
static int tn_launch(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, float* dbias, int R, int I, int J, float alpha, float* ws,
                     long ws_floats, cxr_tn_pending* pending, hipStream_t stream) {
    if (R <= 0 || I <= 0 || J <= 0 || (I % 8) || (J % 8) || (ldp % 8) || (ldq % 8)) return CXR_ERR_ARG;
    if (pending) memset(pending, 0, sizeof(*pending));
    GemmTnArgs g;
    g.P = (const bf16_t*)P; g.ldp = ldp; g.Q = (const bf16_t*)Q; g.ldq = ldq; g.C = C; g.ldc = ldc; g.dbias = dbias;
    g.R = R; g.I = I; g.J = J; g.alpha = alpha;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("CXR_TN_DEBUG"); dbg = e ? atoi(e) : 0; } g.debug = dbg; }
    const TnPlan pl = tn_plan(R, I, J);
    const bool big = pl.big;
    const int tiles_i = pl.tiles_i, blocks_j = pl.blocks_j, tiles = pl.wgs_per_split;
    g.tiles_j = pl.tiles_j; g.rt_per_split = pl.rt_per_split; g.splits = pl.splits;
    // accumulation: one split -> atomics (each element receives exactly one add per launch: order-free); several splits -> partial tiles into the
    // caller's workspace + a reduce that adds them in a fixed order (deterministic; float atomics from several splits were measured in round 5:
    // 75 - 119 ms per step, the 256 x 256 blocks of the splits collide on the same rows); no / too small a workspace: atomics
    const int Ip = tiles_i * 128, Jp = g.tiles_j * 128;
    const long need = pl.need;
    g.ws = nullptr; g.wsb = nullptr;
    g.mode = 0;                                            // one split: every element gets exactly ONE atomic add per launch -- deterministic as it is
    if (g.splits > 1 && g.splits <= 192 && ws && ws_floats >= need && (J % 4) == 0) { g.mode = 2; g.ws = ws; g.wsb = ws + (long)g.splits * Ip * Jp; }
    static int stages = -1;                                // CXR_TN_STAGES = 2 | 4 (LDS 32 | 64 KB per workgroup)
    if (stages < 0) { const char* e = getenv("CXR_TN_STAGES"); stages = e ? atoi(e) : 4; }
    static int st5 = -1;                                   // CXR_TN5_STAGES = 2 | 3 (LDS 96 (epilogue) | 120 KB per workgroup)
    if (st5 < 0) { const char* e = getenv("CXR_TN5_STAGES"); st5 = (e && atoi(e) == 2) ? 2 : 3; }
    if (pl.b5 && st5 == 2) CXR_LAUNCH(gemm_tn5_kernel<2>, dim3(tiles * g.splits), dim3(512), 0, stream, g, blocks_j);
    else if (pl.b5)       CXR_LAUNCH(gemm_tn5_kernel<3>, dim3(tiles * g.splits), dim3(512), 0, stream, g, blocks_j);
    else if (big)         CXR_LAUNCH(gemm_tn2_kernel, dim3(tiles * g.splits), dim3(512), 0, stream, g, blocks_j);
    else if (stages == 2) CXR_LAUNCH(gemm_tn_kernel<2>, dim3(tiles * g.splits), dim3(256), 0, stream, g);
    else                  CXR_LAUNCH(gemm_tn_kernel<4>, dim3(tiles * g.splits), dim3(256), 0, stream, g);
    if (g.mode == 2 && pending) {
        pending->ws = g.ws; pending->wsb = g.wsb; pending->C = C; pending->dbias = dbias; pending->ldc = ldc;
        pending->I = I; pending->J = J; pending->Ip = Ip; pending->Jp = Jp; pending->splits = g.splits;
    } else if (g.mode == 2) {
        const long total = (long)I * (J / 4) + (dbias ? I : 0);
        const int sl = tn_sl(g.splits);
#define TN_RED(SL_) CXR_LAUNCH(gemm_tn_reduce_kernel<SL_>, dim3((unsigned)cdiv(total * SL_, 256)), dim3(256), 0, stream, g.ws, g.wsb, C, ldc, dbias, I, J, Ip, Jp, g.splits)
        if (sl == 1) TN_RED(1); else if (sl == 4) TN_RED(4); else TN_RED(16);
#undef TN_RED
    }
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_gemm_tn_bf16(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, float* dbias, int R, int I, int J,
                                float alpha, float* ws, long ws_floats, hipStream_t stream) {
    return tn_launch(P, ldp, Q, ldq, C, ldc, dbias, R, I, J, alpha, ws, ws_floats, nullptr, stream);
}

extern "C" int cxr_gemm_tn_partial_bf16(const void* P, long ldp, const void* Q, long ldq, float* C, long ldc, float* dbias, int R, int I, int J,
                                        float alpha, float* ws, long ws_floats, cxr_tn_pending* pending, hipStream_t stream) {
    if (!pending) return CXR_ERR_ARG;
    return tn_launch(P, ldp, Q, ldq, C, ldc, dbias, R, I, J, alpha, ws, ws_floats, pending, stream);
}

extern "C" int cxr_gemm_tn_reduce_batch(const cxr_tn_pending* pending, int n, hipStream_t stream) {
    if (n < 0 || (n > 0 && !pending)) return CXR_ERR_ARG;
    int i = 0;
    while (i < n) {
        TnRedBatch bt;
        bt.n = 0;
        unsigned blocks = 0;
        for (; i < n && bt.n < TN_BATCH; ++i) {
            const cxr_tn_pending& p = pending[i];
            if (p.splits <= 1) continue;                           // nothing pending for this one
            if (!p.ws || !p.C || p.I <= 0 || p.J <= 0 || (p.J % 4) || p.splits > 192 || (p.dbias && !p.wsb)) return CXR_ERR_ARG;
            TnRedItem& it = bt.d[bt.n++];
            it.ws = p.ws; it.wsb = p.wsb; it.C = p.C; it.dbias = p.dbias; it.ldc = p.ldc;
            it.I = p.I; it.J = p.J; it.Ip = p.Ip; it.Jp = p.Jp; it.splits = p.splits; it.sl = tn_sl(p.splits);
            it.block0 = blocks; it.pad = 0;
            const long total = (long)p.I * (p.J / 4) + (p.dbias ? p.I : 0);
            blocks += (unsigned)cdiv(total * it.sl, 256);
        }
        if (bt.n) CXR_LAUNCH(gemm_tn_reduce_batch_kernel, dim3(blocks), dim3(256), 0, stream, bt);
    }
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// 2-D transpose of a bf16 matrix (used by the backward pass to present dY^T / X^T / W^T as K-contiguous operands).
// in [R, C] (ld_in) -> out [C, R] (ld_out). 64x64 tile through LDS, 16-byte global accesses on both sides.
__device__ __forceinline__ void transpose_tile(const bf16_t* __restrict__ in, long ld_in, bf16_t* __restrict__ out, long ld_out, int R, int C,
                                               int r0, int c0, bf16_t (*tile)[64 + 2]) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int s = p * 256 + tid;            // 512 chunks of 8 elements
        const int r = s >> 3, c8 = (s & 7) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < R && c0 + c8 < C) v = *reinterpret_cast<const uint4*>(in + (long)(r0 + r) * ld_in + c0 + c8);
        const bf16_t* e = reinterpret_cast<const bf16_t*>(&v);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[r][c8 + j] = e[j];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int s = p * 256 + tid;
        const int c = s >> 3, r8 = (s & 7) * 8;   // output row = input column
        if (c0 + c < C && r0 + r8 < R) {
            bf16_t e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = tile[r8 + j][c];
            if (r0 + r8 + 8 <= R) {
                *reinterpret_cast<uint4*>(out + (long)(c0 + c) * ld_out + r0 + r8) = *reinterpret_cast<const uint4*>(e);
            } else {
                for (int j = 0; j < 8 && r0 + r8 + j < R; ++j) out[(long)(c0 + c) * ld_out + r0 + r8 + j] = e[j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ in, long ld_in,
                                                             bf16_t* __restrict__ out, long ld_out, int R, int C) {
    __shared__ bf16_t tile[64][64 + 2];
    transpose_tile(in, ld_in, out, ld_out, R, C, blockIdx.y * 64, blockIdx.x * 64, tile);
}

// Many matrices in ONE launch (the per-step W^T refresh of ~190 weight matrices: each is 0.3-5 MB, i.e. launch-latency-bound on its own).
// table[i] = {in, out, ld_in, ld_out, R, C, first_tile, tiles_x} (int64 x 8, device memory); grid = total number of 64x64 tiles.
__global__ __launch_bounds__(256) void transpose_batched_bf16_kernel(const long* __restrict__ table, int n) {
    __shared__ bf16_t tile[64][64 + 2];
    const long t = blockIdx.x;
    int lo = 0, hi = n - 1;                                  // last matrix whose first_tile <= t
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid * 8 + 6] <= t) lo = mid; else hi = mid - 1;
    }
    const long* e = table + lo * 8;
    const long local = t - e[6];
    const int tx = (int)(local % e[7]), ty = (int)(local / e[7]);
    transpose_tile(reinterpret_cast<const bf16_t*>(e[0]), e[2], reinterpret_cast<bf16_t*>(e[1]), e[3], (int)e[4], (int)e[5], ty * 64, tx * 64, tile);
}

extern "C" int cxr_transpose_batched_bf16(const long* table, int n, long total_tiles, hipStream_t stream) {
    if (n <= 0 || total_tiles <= 0 || !table) return CXR_ERR_ARG;
    CXR_LAUNCH(transpose_batched_bf16_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, table, n);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_transpose_bf16(const void* in, long ld_in, void* out, long ld_out, int R, int C, hipStream_t stream) {
    if (R <= 0 || C <= 0 || (C % 8) || (ld_in % 8) || (ld_out % 8)) return CXR_ERR_ARG;
    dim3 grid(cdiv(C, 64), cdiv(R, 64));
    CXR_LAUNCH(transpose_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, R, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// column sums of a bf16 matrix into f32 (bias gradients): out[c] (+)= sum_r in[r][c]
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ in, long ld, float* __restrict__ out,
                                                          int R, int C, int rows_per_block) {
    // block handles 32 column-chunks(8 wide) x rows_per_block rows; 256 threads = 32 chunk-lanes x 8 row-lanes
    __shared__ float red[8][256 + 8];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c8 = (blockIdx.x * 32 + cl) * 8;
    const int rbeg = blockIdx.y * rows_per_block, rend = min(R, rbeg + rows_per_block);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (c8 < C) {
        for (int r = rbeg + rl; r < rend; r += 8) {
            const uint4 v = *reinterpret_cast<const uint4*>(in + (long)r * ld + c8);
            float f[8]; unpack8(v, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][cl * 8 + j] = acc[j];
    __syncthreads();
    const int col = threadIdx.x;                 // 256 columns per block
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) s += red[r][col];
    const int c = blockIdx.x * 256 + col;
    if (c < C) atomicAdd(out + c, s);
}

extern "C" int cxr_colsum_bf16(const void* in, long ld, float* out, int R, int C, hipStream_t stream) {
    if (R <= 0 || C <= 0 || (C % 8) || (ld % 8)) return CXR_ERR_ARG;
    const int rows_per_block = 512;
    dim3 grid(cdiv(C, 256), cdiv(R, rows_per_block));
    CXR_LAUNCH(colsum_bf16_kernel, grid, dim3(256), 0, stream, (const bf16_t*)in, ld, out, R, C, rows_per_block);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
