// Row-strip bf16 MFMA GEMM for the 384-wide outputs of CvT stage 3 (round 6):
//
//   C[M,384] = epilogue( alpha * A[M,K] . W[384,K]^T )           (same contract and epilogue arithmetic as gemm_nt_kernel; K % 64 == 0)
//
// attention-output and FFN down projections and the input-gradient twins of the q / k / v / FFN-up projections: 36928 x 384 x {384, 1536} and
// 9280 x 384 x 384 at the benchmark batch, 6.4 ms of the training step's main stream at 150 - 470 TFLOP/s. What bounds the tiled kernels on these
// shapes is neither HBM nor the matrix pipe but what ONE compute unit can take in (MI355X_MICROARCH.md "Indexed rows": 24 - 33 GB/s per CU from HBM /
// the Infinity Cache, 66 - 73 from its XCD's L2): with 128 x 128 tiles every CU ingests 128 rows of A AND 128 rows of W per tile (64 FLOP per byte),
// and the three column tiles of a row block fetch the same A rows three times (twice from L2 at best).
// Here a workgroup owns a STRIP of 16 MT rows and ALL 384 columns: A comes in exactly once (from HBM), the 384 x K weight matrix streams from the
// XCD's L2 (0.3 - 1.2 MB: resident in every L2), 16 MT x 384 x 2 / (16 MT + 384) = 226 FLOP per ingested byte at MT = 10. One strip per workgroup,
// strips dispatched by the hardware as CUs become free (no static partition: beside the weight-gradient stream some CUs are taken).
//   * 8 waves as 2 (rows) x 4 (columns): a wave owns 8 MT rows x 96 columns = (MT / 2) x 6 MFMA tiles (16x16x32), issued as D^T = W . A^T so that a
//     lane owns 4 consecutive output columns (as gemm_nt_kernel);
//   * BK-deep steps (64 by default, two LDS stages; 32 with 2 - 4 stages for A/B), filled by 16-byte LDS-DMA NST - 1 steps ahead (counted vmcnt, one raw
//     barrier per step), the images XOR-swizzled on the source address as in gemm_nt_kernel (Swz<BK>): conflict-free ds_read_b128 fragment reads;
//   * epilogue through the idle staging LDS, one 16-row MFMA tile row per wave and pass: 16-byte accesses, 192 contiguous bytes per row and wave.
#include "gemm_args.h"
#include "../../include/cxrmate_hip.h"
#include <stdlib.h>

template <int MT, int NST, int BK = 32, int NTW = 6>            // NTW = 16-column MFMA tiles per wave: 6 (N = 384) | 3 (N = 192)
__device__ __forceinline__ void gemm_strip_body(const GemmArgs& g, const int strip, const int dbg) {      // dbg (timing experiments): 1 no MFMA, 2 no refills, 4 no fragment reads
    CXR_PRIO_MAIN();
    constexpr int CPR = BK / 8;                                  // 16-byte chunks per tile row (4 at BK = 32)
    constexpr int BM = 16 * MT, MW = MT / 2;                     // rows of the strip; MFMA tile rows per wave
    constexpr int NCOLS = 64 * NTW, WCOLS = 16 * NTW;               // columns of the output / of a wave
    constexpr int A_PASSES = (BM * CPR + 511) / 512, W_PASSES = (NCOLS * CPR + 511) / 512;      // LDS-DMA instructions per thread per stage
    constexpr int A_BYTES = A_PASSES * 512 * 16, W_BYTES = W_PASSES * 512 * 16, STAGE = A_BYTES + W_BYTES;
    constexpr int LPS = A_PASSES + W_PASSES;
    constexpr int ESTR = WCOLS + 4;                              // epilogue row stride in floats (pad: the 16 rows of a write land on distinct banks)
    constexpr int EPI_BYTES = 8 * 16 * ESTR * 4;                 // epilogue: 8 waves x 16 rows x (WCOLS + 4 pad) floats
    constexpr int LDS_BYTES = NST * STAGE > EPI_BYTES ? NST * STAGE : EPI_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    static_assert(MT % 2 == 0 && MT >= 2 && MT <= 16 && MT * NTW <= 60 && NST >= 2 && NST <= 5 && (NTW == 6 || NTW == 3), "strip geometry");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = strip * BM;

    // per-thread staging sources: slot s = p * 512 + tid -> tile row s / 4, physical chunk s % 4 holds logical chunk (s % 4) ^ Swz<32>::f(row)
    const bf16_t* srcA[A_PASSES];
    const bf16_t* srcW[W_PASSES];
#pragma unroll
    for (int p = 0; p < A_PASSES; ++p) {
        const int s = p * 512 + tid;
        // slots past the strip (the last pass of a strip whose 64 MT slots are not a multiple of 512) re-load the strip's last row into LDS padding:
        // every wave issues the same number of LDS-DMA instructions per stage, which is what the counted vmcnt below relies on
        const int row = s / CPR < BM ? s / CPR : BM - 1, c = (s % CPR) ^ Swz<BK>::f(row);
        int ra = m0 + row; ra = ra < g.M ? ra : g.M - 1;         // rows past the matrix: clamped, never stored
        srcA[p] = g.A + (long)ra * g.lda + c * 8;
    }
#pragma unroll
    for (int p = 0; p < W_PASSES; ++p) {
        const int s = p * 512 + tid;
        const int row = s / CPR < NCOLS ? s / CPR : NCOLS - 1, c = (s % CPR) ^ Swz<BK>::f(row);
        srcW[p] = g.W + (long)row * g.ldw + c * 8;
    }
    auto stage = [&](int buf, int kt) {
        unsigned char* la = lds + buf * STAGE;
        unsigned char* lw = la + A_BYTES;
        const long k0 = (long)kt * BK;
#pragma unroll
        for (int p = 0; p < A_PASSES; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcA[p] + k0),
                                             (__attribute__((address_space(3))) void*)(la + (p * 512 + wave * 64) * 16), 16, 0, 0);
#pragma unroll
        for (int p = 0; p < W_PASSES; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srcW[p] + k0),
                                             (__attribute__((address_space(3))) void*)(lw + (p * 512 + wave * 64) * 16), 16, 0, 0);
    };

    f32x4_t acc[NTW][MW];
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int j = 0; j < MW; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nk = g.K / BK;
#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) stage(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        // steps kt + 1 .. kt + NST - 2 may stay in flight; step kt must have landed (vmcnt counts in issue order)
        const int ahead = nk - 1 - kt;
        if (ahead >= NST - 2) wait_vmcnt<LPS * (NST - 2)>();
        else if (NST > 3 && ahead == 2) wait_vmcnt<LPS * 2>();
        else if (NST > 3 && ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nk && !(dbg & 2)) stage((kt + NST - 1) % NST, kt + NST - 1);          // refills the stage consumed in step kt - 1
        const unsigned char* la = lds + (kt % NST) * STAGE;
        const unsigned char* lw = la + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 32; ++kk) {
            bf16x8_t fa[MW], fw[NTW];
            if (!(dbg & 4) || kt == 0) {
#pragma unroll
                for (int t = 0; t < MW; ++t) {
                    const int ra = wm * (BM / 2) + t * 16 + fr;
                    fa[t] = *reinterpret_cast<const bf16x8_t*>(la + (ra * CPR + ((kk * 4 + fq) ^ Swz<BK>::f(ra))) * 16);
                }
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    const int rw = wn * WCOLS + t * 16 + fr;
                    fw[t] = *reinterpret_cast<const bf16x8_t*>(lw + (rw * CPR + ((kk * 4 + fq) ^ Swz<BK>::f(rw))) * 16);
                }
            }
            if (!(dbg & 1)) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MW; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fa[mt], acc[nt][mt], 0, 0, 0);
            } else {
                acc[0][0][0] += (float)fa[0][0] + (float)fw[NTW - 1][7];
            }
        }
    }

    // ---- epilogue: lane (fr, fq) owns C[m = .. + mt * 16 + fr][n = wn * 96 + nt * 16 + 4 fq .. + 3]. Per MFMA tile row the wave parks its 16 x 96 fp32
    // values (bias added) in its own 6.4 KB of the idle staging LDS (row stride 100 floats: the 16 rows of a write land on distinct banks) and reads
    // them back as 8 columns per lane, 12 lanes per row: every global access is 16 bytes per lane, 192 contiguous bytes per row.
    __syncthreads();                                             // every wave has finished reading the last step's fragments
    float* stg = reinterpret_cast<float*>(lds) + wave * (16 * ESTR);
    float4 bv[NTW];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt) {
        bv[nt] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (g.bias) bv[nt] = *reinterpret_cast<const float4*>(g.bias + wn * WCOLS + nt * 16 + fq * 4);
    }
    const bool rd_aux = g.act == 2;
    const bf16_t* opp = rd_aux ? g.aux : g.residual;
    const long ldop = rd_aux ? g.ldaux : g.ldr;
#pragma unroll
    for (int mt = 0; mt < MW; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const float4 v = make_float4(acc[nt][mt][0] * g.alpha + bv[nt].x, acc[nt][mt][1] * g.alpha + bv[nt].y, acc[nt][mt][2] * g.alpha + bv[nt].z,
                                         acc[nt][mt][3] * g.alpha + bv[nt].w);
            *reinterpret_cast<float4*>(stg + fr * ESTR + nt * 16 + fq * 4) = v;
        }
        // the wave reads back only what it wrote itself: no workgroup barrier, the reads below wait for the writes through lgkmcnt
        constexpr int CPW = WCOLS / 8, EIT = (16 * CPW + 63) / 64;      // 8-column chunks per wave row; read-back passes (16 rows x CPW slots)
        uint4 rop[EIT];
        float rsc[EIT];
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int sl0 = it * 64 + lane, sl = sl0 < 16 * CPW ? sl0 : 16 * CPW - 1;
            const int row = sl / CPW, cc = sl % CPW;
            int m = m0 + wm * (BM / 2) + mt * 16 + row; m = m < g.M ? m : g.M - 1;
            const int n = wn * WCOLS + cc * 8;
            if (opp) rop[it] = *reinterpret_cast<const uint4*>(opp + (long)m * ldop + n);
            if (g.row_scale) rsc[it] = g.row_scale[(unsigned)m / (unsigned)g.rs_rows];
        }
#pragma unroll
        for (int it = 0; it < EIT; ++it) {
            const int sl0 = it * 64 + lane, sl = sl0 < 16 * CPW ? sl0 : 16 * CPW - 1;
            const int row = sl / CPW, cc = sl % CPW;
            const int m = m0 + wm * (BM / 2) + mt * 16 + row;
            const int n = wn * WCOLS + cc * 8;
            const float4 lo = *reinterpret_cast<const float4*>(stg + row * ESTR + cc * 8);
            const float4 hi = *reinterpret_cast<const float4*>(stg + row * ESTR + cc * 8 + 4);
            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            if (rd_aux) {
                float a8[8];
                unpack8(rop[it], a8);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] *= gelu_grad_f(a8[j]);
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
            if (m < g.M && sl0 < 16 * CPW) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = pack8(v);
        }
        // the next tile row overwrites the parked values: this wave's reads above have returned (their values were consumed)
    }
}

template <int MT, int NST, int BK = 32, int NTW = 6>
__global__ __launch_bounds__(512) void gemm_strip384_kernel(const GemmArgs g, const int dbg) {
    gemm_strip_body<MT, NST, BK, NTW>(g, blockIdx.x, dbg);
}

// Up to three problems with the same N = 384 and K in ONE launch (the query / key / value projections of a CvT stage-3 layer and their input gradients:
// 36928 + 9280 + 9280 rows): workgroups [start[p], start[p + 1]) own the strips of problem p. As separate launches the two 9280-row problems run on a
// third of the chip behind the big one (measured: -0.7 ms of the strip kernel's gain, scripts/r6/call17.sh); grouped, their strips fill the tail of its wave.
struct StripGroupArgs { GemmArgs g[3]; int start[3]; int n; };
template <int MT, int NST, int BK>
__global__ __launch_bounds__(512) void gemm_strip384_group_kernel(const StripGroupArgs gg, const int dbg) {
    const int b = blockIdx.x;
    const int p = (gg.n > 2 && b >= gg.start[2]) ? 2 : ((gg.n > 1 && b >= gg.start[1]) ? 1 : 0);
    if (p == 0)      gemm_strip_body<MT, NST, BK, 6>(gg.g[0], b, dbg);
    else if (p == 1) gemm_strip_body<MT, NST, BK, 6>(gg.g[1], b - gg.start[1], dbg);
    else             gemm_strip_body<MT, NST, BK, 6>(gg.g[2], b - gg.start[2], dbg);
}

// (A staggered form of this loop -- the two waves of every SIMD half a step apart, one feeding the matrix pipe while the other reads fragments and issues
// its LDS-DMA share, two barriers per step -- was built, bit-identical, and measured SLOWER: 79.9 vs 62.2 us at K = 1536, TF step +0.6 ms. A barrier
// phase costs ~0.2 - 0.4 us by itself here (the empty phase loop: 51 us against 21), more than the overlap it buys: profiles/r06_gemm_strip.txt.)

static int strip_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int strip_enabled = -1, strip_force_mt = 0, strip_min_rows = 24577, strip_min_rows192 = 32768, strip_stages = 0, strip_dbg = 0;
static void strip_init() {
    if (strip_enabled >= 0) return;
    strip_enabled = strip_env("CXR_GEMM_STRIP", 1);             // CXR_GEMM_STRIP=0: the tiled / persistent kernels of rounds 1-5 (A/B)
    strip_force_mt = strip_env("CXR_STRIP_MT", 0);              // 2 | 4 | 6 | 10: force the strip height (16 MT rows)
    // automatic use from 24577 rows (= more than 6 x 256 row groups: the 160-row strips then give >= 154 workgroups): measured alone against the tiled /
    // persistent kernels (scripts/r6/strip_micro.py, profiles/r06_gemm_strip.txt) 36928 rows: 64 vs 70 / 62 us at K = 1536, 22 vs 28 / 26 at K = 384;
    // 18464 rows: 42 vs 36 / 35 and 16.4 vs 16.9 / 14.6; 9280 rows: 12.3 vs 14.9 / 10.4 -- the shorter strips re-stream the weights too often
    strip_min_rows = strip_env("CXR_STRIP_MIN_M", 24577);
    // N = 192 (CvT stage 2) strips of 192 rows from 32768 rows: alone 147456 x 192 x 768 111 (tiled) / 91 (persistent) -> 80 us, x 192: 44 / 38 -> 36,
    // 36864 x 192 x 192: 16 / 17 -> 12 us; TF step -0.03 .. -0.2 ms (call 19)
    strip_min_rows192 = strip_env("CXR_STRIP_MIN_M192", 32768);
    strip_stages = strip_env("CXR_STRIP_STAGES", 0);            // 0: automatic; 2 | 3 | 4: that many stages of 32-deep steps
    strip_dbg = strip_env("CXR_STRIP_DEBUG", 0);                // timing experiments (wrong results): 1 no MFMA, 2 no LDS-DMA refills, 4 no fragment reads
}

// tuning / A-B aid (like cxr_gemm_pk_config): enabled 0 | 1, mt 0 (automatic) | 2 | 4 | 6 | 10 (N = 384) | 8 | 12 | 16 (N = 192), min_rows (-2: the shipped thresholds), stages 0 (automatic: mt 10 = two stages of 64-deep steps, else four of 32) | 2 | 3 | 4 (stages of 32-deep steps; 3, 4: mt 10 only); negative = keep
extern "C" int cxr_gemm_strip_config(int enabled, int mt, int min_rows, int stages) {
    strip_init();
    if (mt > 0 && mt != 2 && mt != 4 && mt != 6 && mt != 10 && mt != 8 && mt != 12 && mt != 16) return CXR_ERR_ARG;      // (8 / 12 / 16: the N = 192 form)
    if (enabled >= 0) strip_enabled = enabled != 0;
    if (mt >= 0) strip_force_mt = mt;
    if (min_rows >= 0) { strip_min_rows = min_rows; strip_min_rows192 = min_rows; }
    else if (min_rows == -2) { strip_min_rows = 24577; strip_min_rows192 = 32768; }       // back to the shipped thresholds
    if (stages >= 0) strip_stages = stages;                       // (64: two stages of 64-deep steps, mt 10)
    return CXR_OK;
}

// true when the row-strip kernel took the problem: N == 384, K % 64 == 0, bf16 output, 16-byte aligned rows, no GELU / dropout in the epilogue
bool gemm_strip_launch(const GemmArgs& g, hipStream_t stream) {
    strip_init();
    const int enabled = strip_enabled, force_mt = strip_force_mt, min_rows = strip_min_rows, stages = strip_stages;
    if (!enabled) return false;
    if (g.N == 192) {
        // N = 192 (CvT stage 2): strips of 256 rows x all 192 columns (8 x 3 MFMA tiles per wave), two stages of 64-deep steps
        if ((g.K % 64) || g.M < strip_min_rows192 || g.out_f32 || g.act == 1 || g.drop_thr16 || !g.lds_epilogue || (g.act == 2 && g.residual)) return false;
        if ((g.lda % 8) || (g.ldw % 8) || (g.ldc % 8) || (((size_t)g.A | (size_t)g.W | (size_t)g.C) & 15)) return false;
        if (force_mt && force_mt != 16 && force_mt != 12 && force_mt != 8) return false;
        const int mt192 = force_mt ? force_mt : 12;
        const int grid192 = cdiv(g.M, 16 * mt192);
        if (mt192 == 16)      CXR_LAUNCH((gemm_strip384_kernel<16, 2, 64, 3>), dim3(grid192), dim3(512), 0, stream, g, strip_dbg);
        else if (mt192 == 12) CXR_LAUNCH((gemm_strip384_kernel<12, 2, 64, 3>), dim3(grid192), dim3(512), 0, stream, g, strip_dbg);
        else                  CXR_LAUNCH((gemm_strip384_kernel<8, 2, 64, 3>), dim3(grid192), dim3(512), 0, stream, g, strip_dbg);
        return true;
    }
    if (g.N != 384 || (g.K % 64) || g.M < min_rows || g.out_f32 || g.act == 1 || g.drop_thr16 || !g.lds_epilogue || (g.act == 2 && g.residual)) return false;
    if ((g.lda % 8) || (g.ldw % 8) || (g.ldc % 8) || (((size_t)g.A | (size_t)g.W | (size_t)g.C) & 15)) return false;
    // strip height: the largest that still gives every CU a strip (one round), at least 32 rows
    if (force_mt && force_mt != 2 && force_mt != 4 && force_mt != 6 && force_mt != 10) return false;      // (8 / 12 / 16 belong to the N = 192 form)
    const int rg = cdiv(g.M, 16);
    int mt = force_mt;
    if (!mt) mt = rg > 256 * 6 ? 10 : (rg > 256 * 4 ? 6 : (rg > 256 * 2 ? 4 : 2));
    const int grid = cdiv(g.M, 16 * mt);
    switch (mt) {
        // default: two stages of 64-deep steps (half the barriers of the 32-deep loop: 64.8 -> 56.2 us at K = 1536, TF step -0.22 ms; call 16)
        case 10: if (stages == 2) CXR_LAUNCH((gemm_strip384_kernel<10, 2>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 else if (stages == 3) CXR_LAUNCH((gemm_strip384_kernel<10, 3>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 else if (stages == 4) CXR_LAUNCH((gemm_strip384_kernel<10, 4>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 else CXR_LAUNCH((gemm_strip384_kernel<10, 2, 64>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 break;
        case 6:  if (stages == 2) CXR_LAUNCH((gemm_strip384_kernel<6, 2>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 else CXR_LAUNCH((gemm_strip384_kernel<6, 4>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 break;
        case 4:  if (stages == 2) CXR_LAUNCH((gemm_strip384_kernel<4, 2>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 else CXR_LAUNCH((gemm_strip384_kernel<4, 4>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 break;
        default: if (stages == 2) CXR_LAUNCH((gemm_strip384_kernel<2, 2>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 else CXR_LAUNCH((gemm_strip384_kernel<2, 4>), dim3(grid), dim3(512), 0, stream, g, strip_dbg);
                 break;
    }
    return true;
}

static bool strip_takes(const GemmArgs& g) {
    return g.N == 384 && (g.K % 64) == 0 && !g.out_f32 && g.act != 1 && !g.drop_thr16 && g.lds_epilogue && !(g.act == 2 && g.residual) && !(g.lda % 8) && !(g.ldw % 8) &&
           !(g.ldc % 8) && !(((size_t)g.A | (size_t)g.W | (size_t)g.C) & 15);
}

// true when the grouped row-strip launch took ALL n problems (same N = 384 and K; the largest has at least the automatic row threshold)
bool gemm_strip_group_launch(const GemmArgs* g, int n, hipStream_t stream) {
    strip_init();
    static int grp = -1;
    if (grp < 0) grp = strip_env("CXR_STRIP_GROUP", 1);          // CXR_STRIP_GROUP=0: grouped problems stay on gemm_nt_group_kernel (A/B)
    if (!strip_enabled || !grp || n < 1 || n > 3 || strip_force_mt) return false;
    int mmax = 0;
    for (int i = 0; i < n; ++i) {
        if (!strip_takes(g[i]) || g[i].K != g[0].K) return false;
        mmax = g[i].M > mmax ? g[i].M : mmax;
    }
    if (mmax < strip_min_rows) return false;
    StripGroupArgs gg;
    int grid = 0;
    for (int i = 0; i < 3; ++i) {
        gg.g[i] = g[i < n ? i : 0];
        gg.start[i] = grid;
        if (i < n) grid += cdiv(g[i].M, 160);
    }
    gg.n = n;
    CXR_LAUNCH((gemm_strip384_group_kernel<10, 2, 64>), dim3(grid), dim3(512), 0, stream, gg, strip_dbg);
    return true;
}
