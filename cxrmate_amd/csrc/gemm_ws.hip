// W-stationary bf16 MFMA GEMM for the channel-width products of CvT stage 3 (K = 384, N a multiple of 384):
//
//   C[M,N] = epilogue( alpha * A[M,384] . W[N,384]^T )         (same contract, epilogue and bit-for-bit results as gemm_nt_kernel)
//
// q / k / v / output projections, the FFN up-projection and their input-gradient twins are 36928 x {384, 1536} x 384 at the benchmark batch:
// 57 - 254 MB for 11 - 22 GFLOP, i.e. on the HBM side of the ridge. Tiled kernels (gemm.hip, gemm_pk.hip) spend their LDS ring on the W
// panel (re-streamed from L2 for every row tile) and on A rows that three column-tile workgroups fetch at the same time: only a third of the
// bytes they keep in flight are distinct HBM bytes, and they ran these shapes at 1.4 - 2.6 TB/s (profiles/r03_gemm_shapes_baseline.txt;
// in-kernel stamps: 0.9 us per 48 KB ring step against 0.5 us of MFMA work, 2.7 us epilogues throttled by every workgroup writing at once).
// Here the weights never move:
//   * one persistent workgroup of 8 waves per CU owns ONE 384-column slice of W for the whole launch; wave w keeps the 48 x 384 sub-panel of
//     its 48 output columns in 144 VGPRs, already in MFMA operand order (36 x 16-byte loads per lane, once);
//   * the LDS holds nothing but A: a ring of 3 (BM = 64) or 6 (BM = 32) row blocks x 384 channels, filled by LDS-DMA two blocks ahead
//     (96 KB of distinct HBM bytes in flight per CU), each block laid out as three [rows][256 B] sub-images (a 1 KB DMA piece = 4 rows x 256 B,
//     never across a row) with the 16-byte chunk index XOR-ed by the row so that fragment reads are bank-conflict free;
//   * ONE barrier per row block (not per K step): the waves of a CU drift apart inside a block, so one wave's epilogue VALU / stores run
//     under its SIMD partner's MFMAs, and HBM sees reads and writes mixed instead of in chip-wide phases;
//   * N = 1536 (FFN up / its GELU' twin): four workgroups, next to each other in the XCD-contiguous order, cover the four slices of the same
//     rows (A comes from HBM once, L2 serves the other three);
//   * register-direct epilogue (lanes own 8 + 4 consecutive columns through the W-row deal of gemm_pk.hip); residual / saved pre-activation /
//     DropPath operands (BM = 32 variant, which has the registers) come by inline-asm loads issued before the block's MFMAs.
#include "gemm_args.h"
#include "../../include/cxrmate_hip.h"
#include <stdlib.h>

struct WsSched { int slices, groups, nblocks; int dbg; };

typedef __attribute__((address_space(1))) const void* ws_gptr_t;
typedef __attribute__((address_space(3))) void* ws_lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int ws_u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int ws_u32x2_t;

// timing experiments (dbg & 4): s_memtime stamps of wave 0 of every workgroup: [workgroup][64] = stamps (tag in the low 2 bits), [63] = count
__device__ unsigned long long g_ws_stamps[512 * 64];

template <int N> __device__ __forceinline__ void ws_wait_pieces(int groups) {       // all but the `groups` youngest requests (3 pieces each) have landed
    if (groups >= N) wait_vmcnt<3 * N>();
    else if constexpr (N > 0) ws_wait_pieces<N - 1>(groups);
}

template <bool ROP>
__global__ __launch_bounds__(512, 2) void gemm_ws384_kernel(const GemmArgs g, const WsSched sc) {
    constexpr int BM = 32, K = 384, KS = K / 32, MT = BM / 16, NTW = 3;
    constexpr int BLOCK_BYTES = BM * K * 2, SUB_BYTES = BM * 256;  // one [32 rows x 384] bf16 block = three [32][256 B] sub-images = 24 KB
    constexpr int NA = ROP ? 3 : 6;                                // ring slots for A blocks; with a second [M,N] operand: 3 + 3 slots for its blocks
    constexpr int DEPTH = ROP ? 4 : 5;                             // requests in flight ahead of their use (see the phase loop)
    constexpr int BIAS_OFF = 6 * BLOCK_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char lds[6 * BLOCK_BYTES + 384 * 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    int nstamp = 0;
    auto stamp = [&](int tag) {
        if ((sc.dbg & 4) && nstamp < 63 && blockIdx.x < 512) {
            if (threadIdx.x == 0) g_ws_stamps[blockIdx.x * 64 + nstamp] = (__builtin_amdgcn_s_memtime() & ~3ull) | (unsigned)tag;
            ++nstamp;
        }
    };
    stamp(0);

    int L;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int slice = L % sc.slices, group = L / sc.slices;
    const int b0 = (int)((long)group * sc.nblocks / sc.groups), b1 = (int)((long)(group + 1) * sc.nblocks / sc.groups);
    const int nb = b1 - b0;                                        // row blocks of this workgroup (contiguous run)
    if (nb <= 0) return;
    const int ncol0 = slice * 384 + wave * 48;                     // first output column of this wave

    // ---- requests. One request = one 24 KB block brought in by LDS-DMA, 3 pieces of 1 KB per wave: the A rows of a block, or (second [M,N]
    // operand: residual / saved pre-activation) its rows x this slice's 384 columns; both land as [3 sub-images][32 rows][256 B]: piece p =
    // sub-image p / 8, rows 4 * (p % 8) .. + 3; lane l -> row + l / 16, physical 16-byte chunk l % 16 holds logical chunk (l % 16) ^ (row & 15).
    // Without a second operand request x = A block x; with one, request 2x = A block x and request 2x + 1 = operand block x.
    const bool rd_aux = g.act == 2;
    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Rb = reinterpret_cast<const char*>(rd_aux ? g.aux : g.residual) + (long)slice * 768;
    const long lda2 = g.lda * 2, ldr2 = (rd_aux ? g.ldaux : g.ldr) * 2;
    const int nreq = ROP ? 2 * nb : nb;
    auto issue_req = [&](int r) {
        const bool second = ROP && (r & 1);
        const int lb = ROP ? r >> 1 : r;
        unsigned char* base = lds + ((second ? 3 : 0) + lb % NA) * BLOCK_BYTES;
        const int m0 = (b0 + lb) * BM;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int p = wave * 3 + i;                            // wave-uniform piece index 0 .. 23
            const int sub = p / 8, r4 = (p % 8) * 4;
            const int row = r4 + (lane >> 4);
            int m = m0 + row; m = m < g.M ? m : g.M - 1;
            const int logical = (lane & 15) ^ (row & 15);
            const char* src = (second ? Rb + m * ldr2 : Ab + m * lda2) + sub * 256 + logical * 16;
            __builtin_amdgcn_global_load_lds((ws_gptr_t)src, (ws_lptr_t)(base + p * 1024), 16, 0, 0);
        }
    };
    // ---- the wave's W sub-panel -> registers, in MFMA operand order (the MFMA is issued as D^T = W . A^T: W is the "A" operand, row = lane & 15).
    // Tiles 0 and 1 cover columns 0 .. 31 of the wave dealt as 8q + {0..3} / 8q + {4..7} (a lane then owns 8 consecutive columns), tile 2 = columns 32 .. 47.
    // Straight from global memory a fragment load touches 16 rows x 64 bytes per instruction and the eight waves' 288 loads took ~30 k cycles
    // (in-kernel stamps); instead the slice's 384 rows come through the (still empty) LDS ring by LDS-DMA in two halves of 192 rows = 4 waves
    // (whole 768-byte rows, 16-byte chunks XOR-ed with the row key below so that the fragment reads are bank-conflict free), ~10 k cycles.
    bf16x8_t wf[NTW][KS];
    {
        const char* Wb = reinterpret_cast<const char*>(g.W) + (long)slice * 384 * g.ldw * 2;
        const long ldw2 = g.ldw * 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // 144 pieces of 1 KB: LDS byte o = piece * 1024 + lane * 16 -> row o / 768 of the half, physical chunk (o % 768) / 16
#pragma unroll
            for (int i = 0; i < 18; ++i) {
                const int o = (wave * 18 + i) * 1024 + lane * 16;
                const int row = o / 768, pc = (o % 768) >> 4;
                const int key = (row & 3) | (((row >> 3) & 3) << 2);
                const int lc = (pc & ~15) | ((pc & 15) ^ key);
                const char* src = Wb + (long)(h * 192 + row) * ldw2 + lc * 16;
                __builtin_amdgcn_global_load_lds((ws_gptr_t)src, (ws_lptr_t)(lds + (wave * 18 + i) * 1024), 16, 0, 0);
            }
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            if ((wave >> 2) == h) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const int row = (wave & 3) * 48 + (nt < 2 ? (fr >> 2) * 8 + nt * 4 + (fr & 3) : 32 + fr);
                    const int key = (row & 3) | (((row >> 3) & 3) << 2);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        const int c = ks * 4 + fq;
                        wf[nt][ks] = *reinterpret_cast<const bf16x8_t*>(lds + row * 768 + (((c & ~15) | ((c & 15) ^ key)) << 4));
                    }
                }
            }
            __builtin_amdgcn_s_barrier();                          // (the reads above are complete for every wave: next half / the A ring may overwrite)
        }
    }
    // the slice's bias (384 floats) goes to LDS; read back 12 values per lane in the epilogue
    if (tid < 384) reinterpret_cast<float*>(lds + BIAS_OFF)[tid] = g.bias ? g.bias[slice * 384 + tid] : 0.f;
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(wf[nt][ks]));
    stamp(0);
#pragma unroll
    for (int r = 0; r < DEPTH; ++r)
        if (r < nreq) issue_req(r);

    const bool has_rs = g.row_scale != nullptr;
    float rsv[MT];
    f32x4_t acc[NTW][MT];

    // ---- two wave groups in anti-phase. Waves w and w + 4 share a SIMD (its matrix pipe and its VALU issue): group 0 = waves 0-3 runs the
    // MFMAs of block b in phase 2b and its epilogue (VALU, stores) in phase 2b + 1; group 1 = waves 4-7 one phase later. In every phase one
    // wave of a SIMD feeds the matrix pipe while the other runs an epilogue, instead of both doing the same thing at half speed each.
    // Phases are separated by workgroup barriers, and every wave does the same memory bookkeeping in every phase p:
    //   * before the barrier: the request that phase p needs has landed (A block b for p = 2b; with a second operand, its block b for p = 2b + 1,
    //     read by group 0's epilogue in that phase and by group 1's in the next) -- a counted vmcnt on the wave's own pieces;
    //   * after the barrier: the next request goes out (DEPTH requests ahead). Its slot was last read in phase p - 1: A block x - NA in phase
    //     2(x - NA) + 1, operand block x - 3 in phase 2(x - 3) + 2.
    const int grp = wave >> 2;
    for (int p = 0; p <= 2 * nb; ++p) {
        const int need = ROP ? p : (p >> 1);                       // request that must have landed for this phase
        if ((ROP || !(p & 1)) && need < nreq) {
            const int last = ROP ? p - 1 + DEPTH : (p == 0 ? DEPTH - 1 : ((p - 1) >> 1) + DEPTH);     // newest request issued so far
            ws_wait_pieces<DEPTH>((last < nreq - 1 ? last : nreq - 1) - need);
        }
        __builtin_amdgcn_s_barrier();
        bool issued = false;
        if (ROP || !(p & 1)) {
            const int r = ROP ? p + DEPTH : (p >> 1) + DEPTH;
            if (r < nreq) { issue_req(r); issued = true; }
        }
        const int q = p - grp;
        const int lb = q >> 1;
        if (q < 0 || lb >= nb) continue;
        const int m0 = (b0 + lb) * BM;
        if (!(q & 1)) {
            // ---------------- MFMA phase of block lb
            __builtin_amdgcn_s_setprio(0);
            stamp(1);
            if (has_rs) {                                          // DropPath factors of the lane's rows (inline asm: not in the compiler's wait counting)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int m = m0 + mt * 16 + fr; m = m < g.M ? m : g.M - 1;
                    const float* pr = g.row_scale + (unsigned)m / (unsigned)g.rs_rows;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(rsv[mt]) : "v"(pr) : "memory");
                }
            }
            const unsigned char* blk = lds + (lb % NA) * BLOCK_BYTES;
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            // A fragments two K steps deep, order pinned (left to itself hipcc hoists the reads of all 12 steps: +80 registers, spills)
            bf16x8_t fa[2][MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                fa[0][mt] = *reinterpret_cast<const bf16x8_t*>(blk + (mt * 16 + fr) * 256 + ((fq ^ fr) << 4));
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        fa[(ks + 1) & 1][mt] = *reinterpret_cast<const bf16x8_t*>(blk + ((ks + 1) >> 2) * SUB_BYTES + (mt * 16 + fr) * 256 + (((((ks + 1) & 3) * 4 + fq) ^ fr) << 4));
                }
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], fa[ks & 1][mt], acc[nt][mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            stamp(2);
            continue;
        }
        // ---------------- epilogue phase of block lb: lane (fr, fq) owns row mt*16 + fr and columns fq*8 .. + 7 (acc[0], acc[1]) and 32 + fq*4 .. + 3 (acc[2]) of the wave's 48
        if (sc.dbg & 1) { if (acc[0][0][0] == 123.456f && acc[NTW - 1][MT - 1][3] == 1.5f) reinterpret_cast<float*>(g.C)[0] = 1.f; continue; }
        stamp(0);
        if (sc.dbg & 8) __builtin_amdgcn_s_setprio(2);             // (experiment: the epilogue wave wins the SIMD's vector issue against its partner's MFMAs)
        if (has_rs) {
            // the DropPath loads went out in the previous phase, behind that phase's request and ahead of this phase's
            if (issued) wait_vmcnt<3>(); else wait_vmcnt<0>();
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(rsv[mt]));
        }
        // bias and second operand from LDS (inline asm: in front of a compiler-visible ds_read hipcc waits vmcnt(0) while LDS-DMA is in flight)
        f32x4_t bq0, bq1, bq2;
        {
            const unsigned ba = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(lds + BIAS_OFF) + (wave * 48 + fq * 8) * 4;
            const unsigned bb = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(lds + BIAS_OFF) + (wave * 48 + 32 + fq * 4) * 4;
            asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b128 %2, %4\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(bq0), "=&v"(bq1), "=&v"(bq2) : "v"(ba), "v"(bb) : "memory");
        }
        const float bv[12] = {bq0[0], bq0[1], bq0[2], bq0[3], bq1[0], bq1[1], bq1[2], bq1[3], bq2[0], bq2[1], bq2[2], bq2[3]};
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + mt * 16 + fr;
            const bool ok = m < g.M;
            float v[12];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[0][mt][e] * g.alpha + bv[e];
                v[4 + e] = acc[1][mt][e] * g.alpha + bv[4 + e];
                v[8 + e] = acc[2][mt][e] * g.alpha + bv[8 + e];
            }
            float a12[12];
            if constexpr (ROP) {
                // the lane's 8 + 4 values of the second operand: row mt*16 + fr of its block, 16-byte chunks wave*6 + fq and wave*6 + 4 + fq/2
                const int row = mt * 16 + fr;
                const int c8 = wave * 6 + fq, c4 = wave * 6 + 4 + (fq >> 1);
                const unsigned char* rblk = lds + (3 + lb % 3) * BLOCK_BYTES;
                const unsigned a8 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(rblk + (c8 >> 4) * SUB_BYTES + row * 256 + (((c8 & 15) ^ fr) << 4));
                const unsigned a4 = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)(rblk + (c4 >> 4) * SUB_BYTES + row * 256 + (((c4 & 15) ^ fr) << 4) + (fq & 1) * 8);
                ws_u32x4_t r8;
                ws_u32x2_t r4;
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(r8), "=&v"(r4) : "v"(a8), "v"(a4) : "memory");
                unpack8(make_uint4(r8[0], r8[1], r8[2], r8[3]), a12);
                a12[8] = __uint_as_float(r4[0] << 16); a12[9] = __uint_as_float(r4[0] & 0xffff0000u);
                a12[10] = __uint_as_float(r4[1] << 16); a12[11] = __uint_as_float(r4[1] & 0xffff0000u);
            }
            const long n8 = ncol0 + fq * 8, n4 = ncol0 + 32 + fq * 4;
            if (g.act == 1) {
                if (g.aux && ok) {
                    *reinterpret_cast<uint4*>(g.aux + (long)m * g.ldaux + n8) = pack8(v);
                    uint2 pk; pk.x = pack2bf(v[8], v[9]); pk.y = pack2bf(v[10], v[11]);
                    *reinterpret_cast<uint2*>(g.aux + (long)m * g.ldaux + n4) = pk;
                }
#pragma unroll
                for (int e = 0; e < 12; ++e) v[e] = gelu_f(v[e]);
            } else if (rd_aux) {
                if constexpr (ROP) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) v[e] *= gelu_grad_f(a12[e]);
                }
            }
            const float rsc = has_rs ? rsv[mt] : 1.0f;
            if (has_rs && !g.rs_after) {
#pragma unroll
                for (int e = 0; e < 12; ++e) v[e] *= rsc;
            }
            if constexpr (ROP) {
                if (g.residual) {
#pragma unroll
                    for (int e = 0; e < 12; ++e) v[e] += a12[e];
                }
            }
            if (has_rs && g.rs_after) {
#pragma unroll
                for (int e = 0; e < 12; ++e) v[e] *= rsc;
            }
            if (ok && !((sc.dbg & 2) && v[0] != 123.456f)) {
                bf16_t* crow = reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc;
                *reinterpret_cast<uint4*>(crow + n8) = pack8(v);
                uint2 pk; pk.x = pack2bf(v[8], v[9]); pk.y = pack2bf(v[10], v[11]);
                *reinterpret_cast<uint2*>(crow + n4) = pk;
            }
        }
        stamp(3);
    }
    stamp(0);
    if ((sc.dbg & 4) && threadIdx.x == 0 && blockIdx.x < 512) g_ws_stamps[blockIdx.x * 64 + 63] = nstamp;
}

extern "C" int cxr_gemm_ws_stamps(void* out, long bytes) {
    if (bytes > (long)sizeof(unsigned long long) * 512 * 64) return CXR_ERR_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ws_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? CXR_OK : CXR_ERR_LAUNCH;
}

static int ws_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int ws_enabled = -1, ws_min_rows = 2048, ws_wgs = 256, ws_dbg = 0, ws_force = 0;

// tuning / A-B aid: enabled 0|1, bm 0 (automatic) | 32 | 64, wgs = workgroups of a launch (multiple of N / 384); negative = keep
extern "C" int cxr_gemm_ws_config(int enabled, int bm, int min_rows, int wgs, int dbg) {
    if (ws_enabled < 0) ws_enabled = ws_env("CXR_GEMM_WS", 1);
    if (enabled >= 0) ws_enabled = enabled;
    if (bm >= 0) ws_force = bm != 0;                               // bm != 0: also take the shapes the automatic rule leaves to the tiled kernels
    if (min_rows >= 0) ws_min_rows = min_rows;
    if (wgs > 0) ws_wgs = wgs;
    if (dbg >= 0) ws_dbg = dbg;
    return CXR_OK;
}

bool gemm_ws_launch(const GemmArgs& g, hipStream_t stream) {
    if (ws_enabled < 0) ws_enabled = ws_env("CXR_GEMM_WS", 1);
    if (!ws_enabled) return false;
    if (g.K != 384 || (g.N % 384) || g.M < ws_min_rows || g.out_f32 || g.drop_thr16 || !g.lds_epilogue || (g.act == 2 && g.residual)) return false;
    const bool rop = g.residual != nullptr || g.act == 2;
    // measured (scripts/ws_lab.py, cache-cold, MI355X): against gemm_nt_kernel 36928 x 1536 x 384 + GELU 97 -> 77 us (+ saved pre-activation 110 -> 85),
    // 36864 x 768 x 384 50 -> 36 us; N = 384 (27 vs 29 us) and the products with a second [M,N] operand (its 24 KB blocks halve the ring) do not
    // pay for the ~12 k cycles it takes to bring the weights in: they stay on the tiled kernels unless forced (ws_force)
    if (!ws_force && (g.N < 768 || rop)) return false;
    WsSched sc;
    sc.dbg = ws_dbg;
    sc.slices = g.N / 384;
    if (sc.slices > ws_wgs) return false;
    sc.nblocks = cdiv(g.M, 32);
    sc.groups = ws_wgs / sc.slices;
    if (sc.groups > sc.nblocks) sc.groups = sc.nblocks;
    const int grid = sc.groups * sc.slices;
    if (rop) CXR_LAUNCH((gemm_ws384_kernel<true>), dim3(grid), dim3(512), 0, stream, g, sc);
    else     CXR_LAUNCH((gemm_ws384_kernel<false>), dim3(grid), dim3(512), 0, stream, g, sc);
    return true;
}
