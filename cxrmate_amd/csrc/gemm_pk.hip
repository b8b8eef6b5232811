// Persistent big-tile bf16 MFMA GEMM (same contract and epilogue as gemm_nt_kernel of gemm.hip):
//
//   C[M,N] = epilogue( alpha * A[M,K] . W[N,K]^T )
//
// Why a second kernel. The products of this model are short in K (CvT: K = 64 .. 384, 1 .. 6 steps of 64) and tall in M (37 k .. 590 k rows):
// most of them sit at or below the HBM ridge (36928 x 384 x 384 moves 57 MB for 10.9 GFLOP), so what decides their time is how many bytes a CU
// keeps in flight and whether the load stream ever stops. gemm_nt_kernel (128 x 128 tile, 2 workgroups per CU, one K step in flight, every tile
// with its own prologue, barrier-per-step loop and epilogue) kept 1.4 - 2.6 TB/s of algorithmic traffic moving on those shapes (round-3 table,
// profiles/r03_gemm_shapes_baseline.txt). Here:
//   * ONE workgroup of 8 waves per CU, 256 x 128 (or 256 x 256) tile: half the L2 -> LDS bytes per FLOP of the 128 x 128 tile;
//   * the workgroup is PERSISTENT: it walks a list of tiles and its LDS-DMA ring (3 x 48 KB or 2 x 64 KB stages) runs across tile boundaries
//     -- while a tile's epilogue runs, the first K steps of the next tile are already in flight (96 KB per CU);
//   * counted vmcnt + raw s_barrier: a stage is waited for only when it is consumed; the LDS-DMA pieces of a step (6 - 8 per wave, 60 - 180
//     issue cycles each) are interleaved with the step's MFMAs by sched_group_barrier -- issued as one block in front of them they made the
//     step DMA-issue time PLUS MFMA time (all 8 waves run the same phase between two barriers: measured 0.96 us per step against 0.49 / 0.55
//     for the loads / the MFMAs alone);
//   * register-direct epilogue: the W rows of a wave's 32-column group are dealt to the two MFMAs that cover it so that a lane ends up with
//     EIGHT consecutive output columns (rows 8q + {0..3} of the group feed MFMA 0, rows 8q + {4..7} MFMA 1): bias, residual, activation and the
//     store are 16 bytes per lane straight from the accumulators, no LDS round trip, no barrier. The bias lives in registers for the whole
//     launch; residual / saved-pre-activation rows are fetched by inline-asm loads at the START of a tile's last K step, so they are older than
//     that step's LDS-DMA and a counted vmcnt retires them without draining the ring (a compiler-visible load beside LDS-DMA costs vmcnt(0));
//   * work partition without a tail: the row range is cut into `teams` equal runs of 16-row groups (to within one group), a team = the
//     workgroups that cover the N tiles of one run. They sit next to each other in the XCD-contiguous order, walk their rows in the same order
//     and so share the A rows through one L2, and each of them keeps ONE W panel for the whole launch.
// MFMA orientation and the K order of the accumulation are those of gemm_nt_kernel: results are bit-identical to it.
#include "gemm_args.h"
#include "../../include/cxrmate_hip.h"
#include <stdlib.h>
#include <type_traits>

struct PkSched { int tiles_n, teams, slots, rg_total; int dbg; int slot_major; };      // dbg (timing experiments only): 1 no epilogue, 2 no epilogue stores, 4 stamps, 8 (host) every workgroup
                                                                        // covers all N tiles of its own rows, 16 odd workgroups walk their M tiles backwards

// One configuration of the kernel: BM x BN tile, BK-deep ring stages (NST of them), WM x WN waves (wave tile (BM / WM) x 64), OCC workgroups
// per CU (for the register bound only).
template <int BM_, int BN_, int BK_, int NST_, int WM_, int WN_, int OCC_>
struct PkCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, NST = NST_, WM = WM_, WN = WN_, OCC = OCC_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int WTM = BM / WM, MT = WTM / 16, NT = 4, KK = BK / 32;
    static_assert(BN / WN == 64, "a wave covers 64 columns");
    static constexpr int ROWB = BK * 2, CPR = BK / 8;                  // bytes / 16-byte chunks per tile row
    static constexpr int A_BYTES = BM * ROWB, W_BYTES = BN * ROWB, STAGE = A_BYTES + W_BYTES;
    static constexpr int APASS = BM * CPR / THREADS, WPASS = BN * CPR / THREADS, LPS = APASS + WPASS;
    static constexpr int RPP = THREADS / CPR;                          // tile rows per LDS-DMA pass
    static_assert(RPP % 32 == 0 && BM % RPP == 0 && BN % RPP == 0, "pass geometry");
    static constexpr int BIAS_FLOATS = 1536 > BN ? 1536 : BN;          // LDS bias area: all N tiles of a workgroup when they fit, else one at a time
    static constexpr bool ROP = MT <= 4;                               // residual / saved pre-activation / DropPath operand supported (register budget)
};

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// XOR swizzle of the 16-byte chunk index inside a tile row. 128-byte rows (BK = 64): two rows per 256-byte bank row; 64-byte rows (BK = 32): four.
// `key` = the 4 bits that tell the 16 rows of one fragment read apart: row & 15 for the A tile (16 consecutive rows), and for the W tile, whose
// fragment reads take rows 8q + {0..3} (+ 4) of a 32-row group, (row & 3) | ((row >> 3) & 3) << 2 -- so that in terms of the reading lane
// (fr = lane & 15) both operands use pk_swz(fr).
template <int BK> __device__ __forceinline__ int pk_swz(int key) {
    if constexpr (BK == 64) return (key >> 1) & 7;
    else return (0x1320 >> (((key >> 2) & 3) * 4)) & 3;
}
__device__ __forceinline__ int pk_wkey(int row) { return (row & 3) | (((row >> 3) & 3) << 2); }

// timing experiments (dbg & 4): wave 0 of every workgroup stamps s_memtime at kernel entry, after each step's barrier, at the head and the end of
// each tile epilogue and at exit (kept in LDS, written out once at the end): [workgroup][64] = {count, stamps...}; tag in the low 2 bits
__device__ unsigned long long g_pk_stamps[512 * 64];

template <class C>
__global__ __launch_bounds__(C::THREADS, (C::OCC * C::WM * C::WN + 3) / 4) void gemm_nt_pk_kernel(const GemmArgs g, const PkSched sc) {
    constexpr int NST = C::NST, MT = C::MT, NT = C::NT, LPS = C::LPS, BN = C::BN, KK = C::KK;
    __shared__ __attribute__((aligned(16))) unsigned char lds[NST * C::STAGE + C::BIAS_FLOATS * 4 + 512];      // ring + the workgroup's bias panel(s) (fp32) + stamps
    float* lbias = reinterpret_cast<float*>(lds + NST * C::STAGE);
    int nstamp = 0;
    auto stamp = [&](int tag) {
        if ((sc.dbg & 4) && nstamp < 63) {
            const unsigned long long t = (__builtin_amdgcn_s_memtime() & ~3ull) | (unsigned)tag;
            const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)(lds + NST * C::STAGE + C::BIAS_FLOATS * 4) + (nstamp & 63) * 8;
            if (threadIdx.x == 0) asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(t) : "memory");
            ++nstamp;
        }
    };
    stamp(0);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % C::WM, wn = wave / C::WM;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- this workgroup's tile list
    int L;
    {
        const int nwg = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    // team-major (default): a team's workgroups (one per N tile) are neighbours in L, i.e. share an XCD and its L2 copy of the team's A rows.
    // slot-major (W far larger than an L2: the 30000-column LM head): the teams that work on the SAME N tiles are neighbours instead, so an XCD holds
    // few W tiles at a time and every one of them is fetched once per XCD, not once per workgroup and row tile (PMC: 2.2 GB -> see DESIGN.md section 6)
    const int team = sc.slot_major ? L % sc.teams : L / sc.slots, slot = sc.slot_major ? L / sc.teams : L % sc.slots;
    const int rg0 = (int)((long)team * sc.rg_total / sc.teams), rg1 = (int)((long)(team + 1) * sc.rg_total / sc.teams);
    const int m_begin = rg0 * 16, m_end = min(g.M, rg1 * 16);
    const int mtiles = (m_end - m_begin + C::BM - 1) / C::BM;
    const int ntl = (sc.tiles_n - slot + sc.slots - 1) / sc.slots;
    const int nk = g.K / C::BK;
    const int ntiles = mtiles * ntl;
    const int S = ntiles * nk;                                     // ring steps of this workgroup
    if (S <= 0) return;
    // walk order: one N tile per workgroup (ntl == 1, the usual case) -> M tiles in order; several N tiles -> the N tiles of an M tile back to back
    // (its A rows are re-read from L2 while they are hot)
    const bool rev = (sc.dbg & 16) && (L & 1);
    auto tile_ij = [&](int t, int& i, int& j) {
        i = t / ntl; j = t - i * ntl;
        if (rev) i = mtiles - 1 - i;
    };

    // ---- LDS-DMA sources: slot s = pass * THREADS + tid -> tile row = pass * RPP + tid / CPR, physical 16-byte chunk = tid % CPR holds the
    // logical chunk (tid % CPR) ^ swizzle(row) (RPP is a multiple of 32, so the swizzle does not depend on the pass). With these swizzles the 16
    // lanes that the LDS serves together in a fragment read hit 16 different 16-byte slots of the 256-byte bank row.
    const int r8 = tid / C::CPR;
    const int csrcA = (tid % C::CPR) ^ pk_swz<C::BK>(r8 & 15);
    const int csrcW = (tid % C::CPR) ^ pk_swz<C::BK>(pk_wkey(r8));
    uint32_t offA[C::APASS], offW[C::WPASS];
    auto set_sources = [&](int t) {
        int i, j;
        tile_ij(t, i, j);
        const int m0 = m_begin + i * C::BM, n0 = (slot + j * sc.slots) * BN;
#pragma unroll
        for (int p = 0; p < C::APASS; ++p) {
            int row = m0 + p * C::RPP + r8; row = row < g.M ? row : g.M - 1;
            offA[p] = (uint32_t)row * (uint32_t)(g.lda * 2) + csrcA * 16;
        }
#pragma unroll
        for (int p = 0; p < C::WPASS; ++p) {
            int row = n0 + p * C::RPP + r8; row = row < g.N ? row : g.N - 1;
            offW[p] = (uint32_t)row * (uint32_t)(g.ldw * 2) + csrcW * 16;
        }
    };
    const char* Ab = reinterpret_cast<const char*>(g.A);
    const char* Wb = reinterpret_cast<const char*>(g.W);
    // one LDS-DMA piece (1 KB per wave-instruction) of stage (buf, kt): pieces 0 .. APASS-1 = A passes, then the W passes
    auto issue_piece = [&](int piece, int buf, int kt) {
        unsigned char* la = lds + buf * C::STAGE;
        const uint32_t kb = (uint32_t)kt * (C::BK * 2);
        if (piece < C::APASS)
            __builtin_amdgcn_global_load_lds((gptr_t)(Ab + (size_t)(offA[piece] + kb)), (lptr_t)(la + (piece * C::THREADS + wave * 64) * 16), 16, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((gptr_t)(Wb + (size_t)(offW[piece - C::APASS] + kb)),
                                             (lptr_t)(la + C::A_BYTES + ((piece - C::APASS) * C::THREADS + wave * 64) * 16), 16, 0, 0);
    };
    auto issue = [&](int buf, int kt) {
#pragma unroll
        for (int p = 0; p < LPS; ++p) issue_piece(p, buf, kt);
    };

    f32x4_t acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets inside a stage (bytes): row * ROWB + ((kk * 4 + fq) ^ pk_swz(fr)) * 16 for both operands.
    // A fragment mt: row wm * WTM + mt * 16 + fr. W fragment nt = 2u + t: row wn * 64 + u * 32 + (fr >> 2) * 8 + t * 4 + (fr & 3).
    const int fsw = pk_swz<C::BK>(fr);
    const int fragA = (wm * C::WTM + fr) * C::ROWB;
    const int fragW = C::A_BYTES + (wn * 64 + (fr >> 2) * 8 + (fr & 3)) * C::ROWB;
    int chk[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) chk[kk] = ((kk * 4 + fq) ^ fsw) << 4;

    // ---- the bias of the workgroup's N tile lives in LDS (read back 2 x 32 bytes per lane in the epilogue): a compiler-visible global load
    // beside LDS-DMA is waited for with vmcnt(0), which would drain the ring in every epilogue, and 16 more registers do not fit the 256-column tile
    const bool bias_all = ntl * BN <= C::BIAS_FLOATS;
    auto load_bias = [&](int j) {
        const int cnt = bias_all ? ntl * BN : BN;
        for (int idx = tid; idx < cnt; idx += C::THREADS) {
            const int jj = bias_all ? idx / BN : j;
            int n = (slot + jj * sc.slots) * BN + idx % BN;
            n = n < g.N ? n : g.N - 1;
            lbias[idx] = g.bias ? g.bias[n] : 0.f;
        }
    };
    load_bias(0);                                                  // visible to every wave after the first step's barrier
    const uint32_t dseed = g.drop_thr16 ? *g.drop_seed : 0u;
    const bool rd_aux = g.act == 2;
    const bf16_t* opp = rd_aux ? g.aux : g.residual;
    const long ldop = rd_aux ? g.ldaux : g.ldr;
    const bool has_rop = C::ROP && opp != nullptr;
    u32x4_t rop[C::ROP ? MT : 1][2];
    float rsv[C::ROP ? MT : 1];                                    // DropPath factor of the lane's rows (same treatment as the residual)
    const bool has_rs = C::ROP && g.row_scale != nullptr;

    // ---- the ring: step s = (tile, k step) in walk order; stage s lives in buffer s % NST
    int ti = 0, ki = 0;                                            // issue cursor
    set_sources(0);
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) {
        if (s < S) {
            issue(s, ki);
            if (++ki == nk) { ki = 0; if (++ti < ntiles) set_sources(ti); }
        }
    }
    int tc = 0, kc = 0;                                            // compute cursor
    int bc = 0, bi = NST - 1;                                      // buffer of the stage being consumed / refilled
    bool landed = false;                                           // the stage of the coming step was already waited for (by a tile epilogue)

    // one ring step. ISSUE: refill the buffer consumed in the previous step with stage s + NST - 1 (main part of the walk) or not (the last
    // NST - 1 steps).
    auto step = [&](auto issue_tag, int s) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        if (!landed) {                                             // all but the stages younger than stage s have landed
            const int younger = S - 1 - s;
            if (ISSUE || younger >= NST - 2) wait_vmcnt<LPS * (NST - 2)>();
            else if (NST > 3 && younger == 1) wait_vmcnt<LPS>();
            else wait_vmcnt<0>();
        }
        landed = false;
        __builtin_amdgcn_s_barrier();
        stamp(1);
        const bool last = kc + 1 == nk;
        int i, j;
        tile_ij(tc, i, j);
        const int m0 = m_begin + i * C::BM + wm * C::WTM, n0 = (slot + j * sc.slots) * BN + wn * 64;
        if constexpr (C::ROP) {
            if (last && has_rop) {
                // residual / saved pre-activation of this tile, 16 bytes per lane in the accumulator layout, rows and columns clamped. Hidden from
                // the compiler's wait-count bookkeeping (inline asm): they are waited for by the counted vmcnt at the head of the epilogue.
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int m = m0 + mt * 16 + fr; m = m < g.M ? m : g.M - 1;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        int n = n0 + u * 32 + fq * 8; n = n < g.N ? n : g.N - 8;
                        const bf16_t* p = opp + (long)m * ldop + n;
                        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(rop[mt][u]) : "v"(p) : "memory");
                    }
                }
            }
            if (last && has_rs) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int m = m0 + mt * 16 + fr; m = m < g.M ? m : g.M - 1;
                    const float* p = g.row_scale + (unsigned)m / (unsigned)g.rs_rows;
                    asm volatile("global_load_dword %0, %1, off" : "=v"(rsv[mt]) : "v"(p) : "memory");
                }
            }
        }
        const unsigned char* st = lds + bc * C::STAGE;
        {
            // Issue order pinned by hand (sched_barrier between groups; the compiler still places the lgkmcnt waits): the first K half's fragments,
            // then groups of 4 MFMAs (one A fragment against the 4 W fragments). Behind each group of the first half: the A fragment of the
            // second half that replaces the one just used up, a W fragment of the second half, and LDS-DMA pieces of the stage being refilled --
            // spread over the groups so that a wave never sits in more than one or two 60 - 180-cycle DMA issues between MFMAs.
            constexpr int GROUPS = KK * MT;                        // MFMA groups of a step
            bf16x8_t fw[KK][NT], fa[KK][MT];
#pragma unroll
            for (int t = 0; t < NT; ++t) fw[0][t] = *reinterpret_cast<const bf16x8_t*>(st + fragW + ((t >> 1) * 32 + (t & 1) * 4) * C::ROWB + chk[0]);
#pragma unroll
            for (int t = 0; t < MT; ++t) fa[0][t] = *reinterpret_cast<const bf16x8_t*>(st + fragA + t * 16 * C::ROWB + chk[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gi = 0; gi < GROUPS; ++gi) {
                const int kk = gi / MT, mt = gi % MT;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[kk][nt], fa[kk][mt], acc[nt][mt], 0, 0, 0);
                if (kk + 1 < KK) {
                    fa[kk + 1][mt] = *reinterpret_cast<const bf16x8_t*>(st + fragA + mt * 16 * C::ROWB + chk[kk + 1 < KK ? kk + 1 : 0]);
                    if (mt < NT) fw[kk + 1][mt] = *reinterpret_cast<const bf16x8_t*>(st + fragW + ((mt >> 1) * 32 + (mt & 1) * 4) * C::ROWB + chk[kk + 1 < KK ? kk + 1 : 0]);
                }
                if constexpr (ISSUE) {
                    // pieces p with p * GROUPS / LPS == gi: evenly spread, the first right behind the first group
#pragma unroll
                    for (int p = 0; p < LPS; ++p)
                        if (p * GROUPS / LPS == gi) issue_piece(p, bi, ki);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if constexpr (ISSUE) {
            if (++ki == nk) { ki = 0; if (++ti < ntiles) set_sources(ti); }
            bi = bi + 1 == NST ? 0 : bi + 1;
        }
        bc = bc + 1 == NST ? 0 : bc + 1;
        if (!last) { ++kc; return; }
        kc = 0;
        ++tc;
        stamp(2);
        if (sc.dbg & 1) { if (acc[0][0][0] == 123.456f && acc[NT - 1][MT - 1][3] == 1.5f) reinterpret_cast<float*>(g.C)[0] = 1.f; return; }
        // ---- tile epilogue, straight from the accumulators: lane (fr, fq) owns rows mt*16 + fr and, per 32-column group u, columns u*32 + fq*8 .. + 7
        // (acc[2u][mt] = the first four, acc[2u+1][mt] = the last four)
        if constexpr (C::ROP) {
            // everything older than this step's LDS-DMA has landed after this wait: the residual loads AND the stage of the coming step
            if (ISSUE) {
                wait_vmcnt<LPS>();
                landed = NST > 2;
            } else {
                wait_vmcnt<0>();
            }
            if (has_rop) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    asm volatile("" : "+v"(rop[mt][0]), "+v"(rop[mt][1]));
            }
            if (has_rs) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    asm volatile("" : "+v"(rsv[mt]));
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + mt * 16 + fr;
            float rsc = 1.0f;
            if constexpr (C::ROP) {
                if (has_rs) rsc = rsv[mt];
            } else {
                if (g.row_scale) rsc = g.row_scale[(unsigned)(m < g.M ? m : g.M - 1) / (unsigned)g.rs_rows];
            }
            uint32_t dkey = 0;
            if (g.drop_thr16) dkey = dropout_row_key(dseed, g.drop_site, (uint32_t)(m / g.drop_rows_per_b), (uint32_t)(g.drop_t0 + m % g.drop_rows_per_b));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = n0 + u * 32 + fq * 8;
                const bool ok = m < m_end && n < g.N;
                float v[8];
                // (inline asm: in front of a compiler-visible ds_read hipcc puts s_waitcnt vmcnt(0) while this step's LDS-DMA is in flight)
                f32x4_t b0, b1;
                const unsigned baddr = (unsigned)(size_t)(__attribute__((address_space(3))) const float*)(lbias + (bias_all ? j * BN : 0) + wn * 64 + u * 32 + fq * 8);
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)" : "=&v"(b0), "=&v"(b1) : "v"(baddr) : "memory");
                const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[2 * u][mt][e] * g.alpha + bv[e];
                    v[4 + e] = acc[2 * u + 1][mt][e] * g.alpha + bv[4 + e];
                }
                acc[2 * u][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                acc[2 * u + 1][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                uint4 ro = make_uint4(0, 0, 0, 0);
                if constexpr (C::ROP) {
                    if (has_rop) ro = make_uint4(rop[mt][u][0], rop[mt][u][1], rop[mt][u][2], rop[mt][u][3]);
                }
                if (g.act == 1) {
                    if (g.aux && ok) *reinterpret_cast<uint4*>(g.aux + (long)m * g.ldaux + n) = pack8(v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
                } else if (rd_aux) {
                    float a8[8];
                    unpack8(ro, a8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_f(a8[e]);
                }
                if (g.drop_thr16) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const uint32_t bits = dropout_pair_bits(dkey, (uint32_t)(n + e) >> 1);
                        v[e] = (bits & 0xffffu) >= g.drop_thr16 ? v[e] * g.drop_inv : 0.f;
                        v[e + 1] = (bits >> 16) >= g.drop_thr16 ? v[e + 1] * g.drop_inv : 0.f;
                    }
                }
                if (g.row_scale && !g.rs_after) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= rsc;
                }
                if (g.residual) {
                    float a8[8];
                    unpack8(ro, a8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += a8[e];
                }
                if (g.row_scale && g.rs_after) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= rsc;
                }
                if (g.out_f32) {
                    float* cp = reinterpret_cast<float*>(g.C) + (long)(ok ? m : 0) * g.ldc + (ok ? n : 0);
                    float4 o0 = make_float4(v[0], v[1], v[2], v[3]), o1 = make_float4(v[4], v[5], v[6], v[7]);
                    if (g.accumulate && ok) {
                        const float4 p0 = *reinterpret_cast<const float4*>(cp), p1 = *reinterpret_cast<const float4*>(cp + 4);
                        o0.x += p0.x; o0.y += p0.y; o0.z += p0.z; o0.w += p0.w; o1.x += p1.x; o1.y += p1.y; o1.z += p1.z; o1.w += p1.w;
                    }
                    if (ok) { *reinterpret_cast<float4*>(cp) = o0; *reinterpret_cast<float4*>(cp + 4) = o1; }
                } else if (ok && !((sc.dbg & 2) && v[0] != 123.456f)) {
                    *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(g.C) + (long)m * g.ldc + n) = pack8(v);
                }
            }
        }
        stamp(3);
        if (!bias_all && tc < ntiles) {                            // another N tile comes and the panels do not all fit: swap the panel
            __builtin_amdgcn_s_barrier();                          // every wave has read the old panel
            int i2, j2;
            tile_ij(tc, i2, j2);
            load_bias(j2);                                         // (the next step's barrier publishes it)
        }
    };

    int s = 0;
    for (; s + NST - 1 < S; ++s) step(std::true_type{}, s);
    for (; s < S; ++s) step(std::false_type{}, s);
    if (sc.dbg & 4) {
        stamp(0);
        __syncthreads();
        if (threadIdx.x < 64 && blockIdx.x < 512) {
            const unsigned long long* ls = reinterpret_cast<const unsigned long long*>(lds + NST * C::STAGE + C::BIAS_FLOATS * 4);
            g_pk_stamps[blockIdx.x * 64 + threadIdx.x] = threadIdx.x == 63 ? (unsigned long long)nstamp : (threadIdx.x < nstamp ? ls[threadIdx.x] : 0ull);
        }
    }
}

extern "C" int cxr_gemm_pk_stamps(void* out, long bytes) {
    if (bytes > (long)sizeof(unsigned long long) * 512 * 64) return CXR_ERR_ARG;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pk_stamps), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? CXR_OK : CXR_ERR_LAUNCH;
}

static int pk_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}

static int pk_enabled = -1, pk_force = 0, pk_min_rows = 2048, pk_wgs = 0, pk_dbg = -1;

// the configurations that are built (index = `1000 + i` in cxr_gemm_pk_config / CXR_PK_CFG)
typedef PkCfg<256, 128, 64, 3, 4, 2, 1> PkC0;     // 8 waves, one workgroup per CU, 144 KB ring
typedef PkCfg<256, 256, 64, 2, 2, 4, 1> PkC1;     // 8 waves, one per CU, 128 KB ring (no second [M,N] operand)
typedef PkCfg<128, 128, 64, 2, 2, 2, 2> PkC2;     // 4 waves, two per CU, 64 KB ring
typedef PkCfg<128, 128, 32, 4, 2, 2, 2> PkC3;     // 4 waves, two per CU, 64 KB ring of 16 KB stages
typedef PkCfg<256, 128, 32, 3, 2, 2, 2> PkC4;     // 4 waves (wave tile 128 x 64), two per CU, 72 KB ring (no second [M,N] operand)
typedef PkCfg<128, 128, 32, 3, 2, 2, 3> PkC5;     // 4 waves, three per CU, 48 KB ring
typedef PkCfg<128, 128, 64, 4, 4, 2, 1> PkC6;     // 8 waves (wave tile 32 x 64), one per CU, 128 KB ring: 96 KB in flight

// tuning / A-B aid (like cxr_gemm_set_regstage): enabled 0 = every NT GEMM on gemm_nt_kernel; bn 0 = automatic choice, 128 / 256 = the one-per-CU
// configurations, 1000 + i = configuration i; wgs 0 = one launch fills the chip; negative = keep
extern "C" int cxr_gemm_pk_config(int enabled, int bn, int min_rows, int wgs) {
    if (pk_enabled < 0) pk_enabled = pk_env("CXR_GEMM_PK", 1);
    if (enabled >= 0) pk_enabled = enabled;
    if (bn >= 0) { if (bn != 0 && bn != 128 && bn != 256 && !(bn >= 1000 && bn <= 1006)) return CXR_ERR_ARG; pk_force = bn; }
    if (min_rows >= 0) pk_min_rows = min_rows;
    if (wgs >= 0) pk_wgs = wgs;
    if (wgs <= -100) pk_dbg = -wgs - 100;                          // (timing experiments: wgs = -(100 + debug bits))
    return CXR_OK;
}

template <class C>
static void pk_launch_cfg(const GemmArgs& g, int wgs, int dbg, hipStream_t stream) {
    if (wgs <= 0) wgs = 256 * C::OCC;
    PkSched sc;
    sc.dbg = dbg;
    sc.slot_major = 0;
    sc.tiles_n = cdiv(g.N, C::BN);
    sc.rg_total = cdiv(g.M, 16);
    if (dbg & 8) {
        sc.slots = 1;
        sc.teams = wgs;
        const int max_teams = cdiv(g.M, 64);
        if (sc.teams > max_teams) sc.teams = max_teams;
    } else if (!(dbg & 32) && (long)g.N * g.K * 2 > (16L << 20) && g.M >= 8 * C::BM && wgs >= 64 && sc.tiles_n * 8 >= wgs) {
        // weights far beyond the 4-MB L2s (LM head: 46 MB): 8 row teams x (wgs / 8 rounded down to a divisor-friendly count) column slots, slot-major
        sc.slot_major = 1;
        sc.teams = 8;
        sc.slots = wgs / 8;
        const int per = cdiv(sc.tiles_n, sc.slots);               // N tiles per slot; fewer slots with the same count leave less imbalance
        sc.slots = cdiv(sc.tiles_n, per);
    } else if (sc.tiles_n <= wgs) {
        sc.slots = sc.tiles_n;
        sc.teams = wgs / sc.tiles_n;
        const int max_teams = cdiv(g.M, 64);                     // no team below 64 rows
        if (sc.teams > max_teams) sc.teams = max_teams;
    } else {
        sc.slots = wgs;
        sc.teams = 1;
    }
    CXR_LAUNCH((gemm_nt_pk_kernel<C>), dim3(sc.teams * sc.slots), dim3(C::THREADS), 0, stream, g, sc);
}

bool gemm_pk_launch(const GemmArgs& g, hipStream_t stream) {
    if (pk_enabled < 0) {
        pk_enabled = pk_env("CXR_GEMM_PK", 1);
        pk_force = pk_env("CXR_PK_CFG", 0);
        pk_min_rows = pk_env("CXR_PK_MIN_M", 2048);
        pk_wgs = pk_env("CXR_PK_WGS", 0);
    }
    if (!pk_enabled) return false;
    if (g.M < pk_min_rows || (g.K % 64) || (g.N % 8) || !g.lds_epilogue || (g.act == 2 && g.residual)) return false;
    // 32-bit byte offsets of the LDS-DMA sources
    if ((long)g.M * g.lda * 2 >= (1L << 32) || (long)g.N * g.ldw * 2 >= (1L << 32)) return false;
    const bool rop = g.residual != nullptr || g.act == 2 || g.row_scale != nullptr;      // a second [M,N] operand is read: MT <= 4 configurations only
    int cfg = pk_force >= 1000 ? pk_force - 1000 : (pk_force == 128 ? 0 : (pk_force == 256 ? 1 : -1));
    if (cfg < 0) {
        // automatic choice, from the cache-cold shape table of the 2-image step (scripts/pk_lab.py on MI355X; us, gemm_nt_kernel -> here):
        //   256 x 256 tile: 36864 x 9216 x 768 763 -> 629, 8192 x 30000 x 768 578 -> 463, 147456 x 768 x 192 + GELU 144 -> 122, 589824 x 256 x 64 + GELU 150 -> 127
        //   256 x 128 tile: 36928 x 384 x 1536 + residual 72 -> 64, 147456 x 192 x 768 + residual 110 -> 95, 147456 x 192 x {192, 576} 46 -> 41 / 80 -> 68,
        //                   36928 x 384 x 384 29 -> 26, 36928 x 384 x 1728 72 -> 64
        //   left to gemm_nt_kernel: the decoder's 8192-row products (equal or slower here), 589824 x 64 x 256 + residual (94 vs 105), anything under 32 k rows
        if (!rop && (g.N >= 4096 || (g.M >= 65536 && g.N >= 256))) cfg = 1;
        else if (g.M >= 32768 && !(g.N <= 64 && g.K >= 256)) cfg = 0;
        else return false;
    }
    if (rop && cfg == 1) cfg = 0;                                  // (forced configurations without registers for a second [M,N] operand)
    if (rop && cfg == 4) cfg = 3;
    static int dbg_env = -1;
    if (dbg_env < 0) dbg_env = pk_env("CXR_PK_DEBUG", 0);
    const int dbg = pk_dbg >= 0 ? pk_dbg : dbg_env;
    switch (cfg) {
        case 0: pk_launch_cfg<PkC0>(g, pk_wgs, dbg, stream); break;
        case 1: pk_launch_cfg<PkC1>(g, pk_wgs, dbg, stream); break;
        case 2: pk_launch_cfg<PkC2>(g, pk_wgs, dbg, stream); break;
        case 3: pk_launch_cfg<PkC3>(g, pk_wgs, dbg, stream); break;
        case 4: pk_launch_cfg<PkC4>(g, pk_wgs, dbg, stream); break;
        case 5: pk_launch_cfg<PkC5>(g, pk_wgs, dbg, stream); break;
        default: pk_launch_cfg<PkC6>(g, pk_wgs, dbg, stream); break;
    }
    return true;
}
