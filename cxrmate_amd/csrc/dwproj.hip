// CvT convolutional projections, fused over query / key / value  (TF5 modeling_cvt.py:93-110,113-130: depthwise 3x3 conv, pad 1, stride 1 for
// the query and 2 for key / value, each followed by BatchNorm2d, all three reading the SAME layer-normed activation).
//
// One workgroup = (image b, band of input rows, 64-channel slice). It stages its band of the activation ONCE into LDS as
// [band + 2 halo rows][W + 2 zero-padded columns][64 channels = one 128-byte line per token] and serves every tap of every projection from
// there: the earlier one-thread-per-output kernels (conv.hip) fetched 9 x 16 B of activation plus 9 x 32 B of taps through the L1 for every
// 16 B they produced and ran 5-10x off the HBM roofline. Lane = (pixel lane, 16-byte chunk): 8 consecutive lanes cover one token line, so a
// ds_read_b128 lane group reads two whole adjacent lines (conflict-free: MI355X_MICROARCH.md, LDS table) and global traffic is full lines.
// The taps of the projection being processed live in registers (72 VGPRs), loaded once per workgroup and projection.
//
// Train-mode BatchNorm (batch statistics) needs per-channel reductions over the whole launch: every workgroup writes ONE partial row per
// reduction and a 1024-thread kernel adds the rows and finishes the per-channel algebra (deterministic: no atomics).
//   forward : dw3_stats (sum c, sum c^2)        -> dw3_reduce_finalize (mean, rstd, running stats, folded taps) -> dw3_apply
//   backward: dw3_bwd_stats (sum dy, sum dy*c)  -> dw3_reduce_coef (dgamma, dbeta, coefficients of dc = a*dy + kb + kc*c)
//             -> dw3_dc_taps (dc in place + tap sums G[t] = sum dc * x_t, the raw-tap gradient) -> dw3_reduce_taps -> dw3_dx
// Eval mode (running statistics folded into the taps) uses dw3_apply / dw3_dc_taps without coefficients / dw3_dx with the folded taps.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/cxrmate_hip.h"

namespace {

constexpr int DW3_THREADS = 256;
constexpr int DW3_PL = 32;                         // pixel lanes per workgroup (256 threads / 8 chunks)
constexpr int DW3_LDS_CAP = 80 * 1024;             // two workgroups per CU (160 KB): dynamic tile + static reduction scratch
constexpr int DW3_LDS_MAX = 156 * 1024;            // one workgroup per CU: only when not even a 2-row band fits DW3_LDS_CAP (W = 96 input gradient)

struct Dw3Geo {
    const bf16_t* x; long x_bs, x_rs;              // activation / dx / staged gradient: [Bn, tok0 + H*W, C]
    int Bn, C, H, W, tok0, band, nbands, nslices, nproj;
};
struct Dw3P {                                      // one projection, device view
    const float* taps; const float* aux;           // [9][C] taps; aux = shift [C] (apply) or coef [3][C] (dc; may be null)
    bf16_t* y; long y_bs, y_rs;
    int stride, Ho, Wo;
    float inv8;                                    // apply only: > 0 -> y is an e4m3 matrix (strides in bytes) holding output * inv8
    // backward statistics only: the projection's FORWARD output (saved for the Linear layer's weight gradient anyway) and its BatchNorm parameters:
    // c = mean + (yf - beta) / (gamma * rstd) replaces the recomputed convolution wherever |gamma * rstd| is not tiny and |beta| <= 8 |gamma|
    const bf16_t* yf; long yf_bs, yf_rs; const float* gm; const float* bt; const float* mn; const float* rs;
};
struct Dw3Blk { int b, band_i, slice, ch, pl, c0; };

__device__ __forceinline__ Dw3Blk dw3_block(const Dw3Geo& g) {
    Dw3Blk k;
    const int bid = blockIdx.x;
    k.slice = bid % g.nslices;
    const int r = bid / g.nslices;
    k.band_i = r % g.nbands;
    k.b = r / g.nbands;
    k.ch = threadIdx.x & 7;
    k.pl = threadIdx.x >> 3;
    k.c0 = k.slice * 64 + k.ch * 8;
    return k;
}

// tile[rows][cols][8 chunks] <- token map rows [iy0, iy0+rows) x cols [ix0, ix0+cols) of `src` (pointing at spatial token 0, channel slice 0),
// zeros outside the H x W map. Four independent 16-byte loads in flight per thread.
__device__ __forceinline__ void dw3_stage(uint4* tile, const bf16_t* __restrict__ src, long rs, int H, int W, int iy0, int rows, int ix0, int cols) {
    const int total = rows * cols * 8;
    for (int s0 = threadIdx.x; s0 < total; s0 += DW3_THREADS * 4) {
        // UNCONDITIONAL loads (coordinates clamped into the map, the zero padding applied by a select afterwards): guarded by the per-lane
        // "inside the map" test, hipcc branches around every load and waits vmcnt(0) behind it -- the four loads became four dependent round trips
        uint4 v[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u * DW3_THREADS;
            const int sc = s < total ? s : 0;
            const int ch = sc & 7, pix = sc >> 3;
            const int rr = pix / cols, cc = pix - rr * cols;
            const int iy = iy0 + rr, ix = ix0 + cc;
            ok[u] = s < total && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int iyc = iy < 0 ? 0 : (iy < H ? iy : H - 1), ixc = ix < 0 ? 0 : (ix < W ? ix : W - 1);
            v[u] = *reinterpret_cast<const uint4*>(src + (long)(iyc * W + ixc) * rs + ch * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u * DW3_THREADS;
            if (s < total) tile[s] = ok[u] ? v[u] : make_uint4(0, 0, 0, 0);
        }
    }
}

__device__ __forceinline__ void dw3_load_taps(const float* __restrict__ taps, int C, int c0, float (&w)[9][8]) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const float4 a = *reinterpret_cast<const float4*>(taps + t * C + c0), b = *reinterpret_cast<const float4*>(taps + t * C + c0 + 4);
        w[t][0] = a.x; w[t][1] = a.y; w[t][2] = a.z; w[t][3] = a.w; w[t][4] = b.x; w[t][5] = b.y; w[t][6] = b.z; w[t][7] = b.w;
    }
}
__device__ __forceinline__ void dw3_load8(const float* __restrict__ p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// outputs of projection P that this band owns: rows [oy0, oy0 + n_oy); tap (ky, kx) of local output (oyl, ox) sits at tile slot
// ((oyl*st + ky) * pitch + ox*st + kx) * 8 + ch   (tile row 0 = input row band_i*band - 1, tile column 0 = input column -1)
__device__ __forceinline__ int dw3_band_outputs(const Dw3Geo& g, const Dw3P& P, int band_i, int& oy0) {
    oy0 = (band_i * g.band) / P.stride;
    int n = g.band / P.stride;
    if (oy0 + n > P.Ho) n = P.Ho - oy0;
    return n > 0 ? n * P.Wo : 0;
}

__device__ __forceinline__ void dw3_conv(const uint4* tile, int base, int pitch8, const float (&w)[9][8], float (&c)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c[j] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            float f[8];
            unpack8(tile[base + ky * pitch8 + kx * 8], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = fmaf(f[j], w[ky * 3 + kx][j], c[j]);
        }
}

// taps (9 rows) + `naux` rows of P.aux of this workgroup's 64-channel slice -> wl[q][12][64] (LDS): kernels whose accumulators leave no room for
// 72 tap registers read the taps from LDS per output instead (18 ds_read_b128; the LDS pipe is otherwise idle).
__device__ __forceinline__ void dw3_stage_taps(float (*wl)[12][64], const Dw3Geo& g, const Dw3P& p0, const Dw3P& p1, const Dw3P& p2, int slice, int naux) {
    const int rows = 9 + naux;
    for (int i = threadIdx.x; i < g.nproj * rows * 64; i += DW3_THREADS) {
        const int q = i / (rows * 64), r = (i / 64) % rows, c = i % 64;
        const Dw3P& P = q == 0 ? p0 : (q == 1 ? p1 : p2);
        // (unconditional load through a selected pointer: see dw3_stage)
        const float* sp = r < 9 ? P.taps + r * g.C : (P.aux ? P.aux + (r - 9) * g.C : P.taps);
        const float v = sp[slice * 64 + c];
        wl[q][r][c] = (r < 9 || P.aux) ? v : 0.f;
    }
}

// ---- workgroup reductions over the 32 pixel lanes: lanes differing in bits 3..5 inside a wave, then the 4 waves through `dw3_red`
// (a __shared__ float array of the calling kernel: DW3_RED_SMALL / DW3_RED_TAPS elements)
constexpr int DW3_RED_SMALL = 4 * 16 * 8, DW3_RED_TAPS = 4 * 80 * 8;      // floats

// out[(v * C) + c0 + j-th channel]: value v of chunk ch lives at red[wave][v][ch]; NV8 = number of values (each 1 float per channel) * 8
template <int NV>
__device__ __forceinline__ void dw3_reduce_small(float (&v)[NV], float* dw3_red, float* __restrict__ out_row, int C, int slice) {
    // NV = nvals * 8 (value-major, 8 channels of the thread's chunk each)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] += __shfl_xor(v[i], 8, 64);
        v[i] += __shfl_xor(v[i], 16, 64);
        v[i] += __shfl_xor(v[i], 32, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();                                       // previous use of dw3_red is over
    if (lane < 8) {
#pragma unroll
        for (int i = 0; i < NV; ++i) dw3_red[(wave * NV + i) * 8 + lane] = v[i];
    }
    __syncthreads();
    // NV*8 outputs: index o = (val * 8 + j) * 8 + ch  <->  red[wave][val*8 + j][ch]; channel = slice*64 + ch*8 + j
    for (int o = threadIdx.x; o < NV * 8; o += DW3_THREADS) {
        const int ch = o & 7, i = o >> 3;
        const float s = (dw3_red[(0 * NV + i) * 8 + ch] + dw3_red[(1 * NV + i) * 8 + ch]) + (dw3_red[(2 * NV + i) * 8 + ch] + dw3_red[(3 * NV + i) * 8 + ch]);
        out_row[(i >> 3) * C + slice * 64 + ch * 8 + (i & 7)] = s;
    }
}

// 80 values per lane (10 rows x 8 channels): first level by DPP (row_ror:8 = xor 8 inside a 16-lane row), the two upper levels exchange HALF
// of the remaining values each (recursive halving: 40 + 20 cross-lane moves instead of 160)
__device__ __forceinline__ float dw3_dpp_xor8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
}
__device__ __forceinline__ void dw3_reduce_taps(float (&v)[80], float* dw3_red, float* __restrict__ out_row, int C, int slice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool h16 = (lane >> 4) & 1, h32 = (lane >> 5) & 1;
#pragma unroll
    for (int i = 0; i < 80; ++i) v[i] += dw3_dpp_xor8(v[i]);
    float u[40];
#pragma unroll
    for (int i = 0; i < 40; ++i) {
        const float send = h16 ? v[i] : v[40 + i], keep = h16 ? v[40 + i] : v[i];
        u[i] = keep + __shfl_xor(send, 16, 64);
    }
    float r[20];
#pragma unroll
    for (int i = 0; i < 20; ++i) {
        const float send = h32 ? u[i] : u[20 + i], keep = h32 ? u[20 + i] : u[i];
        r[i] = keep + __shfl_xor(send, 32, 64);
    }
    __syncthreads();                                       // previous use of dw3_red is over
    if ((lane & 8) == 0) {
        const int base = (h16 ? 40 : 0) + (h32 ? 20 : 0);
#pragma unroll
        for (int i = 0; i < 20; ++i) dw3_red[(wave * 80 + base + i) * 8 + (lane & 7)] = r[i];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 80 * 8; o += DW3_THREADS) {
        const int ch = o & 7, i = o >> 3;
        const float s = (dw3_red[(0 * 80 + i) * 8 + ch] + dw3_red[(1 * 80 + i) * 8 + ch]) + (dw3_red[(2 * 80 + i) * 8 + ch] + dw3_red[(3 * 80 + i) * 8 + ch]);
        out_row[(i >> 3) * C + slice * 64 + ch * 8 + (i & 7)] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------ forward statistics
// ws row (b*nbands + band) of [nproj][2][C]: (sum c, sum c^2) over this workgroup's outputs
__global__ __launch_bounds__(DW3_THREADS, 2) void dw3_stats_kernel(const Dw3Geo g, const Dw3P p0, const Dw3P p1, const Dw3P p2, float* __restrict__ ws) {
    CXR_PRIO_MAIN();
    __shared__ float red[DW3_RED_SMALL];
    extern __shared__ uint4 dw3_tile[];
    const Dw3Blk k = dw3_block(g);
    const int pitch = g.W + 2, pitch8 = pitch * 8;
    dw3_stage(dw3_tile, g.x + (long)k.b * g.x_bs + (long)g.tok0 * g.x_rs + k.slice * 64, g.x_rs, g.H, g.W, k.band_i * g.band - 1, g.band + 2, -1, pitch);
    __syncthreads();
    float* row = ws + (long)(k.b * g.nbands + k.band_i) * g.nproj * 2 * g.C;
#pragma unroll 1
    for (int q = 0; q < g.nproj; ++q) {
        const Dw3P P = q == 0 ? p0 : (q == 1 ? p1 : p2);
        float w[9][8];
        dw3_load_taps(P.taps, g.C, k.c0, w);
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        int oy0;
        const int npo = dw3_band_outputs(g, P, k.band_i, oy0);
        for (int o = k.pl; o < npo; o += DW3_PL) {
            const int oyl = o / P.Wo, ox = o - oyl * P.Wo;
            float c[8];
            dw3_conv(dw3_tile, (oyl * P.stride * pitch + ox * P.stride) * 8 + k.ch, pitch8, w, c);
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc[j] += c[j]; acc[8 + j] = fmaf(c[j], c[j], acc[8 + j]); }
        }
        dw3_reduce_small<16>(acc, red, row + (long)q * 2 * g.C, g.C, k.slice);
    }
}

// ------------------------------------------------------------------------------------------------------------------ forward apply
// y_q = conv(x; folded taps) + shift for every projection; class-token row copied through (TF5:cvt:195-198)
__global__ __launch_bounds__(DW3_THREADS, 2) void dw3_apply_kernel(const Dw3Geo g, const Dw3P p0, const Dw3P p1, const Dw3P p2) {
    CXR_PRIO_MAIN();
    extern __shared__ uint4 dw3_tile[];
    const Dw3Blk k = dw3_block(g);
    const int pitch = g.W + 2, pitch8 = pitch * 8;
    const bf16_t* xb = g.x + (long)k.b * g.x_bs + k.slice * 64;
    dw3_stage(dw3_tile, xb + (long)g.tok0 * g.x_rs, g.x_rs, g.H, g.W, k.band_i * g.band - 1, g.band + 2, -1, pitch);
    __syncthreads();
#pragma unroll 1
    for (int q = 0; q < g.nproj; ++q) {
        const Dw3P P = q == 0 ? p0 : (q == 1 ? p1 : p2);
        bf16_t* yb = P.y + (long)k.b * P.y_bs + k.c0;
        unsigned char* yb8 = reinterpret_cast<unsigned char*>(P.y) + (long)k.b * P.y_bs + k.c0;      // (e4m3 output: same element strides, 1-byte elements)
        const bool q8 = P.inv8 > 0.f;
        if (g.tok0 && k.band_i == 0 && threadIdx.x < 8) {
            for (int t = 0; t < g.tok0; ++t) {
                const uint4 cls = *reinterpret_cast<const uint4*>(xb + (long)t * g.x_rs + k.ch * 8);
                if (q8) { float f[8]; unpack8(cls, f); *reinterpret_cast<uint2*>(yb8 + (long)t * P.y_rs) = pack8_fp8(f, P.inv8); }
                else *reinterpret_cast<uint4*>(yb + (long)t * P.y_rs) = cls;
            }
        }
        float w[9][8], sh[8];
        dw3_load_taps(P.taps, g.C, k.c0, w);
        dw3_load8(P.aux + k.c0, sh);
        int oy0;
        const int npo = dw3_band_outputs(g, P, k.band_i, oy0);
        for (int o = k.pl; o < npo; o += DW3_PL) {
            const int oyl = o / P.Wo, ox = o - oyl * P.Wo;
            float c[8];
            dw3_conv(dw3_tile, (oyl * P.stride * pitch + ox * P.stride) * 8 + k.ch, pitch8, w, c);
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] += sh[j];
            if (q8) *reinterpret_cast<uint2*>(yb8 + (long)(g.tok0 + (oy0 + oyl) * P.Wo + ox) * P.y_rs) = pack8_fp8(c, P.inv8);
            else *reinterpret_cast<uint4*>(yb + (long)(g.tok0 + (oy0 + oyl) * P.Wo + ox) * P.y_rs) = pack8(c);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------ backward statistics
// ws row of [nproj][2][C]: (sum dy, sum dy*c). c is the projection's raw convolution output: recomputed from the staged activation and the raw taps, or --
// when the caller hands over the projection's forward output yf = (c - mean) * gamma * rstd + beta (bf16, saved for the weight gradient of the Linear layer
// that follows) and, for all 64 channels of the slice, |gamma * rstd| >= 1e-3 and |beta| <= 8 |gamma| -- recovered as c = mean + (yf - beta) / (gamma * rstd):
// the pass is then a plain streaming reduction over (dy, yf) without the LDS tile, the 9 window reads and the 108 VALU instructions of the convolution
// per 8 outputs. The bf16 rounding of yf is 2^-9 |yf| (of the whole stored value, beta included), so the normalised activation xhat = (yf - beta) / gamma
// is recovered with an error of up to 2^-9 (|xhat| + |beta| / |gamma|): the second gate bounds that amplification (<= 2^-9 * (|xhat| + 8): the size of the
// bf16 rounding the activations carry anyway); a slice with a larger |beta| / |gamma| (pretrained BatchNorm parameters can have one) recomputes the
// convolution. The remaining per-element error is unbiased and averages out in the two sums (tests: same tolerances as the recomputed path).
__global__ __launch_bounds__(DW3_THREADS, 2) void dw3_bwd_stats_kernel(const Dw3Geo g, const Dw3P p0, const Dw3P p1, const Dw3P p2, float* __restrict__ ws) {
    CXR_PRIO_MAIN();
    __shared__ float red[DW3_RED_SMALL];
    extern __shared__ uint4 dw3_tile[];
    const Dw3Blk k = dw3_block(g);
    const int pitch = g.W + 2, pitch8 = pitch * 8;
    // which projections can take c from their forward output (workgroup-uniform: one decision per 64-channel slice)
    bool from_y[3] = {false, false, false};
    bool need_tile = false;
#pragma unroll 1
    for (int q = 0; q < g.nproj; ++q) {
        const Dw3P& P = q == 0 ? p0 : (q == 1 ? p1 : p2);
        int safe = 0;
        if (P.yf) {                                                 // (kernel-argument uniform)
            float gm[8], rs[8], bt[8];
            dw3_load8(P.gm + k.c0, gm); dw3_load8(P.rs + k.c0, rs); dw3_load8(P.bt + k.c0, bt);
            safe = 1;
#pragma unroll
            for (int j = 0; j < 8; ++j) safe &= (fabsf(gm[j] * rs[j]) >= 1e-3f && fabsf(bt[j]) <= 8.0f * fabsf(gm[j])) ? 1 : 0;
        }
        from_y[q] = __syncthreads_and(safe) != 0;
        need_tile = need_tile || !from_y[q];
    }
    if (need_tile) {
        dw3_stage(dw3_tile, g.x + (long)k.b * g.x_bs + (long)g.tok0 * g.x_rs + k.slice * 64, g.x_rs, g.H, g.W, k.band_i * g.band - 1, g.band + 2, -1, pitch);
        __syncthreads();
    }
    float* row = ws + (long)(k.b * g.nbands + k.band_i) * g.nproj * 2 * g.C;
#pragma unroll 1
    for (int q = 0; q < g.nproj; ++q) {
        const Dw3P P = q == 0 ? p0 : (q == 1 ? p1 : p2);
        const bf16_t* yb = P.y + (long)k.b * P.y_bs + k.c0;
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        int oy0;
        const int npo = dw3_band_outputs(g, P, k.band_i, oy0);
        // the gradient rows are prefetched two iterations ahead (unconditional, index clamped): loaded where it is used, every iteration of
        // this loop was one exposed global round trip
        const bf16_t* ybase = yb + (long)(g.tok0 + oy0 * P.Wo) * P.y_rs;
        const int last = npo > 0 ? npo - 1 : 0;
        if (from_y[q]) {
            float k0[8], k1[8];                                     // c = yf * k0 + k1
            {
                float gm[8], rs[8], bt[8], mn[8];
                dw3_load8(P.gm + k.c0, gm); dw3_load8(P.rs + k.c0, rs); dw3_load8(P.bt + k.c0, bt); dw3_load8(P.mn + k.c0, mn);
#pragma unroll
                for (int j = 0; j < 8; ++j) { k0[j] = 1.0f / (gm[j] * rs[j]); k1[j] = mn[j] - bt[j] * k0[j]; }
            }
            const bf16_t* fbase = P.yf + (long)k.b * P.yf_bs + k.c0 + (long)(g.tok0 + oy0 * P.Wo) * P.yf_rs;
            uint4 dq[4], fq[4];                                     // four outputs in flight per thread (8 independent 16-byte loads)
            for (int o0 = k.pl; o0 < npo; o0 += 4 * DW3_PL) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int o = min(o0 + u * DW3_PL, last);
                    dq[u] = ld_stream16(ybase + (long)o * P.y_rs);
                    fq[u] = ld_stream16(fbase + (long)o * P.yf_rs);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool on = o0 + u * DW3_PL < npo;
                    float d[8], f[8];
                    unpack8(dq[u], d); unpack8(fq[u], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float dj = on ? d[j] : 0.f;
                        acc[j] += dj;
                        acc[8 + j] = fmaf(dj, fmaf(f[j], k0[j], k1[j]), acc[8 + j]);
                    }
                }
            }
        } else {
            float w[9][8];
            dw3_load_taps(P.taps, g.C, k.c0, w);
            uint4 dq0 = make_uint4(0, 0, 0, 0), dq1 = dq0;
            if (npo > 0) {                                             // (block-uniform: a band past the last output row reads nothing)
                dq0 = *reinterpret_cast<const uint4*>(ybase + (long)min(k.pl, last) * P.y_rs);
                dq1 = *reinterpret_cast<const uint4*>(ybase + (long)min(k.pl + DW3_PL, last) * P.y_rs);
            }
            for (int o = k.pl; o < npo; o += DW3_PL) {
                const int oyl = o / P.Wo, ox = o - oyl * P.Wo;
                float d[8], c[8];
                const uint4 dcur = dq0;
                dq0 = dq1;
                dq1 = *reinterpret_cast<const uint4*>(ybase + (long)min(o + 2 * DW3_PL, last) * P.y_rs);
                unpack8(dcur, d);
                dw3_conv(dw3_tile, (oyl * P.stride * pitch + ox * P.stride) * 8 + k.ch, pitch8, w, c);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[j] += d[j]; acc[8 + j] = fmaf(d[j], c[j], acc[8 + j]); }
            }
        }
        dw3_reduce_small<16>(acc, red, row + (long)q * 2 * g.C, g.C, k.slice);
    }
}

// ------------------------------------------------------------------------------------------------------------------ dc + tap sums
// With coefficients (train mode): dy <- dc = a*dy + kb + kc*c in place (class rows untouched). Always: ws row of [nproj][10][C] =
// (G[t] = sum dc * x_t for the 9 taps, S = sum dc)
// (Round 4, measured and removed: c from the projection's forward output as in dw3_bwd_stats_kernel -- 72 FMAs and 18 LDS reads of tap weights less per
// output, one more 16-byte stream per projection. The window is unpacked for the tap sums either way, and the step was 0.18 ms SLOWER with it
// (42.26 -> 42.44 ms, three alternations on one box, scripts/r4/call13.sh): the pass is bound by its memory streams beside the weight-gradient stream,
// not by the convolution arithmetic.)
__global__ __launch_bounds__(DW3_THREADS, 2) void dw3_dc_taps_kernel(const Dw3Geo g, const Dw3P p0, const Dw3P p1, const Dw3P p2, float* __restrict__ ws) {
    CXR_PRIO_MAIN();
    __shared__ float red[DW3_RED_TAPS];
    __shared__ __attribute__((aligned(16))) float wl[3][12][64];      // per projection: 9 raw taps + (a, kb, kc) of this slice
    extern __shared__ uint4 dw3_tile[];
    const Dw3Blk k = dw3_block(g);
    const int pitch = g.W + 2, pitch8 = pitch * 8;
    dw3_stage_taps(wl, g, p0, p1, p2, k.slice, 3);
    dw3_stage(dw3_tile, g.x + (long)k.b * g.x_bs + (long)g.tok0 * g.x_rs + k.slice * 64, g.x_rs, g.H, g.W, k.band_i * g.band - 1, g.band + 2, -1, pitch);
    __syncthreads();
    float* row = ws + (long)(k.b * g.nbands + k.band_i) * g.nproj * 10 * g.C;
#pragma unroll 1
    for (int q = 0; q < g.nproj; ++q) {
        const Dw3P P = q == 0 ? p0 : (q == 1 ? p1 : p2);
        bf16_t* yb = P.y + (long)k.b * P.y_bs + k.c0;
        const bool train = P.aux != nullptr;
        const float* wq = &wl[q][0][k.ch * 8];
        float G[80];
#pragma unroll
        for (int i = 0; i < 80; ++i) G[i] = 0.f;
        int oy0;
        const int npo = dw3_band_outputs(g, P, k.band_i, oy0);
        // (gradient rows prefetched two iterations ahead, as in dw3_bwd_stats_kernel; every pixel is read and rewritten by its own iteration only)
        bf16_t* ybase = yb + (long)(g.tok0 + oy0 * P.Wo) * P.y_rs;
        const int last = npo > 0 ? npo - 1 : 0;
        uint4 dq0 = make_uint4(0, 0, 0, 0), dq1 = dq0;
        if (npo > 0) {                                             // (block-uniform: a band past the last output row reads nothing)
            dq0 = *reinterpret_cast<const uint4*>(ybase + (long)min(k.pl, last) * P.y_rs);
            dq1 = *reinterpret_cast<const uint4*>(ybase + (long)min(k.pl + DW3_PL, last) * P.y_rs);
        }
        for (int o = k.pl; o < npo; o += DW3_PL) {
            const int oyl = o / P.Wo, ox = o - oyl * P.Wo;
            const int base = (oyl * P.stride * pitch + ox * P.stride) * 8 + k.ch;
            bf16_t* dp = ybase + (long)o * P.y_rs;
            float d[8];
            const uint4 dcur = dq0;
            dq0 = dq1;
            dq1 = *reinterpret_cast<const uint4*>(ybase + (long)min(o + 2 * DW3_PL, last) * P.y_rs);
            unpack8(dcur, d);
            float f[9][8];                                       // the 3 x 3 input window, unpacked ONCE for the conv and the tap sums
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) unpack8(dw3_tile[base + ky * pitch8 + kx * 8], f[ky * 3 + kx]);
            if (train) {
                float c[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    float wt[8];
                    dw3_load8(wq + t * 64, wt);
#pragma unroll
                    for (int j = 0; j < 8; ++j) c[j] = fmaf(f[t][j], wt[j], c[j]);
                }
                float ca[8], cb[8], cc[8];
                dw3_load8(wq + 9 * 64, ca); dw3_load8(wq + 10 * 64, cb); dw3_load8(wq + 11 * 64, cc);
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = fmaf(ca[j], d[j], fmaf(cc[j], c[j], cb[j]));
                const uint4 pk = pack8(d);
                *reinterpret_cast<uint4*>(dp) = pk;
                unpack8(pk, d);                                  // the tap sums see the bf16 dc that dx and the GEMMs see
            }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) G[t * 8 + j] = fmaf(d[j], f[t][j], G[t * 8 + j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) G[72 + j] += d[j];
        }
        dw3_reduce_taps(G, red, row + (long)q * 10 * g.C, g.C, k.slice);
    }
}

// ------------------------------------------------------------------------------------------------------------------ input gradient
// w0: the taps of projection 0 in REGISTERS when it has stride 1 (the query projection: 9 of the ~13.5 taps a pixel gathers) -- from LDS each tap costs two
// ds_read_b128 per pixel on top of the data read; the stride-2 projections' taps (1-4 per pixel and projection) stay in LDS
template <int py, int px>
__device__ __forceinline__ void dw3_dx_class(const Dw3Geo& g, const Dw3P& p0, const Dw3P& p1, const Dw3P& p2, const uint4* dw3_tile, const float (*wl)[12][64],
                                             int toff1, int toff2, int nrow, int ch, bf16_t* drow, const float (&w0)[9][8], const bool w0_regs) {
    const int wave = threadIdx.x >> 6, lpl = (threadIdx.x & 63) >> 3;
    const int nr = (nrow - py + 1) >> 1, nc = (g.W - px + 1) >> 1;
    const int n = nr * nc;
    for (int grp = (wave - (py * 2 + px)) & 3; grp * 8 < n; grp += 4) {
        const int idx = grp * 8 + lpl;
        if (idx >= n) continue;
        const int ry = idx / nc, rx = idx - ry * nc;
        const int iyl = 2 * ry + py, ix = 2 * rx + px;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (w0_regs) {                                              // projection 0, stride 1, taps in registers (block-uniform)
            const int cols8 = (g.W + 2) * 8;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    float f[8];
                    unpack8(dw3_tile[(iyl + 2 - ky) * cols8 + (ix + 2 - kx) * 8 + ch], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] = fmaf(f[j], w0[ky * 3 + kx][j], acc[j]);
                }
        }
#pragma unroll 1
        for (int q = w0_regs ? 1 : 0; q < g.nproj; ++q) {
            const int stride = q == 0 ? p0.stride : (q == 1 ? p1.stride : p2.stride);
            const float* wq = &wl[q][0][ch * 8];
            const uint4* tile = dw3_tile + (q == 0 ? 0 : (q == 1 ? toff1 : toff2));
            const int cols8 = (stride == 1 ? g.W + 2 : g.W / 2 + 1) * 8;
            if (stride == 1) {
                // source (iy+1-ky, ix+1-kx) -> tile row iyl + 2 - ky, column ix + 2 - kx
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        float f[8], wt[8];
                        unpack8(tile[(iyl + 2 - ky) * cols8 + (ix + 2 - kx) * 8 + ch], f);
                        dw3_load8(wq + (ky * 3 + kx) * 64, wt);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(f[j], wt[j], acc[j]);
                    }
            } else {
                // band is even, so the parity of iy + 1 - ky is that of iyl + 1 - ky; tile row = (iyl + 1 - ky) / 2, column = (ix + 1 - kx) / 2
#pragma unroll
                for (int ky = (py ? 0 : 1); ky < 3; ky += 2)
#pragma unroll
                    for (int kx = (px ? 0 : 1); kx < 3; kx += 2) {
                        float f[8], wt[8];
                        unpack8(tile[((iyl + 1 - ky) >> 1) * cols8 + ((ix + 1 - kx) >> 1) * 8 + ch], f);
                        dw3_load8(wq + (ky * 3 + kx) * 64, wt);
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] = fmaf(f[j], wt[j], acc[j]);
                    }
            }
        }
        *reinterpret_cast<uint4*>(drow + (long)(iyl * g.W + ix) * g.x_rs) = pack8(acc);
    }
}


// dx[b,(iy,ix)] = sum_q sum_taps taps_q[ky][kx] * dc_q[b, ((iy+1-ky)/st, (ix+1-kx)/st)] (terms with a non-integer or out-of-range source
// vanish); class row: dx[b,0] = sum_q dc_q[b,0]. g.x is the OUTPUT here; each projection's gradient band is staged in LDS.
__global__ __launch_bounds__(DW3_THREADS, 2) void dw3_dx_kernel(const Dw3Geo g, const Dw3P p0, const Dw3P p1, const Dw3P p2) {
    CXR_PRIO_MAIN();
    __shared__ __attribute__((aligned(16))) float wl[3][12][64];      // taps of this slice (the class loops index them by compile-time tap)
    extern __shared__ uint4 dw3_tile[];
    const Dw3Blk k = dw3_block(g);
    // tile of projection q: stride 1 -> [band + 2][W + 2] starting at (band_i*band - 1, -1); stride 2 -> [band/2 + 1][W/2 + 1] at (band_i*band/2, 0)
    auto t_rows = [&](const Dw3P& P) { return P.stride == 1 ? g.band + 2 : g.band / 2 + 1; };
    auto t_cols = [&](const Dw3P& P) { return P.stride == 1 ? g.W + 2 : g.W / 2 + 1; };
    const int toff1 = t_rows(p0) * t_cols(p0) * 8, toff2 = toff1 + t_rows(p1) * t_cols(p1) * 8;
#pragma unroll 1
    for (int q = 0; q < g.nproj; ++q) {
        const Dw3P P = q == 0 ? p0 : (q == 1 ? p1 : p2);
        const int off = q == 0 ? 0 : (q == 1 ? toff1 : toff2);
        dw3_stage(dw3_tile + off, P.y + (long)k.b * P.y_bs + (long)g.tok0 * P.y_rs + k.slice * 64, P.y_rs, P.Ho, P.Wo,
                  P.stride == 1 ? k.band_i * g.band - 1 : (k.band_i * g.band) / 2, t_rows(P), P.stride == 1 ? -1 : 0, t_cols(P));
    }
    dw3_stage_taps(wl, g, p0, p1, p2, k.slice, 0);
    __syncthreads();
    bf16_t* db = const_cast<bf16_t*>(g.x) + (long)k.b * g.x_bs + k.c0;
    if (g.tok0 && k.band_i == 0 && threadIdx.x < 8) {
        for (int t = 0; t < g.tok0; ++t) {
            float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int q = 0; q < g.nproj; ++q) {
                const Dw3P P = q == 0 ? p0 : (q == 1 ? p1 : p2);
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(P.y + (long)k.b * P.y_bs + (long)t * P.y_rs + k.c0), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += f[j];
            }
            *reinterpret_cast<uint4*>(db + (long)t * g.x_rs) = pack8(a);
        }
    }
    int nrow = g.band;
    if (k.band_i * g.band + nrow > g.H) nrow = g.H - k.band_i * g.band;
    if (nrow <= 0) return;
    // A stride-2 projection reaches input pixel (iy, ix) only through the taps whose source (iy+1-ky, ix+1-kx) is even in both coordinates:
    // 1, 2, 2 or 4 of the 9, decided by the parities of iy and ix. Pixels are therefore processed per parity class with the class's tap list
    // compiled in (no per-lane predication: 2.25 taps per pixel on average instead of 9 masked ones); every wave serves all four classes, taking
    // every 4th group of 8 pixels with a class-dependent rotation so the waves stay balanced.
    bf16_t* drow = db + (long)(g.tok0 + k.band_i * g.band * g.W) * g.x_rs;
    float w0[9][8];
    const bool w0_regs = p0.stride == 1;
#pragma unroll
    for (int t = 0; t < 9; ++t) dw3_load8(&wl[0][t][k.ch * 8], w0[t]);
    dw3_dx_class<0, 0>(g, p0, p1, p2, dw3_tile, wl, toff1, toff2, nrow, k.ch, drow, w0, w0_regs);
    dw3_dx_class<0, 1>(g, p0, p1, p2, dw3_tile, wl, toff1, toff2, nrow, k.ch, drow, w0, w0_regs);
    dw3_dx_class<1, 0>(g, p0, p1, p2, dw3_tile, wl, toff1, toff2, nrow, k.ch, drow, w0, w0_regs);
    dw3_dx_class<1, 1>(g, p0, p1, p2, dw3_tile, wl, toff1, toff2, nrow, k.ch, drow, w0, w0_regs);
}

// ------------------------------------------------------------------------------------------------------------------ row reductions
struct Dw3Fin { const float* w; const float* g; const float* b; float* run_mean; float* run_var; float* mean; float* rstd; float* wf; float* sh; float count; };
struct Dw3Coef { const float* g; const float* mean; const float* rstd; float* dg; float* db; float* coef; float count; };
struct Dw3Taps { float* GS; float* dw; };

// column sums of two columns of ws [G][K] (16 row lanes x 64 columns per workgroup, 8 loads in flight per lane)
__device__ __forceinline__ void dw3_rows_sum2(const float* __restrict__ ws, int G, int K, int col0, int col1, float& s0, float& s1, float (*red)[64][2]) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float u0[4] = {0, 0, 0, 0}, u1[4] = {0, 0, 0, 0};
    int r = ty;
    for (; r + 3 * 16 < G; r += 4 * 16) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { u0[u] += ws[(long)(r + 16 * u) * K + col0]; u1[u] += ws[(long)(r + 16 * u) * K + col1]; }
    }
    for (; r < G; r += 16) { u0[0] += ws[(long)r * K + col0]; u1[0] += ws[(long)r * K + col1]; }
    red[ty][tx][0] = (u0[0] + u0[1]) + (u0[2] + u0[3]);
    red[ty][tx][1] = (u1[0] + u1[1]) + (u1[2] + u1[3]);
    __syncthreads();
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s0 += red[q][tx][0]; s1 += red[q][tx][1]; }
}

__global__ __launch_bounds__(1024) void dw3_reduce_finalize_kernel(const float* __restrict__ ws, int G, int C, int nproj, float eps, float momentum,
                                                                   Dw3Fin f0, Dw3Fin f1, Dw3Fin f2) {
    __shared__ float red[16][64][2];
    const int q = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63);
    const Dw3Fin P = q == 0 ? f0 : (q == 1 ? f1 : f2);
    const int K = nproj * 2 * C;
    float s0, s1;
    dw3_rows_sum2(ws, G, K, (q * 2 + 0) * C + c, (q * 2 + 1) * C + c, s0, s1, red);
    if ((threadIdx.x >> 6) != 0) return;
    const float mean = s0 / P.count;
    const float var = fmaxf(s1 / P.count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    P.mean[c] = mean; P.rstd[c] = rstd;
    if (momentum > 0.f) {
        P.run_mean[c] = (1.f - momentum) * P.run_mean[c] + momentum * mean;
        P.run_var[c] = (1.f - momentum) * P.run_var[c] + momentum * var * (P.count > 1.f ? P.count / (P.count - 1.f) : 1.f);
    }
    const float sc = P.g[c] * rstd;
#pragma unroll
    for (int t = 0; t < 9; ++t) P.wf[t * C + c] = P.w[c * 9 + t] * sc;
    P.sh[c] = P.b[c] - mean * sc;
}

__global__ __launch_bounds__(1024) void dw3_reduce_coef_kernel(const float* __restrict__ ws, int G, int C, int nproj, Dw3Coef f0, Dw3Coef f1, Dw3Coef f2) {
    __shared__ float red[16][64][2];
    const int q = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63);
    const Dw3Coef P = q == 0 ? f0 : (q == 1 ? f1 : f2);
    const int K = nproj * 2 * C;
    float S, D;
    dw3_rows_sum2(ws, G, K, (q * 2 + 0) * C + c, (q * 2 + 1) * C + c, S, D, red);
    if ((threadIdx.x >> 6) != 0) return;
    const float r = P.rstd[c], mu = P.mean[c];
    const float dgam = r * (D - mu * S);
    P.dg[c] += dgam;
    P.db[c] += S;
    const float a = P.g[c] * r, m1 = S / P.count, m2 = dgam / P.count;
    const float kc = -a * m2 * r;
    P.coef[c] = a; P.coef[C + c] = -a * m1 - kc * mu; P.coef[2 * C + c] = kc;
}

// GS[q] [10][C] <- column sums of ws [G][nproj][10][C]; dw[q] [C][9] += G rows when given (raw-tap gradient in parameter layout)
__global__ __launch_bounds__(1024) void dw3_reduce_taps_kernel(const float* __restrict__ ws, int G, int C, int nproj, Dw3Taps f0, Dw3Taps f1, Dw3Taps f2) {
    __shared__ float red[16][64];
    const int q = blockIdx.y, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + tx;                    // 0 .. 10*C (C % 64 == 0: no ragged block)
    const Dw3Taps P = q == 0 ? f0 : (q == 1 ? f1 : f2);
    const long K = (long)nproj * 10 * C;
    const float* src = ws + (long)q * 10 * C + col;
    float u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int r = ty;
    for (; r + 7 * 16 < G; r += 8 * 16) {
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] += src[(long)(r + 16 * i) * K];
    }
    for (; r < G; r += 16) u[0] += src[(long)r * K];
    red[ty][tx] = ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
    __syncthreads();
    if (ty != 0) return;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += red[i][tx];
    if (P.GS) P.GS[col] = s;
    const int t = col / C, c = col - t * C;
    if (P.dw && t < 9) P.dw[c * 9 + t] += s;
}

// ------------------------------------------------------------------------------------------------------------------ host side
struct Dw3Plan { Dw3Geo g; Dw3P p[3]; size_t lds; int grid; };

static int dw3_pick_band(int Bn, int H, int W, int nslices, size_t (*bytes)(int band, int W, const int* strides, int nproj), const int* strides, int nproj,
                         size_t static_lds, size_t cap) {
    // even band that fits the LDS budget; among those giving >= 384 workgroups (1.5 per CU) the one staging the fewest rows in total
    // (nbands * (band + 2)), otherwise the one giving the most workgroups
    {
        static int forced = -1;
        if (forced < 0) { const char* e = getenv("CXR_DW3_BAND"); forced = e ? atoi(e) : 0; }        // tuning aid: force the band height
        if (forced >= 2 && !(forced & 1) && bytes(forced, W, strides, nproj) + static_lds <= cap) return forced;
    }
    int best = 0, fallback = 0;
    long best_cost = 0;
    const int hmax = (H + 1) & ~1;
    for (int band = 2; band <= hmax; band += 2) {
        if (bytes(band, W, strides, nproj) + static_lds > cap) break;
        const int nb = (H + band - 1) / band;
        if (!fallback) fallback = band;
        if ((long)Bn * nb * nslices < 384) continue;
        const long cost = (long)nb * (band + 2);
        if (!best || cost <= best_cost) { best = band; best_cost = cost; }       // (equal cost: the taller band -- fewer, longer workgroups: stage 2 of CvT-21, 8 vs 10 rows, is 5-15 % faster per pass)
    }
    return best ? best : fallback;
}
static size_t dw3_bytes_x(int band, int W, const int*, int) { return (size_t)(band + 2) * (W + 2) * 128; }
static size_t dw3_bytes_dx(int band, int W, const int* strides, int nproj) {
    size_t b = 0;
    for (int q = 0; q < nproj; ++q) b += strides[q] == 1 ? (size_t)(band + 2) * (W + 2) * 128 : (size_t)(band / 2 + 1) * (W / 2 + 1) * 128;
    return b;
}

// which = 0: y/dy from cxr_dwproj.y, taps + shift (apply) | 1: raw taps (+ coef when present)
static int dw3_plan(Dw3Plan& pl, const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj, bool for_dx,
                    int red_floats) {
    if (Bn <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % 64) || nproj < 1 || nproj > 3 || !projs || !x || (x_rs % 8) || (x_bs % 8) || tok0 < 0) return CXR_ERR_ARG;
    int strides[3] = {1, 1, 1};
    for (int q = 0; q < nproj; ++q) {
        if (projs[q].stride != 1 && projs[q].stride != 2) return CXR_ERR_ARG;
        strides[q] = projs[q].stride;
    }
    Dw3Geo& g = pl.g;
    g.x = (const bf16_t*)x; g.x_bs = x_bs; g.x_rs = x_rs; g.Bn = Bn; g.C = C; g.H = H; g.W = W; g.tok0 = tok0; g.nproj = nproj;
    g.nslices = C / 64;
    g.band = dw3_pick_band(Bn, H, W, g.nslices, for_dx ? dw3_bytes_dx : dw3_bytes_x, strides, nproj, (size_t)red_floats * 4, DW3_LDS_CAP);
    if (g.band < 2) g.band = dw3_pick_band(Bn, H, W, g.nslices, for_dx ? dw3_bytes_dx : dw3_bytes_x, strides, nproj, (size_t)red_floats * 4, DW3_LDS_MAX);
    if (g.band < 2) return CXR_ERR_ARG;                     // a 2-row band does not fit the LDS budget: W too large for this kernel family
    g.nbands = (H + g.band - 1) / g.band;
    pl.lds = for_dx ? dw3_bytes_dx(g.band, W, strides, nproj) : dw3_bytes_x(g.band, W, strides, nproj);
    pl.grid = Bn * g.nbands * g.nslices;
    for (int q = 0; q < 3; ++q) {
        const cxr_dwproj& s = projs[q < nproj ? q : 0];
        Dw3P& d = pl.p[q];
        d.taps = s.taps; d.aux = nullptr; d.y = (bf16_t*)s.y; d.y_bs = s.y_bs; d.y_rs = s.y_rs; d.stride = s.stride; d.inv8 = 0.f;
        d.yf = nullptr; d.yf_bs = d.yf_rs = 0; d.gm = d.bt = d.mn = d.rs = nullptr;
        d.Ho = (H + 2 - 3) / s.stride + 1; d.Wo = (W + 2 - 3) / s.stride + 1;
        if (!s.taps) return CXR_ERR_ARG;
    }
    return CXR_OK;
}
static bool dw3_y_ok(const cxr_dwproj* projs, int nproj) {
    for (int q = 0; q < nproj; ++q)
        if (!projs[q].y || (projs[q].y_rs % 8) || (projs[q].y_bs % 8) || (((size_t)projs[q].y) % 16)) return false;
    return true;
}

template <typename K>
static void dw3_allow_big_lds(K kernel, bool& done) {
    if (!done) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DW3_LDS_MAX); done = true; }
}

}  // namespace

// scratch of the statistics / tap-sum passes, fp32 elements: rows = Bn * nbands <= Bn * ceil(H/2), widest row = 3 projections x 10 x C
extern "C" int cxr_dwproj_ws_floats(int Bn, int C, int H, int W) {
    (void)W;
    if (Bn <= 0 || C <= 0 || H <= 0) return CXR_ERR_ARG;
    const long n = (long)Bn * ((H + 1) / 2) * 3 * 10 * C;
    return n > 0x7fffffffL ? CXR_ERR_ARG : (int)n;
}

extern "C" int cxr_dwproj_apply_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                                     hipStream_t stream) {
    Dw3Plan pl;
    const int rc = dw3_plan(pl, x, x_bs, x_rs, Bn, C, H, W, tok0, projs, nproj, false, 0);
    if (rc) return rc;
    if (!dw3_y_ok(projs, nproj) || (((size_t)x) % 16)) return CXR_ERR_ARG;
    for (int q = 0; q < nproj; ++q) { if (!projs[q].shift) return CXR_ERR_ARG; pl.p[q].aux = projs[q].shift; }
    { static bool big = false; dw3_allow_big_lds(dw3_apply_kernel, big); }
    CXR_LAUNCH(dw3_apply_kernel, dim3(pl.grid), dim3(DW3_THREADS), pl.lds, stream, pl.g, pl.p[0], pl.p[1], pl.p[2]);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// the same pass with e4m3 outputs for consumers that are e4m3 GEMMs: projs[q].y is a 1-byte matrix (y_bs / y_rs in bytes), value * inv_scale[q]
extern "C" int cxr_dwproj_apply_q8(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                                   const float* inv_scale, hipStream_t stream) {
    Dw3Plan pl;
    const int rc = dw3_plan(pl, x, x_bs, x_rs, Bn, C, H, W, tok0, projs, nproj, false, 0);
    if (rc) return rc;
    if (!inv_scale || (((size_t)x) % 16)) return CXR_ERR_ARG;
    for (int q = 0; q < nproj; ++q) {
        if (!projs[q].shift || !projs[q].y || (projs[q].y_rs % 8) || (projs[q].y_bs % 8) || (((size_t)projs[q].y) % 8) || !(inv_scale[q] > 0.f)) return CXR_ERR_ARG;
        pl.p[q].aux = projs[q].shift; pl.p[q].inv8 = inv_scale[q];
    }
    { static bool big = false; dw3_allow_big_lds(dw3_apply_kernel, big); }
    CXR_LAUNCH(dw3_apply_kernel, dim3(pl.grid), dim3(DW3_THREADS), pl.lds, stream, pl.g, pl.p[0], pl.p[1], pl.p[2]);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dwproj_bn_train_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, float eps, float momentum,
                                              const cxr_dwproj* projs, int nproj, float* ws, hipStream_t stream) {
    Dw3Plan pl;
    const int rc = dw3_plan(pl, x, x_bs, x_rs, Bn, C, H, W, tok0, projs, nproj, false, DW3_RED_SMALL);
    if (rc) return rc;
    if (!ws || (((size_t)x) % 16)) return CXR_ERR_ARG;
    Dw3Fin f[3];
    for (int q = 0; q < 3; ++q) {
        const cxr_dwproj& s = projs[q < nproj ? q : 0];
        if (!s.w || !s.gamma || !s.beta || !s.mean || !s.rstd || !s.taps_out || !s.shift_out || (momentum > 0.f && (!s.run_mean || !s.run_var))) return CXR_ERR_ARG;
        f[q] = Dw3Fin{s.w, s.gamma, s.beta, s.run_mean, s.run_var, s.mean, s.rstd, s.taps_out, s.shift_out, (float)((long)Bn * pl.p[q].Ho * pl.p[q].Wo)};
    }
    { static bool big = false; dw3_allow_big_lds(dw3_stats_kernel, big); }
    CXR_LAUNCH(dw3_stats_kernel, dim3(pl.grid), dim3(DW3_THREADS), pl.lds, stream, pl.g, pl.p[0], pl.p[1], pl.p[2], ws);
    CXR_LAUNCH(dw3_reduce_finalize_kernel, dim3(C / 64, nproj), dim3(1024), 0, stream, ws, Bn * pl.g.nbands, C, nproj, eps, momentum, f[0], f[1], f[2]);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dwproj_bn_train_bwd_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs,
                                                  int nproj, float* ws, hipStream_t stream) {
    Dw3Plan pl;
    const int rc = dw3_plan(pl, x, x_bs, x_rs, Bn, C, H, W, tok0, projs, nproj, false, DW3_RED_SMALL);
    if (rc) return rc;
    if (!ws || !dw3_y_ok(projs, nproj) || (((size_t)x) % 16)) return CXR_ERR_ARG;
    Dw3Coef f[3];
    for (int q = 0; q < 3; ++q) {
        const cxr_dwproj& s = projs[q < nproj ? q : 0];
        if (!s.gamma || !s.mean || !s.rstd || !s.dgamma || !s.dbeta || !s.coef) return CXR_ERR_ARG;
        f[q] = Dw3Coef{s.gamma, s.mean, s.rstd, s.dgamma, s.dbeta, s.coef, (float)((long)Bn * pl.p[q].Ho * pl.p[q].Wo)};
        if (q < nproj && s.yf && s.beta) {                          // forward output handed over: c from it instead of the recomputed convolution
            if ((s.yf_rs % 8) || (s.yf_bs % 8) || (((size_t)s.yf) % 16)) return CXR_ERR_ARG;
            pl.p[q].yf = (const bf16_t*)s.yf; pl.p[q].yf_bs = s.yf_bs; pl.p[q].yf_rs = s.yf_rs;
            pl.p[q].gm = s.gamma; pl.p[q].bt = s.beta; pl.p[q].mn = s.mean; pl.p[q].rs = s.rstd;
        }
    }
    { static bool big = false; dw3_allow_big_lds(dw3_bwd_stats_kernel, big); }
    CXR_LAUNCH(dw3_bwd_stats_kernel, dim3(pl.grid), dim3(DW3_THREADS), pl.lds, stream, pl.g, pl.p[0], pl.p[1], pl.p[2], ws);
    CXR_LAUNCH(dw3_reduce_coef_kernel, dim3(C / 64, nproj), dim3(1024), 0, stream, ws, Bn * pl.g.nbands, C, nproj, f[0], f[1], f[2]);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dwproj_dc_taps_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                                       float* ws, hipStream_t stream) {
    Dw3Plan pl;
    const int rc = dw3_plan(pl, x, x_bs, x_rs, Bn, C, H, W, tok0, projs, nproj, false, DW3_RED_TAPS + 3 * 12 * 64);
    if (rc) return rc;
    if (!ws || !dw3_y_ok(projs, nproj) || (((size_t)x) % 16)) return CXR_ERR_ARG;
    Dw3Taps f[3];
    for (int q = 0; q < 3; ++q) {
        const cxr_dwproj& s = projs[q < nproj ? q : 0];
        if (!s.GS && !s.dw) return CXR_ERR_ARG;
        pl.p[q].aux = s.coef;                                // null = eval mode: dc = dy, not rewritten
        f[q] = Dw3Taps{s.GS, s.dw};
    }
    { static bool big = false; dw3_allow_big_lds(dw3_dc_taps_kernel, big); }
    CXR_LAUNCH(dw3_dc_taps_kernel, dim3(pl.grid), dim3(DW3_THREADS), pl.lds, stream, pl.g, pl.p[0], pl.p[1], pl.p[2], ws);
    CXR_LAUNCH(dw3_reduce_taps_kernel, dim3(10 * C / 64, nproj), dim3(1024), 0, stream, ws, Bn * pl.g.nbands, C, nproj, f[0], f[1], f[2]);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dwproj_dx_bf16(void* dx, long dx_bs, long dx_rs, int Bn, int C, int H, int W, int tok0, const cxr_dwproj* projs, int nproj,
                                  hipStream_t stream) {
    Dw3Plan pl;
    const int rc = dw3_plan(pl, dx, dx_bs, dx_rs, Bn, C, H, W, tok0, projs, nproj, true, 3 * 12 * 64);
    if (rc) return rc;
    if (!dw3_y_ok(projs, nproj) || (((size_t)dx) % 16)) return CXR_ERR_ARG;
    { static bool big = false; dw3_allow_big_lds(dw3_dx_kernel, big); }
    CXR_LAUNCH(dw3_dx_kernel, dim3(pl.grid), dim3(DW3_THREADS), pl.lds, stream, pl.g, pl.p[0], pl.p[1], pl.p[2]);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// Raw depthwise taps of many projections from the parameter layout [C, 9] to the conv layout [9, C] the kernels above read, in one launch
// (63 projections per CvT-21; re-run once per weight version). table (device) int64 [n][4] = {src, dst, C, first_block}; one workgroup per
// 256 consecutive elements of a destination.
__global__ __launch_bounds__(256) void dw3_taps_layout_kernel(const long* __restrict__ table, int n) {
    int e = 0;
    while (e + 1 < n && (long)blockIdx.x >= table[(e + 1) * 4 + 3]) ++e;
    const float* src = reinterpret_cast<const float*>(table[e * 4 + 0]);
    float* dst = reinterpret_cast<float*>(table[e * 4 + 1]);
    const int C = (int)table[e * 4 + 2];
    const int i = ((int)blockIdx.x - (int)table[e * 4 + 3]) * 256 + threadIdx.x;      // destination index t*C + c
    if (i < 9 * C) {
        const int t = i / C, c = i - t * C;
        dst[i] = src[c * 9 + t];
    }
}

extern "C" int cxr_dwproj_taps_layout(const long* table, int n, int total_blocks, hipStream_t stream) {
    if (!table || n <= 0 || total_blocks <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(dw3_taps_layout_kernel, dim3(total_blocks), dim3(256), 0, stream, table, n);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
