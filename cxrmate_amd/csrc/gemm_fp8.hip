// FP8 (OCP e4m3fn) linear layers of the FROZEN encoder (BASELINE.json configs[4]: "fp8 MFMA encoder"; caller: the reference's
// gen-prompt SCST, modules/lightning_modules/longitudinal/scst/gen_prompt.py:174-259, whose encoder runs without gradients).
//
//   C[M,N] = epi( scale * A8[M,K] . W8[N,K]^T )      A8, W8 e4m3 with one scale per tensor (scale = s_A * s_W), fp32 accumulation
//
// on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: the block-scaled form is the one that runs e4m3 at twice the bf16 rate on gfx950
// (MI355X_MICROARCH.md "Matrix cores": the non-scaled fp8 MFMA runs at the bf16 rate). Operand map, probed with exact integer data
// (scripts/lab/mfma_fp8_probe.hip): lane l holds 32 k-bytes of row / column l & 31; the products pair byte j of lane half l >> 5 of A with the
// same byte of the same half of B, so ANY k assignment works as long as both operands use it: here lane half h takes k = 32 h .. 32 h + 31 of
// the 64-wide step, two 16-byte LDS reads. C/D: column = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5).
//
// Tile 128 x BN (BN = 128 / 64) x 64, 256 threads = 2 x 2 waves of 64 x BN/2, register-staged double buffering (the next tile's global loads
// are issued before the MFMAs of the current one). K % 64 == 0, lda / ldw % 16 == 0. Rows past M / N are clamped on the load side (no
// conditional loads: hipcc parks s_waitcnt vmcnt(0) behind a guarded load) and dropped on the store side.
// Epilogue: bias, GELU, residual (bf16), output as bf16 and / or as e4m3 with its own scale (the next fp8 layer's input: no quantisation pass).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct Fp8Args {
    const uint8_t* A; long lda;
    const uint8_t* W; long ldw;
    bf16_t* C; long ldc;                 // bf16 output or null
    uint8_t* C8; long ldc8; float c8_inv; // e4m3 output (value / its scale) or null
    const float* bias; const bf16_t* residual; long ldr;
    int M, N, K, act;
    float scale;
    const float* row_scale; int rs_rows, rs_after;   // per-image DropPath factor row_scale[m / rs_rows]: on the branch (before the residual) or on the sum
    int vec_epilogue;                    // N, ldc, ldc8, ldr multiples of 8 and 16-byte aligned bases: rows are written in 16-byte chunks through LDS
};

constexpr int LDSROW = 80;               // 64 k-bytes + 16 of padding: the 16-byte fragment reads of a wave spread over all banks

// (pack_fp8x4 / to_fp8: common.h)

template <int BN>
__global__ __launch_bounds__(256) void gemm_fp8_kernel(const Fp8Args g) {
    constexpr int TN = BN / 64;                                   // 32-column tiles per wave
    __shared__ __attribute__((aligned(16))) uint8_t smem[2 * (128 + BN) * LDSROW];     // 40 KB / 30 KB; re-used by the epilogue
    uint8_t (*As)[128 * LDSROW] = reinterpret_cast<uint8_t (*)[128 * LDSROW]>(smem);
    uint8_t (*Ws)[BN * LDSROW] = reinterpret_cast<uint8_t (*)[BN * LDSROW]>(smem + 2 * 128 * LDSROW);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order: consecutive workgroup ids land on different XCDs (each with its own L2); give every XCD a contiguous run of row
    // panels so that the W panel and the A rows it re-reads stay in ONE L2
    const int nbn = (g.N + BN - 1) / BN, nbm = (g.M + 127) / 128, ntile = nbn * nbm;
    int wg = blockIdx.x;
    {
        const int per = (ntile + 7) / 8, x = wg & 7, i = wg >> 3;
        const int t = x * per + i;
        if (i >= per || t >= ntile) return;
        wg = t;
    }
    const int bm = wg / nbn, bn = wg % nbn;
    const int m0 = bm * 128, n0 = bn * BN;
    // staging map: the tile's rows are 64 bytes = 4 chunks of 16; thread t moves chunks t and t + 256 of A (and of W when BN = 128)
    const int r0 = tid >> 2, part = (tid & 3) * 16;
    const int am0 = min(m0 + r0, g.M - 1), am1 = min(m0 + r0 + 64, g.M - 1);
    const int wn0 = min(n0 + r0, g.N - 1), wn1 = min(n0 + r0 + 64, g.N - 1);
    const uint8_t* pa0 = g.A + (long)am0 * g.lda + part; const uint8_t* pa1 = g.A + (long)am1 * g.lda + part;
    const uint8_t* pw0 = g.W + (long)wn0 * g.ldw + part; const uint8_t* pw1 = g.W + (long)wn1 * g.ldw + part;
    uint4 ra0, ra1, rw0, rw1;
    ra0 = *reinterpret_cast<const uint4*>(pa0); ra1 = *reinterpret_cast<const uint4*>(pa1);
    rw0 = *reinterpret_cast<const uint4*>(pw0);
    if (BN == 128) rw1 = *reinterpret_cast<const uint4*>(pw1);
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int KT = g.K / 64;
    *reinterpret_cast<uint4*>(&As[0][r0 * LDSROW + part]) = ra0;
    *reinterpret_cast<uint4*>(&As[0][(r0 + 64) * LDSROW + part]) = ra1;
    *reinterpret_cast<uint4*>(&Ws[0][r0 * LDSROW + part]) = rw0;
    if (BN == 128) *reinterpret_cast<uint4*>(&Ws[0][(r0 + 64) * LDSROW + part]) = rw1;
    __syncthreads();
    const int fr = lane & 31, fh = (lane >> 5) * 32;
    for (int kt = 0; kt < KT; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < KT) {                                        // (uniform branch) next tile: global -> registers while this one multiplies
            const long ko = (long)(kt + 1) * 64;
            ra0 = *reinterpret_cast<const uint4*>(pa0 + ko); ra1 = *reinterpret_cast<const uint4*>(pa1 + ko);
            rw0 = *reinterpret_cast<const uint4*>(pw0 + ko);
            if (BN == 128) rw1 = *reinterpret_cast<const uint4*>(pw1 + ko);
        }
        i32x8 af[2], bf[TN];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint8_t* p = &As[cur][(wm * 64 + i * 32 + fr) * LDSROW + fh];
            const uint4 lo = *reinterpret_cast<const uint4*>(p), hi = *reinterpret_cast<const uint4*>(p + 16);
            af[i] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const uint8_t* p = &Ws[cur][(wn * (BN / 2) + j * 32 + fr) * LDSROW + fh];
            const uint4 lo = *reinterpret_cast<const uint4*>(p), hi = *reinterpret_cast<const uint4*>(p + 16);
            bf[j] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[i], bf[j], acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        if (kt + 1 < KT) {
            const int nx = cur ^ 1;
            *reinterpret_cast<uint4*>(&As[nx][r0 * LDSROW + part]) = ra0;
            *reinterpret_cast<uint4*>(&As[nx][(r0 + 64) * LDSROW + part]) = ra1;
            *reinterpret_cast<uint4*>(&Ws[nx][r0 * LDSROW + part]) = rw0;
            if (BN == 128) *reinterpret_cast<uint4*>(&Ws[nx][(r0 + 64) * LDSROW + part]) = rw1;
        }
        __syncthreads();
    }
    // ---- epilogue. The accumulators hold a COLUMN per lane (16 rows in the registers): written as they are, every store instruction would
    // touch 32 rows x 64 bytes. Each wave instead turns its tile through LDS (fp32, 32 rows at a time, in the staging buffers the K loop is
    // done with) and writes whole 16-byte chunks of rows; the residual is read the same way.
    const int h4 = (lane >> 5) * 4;
    constexpr int WCOLS = BN / 2;                                  // columns of a wave's tile
    if (g.vec_epilogue) {
        float* T = reinterpret_cast<float*>(smem) + wave * (32 * WCOLS);           // 4 waves x 32 rows x 64 (32) columns x 4 B = 32 (16) KB
        float bias[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) bias[j] = g.bias ? g.bias[min(n0 + wn * WCOLS + j * 32 + fr, g.N - 1)] : 0.f;
        constexpr int CPR = WCOLS / 8;                             // 8-column chunks per row
        constexpr int RPI = 64 / CPR;                              // rows per pass of the wave
        const int cr = lane / CPR, cc = (lane % CPR) * 8;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            // residual chunks and DropPath factors of this half's passes: issued up front, unconditional (clamped row / column) -- inside the
            // per-lane `m < M && n < N` guard below every one of them would be a branch + load + s_waitcnt vmcnt(0)
            uint4 rres[32 / RPI];
            float rsv[32 / RPI];
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int mc = min(m0 + wm * 64 + i * 32 + it * RPI + cr, g.M - 1), nc = min(n0 + wn * WCOLS + cc, g.N - 8);
                rres[it] = make_uint4(0, 0, 0, 0);
                if (g.residual) rres[it] = *reinterpret_cast<const uint4*>(g.residual + (long)mc * g.ldr + nc);      // (uniform branch)
                rsv[it] = g.row_scale ? g.row_scale[(unsigned)mc / (unsigned)g.rs_rows] : 1.f;
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float v = acc[i][j][e] * g.scale + bias[j];
                    if (g.act == 1) v = gelu_f(v);
                    T[((e & 3) + 8 * (e >> 2) + h4) * WCOLS + j * 32 + fr] = v;
                }
            // (a wave reads only what it wrote itself: no workgroup barrier, the LDS counter wait the compiler inserts is enough)
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int rr = it * RPI + cr;
                const int m = m0 + wm * 64 + i * 32 + rr, n = n0 + wn * WCOLS + cc;
                const float4 lo = *reinterpret_cast<const float4*>(&T[rr * WCOLS + cc]), hi = *reinterpret_cast<const float4*>(&T[rr * WCOLS + cc + 4]);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (m < g.M && n < g.N) {
                    const float rs = rsv[it];
                    if (!g.rs_after) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] *= rs;
                    }
                    if (g.residual) {
                        float r[8];
                        unpack8(rres[it], r);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += r[q];
                    }
                    if (g.rs_after) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] *= rs;
                    }
                    if (g.C) *reinterpret_cast<uint4*>(g.C + (long)m * g.ldc + n) = pack8(v);
                    if (g.C8) {
                        const float ci = g.c8_inv;
                        *reinterpret_cast<uint2*>(g.C8 + (long)m * g.ldc8 + n) =
                            make_uint2(pack_fp8x4(v[0] * ci, v[1] * ci, v[2] * ci, v[3] * ci), pack_fp8x4(v[4] * ci, v[5] * ci, v[6] * ci, v[7] * ci));
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + fr;
        const bool n_ok = n < g.N;
        const int nc = n_ok ? n : g.N - 1;
        const float bias = g.bias ? g.bias[nc] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + h4;
                if (!n_ok || m >= g.M) continue;
                float v = acc[i][j][e] * g.scale + bias;
                if (g.act == 1) v = gelu_f(v);
                const float rs = g.row_scale ? g.row_scale[m / g.rs_rows] : 1.f;
                if (!g.rs_after) v *= rs;
                if (g.residual) v += bf2f(g.residual[(long)m * g.ldr + n]);
                if (g.rs_after) v *= rs;
                if (g.C) g.C[(long)m * g.ldc + n] = f2bf(v);
                if (g.C8) g.C8[(long)m * g.ldc8 + n] = to_fp8(v * g.c8_inv);
            }
        }
    }
}

// bf16 [M,K] -> e4m3 [M,K] (x * inv_scale, saturating); 16 elements per thread
__global__ __launch_bounds__(256) void quantize_fp8_kernel(const bf16_t* __restrict__ x, long ldx, uint8_t* __restrict__ out, long ldo, int M, int K,
                                                           float inv) {
    const int kc = K / 16;
    const long total = (long)M * kc;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / kc; const int c = (int)(i % kc) * 16;
        const uint4 a = *reinterpret_cast<const uint4*>(x + m * ldx + c), b = *reinterpret_cast<const uint4*>(x + m * ldx + c + 8);
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t w0 = w[2 * q], w1 = w[2 * q + 1];
            o[q] = pack_fp8x4(__uint_as_float(w0 << 16) * inv, __uint_as_float(w0 & 0xffff0000u) * inv, __uint_as_float(w1 << 16) * inv,
                              __uint_as_float(w1 & 0xffff0000u) * inv);
        }
        *reinterpret_cast<uint4*>(out + m * ldo + c) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

}  // namespace

extern "C" int cxr_gemm_nt_fp8(const void* A8, long lda, const void* W8, long ldw, void* C, long ldc, void* C8, long ldc8, float c8_inv_scale,
                               int M, int N, int K, float scale, const float* bias, const void* residual, long ldr, int act,
                               const float* row_scale, int rs_rows, int rs_after, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K % 64) || (lda % 16) || (ldw % 16) || ((uintptr_t)A8 % 16) || ((uintptr_t)W8 % 16) || (!C && !C8) ||
        (act != 0 && act != 1) || (row_scale && rs_rows <= 0))
        return CXR_ERR_ARG;
    Fp8Args g{(const uint8_t*)A8, lda, (const uint8_t*)W8, ldw, (bf16_t*)C, ldc, (uint8_t*)C8, ldc8, c8_inv_scale, bias, (const bf16_t*)residual, ldr,
              M, N, K, act, scale, row_scale, rs_rows > 0 ? rs_rows : 1, rs_after, 0};
    g.vec_epilogue = (N % 8 == 0) && (!C || (ldc % 8 == 0 && (uintptr_t)C % 16 == 0)) && (!C8 || (ldc8 % 8 == 0 && (uintptr_t)C8 % 8 == 0)) &&
                     (!residual || (ldr % 8 == 0 && (uintptr_t)residual % 16 == 0));
    // 128 x 64 tiles where 128-wide ones would waste half a tile or leave CUs idle
    const bool narrow = (N % 128 != 0 && N % 128 <= 64) || ((long)cdiv(M, 128) * cdiv(N, 128) < 384);
    const int bn = narrow ? 64 : 128;
    const int ntile = cdiv(M, 128) * cdiv(N, bn);
    const int grid = cdiv(ntile, 8) * 8;
    if (narrow) CXR_LAUNCH(gemm_fp8_kernel<64>, dim3(grid), dim3(256), 0, stream, g);
    else CXR_LAUNCH(gemm_fp8_kernel<128>, dim3(grid), dim3(256), 0, stream, g);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_quantize_fp8(const void* x, long ldx, void* out, long ldo, int M, int K, float inv_scale, hipStream_t stream) {
    if (M <= 0 || K <= 0 || (K % 16) || (ldx % 8) || (ldo % 16) || ((uintptr_t)x % 16) || ((uintptr_t)out % 16)) return CXR_ERR_ARG;
    const long total = (long)M * (K / 16);
    const int grid = (int)(cdiv(total, 256) < 4096 ? cdiv(total, 256) : 4096);
    CXR_LAUNCH(quantize_fp8_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, ldx, (uint8_t*)out, ldo, M, K, inv_scale);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
