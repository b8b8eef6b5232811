// Convolutional pieces of the CvT encoder, all on token-major (NHWC) bf16 activations:
//   * im2col for the three patch-embedding convs (7x7 s4 p2 on NCHW fp32 pixels; 3x3 s2 p1 on tokens) -> MFMA GEMM
//   * col2im (gather form, no atomics) for the embedding convs' input gradient
//   * depthwise 3x3 conv + folded BatchNorm producing q (stride 1) or k AND v (stride 2, one read of the input)
//   * its backward (input gradient + per-channel tap/shift gradient sums)
// These are HBM-bound: channel-contiguous 16-byte accesses, taps re-read through L1/L2.
#include "common.h"

// ---------------------------------------------------------------------------------------------- stage-1 im2col
// pixels [Bn, Cin, H, W] fp32 NCHW  ->  col [Bn*Ho*Wo, Kpad] bf16, k = c*KS*KS + ky*KS + kx (= weight.view(Cout,-1)), zero padded
__global__ __launch_bounds__(256) void im2col_nchw_kernel(const float* __restrict__ px, bf16_t* __restrict__ col, int Bn, int Cin, int H, int W,
                                                          int KS, int stride, int pad, int Ho, int Wo, int Kreal, int Kpad) {
    const int chunks = Kpad / 8;
    const long total = (long)Bn * Ho * Wo * chunks;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ch = (int)(idx % chunks);
        const long pix = idx / chunks;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = ch * 8 + j;
            float v = 0.f;
            if (k < Kreal) {
                const int c = k / (KS * KS), rem = k % (KS * KS), ky = rem / KS, kx = rem % KS;
                const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = px[(((long)b * Cin + c) * H + iy) * W + ix];
            }
            o[j] = v;
        }
        *reinterpret_cast<uint4*>(col + pix * Kpad + ch * 8) = pack8(o);
    }
}

// The same with ONE workgroup per output row (b, oy): the Cin * KS input rows that row's patches touch are staged in LDS with coalesced 16-byte
// loads (zero borders included), the col entries are then assembled from LDS through a k -> LDS-offset table (no per-element division, no
// scattered 4-byte global gathers: the gather kernel above ran at 2 TB/s of fabric traffic, 2.6x its unique input bytes).
__global__ __launch_bounds__(256) void im2col_nchw_rows_kernel(const float* __restrict__ px, bf16_t* __restrict__ col, int Cin, int H, int W, int KS,
                                                               int stride, int pad, int Ho, int Wo, int Kreal, int Kpad) {
    extern __shared__ __attribute__((aligned(16))) float im_rows[];          // [Cin * KS][Wp], Wp = W + 2 * pad, then the table int[Kpad]
    const int Wp = W + 2 * pad, nrows = Cin * KS;
    int* tbl = reinterpret_cast<int*>(im_rows + (long)nrows * Wp);
    const int tid = threadIdx.x;
    const int oy = blockIdx.x % Ho, b = blockIdx.x / Ho;
    for (int k = tid; k < Kpad; k += 256) {
        int off = -1;
        if (k < Kreal) { const int c = k / (KS * KS), rem = k % (KS * KS); off = (c * KS + rem / KS) * Wp + rem % KS; }
        tbl[k] = off;
    }
    // borders of every staged row
    for (int e = tid; e < nrows * 2 * pad; e += 256) {
        const int r = e / (2 * pad), j = e % (2 * pad);
        im_rows[r * Wp + (j < pad ? j : W + j)] = 0.f;
    }
    const bool vec = (W % 4) == 0 && (pad % 2) == 0 && (((size_t)px) % 16) == 0;
    if (vec) {
        const int w4 = W / 4;
        for (int e = tid; e < nrows * w4; e += 256) {
            const int r = e / w4, x4 = (e % w4) * 4;
            const int c = r / KS, ky = r % KS, iy = oy * stride - pad + ky;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (iy >= 0 && iy < H) v = *reinterpret_cast<const float4*>(px + (((long)b * Cin + c) * H + iy) * W + x4);
            float* d = im_rows + r * Wp + pad + x4;                         // (8-byte aligned: Wp and pad even)
            *reinterpret_cast<float2*>(d) = make_float2(v.x, v.y);
            *reinterpret_cast<float2*>(d + 2) = make_float2(v.z, v.w);
        }
    } else {
        for (int e = tid; e < nrows * W; e += 256) {
            const int r = e / W, x = e % W;
            const int c = r / KS, ky = r % KS, iy = oy * stride - pad + ky;
            im_rows[r * Wp + pad + x] = (iy >= 0 && iy < H) ? px[(((long)b * Cin + c) * H + iy) * W + x] : 0.f;
        }
    }
    __syncthreads();
    const int chunks = Kpad / 8;
    bf16_t* out = col + ((long)b * Ho + oy) * Wo * Kpad;
    for (int e = tid; e < Wo * chunks; e += 256) {
        const int ox = e / chunks, ch = e % chunks;
        const float* base = im_rows + ox * stride;
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int off = tbl[ch * 8 + j];
            o[j] = off >= 0 ? base[off] : 0.f;
        }
        *reinterpret_cast<uint4*>(out + (long)e * 8) = pack8(o);
    }
}

// tokens [Bn, tok_rs rows..] bf16, spatial token (y,x) at row y*W+x  ->  col [Bn*Ho*Wo, 9*Cin], k = (ky*3+kx)*Cin + c
__global__ __launch_bounds__(256) void im2col_tok_kernel(const bf16_t* __restrict__ x, long x_bs, long x_rs, bf16_t* __restrict__ col,
                                                         int Bn, int Cin, int H, int W, int stride, int pad, int Ho, int Wo) {
    const int cch = Cin / 8, chunks = 9 * cch;
    const long total = (long)Bn * Ho * Wo * chunks;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ch = (int)(idx % chunks);
        const long pix = idx / chunks;
        const int tap = ch / cch, c8 = (ch % cch) * 8, ky = tap / 3, kx = tap % 3;
        const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
        const int iy = oy * stride - pad + ky, ix = ox * stride - pad + kx;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const uint4*>(x + (long)b * x_bs + ((long)iy * W + ix) * x_rs + c8);
        *reinterpret_cast<uint4*>(col + pix * (long)(9 * Cin) + tap * Cin + c8) = v;
    }
}

// gather-form col2im: dx[b,(iy,ix),c] = sum over taps/out positions covering (iy,ix) of dcol[b,(oy,ox), tap, c]
__global__ __launch_bounds__(256) void col2im_tok_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dx, long dx_bs, long dx_rs,
                                                         int Bn, int Cin, int H, int W, int stride, int pad, int Ho, int Wo) {
    const int cch = Cin / 8;
    const long total = (long)Bn * H * W * cch;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % cch) * 8;
        const long pix = idx / cch;
        const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + pad - ky;
            if (ty < 0 || (ty % stride) || ty / stride >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + pad - kx;
                if (tx < 0 || (tx % stride) || tx / stride >= Wo) continue;
                const long opix = ((long)b * Ho + ty / stride) * Wo + tx / stride;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(dcol + opix * (long)(9 * Cin) + (ky * 3 + kx) * Cin + c8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
        }
        *reinterpret_cast<uint4*>(dx + (long)b * dx_bs + ((long)iy * W + ix) * dx_rs + c8) = pack8(acc);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Round 4: the stage-1 patch embedding as ONE kernel -- Conv2d(3, 64, kernel 7, stride 4, padding 2) on the fp32 NCHW pixels as an implicit GEMM on the
// matrix cores, + bias, + LayerNorm(64) over the channels in the epilogue (TF5 modeling_cvt.py:77-90: projection -> flatten -> LayerNorm).
// Replaces im2col (147 us: 226 MB written, read again by the GEMM) + GEMM + LayerNorm launch (three passes over the 75 MB token map).
//   workgroup  = (image, band of PE_RB output rows): the 4 PE_RB + 3 input rows of all three channels are staged ONCE in LDS as bf16, two zero
//                columns on each side, so that the 8 horizontal taps kx = 0..7 of output pixel ox start at element 4 ox of a row (8-byte aligned).
//   K order    = (c, ky) group-major with kx padded from 7 to 8 (the 8th tap has a zero weight): one 8-element MFMA operand fragment of a pixel is
//                ONE contiguous 16-byte piece of one staged row -- no gather, no per-element address arithmetic. K = 21 groups x 8 = 168 -> 192.
//   MFMA       = D^T[channel, pixel] = W'[channel, k] . X^T[k, pixel] (v_mfma_f32_16x16x32_bf16): the packed weights (64 x 192, 24 KB) live in
//                registers as A fragments; a lane ends up with 4 x 4 CONSECUTIVE channels of ONE pixel, so the LayerNorm statistics of a pixel are 16
//                in-lane adds and two cross-lane steps.
// Written: y = LayerNorm(e) (bf16 token-major), and for a training forward e = conv + bias (bf16, the LayerNorm backward's input) + (mean, rstd).
constexpr int PE_RB = 4;                     // output rows per workgroup
constexpr int PE_WS = 392;                   // LDS row stride (bf16 elements): 2 + 384 + 2 zero columns, padded to a multiple of 8

__global__ __launch_bounds__(256) void patch_embed_s1_kernel(const float* __restrict__ px, const bf16_t* __restrict__ Wpk, const float* __restrict__ bias,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                             bf16_t* __restrict__ e_out, bf16_t* __restrict__ y_out, float* __restrict__ stats,
                                                             int H, int W, int Ho, int Wo) {
    constexpr int NR = 4 * PE_RB + 3;
    __shared__ __attribute__((aligned(16))) bf16_t img[3 * NR * PE_WS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nbands = (Ho + PE_RB - 1) / PE_RB;
    const int b = blockIdx.x / nbands, oy0 = (blockIdx.x % nbands) * PE_RB;
    const int iy0 = 4 * oy0 - 2;
    // ---- the packed weights of this lane: A fragments, channel tile ct, k-step s: lane (m = lane & 15, kg = lane >> 4) holds W'[16 ct + m][32 s + 8 kg ..]
    const int fm = lane & 15, fg = lane >> 4;
    bf16x8_t wf[4][6];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) wf[ct][s2] = *reinterpret_cast<const bf16x8_t*>(Wpk + (long)(16 * ct + fm) * 192 + 32 * s2 + 8 * fg);
    // ---- stage the band: fp32 rows -> bf16 LDS rows (zero outside the image), 16 bytes of pixels per thread and step
    const int w4 = W / 4;
    for (int e = tid; e < 3 * NR * w4; e += 256) {
        const int r = e / w4, x4 = (e - r * w4) * 4;
        const int c = r / NR, rr = r - c * NR, iy = iy0 + rr;
        const bool in = iy >= 0 && iy < H;
        const float4 v = *reinterpret_cast<const float4*>(px + (((long)b * 3 + c) * H + (in ? iy : 0)) * W + x4);      // (unconditional: clamped row)
        uint32_t* d = reinterpret_cast<uint32_t*>(img + r * PE_WS + 2 + x4);                                           // (4-byte aligned)
        d[0] = in ? pack2bf(v.x, v.y) : 0u;
        d[1] = in ? pack2bf(v.z, v.w) : 0u;
    }
    for (int r = tid; r < 3 * NR; r += 256) {                      // the zero columns left and right of every row
        *reinterpret_cast<uint32_t*>(img + r * PE_WS) = 0u;
#pragma unroll
        for (int j = 0; j < (PE_WS - 386) / 2; ++j) *reinterpret_cast<uint32_t*>(img + r * PE_WS + 2 + 384 + 2 * j) = 0u;
    }
    if (W < 384) {                                                 // narrower images: zeros behind the last pixel up to the widest tap read
        for (int e = tid; e < 3 * NR * 8; e += 256) img[(e >> 3) * PE_WS + 2 + W + (e & 7)] = 0;
    }
    // per-lane epilogue constants: channels 16 ct + 4 fg + r of the lane's pixel
    float bs[16], gm[16], bt[16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const float4 b4 = *reinterpret_cast<const float4*>(bias + 16 * ct + 4 * fg), g4 = *reinterpret_cast<const float4*>(gamma + 16 * ct + 4 * fg),
                     t4 = *reinterpret_cast<const float4*>(beta + 16 * ct + 4 * fg);
        bs[4 * ct] = b4.x; bs[4 * ct + 1] = b4.y; bs[4 * ct + 2] = b4.z; bs[4 * ct + 3] = b4.w;
        gm[4 * ct] = g4.x; gm[4 * ct + 1] = g4.y; gm[4 * ct + 2] = g4.z; gm[4 * ct + 3] = g4.w;
        bt[4 * ct] = t4.x; bt[4 * ct + 1] = t4.y; bt[4 * ct + 2] = t4.z; bt[4 * ct + 3] = t4.w;
    }
    __syncthreads();
    const int tiles = (Wo + 15) >> 4;
    for (int it = wave; it < PE_RB * tiles; it += 4) {             // (wave-uniform) one (output row, 16-pixel tile) at a time
        const int oyl = it / tiles, pt = it - oyl * tiles;
        const int oy = oy0 + oyl;
        if (oy >= Ho) continue;
        const int ox = pt * 16 + fm;                               // this lane's pixel (B operand column n = lane & 15)
        const int oxc = ox < Wo ? ox : Wo - 1;
        f32x4_t acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < 6; ++s2) {
            int grp = 4 * s2 + fg; grp = grp < 21 ? grp : 20;      // (groups 21..23 are zero padding of K: any staged row will do)
            const int c = grp / 7, ky = grp - 7 * c;
            const bf16_t* src = img + (c * NR + 4 * oyl + ky) * PE_WS + 4 * oxc;
            const uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 4);
            const bf16x8_t xf = __builtin_bit_cast(bf16x8_t, make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct][s2], xf, acc[ct], 0, 0, 0);
        }
        // D^T tile: lane (n = pixel fm, row group fg) holds channels 16 ct + 4 fg + r -> e = conv + bias, rounded to bf16 (what the LayerNorm -- and its
        // backward -- see, as when the GEMM wrote e and a LayerNorm kernel read it back)
        float ev[16];
        float sum = 0.f;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = bf2f(f2bf(acc[ct][r] + bs[4 * ct + r]));
                ev[4 * ct + r] = v;
                sum += v;
            }
        sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
        const float mean = sum * (1.0f / 64.0f);
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const float d = ev[i] - mean; sq = fmaf(d, d, sq); }
        sq += __shfl_xor(sq, 16, 64); sq += __shfl_xor(sq, 32, 64);
        const float rstd = rsqrtf(sq * (1.0f / 64.0f) + eps);
        if (ox < Wo) {
            const long pix = ((long)b * Ho + oy) * Wo + ox;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int ch = 16 * ct + 4 * fg;
                if (e_out) *reinterpret_cast<uint2*>(e_out + pix * 64 + ch) = make_uint2(pack2bf(ev[4 * ct], ev[4 * ct + 1]), pack2bf(ev[4 * ct + 2], ev[4 * ct + 3]));
                float y[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) y[r] = (ev[4 * ct + r] - mean) * rstd * gm[4 * ct + r] + bt[4 * ct + r];
                *reinterpret_cast<uint2*>(y_out + pix * 64 + ch) = make_uint2(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]));
            }
            if (stats && fg == 0) *reinterpret_cast<float2*>(stats + pix * 2) = make_float2(mean, rstd);
        }
    }
}

// weights [64, 3, 7, 7] fp32 (nn.Conv2d layout) -> the packed bf16 A operand [64][24 groups of (c, ky)][8 taps kx] of patch_embed_s1_kernel (zero where kx = 7 or
// group >= 21)
__global__ __launch_bounds__(256) void patch_embed_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 64 * 192) return;
    const int n = i / 192, k = i % 192, grp = k >> 3, kx = k & 7;
    out[i] = (grp < 21 && kx < 7) ? f2bf(w[(n * 21 + grp) * 7 + kx]) : (bf16_t)0;
}

extern "C" int cxr_patch_embed_pack_f32(const float* w, void* out, hipStream_t stream) {
    if (!w || !out) return CXR_ERR_ARG;
    CXR_LAUNCH(patch_embed_pack_kernel, dim3(cdiv(64 * 192, 256)), dim3(256), 0, stream, w, (bf16_t*)out);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_patch_embed_s1_f32(const float* px, const void* Wpk, const float* bias, const float* gamma, const float* beta, float eps,
                                      void* e_out, void* y_out, float* stats, int Bn, int H, int W, hipStream_t stream) {
    if (!px || !Wpk || !bias || !gamma || !beta || !y_out || Bn <= 0 || H < 4 || W < 4 || (H % 4) || (W % 4) || W > 384) return CXR_ERR_ARG;
    if ((((size_t)px) % 16) || (((size_t)Wpk) % 16) || (((size_t)y_out) % 8) || (e_out && (((size_t)e_out) % 8)) || (stats && !e_out)) return CXR_ERR_ARG;
    const int Ho = H / 4, Wo = W / 4;                              // (H + 2*2 - 7) / 4 + 1 for H % 4 == 0
    const long grid = (long)Bn * cdiv(Ho, PE_RB);
    if (grid > 0x7fffffffL) return CXR_ERR_ARG;
    CXR_LAUNCH(patch_embed_s1_kernel, dim3((unsigned)grid), dim3(256), 0, stream, px, (const bf16_t*)Wpk, bias, gamma, beta, eps, (bf16_t*)e_out,
               (bf16_t*)y_out, stats, H, W, Ho, Wo);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_im2col_nchw_f32(const float* px, void* col, int Bn, int Cin, int H, int W, int KS, int stride, int pad,
                                   int Ho, int Wo, int Kpad, hipStream_t stream) {
    if (Bn <= 0 || (Kpad % 8) || Kpad < Cin * KS * KS) return CXR_ERR_ARG;
    const long total = (long)Bn * Ho * Wo * (Kpad / 8);
    const size_t lds = ((size_t)Cin * KS * (W + 2 * pad) + Kpad) * 4;
    static int rows_on = -1;                                       // CXR_IM2COL_ROWS=0: the gather kernel (A/B)
    if (rows_on < 0) { const char* e = getenv("CXR_IM2COL_ROWS"); rows_on = (e && e[0] == '0') ? 0 : 1; }
    if (rows_on && lds <= 64 * 1024 && pad > 0 && (long)Bn * Ho < (1L << 31)) {
        CXR_LAUNCH(im2col_nchw_rows_kernel, dim3((unsigned)(Bn * Ho)), dim3(256), lds, stream, px, (bf16_t*)col, Cin, H, W, KS, stride, pad, Ho, Wo,
                   Cin * KS * KS, Kpad);
        CXR_LAUNCH_CHECK();
        return CXR_OK;
    }
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(im2col_nchw_kernel, dim3(grid), dim3(256), 0, stream, px, (bf16_t*)col, Bn, Cin, H, W, KS, stride, pad, Ho, Wo,
                       Cin * KS * KS, Kpad);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_im2col_tok_bf16(const void* x, long x_bs, long x_rs, void* col, int Bn, int Cin, int H, int W, int stride, int pad,
                                   int Ho, int Wo, hipStream_t stream) {
    if (Bn <= 0 || (Cin % 8) || (x_rs % 8) || (x_bs % 8)) return CXR_ERR_ARG;
    const long total = (long)Bn * Ho * Wo * 9 * (Cin / 8);
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(im2col_tok_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, x_bs, x_rs, (bf16_t*)col, Bn, Cin, H, W,
                       stride, pad, Ho, Wo);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_col2im_tok_bf16(const void* dcol, void* dx, long dx_bs, long dx_rs, int Bn, int Cin, int H, int W, int stride, int pad,
                                   int Ho, int Wo, hipStream_t stream) {
    if (Bn <= 0 || (Cin % 8) || (dx_rs % 8) || (dx_bs % 8)) return CXR_ERR_ARG;
    const long total = (long)Bn * H * W * (Cin / 8);
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(col2im_tok_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dcol, (bf16_t*)dx, dx_bs, dx_rs, Bn, Cin, H, W,
                       stride, pad, Ho, Wo);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- depthwise 3x3 + BN
// Folded parameters per projection: wf[9][C] = conv_w[c][tap] * bn_scale[c], sh[C] = bn_bias - mean*bn_scale  (fp32)
// (eval-mode BatchNorm, TF5 modeling_cvt.py:93-110; the batch-statistics variant feeds batch mean/var through the same fold)
__global__ void bn_fold_kernel(const float* __restrict__ w /*[C,9]*/, const float* __restrict__ g, const float* __restrict__ b,
                               const float* __restrict__ mean, const float* __restrict__ var, float eps, float* __restrict__ wf,
                               float* __restrict__ sh, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = g[c] * rsqrtf(var[c] + eps);
#pragma unroll
    for (int t = 0; t < 9; ++t) wf[t * C + c] = w[c * 9 + t] * s;
    sh[c] = b[c] - mean[c] * s;
}

extern "C" int cxr_bn_fold(const float* w, const float* g, const float* b, const float* mean, const float* var, float eps, float* wf,
                           float* sh, int C, hipStream_t stream) {
    CXR_LAUNCH(bn_fold_kernel, dim3(cdiv(C, 128)), dim3(128), 0, stream, w, g, b, mean, var, eps, wf, sh, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// param gradients from the tap sums: G[9][C] = sum dy*x_tap, S[C] = sum dy
//   dw[c][t] = s*G[t][c];  dgamma = r*(sum_t w[c][t]*G[t][c] - mean*S);  dbeta = S      (s = gamma*r, r = rsqrt(var+eps))
__global__ void bn_fold_bwd_kernel(const float* __restrict__ w, const float* __restrict__ g, const float* __restrict__ mean,
                                   const float* __restrict__ var, float eps, const float* __restrict__ G, const float* __restrict__ S,
                                   float* __restrict__ dw, float* __restrict__ dg, float* __restrict__ db, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = rsqrtf(var[c] + eps), s = g[c] * r;
    float dot = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) { dw[c * 9 + t] += s * G[t * C + c]; dot += w[c * 9 + t] * G[t * C + c]; }
    dg[c] += r * (dot - mean[c] * S[c]);
    db[c] += S[c];
}

extern "C" int cxr_bn_fold_bwd(const float* w, const float* g, const float* mean, const float* var, float eps, const float* G,
                               const float* S, float* dw, float* dg, float* db, int C, hipStream_t stream) {
    CXR_LAUNCH(bn_fold_bwd_kernel, dim3(cdiv(C, 128)), dim3(128), 0, stream, w, g, mean, var, eps, G, S, dw, dg, db, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// x: [Bn, x_rs-strided rows], spatial token (y,x) at row `tok0 + y*W + x`; outputs y0 (and y1 when NOUT==2) at row `tok0 + oy*Wo + ox`.
// When tok0 == 1 the class-token row 0 is copied through unchanged (TF5 modeling_cvt.py:195-198).
template <int NOUT>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const bf16_t* __restrict__ x, long x_bs, long x_rs, const float* __restrict__ wf0,
                                                         const float* __restrict__ sh0, const float* __restrict__ wf1,
                                                         const float* __restrict__ sh1, bf16_t* __restrict__ y0, bf16_t* __restrict__ y1,
                                                         long y_bs, long y_rs, int Bn, int C, int H, int W, int stride, int Ho, int Wo, int tok0) {
    const int cch = C / 8;
    const long total = (long)Bn * (Ho * Wo + tok0) * cch;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % cch) * 8;
        const long t = idx / cch;
        const int tok = (int)(t % (Ho * Wo + tok0)), b = (int)(t / (Ho * Wo + tok0));
        const bf16_t* xb = x + (long)b * x_bs;
        if (tok < tok0) {                                                   // class token passthrough
            const uint4 v = *reinterpret_cast<const uint4*>(xb + c8);
            *reinterpret_cast<uint4*>(y0 + (long)b * y_bs + c8) = v;
            if (NOUT == 2) *reinterpret_cast<uint4*>(y1 + (long)b * y_bs + c8) = v;
            continue;
        }
        const int p = tok - tok0, oy = p / Wo, ox = p % Wo;
        float a0[8], a1[8];
        {
            const float4 s0 = *reinterpret_cast<const float4*>(sh0 + c8), s1 = *reinterpret_cast<const float4*>(sh0 + c8 + 4);
            a0[0] = s0.x; a0[1] = s0.y; a0[2] = s0.z; a0[3] = s0.w; a0[4] = s1.x; a0[5] = s1.y; a0[6] = s1.z; a0[7] = s1.w;
            if (NOUT == 2) {
                const float4 t0 = *reinterpret_cast<const float4*>(sh1 + c8), t1 = *reinterpret_cast<const float4*>(sh1 + c8 + 4);
                a1[0] = t0.x; a1[1] = t0.y; a1[2] = t0.z; a1[3] = t0.w; a1[4] = t1.x; a1[5] = t1.y; a1[6] = t1.z; a1[7] = t1.w;
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(xb + (long)(tok0 + iy * W + ix) * x_rs + c8), f);
                const float* w0 = wf0 + (ky * 3 + kx) * C + c8;
                const float4 u0 = *reinterpret_cast<const float4*>(w0), u1 = *reinterpret_cast<const float4*>(w0 + 4);
                a0[0] += f[0] * u0.x; a0[1] += f[1] * u0.y; a0[2] += f[2] * u0.z; a0[3] += f[3] * u0.w;
                a0[4] += f[4] * u1.x; a0[5] += f[5] * u1.y; a0[6] += f[6] * u1.z; a0[7] += f[7] * u1.w;
                if (NOUT == 2) {
                    const float* w1 = wf1 + (ky * 3 + kx) * C + c8;
                    const float4 q0 = *reinterpret_cast<const float4*>(w1), q1 = *reinterpret_cast<const float4*>(w1 + 4);
                    a1[0] += f[0] * q0.x; a1[1] += f[1] * q0.y; a1[2] += f[2] * q0.z; a1[3] += f[3] * q0.w;
                    a1[4] += f[4] * q1.x; a1[5] += f[5] * q1.y; a1[6] += f[6] * q1.z; a1[7] += f[7] * q1.w;
                }
            }
        }
        *reinterpret_cast<uint4*>(y0 + (long)b * y_bs + (long)tok * y_rs + c8) = pack8(a0);
        if (NOUT == 2) *reinterpret_cast<uint4*>(y1 + (long)b * y_bs + (long)tok * y_rs + c8) = pack8(a1);
    }
}

extern "C" int cxr_dwconv_bn_fwd_bf16(const void* x, long x_bs, long x_rs, const float* wf0, const float* sh0, const float* wf1,
                                      const float* sh1, void* y0, void* y1, long y_bs, long y_rs, int Bn, int C, int H, int W, int stride,
                                      int tok0, hipStream_t stream) {
    if (Bn <= 0 || (C % 8) || (x_rs % 8) || (y_rs % 8) || (x_bs % 8) || (y_bs % 8) || (stride != 1 && stride != 2)) return CXR_ERR_ARG;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long total = (long)Bn * (Ho * Wo + tok0) * (C / 8);
    const int grid = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
    if (y1) CXR_LAUNCH((dwconv_fwd_kernel<2>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, x_bs, x_rs, wf0, sh0, wf1, sh1,
                               (bf16_t*)y0, (bf16_t*)y1, y_bs, y_rs, Bn, C, H, W, stride, Ho, Wo, tok0);
    else    CXR_LAUNCH((dwconv_fwd_kernel<1>), dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, x_bs, x_rs, wf0, sh0, wf0, sh0,
                               (bf16_t*)y0, (bf16_t*)nullptr, y_bs, y_rs, Bn, C, H, W, stride, Ho, Wo, tok0);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// Backward, input gradient (gather form): dx[b,(iy,ix),c] (+)= sum_taps wf[tap][c] * dy[b,(oy,ox),c]; class row: dx[0] (+)= dy[0].
// Up to three projections (q stride 1; k, v stride 2) are folded into one pass so dx is written once.
struct DwBwdProj { const bf16_t* dy; const float* wf; long bs, rs; int stride, Ho, Wo; };
__global__ __launch_bounds__(256) void dwconv_bwd_dx_kernel(DwBwdProj p0, DwBwdProj p1, DwBwdProj p2, int nproj, bf16_t* __restrict__ dx,
                                                            long dx_bs, long dx_rs, int Bn, int C, int H, int W, int tok0) {
    const int cch = C / 8;
    const long total = (long)Bn * (H * W + tok0) * cch;
    const DwBwdProj pr[3] = {p0, p1, p2};
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % cch) * 8;
        const long t = idx / cch;
        const int tok = (int)(t % (H * W + tok0)), b = (int)(t / (H * W + tok0));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tok < tok0) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q >= nproj) break;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(pr[q].dy + (long)b * pr[q].bs + c8), f);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += f[j];
            }
        } else {
            const int p = tok - tok0, iy = p / W, ix = p % W;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                if (q >= nproj) break;
                const int st = pr[q].stride;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int ty = iy + 1 - ky;
                    if (ty < 0 || (ty % st) || ty / st >= pr[q].Ho) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int tx = ix + 1 - kx;
                        if (tx < 0 || (tx % st) || tx / st >= pr[q].Wo) continue;
                        float f[8];
                        unpack8(*reinterpret_cast<const uint4*>(pr[q].dy + (long)b * pr[q].bs +
                                                                (long)(tok0 + (ty / st) * pr[q].Wo + tx / st) * pr[q].rs + c8), f);
                        const float* w = pr[q].wf + (ky * 3 + kx) * C + c8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[j] += f[j] * w[j];
                    }
                }
            }
        }
        *reinterpret_cast<uint4*>(dx + (long)b * dx_bs + (long)tok * dx_rs + c8) = pack8(acc);
    }
}

extern "C" int cxr_dwconv_bn_bwd_dx_bf16(const void* dy0, const float* wf0, long bs0, long rs0, int stride0,
                                         const void* dy1, const float* wf1, long bs1, long rs1, int stride1,
                                         const void* dy2, const float* wf2, long bs2, long rs2, int stride2,
                                         int nproj, void* dx, long dx_bs, long dx_rs, int Bn, int C, int H, int W, int tok0,
                                         hipStream_t stream) {
    if (Bn <= 0 || (C % 8) || nproj < 1 || nproj > 3) return CXR_ERR_ARG;
    auto mk = [&](const void* dy, const float* wf, long bs, long rs, int st) {
        DwBwdProj p; p.dy = (const bf16_t*)dy; p.wf = wf; p.bs = bs; p.rs = rs; p.stride = st > 0 ? st : 1;
        p.Ho = (H + 2 - 3) / p.stride + 1; p.Wo = (W + 2 - 3) / p.stride + 1; return p;
    };
    const long total = (long)Bn * (H * W + tok0) * (C / 8);
    const int grid = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
    CXR_LAUNCH(dwconv_bwd_dx_kernel, dim3(grid), dim3(256), 0, stream, mk(dy0, wf0, bs0, rs0, stride0), mk(dy1, wf1, bs1, rs1, stride1),
                       mk(dy2, wf2, bs2, rs2, stride2), nproj, (bf16_t*)dx, dx_bs, dx_rs, Bn, C, H, W, tok0);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// out[k] = sum_g ws[g][k]: second stage of the per-channel reductions below. Workgroups write one partial row each and this kernel adds the
// rows -- same-address fp32 atomics from ~1000 workgroups serialise at ~30 ns each and were the whole cost of those kernels.
__global__ __launch_bounds__(1024) void partial_rows_sum_kernel(const float* __restrict__ ws, int G, int K, float* __restrict__ out) {
    __shared__ float red[16][64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 columns x 16 row lanes: <= 64 rows per thread, 8 loads in flight
    const int k = blockIdx.x * 64 + tx;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (k < K) {
        int g = ty;
        for (; g + 7 * 16 < G; g += 8 * 16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += ws[(long)(g + 16 * u) * K + k];
        }
        for (; g < G; g += 16) acc[0] += ws[(long)g * K + k];
    }
    red[ty][tx] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (ty == 0 && k < K) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][tx];
        out[k] = s;
    }
}

static inline int dw_partial_rows(long npix, int npl) {       // workgroups of a per-channel reduction pass: 256..1024, >= 4 pixels per lane
    long g = npix / (4L * npl);
    g = g < 256 ? 256 : (g > 1024 ? 1024 : g);
    const long cap = (npix + npl - 1) / npl;                 // at least one pixel per lane group
    return (int)(g < cap ? g : cap);
}

// Backward, tap sums: G[tap][c] = sum_{b,oy,ox} dy[b,(oy,ox),c] * x[b,(iy,ix),c],  S[c] = sum dy.  Each workgroup writes its partial
// [10][C] row (taps 0-8, then S) to `ws`; partial_rows_sum_kernel adds the rows (deterministic, no atomics).
__global__ __launch_bounds__(256) void dwconv_bwd_w_kernel(const bf16_t* __restrict__ x, long x_bs, long x_rs, const bf16_t* __restrict__ dy,
                                                           long dy_bs, long dy_rs, float* __restrict__ ws,
                                                           int Bn, int C, int H, int W, int stride, int Ho, int Wo, int tok0, int pix_per_block) {
    // block = (C/8 channel chunks) x (256/(C/8) pixel lanes); each thread keeps 10x8 partial sums
    const int cch = C / 8;
    const int cl = threadIdx.x % cch, pl = threadIdx.x / cch, npl = 256 / cch;
    const int c8 = cl * 8;
    float g[10][8];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) g[t][j] = 0.f;
    const long npix = (long)Bn * Ho * Wo;
    const long beg = (long)blockIdx.x * pix_per_block, end = beg + pix_per_block < npix ? beg + pix_per_block : npix;
    if (pl < npl) {
        for (long pix = beg + pl; pix < end; pix += npl) {
            const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
            float d[8];
            unpack8(*reinterpret_cast<const uint4*>(dy + (long)b * dy_bs + (long)(tok0 + oy * Wo + ox) * dy_rs + c8), d);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[9][j] += d[j];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * stride - 1 + ky;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * stride - 1 + kx;
                    if (ix < 0 || ix >= W) continue;
                    float f[8];
                    unpack8(*reinterpret_cast<const uint4*>(x + (long)b * x_bs + (long)(tok0 + iy * W + ix) * x_rs + c8), f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g[ky * 3 + kx][j] += d[j] * f[j];
                }
            }
        }
    }
    // block reduction over the pixel lanes without atomics: two passes of 5 taps through a [5][npl][C] fp32 LDS image (<= 40 KB)
    __shared__ __attribute__((aligned(16))) float red[10240];
#pragma unroll
    for (int tg = 0; tg < 2; ++tg) {
        if (pl < npl) {
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                float* dst = red + ((t * npl + pl) * C + c8);
                *reinterpret_cast<float4*>(dst) = make_float4(g[5 * tg + t][0], g[5 * tg + t][1], g[5 * tg + t][2], g[5 * tg + t][3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(g[5 * tg + t][4], g[5 * tg + t][5], g[5 * tg + t][6], g[5 * tg + t][7]);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 5 * C; i += 256) {
            const int t = i / C, c = i % C;
            float sum = 0.f;
            for (int q = 0; q < npl; ++q) sum += red[(t * npl + q) * C + c];
            ws[(long)blockIdx.x * 10 * C + (5 * tg + t) * C + c] = sum;
        }
        __syncthreads();
    }
}

// GS [10][C] (rows 0-8 = G, row 9 = S) is OVERWRITTEN; ws = scratch of cxr_dwconv_ws_floats(C) fp32 elements.
extern "C" int cxr_dwconv_bn_bwd_w_bf16(const void* x, long x_bs, long x_rs, const void* dy, long dy_bs, long dy_rs, float* GS, float* ws,
                                        int Bn, int C, int H, int W, int stride, int tok0, hipStream_t stream) {
    if (Bn <= 0 || (C % 8) || C > 384 || (256 % (C / 8) != 0 && C / 8 > 256) || !ws) return CXR_ERR_ARG;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long npix = (long)Bn * Ho * Wo;
    const int rows = dw_partial_rows(npix, 256 / (C / 8));
    const int ppb = (int)cdiv(npix, rows), grid = cdiv(npix, ppb);
    CXR_LAUNCH(dwconv_bwd_w_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, x_bs, x_rs, (const bf16_t*)dy, dy_bs,
                       dy_rs, ws, Bn, C, H, W, stride, Ho, Wo, tok0, ppb);
    CXR_LAUNCH(partial_rows_sum_kernel, dim3(cdiv(10 * C, 64)), dim3(1024), 0, stream, ws, grid, 10 * C, GS);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_dwconv_ws_floats(int C) { return 1024 * 10 * C; }     // scratch elements of the per-channel reduction passes

// ---------------------------------------------------------------------------------------------- train-mode BatchNorm (batch statistics)
// nn.BatchNorm2d in training mode (TF5 modeling_cvt.py:93-110 under model.train(); SURVEY.md quirk Q7: the "frozen" encoder of the SCST
// stage still runs it). y = gamma*(c - mean_B)/sqrt(var_B + eps) + beta with c the raw depthwise conv output and (mean_B, var_B) the biased
// statistics over all Bn*Ho*Wo positions of the launch; running stats move by `momentum` with the UNBIASED variance.
// Forward = statistics pass (below: one read of the activation, conv outputs are not written) + finalize (stats -> folded taps) + the
// ordinary folded dwconv_fwd_kernel. Cheaper than materialising c: R + (R+W) instead of (R+W) + (R+W).
// WITH_DY = false: stats = (sum c, sum c^2)   -- forward statistics
// WITH_DY = true : stats = (sum dy, sum dy*c)  -- the two reductions the backward through the batch statistics needs (c recomputed on the fly)
template <int NOUT, bool WITH_DY>
__global__ __launch_bounds__(256) void dwconv_stats_kernel(const bf16_t* __restrict__ x, long x_bs, long x_rs, const float* __restrict__ w0 /*[9][C] raw*/,
                                                           const float* __restrict__ w1, const bf16_t* __restrict__ dy0, const bf16_t* __restrict__ dy1,
                                                           long dy_bs, long dy_rs, float* __restrict__ ws /*[grid][NOUT][2][C] partial rows*/,
                                                           int Bn, int C, int H, int W, int stride, int Ho, int Wo, int tok0, int pix_per_block) {
    const int cch = C / 8;
    const int cl = threadIdx.x % cch, pl = threadIdx.x / cch, npl = 256 / cch;
    const int c8 = cl * 8;
    float sm[NOUT][8], sq[NOUT][8];
#pragma unroll
    for (int q = 0; q < NOUT; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) { sm[q][j] = 0.f; sq[q][j] = 0.f; }
    const long npix = (long)Bn * Ho * Wo;
    const long beg = (long)blockIdx.x * pix_per_block, end = beg + pix_per_block < npix ? beg + pix_per_block : npix;
    if (pl < npl) {
        for (long pix = beg + pl; pix < end; pix += npl) {
            const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
            float a[NOUT][8];
#pragma unroll
            for (int q = 0; q < NOUT; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[q][j] = 0.f;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * stride - 1 + ky;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * stride - 1 + kx;
                    if (ix < 0 || ix >= W) continue;
                    float f[8];
                    unpack8(*reinterpret_cast<const uint4*>(x + (long)b * x_bs + (long)(tok0 + iy * W + ix) * x_rs + c8), f);
                    const float* u = w0 + (ky * 3 + kx) * C + c8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[0][j] += f[j] * u[j];
                    if (NOUT == 2) {
                        const float* u1 = w1 + (ky * 3 + kx) * C + c8;
#pragma unroll
                        for (int j = 0; j < 8; ++j) a[NOUT - 1][j] += f[j] * u1[j];
                    }
                }
            }
            if (WITH_DY) {
#pragma unroll
                for (int q = 0; q < NOUT; ++q) {
                    float d[8];
                    unpack8(*reinterpret_cast<const uint4*>((q == 0 ? dy0 : dy1) + (long)b * dy_bs + (long)(tok0 + oy * Wo + ox) * dy_rs + c8), d);
#pragma unroll
                    for (int j = 0; j < 8; ++j) { sm[q][j] += d[j]; sq[q][j] += d[j] * a[q][j]; }
                }
            } else {
#pragma unroll
                for (int q = 0; q < NOUT; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { sm[q][j] += a[q][j]; sq[q][j] += a[q][j] * a[q][j]; }
            }
        }
    }
    __shared__ __attribute__((aligned(16))) float red[4 * 2048];            // [NOUT*2][npl][C], npl*C <= 2048 for C in {64,192,384}
    if (pl < npl) {
#pragma unroll
        for (int q = 0; q < NOUT; ++q) {
            float* d0 = red + (((q * 2 + 0) * npl + pl) * C + c8);
            float* d1 = red + (((q * 2 + 1) * npl + pl) * C + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { d0[j] = sm[q][j]; d1[j] = sq[q][j]; }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NOUT * 2 * C; i += 256) {
        const int k = i / C, c = i % C;
        float sum = 0.f;
        for (int q = 0; q < npl; ++q) sum += red[(k * npl + q) * C + c];
        ws[(long)blockIdx.x * NOUT * 2 * C + i] = sum;
    }
}

// stats [nproj][2][C] (overwritten) <- per-channel reductions over the raw depthwise conv outputs c of one (w1 == NULL) or two
// projections: (sum c, sum c^2) when dy0 == NULL (forward), (sum dy, sum dy*c) otherwise (backward; dy1 pairs with w1, same strides).
// ws = scratch of cxr_dwconv_ws_floats(C) fp32 elements.
extern "C" int cxr_dwconv_stats_bf16(const void* x, long x_bs, long x_rs, const float* w0, const float* w1, const void* dy0, const void* dy1,
                                     long dy_bs, long dy_rs, float* stats, float* ws, int Bn, int C, int H, int W, int stride, int tok0,
                                     hipStream_t stream) {
    if (Bn <= 0 || (C % 8) || C > 384 || (256 / (C / 8)) * C > 2048 || (stride != 1 && stride != 2) || !ws) return CXR_ERR_ARG;
    if ((w1 != nullptr) != (dy1 != nullptr) && dy0) return CXR_ERR_ARG;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long npix = (long)Bn * Ho * Wo;
    const int rows = dw_partial_rows(npix, 256 / (C / 8));
    const int ppb = (int)cdiv(npix, rows);
    const int nblk = cdiv(npix, ppb);
    const dim3 grid(nblk);
#define STATS(N_, D_) CXR_LAUNCH((dwconv_stats_kernel<N_, D_>), grid, dim3(256), 0, stream, (const bf16_t*)x, x_bs, x_rs, w0, w1 ? w1 : w0,        \
                                 (const bf16_t*)dy0, (const bf16_t*)(dy1 ? dy1 : dy0), dy_bs, dy_rs, ws, Bn, C, H, W, stride, Ho, Wo, tok0, ppb)
    if (dy0) { if (w1) STATS(2, true); else STATS(1, true); }
    else     { if (w1) STATS(2, false); else STATS(1, false); }
#undef STATS
    const int K = (w1 ? 2 : 1) * 2 * C;
    CXR_LAUNCH(partial_rows_sum_kernel, dim3(cdiv(K, 64)), dim3(1024), 0, stream, ws, nblk, K, stats);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// stats -> batch mean / rstd (kept for backward), running-stat update (in place, fp32 master), folded taps for dwconv_fwd_kernel
__global__ void bn_train_finalize_kernel(const float* __restrict__ stats /*[2][C]*/, float count, const float* __restrict__ w /*[C,9]*/,
                                         const float* __restrict__ g, const float* __restrict__ b, float eps, float momentum,
                                         float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ mean_out,
                                         float* __restrict__ rstd_out, float* __restrict__ wf, float* __restrict__ sh, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float mean = stats[c] / count;
    const float var = fmaxf(stats[C + c] / count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    mean_out[c] = mean; rstd_out[c] = rstd;
    if (momentum > 0.f) {
        run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
        run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (count > 1.f ? count / (count - 1.f) : 1.f);
    }
    const float s = g[c] * rstd;
#pragma unroll
    for (int t = 0; t < 9; ++t) wf[t * C + c] = w[c * 9 + t] * s;
    sh[c] = b[c] - mean * s;
}

extern "C" int cxr_bn_train_finalize(const float* stats, long count, const float* w, const float* g, const float* b, float eps, float momentum,
                                     float* run_mean, float* run_var, float* mean_out, float* rstd_out, float* wf, float* sh, int C,
                                     hipStream_t stream) {
    if (count <= 0 || C <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(bn_train_finalize_kernel, dim3(cdiv(C, 128)), dim3(128), 0, stream, stats, (float)count, w, g, b, eps, momentum, run_mean, run_var,
                       mean_out, rstd_out, wf, sh, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// Backward through the batch statistics. With dy = grad wrt the BN output, x^ = (c - mean)*rstd, M positions:
//   dgamma = sum dy*x^,  dbeta = sum dy,  dc = gamma*rstd*(dy - dbeta/M - x^ * dgamma/M)  =  a*dy + kb + kc*c
// SD = (sum dy, sum dy*c) per channel from cxr_dwconv_stats_bf16 with dy.
__global__ void bn_train_bwd_coef_kernel(const float* __restrict__ g, const float* __restrict__ mean, const float* __restrict__ rstd,
                                         const float* __restrict__ SD /*[2][C]*/, float count, float* __restrict__ dg, float* __restrict__ db,
                                         float* __restrict__ coef /*[3][C]: a, kb, kc*/, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float r = rstd[c], mu = mean[c], S = SD[c];
    const float dgam = r * (SD[C + c] - mu * S);
    dg[c] += dgam;
    db[c] += S;
    const float a = g[c] * r, m1 = S / count, m2 = dgam / count;
    const float kc = -a * m2 * r;
    coef[c] = a; coef[C + c] = -a * m1 - kc * mu; coef[2 * C + c] = kc;
}

extern "C" int cxr_bn_train_bwd_coef(const float* g, const float* mean, const float* rstd, const float* SD, long count, float* dg, float* db,
                                     float* coef, int C, hipStream_t stream) {
    if (count <= 0 || C <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(bn_train_bwd_coef_kernel, dim3(cdiv(C, 128)), dim3(128), 0, stream, g, mean, rstd, SD, (float)count, dg, db, coef, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---- statistics pass + row sum + per-channel epilogue in one C-ABI call (two launches instead of 3-4 per projection pair): the row-sum kernel
// of the statistics partials finishes with the per-channel work that used to be separate 4-us launches on the critical path
// (bn_train_finalize / bn_train_bwd_coef): 126 launches per training step.
struct BnFwdProj { const float* w; const float* g; const float* b; float* run_mean; float* run_var; float* mean_out; float* rstd_out; float* wf; float* sh; };
struct BnBwdProj { const float* g; const float* mean; const float* rstd; float* dg; float* db; float* coef; };

// ws [G][nproj][2][C] partial rows -> per projection q (blockIdx.y) and channel c: (sum0, sum1) = column sums of [q][0][c], [q][1][c]
__device__ __forceinline__ void bn_rows_sum2(const float* __restrict__ ws, int G, int K, int col0, int col1, float& s0, float& s1, float (*red)[64][2]) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    float a0 = 0.f, a1 = 0.f;
    {
        float u0[4] = {0, 0, 0, 0}, u1[4] = {0, 0, 0, 0};                    // 8 independent loads in flight per lane
        int g = ty;
        for (; g + 3 * 16 < G; g += 4 * 16) {
#pragma unroll
            for (int u = 0; u < 4; ++u) { u0[u] += ws[(long)(g + 16 * u) * K + col0]; u1[u] += ws[(long)(g + 16 * u) * K + col1]; }
        }
        for (; g < G; g += 16) { u0[0] += ws[(long)g * K + col0]; u1[0] += ws[(long)g * K + col1]; }
        a0 = (u0[0] + u0[1]) + (u0[2] + u0[3]); a1 = (u1[0] + u1[1]) + (u1[2] + u1[3]);
    }
    red[ty][tx][0] = a0; red[ty][tx][1] = a1;
    __syncthreads();
    s0 = 0.f; s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s0 += red[q][tx][0]; s1 += red[q][tx][1]; }
}

__global__ __launch_bounds__(1024) void bn_train_reduce_finalize_kernel(const float* __restrict__ ws, int G, int C, int nproj, float count, float eps,
                                                                        float momentum, BnFwdProj p0, BnFwdProj p1, float* __restrict__ stats) {
    __shared__ float red[16][64][2];
    const int q = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63);
    const BnFwdProj P = q == 0 ? p0 : p1;
    const int K = nproj * 2 * C;
    const int cc = c < C ? c : C - 1;
    float s0, s1;
    bn_rows_sum2(ws, G, K, (q * 2 + 0) * C + cc, (q * 2 + 1) * C + cc, s0, s1, red);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
    if (stats) { stats[(q * 2 + 0) * C + c] = s0; stats[(q * 2 + 1) * C + c] = s1; }
    const float mean = s0 / count;
    const float var = fmaxf(s1 / count - mean * mean, 0.f);
    const float rstd = rsqrtf(var + eps);
    P.mean_out[c] = mean; P.rstd_out[c] = rstd;
    if (momentum > 0.f) {
        P.run_mean[c] = (1.f - momentum) * P.run_mean[c] + momentum * mean;
        P.run_var[c] = (1.f - momentum) * P.run_var[c] + momentum * var * (count > 1.f ? count / (count - 1.f) : 1.f);
    }
    const float sc = P.g[c] * rstd;
#pragma unroll
    for (int t = 0; t < 9; ++t) P.wf[t * C + c] = P.w[c * 9 + t] * sc;
    P.sh[c] = P.b[c] - mean * sc;
}

__global__ __launch_bounds__(1024) void bn_train_reduce_coef_kernel(const float* __restrict__ ws, int G, int C, int nproj, float count, BnBwdProj p0,
                                                                    BnBwdProj p1) {
    __shared__ float red[16][64][2];
    const int q = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63);
    const BnBwdProj P = q == 0 ? p0 : p1;
    const int K = nproj * 2 * C;
    const int cc = c < C ? c : C - 1;
    float S, D;
    bn_rows_sum2(ws, G, K, (q * 2 + 0) * C + cc, (q * 2 + 1) * C + cc, S, D, red);
    if ((threadIdx.x >> 6) != 0 || c >= C) return;
    const float r = P.rstd[c], mu = P.mean[c];
    const float dgam = r * (D - mu * S);
    P.dg[c] += dgam;
    P.db[c] += S;
    const float a = P.g[c] * r, m1 = S / count, m2 = dgam / count;
    const float kc = -a * m2 * r;
    P.coef[c] = a; P.coef[C + c] = -a * m1 - kc * mu; P.coef[2 * C + c] = kc;
}

static int launch_dwconv_stats_partials(const void* x, long x_bs, long x_rs, const float* w0, const float* w1, const void* dy0, const void* dy1, long dy_bs,
                                        long dy_rs, float* ws, int Bn, int C, int H, int W, int stride, int tok0, hipStream_t stream, int* nblk_out,
                                        long* count_out) {
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long npix = (long)Bn * Ho * Wo;
    const int rows = dw_partial_rows(npix, 256 / (C / 8));
    const int ppb = (int)cdiv(npix, rows);
    const int nblk = cdiv(npix, ppb);
    const dim3 grid(nblk);
#define STATS(N_, D_) CXR_LAUNCH((dwconv_stats_kernel<N_, D_>), grid, dim3(256), 0, stream, (const bf16_t*)x, x_bs, x_rs, w0, w1 ? w1 : w0,        \
                                 (const bf16_t*)dy0, (const bf16_t*)(dy1 ? dy1 : dy0), dy_bs, dy_rs, ws, Bn, C, H, W, stride, Ho, Wo, tok0, ppb)
    if (dy0) { if (w1) STATS(2, true); else STATS(1, true); }
    else     { if (w1) STATS(2, false); else STATS(1, false); }
#undef STATS
    *nblk_out = nblk; *count_out = npix;
    return CXR_OK;
}

// Forward of train-mode BatchNorm for one (wraw1 == NULL) or two projections: statistics of the raw conv outputs, batch mean / rstd, running-stat
// update and the folded taps for cxr_dwconv_bn_fwd_bf16. Per projection i: wt_i raw taps [9][C] (conv layout for the kernels), w_i [C,9] (parameter
// layout), gamma, beta, running mean / var (updated in place), outputs mean_i / rstd_i [C], wf_i [9][C], sh_i [C].
extern "C" int cxr_dwconv_bn_train_fwd_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int stride, int tok0, float eps,
                                                  float momentum, float* ws, const float* wt0, const float* w0, const float* g0, const float* b0,
                                                  float* rm0, float* rv0, float* mean0, float* rstd0, float* wf0, float* sh0, const float* wt1,
                                                  const float* w1, const float* g1, const float* b1, float* rm1, float* rv1, float* mean1, float* rstd1,
                                                  float* wf1, float* sh1, hipStream_t stream) {
    if (Bn <= 0 || (C % 8) || C > 384 || (256 / (C / 8)) * C > 2048 || (stride != 1 && stride != 2) || !ws || !wt0) return CXR_ERR_ARG;
    int nblk; long count;
    launch_dwconv_stats_partials(x, x_bs, x_rs, wt0, wt1, nullptr, nullptr, 0, 0, ws, Bn, C, H, W, stride, tok0, stream, &nblk, &count);
    BnFwdProj p0{w0, g0, b0, rm0, rv0, mean0, rstd0, wf0, sh0}, p1{w1, g1, b1, rm1, rv1, mean1, rstd1, wf1, sh1};
    const int nproj = wt1 ? 2 : 1;
    CXR_LAUNCH(bn_train_reduce_finalize_kernel, dim3(cdiv(C, 64), nproj), dim3(1024), 0, stream, ws, nblk, C, nproj, (float)count, eps, momentum, p0,
                       nproj == 2 ? p1 : p0, (float*)nullptr);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// Backward counterpart: (sum dy, sum dy*c) per channel, then dgamma / dbeta (accumulated) and the coefficients (a, kb, kc) [3][C] of
// dc = a*dy + kb + kc*c for cxr_dwconv_bn_train_dc_bf16.
extern "C" int cxr_dwconv_bn_train_bwd_stats_bf16(const void* x, long x_bs, long x_rs, int Bn, int C, int H, int W, int stride, int tok0, float* ws,
                                                  const float* wt0, const void* dy0, const float* g0, const float* mean0, const float* rstd0, float* dg0,
                                                  float* db0, float* coef0, const float* wt1, const void* dy1, const float* g1, const float* mean1,
                                                  const float* rstd1, float* dg1, float* db1, float* coef1, long dy_bs, long dy_rs, hipStream_t stream) {
    if (Bn <= 0 || (C % 8) || C > 384 || (256 / (C / 8)) * C > 2048 || (stride != 1 && stride != 2) || !ws || !wt0 || !dy0) return CXR_ERR_ARG;
    if ((wt1 != nullptr) != (dy1 != nullptr)) return CXR_ERR_ARG;
    int nblk; long count;
    launch_dwconv_stats_partials(x, x_bs, x_rs, wt0, wt1, dy0, dy1, dy_bs, dy_rs, ws, Bn, C, H, W, stride, tok0, stream, &nblk, &count);
    BnBwdProj p0{g0, mean0, rstd0, dg0, db0, coef0}, p1{g1, mean1, rstd1, dg1, db1, coef1};
    const int nproj = wt1 ? 2 : 1;
    CXR_LAUNCH(bn_train_reduce_coef_kernel, dim3(cdiv(C, 64), nproj), dim3(1024), 0, stream, ws, nblk, C, nproj, (float)count, p0, nproj == 2 ? p1 : p0);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// dy <- dc = a*dy + kb + kc*c in place, c recomputed from the activation and the raw taps (class-token rows are untouched: no BN there)
__global__ __launch_bounds__(256) void dwconv_bn_train_dc_kernel(const bf16_t* __restrict__ x, long x_bs, long x_rs, const float* __restrict__ wr /*[9][C]*/,
                                                                 const float* __restrict__ coef, bf16_t* __restrict__ dy, long dy_bs, long dy_rs,
                                                                 int Bn, int C, int H, int W, int stride, int Ho, int Wo, int tok0) {
    const int cch = C / 8;
    const long total = (long)Bn * Ho * Wo * cch;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % cch) * 8;
        const long t = idx / cch;
        const int p = (int)(t % (Ho * Wo)), b = (int)(t / (Ho * Wo));
        const int oy = p / Wo, ox = p % Wo;
        const bf16_t* xb = x + (long)b * x_bs;
        float c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * stride - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * stride - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                float f[8];
                unpack8(*reinterpret_cast<const uint4*>(xb + (long)(tok0 + iy * W + ix) * x_rs + c8), f);
                const float* u = wr + (ky * 3 + kx) * C + c8;
#pragma unroll
                for (int j = 0; j < 8; ++j) c[j] += f[j] * u[j];
            }
        }
        bf16_t* dp = dy + (long)b * dy_bs + (long)(tok0 + p) * dy_rs + c8;
        float d[8];
        unpack8(*reinterpret_cast<const uint4*>(dp), d);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = coef[c8 + j] * d[j] + coef[C + c8 + j] + coef[2 * C + c8 + j] * c[j];
        *reinterpret_cast<uint4*>(dp) = pack8(d);
    }
}

extern "C" int cxr_dwconv_bn_train_dc_bf16(const void* x, long x_bs, long x_rs, const float* wr, const float* coef, void* dy, long dy_bs,
                                           long dy_rs, int Bn, int C, int H, int W, int stride, int tok0, hipStream_t stream) {
    if (Bn <= 0 || (C % 8) || (stride != 1 && stride != 2)) return CXR_ERR_ARG;
    const int Ho = (H + 2 - 3) / stride + 1, Wo = (W + 2 - 3) / stride + 1;
    const long total = (long)Bn * Ho * Wo * (C / 8);
    const int grid = (int)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384);
    CXR_LAUNCH(dwconv_bn_train_dc_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x, x_bs, x_rs, wr, coef, (bf16_t*)dy, dy_bs, dy_rs,
                       Bn, C, H, W, stride, Ho, Wo, tok0);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// dw[c][t] += G[t][c]  (tap sums of dc against the activation = gradient of the raw depthwise taps)
__global__ void tap_grad_accum_kernel(const float* __restrict__ G, float* __restrict__ dw, int C) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 9 * C) return;
    const int c = i / 9, t = i % 9;
    dw[i] += G[t * C + c];
}

extern "C" int cxr_tap_grad_accum(const float* G, float* dw, int C, hipStream_t stream) {
    CXR_LAUNCH(tap_grad_accum_kernel, dim3(cdiv(9 * C, 256)), dim3(256), 0, stream, G, dw, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
