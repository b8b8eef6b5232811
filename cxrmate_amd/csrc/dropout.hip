// Train-mode stochastic regularisers of the hot path: nn.Dropout on hidden states (TF5 modeling_bert.py:106,298,464 -- embeddings, attention
// output, FFN output), DropPath per image (TF5 modeling_cvt.py:297-316) and the mask generator the parity tests feed to the CPU oracle.
// The keep decision is a pure function of (seed, site, b, t, col) -- common.h -- so backward and the teacher-forced re-scoring of a sampled
// sequence regenerate the forward masks instead of storing them.
#include "common.h"

// out[r, :] = (resid ? resid[r, :] : 0) + f(r, :) * y[r, :]
//   element mode (row_scale == NULL): f = keep(seed, site, b, t, col) / (1 - p),  b = r / rows_per_b, t = t0 + r % rows_per_b
//   row mode:                         f = row_scale[r / rows_per_b]               (DropPath: 0 or 1/keep_prob per image)
__global__ __launch_bounds__(256) void dropout_add_kernel(const bf16_t* __restrict__ y, long ldy, const bf16_t* __restrict__ resid, long ldr,
                                                          bf16_t* __restrict__ out, long ldo, long R, int C, uint32_t thr16, float inv_keep,
                                                          const uint32_t* __restrict__ seed_ptr, uint32_t site, int rows_per_b, int t0,
                                                          const float* __restrict__ row_scale) {
    const int cch = C / 8;
    const long total = R * cch;
    const uint32_t seed = seed_ptr ? *seed_ptr : 0u;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / cch;
        const int c8 = (int)(idx % cch) * 8;
        float v[8], f[8];
        unpack8(*reinterpret_cast<const uint4*>(y + r * ldy + c8), v);
        const uint32_t b = (uint32_t)(r / rows_per_b), t = (uint32_t)(t0 + (int)(r % rows_per_b));
        if (row_scale) {
            const float s = row_scale[b];
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = s;
        } else {
            const uint32_t key = dropout_row_key(seed, site, b, t);
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const uint32_t bits = dropout_pair_bits(key, (uint32_t)(c8 + j) >> 1);
                f[j] = (bits & 0xffffu) >= thr16 ? inv_keep : 0.f;
                f[j + 1] = (bits >> 16) >= thr16 ? inv_keep : 0.f;
            }
        }
        float o[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (resid) unpack8(*reinterpret_cast<const uint4*>(resid + r * ldr + c8), o);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] += f[j] * v[j];
        *reinterpret_cast<uint4*>(out + r * ldo + c8) = pack8(o);
    }
}

extern "C" int cxr_dropout_add_bf16(const void* y, long ldy, const void* resid, long ldr, void* out, long ldo, long R, int C, float p,
                                    const unsigned int* seed, unsigned int site, int rows_per_b, int t0, const float* row_scale,
                                    hipStream_t stream) {
    if (R <= 0 || (C % 8) || rows_per_b <= 0 || p < 0.f || p >= 1.f || (!row_scale && !seed)) return CXR_ERR_ARG;
    const long total = R * (C / 8);
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(dropout_add_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)y, ldy, (const bf16_t*)resid, ldr, (bf16_t*)out, ldo, R, C,
                       dropout_thr16(p), 1.0f / (1.0f - p), seed, site, rows_per_b, t0, row_scale);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// keep mask of one site as bytes [R, C] (and/or the fp32 factor keep/(1-p)): what the kernels above / the attention kernels apply.
__global__ __launch_bounds__(256) void dropout_mask_kernel(unsigned char* __restrict__ mask, float* __restrict__ factor, long R, int C, uint32_t thr16,
                                                           float inv_keep, const uint32_t* __restrict__ seed_ptr, uint32_t site, int rows_per_b, int t0) {
    const long total = R * C;
    const uint32_t seed = *seed_ptr;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / C;
        const uint32_t c = (uint32_t)(idx % C);
        const bool keep = dropout_keep(dropout_row_key(seed, site, (uint32_t)(r / rows_per_b), (uint32_t)(t0 + (int)(r % rows_per_b))), c, thr16);
        if (mask) mask[idx] = keep ? 1 : 0;
        if (factor) factor[idx] = keep ? inv_keep : 0.f;
    }
}

extern "C" int cxr_dropout_mask(unsigned char* mask, float* factor, long R, int C, float p, const unsigned int* seed, unsigned int site,
                                int rows_per_b, int t0, hipStream_t stream) {
    if (R <= 0 || C <= 0 || rows_per_b <= 0 || p < 0.f || p >= 1.f || !seed) return CXR_ERR_ARG;
    const long total = R * C;
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(dropout_mask_kernel, dim3(grid), dim3(256), 0, stream, mask, factor, R, C, dropout_thr16(p), 1.0f / (1.0f - p), seed, site,
                       rows_per_b, t0);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// DropPath factors of `nsites` consecutive sites in one launch: factor[s][b] = keep(seed, site0 + s, sequence b, position 0, column 0) / (1 - p)
// -- exactly what cxr_dropout_mask(R = Bn, C = 1, rows_per_b = 1, site = site0 + s) writes, for all layers of a CvT stage at once.
__global__ __launch_bounds__(256) void dropout_site_factors_kernel(float* __restrict__ factor, int nsites, int Bn, uint32_t thr16, float inv_keep,
                                                                   const uint32_t* __restrict__ seed_ptr, uint32_t site0) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nsites * Bn) return;
    const int s = idx / Bn, b = idx - s * Bn;
    factor[idx] = dropout_keep(dropout_row_key(*seed_ptr, site0 + (uint32_t)s, (uint32_t)b, 0u), 0u, thr16) ? inv_keep : 0.f;
}

extern "C" int cxr_dropout_site_factors(float* factor, int nsites, int Bn, float p, const unsigned int* seed, unsigned int site0, hipStream_t stream) {
    if (!factor || nsites <= 0 || Bn <= 0 || p < 0.f || p >= 1.f || !seed) return CXR_ERR_ARG;
    CXR_LAUNCH(dropout_site_factors_kernel, dim3(cdiv(nsites * Bn, 256)), dim3(256), 0, stream, factor, nsites, Bn, dropout_thr16(p), 1.0f / (1.0f - p),
               seed, site0);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
