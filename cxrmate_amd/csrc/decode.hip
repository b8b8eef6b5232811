// Autoregressive-decode helpers: beam reordering of the self-attention KV cache and the
// top-2*beams continuation search of beam search (TF5 generation/utils.py:3388-3435).
#include "common.h"

// out[b, r, :] = in[idx[b], r, :]  for r < rows   (cache reorder after a beam step; idx is int64)
__global__ __launch_bounds__(256) void gather_batch_kernel(const bf16_t* __restrict__ in, long in_bs, long in_rs, bf16_t* __restrict__ out,
                                                           long out_bs, long out_rs, const long* __restrict__ idx, int B, int rows, int C) {
    const int cch = C / 8;
    const long total = (long)B * rows * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % cch) * 8;
        const long t = i / cch;
        const int r = (int)(t % rows), b = (int)(t / rows);
        *reinterpret_cast<uint4*>(out + (long)b * out_bs + (long)r * out_rs + c8) =
            *reinterpret_cast<const uint4*>(in + idx[b] * in_bs + (long)r * in_rs + c8);
    }
}
extern "C" int cxr_gather_batch_bf16(const void* in, long in_bs, long in_rs, void* out, long out_bs, long out_rs, const long* idx, int B,
                                     int rows, int C, hipStream_t stream) {
    if (B <= 0 || rows <= 0 || (C % 8)) return CXR_ERR_ARG;
    const long total = (long)B * rows * (C / 8);
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(gather_batch_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)in, in_bs, in_rs, (bf16_t*)out, out_bs, out_rs, idx,
                       B, rows, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// Per study: the K largest entries of x[b, 0 .. n) (n = beams*V accumulated log-probs), in descending order, ties -> lowest index
// (torch.topk order on distinct values). K <= 16. One block per study; K rounds of a block-wide arg-max with exclusion.
__global__ __launch_bounds__(256) void topk_rows_kernel(const float* __restrict__ x, long ld, int n, int K, float* __restrict__ vals,
                                                        long* __restrict__ inds) {
    __shared__ float shv[4];
    __shared__ int shi[4];
    __shared__ int chosen[16];
    const float* row = x + (long)blockIdx.x * ld;
    for (int k = 0; k < K; ++k) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int i = threadIdx.x; i < n; i += 256) {
            const float a = row[i];
            bool skip = false;
            for (int j = 0; j < k; ++j) skip |= (chosen[j] == i);
            if (!skip && (a > best || (a == best && i < bi))) { best = a; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = best; shi[threadIdx.x >> 6] = bi; }
        __syncthreads();
        if (threadIdx.x == 0) {
            best = shv[0]; bi = shi[0];
            for (int w = 1; w < 4; ++w) if (shv[w] > best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
            chosen[k] = bi;
            vals[(long)blockIdx.x * K + k] = best;
            inds[(long)blockIdx.x * K + k] = bi;
        }
        __syncthreads();
    }
}
extern "C" int cxr_topk_rows(const float* x, long ld, long R, int n, int K, float* vals, long* inds, hipStream_t stream) {
    if (R <= 0 || n <= 0 || K <= 0 || K > 16 || K > n) return CXR_ERR_ARG;
    CXR_LAUNCH(topk_rows_kernel, dim3((unsigned)R), dim3(256), 0, stream, x, ld, n, K, vals, inds);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
