// Gather / integer / optimiser kernels of the CXRMate hot path.
#include "common.h"

extern "C" { int g_cxr_last_hip_error = 0; }
extern "C" int cxr_last_hip_error() { return g_cxr_last_hip_error; }                       // hipError_t of the last failed launch
extern "C" const char* cxr_last_hip_error_string() { return hipGetErrorString((hipError_t)g_cxr_last_hip_error); }

// ---------------------------------------------------------------------------------------------- BERT embeddings (K8)
// out[r] = LayerNorm(word[ids[r]] + type[tt[r]] + pos[pid[r]])   (TF5 modeling_bert.py:70-108); one wave per row, C = 768.
// pre-LN sum is optionally kept for the backward pass.
__global__ __launch_bounds__(256) void bert_embed_fwd_kernel(const long* __restrict__ ids, const long* __restrict__ tt, const long* __restrict__ pid,
                                                             const bf16_t* __restrict__ word, const bf16_t* __restrict__ type,
                                                             const bf16_t* __restrict__ posw, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float eps, bf16_t* __restrict__ sum_out,
                                                             bf16_t* __restrict__ out, float* __restrict__ stats, long R, int T, int pos_offset,
                                                             const uint32_t* __restrict__ drop_seed, uint32_t drop_site, uint32_t drop_thr16,
                                                             float drop_inv, int out_mt) {
    constexpr int C = 768, CH = 96;
    const uint32_t dseed = drop_thr16 ? *drop_seed : 0u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wave; row < R; row += (long)gridDim.x * 4) {
        const long id = ids[row];
        const long ty = tt ? tt[row] : 0;
        const long ps = pid ? pid[row] : (row % T) + pos_offset;
        float v[2][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = lane + i * 64;
            if (ch < CH) {
                float a[8], b[8], c[8];
                unpack8(*reinterpret_cast<const uint4*>(word + id * C + ch * 8), a);
                unpack8(*reinterpret_cast<const uint4*>(type + ty * C + ch * 8), b);
                unpack8(*reinterpret_cast<const uint4*>(posw + ps * C + ch * 8), c);
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[i][j] = (a[j] + b[j]) + c[j]; s += v[i][j]; }
                if (sum_out) *reinterpret_cast<uint4*>(sum_out + row * C + ch * 8) = pack8(v[i]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
            }
        }
        const float mean = group_sum<64>(s) * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lane + i * 64 < CH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(group_sum<64>(q) * (1.0f / C) + eps);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = lane + i * 64;
            if (ch < CH) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gamma[ch * 8 + j] + beta[ch * 8 + j];
                if (drop_thr16) {                                  // embeddings dropout (TF5:bert:106), row = (sequence row / T, pos_offset + row % T)
                    const uint32_t key = dropout_row_key(dseed, drop_site, (uint32_t)(row / T), (uint32_t)(pos_offset + (int)(row % T)));
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const uint32_t bits = dropout_pair_bits(key, (uint32_t)(ch * 8 + j) >> 1);
                        o[j] = (bits & 0xffffu) >= drop_thr16 ? o[j] * drop_inv : 0.f;
                        o[j + 1] = (bits >> 16) >= drop_thr16 ? o[j + 1] * drop_inv : 0.f;
                    }
                }
                *reinterpret_cast<uint4*>(out + (out_mt ? dal_off((int)row, ch * 8, out_mt) : row * C + ch * 8)) = pack8(o);
            }
        }
        if (stats && lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
    }
}

extern "C" int cxr_bert_embed_fwd(const long* ids, const long* tt, const long* pid, const void* word, const void* type, const void* posw,
                                  const float* gamma, const float* beta, float eps, void* sum_out, void* out, float* stats, long R, int T,
                                  int pos_offset, int C, float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int out_dal,
                                  hipStream_t stream) {
    if (R <= 0 || C != 768 || T <= 0 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed) || (out_dal && R > 64)) return CXR_ERR_ARG;
    const int out_mt = out_dal ? (cdiv(R, 16) == 3 ? 4 : cdiv(R, 16)) : 0;
    const int grid = (int)(cdiv(R, 4) < 4096 ? cdiv(R, 4) : 4096);
    CXR_LAUNCH(bert_embed_fwd_kernel, dim3(grid), dim3(256), 0, stream, ids, tt, pid, (const bf16_t*)word, (const bf16_t*)type,
                       (const bf16_t*)posw, gamma, beta, eps, (bf16_t*)sum_out, (bf16_t*)out, stats, R, T, pos_offset, drop_seed, drop_site,
                       drop_p > 0.f ? dropout_thr16(drop_p) : 0u, 1.0f / (1.0f - drop_p), out_mt);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// scatter-add of d(sum) into the three tables (fp32 atomics, lane-contiguous dwords). padding_idx row of the word table gets no gradient.
__global__ __launch_bounds__(256) void bert_embed_bwd_kernel(const bf16_t* __restrict__ dsum, const long* __restrict__ ids, const long* __restrict__ tt,
                                                             const long* __restrict__ pid, float* __restrict__ dword, float* __restrict__ dtype,
                                                             float* __restrict__ dpos, long R, int T, int pos_offset, long padding_idx) {
    constexpr int C = 768;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long row = (long)blockIdx.x * 4 + wave; row < R; row += (long)gridDim.x * 4) {
        const long id = ids[row];
        const long ty = tt ? tt[row] : 0;
        const long ps = pid ? pid[row] : (row % T) + pos_offset;
#pragma unroll
        for (int i = 0; i < C / 64; ++i) {
            const int c = lane + i * 64;
            const float g = bf2f(dsum[row * C + c]);
            if (dword && id != padding_idx) atomicAdd(dword + id * C + c, g);
            if (dtype) atomicAdd(dtype + ty * C + c, g);
            if (dpos) atomicAdd(dpos + ps * C + c, g);
        }
    }
}

extern "C" int cxr_bert_embed_bwd(const void* dsum, const long* ids, const long* tt, const long* pid, float* dword, float* dtype, float* dpos,
                                  long R, int T, int pos_offset, long padding_idx, int C, hipStream_t stream) {
    if (R <= 0 || C != 768) return CXR_ERR_ARG;
    const int grid = (int)(cdiv(R, 4) < 4096 ? cdiv(R, 4) : 4096);
    CXR_LAUNCH(bert_embed_bwd_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dsum, ids, tt, pid, dword, dtype, dpos, R, T,
                       pos_offset, padding_idx);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- token-type / position ids (K15) -- bit-exact
// reference modelling_longitudinal.py:297-338: for each special id (in order) the section after its FIRST occurrence (excluded when that
// occurrence is column 0 or the last column) gets sections[i+1]; later special ids overwrite earlier ones.
__global__ __launch_bounds__(64) void token_type_ids_kernel(const long* __restrict__ ids, long ld, int B, int T, const long* __restrict__ special,
                                                            const long* __restrict__ sections, int nspecial, long* __restrict__ out, long ldo) {
    const int b = blockIdx.x, lane = threadIdx.x;
    __shared__ int start[16];
    for (int i = 0; i < nspecial; ++i) {
        int first = 0x7fffffff;
        for (int t = lane; t < T; t += 64) if (ids[(long)b * ld + t] == special[i]) { first = t; break; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
        int col = (first == 0x7fffffff ? 0 : first) + 1;                // argmax of an all-zero row is 0
        if (lane == 0) start[i] = (col != 1 && col < T) ? col : 0x7fffffff;
    }
    __syncthreads();
    for (int t = lane; t < T; t += 64) {
        long v = sections[0];
        for (int i = 0; i < nspecial; ++i) if (t >= start[i]) v = sections[i + 1];
        out[(long)b * ldo + t] = v;
    }
}

// reference :340-364  -> [B,1]; ids[:, :-1] is searched
__global__ __launch_bounds__(64) void token_type_ids_past_kernel(const long* __restrict__ ids, long ld, int B, int T, const long* __restrict__ special,
                                                                 const long* __restrict__ sections, int nspecial, long* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    long v = sections[0];
    for (int i = 0; i < nspecial; ++i) {
        int any = 0;
        for (int t = lane; t < T - 1; t += 64) any |= (ids[(long)b * ld + t] == special[i]);
        any = __any(any);
        if (any) v = sections[i + 1];
    }
    if (lane == 0) out[b] = v;
}

extern "C" int cxr_token_type_ids(const long* ids, long ld, int B, int T, const long* special, const long* sections, int nspecial, long* out,
                                  long ldo, int past, hipStream_t stream) {
    if (B <= 0 || T <= 0 || nspecial < 0 || nspecial > 16) return CXR_ERR_ARG;
    if (past) CXR_LAUNCH(token_type_ids_past_kernel, dim3(B), dim3(64), 0, stream, ids, ld, B, T, special, sections, nspecial, out);
    else      CXR_LAUNCH(token_type_ids_kernel, dim3(B), dim3(64), 0, stream, ids, ld, B, T, special, sections, nspecial, out, ldo);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// mask = (ids != mask_token_id) ; position = relu(cumsum(mask) - 1)   (reference modelling_longitudinal.py:274-277). One wave per row.
__global__ __launch_bounds__(64) void mask_position_ids_kernel(const long* __restrict__ ids, long ld, int B, int T, long mask_token_id,
                                                               unsigned char* __restrict__ mask, long ldm, long* __restrict__ pos, long ldp) {
    const int b = blockIdx.x, lane = threadIdx.x;
    int carry = 0;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        const int m = (t < T) ? (ids[(long)b * ld + t] != mask_token_id) : 0;
        int x = m;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
        if (t < T) {
            if (mask) mask[(long)b * ldm + t] = (unsigned char)m;
            const int c = carry + x - 1;
            if (pos) pos[(long)b * ldp + t] = c > 0 ? c : 0;
        }
        carry += __shfl(x, 63, 64);
    }
}

extern "C" int cxr_mask_position_ids(const long* ids, long ld, int B, int T, long mask_token_id, void* mask, long ldm, long* pos, long ldp,
                                     hipStream_t stream) {
    if (B <= 0 || T <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(mask_position_ids_kernel, dim3(B), dim3(64), 0, stream, ids, ld, B, T, mask_token_id, (unsigned char*)mask, ldm, pos, ldp);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// encoder attention mask: (pixel_values[:, :, 0, 0, 0] != 0).repeat_interleave(tokens)   (reference modelling_multi.py:80, quirk Q3)
__global__ void image_mask_kernel(const float* __restrict__ px, long img_stride, int BN, int tokens, unsigned char* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)BN * tokens) return;
    out[i] = px[(i / tokens) * img_stride] != 0.0f;
}

extern "C" int cxr_image_mask(const float* px, long img_stride, int BN, int tokens, void* out, hipStream_t stream) {
    if (BN <= 0 || tokens <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(image_mask_kernel, dim3(cdiv((long)BN * tokens, 256)), dim3(256), 0, stream, px, img_stride, BN, tokens, (unsigned char*)out);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- AdamW (K17) + casts
// torch.optim.AdamW semantics (reference single.py:426-431: default betas/eps, weight_decay=0.01 on every parameter):
//   p *= 1 - lr*wd ; m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g^2 ; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
// One pass over the flat fp32 master buffer; also refreshes the bf16 shadow used by the MFMA kernels.
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ p16, long n, float lr, float b1, float b2, float eps, float wd,
                                                    float bc1, float bc2_sqrt, float gscale, const int* __restrict__ step_ptr) {
    if (step_ptr) {                                   // device-resident step counter (hipGraph replay keeps the host out of the loop)
        const float t = (float)(*step_ptr);
        bc1 = 1.0f - powf(b1, t);
        bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    }
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        float4 P = *reinterpret_cast<float4*>(p + i);
        const float4 G = *reinterpret_cast<const float4*>(g + i);
        float4 M = *reinterpret_cast<float4*>(m + i), V = *reinterpret_cast<float4*>(v + i);
        float* pp = &P.x; const float* gg = &G.x; float* mm = &M.x; float* vv = &V.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = gg[j] * gscale;
            pp[j] *= 1.0f - lr * wd;
            mm[j] = b1 * mm[j] + (1.0f - b1) * gr;
            vv[j] = b2 * vv[j] + (1.0f - b2) * gr * gr;
            pp[j] -= (lr / bc1) * mm[j] / (sqrtf(vv[j]) / bc2_sqrt + eps);
        }
        *reinterpret_cast<float4*>(p + i) = P;
        *reinterpret_cast<float4*>(m + i) = M;
        *reinterpret_cast<float4*>(v + i) = V;
        if (p16) { uint2 o; o.x = pack2bf(P.x, P.y); o.y = pack2bf(P.z, P.w); *reinterpret_cast<uint2*>(p16 + i) = o; }
    }
}

__global__ void increment_i32_kernel(int* p) { *p += 1; }
extern "C" int cxr_increment_i32(int* p, hipStream_t stream) {
    CXR_LAUNCH(increment_i32_kernel, dim3(1), dim3(1), 0, stream, p);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// step >= 1: host-side step count; step == 0: read the (already incremented) count from the device word `step_ptr`
extern "C" int cxr_adamw_step(float* p, const float* g, float* m, float* v, void* p16, long n, float lr, float b1, float b2, float eps, float wd,
                              int step, const int* step_ptr, float gscale, hipStream_t stream) {
    if (n <= 0 || (n % 4) || (step < 1 && !step_ptr)) return CXR_ERR_ARG;
    const float bc1 = 1.0f - powf(b1, (float)(step < 1 ? 1 : step)), bc2 = 1.0f - powf(b2, (float)(step < 1 ? 1 : step));
    const int grid = (int)(cdiv(n, 1024) < 8192 ? cdiv(n, 1024) : 8192);
    CXR_LAUNCH(adamw_kernel, dim3(grid), dim3(256), 0, stream, p, g, m, v, (bf16_t*)p16, n, lr, b1, b2, eps, wd, bc1, sqrtf(bc2), gscale,
               step < 1 ? step_ptr : (const int*)nullptr);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, long n) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4*>(in + i);
            uint2 o; o.x = pack2bf(v.x, v.y); o.y = pack2bf(v.z, v.w);
            *reinterpret_cast<uint2*>(out + i) = o;
        } else {
            for (long j = i; j < n; ++j) out[j] = f2bf(in[j]);
        }
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* __restrict__ in, float* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) out[i] = bf2f(in[i]);
}

extern "C" int cxr_cast_f32_to_bf16(const float* in, void* out, long n, hipStream_t stream) {
    if (n <= 0) return CXR_ERR_ARG;
    const int grid = (int)(cdiv(n, 1024) < 8192 ? cdiv(n, 1024) : 8192);
    CXR_LAUNCH(cast_f32_bf16_kernel, dim3(grid), dim3(256), 0, stream, in, (bf16_t*)out, n);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
extern "C" int cxr_cast_bf16_to_f32(const void* in, float* out, long n, hipStream_t stream) {
    if (n <= 0) return CXR_ERR_ARG;
    const int grid = (int)(cdiv(n, 256) < 8192 ? cdiv(n, 256) : 8192);
    CXR_LAUNCH(cast_bf16_f32_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)in, out, n);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// out[r, c] (+)= a[r, c] for bf16 rows (residual-gradient joins); rows are 8-element aligned
__global__ __launch_bounds__(256) void add_bf16_kernel(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b, long ldb,
                                                       bf16_t* __restrict__ out, long ldo, long rows, int C) {
    const int cch = C / 8;
    const long total = rows * cch;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long r = idx / cch; const int c8 = (int)(idx % cch) * 8;
        float x[8], y[8];
        unpack8(*reinterpret_cast<const uint4*>(a + r * lda + c8), x);
        unpack8(*reinterpret_cast<const uint4*>(b + r * ldb + c8), y);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] += y[j];
        *reinterpret_cast<uint4*>(out + r * ldo + c8) = pack8(x);
    }
}
extern "C" int cxr_add_bf16(const void* a, long lda, const void* b, long ldb, void* out, long ldo, long rows, int C, hipStream_t stream) {
    if (rows <= 0 || (C % 8) || (lda % 8) || (ldb % 8) || (ldo % 8)) return CXR_ERR_ARG;
    const long total = rows * (C / 8);
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(add_bf16_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, ldo, rows, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// strided bf16 row copy: out[r, :C] = in[r, :C]  (class-token concat / split, KV-cache append)
__global__ __launch_bounds__(256) void copy_rows_bf16_kernel(const bf16_t* __restrict__ in, long in_bs, long in_rs, bf16_t* __restrict__ out,
                                                             long out_bs, long out_rs, int B, int rows, int C) {
    const int cch = C / 8;
    const long total = (long)B * rows * cch;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c8 = (int)(idx % cch) * 8;
        const long t = idx / cch;
        const int r = (int)(t % rows), b = (int)(t / rows);
        *reinterpret_cast<uint4*>(out + (long)b * out_bs + (long)r * out_rs + c8) =
            *reinterpret_cast<const uint4*>(in + (long)b * in_bs + (long)r * in_rs + c8);
    }
}
extern "C" int cxr_copy_rows_bf16(const void* in, long in_bs, long in_rs, void* out, long out_bs, long out_rs, int B, int rows, int C,
                                  hipStream_t stream) {
    if (B <= 0 || rows <= 0 || (C % 8) || (in_rs % 8) || (out_rs % 8) || (in_bs % 8) || (out_bs % 8)) return CXR_ERR_ARG;
    const long total = (long)B * rows * (C / 8);
    const int grid = (int)(cdiv(total, 256) < 8192 ? cdiv(total, 256) : 8192);
    CXR_LAUNCH(copy_rows_bf16_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)in, in_bs, in_rs, (bf16_t*)out, out_bs, out_rs, B,
                       rows, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// broadcast a single fp32 row (class token parameter) into row 0 of every batch element as bf16; and its gradient (sum over batch)
__global__ void bcast_row_kernel(const float* __restrict__ row, bf16_t* __restrict__ out, long out_bs, int B, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * C) return;
    out[(i / C) * out_bs + (i % C)] = f2bf(row[i % C]);
}
extern "C" int cxr_bcast_row_f32_bf16(const float* row, void* out, long out_bs, int B, int C, hipStream_t stream) {
    CXR_LAUNCH(bcast_row_kernel, dim3(cdiv((long)B * C, 256)), dim3(256), 0, stream, row, (bf16_t*)out, out_bs, B, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
__global__ void sum_row0_kernel(const bf16_t* __restrict__ in, long in_bs, float* __restrict__ out, int B, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += bf2f(in[(long)b * in_bs + c]);
    out[c] += s;
}
extern "C" int cxr_sum_row0_bf16_f32(const void* in, long in_bs, float* out, int B, int C, hipStream_t stream) {
    CXR_LAUNCH(sum_row0_kernel, dim3(cdiv(C, 128)), dim3(128), 0, stream, (const bf16_t*)in, in_bs, out, B, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// dx = dy * GELU'(u)   (LM-head transform backward, TF5 modeling_bert.py:466-481)
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ u, bf16_t* __restrict__ dx, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        float a[8], b[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + i * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(u + i * 8), b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] *= gelu_grad_f(b[j]);
        *reinterpret_cast<uint4*>(dx + i * 8) = pack8(a);
    }
}
extern "C" int cxr_gelu_bwd_bf16(const void* dy, const void* u, void* dx, long n, hipStream_t stream) {
    if (n <= 0 || (n % 8)) return CXR_ERR_ARG;
    const int grid = (int)(cdiv(n / 8, 256) < 8192 ? cdiv(n / 8, 256) : 8192);
    CXR_LAUNCH(gelu_bwd_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)dy, (const bf16_t*)u, (bf16_t*)dx, n / 8);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// cosine similarity of fp32 rows: out[r] = <a_r, b_r> / (max(|a_r|, eps) * max(|b_r|, eps))   (torch.nn.functional.cosine_similarity,
// reference tools/rewards/cxrbert.py:66-71). One wave per row.
__global__ __launch_bounds__(256) void cosine_rows_kernel(const float* __restrict__ a, long lda, const float* __restrict__ b, long ldb,
                                                          float* __restrict__ out, long R, int C, float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 4 + wave;
    float ab = 0.f, aa = 0.f, bb = 0.f;
    if (r < R)
        for (int c = lane; c < C; c += 64) { const float x = a[r * lda + c], y = b[r * ldb + c]; ab += x * y; aa += x * x; bb += y * y; }
    ab = group_sum<64>(ab); aa = group_sum<64>(aa); bb = group_sum<64>(bb);
    if (r < R && lane == 0) out[r] = ab / (fmaxf(sqrtf(aa), eps) * fmaxf(sqrtf(bb), eps));
}
extern "C" int cxr_cosine_rows_f32(const float* a, long lda, const float* b, long ldb, float* out, long R, int C, float eps, hipStream_t stream) {
    if (R <= 0 || C <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(cosine_rows_kernel, dim3(cdiv(R, 4)), dim3(256), 0, stream, a, lda, b, ldb, out, R, C, eps);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- input pipeline tail on the GPU
// ToTensor + Normalize + pad_sequence of the reference's collate (modules/lightning_modules/single.py:248-262 test/train transforms,
// multi.py:155-164 collate_fn): decoded, resized and cropped uint8 HWC images of all studies of a batch, packed back to back, become the
// fp32 NCHW `images` tensor [B, Nmax, 3, H, W]; studies with fewer than Nmax images are padded with 0.0 images (which is what the
// cross-attention mask keys on, quirk Q3). Ships 1 byte per sample over PCIe instead of 4.
__global__ __launch_bounds__(256) void pixels_u8_to_f32_kernel(const unsigned char* __restrict__ src, const long* __restrict__ first_image /*[B+1]*/,
                                                               float* __restrict__ dst, int B, int Nmax, int H, int W, float3 scale, float3 shift) {
    const long hw = (long)H * W;
    const long total = (long)B * Nmax * hw;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pix = idx % hw;
        const int n = (int)((idx / hw) % Nmax), b = (int)(idx / (hw * Nmax));
        float* o = dst + ((long)(b * Nmax + n) * 3) * hw + pix;
        const long img = first_image[b] + n;
        if (img < first_image[b + 1]) {
            const unsigned char* p = src + (img * hw + pix) * 3;
            o[0] = p[0] * scale.x + shift.x; o[hw] = p[1] * scale.y + shift.y; o[2 * hw] = p[2] * scale.z + shift.z;
        } else {
            o[0] = 0.f; o[hw] = 0.f; o[2 * hw] = 0.f;
        }
    }
}

extern "C" int cxr_pixels_u8_to_f32(const void* src, const long* first_image, float* dst, int B, int Nmax, int H, int W, float mean0,
                                    float mean1, float mean2, float std0, float std1, float std2, hipStream_t stream) {
    if (B <= 0 || Nmax <= 0 || H <= 0 || W <= 0 || std0 == 0.f || std1 == 0.f || std2 == 0.f) return CXR_ERR_ARG;
    // (u/255 - mean)/std = u * (1/(255 std)) - mean/std
    const float3 scale = make_float3(1.0f / (255.0f * std0), 1.0f / (255.0f * std1), 1.0f / (255.0f * std2));
    const float3 shift = make_float3(-mean0 / std0, -mean1 / std1, -mean2 / std2);
    const long total = (long)B * Nmax * H * W;
    CXR_LAUNCH(pixels_u8_to_f32_kernel, dim3((unsigned)(cdiv(total, 256) < 16384 ? cdiv(total, 256) : 16384)), dim3(256), 0, stream,
                       (const unsigned char*)src, first_image, dst, B, Nmax, H, W, scale, shift);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// argmax over consecutive column segments of each row: out[r][s] = argmax x[r][off[s] .. off[s+1])  (lowest index wins ties, like torch.argmax)
// -- the 14 CheXbert heads (13 x 4 classes + 1 x 2, reference tools/chexbert.py:74-81) evaluated as ONE GEMM + this kernel.
__global__ void segment_argmax_kernel(const float* __restrict__ x, long ld, const int* __restrict__ off, int nseg, long* __restrict__ out, long R) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * nseg) return;
    const long r = i / nseg;
    const int s = (int)(i % nseg);
    int best = off[s];
    float bv = x[r * ld + best];
    for (int c = off[s] + 1; c < off[s + 1]; ++c) {
        const float v = x[r * ld + c];
        if (v > bv) { bv = v; best = c; }
    }
    out[i] = best - off[s];
}

extern "C" int cxr_segment_argmax_f32(const float* x, long ld, const int* offsets, int nseg, long* out, long R, hipStream_t stream) {
    if (R <= 0 || nseg <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(segment_argmax_kernel, dim3(cdiv(R * nseg, 256)), dim3(256), 0, stream, x, ld, offsets, nseg, out, R);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- cached-decode step inputs, one launch
// Everything `prepare_inputs_for_generation` assembles for a cached step (reference modules/transformers/longitudinal_model/
// modelling_longitudinal.py:251-295 and the single/multi variants): from the running token buffer ids[r, strip:cur]
//   new_id[r]    = ids[r, cur-1]                                      (input_ids[:, -1:])
//   tt[r]        = token_ids_to_token_type_ids_past(ids[r, strip:cur])  (:340-364; separator sets may differ between the two row halves:
//                  the sampled and the greedy decode of one SCST step run as one batch)
//   mask[r, j]   = ids[r, strip+j] != mask_token_id,  pos[r] = relu(cumsum(mask) - 1)[-1]     (:274-277; skipped when mask == NULL)
// tt / pos are also appended to per-row histories (column cur) for the teacher-forced re-scoring of the sampled rows.
__global__ __launch_bounds__(64) void decode_step_inputs_kernel(const long* __restrict__ ids, long ld, int rows, int strip, int cur,
                                                                const long* __restrict__ special0, int n0, const long* __restrict__ special1, int n1,
                                                                const long* __restrict__ sections, int half_rows, long mask_token_id,
                                                                long* __restrict__ new_id, long* __restrict__ tt, long* __restrict__ pos,
                                                                unsigned char* __restrict__ mask, long ldm, long* __restrict__ tt_hist,
                                                                long* __restrict__ pos_hist, long ldh) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const long* row = ids + (long)r * ld + strip;
    const int T = cur - strip;
    const long* special = r < half_rows ? special0 : special1;
    const int ns = r < half_rows ? n0 : n1;
    long v = sections[0];
    for (int i = 0; i < ns; ++i) {                            // separators searched in all but the last position, later separators win
        int any = 0;
        for (int t = lane; t < T - 1; t += 64) any |= (row[t] == special[i]);
        if (__any(any)) v = sections[i + 1];
    }
    int count = 0;
    if (mask) {
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            const int m = t < T ? (row[t] != mask_token_id) : 0;
            if (t < T) mask[(long)r * ldm + t] = (unsigned char)m;
            count += __popcll(__ballot(m));
        }
    }
    if (lane == 0) {
        new_id[r] = row[T - 1];
        tt[r] = v;
        if (tt_hist) tt_hist[(long)r * ldh + cur] = v;
        if (mask) {
            const long p = count > 0 ? count - 1 : 0;
            pos[r] = p;
            if (pos_hist) pos_hist[(long)r * ldh + cur] = p;
        }
    }
}

// The same input assembly FUSED with the BERT embeddings of the new token (TF5:bert:70-108: word + token-type + position gather, LayerNorm,
// dropout): one wave per row, the row's ids are read once into registers, the embedding output goes straight to the decode activation layout
// of cxr_dec_gemm_bf16. One launch per token instead of two.
__global__ __launch_bounds__(64) void decode_step_embed_kernel(const long* __restrict__ ids, long ld, int rows, int strip, int cur,
                                                               const long* __restrict__ special0, int n0, const long* __restrict__ special1, int n1,
                                                               const long* __restrict__ sections, int half_rows, long mask_token_id,
                                                               long* __restrict__ new_id, long* __restrict__ tt, long* __restrict__ pos,
                                                               unsigned char* __restrict__ mask, long ldm, long* __restrict__ tt_hist,
                                                               long* __restrict__ pos_hist, long ldh, const bf16_t* __restrict__ word,
                                                               const bf16_t* __restrict__ type, const bf16_t* __restrict__ posw,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               bf16_t* __restrict__ out, int out_mt, const uint32_t* __restrict__ drop_seed,
                                                               uint32_t drop_site, uint32_t drop_thr16, float drop_inv) {
    constexpr int C = 768, CH = 96, PER = 8;                 // up to 64 * PER = 512 tokens per row (max_position_embeddings)
    const int r = blockIdx.x, lane = threadIdx.x;
    const long* row = ids + (long)r * ld + strip;
    const int T = cur - strip;
    long tok[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) { const int t = j * 64 + lane; tok[j] = row[t < T ? t : T - 1]; }
    const long* special = r < half_rows ? special0 : special1;
    const int ns = r < half_rows ? n0 : n1;
    long sp[4] = {special[0], special[ns > 1 ? 1 : 0], special[ns > 2 ? 2 : 0], special[ns > 3 ? 3 : 0]};
    long v = sections[0];
#pragma unroll
    for (int i = 0; i < 4; ++i) {                           // separators searched in all but the last position, later separators win
        int any = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) any |= (j * 64 + lane < T - 1) && (tok[j] == sp[i]);
        if (i < ns && __any(any)) v = sections[i + 1];
    }
    int count = 0;
    if (mask) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int t = j * 64 + lane;
            const int m = t < T ? (tok[j] != mask_token_id) : 0;
            if (t < T) mask[(long)r * ldm + t] = (unsigned char)m;
            count += __popcll(__ballot(m));
        }
    }
    const long last = row[T - 1];
    const long p = mask ? (count > 0 ? count - 1 : 0) : (long)(T - 1);       // longitudinal: relu(cumsum(mask) - 1)[-1]; else the absolute position
    if (lane == 0) {
        new_id[r] = last;
        tt[r] = v;
        if (tt_hist) tt_hist[(long)r * ldh + cur] = v;
        if (mask) {
            pos[r] = p;
            if (pos_hist) pos_hist[(long)r * ldh + cur] = p;
        }
    }
    // ---- embeddings of the new token
    const uint32_t dseed = drop_thr16 ? *drop_seed : 0u;
    float x[2][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = lane + i * 64, chc = ch < CH ? ch : CH - 1;
        float a[8], b[8], c[8];
        unpack8(*reinterpret_cast<const uint4*>(word + last * C + chc * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(type + v * C + chc * 8), b);
        unpack8(*reinterpret_cast<const uint4*>(posw + p * C + chc * 8), c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[i][j] = ch < CH ? (a[j] + b[j]) + c[j] : 0.f; s += x[i][j]; }
    }
    const float mean = group_sum<64>(s) * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = (lane + i * 64 < CH) ? x[i][j] - mean : 0.f; q += d * d; }
    const float rstd = rsqrtf(group_sum<64>(q) * (1.0f / C) + eps);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ch = lane + i * 64;
        if (ch < CH) {
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (x[i][j] - mean) * rstd * gamma[ch * 8 + j] + beta[ch * 8 + j];
            if (drop_thr16) {                                  // embeddings dropout (TF5:bert:106): (sequence r, absolute position T-1)
                const uint32_t key = dropout_row_key(dseed, drop_site, (uint32_t)r, (uint32_t)(T - 1));
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    const uint32_t bits = dropout_pair_bits(key, (uint32_t)(ch * 8 + j) >> 1);
                    o[j] = (bits & 0xffffu) >= drop_thr16 ? o[j] * drop_inv : 0.f;
                    o[j + 1] = (bits >> 16) >= drop_thr16 ? o[j + 1] * drop_inv : 0.f;
                }
            }
            *reinterpret_cast<uint4*>(out + (out_mt ? dal_off(r, ch * 8, out_mt) : (long)r * C + ch * 8)) = pack8(o);
        }
    }
}

extern "C" int cxr_decode_step_embed(const long* ids, long ld, int rows, int strip, int cur, const long* special0, int n0, const long* special1,
                                     int n1, const long* sections, int half_rows, long mask_token_id, long* new_id, long* tt, long* pos,
                                     void* mask, long ldm, long* tt_hist, long* pos_hist, long ldh, const void* word, const void* type,
                                     const void* posw, const float* gamma, const float* beta, float eps, void* out, int out_dal, float drop_p,
                                     const unsigned int* drop_seed, unsigned int drop_site, hipStream_t stream) {
    if (rows <= 0 || cur - strip < 1 || cur - strip > 512 || !special0 || !special1 || !sections || !new_id || !tt || (mask && !pos)) return CXR_ERR_ARG;
    if (n0 < 1 || n0 > 4 || n1 < 1 || n1 > 4 || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed) || (out_dal && rows > 64)) return CXR_ERR_ARG;
    const int out_mt = out_dal ? (cdiv(rows, 16) == 3 ? 4 : cdiv(rows, 16)) : 0;
    CXR_LAUNCH(decode_step_embed_kernel, dim3(rows), dim3(64), 0, stream, ids, ld, rows, strip, cur, special0, n0, special1, n1, sections, half_rows,
                       mask_token_id, new_id, tt, pos, (unsigned char*)mask, ldm, tt_hist, pos_hist, ldh, (const bf16_t*)word, (const bf16_t*)type,
                       (const bf16_t*)posw, gamma, beta, eps, (bf16_t*)out, out_mt, drop_seed, drop_site,
                       drop_p > 0.f ? dropout_thr16(drop_p) : 0u, 1.0f / (1.0f - drop_p));
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_decode_step_inputs(const long* ids, long ld, int rows, int strip, int cur, const long* special0, int n0, const long* special1,
                                      int n1, const long* sections, int half_rows, long mask_token_id, long* new_id, long* tt, long* pos,
                                      void* mask, long ldm, long* tt_hist, long* pos_hist, long ldh, hipStream_t stream) {
    if (rows <= 0 || cur - strip < 1 || !special0 || !special1 || !sections || !new_id || !tt || (mask && !pos)) return CXR_ERR_ARG;
    CXR_LAUNCH(decode_step_inputs_kernel, dim3(rows), dim3(64), 0, stream, ids, ld, rows, strip, cur, special0, n0, special1, n1, sections, half_rows,
                       mask_token_id, new_id, tt, pos, (unsigned char*)mask, ldm, tt_hist, pos_hist, ldh);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
