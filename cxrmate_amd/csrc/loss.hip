// Vocabulary-row kernels: cross-entropy & REINFORCE loss + logits gradient, exact top-k threshold,
// greedy argmax (lowest index wins ties, like torch.argmax), and top-k multinomial sampling. One 256-thread block per logits row
// (V = 30000 fp32 = 120 KB: first pass from HBM, later passes from L2).
#include "common.h"
#ifndef CXR_STAMP
#define CXR_STAMP(i)            /* scripts/lab defines it to record s_memrealtime per wave; nothing in the product build */
#endif

__device__ __forceinline__ float block_max(float v, float* sh) {
    v = group_max<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}
__device__ __forceinline__ float block_sum(float v, float* sh) {
    v = group_sum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

// row_loss[r] = logsumexp_kept(logits[r]) - logits[r][label]   (0 for ignored rows)
// dlogits[r][v] = row_w[r] * (softmax_kept - onehot)            (0 for ignored rows and for filtered entries)
// "kept" = entries >= thr[r] when thr != null (top-k filtered distribution of the SCST sampler, reference scst/gt_prompt.py:189,230-235).
// The label of a row IS in its kept set (the reference's scores are the ones the token was drawn from): where the re-scored logit of the drawn
// token fell below the re-computed threshold (last bf16 bits at the edge of the top-k) it takes the place of the k-th entry -- kept = { > thr }
// + label, still k finite entries -- see kept_threshold().
__device__ __forceinline__ float kept_threshold(const float* __restrict__ thr, long r, float xlab) {
    if (!thr) return -INFINITY;
    const float t = thr[r];
    return xlab < t ? nextafterf(t, INFINITY) : t;
}
__global__ __launch_bounds__(256) void softmax_ce_kernel(const float* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                         long ignore_index, const float* __restrict__ thr, const float* __restrict__ row_w,
                                                         float* __restrict__ row_loss, bf16_t* __restrict__ dlogits, long lddl, int V) {
    __shared__ float sh[4];
    const long r = blockIdx.x;
    const float* x = logits + r * ld;
    const long label = labels[r];
    const bool ignored = label == ignore_index;
    if (ignored) {
        if (row_loss && threadIdx.x == 0) row_loss[r] = 0.f;
        if (dlogits) for (int v = threadIdx.x * 8; v < (int)lddl; v += 2048)
            *reinterpret_cast<uint4*>(dlogits + r * lddl + v) = make_uint4(0, 0, 0, 0);
        return;
    }
    const float xlab = x[label];
    const float t = kept_threshold(thr, r, xlab);
    float mx = xlab;
    for (int v = threadIdx.x; v < V; v += 256) { const float a = x[v]; if (a >= t) mx = fmaxf(mx, a); }
    mx = block_max(mx, sh);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) { const float a = x[v]; if (a >= t || v == (int)label) s += __expf(a - mx); }
    s = block_sum(s, sh);
    const float lse = mx + __logf(s);
    if (row_loss && threadIdx.x == 0) row_loss[r] = lse - x[label];
    if (dlogits) {
        const float w = row_w[r], inv = 1.0f / s;
        for (int v = threadIdx.x * 8; v < (int)lddl; v += 2048) {      // columns [V, lddl) are zero padding (K of the next GEMM)
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int vv = v + j;
                float gv = 0.f;
                if (vv < V) { const float a = x[vv]; if (a >= t || vv == label) gv = __expf(a - mx) * inv; if (vv == label) gv -= 1.0f; }
                o[j] = gv * w;
            }
            *reinterpret_cast<uint4*>(dlogits + r * lddl + v) = pack8(o);
        }
    }
}

// Register-resident variant (V <= 32768, ld % 4 == 0): 1024 threads hold the whole row (<= 8 float4 each), so the logits are read from HBM
// exactly once (the 256-thread kernel above re-reads the 120-KB row for the sum and for the gradient).
// BF16IN: the logits are bf16 (what the LM head produces under the reference's bf16 autocast; the softmax still runs in fp32): half the bytes of
// the largest tensor of a training step, written by the LM-head GEMM and read here.
template <bool BF16IN>
__device__ __forceinline__ void softmax_ce_reg_body(const void* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                    long ignore_index, const float* __restrict__ thr, const float* __restrict__ row_w,
                                                    float* __restrict__ row_loss, bf16_t* __restrict__ dlogits, long lddl, int V, float* sh, float& xl) {
    const long r = blockIdx.x;
    const float* x = reinterpret_cast<const float*>(logits) + r * ld;
    const bf16_t* x16 = reinterpret_cast<const bf16_t*>(logits) + r * ld;
    const long label = labels[r];
    const int tid = threadIdx.x;
    if (label == ignore_index) {
        if (row_loss && tid == 0) row_loss[r] = 0.f;
        if (dlogits) for (int v = tid * 8; v < (int)lddl; v += 8192)
            *reinterpret_cast<uint4*>(dlogits + r * lddl + v) = make_uint4(0, 0, 0, 0);
        return;
    }
    const float xlab = thr ? (BF16IN ? bf2f(x16[label]) : x[label]) : -INFINITY;
    const float t = kept_threshold(thr, r, xlab);
    // thread owns columns [8*tid + 8192*i, +8): two float4 per chunk, 4 chunks
    float4 a[4][2];
    float mx = xlab;                                                         // the label is always kept
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tid * 8 + 8192 * i;
        float f8[8];
        if (BF16IN) {
#pragma unroll
            for (int j = 0; j < 8; ++j) f8[j] = -INFINITY;
            if (v + 7 < V) unpack8(*reinterpret_cast<const uint4*>(x16 + v), f8);
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) if (v + j < V) f8[j] = bf2f(x16[v + j]);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 q = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            const int vv = v + 4 * h;
            if (BF16IN) q = make_float4(f8[4 * h], f8[4 * h + 1], f8[4 * h + 2], f8[4 * h + 3]);
            else if (vv + 3 < V) q = *reinterpret_cast<const float4*>(x + vv);
            else {
                if (vv < V) q.x = x[vv];
                if (vv + 1 < V) q.y = x[vv + 1];
                if (vv + 2 < V) q.z = x[vv + 2];
            }
            a[i][h] = q;
            if (q.x >= t) mx = fmaxf(mx, q.x);
            if (q.y >= t) mx = fmaxf(mx, q.y);
            if (q.z >= t) mx = fmaxf(mx, q.z);
            if (q.w >= t) mx = fmaxf(mx, q.w);
        }
    }
    mx = group_max<64>(mx);
    if ((tid & 63) == 0) sh[tid >> 6] = mx;
    __syncthreads();
    mx = sh[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, sh[w]);
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float* q = reinterpret_cast<float*>(&a[i][h]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int vv = tid * 8 + 8192 * i + 4 * h + j;
                const float e = (q[j] >= t || vv == (int)label) ? __expf(q[j] - mx) : 0.f;       // -inf padding / filtered entries -> 0
                if (vv == (int)label) xl = q[j];
                q[j] = e;
            }
            s += (q[0] + q[1]) + (q[2] + q[3]);
        }
    s = group_sum<64>(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += sh[w];
    if (row_loss && tid == 0) row_loss[r] = mx + __logf(s) - xl;
    if (dlogits) {
        const float w = row_w[r], inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid * 8 + 8192 * i;
            if (v >= (int)lddl) continue;                                   // columns [V, lddl) are zero padding (K of the next GEMM)
            float o[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float* q = reinterpret_cast<const float*>(&a[i][h]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int vv = v + 4 * h + j;
                    float gv = q[j] * inv;
                    if (vv == (int)label) gv -= 1.0f;
                    o[4 * h + j] = vv < V ? gv * w : 0.f;
                }
            }
            *reinterpret_cast<uint4*>(dlogits + r * lddl + v) = pack8(o);
        }
    }
}

template <bool BF16IN>
__global__ __launch_bounds__(1024) void softmax_ce_reg_kernel(const void* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                                              long ignore_index, const float* __restrict__ thr,
                                                                              const float* __restrict__ row_w, float* __restrict__ row_loss,
                                                                              bf16_t* __restrict__ dlogits, long lddl, int V) {
    __shared__ float sh[16];
    __shared__ float xl;
    softmax_ce_reg_body<BF16IN>(logits, ld, labels, ignore_index, thr, row_w, row_loss, dlogits, lddl, V, sh, xl);
}

// bf16 rows of a multiple of 8 columns (the training step's 30000-column logits): the row stays PACKED in registers (4 x 16 bytes per thread) and the
// exponentials are recomputed for the gradient instead of kept as fp32 -- 64 registers = 8 waves per SIMD, so TWO 1024-thread workgroups share a CU and
// one's load / store phase runs under the other's reductions (the fp32-resident version: 72 registers, one workgroup per CU). All four loads are
// unconditional (chunks past the row re-read its last one and are masked) and in flight together. Same arithmetic as softmax_ce_reg_kernel<true>.
__global__ __launch_bounds__(1024, 8) void softmax_ce_bf16row_kernel(const bf16_t* __restrict__ logits, long ld, const long* __restrict__ labels,
                                                                     long ignore_index, const float* __restrict__ thr, const float* __restrict__ row_w,
                                                                     float* __restrict__ row_loss, bf16_t* __restrict__ dlogits, long lddl, int V) {
    __shared__ float sh[16];
    __shared__ float xl;
    const long r = blockIdx.x;
    const bf16_t* x16 = logits + r * ld;
    const long label = labels[r];
    const int tid = threadIdx.x;
    if (label == ignore_index) {
        if (row_loss && tid == 0) row_loss[r] = 0.f;
        if (dlogits) for (int v = tid * 8; v < (int)lddl; v += 8192)
            *reinterpret_cast<uint4*>(dlogits + r * lddl + v) = make_uint4(0, 0, 0, 0);
        return;
    }
    const float xlab = thr ? bf2f(x16[label]) : -INFINITY;
    const float t = kept_threshold(thr, r, xlab);
    uint4 raw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tid * 8 + 8192 * i;
        raw[i] = *reinterpret_cast<const uint4*>(x16 + (v < V ? v : V - 8));
    }
    float mx = xlab;                                                         // the label is always kept
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (tid * 8 + 8192 * i >= V) continue;
        float f[8];
        unpack8(raw[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (f[j] >= t) mx = fmaxf(mx, f[j]);
    }
    mx = group_max<64>(mx);
    if ((tid & 63) == 0) sh[tid >> 6] = mx;
    __syncthreads();
    mx = sh[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, sh[w]);
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tid * 8 + 8192 * i;
        if (v >= V) continue;
        float f[8], e[8];
        unpack8(raw[i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            e[j] = (f[j] >= t || v + j == (int)label) ? __expf(f[j] - mx) : 0.f;
            if (v + j == (int)label) xl = f[j];
        }
        s += (e[0] + e[1]) + (e[2] + e[3]);                              // (the summation order of softmax_ce_reg_kernel: bit-identical results)
        s += (e[4] + e[5]) + (e[6] + e[7]);
    }
    s = group_sum<64>(s);
    if ((tid & 63) == 0) sh[tid >> 6] = s;
    __syncthreads();
    s = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += sh[w];
    if (row_loss && tid == 0) row_loss[r] = mx + __logf(s) - xl;
    if (dlogits) {
        const float w = row_w[r], inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid * 8 + 8192 * i;
            if (v >= (int)lddl) continue;                                   // columns [V, lddl) are zero padding (K of the next GEMM)
            float f[8], o[8];
            unpack8(raw[i], f);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float gv = ((f[j] >= t || v + j == (int)label) ? __expf(f[j] - mx) : 0.f) * inv;
                if (v + j == (int)label) gv -= 1.0f;
                o[j] = v < V ? gv * w : 0.f;
            }
            *reinterpret_cast<uint4*>(dlogits + r * lddl + v) = pack8(o);
        }
    }
}

extern "C" int cxr_softmax_ce(const void* logits, long ld, const long* labels, long ignore_index, const float* thr, const float* row_w,
                              float* row_loss, void* dlogits, long lddl, long R, int V, int logits_bf16, hipStream_t stream) {
    if (R <= 0 || V <= 0 || (dlogits && (!row_w || (lddl % 8)))) return CXR_ERR_ARG;
    if (logits_bf16) {
        if (V > 32768 || lddl > 32768 || (ld % 8) || (((size_t)logits) % 16)) return CXR_ERR_ARG;
        static int row_on = -1;                                    // CXR_CE_BF16ROW=0: the fp32-resident kernel (A/B)
        if (row_on < 0) { const char* e = getenv("CXR_CE_BF16ROW"); row_on = (e && e[0] == '0') ? 0 : 1; }
        if (row_on && (V & 7) == 0 && V >= 8)
            CXR_LAUNCH(softmax_ce_bf16row_kernel, dim3((unsigned)R), dim3(1024), 0, stream, (const bf16_t*)logits, ld, labels, ignore_index, thr, row_w,
                       row_loss, (bf16_t*)dlogits, lddl, V);
        else
            CXR_LAUNCH((softmax_ce_reg_kernel<true>), dim3((unsigned)R), dim3(1024), 0, stream, logits, ld, labels, ignore_index, thr, row_w, row_loss,
                       (bf16_t*)dlogits, lddl, V);
    } else if (V <= 32768 && lddl <= 32768 && (ld % 4) == 0 && (((size_t)logits) % 16) == 0)
        CXR_LAUNCH((softmax_ce_reg_kernel<false>), dim3((unsigned)R), dim3(1024), 0, stream, logits, ld, labels, ignore_index, thr, row_w, row_loss,
                           (bf16_t*)dlogits, lddl, V);
    else
        CXR_LAUNCH(softmax_ce_kernel, dim3((unsigned)R), dim3(256), 0, stream, (const float*)logits, ld, labels, ignore_index, thr, row_w, row_loss,
                           (bf16_t*)dlogits, lddl, V);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// Row weights + scalar loss.
//   mode 0 (teacher forcing, reference single.py:467-469): w[r] = 1/count(labels != ignore);  loss = sum row_loss / count
//   mode 1 (REINFORCE, reference scst/gt_prompt.py:238-244): w[r] = reward[r / T] / B;          loss = mean_b(reward_b * sum_t row_loss)
__global__ __launch_bounds__(256) void ce_weights_kernel(const long* __restrict__ labels, long R, long ignore_index, int mode,
                                                         const float* __restrict__ reward, int T, float* __restrict__ row_w) {
    __shared__ float sh[4];
    float cnt = 0.f;
    if (mode == 0) {
        for (long r = threadIdx.x; r < R; r += 256) cnt += labels[r] != ignore_index;
        cnt = block_sum(cnt, sh);
    }
    const int B = (int)(R / (T > 0 ? T : 1));
    for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < R; r += (long)gridDim.x * 256) {
        const bool ig = labels[r] == ignore_index;
        row_w[r] = ig ? 0.f : (mode == 0 ? 1.0f / fmaxf(cnt, 1.0f) : reward[r / T] / (float)B);
    }
}
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss, const float* __restrict__ row_w, long R,
                                                        float* __restrict__ loss) {
    __shared__ float sh[4];
    float s = 0.f;
    for (long r = threadIdx.x; r < R; r += 256) s += row_loss[r] * row_w[r];
    s = block_sum(s, sh);
    if (threadIdx.x == 0) *loss = s;
}
extern "C" int cxr_ce_weights(const long* labels, long R, long ignore_index, int mode, const float* reward, int T, float* row_w,
                              hipStream_t stream) {
    if (R <= 0 || (mode == 1 && (!reward || T <= 0))) return CXR_ERR_ARG;
    CXR_LAUNCH(ce_weights_kernel, dim3(cdiv(R, 256) < 64 ? cdiv(R, 256) : 64), dim3(256), 0, stream, labels, R, ignore_index, mode,
                       reward, T, row_w);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
extern "C" int cxr_ce_reduce(const float* row_loss, const float* row_w, long R, float* loss, hipStream_t stream) {
    if (R <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(ce_reduce_kernel, dim3(1), dim3(256), 0, stream, row_loss, row_w, R, loss);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- exact k-th largest (TopKLogitsWarper threshold)
__device__ __forceinline__ unsigned f2ord(float f) { const unsigned u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// radix select, 4 passes of 8 bits over the order-preserving integer image of the floats; returns the k-th largest value of x[0..V)
// (Round 4, measured and removed: one LDS atomic per distinct bin of a wave -- ballot / readlane loop over the 3-6 bins the leading digits of a row of
// logits fall into -- instead of one per element: topk_threshold_kernel on 4080 x 30000 went from 515 to 1035 us. Same-address LDS atomics of a wave are
// not what the first radix pass costs.)
constexpr int KTH_CAP = 2048;            // candidates (ordered keys) kept in LDS after the second radix pass

__device__ float kth_largest(const float* __restrict__ x, int V, int k, unsigned* hist /*[256]*/, unsigned* bcast /*[2]*/) {
    // Exact k-th largest value of a row: MSD radix select on the order-preserving key, 8 bits per pass. The first two passes scan the row (16-byte
    // loads, two per thread in flight: with one 4-byte load per trip and the LDS atomic behind it every trip was one exposed memory round trip --
    // 515 us for the 4080 x 30000 scores of an SCST re-scoring pass); the second one also keeps the keys that share the leading digit in LDS, and
    // the last two passes read that list (a few hundred entries for k = 50 of 30000) instead of the row. 515 -> 369 (loads) -> see profiles/.
    __shared__ unsigned cand[KTH_CAP];
    __shared__ unsigned ncand;
    unsigned prefix = 0u, mask = 0u;
    int remaining = k;
    const bool vec = (V & 3) == 0 && ((size_t)x & 15) == 0;
    bool listed = false;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        __syncthreads();
        hist[threadIdx.x] = 0u;
        if (pass == 1 && threadIdx.x == 0) ncand = 0u;
        __syncthreads();
        if (listed) {
            const int n = (int)ncand;
            for (int i = threadIdx.x; i < n; i += 256) {
                const unsigned o = cand[i];
                if ((o & mask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
            }
        } else if (vec) {
            const float4* x4 = reinterpret_cast<const float4*>(x);
            const int n4 = V >> 2;
            for (int i0 = 0; i0 < n4; i0 += 512) {
                const int ia = i0 + threadIdx.x, ib = ia + 256;
                const float4 fa = x4[ia < n4 ? ia : n4 - 1], fb = x4[ib < n4 ? ib : n4 - 1];
                const float e[8] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned o = f2ord(e[j]);
                    if ((j < 4 ? ia : ib) < n4 && (o & mask) == prefix) {
                        atomicAdd(&hist[(o >> shift) & 255u], 1u);
                        if (pass == 1) { const unsigned c = atomicAdd(&ncand, 1u); if (c < (unsigned)KTH_CAP) cand[c] = o; }
                    }
                }
            }
        } else {
            for (int v = threadIdx.x; v < V; v += 256) {
                const unsigned o = f2ord(x[v]);
                if ((o & mask) == prefix) {
                    atomicAdd(&hist[(o >> shift) & 255u], 1u);
                    if (pass == 1) { const unsigned c = atomicAdd(&ncand, 1u); if (c < (unsigned)KTH_CAP) cand[c] = o; }
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0, d = 255;
            for (; d > 0; --d) { if (acc + (int)hist[d] >= remaining) break; acc += hist[d]; }
            bcast[0] = (unsigned)d; bcast[1] = (unsigned)acc;
        }
        __syncthreads();
        prefix |= bcast[0] << shift;
        mask |= 255u << shift;
        remaining -= (int)bcast[1];
        if (pass == 1) listed = ncand <= (unsigned)KTH_CAP;          // (block-uniform: read after the barrier)
    }
    return ord2f(prefix);
}

// TopPLogitsWarper (TF5 generation/logits_process.py, applied after the top-k warper): with the kept entries sorted ascending, entries whose
// cumulative softmax probability is <= 1 - top_p are removed (the largest entry always stays). The survivors are a suffix of that order, so the
// filter is a value threshold: returns the smallest surviving value among the candidates >= t_k held in LDS (cval / cidx, n entries, n small:
// the top-k set). Ties are ordered by vocabulary index. Called by every thread of the block; *tmin_ord is LDS scratch.
__device__ float topp_threshold_from_list(const float* cval, const int* cidx, int n, float t_k, float top_p, float invt, unsigned* tmin_ord) {
    if (threadIdx.x == 0) *tmin_ord = 0xffffffffu;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float vi = cval[i];
        if (!(vi >= t_k)) continue;
        const int ii = cidx[i];
        float mx = -INFINITY;
        for (int j = 0; j < n; ++j) if (cval[j] >= t_k) mx = fmaxf(mx, cval[j] * invt);
        float tot = 0.f, cum = 0.f;
        bool is_max = true;
        for (int j = 0; j < n; ++j) {
            const float vj = cval[j];
            if (!(vj >= t_k)) continue;
            const float e = __expf(vj * invt - mx);
            tot += e;
            const bool before = vj < vi || (vj == vi && cidx[j] <= ii);        // j sorts at or before i (ascending, index breaks ties)
            if (before) cum += e; else is_max = false;
        }
        const bool removed = !is_max && (cum / tot <= 1.0f - top_p);
        if (!removed) atomicMin(tmin_ord, f2ord(vi));
    }
    __syncthreads();
    const float t = ord2f(*tmin_ord);
    __syncthreads();
    return fmaxf(t, t_k);
}

// TopPLogitsWarper over an arbitrarily large kept set (no top-k, or a top-k set too large for the list form above): the smallest kept VALUE
// among the entries >= t_k of x at temperature 1/invt. HF keeps a token iff the probability mass ranked strictly above it is < top_p
// (TF5 generation/logits_process.py TopPLogitsWarper: ascending sort, drop cumulative <= 1 - top_p), i.e. the minimal top set whose mass reaches
// top_p. Radix select on the order-preserving integer image, 4 passes of 8 bits, each pass histogramming probability MASS per digit (fp32 LDS
// atomics); entries equal to the threshold value are all kept. Scratch: histf [256], sf [2], su [2] in LDS. Works for any blockDim.
__device__ float topp_threshold_mass(const float* x, int V, float t_k, float top_p, float invt, float* histf, float* sf, unsigned* su) {
    const int nt = blockDim.x;
    if (threadIdx.x == 0) { su[0] = 0u; sf[0] = 0.f; }
    __syncthreads();
    unsigned lm = 0u;
    for (int v = threadIdx.x; v < V; v += nt) { const float a = x[v]; if (a >= t_k) { const unsigned o = f2ord(a * invt); lm = o > lm ? o : lm; } }
    atomicMax(&su[0], lm);
    __syncthreads();
    const float mx = ord2f(su[0]);
    float z = 0.f;
    for (int v = threadIdx.x; v < V; v += nt) { const float a = x[v]; if (a >= t_k) z += __expf(a * invt - mx); }
    z = group_sum<64>(z);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sf[0], z);
    __syncthreads();
    const float target = top_p * sf[0];
    unsigned prefix = 0u, mask = 0u;
    float above = 0.f;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        __syncthreads();
        for (int i = threadIdx.x; i < 256; i += nt) histf[i] = 0.f;
        __syncthreads();
        for (int v = threadIdx.x; v < V; v += nt) {
            const float a = x[v];
            if (!(a >= t_k)) continue;
            const unsigned o = f2ord(a);
            if ((o & mask) == prefix) atomicAdd(&histf[(o >> shift) & 255u], __expf(a * invt - mx));
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float acc = above;
            int d = 255;
            for (; d > 0; --d) { if (acc + histf[d] >= target) break; acc += histf[d]; }
            su[1] = (unsigned)d; sf[1] = acc;
        }
        __syncthreads();
        prefix |= su[1] << shift;
        mask |= 255u << shift;
        above = sf[1];
    }
    __syncthreads();
    return fmaxf(ord2f(prefix), t_k);
}

// Round 5: the k-th largest of a row in ONE pass over it, for the shape the SCST re-scoring pass has (k = 50 of 30000, 4080 rows: the radix select
// above reads the row twice and issues one LDS atomic per element into the 3-6 histogram bins a row of logits falls into: 342 us in the step for
// 490 MB = 4x its byte floor). The row is held in REGISTERS (30 x 16-byte loads per thread, all in flight before the first use; order-preserving
// integer keys). Every thread's own maximum is a candidate; the k-th largest of those 256 group maxima, m, is a LOWER bound of the row's k-th largest
// (k of the maxima are >= m), so the answer lies among the entries >= m: ~k (1 + k / 256 / ...) of them for distinct values -- 55 for k = 50 -- which
// are compacted into LDS and ranked by counting. Exact, ties included (an entry is the k-th largest iff fewer than k entries are greater and at
// least k are greater or equal). Returns false (block-uniformly, nothing written) where the form does not apply or the candidate list overflows
// (k > 128, rows wider than 30720 or not 16-byte aligned, long runs of equal values around the threshold): the radix select takes over.
constexpr int K1_NV = 30, K1_CAP = 1024;
__device__ bool kth_largest_onepass(const float* __restrict__ x, int V, int k, float* out, unsigned* gmax /*[256]*/, unsigned* cand /*[K1_CAP]*/,
                                    unsigned* sh /*[3]*/) {
    const int n4 = V >> 2;
    if ((V & 3) != 0 || ((size_t)x & 15) != 0 || n4 < 256 || n4 > 256 * K1_NV || k < 1 || k > 128) return false;
    const int tid = threadIdx.x;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4 f[K1_NV];
#pragma unroll
    for (int u = 0; u < K1_NV; ++u) { const int i = tid + 256 * u; f[u] = x4[i < n4 ? i : n4 - 1]; }
    unsigned key[K1_NV * 4];
    unsigned mx = 0u;
#pragma unroll
    for (int u = 0; u < K1_NV; ++u) {
        const bool ok = tid + 256 * u < n4;                      // (key 0 is below the key of every float, -inf and NaNs included)
        key[4 * u + 0] = ok ? f2ord(f[u].x) : 0u; key[4 * u + 1] = ok ? f2ord(f[u].y) : 0u;
        key[4 * u + 2] = ok ? f2ord(f[u].z) : 0u; key[4 * u + 3] = ok ? f2ord(f[u].w) : 0u;
        mx = max(max(mx, max(key[4 * u], key[4 * u + 1])), max(key[4 * u + 2], key[4 * u + 3]));
    }
    gmax[tid] = mx;
    if (tid == 0) { sh[0] = 0u; sh[1] = 0u; sh[2] = 0u; }
    __syncthreads();
    {   // rank of this thread's maximum among the 256: the one(s) with  #greater < k <= #greater-or-equal  is m
        int gt = 0, ge = 0;
        const uint4* g4 = reinterpret_cast<const uint4*>(gmax);
#pragma unroll 8
        for (int j = 0; j < 64; ++j) {
            const uint4 g = g4[j];
            gt += (g.x > mx) + (g.y > mx) + (g.z > mx) + (g.w > mx);
            ge += (g.x >= mx) + (g.y >= mx) + (g.z >= mx) + (g.w >= mx);
        }
        if (gt < k && k <= ge) sh[1] = mx;
    }
    __syncthreads();
    const unsigned m = sh[1];
#pragma unroll
    for (int e = 0; e < K1_NV * 4; ++e)
        if (key[e] >= m && key[e] != 0u) { const unsigned c = atomicAdd(&sh[0], 1u); if (c < (unsigned)K1_CAP) cand[c] = key[e]; }
    __syncthreads();
    const int n = (int)sh[0];
    if (n > K1_CAP) return false;
    for (int i = tid; i < n; i += 256) {
        const unsigned my = cand[i];
        int gt = 0, ge = 0;
        for (int j = 0; j < n; ++j) { const unsigned c = cand[j]; gt += c > my; ge += c >= my; }
        if (gt < k && k <= ge) sh[2] = my;
    }
    __syncthreads();
    *out = ord2f(sh[2]);
    return true;
}

__global__ __launch_bounds__(256) void topk_threshold_kernel(const float* __restrict__ logits, long ld, int V, int k, float top_p, float invt,
                                                             float* __restrict__ thr) {
    __shared__ __attribute__((aligned(16))) unsigned hist[256];
    __shared__ unsigned bc[4];
    __shared__ float cval[1024];
    __shared__ int cidx[1024];
    __shared__ int ncand;
    __shared__ unsigned tmin;
    __shared__ float histf[256];
    __shared__ float sf[2];
    const float* x = logits + (long)blockIdx.x * ld;
    float t = -INFINITY;
    if (k > 0 && k < V) {
        if (!kth_largest_onepass(x, V, k, &t, hist, reinterpret_cast<unsigned*>(cval), bc)) {
            __syncthreads();
            t = kth_largest(x, V, k, hist, bc);
        }
        __syncthreads();
    }
    if (top_p < 1.0f) {                                          // top-p on top of the top-k set: list form up to 1024 entries, mass radix select beyond
        if (threadIdx.x == 0) ncand = 0;
        __syncthreads();
        for (int v = threadIdx.x; v < V; v += 256) {
            const float a = x[v];
            if (a >= t) { const int i = atomicAdd(&ncand, 1); if (i < 1024) { cval[i] = a; cidx[i] = v; } }
        }
        __syncthreads();
        const int n = ncand;
        if (n <= 1024) t = topp_threshold_from_list(cval, cidx, n, t, top_p, invt, &tmin);
        else t = topp_threshold_mass(x, V, t, top_p, invt, histf, sf, bc);
    }
    if (threadIdx.x == 0) thr[blockIdx.x] = t;
}
// thr[r] = the value below which TopKLogitsWarper(k) followed by TopPLogitsWarper(top_p) (at the given temperature) remove row r's entries
extern "C" int cxr_topk_threshold(const float* logits, long ld, long R, int V, int k, float top_p, float temperature, float* thr, hipStream_t stream) {
    if (R <= 0 || V <= 0 || (k <= 0 && !(top_p < 1.f)) || !(top_p > 0.f) || temperature <= 0.f) return CXR_ERR_ARG;      // k <= 0: no top-k (top-p only)
    CXR_LAUNCH(topk_threshold_kernel, dim3((unsigned)R), dim3(256), 0, stream, logits, ld, V, k, top_p, 1.0f / temperature, thr);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- next-token selection (K14)
// mode 0: argmax (first maximal index).  mode 1: temperature -> top-k -> softmax -> inverse-CDF draw with the caller's uniform u[r]
// (index order = torch.multinomial's category order). Finished rows (unfinished[r]==0) emit pad (TF5 generation/utils.py:2932-2933),
// and a row that emits eos clears its unfinished flag.
__global__ __launch_bounds__(256) void select_token_kernel(const float* __restrict__ logits, long ld, int V, int mode, float temperature, int top_k,
                                                           const float* __restrict__ u, long* __restrict__ next, long next_stride,
                                                           int* __restrict__ unfinished, long eos, long pad, float* __restrict__ margin) {
    __shared__ unsigned hist[256];
    __shared__ unsigned bc[2];
    __shared__ float shf[4];
    __shared__ int shi[4];
    __shared__ float wsum[4];
    const long r = blockIdx.x;
    const float* x = logits + r * ld;
    long tok;
    if (mode == 0) {
        float best = -INFINITY; int bi = 0x7fffffff;
        if ((V & 3) == 0 && ((size_t)x & 15) == 0) {                       // 16-byte loads, 4 independent loads in flight per lane
            const float4* x4 = reinterpret_cast<const float4*>(x);
            const int n4 = V >> 2;
            for (int i0 = threadIdx.x; i0 < n4; i0 += 1024) {
                float4 f[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int i = i0 + u * 256; f[u] = i < n4 ? x4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY); }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int v = (i0 + u * 256) * 4;
                    if (f[u].x > best) { best = f[u].x; bi = v; }
                    if (f[u].y > best) { best = f[u].y; bi = v + 1; }
                    if (f[u].z > best) { best = f[u].z; bi = v + 2; }
                    if (f[u].w > best) { best = f[u].w; bi = v + 3; }
                }
            }
        } else {
            for (int v = threadIdx.x; v < V; v += 256) { const float a = x[v]; if (a > best) { best = a; bi = v; } }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((threadIdx.x & 63) == 0) { shf[threadIdx.x >> 6] = best; shi[threadIdx.x >> 6] = bi; }
        __syncthreads();
        best = shf[0]; bi = shi[0];
        for (int w = 1; w < 4; ++w) if (shf[w] > best || (shf[w] == best && shi[w] < bi)) { best = shf[w]; bi = shi[w]; }
        tok = bi;
        if (margin) {                                   // top-1 / top-2 gap (parity gating of reduced-precision decode)
            float second = -INFINITY;
            for (int v = threadIdx.x; v < V; v += 256) { const float a = x[v]; if (v != bi) second = fmaxf(second, a); }
            second = block_max(second, wsum);
            if (threadIdx.x == 0) margin[r] = best - second;
        }
    } else {
        const float invt = 1.0f / temperature;
        float t = -INFINITY;
        if (top_k > 0 && top_k < V) t = kth_largest(x, V, top_k, hist, bc);
        float mx = -INFINITY;
        for (int v = threadIdx.x; v < V; v += 256) { const float a = x[v]; if (a >= t) mx = fmaxf(mx, a * invt); }
        mx = block_max(mx, shf);
        // contiguous slice per thread so that the prefix order is the vocabulary order
        const int per = (V + 255) / 256, beg = threadIdx.x * per, end = min(V, beg + per);
        float mine = 0.f;
        for (int v = beg; v < end; ++v) { const float a = x[v]; if (a >= t) mine += __expf(a * invt - mx); }
        // block-wide exclusive scan of `mine`
        float incl = mine;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        float woff = 0.f;
        for (int w = 0; w < wv; ++w) woff += wsum[w];
        const float total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const float target = u[r] * total;
        const float excl = woff + incl - mine;
        if (threadIdx.x == 0) shi[0] = -1;
        __syncthreads();
        if (mine > 0.f && target >= excl && target < excl + mine) {
            float c = excl; int pick = -1;
            for (int v = beg; v < end; ++v) { const float a = x[v]; if (a >= t) { c += __expf(a * invt - mx); pick = v; if (target < c) break; } }
            shi[0] = pick;
        }
        __syncthreads();
        const int pick0 = shi[0];
        __syncthreads();
        if (pick0 < 0) {                                // numerical edge (u ~ 1): take the last kept entry
            int last = -1;
            for (int v = threadIdx.x; v < V; v += 256) if (x[v] >= t) last = max(last, v);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
            if (lane == 0) atomicMax(&shi[0], last);
            __syncthreads();
        }
        tok = shi[0];
    }
    if (threadIdx.x == 0) {
        if (unfinished) {
            const int uf = unfinished[r];
            if (!uf) tok = pad;
            else if (tok == eos) unfinished[r] = 0;
        }
        next[r * next_stride] = tok;
    }
}

__global__ void sample_topk_lds_kernel(const float* __restrict__ logits, long ld, int V, float temperature, int top_k, float top_p,
                                       const float* __restrict__ u, long* __restrict__ next, long next_stride, int* __restrict__ unfinished, long eos,
                                       long pad, int n_sample);

extern "C" int cxr_select_token(const float* logits, long ld, long R, int V, int mode, float temperature, int top_k, float top_p, const float* u,
                                long* next, long next_stride, int* unfinished, long eos, long pad, float* margin, int n_sample, hipStream_t stream) {
    if (R <= 0 || V <= 0 || next_stride <= 0 || (mode == 1 && (!u || temperature <= 0.f || !(top_p > 0.f)))) return CXR_ERR_ARG;
    if (n_sample < 0 || n_sample > R) n_sample = (int)R;                      // rows [0, n_sample) sample, rows [n_sample, R) take the argmax (mode 1)
    if (mode == 1 && n_sample < R && !(!margin && (size_t)V * sizeof(float) <= 130 * 1024)) return CXR_ERR_ARG;
    if (mode == 1 && top_p < 1.0f && !(!margin && (size_t)V * sizeof(float) <= 130 * 1024)) return CXR_ERR_ARG;       // top-p lives in the row-in-LDS kernel
    if (mode == 1 && !margin && (size_t)V * sizeof(float) <= 130 * 1024) {           // sampling: row-resident-in-LDS kernel
        static bool attr_set = false;
        if (!attr_set) { (void)hipFuncSetAttribute((const void*)sample_topk_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); attr_set = true; }
        CXR_LAUNCH(sample_topk_lds_kernel, dim3((unsigned)R), dim3(1024), (size_t)V * sizeof(float), stream, logits, ld, V, temperature, top_k, top_p, u,
                   next, next_stride, unfinished, eos, pad, n_sample);
        CXR_LAUNCH_CHECK();
        return CXR_OK;
    }
    CXR_LAUNCH(select_token_kernel, dim3((unsigned)R), dim3(256), 0, stream, logits, ld, V, mode, temperature, top_k, u, next, next_stride, unfinished,
                       eos, pad, margin);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ---------------------------------------------------------------------------------------------- top-k sampling with the row held in LDS
// The decode loop samples one row per sequence (B = 16..64 rows of V = 30000 logits): too few rows to fill the chip, so each row's seven
// passes (4 radix-select passes, max, sum, inverse-CDF search) must not go back to L2/HBM. One 1024-thread workgroup loads its row ONCE
// into LDS (120 KB of the 160 KB) with 16-byte loads and runs every pass from there.
__global__ __launch_bounds__(1024) void sample_topk_lds_kernel(const float* __restrict__ logits, long ld, int V, float temperature, int top_k,
                                                               float top_p, const float* __restrict__ u, long* __restrict__ next, long next_stride,
                                                               int* __restrict__ unfinished, long eos, long pad, int n_sample) {
    extern __shared__ __attribute__((aligned(16))) float row[];          // [V]
    __shared__ unsigned hist[256];
    __shared__ unsigned bc[2];
    __shared__ float wred[16];
    __shared__ int pick;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long r = blockIdx.x;
    const float* x = logits + r * ld;
    const float invt = 1.0f / temperature;
    CXR_STAMP(0);
    // the row as 8 UNCONDITIONAL 16-byte loads per lane, all in flight together (indices past the row re-read its last vector; a load guarded by
    // a per-lane condition makes hipcc wait for each one before issuing the next: 8 dependent round trips). The values stay in registers for
    // the candidate-filter path below (its passes over the row read registers, not LDS); the LDS copy serves the general path.
    const bool vec_row = (V & 3) == 0 && ((size_t)x & 15) == 0 && V <= 8 * 4096;       // wave-uniform
    float4 f[8];
    if (vec_row) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int i = (tid + it * 1024) * 4; i = i < V ? i : V - 4;
            f[it] = *reinterpret_cast<const float4*>(x + i);
        }
        // (pin the loaded values: keeps hipcc from sinking each load into the guarded store below; a "memory" clobber here put f[] in scratch)
#pragma unroll
        for (int it = 0; it < 8; ++it) asm volatile("" : "+v"(f[it].x), "+v"(f[it].y), "+v"(f[it].z), "+v"(f[it].w));
        CXR_STAMP(1);
#pragma unroll
        for (int it = 0; it < 8; ++it)
            if ((tid + it * 1024) * 4 >= V) f[it] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);      // past the row: never a maximum / candidate
        // (the LDS copy of the row is written only if the general path below is taken: ROW_TO_LDS)
    } else {
        for (int i = tid * 4; i < V; i += 4096) {
            if (i + 4 <= V && ((size_t)(x + i) & 15) == 0) *reinterpret_cast<float4*>(row + i) = *reinterpret_cast<const float4*>(x + i);
            else for (int j = i; j < V && j < i + 4; ++j) row[j] = x[j];
        }
    }
#define ROW_TO_LDS()                                                                                          \
    do {                                                                                                      \
        if (vec_row) {                                                                                        \
            _Pragma("unroll") for (int it = 0; it < 8; ++it) {                                                \
                const int i = (tid + it * 1024) * 4;                                                          \
                if (i < V) *reinterpret_cast<float4*>(row + i) = f[it];                                       \
            }                                                                                                 \
        }                                                                                                     \
        __syncthreads();                                                                                      \
    } while (0)
    if (r >= n_sample) {
        // argmax row in the same launch (the greedy half of an SCST decode batch): first maximal index, like torch.argmax
        __shared__ float gbest[16];
        __shared__ int gidx[16];
        float best = -INFINITY; int bi = 0x7fffffff;
        if (vec_row) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {                        // ascending index: first maximum per thread
                const int v = (tid + it * 1024) * 4;
                if (f[it].x > best) { best = f[it].x; bi = v; }
                if (f[it].y > best) { best = f[it].y; bi = v + 1; }
                if (f[it].z > best) { best = f[it].z; bi = v + 2; }
                if (f[it].w > best) { best = f[it].w; bi = v + 3; }
            }
        } else {
            __syncthreads();
            for (int v = tid; v < V; v += 1024) { const float a = row[v]; if (a > best) { best = a; bi = v; } }     // ascending v: first maximum per thread
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (lane == 0) { gbest[wave] = best; gidx[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w) if (gbest[w] > best || (gbest[w] == best && gidx[w] < bi)) { best = gbest[w]; bi = gidx[w]; }
            long tok = bi;
            if (unfinished) {
                const int uf = unfinished[r];
                if (!uf) tok = pad;
                else if (tok == eos) unfinished[r] = 0;
            }
            next[r * next_stride] = tok;
        }
        return;
    }
    // ---- fast path (top_k <= 256): candidate filtering instead of four contended-histogram radix passes over the row.
    //   L = k-th largest of 256 group maxima (a lower bound of the k-th largest element: those are k distinct elements >= L), so every
    //   kept entry is among the few elements >= L; the exact threshold, the softmax and the inverse-CDF walk then run on that short list.
    constexpr int CAND = 1024;
    __shared__ __attribute__((aligned(16))) unsigned gkey[256];
    __shared__ float cval[CAND];
    __shared__ int cidx[CAND];
    __shared__ float sval[CAND];
    __shared__ int sidx[CAND];
    __shared__ int ncand;
    __shared__ float thr_s;
    __shared__ unsigned lkey;
    if (top_k > 0 && top_k <= 256 && top_k < V && vec_row) {
        float lm = -INFINITY;
#pragma unroll
        for (int it = 0; it < 8; ++it) lm = fmaxf(lm, fmaxf(fmaxf(f[it].x, f[it].y), fmaxf(f[it].z, f[it].w)));
        const float own_max = lm;
        lm = fmaxf(lm, __shfl_xor(lm, 1, 64));
        lm = fmaxf(lm, __shfl_xor(lm, 2, 64));
        // group maxima as DISTINCT ordered keys: the order-preserving integer image of the value with its low 8 bits replaced by 255 - group
        // (ties -> lower group first). Dropping 8 mantissa bits only lowers the bound L a little (a few more candidates).
        if ((tid & 3) == 0) gkey[tid >> 2] = (f2ord(lm) & 0xffffff00u) | (255u - (unsigned)(tid >> 2));
        if (tid == 0) { ncand = 0; thr_s = -INFINITY; lkey = 0u; pick = -1; }
        CXR_STAMP(2);
        __syncthreads();
        {
            // rank of every group key (number of larger keys): 4 lanes per group, each compares against a quarter of the 256 keys with 16-byte LDS
            // reads, one compare + one add-with-carry per key (a one-thread-per-group loop of 256 dependent 4-byte reads took 7 us: one LDS latency
            // per iteration; value + index comparisons on all 1024 threads 3.5 us: the CU's VALU rate)
            const int grp = tid >> 2, part = tid & 3;
            const unsigned mine = gkey[grp];
            int rank = 0;
#pragma unroll
            for (int j4 = 0; j4 < 16; ++j4) {
                const uint4 o = *reinterpret_cast<const uint4*>(&gkey[(part * 16 + j4) * 4]);
                rank += o.x > mine; rank += o.y > mine; rank += o.z > mine; rank += o.w > mine;
            }
            rank += __shfl_xor(rank, 1, 64);
            rank += __shfl_xor(rank, 2, 64);
            if (part == 0 && rank == top_k - 1) lkey = mine & 0xffffff00u;
        }
        __syncthreads();
        CXR_STAMP(3);
        float L = ord2f(lkey);                                     // <= the value of top_k distinct elements: every kept entry is >= L
        if (!(L == L)) L = -INFINITY;                              // (fewer than top_k groups hold data: the truncated key of -inf decodes to a NaN)
        {
            // candidates straight from the registers, compacted with ONE LDS atomic per wave (a same-address atomic per candidate serialised:
            // ~100 x 40 cycles)
            uint32_t m = 0u;
            if (own_max >= L) {
#pragma unroll
                for (int it = 0; it < 8; ++it)
                    if ((tid + it * 1024) * 4 < V)                 // (vectors past the row were set to -inf: not candidates even when L is -inf)
                        m |= ((f[it].x >= L ? 1u : 0u) | (f[it].y >= L ? 2u : 0u) | (f[it].z >= L ? 4u : 0u) | (f[it].w >= L ? 8u : 0u)) << (it * 4);
            }
            const int cnt = __popc(m);
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
            const int total = __shfl(incl, 63, 64);
            int base = 0;
            if (lane == 63 && total) base = atomicAdd(&ncand, total);
            base = __shfl(base, 63, 64);
            int off = base + incl - cnt;
            if (m) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int v = (tid + it * 1024) * 4;
                    const float a4[4] = {f[it].x, f[it].y, f[it].z, f[it].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((m >> (it * 4 + q)) & 1u) { if (off < CAND) { cval[off] = a4[q]; cidx[off] = v + q; } ++off; }
                }
            }
        }
        __syncthreads();
        CXR_STAMP(4);
        const int nc = ncand;
        if (nc <= CAND) {
            // one pass per candidate (8 lanes split the comparisons): its rank by value -> the k-th largest VALUE (entries equal to it are all
            // kept), and its position by vocabulary index -> the candidate list in torch.multinomial's category order
            for (int c0 = 0; c0 < nc; c0 += 128) {
                const int c = c0 + (tid >> 3), part = tid & 7;
                const float a = c < nc ? cval[c] : 0.f;
                const int myi = c < nc ? cidx[c] : 0;
                int gt = 0, ge = 0, pos = 0;
                for (int j = part; j < nc; j += 8) { const float o = cval[j]; gt += o > a; ge += o >= a; pos += cidx[j] < myi; }
                gt += __shfl_xor(gt, 1, 64); ge += __shfl_xor(ge, 1, 64); pos += __shfl_xor(pos, 1, 64);
                gt += __shfl_xor(gt, 2, 64); ge += __shfl_xor(ge, 2, 64); pos += __shfl_xor(pos, 2, 64);
                gt += __shfl_xor(gt, 4, 64); ge += __shfl_xor(ge, 4, 64); pos += __shfl_xor(pos, 4, 64);
                if (c < nc && part == 0) {
                    sval[pos] = a; sidx[pos] = myi;
                    if (gt < top_k && top_k <= ge) thr_s = a;      // every candidate holding that value writes the same number
                }
            }
            __syncthreads();
            CXR_STAMP(5);
            float t = thr_s;
            if (top_p < 1.0f) {                                    // TopPLogitsWarper on the top-k set (uniform branch)
                __shared__ unsigned tmin;
                t = topp_threshold_from_list(cval, cidx, nc, t, top_p, invt, &tmin);
            }
            CXR_STAMP(6);
            if (wave == 0) {                                       // softmax over the kept entries (value >= t) + inverse-CDF walk in index order
                float mx = -INFINITY;
                int last = -1;
                for (int i = lane; i < nc; i += 64) if (sval[i] >= t) { mx = fmaxf(mx, sval[i] * invt); last = i; }
                mx = group_max<64>(mx);
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
                float tot = 0.f;
                for (int i = lane; i < nc; i += 64) tot += sval[i] >= t ? __expf(sval[i] * invt - mx) : 0.f;
                tot = group_sum<64>(tot);
                const float target = u[r] * tot;
                float carry = 0.f;
                int found = -1;
                for (int i0 = 0; i0 < nc && found < 0; i0 += 64) {
                    const int i = i0 + lane;
                    const bool kept = i < nc && sval[i] >= t;
                    const float e = kept ? __expf(sval[i] * invt - mx) : 0.f;
                    float incl = e;
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
                    const bool hit = kept && target < carry + incl;
                    const unsigned long long bal = __ballot(hit);
                    if (bal) found = i0 + __ffsll((long long)bal) - 1;
                    carry += __shfl(incl, 63, 64);
                }
                if (found < 0) found = last;                        // numerical edge (u ~ 1): the last kept entry
                if (lane == 0) {
                    long tok = sidx[found];
                    if (unfinished) {
                        const int uf = unfinished[r];
                        if (!uf) tok = pad;
                        else if (tok == eos) unfinished[r] = 0;
                    }
                    next[r * next_stride] = tok;
                }
                CXR_STAMP(7);
            }
            return;
        }
        __syncthreads();                                           // pathological ties (more than CAND elements >= L): general path below
    }
    ROW_TO_LDS();                                                  // general path: every pass below reads the row from LDS
#undef ROW_TO_LDS
    // exact k-th largest (radix select on the order-preserving integer image)
    float t = -INFINITY;
    if (top_k > 0 && top_k < V) {
        unsigned prefix = 0u, mask = 0u;
        int remaining = top_k;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            if (tid < 256) hist[tid] = 0u;
            __syncthreads();
            for (int v = tid; v < V; v += 1024) {
                const unsigned o = f2ord(row[v]);
                if ((o & mask) == prefix) atomicAdd(&hist[(o >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                int acc = 0, d = 255;
                for (; d > 0; --d) { if (acc + (int)hist[d] >= remaining) break; acc += hist[d]; }
                bc[0] = (unsigned)d; bc[1] = (unsigned)acc;
            }
            __syncthreads();
            prefix |= bc[0] << shift;
            mask |= 255u << shift;
            remaining -= (int)bc[1];
            __syncthreads();
        }
        t = ord2f(prefix);
    }
    if (top_p < 1.0f) {                                  // nucleus over a large kept set (no top-k, top-k > 256, or > 1024 ties): mass radix select
        __shared__ float histf[256];
        __shared__ float sf[2];
        __syncthreads();
        t = topp_threshold_mass(row, V, t, top_p, invt, histf, sf, bc);
    }
    // max of the kept entries
    float mx = -INFINITY;
    for (int v = tid; v < V; v += 1024) { const float a = row[v]; if (a >= t) mx = fmaxf(mx, a * invt); }
    mx = group_max<64>(mx);
    if (lane == 0) wred[wave] = mx;
    __syncthreads();
    mx = wred[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, wred[w]);
    __syncthreads();
    // contiguous slice per thread -> prefix order == vocabulary order (torch.multinomial's category order)
    const int per = (V + 1023) / 1024, beg = tid * per, end = min(V, beg + per);
    float mine = 0.f;
    for (int v = beg; v < end; ++v) { const float a = row[v]; if (a >= t) mine += __expf(a * invt - mx); }
    float incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const float y = __shfl_up(incl, o, 64); if (lane >= o) incl += y; }
    if (lane == 63) wred[wave] = incl;
    if (tid == 0) pick = -1;
    __syncthreads();
    float woff = 0.f, total = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { if (w < wave) woff += wred[w]; total += wred[w]; }
    const float target = u[r] * total;
    const float excl = woff + incl - mine;
    if (mine > 0.f && target >= excl && target < excl + mine) {
        float c = excl; int p = -1;
        for (int v = beg; v < end; ++v) { const float a = row[v]; if (a >= t) { c += __expf(a * invt - mx); p = v; if (target < c) break; } }
        pick = p;
    }
    __syncthreads();
    if (pick < 0) {                                     // numerical edge (u ~ 1): the last kept entry
        int last = -1;
        for (int v = tid; v < V; v += 1024) if (row[v] >= t) last = max(last, v);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
        __syncthreads();
        if (lane == 0) atomicMax(&pick, last);
        __syncthreads();
    }
    if (tid == 0) {
        long tok = pick;
        if (unfinished) {
            const int uf = unfinished[r];
            if (!uf) tok = pad;
            else if (tok == eos) unfinished[r] = 0;
        }
        next[r * next_stride] = tok;
    }
}

// log_softmax over the vocabulary, in place on fp32 rows (beam search scoring, TF5 generation/utils.py:3380)
__global__ __launch_bounds__(256) void log_softmax_kernel(float* __restrict__ x, long ld, int V, const float* __restrict__ add_row) {
    __shared__ float sh[4];
    float* row = x + (long)blockIdx.x * ld;
    float mx = -INFINITY;
    for (int v = threadIdx.x; v < V; v += 256) mx = fmaxf(mx, row[v]);
    mx = block_max(mx, sh);
    float s = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) s += __expf(row[v] - mx);
    s = block_sum(s, sh);
    const float off = mx + __logf(s) - (add_row ? add_row[blockIdx.x] : 0.f);
    for (int v = threadIdx.x; v < V; v += 256) row[v] = row[v] - off;
}
extern "C" int cxr_log_softmax_rows(float* x, long ld, long R, int V, const float* add_row, hipStream_t stream) {
    if (R <= 0 || V <= 0) return CXR_ERR_ARG;
    CXR_LAUNCH(log_softmax_kernel, dim3((unsigned)R), dim3(256), 0, stream, x, ld, V, add_row);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
