// Autoregressive-decode helpers: beam reordering of the self-attention KV cache and the
// top-2*beams continuation search of beam search (TF5 generation/utils.py:3388-3435).
#include "common.h"
#include <stdlib.h>

#ifndef CXR_STAMP
#define CXR_STAMP(i)            /* scripts/lab defines it to record s_memrealtime per wave; nothing in the product build */
#endif

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


// ------------------------------------------------------------------------------------------------------------------
// Device-side beam search (reference: generate(num_beams=4) -> TF5 generation/utils.py:3208-3560 `_beam_search`, do_sample=False,
// early_stopping=False, one EOS id, num_return_sequences=1). One launch per decode step does what the host loop of the library does with
// ~40 small tensor ops and two host synchronisations: log-softmax of the study's beams + running scores, the 2*beams best continuations,
// the split into running / finished beams, the merge into the finished set, the improvement test, and the row indices the KV cache is
// reordered by. Rows are BEAM-MAJOR (row = beam * B + study): the beams of a study then share its cross-attention K/V in the decode
// attention kernel (G queries per K/V stream). State buffers are updated in place (every column of a study is owned by one thread);
// the stop flags are double-buffered by step parity because workgroups of one launch read each other's flags of the previous step.
// After the search has stopped (no study can improve, or every candidate hit the length limit) later launches leave the state alone,
// which lets the host poll the flags asynchronously every few steps instead of synchronising per token.
struct BeamArgs {
    const float* logits; long ld;          // fp32 [beams*B, V], raw
    long* running; long* sequences;        // int64 [beams, B, L]
    float* run_scores; float* beam_scores; // fp32 [B, beams]
    unsigned char* finished;               // [B, beams]
    int* unsat; int* allhit;               // [2, B] each, indexed by step parity
    long* beam_idx;                        // int64 [beams*B]: the cache rows of the new running beams
    float* ws;                             // scan results: per (row, chunk) max, sum of exp, 2*beams best raw logits and their token ids
    int B, nb, V, L, cur, max_length, par, nch;
    long eos;
    float div;                             // (cur + 1 - prompt_len) ^ length_penalty
};
constexpr int BEAM_CHUNK = 4096;           // vocabulary entries per scan workgroup (16 per thread, held in registers)

__device__ __forceinline__ bool beam_stopped(const BeamArgs& a) {
    const int* unsat_prev = a.unsat + (a.par ^ 1) * a.B; const int* hit_prev = a.allhit + (a.par ^ 1) * a.B;
    int any_unsat = 0, all_hit = 1;
    for (int i = 0; i < a.B; ++i) { any_unsat |= unsat_prev[i]; all_hit &= hit_prev[i]; }
    return !(any_unsat && !all_hit);
}

// Scan: workgroup (chunk, row) reduces its slice of the row's logits to (max, sum exp(x - max)) and its K2 largest entries (descending, ties ->
// lowest token id). Within a row the order of the raw logits IS the order of the log-probabilities, so the study's 2*beams best continuations
// are among the per-row top 2*beams: the step kernel below only looks at beams * chunks * K2 candidates.
template <int K2>
__global__ __launch_bounds__(256) void beam_scan_kernel(const BeamArgs a) {
    __shared__ float shv[4]; __shared__ int shi[4]; __shared__ float shs[4];
    const int c = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
    const int len = (a.V + a.nch - 1) / a.nch, lo = c * len, hi = min(a.V, lo + len);
    const float* x = a.logits + (long)row * a.ld;
    float v[16];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lo + tid + 256 * j;
        const float t = x[i < hi ? i : hi - 1];
        v[j] = i < hi ? t : -INFINITY;
        mx = fmaxf(mx, v[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) shv[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(shv[0], shv[1]), fmaxf(shv[2], shv[3]));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) sum += __expf(v[j] - mx);              // exp(-inf) = 0 for the padding
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if ((tid & 63) == 0) shs[tid >> 6] = sum;
    __syncthreads();
    float* out = a.ws + ((long)row * a.nch + c) * (2 + 2 * K2);
    if (tid == 0) { out[0] = mx; out[1] = shs[0] + shs[1] + shs[2] + shs[3]; }
    for (int k = 0; k < K2; ++k) {
        float best = -INFINITY; int bj = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) if (v[j] > best) { best = v[j]; bj = j; }        // ascending j = ascending token id: first maximum wins
        int bi = lo + tid + 256 * bj;
        const int mine = bi;
        if (!(best > -INFINITY)) bi = 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { shv[tid >> 6] = best; shi[tid >> 6] = bi; }
        __syncthreads();
        best = shv[0]; bi = shi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (shv[w] > best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
        if (bi == mine) {
#pragma unroll
            for (int j = 0; j < 16; ++j) if (j == bj) v[j] = -INFINITY;                // taken
        }
        if (tid == 0) { out[2 + k] = best; reinterpret_cast<int*>(out)[2 + K2 + k] = bi == 0x7fffffff ? lo : bi; }
    }
}

template <int NB>
__global__ __launch_bounds__(256) void beam_step_kernel(const BeamArgs a) {
    constexpr int K2 = 2 * NB;
    constexpr int MAXC = NB * 32 * K2;                                    // candidates: beams x chunks (<= 32) x K2
    __shared__ float shv[4]; __shared__ int shi[4];
    __shared__ float off[NB];
    __shared__ float cand[MAXC]; __shared__ int cidx[MAXC];               // log-prob + running score, flat index beam * V + token
    __shared__ float c_lp[K2]; __shared__ int c_idx[K2];
    __shared__ int s_parent[NB], s_tok[NB], s_src[NB], s_stop;
    __shared__ int c_beam[K2], c_tok[K2];
    const int b = blockIdx.x, tid = threadIdx.x, B = a.B, V = a.V, nch = a.nch;
    const int* unsat_prev = a.unsat + (a.par ^ 1) * B; const int* hit_prev = a.allhit + (a.par ^ 1) * B;
    int* unsat_new = a.unsat + a.par * B; int* hit_new = a.allhit + a.par * B;
    if (tid == 0) s_stop = beam_stopped(a);
    __syncthreads();
    if (s_stop) {                           // frozen: identity reorder, flags carried over
        if (tid < NB) a.beam_idx[tid * B + b] = tid * B + b;
        if (tid == 0) { unsat_new[b] = unsat_prev[b]; hit_new[b] = hit_prev[b]; }
        return;
    }
    // ---- log-softmax offsets of the study's beams from the chunk partials: lp = x - (max + log(sum exp(x - max)) - running score)
    if (tid < NB) {
        const float* pr = a.ws + (long)(tid * B + b) * nch * (2 + 2 * K2);
        float mx = -INFINITY;
        for (int c = 0; c < nch; ++c) mx = fmaxf(mx, pr[c * (2 + 2 * K2)]);
        float t = 0.f;
        for (int c = 0; c < nch; ++c) t += pr[c * (2 + 2 * K2) + 1] * __expf(pr[c * (2 + 2 * K2)] - mx);
        off[tid] = mx + __logf(t) - a.run_scores[b * NB + tid];
    }
    __syncthreads();
    const int ncand = NB * nch * K2;
    for (int i = tid; i < ncand; i += 256) {
        const int g = i / (nch * K2), r = i % (nch * K2), c = r / K2, k = r % K2;
        const float* pr = a.ws + ((long)(g * B + b) * nch + c) * (2 + 2 * K2);
        cand[i] = pr[2 + k] - off[g];
        cidx[i] = g * V + reinterpret_cast<const int*>(pr)[2 + K2 + k];
    }
    __syncthreads();
    // ---- the 2*beams best continuations, descending, ties -> lowest flat index (beam * V + token)
    float prev = INFINITY; int prev_i = -1;
    for (int k = 0; k < K2; ++k) {
        float best = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < ncand; i += 256) {
            const float x = cand[i]; const int fi = cidx[i];
            const bool ok = (x < prev) || (x == prev && fi > prev_i);
            if (ok && (x > best || (x == best && fi < bi))) { best = x; bi = fi; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { shv[tid >> 6] = best; shi[tid >> 6] = bi; }
        __syncthreads();
        best = shv[0]; bi = shi[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) if (shv[w] > best || (shv[w] == best && shi[w] < bi)) { best = shv[w]; bi = shi[w]; }
        prev = best; prev_i = bi;
        if (tid == 0) { c_lp[k] = best; c_idx[k] = bi == 0x7fffffff ? 0 : bi; }       // (all-NaN rows: keep the indices in range)
    }
    __syncthreads();
    // ---- bookkeeping of the step (a few dozen scalar operations)
    if (tid == 0) {
        const float NEG = -1.0e9f;
        float run_lp[K2], fin_lp[K2]; bool hits[K2]; bool all_hits = true;
        const bool unsat = unsat_prev[b] != 0;
        for (int j = 0; j < K2; ++j) {
            c_beam[j] = c_idx[j] / V; c_tok[j] = c_idx[j] % V;
            hits[j] = (c_tok[j] == (int)a.eos) || (a.cur + 1 >= a.max_length);
            all_hits = all_hits && hits[j];
            run_lp[j] = c_lp[j] + (hits[j] ? NEG : -0.0f);
            const bool just = hits[j] && j < NB;
            fin_lp[j] = c_lp[j] / a.div;
            fin_lp[j] = fin_lp[j] + (unsat ? -0.0f : NEG);
            fin_lp[j] = fin_lp[j] + (just ? -0.0f : NEG);
        }
        // the beams that keep running: the `beams` best continuations that did not end (stable: lowest candidate first among ties)
        unsigned taken = 0; float new_run[NB];
        for (int i = 0; i < NB; ++i) {
            int bj = -1;
            for (int j = 0; j < K2; ++j) if (!((taken >> j) & 1) && (bj < 0 || run_lp[j] > run_lp[bj])) bj = j;
            taken |= 1u << bj;
            s_parent[i] = c_beam[bj]; s_tok[i] = c_tok[bj]; new_run[i] = run_lp[bj];
        }
        // finished set: the `beams` best of (held finished beams, continuations that ended among the first `beams` candidates)
        float m_sc[NB + K2]; bool m_fin[NB + K2];
        for (int i = 0; i < NB; ++i) { m_sc[i] = a.beam_scores[b * NB + i]; m_fin[i] = a.finished[b * NB + i] != 0; }
        for (int j = 0; j < K2; ++j) { m_sc[NB + j] = fin_lp[j]; m_fin[NB + j] = hits[j] && j < NB; }
        unsigned tk = 0; float nsc[NB]; bool nfin[NB];
        for (int i = 0; i < NB; ++i) {
            int bj = -1;
            for (int j = 0; j < NB + K2; ++j) if (!((tk >> j) & 1) && (bj < 0 || m_sc[j] > m_sc[bj])) bj = j;
            tk |= 1u << bj;
            s_src[i] = bj; nsc[i] = m_sc[bj]; nfin[i] = m_fin[bj];
        }
        float mn = nsc[0];
        for (int i = 1; i < NB; ++i) mn = fminf(mn, nsc[i]);
        const float best_run = new_run[0] / a.div;      // the library divides by (cur - prompt)^lp AFTER its cur += 1: the same number
        bool improve = false;
        for (int i = 0; i < NB; ++i) improve = improve || (best_run > (nfin[i] ? mn : NEG));
        for (int i = 0; i < NB; ++i) {
            a.run_scores[b * NB + i] = new_run[i]; a.beam_scores[b * NB + i] = nsc[i]; a.finished[b * NB + i] = nfin[i] ? 1 : 0;
            a.beam_idx[i * B + b] = (long)s_parent[i] * B + b;
        }
        unsat_new[b] = (unsat && improve) ? 1 : 0;
        hit_new[b] = all_hits ? 1 : 0;
    }
    __syncthreads();
    // ---- token rows: column t of every beam of the study is read, then written, by one thread (in place)
    for (int t = tid; t <= a.cur; t += 256) {
        long oldr[NB], olds[NB];
#pragma unroll
        for (int g = 0; g < NB; ++g) {
            oldr[g] = a.running[((long)g * B + b) * a.L + t];
            olds[g] = a.sequences[((long)g * B + b) * a.L + t];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            long r = 0, q = 0;
            const int par = s_parent[i], src = s_src[i];
            const int cb = src >= NB ? c_beam[src - NB] : 0;
#pragma unroll
            for (int g = 0; g < NB; ++g) {
                if (g == par) r = oldr[g];
                if (src < NB ? g == src : g == cb) q = src < NB ? olds[g] : oldr[g];
            }
            if (t == a.cur) { r = s_tok[i]; if (src >= NB) q = c_tok[src - NB]; }
            a.running[((long)i * B + b) * a.L + t] = r;
            a.sequences[((long)i * B + b) * a.L + t] = q;
        }
    }
}
// div = (cur + 1 - prompt_len) ** length_penalty, evaluated in double by the caller as the library evaluates the Python power, rounded to
// fp32 as its tensor division does. ws: beams*B * ceil(V / 4096) * (2 + 4*beams) floats.
extern "C" int cxr_beam_step(const float* logits, long ld, long* running, long* sequences, float* run_scores, float* beam_scores,
                             unsigned char* finished, int* unsat, int* allhit, long* beam_idx, float* ws, int B, int nb, int V, int L, int cur,
                             int max_length, long eos, float div, hipStream_t stream) {
    if (B <= 0 || V <= 0 || cur < 0 || cur >= L || (nb != 1 && nb != 2 && nb != 3 && nb != 4 && nb != 5 && nb != 8) || (long)nb * V >= (1L << 31) || !ws)
        return CXR_ERR_ARG;
    const int nch = cdiv(V, BEAM_CHUNK);
    if (nch > 32) return CXR_ERR_ARG;
    BeamArgs a{logits, ld, running, sequences, run_scores, beam_scores, finished, unsat, allhit, beam_idx, ws, B, nb, V, L, cur, max_length, cur & 1,
               nch, eos, div};
    const dim3 sg(nch, nb * B);
#define BEAM_CASE(NB_) case NB_: CXR_LAUNCH(beam_scan_kernel<2 * NB_>, sg, dim3(256), 0, stream, a);                       \
                                 CXR_LAUNCH(beam_step_kernel<NB_>, dim3(B), dim3(256), 0, stream, a); break
    switch (nb) { BEAM_CASE(1); BEAM_CASE(2); BEAM_CASE(3); BEAM_CASE(4); BEAM_CASE(5); default: BEAM_CASE(8); }
#undef BEAM_CASE
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// The KV-cache reorder of a beam step for ALL layers in one launch: out[i][r, :rows, :] = in[i][idx[r], :rows, :] for up to 16 tensors of
// one geometry (blockIdx.y = tensor).
struct GatherMulti { const bf16_t* in[16]; bf16_t* out[16]; };
__global__ __launch_bounds__(256) void gather_multi_kernel(const GatherMulti g, long bs, long rs, const long* __restrict__ idx, int B, int rows,
                                                           int C) {
    const bf16_t* in = g.in[blockIdx.y]; bf16_t* out = g.out[blockIdx.y];
    const int cch = C / 8;
    const long total = (long)B * rows * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c8 = (int)(i % cch) * 8;
        const long t = i / cch;
        const int r = (int)(t % rows), b = (int)(t / rows);
        *reinterpret_cast<uint4*>(out + (long)b * bs + (long)r * rs + c8) = *reinterpret_cast<const uint4*>(in + idx[b] * bs + (long)r * rs + c8);
    }
}
extern "C" int cxr_gather_batch_multi_bf16(const void* const* in, void* const* out, int n, long bs, long rs, const long* idx, int B, int rows,
                                           int C, hipStream_t stream) {
    if (n <= 0 || n > 16 || B <= 0 || rows <= 0 || (C % 8)) return CXR_ERR_ARG;
    GatherMulti g;
    for (int i = 0; i < 16; ++i) { g.in[i] = (const bf16_t*)in[i < n ? i : 0]; g.out[i] = (bf16_t*)out[i < n ? i : 0]; }
    const long total = (long)B * rows * (C / 8);
    const int gx = (int)(cdiv(total, 256) < 2048 ? cdiv(total, 256) : 2048);
    CXR_LAUNCH(gather_multi_kernel, dim3(gx, n), dim3(256), 0, stream, g, bs, rs, idx, B, rows, C);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Skinny GEMM for autoregressive decode: C[M,N] = epi(A[M,K] . W[N,K]^T), M <= 64 (B*beams rows, one token each).
// Weight-streaming regime (cdna_hip_programming.md section 5, "GEMV / M <= 16 decode weights"): every W element is read exactly once
// chip-wide, straight from HBM/L2 into MFMA B fragments (no LDS round trip); a workgroup owns 16 output columns, its 4 waves split K
// and combine through LDS; all of a wave's loads are issued before its first MFMA (deep memory-level parallelism instead of occupancy).
struct SkinnyProb { const bf16_t* W; const float* bias; void* C; long ldw, ldc; int N;
                    const float* lr_t; const bf16_t* lr_B;        // optional rank-8 term: C[m,n] += sum_r lr_t[m][r] * lr_B[n][r]  (LoRA, train mode)
                    const bf16_t* lr_A; uint32_t lr_site; };      // lr_A [8,K] given (LNA kernels): t = lr_scale * dropout(LN(a)) . lr_A^T is computed HERE
struct SkinnyArgs {
    const bf16_t* A; long lda;
    const bf16_t* residual; long ldr;      // added to problem 0 only
    SkinnyProb p[3];
    int nprob, M, K, act, out_f32;
    // train-mode dropout of the dense output before the residual (BertSelfOutput / BertOutput): row m = sequence, position drop_t
    const uint32_t* drop_seed; uint32_t drop_site, drop_thr16; float drop_inv; int drop_t;       // drop_thr16 == 0: off
    // LayerNorm folded into its consumers (a decode step is launch-bound: ~19 LayerNorm launches per token disappear):
    //   lnA_*: A is the RAW pre-LayerNorm sum; every workgroup holds its whole K = 768 slice of A in registers anyway, so it computes the
    //          row statistics (two-pass, fp32) and normalises the fragments before the MFMAs; workgroup 0 publishes (mean, rstd) per row.
    //   lnR_*: the residual operand is LayerNorm(raw residual) with the published statistics.
    const float* lnA_g; const float* lnA_b; float lnA_eps; float* lnA_stats;
    const float* lnR_stats; const float* lnR_g; const float* lnR_b;
    // in-kernel LoRA down-projection (SkinnyProb::lr_A): dropout of the branch input keyed by (lr_seed, site, row, position lr_t) as in lora.hip
    const uint32_t* lr_seed; uint32_t lr_thr16; float lr_inv, lr_scale; int lr_t;
};

__device__ __forceinline__ float bfv(const bf16x8_t& v, int j) {
    return __uint_as_float(((uint32_t)(uint16_t)__builtin_bit_cast(s16x8_t, v)[j]) << 16);
}

// MT = number of 16-row tiles of A; LNA = LayerNorm on the A operand (K == 768); NW = waves per workgroup splitting K (8 for K >= 2048: the
// 3072-wide FFN output projection otherwise runs three dependent load rounds per wave)
template <int MT, bool LNA, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(const SkinnyArgs g) {
    __shared__ float red[NW][MT][64][4];
    __shared__ float redl[NW == 4 ? NW : 1][NW == 4 ? MT : 1][64][4];  // LoRA down-projection partials (rank 8 = columns 0..7 of a 16-wide tile; K = 768 kernels)
    __shared__ float lnbuf[2][4][MT][16];
    CXR_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // grouped launch: blocks [0, N0/16) -> problem 0, next N1/16 -> problem 1, ... (q / k / v projections share A and one launch)
    int blk = blockIdx.x, pi = 0;
    while (pi + 1 < g.nprob && blk >= (g.p[pi].N + 15) / 16) { blk -= (g.p[pi].N + 15) / 16; ++pi; }
    const SkinnyProb P = g.p[pi];
    const int n0 = blk * 16;
    const int fr = lane & 15, fq = lane >> 4;
    const int kslice = g.K / NW, k0 = wave * kslice;
    const int nw = n0 + fr < P.N ? n0 + fr : P.N - 1;
    const bf16_t* wp = P.W + (long)nw * P.ldw + k0 + fq * 8;
    // epilogue operands are fetched up front by wave 0 so their latency overlaps the weight stream
    const int n = n0 + fr;
    float bv = 0.f, rv[MT][4];
    if (wave == 0) {
        if (P.bias && n < P.N) bv = P.bias[n];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = t * 16 + fq * 4 + r;
                rv[t][r] = (g.residual && pi == 0 && m < g.M && n < P.N) ? bf2f(g.residual[(long)m * g.ldr + n]) : 0.f;
            }
    }
    float rg = 1.f, rb = 0.f, rmean[MT][4], rrstd[MT][4], lrB[8];
    uint32_t dseed = 0u;
    if (wave == 0) {                                 // every epilogue operand is requested now: no dependent global load after the MFMAs
        if (g.drop_thr16) dseed = *g.drop_seed;
        const bool lnr = g.lnR_stats && pi == 0 && n < P.N;
        if (lnr) { rg = g.lnR_g[n]; rb = g.lnR_b[n]; }
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = t * 16 + fq * 4 + r;
                rmean[t][r] = (lnr && m < g.M) ? g.lnR_stats[2 * m] : 0.f;
                rrstd[t][r] = (lnr && m < g.M) ? g.lnR_stats[2 * m + 1] : 1.f;
            }
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) lrB[r8] = ((P.lr_t || P.lr_A) && n < P.N) ? bf2f(P.lr_B[(long)n * 8 + r8]) : 0.f;
    }
    f32x4_t acc[MT], accl[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) { acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accl[t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const bool lora_in = NW == 4 && P.lr_A != nullptr;
    const bf16_t* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        int m = t * 16 + fr; m = m < g.M ? m : g.M - 1;
        ap[t] = g.A + (long)m * g.lda + k0 + fq * 8;
    }
    constexpr int KB = LNA ? 6 : (MT == 1 ? 12 : (MT == 2 ? (NW == 8 ? 12 : 8) : 4));   // k-steps (of 32) whose loads are in flight together
    for (int kb = 0; kb < kslice; kb += 32 * KB) {
        bf16x8_t wf[KB], af[MT][KB];
#pragma unroll
        for (int s2 = 0; s2 < KB; ++s2) {
            const int k = kb + s2 * 32;
            if (k < kslice) {
                wf[s2] = *reinterpret_cast<const bf16x8_t*>(wp + k);
#pragma unroll
                for (int t = 0; t < MT; ++t) af[t][s2] = *reinterpret_cast<const bf16x8_t*>(ap[t] + k);
            }
        }
        CXR_STAMP(1);
        if (LNA) {                                                        // kslice == 192 == 32*KB: the loop body runs once
            float mean[MT], rstd[MT];
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    float acc1 = 0.f;
#pragma unroll
                    for (int s2 = 0; s2 < KB; ++s2)
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const float x = bfv(af[t][s2], j);
                            acc1 += pass == 0 ? x : (x - mean[t]) * (x - mean[t]);
                        }
                    acc1 += __shfl_xor(acc1, 16, 64);
                    acc1 += __shfl_xor(acc1, 32, 64);
                    if (fq == 0) lnbuf[pass][wave][t][fr] = acc1;
                }
                __syncthreads();
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const float tot = (lnbuf[pass][0][t][fr] + lnbuf[pass][1][t][fr]) + (lnbuf[pass][2][t][fr] + lnbuf[pass][3][t][fr]);
                    if (pass == 0) mean[t] = tot * (1.0f / 768.0f);
                    else rstd[t] = rsqrtf(tot * (1.0f / 768.0f) + g.lnA_eps);
                }
            }
            if (g.lnA_stats && blockIdx.x == 0 && wave == 0 && fq == 0) {
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const int m = t * 16 + fr;
                    if (m < g.M) { g.lnA_stats[2 * m] = mean[t]; g.lnA_stats[2 * m + 1] = rstd[t]; }
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < KB; ++s2) {
                const float* gp = g.lnA_g + k0 + s2 * 32 + fq * 8;
                const float* bp = g.lnA_b + k0 + s2 * 32 + fq * 8;
                const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
                const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
                const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    s16x8_t o;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (short)f2bf((bfv(af[t][s2], j) - mean[t]) * rstd[t] * gg[j] + bb[j]);
                    af[t][s2] = __builtin_bit_cast(bf16x8_t, o);
                }
            }
        }
        CXR_STAMP(2);
        if (lora_in) {
            // rank-8 LoRA down-projection of the SAME normalised rows, with the branch's own input dropout: one more 16-wide MFMA tile per
            // k-step whose B operand is lr_A (rows 0..7; lanes 8..15 of a row group feed zeros) -- replaces a separate launch per layer
            const uint32_t lseed = g.lr_thr16 ? *g.lr_seed : 0u;
#pragma unroll
            for (int s2 = 0; s2 < KB; ++s2) {
                if (kb + s2 * 32 >= kslice) continue;
                const int kcol = k0 + kb + s2 * 32 + fq * 8;
                bf16x8_t wl = __builtin_bit_cast(bf16x8_t, s16x8_t{0, 0, 0, 0, 0, 0, 0, 0});
                if (fr < 8) wl = *reinterpret_cast<const bf16x8_t*>(P.lr_A + (long)fr * g.K + kcol);
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    bf16x8_t ad = af[t][s2];
                    if (g.lr_thr16) {
                        int m = t * 16 + fr; m = m < g.M ? m : g.M - 1;
                        const uint32_t key = dropout_row_key(lseed, P.lr_site, (uint32_t)m, (uint32_t)g.lr_t);
                        s16x8_t o;
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            const uint32_t bits = dropout_pair_bits(key, (uint32_t)(kcol + j) >> 1);
                            o[j] = (short)f2bf((bits & 0xffffu) >= g.lr_thr16 ? bfv(ad, j) * g.lr_inv : 0.f);
                            o[j + 1] = (short)f2bf((bits >> 16) >= g.lr_thr16 ? bfv(ad, j + 1) * g.lr_inv : 0.f);
                        }
                        ad = __builtin_bit_cast(bf16x8_t, o);
                    }
                    accl[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ad, wl, accl[t], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int s2 = 0; s2 < KB; ++s2) {
            if (kb + s2 * 32 < kslice) {
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[t][s2], wf[s2], acc[t], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            red[wave][t][lane][r] = acc[t][r];
            if (NW == 4) { if (lora_in) redl[wave][t][lane][r] = accl[t][r]; }
        }
    CXR_STAMP(3);
    __syncthreads();
    CXR_STAMP(4);
    if (wave != 0 || n >= P.N) return;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = t * 16 + fq * 4 + r;                       // D[m][n]: lane owns column n, rows (lane>>4)*4 + r
            if (m >= g.M) continue;
            float v = bv;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w][t][lane][r];
            if (P.lr_t) {
#pragma unroll
                for (int r8 = 0; r8 < 8; ++r8) v += P.lr_t[m * 8 + r8] * lrB[r8];
            }
            if (NW == 4) {
                if (lora_in) {                                   // t[m][r8] sits in column r8 of the LoRA tile: lane fq*16 + r8, element r
                    float la = 0.f;
#pragma unroll
                    for (int r8 = 0; r8 < 8; ++r8) {
                        float tv = 0.f;
#pragma unroll
                        for (int w = 0; w < NW; ++w) tv += redl[w][t][fq * 16 + r8][r];
                        la += tv * lrB[r8];
                    }
                    v += la * g.lr_scale;
                }
            }
            if (g.act == 1) v = gelu_f(v);
            if (g.drop_thr16)
                v = dropout_keep(dropout_row_key(dseed, g.drop_site, (uint32_t)m, (uint32_t)g.drop_t), (uint32_t)n, g.drop_thr16) ? v * g.drop_inv : 0.f;
            float res = rv[t][r];
            if (g.lnR_stats && pi == 0) res = (res - rmean[t][r]) * rrstd[t][r] * rg + rb;
            v += res;
            if (g.out_f32) reinterpret_cast<float*>(P.C)[(long)m * P.ldc + n] = v;
            else reinterpret_cast<bf16_t*>(P.C)[(long)m * P.ldc + n] = f2bf(v);
        }
    CXR_STAMP(5);
}

static int launch_skinny(const SkinnyArgs& g, hipStream_t stream) {
    int grid = 0;
    for (int i = 0; i < g.nprob; ++i) grid += cdiv(g.p[i].N, 16);
    const int mt = cdiv(g.M, 16);
    if (g.lnA_g && g.K != 768) return CXR_ERR_ARG;
    const bool wide = !g.lnA_g && g.K >= 2048 && (g.K % 256) == 0;
#define SKINNY(MT_) do { if (g.lnA_g)  CXR_LAUNCH((gemm_skinny_kernel<MT_, true, 4>), dim3(grid), dim3(256), 0, stream, g);        \
                         else if (wide) CXR_LAUNCH((gemm_skinny_kernel<MT_, false, 8>), dim3(grid), dim3(512), 0, stream, g);      \
                         else          CXR_LAUNCH((gemm_skinny_kernel<MT_, false, 4>), dim3(grid), dim3(256), 0, stream, g); } while (0)
    if (mt == 1) SKINNY(1); else if (mt == 2) SKINNY(2); else if (mt == 3) SKINNY(3); else SKINNY(4);
#undef SKINNY
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

extern "C" int cxr_gemm_skinny_bf16(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const float* bias, const void* residual,
                                    long ldr, int M, int N, int K, int act, int out_f32, const float* lnA_gamma, const float* lnA_beta,
                                    float lnA_eps, float* lnA_stats, const float* lnR_stats, const float* lnR_gamma, const float* lnR_beta,
                                    float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t, hipStream_t stream) {
    if (M <= 0 || M > 64 || N <= 0 || K <= 0 || (K % 128) || (lda % 8) || (ldw % 8)) return CXR_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed)) return CXR_ERR_ARG;
    SkinnyArgs g;
    g.drop_seed = drop_seed; g.drop_site = drop_site; g.drop_thr16 = drop_p > 0.f ? dropout_thr16(drop_p) : 0u;
    g.drop_inv = 1.0f / (1.0f - drop_p); g.drop_t = drop_t;
    if ((lnA_gamma && !lnA_beta) || (lnR_stats && (!lnR_gamma || !lnR_beta || !residual))) return CXR_ERR_ARG;
    g.lnA_g = lnA_gamma; g.lnA_b = lnA_beta; g.lnA_eps = lnA_eps; g.lnA_stats = lnA_stats;
    g.lnR_stats = lnR_stats; g.lnR_g = lnR_gamma; g.lnR_b = lnR_beta;
    g.A = (const bf16_t*)A; g.lda = lda; g.residual = (const bf16_t*)residual; g.ldr = ldr;
    g.p[0].W = (const bf16_t*)W; g.p[0].bias = bias; g.p[0].C = C; g.p[0].ldw = ldw; g.p[0].ldc = ldc; g.p[0].N = N;
    g.p[0].lr_t = nullptr; g.p[0].lr_B = nullptr; g.p[0].lr_A = nullptr; g.p[0].lr_site = 0;
    g.lr_seed = nullptr; g.lr_thr16 = 0u; g.lr_inv = 1.f; g.lr_scale = 0.f; g.lr_t = 0;
    g.p[1] = g.p[0]; g.p[2] = g.p[0];
    g.nprob = 1; g.M = M; g.K = K; g.act = act; g.out_f32 = out_f32;
    return launch_skinny(g, stream);
}

// three projections of the same activations in one launch (decode-step q / k / v: k and v land directly in their KV-cache rows)
extern "C" int cxr_gemm_skinny3_bf16(const void* A, long lda, const void* W0, const float* b0, void* C0, long ldc0, const void* W1,
                                     const float* b1, void* C1, long ldc1, const void* W2, const float* b2, void* C2, long ldc2, long ldw,
                                     int M, int N, int K, const float* lnA_gamma, const float* lnA_beta, float lnA_eps, float* lnA_stats,
                                     const float* lr_t0, const void* lr_B0, const float* lr_t1, const void* lr_B1, const void* lr_A0, const void* lr_A1,
                                     float lr_p, const unsigned int* lr_seed, unsigned int lr_site0, unsigned int lr_site1, int lr_tpos, float lr_scale,
                                     hipStream_t stream) {
    if (M <= 0 || M > 64 || N <= 0 || K <= 0 || (K % 128) || (lda % 8) || (ldw % 8)) return CXR_ERR_ARG;
    if ((lr_A0 || lr_A1) && (K != 768 || lr_p < 0.f || lr_p >= 1.f || (lr_p > 0.f && !lr_seed) || (lr_A0 && !lr_B0) || (lr_A1 && !lr_B1))) return CXR_ERR_ARG;
    SkinnyArgs g;
    g.lr_seed = lr_seed; g.lr_thr16 = lr_p > 0.f ? dropout_thr16(lr_p) : 0u; g.lr_inv = 1.0f / (1.0f - lr_p); g.lr_scale = lr_scale; g.lr_t = lr_tpos;
    g.drop_seed = nullptr; g.drop_site = 0; g.drop_thr16 = 0; g.drop_inv = 1.f; g.drop_t = 0;
    if (lnA_gamma && !lnA_beta) return CXR_ERR_ARG;
    g.lnA_g = lnA_gamma; g.lnA_b = lnA_beta; g.lnA_eps = lnA_eps; g.lnA_stats = lnA_stats;
    g.lnR_stats = nullptr; g.lnR_g = nullptr; g.lnR_b = nullptr;
    g.A = (const bf16_t*)A; g.lda = lda; g.residual = nullptr; g.ldr = 0;
    const void* W[3] = {W0, W1, W2}; const float* b[3] = {b0, b1, b2}; void* C[3] = {C0, C1, C2}; const long ldc[3] = {ldc0, ldc1, ldc2};
    for (int i = 0; i < 3; ++i) { g.p[i].W = (const bf16_t*)W[i]; g.p[i].bias = b[i]; g.p[i].C = C[i]; g.p[i].ldw = ldw; g.p[i].ldc = ldc[i]; g.p[i].N = N;
                                  g.p[i].lr_t = nullptr; g.p[i].lr_B = nullptr; g.p[i].lr_A = nullptr; g.p[i].lr_site = 0; }
    g.p[0].lr_t = lr_t0; g.p[0].lr_B = (const bf16_t*)lr_B0; g.p[1].lr_t = lr_t1; g.p[1].lr_B = (const bf16_t*)lr_B1;     // LoRA on query / key
    g.p[0].lr_A = (const bf16_t*)lr_A0; g.p[0].lr_site = lr_site0; g.p[1].lr_A = (const bf16_t*)lr_A1; g.p[1].lr_site = lr_site1;
    g.nprob = 3; g.M = M; g.K = K; g.act = 0; g.out_f32 = 0;
    return launch_skinny(g, stream);
}

// ------------------------------------------------------------------------------------------------------------------
// Single-query attention for cached decode (self-attention over the KV cache, cross-attention over N*576 encoder tokens):
// one workgroup per (batch row, head); K and V rows are streamed once, 8 lanes per 128-byte row (16 B each), HBM-bound
// (cross-attention K/V = B*6*2*(N*576)*768*2 B per token is the dominant decode traffic).
// Masked keys follow the teacher-forced kernel's convention (finite sentinel -> uniform over masked-only rows).
// K/V rows are read exactly once per token: non-temporal loads keep them from displacing the decoder weights (160 MB, re-read every token)
// in the 256 MB Infinity Cache.
__device__ __forceinline__ uint4 nt_load16(const bf16_t* p) {
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// G queries share one K/V stream: query rows b, b + Bkv, ... (b < Bkv) attend to K/V row b. G = 2 is the SCST step, where the sampled and
// the greedy decode of the same studies run as one batch and read identical cross-attention K/V (340 MB per token at 16 x 2 images).
// NG = 8-lane key groups per workgroup (NG * 8 threads), KU = keys per group per pass: one pass covers NG * KU keys with 2 * KU independent
// 16-byte loads in flight per lane. KU = 9 tiles the encoder's 576 tokens per image exactly (64 groups x 9).
//
// The load phase is BRANCH-FREE: keys past the range re-read the last valid row (an L1 hit) and are discarded by a select, the key-padding
// byte is always read (from a stand-in address when there is no mask). hipcc guards a conditional load with an exec-mask branch and parks
// `s_waitcnt vmcnt(0)` behind it: the round-1 kernel thereby waited for every K/V row before requesting the next one -- eight dependent HBM
// round trips per workgroup, 31 us for 57 MB (scripts/lab/decode_lab.hip).
struct AttnDecArgs {
    const bf16_t* Q; const bf16_t* K; const bf16_t* V; bf16_t* O; const unsigned char* kpm; const uint32_t* drop_seed; float* ws;
    long q_bs, k_bs, k_rs, v_bs, v_rs, o_bs, kpm_bs, kv_hs;
    int H, Tk, Bkv, nsplit, chunk, drop_t;
    float scale, drop_inv; uint32_t drop_site, drop_thr16, has_kpm; int o_mt;          // o_mt > 0: O in the decode activation layout of [16*o_mt, H*64]
};

// MASK: 0 = no key-padding mask, 1 = mask bytes at any alignment (KU byte loads per lane), 2 = mask rows 8-byte aligned and KU == 8 (one 8-byte
// load per lane), 3 = `kpm` holds BIT words (uint32 per 32 keys, row stride kpm_bs BYTES; cxr_pack_mask_bits): two dword loads per lane.
// A lane group owns KU CONSECUTIVE keys of a pass (keys k0 + grp*KU .. +KU), so its mask bits are one short run.
// LOOP = false: the host guarantees one pass per workgroup (chunk <= NG * KU): without the loop-carried row registers the kernel needs ~40
// registers less.
template <int G, int KU, int NG, int MASK, bool LOOP>
__global__ __launch_bounds__(NG * 8) void attn_decode_kernel(const AttnDecArgs a) {
    __shared__ float gm[G][NG / 8], gl[G][NG / 8];      // per-wave partial states
    __shared__ float go[G][NG / 8][64];
    CXR_STAMP(0);
    asm volatile("" :: "s"(a.Q), "s"(a.K), "s"(a.V), "s"(a.kpm), "s"(a.drop_seed), "s"(a.q_bs), "s"(a.k_bs), "s"(a.k_rs), "s"(a.v_bs), "s"(a.v_rs),
                 "s"(a.kpm_bs), "s"(a.kv_hs), "s"(a.H), "s"(a.Tk), "s"(a.Bkv), "s"(a.nsplit), "s"(a.chunk));
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const int tid = threadIdx.x;
    // nsplit > 1 (flash-decoding across workgroups): workgroup (b, h, split) covers keys [split*chunk, +chunk) and leaves its
    // un-normalised state (max, denominator, numerator[64]) in ws for attn_decode_merge_kernel
    const int split = blockIdx.x % a.nsplit, bh = blockIdx.x / a.nsplit;
    const int h = bh % a.H, b = bh / a.H;                            // b indexes K/V (and the key-padding mask)
    const int k_lo = split * a.chunk, k_hi = (k_lo + a.chunk < a.Tk) ? k_lo + a.chunk : a.Tk;
    const int sub = tid & 7, grp = tid >> 3;
    // wave-uniform bases (SGPRs) + 32-bit per-lane element offsets: one address register per load instead of a 64-bit pair (the key range of
    // one (row, head) spans < 2^31 elements: Tk <= 8192 rows)
    const bf16_t* kb = a.K + (long)b * a.k_bs + h * a.kv_hs;              // kv_hs = 64 for token-major [B,T,H*64], T*64 for head-major [B,H,T,64]
    const bf16_t* vb = a.V + (long)b * a.v_bs + h * a.kv_hs;
    const uint32_t k_rs32 = (uint32_t)a.k_rs, v_rs32 = (uint32_t)a.v_rs, sub8 = (uint32_t)sub * 8u;
    const unsigned char* mrow = a.kpm + (long)b * a.kpm_bs;
    float m_run[G], l_run[G], o[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m_run[g] = -1.0e30f; l_run[g] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[g][j] = 0.f;
    }
    uint4 kr[KU], vr[KU];
    uint32_t mw[MASK == 1 ? KU : 2];                     // mask bytes / words of this lane's KU keys
    // the K/V rows of a pass are requested FIRST (oldest in the memory queue), everything small behind them; keys past the range re-read the
    // last valid row (an L1 hit) and are discarded by a select: no branch in the load phase
#define ATTN_DEC_LOAD(k0_)                                                                                   \
    {                                                                                                        \
        const int key0 = (k0_) + grp * KU;                                                                   \
        _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                                     \
            int key = key0 + u; key = key < k_hi ? key : k_hi - 1;                                           \
            kr[u] = nt_load16(kb + ((uint32_t)key * k_rs32 + sub8));                                         \
            vr[u] = nt_load16(vb + ((uint32_t)key * v_rs32 + sub8));                                         \
        }                                                                                                    \
        asm volatile("" ::: "memory");                                                                       \
        if (MASK == 1) {                                                                                     \
            _Pragma("unroll") for (int u = 0; u < KU; ++u) {                                                 \
                int key = key0 + u; key = key < k_hi ? key : k_hi - 1;                                       \
                mw[u] = mrow[key];                                                                           \
            }                                                                                                \
        } else if (MASK == 2) {                                                                              \
            const int kc = key0 < k_hi ? key0 : 0;          /* (a wholly dead group reads the row start) */ \
            const uint2 t2 = *reinterpret_cast<const uint2*>(mrow + kc);                                     \
            mw[0] = t2.x; mw[1] = t2.y;                                                                      \
        } else if (MASK == 3) {                                                                              \
            const int w0 = (key0 < k_hi ? key0 : 0) >> 5, wl = (k_hi - 1) >> 5;                              \
            mw[0] = reinterpret_cast<const uint32_t*>(mrow)[w0];                                             \
            mw[1] = reinterpret_cast<const uint32_t*>(mrow)[w0 < wl ? w0 + 1 : wl];                          \
        }                                                                                                    \
    }
    ATTN_DEC_LOAD(k_lo);
    asm volatile("" ::: "memory");
    const uint32_t dseed = a.drop_seed[vzero];
    uint4 qraw[G];
#pragma unroll
    for (int g = 0; g < G; ++g) qraw[g] = *reinterpret_cast<const uint4*>(a.Q + (long)(b + g * a.Bkv) * a.q_bs + h * 64 + sub * 8);
    CXR_STAMP(1);
    uint32_t drop_key[G];
    const float qscale = a.scale * 1.4426950408889634f;                         // scores directly in the exp2 domain
#pragma unroll
    for (int g = 0; g < G; ++g)     // train-mode dropout on the probabilities: same (b*H+h, query position, key) hash as the tiled kernels (attention.hip)
        drop_key[g] = dropout_row_key(dseed, a.drop_site, (uint32_t)((b + g * a.Bkv) * a.H + h), (uint32_t)a.drop_t);
    for (int k0 = k_lo; k0 < k_hi; k0 += NG * KU) {
        const int key0 = k0 + grp * KU;
        // bit u of `okbits`: key0 + u may be attended to
        uint32_t okbits = (1u << KU) - 1u;
        if (MASK == 1) {
            okbits = 0u;
#pragma unroll
            for (int u = 0; u < KU; ++u) okbits |= (mw[u] != 0u ? 1u : 0u) << u;
        } else if (MASK == 2) {
            okbits = 0u;
#pragma unroll
            for (int u = 0; u < KU; ++u) okbits |= (((u < 4 ? mw[0] >> (8 * u) : mw[1] >> (8 * (u - 4))) & 0xffu) != 0u ? 1u : 0u) << u;
        } else if (MASK == 3) {
            const uint64_t win = ((uint64_t)mw[1] << 32) | mw[0];
            const int w0 = (key0 < k_hi ? key0 : 0) >> 5;
            okbits = (uint32_t)(win >> (key0 - (w0 << 5))) & ((1u << KU) - 1u);          // (key0 >= k_hi: every key of the group is dead anyway)
        }
        float sv[G][KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            float kv[8];
            unpack8(kr[u], kv);
            const bool live = key0 + u < k_hi;
            const bool ok = (okbits >> u) & 1u;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float qv[8], d = 0.f;                                          // the query stays packed (4 registers per row)
                unpack8(qraw[g], qv);
#pragma unroll
                for (int j = 0; j < 8; ++j) d += qv[j] * kv[j];
                d = group_sum<8>(d) * qscale;
                sv[g][u] = live ? (ok ? d : -1.0e30f) : -3.0e38f;            // masked: finite sentinel; beyond the range: never wins the max, p = 0
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float mloc = m_run[g];
#pragma unroll
            for (int u = 0; u < KU; ++u) mloc = fmaxf(mloc, sv[g][u]);
            const float alpha = __builtin_amdgcn_exp2f(m_run[g] - mloc);
            m_run[g] = mloc;
            l_run[g] *= alpha;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] *= alpha;
        }
        // dropout keep bits of this group's KU keys: each of the 8 lanes of a group hashes ONE (or two) of them, the group shares them by
        // shuffle (every lane hashing every key was a quarter of the kernel's VALU work)
        uint32_t keepbits[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            keepbits[g] = (1u << KU) - 1u;
            if (a.drop_thr16) {
                uint32_t mine = 0u;
#pragma unroll
                for (int u = sub; u < KU; u += 8) mine |= (dropout_keep(drop_key[g], (uint32_t)(key0 + u), a.drop_thr16) ? 1u : 0u) << u;
                mine |= __shfl_xor(mine, 1, 64); mine |= __shfl_xor(mine, 2, 64); mine |= __shfl_xor(mine, 4, 64);
                keepbits[g] = mine;
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            float vv[8];
            unpack8(vr[u], vv);
            const bool live = key0 + u < k_hi;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float p = live ? __builtin_amdgcn_exp2f(sv[g][u] - m_run[g]) : 0.f;
                l_run[g] += p;
                const float pd = ((keepbits[g] >> u) & 1u) ? p * a.drop_inv : 0.f;      // drop_inv == 1 without dropout
#pragma unroll
                for (int j = 0; j < 8; ++j) o[g][j] += pd * vv[j];
            }
        }
        if (!LOOP) break;
        if (k0 + NG * KU < k_hi) ATTN_DEC_LOAD(k0 + NG * KU);            // wave-uniform: the next pass's rows (the registers are free again)
    }
#undef ATTN_DEC_LOAD
    CXR_STAMP(2);
    // merge the running states: first the 8 key groups of each wave by butterfly (lanes 8, 16, 32 apart hold the same 8 output dims of other
    // groups), then the NG/8 wave states through LDS (a serial loop over all NG groups by 64*G threads was 2 us of the kernel)
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int off = 8; off < 64; off <<= 1) {
            const float mo = __shfl_xor(m_run[g], off, 64), lo = __shfl_xor(l_run[g], off, 64);
            const float mn = fmaxf(m_run[g], mo);
            const float wa = __builtin_amdgcn_exp2f(m_run[g] - mn), wb = __builtin_amdgcn_exp2f(mo - mn);
            l_run[g] = l_run[g] * wa + lo * wb;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[g][j] = o[g][j] * wa + __shfl_xor(o[g][j], off, 64) * wb;
            m_run[g] = mn;
        }
    }
    constexpr int NWV = NG / 8;                          // waves per workgroup
    const int wv = tid >> 6;
    if ((tid & 63) < 8) {                                // lanes 0..7 of each wave hold the wave's state (dims sub*8 .. +8)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (sub == 0) { gm[g][wv] = m_run[g]; gl[g][wv] = l_run[g]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) go[g][wv][sub * 8 + j] = o[g][j];
        }
    }
    __syncthreads();
    CXR_STAMP(3);
    for (int e = tid; e < 64 * G; e += NG * 8) {
        const int g = e >> 6, d = e & 63;
        float M = -1.0e30f;
#pragma unroll
        for (int q = 0; q < NWV; ++q) M = fmaxf(M, gm[g][q]);
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int q = 0; q < NWV; ++q) {
            const float w = __builtin_amdgcn_exp2f(gm[g][q] - M);
            num += w * go[g][q][d];
            den += w * gl[g][q];
        }
        if (a.nsplit == 1) {
            const int row = b + g * a.Bkv;
            a.O[a.o_mt ? dal_off(row, h * 64 + d, a.o_mt) : (long)row * a.o_bs + h * 64 + d] = f2bf(num / den);
        } else {
            float* w = a.ws + ((((long)(b + g * a.Bkv) * a.H + h) * a.nsplit) + split) * 66;
            w[2 + d] = num;
            if (d == 0) { w[0] = M; w[1] = den; }
        }
    }
    CXR_STAMP(4);
}

// key-padding mask bytes [B, T] -> bit words uint32 [B, words] (bit k%32 of word k/32 = key k may be attended to); one thread per word
__global__ __launch_bounds__(256) void pack_mask_bits_kernel(const unsigned char* __restrict__ kpm, long kpm_bs, int B, int T, uint32_t* __restrict__ out, int words) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * words) return;
    const int b = i / words, w = i % words;
    uint32_t bits = 0u;
    for (int j = 0; j < 32; ++j) {
        const int k = w * 32 + j;
        if (k < T && kpm[(long)b * kpm_bs + k] != 0) bits |= 1u << j;
    }
    out[i] = bits;
}
extern "C" int cxr_pack_mask_bits(const void* kpm, long kpm_bs, int B, int T, unsigned int* out, int words, hipStream_t stream) {
    if (B <= 0 || T <= 0 || words < cdiv(T, 32)) return CXR_ERR_ARG;
    CXR_LAUNCH(pack_mask_bits_kernel, dim3(cdiv((long)B * words, 256)), dim3(256), 0, stream, (const unsigned char*)kpm, kpm_bs, B, T, out, words);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// Cross-attention of a cached decode step on the matrix cores. The cross K/V of a study are static for a whole decode and shared by all of its
// rows (sample + greedy: 2, beams: 4), so V is ALSO kept transposed per head (VT [Bkv, H*64, Tk], written once at prefill): then both products of
// a 32-key block are MFMAs whose operands are single 16-byte loads per lane straight from HBM, with no data movement in between:
//   S^T[keys x queries] = K . Q^T   A = K rows (16 keys x 32 dims per instruction), B = Q^T (dims x 16 query columns, G of them real)
//   O[queries x dims]   = P . V     A = P (the S^T accumulators ARE this fragment when a block's two 16-key tiles interleave the keys
//                                       {8g + r} / {8g + 4 + r}: lane (query, g) then owns keys 8g .. 8g+7 in order), B = VT rows
// Up to 16 query rows cost what one costs. One workgroup = (study, head), 12 waves, wave w owns key blocks w, w+12, w+24 (Tk <= 1152 = a 2-image
// study), all 24 loads of a wave in flight before the first MFMA; softmax over the whole key range with two workgroup reductions (max, then sum and
// numerators); probabilities enter P.V as bf16 (as in the teacher-forced kernels). The VALU kernel above spends 4.8 us per (study, head) on the
// dot products / softmax and 3.8 us merging 16 wave states; this one has ~60 MFMAs and two LDS reductions.
struct AttnXArgs {
    const bf16_t* Q; const bf16_t* Kp; const bf16_t* Vp; bf16_t* O; const uint32_t* mbits; const uint32_t* drop_seed;
    long q_bs, o_bs, mb_bs;                         // mb_bs in 32-bit words
    float* ws;                                      // nsplit > 1: partial states (max, denominator, numerator[64]) per (row, head, split) for attn_decode_merge_kernel
    int H, Tk, Bkv, G, drop_t, o_mt, nsplit, bps;   // bps = key blocks per split (<= 36)
    float scale_log2e, drop_inv; uint32_t drop_site, drop_thr16, has_mask;
    // QPROJ: the query projection q = LN(x) Wq^T + b computed by the workgroup itself (x raw in the decode activation layout, LayerNorm folded into the
    // packed weights, row statistics from the producer's partials: the operands of cxr_dec_gemm_bf16) -- Q is unused then
    const bf16_t* xA; const float* xstats; const bf16_t* qWp; const float2* qbc; int x_mtl, x_M, x_tiles; float x_eps;
};

// Fragment-ordered copies of a study's cross-attention K and V (written once at prefill, like the packed decode weights): block = 32 keys of one
// head. Kp[(b*H + h)*nblk + blk][t][ks][lane][8] = K[key = blk*32 + 8*(m>>2) + 4*t + (m&3)][h*64 + ks*32 + 8*g + j] (m = lane & 15, g = lane >> 4):
// the A fragments of the block's two interleaved 16-key tiles; Vp[(b*H + h)*nblk + blk][dt][lane][8] = V[key = blk*32 + 8*g + j][h*64 + dt*16 + m]:
// the B fragments of P.V. Every fragment is 1 KB contiguous: the decode kernel's loads are 16 bytes per lane, fully coalesced, and can be
// non-temporal (a row-major K gives the MFMA operand layout 64-byte pieces, which only merge through L1 -- i.e. not with streaming loads, and
// plain loads push the decoder weights out of the Infinity Cache: measured +15 us per token-step on the GEMMs that follow).
__global__ __launch_bounds__(256) void pack_cross_kv_kernel(const bf16_t* __restrict__ K, const bf16_t* __restrict__ V, long kv_bs, long kv_rs,
                                                            bf16_t* __restrict__ Kp, bf16_t* __restrict__ Vp, int H, int nblk) {
    __shared__ __attribute__((aligned(16))) bf16_t ks_[32][64 + 8], vs_[32][64 + 8];
    const int blk = blockIdx.x % nblk, bh = blockIdx.x / nblk, h = bh % H, b = bh / H;
    const int tid = threadIdx.x;
    {
        const int r = tid >> 3, c = (tid & 7) * 8;                                                   // 32 rows x 8 chunks of 16 bytes
        const long off = (long)b * kv_bs + (long)(blk * 32 + r) * kv_rs + h * 64 + c;
        *reinterpret_cast<uint4*>(&ks_[r][c]) = *reinterpret_cast<const uint4*>(K + off);
        *reinterpret_cast<uint4*>(&vs_[r][c]) = *reinterpret_cast<const uint4*>(V + off);
    }
    __syncthreads();
    const int lane = tid & 63, f = tid >> 6, m = lane & 15, g = lane >> 4;                          // fragment f of 4, lane of 64
    {
        const int t = f >> 1, ksx = f & 1;
        const int key = 8 * (m >> 2) + 4 * t + (m & 3);
        *reinterpret_cast<uint4*>(Kp + (((long)blockIdx.x * 4 + f) * 64 + lane) * 8) = *reinterpret_cast<const uint4*>(&ks_[key][ksx * 32 + g * 8]);
    }
    {
        const int dt = f;
        s16x8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (short)vs_[8 * g + j][dt * 16 + m];
        *reinterpret_cast<s16x8_t*>(Vp + (((long)blockIdx.x * 4 + f) * 64 + lane) * 8) = o;
    }
}
extern "C" int cxr_pack_cross_kv_bf16(const void* K, const void* V, long kv_bs, long kv_rs, void* Kp, void* Vp, int Bkv, int H, int Tk, hipStream_t stream) {
    if (Bkv <= 0 || H <= 0 || Tk <= 0 || (Tk % 32) || (kv_rs % 8) || (kv_bs % 8)) return CXR_ERR_ARG;
    CXR_LAUNCH(pack_cross_kv_kernel, dim3((unsigned)(Bkv * H * (Tk / 32))), dim3(256), 0, stream, (const bf16_t*)K, (const bf16_t*)V, kv_bs, kv_rs, (bf16_t*)Kp,
               (bf16_t*)Vp, H, Tk / 32);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// PHASED (MAXB = 5: up to 60 key blocks = 1920 keys in ONE workgroup, e.g. the 1728 keys of a 3-image study): the V fragments are requested after the
// scores are computed instead of up front -- K and V fragments of 5 blocks per wave do not fit the 170 registers of a 12-wave workgroup together. The
// V round trip then sits behind the score phase of the same wave, but other workgroups of the launch are in their K phase meanwhile (the launch as a
// whole stays at the HBM rate), and the split + partial states + merge launch of the > 36-block case is gone.
// QPROJ (G <= 4 query rows per study, K = 768): the cross-attention QUERY projection of the cached step runs inside this kernel. Every (study, head)
// workgroup needs only its head's 64 query columns of its G rows: 12 waves split K (2 k-steps of 32 each) over the 4 x 16 output columns, with the
// packed weights as the A operand and the activation fragments as the B operand (both are laid out as either), partial sums meet in LDS, the folded
// LayerNorm is applied as in dec_gemm_kernel (Chan combination of the producer's 48 partial statistics by one wave per row). The weight fragments are
// requested together with the K fragments, so the K stream is not delayed; the separate 4.7-us query launch of every layer is gone. V is requested
// after the projection (as in PHASED): K + V + the projection's operands do not fit 170 registers.
template <int MAXB, bool PHASED = false, bool QPROJ = false>
__global__ __launch_bounds__(768) void attn_cross_mfma_kernel(const AttnXArgs a) {
    constexpr bool VLATE = PHASED || QPROJ;
    __shared__ float qred[QPROJ ? 12 : 1][4][64];
    __shared__ float q_mean[4], q_rstd[4];
    __shared__ __attribute__((aligned(16))) bf16_t q_s[4][64];
    __shared__ float wmax[12][16], wsum[12][16], wmax_all[16];
    __shared__ float wo[12][4][64];                  // per-wave numerators of up to 4 query rows... (G <= 4 here; see the entry point)
    asm volatile("" :: "s"(a.Q), "s"(a.Kp), "s"(a.Vp), "s"(a.mbits), "s"(a.drop_seed), "s"(a.q_bs), "s"(a.mb_bs), "s"(a.H), "s"(a.Tk), "s"(a.Bkv), "s"(a.G));
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int split = blockIdx.x % a.nsplit, bh = blockIdx.x / a.nsplit;
    const int h = bh % a.H, b = bh / a.H;
    const int qi = lane & 15, g4 = lane >> 4;
    const int nblk_all = a.Tk >> 5, blk0 = split * a.bps;
    const int nblk = min(a.bps, nblk_all - blk0);                                        // this workgroup's blocks: blk0 .. blk0 + nblk
    const bf16_t* kp = a.Kp + ((long)bh * nblk_all + blk0) * 2048 + lane * 8;            // 4 fragments of 64 x 8 elements per block
    const bf16_t* vp = a.Vp + ((long)bh * nblk_all + blk0) * 2048 + lane * 8;
    // ---- every load of the wave first: K fragments (2 tiles x 2 k-steps) and V fragments (4 dim tiles) of its blocks, 1 KB contiguous each
    bf16x8_t kf[MAXB][2][2], vf[MAXB][4];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        int blk = wave + 12 * i; blk = blk < nblk ? blk : nblk - 1;                       // (blocks past the range re-read the last one; masked below)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) kf[i][t][ks] = __builtin_bit_cast(bf16x8_t, nt_load16(kp + ((long)blk * 4 + t * 2 + ks) * 512));
        if (!VLATE) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vf[i][dt] = __builtin_bit_cast(bf16x8_t, nt_load16(vp + ((long)blk * 4 + dt) * 512));
        }
    }
    // queries (B operand: column qi = query row b + qi * Bkv; columns >= G are zero), mask words, seed
    const int qrow = b + (qi < a.G ? qi : 0) * a.Bkv;
    bf16x8_t qf[2];
    if (QPROJ) {
        // ---- q[g][64 h + d] for the G (<= 4) rows m_g = b + g * Bkv. D'[d][m] = sum_k W'[64 h + d][k] * x[m][k]: A = packed weight fragment (16 columns d),
        // B = activation fragment of row m_g's 16-row tile; this wave's k-steps are 2 * wave, 2 * wave + 1 of the 24
        // ONE B operand for all G rows: output column g = row m_g (every lane fetches the 16 bytes of "its" row straight from the decode activation
        // layout; columns >= G repeat row m_0 and are dropped) -- a fragment per row would cost G times the MFMAs, accumulators and LDS partials
        bf16x8_t wq[4][2], xf[2];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
                wq[jj][ss] = *reinterpret_cast<const bf16x8_t*>(a.qWp + ((long)((4 * h + jj) * 24 + 2 * wave + ss) * 64 + lane) * 8);
        {
            const int gl = lane & 15;
            const int mrow = b + (gl < a.G ? gl : 0) * a.Bkv;
#pragma unroll
            for (int ss = 0; ss < 2; ++ss)
                xf[ss] = *reinterpret_cast<const bf16x8_t*>(a.xA + ((long)((2 * wave + ss) * a.x_mtl + (mrow >> 4)) * 64 + (lane & 48) + (mrow & 15)) * 8);
        }
        // partial row statistics of the G rows: wave g combines row g's (<= 64) 16-column partials
        float2 pst = make_float2(0.f, 0.f);
        {
            const int myg = wave < a.G ? wave : 0;
            const int tile = lane < a.x_tiles ? lane : 0;
            pst = *reinterpret_cast<const float2*>(a.xstats + ((long)tile * a.x_M + b + myg * a.Bkv) * 2);
        }
        __builtin_amdgcn_sched_barrier(0);
        f32x4_t qa[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            qa[jj] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) qa[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[jj][ss], xf[ss], qa[jj], 0, 0, 0);
        }
        // D' element (row d = 16 jj + 4 (lane >> 4) + r, column lane & 15): columns 0 .. G - 1 are the query rows
        if ((lane & 15) < 4) {
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int r = 0; r < 4; ++r) qred[wave][lane & 15][16 * jj + 4 * (lane >> 4) + r] = qa[jj][r];
        }
        if (wave < 4) {                                             // Chan et al., as in dec_gemm_kernel: n_i = 16 per partial
            const bool on = lane < a.x_tiles;
            const float ntot = 16.0f * (float)a.x_tiles;
            const float mean = group_sum<64>(on ? pst.x : 0.f) / ntot;
            const float dlt = pst.x * (1.0f / 16.0f) - mean;
            const float Qs = group_sum<64>(on ? pst.y + 16.0f * dlt * dlt : 0.f);
            if (lane == 0) { q_mean[wave] = mean; q_rstd[wave] = rsqrtf(Qs / ntot + a.x_eps); }
        }
        __syncthreads();
        if (tid < 256) {
            const int g = tid >> 6, d = tid & 63;
            float raw = 0.f;
#pragma unroll
            for (int w = 0; w < 12; ++w) raw += qred[w][g][d];
            const float2 bcv = a.qbc[h * 64 + d];                  // (bias', column sum of W')
            q_s[g][d] = f2bf(q_rstd[g] * (raw - q_mean[g] * bcv.y) + bcv.x);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 qv = *reinterpret_cast<const uint4*>(&q_s[qi < 4 ? qi : 0][ks * 32 + g4 * 8]);
            if (qi >= a.G) qv = make_uint4(0, 0, 0, 0);
            qf[ks] = __builtin_bit_cast(bf16x8_t, qv);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 qv = *reinterpret_cast<const uint4*>(a.Q + (long)qrow * a.q_bs + h * 64 + ks * 32 + g4 * 8);
            if (qi >= a.G) qv = make_uint4(0, 0, 0, 0);
            qf[ks] = __builtin_bit_cast(bf16x8_t, qv);
        }
    }
    uint32_t mw[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        int blk = wave + 12 * i; blk = blk < nblk ? blk : nblk - 1;
        mw[i] = (a.has_mask ? a.mbits + (long)b * a.mb_bs : a.drop_seed)[a.has_mask ? blk0 + blk : 0];
    }
    const uint32_t dseed = a.drop_seed[vzero];
    __builtin_amdgcn_sched_barrier(0);
    // ---- scores (exp2 domain), masked
    f32x4_t sc[MAXB][2];
    float mx = -3.0e38f;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const bool live = wave + 12 * i < nblk;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][t][0], qf[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i][t][1], qf[1], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kin = 8 * g4 + 4 * t + r;                                          // key inside the block
                const bool ok = !a.has_mask || ((mw[i] >> kin) & 1u);
                const float v = live ? (ok ? acc[r] * a.scale_log2e : -1.0e30f) : -3.0e38f;  // masked: finite sentinel; past the range: p = 0
                acc[r] = v;
                mx = fmaxf(mx, v);
            }
            sc[i][t] = acc;
        }
    }
    if (VLATE) {                                                                          // the K fragments are dead: their registers take the V fragments
#pragma unroll
        for (int i = 0; i < MAXB; ++i) {
            int blk = wave + 12 * i; blk = blk < nblk ? blk : nblk - 1;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) vf[i][dt] = __builtin_bit_cast(bf16x8_t, nt_load16(vp + ((long)blk * 4 + dt) * 512));
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    if (g4 == 0) wmax[wave][qi] = mx;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < 12; ++w) mx = fmaxf(mx, wmax[w][qi]);
    if (wave == 0 && g4 == 0) wmax_all[qi] = mx;                                         // (read after the next barrier by the split epilogue)
    // ---- probabilities, dropout (train mode), P . V
    const uint32_t dkey = dropout_row_key(dseed, a.drop_site, (uint32_t)(qrow * a.H + h), (uint32_t)a.drop_t);
    f32x4_t o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float lsum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int blk = blk0 + wave + 12 * i;
        float p[8];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(sc[i][t][r] - mx);
                lsum += e;
                p[4 * t + r] = e;
            }
        if (a.drop_thr16) {
#pragma unroll
            for (int j = 0; j < 8; j += 2) {                                                 // keys 8 g4 + j, + j + 1: one hash word per pair
                const uint32_t bits = dropout_pair_bits(dkey, (uint32_t)(blk * 32 + 8 * g4 + j) >> 1);
                p[j] = (bits & 0xffffu) >= a.drop_thr16 ? p[j] * a.drop_inv : 0.f;
                p[j + 1] = (bits >> 16) >= a.drop_thr16 ? p[j + 1] * a.drop_inv : 0.f;
            }
        }
        s16x8_t pv;
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = (short)f2bf(p[j]);
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pv);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, vf[i][dt], o[dt], 0, 0, 0);
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (g4 == 0) wsum[wave][qi] = lsum;
    // numerators: C layout of o[dt] = (query row 4 g4 + r, dim dt * 16 + qi): the G <= 4 real queries sit in lane group g4 == 0
    if (g4 == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) wo[wave][r][dt * 16 + qi] = o[dt][r];
    }
    __syncthreads();
    for (int e = tid; e < 64 * a.G; e += 768) {
        const int g = e >> 6, d = e & 63;
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 12; ++w) { num += wo[w][g][d]; den += wsum[w][g]; }
        const int row = b + g * a.Bkv;
        if (a.nsplit == 1) a.O[a.o_mt ? dal_off(row, h * 64 + d, a.o_mt) : (long)row * a.o_bs + h * 64 + d] = f2bf(num / den);
        else {
            float* w = a.ws + ((((long)row * a.H + h) * a.nsplit) + split) * 66;
            w[2 + d] = num;
            if (d == 0) { w[0] = wmax_all[g]; w[1] = den; }
        }
    }
}

// O[b,h,:] from the nsplit (<= 8) partial states of attn_decode_kernel: one 64-lane wave per (query row, head). All partial states are
// requested up front (unused slots re-read slot 0), so the launch costs one memory round trip.
__global__ __launch_bounds__(256) void attn_decode_merge_kernel(const float* __restrict__ ws, bf16_t* __restrict__ O, long o_bs, int H, int nsplit, int rows,
                                                                int o_mt) {
    asm volatile("" :: "s"(ws), "s"(O), "s"(o_bs), "s"(H), "s"(nsplit), "s"(rows));
    int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int d = threadIdx.x & 63;          // r = b*H + h
    const bool live = r < rows;
    r = live ? r : rows - 1;
    const float* w = ws + (long)r * nsplit * 66;
    float m[8], den[8], num[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int sc = s < nsplit ? s : 0;
        m[s] = w[sc * 66]; den[s] = w[sc * 66 + 1]; num[s] = w[sc * 66 + 2 + d];
    }
    asm volatile("" ::: "memory");
    float M = -1.0e30f;
#pragma unroll
    for (int s = 0; s < 8; ++s) M = fmaxf(M, s < nsplit ? m[s] : -1.0e30f);
    float n_ = 0.f, d_ = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float e = s < nsplit ? __builtin_amdgcn_exp2f(m[s] - M) : 0.f;
        n_ += e * num[s];
        d_ += e * den[s];
    }
    if (live) O[o_mt ? dal_off(r / H, (r % H) * 64 + d, o_mt) : (long)(r / H) * o_bs + (r % H) * 64 + d] = f2bf(n_ / d_);
}

// B query rows; K, V (and kpm) have B / kv_share rows: query rows b and b + B/kv_share read K/V row b (kv_share = 1, 2 or 4: the beams of a study).
extern "C" int cxr_attn_decode_bf16(const void* Q, const void* K, const void* V, void* O, const void* kpm, long q_bs, long k_bs, long k_rs,
                                    long v_bs, long v_rs, long o_bs, long kpm_bs, int B, int H, int Tk, float scale, int kv_share, float* ws,
                                    long kv_hs, float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t, int wg_keys,
                                    int o_dal, int kpm_bits, hipStream_t stream) {
    if (B <= 0 || H <= 0 || Tk <= 0 || Tk > 8192 || (k_rs % 8) || (v_rs % 8) || (q_bs % 8) || (k_bs % 8) || (v_bs % 8)) return CXR_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed) || (kv_share != 1 && kv_share != 2 && kv_share != 4) || (B % kv_share)) return CXR_ERR_ARG;
    if (k_rs <= 0 || v_rs <= 0 || (long)Tk * k_rs >= (1L << 31) || (long)Tk * v_rs >= (1L << 31)) return CXR_ERR_ARG;      // 32-bit in-range offsets
    if (kpm_bits && (!kpm || (kpm_bs % 4) || ((uintptr_t)kpm % 4))) return CXR_ERR_ARG;
    const int Bkv = B / kv_share;
    // keys per workgroup pass (= NG key groups x KU consecutive keys): 256 = 32 x 8 (KV-cache self-attention: one pass up to 256 cached tokens),
    // 576 = 64 x 9 and 1152 = 128 x 9 (cross-attention: the encoder emits 576 tokens per image; 1152 = a 2-image study in ONE pass: no split,
    // no merge launch). With `ws`, ranges longer than one pass are split over workgroups (flash-decoding, B*H*8*66 floats) + the merge kernel;
    // negative wg_keys (or no ws): the workgroup loops.
    const bool no_split = wg_keys < 0;
    if (no_split) wg_keys = -wg_keys;
    if (kv_share == 4) wg_keys = (Tk % 576 == 0) ? 576 : 288;     // 4 beams of a study on one K/V stream: the register budget of the two smaller geometries
    if (wg_keys == 0) wg_keys = (Tk % 576 == 0) ? (Tk == 1152 ? 1152 : 576) : 256;       // whole studies of 1 / 2 images in one pass; more: 576-key splits
    if (wg_keys != 256 && wg_keys != 288 && wg_keys != 576 && wg_keys != 1152) return CXR_ERR_ARG;
    int nsplit = 1, chunk = Tk;
    if (ws && Tk > wg_keys && cdiv(Tk, wg_keys) <= 8 && !no_split) { nsplit = cdiv(Tk, wg_keys); chunk = wg_keys; }
    // a workgroup that LOOPS over passes keeps its running softmax state next to the 2 * KU loads in flight: the widest geometry of each sharing
    // degree (1152 keys = 1024 threads = 128 registers; 576 keys with 4 rows per stream) then spills 200-456 bytes per lane -- looping launches
    // step down to the next geometry (576 / 288 keys per pass), which fits
    if (chunk > wg_keys) {
        if (kv_share == 4 && wg_keys == 576) wg_keys = 288;
        else if (kv_share != 4 && wg_keys == 1152) wg_keys = 576;
    }
    const bool loop = chunk > wg_keys;
    AttnDecArgs a;
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O;
    a.kpm = kpm ? (const unsigned char*)kpm : (const unsigned char*)K; a.has_kpm = kpm ? 1u : 0u; a.kpm_bs = kpm ? kpm_bs : 0;
    a.drop_seed = drop_seed ? drop_seed : (const uint32_t*)K; a.ws = ws;
    a.q_bs = q_bs; a.k_bs = k_bs; a.k_rs = k_rs; a.v_bs = v_bs; a.v_rs = v_rs; a.o_bs = o_bs; a.kv_hs = kv_hs;
    a.H = H; a.Tk = Tk; a.Bkv = Bkv; a.nsplit = nsplit; a.chunk = chunk; a.drop_t = drop_t; a.scale = scale;
    a.drop_inv = 1.0f / (1.0f - drop_p); a.drop_site = drop_site; a.drop_thr16 = drop_p > 0.f ? dropout_thr16(drop_p) : 0u;
    a.o_mt = o_dal ? (cdiv(B, 16) == 3 ? 4 : cdiv(B, 16)) : 0;
    if (o_dal && B > 64) return CXR_ERR_ARG;
    // mask form: bit words (two loads per lane), 8-byte-aligned byte rows with 8 keys per lane (one load), bytes (KU loads), none
    const int mask = !kpm ? 0 : (kpm_bits ? 3 : ((wg_keys == 256 && (kpm_bs % 8) == 0 && ((uintptr_t)kpm % 8) == 0 && (nsplit == 1 || (chunk % 8) == 0)) ? 2 : 1));
    const dim3 grid(Bkv * H * nsplit);
#define ATTN_DEC(G_, KU_, NG_, MK_, L_) CXR_LAUNCH((attn_decode_kernel<G_, KU_, NG_, MK_, L_>), grid, dim3(NG_ * 8), 0, stream, a)
#define ATTN_DEC_L(G_, KU_, NG_, MK_) do { if (loop && !((NG_) == 128 || ((G_) == 4 && (NG_) == 64))) ATTN_DEC(G_, KU_, (((NG_) == 128 || ((G_) == 4 && (NG_) == 64)) ? 32 : (NG_)), MK_, true); \
                                           else ATTN_DEC(G_, KU_, NG_, MK_, false); } while (0)      /* (the widest geometries never loop: see above) */
#define ATTN_DEC_M(G_, KU_, NG_) do { if (mask == 0) ATTN_DEC_L(G_, KU_, NG_, 0); else if (mask == 1) ATTN_DEC_L(G_, KU_, NG_, 1);          \
                                      else if (mask == 3) ATTN_DEC_L(G_, KU_, NG_, 3); else ATTN_DEC_L(G_, 8, 32, 2); } while (0)
#define ATTN_DEC_G(KU_, NG_) do { if (kv_share == 2) ATTN_DEC_M(2, KU_, NG_); else ATTN_DEC_M(1, KU_, NG_); } while (0)
#define ATTN_DEC_M4(KU_, NG_) do { if (mask == 0) ATTN_DEC_L(4, KU_, NG_, 0); else if (mask == 1) ATTN_DEC_L(4, KU_, NG_, 1);                \
                                   else ATTN_DEC_L(4, KU_, NG_, 3); } while (0)
    if (kv_share == 4) { if (wg_keys == 288) ATTN_DEC_M4(9, 32); else ATTN_DEC_M4(9, 64); }
    else if (wg_keys == 256) ATTN_DEC_G(8, 32);
    else if (wg_keys == 288) ATTN_DEC_G(9, 32);
    else if (wg_keys == 576) ATTN_DEC_G(9, 64);
    else ATTN_DEC_G(9, 128);
#undef ATTN_DEC_M4
#undef ATTN_DEC_G
#undef ATTN_DEC_M
#undef ATTN_DEC_L
#undef ATTN_DEC
    if (nsplit > 1) CXR_LAUNCH(attn_decode_merge_kernel, dim3(cdiv(B * H, 4)), dim3(256), 0, stream, ws, (bf16_t*)O, o_bs, H, nsplit, B * H, a.o_mt);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// Cross-attention decode step on the matrix cores (attn_cross_mfma_kernel). Q [B rows, H*64] bf16 (row stride q_bs); Kp / Vp = fragment-ordered K / V of
// the Bkv studies (cxr_pack_cross_kv_bf16, Bkv*Tk*H*64 elements each); kpm_bits uint32 [Bkv][mb_words] or NULL; O as in cxr_attn_decode_bf16
// (o_dal: decode activation layout). kv_share = B / Bkv query rows per K/V stream, <= 4. Tk % 32 == 0; Tk > 1152 needs ws (B*H*8*66 floats).
extern "C" int cxr_attn_cross_mfma_bf16(const void* Q, const void* Kp, const void* Vp, void* O, const unsigned int* kpm_bits, long q_bs, long o_bs,
                                        long mb_words, int B, int H, int Tk, float scale, int kv_share, float drop_p, const unsigned int* drop_seed,
                                        unsigned int drop_site, int drop_t, int o_dal, float* ws, hipStream_t stream) {
    if (B <= 0 || H <= 0 || Tk <= 0 || (Tk % 32) || Tk > 8 * 1152 || (Tk > 1920 && !ws) || kv_share < 1 || kv_share > 4 || (B % kv_share) || (q_bs % 8) || ((uintptr_t)Kp % 16) ||
        ((uintptr_t)Vp % 16) || drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed) || (o_dal && B > 64))
        return CXR_ERR_ARG;
    AttnXArgs a;
    a.Q = (const bf16_t*)Q; a.Kp = (const bf16_t*)Kp; a.Vp = (const bf16_t*)Vp; a.O = (bf16_t*)O; a.mbits = kpm_bits;
    a.drop_seed = drop_seed ? drop_seed : (const uint32_t*)Kp;
    a.q_bs = q_bs; a.o_bs = o_bs; a.mb_bs = mb_words;
    a.H = H; a.Tk = Tk; a.Bkv = B / kv_share; a.G = kv_share; a.drop_t = drop_t;
    a.o_mt = o_dal ? (cdiv(B, 16) == 3 ? 4 : cdiv(B, 16)) : 0;
    a.scale_log2e = scale * 1.4426950408889634f; a.drop_inv = 1.0f / (1.0f - drop_p); a.drop_site = drop_site;
    a.drop_thr16 = drop_p > 0.f ? dropout_thr16(drop_p) : 0u; a.has_mask = kpm_bits ? 1u : 0u;
    a.xA = nullptr; a.xstats = nullptr; a.qWp = nullptr; a.qbc = nullptr; a.x_mtl = a.x_M = a.x_tiles = 0; a.x_eps = 0.f;
    // more than 36 key blocks (1152 keys) per study: the blocks are split over nsplit workgroups per (study, head), partial states -> ws
    // (B*H*nsplit*66 floats) -> attn_decode_merge_kernel
    const int nblk_all = Tk / 32;
    a.nsplit = cdiv(nblk_all, 36); a.bps = cdiv(nblk_all, a.nsplit); a.ws = ws;
    if (nblk_all > 36 && nblk_all <= 60) {              // one workgroup per (study, head), K phase then V phase: no partial states, no merge launch
        a.nsplit = 1; a.bps = nblk_all;
        CXR_LAUNCH((attn_cross_mfma_kernel<5, true>), dim3(a.Bkv * H), dim3(768), 0, stream, a);
        CXR_LAUNCH_CHECK();
        return CXR_OK;
    }
    const dim3 grid(a.Bkv * H * a.nsplit);
    if (a.bps <= 12) CXR_LAUNCH(attn_cross_mfma_kernel<1>, grid, dim3(768), 0, stream, a);
    else if (a.bps <= 24) CXR_LAUNCH(attn_cross_mfma_kernel<2>, grid, dim3(768), 0, stream, a);
    else CXR_LAUNCH(attn_cross_mfma_kernel<3>, grid, dim3(768), 0, stream, a);
    if (a.nsplit > 1) CXR_LAUNCH(attn_decode_merge_kernel, dim3(cdiv(B * H, 4)), dim3(256), 0, stream, ws, (bf16_t*)O, o_bs, H, a.nsplit, B * H, a.o_mt);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}

// The same step with the cross-attention QUERY projection inside the kernel (attn_cross_mfma_kernel<.., QPROJ>): xA = the raw hidden rows [x_M rows, 768]
// in the decode activation layout (x_mtl 16-row tiles), xstats = their producer's partial row statistics fp32 [x_tiles][x_M][2], qWp / qbc = the query
// Linear packed by cxr_dec_pack_weight_bf16 with the LayerNorm folded in. Requires H * 64 == 768, kv_share <= 4, Tk <= 1920 (one workgroup per
// (study, head)). Replaces one cxr_dec_gemm_bf16 launch + cxr_attn_cross_mfma_bf16 per layer and token-step.
extern "C" int cxr_attn_cross_mfma_q_bf16(const void* xA, int x_mtl, int x_M, const float* xstats, int x_tiles, float x_eps, const void* qWp, const float* qbc,
                                          const void* Kp, const void* Vp, void* O, const unsigned int* kpm_bits, long o_bs, long mb_words, int B, int H, int Tk,
                                          float scale, int kv_share, float drop_p, const unsigned int* drop_seed, unsigned int drop_site, int drop_t, int o_dal,
                                          hipStream_t stream) {
    if (B <= 0 || H * 64 != 768 || Tk <= 0 || (Tk % 32) || Tk > 1920 || kv_share < 1 || kv_share > 4 || (B % kv_share) || ((uintptr_t)Kp % 16) || ((uintptr_t)Vp % 16) ||
        drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && !drop_seed) || (o_dal && B > 64) || !xA || !xstats || !qWp || !qbc || x_tiles <= 0 || x_tiles > 64 ||
        x_M < B || x_mtl < (B + 15) / 16 || ((uintptr_t)xA % 16) || ((uintptr_t)qWp % 16))
        return CXR_ERR_ARG;
    AttnXArgs a;
    a.Q = (const bf16_t*)Kp; a.Kp = (const bf16_t*)Kp; a.Vp = (const bf16_t*)Vp; a.O = (bf16_t*)O; a.mbits = kpm_bits;
    a.drop_seed = drop_seed ? drop_seed : (const uint32_t*)Kp;
    a.q_bs = 0; a.o_bs = o_bs; a.mb_bs = mb_words;
    a.H = H; a.Tk = Tk; a.Bkv = B / kv_share; a.G = kv_share; a.drop_t = drop_t;
    a.o_mt = o_dal ? (cdiv(B, 16) == 3 ? 4 : cdiv(B, 16)) : 0;
    a.scale_log2e = scale * 1.4426950408889634f; a.drop_inv = 1.0f / (1.0f - drop_p); a.drop_site = drop_site;
    a.drop_thr16 = drop_p > 0.f ? dropout_thr16(drop_p) : 0u; a.has_mask = kpm_bits ? 1u : 0u;
    a.xA = (const bf16_t*)xA; a.xstats = xstats; a.qWp = (const bf16_t*)qWp; a.qbc = (const float2*)qbc; a.x_mtl = x_mtl; a.x_M = x_M; a.x_tiles = x_tiles; a.x_eps = x_eps;
    const int nblk_all = Tk / 32;
    a.nsplit = 1; a.bps = nblk_all; a.ws = nullptr;
    const dim3 grid(a.Bkv * H);
    if (nblk_all <= 12) CXR_LAUNCH((attn_cross_mfma_kernel<1, false, true>), grid, dim3(768), 0, stream, a);
    else if (nblk_all <= 24) CXR_LAUNCH((attn_cross_mfma_kernel<2, false, true>), grid, dim3(768), 0, stream, a);
    else if (nblk_all <= 36) CXR_LAUNCH((attn_cross_mfma_kernel<3, false, true>), grid, dim3(768), 0, stream, a);
    else CXR_LAUNCH((attn_cross_mfma_kernel<5, true, true>), grid, dim3(768), 0, stream, a);
    CXR_LAUNCH_CHECK();
    return CXR_OK;
}
