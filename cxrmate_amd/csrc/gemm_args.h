// Shared between gemm.hip (128x128 tiles, 2 workgroups per CU) and gemm_pk.hip (persistent 256-row tiles, one workgroup per CU):
// the argument block of an NT GEMM launch and the LDS swizzle of a K-contiguous operand tile.
#pragma once
#include "common.h"

struct GemmArgs {
    const bf16_t* A; long lda;
    const bf16_t* W; long ldw;
    void* C; long ldc;
    const float* bias;            // [N] or null
    const bf16_t* residual; long ldr;   // [M,N] or null, added after the activation
    bf16_t* aux; long ldaux;      // act==1 && aux: pre-activation is stored here; act==2: pre-activation is read from here
    int M, N, K;
    float alpha;
    int act;                      // 0 none, 1 GELU(erf), 2 multiply by GELU'(aux)  (backward of 1)
    int out_f32;                  // 0: C is bf16, 1: C is f32
    int accumulate;               // out_f32 only: C += result
    // train-mode regularisers folded into the epilogue (after the activation): element dropout by the counter-based hash of common.h
    // (row m = sequence m / drop_rows_per_b at position drop_t0 + m % drop_rows_per_b), or a per-image DropPath factor row_scale[m / rs_rows];
    // rs_after != 0 applies the row scale AFTER the residual (CvT's second DropPath scales the whole layer output, quirk Q12)
    const uint32_t* drop_seed; uint32_t drop_site, drop_thr16; float drop_inv; int drop_rows_per_b, drop_t0;
    const float* row_scale; int rs_rows, rs_after;
    int lds_epilogue;             // 1: every memory-facing epilogue access is 16-byte aligned -> the tile goes through LDS and is written in full rows
};

template <int BK> struct Swz;
template <> struct Swz<32> { static __device__ __forceinline__ int f(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; } };
template <> struct Swz<64> { static __device__ __forceinline__ int f(int row) { return (row >> 1) & 7; } };

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// defined in gemm_pk.hip: true when the persistent kernel took the problem (it needs K % 64 == 0, N % 8 == 0, 16-byte aligned rows)
bool gemm_pk_launch(const GemmArgs& g, hipStream_t stream);
// defined in gemm_ws.hip: true when the W-stationary kernel took the problem (K == 384, N % 384 == 0, bf16 output, no dropout)
bool gemm_ws_launch(const GemmArgs& g, hipStream_t stream);
// defined in gemm_strip.hip: true when the row-strip kernel took the problem (N == 384, K % 64 == 0, bf16 output, no GELU / dropout epilogue)
bool gemm_strip_launch(const GemmArgs& g, hipStream_t stream);
// ... and the grouped form: true when ONE row-strip launch took all n (<= 3) problems
bool gemm_strip_group_launch(const GemmArgs* g, int n, hipStream_t stream);
