// Shared device helpers for the CXRMate gfx950 kernels (wave64, MFMA, bf16 storage / fp32 accumulate).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;                                              // raw bf16 bits in HBM
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;          // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;            // 16x16 accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;          // 32x32 accumulator
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;

#define CXR_OK 0
#define CXR_ERR_ARG (-1)
#define CXR_ERR_LAUNCH (-2)

// The kernels of the training step's MAIN stream (NT GEMMs, attention, depthwise projections, LayerNorm) raise their waves' issue priority once, at
// entry: on a SIMD they share with a wave of the weight-gradient stream (priority 0) their MFMA / VALU instructions are arbitrated first
// (MI355X_MICROARCH.md, "Two waves per SIMD", item 2). Same-box alternation of three builds (-DCXR_MAIN_PRIO=0 / 1 / 3, scripts/r4/build_prio.sh,
// profiles/r04_ab_wgrad_stream.txt): 41.17 / 41.08 / 41.07 ms per step -- the two streams mostly contend for CU slots, LDS and bandwidth, not for issue
// slots, so it is worth 0.1 ms, no more. 0 compiles every s_setprio out.
#ifndef CXR_MAIN_PRIO
#define CXR_MAIN_PRIO 3
#endif
#define CXR_PRIO_MAIN() do { if (CXR_MAIN_PRIO) __builtin_amdgcn_s_setprio(CXR_MAIN_PRIO); } while (0)

// torch (and anything else in the process) may leave a stale error in the HIP runtime's per-thread slot: clear it before launching so that
// CXR_LAUNCH_CHECK reports only OUR launch failures.
#define CXR_LAUNCH(...)                                      \
    do {                                                     \
        (void)hipGetLastError();                             \
        hipLaunchKernelGGL(__VA_ARGS__);                     \
    } while (0)

extern "C" int g_cxr_last_hip_error;    // defined in misc.hip; exposed through cxr_last_hip_error()
#define CXR_LAUNCH_CHECK()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) { g_cxr_last_hip_error = (int)e__; return CXR_ERR_LAUNCH; } \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs NaN
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// 16-byte load of data that is read once (streams): non-temporal, so it does not displace reusable lines (weights, the producer->consumer
// activations of the next kernel) in L2 / Infinity Cache. CXR_NT_STREAMS=0 at build time turns the hint off (A/B measurements).
#ifndef CXR_NT_STREAMS
#define CXR_NT_STREAMS 1
#endif
__device__ __forceinline__ uint4 ld_stream16(const void* p) {
#if CXR_NT_STREAMS
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v[0], v[1], v[2], v[3]);
#else
    return *reinterpret_cast<const uint4*>(p);
#endif
}

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
    f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
    f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 v;
    v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
    return v;
}

// ---- e4m3 (OCP fp8) outputs: value * inverse scale, saturating at +-448; used by the fp8 GEMM and by every producer that hands its output
// straight to an e4m3 GEMM (LayerNorm, depthwise projections, attention: no separate quantisation pass)
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
    c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
__device__ __forceinline__ uint8_t to_fp8(float a) {
    a = fminf(fmaxf(a, -448.f), 448.f);
    return (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(a, 0.f, 0, false) & 0xff);
}
__device__ __forceinline__ uint2 pack8_fp8(const float (&f)[8], float inv) {
    return make_uint2(pack_fp8x4(f[0] * inv, f[1] * inv, f[2] * inv, f[3] * inv), pack_fp8x4(f[4] * inv, f[5] * inv, f[6] * inv, f[7] * inv));
}

// GELU(x) = x * Phi(x) with Phi(x) ~= sigmoid(x * (a1 + a3 x^2 + a5 x^4)), minimax-fitted to the exact (erf) GELU that torch.nn.GELU() /
// ACT2FN["gelu"] compute: max |error| 2.5e-5 on the value and 1.1e-4 on the derivative (the derivative below is the exact derivative
// of this function, so forward and backward stay consistent). 9 instructions (one v_exp, one v_rcp) instead of ~40 for libm erff: the
// FFN GEMMs of this model have K = 384/768, i.e. only 6-12 MFMA steps to hide the epilogue of 64 outputs per lane behind.
// The odd polynomial is evaluated on clamp(x, +-8) (it turns around beyond |x| ~ 8.3, where Phi is 0/1 to fp32 precision anyway).
#define CXR_GELU_A1 1.595015769f
#define CXR_GELU_A3 0.07401129203f
#define CXR_GELU_A5 (-0.0007030335804f)
__device__ __forceinline__ float gelu_sigmoid(float x, float& x2) {
    const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    x2 = xc * xc;
    const float t = fmaf(x2, fmaf(x2, CXR_GELU_A5 * -1.4426950408889634f, CXR_GELU_A3 * -1.4426950408889634f), CXR_GELU_A1 * -1.4426950408889634f);
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(xc * t));
}
__device__ __forceinline__ float gelu_f(float x) { float x2; return x * gelu_sigmoid(x, x2); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    float x2;
    const float s = gelu_sigmoid(x, x2);
    const float dp = fmaf(x2, fmaf(x2, 5.0f * CXR_GELU_A5, 3.0f * CXR_GELU_A3), CXR_GELU_A1);
    return s * fmaf(x * (1.0f - s), dp, 1.0f);
}

// ---- counter-based dropout: every kernel that applies (or re-applies, in backward / in the teacher-forced re-scoring of a sampled sequence)
// a dropout mask derives the keep decision of element (site, b, t, col) from the same hash, so masks are never stored. `seed` lives in device
// memory (graph replays see fresh seeds); site = which dropout of the network; (b, t) = sequence index / absolute position (or b*H+h, query)
// for attention probabilities; one 32-bit hash serves two adjacent columns with 16 bits each: keep iff field >= round(p * 65536).
__device__ __forceinline__ uint32_t cxr_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t dropout_row_key(uint32_t seed, uint32_t site, uint32_t b, uint32_t t) {
    return cxr_mix32(cxr_mix32(seed ^ (site * 0x9E3779B9u)) + b * 0x7FEB352Du + t * 0x846CA68Bu);
}
__device__ __forceinline__ uint32_t dropout_pair_bits(uint32_t row_key, uint32_t col_pair) {
    return cxr_mix32(row_key ^ (col_pair * 0x27D4EB2Fu + 0x165667B1u));
}
__device__ __forceinline__ bool dropout_keep(uint32_t row_key, uint32_t col, uint32_t thr16) {
    const uint32_t bits = dropout_pair_bits(row_key, col >> 1);
    return ((col & 1u) ? (bits >> 16) : (bits & 0xffffu)) >= thr16;
}
static inline uint32_t dropout_thr16(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }

// Row-contiguous store of a 32-row x 64-column bf16 tile held in the swapped-MFMA accumulator layout of the attention kernels (lane = row
// (lane & 31), half hh = lane >> 5; for dt in {0,1}, rg in 0..3 the lane owns columns dt*32 + 8*rg + 4*hh .. +3). Storing straight from that
// layout writes 8 bytes per lane to 64 different rows per instruction; through a 4-KB LDS tile (8-byte slots XOR-swizzled by the row) every
// global store is 16 bytes per lane and a full 128-byte row per 8 lanes. Usage: TILE_PUT for the 8 (dt, rg) groups, then tile_rows_store.
#define TILE_PUT(tile, lane, dt, rg, pk) \
    (*reinterpret_cast<uint2*>((tile) + ((lane) & 31) * 64 + ((((dt) * 8 + 2 * (rg) + ((lane) >> 5)) ^ ((lane) & 15)) << 2)) = (pk))
__device__ __forceinline__ void tile_rows_store(const bf16_t* tile, int lane, bf16_t* gbase, long row_stride, int rows_valid) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 64 + lane;
        const int r = c >> 3, cc = c & 7;
        const uint2 lo = *reinterpret_cast<const uint2*>(tile + r * 64 + (((2 * cc) ^ (r & 15)) << 2));
        const uint2 hi = *reinterpret_cast<const uint2*>(tile + r * 64 + (((2 * cc + 1) ^ (r & 15)) << 2));
        if (r < rows_valid) *reinterpret_cast<uint4*>(gbase + (long)r * row_stride + cc * 8) = make_uint4(lo.x, lo.y, hi.x, hi.y);
    }
}

// the same tile written as e4m3 rows (value * inv, saturating): 8 bytes per lane, one 64-byte head row per 8 lanes
__device__ __forceinline__ void tile_rows_store_q8(const bf16_t* tile, int lane, unsigned char* gbase, long row_stride, int rows_valid, float inv) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int c = it * 64 + lane;
        const int r = c >> 3, cc = c & 7;
        const uint2 lo = *reinterpret_cast<const uint2*>(tile + r * 64 + (((2 * cc) ^ (r & 15)) << 2));
        const uint2 hi = *reinterpret_cast<const uint2*>(tile + r * 64 + (((2 * cc + 1) ^ (r & 15)) << 2));
        float f[8];
        unpack8(make_uint4(lo.x, lo.y, hi.x, hi.y), f);
        if (r < rows_valid) *reinterpret_cast<uint2*>(gbase + (long)r * row_stride + cc * 8) = pack8_fp8(f, inv);
    }
}

// XCD-aware workgroup order for 3-D grids whose x index walks the blocks that SHARE operands (the query blocks of one (image, head) all stream
// that head's K / V; its key blocks all stream its Q / dO): hardware deals workgroups round-robin over the 8 XCDs in linear order, so neighbours in
// x land on 8 different L2s and every L2 ends up holding the operands of every image in flight (PMC: 2-3x the unique bytes fetched from the fabric).
// The linear id is remapped (bijectively, remainder included) so that each XCD runs a CONTIGUOUS run of the linear order.
__device__ __forceinline__ void xcd_block_remap(int& bx, int& by, int& bz) {
    const int nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
    const int L = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    const int xcd = L & 7, q = n >> 3, r = n & 7;
    const int Lp = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    bx = Lp % nx;
    const int t = Lp / nx;
    by = t % ny;
    bz = t / ny;
}

template <int W>
__device__ __forceinline__ float group_sum(float v) {     // butterfly over W lanes (W power of two <= 64)
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int W>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// "decode activation layout" (csrc/decode_gemm.hip): element offset of (m, k) in an [16*MT, K] matrix stored in MFMA A-fragment order
__device__ __forceinline__ long dal_off(int m, int k, int MT) {
    return ((long)((k >> 5) * MT + (m >> 4)) << 9) + (((((k & 31) >> 3) << 4) + (m & 15)) << 3) + (k & 7);
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
