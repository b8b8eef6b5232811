// Lab: can consecutive dependent kernels of a decode step overlap their launch + weight fetch with the predecessor's execution?
// A chain of N "skinny GEMM"-shaped kernels (each workgroup streams its 24-KB weight slice, reads the previous kernel's 48-KB activation, writes its
// 16 output columns) replayed from a hipGraph, two ways:
//   mode 0: one stream, every kernel depends on its predecessor through the queue (what the cached decode step does today)
//   mode 1: kernels alternate between two streams (kernel k has a graph edge to k-2 only) and order themselves through a device counter:
//           producer workgroups release + atomicAdd, consumer workgroups issue their weight loads FIRST, then spin, acquire, read the activation
// hipcc --offload-arch=gfx950 -O3 -o overlap_lab scripts/lab/overlap_lab.hip && ./overlap_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 32, K = 768, NCOL = 768, WG_COLS = 16, NWG = NCOL / WG_COLS;     // 48 workgroups of 256 threads

__global__ __launch_bounds__(256) void step_kernel(const unsigned short* __restrict__ W, const float* __restrict__ xin, float* __restrict__ xout,
                                                   int* done, int k, int flagged) {
    __shared__ float red[256];
    const int tid = threadIdx.x, wg = blockIdx.x;
    // this workgroup's weight slice: 16 columns x 768 = 12288 bf16 = 24 KB -> 6 x 16 B per thread, all issued before anything else
    const uint4* wp = reinterpret_cast<const uint4*>(W + (size_t)wg * WG_COLS * K);
    uint4 w[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = wp[i * 256 + tid];
    if (flagged && k > 0) {
        if (tid == 0) {
            int spins = 0;                                   // bounded: a runtime that serialises kernel k BEFORE k-1 must not hang the box
            while (__hip_atomic_load(&done[k - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NWG && ++spins < 200000) __builtin_amdgcn_s_sleep(1);
            if (spins >= 200000) __hip_atomic_fetch_add(&done[63], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // gave up
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // "GEMM": every thread mixes its weights with a few activation values (enough to keep the loads alive); row sums through LDS
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float x0 = xin[(i * 256 + tid) % (ROWS * K)];
        acc += x0 * (float)(w[i].x & 0xff) + (float)(w[i].y & 0xf) + (float)(w[i].z & 0x3) + (float)(w[i].w & 0x1);
    }
    red[tid] = acc;
    __syncthreads();
    if (tid < WG_COLS * 2) {
        float s = 0.f;
        for (int i = tid; i < 256; i += 32) s += red[i];
        for (int r = 0; r < ROWS / 2; ++r) xout[(r * 2 + (tid & 1)) * NCOL + wg * WG_COLS + (tid >> 1)] = s * 1e-6f + r;
    }
    if (flagged) {
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(&done[k], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 36, reps = 200;
    unsigned short* W; float* act; int* done;
    CK(hipMalloc(&W, (size_t)N * NCOL * K * 2)); CK(hipMemset(W, 1, (size_t)N * NCOL * K * 2));
    CK(hipMalloc(&act, (size_t)(N + 1) * ROWS * NCOL * 4)); CK(hipMemset(act, 0, (size_t)(N + 1) * ROWS * NCOL * 4));
    CK(hipMalloc(&done, 64 * sizeof(int))); CK(hipMemset(done, 0, 64 * sizeof(int)));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    if (N > 60) return 1;
    for (int mode = 0; mode < 3; ++mode) {                          // 2 = mode 1 without a graph (plain launches on the two streams)
        if (mode == 2) {
            const int r2 = 50;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1));
            CK(hipEventRecord(e0, s0));
            for (int i = 0; i < r2; ++i) {
                CK(hipMemsetAsync(done, 0, N * sizeof(int), s0));
                hipEvent_t f, j; CK(hipEventCreateWithFlags(&f, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j, hipEventDisableTiming));
                CK(hipEventRecord(f, s0)); CK(hipStreamWaitEvent(s1, f, 0));
                for (int k = 0; k < N; ++k)
                    hipLaunchKernelGGL(step_kernel, dim3(NWG), dim3(256), 0, (k & 1) ? s1 : s0, W + (size_t)k * NCOL * K, act + (size_t)k * ROWS * NCOL,
                                       act + (size_t)(k + 1) * ROWS * NCOL, done, k, 1);
                CK(hipEventRecord(j, s1)); CK(hipStreamWaitEvent(s0, j, 0));
            }
            CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int gave = 0; CK(hipMemcpy(&gave, done + 63, 4, hipMemcpyDeviceToHost));
            printf("mode 2 (two streams + arrival counters, NO graph): %.2f us per chain, %.2f us per kernel (host-launch bound?); spin give-ups so far %d\n",
                   ms * 1e3 / r2, ms * 1e3 / r2 / N, gave); fflush(stdout);
            continue;
        }
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
        CK(hipMemsetAsync(done, 0, N * sizeof(int), s0));
        hipEvent_t fork, join; CK(hipEventCreate(&fork)); CK(hipEventCreate(&join));
        if (mode == 1) { CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); }
        for (int k = 0; k < N; ++k) {
            hipStream_t st = (mode == 1 && (k & 1)) ? s1 : s0;
            hipLaunchKernelGGL(step_kernel, dim3(NWG), dim3(256), 0, st, W + (size_t)k * NCOL * K, act + (size_t)k * ROWS * NCOL,
                               act + (size_t)(k + 1) * ROWS * NCOL, done, k, mode);
        }
        if (mode == 1) { CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0)); }
        CK(hipStreamEndCapture(s0, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(exec, s0));
        CK(hipStreamSynchronize(s0));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, s0));
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(exec, s0));
        CK(hipEventRecord(e1, s0)); CK(hipStreamSynchronize(s0));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<float> h(ROWS * NCOL);
        CK(hipMemcpy(h.data(), act + (size_t)N * ROWS * NCOL, h.size() * 4, hipMemcpyDeviceToHost));
        printf("mode %d (%s): %d kernels per replay, %.2f us per replay, %.2f us per kernel; out[5] = %.6f\n", mode,
               mode ? "two streams + arrival counters" : "one stream", N, ms * 1e3 / reps, ms * 1e3 / reps / N, h[5]);
        { int gave = 0; CK(hipMemcpy(&gave, done + 63, 4, hipMemcpyDeviceToHost)); printf("   spin give-ups so far: %d\n", gave); }
        fflush(stdout);
    }
    return 0;
}
