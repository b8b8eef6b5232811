// Lab (round 6): a stand-in for the LOCAL memory traffic of an RCCL all-reduce on a third stream. One call moves `n16` 16-byte pieces src -> dst with
// `workgroups` workgroups of 256 threads (the pace: RCCL runs a collective on a few dozen workgroups, one per channel), four pieces in flight per lane.
// Built as a shared library by scripts/lab/build.sh and loaded through ctypes by scripts/r6/third_stream_ab.py; not part of the product library.
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void paced_copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long n16) {
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int lab_paced_copy(const void* src, void* dst, long bytes, int workgroups, void* stream) {
    if (!src || !dst || bytes <= 0 || workgroups <= 0) return -1;
    paced_copy_kernel<<<workgroups, 256, 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, bytes / 16);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
