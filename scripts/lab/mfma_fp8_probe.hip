// Probe of the A/B operand lane maps of v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands, unit e8m0 scales) with exact integer data.
// build: hipcc --offload-arch=gfx950 -O2 -o mfma_fp8_probe mfma_fp8_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__global__ void probe(const uint8_t* A, const uint8_t* B, float* D, int hyp) {
    // A [32][64] row-major fp8, B^T [32 cols][64 k] fp8. D [32][32].
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { i32x8 v; uint8_t b[32]; } a, b;
    for (int j = 0; j < 32; ++j) {
        int k = hyp == 0 ? 32 * h + j : (16 * h + (j & 15) + 32 * (j >> 4));
        a.b[j] = A[r * 64 + k];
        b.b[j] = B[r * 64 + k];
    }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D[row * 32 + r] = c[i];
    }
}

static uint8_t enc(int v) {      // small integers -4..4 as e4m3
    static const uint8_t pos[5] = {0x00, 0x38, 0x40, 0x44, 0x48};
    return v >= 0 ? pos[v] : (uint8_t)(pos[-v] | 0x80);
}

int main() {
    int Ai[32][64], Bi[32][64];
    uint8_t Ah[32 * 64], Bh[32 * 64];
    srand(1);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 64; ++k) {
        Ai[i][k] = rand() % 9 - 4; Bi[i][k] = rand() % 9 - 4;
        Ah[i * 64 + k] = enc(Ai[i][k]); Bh[i * 64 + k] = enc(Bi[i][k]);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof Ah); hipMalloc(&dB, sizeof Bh); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, Ah, sizeof Ah, hipMemcpyHostToDevice); hipMemcpy(dB, Bh, sizeof Bh, hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp) {
        probe<<<1, 64>>>(dA, dB, dD, hyp);
        float D[32 * 32];
        hipMemcpy(D, dD, sizeof D, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            int ref = 0;
            for (int k = 0; k < 64; ++k) ref += Ai[i][k] * Bi[j][k];
            bad += (D[i * 32 + j] != (float)ref);
        }
        printf("hypothesis %d (%s): %d mismatches of 1024\n", hyp, hyp == 0 ? "k = 32*(lane>>5) + byte" : "k = 16*(lane>>5) + byte%%16 + 32*(byte/16)", bad);
    }
    return 0;
}
