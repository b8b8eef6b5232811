// Lab (round 5, moved out of the product library in round 6): which compute units does bit i of a hipExtStreamCreateWithCUMask mask select, and can a
// masked stream be kept out of an XCD? Launches a probe kernel on streams with different masks and prints the XCC ids / HW_ID fields of its workgroups
// (gfx950: 256 CUs in 8 XCDs). Finding: profiles/r05_cu_mask_probe.txt. Build: scripts/lab/build.sh; run: scripts/lab/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <tuple>
#include <vector>

__global__ __launch_bounds__(256) void probe_placement_kernel(unsigned int* __restrict__ out, int spin) {
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);        // HW_REG_XCC_ID[3:0]
    const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID (wave / SIMD / CU / SH / SE ids)
    const unsigned t0 = (unsigned)__builtin_amdgcn_s_memtime();
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0000001f + 0.5f;                          // keeps the workgroup resident so that a launch spreads over its CUs
    if (threadIdx.x == 0) { out[3 * blockIdx.x] = xcc; out[3 * blockIdx.x + 1] = hwid; out[3 * blockIdx.x + 2] = t0 + (x == 123.f ? 1u : 0u); }
}

static void show(const char* name, const std::vector<int>& bits, int wgs) {
    std::vector<uint32_t> mask(8, 0u);
    for (int b : bits) mask[b / 32] |= 1u << (b % 32);
    hipStream_t st = nullptr;
    if (!bits.empty()) { if (hipExtStreamCreateWithCUMask(&st, 8, mask.data()) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", name); return; } }
    unsigned int* d = nullptr;
    hipMalloc(&d, sizeof(unsigned int) * 3 * wgs);
    hipMemset(d, 0, sizeof(unsigned int) * 3 * wgs);
    probe_placement_kernel<<<wgs, 256, 0, st>>>(d, 20000);
    hipStreamSynchronize(st);
    std::vector<unsigned int> h(3 * wgs);
    hipMemcpy(h.data(), d, sizeof(unsigned int) * 3 * wgs, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_xcc;
    std::set<std::tuple<unsigned, unsigned, unsigned, unsigned>> cus;
    for (int w = 0; w < wgs; ++w) {
        const unsigned x = h[3 * w], id = h[3 * w + 1];
        ++per_xcc[x];
        cus.insert({x, (id >> 13) & 7, (id >> 12) & 1, (id >> 8) & 15});
    }
    printf("%-28s bits=%3zu  workgroups per XCC {", name, bits.size());
    for (auto& kv : per_xcc) printf(" %u: %d", kv.first, kv.second);
    printf(" }  distinct (xcc, se, sh, cu): %zu\n", cus.size());
    hipFree(d);
    if (st) hipStreamDestroy(st);
}

int main() {
    show("unmasked, 2048 workgroups", {}, 2048);
    for (int b : {0, 1, 2, 7, 8, 9, 31, 32, 33, 64, 255}) { char n[32]; snprintf(n, sizeof n, "single bit %d", b); show(n, {b}, 64); }
    std::vector<int> low, even;
    for (int b = 0; b < 128; ++b) low.push_back(b);
    for (int b = 0; b < 256; b += 2) even.push_back(b);
    show("bits 0..127", low, 512);
    show("even bits", even, 512);
    return 0;
}
