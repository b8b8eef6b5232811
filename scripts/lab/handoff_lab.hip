// Lab (round 6, review item 2 -- the persistent decoder block): what does ONE dependency edge of a cached decode token-step cost inside a persistent
// launch, against the kernel boundary it would replace, AT THIS MODEL'S SIZES?
//
// A token-step of the BERT-6 decoder at 32 rows is a chain of 42 weight-streaming kernels (7 per layer): each of ~256 workgroups streams its slice of
// a weight matrix (0.6 - 4.7 MB per matrix: 2 - 18 KB per workgroup) and needs the WHOLE activation vector of the previous kernel -- 32 rows x 768
// bf16 = 48 KB (x 3072 = 192 KB behind the FFN up-projection). The kernels take 4.9 - 7.4 us each for ~1 us of streaming: launch + dependent round trips.
// MI355X_MICROARCH.md prices the alternatives for 4 - 32 KB vectors (rows handoff-*, allgather, prefetch-credit, engine-vs-launches: a persistent
// batch-1 layer = 0.87 - 0.89 x of five launches). This program measures them for OUR vector sizes with the same skeleton in every variant:
//
//   phase e of workgroup w:  stream its weight slice (Wb bytes, HBM-cold)  +  read ALL of V[e-1] (P bytes)  ->  write its 1/256 of V[e]
//     V[e][i] = mix(e, i, xor of every word of V[e-1], xor of the workgroup's weight words)      (any lost / stale word changes every later vector)
//
//   L  launches   : E kernels replayed from one hipGraph (what the decode loop does today)
//   G  granules   : ONE persistent launch; V travels as 8-byte {tag = e + 1, value} granules, sc1 stores, consumers sweep with sc1 loads until every tag
//                   matches (Guideline 16 R2); the next phase's weight loads are issued BEFORE the sweep (prefetch-credit)
//   F  flags      : ONE persistent launch; V as 16-byte sc1 stores, every storing wave drains, one flag word per workgroup; one wave polls the 256 flags,
//                   then every wave reads V with sc1 loads (Guideline 16 R1 + the sc1-load form); weights prefetched as in G
//
// Build: scripts/lab/build.sh.  Run: scripts/lab/handoff_lab            (prints us per edge for P = 4 / 48 / 192 KB, with and without the weight stream)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NWG = 256, NT = 256;

struct Args {
    const uint4* W;            // weight pool (HBM-cold: every phase of every repetition reads another region)
    long w_words16;            // 16-byte words of the pool
    int wb16;                  // 16-byte words of a workgroup's slice per phase (0: no weight stream)
    unsigned* V0; unsigned* V1;      // plain vectors (variants L, F): two buffers, phase e writes buffer e & 1
    unsigned long long* G0; unsigned long long* G1;      // granule vectors (variant G)
    unsigned* flags;           // [2][NWG] (variant F)
    unsigned* tmo;             // timeout word (bounded spins)
    int pw;                    // 32-bit words of V (P / 4)
    int E;                     // phases
    long rep_off;              // pool offset of this repetition
};

__device__ __forceinline__ unsigned mix(unsigned e, unsigned i, unsigned s, unsigned w) {
    unsigned x = (e + 1u) * 0x9E3779B9u ^ (i * 0x85EBCA6Bu) ^ s ^ (w * 0xC2B2AE35u);
    x ^= x >> 15; x *= 0x2C1B3C6Du; x ^= x >> 12;
    return x;
}

__device__ __forceinline__ unsigned wg_xor(unsigned v, unsigned* red) {          // xor over the workgroup (4 waves)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v ^= __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] ^ red[1] ^ red[2] ^ red[3];
}

// the workgroup's weight slice of phase e: wb16 16-byte words, strided by thread; returns the xor of the loaded words (after they landed)
#define W_ISSUE(e_, buf_)                                                                                                              \
    do {                                                                                                                               \
        const long base_ = (a.rep_off + ((long)(e_) * NWG + blockIdx.x) * (long)a.wb16) % (a.w_words16 - a.wb16 - NT);                 \
        _Pragma("unroll") for (int k = 0; k < 5; ++k) { const int idx = threadIdx.x + k * NT; buf_[k] = idx < a.wb16 ? a.W[base_ + idx] : make_uint4(0, 0, 0, 0); } \
    } while (0)
#define W_XOR(buf_) (buf_[0].x ^ buf_[1].y ^ buf_[2].z ^ buf_[3].w ^ buf_[4].x ^ buf_[0].w ^ buf_[1].x ^ buf_[2].y ^ buf_[3].z)

// ---------------------------------------------------------------------------------------------- L: one kernel per phase
__global__ __launch_bounds__(NT) void phase_kernel(const Args a, const int e) {
    __shared__ unsigned red[4];
    uint4 wbuf[5];
    W_ISSUE(e, wbuf);
    const unsigned* Vin = (e & 1) ? a.V0 : a.V1;             // phase e reads what phase e - 1 wrote: buffer (e - 1) & 1
    unsigned* Vout = (e & 1) ? a.V1 : a.V0;
    unsigned s = 0u;
    if (e > 0) {
        const uint4* v4 = reinterpret_cast<const uint4*>(Vin);
        for (int i = threadIdx.x; i < a.pw / 4; i += NT) { const uint4 v = v4[i]; s ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    s = wg_xor(s, red);
    const unsigned w = wg_xor(W_XOR(wbuf), red);
    const int per = a.pw / NWG;
    for (int i = threadIdx.x; i < per; i += NT) Vout[blockIdx.x * per + i] = mix(e, blockIdx.x * per + i, s, w);
}

// ---------------------------------------------------------------------------------------------- G: persistent, tagged granules
__global__ __launch_bounds__(NT) void granule_kernel(const Args a) {
    __shared__ unsigned red[4];
    uint4 wbuf[5];
    W_ISSUE(0, wbuf);
    const int per = a.pw / NWG;
    for (int e = 0; e < a.E; ++e) {
        gu64* Gin = (gu64*)((e & 1) ? a.G0 : a.G1);
        gu64* Gout = (gu64*)((e & 1) ? a.G1 : a.G0);
        unsigned s = 0u;
        if (e > 0) {
            // sweep: every lane owns pw / NT granules (strided); re-read those not yet tagged e; bounded
            const unsigned tag = (unsigned)e;                // phase e - 1 wrote tag (e - 1) + 1
            for (int i0 = threadIdx.x; i0 < a.pw; i0 += NT * 8) {
                unsigned long long x[8];
                unsigned spins = 0;
                bool ok;
                do {
                    ok = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const int i = i0 + k * NT;
                        x[k] = i < a.pw ? __hip_atomic_load(Gin + i, RLX_AGENT) : ((unsigned long long)tag << 32);
                        ok &= (unsigned)(x[k] >> 32) == tag;
                    }
                    if (!ok && ++spins > 2000000u) { a.tmo[0] = 1u; ok = true; }
                } while (!ok);
#pragma unroll
                for (int k = 0; k < 8; ++k) s ^= (i0 + k * NT < a.pw) ? (unsigned)x[k] : 0u;
            }
        }
        s = wg_xor(s, red);
        const unsigned w = wg_xor(W_XOR(wbuf), red);
        if (e + 1 < a.E) W_ISSUE(e + 1, wbuf);               // the next phase's weights are on their way while this phase publishes and the next sweep polls
        for (int i = threadIdx.x; i < per; i += NT) {
            const int gi = blockIdx.x * per + i;
            __hip_atomic_store(Gout + gi, ((unsigned long long)(e + 1) << 32) | mix(e, gi, s, w), RLX_AGENT);
        }
    }
}

// ---------------------------------------------------------------------------------------------- F: persistent, sc1 payload + one flag per workgroup
__global__ __launch_bounds__(NT) void flag_kernel(const Args a) {
    __shared__ unsigned red[4];
    uint4 wbuf[5];
    W_ISSUE(0, wbuf);
    const int per = a.pw / NWG;
    for (int e = 0; e < a.E; ++e) {
        const unsigned* Vin = (e & 1) ? a.V0 : a.V1;
        unsigned* Vout = (e & 1) ? a.V1 : a.V0;
        gu32* fin = (gu32*)(a.flags + ((e - 1) & 1) * NWG);
        gu32* fout = (gu32*)(a.flags + (e & 1) * NWG);
        unsigned s = 0u;
        if (e > 0) {
            if (threadIdx.x < 64) {                          // ONE wave polls the 256 flag words (4 per lane), relaxed sc1 loads
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) ok &= __hip_atomic_load(fin + threadIdx.x * 4 + k, RLX_AGENT) == (unsigned)e;
                    if (__all(ok)) break;
                    if (++spins > 2000000u) { a.tmo[0] = 2u; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            // every load of the handed-off bytes is an sc1 load to registers (MI355X_MICROARCH.md, valid forms): 16-byte buffer loads with aux = sc1,
            // four in flight per lane
            const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vin, 0, a.pw * 4, 0x00020000);
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            const int n16 = a.pw / 4;
            int i = threadIdx.x;
            for (; i + 3 * NT < n16; i += 4 * NT) {
                const u32x4 v0 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i * 16, 0, 16), v1 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i + NT) * 16, 0, 16);
                const u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i + 2 * NT) * 16, 0, 16), v3 = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i + 3 * NT) * 16, 0, 16);
                s ^= v0[0] ^ v0[1] ^ v0[2] ^ v0[3] ^ v1[0] ^ v1[1] ^ v1[2] ^ v1[3] ^ v2[0] ^ v2[1] ^ v2[2] ^ v2[3] ^ v3[0] ^ v3[1] ^ v3[2] ^ v3[3];
            }
            for (; i < n16; i += NT) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, i * 16, 0, 16); s ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
        }
        s = wg_xor(s, red);
        const unsigned w = wg_xor(W_XOR(wbuf), red);
        if (e + 1 < a.E) W_ISSUE(e + 1, wbuf);
        {
            typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
            const auto wsrc = __builtin_amdgcn_make_buffer_rsrc((void*)Vout, 0, a.pw * 4, 0x00020000);
            for (int i4 = threadIdx.x; i4 < per / 4; i4 += NT) {        // 16-byte write-through stores (per is a multiple of 4 words: 4 KB / 256 = 16 B)
                const int i = blockIdx.x * per + i4 * 4;
                u32x4 v; v[0] = mix(e, i, s, w); v[1] = mix(e, i + 1, s, w); v[2] = mix(e, i + 2, s, w); v[3] = mix(e, i + 3, s, w);
                __builtin_amdgcn_raw_buffer_store_b128(v, wsrc, i * 4, 0, 16);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(fout + blockIdx.x, (unsigned)(e + 1), RLX_AGENT);
    }
}

static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const int E = 42, REPS = 21;
    const long pool_bytes = 1L << 30;
    uint4* W; HC(hipMalloc(&W, pool_bytes));
    HC(hipMemset(W, 0x5a, pool_bytes));
    unsigned *V0, *V1, *flags, *tmo; unsigned long long *G0, *G1;
    const int maxpw = 192 * 1024 / 4;
    HC(hipMalloc(&V0, maxpw * 4)); HC(hipMalloc(&V1, maxpw * 4)); HC(hipMalloc(&G0, maxpw * 8)); HC(hipMalloc(&G1, maxpw * 8));
    HC(hipMalloc(&flags, 2 * NWG * 4)); HC(hipMalloc(&tmo, 16));
    hipStream_t st; HC(hipStreamCreate(&st));
    printf("%d phases per chain, %d workgroups x %d threads, one per CU; us per edge = chain time / %d (median of %d chains, weight pool 1 GiB, every chain another region)\n", E, NWG, NT, E, REPS);
    for (int wbk : {0, 18}) {
        for (int pk : {4, 48, 192}) {
            Args a; a.W = W; a.w_words16 = pool_bytes / 16; a.wb16 = wbk * 1024 / 16; a.V0 = V0; a.V1 = V1; a.G0 = G0; a.G1 = G1; a.flags = flags; a.tmo = tmo;
            a.pw = pk * 1024 / 4; a.E = E; a.rep_off = 0;
            unsigned sums[3] = {0, 0, 0};
            double us[3] = {0, 0, 0};
            for (int variant = 0; variant < 3; ++variant) {
                // variant L as a graph of E launches (captured once per repetition offset would freeze rep_off: the graph is re-captured per repetition, its
                // instantiation outside the timed region)
                std::vector<double> t;
                for (int rep = 0; rep < REPS; ++rep) {
                    a.rep_off = ((long)rep * 7919L + variant * 131L) * (long)(a.wb16 + 1) * NWG * E % (a.w_words16 / 2);
                    HC(hipMemsetAsync(flags, 0, 2 * NWG * 4, st)); HC(hipMemsetAsync(tmo, 0, 16, st));
                    HC(hipMemsetAsync(G0, 0, maxpw * 8, st)); HC(hipMemsetAsync(G1, 0, maxpw * 8, st));
                    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
                    if (variant == 0) {
                        HC(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                        for (int e = 0; e < E; ++e) phase_kernel<<<NWG, NT, 0, st>>>(a, e);
                        HC(hipStreamEndCapture(st, &g));
                        HC(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    }
                    HC(hipStreamSynchronize(st));
                    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
                    HC(hipEventRecord(e0, st));
                    if (variant == 0) HC(hipGraphLaunch(ge, st));
                    else if (variant == 1) granule_kernel<<<NWG, NT, 0, st>>>(a);
                    else flag_kernel<<<NWG, NT, 0, st>>>(a);
                    HC(hipEventRecord(e1, st));
                    HC(hipStreamSynchronize(st));
                    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0) t.push_back(ms * 1e3 / E);
                    HC(hipEventDestroy(e0)); HC(hipEventDestroy(e1));
                    if (ge) { HC(hipGraphExecDestroy(ge)); HC(hipGraphDestroy(g)); }
                }
                us[variant] = median(t);
                // checksum of the last vector (rep_off differs per variant -> weight xor equal anyway: the pool is constant bytes)
                std::vector<unsigned> h(a.pw);
                if (variant == 1) {
                    std::vector<unsigned long long> hg(a.pw);
                    HC(hipMemcpy(hg.data(), ((E - 1) & 1) ? G1 : G0, a.pw * 8, hipMemcpyDeviceToHost));
                    for (int i = 0; i < a.pw; ++i) h[i] = (unsigned)hg[i];
                } else HC(hipMemcpy(h.data(), ((E - 1) & 1) ? V1 : V0, a.pw * 4, hipMemcpyDeviceToHost));
                unsigned c = 0; for (unsigned v : h) c = c * 31u + v;
                sums[variant] = c;
                unsigned ht[4]; HC(hipMemcpy(ht, tmo, 16, hipMemcpyDeviceToHost));
                if (ht[0]) printf("  !! variant %d timed out (code %u)\n", variant, ht[0]);
            }
            printf("weights %2d KB/WG/phase  vector %3d KB:  launches %6.2f   granules %6.2f (%.2fx)   flags %6.2f (%.2fx)   checksums %s\n", wbk, pk, us[0], us[1], us[1] / us[0],
                   us[2], us[2] / us[0], (sums[0] == sums[1] && sums[1] == sums[2]) ? "equal" : "DIFFER");
        }
    }
    return 0;
}
