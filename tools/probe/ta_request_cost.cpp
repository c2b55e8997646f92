// What a block of row requests costs one CU, by how the rows beyond a node's end are handled (a bf 1000 level
// is bound by the requests for rows that do not exist).  Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/ta tools/probe/ta_request_cost.cpp && /tmp/ta
//
// One workgroup of 256 threads, the layout of node_best's block path: 16 lanes per row (16 B each), 16 rows per pass, 16 passes
// = one block of 256 rows x 256 B.  Per iteration one block is requested and waited for (the data is summed so the
// loads cannot be dropped); `live` rows of the 256 exist.  Variants for the rows beyond `live`:
//   0 all      every row requested from its own address (what the engine did until round 4's `len` walk)
//   1 clamped  requests beyond the end clamped to the last live row (the engine now)
//   2 masked   lanes beyond the end do not load (a per-lane branch: exec mask)
//   3 bounded  raw buffer loads, num_records = live x 256: lanes beyond the end are out of range and return 0
// Cycles (s_memtime) per block, thread 0; the checksum of the data that arrived must agree between `masked` and `bounded` at
// the same `live` (rows beyond the end read as zero in both) and between `all` and every variant at live = 256.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_req(const uint8_t* __restrict__ cent, int nodes, int live, int iters, unsigned long long* out) {
    const int tid = threadIdx.x, l = tid & 15, g = tid >> 4;
    u32x4 acc = (u32x4)(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const uint8_t* node = cent + (size_t)(it % nodes) * 256 * 256;
        u32x4 d[16];
        if constexpr (MODE == 3) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)node, 0, live * 256, 0x00020000);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int r = p * 16 + g;
                d[p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, r * 256 + l * 16, 0, 0));
            }
        } else {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                int r = p * 16 + g;
                if constexpr (MODE == 1) r = r < live ? r : live - 1;
                if constexpr (MODE == 2) {
                    d[p] = (u32x4)(0);
                    if (r < live) d[p] = *(const u32x4*)(node + (size_t)r * 256 + l * 16);
                } else {
                    d[p] = *(const u32x4*)(node + (size_t)r * 256 + l * 16);
                }
            }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) acc += d[p];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[0] = t1 - t0;
    atomicAdd(&out[1], (unsigned long long)(acc.x ^ acc.y ^ acc.z ^ acc.w) * (unsigned long long)(tid + 1));  // (keeps the loads; a checksum of what arrived)
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int nodes = 8, iters = 4000;  // 8 nodes x 64 KB: L2-resident after the first pass
    std::vector<uint8_t> h((size_t)nodes * 256 * 256);
    uint32_t s = 777u;
    for (auto& b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    uint8_t* cent;
    unsigned long long* out;
    CK(hipMalloc(&cent, h.size()));
    CK(hipMalloc(&out, 16));
    CK(hipMemcpy(cent, h.data(), h.size(), hipMemcpyHostToDevice));
    const char* names[4] = {"all", "clamped", "masked", "bounded"};
    printf("%8s %5s | %s\n", "variant", "live", "cycles per 256-row block (request + wait), one workgroup");
    const int lives[] = {256, 192, 128, 64, 16, 4};
    for (int mode = 0; mode < 4; ++mode)
        for (int live : lives) {
            unsigned long long best = ~0ull, sum = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(out, 0, 16));
                switch (mode) {
                    case 0: hipLaunchKernelGGL(k_req<0>, dim3(1), dim3(256), 0, 0, cent, nodes, live, iters, out); break;
                    case 1: hipLaunchKernelGGL(k_req<1>, dim3(1), dim3(256), 0, 0, cent, nodes, live, iters, out); break;
                    case 2: hipLaunchKernelGGL(k_req<2>, dim3(1), dim3(256), 0, 0, cent, nodes, live, iters, out); break;
                    default: hipLaunchKernelGGL(k_req<3>, dim3(1), dim3(256), 0, 0, cent, nodes, live, iters, out); break;
                }
                CK(hipDeviceSynchronize());
                unsigned long long c[2] = {0, 0};
                CK(hipMemcpy(c, out, 16, hipMemcpyDeviceToHost));
                best = c[0] < best ? c[0] : best;
                sum = c[1];
            }
            printf("%8s %5d | %8.0f   data checksum %016llx\n", names[mode], live, (double)best / iters, sum);
        }
    return 0;
}
