// What one tree level would cost if a node's compare were spread over several workgroups (VERDICT r4 items 7 / 9:
// "use more than one CU for one tree", "bf 1000 across CUs").  Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/xcu tools/probe/xcu_node_compare.cpp && /tmp/xcu
//
// A chain of ITER dependent queries (the next query vector is chosen by the previous best row, as the next insertion's
// descent depends on the previous one's updates).  Per query, P workgroups of 256 threads each take rows
// [p R / P, (p+1) R / P) of an R-row node of 256-byte centroids (16 lanes per row, 16 B per lane, popcounts met by
// DPP - the layout of node_best in bb_tree.hip), the master (workgroup 0) publishes the query with a release store
// of a sequence number, the helpers spin on it, write their slice's best key and the master merges.  P = 1 is the
// single-workgroup cost; R = 0 is the bare rendezvous.  `stride` 8 puts every participant on the master's XCD
// (workgroups are dealt round-robin to the 8 XCDs), stride 1 on different XCDs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Mail {
    uint32_t seq;       // published query number (release / acquire, agent scope)
    uint32_t q;         // the query's row in `queries`
    uint32_t pad[30];
    struct { uint32_t seq; uint32_t pad0; unsigned long long key; uint32_t pad[28]; } res[64];  // one 128 B line per helper
};

template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t x) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t row16_sum(uint32_t x) {
    x += dpp<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp<0x141>(x);  // row_half_mirror
    x += dpp<0x140>(x);  // row_mirror
    return x;
}

template <int REL, int ACQ>
__global__ __launch_bounds__(256) void k_fan(const uint8_t* __restrict__ cent, const uint8_t* __restrict__ queries, Mail* mail,
                                             unsigned long long* out, int R, int NN, int NQ, int P, int stride, int iters) {
    if (blockIdx.x % stride != 0) return;
    const int p = blockIdx.x / stride;
    if (p >= P) return;
    __shared__ unsigned long long s_best[4];
    __shared__ unsigned long long s_h[64];
    __shared__ uint32_t s_q;
    const int tid = threadIdx.x, l = tid & 15, g = tid >> 4, w = tid >> 6;
    const int r_lo = (int)((long long)p * R / P), r_hi = (int)((long long)(p + 1) * R / P);
    uint32_t q = 0;
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t want = (uint32_t)it + 1u;
        if (P > 1) {
            if (p == 0) {
                if (tid == 0) {
                    __hip_atomic_store(&mail->q, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&mail->seq, want, REL, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (tid == 0) {
                    while (__hip_atomic_load(&mail->seq, ACQ, __HIP_MEMORY_SCOPE_AGENT) != want) {}
                    s_q = __hip_atomic_load(&mail->q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                q = s_q;
            }
        }
        // this workgroup's slice of node (it % NN)
        const uint8_t* node = cent + (size_t)(it % NN) * (size_t)(R > 0 ? R : 1) * 256;
        const u32x4 xv = *(const u32x4*)(queries + (size_t)q * 256 + l * 16);
        unsigned long long best = 0;
        for (int r0 = r_lo; r0 < r_hi; r0 += 64) {
            u32x4 d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int r = r0 + k * 16 + g;
                r = r < r_hi ? r : r_hi - 1;
                d[k] = __builtin_nontemporal_load((const u32x4*)(node + (size_t)r * 256 + l * 16));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = r0 + k * 16 + g;
                const u32x4 a = d[k] & xv;
                const uint32_t pi = __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w);
                const uint32_t pr = __popc(d[k].x) + __popc(d[k].y) + __popc(d[k].z) + __popc(d[k].w);
                const uint32_t both = row16_sum(pi + (pr << 16));
                const uint32_t inter = both & 0xFFFFu;
                uint32_t un = (both >> 16) + 1024u - inter;
                un = un < 1u ? 1u : un;
                const unsigned long long key = r < r_hi ? (((unsigned long long)((inter << 16) / un)) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)r) : 0ull;
                best = key > best ? key : best;
            }
        }
        // the wave's best, then the workgroup's
        for (int o = 32; o >= 16; o >>= 1) {
            const unsigned long long other = __shfl_xor(best, o);
            best = other > best ? other : best;
        }
        if ((tid & 63) == 0) s_best[w] = best;
        __syncthreads();
        unsigned long long wg = s_best[0];
        for (int k = 1; k < 4; ++k) wg = s_best[k] > wg ? s_best[k] : wg;
        if (P > 1) {
            if (p != 0) {
                if (tid == 0) {
                    __hip_atomic_store(&mail->res[p].key, wg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&mail->res[p].seq, want, REL, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                if (tid > 0 && tid < P) {
                    while (__hip_atomic_load(&mail->res[tid].seq, ACQ, __HIP_MEMORY_SCOPE_AGENT) != want) {}
                    s_h[tid] = __hip_atomic_load(&mail->res[tid].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                for (int k = 1; k < P; ++k) wg = s_h[k] > wg ? s_h[k] : wg;
            }
        }
        __syncthreads();
        acc += wg;
        q = (uint32_t)((wg ^ (wg >> 32) ^ (unsigned long long)it) % (unsigned long long)NQ);  // the next query depends on this result
    }
    if (p == 0 && tid == 0) out[0] = acc;
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    const int NQ = 4096, NNMAX = 512, RMAX = 1001;
    std::vector<uint8_t> h((size_t)NNMAX * RMAX * 256);
    uint32_t s = 12345u;
    for (auto& b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24) & (uint8_t)(s >> 16); }
    std::vector<uint8_t> hq((size_t)NQ * 256);
    for (auto& b : hq) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    uint8_t *cent, *queries;
    Mail* mail;
    unsigned long long* out;
    CK(hipMalloc(&cent, h.size()));
    CK(hipMalloc(&queries, hq.size()));
    CK(hipMalloc(&mail, sizeof(Mail)));
    CK(hipMalloc(&out, 8));
    CK(hipMemcpy(cent, h.data(), h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(queries, hq.data(), hq.size(), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%4s %5s %4s %3s %6s | %9s  %s\n", "mode", "rows", "NN", "P", "stride", "us/query", "checksum");
    const int Rs[] = {0, 254, 1001};
    const int NNs[] = {8, 512};
    const int Ps[] = {1, 2, 4, 8, 16, 32};
    const int strides[] = {1, 8};
    // mode 0: release / acquire at agent scope (what a real exchange of updated rows needs: L2 write-back / invalidate across
    // XCDs); mode 1: relaxed atomics on the mailbox only (no cache maintenance: the floor of the rendezvous itself - correct
    // here only because the probe's node rows never change)
    for (int R : Rs)
        for (int NN : NNs) {
            if (R == 0 && NN != 8) continue;
            unsigned long long ref = 0;
            for (int P : Ps)
                for (int stride : strides)
                for (int mode = 0; mode < 2; ++mode) {
                    if (P == 1 && (stride != 1 || mode != 0)) continue;
                    float best_ms = 1e30f;
                    unsigned long long sum = 0;
                    for (int rep = 0; rep < 2; ++rep) {
                        CK(hipMemset(mail, 0, sizeof(Mail)));
                        CK(hipDeviceSynchronize());
                        CK(hipEventRecord(e0));
                        if (mode == 0)
                            hipLaunchKernelGGL((k_fan<__ATOMIC_RELEASE, __ATOMIC_ACQUIRE>), dim3(P * stride), dim3(256), 0, 0, cent, queries, mail, out, R, NN, NQ, P, stride, iters);
                        else
                            hipLaunchKernelGGL((k_fan<__ATOMIC_RELAXED, __ATOMIC_RELAXED>), dim3(P * stride), dim3(256), 0, 0, cent, queries, mail, out, R, NN, NQ, P, stride, iters);
                        CK(hipEventRecord(e1));
                        CK(hipEventSynchronize(e1));
                        float ms = 0;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        best_ms = ms < best_ms ? ms : best_ms;
                        CK(hipMemcpy(&sum, out, 8, hipMemcpyDeviceToHost));
                    }
                    if (P == 1) ref = sum;
                    printf("%4d %5d %4d %3d %6d | %9.3f  %016llx%s\n", mode, R, NN, P, stride, 1e3 * best_ms / iters, sum,
                           sum == ref ? "" : "  (differs from P=1)");
                }
        }
    return 0;
}
