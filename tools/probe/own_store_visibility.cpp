// Does a workgroup see its OWN global stores?  Thread 0 (wave 0) stores, a thread of another wave loads the same word after a
// __syncthreads() - which on gfx950 waits for LDS only (s_waitcnt lgkmcnt(0); s_barrier), not for the store (vmcnt).
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/osv tools/probe/own_store_visibility.cpp && /tmp/osv
// mode 0: store, barrier, load                      (what the tree engines do between two steps)
// mode 1: store, s_waitcnt vmcnt(0), barrier, load  (the store has been acknowledged before anyone loads)
// mode 2: as 1, plus an agent-scope acquire (buffer_inv sc1) before the load
// Each iteration first LOADS the word (so that its line sits in the CU's L1), then stores word + 1, then re-reads it; other
// traffic (a 64 KB sweep every `sweep` iterations) moves lines in and out of the L1.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// LOAD: 0 plain global load, 1 relaxed agent-scope atomic load (sc1), 2 volatile (flat sc0 sc1: system scope)
template <int LOAD>
__device__ __forceinline__ uint32_t ld(uint32_t* p) {
    uint32_t v;
    if (LOAD == 0) {
        asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    } else if (LOAD == 1) {
        v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        v = *(volatile uint32_t*)p;
    }
    return v;
}

template <int MODE, int LOAD>
__global__ __launch_bounds__(256) void k(uint32_t* words, const uint32_t* filler, uint32_t* bad, int iters, int nwords, int sweep) {
    __shared__ uint32_t s_v;
    const int tid = threadIdx.x;
    uint32_t* w = words + (size_t)blockIdx.x * nwords * 64;
    uint32_t acc = 0, stale = 0;
    for (int it = 0; it < iters; ++it) {
        const int i = (it * 7) % nwords;
        uint32_t before = 0;
        if (tid == 200) before = ld<LOAD>(w + i * 64);  // (256 B apart: a line each)
        if (tid == 200) s_v = before;
        __syncthreads();
        if (tid == 0) {
            w[i * 64] = s_v + 1;
            if (MODE >= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (sweep && (it % sweep) == 0) acc += filler[(it * 256 + tid) & 0xFFFF];
        __syncthreads();
        if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (tid == 200) {
            const uint32_t after = ld<LOAD>(w + i * 64);
            if (after != before + 1) stale++;
        }
        __syncthreads();
    }
    if (tid == 200 && stale) atomicAdd(bad, stale);
    if (acc == 0xFFFFFFFFu) bad[1] = acc;
}

int main() {
    const int nwords = 64, blocks = 64, iters = 200000;
    uint32_t *words, *filler, *bad;
    CK(hipMalloc(&words, (size_t)blocks * nwords * 64 * 4));
    CK(hipMalloc(&filler, 65536 * 4));
    CK(hipMalloc(&bad, 8));
    CK(hipMemset(filler, 1, 65536 * 4));
    for (int sweep : {0, 16})
      for (int load = 0; load < 3; ++load)
        for (int mode = 0; mode < 3; ++mode) {
            CK(hipMemset(words, 0, (size_t)blocks * nwords * 64 * 4));
            CK(hipMemset(bad, 0, 8));
#define L(M, Q) if (mode == M && load == Q) hipLaunchKernelGGL((k<M, Q>), dim3(blocks), dim3(256), 0, 0, words, filler, bad, iters, nwords, sweep);
            L(0, 0) L(1, 0) L(2, 0) L(0, 1) L(1, 1) L(2, 1) L(0, 2) L(1, 2) L(2, 2)
            CK(hipDeviceSynchronize());
            uint32_t h = 0;
            CK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost));
            printf("sweep %2d load %s mode %d (%s): %u stale re-reads of %lld\n", sweep, load == 0 ? "plain " : (load == 1 ? "sc1   " : "sc0sc1"), mode,
                   mode == 0 ? "store, barrier, load" : (mode == 1 ? "store, vmcnt(0), barrier, load" : "store, vmcnt(0), barrier, acquire, load"), h, (long long)blocks * iters);
        }
    return 0;
}
