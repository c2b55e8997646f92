// ONE-WAY streaming between workgroups inside a running kernel ("probe one-way streaming mailboxes
// between workgroups - messages per second, not microseconds per round trip").  Run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/oneway tools/probe/oneway_stream.cpp && /tmp/oneway
//
// Shape of the level-systolic pipeline it prices: a chain of STAGES.  Stage 0 is one workgroup (the root's owner) that
// emits N elements in order; stage s has K_s workgroups; every element goes from the workgroup that handled it at stage
// s to workgroup hash(element, s + 1) % K_{s+1} of the next stage through a single-producer / single-consumer ring in
// HBM / L2 (one ring per (producer, consumer) pair, 16-byte messages).  A message is two naturally aligned 8-byte halves,
// each written by one `sc1` store and each carrying the ring pass's generation bit, so a half is never torn and a message
// is complete when both halves show the expected generation - no head / tail words, no fences, nothing flows back except
// a consumed counter published every 64 messages (the producer looks at it only when its ring could be full).
// The consumer's wave 0 polls all its producers at once (lane p reads the head slot of ring p -> me with `sc1` loads).
// "work" = what a stage does per element besides the hand-over: nothing, or the element's 256-byte row read from HBM
// and AND-popcounted against 51 rows held in LDS (a bf 50 node compare), by the whole workgroup.
// Checked: every element arrives exactly once at the last stage, and the elements a consumer gets from one producer
// arrive in increasing order wherever that producer has a single producer itself (stages 1 and 2: per-node FIFO = the
// reference's order per node; a tree node has exactly ONE producer, its parent's owner.  A stage-2 workgroup of this probe
// interleaves the streams of several stage-1 workgroups, so what it sends on is not globally increasing - the first version
// of the probe flagged exactly that at stage 3 and nowhere else).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
constexpr int R = 1024;        // ring entries (16 B each)
constexpr int MAXG = 64;       // producers a consumer polls with one wave-wide load
constexpr u64 SPIN_LIMIT = 2400000000ull;  // s_memtime ticks (shader clock, ~2.4 GHz: ~2 s) before a wait gives up (never hang the box; 0.17 s tripped once on a loaded run)

struct Params {
    u64* rings;        // [consumer][producer][R][2]
    uint32_t* consumed;  // [consumer][producer]
    uint32_t* abort_flag;
    uint32_t* last_seen;  // [wg][MAXG] order check at every consumer
    uint32_t* arrived;    // [N] at the last stage
    uint32_t* errors;
    uint32_t* diag;
    u64* diag64;       // first timeout: what the waiting side saw
    u64* state;        // [wg][4]: phase word (where thread 0 is), elements taken, exit reason, last element
    const uint8_t* rows;  // element rows (256 B each), N_ROWS of them
    u64* sink;
    int n_rows;
    int n_elems;
    int stages;         // number of stages including stage 0
    int k[8];           // workgroups per stage (k[0] = 1)
    int first[8];       // first workgroup id of a stage (in units of `stride`)
    int stride;         // 1: stages dealt round-robin over the XCDs; 8: everything on one XCD
    int work;           // 0 none, 1 row + 51-row compare
    int poll;           // 0: relaxed agent-scope (sc1) loads only; 1: + an agent-scope acquire fence (buffer_inv sc1) after every poll that
                        // found nothing; 2: system-scope (sc0 sc1) loads
};

__device__ __forceinline__ u64 ld_sc1(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ __launch_bounds__(256) void k_stream(Params P) {
    if (blockIdx.x % P.stride != 0) return;
    const int wg = blockIdx.x / P.stride;
    int stage = -1;
    for (int s = 0; s < P.stages; ++s)
        if (wg >= P.first[s] && wg < P.first[s] + P.k[s]) stage = s;
    if (stage < 0) return;
    const int me = wg - P.first[stage];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    __shared__ uint32_t s_node[51 * 64];   // a bf 50 node's centroids
    __shared__ uint32_t s_msg[4];
    __shared__ uint32_t s_part[4];
    for (int i = tid; i < 51 * 64; i += 256) s_node[i] = hash32(i * 2654435761u + wg);
    __syncthreads();
    const bool last = stage == P.stages - 1;
    const int nprod = stage == 0 ? 0 : P.k[stage - 1];
    const int ncons = last ? 0 : P.k[stage + 1];
    // producer side: my tail per consumer (lane c of wave 0 owns ring me -> c)
    uint32_t tail = 0, cons_seen = 0;
    // consumer side: my head per producer (lane p of wave 0)
    uint32_t head = 0;
    u64 acc = 0;
    const uint32_t my_share_end = (uint32_t)P.n_elems;
    uint32_t next_elem = 0;  // stage 0 only
    u64 t_last = __builtin_amdgcn_s_memtime();
    u64 taken = 0;
    int exit_reason = 0;
#define MARK(ph) do { if (tid == 0) st_sc1(P.state + (size_t)wg * 4, ((u64)(ph) << 32) | (uint32_t)taken); } while (0)
    while (true) {
        // ---- get the next element --------------------------------------------------------------
        uint32_t elem = 0xFFFFFFFFu, from = 0;
        MARK(1);
        if (stage == 0) {
            if (next_elem >= my_share_end) break;
            elem = next_elem++;
        } else {
            if (wave == 0) {
                u64 a = 0, b = 0;
                const u64* slot = P.rings + (((size_t)(P.first[stage] + me) * MAXG + (lane < nprod ? lane : 0)) * R + (head % R)) * 2;
                bool ready = false;
                while (true) {
                    if (lane < nprod) {
                        if (P.poll == 2) {
                            a = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            b = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        } else {
                            a = ld_sc1(slot);
                            b = ld_sc1(slot + 1);
                        }
                        const u64 gen = ((head / R) + 1) & 1;
                        ready = (a >> 63) == gen && (b >> 63) == gen;
                    }
                    const u64 m = __ballot(ready);
                    if (m) {
                        const int p = __ffsll((long long)m) - 1;
                        const uint32_t e = (uint32_t)__shfl((unsigned)(a & 0xFFFFFFFFu), p);
                        if (lane == p) {
                            head++;
                            if ((head & 63) == 0)
                                __hip_atomic_store(P.consumed + (size_t)(P.first[stage] + me) * MAXG + p, head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        if (lane == 0) { s_msg[0] = e; s_msg[1] = (uint32_t)p; }
                        t_last = __builtin_amdgcn_s_memtime();
                        break;
                    }
                    // nothing there: finished? (the abort flag doubles as the "all elements arrived" broadcast)
                    const uint32_t ab = __hip_atomic_load(P.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (ab != 0) { if (lane == 0) s_msg[0] = 0xFFFFFFFFu; break; }
                    if (__builtin_amdgcn_s_memtime() - t_last > SPIN_LIMIT) {
                        // what does the slot look like now - through the same sc1 loads, after an agent-scope acquire, and at system scope?
                        uint32_t first = 0;
                        if (lane == 0) first = atomicAdd(P.errors, 1000000u);
                        first = __shfl(first, 0);
                        if (lane < nprod && lane < 2) {  // (every workgroup that gives up: the one whose view is stale is not the first)
                            u64* dg = P.diag64 + ((size_t)wg * 2 + lane) * 8;
                            dg[0] = ((u64)stage << 48) | ((u64)me << 32) | head;
                            dg[1] = a; dg[2] = b;
                            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                            dg[3] = ld_sc1(slot); dg[4] = ld_sc1(slot + 1);
                            dg[5] = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            dg[6] = __hip_atomic_load(slot + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            dg[7] = 1;
                        }
                        if (lane == 0) { __hip_atomic_store(P.abort_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); s_msg[0] = 0xFFFFFFFFu; }
                        break;
                    }
                    if (P.poll == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    else __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            elem = s_msg[0];
            from = s_msg[1];
            __syncthreads();
            if (elem == 0xFFFFFFFFu) { exit_reason = 1; break; }
            // order per producer
            if (tid == 0 && stage <= 2) {
                uint32_t* ls = P.last_seen + (size_t)(P.first[stage] + me) * MAXG + from;
                const uint32_t prev = *ls;
                if (prev != 0 && elem + 1 <= prev) {
                    const uint32_t k = atomicAdd(P.errors, 1u);
                    if (k < 8) { P.diag[k * 4 + 0] = (uint32_t)stage; P.diag[k * 4 + 1] = (uint32_t)me * 64u + from; P.diag[k * 4 + 2] = elem; P.diag[k * 4 + 3] = prev - 1; }
                }
                *ls = elem + 1;
            }
        }
        taken++;
        MARK(2);
        // ---- the stage's own work ------------------------------------------------------------------
        if (P.work) {
            const uint32_t r = hash32(elem) % (uint32_t)P.n_rows;
            const uint32_t x = ((const uint32_t*)(P.rows + (size_t)r * 256))[lane];  // every wave reads the row (64 dwords)
            uint32_t best = 0;
            for (int row = wave; row < 51; row += 4) {
                uint32_t c = __popc(x & s_node[row * 64 + lane]);
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
                best = c > best ? c : best;
            }
            if (lane == 0) s_part[wave] = best;
            __syncthreads();
            acc += s_part[0] + s_part[1] + s_part[2] + s_part[3];
            __syncthreads();
        }
        // ---- hand over -----------------------------------------------------------------------------
        MARK(3);
        if (last) {
            if (tid == 0) {
                atomicAdd(P.arrived + elem, 1u);
                atomicAdd(P.errors + 1, 1u);  // errors[1] = elements that reached the end (the watcher ends the run on it)
            }
        } else if (wave == 0) {
            const int c = (int)(hash32(elem * 7919u + (uint32_t)stage) % (uint32_t)ncons);
            const int cw = P.first[stage + 1] + c;
            if (lane == c) {
                if (tail - cons_seen >= (uint32_t)(R - 64)) {
                    const u64 t0 = __builtin_amdgcn_s_memtime();
                    while (true) {
                        if (P.poll == 2) cons_seen = __hip_atomic_load(P.consumed + (size_t)cw * MAXG + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        else cons_seen = __hip_atomic_load(P.consumed + (size_t)cw * MAXG + me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (tail - cons_seen < (uint32_t)(R - 64)) break;
                        if (P.poll == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        if (__builtin_amdgcn_s_memtime() - t0 > SPIN_LIMIT ||
                            __hip_atomic_load(P.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 2u) {
                            if (atomicAdd(P.errors, 1000000u) < 1000000u) {
                                u64* dg = P.diag64 + 512 * 8;
                                dg[0] = ((u64)stage << 48) | ((u64)me << 32) | tail; dg[1] = ((u64)c << 32) | cons_seen; dg[7] = 2;
                            }
                            __hip_atomic_store(P.abort_flag, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                }
                u64* slot = P.rings + (((size_t)cw * MAXG + me) * R + (tail % R)) * 2;
                const u64 gen = ((tail / R) + 1) & 1;
                st_sc1(slot, (gen << 63) | (u64)elem);
                st_sc1(slot + 1, (gen << 63) | ((u64)stage << 32) | (u64)(elem ^ 0x5a5a5a5au));
                tail++;
            }
        }
    }
    if (tid == 0) {
        P.state[(size_t)wg * 4 + 1] = taken;
        P.state[(size_t)wg * 4 + 2] = 100 + exit_reason;
        P.sink[blockIdx.x] = acc;
    }
}

// the host ends the run: a tiny kernel on a second stream raises the "done" flag once errors[1] == N (or a timeout)
__global__ void k_watch(uint32_t* errors, uint32_t* abort_flag, uint32_t n) {
    const u64 t0 = __builtin_amdgcn_s_memtime();
    while (true) {
        const uint32_t d = __hip_atomic_load(errors + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d >= n) break;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
        if (__builtin_amdgcn_s_memtime() - t0 > 20 * SPIN_LIMIT) break;
        __builtin_amdgcn_s_sleep(8);
    }
    __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int main(int argc, char** argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int N = argc > 1 ? atoi(argv[1]) : 400000;
    const int REPS = argc > 2 ? atoi(argv[2]) : 2;
    const int ONLY_WORK = argc > 3 ? atoi(argv[3]) : -1;
    const char* ONLY_CFG = argc > 4 ? argv[4] : nullptr;
    const int POLL = argc > 5 ? atoi(argv[5]) : 1;
    const int NROWS = 65536;
    const size_t total_wg = 256;
    u64* rings; uint32_t *consumed, *abort_flag, *last_seen, *arrived, *errors, *diag; uint8_t* rows; u64* sink;
    const size_t ring_bytes = total_wg * MAXG * R * 16;
    CK(hipMalloc(&rings, ring_bytes));
    CK(hipMalloc(&consumed, total_wg * MAXG * 4));
    CK(hipMalloc(&abort_flag, 4));
    CK(hipMalloc(&last_seen, total_wg * MAXG * 4));
    CK(hipMalloc(&arrived, (size_t)N * 4));
    CK(hipMalloc(&errors, 8));
    CK(hipMalloc(&diag, 32 * 4));
    u64* diag64;
    CK(hipMalloc(&diag64, 513 * 8 * 8));
    u64* state;
    CK(hipMalloc(&state, 2048 * 4 * 8));
    CK(hipMalloc(&rows, (size_t)NROWS * 256));
    CK(hipMalloc(&sink, 2048 * 8));
    std::vector<uint8_t> h((size_t)NROWS * 256);
    uint32_t s = 777u;
    for (auto& b : h) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    CK(hipMemcpy(rows, h.data(), h.size(), hipMemcpyHostToDevice));
    hipStream_t s1, s2;
    CK(hipStreamCreate(&s1));
    CK(hipStreamCreate(&s2));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct Cfg { const char* name; int stages; int k[8]; };
    const Cfg cfgs[] = {
        {"1 -> 1", 2, {1, 1}},
        {"1 -> 8", 2, {1, 8}},
        {"1 -> 64", 2, {1, 64}},
        {"1 -> 1 -> 1", 3, {1, 1, 1}},
        {"1 -> 8 -> 64", 3, {1, 8, 64}},
        {"1 -> 4 -> 16 -> 64 (root, two levels, leaves)", 4, {1, 4, 16, 64}},
        {"1 -> 16 -> 32 -> 64", 4, {1, 16, 32, 64}},
    };
    printf("poll mode %d (0: sc1 loads; 1: + agent acquire fence after an empty poll; 2: system-scope loads), %d elements, %d repeats\n", POLL, N, REPS);
    printf("%-48s %6s %5s | %10s %10s  %s\n", "stages (workgroups per stage)", "stride", "work", "M msgs/s", "us/elem", "check");
    for (const Cfg& c : cfgs)
        for (int stride : {1, 8}) {
            if (ONLY_CFG && std::strcmp(c.name, ONLY_CFG) != 0) continue;
            for (int work : {0, 1}) {
                if (ONLY_WORK >= 0 && work != ONLY_WORK) continue;
                Params P{};
                P.rings = rings; P.consumed = consumed; P.abort_flag = abort_flag; P.last_seen = last_seen; P.arrived = arrived;
                P.errors = errors; P.diag = diag; P.diag64 = diag64; P.state = state; P.rows = rows; P.sink = sink; P.n_rows = NROWS; P.n_elems = N; P.stages = c.stages;
                int f = 0;
                for (int i = 0; i < c.stages; ++i) { P.k[i] = c.k[i]; P.first[i] = f; f += c.k[i]; }
                P.stride = stride; P.work = work; P.poll = POLL;
                if ((size_t)f * stride > 2048 || (stride == 8 && f > 32)) continue;  // one XCD has 32 CUs
                float best = 1e30f;
                uint32_t herr[2] = {0, 0};
                bool ok = true;
                for (int rep = 0; rep < REPS; ++rep) {
                    CK(hipMemset(rings, 0, ring_bytes));
                    CK(hipMemset(consumed, 0, total_wg * MAXG * 4));
                    CK(hipMemset(abort_flag, 0, 4));
                    CK(hipMemset(last_seen, 0, total_wg * MAXG * 4));
                    CK(hipMemset(arrived, 0, (size_t)N * 4));
                    CK(hipMemset(errors, 0, 8));
                    CK(hipMemset(diag, 0, 32 * 4));
                    CK(hipMemset(diag64, 0, 513 * 8 * 8));
                    CK(hipMemset(state, 0, 2048 * 4 * 8));
                    CK(hipDeviceSynchronize());
                    CK(hipEventRecord(e0, s1));
                    hipLaunchKernelGGL(k_stream, dim3(f * stride), dim3(256), 0, s1, P);
                    hipLaunchKernelGGL(k_watch, dim3(1), dim3(1), 0, s2, errors, abort_flag, (uint32_t)N);
                    CK(hipEventRecord(e1, s1));
                    CK(hipEventSynchronize(e1));
                    CK(hipDeviceSynchronize());
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    CK(hipMemcpy(herr, errors, 8, hipMemcpyDeviceToHost));
                    std::vector<uint32_t> ha(N);
                    CK(hipMemcpy(ha.data(), arrived, (size_t)N * 4, hipMemcpyDeviceToHost));
                    size_t bad = 0;
                    for (int i = 0; i < N; ++i) bad += ha[i] != 1u;
                    if (bad || herr[0]) ok = false;
                    if (bad || herr[0]) {
                        printf("   (rep %d: %zu elements not delivered exactly once, order/timeouts word %u, reached end %u)\n", rep, bad, herr[0], herr[1]);
                        std::vector<u64> h64(513 * 8);
                        CK(hipMemcpy(h64.data(), diag64, 513 * 8 * 8, hipMemcpyDeviceToHost));
                        for (int q = 0; q < 513; ++q)
                            if (h64[q * 8 + 7] == 1)
                                printf("      consumer timeout: stage %llu me %llu [wg*2+producer %d] head %llu (slot %llu, expected gen %llu): saw %016llx %016llx, after acquire %016llx %016llx, system scope %016llx %016llx\n",
                                       h64[q * 8] >> 48, (h64[q * 8] >> 32) & 0xFFFF, q, h64[q * 8] & 0xFFFFFFFFull, (h64[q * 8] & 0xFFFFFFFFull) % R,
                                       (((h64[q * 8] & 0xFFFFFFFFull) / R) + 1) & 1, h64[q * 8 + 1], h64[q * 8 + 2], h64[q * 8 + 3], h64[q * 8 + 4], h64[q * 8 + 5], h64[q * 8 + 6]);
                            else if (h64[q * 8 + 7] == 2)
                                printf("      producer timeout: stage %llu wg %llu tail %llu -> consumer %llu, consumed seen %llu\n", h64[q * 8] >> 48,
                                       (h64[q * 8] >> 32) & 0xFFFF, h64[q * 8] & 0xFFFFFFFFull, h64[q * 8 + 1] >> 32, h64[q * 8 + 1] & 0xFFFFFFFFull);
                        std::vector<u64> hs(2048 * 4);
                        CK(hipMemcpy(hs.data(), state, hs.size() * 8, hipMemcpyDeviceToHost));
                        for (int w = 0; w < f; ++w)
                            printf("      wg %3d: phase %llu taken(mark) %llu taken(exit) %llu exit %llu\n", w, hs[w * 4] >> 32, hs[w * 4] & 0xFFFFFFFFull, hs[w * 4 + 1], hs[w * 4 + 2]);
                        uint32_t hd[32];
                        CK(hipMemcpy(hd, diag, sizeof(hd), hipMemcpyDeviceToHost));
                        for (int q = 0; q < 8 && q < (int)(herr[0] % 1000000u); ++q)
                            printf("      order: stage %u consumer %u producer %u got element %u after %u\n", hd[q * 4], hd[q * 4 + 1] / 64, hd[q * 4 + 1] % 64, hd[q * 4 + 2], hd[q * 4 + 3]);
                    }
                }
                printf("%-48s %6d %5d | %10.3f %10.3f  %s\n", c.name, stride, work, N / (best * 1e3), best * 1e3 / N, ok ? "ok" : "FAILED");
            }
        }
    return 0;
}
