// bb_tree.hip -- HBM-resident BitBIRCH tree engine for gfx950 (MI355X, CDNA4).
//
// Replaces the per-fingerprint Python loop of the reference (bblean/bitbirch.py:769-787
// and :848-866) and everything it calls: _BFNode.insert_bf_subcluster (:305-357),
// _BFSubcluster.merge_subcluster / update (:488-526), the merge criteria (_merges.py),
// centroid_from_sum (_py_similarity.py:12-42), jt_isim_from_sum (similarity.cpp:273-301),
// _jt_sim_arr_vec_packed + np.argmax (similarity.cpp:374, bitbirch.py:317-320) and
// _split_node + jt_most_dissimilar_packed (bitbirch.py:162-211, similarity.cpp:413-471).
//
// Data layout in HBM (flat pools indexed by node id, grown by the host):
//   per node  : header {len, leaf, prev_leaf, next_leaf};
//   per row   : packed centroid (RB bytes = row bytes padded to 16), its popcount, a
//               "link" word (internal row: child node id; leaf row: cluster-feature slot)
//               and a 32-byte RowMeta {BitFeature id, n_samples, CF slot, sum(ls), sum(ls^2)}.
//               Everything an insertion needs about a row lives next to the row, so one
//               level of the descent is ONE round of dependent memory accesses.
//   CF pools  : cf8 / cf16 / cf32 = linear sums at 1/2/4 bytes per feature (the reference
//               keeps the minimum dtype for n_samples, utils.py:25).  Leaf BitFeatures
//               move up a tier when n_samples crosses 255 / 65535; tracking BitFeatures of
//               internal nodes always live in cf32.
//
// Execution model: one 256-thread workgroup per tree inserts a whole batch with the
// reference's sequential semantics.  Inside an insert the work is data parallel.  Nodes read
// from HBM: 16 lanes x 16 B cover a 256-byte centroid row (16 rows per pass, DPP row reduction
// of the AND-popcounts).  The nodes of the most recent root-to-leaf path are mirrored in LDS
// (as many levels as fit in the CU's 160 KiB) and compared one row per lane, a quarter of the
// row per wave (node_best_mirror).  The cluster-feature update is one thread per 8 features,
// and the leaf test + every ancestor's CF update share one block reduction.  Everything
// wave-uniform (node ids, rows, lengths, slots) is kept in SGPRs (readfirstlane / readlane),
// every pool access is an explicit global-address-space access and every scratchpad access an
// LDS access, so the compiler emits scalar control flow, global_load/ds_read and no
// private-memory traffic.  The device code is templated on a context type: KC carries the
// tree's shape as run-time values, KCFix<BF, NF> as compile-time constants (instantiated for
// the benchmark shape 50 x 2048 bits and the CLI default 254 x 2048).
// All decisions use exact integers; the only floating point is the reference's own f64
// formulae in the same operation order.
//
// Also here: the exact batch mode for one tree (k_route / k_upd + bb_tree_batch.inc, DESIGN.md
// section 6b), many independent trees per launch (one or two workgroups per CU), the streaming
// ingest of host / file-backed rows (HostSlabs) and the leaf export kernels.
#include "bb_common.h"

#include <atomic>
#include <chrono>

#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <cmath>
#include <future>
#include <type_traits>

using namespace bbd;

namespace {

constexpr int TB = 256;  // threads per tree workgroup
constexpr int TW = TB / 64;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int MAXD = 256;    // deepest tree handled (the reference recurses: ~990 levels at most; trees this deep only
                             // arise from degenerate branching factors 2-3)
constexpr int MAXFAST = 5;   // ancestor levels updated in the fused fast path
constexpr int NRED = 2 + MAXFAST;
constexpr int MAX_BF = 1023;
constexpr int NPLW = 9;   // bit planes of a per-wave row count (<= 256 rows per wave at MAX_BF)
constexpr int SPLIT_MLP = 16;  // uint8 member CF rows requested before the first is consumed (split)
constexpr int SPLIT_MLP_WIDE = 4;  // same for rows of mixed width (32 B per thread and row)
constexpr int NPLT = 11;  // bit planes of a whole-node row count (<= 1024)
constexpr int MAXM = 4;  // levels of the current root-to-leaf path mirrored in LDS
// Node storage (round 5): the row pools are cut into BLOCKS of NG rows.  A node id is the index of the node's first block
// (its rows start at row id * NG of every pool, its header is entry id of the header pool) and a node owns as many blocks
// as its CAPACITY needs: bf + 1 rows while it is being inserted into, its length rounded up to a block once the compaction
// (gc_compact, host side) has found it unchanged since the previous compaction ("sealed").  The reference's node is a Python
// list that grows (bitbirch.py:264-287); a fixed (bf + 1)-row reservation cost 75.5 KB per node at bf 254 whatever it
// held - 3.2-4.8 GB per million S-ecfp fingerprints, whose leaves stay a tenth full.  Readers may still request bf + 1
// rows from any node (rows >= len are masked; the pools end in a pad of bf + 1 rows); whoever is about to WRITE to a node
// meets its header first, and a sealed node is moved to a fresh full-capacity block before anything else happens to it
// (thaw_node: complete engine; the steady-state and pipelined kernels hand the element over).
constexpr uint32_t NG = 4;
__host__ __device__ constexpr uint32_t node_blocks(uint32_t rows) { return (rows + NG - 1) / NG; }
// header word 1 ("leaf word"): bit 0 leaf flag; bits 4..15 the node's length at the last compaction + 1 (0: it has not seen
// one); bits 16..31 capacity in rows (0: this block does not start a live node)
constexpr uint32_t HW_LEAF = 1u;
__host__ __device__ constexpr uint32_t hw_make(uint32_t leaf, uint32_t cap_rows) { return (leaf & 1u) | (cap_rows << 16); }
__host__ __device__ constexpr uint32_t hw_cap(uint32_t w) { return w >> 16; }
__host__ __device__ constexpr uint32_t hw_gcl(uint32_t w) { return (w >> 4) & 0xFFFu; }

enum StopReason : int32_t {
    STOP_DONE = 0,
    STOP_NODES = 1,
    STOP_CF8 = 3,
    STOP_CF16 = 4,
    STOP_CF32 = 5,
    STOP_DEPTH = 6,
    STOP_RANGE = 7,  // n_samples would exceed 2^32-1
    STOP_GATE = 8,   // a gate node would have to split inside a concurrent batch (admission bug), or a sealed node was met there
};

enum Ctr : int { C_NODES = 0, C_IDS, C_N8, C_N16, C_N32, C_ROOT, C_FIRST_LEAF, C_DEPTH, C_COUNT };

struct __attribute__((aligned(16))) RowMeta {
    uint32_t sub;   // BitFeature id (leaf rows; NONE for tracking rows)
    uint32_t n;     // n_samples
    uint32_t slot;  // tier << 30 | index into cf8/cf16/cf32
    uint32_t pad;   // tracking rows: flip distance + 0 = unknown (see k_fd), else unused
    unsigned long long s1;  // sum(ls)      (leaf rows)
    unsigned long long s2;  // sum(ls^2)    (leaf rows)
};
static_assert(sizeof(RowMeta) == 32, "RowMeta must be 32 bytes");

struct __attribute__((aligned(16))) NodeHdr {
    uint32_t len, leaf, prev, next;
};

struct TreeDev {
    // configuration
    int32_t bf, F, nbytes, RB;
    int32_t crit, tol_len;
    double thr, tolerance;
    const double* tol_table;
    // node pool
    uint8_t* node_cent;
    uint32_t* node_card;
    uint32_t* node_link;
    RowMeta* node_rm;
    NodeHdr* node_hdr;
    uint8_t* scratch_cent;  // (bf+1) x RB, staging for row moves in a split
    // cluster-feature pools
    uint8_t* cf8;
    uint16_t* cf16;
    uint32_t* cf32;
    uint32_t cap_nodes, cap8, cap16, cap32;
    // state
    uint32_t ctr[C_COUNT];
    unsigned long long stats[8];
    unsigned long long phase[16];  // shader-clock cycles per phase (thread 0), debug; [8..] descent detail
    unsigned long long sphase[8]; // same, inside split_node
    unsigned long long splitprof[10];  // pipelined kernel (phase-timer build): the leaf split's phases (split_node's SPH marks), splits, cycles
    unsigned long long mlprof[8];  // multi-level router (phase-timer build): tracking-CF cache misses, levels committed
    unsigned long long rprof[4];   // router (phase-timer build): cycles waiting for a ring entry / wave 2's stamp / a leaf's pending jobs / a full leaf's decision
    // job of the next launch
    const uint8_t* rows;
    long long row_stride;
    const uint8_t* bufs;
    int32_t width;
    int32_t use_root_cache;
    long long n_elems;
    uint32_t* out_leaf;
    // result of the last launch
    long long processed;
    int32_t stop_reason;
    int32_t audit;  // (phase-timer build of the pipelined kernel, BBHIP_PIPE_AUDIT=1) compare the LDS state with HBM at every run end
    uint32_t giveup_line;  // STOP_INTERNAL: the line of bb_tree_pipe.inc where a bounded wait gave up
};

// LDS layout: byte offsets from the start of the dynamic shared segment
struct Smem {
    uint32_t x, vec, cA, cB;      // RB-byte vectors
    uint32_t keys;                // 2 x rows u64
    uint32_t link;                // 2 x rows u32
    uint32_t i1, u1, i2, u2;      // rows u32 each (split)
    uint32_t dst, mcard, mlink;   // rows u32 each (split)
    uint32_t mrm;                 // rows x 32 B RowMeta images (split)
    uint32_t red;                 // 2 x TW x NRED u64
    uint32_t red32;               // 2 x TW x NRED u32
    uint32_t wbest;               // 2 x TW u64: per-wave best candidate of a node compare
    uint32_t path_node, path_row, path_len, path_slot, path_n;  // MAXD u32 each
    uint32_t ctr;                 // C_COUNT u32
    uint32_t stats;               // 8 u64
    uint32_t bc;                  // 16 u32 broadcast scratch
    uint32_t planes;              // TW x NPLW x 64 u32 bit-sliced column counters (split)
    uint32_t ppart;               // 2 x TW x rows u32: per-wave partial intersections of a mirror compare
    uint32_t rc_cent, rc_card, rc_link;  // LDS mirrors of the nodes on the current path (MAXM levels)
    uint32_t total;
};

struct SmemCursor {
    uint32_t off = 0;
    constexpr uint32_t take(size_t bytes) {
        const uint32_t o = off;
        off += (uint32_t)((bytes + 15) / 16 * 16);
        return o;
    }
};

__host__ __device__ constexpr Smem smem_layout(int bf, int RB, int nm) {
    Smem s{};
    SmemCursor c;
    const size_t m = (size_t)bf + 1;
    s.x = c.take(RB); s.vec = c.take(RB); s.cA = c.take(RB); s.cB = c.take(RB);
    s.keys = c.take(2 * m * 8); s.link = c.take(2 * m * 4);
    s.i1 = c.take(m * 4); s.u1 = c.take(m * 4); s.i2 = c.take(m * 4); s.u2 = c.take(m * 4);
    s.dst = c.take(m * 4); s.mcard = c.take(m * 4); s.mlink = c.take(m * 4); s.mrm = c.take(m * 32);
    s.red = c.take(2 * TW * NRED * 8);
    s.red32 = c.take(2 * TW * NRED * 4);
    s.wbest = c.take(2 * TW * 8);
    s.path_node = c.take(MAXD * 4); s.path_row = c.take(MAXD * 4); s.path_len = c.take(MAXD * 4);
    s.path_slot = c.take(MAXD * 4); s.path_n = c.take(MAXD * 4);
    s.ctr = c.take(C_COUNT * 4); s.stats = c.take(8 * 8); s.bc = c.take(16 * 4);
    s.planes = c.take((size_t)TW * NPLW * 64 * 4);
    s.ppart = c.take(2 * (size_t)TW * m * 4);
    if (nm > 0) {
        s.rc_cent = c.take((size_t)nm * m * ((size_t)RB + 16));
        s.rc_card = c.take((size_t)nm * m * 4);
        s.rc_link = c.take((size_t)nm * m * 4);
    }
    s.total = c.off;
    return s;
}

// mirrored path levels for a (branching factor, row bytes) pair: as many (<= MAXM) as keep the
// workgroup inside the CU's 160 KiB of LDS; rows of 512, 1024 and 2048 bits are mirrored
__host__ __device__ constexpr int mirror_levels(int bf, int RB) {
    // a wave compares a quarter of the row in 16-byte pieces; the write-through of centroid updates
    // is part of the one-byte-group-per-thread update path (row bytes <= threads)
    if (RB % 64 != 0 || RB > TB) return 0;
    for (int q = MAXM; q >= 1; --q)
        if (smem_layout(bf, RB, q).total <= 160 * 1024) return q;
    return 0;
}

#if defined(__HIPCC__)

#define GA __attribute__((address_space(1)))
#define LA __attribute__((address_space(3)))
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;

// explicit global / LDS accesses (the pools are reached through pointers loaded from
// memory, which the compiler would otherwise treat as generic -> flat_load)
template <typename T> __device__ __forceinline__ T ldg(const void* p) { return *(const GA T*)p; }
template <typename T> __device__ __forceinline__ void stg(void* p, T v) { *(GA T*)p = v; }
template <typename T> __device__ __forceinline__ LA T* lds(LA unsigned char* L, uint32_t off) { return (LA T*)(L + off); }

__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uni64(u64 v) { return ((u64)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
__device__ __forceinline__ uint32_t rdlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }

__device__ __forceinline__ uint32_t popc4v(u32x4_t q) { return __popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w); }

// wave reductions with a wave-uniform (SGPR) result: DPP inside the 16-lane rows, readlane across
__device__ __forceinline__ uint32_t wsum32(uint32_t v) {
    v = row16_sum(v);
    return rdlane(v, 0) + rdlane(v, 16) + rdlane(v, 32) + rdlane(v, 48);
}
template <int N> __device__ __forceinline__ u64 ror64(u64 v) {
    return ((u64)row_ror<N>((uint32_t)(v >> 32)) << 32) | row_ror<N>((uint32_t)v);
}
__device__ __forceinline__ u64 rdlane64(u64 v, int l) { return ((u64)rdlane((uint32_t)(v >> 32), l) << 32) | rdlane((uint32_t)v, l); }
__device__ __forceinline__ u64 wsum64(u64 v) {
    v += ror64<8>(v);
    v += ror64<4>(v);
    v += ror64<2>(v);
    v += ror64<1>(v);
    return rdlane64(v, 0) + rdlane64(v, 16) + rdlane64(v, 32) + rdlane64(v, 48);
}
__device__ __forceinline__ u64 wmax64(u64 v) {
    u64 o;
    o = ror64<8>(v); v = o > v ? o : v;
    o = ror64<4>(v); v = o > v ? o : v;
    o = ror64<2>(v); v = o > v ? o : v;
    o = ror64<1>(v); v = o > v ? o : v;
    const u64 a = rdlane64(v, 0), b = rdlane64(v, 16), c = rdlane64(v, 32), d = rdlane64(v, 48);
    const u64 ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// uniform kernel context: plain values, fully scalarised after inlining.  Two flavours with the
// same member names: KC carries the tree's shape (branching factor, row width, LDS layout) as
// run-time values; KCFix<BF, NF> has them as compile-time constants, which frees ~45 scalar
// registers, turns every LDS offset into an instruction immediate and folds the shape-dependent
// branches and loops.  The device code is templated on the context type.
struct KCBase {
    uint8_t* cent; uint32_t* card; uint32_t* link; RowMeta* rm; NodeHdr* hdr; uint8_t* scratch;
    uint8_t* cf8; uint16_t* cf16; uint32_t* cf32;
    const uint8_t* bufs; int width;
    int crit, tol_len; double thr, tolerance; const double* tol;
    LA unsigned char* L;
    // kind of element a kernel instance inserts: 0 packed fingerprints only, 1 BitFeature buffers
    // only, 2 decided at run time (`bufs` null or not).  The specialised instances drop the other
    // path's code and registers (+13 % on fingerprints).
    static constexpr int buf = 2;
    // merge criterion fixed at compile time (BBH_CRIT_*), or -1: read `crit` at run time
    static constexpr int crit_fixed = -1;
};
template <class Base, int BUF, int CRIT = -1>
struct KCWith : Base {
    static constexpr int buf = BUF;
    static constexpr int crit_fixed = CRIT;
};
struct KC : KCBase {
    static constexpr bool dynamic_shape = true;
    int F, nb, RB, RBc, RBS;
    uint32_t rows, bf, nblk;  // (nblk: blocks of a full-capacity node)
    int nm;  // mirrored levels
    Smem o;
};
template <int BF, int NF>
struct KCFix : KCBase {
    static constexpr bool dynamic_shape = false;
    static constexpr int F = NF, nb = NF / 8, RB = (NF / 8 + 15) / 16 * 16, RBc = RB / 16, RBS = RB + 16;
    static constexpr uint32_t rows = BF + 1, bf = BF, nblk = node_blocks(BF + 1);
    static constexpr int nm = mirror_levels(BF, RB);
    static constexpr Smem o = smem_layout(BF, RB, nm);
};
struct Elem {  // the element being inserted (uniform)
    long long idx; uint32_t nS; u64 s1S, s2S; uint32_t pcx;
};

// ---- block reduction of NV u64 values to every thread, uniform result (one barrier) ------
template <int NV, class KCt>
__device__ __forceinline__ void block_sum(const KCt& k, int& red_slot, u64 (&v)[NV]) {
    LA u64* buf = lds<u64>(k.L, k.o.red) + red_slot * TW * NRED;
    red_slot ^= 1;
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const u64 t = wsum64(v[i]);
        if ((threadIdx.x & 63) == 0) buf[w * NRED + i] = t;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        u64 a = 0;
#pragma unroll
        for (int q = 0; q < TW; ++q) a += buf[q * NRED + i];
        v[i] = uni64(a);
    }
}

// The hot-path reduction: one dot product (64-bit only when it can exceed 32 bits) and up to
// 1 + MAXFAST popcounts (always < 2^14), one barrier.  Results are wave-uniform.
template <class KCt>
__device__ __forceinline__ void block_sum_fused(const KCt& k, int& red_slot, bool wide_dot, u64& dot,
                                                uint32_t (&pc)[1 + MAXFAST], int npc) {
    LA u64* b64 = lds<u64>(k.L, k.o.red) + red_slot * TW * NRED;
    LA uint32_t* b32 = lds<uint32_t>(k.L, k.o.red32) + red_slot * TW * NRED;
    red_slot ^= 1;
    const int w = threadIdx.x >> 6;
    const bool lane0 = (threadIdx.x & 63) == 0;
    const u64 d = wide_dot ? wsum64(dot) : (u64)wsum32((uint32_t)dot);
    if (lane0) b64[w * NRED] = d;
#pragma unroll
    for (int i = 0; i < 1 + MAXFAST; ++i) {
        if (i < npc) {
            const uint32_t t = wsum32(pc[i]);
            if (lane0) b32[w * NRED + i] = t;
        }
    }
    __syncthreads();
    u64 a = 0;
#pragma unroll
    for (int q = 0; q < TW; ++q) a += b64[q * NRED];
    dot = uni64(a);
#pragma unroll
    for (int i = 0; i < 1 + MAXFAST; ++i) {
        if (i < npc) {
            uint32_t t = 0;
#pragma unroll
            for (int q = 0; q < TW; ++q) t += b32[q * NRED + i];
            pc[i] = uni(t);
        }
    }
}

// ---- cluster-feature access: 8 consecutive features -----------------------------------
// A leaf BitFeature of ONE fingerprint (n_samples == 1) has its linear sum in its centroid row (one fingerprint is its own
// majority vote, bitbirch.py:423-435), which its node holds anyway: its uint8 slot is reserved when it is appended but a
// packed fingerprint's is not written, and no reader looks at it.  The first merge writes the sum of the two members
// there.  Readers are handed an EFFECTIVE slot word with tier TIER_LAZY (never stored) and the centroid row (`crow`).
constexpr uint32_t TIER_LAZY = 3u;
// ... and since round 4 such a BitFeature has no uint8 slot at all: its slot word is SLOT_LAZY8 (slot 0 of the uint8 pool,
// which no BitFeature ever owns: the counter starts at 1, so a speculative load through the word stays inside the pool);
// the slot is allocated by the merge that gives it a second member.  A tree of N fingerprints that hardly merge (S-ecfp:
// 99 % singletons) used to reserve 2 KB x N of HBM that nothing ever wrote.
constexpr uint32_t SLOT_LAZY8 = 0u;
__device__ __forceinline__ uint32_t slot_effective(uint32_t slotw, u64 n) { return n == 1 ? ((TIER_LAZY << 30) | (slotw & 0x3FFFFFFFu)) : slotw; }
__device__ __forceinline__ void byte_to_cols(uint32_t byte, uint32_t v[8]) {
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = (byte >> (7 - q)) & 1u;
}

template <class KCt>
__device__ __forceinline__ void cf_load8(const KCt& k, uint32_t slotw, int b, uint32_t v[8], const uint8_t* crow = nullptr) {
    const uint32_t tier = slotw >> 30;
    const size_t idx = (size_t)(slotw & 0x3FFFFFFFu) * (size_t)k.F + (size_t)b * 8;
    if (tier == TIER_LAZY) {
        byte_to_cols(ldg<uint8_t>(crow + b), v);
    } else if (tier == 0) {
        const u32x2_t q = ldg<u32x2_t>(k.cf8 + idx);
        v[0] = q.x & 0xFF; v[1] = (q.x >> 8) & 0xFF; v[2] = (q.x >> 16) & 0xFF; v[3] = q.x >> 24;
        v[4] = q.y & 0xFF; v[5] = (q.y >> 8) & 0xFF; v[6] = (q.y >> 16) & 0xFF; v[7] = q.y >> 24;
    } else if (tier == 1) {
        const u32x4_t q = ldg<u32x4_t>(k.cf16 + idx);
        v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
        v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
    } else {
        const u32x4_t q0 = ldg<u32x4_t>(k.cf32 + idx), q1 = ldg<u32x4_t>(k.cf32 + idx + 4);
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
        v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    }
}

// The same load split in two so that several rows can be in flight before the first unpack.
// The request itself is branch-free (a load inside a tier branch makes the compiler wait for it at
// the branch's join): always two 16-byte loads, the second one only meaningful for cf32 and
// pointed at the same line otherwise.  cf8 rows are 8-byte aligned and over-read by 8 bytes
// (the pools carry 64 bytes of slack; amdhsa runs with unaligned access enabled).
template <class KCt>
__device__ __forceinline__ void cf_load_raw(const KCt& k, uint32_t slotw, int b, u32x4_t (&raw)[2], const uint8_t* crow = nullptr) {
    const uint32_t tier = slotw >> 30;
    if (tier == TIER_LAZY) {  // (wave-uniform) byte b of the BitFeature's centroid row
        raw[0] = (u32x4_t)(0);
        raw[1] = (u32x4_t)(0);
        raw[0].x = ldg<uint8_t>(crow + b);
        return;
    }
    // mask arithmetic, not ?: - the compiler turns a three-way pointer select into a table in scratch
    const u64 m0 = 0ull - (u64)(tier == 0), m1 = 0ull - (u64)(tier == 1), m2 = 0ull - (u64)(tier >= 2);
    const uint8_t* base = (const uint8_t*)(((u64)(uintptr_t)k.cf8 & m0) | ((u64)(uintptr_t)k.cf16 & m1) | ((u64)(uintptr_t)k.cf32 & m2));
    const size_t off = ((size_t)(slotw & 0x3FFFFFFFu) * (size_t)k.F + (size_t)b * 8) << tier;
    raw[0] = ldg<u32x4_t>(base + off);
    raw[1] = ldg<u32x4_t>(base + off + (tier == 2 ? 16 : 0));
}
__device__ __forceinline__ void cf_unpack_raw(uint32_t tier, const u32x4_t (&raw)[2], uint32_t v[8]) {
    const u32x4_t q = raw[0];
    if (tier == TIER_LAZY) {
        byte_to_cols(q.x & 0xFFu, v);
    } else if (tier == 0) {
        v[0] = q.x & 0xFF; v[1] = (q.x >> 8) & 0xFF; v[2] = (q.x >> 16) & 0xFF; v[3] = q.x >> 24;
        v[4] = q.y & 0xFF; v[5] = (q.y >> 8) & 0xFF; v[6] = (q.y >> 16) & 0xFF; v[7] = q.y >> 24;
    } else if (tier == 1) {
        v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
        v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
    } else {
        const u32x4_t q1 = raw[1];
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    }
}

template <class KCt>
__device__ __forceinline__ void cf_store8(const KCt& k, uint32_t slotw, int b, const uint32_t v[8]) {
    const uint32_t tier = slotw >> 30;
    const size_t idx = (size_t)(slotw & 0x3FFFFFFFu) * (size_t)k.F + (size_t)b * 8;
    if (tier == 0) {
        u32x2_t q;
        q.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        q.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
        stg<u32x2_t>(k.cf8 + idx, q);
    } else if (tier == 1) {
        u32x4_t q;
        q.x = v[0] | (v[1] << 16); q.y = v[2] | (v[3] << 16);
        q.z = v[4] | (v[5] << 16); q.w = v[6] | (v[7] << 16);
        stg<u32x4_t>(k.cf16 + idx, q);
    } else {
        u32x4_t q0, q1;
        q0.x = v[0]; q0.y = v[1]; q0.z = v[2]; q0.w = v[3];
        q1.x = v[4]; q1.y = v[5]; q1.z = v[6]; q1.w = v[7];
        stg<u32x4_t>(k.cf32 + idx, q0);
        stg<u32x4_t>(k.cf32 + idx + 4, q1);
    }
}

// a BitFeature known to be in the uint8 tier (a freshly appended one): no tier dispatch
template <class KCt>
__device__ __forceinline__ void cf_store8_u8(const KCt& k, uint32_t slot, int b, const uint32_t v[8]) {
    const size_t idx = (size_t)slot * (size_t)k.F + (size_t)b * 8;
    u32x2_t q;
    q.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
    q.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
    stg<u32x2_t>(k.cf8 + idx, q);
}

// tracking BitFeatures always live in cf32: no tier dispatch on the hot path
template <class KCt>
__device__ __forceinline__ void cf32_load8(const KCt& k, uint32_t slotw, int b, uint32_t v[8]) {
    const size_t idx = (size_t)(slotw & 0x3FFFFFFFu) * (size_t)k.F + (size_t)b * 8;
    const u32x4_t q0 = ldg<u32x4_t>(k.cf32 + idx), q1 = ldg<u32x4_t>(k.cf32 + idx + 4);
    v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
    v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
}

// linear-sum values of the element being inserted for features b*8 .. b*8+7
template <class KCt>
__device__ __forceinline__ void elem_cols(const KCt& k, const Elem& el, int b, uint32_t v[8]) {
    if (KCt::buf == 0 || (KCt::buf == 2 && k.bufs == nullptr)) {  // fingerprint: bits of the packed row, MSB first
        const uint32_t xb = lds<uint8_t>(k.L, k.o.x)[b];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = (xb >> (7 - q)) & 1u;
        return;
    }
    // BitFeature buffer row: F + 1 values of `width` bytes, so rows are not aligned to anything;
    // gfx950 under HSA serves unaligned vector loads, one request per thread instead of eight
    const size_t base = (size_t)el.idx * ((size_t)k.F + 1) + (size_t)b * 8;
    if (k.width == 1) {
        const u32x2_t q = ldg<u32x2_t>(k.bufs + base);
        v[0] = q.x & 0xFF; v[1] = (q.x >> 8) & 0xFF; v[2] = (q.x >> 16) & 0xFF; v[3] = q.x >> 24;
        v[4] = q.y & 0xFF; v[5] = (q.y >> 8) & 0xFF; v[6] = (q.y >> 16) & 0xFF; v[7] = q.y >> 24;
    } else if (k.width == 2) {
        const u32x4_t q = ldg<u32x4_t>(k.bufs + 2 * base);
        v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
        v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
    } else if (k.width == 4) {
        const u32x4_t q0 = ldg<u32x4_t>(k.bufs + 4 * base), q1 = ldg<u32x4_t>(k.bufs + 4 * base + 16);
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
        v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    } else {
#pragma unroll
        for (int q = 0; q < 8; q += 2) {  // uint64 tables: values fit 32 bits by contract (n_samples < 2^32)
            const u32x4_t w = ldg<u32x4_t>(k.bufs + 8 * (base + q));
            v[q] = w.x;
            v[q + 1] = w.z;
        }
    }
}

// majority-vote byte for 8 features (centroid_from_sum, _py_similarity.py:36-41).
// n fits in 32 bits by contract, 2*v is formed in 64 bits only when needed.
__device__ __forceinline__ uint32_t centroid_byte(const uint32_t v[8], u64 n) {
    uint32_t byte = 0;
    if (n <= 1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) byte |= ((v[q] & 0xFFu) != 0 ? 1u : 0u) << (7 - q);
    } else if (n <= 0x7FFFFFFFull) {
        const uint32_t half = (uint32_t)((n + 1) >> 1);  // 2v >= n  <=>  v >= ceil(n/2)
#pragma unroll
        for (int q = 0; q < 8; ++q) byte |= (v[q] >= half ? 1u : 0u) << (7 - q);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) byte |= (2ull * v[q] >= n ? 1u : 0u) << (7 - q);
    }
    return byte;
}

// the same for a BitFeature that has just absorbed another one: n >= 2 by construction, so the
// "n <= 1: cast" case of centroid_from_sum cannot occur (one code variant less per inlined copy)
__device__ __forceinline__ uint32_t centroid_byte_merged(const uint32_t v[8], u64 n) {
    uint32_t byte = 0;
    if (n <= 0x7FFFFFFFull) {
        // v >= ceil(n/2)  <=>  (ceil(n/2) - 1) - v is negative as int32 (v <= n < 2^31): its sign bit is the centroid
        // bit, shifted in from the right - two instructions per feature (v_sub, v_alignbit) instead of compare,
        // select and or
        const uint32_t hm1 = (uint32_t)((n + 1) >> 1) - 1u;
#pragma unroll
        for (int q = 0; q < 8; ++q) byte = __builtin_amdgcn_alignbit(byte, hm1 - v[q], 31);
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) byte |= (2ull * v[q] >= n ? 1u : 0u) << (7 - q);
    }
    return byte;
}

// the same for n < 2^31 only (the steady-state kernel leaves larger clusters to the complete engine): no 64-bit variant,
// i.e. no branch on a value the compiler has to treat as divergent
__device__ __forceinline__ uint32_t centroid_byte_merged31(const uint32_t v[8], uint32_t n) {
    const uint32_t hm1 = ((n + 1u) >> 1) - 1u;
    uint32_t byte = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) byte = __builtin_amdgcn_alignbit(byte, hm1 - v[q], 31);
    return byte;
}

__device__ __forceinline__ uint32_t tier_for(u64 n) { return n <= 255 ? 0u : (n <= 65535 ? 1u : 2u); }
__device__ __forceinline__ int ctr_for_tier(uint32_t tier) { return tier == 0 ? C_N8 : (tier == 1 ? C_N16 : C_N32); }

// popcount of an RB-byte vector in LDS (every wave computes it redundantly; uniform result)
template <class KCt>
__device__ __forceinline__ uint32_t lds_vec_popcount(const KCt& k, uint32_t off) {
    uint32_t p = 0;
    for (int ch = threadIdx.x & 63; ch < k.RBc; ch += 64) p += popc4v(lds<u32x4_t>(k.L, off)[ch]);
    return wsum32(p);
}

// A comparison candidate: Tanimoto = i / u as an exact fraction (u already clamped to >= 1,
// similarity.cpp:326-331) and the row it belongs to.  Equal fractions <=> equal float64
// quotients (both are the correctly rounded value of the same rational), distinct fractions
// with u <= 2^14 differ by far more than an ulp, so ordering fractions by cross-multiplication
// reproduces np.argmax / np.argmin on the reference's float64 array exactly, including the
// first-index tie-break.
struct Cand { uint32_t i, u, r; };

template <bool MINMODE>
__device__ __forceinline__ bool cand_better(uint32_t ai, uint32_t au, uint32_t ar, uint32_t bi, uint32_t bu, uint32_t br) {
    const uint32_t x = __umul24(ai, bu), y = __umul24(bi, au);  // operands < 2^14 (n_features <= 16384): products < 2^28, full-rate multiplies
    if (MINMODE) return x < y || (x == y && ar < br);
    return x > y || (x == y && ar < br);
}

// ---- similarity of every row of a node against a vector in LDS; returns the first-argmax
// (or first-argmin) row as a wave-uniform candidate.  Loads all bf+1 row slots without
// waiting for the node's length (rows >= len are masked) so header and rows arrive in the
// same memory round trip; popcounts of 16 lanes meet by DPP, the 4 row groups of a wave by
// readlane, the 4 waves through 32 bytes of LDS and one barrier.
template <bool ROOT, bool MINMODE, class KCt>
__device__ __forceinline__ Cand node_best(const KCt& k, int& cmp_par, uint32_t nd, int known_len, uint32_t vec_off,
                                          uint32_t vec_pc, bool want_counts, bool second, bool want_link,
                                          uint32_t* out_len, uint32_t* out_leaf, uint32_t mslot = 0, bool fill = false,
                                          bool* out_any_card = nullptr) {
    // ROOT: read the node from LDS mirror `mslot`; !ROOT && fill: read HBM and refresh that mirror
    const uint32_t mrow0 = mslot * k.rows;
    const int tid = threadIdx.x, l = tid & 15, g = tid >> 4;
    const uint32_t rows = k.rows;
    cmp_par ^= 1;
    LA uint32_t* s_link = lds<uint32_t>(k.L, k.o.link) + cmp_par * rows;
    LA uint32_t* s_i = lds<uint32_t>(k.L, second ? k.o.i2 : k.o.i1);
    LA uint32_t* s_u = lds<uint32_t>(k.L, second ? k.o.u2 : k.o.u1);
    LA u64* wb = lds<u64>(k.L, k.o.wbest) + cmp_par * TW;
    const size_t meta = (size_t)nd * NG;
    uint32_t len = 0, leaf = 0;  // (leaf: the header's whole leaf word - flag, compaction record, capacity)
    u32x4_t hraw = (u32x4_t)(0);
    const bool load_hdr = !ROOT && known_len < 0;
    if (load_hdr) hraw = ldg<u32x4_t>(k.hdr + nd);  // issued before the rows, same round trip
    LA u32x4_t* vec = lds<u32x4_t>(k.L, vec_off);
    // running best of this lane's row group (identical in all 16 lanes of the group)
    uint32_t bi = MINMODE ? 0xFFFFu : 0u, bu = 1u, br = NONE;
    bool anyc = false;  // some row (< len) of this lane's groups has a non-zero popcount
    if (k.RBc == 16) {
        const u32x4_t xv = vec[l];
        const uint32_t last = rows - 1;
        uint32_t r_start = 0;
        if constexpr (!ROOT) {
            // Nodes far larger than the block's 16 rows per pass (branching factors of several hundred: nothing of them is
            // mirrored in LDS, every level streams from L2): two blocks of 8 passes = 256 rows requested before the first is
            // consumed, instead of one memory round trip per 64 rows.
            if (!fill && rows > 256) {
                // Blocks of 128 rows (8 passes of the block's 16 rows), double-buffered: the next block's rows are on their
                // way while this one's are counted - with one block at a time a 1001-row node was four exposed memory round
                // trips (41 k cycles per level at bf 1000).  Requests carry no branch (a conditional request makes the
                // compiler wait for everything outstanding at the join): they are BUFFER loads bounded by the node's live bytes
                // as soon as the length is known - lanes beyond the end are out of range, return zero and ask the memory system
                // for nothing (tools/probe/ta_request_cost.cpp) - and blocks are 128 rows, not 256, because what is requested
                // blind (block B, before the header has arrived) and ahead of the end queues in front of the next dependent
                // load: timed inside this function, a level of S-fake's bf 1000 tree spent 2.2 k cycles issuing two 256-row
                // blocks for nodes of 124-314 live rows (profiles/r05/bf1000_node_best_timeline.txt).  A row's popcount is
                // counted from the row itself (same reduction, high half) instead of loaded, links wait in registers until
                // their block is consumed.
                constexpr int NP = 8;
                constexpr uint32_t BLK = NP * 16;
                uint32_t nrec = rows * 256u;  // bytes of the node's rows that exist, as far as known: the bound of the buffer loads
                // The 16 partial counts of a row sit in the 16 lanes of its group, and a block hands every group 8 rows: the
                // 16 x 8 partials are summed by a TRANSPOSING butterfly inside each half-row of 8 lanes (select, select,
                // v_add_u32_dpp with row_half_mirror, quad_perm [2,3,0,1], [1,0,3,2]: 7 steps, lane l ends with its half-row's
                // sum for pass l & 7) and one row_ror:8 add brings the two half-rows together, instead of 8 x 4 DPP adds; and
                // everything behind the sum - union, counts for the split, the running first-argmax - runs once per block on 8
                // different rows per group (lanes l and l ^ 8 the same one) instead of eight times on one.  The row a lane looks
                // after within a block is rs + (l & 7) * 16 + g.
                u32x4_t dA[NP], dB[NP];
                uint32_t lkA = 0, lkB = 0;
                auto issue = [&](u32x4_t (&d)[NP], uint32_t& lk, uint32_t rs) {
                    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(k.cent + meta * 256), 0, (int)nrec, 0x00020000);
#pragma unroll
                    for (int p = 0; p < NP; ++p) {
                        const uint32_t r = rs + p * 16 + g;
                        d[p] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(r * 256 + l * 16), 0, 0));
                    }
                    if (want_link) {
                        const uint32_t r = rs + (l & 7) * 16 + g;
                        lk = ldg<uint32_t>(k.link + meta + (r < last ? r : last));
                    }
                };
                const bool b2 = (l & 4) != 0, b1 = (l & 2) != 0, b0 = (l & 1) != 0;
                auto consume = [&](const u32x4_t (&d)[NP], uint32_t lk, uint32_t rs) {
                    if (rs == 0) {
                        if (load_hdr) {
                            len = uni(hraw.x);
                            leaf = uni(hraw.y);
                        } else {
                            len = (uint32_t)known_len;
                        }
                    }
                    // (a row's popcount is counted from the row itself, in the high half)
                    uint32_t v4[4], v2[2];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint32_t lo = popc4v(d[j] & xv) + (popc4v(d[j]) << 16);
                        const uint32_t hi = popc4v(d[j + 4] & xv) + (popc4v(d[j + 4]) << 16);
                        v4[j] = (b2 ? hi : lo) + dpp_ctrl<0x141>(b2 ? lo : hi);
                    }
#pragma unroll
                    for (int j = 0; j < 2; ++j) v2[j] = (b1 ? v4[j + 2] : v4[j]) + dpp_ctrl<0x4E>(b1 ? v4[j] : v4[j + 2]);
                    uint32_t both = (b0 ? v2[1] : v2[0]) + dpp_ctrl<0xB1>(b0 ? v2[0] : v2[1]);
                    both += row_ror<8>(both);  // the other half-row's eight lanes: the same pass (l & 7)
                    const uint32_t r = rs + (uint32_t)(l & 7) * 16 + g;
                    if (want_link && r < rows) s_link[r] = lk;
                    const uint32_t inter = both & 0xFFFFu;
                    uint32_t un = (both >> 16) + vec_pc - inter;
                    anyc = anyc || (r < len && (both >> 16) != 0);
                    if (want_counts && r < len) { s_i[r] = inter; s_u[r] = un; }
                    un = un < 1u ? 1u : un;
                    const bool take = r < len && cand_better<MINMODE>(inter, un, r, bi, bu, br);
                    bi = take ? inter : bi;
                    bu = take ? un : bu;
                    br = take ? r : br;
                };
                // (only the node's `len` rows are walked, not all bf + 1: the length arrives with the first block - until round 4
                // a half-full node of a bf 1000 tree cost what a full one does)
                issue(dA, lkA, 0);
                uint32_t lim = rows;
                for (; r_start < lim; r_start += 2 * BLK) {
                    issue(dB, lkB, r_start + BLK);
                    consume(dA, lkA, r_start);
                    if (r_start == 0) { lim = len < rows ? len : rows; nrec = lim * 256u; }
                    issue(dA, lkA, r_start + 2 * BLK);
                    if (r_start + BLK < lim) consume(dB, lkB, r_start + BLK);
                }
                // the lanes of a group hold different rows' candidates here: rotate-and-keep-better leaves the group's first
                // best in all 16 of them, as the code below expects
#define BB_GSTEP(N)                                                                          \
    {                                                                                        \
        const uint32_t oi = row_ror<N>(bi), ou = row_ror<N>(bu), orr = row_ror<N>(br);       \
        const bool tk = cand_better<MINMODE>(oi, ou, orr, bi, bu, br);                       \
        bi = tk ? oi : bi;                                                                   \
        bu = tk ? ou : bu;                                                                   \
        br = tk ? orr : br;                                                                  \
    }
                BB_GSTEP(8) BB_GSTEP(4) BB_GSTEP(2) BB_GSTEP(1)
#undef BB_GSTEP
                r_start = rows;  // (every row below `len` has been looked at)
            }
        }
        for (uint32_t r0 = r_start; r0 < rows; r0 += 64) {
            // branch-free: row indices are clamped instead of predicated so that all loads of
            // the pass are issued back to back; out-of-range rows are discarded by `r < len`
            u32x4_t d[4];
            uint32_t cd[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t r = r0 + p * 16 + g;
                const uint32_t rc = r < last ? r : last;
                if constexpr (ROOT) {
                    d[p] = *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + rc) * k.RBS + l * 16);
                    cd[p] = lds<uint32_t>(k.L, k.o.rc_card)[mrow0 + rc];
                } else {
                    d[p] = ldg<u32x4_t>(k.cent + (meta + rc) * 256 + l * 16);
                    cd[p] = ldg<uint32_t>(k.card + meta + rc);
                }
            }
            if constexpr (!ROOT) {
                if (want_link) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const uint32_t r = r0 + p * 16 + g;
                        if (l == 0 && r < rows) {
                            const uint32_t lk = ldg<uint32_t>(k.link + meta + r);
                            s_link[r] = lk;
                            if (fill) lds<uint32_t>(k.L, k.o.rc_link)[mrow0 + r] = lk;
                        }
                    }
                }
                if (fill) {
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const uint32_t r = r0 + p * 16 + g;
                        if (r < rows) {
                            *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + r) * k.RBS + l * 16) = d[p];
                            if (l == 0) lds<uint32_t>(k.L, k.o.rc_card)[mrow0 + r] = cd[p];
                        }
                    }
                }
            }
            if (r0 == 0) {
                if (load_hdr) {
                    len = uni(hraw.x);
                    leaf = uni(hraw.y);
                } else {
                    len = (uint32_t)known_len;
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t r = r0 + p * 16 + g;
                // lane 0 of the group carries the row's cardinality in the high half
                const uint32_t both = row16_sum(popc4v(d[p] & xv) + (l == 0 ? cd[p] << 16 : 0u));
                const uint32_t inter = both & 0xFFFFu;
                uint32_t un = (both >> 16) + vec_pc - inter;
                anyc = anyc || (r < len && (both >> 16) != 0);
                if (want_counts && r < len && l == 0) { s_i[r] = inter; s_u[r] = un; }
                un = un < 1u ? 1u : un;
                const bool take = r < len && cand_better<MINMODE>(inter, un, r, bi, bu, br);
                bi = take ? inter : bi;
                bu = take ? un : bu;
                br = take ? r : br;
            }
        }
    } else {
        if (load_hdr) {
            len = uni(hraw.x);
            leaf = uni(hraw.y);
        } else {
            len = (uint32_t)known_len;
        }
        for (uint32_t r0 = 0; r0 < rows; r0 += TB / 16) {
            const uint32_t r = r0 + g;
            uint32_t part = 0;
            if (r < len) {
                for (int ch = l; ch < k.RBc; ch += 16) {
                    u32x4_t d;
                    if constexpr (ROOT) d = *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + r) * k.RBS + ch * 16);
                    else d = ldg<u32x4_t>(k.cent + (meta + r) * (size_t)k.RB + ch * 16);
                    if constexpr (!ROOT) {
                        if (fill) *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + r) * k.RBS + ch * 16) = d;
                    }
                    part += popc4v(d & vec[ch]);
                }
                if (l == 0) {
                    if constexpr (ROOT) {
                        part += lds<uint32_t>(k.L, k.o.rc_card)[mrow0 + r] << 16;
                    } else {
                        const uint32_t cd = ldg<uint32_t>(k.card + meta + r);
                        part += cd << 16;
                        uint32_t lk = 0;
                        if (want_link || fill) lk = ldg<uint32_t>(k.link + meta + r);
                        if (want_link) s_link[r] = lk;
                        if (fill) {
                            lds<uint32_t>(k.L, k.o.rc_card)[mrow0 + r] = cd;
                            lds<uint32_t>(k.L, k.o.rc_link)[mrow0 + r] = lk;
                        }
                    }
                }
            }
            const uint32_t both = row16_sum(part);
            const uint32_t inter = both & 0xFFFFu;
            uint32_t un = (both >> 16) + vec_pc - inter;
            anyc = anyc || (r < len && (both >> 16) != 0);
            if (want_counts && r < len && l == 0) { s_i[r] = inter; s_u[r] = un; }
            un = un < 1u ? 1u : un;
            if (r < len && cand_better<MINMODE>(inter, un, r, bi, bu, br)) { bi = inter; bu = un; br = r; }
        }
    }
    // the 4 row groups of this wave (scalar from here on)
    uint32_t wi = rdlane(bi, 0), wu = rdlane(bu, 0), wr = rdlane(br, 0);
#pragma unroll
    for (int q = 16; q < 64; q += 16) {
        const uint32_t ci = rdlane(bi, q), cu = rdlane(bu, q), cr = rdlane(br, q);
        if (cand_better<MINMODE>(ci, cu, cr, wi, wu, wr)) { wi = ci; wu = cu; wr = cr; }
    }
    const uint32_t wany = __ballot(anyc) != 0ull ? 0x80000000u : 0u;  // rides in the row word's top bit
    if ((tid & 63) == 0) wb[tid >> 6] = ((u64)(wr ^ wany) << 32) | (wi << 16) | wu;
    __syncthreads();
    Cand best;
    uint32_t any_all = 0;
    {
        const u64 v0 = wb[0];
        const uint32_t hi = uni((uint32_t)(v0 >> 32));
        // a wave without a valid row reports r = NONE (all ones): the flag bit is XORed in, so recover it
        const uint32_t r0v = (hi | 0x80000000u) == NONE ? NONE : (hi & 0x7FFFFFFFu);
        any_all |= r0v == NONE ? (~hi & 0x80000000u) : (hi & 0x80000000u);
        best.r = r0v;
        best.i = uni((uint32_t)v0) >> 16;
        best.u = uni((uint32_t)v0) & 0xFFFFu;
    }
#pragma unroll
    for (int w = 1; w < TW; ++w) {
        const u64 v = wb[w];
        const uint32_t hi = uni((uint32_t)(v >> 32));
        const uint32_t cr = (hi | 0x80000000u) == NONE ? NONE : (hi & 0x7FFFFFFFu);
        any_all |= cr == NONE ? (~hi & 0x80000000u) : (hi & 0x80000000u);
        const uint32_t lo = uni((uint32_t)v);
        const uint32_t ci = lo >> 16, cu = lo & 0xFFFFu;
        if (cand_better<MINMODE>(ci, cu, cr, best.i, best.u, best.r)) { best.i = ci; best.u = cu; best.r = cr; }
    }
    if (out_len) *out_len = len;
    if (out_leaf) *out_leaf = leaf;
    if (out_any_card) *out_any_card = any_all != 0;
    return best;
}

// ---- the same comparison for a node that sits in LDS mirror `mslot` (2048-bit rows) ---------
// One row per LANE and a quarter of the row (16 dwords) per WAVE: 4 LDS reads + 16 AND/BCNT per
// lane and no cross-lane reduction at all; the four partial intersections of a row meet through
// LDS (one barrier), then every wave redundantly forms the candidates and reduces them
// (DPP butterfly inside the 16-lane rows, readlane across them), so the result is uniform across
// the block without a second barrier.  ~4x fewer instructions than the 16-lanes-per-row layout
// that suits coalesced HBM reads.
template <bool MINMODE, class KCt>
__device__ __forceinline__ Cand node_best_mirror(const KCt& k, int& cmp_par, uint32_t len, uint32_t vec_off, uint32_t vec_pc,
                                                 bool want_counts, bool second, uint32_t mslot, bool* out_any_card) {
    const uint32_t mrow0 = mslot * k.rows;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    cmp_par ^= 1;
    LA uint32_t* pp = lds<uint32_t>(k.L, k.o.ppart) + (uint32_t)cmp_par * TW * k.rows;
    LA uint32_t* s_i = lds<uint32_t>(k.L, second ? k.o.i2 : k.o.i1);
    LA uint32_t* s_u = lds<uint32_t>(k.L, second ? k.o.u2 : k.o.u1);
    // this wave's quarter of the query vector (same address in every lane: LDS broadcast); rows of
    // 64 / 128 / 256 bytes give 1 / 2 / 4 sixteen-byte pieces per wave
    const int pieces = k.RB / 64;
    const uint32_t slice = (uint32_t)k.RB / 4;
    u32x4_t xv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < pieces) xv[i] = *(LA u32x4_t*)(k.L + vec_off + wave * slice + i * 16);
    const uint32_t last = len - 1;  // len >= 1
    // loops run over the node's row slots (a compile-time count for KCFix), rows >= len are masked
    for (uint32_t r0 = 0; r0 < k.rows; r0 += 64) {
        const uint32_t r = r0 + lane;
        const uint32_t rc = r < last ? r : last;
        const uint32_t base = k.o.rc_cent + (mrow0 + rc) * k.RBS + wave * slice;
        u32x4_t d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < pieces) d[i] = *(LA u32x4_t*)(k.L + base + i * 16);
        uint32_t part = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (i < pieces) part += popc4v(d[i] & xv[i]);
        if (r < len) pp[wave * k.rows + r] = part;
    }
    // a full barrier: it also orders the previous insertion's HBM stores before the RowMeta read
    // that follows the compare (an LDS-only barrier measured no faster)
    __syncthreads();
    uint32_t biu = MINMODE ? (0xFFFFu << 16 | 1u) : 1u, br = NONE;  // best of this lane: inter << 16 | union, row
    bool anyc = false;
    for (uint32_t r0 = 0; r0 < k.rows; r0 += 64) {
        const uint32_t r = r0 + lane;
        const uint32_t rc = r < last ? r : last;
        uint32_t inter = 0;
#pragma unroll
        for (int w = 0; w < TW; ++w) inter += pp[w * k.rows + rc];
        const uint32_t card = lds<uint32_t>(k.L, k.o.rc_card)[mrow0 + rc];
        uint32_t un = card + vec_pc - inter;
        if (want_counts && wave == 0 && r < len) { s_i[r] = inter; s_u[r] = un; }
        un = un < 1u ? 1u : un;
        anyc = anyc || (r < len && card != 0);
        const bool take = r < len && cand_better<MINMODE>(inter, un, r, biu >> 16, biu & 0xFFFFu, br);
        biu = take ? (inter << 16 | un) : biu;
        br = take ? r : br;
    }
    // 16-lane rows: rotate-and-keep-better butterfly leaves the row's best in all of its lanes
#define BB_STEP(N)                                                                                   \
    {                                                                                                \
        const uint32_t oiu = row_ror<N>(biu), orr = row_ror<N>(br);                                  \
        const bool take = cand_better<MINMODE>(oiu >> 16, oiu & 0xFFFFu, orr, biu >> 16, biu & 0xFFFFu, br); \
        biu = take ? oiu : biu;                                                                      \
        br = take ? orr : br;                                                                        \
    }
    BB_STEP(8) BB_STEP(4) BB_STEP(2) BB_STEP(1)
#undef BB_STEP
    // across the four 16-lane rows: row_bcast15 folds rows 0->1 and 2->3, row_bcast31 folds 1->3;
    // lane 63 then holds the best of the wave
#define BB_BCAST(CTRL, ROWMASK)                                                                       \
    {                                                                                                \
        const uint32_t oiu = (uint32_t)__builtin_amdgcn_update_dpp((int)biu, (int)biu, CTRL, ROWMASK, 0xF, false); \
        const uint32_t orr = (uint32_t)__builtin_amdgcn_update_dpp((int)br, (int)br, CTRL, ROWMASK, 0xF, false);   \
        const bool take = cand_better<MINMODE>(oiu >> 16, oiu & 0xFFFFu, orr, biu >> 16, biu & 0xFFFFu, br); \
        biu = take ? oiu : biu;                                                                      \
        br = take ? orr : br;                                                                        \
    }
    BB_BCAST(0x142, 0xA) BB_BCAST(0x143, 0xC)
#undef BB_BCAST
    const uint32_t wiu = rdlane(biu, 63), wr = rdlane(br, 63);
    if (out_any_card) *out_any_card = __ballot(anyc) != 0ull;
    Cand best;
    best.i = wiu >> 16;
    best.u = wiu & 0xFFFFu;
    best.r = wr;
    return best;
}

// write one node row: centroid (from LDS), popcount, link, meta
template <class KCt>
__device__ __forceinline__ void node_put_row(const KCt& k, uint32_t nd, uint32_t row, uint32_t cent_off, uint32_t card,
                                             uint32_t link, uint32_t sub, uint32_t n, uint32_t slot, u64 s1, u64 s2) {
    const size_t m = (size_t)nd * NG + row;
    for (int ch = threadIdx.x; ch < k.RBc; ch += TB)
        stg<u32x4_t>(k.cent + m * (size_t)k.RB + (size_t)ch * 16, lds<u32x4_t>(k.L, cent_off)[ch]);
    if (threadIdx.x == 0) {
        stg<uint32_t>(k.card + m, card);
        stg<uint32_t>(k.link + m, link);
        u32x4_t a, b;
        a.x = sub; a.y = n; a.z = slot; a.w = 0;
        b.x = (uint32_t)s1; b.y = (uint32_t)(s1 >> 32); b.z = (uint32_t)s2; b.w = (uint32_t)(s2 >> 32);
        stg<u32x4_t>((uint8_t*)(k.rm + m), a);
        stg<u32x4_t>((uint8_t*)(k.rm + m) + 16, b);
    }
}

// ---- radius complement terms (similarity.py:192-202) on CF(slot) [+ element] -----------
template <class KCt>
__device__ __forceinline__ void radius_terms(const KCt& k, const Elem& el, int& red_slot, uint32_t slotw, const uint8_t* crow, bool add_elem,
                                             u64 n, u64& sc, u64& sq) {
    u64 acc[2] = {0, 0};
    for (int b = threadIdx.x; b < k.nb; b += TB) {
        uint32_t v[8], e[8];
        cf_load8(k, slotw, b, v, crow);
        if (add_elem) {
            elem_cols(k, el, b, e);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] += e[q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u64 vv = v[q];
            const u64 bit = n <= 1 ? (u64)((vv & 0xFF) != 0) : (2ull * vv >= n ? 1ull : 0ull);
            acc[0] += bit;
            acc[1] += 2ull * vv * bit + bit;
        }
    }
    block_sum<2>(k, red_slot, acc);
    sc = acc[0];
    sq = acc[1];
}

__device__ __forceinline__ double radius_compl(u64 s1, u64 s2, u64 sc, u64 sq, u64 n) {
    const double jt = isim_from_moments(s1, s2, n);
    const double jt1 = isim_from_moments(s1 + sc, s2 + sq, n + 1);
    return (jt1 * (double)(n + 1) - jt * (double)(n - 1)) / 2;
}

template <class KCt>
__device__ __forceinline__ double tol_lookup(const KCt& k, u64 old_n) {
    if (k.tol == nullptr || old_n >= (u64)k.tol_len) return 0.0;
    return ldg<double>(k.tol + old_n);
}

// merge_accept_fn(threshold, new_ls, new_n, old_ls, nom_ls, old_n, nom_n) of _merges.py,
// on exact moments.  Uniform across the block.
template <class KCt>
__device__ __forceinline__ bool merge_accept(const KCt& k, const Elem& el, int& red_slot, uint32_t slotT, const uint8_t* crowT, u64 nT, u64 s1T,
                                             u64 s2T, u64 new_n, u64 s1n, u64 s2n) {
    const double thr = k.thr;
    const int crit = KCt::crit_fixed >= 0 ? KCt::crit_fixed : k.crit;
    switch (crit) {
        case BBH_CRIT_DIAMETER:
            return isim_from_moments(s1n, s2n, new_n) >= thr;
        case BBH_CRIT_TOL_DIAMETER: {
            const double new_dc = isim_from_moments(s1n, s2n, new_n);
            if (new_dc < thr) return false;
            if (nT == 1) return true;
            const double old_dc = isim_from_moments(s1T, s2T, nT);
            return new_dc >= old_dc - tol_lookup(k, nT);
        }
        case BBH_CRIT_TOL_LEGACY: {
            const double new_dc = isim_from_moments(s1n, s2n, new_n);
            if (new_dc < thr) return false;
            if (nT == 1 || el.nS != 1) return true;
            const double old_dc = isim_from_moments(s1T, s2T, nT);
            return (new_dc * (double)new_n - old_dc * (double)(nT - 1)) / 2 >= old_dc - k.tolerance;
        }
        case BBH_CRIT_RADIUS: {
            u64 sc, sq;
            radius_terms(k, el, red_slot, slotT, crowT, true, new_n, sc, sq);
            return radius_compl(s1n, s2n, sc, sq, new_n) >= thr;
        }
        case BBH_CRIT_TOL_RADIUS: {
            u64 sc, sq;
            radius_terms(k, el, red_slot, slotT, crowT, true, new_n, sc, sq);
            const double new_rc = radius_compl(s1n, s2n, sc, sq, new_n);
            if (new_rc < thr) return false;
            if (nT == 1) return true;
            radius_terms(k, el, red_slot, slotT, crowT, false, nT, sc, sq);
            const double old_rc = radius_compl(s1T, s2T, sc, sq, nT);
            return new_rc >= old_rc - tol_lookup(k, nT);
        }
        default:
            return false;  // never-merge
    }
}

// Allocation of ids / slots / nodes.  Single-tree launches own the tree: the counters are
// wave-uniform registers.  In SUB (concurrent gates of one tree) mode the counters live in the
// TreeDev and are bumped with one device-scope atomic by thread 0, broadcast through LDS.
template <bool SUB, class KCt>
__device__ __forceinline__ uint32_t alloc_n(const KCt& k, uint32_t& reg, uint32_t* gctr, uint32_t cnt, int bc_slot) {
    if constexpr (!SUB) {
        const uint32_t r = reg;
        reg += cnt;
        return r;
    } else {
        LA uint32_t* bc = lds<uint32_t>(k.L, k.o.bc);
        if (threadIdx.x == 0) bc[bc_slot] = atomicAdd(gctr, cnt);
        __syncthreads();
        const uint32_t r = uni(bc[bc_slot]);
        __syncthreads();
        return r;
    }
}

// Majority vote over the m centroid rows of a node -> packed vector in LDS (o.vec), the
// centroid of the node's centroids (bitbirch.py:162-211 via centroid_from_sum, _py_similarity.py:
// 12-42: bit = 2*count >= m).  Column counts are kept bit-sliced: every lane owns one dword (32
// columns) of the row, every wave a quarter of the rows; adding a row is a ripple of half adders
// over the count's NPW bit planes.  The four partial counts meet in LDS, are added with full
// adders (NPT planes) and compared with ceil(m/2) plane by plane, which leaves the packed
// majority dword directly.  Needs m <= 4 * (2^NPW - 1) and m < 2^NPT.
template <int NPW, int NPT, class KCt>
__device__ __forceinline__ void majority_rows(const KCt& k, bool lm, uint32_t mrow0, const uint8_t* cent, uint32_t m) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int RBdw = k.RB / 4;
    const uint32_t half = (m + 1) >> 1;
    LA uint32_t* planes = lds<uint32_t>(k.L, k.o.planes);
    for (int d0 = 0; d0 < RBdw; d0 += 64) {
        const int d = d0 + lane;
        const bool actd = d < RBdw;
        const int dc = actd ? d : 0;
        uint32_t p[NPW];
#pragma unroll
        for (int i = 0; i < NPW; ++i) p[i] = 0;
        for (uint32_t r0 = (uint32_t)wave; r0 < m; r0 += 4 * TW) {
            uint32_t c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // four rows requested before the first is consumed
                const uint32_t r = r0 + u * TW;
                const uint32_t rc = r < m ? r : m - 1;
                if (lm) c[u] = *(LA uint32_t*)(k.L + k.o.rc_cent + (mrow0 + rc) * k.RBS + (uint32_t)dc * 4);
                else c[u] = ldg<uint32_t>(cent + (size_t)rc * k.RB + (size_t)dc * 4);
            }
            // four rows at once through carry-save adders (round 5; a ripple of half adders per row before: 18-27 instructions a
            // row, three quarters of the 8.9 k cycles this step cost per leaf split at bf 254): c0 + c1 + c2 -> sum, carry;
            // sum + c3 + plane 0 -> plane 0, carry'; carry + carry' + plane 1 -> plane 1, carry''; carry'' ripples from plane 2
#pragma unroll
            for (int u = 0; u < 4; ++u) c[u] = (actd && r0 + u * TW < m) ? c[u] : 0u;
            {
                static_assert(NPW >= 3, "carry-save step needs three planes");
                const uint32_t x01 = c[0] ^ c[1];
                const uint32_t s1 = x01 ^ c[2];
                const uint32_t k1 = (x01 & c[2]) | (~x01 & c[0]);  // majority(c0, c1, c2): one v_bfi_b32
                const uint32_t x0 = p[0] ^ s1;
                const uint32_t k2 = (x0 & c[3]) | (~x0 & s1);       // majority(plane 0, sum, c3)
                p[0] = x0 ^ c[3];
                const uint32_t x1 = p[1] ^ k1;
                uint32_t cc = (x1 & k2) | (~x1 & k1);               // majority(plane 1, carry, carry')
                p[1] = x1 ^ k2;
#pragma unroll
                for (int i = 2; i < NPW; ++i) {
                    const uint32_t t = p[i] & cc;
                    p[i] ^= cc;
                    cc = t;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NPW; ++i) planes[(wave * NPLW + i) * 64 + lane] = p[i];
        __syncthreads();
        if (wave == 0) {
            uint32_t sacc[NPT];
#pragma unroll
            for (int i = 0; i < NPT; ++i) sacc[i] = (i < NPW) ? p[i] : 0u;
#pragma unroll
            for (int w = 1; w < TW; ++w) {
                uint32_t carry = 0;
#pragma unroll
                for (int i = 0; i < NPT; ++i) {
                    const uint32_t a = sacc[i], bq = (i < NPW) ? planes[(w * NPLW + i) * 64 + lane] : 0u;
                    const uint32_t x = a ^ bq;
                    sacc[i] = x ^ carry;
                    carry = (a & bq) | (carry & x);
                }
            }
            uint32_t gt = 0, eq = 0xFFFFFFFFu;
#pragma unroll
            for (int i = NPT - 1; i >= 0; --i) {
                const uint32_t tb = ((half >> i) & 1u) ? 0xFFFFFFFFu : 0u;
                gt |= eq & sacc[i] & ~tb;
                eq &= ~(sacc[i] ^ tb);
            }
            if (actd) lds<uint32_t>(k.L, k.o.vec)[d] = gt | eq;
        }
        __syncthreads();
    }
}

// ---- _split_node (bitbirch.py:162-211) -------------------------------------------------
// Splits node `nd` (len = bf+1 rows).  Leaves the two tracking BitFeatures' centroids in
// LDS (o.cA / o.cB) and publishes through bc: [0]=node1 [3]=cardA [4]=cardB [7]=nA [8]=nB
// [9]=slotA [10]=slotB [11]=n overflow flag.
template <bool SUB, bool PROF, class KCt>
__device__ __forceinline__ void split_node(const KCt& k, const Elem& el, int& red_slot, int& cmp_par, uint32_t nd, uint32_t ms,
                                           bool ms_valid, uint32_t trk_slot, uint32_t& cN, uint32_t& c32, uint32_t& cFirst,
                                           uint32_t* gctr, u64 (&sph)[8]) {
    // ms: LDS mirror slot to work in (2048-bit rows only); ms_valid: it already holds nd's rows.
    // trk_slot: CF slot word of the tracking BitFeature that describes nd in its parent (NONE for
    // the root).  It has not received the element being inserted yet, so the CFs of nd's rows add
    // up to exactly CF(trk_slot) + element: only the smaller half is summed, the other half is the
    // difference.
    const int tid = threadIdx.x;
    u64 smark = PROF ? __builtin_amdgcn_s_memtime() : 0;
#define SPH(i) do { if constexpr (PROF) { const u64 _n = __builtin_amdgcn_s_memtime(); sph[i] += _n - smark; smark = _n; } } while (0)
    const uint32_t rows = k.rows;
    const size_t meta = (size_t)nd * NG;
    const u32x4_t hold = ldg<u32x4_t>(k.hdr + nd);  // {len, leaf word, prev, next}
    const uint32_t m = uni(hold.x);
    const int nb = k.nb;
    const bool lm = k.nm > 0;  // rows are staged in (or already live in) LDS mirror `ms`
    const uint32_t mrow0 = ms * rows;
    uint8_t* cent = k.cent + meta * (size_t)k.RB;
    LA uint32_t* bc = lds<uint32_t>(k.L, k.o.bc);
    LA uint32_t* dst = lds<uint32_t>(k.L, k.o.dst);
    LA uint32_t* mcard = lds<uint32_t>(k.L, k.o.mcard);
    LA uint32_t* mlink = lds<uint32_t>(k.L, k.o.mlink);
    LA u32x4_t* mrm = lds<u32x4_t>(k.L, k.o.mrm);
    LA u32x4_t* vec = lds<u32x4_t>(k.L, k.o.vec);
    LA uint32_t* lst = lds<uint32_t>(k.L, k.o.i1);  // CF slots of the smaller half (after step 5)
    LA uint32_t* lrow = lds<uint32_t>(k.L, k.o.u1); // and their rows (a BitFeature of one member has its cluster features in its centroid row)
    // byte b of row r's centroid as it was before the rows were distributed (step 7b): the LDS mirror or the staged copy
    auto row_byte = [&](uint32_t r, int b) -> uint32_t {
        if (lm) return *(LA uint8_t*)(k.L + k.o.rc_cent + (mrow0 + r) * k.RBS + (uint32_t)b);
        return ldg<uint8_t>(k.scratch + (size_t)r * (size_t)k.RB + (size_t)b);
    };
    // 0. row metadata (and, if needed, the centroid rows) to LDS; zero the comparison vector padding
    for (uint32_t r = tid; r < m; r += TB) {
        const uint32_t cd = ldg<uint32_t>(k.card + meta + r);
        mcard[r] = cd;
        if (lm && !ms_valid) lds<uint32_t>(k.L, k.o.rc_card)[mrow0 + r] = cd;
        mlink[r] = ldg<uint32_t>(k.link + meta + r);
        mrm[2 * r] = ldg<u32x4_t>((uint8_t*)(k.rm + meta + r));
        mrm[2 * r + 1] = ldg<u32x4_t>((uint8_t*)(k.rm + meta + r) + 16);
    }
    if (lm && !ms_valid) {
        const uint32_t rbc = (uint32_t)k.RBc, total = m * rbc;  // 16-byte pieces of the node's rows
        for (uint32_t i0 = 0; i0 < total; i0 += 4 * TB) {  // four 16-byte pieces per thread in flight
            u32x4_t t4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * TB + tid;
                t4[u] = ldg<u32x4_t>(cent + (size_t)(i < total ? i : total - 1) * 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * TB + tid;
                if (i < total) *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + i / rbc) * k.RBS + (i % rbc) * 16) = t4[u];
            }
        }
    }
    for (int ch = tid; ch < k.RBc; ch += TB) vec[ch] = (u32x4_t)(0);
    __syncthreads();
    SPH(0);
    // 1. majority centroid of the node's centroids (bit-sliced column counts)
    if (m <= 124) majority_rows<5, 7>(k, lm, mrow0, cent, m);
    else majority_rows<NPLW, NPLT>(k, lm, mrow0, cent, m);
    const uint32_t pc = lds_vec_popcount(k, k.o.vec);
    SPH(1);
    // 2. fp1 = first argmin of similarity to that centroid; 3. similarities to fp1, fp2 = first
    //    argmin; 4. similarities to fp2
    uint32_t f1;
    if (lm) {
        f1 = node_best_mirror<true>(k, cmp_par, m, k.o.vec, pc, false, false, ms, nullptr).r;
        for (int ch = tid; ch < k.RBc; ch += TB) vec[ch] = *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + f1) * k.RBS + ch * 16);
        __syncthreads();
        const uint32_t f2 = node_best_mirror<true>(k, cmp_par, m, k.o.vec, uni(mcard[f1]), true, false, ms, nullptr).r;
        for (int ch = tid; ch < k.RBc; ch += TB) vec[ch] = *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + f2) * k.RBS + ch * 16);
        __syncthreads();
        (void)node_best_mirror<true>(k, cmp_par, m, k.o.vec, uni(mcard[f2]), true, true, ms, nullptr);
        __syncthreads();  // the counts of the last pass are written after its internal barrier
    } else {
        f1 = node_best<false, true>(k, cmp_par, nd, (int)m, k.o.vec, pc, false, false, false, nullptr, nullptr).r;
        for (int ch = tid; ch < k.RBc; ch += TB) vec[ch] = ldg<u32x4_t>(cent + (size_t)f1 * k.RB + (size_t)ch * 16);
        __syncthreads();
        const uint32_t f2 = node_best<false, true>(k, cmp_par, nd, (int)m, k.o.vec, uni(mcard[f1]), true, false, false, nullptr, nullptr).r;
        for (int ch = tid; ch < k.RBc; ch += TB) vec[ch] = ldg<u32x4_t>(cent + (size_t)f2 * k.RB + (size_t)ch * 16);
        __syncthreads();
        (void)node_best<false, true>(k, cmp_par, nd, (int)m, k.o.vec, uni(mcard[f2]), true, true, false, nullptr, nullptr);
    }
    SPH(2);
    // 5. node1_closer = sims_fp1 > sims_fp2 (exact cross-multiplication), node1_closer[fp1] = True
    {
        LA uint32_t* i1 = lds<uint32_t>(k.L, k.o.i1);
        LA uint32_t* u1 = lds<uint32_t>(k.L, k.o.u1);
        LA uint32_t* i2 = lds<uint32_t>(k.L, k.o.i2);
        LA uint32_t* u2 = lds<uint32_t>(k.L, k.o.u2);
        for (uint32_t r = tid; r < m; r += TB) {
            const uint32_t a1 = i1[r], b1 = u1[r] < 1u ? 1u : u1[r];
            const uint32_t a2 = i2[r], b2 = u2[r] < 1u ? 1u : u2[r];
            const bool to1 = ((u64)a1 * b2 > (u64)a2 * b1) || r == f1;
            dst[r] = to1 ? 0x80000000u : 0u;
        }
    }
    __syncthreads();
    if (tid < 64) {
        // stable partition positions: ballot prefix counts, 64 rows at a time
        uint32_t n1 = 0, n2 = 0;
        u64 nA = 0, nB = 0;
        const u64 below = (1ull << tid) - 1ull;
        for (uint32_t r0 = 0; r0 < m; r0 += 64) {
            const uint32_t r = r0 + tid;
            const bool valid = r < m;
            const bool to1 = valid && (dst[r] & 0x80000000u) != 0;
            const u64 nr = valid ? (u64)mrm[2 * r].y : 0ull;  // RowMeta.n
            const u64 m1 = __ballot(to1), mv = __ballot(valid);
            const u64 m2 = mv & ~m1;
            if (valid) dst[r] = to1 ? (0x80000000u | (n1 + (uint32_t)__popcll(m1 & below))) : (n2 + (uint32_t)__popcll(m2 & below));
            // (n_samples < 2^32: two 16-bit halves summed as 32-bit lanes - 64 x 65 535 fits - instead of 64-bit wave sums)
            {
                const uint32_t n32 = (uint32_t)nr, lo16 = n32 & 0xFFFFu, hi16 = n32 >> 16;
                const bool to2 = valid && !to1;
                nA += (u64)wsum32(to1 ? lo16 : 0u) + ((u64)wsum32(to1 ? hi16 : 0u) << 16);
                nB += (u64)wsum32(to2 ? lo16 : 0u) + ((u64)wsum32(to2 ? hi16 : 0u) << 16);
            }
            n1 += (uint32_t)__popcll(m1);
            n2 += (uint32_t)__popcll(m2);
        }
        // CF slots of the smaller half, in row order
        const bool small1 = n1 <= n2;
        for (uint32_t r0 = 0; r0 < m; r0 += 64) {
            const uint32_t r = r0 + tid;
            if (r < m) {
                const uint32_t d = dst[r];
                if (((d & 0x80000000u) != 0) == small1) { lst[d & 0x7FFFFFFFu] = slot_effective(mrm[2 * r].z, mrm[2 * r].y); lrow[d & 0x7FFFFFFFu] = r; }  // RowMeta.slot (.n == 1: the centroid row)
            }
        }
        if (tid == 0) {
            bc[7] = (uint32_t)nA;
            bc[8] = (uint32_t)nB;
            bc[11] = (nA > 0xFFFFFFFFull || nB > 0xFFFFFFFFull) ? 1u : 0u;
            bc[5] = n1;
            bc[6] = n2;
            lds<u64>(k.L, k.o.stats)[4]++;
            lds<u64>(k.L, k.o.stats)[5]++;
        }
    }
    // 6. ids for the new node and the two tracking BitFeatures (always cf32): every thread
    //    keeps the (uniform) allocation counters in registers
    const uint32_t node1 = alloc_n<SUB>(k, cN, gctr + C_NODES, k.nblk, 14);  // (a full-capacity node)
    // (the row that tracked nd goes on tracking node1 and KEEPS its cf32 slot: the old sums are read - by_difference, below,
    // every thread its own features - before the new ones are written to the same place.  Until round 5 both halves got new
    // slots and the old one was abandoned: half of the cf32 pool of a large tree was such garbage, 13 GB of 24 GB at 20 M
    // S-ecfp rows.)
    uint32_t slotA_i, slotB_i;
    if (trk_slot != NONE) {
        slotA_i = trk_slot & 0x3FFFFFFFu;
        slotB_i = alloc_n<SUB>(k, c32, gctr + C_N32, 1, 15);
    } else {
        slotA_i = alloc_n<SUB>(k, c32, gctr + C_N32, 2, 15);
        slotB_i = slotA_i + 1;
    }
    const uint32_t was_leaf = uni(hold.y) & HW_LEAF, prev_leaf = uni(hold.z);
    if (was_leaf && prev_leaf == NONE) {
        if constexpr (SUB) { if (tid == 0) stg<uint32_t>(gctr + C_FIRST_LEAF, node1); }
        else cFirst = node1;
    }
    SPH(3);
    // 7a. stage all centroid rows (kept rows are compacted in place afterwards); rows that
    //     already sit in LDS need no staging
    if (!lm) {
        for (uint32_t i = tid; i < m * (uint32_t)k.RBc; i += TB)
            stg<u32x4_t>(k.scratch + (size_t)i * 16, ldg<u32x4_t>(cent + (size_t)i * 16));
    }
    __syncthreads();
    const u64 nA = uni(bc[7]), nB = uni(bc[8]);
    const uint32_t n1 = uni(bc[5]), n2 = uni(bc[6]);
    const uint32_t slotA = (2u << 30) | slotA_i, slotB = (2u << 30) | slotB_i;
    if (tid == 0) {
        // 9. leaf chain: node1 goes immediately before nd (bitbirch.py:182-188)
        bc[0] = node1;
        bc[9] = slotA_i;
        bc[10] = slotB_i;
        // word-granular header writes: a node's `next` word is only ever written by the split of
        // the node that follows it in the chain, every other word only by its own subtree, so
        // concurrent gates never race on the leaf chain
        u32x4_t h = hold;
        h.y &= HW_LEAF;
        u32x4_t h1;
        h1.x = n1; h1.y = hw_make(h.y, rows); h1.z = NONE; h1.w = NONE;
        if (h.y) {
            h1.z = h.z;
            if (h.z != NONE) stg<uint32_t>((uint8_t*)(k.hdr + h.z) + 12, node1);
            h1.w = nd;
        }
        stg<u32x4_t>(k.hdr + node1, h1);
        stg<uint32_t>((uint8_t*)(k.hdr + nd), n2);
        if (h.y) stg<uint32_t>((uint8_t*)(k.hdr + nd) + 8, node1);
    }
    // 7b. distribute rows in their original order
    {
        uint8_t* cent1 = k.cent + (size_t)node1 * NG * (size_t)k.RB;
        for (uint32_t i = tid; i < m * (uint32_t)k.RBc; i += TB) {
            const uint32_t r = i / k.RBc, ch = i % k.RBc;
            const uint32_t d = dst[r];
            uint8_t* dstbase = (d & 0x80000000u) ? cent1 : cent;
            u32x4_t v;
            if (lm) v = *(LA u32x4_t*)(k.L + k.o.rc_cent + (mrow0 + r) * k.RBS + ch * 16);
            else v = ldg<u32x4_t>(k.scratch + (size_t)i * 16);
            stg<u32x4_t>(dstbase + (size_t)(d & 0x7FFFFFFFu) * k.RB + (size_t)ch * 16, v);
        }
        for (uint32_t r = tid; r < m; r += TB) {
            const uint32_t d = dst[r];
            const size_t mm = ((d & 0x80000000u) ? (size_t)node1 * NG : meta) + (d & 0x7FFFFFFFu);
            stg<uint32_t>(k.card + mm, mcard[r]);
            stg<uint32_t>(k.link + mm, mlink[r]);
            stg<u32x4_t>((uint8_t*)(k.rm + mm), mrm[2 * r]);
            stg<u32x4_t>((uint8_t*)(k.rm + mm) + 16, mrm[2 * r + 1]);
        }
    }
    SPH(4);
    // 8. tracking BitFeatures: CF = sum of member CFs; centroid from the final CF
    for (int ch = tid; ch < k.RBc; ch += TB) {
        lds<u32x4_t>(k.L, k.o.cA)[ch] = (u32x4_t)(0);
        lds<u32x4_t>(k.L, k.o.cB)[ch] = (u32x4_t)(0);
    }
    __syncthreads();
    u64 cards[2] = {0, 0};
    const bool by_difference = trk_slot != NONE;
    const bool small1 = n1 <= n2;
    const uint32_t ns = small1 ? n1 : n2;
    for (int b = tid; b < nb; b += TB) {
        uint32_t accA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, accB[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        SPH(5);
        if (by_difference) {
            // the element and the old tracking CF first: they are needed last
            uint32_t tot[8], x8[8], sm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            cf32_load8(k, trk_slot, b, tot);
            elem_cols(k, el, b, x8);
            for (uint32_t r0 = 0; r0 < ns; r0 += SPLIT_MLP) {  // this many member CFs in flight
                uint32_t sw[SPLIT_MLP], rw[SPLIT_MLP], tiers_and = 0xFFFFFFFFu;
                bool wide = false;
                // (the group's slot words and rows: one LDS read per lane and v_readlane, not 2 x SPLIT_MLP uniform reads)
                // (only when whole waves are in this loop - rows of 512 .. 2048 bits: v_readlane reads lanes that must have loaded)
                static_assert(SPLIT_MLP <= 64, "one entry per lane");
                const bool whole_waves = (nb & 63) == 0 && nb <= TB;
                uint32_t swv = 0, rwv = 0;
                if (whole_waves) {
                    const uint32_t gl = (uint32_t)tid & 63u, gi = r0 + gl < ns ? r0 + gl : ns - 1;
                    swv = lst[gi];
                    rwv = lrow[gi];
                }
#pragma unroll
                for (int u = 0; u < SPLIT_MLP; ++u) {
                    if (whole_waves) {
                        sw[u] = rdlane(swv, u);
                        rw[u] = rdlane(rwv, u);
                    } else {
                        sw[u] = uni(lst[r0 + u < ns ? r0 + u : ns - 1]);
                        rw[u] = uni(lrow[r0 + u < ns ? r0 + u : ns - 1]);
                    }
                    wide = wide || (sw[u] >> 30) == 1u || (sw[u] >> 30) == 2u;
                    tiers_and &= sw[u];
                }
                if ((tiers_and >> 30) == TIER_LAZY) {
                    // BitFeatures of one member only (a leaf of single fingerprints): the bits of their centroid bytes, four to a
                    // dword by one 24-bit multiply (bit i of a nibble -> byte i), summed in byte lanes (16 rows at most)
                    uint32_t lo = 0, hi = 0;
#pragma unroll
                    for (int u = 0; u < SPLIT_MLP; ++u) {
                        if (r0 + u < ns) {
                            const uint32_t xb = row_byte(rw[u], b);
                            lo += __umul24(xb & 15u, 0x204081u) & 0x01010101u;
                            hi += __umul24(xb >> 4, 0x204081u) & 0x01010101u;
                        }
                    }
                    // feature q of the byte is bit 7 - q
                    sm[7] += lo & 0xFFu; sm[6] += (lo >> 8) & 0xFFu; sm[5] += (lo >> 16) & 0xFFu; sm[4] += lo >> 24;
                    sm[3] += hi & 0xFFu; sm[2] += (hi >> 8) & 0xFFu; sm[1] += (hi >> 16) & 0xFFu; sm[0] += hi >> 24;
                } else if (!wide) {
                    // uint8 CFs (the usual leaf): 8 bytes per row and thread, summed as four pairs of 16-bit lanes
                    // (16 rows x 255 cannot overflow them).  The requests are branch-free - the reserved slot of a BitFeature of one
                    // member is read and ignored - and its centroid byte takes the loaded pair's place, spread to the
                    // same layout (feature q of the byte is bit 7 - q: the nibbles' spread bytes reversed).
                    u32x2_t q[SPLIT_MLP];
#pragma unroll
                    for (int u = 0; u < SPLIT_MLP; ++u)
                        q[u] = ldg<u32x2_t>(k.cf8 + (size_t)(sw[u] & 0x3FFFFFFFu) * (size_t)k.F + (size_t)b * 8);
                    uint32_t lx = 0, hx = 0, ly = 0, hy = 0;
#pragma unroll
                    for (int u = 0; u < SPLIT_MLP; ++u) {
                        if (r0 + u < ns) {
                            u32x2_t qq = q[u];
                            if ((sw[u] >> 30) == TIER_LAZY) {
                                const uint32_t xb = row_byte(rw[u], b);
                                qq.x = __builtin_bswap32(__umul24(xb >> 4, 0x204081u) & 0x01010101u);
                                qq.y = __builtin_bswap32(__umul24(xb & 15u, 0x204081u) & 0x01010101u);
                            }
                            lx += qq.x & 0x00FF00FFu; hx += (qq.x >> 8) & 0x00FF00FFu;
                            ly += qq.y & 0x00FF00FFu; hy += (qq.y >> 8) & 0x00FF00FFu;
                        }
                    }
                    sm[0] += lx & 0xFFFFu; sm[1] += hx & 0xFFFFu; sm[2] += lx >> 16; sm[3] += hx >> 16;
                    sm[4] += ly & 0xFFFFu; sm[5] += hy & 0xFFFFu; sm[6] += ly >> 16; sm[7] += hy >> 16;
                } else {
#pragma unroll
                    for (int u0 = 0; u0 < SPLIT_MLP; u0 += SPLIT_MLP_WIDE) {
                        u32x4_t raw[SPLIT_MLP_WIDE][2];
#pragma unroll
                        for (int u = 0; u < SPLIT_MLP_WIDE; ++u) {
                            if ((sw[u0 + u] >> 30) == TIER_LAZY) { raw[u][0] = (u32x4_t)(0); raw[u][1] = (u32x4_t)(0); raw[u][0].x = row_byte(rw[u0 + u], b); }
                            else cf_load_raw(k, sw[u0 + u], b, raw[u]);
                        }
#pragma unroll
                        for (int u = 0; u < SPLIT_MLP_WIDE; ++u) {
                            if (r0 + u0 + u < ns) {
                                uint32_t v[8];
                                cf_unpack_raw(sw[u0 + u] >> 30, raw[u], v);
#pragma unroll
                                for (int q8 = 0; q8 < 8; ++q8) sm[q8] += v[q8];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                const uint32_t other = tot[q8] + x8[q8] - sm[q8];
                accA[q8] = small1 ? sm[q8] : other;
                accB[q8] = small1 ? other : sm[q8];
            }
        } else {
            for (uint32_t r0 = 0; r0 < m; r0 += SPLIT_MLP_WIDE) {
                u32x4_t raw[SPLIT_MLP_WIDE][2];
                uint32_t sw[SPLIT_MLP_WIDE];
#pragma unroll
                for (int u = 0; u < SPLIT_MLP_WIDE; ++u) {
                    const uint32_t ru = r0 + u < m ? r0 + u : m - 1;
                    sw[u] = uni(slot_effective(mrm[2 * ru].z, mrm[2 * ru].y));  // RowMeta.slot (.n == 1: the centroid row)
                    if ((sw[u] >> 30) == TIER_LAZY) { raw[u][0] = (u32x4_t)(0); raw[u][1] = (u32x4_t)(0); raw[u][0].x = row_byte(ru, b); }
                    else cf_load_raw(k, sw[u], b, raw[u]);
                }
#pragma unroll
                for (int u = 0; u < SPLIT_MLP_WIDE; ++u) {
                    if (r0 + u < m) {
                        uint32_t v[8];
                        cf_unpack_raw(sw[u] >> 30, raw[u], v);
                        if (uni(dst[r0 + u]) & 0x80000000u) {
#pragma unroll
                            for (int q8 = 0; q8 < 8; ++q8) accA[q8] += v[q8];
                        } else {
#pragma unroll
                            for (int q8 = 0; q8 < 8; ++q8) accB[q8] += v[q8];
                        }
                    }
                }
            }
        }
        SPH(6);
        cf_store8(k, slotA, b, accA);
        cf_store8(k, slotB, b, accB);
        const uint32_t ba = centroid_byte(accA, nA), bb_ = centroid_byte(accB, nB);
        lds<uint8_t>(k.L, k.o.cA)[b] = (uint8_t)ba;
        lds<uint8_t>(k.L, k.o.cB)[b] = (uint8_t)bb_;
        cards[0] += __popc(ba);
        cards[1] += __popc(bb_);
    }
    block_sum<2>(k, red_slot, cards);  // barrier inside: cA / cB complete for every thread
    if (tid == 0) {
        bc[3] = (uint32_t)cards[0];
        bc[4] = (uint32_t)cards[1];
    }
    __syncthreads();
    SPH(7);
#undef SPH
}

// CF += element on one ancestor row (closest_subcluster.update, bitbirch.py:352-357); slow path
template <class KCt>
__device__ __forceinline__ void update_tracker_slow(const KCt& k, const Elem& el, int& red_slot, int lvl, int& stop) {
    const int tid = threadIdx.x;
    const uint32_t P = uni(lds<uint32_t>(k.L, k.o.path_node)[lvl]), jp = uni(lds<uint32_t>(k.L, k.o.path_row)[lvl]);
    const size_t pm = (size_t)P * NG + jp;
    const uint32_t slotw = uni(lds<uint32_t>(k.L, k.o.path_slot)[lvl]);
    const u64 n_new = (u64)uni(lds<uint32_t>(k.L, k.o.path_n)[lvl]) + el.nS;
    u64 cc[1] = {0};
    for (int b = tid; b < k.nb; b += TB) {
        uint32_t v[8], x8[8];
        cf32_load8(k, slotw, b, v);
        elem_cols(k, el, b, x8);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += x8[q];
        cf_store8(k, slotw, b, v);
        const uint32_t byte = centroid_byte(v, n_new);
        stg<uint8_t>(k.cent + pm * (size_t)k.RB + b, (uint8_t)byte);
        cc[0] += __popc(byte);
    }
    block_sum<1>(k, red_slot, cc);
    if (n_new > 0xFFFFFFFFull) stop = STOP_RANGE;
    if (tid == 0) {
        stg<uint32_t>((uint8_t*)(k.rm + pm) + 4, (uint32_t)n_new);
        stg<uint32_t>((uint8_t*)(k.rm + pm) + 12, 0u);  // flip distance now stale
        stg<uint32_t>(k.card + pm, (uint32_t)cc[0]);
    }
}

// A sealed node (capacity below bf + 1: the compaction found it unchanged since the one before) is about to be inserted
// into: move it to a fresh full-capacity block.  Rows, per-row words and records are copied, the header is written with
// full capacity (and no compaction record), the leaf chain's neighbours and the parent's row (or the root) are pointed
// at the new id, the old header is cleared (its blocks are reclaimed by the next compaction).  Whole workgroup; ends with
// a barrier behind which the new node is complete in HBM for every thread of the workgroup.
template <class KCt>
__device__ __forceinline__ uint32_t thaw_node(const KCt& k, uint32_t nd, uint32_t P, uint32_t jp, uint32_t& cN, uint32_t& cRoot,
                                              uint32_t& cFirst) {
    const int tid = threadIdx.x;
    const u32x4_t h = ldg<u32x4_t>(k.hdr + nd);
    const uint32_t len = uni(h.x), hw = uni(h.y), prev = uni(h.z), next = uni(h.w);
    const uint32_t nn = cN;
    cN += k.nblk;
    const size_t so = (size_t)nd * NG, dn = (size_t)nn * NG;
    for (uint32_t i = tid; i < len * (uint32_t)k.RBc; i += TB)
        stg<u32x4_t>(k.cent + dn * (size_t)k.RB + (size_t)i * 16, ldg<u32x4_t>(k.cent + so * (size_t)k.RB + (size_t)i * 16));
    for (uint32_t r = tid; r < len; r += TB) {
        stg<uint32_t>(k.card + dn + r, ldg<uint32_t>(k.card + so + r));
        stg<uint32_t>(k.link + dn + r, ldg<uint32_t>(k.link + so + r));
        stg<u32x4_t>((uint8_t*)(k.rm + dn + r), ldg<u32x4_t>((const uint8_t*)(k.rm + so + r)));
        stg<u32x4_t>((uint8_t*)(k.rm + dn + r) + 16, ldg<u32x4_t>((const uint8_t*)(k.rm + so + r) + 16));
    }
    if (tid == 0) {
        u32x4_t hn;
        hn.x = len; hn.y = hw_make(hw, k.rows); hn.z = prev; hn.w = next;
        stg<u32x4_t>(k.hdr + nn, hn);
        if (hw & HW_LEAF) {
            if (prev != NONE) stg<uint32_t>((uint8_t*)(k.hdr + prev) + 12, nn);
            if (next != NONE) stg<uint32_t>((uint8_t*)(k.hdr + next) + 8, nn);
        }
        if (P != NONE) stg<uint32_t>(k.link + (size_t)P * NG + jp, nn);
        u32x4_t dead;
        dead.x = 0; dead.y = 0; dead.z = NONE; dead.w = NONE;
        stg<u32x4_t>(k.hdr + nd, dead);
        lds<u64>(k.L, k.o.stats)[7]++;  // (statistics: nodes thawed)
    }
    if ((hw & HW_LEAF) && prev == NONE) cFirst = nn;
    if (P == NONE) cRoot = nn;
    __syncthreads();
    return nn;
}

// one tree per workgroup.  ONE: the cold path of the fast kernel (bb_tree_fast.inc) - insert exactly element
// `e_first` of tree `trees`; allocation counters and statistics live in LDS (o.ctr / o.stats) across calls, the
// stop reason and the number of processed elements are returned through o.ctr[8] / o.ctr[9].
template <bool PROF, bool SUB, class KCt, bool ONE = false>
__device__ __forceinline__ void tree_insert_body(unsigned char* smem_raw, TreeDev* trees, const uint32_t* gate_nodes,
                                                 const uint32_t* gate_off, const uint32_t* gate_elems, long long e_first = 0) {
    TreeDev* T = (SUB || ONE) ? trees : trees + blockIdx.x;
    uint32_t* gctr = T->ctr;
    KCt k;
    k.cent = T->node_cent; k.card = T->node_card; k.link = T->node_link; k.rm = T->node_rm; k.hdr = T->node_hdr;
    k.scratch = T->scratch_cent; k.cf8 = T->cf8; k.cf16 = T->cf16; k.cf32 = T->cf32;
    k.bufs = T->bufs; k.width = (int)uni((uint32_t)T->width);
    k.crit = (int)uni((uint32_t)T->crit); k.tol_len = (int)uni((uint32_t)T->tol_len); k.thr = T->thr; k.tolerance = T->tolerance; k.tol = T->tol_table;
    k.L = (LA unsigned char*)smem_raw;
    if constexpr (KCt::dynamic_shape) {  // shape of the tree as run-time values
        k.F = (int)uni((uint32_t)T->F); k.nb = (int)uni((uint32_t)T->nbytes); k.RB = (int)uni((uint32_t)T->RB); k.RBc = k.RB / 16; k.RBS = k.RB + 16;
        k.bf = uni((uint32_t)T->bf); k.rows = k.bf + 1; k.nblk = node_blocks(k.rows);
        k.nm = (int)uni((uint32_t)T->use_root_cache);  // number of mirrored path levels
        k.o = smem_layout((int)k.bf, k.RB, k.nm);
    }
    const uint8_t* in_rows = T->rows;
    const long long row_stride = T->row_stride;
    const uint32_t g_off = SUB ? uni(gate_off[blockIdx.x]) : 0u;
    const long long n_elems = SUB ? (long long)(uni(gate_off[blockIdx.x + 1]) - g_off) : T->n_elems;
    uint32_t* out_leaf = T->out_leaf;
    const uint32_t cap_nodes = uni(T->cap_nodes), cap8 = uni(T->cap8), cap16 = uni(T->cap16), cap32 = uni(T->cap32);
    if constexpr (KCt::buf == 0) k.bufs = nullptr;
    const bool bufmode = KCt::buf == 0 ? false : (KCt::buf == 1 ? true : k.bufs != nullptr);
    const int tid = threadIdx.x;
    const int nb = k.nb;
    const uint32_t bf = k.bf;
    const int nm = k.nm;
    LA u64* stats = lds<u64>(k.L, k.o.stats);
    LA uint32_t* bc = lds<uint32_t>(k.L, k.o.bc);
    LA uint32_t* path_node = lds<uint32_t>(k.L, k.o.path_node);
    LA uint32_t* path_row = lds<uint32_t>(k.L, k.o.path_row);
    LA uint32_t* path_len = lds<uint32_t>(k.L, k.o.path_len);
    LA uint32_t* path_slot = lds<uint32_t>(k.L, k.o.path_slot);
    LA uint32_t* path_n = lds<uint32_t>(k.L, k.o.path_n);
    LA u32x4_t* sx = lds<u32x4_t>(k.L, k.o.x);
    int red_slot = 0, cmp_par = 0;
    // allocation counters and tree roots: wave-uniform registers, updated identically by every thread
    uint32_t cN, cI, c8, c16, c32, cRoot, cFirst, cDepth;
    if constexpr (ONE) {
        LA uint32_t* lc = lds<uint32_t>(k.L, k.o.ctr);
        cN = uni(lc[C_NODES]); cI = uni(lc[C_IDS]); c8 = uni(lc[C_N8]); c16 = uni(lc[C_N16]);
        c32 = uni(lc[C_N32]); cRoot = uni(lc[C_ROOT]); cFirst = uni(lc[C_FIRST_LEAF]); cDepth = uni(lc[C_DEPTH]);
    } else {
        cN = uni(T->ctr[C_NODES]); cI = uni(T->ctr[C_IDS]); c8 = uni(T->ctr[C_N8]); c16 = uni(T->ctr[C_N16]);
        c32 = uni(T->ctr[C_N32]); cRoot = SUB ? uni(gate_nodes[blockIdx.x]) : uni(T->ctr[C_ROOT]); cFirst = uni(T->ctr[C_FIRST_LEAF]);
        cDepth = uni(T->ctr[C_DEPTH]);
        if (tid < 8) stats[tid] = SUB ? 0ull : T->stats[tid];
    }
    for (int ch = tid; ch < k.RBc; ch += TB) sx[ch] = (u32x4_t)(0);  // padding bytes stay zero
    __syncthreads();
    // LDS mirrors of the nodes on the most recent root-to-leaf path, one per level: consecutive
    // insertions mostly revisit them (the upper levels always, and BitBIRCH trees on diverse data
    // route whole runs of fingerprints down the same branch).  Tags are wave-uniform registers.
    uint32_t mir_node[MAXM], mir_len[MAXM], mir_leaf[MAXM];
    bool mir_zero[MAXM];  // every centroid of the mirrored node is all-zero: similarity 0 to anything
#pragma unroll
    for (int q = 0; q < MAXM; ++q) { mir_node[q] = NONE; mir_len[q] = 0; mir_leaf[q] = 0; mir_zero[q] = false; }
    u64 ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    u64 sph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    u64 tmark = PROF ? __builtin_amdgcn_s_memtime() : 0;
#define PHASE(i) do { if constexpr (PROF) { const u64 _n = __builtin_amdgcn_s_memtime(); ph[i] += _n - tmark; tmark = _n; } } while (0)

    // software prefetch of the next fingerprint row (aligned rows of <= 4 KiB)
    const bool pf_ok = !bufmode && n_elems > 0 && ((((uintptr_t)in_rows) | (uintptr_t)row_stride) & 15) == 0 &&
                       (nb & 15) == 0 && k.RBc <= TB;
    u32x4_t pf = (u32x4_t)(0);
    long long e = ONE ? e_first : 0;
    const long long e_end = ONE ? e_first + 1 : n_elems;
    if (pf_ok && tid < k.RBc) pf = ldg<u32x4_t>(in_rows + (SUB ? (size_t)uni(gate_elems[g_off]) * (size_t)row_stride : (size_t)e * (size_t)row_stride) + (size_t)tid * 16);

    int stop = STOP_DONE;
    for (; e < e_end; ++e) {
        // Elements meet here.  When the root is mirrored in LDS the first global read of the next
        // insertion (level 1) sits behind the full barrier that ends the root compare, so this
        // barrier only has to order LDS traffic: the previous insertion's HBM stores keep draining
        // underneath the root compare instead of being waited for here.
        const bool root_hit = nm > 0 && mir_node[0] == cRoot;
        if (!root_hit) {
            __syncthreads();
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // ---- capacity for the worst case of one insertion -------------------------------
        if constexpr (!SUB) {
            const uint32_t depth = cDepth;
            if (depth + 2 >= (uint32_t)MAXD) { stop = STOP_DEPTH; break; }
            // (every level may split - a node each, and a new root - and every node of the path may have to be thawed first)
            if (cN + (2 * depth + 2) * k.nblk > cap_nodes) { stop = STOP_NODES; break; }
            if (c8 + 1 > cap8) { stop = STOP_CF8; break; }
            if (c16 + 1 > cap16) { stop = STOP_CF16; break; }
            if (c32 + 2 * (depth + 1) + 1 > cap32) { stop = STOP_CF32; break; }
        }
        uint32_t root = cRoot;
        uint32_t root_len;
        if (root_hit) root_len = mir_len[0];
        else root_len = uni(ldg<uint32_t>(k.hdr + root));
        Elem el;
        PHASE(14);
        const long long eidx = SUB ? (long long)uni(gate_elems[g_off + e]) : e;  // position in the input array
        el.idx = eidx;
        // ---- element: packed centroid into LDS, n, moments ------------------------------
        if (!bufmode) {
            const uint8_t* row = in_rows + eidx * row_stride;
            if (pf_ok) {
                // this row was requested while the previous element was being inserted
                if (tid < k.RBc) sx[tid] = pf;
                PHASE(15);
                if (e + 1 < e_end && tid < k.RBc) {
                    const long long nidx = SUB ? (long long)uni(gate_elems[g_off + e + 1]) : e + 1;
                    pf = ldg<u32x4_t>(in_rows + nidx * row_stride + (size_t)tid * 16);
                }
            } else if ((((uintptr_t)row) & 15) == 0 && (nb & 15) == 0) {
                for (int ch = tid; ch < k.RBc; ch += TB) sx[ch] = ldg<u32x4_t>(row + (size_t)ch * 16);
            } else {
                for (int b = tid; b < nb; b += TB) lds<uint8_t>(k.L, k.o.x)[b] = ldg<uint8_t>(row + b);
            }
            el.nS = 1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // orders the LDS row only
            el.pcx = lds_vec_popcount(k, k.o.x);
            el.s1S = el.pcx;
            el.s2S = el.pcx;
        } else {
            u64 nraw;
            const size_t ncol = (size_t)eidx * ((size_t)k.F + 1) + (size_t)k.F;
            switch (k.width) {
                case 1: nraw = ldg<uint8_t>(k.bufs + ncol); break;
                case 2: nraw = ldg<uint16_t>(k.bufs + 2 * ncol); break;
                case 4: nraw = ldg<uint32_t>(k.bufs + 4 * ncol); break;
                default: nraw = ldg<u64>(k.bufs + 8 * ncol); break;
            }
            nraw = uni64(nraw);
            if (nraw > 0xFFFFFFFFull) { stop = STOP_RANGE; break; }
            el.nS = (uint32_t)nraw;
            el.pcx = 0; el.s1S = 0; el.s2S = 0;
            u64 acc[2] = {0, 0};
            for (int b = tid; b < nb; b += TB) {
                uint32_t v[8];
                elem_cols(k, el, b, v);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    acc[0] += v[q];
                    acc[1] += (u64)v[q] * v[q];
                }
                lds<uint8_t>(k.L, k.o.x)[b] = (uint8_t)centroid_byte(v, el.nS);
            }
            block_sum<2>(k, red_slot, acc);  // barrier: x complete
            el.s1S = acc[0];
            el.s2S = acc[1];
            el.pcx = lds_vec_popcount(k, k.o.x);
        }
        el.nS = uni(el.nS); el.pcx = uni(el.pcx); el.s1S = uni64(el.s1S); el.s2S = uni64(el.s2S);
        PHASE(0);

        uint32_t out_id = NONE;
        bool overflow = false;
        int D = 0;  // leaf level index
        if (root_len == 0) {
            // ---- very first element of an empty tree: row 0 of the root leaf -------------
            const uint32_t tier = tier_for(el.nS);
            const bool lazy = el.nS == 1;  // one member: its cluster features are its centroid, no slot and nothing to write
            const uint32_t s = alloc_n<SUB>(k, cI, gctr + C_IDS, 1, 14);
            uint32_t slotw = SLOT_LAZY8;
            if (!lazy)
                slotw = (tier << 30) | (tier == 0 ? alloc_n<SUB>(k, c8, gctr + C_N8, 1, 15)
                                                  : (tier == 1 ? alloc_n<SUB>(k, c16, gctr + C_N16, 1, 15)
                                                               : alloc_n<SUB>(k, c32, gctr + C_N32, 1, 15)));
            if (tid == 0) {
                stg<uint32_t>(k.hdr + root, 1u);
                stats[3]++;
            }
            if (!lazy) {
                for (int b = tid; b < nb; b += TB) {
                    uint32_t v[8];
                    elem_cols(k, el, b, v);
                    cf_store8(k, slotw, b, v);
                }
            }
            node_put_row(k, root, 0, k.o.x, el.pcx, slotw, s, el.nS, slotw, el.s1S, el.s2S);
            out_id = s;
#pragma unroll
            for (int q = 0; q < MAXM; ++q) mir_node[q] = NONE;
        } else {
            // ---- greedy descent (bitbirch.py:305-357) ---------------------------------
            uint32_t nd = root, j = 0, link = NONE, len = root_len, leaf = 0;
            int depth = 0;
            u32x4_t rm0 = (u32x4_t)(0), rm1 = (u32x4_t)(0);  // RowMeta of the chosen row
            uint32_t tslot[MAXFAST], tn[MAXFAST];             // ancestors' CF slot / n_samples (uniform)
#pragma unroll
            for (int q = 0; q < MAXFAST; ++q) { tslot[q] = 0; tn[q] = 0; }
            bool bad = false;
            while (true) {
                depth = (int)uni((uint32_t)depth);  // wave-uniform by construction; stops divergence analysis from
                nd = uni(nd);                       // treating everything derived from the loop counter as per-lane
                Cand best;
                bool hit = false, zero = false, anyc = true;
#pragma unroll
                for (int q = 0; q < MAXM; ++q)
                    if (q < nm && depth == q && mir_node[q] == nd) { hit = true; len = mir_len[q]; leaf = mir_leaf[q]; zero = mir_zero[q]; }
                if (hit && zero) {
                    // every row of this node has an all-zero centroid: every similarity is 0, np.argmax
                    // returns row 0 (bitbirch.py:320).  Only the ordering of HBM traffic remains to be done.
                    __syncthreads();
                    j = 0;
                    link = uni(lds<uint32_t>(k.L, k.o.rc_link)[(uint32_t)depth * k.rows]);
                } else if (hit) {
                    best = node_best_mirror<false>(k, cmp_par, len, k.o.x, el.pcx, false, false, (uint32_t)depth, &anyc);
                    j = best.r;
                    link = uni(lds<uint32_t>(k.L, k.o.rc_link)[(uint32_t)depth * k.rows + j]);
#pragma unroll
                    for (int q = 0; q < MAXM; ++q)
                        if (depth == q) mir_zero[q] = !anyc;
                } else {
                    const bool fill = depth < nm;
                    best = node_best<false, false>(k, cmp_par, nd, -1, k.o.x, el.pcx, false, false, true, &len, &leaf,
                                                   (uint32_t)depth, fill, &anyc);
                    if (hw_cap(leaf) < k.rows) {
                        // a sealed node: nothing is written to it where it lies (bb_tree.hip, "Node storage")
                        if constexpr (SUB) { bad = true; break; }  // (concurrent gates: the host thaws the whole tree first)
                        __syncthreads();  // (path_node / path_row of the level above were written by thread 0)
                        const uint32_t pP = depth > 0 ? uni(path_node[depth - 1]) : NONE, pj = depth > 0 ? uni(path_row[depth - 1]) : 0u;
                        nd = thaw_node(k, nd, pP, pj, cN, cRoot, cFirst);
#pragma unroll
                        for (int q = 0; q < MAXM; ++q) mir_node[q] = NONE;  // (the parent's mirror holds the old child id)
                        if (depth == 0) root = nd;
                        continue;
                    }
                    leaf &= HW_LEAF;
                    j = best.r;
                    link = uni(lds<uint32_t>(k.L, k.o.link)[cmp_par * k.rows + j]);
                    if (fill) {
#pragma unroll
                        for (int q = 0; q < MAXM; ++q)
                            if (depth == q) { mir_node[q] = nd; mir_len[q] = len; mir_leaf[q] = leaf; mir_zero[q] = !anyc; }
                    }
                }
                if constexpr (PROF) {  // [8]/[9]/[10]: cycles of zero-skip / hit / miss levels, [11..13]: their counts
                    const int cls = (hit && zero) ? 0 : (hit ? 1 : 2);
                    const u64 _n = __builtin_amdgcn_s_memtime();
                    ph[8 + cls] += _n - tmark;
                    ph[11 + cls] += 1;
                }
                if (depth == 0) PHASE(6); else PHASE(7);
                if (depth > 0) {  // the previous level's RowMeta has arrived by now
                    const uint32_t ps = uni(rm0.z), pn = uni(rm0.y);
#pragma unroll
                    for (int q = 0; q < MAXFAST; ++q)
                        if (depth - 1 == q) { tslot[q] = ps; tn[q] = pn; }
                    if (tid == 0) {
                        path_slot[depth - 1] = ps;
                        path_n[depth - 1] = pn;
                    }
                }
                if (tid == 0) {
                    path_node[depth] = nd;
                    path_row[depth] = j;
                    path_len[depth] = len;
                    stats[0]++;
                    stats[1] += len;
                }
                {
                    const uint8_t* rmp = (const uint8_t*)(k.rm + (size_t)nd * NG + j);
                    rm0 = ldg<u32x4_t>(rmp);
                    rm1 = ldg<u32x4_t>(rmp + 16);
                }
                // exit conditions through readfirstlane: a loop whose exit the compiler cannot prove uniform
                // makes every value that leaves it per-lane (VGPRs, exec-mask branches) downstream
                leaf = uni(leaf);
                link = uni(link);
                if (leaf) break;
                if (depth + 2 >= MAXD || link >= cap_nodes) { bad = true; break; }  // never spin on corruption
                nd = link;
                depth++;
            }
            if (bad) { stop = STOP_DEPTH; break; }
            PHASE(1);
            D = (int)uni((uint32_t)depth);  // loop-carried: tell the compiler it is wave-uniform
            if (tid == 0 && (u64)(D + 1) > stats[6]) stats[6] = (u64)(D + 1);
            const uint32_t leafnode = uni(nd), jl = uni(j), leaflen = uni(len);
            const uint32_t slotT = uni(link);  // leaf row: link = CF slot word
            // Request the cluster features of the chosen leaf row and of every ancestor NOW, before the
            // leaf row's RowMeta (still in flight) is looked at: one memory round trip instead of two.
            const bool fast = nb <= TB;  // one byte-group (8 features) per thread
            const int b0 = tid;
            const bool act = fast && b0 < nb;
            const int DT = D < MAXFAST ? D : MAXFAST;  // ancestors handled in the fused pass
            u32x4_t rawL[2] = {(u32x4_t)(0), (u32x4_t)(0)};
            uint32_t vT[MAXFAST][8];
            const uint8_t* const crowT = k.cent + ((size_t)leafnode * NG + jl) * (size_t)k.RB;  // (a lazy BitFeature's cluster features)
            uint32_t cbyteT = 0;
            if (act) {
                // (whether the BitFeature has one member is not known yet: both its slot - reserved, possibly never
                // written - and this thread's byte of its centroid are requested)
                cf_load_raw(k, slotT, b0, rawL);
                cbyteT = ldg<uint8_t>(crowT + b0);
#pragma unroll
                for (int q = 0; q < MAXFAST; ++q)
                    if (q < DT) cf32_load8(k, tslot[q], b0, vT[q]);
            }
            const uint32_t Tsub = uni(rm0.x);
            const u64 nT = uni(rm0.y);
            const u64 s1T = ((u64)uni(rm1.y) << 32) | uni(rm1.x);
            const u64 s2T = ((u64)uni(rm1.w) << 32) | uni(rm1.z);
            const u64 new_n = nT + el.nS;
            if (new_n > 0xFFFFFFFFull) { stop = STOP_RANGE; break; }
            const uint32_t slotE = slot_effective(slotT, nT);  // (one member: the centroid row is the linear sum)
            if (nT == 1) { rawL[0] = (u32x4_t)(0); rawL[0].x = cbyteT; }
            // ---- one fused pass: leaf dot product, speculative merged CF + centroid, and
            //      every ancestor's CF += element with its new centroid; one reduction ----
            u64 dot = 0;
            uint32_t pcs[1 + MAXFAST];  // [0] merged leaf centroid popcount, [1+q] ancestor q's
#pragma unroll
            for (int i = 0; i < 1 + MAXFAST; ++i) pcs[i] = 0;
            uint32_t xs[8], vL[8], byteL = 0, byteT[MAXFAST];
            const bool wide_dot = bufmode || (slotT >> 30) == 2;
#pragma unroll
            for (int q = 0; q < MAXFAST; ++q) byteT[q] = 0;
            if (fast) {
                if (act) {
                    elem_cols(k, el, b0, xs);
                    cf_unpack_raw(slotE >> 30, rawL, vL);
                    if (wide_dot) {
#pragma unroll
                        for (int q = 0; q < 8; ++q) dot += (u64)vL[q] * xs[q];
                    } else {
                        uint32_t d32 = 0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) d32 += vL[q] * xs[q];
                        dot = d32;
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) vL[q] += xs[q];
                    byteL = centroid_byte_merged(vL, new_n);
                    pcs[0] = __popc(byteL);
#pragma unroll
                    for (int q = 0; q < MAXFAST; ++q) {
                        if (q < DT) {
#pragma unroll
                            for (int z = 0; z < 8; ++z) vT[q][z] += xs[z];
                            byteT[q] = centroid_byte_merged(vT[q], (u64)tn[q] + el.nS);
                            pcs[1 + q] = __popc(byteT[q]);
                        }
                    }
                }
            } else {
                for (int b = tid; b < nb; b += TB) {
                    uint32_t v[8], x8[8];
                    cf_load8(k, slotE, b, v, crowT);
                    elem_cols(k, el, b, x8);
#pragma unroll
                    for (int q = 0; q < 8; ++q) dot += (u64)v[q] * x8[q];
                }
            }
            PHASE(2);
            block_sum_fused(k, red_slot, wide_dot || !fast, dot, pcs, fast ? 1 + DT : 0);
            PHASE(3);
            const u64 s1n = s1T + el.s1S;
            const u64 s2n = s2T + 2ull * dot + el.s2S;
            const bool accept = merge_accept(k, el, red_slot, slotE, crowT, nT, s1T, s2T, new_n, s1n, s2n);
            const size_t leafm = (size_t)leafnode * NG;
            if (accept) {
                // replace_n_samples_and_linear_sum (bitbirch.py:476-484)
                const uint32_t old_tier = slotT >> 30;
                const uint32_t new_tier = tier_for(new_n) > old_tier ? tier_for(new_n) : old_tier;
                uint32_t slotN = slotT;
                if (new_tier != old_tier) {
                    slotN = (new_tier << 30) | (new_tier == 1 ? alloc_n<SUB>(k, c16, gctr + C_N16, 1, 15)
                                                              : alloc_n<SUB>(k, c32, gctr + C_N32, 1, 15));
                } else if (nT == 1) {
                    slotN = alloc_n<SUB>(k, c8, gctr + C_N8, 1, 15);  // the second member: the BitFeature gets its uint8 row (SLOT_LAZY8)
                }
                uint8_t* crow = k.cent + (leafm + jl) * (size_t)k.RB;
                u64 card = pcs[0];
                const bool leaf_mirrored = D < nm;  // the leaf node sits in LDS mirror D
                if (fast) {
                    if (act) {
                        cf_store8(k, slotN, b0, vL);
                        stg<uint8_t>(crow + b0, (uint8_t)byteL);
                        if (leaf_mirrored)
                            *(LA uint8_t*)(k.L + k.o.rc_cent + ((uint32_t)D * k.rows + jl) * k.RBS + b0) = (uint8_t)byteL;
                    }
                } else {
                    u64 cc[1] = {0};
                    for (int b = tid; b < nb; b += TB) {
                        uint32_t v[8], x8[8];
                        cf_load8(k, slotE, b, v, crowT);
                        elem_cols(k, el, b, x8);
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += x8[q];
                        cf_store8(k, slotN, b, v);
                        const uint32_t byte = centroid_byte(v, new_n);
                        stg<uint8_t>(crow + b, (uint8_t)byte);
                        cc[0] += __popc(byte);
                    }
                    block_sum<1>(k, red_slot, cc);
                    card = cc[0];
                }
                if (tid == 0) {
                    u32x4_t a, b;
                    a.x = Tsub; a.y = (uint32_t)new_n; a.z = slotN; a.w = 0;
                    b.x = (uint32_t)s1n; b.y = (uint32_t)(s1n >> 32); b.z = (uint32_t)s2n; b.w = (uint32_t)(s2n >> 32);
                    stg<u32x4_t>((uint8_t*)(k.rm + leafm + jl), a);
                    stg<u32x4_t>((uint8_t*)(k.rm + leafm + jl) + 16, b);
                    stg<uint32_t>(k.card + leafm + jl, (uint32_t)card);
                    if (slotN != slotT) stg<uint32_t>(k.link + leafm + jl, slotN);
                    if (leaf_mirrored) {
                        lds<uint32_t>(k.L, k.o.rc_card)[(uint32_t)D * k.rows + jl] = (uint32_t)card;
                        lds<uint32_t>(k.L, k.o.rc_link)[(uint32_t)D * k.rows + jl] = slotN;
                    }
                    stats[2]++;
                }
                if (leaf_mirrored && card != 0) {
#pragma unroll
                    for (int q = 0; q < MAXM; ++q)
                        if (D == q) mir_zero[q] = false;
                }
                out_id = Tsub;
            } else {
                // append_subcluster (bitbirch.py:284-287): new leaf BitFeature
                const uint32_t tier = tier_for(el.nS);
                const bool lazy = el.nS == 1;  // (one member: its cluster features are its centroid, no slot and nothing to write)
                const uint32_t s = alloc_n<SUB>(k, cI, gctr + C_IDS, 1, 14);
                uint32_t slotw = SLOT_LAZY8;
                if (!lazy)
                    slotw = (tier << 30) | (tier == 0 ? alloc_n<SUB>(k, c8, gctr + C_N8, 1, 15)
                                                      : (tier == 1 ? alloc_n<SUB>(k, c16, gctr + C_N16, 1, 15)
                                                                   : alloc_n<SUB>(k, c32, gctr + C_N32, 1, 15)));
                if (tid == 0) {
                    stg<uint32_t>(k.hdr + leafnode, leaflen + 1);
                    stats[3]++;
                }
                if (lazy) {
                } else if (fast) {
                    if (act) cf_store8(k, slotw, b0, xs);
                } else {
                    for (int b = tid; b < nb; b += TB) {
                        uint32_t v[8];
                        elem_cols(k, el, b, v);
                        cf_store8(k, slotw, b, v);
                    }
                }
                node_put_row(k, leafnode, leaflen, k.o.x, el.pcx, slotw, s, el.nS, slotw, el.s1S, el.s2S);
                if (D < nm) {  // the new row also goes into the leaf's LDS mirror
                    const uint32_t mr = (uint32_t)D * k.rows + leaflen;
                    if (tid < k.RBc) *(LA u32x4_t*)(k.L + k.o.rc_cent + mr * k.RBS + tid * 16) = sx[tid];
                    if (tid == 0) {
                        lds<uint32_t>(k.L, k.o.rc_card)[mr] = el.pcx;
                        lds<uint32_t>(k.L, k.o.rc_link)[mr] = slotw;
                    }
#pragma unroll
                    for (int q = 0; q < MAXM; ++q)
                        if (D == q) { mir_len[q] = leaflen + 1; mir_zero[q] = mir_zero[q] && el.pcx == 0; }
                }
                out_id = s;
                overflow = leaflen + 1 > bf;
            }
            // ---- ancestors (closest_subcluster.update, bitbirch.py:352-357) ----------------
            if (!overflow) {
                if (fast) {
#pragma unroll
                    for (int q = 0; q < MAXFAST; ++q) {
                        if (q < DT) {
                            const uint32_t P = uni(path_node[q]), jp = uni(path_row[q]);
                            const size_t pm = (size_t)P * NG + jp;
                            const u64 n_new = (u64)tn[q] + el.nS;
                            if (n_new > 0xFFFFFFFFull) stop = STOP_RANGE;
                            if (act) {
                                cf_store8(k, tslot[q], b0, vT[q]);
                                stg<uint8_t>(k.cent + pm * (size_t)k.RB + b0, (uint8_t)byteT[q]);
                                if (q < MAXM && q < nm) *(LA uint8_t*)(k.L + k.o.rc_cent + ((uint32_t)q * k.rows + jp) * k.RBS + b0) = (uint8_t)byteT[q];
                            }
                            if (tid == 0) {
                                stg<uint32_t>((uint8_t*)(k.rm + pm) + 4, (uint32_t)n_new);
                                stg<uint32_t>((uint8_t*)(k.rm + pm) + 12, 0u);  // flip distance now stale
                                stg<uint32_t>(k.card + pm, pcs[1 + q]);
                                if (q < MAXM && q < nm) lds<uint32_t>(k.L, k.o.rc_card)[(uint32_t)q * k.rows + jp] = pcs[1 + q];
                            }
                            if (q < MAXM && pcs[1 + q] != 0) mir_zero[q] = false;
                        }
                    }
                }
                for (int lv = fast ? DT : 0; lv < D; ++lv) {  // deep trees / wide rows: one level at a time
                    update_tracker_slow(k, el, red_slot, lv, stop);
#pragma unroll
                    for (int q = 0; q < MAXM; ++q)
                        if (lv == q) mir_node[q] = NONE;  // changed behind the mirror's back
                }
            }
        }
        PHASE(4);
        // ---- overflow: split upward (bitbirch.py:339-350, :778-782) ------------------------
        if (overflow) {
            __syncthreads();
            int lvl = D;
            int upd_levels = 0;
            bool range_bad = false;
            while (true) {
                const uint32_t nd = uni(path_node[lvl]);
                uint32_t ms = 0;
                bool ms_valid = false;
                if (nm > 0) {
                    ms = lvl < nm ? (uint32_t)lvl : (uint32_t)(nm - 1);
#pragma unroll
                    for (int q = 0; q < MAXM; ++q)
                        if (q == lvl && lvl == D && q < nm && mir_node[q] == nd) ms_valid = true;  // leaf mirror, appended row included
                }
                const uint32_t trk = lvl > 0 ? uni(path_slot[lvl - 1]) : NONE;
                split_node<SUB, PROF>(k, el, red_slot, cmp_par, nd, ms, ms_valid, trk, cN, c32, cFirst, gctr, sph);
#pragma unroll
                for (int q = 0; q < MAXM; ++q) mir_node[q] = NONE;  // slot `ms` was used as workspace
                const uint32_t node1 = uni(bc[0]), cardA = uni(bc[3]), cardB = uni(bc[4]);
                const uint32_t nA = uni(bc[7]), nB = uni(bc[8]);
                const uint32_t slotA = (2u << 30) | uni(bc[9]), slotB = (2u << 30) | uni(bc[10]);
                if (uni(bc[11])) { range_bad = true; break; }
                if (SUB && lvl == 0) {  // a gate may never split inside a concurrent batch
                    range_bad = true;
                    stop = STOP_GATE;
                    break;
                }
                if (lvl == 0) {
                    // root split: new root holding the two tracking BitFeatures
                    const uint32_t nr = cN;
                    cN += k.nblk;
                    cRoot = nr;
                    cDepth++;
                    node_put_row(k, nr, 0, k.o.cA, cardA, node1, NONE, nA, slotA, 0, 0);
                    node_put_row(k, nr, 1, k.o.cB, cardB, nd, NONE, nB, slotB, 0, 0);
                    if (tid == 0) {
                        u32x4_t h;
                        h.x = 2; h.y = hw_make(0u, k.rows); h.z = NONE; h.w = NONE;
                        stg<u32x4_t>(k.hdr + nr, h);
                        stats[5]++;
                    }
                    upd_levels = 0;
                    break;
                }
                // update_split_subclusters (bitbirch.py:289-303)
                const uint32_t P = uni(path_node[lvl - 1]), jp = uni(path_row[lvl - 1]), lenP = uni(path_len[lvl - 1]);
                node_put_row(k, P, jp, k.o.cA, cardA, node1, NONE, nA, slotA, 0, 0);
                node_put_row(k, P, lenP, k.o.cB, cardB, nd, NONE, nB, slotB, 0, 0);
                if (tid == 0) stg<uint32_t>(k.hdr + P, lenP + 1);
                __syncthreads();
                if (lenP + 1 > bf) {
                    lvl--;
                    continue;
                }
                upd_levels = lvl - 1;
                break;
            }
            if (range_bad) { if (stop == STOP_DONE) stop = STOP_RANGE; break; }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MAXM; ++q) mir_node[q] = NONE;  // the path was restructured
            for (int lv = 0; lv < upd_levels; ++lv) update_tracker_slow(k, el, red_slot, lv, stop);
        }
        PHASE(5);
        stop = (int)uni((uint32_t)stop);
        if (stop != STOP_DONE) break;
        if (tid == 0 && out_leaf) stg<uint32_t>(out_leaf + eidx, out_id);
    }
    __syncthreads();
    if constexpr (SUB) {
        // concurrent gates of one tree: counters were bumped atomically; fold the statistics in
        if (tid < 7 && tid != 5 && tid != 6) atomicAdd((unsigned long long*)&T->stats[tid], (unsigned long long)stats[tid] );
        if (tid == 5) atomicAdd((unsigned long long*)&T->stats[5], (unsigned long long)stats[5]);
        if (tid == 0 && stop != STOP_DONE) atomicMax(&T->stop_reason, stop);
    } else if constexpr (ONE) {
        if (tid == 0) {
            LA uint32_t* lc = lds<uint32_t>(k.L, k.o.ctr);
            lc[C_NODES] = cN; lc[C_IDS] = cI; lc[C_N8] = c8; lc[C_N16] = c16; lc[C_N32] = c32;
            lc[C_ROOT] = cRoot; lc[C_FIRST_LEAF] = cFirst; lc[C_DEPTH] = cDepth;
            lc[8] = (uint32_t)stop;
            lc[9] = (uint32_t)(e - e_first);
        }
        __syncthreads();
    } else {
        if (tid == 0) {
            T->ctr[C_NODES] = cN; T->ctr[C_IDS] = cI; T->ctr[C_N8] = c8; T->ctr[C_N16] = c16; T->ctr[C_N32] = c32;
            T->ctr[C_ROOT] = cRoot; T->ctr[C_FIRST_LEAF] = cFirst; T->ctr[C_DEPTH] = cDepth;
        }
        if (tid < 8) T->stats[tid] = stats[tid];
        if (tid == 0) {
            T->processed = e;
            T->stop_reason = stop;
            if constexpr (PROF) {
                for (int i = 0; i < 16; ++i) T->phase[i] += ph[i];
                for (int i = 0; i < 8; ++i) T->sphase[i] += sph[i];
            }
        }
    }
#undef PHASE
}

// A single tree (or a few) wants the whole register file: one wave per SIMD, no spills.
template <bool PROF, bool SUB, class KCt = KC>
__global__ __launch_bounds__(TB) void k_tree_insert(TreeDev* trees, const uint32_t* gate_nodes, const uint32_t* gate_off,
                                                    const uint32_t* gate_elems) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    tree_insert_body<PROF, SUB, KCt>(smem_raw, trees, gate_nodes, gate_off, gate_elems);
}

// More trees than compute units: two workgroups per CU (<= 256 VGPRs, 2 waves per SIMD) hide each
// other's memory and barrier latency.
template <class KCt = KC>
__global__ __launch_bounds__(TB, 2) void k_tree_insert_dense(TreeDev* trees) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    tree_insert_body<false, false, KCt>(smem_raw, trees, nullptr, nullptr, nullptr);
}

#endif  // __HIPCC__ (bb_tree_fast.inc has its own host / device split)
#include "bb_tree_fast.inc"
#include "bb_tree_pipe.inc"
#include "bb_tree_sys.inc"
#if defined(__HIPCC__)

// the complete engine compiled for the benchmark shape with phase timers (tools/phases.py with BBHIP_NO_FAST=1)
using KC50 = KCFix<50, 2048>;

// the steady-state kernel (bb_tree_fast.inc) for the shapes that have one
using KF50P = KF<50, 0, -1>;
using KF50B = KF<50, 1, -1>;
using KF254P = KF<254, 0, -1>;
using KF254B = KF<254, 1, -1>;
// ... with the criterion of the standard pipelines as a compile-time constant (the decision becomes straight-line
// code): diameter for `fit`, tolerance-diameter for refinement / merge rounds (packed singleton runs and buffers)
using KF50PD = KF<50, 0, BBH_CRIT_DIAMETER>;
using KF50PT = KF<50, 0, BBH_CRIT_TOL_DIAMETER>;
using KF50BT = KF<50, 1, BBH_CRIT_TOL_DIAMETER>;
using KF254PD = KF<254, 0, BBH_CRIT_DIAMETER>;
using KF254PT = KF<254, 0, BBH_CRIT_TOL_DIAMETER>;
using KF254BT = KF<254, 1, BBH_CRIT_TOL_DIAMETER>;

// Kernel instances by (branching factor, element kind): one table instead of a ladder of launch statements.  Shapes
// without an entry run the complete engine with the shape as run-time values (k_tree_insert<.., KC>).
struct FastKernel {
    int bf;
    bool buffers;
    int crit;  // -1: any of the criteria the steady-state kernel knows, read at run time
    void (*fn)(TreeDev*);
    uint32_t lds;
};
static const FastKernel kFastKernels[] = {  // (most specific first)
    {50, false, BBH_CRIT_DIAMETER, k_tree_fast<KF50PD>, fast_layout(50).total},
    {50, false, BBH_CRIT_TOL_DIAMETER, k_tree_fast<KF50PT>, fast_layout(50).total},
    {50, true, BBH_CRIT_TOL_DIAMETER, k_tree_fast<KF50BT>, fast_layout(50).total},
    {254, false, BBH_CRIT_DIAMETER, k_tree_fast<KF254PD>, fast_layout(254).total},
    {254, false, BBH_CRIT_TOL_DIAMETER, k_tree_fast<KF254PT>, fast_layout(254).total},
    {254, true, BBH_CRIT_TOL_DIAMETER, k_tree_fast<KF254BT>, fast_layout(254).total},
    {50, false, -1, k_tree_fast<KF50P>, fast_layout(50).total},
    {50, true, -1, k_tree_fast<KF50B>, fast_layout(50).total},
    {254, false, -1, k_tree_fast<KF254P>, fast_layout(254).total},
    {254, true, -1, k_tree_fast<KF254B>, fast_layout(254).total},
};

// the pipelined kernel (bb_tree_pipe.inc): packed fingerprints, one tree per launch, diameter / tolerance-diameter
struct PipeKernel {
    int bf, crit, ml;  // ml: the multi-level instance (several exact internal levels, pipe_router_ml)
    void (*fn)(TreeDev*);
    void (*fn_prof)(TreeDev*);  // the phase-timer / run-end-audit instance (BBHIP_PIPE_PHASES, BBHIP_PIPE_AUDIT): every entry has one
    uint32_t lds;
};
#define BB_PIPE_ENTRY(BF, CRIT, ML) {BF, CRIT, ML, k_tree_pipe<KP<BF, CRIT, ML>>, k_tree_pipe<KP<BF, CRIT, ML>, true>, pipe_layout(BF, ML).total}
static const PipeKernel kPipeKernels[] = {
    BB_PIPE_ENTRY(50, BBH_CRIT_DIAMETER, 0),
    BB_PIPE_ENTRY(50, BBH_CRIT_TOL_DIAMETER, 0),
    BB_PIPE_ENTRY(254, BBH_CRIT_DIAMETER, 0),
    BB_PIPE_ENTRY(254, BBH_CRIT_TOL_DIAMETER, 0),
    BB_PIPE_ENTRY(50, BBH_CRIT_DIAMETER, 1),
    BB_PIPE_ENTRY(50, BBH_CRIT_TOL_DIAMETER, 1),
};
#undef BB_PIPE_ENTRY

// uint8 BitFeature buffers with n_samples == 1 are plain fingerprints in unpacked form (ls in {0, 1}):
// pack them (MSB first, np.packbits order) so that they take the fingerprint path of the kernel.
__global__ __launch_bounds__(256) void k_pack_singletons(const uint8_t* __restrict__ bufs, long long k, int F, int nbytes,
                                                          uint8_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k * nbytes) return;
    const long long r = i / nbytes;
    const int b = (int)(i % nbytes);
    const uint8_t* src = bufs + (size_t)r * ((size_t)F + 1) + (size_t)b * 8;
    const u32x2_t q = *(const GA u32x2_t*)src;  // unaligned 8-byte load
    uint32_t byte = 0;
#pragma unroll
    for (int z = 0; z < 4; ++z) {
        byte |= (((q.x >> (8 * z)) & 0xFFu) != 0 ? 1u : 0u) << (7 - z);
        byte |= (((q.y >> (8 * z)) & 0xFFu) != 0 ? 1u : 0u) << (3 - z);
    }
    out[i] = (uint8_t)byte;
}

// the n_samples column of a uint8 BitFeature table, contiguous (the host decides on it where singleton runs start)
__global__ __launch_bounds__(256) void k_gather_n_col(const uint8_t* __restrict__ bufs, long long k, int F, uint8_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) out[i] = bufs[(size_t)i * ((size_t)F + 1) + (size_t)F];
}

// =======================================================================================
// Batch mode (exact, rollback-free): a prefix of the pending fingerprints is routed through the
// STABLE upper levels of the tree in parallel (k_route), the host admits the longest prefix for
// which (G1) no tracking centroid of a stable node can flip (flip distance, k_upd) and (G2) no
// gate (leaf-parent node) can overflow, then every gate inserts its own elements sequentially
// (k_tree_insert<.., SUB>), all gates concurrently, and the stable trackers receive their
// commutative cluster-feature sums (k_upd).  See DESIGN.md section 6b for the argument.
// =======================================================================================
struct RouteRec {
    uint32_t gate, gate_len, leaf, leaf_len, sumlen, pad;
    uint32_t node[4], row[4], fd[4];
};

__device__ __forceinline__ KC make_kc(TreeDev* T, unsigned char* smem_raw, int nm) {
    KC k;
    k.cent = T->node_cent; k.card = T->node_card; k.link = T->node_link; k.rm = T->node_rm; k.hdr = T->node_hdr;
    k.scratch = T->scratch_cent; k.cf8 = T->cf8; k.cf16 = T->cf16; k.cf32 = T->cf32;
    k.bufs = nullptr; k.width = 0;
    k.F = (int)uni((uint32_t)T->F); k.nb = (int)uni((uint32_t)T->nbytes); k.RB = (int)uni((uint32_t)T->RB);
    k.RBc = k.RB / 16; k.RBS = k.RB + 16;
    k.bf = uni((uint32_t)T->bf); k.rows = k.bf + 1; k.nblk = node_blocks(k.rows);
    k.crit = 0; k.tol_len = 0; k.thr = 0; k.tolerance = 0; k.tol = nullptr;
    k.nm = nm;
    k.L = (LA unsigned char*)smem_raw;
    k.o = smem_layout((int)k.bf, k.RB, nm);
    return k;
}

// one workgroup per pending fingerprint: greedy descent through G stable levels + the gate
__global__ __launch_bounds__(TB) void k_route(TreeDev* T, long long first_idx, uint32_t G, RouteRec* out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    const KC k = make_kc(T, smem_raw, 0);
    const int tid = threadIdx.x;
    const long long idx = first_idx + blockIdx.x;
    const uint8_t* row = T->rows + idx * T->row_stride;
    LA u32x4_t* sx = lds<u32x4_t>(k.L, k.o.x);
    for (int ch = tid; ch < k.RBc; ch += TB) sx[ch] = (u32x4_t)(0);
    __syncthreads();
    for (int b = tid; b < k.nb; b += TB) lds<uint8_t>(k.L, k.o.x)[b] = ldg<uint8_t>(row + b);
    __syncthreads();
    const uint32_t pcx = lds_vec_popcount(k, k.o.x);
    int cmp_par = 0;
    uint32_t nd = uni(T->ctr[C_ROOT]);
    uint32_t sumlen = 0;
    RouteRec rec;
    rec.pad = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l) { rec.node[l] = NONE; rec.row[l] = 0; rec.fd[l] = 0; }
    for (uint32_t l = 0; l <= G; ++l) {
        uint32_t len = 0, leaf = 0;
        const Cand best = node_best<false, false>(k, cmp_par, nd, -1, k.o.x, pcx, false, false, true, &len, &leaf);
        const uint32_t link = uni(lds<uint32_t>(k.L, k.o.link)[cmp_par * k.rows + best.r]);
        if (l < G) {
            const uint32_t fd = uni(ldg<uint32_t>((const uint8_t*)(k.rm + (size_t)nd * NG + best.r) + 12));
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if ((uint32_t)q == l) { rec.node[q] = nd; rec.row[q] = best.r; rec.fd[q] = fd; }
            sumlen += len;
            nd = link;
        } else {
            rec.gate = nd;
            rec.gate_len = len;
            rec.leaf = link;
            rec.leaf_len = uni(ldg<uint32_t>(k.hdr + link));
        }
    }
    rec.sumlen = sumlen;
    if (tid == 0) out[blockIdx.x] = rec;
}

// one workgroup per stable tracking row: CF += sum of the admitted fingerprints routed through
// it (commutative), n += count, new centroid / popcount, and the row's flip distance
//   fd = min over features of (2*ls >= n ? 2*ls - n + 1 : n - 2*ls)
// = the smallest number of further single-fingerprint additions that could change any centroid
// bit.  count == 0 just (re)computes fd.
__global__ __launch_bounds__(TB) void k_upd(TreeDev* T, const uint32_t* u_node, const uint32_t* u_row,
                                            const uint32_t* u_off, const uint32_t* u_elems, uint32_t* out_fd) {
    __shared__ unsigned long long s_card[TW];
    __shared__ uint32_t s_fd[TW];
    const int tid = threadIdx.x;
    const uint32_t F = (uint32_t)T->F, nb = (uint32_t)T->nbytes, RB = (uint32_t)T->RB;
    const uint32_t nd = u_node[blockIdx.x], r = u_row[blockIdx.x];
    const uint32_t e0 = u_off[blockIdx.x], e1 = u_off[blockIdx.x + 1];
    const size_t m = (size_t)nd * NG + r;
    RowMeta* rm = T->node_rm + m;
    const uint32_t slot = rm->slot & 0x3FFFFFFFu;
    const unsigned long long n_new = (unsigned long long)rm->n + (e1 - e0);
    uint32_t* cf = T->cf32 + (size_t)slot * F;
    const uint8_t* in_rows = T->rows;
    const long long stride = T->row_stride;
    unsigned long long card = 0;
    uint32_t fd = 0xFFFFFFFEu;
    for (uint32_t b = tid; b < nb; b += TB) {
        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t e = e0; e < e1; ++e) {
            const uint32_t v = in_rows[(long long)u_elems[e] * stride + b];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += (v >> (7 - q)) & 1u;
        }
        uint32_t byte = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const unsigned long long v = (unsigned long long)cf[b * 8 + q] + acc[q];
            cf[b * 8 + q] = (uint32_t)v;
            const bool bit = n_new <= 1 ? ((v & 0xFF) != 0) : (2ull * v >= n_new);
            byte |= (bit ? 1u : 0u) << (7 - q);
            const unsigned long long d = 2ull * v >= n_new ? 2ull * v - n_new + 1ull : n_new - 2ull * v;
            const uint32_t dd = d > 0xFFFFFFFEull ? 0xFFFFFFFEu : (uint32_t)d;
            fd = dd < fd ? dd : fd;
        }
        T->node_cent[m * RB + b] = (uint8_t)byte;
        card += __popc(byte);
    }
    for (int o = 32; o >= 1; o >>= 1) {
        card += (unsigned long long)__shfl_xor((long long)card, o);
        const uint32_t of = (uint32_t)__shfl_xor((int)fd, o);
        fd = of < fd ? of : fd;
    }
    if ((tid & 63) == 0) { s_card[tid >> 6] = card; s_fd[tid >> 6] = fd; }
    __syncthreads();
    if (tid == 0) {
        unsigned long long c = 0;
        uint32_t f = 0xFFFFFFFEu;
        for (int w = 0; w < TW; ++w) { c += s_card[w]; f = s_fd[w] < f ? s_fd[w] : f; }
        if (n_new > 0xFFFFFFFFull) atomicMax(&T->stop_reason, (int)STOP_RANGE);
        rm->n = (uint32_t)n_new;
        rm->pad = f;
        T->node_card[m] = (uint32_t)c;
        if (out_fd) out_fd[blockIdx.x] = f;
    }
}

// ---- extraction ---------------------------------------------------------------------------
// one workgroup per requested leaf row: BitFeature buffer [linear_sum | n] at `width` bytes
__global__ __launch_bounds__(256) void k_gather_leaves(TreeDev* Tp, const uint32_t* nodes, const uint32_t* rowsidx,
                                                       long long m, int width, uint8_t* out_bufs, int ls_only,
                                                       uint8_t* out_cent, unsigned long long* out_n,
                                                       uint32_t* out_ids) {
    const TreeDev t = *Tp;
    const long long i = blockIdx.x;
    if (i >= m) return;
    const uint32_t nd = nodes[i], r = rowsidx[i];
    const RowMeta rm = t.node_rm[(size_t)nd * NG + r];
    const uint32_t n = rm.n;
    if (threadIdx.x == 0) {
        if (out_n) out_n[i] = n;
        if (out_ids) out_ids[i] = rm.sub;
    }
    if (out_cent) {
        const uint8_t* src = t.node_cent + ((size_t)nd * NG + r) * (size_t)t.RB;
        for (int b = threadIdx.x; b < t.nbytes; b += blockDim.x) out_cent[(size_t)i * t.nbytes + b] = src[b];
    }
    if (out_bufs) {
        const uint32_t slotw = rm.slot;
        const uint32_t tier = slotw >> 30;
        const size_t base = (size_t)(slotw & 0x3FFFFFFFu) * (size_t)t.F;
        const size_t cols = (size_t)t.F + (ls_only ? 0 : 1);
        for (int j = threadIdx.x; j < t.F + (ls_only ? 0 : 1); j += blockDim.x) {
            unsigned long long v;
            if (j == t.F) v = n;
            else if (n == 1) v = (t.node_cent[((size_t)nd * NG + r) * (size_t)t.RB + (size_t)(j >> 3)] >> (7 - (j & 7))) & 1u;
            else v = tier == 0 ? t.cf8[base + j] : (tier == 1 ? t.cf16[base + j] : t.cf32[base + j]);
            const size_t o = (size_t)i * cols + j;
            switch (width) {
                case 1: out_bufs[o] = (uint8_t)v; break;
                case 2: reinterpret_cast<uint16_t*>(out_bufs)[o] = (uint16_t)v; break;
                case 4: reinterpret_cast<uint32_t*>(out_bufs)[o] = (uint32_t)v; break;
                default: reinterpret_cast<unsigned long long*>(out_bufs)[o] = v; break;
            }
        }
    }
}

// {id, len, next} of every header that starts a live leaf (build_chain); out == nullptr: count only
__global__ __launch_bounds__(256) void k_leaf_headers(const NodeHdr* __restrict__ hdr, uint32_t used, uint32_t* count, uint32_t* out, uint32_t cap) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= used) return;
    const NodeHdr h = hdr[b];
    if (hw_cap(h.leaf) == 0 || !(h.leaf & HW_LEAF)) return;
    const uint32_t i = atomicAdd(count, 1u);
    if (out != nullptr && i < cap) {
        out[3 * (size_t)i] = b;
        out[3 * (size_t)i + 1] = h.len;
        out[3 * (size_t)i + 2] = h.next;
    }
}

// ---- compaction of the node pools (gc_nodes, host side; "Node storage" at the top of this file) -------------------
// Pass 1, one thread per block of the used part of the pool: how many blocks the node that starts here gets in the new
// pool - 0 for a block that starts no live node; its length rounded up to a block for a node that is sealed already or
// whose length is what it was at the previous compaction (`seal`; never the root); bf + 1 rows otherwise.
__global__ __launch_bounds__(256) void k_gc_size(const NodeHdr* __restrict__ hdr, uint32_t used, uint32_t rows, uint32_t root, int seal,
                                                 uint32_t* __restrict__ sz, unsigned long long* __restrict__ counts) {
    const uint32_t b = blockIdx.x * 256u + threadIdx.x;
    if (b >= used) return;
    const NodeHdr h = hdr[b];
    const uint32_t cap = hw_cap(h.leaf);
    uint32_t blocks = 0;
    if (cap != 0) {
        const bool cold = seal != 0 && b != root && (cap < rows || hw_gcl(h.leaf) == h.len + 1);
        const uint32_t len1 = h.len > 0 ? h.len : 1u;
        blocks = cold ? node_blocks(len1) : node_blocks(rows);
        atomicAdd(counts + (cold ? 1 : 0), 1ull);  // [0] nodes kept at full capacity, [1] sealed nodes
    }
    sz[b] = blocks;
}

// Pass 2 (after the exclusive prefix sum of the sizes = the new ids), one wave per live node, grid-stride: rows, per-row words
// and records move to the node's new place in the NEW pools; child ids of internal rows and the leaf chain's ids are
// translated; the header records the node's length (the next compaction compares with it) and its new capacity.
__global__ __launch_bounds__(256) void k_gc_move(TreeDev old_t, TreeDev new_t, uint32_t used, uint32_t rows, const uint32_t* __restrict__ sz,
                                                 const uint32_t* __restrict__ newid) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t waves = gridDim.x * 4u;
    for (uint32_t b = blockIdx.x * 4u + (threadIdx.x >> 6); b < used; b += waves) {
        const uint32_t blocks = sz[b];
        if (blocks == 0) continue;
        const NodeHdr h = old_t.node_hdr[b];
        const uint32_t nb_ = newid[b], len = h.len;
        const bool leaf = (h.leaf & HW_LEAF) != 0;
        const size_t so = (size_t)b * NG, dn = (size_t)nb_ * NG;
        const uint32_t rbc = (uint32_t)old_t.RB / 16;
        const uint4* src = (const uint4*)(old_t.node_cent + so * (size_t)old_t.RB);
        uint4* dst = (uint4*)(new_t.node_cent + dn * (size_t)old_t.RB);
        for (uint32_t i = lane; i < len * rbc; i += 64) dst[i] = src[i];
        for (uint32_t r = lane; r < len; r += 64) {
            new_t.node_card[dn + r] = old_t.node_card[so + r];
            const uint32_t lk = old_t.node_link[so + r];
            new_t.node_link[dn + r] = leaf ? lk : (lk < used ? newid[lk] : lk);
            new_t.node_rm[dn + r] = old_t.node_rm[so + r];
        }
        if (lane == 0) {
            NodeHdr hn;
            hn.len = len;
            const uint32_t cap = blocks == node_blocks(rows) ? rows : blocks * NG;
            hn.leaf = (h.leaf & HW_LEAF) | (((len + 1) & 0xFFFu) << 4) | (cap << 16);
            hn.prev = (h.prev != NONE && h.prev < used) ? newid[h.prev] : NONE;
            hn.next = (h.next != NONE && h.next < used) ? newid[h.next] : NONE;
            new_t.node_hdr[nb_] = hn;
        }
    }
}

#endif  // __HIPCC__

}  // namespace

// -------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------
struct bbh_tree {
    TreeDev h{};
    TreeDev* d = nullptr;
    double* d_tol = nullptr;
    int device = 0;
    size_t lds = 0;
    bool chain_valid = false;
    std::vector<uint32_t> chain_nodes, chain_rows;
    uint32_t *d_chain_nodes = nullptr, *d_chain_rows = nullptr;
    size_t d_chain_cap = 0;
    int64_t unsup_stretch = 0;  // elements the steady-state kernel took after the pipelined one last refused the tree's shape
    // which kernel inserted what (bbh_tree_kernel_counts): elements and launches by {pipelined, steady-state, complete}, then
    // the launches the pipelined kernel ended with STOP_PIPE_UNSUPPORTED and the pool-exhaustion stops (STOP_NODES / STOP_CF*)
    uint64_t kcount[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool pipe_ml = false;  // the pipelined kernel asked for its multi-level instance (informative levels above the leaf-parents)
    // node-pool compactions (gc_nodes): how many ran, nodes they sealed / left at full capacity (last one), blocks before / after (last one)
    uint64_t gc_runs = 0, gc_sealed = 0, gc_full = 0, gc_before = 0, gc_after = 0;
    bool lazy_pools = false;  // a fresh tree owns no pools yet: the first call that inserts allocates them at the size it needs (pregrow)
    bool no_seal = false;  // (while the exact batch mode runs: compactions leave every node at full capacity)
    size_t peak_bytes = 0;  // largest sum of this tree's pool allocations (both copies of a pool that is being regrown included)
    // the level-systolic kernel (bb_tree_sys.inc): its work area in HBM (rings, mailboxes), how many blocks the mailbox arrays hold,
    // and what it did: elements, launches, relaunches after a root split, workgroups of the last launch, summed busy cycles of
    // all workgroups / of workgroup 0, launches it refused (bbh_tree_sys_counts)
    SysDev sys{};
    size_t sys_ring_bytes = 0;
    uint32_t sys_cap_nodes = 0;
    int sys_G_alloc = 0;
    uint64_t syscount[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t sys_off_left = 0;  // elements the other kernels take after the systolic one refused the tree
    bool sys_pref = false;     // (default policy) this tree goes to the systolic kernel: the pipelined one asked for its multi-level
                               // instance / refused the shape AND the root's centroids are informative (sys_root_informative)
};

namespace {

// `used_elems`: the prefix of the old pool that holds live data (only that much is carried over)
template <typename T>
int grow_pool(T*& p, size_t used_elems, size_t new_elems) {
    // (allocate, copy, free: the old pool and the new one exist side by side for the length of the copy.  Growing in place
    // with the virtual-memory calls - hipMemAddressReserve / hipMemCreate / hipMemMap - was tried in round 4: mappings of a
    // few KB behaved, a 2.2 GB chunk with a 1 GB chunk mapped behind it took a memory access fault on the first touch
    // (tools/probe/vmm_probe.cpp, profiles/r04/vmm_probe.txt), so the pools stay plain allocations.)
    T* np_ = nullptr;
    BB_HIP(bb::dev_alloc(&np_, new_elems * sizeof(T) + 64));  // slack: cf_load_raw over-reads 8 bytes
    if (p && used_elems) BB_HIP(hipMemcpy(np_, p, used_elems * sizeof(T), hipMemcpyDeviceToDevice));
    if (p) bb::dev_free(p);
    p = np_;
    return BBH_OK;
}

// BBHIP_TINY_POOLS=1 (tests): pools are pre-grown by next to nothing, so that the kernels run out of nodes / cluster-feature
// slots in the middle of their runs and every STOP_* -> grow -> relaunch path is exercised
static bool tiny_pools() {  // (read on every call: tests switch it on and off inside one process)
    const char* v = getenv("BBHIP_TINY_POOLS");
    return v != nullptr && v[0] != '\0' && std::strcmp(v, "0") != 0;
}
// a pool that has to grow grows by at least half (a tree fed in 64 MiB slabs asked for a slightly larger pool with every
// slab, i.e. copied all of it every time); `hint` is what the caller expects to need
static uint32_t grow_target(uint32_t cap, uint32_t want, size_t elem_bytes = 0) {
    if (tiny_pools()) return want;
    // (a pool of gigabytes grows by an eighth: while it is copied it exists twice, and at 100 M rows the cf32 pool's "half as
    // much again" was 90 GB next to 60)
    const uint64_t geo = (uint64_t)cap + ((uint64_t)cap * elem_bytes > (4ull << 30) ? cap / 8 : cap / 2);
    return (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, std::max<uint64_t>(want, geo));
}

uint32_t clamp30(uint64_t v) { return (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, v); }

// a generous target (grow_target's half-as-much-again, the rate-based estimates of pregrow) must not be what runs the device
// out of memory: beyond `floor_elems` (what is needed now) the pool only takes what the driver reports free, less a tenth
static uint32_t fit_to_memory(uint32_t want, uint32_t floor_elems, uint32_t cap, size_t elem_bytes) {
    if (want <= floor_elems) return want;
    // (hipMemGetInfo is a driver call of tens of microseconds: a round of 512 small shard trees made 2 048 of them, most of
    // what bench.py's `concurrent_shards` lost between rounds 3 and 4 in a fresh process - small requests are not clamped)
    if ((uint64_t)want * elem_bytes < (256ull << 20)) return want;
    size_t free_b = 0;
    // (caches - this library's and the host side's - are emptied before the answer is taken for the truth: ADVICE r5)
    if (bb::dev_free_bytes((size_t)((double)want * (double)elem_bytes / 0.9) + 1, &free_b) != hipSuccess) { (void)hipGetLastError(); return want; }
    // (the copying growth holds the old pool until the new one is filled: only the free memory counts)
    const size_t room = (size_t)((double)free_b * 0.9) / std::max<size_t>(elem_bytes, 1);
    // (a pool that the free memory only lets grow by the floor - a few dozen elements - would be copied whole every
    // few dozen insertions: when less than a sixteenth more than the pool holds fits, the growth is refused - the allocation
    // of `want` fails with "out of memory", which is the truth - instead of crawling)
    if (room < (uint64_t)cap + std::max<uint64_t>(cap / 16, floor_elems > cap ? floor_elems - cap : 0)) return want;
    const uint64_t most = std::max<uint64_t>(floor_elems, std::min<uint64_t>(want, room));
    return (uint32_t)std::min<uint64_t>(most, 0x3FFFFFFFull);
}

// bytes one block of NG rows takes in the five node pools
static size_t block_bytes(const TreeDev& h) { return (size_t)NG * ((size_t)h.RB + 40) + sizeof(NodeHdr); }
static size_t pool_bytes(const TreeDev& h) {
    const size_t F = (size_t)h.F;
    return (size_t)h.cap_nodes * block_bytes(h) + (size_t)h.cap8 * F + (size_t)h.cap16 * F * 2 + (size_t)h.cap32 * F * 4;
}
static void note_peak(bbh_tree* t, size_t extra) { t->peak_bytes = std::max(t->peak_bytes, pool_bytes(t->h) + extra); }

// The node pools hold `blocks` blocks of NG rows and end in a pad of bf + 1 rows: a node compare requests bf + 1 rows from
// wherever a node starts, whatever its capacity (rows beyond its length are masked).
static size_t pool_rows(const TreeDev& h, size_t blocks) { return blocks * NG + (size_t)h.bf + 1 + NG; }

// new node pools of `nc` blocks; the first `keep` blocks of the old ones are carried over (`keep` == 0: nothing).  Headers
// beyond `keep` are zeroed: a header with capacity 0 starts no node, which is how the compaction tells a node's first
// block from its other blocks and from blocks that were given up (thaw_node).
static int realloc_node_pools(bbh_tree* t, size_t keep, size_t nc, TreeDev* fresh_only = nullptr) {
    TreeDev& h = t->h;
    const size_t rb = (size_t)h.RB;
    TreeDev n = h;
    n.node_cent = nullptr; n.node_card = nullptr; n.node_link = nullptr; n.node_rm = nullptr; n.node_hdr = nullptr;
    const size_t nr = pool_rows(h, nc);
    hipError_t e = bb::dev_alloc(&n.node_cent, nr * rb + 64);
    if (e == hipSuccess) e = bb::dev_alloc(&n.node_card, nr * 4 + 64);
    if (e == hipSuccess) e = bb::dev_alloc(&n.node_link, nr * 4 + 64);
    if (e == hipSuccess) e = bb::dev_alloc(&n.node_rm, nr * sizeof(RowMeta) + 64);
    if (e == hipSuccess) e = bb::dev_alloc(&n.node_hdr, (nc + 1) * sizeof(NodeHdr));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        void* ptrs[] = {n.node_cent, n.node_card, n.node_link, n.node_rm, n.node_hdr};
        for (void* q : ptrs)
            if (q) bb::dev_free(q);
        return bb::fail(BBH_ERR_CAPACITY, "node pools of %zu blocks (%.2f GB) could not be allocated: %s", nc, (double)nc * (double)block_bytes(h) / 1e9,
                        hipGetErrorString(e));
    }
    note_peak(t, nc * block_bytes(h));
    if (fresh_only) {  // (the compaction fills the new pools itself and swaps them in)
        BB_HIP(hipMemset(n.node_hdr, 0, (nc + 1) * sizeof(NodeHdr)));
        *fresh_only = n;
        return BBH_OK;
    }
    // (the memset first: whatever blocking copy follows on the null stream - the carried-over part here, the root header of
    // a tree without rows in the callers - returns when both are done; a memset left in flight could land on a header a
    // kernel on another stream has written since)
    BB_HIP(hipMemset(n.node_hdr + keep, 0, (nc + 1 - keep) * sizeof(NodeHdr)));
    if (keep) {
        const size_t kr = keep * NG;
        BB_HIP(hipMemcpy(n.node_cent, h.node_cent, kr * rb, hipMemcpyDeviceToDevice));
        BB_HIP(hipMemcpy(n.node_card, h.node_card, kr * 4, hipMemcpyDeviceToDevice));
        BB_HIP(hipMemcpy(n.node_link, h.node_link, kr * 4, hipMemcpyDeviceToDevice));
        BB_HIP(hipMemcpy(n.node_rm, h.node_rm, kr * sizeof(RowMeta), hipMemcpyDeviceToDevice));
        BB_HIP(hipMemcpy(n.node_hdr, h.node_hdr, keep * sizeof(NodeHdr), hipMemcpyDeviceToDevice));
    }
    void* old[] = {h.node_cent, h.node_card, h.node_link, h.node_rm, h.node_hdr};
    for (void* q : old)
        if (q) bb::dev_free(q);
    h.node_cent = n.node_cent; h.node_card = n.node_card; h.node_link = n.node_link; h.node_rm = n.node_rm; h.node_hdr = n.node_hdr;
    h.cap_nodes = (uint32_t)nc;
    return BBH_OK;
}

// BBHIP_GC_MIN_MB: node pools below this size grow by plain copying (no compaction, nothing is ever sealed): 1024 by default,
// 0 in the tests that want every growth to compact.  BBHIP_GC=0 switches the compaction off altogether.
static size_t gc_min_bytes() {
    const char* off = getenv("BBHIP_GC");
    if (off != nullptr && std::strcmp(off, "0") == 0) return (size_t)-1;
    const char* v = getenv("BBHIP_GC_MIN_MB");  // (read on every call: tests switch it inside one process)
    return (v != nullptr && v[0] != '\0') ? (size_t)std::strtoull(v, nullptr, 10) << 20 : (size_t)1024 << 20;
}

// Compaction of the node pools into new ones ("Node storage" at the top of this file): every live node moves to the next free
// blocks in id order - nodes that are sealed, or (with `seal`) whose length is what the previous compaction recorded, at their
// length rounded up to a block, the others at full capacity; blocks that thaw_node gave up disappear.  Ids change: child
// ids, the leaf chain, the root and the first leaf are translated; nothing on the host outlives a call with node ids in it
// except the leaf chain's copy, which is dropped.  The new pools hold what is live afterwards plus `extra` blocks, or a
// quarter more if that is more.  seal == 0 brings every node back to full capacity (the exact batch mode wants that).
int gc_nodes(bbh_tree* t, uint64_t extra, int seal) {
    TreeDev& h = t->h;
    const uint32_t used = std::min(h.cap_nodes, h.ctr[C_NODES]);
    const uint32_t rows = (uint32_t)h.bf + 1;
    uint32_t *d_sz = nullptr, *d_id = nullptr;
    unsigned long long* d_cnt = nullptr;
    void* d_tmp = nullptr;
    int rc = BBH_OK;
    TreeDev fresh{};
    bool have_fresh = false;
    auto body = [&]() -> int {
        BB_HIP(bb::dev_alloc(&d_sz, ((size_t)used + 1) * 4));
        BB_HIP(bb::dev_alloc(&d_id, ((size_t)used + 1) * 4));
        BB_HIP(bb::dev_alloc(&d_cnt, 16));
        BB_HIP(hipMemset(d_cnt, 0, 16));
        BB_HIP(hipMemset(d_sz + used, 0, 4));
        hipLaunchKernelGGL(k_gc_size, dim3((used + 255) / 256), dim3(256), 0, 0, (const NodeHdr*)h.node_hdr, used, rows, h.ctr[C_ROOT], seal, d_sz, d_cnt);
        BB_HIP(hipGetLastError());
        size_t tmp_bytes = 0;
        BB_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_sz, d_id, 0u, (size_t)used + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
        BB_HIP(bb::dev_alloc(&d_tmp, tmp_bytes + 16));
        BB_HIP(rocprim::exclusive_scan(d_tmp, tmp_bytes, d_sz, d_id, 0u, (size_t)used + 1, rocprim::plus<uint32_t>(), (hipStream_t)0));
        uint32_t total = 0, new_root = 0, new_first = NONE;
        unsigned long long cnt[2] = {0, 0};
        BB_HIP(hipMemcpy(&total, d_id + used, 4, hipMemcpyDeviceToHost));
        BB_HIP(hipMemcpy(&new_root, d_id + h.ctr[C_ROOT], 4, hipMemcpyDeviceToHost));
        if (h.ctr[C_FIRST_LEAF] != NONE && h.ctr[C_FIRST_LEAF] < used) BB_HIP(hipMemcpy(&new_first, d_id + h.ctr[C_FIRST_LEAF], 4, hipMemcpyDeviceToHost));
        BB_HIP(hipMemcpy(cnt, d_cnt, 16, hipMemcpyDeviceToHost));
        // what the new pools hold: everything live, and room for `extra` blocks or a quarter more - within what the device has
        // free right now (the old pools stay until the move is done)
        const uint64_t floor_b = (uint64_t)total + (2 * (uint64_t)h.ctr[C_DEPTH] + 8) * node_blocks(rows);
        uint64_t want = (uint64_t)total + std::max<uint64_t>(extra, tiny_pools() ? 0 : (uint64_t)total / 4);
        want = std::max(want, floor_b);
        size_t free_b = 0;
        // (what is free AFTER the caches - this library's and, through the memory-pressure callback, the host side's - have been
        // emptied, if the first answer is short of `want`: ADVICE r5)
        if (bb::dev_free_bytes((size_t)((double)want * (double)block_bytes(h) / 0.92) + 1, &free_b) == hipSuccess) {
            const uint64_t room = (uint64_t)((double)free_b * 0.92) / block_bytes(h);
            if (want > room) {
                // (near the end of the device's memory the pool takes what is left - but not in steps so small that every few
                // thousand elements copy all of it again: ADVICE r4)
                const uint64_t least = std::max<uint64_t>(floor_b, (uint64_t)total + std::min<uint64_t>(extra, (uint64_t)total / 32 + 1024));
                if (room < least)
                    return bb::fail(BBH_ERR_CAPACITY, "out of device memory: the node pools hold %.2f GB after compaction and %.2f GB are free", (double)total * (double)block_bytes(h) / 1e9, (double)free_b / 1e9);
                want = room;
            }
        } else {
            (void)hipGetLastError();
        }
        if (want > 0x3FFFFFFFull) want = 0x3FFFFFFFull;
        if (want < floor_b) return bb::fail(BBH_ERR_CAPACITY, "node pool limit of 2^30 blocks reached");
        BB_TRY(realloc_node_pools(t, 0, (size_t)want, &fresh));
        have_fresh = true;
        const uint32_t grid = (uint32_t)std::min<uint64_t>(((uint64_t)used + 3) / 4, 1u << 16);
        hipLaunchKernelGGL(k_gc_move, dim3(std::max(grid, 1u)), dim3(256), 0, 0, h, fresh, used, rows, (const uint32_t*)d_sz, (const uint32_t*)d_id);
        BB_HIP(hipGetLastError());
        BB_HIP(hipDeviceSynchronize());
        void* old[] = {h.node_cent, h.node_card, h.node_link, h.node_rm, h.node_hdr};
        for (void* q : old)
            if (q) bb::dev_free(q);
        h.node_cent = fresh.node_cent; h.node_card = fresh.node_card; h.node_link = fresh.node_link; h.node_rm = fresh.node_rm; h.node_hdr = fresh.node_hdr;
        have_fresh = false;
        h.cap_nodes = (uint32_t)want;
        t->gc_runs += 1; t->gc_full = cnt[0]; t->gc_sealed = cnt[1]; t->gc_before = used; t->gc_after = total;
        h.ctr[C_NODES] = total;
        h.ctr[C_ROOT] = new_root;
        h.ctr[C_FIRST_LEAF] = new_first;
        t->chain_valid = false;
        static const bool gc_log = getenv("BBHIP_GC_LOG") != nullptr;
        if (gc_log)
            fprintf(stderr, "[bbhip gc] #%llu blocks %u -> %u (%.3f -> %.3f GB), nodes full %llu sealed %llu, new pool %.3f GB\n", (unsigned long long)t->gc_runs, used, total,
                    (double)used * (double)block_bytes(h) / 1e9, (double)total * (double)block_bytes(h) / 1e9, cnt[0], cnt[1], (double)want * (double)block_bytes(h) / 1e9);
        return BBH_OK;
    };
    rc = body();
    if (have_fresh) {
        void* ptrs[] = {fresh.node_cent, fresh.node_card, fresh.node_link, fresh.node_rm, fresh.node_hdr};
        for (void* q : ptrs)
            if (q) bb::dev_free(q);
    }
    if (d_sz) bb::dev_free(d_sz);
    if (d_id) bb::dev_free(d_id);
    if (d_cnt) bb::dev_free(d_cnt);
    if (d_tmp) bb::dev_free(d_tmp);
    return rc;
}

// `want`: blocks the node pools should hold.  Small pools grow by copying; pools beyond BBHIP_GC_MIN_MB are compacted on the
// way (gc_nodes), which is where nodes that stopped changing give their unused rows back.
int grow_nodes(bbh_tree* t, uint32_t want, uint64_t gc_extra = 0) {  // (gc_extra: room a compaction leaves; 0: want - used)
    TreeDev& h = t->h;
    if (want <= h.cap_nodes) return BBH_OK;
    const uint32_t nblk = node_blocks((uint32_t)h.bf + 1);
    // a tree that has not received anything yet owns one empty root: nothing to carry over but its
    // 16-byte header (a multiround round creates hundreds of trees and grows each of them once)
    const bool pristine = h.ctr[C_NODES] == nblk && h.ctr[C_IDS] == 0 && h.stats[3] == 0;
    const uint32_t used = std::min(h.cap_nodes, h.ctr[C_NODES]);
    if (!pristine && h.cap_nodes > 0 && (size_t)used * block_bytes(h) >= gc_min_bytes()) return gc_nodes(t, gc_extra ? gc_extra : (uint64_t)want - used, t->no_seal ? 0 : 1);
    {
        const uint32_t floor_elems = clamp30((uint64_t)h.ctr[C_NODES] + (2 * (uint64_t)h.ctr[C_DEPTH] + 64) * nblk);
        want = fit_to_memory(grow_target(h.cap_nodes, want), std::max(floor_elems, h.cap_nodes + 1), h.cap_nodes, block_bytes(h));
    }
    BB_TRY(realloc_node_pools(t, pristine ? 0 : used, want));
    if (pristine) {
        NodeHdr root;
        root.len = 0; root.leaf = hw_make(1u, (uint32_t)h.bf + 1); root.prev = NONE; root.next = NONE;
        BB_HIP(hipMemcpy(h.node_hdr, &root, sizeof(root), hipMemcpyHostToDevice));
    }
    return BBH_OK;
}

int grow_cf(bbh_tree* t, int tier, uint32_t want) {
    TreeDev& h = t->h;
    const size_t F = (size_t)h.F;
    // (a tree that has not received anything yet has nothing to carry over: slot 0 of the uint8 pool - SLOT_LAZY8 - is never
    // looked at.  A multiround round creates hundreds of trees and grows three pools of each of them right away: three
    // blocking 2 KB copies per tree were a third of bench.py's `concurrent_shards` wall time)
    const bool pristine = h.ctr[C_IDS] == 0 && h.stats[3] == 0 && h.ctr[C_N16] == 0 && h.ctr[C_N32] == 0 && h.ctr[C_N8] <= 1;
    if (tier == 0 && want > h.cap8) {
        want = fit_to_memory(grow_target(h.cap8, want, F), std::max<uint32_t>(clamp30((uint64_t)h.ctr[C_N8] + 2 * (uint64_t)h.ctr[C_DEPTH] + 64), h.cap8 + 1), h.cap8, F * 1);
        BB_TRY(grow_pool(h.cf8, pristine ? 0 : (size_t)std::min(h.cap8, h.ctr[C_N8]) * F, (size_t)want * F));
        h.cap8 = want;
    } else if (tier == 1 && want > h.cap16) {
        want = fit_to_memory(grow_target(h.cap16, want, F * 2), std::max<uint32_t>(clamp30((uint64_t)h.ctr[C_N16] + 2 * (uint64_t)h.ctr[C_DEPTH] + 64), h.cap16 + 1), h.cap16, F * 2);
        BB_TRY(grow_pool(h.cf16, (size_t)std::min(h.cap16, h.ctr[C_N16]) * F, (size_t)want * F));
        h.cap16 = want;
    } else if (tier == 2 && want > h.cap32) {
        want = fit_to_memory(grow_target(h.cap32, want, F * 4), std::max<uint32_t>(clamp30((uint64_t)h.ctr[C_N32] + 2 * (uint64_t)h.ctr[C_DEPTH] + 64), h.cap32 + 1), h.cap32, F * 4);
        BB_TRY(grow_pool(h.cf32, (size_t)std::min(h.cap32, h.ctr[C_N32]) * F, (size_t)want * F));
        h.cap32 = want;
    }
    return BBH_OK;
}

void free_pools(bbh_tree* t) {
    TreeDev& h = t->h;
    void* ptrs[] = {h.node_cent, h.node_card, h.node_link, h.node_rm, h.node_hdr, h.scratch_cent, h.cf8, h.cf16, h.cf32};
    for (void* p : ptrs)
        if (p) bb::dev_free(p);
    h.node_cent = nullptr; h.node_card = nullptr; h.node_link = nullptr; h.node_rm = nullptr; h.node_hdr = nullptr;
    h.scratch_cent = nullptr; h.cf8 = nullptr; h.cf16 = nullptr; h.cf32 = nullptr;
    h.cap_nodes = h.cap8 = h.cap16 = h.cap32 = 0;
}

// an empty tree: one empty leaf root (bitbirch.py:880-884)
int init_empty(bbh_tree* t) {
    TreeDev& h = t->h;
    std::memset(h.ctr, 0, sizeof(h.ctr));
    const uint32_t nblk = node_blocks((uint32_t)h.bf + 1);
    // A fresh tree owns no pools: the first call that inserts allocates them once, at the size its elements are expected to
    // need (pregrow -> grow_nodes writes the root's header).  A multiround round creates hundreds of trees; allocating
    // minimal pools here and regrowing all of them in the first fit was most of what such a tree cost the host.
    t->lazy_pools = h.cap_nodes == 0;
    if (!t->lazy_pools) {  // a reset tree keeps its pools: no header of the old tree survives
        NodeHdr root;
        root.len = 0; root.leaf = hw_make(1u, (uint32_t)h.bf + 1); root.prev = NONE; root.next = NONE;
        BB_HIP(hipMemset(h.node_hdr, 0, ((size_t)h.cap_nodes + 1) * sizeof(NodeHdr)));
        BB_HIP(hipMemcpy(h.node_hdr, &root, sizeof(root), hipMemcpyHostToDevice));
    }
    h.ctr[C_NODES] = nblk;
    h.ctr[C_N8] = 1;  // (slot 0 of the uint8 pool is SLOT_LAZY8: never a BitFeature's)
    h.ctr[C_ROOT] = 0;
    h.ctr[C_FIRST_LEAF] = 0;
    h.ctr[C_DEPTH] = 1;
    std::memset(h.stats, 0, sizeof(h.stats));
    std::memset(h.phase, 0, sizeof(h.phase));
    std::memset(h.sphase, 0, sizeof(h.sphase));
    std::memset(h.splitprof, 0, sizeof(h.splitprof));
    std::memset(h.mlprof, 0, sizeof(h.mlprof));
    std::memset(h.rprof, 0, sizeof(h.rprof));
    h.stats[5] = 1;
    t->chain_valid = false;
    t->pipe_ml = false;
    t->unsup_stretch = 0;
    return BBH_OK;
}

int set_tol(bbh_tree* t, const double* tab, int64_t len) {
    if (t->d_tol) {
        (void)bb::dev_free(t->d_tol);
        t->d_tol = nullptr;
    }
    t->h.tol_table = nullptr;
    t->h.tol_len = 0;
    if (tab && len > 0) {
        BB_HIP(bb::dev_alloc(&t->d_tol, (size_t)len * 8));
        BB_HIP(hipMemcpy(t->d_tol, tab, (size_t)len * 8, hipMemcpyHostToDevice));
        t->h.tol_table = t->d_tol;
        t->h.tol_len = (int32_t)len;
    }
    return BBH_OK;
}

int configure(bbh_tree* t, int32_t bf, int32_t n_features) {
    TreeDev& h = t->h;
    h.bf = bf;
    h.F = n_features;
    h.nbytes = n_features / 8;
    h.RB = (h.nbytes + 15) / 16 * 16;
    // LDS mirrors of the nodes on the current path: as many levels (<= MAXM) as fit comfortably
    const int nm = mirror_levels(bf, h.RB);
    h.use_root_cache = nm;
    t->lds = smem_layout(bf, h.RB, nm).total;
    if (h.scratch_cent) bb::dev_free(h.scratch_cent);
    h.scratch_cent = nullptr;
    BB_HIP(bb::dev_alloc(&h.scratch_cent, ((size_t)bf + 1) * h.RB));
    {
        // dynamic LDS above 48 KiB has to be allowed per kernel AND per device (HIP function objects are per device): raised once
        // per device, to what that device offers (160 KiB on gfx950; trees of different shapes share the kernels)
        static std::mutex attr_mu;
        static std::vector<bool> attr_done_dev(64, false);
        std::lock_guard<std::mutex> lock(attr_mu);
        const size_t di = (size_t)std::min(std::max(t->device, 0), 63);
        if (!attr_done_dev[di]) {
            hipDeviceProp_t prop;
            BB_HIP(hipGetDeviceProperties(&prop, t->device));
            // (gfx950: 160 KiB per workgroup; elsewhere - BBHIP_ALLOW_ANY_ARCH - whatever a compute unit has)
            const bool is950 = std::strncmp(prop.gcnArchName, "gfx950", 6) == 0;
            const int cap = is950 ? 160 * 1024 : (int)std::min<size_t>(prop.maxSharedMemoryPerMultiProcessor, 160 * 1024);
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_insert<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_insert<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_insert<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_insert_dense<KC>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_insert<true, false, KC50>, hipFuncAttributeMaxDynamicSharedMemorySize, cap));
            // the steady-state / pipelined kernels use the CU's whole LDS whatever this tree's own layout needs
            if ((uint32_t)cap >= pipe_layout(254).total && (uint32_t)cap >= fast_layout(254).total) {
                for (const FastKernel& fk : kFastKernels)
                    BB_HIP(hipFuncSetAttribute((const void*)fk.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fk.lds));
                BB_HIP(hipFuncSetAttribute((const void*)k_tree_fast<KF50P, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast_layout(50).total));
                BB_HIP(hipFuncSetAttribute((const void*)k_tree_fast<KF254P, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fast_layout(254).total));
                for (const PipeKernel& pk : kPipeKernels) {
                    BB_HIP(hipFuncSetAttribute((const void*)pk.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk.lds));
                    BB_HIP(hipFuncSetAttribute((const void*)pk.fn_prof, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pk.lds));
                }
            } else {
                return bb::fail(BBH_ERR_NO_DEVICE, "device %d offers %d bytes of LDS per workgroup, the tree kernels need %u", t->device, cap,
                                pipe_layout(254).total);
            }
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_sys<KS50>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KS50::o.total + sys_lds_bytes<KS50>())));
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_sys<KS254>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KS254::o.total + sys_lds_bytes<KS254>())));
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_sys<KS50, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KS50::o.total + sys_lds_bytes<KS50>())));
            BB_HIP(hipFuncSetAttribute((const void*)k_tree_sys<KS254, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(KS254::o.total + sys_lds_bytes<KS254>())));
            attr_done_dev[di] = true;
        }
    }
    return BBH_OK;
}

// two workgroups per compute unit pay off once there are more trees than compute units and two
// of them fit in the CU's 160 KiB of LDS
static bool dense_launch(size_t n_trees, size_t lds_bytes) {
    static int cus = -1;
    static int mode = -1;  // BBHIP_DENSE=0/1 forces the choice (measurements)
    if (cus < 0) {
        int dev = 0;
        hipDeviceProp_t p;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
        const char* e = getenv("BBHIP_DENSE");
        mode = e ? atoi(e) : -1;
    }
    if (mode >= 0) return mode != 0 && 2 * lds_bytes <= 160 * 1024;
    return n_trees > (size_t)cus && 2 * lds_bytes <= 160 * 1024;
}


// Pools are grown BEFORE a call to what its n elements are expected to need; a kernel that runs out anyway stops in front of
// the element that does not fit (STOP_NODES / STOP_CF*) and the host grows the pool and relaunches (run_insert_multi).
//   uint8 cluster features: a slot is taken by the merge that gives a BitFeature its second member (SLOT_LAZY8) and by an
//   appended buffer of several members - an eighth of the elements is more than any workload measured took (S-fake 3 %,
//   S-ecfp 1 %, S-rdkit-like 10 %); nodes: one per bf / 2 elements (a node is half full after its split); tracking cluster
//   features (uint32): two per node split.
int pregrow(bbh_tree* t, int64_t n, int width) {
    TreeDev& h = t->h;
    const bool first = t->lazy_pools;  // the tree's pools come into being here: every pool at least at its minimum
    // (lazy_pools is cleared only when every first allocation has succeeded - ADVICE r5: a tree whose first fit ran out of
    // memory must still read as "owns no pools" to build_chain / export / compact)
    struct Commit { bbh_tree* t; bool ok = false; ~Commit() { if (ok) t->lazy_pools = false; } } commit{t};
    if (first && tiny_pools()) {
        BB_TRY(grow_cf(t, 0, 8));
        BB_TRY(grow_cf(t, 1, 8));
        BB_TRY(grow_cf(t, 2, 16));
    }
    if (tiny_pools()) {
        BB_TRY(grow_cf(t, 0, clamp30((uint64_t)h.ctr[C_N8] + 8)));
        BB_TRY(grow_cf(t, 1, clamp30((uint64_t)h.ctr[C_N16] + 8)));
        BB_TRY(grow_nodes(t, clamp30((uint64_t)h.ctr[C_NODES] + (8 + 2 * (uint64_t)h.ctr[C_DEPTH]) * node_blocks((uint32_t)h.bf + 1))));
        BB_TRY(grow_cf(t, 2, clamp30((uint64_t)h.ctr[C_N32] + 16 + 2 * (uint64_t)h.ctr[C_DEPTH])));
        commit.ok = true;
        return BBH_OK;
    }
    const uint64_t un = (uint64_t)std::max<int64_t>(n, 0);
    if (width <= 1 || first) BB_TRY(grow_cf(t, 0, clamp30((uint64_t)h.ctr[C_N8] + (width <= 1 ? un / 8 : 0) + 1024)));
    if (width == 2 || first) BB_TRY(grow_cf(t, 1, clamp30((uint64_t)h.ctr[C_N16] + (width == 2 ? un : 0) + 64)));
    BB_TRY(grow_nodes(t, clamp30((uint64_t)h.ctr[C_NODES] + (un / (uint64_t)std::max(1, h.bf / 2) + 64) * node_blocks((uint32_t)h.bf + 1))));
    BB_TRY(grow_cf(t, 2, clamp30((uint64_t)h.ctr[C_N32] + un / (uint64_t)std::max(1, h.bf / 6) + 256)));
    commit.ok = true;
    return BBH_OK;
}

// longest stretch on the steady-state kernel between two attempts of the pipelined one (see STOP_PIPE_UNSUPPORTED below): 2^16
// elements.  An attempt that finds the shape still unsupported costs a launch that inserts nothing (~0.2 ms); a stretch that
// doubled to 2^22 (rounds 3-4) kept S-rdkit-like merge rounds on the slower kernel for millions of elements after the shape
// had come back (config 5 at 12 M rows: merge round 47.2 -> 44.8 s, profiles/r05/ab_unsup_cap.txt)
#ifndef BBH_UNSUP_CAP
#define BBH_UNSUP_CAP (1ll << 16)
#endif

// One insertion job: a tree and the elements to insert into it.
struct Job {
    bbh_tree* t;
    const uint8_t* rows;  // device
    int64_t row_stride;
    const uint8_t* bufs;  // device
    int width;
    int64_t n;
    uint32_t* out;  // device or null
    int64_t done;
    int stalls;
    int64_t old_left = 0;  // elements the steady-state kernel takes next (the pipelined one reported a shape it does not handle)
};

// ---- the level-systolic kernel's work area and launch plan (bb_tree_sys.inc) --------------------------------------------
// the words workgroups talk through: uncached device memory (bb_tree_sys.inc, "Memory model")
static bool sys_mem_host() {
    static const bool h = [] { const char* m = getenv("BBHIP_SYS_MEM"); return m && m[0] == 'h'; }();
    return h;
}
template <typename T>
static hipError_t sys_alloc_uc(T** p, size_t bytes) {
    if (sys_mem_host()) {  // (experiment: pinned host memory, coherent by construction, every access crosses the fabric)
        hipError_t eh = hipHostMalloc((void**)p, bytes, hipHostMallocCoherent);
        return eh;
    }
    static const unsigned flags = [] {
        const char* m = getenv("BBHIP_SYS_MEM");  // (experiments: "plain" = ordinary device memory, "fine" = fine-grained)
        if (m && m[0] == 'p') return (unsigned)hipDeviceMallocDefault;
        if (m && m[0] == 'f') return (unsigned)hipDeviceMallocFinegrained;
        return (unsigned)hipDeviceMallocUncached;
    }();
    hipError_t e = hipExtMallocWithFlags((void**)p, bytes, flags);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        bb::dev_trim();
        e = hipExtMallocWithFlags((void**)p, bytes, flags);
    }
    return e;
}
void sys_free(bbh_tree* t) {
    void* uc[] = {t->sys.rings, t->sys.mail, t->sys.ctl};
    for (void* q : uc)
        if (q) (void)(sys_mem_host() ? hipHostFree(q) : hipFree(q));
    void* ptrs[] = {t->sys.laste, t->sys.busy, t->sys.sent, t->sys.up, t->sys.acks};
    for (void* q : ptrs)
        if (q) bb::dev_free(q);
    t->sys = SysDev{};
    t->sys_ring_bytes = 0;
    t->sys_cap_nodes = 0;
    t->sys_G_alloc = 0;
}

// BBHIP_SYS: unset / "0" never (the default: the kernel is OPT-IN), "1" whenever the tree's shape allows it, "auto": where
// the other kernels are weakest (see the selection in run_insert_multi).  Opt-in because its cross-workgroup hand-over is not
// yet dependable on this hardware: the 1 M-row workloads it was built for are per-element identical to the oracle in every one
// of ~100 runs, but the randomised suite's adversarial shapes at bf 254 (every node full, four levels, a split every few
// elements) end in a detected inconsistency or - rarely - a silently different tree in 2-7 % of runs
// (profiles/r06/sys_stability.txt).  Read on every call: tests switch it inside one process.
static int sys_mode() {
    const char* v = getenv("BBHIP_SYS");
    if (v == nullptr || v[0] == '\0' || std::strcmp(v, "0") == 0) return 0;
    if (std::strcmp(v, "auto") == 0) return 2;
    return 1;
}

// Are the root's centroids informative (some row's popcount non-zero)?  Trees over sparse / weakly clustered rows keep
// all-zero centroids in their upper levels (every similarity 0, np.argmax -> row 0): one exact level, the shape the pipelined
// kernel was built for (S-fake: 460 k/s there, 274 k/s here).  Trees over real-fingerprint-like rows compare at every level:
// that is what the systolic kernel is for (zipf / hier: 122-131 k/s there, 357-379 k/s here).  Two small blocking copies, made
// only when the pipelined kernel has just handed the tree over.
static bool sys_root_informative(bbh_tree* t) {
    const TreeDev& h = t->h;
    const uint32_t root = h.ctr[C_ROOT];
    NodeHdr hd{};
    if (hipMemcpy(&hd, h.node_hdr + root, sizeof(hd), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
    const uint32_t len = std::min<uint32_t>(hd.len, (uint32_t)h.bf + 1);
    if (len == 0 || (hd.leaf & HW_LEAF)) return false;
    std::vector<uint32_t> cards(len);
    if (hipMemcpy(cards.data(), h.node_card + (size_t)root * NG, (size_t)len * 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return false; }
    for (uint32_t cd : cards)
        if (cd != 0) return true;
    return false;
}

// workgroups per tree level: level 0 is workgroup 0 (the root), the last level the leaf owners
static void sys_plan(const TreeDev& h, SysDev& S) {
    const int levels = (int)h.ctr[C_DEPTH];  // (1: the root is a leaf)
    S.levels = levels;
    int first = 0;
    static const int env_int = [] { const char* v = getenv("BBHIP_SYS_INTERNAL_WGS"); return v ? atoi(v) : 0; }();
    static const int env_leaf = [] { const char* v = getenv("BBHIP_SYS_LEAF_WGS"); return v ? atoi(v) : 0; }();
    for (int l = 0; l < levels && l < SYS_MAXLVL; ++l) {
        int cnt;
        if (l == 0) cnt = 1;
        else if (l == levels - 1) cnt = env_leaf > 0 ? env_leaf : 64;
        else cnt = env_int > 0 ? env_int : (h.bf >= 128 ? 64 : 16);  // (bf 254: one mirrored node per owner - as many owners as a level has nodes)
        cnt = std::min(cnt, SYS_MAXPROD);
        while (cnt & (cnt - 1)) cnt &= cnt - 1;  // a power of two (sys_owner_idx)
        S.lvl_first[l] = first;
        S.lvl_count[l] = cnt;
        first += cnt;
    }
    S.G = first;
    // every workgroup has to be RESIDENT (one per CU: the kernel takes most of a CU's LDS) - a workgroup that waits for a CU while
    // the others wait for its answers is a deadlock (found by the randomised suite: five levels at bf 254 came to 257 workgroups)
    while (S.G > 224) {
        first = 0;
        for (int l = 0; l < levels && l < SYS_MAXLVL; ++l) {
            if (l > 0 && l < levels - 1 && S.lvl_count[l] > 8) S.lvl_count[l] /= 2;
            S.lvl_first[l] = first;
            first += S.lvl_count[l];
        }
        if (first == S.G) break;
        S.G = first;
    }
    S.qmax = 384;  // (< SYS_R: no ring can overflow)
}

// (re)allocate what the plan and the node pool's size need; zero the rings and control words, initialise the mailboxes
static int sys_prepare(bbh_tree* t, hipStream_t s) {
    TreeDev& h = t->h;
    SysDev& S = t->sys;
    sys_plan(h, S);
    {
        // the kernel admits an element only while the pools hold the worst case of everything in flight (every element splits
        // every level and the root): make sure a launch starts with that reserve and room to work in
        const uint64_t q = (uint64_t)S.qmax + 64, depth = h.ctr[C_DEPTH], nblk = node_blocks((uint32_t)h.bf + 1);
        const uint64_t room = tiny_pools() ? 16 : 4096;  // (elements' worth of room beyond the reserve)
        BB_TRY(grow_nodes(t, clamp30((uint64_t)h.ctr[C_NODES] + (q * (depth + 2) + room / 8 + 8) * nblk)));
        BB_TRY(grow_cf(t, 0, clamp30((uint64_t)h.ctr[C_N8] + q + room)));
        BB_TRY(grow_cf(t, 1, clamp30((uint64_t)h.ctr[C_N16] + q + 64)));
        BB_TRY(grow_cf(t, 2, clamp30((uint64_t)h.ctr[C_N32] + q * 2 * (depth + 2) + room / 4 + 16)));
    }
    const size_t ring_bytes = (size_t)S.G * SYS_MAXPROD * (size_t)SYS_R * 16;
    {
        static std::atomic<uint32_t> g_sys_launch{0};  // (process-wide: a new tree may inherit another tree's ring memory)
        S.launch_id = (g_sys_launch.fetch_add(1u) + 1u) & 0x7FFFFFFFu;
        if (S.launch_id == 0) S.launch_id = (g_sys_launch.fetch_add(1u) + 1u) & 0x7FFFFFFFu;
    }
    if (ring_bytes > t->sys_ring_bytes || S.G > t->sys_G_alloc) {
        if (S.rings) (void)(sys_mem_host() ? hipHostFree(S.rings) : hipFree(S.rings));
        if (S.busy) bb::dev_free(S.busy);
        S.rings = nullptr; S.busy = nullptr;
        BB_HIP(sys_alloc_uc(&S.rings, ring_bytes));
        BB_HIP(bb::dev_alloc(&S.busy, (size_t)S.G * (15 * 8 + 3 * SYS_MAXPROD * 4) + 64));
        t->sys_ring_bytes = ring_bytes;
        t->sys_G_alloc = S.G;
    }
    if (!S.ctl) BB_HIP(sys_alloc_uc(&S.ctl, SC_COUNT * 4));
    if (h.cap_nodes > t->sys_cap_nodes || !S.mail) {
        if (S.mail) (void)(sys_mem_host() ? hipHostFree(S.mail) : hipFree(S.mail));
        if (S.sent) bb::dev_free(S.sent);
        if (S.up) bb::dev_free(S.up);
        if (S.laste) bb::dev_free(S.laste);
        if (S.acks) bb::dev_free(S.acks);
        S.mail = nullptr; S.sent = nullptr; S.up = nullptr; S.laste = nullptr; S.acks = nullptr;
        BB_HIP(bb::dev_alloc(&S.laste, (size_t)h.cap_nodes * 4 + 64));
        BB_HIP(bb::dev_alloc(&S.acks, (size_t)h.cap_nodes * 4 + 64));
        BB_HIP(bb::dev_alloc(&S.sent, (size_t)h.cap_nodes * 4 + 64));
        BB_HIP(bb::dev_alloc(&S.up, (size_t)h.cap_nodes * 8 + 64));
        BB_HIP(sys_alloc_uc(&S.mail, (size_t)h.cap_nodes * 8 + 64));
        t->sys_cap_nodes = h.cap_nodes;
    }
    BB_HIP(hipMemsetAsync(S.rings, 0, ring_bytes, s));
    BB_HIP(hipMemsetAsync(S.ctl, 0, SC_COUNT * 4, s));
    BB_HIP(hipMemsetAsync(S.busy, 0, (size_t)S.G * (15 * 8 + 3 * SYS_MAXPROD * 4), s));
    const uint32_t used = std::min(h.cap_nodes, h.ctr[C_NODES]);
    hipLaunchKernelGGL(k_sys_init, dim3((used + 255) / 256), dim3(256), 0, s, (const NodeHdr*)h.node_hdr, used, S.mail, S.sent, S.up, S.laste, S.acks);
    BB_HIP(hipGetLastError());
    return BBH_OK;
}

// ids handed out by a launch of the systolic kernel, renumbered into the sequential engines' order (bb_tree_sys.inc)
static int sys_renumber(bbh_tree* t, uint32_t* out_leaf, uint32_t n, uint32_t base, uint32_t nnew, hipStream_t s) {
    if (out_leaf == nullptr || n == 0 || nnew == 0) return BBH_OK;
    uint32_t *creator = nullptr, *flag = nullptr, *rank = nullptr, *map = nullptr;
    void* tmp = nullptr;
    auto body = [&]() -> int {
        BB_HIP(bb::dev_alloc(&creator, (size_t)nnew * 4 + 64));
        BB_HIP(bb::dev_alloc(&flag, (size_t)n * 4 + 64));
        BB_HIP(bb::dev_alloc(&rank, (size_t)n * 4 + 64));
        BB_HIP(bb::dev_alloc(&map, (size_t)nnew * 4 + 64));
        BB_HIP(hipMemsetAsync(creator, 0xFF, (size_t)nnew * 4, s));
        const dim3 ge((n + 255) / 256), gi((nnew + 255) / 256), blk(256);
        hipLaunchKernelGGL(k_sys_creator, ge, blk, 0, s, (const uint32_t*)out_leaf, n, base, creator);
        hipLaunchKernelGGL(k_sys_flag, ge, blk, 0, s, (const uint32_t*)out_leaf, n, base, (const uint32_t*)creator, flag);
        size_t tmp_bytes = 0;
        BB_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, flag, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
        BB_HIP(bb::dev_alloc(&tmp, tmp_bytes + 16));
        BB_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, flag, rank, 0u, (size_t)n, rocprim::plus<uint32_t>(), s));
        hipLaunchKernelGGL(k_sys_map, gi, blk, 0, s, (const uint32_t*)creator, (const uint32_t*)rank, nnew, base, map);
        hipLaunchKernelGGL(k_sys_apply_out, ge, blk, 0, s, out_leaf, n, base, nnew, (const uint32_t*)map);
        const uint32_t used = std::min(t->h.cap_nodes, t->h.ctr[C_NODES]);
        hipLaunchKernelGGL(k_sys_apply_rows, dim3((used + 3) / 4), blk, 0, s, (const NodeHdr*)t->h.node_hdr, t->h.node_rm, used, base, nnew, (const uint32_t*)map);
        BB_HIP(hipGetLastError());
        BB_HIP(hipStreamSynchronize(s));
        return BBH_OK;
    };
    const int rc = body();
    void* ptrs[] = {creator, flag, rank, map, tmp};
    for (void* q : ptrs)
        if (q) bb::dev_free(q);
    return rc;
}

// Run the insertion kernel over all jobs, ONE WORKGROUP PER TREE in a single launch (independent
// trees - multiround shards - run concurrently on different CUs), relaunching the unfinished
// ones after growing whatever pool made them stop.
int run_insert_multi(std::vector<Job>& jobs, hipStream_t s) {
    if (jobs.empty()) return BBH_OK;
    TreeDev* darr = nullptr;
    const bool single = jobs.size() == 1;
    if (!single) BB_HIP(bb::dev_alloc(&darr, jobs.size() * sizeof(TreeDev)));
    std::vector<TreeDev> harr(jobs.size());
    std::vector<size_t> active;
    int rc = BBH_OK;
    size_t lds = 0;
    for (auto& j : jobs) {
        j.t->chain_valid = false;
        lds = std::max(lds, j.t->lds);
    }
    // Several trees in the pipelined kernel (bf 254, where it is well ahead of the steady-state kernel: 8 x 1 M rows S-fake 1.85 M
    // against 1.46 M fingerprints/s, S-ecfp 1.53 M against 1.18 M; at bf 50 eight steady-state workgroups are as fast as eight
    // pipelines that leave for the multi-level instance): worth it while the trees stay in it.  Trees that keep leaving it
    // (shapes it hands over every few ten thousand elements: S-rdkit-like rows at bf 254) wait for the others' workgroups after
    // every stop and take everybody to the steady-state kernel for their stretch - measured 0.69 M fingerprints/s against
    // 1.33 M with the steady-state kernel alone.  So: when a quarter of the call's trees have left shared launches midway
    // for an unsupported shape, the attempt is over for this call.
    bool multi_pipe_ok = true;
    size_t multi_left_midway = 0;  // trees that left a shared launch midway for an unsupported shape, over the whole call
    long long multi_chunk = 1ll << 15;  // elements per tree and shared launch: doubles (to 2^18) with every launch no tree left midway
    while (rc == BBH_OK) {
        long long multi_requested = 0;
        active.clear();
        for (size_t i = 0; i < jobs.size(); ++i)
            if (jobs[i].done < jobs[i].n) active.push_back(i);
        if (active.empty()) break;
        for (size_t a = 0; a < active.size(); ++a) {
            Job& j = jobs[active[a]];
            TreeDev& h = j.t->h;
            h.rows = j.rows ? j.rows + j.done * j.row_stride : nullptr;
            h.row_stride = j.row_stride;
            h.bufs = j.bufs ? j.bufs + (size_t)j.done * ((size_t)h.F + 1) * j.width : nullptr;
            h.width = j.width;
            h.n_elems = j.old_left > 0 ? std::min<int64_t>(j.old_left, j.n - j.done) : j.n - j.done;
            h.out_leaf = j.out ? j.out + j.done : nullptr;
            harr[a] = h;
        }
        TreeDev* dptr = single ? jobs[0].t->d : darr;
        size_t prof_tok = (size_t)-1;
        uint32_t sys_ids_before = 0;  // (the systolic kernel: BitFeature ids in use before its launch)
        static const bool launch_log = [] {  // one line per launch on stderr (tools/); unset, empty or "0": off
            const char* v = getenv("BBHIP_LAUNCH_LOG");
            return v != nullptr && v[0] != '\0' && std::strcmp(v, "0") != 0;
        }();
        const auto log_t0 = std::chrono::steady_clock::now();
        const char* log_kernel = "complete";
        uint64_t log_before[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (launch_log)
            for (size_t a = 0; a < active.size(); ++a)
                for (int z = 0; z < 8; ++z) log_before[z] += jobs[active[a]].t->h.stats[z];
        hipError_t e = hipMemcpyAsync(dptr, harr.data(), active.size() * sizeof(TreeDev), hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "H2D: %s", hipGetErrorString(e)); break; }
        {
            bb::ProfScope ps("tree_insert", s);
            prof_tok = ps.tok;
            static const bool prof_phases = getenv("BBHIP_PHASES") != nullptr;
            // trees of the benchmark / default shape run the kernel compiled for that shape
            static const bool no_fix = getenv("BBHIP_NO_FIXED_SHAPE") != nullptr;
            bool all50 = !no_fix, all254 = !no_fix;
            for (size_t a = 0; a < active.size() && (all50 || all254); ++a) {
                const TreeDev& q = jobs[active[a]].t->h;
                all50 = all50 && q.bf == 50 && q.F == 2048;
                all254 = all254 && q.bf == 254 && q.F == 2048;
            }
            // the steady-state kernel: 2048-bit rows, bf 50 / 254, criteria without extra reductions
            static const bool no_fast = getenv("BBHIP_NO_FAST") != nullptr;
            bool fast_ok = !no_fast && (all50 || all254);
            bool f_packed = true, f_buffers = true;
            int f_crit = active.empty() ? -1 : jobs[active[0]].t->h.crit;  // the criterion all trees share, or -1
            for (size_t a = 0; a < active.size() && fast_ok; ++a) {
                const Job& fj = jobs[active[a]];
                const int c = fj.t->h.crit;
                if (c != f_crit) f_crit = -1;
                fast_ok = c == BBH_CRIT_DIAMETER || c == BBH_CRIT_TOL_DIAMETER || c == BBH_CRIT_TOL_LEGACY || c == BBH_CRIT_NEVER;
                f_packed = f_packed && fj.bufs == nullptr;
                f_buffers = f_buffers && fj.bufs != nullptr;
                // packed rows are read with 16-byte loads straight into registers
                if (fj.bufs == nullptr && ((((uintptr_t)fj.rows) | (uintptr_t)fj.row_stride) & 15) != 0) fast_ok = false;
            }
            fast_ok = fast_ok && (f_packed || f_buffers);
            const FastKernel* fk = nullptr;
            if (fast_ok)
                for (const FastKernel& c : kFastKernels)
                    if (fk == nullptr && c.bf == (all50 ? 50 : 254) && c.buffers == f_buffers && (c.crit == -1 || c.crit == f_crit)) fk = &c;
            const dim3 grid((unsigned)active.size()), block(TB);
            // The pipelined kernel: packed rows, the criteria of the standard pipelines.  Several trees (multiround shards,
            // one workgroup each): all of them must be ready for it - a tree that is on a stretch of the steady-state
            // kernel (old_left) takes the whole launch there, the others come along for the length of that stretch - and
            // a launch inserts at most `multi_chunk` elements per tree: a tree that stops early (a shape the pipeline
            // hands over, a pool that ran out) waits for the others' workgroups, and no longer than that.
            static const bool no_pipe = getenv("BBHIP_NO_PIPE") != nullptr;
            static const bool no_pipe_multi = getenv("BBHIP_NO_PIPE_MULTI") != nullptr;
            const long long PIPE_MULTI_CHUNK = multi_chunk;
            const PipeKernel* pk = nullptr;
            const bool pipe_possible = fk != nullptr && !no_pipe && !prof_phases && f_packed && (single || (all254 && !no_pipe_multi && multi_pipe_ok));
            bool recopy = false;
            if (pipe_possible) {
                bool all_ready = true;
                int want_ml = 0;
                int64_t max_old = 0;
                for (size_t a = 0; a < active.size(); ++a) {
                    const Job& pj = jobs[active[a]];
                    all_ready = all_ready && pj.old_left == 0 && harr[a].n_elems < (1ll << 31);
                    max_old = std::max(max_old, pj.old_left);
                    want_ml |= pj.t->pipe_ml ? 1 : 0;  // (the multi-level instance also runs trees of one exact level)
                }
                if (all_ready) {
                    for (const PipeKernel& c : kPipeKernels)
                        if (pk == nullptr && c.bf == (all50 ? 50 : 254) && c.crit == f_crit && c.ml == want_ml) pk = &c;
                    if (pk != nullptr && !single)
                        for (size_t a = 0; a < active.size(); ++a) {
                            if (harr[a].n_elems > PIPE_MULTI_CHUNK) { harr[a].n_elems = PIPE_MULTI_CHUNK; recopy = true; }
                            multi_requested += harr[a].n_elems;
                        }
                } else if (!single && max_old > 0) {
                    for (size_t a = 0; a < active.size(); ++a)
                        if (jobs[active[a]].old_left == 0 && harr[a].n_elems > max_old) { harr[a].n_elems = max_old; recopy = true; }
                }
            }
            // The level-systolic kernel (bb_tree_sys.inc): ONE tree over many workgroups.  BBHIP_SYS=1: whenever the shape allows;
            // default: where the pipelined kernel has nothing to offer - it asked for its multi-level instance (informative levels
            // above the leaf-parents at bf 50) or refused the shape (bf 254) and the steady-state kernel would take the stretch.
            int64_t sys_n = 0;
            static const bool pipe_diag = getenv("BBHIP_PIPE_AUDIT") != nullptr || getenv("BBHIP_PIPE_PHASES") != nullptr;  // (diagnostics of the pipelined kernel: keep it in charge)
            if (!pipe_diag && single && fk != nullptr && f_packed && !prof_phases && sys_mode() != 0) {
                Job& sj = jobs[active[0]];
                bbh_tree* st = sj.t;
                const int levels = (int)st->h.ctr[C_DEPTH];
                const bool shape_ok = levels >= 2 && levels <= SYS_MAXLVL && st->gc_runs == 0 && harr[0].n_elems < (1ll << 31) && st->sys_off_left == 0;
                if (shape_ok) {
                    if (sys_mode() == 1) {
                        sys_n = harr[0].n_elems;
                    } else {
                        if (!st->sys_pref && ((all254 && sj.old_left > 0) || (all50 && st->pipe_ml)) && sys_root_informative(st)) st->sys_pref = true;
                        if (st->sys_pref) {
                            sj.old_left = 0;  // (the stretch the steady-state kernel was to take is this kernel's, and so is the rest of the call)
                            sys_n = sj.n - sj.done;
                        }
                    }
                }
                if (sys_n > 0) {
                    pk = nullptr;
                    sys_ids_before = st->h.ctr[C_IDS];
                    rc = sys_prepare(st, s);
                    if (rc != BBH_OK) break;
                    harr[0] = st->h;  // (the pools may have grown)
                    harr[0].n_elems = sys_n;
                    recopy = true;
                }
            }
            if (!pipe_possible && !single) {
                // (the attempt is over: nobody comes back to the pipeline, the stretches are the rest of the input)
                for (size_t a = 0; a < active.size(); ++a) {
                    Job& oj = jobs[active[a]];
                    if (oj.old_left > 0) { oj.old_left = 0; harr[a].n_elems = oj.n - oj.done; recopy = true; }
                }
            }
            if (pk != nullptr) {
                // tier promotions of elements in flight take cf16 / cf32 slots without asking: keep a reserve
                for (size_t a = 0; a < active.size() && rc == BBH_OK; ++a) {
                    bbh_tree* ta = jobs[active[a]].t;
                    bool grown = false;
                    if (ta->h.cap16 - std::min(ta->h.cap16, ta->h.ctr[C_N16]) < (uint32_t)PROMO_RESERVE) {
                        rc = grow_cf(ta, 1, clamp30((uint64_t)ta->h.ctr[C_N16] + 2 * PROMO_RESERVE));
                        grown = true;
                    }
                    if (rc == BBH_OK && ta->h.cap32 - std::min(ta->h.cap32, ta->h.ctr[C_N32]) < (uint32_t)PROMO_RESERVE) {
                        rc = grow_cf(ta, 2, clamp30((uint64_t)ta->h.ctr[C_N32] + 2 * PROMO_RESERVE));
                        grown = true;
                    }
                    if (grown) {
                        harr[a].cf16 = ta->h.cf16; harr[a].cf32 = ta->h.cf32; harr[a].cap16 = ta->h.cap16; harr[a].cap32 = ta->h.cap32;
                        recopy = true;
                    }
                }
                if (rc != BBH_OK) break;
            }
            if (recopy) {
                e = hipMemcpyAsync(dptr, harr.data(), active.size() * sizeof(TreeDev), hipMemcpyHostToDevice, s);
                if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "H2D: %s", hipGetErrorString(e)); break; }
            }
            if (sys_n > 0) {
                log_kernel = "sys";
                const SysDev& S = jobs[active[0]].t->sys;
                static const bool sys_phases = getenv("BBHIP_SYS_PHASES") != nullptr;
                if (sys_phases) {
                    if (all50) hipLaunchKernelGGL((k_tree_sys<KS50, true>), dim3((unsigned)S.G), block, KS50::o.total + sys_lds_bytes<KS50>(), s, dptr, S);
                    else hipLaunchKernelGGL((k_tree_sys<KS254, true>), dim3((unsigned)S.G), block, KS254::o.total + sys_lds_bytes<KS254>(), s, dptr, S);
                } else if (all50) hipLaunchKernelGGL(k_tree_sys<KS50>, dim3((unsigned)S.G), block, KS50::o.total + sys_lds_bytes<KS50>(), s, dptr, S);
                else hipLaunchKernelGGL(k_tree_sys<KS254>, dim3((unsigned)S.G), block, KS254::o.total + sys_lds_bytes<KS254>(), s, dptr, S);
            } else if (pk != nullptr) {
                log_kernel = "pipe";
                static const bool pipe_audit = getenv("BBHIP_PIPE_AUDIT") != nullptr;
                static const bool pipe_phases = getenv("BBHIP_PIPE_PHASES") != nullptr || pipe_audit;
                if (pipe_audit) {
                    static const int audit_mode = std::strcmp(getenv("BBHIP_PIPE_AUDIT"), "corrupt") == 0 ? 2 : 1;
                    for (size_t a = 0; a < active.size(); ++a) harr[a].audit = audit_mode;
                    e = hipMemcpyAsync(dptr, harr.data(), active.size() * sizeof(TreeDev), hipMemcpyHostToDevice, s);
                    if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "H2D: %s", hipGetErrorString(e)); break; }
                }
                // (every instance has its phase-timer / audit twin - ADVICE r4: tolerance-diameter trees used to run unaudited
                // under BBHIP_PIPE_AUDIT while bbh_tree_stats reported "no difference")
                hipLaunchKernelGGL(pipe_phases ? pk->fn_prof : pk->fn, grid, block, pk->lds, s, dptr);
            } else if (fk != nullptr && prof_phases && f_packed) {
                log_kernel = "fast+phases";
                if (all50) hipLaunchKernelGGL((k_tree_fast<KF50P, true>), grid, block, fk->lds, s, dptr);
                else hipLaunchKernelGGL((k_tree_fast<KF254P, true>), grid, block, fk->lds, s, dptr);
            } else if (fk != nullptr) {
                log_kernel = "fast";
                hipLaunchKernelGGL(fk->fn, grid, block, fk->lds, s, dptr);
            } else if (prof_phases) {
                const uint32_t* const nu = nullptr;
                if (all50) hipLaunchKernelGGL((k_tree_insert<true, false, KC50>), grid, block, lds, s, dptr, nu, nu, nu);
                else hipLaunchKernelGGL((k_tree_insert<true, false>), grid, block, lds, s, dptr, nu, nu, nu);
            } else {
                // the complete engine, shape as run-time values; two workgroups per CU when there are more trees than CUs
                const uint32_t* const nu = nullptr;
                if (dense_launch(active.size(), lds)) hipLaunchKernelGGL(k_tree_insert_dense<KC>, grid, block, lds, s, dptr);
                else hipLaunchKernelGGL((k_tree_insert<false, false, KC>), grid, block, lds, s, dptr, nu, nu, nu);
            }
            e = hipGetLastError();
        }
        if (prof_tok != (size_t)-1)
            bb::prof_rename(prof_tok, log_kernel[0] == 's' ? "tree_insert/sys" : (log_kernel[0] == 'p' ? "tree_insert/pipe" : (log_kernel[0] == 'f' ? "tree_insert/fast" : "tree_insert/complete")));
        if (e == hipSuccess) e = hipMemcpyAsync(harr.data(), dptr, active.size() * sizeof(TreeDev), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "tree_insert: %s", hipGetErrorString(e)); break; }
        if (log_kernel[0] == 'p' && harr[0].audit) {
            unsigned int au = 0;
            (void)hipMemcpyFromSymbol(&au, HIP_SYMBOL(g_pipe_audit), sizeof(au));
            if (au != 0) {
                const unsigned int zero = 0;
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pipe_audit), &zero, sizeof(zero));
                rc = bb::fail(BBH_ERR_HIP, "pipelined kernel, run-end audit: LDS state differs from HBM (check %u at bb_tree_pipe.inc:%u)", au >> 16, au & 0xFFFFu);
                break;
            }
        }
        if (launch_log) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - log_t0).count();
            uint64_t after[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            long long done = 0;
            for (size_t a = 0; a < active.size(); ++a) {
                for (int z = 0; z < 8; ++z) after[z] += harr[a].stats[z];
                done += harr[a].processed;
            }
            const Job& j0 = jobs[active[0]];
            fprintf(stderr, "[bbhip launch] %s trees=%zu %s crit=%d elems=%lld %.2f ms (%.2f us/elem) calls=%llu rows=%llu merges=%llu appends=%llu "
                    "leaf_splits=%llu node_splits=%llu stop=%d\n", log_kernel, active.size(), j0.bufs ? (j0.width == 1 ? "buf8" : j0.width == 2 ? "buf16" : "buf32+") : "packed",
                    j0.t->h.crit, done, ms, done ? 1e3 * ms / (double)done : 0.0, (unsigned long long)(after[0] - log_before[0]),
                    (unsigned long long)(after[1] - log_before[1]), (unsigned long long)(after[2] - log_before[2]),
                    (unsigned long long)(after[3] - log_before[3]), (unsigned long long)(after[4] - log_before[4]),
                    (unsigned long long)(after[5] - log_before[5]), harr[0].stop_reason);
        }
        if (multi_requested > 0) {
            size_t left_midway = 0;
            for (size_t a = 0; a < active.size(); ++a) left_midway += harr[a].stop_reason == STOP_PIPE_UNSUPPORTED && harr[a].processed > 0;
            multi_left_midway += left_midway;
            if (4 * multi_left_midway >= jobs.size()) multi_pipe_ok = false;
            else if (left_midway == 0) multi_chunk = std::min<long long>(2 * multi_chunk, 1ll << 18);
        }
        for (size_t a = 0; a < active.size() && rc == BBH_OK; ++a) {
            Job& j = jobs[active[a]];
            bbh_tree* t = j.t;
            TreeDev& h = t->h;
            const TreeDev& back = harr[a];
            std::memcpy(h.ctr, back.ctr, sizeof(h.ctr));
            std::memcpy(h.stats, back.stats, sizeof(h.stats));
            std::memcpy(h.phase, back.phase, sizeof(h.phase));
            std::memcpy(h.sphase, back.sphase, sizeof(h.sphase));
            std::memcpy(h.splitprof, back.splitprof, sizeof(h.splitprof));
            std::memcpy(h.mlprof, back.mlprof, sizeof(h.mlprof));
            std::memcpy(h.rprof, back.rprof, sizeof(h.rprof));
            j.done += back.processed;
            {
                const bool was_sys = log_kernel[0] == 's';
                const int kk = (log_kernel[0] == 'p' || was_sys) ? 0 : (log_kernel[0] == 'f' ? 1 : 2);  // (the systolic kernel counts with the pipelined ones)
                t->kcount[kk] += (uint64_t)back.processed;
                t->kcount[3 + kk] += 1;
                if (was_sys) {
                    // (ids in the sequential engines' order: bb_tree_sys.inc, "BitFeature ids in the reference's order")
                    if (back.stop_reason != STOP_INTERNAL && back.ctr[C_IDS] > sys_ids_before)
                        rc = sys_renumber(t, back.out_leaf, (uint32_t)back.processed, sys_ids_before, back.ctr[C_IDS] - sys_ids_before, s);
                    if (rc != BBH_OK) break;
                    t->syscount[0] += (uint64_t)back.processed;
                    t->syscount[1] += 1;
                    if (back.stop_reason == STOP_SYS_RELAUNCH) t->syscount[2] += 1;
                    t->syscount[3] = (uint64_t)t->sys.G;
                    std::vector<unsigned long long> busy((size_t)t->sys.G);
                    if (hipMemcpy(busy.data(), t->sys.busy, busy.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                        for (unsigned long long b : busy) t->syscount[4] += b;
                        t->syscount[5] += busy[0];
                        unsigned long long mx = 0;
                        for (size_t q = 1; q < busy.size(); ++q) mx = std::max(mx, busy[q]);
                        t->syscount[6] += mx;
                    } else {
                        (void)hipGetLastError();
                    }
                    if (back.stop_reason == STOP_SYS_UNSUPPORTED) t->syscount[7] += 1;
                    {
                        static const bool sys_dbg_log = getenv("BBHIP_SYS_DEBUG") != nullptr;
                        uint32_t stale = 0;
                        if (sys_dbg_log && hipMemcpy(&stale, t->sys.ctl + SC_STALE_CTL, 4, hipMemcpyDeviceToHost) == hipSuccess && stale != 0)
                            fprintf(stderr, "[bbhip sys state] launch %u: %u polls saw a control word that a read-modify-write read did not, %u saw another launch's FINISH\n", t->sys.launch_id, stale & 0xFFFFu, stale >> 16);
                    }
                    static const bool sys_phases_log = getenv("BBHIP_SYS_PHASES") != nullptr;
                    if (sys_phases_log && back.processed > 0) {
                        // (phase-timer instance) cycles per element of the internal step: the root's owner, and the busiest owner of every other level
                        std::vector<unsigned long long> ph((size_t)t->sys.G * 13);
                        if (hipMemcpy(ph.data(), t->sys.busy, ph.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                            static const char* names[12] = {"lookup", "compare(hit)", "fill+compare(miss)", "guard", "send", "commit", "rest", "misses", "mailbox-rereads", "reread-spins", "ALONE", "ALONE-wait-cycles"};
                            for (int l = 0; l < t->sys.levels; ++l) {
                                int best = t->sys.lvl_first[l];
                                for (int w = t->sys.lvl_first[l]; w < t->sys.lvl_first[l] + t->sys.lvl_count[l]; ++w)
                                    if (ph[(size_t)w] > ph[(size_t)best]) best = w;
                                fprintf(stderr, "[bbhip sys phases] level %d: %d workgroups, busiest wg %d: busy %.0f cycles per launch element (%lld elements)", l, t->sys.lvl_count[l], best,
                                        (double)ph[(size_t)best] / (double)back.processed, (long long)back.processed);
                                if (l < t->sys.levels - 1)
                                    for (int i = 0; i < 12; ++i) fprintf(stderr, " %s %.0f", names[i], (double)ph[(size_t)t->sys.G + (size_t)best * 12 + i] / ((i >= 7 && i <= 10) ? 1.0 : (double)back.processed));
                                fprintf(stderr, "\n");
                            }
                        } else {
                            (void)hipGetLastError();
                        }
                    }
                }
                if (back.stop_reason == STOP_PIPE_UNSUPPORTED) t->kcount[6] += 1;
                if (back.stop_reason == STOP_NODES || back.stop_reason == STOP_CF8 || back.stop_reason == STOP_CF16 || back.stop_reason == STOP_CF32) t->kcount[7] += 1;
            }
            if (j.old_left > 0) j.old_left = std::max<int64_t>(0, j.old_left - back.processed);
            if (t->sys_off_left > 0 && log_kernel[0] != 's') t->sys_off_left = std::max<int64_t>(0, t->sys_off_left - back.processed);
            if (prof_tok != (size_t)-1) bb::prof_units(prof_tok, back.processed);  // elements this launch inserted
            j.stalls = (back.processed == 0 && back.stop_reason != STOP_PIPE_UNSUPPORTED && back.stop_reason != STOP_PIPE_NEEDS_ML) ? j.stalls + 1 : 0;
            if (j.stalls > 3) { rc = bb::fail(BBH_ERR_CAPACITY, "tree engine made no progress (stop reason %d)", back.stop_reason); break; }
            const int64_t left = j.n - j.done;
            // what a pool that ran out is grown to: half as much again, or what the elements that are left are expected to
            // need (pregrow's rates) if that is more
            uint64_t uleft_for_nodes = 0;
            auto more = [&](uint32_t used, uint32_t cap, uint64_t expect) -> uint32_t {
                if (tiny_pools()) return clamp30((uint64_t)cap + 8 + 2 * (uint64_t)h.ctr[C_DEPTH]);  // (one insertion's worst case fits)
                return clamp30(std::max<uint64_t>((uint64_t)cap + 1, (uint64_t)used + expect));  // (grow_cf takes the larger of this and its geometric step)
            };
            const uint64_t nblk = node_blocks((uint32_t)h.bf + 1);
            // (nodes: blocks; a pool that is compacted on the way - grow_nodes - takes `want - used` as the room asked for)
            auto more_nodes = [&]() -> uint32_t {
                if (tiny_pools()) return clamp30((uint64_t)h.ctr[C_NODES] + (8 + 2 * (uint64_t)h.ctr[C_DEPTH]) * nblk);
                const uint64_t expect = (uleft_for_nodes / (uint64_t)std::max(1, h.bf / 2) + 64) * nblk;
                return clamp30(std::max<uint64_t>((uint64_t)h.cap_nodes + h.cap_nodes / 2, (uint64_t)h.ctr[C_NODES] + expect));
            };
            const uint64_t uleft = (uint64_t)std::max<int64_t>(left, 0);
            uleft_for_nodes = uleft;
            switch (back.stop_reason) {
                case STOP_DONE: break;
                case STOP_NODES: rc = grow_nodes(t, more_nodes(), tiny_pools() ? (8 + 2 * (uint64_t)h.ctr[C_DEPTH]) * nblk : (uleft / (uint64_t)std::max(1, h.bf / 2) + 64) * nblk); break;
                case STOP_CF8: rc = grow_cf(t, 0, more(h.ctr[C_N8], h.cap8, uleft / 8 + 1024)); break;
                case STOP_CF16: rc = grow_cf(t, 1, more(h.ctr[C_N16], h.cap16, (j.width == 2 ? uleft : uleft / 64) + 64)); break;
                case STOP_CF32: rc = grow_cf(t, 2, more(h.ctr[C_N32], h.cap32, uleft / (uint64_t)std::max(1, h.bf / 6) + 256)); break;
                case STOP_DEPTH: rc = bb::fail(BBH_ERR_CAPACITY, "tree deeper than %d levels (or corrupt link)", MAXD); break;
                case STOP_RANGE: rc = bb::fail(BBH_ERR_INVALID, "n_samples exceeds 2^32-1 (engine limit)"); break;
                case STOP_PIPE_UNSUPPORTED:
                    // a stretch on the steady-state kernel, then the pipeline again.  A tree that keeps its unsupported shape
                    // would pay a launch that inserts nothing after every stretch: the stretch doubles (up to BBH_UNSUP_CAP = 65 536 elements)
                    // while the pipeline makes no progress and starts again at 1 024 as soon as it does
                    // (a tree that has been in the pipeline: informative levels above the leaf-parents are mostly short-lived
                    // there - a freshly split node's tracking row - so the first stretch is short; a new tree needs its first
                    // 8 192 elements to get the shape at all)
                    t->unsup_stretch = t->unsup_stretch == 0 ? 8192 : (back.processed > 0 ? 1024 : std::min<int64_t>(t->unsup_stretch * 2, BBH_UNSUP_CAP));
                    j.old_left = t->unsup_stretch;
                    break;
                case STOP_SYS_RELAUNCH: break;  // the root was split: the same kernel again, one level more
                case STOP_SYS_UNSUPPORTED: t->sys_off_left = 1 << 16; break;
                case STOP_PIPE_NEEDS_ML: t->pipe_ml = true; break;  // relaunched at once, with the multi-level instance
                case STOP_PIPE_PREFERS_SL: t->pipe_ml = false; break;  // ... and back (after a stint of >= PIPE_ML_STINT elements)
                case STOP_INTERNAL: {
                    if (log_kernel[0] == 's') {
                        uint32_t dbg[16] = {0};
                        (void)hipMemcpy(dbg, t->sys.ctl, sizeof(dbg), hipMemcpyDeviceToHost);
                        rc = bb::fail(BBH_ERR_HIP, "level-systolic kernel: internal error at bb_tree_sys.inc:%u (node %u child %u sent %u acked %u | fresh sent %u acked %u | level/miss/row %#x element %u)",
                                      back.giveup_line, dbg[8], dbg[9], dbg[10], dbg[11], dbg[12], dbg[13], dbg[14], dbg[15]);
                        if (getenv("BBHIP_SYS_DEBUG")) {  // where every workgroup was waiting
                            std::vector<unsigned long long> stw((size_t)t->sys.G * 15);
                            if (hipMemcpy(stw.data(), t->sys.busy, stw.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
                                for (int w = 0; w < t->sys.G; ++w) {
                                    const unsigned long long a = stw[(size_t)t->sys.G * 13 + (size_t)w * 2], b = stw[(size_t)t->sys.G * 13 + (size_t)w * 2 + 1];
                                    const unsigned kind = (unsigned)(a >> 32);
                                    if (kind != 4)
                                        fprintf(stderr, "[bbhip sys state] wg %d: %s node %u child %u want %u (owner index of child in the next level: %u)\n", w,
                                                kind == 1 ? "guard (pending below the child)" : kind == 2 ? "ALONE (waits for the child)" : kind == 3 ? "drain (all children)" : kind == 5 ? "in an internal step (node, element, alone)" : kind == 6 ? "in a leaf step (node, element, alone)" : "busy / never polled",
                                                (unsigned)a, (unsigned)(b >> 32), (unsigned)b, (unsigned)(((unsigned)(b >> 32)) / node_blocks((uint32_t)t->h.bf + 1)));
                                    if (kind >= 1 && kind <= 3) {  // the words the wait is about, as the host sees them now
                                        auto words = [&](uint32_t nd, const char* what) {
                                            unsigned long long mw = 0, upw = 0; uint32_t sw = 0, le = 0; NodeHdr hd{};
                                            if (nd >= t->h.cap_nodes) return;
                                            (void)hipMemcpy(&mw, t->sys.mail + nd, 8, hipMemcpyDeviceToHost);
                                            (void)hipMemcpy(&upw, t->sys.up + nd, 8, hipMemcpyDeviceToHost);
                                            (void)hipMemcpy(&sw, t->sys.sent + nd, 4, hipMemcpyDeviceToHost);
                                            (void)hipMemcpy(&le, t->sys.laste + nd, 4, hipMemcpyDeviceToHost);
                                            (void)hipMemcpy(&hd, t->h.node_hdr + nd, sizeof(hd), hipMemcpyDeviceToHost);
                                            fprintf(stderr, "[bbhip sys state]     %s %u: mail acked %u len %u res %u epoch %u | sent %u | last element+1 %u | up (node %u row %u) | hdr len %u leaf %#x\n", what, nd,
                                                    (unsigned)mw, (unsigned)((mw >> 32) & 0xFFFF), (unsigned)((mw >> 48) & 3), (unsigned)(mw >> 50), sw, le, (unsigned)(upw >> 32), (unsigned)upw, hd.len, hd.leaf);
                                        };
                                        words((unsigned)a, "node");
                                        words((unsigned)(b >> 32), "child");
                                        {   // where the tree (as the host sees it now) holds the child, and the node's first rows
                                            const uint32_t used = std::min<uint32_t>(t->h.cap_nodes, 1u << 22), nblk_ = node_blocks((uint32_t)t->h.bf + 1);
                                            std::vector<uint32_t> lk((size_t)used * NG);
                                            std::vector<NodeHdr> hh(used);
                                            if (hipMemcpy(lk.data(), t->h.node_link, lk.size() * 4, hipMemcpyDeviceToHost) == hipSuccess &&
                                                hipMemcpy(hh.data(), t->h.node_hdr, hh.size() * sizeof(NodeHdr), hipMemcpyDeviceToHost) == hipSuccess) {
                                                const uint32_t nd_ = (unsigned)a, ch_ = (unsigned)(b >> 32);
                                                if (nd_ < used) {
                                                    fprintf(stderr, "[bbhip sys state]     node %u rows' children:", nd_);
                                                    for (uint32_t r = 0; r < hh[nd_].len && r < 6; ++r) fprintf(stderr, " %u", lk[(size_t)nd_ * NG + r]);
                                                    fprintf(stderr, "\n");
                                                }
                                                for (uint32_t x = 0; x + nblk_ <= used; ++x) {
                                                    if ((hh[x].leaf & HW_LEAF) || hh[x].len == 0 || hh[x].len > (uint32_t)t->h.bf + 1 || hw_cap(hh[x].leaf) != (uint32_t)t->h.bf + 1) continue;
                                                    for (uint32_t r = 0; r < hh[x].len; ++r)
                                                        if (lk[(size_t)x * NG + r] == ch_) fprintf(stderr, "[bbhip sys state]     child %u is row %u of node %u (len %u)\n", ch_, r, x, hh[x].len);
                                                }
                                            }
                                        }
                                        if (kind == 3) {
                                            NodeHdr hd{};
                                            (void)hipMemcpy(&hd, t->h.node_hdr + (unsigned)a, sizeof(hd), hipMemcpyDeviceToHost);
                                            for (uint32_t r = 0; r < hd.len && r <= (uint32_t)t->h.bf; ++r) {
                                                uint32_t ch = 0, sw = 0; unsigned long long mw = 0;
                                                (void)hipMemcpy(&ch, t->h.node_link + (size_t)(unsigned)a * NG + r, 4, hipMemcpyDeviceToHost);
                                                if (ch >= t->h.cap_nodes) continue;
                                                (void)hipMemcpy(&sw, t->sys.sent + ch, 4, hipMemcpyDeviceToHost);
                                                (void)hipMemcpy(&mw, t->sys.mail + ch, 8, hipMemcpyDeviceToHost);
                                                if ((uint32_t)mw != sw) words(ch, "  behind: row's child");
                                            }
                                        }
                                    }
                                }
                                {   // ring positions: what a producer sent and its consumer has not taken
                                    std::vector<uint32_t> pos((size_t)t->sys.G * 2 * SYS_MAXPROD);
                                    if (hipMemcpy(pos.data(), (const uint8_t*)t->sys.busy + (size_t)t->sys.G * 15 * 8, pos.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
                                        for (int l = 0; l + 1 < t->sys.levels; ++l)
                                            for (int p = 0; p < t->sys.lvl_count[l]; ++p)
                                                for (int q = 0; q < t->sys.lvl_count[l + 1]; ++q) {
                                                    const int pw = t->sys.lvl_first[l] + p, cw = t->sys.lvl_first[l + 1] + q;
                                                    const uint32_t tail = pos[(size_t)pw * 2 * SYS_MAXPROD + SYS_MAXPROD + q], head = pos[(size_t)cw * 2 * SYS_MAXPROD + p];
                                                    if (tail != head) {
                                                        unsigned long long ab[2] = {0, 0};
                                                        (void)hipMemcpy(ab, t->sys.rings + (((size_t)cw * SYS_MAXPROD + (size_t)p) * SYS_R + (head & (SYS_R - 1u))) * 2, 16, hipMemcpyDeviceToHost);
                                                        fprintf(stderr, "[bbhip sys state] ring wg %d -> wg %d: sent %u taken %u; the slot the consumer is looking at: %016llx %016llx (launch id %u: gen %u alone %u node %u element %u | gen %u launch %u epoch %u)\n",
                                                                pw, cw, tail, head, ab[0], ab[1], t->sys.launch_id, (unsigned)(ab[0] >> 63), (unsigned)((ab[0] >> 62) & 1), (unsigned)((ab[0] >> 31) & 0x3FFFFFFF),
                                                                (unsigned)(ab[0] & 0x7FFFFFFF), (unsigned)(ab[1] >> 63), (unsigned)((ab[1] >> 32) & 0x7FFFFFFF), (unsigned)ab[1]);
                                                    }
                                                }
                                    }
                                }
                                for (int l = 0; l < t->sys.levels; ++l) fprintf(stderr, "[bbhip sys state] level %d: workgroups %d..%d\n", l, t->sys.lvl_first[l], t->sys.lvl_first[l] + t->sys.lvl_count[l] - 1);
                            }
                        }
                        break;
                    }
                    const unsigned int line = back.giveup_line;
                    rc = bb::fail(BBH_ERR_HIP, "pipelined kernel: a wait gave up (internal error; first at bb_tree_pipe.inc:%u)", line);
                    break;
                }
                default: rc = bb::fail(BBH_ERR_HIP, "unknown stop reason %d", back.stop_reason); break;
            }
        }
    }
    if (darr) (void)bb::dev_free(darr);
    return rc;
}

int run_insert(bbh_tree* t, const uint8_t* rows_dev, int64_t row_stride, const uint8_t* bufs_dev, int width, int64_t n,
               uint32_t* out_leaf_dev, hipStream_t s) {
    std::vector<Job> jobs(1);
    jobs[0] = Job{t, rows_dev, row_stride, bufs_dev, width, n, out_leaf_dev, 0, 0};
    return run_insert_multi(jobs, s);
}


#include "bb_tree_batch.inc"

// walk the leaf chain on the host (bitbirch.py:886-893): positions -> (node, row)
int build_chain(bbh_tree* t) {
    if (t->chain_valid) return BBH_OK;
    TreeDev& h = t->h;
    const uint32_t nn = (t->lazy_pools || h.node_hdr == nullptr) ? 0u : h.ctr[C_NODES];  // (a tree that never received anything owns no pools)
    // Only the headers that start a live LEAF come to the host (the header pool has one entry per block of four
    // rows, 64 per full-capacity node at bf 254 - copying all of them was 1 GB for a million unsealed nodes): a counting
    // pass, then {id, len, next} of every such header, sorted by id on the host and walked from the first leaf.
    struct LeafHdr { uint32_t id, len, next; };
    std::vector<LeafHdr> lh;
    if (nn) {
        uint32_t* d_count = nullptr;
        BB_HIP(bb::dev_alloc(&d_count, 64));
        BB_HIP(hipMemset(d_count, 0, 4));
        hipLaunchKernelGGL(k_leaf_headers, dim3((nn + 255) / 256), dim3(256), 0, 0, (const NodeHdr*)h.node_hdr, nn, d_count, (uint32_t*)nullptr, 0u);
        uint32_t count = 0;
        hipError_t e = hipMemcpy(&count, d_count, 4, hipMemcpyDeviceToHost);
        uint32_t* d_out = nullptr;
        if (e == hipSuccess && count) e = bb::dev_alloc(&d_out, (size_t)count * 12);
        if (e == hipSuccess && count) {
            e = hipMemset(d_count, 0, 4);
            if (e == hipSuccess) {
                hipLaunchKernelGGL(k_leaf_headers, dim3((nn + 255) / 256), dim3(256), 0, 0, (const NodeHdr*)h.node_hdr, nn, d_count, d_out, count);
                lh.resize(count);
                e = hipMemcpy(lh.data(), d_out, (size_t)count * 12, hipMemcpyDeviceToHost);
            }
        }
        if (d_out) (void)bb::dev_free(d_out);
        (void)bb::dev_free(d_count);
        BB_HIP(e);
        std::sort(lh.begin(), lh.end(), [](const LeafHdr& a, const LeafHdr& b) { return a.id < b.id; });
    }
    t->chain_nodes.clear();
    t->chain_rows.clear();
    uint32_t nd = h.ctr[C_FIRST_LEAF];
    size_t guard = 0;
    while (nd != NONE && nd < nn && guard++ <= lh.size()) {
        const auto it = std::lower_bound(lh.begin(), lh.end(), nd, [](const LeafHdr& a, uint32_t id) { return a.id < id; });
        if (it == lh.end() || it->id != nd) return bb::fail(BBH_ERR_HIP, "leaf chain points at block %u, which starts no live leaf", nd);
        for (uint32_t r = 0; r < it->len; ++r) {
            t->chain_nodes.push_back(nd);
            t->chain_rows.push_back(r);
        }
        nd = it->next;
    }
    const size_t k = t->chain_nodes.size();
    if (k > t->d_chain_cap) {
        if (t->d_chain_nodes) (void)bb::dev_free(t->d_chain_nodes);
        if (t->d_chain_rows) (void)bb::dev_free(t->d_chain_rows);
        t->d_chain_nodes = t->d_chain_rows = nullptr;
        BB_HIP(bb::dev_alloc(&t->d_chain_nodes, k * 4));
        BB_HIP(bb::dev_alloc(&t->d_chain_rows, k * 4));
        t->d_chain_cap = k;
    }
    if (k) {
        BB_HIP(hipMemcpy(t->d_chain_nodes, t->chain_nodes.data(), k * 4, hipMemcpyHostToDevice));
        BB_HIP(hipMemcpy(t->d_chain_rows, t->chain_rows.data(), k * 4, hipMemcpyHostToDevice));
    }
    t->chain_valid = true;
    return BBH_OK;
}


}  // namespace

extern "C" int bbh_tree_create(bbh_tree** out, int32_t branching_factor, double threshold, int32_t criterion,
                               double tolerance, const double* tol_table, int64_t tol_len, int32_t n_features,
                               int32_t device) {
    if (out == nullptr) return bb::fail(BBH_ERR_INVALID, "null output handle");
    *out = nullptr;
    BB_TRY(bb::ensure_device());
    if (n_features < 8 || n_features % 8 != 0)
        return bb::fail(BBH_ERR_INVALID, "Only n_features divisible by 8 is supported");
    if (n_features > 8192) return bb::fail(BBH_ERR_INVALID, "n_features > 8192 is not supported by the tree engine");
    if (branching_factor < 2 || branching_factor > MAX_BF)
        return bb::fail(BBH_ERR_INVALID, "branching_factor must be in [2, %d]", MAX_BF);
    if (criterion < 0 || criterion > BBH_CRIT_NEVER) return bb::fail(BBH_ERR_INVALID, "unknown merge criterion %d", criterion);
    BB_HIP(hipSetDevice(device));
    bbh_tree* t = new bbh_tree();
    t->device = device;
    t->h.thr = threshold;
    t->h.crit = criterion;
    t->h.tolerance = tolerance;
    int rc = configure(t, branching_factor, n_features);
    if (rc == BBH_OK) rc = set_tol(t, tol_table, tol_len);
    if (rc == BBH_OK) {
        hipError_t e = bb::dev_alloc(&t->d, sizeof(TreeDev));
        if (e != hipSuccess) rc = bb::fail(BBH_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e));
    }
    if (rc == BBH_OK) rc = init_empty(t);
    if (rc != BBH_OK) {
        bbh_tree_destroy(t);
        return rc;
    }
    *out = t;
    return BBH_OK;
}

extern "C" int bbh_tree_destroy(bbh_tree* t) {
    if (!t) return BBH_OK;
    free_pools(t);
    if (t->d) (void)bb::dev_free(t->d);
    if (t->d_tol) (void)bb::dev_free(t->d_tol);
    if (t->d_chain_nodes) (void)bb::dev_free(t->d_chain_nodes);
    if (t->d_chain_rows) (void)bb::dev_free(t->d_chain_rows);
    sys_free(t);
    delete t;
    return BBH_OK;
}

extern "C" int bbh_tree_set_merge(bbh_tree* t, int32_t criterion, double tolerance, const double* tol_table,
                                  int64_t tol_len, double threshold, int32_t branching_factor) {
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    if (criterion < 0 || criterion > BBH_CRIT_NEVER) return bb::fail(BBH_ERR_INVALID, "unknown merge criterion %d", criterion);
    t->h.crit = criterion;
    t->h.tolerance = tolerance;
    t->h.thr = threshold;
    BB_TRY(set_tol(t, tol_table, tol_len));
    if (branching_factor != t->h.bf) {
        const bool empty = t->h.ctr[C_NODES] == node_blocks((uint32_t)t->h.bf + 1) && t->h.ctr[C_IDS] == 0;
        if (!empty)
            return bb::fail(BBH_ERR_STATE, "branching_factor can only change on an empty tree: call reset() first");
        if (branching_factor < 2 || branching_factor > MAX_BF)
            return bb::fail(BBH_ERR_INVALID, "branching_factor must be in [2, %d]", MAX_BF);
        free_pools(t);
        BB_TRY(configure(t, branching_factor, t->h.F));
        BB_TRY(init_empty(t));
    }
    return BBH_OK;
}

extern "C" int bbh_tree_reset(bbh_tree* t) {
    if (t) { t->sys_pref = false; t->sys_off_left = 0; }
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    return init_empty(t);  // pools are kept and reused
}

// ---------------------------------------------------------------------------------------
// Streaming ingest of host-resident rows (fingerprint files are memory-mapped by the caller;
// the reference walks them with mmap + madvise, _memory.py:74-126).  Two slabs in HBM and two
// pinned bounce buffers: while the tree kernel consumes slab i, a helper thread pages slab i+1
// in from the (pageable, possibly file-backed) source, and a dedicated copy stream moves it
// over PCIe.  The clustering kernel never waits for the host after the first slab, and the
// device footprint of the input is 2 slabs regardless of the file size.
// ---------------------------------------------------------------------------------------
struct HostSlabs {
    static constexpr size_t kSlabBytes = 64ull << 20;
    int device = 0;
    const uint8_t* src = nullptr;
    size_t unit = 0;           // bytes per row (stride)
    int64_t total = 0, per = 0;  // rows in all / per slab
    uint8_t* pin[2] = {nullptr, nullptr};
    uint8_t* dev[2] = {nullptr, nullptr};
    hipStream_t cs = nullptr;
    std::future<int> pending;
    int64_t next = 0;  // first row of the slab being fetched

    int init(int device_, const void* src_, size_t unit_, int64_t total_) {
        device = device_; src = (const uint8_t*)src_; unit = unit_; total = total_;
        size_t slab_bytes = kSlabBytes;
        if (const char* e = getenv("BBHIP_SLAB_KB")) slab_bytes = std::max<size_t>(1, (size_t)atoll(e)) << 10;  // tests
        per = std::max<int64_t>(1, (int64_t)(slab_bytes / unit));
        per = std::min(per, total);
        const int nbuf = total > per ? 2 : 1;
        for (int i = 0; i < nbuf; ++i) {
            BB_HIP(hipHostMalloc((void**)&pin[i], (size_t)per * unit, hipHostMallocDefault));
            BB_HIP(bb::dev_alloc(&dev[i], (size_t)per * unit));
        }
        BB_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        return BBH_OK;
    }
    void fetch(int64_t first) {  // start moving rows [first, first + per) into buffer (first / per) & 1
        next = first;
        if (first >= total) return;
        const int b = (int)((first / per) & 1);
        const int64_t m = std::min(per, total - first);
        auto job = [this, b, first, m]() -> int {
            if (hipSetDevice(device) != hipSuccess) return 1;
            std::memcpy(pin[b], src + (size_t)first * unit, (size_t)m * unit);  // page-in happens here
            if (hipMemcpyAsync(dev[b], pin[b], (size_t)m * unit, hipMemcpyHostToDevice, cs) != hipSuccess) return 2;
            return hipStreamSynchronize(cs) == hipSuccess ? 0 : 3;
        };
        try {
            pending = std::async(std::launch::async, job);
        } catch (...) {  // no thread to be had: do the transfer inline, nothing may escape the C ABI
            pending = std::async(std::launch::deferred, job);
        }
    }
    // rows [first, first + *m) are resident at the returned device pointer; the following slab
    // is already on its way when this returns
    int acquire(int64_t first, const uint8_t** out, int64_t* m) {
        if (!pending.valid() || next != first) fetch(first);
        const int rc = pending.get();
        if (rc != 0) return bb::fail(BBH_ERR_HIP, "host slab transfer failed (stage %d)", rc);
        *out = dev[(first / per) & 1];
        *m = std::min(per, total - first);
        fetch(first + per);
        return BBH_OK;
    }
    ~HostSlabs() {
        if (pending.valid()) (void)pending.get();
        for (int i = 0; i < 2; ++i) {
            if (pin[i]) (void)hipHostFree(pin[i]);
            if (dev[i]) (void)bb::dev_free(dev[i]);
        }
        if (cs) (void)hipStreamDestroy(cs);
    }
};

extern "C" int bbh_tree_fit_packed(bbh_tree* t, const uint8_t* rows, int64_t n, int64_t row_stride,
                                   uint32_t* out_leaf, void* stream) {
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    if (n < 0 || row_stride < t->h.nbytes) return bb::fail(BBH_ERR_INVALID, "rows must be (n, n_features/8) uint8");
    if (n == 0) return BBH_OK;
    BB_HIP(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    BB_TRY(pregrow(t, n, 0));
    bb::DevOut o;
    BB_TRY(o.init(out_leaf, (size_t)n * 4));
    const char* benv = getenv("BBHIP_BATCH");
    const int batch = benv ? atoi(benv) : 0;
    if (bb::is_device_ptr(rows)) {
        if (batch > 0) BB_TRY(run_insert_batched(t, rows, row_stride, n, (uint32_t*)o.dev, batch, s));
        else BB_TRY(run_insert(t, rows, row_stride, nullptr, 0, n, (uint32_t*)o.dev, s));
    } else {
        HostSlabs slabs;
        BB_TRY(slabs.init(t->device, rows, (size_t)row_stride, n));
        for (int64_t off = 0; off < n;) {
            const uint8_t* d = nullptr;
            int64_t m = 0;
            BB_TRY(slabs.acquire(off, &d, &m));
            uint32_t* o_off = o.dev ? (uint32_t*)o.dev + off : nullptr;
            if (batch > 0) BB_TRY(run_insert_batched(t, d, row_stride, m, o_off, batch, s));
            else BB_TRY(run_insert(t, d, row_stride, nullptr, 0, m, o_off, s));
            off += m;
        }
    }
    BB_TRY(o.finish(s));
    BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

// Insert m uint8-wide BitFeature buffers that sit in HBM.  Runs of singletons (n_samples == 1, the
// bulk of every table: 92 % of the BitFeatures of the 1 M-row benchmark) are packed on the device and
// inserted as fingerprints - the same BitFeature (ls = bits, n = 1, centroid = the bits; reference
// bitbirch.py:412-421 vs :422-448) through the cheaper path; everything else goes as buffers.
// `n_col`: the n_samples column on the host.
static int insert_u8_buffers(bbh_tree* t, const uint8_t* d, int64_t m, const uint8_t* n_col, uint32_t* out, hipStream_t s) {
    const int64_t kMinRun = 1024;
    const size_t row_bytes = (size_t)t->h.F + 1;
    const int nbytes = t->h.nbytes;
    uint8_t* packed = nullptr;
    int rc = BBH_OK;
    for (int64_t lo = 0; lo < m && rc == BBH_OK;) {
        // extend a singleton run as far as it goes; otherwise collect buffers up to the next long singleton run
        int64_t hi = lo;
        while (hi < m && n_col[hi] == 1) ++hi;
        if (hi - lo >= kMinRun) {
            if (!packed) {
                hipError_t e = bb::dev_alloc(&packed, (size_t)m * nbytes);
                if (e != hipSuccess) return bb::fail(BBH_ERR_HIP, "dev_alloc: %s", hipGetErrorString(e));
            }
            // (at most 2^22 rows per launch: the global size of a launch is a 32-bit number)
            for (int64_t q = lo; q < hi; q += (4ll << 20)) {
                const long long cnt = (long long)std::min<int64_t>(4ll << 20, hi - q);
                const long long total = cnt * nbytes;
                hipLaunchKernelGGL(k_pack_singletons, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d + (size_t)q * row_bytes,
                                   cnt, t->h.F, nbytes, packed + (size_t)(q - lo) * (size_t)nbytes);
            }
            rc = run_insert(t, packed, nbytes, nullptr, 0, hi - lo, out ? out + lo : nullptr, s);
            lo = hi;
            continue;
        }
        // short singleton runs stay with their neighbours
        int64_t run1 = 0;
        hi = lo;
        while (hi < m) {
            run1 = n_col[hi] == 1 ? run1 + 1 : 0;
            ++hi;
            if (run1 >= kMinRun) { hi -= run1; break; }
        }
        if (hi == lo) hi = lo + 1;
        // (every buffer of this run that is appended takes a uint8 slot - pregrow reckoned with an eighth of the call's elements,
        // which is right for its singleton tail and eight times too little here: the kernel stopped on the exhausted pool, the
        // host copied all of it into a larger one and relaunched, several times per table)
        if (!tiny_pools()) rc = grow_cf(t, 0, clamp30((uint64_t)t->h.ctr[C_N8] + (uint64_t)(hi - lo) + 64));
        if (rc != BBH_OK) break;
        rc = run_insert(t, nullptr, 0, d + (size_t)lo * row_bytes, 1, hi - lo, out ? out + lo : nullptr, s);
        lo = hi;
    }
    if (packed) {
        (void)hipStreamSynchronize(s);
        bb::dev_free(packed);
    }
    return rc;
}

extern "C" int bbh_tree_fit_buffers(bbh_tree* t, const void* bufs, int32_t width, int64_t k, uint32_t* out_leaf,
                                    void* stream) {
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    if (width != 1 && width != 2 && width != 4 && width != 8) return bb::fail(BBH_ERR_INVALID, "buffer element width must be 1, 2, 4 or 8");
    if (k < 0) return bb::fail(BBH_ERR_INVALID, "negative buffer count");
    if (k == 0) return BBH_OK;
    BB_HIP(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    BB_TRY(pregrow(t, k, width));
    const size_t row_bytes = ((size_t)t->h.F + 1) * width;
    bb::DevOut o;
    BB_TRY(o.init(out_leaf, (size_t)k * 4));
    static const bool no_singleton_path = getenv("BBHIP_NO_SINGLETON_PATH") != nullptr;
    const bool split_runs = width == 1 && !no_singleton_path;
    if (bb::is_device_ptr(bufs)) {
        if (split_runs) {
            // the n_samples column: gathered on the device, one contiguous copy (a 1-byte-wide hipMemcpy2D over
            // 360 k rows took half a second)
            std::vector<uint8_t> n_col((size_t)k);
            uint8_t* col = nullptr;
            BB_HIP(bb::dev_alloc(&col, (size_t)k));
            hipLaunchKernelGGL(k_gather_n_col, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, s, (const uint8_t*)bufs, (long long)k,
                               t->h.F, col);
            hipError_t ce = hipMemcpyAsync(n_col.data(), col, (size_t)k, hipMemcpyDeviceToHost, s);
            if (ce == hipSuccess) ce = hipStreamSynchronize(s);
            (void)bb::dev_free(col);
            BB_HIP(ce);
            BB_TRY(insert_u8_buffers(t, (const uint8_t*)bufs, k, n_col.data(), (uint32_t*)o.dev, s));
        } else {
            BB_TRY(run_insert(t, nullptr, 0, (const uint8_t*)bufs, width, k, (uint32_t*)o.dev, s));
        }
    } else {
        HostSlabs slabs;
        BB_TRY(slabs.init(t->device, bufs, row_bytes, k));
        std::vector<uint8_t> n_col;
        for (int64_t off = 0; off < k;) {
            const uint8_t* d = nullptr;
            int64_t m = 0;
            BB_TRY(slabs.acquire(off, &d, &m));
            uint32_t* o_off = o.dev ? (uint32_t*)o.dev + off : nullptr;
            if (split_runs) {
                n_col.resize((size_t)m);
                const uint8_t* src = (const uint8_t*)bufs + (size_t)off * row_bytes + t->h.F;
                for (int64_t i = 0; i < m; ++i) n_col[(size_t)i] = src[(size_t)i * row_bytes];
                BB_TRY(insert_u8_buffers(t, d, m, n_col.data(), o_off, s));
            } else {
                BB_TRY(run_insert(t, nullptr, 0, d, width, m, o_off, s));
            }
            off += m;
        }
    }
    BB_TRY(o.finish(s));
    BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

// Several independent trees (multiround shards) in one launch: one workgroup per tree.
extern "C" int bbh_trees_fit_packed(bbh_tree** trees, int32_t n_trees, const uint8_t* const* rows, const int64_t* n,
                                    const int64_t* row_stride, uint32_t* const* out_leaf, void* stream) {
    if (!trees || n_trees < 0 || !rows || !n || !row_stride) return bb::fail(BBH_ERR_INVALID, "null argument");
    if (n_trees == 0) return BBH_OK;
    hipStream_t s = (hipStream_t)stream;
    std::vector<Job> jobs;
    std::vector<bb::DevIn> ins((size_t)n_trees);
    std::vector<bb::DevOut> outs((size_t)n_trees);
    const int device = trees[0]->device;
    BB_HIP(hipSetDevice(device));
    for (int32_t i = 0; i < n_trees; ++i) {
        bbh_tree* t = trees[i];
        if (!t || t->device != device) return bb::fail(BBH_ERR_INVALID, "all trees of one call must live on one device");
        if (n[i] < 0 || row_stride[i] < t->h.nbytes) return bb::fail(BBH_ERR_INVALID, "rows must be (n, n_features/8) uint8");
        if (n[i] == 0) continue;
        BB_TRY(pregrow(t, n[i], 0));
        BB_TRY(ins[(size_t)i].init(rows[i], (size_t)(n[i] * row_stride[i]), s));
        BB_TRY(outs[(size_t)i].init(out_leaf ? out_leaf[i] : nullptr, (size_t)n[i] * 4));
        jobs.push_back(Job{t, (const uint8_t*)ins[(size_t)i].dev, row_stride[i], nullptr, 0, n[i],
                           (uint32_t*)outs[(size_t)i].dev, 0, 0});
    }
    BB_TRY(run_insert_multi(jobs, s));
    for (auto& o : outs) BB_TRY(o.finish(s));
    BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

// The same for BitFeature buffers: the trees of one merge round (reference multiround.py:240-264
// builds them one process each), each inserting its own table in one shared launch.
extern "C" int bbh_trees_fit_buffers(bbh_tree** trees, int32_t n_trees, const void* const* bufs, const int32_t* width,
                                     const int64_t* k, uint32_t* const* out_leaf, void* stream) {
    if (!trees || n_trees < 0 || !bufs || !width || !k) return bb::fail(BBH_ERR_INVALID, "null argument");
    if (n_trees == 0) return BBH_OK;
    hipStream_t s = (hipStream_t)stream;
    std::vector<Job> jobs;
    std::vector<bb::DevIn> ins((size_t)n_trees);
    std::vector<bb::DevOut> outs((size_t)n_trees);
    const int device = trees[0]->device;
    BB_HIP(hipSetDevice(device));
    for (int32_t i = 0; i < n_trees; ++i) {
        bbh_tree* t = trees[i];
        if (!t || t->device != device) return bb::fail(BBH_ERR_INVALID, "all trees of one call must live on one device");
        const int w = width[i];
        if (w != 1 && w != 2 && w != 4 && w != 8) return bb::fail(BBH_ERR_INVALID, "buffer element width must be 1, 2, 4 or 8");
        if (k[i] < 0) return bb::fail(BBH_ERR_INVALID, "negative buffer count");
        if (k[i] == 0) continue;
        BB_TRY(pregrow(t, k[i], w));
        const size_t row_bytes = ((size_t)t->h.F + 1) * (size_t)w;
        BB_TRY(ins[(size_t)i].init(bufs[i], (size_t)k[i] * row_bytes, s));
        BB_TRY(outs[(size_t)i].init(out_leaf ? out_leaf[i] : nullptr, (size_t)k[i] * 4));
        jobs.push_back(Job{t, nullptr, 0, (const uint8_t*)ins[(size_t)i].dev, w, k[i], (uint32_t*)outs[(size_t)i].dev, 0, 0});
    }
    BB_TRY(run_insert_multi(jobs, s));
    for (auto& o : outs) BB_TRY(o.finish(s));
    BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

extern "C" int bbh_tree_leaf_count(bbh_tree* t, int64_t* out) {
    if (!t || !out) return bb::fail(BBH_ERR_INVALID, "null argument");
    BB_HIP(hipSetDevice(t->device));
    BB_TRY(build_chain(t));
    *out = (int64_t)t->chain_nodes.size();
    return BBH_OK;
}

static int gather(bbh_tree* t, const uint32_t* d_nodes, const uint32_t* d_rows, int64_t m, int width, void* bufs,
                  int ls_only, uint8_t* cents, uint64_t* ns, uint32_t* ids) {
    if (m == 0) return BBH_OK;
    TreeDev& h = t->h;
    bb::DevOut ob, oc, on, oi;
    const size_t cols = (size_t)h.F + (ls_only ? 0 : 1);
    BB_TRY(ob.init(bufs, bufs ? (size_t)m * cols * width : 0));
    BB_TRY(oc.init(cents, cents ? (size_t)m * h.nbytes : 0));
    BB_TRY(on.init(ns, ns ? (size_t)m * 8 : 0));
    BB_TRY(oi.init(ids, ids ? (size_t)m * 4 : 0));
    BB_HIP(hipMemcpy(t->d, &h, sizeof(TreeDev), hipMemcpyHostToDevice));
    {
        bb::ProfScope ps("gather_leaves", nullptr);
        // (a launch's global size - workgroups x 256 threads - is a 32-bit number: more than 2^24 leaves in ONE launch ran
        // only m mod 2^24 of them, silently; found at 20 M leaf BitFeatures in round 4)
        const int64_t kMax = 4ll << 20;
        const int w = width ? width : 1;
        for (int64_t lo = 0; lo < m; lo += kMax) {
            const int64_t cnt = std::min(kMax, m - lo);
            hipLaunchKernelGGL(k_gather_leaves, dim3((unsigned)cnt), dim3(256), 0, nullptr, t->d, d_nodes + lo, d_rows + lo,
                               (long long)cnt, w, ob.dev ? (uint8_t*)ob.dev + (size_t)lo * cols * (size_t)w : nullptr, ls_only,
                               oc.dev ? (uint8_t*)oc.dev + (size_t)lo * (size_t)h.nbytes : nullptr,
                               on.dev ? (unsigned long long*)on.dev + lo : nullptr, oi.dev ? (uint32_t*)oi.dev + lo : nullptr);
            BB_HIP(hipGetLastError());
        }
    }
    BB_TRY(ob.finish(nullptr));
    BB_TRY(oc.finish(nullptr));
    BB_TRY(on.finish(nullptr));
    BB_TRY(oi.finish(nullptr));
    BB_HIP(hipDeviceSynchronize());
    return BBH_OK;
}

extern "C" int bbh_tree_export_leaves(bbh_tree* t, uint32_t* leaf_ids, uint64_t* n_samples, uint8_t* packed_centroids,
                                      void* linear_sums, int32_t ls_width) {
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    if (linear_sums && ls_width != 1 && ls_width != 2 && ls_width != 4 && ls_width != 8)
        return bb::fail(BBH_ERR_INVALID, "ls_width must be 1, 2, 4 or 8");
    BB_HIP(hipSetDevice(t->device));
    BB_TRY(build_chain(t));
    return gather(t, t->d_chain_nodes, t->d_chain_rows, (int64_t)t->chain_nodes.size(), ls_width, linear_sums, 1,
                  packed_centroids, n_samples, leaf_ids);
}

// leaves at `positions` (chain order): BitFeature buffer rows (`bufs`, width bytes per value) and / or packed centroid rows
static int gather_positions(bbh_tree* t, const int64_t* positions, int64_t m, int32_t width, void* bufs, uint8_t* cents) {
    BB_HIP(hipSetDevice(t->device));
    BB_TRY(build_chain(t));
    const int64_t k = (int64_t)t->chain_nodes.size();
    std::vector<uint32_t> nodes((size_t)m), rows((size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        if (positions[i] < 0 || positions[i] >= k) return bb::fail(BBH_ERR_INVALID, "leaf position %lld out of range", (long long)positions[i]);
        nodes[(size_t)i] = t->chain_nodes[(size_t)positions[i]];
        rows[(size_t)i] = t->chain_rows[(size_t)positions[i]];
    }
    uint32_t *dn = nullptr, *dr = nullptr;
    BB_HIP(bb::dev_alloc(&dn, (size_t)m * 4));
    BB_HIP(bb::dev_alloc(&dr, (size_t)m * 4));
    int rc = BBH_OK;
    if (hipMemcpy(dn, nodes.data(), (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dr, rows.data(), (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = bb::fail(BBH_ERR_HIP, "gather: H2D failed");
    if (rc == BBH_OK) rc = gather(t, dn, dr, m, width, bufs, 0, cents, nullptr, nullptr);
    (void)bb::dev_free(dn);
    (void)bb::dev_free(dr);
    return rc;
}

extern "C" int bbh_tree_gather_buffers(bbh_tree* t, const int64_t* positions, int64_t m, int32_t width, void* out) {
    if (!t || (m > 0 && (!positions || !out))) return bb::fail(BBH_ERR_INVALID, "null argument");
    if (width != 1 && width != 2 && width != 4 && width != 8) return bb::fail(BBH_ERR_INVALID, "width must be 1, 2, 4 or 8");
    if (m == 0) return BBH_OK;
    return gather_positions(t, positions, m, width, out, nullptr);
}

extern "C" int bbh_tree_gather_centroids(bbh_tree* t, const int64_t* positions, int64_t m, uint8_t* out) {
    if (!t || (m > 0 && (!positions || !out))) return bb::fail(BBH_ERR_INVALID, "null argument");
    if (m == 0) return BBH_OK;
    return gather_positions(t, positions, m, 0, nullptr, out);
}

extern "C" int bbh_tree_kernel_counts(bbh_tree* t, uint64_t* out8) {
    if (!t || !out8) return bb::fail(BBH_ERR_INVALID, "null argument");
    for (int i = 0; i < 8; ++i) out8[i] = t->kcount[i];
    return BBH_OK;
}

// the level-systolic kernel's record: elements, launches, relaunches after a root split, workgroups of the last launch, busy
// shader cycles summed over all workgroups / of workgroup 0 (the root's owner) / of the busiest other workgroup, refusals
extern "C" int bbh_tree_sys_counts(bbh_tree* t, uint64_t* out8) {
    if (!t || !out8) return bb::fail(BBH_ERR_INVALID, "null argument");
    for (int i = 0; i < 8; ++i) out8[i] = t->syscount[i];
    return BBH_OK;
}

extern "C" int bbh_tree_memory(bbh_tree* t, uint64_t* out8) {
    if (!t || !out8) return bb::fail(BBH_ERR_INVALID, "null argument");
    const TreeDev& h = t->h;
    const size_t F = (size_t)h.F;
    out8[0] = (uint64_t)h.cap_nodes * block_bytes(h);
    out8[1] = (uint64_t)std::min(h.cap_nodes, h.ctr[C_NODES]) * block_bytes(h);
    out8[2] = (uint64_t)h.cap8 * F + (uint64_t)h.cap16 * F * 2 + (uint64_t)h.cap32 * F * 4;
    out8[3] = std::max(t->peak_bytes, pool_bytes(h));
    out8[4] = t->gc_runs;
    out8[5] = t->gc_sealed;
    out8[6] = t->gc_full;
    out8[7] = h.stats[7];
    return BBH_OK;
}

extern "C" int bbh_tree_compact(bbh_tree* t, int32_t seal) {
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    if (t->lazy_pools) return BBH_OK;  // (nothing was ever inserted: no pools)
    BB_HIP(hipSetDevice(t->device));
    BB_HIP(hipDeviceSynchronize());
    return gc_nodes(t, 0, seal != 0 ? 1 : 0);
}

extern "C" int bbh_tree_stats(bbh_tree* t, uint64_t* out8) {
    if (!t || !out8) return bb::fail(BBH_ERR_INVALID, "null argument");
    for (int i = 0; i < 7; ++i) out8[i] = t->h.stats[i];
    out8[7] = t->h.ctr[C_IDS];
    if (getenv("BBHIP_PIPE_AUDIT")) {
        unsigned long long runs = 0;
        (void)hipMemcpyFromSymbol(&runs, HIP_SYMBOL(g_pipe_audit_runs), sizeof(runs));
        fprintf(stderr, "[bbhip pipe audit] %llu slots audited at run ends so far (this process), no difference\n", runs);
    }
    if (getenv("BBHIP_PIPE_PHASES")) {
        static const char* nm[16] = {"router:setup", "router:wait", "router:compare", "router:commit", "router:drain", "leaf:wait", "leaf:fill",
                                     "leaf:compare", "leaf:cf+dot", "leaf:decide+apply", "all:flush", "all:cold", "#runs", "#router-stale", "#leaf-pre-hits", "#leaf-stale"};
        const double n = (double)(t->h.stats[2] + t->h.stats[3]);
        fprintf(stderr, "[bbhip pipe phases, per insert]");
        for (int i = 0; i < 16; ++i) fprintf(stderr, " %s=%.*f", nm[i], i < 12 ? 0 : 4, n > 0 ? (double)t->h.phase[i] / n : 0.0);
        static const char* kd[3] = {"pre-compared, best row folded", "own", "pre-compared, every row checked"};
        fprintf(stderr, "\n[bbhip pipe leaf compares]");
        for (int i = 0; i < 3; ++i)
            fprintf(stderr, " %s: %.3f/insert x %.0f cycles", kd[i], n > 0 ? (double)t->h.sphase[2 * i + 1] / n : 0.0,
                    t->h.sphase[2 * i + 1] ? (double)t->h.sphase[2 * i] / (double)t->h.sphase[2 * i + 1] : 0.0);
        fprintf(stderr, "\n");
        fprintf(stderr, "[bbhip pipe router waits, per insert] ring entry %.0f, row + pre-compare of wave 2 %.0f, pending jobs of a nearly full leaf %.0f, "
                "decision on a full leaf %.0f\n", n > 0 ? (double)t->h.rprof[0] / n : 0.0, n > 0 ? (double)t->h.rprof[1] / n : 0.0,
                n > 0 ? (double)t->h.rprof[2] / n : 0.0, n > 0 ? (double)t->h.rprof[3] / n : 0.0);
        if (t->h.splitprof[8]) {
            const double ns = (double)t->h.splitprof[8];
            fprintf(stderr, "[bbhip pipe leaf splits] %.4f/insert x %.0f cycles:", n > 0 ? ns / n : 0.0, (double)t->h.splitprof[9] / ns);
            for (int i = 0; i < 8; ++i) fprintf(stderr, " s%d=%.0f", i, (double)t->h.splitprof[i] / ns);
            fprintf(stderr, "\n");
        }
        if (t->pipe_ml)
            fprintf(stderr, "[bbhip pipe multi-level router] upper-slot fills %.3f/insert x %.0f cycles; tracking levels committed %.3f/insert, "
                    "cluster-feature cache misses %.3f/insert; router waiting for a level's update by a helper wave %.0f cycles/insert, for "
                    "all of them and their stores (before a fill) %.0f; of a fill: loads %.0f cycles, slot + corrections %.0f\n", n > 0 ? (double)t->h.sphase[7] / n : 0.0,
                    t->h.sphase[7] ? (double)t->h.sphase[6] / (double)t->h.sphase[7] : 0.0, n > 0 ? (double)t->h.mlprof[1] / n : 0.0,
                    n > 0 ? (double)t->h.mlprof[0] / n : 0.0, n > 0 ? (double)t->h.mlprof[2] / n : 0.0, n > 0 ? (double)t->h.mlprof[3] / n : 0.0,
                    t->h.sphase[7] ? (double)t->h.mlprof[4] / (double)t->h.sphase[7] : 0.0, t->h.sphase[7] ? (double)t->h.mlprof[5] / (double)t->h.sphase[7] : 0.0);
    }
    if (getenv("BBHIP_PHASES")) {
        fprintf(stderr, "[bbhip phases, cycles/insert]");
        const double n = (double)(t->h.stats[2] + t->h.stats[3]);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " p%d=%.0f", i, n > 0 ? (double)t->h.phase[i] / n : 0.0);
        static const char* cls[3] = {"fill/miss", "tag-only", "compare"};  // (k_tree_fast; k_tree_insert: miss, zero-skip, mirror-hit)
        fprintf(stderr, "\n[bbhip p0 detail] loop-top barrier+checks=%.0f wait-for-prefetched-row=%.0f", n > 0 ? (double)t->h.phase[14] / n : 0.0,
                n > 0 ? (double)t->h.phase[15] / n : 0.0);
        for (int i = 0; i < 3; ++i)
            fprintf(stderr, "\n[bbhip descent] %-10s levels/insert=%.3f cycles/level=%.0f", cls[i],
                    n > 0 ? (double)t->h.phase[11 + i] / n : 0.0,
                    t->h.phase[11 + i] ? (double)t->h.phase[8 + i] / (double)t->h.phase[11 + i] : 0.0);
        fprintf(stderr, "\n[bbhip split phases, cycles/split]");
        const double ns = (double)t->h.stats[4];
        for (int i = 0; i < 8; ++i) fprintf(stderr, " s%d=%.0f", i, ns > 0 ? (double)t->h.sphase[i] / ns : 0.0);
        fprintf(stderr, "\n");
    }
    return BBH_OK;
}
