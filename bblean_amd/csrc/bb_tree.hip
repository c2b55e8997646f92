// bb_tree.hip -- HBM-resident BitBIRCH tree engine for gfx950 (MI355X, CDNA4).
//
// Replaces the per-fingerprint Python loop of the reference (bblean/bitbirch.py:769-787
// and :848-866) and everything it calls: _BFNode.insert_bf_subcluster (:305-357),
// _BFSubcluster.merge_subcluster / update (:488-526), the merge criteria (_merges.py),
// centroid_from_sum (_py_similarity.py:12-42), jt_isim_from_sum (similarity.cpp:273-301),
// _jt_sim_arr_vec_packed + np.argmax (similarity.cpp:374, bitbirch.py:317-320) and
// _split_node + jt_most_dissimilar_packed (bitbirch.py:162-211, similarity.cpp:413-471).
//
// Data layout in HBM (all pools are flat arrays indexed by id, grown by the host):
//   node pool : per node (bf+1) packed centroid rows of RB bytes (RB = row bytes padded to
//               16), plus per row: BitFeature id, child node id, centroid popcount; per
//               node: row count, leaf flag, prev/next leaf (the leaf chain).
//   BF pool   : per BitFeature n_samples, the exact moments sum(ls) and sum(ls^2) (u64)
//               and a slot word (tier | index) into one of three cluster-feature pools:
//               cf8 / cf16 / cf32 = linear sums at 1/2/4 bytes per feature (the reference
//               keeps the minimum dtype for n_samples, utils.py:25; leaf BitFeatures move
//               up a tier when n_samples crosses 255 / 65535, tracking BitFeatures of
//               internal nodes always live in cf32).
//
// Execution model: ONE 256-thread workgroup walks one tree and inserts a whole batch
// with the reference's sequential semantics.  Inside an insert the work is data
// parallel: 16 lanes x 16 B cover a 256-byte centroid row (16 rows per pass, DPP row
// reduction of the AND-popcounts), keys are combined through LDS, the cluster-feature
// update is one thread per 8 features.  All decisions use exact integers; the only
// floating point is the reference's own f64 formulae, evaluated in the same order.
#include "bb_common.h"

#include <algorithm>
#include <cmath>

using namespace bbd;

namespace {

constexpr int TB = 256;  // threads per tree workgroup
constexpr int TW = TB / 64;
constexpr uint32_t NONE = 0xFFFFFFFFu;
constexpr int MAXD = 64;  // deepest tree handled
constexpr int MAX_BF = 1023;

enum StopReason : int32_t {
    STOP_DONE = 0,
    STOP_NODES = 1,
    STOP_SUBS = 2,
    STOP_CF8 = 3,
    STOP_CF16 = 4,
    STOP_CF32 = 5,
    STOP_DEPTH = 6,
    STOP_RANGE = 7,  // n_samples would exceed 2^32-1
};

enum Ctr : int { C_NODES = 0, C_SUBS, C_N8, C_N16, C_N32, C_ROOT, C_FIRST_LEAF, C_DEPTH, C_COUNT };

struct TreeDev {
    // configuration
    int32_t bf, F, nbytes, RB;
    int32_t crit, tol_len;
    double thr, tolerance;
    const double* tol_table;
    // node pool
    uint8_t* node_cent;
    uint32_t* node_sub;
    uint32_t* node_child;
    uint32_t* node_card;
    uint32_t* node_len;
    uint32_t* node_prev;
    uint32_t* node_next;
    uint32_t* node_leaf;
    uint8_t* scratch_cent;  // (bf+1) x RB, staging for row moves in a split
    // BitFeature pool
    uint32_t* sub_n;
    unsigned long long* sub_s1;
    unsigned long long* sub_s2;
    uint32_t* sub_slot;
    uint8_t* cf8;
    uint16_t* cf16;
    uint32_t* cf32;
    uint32_t cap_nodes, cap_subs, cap8, cap16, cap32;
    // state
    uint32_t ctr[C_COUNT];
    unsigned long long stats[8];
    // per launch result
    long long processed;
    int32_t stop_reason;
};

// -------------------------------------------------------------------------------------
// device side
// -------------------------------------------------------------------------------------
struct Smem {
    uint4* x;        // packed centroid of the element being inserted, RB bytes
    uint4* vec;      // comparison vector during a split
    uint4* cA;       // centroid of tracking BitFeature A / B after a split
    uint4* cB;
    unsigned long long* keys;  // 2 x (bf+1)
    uint32_t* child;           // 2 x (bf+1)
    uint32_t* sub;             // 2 x (bf+1)
    uint32_t* i1;              // (bf+1) each
    uint32_t* u1;
    uint32_t* i2;
    uint32_t* u2;
    uint32_t* dst;    // split: flag << 31 | destination row
    uint32_t* mslot;  // split: slot word of each row's BitFeature
    uint32_t* mn;     // split: n_samples of each row's BitFeature
    uint32_t* msub;
    uint32_t* mchild;
    uint32_t* mcard;
    unsigned long long* red;  // 2 x TW x 4
    uint32_t* path_node;      // MAXD each
    uint32_t* path_row;
    uint32_t* path_sub;
    uint32_t* path_len;
    uint32_t* ctr;  // C_COUNT
    unsigned long long* stats;
    uint32_t* bc;  // broadcast scratch, 16
};

__host__ __device__ inline size_t smem_layout(int bf, int RB, Smem* s, unsigned char* base) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
        size_t o = off;
        off += (bytes + 15) / 16 * 16;
        return o;
    };
    const size_t m = (size_t)bf + 1;
    size_t o_x = take(RB), o_vec = take(RB), o_cA = take(RB), o_cB = take(RB);
    size_t o_keys = take(2 * m * 8), o_child = take(2 * m * 4), o_sub = take(2 * m * 4);
    size_t o_i1 = take(m * 4), o_u1 = take(m * 4), o_i2 = take(m * 4), o_u2 = take(m * 4);
    size_t o_dst = take(m * 4), o_mslot = take(m * 4), o_mn = take(m * 4), o_msub = take(m * 4);
    size_t o_mchild = take(m * 4), o_mcard = take(m * 4);
    size_t o_red = take(2 * TW * 4 * 8);
    size_t o_pn = take(MAXD * 4), o_pr = take(MAXD * 4), o_ps = take(MAXD * 4), o_pl = take(MAXD * 4);
    size_t o_ctr = take(C_COUNT * 4), o_stats = take(8 * 8), o_bc = take(16 * 4);
    if (s) {
        s->x = (uint4*)(base + o_x);
        s->vec = (uint4*)(base + o_vec);
        s->cA = (uint4*)(base + o_cA);
        s->cB = (uint4*)(base + o_cB);
        s->keys = (unsigned long long*)(base + o_keys);
        s->child = (uint32_t*)(base + o_child);
        s->sub = (uint32_t*)(base + o_sub);
        s->i1 = (uint32_t*)(base + o_i1);
        s->u1 = (uint32_t*)(base + o_u1);
        s->i2 = (uint32_t*)(base + o_i2);
        s->u2 = (uint32_t*)(base + o_u2);
        s->dst = (uint32_t*)(base + o_dst);
        s->mslot = (uint32_t*)(base + o_mslot);
        s->mn = (uint32_t*)(base + o_mn);
        s->msub = (uint32_t*)(base + o_msub);
        s->mchild = (uint32_t*)(base + o_mchild);
        s->mcard = (uint32_t*)(base + o_mcard);
        s->red = (unsigned long long*)(base + o_red);
        s->path_node = (uint32_t*)(base + o_pn);
        s->path_row = (uint32_t*)(base + o_pr);
        s->path_sub = (uint32_t*)(base + o_ps);
        s->path_len = (uint32_t*)(base + o_pl);
        s->ctr = (uint32_t*)(base + o_ctr);
        s->stats = (unsigned long long*)(base + o_stats);
        s->bc = (uint32_t*)(base + o_bc);
    }
    return off;
}

struct Ctx {
    TreeDev t;  // by-value copy of pointers and config (lives in SGPRs / scalar loads)
    Smem s;
    int RBc;           // 16-byte chunks per row
    size_t node_rows;  // bf + 1
    int red_slot;
    int cmp_par;
    // current element
    const uint8_t* bufs;  // BitFeature buffer table (NULL = fingerprint mode)
    int width;
    long long elem;
    uint32_t nS;
    unsigned long long s1S, s2S;
    uint32_t pcx;
};

#if defined(__HIPCC__)

// ---- block reductions (sum of up to 4 u64 values to every thread) ---------------------
__device__ __forceinline__ void block_sum4(Ctx& c, unsigned long long& a, unsigned long long& b,
                                           unsigned long long& d, unsigned long long& e) {
    a = wave_sum_u64(a);
    b = wave_sum_u64(b);
    d = wave_sum_u64(d);
    e = wave_sum_u64(e);
    unsigned long long* buf = c.s.red + (size_t)c.red_slot * TW * 4;
    c.red_slot ^= 1;
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        buf[w * 4 + 0] = a;
        buf[w * 4 + 1] = b;
        buf[w * 4 + 2] = d;
        buf[w * 4 + 3] = e;
    }
    __syncthreads();
    a = b = d = e = 0;
#pragma unroll
    for (int i = 0; i < TW; ++i) {
        a += buf[i * 4 + 0];
        b += buf[i * 4 + 1];
        d += buf[i * 4 + 2];
        e += buf[i * 4 + 3];
    }
}

__device__ __forceinline__ void block_sum2(Ctx& c, unsigned long long& a, unsigned long long& b) {
    unsigned long long z0 = 0, z1 = 0;
    block_sum4(c, a, b, z0, z1);
}

// ---- cluster-feature access: 8 consecutive features of one BitFeature ------------------
__device__ __forceinline__ void cf_load8(const Ctx& c, uint32_t slotw, int b, uint32_t v[8]) {
    const uint32_t tier = slotw >> 30;
    const size_t idx = (size_t)(slotw & 0x3FFFFFFFu) * (size_t)c.t.F + (size_t)b * 8;
    if (tier == 0) {
        uint2 q = *reinterpret_cast<const uint2*>(c.t.cf8 + idx);
        v[0] = q.x & 0xFF; v[1] = (q.x >> 8) & 0xFF; v[2] = (q.x >> 16) & 0xFF; v[3] = q.x >> 24;
        v[4] = q.y & 0xFF; v[5] = (q.y >> 8) & 0xFF; v[6] = (q.y >> 16) & 0xFF; v[7] = q.y >> 24;
    } else if (tier == 1) {
        uint4 q = *reinterpret_cast<const uint4*>(c.t.cf16 + idx);
        v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
        v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
    } else {
        const uint4* p = reinterpret_cast<const uint4*>(c.t.cf32 + idx);
        uint4 q0 = p[0], q1 = p[1];
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w;
        v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    }
}

__device__ __forceinline__ void cf_store8(const Ctx& c, uint32_t slotw, int b, const uint32_t v[8]) {
    const uint32_t tier = slotw >> 30;
    const size_t idx = (size_t)(slotw & 0x3FFFFFFFu) * (size_t)c.t.F + (size_t)b * 8;
    if (tier == 0) {
        uint2 q;
        q.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
        q.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
        *reinterpret_cast<uint2*>(c.t.cf8 + idx) = q;
    } else if (tier == 1) {
        uint4 q;
        q.x = v[0] | (v[1] << 16); q.y = v[2] | (v[3] << 16);
        q.z = v[4] | (v[5] << 16); q.w = v[6] | (v[7] << 16);
        *reinterpret_cast<uint4*>(c.t.cf16 + idx) = q;
    } else {
        uint4* p = reinterpret_cast<uint4*>(c.t.cf32 + idx);
        p[0] = make_uint4(v[0], v[1], v[2], v[3]);
        p[1] = make_uint4(v[4], v[5], v[6], v[7]);
    }
}

// linear-sum values of the element being inserted for features b*8 .. b*8+7
__device__ __forceinline__ void elem_cols(const Ctx& c, int b, uint32_t v[8]) {
    if (c.bufs == nullptr) {  // fingerprint: bits of the packed row, MSB first
        const uint32_t xb = reinterpret_cast<const uint8_t*>(c.s.x)[b];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (xb >> (7 - k)) & 1u;
        return;
    }
    const size_t base = (size_t)c.elem * ((size_t)c.t.F + 1) + (size_t)b * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        switch (c.width) {
            case 1: v[k] = c.bufs[base + k]; break;
            case 2: v[k] = reinterpret_cast<const uint16_t*>(c.bufs)[base + k]; break;
            case 4: v[k] = reinterpret_cast<const uint32_t*>(c.bufs)[base + k]; break;
            default: v[k] = (uint32_t) reinterpret_cast<const unsigned long long*>(c.bufs)[base + k]; break;
        }
    }
}

// majority-vote byte for 8 features (centroid_from_sum, _py_similarity.py:36-41)
__device__ __forceinline__ uint32_t centroid_byte(const uint32_t v[8], unsigned long long n) {
    uint32_t byte = 0;
    if (n <= 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) byte |= ((v[k] & 0xFFu) != 0 ? 1u : 0u) << (7 - k);
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) byte |= (2ull * v[k] >= n ? 1u : 0u) << (7 - k);
    }
    return byte;
}

__device__ __forceinline__ uint32_t tier_for(unsigned long long n) { return n <= 255 ? 0u : (n <= 65535 ? 1u : 2u); }

// popcount of an RB-byte vector in LDS (every wave computes it redundantly)
__device__ __forceinline__ uint32_t lds_vec_popcount(const Ctx& c, const uint4* v) {
    uint32_t p = 0;
    for (int ch = threadIdx.x & 63; ch < c.RBc; ch += 64) p += popc4(v[ch]);
    return wave_sum_u32(p);
}

// ---- similarity of every row of a node against a vector in LDS -------------------------
// mode 0: keys for first-argmax; mode 1: keys for first-argmin.  Ends with a barrier.
__device__ void node_compare(Ctx& c, uint32_t nd, uint32_t len, const uint4* vec, uint32_t vec_pc, int mode,
                             uint32_t* s_i, uint32_t* s_u, bool load_meta) {
    const int tid = threadIdx.x, l = tid & 15, g = tid >> 4;
    const size_t rows = c.node_rows;
    const uint8_t* base = c.t.node_cent + (size_t)nd * rows * (size_t)c.t.RB;
    c.cmp_par ^= 1;
    unsigned long long* keys = c.s.keys + (size_t)c.cmp_par * rows;
    uint32_t* s_child = c.s.child + (size_t)c.cmp_par * rows;
    uint32_t* s_sub = c.s.sub + (size_t)c.cmp_par * rows;
    const size_t meta = (size_t)nd * rows;
    if (c.RBc == 16) {
        const uint4 xv = vec[l];
        for (uint32_t r0 = 0; r0 < len; r0 += 64) {
            uint4 d[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t r = r0 + p * 16 + g;
                d[p] = make_uint4(0, 0, 0, 0);
                if (r < len) d[p] = reinterpret_cast<const uint4*>(base + (size_t)r * 256)[l];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t r = r0 + p * 16 + g;
                const uint32_t inter = row16_sum(popc4(and4(d[p], xv)));
                if (r < len && l == 0) {
                    const uint32_t un = c.t.node_card[meta + r] + vec_pc - inter;
                    const unsigned long long bits = (unsigned long long)__double_as_longlong(jt_from_counts(inter, un));
                    keys[r] = (bits & ~0xFFFull) | (unsigned long long)(mode == 0 ? (0xFFFu - r) : r);
                    if (s_i) { s_i[r] = inter; s_u[r] = un; }
                    if (load_meta) { s_child[r] = c.t.node_child[meta + r]; s_sub[r] = c.t.node_sub[meta + r]; }
                }
            }
        }
    } else {
        for (uint32_t r0 = 0; r0 < len; r0 += TB / 16) {
            const uint32_t r = r0 + g;
            uint32_t inter = 0;
            if (r < len) {
                const uint4* row = reinterpret_cast<const uint4*>(base + (size_t)r * c.t.RB);
                for (int ch = l; ch < c.RBc; ch += 16) inter += popc4(and4(row[ch], vec[ch]));
            }
            inter = row16_sum(inter);
            if (r < len && l == 0) {
                const uint32_t un = c.t.node_card[meta + r] + vec_pc - inter;
                const unsigned long long bits = (unsigned long long)__double_as_longlong(jt_from_counts(inter, un));
                keys[r] = (bits & ~0xFFFull) | (unsigned long long)(mode == 0 ? (0xFFFu - r) : r);
                if (s_i) { s_i[r] = inter; s_u[r] = un; }
                if (load_meta) { s_child[r] = c.t.node_child[meta + r]; s_sub[r] = c.t.node_sub[meta + r]; }
            }
        }
    }
    __syncthreads();
}

// first index of the max (mode 0) / min (mode 1) similarity among the keys just written
__device__ __forceinline__ uint32_t pick_best(const Ctx& c, uint32_t len, int mode) {
    const unsigned long long* keys = c.s.keys + (size_t)c.cmp_par * c.node_rows;
    unsigned long long k = mode == 0 ? 0ull : ~0ull;
    for (uint32_t r = threadIdx.x & 63; r < len; r += 64) {
        const unsigned long long v = keys[r];
        k = mode == 0 ? (v > k ? v : k) : (v < k ? v : k);
    }
    k = mode == 0 ? wave_max_u64(k) : wave_min_u64(k);
    const uint32_t low = (uint32_t)(k & 0xFFFull);
    return mode == 0 ? (0xFFFu - low) : low;
}

// write one node row: centroid (from LDS), popcount, BitFeature id, child node id
__device__ __forceinline__ void node_put_row(const Ctx& c, uint32_t nd, uint32_t row, const uint4* cent, uint32_t card,
                                             uint32_t sub, uint32_t child) {
    uint4* dstp = reinterpret_cast<uint4*>(c.t.node_cent + ((size_t)nd * c.node_rows + row) * (size_t)c.t.RB);
    for (int ch = threadIdx.x; ch < c.RBc; ch += TB) dstp[ch] = cent[ch];
    if (threadIdx.x == 0) {
        const size_t m = (size_t)nd * c.node_rows + row;
        c.t.node_card[m] = card;
        c.t.node_sub[m] = sub;
        c.t.node_child[m] = child;
    }
}

// ---- radius complement terms (similarity.py:192-202) on CF(slot) [+ element] -----------
// returns sum(c) and sum(2*v*c + c) where c = majority bit of v for n >= 2
__device__ void radius_terms(Ctx& c, uint32_t slotw, bool add_elem, unsigned long long n,
                             unsigned long long& sc, unsigned long long& sq) {
    unsigned long long a = 0, q = 0;
    const int nb = c.t.nbytes;
    for (int b = threadIdx.x; b < nb; b += TB) {
        uint32_t v[8], e[8];
        cf_load8(c, slotw, b, v);
        if (add_elem) {
            elem_cols(c, b, e);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] += e[k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long vv = v[k];
            const unsigned long long bit = n <= 1 ? (unsigned long long)((vv & 0xFF) != 0) : (2ull * vv >= n ? 1ull : 0ull);
            // reference adds the centroid VALUE (cast) for n<=1; it is 0/1 in every reachable case
            a += bit;
            q += 2ull * vv * bit + bit;
        }
    }
    block_sum2(c, a, q);
    sc = a;
    sq = q;
}

__device__ __forceinline__ double radius_compl(unsigned long long s1, unsigned long long s2, unsigned long long sc,
                                               unsigned long long sq, unsigned long long n) {
    const double jt = isim_from_moments(s1, s2, n);
    const double jt1 = isim_from_moments(s1 + sc, s2 + sq, n + 1);
    return (jt1 * (double)(n + 1) - jt * (double)(n - 1)) / 2;
}

__device__ __forceinline__ double tol_lookup(const Ctx& c, unsigned long long old_n) {
    if (c.t.tol_table == nullptr || old_n >= (unsigned long long)c.t.tol_len) return 0.0;
    return c.t.tol_table[old_n];
}

// merge_accept_fn(threshold, new_ls, new_n, old_ls, nom_ls, old_n, nom_n) of _merges.py,
// on exact moments.  Uniform across the block.
__device__ bool merge_accept(Ctx& c, uint32_t slotT, unsigned long long nT, unsigned long long s1T,
                             unsigned long long s2T, unsigned long long new_n, unsigned long long s1n,
                             unsigned long long s2n) {
    const double thr = c.t.thr;
    switch (c.t.crit) {
        case BBH_CRIT_DIAMETER:
            return isim_from_moments(s1n, s2n, new_n) >= thr;
        case BBH_CRIT_TOL_DIAMETER: {
            const double new_dc = isim_from_moments(s1n, s2n, new_n);
            if (new_dc < thr) return false;
            if (nT == 1) return true;
            const double old_dc = isim_from_moments(s1T, s2T, nT);
            return new_dc >= old_dc - tol_lookup(c, nT);
        }
        case BBH_CRIT_TOL_LEGACY: {
            const double new_dc = isim_from_moments(s1n, s2n, new_n);
            if (new_dc < thr) return false;
            if (nT == 1 || c.nS != 1) return true;
            const double old_dc = isim_from_moments(s1T, s2T, nT);
            return (new_dc * (double)new_n - old_dc * (double)(nT - 1)) / 2 >= old_dc - c.t.tolerance;
        }
        case BBH_CRIT_RADIUS: {
            unsigned long long sc, sq;
            radius_terms(c, slotT, true, new_n, sc, sq);
            return radius_compl(s1n, s2n, sc, sq, new_n) >= thr;
        }
        case BBH_CRIT_TOL_RADIUS: {
            unsigned long long sc, sq;
            radius_terms(c, slotT, true, new_n, sc, sq);
            const double new_rc = radius_compl(s1n, s2n, sc, sq, new_n);
            if (new_rc < thr) return false;  // uniform: every thread took part in the reduction above
            if (nT == 1) return true;
            radius_terms(c, slotT, false, nT, sc, sq);
            const double old_rc = radius_compl(s1T, s2T, sc, sq, nT);
            return new_rc >= old_rc - tol_lookup(c, nT);
        }
        default:
            return false;  // never-merge
    }
}

// ---- _split_node (bitbirch.py:162-211) -------------------------------------------------
// Splits node `nd` (len = bf+1 rows).  Leaves the two tracking BitFeatures' centroids in
// s.cA / s.cB and returns ids through s.bc: [0]=node1 [1]=A [2]=B [3]=cardA [4]=cardB.
__device__ void split_node(Ctx& c, uint32_t nd) {
    const int tid = threadIdx.x;
    const uint32_t m = c.t.node_len[nd];
    const size_t rows = c.node_rows;
    const size_t meta = (size_t)nd * rows;
    const int nb = c.t.nbytes;
    uint8_t* cent = c.t.node_cent + meta * (size_t)c.t.RB;
    // 0. row metadata to LDS; zero the comparison vector padding
    for (uint32_t r = tid; r < m; r += TB) {
        const uint32_t sb = c.t.node_sub[meta + r];
        c.s.msub[r] = sb;
        c.s.mchild[r] = c.t.node_child[meta + r];
        c.s.mcard[r] = c.t.node_card[meta + r];
        c.s.mslot[r] = c.t.sub_slot[sb];
        c.s.mn[r] = c.t.sub_n[sb];
    }
    for (int ch = tid; ch < c.RBc; ch += TB) c.s.vec[ch] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    // 1. majority centroid of the node's centroids (column sums of the unpacked rows)
    for (int b = tid; b < nb; b += TB) {
        uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t r = 0; r < m; ++r) {
            const uint32_t v = cent[(size_t)r * c.t.RB + b];
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += (v >> (7 - k)) & 1u;
        }
        reinterpret_cast<uint8_t*>(c.s.vec)[b] = (uint8_t)centroid_byte(acc, m);
    }
    __syncthreads();
    uint32_t pc = lds_vec_popcount(c, c.s.vec);
    // 2. fp1 = first argmin of similarity to that centroid
    node_compare(c, nd, m, c.s.vec, pc, 1, nullptr, nullptr, false);
    const uint32_t f1 = pick_best(c, m, 1);
    // 3. similarities to fp1; fp2 = first argmin
    for (int ch = tid; ch < c.RBc; ch += TB) c.s.vec[ch] = reinterpret_cast<const uint4*>(cent + (size_t)f1 * c.t.RB)[ch];
    __syncthreads();
    node_compare(c, nd, m, c.s.vec, c.s.mcard[f1], 1, c.s.i1, c.s.u1, false);
    const uint32_t f2 = pick_best(c, m, 1);
    // 4. similarities to fp2
    for (int ch = tid; ch < c.RBc; ch += TB) c.s.vec[ch] = reinterpret_cast<const uint4*>(cent + (size_t)f2 * c.t.RB)[ch];
    __syncthreads();
    node_compare(c, nd, m, c.s.vec, c.s.mcard[f2], 1, c.s.i2, c.s.u2, false);
    // 5. node1_closer = sims_fp1 > sims_fp2 (exact cross-multiplication), node1_closer[fp1] = True
    for (uint32_t r = tid; r < m; r += TB) {
        const uint32_t a1 = c.s.i1[r], b1 = c.s.u1[r] < 1u ? 1u : c.s.u1[r];
        const uint32_t a2 = c.s.i2[r], b2 = c.s.u2[r] < 1u ? 1u : c.s.u2[r];
        const bool to1 = ((unsigned long long)a1 * b2 > (unsigned long long)a2 * b1) || r == f1;
        c.s.dst[r] = to1 ? 0x80000000u : 0u;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t n1 = 0, n2 = 0;
        unsigned long long nA = 0, nB = 0;
        for (uint32_t r = 0; r < m; ++r) {
            if (c.s.dst[r] & 0x80000000u) {
                c.s.dst[r] = 0x80000000u | n1++;
                nA += c.s.mn[r];
            } else {
                c.s.dst[r] = n2++;
                nB += c.s.mn[r];
            }
        }
        // 6. ids for the new node and the two tracking BitFeatures (always cf32)
        const uint32_t node1 = c.s.ctr[C_NODES]++;
        const uint32_t A = c.s.ctr[C_SUBS], B = A + 1;
        c.s.ctr[C_SUBS] += 2;
        const uint32_t slotA = c.s.ctr[C_N32], slotB = slotA + 1;
        c.s.ctr[C_N32] += 2;
        c.s.bc[0] = node1;
        c.s.bc[1] = A;
        c.s.bc[2] = B;
        c.s.bc[5] = n1;
        c.s.bc[6] = n2;
        c.s.bc[7] = (uint32_t)nA;
        c.s.bc[8] = (uint32_t)nB;
        c.s.bc[9] = slotA;
        c.s.bc[10] = slotB;
        c.t.sub_n[A] = (uint32_t)nA;
        c.t.sub_n[B] = (uint32_t)nB;
        c.t.sub_slot[A] = (2u << 30) | slotA;
        c.t.sub_slot[B] = (2u << 30) | slotB;
        c.t.sub_s1[A] = c.t.sub_s2[A] = c.t.sub_s1[B] = c.t.sub_s2[B] = 0;
        // 9. leaf chain: node1 goes immediately before nd (bitbirch.py:182-188)
        const uint32_t leaf = c.t.node_leaf[nd];
        c.t.node_leaf[node1] = leaf;
        c.t.node_len[node1] = n1;
        c.t.node_len[nd] = n2;
        if (leaf) {
            const uint32_t prev = c.t.node_prev[nd];
            c.t.node_prev[node1] = prev;
            if (prev != NONE) c.t.node_next[prev] = node1; else c.s.ctr[C_FIRST_LEAF] = node1;
            c.t.node_next[node1] = nd;
            c.t.node_prev[nd] = node1;
        } else {
            c.t.node_prev[node1] = NONE;
            c.t.node_next[node1] = NONE;
        }
        c.s.stats[4]++;
        c.s.stats[5]++;
    }
    // 7a. stage all centroid rows (the kept ones are compacted in place afterwards)
    for (size_t i = tid; i < (size_t)m * c.RBc; i += TB)
        reinterpret_cast<uint4*>(c.t.scratch_cent)[i] = reinterpret_cast<const uint4*>(cent)[i];
    __syncthreads();
    const uint32_t node1 = c.s.bc[0];
    const unsigned long long nA = c.s.bc[7], nB = c.s.bc[8];
    const uint32_t slotA = (2u << 30) | c.s.bc[9], slotB = (2u << 30) | c.s.bc[10];
    // 7b. distribute rows in their original order
    {
        uint8_t* cent1 = c.t.node_cent + (size_t)node1 * rows * (size_t)c.t.RB;
        for (size_t i = tid; i < (size_t)m * c.RBc; i += TB) {
            const uint32_t r = (uint32_t)(i / c.RBc), ch = (uint32_t)(i % c.RBc);
            const uint32_t d = c.s.dst[r];
            uint8_t* dstbase = (d & 0x80000000u) ? cent1 : cent;
            reinterpret_cast<uint4*>(dstbase + (size_t)(d & 0x7FFFFFFFu) * c.t.RB)[ch] =
                reinterpret_cast<const uint4*>(c.t.scratch_cent)[i];
        }
        for (uint32_t r = tid; r < m; r += TB) {
            const uint32_t d = c.s.dst[r];
            const size_t mm = ((d & 0x80000000u) ? (size_t)node1 * rows : meta) + (d & 0x7FFFFFFFu);
            c.t.node_sub[mm] = c.s.msub[r];
            c.t.node_child[mm] = c.s.mchild[r];
            c.t.node_card[mm] = c.s.mcard[r];
        }
    }
    // 8. tracking BitFeatures: CF = sum of member CFs; centroid from the final CF
    for (int ch = tid; ch < c.RBc; ch += TB) {
        c.s.cA[ch] = make_uint4(0, 0, 0, 0);
        c.s.cB[ch] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    unsigned long long cardA = 0, cardB = 0;
    for (int b = tid; b < nb; b += TB) {
        uint32_t accA[8] = {0, 0, 0, 0, 0, 0, 0, 0}, accB[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t r = 0; r < m; ++r) {
            uint32_t v[8];
            cf_load8(c, c.s.mslot[r], b, v);
            if (c.s.dst[r] & 0x80000000u) {
#pragma unroll
                for (int k = 0; k < 8; ++k) accA[k] += v[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) accB[k] += v[k];
            }
        }
        cf_store8(c, slotA, b, accA);
        cf_store8(c, slotB, b, accB);
        const uint32_t ba = centroid_byte(accA, nA), bb_ = centroid_byte(accB, nB);
        reinterpret_cast<uint8_t*>(c.s.cA)[b] = (uint8_t)ba;
        reinterpret_cast<uint8_t*>(c.s.cB)[b] = (uint8_t)bb_;
        cardA += __popc(ba);
        cardB += __popc(bb_);
    }
    block_sum2(c, cardA, cardB);  // barrier inside: cA / cB complete for every thread
    if (tid == 0) {
        c.s.bc[3] = (uint32_t)cardA;
        c.s.bc[4] = (uint32_t)cardB;
    }
    __syncthreads();
}

// one tree per workgroup
__global__ __launch_bounds__(TB) void k_tree_insert(TreeDev* trees, const uint8_t* rows, long long row_stride,
                                                    const uint8_t* bufs, int width, long long n_elems,
                                                    uint32_t* out_leaf) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    TreeDev* T = trees + blockIdx.x;
    Ctx c;
    c.t = *T;
    smem_layout(c.t.bf, c.t.RB, &c.s, smem_raw);
    c.RBc = c.t.RB / 16;
    c.node_rows = (size_t)c.t.bf + 1;
    c.red_slot = 0;
    c.cmp_par = 0;
    c.bufs = bufs;
    c.width = width;
    const int tid = threadIdx.x;
    const int nb = c.t.nbytes;
    const uint32_t bf = (uint32_t)c.t.bf;
    if (tid < C_COUNT) c.s.ctr[tid] = c.t.ctr[tid];
    if (tid < 8) c.s.stats[tid] = c.t.stats[tid];
    __syncthreads();

    long long e = 0;
    int stop = STOP_DONE;
    for (; e < n_elems; ++e) {
        __syncthreads();
        // ---- capacity for the worst case of one insertion -------------------------------
        {
            const uint32_t depth = c.s.ctr[C_DEPTH];
            const uint32_t need_nodes = depth + 2, need_subs = 2 * (depth + 1) + 1;
            if (depth + 2 >= (uint32_t)MAXD) { stop = STOP_DEPTH; break; }
            if (c.s.ctr[C_NODES] + need_nodes > c.t.cap_nodes) { stop = STOP_NODES; break; }
            if (c.s.ctr[C_SUBS] + need_subs > c.t.cap_subs) { stop = STOP_SUBS; break; }
            if (c.s.ctr[C_N8] + 1 > c.t.cap8) { stop = STOP_CF8; break; }
            if (c.s.ctr[C_N16] + 1 > c.t.cap16) { stop = STOP_CF16; break; }
            if (c.s.ctr[C_N32] + need_subs > c.t.cap32) { stop = STOP_CF32; break; }
        }
        c.elem = e;
        // ---- element: packed centroid into LDS, n, moments ------------------------------
        for (int ch = tid; ch < c.RBc; ch += TB) c.s.x[ch] = make_uint4(0, 0, 0, 0);
        __syncthreads();
        if (bufs == nullptr) {
            const uint8_t* row = rows + e * row_stride;
            if ((((uintptr_t)row) & 15) == 0 && (nb & 15) == 0) {
                for (int ch = tid; ch < c.RBc; ch += TB) c.s.x[ch] = reinterpret_cast<const uint4*>(row)[ch];
            } else {
                for (int b = tid; b < nb; b += TB) reinterpret_cast<uint8_t*>(c.s.x)[b] = row[b];
            }
            c.nS = 1;
            __syncthreads();
            c.pcx = lds_vec_popcount(c, c.s.x);
            c.s1S = c.pcx;
            c.s2S = c.pcx;
        } else {
            unsigned long long nraw;
            const size_t ncol = (size_t)e * ((size_t)c.t.F + 1) + (size_t)c.t.F;
            switch (width) {
                case 1: nraw = bufs[ncol]; break;
                case 2: nraw = reinterpret_cast<const uint16_t*>(bufs)[ncol]; break;
                case 4: nraw = reinterpret_cast<const uint32_t*>(bufs)[ncol]; break;
                default: nraw = reinterpret_cast<const unsigned long long*>(bufs)[ncol]; break;
            }
            if (nraw > 0xFFFFFFFFull) { stop = STOP_RANGE; break; }
            c.nS = (uint32_t)nraw;
            unsigned long long a = 0, q = 0;
            for (int b = tid; b < nb; b += TB) {
                uint32_t v[8];
                elem_cols(c, b, v);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    a += v[k];
                    q += (unsigned long long)v[k] * v[k];
                }
                reinterpret_cast<uint8_t*>(c.s.x)[b] = (uint8_t)centroid_byte(v, c.nS);
            }
            block_sum2(c, a, q);  // barrier: s.x complete
            c.s1S = a;
            c.s2S = q;
            c.pcx = lds_vec_popcount(c, c.s.x);
        }

        uint32_t out_id;
        bool overflow = false;
        int D = 0;  // leaf level index
        const uint32_t root = c.s.ctr[C_ROOT];
        uint32_t len = c.t.node_len[root];
        if (len == 0) {
            // ---- very first element of an empty tree: becomes row 0 of the root leaf -----
            const uint32_t tier = tier_for(c.nS);
            const uint32_t s = c.s.ctr[C_SUBS];
            const uint32_t slot = c.s.ctr[tier == 0 ? C_N8 : (tier == 1 ? C_N16 : C_N32)];
            const uint32_t slotw = (tier << 30) | slot;
            __syncthreads();
            if (tid == 0) {
                c.s.ctr[C_SUBS]++;
                c.s.ctr[tier == 0 ? C_N8 : (tier == 1 ? C_N16 : C_N32)]++;
                c.t.sub_n[s] = c.nS;
                c.t.sub_s1[s] = c.s1S;
                c.t.sub_s2[s] = c.s2S;
                c.t.sub_slot[s] = slotw;
                c.t.node_len[root] = 1;
                c.s.stats[3]++;
            }
            for (int b = tid; b < nb; b += TB) {
                uint32_t v[8];
                elem_cols(c, b, v);
                cf_store8(c, slotw, b, v);
            }
            node_put_row(c, root, 0, c.s.x, c.pcx, s, NONE);
            out_id = s;
        } else {
            // ---- greedy descent (bitbirch.py:305-357) ---------------------------------
            uint32_t nd = root, j, child, sub;
            int depth = 0;
            while (true) {
                node_compare(c, nd, len, c.s.x, c.pcx, 0, nullptr, nullptr, true);
                j = pick_best(c, len, 0);
                child = c.s.child[(size_t)c.cmp_par * c.node_rows + j];
                sub = c.s.sub[(size_t)c.cmp_par * c.node_rows + j];
                if (tid == 0) {
                    c.s.path_node[depth] = nd;
                    c.s.path_row[depth] = j;
                    c.s.path_sub[depth] = sub;
                    c.s.path_len[depth] = len;
                    c.s.stats[0]++;
                    c.s.stats[1] += len;
                }
                if (child == NONE) break;
                if (depth + 2 >= MAXD || child >= c.t.cap_nodes) {  // corrupt or too deep: never spin
                    stop = STOP_DEPTH;
                    break;
                }
                nd = child;
                len = c.t.node_len[nd];
                depth++;
            }
            if (stop != STOP_DONE) break;
            D = depth;
            if ((unsigned long long)(D + 1) > c.s.stats[6] && tid == 0) c.s.stats[6] = (unsigned long long)(D + 1);
            const uint32_t leafnode = nd, jl = j, Tsub = sub, leaflen = len;
            // ---- leaf merge test (merge_subcluster, bitbirch.py:507-526) ----------------
            const uint32_t slotT = c.t.sub_slot[Tsub];
            const unsigned long long nT = c.t.sub_n[Tsub];
            const unsigned long long s1T = c.t.sub_s1[Tsub], s2T = c.t.sub_s2[Tsub];
            unsigned long long dot = 0, zz = 0;
            for (int b = tid; b < nb; b += TB) {
                uint32_t v[8], x8[8];
                cf_load8(c, slotT, b, v);
                elem_cols(c, b, x8);
#pragma unroll
                for (int k = 0; k < 8; ++k) dot += (unsigned long long)v[k] * x8[k];
            }
            block_sum2(c, dot, zz);
            const unsigned long long new_n = nT + c.nS;
            if (new_n > 0xFFFFFFFFull) { stop = STOP_RANGE; break; }
            const unsigned long long s1n = s1T + c.s1S;
            const unsigned long long s2n = s2T + 2ull * dot + c.s2S;
            const bool accept = merge_accept(c, slotT, nT, s1T, s2T, new_n, s1n, s2n);
            if (accept) {
                // replace_n_samples_and_linear_sum (bitbirch.py:476-484)
                const uint32_t old_tier = slotT >> 30, new_tier = tier_for(new_n) > old_tier ? tier_for(new_n) : old_tier;
                uint32_t slotN = slotT;
                if (new_tier != old_tier) {
                    const int which = new_tier == 1 ? C_N16 : C_N32;
                    slotN = (new_tier << 30) | c.s.ctr[which];
                    __syncthreads();
                    if (tid == 0) c.s.ctr[which]++;
                }
                unsigned long long card = 0, z2 = 0;
                uint8_t* crow = c.t.node_cent + ((size_t)leafnode * c.node_rows + jl) * (size_t)c.t.RB;
                for (int b = tid; b < nb; b += TB) {
                    uint32_t v[8], x8[8];
                    cf_load8(c, slotT, b, v);
                    elem_cols(c, b, x8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += x8[k];
                    cf_store8(c, slotN, b, v);
                    const uint32_t byte = centroid_byte(v, new_n);
                    crow[b] = (uint8_t)byte;
                    card += __popc(byte);
                }
                block_sum2(c, card, z2);
                if (tid == 0) {
                    c.t.sub_n[Tsub] = (uint32_t)new_n;
                    c.t.sub_s1[Tsub] = s1n;
                    c.t.sub_s2[Tsub] = s2n;
                    c.t.sub_slot[Tsub] = slotN;
                    c.t.node_card[(size_t)leafnode * c.node_rows + jl] = (uint32_t)card;
                    c.s.stats[2]++;
                }
                out_id = Tsub;
            } else {
                // append_subcluster (bitbirch.py:284-287): new leaf BitFeature
                const uint32_t tier = tier_for(c.nS);
                const int which = tier == 0 ? C_N8 : (tier == 1 ? C_N16 : C_N32);
                const uint32_t s = c.s.ctr[C_SUBS];
                const uint32_t slotw = (tier << 30) | c.s.ctr[which];
                __syncthreads();
                if (tid == 0) {
                    c.s.ctr[C_SUBS]++;
                    c.s.ctr[which]++;
                    c.t.sub_n[s] = c.nS;
                    c.t.sub_s1[s] = c.s1S;
                    c.t.sub_s2[s] = c.s2S;
                    c.t.sub_slot[s] = slotw;
                    c.t.node_len[leafnode] = leaflen + 1;
                    c.s.stats[3]++;
                }
                for (int b = tid; b < nb; b += TB) {
                    uint32_t v[8];
                    elem_cols(c, b, v);
                    cf_store8(c, slotw, b, v);
                }
                node_put_row(c, leafnode, leaflen, c.s.x, c.pcx, s, NONE);
                out_id = s;
                overflow = leaflen + 1 > bf;
            }
        }
        __syncthreads();
        // ---- upward pass -----------------------------------------------------------------
        int upd_levels = D;  // tracking BitFeatures at levels [0, upd_levels) get CF += element
        if (overflow) {
            int lvl = D;
            while (true) {
                const uint32_t nd = c.s.path_node[lvl];
                split_node(c, nd);
                const uint32_t node1 = c.s.bc[0], A = c.s.bc[1], B = c.s.bc[2], cardA = c.s.bc[3], cardB = c.s.bc[4];
                if (lvl == 0) {
                    // root split: new root holding the two tracking BitFeatures (bitbirch.py:778-782)
                    const uint32_t nr = c.s.ctr[C_NODES];
                    __syncthreads();
                    node_put_row(c, nr, 0, c.s.cA, cardA, A, node1);
                    node_put_row(c, nr, 1, c.s.cB, cardB, B, nd);
                    if (tid == 0) {
                        c.s.ctr[C_NODES]++;
                        c.t.node_len[nr] = 2;
                        c.t.node_leaf[nr] = 0;
                        c.t.node_prev[nr] = NONE;
                        c.t.node_next[nr] = NONE;
                        c.s.ctr[C_ROOT] = nr;
                        c.s.ctr[C_DEPTH]++;
                        c.s.stats[5]++;
                    }
                    upd_levels = 0;
                    break;
                }
                // update_split_subclusters (bitbirch.py:289-303)
                const uint32_t P = c.s.path_node[lvl - 1], jp = c.s.path_row[lvl - 1], lenP = c.s.path_len[lvl - 1];
                node_put_row(c, P, jp, c.s.cA, cardA, A, node1);
                node_put_row(c, P, lenP, c.s.cB, cardB, B, nd);
                if (tid == 0) c.t.node_len[P] = lenP + 1;
                __syncthreads();
                if (lenP + 1 > bf) {
                    lvl--;
                    continue;
                }
                upd_levels = lvl - 1;
                break;
            }
            __syncthreads();
        }
        // closest_subcluster.update(subcluster) for the untouched ancestors (bitbirch.py:352-357)
        for (int k = 0; k < upd_levels; ++k) {
            const uint32_t t = c.s.path_sub[k], P = c.s.path_node[k], jp = c.s.path_row[k];
            const uint32_t slotw = c.t.sub_slot[t];
            const unsigned long long n_new = (unsigned long long)c.t.sub_n[t] + c.nS;
            if (n_new > 0xFFFFFFFFull) { stop = STOP_RANGE; break; }
            unsigned long long card = 0, z2 = 0;
            uint8_t* crow = c.t.node_cent + ((size_t)P * c.node_rows + jp) * (size_t)c.t.RB;
            for (int b = tid; b < nb; b += TB) {
                uint32_t v[8], x8[8];
                cf_load8(c, slotw, b, v);
                elem_cols(c, b, x8);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) v[kk] += x8[kk];
                cf_store8(c, slotw, b, v);
                const uint32_t byte = centroid_byte(v, n_new);
                crow[b] = (uint8_t)byte;
                card += __popc(byte);
            }
            block_sum2(c, card, z2);
            if (tid == 0) {
                c.t.sub_n[t] = (uint32_t)n_new;
                c.t.node_card[(size_t)P * c.node_rows + jp] = (uint32_t)card;
            }
        }
        if (stop != STOP_DONE) break;
        if (tid == 0 && out_leaf) out_leaf[e] = out_id;
    }
    __syncthreads();
    if (tid < C_COUNT) T->ctr[tid] = c.s.ctr[tid];
    if (tid < 8) T->stats[tid] = c.s.stats[tid];
    if (tid == 0) {
        T->processed = e;
        T->stop_reason = stop;
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
    const size_t rows = (size_t)t.bf + 1;
    const uint32_t nd = nodes[i], r = rowsidx[i];
    const uint32_t sub = t.node_sub[(size_t)nd * rows + r];
    const uint32_t n = t.sub_n[sub];
    if (threadIdx.x == 0) {
        if (out_n) out_n[i] = n;
        if (out_ids) out_ids[i] = sub;
    }
    if (out_cent) {
        const uint8_t* src = t.node_cent + ((size_t)nd * rows + r) * (size_t)t.RB;
        for (int b = threadIdx.x; b < t.nbytes; b += blockDim.x) out_cent[(size_t)i * t.nbytes + b] = src[b];
    }
    if (out_bufs) {
        const uint32_t slotw = t.sub_slot[sub];
        const uint32_t tier = slotw >> 30;
        const size_t base = (size_t)(slotw & 0x3FFFFFFFu) * (size_t)t.F;
        const size_t cols = (size_t)t.F + (ls_only ? 0 : 1);
        for (int j = threadIdx.x; j < t.F + (ls_only ? 0 : 1); j += blockDim.x) {
            unsigned long long v;
            if (j == t.F) v = n;
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
};

namespace {

template <typename T>
int grow_pool(T*& p, size_t old_elems, size_t new_elems) {
    T* np_ = nullptr;
    BB_HIP(hipMalloc(&np_, new_elems * sizeof(T)));
    if (p && old_elems) BB_HIP(hipMemcpy(np_, p, old_elems * sizeof(T), hipMemcpyDeviceToDevice));
    if (p) BB_HIP(hipFree(p));
    p = np_;
    return BBH_OK;
}

int grow_nodes(bbh_tree* t, uint32_t want) {
    TreeDev& h = t->h;
    if (want <= h.cap_nodes) return BBH_OK;
    const size_t rows = (size_t)h.bf + 1;
    const size_t oc = h.cap_nodes, nc = want;
    BB_TRY(grow_pool(h.node_cent, oc * rows * h.RB, nc * rows * h.RB));
    BB_TRY(grow_pool(h.node_sub, oc * rows, nc * rows));
    BB_TRY(grow_pool(h.node_child, oc * rows, nc * rows));
    BB_TRY(grow_pool(h.node_card, oc * rows, nc * rows));
    BB_TRY(grow_pool(h.node_len, oc, nc));
    BB_TRY(grow_pool(h.node_prev, oc, nc));
    BB_TRY(grow_pool(h.node_next, oc, nc));
    BB_TRY(grow_pool(h.node_leaf, oc, nc));
    // new nodes must start empty
    BB_HIP(hipMemset(h.node_len + oc, 0, (nc - oc) * sizeof(uint32_t)));
    h.cap_nodes = (uint32_t)nc;
    return BBH_OK;
}

int grow_subs(bbh_tree* t, uint32_t want) {
    TreeDev& h = t->h;
    if (want <= h.cap_subs) return BBH_OK;
    const size_t oc = h.cap_subs, nc = want;
    BB_TRY(grow_pool(h.sub_n, oc, nc));
    BB_TRY(grow_pool(h.sub_s1, oc, nc));
    BB_TRY(grow_pool(h.sub_s2, oc, nc));
    BB_TRY(grow_pool(h.sub_slot, oc, nc));
    h.cap_subs = (uint32_t)nc;
    return BBH_OK;
}

int grow_cf(bbh_tree* t, int tier, uint32_t want) {
    TreeDev& h = t->h;
    const size_t F = (size_t)h.F;
    if (tier == 0 && want > h.cap8) {
        BB_TRY(grow_pool(h.cf8, (size_t)h.cap8 * F, (size_t)want * F));
        h.cap8 = want;
    } else if (tier == 1 && want > h.cap16) {
        BB_TRY(grow_pool(h.cf16, (size_t)h.cap16 * F, (size_t)want * F));
        h.cap16 = want;
    } else if (tier == 2 && want > h.cap32) {
        BB_TRY(grow_pool(h.cf32, (size_t)h.cap32 * F, (size_t)want * F));
        h.cap32 = want;
    }
    return BBH_OK;
}

void free_pools(bbh_tree* t) {
    TreeDev& h = t->h;
    void* ptrs[] = {h.node_cent, h.node_sub, h.node_child, h.node_card, h.node_len, h.node_prev, h.node_next,
                    h.node_leaf, h.scratch_cent, h.sub_n, h.sub_s1, h.sub_s2, h.sub_slot, h.cf8, h.cf16, h.cf32};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    h.node_cent = nullptr; h.node_sub = h.node_child = h.node_card = h.node_len = h.node_prev = h.node_next = nullptr;
    h.node_leaf = nullptr; h.scratch_cent = nullptr; h.sub_n = nullptr; h.sub_s1 = h.sub_s2 = nullptr;
    h.sub_slot = nullptr; h.cf8 = nullptr; h.cf16 = nullptr; h.cf32 = nullptr;
    h.cap_nodes = h.cap_subs = h.cap8 = h.cap16 = h.cap32 = 0;
}

// an empty tree: one empty leaf root (bitbirch.py:880-884)
int init_empty(bbh_tree* t) {
    TreeDev& h = t->h;
    std::memset(h.ctr, 0, sizeof(h.ctr));
    BB_TRY(grow_nodes(t, std::max<uint32_t>(h.cap_nodes, 64)));
    BB_TRY(grow_subs(t, std::max<uint32_t>(h.cap_subs, 1024)));
    BB_TRY(grow_cf(t, 0, std::max<uint32_t>(h.cap8, 1024)));
    BB_TRY(grow_cf(t, 1, std::max<uint32_t>(h.cap16, 64)));
    BB_TRY(grow_cf(t, 2, std::max<uint32_t>(h.cap32, 256)));
    const uint32_t zero = 0, one = 1, none = NONE;
    BB_HIP(hipMemcpy(h.node_len, &zero, 4, hipMemcpyHostToDevice));
    BB_HIP(hipMemcpy(h.node_leaf, &one, 4, hipMemcpyHostToDevice));
    BB_HIP(hipMemcpy(h.node_prev, &none, 4, hipMemcpyHostToDevice));
    BB_HIP(hipMemcpy(h.node_next, &none, 4, hipMemcpyHostToDevice));
    h.ctr[C_NODES] = 1;
    h.ctr[C_ROOT] = 0;
    h.ctr[C_FIRST_LEAF] = 0;
    h.ctr[C_DEPTH] = 1;
    std::memset(h.stats, 0, sizeof(h.stats));
    h.stats[5] = 1;
    t->chain_valid = false;
    return BBH_OK;
}

int set_tol(bbh_tree* t, const double* tab, int64_t len) {
    if (t->d_tol) {
        (void)hipFree(t->d_tol);
        t->d_tol = nullptr;
    }
    t->h.tol_table = nullptr;
    t->h.tol_len = 0;
    if (tab && len > 0) {
        BB_HIP(hipMalloc(&t->d_tol, (size_t)len * 8));
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
    t->lds = smem_layout(bf, h.RB, nullptr, nullptr);
    if (h.scratch_cent) (void)hipFree(h.scratch_cent);
    h.scratch_cent = nullptr;
    BB_HIP(hipMalloc(&h.scratch_cent, ((size_t)bf + 1) * h.RB));
    if (t->lds > 48 * 1024)
        BB_HIP(hipFuncSetAttribute((const void*)k_tree_insert, hipFuncAttributeMaxDynamicSharedMemorySize, (int)t->lds));
    return BBH_OK;
}

// run the insertion kernel over n elements, growing pools whenever it stops for capacity
int run_insert(bbh_tree* t, const uint8_t* rows_dev, int64_t row_stride, const uint8_t* bufs_dev, int width, int64_t n,
               uint32_t* out_leaf_dev, hipStream_t s) {
    TreeDev& h = t->h;
    t->chain_valid = false;
    int64_t done = 0;
    int stalls = 0;
    while (done < n) {
        BB_HIP(hipMemcpyAsync(t->d, &h, sizeof(TreeDev), hipMemcpyHostToDevice, s));
        const int64_t chunk = n - done;
        {
            bb::ProfScope ps("tree_insert", s);
            hipLaunchKernelGGL(k_tree_insert, dim3(1), dim3(TB), t->lds, s, t->d,
                               rows_dev ? rows_dev + done * row_stride : nullptr, (long long)row_stride,
                               bufs_dev ? bufs_dev + (size_t)done * ((size_t)h.F + 1) * width : nullptr, width,
                               (long long)chunk, out_leaf_dev ? out_leaf_dev + done : nullptr);
            BB_HIP(hipGetLastError());
        }
        TreeDev back;
        BB_HIP(hipMemcpyAsync(&back, t->d, sizeof(TreeDev), hipMemcpyDeviceToHost, s));
        BB_HIP(hipStreamSynchronize(s));
        std::memcpy(h.ctr, back.ctr, sizeof(h.ctr));
        std::memcpy(h.stats, back.stats, sizeof(h.stats));
        done += back.processed;
        stalls = back.processed == 0 ? stalls + 1 : 0;
        if (stalls > 3) return bb::fail(BBH_ERR_CAPACITY, "tree engine made no progress (stop reason %d)", back.stop_reason);
        const int64_t left = n - done;
        auto more = [&](uint32_t used, uint32_t cap, int64_t per_elem_hint) -> uint32_t {
            uint64_t want = (uint64_t)cap * 2;
            uint64_t est = (uint64_t)used + (uint64_t)(left * per_elem_hint) / 4 + 4096;
            if (est > want) want = est;
            if (want > 0x3FFFFFFFull) want = 0x3FFFFFFFull;
            return (uint32_t)want;
        };
        switch (back.stop_reason) {
            case STOP_DONE: break;
            case STOP_NODES: BB_TRY(grow_nodes(t, more(h.ctr[C_NODES], h.cap_nodes, 1))); break;
            case STOP_SUBS: BB_TRY(grow_subs(t, more(h.ctr[C_SUBS], h.cap_subs, 4))); break;
            case STOP_CF8: BB_TRY(grow_cf(t, 0, more(h.ctr[C_N8], h.cap8, 4))); break;
            case STOP_CF16: BB_TRY(grow_cf(t, 1, more(h.ctr[C_N16], h.cap16, 1))); break;
            case STOP_CF32: BB_TRY(grow_cf(t, 2, more(h.ctr[C_N32], h.cap32, 1))); break;
            case STOP_DEPTH: return bb::fail(BBH_ERR_CAPACITY, "tree deeper than %d levels", MAXD);
            case STOP_RANGE: return bb::fail(BBH_ERR_INVALID, "n_samples exceeds 2^32-1 (engine limit)");
            default: return bb::fail(BBH_ERR_HIP, "unknown stop reason %d", back.stop_reason);
        }
    }
    return BBH_OK;
}

// walk the leaf chain on the host (bitbirch.py:886-893): positions -> (node, row)
int build_chain(bbh_tree* t) {
    if (t->chain_valid) return BBH_OK;
    TreeDev& h = t->h;
    const uint32_t nn = h.ctr[C_NODES];
    std::vector<uint32_t> len(nn), next(nn);
    BB_HIP(hipMemcpy(len.data(), h.node_len, (size_t)nn * 4, hipMemcpyDeviceToHost));
    BB_HIP(hipMemcpy(next.data(), h.node_next, (size_t)nn * 4, hipMemcpyDeviceToHost));
    t->chain_nodes.clear();
    t->chain_rows.clear();
    uint32_t nd = h.ctr[C_FIRST_LEAF];
    size_t guard = 0;
    while (nd != NONE && guard++ <= nn) {
        for (uint32_t r = 0; r < len[nd]; ++r) {
            t->chain_nodes.push_back(nd);
            t->chain_rows.push_back(r);
        }
        nd = next[nd];
    }
    const size_t k = t->chain_nodes.size();
    if (k > t->d_chain_cap) {
        if (t->d_chain_nodes) (void)hipFree(t->d_chain_nodes);
        if (t->d_chain_rows) (void)hipFree(t->d_chain_rows);
        t->d_chain_nodes = t->d_chain_rows = nullptr;
        BB_HIP(hipMalloc(&t->d_chain_nodes, k * 4));
        BB_HIP(hipMalloc(&t->d_chain_rows, k * 4));
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
        hipError_t e = hipMalloc(&t->d, sizeof(TreeDev));
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
    if (t->d) (void)hipFree(t->d);
    if (t->d_tol) (void)hipFree(t->d_tol);
    if (t->d_chain_nodes) (void)hipFree(t->d_chain_nodes);
    if (t->d_chain_rows) (void)hipFree(t->d_chain_rows);
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
        const bool empty = t->h.ctr[C_NODES] == 1 && t->h.ctr[C_SUBS] == 0;
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
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    return init_empty(t);  // pools are kept and reused
}

extern "C" int bbh_tree_fit_packed(bbh_tree* t, const uint8_t* rows, int64_t n, int64_t row_stride,
                                   uint32_t* out_leaf, void* stream) {
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    if (n < 0 || row_stride < t->h.nbytes) return bb::fail(BBH_ERR_INVALID, "rows must be (n, n_features/8) uint8");
    if (n == 0) return BBH_OK;
    BB_HIP(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    // every fingerprint can become a new leaf BitFeature at tier 0
    BB_TRY(grow_subs(t, (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, (uint64_t)t->h.ctr[C_SUBS] + (uint64_t)n + (uint64_t)n / 8 + 1024)));
    BB_TRY(grow_cf(t, 0, (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, (uint64_t)t->h.ctr[C_N8] + (uint64_t)n + 64)));
    BB_TRY(grow_nodes(t, (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, (uint64_t)t->h.ctr[C_NODES] + (uint64_t)n / std::max(1, t->h.bf / 3) + 64)));
    bb::DevOut o;
    BB_TRY(o.init(out_leaf, (size_t)n * 4));
    if (bb::is_device_ptr(rows)) {
        BB_TRY(run_insert(t, rows, row_stride, nullptr, 0, n, (uint32_t*)o.dev, s));
    } else {
        // stage host rows through HBM in slabs (PCIe is outside the engine's hot loop)
        const int64_t slab = std::max<int64_t>(1, (int64_t)(1ull << 30) / row_stride);
        uint8_t* stage = nullptr;
        BB_HIP(hipMalloc(&stage, (size_t)std::min(slab, n) * row_stride));
        int rc = BBH_OK;
        for (int64_t off = 0; off < n && rc == BBH_OK; off += slab) {
            const int64_t m = std::min(slab, n - off);
            hipError_t e = hipMemcpyAsync(stage, rows + off * row_stride, (size_t)m * row_stride, hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "H2D: %s", hipGetErrorString(e)); break; }
            rc = run_insert(t, stage, row_stride, nullptr, 0, m, o.dev ? (uint32_t*)o.dev + off : nullptr, s);
        }
        (void)hipStreamSynchronize(s);
        (void)hipFree(stage);
        BB_TRY(rc);
    }
    BB_TRY(o.finish(s));
    BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

extern "C" int bbh_tree_fit_buffers(bbh_tree* t, const void* bufs, int32_t width, int64_t k, uint32_t* out_leaf,
                                    void* stream) {
    if (!t) return bb::fail(BBH_ERR_INVALID, "null tree");
    if (width != 1 && width != 2 && width != 4 && width != 8) return bb::fail(BBH_ERR_INVALID, "buffer element width must be 1, 2, 4 or 8");
    if (k < 0) return bb::fail(BBH_ERR_INVALID, "negative buffer count");
    if (k == 0) return BBH_OK;
    BB_HIP(hipSetDevice(t->device));
    hipStream_t s = (hipStream_t)stream;
    BB_TRY(grow_subs(t, (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, (uint64_t)t->h.ctr[C_SUBS] + (uint64_t)k + (uint64_t)k / 8 + 1024)));
    if (width == 1) BB_TRY(grow_cf(t, 0, (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, (uint64_t)t->h.ctr[C_N8] + (uint64_t)k + 64)));
    if (width == 2) BB_TRY(grow_cf(t, 1, (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, (uint64_t)t->h.ctr[C_N16] + (uint64_t)k + 64)));
    BB_TRY(grow_nodes(t, (uint32_t)std::min<uint64_t>(0x3FFFFFFFull, (uint64_t)t->h.ctr[C_NODES] + (uint64_t)k / std::max(1, t->h.bf / 3) + 64)));
    const size_t row_bytes = ((size_t)t->h.F + 1) * width;
    bb::DevOut o;
    BB_TRY(o.init(out_leaf, (size_t)k * 4));
    if (bb::is_device_ptr(bufs)) {
        BB_TRY(run_insert(t, nullptr, 0, (const uint8_t*)bufs, width, k, (uint32_t*)o.dev, s));
    } else {
        const int64_t slab = std::max<int64_t>(1, (int64_t)((1ull << 30) / row_bytes));
        uint8_t* stage = nullptr;
        BB_HIP(hipMalloc(&stage, (size_t)std::min(slab, k) * row_bytes));
        int rc = BBH_OK;
        for (int64_t off = 0; off < k && rc == BBH_OK; off += slab) {
            const int64_t m = std::min(slab, k - off);
            hipError_t e = hipMemcpyAsync(stage, (const uint8_t*)bufs + (size_t)off * row_bytes, (size_t)m * row_bytes,
                                          hipMemcpyHostToDevice, s);
            if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "H2D: %s", hipGetErrorString(e)); break; }
            rc = run_insert(t, nullptr, 0, stage, width, m, o.dev ? (uint32_t*)o.dev + off : nullptr, s);
        }
        (void)hipStreamSynchronize(s);
        (void)hipFree(stage);
        BB_TRY(rc);
    }
    BB_TRY(o.finish(s));
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
        hipLaunchKernelGGL(k_gather_leaves, dim3((unsigned)m), dim3(256), 0, nullptr, t->d, d_nodes, d_rows,
                           (long long)m, width ? width : 1, (uint8_t*)ob.dev, ls_only, (uint8_t*)oc.dev,
                           (unsigned long long*)on.dev, (uint32_t*)oi.dev);
        BB_HIP(hipGetLastError());
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

extern "C" int bbh_tree_gather_buffers(bbh_tree* t, const int64_t* positions, int64_t m, int32_t width, void* out) {
    if (!t || (m > 0 && (!positions || !out))) return bb::fail(BBH_ERR_INVALID, "null argument");
    if (width != 1 && width != 2 && width != 4 && width != 8) return bb::fail(BBH_ERR_INVALID, "width must be 1, 2, 4 or 8");
    if (m == 0) return BBH_OK;
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
    BB_HIP(hipMalloc(&dn, (size_t)m * 4));
    BB_HIP(hipMalloc(&dr, (size_t)m * 4));
    int rc = BBH_OK;
    if (hipMemcpy(dn, nodes.data(), (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(dr, rows.data(), (size_t)m * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = bb::fail(BBH_ERR_HIP, "gather_buffers: H2D failed");
    if (rc == BBH_OK) rc = gather(t, dn, dr, m, width, out, 0, nullptr, nullptr, nullptr);
    (void)hipFree(dn);
    (void)hipFree(dr);
    return rc;
}

extern "C" int bbh_tree_stats(bbh_tree* t, uint64_t* out8) {
    if (!t || !out8) return bb::fail(BBH_ERR_INVALID, "null argument");
    for (int i = 0; i < 7; ++i) out8[i] = t->h.stats[i];
    out8[7] = t->h.ctr[C_SUBS];
    return BBH_OK;
}
