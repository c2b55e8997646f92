// bb_kernels.hip -- stateless similarity kernels of libbbhip.so, hand-written for
// gfx950 (MI355X, CDNA4): wave64, 16-byte per-lane coalesced row loads, DPP row
// reductions, popcount on the VALU (v_bcnt_u32_b32).  No MFMA: this is bit counting.
//
// Each extern "C" entry point replaces one pybind11 binding of the reference's
// bblean/csrc/similarity.cpp (cited at each function and in include/bbhip.h).
#include "bb_common.h"

#include <atomic>

#include <cmath>

using namespace bbd;

namespace bb {

thread_local char g_err[512] = "";
bool g_prof_on = false;
static std::mutex g_prof_mu;
static std::vector<ProfRec> g_prof;

void prof_begin(const char* name, hipStream_t s, size_t* token) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
        *token = (size_t)-1;
        return;
    }
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    *token = g_prof.size() - 1;
}

void prof_rename(size_t token, const char* name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (token < g_prof.size()) g_prof[token].name = name;
}

// a record named "tree_insert/pipe" answers to "tree_insert/pipe" and to "tree_insert"
static bool prof_match(const std::string& rec, const char* query) {
    const size_t n = std::strlen(query);
    return rec.compare(0, n, query) == 0 && (rec.size() == n || rec[n] == '/');
}

void prof_units(size_t token, long long units) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (token < g_prof.size()) g_prof[token].units += units;
}

void prof_end(size_t token, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (token < g_prof.size()) (void)hipEventRecord(g_prof[token].b, s);
}

static int g_device_checked = -1;

// ---- caching device allocator (bb_common.h) ------------------------------------------------
namespace {
struct DevCache {
    std::mutex mu;
    std::map<std::pair<int, size_t>, std::vector<void*>> free_blocks;  // (device, class bytes) -> blocks
    std::map<void*, std::pair<int, size_t>> live;                      // block -> (device, class bytes)
    size_t cached = 0;
    size_t limit = 4ull << 30;
    // Only blocks up to kMaxCached are recycled: the pools of a small tree (a multiround round creates hundreds of them).  A
    // tree of millions of rows frees blocks of gigabytes whose exact size nobody asks for again; retained, they filled the
    // cache to its limit - after which every small block went back to the driver (hipFree synchronises) and came from it
    // again (hipMalloc): bench.py's 512 concurrent shard trees took 0.34 s instead of 0.05 s once a 10 M-row tree had been
    // built and dropped in the same process (round 4's `concurrent_shards` regression) - and kept up to 16 GiB of HBM from
    // whoever needed it next.
    static constexpr size_t kMaxCached = 64ull << 20;
    DevCache() {
        if (const char* e = getenv("BBHIP_CACHE_MB")) limit = (size_t)atoll(e) << 20;
    }
    static size_t size_class(size_t bytes) {
        if (bytes <= 512) return 512;
        if (bytes <= kMaxCached) {
            // powers of two in four steps (1, 1.25, 1.5, 1.75 x 2^k): at most a quarter wasted, sizes meet again
            size_t c = 512;
            while (c * 2 < bytes) c <<= 1;
            if (c >= bytes) return c;
            const size_t q = c / 4;
            return c + (bytes - c + q - 1) / q * q;
        }
        const size_t mb = 1u << 20;
        return (bytes + mb - 1) / mb * mb;
    }
};
DevCache& dev_cache() {
    static DevCache* c = new DevCache();  // never destroyed: blocks may outlive static destructors
    return *c;
}
}  // namespace

static std::atomic<void (*)(void)> g_pressure_cb{nullptr};
void set_pressure_callback(void (*fn)(void)) { g_pressure_cb.store(fn); }

// Free device memory as far as a growing pool may count on it: when the driver reports less than `need` free, the blocks
// this library's cache retains and the host side's (the Python shim registers torch.cuda.empty_cache) are given back first
// and the driver is asked again (gc_nodes / fit_to_memory clamped or refused a pool on the first answer, while
// gigabytes sat in caches that only a FAILED hipMalloc would have emptied).
hipError_t dev_free_bytes(size_t need, size_t* free_b) {
    size_t total_b = 0;
    hipError_t e = hipMemGetInfo(free_b, &total_b);
    if (e != hipSuccess || *free_b >= need) return e;
    dev_trim();
    if (void (*cb)(void) = g_pressure_cb.load()) cb();
    return hipMemGetInfo(free_b, &total_b);
}

// (experiment switch BBHIP_FINE_POOLS=1: every pool of the library in fine-grained device memory - coherent across XCDs by
// memory type instead of by fences; profiles/r06/sys_stability.txt)
static hipError_t raw_device_malloc(void** p, size_t bytes) {
    static const bool fine = [] { const char* v = getenv("BBHIP_FINE_POOLS"); return v && v[0] == '1'; }();
    return fine ? hipExtMallocWithFlags(p, bytes, hipDeviceMallocFinegrained) : hipMalloc(p, bytes);
}

hipError_t dev_alloc(void** p, size_t bytes) {
    DevCache& c = dev_cache();
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t cls = DevCache::size_class(bytes);
    {
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.free_blocks.find({dev, cls});
        if (it != c.free_blocks.end() && !it->second.empty()) {
            *p = it->second.back();
            it->second.pop_back();
            c.cached -= cls;
            c.live[*p] = {dev, cls};
            return hipSuccess;
        }
    }
    e = raw_device_malloc(p, cls);
    if (e != hipSuccess) {  // give the cached blocks back to the driver - ours and the host side's - and retry once
        (void)hipGetLastError();
        dev_trim();
        if (void (*cb)(void) = g_pressure_cb.load()) cb();
        e = raw_device_malloc(p, cls);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> g(c.mu);
    c.live[*p] = {dev, cls};
    return hipSuccess;
}

void dev_free(void* p) {
    if (!p) return;
    DevCache& c = dev_cache();
    {
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.live.find(p);
        if (it != c.live.end()) {
            const std::pair<int, size_t> key = it->second;
            c.live.erase(it);
            if (key.second <= DevCache::kMaxCached && c.cached + key.second <= c.limit) {
                c.free_blocks[key].push_back(p);
                c.cached += key.second;
                return;
            }
        }
    }
    (void)hipFree(p);
}

void dev_trim() {
    DevCache& c = dev_cache();
    std::vector<void*> victims;
    {
        std::lock_guard<std::mutex> g(c.mu);
        for (auto& kv : c.free_blocks)
            for (void* b : kv.second) victims.push_back(b);
        c.free_blocks.clear();
        c.cached = 0;
    }
    for (void* b : victims) (void)hipFree(b);
}

int ensure_device() {
    if (g_device_checked >= 0) return BBH_OK;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0) {
        (void)hipGetLastError();
        return fail(BBH_ERR_NO_DEVICE, "no HIP device visible (libbbhip has no CPU fallback)");
    }
    int dev = 0;
    BB_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    BB_HIP(hipGetDeviceProperties(&p, dev));
    if (std::strncmp(p.gcnArchName, "gfx950", 6) != 0 && getenv("BBHIP_ALLOW_ANY_ARCH") == nullptr)
        return fail(BBH_ERR_NO_DEVICE, "device %d is %s, libbbhip is built for gfx950 only", dev,
                    p.gcnArchName);
    g_device_checked = dev;
    return BBH_OK;
}

}  // namespace bb

// =======================================================================================
// K1: arr-vec Tanimoto / row popcount.  HBM-bound: 256 B read + 8 B written per row
// at 2048 bits.  One wave owns a tile of 64 rows: LPR lanes (16 B each) cover a row,
// 64/LPR rows per load instruction (1 KiB contiguous), all tile loads issued up front;
// the per-row sums are reduced inside the LPR-lane group (DPP when LPR == 16) and
// parked so that after the tile each of the 64 lanes owns one row: one f64 division
// and one fully coalesced 512-byte store per wave.
// =======================================================================================
template <int LPR>
__device__ __forceinline__ uint32_t group_sum(uint32_t v) {
    if constexpr (LPR == 16) {
        return row16_sum(v);
    } else {
#pragma unroll
        for (int m = LPR / 2; m >= 1; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m);
        return v;
    }
}

template <int LPR, bool HAS_VEC>
__global__ __launch_bounds__(256) void k_arr_vec(const uint8_t* __restrict__ arr, int64_t n,
                                                 int chunks_per_row, int64_t row_stride,
                                                 const uint8_t* __restrict__ vec,
                                                 const uint32_t* __restrict__ card_in,
                                                 double* __restrict__ out_sim,
                                                 uint32_t* __restrict__ out_inter,
                                                 uint32_t* __restrict__ out_union,
                                                 uint32_t* __restrict__ out_card) {
    constexpr int RPI = 64 / LPR;  // rows per load instruction
    constexpr int IT = LPR;        // iterations per tile -> 64 rows
    const int lane = threadIdx.x & 63;
    const int l = lane % LPR;  // chunk within the row
    const int g = lane / LPR;  // row within the instruction
    const bool chunk_ok = l < chunks_per_row;
    uint4 v = make_uint4(0, 0, 0, 0);
    uint32_t vec_pc = 0;
    if constexpr (HAS_VEC) {
        if (chunk_ok) v = reinterpret_cast<const uint4*>(vec)[l];
        vec_pc = group_sum<LPR>(popc4(v));
    }
    const int64_t n_tiles = (n + 63) / 64;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
    for (int64_t tile = wave_id; tile < n_tiles; tile += n_waves) {
        const int64_t base = tile * 64;
        uint4 d[IT];
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int64_t row = base + i * RPI + g;
            d[i] = make_uint4(0, 0, 0, 0);
            if (chunk_ok && row < n)
                d[i] = ld_nt16(arr + row * row_stride + (size_t)l * 16);
        }
        uint32_t mine = 0;  // inter | card << 16 of the row this lane ends up owning
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            uint32_t c = popc4(d[i]);
            uint32_t packed = c << 16;
            if constexpr (HAS_VEC) packed |= popc4(and4(d[i], v));
            packed = group_sum<LPR>(packed);
            if (l == i) mine = packed;
        }
        // lane (g, l) owns row base + l * RPI + g
        const int64_t row = base + (int64_t)l * RPI + g;
        if (row < n) {
            uint32_t inter = mine & 0xFFFFu;
            uint32_t card = card_in ? card_in[row] : (mine >> 16);
            if (out_card) out_card[row] = card;
            if constexpr (HAS_VEC) {
                uint32_t un = card + vec_pc - inter;
                if (out_sim) out_sim[row] = jt_from_counts(inter, un);
                if (out_inter) out_inter[row] = inter;
                if (out_union) out_union[row] = un;
            }
        }
    }
}

// generic path: any width / alignment; one wave per row, byte-granular.
template <bool HAS_VEC>
__global__ __launch_bounds__(256) void k_arr_vec_generic(const uint8_t* __restrict__ arr, int64_t n,
                                                         int64_t nbytes, int64_t row_stride,
                                                         const uint8_t* __restrict__ vec,
                                                         const uint32_t* __restrict__ card_in,
                                                         double* __restrict__ out_sim,
                                                         uint32_t* __restrict__ out_inter,
                                                         uint32_t* __restrict__ out_union,
                                                         uint32_t* __restrict__ out_card) {
    const int lane = threadIdx.x & 63;
    const int64_t wave_id = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * (blockDim.x / 64);
    uint32_t vpc = 0;
    if constexpr (HAS_VEC) {
        for (int64_t j = lane; j < nbytes; j += 64) vpc += __popc((uint32_t)vec[j]);
        vpc = wave_sum_u32(vpc);
    }
    for (int64_t row = wave_id; row < n; row += n_waves) {
        const uint8_t* r = arr + row * row_stride;
        uint32_t c = 0, in = 0;
        for (int64_t j = lane; j < nbytes; j += 64) {
            uint32_t b = r[j];
            c += __popc(b);
            if constexpr (HAS_VEC) in += __popc(b & (uint32_t)vec[j]);
        }
        c = wave_sum_u32(c);
        in = wave_sum_u32(in);
        if (lane == 0) {
            uint32_t card = card_in ? card_in[row] : c;
            if (out_card) out_card[row] = card;
            if constexpr (HAS_VEC) {
                uint32_t un = card + vpc - in;
                if (out_sim) out_sim[row] = jt_from_counts(in, un);
                if (out_inter) out_inter[row] = in;
                if (out_union) out_union[row] = un;
            }
        }
    }
}

template <bool HAS_VEC>
static int launch_arr_vec(const uint8_t* arr, int64_t n, int64_t nbytes, int64_t stride,
                          const uint8_t* vec, const uint32_t* card, double* sim, uint32_t* inter,
                          uint32_t* un, uint32_t* out_card, hipStream_t s) {
    if (n == 0) return BBH_OK;
    const bool fast = (nbytes % 16 == 0) && (stride % 16 == 0) && nbytes <= 256 &&
                      ((uintptr_t)arr % 16 == 0) && (!HAS_VEC || (uintptr_t)vec % 16 == 0);
    const int64_t tiles = (n + 63) / 64;
    int64_t blocks = (tiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (!fast) {
        int64_t gb = (n + 3) / 4;
        if (gb > 256 * 8) gb = 256 * 8;
        hipLaunchKernelGGL((k_arr_vec_generic<HAS_VEC>), dim3((unsigned)gb), dim3(256), 0, s, arr, n,
                           nbytes, stride, vec, card, sim, inter, un, out_card);
        BB_HIP(hipGetLastError());
        return BBH_OK;
    }
    const int cpr = (int)(nbytes / 16);
#define BB_LAUNCH_AV(L)                                                                      \
    hipLaunchKernelGGL((k_arr_vec<L, HAS_VEC>), dim3((unsigned)blocks), dim3(256), 0, s, arr, n, \
                       cpr, stride, vec, card, sim, inter, un, out_card)
    if (cpr <= 1) BB_LAUNCH_AV(1);
    else if (cpr <= 2) BB_LAUNCH_AV(2);
    else if (cpr <= 4) BB_LAUNCH_AV(4);
    else if (cpr <= 8) BB_LAUNCH_AV(8);
    else if (cpr <= 16) BB_LAUNCH_AV(16);
    else if (cpr <= 32) BB_LAUNCH_AV(32);
    else BB_LAUNCH_AV(64);
#undef BB_LAUNCH_AV
    BB_HIP(hipGetLastError());
    return BBH_OK;
}

extern "C" int bbh_popcount_rows(const uint8_t* arr, int64_t n, int64_t nbytes, int64_t row_stride,
                                 uint32_t* out, void* stream) {
    BB_TRY(bb::ensure_device());
    if (n < 0 || nbytes <= 0 || row_stride < nbytes)
        return bb::fail(BBH_ERR_INVALID, "Input array must be 2-dimensional");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn a;
    bb::DevOut o;
    BB_TRY(a.init(arr, (size_t)(n * row_stride), s));
    BB_TRY(o.init(out, (size_t)n * 4));
    {
        bb::ProfScope ps("popcount_rows", s);
        BB_TRY(launch_arr_vec<false>((const uint8_t*)a.dev, n, nbytes, row_stride, nullptr, nullptr,
                                     nullptr, nullptr, nullptr, (uint32_t*)o.dev, s));
    }
    BB_TRY(o.finish(s));
    if (a.owned || o.needs_copy()) BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

extern "C" int bbh_jt_arr_vec(const uint8_t* arr, int64_t n, int64_t nbytes, int64_t row_stride,
                              const uint8_t* vec, const uint32_t* card, double* out_sim,
                              uint32_t* out_inter, uint32_t* out_union, void* stream) {
    BB_TRY(bb::ensure_device());
    if (n < 0 || nbytes <= 0 || row_stride < nbytes || vec == nullptr)
        return bb::fail(BBH_ERR_INVALID, "arr must be 2D, vec must be 1D");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn a, v, c;
    bb::DevOut os, oi, ou;
    BB_TRY(a.init(arr, (size_t)(n * row_stride), s));
    BB_TRY(v.init(vec, (size_t)nbytes, s));
    BB_TRY(c.init(card, card ? (size_t)n * 4 : 0, s));
    BB_TRY(os.init(out_sim, (size_t)n * 8));
    BB_TRY(oi.init(out_inter, (size_t)n * 4));
    BB_TRY(ou.init(out_union, (size_t)n * 4));
    {
        bb::ProfScope ps("jt_arr_vec", s);
        BB_TRY(launch_arr_vec<true>((const uint8_t*)a.dev, n, nbytes, row_stride,
                                    (const uint8_t*)v.dev, (const uint32_t*)c.dev, (double*)os.dev,
                                    (uint32_t*)oi.dev, (uint32_t*)ou.dev, nullptr, s));
    }
    BB_TRY(os.finish(s));
    BB_TRY(oi.finish(s));
    BB_TRY(ou.finish(s));
    if (a.owned || v.owned || c.owned || os.needs_copy() || oi.needs_copy() || ou.needs_copy())
        BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

// =======================================================================================
// K2: batched best match.  VALU(v_bcnt)-bound for nc >~ 10: every
// lane owns one query row in registers; centroid rows are wave-uniform, so the compiler
// fetches them through the scalar cache (s_load) and the inner loop is AND + bcnt only,
// with no cross-lane reduction.  First-index argmax by exact integer cross-multiply.
// =======================================================================================
template <int W32>  // 32-bit words per row held in registers (64 for 2048 bits)
__global__ __launch_bounds__(256) void k_best_match(const uint32_t* __restrict__ q, int64_t nq,
                                                    const uint32_t* __restrict__ c, int nc,
                                                    const uint32_t* __restrict__ ccard,
                                                    int32_t* __restrict__ out_idx,
                                                    uint32_t* __restrict__ out_inter,
                                                    uint32_t* __restrict__ out_union,
                                                    double* __restrict__ out_sims) {
    const int64_t qi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = qi < nq;
    uint32_t x[W32];
    uint32_t qc = 0;
    const uint4* src = reinterpret_cast<const uint4*>(q + (ok ? qi : 0) * W32);
#pragma unroll
    for (int w = 0; w < W32 / 4; ++w) {
        uint4 t = ok ? src[w] : make_uint4(0, 0, 0, 0);
        x[4 * w] = t.x;
        x[4 * w + 1] = t.y;
        x[4 * w + 2] = t.z;
        x[4 * w + 3] = t.w;
        qc += popc4(t);
    }
    uint32_t best_i = 0, best_u = 1;
    int best = 0;
    for (int m = 0; m < nc; ++m) {
        const uint32_t* cr = c + (size_t)m * W32;  // wave-uniform address
        uint32_t inter = 0;
#pragma unroll
        for (int w = 0; w < W32; ++w) inter += __popc(x[w] & cr[w]);
        uint32_t un = qc + ccard[m] - inter;
        uint32_t unc = un < 1u ? 1u : un;
        // inter/unc > best_i/best_u  <=>  inter*best_u > best_i*unc (exact; <= 2^26)
        if (m == 0 || inter * best_u > best_i * unc) {
            best_i = inter;
            best_u = unc;
            best = m;
        }
        if (out_sims && ok) out_sims[qi * nc + m] = jt_from_counts(inter, un);
    }
    if (ok) {
        out_idx[qi] = best;
        if (out_inter) out_inter[qi] = best_i;
        // report the true (unclamped) union like the reference's denominator
        if (out_union) out_union[qi] = (best_i == 0 && best_u == 1) ? (qc + ccard[best]) : best_u;
    }
}

// generic width: one wave per query, lanes stride over bytes of every centroid.
__global__ __launch_bounds__(256) void k_best_match_generic(const uint8_t* __restrict__ q, int64_t nq,
                                                            const uint8_t* __restrict__ c, int nc,
                                                            int64_t nbytes,
                                                            const uint32_t* __restrict__ ccard,
                                                            int32_t* __restrict__ out_idx,
                                                            uint32_t* __restrict__ out_inter,
                                                            uint32_t* __restrict__ out_union,
                                                            double* __restrict__ out_sims) {
    const int lane = threadIdx.x & 63;
    const int64_t qi = (int64_t)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (qi >= nq) return;
    const uint8_t* qr = q + qi * nbytes;
    uint32_t qc = 0;
    for (int64_t j = lane; j < nbytes; j += 64) qc += __popc((uint32_t)qr[j]);
    qc = wave_sum_u32(qc);
    uint32_t best_i = 0, best_u = 1, best_true_u = 0;
    int best = 0;
    for (int m = 0; m < nc; ++m) {
        const uint8_t* cr = c + (size_t)m * nbytes;
        uint32_t inter = 0;
        for (int64_t j = lane; j < nbytes; j += 64) inter += __popc((uint32_t)(qr[j] & cr[j]));
        inter = wave_sum_u32(inter);
        uint32_t un = qc + ccard[m] - inter;
        uint32_t unc = un < 1u ? 1u : un;
        if (m == 0 || inter * best_u > best_i * unc) {
            best_i = inter;
            best_u = unc;
            best_true_u = un;
            best = m;
        }
        if (out_sims && lane == 0) out_sims[qi * nc + m] = jt_from_counts(inter, un);
    }
    if (lane == 0) {
        out_idx[qi] = best;
        if (out_inter) out_inter[qi] = best_i;
        if (out_union) out_union[qi] = best_true_u;
    }
}

extern "C" int bbh_jt_best_match(const uint8_t* queries, int64_t nq, const uint8_t* cents, int64_t nc,
                                 int64_t nbytes, int32_t* out_idx, uint32_t* out_inter,
                                 uint32_t* out_union, double* out_sims, void* stream) {
    BB_TRY(bb::ensure_device());
    if (nq < 0 || nc <= 0 || nbytes <= 0 || out_idx == nullptr)
        return bb::fail(BBH_ERR_INVALID, "best_match: need nq >= 0, nc >= 1 and an index output");
    if (nc > (1 << 20)) return bb::fail(BBH_ERR_INVALID, "best_match: too many centroid rows");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn q, c;
    bb::DevOut oi, on, ou, os;
    BB_TRY(q.init(queries, (size_t)(nq * nbytes), s));
    BB_TRY(c.init(cents, (size_t)(nc * nbytes), s));
    BB_TRY(oi.init(out_idx, (size_t)nq * 4));
    BB_TRY(on.init(out_inter, (size_t)nq * 4));
    BB_TRY(ou.init(out_union, (size_t)nq * 4));
    BB_TRY(os.init(out_sims, (size_t)nq * (size_t)nc * 8));
    uint32_t* ccard = nullptr;
    BB_HIP(bb::dev_alloc(&ccard, (size_t)nc * 4));
    int rc = launch_arr_vec<false>((const uint8_t*)c.dev, nc, nbytes, nbytes, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, ccard, s);
    if (rc == BBH_OK && nq > 0) {
        bb::ProfScope ps("jt_best_match", s);
        const bool fast = nbytes == 256 && ((uintptr_t)q.dev % 16 == 0) && ((uintptr_t)c.dev % 16 == 0);
        if (fast) {
            int64_t blocks = (nq + 255) / 256;
            hipLaunchKernelGGL((k_best_match<64>), dim3((unsigned)blocks), dim3(256), 0, s,
                               (const uint32_t*)q.dev, nq, (const uint32_t*)c.dev, (int)nc, ccard,
                               (int32_t*)oi.dev, (uint32_t*)on.dev, (uint32_t*)ou.dev, (double*)os.dev);
        } else {
            int64_t blocks = (nq + 3) / 4;
            hipLaunchKernelGGL(k_best_match_generic, dim3((unsigned)blocks), dim3(256), 0, s,
                               (const uint8_t*)q.dev, nq, (const uint8_t*)c.dev, (int)nc, nbytes, ccard,
                               (int32_t*)oi.dev, (uint32_t*)on.dev, (uint32_t*)ou.dev, (double*)os.dev);
        }
        if (hipGetLastError() != hipSuccess) rc = bb::fail(BBH_ERR_HIP, "best_match launch failed");
    }
    if (rc == BBH_OK) rc = oi.finish(s);
    if (rc == BBH_OK) rc = on.finish(s);
    if (rc == BBH_OK) rc = ou.finish(s);
    if (rc == BBH_OK) rc = os.finish(s);
    hipError_t e = hipStreamSynchronize(s);
    bb::dev_free(ccard);
    if (rc == BBH_OK && e != hipSuccess) rc = bb::fail(BBH_ERR_HIP, "best_match: %s", hipGetErrorString(e));
    return rc;
}

// =======================================================================================
// unpack / add_rows / centroid_from_sum / isim_from_sum
// =======================================================================================
__global__ void k_unpack(const uint8_t* __restrict__ in, int64_t n, int64_t nbytes, int64_t out_bytes,
                         uint8_t* __restrict__ out) {
    // one thread per packed byte -> 8 output bytes, MSB first (similarity.cpp:145-155)
    const int64_t total = n * out_bytes;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / out_bytes, b = t % out_bytes;
        const uint32_t v = in[row * nbytes + b];
        uint8_t* o = out + (row * out_bytes + b) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = (uint8_t)((v >> (7 - k)) & 1u);
    }
}

extern "C" int bbh_unpack(const uint8_t* packed, int64_t n, int64_t nbytes, int64_t n_features,
                          uint8_t* out, void* stream) {
    BB_TRY(bb::ensure_device());
    if (n_features % 8 != 0) return bb::fail(BBH_ERR_INVALID, "Only n_features divisible by 8 is supported");
    if (n < 0 || nbytes <= 0 || n_features > nbytes * 8)
        return bb::fail(BBH_ERR_INVALID, "Input array must be 1- or 2-dimensional");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn a;
    bb::DevOut o;
    BB_TRY(a.init(packed, (size_t)(n * nbytes), s));
    BB_TRY(o.init(out, (size_t)(n * n_features)));
    if (n > 0) {
        bb::ProfScope ps("unpack", s);
        int64_t total = n * (n_features / 8);
        int64_t blocks = (total + 255) / 256;
        if (blocks > 256 * 16) blocks = 256 * 16;
        hipLaunchKernelGGL(k_unpack, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)a.dev, n,
                           nbytes, n_features / 8, (uint8_t*)o.dev);
        BB_HIP(hipGetLastError());
    }
    BB_TRY(o.finish(s));
    if (a.owned || o.needs_copy()) BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

// np.packbits(axis=-1), MSB first (fingerprints.py:46-49): one thread per output byte
__global__ void k_pack(const uint8_t* __restrict__ in, int64_t n, int64_t n_features, int64_t out_bytes,
                       uint8_t* __restrict__ out) {
    const int64_t total = n * out_bytes;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = t / out_bytes, b = t % out_bytes;
        const uint8_t* src = in + row * n_features + b * 8;
        const int64_t left = n_features - b * 8;  // the last byte is zero-padded like np.packbits
        uint32_t v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (k < left) v |= (src[k] != 0 ? 1u : 0u) << (7 - k);
        out[t] = (uint8_t)v;
    }
}

extern "C" int bbh_pack(const uint8_t* unpacked, int64_t n, int64_t n_features, uint8_t* out, void* stream) {
    BB_TRY(bb::ensure_device());
    if (n < 0 || n_features <= 0) return bb::fail(BBH_ERR_INVALID, "Input array must be 1- or 2-dimensional");
    hipStream_t s = (hipStream_t)stream;
    const int64_t out_bytes = (n_features + 7) / 8;
    bb::DevIn a;
    bb::DevOut o;
    BB_TRY(a.init(unpacked, (size_t)(n * n_features), s));
    BB_TRY(o.init(out, (size_t)(n * out_bytes)));
    if (n > 0) {
        bb::ProfScope ps("pack", s);
        ps.units(n);
        int64_t blocks = (n * out_bytes + 255) / 256;
        if (blocks > 65535 * 16) blocks = 65535 * 16;
        hipLaunchKernelGGL(k_pack, dim3((unsigned)blocks), dim3(256), 0, s, (const uint8_t*)a.dev, n, n_features, out_bytes,
                           (uint8_t*)o.dev);
        BB_HIP(hipGetLastError());
    }
    BB_TRY(o.finish(s));
    if (a.owned || o.needs_copy()) BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

// column sums: thread per output byte-group (8 columns when packed, 1 column otherwise),
// rows split over gridDim.y, partials combined with 64-bit atomics.
__global__ void k_add_rows_packed(const uint8_t* __restrict__ arr, int64_t n, int64_t nbytes,
                                  int64_t used_bytes, unsigned long long* __restrict__ out) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= used_bytes) return;
    const int64_t rows_per = (n + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    const int64_t r1 = r0 + rows_per < n ? r0 + rows_per : n;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long big[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t since = 0;
    for (int64_t r = r0; r < r1; ++r) {
        const uint32_t v = arr[r * nbytes + b];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += (v >> (7 - k)) & 1u;
        if (++since == (1 << 30)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { big[k] += acc[k]; acc[k] = 0; }
            since = 0;
        }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned long long t = big[k] + acc[k];
        if (t) atomicAdd(&out[b * 8 + k], t);
    }
}

__global__ void k_add_rows_u8(const uint8_t* __restrict__ arr, int64_t n, int64_t ncols,
                              unsigned long long* __restrict__ out) {
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    const int64_t rows_per = (n + gridDim.y - 1) / gridDim.y;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per;
    const int64_t r1 = r0 + rows_per < n ? r0 + rows_per : n;
    unsigned long long acc = 0;
    for (int64_t r = r0; r < r1; ++r) acc += arr[r * ncols + c];
    if (acc) atomicAdd(&out[c], acc);
}

static int add_rows_dev(const uint8_t* arr, int64_t n, int64_t n_cols, int packed, int64_t n_features,
                        unsigned long long* out_dev, hipStream_t s) {
    BB_HIP(hipMemsetAsync(out_dev, 0, (size_t)n_features * 8, s));
    if (n == 0) return BBH_OK;
    int64_t splits = n / 256;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    bb::ProfScope ps("add_rows", s);
    if (packed) {
        int64_t used = n_features / 8;
        hipLaunchKernelGGL(k_add_rows_packed, dim3((unsigned)((used + 63) / 64), (unsigned)splits),
                           dim3(64), 0, s, arr, n, n_cols, used, out_dev);
    } else {
        hipLaunchKernelGGL(k_add_rows_u8, dim3((unsigned)((n_cols + 63) / 64), (unsigned)splits), dim3(64),
                           0, s, arr, n, n_cols, out_dev);
    }
    BB_HIP(hipGetLastError());
    return BBH_OK;
}

extern "C" int bbh_add_rows(const uint8_t* arr, int64_t n, int64_t n_cols, int packed,
                            int64_t n_features, uint64_t* out, void* stream) {
    BB_TRY(bb::ensure_device());
    if (n < 0 || n_cols <= 0) return bb::fail(BBH_ERR_INVALID, "Input array must be 2-dimensional");
    if (packed && (n_features % 8 != 0 || n_features > n_cols * 8))
        return bb::fail(BBH_ERR_INVALID, "Only n_features divisible by 8 is supported");
    if (!packed && n_features != n_cols) return bb::fail(BBH_ERR_INVALID, "n_features must equal the column count");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn a;
    bb::DevOut o;
    BB_TRY(a.init(arr, (size_t)(n * n_cols), s));
    BB_TRY(o.init(out, (size_t)n_features * 8));
    BB_TRY(add_rows_dev((const uint8_t*)a.dev, n, n_cols, packed, n_features, (unsigned long long*)o.dev, s));
    BB_TRY(o.finish(s));
    if (a.owned || o.needs_copy()) BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

__device__ __forceinline__ unsigned long long load_ls(const void* p, int width, int64_t j) {
    switch (width) {
        case 1: return ((const uint8_t*)p)[j];
        case 2: return ((const uint16_t*)p)[j];
        case 4: return ((const uint32_t*)p)[j];
        default: return ((const unsigned long long*)p)[j];
    }
}

// one thread per output byte (pack) or per feature (no pack)
__global__ void k_centroid_from_sum(const void* __restrict__ ls, int width, int64_t nf, int64_t n_samples,
                                    int pack, uint8_t* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (!pack) {
        if (t >= nf) return;
        unsigned long long v = load_ls(ls, width, t);
        out[t] = n_samples <= 1 ? (uint8_t)v : (uint8_t)((double)v >= (double)n_samples * 0.5);
        return;
    }
    const int64_t nb = (nf + 7) / 8;
    if (t >= nb) return;
    uint32_t byte = 0;
    for (int k = 0; k < 8; ++k) {
        const int64_t j = t * 8 + k;
        if (j >= nf) break;
        unsigned long long v = load_ls(ls, width, j);
        // n<=1: cast to uint8, packbits treats non-zero as 1 (_py_similarity.py:36-41)
        bool bit = n_samples <= 1 ? ((uint8_t)v != 0) : ((double)v >= (double)n_samples * 0.5);
        if (bit) byte |= 0x80u >> k;
    }
    out[t] = (uint8_t)byte;
}

extern "C" int bbh_centroid_from_sum(const void* linear_sum, int32_t ls_width, int64_t n_features,
                                     int64_t n_samples, int pack, uint8_t* out, void* stream) {
    BB_TRY(bb::ensure_device());
    if (n_features <= 0 || (ls_width != 1 && ls_width != 2 && ls_width != 4 && ls_width != 8))
        return bb::fail(BBH_ERR_INVALID, "linear_sum must be 1-dimensional");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn a;
    bb::DevOut o;
    const int64_t out_n = pack ? (n_features + 7) / 8 : n_features;
    BB_TRY(a.init(linear_sum, (size_t)(n_features * ls_width), s));
    BB_TRY(o.init(out, (size_t)out_n));
    {
        bb::ProfScope ps("centroid_from_sum", s);
        hipLaunchKernelGGL(k_centroid_from_sum, dim3((unsigned)((out_n + 255) / 256)), dim3(256), 0, s, a.dev,
                           (int)ls_width, n_features, n_samples, pack, (uint8_t*)o.dev);
        BB_HIP(hipGetLastError());
    }
    BB_TRY(o.finish(s));
    if (a.owned || o.needs_copy()) BB_HIP(hipStreamSynchronize(s));
    return BBH_OK;
}

// exact u64 moments (wrap-around mod 2^64 like the reference's uint64 accumulators),
// then the f64 formula; single block.
__global__ __launch_bounds__(256) void k_isim_from_sum(const void* __restrict__ ls, int width, int64_t nf,
                                                       long long n_objects, double* __restrict__ out) {
    __shared__ unsigned long long r1[4], r2[4];
    unsigned long long s1 = 0, s2 = 0;
    for (int64_t j = threadIdx.x; j < nf; j += blockDim.x) {
        unsigned long long v = load_ls(ls, width, j);
        s1 += v;
        s2 += v * v;
    }
    s1 = wave_sum_u64(s1);
    s2 = wave_sum_u64(s2);
    if ((threadIdx.x & 63) == 0) {
        r1[threadIdx.x >> 6] = s1;
        r2[threadIdx.x >> 6] = s2;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long a = r1[0] + r1[1] + r1[2] + r1[3];
        unsigned long long b = r2[0] + r2[1] + r2[2] + r2[3];
        *out = isim_from_moments(a, b, (unsigned long long)n_objects);
    }
}

static int isim_dev(const void* ls_dev, int width, int64_t nf, int64_t n_objects, double* out_host,
                    int* warn, hipStream_t s) {
    if (warn) *warn = 0;
    if (n_objects < 2) {  // similarity.cpp:275-279
        if (warn) *warn = 1;
        *out_host = NAN;
        return BBH_OK;
    }
    double* d = nullptr;
    BB_HIP(bb::dev_alloc(&d, 8));
    {
        bb::ProfScope ps("isim_from_sum", s);
        hipLaunchKernelGGL(k_isim_from_sum, dim3(1), dim3(256), 0, s, ls_dev, width, nf, (long long)n_objects, d);
    }
    hipError_t e = hipMemcpyAsync(out_host, d, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    bb::dev_free(d);
    if (e != hipSuccess) return bb::fail(BBH_ERR_HIP, "isim_from_sum: %s", hipGetErrorString(e));
    return BBH_OK;
}

extern "C" int bbh_isim_from_sum(const void* linear_sum, int32_t ls_width, int64_t n_features,
                                 int64_t n_objects, double* out, int* warn, void* stream) {
    BB_TRY(bb::ensure_device());
    if (out == nullptr || n_features <= 0 ||
        (ls_width != 1 && ls_width != 2 && ls_width != 4 && ls_width != 8))
        return bb::fail(BBH_ERR_INVALID, "linear_sum must be a 1D array");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn a;
    BB_TRY(a.init(linear_sum, (size_t)(n_features * ls_width), s));
    return isim_dev(a.dev, ls_width, n_features, n_objects, out, warn, s);
}

extern "C" int bbh_isim_rows(const uint8_t* arr, int64_t n, int64_t n_cols, int packed,
                             int64_t n_features, double* out, int* warn, void* stream) {
    BB_TRY(bb::ensure_device());
    if (out == nullptr || n < 0 || n_cols <= 0) return bb::fail(BBH_ERR_INVALID, "Input array must be 2-dimensional");
    if (packed && (n_features % 8 != 0 || n_features > n_cols * 8))
        return bb::fail(BBH_ERR_INVALID, "Only n_features divisible by 8 is supported");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn a;
    BB_TRY(a.init(arr, (size_t)(n * n_cols), s));
    unsigned long long* ls = nullptr;
    BB_HIP(bb::dev_alloc(&ls, (size_t)n_features * 8));
    int rc = add_rows_dev((const uint8_t*)a.dev, n, n_cols, packed, n_features, ls, s);
    if (rc == BBH_OK) rc = isim_dev(ls, 8, n_features, n, out, warn, s);
    (void)hipStreamSynchronize(s);
    bb::dev_free(ls);
    return rc;
}

// =======================================================================================
// The pair loop of metrics.jt_isim_dunn (bblean/metrics.py:186-199): for every pair of clusters i < j the iSIM of their
// combined column sums, 1 - iSIM = the pair's gap, the minimum over all pairs.  With a = sums_i, b = sums_j:
// sum(a + b) = s1_i + s1_j and sum((a + b)^2) = s2_i + s2_j + 2 a.b, so a pair needs ONE exact uint64 dot product (one
// wave per pair, a cluster's row 64 lanes x F / 64 features) and the reference's float64 formula in its operation order
// (similarity.cpp:297-300).  Gaps are >= 0: their bit patterns order like the values, the minimum is an atomicMin.
// =======================================================================================
__global__ __launch_bounds__(256) void k_isim_pair_min(const unsigned long long* __restrict__ sums, const unsigned long long* __restrict__ sizes,
                                                       const unsigned long long* __restrict__ s1, const unsigned long long* __restrict__ s2,
                                                       long long k, long long F, unsigned long long* __restrict__ out_bits) {
    const long long i = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned long long* a = sums + (size_t)i * (size_t)F;
    unsigned long long best = 0x3FF0000000000000ull;  // 1.0
    for (long long j = i + 1 + (long long)blockIdx.x * 4 + wave; j < k; j += (long long)gridDim.x * 4) {
        const unsigned long long* b = sums + (size_t)j * (size_t)F;
        unsigned long long dot = 0;
        for (long long f = lane; f < F; f += 64) dot += a[f] * b[f];
        dot = wave_sum_u64(dot);
        const unsigned long long t1 = s1[i] + s1[j], t2 = s2[i] + s2[j] + 2ull * dot, n = sizes[i] + sizes[j];
        const double gap = 1.0 - isim_from_moments(t1, t2, n);
        const unsigned long long bits = (unsigned long long)__double_as_longlong(gap < 0.0 ? 0.0 : gap);
        best = bits < best ? bits : best;
    }
    if (lane == 0) atomicMin(out_bits, best);
}

extern "C" int bbh_isim_pair_min_gap(const uint64_t* sums, const uint64_t* sizes, int64_t k, int64_t n_features, double* out,
                                     void* stream) {
    BB_TRY(bb::ensure_device());
    if (!sums || !sizes || !out || k < 0 || n_features <= 0) return bb::fail(BBH_ERR_INVALID, "null or empty argument");
    *out = 1.0;
    if (k < 2) return BBH_OK;
    hipStream_t s = (hipStream_t)stream;
    // per-cluster moments on the host (exact uint64, k x F additions): the kernel's work is the k^2 / 2 dot products
    std::vector<unsigned long long> h1((size_t)k), h2((size_t)k);
    std::vector<unsigned long long> hs;
    const unsigned long long* hsums = nullptr;
    if (bb::is_device_ptr(sums)) {
        hs.resize((size_t)k * (size_t)n_features);
        BB_HIP(hipMemcpy(hs.data(), sums, hs.size() * 8, hipMemcpyDeviceToHost));
        hsums = hs.data();
    } else {
        hsums = (const unsigned long long*)sums;
    }
    for (int64_t c = 0; c < k; ++c) {
        unsigned long long a1 = 0, a2 = 0;
        for (int64_t f = 0; f < n_features; ++f) {
            const unsigned long long v = hsums[(size_t)c * (size_t)n_features + (size_t)f];
            a1 += v;
            a2 += v * v;
        }
        h1[(size_t)c] = a1;
        h2[(size_t)c] = a2;
    }
    bb::DevIn dsums, dsizes, d1, d2;
    BB_TRY(dsums.init(sums, (size_t)k * (size_t)n_features * 8, s));
    BB_TRY(dsizes.init(sizes, (size_t)k * 8, s));
    BB_TRY(d1.init(h1.data(), (size_t)k * 8, s));
    BB_TRY(d2.init(h2.data(), (size_t)k * 8, s));
    unsigned long long* dout = nullptr;
    BB_HIP(bb::dev_alloc(&dout, 8));
    const unsigned long long one = 0x3FF0000000000000ull;
    int rc = BBH_OK;
    do {
        hipError_t e = hipMemcpyAsync(dout, &one, 8, hipMemcpyHostToDevice, s);
        if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "H2D: %s", hipGetErrorString(e)); break; }
        bb::ProfScope ps("isim_pair_min", s);
        // (grid.y = the pair's first cluster, at most 65 535 per launch; grid.x spreads its partners over workgroups)
        for (int64_t i0 = 0; i0 + 1 < k && rc == BBH_OK; i0 += 65535) {
            const unsigned gy = (unsigned)std::min<int64_t>(65535, k - 1 - i0);
            const unsigned gx = (unsigned)std::min<int64_t>(64, (k + 3) / 4);
            hipLaunchKernelGGL(k_isim_pair_min, dim3(gx, gy), dim3(256), 0, s, (const unsigned long long*)dsums.dev + (size_t)i0 * (size_t)n_features,
                               (const unsigned long long*)dsizes.dev + i0, (const unsigned long long*)d1.dev + i0, (const unsigned long long*)d2.dev + i0,
                               (long long)(k - i0), (long long)n_features, dout);
            e = hipGetLastError();
            if (e != hipSuccess) rc = bb::fail(BBH_ERR_HIP, "k_isim_pair_min: %s", hipGetErrorString(e));
        }
        if (rc != BBH_OK) break;
        unsigned long long bits = one;
        e = hipMemcpyAsync(&bits, dout, 8, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { rc = bb::fail(BBH_ERR_HIP, "D2H: %s", hipGetErrorString(e)); break; }
        std::memcpy(out, &bits, 8);
    } while (false);
    (void)hipStreamSynchronize(s);
    bb::dev_free(dout);
    return rc;
}

// =======================================================================================
// jt_most_dissimilar_packed (similarity.cpp:413-471) composed from the kernels above.
// The tree engine has its own fused in-kernel version (bb_tree.hip, split_node).
// =======================================================================================
// first index of the minimum (std::min_element / np.argmin: the first one wins), one block; then the row it names
// becomes the comparison vector of the next pass - both on the device, no host round trip in between
__global__ __launch_bounds__(256) void k_first_argmin_row(const double* __restrict__ v, long long n, const uint8_t* __restrict__ Y,
                                                          long long nbytes, long long* __restrict__ out_idx,
                                                          uint8_t* __restrict__ out_row) {
    __shared__ double sv[256];
    __shared__ long long si[256];
    __shared__ long long best;
    const int t = threadIdx.x;
    double bv = 0.0;
    long long bi = -1;
    for (long long i = t; i < n; i += 256) {
        const double x = v[i];
        if (bi < 0 || x < bv) { bv = x; bi = i; }  // ascending i per thread: strict < keeps the first
    }
    sv[t] = bv;
    si[t] = bi;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) {
            const long long oi = si[t + w];
            const double ov = sv[t + w];
            if (oi >= 0 && (si[t] < 0 || ov < sv[t] || (ov == sv[t] && oi < si[t]))) { sv[t] = ov; si[t] = oi; }
        }
        __syncthreads();
    }
    if (t == 0) { best = si[0]; *out_idx = si[0]; }
    __syncthreads();
    const long long b = best;
    for (long long i = t; i < nbytes; i += 256) out_row[i] = Y[b * nbytes + i];
}

extern "C" int bbh_most_dissimilar(const uint8_t* Y, int64_t n, int64_t nbytes, int64_t n_features,
                                   int64_t* idx1, int64_t* idx2, double* sims1, double* sims2,
                                   void* stream) {
    BB_TRY(bb::ensure_device());
    if (n <= 0 || nbytes <= 0 || idx1 == nullptr || idx2 == nullptr)
        return bb::fail(BBH_ERR_INVALID, "Input array must be 2-dimensional");
    if (n_features % 8 != 0 || n_features > nbytes * 8)
        return bb::fail(BBH_ERR_INVALID, "Only features divisible by 8 is supported");
    hipStream_t s = (hipStream_t)stream;
    bb::DevIn y;
    bb::DevOut o1, o2;
    BB_TRY(y.init(Y, (size_t)(n * nbytes), s));
    BB_TRY(o1.init(sims1, (size_t)n * 8));
    BB_TRY(o2.init(sims2, (size_t)n * 8));
    const uint8_t* yd = (const uint8_t*)y.dev;
    void* scratch = nullptr;
    const size_t rowb = ((size_t)nbytes + 15) / 16 * 16;
    const size_t sz_ls = (size_t)nbytes * 8 * 8, sz_card = (size_t)n * 4, sz_sim = (size_t)n * 8;
    const size_t off_cen = sz_ls, off_r1 = off_cen + rowb, off_r2 = off_r1 + rowb, off_idx = off_r2 + rowb;
    const size_t off_card = off_idx + 16;
    const size_t off_sim = (off_card + sz_card + 15) / 16 * 16;
    BB_HIP(bb::dev_alloc(&scratch, off_sim + 2 * sz_sim));
    auto* ls = (unsigned long long*)scratch;
    auto* cen = (uint8_t*)scratch + off_cen;
    auto* row1 = (uint8_t*)scratch + off_r1;
    auto* row2 = (uint8_t*)scratch + off_r2;
    auto* didx = (long long*)((uint8_t*)scratch + off_idx);
    auto* card = (uint32_t*)((uint8_t*)scratch + off_card);
    auto* simc = (double*)((uint8_t*)scratch + off_sim);
    auto* sim_tmp = simc + n;
    long long hidx[2] = {0, 0};
    int rc = BBH_OK;
    auto run = [&]() -> int {
        BB_HIP(hipMemsetAsync(cen, 0, (size_t)nbytes, s));
        BB_TRY(add_rows_dev(yd, n, nbytes, 1, n_features, ls, s));
        hipLaunchKernelGGL(k_centroid_from_sum, dim3((unsigned)((n_features / 8 + 255) / 256)), dim3(256), 0, s,
                           (const void*)ls, 8, n_features, n, 1, cen);
        BB_TRY(launch_arr_vec<false>(yd, n, nbytes, nbytes, nullptr, nullptr, nullptr, nullptr, nullptr, card, s));
        BB_TRY(launch_arr_vec<true>(yd, n, nbytes, nbytes, cen, card, simc, nullptr, nullptr, nullptr, s));
        hipLaunchKernelGGL(k_first_argmin_row, dim3(1), dim3(256), 0, s, (const double*)simc, (long long)n, yd, (long long)nbytes,
                           didx, row1);
        double* d1 = o1.dev ? (double*)o1.dev : sim_tmp;
        BB_TRY(launch_arr_vec<true>(yd, n, nbytes, nbytes, row1, card, d1, nullptr, nullptr, nullptr, s));
        hipLaunchKernelGGL(k_first_argmin_row, dim3(1), dim3(256), 0, s, (const double*)d1, (long long)n, yd, (long long)nbytes,
                           didx + 1, row2);
        if (o2.dev)
            BB_TRY(launch_arr_vec<true>(yd, n, nbytes, nbytes, row2, card, (double*)o2.dev, nullptr, nullptr, nullptr, s));
        BB_HIP(hipGetLastError());
        BB_HIP(hipMemcpyAsync(hidx, didx, 16, hipMemcpyDeviceToHost, s));
        return BBH_OK;
    };
    {
        bb::ProfScope ps("most_dissimilar", s);
        rc = run();
    }
    if (rc == BBH_OK) rc = o1.finish(s);
    if (rc == BBH_OK) rc = o2.finish(s);
    hipError_t e = hipStreamSynchronize(s);
    bb::dev_free(scratch);
    if (rc == BBH_OK && e != hipSuccess) rc = bb::fail(BBH_ERR_HIP, "most_dissimilar: %s", hipGetErrorString(e));
    if (rc == BBH_OK) {
        *idx1 = hidx[0];
        *idx2 = hidx[1];
    }
    return rc;
}

// =======================================================================================
// misc C ABI
// =======================================================================================
extern "C" const char* bbh_last_error(void) { return bb::g_err; }

extern "C" int bbh_set_memory_pressure_callback(void (*fn)(void)) {
    bb::set_pressure_callback(fn);
    return BBH_OK;
}

extern "C" int bbh_trim_cache(void) {
    bb::dev_trim();
    return BBH_OK;
}

extern "C" int bbh_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int bbh_device_info(int device, char* buf, size_t buflen) {
    hipDeviceProp_t p;
    BB_HIP(hipGetDeviceProperties(&p, device));
    snprintf(buf, buflen, "%s %s CUs=%d HBM=%.1fGiB clock=%dMHz", p.gcnArchName, p.name,
             p.multiProcessorCount, (double)p.totalGlobalMem / (1024.0 * 1024.0 * 1024.0), p.clockRate / 1000);
    return BBH_OK;
}

extern "C" int bbh_profile_enable(int on) {
    bb::g_prof_on = on != 0;
    return BBH_OK;
}

extern "C" int bbh_profile_reset(void) {
    std::lock_guard<std::mutex> lk(bb::g_prof_mu);
    for (auto& r : bb::g_prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    bb::g_prof.clear();
    return BBH_OK;
}

extern "C" int bbh_profile_units(const char* name, int64_t* units) {
    std::lock_guard<std::mutex> lk(bb::g_prof_mu);
    int64_t u = 0;
    for (auto& r : bb::g_prof)
        if (bb::prof_match(r.name, name)) u += r.units;
    if (units) *units = u;
    return BBH_OK;
}

extern "C" int bbh_profile_longest(const char* name, double* ms_out, int64_t* units_out) {
    std::lock_guard<std::mutex> lk(bb::g_prof_mu);
    double best = -1.0;
    int64_t u = 0;
    for (auto& r : bb::g_prof) {
        if (!bb::prof_match(r.name, name)) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess && (double)t > best) {
            best = t;
            u = r.units;
        }
    }
    if (ms_out) *ms_out = best < 0.0 ? 0.0 : best;
    if (units_out) *units_out = u;
    return BBH_OK;
}

extern "C" int bbh_profile_get(const char* name, int64_t* launches, double* total_ms) {
    std::lock_guard<std::mutex> lk(bb::g_prof_mu);
    int64_t cnt = 0;
    double ms = 0.0;
    for (auto& r : bb::g_prof) {
        if (!bb::prof_match(r.name, name)) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms += t;
            cnt++;
        }
    }
    if (launches) *launches = cnt;
    if (total_ms) *total_ms = ms;
    return BBH_OK;
}
