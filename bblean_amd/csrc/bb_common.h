// bb_common.h -- shared host/device helpers of libbbhip.so (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bbhip.h"

// ---------------------------------------------------------------------------------------
// error handling: status codes + thread-local message (include/bbhip.h conventions)
// ---------------------------------------------------------------------------------------
namespace bb {

extern thread_local char g_err[512];

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define BB_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return bb::fail(BBH_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                   \
                            hipGetErrorString(_e), __FILE__, __LINE__);                    \
    } while (0)

#define BB_TRY(expr)                 \
    do {                             \
        int _rc = (expr);            \
        if (_rc != BBH_OK) return _rc; \
    } while (0)

// Is p a device-accessible allocation (hipMalloc / torch tensor)?  Plain host memory
// makes hipPointerGetAttributes fail; that error is swallowed here.
inline bool is_device_ptr(const void* p) {
    if (p == nullptr) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

int ensure_device();  // picks device 0 lazily, checks it is gfx950

// ---------------------------------------------------------------------------------------
// Caching device allocator.  hipFree synchronises the device and hipMalloc of small blocks
// costs tens of microseconds; a tree owns eight pools that are re-allocated as it grows and a
// multiround round creates hundreds of trees, so blocks are recycled by size class instead of
// being returned to the driver (bbh_trim_cache releases them): blocks of up to 64 MiB, at most BBHIP_CACHE_MB (default 4 GiB)
// per process; larger blocks go straight back to the driver.
// ---------------------------------------------------------------------------------------
hipError_t dev_alloc(void** p, size_t bytes);
void dev_free(void* p);
void dev_trim();
hipError_t dev_free_bytes(size_t need, size_t* free_b);  // hipMemGetInfo's free bytes; caches are emptied first when that is less than `need`
void set_pressure_callback(void (*fn)(void));  // called when hipMalloc fails, before the one retry (bbh_set_memory_pressure_callback)
template <typename T>
inline hipError_t dev_alloc(T** p, size_t bytes) { return dev_alloc((void**)p, bytes); }

// Input staged to HBM when the caller handed a host pointer.
struct DevIn {
    const void* dev = nullptr;
    void* owned = nullptr;
    int init(const void* p, size_t bytes, hipStream_t s) {
        if (p == nullptr || bytes == 0) {
            dev = p;
            return BBH_OK;
        }
        if (is_device_ptr(p)) {
            dev = p;
            return BBH_OK;
        }
        BB_HIP(dev_alloc(&owned, bytes));
        BB_HIP(hipMemcpyAsync(owned, p, bytes, hipMemcpyHostToDevice, s));
        dev = owned;
        return BBH_OK;
    }
    ~DevIn() {
        if (owned) dev_free(owned);
    }
};

// Output produced in HBM, copied back when the caller handed a host pointer.
struct DevOut {
    void* dev = nullptr;
    void* owned = nullptr;
    void* host = nullptr;
    size_t bytes = 0;
    int init(void* p, size_t nbytes) {
        bytes = nbytes;
        if (p == nullptr || nbytes == 0) {
            dev = nullptr;
            return BBH_OK;
        }
        if (is_device_ptr(p)) {
            dev = p;
            return BBH_OK;
        }
        host = p;
        BB_HIP(dev_alloc(&owned, nbytes));
        dev = owned;
        return BBH_OK;
    }
    bool needs_copy() const { return host != nullptr; }
    int finish(hipStream_t s) {
        if (host) {
            BB_HIP(hipMemcpyAsync(host, owned, bytes, hipMemcpyDeviceToHost, s));
        }
        return BBH_OK;
    }
    ~DevOut() {
        if (owned) dev_free(owned);
    }
};

// ---------------------------------------------------------------------------------------
// per-kernel timing with HIP events on the launch stream (bbh_profile_*)
// ---------------------------------------------------------------------------------------
struct ProfRec {
    hipEvent_t a, b;
    std::string name;
    long long units = 0;  // work units of the launch (rows, inserted elements ...), for bbh_profile_units
};
extern bool g_prof_on;
void prof_begin(const char* name, hipStream_t s, size_t* token);
void prof_units(size_t token, long long units);
void prof_rename(size_t token, const char* name);  // ("tree_insert/pipe": also counted under "tree_insert")
void prof_end(size_t token, hipStream_t s);

struct ProfScope {
    size_t tok = (size_t)-1;
    hipStream_t s;
    ProfScope(const char* name, hipStream_t stream) : s(stream) {
        if (g_prof_on) prof_begin(name, stream, &tok);
    }
    void units(long long u) {
        if (tok != (size_t)-1) prof_units(tok, u);
    }
    ~ProfScope() {
        if (tok != (size_t)-1) prof_end(tok, s);
    }
};

}  // namespace bb

// ---------------------------------------------------------------------------------------
// device helpers: wave64 cross-lane primitives
// ---------------------------------------------------------------------------------------
#if defined(__HIPCC__)

namespace bbd {

constexpr int WAVE = 64;

// DPP row rotate right by N within each 16-lane row (ctrl 0x120 + N); every lane valid.
template <int N>
__device__ __forceinline__ uint32_t row_ror(uint32_t v) {
    // (`old` = 0, the add's identity: the compiler folds the move into ONE v_add_u32_dpp; with `old` = v it emits a move and an add)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xF, 0xF, false);
}

// any other DPP control word inside a 16-lane row (0x141 row_half_mirror, 0x00-0xFF quad_perm), folded into an add the same way
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_ctrl(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

// Sum over the 16 lanes of a DPP row; result in every lane of the row.
__device__ __forceinline__ uint32_t row16_sum(uint32_t v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v = row16_sum(v);
    v += (uint32_t)__shfl_xor((int)v, 16);
    v += (uint32_t)__shfl_xor((int)v, 32);
    return v;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) v += (unsigned long long)__shfl_xor((long long)v, m);
    return v;
}

__device__ __forceinline__ unsigned long long dpp_ror_u64(unsigned long long v, int n) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    switch (n) {
        case 8: lo = row_ror<8>(lo); hi = row_ror<8>(hi); break;
        case 4: lo = row_ror<4>(lo); hi = row_ror<4>(hi); break;
        case 2: lo = row_ror<2>(lo); hi = row_ror<2>(hi); break;
        default: lo = row_ror<1>(lo); hi = row_ror<1>(hi); break;
    }
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    unsigned long long o;
    o = dpp_ror_u64(v, 8); v = o > v ? o : v;
    o = dpp_ror_u64(v, 4); v = o > v ? o : v;
    o = dpp_ror_u64(v, 2); v = o > v ? o : v;
    o = dpp_ror_u64(v, 1); v = o > v ? o : v;
    o = (unsigned long long)__shfl_xor((long long)v, 16); v = o > v ? o : v;
    o = (unsigned long long)__shfl_xor((long long)v, 32); v = o > v ? o : v;
    return v;
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
    return ~wave_max_u64(~v);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// 16-byte streaming load (read-once data: do not keep the line in cache)
__device__ __forceinline__ uint4 ld_nt16(const void* p) {
    u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ uint32_t popc4(uint4 q) {
    return __popc(q.x) + __popc(q.y) + __popc(q.z) + __popc(q.w);
}

__device__ __forceinline__ uint4 and4(uint4 a, uint4 b) {
    return make_uint4(a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w);
}

// Tanimoto exactly as similarity.cpp:326-331: uint32 denominator, clamp in f64, one
// IEEE-754 double division (no fast-math, no contraction).
__device__ __forceinline__ double jt_from_counts(uint32_t inter, uint32_t un) {
    double d = (double)un;
    d = d < 1.0 ? 1.0 : d;
    return (double)inter / d;
}

// iSIM from exact u64 moments, similarity.cpp:297-300 operation order.
__device__ __forceinline__ double isim_from_moments(unsigned long long s1,
                                                    unsigned long long s2,
                                                    unsigned long long n) {
    if (s1 == 0ull) return 1.0;
    double a = (double)(s2 - s1) / 2.0;
    return a / ((a + (double)(n * s1)) - (double)s2);
}

}  // namespace bbd

#endif  // __HIPCC__
