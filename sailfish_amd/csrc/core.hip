// core.hip -- error slot, logger hook, device info for libsfgpu.
#include "common.h"

#include <mutex>
#include <unordered_map>
#include <vector>

namespace sfgpu {

static thread_local char g_err[512] = "";
static void (*g_logger)(int, const char*) = nullptr;

void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    if (g_logger) g_logger(2, g_err);
}

void log_msg(int level, const char* fmt, ...) {
    if (!g_logger) return;
    char buf[512];
    va_list ap; va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_logger(level, buf);
}


// ---- caching allocators (see common.h) ----
// Device blocks, pinned host blocks and non-blocking streams are all expensive to create and destroy
// (hipFree / hipHostFree / hipStreamDestroy synchronise the device: 0.3-0.5 ms each, measured), and a
// quantification step creates and drops dozens of them: all three are recycled per device.
namespace {
std::mutex g_pool_mu;
struct Key { int dev; int kind; size_t sz; bool operator==(const Key& o) const { return dev == o.dev && kind == o.kind && sz == o.sz; } };
struct KeyHash { size_t operator()(const Key& k) const { return std::hash<size_t>()(k.sz * 31u + (size_t)k.dev * 2u + (size_t)k.kind); } };
std::unordered_map<Key, std::vector<void*>, KeyHash> g_pool_free;   // (device, kind, rounded size) -> free blocks
std::unordered_map<void*, Key> g_pool_key;                           // live or cached block -> its key
std::unordered_map<int, std::vector<hipStream_t>> g_streams;         // device -> idle non-blocking streams
enum { kDeviceMem = 0, kPinnedMem = 1, kUncachedMem = 2 };      // (uncached: device memory no L2 may hold a line of -- the persistent EM loop's exchange buffer)
// size classes: powers of two up to 1 GiB; above that eight steps per octave (a 38 GB block is 40 GB, not 64: mapping a block
// for the first time costs ~15 ms per GB) -- still few distinct keys, so freed blocks keep finding takers
size_t round_up_pow2(size_t n) {
    size_t r = 256;
    while (r < n && r < ((size_t)1 << 30)) r <<= 1;
    if (r >= n) return r;
    while ((r << 1) < n) r <<= 1;                 // r < n <= 2 r
    const size_t step = r >> 3;
    return r + (n - r + step - 1) / step * step;
}
int cur_device() { int d = 0; (void)hipGetDevice(&d); return d; }

// LARGE device blocks (>= 1 GiB: the Gibbs sampler's chain state is 4 x nnz x chains bytes, 38 GB at cfg3) are cached only up
// to a budget: memory parked here is invisible to every other allocator of the process (torch's caching allocator cannot
// reclaim it and would report out-of-memory).  Default: a quarter of the device's memory, at most 64 GiB;
// SFGPU_POOL_LARGE_LIMIT_GB / sfgpu_pool_set_large_limit() change it (0: large blocks are never cached).
constexpr size_t kLargeBlock = (size_t)1 << 30;
std::unordered_map<int, size_t> g_large_cached;                      // device -> bytes of cached (free) large blocks
long long g_large_limit_user = -1;                                   // bytes, set by sfgpu_pool_set_large_limit / the environment for EVERY device; -1: per-device default
std::unordered_map<int, long long> g_large_limit_dev;                // device -> its default (a quarter of ITS memory, at most 64 GiB)
// (the limit of device `dev`: decided with that device current -- a node with unlike devices, or a first call made on the small one,
//  gave every device the first device's quarter before)
size_t large_limit_locked(int dev) {
    static const long long env_lim = []() -> long long {
        if (const char* e = getenv("SFGPU_POOL_LARGE_LIMIT_GB")) { const double g = atof(e); if (g >= 0) return (long long)(g * (double)(1ull << 30)); }
        return -1;
    }();
    if (g_large_limit_user >= 0) return (size_t)g_large_limit_user;
    if (env_lim >= 0) return (size_t)env_lim;
    auto it = g_large_limit_dev.find(dev);
    if (it != g_large_limit_dev.end()) return (size_t)it->second;
    long long lim = 64ll << 30;
    int cur = 0;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess;
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && (long long)(tot / 4) < lim) lim = (long long)(tot / 4);
    (void)hipGetLastError();
    if (switched) (void)hipSetDevice(cur);
    g_large_limit_dev[dev] = lim;
    return (size_t)lim;
}
// a large block goes back to the driver because of the budget: said once (mapping it again costs ~15 ms per GB).  The note is only
// RECORDED under g_pool_mu; the logger callback runs after the lock is released (flush_large_note): a logger that allocates or frees
// through the pool, or calls sfgpu_pool_*, would deadlock otherwise.
struct LargeNote { bool pending = false, said = false; size_t sz = 0; int dev = 0; size_t limit = 0; } g_large_note;
void note_large_release(size_t sz, int dev) {
    if (g_large_note.said) return;
    g_large_note.said = true; g_large_note.pending = true;
    g_large_note.sz = sz; g_large_note.dev = dev; g_large_note.limit = large_limit_locked(dev);
}
void flush_large_note() {                                      // (called WITHOUT g_pool_mu)
    LargeNote n;
    { std::lock_guard<std::mutex> lk(g_pool_mu); if (!g_large_note.pending) return; n = g_large_note; g_large_note.pending = false; }
    log_msg(0, "device block of %.1f GB released instead of cached (large-block budget of device %d: %.1f GB; SFGPU_POOL_LARGE_LIMIT_GB / sfgpu_pool_set_large_limit change it)",
            (double)n.sz / (double)(1ull << 30), n.dev, (double)n.limit / (double)(1ull << 30));
}

struct Pending { void* p; hipEvent_t ev; bool owns; };      // (blocks freed together share an event; the last one of them gives it back)
std::vector<Pending> g_pending;
std::vector<hipEvent_t> g_events;
void reap_pending_locked(bool wait) {
    size_t keep = 0;
    hipEvent_t seen = nullptr; hipError_t seen_q = hipSuccess;
    for (size_t i = 0; i < g_pending.size(); ++i) {
        Pending& x = g_pending[i];
        hipError_t q = (x.ev == seen) ? seen_q : (wait ? hipEventSynchronize(x.ev) : hipEventQuery(x.ev));
        seen = x.ev; seen_q = q;
        if (q == hipErrorNotReady) { g_pending[keep++] = x; continue; }
        (void)hipGetLastError();
        auto it = g_pool_key.find(x.p);
        if (it == g_pool_key.end()) (void)hipFree(x.p);
        else if (it->second.kind == kDeviceMem && it->second.sz >= kLargeBlock && g_large_cached[it->second.dev] + it->second.sz > large_limit_locked(it->second.dev)) {
            note_large_release(it->second.sz, it->second.dev);
            g_pool_key.erase(it); (void)hipFree(x.p);                        // over the large-block budget: back to the driver
        } else {
            if (it->second.kind == kDeviceMem && it->second.sz >= kLargeBlock) g_large_cached[it->second.dev] += it->second.sz;
            g_pool_free[it->second].push_back(x.p);
        }
        if (x.owns) g_events.push_back(x.ev);
    }
    g_pending.resize(keep);
}

hipError_t pool_get(void** p, size_t bytes, int kind) {
    const Key key{cur_device(), kind, round_up_pow2(bytes ? bytes : 1)};
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        // (blocks freed stream-ordered are looked for only when the cache has none of this size, or when many are waiting: a query
        //  per waiting block in front of EVERY allocation was hundreds of event queries per plan)
        for (int pass = 0; pass < 2; ++pass) {
            auto it = g_pool_free.find(key);
            if (it != g_pool_free.end() && !it->second.empty()) {
                *p = it->second.back(); it->second.pop_back();
                if (kind == kDeviceMem && key.sz >= kLargeBlock) g_large_cached[key.dev] -= key.sz;
                if (pass == 0 && g_pending.size() > 64) reap_pending_locked(false);
                return hipSuccess;
            }
            if (pass == 0) { if (g_pending.empty()) break; reap_pending_locked(false); }
        }
    }
    flush_large_note();
    void* q = nullptr;
    auto alloc = [&] {
        if (kind == kDeviceMem) return hipMalloc(&q, key.sz);
        if (kind == kUncachedMem) return hipExtMallocWithFlags(&q, key.sz, hipDeviceMallocUncached);
        return hipHostMalloc(&q, key.sz, hipHostMallocDefault);
    };
    hipError_t e = alloc();
    if (e != hipSuccess) {              // out of memory: give the cache back and retry once
        (void)hipGetLastError();
        pool_trim();
        e = alloc();
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_key[q] = key;
    *p = q;
    return hipSuccess;
}
void pool_put_locked(void* p, int kind) {
    auto it = g_pool_key.find(p);
    if (it == g_pool_key.end()) { if (kind != kPinnedMem) (void)hipFree(p); else (void)hipHostFree(p); return; }
    if (kind == kDeviceMem && it->second.sz >= kLargeBlock) {
        if (g_large_cached[it->second.dev] + it->second.sz > large_limit_locked(it->second.dev)) { note_large_release(it->second.sz, it->second.dev); g_pool_key.erase(it); (void)hipFree(p); return; }
        g_large_cached[it->second.dev] += it->second.sz;
    }
    g_pool_free[it->second].push_back(p);
}
void pool_put(void* p, int kind) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(g_pool_mu); pool_put_locked(p, kind); }
    flush_large_note();
}
}  // namespace

// Stream-ordered free: the block goes back to the cache once the work enqueued on `s` so far has completed (an event is
// recorded now and polled when the allocator next runs) -- callers that would otherwise synchronise just to free a scratch
// buffer (the rocPRIM wrappers: ~25 us of idle GPU each time) use this.
void pool_free_on_many(void* const* ps, int n, hipStream_t s) {
    int live = 0;
    for (int i = 0; i < n; ++i) if (ps[i]) ++live;
    if (!live) return;
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_events.empty()) { ev = g_events.back(); g_events.pop_back(); }
    }
    auto fallback = [&]() { (void)hipStreamSynchronize(s); for (int i = 0; i < n; ++i) if (ps[i]) pool_free(ps[i]); };
    if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { fallback(); return; }
    if (hipEventRecord(ev, s) != hipSuccess) { fallback(); std::lock_guard<std::mutex> lk(g_pool_mu); g_events.push_back(ev); return; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (int i = 0; i < n; ++i) if (ps[i]) g_pending.push_back({ps[i], ev, --live == 0});
}
void pool_free_on(void* p, hipStream_t s) { pool_free_on_many(&p, 1, s); }

hipError_t pool_malloc(void** p, size_t bytes) { return pool_get(p, bytes, kDeviceMem); }
void pool_free(void* p) { pool_put(p, kDeviceMem); }
hipError_t pinned_malloc(void** p, size_t bytes) { return pool_get(p, bytes, kPinnedMem); }
void pinned_free(void* p) { pool_put(p, kPinnedMem); }
hipError_t uncached_malloc(void** p, size_t bytes) { return pool_get(p, bytes, kUncachedMem); }
void uncached_free(void* p) { pool_put(p, kUncachedMem); }

hipError_t stream_acquire(hipStream_t* s) {
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto& v = g_streams[cur_device()];
        if (!v.empty()) { *s = v.back(); v.pop_back(); return hipSuccess; }
    }
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
void stream_release(hipStream_t s) {          // the caller has synchronised it
    if (!s) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_streams[cur_device()].push_back(s);
}

void pool_trim() {
    struct Flush { ~Flush() { flush_large_note(); } } flush_after_unlock;      // (destroyed after the lock below)
    std::lock_guard<std::mutex> lk(g_pool_mu);
    reap_pending_locked(true);
    for (hipEvent_t ev : g_events) (void)hipEventDestroy(ev);
    g_events.clear();
    for (auto& kv : g_pool_free) {
        for (void* q : kv.second) {
            g_pool_key.erase(q);
            if (kv.first.kind != kPinnedMem) (void)hipFree(q); else (void)hipHostFree(q);
        }
        kv.second.clear();
    }
    g_large_cached.clear();
    for (auto& kv : g_streams) { for (hipStream_t s : kv.second) (void)hipStreamDestroy(s); kv.second.clear(); }
}
void pool_set_large_limit(long long bytes) { std::lock_guard<std::mutex> lk(g_pool_mu); g_large_limit_user = bytes; }

}  // namespace sfgpu

extern "C" {

int sfgpu_pool_trim(void) { sfgpu::pool_trim(); return SFGPU_OK; }
int sfgpu_pool_set_large_limit(long long bytes) { sfgpu::pool_set_large_limit(bytes); return SFGPU_OK; }


int sfgpu_version(void) { return SFGPU_VERSION; }
// 1: a build with -DSFGPU_VARIANTS (tools/*_variants.sh): the alternative kernel forms that lost their A/Bs and the tuning switches
// are compiled in; the product library answers 0
int sfgpu_has_variants(void) {
#ifdef SFGPU_VARIANTS
    return 1;
#else
    return 0;
#endif
}
const char* sfgpu_last_error(void) { return sfgpu::g_err; }
void sfgpu_set_logger(void (*log)(int level, const char* msg)) { sfgpu::g_logger = log; }

int sfgpu_device_info(char* name, int name_len, int* n_cu, uint64_t* hbm_bytes) {
    int dev = 0;
    SF_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    SF_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, (size_t)name_len - 1); name[name_len - 1] = 0; }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)p.totalGlobalMem;
    return SFGPU_OK;
}

}  // extern "C"
