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


// ---- caching device allocator (see common.h) ----
namespace {
std::mutex g_pool_mu;
std::unordered_map<size_t, std::vector<void*>> g_pool_free;    // rounded size -> free blocks
std::unordered_map<void*, size_t> g_pool_size;                  // live or cached block -> rounded size
size_t round_up_pow2(size_t n) { size_t r = 256; while (r < n) r <<= 1; return r; }
}  // namespace

hipError_t pool_malloc(void** p, size_t bytes) {
    const size_t sz = round_up_pow2(bytes ? bytes : 1);
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_free.find(sz);
        if (it != g_pool_free.end() && !it->second.empty()) { *p = it->second.back(); it->second.pop_back(); return hipSuccess; }
    }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, sz);
    if (e != hipSuccess) {              // out of memory: give the cache back and retry once
        pool_trim();
        e = hipMalloc(&q, sz);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_size[q] = sz;
    *p = q;
    return hipSuccess;
}

void pool_free(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_size.find(p);
    if (it == g_pool_size.end()) { (void)hipFree(p); return; }
    g_pool_free[it->second].push_back(p);
}

void pool_trim() {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (auto& kv : g_pool_free) {
        for (void* q : kv.second) { g_pool_size.erase(q); (void)hipFree(q); }
        kv.second.clear();
    }
}

}  // namespace sfgpu

extern "C" {

int sfgpu_pool_trim(void) { sfgpu::pool_trim(); return SFGPU_OK; }


int sfgpu_version(void) { return SFGPU_VERSION; }
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
