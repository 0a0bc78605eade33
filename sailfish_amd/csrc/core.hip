// core.hip -- error slot, logger hook, device info for libsfgpu.
#include "common.h"

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

}  // namespace sfgpu

extern "C" {

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
