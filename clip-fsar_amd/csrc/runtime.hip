// Error plumbing, version and the per-device launch-attribute cache of libclipfsar_hip.  The library holds no device memory
// and no stream; the only process-wide state is this cache of facts about the devices (CU count, the dynamic-LDS limit already
// raised for a kernel), keyed by device ordinal, so that one process may drive several GPUs.
#include <stdarg.h>

#include <atomic>

#include "common.h"

thread_local char cfsar_err_buf[512] = {0};

int cfsar_fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(cfsar_err_buf, sizeof(cfsar_err_buf), fmt, ap);
    va_end(ap);
    return 1;
}

int cfsar_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return cfsar_fail("%s: %s", what, hipGetErrorString(e));
    return 0;
}

namespace {
constexpr int kMaxDev = 64, kSlots = 256;
struct AttrSlot { std::atomic<const void*> fn; std::atomic<int> bytes; };
AttrSlot g_attr[kMaxDev][kSlots];
std::atomic<int> g_cus[kMaxDev];
}  // namespace

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: raise it once per (device, kernel).
int cfsar_ensure_lds(const void* fn, int bytes, const char* what) {
    if (bytes <= 48 * 1024) return 0;                                  // within the default limit
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return cfsar_fail("%s: bad current device", what);
    size_t h = (reinterpret_cast<size_t>(fn) >> 4) % kSlots;
    for (int probe = 0; probe < kSlots; ++probe, h = (h + 1) % kSlots) {
        AttrSlot& sl = g_attr[dev][h];
        const void* cur = sl.fn.load(std::memory_order_acquire);
        if (cur == nullptr) {
            const void* expect = nullptr;
            if (!sl.fn.compare_exchange_strong(expect, fn, std::memory_order_acq_rel) && expect != fn) continue;
            cur = fn;
        }
        if (cur != fn) continue;
        if (sl.bytes.load(std::memory_order_acquire) >= bytes) return 0;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return cfsar_fail("%s: set LDS size %d: %s", what, bytes, hipGetErrorString(e));
        sl.bytes.store(bytes, std::memory_order_release);              // racing threads set the same value: harmless
        return 0;
    }
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);   // table full: uncached
    return e == hipSuccess ? 0 : cfsar_fail("%s: set LDS size %d: %s", what, bytes, hipGetErrorString(e));
}

// compute units of the CURRENT device (persistent kernels launch one workgroup per CU)
int cfsar_num_cus() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int n = g_cus[dev].load(std::memory_order_relaxed);
    if (n <= 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        g_cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

extern "C" int cfsar_version(void) { return 400; /* 0.4.0 */ }
extern "C" int cfsar_abi_version(void) { return CFSAR_ABI_VERSION; }
extern "C" const char* cfsar_last_error(void) { return cfsar_err_buf; }
