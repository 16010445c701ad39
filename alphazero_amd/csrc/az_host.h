// Host-side helpers shared by the translation units of libazhip.so: the error text behind
// az_last_error() and the per-device "large dynamic LDS" attribute of the big kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>

// records `what` (plus the HIP error text, if any) as this thread's az_last_error() and returns -1
int az_fail(const char* what, hipError_t e = hipSuccess);

#define AZ_HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return az_fail(#x, _e); } while (0)
#define AZ_LAUNCH_CHECK(name) \
    do { hipError_t _e = hipGetLastError(); if (_e != hipSuccess) return az_fail("launch of " name, _e); } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device attribute of a kernel: set it the first time the
// kernel is launched on each device.  `done` is the kernel's own bitmap (bit d = device d), updated atomically
// so that two host threads driving different engines cannot lose each other's bit.
inline int az_set_kernel_lds(const void* kernel, int bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    AZ_HIPCHK(hipGetDevice(&dev));
    if (dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return 0;
    AZ_HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return 0;
}
