// devstate.h — launch-time state that belongs to a DEVICE, not to the process.
//
// include/esahrnet.h promises one handle per device and nothing shared between handles; the reference's own
// multi-GPU wrapper is single-process nn.DataParallel (val.py:382, main.py:254: one replica and one Python thread
// per GPU).  Two things the launchers need are per-device facts: the raised dynamic-LDS limit of a kernel
// (hipFuncSetAttribute acts on the current device's copy of the function) and the CU count that sizes persistent
// grids.  Both are cached here per device ordinal, behind one mutex, so that a second device gets its own
// attribute calls and two replica threads cannot race on a `static bool`.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <utility>

namespace esa {

struct DevState {
    std::mutex mu;
    std::map<std::pair<const void*, int>, int> lds;     // (kernel, device) -> bytes already granted
    std::map<int, int> cus;                             // device -> multiProcessorCount
};
inline DevState& dev_state() {
    static DevState s;
    return s;
}

// ordinal of the calling thread's current device (-1 if the runtime has none)
inline int current_device() {
    int dev = -1;
    return hipGetDevice(&dev) == hipSuccess ? dev : -1;
}

// Raise the dynamic shared memory limit of `kern` to `bytes` on the current device, once per (kernel, device).
// Returns hipError_t as int.
inline int ensure_dyn_lds(const void* kern, int bytes) {
    const int dev = current_device();
    DevState& s = dev_state();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.lds.find({kern, dev});
    if (it != s.lds.end() && it->second >= bytes) return 0;
    const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    s.lds[{kern, dev}] = bytes;
    return 0;
}

// CU count of the current device (256 on MI355X; the fallback if the query fails)
inline int device_cus() {
    const int dev = current_device();
    DevState& s = dev_state();
    std::lock_guard<std::mutex> lk(s.mu);
    auto it = s.cus.find(dev);
    if (it != s.cus.end()) return it->second;
    int cus = 256;
    hipDeviceProp_t prop;
    if (dev >= 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    s.cus[dev] = cus;
    return cus;
}

// test hook (esahrnet_debug_devstate): number of (kernel, device) attribute entries and of devices seen
inline void dev_state_counts(int* kernels, int* devices) {
    DevState& s = dev_state();
    std::lock_guard<std::mutex> lk(s.mu);
    if (kernels) *kernels = (int)s.lds.size();
    if (devices) *devices = (int)s.cus.size();
}

}  // namespace esa
