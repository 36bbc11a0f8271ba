// The library's built-in workspace, for the entry point that must not allocate
// (rt_render_device_static): tile queue + one deep-stack region per wave of the largest
// launch grid.  It is a file-scope device array in a code object of its own, which holds
// nothing else: the HIP runtime loads a code object when something in it is first asked for,
// so a process that never calls that entry point never pays for the array (52.5 MiB of device
// memory per GPU and about 18 ms at load), and one that does pays at its first such call on a
// device, not in rt_create.  Each GPU holds its own copy at its own address until the process
// ends.  Host code never uses the array's name as an address: static_workspace() asks the
// runtime for the device address, once per device.
#include <mutex>

#include "whitted_device.h"

namespace rt {

// (no device code in this file names the array: `used` keeps it in the code object)
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kWorkspaceMaxBytes];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[kMaxDevices];
}  // namespace

// The array's address on `device`, which must be the calling thread's current device.
hipError_t static_workspace(int device, void** ptr) {
    *ptr = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));  // of the current device
        if (e != hipSuccess) return e;
        g_scratch_ptr[device] = p;
    }
    *ptr = g_scratch_ptr[device];
    return hipSuccess;
}

}  // namespace rt
