// mrca_hostutil.h -- host-side helpers shared by the translation units of libmrca_env.so (mrca_abi.hip,
// mrca_policy.hip, mrca_policy_bwd.hip): the error string behind mrca_last_error(), and running a launch on the device
// its buffers live on whatever the caller's current device is.
#pragma once
#include <hip/hip_runtime.h>

namespace mrca {

// Records the message mrca_last_error() returns (thread-local) and hands `code` back.  Defined in mrca_abi.hip.
int set_error(int code, const char* fmt, ...);

// Launches must target the device of the buffers whatever the caller's current device is (two envs on two GPUs in one
// process; a torch caller whose current device differs): switch for the duration of the call, then restore.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int want) {
        if (want >= 0 && hipGetDevice(&prev) == hipSuccess && prev != want) switched = hipSetDevice(want) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// The device a device pointer belongs to (-1 if the runtime does not know the pointer: the caller's current device is
// used then, as before).
inline int device_of(const void* dev_ptr) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, dev_ptr) != hipSuccess) {
        (void)hipGetLastError();      // clear the sticky error of a failed query
        return -1;
    }
    return a.device;
}

}  // namespace mrca
