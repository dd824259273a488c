// misc.hip -- library identity and device probing.
#include "common.h"

#include <cstring>

extern "C" const char* eae_hip_version(void) { return "eae_hip 1.0 (gfx950, f32 MFMA, bit-exact vs oracle/transforms_oracle.c)"; }

extern "C" int eae_hip_device_info(char* name, int name_cap, int* compute_units, int* clock_mhz, int64_t* hbm_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    hipDeviceProp_t p;
    e = hipGetDeviceProperties(&p, dev);
    if (e != hipSuccess) return (int)e;
    if (name && name_cap > 0) {
        std::strncpy(name, p.gcnArchName, (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (clock_mhz) *clock_mhz = p.clockRate / 1000;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return EAE_HIP_OK;
}
