// device_ctx.hip -- the process-wide device context behind include/nbody_hip.h.
//
// Stands where the reference has src/lib/vulkan_ctx.c (device pick behind a `static bool`, vulkan_ctx.c:11,187-191;
// the reference always takes device 0, vulkan_ctx.c:83-84).  One process drives one GPU; ranks of a sharded run pick
// theirs with nb_hip_set_device(LOCAL_RANK) before anything touches the device.  Nothing here runs until a pipeline
// first needs the GPU (SetSimulationData), so CPU-only worlds never initialise HIP.
#include "pipeline_internal.h"

#define NB_HIP_VERSION 300  // 0.3.0: launch-shape / experiment knobs left the ABI (nbody_hip_tuning.h); + nb_hip_probe_clock, nb_hip_clock_sampler_*

namespace nbi {

DeviceCtx g_dev;
int g_requested_ordinal = -1;

void ensure_device() {
    if (g_dev.ready) return;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    NB_ASSERT(e == hipSuccess && count > 0,
              "no HIP device visible (hipGetDeviceCount: %s, count %d); the GPU path has no CPU fallback",
              hipGetErrorString(e), count);
    int ord = g_requested_ordinal >= 0 ? g_requested_ordinal : 0;
    NB_ASSERT(ord < count, "device ordinal %d requested, %d visible", ord, count);
    ASSERT_HIP(hipSetDevice(ord), "hipSetDevice(%d)", ord);
    // PerformSimUpdate is synchronous by contract (the reference blocks on its fence, sim_gpu.c:353): how fast the host
    // notices completion is part of a short step.  NB_HIP_WAIT=spin|yield|block picks the runtime's wait policy before
    // the context exists; default: the runtime's.
#ifdef NB_TUNING_SHAPES
    if (const char *wp = getenv("NB_HIP_WAIT")) {
        const unsigned flag = !strcmp(wp, "spin") ? hipDeviceScheduleSpin
                              : !strcmp(wp, "yield") ? hipDeviceScheduleYield
                              : !strcmp(wp, "block") ? hipDeviceScheduleBlockingSync
                                                     : hipDeviceScheduleAuto;
        if (hipSetDeviceFlags(flag) != hipSuccess) (void)hipGetLastError();  // context already live: keep its policy
    }
#endif
    hipDeviceProp_t prop;
    ASSERT_HIP(hipGetDeviceProperties(&prop, ord), "hipGetDeviceProperties(%d)", ord);
    NB_ASSERT(strncmp(prop.gcnArchName, "gfx950", 6) == 0,
              "device %d is %s; this library ships gfx950 (MI355X) code objects only", ord, prop.gcnArchName);
    g_dev.ordinal = ord;
    g_dev.compute_units = prop.multiProcessorCount;
    // the PCI address is what tells two ranks' devices apart when each process sees its own GPU as ordinal 0
    snprintf(g_dev.info, sizeof g_dev.info, "%s %s %d %d pci=%04x:%02x:%02x", prop.name[0] ? prop.name : "AMD-GPU", prop.gcnArchName,
             prop.multiProcessorCount, prop.clockRate / 1000, prop.pciDomainID, prop.pciBusID, prop.pciDeviceID);
    g_dev.ready = true;
}

// HIP's current device is per THREAD: every entry point that allocates, launches or copies re-selects the
// process' device, so a call from another thread than the first one lands on the same GPU (ordinal > 0 matters:
// sharded ranks use LOCAL_RANK).
void use_device() {
    ensure_device();
    ASSERT_HIP(hipSetDevice(g_dev.ordinal), "hipSetDevice(%d)", g_dev.ordinal);
}

void *dev_alloc_bytes(size_t bytes) {
    void *p = nullptr;
    ASSERT_HIP(hipMalloc(&p, bytes ? bytes : 1), "hipMalloc of %zu bytes", bytes);
    return p;
}

void dev_free(void *p) {
    if (p) ASSERT_HIP(hipFree(p), "hipFree");
}

}  // namespace nbi

extern "C" {

int nb_hip_version(void) { return NB_HIP_VERSION; }

int nb_hip_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

void nb_hip_set_device(int ordinal) {
    NB_ASSERT(!nbi::g_dev.ready || nbi::g_dev.ordinal == ordinal, "device %d already in use, cannot switch to %d",
              nbi::g_dev.ordinal, ordinal);
    nbi::g_requested_ordinal = ordinal;
}

void nb_hip_device_info(char *buf, uint32_t len) {
    nbi::ensure_device();
    if (buf && len) snprintf(buf, len, "%s", nbi::g_dev.info);
}

int nb_hip_runtime_version(void) {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) return 0;
    return v;
}

}  // extern "C"
