// device_ctx.hip -- the process-wide device context behind include/nbody_hip.h.
//
// Stands where the reference has src/lib/vulkan_ctx.c (device pick behind a `static bool`, vulkan_ctx.c:11,187-191;
// the reference always takes device 0, vulkan_ctx.c:83-84).  One process drives one GPU; ranks of a sharded run pick
// theirs with nb_hip_set_device(LOCAL_RANK) before anything touches the device.  Nothing here runs until a pipeline
// first needs the GPU (SetSimulationData), so CPU-only worlds never initialise HIP.
#include <chrono>

#include "pipeline_internal.h"

#define NB_HIP_VERSION 301  // 0.3.1: + nb_hip_preflight_* / nb_hip_comm_bringup (multi-GPU bring-up diagnostics that report instead of aborting)

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

// ---- preflight: what a harness asks BEFORE the first real contact between ranks; every probe reports, none aborts --------

int nb_hip_preflight_peers(int *row, int len) {
    nbi::RandGuard keep_callers_rand_stream;  // may be this process' first touch of the GPU
    nbi::use_device();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) count = 0;
    for (int q = 0; row && q < len; q++) {
        row[q] = -2;  // no such device
        if (q >= count) continue;
        int can = 0;
        if (q == nbi::g_dev.ordinal)
            row[q] = 1;
        else if (hipDeviceCanAccessPeer(&can, nbi::g_dev.ordinal, q) == hipSuccess)
            row[q] = can ? 1 : 0;
        else {
            (void)hipGetLastError();
            row[q] = -1;
        }
    }
    return count;
}

namespace {
uint32_t *g_preflight_word = nullptr;  // the 4-byte device word whose IPC handle the probe exports
}

int nb_hip_preflight_ipc_export(void *handle64, uint32_t tag) {
    nbi::RandGuard keep_callers_rand_stream;
    nbi::use_device();
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the preflight ABI carries IPC handles as 64 bytes");
    hipError_t e = hipSuccess;
    if (g_preflight_word == nullptr) e = hipMalloc(reinterpret_cast<void **>(&g_preflight_word), 4096);
    if (e == hipSuccess) e = hipMemcpy(g_preflight_word, &tag, sizeof tag, hipMemcpyHostToDevice);
    hipIpcMemHandle_t h;
    memset(&h, 0, sizeof h);
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, g_preflight_word);
    if (e != hipSuccess) (void)hipGetLastError();
    if (handle64) memcpy(handle64, &h, sizeof h);
    return (int)e;
}

int nb_hip_preflight_ipc_open(const void *handle64, uint32_t expect_tag, double *ms) {
    nbi::RandGuard keep_callers_rand_stream;
    nbi::use_device();
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof h);
    const auto t0 = std::chrono::steady_clock::now();
    void *mapped = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&mapped, h, hipIpcMemLazyEnablePeerAccess);
    uint32_t seen = ~expect_tag;
    if (e == hipSuccess) e = hipMemcpy(&seen, mapped, sizeof seen, hipMemcpyDeviceToHost);
    if (mapped) {
        const hipError_t c = hipIpcCloseMemHandle(mapped);
        if (e == hipSuccess) e = c;
    }
    if (ms) *ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return (int)e;
    }
    return seen == expect_tag ? 0 : -1;  // -1: mapped and read, but not the peer's word
}

void nb_hip_preflight_ipc_release(void) {
    if (g_preflight_word) (void)hipFree(g_preflight_word);
    g_preflight_word = nullptr;
}

const char *nb_hip_error_string(int hip_error) { return hip_error == -1 ? "mapped, but the word read is not the peer's" : hipGetErrorString((hipError_t)hip_error); }

int nb_hip_runtime_version(void) {
    int v = 0;
    if (hipRuntimeGetVersion(&v) != hipSuccess) return 0;
    return v;
}

}  // extern "C"
