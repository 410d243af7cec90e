// ubench6: what a dependent kernel launch costs on this box -- the per-step floor of a step chain.
// Chains of 200 dependent launches of (a) an empty kernel, (b) a kernel that reads and writes 16 KB through HBM
// pointers (a stand-in for "load receivers, store results"), as plain stream launches and as a hipGraph replay.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void empty_kernel(float *) {}
__global__ void touch_kernel(float *p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.0f; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    float *buf;
    CHECK(hipMalloc(&buf, 1 << 20));
    CHECK(hipMemset(buf, 0, 1 << 20));
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int N = 200;
    struct Case { const char *name; void (*fn)(float *); dim3 grid, block; } cases[] = {
        {"empty 1x64", empty_kernel, dim3(1), dim3(64)},
        {"empty 64x1024", empty_kernel, dim3(64), dim3(1024)},
        {"touch 4x1024 (16 KB rw)", touch_kernel, dim3(4), dim3(1024)},
        {"touch 64x1024 (256 KB rw)", touch_kernel, dim3(64), dim3(1024)},
    };
    for (auto &c : cases) {
        void *args[] = {&buf};
        // plain launches
        double best_plain = 1e30, best_graph = 1e30;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipStreamSynchronize(st));
            const double t0 = now_us();
            for (int i = 0; i < N; i++) CHECK(hipLaunchKernel((const void *)c.fn, c.grid, c.block, args, 0, st));
            CHECK(hipStreamSynchronize(st));
            const double t = (now_us() - t0) / N;
            if (t < best_plain) best_plain = t;
        }
        hipGraph_t g;
        CHECK(hipGraphCreate(&g, 0));
        hipGraphNode_t prev = nullptr;
        for (int i = 0; i < N; i++) {
            hipKernelNodeParams kp = {};
            kp.func = (void *)c.fn; kp.gridDim = c.grid; kp.blockDim = c.block; kp.kernelParams = args;
            hipGraphNode_t node;
            CHECK(hipGraphAddKernelNode(&node, g, prev ? &prev : nullptr, prev ? 1 : 0, &kp));
            prev = node;
        }
        hipGraphExec_t ge;
        const double tb = now_us();
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        const double build = now_us() - tb;
        for (int rep = 0; rep < 5; rep++) {
            CHECK(hipStreamSynchronize(st));
            const double t0 = now_us();
            CHECK(hipGraphLaunch(ge, st));
            CHECK(hipStreamSynchronize(st));
            const double t = (now_us() - t0) / N;
            if (t < best_graph) best_graph = t;
        }
        printf("%-28s plain %6.2f us/launch   graph replay %6.2f us/launch   (instantiate %7.1f us for %d nodes)\n", c.name, best_plain, best_graph, build, N);
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
    }
    return 0;
}
