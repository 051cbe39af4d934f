// launch_latency.hip -- what a per-frame call is made of on this box: one tiny dispatch + a stream synchronisation, a chain of k tiny
// dispatches, small pinned copies in either direction, and a kernel that leaves its result in host-mapped memory instead of a D2H copy.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/launch_latency tools/microbench/launch_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_tiny(float *p, int v) { if (threadIdx.x == 0) p[0] = (float)v; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == 's') { CHECK(hipSetDeviceFlags(hipDeviceScheduleSpin)); printf("hipDeviceScheduleSpin\n"); }
    if (argc > 1 && argv[1][0] == 'y') { CHECK(hipSetDeviceFlags(hipDeviceScheduleYield)); printf("hipDeviceScheduleYield\n"); }
    float *dev, *host, *mapped;
    CHECK(hipMalloc(&dev, 4096)); CHECK(hipHostMalloc(&host, 4096)); CHECK(hipHostMalloc(&mapped, 4096, hipHostMallocMapped));
    hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int reps = 2000;
    auto run = [&](const char *name, auto body) {
        for (int i = 0; i < 50; ++i) body(i);
        const double t0 = now();
        for (int i = 0; i < reps; ++i) body(i);
        printf("%-72s %7.2f us per call\n", name, (now() - t0) / reps);
        return 0;
    };
    run("1 dispatch + sync", [&](int i) { k_tiny<<<1, 64, 0, s>>>(dev, i); (void)hipStreamSynchronize(s); });
    run("2 dispatches + sync", [&](int i) { k_tiny<<<1, 64, 0, s>>>(dev, i); k_tiny<<<1, 64, 0, s>>>(dev, i); (void)hipStreamSynchronize(s); });
    run("3 dispatches + sync", [&](int i) { for (int k = 0; k < 3; ++k) k_tiny<<<1, 64, 0, s>>>(dev, i); (void)hipStreamSynchronize(s); });
    run("1 dispatch writing host-mapped memory + sync", [&](int i) { k_tiny<<<1, 64, 0, s>>>(mapped, i); (void)hipStreamSynchronize(s); });
    run("H2D 256 B + sync", [&](int i) { host[0] = (float)i; (void)hipMemcpyAsync(dev, host, 256, hipMemcpyHostToDevice, s); (void)hipStreamSynchronize(s); });
    run("D2H 256 B + sync", [&](int i) { (void)hipMemcpyAsync(host, dev, 256, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
    run("H2D + 2 dispatches + D2H + sync (a per-frame call today)", [&](int i) {
        host[0] = (float)i; (void)hipMemcpyAsync(dev, host, 256, hipMemcpyHostToDevice, s);
        k_tiny<<<1, 64, 0, s>>>(dev + 64, i); k_tiny<<<1, 64, 0, s>>>(dev + 64, i);
        (void)hipMemcpyAsync(host, dev, 256, hipMemcpyDeviceToHost, s); (void)hipStreamSynchronize(s); });
    run("1 dispatch + spin on host-mapped word (no sync call)", [&](int i) {
        volatile float *w = mapped; k_tiny<<<1, 64, 0, s>>>(mapped, i + 1); while (*w != (float)(i + 1)) { } });
    {
        hipEvent_t ev; CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        run("1 dispatch + event record + spin on hipEventQuery", [&](int i) { k_tiny<<<1, 64, 0, s>>>(dev, i); (void)hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) { } });
        run("1 dispatch + spin on hipStreamQuery", [&](int i) { k_tiny<<<1, 64, 0, s>>>(dev, i); while (hipStreamQuery(s) == hipErrorNotReady) { } });
        uint32_t *sig = nullptr;
        if (hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory) == hipSuccess) {
            *sig = 0;
            run("1 dispatch + hipStreamWriteValue32 + spin on the word", [&](int i) { volatile uint32_t *w = sig; k_tiny<<<1, 64, 0, s>>>(dev, i); (void)hipStreamWriteValue32(s, sig, (uint32_t)(i + 7), 0); while (*w != (uint32_t)(i + 7)) { } });
        } else printf("no signal memory\n");
    }
    (void)hipStreamSynchronize(s);
    return 0;
}
