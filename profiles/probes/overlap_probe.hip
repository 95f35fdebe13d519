// Micro-probe (round 4): which launch forms let two INDEPENDENT small kernels run at the same time on this stack?
//   (1) eager stream order, the second launch with hipExtAnyOrderLaunch (AQL barrier bit cleared)
//   (2) the same pair recorded by stream capture and replayed as a hipGraph
//   (3) an explicit graph whose two kernel nodes have no edge between them (fork / join by the runtime)
//   (4) two captured chains replayed on two streams
// Every kernel stamps s_memrealtime (100 MHz) at its start and end; overlap = second.begin < first.end.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void spin_k(unsigned long long* ts, int slot, int ticks, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long t = t0;
    float acc = 0.f;
    while ((long long)(t - t0) < ticks) { acc += 1.f; t = __builtin_amdgcn_s_memrealtime(); }
    if (threadIdx.x == 0) {
        atomicMin(&ts[2 * slot], t0);
        atomicMax(&ts[2 * slot + 1], t);
        if (acc < 0.f) sink[0] = acc;
    }
}

static unsigned long long* d_ts; static float* d_sink;
static const int SLOTS = 256;
static void reset_ts() {
    std::vector<unsigned long long> h(2 * SLOTS);
    for (int i = 0; i < SLOTS; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
    hipMemcpy(d_ts, h.data(), h.size() * 8, hipMemcpyHostToDevice);
}
static std::vector<unsigned long long> read_ts() {
    std::vector<unsigned long long> h(2 * SLOTS);
    hipMemcpy(h.data(), d_ts, h.size() * 8, hipMemcpyDeviceToHost);
    return h;
}
static void launch(hipStream_t s, int slot, int us, int grid, int flags) {
    int ticks = us * 100;
    void* args[] = {&d_ts, &slot, &ticks, &d_sink};
    hipExtLaunchKernel((const void*)spin_k, dim3(grid), dim3(256), args, 0, s, nullptr, nullptr, flags);
}
static void report(const char* name, int n) {
    auto h = read_ts();
    unsigned long long base = ~0ull;
    for (int i = 0; i < n; ++i) base = std::min(base, h[2 * i]);
    printf("%s\n", name);
    for (int i = 0; i < n; ++i)
        printf("   kernel %2d: begin %8.2f us  end %8.2f us\n", i, (h[2 * i] - base) / 100.0, (h[2 * i + 1] - base) / 100.0);
}

int main() {
    hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
    CK(hipMalloc(&d_ts, 2 * SLOTS * 8)); CK(hipMalloc(&d_sink, 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    // warm
    launch(s, 0, 5, 16, 0); hipStreamSynchronize(s);

    // (1) eager: A 30us, B 10us any-order, C 5us
    reset_ts();
    launch(s, 0, 30, 16, 0); launch(s, 1, 10, 16, hipExtAnyOrderLaunch); launch(s, 2, 5, 16, 0);
    CK(hipStreamSynchronize(s));
    report("(1) eager: A(30us) ; B(10us, any-order) ; C(5us)", 3);

    // (2) captured
    {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        launch(s, 0, 30, 16, 0); launch(s, 1, 10, 16, hipExtAnyOrderLaunch); launch(s, 2, 5, 16, 0);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) { reset_ts(); CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s)); }
        report("(2) stream-captured graph of the same three launches (third replay)", 3);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    // (3) explicit graph: A, B roots; C after both.  Then a chain of 20 such (host 20us, rider 10us) pairs against the serial chain
    {
        auto build = [&](int pairs, bool parallel, int host_us, int rider_us, hipGraphExec_t* ge) {
            hipGraph_t g; hipGraphCreate(&g, 0);
            std::vector<hipGraphNode_t> prev;
            static int slots[512]; static int ticks[512];
            static void* argv[512][4];
            int q = 0;
            for (int p = 0; p < pairs; ++p) {
                hipGraphNode_t a, b;
                for (int w = 0; w < 2; ++w) {
                    slots[q] = q < SLOTS ? q : SLOTS - 1; ticks[q] = (w == 0 ? host_us : rider_us) * 100;
                    argv[q][0] = &d_ts; argv[q][1] = &slots[q]; argv[q][2] = &ticks[q]; argv[q][3] = &d_sink;
                    hipKernelNodeParams kp{}; kp.func = (void*)spin_k; kp.gridDim = dim3(16); kp.blockDim = dim3(256);
                    kp.kernelParams = argv[q]; kp.sharedMemBytes = 0; kp.extra = nullptr;
                    std::vector<hipGraphNode_t> deps = prev;
                    if (w == 1 && !parallel) deps = {a};
                    hipGraphAddKernelNode(w == 0 ? &a : &b, g, deps.data(), deps.size(), &kp);
                    ++q;
                }
                prev = parallel ? std::vector<hipGraphNode_t>{a, b} : std::vector<hipGraphNode_t>{b};
            }
            hipGraphInstantiate(ge, g, nullptr, nullptr, 0);
        };
        for (int parallel = 0; parallel < 2; ++parallel) {
            hipGraphExec_t ge; build(20, parallel, 20, 10, &ge);
            for (int r = 0; r < 3; ++r) { reset_ts(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
            hipEventRecord(e0, s);
            for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
            hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            printf("(3) explicit graph, 20 x (host 20us, rider 10us), %s: %.1f us per replay (serial ideal 600 + nodes, parallel ideal 400)\n",
                   parallel ? "rider has no edge to its host" : "serial chain", ms * 1e3 / 20);
            if (parallel) report("    first pairs of the last replay", 6);
            hipGraphExecDestroy(ge);
        }
    }
    // (2b) captured chain of 20 (host normal, rider any-order) pairs vs all-normal
    for (int any = 0; any < 2; ++any) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < 20; ++p) { launch(s, 2 * p, 20, 16, 0); launch(s, 2 * p + 1, 10, 16, any ? hipExtAnyOrderLaunch : 0); }
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) { hipGraphLaunch(ge, s); hipStreamSynchronize(s); }
        hipEventRecord(e0, s);
        for (int r = 0; r < 20; ++r) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("(2b) captured chain of 20 pairs, riders %s: %.1f us per replay\n", any ? "any-order" : "in order", ms * 1e3 / 20);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    // (1b) eager chain of 20 pairs
    for (int any = 0; any < 2; ++any) {
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < 10; ++r)
            for (int p = 0; p < 20; ++p) { launch(s, 2 * p, 20, 16, 0); launch(s, 2 * p + 1, 10, 16, any ? hipExtAnyOrderLaunch : 0); }
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("(1b) eager chain of 20 pairs, riders %s: %.1f us per 20 pairs\n", any ? "any-order" : "in order", ms * 1e3 / 10);
    }
    // (4) two captured chains on two streams
    {
        hipGraph_t g1, g2; hipGraphExec_t ge1, ge2;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < 20; ++p) launch(s, p, 20, 16, 0);
        CK(hipStreamEndCapture(s, &g1)); CK(hipGraphInstantiate(&ge1, g1, nullptr, nullptr, 0));
        CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < 20; ++p) launch(s2, 32 + p, 20, 16, 0);
        CK(hipStreamEndCapture(s2, &g2)); CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        for (int r = 0; r < 2; ++r) { hipGraphLaunch(ge1, s); hipGraphLaunch(ge2, s2); hipDeviceSynchronize(); }
        hipDeviceSynchronize();
        hipEventRecord(e0, s);
        for (int r = 0; r < 10; ++r) hipGraphLaunch(ge1, s);
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("(4) one chain of 20 x 20us alone: %.1f us per replay\n", ms * 1e3 / 10);
        reset_ts();
        hipDeviceSynchronize();
        hipEventRecord(e0, s);
        for (int r = 0; r < 10; ++r) { hipGraphLaunch(ge1, s); hipGraphLaunch(ge2, s2); }
        hipDeviceSynchronize();
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("(4) the two chains on two streams, 10 replays each: %.1f us per pair of replays\n", ms * 1e3 / 10);
        auto h = read_ts();
        printf("    last replay: chain 1 kernel 0 [%.1f, %.1f], chain 2 kernel 0 [%.1f, %.1f] (us, relative)\n", 0.0,
               (h[1] - h[0]) / 100.0, ((long long)h[64] - (long long)h[0]) / 100.0, ((long long)h[65] - (long long)h[0]) / 100.0);
        // eager two streams
        hipDeviceSynchronize();
        hipEventRecord(e0, s);
        for (int r = 0; r < 10; ++r) for (int p = 0; p < 20; ++p) { launch(s, p, 20, 16, 0); launch(s2, 32 + p, 20, 16, 0); }
        hipDeviceSynchronize();
        hipEventRecord(e1, s); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("(4b) eager, two streams, 20 x 20us each: %.1f us per 20+20 launches (serial 800, overlapped 400)\n", ms * 1e3 / 10);
    }
    return 0;
}
