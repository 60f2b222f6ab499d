// launch_chain.hip -- what a chain of DEPENDENT kernel launches costs on this GPU, as stream launches and as one hipGraph launch:
// the shape of a frame rendered one at a time (nine dependent stage launches, profiles/r05_notes.md section 13).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/launch_chain.hip -o tools/microbench/launch_chain && tools/microbench/launch_chain
// Each kernel spins for `busy_us` on every CU-filling block (a stage with work) or returns at once (busy_us = 0: the pure launch path).
// Reported per chain of N launches: GPU time from the first kernel's start to the last one's end (events on the stream), and host time
// to submit the chain.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            std::exit(1);                                                                         \
        }                                                                                         \
    } while (0)

__global__ void spin(long long ticks, unsigned *sink) { // wall_clock64: 100 MHz
    const long long t0 = wall_clock64();
    unsigned v = threadIdx.x;
    while (wall_clock64() - t0 < ticks) v = v * 1664525u + 1013904223u;
    if (v == 0xFFFFFFFFu && sink) *sink = v;
}

int main(int argc, char **argv) {
    const int n_chain = argc > 1 ? std::atoi(argv[1]) : 9, reps = 200;
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned *sink = nullptr;
    CK(hipMalloc(&sink, 4));
    std::printf("%s, %d CUs; chains of %d dependent launches, %d repetitions\n", prop.name, prop.multiProcessorCount, n_chain, reps);
    for (int busy_us : {0, 20, 100}) {
        for (int blocks : {1, prop.multiProcessorCount * 4}) {
            const long long ticks = (long long)busy_us * 100;
            auto chain = [&] {
                for (int k = 0; k < n_chain; ++k) hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, st, ticks, sink);
            };
            for (int w = 0; w < 5; ++w) chain();
            CK(hipStreamSynchronize(st));
            // ---- stream launches
            double gpu_ms = 0.0, host_us = 0.0;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0, st));
                const auto h0 = std::chrono::steady_clock::now();
                chain();
                host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                gpu_ms += ms;
            }
            const double s_gpu = gpu_ms / reps * 1e3, s_host = host_us / reps;
            // ---- the same chain captured once, launched as a graph
            hipGraph_t graph;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            chain();
            CK(hipStreamEndCapture(st, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(exec, st));
            CK(hipStreamSynchronize(st));
            gpu_ms = 0.0;
            host_us = 0.0;
            for (int r = 0; r < reps; ++r) {
                CK(hipEventRecord(e0, st));
                const auto h0 = std::chrono::steady_clock::now();
                CK(hipGraphLaunch(exec, st));
                host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms = 0.f;
                CK(hipEventElapsedTime(&ms, e0, e1));
                gpu_ms += ms;
            }
            const double g_gpu = gpu_ms / reps * 1e3, g_host = host_us / reps;
            CK(hipGraphExecDestroy(exec));
            CK(hipGraphDestroy(graph));
            const double work = double(busy_us) * n_chain;
            std::printf("busy %3d us x %4d blocks: stream launches %7.1f us GPU (%5.1f us per launch beyond the work), %6.1f us host | graph %7.1f us GPU (%5.1f), %6.1f us host\n",
                        busy_us, blocks, s_gpu, (s_gpu - work) / n_chain, s_host, g_gpu, (g_gpu - work) / n_chain, g_host);
        }
    }
    return 0;
}
