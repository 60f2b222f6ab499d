// Microbenchmark: dependent random gathers of 64-byte records (4 x dwordx4 per lane), the
// memory access pattern of a BVH2 node visit. Reports ns per step and lane-fetches/s for
// several working-set sizes and occupancies. Build: hipcc --offload-arch=gfx950 -O3 gather64.hip -o gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>

template <int NLOADS>
__global__ __launch_bounds__(256) void chase(const float4 *__restrict__ nodes, uint32_t mask, int steps, uint32_t *out, int coherent, int active_lanes) {
    uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cur = coherent ? ((tid >> 6) * 2654435761u) & mask : (tid * 2654435761u) & mask;
    float acc = 0.f;
    if ((int)(threadIdx.x & 63u) >= active_lanes) steps = 0;
    for (int s = 0; s < steps; ++s) {
        const float4 *np = nodes + (size_t)cur * NLOADS; // record stride = record size
        float4 a = np[0];
        float4 b = NLOADS > 1 ? np[1] : a;
        float4 c = NLOADS > 2 ? np[2] : a;
        float4 d = NLOADS > 3 ? np[3] : a;
        acc += a.x + b.y + c.z;
        uint32_t nxt = __float_as_uint(NLOADS > 3 ? d.w : (NLOADS > 1 ? b.w : a.w));
        cur = (nxt + (coherent ? 0u : (tid & 63u) * 40503u)) & mask;
    }
    out[tid] = cur + (uint32_t)acc;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int cus = prop.multiProcessorCount;
    const size_t max_nodes = 1u << 22; // 256 MiB
    std::vector<float4> h(max_nodes * 4);
    uint32_t x = 12345;
    for (size_t i = 0; i < max_nodes; ++i) {
        for (int k = 0; k < 4; ++k) h[i * 4 + k] = make_float4(1.f, 2.f, 3.f, 0.f);
        x = x * 1664525u + 1013904223u;
        uint32_t nxt = x >> 4;
        for (int k = 0; k < 4; ++k) memcpy(&h[i * 4 + k].w, &nxt, 4);
    }
    float4 *d; hipMalloc(&d, max_nodes * 64); hipMemcpy(d, h.data(), max_nodes * 64, hipMemcpyHostToDevice);
    uint32_t *out; hipMalloc(&out, (size_t)cus * 8 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 2000;
    printf("CUs %d\n", cus);
    for (int coherent = 0; coherent < 1; ++coherent)
    for (int nl : {4, 2})
    for (int active_lanes : {64, 26})
    for (size_t ws_nodes : {size_t(1) << 12, size_t(1) << 19, size_t(1) << 21}) {
        for (int bpc : {1, 5, 8}) {
            int grid = cus * bpc;
            auto launch = [&]() {
                if (nl == 4) hipLaunchKernelGGL(chase<4>, dim3(grid), dim3(256), 0, 0, d, (uint32_t)(ws_nodes - 1), steps, out, coherent, active_lanes);
                else if (nl == 2) hipLaunchKernelGGL(chase<2>, dim3(grid), dim3(256), 0, 0, d, (uint32_t)(ws_nodes - 1), steps, out, coherent, active_lanes);
                else hipLaunchKernelGGL(chase<1>, dim3(grid), dim3(256), 0, 0, d, (uint32_t)(ws_nodes - 1), steps, out, coherent, active_lanes);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fetches = (double)grid * 256 * steps;
            printf("active %2d %s loads/lane %d  ws %8.2f MiB  blocks/CU %d : %7.1f ns/step  %7.2f Gfetch/s  %7.2f TB/s\n", active_lanes, coherent ? "wave-uniform" : "divergent   ", nl,
                   ws_nodes * 16.0 * nl / 1048576, bpc, ms * 1e6 / steps, fetches / ms / 1e6, fetches * nl * 16 / ms / 1e9);
        }
    }
    return 0;
}
