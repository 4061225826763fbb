// Measures the floor of a chain of dependent kernel launches on one stream (empty kernels, and kernels that
// only stream N doubles), to separate launch/dependency overhead from work in the particle-filter timestep.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_empty(int) {}
__global__ void k_touch(double* p, const double* q, long n) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i < n) { double2 v = *reinterpret_cast<const double2*>(q + i); v.x += 1.0; v.y += 1.0; *reinterpret_cast<double2*>(p + i) = v; }
}
__global__ void k_dep(double* p, const double* q) {   // one dependent global round trip per block
    if (threadIdx.x == 0) p[blockIdx.x] = q[blockIdx.x] + 1.0;
}
int main() {
    const long n = 1 << 20;
    double *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char* name, auto fn, int reps) {
        for (int i = 0; i < 50; ++i) fn();
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) fn();
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.2f us per launch\n", name, 1e3 * ms / reps);
    };
    time("empty, grid 1 x 64", [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, 0); }, 3000);
    time("empty, grid 977 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(977), dim3(256), 0, s, 0); }, 3000);
    time("empty, grid 1954 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(1954), dim3(256), 0, s, 0); }, 3000);
    time("one dependent 8-B load+store per block, 977", [&] { hipLaunchKernelGGL(k_dep, dim3(977), dim3(256), 0, s, a, b); }, 3000);
    time("stream 8 MB read + 8 MB write (1e6 doubles)", [&] { hipLaunchKernelGGL(k_touch, dim3((n / 2 + 255) / 256), dim3(256), 0, s, a, b, n); }, 3000);
    // the same chains captured once into a hipGraph and replayed: does a graph shorten the dependent-launch gap?
    auto time_graph = [&](const char* name, auto fn, int chain, int reps) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < chain; ++i) fn();
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipGraphLaunch(ge, s);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("graph: %-37s %8.2f us per launch\n", name, 1e3 * ms / (reps * chain));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    };
    time_graph("empty, grid 977 x 256", [&] { hipLaunchKernelGGL(k_empty, dim3(977), dim3(256), 0, s, 0); }, 1000, 5);
    time_graph("one dependent load+store per block, 977", [&] { hipLaunchKernelGGL(k_dep, dim3(977), dim3(256), 0, s, a, b); }, 1000, 5);
    time_graph("stream 8 MB read + 8 MB write", [&] { hipLaunchKernelGGL(k_touch, dim3((n / 2 + 255) / 256), dim3(256), 0, s, a, b, n); }, 1000, 5);
    return 0;
}
