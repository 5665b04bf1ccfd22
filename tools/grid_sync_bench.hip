// Diagnostic: what a grid-wide barrier inside ONE cooperative launch costs on this stack, against a kernel boundary
// (tools/launch_floor.hip) -- the question behind "one 256 x 512 frame forward + backward as one launch" (VERDICT r4 item 6).
// A cooperative kernel of 256 blocks of 256 threads (one per CU: all resident; hipLaunchCooperativeKernel checks it) that
// touches 1 MiB, then does n grid syncs with a dependent read-modify-write of another block's data between them.
//   hipcc -O3 --offload-arch=gfx950 tools/grid_sync_bench.hip -o /tmp/grid_sync_bench && /tmp/grid_sync_bench
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ __launch_bounds__(256) void phases_kernel(float *p, int nsync)
{
    cg::grid_group grid = cg::this_grid();
    const int n = gridDim.x * blockDim.x;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    p[i] += 1.f;
    for (int s = 0; s < nsync; ++s) {
        grid.sync();
        i = (i + 4099 * blockDim.x) % n; // another block's data (another XCD's, mostly)
        p[i] += 1.f;
    }
}
__global__ __launch_bounds__(256) void phase_kernel(float *p, int shift)
{
    const int n = gridDim.x * blockDim.x;
    const int i = (blockIdx.x * blockDim.x + threadIdx.x + shift) % n;
    p[i] += 1.f;
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

int main()
{
    const int blocks = 256, threads = 256, n = blocks * threads;
    float *p = nullptr;
    CK(hipMalloc(&p, sizeof(float) * n));
    CK(hipMemset(p, 0, sizeof(float) * n));
    const int N = 2000;
    for (int nsync = 0; nsync <= 4; ++nsync) {
        void *args[] = {(void *)&p, (void *)&nsync};
        for (int i = 0; i < 20; ++i) CK(hipLaunchCooperativeKernel((const void *)phases_kernel, dim3(blocks), dim3(threads), args, 0, 0));
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) CK(hipLaunchCooperativeKernel((const void *)phases_kernel, dim3(blocks), dim3(threads), args, 0, 0));
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        // the same phases as nsync + 1 ordinary launches
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(phase_kernel, dim3(blocks), dim3(threads), 0, 0, p, 0);
        CK(hipDeviceSynchronize());
        const auto t1 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i)
            for (int s = 0; s <= nsync; ++s) hipLaunchKernelGGL(phase_kernel, dim3(blocks), dim3(threads), 0, 0, p, s * 4099 * threads);
        CK(hipDeviceSynchronize());
        const double us2 = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t1).count() / N;
        printf("%d grid sync(s): cooperative launch %7.2f us   |   %d ordinary launch(es) %7.2f us\n", nsync, us, nsync + 1, us2);
    }
    (void)hipFree(p);
    return 0;
}
