// Diagnostic: what ONE more kernel in a stream costs on this stack.  Empty kernels and kernels that touch a little memory,
// issued back to back (the stream orders them: each waits for the one before), timed over many launches; and the same kernel
// bracketed by events, one at a time (what tools/time_step.py's per-kernel figures include).
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void empty_kernel() {}
__global__ void touch_kernel(float *p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1.f;
}

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

template <class F>
static double stream_us(F launch, int n)
{
    for (int i = 0; i < 50; ++i) launch();
    (void)hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) launch();
    (void)hipDeviceSynchronize();
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

template <class F>
static double event_us(F launch, int n)
{
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < n + 10; ++i) {
        (void)hipEventRecord(a, 0);
        launch();
        (void)hipEventRecord(b, 0);
        (void)hipEventSynchronize(b);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, a, b);
        if (i >= 10) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    float *p = nullptr;
    const int n = 1 << 20;
    CK(hipMalloc(&p, sizeof(float) * n));
    CK(hipMemset(p, 0, sizeof(float) * n));
    const int N = 20000;
    printf("%-44s %8s %10s\n", "kernel", "stream", "by events");
    printf("%-44s %8.2f %10.2f us\n", "empty, 1 block of 64", stream_us([] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }, N),
           event_us([] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }, 300));
    printf("%-44s %8.2f %10.2f us\n", "empty, 2048 blocks of 256", stream_us([] { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0); }, N),
           event_us([] { hipLaunchKernelGGL(empty_kernel, dim3(2048), dim3(256), 0, 0); }, 300));
    printf("%-44s %8.2f %10.2f us\n", "empty, 16640 blocks of 256", stream_us([] { hipLaunchKernelGGL(empty_kernel, dim3(16640), dim3(256), 0, 0); }, N),
           event_us([] { hipLaunchKernelGGL(empty_kernel, dim3(16640), dim3(256), 0, 0); }, 300));
    printf("%-44s %8.2f %10.2f us\n", "p[i] += 1 over 4 MiB, 4096 blocks of 256", stream_us([=] { hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, 0, p, n); }, N),
           event_us([=] { hipLaunchKernelGGL(touch_kernel, dim3(n / 256), dim3(256), 0, 0, p, n); }, 300));
    printf("%-44s %8.2f %10.2f us\n", "p[i] += 1 over 256 B, 1 block of 64", stream_us([=] { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, 0, p, 64); }, N),
           event_us([=] { hipLaunchKernelGGL(touch_kernel, dim3(1), dim3(64), 0, 0, p, 64); }, 300));
    (void)hipFree(p);
    return 0;
}
