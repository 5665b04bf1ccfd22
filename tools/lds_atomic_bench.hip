// Microbenchmark: LDS atomic throughput on gfx950 (ds_add_f32 / ds_add_u32 / ds_add_u64 / ds_add_f64 vs plain RMW),
// for three address patterns.  Diagnostic tool for the scatter design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kIters = 256;
template <int MODE, int PAT, int LM = 0> // LM: only lanes with (lane & LM) == 0 issue the atomic (does the cost follow the active lanes?)
__global__ __launch_bounds__(256) void k(const int *__restrict__ addr, float *out)
{
    __shared__ float tf[8192];
    int *ti = reinterpret_cast<int *>(tf);
    for (int e = threadIdx.x; e < 8192; e += 256) tf[e] = 0.f;
    __syncthreads();
    int a[8];
    for (int j = 0; j < 8; ++j) a[j] = addr[(PAT * 8 + j) * 256 + threadIdx.x];
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int e = (a[j] + it * 37) & 8191;
            if (PAT >= 2) e = a[j];
            if (LM != 0 && (threadIdx.x & LM) != 0) continue;
            if (MODE == 0) atomicAdd(&tf[e], 1.0f);
            if (MODE == 1) atomicAdd(&ti[e], 3);
            if (MODE == 2) tf[e] += 1.0f; // racy plain RMW: rate reference only
            if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long *>(tf) + (e >> 1), 3ull);   // ds_add_u64
            if (MODE == 4) atomicAdd(reinterpret_cast<double *>(tf) + (e >> 1), 1.0);                // ds_add_f64
            if (MODE == 5) atomicMax(&ti[e], it);                                                     // ds_max_i32
        }
    }
    __syncthreads();
    float s = 0;
    for (int e = threadIdx.x; e < 8192; e += 256) s += tf[e];
    if (s == 12345.f) out[0] = s;
}

int main()
{
    std::vector<int> h(5 * 8 * 256);
    srand(1);
    for (int j = 0; j < 8; ++j)
        for (int t = 0; t < 256; ++t) {
            h[(0 * 8 + j) * 256 + t] = t * 1 + j * 256;          // conflict-free, distinct addresses
            h[(1 * 8 + j) * 256 + t] = rand() & 8191;            // random
            h[(2 * 8 + j) * 256 + t] = ((t >> 2) * 33 + j) & 8191; // 4 lanes share an address (adjacent rays)
            h[(3 * 8 + j) * 256 + t] = ((t >> 1) * 33 + j) & 8191; // 2 lanes share an address
            h[(4 * 8 + j) * 256 + t] = ((t >> 3) * 33 + j) & 8191; // 8 lanes share an address
        }
    int *d; float *o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8;
    auto run = [&](auto kern, const char *name) {
        kern<<<blocks, 256>>>(d, o);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        kern<<<blocks, 256>>>(d, o);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double lane_ops = (double)blocks * 256 * 8 * kIters;
        // cycles per wave-instruction per CU: time * 2.4e9 / (wave-instr per CU)
        double wi_per_cu = (double)blocks * 4 * 8 * kIters / 256.0;
        printf("%-28s %8.3f ms  %7.2f Glane-ops/s  ~%6.1f cyc/wave-instr/CU\n", name, ms, lane_ops / ms / 1e6,
               ms * 1e-3 * 2.4e9 / wi_per_cu);
    };
    run(k<0, 0>, "ds_add_f32 conflict-free");
    run(k<0, 1>, "ds_add_f32 random");
    run(k<0, 2>, "ds_add_f32 4-lanes-same");
    run(k<1, 0>, "ds_add_u32 conflict-free");
    run(k<1, 1>, "ds_add_u32 random");
    run(k<1, 2>, "ds_add_u32 4-lanes-same");
    run(k<3, 0>, "ds_add_u64 conflict-free");
    run(k<3, 1>, "ds_add_u64 random");
    run(k<3, 2>, "ds_add_u64 4-lanes-same");
    run(k<4, 0>, "ds_add_f64 conflict-free");
    run(k<4, 1>, "ds_add_f64 random");
    run(k<4, 2>, "ds_add_f64 4-lanes-same");
    run(k<4, 3>, "ds_add_f64 2-lanes-same");
    run(k<4, 4>, "ds_add_f64 8-lanes-same");
    run(k<4, 0, 1>, "ds_add_f64 c-free, 32 lanes");
    run(k<4, 0, 3>, "ds_add_f64 c-free, 16 lanes");
    run(k<4, 0, 7>, "ds_add_f64 c-free, 8 lanes");
    run(k<4, 0, 63>, "ds_add_f64 c-free, 1 lane");
    run(k<4, 2, 3>, "ds_add_f64 4-same, 1 of each");
    run(k<1, 0, 3>, "ds_add_u32 c-free, 16 lanes");
    run(k<3, 0, 3>, "ds_add_u64 c-free, 16 lanes");
    run(k<3, 4>, "ds_add_u64 8-lanes-same");
    run(k<1, 4>, "ds_add_u32 8-lanes-same");
    run(k<5, 1>, "ds_max_i32 random");
    run(k<2, 0>, "plain rmw conflict-free");
    run(k<2, 1>, "plain rmw random");
    return 0;
}
