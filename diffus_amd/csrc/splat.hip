// splat.hip -- scan conversion (differentiable_splat) and the benchmark loss, with their C-ABI entry points
#include "diffus_host.hpp"

namespace {

// ----------------------------------------------------------------------------
// SCAN CONVERSION  (SURVEY §8f row 1): differentiable_splat, reference src/renderer.py:694-737.
//   image[idx1, idx0] "+=" intensities is an index_put WITHOUT accumulation: of the samples that
//   round to one pixel the LAST one in flattened order wins, and weight is 1 where any sample
//   landed (:717-722).  Then both are blurred with a normalised Gaussian (zero padding) and divided
//   (:725-735); the result is returned transposed (:737).  Autograd hands every sample the gradient
//   of its pixel, winners and losers alike (index_put's backward is a gather).
// Pipeline: winner (atomicMax of the sample index) -> compose (image, weight) -> separable blur ->
// divide + transpose.  Backward: q = gout^T / (blur(weight) + eps) -> blur -> gather per sample.
constexpr int kSplatTile = 32;
constexpr int kSplatMaxHalf = 24; // kernel half-width int(6 sigma)|1 >> 1  =>  sigma <= 8

__device__ __forceinline__ int splat_pixel(float c0, float c1, int H, int W)
{
    // clamp(round(coord).long(), 0, size-1) with round-half-even (reference :717-718)
    return nearest_index(c1, H) * W + nearest_index(c0, W);
}

// winner[pixel] = largest sample index landing on it.  Samples are visited from the LAST to the
// first and a sample first looks (plain load) whether a later one already owns its pixel: a stale
// look only costs a redundant atomic, never a wrong answer, and it removes almost all of the
// same-address atomics around the fan apex (4.2 M contended atomicMax took 1.6 ms without it).
__global__ __launch_bounds__(kBlock) void splat_winner_kernel(const float *__restrict__ c0, const float *__restrict__ c1,
                                                              long n, int H, int W, int *winner)
{
    const long pz = blockIdx.y;
    for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) {
        const long s = n - 1 - t;
        int *w = &winner[pz * H * W + splat_pixel(c0[pz * n + s], c1[pz * n + s], H, W)];
        if (__builtin_nontemporal_load(w) < (int)s) atomicMax(w, (int)s);
    }
}

// The same, privatised: a block takes a patch of kPatchRays rows x kPatchSteps columns of the (rows, cols) sample
// grid (adjacent rays x consecutive steps: a compact pixel footprint), resolves the winner per
// pixel in an LDS tile over the patch's pixel bounding box (ds_max_i32), and issues ONE global
// atomicMax per touched pixel.  Falls back to the direct form when the box does not fit.
__global__ __launch_bounds__(kBlock) void splat_winner_patch_kernel(const float *__restrict__ c0,
                                                                    const float *__restrict__ c1, int rows, int cols,
                                                                    int H, int W, int *winner, int row_groups,
                                                                    int col_groups)
{
    __shared__ int tile[kTileCap];
    __shared__ int s_lo[2], s_hi[2];
    const long pz = blockIdx.y, n = (long)rows * cols;
    const int cg = blockIdx.x % col_groups, rg = blockIdx.x / col_groups;
    const int tid = threadIdx.x;
    const int row = rg * kPatchRays + tid / (kPatchSteps / kSamplesPerThread);
    const int cbase = cg * kPatchSteps + (tid % (kPatchSteps / kSamplesPerThread)) * kSamplesPerThread;
    if (tid < 2) {
        s_lo[tid] = 0x7fffffff;
        s_hi[tid] = -1;
    }
    int px[kSamplesPerThread], py[kSamplesPerThread], sid[kSamplesPerThread];
    int lo0 = 0x7fffffff, lo1 = 0x7fffffff, hi0 = -1, hi1 = -1;
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
        int c = cbase + q;
        sid[q] = -1;
        px[q] = py[q] = 0;
        if (row < rows && c < cols) {
            long s = (long)row * cols + c;
            sid[q] = (int)s;
            px[q] = nearest_index(c0[pz * n + s], W);
            py[q] = nearest_index(c1[pz * n + s], H);
            lo0 = min(lo0, px[q]); hi0 = max(hi0, px[q]);
            lo1 = min(lo1, py[q]); hi1 = max(hi1, py[q]);
        }
    }
    lo0 = wave_reduce_minmax<true>(lo0); hi0 = wave_reduce_minmax<false>(hi0);
    lo1 = wave_reduce_minmax<true>(lo1); hi1 = wave_reduce_minmax<false>(hi1);
    __syncthreads();
    if ((tid & 63) == 0) {
        atomicMin(&s_lo[0], lo0); atomicMax(&s_hi[0], hi0);
        atomicMin(&s_lo[1], lo1); atomicMax(&s_hi[1], hi1);
    }
    __syncthreads();
    if (s_hi[0] < 0) return;
    const int l0 = s_lo[0], l1 = s_lo[1], b0 = s_hi[0] - l0 + 1, b1 = s_hi[1] - l1 + 1;
    int *wz = winner + pz * H * W;
    if ((long)b0 * b1 > kTileCap) {
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
            if (sid[q] >= 0) {
                int *w = &wz[py[q] * W + px[q]];
                if (__builtin_nontemporal_load(w) < sid[q]) atomicMax(w, sid[q]);
            }
        return;
    }
    const int nt = b0 * b1;
    for (int e = tid; e < nt; e += kBlock) tile[e] = -1;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q)
        if (sid[q] >= 0) atomicMax(&tile[(py[q] - l1) * b0 + (px[q] - l0)], sid[q]);
    __syncthreads();
    const float rb0 = __frcp_rn((float)b0);
    for (int e = tid; e < nt; e += kBlock) {
        int v = tile[e];
        if (v >= 0) {
            int y = __float2int_rz(((float)e + 0.5f) * rb0), x = e - y * b0;
            // no look-before-you-leap here: after the tile only the few patches that overlap a pixel meet on it, and
            // a load in front of every atomic put a global round trip into each trip of this loop (83 -> 3x us)
            atomicMax(&wz[(l1 + y) * W + (l0 + x)], v);
        }
    }
}

// backward only needs WHERE samples landed (the weight plane): plain stores of 1, no atomics
__global__ __launch_bounds__(kBlock) void splat_mark_kernel(const float *__restrict__ c0, const float *__restrict__ c1,
                                                            long n, int H, int W, float *__restrict__ planes)
{
    const long pz = blockIdx.y, hw = (long)H * W;
    for (long s = (long)blockIdx.x * kBlock + threadIdx.x; s < n; s += (long)gridDim.x * kBlock)
        planes[(pz * 2 + 1) * hw + splat_pixel(c0[pz * n + s], c1[pz * n + s], H, W)] = 1.f;
}

// planes (P,2,H,W): [0] = image, [1] = weight
__global__ __launch_bounds__(kBlock) void splat_compose_kernel(const int *__restrict__ winner, const float *__restrict__ val,
                                                               long n, long hw, float *__restrict__ planes)
{
    const long pz = blockIdx.y;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < hw; i += (long)gridDim.x * kBlock) {
        int wn = winner[pz * hw + i];
        planes[(pz * 2 + 0) * hw + i] = wn >= 0 ? (val ? val[pz * n + wn] : 0.f) : 0.f;
        planes[(pz * 2 + 1) * hw + i] = wn >= 0 ? 1.f : 0.f;
    }
}

// Separable zero-padded Gaussian blur of planes of H x W (plane z of the grid at in + z * in_stride -> out + z *
// out_stride), one 32x32 tile per block.  Generic in the kernel half-width (up to kSplatMaxHalf).
__global__ __launch_bounds__(kBlock) void blur2d_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W,
                                                        int half, float sigma, long in_stride, long out_stride)
{
    __shared__ float kw[2 * kSplatMaxHalf + 1];
    __shared__ float src[kSplatTile + 2 * kSplatMaxHalf][kSplatTile + 2 * kSplatMaxHalf + 1];
    __shared__ float mid[kSplatTile + 2 * kSplatMaxHalf][kSplatTile + 1];
    const int size = 2 * half + 1, ext = kSplatTile + 2 * half;
    const float *pin = in + (long)blockIdx.z * in_stride;
    float *pout = out + (long)blockIdx.z * out_stride;
    const int x0 = blockIdx.x * kSplatTile, y0 = blockIdx.y * kSplatTile;
    if ((int)threadIdx.x < size) { // exp(-0.5 (c/sigma)^2) / sum, reference :726-728
        float c = (float)((int)threadIdx.x - half) / sigma;
        kw[threadIdx.x] = expf(-0.5f * c * c);
    }
    __syncthreads();
    float ksum = 0.f;
    for (int i = 0; i < size; ++i) ksum += kw[i];
    __syncthreads();
    if ((int)threadIdx.x < size) kw[threadIdx.x] = kw[threadIdx.x] / ksum;
    for (int e = threadIdx.x; e < ext * ext; e += kBlock) {
        int r = e / ext, c = e - r * ext;
        int y = y0 + r - half, x = x0 + c - half;
        src[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? pin[(long)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < ext * kSplatTile; e += kBlock) { // along x
        int r = e / kSplatTile, c = e - r * kSplatTile;
        float a = 0.f;
        for (int t = 0; t < size; ++t) a = __builtin_fmaf(kw[t], src[r][c + t], a);
        mid[r][c] = a;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kSplatTile * kSplatTile; e += kBlock) { // along y
        int r = e / kSplatTile, c = e - r * kSplatTile;
        int y = y0 + r, x = x0 + c;
        if (y < H && x < W) {
            float a = 0.f;
            for (int t = 0; t < size; ++t) a = __builtin_fmaf(kw[t], mid[r + t][c], a);
            pout[(long)y * W + x] = a;
        }
    }
}

// The same for half-widths up to 8 (sigma < 3: every demo uses 1 or 2), register-tiled: a thread produces 8
// consecutive outputs of a row (x pass) or 4 of a column (y pass) from one sliding window of LDS reads -- 20 reads for
// 8 x 13 multiply-adds at sigma = 2 instead of 104.  Same summation order per output as the generic kernel.
template <int HALF>
__global__ __launch_bounds__(kBlock) void blur2d_tiled_kernel(const float *__restrict__ in, float *__restrict__ out, int H,
                                                              int W, float sigma, long in_stride, long out_stride)
{
    constexpr int SIZE = 2 * HALF + 1, EXT = kSplatTile + 2 * HALF;
    __shared__ float kw[SIZE];
    __shared__ float src[EXT][EXT + 1];
    __shared__ float mid[EXT][kSplatTile + 1];
    const float *pin = in + (long)blockIdx.z * in_stride;
    float *pout = out + (long)blockIdx.z * out_stride;
    const int x0 = blockIdx.x * kSplatTile, y0 = blockIdx.y * kSplatTile;
    const int tid = threadIdx.x;
    if (tid < SIZE) {
        float c = (float)(tid - HALF) / sigma;
        kw[tid] = expf(-0.5f * c * c);
    }
    for (int e = tid; e < EXT * EXT; e += kBlock) {
        int r = e / EXT, c = e - r * EXT;
        int y = y0 + r - HALF, x = x0 + c - HALF;
        src[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? pin[(long)y * W + x] : 0.f;
    }
    __syncthreads();
    float k[SIZE], ksum = 0.f;
#pragma unroll
    for (int i = 0; i < SIZE; ++i) ksum += kw[i];
#pragma unroll
    for (int i = 0; i < SIZE; ++i) k[i] = kw[i] / ksum;
    // x pass: EXT rows x 4 groups of 8 columns
    for (int item = tid; item < EXT * (kSplatTile / 8); item += kBlock) {
        const int r = item / (kSplatTile / 8), c0 = (item - r * (kSplatTile / 8)) * 8;
        float win[8 + 2 * HALF];
#pragma unroll
        for (int i = 0; i < 8 + 2 * HALF; ++i) win[i] = src[r][c0 + i];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < SIZE; ++t) a = __builtin_fmaf(k[t], win[o + t], a);
            mid[r][c0 + o] = a;
        }
    }
    __syncthreads();
    // y pass: 32 columns x 8 groups of 4 rows; a wave covers two row groups of all 32 columns (128-B store runs)
    {
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        float win[4 + 2 * HALF];
#pragma unroll
        for (int i = 0; i < 4 + 2 * HALF; ++i) win[i] = mid[r0 + i][c];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < SIZE; ++t) a = __builtin_fmaf(k[t], win[o + t], a);
            const int y = y0 + r0 + o, x = x0 + c;
            if (y < H && x < W) pout[(long)y * W + x] = a;
        }
    }
    static_assert(kBlock == 256 && kSplatTile == 32, "the y pass maps 256 threads to 32 columns x 8 row groups");
}

// launch the blur of `planes` planes
int launch_blur(const float *in, float *out, int H, int W, int half, float sigma, long in_stride, long out_stride,
                int planes, hipStream_t st)
{
    dim3 tiles((W + kSplatTile - 1) / kSplatTile, (H + kSplatTile - 1) / kSplatTile, planes);
#define DIFFUS_BLUR_CASE(HF)                                                                                          \
    case HF:                                                                                                          \
        hipLaunchKernelGGL(blur2d_tiled_kernel<HF>, tiles, dim3(kBlock), 0, st, in, out, H, W, sigma, in_stride,      \
                           out_stride);                                                                               \
        break;
    switch (half) {
        DIFFUS_BLUR_CASE(1) DIFFUS_BLUR_CASE(2) DIFFUS_BLUR_CASE(3) DIFFUS_BLUR_CASE(4)
        DIFFUS_BLUR_CASE(5) DIFFUS_BLUR_CASE(6) DIFFUS_BLUR_CASE(7) DIFFUS_BLUR_CASE(8)
    default:
        hipLaunchKernelGGL(blur2d_kernel, tiles, dim3(kBlock), 0, st, in, out, H, W, half, sigma, in_stride, out_stride);
    }
#undef DIFFUS_BLUR_CASE
    return last_launch();
}

// The tail of the splat in ONE launch per direction (half-widths up to 8): a graph node costs ~4.8 us on this stack
// whatever it does, and compose -> blur -> divide were three of them (mark -> blur -> divide in the backward).
//   FWD: image and weight planes straight from the winner raster (val[winner] / 1 where a sample landed), both blurred in
//        LDS, out[p, x, y] = bimg / (bw + 1e-8), stored transposed.
//   BWD: the weight plane from the winner raster the forward kept, blurred, q[p, y, x] = gout[p, x, y] / (bw + 1e-8).
// Same operations in the same order per output as splat_compose / blur2d_tiled / splat_divide: bit-identical.
template <int HALF, bool BWD>
__global__ __launch_bounds__(kBlock) void splat_fused_kernel(const int *__restrict__ winner, const float *__restrict__ val, long n,
                                                             const float *__restrict__ gout, float *__restrict__ out, int H,
                                                             int W, float sigma)
{
    constexpr int SIZE = 2 * HALF + 1, EXT = kSplatTile + 2 * HALF, NP = BWD ? 1 : 2;
    __shared__ float kw[SIZE];
    __shared__ float src[NP][EXT][EXT + 1];
    __shared__ float mid[NP][EXT][kSplatTile + 1];
    __shared__ float t[kSplatTile][kSplatTile + 1];
    const long pz = blockIdx.z, hw = (long)H * W;
    const int x0 = blockIdx.x * kSplatTile, y0 = blockIdx.y * kSplatTile;
    const int tid = threadIdx.x;
    if (tid < SIZE) {
        float c = (float)(tid - HALF) / sigma;
        kw[tid] = expf(-0.5f * c * c);
    }
    for (int e = tid; e < EXT * EXT; e += kBlock) {
        int r = e / EXT, c = e - r * EXT;
        int y = y0 + r - HALF, x = x0 + c - HALF;
        const int wn = (y >= 0 && y < H && x >= 0 && x < W) ? winner[pz * hw + (long)y * W + x] : -1;
        if (BWD) {
            src[0][r][c] = wn >= 0 ? 1.f : 0.f;
        } else {
            src[0][r][c] = wn >= 0 ? (val ? val[pz * n + wn] : 0.f) : 0.f;
            src[NP - 1][r][c] = wn >= 0 ? 1.f : 0.f;
        }
    }
    if (BWD) { // gout arrives transposed (p, x, y): through the tile, both sides coalesced
        for (int e = tid; e < kSplatTile * kSplatTile; e += kBlock) {
            int c = e / kSplatTile, r = e - c * kSplatTile, y = y0 + r, x = x0 + c;
            t[r][c] = (y < H && x < W) ? gout[pz * hw + (long)x * H + y] : 0.f;
        }
    }
    __syncthreads();
    float k[SIZE], ksum = 0.f;
#pragma unroll
    for (int i = 0; i < SIZE; ++i) ksum += kw[i];
#pragma unroll
    for (int i = 0; i < SIZE; ++i) k[i] = kw[i] / ksum;
    for (int item = tid; item < NP * EXT * (kSplatTile / 8); item += kBlock) { // x pass: EXT rows x 4 groups of 8 columns, per plane
        const int pl = item / (EXT * (kSplatTile / 8)), it = item - pl * (EXT * (kSplatTile / 8));
        const int r = it / (kSplatTile / 8), c0 = (it - r * (kSplatTile / 8)) * 8;
        float win[8 + 2 * HALF];
#pragma unroll
        for (int i = 0; i < 8 + 2 * HALF; ++i) win[i] = src[pl][r][c0 + i];
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            float a = 0.f;
#pragma unroll
            for (int q = 0; q < SIZE; ++q) a = __builtin_fmaf(k[q], win[o + q], a);
            mid[pl][r][c0 + o] = a;
        }
    }
    __syncthreads();
    {   // y pass: 32 columns x 8 groups of 4 rows
        const int c = tid & 31, r0 = (tid >> 5) * 4;
        float res[NP][4];
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) {
            float win[4 + 2 * HALF];
#pragma unroll
            for (int i = 0; i < 4 + 2 * HALF; ++i) win[i] = mid[pl][r0 + i][c];
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < SIZE; ++q) a = __builtin_fmaf(k[q], win[o + q], a);
                res[pl][o] = a;
            }
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const int y = y0 + r0 + o, x = x0 + c;
            if (BWD) {
                if (y < H && x < W) out[pz * hw + (long)y * W + x] = t[r0 + o][c] / (res[0][o] + 1e-8f);
            } else {
                t[r0 + o][c] = res[0][o] / (res[NP - 1][o] + 1e-8f);
            }
        }
    }
    if (!BWD) {
        __syncthreads();
        for (int e = tid; e < kSplatTile * kSplatTile; e += kBlock) {
            int c = e / kSplatTile, r = e - c * kSplatTile, y = y0 + r, x = x0 + c;
            if (y < H && x < W) out[pz * hw + (long)x * H + y] = t[r][c];
        }
    }
    static_assert(kBlock == 256 && kSplatTile == 32, "the y pass maps 256 threads to 32 columns x 8 row groups");
}

template <bool BWD>
int launch_splat_fused(const int *winner, const float *val, long n, const float *gout, float *out, int H, int W, int half,
                       float sigma, int P, hipStream_t st)
{
    dim3 tiles((W + kSplatTile - 1) / kSplatTile, (H + kSplatTile - 1) / kSplatTile, P);
#define DIFFUS_FUSED_CASE(HF)                                                                                                   \
    case HF:                                                                                                                    \
        hipLaunchKernelGGL((splat_fused_kernel<HF, BWD>), tiles, dim3(kBlock), 0, st, winner, val, n, gout, out, H, W, sigma);  \
        break;
    switch (half) {
        DIFFUS_FUSED_CASE(1) DIFFUS_FUSED_CASE(2) DIFFUS_FUSED_CASE(3) DIFFUS_FUSED_CASE(4)
        DIFFUS_FUSED_CASE(5) DIFFUS_FUSED_CASE(6) DIFFUS_FUSED_CASE(7) DIFFUS_FUSED_CASE(8)
    default: return DIFFUS_EUNSUPPORTED;
    }
#undef DIFFUS_FUSED_CASE
    return last_launch();
}
constexpr int kSplatFusedMaxHalf = 8;

// forward: out[p, x, y] = bimg[y, x] / (bw[y, x] + 1e-8)          (the .T of reference :737)
// backward: q[p, y, x]  = gout[p, x, y] / (bw[y, x] + 1e-8)
template <bool BWD>
__global__ __launch_bounds__(kBlock) void splat_divide_kernel(const float *__restrict__ blurred, const float *__restrict__ gout,
                                                              float *__restrict__ out, int H, int W)
{
    __shared__ float t[kSplatTile][kSplatTile + 1];
    const long pz = blockIdx.z, hw = (long)H * W;
    const int x0 = blockIdx.x * kSplatTile, y0 = blockIdx.y * kSplatTile;
    const float *bimg = blurred + (pz * 2 + 0) * hw, *bw = blurred + (pz * 2 + 1) * hw;
    if (!BWD) {
        for (int e = threadIdx.x; e < kSplatTile * kSplatTile; e += kBlock) {
            int r = e / kSplatTile, c = e - r * kSplatTile, y = y0 + r, x = x0 + c;
            t[r][c] = (y < H && x < W) ? bimg[(long)y * W + x] / (bw[(long)y * W + x] + 1e-8f) : 0.f;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kSplatTile * kSplatTile; e += kBlock) {
            int c = e / kSplatTile, r = e - c * kSplatTile, y = y0 + r, x = x0 + c;
            if (y < H && x < W) out[pz * hw + (long)x * H + y] = t[r][c];
        }
    } else {
        for (int e = threadIdx.x; e < kSplatTile * kSplatTile; e += kBlock) {
            int c = e / kSplatTile, r = e - c * kSplatTile, y = y0 + r, x = x0 + c;
            t[r][c] = (y < H && x < W) ? gout[pz * hw + (long)x * H + y] : 0.f;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < kSplatTile * kSplatTile; e += kBlock) {
            int r = e / kSplatTile, c = e - r * kSplatTile, y = y0 + r, x = x0 + c;
            if (y < H && x < W) out[pz * hw + (long)y * W + x] = t[r][c] / (bw[(long)y * W + x] + 1e-8f);
        }
    }
}

__global__ __launch_bounds__(kBlock) void splat_gather_kernel(const float *__restrict__ c0, const float *__restrict__ c1,
                                                              long n, int H, int W, const float *__restrict__ gimg,
                                                              float *__restrict__ gval)
{
    const long pz = blockIdx.y;
    for (long s = (long)blockIdx.x * kBlock + threadIdx.x; s < n; s += (long)gridDim.x * kBlock)
        gval[pz * n + s] = gimg[pz * H * W + splat_pixel(c0[pz * n + s], c1[pz * n + s], H, W)];
}

// ----------------------------------------------------------------------------
// Which two coordinate axes span the image: the two of largest variance, largest first, ties in axis order (reference
// src/renderer.py:702-707: `variances = [c.float().var().item() ...]`, `sorted(range(3), key=lambda i: -variances[i])[:2]`).
// The reference decides on the HOST (three .item() syncs); here the decision stays on the device -- one block sums the
// three planes (values cast to float32 like `.float()`, sums in float64), writes the two axis numbers, and a second
// launch casts the two chosen planes to float32 for the splat kernels -- so render -> splat -> loss -> backward can be
// captured as one hipGraph.
__device__ __forceinline__ float coord_f32(const void *p, int dtype, long i)
{
    if (dtype == DIFFUS_F32) return ((const float *)p)[i];
    if (dtype == DIFFUS_F64) return (float)((const double *)p)[i];
    return (float)((const long long *)p)[i]; // DIFFUS_I64
}

__device__ __forceinline__ void axes_from_variances(const double (&var)[3], int *__restrict__ axes)
{
    // stable descending order of three keys; a NaN variance (NaN coordinates) sorts last
    int o[3] = {0, 1, 2};
    auto before = [&](int a, int b) { // a strictly before b
        const double va = var[a], vb = var[b];
        if (va != va) return false;
        if (vb != vb) return true;
        return va > vb;
    };
    for (int i = 1; i < 3; ++i)
        for (int j = i; j > 0 && before(o[j], o[j - 1]); --j) {
            const int t = o[j]; o[j] = o[j - 1]; o[j - 1] = t;
        }
    axes[0] = o[0];
    axes[1] = o[1];
}

constexpr int kAxesThreads = 1024;
__global__ __launch_bounds__(kAxesThreads) void splat_axes_kernel(const void *x, const void *y, const void *z, int dx, int dy,
                                                                  int dz, long n, int *__restrict__ axes)
{
    __shared__ double sm[6][kAxesThreads / kWave];
    __shared__ double var[3];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // ONE pass over the three planes (six running sums): the block's time is the latency of its load chain.
    // Shifted sums (first element as the pivot): no cancellation for coordinates far from 0.
    const double p0 = (double)coord_f32(x, dx, 0), p1 = (double)coord_f32(y, dy, 0), p2 = (double)coord_f32(z, dz, 0);
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (long i = threadIdx.x; i < n; i += kAxesThreads) {
        const double a = (double)coord_f32(x, dx, i) - p0, b = (double)coord_f32(y, dy, i) - p1, c = (double)coord_f32(z, dz, i) - p2;
        acc[0] += a; acc[1] += a * a;
        acc[2] += b; acc[3] += b * b;
        acc[4] += c; acc[5] += c * c;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        acc[k] = wave_sum_to_lane63(acc[k]);
        if (lane == kWave - 1) sm[k][wv] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double S = 0.0, Q = 0.0;
        for (int w = 0; w < kAxesThreads / kWave; ++w) { // fixed order: deterministic
            S += sm[2 * threadIdx.x][w];
            Q += sm[2 * threadIdx.x + 1][w];
        }
        var[threadIdx.x] = (n > 1) ? (Q - S * S / (double)n) / (double)(n - 1) : 0.0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double v3[3] = {var[0], var[1], var[2]};
        axes_from_variances(v3, axes);
    }
}

// The same in two launches for large n (one 1024-thread block walking 37 k int64 samples took 25-45 us, the longest kernel of
// the demo frame): kAxesBlocks blocks leave six partial sums each (same pivot: element 0), a 64-thread block adds them in
// block order and sorts.  `part` is borrowed from the c0 plane, which splat_select_kernel overwrites afterwards.
constexpr int kAxesBlocks = 64;
__global__ __launch_bounds__(kBlock) void splat_axes_partial_kernel(const void *x, const void *y, const void *z, int dx, int dy,
                                                                    int dz, long n, double *__restrict__ part)
{
    __shared__ double sm[6][kWavesPerBlock];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double p0 = (double)coord_f32(x, dx, 0), p1 = (double)coord_f32(y, dy, 0), p2 = (double)coord_f32(z, dz, 0);
    double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)kAxesBlocks * kBlock) {
        const double a = (double)coord_f32(x, dx, i) - p0, b = (double)coord_f32(y, dy, i) - p1, c = (double)coord_f32(z, dz, i) - p2;
        acc[0] += a; acc[1] += a * a;
        acc[2] += b; acc[3] += b * b;
        acc[4] += c; acc[5] += c * c;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        acc[k] = wave_sum_to_lane63(acc[k]);
        if (lane == kWave - 1) sm[k][wv] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        double t = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) t += sm[threadIdx.x][w];
        part[blockIdx.x * 6 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(kWave) void splat_axes_final_kernel(const double *__restrict__ part, long n, int *__restrict__ axes)
{
    // one wave: lane b holds block b's six sums (all loads in flight at once), a DPP tree adds them -- a fixed order
    static_assert(kAxesBlocks == kWave, "one lane per partial");
    double t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = part[threadIdx.x * 6 + k];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = wave_sum_to_lane63(t[k]);
    if (threadIdx.x == kWave - 1) {
        double var[3];
        for (int a = 0; a < 3; ++a) var[a] = (n > 1) ? (t[2 * a + 1] - t[2 * a] * t[2 * a] / (double)n) / (double)(n - 1) : 0.0;
        axes_from_variances(var, axes);
    }
}

__global__ __launch_bounds__(kBlock) void splat_select_kernel(const void *x, const void *y, const void *z, int dx, int dy, int dz,
                                                              long n, const int *__restrict__ axes, float *__restrict__ c0,
                                                              float *__restrict__ c1)
{
    const int a0 = axes[0], a1 = axes[1]; // uniform: scalar loads
    const void *p0 = a0 == 0 ? x : (a0 == 1 ? y : z), *p1 = a1 == 0 ? x : (a1 == 1 ? y : z);
    const int d0 = a0 == 0 ? dx : (a0 == 1 ? dy : dz), d1 = a1 == 0 ? dx : (a1 == 1 ? dy : dz);
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        c0[i] = coord_f32(p0, d0, i);
        c1[i] = coord_f32(p1, d1, i);
    }
}

// rotate_around_apex (reference src/renderer.py:655-692) in one launch: every point (x - shift, z) turned by the angle
// between (0, 1) and `median`, then moved to `apex`.  apex / median are read from device memory (no host values: the
// call sits inside captured graphs); the angle goes through atan2 / cos / sin like the reference's.
__global__ __launch_bounds__(kBlock) void rotate_apex_kernel(const float *__restrict__ x, const float *__restrict__ z, long n,
                                                             const float *__restrict__ apex, const float *__restrict__ median,
                                                             float shift, float *__restrict__ xr, float *__restrict__ zr)
{
    const float m0 = median[0], m1 = median[1];
    const float nrm = sqrtf(m0 * m0 + m1 * m1);
    const float turn = atan2f(__fdiv_rn(m0, nrm), __fdiv_rn(m1, nrm));
    const float c = cosf(turn), sn = sinf(turn);
    const float a0 = apex[0], a1 = apex[1];
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float u = x[i] - shift, v = z[i];
        xr[i] = __fadd_rn(__fadd_rn(__fmul_rn(c, u), __fmul_rn(-sn, v)), a0);
        zr[i] = __fadd_rn(__fadd_rn(__fmul_rn(sn, u), __fmul_rn(c, v)), a1);
    }
}

// ----------------------------------------------------------------------------
// Energy loss used by the benchmarks and examples: loss[p] = sum(frame[p]^2), gframe = 2 * frame,
// in one streaming pass.  kLossSplit blocks per pose write partial sums, the last of them to arrive adds
// them in a fixed order (deterministic; one block per pose alone used only P of the 256 CUs).
constexpr int kLossSplit = 16; // blocks per pose.  (64 streamed faster, 5 us, but 64 same-address counter adds per pose
                               // serialise at the memory side: 25 us)
// The kLossSplit partial sums of a pose are handed to whichever of its blocks arrives last, which adds them in a
// fixed order (deterministic) -- no second launch.  The hand-off is 4 bytes per block, so it uses write-through
// (sc1) stores and loads around one relaxed agent-scope counter instead of fences (a release fence would write
// back the XCD's whole L2, freshly dirtied by gframe).  The counter returns to 0 for the next call.
__global__ __launch_bounds__(kBlock) void loss_sumsq_kernel(const float *__restrict__ frame, float *part, int *cnt,
                                                            float *__restrict__ loss, float *__restrict__ gframe, long n)
{
    __shared__ float sm[kWavesPerBlock];
    const long pz = blockIdx.y;
    const float *f = frame + pz * n;
    float *g = gframe ? gframe + pz * n : nullptr;
    float acc = 0.f;
    const bool vec = ((n & 3) == 0) && ((((uintptr_t)f) & 15) == 0) && (!g || (((uintptr_t)g) & 15) == 0);
    if (vec) {
        const float4 *f4 = reinterpret_cast<const float4 *>(f);
        float4 *g4 = reinterpret_cast<float4 *>(g);
#pragma unroll 4
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n / 4; i += (long)kLossSplit * kBlock) {
            float4 v = f4[i];
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            if (g) g4[i] = make_float4(2.f * v.x, 2.f * v.y, 2.f * v.z, 2.f * v.w);
        }
    } else {
        for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)kLossSplit * kBlock) {
            float v = f[i];
            acc += v * v;
            if (g) g[i] = 2.f * v;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, kWave);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) {
        __hip_atomic_store(part + pz * kLossSplit + blockIdx.x, (sm[0] + sm[1]) + (sm[2] + sm[3]), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the partial has left this CU before the counter moves
        const int prev = __hip_atomic_fetch_add(cnt + pz, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = prev == kLossSplit - 1; // every other block of this pose has published its partial
    }
    __syncthreads();
    if (s_last && threadIdx.x < kWave) { // one wave: all partials in one round trip, then a fixed-order tree
        static_assert(kLossSplit <= kWave, "one partial per lane");
        float t = threadIdx.x < kLossSplit
                      ? __hip_atomic_load(part + pz * kLossSplit + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : 0.f;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, kWave);
        if (threadIdx.x == 0) {
            loss[pz] = t;
            __hip_atomic_store(cnt + pz, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace

extern "C" {

static int splat_half(float sigma) { return ((int)(6.f * sigma) | 1) / 2; } // size = int(6 sigma) | 1, reference :725

size_t diffus_splat_workspace_bytes(int P, int H, int W)
{
    if (P <= 0 || H <= 0 || W <= 0) return 0;
    return align256(sizeof(int) * (size_t)P * H * W) + 2 * align256(sizeof(float) * (size_t)P * 2 * H * W);
}

int diffus_rotate_around_apex(const float *x, const float *z, long n, const float *apex, const float *median, float shift,
                              float *x_rot, float *z_rot, diffus_stream_t stream)
{
    if (!x || !z || !apex || !median || !x_rot || !z_rot || n <= 0) return DIFFUS_EINVAL;
    unsigned nb = (unsigned)((n + kBlock - 1) / kBlock); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(rotate_apex_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, x, z, n, apex, median, shift, x_rot, z_rot);
    return last_launch();
}

int diffus_splat_axes(const void *x, int x_dtype, const void *y, int y_dtype, const void *z, int z_dtype, long n, int *axes,
                      float *c0, float *c1, diffus_stream_t stream)
{
    if (!x || !y || !z || !axes || n <= 0) return DIFFUS_EINVAL;
    for (int d : {x_dtype, y_dtype, z_dtype})
        if (d != DIFFUS_F32 && d != DIFFUS_F64 && d != DIFFUS_I64) return DIFFUS_EINVAL;
    if ((c0 == nullptr) != (c1 == nullptr)) return DIFFUS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (c0 && n >= (long)kAxesBlocks * 6 * 2 && n >= 8192 && (reinterpret_cast<size_t>(c0) & 7) == 0) {
        double *part = reinterpret_cast<double *>(c0); // kAxesBlocks x 6 doubles fit (n floats), rewritten by the select below
        hipLaunchKernelGGL(splat_axes_partial_kernel, dim3(kAxesBlocks), dim3(kBlock), 0, st, x, y, z, x_dtype, y_dtype, z_dtype, n, part);
        hipLaunchKernelGGL(splat_axes_final_kernel, dim3(1), dim3(kWave), 0, st, part, n, axes);
    } else {
        hipLaunchKernelGGL(splat_axes_kernel, dim3(1), dim3(kAxesThreads), 0, st, x, y, z, x_dtype, y_dtype, z_dtype, n, axes);
    }
    if (c0) {
        unsigned nb = (unsigned)((n + kBlock - 1) / kBlock); if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(splat_select_kernel, dim3(nb), dim3(kBlock), 0, st, x, y, z, x_dtype, y_dtype, z_dtype, n, axes, c0, c1);
    }
    return last_launch();
}

int diffus_splat_fwd(const float *c0, const float *c1, const float *val, int P, long n, int cols, int H, int W,
                     float sigma, float *out, int *winner_keep, void *workspace, size_t workspace_bytes,
                     diffus_stream_t stream)
{
    if (!c0 || !c1 || !val || !out || P <= 0 || n <= 0 || H <= 0 || W <= 0 || !(sigma > 0.f)) return DIFFUS_EINVAL;
    if (cols < 0 || (cols > 0 && n % cols != 0)) return DIFFUS_EINVAL;
    if (n > 0x7fffffffL || (long)H * W > 0x3fffffffL) return DIFFUS_EUNSUPPORTED;
    const int half = splat_half(sigma);
    if (half > kSplatMaxHalf) return DIFFUS_EUNSUPPORTED;
    if (!workspace || workspace_bytes < diffus_splat_workspace_bytes(P, H, W)) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)H * W;
    int *winner = winner_keep ? winner_keep : (int *)workspace; // kept for the backward when the caller gives it a home
    float *planes = (float *)((char *)workspace + align256(sizeof(int) * (size_t)P * hw));
    float *blurred = (float *)((char *)planes + align256(sizeof(float) * (size_t)P * 2 * hw));
    if (hipMemsetAsync(winner, 0xff, sizeof(int) * (size_t)P * hw, st) != hipSuccess) return DIFFUS_ELAUNCH;
    if (cols >= kPatchSteps / 2 && n / cols >= 2) { // a (rows, cols) grid of samples: privatised patches
        const int rows = (int)(n / cols);
        const int rgs = (rows + kPatchRays - 1) / kPatchRays, cgs = (cols + kPatchSteps - 1) / kPatchSteps;
        hipLaunchKernelGGL(splat_winner_patch_kernel, dim3((unsigned)(rgs * cgs), P), dim3(kBlock), 0, st, c0, c1, rows,
                           cols, H, W, winner, rgs, cgs);
    } else {
        unsigned nb = (unsigned)((n + kBlock - 1) / kBlock); if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(splat_winner_kernel, dim3(nb, P), dim3(kBlock), 0, st, c0, c1, n, H, W, winner);
    }
    if (half >= 1 && half <= kSplatFusedMaxHalf) // compose + both blurs + divide in one launch (half == 0, sigma < 1/3: the size-1 kernel of the general path)
        return launch_splat_fused<false>(winner, val, n, nullptr, out, H, W, half, sigma, P, st);
    unsigned pb = (unsigned)((hw + kBlock - 1) / kBlock); if (pb > 4096) pb = 4096;
    hipLaunchKernelGGL(splat_compose_kernel, dim3(pb, P), dim3(kBlock), 0, st, winner, val, n, hw, planes);
    if (launch_blur(planes, blurred, H, W, half, sigma, hw, hw, P * 2, st)) return DIFFUS_ELAUNCH;
    dim3 tiles((W + kSplatTile - 1) / kSplatTile, (H + kSplatTile - 1) / kSplatTile, P);
    hipLaunchKernelGGL(splat_divide_kernel<false>, tiles, dim3(kBlock), 0, st, blurred, nullptr, out, H, W);
    return last_launch();
}

int diffus_splat_bwd(const float *c0, const float *c1, int P, long n, int H, int W, float sigma, const float *gout,
                     float *gval, const int *winner_kept, void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    if (!c0 || !c1 || !gout || !gval || P <= 0 || n <= 0 || H <= 0 || W <= 0 || !(sigma > 0.f)) return DIFFUS_EINVAL;
    if (n > 0x7fffffffL || (long)H * W > 0x3fffffffL) return DIFFUS_EUNSUPPORTED;
    const int half = splat_half(sigma);
    if (half > kSplatMaxHalf) return DIFFUS_EUNSUPPORTED;
    if (!workspace || workspace_bytes < diffus_splat_workspace_bytes(P, H, W)) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)H * W;
    int *winner = (int *)workspace;
    float *planes = (float *)((char *)workspace + align256(sizeof(int) * (size_t)P * hw));
    float *blurred = (float *)((char *)planes + align256(sizeof(float) * (size_t)P * 2 * hw));
    (void)winner;
    unsigned nb = (unsigned)((n + kBlock - 1) / kBlock); if (nb > 4096) nb = 4096;
    if (winner_kept && half >= 1 && half <= kSplatFusedMaxHalf) {
        // the forward's winner raster says where samples landed: weight plane, its blur and q = gout^T / (bw + eps) in one
        // launch; then blur q and gather per sample
        int rc = launch_splat_fused<true>(winner_kept, nullptr, n, gout, planes, H, W, half, sigma, P, st);
        if (rc) return rc;
        if (launch_blur(planes, blurred, H, W, half, sigma, hw, hw, P, st)) return DIFFUS_ELAUNCH;
        hipLaunchKernelGGL(splat_gather_kernel, dim3(nb, P), dim3(kBlock), 0, st, c0, c1, n, H, W, blurred, gval);
        return last_launch();
    }
    // no raster kept: recompute the weight plane and its blur
    if (hipMemsetAsync(planes, 0, sizeof(float) * (size_t)P * 2 * hw, st) != hipSuccess) return DIFFUS_ELAUNCH;
    hipLaunchKernelGGL(splat_mark_kernel, dim3(nb, P), dim3(kBlock), 0, st, c0, c1, n, H, W, planes);
    // only the weight planes (planes[:, 1]) carry anything: blur those alone, into blurred[:, 1]
    if (launch_blur(planes + hw, blurred + hw, H, W, half, sigma, 2 * hw, 2 * hw, P, st)) return DIFFUS_ELAUNCH;
    // q = gout^T / (bw + eps) into planes[:, 0]; blur it into blurred[:, 0]; gather per sample
    dim3 tiles((W + kSplatTile - 1) / kSplatTile, (H + kSplatTile - 1) / kSplatTile, P);
    hipLaunchKernelGGL(splat_divide_kernel<true>, tiles, dim3(kBlock), 0, st, blurred, gout, planes, H, W);
    // planes is now (P,1,H,W) q; blur plane-wise into `blurred` viewed as (P,H,W)
    if (launch_blur(planes, blurred, H, W, half, sigma, hw, hw, P, st)) return DIFFUS_ELAUNCH;
    hipLaunchKernelGGL(splat_gather_kernel, dim3(nb, P), dim3(kBlock), 0, st, c0, c1, n, H, W, blurred, gval);
    return last_launch();
}

int diffus_loss_sumsq(const float *frame, int P, long n, float *loss, float *gframe, void *workspace,
                      size_t workspace_bytes, diffus_stream_t stream)
{
    if (!frame || !loss || P <= 0 || n <= 0) return DIFFUS_EINVAL;
    if (!workspace || workspace_bytes < (size_t)512 * P) return DIFFUS_EWORKSPACE;
    int *cnt = (int *)workspace;                      // P arrival counters (zero between calls), then the partials
    float *part = (float *)((char *)workspace + (size_t)64 * P);   // kLossSplit floats per pose
    hipLaunchKernelGGL(loss_sumsq_kernel, dim3(kLossSplit, P), dim3(kBlock), 0, (hipStream_t)stream, frame, part, cnt, loss,
                       gframe, n);
    return last_launch();
}

} // extern "C"
