// impedance.hip -- the MRI -> acoustic-impedance stage that feeds the renderer:
//   ImpedanceEstimator.forward (+ autograd)          reference src/impedance.py:6-17     1 -> 32 -> 32 -> 1 MLP, ReLU
//   create_brain_mask                                reference src/utils.py:12-21        threshold, dilate x2, erode x2
//   zscore_normalize (mean / unbiased std over mask)  reference src/utils.py:23-39
//   ImpedanceEstimator.compute_impedance_volume       reference src/impedance.py:38-53    = the three above, fused
//
// The MLP is the one GEMM-shaped piece of the whole package: per voxel 32 + 32x32 + 32 MACs, 16.7 M voxels.  The
// hidden layer runs on the f32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMA chains, 157 TFLOP/s peak),
// activations never leave registers:
//   tile = 32 voxels.  D[k][v] = sum_j W2[k][j] h1[v][j]  as 16 MFMAs with A = W2 (held in 16 VGPRs for the whole
//   kernel) and B = h1, computed on the fly in exactly the lane layout the B operand wants.
//   An accumulator register i of lane l (r = l & 31, h = l >> 5) holds row idx(i,h) = 8(i>>2) + 4h + (i&3), column r.
//   Every contraction index is enumerated in that same (i,h) order, so accumulators feed the next MFMA directly.
// Backward recomputes the forward twice -- D (lane = voxel) and, by swapping the MFMA operands, its transpose
// D' (lane = hidden unit) -- which gives dL/dh2 in both operand layouts without an LDS transpose; then
//   dW2 += G2 H1   (contraction over the tile's voxels)      G1 = W2^T G2   (contraction over hidden units)
// 64 MFMAs per 32 voxels.  Parameter gradients accumulate in registers over the wave's tiles, are combined per block
// in a fixed order, and reduced over blocks by a second kernel: deterministic.
#include "diffus_host.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kH = DIFFUS_MLP_HIDDEN;           // 32
constexpr int kNP = DIFFUS_MLP_PARAMS;          // 1153
constexpr int oW1 = 0, oB1 = 32, oW2 = 64, oB2 = 64 + 1024, oW3 = oB2 + 32, oB3 = oW3 + 32;
constexpr int kMlpMaxBlocks = 2048;             // bound of the persistent grids (sizes the backward's workspace)

__device__ __forceinline__ int hidx(int i, int h) { return 8 * (i >> 2) + 4 * h + (i & 3); }

struct MlpArgs {
    const float *x;
    const unsigned char *mask; // nullable
    size_t n;
    const float *params;
    float shift, div, out_scale, fill;
    float *y;
    const float *gy;
    size_t y_stride, gy_stride; // elements between consecutive outputs / upstream gradients (1 = contiguous)
    float *gx;      // nullable
    float *partial; // (gridDim.x, kNP)
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// y = mask ? out_scale * mlp((x - shift) / div) : fill.   One wave = 64 consecutive voxels = two tiles per trip.
#ifndef DIFFUS_MLP_FWD_WAVES // 3 waves per SIMD: 134 VGPRs, no scratch, 0.363 ms at 256^3; 4 forced 128 VGPRs + 28 B of scratch: 0.382 ms
#define DIFFUS_MLP_FWD_WAVES 3
#endif
__global__ __launch_bounds__(kBlock, DIFFUS_MLP_FWD_WAVES) void mlp_fwd_kernel(MlpArgs A)
{
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float *P = A.params;
    float aW2[16], w1h[16], b1h[16], b2h[16], w3h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = hidx(i, h);
        aW2[i] = P[oW2 + r * kH + j]; // A operand: W2[k = r][j = idx(i,h)]
        w1h[i] = P[oW1 + j];
        b1h[i] = P[oB1 + j];
        b2h[i] = P[oB2 + j];
        w3h[i] = P[oW3 + j];
    }
    const float b3 = P[oB3];
    const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock, wave = (size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const size_t ngroups = (A.n + 63) / 64;
    // software prefetch: the next trip's voxels are in flight while this trip's 32 MFMAs run.  Branch-free loads
    // (index clamped, mask and x fetched together): a load behind a branch on another load serialises two round trips.
    const bool has_mask = A.mask != nullptr;
    auto fetch = [&](size_t g, float &xv, unsigned &mk) {
        size_t i0 = g * 64 + lane;
        i0 = i0 < A.n ? i0 : A.n - 1;
        xv = A.x[i0];
        mk = has_mask ? A.mask[i0] : 1u;
    };
    float xn;
    unsigned mkn;
    fetch(wave, xn, mkn);
    for (size_t g = wave; g < ngroups; g += nwaves) {
        const size_t i0 = g * 64 + lane;
        const bool in = i0 < A.n, live = in && mkn;
        float xv = live ? xn : A.shift;
        fetch(g + nwaves, xn, mkn);
        if (__ballot(live) == 0) { // all air: nothing to evaluate (most of a head volume)
            if (in) A.y[i0 * A.y_stride] = A.fill;
            continue;
        }
        xv = __fdiv_rn(xv - A.shift, A.div); // zscore_normalize, reference src/utils.py:38
        const float x0 = __shfl(xv, r, kWave), x1 = __shfl(xv, 32 + r, kWave);
        f32x16 D0, D1; // accumulators start at the bias b2 (row idx(i,h) lives in register i): no add afterwards
#pragma unroll
        for (int i = 0; i < 16; ++i) D0[i] = D1[i] = b2h[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float hb0 = fmaxf(__builtin_fmaf(w1h[i], x0, b1h[i]), 0.f); // B operand: h1[voxel r][j = idx(i,h)]
            const float hb1 = fmaxf(__builtin_fmaf(w1h[i], x1, b1h[i]), 0.f);
            D0 = mfma(aW2[i], hb0, D0);
            D1 = mfma(aW2[i], hb1, D1);
        }
        float s0 = 0.f, s1 = 0.f; // D[i] = pre-activation of hidden unit idx(i,h) at voxel r
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s0 = __builtin_fmaf(w3h[i], fmaxf(D0[i], 0.f), s0);
            s1 = __builtin_fmaf(w3h[i], fmaxf(D1[i], 0.f), s1);
        }
        s0 += __shfl_xor(s0, 32, kWave);
        s1 += __shfl_xor(s1, 32, kWave);
        const float yv = ((h ? s1 : s0) + b3) * A.out_scale;
        if (in) A.y[i0 * A.y_stride] = live ? yv : A.fill;
    }
}

// Backward of y = out_scale * mlp((x - shift)/div) for upstream gy: parameter gradients (per-block partials) and,
// if asked, gx.  One tile (32 voxels) per trip.
__global__ __launch_bounds__(kBlock) void mlp_bwd_kernel(MlpArgs A)
{
    __shared__ float sm[kWavesPerBlock][kNP];
    const int wib = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float *P = A.params;
    float aW2[16], aW2T[16], w1h[16], b1h[16], b2h[16], w3h[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int j = hidx(i, h);
        aW2[i] = P[oW2 + r * kH + j];  // W2[k = r][j = idx(i,h)]
        aW2T[i] = P[oW2 + j * kH + r]; // W2[k = idx(i,h)][j = r]
        w1h[i] = P[oW1 + j];
        b1h[i] = P[oB1 + j];
        b2h[i] = P[oB2 + j];
        w3h[i] = P[oW3 + j];
    }
    const float w1l = P[oW1 + r], b1l = P[oB1 + r], b2l = P[oB2 + r], w3l = P[oW3 + r];

    f32x16 dW2 = {0};                 // [k = idx(i,h)][j = r]
    float aw1[16], ab1[16];           // per-lane (voxel) partial sums for hidden unit idx(i,h)
#pragma unroll
    for (int i = 0; i < 16; ++i) aw1[i] = ab1[i] = 0.f;
    float aw3 = 0.f, ab2 = 0.f, ab3 = 0.f; // lane = hidden unit r (both halves hold partial sums)

    const size_t nwaves = (size_t)gridDim.x * kWavesPerBlock, wave = (size_t)blockIdx.x * kWavesPerBlock + wib;
    const size_t ntiles = (A.n + 31) / 32;
    // One wave per SIMD (the accumulators fill the register file), so nothing hides a load but the wave itself:
    // the next tile's x / gy / mask are fetched (branch-free, index clamped) before this tile's 64 MFMAs start.
    const bool has_mask = A.mask != nullptr;
    auto fetch = [&](size_t t, float &xv, float &gv, unsigned &mk) {
        size_t i = t * 32 + r;
        i = i < A.n ? i : A.n - 1;
        xv = A.x[i];
        gv = A.gy[i * A.gy_stride];
        mk = has_mask ? A.mask[i] : 1u;
    };
    float xnx, gnx;
    unsigned mnx;
    fetch(wave, xnx, gnx, mnx);
    for (size_t t = wave; t < ntiles; t += nwaves) {
        const size_t base = t * 32;
        // lane = voxel view
        const size_t il = base + r;
        const bool livel = il < A.n && mnx;
        const float gyl = livel ? gnx * A.out_scale : 0.f;
        const float xraw = livel ? xnx : A.shift;
        fetch(t + nwaves, xnx, gnx, mnx);
        if (__ballot(gyl != 0.f) == 0) { // nothing flows back through this tile
            if (A.gx && h == 0 && il < A.n) A.gx[il] = 0.f;
            continue;
        }
        const float xl = __fdiv_rn(xraw - A.shift, A.div);
        // (register, half) = voxel view of the same 32 voxels: lane idx(i,h) holds that voxel
        float xq[16], gq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            xq[i] = __shfl(xl, hidx(i, h), kWave);
            gq[i] = __shfl(gyl, hidx(i, h), kWave);
        }
        // forward, both orientations
        float hb[16];
        f32x16 D, Dt; // start at the bias: row k = idx(i,h) of D in register i, column k = r of Dt in every register
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            D[i] = b2h[i];
            Dt[i] = b2l;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            hb[i] = fmaxf(__builtin_fmaf(w1h[i], xl, b1h[i]), 0.f); // h1[voxel r][j = idx(i,h)]
            D = mfma(aW2[i], hb[i], D);                             // D[k = idx(.,h)][voxel r]
            Dt = mfma(hb[i], aW2[i], Dt);                           // Dt[voxel idx(.,h)][k = r]
        }
        // dL/d(pre-activation 2) in both layouts; W3, b2 gradients from the transposed one
        float g2[16], g2q[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            g2[i] = (D[i] > 0.f) ? w3h[i] * gyl : 0.f;
            const float a2 = Dt[i];
            g2q[i] = (a2 > 0.f) ? w3l * gq[i] : 0.f;
            aw3 = __builtin_fmaf(fmaxf(a2, 0.f), gq[i], aw3);
            ab2 += g2q[i];
        }
        if (h == 0) ab3 += gyl;
        // dW2[k][j] += sum_v g2[k][v] h1[v][j];   G1[j][v] = sum_k W2[k][j] g2[k][v]
        f32x16 G1 = {0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float h1q = fmaxf(__builtin_fmaf(w1l, xq[i], b1l), 0.f); // h1[voxel idx(i,h)][j = r]
            dW2 = mfma(g2q[i], h1q, dW2);
            G1 = mfma(aW2T[i], g2[i], G1);                                 // G1[j = idx(.,h)][voxel r]
        }
        float gxs = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float g1 = (hb[i] > 0.f) ? G1[i] : 0.f;
            aw1[i] = __builtin_fmaf(g1, xl, aw1[i]);
            ab1[i] += g1;
            gxs = __builtin_fmaf(w1h[i], g1, gxs);
        }
        if (A.gx) {
            gxs += __shfl_xor(gxs, 32, kWave);
            if (h == 0 && il < A.n) A.gx[il] = __fdiv_rn(gxs, A.div);
        }
    }

    // ---- per-wave results -> LDS, waves combined in a fixed order -> this block's partial ----
    float *my = sm[wib];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float a = aw1[i], b = ab1[i];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) { // over the 32 voxels of the lane half
            a += __shfl_xor(a, off, kWave);
            b += __shfl_xor(b, off, kWave);
        }
        if (r == 0) {
            my[oW1 + hidx(i, h)] = a;
            my[oB1 + hidx(i, h)] = b;
        }
        my[oW2 + hidx(i, h) * kH + r] = dW2[i];
    }
    aw3 += __shfl_xor(aw3, 32, kWave);
    ab2 += __shfl_xor(ab2, 32, kWave);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ab3 += __shfl_xor(ab3, off, kWave);
    if (h == 0) {
        my[oW3 + r] = aw3;
        my[oB2 + r] = ab2;
    }
    if (lane == 0) my[oB3] = ab3;
    __syncthreads();
    for (int q = threadIdx.x; q < kNP; q += kBlock) {
        float s = sm[0][q];
#pragma unroll
        for (int wv = 1; wv < kWavesPerBlock; ++wv) s += sm[wv][q];
        A.partial[(size_t)blockIdx.x * kNP + q] = s;
    }
}

// gparams[q] = sum over blocks of partial[b][q], in a fixed order (deterministic).  A block takes 16 parameters; its 256
// threads are 16 slices of the block list x 16 parameters, each slice summed serially, the 16 slice sums added in
// order.  (One thread per parameter walking all ~1000 partial blocks was a chain of dependent loads: 61 us -- for a
// 256 x 256 slice more than the rest of a whole training iteration.)
constexpr int kRedParams = 16, kRedSlices = kBlock / kRedParams;
__global__ __launch_bounds__(kBlock) void mlp_reduce_kernel(const float *__restrict__ partial, int nblk, float *__restrict__ gparams)
{
    __shared__ float sm[kRedSlices][kRedParams];
    const int p = threadIdx.x % kRedParams, sl = threadIdx.x / kRedParams;
    const int q = blockIdx.x * kRedParams + p;
    float s = 0.f;
    if (q < kNP)
        for (int b = sl; b < nblk; b += kRedSlices) s += partial[(size_t)b * kNP + q];
    sm[sl][p] = s;
    __syncthreads();
    if (sl == 0 && q < kNP) {
        float t = sm[0][p];
#pragma unroll
        for (int k = 1; k < kRedSlices; ++k) t += sm[k][p];
        gparams[q] = t;
    }
}

// ---- create_brain_mask: threshold, then 6-neighbourhood dilations / erosions (outside = 0, like SciPy) ----
__global__ __launch_bounds__(kBlock) void threshold_kernel(const float *__restrict__ vol, size_t n, float thr,
                                                           unsigned char *__restrict__ m)
{
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) m[i] = vol[i] > thr;
}

template <bool DILATE>
__global__ __launch_bounds__(kBlock) void morph_kernel(const unsigned char *__restrict__ in, unsigned char *__restrict__ out,
                                                       int d0, int d1, int d2)
{
    const size_t n = (size_t)d0 * d1 * d2;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        const int c = (int)(i % d2), b = (int)((i / d2) % d1), a = (int)(i / ((size_t)d1 * d2));
        const size_t s1 = d2, s0 = (size_t)d1 * d2;
        auto at = [&](bool inside, size_t j) -> unsigned { return inside ? in[j] : 0u; };
        unsigned v = in[i];
        unsigned nb[6] = {at(a > 0, i - s0), at(a + 1 < d0, i + s0), at(b > 0, i - s1),
                          at(b + 1 < d1, i + s1), at(c > 0, i - 1), at(c + 1 < d2, i + 1)};
#pragma unroll
        for (int q = 0; q < 6; ++q) v = DILATE ? (v | nb[q]) : (v & nb[q]);
        out[i] = (unsigned char)v;
    }
}

// ---- masked mean / unbiased std (zscore_normalize): f64 sums, fixed-order two-stage reduction ----
constexpr int kStatBlocks = 256;
__global__ __launch_bounds__(kBlock) void stats_partial_kernel(const float *__restrict__ vol, const unsigned char *__restrict__ m,
                                                               size_t n, double *__restrict__ part)
{
    __shared__ double sh[3][kBlock];
    double s = 0.0, ss = 0.0, c = 0.0;
    for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (size_t)gridDim.x * kBlock) {
        if (!m || m[i]) {
            const double v = vol[i];
            s += v; ss += v * v; c += 1.0;
        }
    }
    sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = ss; sh[2][threadIdx.x] = c;
    __syncthreads();
    for (int st = kBlock / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int q = 0; q < 3; ++q) sh[q][threadIdx.x] += sh[q][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 3) part[blockIdx.x * 3 + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ void stats_finish_kernel(const double *__restrict__ part, int nblk, double *__restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    double s = 0.0, ss = 0.0, c = 0.0;
    for (int b = 0; b < nblk; ++b) { s += part[b * 3]; ss += part[b * 3 + 1]; c += part[b * 3 + 2]; }
    const double mean = s / c;
    const double var = (ss - s * mean) / (c - 1.0); // unbiased, like torch.std()
    out[0] = mean;
    out[1] = sqrt(var > 0.0 ? var : 0.0);
    out[2] = c;
}

// Persistent grids are sized to what is resident at once (blocks per CU from the occupancy query x CUs): a grid
// larger than that runs in rounds and the last round leaves most of the chip idle.
template <typename K>
unsigned resident_blocks(K kernel)
{
    int dev = 0, cus = 256, per_cu = 1;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, 0) != hipSuccess || per_cu < 1) per_cu = 1;
    long nb = (long)cus * per_cu;
    return (unsigned)(nb < kMlpMaxBlocks ? nb : kMlpMaxBlocks);
}

unsigned grid_for(size_t n)
{
    size_t nb = (n + kBlock - 1) / kBlock;
    return (unsigned)(nb < 4096 ? (nb ? nb : 1) : 4096);
}

} // namespace

extern "C" {

int diffus_mlp_fwd(const float *x, const unsigned char *mask, size_t n, const float *params, float in_shift, float in_div,
                   float out_scale, float fill, float *y, size_t y_stride, diffus_stream_t stream)
{
    if (!x || !params || !y || n == 0 || y_stride == 0) return DIFFUS_EINVAL;
    if (!(in_div != 0.f)) return DIFFUS_EINVAL;
    MlpArgs A{};
    A.x = x; A.mask = mask; A.n = n; A.params = params;
    A.shift = in_shift; A.div = in_div; A.out_scale = out_scale; A.fill = fill; A.y = y; A.y_stride = y_stride;
    const size_t ngroups = (n + 63) / 64;
    static const unsigned resident = resident_blocks(mlp_fwd_kernel);
    const size_t want = (ngroups + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned nblk = (unsigned)(want < resident ? want : resident);
    hipLaunchKernelGGL(mlp_fwd_kernel, dim3(nblk), dim3(kBlock), 0, (hipStream_t)stream, A);
    return last_launch();
}

size_t diffus_mlp_workspace_bytes(void) { return align256(sizeof(float) * (size_t)kMlpMaxBlocks * kNP); }

int diffus_mlp_bwd(const float *x, const unsigned char *mask, size_t n, const float *params, float in_shift, float in_div,
                   float out_scale, const float *gy, size_t gy_stride, float *gparams, float *gx, void *workspace,
                   size_t workspace_bytes, diffus_stream_t stream)
{
    if (!x || !params || !gy || !gparams || n == 0 || gy_stride == 0) return DIFFUS_EINVAL;
    if (!(in_div != 0.f)) return DIFFUS_EINVAL;
    if (!workspace || workspace_bytes < diffus_mlp_workspace_bytes()) return DIFFUS_EWORKSPACE;
    MlpArgs A{};
    A.x = x; A.mask = mask; A.n = n; A.params = params;
    A.shift = in_shift; A.div = in_div; A.out_scale = out_scale;
    A.gy = gy; A.gy_stride = gy_stride; A.gx = gx; A.partial = (float *)workspace;
    const size_t ntiles = (n + 31) / 32;
    static const unsigned resident = resident_blocks(mlp_bwd_kernel);
    const size_t want = (ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned nblk = (unsigned)(want < resident ? want : resident);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(mlp_bwd_kernel, dim3(nblk), dim3(kBlock), 0, st, A);
    if (hipGetLastError() != hipSuccess) return DIFFUS_ELAUNCH;
    hipLaunchKernelGGL(mlp_reduce_kernel, dim3((kNP + kRedParams - 1) / kRedParams), dim3(kBlock), 0, st, A.partial, (int)nblk, gparams);
    return last_launch();
}

size_t diffus_brain_mask_workspace_bytes(int d0, int d1, int d2)
{
    if (d0 <= 0 || d1 <= 0 || d2 <= 0) return 0;
    return align256((size_t)d0 * d1 * d2);
}

int diffus_brain_mask(const float *vol, int d0, int d1, int d2, float threshold, int iterations, unsigned char *mask,
                      void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    if (!vol || !mask || d0 <= 0 || d1 <= 0 || d2 <= 0 || iterations < 0) return DIFFUS_EINVAL;
    if (iterations > 0 && (!workspace || workspace_bytes < diffus_brain_mask_workspace_bytes(d0, d1, d2))) return DIFFUS_EWORKSPACE;
    const size_t n = (size_t)d0 * d1 * d2;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nb = grid_for(n);
    unsigned char *a = mask, *b = (unsigned char *)workspace;
    // 2*iterations passes ping-pong between the two buffers; start so that the last pass lands in `mask`
    hipLaunchKernelGGL(threshold_kernel, dim3(nb), dim3(kBlock), 0, st, vol, n, threshold, a);
    for (int pass = 0; pass < 2 * iterations; ++pass) {
        if (pass < iterations)
            hipLaunchKernelGGL(morph_kernel<true>, dim3(nb), dim3(kBlock), 0, st, a, b, d0, d1, d2);
        else
            hipLaunchKernelGGL(morph_kernel<false>, dim3(nb), dim3(kBlock), 0, st, a, b, d0, d1, d2);
        unsigned char *t = a; a = b; b = t;
    }
    (void)b; // 2*iterations is even: the result is back in `mask`
    return last_launch();
}

size_t diffus_masked_stats_workspace_bytes(void) { return align256(sizeof(double) * 3 * kStatBlocks); }

int diffus_masked_stats(const float *vol, const unsigned char *mask, size_t n, double *out, void *workspace,
                        size_t workspace_bytes, diffus_stream_t stream)
{
    if (!vol || !out || n == 0) return DIFFUS_EINVAL;
    if (!workspace || workspace_bytes < diffus_masked_stats_workspace_bytes()) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(stats_partial_kernel, dim3(kStatBlocks), dim3(kBlock), 0, st, vol, mask, n, (double *)workspace);
    if (hipGetLastError() != hipSuccess) return DIFFUS_ELAUNCH;
    hipLaunchKernelGGL(stats_finish_kernel, dim3(1), dim3(64), 0, st, (const double *)workspace, kStatBlocks, out);
    return last_launch();
}

} // extern "C"
