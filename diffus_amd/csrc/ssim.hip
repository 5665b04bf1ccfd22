// ssim.hip -- the loss the reference's training notebook attaches after the scan conversion:
//     synth = (img - img.min()) / (img.max() - img.min() + 1e-8);   loss = 1 - piq.ssim(synth, real, data_range=1.0)
// (reference notebooks/[DEMO] Train MRI to Impedance MLP - GPU.ipynb cell 16, `UltrasoundSynthesisModel.loss`), forward and
// backward, fused into five small launches so that the whole training iteration fits one short hipGraph (as torch ops it is
// ~110 kernels, eight of them MIOpen convolutions of ~20 us).  SSIM = Wang, Bovik, Sheikh & Simoncelli, IEEE TIP 2004,
// with piq's defaults: 11 x 11 Gaussian window of sigma 1.5 ('valid' correlation), K1 = 0.01, K2 = 0.03, mean over the
// map.  piq is a third-party package (not in the reference's requirements.txt, not installed here): the formula is
// restated from the paper; examples/losses.py holds the same thing in plain torch and is what the tests compare with.
#include "diffus_host.hpp"

namespace {

constexpr int kSsimMaxWin = 15;
constexpr int kSsimTile = 16;
constexpr int kSsimPatch = kSsimTile + kSsimMaxWin - 1;

struct SsimArgs {
    const float *img, *ref;
    int H, W, Hm, Wm, win;
    float c1, c2;
    int normalise;
    int reuse_stats; // backward: stats[0..3] still hold the forward's min / max / tie counts of this very image
    float *stats; // [0] lo [1] hi [2] ties of lo [3] ties of hi; as ints: [8..23] arrival counters of the 16 block groups, [24] of the groups
    float *part;  // (3, nblk) per-block partial sums: SSIM map | dL/dlo | dL/dhi  (nblk = 16 x 16 tiles of the IMAGE)
    int nblk;
    float *loss;
    const float *gloss; // nullable: upstream gradient of the loss (device scalar)
    float *gmap;        // (3, Hm, Wm): dL/d mu_x, dL/d E[x^2], dL/d E[xy] per map position
    float *gimg;
    float w1d[kSsimMaxWin]; // normalised 1-D Gaussian: the 2-D window is its outer product
};

// lo / hi of the image and how many pixels attain them (torch's min() / max() backward shares the gradient equally
// among ties -- and a splat image has thousands of exact zeros); also clears the accumulators of the other kernels.
// One block; its time is the latency of its load chain.  Images of up to 64 Ki pixels (the notebook's 256 x 256) are read
// ONCE: sixteen independent 16-byte loads per thread, all in flight together, and the values stay in registers for the
// tie count (round 3 first walked the image twice, four dwords per trip: 12 us for 65 536 pixels; a launch costs ~5).
constexpr int kMmThreads = 1024, kMmVec = 16; // 1024 threads x 16 float4
__global__ __launch_bounds__(kMmThreads) void ssim_minmax_kernel(SsimArgs A)
{
    __shared__ float s_lo[16], s_hi[16];
    __shared__ int s_cl[16], s_ch[16];
    __shared__ float b_lo, b_hi;
    const long n = (long)A.H * A.W;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool in_regs = n <= (long)kMmThreads * kMmVec * 4 && (reinterpret_cast<size_t>(A.img) & 15) == 0;
    float4 keep[kMmVec];
    float lo = __builtin_inff(), hi = -__builtin_inff();
    bool nan = false;
    const long n4 = n & ~3L;
    auto see = [&](float v) {
        nan |= v != v;
        lo = fminf(lo, v);
        hi = fmaxf(hi, v);
    };
    if (in_regs) {
        const float4 *img4 = reinterpret_cast<const float4 *>(A.img);
#pragma unroll
        for (int k = 0; k < kMmVec; ++k) {
            const long i = 4L * (threadIdx.x + (long)k * kMmThreads);
            // outside the image: a copy of pixel 0..3 region is not available in general -- use NaN-free neutral values
            keep[k] = (i < n4) ? img4[i >> 2] : make_float4(__builtin_inff(), __builtin_inff(), -__builtin_inff(), -__builtin_inff());
        }
#pragma unroll
        for (int k = 0; k < kMmVec; ++k) {
            const long i = 4L * (threadIdx.x + (long)k * kMmThreads);
            if (i < n4) { see(keep[k].x); see(keep[k].y); see(keep[k].z); see(keep[k].w); }
        }
    } else {
        for (long i = 4L * threadIdx.x; i < n4; i += 4L * kMmThreads) {
            see(A.img[i]); see(A.img[i + 1]); see(A.img[i + 2]); see(A.img[i + 3]);
        }
    }
    for (long i = n4 + threadIdx.x; i < n; i += kMmThreads) see(A.img[i]);
    if (nan) lo = hi = __builtin_nanf(""); // torch.min / max propagate NaN
    for (int o = 32; o > 0; o >>= 1) {
        const float l2 = __shfl_xor(lo, o), h2 = __shfl_xor(hi, o);
        lo = (lo != lo || l2 != l2) ? __builtin_nanf("") : fminf(lo, l2);
        hi = (hi != hi || h2 != h2) ? __builtin_nanf("") : fmaxf(hi, h2);
    }
    if (lane == 0) { s_lo[wv] = lo; s_hi[wv] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float l = s_lo[0], h = s_hi[0];
        for (int w = 1; w < 16; ++w) {
            l = (l != l || s_lo[w] != s_lo[w]) ? __builtin_nanf("") : fminf(l, s_lo[w]);
            h = (h != h || s_hi[w] != s_hi[w]) ? __builtin_nanf("") : fmaxf(h, s_hi[w]);
        }
        b_lo = l; b_hi = h;
    }
    __syncthreads();
    lo = b_lo; hi = b_hi;
    int cl = 0, ch = 0;
    auto count = [&](float v) {
        cl += v == lo;
        ch += v == hi;
    };
    if (in_regs) {
#pragma unroll
        for (int k = 0; k < kMmVec; ++k) {
            const long i = 4L * (threadIdx.x + (long)k * kMmThreads);
            if (i < n4) { count(keep[k].x); count(keep[k].y); count(keep[k].z); count(keep[k].w); }
        }
    } else {
        for (long i = 4L * threadIdx.x; i < n4; i += 4L * kMmThreads) {
            count(A.img[i]); count(A.img[i + 1]); count(A.img[i + 2]); count(A.img[i + 3]);
        }
    }
    for (long i = n4 + threadIdx.x; i < n; i += kMmThreads) count(A.img[i]);
    for (int o = 32; o > 0; o >>= 1) { cl += __shfl_xor(cl, o); ch += __shfl_xor(ch, o); }
    if (lane == 0) { s_cl[wv] = cl; s_ch[wv] = ch; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int tl = 0, th = 0;
        for (int w = 0; w < 16; ++w) { tl += s_cl[w]; th += s_ch[w]; }
        A.stats[0] = lo; A.stats[1] = hi; A.stats[2] = (float)tl; A.stats[3] = (float)th;
    }
    if (threadIdx.x >= 8 && threadIdx.x <= 24) reinterpret_cast<int *>(A.stats)[threadIdx.x] = 0; // the arrival counters
}

__device__ __forceinline__ float ssim_x(const SsimArgs &A, float v, float lo, float den)
{
    return A.normalise ? __fdiv_rn(v - lo, den) : v;
}

// One map position per thread, 16 x 16 positions per block, the (16 + win - 1)^2 patches of x and y in LDS.
// GRAD = false: sum of the SSIM map (-> loss by the last block).  GRAD = true: the three partial derivatives per position.
// WIN: the window size at compile time (11 = piq's and the notebook's default: the taps then sit in SGPRs and the loops
// unroll), 0 = read it from the arguments (every weight access is then a scalar load from the argument block).
template <bool GRAD, int WIN>
__global__ __launch_bounds__(kSsimTile *kSsimTile) void ssim_map_kernel(SsimArgs A)
{
    const int win = WIN ? WIN : A.win;
    __shared__ float px[kSsimPatch][kSsimPatch + 1], py[kSsimPatch][kSsimPatch + 1];
    __shared__ float rows[5][kSsimPatch][kSsimTile + 1]; // the five row sums of every patch row at the tile's 16 columns
    __shared__ float s_part[4];
    const int tx = threadIdx.x % kSsimTile, ty = threadIdx.x / kSsimTile;
    const int r0 = blockIdx.y * kSsimTile, c0 = blockIdx.x * kSsimTile;
    const int pw = kSsimTile + win - 1;
    const float lo = A.normalise ? A.stats[0] : 0.f;
    const float den = A.normalise ? __fadd_rn(A.stats[1] - lo, 1e-8f) : 1.f;
    for (int e = threadIdx.x; e < pw * pw; e += kSsimTile * kSsimTile) {
        const int r = e / pw, c = e - r * pw, gr = r0 + r, gc = c0 + c;
        const bool in = gr < A.H && gc < A.W;
        px[r][c] = in ? ssim_x(A, A.img[(long)gr * A.W + gc], lo, den) : 0.f;
        py[r][c] = in ? A.ref[(long)gr * A.W + gc] : 0.f;
    }
    __syncthreads();
    // The window is an outer product: the row sums of a patch row serve the 11 map rows it lies under.  Round 3 first had
    // every thread form the row sums of its own 11 rows (605 multiply-adds and 242 LDS reads per map position, 15 us for a
    // 256 x 256 image); shared through LDS it is ~110 + 55, in the same order of operations (bit-identical).
    for (int e = threadIdx.x; e < pw * kSsimTile; e += kSsimTile * kSsimTile) {
        const int r = e / kSsimTile, c = e - r * kSsimTile;
        float rx = 0.f, ry = 0.f, rxx = 0.f, ryy = 0.f, rxy = 0.f;
#pragma unroll
        for (int j = 0; j < win; ++j) {
            const float w = A.w1d[j], x = px[r][c + j], y = py[r][c + j];
            rx = __builtin_fmaf(w, x, rx); ry = __builtin_fmaf(w, y, ry);
            rxx = __builtin_fmaf(w, x * x, rxx); ryy = __builtin_fmaf(w, y * y, ryy); rxy = __builtin_fmaf(w, x * y, rxy);
        }
        rows[0][r][c] = rx; rows[1][r][c] = ry; rows[2][r][c] = rxx; rows[3][r][c] = ryy; rows[4][r][c] = rxy;
    }
    __syncthreads();
    const int r = r0 + ty, c = c0 + tx;
    const bool live = r < A.Hm && c < A.Wm;
    float mx = 0.f, my = 0.f, exx = 0.f, eyy = 0.f, exy = 0.f;
#pragma unroll
    for (int i = 0; i < win; ++i) {
        const float w = A.w1d[i];
        mx = __builtin_fmaf(w, rows[0][ty + i][tx], mx); my = __builtin_fmaf(w, rows[1][ty + i][tx], my);
        exx = __builtin_fmaf(w, rows[2][ty + i][tx], exx); eyy = __builtin_fmaf(w, rows[3][ty + i][tx], eyy);
        exy = __builtin_fmaf(w, rows[4][ty + i][tx], exy);
    }
    const float sxx = exx - mx * mx, syy = eyy - my * my, sxy = exy - mx * my;
    const float a1 = 2.f * mx * my + A.c1, a2 = 2.f * sxy + A.c2, b1 = mx * mx + my * my + A.c1, b2 = sxx + syy + A.c2;
    const float ib = __fdiv_rn(1.f, b1 * b2);
    const float S = a1 * a2 * ib;
    if constexpr (!GRAD) {
        // Sum of the map, without float atomics (deterministic) and without 2 x 256 returning atomics on ONE address
        // (they serialise at the L2: ~5 us of this kernel): every block stores its partial; arrival is counted in 16
        // groups and then once per group; the block that arrives last adds the partials in block order.
        __shared__ int s_last;
        float v = live ? S : 0.f;
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = v;
        __syncthreads();
        const int nb = (int)(gridDim.x * gridDim.y), bid = (int)(blockIdx.y * gridDim.x + blockIdx.x);
        if (threadIdx.x == 0) {
            A.part[bid] = s_part[0] + s_part[1] + s_part[2] + s_part[3];
            __threadfence(); // (3.2 us of this kernel, measured by leaving it out; a finishing launch of its own would cost ~4.8)
            int *cnt = reinterpret_cast<int *>(A.stats) + 8;
            const int g = bid & 15, gsize = (nb - g + 15) >> 4, ngroups = nb < 16 ? nb : 16;
            int last = 0;
            if (atomicAdd(cnt + g, 1) == gsize - 1) {
                __threadfence();
                last = atomicAdd(cnt + 16, 1) == ngroups - 1;
            }
            s_last = last;
        }
        __syncthreads();
        if (s_last) { // block-uniform
            __threadfence();
            float t = 0.f;
            for (int b = threadIdx.x; b < nb; b += kSsimTile * kSsimTile) t += __builtin_nontemporal_load(A.part + b);
            for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = t;
            __syncthreads();
            if (threadIdx.x == 0)
                A.loss[0] = 1.f - ((s_part[0] + s_part[1]) + (s_part[2] + s_part[3])) / (float)((long)A.Hm * A.Wm);
        }
    } else {
        if (live) {
            // loss = 1 - mean(S): every position weighs -gloss / N
            const float k = -(A.gloss ? A.gloss[0] : 1.f) / (float)((long)A.Hm * A.Wm);
            const float dmx = (2.f * my * (a2 - a1)) * ib - 2.f * mx * S / b1 + 2.f * mx * S / b2;
            const float dexx = -S / b2;
            const float dexy = 2.f * a1 * ib;
            const long q = (long)r * A.Wm + c, hw = (long)A.Hm * A.Wm;
            A.gmap[q] = k * dmx;
            A.gmap[hw + q] = k * dexx;
            A.gmap[2 * hw + q] = k * dexy;
        }
    }
}

// dL/dx(p) = sum over the map positions q whose window covers p of w(p - q) (G_mu(q) + 2 x(p) G_xx(q) + y(p) G_xy(q)),
// then through the normalisation: dL/dimg = dL/dx / den, and the sums that go to the min / max pixels
template <int WIN>
__global__ __launch_bounds__(kSsimTile *kSsimTile) void ssim_back_kernel(SsimArgs A)
{
    const int win = WIN ? WIN : A.win;
    __shared__ float g[3][kSsimPatch][kSsimPatch + 1];
    __shared__ float urow[3][kSsimPatch][kSsimTile + 1]; // row sums, shared like in ssim_map_kernel
    __shared__ float s_a[4], s_b[4];
    const int tx = threadIdx.x % kSsimTile, ty = threadIdx.x / kSsimTile;
    const int r0 = blockIdx.y * kSsimTile, c0 = blockIdx.x * kSsimTile;
    const int pw = kSsimTile + win - 1, off = win - 1;
    const long hw = (long)A.Hm * A.Wm;
    for (int e = threadIdx.x; e < pw * pw; e += kSsimTile * kSsimTile) {
        const int r = e / pw, c = e - r * pw, qr = r0 + r - off, qc = c0 + c - off;
        const bool in = qr >= 0 && qc >= 0 && qr < A.Hm && qc < A.Wm;
        const long q = (long)qr * A.Wm + qc;
        g[0][r][c] = in ? A.gmap[q] : 0.f;
        g[1][r][c] = in ? A.gmap[hw + q] : 0.f;
        g[2][r][c] = in ? A.gmap[2 * hw + q] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < pw * kSsimTile; e += kSsimTile * kSsimTile) { // patch row r, tile column c
        const int r = e / kSsimTile, c = e - r * kSsimTile;
        float u0 = 0.f, u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int j = 0; j < win; ++j) { // q = p - (i, j): patch column c + off - j
            const float w = A.w1d[j];
            u0 = __builtin_fmaf(w, g[0][r][c + off - j], u0);
            u1 = __builtin_fmaf(w, g[1][r][c + off - j], u1);
            u2 = __builtin_fmaf(w, g[2][r][c + off - j], u2);
        }
        urow[0][r][c] = u0; urow[1][r][c] = u1; urow[2][r][c] = u2;
    }
    __syncthreads();
    const int r = r0 + ty, c = c0 + tx;
    const bool live = r < A.H && c < A.W;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < win; ++i) {
        const float w = A.w1d[i];
        t0 = __builtin_fmaf(w, urow[0][ty + off - i][tx], t0);
        t1 = __builtin_fmaf(w, urow[1][ty + off - i][tx], t1);
        t2 = __builtin_fmaf(w, urow[2][ty + off - i][tx], t2);
    }
    float da = 0.f, db = 0.f;
    if (live) {
        const long p = (long)r * A.W + c;
        const float lo = A.normalise ? A.stats[0] : 0.f;
        const float den = A.normalise ? __fadd_rn(A.stats[1] - lo, 1e-8f) : 1.f;
        const float x = ssim_x(A, A.img[p], lo, den), y = A.ref[p];
        const float gx = t0 + 2.f * x * t1 + y * t2;
        if (A.normalise) {
            const float s = __fdiv_rn(1.f, den);
            A.gimg[p] = gx * s;
            da = -gx * s * (1.f - x); // d x / d lo = -(1 - x) / den
            db = -gx * s * x;         // d x / d hi = -x / den
        } else {
            A.gimg[p] = gx;
        }
    }
    if (A.normalise) {
        for (int o = 32; o > 0; o >>= 1) { da += __shfl_xor(da, o); db += __shfl_xor(db, o); }
        if ((threadIdx.x & 63) == 0) { s_a[threadIdx.x >> 6] = da; s_b[threadIdx.x >> 6] = db; }
        __syncthreads();
        if (threadIdx.x == 0) { // per-block partials, summed in block order by ssim_ties_kernel: no atomics, deterministic
            const int bid = (int)(blockIdx.y * gridDim.x + blockIdx.x);
            A.part[A.nblk + bid] = s_a[0] + s_a[1] + s_a[2] + s_a[3];
            A.part[2 * A.nblk + bid] = s_b[0] + s_b[1] + s_b[2] + s_b[3];
        }
    }
}

// the gradient of img.min() / img.max(): shared equally by the pixels that attain them (torch: evenly_distribute_backward)
__global__ __launch_bounds__(kBlock) void ssim_ties_kernel(SsimArgs A)
{
    __shared__ float s_t[2][kWavesPerBlock];
    const long n = (long)A.H * A.W;
    const float lo = A.stats[0], hi = A.stats[1];
    float ta = 0.f, tb = 0.f; // dL/dlo, dL/dhi: the blocks' partials of ssim_back_kernel in block order (every block, the same sum)
    for (int b = threadIdx.x; b < A.nblk; b += kBlock) { ta += A.part[A.nblk + b]; tb += A.part[2 * A.nblk + b]; }
    for (int o = 32; o > 0; o >>= 1) { ta += __shfl_xor(ta, o); tb += __shfl_xor(tb, o); }
    if ((threadIdx.x & 63) == 0) { s_t[0][threadIdx.x >> 6] = ta; s_t[1][threadIdx.x >> 6] = tb; }
    __syncthreads();
    ta = tb = 0.f;
    for (int w = 0; w < kWavesPerBlock; ++w) { ta += s_t[0][w]; tb += s_t[1][w]; }
    const float glo = ta / A.stats[2], ghi = tb / A.stats[3];
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float v = A.img[i];
        float add = 0.f;
        if (v == lo) add += glo;
        if (v == hi) add += ghi;
        if (add != 0.f) A.gimg[i] += add;
    }
}

int ssim_setup(SsimArgs &A, const float *img, const float *ref, int H, int W, int normalise, int win, float sigma, float k1,
               float k2, void *workspace, size_t workspace_bytes)
{
    if (!img || !ref || H <= 0 || W <= 0 || win < 1 || win > kSsimMaxWin || !(win & 1) || !(sigma > 0.f)) return DIFFUS_EINVAL;
    if (H < win || W < win) return DIFFUS_EINVAL;
    A.img = img; A.ref = ref; A.H = H; A.W = W; A.Hm = H - win + 1; A.Wm = W - win + 1; A.win = win;
    A.c1 = k1 * k1; A.c2 = k2 * k2; A.normalise = normalise != 0;
    A.nblk = ((W + kSsimTile - 1) / kSsimTile) * ((H + kSsimTile - 1) / kSsimTile);
    const size_t head = align256(32 * sizeof(float)), parts = align256(sizeof(float) * 3 * (size_t)A.nblk);
    if (!workspace || workspace_bytes < head + parts + sizeof(float) * 3 * (size_t)A.Hm * A.Wm) return DIFFUS_EWORKSPACE;
    A.stats = (float *)workspace;
    A.part = (float *)((char *)workspace + head);
    A.gmap = (float *)((char *)workspace + head + parts);
    // normalised 2-D Gaussian exp(-(i^2 + j^2) / (2 sigma^2)) / sum = outer product of the normalised 1-D one
    double w[kSsimMaxWin], tot = 0.0;
    for (int i = 0; i < win; ++i) {
        const double c = i - (win - 1) / 2.0;
        w[i] = exp(-(c * c) / (2.0 * (double)sigma * sigma));
        tot += w[i];
    }
    for (int i = 0; i < kSsimMaxWin; ++i) A.w1d[i] = i < win ? (float)(w[i] / tot) : 0.f;
    return DIFFUS_OK;
}

} // namespace

extern "C" {

size_t diffus_ssim_workspace_bytes(int H, int W, int win)
{
    if (H <= 0 || W <= 0 || win < 1 || H < win || W < win) return 0;
    const size_t nblk = (size_t)((W + kSsimTile - 1) / kSsimTile) * ((H + kSsimTile - 1) / kSsimTile);
    return align256(32 * sizeof(float)) + align256(sizeof(float) * 3 * nblk) + align256(sizeof(float) * 3 * (size_t)(H - win + 1) * (W - win + 1));
}

int diffus_ssim_loss_fwd(const float *img, const float *ref, int H, int W, int normalise, int win, float sigma, float k1, float k2,
                         float *loss, void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    SsimArgs A{};
    int rc = ssim_setup(A, img, ref, H, W, normalise, win, sigma, k1, k2, workspace, workspace_bytes);
    if (rc) return rc;
    if (!loss) return DIFFUS_EINVAL;
    A.loss = loss;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ssim_minmax_kernel, dim3(1), dim3(kMmThreads), 0, st, A); // (also clears the accumulators: needed without normalisation too)
    dim3 grid((A.Wm + kSsimTile - 1) / kSsimTile, (A.Hm + kSsimTile - 1) / kSsimTile);
    if (win == 11)
        hipLaunchKernelGGL((ssim_map_kernel<false, 11>), grid, dim3(kSsimTile * kSsimTile), 0, st, A);
    else
        hipLaunchKernelGGL((ssim_map_kernel<false, 0>), grid, dim3(kSsimTile * kSsimTile), 0, st, A);
    return last_launch();
}

int diffus_ssim_loss_bwd(const float *img, const float *ref, int H, int W, int normalise, int win, float sigma, float k1, float k2,
                         const float *gloss, float *gimg, int reuse_stats, void *workspace, size_t workspace_bytes,
                         diffus_stream_t stream)
{
    SsimArgs A{};
    int rc = ssim_setup(A, img, ref, H, W, normalise, win, sigma, k1, k2, workspace, workspace_bytes);
    if (rc) return rc;
    if (!gimg) return DIFFUS_EINVAL;
    A.gloss = gloss; A.gimg = gimg;
    A.reuse_stats = (reuse_stats != 0) && A.normalise;
    hipStream_t st = (hipStream_t)stream;
    // min / max and the tie counts: recomputed, unless the caller vouches that `workspace` still holds the forward's
    if (A.normalise && !A.reuse_stats) hipLaunchKernelGGL(ssim_minmax_kernel, dim3(1), dim3(kMmThreads), 0, st, A);
    dim3 gm((A.Wm + kSsimTile - 1) / kSsimTile, (A.Hm + kSsimTile - 1) / kSsimTile);
    if (win == 11)
        hipLaunchKernelGGL((ssim_map_kernel<true, 11>), gm, dim3(kSsimTile * kSsimTile), 0, st, A);
    else
        hipLaunchKernelGGL((ssim_map_kernel<true, 0>), gm, dim3(kSsimTile * kSsimTile), 0, st, A);
    dim3 gi((W + kSsimTile - 1) / kSsimTile, (H + kSsimTile - 1) / kSsimTile);
    if (win == 11)
        hipLaunchKernelGGL(ssim_back_kernel<11>, gi, dim3(kSsimTile * kSsimTile), 0, st, A);
    else
        hipLaunchKernelGGL(ssim_back_kernel<0>, gi, dim3(kSsimTile * kSsimTile), 0, st, A);
    if (A.normalise) {
        unsigned nb = (unsigned)(((long)H * W + kBlock - 1) / kBlock); if (nb > 1024) nb = 1024;
        hipLaunchKernelGGL(ssim_ties_kernel, dim3(nb), dim3(kBlock), 0, st, A);
    }
    return last_launch();
}

} // extern "C"
