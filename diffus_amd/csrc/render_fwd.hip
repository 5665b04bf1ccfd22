// render_fwd.hip -- forward kernel (plot_beam_frame), the stage-wise kernels, and their C-ABI entry points
#include "diffus_host.hpp"

namespace {

#ifdef DIFFUS_STAMP // diagnostic build only (tools/fwd_stamps.py): per-wave phase timestamps of the forward kernel
__device__ unsigned long long *g_fwd_stamps = nullptr;
#define STAMPW(i)                                                                                             \
    do {                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        if (g_fwd_stamps && lane == 0) g_fwd_stamps[(size_t)w * 8 + (i)] = __builtin_readcyclecounter();      \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
    } while (0)
#else
#define STAMPW(i) ((void)0)
#endif

// ----------------------------------------------------------------------------
// FORWARD  (replaces reference src/renderer.py:201-275 with artifacts=False)
#ifndef DIFFUS_FWD_MIN_WAVES
#define DIFFUS_FWD_MIN_WAVES 1
#endif
// SEG = true: the launch covers one 1024-sample segment of a longer ray (carries in the workspace); only
// instantiated for C = 16.  SEG = false compiles every carry path away.
// SPLIT = 2: the two waves of a 128-thread block take the two halves of one ray (see render_bwd_kernel); the first
// half hands its last impedance sample and its total transfer-matrix product to the second through LDS.
template <int C, int SAMPLER, int LAYOUT, int WPB, int PM, bool SEG = false, int SPLIT = 1>
__global__ __launch_bounds__(kWave *WPB, (C <= 8 ? DIFFUS_FWD_MIN_WAVES : 1)) void render_fwd_kernel(Args A)
{
    static_assert(SPLIT == 1 || (SPLIT == 2 && WPB == 2 && !SEG), "SPLIT: one ray per block of two waves");
    __shared__ float s_c[5]; // SPLIT exchange: total product of the first half (4) + its last sample
    const float *const cin = SEG ? A.cin : nullptr;
    float *const cout = SEG ? A.cout : nullptr;
    __shared__ __attribute__((aligned(16))) float lds[WPB][kWave * C];
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: ray-derived addresses stay scalar
    const int part = (SPLIT > 1) ? wib : 0;
    const long w = (SPLIT > 1) ? (long)xcd_remap(blockIdx.x, gridDim.x) : (long)xcd_remap(blockIdx.x, gridDim.x) * WPB + wib;
    if (w >= (long)A.P * A.R) return; // wave-uniform; SPLIT: both waves of the block leave together
    const int seg0 = SEG ? A.seg0 : part * (kWave * C);
    const int segN = SEG ? A.segN : ((SPLIT > 1) ? min(A.N1 - seg0, kWave * C) : A.N1);
    const int lane = threadIdx.x & 63;
    const long pose = w / A.R;
    const int n0 = lane * C;
    float *wb = lds[wib];

    STAMPW(0);
    Pose ps;
    load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);

    float zi[C], z[C], r[C], e[C], u0[C], u1[C], u2[C];
#ifdef DIFFUS_ABLATE_GATHER
#pragma unroll
    for (int j = 0; j < C; ++j) zi[j] = ps.sf[0] + (float)(j * kWave + lane) * ps.df[1];
#else
    gather_interleaved<C, SAMPLER, LAYOUT, false, PM>(A, seg0, segN, ps, lane, zi, u0, u1, u2);
#endif
    STAMPW(1);
#ifdef DIFFUS_ABLATE_TRANSPOSE
#pragma unroll
    for (int j = 0; j < C; ++j) z[j] = zi[j];
#else
    to_chunked<C>(wb, lane, zi, z);
#endif
    STAMPW(2);
    float zprev = lane_prev(z[C - 1], z[C - 1]); // last sample of the lane below (lane 0: unused unless a carry comes in)
    if (SPLIT > 1) {
        if (part == 0 && lane == kWave - 1) s_c[4] = z[C - 1];
        __syncthreads();
        if (part == 1 && lane == 0) zprev = s_c[4];
    }
    Mat K = mat_identity(), Klast = mat_identity();
    if (cin) { // carry of the earlier segments: running product and the sample just before this segment
        K = Mat{cin[w * 5 + 0], cin[w * 5 + 1], cin[w * 5 + 2], cin[w * 5 + 3]};
        if (lane == 0) zprev = cin[w * 5 + 4];
    }
    float medv = (A.start > 0) ? A.med[pose] : 0.f;
    reflect_chunk<C>(A, seg0, segN, n0, z, zprev, medv, r);
    STAMPW(3);
#ifdef DIFFUS_ABLATE_SCAN
#pragma unroll
    for (int j = 0; j < C; ++j) e[j] = r[j];
#else
    if (SPLIT > 1) {
        auto exchange = [&](const Mat &Lincl, Mat &carry) -> bool {
            if (part == 0 && lane == kWave - 1) {
                s_c[0] = Lincl.a; s_c[1] = Lincl.b; s_c[2] = Lincl.c; s_c[3] = Lincl.d;
            }
            __syncthreads();
            if (part == 0) return false;
            carry = Mat{s_c[0], s_c[1], s_c[2], s_c[3]};
            return true;
        };
        echo_chunk<C, true>(r, lane, e, nullptr, -1, nullptr, exchange);
    } else {
        echo_chunk<C, true>(r, lane, e, cin ? &K : nullptr, segN - 1, cout ? &Klast : nullptr);
        // an ill-conditioned ray (wave-uniform, rare): the same scan in float64 (diffus_device.hpp).  A ray of one launch: in place; a
        // ray of several (SEG) is flagged and walked again from its first sample by render_fwd_long_repair_kernel, its running
        // product carried in float64 (a float32 carry would bring its own rounding times the ray's condition number along)
        if (__builtin_expect(echo_needs_f64<C>(e), 0)) {
            if (!SEG) echo_f64_rare<C, SAMPLER, LAYOUT, PM>(A, ps, seg0, segN, n0, medv, e);
            else if (A.rflag && lane == 0) A.rflag[w * 2] = 1; // a ray of several launches: render_fwd_long_repair_kernel walks it again
        }
    }
#endif
    if (cout) { // hand the running product and the last impedance sample to the next segment
        const int last = segN - 1;
        if (lane == last / C) {
            float zl = z[0];
#pragma unroll
            for (int j = 1; j < C; ++j) zl = (j == last % C) ? z[j] : zl;
            cout[w * 5 + 0] = Klast.a; cout[w * 5 + 1] = Klast.b; cout[w * 5 + 2] = Klast.c; cout[w * 5 + 3] = Klast.d;
            cout[w * 5 + 4] = zl;
        }
    }
    if (SEG && !A.frame) return; // carry-only pass (first half of a segmented backward)
    STAMPW(4);
    {   // attenuation, reference :256-259: exp(-alpha * n) (chunk_attenuation: one v_exp_f32 per lane)
        float att[C];
        chunk_attenuation<C>(A, seg0 + n0, att);
#pragma unroll
        for (int j = 0; j < C; ++j) e[j] = __fmul_rn(e[j], att[j]);
    }
    STAMPW(5);
    // the frame row leaves from the CHUNKED mapping: two 16-byte stores per lane (a wave's 2 KiB contiguous) instead of
    // an LDS transpose and eight dword stores
    float *out = A.frame + w * A.N1 + seg0;
#ifdef DIFFUS_ABLATE_STORE
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) acc += e[j];
    if (acc == 123.456f) out[lane] = acc;
#else
    store_chunk<C, true>(out, n0, segN, e);
#endif

    STAMPW(6);
    if (A.idx) {
        const long plane = (long)A.P * A.R * A.N1;
        long long *ix = A.idx + w * A.N1 + seg0;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = j * kWave + lane;
            if (n < segN) {
                int k = A.start + seg0 + n;
                ix[n] = nearest_index(ray_point<PM>(ps, 0, k), A.G.d0);
                ix[plane + n] = nearest_index(ray_point<PM>(ps, 1, k), A.G.d1);
                ix[2 * plane + n] = nearest_index(ray_point<PM>(ps, 2, k), A.G.d2);
            }
        }
    }
}

// ----------------------------------------------------------------------------
// Standalone stages (rows a3-a6 and a7-a9 of SURVEY §8a), exposed so that each
// can be checked against the reference's golden vectors in isolation.

// trace_ray + custom_nearest_sampler + compute_reflection_coeff
// (reference src/renderer.py:90-180, :741-759, :27-33, :65-68): one thread per sample.
template <int SAMPLER, int LAYOUT>
__global__ __launch_bounds__(kBlock) void trace_rays_kernel(Args A, float *__restrict__ imp, float *__restrict__ refl,
                                                            long long *__restrict__ idx)
{
    const long total = (long)A.P * A.R * A.S;
    for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < total; t += (long)gridDim.x * kBlock) {
        const long w = t / A.S;
        const int k = (int)(t - w * A.S);
        Pose ps;
        load_pose(ps, A.src, A.src_f64, A.dirs, A.dir_f64, w / A.R, w);
        float zz[2];
        const int nq = (refl && k + 1 < A.S) ? 2 : 1;
        for (int q = 0; q < nq; ++q) {
            float p0 = ray_point(ps, 0, k + q), p1 = ray_point(ps, 1, k + q), p2 = ray_point(ps, 2, k + q);
            int i0 = nearest_index(p0, A.G.d0), i1 = nearest_index(p1, A.G.d1), i2 = nearest_index(p2, A.G.d2);
            if (q == 0 && idx) {
                idx[t] = i0;
                idx[total + t] = i1;
                idx[2 * total + t] = i2;
            }
            if (SAMPLER == DIFFUS_NEAREST)
                zz[q] = A.vol[vox_off<LAYOUT>(A.G, i0, i1, i2)];
            else
                zz[q] = tri_sample<LAYOUT, false>(A.vol, A.G, p0, p1, p2).v;
        }
        if (imp) imp[t] = zz[0];
        if (refl && k + 1 < A.S) refl[w * (A.S - 1) + k] = reflect(zz[0], zz[1]);
    }
}

// compute_echo_traces (reference src/renderer.py:439-457): r (B,N) -> echo (B,N+1),
// one wave per row.
template <int C>
__global__ __launch_bounds__(kBlock) void echo_traces_kernel(const float *__restrict__ rin, float *__restrict__ echo,
                                                             int B, int N)
{
    const long w = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (w >= B) return;
    const int lane = threadIdx.x & 63;
    // rows longer than 64*C samples: the wave walks them in pieces, the running product stays in registers
    Mat K = mat_identity();
    bool again = false; // a row of several pieces with an ill-conditioned echo somewhere: walked a second time in float64 below
    // Strong reflectors at MANY samples (coefficients nothing like tissue's: white noise of |r| ~ 0.3 and more) make the running
    // product numerically rank one -- det P = prod (1 - r^2) against entries of order one --, and the wave scan multiplies PARTIAL
    // products of that kind with each other: up to 200x the error of a sequential float32 evaluation (tools/fuzz_echo.py), with
    // |echo| itself unremarkable.  bits = -log2 det P so far; a skull crossed twice and an air interface are ~15, 512 samples of
    // |r| <= 0.3 are ~33: beyond kCondBits the row takes the float64 path like an ill-conditioned echo does.
    constexpr float kCondBits = 20.f;
    float bits = 0.f;
    for (int base = 0; base <= N; base += kWave * C) {
        const int n0 = base + lane * C;
        float r[C], e[C];
        float lb = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = n0 + j;
            r[j] = (n >= 1 && n <= N) ? rin[w * N + n - 1] : 0.f;
            lb -= __builtin_amdgcn_logf(fmaxf(fabsf(1.f - r[j] * r[j]), 0x1p-24f)); // (v_log_f32: log2; a NaN coefficient counts 24 bits)
        }
        bits += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wave_sum_to_lane63(lb)), kWave - 1));
        Mat Kl = K;
        echo_chunk<C>(r, lane, e, base ? &K : nullptr, kWave * C - 1, &Kl);
        K = mat_lane_bcast(Kl, kWave - 1);
        if (__builtin_expect(echo_needs_f64<C>(e) || bits > kCondBits, 0)) {
            if (N < kWave * C) echo_chunk_f64<C>(r, lane, e); // rows of one piece: in place; diffus_device.hpp (a stage-wise kernel: inlined)
            else again = true;
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = n0 + j;
            if (n <= N) echo[w * (N + 1) + n] = e[j];
        }
    }
    if (__builtin_expect(again, 0)) { // (wave-uniform) the whole row again, the running product carried from piece to piece in float64
        DMat Kd{1.0, 0.0, 0.0, 1.0};
#pragma unroll 1
        for (int base = 0; base <= N; base += kWave * C) {
            const int n0 = base + lane * C;
            float r[C], e[C];
#pragma unroll
            for (int j = 0; j < C; ++j) {
                int n = n0 + j;
                r[j] = (n >= 1 && n <= N) ? rin[w * N + n - 1] : 0.f;
            }
            DMat nextK;
            echo_chunk_f64<C>(r, lane, e, base ? &Kd : nullptr, &nextK);
            Kd = nextK;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                int n = n0 + j;
                if (n <= N) echo[w * (N + 1) + n] = e[j];
            }
        }
    }
}

// Rows correlated with a short kernel, zero padding (torch conv1d semantics: no flip): the pulse stage of
// compute_gaussian_pulse, reference src/renderer.py:477.  out[b][m] = sum_t k[t] * in[b][m + t - pad], m < M.
__global__ __launch_bounds__(kBlock) void rows_conv1d_kernel(const float *__restrict__ in, const float *__restrict__ k,
                                                             float *__restrict__ out, int B, int N, int L, int pad, int M)
{
    const long total = (long)B * M;
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const int b = (int)(e / M), m = (int)(e - (long)b * M);
        float acc = 0.f;
        for (int t = 0; t < L; ++t) {
            const int j = m + t - pad;
            if (j >= 0 && j < N) acc = __builtin_fmaf(k[t], in[(long)b * N + j], acc);
        }
        out[e] = acc;
    }
}

// prop_single_ray (reference src/renderer.py:367-410) in closed form: the full solution w = [g0,d0,...,gN,dN] of the
// 2(N+1) x 2(N+1) system that the reference builds and hands to torch.linalg.solve, one thread per ray, O(N).
// With rho_k = d_k / g_k (what interface k reflects of what reaches it; rho_N = 0 because d_N = 0) the two rows
// per interface (:397-405 with :380-382) give
//     rho_k   = (r_k + (1 - 2 r_k^2) rho_{k+1}) / (1 - r_k rho_{k+1})          k = N-1 .. 0
//     g_{k+1} = (1 + r_k) g_k / (1 - r_k rho_{k+1}),  d_k = rho_k g_k,  g_0 = 1   k = 0 .. N-1
// A non-finite coefficient anywhere makes the reference's whole solution NaN, which nan_to_num (:408) turns into
// zeros: reproduced as an all-zero row.  Remaining NaNs (0/0 of a singular system) become 0 like there.
template <typename T>
__global__ __launch_bounds__(kBlock) void prop_single_ray_kernel(const T *__restrict__ rin, T *__restrict__ w, int B, int N)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const T *r = rin + (long)b * N;
    T *o = w + (long)b * (2 * N + 2);
    bool ok = true;
    for (int k = 0; k < N; ++k) ok = ok && (r[k] - r[k] == T(0)); // finite
    if (!ok) {
        for (int k = 0; k < 2 * N + 2; ++k) o[k] = T(0);
        return;
    }
    T rho = T(0);
    o[2 * N + 1] = T(0);
    for (int k = N - 1; k >= 0; --k) { // rho_k parked in the d_k slot
        const T rk = r[k];
        rho = (rk + (T(1) - T(2) * rk * rk) * rho) / (T(1) - rk * rho);
        o[2 * k + 1] = rho;
    }
    T g = T(1);
    o[0] = g;
    for (int k = 0; k < N; ++k) {
        const T rho_k = o[2 * k + 1], rho_n = o[2 * k + 3];
        o[2 * k + 1] = rho_k * g;
        g = (T(1) + r[k]) * g / (T(1) - r[k] * rho_n);
        o[2 * k + 2] = g;
    }
    // d_N = rho_N g_N = 0 stays
    for (int k = 0; k < 2 * N + 2; ++k)
        if (o[k] != o[k]) o[k] = T(0);
}

// Backward of compute_echo_traces (what torch autograd does through the reference's N+1 linalg.solve nodes,
// src/renderer.py:407,430,454): r (B,N), dL/d echo (B,N+1) -> dL/d r (B,N).  SURVEY App. A.4 with the rescaling made
// explicit: forward P'_n = 2^-e_n P'_{n-1} M(r_{n-1}), echo_n = b'_n / d'_n; reverse sweep
//     Pbar_n += gbar_n [[0, 1/d'],[0, -b'/d'^2]],  Mbar = 2^-e_n P'_{n-1}^T Pbar_n,  Pbar_{n-1} = 2^-e_n Pbar_n M^T,
//     rbar_{n-1} = -4 r Mbar_00 + Mbar_01 - Mbar_10.
// One thread per ray, float64 inside (this is the module-level API, not the hot path: the fused render_bwd_kernel does
// the same adjoint as a wave scan in float32).  The forward products wait in the workspace, (N+1) x B x 4 doubles +
// exponents, laid out step-major so that neighbouring threads touch neighbouring addresses.  An echo that nan_to_num
// (:408) turned into the constant 0 passes no gradient; nothing flows through a non-finite coefficient.
__global__ __launch_bounds__(kBlock) void echo_traces_bwd_kernel(const float *__restrict__ rin, const float *__restrict__ gecho,
                                                                 float *__restrict__ gr, double *__restrict__ wsP,
                                                                 int *__restrict__ wsE, int B, int N)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    const float *r = rin + (long)b * N;
    double p00 = 1.0, p01 = 0.0, p10 = 0.0, p11 = 1.0;
    auto slot = [&](int n) { return ((size_t)n * B + b) * 4; };
    {
        double *q = wsP + slot(0);
        q[0] = p00; q[1] = p01; q[2] = p10; q[3] = p11;
        wsE[(size_t)0 * B + b] = 0;
    }
    for (int n = 1; n <= N; ++n) {
        const double rk = (double)r[n - 1], a = 1.0 - 2.0 * rk * rk;
        const double q00 = p00 * a - p01 * rk, q01 = p00 * rk + p01, q10 = p10 * a - p11 * rk, q11 = p10 * rk + p11;
        const double mx = fmax(fmax(fabs(q00), fabs(q01)), fmax(fabs(q10), fabs(q11)));
        int e = 0;
        if (mx > 0.0 && mx < __builtin_inf()) frexp(mx, &e);
        p00 = ldexp(q00, -e); p01 = ldexp(q01, -e); p10 = ldexp(q10, -e); p11 = ldexp(q11, -e);
        double *q = wsP + slot(n);
        q[0] = p00; q[1] = p01; q[2] = p10; q[3] = p11;
        wsE[(size_t)n * B + b] = e;
    }
    double u00 = 0.0, u01 = 0.0, u10 = 0.0, u11 = 0.0; // Pbar_n
    for (int n = N; n >= 1; --n) {
        const double *q = wsP + slot(n);
        const double bb = q[1], dd = q[3];
        const double echo = bb / dd;
        const double g = (double)gecho[(long)b * (N + 1) + n];
        if (echo == echo && g != 0.0) { // a NaN echo became the constant 0: no gradient through it
            const double s = g / dd, t = -g * echo / dd;
            if (s - s == 0.0 && t - t == 0.0) { // finite seeds only
                u01 += s;
                u11 += t;
            }
        }
        const double rk = (double)r[n - 1];
        if (!(rk - rk == 0.0)) { // non-finite coefficient: everything after it is the constant 0, nothing flows
            gr[(long)b * N + n - 1] = 0.f;
            u00 = u01 = u10 = u11 = 0.0;
            continue;
        }
        const double *pm = wsP + slot(n - 1);
        const int e = wsE[(size_t)n * B + b];
        // Mbar = 2^-e P'_{n-1}^T Pbar_n
        const double m00 = pm[0] * u00 + pm[2] * u10, m01 = pm[0] * u01 + pm[2] * u11;
        const double m10 = pm[1] * u00 + pm[3] * u10;
        double rb = ldexp(-4.0 * rk * m00 + m01 - m10, -e);
        if (!(rb - rb == 0.0)) rb = 0.0;
        gr[(long)b * N + n - 1] = (float)rb;
        // Pbar_{n-1} = 2^-e Pbar_n M^T,  M = [[a, r],[-r, 1]]
        const double a = 1.0 - 2.0 * rk * rk;
        const double v00 = u00 * a + u01 * rk, v01 = -u00 * rk + u01, v10 = u10 * a + u11 * rk, v11 = -u10 * rk + u11;
        u00 = ldexp(v00, -e); u01 = ldexp(v01, -e); u10 = ldexp(v10, -e); u11 = ldexp(v11, -e);
        if (!(u00 - u00 == 0.0 && u01 - u01 == 0.0 && u10 - u10 == 0.0 && u11 - u11 == 0.0)) u00 = u01 = u10 = u11 = 0.0;
    }
}

// running sum along each row, in place: the cumsum of propagate_full_rays_batched (reference :435)
__global__ __launch_bounds__(kBlock) void rows_cumsum_kernel(float *__restrict__ a, int B, int M)
{
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= B) return;
    float *row = a + (long)b * M;
    float acc = 0.f;
    for (int k = 0; k < M; ++k) {
        acc += row[k];
        row[k] = acc;
    }
}

// custom_nearest_sampler (reference src/renderer.py:741-759) at ARBITRARY points: round half to even, clamp, gather.
template <int SAMPLER, int LAYOUT>
__global__ __launch_bounds__(kBlock) void sample_points_kernel(const float *__restrict__ vol, Geom G, const float *__restrict__ pts,
                                                               long n, float *__restrict__ val, long long *__restrict__ idx)
{
    for (long t = (long)blockIdx.x * kBlock + threadIdx.x; t < n; t += (long)gridDim.x * kBlock) {
        const float p0 = pts[t * 3], p1 = pts[t * 3 + 1], p2 = pts[t * 3 + 2];
        const int i0 = nearest_index(p0, G.d0), i1 = nearest_index(p1, G.d1), i2 = nearest_index(p2, G.d2);
        if (idx) {
            idx[t] = i0;
            idx[n + t] = i1;
            idx[2 * n + t] = i2;
        }
        if (val) val[t] = (SAMPLER == DIFFUS_NEAREST) ? vol[vox_off<LAYOUT>(G, i0, i1, i2)] : tri_sample<LAYOUT, false>(vol, G, p0, p1, p2).v;
    }
}

// Rays of more than DIFFUS_MAX_SAMPLES samples go through render_fwd_kernel in chained launches whose carries are float32.  Those a
// launch flagged (|echo| > kEchoRecheck somewhere) are walked again here, one wave per ray, piece after piece from the first sample,
// the running product carried in float64 and the samples re-taken with the stage-wise sampler (echo_f64_rare): the frame row is
// rewritten whole.
template <int SAMPLER, int LAYOUT, int PM>
__global__ __launch_bounds__(kBlock) void render_fwd_long_repair_kernel(Args A)
{
    constexpr int C = 16;
    const long w = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (w >= (long)A.P * A.R) return;
    if (__builtin_amdgcn_readfirstlane(A.rflag[w * 2]) == 0) return; // (wave-uniform)
    const int lane = threadIdx.x & 63, n0 = lane * C;
    const long pose = w / A.R;
    Pose ps;
    load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
    const float medv = (A.start > 0) ? A.med[pose] : 0.f;
    DMat K{1.0, 0.0, 0.0, 1.0};
    float zc = 0.f;
#pragma unroll 1
    for (int seg0 = 0; seg0 < A.N1; seg0 += kWave * C) {
        const int segN = min(A.N1 - seg0, kWave * C);
        float e[C], znext;
        DMat Knext;
        echo_f64_rare<C, SAMPLER, LAYOUT, PM>(A, ps, seg0, segN, n0, medv, e, seg0 ? &K : nullptr, &Knext, seg0 ? &zc : nullptr, &znext);
        K = Knext;
        zc = znext;
        float att[C];
        chunk_attenuation<C>(A, seg0 + n0, att);
#pragma unroll
        for (int j = 0; j < C; ++j) e[j] = __fmul_rn(e[j], att[j]);
        store_chunk<C, true>(A.frame + w * A.N1 + seg0, n0, segN, e);
    }
}

template <int SM, int LY, int PM>
int launch_fwd_long_repair_t(const Args &A, hipStream_t st)
{
    const long waves = (long)A.P * A.R;
    hipLaunchKernelGGL((render_fwd_long_repair_kernel<SM, LY, PM>), dim3((unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock)), dim3(kBlock), 0, st, A);
    return last_launch();
}

template <int SM, int LY, int PM>
int launch_fwd_t(const Args &A, hipStream_t st)
{
    const long waves = (long)A.P * A.R;
    const unsigned nblk = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    switch (chunk_for(A.N1)) {
    case 2: hipLaunchKernelGGL((render_fwd_kernel<2, SM, LY, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    case 4: hipLaunchKernelGGL((render_fwd_kernel<4, SM, LY, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    case 8: hipLaunchKernelGGL((render_fwd_kernel<8, SM, LY, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    default: // 512 < N1 <= 1024: one wave with 16 samples per lane (100 VGPRs).  The two-wave SPLIT form that pays for
             // the backward (render_bwd.hip) does not here: 31.4 against 21.9 us at 8 poses x 512 rays x 1024 steps
#ifdef DIFFUS_FWD_SPLIT
        hipLaunchKernelGGL((render_fwd_kernel<8, SM, LY, 2, PM, false, 2>), dim3((unsigned)waves), dim3(2 * kWave), 0, st, A);
#else
        hipLaunchKernelGGL((render_fwd_kernel<16, SM, LY, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A);
#endif
        break;
    }
    return last_launch();
}

template <int SM, int LY, int PM>
int launch_fwd_seg(const Args &A, hipStream_t st)
{
    const long waves = (long)A.P * A.R;
    const unsigned nblk = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL((render_fwd_kernel<16, SM, LY, kWavesPerBlock, PM, true>), dim3(nblk), dim3(kBlock), 0, st, A);
    return last_launch();
}

} // namespace

// one launch of the forward kernel over the segment [A.seg0, A.seg0 + A.segN); also used by the backward of
// long rays (render_bwd.hip) as its carry-only pass
int diffus::launch_fwd(const Args &A, int sampler, int layout, hipStream_t st)
{
    const bool f32 = !A.src_f64 && !A.dir_f64;
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        constexpr int SM = decltype(S_)::value, LY = decltype(L_)::value;
        if (A.N1 > DIFFUS_MAX_SAMPLES) return f32 ? launch_fwd_seg<SM, LY, 0>(A, st) : launch_fwd_seg<SM, LY, 1>(A, st);
        return f32 ? launch_fwd_t<SM, LY, 0>(A, st) : launch_fwd_t<SM, LY, 1>(A, st);
    });
}

namespace {
int launch_fwd_long_repair(const Args &A, int sampler, int layout, hipStream_t st)
{
    const bool f32 = !A.src_f64 && !A.dir_f64;
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        constexpr int SM = decltype(S_)::value, LY = decltype(L_)::value;
        return f32 ? launch_fwd_long_repair_t<SM, LY, 0>(A, st) : launch_fwd_long_repair_t<SM, LY, 1>(A, st);
    });
}
} // namespace

extern "C" {

int diffus_abi_version(void) { return DIFFUS_ABI_VERSION; }

const char *diffus_strerror(int code)
{
    switch (code) {
    case DIFFUS_OK: return "ok";
    case DIFFUS_EINVAL: return "invalid argument";
    case DIFFUS_EUNSUPPORTED: return "unsupported shape (S - start > 65536, a volume edge > 2^24, or too many rays for start > 0)";
    case DIFFUS_ELAUNCH: return "HIP launch failure";
    case DIFFUS_EWORKSPACE: return "workspace too small (see diffus_workspace_bytes)";
    default: return "unknown diffus error";
    }
}

size_t diffus_workspace_bytes(int P, int R, int S, int start)
{
    if (P <= 0 || R <= 0 || S <= 0 || start < 0 || start >= S) return 0;
    return carve(nullptr, P, R, S - start).bytes;
}

size_t diffus_workspace_zbar_offset(int P, int R, int S, int start)
{
    if (P <= 0 || R <= 0 || S <= 0 || start < 0 || start >= S) return 0;
    Workspace ws = carve(nullptr, P, R, S - start);
    return (size_t)((char *)ws.zbar - (char *)nullptr);
}

int diffus_render_fwd(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype, int P, int R, int S, int start, float alpha, int sampler,
                      float *frame, int64_t *idx, void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    int rc = check_common(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, start, sampler, layout, true);
    if (rc) return rc;
    if (!frame) return DIFFUS_EINVAL;
    Workspace ws = carve(workspace, P, R, S - start);
    if ((start > 0 || ws.nseg > 1) && (!workspace || workspace_bytes < ws.bytes)) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Args A = make_args(vol, d0, d1, d2, layout, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, ws);
    A.frame = frame;
    A.idx = (long long *)idx;
    if (start > 0) {
        rc = launch_median(A, sampler, layout, st);
        if (rc) return rc;
    }
    // rays longer than one launch covers: segments of DIFFUS_MAX_SAMPLES chained through the running product
    const size_t wr = (size_t)P * R;
    if (ws.nseg > 1) { // ... whose launches flag the rays with an ill-conditioned echo for the float64 walk below
        if (hipMemsetAsync(ws.rflag, 0, sizeof(int) * wr * 2, st) != hipSuccess) return DIFFUS_ELAUNCH;
        A.rflag = ws.rflag;
    }
    for (int s = 0; s < ws.nseg; ++s) {
        A.seg0 = s * DIFFUS_MAX_SAMPLES;
        A.segN = (s == ws.nseg - 1) ? A.N1 - A.seg0 : DIFFUS_MAX_SAMPLES;
        A.cin = s ? ws.carry + (size_t)(s - 1) * wr * 5 : nullptr;
        A.cout = (s + 1 < ws.nseg) ? ws.carry + (size_t)s * wr * 5 : nullptr;
        rc = diffus::launch_fwd(A, sampler, layout, st);
        if (rc) return rc;
    }
    if (ws.nseg > 1) {
        A.seg0 = 0;
        A.segN = A.N1;
        rc = launch_fwd_long_repair(A, sampler, layout, st);
        if (rc) return rc;
    }
    return DIFFUS_OK;
}

int diffus_trace_rays(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype, int P, int R, int S, int sampler, float *imp, float *refl,
                      int64_t *idx, diffus_stream_t stream)
{
    int rc = check_common(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, 0, sampler, layout, false);
    if (rc) return rc;
    if (!imp && !refl && !idx) return DIFFUS_OK;
    Workspace ws = carve(nullptr, P, R, S);
    Args A = make_args(vol, d0, d1, d2, layout, src, src_dtype, dirs, dirs_dtype, P, R, S, 0, 0.f, ws);
    const long total = (long)P * R * S;
    unsigned nblk = (unsigned)((total + kBlock - 1) / kBlock);
    if (nblk > 256u * 16u) nblk = 256u * 16u;
    hipStream_t st = (hipStream_t)stream;
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        hipLaunchKernelGGL((trace_rays_kernel<decltype(S_)::value, decltype(L_)::value>), dim3(nblk), dim3(kBlock), 0, st,
                           A, imp, refl, (long long *)idx);
        return last_launch();
    });
}

#ifdef DIFFUS_STAMP
int diffus_debug_set_fwd_stamps(unsigned long long *p)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif


int diffus_echo_traces(const float *refl, int B, int N, float *echo, diffus_stream_t stream)
{
    if (!refl && N > 0) return DIFFUS_EINVAL;
    if (!echo || B <= 0 || N < 0) return DIFFUS_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const unsigned nblk = (unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock);
    switch (chunk_for(N + 1)) {
    case 2: hipLaunchKernelGGL(echo_traces_kernel<2>, dim3(nblk), dim3(kBlock), 0, st, refl, echo, B, N); break;
    case 4: hipLaunchKernelGGL(echo_traces_kernel<4>, dim3(nblk), dim3(kBlock), 0, st, refl, echo, B, N); break;
    case 8: hipLaunchKernelGGL(echo_traces_kernel<8>, dim3(nblk), dim3(kBlock), 0, st, refl, echo, B, N); break;
    default: hipLaunchKernelGGL(echo_traces_kernel<16>, dim3(nblk), dim3(kBlock), 0, st, refl, echo, B, N); break;
    }
    return last_launch();
}

size_t diffus_echo_bwd_workspace_bytes(int B, int N)
{
    if (B <= 0 || N < 0) return 0;
    return align256((size_t)B * (N + 1) * 4 * sizeof(double)) + align256((size_t)B * (N + 1) * sizeof(int));
}

int diffus_echo_traces_bwd(const float *refl, int B, int N, const float *gecho, float *grefl, void *workspace,
                           size_t workspace_bytes, diffus_stream_t stream)
{
    if (B <= 0 || N < 0 || !gecho) return DIFFUS_EINVAL;
    if (N == 0) return DIFFUS_OK; // no coefficient, no gradient
    if (!refl || !grefl) return DIFFUS_EINVAL;
    if (!workspace || workspace_bytes < diffus_echo_bwd_workspace_bytes(B, N)) return DIFFUS_EWORKSPACE;
    double *wsP = (double *)workspace;
    int *wsE = (int *)((char *)workspace + align256((size_t)B * (N + 1) * 4 * sizeof(double)));
    hipLaunchKernelGGL(echo_traces_bwd_kernel, dim3((unsigned)((B + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       refl, gecho, grefl, wsP, wsE, B, N);
    return last_launch();
}

int diffus_prop_single_ray(const void *refl, int dtype, int B, int N, void *w, diffus_stream_t stream)
{
    if (!w || B <= 0 || N < 0 || (!refl && N > 0)) return DIFFUS_EINVAL;
    if (dtype != DIFFUS_F32 && dtype != DIFFUS_F64) return DIFFUS_EINVAL;
    const unsigned nb = (unsigned)((B + kBlock - 1) / kBlock);
    if (dtype == DIFFUS_F32)
        hipLaunchKernelGGL(prop_single_ray_kernel<float>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, (const float *)refl, (float *)w, B, N);
    else
        hipLaunchKernelGGL(prop_single_ray_kernel<double>, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, (const double *)refl, (double *)w, B, N);
    return last_launch();
}

int diffus_propagate_rays(const float *refl, int B, int N, float *d0_cum, diffus_stream_t stream)
{
    int rc = diffus_echo_traces(refl, B, N, d0_cum, stream); // d0^(n) per truncation depth (d0^(0) = 0)
    if (rc) return rc;
    hipLaunchKernelGGL(rows_cumsum_kernel, dim3((unsigned)((B + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, d0_cum, B, N + 1);
    return last_launch();
}

int diffus_sample_points(const float *vol, int d0, int d1, int d2, int layout, const float *points, long n, int sampler,
                         float *values, int64_t *idx, diffus_stream_t stream)
{
    if (!vol || !points || n <= 0 || d0 <= 0 || d1 <= 0 || d2 <= 0) return DIFFUS_EINVAL;
    if (sampler != DIFFUS_NEAREST && sampler != DIFFUS_TRILINEAR) return DIFFUS_EINVAL;
    if (layout != DIFFUS_CANONICAL && layout != DIFFUS_BRICKED && layout != DIFFUS_PAIRED) return DIFFUS_EINVAL;
    if (d0 > (1 << 24) || d1 > (1 << 24) || d2 > (1 << 24) || bricked_floats(d0, d1, d2) >= ((size_t)1 << 30)) return DIFFUS_EUNSUPPORTED;
    if (!values && !idx) return DIFFUS_OK;
    const Geom G = make_geom(d0, d1, d2, layout);
    unsigned nb = (unsigned)((n + kBlock - 1) / kBlock);
    if (nb > 256u * 16u) nb = 256u * 16u;
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        hipLaunchKernelGGL((sample_points_kernel<decltype(S_)::value, decltype(L_)::value>), dim3(nb), dim3(kBlock), 0,
                           (hipStream_t)stream, vol, G, points, n, values, (long long *)idx);
        return last_launch();
    });
}

int diffus_rows_conv1d(const float *in, int B, int N, const float *kernel, int L, int pad, float *out,
                       diffus_stream_t stream)
{
    if (!in || !kernel || !out || B <= 0 || N <= 0 || L <= 0 || pad < 0) return DIFFUS_EINVAL;
    const long M = (long)N + 2L * pad - L + 1;
    if (M <= 0 || M > 0x7fffffffL) return DIFFUS_EINVAL;
    const long total = (long)B * M;
    unsigned nb = (unsigned)((total + kBlock - 1) / kBlock);
    if (nb > 4096u) nb = 4096u;
    hipLaunchKernelGGL(rows_conv1d_kernel, dim3(nb), dim3(kBlock), 0, (hipStream_t)stream, in, kernel, out, B, N, L, pad, (int)M);
    return last_launch();
}

} // extern "C"
