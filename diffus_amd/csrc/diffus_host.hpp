// diffus_host.hpp -- host-side helpers shared by the translation units: workspace carving, argument
// validation, (sampler, layout) -> template dispatch, and the per-pose median kernel that both the
// forward and the backward launch when start > 0.
#pragma once
#include <cmath>

#include "diffus_device.hpp"

namespace {

// ----------------------------------------------------------------------------
// host side
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// byte strides of the gather's separable offsets (diffus_device.hpp part_x / part_y) for a volume layout
size_t stride_x_bytes(int layout, int d1, int d2)
{
    const size_t nb1 = (size_t)(d1 + 3) / 4, nb2 = (size_t)(d2 + 1) / 2;
    if (layout == DIFFUS_CANONICAL) return (size_t)d1 * d2 * 4;
    return layout == DIFFUS_PAIRED ? nb1 * d2 * (kPairFloats * 4) : nb1 * nb2 * 128;
}
size_t stride_y_bytes(int layout, int d2)
{
    if (layout == DIFFUS_CANONICAL) return (size_t)d2 * 4;
    return layout == DIFFUS_PAIRED ? (size_t)d2 * (kPairFloats * 4) : (size_t)((d2 + 1) / 2) * 128;
}

Geom make_geom(int d0, int d1, int d2, int layout = DIFFUS_CANONICAL)
{
    Geom G;
    G.d0 = d0; G.d1 = d1; G.d2 = d2;
    G.nb1 = (d1 + 3) / 4;
    G.nb2 = (d2 + 1) / 2;
    G.sxB = (unsigned)stride_x_bytes(layout, d1, d2);
    G.syB = (unsigned)stride_y_bytes(layout, d2);
    return G;
}

size_t bricked_floats(int d0, int d1, int d2)
{
    Geom G = make_geom(d0, d1, d2);
    return (size_t)((d0 + 3) / 4) * G.nb1 * G.nb2 * kBrickFloats;
}

size_t paired_floats(int d0, int d1, int d2)
{
    Geom G = make_geom(d0, d1, d2);
    return (size_t)((d0 + 3) / 4) * G.nb1 * d2 * kPairFloats;
}

struct Workspace {
    float *med;
    int *who;
    float *gmed;
    float *medinfo;
    float *loss_part; // (P,R,2) fused-loss partial sums
    int *rflag;       // (P,R,2) one-pass step: ray (half) whose frame row pose_finish_block evaluates again in float64
    float *gsrc_part;
    float *zbar;
    float *carry;  // (nseg-1, P*R, 5) per-segment carry-in of long rays
    float *ucarry; // (2, P*R, 4) adjoint carry, ping-pong
    float *zcarry; // (2, P*R)
    int nseg;      // launches per ray: ceil(N1 / DIFFUS_MAX_SAMPLES)
    size_t bytes;
};

Workspace carve(void *base, int P, int R, int N1)
{
    Workspace ws;
    char *p = (char *)base;
    size_t o = 0;
    ws.med = (float *)(p + o); o += align256(sizeof(float) * (size_t)P);
    ws.who = (int *)(p + o);   o += align256(sizeof(int) * (size_t)P);
    ws.gmed = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R);
    ws.medinfo = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * 8);
    ws.loss_part = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R * 2);
    ws.rflag = (int *)(p + o); o += align256(sizeof(int) * (size_t)P * R * 2);
    ws.gsrc_part = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R * 3);
    ws.zbar = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R * (N1 > 0 ? N1 : 0));
    ws.nseg = N1 > 0 ? (N1 + DIFFUS_MAX_SAMPLES - 1) / DIFFUS_MAX_SAMPLES : 1;
    ws.carry = ws.ucarry = ws.zcarry = nullptr;
    if (ws.nseg > 1) {
        ws.carry = (float *)(p + o);  o += align256(sizeof(float) * (size_t)P * R * 5 * (ws.nseg - 1));
        ws.ucarry = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R * 4 * 2);
        ws.zcarry = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R * 2);
    }
    ws.bytes = o;
    return ws;
}

int chunk_for(int N1)
{
    int c = (N1 + kWave - 1) / kWave;
    if (c <= 2) return 2;
    if (c <= 4) return 4;
    if (c <= 8) return 8;
    return 16;
}

int check_common(const float *vol, int d0, int d1, int d2, const void *src, int src_dtype, const void *dirs,
                 int dirs_dtype, int P, int R, int S, int start, int sampler, int layout, bool need_scan)
{
    if (!vol || !src || !dirs) return DIFFUS_EINVAL;
    if (d0 <= 0 || d1 <= 0 || d2 <= 0 || P <= 0 || R <= 0 || S <= 0) return DIFFUS_EINVAL;
    if ((src_dtype != DIFFUS_F32 && src_dtype != DIFFUS_F64) || (dirs_dtype != DIFFUS_F32 && dirs_dtype != DIFFUS_F64))
        return DIFFUS_EINVAL;
    if (sampler != DIFFUS_NEAREST && sampler != DIFFUS_TRILINEAR) return DIFFUS_EINVAL;
    if (layout != DIFFUS_CANONICAL && layout != DIFFUS_BRICKED && layout != DIFFUS_PAIRED) return DIFFUS_EINVAL;
    if (start < 0 || start > S - 1) return DIFFUS_EINVAL;
    if (start > 0 && start > S - 2) return DIFFUS_EINVAL; // reference raises IndexError at :243
    if (d0 > (1 << 24) || d1 > (1 << 24) || d2 > (1 << 24)) return DIFFUS_EUNSUPPORTED; // float(dim-1) must be exact
    if (bricked_floats(d0, d1, d2) >= ((size_t)1 << 30)) return DIFFUS_EUNSUPPORTED;    // 32-bit element offsets
    if (layout == DIFFUS_PAIRED && paired_floats(d0, d1, d2) >= ((size_t)1 << 30)) return DIFFUS_EUNSUPPORTED;
    // bricked / paired: the brick-row stride is the 24-bit operand of a v_mul_u32_u24 (part_x); slices of 2^17 bricks
    // or more (e.g. 2048 x 1024) are not bricked -- the caller keeps such a volume canonical
    if (layout != DIFFUS_CANONICAL && stride_x_bytes(layout, d1, d2) >= ((size_t)1 << 24)) return DIFFUS_EUNSUPPORTED;
    if (need_scan && S - start > DIFFUS_MAX_SAMPLES * DIFFUS_MAX_SEGMENTS) return DIFFUS_EUNSUPPORTED;
    if (start > 0 && (size_t)R * sizeof(float) > 64 * 1024) return DIFFUS_EUNSUPPORTED; // median LDS
    return DIFFUS_OK;
}

Args make_args(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype, const void *dirs,
               int dirs_dtype, int P, int R, int S, int start, float alpha, const Workspace &ws)
{
    Args A{};
    A.vol = vol;
    A.G = make_geom(d0, d1, d2, layout);
    A.src = src; A.dirs = dirs;
    A.src_f64 = src_dtype == DIFFUS_F64; A.dir_f64 = dirs_dtype == DIFFUS_F64;
    A.P = P; A.R = R; A.S = S; A.start = start; A.N1 = S - start;
    A.seg0 = 0; A.segN = A.N1; // one launch covers the ray unless the caller loops over segments
    A.neg_alpha = -alpha;
    A.neg_alpha_l2e = (float)(-(double)alpha * 1.4426950408889634);
    for (int j = 0; j < 16; ++j) A.att_step[j] = exp2f(A.neg_alpha_l2e * (float)j);
    A.med = ws.med; A.who = ws.who; A.gmed = ws.gmed; A.medinfo = ws.medinfo;
    return A;
}

int last_launch() { return hipGetLastError() == hipSuccess ? DIFFUS_OK : DIFFUS_ELAUNCH; }

// (sampler, layout) -> compile-time constants
template <int SM, typename F>
int dispatch_layout(int layout, F &&f)
{
    using S_ = std::integral_constant<int, SM>;
    switch (layout) {
    case DIFFUS_CANONICAL: return f(S_{}, std::integral_constant<int, DIFFUS_CANONICAL>{});
    case DIFFUS_BRICKED: return f(S_{}, std::integral_constant<int, DIFFUS_BRICKED>{});
    default: return f(S_{}, std::integral_constant<int, DIFFUS_PAIRED>{});
    }
}

template <typename F>
int dispatch_sl(int sampler, int layout, F &&f)
{
    return sampler == DIFFUS_NEAREST ? dispatch_layout<DIFFUS_NEAREST>(layout, f)
                                     : dispatch_layout<DIFFUS_TRILINEAR>(layout, f);
}

// ----------------------------------------------------------------------------
// start > 0: median over rays of r[:, start] (reference :243), one block per pose.
// Lower median like torch.median; NaN if any NaN.  Leaves the median ray's samples in medinfo for the backward.
template <int SAMPLER, int LAYOUT>
__global__ __launch_bounds__(kBlock) void median_kernel(Args A)
{
    extern __shared__ __attribute__((aligned(16))) float vals[];
    __shared__ int s_nan;
    const int pose = blockIdx.x;
    if (threadIdx.x == 0) s_nan = 0;
    __syncthreads();
    // the two samples (steps start, start + 1) of a ray and, trilinear, their spatial gradients: mi[8] as in medinfo
    auto sample_ray = [&](int i, float (&mi)[8]) {
        Pose ps;
        load_pose(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, (long)pose * A.R + i);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int k = A.start + q;
            const float p0 = ray_point(ps, 0, k), p1 = ray_point(ps, 1, k), p2 = ray_point(ps, 2, k);
            if (SAMPLER == DIFFUS_NEAREST) {
                const int i0 = nearest_index(p0, A.G.d0), i1 = nearest_index(p1, A.G.d1), i2 = nearest_index(p2, A.G.d2);
                mi[q] = A.vol[vox_off<LAYOUT>(A.G, i0, i1, i2)];
                mi[2 + 3 * q] = mi[3 + 3 * q] = mi[4 + 3 * q] = 0.f;
            } else {
                const TriSample sm = tri_sample<LAYOUT, true>(A.vol, A.G, p0, p1, p2);
                mi[q] = sm.v;
                mi[2 + 3 * q] = sm.g0; mi[3 + 3 * q] = sm.g1; mi[4 + 3 * q] = sm.g2;
            }
        }
    };
    // The first ray of a thread (every ray when R <= 256) keeps its samples in registers: the thread that turns out to
    // hold the median writes them out without a second round trip to memory.
    float mine[8];
    for (int i = threadIdx.x; i < A.R; i += blockDim.x) {
        float mi[8];
        sample_ray(i, mi);
        if (i == (int)threadIdx.x) {
#pragma unroll
            for (int q = 0; q < 8; ++q) mine[q] = mi[q];
        }
        const float v = reflect(mi[0], mi[1]);
        vals[i] = v;
        if (v != v) atomicOr(&s_nan, 1);
    }
    __syncthreads();
    if (s_nan) {
        if (threadIdx.x == 0) {
            A.med[pose] = __builtin_nanf("");
            A.who[pose] = -1;
        }
        return;
    }
    const int target = (A.R - 1) / 2;
    for (int i = threadIdx.x; i < A.R; i += blockDim.x) {
        const float v = vals[i];
        int rank = 0;
        int j = 0;
#pragma unroll 4
        for (; j + 4 <= A.R; j += 4) { // vals is 16-byte aligned dynamic LDS: one ds_read_b128 (a broadcast) per 4 values
            const float4 u = *reinterpret_cast<const float4 *>(vals + j);
            rank += (u.x < v) || (u.x == v && j < i);
            rank += (u.y < v) || (u.y == v && j + 1 < i);
            rank += (u.z < v) || (u.z == v && j + 2 < i);
            rank += (u.w < v) || (u.w == v && j + 3 < i);
        }
        for (; j < A.R; ++j) {
            const float u = vals[j];
            rank += (u < v) || (u == v && j < i);
        }
        if (rank == target) { // exactly one i satisfies this
            A.med[pose] = v;
            A.who[pose] = i;
            // what the backward needs to route d/d median to this ray (pose_finish_block)
            float mi[8];
            if (i == (int)threadIdx.x) {
#pragma unroll
                for (int q = 0; q < 8; ++q) mi[q] = mine[q];
            } else {
                sample_ray(i, mi);
            }
            float *out = A.medinfo + (long)pose * 8;
#pragma unroll
            for (int q = 0; q < 8; ++q) out[q] = mi[q];
        }
    }
}

int launch_median(const Args &A, int sampler, int layout, hipStream_t st)
{
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        hipLaunchKernelGGL((median_kernel<decltype(S_)::value, decltype(L_)::value>), dim3(A.P), dim3(kBlock),
                           sizeof(float) * (size_t)A.R, st, A);
        return last_launch();
    });
}

} // namespace

// cross-unit calls: the volume-gradient scatter of the backward (scatter.hip) and the forward launch
namespace diffus {
int launch_scatter(const Args &A, int sampler, int layout, hipStream_t st);
int launch_fwd(const Args &A, int sampler, int layout, hipStream_t st); // render_fwd.hip
}
