// pose.hip -- the probe-pose parameterisation on the device, with its C-ABI entry points
#include "diffus_host.hpp"

namespace {

// ----------------------------------------------------------------------------
// FAN DIRECTIONS FROM POSE PARAMETERS, AND BACK.
//   The reference builds a fan once, on the host: generate_cone_directions (src/cone.py:242-258) -- ray i of R has
//   the in-plane angle a_i in linspace(-opening / 2, opening / 2, R) about the median direction, third component 0 --
//   and cone_us_to_mri_world (:187-209) places it; nothing there is differentiable.  A registration loop
//   that descends d loss / d directions (the HIP backward's product) needs the map from the pose PARAMETERS -- median
//   angle m, opening angle, rotation vector rho (axis x angle: the fan's plane rolled / pitched out of the slice) --
//   to the directions and its adjoint on every iteration; as ~80 small torch ops that pair is 0.95 ms of a 1.56 ms
//   iteration at a 256 x 512 frame (tools/prof_registration.py), 40 x the render's own forward launch.  Here: one
//   launch each way, a block per pose, everything in float64 (a few hundred flops per ray), float32 in and out.
//
//     dir_i = R(rho) (cos(m + a_i), sin(m + a_i), 0),    a_i = opening * lin_i,   lin = linspace(-1/2, 1/2, R)
//     R(rho) = I + A K + B K^2,  K = [rho]x,  A = sin t / t,  B = (1 - cos t) / t^2,  t = |rho|   (Rodrigues)
//   Adjoint, with g_i = dL / d dir_i and w_i = R (-sin(m + a_i), cos(m + a_i), 0) = d dir_i / d m:
//     dL/dm = sum_i g_i . w_i,      dL/dopening = sum_i lin_i g_i . w_i,
//     dL/drho = J_l(rho)^T sum_i dir_i x g_i,   J_l = I + B K + C K^2,  C = (t - sin t) / t^3   (left Jacobian of SO(3):
//     R(rho + d) = exp([J_l d]x) R(rho) to first order)
//   A, B, C from their series below t^2 = 1e-6, so that value and gradient are exact at the identity.
struct RotCoef {
    double A, B, C;
};
__device__ __forceinline__ RotCoef rot_coef(double t2)
{
    RotCoef c;
    if (t2 < 1e-6) {
        c.A = 1.0 - t2 / 6.0 + t2 * t2 / 120.0;
        c.B = 0.5 - t2 / 24.0 + t2 * t2 / 720.0;
        c.C = 1.0 / 6.0 - t2 / 120.0 + t2 * t2 / 5040.0;
    } else {
        const double t = sqrt(t2);
        c.A = sin(t) / t;
        c.B = (1.0 - cos(t)) / t2;
        c.C = (t - sin(t)) / (t2 * t);
    }
    return c;
}
// M = I + a K + b K^2 for K = [rho]x, row-major
__device__ __forceinline__ void rot_poly(const double (&r)[3], double a, double b, double (&M)[9])
{
    const double x = r[0], y = r[1], z = r[2];
    // K^2 = rho rho^T - |rho|^2 I
    const double t2 = x * x + y * y + z * z;
    M[0] = 1.0 + b * (x * x - t2); M[1] = -a * z + b * x * y;     M[2] = a * y + b * x * z;
    M[3] = a * z + b * x * y;      M[4] = 1.0 + b * (y * y - t2); M[5] = -a * x + b * y * z;
    M[6] = -a * y + b * x * z;     M[7] = a * x + b * y * z;      M[8] = 1.0 + b * (z * z - t2);
}
// torch.linspace(-0.5, 0.5, R): from the start in the lower half, from the end in the upper half (symmetric to the last bit)
__device__ __forceinline__ double fan_lin(int i, int R)
{
    if (R <= 1) return -0.5;
    const double step = 1.0 / (double)(R - 1);
    return (2 * i < R) ? -0.5 + (double)i * step : 0.5 - (double)(R - 1 - i) * step;
}
__device__ __forceinline__ void load_rho(const float *rotvec, int p, double (&r)[3])
{
    r[0] = r[1] = r[2] = 0.0;
    if (rotvec) {
#pragma unroll
        for (int c = 0; c < 3; ++c) r[c] = (double)rotvec[3 * (long)p + c];
    }
}

__global__ __launch_bounds__(kBlock) void fan_pose_fwd_kernel(const float *__restrict__ median, const float *__restrict__ opening,
                                                              int opening_stride, const float *__restrict__ rotvec, int R, float *__restrict__ dirs)
{
    const int p = blockIdx.x;
    double r[3], M[9];
    load_rho(rotvec, p, r);
    const RotCoef c = rot_coef(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    rot_poly(r, c.A, c.B, M);
    const double m = (double)median[p], th = (double)opening[(long)p * opening_stride];
    float *out = dirs + (long)p * R * 3;
    for (int i = threadIdx.x; i < R; i += kBlock) {
        double sn, cs;
        sincos(m + th * fan_lin(i, R), &sn, &cs);
        out[3 * i + 0] = (float)(cs * M[0] + sn * M[1]);
        out[3 * i + 1] = (float)(cs * M[3] + sn * M[4]);
        out[3 * i + 2] = rotvec ? (float)(cs * M[6] + sn * M[7]) : 0.f; // no rotation vector: the reference's fan, an exact +0
    }
}

__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}

__global__ __launch_bounds__(kBlock) void fan_pose_bwd_kernel(const float *__restrict__ median, const float *__restrict__ opening,
                                                              int opening_stride, const float *__restrict__ rotvec, const float *__restrict__ gdirs,
                                                              int R, float *__restrict__ g_median, float *__restrict__ g_opening, float *__restrict__ g_rotvec)
{
    __shared__ double part[kWavesPerBlock][5];
    const int p = blockIdx.x;
    double r[3], M[9];
    load_rho(rotvec, p, r);
    const RotCoef c = rot_coef(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    rot_poly(r, c.A, c.B, M);
    const double m = (double)median[p], th = (double)opening[(long)p * opening_stride];
    const float *g = gdirs + (long)p * R * 3;
    double acc[5] = {0.0, 0.0, 0.0, 0.0, 0.0}; // dL/dm, dL/dopening, torque
    for (int i = threadIdx.x; i < R; i += kBlock) {
        const double lin = fan_lin(i, R);
        double sn, cs;
        sincos(m + th * lin, &sn, &cs);
        const double g0 = (double)g[3 * i + 0], g1 = (double)g[3 * i + 1], g2 = (double)g[3 * i + 2];
        // g . (first column of R), g . (second column)
        const double c0 = g0 * M[0] + g1 * M[3] + g2 * M[6], c1 = g0 * M[1] + g1 * M[4] + g2 * M[7];
        const double s = cs * c1 - sn * c0;
        acc[0] += s;
        acc[1] += lin * s;
        const double d0 = cs * M[0] + sn * M[1], d1 = cs * M[3] + sn * M[4], d2 = cs * M[6] + sn * M[7];
        acc[2] += d1 * g2 - d2 * g1;
        acc[3] += d2 * g0 - d0 * g2;
        acc[4] += d0 * g1 - d1 * g0;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) acc[k] = wave_sum_d(acc[k]);
    if ((threadIdx.x & (kWave - 1)) == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) part[threadIdx.x / kWave][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            tot[k] = 0.0;
            for (int w = 0; w < kWavesPerBlock; ++w) tot[k] += part[w][k];
        }
        if (g_median) g_median[p] = (float)tot[0];
        if (g_opening) g_opening[p] = (float)tot[1];
        if (g_rotvec) {
            double J[9];
            rot_poly(r, c.B, c.C, J); // J_l; its transpose applied to the torque
            g_rotvec[3 * (long)p + 0] = (float)(J[0] * tot[2] + J[3] * tot[3] + J[6] * tot[4]);
            g_rotvec[3 * (long)p + 1] = (float)(J[1] * tot[2] + J[4] * tot[3] + J[7] * tot[4]);
            g_rotvec[3 * (long)p + 2] = (float)(J[2] * tot[2] + J[5] * tot[3] + J[8] * tot[4]);
        }
    }
}

} // namespace

extern "C" {

int diffus_fan_pose_fwd(const float *median, const float *opening, int opening_stride, const float *rotvec, int n_poses,
                        int n_rays, float *dirs, diffus_stream_t stream)
{
    if (!median || !opening || !dirs || n_poses <= 0 || n_rays <= 0 || (opening_stride != 0 && opening_stride != 1)) return DIFFUS_EINVAL;
    hipLaunchKernelGGL(fan_pose_fwd_kernel, dim3((unsigned)n_poses), dim3(kBlock), 0, (hipStream_t)stream, median, opening, opening_stride,
                       rotvec, n_rays, dirs);
    return last_launch();
}

int diffus_fan_pose_bwd(const float *median, const float *opening, int opening_stride, const float *rotvec, const float *gdirs,
                        int n_poses, int n_rays, float *g_median, float *g_opening, float *g_rotvec, diffus_stream_t stream)
{
    if (!median || !opening || !gdirs || n_poses <= 0 || n_rays <= 0 || (opening_stride != 0 && opening_stride != 1)) return DIFFUS_EINVAL;
    if (g_rotvec && !rotvec) return DIFFUS_EINVAL;
    if (!g_median && !g_opening && !g_rotvec) return DIFFUS_OK;
    hipLaunchKernelGGL(fan_pose_bwd_kernel, dim3((unsigned)n_poses), dim3(kBlock), 0, (hipStream_t)stream, median, opening, opening_stride,
                       rotvec, gdirs, n_rays, g_median, g_opening, g_rotvec);
    return last_launch();
}

} // extern "C"
