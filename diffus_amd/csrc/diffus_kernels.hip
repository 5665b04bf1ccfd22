// diffus_kernels.hip -- gfx950 (MI355X / CDNA4) kernels + C ABI for the DiffUS
// plot_beam_frame hot path.  See include/diffus_hip.h for the boundary and
// DESIGN.md for the data layout and the roofline of each kernel.
//
// Execution model (CDNA4-first, not a port of the reference's ATen call chain):
//   * one 64-lane wavefront marches one ray; lane l owns C = ceil(N1/64)
//     CONSECUTIVE samples n = l*C .. l*C+C-1 of the cropped ray (N1 = S-start);
//   * the reference's N+1 dense solves (src/renderer.py:367-457) collapse to a
//     running product of 2x2 transfer matrices: each lane
//     multiplies its C matrices, the wave does a 6-round Hillis-Steele scan of
//     2x2 products with cross-lane shuffles, each lane then sweeps its chunk from
//     its exclusive prefix.  No A matrix, no LU, no global scratch;
//   * backward recomputes the forward in-kernel and runs the adjoint recursion
//      as a reverse affine scan over the same lanes;
//   * memory is touched in an INTERLEAVED lane mapping (consecutive lanes =
//     consecutive steps) and the scan runs in a CHUNKED one; a per-wave LDS
//     buffer transposes between them;
//   * the volume (and its gradient) can live in a BRICKED layout: 4x4x2-voxel
//     bricks = one 128-B line, so a fan sheet uses whole lines instead of 8 B of
//     each; the gradient scatter is privatised in LDS tiles per patch of rays;
//   * a pose's rays are kept on one XCD (blockIdx remap) so neighbouring rays,
//     which touch the same voxels near the apex, share that XCD's L2.
// This is gather/accumulate work: HBM/L2-bound, no MFMA anywhere.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "diffus_hip.h"

namespace {

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = kWave * kWavesPerBlock;

// ----------------------------------------------------------------------------
// 2x2 matrices  [[a b],[c d]]
struct Mat {
    float a, b, c, d;
};

__device__ __forceinline__ Mat mat_identity() { return Mat{1.f, 0.f, 0.f, 1.f}; }

__device__ __forceinline__ Mat mat_mul(const Mat &x, const Mat &y)
{
    Mat o;
    o.a = __builtin_fmaf(x.a, y.a, x.b * y.c);
    o.b = __builtin_fmaf(x.a, y.b, x.b * y.d);
    o.c = __builtin_fmaf(x.c, y.a, x.d * y.c);
    o.d = __builtin_fmaf(x.c, y.b, x.d * y.d);
    return o;
}

// x * y^T
__device__ __forceinline__ Mat mat_mul_bt(const Mat &x, const Mat &y)
{
    Mat o;
    o.a = __builtin_fmaf(x.a, y.a, x.b * y.b);
    o.b = __builtin_fmaf(x.a, y.c, x.b * y.d);
    o.c = __builtin_fmaf(x.c, y.a, x.d * y.b);
    o.d = __builtin_fmaf(x.c, y.c, x.d * y.d);
    return o;
}

// x^T * y
__device__ __forceinline__ Mat mat_mul_at(const Mat &x, const Mat &y)
{
    Mat o;
    o.a = __builtin_fmaf(x.a, y.a, x.c * y.c);
    o.b = __builtin_fmaf(x.a, y.b, x.c * y.d);
    o.c = __builtin_fmaf(x.b, y.a, x.d * y.c);
    o.d = __builtin_fmaf(x.b, y.b, x.d * y.d);
    return o;
}

// Transfer matrix of one interface, M(r) = [[1-2r^2, r], [-r, 1]] (SURVEY A.3,
// from the rows written at reference src/renderer.py:397-405 with :380-382).
__device__ __forceinline__ Mat mat_of_r(float r) { return Mat{1.f - (2.f * r) * r, r, -r, 1.f}; }

// P * M(r) without forming M
__device__ __forceinline__ Mat mat_step(const Mat &p, float r)
{
    float a = 1.f - (2.f * r) * r;
    Mat o;
    o.a = __builtin_fmaf(p.a, a, -(p.b * r));
    o.b = __builtin_fmaf(p.a, r, p.b);
    o.c = __builtin_fmaf(p.c, a, -(p.d * r));
    o.d = __builtin_fmaf(p.c, r, p.d);
    return o;
}

__device__ __forceinline__ Mat mat_scale(const Mat &m, int e)
{
    return Mat{ldexpf(m.a, e), ldexpf(m.b, e), ldexpf(m.c, e), ldexpf(m.d, e)};
}

// Rescale by an exact power of two so that max|entry| is in [0.5,1).  Returns ex
// with  m_out = m_in * 2^-ex  (0 when m is all zero / non finite).
__device__ __forceinline__ int mat_renorm(Mat &m)
{
    float mx = fmaxf(fmaxf(fabsf(m.a), fabsf(m.b)), fmaxf(fabsf(m.c), fabsf(m.d)));
    // v_frexp_exp_i32_f32 returns 0 for +-0, inf and NaN: no branch needed, ldexp(x, 0) is a no-op
    int ex = __builtin_amdgcn_frexp_expf(mx);
    m = mat_scale(m, -ex);
    return ex;
}

__device__ __forceinline__ bool finitef(float x) { return fabsf(x) < __builtin_inff(); }
__device__ __forceinline__ bool mat_finite(const Mat &m) { return finitef(m.a) && finitef(m.b) && finitef(m.c) && finitef(m.d); }

// Wave-wide min/max of an int by the classic DPP ladder (row_shr 1,2,3,4,8, row_bcast 15,31):
// 7 VALU+DPP steps instead of 6 ds_bpermute round trips.  Result valid in every lane (readlane 63).
template <bool IS_MIN>
__device__ __forceinline__ int wave_reduce_minmax(int v)
{
#define DIFFUS_DPP_STEP(ctrl, rmask, bmask)                                                   \
    {                                                                                         \
        int o = __builtin_amdgcn_update_dpp(v, v, ctrl, rmask, bmask, false);                 \
        v = IS_MIN ? min(v, o) : max(v, o);                                                   \
    }
    DIFFUS_DPP_STEP(0x111, 0xf, 0xf) // row_shr:1
    DIFFUS_DPP_STEP(0x112, 0xf, 0xf) // row_shr:2
    DIFFUS_DPP_STEP(0x113, 0xf, 0xf) // row_shr:3
    DIFFUS_DPP_STEP(0x114, 0xf, 0xe) // row_shr:4
    DIFFUS_DPP_STEP(0x118, 0xf, 0xc) // row_shr:8
    DIFFUS_DPP_STEP(0x142, 0xa, 0xf) // row_bcast:15
    DIFFUS_DPP_STEP(0x143, 0xc, 0xf) // row_bcast:31
#undef DIFFUS_DPP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ Mat mat_shfl_up(const Mat &m, int off)
{
    return Mat{__shfl_up(m.a, off, kWave), __shfl_up(m.b, off, kWave), __shfl_up(m.c, off, kWave),
               __shfl_up(m.d, off, kWave)};
}
__device__ __forceinline__ Mat mat_shfl_down(const Mat &m, int off)
{
    return Mat{__shfl_down(m.a, off, kWave), __shfl_down(m.b, off, kWave), __shfl_down(m.c, off, kWave),
               __shfl_down(m.d, off, kWave)};
}

// ----------------------------------------------------------------------------
struct Pose {
    // source and direction of this wave's ray, kept in both precisions; pmode
    // says which roundings the reference would perform (diffus_oracle.c orc_point)
    float sf[3], df[3];
    double sd[3], dd[3];
    int pmode; // 0: f32 src, f32 dir; 1: f64 src, f32 dir; 2: f64 dir (src any)
};

// PM = 0: both inputs are f32 (the common case) -- compile-time: no f64 code, no mode branches.
// PM = 1: dtypes resolved at run time.
template <int PM = 1>
__device__ __forceinline__ void load_pose(Pose &ps, const void *src, int src_f64, const void *dirs, int dir_f64,
                                          long pose, long ray_lin)
{
    if (PM == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ps.sf[c] = ((const float *)src)[pose * 3 + c];
            ps.df[c] = ((const float *)dirs)[ray_lin * 3 + c];
        }
        ps.pmode = 0;
        return;
    }
    for (int c = 0; c < 3; ++c) {
        if (src_f64) {
            ps.sd[c] = ((const double *)src)[pose * 3 + c];
            ps.sf[c] = (float)ps.sd[c];
        } else {
            ps.sf[c] = ((const float *)src)[pose * 3 + c];
            ps.sd[c] = (double)ps.sf[c];
        }
        if (dir_f64) {
            ps.dd[c] = ((const double *)dirs)[ray_lin * 3 + c];
            ps.df[c] = (float)ps.dd[c];
        } else {
            ps.df[c] = ((const float *)dirs)[ray_lin * 3 + c];
            ps.dd[c] = (double)ps.df[c];
        }
    }
    ps.pmode = dir_f64 ? 2 : (src_f64 ? 1 : 0);
}

// p_c = source_c + float(k) * dir_c with the reference's rounding sequence
// (src/renderer.py:119-124, cast to f32 at :751).  No FMA contraction.
template <int PM = 1>
__device__ __forceinline__ float ray_point(const Pose &ps, int c, int k)
{
    float stepf = (float)k;
    if (PM == 0 || ps.pmode == 0) {
        return __fadd_rn(ps.sf[c], __fmul_rn(stepf, ps.df[c]));
    } else if (ps.pmode == 1) {
        float t = __fmul_rn(stepf, ps.df[c]);
        return (float)__dadd_rn(ps.sd[c], (double)t);
    } else {
        return (float)__dadd_rn(ps.sd[c], __dmul_rn((double)stepf, ps.dd[c]));
    }
}

// round-half-even -> clamp (src/renderer.py:754-756); NaN / beyond-int64 -> 0
// like x86's float->int64 conversion followed by the clamp.
__device__ __forceinline__ int nearest_index(float p, int dim)
{
    float r = rintf(p);
    if (!(r > -9.2233720368547758e18f && r < 9.2233720368547758e18f)) return 0;
    float hi = (float)(dim - 1);
    r = fminf(fmaxf(r, 0.f), hi);
    return (int)r;
}

// ----------------------------------------------------------------------------
// Volume layouts.
//   CANONICAL: the caller's (d0,d1,d2) row-major tensor, dim 2 contiguous.
//   BRICKED:   4x4x2-voxel bricks of 32 floats = one 128-B cache line, bricks in
//              row-major order over (ceil(d0/4), ceil(d1/4), ceil(d2/2)).  Every
//              demo fan lies in a plane of constant dim-2 (reference src/cone.py:258):
//              in the canonical layout each sample then uses 8 bytes of every
//              128-B line it touches; in a brick the same sheet uses the whole line,
//              and neighbouring steps/rays land in the same line.
constexpr int kBrickFloats = 32;

//   PAIRED:    (volume only) 4x4 columns x ONE depth z, each voxel stored as the float2
//              (v[z], v[min(z+1, d2-1)]): every trilinear column is one aligned 8-byte load whatever
//              the parity of z, and a fan sheet uses all 128 bytes of each line it touches.  Twice
//              the memory of the volume; the gradient of a PAIRED volume is BRICKED.
struct Geom {
    int d0, d1, d2;
    int nb1, nb2; // bricks along dim 1 / dim 2
};

// layout of the gradient buffer that goes with a volume layout
template <int LAYOUT>
struct GradLayout {
    static constexpr int value = (LAYOUT == DIFFUS_PAIRED) ? DIFFUS_BRICKED : LAYOUT;
};

// 32-bit element offsets (the host refuses volumes of 2^30 floats or more): 64-bit address
// arithmetic was 28 % of the forward kernel's instructions.
template <int LAYOUT>
__device__ __forceinline__ unsigned vox_off(const Geom &G, int x, int y, int z)
{
    if (LAYOUT == DIFFUS_CANONICAL) {
        return ((unsigned)x * (unsigned)G.d1 + (unsigned)y) * (unsigned)G.d2 + (unsigned)z;
    } else if (LAYOUT == DIFFUS_PAIRED) { // index of the .x half of the pair at (x,y,z)
        unsigned col = ((unsigned)(x >> 2) * (unsigned)G.nb1 + (unsigned)(y >> 2)) * (unsigned)G.d2 + (unsigned)z;
        return col * kBrickFloats + (unsigned)(((x & 3) << 3) | ((y & 3) << 1));
    } else {
        unsigned brick = ((unsigned)(x >> 2) * (unsigned)G.nb1 + (unsigned)(y >> 2)) * (unsigned)G.nb2 + (unsigned)(z >> 1);
        return brick * kBrickFloats + (unsigned)(((x & 3) << 3) | ((y & 3) << 1) | (z & 1));
    }
}

struct Axis {
    int i0, i1;
    float t, m;
};
__device__ __forceinline__ Axis tri_axis(float p, int dim)
{
    Axis a;
    float hi = (float)(dim - 1);
    a.m = (p > 0.f && p < hi) ? 1.f : 0.f;
    float pc = p;
    if (!(pc > 0.f)) pc = 0.f; // also catches NaN
    if (pc > hi) pc = hi;
    float f = floorf(pc);
    a.i0 = (int)f;
    a.t = pc - f;
    a.i1 = min(a.i0 + 1, dim - 1);
    return a;
}

struct __attribute__((packed, aligned(4))) F2 {
    float x, y;
};

// The two dim-2 neighbours of one (x,y) column.  Where they are adjacent in
// memory they come in ONE 8-byte load: half the vector-memory instructions and
// L1 tag lookups of the 8-corner gather.
template <int LAYOUT>
__device__ __forceinline__ void load_zpair(const float *__restrict__ vol, const Geom &G, int x, int y, const Axis &c,
                                           float &lo, float &hi)
{
    if (LAYOUT == DIFFUS_PAIRED) {
        float2 v = *reinterpret_cast<const float2 *>(vol + vox_off<LAYOUT>(G, x, y, c.i0));
        lo = v.x;
        hi = v.y; // = v[min(z0+1, d2-1)] by construction
    } else if (LAYOUT == DIFFUS_CANONICAL) {
        const unsigned row = ((unsigned)x * (unsigned)G.d1 + (unsigned)y) * (unsigned)G.d2;
        if (G.d2 >= 2) {
            int b = min(c.i0, G.d2 - 2);
            F2 v = *reinterpret_cast<const F2 *>(vol + (row + (unsigned)b));
            lo = (c.i0 == b) ? v.x : v.y;
            hi = v.y;
        } else {
            lo = hi = vol[row];
        }
    } else {
        // the aligned pair (z&~1, z|1) of the brick holding z0 in ONE 8-byte load; when z0 is odd
        // its upper neighbour lives in the next brick: one more 4-byte load for those lanes only
        float2 v = *reinterpret_cast<const float2 *>(vol + vox_off<LAYOUT>(G, x, y, c.i0 & ~1));
        if (!(c.i0 & 1)) {
            lo = v.x;
            hi = (c.i1 != c.i0) ? v.y : v.x;
        } else {
            lo = v.y;
            hi = (c.i1 != c.i0) ? vol[vox_off<LAYOUT>(G, x, y, c.i1)] : v.y;
        }
    }
}

struct TriSample {
    float v;          // interpolated impedance
    float g0, g1, g2; // d v / d p (border rule applied)
};

// Trilinear sample at p; lerp order dim 2, dim 1, dim 0, each a + t*(b-a) --
// the exact sequence of oracle/diffus_oracle.c orc_sample_trilinear.
template <int LAYOUT, bool GRAD>
__device__ __forceinline__ TriSample tri_sample(const float *__restrict__ vol, const Geom &G, float p0, float p1,
                                                float p2)
{
    Axis a = tri_axis(p0, G.d0), b = tri_axis(p1, G.d1), c = tri_axis(p2, G.d2);
    float v000, v001, v010, v011, v100, v101, v110, v111;
    load_zpair<LAYOUT>(vol, G, a.i0, b.i0, c, v000, v001);
    load_zpair<LAYOUT>(vol, G, a.i0, b.i1, c, v010, v011);
    load_zpair<LAYOUT>(vol, G, a.i1, b.i0, c, v100, v101);
    load_zpair<LAYOUT>(vol, G, a.i1, b.i1, c, v110, v111);
    float e00 = v001 - v000, e01 = v011 - v010, e10 = v101 - v100, e11 = v111 - v110;
    float c00 = __fadd_rn(v000, __fmul_rn(c.t, e00)), c01 = __fadd_rn(v010, __fmul_rn(c.t, e01));
    float c10 = __fadd_rn(v100, __fmul_rn(c.t, e10)), c11 = __fadd_rn(v110, __fmul_rn(c.t, e11));
    float f0 = c01 - c00, f1 = c11 - c10;
    float q0 = __fadd_rn(c00, __fmul_rn(b.t, f0)), q1 = __fadd_rn(c10, __fmul_rn(b.t, f1));
    float g = q1 - q0;
    TriSample s;
    s.v = __fadd_rn(q0, __fmul_rn(a.t, g));
    if (GRAD) {
        float h0 = __fadd_rn(e00, __fmul_rn(b.t, e01 - e00));
        float h1 = __fadd_rn(e10, __fmul_rn(b.t, e11 - e10));
        s.g0 = g * a.m;
        s.g1 = __fadd_rn(f0, __fmul_rn(a.t, f1 - f0)) * b.m;
        s.g2 = __fadd_rn(h0, __fmul_rn(a.t, h1 - h0)) * c.m;
    } else {
        s.g0 = s.g1 = s.g2 = 0.f;
    }
    return s;
}

// reflection coefficient (reference src/renderer.py:33): IEEE f32 sub, add, div -- used by the
// stage-wise kernels, whose outputs are compared bit for bit with the oracle
__device__ __forceinline__ float reflect(float z1, float z2) { return __fdiv_rn(z2 - z1, z1 + z2); }

// Hot-kernel arithmetic.  The fused kernels are VALU-bound (PMC: VALU busy 57 %, ~4 cycles per
// instruction), and an IEEE f32 division is ~12 instructions, ocml expf ~20.  v_rcp_f32 / v_exp_f32
// are accurate to ~1 ulp, far inside the 1e-5 frame tolerance; 0/0 stays NaN, x/0 stays +-inf.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float reflect_fast(float z1, float z2) { return fast_div(z2 - z1, z1 + z2); }

// ----------------------------------------------------------------------------
struct Args {
    const float *vol;
    Geom G;
    const void *src;
    const void *dirs;
    int src_f64, dir_f64;
    int P, R, S, start, N1;
    float neg_alpha;
    // forward
    float *frame;
    long long *idx;
    // backward
    const float *gframe;
    float *gvol;      // layout that goes with vol's (GradLayout)
    int *gtouched;    // nullable: one flag per gradient brick, set when a launch adds into it (bricked only)
    float *zbar;      // (P,R,N1) d L / d imp per sample, consumed by scatter_patch_kernel
    float *gsrc_part; // (P,R,3) per-ray partials of d/d source
    float *gdirs;
    // start>0 coupling
    float *med;  // (P) median of r[:,start] over rays
    int *who;    // (P) ray that supplied it
    float *gmed; // (P) accumulated d/d median
};

// Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b%8).  Remap so
// that consecutive LOGICAL blocks (= consecutive rays of one pose) sit on one
// XCD and share its L2.  Bijective for any grid size (guide T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk)
{
    unsigned xcd = b & 7u, q = nblk >> 3, rem = nblk & 7u;
    unsigned base = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return base + (b >> 3);
}

// ----------------------------------------------------------------------------
// Two lane <-> sample mappings of one ray's N1 samples over a wave:
//   INTERLEAVED  sample n = j*64 + lane   -- consecutive lanes = consecutive steps:
//                used for everything that touches memory (gathers land in the
//                same bricks / lines, frame & gradient rows are read and written
//                as 256-B runs);
//   CHUNKED      sample n = lane*C + j    -- a lane owns C consecutive samples:
//                used for the scan (15 serial 2x2 products + 6 shuffle rounds
//                instead of 8 x 6 shuffle rounds).
// A per-wave LDS buffer of 64*C floats converts between the two.
__device__ __forceinline__ void wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int C>
__device__ __forceinline__ void to_chunked(float *wb, int lane, const float (&in)[C], float (&out)[C])
{
#pragma unroll
    for (int j = 0; j < C; ++j) wb[j * kWave + lane] = in[j];
    wave_lds_sync();
    if (C >= 4) {
        const float4 *q = reinterpret_cast<const float4 *>(wb + lane * C);
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
            float4 v = q[j];
            out[4 * j] = v.x; out[4 * j + 1] = v.y; out[4 * j + 2] = v.z; out[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) out[j] = wb[lane * C + j];
    }
    wave_lds_sync();
}

template <int C>
__device__ __forceinline__ void to_interleaved(float *wb, int lane, const float (&in)[C], float (&out)[C])
{
    if (C >= 4) {
        float4 *q = reinterpret_cast<float4 *>(wb + lane * C);
#pragma unroll
        for (int j = 0; j < C / 4; ++j) q[j] = make_float4(in[4 * j], in[4 * j + 1], in[4 * j + 2], in[4 * j + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) wb[lane * C + j] = in[j];
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < C; ++j) out[j] = wb[j * kWave + lane];
    wave_lds_sync();
}

// Column / depth parts of a voxel offset: off(x,y,z) = col_off(x,y) + z_off(z).
template <int LAYOUT>
__device__ __forceinline__ unsigned col_off(const Geom &G, int x, int y)
{
    // v_mul_u32_u24 is full rate, v_mul_lo_u32 quarter rate; every factor here is < 2^24 except the
    // final row stride, which is applied as a shift (bricked) or one 32-bit multiply (canonical)
    if (LAYOUT == DIFFUS_CANONICAL)
        return (__umul24((unsigned)x, (unsigned)G.d1) + (unsigned)y) * (unsigned)G.d2;
    if (LAYOUT == DIFFUS_PAIRED)
        return ((__umul24((unsigned)(x >> 2), (unsigned)G.nb1) + (unsigned)(y >> 2)) * (unsigned)G.d2 << 5) +
               (unsigned)(((x & 3) << 3) | ((y & 3) << 1));
    return ((__umul24((unsigned)(x >> 2), (unsigned)G.nb1) + (unsigned)(y >> 2)) * (unsigned)G.nb2 << 5) +
           (unsigned)(((x & 3) << 3) | ((y & 3) << 1));
}
template <int LAYOUT>
__device__ __forceinline__ unsigned z_off(int z)
{
    if (LAYOUT == DIFFUS_CANONICAL) return (unsigned)z;
    if (LAYOUT == DIFFUS_PAIRED) return (unsigned)z << 5;
    return (unsigned)(z >> 1) * kBrickFloats + (unsigned)(z & 1);
}

// lerps of one trilinear sample from its 8 corner values (order 000,001,010,011,100,101,110,111 =
// dim0,dim1,dim2 bits); same operation sequence as oracle/diffus_oracle.c orc_sample_trilinear
template <bool GRAD>
__device__ __forceinline__ TriSample tri_lerp(const float (&v)[8], const Axis &a, const Axis &b, const Axis &c)
{
    float e00 = v[1] - v[0], e01 = v[3] - v[2], e10 = v[5] - v[4], e11 = v[7] - v[6];
    float c00 = __fadd_rn(v[0], __fmul_rn(c.t, e00)), c01 = __fadd_rn(v[2], __fmul_rn(c.t, e01));
    float c10 = __fadd_rn(v[4], __fmul_rn(c.t, e10)), c11 = __fadd_rn(v[6], __fmul_rn(c.t, e11));
    float f0 = c01 - c00, f1 = c11 - c10;
    float q0 = __fadd_rn(c00, __fmul_rn(b.t, f0)), q1 = __fadd_rn(c10, __fmul_rn(b.t, f1));
    float g = q1 - q0;
    TriSample s;
    s.v = __fadd_rn(q0, __fmul_rn(a.t, g));
    if (GRAD) {
        float h0 = __fadd_rn(e00, __fmul_rn(b.t, e01 - e00));
        float h1 = __fadd_rn(e10, __fmul_rn(b.t, e11 - e10));
        s.g0 = g * a.m;
        s.g1 = __fadd_rn(f0, __fmul_rn(a.t, f1 - f0)) * b.m;
        s.g2 = __fadd_rn(h0, __fmul_rn(a.t, h1 - h0)) * c.m;
    } else {
        s.g0 = s.g1 = s.g2 = 0.f;
    }
    return s;
}

#ifndef DIFFUS_GATHER_GROUP
#define DIFFUS_GATHER_GROUP 8
#endif
// Impedance (and for the trilinear backward its spatial gradient) at the wave's samples,
// INTERLEAVED mapping.  Written for memory-level parallelism: phase A computes the addresses of
// up to 8 samples x 8 corners and issues every load with NO branch in between (lanes past the
// end of the ray re-read the last sample instead of being masked), phase B recomputes the cheap
// interpolation weights and consumes the values.  The first version (load -> use per corner,
// behind exec-mask branches) made hipcc emit `s_waitcnt vmcnt(0)` after almost every load:
// ~50 dependent memory round trips per wave, 43 % of wave time in SQ_WAIT_ANY.
template <int C, int SAMPLER, int LAYOUT, bool GRAD, int PM>
__device__ __forceinline__ void gather_interleaved(const Args &A, const Pose &ps, int lane, float (&z)[C],
                                                   float (&g0)[C], float (&g1)[C], float (&g2)[C])
{
    constexpr int G = (C < DIFFUS_GATHER_GROUP) ? C : DIFFUS_GATHER_GROUP;
    constexpr int NV = (SAMPLER == DIFFUS_NEAREST) ? 1 : 8;
    const float *__restrict__ vol = A.vol;
#pragma unroll
    for (int gb = 0; gb < C; gb += G) {
        float raw[G][NV];
        float ta[G], tb[G], tc[G]; // interpolation weights, kept for phase B
        unsigned mk[G];            // border-rule bits of the three axes (gradient only)
        // ---- phase A: addresses + loads
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            int n = min((gb + jj) * kWave + lane, A.N1 - 1);
            int k = A.start + n;
            float p0 = ray_point<PM>(ps, 0, k), p1 = ray_point<PM>(ps, 1, k), p2 = ray_point<PM>(ps, 2, k);
            if constexpr (SAMPLER == DIFFUS_NEAREST) {
                int i0 = nearest_index(p0, A.G.d0), i1 = nearest_index(p1, A.G.d1), i2 = nearest_index(p2, A.G.d2);
                raw[jj][0] = vol[col_off<LAYOUT>(A.G, i0, i1) + z_off<LAYOUT>(i2)];
            } else {
                Axis a = tri_axis(p0, A.G.d0), b = tri_axis(p1, A.G.d1), c = tri_axis(p2, A.G.d2);
                unsigned c00 = col_off<LAYOUT>(A.G, a.i0, b.i0), c01 = col_off<LAYOUT>(A.G, a.i0, b.i1);
                unsigned c10 = col_off<LAYOUT>(A.G, a.i1, b.i0), c11 = col_off<LAYOUT>(A.G, a.i1, b.i1);
                unsigned z0 = z_off<LAYOUT>(c.i0), z1 = z_off<LAYOUT>(c.i1);
                if constexpr (LAYOUT == DIFFUS_PAIRED) { // 4 aligned 8-byte loads: (z0, z0+1) of each column
                    float2 q00 = *reinterpret_cast<const float2 *>(vol + (c00 + z0));
                    float2 q01 = *reinterpret_cast<const float2 *>(vol + (c01 + z0));
                    float2 q10 = *reinterpret_cast<const float2 *>(vol + (c10 + z0));
                    float2 q11 = *reinterpret_cast<const float2 *>(vol + (c11 + z0));
                    raw[jj][0] = q00.x; raw[jj][1] = q00.y; raw[jj][2] = q01.x; raw[jj][3] = q01.y;
                    raw[jj][4] = q10.x; raw[jj][5] = q10.y; raw[jj][6] = q11.x; raw[jj][7] = q11.y;
                    (void)z1;
                } else {
#ifdef DIFFUS_ABLATE_LOADS
                raw[jj][0] = __uint_as_float(c00 + z0); raw[jj][1] = __uint_as_float(c00 + z1);
                raw[jj][2] = __uint_as_float(c01 + z0); raw[jj][3] = __uint_as_float(c01 + z1);
                raw[jj][4] = __uint_as_float(c10 + z0); raw[jj][5] = __uint_as_float(c10 + z1);
                raw[jj][6] = __uint_as_float(c11 + z0); raw[jj][7] = __uint_as_float(c11 + z1);
#else
                raw[jj][0] = vol[c00 + z0]; raw[jj][1] = vol[c00 + z1];
                raw[jj][2] = vol[c01 + z0]; raw[jj][3] = vol[c01 + z1];
                raw[jj][4] = vol[c10 + z0]; raw[jj][5] = vol[c10 + z1];
                raw[jj][6] = vol[c11 + z0]; raw[jj][7] = vol[c11 + z1];
#endif
                }
                ta[jj] = a.t; tb[jj] = b.t; tc[jj] = c.t;
                if (GRAD) mk[jj] = (a.m != 0.f ? 1u : 0u) | (b.m != 0.f ? 2u : 0u) | (c.m != 0.f ? 4u : 0u);
            }
        }
        // ---- phase B: interpolation
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int j = gb + jj;
            const bool live = j * kWave + lane < A.N1;
            if constexpr (SAMPLER == DIFFUS_NEAREST) {
                z[j] = live ? raw[jj][0] : 1.f;
                if (GRAD) g0[j] = g1[j] = g2[j] = 0.f;
            } else {
                Axis a, b, c;
                a.t = ta[jj]; b.t = tb[jj]; c.t = tc[jj];
                if (GRAD) {
                    a.m = (mk[jj] & 1u) ? 1.f : 0.f;
                    b.m = (mk[jj] & 2u) ? 1.f : 0.f;
                    c.m = (mk[jj] & 4u) ? 1.f : 0.f;
                }
#ifdef DIFFUS_ABLATE_LERP
                TriSample sm;
                sm.v = raw[jj][0] + raw[jj][1] + raw[jj][2] + raw[jj][3] + raw[jj][4] + raw[jj][5] + raw[jj][6] + raw[jj][7] + a.t + b.t + c.t;
                sm.g0 = sm.g1 = sm.g2 = 0.f;
#else
                TriSample sm = tri_lerp<GRAD>(raw[jj], a, b, c);
#endif
                z[j] = live ? sm.v : 1.f;
                if (GRAD) {
                    g0[j] = live ? sm.g0 : 0.f;
                    g1[j] = live ? sm.g1 : 0.f;
                    g2[j] = live ? sm.g2 : 0.f;
                }
            }
        }
    }
}

// r'_{n-1} for the lane's samples (CHUNKED): r[j] couples sample n-1 and n
// (n = lane*C+j).  r = 0 (identity transfer matrix) for n = 0 and n >= N1; with
// start > 0 the first kept coefficient is replaced by the per-pose median
// (reference :243-244).
template <int C>
__device__ __forceinline__ void reflect_chunk(const Args &A, int n0, const float (&z)[C], float zprev, float medv,
                                              float (&r)[C])
{
#pragma unroll
    for (int j = 0; j < C; ++j) {
        int n = n0 + j;
        float zp = (j == 0) ? zprev : z[j == 0 ? 0 : j - 1];
        float v = reflect_fast(zp, z[j]);
        if (n == 1 && A.start > 0) v = medv;
        r[j] = (n >= 1 && n < A.N1) ? v : 0.f;
    }
}

// Echo series of one ray spread over a wave (replaces the N+1
// dense solves of reference src/renderer.py:367-457).  r[j] is the reflection
// coefficient entering sample n = lane*C + j (0 where there is none); e[j] gets
// echo_n = (P_n)01/(P_n)11 with NaN -> 0 (reference :408).
template <int C, bool FAST = false>
__device__ __forceinline__ void echo_chunk(const float (&r)[C], int lane, float (&e)[C])
{
    // local product of the chunk, then inclusive scan over lanes (lower lanes on the left)
    Mat L = mat_identity();
#pragma unroll
    for (int j = 0; j < C; ++j) {
        L = mat_step(L, r[j]);
        mat_renorm(L);
    }
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        Mat o = mat_shfl_up(L, off);
        if (lane >= off) {
            L = mat_mul(o, L);
            mat_renorm(L);
        }
    }
    Mat Pm = mat_shfl_up(L, 1);
    if (lane == 0) Pm = mat_identity();
#pragma unroll
    for (int j = 0; j < C; ++j) {
        Pm = mat_step(Pm, r[j]);
        mat_renorm(Pm);
        float v = FAST ? fast_div(Pm.b, Pm.d) : __fdiv_rn(Pm.b, Pm.d);
        e[j] = (v == v) ? v : 0.f; // nan_to_num(nan=0)
    }
}

// ----------------------------------------------------------------------------
// FORWARD  (replaces reference src/renderer.py:201-275 with artifacts=False)
#ifndef DIFFUS_FWD_MIN_WAVES
#define DIFFUS_FWD_MIN_WAVES 1
#endif
template <int C, int SAMPLER, int LAYOUT, int WPB, int PM>
__global__ __launch_bounds__(kWave *WPB, (C <= 8 ? DIFFUS_FWD_MIN_WAVES : 1)) void render_fwd_kernel(Args A)
{
    __shared__ __attribute__((aligned(16))) float lds[WPB][kWave * C];
    const int wib = threadIdx.x >> 6;
    const long w = (long)xcd_remap(blockIdx.x, gridDim.x) * WPB + wib;
    if (w >= (long)A.P * A.R) return; // wave-uniform; no block-level barrier below
    const int lane = threadIdx.x & 63;
    const long pose = w / A.R;
    const int n0 = lane * C;
    float *wb = lds[wib];

    Pose ps;
    load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);

    float zi[C], z[C], r[C], e[C], u0[C], u1[C], u2[C];
#ifdef DIFFUS_ABLATE_GATHER
#pragma unroll
    for (int j = 0; j < C; ++j) zi[j] = ps.sf[0] + (float)(j * kWave + lane) * ps.df[1];
#else
    gather_interleaved<C, SAMPLER, LAYOUT, false, PM>(A, ps, lane, zi, u0, u1, u2);
#endif
#ifdef DIFFUS_ABLATE_TRANSPOSE
#pragma unroll
    for (int j = 0; j < C; ++j) z[j] = zi[j];
#else
    to_chunked<C>(wb, lane, zi, z);
#endif
    float zprev = __shfl_up(z[C - 1], 1, kWave);
    float medv = (A.start > 0) ? A.med[pose] : 0.f;
    reflect_chunk<C>(A, n0, z, zprev, medv, r);
#ifdef DIFFUS_ABLATE_SCAN
#pragma unroll
    for (int j = 0; j < C; ++j) e[j] = r[j];
#else
    echo_chunk<C, true>(r, lane, e);
#endif
#pragma unroll
    for (int j = 0; j < C; ++j) {
        // attenuation, reference :256-259: f32(-alpha) * f32(n), exp, multiply
        float att = fast_exp(__fmul_rn(A.neg_alpha, (float)(n0 + j)));
        e[j] = __fmul_rn(e[j], att);
    }
#ifdef DIFFUS_ABLATE_TRANSPOSE
#pragma unroll
    for (int j = 0; j < C; ++j) zi[j] = e[j];
#else
    to_interleaved<C>(wb, lane, e, zi);
#endif
    float *out = A.frame + w * A.N1;
#ifdef DIFFUS_ABLATE_STORE
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) acc += zi[j];
    if (acc == 123.456f) out[lane] = acc;
#else
#pragma unroll
    for (int j = 0; j < C; ++j) {
        int n = j * kWave + lane;
        if (n < A.N1) out[n] = zi[j];
    }
#endif

    if (A.idx) {
        const long plane = (long)A.P * A.R * A.N1;
        long long *ix = A.idx + w * A.N1;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = j * kWave + lane;
            if (n < A.N1) {
                int k = A.start + n;
                ix[n] = nearest_index(ray_point<PM>(ps, 0, k), A.G.d0);
                ix[plane + n] = nearest_index(ray_point<PM>(ps, 1, k), A.G.d1);
                ix[2 * plane + n] = nearest_index(ray_point<PM>(ps, 2, k), A.G.d2);
            }
        }
    }
}

// ----------------------------------------------------------------------------
// BACKWARD.  Notation (indices in cropped coordinates):
//   T_n = M(r'_{n-1}),  P_n = P_{n-1} T_n,  echo_n = b_n/d_n,  (b_n,d_n) = 2nd column of P_n
//   gbar_n = gframe_n * att_n;  Gbar_n = (gbar_n/d_n) [[0,1],[0,-echo_n]]
//   U_{n-1} = (Gbar_n + U_n) T_n^T, U_N = 0;   Tbar_n = P_{n-1}^T (Gbar_n + U_n)
//   rbar = -4 r Tbar_00 + Tbar_01 - Tbar_10
// All P are carried rescaled by exact powers of two (P'_n = 2^{e_n} P_n); with
// W_n = 2^{e_n-e_{n-1}} (Gbar'_n + U'_n):  Tbar_n = P'_{n-1}^T W_n,  U'_{n-1} = W_n T_n^T.
// The chunk of one lane is an affine map U_in -> U_out; lanes are combined with a
// reverse Hillis-Steele scan of affine maps (A, B, beta):  X -> A + X (B 2^beta)^T.
#ifndef DIFFUS_BWD_MIN_WAVES
#define DIFFUS_BWD_MIN_WAVES 1
#endif
template <int C, int SAMPLER, int LAYOUT, bool GPOSE, int WPB, int PM>
__global__ __launch_bounds__(kWave *WPB, (C <= 8 ? DIFFUS_BWD_MIN_WAVES : 1)) void render_bwd_kernel(Args A)
{
    __shared__ __attribute__((aligned(16))) float lds[WPB][kWave * C];
    constexpr bool KEEP_GRAD = GPOSE && (C < 16); // C = 16: re-gather at the end instead of 48 more registers
    const int wib = threadIdx.x >> 6;
    const long w = (long)xcd_remap(blockIdx.x, gridDim.x) * WPB + wib;
    if (w >= (long)A.P * A.R) return;
    const int lane = threadIdx.x & 63;
    const long pose = w / A.R;
    const int n0 = lane * C;
    float *wb = lds[wib];

    Pose ps;
    load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);

    float zi[C], gi0[C], gi1[C], gi2[C], z[C], r[C], gb[C];
    gather_interleaved<C, SAMPLER, LAYOUT, KEEP_GRAD, PM>(A, ps, lane, zi, gi0, gi1, gi2);
    to_chunked<C>(wb, lane, zi, z);
    {
        // upstream gradient row, read as 256-B runs, attenuation folded in
        const float *gin = A.gframe + w * A.N1;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = j * kWave + lane;
            zi[j] = (n < A.N1) ? gin[n] * fast_exp(__fmul_rn(A.neg_alpha, (float)n)) : 0.f;
        }
        to_chunked<C>(wb, lane, zi, gb);
    }
    const float zprev = __shfl_up(z[C - 1], 1, kWave);
    const float medv = (A.start > 0) ? A.med[pose] : 0.f;
    reflect_chunk<C>(A, n0, z, zprev, medv, r);

    // ---- forward recompute with exponent tracking ----
    Mat L = mat_identity();
    int lam = 0; // L' = L * 2^lam
#pragma unroll
    for (int j = 0; j < C; ++j) {
        L = mat_step(L, r[j]);
        lam -= mat_renorm(L);
    }
    const Mat Lloc = L;   // normalised local product T_first..T_last
    const int lamloc = lam;
    int iota = lam;       // inclusive prefix exponent
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        Mat o = mat_shfl_up(L, off);
        int oe = __shfl_up(iota, off, kWave);
        if (lane >= off) {
            L = mat_mul(o, L);
            iota = oe + iota - mat_renorm(L);
        }
    }
    Mat Pm = mat_shfl_up(L, 1);
    int eps = __shfl_up(iota, 1, kWave); // exponent of the exclusive prefix
    if (lane == 0) {
        Pm = mat_identity();
        eps = 0;
    }

    Mat Pin[C];  // P'_{n-1} as used by this lane
    int ex[C];   // renorm exponent of step n
    float gu[C]; // gbar_n / d'_n
    float rho[C];
    int esum = 0;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        Pin[j] = Pm;
        Pm = mat_step(Pm, r[j]);
        ex[j] = mat_renorm(Pm);
        esum += ex[j];
        float rd = __builtin_amdgcn_rcpf(Pm.d);
        float e = Pm.b * rd;
        float g = (e == e) ? gb[j] : 0.f; // echoes zeroed by nan_to_num are constants
        float q = g * rd;
        gu[j] = (g != 0.f) ? q : 0.f;
        rho[j] = (e == e) ? e : 0.f;
        // past a non-finite reflection coefficient every echo is the constant 0
        if (!finitef(r[j]) || !mat_finite(Pin[j]) || !finitef(gu[j]) || !finitef(rho[j])) gu[j] = 0.f;
    }

    // exponent of this lane's last P' and the hop to the next lane's exclusive prefix
    const int elast = eps - esum;
    int eps_next = __shfl_down(eps, 1, kWave);
    const int delta = (lane == kWave - 1) ? 0 : (eps_next - elast);

    // ---- lane-local affine map: A-part = sweep from U = 0 ----
    auto sweep = [&](Mat U, float *rbar) {
#pragma unroll
        for (int j = C - 1; j >= 0; --j) {
            Mat W;
            W.a = U.a;
            W.b = U.b + gu[j];
            W.c = U.c;
            W.d = U.d - gu[j] * rho[j];
            W = mat_scale(W, -ex[j]);
            if (rbar) {
                Mat Tb = mat_mul_at(Pin[j], W);
                rbar[j] = __builtin_fmaf(-4.f * r[j], Tb.a, Tb.b - Tb.c);
            }
            float rr = finitef(r[j]) ? r[j] : 0.f; // non-finite step: cut the chain (U is zero there anyway)
            U = mat_mul_bt(W, mat_of_r(rr));
        }
        return U;
    };
    Mat Aacc = sweep(Mat{0.f, 0.f, 0.f, 0.f}, nullptr);
    Mat Bn = Lloc;                    // normalised linear part
    int beta = delta - esum - lamloc; // B = Bn * 2^beta
    if (!mat_finite(Bn)) Bn = Mat{0.f, 0.f, 0.f, 0.f}; // a non-finite chunk passes nothing

    // ---- reverse inclusive scan of affine maps: G_l = F_l o F_{l+1} o ... o F_63 ----
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        Mat oA = mat_shfl_down(Aacc, off);
        Mat oB = mat_shfl_down(Bn, off);
        int ob = __shfl_down(beta, off, kWave);
        if (lane + off < kWave) {
            Mat t = mat_scale(mat_mul_bt(oA, Bn), beta);
            Aacc.a += t.a; Aacc.b += t.b; Aacc.c += t.c; Aacc.d += t.d;
            Bn = mat_mul(Bn, oB);
            beta = beta + ob + mat_renorm(Bn); // B = Bn * 2^beta: a rescale of Bn by 2^-ex adds ex
        }
    }
    Mat Uin = mat_shfl_down(Aacc, 1);
    if (lane == kWave - 1) Uin = Mat{0.f, 0.f, 0.f, 0.f};
    Uin = mat_scale(Uin, delta);

    float rbar[C];
    sweep(Uin, rbar);

    // ---- rbar -> zbar (d r / d Z of reference :33) ----
    float zbar[C];
#pragma unroll
    for (int j = 0; j < C; ++j) zbar[j] = 0.f;
    float carry = 0.f, gmed_lane = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        int n = n0 + j;
        bool live = (n >= 1 && n < A.N1);
        float rb = live ? rbar[j] : 0.f;
        if (!finitef(rb)) rb = 0.f; // drop non-finite
        if (n == 1 && A.start > 0) {
            gmed_lane = rb;
            rb = 0.f;
        }
        float zp = (j == 0) ? zprev : z[j == 0 ? 0 : j - 1];
        float s = zp + z[j];
        float inv = __builtin_amdgcn_rcpf(s);
        float dz = 2.f * zp * inv * inv;     // d r / d Z_n
        float dzp = -2.f * z[j] * inv * inv; // d r / d Z_{n-1}
        float c1 = rb * dz, c0 = rb * dzp;
        if (!finitef(c1)) c1 = 0.f;
        if (!finitef(c0)) c0 = 0.f;
        zbar[j] += c1;
        if (j == 0)
            carry = c0;
        else
            zbar[j == 0 ? 0 : j - 1] += c0;
    }
    float cin = __shfl_down(carry, 1, kWave);
    if (lane != kWave - 1) zbar[C - 1] += cin;
    if (gmed_lane != 0.f) atomicAdd(&A.gmed[pose], gmed_lane);

    // ---- back to INTERLEAVED: hand zbar to the scatter kernel, reduce the pose gradient ----
    // The volume scatter is a separate launch (scatter_patch_kernel): its thread <-> sample
    // mapping is chosen for LDS privatisation, not for the scan.
    to_interleaved<C>(wb, lane, zbar, zi);
    if (A.zbar) {
        float *zo = A.zbar + w * A.N1;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = j * kWave + lane;
            if (n < A.N1) zo[n] = zi[j];
        }
    }
    if (GPOSE) {
        float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gd0 = 0.f, gd1 = 0.f, gd2 = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = j * kWave + lane;
            float zb = zi[j];
            if (n < A.N1 && zb != 0.f) {
                int k = A.start + n;
                float q0, q1, q2;
                if (KEEP_GRAD) {
                    q0 = gi0[j]; q1 = gi1[j]; q2 = gi2[j];
                } else {
                    TriSample s = tri_sample<LAYOUT, true>(A.vol, A.G, ray_point<PM>(ps, 0, k), ray_point<PM>(ps, 1, k),
                                                           ray_point<PM>(ps, 2, k));
                    q0 = s.g0; q1 = s.g1; q2 = s.g2;
                }
                float kf = (float)k;
                float a0 = zb * q0, a1 = zb * q1, a2 = zb * q2;
                gs0 += a0; gs1 += a1; gs2 += a2;
                gd0 = __builtin_fmaf(kf, a0, gd0);
                gd1 = __builtin_fmaf(kf, a1, gd1);
                gd2 = __builtin_fmaf(kf, a2, gd2);
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            gs0 += __shfl_xor(gs0, off, kWave);
            gs1 += __shfl_xor(gs1, off, kWave);
            gs2 += __shfl_xor(gs2, off, kWave);
            gd0 += __shfl_xor(gd0, off, kWave);
            gd1 += __shfl_xor(gd1, off, kWave);
            gd2 += __shfl_xor(gd2, off, kWave);
        }
        if (lane == 0) {
            if (A.gsrc_part) {
                A.gsrc_part[w * 3 + 0] = gs0;
                A.gsrc_part[w * 3 + 1] = gs1;
                A.gsrc_part[w * 3 + 2] = gs2;
            }
            if (A.gdirs) {
                A.gdirs[w * 3 + 0] = gd0;
                A.gdirs[w * 3 + 1] = gd1;
                A.gdirs[w * 3 + 2] = gd2;
            }
        }
    }
}

// ----------------------------------------------------------------------------
// VOLUME SCATTER.  gvol += sum over samples of zbar * (interpolation weights).
// Naive per-sample global float atomics run ~10x below even the scattered-atomic
// rate because every fan hammers the few hundred voxels around its apex
// (measured: 17.5 ms for 33 M atomics at config 3).  Instead a block takes a PATCH
// of kPatchRays adjacent rays x kPatchSteps consecutive steps of one pose, whose
// footprint is a small box of voxels; it accumulates the patch into an LDS tile
// covering that box (ds_add_f32) and flushes each touched voxel ONCE.  In the
// bricked layout the tile is a box of whole bricks, so the flush is made of
// 128-B contiguous atomic runs (the full-rate shape of global_atomic_add_f32).
// Patches whose box does not fit the tile fall back to direct global atomics.
#ifndef DIFFUS_PATCH_RAYS
#define DIFFUS_PATCH_RAYS 16
#endif
#ifndef DIFFUS_PATCH_STEPS
#define DIFFUS_PATCH_STEPS 64
#endif
#ifndef DIFFUS_TILE_CAP
#define DIFFUS_TILE_CAP (12 * 1024)
#endif
#ifdef DIFFUS_STAMP // diagnostic build only (tools/): per-block phase timestamps of the scatter kernel
__device__ unsigned long long *g_stamps = nullptr;
#define STAMP(i)                                                                          \
    do {                                                                                  \
        if (threadIdx.x == 0 && g_stamps) g_stamps[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#else
#define STAMP(i) ((void)0)
#endif
constexpr int kPatchRays = DIFFUS_PATCH_RAYS;
constexpr int kPatchSteps = DIFFUS_PATCH_STEPS;
constexpr int kTileCap = DIFFUS_TILE_CAP; // floats (48 KiB: 3 blocks per CU)
constexpr int kSamplesPerThread = kPatchRays * kPatchSteps / kBlock; // 4

struct Cell {
    int i0[3], i1[3];
    float t[3];
};

template <int SAMPLER, int PM = 1>
__device__ __forceinline__ Cell cell_of(const Args &A, const Pose &ps, int k)
{
    Cell c;
    const int dims[3] = {A.G.d0, A.G.d1, A.G.d2};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float p = ray_point<PM>(ps, a, k);
        if (SAMPLER == DIFFUS_NEAREST) {
            c.i0[a] = c.i1[a] = nearest_index(p, dims[a]);
            c.t[a] = 0.f;
        } else {
            Axis ax = tri_axis(p, dims[a]);
            c.i0[a] = ax.i0;
            c.i1[a] = ax.i1;
            c.t[a] = ax.t;
        }
    }
    return c;
}

template <int SAMPLER, typename F>
__device__ __forceinline__ void for_each_corner(const Cell &c, float zb, F &&f)
{
    if (SAMPLER == DIFFUS_NEAREST) {
        f(c.i0[0], c.i0[1], c.i0[2], zb);
    } else {
        float wa1 = c.t[0], wa0 = 1.f - wa1, wb1 = c.t[1], wb0 = 1.f - wb1, wc1 = c.t[2], wc0 = 1.f - wc1;
        float w00 = zb * wa0 * wb0, w01 = zb * wa0 * wb1, w10 = zb * wa1 * wb0, w11 = zb * wa1 * wb1;
        f(c.i0[0], c.i0[1], c.i0[2], w00 * wc0);
        f(c.i0[0], c.i0[1], c.i1[2], w00 * wc1);
        f(c.i0[0], c.i1[1], c.i0[2], w01 * wc0);
        f(c.i0[0], c.i1[1], c.i1[2], w01 * wc1);
        f(c.i1[0], c.i0[1], c.i0[2], w10 * wc0);
        f(c.i1[0], c.i0[1], c.i1[2], w10 * wc1);
        f(c.i1[0], c.i1[1], c.i0[2], w11 * wc0);
        f(c.i1[0], c.i1[1], c.i1[2], w11 * wc1);
    }
}

// tile units: voxels (canonical) or bricks (bricked)
template <int LAYOUT>
__device__ __forceinline__ int tile_unit(int v, int axis)
{
    if (LAYOUT == DIFFUS_CANONICAL) return v;
    return axis == 2 ? (v >> 1) : (v >> 2);
}

template <int SAMPLER, int LAYOUT, int PM>
__global__ __launch_bounds__(kBlock) void scatter_patch_kernel(Args A, int ray_groups, int step_groups)
{
    // Measured on gfx950 (tools/lds_atomic_bench.hip): ds_add_f32 costs ~194 cycles per
    // wave-instruction whatever the addresses (lanes are serialised), ds_add_u32 5-15.
    // The tile therefore accumulates in 32-bit FIXED POINT with a per-patch power-of-two
    // scale 2^fx chosen so that even all 1024 samples landing on one voxel cannot
    // overflow: (sum over the patch of |zbar|) * 2^fx < 2^30 (weights are <= 1, so no voxel can
    // receive more than that sum).  Quantum <= 2^-20 of the patch's largest contribution,
    // typically 2^-23..2^-26; integer adds commute, so a tile sum is bitwise reproducible.
    __shared__ int tile[kTileCap];
    __shared__ int s_lo[3], s_hi[3], s_max;
    __shared__ float s_sum[kWavesPerBlock];
    constexpr int UNIT = (LAYOUT == DIFFUS_CANONICAL) ? 1 : kBrickFloats; // floats per tile unit

    // patch -> (pose, ray group, step group); the XCD remap keeps a pose on one XCD
    const unsigned Lb = xcd_remap(blockIdx.x, gridDim.x);
    const int sg = Lb % step_groups;
    const int rg = (Lb / step_groups) % ray_groups;
    const int pose = Lb / (step_groups * ray_groups);
    const int tid = threadIdx.x;
    // thread -> ray (tid / 16) and 4 consecutive steps ((tid % 16) * 4 ..)
    const int ray = rg * kPatchRays + tid / (kPatchSteps / kSamplesPerThread);
    const int nbase = sg * kPatchSteps + (tid % (kPatchSteps / kSamplesPerThread)) * kSamplesPerThread;
    const bool ray_ok = ray < A.R;
    const long w = (long)pose * A.R + (ray_ok ? ray : 0);

    STAMP(0);
    if (tid < 3) {
        s_lo[tid] = 0x7fffffff;
        s_hi[tid] = -1;
    }
    if (tid == 3) s_max = 0;
    Pose ps;
    load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
    Cell cells[kSamplesPerThread];
    float zb[kSamplesPerThread];
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    float zmax = 0.f;
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) {
        int n = nbase + q;
        zb[q] = 0.f;
        if (ray_ok && n < A.N1) zb[q] = A.zbar[w * A.N1 + n];
        if (!finitef(zb[q])) zb[q] = 0.f;
        cells[q] = cell_of<SAMPLER, PM>(A, ps, A.start + n);
        zmax = fmaxf(zmax, fabsf(zb[q]));
        if (zb[q] != 0.f) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = min(lo[a], tile_unit<LAYOUT>(cells[q].i0[a], a));
                hi[a] = max(hi[a], tile_unit<LAYOUT>(cells[q].i1[a], a));
            }
        }
    }
    STAMP(1);
    // block bounding box: DPP wave reduce, then one LDS atomic per wave
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_minmax<true>(lo[a]);
        hi[a] = wave_reduce_minmax<false>(hi[a]);
    }
    zmax = __int_as_float(wave_reduce_minmax<false>(__float_as_int(zmax))); // zmax >= 0: bits order like floats
    float zsum = 0.f;
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q) zsum += fabsf(zb[q]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) zsum += __shfl_xor(zsum, off, kWave);
    __syncthreads();
    if ((tid & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            atomicMin(&s_lo[a], lo[a]);
            atomicMax(&s_hi[a], hi[a]);
        }
        atomicMax(&s_max, __float_as_int(zmax)); // non-negative floats order like their bit patterns
        s_sum[tid >> 6] = zsum;
    }
    __syncthreads();
    STAMP(2);
    const int l0 = s_lo[0], l1 = s_lo[1], l2 = s_lo[2];
    if (s_hi[0] < 0) return; // nothing to add in this patch (block-uniform)
    const int b0 = s_hi[0] - l0 + 1, b1 = s_hi[1] - l1 + 1, b2 = s_hi[2] - l2 + 1;
    const long vol_tile = (long)b0 * b1 * b2 * UNIT;

    if (vol_tile > kTileCap) { // block-uniform fallback: direct atomics
#pragma unroll
        for (int q = 0; q < kSamplesPerThread; ++q)
            if (zb[q] != 0.f)
                for_each_corner<SAMPLER>(cells[q], zb[q], [&](int i, int j, int k, float v) {
                    if (v != 0.f) {
                        unsigned g = vox_off<LAYOUT>(A.G, i, j, k);
                        atomicAdd(A.gvol + g, v);
                        if (LAYOUT == DIFFUS_BRICKED && A.gtouched) A.gtouched[g >> 5] = 1;
                    }
                });
        return;
    }
    const int nt = (int)vol_tile;
    // (s_sum total) * 2^fx in [2^28, 2^29): headroom for the rounding of each contribution
    const float ztot = fmaxf((s_sum[0] + s_sum[1]) + (s_sum[2] + s_sum[3]), __int_as_float(s_max));
    const int fx = 29 - __builtin_amdgcn_frexp_expf(ztot);
    for (int e = tid; e < nt; e += kBlock) tile[e] = 0;
    __syncthreads();
    STAMP(3);
    // tile index = ex(i) + ey(j) + ez(k): three separable parts, each evaluated for the two
    // coordinates of its axis only (6 small computations per sample instead of 8 full ones)
    auto part = [&](int v, int axis) -> int {
        if (LAYOUT == DIFFUS_CANONICAL)
            return axis == 0 ? (v - l0) * b1 * b2 : (axis == 1 ? (v - l1) * b2 : (v - l2));
        return axis == 0 ? (((v >> 2) - l0) * b1 * b2 * kBrickFloats + ((v & 3) << 3))
                         : (axis == 1 ? (((v >> 2) - l1) * b2 * kBrickFloats + ((v & 3) << 1))
                                      : (((v >> 1) - l2) * kBrickFloats + (v & 1)));
    };
#pragma unroll
    for (int q = 0; q < kSamplesPerThread; ++q)
        if (zb[q] != 0.f) {
            const Cell &c = cells[q];
            const float sc = ldexpf(zb[q], fx);
            if (SAMPLER == DIFFUS_NEAREST) {
                atomicAdd(&tile[part(c.i0[0], 0) + part(c.i0[1], 1) + part(c.i0[2], 2)], __float2int_rn(sc));
            } else {
                const int ex0 = part(c.i0[0], 0), ex1 = part(c.i1[0], 0), ey0 = part(c.i0[1], 1), ey1 = part(c.i1[1], 1),
                          ez0 = part(c.i0[2], 2), ez1 = part(c.i1[2], 2);
                const float wa1 = c.t[0], wa0 = 1.f - wa1, wb1 = c.t[1], wb0 = 1.f - wb1, wc1 = c.t[2], wc0 = 1.f - wc1;
                const float w00 = sc * wa0 * wb0, w01 = sc * wa0 * wb1, w10 = sc * wa1 * wb0, w11 = sc * wa1 * wb1;
                // clamped samples (outside the volume: more than half of a typical fan) have zero
                // weights on half or more of their corners: do not spend an LDS atomic on a zero
                auto add = [&](int e, float v) {
                    int q = __float2int_rn(v);
                    if (q != 0) atomicAdd(&tile[e], q);
                };
                add(ex0 + ey0 + ez0, w00 * wc0);
                add(ex0 + ey0 + ez1, w00 * wc1);
                add(ex0 + ey1 + ez0, w01 * wc0);
                add(ex0 + ey1 + ez1, w01 * wc1);
                add(ex1 + ey0 + ez0, w10 * wc0);
                add(ex1 + ey0 + ez1, w10 * wc1);
                add(ex1 + ey1 + ez0, w11 * wc0);
                add(ex1 + ey1 + ez1, w11 * wc1);
            }
        }
    __syncthreads();
    STAMP(4);
    // Flush every touched entry once.  No integer division per entry (the first version's
    // e -> (i,j,k) by three divisions was 80 us of VALU at config 3): walk (i, m = j*b2+k)
    // and split m with one exact float-reciprocal division by the tiny b2.
    const int b12 = b1 * b2;
    const float rb2 = __frcp_rn((float)b2);
    constexpr int LPU = (UNIT == 1) ? 1 : UNIT;          // lanes per tile unit
    const int o = (UNIT == 1) ? 0 : (tid & (LPU - 1));   // float inside the brick
    const int msub = tid / LPU, mstep = kBlock / LPU;
    for (int i = 0; i < b0; ++i) {
        for (int m = msub; m < b12; m += mstep) {
            int v = tile[(i * b12 + m) * UNIT + o];
            // a half-wave = one brick (bricked) -- skip the address arithmetic for all-zero bricks
            bool any = v != 0;
            if (UNIT != 1) any = (unsigned)(__ballot(v != 0) >> (tid & 32)) != 0u;
            if (any) {
                int j = __float2int_rz(((float)m + 0.5f) * rb2); // exact: m < 2^14, b2 <= 2^14
                int k = m - j * b2;
                unsigned g;
                if (LAYOUT == DIFFUS_CANONICAL)
                    g = ((unsigned)(l0 + i) * (unsigned)A.G.d1 + (unsigned)(l1 + j)) * (unsigned)A.G.d2 + (unsigned)(l2 + k);
                else
                    g = (((unsigned)(l0 + i) * (unsigned)A.G.nb1 + (unsigned)(l1 + j)) * (unsigned)A.G.nb2 + (unsigned)(l2 + k)) * kBrickFloats + (unsigned)o;
                // one plain, idempotent flag store per touched brick; no returning atomic (its latency
                // would sit on the flush path)
                if (LAYOUT == DIFFUS_BRICKED && A.gtouched && o == 0) A.gtouched[g >> 5] = 1;
                if (v != 0) atomicAdd(A.gvol + g, ldexpf((float)v, -fx));
            }
        }
    }
    STAMP(5);
#ifdef DIFFUS_STAMP
    if (threadIdx.x == 0 && g_stamps) {
        g_stamps[(size_t)blockIdx.x * 8 + 6] = (unsigned long long)nt;
        g_stamps[(size_t)blockIdx.x * 8 + 7] = 1;
    }
#endif
}

// ----------------------------------------------------------------------------
// start > 0: median over rays of r[:, start] (reference :243), one block per pose.
// Lower median like torch.median; NaN if any NaN.  Also zeroes gmed[p].
template <int SAMPLER, int LAYOUT>
__global__ __launch_bounds__(kBlock) void median_kernel(Args A)
{
    extern __shared__ float vals[];
    __shared__ int s_nan;
    const int pose = blockIdx.x;
    if (threadIdx.x == 0) s_nan = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < A.R; i += blockDim.x) {
        Pose ps;
        load_pose(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, (long)pose * A.R + i);
        float zz[2];
        for (int q = 0; q < 2; ++q) {
            int k = A.start + q;
            float p0 = ray_point(ps, 0, k), p1 = ray_point(ps, 1, k), p2 = ray_point(ps, 2, k);
            if (SAMPLER == DIFFUS_NEAREST) {
                int i0 = nearest_index(p0, A.G.d0), i1 = nearest_index(p1, A.G.d1), i2 = nearest_index(p2, A.G.d2);
                zz[q] = A.vol[vox_off<LAYOUT>(A.G, i0, i1, i2)];
            } else {
                zz[q] = tri_sample<LAYOUT, false>(A.vol, A.G, p0, p1, p2).v;
            }
        }
        float v = reflect(zz[0], zz[1]);
        vals[i] = v;
        if (v != v) atomicOr(&s_nan, 1);
    }
    __syncthreads();
    if (s_nan) {
        if (threadIdx.x == 0) {
            A.med[pose] = __builtin_nanf("");
            A.who[pose] = -1;
            A.gmed[pose] = 0.f;
        }
        return;
    }
    const int target = (A.R - 1) / 2;
    for (int i = threadIdx.x; i < A.R; i += blockDim.x) {
        float v = vals[i];
        int rank = 0;
        for (int j = 0; j < A.R; ++j) {
            float u = vals[j];
            rank += (u < v) || (u == v && j < i);
        }
        if (rank == target) { // exactly one i satisfies this
            A.med[pose] = v;
            A.who[pose] = i;
            A.gmed[pose] = 0.f;
        }
    }
}

// start > 0, backward: route gmed[p] to the ray that supplied the median
// (torch.median's backward).  One thread per pose; runs after render_bwd_kernel.
template <int SAMPLER, int LAYOUT>
__global__ void median_bwd_kernel(Args A)
{
    const int pose = blockIdx.x * blockDim.x + threadIdx.x;
    if (pose >= A.P) return;
    const int i = A.who[pose];
    const float gm = A.gmed[pose];
    if (i < 0 || gm == 0.f || !finitef(gm)) return;
    const long w = (long)pose * A.R + i;
    Pose ps;
    load_pose(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
    float zz[2], g0[2], g1[2], g2[2];
    for (int q = 0; q < 2; ++q) {
        int k = A.start + q;
        float p0 = ray_point(ps, 0, k), p1 = ray_point(ps, 1, k), p2 = ray_point(ps, 2, k);
        if (SAMPLER == DIFFUS_NEAREST) {
            int i0 = nearest_index(p0, A.G.d0), i1 = nearest_index(p1, A.G.d1), i2 = nearest_index(p2, A.G.d2);
            zz[q] = A.vol[vox_off<LAYOUT>(A.G, i0, i1, i2)];
            g0[q] = g1[q] = g2[q] = 0.f;
        } else {
            TriSample s = tri_sample<LAYOUT, true>(A.vol, A.G, p0, p1, p2);
            zz[q] = s.v; g0[q] = s.g0; g1[q] = s.g1; g2[q] = s.g2;
        }
    }
    float s = zz[0] + zz[1];
    float inv = __fdiv_rn(1.f, s);
    float zb[2] = {gm * (-2.f * zz[1] * inv * inv), gm * (2.f * zz[0] * inv * inv)};
    for (int q = 0; q < 2; ++q) {
        if (!finitef(zb[q]) || zb[q] == 0.f) continue;
        int k = A.start + q;
        if (A.gvol) {
            Cell c = cell_of<SAMPLER>(A, ps, k);
            for_each_corner<SAMPLER>(c, zb[q], [&](int a, int b, int cc, float v) {
                if (v != 0.f) {
                    unsigned g = vox_off<GradLayout<LAYOUT>::value>(A.G, a, b, cc);
                    atomicAdd(A.gvol + g, v);
                    if (GradLayout<LAYOUT>::value == DIFFUS_BRICKED && A.gtouched) A.gtouched[g >> 5] = 1;
                }
            });
        }
        if (SAMPLER == DIFFUS_TRILINEAR) {
            float kf = (float)k;
            if (A.gsrc_part) {
                A.gsrc_part[w * 3 + 0] += zb[q] * g0[q];
                A.gsrc_part[w * 3 + 1] += zb[q] * g1[q];
                A.gsrc_part[w * 3 + 2] += zb[q] * g2[q];
            }
            if (A.gdirs) {
                A.gdirs[w * 3 + 0] += kf * zb[q] * g0[q];
                A.gdirs[w * 3 + 1] += kf * zb[q] * g1[q];
                A.gdirs[w * 3 + 2] += kf * zb[q] * g2[q];
            }
        }
    }
}

// gsrc[p,:] = sum over rays of gsrc_part[p,:,:], fixed order => deterministic.
__global__ __launch_bounds__(kBlock) void reduce_gsrc_kernel(const float *__restrict__ part, float *__restrict__ gsrc,
                                                             int R)
{
    __shared__ float sm[3][kBlock];
    const int pose = blockIdx.x;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int i = threadIdx.x; i < R; i += kBlock) {
        const float *q = part + ((long)pose * R + i) * 3;
        a0 += q[0]; a1 += q[1]; a2 += q[2];
    }
    sm[0][threadIdx.x] = a0; sm[1][threadIdx.x] = a1; sm[2][threadIdx.x] = a2;
    __syncthreads();
    for (int s = kBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sm[0][threadIdx.x] += sm[0][threadIdx.x + s];
            sm[1][threadIdx.x] += sm[1][threadIdx.x + s];
            sm[2][threadIdx.x] += sm[2][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x < 3) gsrc[pose * 3 + threadIdx.x] = sm[threadIdx.x][0];
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
    const int n0 = lane * C;
    float r[C], e[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        int n = n0 + j;
        r[j] = (n >= 1 && n <= N) ? rin[w * N + n - 1] : 0.f;
    }
    echo_chunk<C>(r, lane, e);
#pragma unroll
    for (int j = 0; j < C; ++j) {
        int n = n0 + j;
        if (n <= N) echo[w * (N + 1) + n] = e[j];
    }
}

// ----------------------------------------------------------------------------
// canonical <-> bricked conversion.  A block moves 4 x 4 x 64 voxels (32 bricks,
// 4 KiB): 16 canonical rows of 256 B on one side, 4 KiB contiguous on the other,
// through an LDS transpose so that both sides are coalesced.
constexpr int kConvZ = 64;
template <bool TO_BRICKED, bool ACCUMULATE>
__global__ __launch_bounds__(kBlock) void brick_convert_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                               Geom G)
{
    __shared__ float t[16][kConvZ + 1];
    const int bz0 = blockIdx.x * (kConvZ / 2); // first brick along dim 2
    const int by = blockIdx.y, bx = blockIdx.z;
    const int tid = threadIdx.x;
    const long brick0 = ((long)bx * G.nb1 + by) * G.nb2 + bz0;
    if (TO_BRICKED) {
        for (int e = tid; e < 16 * kConvZ; e += kBlock) {
            int row = e / kConvZ, zz = e - row * kConvZ;
            int x = bx * 4 + (row >> 2), y = by * 4 + (row & 3), z = bz0 * 2 + zz;
            t[row][zz] = (x < G.d0 && y < G.d1 && z < G.d2) ? in[((long)x * G.d1 + y) * G.d2 + z] : 0.f;
        }
        __syncthreads();
        for (int e = tid; e < 16 * kConvZ; e += kBlock) {
            int brick = e >> 5, off = e & 31;
            if (bz0 + brick < G.nb2) out[(brick0 + brick) * kBrickFloats + off] = t[off >> 1][brick * 2 + (off & 1)];
        }
    } else {
        for (int e = tid; e < 16 * kConvZ; e += kBlock) {
            int brick = e >> 5, off = e & 31;
            if (bz0 + brick < G.nb2) t[off >> 1][brick * 2 + (off & 1)] = in[(brick0 + brick) * kBrickFloats + off];
        }
        __syncthreads();
        for (int e = tid; e < 16 * kConvZ; e += kBlock) {
            int row = e / kConvZ, zz = e - row * kConvZ;
            int x = bx * 4 + (row >> 2), y = by * 4 + (row & 3), z = bz0 * 2 + zz;
            if (x < G.d0 && y < G.d1 && z < G.d2) {
                long o = ((long)x * G.d1 + y) * G.d2 + z;
                if (ACCUMULATE)
                    out[o] += t[row][zz];
                else
                    out[o] = t[row][zz];
            }
        }
    }
}

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

// The same, privatised: a block takes a patch of 16 rows x 64 columns of the (rows, cols) sample
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
            int *w = &wz[(l1 + y) * W + (l0 + x)];
            if (__builtin_nontemporal_load(w) < v) atomicMax(w, v);
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

// Separable zero-padded Gaussian blur of `nch` planes of H x W (in -> out), one 32x32 tile per block.
__global__ __launch_bounds__(kBlock) void blur2d_kernel(const float *__restrict__ in, float *__restrict__ out, int H, int W,
                                                        int half, float sigma)
{
    __shared__ float kw[2 * kSplatMaxHalf + 1];
    __shared__ float src[kSplatTile + 2 * kSplatMaxHalf][kSplatTile + 2 * kSplatMaxHalf + 1];
    __shared__ float mid[kSplatTile + 2 * kSplatMaxHalf][kSplatTile + 1];
    const int size = 2 * half + 1, ext = kSplatTile + 2 * half;
    const long plane = (long)blockIdx.z * H * W;
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
        src[r][c] = (y >= 0 && y < H && x >= 0 && x < W) ? in[plane + (long)y * W + x] : 0.f;
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
            out[plane + (long)y * W + x] = a;
        }
    }
}

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
// Energy loss used by the benchmarks and examples: loss[p] = sum(frame[p]^2), gframe = 2 * frame,
// in one streaming pass.  kLossSplit blocks per pose write partial sums, a second tiny kernel adds
// them in a fixed order (deterministic; one block per pose alone used only P of the 256 CUs).
constexpr int kLossSplit = 16;
__global__ __launch_bounds__(kBlock) void loss_sumsq_kernel(const float *__restrict__ frame, float *__restrict__ part,
                                                            float *__restrict__ gframe, long n)
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
    if (threadIdx.x == 0) part[pz * kLossSplit + blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

__global__ void loss_finish_kernel(const float *__restrict__ part, float *__restrict__ loss, int P)
{
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    float t = 0.f;
    for (int i = 0; i < kLossSplit; ++i) t += part[p * kLossSplit + i];
    loss[p] = t;
}

// Sparse bricked gradient -> canonical: one wave per 64 bricks reads their "touched" flags; every
// touched brick is added into (or stored to) the canonical tensor, ZEROED in the bricked buffer and
// its flag cleared, so the bricked buffer and the flags are all-zero again afterwards.  A fan touches
// a few thousand of the 524 288 bricks of a 256^3 volume: this replaces a 64 MiB memset plus a
// 128 MiB dense conversion per step.
__global__ __launch_bounds__(kBlock) void gradbuf_flush_kernel(float *__restrict__ bricked, int *__restrict__ touched,
                                                               float *__restrict__ out, Geom G, long nbricks,
                                                               int accumulate)
{
    __shared__ int s_list[kWavesPerBlock][kWave];
    const int wib = threadIdx.x >> 6;
    const long w = (long)blockIdx.x * kWavesPerBlock + wib;
    const int lane = threadIdx.x & 63;
    const long b0 = w * kWave;
    if (b0 >= nbricks) return;
    const long mine = b0 + lane;
    int f = (mine < nbricks) ? touched[mine] : 0;
    unsigned long long m = __ballot(f != 0);
    if (m == 0) return; // wave-uniform: nothing touched in these 64 bricks
    if (f) {
        touched[mine] = 0;
        s_list[wib][__builtin_popcountll(m & ((1ull << lane) - 1))] = lane; // compact the touched ids
    }
    wave_lds_sync();
    const int cnt = __builtin_popcountll(m);
    const int o = lane & 31, half = lane >> 5;
    // two bricks per step (one per half-wave); iterations are independent so their loads overlap
#pragma unroll 4
    for (int i = half; i < cnt; i += 2) {
        const long brick = b0 + s_list[wib][i];
        float v = bricked[brick * kBrickFloats + o];
        bricked[brick * kBrickFloats + o] = 0.f;
        long bz = brick % G.nb2, t = brick / G.nb2;
        long by = t % G.nb1, bx = t / G.nb1;
        int x = (int)bx * 4 + (o >> 3), y = (int)by * 4 + ((o >> 1) & 3), z = (int)bz * 2 + (o & 1);
        if (x < G.d0 && y < G.d1 && z < G.d2) {
            long a = ((long)x * G.d1 + y) * G.d2 + z;
            out[a] = accumulate ? out[a] + v : v;
        }
    }
}

// canonical -> PAIRED: a block writes 4 x 4 columns x 32 depths (4 KiB contiguous) from 16 canonical
// rows of 33 floats, through LDS.
__global__ __launch_bounds__(kBlock) void pair_convert_kernel(const float *__restrict__ in, float *__restrict__ out, Geom G)
{
    __shared__ float t[16][34];
    const int z0 = blockIdx.x * 32;
    const int by = blockIdx.y, bx = blockIdx.z;
    const int tid = threadIdx.x;
    for (int e = tid; e < 16 * 33; e += kBlock) {
        int row = e / 33, zz = e - row * 33;
        int x = min(bx * 4 + (row >> 2), G.d0 - 1), y = min(by * 4 + (row & 3), G.d1 - 1), z = min(z0 + zz, G.d2 - 1);
        t[row][zz] = in[((long)x * G.d1 + y) * G.d2 + z];
    }
    __syncthreads();
    const long col0 = ((long)bx * G.nb1 + by) * G.d2 + z0;
    for (int e = tid; e < 32 * kBrickFloats; e += kBlock) {
        int zz = e >> 5, off = e & 31;
        if (z0 + zz < G.d2) out[(col0 + zz) * kBrickFloats + off] = t[off >> 1][zz + (off & 1)];
    }
}

// ----------------------------------------------------------------------------
// host side
size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

Geom make_geom(int d0, int d1, int d2)
{
    Geom G;
    G.d0 = d0; G.d1 = d1; G.d2 = d2;
    G.nb1 = (d1 + 3) / 4;
    G.nb2 = (d2 + 1) / 2;
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
    return (size_t)((d0 + 3) / 4) * G.nb1 * d2 * kBrickFloats;
}

struct Workspace {
    float *med;
    int *who;
    float *gmed;
    float *gsrc_part;
    float *zbar;
    size_t bytes;
};

Workspace carve(void *base, int P, int R, int N1)
{
    Workspace ws;
    char *p = (char *)base;
    size_t o = 0;
    ws.med = (float *)(p + o); o += align256(sizeof(float) * (size_t)P);
    ws.who = (int *)(p + o);   o += align256(sizeof(int) * (size_t)P);
    ws.gmed = (float *)(p + o); o += align256(sizeof(float) * (size_t)P);
    ws.gsrc_part = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R * 3);
    ws.zbar = (float *)(p + o); o += align256(sizeof(float) * (size_t)P * R * (N1 > 0 ? N1 : 0));
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
    if (need_scan && S - start > DIFFUS_MAX_SAMPLES) return DIFFUS_EUNSUPPORTED;
    if (start > 0 && (size_t)R * sizeof(float) > 64 * 1024) return DIFFUS_EUNSUPPORTED; // median LDS
    return DIFFUS_OK;
}

Args make_args(const float *vol, int d0, int d1, int d2, const void *src, int src_dtype, const void *dirs,
               int dirs_dtype, int P, int R, int S, int start, float alpha, const Workspace &ws)
{
    Args A{};
    A.vol = vol;
    A.G = make_geom(d0, d1, d2);
    A.src = src; A.dirs = dirs;
    A.src_f64 = src_dtype == DIFFUS_F64; A.dir_f64 = dirs_dtype == DIFFUS_F64;
    A.P = P; A.R = R; A.S = S; A.start = start; A.N1 = S - start;
    A.neg_alpha = -alpha;
    A.med = ws.med; A.who = ws.who; A.gmed = ws.gmed;
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

int launch_median(const Args &A, int sampler, int layout, hipStream_t st)
{
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        hipLaunchKernelGGL((median_kernel<decltype(S_)::value, decltype(L_)::value>), dim3(A.P), dim3(kBlock),
                           sizeof(float) * (size_t)A.R, st, A);
        return last_launch();
    });
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
    default: hipLaunchKernelGGL((render_fwd_kernel<16, SM, LY, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    }
    return last_launch();
}

int launch_fwd(const Args &A, int sampler, int layout, hipStream_t st)
{
    const bool f32 = !A.src_f64 && !A.dir_f64;
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        constexpr int SM = decltype(S_)::value, LY = decltype(L_)::value;
        return f32 ? launch_fwd_t<SM, LY, 0>(A, st) : launch_fwd_t<SM, LY, 1>(A, st);
    });
}

template <int SM, int LY, bool GPOSE, int PM>
int launch_bwd_p(const Args &A, hipStream_t st)
{
    const long waves = (long)A.P * A.R;
    const unsigned nblk = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    switch (chunk_for(A.N1)) {
    case 2: hipLaunchKernelGGL((render_bwd_kernel<2, SM, LY, GPOSE, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    case 4: hipLaunchKernelGGL((render_bwd_kernel<4, SM, LY, GPOSE, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    case 8: hipLaunchKernelGGL((render_bwd_kernel<8, SM, LY, GPOSE, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    default: // 16 samples per lane need ~300 registers: one wave per block so the whole 512-entry file is available
        hipLaunchKernelGGL((render_bwd_kernel<16, SM, LY, GPOSE, 1, PM>), dim3((unsigned)waves), dim3(kWave), 0, st, A);
        break;
    }
    return last_launch();
}

template <int SM, int LY, bool GPOSE>
int launch_bwd_t(const Args &A, hipStream_t st)
{
    return (!A.src_f64 && !A.dir_f64) ? launch_bwd_p<SM, LY, GPOSE, 0>(A, st) : launch_bwd_p<SM, LY, GPOSE, 1>(A, st);
}

int launch_bwd(const Args &A, int sampler, int layout, bool pose, hipStream_t st)
{
    return dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
        constexpr int SM = decltype(S_)::value, LY = decltype(L_)::value;
        if constexpr (SM == DIFFUS_NEAREST) {
            return launch_bwd_t<SM, LY, false>(A, st);
        } else {
            return pose ? launch_bwd_t<SM, LY, true>(A, st) : launch_bwd_t<SM, LY, false>(A, st);
        }
    });
}

} // namespace

// ----------------------------------------------------------------------------
extern "C" {

int diffus_abi_version(void) { return DIFFUS_ABI_VERSION; }

const char *diffus_strerror(int code)
{
    switch (code) {
    case DIFFUS_OK: return "ok";
    case DIFFUS_EINVAL: return "invalid argument";
    case DIFFUS_EUNSUPPORTED: return "unsupported shape (S - start > 1024, a volume edge > 2^24, or too many rays for start > 0)";
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

size_t diffus_bricked_floats(int d0, int d1, int d2)
{
    if (d0 <= 0 || d1 <= 0 || d2 <= 0) return 0;
    return bricked_floats(d0, d1, d2);
}

size_t diffus_brick_count(int d0, int d1, int d2)
{
    if (d0 <= 0 || d1 <= 0 || d2 <= 0) return 0;
    return bricked_floats(d0, d1, d2) / kBrickFloats;
}

int diffus_gradbuf_flush(float *bricked, int *touched, int d0, int d1, int d2, float *vol, int accumulate,
                         diffus_stream_t stream)
{
    if (!bricked || !touched || !vol || d0 <= 0 || d1 <= 0 || d2 <= 0) return DIFFUS_EINVAL;
    Geom G = make_geom(d0, d1, d2);
    const long nbricks = (long)(bricked_floats(d0, d1, d2) / kBrickFloats);
    const long waves = (nbricks + kWave - 1) / kWave;
    const unsigned nblk = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(gradbuf_flush_kernel, dim3(nblk), dim3(kBlock), 0, (hipStream_t)stream, bricked, touched, vol, G,
                       nbricks, accumulate);
    return last_launch();
}

size_t diffus_paired_floats(int d0, int d1, int d2)
{
    if (d0 <= 0 || d1 <= 0 || d2 <= 0) return 0;
    return paired_floats(d0, d1, d2);
}

int diffus_pair_volume(const float *vol, int d0, int d1, int d2, float *paired, diffus_stream_t stream)
{
    if (!vol || !paired || d0 <= 0 || d1 <= 0 || d2 <= 0) return DIFFUS_EINVAL;
    Geom G = make_geom(d0, d1, d2);
    dim3 grid((d2 + 31) / 32, G.nb1, (d0 + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535) return DIFFUS_EUNSUPPORTED;
    hipLaunchKernelGGL(pair_convert_kernel, grid, dim3(kBlock), 0, (hipStream_t)stream, vol, paired, G);
    return last_launch();
}

int diffus_brick_volume(const float *vol, int d0, int d1, int d2, float *bricked, diffus_stream_t stream)
{
    if (!vol || !bricked || d0 <= 0 || d1 <= 0 || d2 <= 0) return DIFFUS_EINVAL;
    Geom G = make_geom(d0, d1, d2);
    dim3 grid((G.nb2 + kConvZ / 2 - 1) / (kConvZ / 2), G.nb1, (d0 + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535) return DIFFUS_EUNSUPPORTED;
    hipLaunchKernelGGL((brick_convert_kernel<true, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, vol, bricked, G);
    return last_launch();
}

int diffus_unbrick_volume(const float *bricked, int d0, int d1, int d2, float *vol, int accumulate,
                          diffus_stream_t stream)
{
    if (!vol || !bricked || d0 <= 0 || d1 <= 0 || d2 <= 0) return DIFFUS_EINVAL;
    Geom G = make_geom(d0, d1, d2);
    dim3 grid((G.nb2 + kConvZ / 2 - 1) / (kConvZ / 2), G.nb1, (d0 + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535) return DIFFUS_EUNSUPPORTED;
    if (accumulate)
        hipLaunchKernelGGL((brick_convert_kernel<false, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, bricked, vol, G);
    else
        hipLaunchKernelGGL((brick_convert_kernel<false, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, bricked, vol, G);
    return last_launch();
}

int diffus_render_fwd(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype, int P, int R, int S, int start, float alpha, int sampler,
                      float *frame, int64_t *idx, void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    int rc = check_common(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, start, sampler, layout, true);
    if (rc) return rc;
    if (!frame) return DIFFUS_EINVAL;
    Workspace ws = carve(workspace, P, R, S - start);
    if (start > 0 && (!workspace || workspace_bytes < ws.bytes)) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    Args A = make_args(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, ws);
    A.frame = frame;
    A.idx = (long long *)idx;
    if (start > 0) {
        rc = launch_median(A, sampler, layout, st);
        if (rc) return rc;
    }
    return launch_fwd(A, sampler, layout, st);
}

int diffus_render_bwd(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype, int P, int R, int S, int start, float alpha, int sampler,
                      const float *gframe, float *gvol, int *gvol_touched, float *gsrc, float *gdirs, int stages,
                      void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    int rc = check_common(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, start, sampler, layout, true);
    if (rc) return rc;
    if (!gframe) return DIFFUS_EINVAL;
    if (stages < 1 || stages > DIFFUS_BWD_ALL) return DIFFUS_EINVAL;
    if (!gvol && !gsrc && !gdirs) return DIFFUS_OK;
    const bool do_scan = stages & DIFFUS_BWD_SCAN, do_scatter = stages & DIFFUS_BWD_SCATTER;
    Workspace ws = carve(workspace, P, R, S - start);
    if (!workspace || workspace_bytes < ws.bytes) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const bool pose = sampler == DIFFUS_TRILINEAR && (gsrc || gdirs);
    if (sampler == DIFFUS_NEAREST && do_scan) { // integer indices: no pose gradient (reference :754-758)
        if (gsrc && hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)P * 3, st) != hipSuccess) return DIFFUS_ELAUNCH;
        if (gdirs && hipMemsetAsync(gdirs, 0, sizeof(float) * (size_t)P * R * 3, st) != hipSuccess) return DIFFUS_ELAUNCH;
    }
    if (sampler == DIFFUS_NEAREST && !gvol) return DIFFUS_OK;
    Args A = make_args(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, ws);
    A.gframe = gframe;
    A.gvol = gvol;
    A.gtouched = (gvol && layout != DIFFUS_CANONICAL) ? gvol_touched : nullptr;
    A.zbar = gvol ? ws.zbar : nullptr;
    A.gsrc_part = (pose && gsrc) ? ws.gsrc_part : nullptr;
    A.gdirs = pose ? gdirs : nullptr;
    if (do_scan) {
        if (start > 0) { // recompute the median (and zero gmed)
            rc = launch_median(A, sampler, layout, st);
            if (rc) return rc;
        }
        rc = launch_bwd(A, sampler, layout, pose, st);
        if (rc) return rc;
    }
    if (gvol && do_scatter) {
        const int rgs = (R + kPatchRays - 1) / kPatchRays, sgs = (A.N1 + kPatchSteps - 1) / kPatchSteps;
        const unsigned nb = (unsigned)((long)P * rgs * sgs);
        const bool f32 = !A.src_f64 && !A.dir_f64;
        const int glayout = layout == DIFFUS_PAIRED ? DIFFUS_BRICKED : layout; // the scatter only sees the gradient
        rc = dispatch_sl(sampler, glayout, [&](auto S_, auto L_) {
            constexpr int SM = decltype(S_)::value, LY = (decltype(L_)::value == DIFFUS_PAIRED) ? DIFFUS_BRICKED : decltype(L_)::value;
            if (f32)
                hipLaunchKernelGGL((scatter_patch_kernel<SM, LY, 0>), dim3(nb), dim3(kBlock), 0, st, A, rgs, sgs);
            else
                hipLaunchKernelGGL((scatter_patch_kernel<SM, LY, 1>), dim3(nb), dim3(kBlock), 0, st, A, rgs, sgs);
            return last_launch();
        });
        if (rc) return rc;
    }
    if (start > 0 && do_scan) {
        const unsigned nb = (unsigned)((P + 63) / 64);
        rc = dispatch_sl(sampler, layout, [&](auto S_, auto L_) {
            hipLaunchKernelGGL((median_bwd_kernel<decltype(S_)::value, decltype(L_)::value>), dim3(nb), dim3(64), 0, st, A);
            return last_launch();
        });
        if (rc) return rc;
    }
    if (pose && gsrc && do_scan) {
        hipLaunchKernelGGL(reduce_gsrc_kernel, dim3(P), dim3(kBlock), 0, st, ws.gsrc_part, gsrc, R);
        if (hipGetLastError() != hipSuccess) return DIFFUS_ELAUNCH;
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
    Args A = make_args(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, 0, 0.f, ws);
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

// ---- scan conversion ----
static int splat_half(float sigma) { return ((int)(6.f * sigma) | 1) / 2; } // size = int(6 sigma) | 1, reference :725

size_t diffus_splat_workspace_bytes(int P, int H, int W)
{
    if (P <= 0 || H <= 0 || W <= 0) return 0;
    return align256(sizeof(int) * (size_t)P * H * W) + 2 * align256(sizeof(float) * (size_t)P * 2 * H * W);
}

int diffus_splat_fwd(const float *c0, const float *c1, const float *val, int P, long n, int cols, int H, int W,
                     float sigma, float *out, void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    if (!c0 || !c1 || !val || !out || P <= 0 || n <= 0 || H <= 0 || W <= 0 || !(sigma > 0.f)) return DIFFUS_EINVAL;
    if (cols < 0 || (cols > 0 && n % cols != 0)) return DIFFUS_EINVAL;
    if (n > 0x7fffffffL || (long)H * W > 0x3fffffffL) return DIFFUS_EUNSUPPORTED;
    const int half = splat_half(sigma);
    if (half > kSplatMaxHalf) return DIFFUS_EUNSUPPORTED;
    if (!workspace || workspace_bytes < diffus_splat_workspace_bytes(P, H, W)) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const long hw = (long)H * W;
    int *winner = (int *)workspace;
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
    unsigned pb = (unsigned)((hw + kBlock - 1) / kBlock); if (pb > 4096) pb = 4096;
    hipLaunchKernelGGL(splat_compose_kernel, dim3(pb, P), dim3(kBlock), 0, st, winner, val, n, hw, planes);
    dim3 tiles((W + kSplatTile - 1) / kSplatTile, (H + kSplatTile - 1) / kSplatTile, P * 2);
    hipLaunchKernelGGL(blur2d_kernel, tiles, dim3(kBlock), 0, st, planes, blurred, H, W, half, sigma);
    tiles.z = P;
    hipLaunchKernelGGL(splat_divide_kernel<false>, tiles, dim3(kBlock), 0, st, blurred, nullptr, out, H, W);
    return last_launch();
}

int diffus_splat_bwd(const float *c0, const float *c1, int P, long n, int H, int W, float sigma, const float *gout,
                     float *gval, void *workspace, size_t workspace_bytes, diffus_stream_t stream)
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
    // recompute the weight plane and its blur (nothing was saved by the forward)
    (void)winner;
    if (hipMemsetAsync(planes, 0, sizeof(float) * (size_t)P * 2 * hw, st) != hipSuccess) return DIFFUS_ELAUNCH;
    unsigned nb = (unsigned)((n + kBlock - 1) / kBlock); if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(splat_mark_kernel, dim3(nb, P), dim3(kBlock), 0, st, c0, c1, n, H, W, planes);
    dim3 tiles((W + kSplatTile - 1) / kSplatTile, (H + kSplatTile - 1) / kSplatTile, P * 2);
    hipLaunchKernelGGL(blur2d_kernel, tiles, dim3(kBlock), 0, st, planes, blurred, H, W, half, sigma);
    // q = gout^T / (bw + eps) into planes[:, 0]; blur it into blurred[:, 0]; gather per sample
    tiles.z = P;
    hipLaunchKernelGGL(splat_divide_kernel<true>, tiles, dim3(kBlock), 0, st, blurred, gout, planes, H, W);
    // planes is now (P,1,H,W) q; blur plane-wise into `blurred` viewed as (P,H,W)
    hipLaunchKernelGGL(blur2d_kernel, tiles, dim3(kBlock), 0, st, planes, blurred, H, W, half, sigma);
    hipLaunchKernelGGL(splat_gather_kernel, dim3(nb, P), dim3(kBlock), 0, st, c0, c1, n, H, W, blurred, gval);
    return last_launch();
}

int diffus_loss_sumsq(const float *frame, int P, long n, float *loss, float *gframe, void *workspace,
                      size_t workspace_bytes, diffus_stream_t stream)
{
    if (!frame || !loss || P <= 0 || n <= 0) return DIFFUS_EINVAL;
    if (!workspace || workspace_bytes < sizeof(float) * (size_t)P * kLossSplit) return DIFFUS_EWORKSPACE;
    float *part = (float *)workspace;
    hipLaunchKernelGGL(loss_sumsq_kernel, dim3(kLossSplit, P), dim3(kBlock), 0, (hipStream_t)stream, frame, part, gframe, n);
    hipLaunchKernelGGL(loss_finish_kernel, dim3((P + 63) / 64), dim3(64), 0, (hipStream_t)stream, part, loss, P);
    return last_launch();
}

#ifdef DIFFUS_STAMP
int diffus_debug_set_stamps(unsigned long long *p)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif

int diffus_echo_traces(const float *refl, int B, int N, float *echo, diffus_stream_t stream)
{
    if (!refl && N > 0) return DIFFUS_EINVAL;
    if (!echo || B <= 0 || N < 0) return DIFFUS_EINVAL;
    if (N + 1 > DIFFUS_MAX_SAMPLES) return DIFFUS_EUNSUPPORTED;
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

} // extern "C"
