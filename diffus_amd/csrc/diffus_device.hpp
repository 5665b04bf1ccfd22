// diffus_device.hpp -- gfx950 (MI355X / CDNA4) kernels + C ABI for the DiffUS
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
// This header holds every __device__ helper shared by the translation units (render_fwd.hip,
// render_bwd.hip, scatter.hip, splat.hip); everything is in an anonymous namespace, so each unit
// gets its own copy and nothing here has external linkage.
#pragma once
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
#ifdef DIFFUS_RENORM_LDEXP // rounds 1-3: v_frexp_exp_i32_f32 + four v_ldexp_f32, 4.25 issue cycles each (tools/valu_issue_bench.hip)
    // v_frexp_exp_i32_f32 returns 0 for +-0, inf and NaN: no branch needed, ldexp(x, 0) is a no-op
    int ex = __builtin_amdgcn_frexp_expf(mx);
    m = mat_scale(m, -ex);
    return ex;
#else
    // The scale 2^-ex as a float, straight from mx's exponent field (two full-rate integer instructions), and four
    // full-rate multiplies: exact like ldexp (a power of two; the products stay normal: mx 2^-ex is in [0.5, 1)).
    // ex = biased exponent - 126 is frexp's exponent for every normal mx.  mx = 0 (an all-zero matrix) scales zeros by
    // 2^126 and reports -126, a NaN matrix stays NaN -- both only ever meet zeros / NaNs downstream; a matrix that has
    // overflowed to infinity was garbage before and is garbage after.
    const unsigned e23 = __float_as_uint(mx) & 0x7f800000u;
    const float sc = __uint_as_float(0x7e800000u - e23);
    m.a *= sc; m.b *= sc; m.c *= sc; m.d *= sc;
    return (int)(e23 >> 23) - 126;
#endif
}

// a * b with DX9 rules: 0 * anything (NaN and infinity included) = 0, an IEEE product otherwise.  One full-rate
// instruction where `a != 0 ? a * b : 0` is a compare and a select (4.25 issue cycles each, tools/valu_issue_bench.hip).
__device__ __forceinline__ float mul_legacy(float a, float b)
{
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// a * b + c on the low 24 bits of a and b (v_mad_u32_u24): ONE half-rate instruction where hipcc turns
// __umul24(a, b) + c + d into a multiply and a three-operand add (two).  b: a wave-uniform value (SGPR) / the constant 40.
__device__ __forceinline__ unsigned mad_u24_s(unsigned a, unsigned b_uniform, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
__device__ __forceinline__ unsigned mad_u24_40(unsigned a, unsigned c)
{
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, 40, %2" : "=v"(r) : "v"(a), "v"(c));
    return r;
}

__device__ __forceinline__ bool finitef(float x) { return fabsf(x) < __builtin_inff(); }
__device__ __forceinline__ bool mat_finite(const Mat &m) { return finitef(m.a) && finitef(m.b) && finitef(m.c) && finitef(m.d); }

// Wave-wide min/max of an int on the DPP path: Hillis-Steele offsets 1, 2, 4, 8 inside the rows of 16, then
// row_bcast:15 / row_bcast:31.  `old` = the identity of the operation, so hipcc folds each move into its
// v_min_i32_dpp / v_max_i32_dpp: 6 instructions (with old = v it was mov + mov_dpp + min per step, 21 in all).
// Result valid in every lane (readlane 63).
template <bool IS_MIN>
__device__ __forceinline__ int wave_reduce_minmax(int v)
{
    constexpr int ident = IS_MIN ? 0x7fffffff : (int)0x80000000;
#define DIFFUS_DPP_STEP(ctrl, rmask)                                                          \
    {                                                                                         \
        int o = __builtin_amdgcn_update_dpp(ident, v, ctrl, rmask, 0xf, false);               \
        v = IS_MIN ? min(v, o) : max(v, o);                                                   \
    }
    DIFFUS_DPP_STEP(0x111, 0xf) // row_shr:1
    DIFFUS_DPP_STEP(0x112, 0xf) // row_shr:2
    DIFFUS_DPP_STEP(0x114, 0xf) // row_shr:4
    DIFFUS_DPP_STEP(0x118, 0xf) // row_shr:8
    DIFFUS_DPP_STEP(0x142, 0xa) // row_bcast:15
    DIFFUS_DPP_STEP(0x143, 0xc) // row_bcast:31
#undef DIFFUS_DPP_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- cross-lane moves on the DPP path ---------------------------------------------------------
// The scans below used ds_bpermute (`__shfl_up/down`): 127 LDS-crossbar round trips per wave in the backward,
// each an address VGPR, an LDS issue slot shared by the CU's four SIMDs and ~100 cycles of latency behind an
// s_waitcnt.  A DPP move is one full-rate VALU instruction with no memory counter at all.  gfx950 is a GFX9
// target, so it has the whole-wave forms too: wave_shr:1 / wave_shl:1 (neighbour lane across row boundaries)
// and row_bcast:15 / row_bcast:31 (last lane of a row / of the lower half to the rows above).
// A lane whose source lane does not exist (or whose row is masked out) keeps `old`.
// The move must execute with every lane active: a source lane that EXEC disables counts as non-existent.
constexpr int kDppRowShl = 0x100, kDppRowShr = 0x110;              // + n, n = 1..15: lane i <- lane i+n / i-n of its row of 16
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;          // lane i <- lane i+1 / i-1 of the wave
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143;            // lane 15 of each row -> next row; lane 31 -> rows 2,3

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_mov(int old, int v)
{
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ Mat mat_dpp(const Mat &old, const Mat &m)
{
    return Mat{dpp_mov<CTRL, ROW_MASK>(old.a, m.a), dpp_mov<CTRL, ROW_MASK>(old.b, m.b), dpp_mov<CTRL, ROW_MASK>(old.c, m.c),
               dpp_mov<CTRL, ROW_MASK>(old.d, m.d)};
}
// The same moves when a lane without a source does not care what it gets (the scans only use the value under
// `if (has source)`): zero for a missing source lane, the register's previous content where the row is masked.
// ONE v_mov_b32_dpp -- dpp_mov has to load `old` into the destination first, a second VALU instruction per move.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_get(int v)
{
    return __builtin_amdgcn_mov_dpp(v, CTRL, ROW_MASK, 0xf, true);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_get(float v)
{
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ Mat mat_dpp_get(const Mat &m)
{
    return Mat{dpp_get<CTRL, ROW_MASK>(m.a), dpp_get<CTRL, ROW_MASK>(m.b), dpp_get<CTRL, ROW_MASK>(m.c), dpp_get<CTRL, ROW_MASK>(m.d)};
}
// The scans' fetch with the operation's IDENTITY where a lane has no source (lower rows of a row_shr round, rows a
// row_mask leaves out): the round then multiplies in every lane and needs no select afterwards -- a v_cndmask costs 4.25
// issue cycles like the DPP move itself, a v_mov of the constant 2.25 (tools/valu_issue_bench.hip), and with a full row
// mask the zeros of the identity come from bound_ctrl for nothing.  1 * a + 0 * c is a exactly (c finite).
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ Mat mat_dpp_ident(const Mat &m)
{
    if constexpr (ROW_MASK == 0xf)
        return Mat{dpp_mov<CTRL>(1.f, m.a), dpp_get<CTRL>(m.b), dpp_get<CTRL>(m.c), dpp_mov<CTRL>(1.f, m.d)};
    else
        return Mat{dpp_mov<CTRL, ROW_MASK>(1.f, m.a), dpp_mov<CTRL, ROW_MASK>(0.f, m.b), dpp_mov<CTRL, ROW_MASK>(0.f, m.c),
                   dpp_mov<CTRL, ROW_MASK>(1.f, m.d)};
}
// ... and for sums: 0 where there is no source (folds into one v_add_u32_dpp)
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_get0(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, true);
}
__device__ __forceinline__ float lane_bcast(float v, int srclane) // wave-uniform value of one lane (v_readlane_b32)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), srclane));
}
__device__ __forceinline__ Mat mat_lane_bcast(const Mat &m, int srclane)
{
    return Mat{lane_bcast(m.a, srclane), lane_bcast(m.b, srclane), lane_bcast(m.c, srclane), lane_bcast(m.d, srclane)};
}
// value of the previous / next lane of the wave; lane 0 / lane 63 gets `edge`
__device__ __forceinline__ float lane_prev(float v, float edge) { return dpp_mov<kDppWaveShr1>(edge, v); }
__device__ __forceinline__ float lane_next(float v, float edge) { return dpp_mov<kDppWaveShl1>(edge, v); }
__device__ __forceinline__ int lane_prev(int v, int edge) { return dpp_mov<kDppWaveShr1>(edge, v); }
__device__ __forceinline__ int lane_next(int v, int edge) { return dpp_mov<kDppWaveShl1>(edge, v); }
__device__ __forceinline__ Mat mat_lane_prev(const Mat &m, const Mat &edge) { return mat_dpp<kDppWaveShr1>(edge, m); }
__device__ __forceinline__ Mat mat_lane_next(const Mat &m, const Mat &edge) { return mat_dpp<kDppWaveShl1>(edge, m); }
// ... with 0 at the edge: one instruction each
__device__ __forceinline__ float lane_prev0(float v) { return dpp_get<kDppWaveShr1>(v); }
__device__ __forceinline__ float lane_next0(float v) { return dpp_get<kDppWaveShl1>(v); }
__device__ __forceinline__ int lane_prev0(int v) { return dpp_get<kDppWaveShr1>(v); }
__device__ __forceinline__ int lane_next0(int v) { return dpp_get<kDppWaveShl1>(v); }
__device__ __forceinline__ Mat mat_lane_next0(const Mat &m) { return mat_dpp_get<kDppWaveShl1>(m); }

// Inclusive scan over the 64 lanes, LOWER lanes on the left: six Hillis-Steele rounds, offsets 1, 2, 4, 8 inside
// the rows of 16 (row_shr), then the last lane of a row / of the lower half broadcast to the rows above it.
// f(has_source, round) is called once per round with every lane active; `fetch` inside it does the DPP moves.
// Round r of this schedule combines exactly the lanes a shuffle scan with offset 2^r would.
// The fourth argument says whether the round rescales its product: every SECOND round is enough -- two products of
// matrices normalised to a largest entry below 1 stay below 8 -- and a rescale is 8 instructions (3 max, frexp, 4 ldexp)
// plus the exponent bookkeeping.  Power-of-two scalings are exact, so the results do not change by a bit.
#define DIFFUS_SCAN_UP_ROUNDS(LANE, ROUND)                                   \
    ROUND(kDppRowShr + 1, 0xf, ((LANE) & 15) >= 1, false)                    \
    ROUND(kDppRowShr + 2, 0xf, ((LANE) & 15) >= 2, true)                     \
    ROUND(kDppRowShr + 4, 0xf, ((LANE) & 15) >= 4, false)                    \
    ROUND(kDppRowShr + 8, 0xf, ((LANE) & 15) >= 8, true)                     \
    ROUND(kDppBcast15, 0xa, ((LANE) & 16) != 0, false)                       \
    ROUND(kDppBcast31, 0xc, (LANE) >= 32, true)

// Sum over the wave on the DPP path; the total is valid in lane 63 only.
__device__ __forceinline__ float wave_sum_to_lane63(float v)
{
#define DIFFUS_SUM_STEP(ctrl, rmask) \
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, rmask, 0xf, true));
    DIFFUS_SUM_STEP(kDppRowShr + 1, 0xf)
    DIFFUS_SUM_STEP(kDppRowShr + 2, 0xf)
    DIFFUS_SUM_STEP(kDppRowShr + 4, 0xf)
    DIFFUS_SUM_STEP(kDppRowShr + 8, 0xf)
    DIFFUS_SUM_STEP(kDppBcast15, 0xa)
    DIFFUS_SUM_STEP(kDppBcast31, 0xc)
#undef DIFFUS_SUM_STEP
    return v;
}

// ... of a double (the two halves move separately); total valid in lane 63 only
__device__ __forceinline__ double wave_sum_to_lane63(double v)
{
#define DIFFUS_SUM_STEP(ctrl, rmask)                                                                     \
    {                                                                                                    \
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, rmask, 0xf, true);        \
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, rmask, 0xf, true);        \
        v += __hiloint2double(hi, lo);                                                                   \
    }
    DIFFUS_SUM_STEP(kDppRowShr + 1, 0xf)
    DIFFUS_SUM_STEP(kDppRowShr + 2, 0xf)
    DIFFUS_SUM_STEP(kDppRowShr + 4, 0xf)
    DIFFUS_SUM_STEP(kDppRowShr + 8, 0xf)
    DIFFUS_SUM_STEP(kDppBcast15, 0xa)
    DIFFUS_SUM_STEP(kDppBcast31, 0xc)
#undef DIFFUS_SUM_STEP
    return v;
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
__device__ __forceinline__ float ray_point_f(const Pose &ps, int c, float stepf) // stepf = float(k), exact below 2^24
{
    if (PM == 0 || ps.pmode == 0) {
        return __fadd_rn(ps.sf[c], __fmul_rn(stepf, ps.df[c]));
    } else if (ps.pmode == 1) {
        float t = __fmul_rn(stepf, ps.df[c]);
        return (float)__dadd_rn(ps.sd[c], (double)t);
    } else {
        return (float)__dadd_rn(ps.sd[c], __dmul_rn((double)stepf, ps.dd[c]));
    }
}
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

//   PAIRED:    (volume only) one record per (4 x 4 column block, depth z): 4 rows of dim 0 x FIVE columns of dim 1
//              x the float2 (v[z], v[min(z+1, d2-1)]) = 40 floats = 160 bytes.  The fifth column repeats the first
//              column of the next block (clamped at the volume's edge), so the two dim-1 neighbours of ANY sample are
//              adjacent: one trilinear sample = TWO 16-byte loads (row x0, row x1), each
//              (v[y0][z0], v[y0][z1], v[y1][z0], v[y1][z1]) with y1 = min(y0+1, d1-1), z1 = min(z0+1, d2-1) --
//              against 8 dword loads canonically and 4 eight-byte loads without the extra column.  The gather
//              runs at the texture addresser's rate, 16 cycles per wave-load whatever it fetches
//              (tools/lds_stage_probe.hip: 13.6 -> 9.9 us for the 4.19 M samples of config 3).  2.5x the
//              memory of the volume; the gradient of a PAIRED volume is BRICKED.
constexpr int kPairFloats = 40;
} // namespace
namespace diffus { // types that cross translation units need linkage
struct Geom {
    int d0, d1, d2;
    int nb1, nb2; // bricks along dim 1 / dim 2
    // BYTE strides of the volume layout in use, for the separable offsets of the gather (part_x/part_y/part_z):
    // canonical: one step along dim 0 / dim 1; bricked and paired: one brick row (4 voxels of dim 0) / brick column
    unsigned sxB, syB;
};
} // namespace diffus
using diffus::Geom;
namespace {

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
        return col * kPairFloats + (unsigned)((x & 3) * 10 + ((y & 3) << 1));
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
    // clamp(p, 0, hi) with NaN -> 0 in ONE instruction: v_med3_f32 returns min3 of its operands when one of them is a NaN,
    // and min3 skips the NaN (0 = min(0, hi)); it was compare + select + min
    const float pc = __builtin_amdgcn_fmed3f(p, 0.f, hi);
    float f = floorf(pc);
    a.i0 = (int)f;
    a.t = pc - f;
    a.i1 = min(a.i0 + 1, dim - 1);
    return a;
}

// The same cell for the fused gathers, cheaper by what the measured issue costs say (tools/valu_issue_bench.hip: compares,
// selects, v_floor, conversions and integer min/max cost 4.25 cycles, FP32 add/sub 2.25):
//   * t = fract(pc) (one instruction; pc >= 0, so pc - floor(pc) is exact and below 1: the same bits), i0 = trunc(pc);
//   * the border rule "no gradient where p <= 0 or p >= dim - 1" needs a test at the LOW side only: at the high side the
//     upper neighbour is the clamped cell itself (i1 = i0, or the repeated entry of a PAIRED record), so the difference
//     is an exact 0 by itself (NaN where the voxel is not finite -- as 0 * NaN was).  `lo` = p > 0 (false for NaN);
// (Addressing the cell itself as its own upper neighbour at the low side would make that difference an exact 0 too, but
// then a non-finite voxel next to the border no longer reaches the value as 0 * NaN = NaN, as it does in grid_sample.)
struct AxisG {
    int i0, i1;
    float t;
    bool lo;
};
__device__ __forceinline__ AxisG tri_axis_g(float p, int dim)
{
    AxisG a;
    const float hi = (float)(dim - 1);
    const float pc = __builtin_amdgcn_fmed3f(p, 0.f, hi);
    a.lo = p > 0.f;
    a.i0 = (int)pc;
    a.t = __builtin_amdgcn_fractf(pc);
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
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float reflect_fast(float z1, float z2) { return fast_div(z2 - z1, z1 + z2); }

// ----------------------------------------------------------------------------
} // namespace
namespace diffus {
struct Args {
    const float *vol;
    Geom G;
    const void *src;
    const void *dirs;
    int src_f64, dir_f64;
    int P, R, S, start, N1;
    // Rays longer than 64*16 samples are processed in SEGMENTS of at most 1024 cropped samples, one
    // launch per segment, chained through per-ray carries in the workspace: [seg0, seg0+segN) is the
    // segment of this launch (seg0 = 0, segN = N1 when the ray fits one launch).
    int seg0, segN;
    const float *cin;  // nullable (P*R,5): running product P (normalised) + impedance of sample seg0-1
    float *cout;       // nullable (P*R,5): the same after this segment's last sample
    const float *cnext; // backward, nullable (P*R,5): carry-in of the NEXT segment (= cout of the carry-only pass)
    const float *uin;  // backward, nullable (P*R,4): adjoint carry U' entering from the next segment
    float *uout;       // backward, nullable (P*R,4): adjoint carry leaving towards the previous segment
    const float *zcin; // backward, nullable (P*R): d L/d imp contribution to this segment's LAST sample
    float *zcout;      // backward, nullable (P*R): contribution of this segment's first r to sample seg0-1
    float *gsrc_out;   // pose_finish_block, nullable (P,3): the per-pose sum of gsrc_part over rays goes here
    int finish_in_scatter; // the scatter launch carries P extra blocks that run pose_finish_block
    int vol_layout;        // layout of `vol` as a run-time value (pose_finish_block's float64 repair samples through it)
    int *rflag;            // nullable (P,R,2): one-pass step, set by the scan for ray halves with |echo| > kEchoRecheck
    int fans_planar;       // the caller vouches that no ray moves along dim 2 (DIFFUS_FANS_PLANAR): the scatter launch without the slab path
    int accum_pose;    // backward: add to (instead of overwrite) the per-ray pose-gradient partials
    float neg_alpha;     // -alpha
    float neg_alpha_l2e; // -alpha * log2(e): attenuation = exp2(neg_alpha_l2e * n), one multiply in front of v_exp_f32
    float att_step[16]; // exp2(neg_alpha_l2e * j), j = 0..15 (host: make_args) -- chunk_attenuation() under -DDIFFUS_ATT_STEPS
    // forward
    float *frame;
    long long *idx;
    // backward
    const float *gframe;
    // fused loss (diffus_render_bwd_mse): L_p = loss_scale * sum((frame_p - target_p)^2); the backward then takes
    // dL/dframe = 2 loss_scale (frame - target) on the fly from `gframe` (= the forward's frame) and `target`
    const float *target;  // nullable: zeros
    float loss_scale;
    int mse;              // 0: gframe is dL/dframe; 1: gframe is the frame
    float *loss_part;     // (P,R,2) per-ray sums of squares (two slots: the two waves of a SPLIT ray)
    float *loss_out;      // nullable (P): pose_finish_block sums loss_part into it
    float *gvol;      // layout that goes with vol's (GradLayout)
    int *gtouched;    // nullable: one flag per gradient brick, set when a launch adds into it (bricked only)
    float *zbar;      // (P,R,N1) d L / d imp per sample, consumed by scatter_patch_kernel
    float *gsrc_part; // (P,R,3) per-ray partials of d/d source
    float *gdirs;
    // start>0 coupling
    float *med;  // (P) median of r[:,start] over rays
    int *who;    // (P) ray that supplied it
    float *gmed; // (P,R) per-ray d/d (first kept coefficient) = each ray's share of d/d median, written by the adjoint scan
                 // (one plain store per ray: 256 float atomics on one address per pose took 50 us), summed by pose_finish_block
    float *medinfo; // (P,8) of the ray that supplied the median: its two impedance samples (steps start, start+1) and,
                    // trilinear, their spatial gradients -- all the backward needs to route d/d median (no re-sampling)
};
} // namespace diffus
using diffus::Args;
namespace {

// Blocks are dealt round-robin over the 8 XCDs (block b -> XCD b%8).  Remap so
// that consecutive LOGICAL blocks (= consecutive rays of one pose) sit on one
// XCD and share its L2.  Bijective for any grid size (guide T1).
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk)
{
    unsigned xcd = b & 7u, q = nblk >> 3, rem = nblk & 7u;
    unsigned base = (xcd < rem) ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q;
    return base + (b >> 3);
}

// Sums of N per-thread values over the block, in a fixed order (deterministic), valid in EVERY thread afterwards: DPP sum
// inside each wave, one LDS word per wave and value, one barrier, then every thread adds the waves' words in wave order.
// (Round 1-2: a log2(256)-step LDS tree with a barrier per step, once per quantity -- 20 barriers and ~8 us for the
// per-pose epilogue, the whole latency of a one-pose backward's last launch.)  sm: N * (blockDim.x / 64) floats.
template <int N>
__device__ __forceinline__ void block_sums(float (&a)[N], float *sm)
{
    const int nw = (int)(blockDim.x >> 6), wib = (int)(threadIdx.x >> 6);
#pragma unroll
    for (int n = 0; n < N; ++n) {
        const float t = wave_sum_to_lane63(a[n]);
        if ((threadIdx.x & 63) == 63) sm[n * nw + wib] = t;
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < N; ++n) {
        float t = 0.f;
        for (int w = 0; w < nw; ++w) t += sm[n * nw + w];
        a[n] = t;
    }
}

// exp(-alpha * n) for the C consecutive samples n = first .. first + C - 1 of a lane (reference src/renderer.py:256-259).
// Default: one v_exp_f32 per sample.  -DDIFFUS_ATT_STEPS: exp2(a * first) * exp2(a * j) with the second factor from the
// kernel arguments -- 85 issue cycles fewer per wave of the adjoint scan and within 2 ulp of the direct form, but another
// rounding: the volume gradient of a short nearest-sampled ray (hundreds of signed terms cancelling on one border voxel)
// moved from 3.45e-4 to 3.66e-4 of its float64 value (tools/diag_grad_noise.py), which put one parity case
// (test_backward_vs_float64_autograd[150-0-nearest]) at 1.10e-3 against its 1e-3.  Not taken.
template <int C>
__device__ __forceinline__ void chunk_attenuation(const Args &A, int first, float (&att)[C])
{
    static_assert(C <= 16, "att_step holds 16 factors");
#ifndef DIFFUS_ATT_STEPS
#pragma unroll
    for (int j = 0; j < C; ++j) att[j] = fast_exp2(A.neg_alpha_l2e * (float)(first + j));
#else
    const float e0 = fast_exp2(A.neg_alpha_l2e * (float)first);
#pragma unroll
    for (int j = 0; j < C; ++j) att[j] = (j == 0) ? e0 : e0 * A.att_step[j];
#endif
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

// Bank swizzle of that buffer.  A lane's chunk is C floats = F = C / 4 sixteen-byte words at a stride of 4 C bytes: the lanes
// l and l + 64 / C of a 16-lane pass of ds_read_b128 / ds_write_b128 sit on the same banks -- F-way conflicts (PMC, round 4:
// 62 % of the adjoint scan's LDS cycles).  So lane l keeps its word k at slot k ^ s(l), s(l) = (l / (64 / C)) mod F: the F
// aliasing lanes of a pass then use F different bank groups.  Seen from the INTERLEAVED side (sample n = j 64 + lane, owner
// lane n / C) the owner's s is (n / 64) mod F = j mod F -- a compile-time constant per access: the float at n lives at
// n ^ ((j mod F) << 2), a permutation inside an aligned group of 8 words (conflict-free like the natural order).
template <int C>
__device__ __forceinline__ int chunk_word(int lane, int k) // index (in floats) of word k of lane's chunk
{
    constexpr int F = C / 4;
    if constexpr (F <= 1) return lane * C + 4 * k;
    return lane * C + 4 * (k ^ ((lane / (kWave / C)) & (F - 1)));
}
template <int C>
__device__ __forceinline__ int inter_index(int lane, int j) // index (in floats) of sample j * 64 + lane
{
    constexpr int F = C / 4;
    if constexpr (F <= 1) return j * kWave + lane;
    return (j * kWave + lane) ^ ((j & (F - 1)) << 2);
}

template <int C>
__device__ __forceinline__ void to_chunked(float *wb, int lane, const float (&in)[C], float (&out)[C])
{
#pragma unroll
    for (int j = 0; j < C; ++j) wb[inter_index<C>(lane, j)] = in[j];
    wave_lds_sync();
    if (C >= 4) {
#pragma unroll
        for (int j = 0; j < C / 4; ++j) {
            float4 v = *reinterpret_cast<const float4 *>(wb + chunk_word<C>(lane, j));
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
#pragma unroll
        for (int j = 0; j < C / 4; ++j)
            *reinterpret_cast<float4 *>(wb + chunk_word<C>(lane, j)) = make_float4(in[4 * j], in[4 * j + 1], in[4 * j + 2], in[4 * j + 3]);
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) wb[lane * C + j] = in[j];
    }
    wave_lds_sync();
#pragma unroll
    for (int j = 0; j < C; ++j) out[j] = wb[inter_index<C>(lane, j)];
    wave_lds_sync();
}

// ---- a lane's C consecutive samples of a row, straight from / to memory (CHUNKED mapping, no LDS transpose) ------
// The texture addresser takes 16 cycles per wave-instruction whatever it moves: 8 dword accesses in the INTERLEAVED
// mapping (256-B runs) cost four times the 2 sixteen-byte ones of the CHUNKED mapping (a lane's 32 contiguous bytes, a
// wave's 2 KiB), and the transpose through LDS goes away with them.  Rows start at any float (N1 need not be a
// multiple of 4), so the accesses are declared dword-aligned; lanes whose chunk straddles the end of the row go
// element by element.
struct __attribute__((aligned(4))) F4a4 {
    float x, y, z, w;
};
struct __attribute__((aligned(4))) F2a4 {
    float x, y;
};
// STREAM: the row is written once and read by a LATER kernel (frame rows, zbar): a non-temporal store goes through the L2
// instead of leaving 33 MB of dirty lines for the end-of-kernel release to write back -- one-pass scan at config 3
// 29.3 -> 28.0 us (tools/time_step.py).
typedef float V4a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float V2a4 __attribute__((ext_vector_type(2), aligned(4)));
template <int C, bool STREAM = false>
__device__ __forceinline__ void store_chunk(float *__restrict__ row, int n0, int segN, const float (&v)[C])
{
    if (n0 + C <= segN) {
        if constexpr (C >= 4) {
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
                if constexpr (STREAM)
                    __builtin_nontemporal_store(V4a4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]}, reinterpret_cast<V4a4 *>(row + n0 + 4 * q));
                else
                    *reinterpret_cast<F4a4 *>(row + n0 + 4 * q) = F4a4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            }
        } else {
            if constexpr (STREAM)
                __builtin_nontemporal_store(V2a4{v[0], v[1]}, reinterpret_cast<V2a4 *>(row + n0));
            else
                *reinterpret_cast<F2a4 *>(row + n0) = F2a4{v[0], v[1]};
        }
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j)
            if (n0 + j < segN) {
                if constexpr (STREAM)
                    __builtin_nontemporal_store(v[j], row + n0 + j);
                else
                    row[n0 + j] = v[j];
            }
    }
}
template <int C>
__device__ __forceinline__ void load_chunk(const float *__restrict__ row, int n0, int segN, float (&v)[C])
{
    if (n0 + C <= segN) {
        if constexpr (C >= 4) {
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
                const F4a4 t = *reinterpret_cast<const F4a4 *>(row + n0 + 4 * q);
                v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
            }
        } else {
            const F2a4 t = *reinterpret_cast<const F2a4 *>(row + n0);
            v[0] = t.x; v[1] = t.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) v[j] = (n0 + j < segN) ? row[n0 + j] : 0.f;
    }
}

// Byte offset of voxel (x,y,z) as three separable parts: off = part_x(x) + part_y(y) + part_z(z).  A trilinear
// sample needs part_x and part_y for two coordinates each and sums them with v_add3_u32 -- 4 x (shift, 24-bit
// multiply, and, shift-add) instead of the 4 x (two multiplies, one of them the quarter-rate v_mul_lo_u32, and ~8
// bit operations) the nested form (x/4 * nb1 + y/4) * stride costs; the offsets come out in bytes, so no shift
// in front of the load either.  Bricked / paired strides are < 2^24 bytes (check_common), coordinates/4 < 2^22.
template <int LAYOUT>
__device__ __forceinline__ unsigned part_x(const Geom &G, int x)
{
    if (LAYOUT == DIFFUS_CANONICAL) return (unsigned)x * G.sxB;
    if (LAYOUT == DIFFUS_PAIRED) return __umul24((unsigned)x >> 2, G.sxB) + __umul24((unsigned)x & 3u, 40u); // 5 columns x 8 B per row
    return __umul24((unsigned)x >> 2, G.sxB) + (((unsigned)x & 3u) << 5);
}
template <int LAYOUT>
__device__ __forceinline__ unsigned part_y(const Geom &G, int y)
{
    if (LAYOUT == DIFFUS_CANONICAL) return (unsigned)y * G.syB;
    return __umul24((unsigned)y >> 2, G.syB) + (((unsigned)y & 3u) << 3);
}
template <int LAYOUT>
__device__ __forceinline__ unsigned part_z(int z)
{
    if (LAYOUT == DIFFUS_CANONICAL) return (unsigned)z << 2;
    if (LAYOUT == DIFFUS_PAIRED) return __umul24((unsigned)z, (unsigned)kPairFloats * 4u); // one 160-byte record per depth
    return (((unsigned)z >> 1) << 7) | (((unsigned)z & 1u) << 2);
}

// lerps of one trilinear sample from its 8 corner values (order 000,001,010,011,100,101,110,111 = dim0,dim1,dim2 bits).
// Each lerp is ONE fused multiply-add, a + t (b - a) rounded once, and the lerps run dim 0 first, then dim 1, then dim 2
// -- the oracle (oracle/diffus_oracle.c orc_sample_trilinear) does dim 2, dim 1, dim 0 with a separate multiply and add.
// Why: the corners arrive as two 16-byte rows (x0 and x1: y0z0, y0z1, y1z0, y1z1 -- a PAIRED record, or two canonical
// pairs), i.e. in register PAIRS (z0, z1).  Dim 0 first keeps every operand in those pairs: the value is 3 v_pk_add +
// 3 v_pk_fma + 2 scalar instructions, all three gradient components 9 more -- 17 against the 25 of the dim-2-first
// order (which hipcc's SLP pass packed too, but behind ~20 register moves per sample: the fused kernels are
// VALU-issue-bound).  Same trilinear polynomial, other rounding: <= 2e-7 relative from the oracle's sequence, inside the
// 1e-5 frame tolerance; the stage-wise kernels (tri_sample) keep the oracle's exact sequence.
typedef float V2f __attribute__((ext_vector_type(2)));
// keep0/1/2: false where the border rule must zero that gradient component and the difference is not an exact 0 by
// itself (tri_axis_g: only the low side of an axis whose upper neighbour is not addressed separately)
template <bool GRAD>
__device__ __forceinline__ TriSample tri_lerp(const float (&v)[8], float at, float bt, float ct, bool keep0, bool keep1, bool keep2)
{
    const V2f r0a = {v[0], v[1]}, r0b = {v[2], v[3]}, r1a = {v[4], v[5]}, r1b = {v[6], v[7]}; // (x, y): its (z0, z1)
    const V2f ta = {at, at}, tb = {bt, bt};
    const V2f da = r1a - r0a, db = r1b - r0b;                                   // d/d dim0 at y0, y1
    const V2f xa = __builtin_elementwise_fma(ta, da, r0a), xb = __builtin_elementwise_fma(ta, db, r0b);
    const V2f ey = xb - xa;                                                     // d/d dim1 at z0, z1
    const V2f yv = __builtin_elementwise_fma(tb, ey, xa);
    const float ez = yv.y - yv.x;                                               // d/d dim2
    TriSample s;
    s.v = __builtin_fmaf(ct, ez, yv.x);
    if (GRAD) {
        const V2f dy = __builtin_elementwise_fma(tb, db - da, da);
        const float g0 = __builtin_fmaf(ct, dy.y - dy.x, dy.x), g1 = __builtin_fmaf(ct, ey.y - ey.x, ey.x);
        s.g0 = keep0 ? g0 : 0.f;
        s.g1 = keep1 ? g1 : 0.f;
        s.g2 = keep2 ? ez : 0.f;
    } else {
        s.g0 = s.g1 = s.g2 = 0.f;
    }
    return s;
}

// Loads at a 32-bit BYTE offset from a wave-uniform base (byte offsets are < 2^32: check_common keeps element
// offsets below 2^30), so that the compiler emits `global_load v, v_off, s[base]` (SGPR base + 32-bit VGPR offset)
// instead of building a 64-bit address per lane per load.
__device__ __forceinline__ float ldb_f32(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
}
__device__ __forceinline__ float2 ldb_f32x2(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
}
__device__ __forceinline__ F2a4 ldb_f32x2u(const float *base, unsigned byte_off) // 8 bytes at a 4-byte-aligned offset
{
    return *reinterpret_cast<const F2a4 *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
}
struct __attribute__((aligned(8))) F4a8 { // a float4 that is only 8-byte aligned (a PAIRED row at an odd column)
    float x, y, z, w;
};
__device__ __forceinline__ F4a8 ldb_f32x4(const float *base, unsigned byte_off)
{
    return *reinterpret_cast<const F4a8 *>(reinterpret_cast<const char *>(base) + (size_t)byte_off);
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
// ZCONST: the ray does not move along dim 2 (direction component exactly 0 -- every fan of the reference,
// src/cone.py:258): p2 = s2 + k*0 = s2 for every sample, so that axis' cell, weight and offset are computed once.
template <int C, int SAMPLER, int LAYOUT, bool GRAD, int PM, bool ZCONST>
__device__ __forceinline__ void gather_interleaved_z(const Args &A, int seg0, int segN, const Pose &ps, int lane, float (&z)[C],
                                                     float (&g0)[C], float (&g1)[C], float (&g2)[C])
{
    constexpr int G = (C < DIFFUS_GATHER_GROUP) ? C : DIFFUS_GATHER_GROUP;
    constexpr int NV = (SAMPLER == DIFFUS_NEAREST) ? 1 : 8;
    const float *__restrict__ vol = A.vol;
    const float kf0 = (float)(A.start + seg0 + lane);
#pragma unroll
    for (int gb = 0; gb < C; gb += G) {
        float raw[G][NV];
        float ta[G], tb[G], tc[G]; // interpolation weights, kept for phase B
        bool k0[G], k1[G], k2[G];  // border rule per axis (gradient only): false = zero that component
        // ---- phase A: addresses + loads
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            // step index as a float (exact: k < 2^24).  Lanes past the end of the ray sample further along it -- any
            // point is clamped into the volume, so the loads stay in bounds -- and are masked in phase B.
            const float kf = kf0 + (float)((gb + jj) * kWave);
            const float p0 = ray_point_f<PM>(ps, 0, kf), p1 = ray_point_f<PM>(ps, 1, kf);
            const float p2 = ray_point_f<PM>(ps, 2, ZCONST ? 0.f : kf); // ZCONST: loop-invariant, hoisted with all that follows from it
            if constexpr (SAMPLER == DIFFUS_NEAREST) {
                int i0 = nearest_index(p0, A.G.d0), i1 = nearest_index(p1, A.G.d1), i2 = nearest_index(p2, A.G.d2);
                raw[jj][0] = ldb_f32(vol, part_x<LAYOUT>(A.G, i0) + part_y<LAYOUT>(A.G, i1) + part_z<LAYOUT>(i2));
            } else {
                const AxisG a = tri_axis_g(p0, A.G.d0), b = tri_axis_g(p1, A.G.d1), c = tri_axis_g(p2, A.G.d2);
                const unsigned x0 = part_x<LAYOUT>(A.G, a.i0), x1 = part_x<LAYOUT>(A.G, a.i1);
                const unsigned y0 = part_y<LAYOUT>(A.G, b.i0), y1 = part_y<LAYOUT>(A.G, b.i1);
                const unsigned z0 = part_z<LAYOUT>(c.i0);
                if constexpr (LAYOUT == DIFFUS_PAIRED) { // TWO 16-byte loads: rows x0 and x1, each (y0, y1) x (z0, z1)
#ifdef DIFFUS_ABLATE_LOADS // timing probe (tools/): the addresses are computed, nothing is loaded
                    F4a8 q0{__uint_as_float(x0 + y0 + z0), 1.f, __uint_as_float(y1), 1.f}, q1{__uint_as_float(x1 + y0 + z0), 1.f, 1.f, 1.f};
#else
                    // (Tried: skipping loads whose weight is 0 for all 64 lanes -- rays that have left the volume --
                    // behind wave-uniform branches, and fetching the second column by an exec-masked extra load
                    // instead of storing it twice: hipcc then waits for the loads at every branch merge, 19.9 -> 24.1
                    // and 13.6 -> 21.2 us.  The loads stay unconditional.)
                    // (x + (y + z) offsets: the dim-0 part as two chained v_mad_u32_u24 on top of the shared (y + z) part --
                    // part_x() + a three-operand add was two multiplies and a v_add3 per row, each a half-rate instruction)
                    const unsigned syB_u = (unsigned)__builtin_amdgcn_readfirstlane((int)A.G.syB);
                    const unsigned yz = mad_u24_s((unsigned)b.i0 >> 2, syB_u, (((unsigned)b.i0 & 3u) << 3) + z0);
                    (void)y0;
                    const unsigned sxB_u = (unsigned)__builtin_amdgcn_readfirstlane((int)A.G.sxB);
                    const F4a8 q0 = ldb_f32x4(vol, mad_u24_s((unsigned)a.i0 >> 2, sxB_u, mad_u24_40((unsigned)a.i0 & 3u, yz)));
                    const F4a8 q1 = ldb_f32x4(vol, mad_u24_s((unsigned)a.i1 >> 2, sxB_u, mad_u24_40((unsigned)a.i1 & 3u, yz)));
                    (void)y1; (void)x0; (void)x1;
#endif
                    raw[jj][0] = q0.x; raw[jj][1] = q0.y; raw[jj][2] = q0.z; raw[jj][3] = q0.w;
                    raw[jj][4] = q1.x; raw[jj][5] = q1.y; raw[jj][6] = q1.z; raw[jj][7] = q1.w;
                } else if constexpr (LAYOUT == DIFFUS_CANONICAL) {
                    // dim 2 is contiguous: the two depth neighbours of a column come in ONE 8-byte load at the pair
                    // (b, b + 1), b = min(z0, d2 - 2) -- four wave-loads per sample instead of eight (the gather runs
                    // at the texture addresser's rate per wave-load).  At the far face (z0 = d2 - 1 = b + 1) both
                    // values are the pair's second; d2 = 1 has no pair and keeps the dword loads.
                    const unsigned c00 = x0 + y0, c01 = x0 + y1, c10 = x1 + y0, c11 = x1 + y1;
                    if (A.G.d2 >= 2) { // kernel argument: uniform over the grid
                        const int bz = min(c.i0, A.G.d2 - 2);
                        const unsigned zb = (unsigned)bz << 2;
                        const bool far = c.i0 != bz;
                        const F2a4 p00 = ldb_f32x2u(vol, c00 + zb), p01 = ldb_f32x2u(vol, c01 + zb);
                        const F2a4 p10 = ldb_f32x2u(vol, c10 + zb), p11 = ldb_f32x2u(vol, c11 + zb);
                        raw[jj][0] = far ? p00.y : p00.x; raw[jj][1] = p00.y;
                        raw[jj][2] = far ? p01.y : p01.x; raw[jj][3] = p01.y;
                        raw[jj][4] = far ? p10.y : p10.x; raw[jj][5] = p10.y;
                        raw[jj][6] = far ? p11.y : p11.x; raw[jj][7] = p11.y;
                    } else {
                        raw[jj][0] = raw[jj][1] = ldb_f32(vol, c00); raw[jj][2] = raw[jj][3] = ldb_f32(vol, c01);
                        raw[jj][4] = raw[jj][5] = ldb_f32(vol, c10); raw[jj][6] = raw[jj][7] = ldb_f32(vol, c11);
                    }
                } else {
                    const unsigned z1 = part_z<LAYOUT>(c.i1);
                    const unsigned c00 = x0 + y0, c01 = x0 + y1, c10 = x1 + y0, c11 = x1 + y1;
#ifdef DIFFUS_ABLATE_LOADS
                raw[jj][0] = __uint_as_float(c00 + z0); raw[jj][1] = __uint_as_float(c00 + z1);
                raw[jj][2] = __uint_as_float(c01 + z0); raw[jj][3] = __uint_as_float(c01 + z1);
                raw[jj][4] = __uint_as_float(c10 + z0); raw[jj][5] = __uint_as_float(c10 + z1);
                raw[jj][6] = __uint_as_float(c11 + z0); raw[jj][7] = __uint_as_float(c11 + z1);
#else
                raw[jj][0] = ldb_f32(vol, c00 + z0); raw[jj][1] = ldb_f32(vol, c00 + z1);
                raw[jj][2] = ldb_f32(vol, c01 + z0); raw[jj][3] = ldb_f32(vol, c01 + z1);
                raw[jj][4] = ldb_f32(vol, c10 + z0); raw[jj][5] = ldb_f32(vol, c10 + z1);
                raw[jj][6] = ldb_f32(vol, c11 + z0); raw[jj][7] = ldb_f32(vol, c11 + z1);
#endif
                }
                ta[jj] = a.t; tb[jj] = b.t; tc[jj] = c.t;
                if (GRAD) { // the low-side tests, as lane masks (SGPR pairs: they cost no VGPR)
                    k0[jj] = a.lo; k1[jj] = b.lo; k2[jj] = c.lo;
                }
            }
        }
        // ---- phase B: interpolation
#pragma unroll
        for (int jj = 0; jj < G; ++jj) {
            const int j = gb + jj;
            // Samples past the end of the ray (j * 64 + lane >= segN) keep whatever their clamped re-read gave: every
            // consumer masks by the sample index itself (reflect_chunk: r = 0 there; the frame, zbar, loss and
            // pose-gradient sums: n < segN) and drops non-finite products, so no select is spent on them here (it was
            // four v_cndmask per sample, 4.25 cycles each: tools/valu_issue_bench.hip).
            (void)segN;
            if constexpr (SAMPLER == DIFFUS_NEAREST) {
                z[j] = raw[jj][0];
                if (GRAD) g0[j] = g1[j] = g2[j] = 0.f;
            } else {
#ifdef DIFFUS_ABLATE_LERP
                TriSample sm;
                sm.v = raw[jj][0] + raw[jj][1] + raw[jj][2] + raw[jj][3] + raw[jj][4] + raw[jj][5] + raw[jj][6] + raw[jj][7] + ta[jj] + tb[jj] + tc[jj];
                sm.g0 = sm.g1 = sm.g2 = 0.f;
#else
                TriSample sm = tri_lerp<GRAD>(raw[jj], ta[jj], tb[jj], tc[jj], GRAD ? k0[jj] : true, GRAD ? k1[jj] : true, GRAD ? k2[jj] : true);
#endif
                z[j] = sm.v;
                if (GRAD) {
                    g0[j] = sm.g0; g1[j] = sm.g1; g2[j] = sm.g2;
                    // Materialise the three gradient components HERE.  Left alone, LLVM sinks these lerps down to
                    // their only use (the pose-gradient sum at the end of the backward) and keeps the 8 corner values
                    // and 3 weights of every sample alive across the whole scan instead: 15 registers per sample, not 3
                    // (render_bwd_kernel<8,...,GPOSE>: 237 VGPRs, 2 waves per SIMD).
                    asm volatile("" : "+v"(g0[j]), "+v"(g1[j]), "+v"(g2[j]));
                }
            }
        }
    }
}

template <int C, int SAMPLER, int LAYOUT, bool GRAD, int PM>
__device__ __forceinline__ void gather_interleaved(const Args &A, int seg0, int segN, const Pose &ps, int lane, float (&z)[C],
                                                   float (&g0)[C], float (&g1)[C], float (&g2)[C])
{
    // wave-uniform (the pose sits in SGPRs): a whole wave takes one of the two copies of the gather
#ifdef DIFFUS_COUNT_PLANAR // tools/kernel_resources.py -DDIFFUS_COUNT_PLANAR=1|0: static count of ONE of the two copies
    const bool planar = DIFFUS_COUNT_PLANAR;
#else
    const bool planar = (PM == 0 || ps.pmode != 2) ? (ps.df[2] == 0.f) : (ps.dd[2] == 0.0);
#endif
    if (planar)
        gather_interleaved_z<C, SAMPLER, LAYOUT, GRAD, PM, true>(A, seg0, segN, ps, lane, z, g0, g1, g2);
    else
        gather_interleaved_z<C, SAMPLER, LAYOUT, GRAD, PM, false>(A, seg0, segN, ps, lane, z, g0, g1, g2);
}

// r'_{n-1} for the lane's samples (CHUNKED): r[j] couples sample n-1 and n
// (n = lane*C+j).  r = 0 (identity transfer matrix) for n = 0 and n >= N1; with
// start > 0 the first kept coefficient is replaced by the per-pose median
// (reference :243-244).
// inv_out (optional): 1 / (Z_{n-1} + Z_n) of every sample as the reflection coefficient used it (the backward needs it
// again for d r / d Z: a second v_rcp_f32 per sample is 8.2 issue cycles, an LDS round trip of the chunk is four slots)
template <int C>
__device__ __forceinline__ void reflect_chunk(const Args &A, int seg0, int segN, int n0, const float (&z)[C], float zprev, float medv,
                                              float (&r)[C], float *inv_out = nullptr)
{
    // n0 = lane*C is the segment-local index of the lane's first sample; the global cropped index is seg0 + n0 + j
#pragma unroll
    for (int j = 0; j < C; ++j) {
        int nl = n0 + j, n = seg0 + nl;
        float zp = (j == 0) ? zprev : z[j == 0 ? 0 : j - 1];
        const float inv = __builtin_amdgcn_rcpf(zp + z[j]);
        float v = (z[j] - zp) * inv; // = reflect_fast(zp, z[j])
        if (inv_out) inv_out[j] = inv;
        if (n == 1 && A.start > 0) v = medv;
        r[j] = (n >= 1 && nl < segN) ? v : 0.f;
    }
}

// Echo series of one ray spread over a wave (replaces the N+1
// dense solves of reference src/renderer.py:367-457).  r[j] is the reflection
// coefficient entering sample n = lane*C + j (0 where there is none); e[j] gets
// echo_n = (P_n)01/(P_n)11 with NaN -> 0 (reference :408).
struct NoScanHook {
    __device__ __forceinline__ bool operator()(const Mat &, Mat &) const { return false; }
};
// after_scan(Lincl, carry): called once between the wave scan and the sweep with this lane's INCLUSIVE prefix; it may
// return true and a matrix that precedes the whole wave (the SPLIT kernels exchange the first half's total there).
template <int C, bool FAST = false, typename Hook = NoScanHook>
__device__ __forceinline__ void echo_chunk(const float (&r)[C], int lane, float (&e)[C], const Mat *carry_in = nullptr,
                                           int last = -1, Mat *carry_out = nullptr, Hook &&after_scan = Hook())
{
    // local product of the chunk, then inclusive scan over lanes (lower lanes on the left)
    // FAST (the render kernel): rescale every 4th step only.  The ratio b/d does not depend on the scale and one
    // step multiplies the largest entry by at most max(2, 2r^2 + |r|), so four steps from a normalised matrix cannot
    // overflow for any |r| < 1e4; the per-step rescale was 14 x 7 of the forward's 1200 VALU instructions.
    Mat L = mat_identity();
#pragma unroll
    for (int j = 0; j < C; ++j) {
        L = mat_step(L, r[j]);
        if (!FAST || (j & 3) == 3 || j == C - 1) mat_renorm(L);
    }
    {
#define DIFFUS_ROUND(CTRL, RMASK, HAS, RN)           \
    {                                                \
        const Mat o = mat_dpp_ident<CTRL, RMASK>(L); \
        L = mat_mul(o, L);                           \
        if (RN) mat_renorm(L);                       \
    }
        DIFFUS_SCAN_UP_ROUNDS(lane, DIFFUS_ROUND)
#undef DIFFUS_ROUND
    }
    Mat Pm = mat_lane_prev(L, mat_identity()); // exclusive prefix; lane 0: identity
    Mat hooked;
    const bool has_hooked = after_scan(L, hooked);
    if (carry_in || has_hooked) { // segment > 0: everything is preceded by the product of the earlier segments
        Pm = mat_mul(carry_in ? *carry_in : hooked, Pm);
        mat_renorm(Pm);
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
        Pm = mat_step(Pm, r[j]);
        if (!FAST || (j & 3) == 3 || j == C - 1) mat_renorm(Pm);
        float v = FAST ? fast_div(Pm.b, Pm.d) : __fdiv_rn(Pm.b, Pm.d);
        e[j] = (v == v) ? v : 0.f; // nan_to_num(nan=0)
        if (carry_out && lane * C + j == last) *carry_out = Pm;
    }
}

// ---- the echo series of an ILL-CONDITIONED ray, in float64 (wave-uniform rare path of the scans) -----------------------
// echo_n = b_n / d_n; on a ray that grazes the skull d_n is nearly cancelled for a sample or two (|echo| = 123 and 287 on the
// two worst rays of BASELINE config 3, ~0.05 elsewhere) and every float32 evaluation carries (condition number) x eps of
// noise there -- the reference's own dense LU 4.2e-5 of the ray's peak (golden G19), the wave scan's tree of 2x2 products
// 2.5x that.  A wave that sees |echo| > kEchoRecheck anywhere evaluates the SAME scan again in float64 and keeps those echoes: the result is then the
// float64 series of its float32 reflection coefficients, i.e. better conditioned than any float32 evaluation, the
// reference's included.  Well-conditioned rays never take the branch and stay bit-identical.
// The threshold: the float32 evaluation's error grows like 1.5e-5 x |echo| of the ray's peak (tools/fuzz_forward.py: 6.0e-5 against
// float64 on a ray with |echo| = 4.0, where the oracle's float32 series is at 1e-5); at 8 (rounds 4-5) rays between 3 and 8
// could exceed the 5e-5 the tests hold a frame to; at 2 one launch in 36 000 still does (a ray with |echo| = 1.8 at 8.8e-5); at 1
// none.  63 rays instead of 9 of config 3's 8192 take the branch: the forward launch 15.4 -> 16.7 us.
#ifndef DIFFUS_ECHO_RECHECK
#define DIFFUS_ECHO_RECHECK 1.f
#endif
constexpr float kEchoRecheck = DIFFUS_ECHO_RECHECK;
struct DMat {
    double a, b, c, d;
};
__device__ __forceinline__ DMat dmat_mul(const DMat &x, const DMat &y)
{
    return DMat{__builtin_fma(x.a, y.a, x.b * y.c), __builtin_fma(x.a, y.b, x.b * y.d), __builtin_fma(x.c, y.a, x.d * y.c),
                __builtin_fma(x.c, y.b, x.d * y.d)};
}
__device__ __forceinline__ DMat dmat_step(const DMat &p, double r) // P * M(r)
{
    const double a = 1.0 - (2.0 * r) * r;
    return DMat{__builtin_fma(p.a, a, -(p.b * r)), __builtin_fma(p.a, r, p.b), __builtin_fma(p.c, a, -(p.d * r)), __builtin_fma(p.c, r, p.d)};
}
__device__ __forceinline__ void dmat_renorm(DMat &m) // exact power-of-two rescale (the ratio b / d does not see it)
{
    const double mx = fmax(fmax(fabs(m.a), fabs(m.b)), fmax(fabs(m.c), fabs(m.d)));
    const int ex = (mx > 0.0 && mx < __builtin_inf()) ? ilogb(mx) : 0;
    m.a = ldexp(m.a, -ex); m.b = ldexp(m.b, -ex); m.c = ldexp(m.c, -ex); m.d = ldexp(m.d, -ex);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_d(double old, double v)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ DMat dmat_dpp_ident(const DMat &m) // the identity where a lane has no source
{
    return DMat{dpp_mov_d<CTRL, ROW_MASK>(1.0, m.a), dpp_mov_d<CTRL, ROW_MASK>(0.0, m.b), dpp_mov_d<CTRL, ROW_MASK>(0.0, m.c),
                dpp_mov_d<CTRL, ROW_MASK>(1.0, m.d)};
}
// r[j]: the reflection coefficient entering sample n = lane * C + j (0 where there is none); e[j] gets echo_n (NaN -> 0,
// reference :408).  carry: a product that precedes the whole wave (nullable).
// last (nullable): gets the product up to and including the wave's last sample, every lane the same value (the carry of the next
// piece of a row that is walked in several).
template <int C, typename T>
__device__ __forceinline__ void echo_chunk_f64(const T (&r)[C], int lane, float (&e)[C], const DMat *carry = nullptr, DMat *last = nullptr)
{
    DMat L{1.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int j = 0; j < C; ++j) L = dmat_step(L, (double)r[j]);
    dmat_renorm(L);
#define DIFFUS_ROUND(CTRL, RMASK, HAS, RN)             \
    {                                                  \
        const DMat o = dmat_dpp_ident<CTRL, RMASK>(L); \
        L = dmat_mul(o, L);                            \
        if (RN) dmat_renorm(L);                        \
    }
    DIFFUS_SCAN_UP_ROUNDS(lane, DIFFUS_ROUND)
#undef DIFFUS_ROUND
    DMat Pm{dpp_mov_d<kDppWaveShr1, 0xf>(1.0, L.a), dpp_mov_d<kDppWaveShr1, 0xf>(0.0, L.b), dpp_mov_d<kDppWaveShr1, 0xf>(0.0, L.c),
            dpp_mov_d<kDppWaveShr1, 0xf>(1.0, L.d)}; // exclusive prefix; lane 0: identity
    if (carry) {
        Pm = dmat_mul(*carry, Pm);
        dmat_renorm(Pm);
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
        Pm = dmat_step(Pm, (double)r[j]);
        if ((j & 3) == 3) dmat_renorm(Pm);
        const double v = Pm.b / Pm.d;
        e[j] = (v == v) ? (float)v : 0.f;
    }
    if (last) {
        dmat_renorm(Pm);
        auto bc = [](double x) {
            const long long b = __double_as_longlong(x);
            const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), kWave - 1), hi = __builtin_amdgcn_readlane((int)(b >> 32), kWave - 1);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
        *last = DMat{bc(Pm.a), bc(Pm.b), bc(Pm.c), bc(Pm.d)};
    }
}
// The float64 series of such a ray from its float32 impedance samples -- the samples taken AGAIN, with the stage-wise sampler
// (tri_sample: the oracle's lerp sequence bit for bit, i.e. the reference's grid_sample / nearest gather): on these rays the
// last bit of a sample is worth as much as the scan's own noise (the fused gather's one-fma lerps differ from that sequence by
// a rounding: 1e-4 of the ray's peak on pose 18 of config 3 with the scan already in float64).
// A cold branch (__builtin_expect at the call sites: what the register allocator must move is moved around THIS block, not
// inside the hot code) -- but a wave that takes it is likely to be the kernel's TAIL, so it is written for its own latency: every
// sample requested before the first is used, no IEEE division.  (As a called function the kernels inherit its 178 VGPRs; with
// rolled loops that sample one point at a time the nine such waves of config 3 held the scan kernel for 25 us.)
// CHUNKED mapping: lane owns samples n0 .. n0 + C - 1 of the segment; e[j] gets echo_n (NaN -> 0).
// A ray walked in several pieces (render_fwd_long_repair_kernel): carry = the float64 product of the earlier pieces, zcarry = their
// last impedance sample; last / zlast receive this piece's (all lanes the same values).
template <int C, int SAMPLER, int LAYOUT, int PM>
__device__ __forceinline__ void echo_f64_rare(const Args &A, const Pose &ps, int seg0, int segN, int n0, float medv, float (&e)[C],
                                              const DMat *carry = nullptr, DMat *last = nullptr, const float *zcarry = nullptr,
                                              float *zlast = nullptr)
{
    const int lane = n0 / C;
    // all samples first (one memory round trip: a wave that takes this path is the kernel's tail), then the coefficients
    float z[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const int k = A.start + seg0 + n0 + j; // (past the end of the ray: any point is clamped into the volume; masked below)
        const float p0 = ray_point<PM>(ps, 0, k), p1 = ray_point<PM>(ps, 1, k), p2 = ray_point<PM>(ps, 2, k);
        if (SAMPLER == DIFFUS_NEAREST)
            z[j] = A.vol[vox_off<LAYOUT>(A.G, nearest_index(p0, A.G.d0), nearest_index(p1, A.G.d1), nearest_index(p2, A.G.d2))];
        else
            z[j] = tri_sample<LAYOUT, false>(A.vol, A.G, p0, p1, p2).v;
    }
    float zprev = lane_prev(z[C - 1], z[C - 1]);
    if (zcarry && lane == 0) zprev = *zcarry;
    if (zlast) { // the sample at segN - 1 (lane and slot wave-uniform)
        const int ll = (segN - 1) / C, lj = (segN - 1) % C;
        float zl = z[0];
#pragma unroll
        for (int j = 1; j < C; ++j) zl = (j == lj) ? z[j] : zl;
        *zlast = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(zl), ll));
    }
    double r[C];
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const int nl = n0 + j, n = seg0 + nl;
        const double zp = (double)((j == 0) ? zprev : z[j == 0 ? 0 : j - 1]), zc = (double)z[j];
        // (zc - zp) / (zp + zc): v_rcp_f64 and two Newton steps (to the last bit or two of a double: far below the float32 samples'
        // own rounding) instead of the ~40 instructions of an IEEE division
        const double den = zp + zc;
        double x = __builtin_amdgcn_rcp(den);
        x = x * __builtin_fma(-den, x, 2.0);
        x = x * __builtin_fma(-den, x, 2.0);
        double v = (zc - zp) * x;
        if (n == 1 && A.start > 0) v = (double)medv;
        r[j] = (n >= 1 && nl < segN) ? v : 0.0;
    }
    DMat L{1.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int j = 0; j < C; ++j) L = dmat_step(L, r[j]);
    dmat_renorm(L);
#define DIFFUS_ROUND(CTRL, RMASK, HAS, RN)             \
    {                                                  \
        const DMat o = dmat_dpp_ident<CTRL, RMASK>(L); \
        L = dmat_mul(o, L);                            \
        if (RN) dmat_renorm(L);                        \
    }
    DIFFUS_SCAN_UP_ROUNDS(lane, DIFFUS_ROUND)
#undef DIFFUS_ROUND
    DMat Pm{dpp_mov_d<kDppWaveShr1, 0xf>(1.0, L.a), dpp_mov_d<kDppWaveShr1, 0xf>(0.0, L.b), dpp_mov_d<kDppWaveShr1, 0xf>(0.0, L.c),
            dpp_mov_d<kDppWaveShr1, 0xf>(1.0, L.d)}; // exclusive prefix; lane 0: identity
    if (carry) {
        Pm = dmat_mul(*carry, Pm);
        dmat_renorm(Pm);
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
        Pm = dmat_step(Pm, r[j]);
        if ((j & 3) == 3) dmat_renorm(Pm);
        // b and d are accurate now; their quotient needs no more than float32 (entries are within 2^4 of 1 in magnitude range)
        const float v = fast_div((float)Pm.b, (float)Pm.d);
        e[j] = (v == v) ? v : 0.f;
    }
    if (last) { // (coefficients past segN are 0: identity steps -- lane 63 ends on the product up to the piece's last sample)
        dmat_renorm(Pm);
        auto bc = [](double x) {
            const long long b = __double_as_longlong(x);
            const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffll), kWave - 1), hi = __builtin_amdgcn_readlane((int)(b >> 32), kWave - 1);
            return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
        };
        *last = DMat{bc(Pm.a), bc(Pm.b), bc(Pm.c), bc(Pm.d)};
    }
}
// wave-uniform: does any lane hold an echo that asks for the float64 evaluation?
template <int C>
__device__ __forceinline__ bool echo_needs_f64(const float (&e)[C])
{
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < C; ++j) m = fmaxf(m, fabsf(e[j])); // (NaN-free: echo_chunk has zeroed them)
    return __builtin_amdgcn_ballot_w64(m > kEchoRecheck) != 0ull;
}

// ---- patches of the (ray, step) grid: shared by the gradient scatter and the splat winner kernel ----
// 32 adjacent rays x 32 consecutive steps: at unit steps and the demos' ray spacing this footprint is about square,
// i.e. the smallest bounding box for 1024 samples (16 x 64: scatter 52 us, 32 x 32: 49, 64 x 16: 53)
#ifndef DIFFUS_PATCH_RAYS
#define DIFFUS_PATCH_RAYS 32
#endif
#ifndef DIFFUS_PATCH_STEPS
#define DIFFUS_PATCH_STEPS 32
#endif
// 24 KiB: 6 blocks per CU.  With the square patches 85 % of the boxes fit (the rest takes 2 or 4 passes) and the extra
// blocks in flight are worth more than the saved passes (scatter at config 3: 48 KiB 49 us, 32 KiB 42, 24 KiB 39, 16 KiB 44)
#ifndef DIFFUS_TILE_CAP
#define DIFFUS_TILE_CAP (6 * 1024)
#endif
constexpr int kPatchRays = DIFFUS_PATCH_RAYS;
constexpr int kPatchSteps = DIFFUS_PATCH_STEPS;
constexpr int kTileCap = DIFFUS_TILE_CAP; // tile entries
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


// ---- float64 repair of an ill-conditioned ray's frame row (one-pass step; called from pose_finish_block by ONE wave, all lanes
// active).  The scan kernel flags rays with |echo| > kEchoRecheck; here the row is evaluated again as the float64 pipeline
// from float32 samples taken with the ORACLE's lerp sequence (the stage-wise tri_sample), the frame row and the ray's loss term are overwritten.  The gradients keep the float32 scan's values (they carry
// the same condition number whatever the arithmetic).  Off the critical path -- the per-pose blocks sit at the head of the
// scatter launch.
// It shares its kernels' register budget (the scatter's patch path: 80) and must not spill: scratch under every wave of a launch
// is paid by all of them (as a called function with 1.2 KB of stack the scatter went from 26 to 98 us), and what the register
// allocator moves for it, it moves in the patch path too (+2.5 us with ~100 registers of unrolled float64 state inlined).  So:
// the ray's samples wait in LDS (zbuf: 1024 floats of the caller's) and every loop over a lane's 16 samples is ROLLED -- two
// 2x2 matrices of doubles and a sample's worth of temporaries are all that is live.  ~10 us per ray, beside the first patches.
// element offset of voxel (x, y, z) for a layout known at run time only, branch-free (all three forms, two selects): the eight
// corner loads of a sample stay in flight together, and the per-pose blocks carry ONE copy of the repair
__device__ __forceinline__ unsigned vox_off_rt(int layout, const Geom &G, int x, int y, int z)
{
    const unsigned c = vox_off<DIFFUS_CANONICAL>(G, x, y, z), p = vox_off<DIFFUS_PAIRED>(G, x, y, z), b = vox_off<DIFFUS_BRICKED>(G, x, y, z);
    return layout == DIFFUS_CANONICAL ? c : (layout == DIFFUS_PAIRED ? p : b);
}
template <int SAMPLER>
__device__ __forceinline__ float sample_rt(const Args &A, const Pose &ps, int k)
{
    const float p0 = ray_point(ps, 0, k), p1 = ray_point(ps, 1, k), p2 = ray_point(ps, 2, k);
    const int lay = A.vol_layout;
    if (SAMPLER == DIFFUS_NEAREST)
        return A.vol[vox_off_rt(lay, A.G, nearest_index(p0, A.G.d0), nearest_index(p1, A.G.d1), nearest_index(p2, A.G.d2))];
    const Axis a = tri_axis(p0, A.G.d0), b = tri_axis(p1, A.G.d1), c = tri_axis(p2, A.G.d2);
    const unsigned o000 = vox_off_rt(lay, A.G, a.i0, b.i0, c.i0), o001 = vox_off_rt(lay, A.G, a.i0, b.i0, c.i1), o010 = vox_off_rt(lay, A.G, a.i0, b.i1, c.i0),
                   o011 = vox_off_rt(lay, A.G, a.i0, b.i1, c.i1), o100 = vox_off_rt(lay, A.G, a.i1, b.i0, c.i0), o101 = vox_off_rt(lay, A.G, a.i1, b.i0, c.i1),
                   o110 = vox_off_rt(lay, A.G, a.i1, b.i1, c.i0), o111 = vox_off_rt(lay, A.G, a.i1, b.i1, c.i1);
    const float v000 = A.vol[o000], v001 = A.vol[o001], v010 = A.vol[o010], v011 = A.vol[o011];
    const float v100 = A.vol[o100], v101 = A.vol[o101], v110 = A.vol[o110], v111 = A.vol[o111];
    // tri_sample's sequence (= oracle/diffus_oracle.c orc_sample_trilinear): dim 2, dim 1, dim 0, a + t (b - a) as a separate multiply and add
    const float e00 = v001 - v000, e01 = v011 - v010, e10 = v101 - v100, e11 = v111 - v110;
    const float c00 = __fadd_rn(v000, __fmul_rn(c.t, e00)), c01 = __fadd_rn(v010, __fmul_rn(c.t, e01));
    const float c10 = __fadd_rn(v100, __fmul_rn(c.t, e10)), c11 = __fadd_rn(v110, __fmul_rn(c.t, e11));
    const float f0 = c01 - c00, f1 = c11 - c10;
    const float q0 = __fadd_rn(c00, __fmul_rn(b.t, f0)), q1 = __fadd_rn(c10, __fmul_rn(b.t, f1));
    return __fadd_rn(q0, __fmul_rn(a.t, q1 - q0));
}
template <int SAMPLER, int C> // C samples per lane: 64 C >= N1 (8 for rays of up to 512 samples: half the serial chain of 16)
__device__ __forceinline__ void repair_ray_f64(const Args &A, long pose, long w, float *zbuf)
{
    const int lane = threadIdx.x & 63;
    const int N1 = A.N1, n0 = lane * C;
    Pose ps;
    load_pose(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
    const float medv = (A.start > 0) ? A.med[pose] : 0.f;
    // samples, INTERLEAVED (n = i 64 + lane), four in flight
#pragma unroll 4
    for (int i = 0; i < C; ++i) {
        const int n = i * kWave + lane; // (past the end of the ray: any point is clamped into the volume; masked by coeff)
        zbuf[n] = sample_rt<SAMPLER>(A, ps, A.start + n);
    }
    wave_lds_sync();
    auto coeff = [&](int n, double zp, double zc) -> double { // r entering sample n (reference :33, :243-244)
        const double den = zp + zc;
        double x = __builtin_amdgcn_rcp(den); // + two Newton steps: to the last bits of a double
        x = x * __builtin_fma(-den, x, 2.0);
        x = x * __builtin_fma(-den, x, 2.0);
        double v = (zc - zp) * x;
        if (n == 1 && A.start > 0) v = (double)medv;
        return (n >= 1 && n < N1) ? v : 0.0;
    };
    DMat L{1.0, 0.0, 0.0, 1.0};
    {
        double zp = (double)zbuf[max(n0 - 1, 0)];
#pragma unroll 1
        for (int j = 0; j < C; ++j) {
            const double zc = (double)zbuf[n0 + j];
            L = dmat_step(L, coeff(n0 + j, zp, zc));
            zp = zc;
            if ((j & 3) == 3) dmat_renorm(L);
        }
    }
#define DIFFUS_ROUND(CTRL, RMASK, HAS, RN)             \
    {                                                  \
        const DMat o = dmat_dpp_ident<CTRL, RMASK>(L); \
        L = dmat_mul(o, L);                            \
        if (RN) dmat_renorm(L);                        \
    }
    DIFFUS_SCAN_UP_ROUNDS(lane, DIFFUS_ROUND)
#undef DIFFUS_ROUND
    DMat Pm{dpp_mov_d<kDppWaveShr1, 0xf>(1.0, L.a), dpp_mov_d<kDppWaveShr1, 0xf>(0.0, L.b), dpp_mov_d<kDppWaveShr1, 0xf>(0.0, L.c),
            dpp_mov_d<kDppWaveShr1, 0xf>(1.0, L.d)}; // exclusive prefix; lane 0: identity
    float ssq = 0.f;
    float *const frow = A.frame + w * N1;
    const float *const trow = A.target ? A.target + w * N1 : nullptr;
    double zp = (double)zbuf[max(n0 - 1, 0)];
#pragma unroll 1
    for (int j = 0; j < C; ++j) {
        const int n = n0 + j;
        const double zc = (double)zbuf[n];
        Pm = dmat_step(Pm, coeff(n, zp, zc));
        zp = zc;
        if ((j & 3) == 3) dmat_renorm(Pm);
        float e = fast_div((float)Pm.b, (float)Pm.d); // b and d are accurate now; their quotient needs no more than float32
        e = (e == e) ? e : 0.f;                       // nan_to_num (reference :408)
        if (n < N1) {
            const float fr = __fmul_rn(e, fast_exp2(A.neg_alpha_l2e * (float)n)); // the scan kernels' attenuation
            const float dlt = fr - (trow ? trow[n] : 0.f);
            ssq = __builtin_fmaf(dlt, dlt, ssq);
            frow[n] = fr;
        }
    }
    ssq = wave_sum_to_lane63(ssq);
    if (lane == kWave - 1) {
        A.loss_part[w * 2] = A.loss_scale * ssq;
        A.loss_part[w * 2 + 1] = 0.f;
    }
    wave_lds_sync(); // zbuf may be refilled by this wave's next ray
}

// What is left of a pose's backward once every ray's adjoint scan has run (one block per pose; the first blocks of the
// scatter launch, or pose_finish_kernel when there is no scatter):
//   * start > 0: the first kept reflection coefficient of every ray was replaced by the per-pose median (reference
//     src/renderer.py:243-244), so their gradients were summed into gmed[pose]; torch.median routes that sum to the
//     ray that supplied the median.  Its two samples and their spatial gradients wait in medinfo (median_kernel), so
//     nothing is sampled again: d r / d Z -> volume gradient (8 corner atomics per sample), d/d source and d/d direction
//     of that ray;
//   * d/d source[pose] = fixed-order sum over rays of the per-ray partials (+ the median ray's extra term).
// GLAYOUT is the GRADIENT layout.  sm: 3 * kBlock floats; REPAIR: (blockDim.x / 64) * DIFFUS_MAX_SAMPLES (a row of samples per wave).
template <int SAMPLER, int GLAYOUT, bool REPAIR>
__device__ __forceinline__ void pose_finish_block(const Args &A, int pose, float *sm)
{
    // every sum of the epilogue at once: d/dsource partials (3), loss partials, the median's gradient shares; all loads
    // are issued before the one barrier
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int nt = (int)blockDim.x, tid = (int)threadIdx.x;
    if (REPAIR && A.rflag && A.frame && A.loss_part) { // one-pass step on request (pose_finish_kernel only): the ill-conditioned rays of this pose (rare), dealt out to the block's waves
        const int lane = tid & 63, wib = tid >> 6, nw = nt >> 6;
        int seen = 0;
        for (int base = 0; base < A.R; base += kWave) { // (block-uniform trip count; the ballots below are wave-uniform)
            const int i = base + lane;
            int f = 0;
            if (i < A.R) {
                const int *rf = A.rflag + ((long)pose * A.R + i) * 2;
                f = rf[0] | rf[1];
            }
            unsigned long long m = __builtin_amdgcn_ballot_w64(f != 0);
            while (m) {
                const int b = __builtin_ctzll(m);
                m &= m - 1;
                if (seen++ % nw == wib) {
                    const long wr = (long)pose * A.R + base + b;
                    float *zbuf = sm + wib * DIFFUS_MAX_SAMPLES;
                    if (A.N1 <= 8 * kWave) repair_ray_f64<SAMPLER, 8>(A, pose, wr, zbuf);
                    else repair_ray_f64<SAMPLER, DIFFUS_MAX_SAMPLES / kWave>(A, pose, wr, zbuf);
                }
            }
        }
        __syncthreads(); // the repaired loss terms are read below by other waves of the block
    }
    if (A.gsrc_out)
        for (int i = tid; i < A.R; i += nt) {
            const float *q = A.gsrc_part + ((long)pose * A.R + i) * 3;
            a[0] += q[0]; a[1] += q[1]; a[2] += q[2];
        }
    if (A.loss_out)
        for (int i = tid; i < 2 * A.R; i += nt) a[3] += A.loss_part[(long)pose * 2 * A.R + i];
    if (A.start > 0)
        for (int i = tid; i < A.R; i += nt) a[4] += A.gmed[(long)pose * A.R + i];
    block_sums<5>(a, sm);
    if (threadIdx.x == 0) {
        float gs[3] = {0.f, 0.f, 0.f};
        if (A.start > 0) {
            const int i = A.who[pose];
            const float gm = a[4];
            if (i >= 0 && gm != 0.f && finitef(gm)) {
                const long w = (long)pose * A.R + i;
                Pose ps;
                load_pose(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
                const float *mi = A.medinfo + (long)pose * 8;
                const float z0 = mi[0], z1 = mi[1];
                const float inv = __fdiv_rn(1.f, z0 + z1);
                const float zb[2] = {gm * (-2.f * z1 * inv * inv), gm * (2.f * z0 * inv * inv)}; // d r / d Z (reference :33)
                float gd[3] = {0.f, 0.f, 0.f};
                for (int q = 0; q < 2; ++q) {
                    if (!finitef(zb[q]) || zb[q] == 0.f) continue;
                    const int k = A.start + q;
                    if (A.gvol) {
                        Cell c = cell_of<SAMPLER>(A, ps, k);
                        for_each_corner<SAMPLER>(c, zb[q], [&](int ci, int cj, int ck, float v) {
                            if (v != 0.f) {
                                unsigned g = vox_off<GLAYOUT>(A.G, ci, cj, ck);
                                atomicAdd(A.gvol + g, v);
                                if (GLAYOUT == DIFFUS_BRICKED && A.gtouched) A.gtouched[g >> 5] = 1;
                            }
                        });
                    }
                    if (SAMPLER == DIFFUS_TRILINEAR) {
                        const float kf = (float)k;
                        for (int c = 0; c < 3; ++c) {
                            const float t = zb[q] * mi[2 + 3 * q + c];
                            gs[c] += t;
                            gd[c] += kf * t;
                        }
                    }
                }
                if (SAMPLER == DIFFUS_TRILINEAR && A.gdirs)
                    for (int c = 0; c < 3; ++c) A.gdirs[w * 3 + c] += gd[c]; // this block is the only writer now
            }
        }
        if (A.gsrc_out)
            for (int c = 0; c < 3; ++c) A.gsrc_out[pose * 3 + c] = a[c] + gs[c];
        if (A.loss_out) A.loss_out[pose] = a[3];
    }
}

} // namespace
