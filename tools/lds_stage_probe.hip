// lds_stage_probe.hip -- experiment for DESIGN.md: does staging the impedance lines through LDS beat gathering them
// straight from global memory?  (north_star: "impedance tiles staged through LDS")
//
// Standalone (hipcc --offload-arch=gfx950 tools/lds_stage_probe.hip -o probe; ./probe).  It samples the SAME points
// as the production forward at BASELINE config 3 -- 32 poses on a ring, 256 rays x 512 unit steps, planar fans, a
// 256^3 volume in the PAIRED layout (4 x 4 columns x one depth per 128-B line, each voxel the pair (v[z], v[z+1])) --
// trilinearly, and writes one float per sample.  Nothing else of the forward (no scan): the question is the gather.
//   A  direct      one wave per ray, lane <-> step (64 consecutive steps per pass), four 8-byte loads per sample
//                  from global memory.  This is the production gather.
//   B  slab-staged VERDICT's variant: the 4 rays of a block stage the lines of their 64-step slab into LDS with
//                  coalesced 16-byte loads (bounding box of the slab in line units), then interpolate from LDS.
//                  A slab whose box exceeds the staging buffer falls back to A for that slab.
//   C  patch-staged a block takes 32 adjacent rays x 32 consecutive steps (the scatter kernel's patch: an about
//                  square footprint), stages the bounding box of its lines, interpolates from LDS and stores 4
//                  consecutive steps per thread.
// All three compute bit-identical values (checked).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e = (x);                                                               \
        if (e != hipSuccess) {                                                            \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                 \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int N = 256, P = 32, R = 256, S = 512;
constexpr int NB = N / 4;                   // lines along dim 0 / dim 1
constexpr unsigned SXB = NB * N * 128u;     // byte stride of a line row (dim 0 / 4)
constexpr unsigned SYB = N * 128u;          // ... of a line column (dim 1 / 4)

struct Ax {
    int i0, i1;
    float t;
};
__device__ __forceinline__ Ax axis(float p, int dim)
{
    const float hi = (float)(dim - 1);
    float pc = p;
    if (!(pc > 0.f)) pc = 0.f;
    pc = fminf(pc, hi);
    const float f = floorf(pc);
    Ax a;
    a.i0 = (int)f;
    a.t = pc - f;
    a.i1 = min(a.i0 + 1, dim - 1);
    return a;
}
__device__ __forceinline__ unsigned px(int x) { return __umul24((unsigned)x >> 2, SXB) + (((unsigned)x & 3u) << 5); }
__device__ __forceinline__ unsigned py(int y) { return __umul24((unsigned)y >> 2, SYB) + (((unsigned)y & 3u) << 3); }
__device__ __forceinline__ float2 ldg2(const float *b, unsigned off)
{
    return *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(b) + (size_t)off);
}
__device__ __forceinline__ float lerp3(float2 q00, float2 q01, float2 q10, float2 q11, float tx, float ty, float tz)
{
    const float c00 = fmaf(tz, q00.y - q00.x, q00.x), c01 = fmaf(tz, q01.y - q01.x, q01.x);
    const float c10 = fmaf(tz, q10.y - q10.x, q10.x), c11 = fmaf(tz, q11.y - q11.x, q11.x);
    const float a0 = fmaf(ty, c01 - c00, c00), a1 = fmaf(ty, c11 - c10, c10);
    return fmaf(tx, a1 - a0, a0);
}

// ---- A: direct gather, ray-major ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_direct(const float *__restrict__ vol, const float *__restrict__ src,
                                                const float *__restrict__ dirs, float *__restrict__ out)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pose = w / R;
    const float s0 = src[pose * 3], s1 = src[pose * 3 + 1], s2 = src[pose * 3 + 2];
    const float d0 = dirs[w * 3], d1 = dirs[w * 3 + 1];
    const Ax c = axis(s2, N);
    const unsigned zoff = (unsigned)c.i0 << 7;
    float v[S / 64];
#pragma unroll
    for (int j = 0; j < S / 64; ++j) {
        const float kf = (float)(j * 64 + lane);
        const Ax a = axis(s0 + kf * d0, N), b = axis(s1 + kf * d1, N);
        const unsigned x0 = px(a.i0), x1 = px(a.i1), y0 = py(b.i0) + zoff, y1 = py(b.i1) + zoff;
        v[j] = lerp3(ldg2(vol, x0 + y0), ldg2(vol, x0 + y1), ldg2(vol, x1 + y0), ldg2(vol, x1 + y1), a.t, b.t, c.t);
    }
#pragma unroll
    for (int j = 0; j < S / 64; ++j) out[(size_t)w * S + j * 64 + lane] = v[j];
}

// ---- block bounding box of (line x, line y) over 256 threads ----------------------------------------------------
template <bool IS_MIN>
__device__ __forceinline__ int wave_minmax(int v)
{
    constexpr int ident = IS_MIN ? 0x7fffffff : (int)0x80000000;
#define STEP(ctrl, rmask)                                                              \
    {                                                                                  \
        int o = __builtin_amdgcn_update_dpp(ident, v, ctrl, rmask, 0xf, false);        \
        v = IS_MIN ? min(v, o) : max(v, o);                                            \
    }
    STEP(0x111, 0xf) STEP(0x112, 0xf) STEP(0x114, 0xf) STEP(0x118, 0xf) STEP(0x142, 0xa) STEP(0x143, 0xc)
#undef STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// ---- B: the 4 rays of a block stage their 64-step slab ------------------------------------------------------------
constexpr int kSlabLines = 96; // 12 KiB staging buffer
__global__ __launch_bounds__(256) void k_slab(const float *__restrict__ vol, const float *__restrict__ src,
                                              const float *__restrict__ dirs, float *__restrict__ out, int *fallbacks)
{
    __shared__ __attribute__((aligned(16))) float stage[kSlabLines * 32];
    __shared__ int box[4][4];
    const int wib = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const int w = blockIdx.x * 4 + wib;
    const int pose = w / R;
    const float s0 = src[pose * 3], s1 = src[pose * 3 + 1], s2 = src[pose * 3 + 2];
    const float d0 = dirs[w * 3], d1 = dirs[w * 3 + 1];
    const Ax c = axis(s2, N);
    const unsigned zoff = (unsigned)c.i0 << 7;
    float v[S / 64];
#pragma unroll
    for (int j = 0; j < S / 64; ++j) {
        const float kf = (float)(j * 64 + lane);
        const Ax a = axis(s0 + kf * d0, N), b = axis(s1 + kf * d1, N);
        // bounding box of the slab in line units (a.i1 >= a.i0)
        const int lx0 = wave_minmax<true>(a.i0 >> 2), lx1 = wave_minmax<false>(a.i1 >> 2);
        const int ly0 = wave_minmax<true>(b.i0 >> 2), ly1 = wave_minmax<false>(b.i1 >> 2);
        if (lane == 0) {
            box[wib][0] = lx0; box[wib][1] = lx1; box[wib][2] = ly0; box[wib][3] = ly1;
        }
        __syncthreads();
        int bx0 = box[0][0], bx1 = box[0][1], by0 = box[0][2], by1 = box[0][3];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            bx0 = min(bx0, box[q][0]); bx1 = max(bx1, box[q][1]); by0 = min(by0, box[q][2]); by1 = max(by1, box[q][3]);
        }
        bx0 = __builtin_amdgcn_readfirstlane(bx0); bx1 = __builtin_amdgcn_readfirstlane(bx1);
        by0 = __builtin_amdgcn_readfirstlane(by0); by1 = __builtin_amdgcn_readfirstlane(by1);
        const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1, nl = bw * bh;
        if (nl <= kSlabLines) {
            // stage: 8 threads per line, 16 bytes each: coalesced 128-B reads
            const float rbh = 1.f / (float)bh;
            for (int l = tid >> 3; l < nl; l += 32) {
                const int li = (int)(((float)l + 0.5f) * rbh), lj = l - li * bh;
                const unsigned g = (unsigned)(bx0 + li) * SXB + (unsigned)(by0 + lj) * SYB + zoff + (unsigned)(tid & 7) * 16u;
                const float4 t = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(vol) + (size_t)g);
                *reinterpret_cast<float4 *>(&stage[l * 32 + (tid & 7) * 4]) = t;
            }
            __syncthreads();
            auto lds2 = [&](int x, int y) -> float2 {
                const int l = ((x >> 2) - bx0) * bh + ((y >> 2) - by0);
                return *reinterpret_cast<const float2 *>(&stage[l * 32 + ((x & 3) << 3) + ((y & 3) << 1)]);
            };
            v[j] = lerp3(lds2(a.i0, b.i0), lds2(a.i0, b.i1), lds2(a.i1, b.i0), lds2(a.i1, b.i1), a.t, b.t, c.t);
        } else {
            if (tid == 0 && fallbacks) atomicAdd(fallbacks, 1);
            const unsigned x0 = px(a.i0), x1 = px(a.i1), y0 = py(b.i0) + zoff, y1 = py(b.i1) + zoff;
            v[j] = lerp3(ldg2(vol, x0 + y0), ldg2(vol, x0 + y1), ldg2(vol, x1 + y0), ldg2(vol, x1 + y1), a.t, b.t, c.t);
        }
        __syncthreads(); // the staging buffer and the boxes are rewritten by the next slab
    }
#pragma unroll
    for (int j = 0; j < S / 64; ++j) out[(size_t)w * S + j * 64 + lane] = v[j];
}

// ---- C: a block stages a patch of 32 rays x 32 steps ------------------------------------------------------------
constexpr int kPatchLines = 384; // 48 KiB staging buffer (3 blocks per CU)
__global__ __launch_bounds__(256) void k_patch(const float *__restrict__ vol, const float *__restrict__ src,
                                               const float *__restrict__ dirs, float *__restrict__ out, int *fallbacks)
{
    __shared__ __attribute__((aligned(16))) float stage[kPatchLines * 32];
    __shared__ int box[4][4];
    const int tid = threadIdx.x, wib = tid >> 6, lane = tid & 63;
    // patch -> (step group slowest, pose, ray group)
    constexpr int RG = R / 32, SG = S / 32;
    const int sg = blockIdx.x / (P * RG), rem = blockIdx.x % (P * RG);
    const int pose = rem / RG, rg = rem % RG;
    const int ray = rg * 32 + (tid >> 3), n0 = sg * 32 + (tid & 7) * 4;
    const int w = pose * R + ray;
    const float s0 = src[pose * 3], s1 = src[pose * 3 + 1], s2 = src[pose * 3 + 2];
    const float d0 = dirs[w * 3], d1 = dirs[w * 3 + 1];
    const Ax c = axis(s2, N);
    const unsigned zoff = (unsigned)c.i0 << 7;
    Ax a[4], b[4];
    int lx0 = 0x7fffffff, lx1 = -1, ly0 = 0x7fffffff, ly1 = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float kf = (float)(n0 + q);
        a[q] = axis(s0 + kf * d0, N);
        b[q] = axis(s1 + kf * d1, N);
        lx0 = min(lx0, a[q].i0 >> 2); lx1 = max(lx1, a[q].i1 >> 2);
        ly0 = min(ly0, b[q].i0 >> 2); ly1 = max(ly1, b[q].i1 >> 2);
    }
    lx0 = wave_minmax<true>(lx0); lx1 = wave_minmax<false>(lx1);
    ly0 = wave_minmax<true>(ly0); ly1 = wave_minmax<false>(ly1);
    if (lane == 0) {
        box[wib][0] = lx0; box[wib][1] = lx1; box[wib][2] = ly0; box[wib][3] = ly1;
    }
    __syncthreads();
    int bx0 = box[0][0], bx1 = box[0][1], by0 = box[0][2], by1 = box[0][3];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        bx0 = min(bx0, box[q][0]); bx1 = max(bx1, box[q][1]); by0 = min(by0, box[q][2]); by1 = max(by1, box[q][3]);
    }
    bx0 = __builtin_amdgcn_readfirstlane(bx0); bx1 = __builtin_amdgcn_readfirstlane(bx1);
    by0 = __builtin_amdgcn_readfirstlane(by0); by1 = __builtin_amdgcn_readfirstlane(by1);
    const int bw = bx1 - bx0 + 1, bh = by1 - by0 + 1, nl = bw * bh;
    float v[4];
    if (nl <= kPatchLines) {
        const float rbh = 1.f / (float)bh;
        for (int l = tid >> 3; l < nl; l += 32) {
            const int li = (int)(((float)l + 0.5f) * rbh), lj = l - li * bh;
            const unsigned g = (unsigned)(bx0 + li) * SXB + (unsigned)(by0 + lj) * SYB + zoff + (unsigned)(tid & 7) * 16u;
            const float4 t = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(vol) + (size_t)g);
            *reinterpret_cast<float4 *>(&stage[l * 32 + (tid & 7) * 4]) = t;
        }
        __syncthreads();
        auto lds2 = [&](int x, int y) -> float2 {
            const int l = ((x >> 2) - bx0) * bh + ((y >> 2) - by0);
            return *reinterpret_cast<const float2 *>(&stage[l * 32 + ((x & 3) << 3) + ((y & 3) << 1)]);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q)
            v[q] = lerp3(lds2(a[q].i0, b[q].i0), lds2(a[q].i0, b[q].i1), lds2(a[q].i1, b[q].i0), lds2(a[q].i1, b[q].i1),
                         a[q].t, b[q].t, c.t);
    } else {
        if (tid == 0 && fallbacks) atomicAdd(fallbacks, 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned x0 = px(a[q].i0), x1 = px(a[q].i1), y0 = py(b[q].i0) + zoff, y1 = py(b[q].i1) + zoff;
            v[q] = lerp3(ldg2(vol, x0 + y0), ldg2(vol, x0 + y1), ldg2(vol, x1 + y0), ldg2(vol, x1 + y1), a[q].t, b[q].t, c.t);
        }
    }
    *reinterpret_cast<float4 *>(&out[(size_t)w * S + n0]) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- D: the patch mapping WITHOUT staging (direct gathers): separates the mapping's effect from the staging's ----
__global__ __launch_bounds__(256) void k_patch_direct(const float *__restrict__ vol, const float *__restrict__ src,
                                                      const float *__restrict__ dirs, float *__restrict__ out)
{
    const int tid = threadIdx.x;
    constexpr int RG = R / 32;
    const int sg = blockIdx.x / (P * RG), rem = blockIdx.x % (P * RG);
    const int pose = rem / RG, rg = rem % RG;
    const int ray = rg * 32 + (tid >> 3), n0 = sg * 32 + (tid & 7) * 4;
    const int w = pose * R + ray;
    const float s0 = src[pose * 3], s1 = src[pose * 3 + 1], s2 = src[pose * 3 + 2];
    const float d0 = dirs[w * 3], d1 = dirs[w * 3 + 1];
    const Ax c = axis(s2, N);
    const unsigned zoff = (unsigned)c.i0 << 7;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float kf = (float)(n0 + q);
        const Ax a = axis(s0 + kf * d0, N), b = axis(s1 + kf * d1, N);
        const unsigned x0 = px(a.i0), x1 = px(a.i1), y0 = py(b.i0) + zoff, y1 = py(b.i1) + zoff;
        v[q] = lerp3(ldg2(vol, x0 + y0), ldg2(vol, x0 + y1), ldg2(vol, x1 + y0), ldg2(vol, x1 + y1), a.t, b.t, c.t);
    }
    *reinterpret_cast<float4 *>(&out[(size_t)w * S + n0]) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- E: direct gather with 16-byte ROW loads: (x, y0) and (x, y0 + 1) are adjacent 8-byte entries of a line unless
// y0 & 3 == 3; those lanes (a quarter) fetch the second column with an extra, exec-masked 8-byte load.  Is a masked
// load cheaper for the texture addresser than a full one?
__device__ __forceinline__ float4 ldg4(const float *b, unsigned off)
{
    return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(b) + (size_t)off);
}
template <bool FIX>
__global__ __launch_bounds__(256) void k_row16(const float *__restrict__ vol, const float *__restrict__ src,
                                               const float *__restrict__ dirs, float *__restrict__ out)
{
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int pose = w / R;
    const float s0 = src[pose * 3], s1 = src[pose * 3 + 1], s2 = src[pose * 3 + 2];
    const float d0 = dirs[w * 3], d1 = dirs[w * 3 + 1];
    const Ax c = axis(s2, N);
    const unsigned zoff = (unsigned)c.i0 << 7;
    float v[S / 64];
    float4 qa[S / 64], qb[S / 64];
    float2 fa[S / 64], fb[S / 64];
    Ax aa[S / 64], bb[S / 64];
#pragma unroll
    for (int j = 0; j < S / 64; ++j) {
        const float kf = (float)(j * 64 + lane);
        const Ax a = axis(s0 + kf * d0, N), b = axis(s1 + kf * d1, N);
        aa[j] = a; bb[j] = b;
        const unsigned x0 = px(a.i0), x1 = px(a.i1), y0 = py(b.i0) + zoff, y1 = py(b.i1) + zoff;
        qa[j] = ldg4(vol, x0 + y0);
        qb[j] = ldg4(vol, x1 + y0);
        fa[j] = make_float2(0.f, 0.f); fb[j] = fa[j];
        if (FIX && ((b.i0 & 3) == 3)) {
            fa[j] = ldg2(vol, x0 + y1);
            fb[j] = ldg2(vol, x1 + y1);
        }
    }
#pragma unroll
    for (int j = 0; j < S / 64; ++j) {
        const bool fix = FIX && ((bb[j].i0 & 3) == 3);
        const float2 q00 = make_float2(qa[j].x, qa[j].y), q10 = make_float2(qb[j].x, qb[j].y);
        const float2 q01 = fix ? fa[j] : make_float2(qa[j].z, qa[j].w), q11 = fix ? fb[j] : make_float2(qb[j].z, qb[j].w);
        v[j] = lerp3(q00, q01, q10, q11, aa[j].t, bb[j].t, c.t);
    }
#pragma unroll
    for (int j = 0; j < S / 64; ++j) out[(size_t)w * S + j * 64 + lane] = v[j];
}

template <typename F>
static float time_us(F &&launch, int iters = 50)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main()
{
    // volume (paired layout) with a smooth deterministic pattern; poses on the bench's ring
    const size_t nlines = (size_t)NB * NB * N;
    std::vector<float> hv(nlines * 32);
    for (int x = 0; x < N; ++x)
        for (int y = 0; y < N; ++y)
            for (int z = 0; z < N; ++z) {
                auto val = [&](int zz) { return 1.5e6f + 2e4f * sinf(0.37f * x) * sinf(0.29f * y) * sinf(0.31f * zz); };
                const size_t line = ((size_t)(x >> 2) * NB + (y >> 2)) * N + z;
                float *p = &hv[line * 32 + ((x & 3) << 3) + ((y & 3) << 1)];
                p[0] = val(z);
                p[1] = val(z + 1 < N ? z + 1 : N - 1);
            }
    std::vector<float> hs(P * 3), hd((size_t)P * R * 3);
    for (int p = 0; p < P; ++p) {
        const double phi = 2.0 * M_PI * p / P;
        hs[p * 3] = (float)(0.5 * N + 0.30 * N * cos(phi));
        hs[p * 3 + 1] = (float)(0.5 * N + 0.30 * N * sin(phi));
        hs[p * 3 + 2] = (float)(0.5 * N + 0.05 * N * sin(3 * phi));
        const double dx = -cos(phi), dy = -sin(phi), ox = -dy, oy = dx, th = M_PI / 3;
        for (int r = 0; r < R; ++r) {
            const double a = -th / 2 + th * r / (R - 1);
            hd[((size_t)p * R + r) * 3] = (float)(cos(a) * dx + sin(a) * ox);
            hd[((size_t)p * R + r) * 3 + 1] = (float)(cos(a) * dy + sin(a) * oy);
            hd[((size_t)p * R + r) * 3 + 2] = 0.f;
        }
    }
    float *vol, *src, *dirs, *oa, *ob;
    int *fb;
    const size_t nout = (size_t)P * R * S;
    CHECK(hipMalloc(&vol, hv.size() * 4 + 64 /* E/F read 8 bytes past a line's last entry */)); CHECK(hipMalloc(&src, hs.size() * 4)); CHECK(hipMalloc(&dirs, hd.size() * 4));
    CHECK(hipMalloc(&oa, nout * 4)); CHECK(hipMalloc(&ob, nout * 4)); CHECK(hipMalloc(&fb, 4));
    CHECK(hipMemcpy(vol, hv.data(), hv.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(src, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dirs, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ra(nout), rb(nout);
    auto same = [&](const char *name) {
        CHECK(hipMemcpy(rb.data(), ob, nout * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < nout; ++i) bad += ra[i] != rb[i];
        printf("   %s vs direct: %zu of %zu samples differ\n", name, bad, nout);
    };
    const int nb_ray = P * R / 4, nb_patch = P * (R / 32) * (S / 32);
    hipLaunchKernelGGL(k_direct, dim3(nb_ray), dim3(256), 0, 0, vol, src, dirs, oa);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(ra.data(), oa, nout * 4, hipMemcpyDeviceToHost));
    printf("lds_stage_probe: %d poses x %d rays x %d steps = %zu samples, %d^3 paired volume\n", P, R, S, nout, N);
    float t = time_us([&] { hipLaunchKernelGGL(k_direct, dim3(nb_ray), dim3(256), 0, 0, vol, src, dirs, oa); });
    printf("A direct gather, ray-major            %7.1f us\n", t);
    CHECK(hipMemset(fb, 0, 4));
    hipLaunchKernelGGL(k_slab, dim3(nb_ray), dim3(256), 0, 0, vol, src, dirs, ob, fb);
    CHECK(hipDeviceSynchronize());
    int nfb = 0;
    CHECK(hipMemcpy(&nfb, fb, 4, hipMemcpyDeviceToHost));
    t = time_us([&] { hipLaunchKernelGGL(k_slab, dim3(nb_ray), dim3(256), 0, 0, vol, src, dirs, ob, (int *)nullptr); });
    printf("B slab-staged (4 rays x 64 steps)     %7.1f us   (%d of %d slabs fell back to direct loads)\n", t, nfb, nb_ray * (S / 64));
    same("B");
    CHECK(hipMemset(fb, 0, 4));
    hipLaunchKernelGGL(k_patch, dim3(nb_patch), dim3(256), 0, 0, vol, src, dirs, ob, fb);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMemcpy(&nfb, fb, 4, hipMemcpyDeviceToHost));
    t = time_us([&] { hipLaunchKernelGGL(k_patch, dim3(nb_patch), dim3(256), 0, 0, vol, src, dirs, ob, (int *)nullptr); });
    printf("C patch-staged (32 rays x 32 steps)   %7.1f us   (%d of %d patches fell back)\n", t, nfb, nb_patch);
    same("C");
    t = time_us([&] { hipLaunchKernelGGL(k_patch_direct, dim3(nb_patch), dim3(256), 0, 0, vol, src, dirs, ob); });
    printf("D patch mapping, direct gathers       %7.1f us\n", t);
    same("D");
    t = time_us([&] { hipLaunchKernelGGL(k_row16<true>, dim3(nb_ray), dim3(256), 0, 0, vol, src, dirs, ob); });
    printf("E 16-byte row loads + masked fix-ups  %7.1f us\n", t);
    same("E");
    t = time_us([&] { hipLaunchKernelGGL(k_row16<false>, dim3(nb_ray), dim3(256), 0, 0, vol, src, dirs, ob); });
    printf("F 16-byte row loads only (wrong at y0 & 3 == 3: the floor of E)  %7.1f us\n", t);
    return 0;
}
