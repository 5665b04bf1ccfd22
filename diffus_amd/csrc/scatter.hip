// scatter.hip -- volume-gradient scatter (LDS-privatised patches), sparse gradient flush, layout conversions
#include "diffus_host.hpp"

namespace {

// ----------------------------------------------------------------------------
// VOLUME SCATTER.  gvol += sum over samples of zbar * (interpolation weights).
// Naive per-sample global float atomics run ~10x below even the scattered-atomic
// rate because every fan hammers the few hundred voxels around its apex
// (measured: 17.5 ms for 33 M atomics at config 3).  Instead a block takes a PATCH
// of kPatchRays adjacent rays x kPatchSteps consecutive steps of one pose, whose
// footprint is a small box of voxels; it accumulates the patch into an LDS tile
// covering that box -- in int32 fixed point with ds_add_u32, see the kernel -- and
// flushes each touched voxel ONCE with global_atomic_add_f32.  In the bricked layout
// the tile is a box of whole bricks, so the flush is made of 128-B contiguous atomic
// runs (the full-rate shape of that atomic).  A patch whose box exceeds the tile goes
// through it as 2 or 4 groups of waves, each with its own box.
// Three paths, tried in this order by every block (bricked gradients):
//   scatter_patch_planar  no ray of the patch moves along dim 2 (every fan of the reference): a 2-D tile of doubles;
//   scatter_patch_slab    the rays lie in one plane through their source (a tilted fan): a height-field tile over that plane,
//                         doubles, one pass per class of samples (inside the volume / clamped onto a face); only in the launch
//                         without the DIFFUS_FANS_PLANAR hint (a larger tile, 4 blocks per CU);
//   the 3-D brick tile in the kernel body: anything else (rays that are not coplanar), and canonical gradients.
#ifdef DIFFUS_STAMP // diagnostic build only (tools/): per-block phase timestamps of the scatter kernel
__device__ unsigned long long *g_stamps = nullptr;
#define STAMP(i)                                                                          \
    do {                                                                                  \
        if (threadIdx.x == 0 && g_stamps) g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); \
    } while (0)
#define STAMPV(i, v)                                                                     \
    do {                                                                                  \
        if (threadIdx.x == 0 && g_stamps) g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = (unsigned long long)(v); \
    } while (0)
#define STAMP_NOW() __builtin_readcyclecounter()
#else
#define STAMP(i) ((void)0)
#define STAMPV(i, v) ((void)0)
#define STAMP_NOW() 0ull
#endif

// tile units: voxels (canonical) or bricks (bricked)
template <int LAYOUT>
__device__ __forceinline__ int tile_unit(int v, int axis)
{
    if (LAYOUT == DIFFUS_CANONICAL) return v;
    return axis == 2 ? (v >> 1) : (v >> 2);
}

// threads per scatter block.  The patch stays kPatchRays x kPatchSteps = 1024 samples and the tile 24 KiB (6 blocks per CU); 512 threads
// (2 samples each, twice the waves per CU) measured 70 us against 58 for 256: the barriers over 8 waves cost more than
// the extra waves hide.
#ifndef DIFFUS_SCATTER_THREADS
#define DIFFUS_SCATTER_THREADS 256
#endif
#ifndef DIFFUS_SC_PATCH_STEPS
#define DIFFUS_SC_PATCH_STEPS DIFFUS_PATCH_STEPS
#endif
#ifndef DIFFUS_SC_PATCH_RAYS
#define DIFFUS_SC_PATCH_RAYS DIFFUS_PATCH_RAYS
#endif
#ifndef DIFFUS_SC_MIN_BLOCKS
#define DIFFUS_SC_MIN_BLOCKS 6
#endif
// stage probe (tools/): -DDIFFUS_SC_EXIT=n makes the planar path return after stage n with its values forced live
#ifdef DIFFUS_SC_EXIT
#define SC_EXIT(n)                                                                                                      \
    if (DIFFUS_SC_EXIT == (n)) {                                                                                        \
        _Pragma("unroll") for (int q_ = 0; q_ < kSPT; ++q_)                                                             \
            asm volatile("" ::"v"(zb[q_]), "v"(tx[q_]), "v"(ty[q_]), "v"(x0[q_]), "v"(y0[q_]), "v"(x1[q_]), "v"(y1[q_])); \
        return true;                                                                                                    \
    }
#else
#define SC_EXIT(n) ((void)0)
#endif
constexpr int kScRays = DIFFUS_SC_PATCH_RAYS, kScSteps = DIFFUS_SC_PATCH_STEPS;
constexpr int kSB = DIFFUS_SCATTER_THREADS, kSW = kSB / kWave, kSPT = kScRays * kScSteps / kSB;
// Tile capacities in 32-bit entries.  kTileCap (24 KiB, 6 blocks per CU): the launch for fans the caller KNOWS to be planar
// in dim 2 (DIFFUS_FANS_PLANAR) and for canonical gradients.  kSlabCap (36 KiB, 4 blocks per CU): the launch that also
// carries the slab path below for fans that leave the slice -- their tile holds several layers per column.
#ifndef DIFFUS_SLAB_CAP
#define DIFFUS_SLAB_CAP 9216
#endif
#ifndef DIFFUS_SLAB_MIN_BLOCKS
#define DIFFUS_SLAB_MIN_BLOCKS 4
#endif
constexpr int kSlabCap = DIFFUS_SLAB_CAP;
#ifndef DIFFUS_SC_ROW_PAD
#define DIFFUS_SC_ROW_PAD 1
#endif
constexpr unsigned kRowPad = DIFFUS_SC_ROW_PAD; // planar tile: padding entries per row (bank spread)

// ---- PLANAR patches (bricked gradient) -----------------------------------------------------------------------------
// No ray of the patch moves along dim 2 -- every fan of the reference (src/cone.py:258) -- so all its samples share ONE
// dim-2 cell (iz0, iz1, tz).  The tile then holds the 2-D footprint only: a plain row-major array of DOUBLES over the
// brick-aligned (dim 0, dim 1) box of the patch, 4 adds per sample instead of 8, and the two depth weights are applied
// once per entry in the flush.
//
// Why doubles: ds_add_f32 costs ~194 cycles per wave-instruction on gfx950 whatever the addresses, ds_add_u32 5-15,
// ds_add_u64 12-56 -- and ds_add_f64 21-89 (tools/lds_atomic_bench.hip, round 3): the LDS float64 atomic is a native
// full-rate path where the float32 one is not.  Rounds 1-2 therefore accumulated in 64-bit FIXED POINT, which cost a
// patch-wide sum of |zbar| (a DPP reduction, an LDS record per wave, scalar bookkeeping) for the common scale and ~10
// VALU instructions per corner for the float -> (mantissa << shift) conversion: 40 of a sample's ~215 instructions in a
// kernel that runs at its VALU issue rate (PMC: 870 VALU instructions per wave, pipes 68 % busy).  A double takes the
// contribution as it is (one v_cvt_f64_f32); accumulation error 2^-53 relative per add instead of a 2^-61 quantum of
// the patch's sum -- both far below the float32 atomics of the flush.
//
// Returns false -- nothing added, tile clear -- when some ray of the block is not planar (the caller then runs the
// general 3-D path); true when the patch is done.
template <int SAMPLER, int PM, int CAP>
__device__ __forceinline__ bool scatter_patch_planar(const Args &A, double *tile, int (*s_box)[4], int *s_planar, int *s_live, const Pose &ps,
                                                     const float *rows, unsigned row_off, bool ray_ok, int nbase, int tid)
{
    // rows: zbar at the block's first ray (block-uniform, a scalar base); row_off: this thread's ray and first step in
    // BYTES from there (32 bits: no 64-bit multiply per lane)
    const int wib = tid >> 6;
    constexpr int kCapD = CAP / 2; // 64-bit entries in the tile
    // ---- loads first: a thread's kSPT consecutive zbar values (one 16-byte load when the row allows it) ...
    float zb[kSPT];
    {
        const char *rb = reinterpret_cast<const char *>(rows);
#ifdef DIFFUS_ABLATE_SC_LOAD // timing probes (tools/): the zbar values are made up, nothing is loaded
        if (true) {
#pragma unroll
            for (int q = 0; q < kSPT; ++q) zb[q] = (ray_ok && nbase + q < A.N1) ? 1e-3f * (float)((tid + q) & 15) : 0.f;
        } else
#endif
        if (ray_ok && nbase + kSPT <= A.N1) {
            if constexpr (kSPT == 4) {
                const F4a4 t = *reinterpret_cast<const F4a4 *>(rb + (size_t)row_off);
                zb[0] = t.x; zb[1] = t.y; zb[2] = t.z; zb[3] = t.w;
            } else {
#pragma unroll
                for (int q = 0; q < kSPT; ++q) zb[q] = ldb_f32(rows, row_off + 4u * q);
            }
        } else {
#pragma unroll
            for (int q = 0; q < kSPT; ++q) zb[q] = (ray_ok && nbase + q < A.N1) ? ldb_f32(rows, row_off + 4u * q) : 0.f;
        }
    }
    // ... and the WHOLE tile is cleared while they are in flight (6 ds_write_b128 per thread)
    {
        int4 *t4 = reinterpret_cast<int4 *>(tile);
#pragma unroll
        for (int e = 0; e < CAP / 4 / kSB; ++e) t4[e * kSB + tid] = make_int4(0, 0, 0, 0);
        static_assert(CAP % (4 * kSB) == 0, "tile clear assumes whole int4 passes");
    }
    const bool ray_planar = (PM == 0 || ps.pmode != 2) ? (ps.df[2] == 0.f) : (ps.dd[2] == 0.0);
    const bool wave_planar = __ballot(ray_planar) == ~0ull;
    // ---- cells along dim 0 and dim 1 (the pose only: worked out while the loads are in flight)
    int x0[kSPT], x1[kSPT], y0[kSPT], y1[kSPT];
    float tx[kSPT], ty[kSPT];
#pragma unroll
    for (int q = 0; q < kSPT; ++q) {
        const float kf = (float)(A.start + nbase + q);
        const float p0 = ray_point_f<PM>(ps, 0, kf), p1 = ray_point_f<PM>(ps, 1, kf);
        if (SAMPLER == DIFFUS_NEAREST) {
            x0[q] = x1[q] = nearest_index(p0, A.G.d0);
            y0[q] = y1[q] = nearest_index(p1, A.G.d1);
            tx[q] = ty[q] = 0.f;
        } else {
            const Axis a = tri_axis(p0, A.G.d0), b = tri_axis(p1, A.G.d1);
            x0[q] = a.i0; x1[q] = a.i1; tx[q] = a.t;
            y0[q] = b.i0; y1[q] = b.i1; ty[q] = b.t;
        }
    }
    // ---- bounding box of the thread's samples.  A ray is a straight line and every step of the chain p -> clamp -> floor
    // is monotone in the step index, so the extremes sit at the first and the last of its consecutive samples.
    // The box takes every sample the patch HAS (ray < R, step < N1), whatever its zbar: nothing before the first barrier
    // waits for the loads (round 3 first boxed the samples with zbar != 0 only: every wave sat on its load before the
    // reduction, and then on the barrier for the slowest wave's).  Whether the patch has anything to add at all is
    // settled later, through one LDS flag.
    const bool has = ray_ok && nbase < A.N1;
    int bx[4];
    bx[0] = has ? min(x0[0], x0[kSPT - 1]) : 0x7fffffff;
    bx[1] = has ? max(x1[0], x1[kSPT - 1]) : -1;
    bx[2] = has ? min(y0[0], y0[kSPT - 1]) : 0x7fffffff;
    bx[3] = has ? max(y1[0], y1[kSPT - 1]) : -1;
    if (tid == 0) *s_live = 0;
    STAMP(1);
    SC_EXIT(1);
    bx[0] = wave_reduce_minmax<true>(bx[0]);
    bx[1] = wave_reduce_minmax<false>(bx[1]);
    bx[2] = wave_reduce_minmax<true>(bx[2]);
    bx[3] = wave_reduce_minmax<false>(bx[3]);
    if ((tid & 63) == 63) {
        s_planar[wib] = wave_planar;
#pragma unroll
        for (int a = 0; a < 4; ++a) s_box[wib][a] = bx[a];
    }
    __syncthreads(); // also: the tile is clear
    STAMP(2);
    SC_EXIT(2);
    // ---- block-uniform bookkeeping, on the scalar unit (readfirstlane): the boxes in BRICK units
    int wb[kSW][4];
    int all_planar = 1;
#pragma unroll
    for (int wv = 0; wv < kSW; ++wv) {
        all_planar &= s_planar[wv];
#pragma unroll
        for (int a = 0; a < 4; ++a) wb[wv][a] = __builtin_amdgcn_readfirstlane(s_box[wv][a]) >> 2; // 0x7fffffff stays huge, -1 stays -1
    }
    if (!__builtin_amdgcn_readfirstlane(all_planar)) return false;
    // box of the waves [w0, w0 + cnt) -> origin (bricks), extent (bricks); entries needed: 0 = nothing to add,
    // kCapD + 1 = does not fit
    auto box_of = [&](int w0, int cnt, int &l0, int &l1, int &b0, int &b1) -> int {
        int mn0 = 0x7fffffff, mx0 = -1, mn1 = 0x7fffffff, mx1 = -1;
#pragma unroll
        for (int wv = 0; wv < kSW; ++wv) {
            const bool in = wv >= w0 && wv < w0 + cnt;
            mn0 = in ? min(mn0, wb[wv][0]) : mn0; mx0 = in ? max(mx0, wb[wv][1]) : mx0;
            mn1 = in ? min(mn1, wb[wv][2]) : mn1; mx1 = in ? max(mx1, wb[wv][3]) : mx1;
        }
        l0 = mn0; l1 = mn1; b0 = mx0 - mn0 + 1; b1 = mx1 - mn1 + 1;
        if (mx0 < 0 || mx1 < 0) return 0;
        const unsigned e0 = (unsigned)min(b0, 0x3fff), e1 = (unsigned)min(b1, 0x3fff); // saturate: only "> kCapD" matters
        return (int)min(4u * e0 * (4u * e1 + kRowPad), (unsigned)kCapD + 1u); // rows of 4 e1 (+ pad) entries
    };
    static_assert(kSW == 4 || kSW == 8, "wave grouping below: 1, 2 or 4 groups of waves");
    int nsub = 1, l0, l1, b0, b1;
    int need = box_of(0, kSW, l0, l1, b0, b1);
    if (need > kCapD) {
        int t0, t1, t2, t3;
        nsub = 2;
        if (box_of(0, kSW / 2, t0, t1, t2, t3) > kCapD || box_of(kSW / 2, kSW / 2, t0, t1, t2, t3) > kCapD) nsub = 4;
    }
    // the patch's dim-2 cell: the same in every thread (p2 = source[2] for every sample)
    int iz0, iz1;
    float tz;
    {
        const float p2 = ray_point_f<PM>(ps, 2, 0.f);
        if (SAMPLER == DIFFUS_NEAREST) {
            iz0 = iz1 = nearest_index(p2, A.G.d2);
            tz = 0.f;
        } else {
            const Axis c = tri_axis(p2, A.G.d2);
            iz0 = c.i0; iz1 = c.i1; tz = c.t;
        }
        iz0 = __builtin_amdgcn_readfirstlane(iz0); iz1 = __builtin_amdgcn_readfirstlane(iz1);
        tz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(tz)));
    }
    const int wpg = kSW / nsub;
    // ---- now the zbar values are needed
    {
        unsigned nz = 0;
#pragma unroll
        for (int q = 0; q < kSPT; ++q) {
            if (!finitef(zb[q])) zb[q] = 0.f;
            nz |= __float_as_uint(zb[q]) & 0x7fffffffu;
        }
        if (__builtin_amdgcn_ballot_w64(nz != 0u) != 0ull && (tid & 63) == 63) *s_live = 1;
    }
#pragma unroll 1
    for (int sp = 0; sp < nsub; ++sp) { // block-uniform trip count and branches
        if (nsub > 1) need = box_of(sp * wpg, wpg, l0, l1, b0, b1);
        if (need == 0) continue; // nothing to add in this group
        const bool mine = (wib / wpg) == sp;
        // keep the per-sample values INSIDE the trip (hoisted out of this almost always single-trip loop they cost registers)
#pragma unroll
        for (int q = 0; q < kSPT; ++q)
            asm volatile("" : "+v"(zb[q]), "+v"(tx[q]), "+v"(ty[q]), "+v"(x0[q]), "+v"(y0[q]), "+v"(x1[q]), "+v"(y1[q]));
        if (need > kCapD) { // a single wave's strip does not fit (never seen with unit steps): direct atomics
            if (mine) {
#pragma unroll
                for (int q = 0; q < kSPT; ++q)
                    if (zb[q] != 0.f) {
                        Cell c;
                        c.i0[0] = x0[q]; c.i1[0] = x1[q]; c.t[0] = tx[q];
                        c.i0[1] = y0[q]; c.i1[1] = y1[q]; c.t[1] = ty[q];
                        c.i0[2] = iz0; c.i1[2] = iz1; c.t[2] = tz;
                        for_each_corner<SAMPLER>(c, zb[q], [&](int i, int j, int k, float v) {
                            if (v != 0.f) {
                                unsigned g = vox_off<DIFFUS_BRICKED>(A.G, i, j, k);
                                atomicAdd(A.gvol + g, v);
                                if (A.gtouched) A.gtouched[g >> 5] = 1;
                            }
                        });
                    }
            }
            continue;
        }
        STAMP(3);
    SC_EXIT(3);
        // tile entry of voxel (x, y) = (x - 4 l0) * BY + (y - 4 l1), BY = 4 b1 voxels per row + 1 of padding: an ODD
        // row stride.  A thread's neighbours in the wave sit 4 steps further along the ray; for a ray that runs along
        // dim 0 that is 4 rows, and 4 rows of 4 b1 doubles are a multiple of the 64 banks whenever b1 is even: the 8
        // step groups of a ray on one bank pair (PMC: 29 % of the LDS pipe's active cycles were bank conflicts).
        const int BY = 4 * b1 + kRowPad;
        const int org = -(4 * l0) * BY - 4 * l1;
        // Same-address lanes of one LDS atomic are served one after the other (ds_add_f64: 21 cycles per wave-instruction
        // with distinct addresses, +23 per duplicate).  The worst case is also a common one: a ray that has left the
        // volume through an edge of the slice is clamped onto ONE voxel for the rest of its steps, all the rays around
        // it onto the same one -- 64 lanes, one address (6.5 % of the adds of config 3 and, at ~1400 cycles each, half
        // of the LDS pipe's time).  So each sample first asks whether the wave's lanes all sit in the same cell (two
        // readfirstlanes and compares); if so the four contributions are summed over the wave (DPP, in double) and ONE
        // lane adds them.
        // byte offsets throughout: corner address = (row base) + 8 y in ONE v_lshl_add_u32
        const int BY8 = BY * 8, org8 = org * 8;
        char *tile_c = reinterpret_cast<char *>(tile);
        auto add_at = [&](int byte_off, double v) { atomicAdd(reinterpret_cast<double *>(tile_c + byte_off), v); };
        if (mine) { // wave-uniform (a wave belongs to one group)
#pragma unroll
        for (int q = 0; q < kSPT; ++q) {
            const bool on = zb[q] != 0.f; // a zero zbar makes every contribution below an exact zero: no select needed
            const int r0 = __mul24(x0[q], BY8) + org8, r1 = __mul24(x1[q], BY8) + org8; // v_mad_i32_i24: full rate (indices < 2^20)
            const int e00 = r0 + 8 * y0[q], e11 = r1 + 8 * y1[q];
            float c00, c01 = 0.f, c10 = 0.f, c11 = 0.f;
            if constexpr (SAMPLER == DIFFUS_NEAREST) {
                c00 = zb[q];
            } else {
                const float wa1 = tx[q], wa0 = 1.f - wa1, wb1 = ty[q], wb0 = 1.f - wb1;
                const float s0 = zb[q] * wa0, s1 = zb[q] * wa1;
                c00 = s0 * wb0; c01 = s0 * wb1; c10 = s1 * wb0; c11 = s1 * wb1;
            }
#ifdef DIFFUS_ABLATE_SC_ADD
            asm volatile("" :: "v"(c00), "v"(c01), "v"(c10), "v"(c11), "v"(e00), "v"(e11));
            continue;
#endif
            const unsigned long long act = __builtin_amdgcn_ballot_w64(on); // (HIP's __ballot goes through an int: 2 VALU more)
            if (act == 0ull) continue; // wave-uniform
            const int lead = __builtin_ctzll(act);
            const int f00 = __builtin_amdgcn_readlane(e00, lead), f11 = __builtin_amdgcn_readlane(e11, lead);
            // (one ballot per compare: a ballot of the conjunction goes through a v_cndmask and a second compare)
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(e00 == f00) & __builtin_amdgcn_ballot_w64(e11 == f11);
            const bool same = (eq & act) == act;
            if (same && __builtin_popcountll(act) > 4) { // wave-uniform: one cell for every live lane
                const int f01 = __builtin_amdgcn_readlane(r0 + 8 * y1[q], lead), f10 = __builtin_amdgcn_readlane(r1 + 8 * y0[q], lead);
                const double t00 = wave_sum_to_lane63((double)c00);
                double t01 = 0.0, t10 = 0.0, t11 = 0.0;
                if constexpr (SAMPLER != DIFFUS_NEAREST) {
                    t01 = wave_sum_to_lane63((double)c01);
                    t10 = wave_sum_to_lane63((double)c10);
                    t11 = wave_sum_to_lane63((double)c11);
                }
                if ((tid & 63) == 63) {
                    if (t00 != 0.0) add_at(f00, t00);
                    if (t01 != 0.0) add_at(f01, t01);
                    if (t10 != 0.0) add_at(f10, t10);
                    if (t11 != 0.0) add_at(f11, t11);
                }
            } else {
                // clamped samples (outside the volume: more than half of a typical fan) have zero weights on half or
                // more of their corners: no LDS atomic is spent on a zero
                if (c00 != 0.f) add_at(e00, (double)c00);
                if constexpr (SAMPLER != DIFFUS_NEAREST) {
                    if (c01 != 0.f) add_at(r0 + 8 * y1[q], (double)c01);
                    if (c10 != 0.f) add_at(r1 + 8 * y0[q], (double)c10);
                    if (c11 != 0.f) add_at(e11, (double)c11);
                }
            }
            // one sample at a time: left alone the scheduler converts all 16 contributions to doubles first (32 VGPRs)
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        __syncthreads();
        STAMP(4);
    SC_EXIT(4);
        if (__builtin_amdgcn_readfirstlane(*s_live) == 0) continue; // no wave had a nonzero zbar: the tile is still clear (block-uniform)
        // flush: a half-wave = one brick column = the 32 floats (x & 3, y & 3, z) of its brick(s); two lanes share a
        // tile entry and apply the two depth weights.  z0 even: one brick, a contiguous 128-B atomic run.
        // A WAVE takes two adjacent columns (cj = 2 cp + h, h = lane >> 5) of one brick row ci, so everything that depends
        // on (ci, cp) is wave-uniform and stays on the scalar unit: the tile address is a scalar base + a per-lane
        // constant, the memory addresses a scalar 64-bit base + a per-lane 32-bit offset.  (Round 3 first had every
        // half-wave derive its own column from a flat index: a float division, two quarter-rate v_mul_lo_u32 and two
        // v_mad_u64_u32 per column per lane -- 24 VALU instructions per wave-atomic, a third of the kernel's VALU count.)
        {
            const int lane = tid & 63, o = lane & 31, h = lane >> 5;
            const int zz = (o & 1) ? iz1 : iz0;
            const float wz = (iz1 == iz0) ? ((o & 1) ? 0.f : 1.f) : ((o & 1) ? tz : 1.f - tz);
            const bool adds = wz != 0.f, h0 = h == 0;
            const unsigned lc_tile = (unsigned)((__mul24(o >> 3, BY) + 4 * h + ((o >> 1) & 3)) * 8);      // bytes, tile
            const unsigned lc_g = (__umul24((unsigned)h, (unsigned)A.G.nb2) + (unsigned)(zz >> 1)) * kBrickFloats * 4u
                                  + (unsigned)(((o >> 1) << 1) + (zz & 1)) * 4u;                     // bytes, gradient
            const unsigned lc_t = (__umul24((unsigned)h, (unsigned)A.G.nb2) + (unsigned)(zz >> 1)) * 4u;     // bytes, touched flags
            unsigned lc_tile_o = lc_tile;
            asm volatile("" : "+v"(lc_tile_o)); // opaque: or else (4 ci + x) * BY is re-associated into a per-column v_mul_lo_u32
            const char *tile_b = reinterpret_cast<const char *>(tile);
            // The four waves split the box 2 x 2: wave wv takes the brick rows ci = wv >> 1, + 2, ... and of each the column
            // pairs cp = wv & 1, + 2, ... -- two plain nested scalar loops.  (Round 3 dealt the flattened pairs p = ci * npr +
            // cp out round-robin and advanced (ci, cp) by a carry loop, four pairs in flight: ~28 scalar instructions per
            // pair -- two 64-bit multiplies for the brick address among them -- in a kernel whose scalar pipe is as busy as
            // its vector pipe: 408 of them per wave, tools/pmc_step.sh.)  Everything that depends on (ci, cp) is scalar: the
            // tile address is a scalar + a lane constant, the brick a scalar 32-bit index, the memory operands a scalar
            // 64-bit base + a 32-bit lane offset.
            const int wv = __builtin_amdgcn_readfirstlane(wib);
            const unsigned nb2u = (unsigned)A.G.nb2;
            char *const gvol_b = reinterpret_cast<char *>(A.gvol);
            char *const gt_b = reinterpret_cast<char *>(A.gtouched);
            auto flush_pair = [&](unsigned trow, unsigned grow, int cp, double &v, unsigned &ta) { // issue the tile read
                const bool second = 2 * cp + 1 < b1; // wave-uniform: the pair's second column exists
                ta = trow + 64u * (unsigned)cp + lc_tile_o;
                v = (h0 || second) ? *reinterpret_cast<const double *>(tile_b + ta) : 0.0;
                (void)grow;
            };
            auto emit_pair = [&](unsigned grow, int cp, double v, unsigned ta) {
                if (nsub > 1 && v != 0.0 && !(o & 1)) *reinterpret_cast<double *>(const_cast<char *>(tile_b) + ta) = 0.0; // leave the tile clean for the next group
                const bool nz = v != 0.0 && adds;
#ifdef DIFFUS_ABLATE_SC_FLUSH
                asm volatile("" :: "s"(grow), "v"((float)v * wz), "v"(nz));
#else
                if (nz) {
                    const unsigned gb = grow + 2u * (unsigned)cp * nb2u; // brick index of the pair's first column at depth brick 0
                    // every adding lane marks its brick (lanes of a brick store the same word: one write)
                    if (A.gtouched) *reinterpret_cast<int *>(gt_b + (size_t)gb * 4u + (size_t)lc_t) = 1;
                    atomicAdd(reinterpret_cast<float *>(gvol_b + (size_t)gb * (kBrickFloats * 4u) + (size_t)lc_g), (float)v * wz);
                }
#endif
            };
            const int npr = (b1 + 1) >> 1; // column pairs per brick row, in the box
            for (int ci = wv >> 1; ci < b0; ci += 2) {
                const unsigned trow = (unsigned)(4 * ci * BY) * 8u;                                                // bytes, tile
                const unsigned grow = ((unsigned)(l0 + ci) * (unsigned)A.G.nb1 + (unsigned)l1) * nb2u;              // bricks
                for (int cp = wv & 1; cp < npr; cp += 4) { // two pairs per trip: two tile reads in flight
                    double v0, v1 = 0.0;
                    unsigned t0, t1 = 0;
                    const bool two = cp + 2 < npr; // wave-uniform
                    flush_pair(trow, grow, cp, v0, t0);
                    if (two) flush_pair(trow, grow, cp + 2, v1, t1);
                    emit_pair(grow, cp, v0, t0);
                    if (two) emit_pair(grow, cp + 2, v1, t1);
                }
            }
        }
        STAMP(5);
#ifdef DIFFUS_STAMP
        if (threadIdx.x == 0 && g_stamps) {
            g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = (unsigned long long)need;
            g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = (unsigned long long)nsub;
        }
#endif
        if (sp + 1 < nsub) __syncthreads();
    }
    return true;
}


// ---- MERGED planar patches: the part of a fan that has left the volume -------------------------------------------------
// Rays that have left the volume are clamped onto its faces (grid_sample's border rule): a patch of them touches a line or a
// corner of border voxels -- a few dozen tile entries -- and still pays the whole skeleton of a block (pose and zbar loads,
// tile clear, boxes, two barriers, bookkeeping, flush walk): 16-18 k cycles whatever the tile holds, tools/scatter_stamps.py;
// 39 % of the blocks of config 3 are of that kind.  A ray that is outside stays outside (the volume is convex), so once the two
// EDGE rays of a ray group are outside at the first step of a step group, the kernel lets ONE block (the "leader") take that
// step group and the following ones up to the next multiple of four, through ONE tile and ONE skeleton; the blocks of the
// groups it covers ("followers") exit at once.  Leader and followers decide by the same rule from the same two rays, so every
// sample is scattered exactly once whatever the rays in between do (if they are NOT all outside, the union box is merely
// larger; if it does not fit the tile, or a ray turns out not to be planar, the leader adds its samples straight to memory --
// correct, slow, and not seen in any fan of the reference).
// rows / row_off / nbase: as in scatter_patch_planar, for the leader's own (first) step group; ng: step groups it takes.
template <int SAMPLER, int PM, int CAP>
__device__ __forceinline__ void scatter_patch_planar_merged(const Args &A, double *tile, int (*s_box)[4], int *s_planar, int *s_live, const Pose &ps,
                                                            const float *rows, unsigned row_off, bool ray_ok, int nbase, int ng, int tid)
{
    constexpr int kCapD = CAP / 2, kMaxG = 4;
    const int wib = tid >> 6;
    // every group's zbar values first (ng x 16 bytes per thread in flight), the tile cleared under them
    float zb[kMaxG][kSPT];
    {
        const char *rb = reinterpret_cast<const char *>(rows);
#pragma unroll
        for (int g = 0; g < kMaxG; ++g) {
            const int n0 = nbase + g * kScSteps;
            const unsigned off = row_off + (unsigned)(g * kScSteps) * 4u;
            if (g < ng && ray_ok && n0 + kSPT <= A.N1) {
                if constexpr (kSPT == 4) {
                    const F4a4 t = *reinterpret_cast<const F4a4 *>(rb + (size_t)off);
                    zb[g][0] = t.x; zb[g][1] = t.y; zb[g][2] = t.z; zb[g][3] = t.w;
                } else {
#pragma unroll
                    for (int q = 0; q < kSPT; ++q) zb[g][q] = ldb_f32(rows, off + 4u * q);
                }
            } else {
#pragma unroll
                for (int q = 0; q < kSPT; ++q) zb[g][q] = (g < ng && ray_ok && n0 + q < A.N1) ? ldb_f32(rows, off + 4u * q) : 0.f;
            }
        }
    }
    {
        int4 *t4 = reinterpret_cast<int4 *>(tile);
#pragma unroll
        for (int e = 0; e < CAP / 4 / kSB; ++e) t4[e * kSB + tid] = make_int4(0, 0, 0, 0);
    }
    const bool ray_planar = (PM == 0 || ps.pmode != 2) ? (ps.df[2] == 0.f) : (ps.dd[2] == 0.0);
    const bool wave_planar = __ballot(ray_planar) == ~0ull;
    auto cell_xy = [&](int n, int &x0, int &x1, int &y0, int &y1, float &tx, float &ty) {
        const float kf = (float)(A.start + n);
        const float p0 = ray_point_f<PM>(ps, 0, kf), p1 = ray_point_f<PM>(ps, 1, kf);
        if (SAMPLER == DIFFUS_NEAREST) {
            x0 = x1 = nearest_index(p0, A.G.d0); y0 = y1 = nearest_index(p1, A.G.d1);
            tx = ty = 0.f;
        } else {
            const Axis a = tri_axis(p0, A.G.d0), b = tri_axis(p1, A.G.d1);
            x0 = a.i0; x1 = a.i1; tx = a.t; y0 = b.i0; y1 = b.i1; ty = b.t;
        }
    };
    // box: a ray is a straight line and clamp / floor are monotone, so the extremes of a thread's samples over the whole merged
    // range sit at its first and its last one
    int bx[4];
    {
        int ax0, ax1, ay0, ay1, bx0, bx1, by0, by1;
        float t0, t1;
        cell_xy(nbase, ax0, ax1, ay0, ay1, t0, t1);
        cell_xy(nbase + (ng - 1) * kScSteps + kSPT - 1, bx0, bx1, by0, by1, t0, t1);
        const bool has = ray_ok && nbase < A.N1;
        bx[0] = has ? min(ax0, bx0) : 0x7fffffff; bx[1] = has ? max(ax1, bx1) : -1;
        bx[2] = has ? min(ay0, by0) : 0x7fffffff; bx[3] = has ? max(ay1, by1) : -1;
    }
    if (tid == 0) *s_live = 0;
    bx[0] = wave_reduce_minmax<true>(bx[0]);
    bx[1] = wave_reduce_minmax<false>(bx[1]);
    bx[2] = wave_reduce_minmax<true>(bx[2]);
    bx[3] = wave_reduce_minmax<false>(bx[3]);
    if ((tid & 63) == 63) {
        s_planar[wib] = wave_planar;
#pragma unroll
        for (int a = 0; a < 4; ++a) s_box[wib][a] = bx[a];
    }
    __syncthreads(); // also: the tile is clear
    int all_planar = 1, mn0 = 0x7fffffff, mx0 = -1, mn1 = 0x7fffffff, mx1 = -1;
#pragma unroll
    for (int wv = 0; wv < kSW; ++wv) {
        all_planar &= s_planar[wv];
        mn0 = min(mn0, __builtin_amdgcn_readfirstlane(s_box[wv][0]) >> 2); mx0 = max(mx0, __builtin_amdgcn_readfirstlane(s_box[wv][1]) >> 2);
        mn1 = min(mn1, __builtin_amdgcn_readfirstlane(s_box[wv][2]) >> 2); mx1 = max(mx1, __builtin_amdgcn_readfirstlane(s_box[wv][3]) >> 2);
    }
    if (mx0 < 0 || mx1 < 0) return; // no sample at all (block-uniform)
    const int l0 = mn0, l1 = mn1, b0 = mx0 - mn0 + 1, b1 = mx1 - mn1 + 1;
    const unsigned need = min(4u * (unsigned)min(b0, 0x3fff) * (4u * (unsigned)min(b1, 0x3fff) + kRowPad), (unsigned)kCapD + 1u);
    if (!__builtin_amdgcn_readfirstlane(all_planar) || need > (unsigned)kCapD) {
        // not planar after all (a wrong DIFFUS_FANS_PLANAR promise), or the union box exceeds the tile: every sample straight to
        // memory with its full 3-D cell
#pragma unroll 1
        for (int g = 0; g < ng; ++g)
#pragma unroll 1
            for (int q = 0; q < kSPT; ++q) {
                float v = 0.f;
#pragma unroll
                for (int gg = 0; gg < kMaxG; ++gg)
#pragma unroll
                    for (int qq = 0; qq < kSPT; ++qq) v = (gg == g && qq == q) ? zb[gg][qq] : v;
                if (!finitef(v) || v == 0.f) continue;
                const Cell c = cell_of<SAMPLER, PM>(A, ps, A.start + nbase + g * kScSteps + q);
                for_each_corner<SAMPLER>(c, v, [&](int i, int j, int k, float w) {
                    if (w != 0.f) {
                        const unsigned gi = vox_off<DIFFUS_BRICKED>(A.G, i, j, k);
                        atomicAdd(A.gvol + gi, w);
                        if (A.gtouched) A.gtouched[gi >> 5] = 1;
                    }
                });
            }
        return;
    }
    // the patch's dim-2 cell: the same for every sample of a planar fan
    int iz0, iz1;
    float tz;
    {
        const float p2 = ray_point_f<PM>(ps, 2, 0.f);
        if (SAMPLER == DIFFUS_NEAREST) {
            iz0 = iz1 = nearest_index(p2, A.G.d2);
            tz = 0.f;
        } else {
            const Axis c = tri_axis(p2, A.G.d2);
            iz0 = c.i0; iz1 = c.i1; tz = c.t;
        }
        iz0 = __builtin_amdgcn_readfirstlane(iz0); iz1 = __builtin_amdgcn_readfirstlane(iz1);
        tz = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(tz)));
    }
    const int BY = 4 * b1 + kRowPad;
    const int org = -(4 * l0) * BY - 4 * l1;
    const int BY8 = BY * 8, org8 = org * 8;
    char *tile_c = reinterpret_cast<char *>(tile);
    auto add_at = [&](int byte_off, double v) { atomicAdd(reinterpret_cast<double *>(tile_c + byte_off), v); };
    unsigned nz = 0;
    // ---- accumulate, group after group (the code of scatter_patch_planar's accumulation; see there for the same-cell path)
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        if (g >= ng) break; // block-uniform
#pragma unroll
        for (int q = 0; q < kSPT; ++q) {
            float zq = zb[g][q];
            if (!finitef(zq)) zq = 0.f;
            nz |= __float_as_uint(zq) & 0x7fffffffu;
            const unsigned long long act = __builtin_amdgcn_ballot_w64(zq != 0.f);
            if (act == 0ull) continue; // wave-uniform
            int x0, x1, y0, y1;
            float tx, ty;
            cell_xy(nbase + g * kScSteps + q, x0, x1, y0, y1, tx, ty);
            const int r0 = __mul24(x0, BY8) + org8, r1 = __mul24(x1, BY8) + org8;
            const int e00 = r0 + 8 * y0, e11 = r1 + 8 * y1;
            float c00, c01 = 0.f, c10 = 0.f, c11 = 0.f;
            if constexpr (SAMPLER == DIFFUS_NEAREST) {
                c00 = zq;
            } else {
                const float wa1 = tx, wa0 = 1.f - wa1, wb1 = ty, wb0 = 1.f - wb1;
                const float s0 = zq * wa0, s1 = zq * wa1;
                c00 = s0 * wb0; c01 = s0 * wb1; c10 = s1 * wb0; c11 = s1 * wb1;
            }
            const int lead = __builtin_ctzll(act);
            const int f00 = __builtin_amdgcn_readlane(e00, lead), f11 = __builtin_amdgcn_readlane(e11, lead);
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(e00 == f00) & __builtin_amdgcn_ballot_w64(e11 == f11);
            if ((eq & act) == act && __builtin_popcountll(act) > 4) { // wave-uniform: one cell for every live lane
                const int f01 = __builtin_amdgcn_readlane(r0 + 8 * y1, lead), f10 = __builtin_amdgcn_readlane(r1 + 8 * y0, lead);
                const double t00 = wave_sum_to_lane63((double)c00);
                double t01 = 0.0, t10 = 0.0, t11 = 0.0;
                if constexpr (SAMPLER != DIFFUS_NEAREST) {
                    t01 = wave_sum_to_lane63((double)c01);
                    t10 = wave_sum_to_lane63((double)c10);
                    t11 = wave_sum_to_lane63((double)c11);
                }
                if ((tid & 63) == 63) {
                    if (t00 != 0.0) add_at(f00, t00);
                    if (t01 != 0.0) add_at(f01, t01);
                    if (t10 != 0.0) add_at(f10, t10);
                    if (t11 != 0.0) add_at(f11, t11);
                }
            } else {
                if (c00 != 0.f) add_at(e00, (double)c00);
                if constexpr (SAMPLER != DIFFUS_NEAREST) {
                    if (c01 != 0.f) add_at(r0 + 8 * y1, (double)c01);
                    if (c10 != 0.f) add_at(r1 + 8 * y0, (double)c10);
                    if (c11 != 0.f) add_at(e11, (double)c11);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (__builtin_amdgcn_ballot_w64(nz != 0u) != 0ull && (tid & 63) == 63) *s_live = 1;
    __syncthreads();
    if (__builtin_amdgcn_readfirstlane(*s_live) == 0) return; // nothing added: the tile is still clear (block-uniform)
    // ---- flush (scatter_patch_planar's, for one group of waves): a wave takes two adjacent brick columns of one brick row
    {
        const int lane = tid & 63, o = lane & 31, h = lane >> 5;
        const int zz = (o & 1) ? iz1 : iz0;
        const float wz = (iz1 == iz0) ? ((o & 1) ? 0.f : 1.f) : ((o & 1) ? tz : 1.f - tz);
        const bool adds = wz != 0.f, h0 = h == 0;
        const unsigned lc_tile = (unsigned)((__mul24(o >> 3, BY) + 4 * h + ((o >> 1) & 3)) * 8);
        const unsigned lc_g = (__umul24((unsigned)h, (unsigned)A.G.nb2) + (unsigned)(zz >> 1)) * kBrickFloats * 4u
                              + (unsigned)(((o >> 1) << 1) + (zz & 1)) * 4u;
        const unsigned lc_t = (__umul24((unsigned)h, (unsigned)A.G.nb2) + (unsigned)(zz >> 1)) * 4u;
        unsigned lc_tile_o = lc_tile;
        asm volatile("" : "+v"(lc_tile_o));
        const char *tile_b = reinterpret_cast<const char *>(tile);
        const int wv = __builtin_amdgcn_readfirstlane(wib);
        const unsigned nb2u = (unsigned)A.G.nb2;
        char *const gvol_b = reinterpret_cast<char *>(A.gvol);
        char *const gt_b = reinterpret_cast<char *>(A.gtouched);
        const int npr = (b1 + 1) >> 1;
        for (int ci = wv >> 1; ci < b0; ci += 2) {
            const unsigned trow = (unsigned)(4 * ci * BY) * 8u;
            const unsigned grow = ((unsigned)(l0 + ci) * (unsigned)A.G.nb1 + (unsigned)l1) * nb2u;
            for (int cp = wv & 1; cp < npr; cp += 2) {
                const bool second = 2 * cp + 1 < b1; // wave-uniform
                const double v = (h0 || second) ? *reinterpret_cast<const double *>(tile_b + trow + 64u * (unsigned)cp + lc_tile_o) : 0.0;
                if (v != 0.0 && adds) {
                    const unsigned gb = grow + 2u * (unsigned)cp * nb2u;
                    if (A.gtouched) *reinterpret_cast<int *>(gt_b + (size_t)gb * 4u + (size_t)lc_t) = 1;
                    atomicAdd(reinterpret_cast<float *>(gvol_b + (size_t)gb * (kBrickFloats * 4u) + (size_t)lc_g), (float)v * wz);
                }
            }
        }
    }
}

// ---- SLAB patches (bricked gradient): fans that are NOT planar in dim 2 --------------------------------------------
// `plot_beam_frame` takes any `directions` (src/renderer.py:119-124, :201-217), and a probe-pose optimisation produces
// exactly such fans: the plane rolled about the central ray, the central ray pitched out of the slice, a fan lying in
// another coordinate plane.  The 3-D brick tile of the general path below holds a box of WHOLE bricks; a patch of a
// plane tilted by 20 degrees crosses ~12 depth layers, its box is ~9 x 9 x 6 bricks against a tile of 192, most waves
// fall through to direct global atomics: 132 us at 20 degrees of roll, 488 us at 20 degrees of pitch against 26 us
// for planar fans (profiles/r05_tilt_before.txt), in 32-bit fixed point.
//
// The rays of a patch share their source and -- for every fan that is a rigid motion of a planar one -- lie in ONE plane
// through it.  Let A be the axis along which that plane's normal is largest ("minor axis") and (U, V) the other two:
// the plane is a height field  h(U, V) = c0 + cu U + cv V  with |cu|, |cv| <= 1.  A sample INSIDE the volume at p touches
// the columns (U, V) in floor(p_U, p_V) + {0, 1}^2 and, in each, the layers floor(p_A) + {0, 1}; since
// |h(U, V) - p_A| <= |cu| + |cv| =: S, every layer a column can receive lies in the window [floor(h - S), floor(h + S) + 1]
// of at most L = ceil(2 S) + 2 consecutive integers.  So the tile is 2-D over the box of touched columns x L SLOTS,
// slot = layer mod L: unambiguous inside a window of L, no per-corner evaluation of the plane in the accumulation; the
// flush recovers the layer from the column's window.  Doubles (ds_add_f64), like the planar tile: the per-voxel error
// bound of the planar path holds here too (tests/test_tilted_fans.py).  The plane is not assumed, it is MEASURED: the
// largest distance dev of any live sample from h (float rounding of the points included) widens S, and a patch whose
// rays are not coplanar enough for L <= kSlabMaxL takes the general path.
//
// Samples OUTSIDE the volume are clamped onto its faces (grid_sample's border rule; reference :754-756 for nearest) and
// leave the plane.  Those clamped on axis a form a 2-D set ON that face: the same tile with minor axis a, two slots
// (layer 0 and layer dim_a - 1).  A patch is therefore scattered in up to four PASSES -- the samples clamped on axis 0,
// those on axis 1 (and not 0), those on axis 2 (and not 0, 1), the inside ones -- of which a typical patch needs one.
// A pass whose column box x L exceeds the tile goes through it in CHUNKS of rows (every thread offers its corners to
// every chunk): no patch ever falls back to per-sample global atomics.
constexpr int kSlabMaxL = 8;
template <int AX> struct SlabAxes { // (U, V) for minor axis AX; V is the axis whose neighbours share a brick line most often
    static constexpr int U = (AX == 2) ? 0 : 2, V = (AX == 1) ? 0 : 1;
};
__device__ __forceinline__ float sel3(float a, float b, float c, int i) { return i == 0 ? a : (i == 1 ? b : c); }
__device__ __forceinline__ int mod_rel(int x, int L, int magic16) { return x - __mul24(__mul24(x, magic16) >> 16, L); } // 0 <= x < 2^12, magic16 = ceil(2^16 / L)

struct SlabSample { // one of a thread's kSPT samples
    int i0[3], i1[3];
    float t[3];
};

// byte offset of voxel (x, y, z) in the bricked gradient as a part that depends on the column (U, V) and a part that depends
// on the layer M along the minor axis AX
template <int AX>
__device__ __forceinline__ unsigned slab_col_part(const Geom &G, int Uc, int Vc)
{
    constexpr int U = SlabAxes<AX>::U;
    const int a = U == 0 ? Uc : Vc, b = U == 0 ? Vc : Uc; // the two column coordinates in axis order (U, V are 0 < 1, 2 > 1, 2 > 0)
    if (AX == 2) return (((unsigned)(a >> 2) * (unsigned)G.nb1 + (unsigned)(b >> 2)) * (unsigned)G.nb2 * kBrickFloats + (unsigned)(((a & 3) << 3) | ((b & 3) << 1))) * 4u; // (x, y)
    if (AX == 1) return ((unsigned)(a >> 2) * (unsigned)G.nb1 * (unsigned)G.nb2 * kBrickFloats + (unsigned)(b >> 1) * kBrickFloats + (unsigned)(((a & 3) << 3) | (b & 1))) * 4u; // (x, z)
    return ((unsigned)(a >> 2) * (unsigned)G.nb2 * kBrickFloats + (unsigned)(b >> 1) * kBrickFloats + (unsigned)(((a & 3) << 1) | (b & 1))) * 4u; // AX == 0: (y, z)
}
template <int AX>
__device__ __forceinline__ unsigned slab_ax_part(const Geom &G, unsigned rowA, int M) // rowA: bricks per unit of M >> 2 (AX 0: nb1 nb2, AX 1: nb2)
{
    if (AX == 2) return ((unsigned)(M >> 1) * kBrickFloats + (unsigned)(M & 1)) * 4u;
    if (AX == 1) return ((unsigned)(M >> 2) * rowA * kBrickFloats + (unsigned)((M & 3) << 1)) * 4u;
    return ((unsigned)(M >> 2) * rowA * kBrickFloats + (unsigned)((M & 3) << 3)) * 4u;
}

// One pass.  `member`: bit q = sample q of this thread belongs to the pass.  FACE: the pass of the samples clamped on
// axis AX (slots: layer 0 / layer dim - 1; the axis carries weight 1, so a sample has FOUR corners); else the inside
// samples on the plane (cu, cv, c0b = c0 - S; L slots, eight corners).
// (u0, u1, v0, v1): the pass's column box (block-uniform).  last: nothing follows this pass (no closing barrier).
template <int SAMPLER, int AX, bool FACE, int CAPD>
__device__ __forceinline__ void slab_pass(const Args &A, double *tile, const SlabSample (&sm)[kSPT], const float (&zb)[kSPT], unsigned member,
                                          float cu, float cv, float c0b, int L, int u0, int u1, int v0, int v1, bool last, int tid,
                                          unsigned long long (&prof)[5])
{
    constexpr int U = SlabAxes<AX>::U, V = SlabAxes<AX>::V;
    (void)prof;
    const int dimA = AX == 0 ? A.G.d0 : (AX == 1 ? A.G.d1 : A.G.d2);
    const int BV = v1 - v0 + 1, BVp = max(BV | 1, 3); // odd row stride (bank spread); >= 3: the magic of 1 does not fit 32 bits
    if ((long)BVp * L > CAPD || BVp > 512 || (u1 - u0) + BVp > 4000) { // (512: the range the approximate magicV below is exact for; 4000: the 16-bit slot magic)
        // A single row of columns exceeds the tile (steps of hundreds of voxels): per-corner global atomics.  ONE copy of the
        // corner code in a rolled loop, the sample picked by selects (unrolled, its 32 weight tests at a time spilled 130 registers)
#pragma unroll 1
        for (int qq = 0; qq < kSPT; ++qq) {
            if (!(member >> qq & 1u)) continue;
            Cell c;
            float zq = 0.f;
#pragma unroll
            for (int q = 0; q < kSPT; ++q)
                if (q == qq) {
                    zq = zb[q];
#pragma unroll
                    for (int a = 0; a < 3; ++a) { c.i0[a] = sm[q].i0[a]; c.i1[a] = sm[q].i1[a]; c.t[a] = sm[q].t[a]; }
                }
            for_each_corner<SAMPLER>(c, zq, [&](int i, int j, int k, float v) {
                if (v != 0.f) {
                    const unsigned g = vox_off<DIFFUS_BRICKED>(A.G, i, j, k);
                    atomicAdd(A.gvol + g, v);
                    if (A.gtouched) A.gtouched[g >> 5] = 1;
                }
            });
        }
        return;
    }
    // rows of columns per chunk = floor(CAPD / (BVp L)), without an integer division (block-uniform, but a division is ~40
    // instructions wherever it runs): float estimate, corrected by one either way
    const int per_row = BVp * L, BU = u1 - u0 + 1;
    int rpc = (int)((float)CAPD * __builtin_amdgcn_rcpf((float)per_row));
    rpc += ((rpc + 1) * per_row <= CAPD) ? 1 : 0;
    rpc -= (rpc * per_row > CAPD) ? 1 : 0;
    rpc = min(rpc, BU);
    const int NC = rpc * BVp;                           // columns per slot plane
    const int NC8 = NC * 8, BVp8 = BVp * 8;
    // slot = layer mod L on layers RELATIVE to a0, a block-uniform lower bound of every column's window (the plane is linear:
    // its minimum over the box is at a corner): a few hundred at most, so x mod L = x - ((x m16) >> 16) L with m16 =
    // ceil(2^16 / L) -- three full-rate instructions where the 32-bit magic costs two quarter-rate multiplies
    const int magic16 = (65535 / L) + 1;
    int a0 = 0;
    if (!FACE) {
        const float h00 = __builtin_fmaf(cv, (float)v0, __builtin_fmaf(cu, (float)u0, c0b)), h01 = __builtin_fmaf(cv, (float)v1, __builtin_fmaf(cu, (float)u0, c0b));
        const float h10 = __builtin_fmaf(cv, (float)v0, __builtin_fmaf(cu, (float)u1, c0b)), h11 = __builtin_fmaf(cv, (float)v1, __builtin_fmaf(cu, (float)u1, c0b));
        a0 = (int)floorf(fminf(fminf(h00, h01), fminf(h10, h11))) - 1; // - 1: the roundings of the two evaluation orders
        a0 = __builtin_amdgcn_readfirstlane(a0);
    }
    // col / BVp for col < 2^13: any magic in [2^32 / BVp, (2^32 + 2^19) / BVp) is exact there; the float quotient is within 2^-22
    // relative (<= 341 for BVp >= 3) of the true one, so + 512 puts it inside that window for every BVp <= 512
    const unsigned magicV = (unsigned)(4294967296.f * __builtin_amdgcn_rcpf((float)BVp)) + 512u;
    static_assert(CAPD <= 8192, "magicV is exact for column indices below 2^13 only");
    char *tile_c = reinterpret_cast<char *>(tile);
#if defined(DIFFUS_SLAB_EXIT) && DIFFUS_SLAB_EXIT == 6
    auto add_at = [&](int byte_off, double v) { asm volatile("" :: "v"(byte_off), "v"(v)); };
#else
    auto add_at = [&](int byte_off, double v) { atomicAdd(reinterpret_cast<double *>(tile_c + byte_off), v); };
#endif
    const unsigned rowA = AX == 0 ? (unsigned)A.G.nb1 * (unsigned)A.G.nb2 : (unsigned)A.G.nb2;
    char *const gvol_b = reinterpret_cast<char *>(A.gvol);
#pragma unroll 1
    for (int uc = u0; uc <= u1; uc += rpc) { // block-uniform
        const int rows = min(rpc, u1 - uc + 1);
        const bool whole = rpc >= BU; // the usual case: one chunk holds the pass, no corner needs a row test
#ifdef DIFFUS_STAMP
        const unsigned long long t_a = STAMP_NOW();
        if (prof[2] == 0) prof[4] = t_a; // when the first accumulation starts
#endif
        // ---- accumulate: 8 (inside) / 4 (face) / 1 (nearest) ds_add_f64 per sample, zero weights skipped
#pragma unroll
        for (int q = 0; q < kSPT; ++q) {
            const bool on = member >> q & 1u;
            if (__builtin_amdgcn_ballot_w64(on) == 0ull) continue; // wave-uniform
            const SlabSample &c = sm[q];
            float zq = on ? zb[q] : 0.f; // a zero makes every weight below an exact zero: no add
            asm volatile("" : "+v"(zq)); // opaque: every weight product depends on it, so none is hoisted out of the chunk and pass loops
            const int ru0 = c.i0[U] - uc, ru1 = c.i1[U] - uc;
            const bool inr[2] = {whole || (unsigned)ru0 < (unsigned)rows, whole || (unsigned)ru1 < (unsigned)rows};
            const int s0 = FACE ? (c.i0[AX] != 0 ? 1 : 0) : mod_rel(c.i0[AX] - a0, L, magic16);
            const int eS0 = __mul24(s0, NC8);
            const int eU[2] = {__mul24(ru0, BVp8), __mul24(ru1, BVp8)};
            const int eV[2] = {(c.i0[V] - v0) * 8, (c.i1[V] - v0) * 8};
            if constexpr (SAMPLER == DIFFUS_NEAREST) {
                if (inr[0] && zq != 0.f) add_at(eS0 + eU[0] + eV[0], (double)zq);
            } else {
                const float tu = c.t[U], tv = c.t[V];
                const float wu[2] = {1.f - tu, tu}, wv[2] = {1.f - tv, tv};
                if constexpr (FACE) {
                    // zq * w0 * w1 * w2 in for_each_corner's association, the clamped axis contributing an exact 1
                    float wc[2][2];
#pragma unroll
                    for (int iu = 0; iu < 2; ++iu)
#pragma unroll
                        for (int iv = 0; iv < 2; ++iv) {
                            const float first = (U < V) ? wu[iu] : wv[iv], second = (U < V) ? wv[iv] : wu[iu]; // lower axis first
                            wc[iu][iv] = inr[iu] ? zq * first * second : 0.f;
                        }
                    // Samples clamped onto an EDGE or a corner of the volume stay on one voxel for the rest of their steps, and
                    // the rays around them on the same one: 64 lanes, one address -- served one after the other (+23 cycles per
                    // duplicate, tools/lds_atomic_bench.hip).  As in the planar path: if every live lane of the wave sits in the
                    // same cell, the contributions are summed over the wave (DPP, double) and one lane adds them.
                    const unsigned long long act = __builtin_amdgcn_ballot_w64(zq != 0.f);
                    if (act == 0ull) continue; // wave-uniform
                    const int lead = __builtin_ctzll(act);
                    const int k0 = eS0 + eU[0] + eV[0], k1 = eU[1] + eV[1];
                    const int f0 = __builtin_amdgcn_readlane(k0, lead), f1 = __builtin_amdgcn_readlane(k1, lead);
                    const unsigned long long eq = __builtin_amdgcn_ballot_w64(k0 == f0) & __builtin_amdgcn_ballot_w64(k1 == f1);
                    if ((eq & act) == act && __builtin_popcountll(act) > 4) {
#pragma unroll
                        for (int iu = 0; iu < 2; ++iu)
#pragma unroll
                            for (int iv = 0; iv < 2; ++iv) {
                                const double t = wave_sum_to_lane63((double)wc[iu][iv]);
                                const int off = __builtin_amdgcn_readlane(eS0 + eU[iu] + eV[iv], lead);
                                if ((tid & 63) == 63 && t != 0.0) add_at(off, t);
                            }
                    } else {
#pragma unroll
                        for (int iu = 0; iu < 2; ++iu)
#pragma unroll
                            for (int iv = 0; iv < 2; ++iv)
                                if (wc[iu][iv] != 0.f) add_at(eS0 + eU[iu] + eV[iv], (double)wc[iu][iv]);
                    }
                } else {
                    const int s1 = (s0 + 1 == L) ? 0 : s0 + 1;
                    const int eS[2] = {eS0, __mul24(s1, NC8)};
                    const float w0[2] = {1.f - c.t[0], c.t[0]}, w1[2] = {1.f - c.t[1], c.t[1]}, w2[2] = {1.f - c.t[2], c.t[2]};
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const float wij = zq * w0[i] * w1[j]; // the association of for_each_corner
#pragma unroll
                            for (int k = 0; k < 2; ++k) {
                                const float wt = wij * w2[k];
                                const int bit[3] = {i, j, k};
                                if (wt != 0.f && inr[bit[U]]) add_at(eS[bit[AX]] + eU[bit[U]] + eV[bit[V]], (double)wt);
                            }
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0); // one sample at a time (register budget)
        }
        __syncthreads();
#ifdef DIFFUS_STAMP
        const unsigned long long t_f = STAMP_NOW();
        prof[0] += t_f - t_a; prof[2] += 1; prof[3] += (unsigned long long)(rows * BVp * L);
#endif
#ifdef DIFFUS_SLAB_EXIT
        if (DIFFUS_SLAB_EXIT == 4) return;
#endif
        // Two lane orders for the flush, chosen per pass (block-uniform).  BLOCK-shaped for face passes, for steep planes (L >= 4:
        // the layers of neighbouring columns differ) and whenever dim 1 is the minor axis (the column-major order then runs
        // along dim 0: a 32-byte stride inside a brick); column-major for flat planes whose columns run along dim 1.  Measured at
        // the config 3 poses, scatter in us, column-major everywhere / this choice: roll 5 deg 56.5 / 55.4, roll 20 deg 64.0 / 58.0,
        // roll 45 deg 73 / 63.6, pitch 20 deg 67.8 / 58.8, fan in the (0,2) plane 64.5 / 58.0, in the (1,2) plane 53 / 52.5
        // (block-shaped there: 57.7; roll 5 deg block-shaped: 60.1).
        const bool block_flush = FACE || L >= 4 || AX == 1;
        // ---- flush, BLOCK-shaped: a wave takes a 4 x 4 block of columns (aligned to multiples of 4 in U and V: the brick grid) and,
        // of each column, four slots -- lane = 4 (column in the block) + slot.  The 64 atomics of one instruction then land in the
        // two to four bricks the block's windows span, where the column-major order below (64 consecutive V) spread them over
        // 16-32 lines: the flush's global atomics were 12.8 us of the kernel at 20 degrees of roll (stage-exit probes), 5x the
        // planar path's.  Recovers the layers from the columns' windows, adds every non-zero entry to the gradient once and
        // leaves the tile all-zero.
        if (block_flush) {
            // (face passes have two slots: 8 x 4 columns x 2 slots per instruction instead of 4 x 4 x 4)
            constexpr int SB = FACE ? 1 : 2, UB = FACE ? 3 : 2; // log2: slots per lane group, U rows of a block
            const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
            const int du = lane >> (SB + 2), dv = (lane >> SB) & 3, s_in = lane & ((1 << SB) - 1);
            const int bu0 = uc >> UB, nbu = ((uc + rows - 1) >> UB) - bu0 + 1, bv0 = v0 >> 2, nbv = (v1 >> 2) - bv0 + 1;
            const float rnbv = __builtin_amdgcn_rcpf((float)nbv);
            const int nblk = nbu * nbv;
#pragma unroll 1
            for (int blk = wv; blk < nblk; blk += kSW) { // wave-uniform
                const int qb = (int)(((float)blk + 0.5f) * rnbv); // blk / nbv: exact for the few hundred blocks of a box
                const int Uc = ((bu0 + qb) << UB) + du, Vc = ((bv0 + (blk - qb * nbv)) << 2) + dv;
                const bool inbox = (unsigned)(Uc - uc) < (unsigned)rows && (unsigned)(Vc - v0) < (unsigned)BV;
                const int col = (Uc - uc) * BVp + (Vc - v0);
                int base = 0, bm = 0;
                if (!FACE) {
                    base = (int)floorf(__builtin_fmaf(cv, (float)Vc, __builtin_fmaf(cu, (float)Uc, c0b)));
                    bm = mod_rel(max(base - a0, 0), L, magic16); // (base >= a0 inside the box)
                }
                const unsigned cpart = slab_col_part<AX>(A.G, Uc, Vc);
#pragma unroll
                for (int sg = 0; sg < (FACE ? 2 : kSlabMaxL); sg += (1 << SB)) {
                    if (sg >= L) break; // wave-uniform
                    const int sl = sg + s_in;
                    const bool mine = inbox && sl < L;
                    const double v = mine ? tile[sl * NC + col] : 0.0;
                    if (v != 0.0) {
                        tile[sl * NC + col] = 0.0;
                        int d = sl - bm;
                        d += (d < 0) ? L : 0;
                        const int M = FACE ? (sl ? dimA - 1 : 0) : base + d;
                        if ((unsigned)M < (unsigned)dimA) { // always, while the window holds; never an out-of-bounds atomic
                            const unsigned gb = cpart + slab_ax_part<AX>(A.G, rowA, M); // bytes
#if defined(DIFFUS_SLAB_EXIT) && DIFFUS_SLAB_EXIT == 5
                            asm volatile("" :: "v"(gb), "v"((float)v));
#else
                            if (A.gtouched) A.gtouched[gb >> 7] = 1;
                            atomicAdd(reinterpret_cast<float *>(gvol_b + (size_t)gb), (float)v);
#endif
                        }
                    }
                }
            }
        } else {
            // ---- flush: a lane takes a column (V fastest: neighbours in a brick line), reads its L slots, recovers the layers
            // from the column's window and adds every non-zero entry to the gradient once; the tile is left all-zero
            const int ncols = rows * BVp;
#pragma unroll 1
            for (int col = tid; col < ncols; col += kSB) {
                const int r = (int)__umulhi((unsigned)col, magicV), cc = col - r * BVp;
                double v[kSlabMaxL];
#pragma unroll
                for (int sl = 0; sl < kSlabMaxL; ++sl) v[sl] = (sl < L) ? tile[sl * NC + col] : 0.0; // (sl < L: wave-uniform)
                const int Uc = uc + r, Vc = v0 + cc;
                int base = 0, bm = 0;
                if (!FACE) {
                    base = (int)floorf(__builtin_fmaf(cv, (float)Vc, __builtin_fmaf(cu, (float)Uc, c0b)));
                    bm = mod_rel(max(base - a0, 0), L, magic16); // (base >= a0 inside the box; the clamp only keeps padding columns tame)
                }
                const unsigned cpart = slab_col_part<AX>(A.G, Uc, Vc);
#pragma unroll
                for (int sl = 0; sl < kSlabMaxL; ++sl) {
                    if (sl < L && v[sl] != 0.0) {
                        tile[sl * NC + col] = 0.0;
                        int d = sl - bm;
                        d += (d < 0) ? L : 0;
                        const int M = FACE ? (sl ? dimA - 1 : 0) : base + d;
                        if ((unsigned)M >= (unsigned)dimA) continue; // cannot happen while the window holds; never an out-of-bounds atomic
                        const unsigned gb = cpart + slab_ax_part<AX>(A.G, rowA, M); // bytes
#if defined(DIFFUS_SLAB_EXIT) && DIFFUS_SLAB_EXIT == 5
                        asm volatile("" :: "v"(gb), "v"((float)v[sl]));
#else
                        if (A.gtouched) A.gtouched[gb >> 7] = 1;
                        atomicAdd(reinterpret_cast<float *>(gvol_b + (size_t)gb), (float)v[sl]);
#endif
                    }
                }
            }
        }
        if (!(last && uc + rpc > u1)) __syncthreads(); // the next chunk / pass adds into the entries this one has just cleared
#ifdef DIFFUS_STAMP
        prof[1] += STAMP_NOW() - t_f;
#endif
    }
}

// Returns false -- nothing added, tile clear -- when the patch's rays are not coplanar enough (the caller then runs the
// general 3-D path); true when the patch is done.  Enters with the tile clear.
template <int SAMPLER, int PM, int CAP>
__device__ __forceinline__ bool scatter_patch_slab(const Args &A, double *tile, int (*s_rec)[18], const Pose &ps, const float (&dfl)[6],
                                                   float (&zb)[kSPT], int nbase, int tid)
{
    // dfl: the directions of the patch's first and last ray; zb: the thread's zbar values (0 where it has no sample) --
    // both requested by the kernel before anything waits on memory
    constexpr int CAPD = CAP / 2;
    const int wib = tid >> 6;
    STAMP(1);
#ifdef DIFFUS_SLAB_EXIT // stage probe (tools/time_slab_exits.py): return after stage n with the values so far forced live
#define SLAB_EXIT(n, ...) if (DIFFUS_SLAB_EXIT == (n)) { asm volatile("" :: __VA_ARGS__); return true; }
#else
#define SLAB_EXIT(n, ...) ((void)0)
#endif
    SLAB_EXIT(1, "v"(zb[0]), "v"(zb[1]), "v"(zb[2]), "v"(zb[3]), "s"(dfl[0]), "s"(dfl[5]));
    bool has = false;
#pragma unroll
    for (int q = 0; q < kSPT; ++q) {
        if (!finitef(zb[q])) zb[q] = 0.f;
        has |= zb[q] != 0.f;
    }
    // ---- the plane of the patch: through the source, spanned by its first and last ray (block-uniform)
    float nrm[3];
    {
        const float *a = dfl, *b = dfl + 3;
        nrm[0] = a[1] * b[2] - a[2] * b[1]; nrm[1] = a[2] * b[0] - a[0] * b[2]; nrm[2] = a[0] * b[1] - a[1] * b[0];
        const float nn = nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2];
        const float aa = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], bb = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
        if (!(nn > 1e-10f * aa * bb)) { // one ray, parallel rays, a zero direction: any plane through the first ray
            const float f0 = fabsf(a[0]), f1 = fabsf(a[1]), f2 = fabsf(a[2]);
            const int e = (f0 <= f1 && f0 <= f2) ? 0 : (f1 <= f2 ? 1 : 2); // a x e_min
            nrm[0] = e == 1 ? -a[2] : (e == 2 ? a[1] : 0.f);
            nrm[1] = e == 0 ? a[2] : (e == 2 ? -a[0] : 0.f);
            nrm[2] = e == 0 ? -a[1] : (e == 1 ? a[0] : 0.f);
            if (!(nrm[0] * nrm[0] + nrm[1] * nrm[1] + nrm[2] * nrm[2] > 0.f)) { nrm[0] = nrm[1] = 0.f; nrm[2] = 1.f; }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) nrm[c] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nrm[c])));
    }
    const float m0 = fabsf(nrm[0]), m1 = fabsf(nrm[1]), m2 = fabsf(nrm[2]);
    const int ax = (m2 >= m0 && m2 >= m1) ? 2 : (m0 >= m1 ? 0 : 1); // dim 2 on ties: the layout's cheap axis
    const int axU = ax == 2 ? 0 : 2, axV = ax == 1 ? 0 : 1;
    const float inv = -1.f / sel3(nrm[0], nrm[1], nrm[2], ax);
    const float cu = sel3(nrm[0], nrm[1], nrm[2], axU) * inv, cv = sel3(nrm[0], nrm[1], nrm[2], axV) * inv;
    const float sA = sel3(ps.sf[0], ps.sf[1], ps.sf[2], ax), sU = sel3(ps.sf[0], ps.sf[1], ps.sf[2], axU), sV = sel3(ps.sf[0], ps.sf[1], ps.sf[2], axV);
    const float c0 = sA - cu * sU - cv * sV;
    // ---- how far this thread's RAY leaves that plane: p - h(p) is linear in the step (the source lies on the plane), so its
    // largest value over the thread's samples is at the last one; plus what the float32 evaluation of the points themselves
    // can be off by (the reference's rounding sequence, f64 poses included)
    float dev = 0.f;
    if (has) {
        const float klast = (float)(A.start + nbase + kSPT - 1);
        const float dA = sel3(ps.df[0], ps.df[1], ps.df[2], ax), dU = sel3(ps.df[0], ps.df[1], ps.df[2], axU), dV = sel3(ps.df[0], ps.df[1], ps.df[2], axV);
        const float pmag = fmaxf(fmaxf(fabsf(ps.sf[0]) + klast * fabsf(ps.df[0]), fabsf(ps.sf[1]) + klast * fabsf(ps.df[1])),
                                 fabsf(ps.sf[2]) + klast * fabsf(ps.df[2]));
        dev = klast * fabsf(dA - cu * dU - cv * dV) + pmag * 0x1p-19f;
    }
    // ---- cells and classes
    SlabSample sm[kSPT];
    unsigned cls = 0; // 2 bits per sample: 0 inside the volume, 1 + a clamped on axis a (the lowest such a)
    const int dims[3] = {A.G.d0, A.G.d1, A.G.d2};
#pragma unroll
    for (int q = 0; q < kSPT; ++q) {
        const float kf = (float)(A.start + nbase + q);
        bool in[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float p = ray_point_f<PM>(ps, a, kf);
            if (SAMPLER == DIFFUS_NEAREST) {
                const int i = nearest_index(p, dims[a]);
                sm[q].i0[a] = sm[q].i1[a] = i;
                sm[q].t[a] = 0.f;
                in[a] = rintf(p) == (float)i;
            } else {
                const Axis x = tri_axis(p, dims[a]);
                sm[q].i0[a] = x.i0; sm[q].i1[a] = x.i1; sm[q].t[a] = x.t;
                in[a] = x.m != 0.f;
            }
        }
        const unsigned c = !in[0] ? 1u : (!in[1] ? 2u : (!in[2] ? 3u : 0u));
        cls |= c << (2 * q);
    }
    STAMP(2);
    SLAB_EXIT(2, "v"(cls), "v"(dev), "v"(sm[0].i0[0]), "v"(sm[1].i0[1]), "v"(sm[2].i0[2]), "v"(sm[3].i1[0]), "v"(sm[3].t[0]), "v"(sm[0].t[1]), "v"(sm[1].t[2]));
    // ---- ONE barrier: which passes the block needs, the plane's measured thickness, and each pass's column box.  Record of a
    // wave: [0] classes present, [1] largest distance from the plane, [2 + 4 c ..] box (U min, U max, V min, V max) of class c
    // in the axes of ITS pass (class 0, the inside samples: the plane's minor axis; class 1 + a: axis a).
    // The box of ALL FOUR samples of a thread is taken for whichever class the thread has a member of (consecutive steps of
    // one ray: a box a few columns too large costs nothing; a select per sample per class costs instructions).
    unsigned present = 0;
#pragma unroll
    for (unsigned c = 0; c < 4; ++c) {
        bool any = false;
#pragma unroll
        for (int q = 0; q < kSPT; ++q) any |= ((cls >> (2 * q)) & 3u) == c && zb[q] != 0.f;
        int bx[4] = {0x7fffffff, -1, 0x7fffffff, -1};
        if (__builtin_amdgcn_ballot_w64(any) != 0ull) { // wave-uniform: a wave works out the boxes of the classes it has (one or two)
            present |= 1u << c;
            const int cU = c == 0 ? axU : (c == 3 ? 0 : 2), cV = c == 0 ? axV : (c == 2 ? 0 : 1); // SlabAxes of the pass
            auto pick = [](const int (&v)[3], int a) { return a == 0 ? v[0] : (a == 1 ? v[1] : v[2]); };
            int ulo = pick(sm[0].i0, cU), uhi = pick(sm[0].i1, cU), vlo = pick(sm[0].i0, cV), vhi = pick(sm[0].i1, cV);
#pragma unroll
            for (int q = 1; q < kSPT; ++q) { // (all four, not just the ends: a NaN or infinite direction is not monotone)
                ulo = min(ulo, pick(sm[q].i0, cU)); uhi = max(uhi, pick(sm[q].i1, cU));
                vlo = min(vlo, pick(sm[q].i0, cV)); vhi = max(vhi, pick(sm[q].i1, cV));
            }
            bx[0] = wave_reduce_minmax<true>(any ? ulo : 0x7fffffff);
            bx[1] = wave_reduce_minmax<false>(any ? uhi : -1);
            bx[2] = wave_reduce_minmax<true>(any ? vlo : 0x7fffffff);
            bx[3] = wave_reduce_minmax<false>(any ? vhi : -1);
        }
        if ((tid & 63) == 63) { // (always written: the passes read all four waves' records without asking who has what)
#pragma unroll
            for (int a = 0; a < 4; ++a) s_rec[wib][2 + 4 * c + a] = bx[a];
        }
    }
    const int devbits = wave_reduce_minmax<false>(__float_as_int(dev)); // dev >= 0 (NaN compares as a huge int: the patch then fails the test below)
    if ((tid & 63) == 63) {
        s_rec[wib][0] = (int)present;
        s_rec[wib][1] = devbits;
    }
    __syncthreads();
    present = 0;
    int dmax = 0;
    {
        int rp[kSW], rd[kSW];
#pragma unroll
        for (int wv = 0; wv < kSW; ++wv) { rp[wv] = s_rec[wv][0]; rd[wv] = s_rec[wv][1]; }
#pragma unroll
        for (int wv = 0; wv < kSW; ++wv) {
            present |= (unsigned)__builtin_amdgcn_readfirstlane(rp[wv]);
            dmax = max(dmax, __builtin_amdgcn_readfirstlane(rd[wv]));
        }
    }
    if (present == 0u) return true; // nothing to add (block-uniform)
    int L = 2;
    float c0b = 0.f;
    if (present & 1u) {
        const float devmax = __int_as_float(dmax);
        const float dU = (float)(axU == 0 ? dims[0] : (axU == 1 ? dims[1] : dims[2])), dV = (float)(axV == 0 ? dims[0] : (axV == 1 ? dims[1] : dims[2]));
        const float T = fabsf(c0) + fabsf(cu) * dU + fabsf(cv) * dV + 2.f;
        const float S = fabsf(cu) + fabsf(cv) + devmax + (1e-4f + T * 0x1p-19f); // + what float32 can err in h
        if (!(S < 0.5f * (float)(kSlabMaxL - 2))) return false; // (also NaN) not a plane within kSlabMaxL layers: the general path
        L = (int)ceilf(2.f * S) + 2;
        c0b = c0 - S;
    }
    SLAB_EXIT(3, "s"(L), "s"(c0b), "v"(cls), "v"(sm[0].i0[0]), "v"(sm[1].i0[1]), "v"(sm[2].i0[2]), "v"(sm[3].i1[0]), "v"(sm[3].t[0]));
#ifdef DIFFUS_STAMP
    const unsigned long long t_rec = STAMP_NOW();
#endif
    STAMP(6);
    unsigned long long prof[5] = {0ull, 0ull, 0ull, 0ull, 0ull}; // -DDIFFUS_STAMP: cycles in the accumulation / the flush, pass-chunks, tile entries
    // the passes in the order they run: classes 1, 2, 3 and the inside samples right after the faces of the plane's minor axis
    const unsigned last_class = (present & 1u) ? ((present >> (ax + 2)) ? (31u - (unsigned)__builtin_clz(present)) : 0u)
                                               : (31u - (unsigned)__builtin_clz(present));
    auto run_pass = [&](auto axis_, auto face_) {
        constexpr int a = decltype(axis_)::value;
        constexpr bool face = decltype(face_)::value;
        const unsigned want = face ? (unsigned)(a + 1) : 0u;
        unsigned member = 0;
#pragma unroll
        for (int q = 0; q < kSPT; ++q) member |= ((((cls >> (2 * q)) & 3u) == want && zb[q] != 0.f) ? 1u : 0u) << q;
        // the four waves' boxes of this class: four unconditional LDS reads in flight at once, merged on the scalar unit
        int rb[kSW][4];
#pragma unroll
        for (int wv = 0; wv < kSW; ++wv)
#pragma unroll
            for (int e = 0; e < 4; ++e) rb[wv][e] = s_rec[wv][2 + 4 * want + e];
        int u0 = 0x7fffffff, u1 = -1, v0 = 0x7fffffff, v1 = -1;
#pragma unroll
        for (int wv = 0; wv < kSW; ++wv) {
            u0 = min(u0, __builtin_amdgcn_readfirstlane(rb[wv][0])); u1 = max(u1, __builtin_amdgcn_readfirstlane(rb[wv][1]));
            v0 = min(v0, __builtin_amdgcn_readfirstlane(rb[wv][2])); v1 = max(v1, __builtin_amdgcn_readfirstlane(rb[wv][3]));
        }
        slab_pass<SAMPLER, a, face, CAPD>(A, tile, sm, zb, member, cu, cv, c0b, face ? 2 : L, u0, u1, v0, v1, want == last_class, tid, prof);
    };
    using T_ = std::true_type;
    using F_ = std::false_type;
    if (present >> 1 & 1u) run_pass(std::integral_constant<int, 0>{}, T_{});
    if ((present & 1u) && ax == 0) run_pass(std::integral_constant<int, 0>{}, F_{});
    if (present >> 2 & 1u) run_pass(std::integral_constant<int, 1>{}, T_{});
    if ((present & 1u) && ax == 1) run_pass(std::integral_constant<int, 1>{}, F_{});
    if (present >> 3 & 1u) run_pass(std::integral_constant<int, 2>{}, T_{});
    if ((present & 1u) && ax == 2) run_pass(std::integral_constant<int, 2>{}, F_{});
#ifdef DIFFUS_STAMP
    STAMPV(3, prof[0]); STAMPV(4, prof[1]); STAMP(5); STAMPV(7, ((prof[3] * 16ull + prof[2]) << 24) | ((prof[4] - t_rec) & 0xffffffull));
#endif
    return true;
}

template <int SAMPLER, int LAYOUT, int PM, bool SLAB = false>
__global__ __launch_bounds__(kSB, SLAB ? DIFFUS_SLAB_MIN_BLOCKS : DIFFUS_SC_MIN_BLOCKS) void scatter_patch_kernel(Args A, int ray_groups, int has_finish, int step_groups, int sg_mul)
{
    constexpr int CAP = SLAB ? kSlabCap : kTileCap; // tile entries (32-bit)
    static_assert(!SLAB || LAYOUT == DIFFUS_BRICKED, "the slab path scatters into a bricked gradient");
    // General (3-D) tile: 32-bit FIXED POINT with a per-patch power-of-two scale 2^fx chosen so that even all 1024 samples
    // landing on one voxel cannot overflow: (sum over the patch of |zbar|) * 2^fx < 2^30 (weights are <= 1, so no voxel
    // can receive more than that sum).  Quantum <= 2^-20 of the patch's largest contribution, typically 2^-23..2^-26;
    // integer adds commute, so a tile sum is bitwise reproducible.  (ds_add_f32 is ~194 cycles per wave-instruction on
    // gfx950, ds_add_u32 5-15: tools/lds_atomic_bench.hip.)  Planar patches -- every fan of the reference -- take the
    // 2-D double-precision tile of scatter_patch_planar instead.
    __shared__ __attribute__((aligned(16))) int tile[CAP];
    __shared__ int s_slab[SLAB ? kSW : 1][18]; // slab path: a record per wave (classes present, distance from the plane, a column box per class)
    __shared__ int s_wlo[kSW][3], s_whi[kSW][3]; // per-WAVE boxes (a wave = 64 / kPatchSteps * kSPT rays x kPatchSteps steps)
    __shared__ float s_sum[kSW];
    __shared__ int s_planar[kSW];
    __shared__ int s_box[kSW][4];
    __shared__ int s_live; // planar path: some wave of the block has a nonzero zbar
    constexpr int UNIT = (LAYOUT == DIFFUS_CANONICAL) ? 1 : kBrickFloats; // floats per tile unit
    constexpr bool kCanPlanar = (LAYOUT == DIFFUS_BRICKED); // nearest sampling too: one add per sample, one depth

    // the FIRST blocks of the launch, one per pose: median routing (start > 0) and d/dsource.  They need nothing from the
    // patches, and at the head of the grid their serial reductions run beside the first patches instead of after the last.
    // Grid: x = (pose, ray group) of one step group, y = step group (+ the finishing row in front): the block reads its
    // step group off blockIdx.y instead of dividing a linear index (uniform integer divisions are ~20 instructions each,
    // and every instruction of this kernel costs the same: fact 23).
    if (has_finish && blockIdx.y == 0) {
        if (blockIdx.x < (unsigned)A.P) pose_finish_block<SAMPLER, LAYOUT, false>(A, (int)blockIdx.x, reinterpret_cast<float *>(tile));
        return;
    }
    // block -> (row, pose, ray group) with the row SLOWEST, and the row -> step group mapping of decode() below: the blocks
    // in flight at any time are then spread over all poses and ray groups (round 1: all depths of a few neighbouring fans
    // at a time was 58 against 52 us) AND over all depths (round 4).  Within a row the XCD remap keeps a pose on one XCD.
    const int tid = threadIdx.x;
    int pose, nbase, sg_blk = 0; // (sg_blk: the block's step group)
    bool ray_ok;
    long w, w0;        // this thread's ray, the block's first ray (block-uniform)
    unsigned row_off;  // bytes from zbar[w0][0] to the thread's first sample
    auto decode = [&](unsigned bx, unsigned by) {
        // gridDim.x is a multiple of 8 (launch_scatter pads it): the hardware's linear block id by * gridDim.x + bx then
        // has the same residue mod 8 -- the XCD -- as bx in every row, so a pose sits on ONE XCD for all its step groups
        const unsigned Lb = xcd_remap(bx, gridDim.x);
        const int rg = Lb % ray_groups;
        pose = Lb / ray_groups;
        // Which step group a block takes: NOT simply its row.  Patches of one depth are alike -- near the apex small boxes
        // full of duplicate addresses (LDS-bound), at mid depth full tiles (bound by the L2's atomic rate in their flush),
        // beyond the volume a few border voxels -- and the dispatcher fills the chip row after row, so with sg = row every
        // generation of resident blocks queued for the same unit while the others idled.  sg = (row * m + 3 * pose) mod
        // groups, m coprime to the group count and about 7/16 of it (a bijection row -> sg for every pose), deals every
        // window of rows AND every row a mix of depths: 28.4 -> 25.5 us at 32 poses, 178.6 -> 145.2 us at 256 (round 4).
        const int row = (int)by - has_finish;
        const int sg = (int)(((unsigned)row * (unsigned)sg_mul + 3u * (unsigned)pose) % (unsigned)step_groups);
        sg_blk = sg;
        // thread -> ray (tid / 8) and 4 consecutive steps ((tid % 8) * 4 ..).  (Tried: a wave taking every 4th ray of the
        // patch instead of 8 adjacent ones, so that near the apex -- adjacent rays less than a voxel apart -- fewer lanes of
        // one LDS atomic share an address: 29.9 -> 30.5 us.)
        const int rl = tid / (kScSteps / kSPT);
        nbase = sg * kScSteps + (tid % (kScSteps / kSPT)) * kSPT;
        ray_ok = rg * kScRays + rl < A.R;
        w0 = (long)pose * A.R + (long)rg * kScRays;
        w = w0 + (ray_ok ? rl : 0);
        row_off = (__umul24((unsigned)(ray_ok ? rl : 0), (unsigned)A.N1) + (unsigned)nbase) * 4u; // N1 < 2^24, rl < 2^6
    };
    unsigned bx = blockIdx.x, by = blockIdx.y;
#ifdef DIFFUS_SC_EXIT
    if (DIFFUS_SC_EXIT == 0) return; // launch + dispatch floor
#endif
    if (xcd_remap(bx, gridDim.x) >= (unsigned)A.P * (unsigned)ray_groups) return; // padding block (block-uniform, before any barrier)
    decode(bx, by);

    STAMP(0);
#ifdef DIFFUS_SC_SALU_PAD // issue-rate probe (tools/): N extra scalar instructions per wave
    {
        int pad = 0;
        asm volatile(".rept %1\n s_add_u32 %0, %0, 1\n .endr" : "+s"(pad) : "n"(DIFFUS_SC_SALU_PAD));
        asm volatile("" :: "s"(pad));
    }
#endif
#ifdef DIFFUS_SC_VALU_PAD // issue-rate probe (tools/): N extra vector instructions per wave
    {
        int pad = tid;
        asm volatile(".rept %1\n v_add_u32 %0, %0, 1\n .endr" : "+v"(pad) : "n"(DIFFUS_SC_VALU_PAD));
        asm volatile("" :: "v"(pad));
    }
#endif
    Pose ps;
    if constexpr (PM == 0) { // float32 pose: the source from the scalar unit, the direction at a 32-bit offset from a scalar base
        const float *sp = (const float *)A.src + (long)pose * 3, *dp = (const float *)A.dirs + w0 * 3;
        const unsigned doff = (unsigned)(w - w0) * 12u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ps.sf[c] = sp[c];
            ps.df[c] = ldb_f32(dp, doff + 4u * c);
        }
        ps.pmode = 0;
    } else {
        load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
    }
#ifndef DIFFUS_SC_MERGE
#define DIFFUS_SC_MERGE 1 // step groups a leader takes at most (1, 2 or 4).  OFF: measured slower at 32 and 64 poses, DESIGN fact 42
#endif
    if constexpr (kCanPlanar && !SLAB && PM == 0 && DIFFUS_SC_MERGE > 1) {
        // The part of the fan that has left the volume: one block per up to four step groups (scatter_patch_planar_merged).
        // The rule, evaluated identically by the leader and by the blocks it relieves: a step group is "outside" when the ray
        // group's two edge rays are outside the slice at its first step; outside groups are taken in runs that end at multiples of
        // four -- the leader of a run is its first group (a multiple of four, or an outside group whose predecessor is not).
        if (A.fans_planar) {
            const float *sp = (const float *)A.src + (long)pose * 3;
            const int nr = min(kScRays, A.R - (int)(w0 - (long)pose * A.R));
            const float *da = (const float *)A.dirs + w0 * 3, *db = (const float *)A.dirs + (w0 + nr - 1) * 3; // block-uniform: scalar loads
            const float hx = (float)(A.G.d0 - 1), hy = (float)(A.G.d1 - 1);
            auto outside = [&](int sgq) -> bool {
                const float kf = (float)(A.start + sgq * kScSteps);
                const float ax = __fadd_rn(sp[0], __fmul_rn(kf, da[0])), ay = __fadd_rn(sp[1], __fmul_rn(kf, da[1]));
                const float bxp = __fadd_rn(sp[0], __fmul_rn(kf, db[0])), byp = __fadd_rn(sp[1], __fmul_rn(kf, db[1]));
                const bool oa = !(ax > 0.f && ax < hx && ay > 0.f && ay < hy), ob = !(bxp > 0.f && bxp < hx && byp > 0.f && byp < hy);
                return oa && ob;
            };
            if (__builtin_amdgcn_readfirstlane((int)outside(sg_blk))) {
                constexpr int kMg = DIFFUS_SC_MERGE;
                const bool leader = (sg_blk & (kMg - 1)) == 0 || !__builtin_amdgcn_readfirstlane((int)outside(sg_blk - 1));
                if (!leader) return; // covered by the leader of its run (block-uniform, before any barrier)
                // its run: the consecutive outside groups from here to the next multiple of four (a source outside the volume makes
                // "outside" true BEFORE the rays enter as well: a run must end where the rule stops relieving blocks)
                const int lim = min(kMg - (sg_blk & (kMg - 1)), step_groups - sg_blk);
                int ng = 1;
                while (ng < lim && __builtin_amdgcn_readfirstlane((int)outside(sg_blk + ng))) ++ng;
                if (ng > 1) {
                    scatter_patch_planar_merged<SAMPLER, PM, CAP>(A, reinterpret_cast<double *>(tile), s_box, s_planar, &s_live, ps, A.zbar + w0 * A.N1,
                                                                  row_off, ray_ok, nbase, ng, tid);
                    return;
                }
            }
        }
    }
    if constexpr (kCanPlanar) {
        // The launch that carries the slab path asks the block's FIRST ray before it tries the planar path (a scalar load, no
        // barrier): a fan that leaves the slice does so with every ray but, at most, its central one.
        bool try_planar = true;
        float dfl[6], zpre[kSPT]; // slab path: directions of the patch's first and last ray (block-uniform), the thread's zbar values
        if constexpr (SLAB) {
            // everything the slab path reads from memory is requested HERE, before the first wait: the zbar values (a vector
            // load), the two directions (scalar loads)
            const long wl = w0 + (long)min(kScRays, A.R - (int)(w0 - (long)pose * A.R)) - 1;
#pragma unroll
            for (int q = 0; q < kSPT; ++q) zpre[q] = (ray_ok && nbase + q < A.N1) ? ldb_f32(A.zbar + w0 * A.N1, row_off + 4u * q) : 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                dfl[c] = A.dir_f64 ? (float)((const double *)A.dirs)[w0 * 3 + c] : ((const float *)A.dirs)[w0 * 3 + c];
                dfl[3 + c] = A.dir_f64 ? (float)((const double *)A.dirs)[wl * 3 + c] : ((const float *)A.dirs)[wl * 3 + c];
            }
            try_planar = __builtin_amdgcn_readfirstlane((int)(dfl[2] == 0.f)) != 0;
        }
        if (try_planar) {
        if (scatter_patch_planar<SAMPLER, PM, CAP>(A, reinterpret_cast<double *>(tile), s_box, s_planar, &s_live, ps, A.zbar + w0 * A.N1, row_off, ray_ok, nbase, tid)) return;
#ifdef DIFFUS_SC_PLANAR_ONLY // timing probe: the general path compiled out (register budget of the planar path alone)
        return;
#endif
        __syncthreads(); // an oblique patch: every wave has read the records above before the general path rewrites them
        // The general path starts from scratch: laundering the block index keeps the compiler from carrying the patch
        // coordinates and the pose across the planar code above in registers it does not have (60 bytes of spills).
        asm volatile("" : "+s"(bx), "+s"(by));
        decode(bx, by);
        load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
        } else { // the tile the planar path would have cleared
            int4 *t4 = reinterpret_cast<int4 *>(tile);
#pragma unroll
            for (int e = 0; e < CAP / 4 / kSB; ++e) t4[e * kSB + tid] = make_int4(0, 0, 0, 0);
        }
        if constexpr (SLAB) { // a fan that leaves the slice: the height-field tile over its plane
            if (scatter_patch_slab<SAMPLER, PM, CAP>(A, reinterpret_cast<double *>(tile), s_slab, ps, dfl, zpre, nbase, tid)) return;
            __syncthreads(); // not a plane: every wave is done with the slab records before the general path starts over
            asm volatile("" : "+s"(bx), "+s"(by));
            decode(bx, by);
            load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);
        }
    }
    Cell cells[kSPT];
    float zb[kSPT];
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
#pragma unroll
    for (int q = 0; q < kSPT; ++q) { // issue the loads first ...
        int n = nbase + q;
        zb[q] = 0.f;
        if (ray_ok && n < A.N1) zb[q] = A.zbar[w * A.N1 + n];
    }
    // ... and clear the WHOLE tile while they are in flight (6 ds_write_b128 per thread, ~400 cycles per block):
    // clearing just the bounding box afterwards was a phase of its own with its own barrier (10 % of the block's time)
    {
        int4 *t4 = reinterpret_cast<int4 *>(tile);
#pragma unroll
        for (int e = 0; e < CAP / 4 / kSB; ++e) t4[e * kSB + tid] = make_int4(0, 0, 0, 0);
        static_assert(CAP % (4 * kSB) == 0, "tile clear assumes whole int4 passes");
    }
    // the cells need the pose only: they are worked out while the zbar loads are still in flight
#pragma unroll
    for (int q = 0; q < kSPT; ++q) cells[q] = cell_of<SAMPLER, PM>(A, ps, A.start + nbase + q);
    float zsum = 0.f;
#pragma unroll
    for (int q = 0; q < kSPT; ++q) {
        if (!finitef(zb[q])) zb[q] = 0.f;
        zsum += fabsf(zb[q]);
        if (zb[q] != 0.f) {
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                lo[a] = min(lo[a], tile_unit<LAYOUT>(cells[q].i0[a], a));
                hi[a] = max(hi[a], tile_unit<LAYOUT>(cells[q].i1[a], a));
            }
        }
    }
    STAMP(1);
    // per-wave bounding boxes and sum of |zbar|: DPP reductions (6 instructions each), one LDS record per wave
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        lo[a] = wave_reduce_minmax<true>(lo[a]);
        hi[a] = wave_reduce_minmax<false>(hi[a]);
    }
    zsum = wave_sum_to_lane63(zsum);
    const int wib = tid >> 6;
    if ((tid & 63) == 63) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_wlo[wib][a] = lo[a];
            s_whi[wib][a] = hi[a];
        }
        s_sum[wib] = zsum;
    }
    __syncthreads(); // also: the tile is clear
    STAMP(2);
    // The patch goes through the tile in ONE pass if its bounding box fits, else as 2 or 4 groups of waves (a wave =
    // 8 rays x 32 steps), each with its own box.  (The first version sent oversized patches -- 1.4 % of
    // them at config 3 -- to direct global atomics: those 58 blocks took 3x as long as the rest and were the kernel's
    // tail; a 64 KiB tile without any fallback ran 62 us against 70.)
    // Everything from here to the accumulation is BLOCK-UNIFORM bookkeeping: it is moved into SGPRs
    // (readfirstlane) so that it runs on the scalar unit, in 32-bit saturating arithmetic -- as 64-bit VALU
    // arithmetic repeated by all 256 threads it was 15 % of a block's time.
    int wlo[kSW][3], whi[kSW][3]; // the wave boxes, wave-uniform
#pragma unroll
    for (int wv = 0; wv < kSW; ++wv)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            wlo[wv][a] = __builtin_amdgcn_readfirstlane(s_wlo[wv][a]);
            whi[wv][a] = __builtin_amdgcn_readfirstlane(s_whi[wv][a]);
        }
    // box of the waves [w0, w0 + cnt); tile entries it needs: 0 for an empty group, CAP + 1 when it does not fit
    auto box_of = [&](int w0, int cnt, int (&l)[3], int (&b)[3]) -> int {
        unsigned v = UNIT;
        bool empty = false;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            int mn = 0x7fffffff, mx = -1;
#pragma unroll
            for (int wv = 0; wv < kSW; ++wv) {
                const bool in = wv >= w0 && wv < w0 + cnt;
                mn = in ? min(mn, wlo[wv][a]) : mn;
                mx = in ? max(mx, whi[wv][a]) : mx;
            }
            l[a] = mn;
            b[a] = mx - mn + 1;
            empty |= mx < 0;
            // only "> CAP" matters: saturate so that the 32-bit product cannot overflow
            const unsigned e = (unsigned)min(max(b[a], 0), 0x7fff);
            v = min(v, (unsigned)CAP + 1u) * e;
        }
        return empty ? 0 : (int)min(v, (unsigned)CAP + 1u);
    };
    static_assert(kSW == 4 || kSW == 8, "wave grouping below: 1, 2 or 4 groups of waves");
    int nsub = 1;
    int lb[3], bb[3];
    int vol_tile = box_of(0, kSW, lb, bb); // the whole patch: 85 % of the patches need nothing else
    if (vol_tile > CAP) {
        int l[3], b[3];
        nsub = 2;
        if (box_of(0, kSW / 2, l, b) > CAP || box_of(kSW / 2, kSW / 2, l, b) > CAP) nsub = 4;
    }
    // (s_sum total) * 2^fx in [2^28, 2^29): headroom for the rounding of each contribution
    float ztot = 0.f; // >= every single |zbar| of the patch (a float sum of non-negative terms is monotone)
#pragma unroll
    for (int wv = 0; wv < kSW; ++wv) ztot += __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(s_sum[wv])));
    const int fx = 29 - __builtin_amdgcn_frexp_expf(ztot);
    const int wpg = kSW / nsub; // waves per group
#pragma unroll 1
    for (int sp = 0; sp < nsub; ++sp) { // block-uniform trip count and branches
    if (nsub > 1) vol_tile = box_of(sp * wpg, wpg, lb, bb);
    if (vol_tile == 0) continue; // nothing to add in this group
    const bool mine = (wib / wpg) == sp;
    // keep the per-sample weights INSIDE the trip: hoisted out of this (almost always single-trip) loop they cost
    // 190 more registers and two of the three blocks per CU
#pragma unroll
    for (int q = 0; q < kSPT; ++q) {
        asm volatile("" : "+v"(zb[q]));
#pragma unroll
        for (int a = 0; a < 3; ++a) asm volatile("" : "+v"(cells[q].t[a]), "+v"(cells[q].i0[a]), "+v"(cells[q].i1[a]));
    }
    if (vol_tile > CAP) { // a single wave's strip does not fit (never seen with unit steps): direct atomics
        if (mine) {
#pragma unroll
            for (int q = 0; q < kSPT; ++q)
                if (zb[q] != 0.f)
                    for_each_corner<SAMPLER>(cells[q], zb[q], [&](int i, int j, int k, float v) {
                        if (v != 0.f) {
                            unsigned g = vox_off<LAYOUT>(A.G, i, j, k);
                            atomicAdd(A.gvol + g, v);
                            if (LAYOUT == DIFFUS_BRICKED && A.gtouched) A.gtouched[g >> 5] = 1;
                        }
                    });
        }
        continue;
    }
    const int l0 = lb[0], l1 = lb[1], l2 = lb[2], b0 = bb[0], b1 = bb[1], b2 = bb[2];
    const int nt = vol_tile;
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
    for (int q = 0; q < kSPT; ++q)
        if (mine && zb[q] != 0.f) {
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
    // e -> (i,j,k) by three divisions was 80 us of VALU at config 3): the flat unit index is split
    // with two exact float-reciprocal divisions, and only for units that hold something.
    const int b12 = b1 * b2, nunits = b0 * b12;
    const float rb2 = __frcp_rn((float)b2), rb12 = __frcp_rn((float)b12);
    constexpr int LPU = (UNIT == 1) ? 1 : UNIT;          // lanes per tile unit
    const int o = (UNIT == 1) ? 0 : (tid & (LPU - 1));   // float inside the brick
    const int msub = tid / LPU, mstep = kSB / LPU;
    constexpr int FU = 4; // tile reads in flight per thread: the LDS latency is paid once per 4 units, not per unit
    for (int q0 = msub; q0 < nunits; q0 += mstep * FU) {
        int v[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int q = q0 + u * mstep;
            v[u] = (q < nunits) ? tile[q * UNIT + o] : 0;
            if (nsub > 1 && v[u] != 0) tile[q * UNIT + o] = 0; // leave the tile clean for the next group
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int q = q0 + u * mstep;
            // a half-wave = one brick (bricked) -- skip the address arithmetic for all-zero bricks
            bool any = v[u] != 0;
            if (UNIT != 1) any = (unsigned)(__ballot(v[u] != 0) >> (tid & 32)) != 0u;
            if (any) {
                // exact float-reciprocal divisions: q < 2^14 and i * b12 <= q, m < b12 = b1 * b2
                const int i = __float2int_rz(((float)q + 0.5f) * rb12);
                const int m = q - i * b12;
                const int j = __float2int_rz(((float)m + 0.5f) * rb2);
                const int k = m - j * b2;
                unsigned g;
                if (LAYOUT == DIFFUS_CANONICAL)
                    g = ((unsigned)(l0 + i) * (unsigned)A.G.d1 + (unsigned)(l1 + j)) * (unsigned)A.G.d2 + (unsigned)(l2 + k);
                else
                    g = (((unsigned)(l0 + i) * (unsigned)A.G.nb1 + (unsigned)(l1 + j)) * (unsigned)A.G.nb2 + (unsigned)(l2 + k)) * kBrickFloats + (unsigned)o;
                // one plain, idempotent flag store per touched brick; no returning atomic (its latency
                // would sit on the flush path)
                if (LAYOUT == DIFFUS_BRICKED && A.gtouched && o == 0) A.gtouched[g >> 5] = 1;
                if (v[u] != 0) atomicAdd(A.gvol + g, ldexpf((float)v[u], -fx));
            }
        }
    }
    STAMP(5);
#ifdef DIFFUS_STAMP
    if (threadIdx.x == 0 && g_stamps) {
        g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = (unsigned long long)nt;
        g_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + 7] = (unsigned long long)nsub;
    }
#endif
    if (sp + 1 < nsub) __syncthreads(); // the next group adds into the tile this one has just read and cleared
    } // groups
}

} // namespace

namespace diffus {
int launch_scatter(const Args &A, int sampler, int layout, hipStream_t st)
{
    const int rgs = (A.R + kScRays - 1) / kScRays, sgs = (A.N1 + kScSteps - 1) / kScSteps;
    int sg_mul = 1; // row -> step group multiplier: odd, about 7/16 of the group count, coprime to it (see the kernel's decode())
    if (sgs >= 3) {
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        sg_mul = ((sgs * 7) / 16) | 1;
        while (sg_mul < sgs && gcd(sg_mul, sgs) != 1) sg_mul += 2;
        if (sg_mul >= sgs) sg_mul = 1;
    }
    const int fin = A.finish_in_scatter ? 1 : 0;
    // x padded to a multiple of 8 (at most 7 idle blocks per row): see the XCD note in the kernel's decode()
    const dim3 grid((unsigned)((((long)A.P * rgs + 7) / 8) * 8), (unsigned)(sgs + fin)); // sgs <= 2048 (DIFFUS_MAX_SAMPLES * SEGMENTS / patch steps)
    const bool f32 = !A.src_f64 && !A.dir_f64;
    const int glayout = layout == DIFFUS_PAIRED ? DIFFUS_BRICKED : layout; // the scatter only sees the gradient
    return dispatch_sl(sampler, glayout, [&](auto S_, auto L_) {
        constexpr int SM = decltype(S_)::value, LY = (decltype(L_)::value == DIFFUS_PAIRED) ? DIFFUS_BRICKED : decltype(L_)::value;
        // Fans the caller knows to be planar in dim 2 (DIFFUS_FANS_PLANAR), and canonical gradients: the 24 KiB tile at 6 blocks
        // per CU.  Otherwise the launch that also carries the slab path (36 KiB, 4 blocks per CU): planar patches in it take
        // the same planar path, ~10 % slower for the two blocks per CU it gives up; oblique ones no longer fall off a cliff.
        if constexpr (LY == DIFFUS_BRICKED) {
            if (!A.fans_planar) {
                if (f32)
                    hipLaunchKernelGGL((scatter_patch_kernel<SM, LY, 0, true>), grid, dim3(kSB), 0, st, A, rgs, fin, sgs, sg_mul);
                else
                    hipLaunchKernelGGL((scatter_patch_kernel<SM, LY, 1, true>), grid, dim3(kSB), 0, st, A, rgs, fin, sgs, sg_mul);
                return last_launch();
            }
        }
        if (f32)
            hipLaunchKernelGGL((scatter_patch_kernel<SM, LY, 0>), grid, dim3(kSB), 0, st, A, rgs, fin, sgs, sg_mul);
        else
            hipLaunchKernelGGL((scatter_patch_kernel<SM, LY, 1>), grid, dim3(kSB), 0, st, A, rgs, fin, sgs, sg_mul);
        return last_launch();
    });
}
} // namespace diffus

namespace {

// ----------------------------------------------------------------------------
// canonical <-> bricked conversion.  A block moves 4 x 4 x 64 voxels (32 bricks,
// 4 KiB): 16 canonical rows of 256 B on one side, 4 KiB contiguous on the other,
// through an LDS transpose so that both sides are coalesced.
constexpr int kConvZ = 64, kConvZThin = 8; // depths per block: whole volumes / thin sub-boxes (diffus_convert_volume_box)
template <bool TO_BRICKED, bool ACCUMULATE, int CZ = kConvZ>
__global__ __launch_bounds__(kBlock) void brick_convert_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                               Geom G, int zblk0 = 0, int by_0 = 0, int bx_0 = 0)
{
    __shared__ float t[16][CZ + 1];
    // (zblk0, by_0, bx_0): the first block of a sub-box conversion (diffus_convert_volume_box); 0 for a whole volume
    const int bz0 = ((int)blockIdx.x + zblk0) * (CZ / 2); // first brick along dim 2
    const int by = blockIdx.y + by_0, bx = blockIdx.z + bx_0;
    const int tid = threadIdx.x;
    const long brick0 = ((long)bx * G.nb1 + by) * G.nb2 + bz0;
    if (TO_BRICKED) {
        for (int e = tid; e < 16 * CZ; e += kBlock) {
            int row = e / CZ, zz = e - row * CZ;
            int x = bx * 4 + (row >> 2), y = by * 4 + (row & 3), z = bz0 * 2 + zz;
            t[row][zz] = (x < G.d0 && y < G.d1 && z < G.d2) ? in[((long)x * G.d1 + y) * G.d2 + z] : 0.f;
        }
        __syncthreads();
        for (int e = tid; e < 16 * CZ; e += kBlock) {
            int brick = e >> 5, off = e & 31;
            if (bz0 + brick < G.nb2) out[(brick0 + brick) * kBrickFloats + off] = t[off >> 1][brick * 2 + (off & 1)];
        }
    } else {
        for (int e = tid; e < 16 * CZ; e += kBlock) {
            int brick = e >> 5, off = e & 31;
            if (bz0 + brick < G.nb2) t[off >> 1][brick * 2 + (off & 1)] = in[(brick0 + brick) * kBrickFloats + off];
        }
        __syncthreads();
        for (int e = tid; e < 16 * CZ; e += kBlock) {
            int row = e / CZ, zz = e - row * CZ;
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

// Bricked gradient scratch -> canonical tensor.  Every touched brick (flag != 0) is added into (or stored to) the
// canonical tensor, ZEROED in the bricked buffer and its flag cleared, so the bricked buffer and the flags are all-zero
// again afterwards.  A fan touches a few thousand of the 524 288 bricks of a 256^3 volume: this replaces a 64 MiB memset
// plus a 128 MiB dense conversion per step.
// mode DIFFUS_FLUSH_PERSISTENT: `out` is a gradient tensor the caller keeps across steps and only this call writes.
// A brick stored this step gets flag 2 ("out holds last step's values, scratch is zero"); if the next step does not
// touch it again (the scatter overwrites the flag with 1) its voxels are zeroed in `out` and the flag cleared.  `out`
// therefore always equals the dense gradient of the latest step without ever being memset.
// mode DIFFUS_FLUSH_DENSE (this kernel): every voxel of `out` is written, one lane per brick.
__global__ __launch_bounds__(kBlock) void gradbuf_flush_dense_kernel(float *__restrict__ bricked, int *__restrict__ touched,
                                                                     float *__restrict__ out, Geom G, long nbricks)
{
    const int wib = threadIdx.x >> 6;
    const long w = (long)blockIdx.x * kWavesPerBlock + wib;
    const int lane = threadIdx.x & 63;
    const long b0 = w * kWave;
    if (b0 >= nbricks) return;
    const long mine = b0 + lane;
    int f = (mine < nbricks) ? touched[mine] : 0;
    {
        // EVERY voxel of `out` is written: a lane takes one brick, the wave 64 consecutive ones -- consecutive along dim 2,
        // so that each of a brick's 16 (x, y) rows is a 512-byte run of the canonical tensor across the wave.
        if (mine >= nbricks) return;
        if (f) touched[mine] = 0;
        const bool live = f == 1; // 2 = left by a PERSISTENT flush: the scratch is already zero there
        const unsigned um = (unsigned)mine, t = um / (unsigned)G.nb2, bz = um - t * (unsigned)G.nb2;
        const unsigned bx = t / (unsigned)G.nb1, by = t - bx * (unsigned)G.nb1;
        float *bsrc = bricked + mine * kBrickFloats;
        const int z = (int)bz * 2;
#pragma unroll 4
        for (int row = 0; row < 16; ++row) {
            const int x = (int)bx * 4 + (row >> 2), y = (int)by * 4 + (row & 3);
            float2 v = make_float2(0.f, 0.f);
            if (live) {
                v = *reinterpret_cast<const float2 *>(bsrc + row * 2);
                *reinterpret_cast<float2 *>(bsrc + row * 2) = make_float2(0.f, 0.f);
            }
            if (x < G.d0 && y < G.d1) {
                float *o = out + ((long)x * G.d1 + y) * G.d2 + z;
                if (z + 1 < G.d2 && !(G.d2 & 1)) {
                    *reinterpret_cast<float2 *>(o) = v;
                } else {
                    o[0] = v.x;
                    if (z + 1 < G.d2) o[1] = v.y;
                }
            }
        }
        return;
    }
}

// The sparse modes.  A wave reads the flags of 256 consecutive bricks (four coalesced loads), compacts the ids of the
// touched ones into LDS and walks them EIGHT per trip: 8 lanes per brick, a lane moving four floats = the z pairs of two
// neighbouring (x, y) rows.  (Rounds 2-3: 64 bricks per wave, two per trip -- 2048 blocks whose launch and flag reads were
// most of the kernel at 32 poses; 256 bricks per wave at two per trip was slower there, the serial walk four times longer.)
constexpr int kFlushBricks = 256;
template <bool VEC2> // VEC2: d2 even and `out` 8-byte aligned -- a z pair is one aligned 8-byte word of the canonical tensor
__global__ __launch_bounds__(kBlock) void gradbuf_flush_kernel(float *__restrict__ bricked, int *__restrict__ touched,
                                                               float *__restrict__ out, Geom G, long nbricks, int mode)
{
    __shared__ short s_list[kWavesPerBlock][kFlushBricks];
    const int wib = threadIdx.x >> 6;
    const long w = (long)blockIdx.x * kWavesPerBlock + wib;
    const int lane = threadIdx.x & 63;
    const long b0 = w * kFlushBricks;
    if (b0 >= nbricks) return;
    int f[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long mine = b0 + c * kWave + lane;
        f[c] = (mine < nbricks) ? touched[mine] : 0;
    }
    if (__ballot((f[0] | f[1] | f[2] | f[3]) != 0) == 0ull) return; // wave-uniform: nothing touched in these 256 bricks
    int cnt = 0;
    const unsigned long long below = (1ull << lane) - 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned long long m = __ballot(f[c] != 0);
        if (f[c]) {
            touched[b0 + c * kWave + lane] = (mode == DIFFUS_FLUSH_PERSISTENT && f[c] == 1) ? 2 : 0;
            // bit 8 marks a stale brick (nothing new this step: clear what the last step left in `out`)
            s_list[wib][cnt + __builtin_popcountll(m & below)] = (short)((c * kWave + lane) | (f[c] == 2 ? 256 : 0));
        }
        cnt += __builtin_popcountll(m);
    }
    wave_lds_sync();
    const int sub = lane & 7, grp = lane >> 3;
    // the lane's two rows inside a brick: x = sub / 2, y = 2 (sub % 2) and the next one; floats 4 sub .. 4 sub + 3
    const int xl = sub >> 1, yl = (sub & 1) * 2;
#pragma unroll 2
    for (int i = grp; i < cnt; i += 8) { // trips are independent: their loads overlap
        const int e = s_list[wib][i];
        const long brick = b0 + (e & 255);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!(e & 256)) { // uniform over the brick's 8 lanes
            float4 *src = reinterpret_cast<float4 *>(bricked + brick * kBrickFloats + sub * 4);
            v = *src;
            *src = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // 32-bit index arithmetic (a volume has fewer than 2^25 bricks)
        const unsigned ub = (unsigned)brick, t = ub / (unsigned)G.nb2, bz = ub - t * (unsigned)G.nb2;
        const unsigned bx = t / (unsigned)G.nb1, by = t - bx * (unsigned)G.nb1;
        const int x = (int)bx * 4 + xl, y = (int)by * 4 + yl, z = (int)bz * 2;
        if (x >= G.d0) continue;
        float *o = out + ((long)x * G.d1 + y) * G.d2 + z;
#pragma unroll
        for (int r = 0; r < 2; ++r, o += G.d2) {
            if (y + r >= G.d1) break;
            const float v0 = r ? v.z : v.x, v1 = r ? v.w : v.y;
            if (VEC2) { // z + 1 < d2 always: d2 is even
                float2 *o2 = reinterpret_cast<float2 *>(o);
                if (mode == DIFFUS_FLUSH_ACCUMULATE) {
                    const float2 q = *o2;
                    *o2 = make_float2(q.x + v0, q.y + v1);
                } else {
                    *o2 = make_float2(v0, v1);
                }
            } else {
                o[0] = (mode == DIFFUS_FLUSH_ACCUMULATE) ? o[0] + v0 : v0;
                if (z + 1 < G.d2) o[1] = (mode == DIFFUS_FLUSH_ACCUMULATE) ? o[1] + v1 : v1;
            }
        }
    }
}

// canonical -> PAIRED: a block writes the records of TWO neighbouring 4 x 4 column blocks for 128 depths (two runs of
// 128 x 160 B) from 4 x 9 canonical rows of 129 floats (the ninth column and the 129th depth are the neighbours the
// records repeat, clamped at the volume's edge), through LDS so that both sides are coalesced: 512-byte row reads,
// 16-byte stores.  256^3: 42 us = 5.5 TB/s of (volume read + records written).  (First version: one column block x 32
// depths per block, 132-byte row reads, a quarter of them the halo column: the L2-side fetch was 1.9x the volume and
// the kernel took 80 us; 64 depths per block: 46 us.)
#ifndef DIFFUS_PC_Z
#define DIFFUS_PC_Z 128
#endif
constexpr int kPcZ = DIFFUS_PC_Z, kPcZThin = 8, kPcCols = 9, kPcRows = 4 * kPcCols; // kPcZThin: thin sub-boxes (diffus_convert_volume_box)
template <int PZ = kPcZ>
__global__ __launch_bounds__(kBlock) void pair_convert_kernel(const float *__restrict__ in, float *__restrict__ out, Geom G,
                                                              int zblk0 = 0, int byp0 = 0, int bx_0 = 0)
{
    __shared__ float t[kPcRows][PZ + 2];
    // (zblk0, byp0, bx_0): the first block of a sub-box conversion (diffus_convert_volume_box); 0 for a whole volume
    const int z0 = ((int)blockIdx.x + zblk0) * PZ;
    const int by0 = ((int)blockIdx.y + byp0) * 2, bx = blockIdx.z + bx_0;
    const int tid = threadIdx.x;
    for (int e = tid; e < kPcRows * (PZ + 1); e += kBlock) {
        int row = e / (PZ + 1), zz = e - row * (PZ + 1); // row = (x & 3) * 9 + column 0..8
        int x = min(bx * 4 + row / kPcCols, G.d0 - 1), y = min(by0 * 4 + row % kPcCols, G.d1 - 1), z = min(z0 + zz, G.d2 - 1);
        t[row][zz] = in[((long)x * G.d1 + y) * G.d2 + z];
    }
    __syncthreads();
    // float4 = the (z, z + 1) pairs of two neighbouring columns of one x-row: 10 per record
    constexpr int V4 = kPairFloats / 4;
    for (int e = tid; e < 2 * PZ * V4; e += kBlock) {
        const int half = e / (PZ * V4), r = e - half * (PZ * V4);
        const int zz = r / V4, q = r - zz * V4;       // q-th float4 of the record: pairs 2q and 2q + 1
        const int by = by0 + half;
        if (by < G.nb1 && z0 + zz < G.d2) {
            const int p0 = 2 * q, p1 = 2 * q + 1;     // pair index = (x & 3) * 5 + column
            const int r0 = (p0 / 5) * kPcCols + half * 4 + p0 % 5, r1 = (p1 / 5) * kPcCols + half * 4 + p1 % 5;
            const float4 v = make_float4(t[r0][zz], t[r0][zz + 1], t[r1][zz], t[r1][zz + 1]);
            const long rec = ((long)bx * G.nb1 + by) * G.d2 + z0 + zz;
            *reinterpret_cast<float4 *>(out + rec * kPairFloats + 4 * q) = v;
        }
    }
}

} // namespace

extern "C" {

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
    if (accumulate < DIFFUS_FLUSH_STORE || accumulate > DIFFUS_FLUSH_DENSE) return DIFFUS_EINVAL;
    Geom G = make_geom(d0, d1, d2);
    const long nbricks = (long)(bricked_floats(d0, d1, d2) / kBrickFloats);
    if (reinterpret_cast<uintptr_t>(bricked) & 15) return DIFFUS_EINVAL; // a brick is read as 16-byte words
    const long per_wave = accumulate == DIFFUS_FLUSH_DENSE ? kWave : kFlushBricks;
    const long waves = (nbricks + per_wave - 1) / per_wave;
    const unsigned nblk = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (accumulate == DIFFUS_FLUSH_DENSE)
        hipLaunchKernelGGL(gradbuf_flush_dense_kernel, dim3(nblk), dim3(kBlock), 0, (hipStream_t)stream, bricked, touched, vol,
                           G, nbricks);
    else if (!(d2 & 1) && !(reinterpret_cast<uintptr_t>(vol) & 7))
        hipLaunchKernelGGL(gradbuf_flush_kernel<true>, dim3(nblk), dim3(kBlock), 0, (hipStream_t)stream, bricked, touched, vol,
                           G, nbricks, accumulate);
    else
        hipLaunchKernelGGL(gradbuf_flush_kernel<false>, dim3(nblk), dim3(kBlock), 0, (hipStream_t)stream, bricked, touched, vol,
                           G, nbricks, accumulate);
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
    dim3 grid((d2 + kPcZ - 1) / kPcZ, (G.nb1 + 1) / 2, (d0 + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535) return DIFFUS_EUNSUPPORTED;
    hipLaunchKernelGGL(pair_convert_kernel<kPcZ>, grid, dim3(kBlock), 0, (hipStream_t)stream, vol, paired, G, 0, 0, 0);
    return last_launch();
}

int diffus_convert_volume_box(const float *vol, int d0, int d1, int d2, int layout, float *converted, int x0, int x1,
                              int y0, int y1, int z0, int z1, diffus_stream_t stream)
{
    if (!vol || !converted || d0 <= 0 || d1 <= 0 || d2 <= 0) return DIFFUS_EINVAL;
    if (layout != DIFFUS_BRICKED && layout != DIFFUS_PAIRED) return DIFFUS_EINVAL;
    if (x0 < 0 || y0 < 0 || z0 < 0 || x1 > d0 || y1 > d1 || z1 > d2) return DIFFUS_EINVAL;
    if (x0 >= x1 || y0 >= y1 || z0 >= z1) return DIFFUS_OK; // an empty box
    Geom G = make_geom(d0, d1, d2);
    const int bx_lo = x0 >> 2, bx_hi = (x1 - 1) >> 2; // brick rows (4 voxels of dim 0): no layout repeats a dim-0 neighbour
    // depths per block: the whole-volume kernels' (128 / 64: long coalesced rows) for a deep box, 8 for a thin one -- a
    // slice of constant dim 2, the plane every fan of the reference lies in, is two depths of records
    auto launch = [&](auto kernel_for, int per_block, int zlo, int zhi, int by_lo, int by_hi) { // depths zlo..zhi; by_lo..by_hi in blocks
        const dim3 grid(zhi / per_block - zlo / per_block + 1, by_hi - by_lo + 1, bx_hi - bx_lo + 1);
        if (grid.y > 65535 || grid.z > 65535) return (int)DIFFUS_EUNSUPPORTED;
        kernel_for(grid, zlo / per_block, by_lo);
        return last_launch();
    };
    hipStream_t st = (hipStream_t)stream;
    if (layout == DIFFUS_PAIRED) {
        // A record (brick column by, depth z) repeats the first column of brick column by + 1 and the depth z + 1: the
        // records that hold a voxel of [y0, y1) x [z0, z1) are brick columns (y0 - 1) / 4 .. (y1 - 1) / 4, depths z0 - 1 .. z1 - 1
        const int by_lo = max(y0 - 1, 0) >> 2, by_hi = (y1 - 1) >> 2, zr_lo = max(z0 - 1, 0), zr_hi = z1 - 1;
        if (zr_hi - zr_lo < 2 * kPcZThin)
            return launch([&](dim3 g, int zb, int byp) { hipLaunchKernelGGL(pair_convert_kernel<kPcZThin>, g, dim3(kBlock), 0, st, vol, converted, G, zb, byp, bx_lo); },
                          kPcZThin, zr_lo, zr_hi, by_lo >> 1, by_hi >> 1);
        return launch([&](dim3 g, int zb, int byp) { hipLaunchKernelGGL(pair_convert_kernel<kPcZ>, g, dim3(kBlock), 0, st, vol, converted, G, zb, byp, bx_lo); },
                      kPcZ, zr_lo, zr_hi, by_lo >> 1, by_hi >> 1);
    }
    const int by_lo = y0 >> 2, by_hi = (y1 - 1) >> 2;
    if (z1 - z0 <= 2 * kConvZThin)
        return launch([&](dim3 g, int zb, int by) { hipLaunchKernelGGL((brick_convert_kernel<true, false, kConvZThin>), g, dim3(kBlock), 0, st, vol, converted, G, zb, by, bx_lo); },
                      kConvZThin, z0, z1 - 1, by_lo, by_hi);
    return launch([&](dim3 g, int zb, int by) { hipLaunchKernelGGL((brick_convert_kernel<true, false, kConvZ>), g, dim3(kBlock), 0, st, vol, converted, G, zb, by, bx_lo); },
                  kConvZ, z0, z1 - 1, by_lo, by_hi);
}

int diffus_brick_volume(const float *vol, int d0, int d1, int d2, float *bricked, diffus_stream_t stream)
{
    if (!vol || !bricked || d0 <= 0 || d1 <= 0 || d2 <= 0) return DIFFUS_EINVAL;
    Geom G = make_geom(d0, d1, d2);
    dim3 grid((G.nb2 + kConvZ / 2 - 1) / (kConvZ / 2), G.nb1, (d0 + 3) / 4);
    if (grid.y > 65535 || grid.z > 65535) return DIFFUS_EUNSUPPORTED;
    hipLaunchKernelGGL((brick_convert_kernel<true, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, vol, bricked, G, 0, 0, 0);
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
        hipLaunchKernelGGL((brick_convert_kernel<false, true>), grid, dim3(kBlock), 0, (hipStream_t)stream, bricked, vol, G, 0, 0, 0);
    else
        hipLaunchKernelGGL((brick_convert_kernel<false, false>), grid, dim3(kBlock), 0, (hipStream_t)stream, bricked, vol, G, 0, 0, 0);
    return last_launch();
}

#ifdef DIFFUS_STAMP
int diffus_debug_set_stamps(unsigned long long *p)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif

} // extern "C"
