// render_bwd.hip -- backward kernel (adjoint scan), median backward, d/dsource reduction, and diffus_render_bwd
#include "diffus_host.hpp"

namespace {

#ifdef DIFFUS_STAMP // diagnostic build only (tools/bwd_stamps.py): per-wave phase timestamps of the adjoint-scan kernel
__device__ unsigned long long *g_bwd_stamps = nullptr;
#define STAMPB(i)                                                                                                \
    do {                                                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
        if (g_bwd_stamps && lane == 0) g_bwd_stamps[((size_t)w * 2 + part) * 12 + (i)] = __builtin_readcyclecounter(); \
        __builtin_amdgcn_sched_barrier(0);                                                                       \
    } while (0)
#define STAMPRT(i)                                                                                               \
    do {                                                                                                         \
        if (g_bwd_stamps && lane == 0) g_bwd_stamps[((size_t)w * 2 + part) * 12 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#elif defined(DIFFUS_PHASE_MARKS) // static analysis only (tools/issue_model.py): phase boundaries as comments in the assembly
#define STAMPB(i)                                        \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        asm volatile("; DIFFUS_PHASE " #i);              \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)
#define STAMPRT(i) ((void)0)
#else
#define STAMPB(i) ((void)0)
#define STAMPRT(i) ((void)0)
#endif

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
#ifdef DIFFUS_COUNT_FAST_ONLY // static analysis only (tools/issue_model.py): the rare wave-uniform repair passes compiled out
#define DIFFUS_RARE(cond) false
#else
#define DIFFUS_RARE(cond) __builtin_expect((cond), 0)
#endif
// Waves per SIMD the C = 8 kernels of a bricked / paired volume are compiled for.  4 = a 128-VGPR budget: the kernel is
// VALU-issue-bound (its time is resident waves x issue cycles), sits at 126-138 registers depending on what the
// register allocator makes of the last change, and three waves per SIMD instead of four is a quarter of the chip idle in
// every generation of waves -- a handful of spilled dwords is the cheaper side of that.  (Canonical volumes -- eight
// address registers per sample in flight -- get THREE waves per SIMD, 164-168 registers: 48.8 -> 44.6 us at config 3;
// forced to 128 they spill 41 dwords: 63 us.)
#ifndef DIFFUS_BWD_MIN_WAVES
#define DIFFUS_BWD_MIN_WAVES 4
#endif
#ifndef DIFFUS_BWD_MIN_WAVES_CANONICAL
#define DIFFUS_BWD_MIN_WAVES_CANONICAL 3
#endif
#ifndef DIFFUS_SPLIT_MIN_WAVES // SPLIT kernels: 3 waves per SIMD, 139 VGPRs, no scratch.  (Forced to 4 waves = 128 VGPRs the
#define DIFFUS_SPLIT_MIN_WAVES 3 // compiler spills 15 dwords: one-pass scan 42.5 against 43.5 us at the config-5 shape, whole step 98.5 against 96 -- a wash; the spill-free build is kept)
#endif
// SEG = true: one 1024-sample segment of a longer ray (see diffus_render_bwd); only instantiated for C = 16.
// SPLIT = 2: the two waves of a 128-thread block take the two halves of ONE ray (64*C samples each) and exchange
// through LDS what the segmented launches exchange through the workspace -- the forward carry (P', last impedance
// sample) from the first half to the second, the adjoint carry (U', boundary term of zbar) back.  Used for
// 512 < N1 <= 1024 instead of one wave with 16 samples per lane: that kernel needs ~250 VGPRs (2 waves per SIMD) and
// a lane's serial sweeps are twice as long; two C = 8 waves need 126 each (4 per SIMD) and overlap their gathers,
// local products, scans and epilogues -- only the second half's adjoint has to finish before the first half's.
// FULL: every wave's row is exactly 64 C samples (N1 = 64 C, or 128 C for SPLIT) -- the benchmark shapes.  segN is then a
// compile-time constant and every "sample index < segN" test (three per sample: reflection, frame, zbar; a compare and a
// select each, 4.25 issue cycles apiece) folds away.
template <int C, int SAMPLER, int LAYOUT, bool GPOSE, int WPB, int PM, bool SEG = false, int SPLIT = 1, bool FULL = false>
__global__ __launch_bounds__(kWave *WPB, (SPLIT > 1 ? DIFFUS_SPLIT_MIN_WAVES : ((C == 8 && LAYOUT != DIFFUS_CANONICAL) ? DIFFUS_BWD_MIN_WAVES : ((C == 8) ? DIFFUS_BWD_MIN_WAVES_CANONICAL : 1)))) void render_bwd_kernel(Args A)
{
    static_assert(SPLIT == 1 || (SPLIT == 2 && WPB == 2 && !SEG), "SPLIT: one ray per block of two waves");
    __shared__ float s_c[5], s_u[4], s_zc, s_pg[6]; // SPLIT exchange: forward carry, adjoint carry, zbar boundary term, pose-gradient partials
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform: ray-derived addresses stay scalar
    const int part = (SPLIT > 1) ? wib : 0; // which half of the ray
    const long w = (SPLIT > 1) ? (long)xcd_remap(blockIdx.x, gridDim.x) : (long)xcd_remap(blockIdx.x, gridDim.x) * WPB + wib;
    if (w >= (long)A.P * A.R) return; // SPLIT: both waves of the block leave together
    const int seg0 = SEG ? A.seg0 : part * (kWave * C);
    const int segN = FULL ? kWave * C : (SEG ? A.segN : ((SPLIT > 1) ? min(A.N1 - seg0, kWave * C) : A.N1));
    // carries: per-ray records in the workspace (SEG) or the block's LDS records (SPLIT)
    const float *const cin = SEG ? A.cin + (A.cin ? w * 5 : 0) : ((SPLIT > 1 && part == 1) ? s_c : nullptr);
    const float *const cnext = SEG ? A.cnext + (A.cnext ? w * 5 : 0) : ((SPLIT > 1 && part == 0) ? s_c : nullptr);
    const float *const uin = SEG ? A.uin + (A.uin ? w * 4 : 0) : ((SPLIT > 1 && part == 0) ? s_u : nullptr);
    const float *const zcin = SEG ? A.zcin + (A.zcin ? w : 0) : ((SPLIT > 1 && part == 0) ? &s_zc : nullptr);
    float *const uout = SEG ? A.uout + (A.uout ? w * 4 : 0) : ((SPLIT > 1 && part == 1) ? s_u : nullptr);
    float *const zcout = SEG ? A.zcout + (A.zcout ? w : 0) : ((SPLIT > 1 && part == 1) ? &s_zc : nullptr);
    const bool accum_pose = SEG && A.accum_pose;
    // per wave: the INTERLEAVED <-> CHUNKED transpose buffer (64 C floats), reused afterwards -- together with a second
    // half -- to park d r / d Z_n and d r / d Z_{n-1} of every sample (lane-private 16-byte slots) from the reflection
    // step to the zbar step at the very end: the samples themselves then die before the scans (8 + 1 registers)
    __shared__ __attribute__((aligned(16))) float lds[WPB][2 * kWave * C];
    constexpr bool KEEP_GRAD = GPOSE && (C < 16); // C = 16: re-gather at the end instead of 12 KiB more LDS per wave
    __shared__ __attribute__((aligned(16))) float stash[KEEP_GRAD ? WPB : 1][KEEP_GRAD ? 3 * kWave * C : 4];
    const int lane = threadIdx.x & 63;
    const long pose = w / A.R;
    const int n0 = lane * C;
    float *wb = lds[wib];
    float *gst = stash[KEEP_GRAD ? wib : 0];

#ifdef DIFFUS_STAGGER_CYCLES // experiment (DESIGN "tried"): de-phase the blocks that share a CU so that one's gather overlaps another's scans
    {
        const unsigned slot = (blockIdx.x >> DIFFUS_STAGGER_SHIFT) & 3u;
        for (unsigned i = 0; i < slot * (DIFFUS_STAGGER_CYCLES / 640); ++i) __builtin_amdgcn_s_sleep(10); // 10 x 64 cycles
    }
#endif
    STAMPRT(10); // 100 MHz wall clock, the same on every XCD: when the wave started ...
    STAMPB(0);
    Pose ps;
    load_pose<PM>(ps, A.src, A.src_f64, A.dirs, A.dir_f64, pose, w);

    float zi[C], z[C], r[C], gb[C], tg[C];
    // The upstream-gradient row (and the target row of the fused loss) are requested FIRST: two 16-byte loads per lane
    // straight into the CHUNKED mapping, in flight while the gather computes its addresses -- loads complete in
    // order, so they are there by the time the first impedance value is consumed.
    if (A.mse != 2) {
        load_chunk<C>(A.gframe + w * A.N1 + seg0, n0, segN, gb);
    } else { // one-pass step: the frame is this kernel's own recomputed forward
#pragma unroll
        for (int j = 0; j < C; ++j) gb[j] = 0.f;
    }
    if (A.mse && A.target) {
        load_chunk<C>(A.target + w * A.N1 + seg0, n0, segN, tg);
    } else {
#pragma unroll
        for (int j = 0; j < C; ++j) tg[j] = 0.f;
    }
    {
        // The spatial gradient of every sample is needed once more, at the very end (pose gradient): it waits in LDS
        // (lane-private slots, conflict-free), not in 3C registers across the whole scan.
        float gi0[C], gi1[C], gi2[C];
        gather_interleaved<C, SAMPLER, LAYOUT, KEEP_GRAD, PM>(A, seg0, segN, ps, lane, zi, gi0, gi1, gi2);
        if (KEEP_GRAD) {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int at = inter_index<C>(lane, j); // (the bank swizzle of the transpose buffer: diffus_device.hpp)
                gst[0 * C * kWave + at] = gi0[j];
                gst[1 * C * kWave + at] = gi1[j];
                gst[2 * C * kWave + at] = gi2[j];
            }
        }
    }
    to_chunked<C>(wb, lane, zi, z);
    STAMPB(1);
    {
        // upstream gradient (loaded above), attenuation folded in
        if (A.mse == 1) { // fused loss: `gframe` is the forward's frame; dL/dframe = 2 s (frame - target), L += s (frame - target)^2
            float ssq = 0.f;
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const float dlt = gb[j] - tg[j]; // both 0 past the end of the row
                ssq = __builtin_fmaf(dlt, dlt, ssq);
                gb[j] = (2.f * A.loss_scale) * dlt;
            }
            ssq = wave_sum_to_lane63(ssq);
            if (lane == kWave - 1) {
                float *lp = A.loss_part + w * 2 + part;
                const float v = A.loss_scale * ssq;
                *lp = (SEG && A.accum_pose) ? *lp + v : v; // later-processed segments of a long ray add up
                if (SPLIT == 1 && !(SEG && A.accum_pose)) lp[1] = 0.f;
            }
        }
        if (A.mse != 2) { // (the one-pass step forms dL/dframe, attenuation included, in the forward sweep below)
            float att[C];
            chunk_attenuation<C>(A, seg0 + n0, att);
#pragma unroll
            for (int j = 0; j < C; ++j) gb[j] *= att[j];
        }
    }
    float zprev = lane_prev(z[C - 1], z[C - 1]);
    if (SPLIT > 1) { // the second half's first coefficient couples to the first half's last sample
        if (part == 0 && lane == kWave - 1) s_c[4] = z[C - 1];
        __syncthreads();
    }
    if (cin && lane == 0) zprev = cin[4]; // last sample of the previous segment
    const float medv = (A.start > 0) ? A.med[pose] : 0.f;
    {
        // d r / d Z_n = 2 Z_{n-1} / (Z_{n-1} + Z_n)^2 and d r / d Z_{n-1} = -2 Z_n / (...)^2 (reference :33) are formed HERE,
        // where the samples and the reciprocal of their sum are at hand, and wait in LDS for the zbar step
        float inv[C], dzn[C], dzp[C];
        reflect_chunk<C>(A, seg0, segN, n0, z, zprev, medv, r, inv);
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const float zp = (j == 0) ? zprev : z[j == 0 ? 0 : j - 1];
            dzn[j] = 2.f * zp * inv[j] * inv[j];
            dzp[j] = -2.f * z[j] * inv[j] * inv[j];
        }
        float *pk = wb + 2 * n0;
        if constexpr (C >= 4) {
            // lane-private 16-byte slots, WORD-major (slot t of lane l at float4 index t 64 + l): consecutive lanes, consecutive
            // words -- conflict-free where the lane-major order (a lane's 2 C floats contiguous: a 64-byte stride at C = 8) had
            // four lanes of every 16-lane pass on the same banks
            float4 *q = reinterpret_cast<float4 *>(wb) + lane;
#pragma unroll
            for (int t = 0; t < C / 4; ++t) {
                q[t * kWave] = make_float4(dzn[4 * t], dzn[4 * t + 1], dzn[4 * t + 2], dzn[4 * t + 3]);
                q[(C / 4 + t) * kWave] = make_float4(dzp[4 * t], dzp[4 * t + 1], dzp[4 * t + 2], dzp[4 * t + 3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                pk[j] = dzn[j];
                pk[C + j] = dzp[j];
            }
        }
    }
    STAMPB(2);

    // ---- forward recompute with exponent tracking ----
    Mat L = mat_identity();
    int lam = 0; // L' = L * 2^lam
#pragma unroll
    for (int j = 0; j < C; ++j) {
        L = mat_step(L, r[j]);
        if ((j & 3) == 3 || j == C - 1) lam -= mat_renorm(L); // every 4th step is enough (echo_chunk, FAST)
    }
    const Mat Lloc = L;   // normalised local product T_first..T_last
    const int lamloc = lam;
    int iota = lam;       // inclusive prefix exponent
    {
        // six scan rounds on the DPP path (diffus_device.hpp): matrix and exponent move together
        // (a lane without a source gets the identity and exponent 0, diffus_device.hpp mat_dpp_ident: no selects)
#define DIFFUS_ROUND(CTRL, RMASK, HAS, RN)                                     \
    {                                                                          \
        const Mat o = mat_dpp_ident<CTRL, RMASK>(L);                           \
        iota += dpp_get0<CTRL, RMASK>(iota);                                   \
        L = mat_mul(o, L);                                                     \
        if (RN) iota -= mat_renorm(L);                                         \
    }
        DIFFUS_SCAN_UP_ROUNDS(lane, DIFFUS_ROUND)
#undef DIFFUS_ROUND
    }
    Mat Pm = mat_lane_prev(L, mat_identity()); // exclusive prefix (lane 0: identity) ...
    int eps = lane_prev0(iota);                // ... and its exponent
    STAMPB(3);
    if (SPLIT > 1) { // the first half's total product (its inclusive scan in lane 63, normalised) is the second half's carry
        if (part == 0 && lane == kWave - 1) {
            s_c[0] = L.a; s_c[1] = L.b; s_c[2] = L.c; s_c[3] = L.d;
        }
        __syncthreads();
    }
    if (cin) { // segment > 0: P'_{seg0-1} of the carry-only forward pass precedes everything (its scale is exponent 0)
        Pm = mat_mul(Mat{cin[0], cin[1], cin[2], cin[3]}, Pm);
        eps -= mat_renorm(Pm);
    }

    Mat Pin[C];  // P'_{n-1} as used by this lane
    int ex[C];   // renorm exponent of step n
    float gu[C]; // gbar_n / d'_n
    float rho[C];
    int esum = 0;
    float ssq2 = 0.f; // one-pass step: this lane's share of sum((frame - target)^2)
    float att_c[C];   // attenuation of the lane's samples (one-pass step only)
    if (A.mse == 2) chunk_attenuation<C>(A, seg0 + n0, att_c);
    // FILTER = true is the loop as the arithmetic defines it.  FILTER = false leaves out everything that only matters when
    // an echo or a seed is not finite -- nan_to_num on the echo, the finite tests of the seed: a compare or a select each, 4.25
    // issue cycles apiece, six per sample -- and reports through `poison` (x * 0: 0 for finite x, NaN otherwise) whether
    // that was legitimate; if not for any lane of the wave, the filtering loop runs again from the same prefix (rare).
    // One-pass step: the frame row replaces the target row in tg as it is formed (no second array alive across the loop);
    // the repair pass loads the target row again.
    auto seeds_pass = [&](auto filter_, const Mat Pstart) -> float {
        constexpr bool FILTER = decltype(filter_)::value;
        float poison = 0.f;
        Pm = Pstart;
        esum = 0;
        ssq2 = 0.f;
        if (FILTER && A.mse == 2) { // the first pass has overwritten the target row with its frame
            if (A.target) {
                load_chunk<C>(A.target + w * A.N1 + seg0, n0, segN, tg);
            } else {
#pragma unroll
                for (int j = 0; j < C; ++j) tg[j] = 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            Pin[j] = Pm;
            Pm = mat_step(Pm, r[j]);
            ex[j] = ((j & 3) == 3 || j == C - 1) ? mat_renorm(Pm) : 0; // compile-time schedule: the zeros fold away
            esum += ex[j];
            const float rd = __builtin_amdgcn_rcpf(Pm.d);
            const float e = Pm.b * rd;
            const bool num = FILTER ? (e == e) : true; // echoes zeroed by nan_to_num are constants: no gradient through them
            float g = gb[j];
            if (A.mse == 2) {
                // ONE-PASS STEP (diffus_render_step_mse): the echo just recomputed IS the forward's (render_fwd_kernel's
                // arithmetic; only the association of the scan may differ -- chunk length, two waves per ray -- i.e. the
                // last bits), so the frame, the loss term and dL/dframe are formed right here and the separate forward
                // launch -- a second gather of every sample -- is not needed.
                const float att = att_c[j];
                const float fr = (n0 + j < segN) ? __fmul_rn(num ? e : 0.f, att) : 0.f;
                const float dlt = fr - tg[j]; // tg is 0 past the end of the row
                ssq2 = __builtin_fmaf(dlt, dlt, ssq2);
                g = (2.f * A.loss_scale) * dlt * att;
                tg[j] = fr; // the frame row, stored after the loop
            }
            const float q = g * rd; // (an echo zeroed by nan_to_num fails the finite test of e below: no select needed here)
            rho[j] = num ? e : 0.f;
            // Nothing but a finite seed may enter the adjoint chain (a NaN in U would wipe out every earlier step as well).
            // Two compares are enough: a non-finite r or P'_{n-1} makes P'_n -- hence e or q -- non-finite too, an infinite
            // echo fails the second test, and 0 * inf = NaN fails the first.  (Per-entry tests of r and P'_{n-1} here
            // were 6 more compares and a branch per sample.)
            if (FILTER) {
                gu[j] = (finitef(q) && finitef(e)) ? q : 0.f;
            } else {
                gu[j] = q;
                poison = __builtin_fmaf(q, 0.f, __builtin_fmaf(e, 0.f, poison));
            }
        }
        return poison;
    };
    {
        const float poison = seeds_pass(std::false_type{}, Pm);
        if (DIFFUS_RARE(__builtin_amdgcn_ballot_w64(poison != 0.f) != 0ull)) seeds_pass(std::true_type{}, Pin[0]); // wave-uniform; Pin[0] = the prefix the first pass started from
    }
    // One-pass step with DIFFUS_BWD_REPAIR_FRAME: a ray with |echo| > kEchoRecheck somewhere is ILL-CONDITIONED (9 rays of 8192 at
    // config 3) and its frame row is evaluated again in float64 by the per-pose epilogue (repair_ray_f64, diffus_device.hpp).
    // Not here: a wave that did it in place was this kernel's tail (+11 us for nine such waves); all that is left is a flag.
    if (A.rflag) { // (kernel argument: uniform over the grid; off by default)
        float emax = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) emax = fmaxf(emax, fabsf(rho[j])); // rho: the echoes, NaN -> 0
        const bool illc = A.mse == 2 && __builtin_amdgcn_ballot_w64(emax > kEchoRecheck) != 0ull; // wave-uniform
        if (lane == 0) {
            A.rflag[w * 2 + part] = illc ? 1 : 0;
            if (SPLIT == 1) A.rflag[w * 2 + 1] = 0; // (the second slot belongs to the second wave of a SPLIT ray)
        }
    }

    if (A.mse == 2) {
        if (A.frame) store_chunk<C, true>(A.frame + w * A.N1 + seg0, n0, segN, tg);
        ssq2 = wave_sum_to_lane63(ssq2);
        if (lane == kWave - 1) {
            float *lp = A.loss_part + w * 2 + part;
            *lp = A.loss_scale * ssq2;
            if (SPLIT == 1) lp[1] = 0.f;
        }
    }
    // exponent of this lane's last P' and the hop to the next lane's exclusive prefix
    const int elast = eps - esum;
    const int eps_next = lane_next0(eps);
    const int delta = (lane == kWave - 1) ? 0 : (eps_next - elast);
    STAMPB(4);

    // ---- lane-local affine map: A-part = sweep from U = 0 ----
    // Filters against non-finite values cost a compare and a select each (4.25 issue cycles apiece, tools/valu_issue_bench.hip)
    // on values that are finite in all but degenerate volumes.  Where the filter is the identity for finite input, the
    // wave first asks whether ANY of its values is non-finite -- x * 0 is 0 for a finite x and NaN otherwise, one
    // full-rate multiply-add per value into a running "poison" -- and only then runs the filtering pass (wave-uniform).
    float rr[C]; // r with non-finite steps cut (U is zero there anyway)
    {
        float poison = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            rr[j] = r[j];
            poison = __builtin_fmaf(r[j], 0.f, poison);
        }
        if (DIFFUS_RARE(__builtin_amdgcn_ballot_w64(poison != 0.f) != 0ull)) {
#pragma unroll
            for (int j = 0; j < C; ++j) rr[j] = finitef(r[j]) ? r[j] : 0.f;
        }
    }
    auto sweep = [&](Mat U, float *rbar) {
#pragma unroll
        for (int j = C - 1; j >= 0; --j) {
            Mat W;
            W.a = U.a;
            W.b = U.b + gu[j];
            W.c = U.c;
            W.d = U.d - gu[j] * rho[j];
            if ((j & 3) == 3 || j == C - 1) W = mat_scale(W, -ex[j]);
            if (rbar) {
                Mat Tb = mat_mul_at(Pin[j], W);
                rbar[j] = __builtin_fmaf(-4.f * r[j], Tb.a, Tb.b - Tb.c);
            }
            U = mat_mul_bt(W, mat_of_r(rr[j]));
        }
        return U;
    };
    Mat Aacc = sweep(Mat{0.f, 0.f, 0.f, 0.f}, nullptr);
    STAMPB(5);
    Mat Bn = Lloc;                    // normalised linear part
    int beta = delta - esum - lamloc; // B = Bn * 2^beta
    if (!mat_finite(Bn)) Bn = Mat{0.f, 0.f, 0.f, 0.f}; // a non-finite chunk passes nothing

    // ---- reverse inclusive scan of affine maps: G_l = F_l o F_{l+1} o ... o F_63 ----
    // Same six rounds mirrored: offsets 1, 2, 4, 8 towards HIGHER lanes inside the rows of 16 (row_shl), then the
    // first lane of the row above (lanes 16 / 48, then lane 32) broadcast downwards.  DPP has no broadcast in that
    // direction, so those two rounds read the lane through an SGPR (v_readlane_b32).
    {
        // (the linear part is rescaled every second round only, like the forward scan: exact either way)
        auto combine = [&](const Mat &oA, const Mat &oB, int ob, bool rn) {
            Mat t = mat_scale(mat_mul_bt(oA, Bn), beta);
            Aacc.a += t.a; Aacc.b += t.b; Aacc.c += t.c; Aacc.d += t.d;
            Bn = mat_mul(Bn, oB);
            beta = beta + ob + (rn ? mat_renorm(Bn) : 0); // B = Bn * 2^beta: a rescale of Bn by 2^-ex adds ex
        };
#define DIFFUS_ROUND_DOWN(N, RN)                                    \
    {                                                               \
        const Mat oA = mat_dpp_get<kDppRowShl + N>(Aacc);           \
        const Mat oB = mat_dpp_get<kDppRowShl + N>(Bn);             \
        const int ob = dpp_get<kDppRowShl + N>(beta);               \
        if ((lane & 15) + N < 16) combine(oA, oB, ob, RN);          \
    }
        DIFFUS_ROUND_DOWN(1, false)
        DIFFUS_ROUND_DOWN(2, true)
        DIFFUS_ROUND_DOWN(4, false)
        DIFFUS_ROUND_DOWN(8, true)
#undef DIFFUS_ROUND_DOWN
        {   // rows 0 and 2 take the row above them (its suffix sits in its first lane)
            const Mat a16 = mat_lane_bcast(Aacc, 16), b16 = mat_lane_bcast(Bn, 16);
            const Mat a48 = mat_lane_bcast(Aacc, 48), b48 = mat_lane_bcast(Bn, 48);
            const int e16 = __builtin_amdgcn_readlane(beta, 16), e48 = __builtin_amdgcn_readlane(beta, 48);
            const bool up = lane >= 32;
            if (!(lane & 16))
                combine(Mat{up ? a48.a : a16.a, up ? a48.b : a16.b, up ? a48.c : a16.c, up ? a48.d : a16.d},
                        Mat{up ? b48.a : b16.a, up ? b48.b : b16.b, up ? b48.c : b16.c, up ? b48.d : b16.d}, up ? e48 : e16, false);
        }
        {   // the lower half takes the upper half
            const Mat a32 = mat_lane_bcast(Aacc, 32), b32 = mat_lane_bcast(Bn, 32);
            const int e32 = __builtin_amdgcn_readlane(beta, 32);
            if (lane < 32) combine(a32, b32, e32, true);
        }
    }
    Mat Uin = mat_lane_next0(Aacc); // lane 63: nothing enters from above (0)
    STAMPB(6);
    if (SPLIT > 1 && part == 0) __syncthreads(); // the second half has published its adjoint carry (it arrives at the matching barrier below)
    if (uin) {
        // Adjoint entering from the next segment.  It was written relative to the scale of that segment's
        // carry-in P' (cnext); lane 63's final P' is the same matrix up to a power of two (idle samples
        // multiply by the identity), so the ratio of their largest entries gives the exact exponent hop.
        Mat Kn{cnext[0], cnext[1], cnext[2], cnext[3]};
        float mk = fmaxf(fmaxf(fabsf(Kn.a), fabsf(Kn.b)), fmaxf(fabsf(Kn.c), fabsf(Kn.d)));
        float mp = fmaxf(fmaxf(fabsf(Pm.a), fabsf(Pm.b)), fmaxf(fabsf(Pm.c), fabsf(Pm.d)));
        mp = lane_bcast(mp, kWave - 1);
        float ratio = mk / mp;
        Mat Uc{uin[0], uin[1], uin[2], uin[3]};
        if (finitef(ratio) && ratio > 0.f && mat_finite(Uc))
            Uc = mat_scale(Uc, (int)rintf(log2f(ratio)));
        else
            Uc = Mat{0.f, 0.f, 0.f, 0.f};
        // U entering lane l = G_{l+1}(Uc) = Aacc_{l+1} + Uc (B_{l+1} 2^beta_{l+1})^T
        const Mat Bs = mat_lane_next(Bn, mat_identity());
        const int bs = lane_next0(beta);
        Mat t = mat_scale(mat_mul_bt(Uc, Bs), bs);
        Uin.a += t.a; Uin.b += t.b; Uin.c += t.c; Uin.d += t.d;
        if (lane == kWave - 1) Uin = Uc;
    }
    Uin = mat_scale(Uin, delta);

    float rbar[C];
    const Mat Uout = sweep(Uin, rbar);
    STAMPB(7);
    if (uout && lane == 0) { // relative to the scale of this segment's carry-in P'
        uout[0] = Uout.a; uout[1] = Uout.b; uout[2] = Uout.c; uout[3] = Uout.d;
    }

    // ---- rbar -> zbar (d r / d Z of reference :33) ----
    float zbar[C];
#pragma unroll
    for (int j = 0; j < C; ++j) zbar[j] = 0.f;
    float carry = 0.f, gmed_lane = 0.f;
    float dzn[C], dzp[C]; // d r / d Z_n, d r / d Z_{n-1}: parked by the reflection step (same lane wrote them: no synchronisation)
    {
        const float *pk = wb + 2 * n0;
        if constexpr (C >= 4) {
            const float4 *q = reinterpret_cast<const float4 *>(wb) + lane; // word-major slots, as written
#pragma unroll
            for (int t = 0; t < C / 4; ++t) {
                const float4 v = q[t * kWave], u = q[(C / 4 + t) * kWave];
                dzn[4 * t] = v.x; dzn[4 * t + 1] = v.y; dzn[4 * t + 2] = v.z; dzn[4 * t + 3] = v.w;
                dzp[4 * t] = u.x; dzp[4 * t + 1] = u.y; dzp[4 * t + 2] = u.z; dzp[4 * t + 3] = u.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                dzn[j] = pk[j];
                dzp[j] = pk[C + j];
            }
        }
    }
    // (FILTER = false: the pass without the per-contribution finite tests; its poison says whether it may stand)
    auto zbar_pass = [&](auto filter_) -> float {
        constexpr bool FILTER = decltype(filter_)::value;
        float poison = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) zbar[j] = 0.f;
#pragma unroll
        for (int j = 0; j < C; ++j) {
            int n = seg0 + n0 + j;
            bool live = (n >= 1 && n0 + j < segN);
            float rb = live ? rbar[j] : 0.f;
            if (n == 1 && A.start > 0) { // the first kept coefficient is the per-pose median: its gradient goes there
                gmed_lane = finitef(rb) ? rb : 0.f;
                rb = 0.f;
            }
            float c1 = rb * dzn[j], c0 = rb * dzp[j];
            if (FILTER) {
                if (!finitef(c1)) c1 = 0.f; // drop non-finite contributions (a non-finite rbar makes both of them so)
                if (!finitef(c0)) c0 = 0.f;
            } else {
                poison = __builtin_fmaf(c1, 0.f, __builtin_fmaf(c0, 0.f, poison));
            }
            zbar[j] += c1;
            if (j == 0)
                carry = c0;
            else
                zbar[j == 0 ? 0 : j - 1] += c0;
        }
        return poison;
    };
    {
        const float poison = zbar_pass(std::false_type{});
        if (DIFFUS_RARE(__builtin_amdgcn_ballot_w64(poison != 0.f) != 0ull)) zbar_pass(std::true_type{}); // wave-uniform, rare
    }
    const float cnb = lane_next0(carry);
    if (lane != kWave - 1) zbar[C - 1] += cnb;
    if (zcout && lane == 0) zcout[0] = carry; // belongs to the last sample of the previous segment
    if (SPLIT > 1 && part == 1) __syncthreads(); // adjoint carry and boundary term are out: releases the first half
    if (zcin) {
        const int last = segN - 1;
        const float zc = zcin[0];
        if (lane == last / C) {
#pragma unroll
            for (int j = 0; j < C; ++j)
                if (j == last % C) zbar[j] += zc;
        }
    }
    // this ray's share of d/d median: sample 1 lives in lane 0 (C >= 2) of the first segment
    if (A.start > 0 && seg0 == 0 && lane == 0) A.gmed[w] = gmed_lane;

    // ---- hand zbar to the scatter kernel, reduce the pose gradient: both from the CHUNKED mapping ----
    // The volume scatter is a separate launch (scatter_patch_kernel): its thread <-> sample
    // mapping is chosen for LDS privatisation, not for the scan.
    if (A.zbar) store_chunk<C, true>(A.zbar + w * A.N1 + seg0, n0, segN, zbar);
    STAMPB(8);
    if (GPOSE) {
        float gs0 = 0.f, gs1 = 0.f, gs2 = 0.f, gd0 = 0.f, gd1 = 0.f, gd2 = 0.f;
        // the spatial gradients parked in LDS by the gather: component c of sample n sits at gst[c * 64 C + n], so
        // a lane's C consecutive samples are contiguous there (ds_read_b128)
        float q0[C], q1[C], q2[C];
        if (KEEP_GRAD) {
            wave_lds_sync();
            auto read_row = [&](int c, float (&q)[C]) { // a lane's C consecutive floats: 16-byte LDS reads
                if constexpr (C >= 4) {
#pragma unroll
                    for (int t = 0; t < C / 4; ++t) {
                        const float4 v = *reinterpret_cast<const float4 *>(gst + c * C * kWave + chunk_word<C>(lane, t));
                        q[4 * t] = v.x; q[4 * t + 1] = v.y; q[4 * t + 2] = v.z; q[4 * t + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < C; ++j) q[j] = gst[c * C * kWave + n0 + j];
                }
            };
            read_row(0, q0); read_row(1, q1); read_row(2, q2);
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int n = n0 + j;
            const int k = A.start + seg0 + n;
            float g0, g1, g2;
            if (KEEP_GRAD) {
                g0 = q0[j]; g1 = q1[j]; g2 = q2[j];
            } else {
                g0 = g1 = g2 = 0.f;
                if (n < segN && zbar[j] != 0.f) {
                    TriSample sm = tri_sample<LAYOUT, true>(A.vol, A.G, ray_point<PM>(ps, 0, k), ray_point<PM>(ps, 1, k),
                                                            ray_point<PM>(ps, 2, k));
                    g0 = sm.g0; g1 = sm.g1; g2 = sm.g2;
                }
            }
            // Nothing flows where zbar is 0 -- past the end of the row (rbar is masked there and non-finite products are
            // dropped, so zbar IS 0) and where the gradient was dropped -- whatever the spatial gradient is there (it may be
            // NaN next to NaN voxels): v_mul_legacy_f32 (0 * anything = 0, an IEEE product otherwise; zbar is finite) does
            // that in the multiply itself.  It was a compare and three selects per sample, 4.25 issue cycles each.
            const float zb = zbar[j];
            const float kf = (float)k;
            const float a0 = mul_legacy(zb, g0), a1 = mul_legacy(zb, g1), a2 = mul_legacy(zb, g2);
            gs0 += a0; gs1 += a1; gs2 += a2;
            gd0 = __builtin_fmaf(kf, a0, gd0);
            gd1 = __builtin_fmaf(kf, a1, gd1);
            gd2 = __builtin_fmaf(kf, a2, gd2);
        }
        gs0 = wave_sum_to_lane63(gs0); gs1 = wave_sum_to_lane63(gs1); gs2 = wave_sum_to_lane63(gs2);
        gd0 = wave_sum_to_lane63(gd0); gd1 = wave_sum_to_lane63(gd1); gd2 = wave_sum_to_lane63(gd2);
        if (SPLIT > 1) { // one ray, two waves: the second half hands its partial sums to the first
            if (part == 1 && lane == kWave - 1) {
                s_pg[0] = gs0; s_pg[1] = gs1; s_pg[2] = gs2; s_pg[3] = gd0; s_pg[4] = gd1; s_pg[5] = gd2;
            }
            __syncthreads();
            if (part == 1) return;
            gs0 += s_pg[0]; gs1 += s_pg[1]; gs2 += s_pg[2]; gd0 += s_pg[3]; gd1 += s_pg[4]; gd2 += s_pg[5];
        }
        if (lane == kWave - 1) { // the DPP ladder leaves the wave's total in its last lane
            if (accum_pose) { // later-processed segment of a long ray: add to the partial sums
                if (A.gsrc_part) {
                    gs0 += A.gsrc_part[w * 3 + 0]; gs1 += A.gsrc_part[w * 3 + 1]; gs2 += A.gsrc_part[w * 3 + 2];
                }
                if (A.gdirs) {
                    gd0 += A.gdirs[w * 3 + 0]; gd1 += A.gdirs[w * 3 + 1]; gd2 += A.gdirs[w * 3 + 2];
                }
            }
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
    STAMPB(9);
    STAMPRT(11); // ... and ended
}

// One block per pose: what pose_finish_block does, as a launch of its own (no scatter launch to ride on).
template <int SAMPLER, int GLAYOUT>
__global__ __launch_bounds__(kBlock) void pose_finish_kernel(Args A)
{
    __shared__ float sm[kWavesPerBlock * DIFFUS_MAX_SAMPLES];
    pose_finish_block<SAMPLER, GLAYOUT, true>(A, blockIdx.x, sm);
}

template <int SM, int LY, bool GPOSE, int PM>
int launch_bwd_p(const Args &A, hipStream_t st)
{
    const long waves = (long)A.P * A.R;
    const unsigned nblk = (unsigned)((waves + kWavesPerBlock - 1) / kWavesPerBlock);
    if (A.N1 > DIFFUS_MAX_SAMPLES) { // a segment of a long ray
        hipLaunchKernelGGL((render_bwd_kernel<16, SM, LY, GPOSE, 1, PM, true>), dim3((unsigned)waves), dim3(kWave), 0, st, A);
        return last_launch();
    }
    switch (chunk_for(A.N1)) {
    case 2: hipLaunchKernelGGL((render_bwd_kernel<2, SM, LY, GPOSE, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    case 4: hipLaunchKernelGGL((render_bwd_kernel<4, SM, LY, GPOSE, kWavesPerBlock, PM>), dim3(nblk), dim3(kBlock), 0, st, A); break;
    // (256 < N1 <= 512 as two waves of 4 samples per lane was measured too: 46.5 against 37 us at config 3 -- the split
    // pays only where it lifts the single wave out of the 2-waves-per-SIMD register regime; for launches of ONE to four frames,
    // where most SIMDs are empty, it changes nothing either: 8.3 us with or without, the event-timed floor of a launch is 6.5)
    case 8: {
#ifndef DIFFUS_BWD_WPB8
#define DIFFUS_BWD_WPB8 kWavesPerBlock
#endif
        constexpr int W8 = DIFFUS_BWD_WPB8; // waves (= rays) per block of the C = 8 kernel
        const dim3 grid((unsigned)((waves + W8 - 1) / W8));
        if (A.N1 == 8 * kWave) // full rows: the instantiation without the end-of-row tests
            hipLaunchKernelGGL((render_bwd_kernel<8, SM, LY, GPOSE, W8, PM, false, 1, true>), grid, dim3(kWave * W8), 0, st, A);
        else
            hipLaunchKernelGGL((render_bwd_kernel<8, SM, LY, GPOSE, W8, PM>), grid, dim3(kWave * W8), 0, st, A);
        break;
    }
    default: // 512 < N1 <= 1024: two waves of 8 samples per lane share a ray (SPLIT), one ray per 128-thread block
        if (A.N1 == 16 * kWave)
            hipLaunchKernelGGL((render_bwd_kernel<8, SM, LY, GPOSE, 2, PM, false, 2, true>), dim3((unsigned)waves), dim3(2 * kWave), 0, st, A);
        else
            hipLaunchKernelGGL((render_bwd_kernel<8, SM, LY, GPOSE, 2, PM, false, 2>), dim3((unsigned)waves), dim3(2 * kWave), 0, st, A);
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

// diffus_render_bwd (mse = false: `gframe` is dL/dframe) and diffus_render_bwd_mse (mse = true: `gframe` is the frame)
int render_bwd_impl(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                    const void *dirs, int dirs_dtype, int P, int R, int S, int start, float alpha, int sampler,
                    const float *gframe, int mse, const float *target, float loss_scale, float *loss, float *frame_out,
                    float *gvol, int *gvol_touched, float *gsrc, float *gdirs, int stages,
                    void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    const bool grad_bricked = (layout & DIFFUS_GRAD_BRICKED) != 0; // the gradient's layout, decoupled from the volume's
    const bool fans_planar = (layout & DIFFUS_FANS_PLANAR) != 0;   // a hint for the scatter launch (include/diffus_hip.h)
    const bool repair = (stages & DIFFUS_BWD_REPAIR_FRAME) != 0;   // one-pass step: float64 frame rows for ill-conditioned rays
    stages &= ~DIFFUS_BWD_REPAIR_FRAME;
    layout &= ~(DIFFUS_GRAD_BRICKED | DIFFUS_FANS_PLANAR);
    const int glayout = (layout == DIFFUS_PAIRED || grad_bricked) ? DIFFUS_BRICKED : layout;
    int rc = check_common(vol, d0, d1, d2, src, src_dtype, dirs, dirs_dtype, P, R, S, start, sampler, layout, true);
    if (rc) return rc;
    if (!gframe && mse != 2) return DIFFUS_EINVAL;
    if (stages < 1 || stages > (DIFFUS_BWD_ALL | DIFFUS_BWD_KEEP_MEDIAN) || !(stages & DIFFUS_BWD_ALL)) return DIFFUS_EINVAL;
    if (!gvol && !gsrc && !gdirs && !(mse && loss) && !(mse == 2 && frame_out)) return DIFFUS_OK;
    const bool do_scan = stages & DIFFUS_BWD_SCAN, do_scatter = stages & DIFFUS_BWD_SCATTER;
    Workspace ws = carve(workspace, P, R, S - start);
    if (!workspace || workspace_bytes < ws.bytes) return DIFFUS_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const bool pose = sampler == DIFFUS_TRILINEAR && (gsrc || gdirs);
    if (sampler == DIFFUS_NEAREST && do_scan) { // integer indices: no pose gradient (reference :754-758)
        if (gsrc && hipMemsetAsync(gsrc, 0, sizeof(float) * (size_t)P * 3, st) != hipSuccess) return DIFFUS_ELAUNCH;
        if (gdirs && hipMemsetAsync(gdirs, 0, sizeof(float) * (size_t)P * R * 3, st) != hipSuccess) return DIFFUS_ELAUNCH;
    }
    if (sampler == DIFFUS_NEAREST && !gvol && !(mse && loss) && !(mse == 2 && frame_out)) return DIFFUS_OK;
    Args A = make_args(vol, d0, d1, d2, layout, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, ws);
    A.gframe = gframe;
    A.mse = mse;
    A.fans_planar = fans_planar;
    A.vol_layout = layout;
    A.rflag = (mse == 2 && ws.nseg == 1 && repair) ? ws.rflag : nullptr; // one-pass step, rays of one launch, on request: ill-conditioned rays are repaired in float64
    A.frame = (mse == 2) ? frame_out : nullptr;
    A.target = target;
    A.loss_scale = loss_scale;
    A.loss_part = ws.loss_part;
    A.loss_out = (mse && do_scan) ? loss : nullptr;
    A.gvol = gvol;
    A.gtouched = (gvol && glayout != DIFFUS_CANONICAL) ? gvol_touched : nullptr;
    A.zbar = gvol ? ws.zbar : nullptr;
    A.gsrc_part = (pose && gsrc) ? ws.gsrc_part : nullptr;
    A.gdirs = pose ? gdirs : nullptr;
    if (do_scan) {
        if (start > 0 && !(stages & DIFFUS_BWD_KEEP_MEDIAN)) { // recompute the median (and zero gmed)
            rc = launch_median(A, sampler, layout, st);
            if (rc) return rc;
        }
        if (ws.nseg == 1) {
            rc = launch_bwd(A, sampler, layout, pose, st);
            if (rc) return rc;
        } else {
            // Long rays.  Pass 1 runs the forward kernel segment by segment to leave every segment's carry-in
            // (P' and the last impedance sample) in the workspace; pass 2 runs the adjoint scan from the last
            // segment to the first, chained through U' and the boundary term of zbar.
            const size_t wr = (size_t)P * R;
            for (int s = 0; s + 1 < ws.nseg; ++s) {
                Args F = A;
                F.frame = nullptr; F.idx = nullptr;
                F.seg0 = s * DIFFUS_MAX_SAMPLES; F.segN = DIFFUS_MAX_SAMPLES;
                F.cin = s ? ws.carry + (size_t)(s - 1) * wr * 5 : nullptr;
                F.cout = ws.carry + (size_t)s * wr * 5;
                rc = diffus::launch_fwd(F, sampler, layout, st);
                if (rc) return rc;
            }
            for (int s = ws.nseg - 1; s >= 0; --s) {
                Args B = A;
                const bool last = s == ws.nseg - 1;
                B.seg0 = s * DIFFUS_MAX_SAMPLES;
                B.segN = last ? A.N1 - B.seg0 : DIFFUS_MAX_SAMPLES;
                B.cin = s ? ws.carry + (size_t)(s - 1) * wr * 5 : nullptr;
                B.cnext = last ? nullptr : ws.carry + (size_t)s * wr * 5;
                B.uin = last ? nullptr : ws.ucarry + (size_t)((s + 1) & 1) * wr * 4;
                B.uout = s ? ws.ucarry + (size_t)(s & 1) * wr * 4 : nullptr;
                B.zcin = last ? nullptr : ws.zcarry + (size_t)((s + 1) & 1) * wr;
                B.zcout = s ? ws.zcarry + (size_t)(s & 1) * wr : nullptr;
                B.accum_pose = !last;
                rc = launch_bwd(B, sampler, layout, pose, st);
                if (rc) return rc;
            }
        }
    }
    // Per-pose epilogue (pose_finish_block): the median's gradient goes to the ray that supplied it (start > 0) and the
    // per-ray d/dsource partials are summed.  It rides along as P extra blocks of the scatter launch when that launch
    // follows in this call, else it is one launch of its own.
    const bool finish = do_scan && (start > 0 || (pose && gsrc) || A.loss_out || A.rflag);
    A.gsrc_out = (pose && gsrc) ? gsrc : nullptr;
    // (with the float64 repair the epilogue is a launch of its own: inside the scatter kernel the repair's code and scalar registers
    // cost the patch path 1.4 us -- measured: SGPR spills and 20 more instructions in its prologue --, whether a ray needs them or not)
    A.finish_in_scatter = finish && gvol && do_scatter && !A.rflag;
    if (gvol && do_scatter) {
        rc = diffus::launch_scatter(A, sampler, glayout, st);
        if (rc) return rc;
    }
    if (finish && !A.finish_in_scatter) {
        rc = dispatch_sl(sampler, glayout, [&](auto S_, auto L_) {
            constexpr int GL = (decltype(L_)::value == DIFFUS_PAIRED) ? DIFFUS_BRICKED : decltype(L_)::value;
            hipLaunchKernelGGL((pose_finish_kernel<decltype(S_)::value, GL>), dim3(P), dim3(kBlock), 0, st, A);
            return last_launch();
        });
        if (rc) return rc;
    }
    return DIFFUS_OK;
}

} // namespace

extern "C" {

#ifdef DIFFUS_STAMP
int diffus_debug_set_bwd_stamps(unsigned long long *p)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(g_bwd_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif

int diffus_render_bwd(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                      const void *dirs, int dirs_dtype, int P, int R, int S, int start, float alpha, int sampler,
                      const float *gframe, float *gvol, int *gvol_touched, float *gsrc, float *gdirs, int stages,
                      void *workspace, size_t workspace_bytes, diffus_stream_t stream)
{
    return render_bwd_impl(vol, d0, d1, d2, layout, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, sampler, gframe,
                           0, nullptr, 0.f, nullptr, nullptr, gvol, gvol_touched, gsrc, gdirs, stages, workspace,
                           workspace_bytes, stream);
}

int diffus_render_bwd_mse(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                          const void *dirs, int dirs_dtype, int P, int R, int S, int start, float alpha, int sampler,
                          const float *frame, const float *target, float loss_scale, float *loss, float *gvol,
                          int *gvol_touched, float *gsrc, float *gdirs, int stages, void *workspace, size_t workspace_bytes,
                          diffus_stream_t stream)
{
    return render_bwd_impl(vol, d0, d1, d2, layout, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, sampler, frame,
                           1, target, loss_scale, loss, nullptr, gvol, gvol_touched, gsrc, gdirs, stages, workspace,
                           workspace_bytes, stream);
}

int diffus_render_step_mse(const float *vol, int d0, int d1, int d2, int layout, const void *src, int src_dtype,
                           const void *dirs, int dirs_dtype, int P, int R, int S, int start, float alpha, int sampler,
                           const float *target, float loss_scale, float *frame, float *loss, float *gvol, int *gvol_touched,
                           float *gsrc, float *gdirs, int stages, void *workspace, size_t workspace_bytes,
                           diffus_stream_t stream)
{
    if (S - start > DIFFUS_MAX_SAMPLES) {
        // rays of more than one launch: the forward and the fused-loss backward as two calls (the segmented backward
        // runs its own carry-only forward passes anyway)
        if (!frame) return DIFFUS_EINVAL;
        if (stages & DIFFUS_BWD_SCAN) {
            int rc = diffus_render_fwd(vol, d0, d1, d2, layout & ~(DIFFUS_GRAD_BRICKED | DIFFUS_FANS_PLANAR), src, src_dtype, dirs, dirs_dtype, P, R, S,
                                       start, alpha, sampler, frame, nullptr, workspace, workspace_bytes, stream);
            if (rc) return rc;
            if (start > 0) stages |= DIFFUS_BWD_KEEP_MEDIAN;
        }
        return diffus_render_bwd_mse(vol, d0, d1, d2, layout, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, sampler,
                                     frame, target, loss_scale, loss, gvol, gvol_touched, gsrc, gdirs, stages, workspace,
                                     workspace_bytes, stream);
    }
    return render_bwd_impl(vol, d0, d1, d2, layout, src, src_dtype, dirs, dirs_dtype, P, R, S, start, alpha, sampler, nullptr,
                           2, target, loss_scale, loss, frame, gvol, gvol_touched, gsrc, gdirs, stages, workspace,
                           workspace_bytes, stream);
}

} // extern "C"
