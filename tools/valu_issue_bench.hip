// Microbenchmark: what one vector (and scalar) instruction COSTS a SIMD of gfx950 (MI355X) in issue cycles, chip-wide,
// as a function of waves per SIMD.  It settles which peak an "issue roofline" may use: the guide
// (MI355X_MICROARCH.md "Wave scheduling") prices a wave64 VALU instruction at 2 cycles on the SIMD-32, round 3's bench.py
// assumed 4.  Measured: BOTH exist -- plain FP32 arithmetic (v_fma/v_mul/v_add_f32 ...) issues every ~2.2 cycles once two
// or more waves share the SIMD, while DPP moves, integer arithmetic, v_ldexp, v_readlane, packed FP32 take ~4, the
// transcendentals 8 -- so a kernel's issue bound is the cost-weighted sum over its instruction mix
// (tools/issue_model.py turns this table + a kernel's disassembly into that bound).
// Every stream is 64 instructions per loop trip on 8 independent destination registers (unless the name says chain);
// every CU gets the same number of resident waves: blocks of 256 k threads with enough LDS that exactly `bpc` blocks fit.
// Build + run (on the GPU box): hipcc -O3 --offload-arch=gfx950 tools/valu_issue_bench.hip -o /tmp/vib && /tmp/vib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

constexpr int kLoop = 256; // loop trips per wave
constexpr int kBody = 64;  // instructions of the measured kind per trip

#define REP8(x) x x x x x x x x
// eight instructions, one per destination register %0..%7; sources %8, %9 (VGPRs), %10 (an SGPR pair), %11 (an SGPR)
#define G8(op, tail) op " %0, " tail "\n" op " %1, " tail "\n" op " %2, " tail "\n" op " %3, " tail "\n" \
                     op " %4, " tail "\n" op " %5, " tail "\n" op " %6, " tail "\n" op " %7, " tail "\n"

#define PROLOGUE                                                                                                   \
    extern __shared__ float pad[];                                                                                 \
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,    \
          a7 = a0 + 7;                                                                                             \
    float m = 1.0000001f, c = 1e-9f;                                                                               \
    unsigned long long smask = 0x5555aaaa5555aaaaull;                                                              \
    int sone = 3;                                                                                                  \
    asm volatile("v_cmp_gt_f32 vcc, %0, %1" ::"v"(a0), "v"(8.0f) : "vcc");                                         \
    unsigned long long t0 = __builtin_readcyclecounter();                                                          \
    unsigned long long r0 = wall_clock64();                                                                        \
    for (int it = 0; it < kLoop; ++it) {
#define EPILOGUE                                                                                                   \
    }                                                                                                              \
    unsigned long long t1 = __builtin_readcyclecounter();                                                          \
    unsigned long long r1 = wall_clock64();                                                                        \
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                               \
    if (s == 12345.678f) out[0] = s + pad[threadIdx.x & 7];                                                        \
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }

#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c), "s"(smask), "s"(sone)

#define DEF(NAME, ASM, ...)                                                                \
    __global__ void NAME(float *out, unsigned long long *clk, float seed)                  \
    {                                                                                      \
        PROLOGUE asm volatile(REP8(ASM) OPS : __VA_ARGS__);                                \
        EPILOGUE                                                                           \
    }

// ---- FP32 arithmetic
DEF(k_fma, G8("v_fma_f32", "%8, %9, %8"), "memory")
DEF(k_fma_acc, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
               "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n", "memory")
DEF(k_fma_chain, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
                 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n", "memory")
DEF(k_mul, G8("v_mul_f32", "%8, %9"), "memory")
DEF(k_add, G8("v_add_f32", "%8, %9"), "memory")
DEF(k_sub, G8("v_sub_f32", "%8, %9"), "memory")
DEF(k_fmac, G8("v_fmac_f32", "%8, %9"), "memory")
DEF(k_max, G8("v_max_f32", "%8, %9"), "memory")
DEF(k_min, G8("v_min_f32", "%8, %9"), "memory")
DEF(k_max3, G8("v_max3_f32", "%8, %9, %8"), "memory")
DEF(k_med3, G8("v_med3_f32", "%8, %9, %8"), "memory")
DEF(k_mov, G8("v_mov_b32", "%8"), "memory")
DEF(k_floor, G8("v_floor_f32", "%8"), "memory")
DEF(k_fract, G8("v_fract_f32", "%8"), "memory")
DEF(k_rndne, G8("v_rndne_f32", "%8"), "memory")
DEF(k_cvt_i32_f32, G8("v_cvt_i32_f32", "%8"), "memory")
DEF(k_cvt_f32_i32, G8("v_cvt_f32_i32", "%8"), "memory")
DEF(k_cvt_f32_u32, G8("v_cvt_f32_u32", "%8"), "memory")
DEF(k_ldexp, G8("v_ldexp_f32", "%8, %9"), "memory")
DEF(k_frexp_exp, G8("v_frexp_exp_i32_f32", "%8"), "memory")
DEF(k_rcp, G8("v_rcp_f32", "%8"), "memory")
DEF(k_exp, G8("v_exp_f32", "%8"), "memory")
DEF(k_abs_and, G8("v_and_b32", "0x7fffffff, %8"), "memory")
// ---- packed FP32
__global__ void k_pk_fma(float *out, unsigned long long *clk, float seed)
{
    PROLOGUE
    double p0, p1, p2, p3, pm;
    { float t[2] = {a0, a1}; memcpy(&p0, t, 8); } { float t[2] = {a2, a3}; memcpy(&p1, t, 8); }
    { float t[2] = {a4, a5}; memcpy(&p2, t, 8); } { float t[2] = {a6, a7}; memcpy(&p3, t, 8); } { float t[2] = {m, m}; memcpy(&pm, t, 8); }
    asm volatile(REP8("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n"
                      "v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4\n")
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
    { float t[2]; memcpy(t, &p0, 8); a0 = t[0] + t[1]; memcpy(t, &p1, 8); a1 = t[0] + t[1]; memcpy(t, &p2, 8); a2 = t[0]; memcpy(t, &p3, 8); a3 = t[0]; }
    EPILOGUE
}
__global__ void k_pk_addmul(float *out, unsigned long long *clk, float seed)
{
    PROLOGUE
    double p0, p1, p2, p3, pm;
    { float t[2] = {a0, a1}; memcpy(&p0, t, 8); } { float t[2] = {a2, a3}; memcpy(&p1, t, 8); }
    { float t[2] = {a4, a5}; memcpy(&p2, t, 8); } { float t[2] = {a6, a7}; memcpy(&p3, t, 8); } { float t[2] = {c, c}; memcpy(&pm, t, 8); }
    asm volatile(REP8("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                      "v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n")
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
    { float t[2]; memcpy(t, &p0, 8); a0 = t[0] + t[1]; memcpy(t, &p1, 8); a1 = t[0] + t[1]; memcpy(t, &p2, 8); a2 = t[0]; memcpy(t, &p3, 8); a3 = t[0]; }
    EPILOGUE
}
// ---- FP64 (the planar scatter tile)
__global__ void k_add_f64(float *out, unsigned long long *clk, float seed)
{
    PROLOGUE
    double p0 = a0, p1 = a1, p2 = a2, p3 = a3, pm = 1.0000001;
    asm volatile(REP8("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n"
                      "v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pm));
    a0 = (float)(p0 + p1 + p2 + p3);
    EPILOGUE
}
__global__ void k_cvt_f64_f32(float *out, unsigned long long *clk, float seed)
{
    PROLOGUE
    double p0 = a0, p1 = a1, p2 = a2, p3 = a3;
    asm volatile(REP8("v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5\n"
                      "v_cvt_f64_f32 %0, %4\n v_cvt_f64_f32 %1, %5\n v_cvt_f64_f32 %2, %4\n v_cvt_f64_f32 %3, %5\n")
                 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(m), "v"(c));
    a0 = (float)(p0 + p1 + p2 + p3);
    EPILOGUE
}
// ---- integer / bit
DEF(k_add_u32, G8("v_add_u32", "%8, %9"), "memory")
DEF(k_sub_u32, G8("v_sub_u32", "%8, %9"), "memory")
DEF(k_and, G8("v_and_b32", "%8, %9"), "memory")
DEF(k_or, G8("v_or_b32", "%8, %9"), "memory")
DEF(k_lshlrev, G8("v_lshlrev_b32", "3, %8"), "memory")
DEF(k_lshrrev, G8("v_lshrrev_b32", "3, %8"), "memory")
DEF(k_lshl_add, G8("v_lshl_add_u32", "%8, 2, %9"), "memory")
DEF(k_add3, G8("v_add3_u32", "%8, %9, %8"), "memory")
DEF(k_and_or, G8("v_and_or_b32", "%8, %9, %8"), "memory")
DEF(k_bfe, G8("v_bfe_u32", "%8, 2, 5"), "memory")
DEF(k_min_i32, G8("v_min_i32", "%8, %9"), "memory")
DEF(k_max_i32, G8("v_max_i32", "%8, %9"), "memory")
DEF(k_mul_u24, G8("v_mul_u32_u24", "%8, %9"), "memory")
DEF(k_mad_u24, G8("v_mad_u32_u24", "%8, %9, %8"), "memory")
DEF(k_mad_i24, G8("v_mad_i32_i24", "%8, %9, %8"), "memory")
DEF(k_mul_lo, G8("v_mul_lo_u32", "%8, %9"), "memory")
// ---- compares and selects
DEF(k_cmp_vcc, "v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n"
               "v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8\n", "vcc")
DEF(k_cmp_sgpr, "v_cmp_gt_f32 s[20:21], %0, %8\n v_cmp_gt_f32 s[22:23], %1, %8\n v_cmp_gt_f32 s[20:21], %2, %8\n v_cmp_gt_f32 s[22:23], %3, %8\n"
                "v_cmp_gt_f32 s[20:21], %4, %8\n v_cmp_gt_f32 s[22:23], %5, %8\n v_cmp_gt_f32 s[20:21], %6, %8\n v_cmp_gt_f32 s[22:23], %7, %8\n", "s20", "s21", "s22", "s23")
DEF(k_cndmask_vcc, G8("v_cndmask_b32", "%8, %9, vcc"), "memory")
DEF(k_cndmask_vcc_rot, "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                       "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n", "memory")
DEF(k_cndmask_sgpr, G8("v_cndmask_b32", "%8, %9, %10"), "memory")
DEF(k_cmp_cndmask, "v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %1, %8, %9, vcc\n v_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %3, %8, %9, vcc\n"
                   "v_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %5, %8, %9, vcc\n v_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %7, %8, %9, vcc\n", "vcc")
DEF(k_cmp_class, "v_cmp_class_f32 vcc, %0, %9\n v_cmp_class_f32 vcc, %1, %9\n v_cmp_class_f32 vcc, %2, %9\n v_cmp_class_f32 vcc, %3, %9\n"
                 "v_cmp_class_f32 vcc, %4, %9\n v_cmp_class_f32 vcc, %5, %9\n v_cmp_class_f32 vcc, %6, %9\n v_cmp_class_f32 vcc, %7, %9\n", "vcc")
// ---- cross-lane
DEF(k_dpp_shr, G8("v_mov_b32_dpp", "%8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), "memory")
DEF(k_dpp_bcast, G8("v_mov_b32_dpp", "%8 row_bcast:15 row_mask:0xa bank_mask:0xf"), "memory")
DEF(k_dpp_wave_shr, G8("v_mov_b32_dpp", "%8 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), "memory")
DEF(k_add_dpp, G8("v_add_f32_dpp", "%8, %9 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), "memory")
DEF(k_readlane, "v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                "v_readlane_b32 s20, %4, 11\n v_readlane_b32 s21, %5, 13\n v_readlane_b32 s22, %6, 15\n v_readlane_b32 s23, %7, 17\n", "s20", "s21", "s22", "s23")
DEF(k_readfirstlane, "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s22, %2\n v_readfirstlane_b32 s23, %3\n"
                     "v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s22, %6\n v_readfirstlane_b32 s23, %7\n", "s20", "s21", "s22", "s23")
// ---- scalar unit (one per CU, shared by its four SIMDs)
DEF(k_salu, "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
            "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n", "s20", "s21", "s22", "s23", "scc")
// (the next two count BOTH kinds: 32 vector + 32 scalar instructions per trip)
DEF(k_fma_salu, "v_fma_f32 %0, %8, %9, %8\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %8, %9, %8\n s_add_u32 s21, s21, 1\n"
                "v_fma_f32 %2, %8, %9, %8\n s_add_u32 s22, s22, 1\n v_fma_f32 %3, %8, %9, %8\n s_add_u32 s23, s23, 1\n", "s20", "s21", "s22", "s23", "scc")
DEF(k_dpp_salu, "v_mov_b32_dpp %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_add_u32 s20, s20, 1\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_add_u32 s21, s21, 1\n"
                "v_mov_b32_dpp %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_add_u32 s22, s22, 1\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_add_u32 s23, s23, 1\n", "s20", "s21", "s22", "s23", "scc")
// ---- mixes
DEF(k_fma_int_mix, "v_fma_f32 %0, %8, %9, %8\n v_add_u32 %1, %8, %9\n v_fma_f32 %2, %8, %9, %8\n v_add_u32 %3, %8, %9\n"
                   "v_fma_f32 %4, %8, %9, %8\n v_add_u32 %5, %8, %9\n v_fma_f32 %6, %8, %9, %8\n v_add_u32 %7, %8, %9\n", "memory")
DEF(k_fma_dpp_mix, "v_fma_f32 %0, %8, %9, %8\n v_mov_b32_dpp %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %2, %8, %9, %8\n v_mov_b32_dpp %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_fma_f32 %4, %8, %9, %8\n v_mov_b32_dpp %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %6, %8, %9, %8\n v_mov_b32_dpp %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n", "memory")
DEF(k_dpp_dep_mix, "v_mov_b32_dpp %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_fma_f32 %0, %4, %8, %0\n v_fma_f32 %1, %4, %9, %1\n v_mov_b32_dpp %5, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                   "v_fma_f32 %2, %5, %8, %2\n v_fma_f32 %3, %5, %9, %3\n v_mul_f32 %6, %0, %2\n v_mul_f32 %7, %1, %3\n", "memory")

struct Row { const char *name; void (*kern)(float *, unsigned long long *, float); };

int main(int argc, char **argv)
{
    (void)hipSetDevice(0);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    int cus = p.multiProcessorCount, wall_khz = 0;
    (void)hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("# device %s  CUs %d  clockRate %d kHz  wall clock %d kHz; %d x %d instructions per wave\n", p.gcnArchName, cus, p.clockRate, wall_khz, kLoop, kBody);
    float *out; unsigned long long *clk;
    (void)hipMalloc(&out, 64); (void)hipMalloc(&clk, 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
#define ROW(k) {#k, k}
    Row rows[] = {
        ROW(k_fma), ROW(k_fma_acc), ROW(k_fma_chain), ROW(k_mul), ROW(k_add), ROW(k_sub), ROW(k_fmac), ROW(k_max), ROW(k_min), ROW(k_max3), ROW(k_med3),
        ROW(k_mov), ROW(k_floor), ROW(k_fract), ROW(k_rndne), ROW(k_cvt_i32_f32), ROW(k_cvt_f32_i32), ROW(k_cvt_f32_u32), ROW(k_ldexp), ROW(k_frexp_exp),
        ROW(k_rcp), ROW(k_exp), ROW(k_abs_and), ROW(k_pk_fma), ROW(k_pk_addmul), ROW(k_add_f64), ROW(k_cvt_f64_f32),
        ROW(k_add_u32), ROW(k_sub_u32), ROW(k_and), ROW(k_or), ROW(k_lshlrev), ROW(k_lshrrev), ROW(k_lshl_add), ROW(k_add3), ROW(k_and_or), ROW(k_bfe),
        ROW(k_min_i32), ROW(k_max_i32), ROW(k_mul_u24), ROW(k_mad_u24), ROW(k_mad_i24), ROW(k_mul_lo),
        ROW(k_cmp_vcc), ROW(k_cmp_sgpr), ROW(k_cndmask_vcc), ROW(k_cndmask_vcc_rot), ROW(k_cndmask_sgpr), ROW(k_cmp_cndmask), ROW(k_cmp_class),
        ROW(k_dpp_shr), ROW(k_dpp_bcast), ROW(k_dpp_wave_shr), ROW(k_add_dpp), ROW(k_readlane), ROW(k_readfirstlane),
        ROW(k_salu), ROW(k_fma_salu), ROW(k_dpp_salu), ROW(k_fma_int_mix), ROW(k_fma_dpp_mix), ROW(k_dpp_dep_mix),
    };
    struct Occ { int threads, bpc; } occ[] = {{256, 1}, {512, 1}, {1024, 1}, {1024, 2}};
    printf("%-18s %6s %9s %13s %13s %11s %8s\n", "stream", "w/SIMD", "ms", "inst/s", "cyc/inst/SIMD", "cyc(1 wave)", "MHz(eff)");
    const char *only = argc > 1 ? argv[1] : nullptr;
    for (const Row &r : rows) {
        if (only && !strstr(r.name, only)) continue;
        for (const Occ &o : occ) {
            size_t lds = (o.bpc == 1) ? 96 * 1024 : 64 * 1024;
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(r.kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            int grid = cus * o.bpc;
            r.kern<<<grid, o.threads, lds>>>(out, clk, 1.0f);
            (void)hipDeviceSynchronize();
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                (void)hipEventRecord(e0);
                r.kern<<<grid, o.threads, lds>>>(out, clk, 1.0f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            unsigned long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            double waves = (double)grid * o.threads / 64;
            double insts = waves * kLoop * kBody;
            double per_s = insts / (best * 1e-3);
            double wps = (double)o.threads * o.bpc / 256;
            double mhz = wall_khz ? (double)h[0] / ((double)h[1] / (wall_khz * 1e3)) / 1e6 : 0; // shader cycles per wall second (wave 0)
            double clk_hz = mhz > 0 ? mhz * 1e6 : 2.4e9;
            double cyc_per_inst_simd = (best * 1e-3) * clk_hz * (cus * 4) / insts;
            double cyc_one_wave = (double)h[0] / (kLoop * kBody);
            printf("%-18s %6.0f %9.4f %13.4e %13.2f %11.2f %8.0f\n", r.name + 2, wps, best, per_s, cyc_per_inst_simd, cyc_one_wave, mhz);
            fflush(stdout);
        }
    }
    printf("# guide: 256 CUs x 4 SIMDs x 2.4e9 / 2 cycles = 1.2288e12 wave-instructions/s (157.3 TFLOP/s f32 fma)\n");
    return 0;
}
