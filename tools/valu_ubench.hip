// valu_ubench.hip -- issue cost of the VALU instruction classes the unit kernel is made of, on gfx950.
//
// Why: k_unit_fb is VALU-issue bound (profiles/r03_pmc_valu.csv) and 36 % of its instructions are
// packed fp32 (v_pk_{add,mul,fma}_f32).  Whether "pack more" is a lever depends on what a packed
// instruction costs next to two plain ones -- this measures it instead of assuming it.
//
// Method: every wave runs ITERS iterations of a block of 64 instructions of one class on 8 independent
// register chains (dependent distance 8: no dependency stall); 256-thread blocks (one wave per SIMD), W
// blocks per CU pinned by the dynamic LDS size, six full rounds of blocks per launch.  Reported: wall ns
// per wave-instruction per SIMD (HIP events over the launch; what a SIMD sustains with W waves resident),
// and s_memtime ticks per instruction of the median wave / W.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_ubench tools/valu_ubench.hip && /tmp/valu_ubench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// 8 chains x 8 = 64 instructions per asm block
#define REP8(S) S S S S S S S S
#define BLOCK1(INS) \
    asm volatile(REP8(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(b), "v"(c))

#define I_FMA(n)   "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_ADD(n)   "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_MUL(n)   "v_mul_f32 %" #n ", %" #n ", %8\n"
#define I_MOV(n)   "v_mov_b32 %" #n ", %8\n"
#define I_DPPMOV(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_DPPADD(n) "v_add_f32_dpp %" #n ", %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define I_CND(n)   "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define I_CMP(n)   "v_cmp_gt_f32 vcc, %" #n ", %8\n"
#define I_RCP(n)   "v_rcp_f32 %" #n ", %" #n "\n"
#define I_EXP(n)   "v_exp_f32 %" #n ", %" #n "\n"
#define I_SQRT(n)  "v_sqrt_f32 %" #n ", %" #n "\n"
#define I_SIN(n)   "v_sin_f32 %" #n ", %" #n "\n"
#define I_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define I_LSHL(n)  "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_AND(n)   "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_ADDU(n)  "v_add_u32 %" #n ", %" #n ", %8\n"
#define I_MED3(n)  "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_MAX3(n)  "v_max3_f32 %" #n ", %" #n ", %8, %9\n"
#define I_CVTFU(n) "v_cvt_f32_u32 %" #n ", %" #n "\n"
#define I_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n"
#define I_FMAC(n)  "v_fmac_f32 %" #n ", %8, %9\n"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define I_CND64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define I_CNDD(n)  "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_ADDS(n)  "v_add_f32 %" #n ", s20, %" #n "\n"
#define I_MULK(n)  "v_mul_f32 %" #n ", 0x3f8ccccd, %" #n "\n"
#define I_MAXI(n)  "v_max_i32 %" #n ", %" #n ", %8\n"
#define I_MUL24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_SUBF(n)  "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_MAXF(n)  "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_CVTIF(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define I_LSHR(n)  "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define I_LSHLV(n) "v_lshlrev_b32 %" #n ", %8, %" #n "\n"
#define I_OR(n)    "v_or_b32 %" #n ", %" #n ", %8\n"
#define I_BFE(n)   "v_bfe_u32 %" #n ", %" #n ", 3, 5\n"
#define I_ADD3(n)  "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_CMP64(n) "v_cmp_gt_f32_e64 s[20:21], %" #n ", %8\n"
#define I_DIVSC(n) "v_div_scale_f32 %" #n ", vcc, %" #n ", %8, %" #n "\n"
#define I_DIVFX(n) "v_div_fixup_f32 %" #n ", %" #n ", %8, %9\n"
#define I_ABSADD(n) "v_add_f32_e64 %" #n ", |%" #n "|, %8\n"
#define I_FMA2(n)  "v_fma_f32 %" #n ", %8, %9, %" #n "\n"
#define I_PKMULS(n) "v_pk_mul_f32 %" #n ", %" #n ", %8 op_sel_hi:[1,0]\n"
#define I_CND64V(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, vcc\n"
#define I_CNDK(n)  "v_cndmask_b32_e64 %" #n ", 0, 1.0, vcc\n"
#define I_CMPCND(n) "v_cmp_gt_f32 vcc, %" #n ", %8\n v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define I_CMPCND64(n) "v_cmp_gt_f32_e64 s[20:21], %" #n ", %8\n v_cndmask_b32_e64 %" #n ", %" #n ", %9, s[20:21]\n"
#define I_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
// packed: operands are 64-bit register pairs
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define I_PKMOV(n) "v_pk_mov_b32 %" #n ", %8, %9\n"
// mixes: one packed, one plain (on different chains)
#define I_MIX_A(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"

enum Op { FMA, ADD, MUL, MOV, DPPMOV, DPPADD, CND, CMP, RCP, EXP, SQRT, SIN, MAD24, MULLO, LSHL, AND, ADDU, MED3,
          MAX3, CVTFU, FLOOR, FMAC, LSHLADD, PKFMA, PKADD, PKMUL, PKMOV, FMA_DEP, PKFMA_DEP, MIX_PK_S,
          CNDV, CND64, CNDD, ADDS, MULK, MAXI, MUL24, SUBF, MAXF, CVTIF, LSHR, LSHLV, OR, BFE, ADD3, CMP64, DIVSC, DIVFX, ABSADD, FMA2, PKMULS,
          CND64V, CNDK, CND1IN8, CND64_1IN8, CMPCND, CMPCND64, ADDCO, NOPS };
static const char *kNames[] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_mov_b32", "v_mov_b32_dpp", "v_add_f32_dpp",
    "v_cndmask_b32", "v_cmp_gt_f32", "v_rcp_f32", "v_exp_f32", "v_sqrt_f32", "v_sin_f32", "v_mad_u32_u24",
    "v_mul_lo_u32", "v_lshlrev_b32", "v_and_b32", "v_add_u32", "v_med3_f32", "v_max3_f32", "v_cvt_f32_u32",
    "v_floor_f32", "v_fmac_f32", "v_lshl_add_u32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_mov_b32",
    "v_fma_f32 (1 dependent chain)", "v_pk_fma_f32 (1 dependent chain)", "alternating v_pk_fma_f32 / v_fma_f32",
    "v_cndmask_b32 (vcc written by v_cmp first)", "v_cndmask_b32_e64 (SGPR-pair mask)", "v_cndmask_b32 (dst != src)",
    "v_add_f32 (SGPR operand)", "v_mul_f32 (32-bit literal)", "v_max_i32", "v_mul_u32_u24", "v_sub_f32", "v_max_f32",
    "v_cvt_i32_f32", "v_lshrrev_b32 (const shift)", "v_lshlrev_b32 (VGPR shift)", "v_or_b32", "v_bfe_u32", "v_add3_u32",
    "v_cmp_gt_f32_e64 (SGPR-pair dst)", "v_div_scale_f32", "v_div_fixup_f32", "v_add_f32_e64 (|abs| modifier)",
    "v_fma_f32 (acc = src2)", "v_pk_mul_f32 (op_sel_hi broadcast)",
    "v_cndmask_b32_e64 (vcc as the e64 mask)", "v_cndmask_b32_e64 0, 1.0, vcc (constants)",
    "1 v_cndmask_b32 (vcc) + 7 v_add_f32", "1 v_cndmask_b32_e64 (SGPR pair) + 7 v_add_f32",
    "pairs: v_cmp vcc + v_cndmask vcc (per PAIR)", "pairs: v_cmp_e64 s[..] + v_cndmask_e64 s[..] (per PAIR)",
    "v_add_co_u32 (writes vcc)"};

template <int OP>
__global__ void __launch_bounds__(256) k_ubench(unsigned long long *out, float *sink, int iters, float seed)
{
    extern __shared__ float dyn_lds[];
    if (seed == 77.0f) dyn_lds[threadIdx.x] = seed;       // the allocation pins the blocks per CU
    const float t = seed + (float)threadIdx.x * 1e-3f;
    unsigned long long t0, t1;
    float r = 0.0f;
    if constexpr ((OP >= PKFMA && OP <= PKMOV) || OP == PKFMA_DEP || OP == PKMULS) {
        f2 a0 = {t, t + 1}, a1 = {t + 2, t}, a2 = {t, t + 3}, a3 = {t + 4, t}, a4 = {t, t + 5}, a5 = {t + 6, t},
           a6 = {t, t + 7}, a7 = {t + 8, t};
        f2 b = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
        asm volatile("" : "+v"(b), "+v"(c));
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            if constexpr (OP == PKFMA) BLOCK1(I_PKFMA);
            if constexpr (OP == PKADD) BLOCK1(I_PKADD);
            if constexpr (OP == PKMUL) BLOCK1(I_PKMUL);
            if constexpr (OP == PKMOV) BLOCK1(I_PKMOV);
            if constexpr (OP == PKMULS) BLOCK1(I_PKMULS);
            if constexpr (OP == PKFMA_DEP)
                asm volatile(REP8(REP8("v_pk_fma_f32 %0, %0, %1, %2\n")) : "+v"(a0) : "v"(b), "v"(c));
        }
        t1 = __builtin_amdgcn_s_memtime();
        const f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
        r = s.x + s.y;
    } else if constexpr (OP == MIX_PK_S) {
        f2 p0 = {t, t + 1}, p1 = {t + 2, t}, p2 = {t, t + 3}, p3 = {t + 4, t};
        float s0 = t, s1 = t + 1, s2 = t + 2, s3 = t + 3;
        f2 b = {1.0001f, 0.9999f}, c = {1e-3f, -1e-3f};
        float bs = 1.0001f, cs = 1e-3f;
        asm volatile("" : "+v"(b), "+v"(c), "+v"(bs), "+v"(cs));
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_fma_f32 %4, %4, %10, %11\n"
                              "v_pk_fma_f32 %1, %1, %8, %9\n v_fma_f32 %5, %5, %10, %11\n"
                              "v_pk_fma_f32 %2, %2, %8, %9\n v_fma_f32 %6, %6, %10, %11\n"
                              "v_pk_fma_f32 %3, %3, %8, %9\n v_fma_f32 %7, %7, %10, %11\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3)
                         : "v"(b), "v"(c), "v"(bs), "v"(cs));
        }
        t1 = __builtin_amdgcn_s_memtime();
        const f2 s = p0 + p1 + p2 + p3;
        r = s.x + s.y + s0 + s1 + s2 + s3;
    } else {
        float a0 = t, a1 = t + 1, a2 = t + 2, a3 = t + 3, a4 = t + 4, a5 = t + 5, a6 = t + 6, a7 = t + 7;
        float b = 1.0001f, c = 1e-3f;
        if constexpr (OP == MAD24 || OP == MULLO || OP == AND || OP == ADDU || OP == LSHLADD || OP == MAXI || OP == MUL24 ||
                      OP == LSHLV || OP == OR || OP == ADD3) {
            b = __builtin_bit_cast(float, 3); c = __builtin_bit_cast(float, 5);
        }
        asm volatile("" : "+v"(b), "+v"(c));
        t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            if constexpr (OP == FMA) BLOCK1(I_FMA);
            if constexpr (OP == ADD) BLOCK1(I_ADD);
            if constexpr (OP == MUL) BLOCK1(I_MUL);
            if constexpr (OP == MOV) BLOCK1(I_MOV);
            if constexpr (OP == DPPMOV) BLOCK1(I_DPPMOV);
            if constexpr (OP == DPPADD) BLOCK1(I_DPPADD);
            if constexpr (OP == CND) {
                asm volatile(REP8(I_CND(0) I_CND(1) I_CND(2) I_CND(3) I_CND(4) I_CND(5) I_CND(6) I_CND(7))
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c) : "vcc");
            }
            if constexpr (OP == CMP) {
                asm volatile(REP8(I_CMP(0) I_CMP(1) I_CMP(2) I_CMP(3) I_CMP(4) I_CMP(5) I_CMP(6) I_CMP(7))
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c) : "vcc");
            }
            if constexpr (OP == RCP) BLOCK1(I_RCP);
            if constexpr (OP == EXP) BLOCK1(I_EXP);
            if constexpr (OP == SQRT) BLOCK1(I_SQRT);
            if constexpr (OP == SIN) BLOCK1(I_SIN);
            if constexpr (OP == MAD24) BLOCK1(I_MAD24);
            if constexpr (OP == MULLO) BLOCK1(I_MULLO);
            if constexpr (OP == LSHL) BLOCK1(I_LSHL);
            if constexpr (OP == AND) BLOCK1(I_AND);
            if constexpr (OP == ADDU) BLOCK1(I_ADDU);
            if constexpr (OP == MED3) BLOCK1(I_MED3);
            if constexpr (OP == MAX3) BLOCK1(I_MAX3);
            if constexpr (OP == CVTFU) BLOCK1(I_CVTFU);
            if constexpr (OP == FLOOR) BLOCK1(I_FLOOR);
            if constexpr (OP == FMAC) BLOCK1(I_FMAC);
            if constexpr (OP == LSHLADD) BLOCK1(I_LSHLADD);
            if constexpr (OP == FMA_DEP)
                asm volatile(REP8(REP8("v_fma_f32 %0, %0, %1, %2\n")) : "+v"(a0) : "v"(b), "v"(c));
#define BLOCKC(INS, PRE, ...) \
    asm volatile(PRE REP8(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(b), "v"(c) : __VA_ARGS__)
            if constexpr (OP == CNDV) BLOCKC(I_CND, "v_cmp_gt_f32 vcc, %0, %8\n", "vcc");
            if constexpr (OP == CND64) BLOCKC(I_CND64, "v_cmp_gt_f32_e64 s[20:21], %0, %8\n", "s20", "s21");
            if constexpr (OP == CNDD) BLOCKC(I_CNDD, "v_cmp_gt_f32 vcc, %0, %8\n", "vcc");
            if constexpr (OP == ADDS) BLOCKC(I_ADDS, "s_mov_b32 s20, 0x3a83126f\n", "s20");
            if constexpr (OP == MULK) BLOCK1(I_MULK);
            if constexpr (OP == MAXI) BLOCK1(I_MAXI);
            if constexpr (OP == MUL24) BLOCK1(I_MUL24);
            if constexpr (OP == SUBF) BLOCK1(I_SUBF);
            if constexpr (OP == MAXF) BLOCK1(I_MAXF);
            if constexpr (OP == CVTIF) BLOCK1(I_CVTIF);
            if constexpr (OP == LSHR) BLOCK1(I_LSHR);
            if constexpr (OP == LSHLV) BLOCK1(I_LSHLV);
            if constexpr (OP == OR) BLOCK1(I_OR);
            if constexpr (OP == BFE) BLOCK1(I_BFE);
            if constexpr (OP == ADD3) BLOCK1(I_ADD3);
            if constexpr (OP == CMP64) BLOCKC(I_CMP64, "", "s20", "s21");
            if constexpr (OP == DIVSC) BLOCKC(I_DIVSC, "", "vcc");
            if constexpr (OP == DIVFX) BLOCK1(I_DIVFX);
            if constexpr (OP == ABSADD) BLOCK1(I_ABSADD);
            if constexpr (OP == FMA2) BLOCK1(I_FMA2);
            if constexpr (OP == CND64V) BLOCKC(I_CND64V, "v_cmp_gt_f32 vcc, %0, %8\n", "vcc");
            if constexpr (OP == CNDK) BLOCKC(I_CNDK, "v_cmp_gt_f32 vcc, %0, %8\n", "vcc");
            if constexpr (OP == CND1IN8)
                asm volatile("v_cmp_gt_f32 vcc, %0, %8\n"
                             REP8(I_CND(0) I_ADD(1) I_ADD(2) I_ADD(3) I_ADD(4) I_ADD(5) I_ADD(6) I_ADD(7))
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c) : "vcc");
            if constexpr (OP == CND64_1IN8)
                asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %8\n"
                             REP8(I_CND64(0) I_ADD(1) I_ADD(2) I_ADD(3) I_ADD(4) I_ADD(5) I_ADD(6) I_ADD(7))
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c) : "s20", "s21");
            if constexpr (OP == CMPCND)
                asm volatile(REP8(I_CMPCND(0) I_CMPCND(1) I_CMPCND(2) I_CMPCND(3))
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c) : "vcc");
            if constexpr (OP == CMPCND64)
                asm volatile(REP8(I_CMPCND64(0) I_CMPCND64(1) I_CMPCND64(2) I_CMPCND64(3))
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                             : "v"(b), "v"(c) : "s20", "s21");
            if constexpr (OP == ADDCO) BLOCKC(I_ADDCO, "", "vcc");
        }
        t1 = __builtin_amdgcn_s_memtime();
        r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (r == 123.456f) sink[0] = r;      // keeps the chains live
}

template <int OP>
static void run_op(int W, int iters, unsigned long long *d_out, float *d_sink, int n_cu, double &cyc, double &wall_ns)
{
    // 256-thread blocks (one wave per SIMD each); W blocks per CU pinned by the dynamic LDS size, a grid of
    // ROUNDS * W * n_cu blocks so that every CU runs W blocks at a time whatever the dispatch order
    constexpr int ROUNDS = 6;
    const dim3 block(256), grid(n_cu * W * ROUNDS);
    const size_t lds = (size_t)(160 * 1024 / W) - 512;
    const int nwaves = grid.x * 4;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ubench<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_ubench<OP>, grid, block, lds, 0, d_out, d_sink, iters / 4, 1.0f);    // warm-up
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_ubench<OP>, grid, block, lds, 0, d_out, d_sink, iters, 1.0f);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(nwaves);
    CHECK(hipMemcpy(h.data(), d_out, nwaves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ninstr = ((OP == CMPCND || OP == CMPCND64) ? 32.0 : 64.0) * iters;
    cyc = (double)h[nwaves / 2] / (ninstr * W);
    // wall: all SIMDs busy the whole time (ROUNDS full rounds): ns per wave-instruction per SIMD
    wall_ns = ms * 1e6 * (n_cu * 4.0) / ((double)nwaves * ninstr);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int OP>
static void sweep(unsigned long long *d_out, float *d_sink, int n_cu)
{
    printf("%-44s", kNames[OP]);
    for (int W : {1, 2, 4, 8}) {
        double cyc, ns;
        run_op<OP>(W, 1500, d_out, d_sink, n_cu, cyc, ns);
        printf("  W=%d: %5.2f ns (%5.2f tick)", W, ns, cyc);
    }
    printf("\n");
    if constexpr (OP + 1 < NOPS) sweep<OP + 1>(d_out, d_sink, n_cu);
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", p.name, n_cu, p.clockRate);
    printf("wall ns per wave-instruction per SIMD with W waves per SIMD resident (256-thread blocks, W per CU pinned by LDS, 6 full rounds);\n"
           "in brackets: s_memtime ticks per instruction of the median wave / W\n");
    unsigned long long *d_out;
    float *d_sink;
    CHECK(hipMalloc(&d_out, (size_t)n_cu * 8 * 6 * 4 * sizeof(unsigned long long)));
    CHECK(hipMalloc(&d_sink, 64));
    if (getenv("UBENCH_NEW")) sweep<CND64V>(d_out, d_sink, n_cu); else sweep<0>(d_out, d_sink, n_cu);
    return 0;
}
