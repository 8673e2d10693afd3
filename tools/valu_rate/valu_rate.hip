/* Issue cost of gfx950 vector instructions, measured: one workgroup of 256 x W threads per CU (W waves on every SIMD) runs a
 * loop of 8 x 8 instructions of ONE kind (8 independent chains); reported = time per wave instruction per SIMD with W = 4
 * (what k_bounce runs at), in ns and relative to v_fma_f64.  usage: valu_rate [waves_per_simd]   (tools/README.md) */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP8(x) x x x x x x x x
#define F32 float, (float)threadIdx.x, 0.999f, 0.001f
#define F64 double, (double)threadIdx.x, 0.999, 0.001
#define PK f2, (f2{(float)threadIdx.x, 1.0f}), (f2{0.999f, 0.001f}), (f2{0.001f, 0.999f})
#define U32 unsigned, threadIdx.x, 12345u, 77u
#define D(k) "%" #k
#define I3(op, k) op " " D(k) ", " D(k) ", %8, %9\n"
#define I2(op, k) op " " D(k) ", " D(k) ", %8\n"
#define I1(op, k) op " " D(k) ", " D(k) "\n"
#define X_fma32(k) I3("v_fma_f32", k)
#define X_fmac32(k) I2("v_fmac_f32", k)
#define X_mul32(k) I2("v_mul_f32", k)
#define X_add32(k) I2("v_add_f32", k)
#define X_min32(k) I2("v_min_f32", k)
#define X_min3_32(k) I3("v_min3_f32", k)
#define X_max3_32(k) I3("v_max3_f32", k)
#define X_rcp32(k) I1("v_rcp_f32", k)
#define X_cmp32(k) "v_cmp_ge_f32 vcc, " D(k) ", %8\n"
#define X_cmp32_e64(k) "v_cmp_ge_f32 s[20:21], " D(k) ", %8\n"
#define X_cndmask(k) "v_cndmask_b32 " D(k) ", " D(k) ", %8, vcc\n"
#define X_cndmask_sdwa(k) "v_cndmask_b32_sdwa " D(k) ", " D(k) ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define X_cnd_e64(k) "v_cndmask_b32_e64 " D(k) ", " D(k) ", %8, s[20:21]\n"
#define X_cnd_indep(k) "v_cndmask_b32 " D(k) ", %8, %9, vcc\n"
#define X_cnd_vccset(k) "s_mov_b64 vcc, exec\nv_cndmask_b32 " D(k) ", " D(k) ", %8, vcc\n"
#define X_cmp_cnd(k) "v_cmp_ge_f32 vcc, " D(k) ", %8\nv_cndmask_b32 " D(k) ", " D(k) ", %9, vcc\n"
#define X_cmp_cnd_s(k) "v_cmp_ge_f32 s[20:21], " D(k) ", %8\nv_cndmask_b32_e64 " D(k) ", " D(k) ", %9, s[20:21]\n"
#define X_cnd_e64_vcc(k) "v_cndmask_b32_e64 " D(k) ", " D(k) ", %8, vcc\n"
#define X_sand_cnd(k) "s_and_b64 vcc, exec, exec\nv_cndmask_b32 " D(k) ", " D(k) ", %8, vcc\n"
#define X_sand_cnd_s(k) "s_and_b64 s[20:21], exec, exec\nv_cndmask_b32_e64 " D(k) ", " D(k) ", %8, s[20:21]\n"
#define X_sand_nop_cnd(k) "s_and_b64 vcc, exec, exec\ns_nop 4\nv_cndmask_b32 " D(k) ", " D(k) ", %8, vcc\n"
#define X_cmp_sand_cnd(k) "v_cmp_ge_f32 vcc, " D(k) ", %8\ns_and_b64 vcc, vcc, exec\nv_cndmask_b32 " D(k) ", " D(k) ", %9, vcc\n"
#define X_cmp_sand_cnd_s(k) "v_cmp_ge_f32 s[20:21], " D(k) ", %8\ns_and_b64 s[20:21], s[20:21], exec\nv_cndmask_b32_e64 " D(k) ", " D(k) ", %9, s[20:21]\n"
#define X_addc(k) "v_addc_co_u32 " D(k) ", vcc, " D(k) ", %8, vcc\n"
#define X_cmp_cnd2(k) "v_cmp_ge_f32 vcc, " D(k) ", %8\nv_cndmask_b32 " D(k) ", " D(k) ", %9, vcc\nv_cndmask_b32 " D(k) ", " D(k) ", %8, vcc\n"
#define X_cmp_gap_cnd(k) "v_cmp_ge_f32 vcc, " D(k) ", %8\nv_add_f32 " D(k) ", " D(k) ", %8\nv_add_f32 " D(k) ", " D(k) ", %8\nv_add_f32 " D(k) ", " D(k) ", %8\nv_cndmask_b32 " D(k) ", " D(k) ", %9, vcc\n"
#define X_cmps_cnd_e32(k) "v_cmp_ge_f32 s[20:21], " D(k) ", %8\ns_mov_b64 vcc, s[20:21]\nv_cndmask_b32 " D(k) ", " D(k) ", %9, vcc\n"
#define X_cmps_cnd_e64(k) "v_cmp_ge_f32 s[20:21], " D(k) ", %8\ns_mov_b64 vcc, s[20:21]\nv_cndmask_b32_e64 " D(k) ", " D(k) ", %9, vcc\n"
#define X_addu32(k) I2("v_add_u32", k)
#define X_lshladd(k) I3("v_lshl_add_u32", k)
#define X_andor(k) I3("v_and_or_b32", k)
#define X_bfi(k) I3("v_bfi_b32", k)
#define X_mov32(k) "v_mov_b32 " D(k) ", %8\n"
#define X_cmpu32(k) "v_cmp_gt_u32 vcc, " D(k) ", %8\n"
#define X_readlane(k) "v_readlane_b32 s20, " D(k) ", 3\n"
#define X_pkfma(k) I3("v_pk_fma_f32", k)
#define X_pkfma_sel(k) "v_pk_fma_f32 " D(k) ", " D(k) ", %8, %8 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n"
#define X_pkmul(k) I2("v_pk_mul_f32", k)
#define X_pkadd(k) I2("v_pk_add_f32", k)
#define X_fma64(k) I3("v_fma_f64", k)
#define X_fma64_s(k) "v_fma_f64 " D(k) ", " D(k) ", %8, s[22:23]\n"
#define X_mul64(k) I2("v_mul_f64", k)
#define X_add64(k) I2("v_add_f64", k)
#define X_min64(k) I2("v_min_f64", k)
#define X_rcp64(k) I1("v_rcp_f64", k)
#define X_rsq64(k) I1("v_rsq_f64", k)
#define X_sqrt64(k) I1("v_sqrt_f64", k)
#define X_divscale64(k) "v_div_scale_f64 " D(k) ", vcc, " D(k) ", %8, %9\n"
#define X_divfmas64(k) I3("v_div_fmas_f64", k)
#define X_divfixup64(k) I3("v_div_fixup_f64", k)
#define X_cmp64(k) "v_cmp_ge_f64 vcc, " D(k) ", %8\n"
#define X_class64(k) "v_cmp_class_f64 vcc, " D(k) ", %10\n"
#define X_cvt_f32_f64(k) "v_cvt_f32_f64 " D(k) ", %[p64]\n"
#define X_mov64(k) "v_mov_b64 " D(k) ", %8\n"
#define X_ldexp64(k) "v_ldexp_f64 " D(k) ", " D(k) ", 1\n"
#define X_frexp64(k) "v_frexp_mant_f64 " D(k) ", " D(k) "\n"
#define X_trig64(k) "v_trig_preop_f64 " D(k) ", " D(k) ", 1\n"
#define X_cvt_i32_f64(k) "v_cvt_i32_f64 %0, " D(k) "\n"
#define X_mulhi(k) I2("v_mul_hi_u32", k)
#define X_mullo(k) I2("v_mul_lo_u32", k)
#define LIST(M)                                                                                                               \
  M(fma32, F32) M(fmac32, F32) M(mul32, F32) M(add32, F32) M(min32, F32) M(min3_32, F32) M(max3_32, F32) M(rcp32, F32)        \
  M(cmp32, F32) M(cmp32_e64, F32) M(cndmask, F32) M(cndmask_sdwa, F32) M(cnd_e64, F32) M(cnd_indep, F32) M(cnd_vccset, F32) M(cmp_cnd, F32) M(cmp_cnd_s, F32) M(cnd_e64_vcc, F32) M(sand_cnd, F32) M(sand_cnd_s, F32) M(sand_nop_cnd, F32) M(cmp_sand_cnd, F32) M(cmp_sand_cnd_s, F32) M(addc, U32) M(cmp_cnd2, F32) M(cmp_gap_cnd, F32) M(cmps_cnd_e32, F32) M(cmps_cnd_e64, F32) M(addu32, U32) M(lshladd, U32) M(andor, U32) M(bfi, U32) \
  M(mov32, U32) M(cmpu32, U32) M(readlane, U32) M(mulhi, U32) M(mullo, U32) M(pkfma, PK) M(pkfma_sel, PK) M(pkmul, PK) M(pkadd, PK) \
  M(fma64, F64) M(fma64_s, F64) M(mul64, F64) M(add64, F64) M(min64, F64) M(rcp64, F64) M(rsq64, F64) M(sqrt64, F64)           \
  M(divscale64, F64) M(divfmas64, F64) M(divfixup64, F64) M(cmp64, F64) M(class64, F64) M(mov64, F64) M(ldexp64, F64)          \
  M(frexp64, F64) M(trig64, F64)
/* T, INIT, P, Q = register type and start values of the eight chain values and of the two extra operands p, q */
#define KERNEL(NAME, T, INIT, P, Q, INS, ...)                                                                                 \
  __global__ void NAME(int iters, double* out) {                                                                              \
    T a0 = INIT, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, p = P, q = Q;                                   \
    for (int it = 0; it < iters; ++it) {                                                                                      \
      REP8(asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7)                                               \
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                     \
                        : "v"(p), "v"(q), "s"(0x1f8)                                                                          \
                        : "vcc", "scc", "s20", "s21", "s22", "s23");)                                                                \
    }                                                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (double)*(float*)&a0 + (double)*(float*)&a1 + (double)*(float*)&a2 +         \
                                                 (double)*(float*)&a3 + (double)*(float*)&a4 + (double)*(float*)&a5 +         \
                                                 (double)*(float*)&a6 + (double)*(float*)&a7;                                 \
  }
#define DEF2(n, ...) KERNEL(k_##n, __VA_ARGS__, X_##n)
#define M_DEF(n, t) DEF2(n, t)
LIST(M_DEF)
typedef void (*kern_t)(int, double*);

/* ---- hand-written blocks (float chains): time per BLOCK, the block's instruction count is in the table */
#define CUSTOM(NAME, BODY)                                                                                                    \
  __global__ void kc_##NAME(int iters, double* out) {                                                                         \
    float a0 = (float)threadIdx.x, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, p = 0.999f, q = 0.001f;      \
    for (int it = 0; it < iters; ++it) {                                                                                      \
      REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                \
                        : "v"(p), "v"(q) : "vcc", "scc", "s20", "s21", "s22", "s23");)                                        \
    }                                                                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                       \
  }
#define CMP(j) "v_cmp_ge_f32 vcc, %" #j ", %8\n"
#define C32(j) "v_cndmask_b32_e32 %" #j ", %" #j ", %9, vcc\n"
#define C64(j) "v_cndmask_b32_e64 %" #j ", %" #j ", %9, vcc\n"
#define CSD(j) "v_cndmask_b32_sdwa %" #j ", %" #j ", %9, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
#define ADD(j) "v_add_f32 %" #j ", %" #j ", %8\n"
CUSTOM(cmp_2c32, CMP(0) C32(0) C32(1) CMP(2) C32(2) C32(3) CMP(4) C32(4) C32(5) CMP(6) C32(6) C32(7))
CUSTOM(cmp_2c64, CMP(0) C64(0) C64(1) CMP(2) C64(2) C64(3) CMP(4) C64(4) C64(5) CMP(6) C64(6) C64(7))
CUSTOM(cmp_2csd, CMP(0) CSD(0) CSD(1) CMP(2) CSD(2) CSD(3) CMP(4) CSD(4) CSD(5) CMP(6) CSD(6) CSD(7))
CUSTOM(cmp_4c32, CMP(0) C32(0) C32(1) C32(2) C32(3) CMP(4) C32(4) C32(5) C32(6) C32(7))
CUSTOM(cmp_4c64, CMP(0) C64(0) C64(1) C64(2) C64(3) CMP(4) C64(4) C64(5) C64(6) C64(7))
CUSTOM(c32_add, C32(0) ADD(1) C32(2) ADD(3) C32(4) ADD(5) C32(6) ADD(7))
CUSTOM(c64_add, C64(0) ADD(1) C64(2) ADD(3) C64(4) ADD(5) C64(6) ADD(7))
CUSTOM(cmp_c32_add_c32, CMP(0) C32(0) ADD(2) C32(1) CMP(4) C32(4) ADD(6) C32(5))
CUSTOM(add8, ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7))

/* odd waves run back-to-back VOP2 v_cndmask_b32 (2 cmp + 8 cnd), even waves 10 adds: does the penalty of the first stall the SIMD
 * or only its own wave? */
__global__ void kc_mix(int iters, double* out) {
  float a0 = (float)threadIdx.x, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, p = 0.999f, q = 0.001f;
  if ((threadIdx.x >> 6) & 1) {
    for (int it = 0; it < iters; ++it) {
      REP8(asm volatile(CMP(0) C32(0) C32(1) C32(2) C32(3) CMP(4) C32(4) C32(5) C32(6) C32(7)
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(q) : "vcc", "scc");)
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      REP8(asm volatile(ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7) ADD(0) ADD(1)
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(p), "v"(q) : "vcc", "scc");)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
CUSTOM(add10, ADD(0) ADD(1) ADD(2) ADD(3) ADD(4) ADD(5) ADD(6) ADD(7) ADD(0) ADD(1))
#define VNOP "v_nop\n"
#define SNOP "s_nop 0\n"
#define MOV(j) "v_mov_b32 %" #j ", %" #j "\n"
CUSTOM(c64_c32, CMP(0) C64(0) C32(1) C64(2) C32(3) CMP(4) C64(4) C32(5) C64(6) C32(7))
CUSTOM(c32_c64, CMP(0) C32(0) C64(1) C32(2) C64(3) CMP(4) C32(4) C64(5) C32(6) C64(7))
CUSTOM(c32_vnop, CMP(0) C32(0) VNOP C32(1) VNOP C32(2) VNOP C32(3) VNOP CMP(4) C32(4) VNOP C32(5) VNOP C32(6) VNOP C32(7) VNOP)
CUSTOM(c32_snop, CMP(0) C32(0) SNOP C32(1) SNOP C32(2) SNOP C32(3) SNOP CMP(4) C32(4) SNOP C32(5) SNOP C32(6) SNOP C32(7) SNOP)
CUSTOM(csd_vnop, CMP(0) CSD(0) VNOP CSD(1) VNOP CSD(2) VNOP CSD(3) VNOP CMP(4) CSD(4) VNOP CSD(5) VNOP CSD(6) VNOP CSD(7) VNOP)
CUSTOM(csd_c64, CMP(0) CSD(0) C64(1) CSD(2) C64(3) CMP(4) CSD(4) C64(5) CSD(6) C64(7))
struct CEntry { const char* name; kern_t fn; int n; };
static CEntry centries[] = {{"cmp_2c32 (4 cmp + 8 cnd e32)", kc_cmp_2c32, 12}, {"cmp_2c64 (4 cmp + 8 cnd e64)", kc_cmp_2c64, 12},
                            {"cmp_2csd (4 cmp + 8 cnd sdwa)", kc_cmp_2csd, 12}, {"cmp_4c32 (2 cmp + 8 cnd e32)", kc_cmp_4c32, 10},
                            {"cmp_4c64 (2 cmp + 8 cnd e64)", kc_cmp_4c64, 10}, {"c32_add (4 cnd e32 + 4 add)", kc_c32_add, 8},
                            {"c64_add (4 cnd e64 + 4 add)", kc_c64_add, 8}, {"cmp_c32_add_c32 (2 cmp 4 cnd e32 2 add)", kc_cmp_c32_add_c32, 8},
                            {"add8 (8 add)", kc_add8, 8}, {"c64_c32 (2 cmp, 4 x (e64, e32))", kc_c64_c32, 10}, {"c32_c64 (2 cmp, 4 x (e32, e64))", kc_c32_c64, 10},
                            {"c32_vnop (2 cmp, 8 x (e32, v_nop))", kc_c32_vnop, 18}, {"c32_snop (2 cmp, 8 x (e32, s_nop))", kc_c32_snop, 10},
                            {"csd_vnop (2 cmp, 8 x (sdwa, v_nop))", kc_csd_vnop, 18}, {"csd_c64 (2 cmp, 4 x (sdwa, e64))", kc_csd_c64, 10}, {"add10 (10 add)", kc_add10, 10},
                            {"mix: odd waves cmp_4c32, even waves add10", kc_mix, 10}};
struct Entry { const char* name; kern_t fn; };
#define M_ENT(n, t) {#n, k_##n},
static Entry entries[] = {LIST(M_ENT)};
static double run(kern_t fn, int waves_per_simd) {
  const int iters = 4000, threads = 256 * waves_per_simd, blocks = 256;
  static double* out = nullptr;
  if (!out) (void)hipMalloc(&out, sizeof(double) * 1024 * 256);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, 50, out);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(threads), 0, 0, iters, out);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e6 / ((double)iters * 64 * waves_per_simd); /* ns per wave instruction per SIMD */
}
int main(int argc, char** argv) {
  const int w = argc > 1 ? atoi(argv[1]) : 4;
  double ref = 0;
  for (const Entry& e : entries) if (!strcmp(e.name, "fma64")) ref = run(e.fn, w);
  printf("waves per SIMD %d; v_fma_f64 = %.3f ns per wave instruction per SIMD (4 cycles at %.2f GHz)\n", w, ref, 4.0 / ref);
  for (const Entry& e : entries) {
    if (argc > 2 && !strstr(e.name, argv[2])) continue;
    const double ns = run(e.fn, w);
    printf("  %-14s %.3f ns  = %.2f x v_fma_f64 = %.1f cycles\n", e.name, ns, ns / ref, 4.0 * ns / ref);
    fflush(stdout);
  }
  for (const CEntry& e : centries) {
    if (argc > 2 && !strstr("custom", argv[2])) continue;
    const double ns = run(e.fn, w) * 8.0; /* run() divides by 64 units per iteration; a block is repeated 8 times */
    printf("  block %-42s %.2f ns = %.1f cycles per block, %.2f per instruction\n", e.name, ns, 4.0 * ns / ref, 4.0 * ns / ref / e.n);
    fflush(stdout);
  }
  return 0;
}
