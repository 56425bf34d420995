// tools/valu_rate.hip — what one wave64 vector instruction of each kind costs a SIMD that has 1, 2 or 4 wavefronts
// to pick from (gfx950).  Each kernel runs a loop of 64 independent instructions of one kind over eight registers;
// the table gives SIMD clocks per instruction, taking v_fma_f32 at 4 wavefronts as 2.0 (its documented rate; the
// first kernel runs while the clock still ramps, so read the other rows against v_add_f32 / v_mul_f32 instead: ~1.6 in
// these units is the 2-clock class).  v_cndmask_b32 alone reads a VCC no instruction of the loop writes and runs at a
// rate it never shows inside real code (see the mixed row).
// Build: hipcc --offload-arch=gfx950 -O2 -o build/valu_rate tools/valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                               INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(s0) : "v"(a), "v"(b) : "vcc", "s20", "s21")

#define KERNEL(NAME, INS)                                                              \
  __global__ void __launch_bounds__(256) NAME(float* out, int iters, float a, float b) { \
    float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
    unsigned s0 = 0;                                                                   \
    for (int i = 0; i < iters; i++) { BODY(INS); }                                     \
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)s0 == 12345.678f) out[0] = r0;  \
  }

#define I_FMA(k) "v_fma_f32 %" #k ", %" #k ", %9, %10\n"
#define I_ADD(k) "v_add_f32 %" #k ", %" #k ", %9\n"
#define I_MUL(k) "v_mul_f32 %" #k ", %" #k ", %9\n"
#define I_MAX(k) "v_max_f32 %" #k ", %" #k ", %9\n"
#define I_AND(k) "v_and_b32 %" #k ", %" #k ", %9\n"
#define I_SHL(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define I_ADDU(k) "v_add_u32 %" #k ", %" #k ", %9\n"
#define I_MOV(k) "v_mov_b32 %" #k ", %9\n"
#define I_CND(k) "v_cndmask_b32 %" #k ", %" #k ", %9, vcc\n"
#define I_CMP(k) "v_cmp_lt_f32 vcc, %" #k ", %9\n"
#define I_CMPS(k) "v_cmp_lt_f32 s[20:21], %" #k ", %9\n"
#define I_RDL(k) "v_readlane_b32 %8, %" #k ", 3\n"
#define I_WRL(k) "v_writelane_b32 %" #k ", %8, 3\n"
#define I_MADU(k) "v_mad_u32_u24 %" #k ", %" #k ", %9, %10\n"
#define I_MULLO(k) "v_mul_lo_u32 %" #k ", %" #k ", %9\n"
#define I_BFE(k) "v_bfe_u32 %" #k ", %" #k ", 3, 5\n"
#define I_CVT(k) "v_cvt_f32_i32 %" #k ", %" #k "\n"
#define I_RCP(k) "v_rcp_f32 %" #k ", %" #k "\n"
#define I_SQRT(k) "v_sqrt_f32 %" #k ", %" #k "\n"
#define I_MBC(k) "v_mbcnt_lo_u32_b32 %" #k ", -1, %" #k "\n"
#define I_SUBR(k) "v_sub_f32 %" #k ", %9, %" #k "\n"
#define I_FMAC(k) "v_fmac_f32 %" #k ", %9, %10\n"
#define I_PKFMA(k) "v_pk_fma_f32 v[" #k "*2+40:" #k "*2+41], v[" #k "*2+40:" #k "*2+41], v[60:61], v[62:63]\n"
#define I_SNOP(k) "s_nop 0\n"
#define I_SADD(k) "s_add_u32 %8, %8, 1\n"
#define I_ABSMAX(k) "v_max_f32 %" #k ", |%" #k "|, %9\n"

KERNEL(k_fma, I_FMA) KERNEL(k_add, I_ADD) KERNEL(k_mul, I_MUL) KERNEL(k_max, I_MAX) KERNEL(k_and, I_AND) KERNEL(k_shl, I_SHL)
KERNEL(k_addu, I_ADDU) KERNEL(k_mov, I_MOV) KERNEL(k_cnd, I_CND) KERNEL(k_cmp, I_CMP) KERNEL(k_cmps, I_CMPS) KERNEL(k_rdl, I_RDL)
KERNEL(k_wrl, I_WRL) KERNEL(k_madu, I_MADU) KERNEL(k_mullo, I_MULLO) KERNEL(k_bfe, I_BFE) KERNEL(k_cvt, I_CVT) KERNEL(k_rcp, I_RCP)
KERNEL(k_sqrt, I_SQRT) KERNEL(k_mbc, I_MBC) KERNEL(k_subr, I_SUBR) KERNEL(k_fmac, I_FMAC) KERNEL(k_snop, I_SNOP) KERNEL(k_sadd, I_SADD)
KERNEL(k_absmax, I_ABSMAX)

// a mix like the sweep's: one compare, one select, one or, one fma per group
#define I_MIX(k) "v_cmp_lt_f32 vcc, %" #k ", %9\nv_cndmask_b32 %" #k ", %" #k ", %9, vcc\nv_fma_f32 %" #k ", %" #k ", %9, %10\nv_and_b32 %" #k ", %" #k ", %9\n"
#define BODY4(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                               : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7), "+s"(s0) : "v"(a), "v"(b) : "vcc", "s20", "s21")
__global__ void __launch_bounds__(256) k_mix(float* out, int iters, float a, float b) {
  float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7;
  unsigned s0 = 0;
  for (int i = 0; i < iters; i++) { BODY4(I_MIX); }
  if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + (float)s0 == 12345.678f) out[0] = r0;
}

struct K { const char* name; void (*fn)(float*, int, float, float); };
int main() {
  K ks[] = {{"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac}, {"v_add_f32", k_add}, {"v_sub_f32", k_subr}, {"v_mul_f32", k_mul}, {"v_max_f32", k_max},
            {"v_max_f32 |x|", k_absmax}, {"v_and_b32", k_and}, {"v_lshlrev_b32", k_shl}, {"v_add_u32", k_addu}, {"v_mov_b32", k_mov}, {"v_cndmask_b32", k_cnd},
            {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmps}, {"v_readlane_b32", k_rdl}, {"v_writelane_b32", k_wrl},
            {"v_mad_u32_u24", k_madu}, {"v_mul_lo_u32", k_mullo}, {"v_bfe_u32", k_bfe}, {"v_cvt_f32_i32", k_cvt}, {"v_rcp_f32", k_rcp},
            {"v_sqrt_f32", k_sqrt}, {"v_mbcnt_lo", k_mbc}, {"s_nop 0", k_snop}, {"mix cmp/cndmask/fma/and (x4)", k_mix}};
  float* out;
  hipMalloc(&out, 4);
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double ref = 0;
  std::vector<std::string> lines;
  printf("%d CUs; clocks per instruction per SIMD (v_fma_f32 at 4 wavefronts per SIMD := 2.0)\n", cus);
  printf("%-32s %10s %10s %10s\n", "instruction", "1 wave", "2 waves", "4 waves");
  for (auto& k : ks) {
    double t[3];
    int wi = 0;
    for (int wps : {1, 2, 4}) {
      // blocks of 256 threads = one wavefront per SIMD; wps blocks per CU
      const int grid = cus * wps;
      k.fn<<<grid, 256>>>(out, 10, 1.0001f, 0.5f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      k.fn<<<grid, 256>>>(out, iters, 1.0001f, 0.5f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double n_ins = (double)iters * 64 * wps * (std::string(k.name).rfind("mix", 0) == 0 ? 1.0 : 1.0);
      t[wi++] = ms * 1e-3 / n_ins;  // seconds per instruction per SIMD
    }
    if (ref == 0) ref = t[2] / 2.0;  // seconds per clock
    printf("%-32s %10.2f %10.2f %10.2f\n", k.name, t[0] / ref, t[1] / ref, t[2] / ref);
  }
  return 0;
}
