// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of the VALU ops the sweep
// kernels are made of, on gfx950.  One workgroup per CU, W waves per SIMD; each wave runs a long
// unrolled stream of one instruction on 8 independent register sets.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define N_ITER 2000

#define KERNEL_D(name, ASM)                                                              \
  __global__ void name(double *out, long long *cyc) {                                    \
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3,         \
           a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                           \
    double b = 1.0000001, c = 1e-9;                                                      \
    int k = 3;                                                                           \
    long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int it = 0; it < N_ITER; it++) {                                                \
      asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),     \
                         "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(k));                   \
    }                                                                                    \
    long long t1 = __builtin_amdgcn_s_memtime();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;  \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                     \
  }

#define OP8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define FMA(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define MUL(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define ADD(i) "v_add_f64 %" #i ", %" #i ", %9\n"
#define LDEXP(i) "v_ldexp_f64 %" #i ", %" #i ", %10\n"
#define FREXPM(i) "v_frexp_mant_f64 %" #i ", %" #i "\n"
#define RNDNE(i) "v_rndne_f64 %" #i ", %" #i "\n"
#define MAXF(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define MOV64(i) "v_mov_b64 %" #i ", %" #i "\n"

KERNEL_D(k_fma, OP8(FMA))
KERNEL_D(k_mul, OP8(MUL))
KERNEL_D(k_add, OP8(ADD))
KERNEL_D(k_ldexp, OP8(LDEXP))
KERNEL_D(k_frexpm, OP8(FREXPM))
KERNEL_D(k_rndne, OP8(RNDNE))
KERNEL_D(k_maxf, OP8(MAXF))
KERNEL_D(k_mov64, OP8(MOV64))

#define KERNEL_I(name, ASM)                                                              \
  __global__ void name(double *out, long long *cyc) {                                    \
    int a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4,            \
        a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;                                           \
    int b = 7, c = 11;                                                                   \
    double dd = threadIdx.x * 0.5 + 3.0;                                                 \
    long long t0 = __builtin_amdgcn_s_memtime();                                         \
    for (int it = 0; it < N_ITER; it++) {                                                \
      asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5),     \
                         "+v"(a6), "+v"(a7) : "v"(b), "v"(c), "v"(dd) : "s10", "s11", "vcc");                  \
    }                                                                                    \
    long long t1 = __builtin_amdgcn_s_memtime();                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;  \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                     \
  }
#define IADD(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define IMAX(i) "v_max_i32 %" #i ", %" #i ", %9\n"
#define CNDM(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define FREXPE(i) "v_frexp_exp_i32_f64 %" #i ", %10\n"
#define CVTI(i) "v_cvt_i32_f64 %" #i ", %10\n"
#define CMPF(i) "v_cmp_gt_f64 vcc, %10, %10\n"
#define CMPI(i) "v_cmp_gt_i32 vcc, %" #i ", %8\n"
#define LSHL(i) "v_lshl_add_u32 %" #i ", %" #i ", 3, %8\n"
#define CNDM64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define CNDMK(i) "v_cndmask_b32_e64 %" #i ", 0, %8, s[10:11]\n"
#define ANDB(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define MOVB(i) "v_mov_b32 %" #i ", %8\n"
#define BFI(i) "v_bfi_b32 %" #i ", %9, %8, %" #i "\n"
#define CVTDI(i) "v_cvt_f64_i32 %10, %" #i "\n"
#define SUBI(i) "v_sub_u32 %" #i ", %" #i ", %8\n"
#define MULF32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
KERNEL_I(k_cndmask64, "s_mov_b32 s10, 0x55555555\ns_mov_b32 s11, 0x33333333\n" OP8(CNDM64))
KERNEL_I(k_cndmaskk, "s_mov_b32 s10, 0x55555555\ns_mov_b32 s11, 0x33333333\n" OP8(CNDMK))
KERNEL_I(k_andb, OP8(ANDB))
KERNEL_I(k_movb, OP8(MOVB))
KERNEL_I(k_bfi, OP8(BFI))
KERNEL_I(k_subi, OP8(SUBI))
KERNEL_I(k_mulf32, OP8(MULF32))
#define CNDM0(i) "v_cndmask_b32_e32 %" #i ", 0, %8, vcc\n"
#define CNDMV64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n"
#define CMPCND(i) "v_cmp_gt_i32 vcc, %" #i ", %9\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
#define CMPCND64(i) "v_cmp_gt_i32_e64 s[10:11], %" #i ", %9\nv_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define CMPCND2(i) "v_cmp_gt_i32 vcc, %" #i ", %9\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n"
#define CMPCND3(i) "v_cmp_gt_i32 vcc, %" #i ", %9\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\nv_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n"
KERNEL_I(k_cmpcnd2, OP8(CMPCND2))
KERNEL_I(k_cmpcnd3, OP8(CMPCND3))
KERNEL_I(k_cndm0, OP8(CNDM0))
KERNEL_I(k_cndmv64, OP8(CNDMV64))
KERNEL_I(k_cmpcnd, OP8(CMPCND))
KERNEL_I(k_cmpcnd64, OP8(CMPCND64))
KERNEL_I(k_iadd, OP8(IADD))
KERNEL_I(k_imax, OP8(IMAX))
KERNEL_I(k_cndmask, OP8(CNDM))
KERNEL_I(k_frexpe, OP8(FREXPE))
KERNEL_I(k_cvti, OP8(CVTI))
KERNEL_I(k_cmpf, OP8(CMPF))
KERNEL_I(k_cmpi, OP8(CMPI))
KERNEL_I(k_lshladd, OP8(LSHL))

// dependent chain of fma_f64 (latency)
__global__ void k_fma_dep(double *out, long long *cyc) {
  double a0 = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 1e-9;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N_ITER; it++) {
    asm volatile(REP8("v_fma_f64 %0, %0, %1, %2\n") : "+v"(a0) : "v"(b), "v"(c));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
__global__ void k_iadd_dep(double *out, long long *cyc) {
  int a0 = threadIdx.x, b = 7;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < N_ITER; it++) {
    asm volatile(REP8("v_add_u32 %0, %0, %1\n") : "+v"(a0) : "v"(b));
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kfn)(double *, long long *);
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 512 * 1024 * sizeof(double));
  hipMalloc(&cyc, 1024 * sizeof(long long));
  struct { const char *name; kfn f; } ks[] = {
    {"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_ldexp_f64", k_ldexp},
    {"v_frexp_mant_f64", k_frexpm}, {"v_rndne_f64", k_rndne}, {"v_max_f64", k_maxf}, {"v_mov_b64", k_mov64},
    {"v_frexp_exp_i32_f64", k_frexpe}, {"v_cvt_i32_f64", k_cvti}, {"v_cmp_gt_f64", k_cmpf},
    {"v_add_u32", k_iadd}, {"v_max_i32", k_imax}, {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_i32", k_cmpi},
    {"v_lshl_add_u32", k_lshladd}, {"v_cndmask_e64 sgpr", k_cndmask64}, {"v_cndmask_e64 0,v,sgpr", k_cndmaskk},
    {"v_cndmask_e32 0,v,vcc", k_cndm0}, {"v_cndmask_e64 ...,vcc", k_cndmv64}, {"cmp(vcc)+cndmask_e32 [x2]", k_cmpcnd}, {"cmp(sgpr)+cndmask_e64 [x2]", k_cmpcnd64},
    {"cmp(vcc)+2 cndmask_e32 [x3]", k_cmpcnd2}, {"cmp(vcc)+3 cndmask_e32 [x4]", k_cmpcnd3},
    {"v_and_b32", k_andb}, {"v_mov_b32", k_movb}, {"v_bfi_b32", k_bfi}, {"v_sub_u32", k_subi}, {"v_mul_f32", k_mulf32}, {"v_fma_f64 (dependent)", k_fma_dep}, {"v_add_u32 (dependent)", k_iadd_dep}};
  printf("wall-clock ns per wave-instruction per SIMD (hipEvent around the launch, 256 CUs busy);\n");
  printf("%-26s %9s %9s %9s %9s   | s_memtime ticks/instr/wave at 1 w/SIMD\n", "instruction", "1 w/SIMD", "2 w/SIMD", "4 w/SIMD", "8 w/SIMD");
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (auto &k : ks) {
    printf("%-26s", k.name);
    double ticks1 = 0;
    for (int wps : {1, 2, 4, 8}) {
      int threads = 256 * (wps > 4 ? 4 : wps);
      int blocks = 256 * (wps > 4 ? wps / 4 : 1);
      hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, out, cyc);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k.f, dim3(blocks), dim3(threads), 0, 0, out, cyc);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> h(256);
      hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
      double avg = 0; for (auto v : h) avg += v; avg /= 256;
      if (wps == 1) ticks1 = avg / (double)(N_ITER * 8);
      printf(" %9.3f", ms * 1e6 / (double)(N_ITER * 8) / wps);
    }
    printf("   | %7.3f\n", ticks1);
  }
  return 0;
}
