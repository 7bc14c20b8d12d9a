// valu_rate.hip -- issue rate of wave64 VALU instructions on gfx950, all SIMDs busy (profiles/r04*_valu_rate.json).
// Every lane keeps 8 independent accumulators; one loop body is 64 instructions of ONE opcode (inline asm, so the compiler
// can neither fuse nor drop them).  Reported: wave-instructions per second for the whole chip and the SIMD cycles per
// instruction that corresponds to at the clock rate the runtime reports.
//   hipcc --offload-arch=gfx950 -O2 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define BODY8(OP) \
  asm volatile(OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" \
               OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n" \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
#define BODY64(OP) BODY8(OP); BODY8(OP); BODY8(OP); BODY8(OP); BODY8(OP); BODY8(OP); BODY8(OP); BODY8(OP)

#define KERNEL32(NAME, OP, T)                                                         \
  __global__ void __launch_bounds__(256) NAME(T *out, int iters, T seed)              \
  {                                                                                   \
    T a0 = seed + (T)threadIdx.x, a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3, a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7, b = seed; \
    for (int i = 0; i < iters; i++) { BODY64(OP); }                                   \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;     \
  }
typedef float float2v __attribute__((ext_vector_type(2)));
#define KERNEL64(NAME, OP)                                                            \
  __global__ void __launch_bounds__(256) NAME(float2v *out, int iters, float seed)    \
  {                                                                                   \
    float2v a0 = { seed + threadIdx.x, seed }, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = { seed, seed }; \
    for (int i = 0; i < iters; i++) { BODY64(OP); }                                   \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;     \
  }

KERNEL32(k_add_u32, "v_add_u32", unsigned)
KERNEL32(k_and_b32, "v_and_b32", unsigned)
KERNEL32(k_lshl_b32, "v_lshlrev_b32", unsigned)
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32", unsigned)
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24", unsigned)
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32", unsigned)
KERNEL32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24", unsigned)
KERNEL32(k_add_f32, "v_add_f32", float)
KERNEL32(k_mul_f32, "v_mul_f32", float)
KERNEL32(k_max_f32, "v_max_f32", float)
KERNEL32(k_cvt_like_min_i32, "v_min_i32", int)
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32")
KERNEL64(k_pk_add_f32, "v_pk_add_f32")

// three-operand forms
__global__ void __launch_bounds__(256) k_fma_f32(float *out, int iters, float seed)
{
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed;
  for (int i = 0; i < iters; i++) {
#define F8 asm volatile("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\nv_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
    F8; F8; F8; F8; F8; F8; F8; F8;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(256) k_pk_fma_f32(float2v *out, int iters, float seed)
{
  float2v a0 = { seed + threadIdx.x, seed }, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f, b = { seed, seed };
  for (int i = 0; i < iters; i++) {
#define P8 asm volatile("v_pk_fma_f32 %0, %0, %8, %8\nv_pk_fma_f32 %1, %1, %8, %8\nv_pk_fma_f32 %2, %2, %8, %8\nv_pk_fma_f32 %3, %3, %8, %8\nv_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\nv_pk_fma_f32 %6, %6, %8, %8\nv_pk_fma_f32 %7, %7, %8, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
    P8; P8; P8; P8; P8; P8; P8; P8;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void __launch_bounds__(256) k_mad_u32_u24(unsigned *out, int iters, unsigned seed)
{
  unsigned a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = seed;
  for (int i = 0; i < iters; i++) {
#define M8 asm volatile("v_mad_u32_u24 %0, %0, %8, %8\nv_mad_u32_u24 %1, %1, %8, %8\nv_mad_u32_u24 %2, %2, %8, %8\nv_mad_u32_u24 %3, %3, %8, %8\nv_mad_u32_u24 %4, %4, %8, %8\nv_mad_u32_u24 %5, %5, %8, %8\nv_mad_u32_u24 %6, %6, %8, %8\nv_mad_u32_u24 %7, %7, %8, %8\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b))
    M8; M8; M8; M8; M8; M8; M8; M8;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// v_cndmask needs vcc: a select chain the way the trellis kernels use it
__global__ void __launch_bounds__(256) k_cmp_cndmask(unsigned *out, int iters, unsigned seed)
{
  unsigned a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = seed;
  for (int i = 0; i < iters; i++) {
#define C8 asm volatile("v_cmp_lt_u32 vcc, %0, %4\nv_cndmask_b32 %0, %0, %4, vcc\nv_cmp_lt_u32 vcc, %1, %4\nv_cndmask_b32 %1, %1, %4, vcc\nv_cmp_lt_u32 vcc, %2, %4\nv_cndmask_b32 %2, %2, %4, vcc\nv_cmp_lt_u32 vcc, %3, %4\nv_cndmask_b32 %3, %3, %4, vcc\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc")
    C8; C8; C8; C8; C8; C8; C8; C8;
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3;
}

// ---- more opcodes (second table): one asm template per kernel, %0 = the accumulator, %1 = a second source ----
#define GEN_KERNEL(NAME, T, LINE)                                                      \
  __global__ void __launch_bounds__(256) NAME(T *out, int iters, T seed)               \
  {                                                                                    \
    T a0 = seed + (T)threadIdx.x, a1 = a0 + (T)1, a2 = a0 + (T)2, a3 = a0 + (T)3, a4 = a0 + (T)4, a5 = a0 + (T)5, a6 = a0 + (T)6, a7 = a0 + (T)7, b = seed; \
    for (int i = 0; i < iters; i++) {                                                  \
      _Pragma("unroll") for (int r = 0; r < 8; r++) {                                  \
        asm volatile(LINE : "+v"(a0) : "v"(b) : "vcc"); asm volatile(LINE : "+v"(a1) : "v"(b) : "vcc");  \
        asm volatile(LINE : "+v"(a2) : "v"(b) : "vcc"); asm volatile(LINE : "+v"(a3) : "v"(b) : "vcc");  \
        asm volatile(LINE : "+v"(a4) : "v"(b) : "vcc"); asm volatile(LINE : "+v"(a5) : "v"(b) : "vcc");  \
        asm volatile(LINE : "+v"(a6) : "v"(b) : "vcc"); asm volatile(LINE : "+v"(a7) : "v"(b) : "vcc");  \
      }                                                                                \
    }                                                                                  \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;      \
  }
GEN_KERNEL(g_or, unsigned, "v_or_b32 %0, %0, %1")
GEN_KERNEL(g_xor, unsigned, "v_xor_b32 %0, %0, %1")
GEN_KERNEL(g_sub, unsigned, "v_sub_u32 %0, %0, %1")
GEN_KERNEL(g_maxu, unsigned, "v_max_u32 %0, %0, %1")
GEN_KERNEL(g_minu, unsigned, "v_min_u32 %0, %0, %1")
GEN_KERNEL(g_lshr, unsigned, "v_lshrrev_b32 %0, %1, %0")
GEN_KERNEL(g_ashr, int, "v_ashrrev_i32 %0, %1, %0")
GEN_KERNEL(g_subf, float, "v_sub_f32 %0, %0, %1")
GEN_KERNEL(g_minf, float, "v_min_f32 %0, %0, %1")
GEN_KERNEL(g_addu16, unsigned, "v_add_u16 %0, %0, %1")
GEN_KERNEL(g_maxi16, unsigned, "v_max_i16 %0, %0, %1")
GEN_KERNEL(g_pkaddu16, unsigned, "v_pk_add_u16 %0, %0, %1")
GEN_KERNEL(g_pkmaxi16, unsigned, "v_pk_max_i16 %0, %0, %1")
GEN_KERNEL(g_bfe, unsigned, "v_bfe_u32 %0, %0, %1, 5")
GEN_KERNEL(g_add3, unsigned, "v_add3_u32 %0, %0, %1, %1")
GEN_KERNEL(g_lshladd, unsigned, "v_lshl_add_u32 %0, %0, 2, %1")
GEN_KERNEL(g_lshlor, unsigned, "v_lshl_or_b32 %0, %0, 2, %1")
GEN_KERNEL(g_or3, unsigned, "v_or3_b32 %0, %0, %1, %1")
GEN_KERNEL(g_andor, unsigned, "v_and_or_b32 %0, %0, %1, %1")
GEN_KERNEL(g_perm, unsigned, "v_perm_b32 %0, %0, %1, %1")
GEN_KERNEL(g_mov, unsigned, "v_mov_b32 %0, %1")
GEN_KERNEL(g_cvtfu, unsigned, "v_cvt_f32_u32 %0, %0")
GEN_KERNEL(g_cvtuf, unsigned, "v_cvt_u32_f32 %0, %0")
GEN_KERNEL(g_cndmask, unsigned, "v_cndmask_b32 %0, %0, %1, vcc")
GEN_KERNEL(g_cmp, unsigned, "v_cmp_lt_u32 vcc, %0, %1")
GEN_KERNEL(g_cmpf, float, "v_cmp_lt_f32 vcc, %0, %1")
GEN_KERNEL(g_movdpp, unsigned, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
GEN_KERNEL(g_ffbh, unsigned, "v_ffbh_u32 %0, %0")
GEN_KERNEL(g_mad24, unsigned, "v_mad_u32_u24 %0, %0, %1, %1")
GEN_KERNEL(g_lshl_add_u64ish, unsigned, "v_add_co_u32 %0, vcc, %0, %1")
GEN_KERNEL(g_readlane_like_bperm, unsigned, "v_mbcnt_lo_u32_b32 %0, %1, %0")

template <class K, class T, class S>
static void run(const char *name, K kernel, T *out, S seed, int waves_per_simd, int ncu, double clock_hz, int lanes_per_instr)
{
  const int iters = 4096, blocks = ncu * waves_per_simd;   // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 64, seed);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters, seed);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double instr = (double)blocks * 4 * iters * 64;   // wave-instructions
  const double rate = instr / (best * 1e-3);
  printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_instr_per_s\": %.4g, \"cycles_per_instr_per_simd\": %.2f, \"lane_ops_per_s\": %.4g}\n",
         name, waves_per_simd, best, rate, ncu * 4 * clock_hz / rate, rate * 64 * lanes_per_instr);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int ncu = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %.0f}\n", prop.name, prop.gcnArchName, ncu, clk / 1e6);
  void *out = nullptr;
  hipMalloc(&out, (size_t)ncu * 8 * 256 * 8);
  for (int w : { 1, 2, 4, 8 }) {
    run("v_add_u32", k_add_u32, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_and_b32", k_and_b32, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_lshlrev_b32", k_lshl_b32, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_min_i32", k_cvt_like_min_i32, (int *)out, 3, w, ncu, clk, 1);
    run("v_cmp+v_cndmask (pairs)", k_cmp_cndmask, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_mul_u32_u24", k_mul_u32_u24, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_mad_u32_u24", k_mad_u32_u24, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_mul_hi_u32_u24", k_mul_hi_u32_u24, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_mul_lo_u32", k_mul_lo_u32, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_mul_hi_u32", k_mul_hi_u32, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_add_f32", k_add_f32, (float *)out, 1.0f, w, ncu, clk, 1);
    run("v_mul_f32", k_mul_f32, (float *)out, 1.0f, w, ncu, clk, 1);
    run("v_max_f32", k_max_f32, (float *)out, 1.0f, w, ncu, clk, 1);
    run("v_fma_f32", k_fma_f32, (float *)out, 1.0f, w, ncu, clk, 1);
    run("v_pk_add_f32", k_pk_add_f32, (float2v *)out, 1.0f, w, ncu, clk, 2);
    run("v_pk_mul_f32", k_pk_mul_f32, (float2v *)out, 1.0f, w, ncu, clk, 2);
    run("v_pk_fma_f32", k_pk_fma_f32, (float2v *)out, 1.0f, w, ncu, clk, 2);
    run("v_or_b32", g_or, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_xor_b32", g_xor, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_sub_u32", g_sub, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_max_u32", g_maxu, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_min_u32", g_minu, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_lshrrev_b32", g_lshr, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_ashrrev_i32", g_ashr, (int *)out, 3, w, ncu, clk, 1);
    run("v_sub_f32", g_subf, (float *)out, 1.0f, w, ncu, clk, 1);
    run("v_min_f32", g_minf, (float *)out, 1.0f, w, ncu, clk, 1);
    run("v_add_u16", g_addu16, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_max_i16", g_maxi16, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_pk_add_u16", g_pkaddu16, (unsigned *)out, 3u, w, ncu, clk, 2);
    run("v_pk_max_i16", g_pkmaxi16, (unsigned *)out, 3u, w, ncu, clk, 2);
    run("v_bfe_u32", g_bfe, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_add3_u32", g_add3, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_lshl_add_u32", g_lshladd, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_lshl_or_b32", g_lshlor, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_or3_b32", g_or3, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_and_or_b32", g_andor, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_perm_b32", g_perm, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_mov_b32", g_mov, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_cvt_f32_u32", g_cvtfu, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_cvt_u32_f32", g_cvtuf, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_cndmask_b32", g_cndmask, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_cmp_lt_u32", g_cmp, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_cmp_lt_f32", g_cmpf, (float *)out, 1.0f, w, ncu, clk, 1);
    run("v_mov_b32_dpp", g_movdpp, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_ffbh_u32", g_ffbh, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_add_co_u32", g_lshl_add_u64ish, (unsigned *)out, 3u, w, ncu, clk, 1);
    run("v_mbcnt_lo_u32_b32", g_readlane_like_bperm, (unsigned *)out, 3u, w, ncu, clk, 1);
  }
  hipFree(out);
  return 0;
}
