// Developer micro-benchmark: VALU issue rates on gfx950 for the instruction kinds the Jacobi
// kernel uses (plain fma, packed fma, DPP mov, fmac with DPP operand, transcendental),
// at 1..8 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND>
__global__ void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = seed * 0.5f, c = seed * 0.25f;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // 8 independent fma chains
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 1) {  // dependent fma chain (1 chain)
      REP16(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(b), "v"(c));)
    } else if (KIND == 2) {  // dpp mov, independent
      REP16(asm volatile("v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_mov_b32_dpp %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
    } else if (KIND == 3) {  // fmac with dpp operand, 8 independent accumulators
      REP16(asm volatile("v_fmac_f32_dpp %0, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %4, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %5, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %6, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %7, %8, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 4) {  // rcp, independent
      REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 5) {  // packed fma: 4 independent pairs
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
      REP16(asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));)
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    } else if (KIND == 6) {  // v_mul + v_fmac pairs dependent within pair, 4 independent pairs (rotation pattern)
      REP16(asm volatile("v_mul_f32 %0, %0, %8\n v_fmac_f32 %0, %4, %9\n v_mul_f32 %1, %1, %8\n v_fmac_f32 %1, %5, %9\n"
                         "v_mul_f32 %2, %2, %8\n v_fmac_f32 %2, %6, %9\n v_mul_f32 %3, %3, %8\n v_fmac_f32 %3, %7, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 7) {  // v_permlane16_swap, 4 independent register pairs
      REP16(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                         "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 8) {  // v_permlane32_swap, 4 independent register pairs
      REP16(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 9) {  // v_cndmask_b32 (vcc), 8 independent
      REP16(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
    } else if (KIND == 10) {  // 4 fma + 4 swaps interleaved (do the swaps hide behind arithmetic?)
      REP16(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_permlane16_swap_b32 %4, %5\n v_fma_f32 %1, %1, %8, %9\n v_permlane16_swap_b32 %6, %7\n"
                         "v_fma_f32 %2, %2, %8, %9\n v_permlane32_swap_b32 %4, %5\n v_fma_f32 %3, %3, %8, %9\n v_permlane32_swap_b32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND> void run(const char* name, int instr_per_iter) {
  float* out;
  hipMalloc(&out, 256 * 8 * 256 * 8 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 3, 4, 8}) {   // waves per SIMD: block of 256 threads = 1 wave per SIMD; blocks per CU = wps
    int blocks = 256 * wps, iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double instr_per_simd = (double)wps * iters * instr_per_iter;   // wave-instructions issued per SIMD
    double cyc = ms * 1e-3 * 2.4e9;
    printf("%-28s waves/SIMD=%d  %.2f ms  %.2f cycles(@2.4GHz)/wave-instr/SIMD\n", name, wps, ms, cyc / instr_per_simd);
  }
  hipFree(out);
}

int main() {
  run<0>("v_fma_f32 x8 indep", 128);
  run<1>("v_fma_f32 dependent", 128);
  run<2>("v_mov_b32_dpp indep", 128);
  run<3>("v_fmac_f32_dpp indep", 128);
  run<4>("v_rcp_f32 indep", 128);
  run<5>("v_pk_fma_f32 x4 indep", 128);
  run<6>("mul+fmac pairs", 128);
  run<7>("v_permlane16_swap indep", 128);
  run<8>("v_permlane32_swap indep", 128);
  run<9>("v_cndmask_b32 indep", 128);
  run<10>("fma + permlane swap 1:1", 128);
  return 0;
}
