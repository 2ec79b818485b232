// Developer micro-benchmark: ds_swizzle_b32 / ds_bpermute_b32 throughput alone and interleaved with v_fma_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void k(float* out, int iters, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7;
  float b = seed * 0.5f, c = seed * 0.25f;
  int addr = ((threadIdx.x & 63) ^ 1) * 4;
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {  // 8 independent swizzles (xor 1), wait once
      REP8(asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n ds_swizzle_b32 %1, %1 offset:0x041F\n ds_swizzle_b32 %2, %2 offset:0x041F\n ds_swizzle_b32 %3, %3 offset:0x041F\n"
                        "ds_swizzle_b32 %4, %4 offset:0x041F\n ds_swizzle_b32 %5, %5 offset:0x041F\n ds_swizzle_b32 %6, %6 offset:0x041F\n ds_swizzle_b32 %7, %7 offset:0x041F\n s_waitcnt lgkmcnt(0)\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 1) {  // bpermute
      REP8(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                        "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(addr));)
    } else if (KIND == 2) {  // 8 swizzles + 24 fma interleaved (ratio 1:3)
      REP8(asm volatile("ds_swizzle_b32 %0, %0 offset:0x041F\n v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n"
                        "ds_swizzle_b32 %1, %1 offset:0x041F\n v_fma_f32 %11, %11, %16, %17\n v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n"
                        "ds_swizzle_b32 %2, %2 offset:0x041F\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n v_fma_f32 %8, %8, %16, %17\n"
                        "ds_swizzle_b32 %3, %3 offset:0x041F\n v_fma_f32 %9, %9, %16, %17\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n"
                        "ds_swizzle_b32 %4, %4 offset:0x041F\n v_fma_f32 %12, %12, %16, %17\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n"
                        "ds_swizzle_b32 %5, %5 offset:0x041F\n v_fma_f32 %15, %15, %16, %17\n v_fma_f32 %8, %8, %16, %17\n v_fma_f32 %9, %9, %16, %17\n"
                        "ds_swizzle_b32 %6, %6 offset:0x041F\n v_fma_f32 %10, %10, %16, %17\n v_fma_f32 %11, %11, %16, %17\n v_fma_f32 %12, %12, %16, %17\n"
                        "ds_swizzle_b32 %7, %7 offset:0x041F\n v_fma_f32 %13, %13, %16, %17\n v_fma_f32 %14, %14, %16, %17\n v_fma_f32 %15, %15, %16, %17\n s_waitcnt lgkmcnt(0)\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),
                          "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(c));)
    } else if (KIND == 3) {  // 24 fma only (baseline for KIND 2)
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(c));)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int KIND> void run(const char* name, int ds_per_iter, int valu_per_iter) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps : {1, 2, 4, 8}) {
    int blocks = 256 * wps, iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters);   // cycles per iteration per wave-slot on a SIMD
    printf("%-26s waves/SIMD=%d  %.2f ms  %.1f cyc/iter/SIMD  (ds %d, valu %d per iter => %.2f cyc per ds per CU, %.2f per valu)\n", name, wps, ms, cyc,
           ds_per_iter, valu_per_iter, ds_per_iter ? cyc / ds_per_iter / 4 : 0.0, valu_per_iter ? cyc / valu_per_iter : 0.0);
  }
  hipFree(out);
}
int main() {
  run<0>("ds_swizzle x8", 64, 0);
  run<1>("ds_bpermute x8", 64, 0);
  run<2>("swizzle:fma 1:3", 64, 192);
  run<3>("fma only (192)", 0, 192);
  return 0;
}
