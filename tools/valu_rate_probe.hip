// valu_rate_probe.hip -- development: issue cost (clocks per wave64 instruction, 8 waves per SIMD resident) of the
// vector instructions the COVID window kernel's dense path spends its time in: v_fma_f64, v_add_f64, v_cvt_f64_i32,
// v_bfe_u32, v_sub_u32.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate_probe.hip -o tools/bin/valu_rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define N 4096

template <int OP>
__global__ void __launch_bounds__(64) k(double* out, int seed) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  int i0 = seed + threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const double m = 1.0000001, c = 1e-9;
#pragma unroll 1
  for (int it = 0; it < N; ++it) {
    if (OP == 0) {  // 8 independent f64 FMAs
      asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                   "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
    } else if (OP == 1) {  // f64 add
      asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                   "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
    } else if (OP == 2) {  // int -> double
      asm volatile("v_cvt_f64_i32 %0, %8\n v_cvt_f64_i32 %1, %9\n v_cvt_f64_i32 %2, %10\n v_cvt_f64_i32 %3, %11\n"
                   "v_cvt_f64_i32 %4, %12\n v_cvt_f64_i32 %5, %13\n v_cvt_f64_i32 %6, %14\n v_cvt_f64_i32 %7, %15"
                   : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3), "=v"(a4), "=v"(a5), "=v"(a6), "=v"(a7)
                   : "v"(i0), "v"(i1), "v"(i2), "v"(i3), "v"(i4), "v"(i5), "v"(i6), "v"(i7));
    } else if (OP == 3) {  // bit-field extract
      asm volatile("v_bfe_u32 %0, %0, 3, 8\n v_bfe_u32 %1, %1, 3, 8\n v_bfe_u32 %2, %2, 3, 8\n v_bfe_u32 %3, %3, 3, 8\n"
                   "v_bfe_u32 %4, %4, 3, 8\n v_bfe_u32 %5, %5, 3, 8\n v_bfe_u32 %6, %6, 3, 8\n v_bfe_u32 %7, %7, 3, 8"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
    } else {  // integer subtract
      asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n"
                   "v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8"
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(seed));
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}

template <int OP>
static void run(const char* name, double* out) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int waves = 8192;  // 8 per SIMD
  hipLaunchKernelGGL(k<OP>, dim3(waves), dim3(64), 0, 0, out, 3);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<OP>, dim3(waves), dim3(64), 0, 0, out, 3);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double insts_per_simd = 8.0 * N * 8;  // 8 waves x N iterations x 8 instructions
  printf("%-16s %8.1f us  -> %.2f clocks per wave instruction at 2.4 GHz\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main() {
  double* out;
  CK(hipMalloc(&out, 8192 * 64 * 8));
  run<0>("v_fma_f64", out);
  run<1>("v_add_f64", out);
  run<2>("v_cvt_f64_i32", out);
  run<3>("v_bfe_u32", out);
  run<4>("v_sub_u32", out);
  return 0;
}
