// store_roof.hip -- what a pure-store (and a 95 % store / 5 % load) kernel reaches on this box: the roof that the
// one-step-economy step kernel (BASELINE configs[4]: 6.8 GB of observation rows per launch, 0.35 GB of reads) is
// measured against.  Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/store_roof.hip -o ...
//
// Every variant writes the same TOTAL bytes, split into one contiguous chunk per workgroup of one wavefront (the
// shape of the step kernel: replica e owns rows [e*n, (e+1)*n) of every observation tensor):
//   aligned16        16 B / lane, 1 KiB per store instruction, 16-byte aligned (the best case)
//   aligned16_nt     the same with non-temporal stores
//   rows452          rows of 113 floats = 452 B (dword aligned only), 29 lanes per row, two rows per instruction,
//                    16 B per lane + a 4-byte tail lane: what ose_store_rows issues (round 2)
//   rows404          the same for the 101-float action-mask rows
//   dword            4 B / lane
//   mix95            aligned16 stores + one 16 B / lane load per 19 stores (5 % of the bytes are reads)
//   persistent16     aligned16 from a grid of 256 x k workgroups of 256 threads striding through the buffer
//   copy16           float4 copy (read + write), the guide's 6.29 TB/s figure, for reference
// `lds` bytes of dynamic LDS per workgroup set how many wavefronts a CU holds (14 KB ~ the step kernel's 11).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                  \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

extern __shared__ uint8_t dyn_lds[];

template <bool NT>
__global__ void __launch_bounds__(64) k_aligned16(uint4* __restrict__ out, int quads_per_wg, uint32_t v) {
  uint4* p = out + (size_t)blockIdx.x * quads_per_wg;
  const uint4 val = {v, v + 1, v + 2, threadIdx.x};
  for (int q = threadIdx.x; q < quads_per_wg; q += 64) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if (NT) __builtin_nontemporal_store(u32x4{val.x, val.y, val.z, val.w}, reinterpret_cast<u32x4*>(p + q));
    else p[q] = val;
  }
  if (v == 0xdeadbeef) dyn_lds[threadIdx.x] = 1;  // keeps the dynamic LDS allocation
}

// rows of F floats, L4 = ceil(F/4) lanes per row, 64 / L4 rows per store instruction (aie_kernels_ose.hip: ose_store_rows)
__global__ void __launch_bounds__(64) k_rows(float* __restrict__ out, int rows_per_wg, int F, float v) {
  float* g = out + (size_t)blockIdx.x * rows_per_wg * F;
  const int L4 = (F + 3) >> 2, rpp = 64 / L4, lane = threadIdx.x;
  const int sub = lane / L4, l = lane - sub * L4, j0 = 4 * l;
  const bool active = sub < rpp;
  const int width = F - j0 >= 4 ? 4 : F - j0;
  for (int r0 = 0; r0 < rows_per_wg; r0 += rpp) {
    const int i = r0 + sub;
    if (!active || i >= rows_per_wg) continue;
    float* d = g + (size_t)i * F + j0;
    if (width == 4) {
      // dword-aligned 16-byte store (global dwordx4 needs dword alignment only on gfx950)
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      const f32x4 val = {v, v, v, v};
      asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(d), "v"(val) : "memory");
    } else {
      d[0] = v;
      if (width > 1) d[1] = v;
      if (width > 2) d[2] = v;
    }
  }
  if (v == 12345.f) dyn_lds[threadIdx.x] = 1;
}

__global__ void __launch_bounds__(64) k_dword(uint32_t* __restrict__ out, int dwords_per_wg, uint32_t v) {
  uint32_t* p = out + (size_t)blockIdx.x * dwords_per_wg;
  for (int q = threadIdx.x; q < dwords_per_wg; q += 64) p[q] = v + q;
  if (v == 0xdeadbeef) dyn_lds[threadIdx.x] = 1;
}

__global__ void __launch_bounds__(64) k_mix95(uint4* __restrict__ out, const uint4* __restrict__ in, int quads_per_wg,
                                              uint32_t v) {
  uint4* p = out + (size_t)blockIdx.x * quads_per_wg;
  const uint4* s = in + (size_t)blockIdx.x * (quads_per_wg / 19 + 64);
  uint4 val = {v, v + 1, v + 2, threadIdx.x};
  int k = 0;
  for (int q = threadIdx.x; q < quads_per_wg; q += 64, ++k) {
    if (k % 19 == 0) {
      const uint4 r = s[(k / 19) * 64 + threadIdx.x];
      val.x ^= r.x;
    }
    p[q] = val;
  }
  if (v == 0xdeadbeef) dyn_lds[threadIdx.x] = 1;
}

__global__ void __launch_bounds__(256) k_persistent16(uint4* __restrict__ out, size_t quads, uint32_t v) {
  const uint4 val = {v, v + 1, v + 2, threadIdx.x};
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += stride) out[q] = val;
}

__global__ void __launch_bounds__(256) k_copy16(uint4* __restrict__ out, const uint4* __restrict__ in, size_t quads) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < quads; q += stride) out[q] = in[q];
}

struct Result {
  std::string name;
  int lds;
  double gbs_best, gbs_median, bytes;
};

template <typename F>
static Result timeit(const char* name, int lds, double bytes, F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<double> ms;
  for (int rep = 0; rep < 7; ++rep) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    CK(hipGetLastError());
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    if (rep > 0) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  Result r{name, lds, bytes / (ms.front() * 1e-3) / 1e9, bytes / (ms[ms.size() / 2] * 1e-3) / 1e9, bytes};
  fprintf(stderr, "%-16s lds %6d  best %8.1f GB/s  median %8.1f GB/s  (%.3f ms)\n", name, lds, r.gbs_best, r.gbs_median,
          ms[ms.size() / 2]);
  return r;
}

int main(int argc, char** argv) {
  const int E = 65536, n = 100;                       // BASELINE configs[4]
  const int F_flat = 113, F_mask = 101;
  const size_t rows_bytes = (size_t)E * n * (F_flat + F_mask) * 4;  // 5.6 GB: the two row tensors of a launch
  const int quads_per_wg = n * (F_flat + F_mask) * 4 / 16;          // 5350 quads = 85 600 B per workgroup
  const size_t total = (size_t)E * quads_per_wg * 16;
  uint4 *buf, *src;
  CK(hipMalloc(reinterpret_cast<void**>(&buf), total + (1 << 20)));
  CK(hipMalloc(reinterpret_cast<void**>(&src), total + (1 << 20)));
  CK(hipMemset(buf, 0, total));
  CK(hipMemset(src, 1, total));
  std::vector<Result> res;
  for (int lds : {0, 14336, 32768}) {
    res.push_back(timeit("aligned16", lds, (double)total, [&] {
      hipLaunchKernelGGL(k_aligned16<false>, dim3(E), dim3(64), lds, 0, buf, quads_per_wg, 7u);
    }));
    res.push_back(timeit("aligned16_nt", lds, (double)total, [&] {
      hipLaunchKernelGGL(k_aligned16<true>, dim3(E), dim3(64), lds, 0, buf, quads_per_wg, 7u);
    }));
    res.push_back(timeit("rows452", lds, (double)E * n * F_flat * 4, [&] {
      hipLaunchKernelGGL(k_rows, dim3(E), dim3(64), lds, 0, reinterpret_cast<float*>(buf), n, F_flat, 1.0f);
    }));
    res.push_back(timeit("rows404", lds, (double)E * n * F_mask * 4, [&] {
      hipLaunchKernelGGL(k_rows, dim3(E), dim3(64), lds, 0, reinterpret_cast<float*>(buf), n, F_mask, 1.0f);
    }));
    res.push_back(timeit("mix95", lds, (double)total * (1.0 + 1.0 / 19), [&] {
      hipLaunchKernelGGL(k_mix95, dim3(E), dim3(64), lds, 0, buf, src, quads_per_wg, 7u);
    }));
  }
  res.push_back(timeit("dword", 0, (double)total, [&] {
    hipLaunchKernelGGL(k_dword, dim3(E), dim3(64), 0, 0, reinterpret_cast<uint32_t*>(buf), quads_per_wg * 4, 7u);
  }));
  for (int k : {4, 8, 16}) {
    char nm[32];
    snprintf(nm, sizeof(nm), "persistent16x%d", k);
    res.push_back(timeit(nm, 0, (double)total, [&] {
      hipLaunchKernelGGL(k_persistent16, dim3(256 * k), dim3(256), 0, 0, buf, total / 16, 7u);
    }));
  }
  res.push_back(timeit("copy16x8", 0, 2.0 * total, [&] {
    hipLaunchKernelGGL(k_copy16, dim3(256 * 8), dim3(256), 0, 0, buf, src, total / 16);
  }));
  res.push_back(timeit("hipMemsetD32", 0, (double)total, [&] { CK(hipMemsetD32Async((hipDeviceptr_t)buf, 7, total / 4, 0)); }));
  (void)rows_bytes;
  printf("{\"what\": \"pure-store / mixed roofs, %zu bytes per launch (BASELINE configs[4] row tensors), one wavefront "
         "per 85.6 KB chunk unless named persistent/copy\", \"results\": [",
         total);
  for (size_t i = 0; i < res.size(); ++i)
    printf("%s{\"variant\": \"%s\", \"lds_bytes\": %d, \"GBps_best\": %.1f, \"GBps_median\": %.1f, \"bytes\": %.0f}",
           i ? ", " : "", res[i].name.c_str(), res[i].lds, res[i].gbs_best, res[i].gbs_median, res[i].bytes);
  printf("]}\n");
  return 0;
}
