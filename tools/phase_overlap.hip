// phase_overlap.hip -- does a "compute, then stream 86 KB" wave shape overlap its compute with other waves' stores?
// Development microbenchmark behind the one-step-economy step kernel's schedule: one wavefront per
// 85.6 KB chunk (65 536 chunks), every wave runs `iters` rounds of ALU work and stores its chunk in one of several
// orders.  Prints ms per launch for: ALU only, stores only, and the combinations.
//   hipcc --offload-arch=gfx950 -O3 tools/phase_overlap.hip -o tools/bin/phase_overlap
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) {                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

extern __shared__ uint8_t dyn_lds[];

__device__ __forceinline__ double alu(double x, int iters) {
  double a = x, b = x + 1, c = x + 2, d = x + 3;
  for (int i = 0; i < iters; ++i) {
    a = __builtin_fma(a, 1.0000001, 0.5);
    b = __builtin_fma(b, 0.9999999, 0.25);
    c = __builtin_fma(c, 1.0000002, 0.125);
    d = __builtin_fma(d, 0.9999998, 0.0625);
  }
  return a + b + c + d;
}

__device__ __forceinline__ void store_quads(uint4* p, int q0, int q1, uint4 val) {
  for (int q = q0 + threadIdx.x; q < q1; q += 64) p[q] = val;
}

// mode 0: ALU then all stores; 1: first 47 % of the stores, ALU, rest; 2: ALU and stores interleaved in `pieces`;
// 3: stores then ALU; 4: like 0 plus a dependent 6 KB load up front (the record) and a 6 KB store at the end
__global__ void __launch_bounds__(64) k_phase(uint4* __restrict__ out, const uint4* __restrict__ in, int quads, int iters,
                                              int mode, int pieces, int do_store, double* sink) {
  uint4* p = out + (size_t)blockIdx.x * quads;
  double x = (double)threadIdx.x;
  uint4 val = {1u, 2u, 3u, threadIdx.x};
  // modes 5..9 dissect mode 4: 5 = load only, 6 = tail store only (other buffer), 7 = load consumed at the end,
  // 8 = load + tail store into another buffer, 9 = load from a small cache-resident region + tail store
  uint4 late[6];
  if (mode == 7) {
    const uint4* s = in + (size_t)blockIdx.x * 384;
#pragma unroll
    for (int k = 0; k < 6; ++k) late[k] = s[k * 64 + threadIdx.x];
  }
  if (mode == 4 || mode == 5 || mode == 8 || mode == 9) {
    const uint4* s = in + (size_t)(mode == 9 ? (blockIdx.x & 255) : blockIdx.x) * 384;
    uint4 v[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) v[k] = s[k * 64 + threadIdx.x];
#pragma unroll
    for (int k = 0; k < 6; ++k) val.x ^= v[k].x;
    x += (double)(val.x & 1u);
  }
  if (mode == 0 || mode >= 4) {
    x = alu(x, iters);
    if (do_store) store_quads(p, 0, quads, val);
    if (mode == 7) {
#pragma unroll
      for (int k = 0; k < 6; ++k) val.x ^= late[k].x;
    }
    if ((mode == 4 || mode == 9) && do_store) store_quads(const_cast<uint4*>(in) + (size_t)blockIdx.x * 384, 0, 384, val);
    if ((mode == 6 || mode == 7 || mode == 8) && do_store) store_quads(out + (size_t)gridDim.x * quads + (size_t)blockIdx.x * 384, 0, 384, val);
  } else if (mode == 1) {
    const int cut = quads * 47 / 100;
    if (do_store) store_quads(p, 0, cut, val);
    x = alu(x, iters);
    if (do_store) store_quads(p, cut, quads, val);
  } else if (mode == 2) {
    for (int k = 0; k < pieces; ++k) {
      x = alu(x, iters / pieces);
      if (do_store) store_quads(p, (int)((long)quads * k / pieces), (int)((long)quads * (k + 1) / pieces), val);
    }
  } else {
    if (do_store) store_quads(p, 0, quads, val);
    x = alu(x, iters);
  }
  if (x == 12345.678) {
    sink[0] = x;
    dyn_lds[threadIdx.x] = 1;
  }
}

template <typename F>
static double timeit(F launch) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<double> ms;
  for (int rep = 0; rep < 6; ++rep) {
    CK(hipEventRecord(a, 0));
    launch();
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float t;
    CK(hipEventElapsedTime(&t, a, b));
    if (rep > 0) ms.push_back(t);
  }
  std::sort(ms.begin(), ms.end());
  return ms[ms.size() / 2];
}

int main(int argc, char** argv) {
  const int E = 65536, quads = 5350;
  const size_t total = (size_t)E * quads * 16;
  uint4 *buf, *src;
  double* sink;
  CK(hipMalloc(reinterpret_cast<void**>(&buf), total + (size_t)E * 384 * 16));
  CK(hipMalloc(reinterpret_cast<void**>(&src), (size_t)E * 384 * 16));
  CK(hipMalloc(reinterpret_cast<void**>(&sink), 64));
  CK(hipMemset(buf, 0, total));
  CK(hipMemset(src, 1, (size_t)E * 384 * 16));
  for (int lds : {10240, 0}) {
    for (int iters : {0, 1000}) {
      auto run = [&](int mode, int pieces, int st) {
        return timeit([&] { hipLaunchKernelGGL(k_phase, dim3(E), dim3(64), lds, 0, buf, src, quads, iters, mode, pieces, st, sink); });
      };
      printf("lds %5d iters %5d | alu only %.3f | stores only %.3f | alu>store %.3f | 47%%>alu>53%% %.3f | x4 %.3f | x16 %.3f | "
             "store>alu %.3f | load>alu>store>store %.3f\n",
             lds, iters, run(0, 1, 0), iters ? -1.0 : run(0, 1, 1), run(0, 1, 1), run(1, 1, 1), run(2, 4, 1), run(2, 16, 1),
             run(3, 1, 1), run(4, 1, 1));
      printf("      dissect mode 4: load only %.3f | tail store only %.3f | late-consumed load + tail %.3f | load + tail elsewhere %.3f | "
             "cached load + tail %.3f\n", run(5, 1, 1), run(6, 1, 1), run(7, 1, 1), run(8, 1, 1), run(9, 1, 1));
      fflush(stdout);
    }
  }
  return 0;
}
