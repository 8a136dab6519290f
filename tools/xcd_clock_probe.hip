// xcd_clock_probe.hip -- development: are the per-XCD workgroup start offsets that tools/block_trace.py sees (mean start
// by blockIdx % 8: 0.2 0.6 2.0 2.0 4.5 5.2 3.3 3.8 us on one MI355X) real dispatch delays, or offsets between the XCDs'
// copies of the 100 MHz counter behind wall_clock64()?
//   1. ping-pong: workgroup 0 (XCD 0) and workgroup k (XCD k) of one 8-workgroup launch exchange flags through device
//      memory (agent-scope atomics) and stamp their clocks: offset_k = t_k - (t_0 + t_0') / 2, NTP style, minimum round
//      trip of R exchanges;
//   2. a 4096 x 128 launch whose workgroups stamp their start: mean start per XCD, raw and corrected by (1);
//   3. the same launch timed by HIP events (20 back to back): the launch cannot be shorter than its real start spread.
// Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/xcd_clock_probe.hip -o tools/bin/xcd_clock_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                  \
      exit(1);                                                                                 \
    }                                                                                          \
  } while (0)

#define ROUNDS 64

__device__ __forceinline__ int ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// mail[2 k] : workgroup 0 -> k, mail[2 k + 1] : k -> 0 (64-byte apart); out[k][r] = {t0, tk, t0'}
__global__ void __launch_bounds__(64) pingpong(int* mail, long long* out, int* xcc) {
  const int b = blockIdx.x;
  if (threadIdx.x != 0) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[b] = (int)(id & 0xf);
  if (b == 0) {
    for (int k = 1; k < (int)gridDim.x; ++k) {
      for (int r = 1; r <= ROUNDS; ++r) {
        const long long t0 = wall_clock64();
        st(mail + 32 * k, r);
        while (ld(mail + 32 * k + 16) != r) {}
        const long long t1 = wall_clock64();
        out[(k * ROUNDS + r - 1) * 3 + 0] = t0;
        out[(k * ROUNDS + r - 1) * 3 + 2] = t1;
      }
    }
  } else {
    for (int r = 1; r <= ROUNDS; ++r) {
      while (ld(mail + 32 * b) != r) {}
      const long long t = wall_clock64();
      st(mail + 32 * b + 16, r);
      out[(b * ROUNDS + r - 1) * 3 + 1] = t;
    }
  }
}

__global__ void __launch_bounds__(1024) stamp(long long* start, int* xcc, int spin) {
  if (threadIdx.x == 0) {
    start[blockIdx.x] = wall_clock64();
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = (int)(id & 0xf);
  }
  // keep the workgroup alive for `spin` clock ticks so that every one of them is resident at once, like the step kernel
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
}

// variants of the stamp kernel: what delays the first instruction of the step kernel's workgroups on some XCDs?
//   mode 1: a scalar load from a parameter block in device memory ahead of the stamp (the step kernel reads params->E and
//           the trace pointer first);  mode 2: private-array scratch (the step kernel spills 132 B per lane);
//   mode 3: both, then a 4 KB record per workgroup HBM -> LDS -> HBM with a second stamp (the load phase)
struct Params { int E; int pad[63]; long long* trace; };
extern __shared__ unsigned char dyn_lds[];
template <int MODE>
__global__ void __launch_bounds__(128) stamp_v(const Params* __restrict__ prm, long long* start, long long* loaded, uint4* rec, int spin) {
  long long* tr = start;
  int E = 4096;
  if (MODE & 1) {
    E = prm->E;
    tr = prm->trace;
  }
  volatile int priv[40];
  if ((MODE & 8) && spin == 123456) {  // mode 8: a private segment that no lane touches (what does ENABLING scratch cost a launch?)
    for (int i = 0; i < 40; ++i) priv[i] = threadIdx.x + i;
    int s = 0;
    for (int i = 0; i < 40; ++i) s += priv[(i + E) % 40];
    if (s == 12345) tr[0] = 0;
  }
  if (MODE & 2) {
    for (int i = 0; i < 40; ++i) priv[i] = threadIdx.x + i;
  }
  const int b = (blockIdx.x & 7) * (E >> 3) + (blockIdx.x >> 3);
  if (threadIdx.x == 0) tr[b] = wall_clock64();
  if (MODE & 4) {
    uint4* l = reinterpret_cast<uint4*>(dyn_lds);
    for (int q = threadIdx.x; q < 266; q += 128) l[q] = rec[(size_t)b * 422 + q];
    __syncthreads();
    if (threadIdx.x == 0) loaded[b] = wall_clock64();
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (MODE & 2) {
    int s = 0;
    for (int i = 0; i < 40; ++i) s += priv[(i + spin) % 40];
    if (s == 12345) tr[0] = 0;
  }
  if (MODE & 4) {
    uint4* l = reinterpret_cast<uint4*>(dyn_lds);
    for (int q = threadIdx.x; q < 422; q += 128) rec[(size_t)b * 422 + q] = l[q % 266];
  }
}

template <int MODE>
static void run_variant(const char* name, Params* dprm, long long* start, long long* loaded, uint4* rec) {
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(stamp_v<MODE>, dim3(4096), dim3(128), 9680, 0, dprm, start, loaded, rec, 1500);
  CK(hipDeviceSynchronize());
  std::vector<long long> s(4096), l(4096);
  CK(hipMemcpy(s.data(), start, 4096 * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(l.data(), loaded, 4096 * 8, hipMemcpyDeviceToHost));
  const long long s0 = *std::min_element(s.begin(), s.end());
  printf("%-44s mean start per XCC (us):", name);
  for (int x = 0; x < 8; ++x) {
    double a = 0;
    for (int j = 0; j < 512; ++j) a += (double)(s[x * 512 + j] - s0);
    printf(" %5.2f", a / 512 / 100.0);
  }
  if (MODE & 4) {
    printf("  | loaded - start:");
    for (int x = 0; x < 8; ++x) {
      double a = 0;
      for (int j = 0; j < 512; ++j) a += (double)(l[x * 512 + j] - s[x * 512 + j]);
      printf(" %5.2f", a / 512 / 100.0);
    }
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(stamp_v<MODE>, dim3(4096), dim3(128), 9680, 0, dprm, start, loaded, rec, 1500);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  | %.2f us per launch\n", ms * 5.0);
}

int main() {
  int* mail;
  long long* out;
  int* xcc;
  CK(hipMalloc(&mail, 32 * 8 * 4 * 2));
  CK(hipMemset(mail, 0, 32 * 8 * 4 * 2));
  CK(hipMalloc(&out, 8 * ROUNDS * 3 * 8));
  CK(hipMemset(out, 0, 8 * ROUNDS * 3 * 8));
  CK(hipMalloc(&xcc, 4096 * 4));
  hipLaunchKernelGGL(pingpong, dim3(8), dim3(64), 0, 0, mail, out, xcc);
  CK(hipDeviceSynchronize());
  std::vector<long long> h(8 * ROUNDS * 3);
  std::vector<int> hx(4096);
  CK(hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hx.data(), xcc, 8 * 4, hipMemcpyDeviceToHost));
  double offset_of_xcc[16] = {0};
  printf("ping-pong against workgroup 0 (XCC %d), clock ticks of 10 ns:\n", hx[0]);
  for (int k = 1; k < 8; ++k) {
    long long best_rt = 1ll << 60;
    double off = 0;
    for (int r = 0; r < ROUNDS; ++r) {
      const long long t0 = h[(k * ROUNDS + r) * 3], tk = h[(k * ROUNDS + r) * 3 + 1], t1 = h[(k * ROUNDS + r) * 3 + 2];
      if (t1 - t0 < best_rt) {
        best_rt = t1 - t0;
        off = (double)tk - 0.5 * (double)(t0 + t1);
      }
    }
    offset_of_xcc[hx[k]] = off;
    printf("  workgroup %d on XCC %d: min round trip %lld ticks, clock offset %+.1f ticks (%+.2f us)\n", k, hx[k], best_rt, off,
           off / 100.0);
  }
  long long* start;
  CK(hipMalloc(&start, 4096 * 8));
  for (int rep = 0; rep < 3; ++rep) {
    const int spin = 1500;  // 15 us
    for (int i = 0; i < (rep == 2 ? 20 : 1); ++i) hipLaunchKernelGGL(stamp, dim3(4096), dim3(128), 9680, 0, start, xcc, spin);
    CK(hipDeviceSynchronize());
    std::vector<long long> s(4096);
    CK(hipMemcpy(s.data(), start, 4096 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hx.data(), xcc, 4096 * 4, hipMemcpyDeviceToHost));
    const long long s0 = *std::min_element(s.begin(), s.end());
    double raw[16] = {0}, cor[16] = {0}, mx_raw = 0, mx_cor = -1e9, mn_cor = 1e9;
    int cnt[16] = {0};
    for (int b = 0; b < 4096; ++b) {
      const double r = (double)(s[b] - s0), c = r - offset_of_xcc[hx[b]];
      raw[hx[b]] += r;
      cor[hx[b]] += c;
      cnt[hx[b]] += 1;
      mx_raw = std::max(mx_raw, r);
      mx_cor = std::max(mx_cor, c);
      mn_cor = std::min(mn_cor, c);
    }
    printf("stamp launch %d (%s): start spread raw %.2f us, corrected %.2f us; mean start per XCC (us), raw | corrected:\n", rep,
           rep == 2 ? "last of 20 back to back" : "single", mx_raw / 100.0, (mx_cor - mn_cor) / 100.0);
    for (int x = 0; x < 8; ++x)
      if (cnt[x]) printf("   XCC %d (%4d workgroups; blockIdx %% 8 of the first: %d): %6.2f | %6.2f\n", x, cnt[x], -1, raw[x] / cnt[x] / 100.0,
                         (cor[x] / cnt[x] - mn_cor) / 100.0);
  }
  // blockIdx % 8 -> XCC map
  printf("blockIdx %% 8 -> XCC:");
  for (int b = 0; b < 8; ++b) printf(" %d", hx[b]);
  printf("\n");
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int spin : {0, 1500}) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(stamp, dim3(4096), dim3(128), 9680, 0, start, xcc, spin);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("200 back-to-back stamp launches (4096 x 128 threads, 9680 B LDS, spin %d ticks): %.2f us per launch\n", spin, ms * 5.0);
  }
  // dispatch cost against the workgroup shape: the same 8192 waves and the same LDS per wave as 4096 x 128, in fewer and
  // larger workgroups (would several replicas per workgroup launch faster?)
  for (int wpw : {2, 4, 8, 16}) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(stamp, dim3(8192 / wpw), dim3(64 * wpw), 4840 * wpw, 0, start, xcc, 0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel, %4d workgroups x %4d threads, %6d B LDS each: %.2f us per launch\n", 8192 / wpw, 64 * wpw, 4840 * wpw, ms * 5.0);
  }
  Params hp;
  hp.E = 4096;
  hp.trace = start;
  Params* dprm;
  CK(hipMalloc(&dprm, sizeof(Params)));
  CK(hipMemcpy(dprm, &hp, sizeof(Params), hipMemcpyHostToDevice));
  long long* loaded;
  CK(hipMalloc(&loaded, 4096 * 8));
  uint4* rec;
  CK(hipMalloc(&rec, (size_t)4096 * 422 * 16));
  CK(hipMemset(rec, 1, (size_t)4096 * 422 * 16));
  printf("variants (last of 20 back-to-back launches, 15 us spin each, XCD-contiguous replica mapping):\n");
  run_variant<0>("0 stamp first", dprm, start, loaded, rec);
  run_variant<1>("1 scalar load of the parameter block first", dprm, start, loaded, rec);
  run_variant<2>("2 scratch", dprm, start, loaded, rec);
  run_variant<3>("3 parameter block + scratch", dprm, start, loaded, rec);
  run_variant<8>("8 scratch enabled, never touched", dprm, start, loaded, rec);
  run_variant<12>("12 scratch enabled, never touched + record", dprm, start, loaded, rec);
  run_variant<4>("4 record in / out (4.2 KB in, 6.7 KB out)", dprm, start, loaded, rec);
  run_variant<7>("7 parameter block + scratch + record", dprm, start, loaded, rec);
  return 0;
}
