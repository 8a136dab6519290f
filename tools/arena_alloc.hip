// Development helper for tools/arena_probe.py: device allocations by allocators other than torch's, so that the launch
// time of the store-bound one-step-economy kernel can be compared per allocator.
//   kind 0: hipMalloc   1: hipExtMallocWithFlags(hipDeviceMallocContiguous)   2: hipMemCreate + hipMemMap (VMM), one
//   physical allocation, granularity = hipMemAllocationGranularityRecommended   3: VMM, mapped in 2 MiB-granule pieces
//   rounded from hipMemAllocationGranularityMinimum
// hipcc --offload-arch=gfx950 -shared -fPIC tools/arena_alloc.hip -o tools/bin/libarena_alloc.so
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

extern "C" __attribute__((visibility("default"))) void* arena_alloc(int kind, size_t n, int device, size_t* granularity_out) {
  void* p = nullptr;
  if (granularity_out) *granularity_out = 0;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  if (kind == 0) return hipMalloc(&p, n) == hipSuccess ? p : nullptr;
  if (kind == 1) return hipExtMallocWithFlags(&p, n, hipDeviceMallocContiguous) == hipSuccess ? p : nullptr;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, kind == 2 ? hipMemAllocationGranularityRecommended
                                                             : hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) {
    fprintf(stderr, "arena_alloc: hipMemGetAllocationGranularity failed\n");
    return nullptr;
  }
  if (kind == 3 && gran < (2u << 20)) gran = 2u << 20;
  if (granularity_out) *granularity_out = gran;
  const size_t total = (n + gran - 1) / gran * gran;
  hipDeviceptr_t va = nullptr;
  if (hipMemAddressReserve(&va, total, gran, nullptr, 0) != hipSuccess) { fprintf(stderr, "arena_alloc: reserve failed\n"); return nullptr; }
  size_t piece = kind == 2 ? total : (size_t)64 * gran;  // kind 3: 128 MiB pieces (ARENA_PIECE_MB overrides), each its own physical handle
  if (kind == 3 && getenv("ARENA_PIECE_MB")) piece = ((size_t)atol(getenv("ARENA_PIECE_MB")) << 20) / gran * gran;
  if (piece < gran) piece = gran;
  for (size_t off = 0; off < total; off += piece) {
    const size_t len = off + piece <= total ? piece : total - off;
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, len, &prop, 0) != hipSuccess) { fprintf(stderr, "arena_alloc: hipMemCreate(%zu) failed\n", len); return nullptr; }
    if (hipMemMap((char*)va + off, len, 0, h, 0) != hipSuccess) { fprintf(stderr, "arena_alloc: hipMemMap failed\n"); return nullptr; }
    (void)hipMemRelease(h);  // (the mapping keeps the memory alive)
  }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(va, total, &acc, 1) != hipSuccess) { fprintf(stderr, "arena_alloc: hipMemSetAccess failed\n"); return nullptr; }
  return va;
}
