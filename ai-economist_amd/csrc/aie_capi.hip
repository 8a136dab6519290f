// aie_capi.hip -- the C ABI declared in include/aie.h: handle management, tensor table,
// host<->device staging, and the kernel launches.  No torch types anywhere.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "aie_kernels.hip"
#include "aie_kernels_ose.hip"
#include "aie_kernels_saez.hip"
#include "aie_kernels_covid.hip"  // last: switches FP contraction off for the rest of the TU
#include "aie_jit.h"

struct aie_env {
  aie_params P{};
  aie_params* d_params;  // device copy of P (kernel parameter block)
  aie_tensor_table tt;
  uint8_t* arena;
  bool owns_arena;
  size_t vmm_total, vmm_piece;  // owns_arena: the arena is a virtual range mapped in pieces (aie_arena_alloc), else 0
  int device;
  size_t lds;
  int spec;        // >= 0: the compile-time instance aie_step_kernel_spec<spec> runs this configuration; -1: generic
  int spec_match;  // the instance that matches the configuration (what AIE_KERNEL_AUTO selects), or -1
  hipModule_t jit_mod;              // aie_specialize: the code object compiled for this configuration, or nullptr
  hipFunction_t jit_step, jit_reset;  // its entry points; the environment then runs as instance AIE_KERNEL_INSTANCE_JIT
  std::shared_ptr<aie_jit::Job> jit_job;  // the background specialisation aie_create started, until its code is loaded
  int pinned_generic;    // aie_select_step_kernel(AIE_KERNEL_GENERIC): stay on the generic kernel, whatever becomes ready
  float* rew_log;        // aie_set_reward_log: caller's ring of n_slots step slots, or nullptr
  int32_t rew_log_slots, rew_log_epoch;  // epoch: bumped by every aie_set_reward_log call (the replicas' slot counters --
                                         // record field o_rew_slot -- restart at 0 when they see a new one)
  int cv_taps_f32;       // COVID: every uploaded filter tap is a float32 value (aie_upload checks): the window-sum kernel
                         // then keeps its LDS tap table in float32
  int log_active;        // aie_set_dense_log_active: the dense-log replicas record events (default) or run with the rest
  char err[512];
};

static thread_local char g_create_err[512] = "";
static inline void aie_jit_poll(aie_env* env);  // adopts the background specialisation once it is ready (below)
static bool aie_jit_eligible(const aie_env* env);
static int aie_jit_request(aie_env* env);

#define AIE_DEV_API __attribute__((visibility("default")))

#define AIE_HIP_CHECK(env, expr)                                                          \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      snprintf((env)->err, sizeof((env)->err), "%s failed: %s", #expr, hipGetErrorString(_e)); \
      return AIE_E_HIP;                                                                   \
    }                                                                                     \
  } while (0)

// Arenas the library allocates itself.  Large ones (AIE_ARENA_VMM_MIN_MB, default 1024 MiB, and up) are a virtual range
// backed by physical allocations of 64 MiB each (hipMemCreate / hipMemMap) instead of one hipMalloc: the store-bound
// one-step-economy launch over its 7 GB arena takes 1.40 ms that way against 1.63 - 1.66 ms on one hipMalloc / torch
// allocation / single VMM allocation, in every one of 8 + 3 fresh processes (profiles/r04_c5_alloc.json; pieces of 2,
// 16, 128, 256, 1024 MiB: 1.57, 1.47, 1.43, 1.48, 1.66 ms -- one big physical allocation lands on the memory channels
// less evenly than many medium ones).  AIE_ARENA_PIECE_MB overrides the piece size, 0 = always hipMalloc.
// Round 5: the piece size does not explain the box-to-box spread of the one-step-economy launch (fresh processes with 16 /
// 64 / 128 MiB pieces: 1.53 / 1.61 / 1.49 ms on one box, 1.616 / 1.622 / 1.633 ms on the next -- tools/c5_piece_experiment.sh,
// DESIGN.md section 4); a create-time store probe that re-mapped the arena with several piece sizes and kept the fastest was
// tried and dropped (a map / unmap / re-map cycle of the 7 GB range faulted on this driver, and the probe had nothing
// consistent to find).  64 MiB stays the default.
static uint8_t* aie_arena_alloc_pieces(int device, size_t bytes, size_t piece_mb, size_t* vmm_total, size_t* vmm_piece);
static uint8_t* aie_arena_alloc(int device, size_t bytes, size_t* vmm_total, size_t* vmm_piece) {
  const char* e_min = getenv("AIE_ARENA_VMM_MIN_MB");
  const char* e_piece = getenv("AIE_ARENA_PIECE_MB");
  const size_t min_mb = e_min ? (size_t)atol(e_min) : 1024;
  return aie_arena_alloc_pieces(device, bytes, bytes >= (min_mb << 20) ? (e_piece ? (size_t)atol(e_piece) : 64) : 0, vmm_total, vmm_piece);
}
static uint8_t* aie_arena_alloc_pieces(int device, size_t bytes, size_t piece_mb, size_t* vmm_total, size_t* vmm_piece) {
  *vmm_total = *vmm_piece = 0;
  void* p = nullptr;
  if (piece_mb > 0) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) == hipSuccess && gran > 0) {
      if (gran < ((size_t)2 << 20)) gran = (size_t)2 << 20;
      size_t piece = (piece_mb << 20) / gran * gran;
      if (piece < gran) piece = gran;
      const size_t total = (bytes + piece - 1) / piece * piece;
      hipDeviceptr_t va = nullptr;
      if (hipMemAddressReserve(&va, total, gran, nullptr, 0) == hipSuccess) {
        size_t off = 0;
        for (; off < total; off += piece) {
          hipMemGenericAllocationHandle_t h;
          if (hipMemCreate(&h, piece, &prop, 0) != hipSuccess) break;
          const hipError_t me = hipMemMap(static_cast<char*>(va) + off, piece, 0, h, 0);
          (void)hipMemRelease(h);  // (the mapping keeps the memory alive)
          if (me != hipSuccess) break;
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if (off >= total && hipMemSetAccess(va, total, &acc, 1) == hipSuccess) {
          *vmm_total = total;
          *vmm_piece = piece;
          return static_cast<uint8_t*>(va);
        }
        for (size_t q = 0; q < off; q += piece) (void)hipMemUnmap(static_cast<char*>(va) + q, piece);  // give up: plain hipMalloc below
        (void)hipMemAddressFree(va, total);
      }
    }
    (void)hipGetLastError();
  }
  return hipMalloc(&p, bytes) == hipSuccess ? static_cast<uint8_t*>(p) : nullptr;
}
static void aie_arena_free(uint8_t* arena, size_t vmm_total, size_t vmm_piece) {
  if (!arena) return;
  if (vmm_total) {
    for (size_t q = 0; q < vmm_total; q += vmm_piece) (void)hipMemUnmap(arena + q, vmm_piece);
    (void)hipMemAddressFree(arena, vmm_total);
  } else {
    (void)hipFree(arena);
  }
}

// cells start with "no house" (owner byte 0xff); everything else zero
__global__ void aie_init_cells_kernel(const aie_params P, uint8_t* __restrict__ arena) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)P.E * P.HW) return;
  const int e = (int)(q / P.HW), cell = (int)(q - (int64_t)e * P.HW);
  reinterpret_cast<uint32_t*>(arena + P.a_records + (int64_t)e * P.rec_bytes + P.o_cells)[cell] = 0x00ff0000u;
}

// packs the static layout flags into byte 3 of every cell word
__global__ void aie_set_flags_kernel(const aie_params P, uint8_t* __restrict__ arena,
                                     const uint8_t* __restrict__ flags, int shared) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= (int64_t)P.E * P.HW) return;
  const int e = (int)(q / P.HW), cell = (int)(q - (int64_t)e * P.HW);
  uint32_t* w = reinterpret_cast<uint32_t*>(arena + P.a_records + (int64_t)e * P.rec_bytes + P.o_cells) + cell;
  const uint32_t fl = flags[shared ? cell : q];
  *w = (*w & 0x00ffffffu) | (fl << 24);
}

// The replicas' source-double lists (record fields o_src_n / o_src_list, aie_layout.h) from the cells' flag bytes in
// device memory: one wavefront per replica, the order of aie::build_src_list (Wood cells ascending, then Stone cells).
// Launched wherever the flags change outside a reset (aie_set_layout, aie_upload of the cells).
__global__ void __launch_bounds__(64) aie_src_list_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena) {
  const aie_params& P = *params;
  const int e = (int)blockIdx.x, lane = (int)threadIdx.x, HW = P.HW;
  if (!P.o_src_list || e >= P.E) return;
  uint8_t* rec = arena + P.a_records + (int64_t)e * P.rec_bytes;
  const uint8_t* cb = rec + P.o_cells;
  uint16_t* lst = reinterpret_cast<uint16_t*>(rec + P.o_src_list);
  for (int k = lane; k < AIE_SRC_CAP; k += 64) lst[k] = 0;
  int base = 0;
  for (int rs = 0; rs < 2; ++rs) {
    const uint32_t bit = rs == 0 ? AIE_CELL_WOOD_SRC : AIE_CELL_STONE_SRC;
    for (int q0 = 0; q0 < HW; q0 += 64) {
      const int q = q0 + lane;
      const bool on = q < HW && (cb[4 * q + 3] & bit);
      const uint64_t mask = __ballot(on);
      const int slot = base + __popcll(mask & ((1ull << lane) - 1ull));
      if (on && slot < AIE_SRC_CAP) lst[slot] = (uint16_t)(rs * HW + q);
      base += __popcll(mask);
    }
  }
  if (lane == 0) *reinterpret_cast<int32_t*>(rec + P.o_src_n) = base;
}
static int aie_rebuild_src_lists(aie_env* env) {
  if (env->P.c.scenario != AIE_SCN_GTB || !env->P.o_src_list) return AIE_OK;
  hipLaunchKernelGGL(aie_src_list_kernel, dim3((unsigned)env->P.E), dim3(64), 0, 0, env->d_params, env->arena);
  AIE_HIP_CHECK(env, hipGetLastError());
  AIE_HIP_CHECK(env, hipDeviceSynchronize());
  return AIE_OK;
}

// Workgroups of `lds` dynamic LDS bytes a gfx950 CU holds at once: 160 KB, allocated in 1280-byte granules, at most
// 16 workgroups of two waves (8 waves per SIMD).
static inline int aie_workgroups_per_cu(size_t lds) {
  const size_t granules = (lds + 1279) / 1280;
  const size_t fit = granules ? (size_t)128 / granules : 16;
  return (int)(fit < 16 ? fit : 16);
}

extern "C" {

int aie_sizeof_config(void) { return (int)sizeof(aie_config); }

int aie_arena_info(const aie_env* env, int64_t* bytes, int32_t* allocator, int64_t* piece_bytes) {
  if (!env) return AIE_E_INVALID;
  if (bytes) *bytes = env->P.arena_bytes;
  if (allocator) *allocator = !env->owns_arena ? AIE_ARENA_CALLER : env->vmm_total ? AIE_ARENA_VMM : AIE_ARENA_HIPMALLOC;
  if (piece_bytes) *piece_bytes = (int64_t)env->vmm_piece;
  return AIE_OK;
}

int64_t aie_arena_bytes(const aie_config* cfg) {
  aie_params P;
  int rc = aie_build_params(cfg, &P, nullptr, g_create_err, sizeof(g_create_err));
  if (rc != AIE_OK) return rc;
  return P.arena_bytes;
}

int aie_create(const aie_config* cfg, int device, void* arena, int64_t arena_bytes, aie_env** out) {
  if (!cfg || !out) {
    snprintf(g_create_err, sizeof(g_create_err), "null argument");
    return AIE_E_INVALID;
  }
  aie_env* env = new aie_env();  // (value-initialised: every plain member zero)
  int rc = aie_build_params(cfg, &env->P, &env->tt, g_create_err, sizeof(g_create_err));
  if (rc != AIE_OK) { delete env; return rc; }
  env->device = device;
  env->spec = -1;
  env->log_active = 1;
  env->cv_taps_f32 = 1;  // (the arena starts zeroed)
  if (cfg->scenario != AIE_SCN_COVID) {  // a compile-time instance exists for exactly this parameter block?
    std::vector<aie_params> norm(1, env->P);  // (heap: the block is ~10 KB; not static: aie_create may run on several threads)
    aie_spec_normalize(&norm[0]);
    for (int k = 0; k < AIE_N_SPECS; ++k)
      if (memcmp(&norm[0], aie_spec_table[k], sizeof(aie_params)) == 0) env->spec = k;
  }
  env->spec_match = env->spec;
  const bool covid = cfg->scenario == AIE_SCN_COVID;
  const bool ose = cfg->scenario == AIE_SCN_ONE_STEP_ECONOMY || covid;  // map-less: no cell words to initialise
  env->lds = covid ? 0 : cfg->scenario == AIE_SCN_ONE_STEP_ECONOMY ? aie::ose_lds_bytes(env->P) : aie::lds_bytes(env->P);
  // a gfx950 workgroup may take the CU's whole 160 KB of LDS (hipDeviceProp.sharedMemPerBlock = 163 840; checked on the
  // device: launches with 65 ... 160 KB of dynamic LDS run without any function attribute)
  if (env->lds + (covid ? 0 : aie::layout_gen_lds_bytes(env->P)) > 160 * 1024) {
    snprintf(g_create_err, sizeof(g_create_err),
             "per-replica working set (%zu B of LDS) exceeds 160 KiB: reduce max_num_orders / world size", env->lds);
    delete env;
    return AIE_E_UNSUPPORTED;
  }
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) {
    snprintf(g_create_err, sizeof(g_create_err), "hipSetDevice(%d): %s", device, hipGetErrorString(he));
    delete env;
    return AIE_E_HIP;
  }
  if (arena) {
    if (arena_bytes < env->P.arena_bytes || (reinterpret_cast<uintptr_t>(arena) & 255u)) {
      snprintf(g_create_err, sizeof(g_create_err), "arena too small (%lld < %lld) or not 256-byte aligned",
               (long long)arena_bytes, (long long)env->P.arena_bytes);
      delete env;
      return AIE_E_INVALID;
    }
    env->arena = static_cast<uint8_t*>(arena);
    env->owns_arena = false;
  } else {
    env->arena = aie_arena_alloc(device, (size_t)env->P.arena_bytes, &env->vmm_total, &env->vmm_piece);
    if (!env->arena) {
      snprintf(g_create_err, sizeof(g_create_err), "arena allocation of %lld bytes failed", (long long)env->P.arena_bytes);
      delete env;
      return AIE_E_NOMEM;
    }
    env->owns_arena = true;
  }
  he = hipMemset(env->arena, 0, (size_t)env->P.arena_bytes);
  if (he == hipSuccess && !ose) {
    const int64_t tot = (int64_t)env->P.E * env->P.HW;
    hipLaunchKernelGGL(aie_init_cells_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, env->P, env->arena);
    he = hipDeviceSynchronize();
  }
  if (he == hipSuccess && env->P.saez_stride) {  // elas_t = elas_tm1 = 0.5 at construction (redistribution.py:263-266)
    std::vector<double> rows((size_t)env->P.E * 2, 0.5);
    he = hipMemcpy2D(env->arena + env->P.a_saez + AIE_SAEZ_OFF_ELAS, (size_t)env->P.saez_stride, rows.data(), 16, 16,
                     (size_t)env->P.E, hipMemcpyHostToDevice);
  }
  if (he != hipSuccess) {
    snprintf(g_create_err, sizeof(g_create_err), "arena init: %s", hipGetErrorString(he));
    if (env->owns_arena) aie_arena_free(env->arena, env->vmm_total, env->vmm_piece);
    delete env;
    return AIE_E_HIP;
  }
  he = hipMalloc(reinterpret_cast<void**>(&env->d_params), sizeof(aie_params));
  if (he == hipSuccess) he = hipMemcpy(env->d_params, &env->P, sizeof(aie_params), hipMemcpyHostToDevice);
  if (he != hipSuccess) {
    snprintf(g_create_err, sizeof(g_create_err), "parameter block: %s", hipGetErrorString(he));
    if (env->owns_arena) aie_arena_free(env->arena, env->vmm_total, env->vmm_piece);
    delete env;
    return AIE_E_HIP;
  }
  for (int i = 0; i < env->tt.n; ++i) env->tt.t[i].data = env->arena + env->tt.t[i].arena_offset;
  // no compile-time instance for this configuration's family: kernels specialised on it are compiled (or fetched from
  // the cache) in the background, the generic kernel runs until they are ready.  AIE_JIT_AUTO=0 switches this off
  // (aie_specialize still does it on request).
  const char* auto_jit = getenv("AIE_JIT_AUTO");
  if (env->spec < 0 && aie_jit_eligible(env) && !(auto_jit && auto_jit[0] == '0')) (void)aie_jit_request(env);
  *out = env;
  return AIE_OK;
}

int aie_destroy(aie_env* env) {
  if (!env) return AIE_OK;
  (void)hipSetDevice(env->device);
  (void)hipDeviceSynchronize();
  if (env->owns_arena && env->arena) aie_arena_free(env->arena, env->vmm_total, env->vmm_piece);
  if (env->d_params) (void)hipFree(env->d_params);
  if (env->jit_mod) (void)hipModuleUnload(env->jit_mod);
  delete env;
  return AIE_OK;
}

const char* aie_last_error(const aie_env* env) { return env ? env->err : g_create_err; }

int aie_num_tensors(const aie_env* env) { return env ? env->tt.n : AIE_E_INVALID; }

int aie_tensor_at(const aie_env* env, int index, aie_tensor_desc* out) {
  if (!env || !out || index < 0 || index >= env->tt.n) return AIE_E_INVALID;
  *out = env->tt.t[index];
  return AIE_OK;
}

static const aie_tensor_desc* find_tensor(const aie_env* env, const char* name) {
  for (int i = 0; i < env->tt.n; ++i)
    if (strcmp(env->tt.t[i].name, name) == 0) return &env->tt.t[i];
  return nullptr;
}

int aie_get_tensor(const aie_env* env, const char* name, aie_tensor_desc* out) {
  if (!env || !name || !out) return AIE_E_INVALID;
  const aie_tensor_desc* d = find_tensor(env, name);
  if (!d) {
    snprintf(const_cast<aie_env*>(env)->err, sizeof(env->err), "no tensor named '%s'", name);
    return AIE_E_NOTFOUND;
  }
  *out = *d;
  return AIE_OK;
}

// Generic strided host<->device copy of one tensor through a host bounce buffer.
static int copy_tensor(aie_env* env, const char* name, void* host, int64_t bytes, bool upload) {
  if (!env || !name || !host) return AIE_E_INVALID;
  const aie_tensor_desc* d = find_tensor(env, name);
  if (!d) {
    snprintf(env->err, sizeof(env->err), "no tensor named '%s'", name);
    return AIE_E_NOTFOUND;
  }
  const int es = aie__dtype_size(d->dtype);
  int64_t count = 1, span = es;
  for (int i = 0; i < d->ndim; ++i) {
    count *= d->shape[i];
    span += (d->shape[i] - 1) * d->stride[i];
  }
  if (bytes != count * es) {
    snprintf(env->err, sizeof(env->err), "'%s': expected %lld bytes, got %lld", name, (long long)(count * es),
             (long long)bytes);
    return AIE_E_INVALID;
  }
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  std::vector<uint8_t> tmp((size_t)span);
  uint8_t* dev = env->arena + d->arena_offset;
  AIE_HIP_CHECK(env, hipDeviceSynchronize());
  AIE_HIP_CHECK(env, hipMemcpy(tmp.data(), dev, (size_t)span, hipMemcpyDeviceToHost));
  int64_t idx[6] = {0, 0, 0, 0, 0, 0};
  uint8_t* h = static_cast<uint8_t*>(host);
  for (int64_t k = 0; k < count; ++k) {
    int64_t off = 0;
    for (int i = 0; i < d->ndim; ++i) off += idx[i] * d->stride[i];
    if (upload) memcpy(tmp.data() + off, h + k * es, (size_t)es);
    else memcpy(h + k * es, tmp.data() + off, (size_t)es);
    for (int i = d->ndim - 1; i >= 0; --i) {
      if (++idx[i] < d->shape[i]) break;
      idx[i] = 0;
    }
  }
  if (upload) AIE_HIP_CHECK(env, hipMemcpy(dev, tmp.data(), (size_t)span, hipMemcpyHostToDevice));
  return AIE_OK;
}

int aie_upload(aie_env* env, const char* name, const void* host, int64_t bytes) {
  const int rc = copy_tensor(env, name, const_cast<void*>(host), bytes, true);
  if (rc == AIE_OK && (strcmp(name, "cells") == 0 || strcmp(name, "cell_flags") == 0))
    return aie_rebuild_src_lists(env);  // the regeneration's source doubles follow the flags (o_src_list)
  if (rc == AIE_OK && env->P.c.scenario == AIE_SCN_COVID && strcmp(name, "model_unemp_conv_filters") == 0) {
    // float32-valued taps (the reference's) let the window-sum kernel keep its LDS tap table in float32
    const double* taps = static_cast<const double*>(host);
    env->cv_taps_f32 = 1;
    for (int64_t q = 0; q < bytes / 8; ++q)
      if ((double)(float)taps[q] != taps[q]) env->cv_taps_f32 = 0;
  }
  if (rc == AIE_OK && env->P.c.scenario == AIE_SCN_COVID &&
      (strcmp(name, "model_stringency_level_history_0") == 0 || strcmp(name, "model_unemp_conv_filters") == 0)) {
    // what every reset derives from these tables is derived once, here (history-format image, the pre-episode change
    // events, the first step's filter sums -- which need the taps as well: either upload refreshes them)
    hipLaunchKernelGGL(aie_covid_prepare_kernel, dim3(1), dim3(AIE_NT), 0, 0, env->d_params, env->arena);
    AIE_HIP_CHECK(env, hipGetLastError());
    AIE_HIP_CHECK(env, hipDeviceSynchronize());
  }
  return rc;
}
int aie_download(aie_env* env, const char* name, void* host, int64_t bytes) {
  return copy_tensor(env, name, host, bytes, false);
}

int aie_set_layout(aie_env* env, const uint8_t* stone_src, const uint8_t* wood_src, const uint8_t* water) {
  if (!env) return AIE_E_INVALID;
  if (env->P.c.scenario != AIE_SCN_GTB) return AIE_OK;  // map-less scenario: nothing to set
  if (!stone_src || !wood_src) return AIE_E_INVALID;
  const aie_params& P = env->P;
  const int shared = P.c.shared_layout ? 1 : 0;
  const int64_t cnt = (shared ? 1 : (int64_t)P.E) * P.HW;
  std::vector<uint8_t> fl((size_t)cnt);
  for (int64_t q = 0; q < cnt; ++q)
    fl[(size_t)q] = (uint8_t)(((water && water[q]) ? AIE_CELL_WATER : 0u) | (stone_src[q] ? AIE_CELL_STONE_SRC : 0u) |
                              (wood_src[q] ? AIE_CELL_WOOD_SRC : 0u));
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  uint8_t* dfl = nullptr;
  AIE_HIP_CHECK(env, hipMalloc(reinterpret_cast<void**>(&dfl), (size_t)cnt));
  AIE_HIP_CHECK(env, hipMemcpy(dfl, fl.data(), (size_t)cnt, hipMemcpyHostToDevice));
  const int64_t tot = (int64_t)P.E * P.HW;
  hipLaunchKernelGGL(aie_set_flags_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, 0, P, env->arena, dfl, shared);
  AIE_HIP_CHECK(env, hipGetLastError());
  AIE_HIP_CHECK(env, hipDeviceSynchronize());
  AIE_HIP_CHECK(env, hipFree(dfl));
  if (const int rc = aie_rebuild_src_lists(env)) return rc;
  if (aie__shared_src_list(&P.c)) {
    // the regeneration's source doubles, once for the whole batch (aie_params.a_src_list): double d of a step's 2 H W
    // np.random.rand values targets Wood cell d (d < H W) or Stone cell d - H W (layout_from_file.py:394-403)
    struct { int32_t count, pad[3]; uint16_t d[AIE_SRC_CAP]; } lst;
    memset(&lst, 0, sizeof(lst));
    for (int r = 0; r < 2; ++r) {
      const uint8_t* plane = r == 0 ? wood_src : stone_src;
      for (int cell = 0; cell < P.HW; ++cell)
        if (plane[cell]) {
          if (lst.count < AIE_SRC_CAP) lst.d[lst.count] = (uint16_t)(r * P.HW + cell);
          lst.count += 1;
        }
    }
    AIE_HIP_CHECK(env, hipMemcpy(env->arena + P.a_src_list, &lst, sizeof(lst), hipMemcpyHostToDevice));
  }
  return AIE_OK;
}

static int aie_seed_impl(aie_env* env, uint64_t base_seed, void* stream) {
  if (env->P.c.scenario == AIE_SCN_COVID) return AIE_OK;  // the COVID simulation draws no random numbers
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  hipLaunchKernelGGL(aie_seed_kernel, dim3((unsigned)((env->P.E + 63) / 64)), dim3(64), 0,
                     static_cast<hipStream_t>(stream), env->P, env->arena, base_seed);
  AIE_HIP_CHECK(env, hipGetLastError());
  return AIE_OK;
}
int aie_seed(aie_env* env, uint32_t base_seed, void* stream) {
  if (!env) return AIE_E_INVALID;
  return aie_seed_impl(env, (uint64_t)base_seed, stream);
}
int aie_seed_fast(aie_env* env, uint64_t seed, int64_t global_env_offset, void* stream) {
  if (!env) return AIE_E_INVALID;
  if (env->P.c.scenario != AIE_SCN_COVID && env->P.c.rng_mode != AIE_RNG_FAST) {
    snprintf(env->err, sizeof(env->err), "aie_seed_fast: this environment was created with rng_mode = AIE_RNG_NUMPY (the "
             "generator is part of the record layout; create it with rng_mode = AIE_RNG_FAST)");
    return AIE_E_UNSUPPORTED;
  }
  return aie_seed_impl(env, seed + (uint64_t)global_env_offset, stream);
}

int aie_set_rng_state(aie_env* env, const uint32_t* key, const int32_t* pos) {
  if (!env || !key || !pos) return AIE_E_INVALID;
  if (env->P.c.scenario == AIE_SCN_COVID) {
    snprintf(env->err, sizeof(env->err), "the COVID scenario has no random stream");
    return AIE_E_UNSUPPORTED;
  }
  const int64_t E = env->P.E;
  int rc = aie_upload(env, "mt", key, E * aie__rng_state_words(&env->P.c) * 4);
  if (rc != AIE_OK) return rc;
  rc = aie_upload(env, "mt_pos", pos, E * 4);
  if (rc != AIE_OK) return rc;
  std::vector<int32_t> z((size_t)E, 0);
  std::vector<double> zd((size_t)E, 0.0);
  rc = aie_upload(env, "mt_has_gauss", z.data(), E * 4);
  if (rc != AIE_OK) return rc;
  return aie_upload(env, "mt_gauss", zd.data(), E * 8);
}

// gather-trade-build reset: the compile-time instance of the environment's configuration if it has one
static void aie_launch_gtb_reset_only(aie_env* env, const uint8_t* d_mask, int keep_rewards, void* stream);
static void aie_launch_gtb_reset(aie_env* env, const uint8_t* d_mask, int keep_rewards, void* stream) {
  aie_launch_gtb_reset_only(env, d_mask, keep_rewards, stream);
  if (aie__layout_staged(&env->P.c)) {  // generated layouts in the counter-stream mode: drawn ahead of their resets
    const int threshold = env->P.E >= 4 ? (int)(env->P.E / 4) : 1;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(aie_layout_decide_kernel, dim3(1), dim3(1), 0, st, env->d_params, env->arena, threshold);
    hipLaunchKernelGGL(aie_layout_refill_kernel, dim3((unsigned)env->P.E), dim3(LG_NW * AIE_NT),
                       env->lds + aie::layout_gen_lds_bytes(env->P), st, env->d_params, env->arena);
  }
}
static void aie_launch_gtb_reset_only(aie_env* env, const uint8_t* d_mask, int keep_rewards, void* stream) {
  // LG_NW wavefronts per replica when the reset draws a new source layout (aie_kernels.hip: layout_generate), else one
  const dim3 g((unsigned)env->P.E), b(env->P.c.layout_gen != AIE_LAYOUT_FIXED ? LG_NW * AIE_NT : AIE_NT);
  const size_t lds = env->lds + aie::layout_gen_lds_bytes(env->P);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (env->spec == AIE_KERNEL_INSTANCE_JIT && env->P.dev_skip_mask == 0) {
    const aie_params* dp = env->d_params;
    uint8_t* ar = env->arena;
    void* args[] = {&dp, &ar, &d_mask, &keep_rewards};
    (void)hipModuleLaunchKernel(env->jit_reset, g.x, 1, 1, b.x, 1, 1, (unsigned)lds, st, args, nullptr);
    return;
  }
#define AIE_SPEC_LAUNCH_RESET(K) \
  case K: hipLaunchKernelGGL(aie_reset_kernel_spec<K>, g, b, lds, st, env->d_params, env->arena, d_mask, keep_rewards); return;
  if (env->spec >= 0 && env->P.dev_skip_mask == 0) {
    switch (env->spec) {
      AIE_SPEC_LIST_GTB(AIE_SPEC_LAUNCH_RESET)
      default: break;
    }
  }
#undef AIE_SPEC_LAUNCH_RESET
  if (env->P.c.layout_gen != AIE_LAYOUT_FIXED)
    hipLaunchKernelGGL(aie_reset_kernel_layout, g, b, lds, st, env->d_params, env->arena, d_mask, keep_rewards);
  else
    hipLaunchKernelGGL(aie_reset_kernel, g, b, lds, st, env->d_params, env->arena, d_mask, keep_rewards);
}

int aie_reset(aie_env* env, const uint8_t* d_env_mask, void* stream) {
  if (!env) return AIE_E_INVALID;
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  // the one place where a finished background specialisation is adopted without being asked for (an episode boundary;
  // hipModuleLoadData synchronises, so never from aie_step and never on a stream that is being captured)
  if (env->jit_job) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs == hipStreamCaptureStatusNone)
      aie_jit_poll(env);
    else
      (void)hipGetLastError();
  }
  if (env->P.c.scenario == AIE_SCN_COVID)
    hipLaunchKernelGGL(aie_covid_reset_kernel, dim3((unsigned)env->P.E), dim3(AIE_NT), 0,
                       static_cast<hipStream_t>(stream), env->d_params, env->arena, d_env_mask, 0);
  else if (env->P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY)
    hipLaunchKernelGGL(aie_ose_reset_kernel, dim3((unsigned)env->P.E), dim3(OSE_NT), env->lds,
                       static_cast<hipStream_t>(stream), env->d_params, env->arena, d_env_mask);
  else
    aie_launch_gtb_reset(env, d_env_mask, 0, stream);
  AIE_HIP_CHECK(env, hipGetLastError());
  return AIE_OK;
}

static int aie_step_impl(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, void* stream,
                         const NextActions& next);

int aie_set_reward_log(aie_env* env, float* d_log, int32_t n_slots) {
  if (!env) return AIE_E_INVALID;
  if (d_log && n_slots < 1) {
    snprintf(env->err, sizeof(env->err), "aie_set_reward_log: needs n_slots >= 1");
    return AIE_E_INVALID;
  }
  env->rew_log = d_log;
  env->rew_log_slots = d_log ? n_slots : 0;
  env->rew_log_epoch += 1;  // (the arena starts zeroed and this starts at 1: never equal to a fresh record's epoch)
  // the kernels read the log's descriptor from the device-side parameter block: launches already captured in a hipGraph
  // follow this call too (the copy is ordered behind whatever the device is running)
  env->P.rew_log = env->rew_log;
  env->P.rew_slots = env->rew_log_slots;
  env->P.rew_epoch = env->rew_log_epoch;
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  AIE_HIP_CHECK(env, hipDeviceSynchronize());
  AIE_HIP_CHECK(env, hipMemcpy(reinterpret_cast<uint8_t*>(env->d_params) + offsetof(aie_params, rew_log), &env->P.rew_log,
                               offsetof(aie_params, rew_epoch) + sizeof(int32_t) - offsetof(aie_params, rew_log),
                               hipMemcpyHostToDevice));
  return AIE_OK;
}

int aie_step(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, void* stream) {
  return aie_step_impl(env, d_actions_a, d_actions_p, stream, NextActions{});
}

int aie_step_range(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, int32_t comp_lo, int32_t comp_hi,
                   int32_t phases, void* stream) {
  if (!env) return AIE_E_INVALID;
  const aie_params& P = env->P;
  if (P.c.scenario != AIE_SCN_GTB || P.saez_stride || (P.ev_replicas > 0 && env->log_active)) {
    snprintf(env->err, sizeof(env->err), "aie_step_range: gather-trade-build scenarios without tax_model \"saez\", and not while a "
             "dense-log replica records");
    return AIE_E_UNSUPPORTED;
  }
  if (comp_lo < 0 || comp_hi < comp_lo || comp_hi > P.c.n_components || phases < 0 || phases > 31 ||
      ((phases & AIE_STEP_OBSERVE) && (phases & ~(AIE_STEP_OBSERVE | AIE_STEP_REBASE | AIE_STEP_RETAX))) ||
      ((phases & (AIE_STEP_REBASE | AIE_STEP_RETAX)) && !(phases & AIE_STEP_OBSERVE))) {
    snprintf(env->err, sizeof(env->err), "aie_step_range: components [%d, %d) of %d, phases %d", comp_lo, comp_hi, P.c.n_components, phases);
    return AIE_E_INVALID;
  }
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  NextActions next{};
  next.E = (int32_t)P.E;
  next.comp_lo = comp_lo;
  next.comp_hi = comp_hi;
  next.phase = phases | 32;  // (never 0 = "a whole step": bit 5 marks a ranged launch)
  hipLaunchKernelGGL(aie_step_kernel_log, dim3((unsigned)P.E), dim3(2 * AIE_NT), env->lds, static_cast<hipStream_t>(stream),
                     env->d_params, env->arena, d_actions_a, d_actions_p, next);
  if ((phases & AIE_STEP_TAIL) && P.auto_reset)  // as behind aie_step: the replicas this step finished restart right behind it
    aie_launch_gtb_reset(env, env->arena + P.a_done, 1, stream);
  AIE_HIP_CHECK(env, hipGetLastError());
  return AIE_OK;
}

int aie_step_sample_next(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, uint64_t seed,
                         int64_t global_env_offset, int32_t* d_next_a, int32_t* d_next_p, void* stream) {
  if (!env) return AIE_E_INVALID;
  if ((d_next_a && d_next_a == d_actions_a) || (d_next_p && d_next_p == d_actions_p)) {
    snprintf(env->err, sizeof(env->err), "aie_step_sample_next: the next-action buffers must differ from the current ones");
    return AIE_E_INVALID;
  }
  NextActions next{};
  next.a = d_next_a;
  next.p = d_next_p;
  next.seed = seed;
  next.env_offset = global_env_offset;
  return aie_step_impl(env, d_actions_a, d_actions_p, stream, next);
}

int aie_step_sample_next_masked(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, uint64_t seed,
                                int64_t global_env_offset, int32_t* d_next_a, int32_t* d_next_p, void* stream) {
  if (!env) return AIE_E_INVALID;
  if (env->P.c.scenario != AIE_SCN_COVID) {
    snprintf(env->err, sizeof(env->err), "aie_step_sample_next_masked: COVID scenario only (elsewhere: aie_step + aie_sample_masked_actions)");
    return AIE_E_UNSUPPORTED;
  }
  if ((d_next_a && d_next_a == d_actions_a) || (d_next_p && d_next_p == d_actions_p)) {
    snprintf(env->err, sizeof(env->err), "aie_step_sample_next_masked: the next-action buffers must differ from the current ones");
    return AIE_E_INVALID;
  }
  NextActions next{};
  next.a = d_next_a;
  next.p = d_next_p;
  next.seed = seed;
  next.env_offset = global_env_offset;
  next.masked = 1;
  return aie_step_impl(env, d_actions_a, d_actions_p, stream, next);
}

static int aie_step_impl(aie_env* env, const int32_t* d_actions_a, const int32_t* d_actions_p, void* stream,
                         const NextActions& next_in) {
  if (!env) return AIE_E_INVALID;
  NextActions next = next_in;
  next.E = (int32_t)env->P.E;
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  if (env->P.saez_stride)  // tax_model "saez": the period-start formula runs ahead of the step (aie_kernels_saez.hip)
    hipLaunchKernelGGL(aie_saez_kernel, dim3((unsigned)env->P.E), dim3(AIE_NT), 0, static_cast<hipStream_t>(stream),
                       env->d_params, env->arena);
  if (env->P.c.scenario == AIE_SCN_GTB && env->P.ev_replicas > 0 && env->log_active) {
    // dense-log replicas whose episode is being logged: they -- and only they -- take the full-featured kernel, which
    // records the AIE_EV_* rows; every other replica runs the environment's fast kernel below, in the same stream
    NextActions lg = next;
    lg.e_lo = 0;
    lg.e_hi = env->P.ev_replicas;
    // The grid covers the workgroups that map to the logged replicas [0, L) and as little else as possible (the others
    // leave at once): replica_of_block is the identity when E is not a multiple of 8, else replica e < E / 8 is workgroup 8 e.
    const int64_t L = env->P.ev_replicas, E8 = env->P.E >> 3;
    const unsigned log_grid = (env->P.E & 7) ? (unsigned)L : (L <= E8 ? (unsigned)(8 * (L - 1) + 1) : (unsigned)env->P.E);
    hipLaunchKernelGGL(aie_step_kernel_log, dim3(log_grid), dim3(2 * AIE_NT), env->lds,
                       static_cast<hipStream_t>(stream), env->d_params, env->arena, d_actions_a, d_actions_p, lg);
    if (env->P.ev_replicas >= env->P.E) goto stepped;
    next.e_lo = env->P.ev_replicas;
    next.e_hi = env->P.E;
  }
  if (env->P.c.scenario == AIE_SCN_COVID) {
    const dim3 g((unsigned)env->P.E), b(AIE_NT);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 gw((unsigned)((env->P.E + AIE_CV_WIN_WAVES - 1) / AIE_CV_WIN_WAVES)), bw(AIE_CV_WIN_WAVES * AIE_NT);
    const size_t lw = aie_covid_win_lds_bytes(env->P, env->cv_taps_f32 ? 4 : 8);
    // window sums: the step, then -- a launch of its own -- the upkeep of the change-event lists and the next step's sums
#define AIE_CV_LAUNCH(FN) \
  case FN: if (env->P.c.covid.filter_recurrence) hipLaunchKernelGGL((aie_covid_step_kernel<FN, true>), g, b, 0, st, env->d_params, env->arena, d_actions_a, d_actions_p, next); \
           else { \
             hipLaunchKernelGGL((aie_covid_step_kernel<FN, false>), g, b, 0, st, env->d_params, env->arena, d_actions_a, d_actions_p, next); \
             if (env->cv_taps_f32) hipLaunchKernelGGL((aie_covid_window_kernel<FN, float>), gw, bw, lw, st, env->d_params, env->arena); \
             else hipLaunchKernelGGL((aie_covid_window_kernel<FN, double>), gw, bw, lw, st, env->d_params, env->arena); \
           } \
           break
    switch (env->P.cv_F) {
      AIE_CV_LAUNCH(1); AIE_CV_LAUNCH(2); AIE_CV_LAUNCH(3); AIE_CV_LAUNCH(4);
      AIE_CV_LAUNCH(5); AIE_CV_LAUNCH(6); AIE_CV_LAUNCH(7); AIE_CV_LAUNCH(8);
      default: return AIE_E_UNSUPPORTED;
    }
#undef AIE_CV_LAUNCH
  } else if (env->P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY && env->spec == AIE_KERNEL_INSTANCE_JIT) {
    const aie_params* dp = env->d_params;
    uint8_t* ar = env->arena;
    NextActions nx = next;
    void* args[] = {&dp, &ar, &d_actions_a, &d_actions_p, &nx};
    AIE_HIP_CHECK(env, hipModuleLaunchKernel(env->jit_step, (unsigned)env->P.E, 1, 1, OSE_NT, 1, 1, (unsigned)env->lds,
                                             static_cast<hipStream_t>(stream), args, nullptr));
  } else if (env->P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY && env->spec >= 0) {
    const dim3 g((unsigned)env->P.E), b(OSE_NT);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define AIE_SPEC_LAUNCH_OSE(K) \
  case K: hipLaunchKernelGGL(aie_ose_step_kernel_spec<K>, g, b, env->lds, st, env->d_params, env->arena, d_actions_a, d_actions_p, next); break;
    switch (env->spec) {
      AIE_SPEC_LIST_OSE(AIE_SPEC_LAUNCH_OSE)
      default: return AIE_E_INVALID;
    }
#undef AIE_SPEC_LAUNCH_OSE
  } else if (env->P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY)
    hipLaunchKernelGGL(aie_ose_step_kernel, dim3((unsigned)env->P.E), dim3(OSE_NT), env->lds,
                       static_cast<hipStream_t>(stream), env->d_params, env->arena, d_actions_a, d_actions_p, next);
#ifdef AIE_DEV
  else if (env->spec >= 0 && !(env->P.dev_skip_mask & (1 << 20)) && (env->P.dev_trace != nullptr || env->P.dev_skip_mask != 0)) {
    const dim3 g((unsigned)env->P.E), b(2 * AIE_NT);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define AIE_SPEC_LAUNCH_TR(K) \
  case K: hipLaunchKernelGGL(aie_step_kernel_spec_trace<K>, g, b, env->lds, st, env->d_params, env->arena, d_actions_a, d_actions_p, next); break;
    switch (env->spec) {
      AIE_SPEC_LIST_GTB(AIE_SPEC_LAUNCH_TR)
      default: return AIE_E_INVALID;
    }
#undef AIE_SPEC_LAUNCH_TR
  }
#endif
  else if (env->P.saez_stride || env->P.M > AIE_NT || env->P.regen_general || env->P.dev_skip_mask != 0 ||
           env->P.dev_trace != nullptr)
    hipLaunchKernelGGL(aie_step_kernel_log, dim3((unsigned)env->P.E), dim3(2 * AIE_NT), env->lds,
                       static_cast<hipStream_t>(stream), env->d_params, env->arena, d_actions_a, d_actions_p, next);
  else if (env->spec == AIE_KERNEL_INSTANCE_JIT) {
    const aie_params* dp = env->d_params;
    uint8_t* ar = env->arena;
    NextActions nx = next;
    void* args[] = {&dp, &ar, &d_actions_a, &d_actions_p, &nx};
    AIE_HIP_CHECK(env, hipModuleLaunchKernel(env->jit_step, (unsigned)env->P.E, 1, 1, 2 * AIE_NT, 1, 1, (unsigned)env->lds,
                                             static_cast<hipStream_t>(stream), args, nullptr));
  } else if (env->spec >= 0) {
    const dim3 g((unsigned)env->P.E), b(2 * AIE_NT);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define AIE_SPEC_LAUNCH(K) \
  case K: hipLaunchKernelGGL(aie_step_kernel_spec<K>, g, b, env->lds, st, env->d_params, env->arena, d_actions_a, d_actions_p, next); break;
    switch (env->spec) {
      AIE_SPEC_LIST_GTB(AIE_SPEC_LAUNCH)
      default: return AIE_E_INVALID;
    }
#undef AIE_SPEC_LAUNCH
  } else if (aie_workgroups_per_cu(env->lds) <= 12)
    hipLaunchKernelGGL(aie_step_kernel_r6, dim3((unsigned)env->P.E), dim3(2 * AIE_NT), env->lds,
                       static_cast<hipStream_t>(stream), env->d_params, env->arena, d_actions_a, d_actions_p, next);
  else
    hipLaunchKernelGGL(aie_step_kernel, dim3((unsigned)env->P.E), dim3(2 * AIE_NT), env->lds,
                       static_cast<hipStream_t>(stream), env->d_params, env->arena, d_actions_a, d_actions_p, next);
stepped:
  if (env->P.auto_reset && env->P.c.scenario != AIE_SCN_ONE_STEP_ECONOMY) {
    // auto-reset: the replicas this step finished restart right behind it on the same stream (mask = the `done`
    // tensor the step just wrote; the reset keeps the terminal rewards / done).  one-step-economy does it inside the
    // step launch itself.
    const uint8_t* done = env->arena + env->P.a_done;
    if (env->P.c.scenario == AIE_SCN_COVID)
      hipLaunchKernelGGL(aie_covid_reset_kernel, dim3((unsigned)env->P.E), dim3(AIE_NT), 0,
                         static_cast<hipStream_t>(stream), env->d_params, env->arena, done, 1);
    else
      aie_launch_gtb_reset(env, done, 1, stream);
  }
  AIE_HIP_CHECK(env, hipGetLastError());
  return AIE_OK;
}

int aie_set_global_saez_buffer(aie_env* env, const double* d_pairs, int64_t n_pairs) {
  if (!env || n_pairs < 0 || (n_pairs > 0 && !d_pairs)) return AIE_E_INVALID;
  if (!env->P.saez_stride) {
    snprintf(env->err, sizeof(env->err), "the global Saez buffer needs tax_model \"saez\"");
    return AIE_E_UNSUPPORTED;
  }
  if (n_pairs > env->P.saez_global_cap) {
    snprintf(env->err, sizeof(env->err), "global Saez buffer of %lld pairs exceeds aie_config.saez_global_capacity = %d",
             (long long)n_pairs, env->P.saez_global_cap);
    return AIE_E_INVALID;
  }
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  AIE_HIP_CHECK(env, hipDeviceSynchronize());
  uint8_t* g = env->arena + env->P.a_saez_global;
  if (n_pairs > 0)
    AIE_HIP_CHECK(env, hipMemcpy(g + 16, d_pairs, (size_t)n_pairs * 16, hipMemcpyDeviceToDevice));
  const int32_t len = (int32_t)n_pairs;
  AIE_HIP_CHECK(env, hipMemcpy(g, &len, 4, hipMemcpyHostToDevice));
  return AIE_OK;
}

int aie_set_dense_log_active(aie_env* env, int on) {
  if (!env) return AIE_E_INVALID;
  env->log_active = on ? 1 : 0;
  return AIE_OK;
}

int aie_set_auto_reset(aie_env* env, int on) {
  if (!env) return AIE_E_INVALID;
  if (on && env->P.c.scenario == AIE_SCN_GTB && !env->P.c.shared_layout && env->P.c.layout_gen == AIE_LAYOUT_FIXED) {
    // per-replica layouts supplied from outside (worlds too large for the device-side generator): the next
    // episode's layout comes from the host
    snprintf(env->err, sizeof(env->err), "auto-reset is not available when the host supplies a new layout per episode");
    return AIE_E_UNSUPPORTED;
  }
  env->P.auto_reset = on ? 1 : 0;
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  AIE_HIP_CHECK(env, hipDeviceSynchronize());
  AIE_HIP_CHECK(env, hipMemcpy(env->d_params, &env->P, sizeof(aie_params), hipMemcpyHostToDevice));
  return AIE_OK;
}

int aie_sample_random_actions(aie_env* env, uint64_t seed, int64_t global_env_offset, int32_t* d_actions_a,
                              int32_t* d_actions_p, void* stream) {
  if (!env) return AIE_E_INVALID;
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  const aie_params& P = env->P;
  const int64_t tot = (int64_t)P.E * (P.n * P.act_a_width + P.act_p_width);
  hipLaunchKernelGGL(aie_sample_actions_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), P, env->arena, seed, global_env_offset, d_actions_a, d_actions_p);
  hipLaunchKernelGGL(aie_sample_advance_kernel, dim3((unsigned)((P.E + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), env->arena, P.a_records, P.rec_bytes, P.o_sample_t, P.E);
  AIE_HIP_CHECK(env, hipGetLastError());
  return AIE_OK;
}

int aie_sample_masked_actions(aie_env* env, uint64_t seed, int64_t global_env_offset, int32_t* d_actions_a,
                              int32_t* d_actions_p, void* stream) {
  if (!env) return AIE_E_INVALID;
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  const aie_params& P = env->P;
  const int64_t tot = (int64_t)P.E * (P.n * P.act_a_width + P.act_p_width);
  hipLaunchKernelGGL(aie_sample_masked_actions_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), P, env->arena, seed, global_env_offset, d_actions_a, d_actions_p);
  hipLaunchKernelGGL(aie_sample_advance_kernel, dim3((unsigned)((P.E + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), env->arena, P.a_records, P.rec_bytes, P.o_sample_t, P.E);
  AIE_HIP_CHECK(env, hipGetLastError());
  return AIE_OK;
}

int aie_sample_policy_actions(aie_env* env, const float* d_logits_a, const float* d_logits_p, uint64_t seed,
                              int64_t global_env_offset, int32_t* d_actions_a, int32_t* d_actions_p, void* stream) {
  if (!env) return AIE_E_INVALID;
  if ((d_actions_a && !d_logits_a) || (d_actions_p && !d_logits_p)) {
    snprintf(env->err, sizeof(env->err), "aie_sample_policy_actions: an action buffer without its logits");
    return AIE_E_INVALID;
  }
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  // waves per replica (1, 2 or 4 of a workgroup's four); AIE_SAMPLER_WAVES_LOG2 in the environment is a development knob
  static const int wpr_log2 = [] {
    const char* v = getenv("AIE_SAMPLER_WAVES_LOG2");
    const int k = v ? atoi(v) : 1;
    int r = k < 0 ? 0 : (k > 2 ? 2 : k);
#ifdef AIE_DEV
    if (const char* sk = getenv("AIE_SAMPLER_DEV_SKIP")) r |= atoi(sk) << 8;
#endif
    return r;
  }();
  const int rpb = 4 >> (wpr_log2 & 255);
  const aie_sampler_args S = aie_sampler_args_of(&env->P, env->d_params);
  using sampler_fn = void (*)(const aie_sampler_args, uint8_t*, const float*, const float*, uint64_t, int64_t, int32_t*, int32_t*, int);
  sampler_fn fn = aie_sample_policy_actions_kernel;  // rows of any shape
  if (!S.ragged && S.agents.len <= 64 && S.planner.len <= 64) {  // every row one aligned lane segment: the fast instances
    static const sampler_fn fast[3][3] = {
        {aie_sample_policy_fast_kernel<4, 4>, aie_sample_policy_fast_kernel<4, 5>, aie_sample_policy_fast_kernel<4, 6>},
        {aie_sample_policy_fast_kernel<5, 4>, aie_sample_policy_fast_kernel<5, 5>, aie_sample_policy_fast_kernel<5, 6>},
        {aie_sample_policy_fast_kernel<6, 4>, aie_sample_policy_fast_kernel<6, 5>, aie_sample_policy_fast_kernel<6, 6>}};
    fn = fast[S.agents.lsh - 4][S.planner.lsh - 4];
  }
  hipLaunchKernelGGL(fn, dim3((unsigned)((env->P.E + rpb - 1) / rpb)), dim3(256), 0, static_cast<hipStream_t>(stream), S,
                     env->arena, d_logits_a, d_logits_p, seed, global_env_offset, d_actions_a, d_actions_p, wpr_log2);
  AIE_HIP_CHECK(env, hipGetLastError());
  return AIE_OK;
}

// Which step kernel runs this environment: >= 0 = compile-time instance (index into aie_spec_generated.h), -1 = generic.
int aie_step_kernel_instance(aie_env* env) { return env ? env->spec : -2; }

// ---- run-time specialisation (aie_jit.h): requested in the background by aie_create, adopted at the next aie_reset ----
static bool aie_jit_eligible(const aie_env* env) {
  const aie_params& P = env->P;
  const bool ose = P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY;
  return !(P.c.scenario == AIE_SCN_COVID || P.saez_stride || (!ose && (P.M > AIE_NT || P.regen_general)));
}
// starts (or joins) the background job that compiles / fetches the code object of this environment's family
static int aie_jit_request(aie_env* env) {
  if (env->jit_job && env->jit_job->pid == (long)getpid()) return AIE_OK;
  env->jit_job.reset();  // (inherited across fork(): the parent's compiler thread does not exist here)
  hipDeviceProp_t prop;
  AIE_HIP_CHECK(env, hipGetDeviceProperties(&prop, env->device));
  std::string arch = prop.gcnArchName;  // "gfx950:sramecc+:xnack-" -> "gfx950"
  arch = arch.substr(0, arch.find(':'));
  std::vector<aie_params> norm(1, env->P);  // (not static: environments specialise from different threads)
  aie_spec_normalize(&norm[0]);
  const bool ose = env->P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY;
  const int wg = aie_workgroups_per_cu(env->lds);
  // gather-trade-build: two waves per workgroup on four SIMDs; one-step-economy: one wave per workgroup
  const int waves = ose ? (wg >= 16 ? 4 : wg >= 12 ? 3 : 2) : ((2 * wg + 3) / 4 < 8 ? (2 * wg + 3) / 4 : 8);
  env->jit_job = aie_jit::start_job(&norm[0], sizeof(aie_params), waves, arch.c_str(), ose);
  return AIE_OK;
}
static bool aie_jit_load(aie_env* env, const std::string& code, bool ose) {
  hipModule_t mod = nullptr;
  hipFunction_t fs = nullptr, fr = nullptr;
  if (hipModuleLoadData(&mod, code.data()) != hipSuccess ||
      hipModuleGetFunction(&fs, mod, ose ? "aie_jit_ose_step" : "aie_jit_step") != hipSuccess ||
      (!ose && hipModuleGetFunction(&fr, mod, "aie_jit_reset") != hipSuccess)) {
    if (mod) (void)hipModuleUnload(mod);
    (void)hipGetLastError();
    return false;
  }
  env->jit_mod = mod;
  env->jit_step = fs;
  env->jit_reset = fr;
  return true;
}
// If the job has finished: load its code object and, unless the caller pinned the generic kernel, switch to it.  Called
// at the top of aie_reset (an episode boundary; the kernels are bit-identical, so the switch is invisible; not while the
// stream is being captured: loading a module synchronises) and by aie_specialize (wait = true).  aie_step never does.  Returns AIE_OK when the environment now has its specialised kernels.
static int aie_jit_adopt(aie_env* env, bool wait) {
  if (!env->jit_job) return env->spec_match == AIE_KERNEL_INSTANCE_JIT ? AIE_OK : AIE_E_UNSUPPORTED;
  std::shared_ptr<aie_jit::Job> job = env->jit_job;
  if (wait) aie_jit::run_job_now_if_queued(job);  // (still in the queue behind other environments' jobs: compile it here)
  int st = job->state.load(std::memory_order_acquire);
  while (wait && st == 0) {
    usleep(2000);
    st = job->state.load(std::memory_order_acquire);
  }
  if (st == 0) return AIE_E_UNSUPPORTED;  // still compiling: the generic kernel carries on
  env->jit_job.reset();
  if (st < 0) {
    snprintf(env->err, sizeof(env->err), "aie_specialize: %s", job->err.c_str());
    return AIE_E_UNSUPPORTED;
  }
  (void)hipSetDevice(env->device);
  bool ok = aie_jit_load(env, job->code, job->ose);
  if (!ok) {
    // a cached code object that does not load (another toolchain, a damaged file): drop it and compile once more
    std::string code, err;
    if (aie_jit::code_object(job->image.data(), job->image.size(), job->waves, job->arch.c_str(), job->ose, code, err, nullptr,
                             /*ignore_cache=*/true))
      ok = aie_jit_load(env, code, job->ose);
  }
  if (!ok) {
    snprintf(env->err, sizeof(env->err), "aie_specialize: the compiled code object could not be loaded");
    return AIE_E_UNSUPPORTED;
  }
  env->spec_match = AIE_KERNEL_INSTANCE_JIT;
  if (!env->pinned_generic) env->spec = AIE_KERNEL_INSTANCE_JIT;
  return AIE_OK;
}
static inline void aie_jit_poll(aie_env* env) {
  if (env->jit_job && env->jit_job->pid != (long)getpid()) env->jit_job.reset();
  if (env->jit_job && env->jit_job->state.load(std::memory_order_acquire) != 0) (void)aie_jit_adopt(env, false);
}

int aie_select_step_kernel(aie_env* env, int which) {
  if (!env) return AIE_E_INVALID;
  if (which != AIE_KERNEL_AUTO && which != AIE_KERNEL_GENERIC) {
    snprintf(env->err, sizeof(env->err), "aie_select_step_kernel: unknown kernel %d", which);
    return AIE_E_INVALID;
  }
  env->pinned_generic = which == AIE_KERNEL_GENERIC;
  env->spec = which == AIE_KERNEL_GENERIC ? -1 : env->spec_match;
  return AIE_OK;
}

int aie_specialize(aie_env* env) {
  if (!env) return AIE_E_INVALID;
  // already on a specialised kernel (a compile-time instance, or an earlier call); AIE_JIT_FORCE=1 compiles anyway
  // (A/B of a run-time against a compile-time instance, tools/jit_timing.py)
  if (env->spec_match >= 0 && !(getenv("AIE_JIT_FORCE") && env->spec_match != AIE_KERNEL_INSTANCE_JIT)) {
    if (!env->pinned_generic) env->spec = env->spec_match;
    return AIE_OK;
  }
  if (!aie_jit_eligible(env)) {
    snprintf(env->err, sizeof(env->err), "aie_specialize: this configuration runs the full-featured step kernel (tax_model "
             "\"saez\", order books beyond a wavefront, general regeneration) or is the COVID scenario");
    return AIE_E_UNSUPPORTED;
  }
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  const int rc = aie_jit_request(env);
  if (rc != AIE_OK) return rc;
  return aie_jit_adopt(env, /*wait=*/true);
}

#ifdef AIE_DEV
// ---- development hooks: compiled only into libaie_hip_dev.so (-DAIE_DEV; tools/ and the tests that need to reach
// inside a launch load that build), never into the shipping library; not part of include/aie.h ----

// Evaluates the device build of aie_glibc_math.h on arrays, so that a GPU test can compare it with the host's libm
// bit for bit.  fn 0: out = pow(x, y); fn 1: out = exp(x); fn 2: out = log(x).
__global__ void aie_test_glibc_math_kernel(int fn, const double* __restrict__ x, const double* __restrict__ y,
                                           double* __restrict__ out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = fn == 0 ? aie_pow_glibc(x[i], y[i]) : fn == 1 ? aie_exp_glibc(x[i]) : aie_log_glibc(x[i]);
}
AIE_DEV_API int aie_test_glibc_math(int fn, const void* d_x, const void* d_y, void* d_out, int64_t n, void* stream) {
  if (n <= 0 || !d_x || !d_out || (fn == 0 && !d_y) || fn < 0 || fn > 2) return AIE_E_INVALID;
  hipLaunchKernelGGL(aie_test_glibc_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), fn, static_cast<const double*>(d_x),
                     static_cast<const double*>(d_y), static_cast<double*>(d_out), n);
  return hipGetLastError() == hipSuccess ? AIE_OK : AIE_E_HIP;
}

// dynamic LDS bytes of a step workgroup (out[0]) and its parts: record image, location map, f64 scratch, staging
// area; out[5]: workgroups per CU that size allows.
AIE_DEV_API int aie_dev_lds_bytes(aie_env* env, int64_t* out) {
  if (!env || !out) return AIE_E_INVALID;
  out[0] = (int64_t)env->lds;
  out[5] = aie_workgroups_per_cu(env->lds);
  out[1] = env->P.o_mt;
  out[2] = env->P.HW;
  out[3] = env->P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY || env->P.c.scenario == AIE_SCN_COVID ? 0 : (int64_t)aie::fscr_doubles(env->P) * 8;
  out[4] = env->P.c.scenario == AIE_SCN_ONE_STEP_ECONOMY || env->P.c.scenario == AIE_SCN_COVID ? 0 : (int64_t)aie::stage_bytes(env->P);
  return AIE_OK;
}

static int aie_dev_push_params(aie_env* env) {
  AIE_HIP_CHECK(env, hipSetDevice(env->device));
  AIE_HIP_CHECK(env, hipDeviceSynchronize());
  AIE_HIP_CHECK(env, hipMemcpy(env->d_params, &env->P, sizeof(aie_params), hipMemcpyHostToDevice));
  return AIE_OK;
}

// device buffer of 12*E uint64 clock stamps per launch
AIE_DEV_API int aie_dev_set_trace(aie_env* env, void* d_buf) {
  if (!env) return AIE_E_INVALID;
  env->P.dev_trace = static_cast<uint64_t*>(d_buf);
  return aie_dev_push_params(env);
}

// extra dynamic LDS per workgroup (lowers residency)
AIE_DEV_API int aie_dev_set_lds_pad(aie_env* env, int bytes) {
  if (!env || bytes < 0) return AIE_E_INVALID;
  env->lds += (size_t)bytes;
  return AIE_OK;
}

// a smaller draw window for the components (the environment then runs aie_step_kernel_log), so that tests reach the
// refill path on every step
AIE_DEV_API int aie_dev_set_draw_window(aie_env* env, int words) {
  if (!env || words < 0) return AIE_E_INVALID;
  env->P.dev_draw_window = words;
  env->P.dev_skip_mask = words ? (env->P.dev_skip_mask | (1 << 20)) : (env->P.dev_skip_mask & ~(1 << 20));
  return aie_dev_push_params(env);
}

// phases of the step kernel to skip
AIE_DEV_API int aie_dev_set_skip_mask(aie_env* env, int mask) {
  if (!env) return AIE_E_INVALID;
  env->P.dev_skip_mask = mask;
  return aie_dev_push_params(env);
}
#endif  // AIE_DEV

}  // extern "C"
