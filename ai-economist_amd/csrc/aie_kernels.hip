// aie_kernels.hip -- hand-written CDNA4 (gfx950) kernels for the batched Foundation
// env.step() / env.reset() of the gather-trade-build family.
//
// Execution model: ONE WAVEFRONT (64 lanes) PER ENV REPLICA, one replica per workgroup.
//   * the replica's state record (layout in aie_layout.h) is streamed HBM -> LDS with
//     16-byte lane loads; its MT19937 key (624 words) goes HBM -> REGISTERS instead
//     (10 VGPRs: lane l of row j holds word 64*j + l),
//   * agent i's scalars (location, inventories, coin, labor, decoded action) live in the
//     registers of lane i; "agent j acts next" is a v_readlane broadcast, never an LDS
//     round trip; random agent orders are lane-distributed permutations,
//   * the order book is matched with wave ballots: "best bid of a not-yet-flagged buyer"
//     and "best ask of another agent" are find-first-set over 64 book slots at a time,
//     insertions / removals / expiry compaction are one-instruction lane shifts,
//   * the MT19937 twist is 10 register rows x (3 ds_bpermute + ~10 VALU) with no memory
//     traffic and no barriers; np.random.rand(H, W) is consumed straight from the rows,
//   * observations are written to their dense [E, ...] tensors with lane-contiguous
//     (coalesced) stores; small vectors are staged in LDS and streamed out.
// The path is integer / branchy and HBM-bound on the observation writes: no MFMA.
//
// Each __device__ function cites the reference function it implements (paths relative
// to the reference tree, F/ = ai_economist/foundation/).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aie_layout.h"
#include "aie_glibc_math.h"

// Floating-point contraction is OFF for every kernel of this library: NumPy (and the C oracle) round a product before
// they add it, and state such as coin, labor and utilities has to come out bit for bit -- the sign of a ~1e-16 reward
// decides an integer state field (layout_from_file.py:553-557).  Fused multiply-adds appear only where libm itself
// fuses them (aie_glibc_math.h, written out with __builtin_fma).
#pragma clang fp contract(off)

// constant images of the parameter block: the build's compile-time instances (aie_spec_generated.h), or -- in a
// run-time specialisation (AIE_JIT, aie_specialize() in aie_capi.hip) -- the one image of the caller's environment
#ifdef AIE_JIT
#include "aie_jit_image.h"
#else
#include "aie_spec_generated.h"
#endif
template <int SPEC>
__device__ __forceinline__ const aie_params& aie_spec_params(const aie_params* run_time) {
  if constexpr (SPEC < 0) return *run_time;
  else return *reinterpret_cast<const aie_params*>(aie_spec_image<SPEC>::bytes);
}

#define AIE_NT 64  // threads per replica (one wavefront)
#define AIE_DIRTY_CAP 64  // map cells one step may change before the incremental map observations give up (= one lane each)
// (AIE_SRC_CAP, aie_layout.h: source-block doubles handled by the gather regen, else row regen)

namespace aie {

// NOTE: every __device__ function below is __forceinline__: a non-inlined call that takes
// `const aie_params&` forces the compiler to copy the whole by-value kernel argument
// (2.7 KB) to scratch memory on every launch -- measured 5x slower.

struct Ctx {
  const aie_params& P;  // everything a kernel needs; in the compile-time instances of the step kernel (aie_spec_generated.h)
                        // a CONSTANT image of the block, so that dims, offsets and component lists fold into the code
  const aie_params& R;  // the run-time block in device memory: what depends on the batch (E, arena offsets a_*)
  uint8_t* rec;      // LDS copy of the record (everything before the MT19937 key)
  int32_t* act_p;    // LDS [AIE_MAX_BRACKETS] decoded planner actions
  uint8_t* locmap;   // LDS [HW] 0 = empty, i+1 = agent i
  double* fscr;      // LDS f64 scratch
  float* stage;      // LDS staging of the small observation vectors (also: MT word dump during regen)
  uint16_t* srcl;    // (unused since round 6: the source list is a record field read straight into registers, src_list_from_record)
  int32_t* srcn;     // LDS [4] scratch words ([2]: the dense log's event rows of this step)
  int32_t* mflags;   // LDS [n] per-agent mask bits
  int32_t* dirty;    // LDS [4 + AIE_DIRTY_CAP/2]: count, moved-agent mask (2 words), pad, uint16 cell list
  uint8_t* snap;     // LDS [2][HW]: pre-step max(map, source block) per resource (P.regen_general only), else nullptr
  uint8_t* met;      // (one-step-economy: this replica's episode accumulators; everybody else: C_MET)
  uint8_t* met_arena;  // GLOBAL: the arena where the replica has episode accumulators (aie_layout.h: a_metrics), else nullptr
  int32_t* ev;       // GLOBAL: this replica's dense-log event rows (a_events), or nullptr (not logged)
  bool saez;         // tax_model == "saez" (compile-time false in the common step kernel, like ev == nullptr)
  bool full;         // compile-time: false in the common step kernel (aie_step_kernel), which leaves out what
                     // only some environments need -- dense-log rows, Saez hooks, order books larger than a
                     // wavefront, development hooks -- to keep its code small (the kernel does not fit the
                     // instruction cache, and rarely taken paths pay for every line they add)
  int tid;
  int e;
  const double* rtab;    // the discretised tax rates (aie_config.tax_disc_rates) and ...
  const uint32_t* mtab;  // ... the action-mask element tests (aie_params.mask_test): LDS copies in the step kernel where
                         // the LDS budget allows (const_tables_in_lds), else the parameter block's own arrays
  uint32_t* mtwin;   // LDS [640]: the generator's current window for the regeneration's sparse reads, or nullptr (mt_window_in_lds)
  int skipm;         // development (-DAIE_DEV builds: aie_dev_set_skip_mask): phases switched off; constant 0 otherwise
};

#define R_F64(c, off) (reinterpret_cast<double*>((c).rec + (c).P.off))
#define R_I32(c, off) (reinterpret_cast<int32_t*>((c).rec + (c).P.off))
#define R_U8(c, off) (reinterpret_cast<uint8_t*>((c).rec + (c).P.off))
#define R_CELLS(c) (reinterpret_cast<uint32_t*>((c).rec + (c).P.o_cells))

// f64 scratch slots: net price history [2][P], then four per-agent vectors
__device__ __forceinline__ double* scr_net_ph(const Ctx& c) { return c.fscr; }
__device__ __forceinline__ double* scr_sorted_inc(const Ctx& c) { return c.fscr + 2 * c.P.P; }
__device__ __forceinline__ double* scr_cmr(const Ctx& c) { return c.fscr + 2 * c.P.P + c.P.n; }
__device__ __forceinline__ double* scr_coin(const Ctx& c) { return c.fscr + 2 * c.P.P + 2 * c.P.n; }
__device__ __forceinline__ double* scr_part(const Ctx& c) { return c.fscr + 2 * c.P.P + 3 * c.P.n; }  // [n+1]
// sort buffer of the planner's gini (n >= 30): its own slot, because the rewards run on the
// second wave of a replica while the first one uses scr_sorted_inc for the tax observations
__device__ __forceinline__ double* scr_gini_sort(const Ctx& c) { return c.fscr + 2 * c.P.P + 4 * c.P.n + 2; }
__host__ __device__ inline int fscr_doubles(const aie_params& P) { return 2 * P.P + 5 * P.n + 2; }

// The LDS image of the record stops where the MT19937 key starts (it lives in VGPRs).  The counter stream's state
// (AIE_RNG_FAST: key32, block number, salt, 0 -- 16 bytes) is part of the image.
__host__ __device__ inline bool rng_fast(const aie_params& P) { return P.c.rng_mode == AIE_RNG_FAST; }
// One list of the regeneration's source doubles for the whole batch (aie_layout.h: aie__shared_src_list, a_src_list)
// instead of a scan of the cells' flag bytes in every step -- used in the counter-stream mode only: A/B on one box
// (tools/ab_variants.sh, round 5) C2f 22.35 -> 22.0 us, but C2 24.7 -> 25.4 us and C3 40.5 -> 40.7 us with MT19937, whose
// second wave already holds the generator's ten rows while the list's registers wait for the regeneration.
__host__ __device__ inline bool shared_src_list(const aie_params& P) {
#ifdef AIE_NO_SHARED_SRC_LIST  // (A/B builds)
  return false;
#endif
  return rng_fast(P) && P.c.scenario == AIE_SCN_GTB && P.c.shared_layout && P.c.layout_gen == AIE_LAYOUT_FIXED;
}
__host__ __device__ inline int rec_lds_bytes(const aie_params& P) { return P.o_mt + (rng_fast(P) ? 16 : 0); }

// staging area: the components' draw window (the next tempered MT19937 words, see MTL) and the small
// observation vectors the flat-vector writer stages
__host__ __device__ inline int pad4(int x) { return (x + 3) & ~3; }
__host__ __device__ inline int stage_window_words(const aie_params& P) {
  // words the components of one step draw at most in the common case: two agent-order permutations (n - 1 masked
  // rejection draws each, < 2 words per draw on average) and a pickup draw per agent and resource; a step that needs
  // more refills the window from the generator state in HBM (rng_refill)
  int w = (4 * P.n + 4 * P.n + 63) / 64 * 64;
  if (w < 128) w = 128;
  return w > 576 ? 576 : w;
}
__host__ __device__ inline size_t stage_bytes(const aie_params& P) {
  // [0, window): tempered words pos ... of the generator's current window (filled by the wave that holds the state);
  // behind it: the planner's flat vector + per-agent fragments (write_flat_observations)
  const size_t b = (size_t)(stage_window_words(P) + pad4(P.n * P.FPA) + pad4(P.FP)) * 4;
  return (b + 15) / 16 * 16;
}

__host__ __device__ inline size_t lds_bytes_base(const aie_params& P) {
  size_t b = (size_t)rec_lds_bytes(P);
  b += AIE_MAX_BRACKETS * 4;
  b = (b + 15) / 16 * 16;
  b += ((size_t)P.HW + 15) / 16 * 16;
  b += (size_t)fscr_doubles(P) * 8;
  b += stage_bytes(P);
  b += 16 + (size_t)pad4(P.n) * 4;  // (the source list is part of the record image since round 6)
  b += 16 + AIE_DIRTY_CAP * 2;
  b = (b + 15) / 16 * 16;
  return b;
}
// The regeneration reads ~80 of a step's 2 500 generator words, scattered over five consecutive windows.  Where a
// workgroup's LDS footprint leaves room for it WITHOUT costing residency (16 workgroups per CU need <= 10 240 B each),
// the second wave publishes each window's ten rows to LDS (ten stores) and every lane fetches its two adjacent words
// with one ds_read2 -- instead of selecting them out of the row registers with ten lane permutes and ten selects per
// word (scenario_step_regen).  Larger records (ten agents and more) keep the register gather.
#define AIE_MT_WINDOW_LDS_BYTES 2560  // 10 rows x 64 lanes x 4 B (row 9's upper 16 lanes are padding)
// Two small tables of the parameter block are indexed with run-time values on the replica's critical path: the
// discretised tax rates (by the planner's latched rate indices: every marginal-rate / tax-due evaluation and the
// curr_rates observation) and the per-element tests of the flattened action mask.  In a compile-time instance the
// block is a constant image in device memory, so each such lookup is a global (or scalar) memory round trip behind the
// LDS read of its index.  The step kernel's second wave copies both tables to LDS while the record streams in --
// where that does not cost residency, like the generator's window below (which takes what room is left).
__host__ __device__ inline size_t const_table_bytes(const aie_params& P) {
  const size_t r = (P.has_tax && P.c.tax_model == AIE_TAX_MODEL_WRAPPER) ? (size_t)P.c.tax_n_disc_rates * 8 : 0;
  return (r + (size_t)P.MA * 4 + 15) / 16 * 16;
}
__host__ __device__ inline bool const_tables_in_lds(const aie_params& P) {
#ifdef AIE_NO_CONST_TABLES_LDS  // (A/B builds)
  return false;
#else
  return lds_bytes_base(P) + const_table_bytes(P) <= 10240;
#endif
}
__host__ __device__ inline bool mt_window_in_lds(const aie_params& P) {
#ifdef AIE_NO_MT_WINDOW_LDS  // (A/B builds)
  return false;
#else
  return !P.regen_general && !rng_fast(P) &&  // (the counter stream addresses its words directly)
         lds_bytes_base(P) + (const_tables_in_lds(P) ? const_table_bytes(P) : 0) + AIE_MT_WINDOW_LDS_BYTES <= 10240;
#endif
}
__host__ __device__ inline size_t lds_bytes(const aie_params& P) {
  size_t b = lds_bytes_base(P);
  if (const_tables_in_lds(P)) b += const_table_bytes(P);
  if (mt_window_in_lds(P)) b += AIE_MT_WINDOW_LDS_BYTES;
  if (P.regen_general) b += 2 * (((size_t)P.HW + 15) / 16 * 16);
  return (b + 15) / 16 * 16;
}

// the accumulators' address where it is used (rarely: a trade, a tax day) instead of in every prologue: a_metrics was a
// scalar load ahead of the step kernel's first record load (0.2 us of the launch, round 6)
#define C_MET(c) ((c).met_arena ? (c).met_arena + (c).R.a_metrics + (int64_t)(c).e * (c).P.met_bytes : nullptr)
__device__ __forceinline__ Ctx make_ctx(const aie_params& P, const aie_params& R, uint8_t* lds, int e, int tid,
                                        uint8_t* arena = nullptr, bool with_events = true, int skipm = 0,
                                        bool lds_tables = false) {
  uint8_t* q = lds + rec_lds_bytes(P);
  int32_t* act_p = reinterpret_cast<int32_t*>(q);
  q += AIE_MAX_BRACKETS * 4;
  q = lds + ((q - lds) + 15) / 16 * 16;
  uint8_t* locmap = q;
  q += (P.HW + 15) / 16 * 16;
  double* fscr = reinterpret_cast<double*>(q);
  q += fscr_doubles(P) * 8;
  float* stage = reinterpret_cast<float*>(q);
  q += stage_bytes(P);
  uint16_t* srcl = nullptr;
  int32_t* srcn = reinterpret_cast<int32_t*>(q);
  q += 16;
  int32_t* mflags = reinterpret_cast<int32_t*>(q);
  q += pad4(P.n) * 4;
  int32_t* dirty = reinterpret_cast<int32_t*>(q);
  q += 16 + AIE_DIRTY_CAP * 2;
  q = lds + ((q - lds) + 15) / 16 * 16;
  // (the kernel that asks for the LDS tables fills them: step_body; everyone else reads the parameter block)
  const double* rtab = R.c.tax_disc_rates;
  const uint32_t* mtab = P.mask_test;
  if (const_tables_in_lds(P)) {
    if (lds_tables) {
      const bool rates = P.has_tax && P.c.tax_model == AIE_TAX_MODEL_WRAPPER;
      rtab = reinterpret_cast<const double*>(q);
      mtab = reinterpret_cast<const uint32_t*>(q + (rates ? P.c.tax_n_disc_rates * 8 : 0));
    }
    q += const_table_bytes(P);
  }
  uint32_t* mtwin = mt_window_in_lds(P) ? reinterpret_cast<uint32_t*>(q) : nullptr;
  if (mt_window_in_lds(P)) q += AIE_MT_WINDOW_LDS_BYTES;
  uint8_t* snap = P.regen_general ? q : nullptr;
  uint8_t* met = nullptr;  // (C_MET)
  // with_events == false is a compile-time constant in the common step kernel: every `if (c.ev)` / `if (c.saez)`
  // folds away (environments with dense-log replicas or tax_model "saez" run aie_step_kernel_log)
  int32_t* ev = (with_events && arena && e < R.ev_replicas)  // (the event buffer is a property of the batch, not of an
                    ? reinterpret_cast<int32_t*>(arena + R.a_events + (int64_t)e * R.ev_stride) : nullptr;  // instance's family)
  const bool saez = with_events && P.c.tax_model == AIE_TAX_SAEZ;
  return Ctx{P, R, lds, act_p, locmap, fscr, stage, srcl, srcn, mflags, dirty, snap, met, arena, ev, saez, with_events, tid, e, rtab, mtab, mtwin, skipm};
}

// ------------------------------------------------------------------------------------
// wave helpers
// ------------------------------------------------------------------------------------
// Ordering point inside a section that ONE wavefront executes (every __device__ function
// below is such a section: lane loops stride by AIE_NT = 64).  LDS instructions of a wave are
// issued and executed in order, so a later ds_read of any lane sees an earlier ds_write of any
// other lane of the same wave; what is needed is that the compiler keeps that order.  Block-
// level barriers (__syncthreads) only appear in the kernels, between sections that different
// waves of a replica's workgroup execute.
#define AIE_WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int bcast(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ uint32_t bcast(uint32_t v, int lane) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
__device__ __forceinline__ double bcast(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// Buffer-descriptor stores: base + num_records in 4 SGPRs, a 32-bit VGPR byte offset and a
// scalar byte offset per instruction (gfx9 raw buffer, DATA_FORMAT = 32 in word 3).
typedef __amdgpu_buffer_rsrc_t BufRsrc;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// (base and size are wave-uniform at every call site.  They go through v_readfirstlane -- free for a value that already
// sits in scalar registers -- because a descriptor the instruction selector happened to build in vector registers makes
// EVERY store through it a "waterfall" loop over its distinct values: round 6 met that when the kernels' first address
// computations moved, +900 vector instructions in one instance.)
__device__ __forceinline__ BufRsrc make_rsrc(void* base, uint32_t bytes) {
  const uint64_t b = reinterpret_cast<uint64_t>(base);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>((uint64_t)lo | ((uint64_t)hi << 32)), 0,
                                           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
__device__ __forceinline__ void buf_store_f32(BufRsrc r, float v, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_i16(BufRsrc r, int v, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, voff, soff, 0);
}
__device__ __forceinline__ void buf_store_u32x2(BufRsrc r, uint32_t a, uint32_t b, int voff, int soff) {
  const u32x2 v = {a, b};
  __builtin_amdgcn_raw_buffer_store_b64(v, r, voff, soff, 0);
}
// 16-byte buffer stores never carry a scalar offset REGISTER: measured on MI355X (round 3), `buffer_store_dwordx4
// v[6:9], v25, s[8:11], s7 offen` followed directly by `v_mov_b32 v6, 2` stored the 2 in lanes 12-15 of every 16 --
// the ">64-bit store data, then VALU write of those VGPRs" hazard exists on this part with an SGPR soffset too, while
// the compiler (ROCm 7.2 clang, GCNHazardRecognizer::createsVALUHazard) only spaces the pair out when soffset is an
// immediate.  With the offset folded into the vector offset and soffset = 0 the compiler inserts the wait state.
// (tests/test_gpu_parity.py::test_reset_is_deterministic_across_environments is the check that found it.)
__device__ __forceinline__ void buf_store_f32x4(BufRsrc r, float a, float b, float c, float d, int voff, int soff) {
  const u32x4 v = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
  __builtin_amdgcn_raw_buffer_store_b128(v, r, voff + soff, 0, 0);
}
// 16 bytes with dword alignment: global dwordx4 accesses need no more on gfx950
struct __attribute__((packed, aligned(4))) f32x4_a4 { float x, y, z, w; };

// LDS (16-byte aligned) -> global (dword aligned) copy of `count` floats
__device__ __forceinline__ void stream_out(const float* __restrict__ lds_src, float* __restrict__ dst, int count, int tid) {
  const int n4 = count >> 2;
  for (int q = tid; q < n4; q += AIE_NT) {
    const float4 v = reinterpret_cast<const float4*>(lds_src)[q];
    f32x4_a4 o = {v.x, v.y, v.z, v.w};
    reinterpret_cast<f32x4_a4*>(dst)[q] = o;
  }
  for (int q = 4 * n4 + tid; q < count; q += AIE_NT) dst[q] = lds_src[q];
}

// q / d for a run-time constant d with host-computed magic (aie_layout.h: aie__magic)
__device__ __forceinline__ int udiv(int q, int d, uint32_t magic) {
  return d == 1 ? q : (int)__umulhi((uint32_t)q, magic);
}
// XCD-aware workgroup -> replica mapping.  The dispatcher places workgroup b on XCD b % 8
// (observed, used for speed only): give every XCD a CONTIGUOUS range of replicas so that
// the 128-byte lines shared by neighbouring replicas in the env-major tensors (rewards,
// done, flat vectors whose per-replica size is not a multiple of 128 B) are completed
// inside one L2 instead of being written back as partial lines by two of them.
__device__ __forceinline__ int replica_of_block(int b, int E) {
  if (E & 7) return b;
  return (b & 7) * (E >> 3) + (b >> 3);
}
__device__ __forceinline__ uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// One action slot j of replica e under the benchmark's uniform random policy (slots: the
// agents' sub-actions in order, then the planner's).
__device__ __forceinline__ void sample_action_slot(const aie_params& P, uint64_t seed, int64_t env_offset, int64_t t,
                                                   int e, int j, int32_t* __restrict__ act_a,
                                                   int32_t* __restrict__ act_p) {
  const uint32_t u = aie_counter_rng(seed, (uint64_t)(env_offset + e), (uint64_t)t, (uint64_t)j);
  if (j < P.n * P.act_a_width) {
    if (!act_a) return;
    int range;
    if (P.c.multi_action_mode_agents) range = P.n_sub_a ? P.sub_a_dim[j % P.act_a_width] + 1 : 1;
    else range = P.A;
    act_a[(int64_t)e * P.n * P.act_a_width + j] = (int32_t)(((uint64_t)u * (uint64_t)range) >> 32);
  } else {
    if (!act_p) return;
    const int range = P.c.multi_action_mode_planner ? P.sub_p_dim + 1 : 1 + P.n_sub_p * P.sub_p_dim;
    act_p[(int64_t)e * P.act_p_width + (j - P.n * P.act_a_width)] = (int32_t)(((uint64_t)u * (uint64_t)range) >> 32);
  }
}

// ------------------------------------------------------------------------------------
// record streaming HBM <-> LDS (16 B per lane, fully coalesced); MT key HBM <-> VGPRs
// ------------------------------------------------------------------------------------
struct MT {
  uint32_t r[10];  // word 64*j + lane (row 9: lanes 0..47)
  int pos;         // wave-uniform index of the next unused word (624 = twist first)
  int twists;      // mt_twist calls since the owner zeroed it (one-step-economy: the rows go back to HBM only then)
  // AIE_RNG_FAST (include/aie.h): the stream is Philox2x32-10 keyed per replica, consumed in blocks of 624 words with
  // the same position bookkeeping; `r` then holds the FINAL words of block `fblk` (no tempering), "twist" = next block
  bool fast;       // (a compile-time constant in the instances: P.c.rng_mode folds)
  uint32_t fkey, fblk, fsalt;  // wave-uniform
};
#define R_U32(c, off) (reinterpret_cast<uint32_t*>((c).rec + (c).P.off))

// ---- AIE_RNG_FAST: Philox2x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11;
// multiplier and Weyl key increment of Random123).  Word g of a replica's stream = element g & 1 of
// philox(counter = (lo32(g >> 1), hi32(g >> 1) | salt), key32).  One 32 x 32 -> 64 multiply and one three-input xor
// per round.
__device__ __forceinline__ void philox2x32_10(uint32_t c0, uint32_t c1, uint32_t key, uint32_t& o0, uint32_t& o1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p = (uint64_t)0xD256D193u * (uint64_t)c0;
    c0 = (uint32_t)(p >> 32) ^ key ^ c1;
    c1 = (uint32_t)p;
    key += 0x9E3779B9u;
  }
  o0 = c0;
  o1 = c1;
}
// words 2 h and 2 h + 1 of block `blk` (h may run past the block's 312 pairs: the stream is linear, block b starts at
// pair 312 b)
__device__ __forceinline__ void fast_pair(uint32_t key, uint32_t blk, uint32_t salt, int h, uint32_t& o0, uint32_t& o1) {
  const uint64_t pair = (uint64_t)blk * 312ull + (uint64_t)(uint32_t)h;
  philox2x32_10((uint32_t)pair, (uint32_t)(pair >> 32) | salt, key, o0, o1);
}

// (Until round 5 this copy also scanned the cells' flag bytes for the regeneration's source doubles, every step: 250 of a
// replica-step's 2 150 vector instructions on BASELINE configs[1].  The list is a record field now, o_src_list, kept by
// the reset kernel.)
// `wave` of `nwaves` copies every nwaves-th 16-byte unit; wave `key_wave` also takes the MT19937 key (registers).
__device__ __forceinline__ void load_record(const Ctx& c, const uint8_t* __restrict__ arena, MT& m, int wave = 0,
                                            int nwaves = 1, int key_wave = 0) {
  // (a_records == 0: the records open the arena, aie_layout.h -- not read from the parameter block: every scalar load a
  // workgroup needs before its first record load costs the launch 0.2 - 0.3 us, round 6)
  const uint8_t* g = arena + (int64_t)c.e * c.P.rec_bytes;
  const uint4* src = reinterpret_cast<const uint4*>(g);
  uint4* dst = reinterpret_cast<uint4*>(c.rec);
  const int nq = rec_lds_bytes(c.P) >> 4;
  for (int q = wave * AIE_NT + c.tid; q < nq; q += nwaves * AIE_NT) dst[q] = src[q];
  if (wave != key_wave || rng_fast(c.P)) return;  // (the counter stream's state came with the image: mt_fast_attach)
  const uint32_t* key = reinterpret_cast<const uint32_t*>(g + c.P.o_mt);
#pragma unroll
  for (int j = 0; j < 9; ++j) m.r[j] = key[64 * j + c.tid];
  m.r[9] = c.tid < 48 ? key[576 + c.tid] : 0u;
}
__device__ __forceinline__ void store_record(const Ctx& c, uint8_t* __restrict__ arena, const MT& m, int wave = 0,
                                             int nwaves = 1, int key_wave = 0) {
  uint8_t* g = arena + (int64_t)c.e * c.P.rec_bytes;  // (a_records == 0, see load_record)
  uint4* dst = reinterpret_cast<uint4*>(g);
  const uint4* src = reinterpret_cast<const uint4*>(c.rec);
  const int nq = rec_lds_bytes(c.P) >> 4;
  for (int q = wave * AIE_NT + c.tid; q < nq; q += nwaves * AIE_NT) dst[q] = src[q];
  if (wave != key_wave || rng_fast(c.P)) return;  // the generator's rows are in that wave's registers
  uint32_t* key = reinterpret_cast<uint32_t*>(g + c.P.o_mt);
#pragma unroll
  for (int j = 0; j < 9; ++j) key[64 * j + c.tid] = m.r[j];
  if (c.tid < 48) key[576 + c.tid] = m.r[9];
}
__device__ __forceinline__ uint16_t* dirty_list(const Ctx& c);
// The step's way out (round 6).  A step changes a handful of the H W map cells -- the ones its change log lists for the
// in-place map observations (dirty_*: every cell whose word or occupant changed) -- so the 16-byte units that hold
// nothing but cells stay as they are in HBM and the listed cells go out one word per lane: 2.4 of the 4.2 KB image of
// BASELINE configs[1] are not written.  A log overflow writes everything.  (The generator's rows left right behind the
// regeneration: store_generator_rows.)
__device__ __forceinline__ void store_record_step(const Ctx& c, uint8_t* __restrict__ arena, int wave, int nwaves) {
  uint8_t* g = arena + (int64_t)c.e * c.P.rec_bytes;
  uint4* dst = reinterpret_cast<uint4*>(g);
  const uint4* src = reinterpret_cast<const uint4*>(c.rec);
  const int nq = rec_lds_bytes(c.P) >> 4;
  const int cnt = uni(c.dirty[0]);
  const int q0 = (c.P.o_cells == 0 && cnt <= AIE_DIRTY_CAP) ? (4 * c.P.HW) >> 4 : 0;  // units [0, q0): cells only
  for (int q = q0 + wave * AIE_NT + c.tid; q < nq; q += nwaves * AIE_NT) dst[q] = src[q];
  if (wave == nwaves - 1 && q0 > 0 && c.tid < cnt) {
    const int cell = (int)dirty_list(c)[c.tid];
    reinterpret_cast<uint32_t*>(g + c.P.o_cells)[cell] = R_CELLS(c)[cell];
  }
}
__device__ __forceinline__ void store_generator_rows(const Ctx& c, uint8_t* __restrict__ arena, const MT& m) {
  if (rng_fast(c.P)) return;
  uint32_t* key = reinterpret_cast<uint32_t*>(arena + (int64_t)c.e * c.P.rec_bytes + c.P.o_mt);
#pragma unroll
  for (int j = 0; j < 9; ++j) key[64 * j + c.tid] = m.r[j];
  if (c.tid < 48) key[576 + c.tid] = m.r[9];
}

// ------------------------------------------------------------------------------------
// NumPy legacy RandomState stream (MT19937), one per replica.
// The reference draws from the process-global np.random (F/base/base_env.py:493,
// F/base/world.py:420, F/components/move.py:138, layout_from_file.py:361-366,400).
// ------------------------------------------------------------------------------------
// one word of the twist: y = (a & UPPER) | (b & LOWER); m ^ (y >> 1) ^ (y & 1 ? MATRIX_A : 0) -- five VALU operations
// (bit-field insert, shift, sign-extended bit 0, one three-input bit operation, xor); the compiler's own selection of
// the plain C expression takes seven, and the twist is a third of the step kernel's vector instructions
__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t m) {
  uint32_t y;
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(y) : "s"(0x7fffffffu), "v"(b), "v"(a));  // (b & mask) | (a & ~mask)
  const uint32_t low = (uint32_t)__builtin_amdgcn_sbfe((int)b, 0, 1);           // y & 1 ? 0xffffffff : 0
  return __builtin_amdgcn_bitop3_b32(low, 0x9908b0dfu, y >> 1, 0x6a) ^ m;        // ((low & MATRIX_A) ^ (y >> 1)) ^ m
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= (y >> 18);
  return y;
}
__device__ __forceinline__ uint32_t lane_get(uint32_t v, int src_lane) { return (uint32_t)__shfl((int)v, src_lane, 64); }

// Whole-state twist in registers.  Word k = 64*J + l of the next state needs words k,
// k+1 and k+397 (mod 624, with the sequential in-place semantics: indices that wrap
// refer to ALREADY UPDATED words).  k+397 = 64*(J+6) + l+13 and k-227 = 64*(J-4) + l+29,
// so every row is two lane rotations (by 13 or by 29) of two other rows.
// The two source rows of a row feed DISJOINT source lanes (l < SPLIT reads lanes [ROT, 64) of ROW_LO, the rest
// lanes [0, ROT) of ROW_HI), so they are merged per source lane first and fetched with one permute; the
// neighbour word k+1 is a one-lane wave rotation (DPP), with the last lane patched from the next row.
__device__ __forceinline__ uint32_t lane_rol1(uint32_t v) {  // lane l <- lane (l + 1) & 63
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x134 /* wave_rol:1 */, 0xf, 0xf, false);
}
// lane l <- lane l + 1 of `v`, lane LAST <- the (wave-uniform) first word of the next row.  LAST == 63: ONE DPP move --
// a wavefront shift leaves the last lane, which has no source, at the destination's previous value (bound_ctrl off);
// a select on `lane == 63` compiles to a branch around a move
template <int LAST>
__device__ __forceinline__ uint32_t lane_next_word(uint32_t v, uint32_t next0, int lane) {
  if (LAST == 63) return (uint32_t)__builtin_amdgcn_update_dpp((int)next0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, false);
  const uint32_t b = lane_rol1(v);
  return (lane == LAST) ? next0 : b;
}
#define AIE_MT_ROW(J, ROT, SPLIT, ROW_LO, ROW_HI, NEXT0)                                      \
  {                                                                                          \
    const uint32_t a = m.r[J];                                                               \
    const uint32_t b = lane_next_word<(J) == 9 ? 47 : 63>(a, (NEXT0), lane);                 \
    const uint32_t src = lane >= (ROT) ? (ROW_LO) : (ROW_HI);                                \
    const uint32_t x = lane_get(src, (lane + (ROT)) & 63);                                   \
    m.r[J] = mt_mix(a, b, x);                                                                \
  }
__device__ __forceinline__ void mt_twist_body(MT& m, int lane) {
  // rows 0..2: old[k+397] from rows J+6 (l < 51) / J+7 (l >= 51), rotation 13
  AIE_MT_ROW(0, 13, 51, m.r[6], m.r[7], bcast(m.r[1], 0))
  AIE_MT_ROW(1, 13, 51, m.r[7], m.r[8], bcast(m.r[2], 0))
  AIE_MT_ROW(2, 13, 51, m.r[8], m.r[9], bcast(m.r[3], 0))
  {  // row 3: l < 35 old row 9 (rotation 13), l >= 35 NEW row 0 (rotation 29)
    const uint32_t a = m.r[3];
    const uint32_t b = lane_next_word<63>(a, bcast(m.r[4], 0), lane);
    const uint32_t x_lo = lane_get(m.r[9], (lane + 13) & 63);
    const uint32_t x_hi = lane_get(m.r[0], (lane + 29) & 63);
    m.r[3] = mt_mix(a, b, lane < 35 ? x_lo : x_hi);
  }
  // rows 4..9: NEW[k-227] from rows J-4 (l < 35) / J-3 (l >= 35), rotation 29
  AIE_MT_ROW(4, 29, 35, m.r[0], m.r[1], bcast(m.r[5], 0))
  AIE_MT_ROW(5, 29, 35, m.r[1], m.r[2], bcast(m.r[6], 0))
  AIE_MT_ROW(6, 29, 35, m.r[2], m.r[3], bcast(m.r[7], 0))
  AIE_MT_ROW(7, 29, 35, m.r[3], m.r[4], bcast(m.r[8], 0))
  AIE_MT_ROW(8, 29, 35, m.r[4], m.r[5], bcast(m.r[9], 0))
  AIE_MT_ROW(9, 29, 35, m.r[5], m.r[6], bcast(m.r[0], 0))  // word 623 pairs with NEW word 0
}

// One out-of-line copy of the twist (rows travel in VGPRs by value): it is reached from
// every draw site, and inlining ~150 instructions x 8 sites would overflow the I-cache.
struct MTRows { uint32_t r[10]; };
__device__ __attribute__((noinline)) MTRows mt_twist_rows(MTRows in, int lane) {
  MT m;
  m.fast = false;
#pragma unroll
  for (int j = 0; j < 10; ++j) m.r[j] = in.r[j];
  m.pos = 0;
  mt_twist_body(m, lane);
  MTRows out;
#pragma unroll
  for (int j = 0; j < 10; ++j) out.r[j] = m.r[j];
  return out;
}
// AIE_RNG_FAST: the 624 words of block m.fblk in the row layout (word 64 J + l in r[J], lane l).  Lane l computes the
// pairs 64 j + l (j = 0..4: words 128 j + 2 l, + 1); row 2 j takes its words from lanes l >> 1, row 2 j + 1 from lanes
// 32 + (l >> 1).  Five Philox blocks and twenty lane permutes per 624 words: costlier than an MT19937 twist (the
// sequential consumers -- resets, layout generation -- pay that; the step kernel never comes here, it addresses the
// words it needs directly).
__device__ __attribute__((noinline)) MTRows mt_fast_rows_of(uint32_t key, uint32_t blk, uint32_t salt, int lane) {
  MTRows out;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    uint32_t x0, x1;
    fast_pair(key, blk, salt, 64 * j + lane, x0, x1);
    const uint32_t lo0 = lane_get(x0, lane >> 1), lo1 = lane_get(x1, lane >> 1);
    const uint32_t hi0 = lane_get(x0, 32 + (lane >> 1)), hi1 = lane_get(x1, 32 + (lane >> 1));
    out.r[2 * j] = (lane & 1) ? lo1 : lo0;
    out.r[2 * j + 1] = (lane & 1) ? hi1 : hi0;
  }
  return out;
}
__device__ __forceinline__ void mt_fast_rows(MT& m, int lane) {
  const MTRows t = mt_fast_rows_of(m.fkey, m.fblk, m.fsalt, lane);
#pragma unroll
  for (int j = 0; j < 10; ++j) m.r[j] = t.r[j];
}
__device__ __forceinline__ void mt_twist(MT& m, int lane) {
  if (m.fast) {
    m.fblk += 1;
    mt_fast_rows(m, lane);
    m.twists += 1;
    return;
  }
  MTRows t;
#pragma unroll
  for (int j = 0; j < 10; ++j) t.r[j] = m.r[j];
  t = mt_twist_rows(t, lane);
#pragma unroll
  for (int j = 0; j < 10; ++j) m.r[j] = t.r[j];
  m.twists += 1;
}

// a row register's value as the stream's word: MT19937 tempers its state words, the counter stream's rows are final
__device__ __forceinline__ uint32_t mt_word(const MT& m, uint32_t raw) { return m.fast ? raw : mt_temper(raw); }
// A generator for a kernel section: MT19937 (rows arrive with load_record) or the counter stream (mt_fast_attach once the
// record is in LDS).
__device__ __forceinline__ void mt_init(MT& m, const aie_params& P) {
  m.fast = rng_fast(P);
  m.pos = AIE_MT_N;
  m.twists = 0;
  m.fkey = m.fblk = m.fsalt = 0u;
}
__device__ __forceinline__ void mt_copy(MT& d, const MT& s) {  // (rows and, for the counter stream, which block they are)
#pragma unroll
  for (int j = 0; j < 10; ++j) d.r[j] = s.r[j];
  d.fast = s.fast;
  d.fkey = s.fkey;
  d.fblk = s.fblk;
  d.fsalt = s.fsalt;
  d.twists = s.twists;
}
// AIE_RNG_FAST, once the record is in LDS and m.pos is set: the stream's identity from the image, and the current
// block's rows if words of it are still to come (pos == 624: the next draw starts a block anyway).  detach: the block
// number goes back into the image (every lane stores the same value).
__device__ __forceinline__ void mt_fast_attach(const Ctx& c, MT& m) {
  if (!m.fast) return;
  const uint32_t* st = R_U32(c, o_mt);
  m.fkey = (uint32_t)uni((int)st[0]);
  m.fblk = (uint32_t)uni((int)st[1]);
  m.fsalt = (uint32_t)uni((int)st[2]);
  if (m.pos < AIE_MT_N) mt_fast_rows(m, c.tid & (AIE_NT - 1));
}
__device__ __forceinline__ void mt_fast_detach(const Ctx& c, const MT& m) {
  if (m.fast) R_U32(c, o_mt)[1] = m.fblk;
}
// sequential draws (wave-uniform: every lane gets the same value)
__device__ __forceinline__ uint32_t rng_u32(MT& m, int lane) {
  if (m.pos >= AIE_MT_N) {
    mt_twist(m, lane);
    m.pos = 0;
  }
  const int row = m.pos >> 6;
  uint32_t v = m.r[0];
#pragma unroll
  for (int j = 1; j < 10; ++j) v = (row == j) ? m.r[j] : v;
  const uint32_t w = bcast(v, m.pos & 63);
  m.pos += 1;
  return mt_word(m, w);
}
__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// numpy float64 add.reduce over n <= 128 contiguous values (pairwise summation with an
// 8-way unrolled head, numpy/_core/src/umath/loops_utils.h.src): decisions such as
// "mean agent reward > 0" (layout_from_file.py:554-557) depend on this exact order.
__device__ __forceinline__ double np_sum_small(const double* a, int n) {
  if (n < 8) {
    double res = -0.0;
    for (int i = 0; i < n; ++i) res += a[i];
    return res;
  }
  double r0 = a[0], r1 = a[1], r2 = a[2], r3 = a[3], r4 = a[4], r5 = a[5], r6 = a[6], r7 = a[7];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 += a[i + 0]; r1 += a[i + 1]; r2 += a[i + 2]; r3 += a[i + 3];
    r4 += a[i + 4]; r5 += a[i + 5]; r6 += a[i + 6]; r7 += a[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; ++i) res += a[i];
  return res;
}

// The same reduction over sequences of up to 1024 elements that are not stored anywhere, evaluated by the whole wave:
// element idx is a[idx] (mode 0) or |a[idx / n] - a[idx % n]| (mode 1: the flattened n x n matrix
// np.abs(endowments[:, None] - endowments[None, :]) of social_metrics.get_gini, social_metrics.py:36-41).
// Lanes 0..7 carry numpy's eight accumulators; blocks above 128 elements split as numpy's recursion does.
// M = 65536 / n + 1 (exact idx / n for idx < 2259 and n <= 29).  The result is wave-uniform.
__device__ __forceinline__ double np_seq_elem(const double* a, int n, int mode, uint32_t M, int idx) {
  if (mode == 0) return a[idx];
  const int r = (int)(((uint32_t)idx * M) >> 16);
  return fabs(a[r] - a[idx - r * n]);
}
__device__ __attribute__((noinline)) double np_sum_leaf(const double* a, int n, int mode, uint32_t M, int o, int m,
                                                        int lane) {
  if (m < 8) {
    double res = -0.0;
    for (int i = 0; i < m; ++i) res += np_seq_elem(a, n, mode, M, o + i);
    return res;
  }
  const int k = lane & 7, body = m - (m % 8);
  double r = np_seq_elem(a, n, mode, M, o + k);
  for (int i = 8; i < body; i += 8) r += np_seq_elem(a, n, mode, M, o + i + k);
  const double r0 = bcast(r, 0), r1 = bcast(r, 1), r2 = bcast(r, 2), r3 = bcast(r, 3);
  const double r4 = bcast(r, 4), r5 = bcast(r, 5), r6 = bcast(r, 6), r7 = bcast(r, 7);
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (int i = body; i < m; ++i) res += np_seq_elem(a, n, mode, M, o + i);
  return res;
}
template <int D>
__device__ __forceinline__ double np_sum_seq(const double* a, int n, int mode, uint32_t M, int o, int m, int lane) {
  if constexpr (D == 0) {
    return np_sum_leaf(a, n, mode, M, o, m, lane);
  } else {
    if (m <= 128) return np_sum_leaf(a, n, mode, M, o, m, lane);
    int n2 = m / 2;
    n2 -= n2 % 8;
    const double lo = np_sum_seq<D - 1>(a, n, mode, M, o, n2, lane);
    return lo + np_sum_seq<D - 1>(a, n, mode, M, o + n2, m - n2, lane);
  }
}

__device__ __forceinline__ double rng_double(MT& m, int lane) {
  const uint32_t a = rng_u32(m, lane);
  const uint32_t b = rng_u32(m, lane);
  return u53(a, b);
}
// random_interval(max): masked rejection on 32-bit words
__device__ __forceinline__ uint32_t rng_interval(MT& m, int lane, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max, v;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  while ((v = (rng_u32(m, lane) & mask)) > max) {}
  return v;
}
// np.random.permutation(n): World.get_random_order_agents, F/base/world.py:418-422.
// Returned lane-distributed: lane k holds the k-th agent of the random order.
__device__ __forceinline__ int rng_permutation(MT& m, int lane, int n) {
  int p = lane;
  for (int i = n - 1; i >= 1; --i) {
    const int j = (int)rng_interval(m, lane, (uint32_t)i);
    const int vi = bcast(p, i), vj = bcast(p, j);
    p = (lane == i) ? vj : ((lane == j) ? vi : p);
  }
  return p;
}
// ---- the same draws for the step kernel's components.  The generator state (624 words) never enters LDS: the
// wave that regenerates the resources afterwards holds it in 10 VGPRs from the record load on, and publishes the
// next stage_window_words() TEMPERED words of the stream in a small LDS draw window for the wave that runs the
// components (register-starved under the 64-VGPR budget): a draw is one LDS broadcast read per 64 draws plus a
// v_readlane.  A step that draws past the window (many agents) or past word 623 refills it from the state in HBM
// (rng_refill; a twist there writes the new state back, and the holder of the registers re-reads it).
struct MTL {
  uint32_t* w;  // LDS draw window: tempered words base ... base + avail - 1 of the generator's current window
  int pos;      // wave-uniform index of the next unused word of the generator's window (624 = twist first)
  // 64 words of the draw window in a register (lane j: word cbase + j): a draw is a v_readlane
  uint32_t cache;
  int cbase;
  // this step's changes to the maps, kept in registers until the components are done (-> c.dirty):
  int dn;             // cells appended to the change list
  uint32_t mv0, mv1;  // agents that moved
  int base, avail;    // what the draw window holds
  int tw;             // twists rng_refill performed this step (-> c.dirty[3])
  uint32_t* gkey;     // the replica's generator state in HBM (record + o_mt); AIE_RNG_FAST: in the record's LDS image
  int cap;            // capacity of the draw window in words
  bool fast;          // AIE_RNG_FAST (compile-time in the instances)
};
// Tempered words [pos, pos + avail) of the state in m -> w[0, avail); returns avail = min(cap, 624 - pos).
__device__ __forceinline__ int draw_window_publish(uint32_t* w, int cap, const MT& m, int pos, int lane) {
  const int avail = cap < AIE_MT_N - pos ? cap : AIE_MT_N - pos;
  const int r0 = pos >> 6;
  // (the rows as opaque register values: left to itself the optimiser turns the select chain below into ONE load with a
  // run-time index -- and the ten rows into a stack array in scratch memory for it)
  uint32_t rr[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    rr[j] = m.r[j];
    asm("" : "+v"(rr[j]));
  }
  for (int d = 0; d * 64 < cap + 64; ++d) {  // lane l of row r holds word 64 r + l
    const int r = r0 + d;
    if (r > 9) break;
    uint32_t v = rr[0];
#pragma unroll
    for (int j = 1; j < 10; ++j) v = (r == j) ? rr[j] : v;
    const int slot = 64 * r + lane - pos;
    if (slot >= 0 && slot < avail) w[slot] = v;  // (raw state words: the reader tempers the 64 it caches, rng_u32)
  }
  AIE_WSYNC();
  return avail;
}
// The same from the state in HBM: only the rows the window covers are fetched (step start: the wave that will hold
// the state later has nothing but the record copy to do, and the fetch rides behind it).
__device__ __forceinline__ int draw_window_publish_from_hbm(uint32_t* w, int cap, const uint32_t* __restrict__ gkey, int pos,
                                                            int lane) {
  const int avail = cap < AIE_MT_N - pos ? cap : AIE_MT_N - pos;
  const int r0 = pos >> 6;
  for (int d = 0; d * 64 < cap + 64; ++d) {
    const int r = r0 + d, i = 64 * r + lane;
    if (r > 9) break;
    const int slot = i - pos;
    if (slot >= 0 && slot < avail) w[slot] = gkey[i];  // (slot < avail implies i < 624; raw: see draw_window_publish)
  }
  return avail;
}
__device__ __forceinline__ void mt_rows_from_hbm(MT& m, const uint32_t* gkey, int lane) {
  // agent-scope loads: the rows may have been rewritten by the other wave of the workgroup during this launch
#pragma unroll
  for (int j = 0; j < 9; ++j) m.r[j] = __hip_atomic_load(gkey + 64 * j + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  m.r[9] = lane < 48 ? __hip_atomic_load(gkey + 576 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
struct Refill { int pos, avail, twisted; };
// Out of line (rare: a step in ~30 crosses word 623 with 4 agents; steps with ~100 agents outrun the window): the
// state comes from HBM -- the record's copy, or what an earlier refill of this step wrote back.
__device__ __attribute__((noinline, cold)) Refill rng_refill(uint32_t* gkey, uint32_t* w, int cap, int pos, int lane) {
  MT m;
  m.fast = false;  // (MT19937 only: the counter stream refills through rng_refill_fast)
  mt_rows_from_hbm(m, gkey, lane);
  int twisted = 0;
  if (pos >= AIE_MT_N) {
    mt_twist_body(m, lane);
    pos = 0;
    twisted = 1;
#pragma unroll
    for (int j = 0; j < 9; ++j) __hip_atomic_store(gkey + 64 * j + lane, m.r[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane < 48) __hip_atomic_store(gkey + 576 + lane, m.r[9], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const int avail = draw_window_publish(w, cap, m, pos, lane);
  return Refill{pos, avail, twisted};
}
// ---- AIE_RNG_FAST: the draw window straight from the counter stream.  Words [pos, pos + avail) of block `blk` -> w[0,
// avail): lane l of pass d computes pair (pos >> 1) + 64 d + l, i.e. window slots 2 (64 d + l) - (pos & 1) and the next.
__device__ __forceinline__ int draw_window_publish_fast(uint32_t* w, int cap, uint32_t key, uint32_t blk, uint32_t salt, int pos,
                                                        int lane) {
  const int npass = (cap + 127) >> 7;
  int avail = cap < AIE_MT_N - pos ? cap : AIE_MT_N - pos;
  if (avail > 128 * npass - (pos & 1)) avail = 128 * npass - (pos & 1);
  for (int d = 0; d < npass; ++d) {
    if (128 * d - (pos & 1) >= avail) break;
    uint32_t x0, x1;
    fast_pair(key, blk, salt, (pos >> 1) + 64 * d + lane, x0, x1);
    const int slot = 2 * (64 * d + lane) - (pos & 1);
    if (slot >= 0 && slot < avail) w[slot] = x0;
    if (slot + 1 < avail) w[slot + 1] = x1;
  }
  return avail;
}
// `st`: the stream's state in the record's LDS image (key32, block number, salt).  The wave that runs the components
// moves to the next block here; the other wave of the replica picks the block number up behind the workgroup barrier.
__device__ __attribute__((noinline, cold)) Refill rng_refill_fast(uint32_t* st, uint32_t* w, int cap, int pos, int lane) {
  uint32_t blk = (uint32_t)uni((int)st[1]);
  int twisted = 0;
  if (pos >= AIE_MT_N) {
    blk += 1u;
    pos = 0;
    twisted = 1;
    if (lane == 0) st[1] = blk;
  }
  const int avail = draw_window_publish_fast(w, cap, (uint32_t)uni((int)st[0]), blk, (uint32_t)uni((int)st[2]), pos, lane);
  AIE_WSYNC();
  return Refill{pos, avail, twisted};
}
__device__ __forceinline__ uint32_t rng_u32(MTL& l, int lane) {
  if (__builtin_expect(l.pos >= l.base + l.avail, 0)) {
    const Refill r = l.fast ? rng_refill_fast(l.gkey, l.w, l.cap, l.pos, lane) : rng_refill(l.gkey, l.w, l.cap, l.pos, lane);
    l.pos = uni(r.pos);  // (a function's results come back in vector registers: keep the bookkeeping scalar)
    l.base = l.pos;
    l.avail = uni(r.avail);
    l.tw += uni(r.twisted);
    l.cbase = -AIE_MT_N;
  }
  int k = l.pos - l.cbase;
  if ((unsigned)k >= (unsigned)AIE_NT) {  // the next (up to) 64 words of the draw window
    const int idx = l.pos - l.base + lane;
    // MT19937's window holds RAW state words (round 6): a step draws one or two dozen of the 128+ the window offers, so the
    // tempering happens here, once per 64 cached words, instead of over the whole window on the way in
    const uint32_t raw = l.w[idx < l.avail ? idx : l.avail - 1];
    l.cache = l.fast ? raw : mt_temper(raw);
    l.cbase = l.pos;
    k = 0;
  }
  l.pos += 1;
  return bcast(l.cache, k);
}
__device__ __forceinline__ double rng_double(MTL& l, int lane) {
  const uint32_t a = rng_u32(l, lane);
  const uint32_t b = rng_u32(l, lane);
  return u53(a, b);
}
// (Tried: finding the first acceptable word of the cached block with one ballot instead of a readlane / compare / branch
// per attempt -- bit-identical, but the unrolled permutation loops grow and the launch gets slower: C2 24.9 -> 25.7 us,
// C3 41.6 -> 43.7 us.)
__device__ __forceinline__ uint32_t rng_interval(MTL& l, int lane, uint32_t max) {
  if (max == 0) return 0;
  uint32_t mask = max, v;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
  while ((v = (rng_u32(l, lane) & mask)) > max) {}
  return v;
}
__device__ __forceinline__ int rng_permutation(MTL& l, int lane, int n) {
  int p = lane;
  for (int i = n - 1; i >= 1; --i) {
    const int j = (int)rng_interval(l, lane, (uint32_t)i);
    const int vi = bcast(p, i), vj = bcast(p, j);
    p = (lane == i) ? vj : ((lane == j) ? vi : p);
  }
  return p;
}
__device__ __forceinline__ double rng_gauss(const Ctx& c, MT& m) {  // legacy_gauss (polar Box-Muller, cached)
  int32_t* has = R_I32(c, o_mt_has_gauss);
  double* g = R_F64(c, o_mt_gauss);
  if (*has) {
    const double t = *g;
    *has = 0;
    *g = 0.0;
    return t;
  }
  double f, x1, x2, r2;
  do {
    x1 = 2.0 * rng_double(m, c.tid) - 1.0;
    x2 = 2.0 * rng_double(m, c.tid) - 1.0;
    r2 = x1 * x1 + x2 * x2;
  } while (r2 >= 1.0 || r2 == 0.0);
  f = sqrt(-2.0 * aie_log_glibc(r2) / r2);  // libm's log bit for bit (aie_glibc_math.h); sqrt and / are IEEE-exact
  *g = f * x1;
  *has = 1;
  return f * x2;
}
__device__ __forceinline__ double rng_pareto(MT& m, int lane, double a) { return aie_exp_glibc(-aie_log_glibc(1.0 - rng_double(m, lane)) / a) - 1.0; }
__device__ __forceinline__ double rng_lognormal(const Ctx& c, MT& m, double mean, double sigma) { return aie_exp_glibc(mean + sigma * rng_gauss(c, m)); }

// ------------------------------------------------------------------------------------
// World helpers (F/base/world.py)
// ------------------------------------------------------------------------------------
// World.can_agent_occupy, world.py:424-440: in bounds, no Water, House only if owner
// (world.py:213-217, 256-258, 300-305), and unoccupied.
__device__ __forceinline__ bool can_agent_occupy(const Ctx& c, int r, int col, int agent) {
  if (r < 0 || r >= c.P.H || col < 0 || col >= c.P.W) return false;
  const int cell = r * c.P.W + col;
  const uint32_t w = R_CELLS(c)[cell];
  if (AIE_CELL_FLAGS(w) & AIE_CELL_WATER) return false;
  const int o = AIE_CELL_OWNER(w);
  if (!(o < 0 || o == agent)) return false;
  const int occ = c.locmap[cell];
  return occ == 0 || occ == agent + 1;
}

// Dense-log event of a logged replica (include/aie.h: AIE_EV_*): wave-uniform arguments, lane 0
// writes the row; the per-step row count lives in LDS (c.srcn[2]) until the components are done.
__device__ __forceinline__ void log_event(const Ctx& c, int type, int a1, int a2, int a3, int a4, int a5,
                                          int a6, int a7, int a8, double f) {
  if (c.tid != 0) return;
  const int k = c.srcn[2];
  if (k >= c.R.ev_cap) return;
  int32_t* row = c.ev + 4 + k * AIE_EV_WORDS;
  row[0] = type; row[1] = a1; row[2] = a2; row[3] = a3; row[4] = a4;
  row[5] = a5; row[6] = a6; row[7] = a7; row[8] = a8; row[9] = 0;
  *reinterpret_cast<double*>(row + 10) = f;
  c.srcn[2] = k + 1;
}

// Build.agent_can_build, F/components/build.py:70-83 (+ world.py:284-293)
__device__ __forceinline__ bool agent_can_build(const Ctx& c, int i) {
  const int n = c.P.n;
  const int32_t* inv = R_I32(c, o_inv_res);
  if (inv[n + i] < 1 || inv[i] < 1) return false;
  const uint32_t w = R_CELLS(c)[R_I32(c, o_loc_r)[i] * c.P.W + R_I32(c, o_loc_c)[i]];
  // no resource, no house (owner byte 0xff = none), no water / source block
  return (w & 0xffffu) == 0 && ((w >> 16) & 0xffu) == 0xffu && (w >> 24) == 0;
}

// ------------------------------------------------------------------------------------
// Agent state in registers: lane i < n holds agent i.
// ------------------------------------------------------------------------------------
struct Agents {
  int lr, lc;            // location
  int inv0, inv1;        // Stone, Wood in inventory
  int esc0, esc1;        // Stone, Wood in escrow
  int no0, no1;          // open orders per commodity (CDA n_orders)
  uint32_t act;          // bit 0 build | gather << 1 (3 bits) | buy0 << 4 | sell0 << 11 | buy1 << 18 | sell1 << 25
  double coin, esc_coin, labor;
};
#define AIE_ACT_BUILD(a) ((a) & 1u)
#define AIE_ACT_GATHER(a) (((a) >> 1) & 7u)
#define AIE_ACT_BUY(a, r) (((a) >> (4 + 14 * (r))) & 0x7fu)
#define AIE_ACT_SELL(a, r) (((a) >> (11 + 14 * (r))) & 0x7fu)

__device__ __forceinline__ void agents_load(const Ctx& c, Agents& A) {
  const int n = c.P.n, i = c.tid < n ? c.tid : 0;
  A.lr = R_I32(c, o_loc_r)[i];
  A.lc = R_I32(c, o_loc_c)[i];
  A.inv0 = R_I32(c, o_inv_res)[i];
  A.inv1 = R_I32(c, o_inv_res)[n + i];
  A.esc0 = R_I32(c, o_esc_res)[i];
  A.esc1 = R_I32(c, o_esc_res)[n + i];
  A.no0 = c.P.has_cda ? R_I32(c, o_cda_n_orders)[i] : 0;
  A.no1 = c.P.has_cda ? R_I32(c, o_cda_n_orders)[n + i] : 0;
  A.coin = R_F64(c, o_inv_coin)[i];
  A.esc_coin = R_F64(c, o_esc_coin)[i];
  A.labor = R_F64(c, o_labor)[i];
}
__device__ __forceinline__ void agents_store(const Ctx& c, const Agents& A) {
  const int n = c.P.n, i = c.tid;
  if (i < n) {
    R_I32(c, o_loc_r)[i] = A.lr;
    R_I32(c, o_loc_c)[i] = A.lc;
    R_I32(c, o_inv_res)[i] = A.inv0;
    R_I32(c, o_inv_res)[n + i] = A.inv1;
    R_I32(c, o_esc_res)[i] = A.esc0;
    R_I32(c, o_esc_res)[n + i] = A.esc1;
    if (c.P.has_cda) {
      R_I32(c, o_cda_n_orders)[i] = A.no0;
      R_I32(c, o_cda_n_orders)[n + i] = A.no1;
    }
    R_F64(c, o_inv_coin)[i] = A.coin;
    R_F64(c, o_esc_coin)[i] = A.esc_coin;
    R_F64(c, o_labor)[i] = A.labor;
  }
}

// parse_actions base_env.py:552-556 -> base_agent.py:407-438: lane i decodes agent i's
// action into its packed per-subspace word; lane b decodes planner bracket b.
// Returns the AIE_ERR_* bits of out-of-range indices (wave-uniform); the caller ORs them into the record's error_flags
// once the record is in LDS -- the action loads themselves leave ahead of it (step_body).
__device__ __forceinline__ int decode_actions(const Ctx& c, Agents& A, const int32_t* __restrict__ aa,
                               const int32_t* __restrict__ ap) {
  const aie_params& P = c.P;
  const int i = c.tid;
  uint32_t act = 0;
  bool bad_a = false, bad_p = false;  // out-of-range indices: NO-OP here, an exception in the reference (AIE_ERR_*)
  if (i < P.n && aa) {
    const int32_t* a = aa + ((int64_t)c.e * P.n + i) * P.act_a_width;
    static const int shift[AIE_N_SUB_SLOTS] = {0, 4, 11, 18, 25, 1, 0};
    if (P.c.multi_action_mode_agents) {
      for (int s = 0; s < P.n_sub_a; ++s) {
        const int v = a[s];
        if (v >= 0 && v <= P.sub_a_dim[s]) act |= (uint32_t)v << shift[P.sub_a_slot[s]];
        else bad_a = true;
      }
    } else {
      const int v = a[0];
      bad_a = v < 0 || v >= P.A;
      for (int s = 0; s < P.n_sub_a; ++s)
        if (v >= P.sub_a_base[s] && v < P.sub_a_base[s] + P.sub_a_dim[s])
          act |= (uint32_t)(v - P.sub_a_base[s] + 1) << shift[P.sub_a_slot[s]];
    }
  }
  A.act = act;
  if (i < AIE_MAX_BRACKETS) {
    int v = 0;
    if (ap && i < P.n_sub_p) {
      const int32_t* a = ap + (int64_t)c.e * P.act_p_width;
      if (P.c.multi_action_mode_planner) {
        v = a[i];
        bad_p = v < 0 || v > P.sub_p_dim;  // (the tax component ignores what its mask forbids; out of range is an error)
        if (bad_p) v = 0;
      } else {
        const int x = a[0];
        bad_p = x < 0 || x >= 1 + P.n_sub_p * P.sub_p_dim;
        if (x >= 1 && x < 1 + P.n_sub_p * P.sub_p_dim && (x - 1) / P.sub_p_dim == i) v = (x - 1) % P.sub_p_dim + 1;
      }
    }
    c.act_p[i] = v;
  }
  return (__ballot(bad_a) ? AIE_ERR_AGENT_ACTION : 0) | (__ballot(bad_p) ? AIE_ERR_PLANNER_ACTION : 0);
}

// ------------------------------------------------------------------------------------
// Change log of one step.  The map observations (egocentric crops, planner map: 85 % of a
// step's output bytes) live in the arena from one step to the next, and a step changes only a
// handful of map cells, so the step kernel rewrites just those (update_spatial_observations)
// instead of all of them.  Every map cell whose packed word or occupant changes is appended
// to a short LDS list, every agent that moves gets a bit in the moved mask; a list overflow
// makes the step fall back to the full rewrite.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t* dirty_list(const Ctx& c) { return reinterpret_cast<uint16_t*>(c.dirty + 4); }
// wave-uniform call site (all lanes pass the same cell): lane 0 appends
__device__ __forceinline__ void dirty_add_uniform(const Ctx& c, MTL& m, int cell) {
  if (c.tid == 0 && m.dn < AIE_DIRTY_CAP) dirty_list(c)[m.dn] = (uint16_t)cell;
  m.dn += 1;
}
// divergent call site (each active lane its own cell)
__device__ __forceinline__ void dirty_add_lane(const Ctx& c, int cell) {
  const int s = atomicAdd(&c.dirty[0], 1);
  if (s < AIE_DIRTY_CAP) dirty_list(c)[s] = (uint16_t)cell;
}
__device__ __forceinline__ void dirty_agent_moved(MTL& m, int i) {
  if (i < 32) m.mv0 |= 1u << i; else m.mv1 |= 1u << (i - 32);
}

// ------------------------------------------------------------------------------------
// Build.component_step, F/components/build.py:112-161.  Wave-uniform control flow; the
// builders are found with one ballot, so the common "nobody builds" step costs only the
// permutation draw (which the reference consumes regardless, build.py:121).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void build_component_step(const Ctx& c, MTL& m, Agents& A) {
  const int n = c.P.n, lane = c.tid;
  const int perm = rng_permutation(m, lane, n);
  const uint64_t builders = __ballot(lane < n && AIE_ACT_BUILD(A.act));
  if (builders == 0) return;
  uint32_t* cells = R_CELLS(c);
  for (int k = 0; k < n; ++k) {
    const int i = bcast(perm, k);
    if (!((builders >> i) & 1ull)) continue;
    // agent_can_build, build.py:70-83 (+ world.py:284-293)
    const int cell = bcast(A.lr, i) * c.P.W + bcast(A.lc, i);
    const uint32_t w = cells[cell];
    const bool ok = bcast(A.inv0, i) >= 1 && bcast(A.inv1, i) >= 1 && (w & 0xffffu) == 0 &&
                    ((w >> 16) & 0xffu) == 0xffu && (w >> 24) == 0;
    if (!ok) continue;
    if (lane == i) {
      A.inv0 -= 1;
      A.inv1 -= 1;
      A.coin += R_F64(c, o_build_payment)[i];
      A.labor += c.R.c.build_labor;
    }
    cells[cell] = (w & 0xff00ffffu) | ((uint32_t)i << 16);  // world.py:474-479 (every lane, same value)
    dirty_add_uniform(c, m, cell);
    if (c.ev) log_event(c, AIE_EV_BUILD, i, cell / c.P.W, cell % c.P.W, 0, 0, 0, 0, 0, R_F64(c, o_build_payment)[i]);
  }
}

// ------------------------------------------------------------------------------------
// Gather.component_step, F/components/move.py:93-153 (wave-uniform control flow)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void gather_component_step_serial(const Ctx& c, MTL& m, Agents& A) {
  const int n = c.P.n, W = c.P.W, H = c.P.H, lane = c.tid;
  const int perm = rng_permutation(m, lane, n);
  uint32_t* cells = R_CELLS(c);
  const double my_bonus = R_F64(c, o_bonus_gather_prob)[lane < n ? lane : 0];
  for (int k = 0; k < n; ++k) {
    const int i = bcast(perm, k);
    const int a = (int)AIE_ACT_GATHER(bcast(A.act, i));
    const int r = bcast(A.lr, i), col = bcast(A.lc, i);
    int land = r * W + col;
    if (a != 0) {
      // 1 Left, 2 Right, 3 Up, 4 Down (move.py:116-123)
      const int nr = r + (a == 3 ? -1 : a == 4 ? 1 : 0);
      const int nc = col + (a == 1 ? -1 : a == 2 ? 1 : 0);
      if (nr >= 0 && nr < H && nc >= 0 && nc < W) {
        // World.can_agent_occupy, world.py:424-440
        const int tcell = nr * W + nc;
        const uint32_t tw = cells[tcell];
        const int occ = c.locmap[tcell];
        const int own = AIE_CELL_OWNER(tw);
        if (!(AIE_CELL_FLAGS(tw) & AIE_CELL_WATER) && (own < 0 || own == i) && occ == 0) {
          c.locmap[land] = 0;
          c.locmap[tcell] = (uint8_t)(i + 1);
          dirty_add_uniform(c, m, land);
          dirty_add_uniform(c, m, tcell);
          dirty_agent_moved(m, i);
          if (lane == i) {
            A.lr = nr;
            A.lc = nc;
            A.labor += c.R.c.move_labor;
          }
          land = tcell;
        }
      }
    }
    // collect on the landing tile, also on a NO-OP (move.py:112-113,136)
    uint32_t w = cells[land];
    if ((w & 0xffffu) != 0) {
      const int health[2] = {(int)AIE_CELL_STONE(w), (int)AIE_CELL_WOOD(w)};
      const double bonus = bcast(my_bonus, i);
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {
        if (health[rs] >= 1) {
          // rand() is consumed even when bonus_gather_prob == 0 (move.py:138)
          const int got = 1 + (rng_double(m, lane) < bonus ? 1 : 0);
          if (lane == i) {
            if (rs == 0) A.inv0 += got; else A.inv1 += got;
            A.labor += c.R.c.collect_labor;
          }
          w -= (1u << (8 * rs));  // consume_resource, world.py:481-483
          if (c.ev) log_event(c, AIE_EV_GATHER, i, rs, got, land / W, land % W, 0, 0, 0, 0.0);
        }
      }
      cells[land] = w;
      dirty_add_uniform(c, m, land);
    }
  }
}

__device__ __forceinline__ void gather_component_step_lookahead(const Ctx& c, MTL& m, Agents& A) {
  const int n = c.P.n, W = c.P.W, H = c.P.H, lane = c.tid;
  const int perm = rng_permutation(m, lane, n);
  uint32_t* cells = R_CELLS(c);
  const double my_bonus = R_F64(c, o_bonus_gather_prob)[lane < n ? lane : 0];
  // The reference resolves the agents one after the other, and what agent i sees -- the target tile's cell word and
  // occupancy, then the landing tile's resources -- are two dependent LDS round trips per agent.  Every lane looks
  // its OWN agent's tiles up first (one round trip for all agents), and the serial loop works on broadcasts of those
  // as long as no earlier agent of this step's order moved out of or into the agent's target tile (`stale`: then the
  // agent goes the slow way, through LDS).  Nothing else an earlier agent does is visible to a later one: resources
  // are only taken from the tile an agent stands on, and nobody else can enter that tile.
  int my_tcell = -1;     // target tile of my move, -1: none (NO-OP or out of the world)
  uint32_t my_tw = 0;    // its cell word
  int my_occ = 1;        // its occupant (0: free)
  uint32_t my_ow = 0;    // the cell word of the tile I stand on
  int my_land = 0;
  if (lane < n) {
    const int a = (int)AIE_ACT_GATHER(A.act);
    my_land = A.lr * W + A.lc;
    my_ow = cells[my_land];
    // 1 Left, 2 Right, 3 Up, 4 Down (move.py:116-123)
    const int nr = A.lr + (a == 3 ? -1 : a == 4 ? 1 : 0);
    const int nc = A.lc + (a == 1 ? -1 : a == 2 ? 1 : 0);
    if (a != 0 && nr >= 0 && nr < H && nc >= 0 && nc < W) {
      my_tcell = nr * W + nc;
      my_tw = cells[my_tcell];
      my_occ = c.locmap[my_tcell];
    }
  }
  // World.can_agent_occupy, world.py:424-440, on the looked-up values
  const int my_own = AIE_CELL_OWNER(my_tw);
  const bool my_can = my_tcell >= 0 && !(AIE_CELL_FLAGS(my_tw) & AIE_CELL_WATER) && (my_own < 0 || my_own == lane) && my_occ == 0;
  const uint32_t my_pk = (uint32_t)my_land | (my_can ? (uint32_t)my_tcell << 16 : 0xffff0000u);  // (cells fit 16 bits: uint16 lists)
  const uint32_t my_lw = my_can ? my_tw : my_ow;  // the landing tile's cell word
  uint64_t stale = 0;
  for (int k = 0; k < n; ++k) {
    const int i = bcast(perm, k);
    int land, tcell;
    uint32_t w;
    bool moves;
    if (!((stale >> i) & 1ull)) {
      const uint32_t pk = bcast(my_pk, i);
      land = (int)(pk & 0xffffu);
      tcell = (int)(pk >> 16);
      moves = tcell != 0xffff;
      w = bcast(my_lw, i);
    } else {
      const int a = (int)AIE_ACT_GATHER(bcast(A.act, i));
      const int r = bcast(A.lr, i), col = bcast(A.lc, i);
      land = r * W + col;
      tcell = 0xffff;
      moves = false;
      if (a != 0) {
        const int nr = r + (a == 3 ? -1 : a == 4 ? 1 : 0);
        const int nc = col + (a == 1 ? -1 : a == 2 ? 1 : 0);
        if (nr >= 0 && nr < H && nc >= 0 && nc < W) {
          const int t = nr * W + nc;
          const uint32_t tw = cells[t];
          const int occ = c.locmap[t];
          const int own = AIE_CELL_OWNER(tw);
          if (!(AIE_CELL_FLAGS(tw) & AIE_CELL_WATER) && (own < 0 || own == i) && occ == 0) {
            tcell = t;
            moves = true;
          }
        }
      }
      w = cells[moves ? tcell : land];
    }
    if (moves) {
      c.locmap[land] = 0;
      c.locmap[tcell] = (uint8_t)(i + 1);
      dirty_add_uniform(c, m, land);
      dirty_add_uniform(c, m, tcell);
      dirty_agent_moved(m, i);
      if (lane == i) {
        A.lr = udiv(tcell, W, c.P.mg_W);
        A.lc = tcell - A.lr * W;
        A.labor += c.R.c.move_labor;
      }
      // whoever targets the tile this agent left or the one it entered no longer knows that tile's occupancy
      stale |= __ballot(my_tcell == land || my_tcell == tcell);
      land = tcell;
    }
    // collect on the landing tile, also on a NO-OP (move.py:112-113,136)
    if ((w & 0xffffu) != 0) {
      const int health[2] = {(int)AIE_CELL_STONE(w), (int)AIE_CELL_WOOD(w)};
      const double bonus = bcast(my_bonus, i);
#pragma unroll
      for (int rs = 0; rs < 2; ++rs) {
        if (health[rs] >= 1) {
          // rand() is consumed even when bonus_gather_prob == 0 (move.py:138)
          const int got = 1 + (rng_double(m, lane) < bonus ? 1 : 0);
          if (lane == i) {
            if (rs == 0) A.inv0 += got; else A.inv1 += got;
            A.labor += c.R.c.collect_labor;
          }
          w -= (1u << (8 * rs));  // consume_resource, world.py:481-483
          if (c.ev) log_event(c, AIE_EV_GATHER, i, rs, got, land / W, land % W, 0, 0, 0, 0.0);
        }
      }
      cells[land] = w;
      dirty_add_uniform(c, m, land);
    }
  }
}

// Few agents: the plain serial loop (two dependent LDS round trips per agent).  Eight and more: every lane looks its
// own agent's tiles up first (measured: 10 agents 43.3 -> 42.3 us per launch, 4 agents 25.2 -> 26.9).
__device__ __forceinline__ void gather_component_step(const Ctx& c, MTL& m, Agents& A) {
  if (c.P.n >= 8) gather_component_step_lookahead(c, m, A);
  else gather_component_step_serial(c, m, A);
}

// ------------------------------------------------------------------------------------
// ContinuousDoubleAuction, F/components/continuous_double_auction.py
// ------------------------------------------------------------------------------------
// price_history *= 0.995 for every (commodity, agent, price) -- :451, all lanes
__device__ __forceinline__ void cda_decay_price_history(const Ctx& c) {
  double* ph = R_F64(c, o_cda_price_history);
  const int tot = 2 * c.P.n * c.P.P;
  for (int q = c.tid; q < tot; q += AIE_NT) ph[q] *= 0.995;
}

// Sort keys: bids by (price desc, lifetime desc, book position asc), asks by
// (price asc, lifetime desc, book position asc) == Python's stable sorted() at :249-256.
template <bool BIDS>
__device__ __forceinline__ bool order_before(int32_t a, int32_t b) {
  const int pa = AIE_ORD_PRICE(a), pb = AIE_ORD_PRICE(b);
  if (pa != pb) return BIDS ? pa > pb : pa < pb;
  return AIE_ORD_LIFE(a) > AIE_ORD_LIFE(b);
}

// Inserts v[q] into the sorted prefix v[0..q): one ballot per 64 slots finds the
// insertion point, one lane shift per 64 slots makes room (high slices first).
template <bool BIDS>
__device__ __forceinline__ void book_insert(int32_t* v, int q, int lane) {
  const int32_t x = v[q];
  int p = 0;
  for (int base = 0; base < q; base += AIE_NT) {
    const int idx = base + lane;
    const bool stays = idx < q && !order_before<BIDS>(x, v[idx < q ? idx : 0]);
    p += __popcll(__ballot(stays));
  }
  if (p == q) return;
  for (int base = ((q - 1) / AIE_NT) * AIE_NT; base >= 0; base -= AIE_NT) {
    const int idx = base + lane;
    const bool mv = idx >= p && idx < q;
    const int32_t t = mv ? v[idx] : 0;
    if (mv) v[idx + 1] = t;
  }
  v[p] = x;
}
// Removes slot `at` from v[0..len): lane shift towards the front, low slices first.
__device__ __forceinline__ void book_remove(int32_t* v, int len, int at, int lane) {
  for (int base = (at / AIE_NT) * AIE_NT; base < len; base += AIE_NT) {
    const int idx = base + lane;
    const bool mv = idx > at && idx < len;
    const int32_t t = mv ? v[idx] : 0;
    if (mv) v[idx - 1] = t;
  }
}
// First slot of v[0..len) whose order satisfies pred (wave ballot + find-first-set).
template <typename Pred>
__device__ __forceinline__ int book_find_first(const int32_t* v, int len, int lane, Pred pred) {
  for (int base = 0; base < len; base += AIE_NT) {
    const int idx = base + lane;
    const uint64_t hit = __ballot(idx < len && pred(v[idx < len ? idx : 0]));
    if (hit) return base + __ffsll((unsigned long long)hit) - 1;
  }
  return -1;
}

// Order-count histograms are bytes; they are updated through 32-bit LDS atomics on the containing word
// (no carry / borrow between bytes: counts stay within [0, max_num_orders]), which, unlike a byte
// read-modify-write, does not make the wave wait for LDS.
__device__ __forceinline__ void hist_add(uint8_t* hist, int idx, int lane) {  // wave-uniform idx
  if (lane == 0) atomicAdd(reinterpret_cast<uint32_t*>(hist + (idx & ~3)), 1u << (8 * (idx & 3)));
}
__device__ __forceinline__ void hist_sub_lane(uint8_t* hist, int idx) {  // one decrement per calling lane
  atomicSub(reinterpret_cast<uint32_t*>(hist + (idx & ~3)), 1u << (8 * (idx & 3)));
}

// ContinuousDoubleAuction.component_step :440-489 (decay of :451 already applied)
__device__ __forceinline__ void cda_component_step(const Ctx& c, Agents& A) {
  const int n = c.P.n, M = c.P.M, P = c.P.P, lane = c.tid;
  const int maxo = c.P.c.cda_max_num_orders, dur = c.R.c.cda_order_duration;
  uint8_t* bid_hist = R_U8(c, o_cda_bid_hist);
  uint8_t* ask_hist = R_U8(c, o_cda_ask_hist);
  int nb[2] = {uni(R_I32(c, o_cda_n_bids)[0]), uni(R_I32(c, o_cda_n_bids)[1])};
  int na[2] = {uni(R_I32(c, o_cda_n_asks)[0]), uni(R_I32(c, o_cda_n_asks)[1])};
  const int nb0[2] = {nb[0], nb[1]}, na0[2] = {na[0], na[1]};

  // ---- create_bid :168-198 / create_ask :200-229, commodity-major, agents in index order
#pragma unroll
  for (int r = 0; r < AIE_N_RES; ++r) {
    int32_t* bids = R_I32(c, o_cda_bids) + r * M;
    int32_t* asks = R_I32(c, o_cda_asks) + r * M;
    const uint32_t buy = AIE_ACT_BUY(A.act, r), sell = AIE_ACT_SELL(A.act, r);
    const uint64_t mb = __ballot(lane < n && buy > 0), ms = __ballot(lane < n && sell > 0);
    uint64_t mm = mb | ms;
    while (mm) {
      const int i = __ffsll((unsigned long long)mm) - 1;
      mm &= mm - 1;
      const int no = bcast(r ? A.no1 : A.no0, i);
      if ((mb >> i) & 1ull) {
        const int price = (int)bcast(buy, i) - 1;
        if (no < maxo && !(bcast(A.coin, i) < (double)price)) {
          bids[nb[r]] = AIE_ORD_PACK(i, price, 0);
          nb[r] += 1;
          hist_add(bid_hist, (r * n + i) * P + price, lane);
          if (lane == i) {
            if (r) A.no1 += 1; else A.no0 += 1;
            const double tr = A.coin < (double)price ? A.coin : (double)price;  // base_agent.py:279-299
            A.coin -= tr;
            A.esc_coin += tr;
            A.labor += c.R.c.cda_order_labor;
          }
        }
      }
      if ((ms >> i) & 1ull) {
        const int price = (int)bcast(sell, i) - 1;
        const int no2 = bcast(r ? A.no1 : A.no0, i);
        if (no2 < maxo && bcast(r ? A.inv1 : A.inv0, i) > 0) {
          asks[na[r]] = AIE_ORD_PACK(i, price, 0);
          na[r] += 1;
          hist_add(ask_hist, (r * n + i) * P + price, lane);
          if (lane == i) {
            if (r) { A.no1 += 1; A.inv1 -= 1; A.esc1 += 1; }
            else { A.no0 += 1; A.inv0 -= 1; A.esc0 += 1; }
            A.labor += c.R.c.cda_order_labor;
          }
        }
      }
    }
  }

  // ---- match_orders :231-350
#pragma unroll
  for (int r = 0; r < AIE_N_RES; ++r) {
    int32_t* bids = R_I32(c, o_cda_bids) + r * M;
    int32_t* asks = R_I32(c, o_cda_asks) + r * M;
    // the book is kept sorted; only this step's new orders (appended) need inserting
    // Without a new order nothing can match: last step's loop ended with every buyer's
    // best bid below the cheapest ask of another agent, and expiry only removes orders.
    if (nb[r] == nb0[r] && na[r] == na0[r]) continue;
    uint64_t possible = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    if (!c.full || M <= AIE_NT) {
      // Both books fit one wavefront: lane q keeps order q in a register.  This step's new orders
      // (appended behind the sorted part) are inserted with a ballot (position) and a one-lane shift; the
      // matching loop works on "alive" lane masks -- the best eligible bid / ask is a ballot +
      // find-first-set, a trade clears two bits; nothing in the loop's control flow waits on LDS.  The
      // survivors go back to LDS once, in order.
      int32_t bv = lane < nb[r] ? bids[lane] : 0;
      int32_t av = lane < na[r] ? asks[lane] : 0;
      for (int q = nb0[r]; q < nb[r]; ++q) {
        const int32_t x = bcast(bv, q);
        const int pb = __popcll(__ballot(lane < q && !order_before<true>(x, bv)));
        const int32_t up = (int32_t)__builtin_amdgcn_update_dpp(0, bv, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        bv = (lane > pb && lane <= q) ? up : bv;
        bv = (lane == pb) ? x : bv;
      }
      for (int q = na0[r]; q < na[r]; ++q) {
        const int32_t x = bcast(av, q);
        const int pa = __popcll(__ballot(lane < q && !order_before<false>(x, av)));
        const int32_t up = (int32_t)__builtin_amdgcn_update_dpp(0, av, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
        av = (lane > pa && lane <= q) ? up : av;
        av = (lane == pa) ? x : av;
      }
      uint64_t alive_b = nb[r] >= 64 ? ~0ull : ((1ull << nb[r]) - 1ull);
      uint64_t alive_a = na[r] >= 64 ? ~0ull : ((1ull << na[r]) - 1ull);
      if (nb[r] == 0 || na[r] == 0) possible = 0;
      const int my_buyer = AIE_ORD_AGENT(bv), my_seller = AIE_ORD_AGENT(av);
      int ntrades = 0, trade = 0;
      while (possible) {
        const uint64_t bh = __ballot((possible >> my_buyer) & 1ull) & alive_b;
        if (!bh) break;  // out of bids to check (:262-264)
        const int ib = __ffsll((unsigned long long)bh) - 1;
        const int32_t bid = bcast(bv, ib);
        const int buyer = AIE_ORD_AGENT(bid), bprice = AIE_ORD_PRICE(bid);
        const uint64_t ah = __ballot(my_seller != buyer) & alive_a;
        int ia = -1;
        int32_t ask = 0;
        if (ah) { ia = __ffsll((unsigned long long)ah) - 1; ask = bcast(av, ia); }
        if (ia < 0 || bprice < AIE_ORD_PRICE(ask)) {  // :273-286
          possible &= ~(1ull << buyer);
          continue;
        }
        // TRADE :289-346.  Only the agents' registers change inside the loop; the book-keeping in LDS
        // (order histograms, price history) and the episode accumulators are applied after it, one lane
        // per trade, through atomics (equal addends / no byte borrow: the order of arrival is immaterial).
        alive_b &= ~(1ull << ib);
        alive_a &= ~(1ull << ia);
        const int seller = AIE_ORD_AGENT(ask), aprice = AIE_ORD_PRICE(ask);
        const bool at_ask = AIE_ORD_LIFE(bid) <= AIE_ORD_LIFE(ask);  // :297-304
        const int price = at_ask ? aprice : bprice;
        if (lane == ntrades) trade = buyer | (seller << 6) | (bprice << 12) | (aprice << 20) | ((int)at_ask << 28);
        ntrades += 1;
        if (c.ev) log_event(c, AIE_EV_TRADE, r, seller, buyer, aprice, bprice, price, AIE_ORD_LIFE(ask), AIE_ORD_LIFE(bid), 0.0);
        if (lane == seller) {
          if (r) { A.no1 -= 1; A.esc1 -= 1; } else { A.no0 -= 1; A.esc0 -= 1; }
          A.coin += (double)price;
        }
        if (lane == buyer) {
          if (r) { A.no1 -= 1; A.inv1 += 1; } else { A.no0 -= 1; A.inv0 += 1; }
          A.esc_coin -= (double)bprice;
          A.coin += (double)(bprice - price);
        }
      }
      if (lane < ntrades) {
        const int buyer = trade & 63, seller = (trade >> 6) & 63, bprice = (trade >> 12) & 255, aprice = (trade >> 20) & 255;
        const int price = ((trade >> 28) & 1) ? aprice : bprice;
        const int bi = (r * n + buyer) * P + bprice, ai = (r * n + seller) * P + aprice;
        atomicSub(reinterpret_cast<uint32_t*>(bid_hist + (bi & ~3)), 1u << (8 * (bi & 3)));
        atomicSub(reinterpret_cast<uint32_t*>(ask_hist + (ai & ~3)), 1u << (8 * (ai & 3)));
        unsafeAtomicAdd(R_F64(c, o_cda_price_history) + (r * n + seller) * P + price, 1.0);
        if (C_MET(c)) {  // get_metrics :585-641: fire-and-forget integer atomics
          int32_t* tm = reinterpret_cast<int32_t*>(C_MET(c) + c.P.mo_cda);
          int32_t* sell = tm + ((0 * AIE_N_RES + r) * n + seller) * 2;
          int32_t* buy = tm + ((1 * AIE_N_RES + r) * n + buyer) * 2;
          atomicAdd(sell, 1); atomicAdd(sell + 1, price);
          atomicAdd(buy, 1); atomicAdd(buy + 1, price);
        }
      }
      // leftover orders keep their (sorted) order: slot = number of survivors in front
      AIE_WSYNC();
      if ((alive_b >> lane) & 1ull) bids[__popcll(alive_b & ((1ull << lane) - 1ull))] = bv;
      if ((alive_a >> lane) & 1ull) asks[__popcll(alive_a & ((1ull << lane) - 1ull))] = av;
      nb[r] = __popcll(alive_b);
      na[r] = __popcll(alive_a);
      AIE_WSYNC();
      continue;
    }
    for (int q = nb0[r]; q < nb[r]; ++q) book_insert<true>(bids, q, lane);
    for (int q = na0[r]; q < na[r]; ++q) book_insert<false>(asks, q, lane);
    if (nb[r] == 0 || na[r] == 0) continue;
    while (possible) {
      const uint64_t poss = possible;
      const int ib = book_find_first(bids, nb[r], lane, [poss](int32_t o) { return ((poss >> AIE_ORD_AGENT(o)) & 1ull) != 0; });
      if (ib < 0) break;  // out of bids to check (:262-264)
      const int32_t bid = uni(bids[ib]);
      const int buyer = AIE_ORD_AGENT(bid), bprice = AIE_ORD_PRICE(bid);
      const int ia = book_find_first(asks, na[r], lane, [buyer](int32_t o) { return AIE_ORD_AGENT(o) != buyer; });
      int32_t ask = 0;
      if (ia >= 0) ask = uni(asks[ia]);
      if (ia < 0 || bprice < AIE_ORD_PRICE(ask)) {  // :273-286
        possible &= ~(1ull << buyer);
        continue;
      }
      // TRADE :289-346
      book_remove(bids, nb[r], ib, lane);
      nb[r] -= 1;
      book_remove(asks, na[r], ia, lane);
      na[r] -= 1;
      const int seller = AIE_ORD_AGENT(ask), aprice = AIE_ORD_PRICE(ask);
      const int price = (AIE_ORD_LIFE(bid) <= AIE_ORD_LIFE(ask)) ? aprice : bprice;  // :297-304
      bid_hist[(r * n + buyer) * P + bprice] -= 1;
      ask_hist[(r * n + seller) * P + aprice] -= 1;
      R_F64(c, o_cda_price_history)[(r * n + seller) * P + price] += 1.0;
      if (lane == 0 && C_MET(c)) {  // get_metrics :585-641: fire-and-forget integer atomics
        int32_t* tm = reinterpret_cast<int32_t*>(C_MET(c) + c.P.mo_cda);
        int32_t* sell = tm + ((0 * AIE_N_RES + r) * n + seller) * 2;
        int32_t* buy = tm + ((1 * AIE_N_RES + r) * n + buyer) * 2;
        atomicAdd(sell, 1); atomicAdd(sell + 1, price);
        atomicAdd(buy, 1); atomicAdd(buy + 1, price);
      }
      if (c.ev) log_event(c, AIE_EV_TRADE, r, seller, buyer, aprice, bprice, price, AIE_ORD_LIFE(ask), AIE_ORD_LIFE(bid), 0.0);
      if (lane == seller) {
        if (r) { A.no1 -= 1; A.esc1 -= 1; } else { A.no0 -= 1; A.esc0 -= 1; }
        A.coin += (double)price;
      }
      if (lane == buyer) {
        if (r) { A.no1 -= 1; A.inv1 += 1; } else { A.no0 -= 1; A.inv0 += 1; }
        A.esc_coin -= (double)bprice;
        A.coin += (double)(bprice - price);
      }
    }
  }

  // ---- remove_expired_orders :352-406: lifetime += 1 everywhere, compaction by ballot
  if (!c.full || M <= AIE_NT) {
    // one lane per resting order: the four books are read back to back, survivors are scattered to their new
    // slots, expired orders decrement their histogram byte themselves (LDS atomics); only the owners'
    // registers are updated one expiry after the other, in book order (coin additions do not commute)
    int32_t ob[2][2];
#pragma unroll
    for (int r = 0; r < AIE_N_RES; ++r) {
      ob[r][0] = lane < nb[r] ? (R_I32(c, o_cda_bids) + r * M)[lane] : 0;
      ob[r][1] = lane < na[r] ? (R_I32(c, o_cda_asks) + r * M)[lane] : 0;
    }
    AIE_WSYNC();
#pragma unroll
    for (int r = 0; r < AIE_N_RES; ++r) {
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        int32_t* v = (side == 0 ? R_I32(c, o_cda_bids) : R_I32(c, o_cda_asks)) + r * M;
        const int len = side == 0 ? nb[r] : na[r];
        const int32_t o = ob[r][side];
        const bool valid = lane < len;
        const int life = AIE_ORD_LIFE(o) + 1;
        const bool keep = valid && life <= dur;
        const uint64_t km = __ballot(keep);
        uint64_t em = __ballot(valid && !keep);
        if (keep) v[__popcll(km & lanemask_lt(lane))] = AIE_ORD_PACK(AIE_ORD_AGENT(o), AIE_ORD_PRICE(o), life);
        if (valid && !keep) hist_sub_lane(side == 0 ? bid_hist : ask_hist, (r * n + AIE_ORD_AGENT(o)) * P + AIE_ORD_PRICE(o));
        while (em) {
          const int q = __ffsll((unsigned long long)em) - 1;
          em &= em - 1;
          const int32_t eo = bcast(o, q);
          const int ag = AIE_ORD_AGENT(eo), pr = AIE_ORD_PRICE(eo);
          if (lane == ag) {
            if (side == 0) {
              const double tr = A.esc_coin < (double)pr ? A.esc_coin : (double)pr;  // escrow_to_inventory
              A.esc_coin -= tr;
              A.coin += tr;
              if (r) A.no1 -= 1; else A.no0 -= 1;
            } else {
              if (r) { A.esc1 -= 1; A.inv1 += 1; A.no1 -= 1; }
              else { A.esc0 -= 1; A.inv0 += 1; A.no0 -= 1; }
            }
          }
        }
        if (side == 0) nb[r] = __popcll(km); else na[r] = __popcll(km);
      }
    }
  } else {
#pragma unroll
  for (int r = 0; r < AIE_N_RES; ++r) {
#pragma unroll
    for (int side = 0; side < 2; ++side) {
      int32_t* v = (side == 0 ? R_I32(c, o_cda_bids) : R_I32(c, o_cda_asks)) + r * M;
      const int len = side == 0 ? nb[r] : na[r];
      int kept = 0;
      for (int base = 0; base < len; base += AIE_NT) {
        const int idx = base + lane;
        const bool valid = idx < len;
        const int32_t o = valid ? v[idx] : 0;
        const int life = AIE_ORD_LIFE(o) + 1;
        const bool keep = valid && life <= dur;
        const uint64_t km = __ballot(keep);
        uint64_t em = __ballot(valid && !keep);
        if (keep) v[kept + __popcll(km & lanemask_lt(lane))] = AIE_ORD_PACK(AIE_ORD_AGENT(o), AIE_ORD_PRICE(o), life);
        kept += __popcll(km);
        while (em) {  // expiries are rare: handled one by one, in book order
          const int q = __ffsll((unsigned long long)em) - 1;
          em &= em - 1;
          const int32_t eo = bcast(o, q);
          const int ag = AIE_ORD_AGENT(eo), pr = AIE_ORD_PRICE(eo);
          if (side == 0) {
            bid_hist[(r * n + ag) * P + pr] -= 1;
            if (lane == ag) {
              const double tr = A.esc_coin < (double)pr ? A.esc_coin : (double)pr;  // escrow_to_inventory
              A.esc_coin -= tr;
              A.coin += tr;
              if (r) A.no1 -= 1; else A.no0 -= 1;
            }
          } else {
            ask_hist[(r * n + ag) * P + pr] -= 1;
            if (lane == ag) {
              if (r) { A.esc1 -= 1; A.inv1 += 1; A.no1 -= 1; }
              else { A.esc0 -= 1; A.inv0 += 1; A.no0 -= 1; }
            }
          }
        }
      }
      if (side == 0) nb[r] = kept; else na[r] = kept;
    }
  }
  }
  R_I32(c, o_cda_n_bids)[0] = nb[0];
  R_I32(c, o_cda_n_bids)[1] = nb[1];
  R_I32(c, o_cda_n_asks)[0] = na[0];
  R_I32(c, o_cda_n_asks)[1] = na[1];
}

// ------------------------------------------------------------------------------------
// PeriodicBracketTax, F/components/redistribution.py
// ------------------------------------------------------------------------------------
// Coin arithmetic of the tax / redistribution components without multiply-add contraction: the
// reference (and the restatement, built with -ffp-contract=off) round every product, and a one-ulp
// difference in an agent's coin flips discrete decisions later on (income residues of +-1e-15 decide
// `income < 0` in marginal_rate and `z_t > 0` in the Saez sample filter).
#pragma clang fp contract(off)
// curr_rate_max :390-394: the annealed limit follows _last_completions, which generate_masks
// refreshes AFTER the observations of a reset are built (:1036-1046) -- kept as a state field.
__device__ __forceinline__ double tax_curr_rate_max(const Ctx& c) {
  return aie_annealed_tax_limit(*R_I32(c, o_tax_last_completions), c.R.c.tax_annealing_warmup,
                                c.R.c.tax_annealing_slope, c.R.c.tax_rate_max);
}
// this replica's Saez block (tax_model "saez", aie_layout.h: a_saez); step / reset kernels only (c.met set)
__device__ __forceinline__ uint8_t* saez_block(const Ctx& c) {
  return C_MET(c) - c.R.a_metrics - (int64_t)c.e * c.P.met_bytes + c.R.a_saez + (int64_t)c.e * c.P.saez_stride;
}
__device__ __forceinline__ double tax_rate(const Ctx& c, int b) {  // curr_marginal_rates :396-417
  if (c.P.c.tax_model == AIE_TAX_MODEL_WRAPPER) return c.rtab[R_I32(c, o_tax_rate_idx)[b]];
  if (c.saez) {  // np.minimum(curr_bracket_tax_rates, curr_rate_max) :406-409
    const double r = R_F64(c, o_tax_saez_rates)[b];
    const double cap = c.P.c.tax_annealing ? tax_curr_rate_max(c) : c.R.c.tax_rate_max;
    return r < cap ? r : cap;
  }
  const double r = c.R.c.tax_fixed_rates[b];
  if (!c.P.c.tax_annealing) return r;
  const double cap = tax_curr_rate_max(c);
  return r < cap ? r : cap;
}
// _curr_rates_obs: the rates the "curr_rates" observation shows (cached at period starts and resets)
__device__ __forceinline__ double tax_rate_obs(const Ctx& c, int b) {
  if (c.saez) return R_F64(c, o_tax_saez_obs_rates)[b];
  return tax_rate(c, b);
}
// planner tax-rate action j of a bracket: allowed by the annealing schedule? (annealed_tax_mask,
// utils.py:59-118, evaluated with the completions count of the episode's reset)
__device__ __forceinline__ bool tax_rate_action_visible(const Ctx& c, int j) {
  if (!c.P.c.tax_annealing) return true;
  double full = 0;
  for (int k = 0; k < c.P.c.tax_n_disc_rates; ++k) full = fmax(full, fabs(c.rtab[k]));
  const double vis = aie_annealed_tax_limit(*R_I32(c, o_tax_last_completions), c.R.c.tax_annealing_warmup,
                                            c.R.c.tax_annealing_slope, full);
  return fabs(c.rtab[j]) <= vis;
}
__device__ __forceinline__ double tax_marginal_rate(const Ctx& c, double income) {  // marginal_rate :837-844
  if (income < 0) return 0.0;
  const int NB = c.P.NB;
  for (int b = 0; b < NB; ++b) {
    const double lo = c.R.c.tax_bracket_cutoffs[b];
    const bool under = (b + 1 < NB) ? (income < c.R.c.tax_bracket_cutoffs[b + 1]) : (income < __builtin_huge_val());
    if (income >= lo && under) return tax_rate(c, b);
  }
  return tax_rate(c, 0);
}
__device__ __forceinline__ double tax_bin(const Ctx& c, double income, int b) {
  const int NB = c.P.NB;
  const double cut = c.R.c.tax_bracket_cutoffs[b];
  const double size = (b + 1 < NB) ? c.R.c.tax_bracket_cutoffs[b + 1] - cut : __builtin_huge_val();
  double past = income - cut;
  if (past < 0) past = 0;
  return tax_rate(c, b) * (size < past ? size : past);
}
__device__ __forceinline__ double tax_due(const Ctx& c, double income) {  // taxes_due :846-851
  const int NB = c.P.NB;
  // np.sum of NB values (pairwise summation, loops_utils.h.src)
  if (NB < 8) {
    double res = -0.0;
    for (int b = 0; b < NB; ++b) res += tax_bin(c, income, b);
    return res;
  }
  double r0 = tax_bin(c, income, 0), r1 = tax_bin(c, income, 1), r2 = tax_bin(c, income, 2),
         r3 = tax_bin(c, income, 3), r4 = tax_bin(c, income, 4), r5 = tax_bin(c, income, 5),
         r6 = tax_bin(c, income, 6), r7 = tax_bin(c, income, 7);
  int b = 8;
  for (; b < NB - (NB % 8); b += 8) {
    r0 += tax_bin(c, income, b + 0); r1 += tax_bin(c, income, b + 1);
    r2 += tax_bin(c, income, b + 2); r3 += tax_bin(c, income, b + 3);
    r4 += tax_bin(c, income, b + 4); r5 += tax_bin(c, income, b + 5);
    r6 += tax_bin(c, income, b + 6); r7 += tax_bin(c, income, b + 7);
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; b < NB; ++b) res += tax_bin(c, income, b);
  return res;
}
// enact_taxes :853-915: lane i computes agent i's income / tax; the revenue is summed
// in agent order (the reference's running `net_tax_revenue +=`).
__device__ __forceinline__ void tax_enact(const Ctx& c, Agents& A) {
  const int n = c.P.n, i = c.tid;
  double eff = 0, eff_rate = 0;
  if (i < n) {
    const double income = (A.coin + A.esc_coin) - R_F64(c, o_tax_last_coin)[i];
    const double due = tax_due(c, income);
    eff = A.coin < due ? A.coin : due;  // escrow is not taxed
    R_F64(c, o_tax_last_marginal_rate)[i] = tax_marginal_rate(c, income);
    R_F64(c, o_tax_last_income)[i] = income;
    A.coin -= eff;
    eff_rate = eff / (income > 0.000001 ? income : 0.000001);  // :880
    if (C_MET(c)) {  // episode accumulators for get_metrics :1141-1186 (no-return atomics)
      unsafeAtomicAdd(reinterpret_cast<double*>(C_MET(c) + c.P.mo_tax_income) + i, income > 0 ? income : 0.0);
      unsafeAtomicAdd(reinterpret_cast<double*>(C_MET(c) + c.P.mo_tax_paid) + i, eff);
      int bin = 0;  // income_bin :828-835
      if (income >= 0)
        for (int b = 0; b < c.P.NB; ++b)
          if (income >= c.R.c.tax_bracket_cutoffs[b] && (b + 1 == c.P.NB || income < c.R.c.tax_bracket_cutoffs[b + 1])) { bin = b; break; }
      atomicAdd(reinterpret_cast<int32_t*>(C_MET(c) + c.P.mo_tax_occ) + bin, 1);
    }
  }
  if (C_MET(c)) {
    if (i < c.P.NB) unsafeAtomicAdd(reinterpret_cast<double*>(C_MET(c) + c.P.mo_tax_sched) + i, tax_rate(c, i));
    double day = 0;
    for (int j = 0; j < n; ++j) day += bcast(eff_rate, j);
    if (i == 0) {
      unsafeAtomicAdd(reinterpret_cast<double*>(C_MET(c) + c.P.mo_tax_eff), day);
      atomicAdd(reinterpret_cast<int32_t*>(C_MET(c) + c.P.mo_tax_days), 1);
    }
  }
  if (c.ev) {
    for (int b = 0; b < c.P.NB; ++b) log_event(c, AIE_EV_TAX_BRACKET, b, 0, 0, 0, 0, 0, 0, 0, tax_rate(c, b));
    for (int j = 0; j < n; ++j) log_event(c, AIE_EV_TAX, j, 0, 0, 0, 0, 0, 0, 0, bcast(eff, j));
  }
  double net = 0;
  for (int j = 0; j < n; ++j) net += bcast(eff, j);
  *R_F64(c, o_tax_total_collected) += net;
  const double lump = net / (double)n;
  if (i < n) {
    A.coin += lump;
    R_F64(c, o_tax_last_coin)[i] = A.coin + A.esc_coin;
  }
  if (c.saez) {  // _update_saez_buffer :533-541 (global memory, tax days only)
    uint8_t* blk = saez_block(c);
    int32_t* hdr = reinterpret_cast<int32_t*>(blk);
    double* buf = reinterpret_cast<double*>(blk + AIE_SAEZ_OFF_BUF);
    int len = uni(hdr[0]);
    if (i < n) {
      buf[2 * (len + i)] = R_F64(c, o_tax_last_income)[i];
      buf[2 * (len + i) + 1] = R_F64(c, o_tax_last_marginal_rate)[i];
    }
    len += n;
    const int size = c.R.c.saez_buffer_size;
    if (len > size) {  // drop the oldest: move down chunk by chunk (a chunk's loads precede its stores;
      const int shift = 2 * (len - size);  // later chunks only read above what earlier ones wrote)
      __builtin_amdgcn_s_waitcnt(0);
      for (int base = 0; base < 2 * size; base += AIE_NT) {
        const int q = base + i;
        double v = 0;
        if (q < 2 * size) v = buf[q + shift];
        __builtin_amdgcn_s_waitcnt(0);
        if (q < 2 * size) buf[q] = v;
      }
      len = size;
    }
    if (i == 0) {
      hdr[0] = len;
      hdr[2] += n;  // _additions_this_episode :541 (only reset_saez_buffers zeroes it)
    }
  }
}
// component_step :945-972 + set_new_period_rates_model :419-434
__device__ __forceinline__ void tax_component_step(const Ctx& c, MTL& ml, Agents& A) {
  int pos = uni(*R_I32(c, o_tax_cycle_pos));
  if (pos == 1 && c.saez) {  // compute_and_set_new_period_rates_from_saez_formula
    uint8_t* blk = saez_block(c);
    if (uni(reinterpret_cast<const int32_t*>(blk)[1])) {  // the formula ran in aie_saez_kernel just before this launch
      if (c.tid < c.P.NB) R_F64(c, o_tax_saez_rates)[c.tid] = reinterpret_cast<const double*>(blk + AIE_SAEZ_OFF_NEXT)[c.tid];
    } else {  // np.random.uniform(low=rate_min, high=curr_rate_max, size=n_brackets) :451-457
      const double lo = c.R.c.tax_rate_min;
      const double hi = c.P.c.tax_annealing ? tax_curr_rate_max(c) : c.R.c.tax_rate_max;
      for (int b = 0; b < c.P.NB; ++b) {
        const double r = lo + (hi - lo) * rng_double(ml, c.tid);
        if (c.tid == b) R_F64(c, o_tax_saez_rates)[b] = r;
      }
    }
    AIE_WSYNC();
    if (c.tid < c.P.NB) R_F64(c, o_tax_saez_obs_rates)[c.tid] = tax_rate(c, c.tid);  // _curr_rates_obs :959
    AIE_WSYNC();
  }
  if (pos == 1 && c.P.c.tax_model == AIE_TAX_MODEL_WRAPPER && !c.P.c.tax_disable) {
    if (c.tid < c.P.NB) {
      const int a = c.act_p[c.tid];
      if (a > 0 && a <= c.P.c.tax_n_disc_rates) R_I32(c, o_tax_rate_idx)[c.tid] = a - 1;
    }
    AIE_WSYNC();
  }
  if (pos >= c.R.c.tax_period) {
    tax_enact(c, A);
    pos = 0;
  }
  *R_I32(c, o_tax_cycle_pos) = pos + 1;
}

// WealthRedistribution.component_step, F/components/redistribution.py:46-65: inventory coin
// := np.sum(inventory + escrow) / n - escrow.  The sum follows NumPy's pairwise order
// (np_sum_small) with the lanes' values read through wave broadcasts.
__device__ __forceinline__ void wealth_component_step(const Ctx& c, Agents& A) {
  const int n = c.P.n;
  const double v = A.coin + A.esc_coin;
  double tot;
  if (n < 8) {
    tot = -0.0;
    for (int j = 0; j < n; ++j) tot += bcast(v, j);
  } else {
    double r[8];
    for (int q = 0; q < 8; ++q) r[q] = bcast(v, q);
    int j = 8;
    for (; j < n - (n % 8); j += 8)
      for (int q = 0; q < 8; ++q) r[q] += bcast(v, j + q);
    tot = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; j < n; ++j) tot += bcast(v, j);
  }
  if (c.tid < n) A.coin = tot / (double)n - A.esc_coin;
}

// (contraction stays off, see the top of the file)

// ------------------------------------------------------------------------------------
// LayoutFromFile.scenario_step, layout_from_file.py:372-410.
// regen_halfwidth == 0: p = regen_weight * max(map, src); only source blocks spawn.
// regen_halfwidth > 0 (dynamic_layout.py:446-463): p from the per-episode window counts.
// np.random.rand(H, W) is consumed for Wood, then for Stone: 4*H*W MT19937 words per
// step, read row by row straight from the state registers (pairs of adjacent lanes
// form one 53-bit double); a double that straddles a twist is carried over.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void regen_cell(const Ctx& c, uint32_t ta, uint32_t tb, int d) {
  const int HW = c.P.HW;
  const int rs = d >= HW ? 0 : 1;  // first H*W doubles are Wood's, then Stone's
  const int cell = d - (d >= HW ? HW : 0);
  uint8_t* cb = reinterpret_cast<uint8_t*>(R_CELLS(c)) + 4 * cell;
  const uint32_t fl = cb[3];
  if (fl & (rs ? AIE_CELL_WOOD_SRC : AIE_CELL_STONE_SRC)) {
    const uint32_t mval = cb[rs];
    const uint32_t health = mval > 1u ? mval : 1u;  // max(map, source block = 1)
    const double u = u53(ta, tb);
    // halfwidth 0: p = regen_weight * health; else regen_p[source blocks in the window] (aie_layout.h: regen_conv)
    double p;
    const int hwid = c.P.regen_conv ? c.P.c.regen_halfwidth[rs] : 0;
    if (c.full && c.snap && hwid > 0 && c.P.c.max_health[rs] > 1) {
      // signal.convolve2d(max(map, source blocks), regen_weight / d^2 * ones(d, d), "same") at this cell, one
      // multiply-add per kernel element in scipy's order (input rows r0+hw .. r0-hw, columns c0+hw .. c0-hw, zeros
      // outside the world), over the pre-step snapshot (dynamic_layout.py:446-463)
      const int dd = 1 + 2 * hwid, W = c.P.W, r0 = cell / W, c0 = cell - r0 * W;
      const double kern = c.R.c.regen_weight[rs] / (double)(dd * dd);
      const uint8_t* sp = c.snap + rs * ((HW + 15) / 16 * 16);
      p = 0.0;
      for (int r = r0 + hwid; r >= r0 - hwid; --r)
        for (int cc = c0 + hwid; cc >= c0 - hwid; --cc)
          if (r >= 0 && r < c.P.H && cc >= 0 && cc < W) p += kern * (double)sp[r * W + cc];
    } else if (hwid > 0) {
      p = c.R.regen_p[rs][R_U8(c, o_regen_count)[rs * HW + cell]];
    } else {
      p = c.R.c.regen_weight[rs] * (double)health;
    }
    if (u < p && mval < (uint32_t)c.P.c.max_health[rs]) {
      cb[rs] = (uint8_t)(mval + 1);
      dirty_add_lane(c, cell);
    }
  }
}
__device__ __forceinline__ void scenario_step_regen_rows(const Ctx& c, MT& m) {
  const int lane = c.tid;
  const int total = 4 * c.P.HW;  // words to consume
  int done = 0;
  bool carry = false;
  uint32_t carry_t = 0;
  int carry_d = 0;
  while (true) {
    if (m.pos >= AIE_MT_N) {
      mt_twist(m, lane);
      m.pos = 0;
    }
    const int pos = m.pos;
    if (carry) {  // second half of a double whose first word was word 623 of the old state
      if (lane == 0) regen_cell(c, carry_t, mt_word(m, m.r[0]), carry_d);
      carry = false;
    }
#pragma unroll
    for (int J = 0; J < 10; ++J) {
      if (64 * J + 63 < pos) continue;              // row fully consumed already
      if (done + (64 * J - pos) >= total) continue;  // row entirely beyond this step's needs
      const int a_idx = 64 * J + lane;
      const int q = done + (a_idx - pos);           // regen-relative word number
      const uint32_t t = mt_word(m, m.r[J]);
      uint32_t tn = lane_get(t, (lane + 1) & 63);
      if (J < 9) tn = (lane == 63) ? mt_word(m, bcast(m.r[J + 1], 0)) : tn;
      const bool first = a_idx >= pos && a_idx < AIE_MT_N - 1 && (q & 1) == 0 && q < total;
      if (first) regen_cell(c, t, tn, q >> 1);
    }
    const int avail = AIE_MT_N - pos;
    const int take = (total - done) < avail ? (total - done) : avail;
    if (take == avail) {  // consumed word 623: is it the first half of a double?
      const int q623 = done + (AIE_MT_N - 1 - pos);
      if ((q623 & 1) == 0 && q623 < total) {
        carry = true;
        carry_t = mt_word(m, bcast(m.r[9], 47));
        carry_d = q623 >> 1;
      }
    }
    done += take;
    m.pos = pos + take;
    if (done >= total) break;
  }
}

// Sparse variant (the normal case: a few dozen source blocks): lane j owns source double d_j and needs stream words
// pos+2*d_j, +1.  The state is advanced window by window (one twist each) in registers; in every window each lane
// fetches the word(s) that fall into it straight from the row registers: one permute per row, the lane keeps the
// one of its own row (word i of a window sits in row i >> 6, lane i & 63).  No LDS copy of the window.
__device__ __forceinline__ uint32_t mt_window_word(const MT& m, int idx) {
  const int row = idx >> 6, ln = idx & 63;
  uint32_t v = 0;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t t = lane_get(m.r[r], ln);
    v = (row == r) ? t : v;
  }
  return v;
}
// The source doubles of this replica: the LDS list load_record collected, or -- fixed layouts shared by the batch -- the
// caller's registers (entry k * 64 + lane of aie_params.a_src_list, fetched while the components ran) and their count.
struct SrcList {
  int S;                            // number of source doubles (may exceed AIE_SRC_CAP: row-by-row regeneration)
  int d[AIE_SRC_CAP / AIE_NT];      // this lane's entries
};
__device__ __forceinline__ SrcList src_list_from_record(const Ctx& c, const uint8_t* __restrict__ arena) {
  SrcList L;
  const uint8_t* g = arena + (int64_t)c.e * c.P.rec_bytes;
  L.S = uni(*reinterpret_cast<const int32_t*>(g + c.P.o_src_n));
  const uint16_t* lst = reinterpret_cast<const uint16_t*>(g + c.P.o_src_list);
#pragma unroll
  for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k) L.d[k] = (int)lst[k * AIE_NT + c.tid];  // (entries past the count are zero)
  return L;
}
__device__ __forceinline__ SrcList src_list_from_arena(const Ctx& c, const uint8_t* __restrict__ arena) {
  SrcList L;
  const uint8_t* g = arena + c.R.a_src_list;
  L.S = uni(*reinterpret_cast<const int32_t*>(g));
  const uint16_t* lst = reinterpret_cast<const uint16_t*>(g + 16);
#pragma unroll
  for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k) L.d[k] = (int)lst[k * AIE_NT + c.tid];  // (entries past the count are zero)
  return L;
}
__device__ __forceinline__ void scenario_step_regen(const Ctx& c, MT& m, const SrcList& src) {
  if (c.full && c.snap) {  // P.regen_general: the planes the reference convolves, before any of this step's respawns
    const int HW = c.P.HW, stride = (HW + 15) / 16 * 16;
    const uint8_t* cb = reinterpret_cast<const uint8_t*>(R_CELLS(c));
    for (int q = c.tid; q < HW; q += AIE_NT) {
      const uint32_t fl = cb[4 * q + 3];
      const uint8_t s0 = cb[4 * q], s1 = cb[4 * q + 1];
      c.snap[q] = (fl & AIE_CELL_STONE_SRC) ? (s0 > 1 ? s0 : 1) : s0;
      c.snap[stride + q] = (fl & AIE_CELL_WOOD_SRC) ? (s1 > 1 ? s1 : 1) : s1;
    }
    AIE_WSYNC();
  }
  const int S = src.S;
  if (S > AIE_SRC_CAP) {
    if (m.fast && m.pos < AIE_MT_N) mt_fast_rows(m, c.tid);  // (the counter stream keeps no rows between steps)
    scenario_step_regen_rows(c, m);
    return;
  }
  const int lane = c.tid;
  const int total = 4 * c.P.HW;
  const int pos0 = m.pos;  // <= 624
  const int nchunk = (S + AIE_NT - 1) / AIE_NT;
  if (m.fast) {
    // AIE_RNG_FAST: source double d is words pos0 + 2 d, + 1 of the linear stream -- computed where they are needed.  One
    // Philox block per lane when pos0 is even (both words in one pair), two when it is odd (wave-uniform).
    const int odd = pos0 & 1;
#pragma unroll
    for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k) {
      if (k >= nchunk) continue;
      const int j = k * AIE_NT + lane;
      const int d = j < S ? src.d[k] : 0;
      const int h = (pos0 >> 1) + d;  // pair that holds word pos0 + 2 d
      uint32_t a0, a1, b0 = 0, b1 = 0;
      fast_pair(m.fkey, m.fblk, m.fsalt, h, a0, a1);
      if (odd) fast_pair(m.fkey, m.fblk, m.fsalt, h + 1, b0, b1);
      if (j < S) regen_cell(c, odd ? a1 : a0, odd ? b0 : a1, d);
    }
    const int last_win = (pos0 + total - 1) / AIE_MT_N;
    m.pos = pos0 + total - last_win * AIE_MT_N;
    m.fblk += (uint32_t)last_win;
    AIE_WSYNC();
    return;
  }
  int off_a[AIE_SRC_CAP / AIE_NT];
  uint32_t wa[AIE_SRC_CAP / AIE_NT], wb[AIE_SRC_CAP / AIE_NT];
#pragma unroll
  for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k) {
    const int j = k * AIE_NT + lane;
    off_a[k] = (k < nchunk && j < S) ? pos0 + 2 * src.d[k] : -16;  // stream offset of word A
    wa[k] = wb[k] = 0;
  }
  const int last_win = (pos0 + total - 1) / AIE_MT_N;
  for (int w = 0; w <= last_win; ++w) {
    if (w > 0) mt_twist_body(m, lane);  // the hot site: inlined (4 twists per step)
    const int lo = w * AIE_MT_N;
    if (c.mtwin) {
      // the window through LDS (mt_window_in_lds): ten row stores, then one two-word read per lane and chunk.  LDS
      // operations of a wave execute in order, so the reads see the stores, and the next window's stores the reads.
#pragma unroll
      for (int j = 0; j < 10; ++j) c.mtwin[64 * j + lane] = m.r[j];
#pragma unroll
      for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k) {
        if (k >= nchunk) continue;
        const int ia = off_a[k] - lo;  // word A's index in this window (-1: A was the previous window's last word)
        const bool ha = (unsigned)ia < (unsigned)AIE_MT_N, hb = (unsigned)(ia + 1) < (unsigned)AIE_MT_N;
        if (__ballot(ha || hb) == 0) continue;
        const int at = ia < 0 ? 0 : (ia > AIE_MT_N - 1 ? AIE_MT_N - 1 : ia);
        const uint32_t x0 = c.mtwin[at], x1 = c.mtwin[at + 1];  // (word 624 is padding: read, never used)
        if (ha) wa[k] = x0;
        if (hb) wb[k] = ia < 0 ? x0 : x1;
      }
      continue;
    }
#pragma unroll
    for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k) {
      if (k >= nchunk) continue;  // (uniform: the usual map lists < 64 draws)
      const int ia = off_a[k] - lo, ib = ia + 1;
      const bool ha = ia >= 0 && ia < AIE_MT_N, hb = off_a[k] >= 0 && ib >= 0 && ib < AIE_MT_N;
      if (__ballot(ha || hb) == 0) continue;
      const uint32_t va = mt_window_word(m, ha ? ia : 0), vb = mt_window_word(m, hb ? ib : 0);
      if (ha) wa[k] = va;
      if (hb) wb[k] = vb;
    }
  }
  m.pos = pos0 + total - last_win * AIE_MT_N;
#pragma unroll
  for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k)
    if (k < nchunk && off_a[k] >= 0) regen_cell(c, mt_temper(wa[k]), mt_temper(wb[k]), (off_a[k] - pos0) >> 1);
  AIE_WSYNC();
}

// ------------------------------------------------------------------------------------
// Utilities / rewards
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double energy_weight(const Ctx& c) {  // layout_from_file.py:249-267
  if (!c.P.sh_energy_warmup) return 1.0;  // (energy_warmup_constant <= 0: a property of the instance's family)
  const int v = c.P.c.energy_warmup_method == AIE_WARMUP_DECAY ? *R_I32(c, o_completions) : *R_I32(c, o_auto_warmup);
  return 1.0 - aie_exp_glibc(-(double)v / c.R.c.energy_warmup_constant);  // libm's exp, bit for bit (aie_glibc_math.h)
}

// get_current_optimization_metrics layout_from_file.py:269-318 with
// rewards.isoelastic_coin_minus_labor (F/scenarios/utils/rewards.py:12-48),
// coin_eq_times_productivity (:84-101), inv_income_weighted_* (:104-133),
// social_metrics.get_gini (social_metrics.py:10-46).
// Lane i < n computes agent i's utility; lane 0 finishes the planner's.
// Results are left in scr_part()[0..n]; must be followed by AIE_WSYNC().
__device__ __forceinline__ void current_metrics(const Ctx& c) {
  const int n = c.P.n, i = c.tid;
  double* coin = scr_coin(c);
  double* out = scr_part(c);
  double* tmp = scr_gini_sort(c);  // (its own slot: only the other planner reward type sorts there, and the flat-vector
                                   // writer -- which may run on the other wave meanwhile -- ranks incomes in scr_cmr)
  const double lcf = energy_weight(c) * c.R.c.energy_cost;
  const double eta = c.R.c.isoelastic_eta;
  if (i < n) {
    const double ci = R_F64(c, o_inv_coin)[i] + R_F64(c, o_esc_coin)[i];
    coin[i] = ci;
    double util_c;
    if (c.P.sh_eta_is_one) util_c = aie_log_glibc(ci > 1 ? ci : 1);
    else util_c = (aie_pow_glibc(ci, 1 - eta) - 1) / (1 - eta);  // libm's pow bit for bit: the sign of a ~1e-16 mean reward
                                                                  // feeds the integer auto_warmup counter (aie_glibc_math.h)
    out[i] = util_c - R_F64(c, o_labor)[i] * lcf;
  }
  AIE_WSYNC();
  const int prt = c.P.c.planner_reward_type;
  // every sum below follows NumPy's pairwise add.reduce order (np_sum_small / np_sum_seq): the planner's utility is
  // state (`util`), compared with the reference bit for bit
  if (prt == AIE_PLANNER_REW_COIN_EQ_TIMES_PROD) {
    double gini;
    const double tot = np_sum_small(coin, n);
    if (n < 30) {
      const double diff = np_sum_seq<3>(coin, n, 1, 65536u / (uint32_t)n + 1u, 0, n * n, i);
      const double unscaled = diff / (2 * n * tot + 1e-10);
      gini = unscaled / ((double)(n - 1) / (double)n);
    } else {
      // sorted-cumsum branch (social_metrics.py:43-46)
      double* s = scr_gini_sort(c);
      if (i == 0) {
        for (int j = 0; j < n; ++j) s[j] = coin[j];
        for (int a = 1; a < n; ++a) {
          double x = s[a]; int b = a - 1;
          while (b >= 0 && s[b] > x) { s[b + 1] = s[b]; --b; }
          s[b + 1] = x;
        }
        const double tots = np_sum_small(s, n) + 1e-10;
        double run = 0;
        for (int j = 0; j < n; ++j) { run += s[j]; s[j] = run / tots; }
      }
      AIE_WSYNC();
      gini = 1 - (2.0 / (n + 1)) * np_sum_small(s, n);
    }
    if (i == 0) {
      const double ew = 1 - c.R.c.mixing_weight_gini_vs_coin;
      out[n] = (ew * (1 - gini) + (1 - ew)) * (tot / n);
    }
  } else {
    if (i < n) tmp[i] = 1 / (coin[i] > 1 ? coin[i] : 1);
    AIE_WSYNC();
    const double sw = np_sum_small(tmp, n);
    AIE_WSYNC();
    if (i < n) tmp[i] = (prt == AIE_PLANNER_REW_INV_INCOME_COIN ? coin[i] : out[i]) * (tmp[i] / sw);
    AIE_WSYNC();
    if (i == 0) out[n] = np_sum_small(tmp, n);
  }
}

// compute_reward layout_from_file.py:519-559
__device__ __forceinline__ void compute_rewards(const Ctx& c, uint8_t* __restrict__ arena, float* rew_log = nullptr) {
  const int n = c.P.n, i = c.tid;
  current_metrics(c);
  AIE_WSYNC();
  double* cur = scr_part(c);
  double* util = R_F64(c, o_util);
  double* rew = scr_coin(c);  // reuse
  if (i <= n) {
    const double r = cur[i] - util[i];
    util[i] = cur[i];
    rew[i] = r;
    if (i < n) reinterpret_cast<float*>(arena + c.R.a_rew_a)[(int64_t)c.e * n + i] = (float)r;
    else reinterpret_cast<float*>(arena + c.R.a_rew_p)[c.e] = (float)r;
    if (rew_log) rew_log[(int64_t)c.e * (n + 2) + i] = (float)r;
  }
  AIE_WSYNC();
  if (i == 0) {
    if (np_sum_small(rew, n) / n > 0) *R_I32(c, o_auto_warmup) += 1;
  }
}

// ------------------------------------------------------------------------------------
// Observations
// ------------------------------------------------------------------------------------
// Channels of Maps.state (world.py:59-93): Stone, Wood, House, [Water], StoneSourceBlock,
// WoodSourceBlock.  `ch[]` receives CM values for one packed cell word.
template <bool WATER>
__device__ __forceinline__ void cell_channels(uint32_t w, float* ch) {
  const uint32_t fl = w >> 24;
  ch[0] = (float)AIE_CELL_STONE(w);
  ch[1] = (float)AIE_CELL_WOOD(w);
  ch[2] = (((w >> 16) & 0xffu) != 0xffu) ? 1.0f : 0.0f;
  int k = 3;
  if (WATER) ch[k++] = (fl & AIE_CELL_WATER) ? 1.0f : 0.0f;
  ch[k++] = (fl & AIE_CELL_STONE_SRC) ? 1.0f : 0.0f;
  ch[k++] = (fl & AIE_CELL_WOOD_SRC) ? 1.0f : 0.0f;
}

// LayoutFromFile.generate_observations layout_from_file.py:412-517: the egocentric
// (2w+1)^2 crop for every agent and the full map for the planner.
//   crop:    one lane per (agent, window cell): one LDS word read feeds CM+1 coalesced f32
//            stores and 2 i16 stores (the out-of-world ring: all 0, in-bounds channel 0);
//   planner: one lane per 4 consecutive cells: one 16-byte LDS read feeds CM 16-byte
//            stores and two 8-byte stores.
// All stores go through buffer descriptors of this replica's observation blocks: one
// 32-bit lane offset per item + a scalar offset per (agent, channel) plane, instead of a
// 64-bit VGPR pointer per plane.
struct SpatialOut {
  BufRsrc amap, aidx, pmap, pidx;
};
template <bool WATER>
__device__ __forceinline__ SpatialOut spatial_out(const Ctx& c, uint8_t* __restrict__ arena) {
  constexpr int CM = WATER ? 6 : 5;
  const int n = c.P.n, HW = c.P.HW, WV2 = c.P.WV * c.P.WV;
  (void)WV2;
  const int plane = c.P.am_h * c.P.am_w;
  const uint32_t amap_bytes = (uint32_t)(n * c.P.am_ch * plane * 4), aidx_bytes = (uint32_t)(n * 2 * plane * 2);
  SpatialOut o;
  o.amap = make_rsrc(arena + c.R.a_obs_a_map + (int64_t)c.e * amap_bytes, amap_bytes);
  o.aidx = make_rsrc(arena + c.R.a_obs_a_idx + (int64_t)c.e * aidx_bytes, aidx_bytes);
  const bool pl = c.P.c.planner_gets_spatial_info != 0;
  o.pmap = make_rsrc(arena + c.R.a_obs_p_map + (int64_t)c.e * CM * HW * 4, pl ? (uint32_t)(CM * HW * 4) : 0u);
  o.pidx = make_rsrc(arena + c.R.a_obs_p_idx + (int64_t)c.e * 2 * HW * 2, pl ? (uint32_t)(2 * HW * 2) : 0u);
  return o;
}

// one (agent i, window cell d) item of the egocentric crop, layout_from_file.py:470-505
template <bool WATER>
__device__ __forceinline__ void crop_item(const Ctx& c, const SpatialOut& o, int i, int d, int r, int col) {
  constexpr int CM = WATER ? 6 : 5;
  const int H = c.P.H, W = c.P.W, WV2 = c.P.WV * c.P.WV;
  const int so = i * (CM + 1) * WV2 * 4, si = i * 2 * WV2 * 2;
  const bool in = (r >= 0) & (r < H) & (col >= 0) & (col < W);
  const int cell = in ? r * W + col : 0;
  const uint32_t cw = in ? R_CELLS(c)[cell] : 0x00ff0000u;  // :480-485
  float ch[6];
  cell_channels<WATER>(cw, ch);
#pragma unroll
  for (int k = 0; k < CM; ++k) buf_store_f32(o.amap, ch[k], 4 * d, so + k * WV2 * 4);
  buf_store_f32(o.amap, in ? 1.0f : 0.0f, 4 * d, so + CM * WV2 * 4);
  const int own = AIE_CELL_OWNER(cw);
  int v0 = own >= 0 ? own + 2 : 0;
  int v1 = in ? (int)c.locmap[cell] : 0;
  v1 = v1 ? v1 + 1 : 0;     // agent k -> k + 2
  if (v0 == i + 2) v0 = 1;  // :503
  if (v1 == i + 2) v1 = 1;
  buf_store_i16(o.aidx, v0, 2 * d, si);
  buf_store_i16(o.aidx, v1, 2 * d, si + WV2 * 2);
}
// the whole crop of agent i (wave-uniform i; lanes over the window cells)
template <bool WATER>
__device__ __forceinline__ void crop_agent(const Ctx& c, const SpatialOut& o, int i) {
  const int WV = c.P.WV, WV2 = WV * WV, w = c.P.c.obs_range;
  const int r0 = R_I32(c, o_loc_r)[i] - w, c0 = R_I32(c, o_loc_c)[i] - w;  // LDS broadcast reads
  for (int d = c.tid; d < WV2; d += AIE_NT) {
    const int dr = udiv(d, WV, c.P.mg_WV), dc = d - dr * WV;
    crop_item<WATER>(c, o, i, d, r0 + dr, c0 + dc);
  }
}
// one cell of the planner's full map (:507-517)
template <bool WATER>
__device__ __forceinline__ void planner_cell(const Ctx& c, const SpatialOut& o, int cell) {
  constexpr int CM = WATER ? 6 : 5;
  const int HW = c.P.HW;
  float ch[6];
  const uint32_t cw = R_CELLS(c)[cell];
  cell_channels<WATER>(cw, ch);
#pragma unroll
  for (int k = 0; k < CM; ++k) buf_store_f32(o.pmap, ch[k], 4 * cell, k * HW * 4);
  const int own = AIE_CELL_OWNER(cw);
  const int occ = c.locmap[cell];
  buf_store_i16(o.pidx, own >= 0 ? own + 2 : 0, 2 * cell, 0);
  buf_store_i16(o.pidx, occ ? occ + 1 : 0, 2 * cell, HW * 2);
}

// LayoutFromFile.generate_observations layout_from_file.py:412-517: the egocentric
// (2w+1)^2 crop for every agent and the full map for the planner.
//   crop:    one lane per window cell of a (wave-uniform) agent: one LDS word read feeds
//            CM+1 coalesced f32 stores and 2 i16 stores (the out-of-world ring: all 0);
//   planner: one lane per 4 consecutive cells: one 16-byte LDS read feeds CM 16-byte
//            stores and two 8-byte stores.
template <bool WATER>
__device__ __forceinline__ void write_spatial_observations_t(const Ctx& c, uint8_t* __restrict__ arena) {
  constexpr int CM = WATER ? 6 : 5;
  const int n = c.P.n, HW = c.P.HW;
  const uint32_t* cells = R_CELLS(c);
  const SpatialOut o = spatial_out<WATER>(c, arena);
  if (c.P.c.full_observability) {
    // layout_from_file.py:466-472: every agent gets the whole map (CM channels, no in-bounds
    // channel) and the owner / occupant planes with its own id replaced by 1
    const uint32_t* lm4f = reinterpret_cast<const uint32_t*>(c.locmap);
    for (int i = 0; i < n; ++i) {
      const int so = i * CM * HW * 4, si = i * 2 * HW * 2;
      for (int q = c.tid; q < (HW >> 2); q += AIE_NT) {
        const uint4 cw = reinterpret_cast<const uint4*>(cells)[q];
        const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
        float ch[4][6];
#pragma unroll
        for (int u = 0; u < 4; ++u) cell_channels<WATER>(cws[u], ch[u]);
#pragma unroll
        for (int k = 0; k < CM; ++k) buf_store_f32x4(o.amap, ch[0][k], ch[1][k], ch[2][k], ch[3][k], 16 * q, so + k * HW * 4);
        const uint32_t occ4 = lm4f[q];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int own = AIE_CELL_OWNER(cws[u]);
          int v0 = own >= 0 ? own + 2 : 0;
          const int occ = (int)((occ4 >> (8 * u)) & 0xffu);
          int v1 = occ ? occ + 1 : 0;
          if (v0 == i + 2) v0 = 1;
          if (v1 == i + 2) v1 = 1;
          buf_store_i16(o.aidx, v0, 8 * q + 2 * u, si);
          buf_store_i16(o.aidx, v1, 8 * q + 2 * u, si + HW * 2);
        }
      }
      for (int cell = 4 * (HW >> 2) + c.tid; cell < HW; cell += AIE_NT) {  // tail cells
        float ch[6];
        const uint32_t cw = cells[cell];
        cell_channels<WATER>(cw, ch);
#pragma unroll
        for (int k = 0; k < CM; ++k) buf_store_f32(o.amap, ch[k], 4 * cell, so + k * HW * 4);
        const int own = AIE_CELL_OWNER(cw);
        int v0 = own >= 0 ? own + 2 : 0;
        const int occ = c.locmap[cell];
        int v1 = occ ? occ + 1 : 0;
        if (v0 == i + 2) v0 = 1;
        if (v1 == i + 2) v1 = 1;
        buf_store_i16(o.aidx, v0, 2 * cell, si);
        buf_store_i16(o.aidx, v1, 2 * cell, si + HW * 2);
      }
    }
  } else {
    for (int i = 0; i < n; ++i) crop_agent<WATER>(c, o, i);
  }
  if (c.P.c.planner_gets_spatial_info) {
    // f32 channel planes: 4 consecutive cells per lane -> CM 16-byte stores.  The planes
    // are only dword-aligned (H*W need not be a multiple of 4): dwordx4 stores need no more
    // than that on gfx950.  The i16 owner / location planes of the same 4 cells: 8-byte stores.
    const int nq4 = HW >> 2;
    const uint32_t* lm4 = reinterpret_cast<const uint32_t*>(c.locmap);
    for (int q = c.tid; q < nq4; q += AIE_NT) {
      const uint4 cw = reinterpret_cast<const uint4*>(cells)[q];
      const uint32_t cws[4] = {cw.x, cw.y, cw.z, cw.w};
      float ch[4][6];
#pragma unroll
      for (int u = 0; u < 4; ++u) cell_channels<WATER>(cws[u], ch[u]);
#pragma unroll
      for (int k = 0; k < CM; ++k) buf_store_f32x4(o.pmap, ch[0][k], ch[1][k], ch[2][k], ch[3][k], 16 * q, k * HW * 4);
      const uint32_t occ4 = lm4[q];
      uint32_t ow[2], oc[2];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int own = AIE_CELL_OWNER(cws[u]);
        const uint32_t vo = (uint32_t)(own >= 0 ? own + 2 : 0);
        const uint32_t occ = (occ4 >> (8 * u)) & 0xffu;
        const uint32_t vc = occ ? occ + 1 : 0;
        if (u & 1) { ow[u >> 1] |= vo << 16; oc[u >> 1] |= vc << 16; }
        else { ow[u >> 1] = vo; oc[u >> 1] = vc; }
      }
      buf_store_u32x2(o.pidx, ow[0], ow[1], 8 * q, 0);
      if ((HW & 1) == 0) {  // second plane starts 4-byte aligned too
        buf_store_u32x2(o.pidx, oc[0], oc[1], 8 * q, HW * 2);
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) buf_store_i16(o.pidx, (int)((oc[u >> 1] >> (16 * (u & 1))) & 0xffffu), 8 * q + 2 * u, HW * 2);
      }
    }
    for (int cell = 4 * nq4 + c.tid; cell < HW; cell += AIE_NT) planner_cell<WATER>(c, o, cell);  // tail cells
  }
}

// one cell of agent i's whole-map observation (full_observability, layout_from_file.py:466-472)
template <bool WATER>
__device__ __forceinline__ void agent_full_cell(const Ctx& c, const SpatialOut& o, int i, int cell) {
  constexpr int CM = WATER ? 6 : 5;
  const int HW = c.P.HW;
  const int so = i * CM * HW * 4, si = i * 2 * HW * 2;
  float ch[6];
  const uint32_t cw = R_CELLS(c)[cell];
  cell_channels<WATER>(cw, ch);
#pragma unroll
  for (int k = 0; k < CM; ++k) buf_store_f32(o.amap, ch[k], 4 * cell, so + k * HW * 4);
  const int own = AIE_CELL_OWNER(cw);
  int v0 = own >= 0 ? own + 2 : 0;
  const int occ = c.locmap[cell];
  int v1 = occ ? occ + 1 : 0;
  if (v0 == i + 2) v0 = 1;
  if (v1 == i + 2) v1 = 1;
  buf_store_i16(o.aidx, v0, 2 * cell, si);
  buf_store_i16(o.aidx, v1, 2 * cell, si + HW * 2);
}

// Incremental form of write_spatial_observations: the tensors still hold the previous step's
// values; rewrite the crops of the agents that moved and every item that shows one of the
// cells the dynamics logged (dirty_*).  One lane per logged cell.
template <bool WATER>
__device__ __forceinline__ void update_spatial_observations_t(const Ctx& c, uint8_t* __restrict__ arena) {
  const int n = c.P.n, W = c.P.W, WV = c.P.WV, w = c.P.c.obs_range;
  const int cnt = uni(c.dirty[0]);
  if (cnt > AIE_DIRTY_CAP) {  // log overflow: start over
    write_spatial_observations_t<WATER>(c, arena);
    return;
  }
  const SpatialOut o = spatial_out<WATER>(c, arena);
  const uint32_t mv0 = (uint32_t)uni(c.dirty[1]), mv1 = (uint32_t)uni(c.dirty[2]);
  const uint16_t* dl = dirty_list(c);
  const int cell = c.tid < cnt ? (int)dl[c.tid] : -1;  // AIE_DIRTY_CAP == wave size
  const int r = cell >= 0 ? udiv(cell, W, c.P.mg_W) : 0, col = cell - r * W;
  for (int i = 0; i < n; ++i) {
    if (c.P.c.full_observability) {
      if (cell >= 0) agent_full_cell<WATER>(c, o, i, cell);
      continue;
    }
    const bool moved = ((i < 32 ? mv0 >> i : mv1 >> (i - 32)) & 1u) != 0;
    if (moved) {
      crop_agent<WATER>(c, o, i);
    } else if (cell >= 0) {
      const int dr = r - (R_I32(c, o_loc_r)[i] - w), dc = col - (R_I32(c, o_loc_c)[i] - w);
      if (dr >= 0 && dr < WV && dc >= 0 && dc < WV) crop_item<WATER>(c, o, i, dr * WV + dc, r, col);
    }
  }
  if (c.P.c.planner_gets_spatial_info && cell >= 0) planner_cell<WATER>(c, o, cell);
}
__device__ __forceinline__ void write_spatial_observations(const Ctx& c, uint8_t* __restrict__ arena) {
  if (c.P.c.has_water) write_spatial_observations_t<true>(c, arena);
  else write_spatial_observations_t<false>(c, arena);
}
__device__ __forceinline__ void update_spatial_observations(const Ctx& c, uint8_t* __restrict__ arena) {
  if (c.P.c.has_water) update_spatial_observations_t<true>(c, arena);
  else update_spatial_observations_t<false>(c, arena);
}

// Component + scalar observations, packed in SORTED key order (base_env.py:561-612):
// Build (build.py:163-178), CDA (continuous_double_auction.py:491-542), Gather
// (move.py:155-165), PeriodicBracketTax (redistribution.py:974-1023), time, world-*.
// Built in LDS (c.stage) then streamed out.
__device__ __forceinline__ void write_flat_observations(const Ctx& c, uint8_t* __restrict__ arena) {
  const aie_params& P = c.P;
  const int n = P.n, tid = c.tid, Pp = P.P, NB = P.NB;
  const double isc = P.c.allow_observation_scaling ? 0.01 : 1.0;
  // The agents' vectors (n x FA floats) go straight to HBM through a buffer descriptor; only the
  // planner's vector (read back by the agents' CDA fragment) and its per-agent fragments are
  // staged in LDS, behind the components' draw window (the first region of the staging area, see step_body).
  float* s_pag = c.stage + stage_window_words(P);
  float* s_pflat = s_pag + pad4(n * P.FPA);
  const BufRsrc aflat = make_rsrc(arena + c.R.a_obs_a_flat + (int64_t)c.e * n * P.FA * 4, (uint32_t)(n * P.FA * 4));
  auto AF = [&](int idx, float v) { buf_store_f32(aflat, v, 4 * idx, 0); };
  const int t = *R_I32(c, o_timestep);
  const float tval = (float)((double)t / (P.c.allow_observation_scaling ? (double)c.R.c.episode_length : 1.0));

  const int skip = c.skipm;
  // ================= stage A: per-(commodity, price) sums, per-agent scalars ===========
  if (P.has_cda && !(skip & 64)) {
    // lanes over (commodity r, price k): net price history and full bid / ask histograms
    double* net_ph = scr_net_ph(c);
    for (int q = tid; q < 2 * Pp; q += AIE_NT) {
      const int r = q >= Pp ? 1 : 0, k = q - r * Pp;
      double s = 0;
      int fa = 0, fb = 0;
      for (int i = 0; i < n; ++i) {
        const double v = R_F64(c, o_cda_price_history)[(r * n + i) * Pp + k];
        s = (i == 0) ? v : s + v;  // np.sum(np.stack(...), axis=0): row by row
        fa += R_U8(c, o_cda_ask_hist)[(r * n + i) * Pp + k];
        fb += R_U8(c, o_cda_bid_hist)[(r * n + i) * Pp + k];
      }
      net_ph[q] = s;
      float* g = s_pflat + P.fp_cda;  // planner: full_asks, full_bids, price_history
      g[0 * Pp + q] = (float)fa;
      g[2 * Pp + q] = (float)fb;
      g[4 * Pp + 2 + q] = (float)(s * isc);
    }
  }
  if (tid < n && !(skip & 64)) {
    const int i = tid;
    const int f0 = i * P.FA;
    const double coin = R_F64(c, o_inv_coin)[i];
    const int inv0 = R_I32(c, o_inv_res)[i], inv1 = R_I32(c, o_inv_res)[n + i];
    const int lr = R_I32(c, o_loc_r)[i], lc = R_I32(c, o_loc_c)[i];
    if (P.has_build) {  // build.py:163-178
      AF(f0 + P.fa_build + 0, (float)(R_F64(c, o_build_payment)[i] / (double)c.R.c.build_payment));
      AF(f0 + P.fa_build + 1, (float)R_F64(c, o_build_skill)[i]);
    }
    if (P.has_gather) AF(f0 + P.fa_gather, (float)R_F64(c, o_bonus_gather_prob)[i]);  // move.py:155-165
    AF(f0 + P.fa_time, tval);
    const float w0 = (float)(coin * isc);
    const float w1 = (float)((double)inv0 * isc);
    const float w2 = (float)((double)inv1 * isc);
    const float w3 = (float)((double)lc / (double)P.W);
    const float w4 = (float)((double)lr / (double)P.H);
    AF(f0 + P.fa_world + 0, w0); AF(f0 + P.fa_world + 1, w1); AF(f0 + P.fa_world + 2, w2);
    float* q = s_pag + i * P.FPA;
    if (!P.c.full_observability) {  // locations and the planner's per-agent fragments: egocentric mode only
      AF(f0 + P.fa_world + 3, w3); AF(f0 + P.fa_world + 4, w4);
      q[P.fpa_world + 0] = w0; q[P.fpa_world + 1] = w1; q[P.fpa_world + 2] = w2;
      if (P.c.planner_gets_spatial_info) { q[P.fpa_world + 3] = w3; q[P.fpa_world + 4] = w4; }
    }
    reinterpret_cast<float*>(arena + c.R.a_obs_a_time)[(int64_t)c.e * n + i] = tval;
    if (P.has_tax) {
      // last_incomes sorted ascending (redistribution.py:908-911): rank by counting
      const double per = (double)c.R.c.tax_period;
      const double x = R_F64(c, o_tax_last_income)[i] / per;
      double* xs = scr_cmr(c);  // (free until the rewards: one division per agent, not one per pair)
      xs[i] = x;
      AIE_WSYNC();
      int rank = 0;
      for (int j = 0; j < n; ++j) {
        const double y = xs[j];
        rank += (y < x || (y == x && j < i)) ? 1 : 0;
      }
      scr_sorted_inc(c)[rank] = x;
      const double cmr = tax_marginal_rate(c, (coin + R_F64(c, o_esc_coin)[i]) - R_F64(c, o_tax_last_coin)[i]);
      AF(f0 + P.fa_tax + NB + 2 + n, (float)cmr);
      q[P.fpa_tax + 0] = (float)cmr;
      q[P.fpa_tax + 1] = (float)x;
      q[P.fpa_tax + 2] = (float)R_F64(c, o_tax_last_marginal_rate)[i];
    }
  }
  if (tid == 0) {
    s_pflat[P.fp_time] = tval;
    s_pflat[P.fp_world + 0] = 0.0f;  // the planner's inventory never changes
    s_pflat[P.fp_world + 1] = 0.0f;
    s_pflat[P.fp_world + 2] = 0.0f;
    reinterpret_cast<float*>(arena + c.R.a_obs_p_time)[c.e] = tval;
  }
  AIE_WSYNC();

  // ================= stage B: fill the vectors, one lane per element =====================
  if (P.has_cda && !(skip & 128)) {
    // continuous_double_auction.py:491-542
    if (tid < 2) {
      const int r = tid;
      const double* a = scr_net_ph(c) + r * Pp;
      double dot = 0;
      for (int k = 0; k < Pp; ++k) dot += (double)k * a[k];
      const double tot = np_sum_small(a, Pp);
      const float mr = (float)(dot / (tot > 0.001 ? tot : 0.001));
      s_pflat[P.fp_cda + 4 * Pp + r] = mr;
      for (int i = 0; i < n; ++i) AF(i * P.FA + P.fa_cda + 4 * Pp + r, mr);
    }
    const float* g = s_pflat + P.fp_cda;
    for (int it = tid; it < n * 2 * Pp; it += AIE_NT) {
      const int i = udiv(it, 2 * Pp, P.mg_2P);
      const int q = it - i * 2 * Pp;  // (commodity, price)
      const int r = q >= Pp ? 1 : 0, k = q - r * Pp;
      const float mya = (float)R_U8(c, o_cda_ask_hist)[(r * n + i) * Pp + k];
      const float myb = (float)R_U8(c, o_cda_bid_hist)[(r * n + i) * Pp + k];
      const int fc = i * P.FA + P.fa_cda;
      AF(fc + 0 * Pp + q, g[0 * Pp + q] - mya);    // available_asks
      AF(fc + 2 * Pp + q, g[2 * Pp + q] - myb);    // available_bids
      AF(fc + 4 * Pp + 2 + q, mya);                // my_asks
      AF(fc + 6 * Pp + 2 + q, myb);                // my_bids
      AF(fc + 8 * Pp + 2 + q, g[4 * Pp + 2 + q]);  // price_history
    }
  }
  if (P.has_tax && !(skip & 256)) {
    // redistribution.py:974-1023; lanes over (agent or planner, element of the fragment)
    const int pos = *R_I32(c, o_tax_cycle_pos);
    const float is_tax_day = pos >= c.R.c.tax_period ? 1.0f : 0.0f;
    const float is_first_day = pos == 1 ? 1.0f : 0.0f;
    const float tax_phase = (float)((double)pos / (double)c.R.c.tax_period);
    const int fragA = NB + n + 4;
    for (int q = tid; q < (n + 1) * fragA; q += AIE_NT) {
      const int i = udiv(q, fragA, P.mg_taxA);
      const int j = q - i * fragA;
      const bool planner = i == n;
      if (j == NB + 2 + n) {  // marginal_rate: written in stage A (agents), absent for the planner
        if (planner) s_pflat[P.fp_tax + NB + 2 + n] = tax_phase;
        continue;
      }
      if (planner && j == NB + 3 + n) continue;
      float v;
      if (j < NB) v = (float)tax_rate_obs(c, j);
      else if (j == NB) v = is_first_day;
      else if (j == NB + 1) v = is_tax_day;
      else if (j < NB + 2 + n) v = (float)scr_sorted_inc(c)[j - NB - 2];
      else v = tax_phase;
      if (planner) s_pflat[P.fp_tax + j] = v;
      else AF(i * P.FA + P.fa_tax + j, v);
    }
  }

  AIE_WSYNC();

  // ---- stream the staged vectors out: 16-byte LDS reads, dword-aligned 16-byte stores ----
  if (!(skip & 1024)) {
    stream_out(s_pag, reinterpret_cast<float*>(arena + c.R.a_obs_p_agents) + (int64_t)c.e * n * P.FPA, n * P.FPA, tid);
    stream_out(s_pflat, reinterpret_cast<float*>(arena + c.R.a_obs_p_flat) + (int64_t)c.e * P.FP, P.FP, tid);
  }
}


// Action masks (own staging slots, own per-agent mask bits): independent of the flat vectors
// above, so the second wave of a replica builds them while the first one does those.
// `all` == false (a step whose observation tensors still show the previous step): only what changed is rewritten -- the
// rows of the agents whose mask bits differ from the ones the tensor shows (record field o_mask_bits), the planner's when
// its "rates may be set today" flag flips (o_mask_p_open; which rates are visible only changes at a reset, and a reset
// writes everything).  Round 6: 1.4 KB of stores and ~100 vector instructions per replica-step less on BASELINE configs[1].
__device__ __forceinline__ void write_action_masks(const Ctx& c, uint8_t* __restrict__ arena, bool all = true) {
  const aie_params& P = c.P;
  const int n = P.n, tid = c.tid, Pp = P.P;
  const int skip = c.skipm;
  if (skip & 512) return;
  if (!P.o_mask_bits) all = true;
  if (tid < n) {
    const int i = tid;
    const double coin = R_F64(c, o_inv_coin)[i];
    const int inv0 = R_I32(c, o_inv_res)[i], inv1 = R_I32(c, o_inv_res)[n + i];
    const int lr = R_I32(c, o_loc_r)[i], lc = R_I32(c, o_loc_c)[i];
    // mask bits: Build build.py:180-193, Gather move.py:167-188, CDA :544-580
    uint32_t mf = 0;
    if (P.has_build) mf |= agent_can_build(c, i) ? 1u : 0u;
    if (P.has_gather) {
      mf |= can_agent_occupy(c, lr, lc - 1, i) ? 2u : 0u;
      mf |= can_agent_occupy(c, lr, lc + 1, i) ? 4u : 0u;
      mf |= can_agent_occupy(c, lr - 1, lc, i) ? 8u : 0u;
      mf |= can_agent_occupy(c, lr + 1, lc, i) ? 16u : 0u;
    }
    if (P.has_cda) {
      int kmax = coin >= (double)(Pp - 1) ? Pp - 1 : (int)coin;  // price k is affordable iff k <= coin
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const bool quota = R_I32(c, o_cda_n_orders)[r * n + i] < P.c.cda_max_num_orders;
        if (quota && (r ? inv1 : inv0) > 0) mf |= 32u << r;
        if (quota) mf |= (uint32_t)(kmax + 1) << (8 + 8 * r);
      }
    }
    // bit 31 of the staged word: this agent's row has to be written
    const bool changed = all || (int32_t)mf != R_I32(c, o_mask_bits)[i];
    if (P.o_mask_bits) R_I32(c, o_mask_bits)[i] = (int32_t)mf;
    c.mflags[i] = (int32_t)(mf | (changed ? 0x80000000u : 0u));
  }
  AIE_WSYNC();
  // ---- masks: _generate_masks base_env.py:706-756 + flatten_masks base_agent.py:440-460,
  // planner PeriodicBracketTax.generate_masks redistribution.py:1025-1104 ----
  // Element m of the flattened mask means the same thing for every agent: one host-built (shift, mask, threshold)
  // test against each agent's mask bits.  Written straight from registers: lane q owns elements 4q .. 4q + 3 of the
  // replica's [n][MA] block (dword-aligned 16-byte stores).
  {
    float* dst = reinterpret_cast<float*>(arena + c.R.a_obs_a_mask) + (int64_t)c.e * n * P.MA;
    const int count = n * P.MA;
    auto elem = [&](int idx) -> float {
      const int i = udiv(idx, P.MA, P.mg_MA), m = idx - i * P.MA;
      const uint32_t t = c.mtab[m];  // host-built test for element m (aie_layout.h)
      const uint32_t sh = t & 31u, msk = (t >> 8) & 0xffu, thr = t >> 16;
      return (((uint32_t)c.mflags[i] >> sh) & msk) >= thr ? 1.0f : 0.0f;
    };
    auto dirty = [&](int idx) -> bool { return c.mflags[udiv(idx, P.MA, P.mg_MA)] < 0; };  // (bit 31)
    const int n4 = count >> 2;
    for (int q = tid; q < n4; q += AIE_NT) {
      if (!all && !dirty(4 * q) && !dirty(4 * q + 3)) continue;  // (a quad spans at most two agents' rows)
      f32x4_a4 o = {elem(4 * q), elem(4 * q + 1), elem(4 * q + 2), elem(4 * q + 3)};
      reinterpret_cast<f32x4_a4*>(dst)[q] = o;
    }
    for (int q = 4 * n4 + tid; q < count; q += AIE_NT)
      if (all || dirty(q)) dst[q] = elem(q);
    float* pdst = reinterpret_cast<float*>(arena + c.R.a_obs_p_mask) + (int64_t)c.e * P.MP;
    const bool pmulti = P.c.multi_action_mode_planner != 0;
    const float open = (P.n_sub_p && *R_I32(c, o_tax_cycle_pos) == 1) ? 1.0f : 0.0f;
    if (P.o_mask_bits) {
      const int was = uni(*R_I32(c, o_mask_p_open));
      AIE_WSYNC();
      if (tid == 0) *R_I32(c, o_mask_p_open) = open != 0.0f ? 1 : 0;
      if (!all && was == (open != 0.0f ? 1 : 0)) return;
    }
    for (int q = tid; q < P.MP; q += AIE_NT) {
      float v;
      int j = -1;  // index of the discretised rate this entry stands for (-1: a NO-OP entry)
      if (P.n_sub_p == 0) j = -1;
      else if (pmulti) j = q - udiv(q, 1 + P.sub_p_dim, P.mg_sub_p) * (1 + P.sub_p_dim) - 1;
      else if (q > 0) j = (q - 1) - udiv(q - 1, P.sub_p_dim, P.mg_sub_p_dim) * P.sub_p_dim;
      if (j < 0) v = 1.0f;
      else v = (open != 0.0f && tax_rate_action_visible(c, j)) ? 1.0f : 0.0f;
      pdst[q] = v;
    }
  }
}

__device__ __forceinline__ void rebuild_locmap(const Ctx& c) {
  uint32_t* lm = reinterpret_cast<uint32_t*>(c.locmap);
  const int nw = (c.P.HW + 3) >> 2;
  for (int q = c.tid; q < nw; q += AIE_NT) lm[q] = 0;
  AIE_WSYNC();
  if (c.tid < c.P.n) {
    const int r = R_I32(c, o_loc_r)[c.tid], col = R_I32(c, o_loc_c)[c.tid];
    if (r >= 0 && col >= 0) c.locmap[r * c.P.W + col] = (uint8_t)(c.tid + 1);
  }
  AIE_WSYNC();
}

// The replica's source doubles from the cells' flag bytes (LDS image), ascending: Wood cells, then Stone cells
// (layout_from_file.py:394-403: np.random.rand(2, H, W) is consumed for Wood first).  One wave; count and list go to
// the record in HBM (o_src_n, o_src_list; a count above AIE_SRC_CAP selects the row-by-row regeneration).
__device__ __forceinline__ void build_src_list(const Ctx& c, uint8_t* __restrict__ arena) {
  if (!c.P.o_src_list) return;
  const int HW = c.P.HW;
  uint8_t* g = arena + (int64_t)c.e * c.P.rec_bytes;
  uint16_t* lst = reinterpret_cast<uint16_t*>(g + c.P.o_src_list);
  const uint8_t* cb = reinterpret_cast<const uint8_t*>(R_CELLS(c));
  for (int k = c.tid; k < AIE_SRC_CAP; k += AIE_NT) lst[k] = 0;
  int base = 0;
  for (int rs = 0; rs < 2; ++rs) {
    const uint32_t bit = rs == 0 ? AIE_CELL_WOOD_SRC : AIE_CELL_STONE_SRC;
    for (int q0 = 0; q0 < HW; q0 += AIE_NT) {
      const int q = q0 + c.tid;
      const bool on = q < HW && (cb[4 * q + 3] & bit);
      const uint64_t mask = __ballot(on);
      const int slot = base + __popcll(mask & lanemask_lt(c.tid));
      if (on && slot < AIE_SRC_CAP) lst[slot] = (uint16_t)(rs * HW + q);
      base += __popcll(mask);
    }
  }
  if (c.tid == 0) *reinterpret_cast<int32_t*>(g + c.P.o_src_n) = base;
}

}  // namespace aie

// ======================================================================================
// Kernels
// ======================================================================================

// BaseEnvironment.step, F/base/base_env.py:929-1032: parse actions, timestep += 1,
// components in list order, scenario_step, observations, masks, rewards, done.
//
// A replica is a workgroup of TWO wavefronts sharing the LDS record (NW = 2; the template parameter remains from the
// one-wave schedule the kernel started with).
//   both   record HBM -> LDS (alternate 16-byte units)  ||  wave 1 also: the next 128+ tempered MT19937 words ->
//                                                        ||         the LDS draw window (rows fetched from HBM)
//   wave 0 action decode, price-history decay          ||  wave 1 occupancy map
//   wave 0 components (draws read the draw window)     ||  wave 1 generator rows HBM -> registers, next step's
//                                                      ||         random actions (opt.)
//   wave 0 flat observation vectors, rewards, done     ||  wave 1 regeneration (rows in registers, 4 twists),
//          (none of them looks at the map)             ||         incremental map observations, action masks
//   both   record LDS -> HBM (wave 1 also the generator's rows)
// After the components the two halves touch disjoint state: wave 0 reads agents / auction / tax
// fields and writes util + warm-up counters, wave 1 reads and writes the map cells and the
// generator.  With <= 64 VGPRs all 2 x 4096 waves of the C2 / C3 batch are resident at once (8 per SIMD).
// (Tried and dropped: writing the map observations speculatively during the dynamics and
// repairing the changed cells afterwards -- parity-clean, but the co-resident second waves'
// instruction stream slows the serial dynamics of the first waves by as much as it saves.)
// The three libm tables behind the utilities (aie_glibc_math.h: pow's log and exp tables, log's: 7 KB) touched once per
// workgroup, 64 bytes apart: the step's second wave calls this while the first one runs the components, so that the
// utilities' dependent table look-ups at the end of the step (two or three per pow, each a round trip to wherever the
// streaming traffic has pushed the tables) find them in this CU's vector cache.  Returns a value to keep alive.
__device__ __forceinline__ uint32_t glibc_tables_touch(int lane) {
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int line = 2 * lane + k;  // 112 lines of 64 bytes
    const uint8_t* p = line < 48 ? reinterpret_cast<const uint8_t*>(aie_glibc_powlog_tab) + 64 * line
                     : line < 80 ? reinterpret_cast<const uint8_t*>(aie_glibc_exp_tab) + 64 * (line - 48)
                                 : reinterpret_cast<const uint8_t*>(aie_glibc_log_tab) + 64 * (line - 80);
    if (line < 112) v ^= *reinterpret_cast<const volatile uint32_t*>(p);
  }
  return v;
}

struct NextActions {  // aie_step_sample_next: where and how to sample the next step's random actions
  // The replica count (and the replica range) FIRST: a workgroup needs them for its very first decision (which replica it
  // is), and here they share the argument segment's first 64 bytes with the parameter block's and the arena's pointers
  // -- one scalar-cache miss instead of two at the head of every workgroup (round 6; as a field of the parameter block
  // the count had been a second, dependent round trip: the caches are cold at every launch).
  int32_t E;
  // replicas this launch steps: [e_lo, e_hi), or all of them when e_hi == 0.  An environment with dense-log replicas
  // whose current episode is being logged steps those replicas with aie_step_kernel_log and the rest with its fast
  // kernel (aie_capi.hip: aie_step_impl)
  int32_t e_lo, e_hi;
  int32_t masked;  // aie_step_sample_next_masked (COVID): the next actions are drawn among what the new masks allow
  int32_t* a;
  int32_t* p;
  uint64_t seed;
  int64_t env_offset;  // (the draw index `t` of the counter RNG is the replica's own record field o_sample_t)
  // aie_step_range (custom host components, include/aie.h): the built-in components [comp_lo, comp_hi) of the list and
  // the parts of a step this launch performs -- AIE_STEP_HEAD (timestep += 1), AIE_STEP_TAIL (regeneration, observations,
  // masks, rewards, done), AIE_STEP_OBSERVE (observations and masks of the state as it stands, nothing else); phase == 0:
  // a whole step.  Honoured by the full-featured kernel only (aie_step_kernel_log); everybody else steps whole steps.
  int32_t comp_lo, comp_hi, phase;
};
// This step's slot of the reward log: the replica's slot counter selects it and moves on (`writer`: the one lane that
// stores the counter back; every lane of the wave calls this with the same fields).
// (`R`: the run-time parameter block in device memory, where aie_set_reward_log keeps the log's address, slot count and
// epoch -- nothing of it travels by value, so a captured launch follows a later aie_set_reward_log call)
__device__ __forceinline__ float* rew_log_claim(const aie_params& R, int32_t* slot_field, int32_t* epoch_field, int E,
                                                int n, bool writer) {
  float* const log = R.rew_log;
  if (!log) return nullptr;
  const int epoch = R.rew_epoch;
  int slot = __builtin_amdgcn_readfirstlane(*slot_field);
  if (__builtin_amdgcn_readfirstlane(*epoch_field) != epoch) slot = 0;
  if (writer) {
    *epoch_field = epoch;
    *slot_field = slot + 1 >= R.rew_slots ? 0 : slot + 1;
  }
  return log + (int64_t)slot * E * (n + 2);
}
// rewards of the step (compute_reward, layout_from_file.py:519-559), the reward log's slot, `done` and the completed-episode
// count: one wave
__device__ __forceinline__ void step_rewards_and_done(const aie::Ctx& c, uint8_t* __restrict__ arena, const NextActions& next, int skip) {
  using namespace aie;
  float* const rew_log = rew_log_claim(c.R, R_I32(c, o_rew_slot), R_I32(c, o_rew_epoch), c.R.E, c.P.n, c.tid == 0);
  if (!(skip & 16)) compute_rewards(c, arena, rew_log);
  AIE_WSYNC();
  if (c.tid == 0) {
    const int done = *R_I32(c, o_timestep) >= c.R.c.episode_length;
    (arena + c.R.a_done)[c.e] = (uint8_t)done;
    if (rew_log) rew_log[(int64_t)c.e * (c.P.n + 2) + c.P.n + 1] = done ? 1.0f : 0.0f;
    if (done) *R_I32(c, o_completions) += 1;
  }
}
// SPEC >= 0: a compile-time instance (aie_spec_generated.h): P is a constant image of the parameter block of one
// configuration -- every dimension, record offset, component list, mask table and magic divisor folds into the
// instruction stream (no scalar loads of parameters, fully unrolled per-agent loops) -- and only what depends on
// the batch (R: E, arena offsets) is read at run time.  SPEC < 0: the generic kernel, P == R == *params.
template <int NW, bool LOG, int SPEC = -1, bool TRACE = LOG>
__device__ __forceinline__ void step_body(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                                          const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p,
                                          uint8_t* lds, const NextActions& next) {
  using namespace aie;
  // The parameter block lives in device memory (uniform scalar loads).  Passing the 2.7 KB
  // struct by value made the compiler copy it to scratch on every launch (5x slower).
  // (round 6: requesting every kernel argument in one batch up here cost 2 us -- scalar-cache misses are served one at a
  // time, and most arguments are not needed before the record is on its way; see the sampler for the opposite case)
  const aie_params& R = *params;
  const aie_params& P = aie_spec_params<SPEC>(params);
  const int wid = NW == 1 ? 0 : uni((int)(threadIdx.x >> 6));
#ifdef AIE_DEV  // phases switched off by aie_dev_set_skip_mask (the generic kernel with hooks and the traced instances)
  const int skip = (LOG || TRACE) ? R.dev_skip_mask : 0;
#else
  const int skip = 0;
#endif
  static_assert(NW == 2, "a replica is a workgroup of two wavefronts");
  const int e_blk = replica_of_block((int)blockIdx.x, next.E);  // (the replica count travels as a kernel argument)
  if (next.e_hi > 0 && (e_blk < next.e_lo || e_blk >= next.e_hi)) return;  // (uniform over the workgroup, ahead of every barrier)
  const bool FAST = rng_fast(P);  // the counter stream (include/aie.h: AIE_RNG_FAST); compile-time in the instances
  const int tid0 = (int)(threadIdx.x & (AIE_NT - 1));
  uint8_t* grec = arena + (int64_t)e_blk * P.rec_bytes;  // (a_records == 0: the records open the arena, aie_layout.h)
  uint32_t* gkey = reinterpret_cast<uint32_t*>(grec + P.o_mt);
  const Ctx c = make_ctx(P, R, lds, e_blk, tid0, arena, LOG, skip,
                         /*lds_tables=*/SPEC >= 0);  // (the generic kernel keeps the parameter block's arrays: a pointer
                                                     // that may be LDS or global at run time costs it flat accesses and spills)
  const bool SHL = shared_src_list(P);  // one list of source doubles for the whole batch (a_src_list)
  uint32_t* const draw_w = reinterpret_cast<uint32_t*>(c.stage);  // the components' draw window (LDS)
  int draw_cap = stage_window_words(P);
  if (LOG && R.dev_draw_window > 0 && R.dev_draw_window < draw_cap) draw_cap = R.dev_draw_window;  // tests: force refills
  // With many agents the first wave's tail (flat vectors of every agent, utilities) is the longer one: it then keeps a
  // priority above the waves that are still loading (measured at 10 agents: 42.3 -> 41.9 us; at 4 agents any
  // priority above 0 costs 0.8-1.4 us)
#ifdef AIE_W0_TAIL_PRIO  // (A/B builds)
  const int w0_tail_prio = AIE_W0_TAIL_PRIO;
#else
  const int w0_tail_prio = P.n >= 8 ? 2 : 0;
#endif
  // Where the rewards run.  With MT19937 the second wave's tail (four twists of regeneration, map observations, masks) is
  // as long as the first wave's (flat vectors, rewards); with the counter stream the regeneration shrinks to a few Philox
  // blocks and the first wave's tail is the long pole (tools/block_trace.py: 7.1 us against 3.7 us), so the rewards move
  // behind the masks on the second wave -- up to 7 agents (A/B on one box, tools/ab_variants.sh: C2f 22.1 -> 21.7 us; with
  // ten agents the utilities' n^2 gini terms make them the longer piece: C3f 35.3 -> 37.4 us, so they stay where they
  // were).  They share no scratch slot with the flat-vector writer (current_metrics).
  const bool REW_ON_W1 = FAST && P.n < 8;
  // partial steps (aie_step_range): compile-time "whole step" everywhere but in the full-featured kernel
  const int ph = LOG ? next.phase : 0;
  const bool HEAD = ph == 0 || (ph & 1), TAIL = ph == 0 || (ph & 2), OBSERVE = ph != 0 && (ph & 4) && !(ph & 2);
  const bool REBASE = OBSERVE && (ph & 8);  // utilities := current (the reward baseline a reset leaves, layout_from_file.py:347-349)
  const bool RETAX = OBSERVE && (ph & 16);  // PeriodicBracketTax's reset-time snapshot of the agents' coin (redistribution.py:1106-1110)
  const int c_lo = ph ? next.comp_lo : 0, c_hi = ph ? next.comp_hi : P.c.n_components;
  // The two waves run two separate ARMS from here to the end, each with its own copy of the four workgroup barriers
  // (the branch is wave-uniform, every wave passes the same number of them): a value that only one wave carries --
  // the agents' registers and the draw cache of the first, the generator's ten rows of the second -- is then live in
  // its own arm only and does not count against the other wave's code in the 64-register budget.
  if (wid == 0) {
    // ---------------- first wave: actions, components, flat vectors, rewards ----------------
    Agents A;
    int act_err = 0;
    if (!(skip & (1 << 18))) act_err = decode_actions(c, A, act_a, act_p);  // (c.act_p is LDS scratch)
    else A.act = 0;
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x] = wall_clock64();
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 9] = wall_clock64();
    // (round 6, tools/ab2.sh: requesting the action words and the wave's whole share of the image before anything waits
    // -- the loop below comes out as load, wait, LDS write per unit, behind decode_actions' two load-wait-decode rounds --
    // made the launch SLOWER, 22.5 -> 23.5 us, and so did milder orders (both action loads first: 22.7; the record ahead of
    // the decode: 22.9): 4096 workgroups start together, and what their first microseconds are short of is the memory
    // system's capacity for requests, not patience -- the loop's own pace spreads them)
    MT none;
    load_record(c, arena, none, 0, NW, /*key_wave=*/-1);
    __syncthreads();  // (2) the record is in LDS
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 8] = wall_clock64();
    MTL ml{draw_w, 0, 0u, -AIE_MT_N, 0, 0u, 0u, 0, 0, 0, FAST ? reinterpret_cast<uint32_t*>(c.rec + P.o_mt) : gkey, draw_cap, FAST};
    ml.pos = ml.base = uni(*R_I32(c, o_mt_pos));
    ml.avail = ml.cap < AIE_MT_N - ml.pos ? ml.cap : AIE_MT_N - ml.pos;  // what the second wave's draw_window_publish left
    if (FAST) {  // (draw_window_publish_fast: whole pairs, one word less when the position is odd)
      const int covered = 128 * ((ml.cap + 127) >> 7) - (ml.pos & 1);
      if (ml.avail > covered) ml.avail = covered;
    }
    agents_load(c, A);
    if (act_err && c.tid == 0) *R_I32(c, o_error_flags) |= act_err;
    bool cda_in_range = ph == 0;  // (the decay opens ContinuousDoubleAuction.component_step: with the component)
    if (ph != 0)
      for (int k = c_lo; k < c_hi; ++k) cda_in_range |= P.c.components[k] == AIE_COMP_CDA;
    if (P.has_cda && !(skip & 1) && cda_in_range) cda_decay_price_history(c);
    __syncthreads();  // (3) occupancy map rebuilt
    if (c.tid == 0 && HEAD) *R_I32(c, o_timestep) += 1;
    if (c.ev && c.tid == 0) c.srcn[2] = 0;
    __builtin_amdgcn_s_setprio(3);  // the serial dynamics are the replica's critical path (at any batch size: round 6 A/B)
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 1] = wall_clock64();
    if (!(skip & 1)) {
      for (int k = c_lo; k < c_hi; ++k) {
        switch (P.c.components[k]) {
          case AIE_COMP_BUILD: if (!(skip & 2048)) build_component_step(c, ml, A); break;
          case AIE_COMP_CDA: if (!(skip & 4096)) cda_component_step(c, A); break;
          case AIE_COMP_GATHER: if (!(skip & 8192)) gather_component_step(c, ml, A); break;
          case AIE_COMP_TAX: if (!(skip & 16384)) tax_component_step(c, ml, A); break;
          case AIE_COMP_WEALTH_REDISTRIBUTION: wealth_component_step(c, A); break;
          default: break;
        }
        if (TRACE && R.dev_trace && c.tid == 0 && k < 4) R.dev_trace[12 * blockIdx.x + 2 + k] = wall_clock64();
      }
    }
    __builtin_amdgcn_s_setprio(0);
    agents_store(c, A);
    if (c.ev && c.tid == 0) c.ev[0] = c.srcn[2];
    if (c.tid == 0) {
      *R_I32(c, o_mt_pos) = ml.pos;
      c.dirty[0] = ml.dn;
      c.dirty[1] = (int32_t)ml.mv0;
      c.dirty[2] = (int32_t)ml.mv1;
      c.dirty[3] = ml.tw;
    }
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 5] = wall_clock64();
    __syncthreads();  // (4) components done; the generator's position (and, after a refill that twisted, its state in HBM) is final
    // flat observation vectors and rewards: neither looks at the map
    if (w0_tail_prio) __builtin_amdgcn_s_setprio(2);
    if (!(skip & 8) && (TAIL || OBSERVE)) write_flat_observations(c, arena);
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 10] = wall_clock64();
    if (!REW_ON_W1 && TAIL) step_rewards_and_done(c, arena, next, skip);
    if (RETAX && P.has_tax && c.tid < P.n)
      R_F64(c, o_tax_last_coin)[c.tid] = R_F64(c, o_inv_coin)[c.tid] + R_F64(c, o_esc_coin)[c.tid];
    if (REBASE) {  // (reset_body's last lines: the metrics of the state as the host's reset hooks left it)
      AIE_WSYNC();
      current_metrics(c);
      AIE_WSYNC();
      if (c.tid <= P.n) R_F64(c, o_util)[c.tid] = scr_part(c)[c.tid];
      AIE_WSYNC();
    }
    if (w0_tail_prio) __builtin_amdgcn_s_setprio(0);
    __syncthreads();  // (5)
    if (!(skip & 32)) store_record_step(c, arena, 0, NW);
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 7] = wall_clock64();
  } else {
    // ---------------- second wave: generator, occupancy map, regeneration, map observations, masks ----------------
    MT m;
    mt_init(m, P);
    const int gpos = *reinterpret_cast<const int32_t*>(grec + P.o_mt_pos);
    load_record(c, arena, m, 1, NW, /*key_wave=*/-1);
    // behind its share of the copy: the words the components will draw -> the LDS draw window (two or three rows of the
    // generator's state, fetched from HBM; the state itself follows while the components run)
    if (skip & (1 << 16)) {}  // (development: the load phase without the draw window)
    else if (FAST) draw_window_publish_fast(draw_w, draw_cap, (uint32_t)uni((int)gkey[0]), (uint32_t)uni((int)gkey[1]), (uint32_t)uni((int)gkey[2]), uni(gpos), c.tid);
    else draw_window_publish_from_hbm(draw_w, draw_cap, gkey, uni(gpos), c.tid);

    if (SPEC >= 0 && const_tables_in_lds(P)) {
      // the small constant tables (Ctx.rtab / mtab) -> LDS, published by the barrier below
      if (P.has_tax && P.c.tax_model == AIE_TAX_MODEL_WRAPPER)
        for (int q = c.tid; q < P.c.tax_n_disc_rates; q += AIE_NT) const_cast<double*>(c.rtab)[q] = R.c.tax_disc_rates[q];
      for (int q = c.tid; q < P.MA; q += AIE_NT) const_cast<uint32_t*>(c.mtab)[q] = P.mask_test[q];
    }
    __syncthreads();  // (2)
    if (!(skip & (1 << 17))) rebuild_locmap(c);
    __syncthreads();  // (3)
    // nothing to do until the components are done but the next step's random actions
    SrcList src;
    src.S = 0;
#pragma unroll
    for (int k = 0; k < AIE_SRC_CAP / AIE_NT; ++k) src.d[k] = 0;
    src = SHL ? src_list_from_arena(c, arena) : src_list_from_record(c, arena);  // (the loads ride under the first wave's dynamics)
    // (round 6: six scalar touches here, for the parameter-block lines the two tails read, cost 1.2 us -- this wave is not
    // idle enough to absorb six misses served one at a time; removing a_records / a_metrics from the prologue and sharing
    // one argument line gave 0.2 - 0.3 us each)
#ifndef AIE_NO_TABLE_TOUCH  // (A/B builds)
    const uint32_t warm = glibc_tables_touch(c.tid);
#else
    const uint32_t warm = 0;
#endif
    // (tried in round 6: all ten rows with the record burst and the window from the registers -- one dependent round trip
    // and 768 redundant bytes less, but 1.7 KB more in the burst every workgroup starts with: C2 23.0 -> 23.5 us)
    if (!FAST && !(skip & (1 << 19))) {  // the generator's rows -> registers (re-read after the barrier if the components twisted the state)
#pragma unroll
      for (int j = 0; j < 9; ++j) m.r[j] = gkey[64 * j + c.tid];
      m.r[9] = c.tid < 48 ? gkey[576 + c.tid] : 0u;
    }
    if ((next.a || next.p) && TAIL) {
      const int per_env = P.n * P.act_a_width + P.act_p_width;
      const int st = uni(*R_I32(c, o_sample_t));  // (the first wave does not touch this field)
      for (int j = c.tid; j < per_env; j += AIE_NT) sample_action_slot(P, next.seed, next.env_offset, (int64_t)st, c.e, j, next.a, next.p);
      if (c.tid == 0) *R_I32(c, o_sample_t) = st + 1;
    }
    asm volatile("" ::"v"(warm));
    __syncthreads();  // (4)
    // resource regeneration (the generator's rows are in this wave's registers), then what depends on the map:
    // incremental map observations, action masks
#ifdef AIE_W1_TAIL_PRIO  // (A/B builds)
    __builtin_amdgcn_s_setprio(AIE_W1_TAIL_PRIO);
#else
    // A batch that fits the device in one go (4096 workgroups at 8 waves per SIMD) is a race to the last workgroup's end:
    // from here on this wave is the critical one (the first has slack) and goes ahead of the waves still loading.  A
    // larger batch is a stream of workgroups, bound by the vector pipes: there a raised tail only delays the next
    // workgroups' start (tools/ab2.sh, round 6: 16 384 replicas 74.5 -> 72.8 us without it, 65 536: 318 -> 314 us;
    // 4096: 22.5 -> 23.0 us WITHOUT it).
    if (next.E > 6144) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(2);
#endif
    if (FAST) {  // the stream's state is in the LDS image (the components may have moved it to the next block)
      const uint32_t* st = R_U32(c, o_mt);
      m.fkey = (uint32_t)uni((int)st[0]);
      m.fblk = (uint32_t)uni((int)st[1]);
      m.fsalt = (uint32_t)uni((int)st[2]);
    } else if (uni(c.dirty[3]) != 0) {  // a refill that twisted (components drew past word 623) left the new state in HBM
      mt_rows_from_hbm(m, gkey, c.tid);
    }
    m.pos = uni(*R_I32(c, o_mt_pos));
    if (TAIL) {
      if (!(skip & 2)) scenario_step_regen(c, m, src);
      if (c.tid == 0) {
        *R_I32(c, o_mt_pos) = m.pos;
        if (FAST) R_U32(c, o_mt)[1] = m.fblk;
      }
      if (!(skip & 32)) store_generator_rows(c, arena, m);  // (the rows' registers are free from here on)
    }
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 6] = wall_clock64();
    AIE_WSYNC();
    // (the masks follow the map observations' rule: in place unless something outside the kernels touched the state)
    const bool masks_all = OBSERVE || !uni(*R_I32(c, o_obs_valid)) || (skip & (4 | 32768)) != 0;
    if (!(skip & 4) && (TAIL || OBSERVE)) {
      // the map observations of the previous step are still in the arena: update them in place,
      // unless something outside the kernels touched the state (obs_valid == 0; an AIE_STEP_OBSERVE launch: always)
      if (uni(*R_I32(c, o_obs_valid)) && !(skip & 32768) && !OBSERVE) update_spatial_observations(c, arena);
      else write_spatial_observations(c, arena);
      if (c.tid == 0) *R_I32(c, o_obs_valid) = 1;
    } else if (!TAIL && !OBSERVE && c.tid == 0) {
      *R_I32(c, o_obs_valid) = 0;  // a partial step changed the state and wrote no observations: the launch that does starts over
    }
    if (!(skip & 8) && (TAIL || OBSERVE)) write_action_masks(c, arena, /*all=*/masks_all);
    if (TRACE && R.dev_trace && c.tid == 0) R.dev_trace[12 * blockIdx.x + 11] = wall_clock64();
    if (REW_ON_W1 && TAIL) step_rewards_and_done(c, arena, next, skip);
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();  // (5)
    if (!(skip & 32)) store_record_step(c, arena, 1, NW);
  }
}

#ifndef AIE_JIT  // (a run-time specialisation compiles the two entry points at the end of this file only)
extern "C" __global__ void __launch_bounds__(2 * AIE_NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
aie_step_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  step_body<2, false>(params, arena, act_a, act_p, lds, next);
}
// the same kernel for records whose LDS footprint keeps a CU at 12 workgroups or fewer anyway (aie_capi.hip:
// aie_step): 6 waves per SIMD buy 80 VGPRs, i.e. no scratch traffic
extern "C" __global__ void __launch_bounds__(2 * AIE_NT) __attribute__((amdgpu_waves_per_eu(6, 6)))
aie_step_kernel_r6(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                   const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  step_body<2, false>(params, arena, act_a, act_p, lds, next);
}
// the same step for environments with dense-log replicas (aie_config.dense_log_replicas > 0): records AIE_EV_* rows
extern "C" __global__ void __launch_bounds__(2 * AIE_NT) __attribute__((amdgpu_waves_per_eu(8, 8)))
aie_step_kernel_log(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                    const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  step_body<2, true>(params, arena, act_a, act_p, lds, next);
}
// compile-time instances for the configurations listed in ai-economist_amd/_specs.py (BASELINE configs[1], [2], ...)
#define AIE_SPEC_WAVES(S) aie_spec_image<S>::waves
template <int SPEC>
__global__ void __launch_bounds__(2 * AIE_NT)
__attribute__((amdgpu_waves_per_eu(AIE_SPEC_WAVES(SPEC), AIE_SPEC_WAVES(SPEC))))
aie_step_kernel_spec(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                     const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  step_body<2, false, SPEC>(params, arena, act_a, act_p, lds, next);
}
#ifdef AIE_DEV
// development: the compile-time instances with per-workgroup clock stamps (tools/block_trace.py, aie_dev_set_trace)
template <int SPEC>
__global__ void __launch_bounds__(2 * AIE_NT)
__attribute__((amdgpu_waves_per_eu(aie_spec_image<SPEC>::waves, aie_spec_image<SPEC>::waves)))
aie_step_kernel_spec_trace(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                           const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  step_body<2, false, SPEC, true>(params, arena, act_a, act_p, lds, next);
}
#endif
#endif  // !AIE_JIT

namespace aie {
// ------------------------------------------------------------------------------------------------------------------
// Reset-time source layouts (uniform/, quadrant/, multi_zone/): FOUR wavefronts per replica.
//
// The algorithm (dynamic_layout.py:313-392) is a chain of whole-plane passes -- rand plane, threshold search, growth by
// 7x7 random-kernel convolutions, coverage check, retry -- whose lengths depend on the draws, so one replica's
// generation cannot be cut shorter than its chain; a masked reset lasts as long as its slowest replica (82 of 4096
// replicas per reset at BASELINE configs[0]'s scenario: round 2's single wave took 0.26 ms on average and 1.2 ms for
// the slowest).  What shortens the chain is running every plane pass cell-parallel over LG_NW x 64 lanes:
//   * every wave keeps its own copy of the generator (three consecutive windows in registers) and advances it
//     identically, so no random word ever crosses a wave boundary;
//   * rand planes: 256 doubles per pass; legacy_gauss: 256 polar attempts per pass (an attempt's acceptance does not
//     depend on the others; the waves exchange their 64-bit accept masks through LDS, rank the accepted attempts with
//     a prefix over the masks and stop behind the attempt that completes the request, cache semantics included);
//   * the threshold search ("tmp *= 0.9 until enough cells pass") advances every cell eight iterations at a time in
//     registers and finds the stopping iteration from eight per-iteration counters: one barrier pair per eight
//     iterations instead of an LDS round trip + ballot per iteration;
//   * convolutions and coverage counts: one cell per lane (15x15: 225 of 256 lanes), counts meet in LDS.
// Results are bit-identical to the sequential restatement (oracle/aie_oracle.c: layout_generate), which the CPU tests
// pin to the live reference: every draw is consumed at the same stream offset, np.mean of a 0/1 plane is count / size,
// signal.convolve2d(x, kernel, "same") is one add per kernel one in scipy's order over the zero-filled 7x7 window.
// ------------------------------------------------------------------------------------------------------------------
#define LG_NW 4
#ifdef AIE_DEV  // per-replica phase clocks of layout_generate (tools/c1_reset_trace.py): dev_trace[12 e + k]
#define LG_T0() const uint64_t lg_t0_ = wall_clock64()
#define LG_ACC(k) do { lg_acc[k] += wall_clock64() - lg_t0_; } while (0)
#define LG_CNT(k) do { lg_acc[k] += 1; } while (0)
#else
#define LG_T0() do { } while (0)
#define LG_ACC(k) do { } while (0)
#define LG_CNT(k) do { } while (0)
#endif

// Lane-parallel reads of the stream: a lane wants the word at stream offset o (counted from the start of window 0,
// o < 3 * 624).  Window k + 1 is the twisted copy of window k (valid for k < have); consuming words moves `pos`
// and shifts the windows down when it passes 624.  Everything here is wave-uniform except `o`.
struct MT3 {
  MT w[3];
  int have;  // valid windows (>= 1)
  int pos;   // next unused word of window 0
};
__device__ __forceinline__ void mt3_need(MT3& s, int k, int lane) {  // make windows 0..k valid (k <= 2)
  if (s.have <= 1 && k >= 1) {
    mt_copy(s.w[1], s.w[0]);
    mt_twist(s.w[1], lane);
    s.have = 2;
  }
  if (s.have <= 2 && k >= 2) {
    mt_copy(s.w[2], s.w[1]);
    mt_twist(s.w[2], lane);
    s.have = 3;
  }
}
// raw word i of one window for the lanes in `want` (a ballot; the lanes' rows lie in [rlo, rhi], wave-uniform)
__device__ __forceinline__ uint32_t mt3_window_word(const MT& m, int i, uint32_t v, bool mine, int rlo, int rhi) {
  const int row = i >> 6, ln = i & 63;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    if (r >= rlo && r <= rhi) {  // (wave-uniform)
      const uint32_t t = lane_get(m.r[r], ln);
      v = (mine && row == r) ? t : v;
    }
  }
  return v;
}
// tempered word at offset o; o must be non-decreasing in the lane index (all call sites: o = base + stride * lane)
__device__ __forceinline__ uint32_t mt3_word(const MT3& s, int o) {
  const int win = o >= 2 * AIE_MT_N ? 2 : (o >= AIE_MT_N ? 1 : 0);
  const int i = o - win * AIE_MT_N;
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint64_t want = __ballot(win == k);
    if (want) {  // (wave-uniform) the first / last lane that reads this window bound the rows it touches
      const int first = __ffsll((unsigned long long)want) - 1, last = 63 - __clzll((long long)want);
      const int rlo = bcast(i, first) >> 6, rhi = bcast(i, last) >> 6;
      v = mt3_window_word(s.w[k], i, v, win == k, rlo, rhi);
    }
  }
  return mt_word(s.w[0], v);
}
// the four tempered words o .. o + 3 (one polar attempt): the window / row bookkeeping once for all four
__device__ __forceinline__ void mt3_words4(const MT3& s, int o, uint32_t out[4]) {
  uint32_t v[4] = {0, 0, 0, 0};
  int win[4], idx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    win[j] = o + j >= 2 * AIE_MT_N ? 2 : (o + j >= AIE_MT_N ? 1 : 0);
    idx[j] = o + j - win[j] * AIE_MT_N;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const uint64_t want = __ballot(win[0] == k || win[3] == k);  // (offsets are monotone in the lane and in j)
    if (want) {
      const int first = __ffsll((unsigned long long)want) - 1, last = 63 - __clzll((long long)want);
      // rows touched in window k: from the first lane's first word that lies in it to the last lane's last word
      const int lo_i = bcast(win[0], first) == k ? bcast(idx[0], first) : 0;
      const int hi_i = bcast(win[3], last) == k ? bcast(idx[3], last) : AIE_MT_N - 1;
      const int rlo = lo_i >> 6, rhi = hi_i >> 6;
#pragma unroll
      for (int r = 0; r < 10; ++r) {
        if (r >= rlo && r <= rhi) {  // (wave-uniform)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t t = lane_get(s.w[k].r[r], idx[j] & 63);
            v[j] = (win[j] == k && (idx[j] >> 6) == r) ? t : v[j];
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = mt_word(s.w[0], v[j]);
}
__device__ __forceinline__ void mt3_consume(MT3& s, int nwords, int lane) {  // nwords <= 2 * 624
  s.pos += nwords;
  while (s.pos >= AIE_MT_N) {
    mt3_need(s, 1, lane);
    mt_copy(s.w[0], s.w[1]);
    mt_copy(s.w[1], s.w[2]);
    s.have -= 1;
    s.pos -= AIE_MT_N;
  }
}

struct LgShared {      // cross-wave scratch in LDS (behind the planes)
  int32_t cnt[2][LG_NW];    // block_count: per-wave counts, two parities
  uint32_t am[2][LG_NW][2]; // gauss: per-wave accept masks, two parities
  int32_t thr[32];          // threshold search: cells on after iteration k of the current block
  uint32_t kbits[2];        // growth kernel: sign bits of the 49 randn values
  double lcache;            // legacy_gauss's one-value cache of a layout stream of its own (aie_layout_stream): the
  int32_t lhas, lpad_;      //   replica's own cache, a record field, belongs to the replica's stream
};
__host__ __device__ inline size_t layout_gen_lds_bytes(const aie_params& P) {
  if (P.c.layout_gen == AIE_LAYOUT_FIXED) return 0;
  const size_t hwp = ((size_t)P.HW + 15) / 16 * 16;
  // tmp, x (f64), two byte planes, the multi_zone grid, the cross-wave scratch
  return 2 * hwp * 8 + 2 * hwp + 256 * 4 + ((sizeof(LgShared) + 15) / 16 * 16);
}
__device__ __forceinline__ LgShared* lg_shared(uint8_t* extra, const aie_params& P) {  // (layout_generate's carve-up of `extra`)
  const size_t hwp = ((size_t)P.HW + 15) / 16 * 16;
  return reinterpret_cast<LgShared*>(extra + 2 * hwp * 8 + 2 * hwp + 256 * 4);
}
// number of lanes of the whole workgroup for which `on` holds (every wave calls it; one barrier)
__device__ __forceinline__ int lg_block_count(bool on, LgShared* sh, int& par, int wave, int lane) {
  const int c = __popcll(__ballot(on));
  if (lane == 0) sh->cnt[par][wave] = c;
  __syncthreads();
  int tot = 0;
#pragma unroll
  for (int w = 0; w < LG_NW; ++w) tot += sh->cnt[par][w];
  par ^= 1;
  return uni(tot);
}
// position (0-based) of the k-th set bit (k >= 1) of a wave-uniform mask, computed by the lanes
__device__ __forceinline__ int lg_kth_bit(uint64_t mask, int k, int lane) {
  const int before = __popcll(mask & lanemask_lt(lane));
  const uint64_t hit = __ballot(((mask >> lane) & 1ull) && before == k - 1);
  return __ffsll((unsigned long long)hit) - 1;
}

// The next `count` values of legacy_gauss (polar Box-Muller with a one-value cache, as rng_gauss): emit(k, value) is
// called once for k = 0 .. count - 1 by the lane that owns the value.  LG_NW x 64 attempts per pass: attempt a reads
// the four words at pos + 4 a.  Called by all waves; contains barriers.
template <typename Emit>
__device__ __forceinline__ void lg_gauss(MT3& s, int count, LgShared* sh, int& par, int wave, int lane, int32_t* has,
                                         double* cache, Emit emit) {
  int produced = 0;
  const bool cached = count > 0 && uni(*has) != 0;
  __syncthreads();  // every wave has looked at the cache flag before anybody changes it
  if (cached) {
    if (wave == 0 && lane == 0) {
      emit(0, *cache);
      *has = 0;
      *cache = 0.0;
    }
    produced = 1;
  }
  while (produced < count) {
    const int pairs = (count - produced + 1) >> 1;  // accepted attempts still needed
    mt3_need(s, (s.pos + 4 * LG_NW * AIE_NT - 1) / AIE_MT_N, lane);
    const int a = wave * AIE_NT + lane, o = s.pos + 4 * a;
    uint32_t w4[4];
    mt3_words4(s, o, w4);
    const double x1 = 2.0 * u53(w4[0], w4[1]) - 1.0;
    const double x2 = 2.0 * u53(w4[2], w4[3]) - 1.0;
    const double r2 = x1 * x1 + x2 * x2;
    const bool acc = !(r2 >= 1.0 || r2 == 0.0);
    const uint64_t am = __ballot(acc);
    if (lane == 0) {
      sh->am[par][wave][0] = (uint32_t)am;
      sh->am[par][wave][1] = (uint32_t)(am >> 32);
    }
    __syncthreads();
    int before_wave = 0, total = 0, used = LG_NW * AIE_NT;  // attempts consumed by this pass
    bool found = false;
#pragma unroll
    for (int w = 0; w < LG_NW; ++w) {
      const uint64_t mw = (uint64_t)sh->am[par][w][0] | ((uint64_t)sh->am[par][w][1] << 32);
      const int cw = __popcll(mw);
      if (w == wave) before_wave = total;
      if (!found && total + cw >= pairs) {  // the attempt holding the pairs-th accepted one ends the request
        used = w * AIE_NT + lg_kth_bit(mw, pairs - total, lane) + 1;
        found = true;
      }
      total += cw;
    }
    par ^= 1;
    if (acc && a < used) {
      const double f = sqrt(-2.0 * aie_log_glibc(r2) / r2);  // libm's log bit for bit; sqrt and / are IEEE-exact
      const int k = produced + 2 * (before_wave + __popcll(am & lanemask_lt(lane)));
      emit(k, f * x2);
      if (k + 1 < count) emit(k + 1, f * x1);
      else {  // the request ends on the first value of the pair: the second one stays cached
        *cache = f * x1;
        *has = 1;
      }
    }
    produced += 2 * (found ? pairs : total);  // (may exceed count by one: the cached value)
    mt3_consume(s, 4 * used, lane);
  }
  __syncthreads();  // emitted values / cache visible to every wave
}

// Source layouts drawn at reset from the replica's own stream: Uniform.reset_starting_layout (dynamic_layout.py:313-392),
// MultiZone's per-reset zone shuffle (:778-872), Quadrant's empty water lines (:992-1024).  Called by all LG_NW waves
// of the replica's workgroup (gtid = 0 .. LG_NW * 64 - 1); `m` is every wave's copy of the generator, advanced
// identically.  c.tid is the lane.
// (cell flags of the record image <- the two source planes, with the checker / water-line cuts of the scenario)
template <typename Planes>
__device__ __forceinline__ void layout_install(const Ctx& c, int gtid, int nthreads, Planes planes) {
  const aie_params& P = c.P;
  const aie_config& g = P.c;
  const int H = P.H, W = P.W, HW = P.HW;
  uint8_t* cb = reinterpret_cast<uint8_t*>(R_CELLS(c));
  for (int cell = gtid; cell < HW; cell += nthreads) {
    const int r = cell / W, col = cell - r * W;
    const uint32_t bits = planes(cell);  // bit 0 Stone, bit 1 Wood
    bool st = (bits & 1u) != 0, wd = (bits & 2u) != 0;
    if (g.layout_checker && ((r & 1) + (col & 1)) != 1) st = wd = false;
    if (g.layout_gen == AIE_LAYOUT_QUADRANT && (col == H / 2 || r == W / 2)) st = wd = false;  // nothing on the water lines
    cb[4 * cell + 3] = (uint8_t)((cb[4 * cell + 3] & AIE_CELL_WATER) | (st ? AIE_CELL_STONE_SRC : 0) | (wd ? AIE_CELL_WOOD_SRC : 0));
  }
}
// `ghas` / `gcache`: legacy_gauss's cache of the stream the layout is drawn from (the record's for the replica's own
// stream, LgShared's for a layout stream).  `stage_out` != nullptr: the planes go to the staging area (one byte per cell:
// bit 0 Stone, bit 1 Wood) instead of into the record image's cell flags.
__device__ __forceinline__ void layout_generate(const Ctx& c, MT& m, const uint8_t* __restrict__ arena, uint8_t* extra, int gtid,
                                                int32_t* ghas, double* gcache, uint8_t* __restrict__ stage_out = nullptr) {
  const aie_params& P = c.P;
  const aie_config& g = P.c;
  const int H = P.H, W = P.W, HW = P.HW, lane = gtid & 63, wave = gtid >> 6;
  constexpr int NT = LG_NW * AIE_NT;
  const int hwp = (HW + 15) / 16 * 16;
  double* tmp = reinterpret_cast<double*>(extra);
  double* x = tmp + hwp;
  uint8_t* mbp[2] = {reinterpret_cast<uint8_t*>(x + hwp), reinterpret_cast<uint8_t*>(x + hwp) + hwp};
  int32_t* grid = reinterpret_cast<int32_t*>(mbp[1] + hwp);
  LgShared* sh = reinterpret_cast<LgShared*>(grid + 256);
  int par = 0, gpar = 0;
  const double* shared_prob = reinterpret_cast<const double*>(arena + c.R.a_layout_prob);
  const bool mz = g.layout_gen == AIE_LAYOUT_MULTI_ZONE;
  double mz_scale[2] = {0.0, 0.0};
  int size_r = 1, size_c = 1;
  if (mz) {  // np.random.shuffle of the flat zone grid, then prob / np.mean(prob) * Wood's coverage
    const int regions = g.mz_rows * g.mz_cols;
    for (int k = gtid; k < regions; k += NT) {
      int z = -1;
      if (k < g.mz_zones[0]) z = 0;
      else if (k < g.mz_zones[0] + g.mz_zones[1]) z = 1;
      else if (k < g.mz_zones[0] + g.mz_zones[1] + g.mz_zones[2]) z = 2;
      grid[k] = z;
    }
    __syncthreads();
    for (int i = regions - 1; i >= 1; --i) {  // every wave draws; the first one swaps (nobody else reads the grid yet)
      const int j = (int)rng_interval(m, lane, (uint32_t)i);
      if (gtid == 0) { const int t = grid[i]; grid[i] = grid[j]; grid[j] = t; }
      AIE_WSYNC();
    }
    __syncthreads();
    size_r = (H + g.mz_rows - 1) / g.mz_rows;
    size_c = (W + g.mz_cols - 1) / g.mz_cols;
    for (int rs = 0; rs < 2; ++rs) {
      const int own = rs == 1 ? 0 : 1;  // zone index: Wood 0, Stone 1, WoodStone 2
      int cnt = 0;
      for (int base = 0; base < HW; base += NT) {
        const int cell = base + gtid;
        bool in = false;
        if (cell < HW) {
          const int r = cell / W, col = cell - r * W;
          const int z = grid[(r / size_r) * g.mz_cols + col / size_c];
          in = z == own || z == 2;
        }
        cnt += lg_block_count(in, sh, par, wave, lane);
      }
      mz_scale[rs] = (1.0 / ((double)cnt / (double)HW)) * c.R.c.layout_coverage[1];
    }
  }
  auto source_prob = [&](int rs, int cell, double clump) -> double {
    double v;
    if (mz) {
      const int r = cell / W, col = cell - r * W;
      const int z = grid[(r / size_r) * g.mz_cols + col / size_c];
      v = (z == (rs == 1 ? 0 : 1) || z == 2) ? mz_scale[rs] : 0.0;
    } else {
      v = shared_prob[rs * HW + cell];
    }
    return v * 0.1 * clump;
  };
#ifdef AIE_DEV
  uint64_t lg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t lg_start = wall_clock64();
#endif
  MT3 s3;
  mt_copy(s3.w[0], m);
  s3.pos = m.pos;
  s3.have = 1;  // (pos == 624, a freshly seeded generator: every word then comes from window 1, the first twist)
  bool happy = false;
  for (int tries = 0; tries < 100 && !happy; ++tries) {
    LG_CNT(5);
    for (int q = 0; q < 2; ++q) {
      const int rs = q == 0 ? 1 : 0;  // ["Wood", "Stone"]
      const double cov = c.R.c.layout_coverage[rs], clump = c.R.c.layout_clump[rs];
      uint8_t* mb = mbp[rs];
      const uint8_t* other = q == 0 ? nullptr : mbp[1];  // empty = nothing placed on the tile yet
      // tmp = rs.rand(H, W): NT doubles per pass, thread t takes words pos + 2 t, + 1
      { LG_T0();
      for (int base = 0; base < HW; base += NT) {
        const int nb = HW - base < NT ? HW - base : NT;
        mt3_need(s3, (s3.pos + 2 * nb - 1) / AIE_MT_N, lane);
        const int o = s3.pos + 2 * (gtid < nb ? gtid : nb - 1);  // (idle threads repeat the last one's words: offsets stay monotone)
        const uint32_t a = mt3_word(s3, o), b = mt3_word(s3, o + 1);
        if (gtid < nb) tmp[base + gtid] = u53(a, b);
        mt3_consume(s3, 2 * nb, lane);
      }
      __syncthreads();
      LG_ACC(1); }
      int count = 0;
      for (int base = 0; base < HW; base += NT) {
        const int cell = base + gtid;
        bool on = false;
        if (cell < HW) {
          on = (tmp[cell] < source_prob(rs, cell, clump)) && !(other && other[cell]);
          mb[cell] = on ? 1 : 0;
        }
        count += lg_block_count(on, sh, par, wave, lane);
      }
      // while np.mean(maybe) < coverage * clump: tmp *= 0.9; maybe = (tmp < prob) * empty; stop after 201 rounds --
      // eight rounds at a time: a cell's eight outcomes become a bit mask (parked in its `maybe` byte), the rounds'
      // counts meet in sh->thr, and the first round that reaches the target is the one the loop would have stopped at
      int n_tries = 0;
      if (HW <= NT) {  // one cell per thread: the cell's value, threshold and "tile still free" stay in registers, 32 rounds a block
        const int cell = gtid < HW ? gtid : HW - 1;
        double t = tmp[cell];
        const double sp = source_prob(rs, cell, clump);
        const bool free_tile = gtid < HW && !(other && other[cell]);
        while ((double)count / (double)HW < cov * clump && n_tries <= 200) {
          LG_T0();
          LG_CNT(7);
          const int B = 201 - n_tries < 32 ? 201 - n_tries : 32;
          if (gtid < 32) sh->thr[gtid] = 0;
          __syncthreads();
          // tmp only shrinks, so a cell that passes stays on: its rounds are summed up by the FIRST round it passes
          // in (32 = not in this block); the rounds' counts are the prefix sums of that histogram
          int first_on = 32;
          {
            double tk = t;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
              tk = tk * 0.9;
              first_on = (first_on == 32 && (tk < sp) && k < B) ? k : first_on;
            }
            t = tk;
          }
          if (!free_tile) first_on = 32;
          if (first_on < 32) atomicAdd(&sh->thr[first_on], 1);
          __syncthreads();
          int mine = lane < 32 ? sh->thr[lane] : 0;  // lane k: cells whose first round is k ...
#pragma unroll
          for (int d = 1; d < 32; d <<= 1) {         // ... inclusive prefix over lanes 0..31: cells on after round k
            const int up = __shfl_up(mine, d, 64);
            mine += lane >= d ? up : 0;
          }
          const uint64_t reached = __ballot(lane < B && !((double)mine / (double)HW < cov * clump));
          const int stop = reached ? __ffsll((unsigned long long)reached) - 1 : B - 1;  // the earliest round that reaches the target
          count = bcast(mine, stop);
          const uint32_t bits = first_on <= stop ? ~0u : 0u;
          n_tries += stop + 1;
          if (gtid < HW) mb[gtid] = (bits >> stop) & 1u;
          __syncthreads();
          LG_ACC(2);
        }
      } else
      while ((double)count / (double)HW < cov * clump && n_tries <= 200) {
        LG_T0();
        LG_CNT(7);
        const int B = 201 - n_tries < 8 ? 201 - n_tries : 8;
        if (gtid < 8) sh->thr[gtid] = 0;
        __syncthreads();
        for (int base = 0; base < HW; base += NT) {
          const int cell = base + gtid;
          uint32_t bits = 0;
          if (cell < HW) {
            double t = tmp[cell];
            const double sp = source_prob(rs, cell, clump);
            const bool free_tile = !(other && other[cell]);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              t = t * 0.9;
              bits |= ((t < sp) && free_tile && k < B) ? (1u << k) : 0u;
            }
            tmp[cell] = t;  // (rounds past the stopping one included: tmp is dead once the search ends)
            mb[cell] = (uint8_t)bits;
          }
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int ck = __popcll(__ballot((bits >> k) & 1u));
            if (lane == 0 && ck) atomicAdd(&sh->thr[k], ck);
          }
        }
        __syncthreads();
        int stop = B - 1;  // the round whose outcome stands
        for (int k = B - 1; k >= 0; --k)
          if (!((double)sh->thr[k] / (double)HW < cov * clump)) stop = k;  // the earliest round that reaches the target
        stop = uni(stop);
        count = uni(sh->thr[stop]);
        n_tries += stop + 1;
        for (int base = 0; base < HW; base += NT) {
          const int cell = base + gtid;
          if (cell < HW) mb[cell] = (mb[cell] >> stop) & 1u;
        }
        __syncthreads();
        LG_ACC(2);
      }
      while ((double)count / (double)HW < cov) {
        // kernel = rs.randn(7, 7) > 0 (row-major), then maybe + 0.2 * rs.randn(H, W) - 0.25: one request of 49 + HW values
        LG_CNT(6);
        if (gtid < 2) sh->kbits[gtid] = 0;
        { LG_T0();
        lg_gauss(s3, 49 + HW, sh, gpar, wave, lane, ghas, gcache, [&](int k, double gs) {
          if (k < 49) {
            if (gs > 0) atomicOr(&sh->kbits[k >> 5], 1u << (k & 31));
          } else {
            const int cell = k - 49;
            x[cell] = ((double)mb[cell] + (0.2 * gs)) - 0.25;
          }
        });
        LG_ACC(3); }
        LG_T0();
        const uint64_t kmask = (uint64_t)sh->kbits[0] | ((uint64_t)sh->kbits[1] << 32);
        count = 0;
        for (int base = 0; base < HW; base += NT) {
          const int cell = base + gtid;
          bool on = false;
          if (cell < HW) {
            const int r0 = cell / W, c0 = cell - r0 * W;
            // the 7x7 window first (49 LDS reads in flight at once: the additions below are a dependent chain, the
            // reads need not be), then one add per kernel one in scipy's order; window cells outside the world add nothing
            double win[49];
            uint32_t rowok = 0, colok = 0;
#pragma unroll
            for (int j = 0; j < 7; ++j) {
              rowok |= (r0 + 3 - j >= 0 && r0 + 3 - j < H) ? (1u << j) : 0u;
              colok |= (c0 + 3 - j >= 0 && c0 + 3 - j < W) ? (1u << j) : 0u;
            }
#pragma unroll
            for (int t = 0; t < 49; ++t) {
              const int j = t / 7, k = t - 7 * j;
              const bool inb = ((rowok >> j) & 1u) && ((colok >> k) & 1u);
              win[t] = inb ? x[cell + (3 - j) * W + (3 - k)] : 0.0;
            }
            double sum = 0.0;
#pragma unroll
            for (int t = 0; t < 49; ++t) {
              const int j = t / 7, k = t - 7 * j;
              if ((kmask >> t) & 1ull) {  // (wave-uniform)
                const bool inb = ((rowok >> j) & 1u) && ((colok >> k) & 1u);
                sum = inb ? sum + win[t] : sum;
              }
            }
            on = ((sum > 0) || mb[cell]) && !(other && other[cell]);
            mb[cell] = on ? 1 : 0;  // (a cell's new value depends on x and its own old value only; the count's barrier publishes it)
          }
          count += lg_block_count(on, sh, par, wave, lane);
        }
        LG_ACC(4);
      }
    }
    happy = true;
    for (int q = 0; q < 2; ++q) {
      const int rs = q == 0 ? 1 : 0;
      int count = 0;
      for (int base = 0; base < HW; base += NT) {
        const int cell = base + gtid;
        count += lg_block_count(cell < HW && mbp[rs][cell], sh, par, wave, lane);
      }
      const double ratio = ((double)count / (double)HW) / c.R.c.layout_coverage[rs];
      if (!((1 / 1.4) <= ratio && ratio <= 1.4)) happy = false;
    }
  }
  mt_copy(m, s3.w[0]);
  m.pos = s3.pos;
  if (stage_out) {
    for (int cell = gtid; cell < HW; cell += NT) stage_out[cell] = (uint8_t)((mbp[0][cell] ? 1u : 0u) | (mbp[1][cell] ? 2u : 0u));
  } else {
    layout_install(c, gtid, NT, [&](int cell) -> uint32_t { return (mbp[0][cell] ? 1u : 0u) | (mbp[1][cell] ? 2u : 0u); });
  }
  __syncthreads();
#ifdef AIE_DEV
  if (c.R.dev_trace && gtid == 0) {
    lg_acc[0] = wall_clock64() - lg_start;
    for (int k = 0; k < 8; ++k) c.R.dev_trace[12 * c.e + k] = lg_acc[k];
  }
#endif
}
}  // namespace aie

// BaseEnvironment.reset, F/base/base_env.py:852-927, with LayoutFromFile
// reset_starting_layout / reset_agent_states / additional_reset_steps
// (layout_from_file.py:323-370, 564-593) and the component resets (build.py:224-254,
// move.py:193-210, continuous_double_auction.py:643-668, redistribution.py:1109-1139).
// Runs once per episode: the sequential part is executed wave-uniformly out of LDS.
namespace aie {
template <int SPEC, bool LAYOUT = true>  // LAYOUT == false: compiled without the layout generator (fixed layouts only)
__device__ __forceinline__ void reset_body(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                                           const uint8_t* __restrict__ mask, int keep_rewards, uint8_t* lds) {
  const aie_params& R = *params;                      // run-time block: replica count, arena offsets
  const aie_params& P = aie_spec_params<SPEC>(params);  // the configuration: a constant image in the instances
  const int e = replica_of_block((int)blockIdx.x, R.E);
  if (mask && !mask[e]) return;
  // One wavefront resets a replica; environments whose reset draws a new source layout (uniform/, quadrant/,
  // multi_zone/) are launched with LG_NW wavefronts per replica: all of them generate the layout, then the first
  // one carries on alone (a barrier only waits for the waves of a workgroup that are still running).
  const int gtid = (int)threadIdx.x, wave = gtid >> 6, nwaves = (int)blockDim.x >> 6;
  const Ctx c = make_ctx(P, R, lds, e, gtid & 63, arena);
  const int n = P.n, HW = P.HW, tid = c.tid;
  if (wave == 0) {
    for (int q = tid; q < (P.met_bytes >> 2); q += AIE_NT) reinterpret_cast<uint32_t*>(C_MET(c))[q] = 0u;  // new episode
    if (c.ev && tid == 0) c.ev[0] = 0;
  }
  MT m;
  mt_init(m, P);
  load_record(c, arena, m, wave, nwaves, wave);  // every wave takes its own copy of the generator's rows
  __syncthreads();
  m.pos = uni(*R_I32(c, o_mt_pos));
  mt_fast_attach(c, m);
  if (LAYOUT && P.c.layout_gen != AIE_LAYOUT_FIXED) {  // a fresh source layout, drawn before anything else of the reset
    if (aie__layout_staged(&P.c)) {
      // the counter-stream mode: this reset's layout is a function of (the replica's stream, resets so far) alone
      // (aie_layout.h: aie_layout_stream) -- already in the staging area if a refill launch came by since the last
      // reset, else drawn here from the same stream; the replica's own stream is not touched either way
      uint32_t st[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) st[k] = (uint32_t)uni((int)R_U32(c, o_mt)[k]);
      const uint64_t tag = aie_layout_tag(st);
      const uint64_t* tags = reinterpret_cast<const uint64_t*>(arena + R.a_layout_tag);
      int32_t* ctl = reinterpret_cast<int32_t*>(arena + R.a_layout_ctl);
      const uint64_t have = tags[e];
      const bool staged = (uint32_t)uni((int)(uint32_t)have) == (uint32_t)tag && (uint32_t)uni((int)(uint32_t)(have >> 32)) == (uint32_t)(tag >> 32);
      __syncthreads();  // (every wave has read the state words before the first one counts the reset)
      if (staged) {
        const uint8_t* sp = arena + R.a_layout_stage + (int64_t)e * R.layout_stage_stride;
        layout_install(c, gtid, LG_NW * AIE_NT, [&](int cell) -> uint32_t { return sp[cell]; });
        __syncthreads();
      } else {
        LgShared* sh = lg_shared(lds + lds_bytes(P), P);
        if (gtid == 0) {
          sh->lhas = 0;
          sh->lcache = 0.0;
        }
        __syncthreads();
        uint32_t ks[2];
        aie_layout_stream(st, ks);
        MT ml;
        mt_init(ml, P);
        ml.fkey = ks[0];
        ml.fblk = 0xffffffffu;
        ml.fsalt = ks[1];
        layout_generate(c, ml, arena, lds + lds_bytes(P), gtid, &sh->lhas, &sh->lcache);
      }
      if (gtid == 0) {
        R_U32(c, o_mt)[3] = st[3] + 1u;  // resets so far
        atomicAdd(&ctl[0], 1);           // one more replica without a staged layout for its coming reset
        atomicAdd(&ctl[staged ? 2 : 3], 1);
      }
    } else {
      layout_generate(c, m, arena, lds + lds_bytes(P), gtid, R_I32(c, o_mt_has_gauss), R_F64(c, o_mt_gauss));
    }
    if (wave != 0) return;
  }
  {  // layout_from_file.py:323-334: resources back on every source block, no houses
    uint32_t* cells = R_CELLS(c);
    for (int q = tid; q < HW; q += AIE_NT) {
      const uint32_t fl = cells[q] >> 24;
      cells[q] = AIE_CELL_PACK((fl & AIE_CELL_STONE_SRC) ? 1 : 0, (fl & AIE_CELL_WOOD_SRC) ? 1 : 0, 0xff, fl);
    }
    if (tid < n) {
      R_I32(c, o_inv_res)[tid] = 0; R_I32(c, o_inv_res)[n + tid] = 0;
      R_I32(c, o_esc_res)[tid] = 0; R_I32(c, o_esc_res)[n + tid] = 0;
      R_F64(c, o_inv_coin)[tid] = c.R.c.starting_agent_coin;
      R_F64(c, o_esc_coin)[tid] = 0;
      R_F64(c, o_labor)[tid] = 0;
      R_I32(c, o_loc_r)[tid] = -1;
      R_I32(c, o_loc_c)[tid] = -1;
      if (!P.has_build) { R_F64(c, o_build_payment)[tid] = 0; R_F64(c, o_build_skill)[tid] = 0; }
      if (!P.has_gather) R_F64(c, o_bonus_gather_prob)[tid] = 0;
    }
    if (P.has_cda) {
      for (int q = tid; q < 2 * P.M; q += AIE_NT) { R_I32(c, o_cda_bids)[q] = 0; R_I32(c, o_cda_asks)[q] = 0; }
      for (int q = tid; q < 2 * n * P.P; q += AIE_NT) {
        R_U8(c, o_cda_bid_hist)[q] = 0; R_U8(c, o_cda_ask_hist)[q] = 0;
        R_F64(c, o_cda_price_history)[q] = 0;
      }
      for (int q = tid; q < 2 * n; q += AIE_NT) R_I32(c, o_cda_n_orders)[q] = 0;
      if (tid < 2) { R_I32(c, o_cda_n_bids)[tid] = 0; R_I32(c, o_cda_n_asks)[tid] = 0; }
    }
  }
  __syncthreads();
  build_src_list(c, arena);  // the regeneration's source doubles of this episode's layout
  if (P.regen_conv) {  // source blocks per d x d window, zero-padded ("same"), once per episode
    const uint8_t* cb = reinterpret_cast<const uint8_t*>(R_CELLS(c));
    for (int q = tid; q < AIE_N_RES * HW; q += AIE_NT) {
      const int rs = q >= HW ? 1 : 0, cell = q - rs * HW, r0 = cell / P.W, c0 = cell - r0 * P.W;
      const int hw = P.c.regen_halfwidth[rs];
      const uint32_t bit = rs ? AIE_CELL_WOOD_SRC : AIE_CELL_STONE_SRC;
      int cnt = 0;
      for (int r = max(r0 - hw, 0); r <= min(r0 + hw, P.H - 1); ++r)
        for (int cc = max(c0 - hw, 0); cc <= min(c0 + hw, P.W - 1); ++cc)
          cnt += (cb[4 * (r * P.W + cc) + 3] & bit) ? 1 : 0;
      R_U8(c, o_regen_count)[q] = (uint8_t)cnt;
    }
  }
  rebuild_locmap(c);  // all agents off the board
  // ---- wave-uniform sequential part (every lane performs the same LDS updates) ----
  *R_I32(c, o_timestep) = 0;
  *R_I32(c, o_error_flags) = 0;
  // layout_from_file.py:360-370 places agents in index order, dynamic_layout.py:420-431
  // (uniform/...) in a random order
  const int place_perm = P.c.reset_random_order ? rng_permutation(m, tid, n) : tid;
  for (int k = 0; k < n; ++k) {
    const int i = bcast(place_perm, k);
    int r = (int)rng_interval(m, tid, (uint32_t)(P.H - 1)), col = (int)rng_interval(m, tid, (uint32_t)(P.W - 1)), tries = 0;
    while (!can_agent_occupy(c, r, col, i)) {
      r = (int)rng_interval(m, tid, (uint32_t)(P.H - 1));
      col = (int)rng_interval(m, tid, (uint32_t)(P.W - 1));
      if (++tries > 200) {  // the reference raises TimeoutError (layout_from_file.py:366-368): flagged, see AIE_ERR_*
        *R_I32(c, o_error_flags) |= AIE_ERR_RESET_PLACEMENT;
        break;
      }
    }
    R_I32(c, o_loc_r)[i] = r;
    R_I32(c, o_loc_c)[i] = col;
    c.locmap[r * P.W + col] = (uint8_t)(i + 1);
  }
  for (int k = 0; k < P.c.n_components; ++k) {
    switch (P.c.components[k]) {
      case AIE_COMP_BUILD:
        for (int i = 0; i < n; ++i) {
          double skill = 1, pay = 1;
          const double pm = (double)c.R.c.build_payment_max_skill_multiplier;
          if (P.c.build_skill_dist == AIE_SKILL_PARETO) {
            skill = rng_pareto(m, tid, 4.0);
            pay = (pm - 1) * skill + 1; if (pm < pay) pay = pm;
          } else if (P.c.build_skill_dist == AIE_SKILL_LOGNORMAL) {
            skill = rng_lognormal(c, m, -1.0, 0.5);
            pay = (pm - 1) * skill + 1; if (pm < pay) pay = pm;
          }
          R_F64(c, o_build_payment)[i] = pay * (double)c.R.c.build_payment;
          R_F64(c, o_build_skill)[i] = skill;
        }
        break;
      case AIE_COMP_GATHER:
        for (int i = 0; i < n; ++i) {
          double b = 0.0;
          if (P.c.gather_skill_dist == AIE_SKILL_PARETO) { b = rng_pareto(m, tid, 3.0); b = (b < 2 ? b : 2) / 2; }
          else if (P.c.gather_skill_dist == AIE_SKILL_LOGNORMAL) { b = rng_lognormal(c, m, -2.022, 0.938); b = (b < 2 ? b : 2) / 2; }
          R_F64(c, o_bonus_gather_prob)[i] = b;
        }
        break;
      case AIE_COMP_TAX:
        for (int b = 0; b < P.NB; ++b) R_I32(c, o_tax_rate_idx)[b] = 0;
        *R_I32(c, o_tax_cycle_pos) = 1;
        for (int i = 0; i < n; ++i) {
          R_F64(c, o_tax_last_coin)[i] = R_F64(c, o_inv_coin)[i] + R_F64(c, o_esc_coin)[i];
          R_F64(c, o_tax_last_income)[i] = 0;
          R_F64(c, o_tax_last_marginal_rate)[i] = 0;
        }
        *R_F64(c, o_tax_total_collected) = 0;
        if (P.c.tax_model == AIE_TAX_SAEZ) {
          // _curr_rates_obs first (:1123, still the previous episode's rates), then
          // curr_bracket_tax_rates = running_avg_tax_rates (:1136-1137)
          for (int b = 0; b < P.NB; ++b) R_F64(c, o_tax_saez_obs_rates)[b] = tax_rate(c, b);
          AIE_WSYNC();
          for (int b = 0; b < P.NB; ++b)
            R_F64(c, o_tax_saez_rates)[b] = reinterpret_cast<const double*>(saez_block(c) + AIE_SAEZ_OFF_AVG)[b];
        }
        break;
      default: break;
    }
  }
  if (P.c.fixed_four_skill_and_loc) {  // layout_from_file.py:582-586
    for (int i = 0; i < n; ++i) {
      c.locmap[R_I32(c, o_loc_r)[i] * P.W + R_I32(c, o_loc_c)[i]] = 0;
      R_I32(c, o_loc_r)[i] = -1;
      R_I32(c, o_loc_c)[i] = -1;
    }
    const int perm = rng_permutation(m, tid, n);
    for (int k = 0; k < n; ++k) {
      const int i = bcast(perm, k);
      const int r = c.R.c.ranked_locs[k][0], col = c.R.c.ranked_locs[k][1];
      if (can_agent_occupy(c, r, col, i)) {
        R_I32(c, o_loc_r)[i] = r;
        R_I32(c, o_loc_c)[i] = col;
        c.locmap[r * P.W + col] = (uint8_t)(i + 1);
      }
      R_F64(c, o_build_payment)[i] = c.R.c.avg_ranked_skill[k];
    }
  }
  if (P.c.split_water_line > 0) {  // SplitLayout.additional_reset_steps layout_from_file.py:759-793
    for (int i = 0; i < n; ++i) {
      c.locmap[R_I32(c, o_loc_r)[i] * P.W + R_I32(c, o_loc_c)[i]] = 0;
      R_I32(c, o_loc_r)[i] = -1;
      R_I32(c, o_loc_c)[i] = -1;
    }
    const int perm = rng_permutation(m, tid, n);
    const int wl = P.c.split_water_line;
    for (int k = 0; k < n; ++k) {
      const int i = bcast(perm, k);
      R_F64(c, o_build_payment)[i] = c.R.c.avg_ranked_skill[k];
      const bool top = (c.R.c.split_top_ranks[k >> 5] >> (k & 31)) & 1u;
      const int r_min = top ? 0 : wl + 1, r_max = top ? wl : P.H;
      int r = r_min + (int)rng_interval(m, tid, (uint32_t)(r_max - r_min - 1));
      int col = (int)rng_interval(m, tid, (uint32_t)(P.W - 1)), tries = 0;
      while (!can_agent_occupy(c, r, col, i)) {
        r = r_min + (int)rng_interval(m, tid, (uint32_t)(r_max - r_min - 1));
        col = (int)rng_interval(m, tid, (uint32_t)(P.W - 1));
        if (++tries > 200) {  // the reference raises TimeoutError
          *R_I32(c, o_error_flags) |= AIE_ERR_RESET_PLACEMENT;
          break;
        }
      }
      R_I32(c, o_loc_r)[i] = r;
      R_I32(c, o_loc_c)[i] = col;
      c.locmap[r * P.W + col] = (uint8_t)(i + 1);
    }
  }
  *R_I32(c, o_mt_pos) = m.pos;
  mt_fast_detach(c, m);
  __syncthreads();
  current_metrics(c);
  __syncthreads();
  if (tid <= n) R_F64(c, o_util)[tid] = scr_part(c)[tid];
  __syncthreads();
  write_spatial_observations(c, arena);
  if (tid == 0) *R_I32(c, o_obs_valid) = 1;
  write_flat_observations(c, arena);
  AIE_WSYNC();
  if (P.has_tax && P.c.tax_annealing && tid == 0) *R_I32(c, o_tax_last_completions) = *R_I32(c, o_completions);  // generate_masks :1036-1046
  AIE_WSYNC();
  write_action_masks(c, arena);
  if (!keep_rewards) {  // (auto-reset right behind the step that ended the episode: its rewards / done stay)
    if (tid < n) reinterpret_cast<float*>(arena + R.a_rew_a)[(int64_t)e * n + tid] = 0.0f;
    if (tid == 0) {
      reinterpret_cast<float*>(arena + R.a_rew_p)[e] = 0.0f;
      (arena + R.a_done)[e] = 0;
    }
  }
  __syncthreads();
  store_record(c, arena, m);
}
}  // namespace aie

#ifndef AIE_JIT
extern "C" __global__ void __launch_bounds__(AIE_NT)
aie_reset_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                 const uint8_t* __restrict__ mask, int keep_rewards) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  aie::reset_body<-1, false>(params, arena, mask, keep_rewards, lds);
}
// environments whose reset draws a new source layout (uniform/, quadrant/, multi_zone/): LG_NW wavefronts per replica
extern "C" __global__ void __launch_bounds__(LG_NW * AIE_NT)
aie_reset_kernel_layout(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                        const uint8_t* __restrict__ mask, int keep_rewards) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  aie::reset_body<-1, true>(params, arena, mask, keep_rewards, lds);
}
// Generated layouts in the counter-stream mode, ahead of their resets (aie_layout.h: a_layout_stage).  Behind every reset
// launch: one thread decides whether a refill pays -- a refill launch lasts as long as its slowest replica (0.3 ms at
// BASELINE configs[0]'s scenario, 1 - 10 tries of the coverage check) however many layouts it draws, so it waits until
// `threshold` replicas have used theirs up (a de-phased rollout resets 2 % of the replicas per launch: one 0.3 ms
// chain per ~12 reset launches instead of one in each) -- then every replica whose coming reset has no staged layout
// yet draws it.  Replicas that reset twice between two refills draw the second layout inside the reset, as before:
// the layout is the same function of (stream, resets so far) wherever it is drawn.
extern "C" __global__ void aie_layout_decide_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena, int threshold) {
  int32_t* ctl = reinterpret_cast<int32_t*>(arena + params->a_layout_ctl);
  const bool go = ctl[0] >= threshold;
  ctl[1] = go ? 1 : 0;
  if (go) ctl[0] = 0;
}
extern "C" __global__ void __launch_bounds__(LG_NW * AIE_NT)
aie_layout_refill_kernel(const aie_params* __restrict__ params, uint8_t* __restrict__ arena) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const aie_params& P = *params;
  const int32_t* ctl = reinterpret_cast<const int32_t*>(arena + P.a_layout_ctl);
  if (!ctl[1]) return;
  const int e = (int)blockIdx.x, gtid = (int)threadIdx.x;
  const uint32_t* stg = reinterpret_cast<const uint32_t*>(arena + P.a_records + (int64_t)e * P.rec_bytes + P.o_mt);
  uint32_t st[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) st[k] = (uint32_t)aie::uni((int)stg[k]);
  const uint64_t tag = aie_layout_tag(st);
  uint64_t* tags = reinterpret_cast<uint64_t*>(arena + P.a_layout_tag);
  const uint64_t have = tags[e];
  if ((uint32_t)aie::uni((int)(uint32_t)have) == (uint32_t)tag && (uint32_t)aie::uni((int)(uint32_t)(have >> 32)) == (uint32_t)(tag >> 32)) return;
  const aie::Ctx c = aie::make_ctx(P, P, lds, e, gtid & 63, arena);
  aie::LgShared* sh = aie::lg_shared(lds + aie::lds_bytes(P), P);
  if (gtid == 0) {
    sh->lhas = 0;
    sh->lcache = 0.0;
  }
  __syncthreads();
  uint32_t ks[2];
  aie_layout_stream(st, ks);
  aie::MT ml;
  aie::mt_init(ml, P);
  ml.fkey = ks[0];
  ml.fblk = 0xffffffffu;
  ml.fsalt = ks[1];
  aie::layout_generate(c, ml, arena, lds + aie::lds_bytes(P), gtid, &sh->lhas, &sh->lcache,
                       arena + P.a_layout_stage + (int64_t)e * P.layout_stage_stride);
  if (gtid == 0) tags[e] = tag;  // (behind layout_generate's closing barrier: the staged bytes are written)
}
// compile-time instances (aie_spec_generated.h), as for the step kernel
template <int SPEC>
__global__ void __launch_bounds__(LG_NW * AIE_NT)
aie_reset_kernel_spec(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
                      const uint8_t* __restrict__ mask, int keep_rewards) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  aie::reset_body<SPEC>(params, arena, mask, keep_rewards, lds);
}

// np.random.seed(base_seed + e): init_genrand (Knuth LCG), pos = 624.
// BaseEnvironment.seed, F/base/base_env.py:481-494.  One thread per replica.
// AIE_RNG_FAST (aie_seed_fast; aie_seed with base_seed < 2^32): the counter stream keyed by base_seed + e (48 bits): key32,
// block number 0 with pos = 624 (the first draw opens block 1), salt.
extern "C" __global__ void aie_seed_kernel(const aie_params P, uint8_t* __restrict__ arena, uint64_t base_seed) {
  const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e >= P.E) return;
  uint8_t* rec = arena + P.a_records + (int64_t)e * P.rec_bytes;
  uint32_t* mt = reinterpret_cast<uint32_t*>(rec + P.o_mt);
  if (aie::rng_fast(P)) {
    const uint64_t s = base_seed + (uint64_t)e;
    mt[0] = (uint32_t)s;
    mt[1] = 0u;
    mt[2] = ((uint32_t)(s >> 32) & 0xffffu) << 16;
    mt[3] = 0u;
  } else {
    uint32_t x = (uint32_t)base_seed + (uint32_t)e;
    mt[0] = x;
    for (int i = 1; i < AIE_MT_N; ++i) {
      x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
      mt[i] = x;
    }
  }
  *reinterpret_cast<int32_t*>(rec + P.o_mt_pos) = AIE_MT_N;
  *reinterpret_cast<int32_t*>(rec + P.o_mt_has_gauss) = 0;
  *reinterpret_cast<double*>(rec + P.o_mt_gauss) = 0.0;
}

// Synthetic uniform random policy of the benchmark: a counter RNG
// keyed (seed, global replica id, t, agent); one thread per (replica, agent slot).
// The draw index t is the replica's own record field o_sample_t (read here, advanced by aie_sample_advance_kernel behind
// this launch: no host-side counter travels by value, the pair can be captured in a hipGraph and replayed).
extern "C" __global__ void aie_sample_actions_kernel(const aie_params P, const uint8_t* __restrict__ arena, uint64_t seed,
                                                     int64_t env_offset, int32_t* __restrict__ act_a,
                                                     int32_t* __restrict__ act_p) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_env = P.n * P.act_a_width + P.act_p_width;
  if (q >= (int64_t)P.E * per_env) return;
  const int e = (int)(q / per_env);
  const int64_t t = *reinterpret_cast<const int32_t*>(arena + P.a_records + (int64_t)e * P.rec_bytes + P.o_sample_t);
  aie::sample_action_slot(P, seed, env_offset, t, e, (int)(q - (int64_t)e * per_env), act_a, act_p);
}
extern "C" __global__ void aie_sample_advance_kernel(uint8_t* __restrict__ arena, int64_t a_records, int rec_bytes,
                                                     int o_sample_t, int E) {
  const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (e < E) *reinterpret_cast<int32_t*>(arena + a_records + (int64_t)e * rec_bytes + o_sample_t) += 1;
}

// Masked uniform random policy (see include/aie.h: aie_sample_masked_actions).  One thread
// per (replica, agent) and one per (replica, planner subspace): count the allowed entries
// of the relevant slice of the flattened mask, pick the floor(u * count)-th one.
extern "C" __global__ void aie_sample_masked_actions_kernel(const aie_params P, const uint8_t* __restrict__ arena,
                                                            uint64_t seed, int64_t env_offset,
                                                            int32_t* __restrict__ act_a, int32_t* __restrict__ act_p) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int per_env = P.n * P.act_a_width + P.act_p_width;
  if (q >= (int64_t)P.E * per_env) return;
  const int e = (int)(q / per_env);
  const int j = (int)(q - (int64_t)e * per_env);
  const int64_t t = *reinterpret_cast<const int32_t*>(arena + P.a_records + (int64_t)e * P.rec_bytes + P.o_sample_t);
  const uint32_t u = aie_counter_rng(seed, (uint64_t)(env_offset + e), (uint64_t)t, (uint64_t)j);
  const float* mask;
  int lo, len, stride = 1;  // mask entry k of the slot's subspace: mask[(lo + k) * stride]
  int32_t* dst;
  const bool covid = P.c.scenario == AIE_SCN_COVID;
  if (j < P.n * P.act_a_width) {
    if (!act_a) return;
    const int i = j / P.act_a_width, s = j - i * P.act_a_width;
    if (covid) {  // collated observations: the states' masks are rows [1 + levels][n] of the replica's block
      mask = reinterpret_cast<const float*>(arena + P.a_cv_obs_a) + ((int64_t)e * P.cv_nrow_obs + AIE_CV_OB_MASK) * P.n + i;
      stride = P.n;
    } else {
      mask = reinterpret_cast<const float*>(arena + P.a_obs_a_mask) + ((int64_t)e * P.n + i) * P.MA;
    }
    if (P.c.multi_action_mode_agents) {
      lo = 0;
      for (int k = 0; k < s; ++k) lo += 1 + P.sub_a_dim[k];
      len = P.n_sub_a ? 1 + P.sub_a_dim[s] : 1;
    } else {
      lo = 0;
      len = covid ? 1 + P.cv_NL : P.MA;
    }
    dst = act_a + (int64_t)e * P.n * P.act_a_width + j;
  } else {
    if (!act_p) return;
    const int s = j - P.n * P.act_a_width;
    if (covid) mask = reinterpret_cast<const float*>(arena + P.a_cv_obs_p) + (int64_t)e * (4 + P.MP) + 4;
    else mask = reinterpret_cast<const float*>(arena + P.a_obs_p_mask) + (int64_t)e * P.MP;
    if (P.c.multi_action_mode_planner) {
      lo = s * (1 + P.sub_p_dim);
      len = P.n_sub_p ? 1 + P.sub_p_dim : 1;
    } else {
      lo = 0;
      len = P.MP;
    }
    dst = act_p + (int64_t)e * P.act_p_width + s;
  }
  int count = 0;
  for (int k = 0; k < len; ++k) count += mask[(lo + k) * stride] > 0.5f ? 1 : 0;
  if (count == 0) { *dst = 0; return; }
  int pick = (int)(((uint64_t)u * (uint64_t)count) >> 32);
  int chosen = 0;
  for (int k = 0; k < len; ++k) {
    if (mask[(lo + k) * stride] > 0.5f) {
      if (pick == 0) { chosen = k; break; }
      --pick;
    }
  }
  *dst = chosen;
}

// Categorical sampling from the caller's policy logits under the current action masks (include/aie.h:
// aie_sample_policy_actions): inverse-CDF sampling (round 6; round 5: Gumbel-max) -- a slot's allowed entries get the
// weights exp(logit_k - max), their prefix sums run in a fixed order, one uniform u per slot from a counter hash keyed
// (seed, global replica, the replica's draw index t, slot) picks the first entry whose sum passes u times the total (NaN
// logits count as masked; nothing allowed: NO-OP).  float32 throughout, every operation a plain IEEE add / multiply / fma
// in a fixed order (aie_layout.h: aie_sampler_expf and the scan's definition), so the CPU restatement picks the same
// entries.  The kernel is bound by the vector instructions it issues (round 6 counters: 530 per wave and 16 waves per
// SIMD were 14 of its 19.7 us), so everything that is the same for a row runs on the scalar unit (the row's addresses,
// its hash, the pick from the ballot) and the cross-lane steps are single DPP instructions: a WORK ITEM is one action
// slot, or -- where the rows of a group are equally long and at most 32 entries (the planner's tax brackets: 22, COVID's
// states: 11) -- as many whole rows as fit the wave's 64 lanes, each in its own aligned segment of 16 or 32 lanes; a lane
// takes one entry (every 64th of a longer row).  `wpr` waves share a replica's items (a workgroup of four waves holds
// 4 / wpr replicas).  The first wave of a replica advances the draw index behind the workgroup's barrier: one launch,
// nothing by value from a host counter (replayable from a hipGraph).
//
// Cross-lane steps, written as instructions because the compiler wraps a float max in two canonicalising copies and does
// not fold a row-masked broadcast into the add (s_nop 4: a DPP source needs 5 wait states after a write of EXEC and 2
// after the VALU write of its source; the assembler text is invisible to the hazard recogniser).
__device__ __forceinline__ float sampler_row16_max(float m) {  // every lane: the maximum over its row of 16 lanes
  asm("s_nop 4\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1"
      : "+v"(m));
  return m;
}
__device__ __forceinline__ float sampler_max_raw(float a, float b) {  // (no NaN reaches the sampler's maxima)
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float sampler_segment_max(float m, int seg) {
  m = sampler_row16_max(m);
  if (seg >= 32) {  // odd rows of one copy <-> even rows of the other: both copies together hold both rows of a pair
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    m = sampler_max_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  if (seg >= 64) {  // the upper half of one copy <-> the lower half of the other
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    m = sampler_max_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
  }
  return m;
}
// Inclusive prefix sums of one 64-entry chunk in the sampler's fixed order (aie_layout.h).
__device__ __forceinline__ float sampler_scan(float v, int seg) {
  asm("s_nop 4\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1"
      : "+v"(v));
  if (seg >= 32) asm("s_nop 4\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1" : "+v"(v));
  if (seg >= 64) asm("s_nop 4\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1" : "+v"(v));
  return v;
}
// One work item in scalars: its first row; the other rows of a packed item follow at fixed strides.
struct SamplerItem {
  const float *lg, *mk;  // the first row's logits and mask entries
  int32_t* dst;          // its action
  int len, lrs, mrs, mks;  // entries; logits / mask stride from row to row; mask stride from entry to entry
  int rows, lsh;           // rows of the item that exist; log2 of the lanes per row
  uint32_t j0;             // the first row's slot in the replica
};
// A replica's agent rows and its planner rows: everything an item's addresses need (aie_sampler_args: filled by the host
// from the parameter block, a few dozen bytes of kernel arguments -- with the 8 KB block itself as the argument its fields
// arrived in a dozen dependent scalar-cache misses, most of the launch's 18 us whatever the arithmetic cost).
struct SamplerGroup {
  const float *lg, *mk;
  int32_t* dst;
  int len, lrs, mrs, mks, rpw, lsh, rows, items, slot0;
};
__device__ __forceinline__ SamplerGroup sampler_group(const aie_sampler_group& g, const uint8_t* __restrict__ arena,
                                                      const float* __restrict__ logits, int32_t* __restrict__ act, int e, int slot0) {
  SamplerGroup G;
  G.lg = logits + (uint64_t)(uint32_t)e * g.lg_estride;
  G.mk = reinterpret_cast<const float*>(arena + g.mk_off) + (uint64_t)(uint32_t)e * g.mk_estride;
  G.dst = act + (uint64_t)(uint32_t)e * (uint32_t)g.rows;
  G.len = g.len;
  G.lrs = g.lrs;
  G.mrs = g.mrs;
  G.mks = g.mks;
  G.lsh = g.lsh;
  G.rpw = 64 >> g.lsh;
  G.rows = g.rows;
  G.items = (act && logits) ? (g.rows + G.rpw - 1) >> (6 - g.lsh) : 0;
  G.slot0 = slot0;
  return G;
}
__device__ __forceinline__ SamplerItem sampler_item(const aie_sampler_args& S, const SamplerGroup& A, const SamplerGroup& Q, int it) {
  SamplerItem d;
  const bool ag = it < A.items;
  const int rpw = ag ? A.rpw : Q.rpw, r0 = (ag ? it : it - A.items) * rpw, rows = ag ? A.rows : Q.rows;
  const float* lg = ag ? A.lg : Q.lg;
  const float* mk = ag ? A.mk : Q.mk;
  d.lrs = ag ? A.lrs : Q.lrs;
  d.mrs = ag ? A.mrs : Q.mrs;
  d.mks = ag ? A.mks : Q.mks;
  d.lsh = ag ? A.lsh : Q.lsh;
  d.len = ag ? A.len : Q.len;
  d.dst = (ag ? A.dst : Q.dst) + r0;
  d.j0 = (uint32_t)(r0 + (ag ? A.slot0 : Q.slot0));
  d.rows = rows - r0 < rpw ? rows - r0 : rpw;
  if (ag && S.ragged) {  // multi-action agents: an agent's rows differ in length (lrs / mrs: the strides from agent to agent)
    const int i = r0 / S.act_a_width, s = r0 - i * S.act_a_width;
    int lo = 0;
    for (int k = 0; k < s; ++k) lo += 1 + S.params->sub_a_dim[k];
    d.len = S.params->n_sub_a ? 1 + S.params->sub_a_dim[s] : 1;
    d.lg = lg + (int64_t)i * d.lrs + lo;
    d.mk = mk + (int64_t)i * d.mrs + lo;
  } else {
    d.lg = lg + (int64_t)r0 * d.lrs;
    d.mk = mk + (int64_t)r0 * d.mrs;
  }
  return d;
}
// the lane's entry of the item's first 64-entry chunk: logit and mask value.  No branch and no use here: a lane without
// an entry reads the row's first one and sampler_item_run ignores it -- the loads of a whole group of items must issue
// back to back (a compare on the loaded value next to the load makes the wave wait right there).
__device__ __forceinline__ void sampler_item_load(const SamplerItem& d, int lane, float& x, float& mv) {
  const int sub = lane >> d.lsh, kk = lane & ((1 << d.lsh) - 1);
  const bool inb = sub < d.rows && kk < d.len;
  x = d.lg[inb ? __mul24(sub, d.lrs) + kk : 0];
  mv = d.mk[inb ? __mul24(sub, d.mrs) + __mul24(kk, d.mks) : 0];
}
__device__ __forceinline__ void sampler_item_run(const SamplerItem& d, int lane, const float x0, const float mv0, uint32_t base) {
  const int lsh = d.lsh, segw = 1 << lsh, rpw = 64 >> lsh, len = d.len;
  const int sub = lane >> lsh, kk = lane & (segw - 1);
  // a lone row's length may differ from row to row (multi-action agents) and exceed 64 (chunks); packed rows fit their segment
  const int nch = rpw > 1 ? 1 : (len + 63) >> 6;
  const int seg = rpw > 1 ? segw : (len > 32 ? 64 : aie_sampler_segment(len));
  const int lgo = __mul24(sub, d.lrs) + kk, mko = __mul24(sub, d.mrs) + __mul24(kk, d.mks);
  // ---- the row maximum over the allowed entries ----
  const bool ok0 = sub < d.rows && kk < len && mv0 > 0.5f && x0 == x0;
  float m = ok0 ? x0 : -INFINITY;
  for (int ch = 1; ch < nch; ++ch)
    if (64 * ch + kk < len) {
      const float x = d.lg[lgo + 64 * ch];
      if (d.mk[mko + 64 * ch * d.mks] > 0.5f && x > m) m = x;  // (x > m: not a NaN)
    }
  const float M = sampler_segment_max(m, seg);
  // ---- one uniform per row (scalar hashes, selected into the row's lanes) ----
  uint32_t rnd = 0u;
  for (int s = 0; s < rpw; ++s) {
    const uint32_t h = aie_sampler_entry_rng(base, d.j0 + (uint32_t)s);
    rnd = sub == s ? h : rnd;
  }
  const float u = aie_sampler_uniform(rnd);
  // ---- weights, prefix sums, the first entry whose sum passes u T ----
  int choice = -1, last_ok = -1, outv = 0;
  float T = 0.0f;
  for (int pass = (nch > 1 ? 0 : 1); pass < 2; ++pass) {  // (rows of more than 64 entries: a first pass for T)
    float carry = 0.0f;
    for (int ch = 0; ch < nch; ++ch) {
      float x = x0;
      bool ok = ok0;
      if (ch > 0) {
        ok = false;
        if (64 * ch + kk < len) {
          x = d.lg[lgo + 64 * ch];
          ok = d.mk[mko + 64 * ch * d.mks] > 0.5f && x == x;
        }
      }
      const float w = ok ? aie_sampler_expf(x - M) : 0.0f;
      const float c = carry + sampler_scan(w, seg);
      if (rpw == 1) {
        const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), seg - 1));
        if (pass == 0) {
          carry = tot;
          continue;
        }
        if (nch == 1) T = tot;
        const uint64_t oks = __ballot(ok), hits = __ballot(ok && c > u * T);
        if (choice < 0 && hits) choice = 64 * ch + (__ffsll((unsigned long long)hits) - 1);
        if (oks) last_ok = 64 * ch + (63 - __clzll((long long)oks));
        carry = tot;
      } else {  // packed rows: one chunk, a total and a pick per segment
        float Ts = 0.0f;
        for (int s = 0; s < rpw; ++s) {
          const float tot = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), (s << lsh) + segw - 1));
          Ts = sub == s ? tot : Ts;
        }
        const uint64_t oks = __ballot(ok), hits = __ballot(ok && c > u * Ts);
        const uint64_t sm = (1ull << segw) - 1ull;
        for (int s = 0; s < rpw; ++s) {
          const uint64_t h = (hits >> (s << lsh)) & sm, o = (oks >> (s << lsh)) & sm;
          const int pick = h ? __ffsll((unsigned long long)h) - 1 : (o ? 63 - __clzll((long long)o) : 0);
          outv = sub == s ? pick : outv;
        }
      }
    }
    if (pass == 0) T = carry;
  }
  if (rpw == 1) outv = choice < 0 ? (last_ok < 0 ? 0 : last_ok) : choice;
  if (sub < d.rows && kk == 0) d.dst[sub] = outv;
}
#define AIE_SAMPLER_GROUP 4  // items whose loads a wave has in flight together
extern "C" __global__ void __launch_bounds__(256)
aie_sample_policy_actions_kernel(const aie_sampler_args S, uint8_t* __restrict__ arena, const float* __restrict__ logits_a,
                                 const float* __restrict__ logits_p, uint64_t seed, int64_t env_offset,
                                 int32_t* __restrict__ act_a, int32_t* __restrict__ act_p, int wpr_log2) {
  const int lane = (int)threadIdx.x & 63, wave = aie::uni((int)threadIdx.x >> 6);
#ifdef AIE_DEV  // development: bits 8.. of the argument switch parts of the kernel off (what is the launch made of?)
  const int dev_skip = wpr_log2 >> 8;
  wpr_log2 &= 255;
  if (dev_skip & 8) return;
#else
  constexpr int dev_skip = 0;
#endif
  const int wpr = 1 << wpr_log2;
  const int e = (int)blockIdx.x * (4 >> wpr_log2) + (wave >> wpr_log2), w_in = wave & (wpr - 1);
  if (e < S.E) {
    // work items: equally long short rows share a wave, each in its own ALIGNED lane segment of 16 or 32 lanes
    const SamplerGroup A = sampler_group(S.agents, arena, logits_a, act_a, e, 0);
    const SamplerGroup Q = sampler_group(S.planner, arena, logits_p, act_p, e, S.agents.rows);
    const int items = A.items + Q.items, per_env = S.agents.rows + S.planner.rows;
    // the kernel waits on memory, not on arithmetic (a wave's items one after the other: 18 us however few instructions):
    // the draw index and a whole group of items' entries are requested before anything is computed
    const int32_t* tfield = reinterpret_cast<const int32_t*>(arena + S.t_off + (int64_t)e * S.rec_bytes);
    const int32_t t_lane = (dev_skip & 2) ? 0 : *tfield;
    uint32_t base = 0u;
    for (int it0 = w_in; it0 < items; it0 += AIE_SAMPLER_GROUP * wpr) {
      SamplerItem d[AIE_SAMPLER_GROUP];
      float x0[AIE_SAMPLER_GROUP], mv0[AIE_SAMPLER_GROUP];
#pragma unroll
      for (int g = 0; g < AIE_SAMPLER_GROUP; ++g) {
        const int it = it0 + g * wpr;
        if (it < items) {
          d[g] = sampler_item(S, A, Q, it);
          if (dev_skip & 1) {
            x0[g] = (float)(lane & 7);
            mv0[g] = 1.0f;
          } else {
            sampler_item_load(d[g], lane, x0[g], mv0[g]);
          }
        }
      }
      if (it0 == w_in) base = aie_counter_rng(seed, (uint64_t)(env_offset + e), (uint64_t)(int64_t)aie::uni(t_lane), (uint64_t)per_env);
#pragma unroll
      for (int g = 0; g < AIE_SAMPLER_GROUP; ++g)
        if (it0 + g * wpr < items) {
          if (dev_skip & 4) {
            if (x0[g] + mv0[g] == 12345.0f) d[g].dst[0] = 1;
          } else {
            sampler_item_run(d[g], lane, x0[g], mv0[g], base);
          }
        }
    }
  }
  __syncthreads();  // every wave of the replica has read the draw index: its first wave advances it
  if (e < S.E && w_in == 0 && lane == 0 && !(dev_skip & 2)) {
    int32_t* tfield = reinterpret_cast<int32_t*>(arena + S.t_off + (int64_t)e * S.rec_bytes);
    *tfield = *tfield + 1;
  }
}
// ---- the sampler's fast instances: every row of both groups is one aligned lane segment --------------------------------
// (no rows of more than 64 entries, no multi-action agents: every BASELINE configuration and COVID.)  A SIMD issues one
// scalar and one vector instruction every fourth clock whichever of its waves they come from, so the launch costs what the
// LONGER of the two instruction streams costs (round 6: 530 vector + 420 scalar instructions per wave and 16 waves per
// SIMD were 14 of the generic kernel's 19.7 us; dropping the vector count to a third moved nothing -- the scalar stream of
// its run-time row shapes, 1500 instructions for four items, had become the longer one).  Here the lanes per row are
// compile-time (LA agents, LQ planner: 16, 32 or 64 = 1 << 4 .. 6): no loops over the rows of an item, no branches on the
// segment size, lane constants per group instead of per item, the pick stored by the segment's first lane above u T, and
// the rows that need a second look (nothing allowed; rounding left no entry above u T) found by one population count.
template <int LSH>
struct SamplerFast {
  const float *lg, *mk;
  int32_t* dst;
  int lrs, mrs, rows, items, slot0;
  uint64_t kkmask;              // lanes whose entry exists (kk < len)
  uint32_t lgo, mko;            // the lane's entry inside an item: byte offsets from the item's first logit / mask entry
  uint32_t below_lo, below_hi;  // the lanes of my segment below me
  int sub, kk;
};
template <int LSH>
__device__ __forceinline__ SamplerFast<LSH> sampler_fast_group(const aie_sampler_group& g, const uint8_t* __restrict__ arena,
                                                               const float* __restrict__ logits, int32_t* __restrict__ act,
                                                               int e, int slot0, int lane) {
  SamplerFast<LSH> G;
  constexpr int SEGW = 1 << LSH;
  G.lg = logits + (uint64_t)(uint32_t)e * g.lg_estride;  // (32 x 32 -> 64 bits: two scalar instructions)
  G.mk = reinterpret_cast<const float*>(arena + g.mk_off) + (uint64_t)(uint32_t)e * g.mk_estride;
  G.dst = act + (uint64_t)(uint32_t)e * (uint32_t)g.rows;
  G.lrs = g.lrs;
  G.mrs = g.mrs;
  G.rows = g.rows;
  G.items = (act && logits) ? (g.rows + (64 >> LSH) - 1) >> (6 - LSH) : 0;
  G.slot0 = slot0;
  G.sub = lane >> LSH;
  G.kk = lane & (SEGW - 1);
  G.kkmask = __ballot(G.kk < g.len);
  G.lgo = 4u * (uint32_t)(__mul24(G.sub, g.lrs) + G.kk);
  G.mko = 4u * (uint32_t)(__mul24(G.sub, g.mrs) + __mul24(G.kk, g.mks));
  const uint64_t below = ((1ull << lane) - 1ull) & ~((1ull << (lane & ~(SEGW - 1))) - 1ull);
  G.below_lo = (uint32_t)below;
  G.below_hi = (uint32_t)(below >> 32);
  return G;
}
template <int LSH>
__device__ __forceinline__ uint64_t sampler_fast_lanes(const SamplerFast<LSH>& G, int it) {  // the lanes of item `it` that hold an entry
  const int rows = G.rows - (it << (6 - LSH));
  return rows >= (64 >> LSH) ? G.kkmask : G.kkmask & ((1ull << (rows << LSH)) - 1ull);
}
// no use of the loaded values here (sampler_item_load); a lane without an entry reads the item's first one
template <int LSH>
__device__ __forceinline__ void sampler_fast_load(const SamplerFast<LSH>& G, int it, float& x, float& mv) {
  const bool in = __builtin_amdgcn_inverse_ballot_w64(sampler_fast_lanes(G, it));
  // (32-bit byte offsets from the replica's first row: one scalar base per group, the rest in the lane's offset register)
  const uint32_t r0 = (uint32_t)it << (6 - LSH), lg0 = 4u * r0 * (uint32_t)G.lrs, mk0 = 4u * r0 * (uint32_t)G.mrs;
  x = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(G.lg) + (lg0 + (in ? G.lgo : 0u)));
  mv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(G.mk) + (mk0 + (in ? G.mko : 0u)));
}
template <int LSH>
__device__ __forceinline__ void sampler_fast_run(const SamplerFast<LSH>& G, int it, int lane, const float x, const float mv, uint32_t base) {
  constexpr int SEG = 1 << LSH, RPW = 64 >> LSH;
  const int r0 = it << (6 - LSH);
  const int rows = G.rows - r0 < RPW ? G.rows - r0 : RPW;
  const bool ok = __builtin_amdgcn_inverse_ballot_w64(sampler_fast_lanes(G, it)) && mv > 0.5f && x == x;
  const float M = sampler_segment_max(ok ? x : -INFINITY, SEG);
  const float w = ok ? aie_sampler_expf(x - M) : 0.0f;
  const float c = sampler_scan(w, SEG);
  const uint32_t slot = (uint32_t)(G.slot0 + r0);
  float thr;
  if (LSH == 6) {  // one row: its uniform and its total are scalars
    const float T = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, c), 63));
    thr = aie_sampler_uniform(aie_sampler_entry_rng(base, slot)) * T;
  } else {         // the segment's last lane to all of it: a swizzle through the LDS crossbar, no memory
    const float T = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, c), LSH == 5 ? 0x3E0 : 0x1F0));
    thr = aie_sampler_uniform(aie_sampler_entry_rng(base, slot + (uint32_t)G.sub)) * T;
  }
  const bool hit = ok && c > thr;
  const uint64_t hits = __ballot(hit);
  const bool first = hit && (((uint32_t)hits & G.below_lo) | ((uint32_t)(hits >> 32) & G.below_hi)) == 0u;
  if (first) *reinterpret_cast<int32_t*>(reinterpret_cast<char*>(G.dst) + 4u * (uint32_t)(r0 + G.sub)) = G.kk;
  if (__popcll(__ballot(first)) != rows) {  // rows without a pick: nothing allowed (NO-OP), or no sum above u T (the last allowed entry)
    const uint64_t oks = __ballot(ok), sm = LSH == 6 ? ~0ull : (1ull << SEG) - 1ull;
    for (int s = 0; s < rows; ++s)
      if (((hits >> (s << LSH)) & sm) == 0ull) {
        const uint64_t o = (oks >> (s << LSH)) & sm;
        if (lane == 0) G.dst[r0 + s] = o ? 63 - __clzll((long long)o) : 0;
      }
  }
}
template <int LA, int LQ>
__global__ void __launch_bounds__(256)
aie_sample_policy_fast_kernel(const aie_sampler_args S, uint8_t* __restrict__ arena, const float* __restrict__ logits_a,
                              const float* __restrict__ logits_p, uint64_t seed, int64_t env_offset,
                              int32_t* __restrict__ act_a, int32_t* __restrict__ act_p, int wpr_log2) {
  const int lane = (int)threadIdx.x & 63, wave = aie::uni((int)threadIdx.x >> 6);
  // every kernel argument is requested here, in one batch: fetched where first used (behind branches) they arrived in
  // four to five dependent trips to the scalar cache, cold at the start of a launch
  asm volatile("" ::"s"(S.agents.mk_off), "s"(S.agents.mk_estride), "s"(S.agents.lg_estride), "s"(S.agents.len), "s"(S.agents.lrs),
               "s"(S.agents.mrs), "s"(S.agents.mks), "s"(S.agents.rows), "s"(S.planner.mk_off), "s"(S.planner.mk_estride),
               "s"(S.planner.lg_estride), "s"(S.planner.len), "s"(S.planner.lrs), "s"(S.planner.mrs), "s"(S.planner.mks),
               "s"(S.planner.rows), "s"(S.t_off), "s"(S.rec_bytes), "s"(S.E), "s"(arena), "s"(logits_a), "s"(logits_p), "s"(seed),
               "s"(env_offset), "s"(act_a), "s"(act_p), "s"(wpr_log2));
#ifdef AIE_DEV  // development: bits 8.. of the argument switch parts of the kernel off (what is the launch made of?)
  const int dev_skip = wpr_log2 >> 8;
  wpr_log2 &= 255;
  if (dev_skip & 8) return;
#else
  constexpr int dev_skip = 0;
#endif
  const int wpr = 1 << wpr_log2;
  const int e = (int)blockIdx.x * (4 >> wpr_log2) + (wave >> wpr_log2), w_in = wave & (wpr - 1);
  if (e < S.E) {
    const SamplerFast<LA> A = sampler_fast_group<LA>(S.agents, arena, logits_a, act_a, e, 0, lane);
    const SamplerFast<LQ> Q = sampler_fast_group<LQ>(S.planner, arena, logits_p, act_p, e, S.agents.rows, lane);
    const int per_env = S.agents.rows + S.planner.rows;
    // the draw index and a whole turn's entries are requested before anything is computed
    const int32_t* tfield = reinterpret_cast<const int32_t*>(arena + S.t_off + (int64_t)e * S.rec_bytes);
    const int32_t t_lane = (dev_skip & 2) ? 0 : *tfield;
    // (per turn: two agent items and two planner items of this wave, all their loads ahead of the arithmetic)
    uint32_t base = 0u;
    for (int it_ = w_in; it_ < A.items || it_ < Q.items; it_ += 2 * wpr) {
      int it = it_;
      asm volatile("" : "+s"(it));  // (opaque: no induction variables derived from it -- the loop's one or two turns
                                    //  do not repay two dozen of them set up in every wave's prologue)
      float xa[2], ma[2], xq[2], mq[2];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (dev_skip & 1) {
          xa[g] = xq[g] = (float)(lane & 7);
          ma[g] = mq[g] = 1.0f;
        } else {
          if (it + g * wpr < A.items) sampler_fast_load<LA>(A, it + g * wpr, xa[g], ma[g]);
          if (it + g * wpr < Q.items) sampler_fast_load<LQ>(Q, it + g * wpr, xq[g], mq[g]);
        }
      }
      if (it_ == w_in) base = aie_counter_rng(seed, (uint64_t)(env_offset + e), (uint64_t)(int64_t)aie::uni(t_lane), (uint64_t)per_env);
      if (dev_skip & 4) {
        if (xa[0] + ma[0] + xa[1] + ma[1] + xq[0] + mq[0] + xq[1] + mq[1] == 12345.0f) A.dst[0] = 1;
        continue;
      }
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (it + g * wpr < A.items) sampler_fast_run<LA>(A, it + g * wpr, lane, xa[g], ma[g], base);
        if (it + g * wpr < Q.items) sampler_fast_run<LQ>(Q, it + g * wpr, lane, xq[g], mq[g], base);
      }
    }
  }
  __syncthreads();  // every wave of the replica has read the draw index: its first wave advances it
  if (e < S.E && w_in == 0 && lane == 0 && !(dev_skip & 2)) {
    int32_t* tfield = reinterpret_cast<int32_t*>(arena + S.t_off + (int64_t)e * S.rec_bytes);
    *tfield = *tfield + 1;
  }
}
#define AIE_SAMPLER_FAST(LA, LQ) \
  template __global__ void aie_sample_policy_fast_kernel<LA, LQ>(const aie_sampler_args, uint8_t*, const float*, const float*, \
                                                                 uint64_t, int64_t, int32_t*, int32_t*, int);
AIE_SAMPLER_FAST(4, 4) AIE_SAMPLER_FAST(4, 5) AIE_SAMPLER_FAST(4, 6)
AIE_SAMPLER_FAST(5, 4) AIE_SAMPLER_FAST(5, 5) AIE_SAMPLER_FAST(5, 6)
AIE_SAMPLER_FAST(6, 4) AIE_SAMPLER_FAST(6, 5) AIE_SAMPLER_FAST(6, 6)
#endif  // !AIE_JIT

#if defined(AIE_JIT) && !defined(AIE_JIT_OSE)
#ifdef AIE_JIT_PAD_NOPS  // development (AIE_JIT_PAD_NOPS=N in the environment): N no-ops ahead of the kernels shift their
                         // placement in the code object -- does a launch's duration follow where its code lies?
extern "C" __global__ void aie_jit_pad() {
#define AIE_STR2(x) #x
#define AIE_STR(x) AIE_STR2(x)
  asm volatile(".rept " AIE_STR(AIE_JIT_PAD_NOPS) "\n s_nop 0\n .endr");
}
#endif
// A second caller, never launched, for the out-of-line helpers.  Round 5 found what rounds 2 - 4 could not: a run-time
// instance was 3 - 6 % slower than the build's instance of the same family because in THIS translation unit the helpers
// have one caller each with compile-time arguments (n, the draw window's capacity ...), interprocedural constant
// propagation specialises them, and the kernels' register allocation around the call sites changes with them (16
// instead of 29 spilled VGPRs, 100 more instructions; same compiler, same flags -- hipcc --genco of this file gave
// hiprtc's code byte for byte).  In the build the helpers are shared by every instance and stay generic.  With this
// caller (run-time arguments; np_sum_leaf's mode is 1 at every call site of the build, too) the step and reset kernels
// come out instruction for instruction as the build's (llvm-objdump, tools/jit_gap_experiment.py).
extern "C" __global__ void aie_jit_keep_helpers_generic(const double* a, int n, uint32_t M, int o, int m, uint32_t* gkey,
                                                        uint32_t* w, int cap, int pos, double* out) {
  const int lane = (int)threadIdx.x & 63;
  out[0] = aie::np_sum_leaf(a, n, 1, M, o, m, lane);
  const aie::Refill r = aie::rng_refill(gkey, w, cap, pos, lane);
  const aie::Refill rf = aie::rng_refill_fast(gkey, w, cap, pos, lane);
  out[1] = (double)(r.pos + r.avail + r.twisted + rf.pos + rf.avail + rf.twisted);
  aie::MTRows t;
  for (int j = 0; j < 10; ++j) t.r[j] = gkey[64 * j + lane];
  t = aie::mt_twist_rows(t, lane);
  const aie::MTRows f = aie::mt_fast_rows_of(gkey[0], gkey[1], gkey[2], lane);
  for (int j = 0; j < 10; ++j) gkey[64 * j + lane] = t.r[j] ^ f.r[j];
}
// Run-time specialisation (aie_specialize): the step and reset kernels with THIS environment's parameter block as the
// constant image (aie_jit_image.h is generated per configuration), exactly what the build's compile-time instances
// are for the BASELINE configurations.
extern "C" __global__ void __launch_bounds__(2 * AIE_NT)
__attribute__((amdgpu_waves_per_eu(aie_spec_image<0>::waves, aie_spec_image<0>::waves)))
aie_jit_step(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
             const int32_t* __restrict__ act_a, const int32_t* __restrict__ act_p, NextActions next) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  step_body<2, false, 0>(params, arena, act_a, act_p, lds, next);
}
extern "C" __global__ void __launch_bounds__(LG_NW * AIE_NT)
aie_jit_reset(const aie_params* __restrict__ params, uint8_t* __restrict__ arena,
              const uint8_t* __restrict__ mask, int keep_rewards) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  aie::reset_body<0>(params, arena, mask, keep_rewards, lds);
}
#endif  // AIE_JIT
