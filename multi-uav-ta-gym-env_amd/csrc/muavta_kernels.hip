// muavta_kernels.hip — gfx950 kernels + the C ABI of include/muavta.h.
//
// One workgroup (one wave64) simulates one env instance with its state blob resident in LDS
// (see muavta_device.h).  Kernels:
//   k_reset     seeds -> initial state                              (MultiUAVEnv.reset)
//   k_step      load blob -> apply actions + step -> store blob     (MultiUAVEnv.step)
//   k_allocate  load blob -> Local-Hungarian -> staged actions      (HungarianAllocator.allocate_tasks)
//   k_rollout   [reset] + n x (allocate -> step) in ONE launch, blob never leaves LDS in between
//   k_metrics   calculate_metrics for every env
//   k_lsap / k_avoid   stand-alone solver / obstacle-avoidance entry points
// There is no CPU fallback: without a HIP device every entry point fails with MUAVTA_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is dlopen()ed by muavta_comm_*

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "muavta_device.h"

using namespace muavta;

namespace {

#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
#define AS4 __attribute__((address_space(4)))
// A pointer loaded from memory (or received by an out-of-line function) is a generic pointer to the compiler, and accesses
// through it become FLAT instructions, which tie up the LDS counter as well as the memory counter.  A round trip through
// the global address space tells the address-space inference what it is.
template <class T> __device__ __forceinline__ T* as_global(T* p) { return (T*)(AS1 T*)p; }

struct ObsPtrs {
  float* tasks;    // [N, 21, max_tasks]  (feature-major)
  unsigned long long* legal;  // [N, A, ceil(max_tasks/64)] bit rows
  uint8_t* pad;    // [N, max_tasks]
  float* agents;   // [N, A, 9]
  float* flags;    // [N, 5]
  double* reward;  // [N]
  uint8_t* done;   // [N]
};

// Launch-invariant context in device memory (one copy per handle).  Kernels get a pointer to it and read it through the
// scalar cache as constant memory (it is written once, by muavta_create); the out-of-line step body of k_rollout gets the
// same pointer instead of ~700 B of by-value arguments.
struct DevCtx {
  DevParams P;
  ObsPtrs O;
  uint32_t* tapes;  // [N][4][1248] MT19937 tapes
  void* blobs;      // EnvState<TL>[N]: the LDS image of every env between launches
  void* cold;       // EnvCold<TL>[N]: the HBM-only part of every env
  uint32_t* pace;   // [PACE_KEYS][16] step counters of the waves resident on each SIMD (k_rollout's issue-priority pacing)
};
enum { PACE_KEYS = 1 << 16 };  // (XCC_ID[3:0], HW_ID[15:4] = se, sh, cu, pipe, simd)
// the context as uniform constant memory: scalar loads, hoistable across the phase barriers
__device__ __forceinline__ const DevCtx& ctx_ref(const DevCtx* p) {
  const uint64_t b = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  return *(const DevCtx*)(const AS4 DevCtx*)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ ObsPtrs obs_ptrs(const DevCtx& c) {
  ObsPtrs o;
  o.tasks = as_global(c.O.tasks); o.legal = as_global(c.O.legal); o.pad = as_global(c.O.pad); o.agents = as_global(c.O.agents);
  o.flags = as_global(c.O.flags); o.reward = as_global(c.O.reward); o.done = as_global(c.O.done);
  return o;
}
template <class TL> __device__ __forceinline__ EnvState<TL>* blob_of(const DevCtx& c, int env) { return as_global(reinterpret_cast<EnvState<TL>*>(c.blobs)) + env; }
template <class TL> __device__ __forceinline__ EnvCold<TL>* cold_of(const DevCtx& c, int env) { return as_global(reinterpret_cast<EnvCold<TL>*>(c.cold)) + env; }
__device__ __forceinline__ uint32_t* tape_of(const DevCtx& c, int env) { return as_global(c.tapes) + (size_t)env * MUAVTA_RNG_STREAMS * MUAVTA_RNG_WORDS; }

template <class TL>
__device__ __forceinline__ void obs_for_env(Sim<TL>& sim, const DevParams& P, const ObsPtrs& O, int env, bool handle_buffer = true) {
  const size_t mt = (size_t)P.max_tasks, nA = (size_t)P.n_agents;
  sim.write_obs(O.tasks + (size_t)env * mt * 21, O.legal + (size_t)env * nA * ((mt + 63) >> 6), O.pad + (size_t)env * mt,
                O.agents + (size_t)env * nA * 9, O.flags + (size_t)env * 5, handle_buffer);
  if (threadIdx.x == 0) {
    O.reward[env] = sim.S.last_reward;
    O.done[env] = (uint8_t)((sim.S.terminated ? 1 : 0) | (sim.S.truncated ? 2 : 0));
  }
}

// 16-byte-per-lane coalesced copy between the HBM blob and LDS.
__device__ __forceinline__ void copy16(void* dst, const void* src, int bytes) {
  uint4* d = reinterpret_cast<uint4*>(dst);
  const uint4* s = reinterpret_cast<const uint4*>(src);
  for (int i = threadIdx.x; i < bytes / 16; i += WG) d[i] = s[i];
}

extern __shared__ __align__(16) unsigned char muavta_smem[];
template <class TL>
struct Lds {
  EnvState<TL>* S;
  Scratch<TL>* X;
  __device__ Lds(unsigned char* base) {
    S = reinterpret_cast<EnvState<TL>*>(base);
    X = reinterpret_cast<Scratch<TL>*>(base + ((sizeof(EnvState<TL>) + 15) & ~size_t(15)));
  }
  static constexpr size_t bytes() { return ((sizeof(EnvState<TL>) + 15) & ~size_t(15)) + sizeof(Scratch<TL>); }
};

#define smem muavta_smem
#ifdef MUAVTA_PROF
#define PROF_AT(sim, i) do { if (threadIdx.x == 0) { unsigned long long t_ = clock64(); (sim).prof_lds()[i] += t_ - (sim).prof_lds()[PROF_N]; (sim).prof_lds()[PROF_N] = t_; } } while (0)
#define PROF_EXTRA_LDS MUAVTA_PROF_LDS_BYTES
#else
#define PROF_AT(sim, i) do { } while (0)
#define PROF_EXTRA_LDS 0
#endif
// The fused kernels own a STATIC LDS block of their tile's size: its address is a compile-time constant (0), so LDS addresses
// fold into the ds_* offset fields.  With the dynamic `extern __shared__` array every address was formed as `0 + x` at run
// time (v_add_u32 v, 0, v / s_add_i32 s, 0, imm: 2 % of the kernel's VALU instructions).  They are launched with no dynamic LDS.
// Experiment (MUAVTA_LDS_ZERO_REG=1, off): a DS instruction takes its address from a VGPR, so every access at a constant address
// is preceded by its own rematerialised `v_mov_b32 v, 0` (887 in the 16-agent rollout kernel, 6 % of its static VALU
// instructions).  Addressing the block through ONE pinned zero register the compiler cannot see through removes 400 of them,
// but loads at uniform addresses then stop being uniform values: 560 scalar branches become exec-mask regions and 490 address
// adds move from the SALU to the VALU.  Measured r3: 219 M env-steps/s against 231 M (config 2), 55.0 against 57.1 M (config 4).
#if MUAVTA_LDS_ZERO_REG
static __device__ __forceinline__ uint32_t lds_zero() { uint32_t z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); return z; }
#else
static __device__ __forceinline__ uint32_t lds_zero() { return 0u; }
#endif
#define KERNEL_LDS(TL) __shared__ __align__(16) unsigned char lds_own[Lds<TL>::bytes() + PROF_EXTRA_LDS]
// Residency on a CU is bound by LDS bytes per env (160 KiB per CU, 1 KiB granule): 16 envs of the 16-agent tile
// (BASELINE configs 2 and 3: 4096 envs = 16 per CU, one round) need <= 10 KiB each.
static_assert(Lds<Tile16>::bytes() <= 10240, "Tile16 no longer fits 16 workgroups per CU");
static_assert(!MUAVTA_TILE24_SLIM || Lds<Tile24>::bytes() <= 10240, "Tile24 no longer fits 16 workgroups per CU");
static_assert(sizeof(EnvState<Tile16>) % 16 == 0 && sizeof(EnvState<Tile24>) % 16 == 0 && sizeof(EnvState<Tile64>) % 16 == 0, "blob copies move 16 B per lane");
static_assert(sizeof(EnvCold<Tile16>) % 16 == 0 && sizeof(EnvCold<Tile24>) % 16 == 0 && sizeof(EnvCold<Tile64>) % 16 == 0, "cold records are 16 B aligned");

// Minimum waves per SIMD the register allocator must leave room for in the fused rollout / step kernels: Tile::MIN_WAVES (4 =>
// at most 128 VGPRs, which is what 16 single-wave workgroups per CU need; 2 => 256 VGPRs on the 64-agent tile).
#ifdef MUAVTA_MIN_WAVES  // (experiments: one value for every tile)
#define TILE_MIN_WAVES(TL) MUAVTA_MIN_WAVES
#else
#define TILE_MIN_WAVES(TL) TL::MIN_WAVES
#endif

// ---- RNG seeding, one LANE per stream --------------------------------------------------------------------------
// CPython's init_by_array is a serial recurrence of 2 x 624 dependent steps; inside k_reset / k_rollout one lane of
// the env's wave would walk it while 63 idle (it was 65 % of a reset).  Here 16 envs x 4 streams share a wave: k_seed
// seeds every env's agent stream (Random(seed), DroneEnv.py:531-533) and draws the three stream seeds from it
// (randint(0, 2^63-1) x 3, :535-538), then the obs / tgt / mission lanes seed theirs.  The reset kernels only load
// the four 624-word states.  Layout: seedbuf [N][4][624] u32 (per-env block contiguous for the coalesced load there).
// init_genrand(19650218), the key-independent half of init_by_array: a compile-time table read through the scalar cache
struct GenrandTable {
  uint32_t v[624];
  constexpr GenrandTable() : v{} {
    uint32_t g = 19650218u;
    v[0] = g;
    for (int i = 1; i < 624; i++) { g = 1812433253u * (g ^ (g >> 30)) + (uint32_t)i; v[i] = g; }
  }
};
__constant__ GenrandTable G_TAB = GenrandTable();
// init_by_array(key[0..len)) into this lane's 624 words of seedbuf (same recurrence as mt_seed).  len is 1 or 2 (seeds below /
// from 2^32): key[j] + j alternates k0, k1 + 1 for len 2 and is k0 for len 1.  The kernel needs no LDS (r2 kept a [624][65]-word
// tile: 162 KB, a whole CU's LDS, which is why k_seed could only start once the previous rollout had drained a CU): the first
// loop's 624 words go through a scratch buffer in GLOBAL memory, 16 words (four 16-byte vectors) at a time in a lane-
// interleaved layout [vector][lane] — every store / load instruction of the wave moves 1 KB of consecutive bytes — and the
// second loop reads them back two chunks ahead of its dependent chain and writes the final state straight into the stream's
// own 624 words (per-lane 64-byte runs: stores, nobody waits for them).  mt[0] is only ever read as the running `prev` (kept
// in a register) and ends as 0x80000000; mt[1] is rewritten by the two wrap-around steps.
typedef uint32_t seed_u4 __attribute__((ext_vector_type(4)));
DEV void seed_stream(uint32_t* mt, seed_u4* tmp /* this wave's [156][64] vectors, already offset by the lane */, uint32_t k0, uint32_t k1, int len) {
  seed_u4* mt4 = reinterpret_cast<seed_u4*>(mt);
  uint32_t prev = 19650218u, m1 = 0;
  const uint32_t add_even = k0, add_odd = len == 2 ? k1 + 1u : k0;  // first loop step i uses j = (i - 1) % len
  for (int base = 0; base < 624; base += 16) {  // first loop, i = 1 .. 623
    uint32_t m[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int i = base + q;
      if (i != 0) prev = (G_TAB.v[i] ^ ((prev ^ (prev >> 30)) * 1664525u)) + (((q - 1) & 1) ? add_odd : add_even);  // (i - 1) & 1 == (q - 1) & 1
      m[q] = i != 0 ? prev : 0x80000000u;
      if (i == 1) m1 = prev;
    }
#pragma unroll
    for (int v = 0; v < 4; v++) tmp[(base / 4 + v) * WG] = seed_u4{m[4 * v], m[4 * v + 1], m[4 * v + 2], m[4 * v + 3]};
  }
  // 624th step of the first loop: i wrapped to 1 (mt[0] = mt[623]), j = 623 % len
  prev = (m1 ^ ((prev ^ (prev >> 30)) * 1664525u)) + (len == 2 ? add_odd : add_even);
  m1 = prev;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunks are read back below (same lane, same addresses)
  seed_u4 n0[4], n1[4];  // the next two chunks, in flight
#pragma unroll
  for (int v = 0; v < 4; v++) { n0[v] = tmp[v * WG]; n1[v] = tmp[(4 + v) * WG]; }
  for (int base = 0; base < 624; base += 16) {  // second loop, i = 2 .. 623
    uint32_t m[16];
#pragma unroll
    for (int v = 0; v < 4; v++) { m[4 * v] = n0[v].x; m[4 * v + 1] = n0[v].y; m[4 * v + 2] = n0[v].z; m[4 * v + 3] = n0[v].w; n0[v] = n1[v]; }
    if (base + 32 < 624) {
#pragma unroll
      for (int v = 0; v < 4; v++) n1[v] = tmp[((base + 32) / 4 + v) * WG];
    }
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int i = base + q;
      if (i >= 2) { prev = (m[q] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i; m[q] = prev; }
    }
#pragma unroll
    for (int v = 0; v < 4; v++) mt4[base / 4 + v] = seed_u4{m[4 * v], m[4 * v + 1], m[4 * v + 2], m[4 * v + 3]};  // (chunk 0 carries mt[0] = 0x80000000 and a stale mt[1])
  }
  // 623rd iteration of the second loop: i wrapped to 1 with mt[0] = mt[623]
  prev = (m1 ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
  mt[1] = prev;  // (after the chunk store above: stores of one lane to one address land in order)
}
// i-th output word of the first block after seeding (i < 227), from this lane's state in global memory
DEV uint32_t seed_output(const uint32_t* mt, int i) {
  const uint32_t y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
  uint32_t v = mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  v ^= (v >> 11); v ^= (v << 7) & 0x9d2c5680u; v ^= (v << 15) & 0xefc60000u; v ^= (v >> 18);
  return v;
}
// One wave = 16 envs x 4 streams, one lane each.  Phase 1: the agent-stream lanes run init_by_array(seed) and draw the three
// stream seeds (Random.randint(0, 2^63-1) == _randbelow(2^63): getrandbits(64) until < 2^63, in the order obs, tgt, mission,
// DroneEnv.py:535-538); phase 2: the obs / tgt / mission lanes run theirs, the seeds handed over by a lane shuffle.  No LDS and
// ~40 VGPRs: the waves find room next to a running rollout, so the seeding of the NEXT launch overlaps all of this one.
__global__ __launch_bounds__(WG, 8) void k_seed(const uint64_t* seeds, int n, int with_obs, uint32_t* seedbuf, uint32_t* seedtmp) {
  const int lane = threadIdx.x, el = lane >> 2, st = lane & 3;
  const int e = blockIdx.x * 16 + el;
  uint32_t* mt = seedbuf + ((size_t)blockIdx.x * WG + lane) * 624;  // [N][4][624]: this lane's stream
  seed_u4* tmp = reinterpret_cast<seed_u4*>(seedtmp) + (size_t)blockIdx.x * 156 * WG + lane;  // this wave's [156][64] scratch vectors
  unsigned long long d0 = 0, d1 = 0, d2 = 0;
  if (e < n && st == ST_AGENT) {
    const uint64_t seed = seeds[e];
    seed_stream(mt, tmp, (uint32_t)seed, (uint32_t)(seed >> 32), (seed >> 32) ? 2 : 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    int i = 0, j = 0;
    while (j < 3 && i <= 220) {  // (110 rejections in a row: the three seeds stay 0)
      const uint64_t lo = seed_output(mt, i), hi = seed_output(mt, i + 1);
      i += 2;
      const uint64_t r = lo | (hi << 32);
      if (r < (1ull << 63)) { if (j == 0) d0 = r; else if (j == 1) d1 = r; else d2 = r; j++; }
    }
  }
  // the agent lane of each env hands its draws to the env's other lanes
  const int src = lane & ~3;
  d0 = __shfl(d0, src); d1 = __shfl(d1, src); d2 = __shfl(d2, src);
  if (e < n && st != ST_AGENT && (st != ST_OBS || with_obs)) {
    const uint64_t sd = st == ST_OBS ? d0 : st == ST_TGT ? d1 : d2;
    seed_stream(mt, tmp, (uint32_t)sd, (uint32_t)(sd >> 32), (sd >> 32) ? 2 : 1);
  }
}

template <class TL>
__global__ __launch_bounds__(WG) void k_reset(const DevCtx* __restrict__ ctxp, const uint64_t* seeds, const uint32_t* seedbuf) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = blockIdx.x;
  Lds<TL> L(smem);
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, tape_of(ctx, env));
  sim.reset(seeds[env], seedbuf + (size_t)env * 4 * 624);
  obs_for_env(sim, ctx.P, obs_ptrs(ctx), env);
  lds_sync();
  copy16(blob_of<TL>(ctx, env), L.S, sizeof(EnvState<TL>));
}

// act_agent == nullptr: use the actions staged in the blob by k_allocate
template <class TL>
__global__ __launch_bounds__(WG, TILE_MIN_WAVES(TL)) void k_step(const DevCtx* __restrict__ ctxp, const int32_t* act_agent, const int32_t* act_index, int act_cap,
                                             double* rel_log, int env_base) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const DevParams& P = ctx.P;
  const int env = env_base + blockIdx.x;  // (env_base: the first env of a sub-batch launched on its own stream, muavta_*_part)
  KERNEL_LDS(TL);
  Lds<TL> L(lds_own + lds_zero());
  EnvState<TL>* blob = blob_of<TL>(ctx, env);
  copy16(L.S, blob, sizeof(EnvState<TL>));
  lds_sync();
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, P, tape_of(ctx, env));
  if (rel_log) sim.rel_log = rel_log + (size_t)env * (1 + MUAVTA_REL_ROW * TL::T);
  if (act_agent) {
    // the env's row of (agent, index) items: the first TL::A of them are staged here, a longer row (muavta_step_lists) is
    // consumed by the action phase A items at a time
    sim.more_agent = act_agent + (size_t)env * act_cap; sim.more_index = act_index + (size_t)env * act_cap;
    sim.more_cap = act_cap; sim.more_pos = 0;
    sim.stage_more();
    if (act_cap <= TL::A) sim.more_agent = nullptr;  // (uniform: nothing beyond the staged items)
  }
  lds_sync();
  sim.step(true);
  obs_for_env(sim, P, obs_ptrs(ctx), env);
  lds_sync();
  copy16(blob, L.S, sizeof(EnvState<TL>));
}

template <class TL>
__global__ __launch_bounds__(WG) void k_allocate(const DevCtx* __restrict__ ctxp, int interval, int use_vis, int mode,
                                                 int32_t* out_agent, int32_t* out_index, int act_cap, int env_base) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = env_base + blockIdx.x;
  Lds<TL> L(smem);
  EnvState<TL>* blob = blob_of<TL>(ctx, env);
  copy16(L.S, blob, sizeof(EnvState<TL>));
  lds_sync();
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, tape_of(ctx, env));
  sim.allocate(interval, use_vis, mode);
  lds_sync();
  if (out_agent) {
    const EnvState<TL>& S = *L.S;
    for (int k = threadIdx.x; k < act_cap; k += WG) {
      out_agent[(size_t)env * act_cap + k] = k < S.n_act ? S.act_agent[k] : -1;
      out_index[(size_t)env * act_cap + k] = k < S.n_act ? S.act_index[k] : 0;
    }
  }
  copy16(blob, L.S, sizeof(EnvState<TL>));
}

template <class TL>
__device__ __forceinline__ typename Sim<TL>::TokPtrs global_tok_ptrs(typename Sim<TL>::TokPtrs K) {  // kernel-argument pointers as global (not FLAT) accesses
  K.task_feats = as_global(K.task_feats); K.task_mask = as_global(K.task_mask); K.task_ids = as_global(K.task_ids);
  K.agent_feats = as_global(K.agent_feats); K.agent_mask = as_global(K.agent_mask); K.agent_ids = as_global(K.agent_ids);
  K.edge_valid = as_global(K.edge_valid); K.n_urgent = as_global(K.n_urgent); K.expert_mask = as_global(K.expert_mask);
  K.replanned = as_global(K.replanned);
  return K;
}

// muavta_allocate_scored: k_allocate with the caller's edge scores / priorities / reserved agents (Sim::allocate<true>).  The
// task list handed to the allocator sits in T bytes of LDS behind the tile.
enum { SCORED_EXTRA_LDS = 128 };
template <class TL>
__global__ __launch_bounds__(WG) void k_allocate_scored(const DevCtx* __restrict__ ctxp, ScoredDev sc, int interval, int use_vis,
                                                        int32_t* out_agent, int32_t* out_index, int act_cap, int env_base) {
  static_assert(TL::T <= SCORED_EXTRA_LDS, "the scored allocator's task list is one byte per slot");
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = env_base + blockIdx.x;
  Lds<TL> L(smem);
  EnvState<TL>* blob = blob_of<TL>(ctx, env);
  copy16(L.S, blob, sizeof(EnvState<TL>));
  lds_sync();
  sc.scores = as_global(sc.scores); sc.pri = as_global(sc.pri); sc.reserved = as_global(sc.reserved);
  sc.selected = as_global(sc.selected); sc.replanned = as_global(sc.replanned);
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, tape_of(ctx, env));
  sim.template allocate<true>(interval, use_vis, 4, &sc, env, smem + Lds<TL>::bytes());
  lds_sync();
  if (out_agent) {
    const EnvState<TL>& S = *L.S;
    for (int k = threadIdx.x; k < act_cap; k += WG) {
      out_agent[(size_t)env * act_cap + k] = k < S.n_act ? S.act_agent[k] : -1;
      out_index[(size_t)env * act_cap + k] = k < S.n_act ? S.act_index[k] : 0;
    }
  }
  copy16(blob, L.S, sizeof(EnvState<TL>));
}

// muavta_rl_step_device: one iteration of run_rl_episode's loop body (experiments/train_pair_cost.py:139-153) per env and launch —
// policy.plan with the caller's scores (allocate<true>) -> _apply_assign -> env.step -> compute_s_wps -> build_tokens (next_tok) —
// with ONE load and ONE store of the env record instead of four (k_tokens, k_allocate_scored, k_step, k_metrics).
template <class TL>
__global__ __launch_bounds__(WG, TILE_MIN_WAVES(TL)) void k_rl_step(const DevCtx* __restrict__ ctxp, ScoredDev sc, typename Sim<TL>::TokPtrs K, int interval, int use_vis,
                                                                    int write_obs, double* s_wps, uint8_t* done, int n_envs, int env_base) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const DevParams& P = ctx.P;
  const int env = env_base + blockIdx.x;
  __shared__ __align__(16) unsigned char lds_own[Lds<TL>::bytes() + SCORED_EXTRA_LDS];
  Lds<TL> L(lds_own + lds_zero());
  EnvState<TL>* blob = blob_of<TL>(ctx, env);
  copy16(L.S, blob, sizeof(EnvState<TL>));
  lds_sync();
  sc.scores = as_global(sc.scores); sc.pri = as_global(sc.pri); sc.reserved = as_global(sc.reserved);
  sc.selected = as_global(sc.selected); sc.replanned = as_global(sc.replanned);
  K = global_tok_ptrs<TL>(K);
  s_wps = as_global(s_wps); done = as_global(done);
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, P, tape_of(ctx, env));
  const double before = sim.s_wps();
  if (!(L.S->terminated || L.S->truncated)) {  // (uniform)  the reference's loop ends with the episode (:139)
    sim.template allocate<true>(interval, use_vis, 4, &sc, env, lds_own + Lds<TL>::bytes());
    lds_sync();
    sim.step(true);
    if (write_obs) obs_for_env(sim, P, obs_ptrs(ctx), env);
    lds_sync();
  } else {
    if (sc.replanned && threadIdx.x == 0) sc.replanned[env] = 0;
    if (sc.selected) for (int i = threadIdx.x; i < sc.MA * sc.MT; i += WG) sc.selected[(size_t)env * sc.MA * sc.MT + i] = 0.f;
  }
  if (threadIdx.x == 0) {
    if (s_wps) { s_wps[env] = before; s_wps[(size_t)n_envs + env] = sim.s_wps(); }
    if (done) done[env] = (uint8_t)((L.S->terminated ? 1 : 0) | (L.S->truncated ? 2 : 0));
  }
  if (K.task_feats) {
    cold_sync();
    sim.tokens(K, env);  // next_tok (:150): the tokens the policy sees at the next iteration
  }
  lds_sync();
  copy16(blob, L.S, sizeof(EnvState<TL>));
}

// muavta_rl_run_device / muavta_step_run: run to the next replan gate.  The reference's trainer and evaluation loops consult the planner
// only when their gate fires and step with EMPTY actions otherwise (experiments/train_pair_cost.py:86-89,139-145; wps_eval.py:248-254,273):
//     if _should_replan(env, events): result = policy.plan(...); actions = _apply_assign(env, result)      <- the launch's FIRST step
//     obs, reward, done, trunc, info = env.step(actions)                                                    <- quiet steps: actions = {}
// One launch per env takes the first step with its plan — src 0: allocate<true> with the caller's scores under the caller's gate (one
// iteration of run_rl_episode: S_WPS before / after, next_tok, done); src 1: the actions muavta_allocate staged; src 2: the caller's
// action rows — and then keeps stepping quietly until the env's own gate fires again, its episode ends, or max_steps steps are taken
// (0: no bound).  Where it stopped: park tokens (what the policy sees next), park flags, the number of steps taken, the summed reward.
struct RunOut {
  double* s_wps;        // [2][N] S_WPS before / after the first step (src 0)
  uint8_t* done;        // [N] done flags after the first step (src 0: ep_done of the pushed transition)
  int32_t* n_stepped;   // [N] env steps this launch took
  uint8_t* park;        // [N] bit 0 terminated, bit 1 truncated, bit 2 stopped at a gate (the next launch plans)
  double* reward_sum;   // [N] rewards of this launch's steps, added in step order
};
enum { RUN_SRC_SCORED = 0, RUN_SRC_STAGED = 1, RUN_SRC_ROWS = 2 };
// The launch's tensors (~45 pointers) are a by-value kernel argument that the body never names: it reads them from the KERNARG SEGMENT
// (constant memory, through the scalar cache) where they are used, like the context.  Named, the compiler loads every argument in the
// prologue and keeps them live across the step loop (first build: 259 SGPR spill stores and 60 spilled VGPRs on the 16-agent tile).
template <class TL> struct RunArgs { ScoredDev sc; typename Sim<TL>::TokPtrs K, KP; RunOut R; };
template <class TL>
__global__ __launch_bounds__(WG, TILE_MIN_WAVES(TL)) void k_run(RunArgs<TL> args_in_kernarg_segment, const DevCtx* __restrict__ ctxp, int src, int gate, int interval, int use_vis,
                                                                int write_obs, int max_steps, const int32_t* act_agent, const int32_t* act_index, int act_cap, int n_envs,
                                                                int env_base) {
  static_assert(alignof(RunArgs<TL>) <= 8, "first kernel argument: offset 0 of the kernarg segment");
  const uint64_t argp = (uint64_t)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();
  const DevCtx& ctx0 = ctx_ref(ctxp);
  const int env = env_base + blockIdx.x;
  __shared__ __align__(16) unsigned char lds_own[Lds<TL>::bytes() + SCORED_EXTRA_LDS];
  Lds<TL> L(lds_own + lds_zero());
  EnvState<TL>* blob = blob_of<TL>(ctx0, env);
  copy16(L.S, blob, sizeof(EnvState<TL>));
  lds_sync();
  const uint32_t arg_lo = __builtin_amdgcn_readfirstlane((uint32_t)argp), arg_hi = __builtin_amdgcn_readfirstlane((uint32_t)(argp >> 32));
  int n = 0;
  bool at_gate = false;
  double rsum = 0.0;
  const bool over0 = L.S->terminated || L.S->truncated;  // (uniform)  the reference's loops end with the episode: such an env is left alone
  for (;;) {  // ONE call site each for the planner, the step and the token builder; every iteration builds its own Sim (lane-derived values
              // and launch constants must not be hoisted across the step: see k_rollout)
    uint32_t lo = arg_lo, hi = arg_hi;
    asm volatile("" : "+s"(lo), "+s"(hi));
    const RunArgs<TL>& G = *(const RunArgs<TL>*)(const AS4 RunArgs<TL>*)(((uint64_t)hi << 32) | (uint64_t)lo);
    const DevCtx& ctx = ctx_ref(ctxp);
    Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, tape_of(ctx, env));
    const bool first = n == 0;
    if (!over0) {
      if (first) {
        if (src == RUN_SRC_SCORED) {
          ScoredDev sc = G.sc;
          sc.scores = as_global(sc.scores); sc.pri = as_global(sc.pri); sc.reserved = as_global(sc.reserved);
          sc.selected = as_global(sc.selected); sc.replanned = as_global(sc.replanned);
          if (threadIdx.x == 0 && G.R.s_wps) as_global(G.R.s_wps)[env] = sim.s_wps();  // before the step
          sim.template allocate<true>(interval, use_vis, 4, &sc, env, lds_own + Lds<TL>::bytes());
          lds_sync();
        } else if (src == RUN_SRC_ROWS) {
          sim.more_agent = act_agent + (size_t)env * act_cap; sim.more_index = act_index + (size_t)env * act_cap;
          sim.more_cap = act_cap; sim.more_pos = 0;
          sim.stage_more();
          if (act_cap <= TL::A) sim.more_agent = nullptr;  // (uniform: nothing beyond the staged items)
          lds_sync();
        }
      } else {
        if (threadIdx.x == 0) L.S->n_act = 0;  // env.step({})
        lds_sync();
      }
      sim.step(true);
      n++;
      lds_sync();
      rsum += L.S->last_reward;
    } else if (src == RUN_SRC_SCORED) {
      const ScoredDev& sc = G.sc;
      if (sc.replanned && threadIdx.x == 0) as_global(sc.replanned)[env] = 0;
      if (sc.selected) for (int i = threadIdx.x; i < sc.MA * sc.MT; i += WG) as_global(sc.selected)[(size_t)env * sc.MA * sc.MT + i] = 0.f;
      if (threadIdx.x == 0 && G.R.s_wps) as_global(G.R.s_wps)[env] = sim.s_wps();
    }
    const bool over = L.S->terminated || L.S->truncated;
    if (!over) at_gate = sim.gate_fires(gate, interval);
    const bool stop = over || at_gate || (max_steps > 0 && n >= max_steps);
    const bool planned = first && src == RUN_SRC_SCORED && !over0 && L.S->gate_step == sim.tnow;  // (uniform) the gate fired at the step just taken: an RL sample
    if (first && src == RUN_SRC_SCORED && threadIdx.x == 0) {
      if (G.R.s_wps) as_global(G.R.s_wps)[(size_t)n_envs + env] = sim.s_wps();
      if (G.R.done) as_global(G.R.done)[env] = (uint8_t)((L.S->terminated ? 1 : 0) | (L.S->truncated ? 2 : 0));
    }
    // next_tok of the planned step (train_pair_cost.py:150-151) and / or the tokens of the state the env stops in
#pragma nounroll
    for (int w = 0; w < 2; w++) {
      const typename Sim<TL>::TokPtrs& Kc = w == 0 ? G.K : G.KP;
      const bool want = Kc.task_feats != nullptr && (w == 0 ? planned : stop);
      if (!want) continue;
      const typename Sim<TL>::TokPtrs cur = global_tok_ptrs<TL>(Kc);
      cold_sync();
      sim.tokens(cur, env);
      lds_sync();
    }
    if (stop) {
      if (write_obs && !over0) { obs_for_env(sim, ctx.P, obs_ptrs(ctx), env); lds_sync(); }
      if (threadIdx.x == 0) {
        if (G.R.n_stepped) as_global(G.R.n_stepped)[env] = n;
        if (G.R.park) as_global(G.R.park)[env] = (uint8_t)((L.S->terminated ? 1 : 0) | (L.S->truncated ? 2 : 0) | (at_gate ? 4 : 0));
        if (G.R.reward_sum) as_global(G.R.reward_sum)[env] = rsum;
      }
      break;
    }
  }
  lds_sync();
  copy16(blob, L.S, sizeof(EnvState<TL>));
}

// The body of the fused rollout, OUT OF LINE on purpose.  Inlined into the 150-step loop of k_rollout the compiler hoists
// loop invariants across the whole body and the kernel needs 255 VGPRs (+188 B/lane of scratch: two waves per SIMD); as a
// function of its own the body fits the 128 VGPRs of FOUR waves per SIMD — with 10 KiB of LDS per env that is 16 resident
// envs per CU, the whole 4096-env batch in one round.  The function must not name the `extern __shared__` array (every
// reference becomes a load from the dynamic-LDS offset table) nor take generic pointers (FLAT accesses): it gets the LDS
// base as a number and rebuilds typed pointers.  An iteration is  step -> observation write -> NEXT step's allocate:
// the function's return waits for all memory operations, and this way the observation stores have drained by then.
#ifndef MUAVTA_PHASE_ATTR
#define MUAVTA_PHASE_ATTR __forceinline__
#define MUAVTA_PHASE_INLINED 1  // the body sees the kernel's own `smem`: LDS addresses fold to constants + lane offsets
#else
#define MUAVTA_PHASE_INLINED 0
#endif
enum { PH_ALLOC = 1, PH_STEP = 2, PH_OBS = 4 };
// muavta_rollout_record: per-step training data of the fused rollout in caller-owned rings [n_slots][N][...] — the token
// tensors / expert labels of the plan staged for step t and S_WPS before step t (experiments/train_pair_cost.py:96-156).
template <class TL>
struct RecordPtrs {
  typename Sim<TL>::TokPtrs K;  // K.task_feats == nullptr: no token rings
  double* s_wps;   // [n_steps + 1][N]
  ObsPtrs O;       // O.tasks == nullptr: no observation rings
  int n_envs;
};
struct RecBlob { uint32_t w[64]; };  // 256 B: a RecordPtrs<TL> by value
__global__ void k_store_rec(RecBlob b, uint32_t* dst) { dst[threadIdx.x] = b.w[threadIdx.x]; }

template <class TL, bool REC>
__device__ MUAVTA_PHASE_ATTR void rollout_phase(const DevCtx* ctxp, unsigned char* lds_own, uint32_t lds_base, int env, int phases, int interval, int use_vis, int mode,
                                                const RecordPtrs<TL>& rec, int slot, int oslot) {
  const DevCtx& ctx = ctx_ref(ctxp);
#if MUAVTA_PHASE_INLINED
  Lds<TL> L(lds_own + lds_zero());
#else
  Lds<TL> L((unsigned char*)(AS3 unsigned char*)(uintptr_t)__builtin_amdgcn_readfirstlane(lds_base));
#endif
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, tape_of(ctx, env));
  if (phases & PH_STEP) sim.step(true);
  if (phases & PH_OBS) {
    ObsPtrs O = obs_ptrs(ctx);
    int at = env;
    bool handle_buffer = true;
    if constexpr (REC) {
      if (oslot >= 0) {  // slot `oslot` of the caller's observation rings instead of the handle's single buffer (a fresh buffer: every row is written)
        handle_buffer = false;
        O.tasks = as_global(rec.O.tasks); O.legal = as_global(rec.O.legal); O.pad = as_global(rec.O.pad); O.agents = as_global(rec.O.agents);
        O.flags = as_global(rec.O.flags); O.reward = as_global(rec.O.reward); O.done = as_global(rec.O.done);
        at = oslot * rec.n_envs + env;
      }
    }
    obs_for_env(sim, ctx.P, O, at, handle_buffer);
  }
  lds_sync();
  if (!ABL(9) && (phases & PH_ALLOC) && !(L.S->terminated || L.S->truncated)) {
    sim.allocate(interval, use_vis, mode);
    if (REC && rec.K.task_feats) {  // the sample of step `slot`: tokens + labels of the plan just staged, S_WPS before the step
      const typename Sim<TL>::TokPtrs K = global_tok_ptrs<TL>(rec.K);
      cold_sync();
      sim.tokens(K, slot * rec.n_envs + env);
      if (threadIdx.x == 0) as_global(rec.s_wps)[(size_t)slot * rec.n_envs + env] = sim.s_wps();
      lds_sync();
    }
  }
  PROF_AT(sim, 20);
}

template <class TL, bool REC>
__global__ __launch_bounds__(WG, TILE_MIN_WAVES(TL)) void k_rollout(const DevCtx* __restrict__ ctxp, const uint64_t* seeds, int n_steps, int interval, int use_vis,
                                                int mode, int write_obs, double* metrics, const uint32_t* seedbuf, const RecordPtrs<TL>* __restrict__ recp, int epoch, int env_base) {
  const DevCtx& ctx = ctx_ref(ctxp);
  // The ring pointers of muavta_rollout_record sit in device memory (one slot per handle, written on the launch's stream just ahead of
  // it) and are read through the scalar cache where they are used, like the context: as by-value kernel arguments they were ~40
  // SGPRs that the recording variants kept alive across the whole step loop (r3: 176-181 SGPR spills, 8 spilled VGPRs + 48 B of
  // scratch on the 24-agent tile).  The plain rollout gets the handle's all-zero slot and never reads it.
  const uint32_t rec_lo = __builtin_amdgcn_readfirstlane((uint32_t)(uint64_t)recp), rec_hi = __builtin_amdgcn_readfirstlane((uint32_t)((uint64_t)recp >> 32));
  // (unsigned halves: v_readfirstlane returns int, and a low word with bit 31 set would sign-extend into the high word)
  const RecordPtrs<TL>& rec = *(const RecordPtrs<TL>*)(const AS4 RecordPtrs<TL>*)(((uint64_t)rec_hi << 32) | (uint64_t)rec_lo);
  const int env = env_base + blockIdx.x;
  KERNEL_LDS(TL);
  Lds<TL> L(lds_own);
  EnvState<TL>* blob = blob_of<TL>(ctx, env);
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, tape_of(ctx, env));
// (r4) The pacing row is read and written with WORKGROUP-scope (plain) accesses: the waves that share a row share a SIMD, hence a CU
// and its vector L1, which is coherent for them; a slightly stale counter only delays a priority change by a step.  With AGENT scope
// (r2, r3) every step's 16-lane read went to the memory side of the fabric (~2 us on this eight-XCD part) and — VMEM loads return in
// order — the step's first `s_waitcnt vmcnt(0)` waited for it: tools/ablate_probe.py measured the whole pacing block at 0.35 ms of a
// 1.19 ms quiet launch.  Headline 240 -> 248 M on one box (profiles/r04_ab_pacing.txt); publishing every 2nd / 4th step changes nothing.
// NOTE on the memory model: the waves that share a row belong to DIFFERENT workgroups (one env per workgroup), and workgroup scope
// promises nothing across workgroups — it works because they share a CU's L1, and would stop working under tgsplit / another CU mode.
// Correctness never depends on a loaded value: the row only feeds s_setprio and (MUAVTA_PACE_HOLD builds) a sleep loop whose polls are
// bounded by MUAVTA_PACE_HOLD_POLLS; a build that holds on these values uses AGENT scope.  -DMUAVTA_PACE_SCOPE overrides.
#ifndef MUAVTA_PACE_SCOPE
#if MUAVTA_PACE_HOLD
#define MUAVTA_PACE_SCOPE __HIP_MEMORY_SCOPE_AGENT
#else
#define MUAVTA_PACE_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
#endif
static_assert(MUAVTA_PACE_HOLD_POLLS > 0 && MUAVTA_PACE_HOLD_POLLS <= (1 << 16), "the hold loop of the pacing block must be poll-bounded");
#ifndef MUAVTA_PACE_EVERY
#define MUAVTA_PACE_EVERY 1  // publish / read / re-rank every n-th step (a power of two)
#endif
#if MUAVTA_PACE_PRIO
  constexpr bool PACED = TL::A <= 32 && !ABL(10);  // the 64-agent tile has one or two waves per SIMD: nothing to pace
  // Pacing: the envs whose waves share a SIMD advance at different speeds (replans, episode length), and the launch ends on
  // the SIMD whose last env runs alone, at a quarter of the SIMD's multi-wave throughput.  Each wave publishes its step
  // counter in a row of the SIMD it runs on (hardware ids), reads its neighbours' and asks for issue priority while nobody
  // on the SIMD is further behind, so that the co-resident envs finish together.  Timing only: results do not depend on it.
  const uint32_t hw_id = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc_id = __builtin_amdgcn_s_getreg(20 | (3 << 11));
  uint32_t* pace_row = as_global(ctx.pace) + ((((xcc_id & 15u) << 12) | ((hw_id >> 4) & 0xFFFu)) << 4);
  const uint32_t pace_tag = (uint32_t)epoch << 16;
#endif
#ifdef MUAVTA_PROF
  sim.prof_begin();
#endif
#ifdef MUAVTA_DIAG_TIMES
  if (threadIdx.x == 0 && env < 65536) as_global(ctx.pace)[(1u << 19) + env] = (uint32_t)__builtin_amdgcn_s_memrealtime();
#endif
  if (seeds) {
    sim.reset(seeds[env], seedbuf + (size_t)env * 4 * 624);
  } else {
    copy16(L.S, blob, sizeof(EnvState<TL>));
    lds_sync();
  }
  uint32_t lds_base = (uint32_t)(uintptr_t)(AS3 unsigned char*)lds_own;
#if !MUAVTA_PHASE_INLINED
  asm volatile("" : "+v"(lds_base));  // opaque: keeps constant propagation from re-introducing the symbol into the callee
#endif
  // schedule: [allocate] , n_steps x [step, observation, next allocate] , [observation if it was not written per step] —
  // driven through ONE call site so that an inlined build carries one copy of the body
  for (int k = n_steps > 0 ? 0 : n_steps + 1; k <= n_steps + 1; k++) {
    int ph;
    if (k == 0) ph = PH_ALLOC;
    else if (k <= n_steps) {
      if (L.S->terminated || L.S->truncated) {  // uniform: read from LDS after a barrier
        if constexpr (REC) {
          // the episode ended during step k - 1: the allocate of that step was skipped, so token slots k-1 .. n_steps-1 have no
          // sample.  They become all-pad rows with replanned = 0 and carry the final S_WPS forward (zero step reward), instead
          // of staying uninitialised ring memory (the observation rings mark theirs with MUAVTA_OBS_UNWRITTEN).
          if (rec.K.task_feats) {
            const typename Sim<TL>::TokPtrs K = global_tok_ptrs<TL>(rec.K);
            const double fin = sim.s_wps();
            for (int sl = k - 1; sl < n_steps; sl++) {
              sim.tokens_pad(K, sl * rec.n_envs + env);
              if (threadIdx.x == 0) as_global(rec.s_wps)[(size_t)sl * rec.n_envs + env] = fin;
            }
          }
        }
        k = n_steps;
        continue;
      }
      ph = PH_STEP | (write_obs ? PH_OBS : 0) | (k < n_steps ? PH_ALLOC : 0);
    } else ph = (write_obs && !(REC && rec.O.tasks)) ? 0 : PH_OBS;  // with observation rings the handle's buffer gets the final one
#if MUAVTA_PACE_PRIO
    uint32_t seen = 0;
    if (PACED && k >= 1 && k <= n_steps && (k & (MUAVTA_PACE_EVERY - 1)) == 0) {
      if (threadIdx.x == 0) __hip_atomic_store(pace_row + (hw_id & 15u), pace_tag | (uint32_t)k, __ATOMIC_RELAXED, MUAVTA_PACE_SCOPE);
      if (threadIdx.x < 16) seen = __hip_atomic_load(pace_row + threadIdx.x, __ATOMIC_RELAXED, MUAVTA_PACE_SCOPE);
    }
#endif
    if (ph) rollout_phase<TL, REC>(ctxp, lds_own, lds_base, env, ph, interval, use_vis, mode, rec, k < n_steps ? k : n_steps,
                                   (REC && rec.O.tasks && k >= 1 && k <= n_steps) ? k - 1 : -1);
#if MUAVTA_PACE_PRIO
    if (PACED && k >= 1 && k <= n_steps && (k & (MUAVTA_PACE_EVERY - 1)) == 0) {  // consumed a step later: the load's latency stays off the env's dependent chain
      const bool behind_me = threadIdx.x < 16 && (seen >> 16) == (uint32_t)epoch && (seen & 0xFFFFu) < (uint32_t)k;
#if MUAVTA_PACE_PRIO == 4  // ranked: 3 for the last env of the SIMD, one less per neighbour that is further behind
      const int n_behind = __popcll(__ballot(behind_me));
      if (n_behind == 0) __builtin_amdgcn_s_setprio(3); else if (n_behind == 1) __builtin_amdgcn_s_setprio(2);
      else if (n_behind == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#else
      if (__ballot(behind_me) != 0ull) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(MUAVTA_PACE_PRIO);
#endif
#if MUAVTA_PACE_HOLD
      // Holding: a wave more than MUAVTA_PACE_HOLD steps AHEAD of a neighbour on its SIMD sleeps until that neighbour has caught up.
      // A wave is latency-bound (it uses about a third of the SIMD's issue slots and runs 1.33x faster alone than as one of
      // four), so an env with heavier steps falls behind whatever its priority and ends the launch running alone; the slots its
      // neighbours give up while they wait are the only thing that speeds it up.  The furthest-behind wave never waits (progress),
      // neighbours further behind than the window are ignored (a later workgroup that took over a wave slot of a multi-round
      // launch), and the polls are bounded.  Timing only.
      for (int polls = 0; polls < MUAVTA_PACE_HOLD_POLLS; polls++) {
        const uint32_t st = seen & 0xFFFFu;
        const bool wait_for = threadIdx.x < 16 && (seen >> 16) == (uint32_t)epoch && st + MUAVTA_PACE_HOLD < (uint32_t)k &&
                              st + MUAVTA_PACE_HOLD_WINDOW >= (uint32_t)k;
        if (__ballot(wait_for) == 0ull) break;
        __builtin_amdgcn_s_sleep(MUAVTA_PACE_HOLD_SLEEP);
        if (threadIdx.x < 16) seen = __hip_atomic_load(pace_row + threadIdx.x, __ATOMIC_RELAXED, MUAVTA_PACE_SCOPE);
      }
#endif
    }
#endif
  }
#if MUAVTA_PACE_PRIO
  if (PACED && threadIdx.x == 0) __hip_atomic_store(pace_row + (hw_id & 15u), pace_tag | 0xFFFFu, __ATOMIC_RELAXED, MUAVTA_PACE_SCOPE);
  __builtin_amdgcn_s_setprio(0);
#endif
  if (REC && rec.K.task_feats) {  // S_WPS after the last step closes the reward series
    if (threadIdx.x == 0) as_global(rec.s_wps)[(size_t)n_steps * rec.n_envs + env] = sim.s_wps();
  }
  sim.sync_clock();
  sim.metrics(as_global(metrics) + (size_t)env * MUAVTA_N_METRICS);
  lds_sync();
  copy16(blob, L.S, sizeof(EnvState<TL>));
#ifdef MUAVTA_PROF
  sim.prof_flush(env);
#endif
#ifdef MUAVTA_DIAG_TIMES  // tools/end_times_probe.py: when each env's wave ended and on which SIMD (rows of the pace table no SIMD key reaches)
  if (threadIdx.x == 0 && env < 65536) {
    as_global(ctx.pace)[(1u << 19) + 65536 + env] = (uint32_t)__builtin_amdgcn_s_memrealtime();
    as_global(ctx.pace)[(1u << 19) + 131072 + env] = __builtin_amdgcn_s_getreg(4 | (31 << 11)) | (__builtin_amdgcn_s_getreg(20 | (3 << 11)) << 16);
  }
#endif
}

template <class TL>
__global__ __launch_bounds__(WG) void k_metrics(const DevCtx* __restrict__ ctxp, double* metrics) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = blockIdx.x;
  Lds<TL> L(smem);
  copy16(L.S, blob_of<TL>(ctx, env), sizeof(EnvState<TL>));
  lds_sync();
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, nullptr);
  sim.metrics(metrics + (size_t)env * MUAVTA_N_METRICS);
}

template <class TL>
__global__ __launch_bounds__(WG) void k_observe(const DevCtx* __restrict__ ctxp) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = blockIdx.x;
  Lds<TL> L(smem);
  copy16(L.S, blob_of<TL>(ctx, env), sizeof(EnvState<TL>));
  lds_sync();
  // muavta_refresh_observation is the FULL rewrite of the handle's observation buffers: pad rows included, whatever the record
  // believes they hold (a caller may have touched the zero-copy views of muavta_device_ptrs in place)
  if (threadIdx.x == 0) L.S->obs_rows = -1;
  lds_sync();
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, nullptr);
  obs_for_env(sim, ctx.P, obs_ptrs(ctx), env);
  lds_sync();
  if (threadIdx.x == 0) {  // the only fields this kernel changes in the record: which rows of the buffer hold pad rows now, and the
    EnvState<TL>* blob = blob_of<TL>(ctx, env);   // task times it rebuilt on the way (refresh_task_times)
    blob->obs_rows = L.S->obs_rows;
    blob->times_dirty = L.S->times_dirty;
  }
}

template <class TL>
__global__ __launch_bounds__(WG) void k_tokens(const DevCtx* __restrict__ ctxp, typename Sim<TL>::TokPtrs K) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = blockIdx.x;
  Lds<TL> L(smem);
  copy16(L.S, blob_of<TL>(ctx, env), sizeof(EnvState<TL>));
  lds_sync();
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, nullptr);
  sim.tokens(K, env);
}

template <class TL>
__global__ __launch_bounds__(WG) void k_context(const DevCtx* __restrict__ ctxp, int raw, int max_tasks, float* out) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = blockIdx.x;
  Lds<TL> L(smem);
  copy16(L.S, blob_of<TL>(ctx, env), sizeof(EnvState<TL>));
  lds_sync();
  Sim<TL> sim(*L.S, *cold_of<TL>(ctx, env), *L.X, ctx.P, nullptr);
  sim.context(raw, max_tasks, as_global(out) + (size_t)env * (raw ? 1 : 8));
}

// muavta_call: one of the reference's out-of-step mutators on ONE env, with the device routines step() itself uses.
struct CallArgs { int32_t op, env, i[8]; double d; };
template <class TL>
__global__ __launch_bounds__(WG) void k_call(const DevCtx* __restrict__ ctxp, CallArgs a, int32_t* out) {
  const DevCtx& ctx = ctx_ref(ctxp);
  const int env = a.env, lane = threadIdx.x;
  Lds<TL> L(smem);
  EnvState<TL>& S = *L.S;
  EnvState<TL>* blob = blob_of<TL>(ctx, env);
  copy16(L.S, blob, sizeof(EnvState<TL>));
  lds_sync();
  Sim<TL> sim(S, *cold_of<TL>(ctx, env), *L.X, ctx.P, tape_of(ctx, env));
  auto slot_of = [&](int id) -> int {  // uniform: the live slot holding task `id`, or -1
    int found = -1;
    for (int base = 0; base < TL::T; base += WG) {
      const int s = base + lane;
      const unsigned long long m = __ballot(id > 0 && s < TL::T && S.t_id[s] == id);
      if (m) found = base + __ffsll((long long)m) - 1;
    }
    return found;
  };
  for (int k = lane; k < MUAVTA_CALL_OUT; k += WG) out[k] = 0;
  __syncthreads();
  const int ag = a.i[0];
  switch (a.op) {
    case MUAVTA_OP_UAV_ALLOCATE: {
      const int s = slot_of(a.i[1]);
      sim.tnow = a.i[2];
      if (lane == 0) out[0] = (s >= 0 && sim.uav_allocate(ag, s)) ? 1 : 0;
    } break;
    case MUAVTA_OP_CREATE_ESCORT: {
      const int s = slot_of(a.i[1]);
      if (lane == 0) {
        S.list_stale = 1;  // (a task may be born behind last_tasks_info: Sim::allocate extends its list)
        if (ctx.P.escort_enabled && (s >= 0 || a.i[1] == 0)) sim.create_escort_for(ag, s);  // rec_task None (id 0): protected_task = None
        const int k = sim.escort_lookup(ag);
        out[0] = (ctx.P.escort_enabled && k >= 0) ? (int)S.esc_id[k] : -1;
      }
    } break;
    case MUAVTA_OP_SYNC_ESCORTS:
      if (lane == 0) S.list_stale = 1;
      lds_sync();
      if (ctx.P.escort_enabled) sim.sync_escorts_coop();
      break;
    case MUAVTA_OP_RETIRE_ESCORT:
      if (lane == 0)
        for (int k = 0; k < S.n_escorts; k++)
          if (S.esc_id[k] == a.i[0]) { sim.retire_escort_entry(k, a.i[1] != 0); break; }
      break;
    case MUAVTA_OP_ESCORT_FIGHTERS_NEAR:
      if (lane == 0) {
        int16_t* who = L.X->remaining;
        const int n = sim.escort_fighters_near(ag, a.d < 0 ? ctx.P.escort_radius : a.d, who, L.X->v);
        out[0] = n;
        for (int k = 0; k < n && k + 1 < MUAVTA_CALL_OUT; k++) out[1 + k] = who[k];
      }
      break;
    case MUAVTA_OP_ACTION_VALID: {
      const int s = slot_of(a.i[1]);
      if (lane == 0) out[0] = (s >= 0 && sim.action_valid(ag, s)) ? 1 : 0;
    } break;
    case MUAVTA_OP_SET_QUEUE: {
      int slots[6];
      for (int k = 0; k < 6; k++) slots[k] = (k < a.i[1]) ? slot_of(a.i[2 + k]) : -1;
      if (lane == 0) {
        int n = 0;
        for (int k = 0; k < a.i[1] && k < 6 && n < TL::Q; k++) {
          if (a.i[2 + k] == 0 || slots[k] < 0) continue;  // task_idle, or a task that is no longer resident
          S.a_qid[ag][n] = (i16)a.i[2 + k]; S.a_qslot[ag][n] = (i8)slots[k]; sim.C.a_qtime[ag][n] = 0.0;
          n++;
        }
        S.a_qlen[ag] = (i8)n;
        if (a.i[7] == 1 && n == 0) {  // UAV.allocate(task_idle) (DroneEnvComponents.py:59-60,85-92): more than the list assignment
          S.a_reeval[ag] = 0; S.a_last_id[ag] = -1; S.a_last_slot[ag] = -1;
          sim.qs().a_nft[ag] = 0.0; sim.qs().a_nfx[ag] = S.a_px[ag]; sim.qs().a_nfy[ag] = S.a_py[ag];
        }
      }
    } break;
    default: break;
  }
  cold_sync();
  sim.refresh_task_times();  // initTime / doneTime follow the allocationDetails the call may have changed
  lds_sync();
  copy16(blob, L.S, sizeof(EnvState<TL>));
}

// Stand-alone LSAP: one problem per workgroup, cost tile staged in LDS (transposed when nc < nr).
// REG: the register-resident solver of the allocator path (rows <= 32, columns <= 64); else the LDS solver (64 x 128).
typedef Tile<32, 64, 16, 16, 16, 8> TileLsapReg;
typedef Tile<64, 128, 16, 16, 16, 8> TileLsapLds;  // keeps the full cost tile in LDS (the env's 64x128 tile evaluates costs on the fly)
template <class TL, bool REG>
__global__ __launch_bounds__(WG) void k_lsap(const double* cost, int nr, int nc, int64_t* row, int64_t* col, int32_t* status) {
  Lds<TL> L(smem);
  DevParams dummy;
  Sim<TL> sim(*L.S, *reinterpret_cast<EnvCold<TL>*>(L.S) /* never touched by the solver */, *L.X, dummy, nullptr);
  const int prob = blockIdx.x;
  const double* c = cost + (size_t)prob * nr * nc;
  const bool tr = nc < nr;
  const int Rr = tr ? nc : nr, Cc = tr ? nr : nc;
  for (int p = threadIdx.x; p < nr * nc; p += WG) {
    int i = p / nc, j = p - i * nc;
    L.X->cost[tr ? (j * Cc + i) : (i * Cc + j)] = c[p];
  }
  if (threadIdx.x == 0) L.S->error = 0;
  lds_sync();
  if constexpr (REG) sim.lsap_reg(Rr, Cc); else sim.lsap(Rr, Cc);
  if (threadIdx.x == 0) {
    status[prob] = L.S->error;  // MUAVTA_ERR_LSAP: no finite assignment (scipy: "cost matrix is infeasible")
    int64_t* r = row + (size_t)prob * Rr;
    int64_t* cc = col + (size_t)prob * Rr;
    int n = 0;
    if (!tr) { for (int i = 0; i < nr; i++) { r[n] = i; cc[n] = L.X->col4row[i]; n++; } }
    else { for (int i = 0; i < nr; i++) if (L.X->row4col[i] >= 0) { r[n] = i; cc[n] = L.X->row4col[i]; n++; } }
  }
}

__global__ void k_domain_math(const double* x, const double* y, int n, double* out_sqrt, double* out_div, double* out_div2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out_sqrt[i] = fsqrt(x[i]);
  out_div[i] = fdiv(x[i], y[i]);
  const double r = frcp_nr(y[i]);
  out_div2[i] = fdiv_r(-x[i], y[i], r);
}

__global__ void k_libm_log(const double* x, int n, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = libm_log(x[i]);
}

__global__ void k_libm_atan2(const double* y, const double* x, int n, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = libm_atan2(y[i], x[i]);
}

__global__ void k_avoid(const double* pos, const double* mov, int n, const double* obst, int n_obs, double* out) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // same arithmetic as Sim::avoid_obstacles (core_sim/src/sim_core.rs:25-59)
  double px = pos[2 * i], py = pos[2 * i + 1], mx = mov[2 * i], my = mov[2 * i + 1];
  double ax = 0.0, ay = 0.0;
  const double PI = 3.14159265358979323846;
  for (int o = 0; o < n_obs; o++) {
    double dx = obst[3 * o] - px, dy = obst[3 * o + 1] - py;
    double d_zone = sqrt(dx * dx + dy * dy) - obst[3 * o + 2];
    if (d_zone < 40.0) {
      double nx = dx / d_zone, ny = dy / d_zone;
      double force = libm_log(fmax(1.05, d_zone));
      force = 0.5 / (1.0 - force);
      double ang = libm_atan2(my, mx) - libm_atan2(dy, dx);
      ang = fmod(ang + PI, 2.0 * PI) - PI;
      double rx, ry;
      if (ang > 0.0) { rx = ny; ry = -nx; } else { rx = -ny; ry = nx; }
      ax += rx * force;
      ay += ry * force;
    }
  }
  out[2 * i] = ax;
  out[2 * i + 1] = ay;
}

// ====================================================================================================
// Host side
// ====================================================================================================
thread_local std::string g_create_error;

#define HIPCHK(env, expr)                                                                         \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      (env)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                             \
      return MUAVTA_E_HIP;                                                                        \
    }                                                                                             \
  } while (0)

enum TileKind { TK16 = 0, TK24 = 1, TK64 = 2 };

// An ABI call runs on its handle's device and leaves the calling thread's current device as it found it (a caller that
// mixes this library with torch on another device must not have its current device changed under it).
struct DeviceScope {
  int prev = -1, want;
  explicit DeviceScope(int d) : want(d) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want) (void)hipSetDevice(want);
  }
  ~DeviceScope() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

}  // namespace

struct MuavtaEnv {
  DevParams P;
  MuavtaParams params;
  int tile = TK16;
  int alloc_mode = 0;  // MUAVTA_ALLOC_*
  void* d_tok = nullptr;  // muavta_tokens staging (host-buffer variant)
  double* d_rel = nullptr;  // release log [N, 1 + MUAVTA_REL_ROW*T] (muavta_set_release_log)
  // Seeding pipeline: seeds upload + k_seed run on their own stream into one of two slots, so that the seeding of launch
  // i+1 overlaps launch i (k_seed uses no LDS and few registers: its waves run next to the rollout's where a SIMD has room).
  uint32_t* d_seedbuf[2] = {nullptr, nullptr};  // [N][4][624] init_by_array states (k_seed)
  uint32_t* d_seedtmp = nullptr;                // k_seed's lane-interleaved scratch
  uint64_t* h_seeds[2] = {nullptr, nullptr};    // pinned staging of the caller's seeds
  hipStream_t seed_stream = nullptr;
  hipEvent_t ev_seed0[2] = {nullptr, nullptr}, ev_seeded[2] = {nullptr, nullptr}, ev_consumed[2] = {nullptr, nullptr};
  bool seed_used[2] = {false, false};
  unsigned seed_seq = 0;
  int last_seed_slot = 0;
  size_t tok_bytes = 0;
  int n_envs = 0, device = 0;
  int A = 0, T = 0, H = 0, E = 0, R = 0, Q = 0;
  size_t state_bytes = 0, cold_bytes = 0, lds_bytes = 0;  // per env: LDS image (EnvState), HBM-only part (EnvCold)
  void* blobs = nullptr;
  void* cold = nullptr;
  uint32_t* tapes = nullptr;
  DevCtx* d_ctx = nullptr;  // device copy of {P, O, tapes}
  uint32_t* d_pace = nullptr;
  uint32_t pace_epoch = 0;
  enum { REC_SLOT = 256 };
  void* d_rec = nullptr;  // [2][REC_SLOT]: slot 0 all zero (plain rollouts), slot 1 the RecordPtrs of the muavta_rollout_record launch in flight
  uint64_t* d_seeds[2] = {nullptr, nullptr};
  int32_t *d_act_agent = nullptr, *d_act_index = nullptr, *d_call_out = nullptr;
  int32_t *d_list_agent = nullptr, *d_list_index = nullptr;  // muavta_step_lists rows [N][list_cap] (grown on demand)
  void* d_run = nullptr;  // muavta_step_run's outputs: [N] f64 reward sums | [N] i32 steps taken | [N] u8 park flags
  int list_cap = 0;
  ncclComm_t comm = nullptr;  // muavta_comm_init
  int comm_rank = 0, comm_ranks = 0;
  void* d_comm = nullptr;     // [64 f64 send | 64 x n_ranks f64 recv | 64 i64 send | 64 i64 recv]
  double* d_metrics = nullptr;
  ObsPtrs O{};
  hipStream_t stream = nullptr;
  // sub-batches (muavta_set_parts): contiguous env ranges, each stepped on its own stream so that the host can decide for one part
  // while the device steps another, and so that one part's slowest env does not hold up the others' launches
  enum { MAX_PARTS = 8 };
  int n_parts = 0;
  hipStream_t part_stream[MAX_PARTS] = {};
  hipEvent_t part_ev[MAX_PARTS] = {};
  hipEvent_t ev_fork = nullptr;
  bool part_busy[MAX_PARTS] = {};         // the part's stream holds work the main stream has not been ordered after yet
  bool part_fork_needed[MAX_PARTS] = {};  // the main stream got work since the part's stream last waited for it
  int32_t *d_part_agent = nullptr, *d_part_index = nullptr;  // action staging of the parts (one [N, A] pair, each part its rows)
  enum { EV_RING = 64 };
  hipEvent_t ev0[EV_RING] = {}, ev1[EV_RING] = {};  // ev0[i] .. ev1[i]: the k_rollout launch number i (mod EV_RING)
  unsigned long long n_rollouts = 0;
  bool timing_stale = false;  // a *_part rollout ran since the last whole-batch one: the event ring describes an older launch
  float last_ms = 0.f;
  bool last_seeded = false;
  bool did_reset = false;
  std::vector<unsigned char> host_blobs, host_cold;  // cache for muavta_get
  bool host_valid = false;
  std::string err;
  // ---- state lanes (muavta_set_lanes) -----------------------------------------------------------------------------------------------
  // Everything above is ONE lane: the env records, tapes, observation buffers, metrics, streams, seeding slots and event rings of a batch.
  // A handle may own a second one (`hl.twin`, a complete MuavtaEnv of the same configuration that no caller ever sees): a seeded rollout
  // issued while the previous one is still running goes to the other lane — launch i + 1's workgroups start in the wave slots launch i's
  // early finishers free instead of waiting for its slowest env.  A flip SWAPS the two objects' contents (everything but `hl` and the
  // communicator), so every entry point keeps working on `*e` = the lane of the latest seeded rollout, without routing.
  int lane_id = 0;  // travels with the lane's contents
  struct HandleLevel {
    MuavtaEnv* twin = nullptr;
    int lanes_mode = 0;  // 0 auto (second lane on demand), 1 one lane only, 2 always alternate
    bool twin_failed = false;
    enum { RING = 64 };
    unsigned char ring_lane[RING] = {};         // rollout launch k (mod RING) of the HANDLE ran on this lane ...
    unsigned long long ring_no[RING] = {};      // ... as that lane's launch number
    unsigned long long n_launches = 0;
  } hl;
};
static int join_parts(MuavtaEnv* e);  // (sub-batches: defined with the other part helpers in front of the C ABI)
extern "C" int muavta_set_parts(MuavtaEnv* e, int32_t n_parts);
extern "C" int muavta_set_release_log(MuavtaEnv* e, int32_t enable);
extern "C" int muavta_create(const MuavtaParams* params, int32_t n_envs, int32_t device, MuavtaEnv** out);
extern "C" int muavta_destroy(MuavtaEnv* e);
extern "C" int muavta_set_slot_cap(MuavtaEnv* e, int32_t cap);

namespace {

template <class TL>
int launch_attr(MuavtaEnv* e) {
  size_t lds = Lds<TL>::bytes();
  e->lds_bytes = lds;
  if (lds > 48 * 1024) {
    const void* ks[] = {reinterpret_cast<const void*>(&k_reset<TL>), reinterpret_cast<const void*>(&k_step<TL>), reinterpret_cast<const void*>(&k_allocate<TL>),
                        reinterpret_cast<const void*>(&k_rollout<TL, false>), reinterpret_cast<const void*>(&k_rollout<TL, true>), reinterpret_cast<const void*>(&k_metrics<TL>), reinterpret_cast<const void*>(&k_observe<TL>),
                        reinterpret_cast<const void*>(&k_tokens<TL>), reinterpret_cast<const void*>(&k_call<TL>), reinterpret_cast<const void*>(&k_context<TL>)};
    for (const void* k : ks) HIPCHK(e, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIPCHK(e, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_allocate_scored<TL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds + SCORED_EXTRA_LDS));
  }
  return MUAVTA_OK;
}

#define DISPATCH(e, CALL)                    \
  switch ((e)->tile) {                       \
    case TK16: { typedef Tile16 TL; CALL; } break; \
    case TK24: { typedef Tile24 TL; CALL; } break; \
    default:   { typedef Tile64 TL; CALL; } break; \
  }

template <class TL>
static void launch_tokens(MuavtaEnv* e, int kind, int max_tasks, int max_agents, float* task_feats, uint8_t* task_mask, int32_t* task_ids,
                          float* agent_feats, uint8_t* agent_mask, int32_t* agent_ids, float* edge_valid, int32_t* n_urgent, float* expert_mask,
                          int32_t* replanned) {
  typename Sim<TL>::TokPtrs K{task_feats, task_mask, task_ids, agent_feats, agent_mask, agent_ids, edge_valid, n_urgent, expert_mask, replanned,
                              kind, max_tasks, max_agents};
  hipLaunchKernelGGL(k_tokens<TL>, dim3(e->n_envs), dim3(WG), Lds<TL>::bytes(), e->stream, (const DevCtx*)e->d_ctx, K);
}

template <class TL>
static void launch_rl_step(MuavtaEnv* e, const ScoredDev& sc, const MuavtaRlStep* rs, hipStream_t stream, int first, int count) {
  typename Sim<TL>::TokPtrs K{rs->task_feats, rs->task_mask, rs->task_ids, rs->agent_feats, rs->agent_mask, rs->agent_ids, rs->edge_valid, rs->n_urgent,
                              nullptr, nullptr, rs->plan.kind, rs->plan.max_tasks, rs->plan.max_agents};
  hipLaunchKernelGGL(k_rl_step<TL>, dim3(count), dim3(WG), 0, stream, (const DevCtx*)e->d_ctx, sc, K, rs->plan.replan_interval, rs->plan.use_visibility,
                     rs->write_obs, rs->s_wps, rs->done, e->n_envs, first);
}

template <class TL>
static void launch_run(MuavtaEnv* e, int src, const ScoredDev& sc, const MuavtaRlStep* rs, const MuavtaRlRun* rr, const RunOut& R, int gate, int interval, int use_vis,
                       int write_obs, int max_steps, const int32_t* da, const int32_t* di, int cap, hipStream_t stream, int first, int count) {
  RunArgs<TL> G;
  memset(&G, 0, sizeof(G));
  G.sc = sc; G.R = R;
  if (rs) {
    typename Sim<TL>::TokPtrs k{rs->task_feats, rs->task_mask, rs->task_ids, rs->agent_feats, rs->agent_mask, rs->agent_ids, rs->edge_valid, rs->n_urgent,
                                nullptr, nullptr, rs->plan.kind, rs->plan.max_tasks, rs->plan.max_agents};
    G.K = k;
  }
  if (rr && rs) {
    typename Sim<TL>::TokPtrs k{rr->park_task_feats, rr->park_task_mask, rr->park_task_ids, rr->park_agent_feats, rr->park_agent_mask, rr->park_agent_ids, rr->park_edge_valid,
                                rr->park_n_urgent, nullptr, nullptr, rs->plan.kind, rs->plan.max_tasks, rs->plan.max_agents};
    G.KP = k;
  }
  hipLaunchKernelGGL(k_run<TL>, dim3(count), dim3(WG), 0, stream, G, (const DevCtx*)e->d_ctx, src, gate, interval, use_vis, write_obs, max_steps, da, di, cap, e->n_envs, first);
}

int fill_dev_params(const MuavtaParams* p, DevParams* d, std::string* err) {
  memset(d, 0, sizeof(*d));
  if (p->abi_version != MUAVTA_ABI_VERSION) { *err = "abi_version mismatch"; return MUAVTA_E_ARG; }
  if (p->n_agent_groups < 1 || p->n_agent_groups > MUAVTA_MAX_GROUPS || p->n_task_groups < 0 || p->n_task_groups > MUAVTA_MAX_GROUPS ||
      p->n_threat_groups < 0 || p->n_threat_groups > MUAVTA_MAX_GROUPS) { *err = "group counts out of range"; return MUAVTA_E_ARG; }
  d->n_agent_groups = p->n_agent_groups; d->n_task_groups = p->n_task_groups; d->n_threat_groups = p->n_threat_groups;
  int nA = 0, nT = 0, nH = 0;
  double possible = 0;  // DroneEnv.py:670-675: summed over the static tasks only (Det tasks are created later, :685)
  for (int g = 0; g < p->n_agent_groups; g++) {
    if (p->agent_type[g] < 0 || p->agent_type[g] > MUAVTA_F2 || p->agent_count[g] < 0) { *err = "bad agent group"; return MUAVTA_E_ARG; }
    d->agent_type[g] = p->agent_type[g]; d->agent_count[g] = p->agent_count[g]; nA += p->agent_count[g];
  }
  for (int g = 0; g < p->n_task_groups; g++) {
    int ty = p->task_type[g];
    if (ty != MUAVTA_HOLD && ty != MUAVTA_REC && ty != MUAVTA_ATT) { *err = "static task types are Hold/Rec/Att"; return MUAVTA_E_ARG; }
    d->task_type[g] = ty; d->task_count[g] = p->task_count[g]; nT += p->task_count[g];
    for (int i = 0; i < p->task_count[g]; i++) possible += 1.0;
  }
  for (int g = 0; g < p->n_threat_groups; g++) {
    int ty = p->threat_type[g];
    if (ty != MUAVTA_T1 && ty != MUAVTA_T2) { *err = "threat types are T1/T2"; return MUAVTA_E_ARG; }
    d->threat_type[g] = ty; d->threat_count[g] = p->threat_count[g]; nH += p->threat_count[g];
  }
  if (nA < 1) { *err = "no agents"; return MUAVTA_E_ARG; }
  if (p->num_obstacles < 0 || p->num_obstacles > 8) { *err = "num_obstacles must be in 0..8"; return MUAVTA_E_ARG; }
  if (p->max_time_steps < 1) { *err = "max_time_steps must be >= 1"; return MUAVTA_E_ARG; }
  // (agent speeds = MAX_SPEED / frame_rate * 0.02 are divisors of the kernels' range-restricted division, see fdiv)
  if (!(p->simulation_frame_rate >= 1e-9 && p->simulation_frame_rate <= 1e9)) { *err = "simulation_frame_rate must be in [1e-9, 1e9]"; return MUAVTA_E_ARG; }
  // time steps, deadlines (t + window_length), reveal times (t + threat_delay), commit locks (t + commit_horizon) and task
  // ids (a few per step) are stored in 16 bits on the device
  if (p->max_time_steps > 20000 || p->window_length > 10000 || p->threat_delay > 10000 || p->commit_horizon > 10000 || p->window_length < -10000 ||
      p->threat_delay < -10000 || p->commit_horizon < -10000) { *err = "max_time_steps <= 20000 and window_length / threat_delay / commit_horizon within +-10000"; return MUAVTA_E_ARG; }
  d->n_agents = nA; d->n_tasks = nT + 1; d->max_tasks = d->n_tasks + 28; d->n_threats = nH;
  d->max_time_steps = p->max_time_steps; d->multiple_tasks_per_agent = p->multiple_tasks_per_agent;
  d->early_terminate = p->early_terminate; d->capability_mask = p->capability_mask; d->saturate_mask = p->saturate_mask;
  d->include_time_windows = p->include_time_windows; d->threat_delay = p->threat_delay; d->hard_windows = p->hard_windows;
  d->window_length = p->window_length; d->burst_mode = p->burst_mode; d->burst_size = p->burst_size;
  d->dual_region_bursts = p->dual_region_bursts; d->share_knowledge = p->share_knowledge; d->escort_enabled = p->escort_enabled;
  d->num_obstacles = p->num_obstacles; d->random_init_pos = p->random_init_pos;
  int need = (int)std::ceil(p->escort_requirement);
  d->escort_required_agents = need > 2 ? need : 2;
  d->escort_mask = p->escort_agent_type_mask;
  d->commit_horizon = p->commit_horizon;
  static const double MAX_SPEED[7] = {5.0, 8.0, 5.0, 20.0, 15.0, 14.0, 12.0};  // MultiDroneEnvData.py:32-38
  for (int t = 0; t < 7; t++) d->speed[t] = MAX_SPEED[t] / p->simulation_frame_rate * 0.02;
  d->threat_prob = 0.7 / p->simulation_frame_rate * 0.02;
  d->reward_norm_factor = (possible * 1 + possible) / 1000;
  // sqrt is correctly rounded and monotone, so `sqrt(v) <= r` is a threshold test on v; find the threshold
  auto sq_bound = [](double r) {
    double v = r * r;
    if (r > 0) {
      while (std::sqrt(v) > r) v = std::nextafter(v, 0.0);
      while (std::sqrt(std::nextafter(v, INFINITY)) <= r) v = std::nextafter(v, INFINITY);
    }
    return v;
  };
  d->sense_sq_bound = sq_bound(p->sense_radius);
  d->escort_sq_bound = sq_bound(p->escort_radius);
  d->fail_rate = p->fail_rate; d->arrival_rate = p->arrival_rate; d->dynamic_idle_penalty = p->dynamic_idle_penalty;
  d->sense_radius = p->sense_radius; d->miss_penalty = p->miss_penalty; d->on_time_bonus = p->on_time_bonus;
  d->reassign_penalty = p->reassign_penalty; d->escort_radius = p->escort_radius; d->escort_requirement = p->escort_requirement;
  d->escort_intercept_radius = p->escort_intercept_radius; d->mutual_support_radius = p->mutual_support_radius;
  for (int i = 0; i < 8; i++) d->rw[i] = p->reward_weights[i];
  d->rw_plain = d->reward_norm_factor > 0 ? 1 : 0;
  for (int i = 0; i < 8; i++) if (!(p->reward_weights[i] >= 0 && std::isfinite(p->reward_weights[i]))) d->rw_plain = 0;
  d->inv_mts = 1.0 / (double)(d->max_time_steps > 1 ? d->max_time_steps : 1);
  d->inv_max_tasks = 1.0 / (double)(d->max_tasks > 1 ? d->max_tasks : 1);
  return MUAVTA_OK;
}

template <class TL>
size_t blob_bytes() { return sizeof(EnvState<TL>); }

int sync_host(MuavtaEnv* e) {
  if (e->host_valid) return MUAVTA_OK;
  if (e->n_parts) { int rc_ = join_parts(e); if (rc_) return rc_; }
  e->host_blobs.resize((size_t)e->n_envs * e->state_bytes);
  e->host_cold.resize((size_t)e->n_envs * e->cold_bytes);
  HIPCHK(e, hipMemcpyAsync(e->host_blobs.data(), e->blobs, e->host_blobs.size(), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->host_cold.data(), e->cold, e->host_cold.size(), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->host_valid = true;
  return MUAVTA_OK;
}

// Gather one field out of the host copy of the blobs.
template <class TL>
int gather(MuavtaEnv* e, MuavtaField f, void* dst, size_t bytes, bool scatter) {
  typedef EnvState<TL> St;
  const int N = e->n_envs, A = e->P.n_agents, T = TL::T, H = e->P.n_threats, Q = TL::Q, E = TL::E, KW = TL::KW;
  St* blobs = reinterpret_cast<St*>(e->host_blobs.data());
  EnvCold<TL>* cold = reinterpret_cast<EnvCold<TL>*>(e->host_cold.data());
  size_t need = 0;
  auto chk = [&](size_t n) { need = n; return bytes == n; };
  double* D = (double*)dst;
  int32_t* I = (int32_t*)dst;
  uint32_t* U = (uint32_t*)dst;
  auto QS = [&](int n) -> QueueSide<TL::A, TL::T, true>& {  // where this tile keeps next_free_* / orgReqs / doneReqs
    if constexpr (TL::SLIM) return static_cast<QueueSide<TL::A, TL::T, true>&>(cold[n]); else return static_cast<QueueSide<TL::A, TL::T, true>&>(blobs[n]);
  };
#define BAD() do { e->err = "muavta_get/set: buffer size mismatch, need " + std::to_string(need) + " bytes"; return MUAVTA_E_ARG; } while (0)
#define RW(dstv, srcv) do { if (scatter) (srcv) = (dstv); else (dstv) = (srcv); } while (0)
  switch (f) {
    case MUAVTA_F_AGENT_POS:
      if (!chk((size_t)N * A * 2 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) { RW(D[((size_t)n * A + a) * 2], blobs[n].a_px[a]); RW(D[((size_t)n * A + a) * 2 + 1], blobs[n].a_py[a]); }
      break;
    case MUAVTA_F_AGENT_NFP:
      if (!chk((size_t)N * A * 2 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) { RW(D[((size_t)n * A + a) * 2], QS(n).a_nfx[a]); RW(D[((size_t)n * A + a) * 2 + 1], QS(n).a_nfy[a]); }
      break;
    case MUAVTA_F_AGENT_NFT:
      if (!chk((size_t)N * A * 8)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) RW(D[(size_t)n * A + a], QS(n).a_nft[a]);
      break;
    case MUAVTA_F_AGENT_DIST:
      if (!chk((size_t)N * A * 8)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) RW(D[(size_t)n * A + a], blobs[n].a_dist[a]);
      break;
    case MUAVTA_F_AGENT_CAPS:
      if (!chk((size_t)N * A * 6 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) for (int c = 0; c < 6; c++) RW(D[((size_t)n * A + a) * 6 + c], blobs[n].a_caps[c][a]);
      break;
    case MUAVTA_F_AGENT_STATE:
      if (!chk((size_t)N * A * 4)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) RW(I[(size_t)n * A + a], blobs[n].a_state[a]);
      break;
    case MUAVTA_F_AGENT_HEAD:
      if (!chk((size_t)N * A * 4)) BAD();
      if (scatter) { e->err = "AGENT_HEAD is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) I[(size_t)n * A + a] = blobs[n].a_qlen[a] > 0 ? blobs[n].a_qid[a][0] : 0;
      break;
    case MUAVTA_F_AGENT_QUEUE:
      if (!chk((size_t)N * A * Q * 4)) BAD();
      if (scatter) { e->err = "AGENT_QUEUE is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) for (int k = 0; k < Q; k++)
        I[((size_t)n * A + a) * Q + k] = k < blobs[n].a_qlen[a] ? blobs[n].a_qid[a][k] : (k == 0 ? 0 : -1);
      break;
    case MUAVTA_F_AGENT_ATTACK_CAP:
      if (!chk((size_t)N * A * 4)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) RW(I[(size_t)n * A + a], blobs[n].a_acap[a]);
      break;
    case MUAVTA_F_AGENT_TYPE:
      if (!chk((size_t)N * A * 4)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) RW(I[(size_t)n * A + a], blobs[n].a_type[a]);
      break;
    case MUAVTA_F_AGENT_NAME_IDX:
      if (!chk((size_t)N * A * 4)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) RW(I[(size_t)n * A + a], blobs[n].a_name[a]);
      break;
    case MUAVTA_F_AGENT_MISC:
      if (!chk((size_t)N * A * 6 * 4)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) {
        int32_t* r = I + ((size_t)n * A + a) * 6;
        RW(r[0], blobs[n].a_task_start[a]); RW(r[1], blobs[n].a_fail[a]); RW(r[2], blobs[n].a_reeval[a]);
        RW(r[3], blobs[n].a_last_id[a]); RW(r[4], blobs[n].a_commit[a]);
        if (!scatter) r[5] = blobs[n].a_qlen[a];
      }
      break;
    case MUAVTA_F_TASK_ID:
      if (!chk((size_t)N * T * 4)) BAD();
      if (scatter) { e->err = "TASK_ID is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) I[(size_t)n * T + s] = blobs[n].t_id[s];
      break;
    case MUAVTA_F_TASK_STATUS:
      if (!chk((size_t)N * T * 4)) BAD();
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) RW(I[(size_t)n * T + s], blobs[n].t_status[s]);
      break;
    case MUAVTA_F_TASK_POS:
      if (!chk((size_t)N * T * 2 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) { RW(D[((size_t)n * T + s) * 2], blobs[n].t_px[s]); RW(D[((size_t)n * T + s) * 2 + 1], blobs[n].t_py[s]); }
      break;
    case MUAVTA_F_TASK_CUR:
      if (!chk((size_t)N * T * 6 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) for (int c = 0; c < 6; c++) RW(D[((size_t)n * T + s) * 6 + c], cold[n].t_cur[c][s]);
      break;
    case MUAVTA_F_TASK_ALLOC:
      if (!chk((size_t)N * T * 6 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) for (int c = 0; c < 6; c++) RW(D[((size_t)n * T + s) * 6 + c], cold[n].t_alloc[c][s]);
      break;
    case MUAVTA_F_TASK_ORG_DONE:
      if (!chk((size_t)N * T * 2 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) { RW(D[((size_t)n * T + s) * 2], QS(n).t_org[s]); RW(D[((size_t)n * T + s) * 2 + 1], QS(n).t_done[s]); }
      break;
    case MUAVTA_F_TASK_TIMES:
      if (!chk((size_t)N * T * 2 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) { RW(D[((size_t)n * T + s) * 2], cold[n].t_init[s]); RW(D[((size_t)n * T + s) * 2 + 1], cold[n].t_dtime[s]); }
      break;
    case MUAVTA_F_TASK_META:
      if (!chk((size_t)N * T * 8 * 4)) BAD();
      for (int n = 0; n < N; n++) for (int s = 0; s < T; s++) {
        int32_t* r = I + ((size_t)n * T + s) * 8;
        St& b = blobs[n];
        if (scatter) { b.t_required[s] = r[3]; continue; }  // required_agents is the only field callers write (test_escort.py:107)
        r[0] = b.t_type[s]; r[1] = (b.t_flags[s] & TF_DEADLINE) ? b.t_deadline[s] : -1; r[2] = b.t_created[s]; r[3] = b.t_required[s];
        r[4] = (b.t_flags[s] & TF_ESCORT) ? 1 : 0; r[5] = b.t_ndet[s]; r[6] = b.t_prot_agent[s];
        r[7] = (b.t_flags[s] & TF_ELIGIBLE) ? (int32_t)b.t_elig[s] : -1;
      }
      break;
    case MUAVTA_F_KNOWN:
      if (!chk((size_t)N * A * KW * 4)) BAD();
      for (int n = 0; n < N; n++) for (int a = 0; a < A; a++) for (int w = 0; w < KW; w++) RW(U[((size_t)n * A + a) * KW + w], blobs[n].known[a][w]);
      if (scatter) for (int n = 0; n < N; n++) for (int sl = 0; sl < T; sl++) blobs[n].t_flags[sl] &= ~TF_KNOWN_ALL;  // caller-written masks: sense again
      break;
    case MUAVTA_F_THREAT_POS:
      if (!chk((size_t)N * H * 2 * 8)) BAD();
      for (int n = 0; n < N; n++) for (int h = 0; h < H; h++) { RW(D[((size_t)n * H + h) * 2], blobs[n].h_px[h]); RW(D[((size_t)n * H + h) * 2 + 1], blobs[n].h_py[h]); }
      break;
    case MUAVTA_F_THREAT_META:
      if (!chk((size_t)N * H * 8 * 4)) BAD();
      if (scatter) { e->err = "THREAT_META is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int h = 0; h < H; h++) {
        int32_t* r = I + ((size_t)n * H + h) * 8;
        St& b = blobs[n];
        r[0] = b.h_status[h]; r[1] = b.h_target[h]; r[2] = b.h_mission[h]; r[3] = b.h_acap[h]; r[4] = b.h_task_id[h]; r[5] = b.h_type[h];
        r[6] = b.h_group[h]; r[7] = b.h_intercept[h];
      }
      break;
    case MUAVTA_F_SCALARS:
      if (!chk((size_t)N * MUAVTA_N_SCALARS * 8)) BAD();
      if (scatter) { e->err = "SCALARS is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) {
        St& b = blobs[n];
        double* s = D + (size_t)n * MUAVTA_N_SCALARS;
        s[0] = b.time_steps; s[1] = b.last_reward; s[2] = b.F_Reward; s[3] = b.total_distance; s[4] = b.n_on_time;
        s[5] = b.n_missed_windows; s[6] = b.n_windowed_tasks; s[7] = b.n_task_switches; s[8] = b.n_reallocations;
        s[9] = b.n_arrivals; s[10] = b.idle_reserve_steps; s[11] = b.conclusion_time; s[12] = b.escort_requests;
        s[13] = b.escort_completed; s[14] = b.escort_failed; s[15] = b.escort_required_steps; s[16] = b.escort_covered_steps;
        s[17] = b.protection_breaches; s[18] = b.threats_intercepted; s[19] = b.recon_losses; s[20] = b.escort_losses;
        s[21] = b.mutual_support_engagements; s[22] = b.protected_rec_completed; s[23] = b.n_replans;
        s[24] = b.pending_reset; s[25] = b.n_reached; s[26] = b.n_pending; s[27] = b.next_task_id - 1;
      }
      break;
    case MUAVTA_F_OPEN_IDS:
      if (!chk((size_t)N * T * 4)) BAD();
      if (scatter) { e->err = "OPEN_IDS is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int k = 0; k < T; k++) I[(size_t)n * T + k] = k < blobs[n].n_open ? blobs[n].t_id[blobs[n].open_slot[k]] : -1;
      break;
    case MUAVTA_F_EVENTS:
      if (!chk((size_t)N * E * 2 * 4)) BAD();
      if (scatter) { e->err = "EVENTS is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int k = 0; k < E; k++) {
        I[((size_t)n * E + k) * 2] = k < blobs[n].n_dev ? blobs[n].dev_tag[k] : -1;
        I[((size_t)n * E + k) * 2 + 1] = k < blobs[n].n_dev ? blobs[n].dev_arg[k] : 0;
      }
      break;
    case MUAVTA_F_EVENT_LIST:
      if (!chk((size_t)N * E * 2 * 4)) BAD();
      if (scatter) { e->err = "EVENT_LIST is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int k = 0; k < E; k++) {
        I[((size_t)n * E + k) * 2] = k < blobs[n].n_events ? blobs[n].ev_tag[k] : -1;
        I[((size_t)n * E + k) * 2 + 1] = k < blobs[n].n_events ? blobs[n].ev_arg[k] : 0;
      }
      break;
    case MUAVTA_F_STAGED_ACTIONS:
      if (!chk((size_t)N * TL::A * 3 * 4)) BAD();
      if (scatter) { e->err = "STAGED_ACTIONS is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int k = 0; k < TL::A; k++) {
        St& b = blobs[n];
        int32_t* r = I + ((size_t)n * TL::A + k) * 3;
        bool v = k < b.n_act;
        r[0] = v ? b.act_agent[k] : -1; r[1] = v && b.act_slot[k] >= 0 ? b.t_id[b.act_slot[k]] : -1; r[2] = v ? b.act_index[k] : -1;
      }
      break;
    case MUAVTA_F_ERROR:
      if (!chk((size_t)N * 4)) BAD();
      if (scatter) { e->err = "ERROR is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) I[n] = blobs[n].error;
      break;
    case MUAVTA_F_ESCORTS:
      if (!chk((size_t)N * TL::A * 2 * 4)) BAD();
      if (scatter) { e->err = "ESCORTS is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++) for (int k = 0; k < TL::A; k++) {
        const bool v = k < blobs[n].n_escorts;
        I[((size_t)n * TL::A + k) * 2] = v ? blobs[n].esc_agent[k] : -1;
        I[((size_t)n * TL::A + k) * 2 + 1] = v ? blobs[n].esc_id[k] : -1;
      }
      break;
    case MUAVTA_F_KNOWN_COUNT:
      if (!chk((size_t)N * A * 4)) BAD();
      if (scatter) { e->err = "KNOWN_COUNT is read-only"; return MUAVTA_E_ARG; }
      for (int n = 0; n < N; n++)
        for (int a = 0; a < A; a++) {
          int c = blobs[n].a_gone[a];
          for (int w = 0; w < TL::KW; w++) c += __builtin_popcount(blobs[n].known[a][w]);
          I[(size_t)n * A + a] = c;
        }
      break;
    default:
      e->err = "unknown field";
      return MUAVTA_E_ARG;
  }
#undef BAD
#undef RW
  return MUAVTA_OK;
}

template <class TL>
void forget_obs_rows(MuavtaEnv* e) {  // the host rewrote the blobs: what the observation buffer holds no longer follows from them
  EnvState<TL>* blobs = reinterpret_cast<EnvState<TL>*>(e->host_blobs.data());
  for (int n = 0; n < e->n_envs; n++) blobs[n].obs_rows = -1;
}

template <class TL>
int check_errors(MuavtaEnv* e) {  // scan the per-env error words after a synchronising call
  typedef EnvState<TL> St;
  St* blobs = reinterpret_cast<St*>(e->host_blobs.data());
  for (int n = 0; n < e->n_envs; n++)
    if (blobs[n].error) {
      e->err = "env " + std::to_string(n) + " overflowed a tile (code " + std::to_string(blobs[n].error) +
               ": 1=task slots 2=agent queue 3=events 4=pending reveals 5=random_position 6=escorts 7=lsap)";
      return MUAVTA_E_CAPACITY;
    }
  return MUAVTA_OK;
}

}  // namespace

// ---- state lanes ----------------------------------------------------------------------------------------------------------------------
static void flip_lanes(MuavtaEnv* e) {  // the other lane's contents move into *e (and this one's into the twin)
  MuavtaEnv* t = e->hl.twin;
  std::swap(*e, *t);
  std::swap(e->hl, t->hl);  // (handle-level state and the RCCL communicator stay with the handle the caller holds)
  std::swap(e->comm, t->comm); std::swap(e->comm_rank, t->comm_rank); std::swap(e->comm_ranks, t->comm_ranks); std::swap(e->d_comm, t->d_comm);
}
static MuavtaEnv* lane_by_id(MuavtaEnv* e, int id) { return e->lane_id == id ? e : e->hl.twin; }
static int ensure_twin(MuavtaEnv* e) {  // create the second lane (same configuration, allocator, sub-batches, release log)
  if (e->hl.twin) return MUAVTA_OK;
  if (e->hl.twin_failed) return MUAVTA_E_HIP;
  MuavtaEnv* t = nullptr;
  int rc = muavta_create(&e->params, e->n_envs, e->device, &t);
  if (rc == MUAVTA_OK && e->n_parts) rc = muavta_set_parts(t, e->n_parts);
  if (rc == MUAVTA_OK && e->d_rel) rc = muavta_set_release_log(t, 1);
  if (rc != MUAVTA_OK) { if (t) muavta_destroy(t); e->hl.twin_failed = true; return rc; }
  t->alloc_mode = e->alloc_mode;
  if (e->P.slot_cap && muavta_set_slot_cap(t, e->P.slot_cap) != MUAVTA_OK) { muavta_destroy(t); e->hl.twin_failed = true; return MUAVTA_E_HIP; }
  t->lane_id = e->lane_id ^ 1;
  t->hl.lanes_mode = 1;  // (a twin never grows a twin)
  e->hl.twin = t;
  return MUAVTA_OK;
}

// ---- sub-batches on their own streams (muavta_set_parts) ------------------------------------------------------------------
// Ordering between the handle's main stream and the part streams: an entry point that works on the main stream first makes it
// wait for whatever the part streams still hold (join_parts) and flags every part to wait for the main stream before its next
// launch (fork_part).  Both are event waits on the device: the host never blocks.
static int join_parts(MuavtaEnv* e) {
  for (int p = 0; p < e->n_parts; p++) {
    if (e->part_busy[p]) {
      HIPCHK(e, hipEventRecord(e->part_ev[p], e->part_stream[p]));
      HIPCHK(e, hipStreamWaitEvent(e->stream, e->part_ev[p], 0));
      e->part_busy[p] = false;
    }
    e->part_fork_needed[p] = true;
  }
  return MUAVTA_OK;
}
#define MAIN_OP(e) do { if ((e)->n_parts) { int rc_ = join_parts(e); if (rc_) return rc_; } } while (0)
static int fork_part(MuavtaEnv* e, int p) {
  if (e->part_fork_needed[p]) {
    HIPCHK(e, hipEventRecord(e->ev_fork, e->stream));
    HIPCHK(e, hipStreamWaitEvent(e->part_stream[p], e->ev_fork, 0));
    e->part_fork_needed[p] = false;
  }
  e->part_busy[p] = true;
  return MUAVTA_OK;
}
static void part_range(const MuavtaEnv* e, int p, int* first, int* count) {
  const long long N = e->n_envs, k = e->n_parts > 0 ? e->n_parts : 1;
  const int lo = (int)(N * p / k), hi = (int)(N * (p + 1) / k);
  *first = lo; *count = hi - lo;
}
static int check_part(MuavtaEnv* e, int p, const char* who) {
  if (!e) return MUAVTA_E_ARG;
  if (e->n_parts < 1 || p < 0 || p >= e->n_parts) { e->err = std::string(who) + ": no such part (muavta_set_parts first)"; return MUAVTA_E_ARG; }
  if (!e->did_reset) { e->err = std::string(who) + " before reset"; return MUAVTA_E_STATE; }
  return MUAVTA_OK;
}

// ====================================================================================================
// C ABI
// ====================================================================================================
extern "C" {

// sizeof() of the ABI structs, so a binding can verify its own layout: out[0] = MuavtaParams, out[1] = MuavtaDims
int muavta_abi_sizes(int32_t* out) {
  if (!out) return MUAVTA_E_ARG;
  out[0] = (int32_t)sizeof(MuavtaParams);
  out[1] = (int32_t)sizeof(MuavtaDims);
  out[2] = MUAVTA_ABI_VERSION;
  return MUAVTA_OK;
}

const char* muavta_last_error(const MuavtaEnv* env) { return env ? env->err.c_str() : g_create_error.c_str(); }

int muavta_create(const MuavtaParams* params, int32_t n_envs, int32_t device, MuavtaEnv** out) {
  if (!params || !out || n_envs < 1) { g_create_error = "muavta_create: bad arguments"; return MUAVTA_E_ARG; }
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    g_create_error = "muavta_create: no HIP device visible; this library is the MI355X path and has no CPU fallback";
    return MUAVTA_E_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) { g_create_error = "muavta_create: device index out of range"; return MUAVTA_E_ARG; }
  MuavtaEnv* e = new (std::nothrow) MuavtaEnv();
  if (!e) { g_create_error = "out of memory"; return MUAVTA_E_ARG; }
  int rc = fill_dev_params(params, &e->P, &g_create_error);
  if (rc) { delete e; return rc; }
  e->params = *params;
  e->n_envs = n_envs;
  e->device = device;
  int ta = params->tile_agents > e->P.n_agents ? params->tile_agents : e->P.n_agents;
  int tt = params->tile_tasks > 0 ? params->tile_tasks : 0;
  int th = params->tile_threats > e->P.n_threats ? params->tile_threats : e->P.n_threats;
  if (ta <= Tile16::A && tt <= Tile16::T && th <= Tile16::H) e->tile = TK16;
  else if (ta <= Tile24::A && tt <= Tile24::T && th <= Tile24::H) e->tile = TK24;
  else if (ta <= Tile64::A && tt <= Tile64::T && th <= Tile64::H) e->tile = TK64;
  else { g_create_error = "muavta_create: requested tile exceeds 64 agents x 128 task slots x 48 threats"; delete e; return MUAVTA_E_ARG; }
  size_t scratch_bytes = 0;
  e->P.slot_cap = 0;  // (live slots an env may use: the tile's; muavta_set_slot_cap lowers it for capacity tests)
  DISPATCH(e, { e->A = TL::A; e->T = TL::T; e->H = TL::H; e->E = TL::E; e->R = TL::R; e->Q = TL::Q; e->state_bytes = sizeof(EnvState<TL>);
                e->cold_bytes = sizeof(EnvCold<TL>); scratch_bytes = sizeof(Scratch<TL>); });
  (void)scratch_bytes;
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); muavta_destroy(e); return MUAVTA_E_HIP; } } while (0)
  DeviceScope scope_(device);
  CK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  for (int i = 0; i < MuavtaEnv::EV_RING; i++) { CK(hipEventCreate(&e->ev0[i])); CK(hipEventCreate(&e->ev1[i])); }
  CK(hipStreamCreateWithFlags(&e->seed_stream, hipStreamNonBlocking));
  for (int b = 0; b < 2; b++) {
    CK(hipEventCreate(&e->ev_seed0[b])); CK(hipEventCreate(&e->ev_seeded[b])); CK(hipEventCreateWithFlags(&e->ev_consumed[b], hipEventDisableTiming));
  }
  // The sub-batch streams (muavta_set_parts) are created HERE, right behind the main and the seeding stream, and touched once:
  // HIP binds a stream to one of its few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) when the stream first gets work,
  // taking the least-loaded queue, and two streams on one queue execute in order.  Created lazily in the middle of a process'
  // life (after the framework's own streams, copy engines ...) two part streams could land on ONE queue: measured r3, two
  // sub-batches ran at 46 M env-steps/s inside bench.py against 70 M in a fresh process, with identical kernels.
  // (r4) Opt-in: MUAVTA_EAGER_PART_STREAMS=n (0..8, default 0) creates n of them here; the rest are created by muavta_set_parts when
  // they are first asked for.  A handle that never uses sub-batches owns two streams, not ten.
  {
    const char* ev = getenv("MUAVTA_EAGER_PART_STREAMS");
    int eager = ev ? atoi(ev) : 0;
    eager = eager < 0 ? 0 : eager > MuavtaEnv::MAX_PARTS ? MuavtaEnv::MAX_PARTS : eager;
    for (int p = 0; p < eager; p++) {
      CK(hipStreamCreateWithFlags(&e->part_stream[p], hipStreamNonBlocking));
      CK(hipEventCreateWithFlags(&e->part_ev[p], hipEventDisableTiming));
      CK(hipEventRecord(e->part_ev[p], e->part_stream[p]));
    }
  }
  CK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  const size_t N = (size_t)n_envs, mt = (size_t)e->P.max_tasks, nA = (size_t)e->P.n_agents;
  CK(hipMalloc(&e->blobs, N * e->state_bytes));
  CK(hipMemsetAsync(e->blobs, 0, N * e->state_bytes, e->stream));
  CK(hipMalloc(&e->cold, N * e->cold_bytes));
  CK(hipMemsetAsync(e->cold, 0, N * e->cold_bytes, e->stream));
  CK(hipMalloc(&e->tapes, N * MUAVTA_RNG_STREAMS * MUAVTA_RNG_WORDS * sizeof(uint32_t)));
  for (int b = 0; b < 2; b++) {
    CK(hipMalloc(&e->d_seeds[b], N * sizeof(uint64_t)));
    CK(hipHostMalloc((void**)&e->h_seeds[b], N * sizeof(uint64_t), hipHostMallocDefault));
  }
  CK(hipMalloc(&e->d_act_agent, N * e->A * sizeof(int32_t)));
  CK(hipMalloc(&e->d_act_index, N * e->A * sizeof(int32_t)));
  CK(hipMalloc(&e->d_metrics, N * MUAVTA_N_METRICS * sizeof(double)));
  CK(hipMalloc(&e->O.tasks, N * mt * 21 * sizeof(float)));
  CK(hipMalloc(&e->O.legal, N * nA * ((mt + 63) / 64) * sizeof(unsigned long long)));
  CK(hipMalloc(&e->O.pad, N * mt));
  CK(hipMalloc(&e->O.agents, N * nA * 9 * sizeof(float)));
  CK(hipMalloc(&e->O.flags, N * 5 * sizeof(float)));
  CK(hipMalloc(&e->O.reward, N * sizeof(double)));
  CK(hipMalloc(&e->O.done, N));
  {
    DevCtx h;
    memset(&h, 0, sizeof(h));
    CK(hipMalloc((void**)&e->d_pace, (size_t)PACE_KEYS * 16 * sizeof(uint32_t)));
    CK(hipMemsetAsync(e->d_pace, 0, (size_t)PACE_KEYS * 16 * sizeof(uint32_t), e->stream));  // epoch 0 is never issued
    h.P = e->P; h.O = e->O; h.tapes = e->tapes; h.blobs = e->blobs; h.cold = e->cold; h.pace = e->d_pace;
    CK(hipMalloc(&e->d_rec, 2 * MuavtaEnv::REC_SLOT));
    CK(hipMemsetAsync(e->d_rec, 0, 2 * MuavtaEnv::REC_SLOT, e->stream));
    CK(hipMalloc((void**)&e->d_ctx, sizeof(DevCtx)));
    CK(hipMemcpyAsync(e->d_ctx, &h, sizeof(DevCtx), hipMemcpyHostToDevice, e->stream));
    CK(hipStreamSynchronize(e->stream));  // `h` is a stack object
  }
#undef CK
  int arc = MUAVTA_OK;
  DISPATCH(e, arc = launch_attr<TL>(e));
  if (arc) { g_create_error = e->err; muavta_destroy(e); return arc; }
  *out = e;
  return MUAVTA_OK;
}

int muavta_destroy(MuavtaEnv* e) {
  if (!e) return MUAVTA_OK;
  muavta_comm_destroy(e);
  if (e->hl.twin) { muavta_destroy(e->hl.twin); e->hl.twin = nullptr; }
  DeviceScope scope_(e->device);
  if (e->seed_stream) hipStreamSynchronize(e->seed_stream);
  for (int p = 0; p < MuavtaEnv::MAX_PARTS; p++) {
    if (e->part_stream[p]) { hipStreamSynchronize(e->part_stream[p]); hipStreamDestroy(e->part_stream[p]); }
    if (e->part_ev[p]) hipEventDestroy(e->part_ev[p]);
  }
  if (e->ev_fork) hipEventDestroy(e->ev_fork);
  hipFree(e->d_part_agent); hipFree(e->d_part_index); if (e->d_run) hipFree(e->d_run);
  if (e->stream) hipStreamSynchronize(e->stream);
  if (e->d_seedtmp) hipFree(e->d_seedtmp); hipFree(e->blobs); hipFree(e->cold); hipFree(e->tapes); hipFree(e->d_ctx); hipFree(e->d_pace); if (e->d_rec) hipFree(e->d_rec); for (int b = 0; b < 2; b++) { hipFree(e->d_seeds[b]); if (e->d_seedbuf[b]) hipFree(e->d_seedbuf[b]); if (e->h_seeds[b]) hipHostFree(e->h_seeds[b]); } hipFree(e->d_act_agent); hipFree(e->d_act_index); if (e->d_list_agent) hipFree(e->d_list_agent); if (e->d_list_index) hipFree(e->d_list_index); hipFree(e->d_call_out); hipFree(e->d_metrics); if (e->d_tok) hipFree(e->d_tok); if (e->d_rel) hipFree(e->d_rel);
  hipFree(e->O.tasks); hipFree(e->O.legal); hipFree(e->O.pad); hipFree(e->O.agents); hipFree(e->O.flags); hipFree(e->O.reward); hipFree(e->O.done);
  for (int i = 0; i < MuavtaEnv::EV_RING; i++) { if (e->ev0[i]) hipEventDestroy(e->ev0[i]); if (e->ev1[i]) hipEventDestroy(e->ev1[i]); }
  for (int b = 0; b < 2; b++) {
    if (e->ev_seed0[b]) hipEventDestroy(e->ev_seed0[b]);
    if (e->ev_seeded[b]) hipEventDestroy(e->ev_seeded[b]);
    if (e->ev_consumed[b]) hipEventDestroy(e->ev_consumed[b]);
  }
  if (e->seed_stream) hipStreamDestroy(e->seed_stream);
  if (e->stream) hipStreamDestroy(e->stream);
  delete e;
  return MUAVTA_OK;
}

int muavta_dims(const MuavtaEnv* e, MuavtaDims* d) {
  if (!e || !d) return MUAVTA_E_ARG;
  d->n_envs = e->n_envs; d->n_agents = e->P.n_agents; d->tile_agents = e->A; d->tile_tasks = e->T; d->tile_threats = e->H;
  d->max_tasks = e->P.max_tasks; d->obs_task_width = 21; d->obs_agent_width = 9; d->queue_cap = e->Q; d->event_cap = e->E;
  d->action_cap = e->A; d->state_bytes = (int64_t)(e->state_bytes + e->cold_bytes);
  d->n_threats = e->P.n_threats; d->known_words = (e->T + 31) / 32; d->lds_bytes = (int32_t)e->lds_bytes; d->legal_words = (e->P.max_tasks + 63) / 64;
  return MUAVTA_OK;
}

// Upload `seeds` and run the seeding kernel on the seed stream into the next slot; the handle's stream waits for it.
// The caller launches the consumer on e->stream and then calls seeding_consumed(e, slot).
static int enqueue_seeding(MuavtaEnv* e, const uint64_t* seeds, const uint64_t** ds, const uint32_t** sb, int* slot) {
  const size_t N = (size_t)e->n_envs;
  const int b = (int)(e->seed_seq & 1u);  // (the slot sequence only advances once the kernel is queued: a failed call leaves it alone)
  if (e->seed_used[b]) {
    HIPCHK(e, hipEventSynchronize(e->ev_seeded[b]));                        // the staging copy of two calls ago has left h_seeds[b]
    HIPCHK(e, hipStreamWaitEvent(e->seed_stream, e->ev_consumed[b], 0));    // ... and its consumer has read d_seeds / d_seedbuf[b]
  }
  memcpy(e->h_seeds[b], seeds, N * sizeof(uint64_t));
  HIPCHK(e, hipMemcpyAsync(e->d_seeds[b], e->h_seeds[b], N * sizeof(uint64_t), hipMemcpyHostToDevice, e->seed_stream));
  const size_t seed_bytes = ((N + 15) / 16) * WG * 624 * sizeof(uint32_t);  // whole waves of 16 envs x 4 streams
  if (!e->d_seedbuf[b]) HIPCHK(e, hipMalloc((void**)&e->d_seedbuf[b], seed_bytes));
  if (!e->d_seedtmp) HIPCHK(e, hipMalloc((void**)&e->d_seedtmp, seed_bytes));  // k_seed's scratch (one: its launches are serialised on the seed stream)
  HIPCHK(e, hipEventRecord(e->ev_seed0[b], e->seed_stream));
  hipLaunchKernelGGL(k_seed, dim3((unsigned)((N + 15) / 16)), dim3(WG), 0, e->seed_stream, (const uint64_t*)e->d_seeds[b], (int)N,
                     (int)(e->P.num_obstacles > 0), e->d_seedbuf[b], e->d_seedtmp);
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipEventRecord(e->ev_seeded[b], e->seed_stream));
  HIPCHK(e, hipStreamWaitEvent(e->stream, e->ev_seeded[b], 0));
  e->seed_used[b] = true;
  e->seed_seq++;
  e->last_seed_slot = b;
  *ds = e->d_seeds[b]; *sb = e->d_seedbuf[b]; *slot = b;
  return MUAVTA_OK;
}
static int seeding_consumed(MuavtaEnv* e, int slot) {
  HIPCHK(e, hipEventRecord(e->ev_consumed[slot], e->stream));
  return MUAVTA_OK;
}

int muavta_reset(MuavtaEnv* e, const uint64_t* seeds) {
  if (!e || !seeds) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  const uint64_t* ds = nullptr;
  const uint32_t* sb = nullptr;
  int slot = 0;
  { int rc = enqueue_seeding(e, seeds, &ds, &sb, &slot); if (rc) return rc; }
  DISPATCH(e, hipLaunchKernelGGL(k_reset<TL>, dim3(e->n_envs), dim3(WG), Lds<TL>::bytes(), e->stream, (const DevCtx*)e->d_ctx, ds, sb));
  HIPCHK(e, hipGetLastError());
  { int rc = seeding_consumed(e, slot); if (rc) return rc; }
  e->last_seeded = true;  // muavta_last_seed_ms reports this reset's k_seed
  e->did_reset = true;
  e->host_valid = false;
  return MUAVTA_OK;
}

static int step_impl(MuavtaEnv* e, const int32_t* aa, const int32_t* ai, int cap = 0) {
  if (!e->did_reset) { e->err = "step before reset"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  const int32_t *da = nullptr, *di = nullptr;
  if (cap <= 0) cap = e->A;
  if (aa) {
    size_t bytes = (size_t)e->n_envs * cap * sizeof(int32_t);
    int32_t *ba = e->d_act_agent, *bi = e->d_act_index;
    if (cap > e->A) {  // rows longer than the handle's action buffers: muavta_step_lists
      if (cap > e->list_cap) {
        HIPCHK(e, hipStreamSynchronize(e->stream));
        if (e->d_list_agent) hipFree(e->d_list_agent);
        if (e->d_list_index) hipFree(e->d_list_index);
        e->d_list_agent = e->d_list_index = nullptr; e->list_cap = 0;
        HIPCHK(e, hipMalloc(&e->d_list_agent, bytes));
        HIPCHK(e, hipMalloc(&e->d_list_index, bytes));
        e->list_cap = cap;
      }
      ba = e->d_list_agent; bi = e->d_list_index;
    }
    HIPCHK(e, hipMemcpyAsync(ba, aa, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(bi, ai, bytes, hipMemcpyHostToDevice, e->stream));
    da = ba; di = bi;
  }
  if (e->d_rel) HIPCHK(e, hipMemsetAsync(e->d_rel, 0, (size_t)e->n_envs * (1 + MUAVTA_REL_ROW * e->T) * sizeof(double), e->stream));
  DISPATCH(e, hipLaunchKernelGGL(k_step<TL>, dim3(e->n_envs), dim3(WG), 0, e->stream, (const DevCtx*)e->d_ctx, da, di, cap, e->d_rel, 0));  // (static LDS)
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  return MUAVTA_OK;
}

int muavta_step_lists(MuavtaEnv* e, const int32_t* act_agent, const int32_t* act_index, int32_t list_cap) {
  if (!e || !act_agent || !act_index) return MUAVTA_E_ARG;
  if (list_cap < 1 || list_cap > 32767) { e->err = "muavta_step_lists: list_cap must be in 1..32767"; return MUAVTA_E_ARG; }
  // agent ids index the per-agent arrays of the env blob on the device: reject anything outside [0, n_agents) up front
  // (the reference's actions dict is keyed by agent name: an unknown name is a KeyError there, DroneEnv.py:813-816)
  for (int n = 0; n < e->n_envs; n++)
    for (int k = 0; k < list_cap; k++) {
      const int a = act_agent[(size_t)n * list_cap + k];
      if (a < 0) break;
      if (a >= e->P.n_agents) {
        e->err = "muavta_step: env " + std::to_string(n) + " names agent id " + std::to_string(a) + ", valid ids are 0.." + std::to_string(e->P.n_agents - 1);
        return MUAVTA_E_ARG;
      }
    }
  return step_impl(e, act_agent, act_index, list_cap);
}
int muavta_step(MuavtaEnv* e, const int32_t* act_agent, const int32_t* act_index) {
  if (!e) return MUAVTA_E_ARG;
  return muavta_step_lists(e, act_agent, act_index, e->A);
}
int muavta_step_staged(MuavtaEnv* e) {
  if (!e) return MUAVTA_E_ARG;
  return step_impl(e, nullptr, nullptr);
}

int muavta_allocate(MuavtaEnv* e, int32_t interval, int32_t use_vis, int32_t* act_agent, int32_t* act_index) {
  if (!e) return MUAVTA_E_ARG;
  if (!e->did_reset) { e->err = "allocate before reset"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  DISPATCH(e, hipLaunchKernelGGL(k_allocate<TL>, dim3(e->n_envs), dim3(WG), Lds<TL>::bytes(), e->stream, (const DevCtx*)e->d_ctx, interval, use_vis, e->alloc_mode,
                                 e->d_act_agent, e->d_act_index, e->A, 0));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  if (act_agent && act_index) {
    size_t bytes = (size_t)e->n_envs * e->A * sizeof(int32_t);
    HIPCHK(e, hipMemcpyAsync(act_agent, e->d_act_agent, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(act_index, e->d_act_index, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
  }
  return MUAVTA_OK;
}

static int token_dims(int kind, int* dt, int* da);

// ---- HungarianAllocator.allocate_tasks with the caller's edge scores / priorities / reserved agents ------------------------------
static int scored_check(MuavtaEnv* e, const MuavtaScored* sp) {
  int dt, da;
  if (!e) return MUAVTA_E_ARG;
  if (!sp || token_dims(sp->kind, &dt, &da) || sp->max_tasks < 1 || sp->max_tasks > 128 || sp->max_agents < 1 || sp->max_agents > 64 ||
      sp->gate < MUAVTA_GATE_FORCE || sp->gate > MUAVTA_GATE_ALLOCATOR || (sp->flags & ~7) ||
      (sp->kind == MUAVTA_TOK_ESCORT && (sp->flags & MUAVTA_SC_FULL_TASK_LIST))) {
    e->err = "muavta_allocate_scored: bad spec (kind 0..2, max_tasks 1..128, max_agents 1..64, gate 0..3, flags 0..7; build_escort_tokens has no untruncated list)";
    return MUAVTA_E_ARG;
  }
  if (!e->did_reset) { e->err = "allocate before reset"; return MUAVTA_E_STATE; }
  return MUAVTA_OK;
}
int muavta_allocate_scored_device(MuavtaEnv* e, const MuavtaScored* sp) {
  if (int rc = scored_check(e, sp)) return rc;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  ScoredDev sc{sp->edge_scores, sp->task_pri, (const unsigned long long*)sp->reserved, sp->selected, sp->replanned, sp->kind, sp->max_tasks,
               sp->max_agents, sp->gate, sp->flags};
  DISPATCH(e, hipLaunchKernelGGL(k_allocate_scored<TL>, dim3(e->n_envs), dim3(WG), Lds<TL>::bytes() + SCORED_EXTRA_LDS, e->stream, (const DevCtx*)e->d_ctx, sc,
                                 sp->replan_interval, sp->use_visibility, e->d_act_agent, e->d_act_index, e->A, 0));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  return MUAVTA_OK;
}
int muavta_allocate_scored(MuavtaEnv* e, const MuavtaScored* sp, int32_t* act_agent, int32_t* act_index) {
  if (int rc = scored_check(e, sp)) return rc;
  DeviceScope scope_(e->device);
  const size_t N = (size_t)e->n_envs, MT = (size_t)sp->max_tasks, MA = (size_t)sp->max_agents;
  const size_t sz[5] = {N * MA * MT * 4, N * MT * 8, N * 8, N * MA * MT * 4, N * 4};  // scores, pri, reserved | selected, replanned
  size_t off[6] = {0};
  for (int i = 0; i < 5; i++) off[i + 1] = off[i] + ((sz[i] + 255) & ~(size_t)255);
  if (off[5] > e->tok_bytes) {  // (shares the staging buffer of muavta_tokens' host variant)
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->d_tok) hipFree(e->d_tok);
    e->d_tok = nullptr; e->tok_bytes = 0;
    HIPCHK(e, hipMalloc(&e->d_tok, off[5]));
    e->tok_bytes = off[5];
  }
  char* b = (char*)e->d_tok;
  const void* in[3] = {sp->edge_scores, sp->task_pri, sp->reserved};
  for (int i = 0; i < 3; i++)
    if (in[i]) HIPCHK(e, hipMemcpyAsync(b + off[i], in[i], sz[i], hipMemcpyHostToDevice, e->stream));
  MuavtaScored d = *sp;
  d.edge_scores = sp->edge_scores ? (const float*)(b + off[0]) : nullptr;
  d.task_pri = sp->task_pri ? (const double*)(b + off[1]) : nullptr;
  d.reserved = sp->reserved ? (const uint64_t*)(b + off[2]) : nullptr;
  d.selected = sp->selected ? (float*)(b + off[3]) : nullptr;
  d.replanned = sp->replanned ? (int32_t*)(b + off[4]) : nullptr;
  if (int rc = muavta_allocate_scored_device(e, &d)) return rc;
  if (sp->selected) HIPCHK(e, hipMemcpyAsync(sp->selected, b + off[3], sz[3], hipMemcpyDeviceToHost, e->stream));
  if (sp->replanned) HIPCHK(e, hipMemcpyAsync(sp->replanned, b + off[4], sz[4], hipMemcpyDeviceToHost, e->stream));
  if (act_agent && act_index) {
    size_t bytes = N * e->A * sizeof(int32_t);
    HIPCHK(e, hipMemcpyAsync(act_agent, e->d_act_agent, bytes, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipMemcpyAsync(act_index, e->d_act_index, bytes, hipMemcpyDeviceToHost, e->stream));
  }
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}
int muavta_rl_step_device(MuavtaEnv* e, const MuavtaRlStep* rs) {
  if (!e || !rs) return MUAVTA_E_ARG;
  const MuavtaScored* sp = &rs->plan;
  if (int rc = scored_check(e, sp)) return rc;
  const bool tok = rs->task_feats != nullptr;
  if (tok && (!rs->task_mask || !rs->task_ids || !rs->agent_feats || !rs->agent_mask || !rs->agent_ids || !rs->edge_valid)) {
    e->err = "muavta_rl_step_device: the next-token outputs come all together or not at all (n_urgent alone is optional)"; return MUAVTA_E_ARG;
  }
  DeviceScope scope_(e->device);
  if (e->d_rel) { e->err = "muavta_rl_step_device: the release log must be off (muavta_set_release_log)"; return MUAVTA_E_STATE; }
  ScoredDev sc{sp->edge_scores, sp->task_pri, (const unsigned long long*)sp->reserved, sp->selected, sp->replanned, sp->kind, sp->max_tasks,
               sp->max_agents, sp->gate, sp->flags};
  hipStream_t stream = e->stream;
  int first = 0, count = e->n_envs;
  if (rs->part > 0) {  // one sub-batch on its own stream (muavta_set_parts): the tensors are the whole batch's, the launch touches the part's rows
    const int part = rs->part - 1;
    { int rc = check_part(e, part, "muavta_rl_step_device"); if (rc) return rc; }
    { int rc = fork_part(e, part); if (rc) return rc; }
    part_range(e, part, &first, &count);
    stream = e->part_stream[part];
  } else {
    MAIN_OP(e);
  }
  DISPATCH(e, launch_rl_step<TL>(e, sc, rs, stream, first, count));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  return MUAVTA_OK;
}
// Run to the next replan gate (k_run; include/muavta.h): the policy in the loop, consulted only where an env's gate fired
int muavta_rl_run_device(MuavtaEnv* e, const MuavtaRlRun* rr) {
  if (!e || !rr) return MUAVTA_E_ARG;
  const MuavtaRlStep* rs = &rr->first;
  const MuavtaScored* sp = &rs->plan;
  if (int rc = scored_check(e, sp)) return rc;
  const bool tok = rs->task_feats != nullptr, ptok = rr->park_task_feats != nullptr;
  if ((tok && (!rs->task_mask || !rs->task_ids || !rs->agent_feats || !rs->agent_mask || !rs->agent_ids || !rs->edge_valid)) ||
      (ptok && (!rr->park_task_mask || !rr->park_task_ids || !rr->park_agent_feats || !rr->park_agent_mask || !rr->park_agent_ids || !rr->park_edge_valid))) {
    e->err = "muavta_rl_run_device: the token outputs (next / park) come all together or not at all (n_urgent alone is optional)"; return MUAVTA_E_ARG;
  }
  if (rr->max_steps < 0) { e->err = "muavta_rl_run_device: max_steps >= 0 (0: until the gate fires or the episode ends)"; return MUAVTA_E_ARG; }
  DeviceScope scope_(e->device);
  if (e->d_rel) { e->err = "muavta_rl_run_device: the release log must be off (muavta_set_release_log)"; return MUAVTA_E_STATE; }
  ScoredDev sc{sp->edge_scores, sp->task_pri, (const unsigned long long*)sp->reserved, sp->selected, sp->replanned, sp->kind, sp->max_tasks,
               sp->max_agents, sp->gate, sp->flags};
  hipStream_t stream = e->stream;
  int first = 0, count = e->n_envs;
  if (rs->part > 0) {
    const int part = rs->part - 1;
    { int rc = check_part(e, part, "muavta_rl_run_device"); if (rc) return rc; }
    { int rc = fork_part(e, part); if (rc) return rc; }
    part_range(e, part, &first, &count);
    stream = e->part_stream[part];
  } else {
    MAIN_OP(e);
  }
  RunOut R{rs->s_wps, rs->done, rr->n_stepped, rr->park, rr->reward_sum};
  DISPATCH(e, launch_run<TL>(e, RUN_SRC_SCORED, sc, rs, rr, R, sp->gate, sp->replan_interval, sp->use_visibility, rs->write_obs, rr->max_steps, nullptr, nullptr, 0, stream, first, count));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  return MUAVTA_OK;
}
int muavta_step_run(MuavtaEnv* e, const int32_t* act_agent, const int32_t* act_index, int32_t gate, int32_t interval, int32_t max_steps, int32_t write_obs,
                    int32_t* n_stepped, uint8_t* park, double* reward_sum) {
  if (!e) return MUAVTA_E_ARG;
  if (!e->did_reset) { e->err = "muavta_step_run before reset"; return MUAVTA_E_STATE; }
  if ((act_agent == nullptr) != (act_index == nullptr) || gate < MUAVTA_GATE_FORCE || gate > MUAVTA_GATE_ALLOCATOR || max_steps < 0) {
    e->err = "muavta_step_run: action rows come as a pair (or both NULL: the staged plan), gate 0..3, max_steps >= 0"; return MUAVTA_E_ARG;
  }
  if (act_agent)
    for (int n = 0; n < e->n_envs; n++)
      for (int k = 0; k < e->A; k++) {
        const int a = act_agent[(size_t)n * e->A + k];
        if (a < 0) break;
        if (a >= e->P.n_agents) { e->err = "muavta_step_run: env " + std::to_string(n) + " names agent id " + std::to_string(a); return MUAVTA_E_ARG; }
      }
  DeviceScope scope_(e->device);
  if (e->d_rel) { e->err = "muavta_step_run: the release log must be off (muavta_set_release_log)"; return MUAVTA_E_STATE; }
  MAIN_OP(e);
  const size_t N = (size_t)e->n_envs;
  if (!e->d_run) HIPCHK(e, hipMalloc(&e->d_run, N * 16));
  double* d_rsum = (double*)e->d_run; int32_t* d_n = (int32_t*)(d_rsum + N); uint8_t* d_park = (uint8_t*)(d_n + N);
  const int32_t *da = nullptr, *di = nullptr;
  if (act_agent) {
    const size_t bytes = N * e->A * sizeof(int32_t);
    HIPCHK(e, hipMemcpyAsync(e->d_act_agent, act_agent, bytes, hipMemcpyHostToDevice, e->stream));
    HIPCHK(e, hipMemcpyAsync(e->d_act_index, act_index, bytes, hipMemcpyHostToDevice, e->stream));
    da = e->d_act_agent; di = e->d_act_index;
  }
  ScoredDev sc{};
  RunOut R{nullptr, nullptr, d_n, d_park, d_rsum};
  DISPATCH(e, launch_run<TL>(e, act_agent ? RUN_SRC_ROWS : RUN_SRC_STAGED, sc, nullptr, nullptr, R, gate, interval, 0, write_obs, max_steps, da, di, e->A, e->stream, 0, e->n_envs));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  if (n_stepped) HIPCHK(e, hipMemcpyAsync(n_stepped, d_n, N * 4, hipMemcpyDeviceToHost, e->stream));
  if (park) HIPCHK(e, hipMemcpyAsync(park, d_park, N, hipMemcpyDeviceToHost, e->stream));
  if (reward_sum) HIPCHK(e, hipMemcpyAsync(reward_sum, d_rsum, N * 8, hipMemcpyDeviceToHost, e->stream));
  if (n_stepped || park || reward_sum) HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}
}  // extern "C" (the launcher below is a template)
template <class TL>
static void launch_rollout(MuavtaEnv* e, const uint64_t* ds, int n_steps, int interval, int use_vis, int write_obs, const uint32_t* sb, size_t extra_lds,
                           const MuavtaRecord* rec, hipStream_t stream, int env_base, int n_launch) {
  RecordPtrs<TL> R;
  memset(&R, 0, sizeof(R));
  const int epoch = (int)(e->pace_epoch++ % 65535u) + 1;  // 1..65535: the zero-filled table matches no launch
  if (rec) {
    if (rec->kind >= 0) {
      typename Sim<TL>::TokPtrs K{rec->task_feats, rec->task_mask, rec->task_ids, rec->agent_feats, rec->agent_mask, rec->agent_ids, rec->edge_valid,
                                  rec->n_urgent, rec->expert_mask, rec->replanned, rec->kind, rec->max_tasks, rec->max_agents};
      R.K = K; R.s_wps = rec->s_wps;
    }
    if (rec->obs_tasks) {
      R.O.tasks = rec->obs_tasks; R.O.legal = (unsigned long long*)rec->obs_legal; R.O.pad = rec->obs_pad; R.O.agents = rec->obs_agents;
      R.O.flags = rec->obs_flags; R.O.reward = rec->obs_reward; R.O.done = rec->obs_done;
    }
    R.n_envs = e->n_envs;
    static_assert(sizeof(RecordPtrs<TL>) <= MuavtaEnv::REC_SLOT, "record-pointer slot too small");
    // The slot is filled by a one-lane kernel that takes the struct BY VALUE (kernel arguments are captured when the launch is queued)
    // — stream-ordered behind the previous launch that read the slot.  NOT hipMemcpyAsync from this stack frame: for pageable memory the
    // runtime may pin the pages and copy after the call has returned, by when the frame is gone (first r4 build: wild ring pointers).
    RecBlob blob;
    memset(&blob, 0, sizeof(blob));
    memcpy(&blob, &R, sizeof(R));
    hipLaunchKernelGGL(k_store_rec, dim3(1), dim3(64), 0, stream, blob, (uint32_t*)((char*)e->d_rec + MuavtaEnv::REC_SLOT));
    hipLaunchKernelGGL((k_rollout<TL, true>), dim3(n_launch), dim3(WG), extra_lds, stream, (const DevCtx*)e->d_ctx, ds,
                       n_steps, interval, use_vis, e->alloc_mode, write_obs, e->d_metrics, sb, (const RecordPtrs<TL>*)((char*)e->d_rec + MuavtaEnv::REC_SLOT), epoch, env_base);
  } else {
    hipLaunchKernelGGL((k_rollout<TL, false>), dim3(n_launch), dim3(WG), extra_lds, stream, (const DevCtx*)e->d_ctx, ds,
                       n_steps, interval, use_vis, e->alloc_mode, write_obs, e->d_metrics, sb, (const RecordPtrs<TL>*)e->d_rec, epoch, env_base);
  }
}
extern "C" {
static int rollout_impl(MuavtaEnv* e, const uint64_t* seeds, int32_t n_steps, int32_t interval, int32_t use_vis, int32_t write_obs, const MuavtaRecord* rec) {
  if (!e || n_steps < 0) return MUAVTA_E_ARG;
  if (!seeds && !e->did_reset) { e->err = "rollout without seeds before reset"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  if (seeds && e->hl.lanes_mode != 1) {
    // a fresh episode batch while this lane's last rollout is still running (or always, in mode 2): it goes to the other lane
    bool want = e->hl.lanes_mode == 2;
    if (!want && e->n_rollouts) { want = hipEventQuery(e->ev1[(e->n_rollouts - 1) % MuavtaEnv::EV_RING]) == hipErrorNotReady; (void)hipGetLastError(); }
    if (want && ensure_twin(e) == MUAVTA_OK) flip_lanes(e);
  }
  MAIN_OP(e);
  const uint64_t* ds = nullptr;
  const uint32_t* sb = nullptr;
  int slot = -1;
  if (seeds) { int rc = enqueue_seeding(e, seeds, &ds, &sb, &slot); if (rc) return rc; }
  e->last_seeded = ds != nullptr;
  const int evi = (int)(e->n_rollouts % MuavtaEnv::EV_RING);
  HIPCHK(e, hipEventRecord(e->ev0[evi], e->stream));
  static const size_t extra_lds = getenv("MUAVTA_EXTRA_LDS") ? (size_t)atoi(getenv("MUAVTA_EXTRA_LDS")) : 0;  // occupancy experiments only
  DISPATCH(e, launch_rollout<TL>(e, ds, n_steps, interval, use_vis, write_obs, sb, extra_lds, rec, e->stream, 0, e->n_envs));
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipEventRecord(e->ev1[evi], e->stream));
  e->hl.ring_lane[e->hl.n_launches % MuavtaEnv::HandleLevel::RING] = (unsigned char)e->lane_id;
  e->hl.ring_no[e->hl.n_launches % MuavtaEnv::HandleLevel::RING] = e->n_rollouts;
  e->hl.n_launches++;
  e->n_rollouts++;
  e->timing_stale = false;
  if (slot >= 0) { int rc = seeding_consumed(e, slot); if (rc) return rc; }
  e->did_reset = true;
  e->host_valid = false;
  return MUAVTA_OK;
}
int muavta_rollout(MuavtaEnv* e, const uint64_t* seeds, int32_t n_steps, int32_t interval, int32_t use_vis, int32_t write_obs) {
  return rollout_impl(e, seeds, n_steps, interval, use_vis, write_obs, nullptr);
}
int muavta_rollout_record(MuavtaEnv* e, const uint64_t* seeds, int32_t n_steps, int32_t interval, int32_t use_vis, int32_t write_obs, const MuavtaRecord* rec) {
  int dt, da;
  bool bad = !e || !rec;
  if (!bad && rec->kind >= 0)
    bad = token_dims(rec->kind, &dt, &da) || rec->max_tasks < 1 || rec->max_agents < 1 || rec->max_tasks > 4096 || rec->max_agents > 4096 || !rec->task_feats ||
          !rec->task_mask || !rec->task_ids || !rec->agent_feats || !rec->agent_mask || !rec->agent_ids || !rec->edge_valid || !rec->s_wps;
  const bool any_obs = !bad && (rec->obs_tasks || rec->obs_legal || rec->obs_pad || rec->obs_agents || rec->obs_flags || rec->obs_reward || rec->obs_done);
  if (any_obs)  // all seven or none, and only with per-step observations switched on
    bad = !(rec->obs_tasks && rec->obs_legal && rec->obs_pad && rec->obs_agents && rec->obs_flags && rec->obs_reward && rec->obs_done) || !write_obs;
  if (!bad && rec->kind < 0 && !any_obs) bad = true;  // nothing to record
  if (bad) {
    if (e) e->err = "muavta_rollout_record: bad argument";
    return MUAVTA_E_ARG;
  }
  if (any_obs && n_steps > 0) {
    DeviceScope scope_(e->device);
    HIPCHK(e, hipMemsetAsync(rec->obs_done, MUAVTA_OBS_UNWRITTEN, (size_t)n_steps * (size_t)e->n_envs, e->stream));
  }
  return rollout_impl(e, seeds, n_steps, interval, use_vis, write_obs, rec);
}

// ---- sub-batches ---------------------------------------------------------------------------------------------------------
int muavta_set_parts(MuavtaEnv* e, int32_t n_parts) {
  if (!e || n_parts < 0 || n_parts > MuavtaEnv::MAX_PARTS || n_parts > e->n_envs) { if (e) e->err = "muavta_set_parts: 0 .. 8 parts, at most one per env"; return MUAVTA_E_ARG; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);  // whatever the old parts hold is ordered in front of the main stream
  if (n_parts == 1) n_parts = 0;
  for (int p = 0; p < n_parts; p++) {
    if (!e->part_stream[p]) {  // (not among the MUAVTA_EAGER_PART_STREAMS created by muavta_create)
      HIPCHK(e, hipStreamCreateWithFlags(&e->part_stream[p], hipStreamNonBlocking));
      HIPCHK(e, hipEventCreateWithFlags(&e->part_ev[p], hipEventDisableTiming));
      HIPCHK(e, hipEventRecord(e->part_ev[p], e->part_stream[p]));
    }
    e->part_busy[p] = false; e->part_fork_needed[p] = true;
  }
  if (n_parts && !e->d_part_agent) {
    HIPCHK(e, hipMalloc((void**)&e->d_part_agent, (size_t)e->n_envs * e->A * sizeof(int32_t)));
    HIPCHK(e, hipMalloc((void**)&e->d_part_index, (size_t)e->n_envs * e->A * sizeof(int32_t)));
  }
  e->n_parts = n_parts;
  if (e->hl.twin) return muavta_set_parts(e->hl.twin, n_parts);
  return MUAVTA_OK;
}
int muavta_part_range(const MuavtaEnv* e, int32_t part, int32_t* first, int32_t* count) {
  if (!e || !first || !count || part < 0 || part >= (e->n_parts > 0 ? e->n_parts : 1)) return MUAVTA_E_ARG;
  int f, c;
  part_range(e, part, &f, &c);
  *first = f; *count = c;
  return MUAVTA_OK;
}
int muavta_rollout_part(MuavtaEnv* e, int32_t part, int32_t n_steps, int32_t interval, int32_t use_vis, int32_t write_obs) {
  { int rc = check_part(e, part, "muavta_rollout_part"); if (rc) return rc; }
  if (n_steps < 0) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  { int rc = fork_part(e, part); if (rc) return rc; }
  int first, count;
  part_range(e, part, &first, &count);
  DISPATCH(e, launch_rollout<TL>(e, nullptr, n_steps, interval, use_vis, write_obs, nullptr, 0, nullptr, e->part_stream[part], first, count));
  HIPCHK(e, hipGetLastError());
  e->timing_stale = true;  // (part launches carry no event pair: muavta_last_kernel_ms / _history refuse until the next whole-batch rollout)
  e->host_valid = false;
  return MUAVTA_OK;
}
int muavta_step_part(MuavtaEnv* e, int32_t part, const int32_t* act_agent, const int32_t* act_index) {
  { int rc = check_part(e, part, "muavta_step_part"); if (rc) return rc; }
  if (e->d_rel) { e->err = "muavta_step_part: the release log is a whole-batch facility (muavta_set_release_log off)"; return MUAVTA_E_STATE; }
  int first, count;
  part_range(e, part, &first, &count);
  if (act_agent && act_index) {
    for (int n = 0; n < count; n++)
      for (int k = 0; k < e->A; k++) {
        const int a = act_agent[(size_t)n * e->A + k];
        if (a < 0) break;
        if (a >= e->P.n_agents) { e->err = "muavta_step_part: agent id out of range"; return MUAVTA_E_ARG; }
      }
  } else if (act_agent || act_index) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  { int rc = fork_part(e, part); if (rc) return rc; }
  hipStream_t st = e->part_stream[part];
  const int32_t *da = nullptr, *di = nullptr;
  if (act_agent) {  // rows first .. first + count of the staging pair; NULL: the actions muavta_allocate_part staged in the blob
    const size_t off = (size_t)first * e->A, bytes = (size_t)count * e->A * sizeof(int32_t);
    HIPCHK(e, hipMemcpyAsync(e->d_part_agent + off, act_agent, bytes, hipMemcpyHostToDevice, st));
    HIPCHK(e, hipMemcpyAsync(e->d_part_index + off, act_index, bytes, hipMemcpyHostToDevice, st));
    da = e->d_part_agent; di = e->d_part_index;
  }
  DISPATCH(e, hipLaunchKernelGGL(k_step<TL>, dim3(count), dim3(WG), 0, st, (const DevCtx*)e->d_ctx, da, di, e->A, (double*)nullptr, first));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  return MUAVTA_OK;
}
int muavta_allocate_part(MuavtaEnv* e, int32_t part, int32_t interval, int32_t use_vis, int32_t* act_agent, int32_t* act_index) {
  { int rc = check_part(e, part, "muavta_allocate_part"); if (rc) return rc; }
  DeviceScope scope_(e->device);
  { int rc = fork_part(e, part); if (rc) return rc; }
  int first, count;
  part_range(e, part, &first, &count);
  hipStream_t st = e->part_stream[part];
  DISPATCH(e, hipLaunchKernelGGL(k_allocate<TL>, dim3(count), dim3(WG), Lds<TL>::bytes(), st, (const DevCtx*)e->d_ctx, interval, use_vis, e->alloc_mode,
                                 e->d_part_agent, e->d_part_index, e->A, first));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;
  if (act_agent && act_index) {
    const size_t off = (size_t)first * e->A, bytes = (size_t)count * e->A * sizeof(int32_t);
    HIPCHK(e, hipMemcpyAsync(act_agent, e->d_part_agent + off, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipMemcpyAsync(act_index, e->d_part_index + off, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(e, hipStreamSynchronize(st));
  }
  return MUAVTA_OK;
}
int muavta_observe_part(MuavtaEnv* e, int32_t part, float* tasks, uint64_t* legal, uint8_t* pad, float* agents, float* flags, double* reward, uint8_t* done) {
  { int rc = check_part(e, part, "muavta_observe_part"); if (rc) return rc; }
  DeviceScope scope_(e->device);
  { int rc = fork_part(e, part); if (rc) return rc; }
  int first, count;
  part_range(e, part, &first, &count);
  hipStream_t st = e->part_stream[part];
  const size_t F = (size_t)first, C = (size_t)count, mt = (size_t)e->P.max_tasks, nA = (size_t)e->P.n_agents, kw = (mt + 63) / 64;
  if (tasks) HIPCHK(e, hipMemcpyAsync(tasks, e->O.tasks + F * mt * 21, C * mt * 21 * sizeof(float), hipMemcpyDeviceToHost, st));
  if (legal) HIPCHK(e, hipMemcpyAsync(legal, e->O.legal + F * nA * kw, C * nA * kw * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  if (pad) HIPCHK(e, hipMemcpyAsync(pad, e->O.pad + F * mt, C * mt, hipMemcpyDeviceToHost, st));
  if (agents) HIPCHK(e, hipMemcpyAsync(agents, e->O.agents + F * nA * 9, C * nA * 9 * sizeof(float), hipMemcpyDeviceToHost, st));
  if (flags) HIPCHK(e, hipMemcpyAsync(flags, e->O.flags + F * 5, C * 5 * sizeof(float), hipMemcpyDeviceToHost, st));
  if (reward) HIPCHK(e, hipMemcpyAsync(reward, e->O.reward + F, C * sizeof(double), hipMemcpyDeviceToHost, st));
  if (done) HIPCHK(e, hipMemcpyAsync(done, e->O.done + F, C, hipMemcpyDeviceToHost, st));
  HIPCHK(e, hipStreamSynchronize(st));
  return MUAVTA_OK;
}
int muavta_wait_part(MuavtaEnv* e, int32_t part) {  // part < 0: every part
  if (!e || part >= e->n_parts) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  for (int p = 0; p < e->n_parts; p++)
    if (part < 0 || p == part) HIPCHK(e, hipStreamSynchronize(e->part_stream[p]));
  return MUAVTA_OK;
}

#ifdef MUAVTA_DIAG_TIMES
int muavta_diag_times(MuavtaEnv* e, uint32_t* out, int32_t n) {  // diagnostic build only: [3][n] start, end (10 ns units), hw ids
  DeviceScope scope_(e->device);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int k = 0; k < 3; k++)
    HIPCHK(e, hipMemcpy(out + (size_t)k * n, e->d_pace + (1u << 19) + 65536u * k, (size_t)n * 4, hipMemcpyDeviceToHost));
  return MUAVTA_OK;
}
#endif
#ifdef MUAVTA_PROF
int muavta_prof_target(int env) { return hipMemcpyToSymbol(HIP_SYMBOL(g_prof_target), &env, sizeof(env)) == hipSuccess ? MUAVTA_OK : MUAVTA_E_HIP; }  // diagnostic build only
int muavta_prof_read(unsigned long long* out, int reset) {  // diagnostic build only
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), PROF_N * sizeof(unsigned long long)) != hipSuccess) return MUAVTA_E_HIP;
  if (reset) { unsigned long long z[PROF_N] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof(z)) != hipSuccess) return MUAVTA_E_HIP; }
  return MUAVTA_OK;
}
#endif

int muavta_set_allocator(MuavtaEnv* e, int32_t mode) {
  if (!e || (mode < MUAVTA_ALLOC_HUNGARIAN || mode > MUAVTA_ALLOC_HUNGARIAN_GATED)) { if (e) e->err = "unknown allocator mode"; return MUAVTA_E_ARG; }
  e->alloc_mode = mode;
  if (e->hl.twin) e->hl.twin->alloc_mode = mode;
  return MUAVTA_OK;
}

int muavta_sync(MuavtaEnv* e) {
  if (!e) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  if (e->hl.twin) { int rc = muavta_sync(e->hl.twin); if (rc) { e->err = e->hl.twin->err; return rc; } }  // everything queued on the handle: both lanes
  MAIN_OP(e);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

int muavta_wait_stream(MuavtaEnv* e, void* other_stream) {  // work queued on the handle from now on starts after what `other_stream` holds now
  if (!e) return MUAVTA_E_ARG;
  if (e->hl.twin) { int rc = muavta_wait_stream(e->hl.twin, other_stream); if (rc) { e->err = e->hl.twin->err; return rc; } }
  DeviceScope scope_(e->device);
  hipEvent_t ev = nullptr;
  HIPCHK(e, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  hipError_t r = hipEventRecord(ev, (hipStream_t)other_stream);
  if (r == hipSuccess) r = hipStreamWaitEvent(e->stream, ev, 0);
  // sub-batches: a part's stream is ordered after the main stream's work at its next launch (fork_part), so the wait carries over
  for (int p = 0; p < e->n_parts; p++) e->part_fork_needed[p] = true;
  hipEventDestroy(ev);  // (destruction is deferred by the runtime until the event has completed)
  if (r != hipSuccess) { e->err = std::string("muavta_wait_stream: ") + hipGetErrorString(r); return MUAVTA_E_HIP; }
  return MUAVTA_OK;
}

int muavta_last_kernel_ms(MuavtaEnv* e, float* ms) {
  if (!e || !ms) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  if (!e->n_rollouts) { e->err = "no rollout launched yet"; return MUAVTA_E_STATE; }
  if (e->timing_stale) { e->err = "muavta_last_kernel_ms: the last rollout was a muavta_rollout_part launch, which records no event pair"; return MUAVTA_E_STATE; }
  const int evi = (int)((e->n_rollouts - 1) % MuavtaEnv::EV_RING);
  HIPCHK(e, hipEventSynchronize(e->ev1[evi]));
  HIPCHK(e, hipEventElapsedTime(ms, e->ev0[evi], e->ev1[evi]));
  e->last_ms = *ms;
  return MUAVTA_OK;
}

int muavta_kernel_ms_history(MuavtaEnv* e, float* ms, int32_t n) {  // durations of the last n rollout launches, oldest first
  if (!e || !ms || n < 1 || n > MuavtaEnv::EV_RING) { if (e) e->err = "muavta_kernel_ms_history: 1 <= n <= 64"; return MUAVTA_E_ARG; }
  if ((unsigned long long)n > e->hl.n_launches) { e->err = "fewer rollouts launched than asked for"; return MUAVTA_E_STATE; }
  if (e->timing_stale) { e->err = "muavta_kernel_ms_history: the last rollout was a muavta_rollout_part launch, which records no event pair"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  for (int k = 0; k < n; k++) {  // (with two state lanes consecutive launches alternate between the lanes' event rings and may overlap on the device)
    const unsigned long long h = e->hl.n_launches - (unsigned long long)n + (unsigned long long)k;
    MuavtaEnv* L = lane_by_id(e, e->hl.ring_lane[h % MuavtaEnv::HandleLevel::RING]);
    const unsigned long long no = e->hl.ring_no[h % MuavtaEnv::HandleLevel::RING];
    if (!L || L->n_rollouts - no > (unsigned long long)MuavtaEnv::EV_RING) { e->err = "muavta_kernel_ms_history: that launch's event pair has been reused"; return MUAVTA_E_STATE; }
    const int evi = (int)(no % MuavtaEnv::EV_RING);
    HIPCHK(e, hipEventSynchronize(L->ev1[evi]));
    HIPCHK(e, hipEventElapsedTime(&ms[k], L->ev0[evi], L->ev1[evi]));
  }
  e->last_ms = ms[n - 1];
  return MUAVTA_OK;
}

int muavta_launch_gaps_ms(MuavtaEnv* e, float* ms, int32_t n) {  // idle time of the handle's stream between the last n rollout launches: n - 1 gaps, oldest first
  if (!e || !ms || n < 2 || n > MuavtaEnv::EV_RING) { if (e) e->err = "muavta_launch_gaps_ms: 2 <= n <= 64"; return MUAVTA_E_ARG; }
  if ((unsigned long long)n > e->hl.n_launches) { e->err = "fewer rollouts launched than asked for"; return MUAVTA_E_STATE; }
  if (e->timing_stale) { e->err = "muavta_launch_gaps_ms: the last rollout was a muavta_rollout_part launch, which records no event pair"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  for (int k = 0; k + 1 < n; k++) {  // end of launch i .. start of launch i + 1 (NEGATIVE when they ran on different lanes and overlapped)
    const unsigned long long ha = e->hl.n_launches - (unsigned long long)n + (unsigned long long)k, hb = ha + 1;
    MuavtaEnv* La = lane_by_id(e, e->hl.ring_lane[ha % MuavtaEnv::HandleLevel::RING]);
    MuavtaEnv* Lb = lane_by_id(e, e->hl.ring_lane[hb % MuavtaEnv::HandleLevel::RING]);
    const unsigned long long na = e->hl.ring_no[ha % MuavtaEnv::HandleLevel::RING], nb = e->hl.ring_no[hb % MuavtaEnv::HandleLevel::RING];
    if (!La || !Lb || La->n_rollouts - na > (unsigned long long)MuavtaEnv::EV_RING || Lb->n_rollouts - nb > (unsigned long long)MuavtaEnv::EV_RING) {
      e->err = "muavta_launch_gaps_ms: an event pair has been reused"; return MUAVTA_E_STATE;
    }
    const int a = (int)(na % MuavtaEnv::EV_RING), b = (int)(nb % MuavtaEnv::EV_RING);
    HIPCHK(e, hipEventSynchronize(Lb->ev0[b]));
    HIPCHK(e, hipEventSynchronize(La->ev1[a]));
    if (La == Lb) HIPCHK(e, hipEventElapsedTime(&ms[k], La->ev1[a], Lb->ev0[b]));
    else {  // events of two streams: elapsed time in either direction, signed
      float fwd = 0.f;
      hipError_t r = hipEventElapsedTime(&fwd, La->ev1[a], Lb->ev0[b]);
      if (r != hipSuccess) { e->err = std::string("hipEventElapsedTime: ") + hipGetErrorString(r); return MUAVTA_E_HIP; }
      ms[k] = fwd;
    }
  }
  return MUAVTA_OK;
}

int muavta_last_seed_ms(MuavtaEnv* e, float* ms) {  // the RNG seeding kernel that preceded the last muavta_rollout (0 without seeds)
  if (!e || !ms) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  *ms = 0.f;
  if (!e->last_seeded) return MUAVTA_OK;
  HIPCHK(e, hipEventSynchronize(e->ev_seeded[e->last_seed_slot]));
  HIPCHK(e, hipEventElapsedTime(ms, e->ev_seed0[e->last_seed_slot], e->ev_seeded[e->last_seed_slot]));
  return MUAVTA_OK;
}

int muavta_observe(MuavtaEnv* e, float* tasks, uint64_t* legal, uint8_t* pad, float* agents, float* flags) {
  if (!e) return MUAVTA_E_ARG;
  if (!e->did_reset) { e->err = "observe before reset"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  const size_t N = (size_t)e->n_envs, mt = (size_t)e->P.max_tasks, nA = (size_t)e->P.n_agents;
  if (tasks) HIPCHK(e, hipMemcpyAsync(tasks, e->O.tasks, N * mt * 21 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (legal) HIPCHK(e, hipMemcpyAsync(legal, e->O.legal, N * nA * ((mt + 63) / 64) * sizeof(uint64_t), hipMemcpyDeviceToHost, e->stream));
  if (pad) HIPCHK(e, hipMemcpyAsync(pad, e->O.pad, N * mt, hipMemcpyDeviceToHost, e->stream));
  if (agents) HIPCHK(e, hipMemcpyAsync(agents, e->O.agents, N * nA * 9 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  if (flags) HIPCHK(e, hipMemcpyAsync(flags, e->O.flags, N * 5 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

// ---- token builders --------------------------------------------------------------------------
static int token_dims(int kind, int* dt, int* da) {
  if (kind == MUAVTA_TOK_PAIR) { *dt = 13; *da = 12; }
  else if (kind == MUAVTA_TOK_PAIR_RAW) { *dt = 9; *da = 11; }
  else if (kind == MUAVTA_TOK_ESCORT) { *dt = 22; *da = 16; }
  else return MUAVTA_E_ARG;
  return MUAVTA_OK;
}
int muavta_tokens_device(MuavtaEnv* e, int32_t kind, int32_t max_tasks, int32_t max_agents, float* task_feats, uint8_t* task_mask,
                         int32_t* task_ids, float* agent_feats, uint8_t* agent_mask, int32_t* agent_ids, float* edge_valid, int32_t* n_urgent,
                         float* expert_mask, int32_t* replanned) {
  int dt, da;
  if (!e || token_dims(kind, &dt, &da) || max_tasks < 1 || max_agents < 1 || max_tasks > 4096 || max_agents > 4096 || !task_feats || !task_mask ||
      !task_ids || !agent_feats || !agent_mask || !agent_ids || !edge_valid) { if (e) e->err = "muavta_tokens: bad argument"; return MUAVTA_E_ARG; }
  if (!e->did_reset) { e->err = "tokens before reset"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  DISPATCH(e, launch_tokens<TL>(e, kind, max_tasks, max_agents, task_feats, task_mask, task_ids, agent_feats, agent_mask, agent_ids, edge_valid, n_urgent, expert_mask, replanned));
  HIPCHK(e, hipGetLastError());
  return MUAVTA_OK;
}
int muavta_tokens(MuavtaEnv* e, int32_t kind, int32_t max_tasks, int32_t max_agents, float* task_feats, uint8_t* task_mask,
                  int32_t* task_ids, float* agent_feats, uint8_t* agent_mask, int32_t* agent_ids, float* edge_valid, int32_t* n_urgent,
                  float* expert_mask, int32_t* replanned) {
  int dt, da;
  if (!e || token_dims(kind, &dt, &da) || max_tasks < 1 || max_agents < 1) { if (e) e->err = "muavta_tokens: bad argument"; return MUAVTA_E_ARG; }
  DeviceScope scope_(e->device);
  const size_t N = (size_t)e->n_envs, MT = (size_t)max_tasks, MA = (size_t)max_agents;
  const size_t sz[10] = {N * MT * dt * 4, N * MT, N * MT * 4, N * MA * da * 4, N * MA, N * MA * 4, N * MA * MT * 4, N * 4, N * MA * MT * 4, N * 4};
  size_t off[11] = {0};
  for (int i = 0; i < 10; i++) off[i + 1] = off[i] + ((sz[i] + 255) & ~(size_t)255);
  if (off[10] > e->tok_bytes) {
    if (e->d_tok) hipFree(e->d_tok);
    e->d_tok = nullptr; e->tok_bytes = 0;
    HIPCHK(e, hipMalloc(&e->d_tok, off[10]));
    e->tok_bytes = off[10];
  }
  char* b = (char*)e->d_tok;
  int rc = muavta_tokens_device(e, kind, max_tasks, max_agents, (float*)(b + off[0]), (uint8_t*)(b + off[1]), (int32_t*)(b + off[2]),
                                (float*)(b + off[3]), (uint8_t*)(b + off[4]), (int32_t*)(b + off[5]), (float*)(b + off[6]), (int32_t*)(b + off[7]),
                                expert_mask ? (float*)(b + off[8]) : nullptr, replanned ? (int32_t*)(b + off[9]) : nullptr);
  if (rc) return rc;
  void* host[10] = {task_feats, task_mask, task_ids, agent_feats, agent_mask, agent_ids, edge_valid, n_urgent, expert_mask, replanned};
  for (int i = 0; i < 10; i++)
    if (host[i]) HIPCHK(e, hipMemcpyAsync(host[i], b + off[i], sz[i], hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

int muavta_context_device(MuavtaEnv* e, int32_t kind, int32_t max_tasks, float* context) {
  if (!e || !context || (kind != MUAVTA_TOK_PAIR && kind != MUAVTA_TOK_PAIR_RAW) || max_tasks < 1 || max_tasks > 4096) {
    if (e) e->err = "muavta_context: kind MUAVTA_TOK_PAIR (8 floats per env) or MUAVTA_TOK_PAIR_RAW (1), max_tasks >= 1"; return MUAVTA_E_ARG;
  }
  if (!e->did_reset) { e->err = "context before reset"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  DISPATCH(e, hipLaunchKernelGGL(k_context<TL>, dim3(e->n_envs), dim3(WG), Lds<TL>::bytes(), e->stream, (const DevCtx*)e->d_ctx, (int)(kind == MUAVTA_TOK_PAIR_RAW), max_tasks, context));
  HIPCHK(e, hipGetLastError());
  return MUAVTA_OK;
}
int muavta_context(MuavtaEnv* e, int32_t kind, int32_t max_tasks, float* context) {
  if (!e || !context) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  const size_t bytes = (size_t)e->n_envs * (kind == MUAVTA_TOK_PAIR_RAW ? 1 : 8) * sizeof(float);
  if (bytes > e->tok_bytes) {  // (shares the staging buffer of muavta_tokens' host variant)
    HIPCHK(e, hipStreamSynchronize(e->stream));
    if (e->d_tok) hipFree(e->d_tok);
    e->d_tok = nullptr; e->tok_bytes = 0;
    HIPCHK(e, hipMalloc(&e->d_tok, bytes));
    e->tok_bytes = bytes;
  }
  if (int rc = muavta_context_device(e, kind, max_tasks, (float*)e->d_tok)) return rc;
  HIPCHK(e, hipMemcpyAsync(context, e->d_tok, bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

int muavta_call(MuavtaEnv* e, int32_t env_index, int32_t op, const int32_t* iargs, double darg, int32_t* out) {
  if (!e || !out || op < 0 || op >= MUAVTA_OP_COUNT_ || env_index < 0 || env_index >= e->n_envs) { if (e) e->err = "muavta_call: bad argument"; return MUAVTA_E_ARG; }
  if (!e->did_reset) { e->err = "muavta_call before reset"; return MUAVTA_E_STATE; }
  CallArgs a;
  memset(&a, 0, sizeof(a));
  a.op = op; a.env = env_index; a.d = darg;
  if (iargs) memcpy(a.i, iargs, sizeof(a.i));
  const bool has_agent = op != MUAVTA_OP_SYNC_ESCORTS && op != MUAVTA_OP_RETIRE_ESCORT;
  if (has_agent && (a.i[0] < 0 || a.i[0] >= e->P.n_agents)) { e->err = "muavta_call: agent id out of range"; return MUAVTA_E_ARG; }
  if (op == MUAVTA_OP_SET_QUEUE && (a.i[1] < 0 || a.i[1] > 6)) { e->err = "muavta_call(SET_QUEUE): at most 6 tasks"; return MUAVTA_E_ARG; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  if (!e->d_call_out) HIPCHK(e, hipMalloc((void**)&e->d_call_out, MUAVTA_CALL_OUT * sizeof(int32_t)));
  DISPATCH(e, hipLaunchKernelGGL(k_call<TL>, dim3(1), dim3(WG), Lds<TL>::bytes(), e->stream, (const DevCtx*)e->d_ctx, a, e->d_call_out));
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(out, e->d_call_out, MUAVTA_CALL_OUT * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->host_valid = false;
  return MUAVTA_OK;
}

// ---- RCCL, loaded on first use --------------------------------------------------------------------------------
namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};
Rccl* rccl() {
  static Rccl R;
  if (R.lib || !R.err.empty()) return &R;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char* n : names) if ((R.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;   // a copy already in the process (PyTorch's)
  if (!R.lib) for (const char* n : names) if ((R.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!R.lib) { R.err = std::string("librccl not found: ") + dlerror(); return &R; }
  *(void**)&R.GetUniqueId = dlsym(R.lib, "ncclGetUniqueId");
  *(void**)&R.CommInitRank = dlsym(R.lib, "ncclCommInitRank");
  *(void**)&R.CommDestroy = dlsym(R.lib, "ncclCommDestroy");
  *(void**)&R.AllReduce = dlsym(R.lib, "ncclAllReduce");
  *(void**)&R.AllGather = dlsym(R.lib, "ncclAllGather");
  *(void**)&R.GetErrorString = dlsym(R.lib, "ncclGetErrorString");
  if (!R.GetUniqueId || !R.CommInitRank || !R.CommDestroy || !R.AllReduce || !R.AllGather || !R.GetErrorString) { R.err = "librccl lacks an expected symbol"; R.lib = nullptr; }
  return &R;
}
}  // namespace
#define NCCLCHK(env, expr)                                                                        \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess) { (env)->err = std::string(#expr) + ": " + rccl()->GetErrorString(r_); return MUAVTA_E_HIP; } \
  } while (0)

int muavta_comm_uid(uint8_t* uid) {
  if (!uid) return MUAVTA_E_ARG;
  Rccl* R = rccl();
  if (!R->lib) { g_create_error = R->err; return MUAVTA_E_NO_DEVICE; }
  static_assert(sizeof(ncclUniqueId) == MUAVTA_COMM_UID_BYTES, "RCCL unique id size");
  ncclUniqueId id;
  ncclResult_t r = R->GetUniqueId(&id);
  if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + R->GetErrorString(r); return MUAVTA_E_HIP; }
  memcpy(uid, &id, sizeof(id));
  return MUAVTA_OK;
}
int muavta_comm_init(MuavtaEnv* e, int32_t rank, int32_t n_ranks, const uint8_t* uid) {
  if (!e || !uid || n_ranks < 1 || rank < 0 || rank >= n_ranks) { if (e) e->err = "muavta_comm_init: bad arguments"; return MUAVTA_E_ARG; }
  if (e->comm) { e->err = "muavta_comm_init: this handle already has a communicator"; return MUAVTA_E_STATE; }
  Rccl* R = rccl();
  if (!R->lib) { e->err = R->err; return MUAVTA_E_NO_DEVICE; }
  DeviceScope scope_(e->device);
  ncclUniqueId id;
  memcpy(&id, uid, sizeof(id));
  // staging first: a failure below must not leave a communicator behind that has nowhere to stage (a retry would be refused
  // as "already has a communicator" and the next all-reduce would touch a null buffer)
  void* staging = nullptr;
  HIPCHK(e, hipMalloc(&staging, (size_t)(64 + 64 * n_ranks + 128) * 8));
  ncclComm_t comm = nullptr;
  const ncclResult_t r = R->CommInitRank(&comm, n_ranks, id, rank);
  if (r != ncclSuccess) {
    hipFree(staging);
    e->err = std::string("ncclCommInitRank: ") + R->GetErrorString(r);
    return MUAVTA_E_HIP;
  }
  e->comm = comm; e->d_comm = staging;
  e->comm_rank = rank; e->comm_ranks = n_ranks;
  return MUAVTA_OK;
}
int muavta_allreduce_metrics(MuavtaEnv* e, const double* f_partials, int32_t nf, const int64_t* counters, int32_t nc, double* f_total, int64_t* c_total) {
  if (!e || nf < 0 || nc < 0 || nf > 64 || nc > 64 || (nf && (!f_partials || !f_total)) || (nc && (!counters || !c_total))) { if (e) e->err = "muavta_allreduce_metrics: bad arguments"; return MUAVTA_E_ARG; }
  if (!e->comm) { e->err = "muavta_allreduce_metrics before muavta_comm_init"; return MUAVTA_E_STATE; }
  Rccl* R = rccl();
  DeviceScope scope_(e->device);
  const int n = e->comm_ranks;
  double* fs = (double*)e->d_comm; double* fr = fs + 64;
  int64_t* cs = (int64_t*)(fr + (size_t)64 * n); int64_t* cr = cs + 64;
  if (nf) {
    HIPCHK(e, hipMemcpyAsync(fs, f_partials, (size_t)nf * 8, hipMemcpyHostToDevice, e->stream));
    NCCLCHK(e, R->AllGather(fs, fr, (size_t)nf, ncclDouble, e->comm, e->stream));
  }
  if (nc) {
    HIPCHK(e, hipMemcpyAsync(cs, counters, (size_t)nc * 8, hipMemcpyHostToDevice, e->stream));
    NCCLCHK(e, R->AllReduce(cs, cr, (size_t)nc, ncclInt64, ncclSum, e->comm, e->stream));
  }
  std::vector<double> gathered((size_t)nf * n);
  if (nf) HIPCHK(e, hipMemcpyAsync(gathered.data(), fr, gathered.size() * 8, hipMemcpyDeviceToHost, e->stream));
  if (nc) HIPCHK(e, hipMemcpyAsync(c_total, cr, (size_t)nc * 8, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  for (int k = 0; k < nf; k++) {  // rank order: the same bits on every rank, whatever the ring order
    double s = 0.0;
    for (int r = 0; r < n; r++) s += gathered[(size_t)r * nf + k];
    f_total[k] = s;
  }
  return MUAVTA_OK;
}
int muavta_comm_destroy(MuavtaEnv* e) {
  if (!e) return MUAVTA_E_ARG;
  if (e->comm) {
    DeviceScope scope_(e->device);
    hipStreamSynchronize(e->stream);
    rccl()->CommDestroy(e->comm);
    e->comm = nullptr;
    hipFree(e->d_comm);
    e->d_comm = nullptr;
  }
  return MUAVTA_OK;
}

int muavta_set_release_log(MuavtaEnv* e, int32_t enable) {
  if (!e) return MUAVTA_E_ARG;
  if (e->hl.twin) { int rc = muavta_set_release_log(e->hl.twin, enable); if (rc) { e->err = e->hl.twin->err; return rc; } }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  if (enable && !e->d_rel) {
    const size_t bytes = (size_t)e->n_envs * (1 + MUAVTA_REL_ROW * e->T) * sizeof(double);
    HIPCHK(e, hipMalloc((void**)&e->d_rel, bytes));
    HIPCHK(e, hipMemset(e->d_rel, 0, bytes));
  } else if (!enable && e->d_rel) {
    hipFree(e->d_rel);
    e->d_rel = nullptr;
  }
  return MUAVTA_OK;
}

int muavta_refresh_observation(MuavtaEnv* e) {  // rebuild the obs tensors from the current state (after muavta_set)
  if (!e) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  DISPATCH(e, hipLaunchKernelGGL(k_observe<TL>, dim3(e->n_envs), dim3(WG), Lds<TL>::bytes(), e->stream, (const DevCtx*)e->d_ctx));
  HIPCHK(e, hipGetLastError());
  e->host_valid = false;  // (the kernel refreshes the derived initTime / doneTime rows of the HBM record)
  return MUAVTA_OK;
}

int muavta_step_result(MuavtaEnv* e, double* reward, uint8_t* done) {
  if (!e) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  if (reward) HIPCHK(e, hipMemcpyAsync(reward, e->O.reward, (size_t)e->n_envs * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  if (done) HIPCHK(e, hipMemcpyAsync(done, e->O.done, (size_t)e->n_envs, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

int muavta_metrics(MuavtaEnv* e, double* out) {
  if (!e || !out) return MUAVTA_E_ARG;
  if (!e->did_reset) { e->err = "metrics before reset"; return MUAVTA_E_STATE; }
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  DISPATCH(e, hipLaunchKernelGGL(k_metrics<TL>, dim3(e->n_envs), dim3(WG), Lds<TL>::bytes(), e->stream, (const DevCtx*)e->d_ctx, e->d_metrics));
  HIPCHK(e, hipGetLastError());
  HIPCHK(e, hipMemcpyAsync(out, e->d_metrics, (size_t)e->n_envs * MUAVTA_N_METRICS * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  int rc = sync_host(e);
  if (rc) return rc;
  DISPATCH(e, rc = check_errors<TL>(e));
  return rc;
}

int muavta_get(MuavtaEnv* e, MuavtaField field, void* dst, size_t bytes) {
  if (!e || !dst) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  if (field == MUAVTA_F_RELEASE_LOG) {
    const size_t want = (size_t)e->n_envs * (1 + MUAVTA_REL_ROW * e->T) * sizeof(double);
    if (!e->d_rel) { e->err = "release log is off (muavta_set_release_log)"; return MUAVTA_E_STATE; }
    if (bytes != want) { e->err = "muavta_get(RELEASE_LOG): wrong size"; return MUAVTA_E_ARG; }
    MAIN_OP(e);
    HIPCHK(e, hipMemcpyAsync(dst, e->d_rel, want, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(e, hipStreamSynchronize(e->stream));
    return MUAVTA_OK;
  }
  int rc = sync_host(e);
  if (rc) return rc;
  DISPATCH(e, rc = gather<TL>(e, field, dst, bytes, false));
  return rc;
}

int muavta_set(MuavtaEnv* e, MuavtaField field, const void* src, size_t bytes) {
  if (!e || !src) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  int rc = sync_host(e);
  if (rc) return rc;
  DISPATCH(e, rc = gather<TL>(e, field, const_cast<void*>(src), bytes, true));
  if (rc) return rc;
  DISPATCH(e, forget_obs_rows<TL>(e));
  HIPCHK(e, hipMemcpyAsync(e->blobs, e->host_blobs.data(), e->host_blobs.size(), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->cold, e->host_cold.data(), e->host_cold.size(), hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

int muavta_get_state(MuavtaEnv* e, void* dst, size_t bytes) {  // [N x EnvState | N x EnvCold]
  if (!e || !dst || bytes != (size_t)e->n_envs * (e->state_bytes + e->cold_bytes)) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  HIPCHK(e, hipMemcpyAsync(dst, e->blobs, (size_t)e->n_envs * e->state_bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipMemcpyAsync((char*)dst + (size_t)e->n_envs * e->state_bytes, e->cold, (size_t)e->n_envs * e->cold_bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}
int muavta_set_state(MuavtaEnv* e, const void* src, size_t bytes) {
  if (!e || !src || bytes != (size_t)e->n_envs * (e->state_bytes + e->cold_bytes)) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  e->host_blobs.assign((const unsigned char*)src, (const unsigned char*)src + (size_t)e->n_envs * e->state_bytes);
  DISPATCH(e, forget_obs_rows<TL>(e));  // (the observation buffer belongs to another moment than the restored state)
  HIPCHK(e, hipMemcpyAsync(e->blobs, e->host_blobs.data(), (size_t)e->n_envs * e->state_bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipMemcpyAsync(e->cold, (const char*)src + (size_t)e->n_envs * e->state_bytes, (size_t)e->n_envs * e->cold_bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->host_valid = false;
  e->did_reset = true;
  return MUAVTA_OK;
}
int muavta_get_rng(MuavtaEnv* e, void* dst, size_t bytes) {  // raw MT tapes, for checkpoint/resume next to get_state
  size_t need = e ? (size_t)e->n_envs * MUAVTA_RNG_STREAMS * MUAVTA_RNG_WORDS * 4 : 0;
  if (!e || !dst || bytes != need) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  HIPCHK(e, hipMemcpyAsync(dst, e->tapes, bytes, hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}
int muavta_set_rng(MuavtaEnv* e, const void* src, size_t bytes) {
  size_t need = e ? (size_t)e->n_envs * MUAVTA_RNG_STREAMS * MUAVTA_RNG_WORDS * 4 : 0;
  if (!e || !src || bytes != need) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  HIPCHK(e, hipMemcpyAsync(e->tapes, src, bytes, hipMemcpyHostToDevice, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

int muavta_device_ptrs(MuavtaEnv* e, void** state, void** obs_tasks, void** obs_legal, void** obs_agents, void** metrics, void** stream) {
  if (!e) return MUAVTA_E_ARG;
  if (state) *state = e->blobs;
  if (obs_tasks) *obs_tasks = e->O.tasks;
  if (obs_legal) *obs_legal = e->O.legal;
  if (obs_agents) *obs_agents = e->O.agents;
  if (metrics) *metrics = e->d_metrics;
  if (stream) *stream = (void*)e->stream;
  return MUAVTA_OK;
}

int muavta_rollout_metrics(MuavtaEnv* e, double* out) {  // metrics written by the last muavta_rollout (no extra kernel)
  if (!e || !out) return MUAVTA_E_ARG;
  DeviceScope scope_(e->device);
  MAIN_OP(e);
  HIPCHK(e, hipMemcpyAsync(out, e->d_metrics, (size_t)e->n_envs * MUAVTA_N_METRICS * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIPCHK(e, hipStreamSynchronize(e->stream));
  return MUAVTA_OK;
}

// the lane that holds the rollout launched just before the latest one, if that launch ran on the OTHER lane (else nullptr: overwritten)
static MuavtaEnv* prev_batch_lane(MuavtaEnv* e) {
  if (!e->hl.twin || e->hl.n_launches < 2) return nullptr;
  const int R = MuavtaEnv::HandleLevel::RING;
  const int last = e->hl.ring_lane[(e->hl.n_launches - 1) % R], prev = e->hl.ring_lane[(e->hl.n_launches - 2) % R];
  if (last == prev) return nullptr;
  MuavtaEnv* t = lane_by_id(e, prev);
  return (t && t != e && t->n_rollouts == e->hl.ring_no[(e->hl.n_launches - 2) % R] + 1) ? t : nullptr;  // (and nothing else was launched on that lane since)
}
int muavta_rollout_metrics_back(MuavtaEnv* e, int32_t back, double* out) {  // back 0: the last seeded batch (= muavta_rollout_metrics); 1: the one before it, on the other lane
  if (!e || !out || back < 0 || back > 1) return MUAVTA_E_ARG;
  if (back == 0) return muavta_rollout_metrics(e, out);
  MuavtaEnv* t = prev_batch_lane(e);
  if (!t) { e->err = "muavta_rollout_metrics_back: the batch before the latest one is gone — it ran on the same lane (the latest rollout found it finished, or there is one lane only); muavta_set_lanes(h, 2) makes seeded rollouts always alternate"; return MUAVTA_E_STATE; }
  int rc = muavta_rollout_metrics(t, out);
  if (rc) e->err = t->err;
  return rc;
}
int muavta_error_flags_back(MuavtaEnv* e, int32_t back, int32_t* out) {  // MUAVTA_F_ERROR of the batch `back` launches ago (0 or 1)
  if (!e || !out || back < 0 || back > 1) return MUAVTA_E_ARG;
  MuavtaEnv* L = back == 0 ? e : prev_batch_lane(e);
  if (!L) { e->err = "muavta_error_flags_back: the batch before the latest one is gone (it ran on the same lane)"; return MUAVTA_E_STATE; }
  int rc = muavta_get(L, MUAVTA_F_ERROR, out, (size_t)L->n_envs * sizeof(int32_t));
  if (rc && L != e) e->err = L->err;
  return rc;
}
int muavta_set_slot_cap(MuavtaEnv* e, int32_t cap) {  // test hook: an env may use at most `cap` of its tile's task slots (0: all of them)
  if (!e || cap < 0 || cap > e->T) { if (e) e->err = "muavta_set_slot_cap: 0 .. the tile's slot count"; return MUAVTA_E_ARG; }
  DeviceScope scope_(e->device);
  if (e->hl.twin) { int rc = muavta_set_slot_cap(e->hl.twin, cap); if (rc) { e->err = e->hl.twin->err; return rc; } }
  MAIN_OP(e);
  HIPCHK(e, hipStreamSynchronize(e->stream));
  e->P.slot_cap = (cap > 0 && cap < e->T) ? cap : 0;
  HIPCHK(e, hipMemcpy((char*)e->d_ctx + offsetof(DevCtx, P) + offsetof(DevParams, slot_cap), &e->P.slot_cap, sizeof(int32_t), hipMemcpyHostToDevice));
  return MUAVTA_OK;
}
int muavta_set_lanes(MuavtaEnv* e, int32_t lanes) {
  if (!e || lanes < 0 || lanes > 2) { if (e) e->err = "muavta_set_lanes: 0 (second lane on demand), 1 (one lane) or 2 (always alternate)"; return MUAVTA_E_ARG; }
  DeviceScope scope_(e->device);
  if (lanes == 2) { int rc = ensure_twin(e); if (rc) { e->err = "muavta_set_lanes: the second lane could not be created: " + g_create_error; return rc; } }
  if (lanes == 1 && e->hl.twin) {  // back to one lane: the second lane's batch completes and its memory is released
    muavta_destroy(e->hl.twin);
    e->hl.twin = nullptr;
  }
  e->hl.lanes_mode = lanes;
  return MUAVTA_OK;
}
int muavta_lanes(const MuavtaEnv* e, int32_t* mode, int32_t* allocated) {
  if (!e) return MUAVTA_E_ARG;
  if (mode) *mode = e->hl.lanes_mode;
  if (allocated) *allocated = e->hl.twin ? 2 : 1;
  return MUAVTA_OK;
}

int muavta_lsap(int32_t device, const double* cost, int32_t n, int32_t nr, int32_t nc, int64_t* row, int64_t* col) {
  return muavta_lsap_impl(device, cost, n, nr, nc, row, col, MUAVTA_LSAP_AUTO);
}
int muavta_lsap_impl(int32_t device, const double* cost, int32_t n, int32_t nr, int32_t nc, int64_t* row, int64_t* col, int32_t impl) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "muavta_lsap: no HIP device"; return MUAVTA_E_NO_DEVICE; }
  if (!cost || !row || !col || n < 1 || nr < 1 || nc < 1) { g_create_error = "muavta_lsap: bad arguments"; return MUAVTA_E_ARG; }
  int mn = nr < nc ? nr : nc, mx = nr < nc ? nc : nr;
  if (mn > Tile64::A || mx > Tile64::T) { g_create_error = "muavta_lsap: at most 64 x 128"; return MUAVTA_E_ARG; }
  // scipy.optimize.linear_sum_assignment raises ValueError("matrix contains invalid numeric entries") for NaN / -inf
  for (size_t i = 0, m = (size_t)n * nr * nc; i < m; i++)
    if (std::isnan(cost[i]) || cost[i] == -INFINITY) { g_create_error = "muavta_lsap: matrix contains invalid numeric entries (NaN or -inf)"; return MUAVTA_E_ARG; }
  const bool fits_reg = mn <= TileLsapReg::A && mx <= TileLsapReg::T;
  if (impl < MUAVTA_LSAP_AUTO || impl > MUAVTA_LSAP_REGISTERS || (impl == MUAVTA_LSAP_REGISTERS && !fits_reg)) {
    g_create_error = "muavta_lsap_impl: unknown solver, or problem beyond 32 x 64 for the register solver"; return MUAVTA_E_ARG;
  }
  const bool use_reg = impl == MUAVTA_LSAP_REGISTERS || (impl == MUAVTA_LSAP_AUTO && fits_reg);
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); hipFree(dc); hipFree(dr); hipFree(dcl); hipFree(dst); return MUAVTA_E_HIP; } } while (0)
  double* dc = nullptr; int64_t *dr = nullptr, *dcl = nullptr; int32_t* dst = nullptr;
  DeviceScope scope_(device);
  size_t cb = (size_t)n * nr * nc * sizeof(double), rb = (size_t)n * mn * sizeof(int64_t);
  CK(hipMalloc(&dc, cb)); CK(hipMalloc(&dr, rb)); CK(hipMalloc(&dcl, rb)); CK(hipMalloc(&dst, (size_t)n * sizeof(int32_t)));
  CK(hipMemcpy(dc, cost, cb, hipMemcpyHostToDevice));
  if (use_reg) {
    size_t lds = Lds<TileLsapReg>::bytes();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsap<TileLsapReg, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_lsap<TileLsapReg, true>), dim3(n), dim3(WG), lds, 0, dc, nr, nc, dr, dcl, dst);
  } else {
    size_t lds = Lds<TileLsapLds>::bytes();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lsap<TileLsapLds, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_lsap<TileLsapLds, false>), dim3(n), dim3(WG), lds, 0, dc, nr, nc, dr, dcl, dst);
  }
  CK(hipGetLastError());
  std::vector<int32_t> status((size_t)n);
  CK(hipMemcpy(status.data(), dst, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  CK(hipMemcpy(row, dr, rb, hipMemcpyDeviceToHost));
  CK(hipMemcpy(col, dcl, rb, hipMemcpyDeviceToHost));
#undef CK
  hipFree(dc); hipFree(dr); hipFree(dcl); hipFree(dst);
  for (int i = 0; i < n; i++)
    if (status[(size_t)i]) {  // scipy: ValueError("cost matrix is infeasible")
      g_create_error = "muavta_lsap: cost matrix " + std::to_string(i) + " is infeasible";
      return MUAVTA_E_ARG;
    }
  return MUAVTA_OK;
}

int muavta_domain_math(int32_t device, const double* x, const double* y, int32_t n, double* out_sqrt, double* out_div, double* out_div_neg) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "muavta_domain_math: no HIP device"; return MUAVTA_E_NO_DEVICE; }
  if (!x || !y || !out_sqrt || !out_div || !out_div_neg || n < 1) return MUAVTA_E_ARG;
  double* d[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); for (double* q : d) hipFree(q); return MUAVTA_E_HIP; } } while (0)
  DeviceScope scope_(device);
  const size_t bytes = (size_t)n * sizeof(double);
  for (double*& q : d) CK(hipMalloc(&q, bytes));
  CK(hipMemcpy(d[0], x, bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(d[1], y, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_domain_math, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], n, d[2], d[3], d[4]);
  CK(hipGetLastError());
  CK(hipMemcpy(out_sqrt, d[2], bytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(out_div, d[3], bytes, hipMemcpyDeviceToHost));
  CK(hipMemcpy(out_div_neg, d[4], bytes, hipMemcpyDeviceToHost));
#undef CK
  for (double* q : d) hipFree(q);
  return MUAVTA_OK;
}

int muavta_domain_log(int32_t device, const double* x, int32_t n, double* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "muavta_domain_log: no HIP device"; return MUAVTA_E_NO_DEVICE; }
  if (!x || !out || n < 1) return MUAVTA_E_ARG;
  double* d[2] = {nullptr, nullptr};
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); for (double* q : d) hipFree(q); return MUAVTA_E_HIP; } } while (0)
  DeviceScope scope_(device);
  const size_t bytes = (size_t)n * sizeof(double);
  for (double*& q : d) CK(hipMalloc(&q, bytes));
  CK(hipMemcpy(d[0], x, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_libm_log, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], n, d[1]);
  CK(hipGetLastError());
  CK(hipMemcpy(out, d[1], bytes, hipMemcpyDeviceToHost));
#undef CK
  for (double* q : d) hipFree(q);
  return MUAVTA_OK;
}

int muavta_domain_atan2(int32_t device, const double* y, const double* x, int32_t n, double* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "muavta_domain_atan2: no HIP device"; return MUAVTA_E_NO_DEVICE; }
  if (!x || !y || !out || n < 1) return MUAVTA_E_ARG;
  double* d[3] = {nullptr, nullptr, nullptr};
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); for (double* q : d) hipFree(q); return MUAVTA_E_HIP; } } while (0)
  DeviceScope scope_(device);
  const size_t bytes = (size_t)n * sizeof(double);
  for (double*& q : d) CK(hipMalloc(&q, bytes));
  CK(hipMemcpy(d[0], y, bytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(d[1], x, bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_libm_atan2, dim3((n + 255) / 256), dim3(256), 0, 0, d[0], d[1], n, d[2]);
  CK(hipGetLastError());
  CK(hipMemcpy(out, d[2], bytes, hipMemcpyDeviceToHost));
#undef CK
  for (double* q : d) hipFree(q);
  return MUAVTA_OK;
}

int muavta_avoid_obstacles(int32_t device, const double* agent_pos, const double* movement, int32_t n, const double* obstacles,
                           int32_t n_obstacles, double* out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { g_create_error = "muavta_avoid_obstacles: no HIP device"; return MUAVTA_E_NO_DEVICE; }
  if (!agent_pos || !movement || !out || n < 1 || n_obstacles < 0 || (n_obstacles > 0 && !obstacles)) return MUAVTA_E_ARG;
  double *dp = nullptr, *dm = nullptr, *dob = nullptr, *dout = nullptr;
#define CK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_create_error = std::string(#expr) + ": " + hipGetErrorString(e_); hipFree(dp); hipFree(dm); hipFree(dob); hipFree(dout); return MUAVTA_E_HIP; } } while (0)
  DeviceScope scope_(device);
  CK(hipMalloc(&dp, (size_t)n * 16)); CK(hipMalloc(&dm, (size_t)n * 16)); CK(hipMalloc(&dout, (size_t)n * 16));
  CK(hipMalloc(&dob, (size_t)(n_obstacles > 0 ? n_obstacles : 1) * 24));
  CK(hipMemcpy(dp, agent_pos, (size_t)n * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(dm, movement, (size_t)n * 16, hipMemcpyHostToDevice));
  if (n_obstacles > 0) CK(hipMemcpy(dob, obstacles, (size_t)n_obstacles * 24, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_avoid, dim3((n + 255) / 256), dim3(256), 0, 0, dp, dm, n, dob, n_obstacles, dout);
  CK(hipGetLastError());
  CK(hipMemcpy(out, dout, (size_t)n * 16, hipMemcpyDeviceToHost));
#undef CK
  hipFree(dp); hipFree(dm); hipFree(dob); hipFree(dout);
  return MUAVTA_OK;
}

}  // extern "C"
