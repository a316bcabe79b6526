// muavta_device.h — device-side simulation of ONE env instance by ONE workgroup (gfx950, wave64).
//
// Execution model: the env's state blob sits in LDS (EnvState).  A step is a sequence of phases separated
// by lds_sync().  Phases whose iterations are independent (pairwise sensing, the agent x task cost tile,
// distances, slot GC, observation rows, MT19937 regeneration) are spread over the 64 lanes.  The
// order-dependent bookkeeping of the reference is parallelised where its order can be kept exactly:
//   * "everything up to the first event, then lane 0 replays that one entity as the reference does, then the
//     wave resumes behind it" — the per-agent state machine and the threat update;
//   * "one entity per lane, order-dependent sums replayed in order" — action application (reward addends via
//     v_readlane, allocatedReqs of a shared task via a same-slot prefix), releaseAllTasks (agent lanes compact
//     their queues, slot lanes replay removeAgentCap in agent order), reveals / window expiry;
//   * the LSAP keeps scipy's scan order as per-column positions and never touches LDS inside the solve.
// What is left on lane 0 is entity creation (arrivals, threat spawns, escorts), engagements, task completion.
// Everything is f64 with -ffp-contract=off; FMAs are explicit where numpy emits them.
//
// Every routine cites the reference lines it restates (mUAV_TA/DroneEnv.py unless another file is
// named).  This file is the product path; it shares no code with oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <type_traits>

#include "muavta_state.h"

#define DEV __device__ __forceinline__
#define DEVN __device__ __noinline__

// 1: the observation rows' HBM operands are requested before the task-time rebuild (latency overlap, +28 live VGPRs there)
// k_rollout's issue-priority pacing of the envs that share a SIMD: 0 off, 1..3 = that priority for the env furthest behind,
// 4 = ranked (3 for the last, one less per neighbour further behind).  Measured on the headline tile (profiles/r02_pacing.txt):
// off 164 M env-steps/s, 1 -> 177 M, 4 -> 182 M; raising priority only while an env replans: 172 M.
#ifndef MUAVTA_PACE_PRIO
#define MUAVTA_PACE_PRIO 4
#endif
#ifndef MUAVTA_PACE_HOLD  // 0 off; n: a wave more than n steps ahead of a neighbour on its SIMD sleeps until it has caught up (muavta_kernels.hip)
#define MUAVTA_PACE_HOLD 0
#endif
#ifndef MUAVTA_PACE_HOLD_WINDOW
#define MUAVTA_PACE_HOLD_WINDOW 48
#endif
#ifndef MUAVTA_PACE_HOLD_SLEEP
#define MUAVTA_PACE_HOLD_SLEEP 16
#endif
#ifndef MUAVTA_PACE_HOLD_POLLS
#define MUAVTA_PACE_HOLD_POLLS 256
#endif
#ifndef MUAVTA_OBS_SADDR
#define MUAVTA_OBS_SADDR 1
#endif
#ifndef MUAVTA_LDS_ZERO_REG  // 1: the rollout phases address the LDS block through one pinned zero register (muavta_kernels.hip: lds_zero) — measured slower
#define MUAVTA_LDS_ZERO_REG 0
#endif
#ifndef MUAVTA_OBS_PREFETCH
#define MUAVTA_OBS_PREFETCH 0
#endif


#ifdef MUAVTA_DIAGNOSTIC_BUILD
#include "muavta_diag.h"   // ABL(), OBS_SKIP(): wrong-result timing switches of diagnostic builds
#else
#if defined(MUAVTA_ABLATE) || defined(MUAVTA_OBS_SKIP)
#error "MUAVTA_ABLATE / MUAVTA_OBS_SKIP produce wrong results: they need -DMUAVTA_DIAGNOSTIC_BUILD and are never part of the shipped library"
#endif
#define ABL(bit) 0
#define OBS_SKIP(mask) false
#endif

namespace muavta {

constexpr int WG = 64;  // one wave64 per env

// Phase separator for a ONE-wave workgroup: LDS instructions of a wave are executed in issue order, so
// a later ds_read of another lane's ds_write needs no s_barrier and, unlike __syncthreads(), no
// vmcnt(0) drain of the observation / tape stores still in flight.  Global-memory hand-offs between
// lanes (MT19937 regeneration) keep the full __syncthreads().
__device__ __forceinline__ void lds_sync() {
  static_assert(WG == 64, "lds_sync() assumes the workgroup is a single wave64");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// Publish point for the HBM-resident part of an env (EnvCold): placed at the entry and exit of every wave-parallel phase
// that reads or writes it, so that a lane's loads are issued only after every other lane's stores have been acknowledged
// by the memory system (one wave, one CU: the L1 is write-through and shared, no cache maintenance is needed).
__device__ __forceinline__ void cold_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

#include "muavta_math.h"

template <class TL>
struct alignas(16) Scratch {
  enum { A = TL::A, T = TL::T,
         COSTN_MIN = 4 * A > 2 * T ? 4 * A : 2 * T,  // act_f: 4 x A doubles; refresh_task_times: 2 x T u64
         COSTN = !TL::NO_COST_TILE ? A * T           // REGC / OTFC: only small staging arrays live here ...
               : (A <= 16 && COSTN_MIN < 144) ? 144 : COSTN_MIN };  // ... and the scratch still has to hold the 3 KB reset RNG window
  double cost[COSTN];  // LSAP cost tile, R x C row-major with R = min(nr, nc) (unless TL::REGC)
  double u[A], v[T], spc[T], resid[T];  // (the register-resident LSAP leaves u/v/spc to their other users)
  double press[TL::OTFC ? T : 1];       // Urgency-Coalition threat pressure per round task where the LDS solver (which owns spc) evaluates costs on the fly
  // index lists of the allocator / LSAP and scratch lists of the serial phases: agent ids, slot ids, row numbers (< 128), -1
  // markers, and Urgency-Pair's (rank | n_know << 8) — 16 bits each
  int16_t path[T], col4row[A], row4col[T], remaining[T], freeA[A], roundT[T];
  uint8_t SR[A], SC[T];
  uint8_t live_rank[A];                  // Urgency-Pair: rank of an agent among the live ones (255 = beyond the token pad)
  int32_t retarget_h;                    // threat whose get_closest_agent() the serial replay of its engagement left to the wave (-1: none)
  int16_t pair_info_big[T > 64 ? T : 1]; // Urgency-Pair per-slot (rank, n_know) when `remaining` is busy (LDS LSAP, T > 64)
};

#include "muavta_rng.h"

#ifdef MUAVTA_PROF
enum { PROF_N = 64 };  // slots 0..47: cycle accumulators, 48..63: event counters (x1000)
__device__ unsigned long long g_prof[PROF_N];
__device__ int g_prof_target = -1;  // >= 0: only this env's stamps are accumulated (tools/phase_profile.py --slowest)
// per-phase cycle accumulators of a diagnostic build: 64 + 1 (last stamp) u64 words in LDS right behind the Scratch tile
#define MUAVTA_PROF_LDS_BYTES 528
#define PROF(i) do { if (threadIdx.x == 0) { unsigned long long t_ = clock64(); prof_lds()[i] += t_ - prof_lds()[PROF_N]; prof_lds()[PROF_N] = t_; } } while (0)
#define PROF_COUNT(i, n) do { if (threadIdx.x == 0) prof_lds()[i] += (n); } while (0)
#else
#define PROF(i) do { } while (0)
#define PROF_COUNT(i, n) do { } while (0)
#endif

// muavta_allocate_scored: the caller's planner inputs in token layout (device pointers, kernel arguments), see include/muavta.h
struct ScoredDev {
  const float* scores;                  // [N, MA, MT] or null
  const double* pri;                    // [N, MT] or null
  const unsigned long long* reserved;   // [N] or null
  float* selected;                      // [N, MA, MT] or null
  int32_t* replanned;                   // [N] or null
  int32_t kind, MT, MA, gate, flags;
};

template <class TL>
struct Sim {
  typedef EnvState<TL> State;
#ifdef MUAVTA_PROF
  DEV unsigned long long* prof_lds() { return reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(&X) + sizeof(Scratch<TL>)); }
  DEV void prof_begin() { if (threadIdx.x < PROF_N) prof_lds()[threadIdx.x] = 0; if (threadIdx.x == 0) prof_lds()[PROF_N] = clock64(); lds_sync(); }
  DEV void prof_flush(int env) { lds_sync(); if (threadIdx.x < PROF_N && (g_prof_target < 0 || g_prof_target == env)) atomicAdd(&g_prof[threadIdx.x], prof_lds()[threadIdx.x]); }
#endif
  enum { A = TL::A, T = TL::T, H = TL::H, R = TL::R, E = TL::E, Q = TL::Q, KW = TL::KW };
  typedef EnvCold<TL> Cold;
  State& S;    // LDS
  Cold& C;     // HBM (L2-resident): requirement vectors, queue-entry times, init/done times, obstacles
  Scratch<TL>& X;
  double* rel_log = nullptr;  // optional per-env release log in HBM (muavta_set_release_log)
  // muavta_step_lists: the (agent, index) items of this env beyond the ones staged in S.act_* (which hold TL::A at a time), still
  // in the caller's rows in HBM: this env's row, its length and the next item to stage.  nullptr everywhere else.
  const int32_t* more_agent = nullptr;
  const int32_t* more_index = nullptr;
  int more_cap = 0, more_pos = 0;
  int tnow;                   // env.time_steps in a (uniform) register: it changes once per step, and every LDS read of it is ~100 cycles of latency
  const DevParams& P;
  uint32_t* tape;  // [4][1248] in HBM
  int lane;
  // LDS window of prefetched raw tape words consulted by next32(): normally the 8-word window in the state
  // blob; reset() points it at a 192-word-per-stream copy in the scratch tile (a reset draws ~100 words)
  const uint32_t* win_ptr;
  uint32_t win_len, win_stride;
  bool in_step = false;  // step() is running in this object (reclaim_slot_serial: which slots may be recycled on demand)

  __device__ Sim(State& s, Cold& c, Scratch<TL>& x, const DevParams& p, uint32_t* t)
      : S(s), C(c), X(x), P(p), tape(t), lane(opaque_lane()), win_ptr(&s.rng_win[0][0]), win_len(8), win_stride(8), tnow(s.time_steps) {}
  // The lane id, opaque to the optimiser.  A Sim built inside the 150-step loop of k_rollout would otherwise have every
  // lane-derived value (ballot prefix masks, lane & 7, lane < n ...) hoisted out of the loop, where they overflow the 128
  // VGPRs and come back as scratch (HBM) reloads at ~150 sites of the step; recomputing them is one or two VALU ops.
  static __device__ __forceinline__ int opaque_lane() {
    int l = threadIdx.x;
    asm volatile("" : "+v"(l));
    return l;
  }
  DEV void sync_clock() { tnow = S.time_steps; }
  // next_free_* / orgReqs / doneReqs / mission areas: LDS or the HBM record, by tile (QueueSide in muavta_state.h)
  DEV QueueSide<A, T, true>& qs() const {
    if constexpr (TL::SLIM) return static_cast<QueueSide<A, T, true>&>(C); else return static_cast<QueueSide<A, T, true>&>(S);
  }  // after the blob was (re)loaded or reset behind this object's back

  DEV void fail(int code) { if (S.error == 0) S.error = code; }
  // S.obs_rows = (rows of the handle's task tensor from which on pad rows are known to be there) | OBS_STATIC, or -1 (unknown).
  // OBS_STATIC: the columns of rows [0, n_open) that only change with a task's requirement / allocation vectors or with the open list
  // itself (id, current_reqs, alloc_reqs, type, unmet) are up to date in the handle's buffer; the observation writer then only
  // rewrites the columns that move every step (position, status, init / end time, age).  Cleared by everything that changes
  // allocatedReqs (the times_dirty sites), currentReqs, orgReqs, or the open list.  Any lane may clear it (same value).
  enum { OBS_STATIC = 1 << 16 };
  DEV void obs_static_clear() { const int v = S.obs_rows; if (v > 0) S.obs_rows = v & (OBS_STATIC - 1); }
  DEV double speed_of(int t) const {  // P.speed[t] without a memory access for a per-lane t
    return t == 0 ? P.speed[0] : t == 1 ? P.speed[1] : t == 2 ? P.speed[2] : t == 3 ? P.speed[3] : t == 4 ? P.speed[4]
         : t == 5 ? P.speed[5] : P.speed[6];
  }


#include "sim/rng.inc"
#include "sim/entities.inc"
#include "sim/reset.inc"
#include "sim/step.inc"
#include "sim/actions.inc"
#include "sim/world.inc"
#include "sim/observe.inc"
#include "sim/allocate.inc"
#include "sim/tokens.inc"
#include "sim/lsap.inc"
#include "sim/metrics.inc"
};


}  // namespace muavta

