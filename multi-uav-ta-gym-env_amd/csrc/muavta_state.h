// muavta_state.h — per-environment state blob, shared by host and device code.
//
// One env instance == one contiguous, 16-byte aligned blob.  The SAME bytes live (a) in HBM as
// `blob[n_envs]` and (b) in the LDS of the workgroup that simulates the env: kernels move it with
// coalesced 16-B/lane copies and run every step against the LDS copy.  Inside the blob every agent /
// task / threat field is its own array (SoA), so a wave reading "all positions" or "all statuses"
// touches consecutive LDS banks.  All geometry is f64 (1e-5 absolute on coordinates up to 1200 rules
// out f32), ids/states are i32, masks are bits.
//
// Reference objects restated here: UAV / Task / Threat (mUAV_TA/DroneEnvComponents.py:7-52,223-263,
// 331-350) and the MultiUAVEnv counters (mUAV_TA/DroneEnv.py:218-243,550-576).
#pragma once
#include <stdint.h>

#include "../../include/muavta.h"


// Per-env error codes stored in EnvState::error (first error wins).
enum {
  MUAVTA_ERR_NONE = 0,
  MUAVTA_ERR_TASK_SLOTS = 1,   // no free task slot (raise tile_tasks)
  MUAVTA_ERR_QUEUE = 2,        // agent queue deeper than Tile::Q
  MUAVTA_ERR_EVENTS = 3,       // event list overflow
  MUAVTA_ERR_PENDING = 4,      // pending-reveal list overflow
  MUAVTA_ERR_POSITION = 5,     // random_position failed 100 times (reference raises ValueError)
  MUAVTA_ERR_ESCORTS = 6,
  MUAVTA_ERR_LSAP = 7,
};

// Task flag bits.
enum { TF_ESCORT = 1, TF_COUNTED = 2, TF_REACHED = 4, TF_ELIGIBLE = 8, TF_DEADLINE = 16,
       TF_KNOWN_ALL = 32 /* the team-wide reveal happened: every agent's known bit is set, sensing has nothing to add */ };

// Kernel-argument block: config + host-derived constants (all computed with the reference's
// operation order on the host so device and oracle agree bit for bit).
struct DevParams {
  int32_t n_agents, n_tasks /* DroneEnv.py:145 */, max_tasks /* :147 */, n_threats;
  int32_t n_agent_groups, agent_type[MUAVTA_MAX_GROUPS], agent_count[MUAVTA_MAX_GROUPS];
  int32_t n_task_groups, task_type[MUAVTA_MAX_GROUPS], task_count[MUAVTA_MAX_GROUPS];
  int32_t n_threat_groups, threat_type[MUAVTA_MAX_GROUPS], threat_count[MUAVTA_MAX_GROUPS];
  int32_t max_time_steps, multiple_tasks_per_agent, early_terminate, capability_mask, saturate_mask,
      include_time_windows, threat_delay, hard_windows, window_length, burst_mode, burst_size,
      dual_region_bursts, share_knowledge, escort_enabled, num_obstacles, random_init_pos,
      escort_required_agents /* max(2, ceil(escort_requirement)), DroneEnv.py:1907 */, commit_horizon;
  uint32_t escort_mask;
  double speed[7];          // maxSpeeds[type] / frame_rate * 0.02            (DroneEnv.py:611,725)
  double threat_prob;       // 0.7 / frame_rate * 0.02                         (:162)
  double reward_norm_factor;// (possible + possible) / 1000                    (:670-675)
  double sense_sq_bound;    // largest v with sqrt(v) <= sense_radius (correctly rounded): d <= r  <=>  d*d-form <= bound
  double fail_rate, arrival_rate, dynamic_idle_penalty, sense_radius, miss_penalty, on_time_bonus,
      reassign_penalty, escort_radius, escort_requirement, escort_intercept_radius, mutual_support_radius;
  double rw[8];
  double escort_sq_bound;   // the same threshold for escort_radius (coverage test of _sync_escorts)
  double inv_mts, inv_max_tasks;  // RN(1 / max(max_time_steps, 1)), RN(1 / max(max_tasks, 1)): Sim::div_small
  int32_t slot_cap;               // live task slots an env may use: MuavtaParams.tile_tasks when it is below the tile's slot count, else the tile's
  int32_t rw_plain;               // every reward weight is finite and >= 0 and reward_norm_factor > 0: a step whose reward terms are all +0.0 has reward +0.0
};

template <int A_, int T_, int H_, int R_, int E_, int Q_, bool REGC_ = false, bool OTFC_ = false, bool SLIM_ = false>
struct Tile {
  static constexpr bool SLIM = SLIM_;  // next_free_* / orgReqs / doneReqs / mission areas live in the HBM record instead of LDS (the 24-agent tile: 16 envs per CU need <= 10 KiB)
  static constexpr bool REGC = REGC_;  // the allocator builds the LSAP cost columns in registers (while they fit one per lane): no A x T cost tile in LDS
  static constexpr bool OTFC = OTFC_;  // beyond 64 columns the (LDS) solver evaluates cost elements on the fly: no A x T cost tile either
  static constexpr bool NO_COST_TILE = REGC_ || OTFC_;
  static constexpr int A = A_;  // agents
  static constexpr int T = T_;  // live task slots
  static constexpr int H = H_;  // threats
  static constexpr int R = R_;  // pending reveals
  static constexpr int E = E_;  // events per list
  static constexpr int Q = Q_;  // agent queue depth (reference max measured on the BASELINE configs: 7 / 10 / 5 on the three tiles; the 30- and 64-agent attention scenarios go past 8)
  static constexpr int KW = (T_ + 31) / 32;  // known-mask words per agent
  // waves per SIMD the fused kernels are compiled for (the VGPR budget is 512 / this).  16 envs of the small tiles share a
  // CU: four single-wave workgroups per SIMD, 128 VGPRs.  The 64-agent tile is bound by LDS to 4-5 envs per CU — at most two
  // waves on a SIMD — so its kernels get 256 VGPRs and keep a whole 64-row LSAP cost column per lane in registers.
  static constexpr int MIN_WAVES = A_ > 32 ? 2 : 4;
  static_assert(A_ <= 64 && T_ <= 128 && H_ <= 64, "ids of agents / slots / threats are stored in int8");
};

template <int A> struct KnowMask { typedef uint64_t type; };  // one bit per agent
template <> struct KnowMask<16> { typedef uint16_t type; };
template <> struct KnowMask<24> { typedef uint32_t type; };
template <int A> struct BucketMask { typedef unsigned long long type; };  // one bit per agent, set by LDS atomics (>= 32 bits)
template <> struct BucketMask<16> { typedef uint32_t type; };
template <> struct BucketMask<24> { typedef uint32_t type; };

// Narrow storage types: every index below is bounded by the tile (agents < 64, slots < 128, threats < 64), task ids and
// time steps by muavta_create's argument checks (max_time_steps <= 20000; ids grow by a few per step).  The device code
// reads them as int: LDS sub-dword loads cost the same as dword loads and the blob shrinks by ~3.5 KB on the 16-agent
// tile — residency on the CU is bound by LDS bytes per env.
typedef int8_t i8;
typedef int16_t i16;
typedef uint8_t u8;

// Fields that change with the queues / on task completion only: in LDS on the tiles that have room (their readers sit on
// the serial paths of a re-plan step), in the HBM record on the SLIM tile.
template <int A, int T, bool HERE> struct QueueSide {
  double a_nfx[A], a_nfy[A], a_nft[A];  // next_free_position, next_free_time
  double t_org[T], t_done[T];           // orgReqs[typeIdx], doneReqs[typeIdx]
  double area[3][3];                    // mission areas: top-left x, y, width (height == width)
};
template <int A, int T> struct QueueSide<A, T, false> {};

// Part of an env that lives in HBM only (one record per env, L2-resident while the rollout runs): the six-component
// requirement vectors of the tasks, the per-queue-entry travel times and the derived init/done times — touched when an
// allocation changes, when a task is created or concluded, and by the observation / token writers (all lanes, coalesced
// within a row), never by the per-step movement / sensing / threat phases.  Keeping them out of LDS is what lets 16 envs
// of the 16-agent tile share one CU's 160 KiB.
template <class TL>
struct alignas(16) EnvCold : QueueSide<TL::A, TL::T, TL::SLIM> {
  enum { A = TL::A, T = TL::T, Q = TL::Q };
  double t_cur[6][T], t_alloc[6][T];  // currentReqs, allocatedReqs
  double t_init[T], t_dtime[T];       // initTime, doneTime
  double a_qtime[A][Q];               // allocationDetails[agent][1] (time_to_task) of each queued task
  double obst[8][3];                  // obstacles (x, y, size): only with num_obstacles > 0
};

template <class TL>
struct alignas(16) EnvState : QueueSide<TL::A, TL::T, !TL::SLIM> {
  enum { A = TL::A, T = TL::T, H = TL::H, R = TL::R, E = TL::E, Q = TL::Q, KW = TL::KW };
  // ---- 8-byte arrays --------------------------------------------------------------------------
  double a_px[A], a_py[A];          // position
  double a_dist[A];                 // env.agent_distances
  double a_caps[6][A];              // currentCap2Task
  double t_px[T], t_py[T];
  double h_px[H], h_py[H];
  // ---- 4-byte arrays --------------------------------------------------------------------------
  typename BucketMask<A>::type t_bucket[T];  // allocation_table[id] as agent bitmask
  uint32_t known[A][KW];            // agent_known_tasks as slot bitmask (bits of free slots are kept clear)
  uint32_t free_slots[KW];          // bit s set <=> slot s is free
  // ---- 2-byte arrays --------------------------------------------------------------------------
  i16 act_index[A];                 // staged actions: index into last_tasks_info as given (clamped to 16 bits when out of range)
  i16 a_qid[A][Q];                  // queued task ids (head first); qlen == 0 <=> [task_idle]
  i16 a_task_start[A], a_fail[A], a_last_id[A], a_commit[A];
  i16 a_gone[A];                    // ids in agent_known_tasks[a] whose slot was released: len(known) = popcount(known[a]) + a_gone[a]
  i16 t_id[T];                      // -1 = free slot
  i16 t_created[T], t_deadline[T], t_prot_id[T];
  i16 h_task_id[H], h_tdeadline[H];
  i16 ev_arg[E], dev_arg[E];
  i16 pend_time[R], pend_id[R];
  i16 esc_id[A], esc_pid[A];
  typename KnowMask<A>::type pend_know[R];  // who knew the task when its slot was released before the reveal came due
  // ---- 1-byte arrays --------------------------------------------------------------------------
  i8 a_qslot[A][Q];                 // slot hint for a_qid (valid iff t_id[slot] == id)
  i8 a_qlen[A];
  i8 a_state[A], a_acap[A], a_type[A], a_name[A], a_reeval[A], a_last_slot[A];
  i8 t_status[T], t_type[T], t_required[T];
  u8 t_flags[T], t_elig[T];
  i8 t_threat[T];                   // relative_threat (threat id) or -1
  i8 t_prot_agent[T], t_prot_slot[T];  // escort: protected agent / Rec task
  i8 t_ndet[T];                     // len(allocationDetails)
  u8 t_order[T];                    // live slots in ascending id (== creation) order
  u8 open_slot[T];                  // env.last_tasks_info (slots), status != 2 at last observation
  u8 t_row[T];                      // inverse of open_slot: row of a slot in last_tasks_info
  i8 h_status[H];                   // -9 = still waiting in its group
  i8 h_target[H], h_mission[H], h_intercept[H];
  i8 h_task_slot[H], h_det_slot[H], h_acap[H], h_type[H], h_group[H];
  u8 h_tflags[H];                   // copy of the Int task's TF_DEADLINE|TF_COUNTED once its slot is freed
  u8 h_order[H];                    // env.threats (spawn order)
  i8 g_next[MUAVTA_MAX_GROUPS], g_end[MUAVTA_MAX_GROUPS];  // threats_groups as [next, end) id ranges
  u8 ev_tag[E];                     // env.event_list (generated this step)
  u8 dev_tag[E];                    // infos['events'] (drained at the start of the last step)
  u8 pend_slot[R];
  i8 esc_agent[A], esc_slot[A], esc_pslot[A];  // _escort_by_recon in insertion order (entries of escorts that expired by
                                               // window are never popped, as in the reference); esc_pid/pslot: protected Rec task
  i8 act_agent[A], act_slot[A];     // actions staged by the allocator
  // ---- scalars (everything above may double as scratch while a reset sets up the RNG) -----------
  unsigned long long esc_mask;      // bit a set <=> recon UAV a has an entry in _escort_by_recon (the esc_* lists above)
  double F_Reward, total_distance, last_reward, step_reward;
  double r_time_penalty, r_alloc;   // reward terms evaluated mid-step (DroneEnv.py:1140-1145), before the world dynamics
  int32_t time_steps, conclusion_time, n_order, n_open, n_active_threats, n_events, n_dev, n_pending, n_escorts, n_act;
  int32_t next_task_id, n_reached, n_retired_empty_buckets;
  int32_t n_reallocations, n_task_switches, n_arrivals, n_missed_windows, n_on_time, n_windowed_tasks,
      idle_reserve_steps, burst_region_toggle, escort_requests, escort_completed, escort_failed,
      escort_required_steps, escort_covered_steps, protection_breaches, threats_intercepted, recon_losses,
      escort_losses, mutual_support_engagements, protected_rec_completed;
  int32_t pending_reset, terminated, truncated, error, did_reset;
  int32_t last_plan_step, n_replans, n_calls;  // HungarianAllocator state
  int32_t gate_step;                           // time step + 1 at which the replan gate last fired (0 = never)
  uint32_t rng_idx[4];                // cursor into each stream's 2x624-word tape (agent, obs, tgt, mission)
  uint32_t rng_win[4][8];             // the next 8 raw words of each stream, prefetched at the step boundary
  uint32_t rng_win_at[4];             // cursor value the window was filled at
  int32_t times_dirty;                // some allocationDetails changed since initTime/doneTime were rebuilt
  int32_t obs_rows;                   // rows [obs_rows, max_tasks) of the HANDLE's observation task tensor are known to hold pad rows
                                      // ({"status": -1}); -1: unknown (after a reset, or when the host rewrote the state)
  int32_t list_stale;                 // an out-of-step call (muavta_call) created tasks since `open_slot` (last_tasks_info) was built: the
                                      // allocator's list — the harness reads env.tasks when it plans — is that list plus the newer tasks
                                      // (fits the tail padding of every tile: the record does not grow)
};

// Standard tiles (BASELINE.json configs).  The 16-agent tile has 40 task slots — the bound at which the reference stops
// creating arrivals for a 16-UAV config (DroneEnv.py:145-147,1646-1689: len(tasks) >= max_tasks - 1) — and queues of 10:
// with 32 slots / queue 8 about 1 in 10,000 seeds of WPS_hard_x2 overflowed (34 slots); events <= 16, pending reveals <= 22
// measured over 4096 seeds.
typedef Tile<16, 40, 16, 48, 40, 10, true> Tile16;
#ifndef MUAVTA_TILE24_SLIM  // experiment knob (0: QueueSide in LDS like the other tiles: 11.6 KB per env, 14 envs per CU)
#define MUAVTA_TILE24_SLIM 1
#endif
typedef Tile<24, 48, 24, 88, 32, 12, true, false, MUAVTA_TILE24_SLIM != 0> Tile24;  // escort 24-UAV config: queue <= 10, pending <= 60, events <= 20 measured
typedef Tile<64, 128, 48, 128, 96, 12, true, true> Tile64;  // register cost columns up to 64 LSAP columns, on-the-fly LDS solver beyond

#define MUAVTA_REL_ROW 29  // doubles per release-log row (include/muavta.h: MUAVTA_F_RELEASE_LOG)
#define MUAVTA_RNG_STREAMS 4
#define MUAVTA_RNG_WORDS 1248  // two consecutive MT19937 blocks per stream
