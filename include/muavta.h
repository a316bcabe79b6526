/*
 * muavta.h — C ABI of the MI355X-native batched mUAV_TA environment.
 *
 * Drop-in boundary for ONE path of andrekuros/Multi-UAV-TA-gym-env: batched reset/step of
 * mUAV_TA.MultiUAVEnv plus the Local-/Coalition-Hungarian allocator that drives it in the WPS and
 * escort evaluation loops.  The reference has no FFI for this path (it is a Python class plus one
 * PyO3 helper), so every entry point cites the reference interface it replaces; the ctypes binding a
 * maintainer would add is shown in INTEGRATION.md.  All functions return 0 on success and a negative
 * MUAVTA_E_* code on failure; muavta_last_error() gives the message.  Plain pointers and sizes only.
 *
 * Host buffers passed in are copied before the call returns; `dst` buffers are host memory filled
 * synchronously.  One handle owns one HIP stream on one device; calls on a handle must be
 * serialized by the caller (the reference env is single-threaded and non-reentrant too).
 */
#ifndef MUAVTA_H
#define MUAVTA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MUAVTA_ABI_VERSION 1

/* UAV / threat type indices = mUAV_TA/MultiDroneEnvData.py:15 (UavTypes order). */
enum { MUAVTA_R1 = 0, MUAVTA_R2 = 1, MUAVTA_E1 = 2, MUAVTA_F1 = 3, MUAVTA_F2 = 4, MUAVTA_T1 = 5, MUAVTA_T2 = 6 };
/* Task type indices = mUAV_TA/MultiDroneEnvData.py:18 (TaskTypes order). */
enum { MUAVTA_HOLD = 0, MUAVTA_REC = 1, MUAVTA_ATT = 2, MUAVTA_DEF = 3, MUAVTA_INT = 4, MUAVTA_DET = 5 };
/* Event tags pushed on MultiUAVEnv.event_list (mUAV_TA/DroneEnv.py:976-977,1639-1640,1914-1915,1950). */
enum { MUAVTA_EV_RESET_ALLOCATION = 0, MUAVTA_EV_NEW_THREAT = 1, MUAVTA_EV_AGENT_FAIL = 2,
       MUAVTA_EV_ESCORT_CREATED = 3, MUAVTA_EV_ESCORT_RETIRED = 4 };

enum {
  MUAVTA_OK = 0,
  MUAVTA_E_ARG = -1,        /* bad argument / unsupported configuration */
  MUAVTA_E_NO_DEVICE = -2,  /* no HIP device: this library has no CPU fallback */
  MUAVTA_E_HIP = -3,        /* HIP runtime error */
  MUAVTA_E_CAPACITY = -4,   /* an env overflowed a tile (task slots / queue / events / pending) */
  MUAVTA_E_STATE = -5       /* call order violated (e.g. step before reset) */
};

#define MUAVTA_MAX_GROUPS 8
#define MUAVTA_N_METRICS 30
#define MUAVTA_N_REWARD_WEIGHTS 8

/*
 * Configuration: mUAV_TA/MultiDroneEnvUtils.py:5-105 (agentEnvOptions) as read by
 * MultiUAVEnv.__init__ (mUAV_TA/DroneEnv.py:73-243), already coerced the way __init__ coerces it
 * (`x or default`).  Dict-valued options keep their insertion ORDER, which the reference's reset
 * depends on (agent naming/creation order DroneEnv.py:124-127,603-612; task creation :641-667).
 */
typedef struct MuavtaParams {
  int32_t abi_version;                       /* MUAVTA_ABI_VERSION */
  int32_t n_agent_groups;                    /* config.agents: (type, count) in dict order */
  int32_t agent_type[MUAVTA_MAX_GROUPS];
  int32_t agent_count[MUAVTA_MAX_GROUPS];
  int32_t n_task_groups;                     /* config.tasks: (type, count) in dict order */
  int32_t task_type[MUAVTA_MAX_GROUPS];
  int32_t task_count[MUAVTA_MAX_GROUPS];
  int32_t n_threat_groups;                   /* config.threats_list: (type, count) */
  int32_t threat_type[MUAVTA_MAX_GROUPS];
  int32_t threat_count[MUAVTA_MAX_GROUPS];
  int32_t max_time_steps;
  int32_t multiple_tasks_per_agent;
  int32_t early_terminate;
  int32_t capability_mask;
  int32_t saturate_mask;
  int32_t include_time_windows;
  int32_t threat_delay;
  int32_t hard_windows;
  int32_t window_length;
  int32_t burst_mode;
  int32_t burst_size;
  int32_t dual_region_bursts;
  int32_t share_knowledge;
  int32_t commit_horizon;
  int32_t escort_enabled;
  uint32_t escort_agent_type_mask;           /* bit t set <=> UAV type t in escort_agent_types */
  int32_t num_obstacles;                     /* <= 8; drawn per env at reset from the rndObsGen stream (DroneEnv.py:579-583) */
  double simulation_frame_rate;
  double fail_rate;
  double reward_weights[MUAVTA_N_REWARD_WEIGHTS]; /* action, distance, quality, s_quality, time, alloc, time_penaulty, step */
  double arrival_rate;
  double dynamic_idle_penalty;
  double sense_radius;
  double miss_penalty;
  double on_time_bonus;
  double reassign_penalty;
  double escort_radius;
  double escort_requirement;
  double escort_intercept_radius;
  double mutual_support_radius;
  /* Tile (capacity) of one env instance in device memory; 0 = derive a default. */
  int32_t tile_agents;   /* >= n_agents; 16 / 24 / 64 in BASELINE configs */
  int32_t tile_tasks;    /* live task slots (open + retired-but-referenced); 40 / 48 / 128.  The library picks the smallest built tile
                            that holds (tile_agents, tile_tasks, tile_threats): any request up to 40 gets the 40-slot tile (MuavtaDims.tile_tasks
                            says what was chosen).  (Round 4 briefly read a value below 40 as a cap on the live slots; that test hook is
                            muavta_set_slot_cap now and this field means what it always meant.) */
  int32_t tile_threats;  /* >= sum(threat_count) */
  int32_t random_init_pos; /* config.random_init_pos (DroneEnv.py:607) */
} MuavtaParams;

/* Geometry the caller needs to size its buffers. */
typedef struct MuavtaDims {
  int32_t n_envs, n_agents, tile_agents, tile_tasks, tile_threats;
  int32_t max_tasks;      /* observation pad length = n_tasks + 28 (DroneEnv.py:145-147) */
  int32_t obs_task_width; /* 21 features per task (leading dim of the feature-major tasks tensor) */
  int32_t obs_agent_width;/* 9 floats per agent row */
  int32_t queue_cap, event_cap, action_cap;
  int64_t state_bytes;    /* bytes of one env's device state: its LDS-image record + its HBM-only record */
  int32_t n_threats;      /* sum(threat_count): leading dim of the THREAT_* fields */
  int32_t known_words;    /* ceil(tile_tasks / 32) */
  int32_t lds_bytes;      /* LDS one workgroup (= one env) occupies */
  int32_t legal_words;    /* ceil(max_tasks / 64): u64 words per agent row of the legal mask */
} MuavtaDims;

/* State fields readable with muavta_get (row-major, leading dim n_envs; A = n_agents, T = tile_tasks,
 * H = n_threats, Q = queue_cap, E = event_cap).  Task fields are indexed by SLOT: the device keeps
 * only live tasks (open, or retired but still queued by a live agent) in T recycled slots; TASK_ID
 * maps slot -> Task.id (-1 = free) and ids stay monotone as in the reference (DroneEnv.py:325-328). */
typedef enum MuavtaField {
  MUAVTA_F_AGENT_POS = 0,      /* f64 [N, A, 2]      UAV.position                            (rw) */
  MUAVTA_F_AGENT_STATE,        /* i32 [N, A]         UAV.state                               (rw) */
  MUAVTA_F_AGENT_HEAD,         /* i32 [N, A]         UAV.tasks[0].id                              */
  MUAVTA_F_AGENT_QUEUE,        /* i32 [N, A, Q]      UAV.tasks ids ([0] == [task_idle]), -1 padded */
  MUAVTA_F_AGENT_NFT,          /* f64 [N, A]         UAV.next_free_time                      (rw) */
  MUAVTA_F_AGENT_NFP,          /* f64 [N, A, 2]      UAV.next_free_position                  (rw) */
  MUAVTA_F_AGENT_CAPS,         /* f64 [N, A, 6]      UAV.currentCap2Task                     (rw) */
  MUAVTA_F_AGENT_ATTACK_CAP,   /* i32 [N, A]         UAV.attackCap                           (rw) */
  MUAVTA_F_AGENT_TYPE,         /* i32 [N, A]         UAV.typeIdx                                  */
  MUAVTA_F_AGENT_NAME_IDX,     /* i32 [N, A]         index of UAV.name in possible_agents         */
  MUAVTA_F_AGENT_DIST,         /* f64 [N, A]         env.agent_distances                          */
  MUAVTA_F_AGENT_MISC,         /* i32 [N, A, 6]      task_start, fail_event, re_eval, last_task id (-1 None), commit_until, len(tasks) (rw except last) */
  MUAVTA_F_TASK_ID,            /* i32 [N, T]         Task.id of the slot (-1 = free)              */
  MUAVTA_F_TASK_STATUS,        /* i32 [N, T]                                                 (rw) */
  MUAVTA_F_TASK_POS,           /* f64 [N, T, 2]                                              (rw) */
  MUAVTA_F_TASK_CUR,           /* f64 [N, T, 6]      Task.currentReqs                        (rw) */
  MUAVTA_F_TASK_ALLOC,         /* f64 [N, T, 6]      Task.allocatedReqs                      (rw) */
  MUAVTA_F_TASK_ORG_DONE,      /* f64 [N, T, 2]      orgReqs[typeIdx], doneReqs[typeIdx]     (rw) */
  MUAVTA_F_TASK_META,          /* i32 [N, T, 8]      type, hard_deadline(-1 none), created_at, required_agents (w), escort, len(allocationDetails), protected_agent(-1), eligible type mask(-1 none) */
  MUAVTA_F_TASK_TIMES,         /* f64 [N, T, 2]      initTime, doneTime                      (rw) */
  MUAVTA_F_KNOWN,              /* u32 [N, A, ceil(T/32)]  bit s of row a: agent a knows the task in slot s (rw) */
  MUAVTA_F_THREAT_POS,         /* f64 [N, H, 2]                                              (rw) */
  MUAVTA_F_THREAT_META,        /* i32 [N, H, 8]      status(-9 not spawned), target agent, mission target, attackCap, task id, type, group, intercepting agent */
  MUAVTA_F_SCALARS,            /* f64 [N, 28]        see MUAVTA_S_* below                         */
  MUAVTA_F_OPEN_IDS,           /* i32 [N, T]         env.last_tasks_info ids in order, -1 padded  */
  MUAVTA_F_EVENTS,             /* i32 [N, E, 2]      events drained by the last step (infos['events']), tag -1 padded */
  MUAVTA_F_EVENT_LIST,         /* i32 [N, E, 2]      env.event_list (generated by the last step, not yet drained)     */
  MUAVTA_F_STAGED_ACTIONS,     /* i32 [N, tile_agents, 3]  (agent, task id, open-list index) left by muavta_allocate, -1 padded */
  MUAVTA_F_ERROR,              /* i32 [N]            0 or the tile that overflowed                */
  MUAVTA_F_RELEASE_LOG,        /* f64 [N, 1 + 29*T]  after muavta_set_release_log(h, 1): first 4 bytes of [0] = i32 number of task slots the
                                                       last step released; then 29-double rows: id, knower mask lo32, hi32 (agents
                                                       that had the id in agent_known_tasks), type, hard_deadline(-1), created_at,
                                                       required_agents, escort, len(allocationDetails), protected_agent(-1),
                                                       eligible mask(-1), x, y, orgReqs[type], doneReqs[type], initTime, doneTime,
                                                       currentReqs[6], allocatedReqs[6]                              (read-only) */
  MUAVTA_F_KNOWN_COUNT,        /* i32 [N, A]         len(agent_known_tasks[a]) incl. ids of released tasks (read-only)  */
  MUAVTA_F_ESCORTS,            /* i32 [N, tile_agents, 2]  env._escort_by_recon in insertion order: (recon UAV.id, escort Task.id), -1 padded (read-only) */
  MUAVTA_F_COUNT_
} MuavtaField;

/* Columns of MUAVTA_F_SCALARS. */
enum {
  MUAVTA_S_TIME_STEPS = 0, MUAVTA_S_REWARD, MUAVTA_S_F_REWARD, MUAVTA_S_TOTAL_DISTANCE, MUAVTA_S_N_ON_TIME,
  MUAVTA_S_N_MISSED, MUAVTA_S_N_WINDOWED, MUAVTA_S_N_SWITCHES, MUAVTA_S_N_REALLOC, MUAVTA_S_N_ARRIVALS,
  MUAVTA_S_IDLE_RESERVE, MUAVTA_S_CONCLUSION_TIME, MUAVTA_S_ESCORT_REQUESTS, MUAVTA_S_ESCORT_COMPLETED,
  MUAVTA_S_ESCORT_FAILED, MUAVTA_S_ESCORT_REQUIRED_STEPS, MUAVTA_S_ESCORT_COVERED_STEPS,
  MUAVTA_S_PROTECTION_BREACHES, MUAVTA_S_THREATS_INTERCEPTED, MUAVTA_S_RECON_LOSSES, MUAVTA_S_ESCORT_LOSSES,
  MUAVTA_S_MUTUAL_SUPPORT, MUAVTA_S_PROTECTED_REC, MUAVTA_S_N_REPLANS, MUAVTA_S_PENDING_RESET, MUAVTA_S_N_REACHED,
  MUAVTA_S_N_PENDING_REVEALS, MUAVTA_S_N_TASKS_CREATED, MUAVTA_N_SCALARS
};

typedef struct MuavtaEnv MuavtaEnv; /* opaque */

/* out[0] = sizeof(MuavtaParams), out[1] = sizeof(MuavtaDims), out[2] = MUAVTA_ABI_VERSION: lets a
 * foreign-language binding verify its struct layout before the first call.  Needs no GPU. */
int muavta_abi_sizes(int32_t* out);

/* MultiUAVEnv(config) for n_envs independent instances on HIP device `device` (DroneEnv.py:73-323).
 * Fails with MUAVTA_E_NO_DEVICE when no GPU is present. */
int muavta_create(const MuavtaParams* params, int32_t n_envs, int32_t device, MuavtaEnv** out);
int muavta_destroy(MuavtaEnv* env);
const char* muavta_last_error(const MuavtaEnv* env); /* env may be NULL: last create() error */
int muavta_dims(const MuavtaEnv* env, MuavtaDims* out);

/* env.reset(seed=seeds[i]) for every instance (DroneEnv.py:522-762). */
int muavta_reset(MuavtaEnv* env, const uint64_t* seeds);

/* env.step(actions) (DroneEnv.py:774-1206).  Actions are the ORDERED (agent, index) items of the
 * reference's actions dict: act_agent[i, k] = UAV.id (position in agents_obj) or -1 to end the
 * list, act_index[i, k] = index into env.last_tasks_info (the open list returned by the previous
 * observation).  Leading dims are [n_envs, action_cap]. */
int muavta_step(MuavtaEnv* env, const int32_t* act_agent, const int32_t* act_index);
/* The same with rows of any length: [n_envs, list_cap] (list_cap 1..32767).  The reference's actions dict takes a LIST of
 * indices per agent (DroneEnv.py:822-825) and applies the items one after the other; rows longer than action_cap are applied
 * action_cap items at a time on the device, in order, inside the one step.  muavta_step(e, a, i) == muavta_step_lists(e, a, i,
 * action_cap). */
int muavta_step_lists(MuavtaEnv* env, const int32_t* act_agent, const int32_t* act_index, int32_t list_cap);

/* HungarianAllocator.allocate_tasks(get_live_agents(), _open_tasks(env), time_steps, events,
 * agent_known_ids=agent_visibility_map()) + _apply_assign  (HungarianAllocator.py:72-208,
 * experiments/wps_eval.py:55-61,123-133).  Writes the actions the harness would pass to step, in
 * muavta_step's layout (host buffers, may be NULL to only stage them on the device for
 * muavta_step_staged).  use_visibility=0 gives Global-Hungarian.  The allocator state
 * (last_plan_step, n_replans) lives in the handle and is cleared by muavta_reset. */
int muavta_allocate(MuavtaEnv* env, int32_t replan_interval, int32_t use_visibility,
                    int32_t* act_agent, int32_t* act_index);
int muavta_step_staged(MuavtaEnv* env); /* step with the actions muavta_allocate left on the device */

/* Which allocator muavta_allocate / muavta_rollout run (default MUAVTA_ALLOC_HUNGARIAN):
 *   MUAVTA_ALLOC_URGENCY_PAIR = UrgencyPair.plan(env, hung, events, force=True) under the harness gate
 *   _should_replan(env, events, 15): engineered float32 edge scores 0.5*urgency + 0.3*scarcity - 0.4*dist, clipped
 *   to +-0.35, on the first 16 live agents x first 32 underfilled tasks, subtracted from the Hungarian cost
 *   (TaskAllocation/Hybrid/PairCostHybrid.py:31-86,520-550; experiments/wps_eval.py:64-74,248-254).
 *   `replan_interval` is ignored in that mode.
 *   MUAVTA_ALLOC_URGENCY_COALITION = UrgencyCoalition.plan(env, hung_force, events, force=True) under
 *   escort_eval._should_replan(env, events, replan_interval): f64 edge scores clip(0.45*urgency + 0.35*threat
 *   pressure*(0.5+0.5*is_escort) + 0.3*min(cap,1) - 0.25*dist [+0.2 role bonus], 0, 1) for every live agent x open
 *   task, agents with commit_until > t held out of the match, assigned agents that hold a real task locked for
 *   commit_horizon steps (TaskAllocation/Hybrid/AttentionEscort.py:32-66,714-767; AttentionCommit.py:24-44;
 *   experiments/escort_eval.py:52-58,175-180). */
/*   MUAVTA_ALLOC_HUNGARIAN_GATED = the trainers' expert / teacher: HungarianAllocator.allocate_tasks(force=True) under
 *   _should_replan(env, events, replan_interval) with tags Reset_Allocation, New_Threat, Agent_Fail
 *   (experiments/train_pair_cost.py:33-43,109-118); use_visibility=0 gives the Global-Hungarian expert. */
enum { MUAVTA_ALLOC_HUNGARIAN = 0, MUAVTA_ALLOC_URGENCY_PAIR = 1, MUAVTA_ALLOC_URGENCY_COALITION = 2, MUAVTA_ALLOC_HUNGARIAN_GATED = 3 };
int muavta_set_allocator(MuavtaEnv* env, int32_t mode);

/* Token builders of the learned/engineered hybrids, batched over all envs straight from the device state
 * (SURVEY §8f rank 2).  kind:
 *   MUAVTA_TOK_PAIR      build_pair_tokens(env, max_tasks, max_agents)            task 13, agent 12 features
 *   MUAVTA_TOK_PAIR_RAW  build_pair_tokens(env, max_tasks, max_agents, raw=True)  task  9, agent 11
 *       (= build_att_tokens + edge_valid; TaskAllocation/Hybrid/AttentionRAH.py:50-173, PairCostHybrid.py:31-65)
 *   MUAVTA_TOK_ESCORT    build_escort_tokens(env, max_tasks, max_agents)          task 22, agent 16
 *       (TaskAllocation/Hybrid/AttentionEscort.py:76-243; rows sorted by _task_priority_key, :69-74)
 * Outputs, reference layout and dtype, one block per env: task_feats f32 [N, max_tasks, Dt]; task_mask u8 [N, max_tasks]
 * (1 = padding); task_ids i32 [N, max_tasks] (-1 = padding; tok["task_ids"]); agent_feats f32 [N, max_agents, Da];
 * agent_mask u8 [N, max_agents]; agent_ids i32 [N, max_agents] (UAV.id of tok["live"][i], -1 = padding); edge_valid f32
 * [N, max_agents, max_tasks]; n_urgent i32 [N] (tok["n_urgent"], 0 for MUAVTA_TOK_ESCORT; may be NULL).
 * Imitation-learning labels (experiments/train_pair_cost.py:54-71,96-129), both may be NULL: expert_mask f32 [N, max_agents,
 * max_tasks] = _expert_mask(tok, pairs) of the plan the last muavta_allocate staged (1 where the plan pairs agent row i
 * with task column j on a valid edge); replanned i32 [N] = that allocate's replan gate fired at the current time step (the trainer builds a sample then).
 * muavta_tokens copies into host buffers (NULL = skip that output) and synchronises; muavta_tokens_device writes to
 * device buffers of the caller (e.g. torch tensors on the same GPU) on the handle's stream without synchronising. */
/* The reference's mutators that callers invoke on the env objects directly, outside step() — experiments/test_escort.py
 * :61-75,95-96,107,236 and the allocators' scaffolding — for ONE env of the batch (env_index).  iargs[8], darg and the
 * meaning of out[MUAVTA_CALL_OUT] depend on `op`; ids are UAV.id / Task.id as everywhere in this ABI:
 *   MUAVTA_OP_UAV_ALLOCATE        UAV.allocate(task, time_step) (DroneEnvComponents.py:55-95): iargs = {agent, task id, time_step};
 *                                 out[0] = its return value (1 = queued now, 0 = already queued / task concluded)
 *   MUAVTA_OP_CREATE_ESCORT       MultiUAVEnv._create_escort_for(recon, rec_task) (DroneEnv.py:1888-1917): iargs = {recon agent, Rec
 *                                 task id, 0 = None: protected_task stays None}; out[0] = id of the (new or existing) escort task,
 *                                 -1 = None (escorts disabled)
 *   MUAVTA_OP_SYNC_ESCORTS        MultiUAVEnv._sync_escorts() (:1964-2000)
 *   MUAVTA_OP_RETIRE_ESCORT       MultiUAVEnv._retire_escort(escort_task, failed) (:1938-1950): iargs = {escort task id, failed}
 *   MUAVTA_OP_ESCORT_FIGHTERS_NEAR  MultiUAVEnv._escort_fighters_near(agent, radius) (:1746-1764): iargs = {agent}, darg = radius
 *                                 (< 0: escort_radius); out[0] = n, out[1..n] = UAV.id nearest first
 *   MUAVTA_OP_ACTION_VALID        MultiUAVEnv._is_task_action_valid(agent, task) (:341-363): iargs = {agent, task id}; out[0] = 0/1
 *   MUAVTA_OP_SET_QUEUE           `agent.tasks = [t0, t1, ...]` (plain list assignment, e.g. test_escort.py:95: no Task bookkeeping):
 *                                 iargs = {agent, n <= 6, id0 .. id5}; [task_idle] (id 0) is the empty queue.  iargs[7] = 1 with
 *                                 n = 0 is UAV.allocate(task_idle) (DroneEnvComponents.py:59-60,85-92): besides `tasks = [idle]` it
 *                                 resets next_free_time = 0, next_free_position = position, re_eval = False, last_task = None.
 *                                 CONTRACT: the reference's assignment leaves Task.allocationDetails alone (an agent stays in the
 *                                 details of a task it no longer queues); the device keeps allocationDetails IN the queues, so the
 *                                 call is exact for what the reference's own callers do with it — scaffolding agents whose queue is
 *                                 [task_idle] (test_escort.py:95) — and drops the old queue's detail entries otherwise
 * Runs on the handle's stream and synchronises.  Returns MUAVTA_E_ARG for ids outside the env. */
typedef enum MuavtaOp {
  MUAVTA_OP_UAV_ALLOCATE = 0, MUAVTA_OP_CREATE_ESCORT, MUAVTA_OP_SYNC_ESCORTS, MUAVTA_OP_RETIRE_ESCORT,
  MUAVTA_OP_ESCORT_FIGHTERS_NEAR, MUAVTA_OP_ACTION_VALID, MUAVTA_OP_SET_QUEUE, MUAVTA_OP_COUNT_
} MuavtaOp;
#define MUAVTA_CALL_OUT 72
int muavta_call(MuavtaEnv* env, int32_t env_index, int32_t op, const int32_t* iargs, double darg, int32_t* out);

/* Multi-GPU: env instances are independent, so each rank owns a contiguous range of global env indices
 * (seed = global index) and the ONLY exchange is the end-of-batch reduction of a small per-rank metric vector.  These entry
 * points run it over RCCL (xGMI between the GPUs of a node) for binders that do not bring torch.distributed:
 *   muavta_comm_uid    rank 0 creates the 128-byte RCCL unique id; the caller ships it to the other ranks out of band
 *   muavta_comm_init   every rank joins (collective: returns when all n_ranks have called it); one communicator per handle
 *   muavta_allreduce_metrics  float partial sums are all-gathered and added in RANK ORDER (the result does not depend on
 *                      the ring order: bit-stable), int64 counters are all-reduced (exact).  nf, nc <= 64.
 * librccl is loaded on first use (the copy already in the process, e.g. PyTorch's, if there is one). */
#define MUAVTA_COMM_UID_BYTES 128
int muavta_comm_uid(uint8_t* uid /* [128] out */);
int muavta_comm_init(MuavtaEnv* env, int32_t rank, int32_t n_ranks, const uint8_t* uid /* [128] */);
int muavta_allreduce_metrics(MuavtaEnv* env, const double* f_partials, int32_t nf, const int64_t* counters, int32_t nc,
                             double* f_total /* [nf] */, int64_t* c_total /* [nc] */);
int muavta_comm_destroy(MuavtaEnv* env);

/* Per-step log of released task slots (off by default; the Python facade turns it on to keep agent_visibility_map()
 * exact for retired ids, DroneEnv.py:1595-1599).  Costs one global atomic per released slot in muavta_step only. */
int muavta_set_release_log(MuavtaEnv* env, int32_t enable);

enum { MUAVTA_TOK_PAIR = 0, MUAVTA_TOK_PAIR_RAW = 1, MUAVTA_TOK_ESCORT = 2 };
int muavta_tokens(MuavtaEnv* env, int32_t kind, int32_t max_tasks, int32_t max_agents, float* task_feats, uint8_t* task_mask,
                  int32_t* task_ids, float* agent_feats, uint8_t* agent_mask, int32_t* agent_ids, float* edge_valid, int32_t* n_urgent,
                  float* expert_mask, int32_t* replanned);
int muavta_tokens_device(MuavtaEnv* env, int32_t kind, int32_t max_tasks, int32_t max_agents, float* task_feats, uint8_t* task_mask,
                         int32_t* task_ids, float* agent_feats, uint8_t* agent_mask, int32_t* agent_ids, float* edge_valid, int32_t* n_urgent,
                         float* expert_mask, int32_t* replanned);

/* The context vector of the ContextPair hybrids, batched: build_context_summary(env, build_pair_tokens(env, max_tasks, ...)) of
 * TaskAllocation/Hybrid/ContextPairHybrid.py:33-78 — kind MUAVTA_TOK_PAIR: f32 [N, 8] = urgent share, min(open / live, 4) / 4, free share, fighter
 * share, left share, right share, |left - right| share of the kept open tasks, mission clock; kind MUAVTA_TOK_PAIR_RAW: f32 [N, 1] = the clock
 * (raw=True).  `max_tasks` is the token pad (the summary runs over the first max_tasks underfilled tasks, as tok["open_tasks"] does).
 * muavta_context fills a HOST buffer and synchronises; muavta_context_device a DEVICE buffer on the handle's stream. */
int muavta_context(MuavtaEnv* env, int32_t kind, int32_t max_tasks, float* context);
int muavta_context_device(MuavtaEnv* env, int32_t kind, int32_t max_tasks, float* context);

/* The learned hybrids' planner call: HungarianAllocator.allocate_tasks(live, tok["open_tasks"], time_step, events, force,
 * task_priorities=..., reserved_agent_names=..., agent_known_ids=..., edge_scores=...) with CALLER-COMPUTED inputs
 * (TaskAllocation/OptimizationBased/HungarianAllocator.py:72-92,123-124,170-179), as PairCostHybrid.plan
 * (TaskAllocation/Hybrid/PairCostHybrid.py:283-294,312-327), AttentionRAH.plan (AttentionRAH.py:395-453),
 * AttentionCommit._plan_from_scores (AttentionCommit.py:266-300) and AttentionEscort._plan_from_scores (AttentionEscort.py:472-515)
 * issue it, followed by _apply_assign (experiments/train_pair_cost.py:46-51).  The inputs are indexed in the token layout the
 * caller's network consumed — muavta_tokens(_device)(kind, max_tasks, max_agents): row i = i-th live agent, column j = j-th
 * token task — so the tensors a policy produces from the token tensors go straight back in:
 *   edge_scores f32 [N, max_agents, max_tasks]  edge_score_dict: cost[i, j] = base - float(score) wherever base < 5e4; pad rows /
 *                                               columns are ignored; NULL = no scores
 *   task_pri    f64 [N, max_tasks]              task_priorities[task_ids[j]] (the -0.4 * priority term of _cost); NULL = none
 *   reserved    u64 [N]                         bit a set <=> UAV.id a is in reserved_agent_names; NULL = none
 * flags: MUAVTA_SC_EDGE_VALID_ONLY  scores count only where the token builder's edge_valid is 1 (PairCostHybrid.edge_score_dict;
 *                                   AttentionEscort's takes every non-pad pair: leave the flag off)
 *        MUAVTA_SC_FULL_TASK_LIST   the allocator gets the token builder's whole open list, not only the max_tasks rows that
 *                                   became tokens (build_att_tokens' "open_tasks": AttentionRAH / AttentionCommit; build_pair_tokens
 *                                   and build_escort_tokens hand over the kept rows: leave the flag off)
 *        MUAVTA_SC_COMMIT           reserved |= committed_names(env); assigned agents that hold a real task are locked for
 *                                   commit_horizon steps (AttentionCommit.py:24-44, as AttentionEscort._plan_from_scores does)
 * gate:  MUAVTA_GATE_FORCE      plan now (the caller evaluated its own gate; force=True)
 *        MUAVTA_GATE_TRAINER    force=True under _should_replan(env, events, replan_interval) with tags Reset_Allocation, New_Threat,
 *                               Agent_Fail (experiments/train_pair_cost.py:33-43; wps_eval.py:64-74 is the same with 15)
 *        MUAVTA_GATE_ESCORT     force=True under escort_eval._should_replan (experiments/escort_eval.py:52-58: any event)
 *        MUAVTA_GATE_ALLOCATOR  force=False: HungarianAllocator.should_replan decides (:27-41)
 * use_visibility = 0 passes agent_known_ids=None.  Outputs: the staged actions (muavta_step_staged applies them; act_agent /
 * act_index as in muavta_allocate, host buffers, may be NULL), selected f32 [N, max_agents, max_tasks] = _selected_mask(tok,
 * result) (PairCostHybrid.py:296-310; may be NULL), replanned i32 [N] = the gate fired (may be NULL).
 * muavta_allocate_scored takes HOST buffers for all of them and synchronises; muavta_allocate_scored_device takes DEVICE buffers
 * (e.g. torch tensors on the handle's GPU), runs on the handle's stream and does not synchronise. */
enum { MUAVTA_GATE_FORCE = 0, MUAVTA_GATE_TRAINER = 1, MUAVTA_GATE_ESCORT = 2, MUAVTA_GATE_ALLOCATOR = 3 };
enum { MUAVTA_SC_EDGE_VALID_ONLY = 1, MUAVTA_SC_FULL_TASK_LIST = 2, MUAVTA_SC_COMMIT = 4 };
typedef struct MuavtaScored {
  int32_t kind, max_tasks, max_agents;  /* MUAVTA_TOK_*, token pads */
  int32_t gate, flags, replan_interval, use_visibility, reserved0;
  const float* edge_scores; const double* task_pri; const uint64_t* reserved;
  float* selected; int32_t* replanned;
} MuavtaScored;
int muavta_allocate_scored(MuavtaEnv* env, const MuavtaScored* spec, int32_t* act_agent, int32_t* act_index);
int muavta_allocate_scored_device(MuavtaEnv* env, const MuavtaScored* spec);

/* One iteration of the RL trainer's loop body for every env in ONE launch (experiments/train_pair_cost.py:139-153, run_rl_episode):
 *   result, tok, ... = policy.plan(env, hung, events, force=True)   <- muavta_allocate_scored_device(plan) with the caller's scores
 *   actions = _apply_assign(env, result); env.step(actions)         <- muavta_step_staged
 *   s_now = env.compute_s_wps(); step_r = (s_now - s_prev) / 20     <- s_wps f64 [2][N]: before / after the step (may be NULL)
 *   next_tok = policy.build_tokens(env)                             <- muavta_tokens_device's outputs (kind and pads of `plan`;
 *                                                                      all seven or none, n_urgent optional)
 *   ep_done                                                         <- done u8 [N] (bit 0 terminated, bit 1 truncated; may be NULL)
 * The caller's network turns the token tensors of call t into the edge scores of call t + 1 on the same GPU: nothing crosses PCIe.
 * write_obs != 0 also refreshes the handle's observation buffers as env.step does.  An env whose episode has ended is left
 * alone (no plan, no step: replanned 0, selected 0, s_wps before == after, tokens of its final state), as the reference's loop
 * ends with the episode.  All pointers are DEVICE pointers; runs on the handle's stream without synchronising. */
typedef struct MuavtaRlStep {
  MuavtaScored plan;
  float* task_feats; uint8_t* task_mask; int32_t* task_ids; float* agent_feats; uint8_t* agent_mask; int32_t* agent_ids;
  float* edge_valid; int32_t* n_urgent;
  double* s_wps; uint8_t* done;
  int32_t write_obs;
  int32_t part;   /* 0 = the whole batch on the handle's stream; p + 1 = sub-batch p (muavta_set_parts) on its own stream: the tensors stay
                     the whole batch's [N, ...] and the launch reads / writes the part's rows — the network runs on part A's tokens
                     while the device steps part B (muavta_wait_part before reading a part's outputs) */
} MuavtaRlStep;
int muavta_rl_step_device(MuavtaEnv* env, const MuavtaRlStep* step);

/* Run to the next replan gate.  The reference's trainer and evaluation loops consult the planner only when their gate fires and step
 * with EMPTY actions otherwise (experiments/train_pair_cost.py:34-43,86-89,139-145; wps_eval.py:64-74,248-254,273):
 *     actions = {}
 *     if _should_replan(env, events): result, tok, ... = policy.plan(env, hung, events, force=True); actions = _apply_assign(env, result)
 *     obs, reward, done, trunc, info = env.step(actions)
 * muavta_rl_run_device is that loop per env inside ONE launch: the FIRST step is muavta_rl_step_device's (`first`: the plan with the caller's
 * scores under `first.plan.gate`, S_WPS before / after, next tokens, done — the transition run_rl_episode pushes, :146-152), then the env
 * keeps stepping with empty actions until ITS OWN gate fires again (the same gate, evaluated on the state after each step: clock +
 * the events the step drained), its episode ends, or max_steps steps have been taken in this launch (0: no bound).  Where it stops it
 * writes the tokens the policy sees next (park_*: muavta_tokens_device's outputs for that state; all seven or none) — so the caller's
 * network runs once per launch on the park tokens and its scores are consumed only by the envs that stopped AT a gate:
 *     n_stepped i32 [N]  env steps this launch took for the env (0: its episode had ended before)
 *     park      u8  [N]  bit 0 terminated, bit 1 truncated, bit 2 = stopped at a gate (the next launch plans for it; clear when the
 *                         episode ended or max_steps cut the quiet run short — the next launch then just continues it, its scores unread)
 *     reward_sum f64 [N] the rewards env.step returned for this launch's steps, added in step order
 * Outputs of `first` for an env that did NOT plan (replanned 0: it was not at a gate, or its episode had ended): selected 0, s_wps / done
 * of its first step, next-token rows untouched.  The launch of a whole batch ends when every env has stopped; envs advance by different
 * numbers of steps (each has its own clock), which is what lets the policy be called once per GATE instead of once per step.
 * first.write_obs != 0 refreshes the handle's observation buffers ONCE per launch, for the state each env stopped in (not after every quiet step).
 * All pointers are DEVICE pointers; asynchronous on the handle's stream (first.part > 0: on that sub-batch's stream, for its rows). */
typedef struct MuavtaRlRun {
  MuavtaRlStep first;
  float* park_task_feats; uint8_t* park_task_mask; int32_t* park_task_ids; float* park_agent_feats; uint8_t* park_agent_mask; int32_t* park_agent_ids;
  float* park_edge_valid; int32_t* park_n_urgent;
  int32_t* n_stepped; uint8_t* park; double* reward_sum;
  int32_t max_steps, reserved;
} MuavtaRlRun;
int muavta_rl_run_device(MuavtaEnv* env, const MuavtaRlRun* run);
/* The same run-ahead for a HOST-side planner (the loop of experiments/wps_eval.py:112-133,248-254,273 with any allocator of the
 * reference behind the facade): env.step(actions) — act_agent / act_index as in muavta_step, or both NULL: the plan muavta_allocate staged —
 * followed by env.step({}) until the env's gate (MUAVTA_GATE_*, replan_interval) fires, its episode ends or max_steps steps were taken
 * (0: no bound).  write_obs != 0 refreshes the handle's observation buffers ONCE, for the state each env stopped in (muavta_observe /
 * muavta_step_result then describe that state's last step).  n_stepped / park / reward_sum as above, HOST buffers (any may be NULL;
 * the call synchronises when one is given). */
int muavta_step_run(MuavtaEnv* env, const int32_t* act_agent, const int32_t* act_index, int32_t gate, int32_t replan_interval, int32_t max_steps,
                    int32_t write_obs, int32_t* n_stepped, uint8_t* park, double* reward_sum);

/* The measured path: reset(seeds) followed by n_steps x (allocate -> step) fused in ONE kernel
 * launch, state resident in LDS (run_wps_episode / run_escort_episode with Local-/Coalition-
 * Hungarian, experiments/wps_eval.py:76-291, experiments/escort_eval.py:86-226).  seeds == NULL
 * continues from the current state without a reset.  write_obs != 0 also writes the batched
 * observation tensors after every step, as env.step does. */
int muavta_rollout(MuavtaEnv* env, const uint64_t* seeds, int32_t n_steps, int32_t replan_interval,
                   int32_t use_visibility, int32_t write_obs);

/* State lanes: episode batches in flight inside ONE handle.  A fused rollout launch lasts as long as its slowest env; the wave slots its early
 * finishers free stay empty until it ends.  The reference's evaluation loop runs episode after episode (experiments/wps_eval.py:528-546: cases x
 * seeds), so the next batch's work exists before this one has finished: a handle may own a SECOND complete set of per-batch device state (env
 * records, RNG tapes, observation buffers, metrics, streams, seeding slots), and a seeded muavta_rollout / muavta_rollout_record that is issued
 * while the handle's previous seeded rollout is still running goes to the other lane — its workgroups start in the free wave slots (+3 % / +17 % /
 * +37 % env-steps/s on BASELINE configs 2 / 4 / 5).  Results are those of the same calls on one lane, bit for bit (env instances are independent).
 *   lanes 0 (default)  the second lane is created the first time a seeded rollout finds the previous one still running (a caller that
 *                      synchronises between rollouts never allocates it);   1  one lane only;   2  create it now, seeded rollouts always alternate
 * After a flip EVERY entry point of the handle refers to the lane of the latest seeded rollout (state, observations, metrics, step calls, ...),
 * exactly as if that rollout had overwritten the previous batch — which is what it does on one lane.  What the second lane adds is reach-back:
 *   muavta_rollout_metrics_back(h, 1, out) / muavta_error_flags_back(h, 1, flags)   the batch BEFORE the latest one (its lane is idle or finishing)
 * (MUAVTA_E_STATE when that batch ran on the SAME lane and has been overwritten: in mode 0 a rollout that finds its predecessor finished does
 * not flip — use mode 2 for a pipeline that relies on reach-back), so `rollout(seeds[i + 1]); read batch i` keeps two batches in flight.  muavta_sync waits for both lanes; muavta_device_ptrs returns the current
 * lane's buffers (they change with every flip: fetch them again after a seeded rollout, or pin the handle to one lane).  Device memory per
 * handle: lanes x n_envs x (MuavtaDims.state_bytes + 19,968 B of RNG tapes + the observation tensors + 240 B of metrics) + 8 MB. */
int muavta_set_lanes(MuavtaEnv* env, int32_t lanes);
int muavta_lanes(const MuavtaEnv* env, int32_t* mode, int32_t* allocated);
int muavta_rollout_metrics_back(MuavtaEnv* env, int32_t back, double* out);
int muavta_error_flags_back(MuavtaEnv* env, int32_t back, int32_t* flags /* [N] */);

/* Test hook for the capacity path (the reference's task list is unbounded, DroneEnv.py:325-328; the tiles are not): an env may use at most
 * `cap` of its tile's task slots (0 = all); one that needs more sets MUAVTA_F_ERROR exactly as on a full tile, which is how
 * BatchedMultiUAVEnv.rollout(escalate=True) and the capped-tile fuzz runs are driven.  Field widths stay the tile's.  Synchronises. */
int muavta_set_slot_cap(MuavtaEnv* env, int32_t cap);

/* Batched observation of the last reset/step (DroneEnv.py:365-415,468-492), feature-major so that
 * the device writes it with contiguous stores (transpose on the host if a row-major view is wanted):
 *   tasks   f32 [N, 21, max_tasks]  rows of the 21 features: id, x, y, status(-1 pad), current_reqs[6],
 *                                   alloc_reqs[6], init_time, end_time, type_idx/6, unmet, age
 *   legal   u64 [N, A, legal_words] legal_mask as bit rows: bit (j & 63) of word (j >> 6) <=> row j legal
 *   pad     u8  [N, max_tasks]      mask
 *   agents  f32 [N, A, 9]           agent_position(2), agent_caps(6), alloc_task
 *   flags   f32 [N, 5]              event_flags
 * Any pointer may be NULL.
 * More than max_tasks open tasks (possible: threat / escort tasks are created past the arrival cap of DroneEnv.py:1646-1689): the
 * reference's lists then simply grow (its pad count goes negative, :410-413); the fixed-width tensors hold the FIRST max_tasks rows.
 * The open list itself (MUAVTA_F_OPEN_IDS, what action indices refer to) is never truncated, and an index >= max_tasks is applied
 * as the reference applies it.  Pinned on reference episodes in that regime by tests/fuzz_reference.py. */
int muavta_observe(MuavtaEnv* env, float* tasks, uint64_t* legal, uint8_t* pad, float* agents, float* flags);

/* rewards / terminations / truncations of the last step: reward f64 [N] (shared by all agents of an
 * env, DroneEnv.py:1162-1178,1202), done u8 [N] (terminated | truncated << 1). */
int muavta_step_result(MuavtaEnv* env, double* reward, uint8_t* done);

/* calculate_metrics() for every env, f64 [N, 30] in the key order of DroneEnv.py:1286-1319. */
int muavta_metrics(MuavtaEnv* env, double* out);

/* Copy one state field of all envs to host / overwrite it from host (tests, object proxies). */
int muavta_get(MuavtaEnv* env, MuavtaField field, void* dst, size_t bytes);
int muavta_set(MuavtaEnv* env, MuavtaField field, const void* src, size_t bytes);

/* Whole-state snapshot (checkpoint/resume): bytes = n_envs * dims.state_bytes, laid out as the n_envs LDS-image records
 * followed by the n_envs HBM-only records (opaque to the caller; same build, same tile). */
int muavta_get_state(MuavtaEnv* env, void* dst, size_t bytes);
int muavta_set_state(MuavtaEnv* env, const void* src, size_t bytes);

/* Stand-alone batched solver with scipy.optimize.linear_sum_assignment's exact tie rules
 * (call site HungarianAllocator.py:181): `n` problems, each cost f64 [nr, nc] row-major (nr, nc <= 64
 * x 128); row/col i64 [n, min(nr, nc)].  Runs on HIP device `device`. */
int muavta_lsap(int32_t device, const double* cost, int32_t n, int32_t nr, int32_t nc, int64_t* row, int64_t* col);
/* Same, with the solver named: the allocator path uses MUAVTA_LSAP_REGISTERS (rows <= 32, columns <= 64 after scipy's
 * transpose; no LDS traffic inside the solve) and MUAVTA_LSAP_LDS beyond that (64 x 128); MUAVTA_LSAP_AUTO picks by size.
 * Both are the same algorithm with the same tie rule; the entry point exists so that tests can pin each one. */
enum { MUAVTA_LSAP_AUTO = 0, MUAVTA_LSAP_LDS = 1, MUAVTA_LSAP_REGISTERS = 2 };
int muavta_lsap_impl(int32_t device, const double* cost, int32_t n_problems, int32_t n_rows, int32_t n_cols, int64_t* row_ind,
                     int64_t* col_ind, int32_t impl);

/* Diagnostic: the kernels' square root and division (sequences restricted to the simulation's operand range, see
 * fsqrt / fdiv in csrc/muavta_device.h) evaluated on the device for n operand pairs: out_sqrt[i] = sqrt(x[i]),
 * out_div[i] = x[i] / y[i], out_div_neg[i] = -x[i] / y[i] (through the shared-reciprocal form).  Inside the documented
 * domain the results equal the IEEE-754 correctly rounded ones bit for bit; the parity tests pin that against numpy. */
int muavta_domain_math(int32_t device, const double* x, const double* y, int32_t n, double* out_sqrt, double* out_div, double* out_div_neg);

/* Diagnostic: the natural logarithm of the obstacle repulsion (core_sim/src/sim_core.rs:44, f64::ln = the host libm's log),
 * evaluated on the device for n positive, finite, normal arguments: libm_log in csrc/muavta_device.h restates the published
 * algorithm of that function (Arm optimized-routines log.c as shipped in glibc) so that device and host agree bit for bit; the
 * parity tests pin that against the host's log. */
int muavta_domain_log(int32_t device, const double* x, int32_t n, double* out);

/* Diagnostic: the two-argument arctangent of the obstacle rule's heading test (core_sim/src/sim_core.rs:46-47, f64::atan2 = the host
 * libm's atan2), evaluated on the device for n (y, x) pairs: libm_atan2 in csrc/muavta_atan2.h restates glibc 2.35's algorithm (which
 * is not correctly rounded) with that library's node table so that device and host agree bit for bit, special operands included;
 * the parity tests pin that against the host's atan2. */
int muavta_domain_atan2(int32_t device, const double* y, const double* x, int32_t n, double* out);

/* core_sim.SimCore.avoid_obstacles (core_sim/src/sim_core.rs:25-59) for n (position, movement)
 * pairs against one obstacle list, evaluated on the device. */
int muavta_avoid_obstacles(int32_t device, const double* agent_pos, const double* movement, int32_t n,
                           const double* obstacles, int32_t n_obstacles, double* out);

/* Device pointers for zero-copy consumers in the same process (torch.from_blob etc.).  The observation buffers are READ-ONLY for
 * the caller: the observation writer leaves rows alone that it knows to hold pad rows already, so a consumer that changes them in
 * place (in-place normalisation, clearing the view) must call muavta_refresh_observation afterwards, which rewrites every row.
 * Read-after-write: `*stream` orders the whole-batch entry points only; with sub-batches active (muavta_set_parts) use muavta_sync /
 * muavta_wait_part before reading. */
int muavta_device_ptrs(MuavtaEnv* env, void** state, void** obs_tasks, void** obs_legal, void** obs_agents,
                       void** metrics, void** stream);

/* The trainers' data loop fused into the rollout (experiments/train_pair_cost.py:96-156: run_il_episode / run_rl_episode):
 * like muavta_rollout, and for every step t = 0 .. n_steps-1, right after the allocator staged its plan and before the env
 * step, slot t of the caller's DEVICE rings receives what muavta_tokens_device would return at that point (token tensors,
 * edge_valid, expert_mask = _expert_mask(tok, plan), replanned = the gate fired at t) and s_wps[t] = compute_s_wps() before the
 * step; s_wps[n_steps] is the value after the last step, so the RL step reward of step t is (s_wps[t+1] - s_wps[t]) / 20.
 * Ring layouts: every muavta_tokens_device output with a leading [n_steps] axis (slot-major: [n_steps][N][...]); s_wps f64
 * [n_steps + 1][N].  n_urgent / expert_mask / replanned may be NULL; kind < 0 leaves the whole token part out (its pointers
 * and s_wps are then ignored).  Use muavta_set_allocator(MUAVTA_ALLOC_HUNGARIAN_GATED) and use_vis = 0 for the
 * reference's expert.
 * Observation rings (optional, all seven or none — obs_tasks == NULL means none; they need write_obs != 0): the observation
 * the env would return from step t (DroneEnv.step -> _get_observations, DroneEnv.py:1226-1243, in muavta_observe's device
 * layouts) goes to slot t of obs_tasks f32 [n_steps][N][21][max_tasks], obs_legal u64 [n_steps][N][A][ceil(max_tasks/64)],
 * obs_pad u8 [n_steps][N][max_tasks], obs_agents f32 [n_steps][N][A][9], obs_flags f32 [n_steps][N][5], obs_reward f64
 * [n_steps][N], obs_done u8 [n_steps][N] (bit 0 terminated, bit 1 truncated) instead of overwriting the handle's single
 * observation buffer each step; the handle's buffer receives the final observation once, at the end.  An env whose episode
 * ended at step t < n_steps - 1 leaves its later slots unwritten: obs_done is pre-filled with MUAVTA_OBS_UNWRITTEN for that.
 * Token rings of such an env (early_terminate, or n_steps beyond max_time_steps): the reference's episode loops stop at `done`
 * (train_pair_cost.py:108,139), so the slots after the last step carry no sample — they are written as all-pad rows (masks 1,
 * ids -1, features / edge_valid / expert_mask 0, n_urgent 0, replanned 0) and s_wps carries the final value forward (the
 * step reward of those slots is exactly 0); no slot of any ring is left uninitialised.
 * Asynchronous on the handle's stream (muavta_sync / an event before reading). */
#define MUAVTA_OBS_UNWRITTEN 0x80
typedef struct MuavtaRecord {
  int32_t kind, max_tasks, max_agents, reserved;  /* MUAVTA_TOK_* or -1, token pads */
  float* task_feats; uint8_t* task_mask; int32_t* task_ids;
  float* agent_feats; uint8_t* agent_mask; int32_t* agent_ids;
  float* edge_valid; int32_t* n_urgent; float* expert_mask; int32_t* replanned;
  double* s_wps;
  float* obs_tasks; uint64_t* obs_legal; uint8_t* obs_pad; float* obs_agents; float* obs_flags; double* obs_reward; uint8_t* obs_done;
} MuavtaRecord;
int muavta_rollout_record(MuavtaEnv* env, const uint64_t* seeds, int32_t n_steps, int32_t replan_interval, int32_t use_visibility,
                          int32_t write_obs, const MuavtaRecord* rec);

/* Sub-batches.  The reference's loop is "observe -> decide -> env.step" for ONE env (experiments/wps_eval.py:112-133,273); a batch
 * stepped by one launch per env step ends every launch on its slowest env (one that replans) while the host can do nothing.
 * muavta_set_parts splits the handle's env range into n_parts contiguous parts (0 or 1: off; at most 8), each with its own
 * stream (created on first use; MUAVTA_EAGER_PART_STREAMS=n in the environment makes muavta_create create n of them up front —
 * HIP binds a stream to one of its GPU_MAX_HW_QUEUES hardware queues, 4 by default, when it first gets work, and two part streams
 * that land on one queue execute in order: see INTEGRATION.md): the *_part calls below are the per-step entry points for ONE part — asynchronous on that part's stream, so the host can
 * decide for part A while the device steps part B, and the parts' launches overlap on the device.  Results are identical to the
 * whole-batch calls (env instances are independent).  The whole-batch entry points stay valid at any time: they are ordered
 * after everything the parts have queued, and a later *_part call is ordered after them (event waits on the device).
 *   muavta_part_range      first env index and env count of a part (contiguous, sizes differ by at most one)
 *   muavta_rollout_part    muavta_rollout(env, NULL, n_steps, ...) for the part's envs: the device-side planner + n_steps fused steps
 *   muavta_allocate_part   muavta_allocate for the part; act_* (may be NULL) receive [count, action_cap] rows and make the call wait
 *   muavta_step_part       muavta_step for the part with [count, action_cap] action rows, or (both NULL) muavta_step_staged
 *   muavta_observe_part    the part's rows of muavta_observe + muavta_step_result ([count, ...]; NULL = skip); waits for that part only
 *   muavta_wait_part       block until the part's stream is idle (part < 0: every part)
 * The per-env release log (muavta_set_release_log) is a whole-batch facility and must be off for muavta_step_part. */
int muavta_set_parts(MuavtaEnv* env, int32_t n_parts);
int muavta_part_range(const MuavtaEnv* env, int32_t part, int32_t* first, int32_t* count);
int muavta_rollout_part(MuavtaEnv* env, int32_t part, int32_t n_steps, int32_t replan_interval, int32_t use_visibility, int32_t write_obs);
int muavta_allocate_part(MuavtaEnv* env, int32_t part, int32_t replan_interval, int32_t use_visibility, int32_t* act_agent, int32_t* act_index);
int muavta_step_part(MuavtaEnv* env, int32_t part, const int32_t* act_agent, const int32_t* act_index);
int muavta_observe_part(MuavtaEnv* env, int32_t part, float* tasks, uint64_t* legal, uint8_t* pad, float* agents, float* flags, double* reward,
                        uint8_t* done);
int muavta_wait_part(MuavtaEnv* env, int32_t part);

/* Duration of the last muavta_rollout launch, measured with HIP events recorded on the handle's own
 * stream around the kernel (ms).  Blocks until that launch has finished.  muavta_rollout_part launches record no event pair:
 * after one, this call and muavta_kernel_ms_history return MUAVTA_E_STATE until the next whole-batch rollout. */
int muavta_last_kernel_ms(MuavtaEnv* env, float* ms);
/* The same for the last n rollout launches (1 <= n <= 64, oldest first): the handle keeps a ring of event pairs, so a
 * caller can queue launches back to back — the seeding of launch i+1 then overlaps launch i — and read the per-launch
 * durations afterwards.  Blocks until the newest of them has finished. */
int muavta_kernel_ms_history(MuavtaEnv* env, float* ms, int32_t n);
/* The gaps BETWEEN the last n rollout launches (2 <= n <= 64): ms[k] = start of launch k + 1 minus end of launch k on the handle's
 * stream, n - 1 values, oldest first — what a queue of back-to-back launches loses to command processing / seeding waits. */
int muavta_launch_gaps_ms(MuavtaEnv* env, float* ms, int32_t n);
/* Same for the RNG seeding kernel (CPython init_by_array of the four random.Random streams per env,
 * DroneEnv.py:531-538) that ran in front of that rollout; 0 when the rollout continued without seeds. */
int muavta_last_seed_ms(MuavtaEnv* env, float* ms);  /* (seed upload + k_seed run on a second stream of the handle: when launches
 * are queued back to back the figure includes the time the kernel waited for CUs the previous launch still held) */
/* Block until everything queued on the handle's stream has finished. */
int muavta_sync(MuavtaEnv* env);
/* Order the handle's (non-blocking) stream after another stream of the same device: everything queued on the handle from now on
 * starts only after the work `other_stream` (a hipStream_t; NULL = the legacy default stream) holds at the time of the call.
 * For callers that hand the library device buffers another stream may still be using — e.g. ring tensors a caching allocator
 * recycled while earlier kernels that read them are queued on the framework's stream (torch.cuda.current_stream().cuda_stream)
 * — before muavta_rollout_record / muavta_tokens_device overwrite them.  The wait also covers the sub-batch streams (their next
 * launch is ordered after the handle's stream).  The other direction is muavta_sync (or, without sub-batches, an event on the
 * stream muavta_device_ptrs returns).  The reference has no counterpart: it is single-threaded host code. */
int muavta_wait_stream(MuavtaEnv* env, void* other_stream);
/* Metrics written by the last muavta_rollout itself (f64 [N, 30]); no extra launch. */
int muavta_rollout_metrics(MuavtaEnv* env, double* out);
/* Rebuild the observation tensors from the current state (after muavta_set / muavta_set_state, or after a consumer wrote into the
 * zero-copy observation views): a FULL rewrite, pad rows included. */
int muavta_refresh_observation(MuavtaEnv* env);
/* Raw MT19937 tapes of the env's random.Random streams (checkpoint/resume next to get/set_state):
 * u32 [N, 4 streams (agent, obs, tgt, mission), 2 blocks, 624]. */
int muavta_get_rng(MuavtaEnv* env, void* dst, size_t bytes);
int muavta_set_rng(MuavtaEnv* env, const void* src, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* MUAVTA_H */
