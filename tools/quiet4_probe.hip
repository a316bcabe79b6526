// quiet4_probe.hip — builder's probe (NOT product): does packing several envs into one wave pay for the part of a
// step every env executes whatever happens in it — the "quiet skeleton"?  The same kernel source is instantiated for EPW = 1 env per
// wave (16 lanes used, one wave per env, 16 envs per CU = four waves per SIMD: the product's layout) and EPW = 4 (16 lanes per env =
// one DPP row, four envs per wave, ONE wave per SIMD: the same 16 envs per CU).  What it executes per env step, with the product's own
// arithmetic (csrc/muavta_device.h: move_parallel, displacement / norm_vector, the distance pass and np_sum_lanes16, the lane-0
// bookkeeping of step_serial_b / step_serial_c) for agents that are idle, returning to base, navigating to their head task or working
// on it:  8 gate reads (the counters a full step consults before it skips its event phases) -> movement -> distances + np.sum ->
// env scalars (total_distance, time_steps, idle-reserve count, reward = 0).  State lives in LDS for the whole launch; LDS per
// workgroup is padded to the product tile's 9,120 B per env so that residency matches.  tools/quiet4_probe.py feeds it product
// snapshots and checks the result bit for bit against the product kernel's own continuation of the same envs.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define DEV __device__ __forceinline__
constexpr double AREA_W = 1200.0, AREA_H = 700.0, BASE_X = 400.0, BASE_Y = 680.0;

DEV double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return __builtin_amdgcn_class(x, 0x260) ? x : g;
}
DEV double frcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
DEV double fdiv_r(double n, double d, double r) { const double q = n * r; return __builtin_fma(__builtin_fma(-d, q, n), r, q); }
DEV double norm2(double x, double y) { return fsqrt(__builtin_fma(y, y, x * x)); }
DEV double dpp_xchg(double v, const int sel) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  if (sel == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, true); }
  else if (sel == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, true); }
  else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, true); }
  return __hiloint2double(hi, lo);
}

struct QEnv {  // one env's share of the LDS block (the rest of its 9,120 B is padding)
  double px[16], py[16], dist[16], tx[16], ty[16], speed[16];
  int32_t state[16], task_start[16], has_task[16];
  double total_distance, last_reward;
  int32_t time_steps, idle_reserve, gates[8];
  // stand-ins for the bookkeeping a quiet step of the product still walks through (values as a quiet env holds them)
  double F_Reward, r_time_penalty, r_alloc, step_reward, rw[8], reward_norm_factor;
  int32_t n_order, n_open, n_pending, n_events, n_act, n_threats, last_plan_step, n_dev, pending_reset, next_task_id, conclusion_time,
      terminated, truncated, max_time_steps, interval, n_tasks;
  uint32_t rng_idx[4], rng_at[4];
  uint8_t t_order[40], t_status[40], t_flags[40], t_type[40];
  int16_t t_deadline[40], pend_time[48];
};
constexpr int ENV_LDS = 9120;
static_assert(sizeof(QEnv) <= ENV_LDS, "probe state exceeds the product tile");

template <int EPW>
__global__ __launch_bounds__(64) void k_quiet(const QEnv* in, QEnv* out, int n_steps) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int sub = EPW == 1 ? 0 : lane >> 4;           // which env of the wave this lane works for
  const int a = lane & 15;                            // agent
  const bool mine = EPW == 4 || lane < 16;
  const int env = blockIdx.x * EPW + sub;
  QEnv& S = *reinterpret_cast<QEnv*>(smem + sub * ENV_LDS);
  {  // load
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (size_t)blockIdx.x * EPW);
    for (int e = 0; e < EPW; e++) {
      uint32_t* dst = reinterpret_cast<uint32_t*>(smem + e * ENV_LDS);
      for (int i = lane; i < (int)(sizeof(QEnv) / 4); i += 64) dst[i] = src[e * (sizeof(QEnv) / 4) + i];
    }
  }
  __syncthreads();
  int tnow = S.time_steps;
  for (int t = 0; t < n_steps; t++) {
    // ---- gates: the counters a step reads to find that there are no events to drain, no actions staged, no threats, no arrivals, no
    // pending reveals / windows, no escorts, nothing to collect, no re-plan due.  A wave leaves the fast path if ANY of its envs does.
    bool slow = false;
#pragma unroll
    for (int g = 0; g < 8; g++) slow |= mine && S.gates[g] != 0;
    // (stand-ins, same shape as the product's quiet path) RNG window check: cursors against the prefetched window, one 16-byte read each
    {
#pragma unroll
      for (int q = 0; q < 4; q++) slow |= mine && (S.rng_idx[q] - S.rng_at[q]) >= 6u;
    }
    // event drain + action gate + allocator gate (should_replan: the interval or an event)
    const int nev = S.n_events, nact = S.n_act;
    slow |= mine && (nev > 0 || nact > 0 || (tnow - S.last_plan_step >= S.interval) || S.n_dev > 0);
    // threat spawn gate / threat pass / arrivals / sensing: configuration switches and counters
    slow |= mine && (S.n_threats > 0);
    if (__ballot(slow) != 0ull) { if (lane == 0) out[0].last_reward = -1.0; }  // (never in the probe's inputs)
    tnow += 1;
    // ---- movement (move_parallel: every operand up front, one pipeline per lane) ----
    double px = S.px[a], py = S.py[a];
    const double prev_x = px, prev_y = py;
    const int st0 = S.state[a], ts0 = S.task_start[a], has = S.has_task[a];
    const double speed = S.speed[a], tpx = S.tx[a], tpy = S.ty[a];
    int new_st = st0, new_ts = ts0;
    double ddx = 0.0, ddy = 0.0;
    if (mine && st0 != -1) {
      const bool to_task = has != 0;
      const bool idle_check = new_st == 0 && !has;
      const bool to_base = !to_task && (idle_check || new_st == 3);
      if (to_task || to_base) {
        const double gx = to_task ? tpx : BASE_X, gy = to_task ? tpy : BASE_Y;
        const double dx = gx - px, dy = gy - py;
        const double dist = norm2(dx, dy);
        double ux = 0, uy = 0;
        const bool zero = to_task ? (fabs(dist) < 1e-12) : (dist == 0);
        if (!zero) { const double r = frcp_nr(dist); ux = fdiv_r(dx, dist, r); uy = fdiv_r(dy, dist, r); }
        // displacement(): avoid_obstacles == (0, 0) without obstacles, then norm_vector(m) * speed
        double mx = ux + 0.0, my = uy + 0.0;
        const double m = norm2(mx, my);
        if (m == 0) { mx = 0; my = 0; } else { const double r = frcp_nr(m); mx = fdiv_r(mx, m, r); my = fdiv_r(my, m, r); }
        const double ndx = mx * speed, ndy = my * speed;
        if (to_task) {
          if (new_st == 1) {
            if (dist < speed) { new_st = 2; new_ts = tnow; px = tpx; py = tpy; }
            else { ddx = ndx; ddy = ndy; }
          } else if (new_st == 2) {
            if (new_ts == -1) { new_ts = tnow; px = tpx; py = tpy; }
          }
        } else {
          if (idle_check && dist > speed + 5) new_st = 3;
          if (new_st == 3) {
            if (dist < speed + 5) new_st = 0;
            else { ddx = ndx; ddy = ndy; }
          }
        }
      }
      S.state[a] = new_st; S.task_start[a] = new_ts;
      px = px + ddx; py = py + ddy;
      px = fmin(fmax(px, 0.0), AREA_W); py = fmin(fmax(py, 0.0), AREA_H);
      S.px[a] = px; S.py[a] = py;
    }
    // ---- distances (:1131-1138) + np.sum over the env's 16 agents: one DPP row ----
    double d = 0.0;
    if (mine) {
      const double dx = px - prev_x, dy = py - prev_y;
      d = fsqrt(dx * dx + dy * dy);
      S.dist[a] += d;
    }
    double r;
    {
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(d), 0x108, 0xf, 0xf, true);   // row_shl:8
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(d), 0x108, 0xf, 0xf, true);
      r = d + __hiloint2double(hi, lo);
      r = r + dpp_xchg(r, 0);
      r = r + dpp_xchg(r, 1);
      r = r + dpp_xchg(r, 2);
    }
    // ---- window / reveal pre-checks (one pass over the pending list and the live slots, four ballots) ----
    bool due = false, expiring = false, blocking = false;
    if (mine) {
      for (int k = a; k < S.n_pending; k += 16) due |= tnow >= S.pend_time[k];
      for (int k = a; k < S.n_order; k += 16) {
        const int sl = S.t_order[k];
        const int fl = S.t_flags[sl], stt = S.t_status[sl], dl = S.t_deadline[sl], ty = S.t_type[sl];
        expiring |= (fl & 16) && stt != 2 && tnow > dl;
        blocking |= !((fl & 1) || ty == 5 || ty == 0 || stt == 2);
      }
    }
    const unsigned long long due_m = __ballot(due), exp_m = __ballot(expiring), blk_m = __ballot(blocking);
    const unsigned long long idle_m = __ballot(mine && new_st != -1 && !has), resp_m = __ballot(mine && new_st != -1 && has);
    const unsigned long long rowmask = 0xffffull << (sub * 16);
    if (((due_m | exp_m) & rowmask) != 0ull && lane == 0) out[0].last_reward = -2.0;  // (never in the probe's inputs)
    // ---- env scalars: lane 0 of the env's row (step_serial_b + step_serial_c: reward assembly, done flags) ----
    if (mine && a == 0) {
      const int n_idle = __popcll(idle_m & rowmask);
      const bool responding = (resp_m & rowmask) != 0ull, all_done_tasks = (blk_m & rowmask) == 0ull;
      S.total_distance += r;
      S.time_steps = tnow;
      const int irs = S.idle_reserve, pr = S.pending_reset, ntid = S.next_task_id, ct = S.conclusion_time;
      const double tp = S.r_time_penalty, al = S.r_alloc, sr = S.step_reward, FR = S.F_Reward;
      const double total = S.rw[0] * 0.0 + S.rw[1] * 0.0 + S.rw[2] * 0.0 + S.rw[3] * 0.0 + S.rw[4] * (double)S.n_tasks * 0.0 + S.rw[5] * al + S.rw[6] * tp + S.rw[7] * sr;
      const double shared = total / S.reward_norm_factor / (double)S.max_time_steps;
      const bool all_done = (ntid > 1) && all_done_tasks;
      const bool timed_out = (tnow >= S.max_time_steps) && (S.max_time_steps > 0);
      S.idle_reserve = irs + n_idle;
      if (pr && responding) S.pending_reset = 0;
      if (all_done && ct > S.max_time_steps) S.conclusion_time = tnow;
      S.terminated = 0; S.truncated = timed_out;
      S.last_reward = timed_out ? FR : shared;
      S.step_reward = 0; S.n_dev = nev; S.n_events = 0;
    }
    // ---- end of step: slot GC fast path (no live slot retired, every live slot listed as open) ----
    {
      bool ret = false;
      if (mine) for (int k = a; k < S.n_order; k += 16) ret |= S.t_status[S.t_order[k]] == 2;
      const unsigned long long ret_m = __ballot(ret);
      if ((ret_m & rowmask) != 0ull && lane == 0) out[0].last_reward = -3.0;
      if (mine && a == 0 && S.n_open == S.n_order) S.n_act = 0;
    }
    __syncthreads();
    // ---- the rollout loop's own bookkeeping: episode-over flags back from LDS ----
    if (__ballot(mine && (S.terminated || S.truncated)) == ~0ull) break;  // (uniform; quiet episodes run to the horizon)
  }
  {  // store
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (size_t)blockIdx.x * EPW);
    for (int e = 0; e < EPW; e++) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(smem + e * ENV_LDS);
      for (int i = lane; i < (int)(sizeof(QEnv) / 4); i += 64) dst[e * (sizeof(QEnv) / 4) + i] = src[i];
    }
  }
}

extern "C" int quiet_probe_env_bytes() { return (int)sizeof(QEnv); }
// in / out: host arrays of n_envs QEnv records; returns the mean kernel time (ms) over `reps` launches in *ms, < 0 on a HIP error
extern "C" int quiet_probe_run(const void* in, void* out, int n_envs, int epw, int n_steps, int reps, float* ms) {
  if ((epw != 1 && epw != 4) || n_envs % epw) return -1;
  QEnv *d_in = nullptr, *d_out = nullptr;
  const size_t bytes = (size_t)n_envs * sizeof(QEnv);
  if (hipMalloc(&d_in, bytes) != hipSuccess || hipMalloc(&d_out, bytes) != hipSuccess) return -2;
  if (hipMemcpy(d_in, in, bytes, hipMemcpyHostToDevice) != hipSuccess) return -3;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)ENV_LDS * epw;
  if (epw == 4) hipFuncSetAttribute(reinterpret_cast<const void*>(&k_quiet<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float total = 0.f;
  for (int r = 0; r < reps + 1; r++) {
    hipEventRecord(e0, 0);
    if (epw == 1) hipLaunchKernelGGL(k_quiet<1>, dim3(n_envs), dim3(64), lds, 0, d_in, d_out, n_steps);
    else hipLaunchKernelGGL(k_quiet<4>, dim3(n_envs / 4), dim3(64), lds, 0, d_in, d_out, n_steps);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -4;
    float t = 0.f;
    hipEventElapsedTime(&t, e0, e1);
    if (r) total += t;  // (first launch: warm-up)
  }
  *ms = total / reps;
  if (hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) return -5;
  hipFree(d_in); hipFree(d_out); hipEventDestroy(e0); hipEventDestroy(e1);
  return 0;
}
