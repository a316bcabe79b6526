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

// ---- scene constants (mUAV_TA/MultiDroneEnvData.py:8-85) ------------------------------------------
constexpr double AREA_W = 1200.0, AREA_H = 700.0, CONTACT_LINE = 550.0, BASE_X = 400.0, BASE_Y = 680.0;
constexpr double MAX_COORD = 1200.0;
__constant__ double CAP_TABLE[7][6] = {
    {0.1, 1.0, 0.0, 0.2, 0.0, 0.0}, {0.1, 0.6, 0.0, 0.1, 0.0, 0.0}, {0.1, 0.8, 0.0, 0.2, 0.0, 1.0},
    {0.1, 0.0, 0.7, 1.0, 1.0, 1.0}, {0.1, 0.0, 1.0, 0.6, 0.8, 1.0}, {0.0, 0.0, 0.2, 0.5, 1.0, 1.0},
    {0.0, 0.0, 0.2, 0.4, 0.8, 0.8}};
__constant__ double ENGAGE_RANGE[7] = {0.0, 0.0, 0.0, 40.0, 30.0, 35.0, 25.0};
__constant__ double FAIL_MULT[7] = {1.2, 0.8, 1.5, 1.5, 0.8, 1.8, 1.0};
__constant__ int TASK_DURATION[6] = {1, 10, 5, 5, 0, 1};
// The same tables as select chains / packed constants: an indexed read of __constant__ (or of a kernarg
// array) with a per-lane index is a vector memory load with hundreds of cycles of exposed latency on the
// order-dependent paths; these cost a few VALU ops.
DEV int task_duration(int ty) { return (int)((0x010005050A01ull >> (8 * ty)) & 0xffull); }  // Hold, Rec, Att, Def, Int, Det
DEV double engage_range(int t) { return t == MUAVTA_F1 ? 40.0 : t == MUAVTA_F2 ? 30.0 : t == MUAVTA_T1 ? 35.0 : t == MUAVTA_T2 ? 25.0 : 0.0; }
DEV double threat_attack(int t) { return 0.2; }                          // UavCapTable[T1/T2][Att]
DEV double threat_defence(int t) { return t == MUAVTA_T1 ? 0.5 : 0.4; }  // UavCapTable[T1/T2][Def]

// Correctly rounded f64 square root and division for operands of this simulation's range: the very instruction sequences
// the compiler expands `sqrt` and `/` into (v_rsq_f64 / v_rcp_f64 seed + FMA refinement + one correction step), without the
// range scaling in front (v_div_scale_f64 x2 / v_cmp + v_ldexp x2 + v_cndmask) and the special-case fix-up behind
// (v_div_fixup_f64).  Those only act on operands the simulation cannot produce — a radicand below 2^-767, a numerator below
// 2^-970 that is not zero, a zero / denormal / infinite divisor, exponents >= 768 apart — so inside the domain every
// intermediate is the same and so is the result, bit for bit (tests/test_gpu_parity.py::test_domain_sqrt_div_bit_exact pins
// them against numpy on the GPU).  sqrt: 13 VALU instead of 20; division: 8 instead of 11, and a second quotient by the same
// divisor costs 3 (the reciprocal refinement is shared).  The kernels are VALU-issue bound.
//   fsqrt(x):       x is +0, -0, +inf, NaN, or >= 2^-767  (sums of squares of coordinate differences: 0 or >= 2^-200)
//   frcp_nr(d):     d finite, 2^-250 <= |d| <= 2^250       (distances >= 1e-12 where the callers guard, speeds, counts)
//   fdiv_r(n,d,r):  n == 0 (either sign: +0 results, callers never hold -0) or 2^-250 <= |n| <= 2^250;  r = frcp_nr(d)
DEV double fsqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = y * 0.5;
  const double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  return __builtin_amdgcn_class(x, 0x260) ? x : g;  // -0, +0, +inf: the radicand itself
}
DEV double frcp_nr(double d) {
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, e, r);
}
DEV double fdiv_r(double n, double d, double r) {
  const double q = n * r;
  return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}
DEV double fdiv(double n, double d) { return fdiv_r(n, d, frcp_nr(d)); }
DEV double norm2(double x, double y) { return fsqrt(fma(y, y, x * x)); }  // np.linalg.norm of a 2-vector

// log(x) as the host's libm computes it — the natural logarithm of Arm's "optimized-routines" (math/log.c + math/log_data.c, N = 128,
// Szabolcs Nagy 2018; the implementation glibc >= 2.28 ships as sysdeps/ieee754/dbl-64/e_log.c), with the multiply-adds fused where
// the FMA build of this image's glibc 2.35 (the variant its ifunc picks on every FMA + AVX2 CPU) fuses them.  Why it is here:
// core_sim's avoid_obstacles (sim_core.rs:25-59) calls f64::ln, i.e. the reference's repulsion force IS this function's value, and
// the oracle calls it through std::log; ocml's log differs from it in the last bit for roughly one argument in seven, which made
// the K > 0 obstacle path the one place where device and oracle agreed to 1e-9 only.  With this restatement they agree bit for
// bit (tests/test_gpu_parity.py::test_libm_log_bit_exact: device vs the host's log on 2 M arguments).  Published algorithm:
//   x = 2^k z, z in [OFF, 2 OFF), i = top 7 mantissa bits of (x - OFF);  r = fma(z, invc[i], -1);  w = k Ln2hi + logc[i];
//   hi = w + r;  lo = (w - hi) + r + k Ln2lo;  log x = lo + r^2 A0 + r r^2 (A1 + r A2 + r^2 (A3 + r A4)) + hi;
//   near 1 (|x - 1| < ~0.0647): a degree-11 polynomial in r = x - 1 with r split at 2^27 for an exact r - r^2 / 2.
// Domain: positive, finite, normal x (the caller passes max(1.05, d_zone) with d_zone < 40).
__constant__ double LIBM_LOG_TAB[128][2] = {  // {invc, logc}: log_data.c, N = 128
    {0x1.734f0c3e0de9fp+0, -0x1.7cc7f79e69000p-2}, {0x1.713786a2ce91fp+0, -0x1.76feec20d0000p-2},
    {0x1.6f26008fab5a0p+0, -0x1.713e31351e000p-2}, {0x1.6d1a61f138c7dp+0, -0x1.6b85b38287800p-2},
    {0x1.6b1490bc5b4d1p+0, -0x1.65d5590807800p-2}, {0x1.69147332f0cbap+0, -0x1.602d076180000p-2},
    {0x1.6719f18224223p+0, -0x1.5a8ca86909000p-2}, {0x1.6524f99a51ed9p+0, -0x1.54f4356035000p-2},
    {0x1.63356aa8f24c4p+0, -0x1.4f637c36b4000p-2}, {0x1.614b36b9ddc14p+0, -0x1.49da7fda85000p-2},
    {0x1.5f66452c65c4cp+0, -0x1.445923989a800p-2}, {0x1.5d867b5912c4fp+0, -0x1.3edf439b0b800p-2},
    {0x1.5babccb5b90dep+0, -0x1.396ce448f7000p-2}, {0x1.59d61f2d91a78p+0, -0x1.3401e17bda000p-2},
    {0x1.5805612465687p+0, -0x1.2e9e2ef468000p-2}, {0x1.56397cee76bd3p+0, -0x1.2941b3830e000p-2},
    {0x1.54725e2a77f93p+0, -0x1.23ec58cda8800p-2}, {0x1.52aff42064583p+0, -0x1.1e9e129279000p-2},
    {0x1.50f22dbb2bddfp+0, -0x1.1956d2b48f800p-2}, {0x1.4f38f4734ded7p+0, -0x1.141679ab9f800p-2},
    {0x1.4d843cfde2840p+0, -0x1.0edd094ef9800p-2}, {0x1.4bd3ec078a3c8p+0, -0x1.09aa518db1000p-2},
    {0x1.4a27fc3e0258ap+0, -0x1.047e65263b800p-2}, {0x1.4880524d48434p+0, -0x1.feb224586f000p-3},
    {0x1.46dce1b192d0bp+0, -0x1.f474a7517b000p-3}, {0x1.453d9d3391854p+0, -0x1.ea4443d103000p-3},
    {0x1.43a2744b4845ap+0, -0x1.e020d44e9b000p-3}, {0x1.420b54115f8fbp+0, -0x1.d60a22977f000p-3},
    {0x1.40782da3ef4b1p+0, -0x1.cc00104959000p-3}, {0x1.3ee8f5d57fe8fp+0, -0x1.c202956891000p-3},
    {0x1.3d5d9a00b4ce9p+0, -0x1.b81178d811000p-3}, {0x1.3bd60c010c12bp+0, -0x1.ae2c9ccd3d000p-3},
    {0x1.3a5242b75dab8p+0, -0x1.a45402e129000p-3}, {0x1.38d22cd9fd002p+0, -0x1.9a877681df000p-3},
    {0x1.3755bc5847a1cp+0, -0x1.90c6d69483000p-3}, {0x1.35dce49ad36e2p+0, -0x1.87120a645c000p-3},
    {0x1.34679984dd440p+0, -0x1.7d68fb4143000p-3}, {0x1.32f5cceffcb24p+0, -0x1.73cb83c627000p-3},
    {0x1.3187775a10d49p+0, -0x1.6a39a9b376000p-3}, {0x1.301c8373e3990p+0, -0x1.60b3154b7a000p-3},
    {0x1.2eb4ebb95f841p+0, -0x1.5737d76243000p-3}, {0x1.2d50a0219a9d1p+0, -0x1.4dc7b8fc23000p-3},
    {0x1.2bef9a8b7fd2ap+0, -0x1.4462c51d20000p-3}, {0x1.2a91c7a0c1babp+0, -0x1.3b08abc830000p-3},
    {0x1.293726014b530p+0, -0x1.31b996b490000p-3}, {0x1.27dfa5757a1f5p+0, -0x1.2875490a44000p-3},
    {0x1.268b39b1d3bbfp+0, -0x1.1f3b9f879a000p-3}, {0x1.2539d838ff5bdp+0, -0x1.160c8252ca000p-3},
    {0x1.23eb7aac9083bp+0, -0x1.0ce7f57f72000p-3}, {0x1.22a012ba940b6p+0, -0x1.03cdc49fea000p-3},
    {0x1.2157996cc4132p+0, -0x1.f57bdbc4b8000p-4}, {0x1.201201dd2fc9bp+0, -0x1.e370896404000p-4},
    {0x1.1ecf4494d480bp+0, -0x1.d17983ef94000p-4}, {0x1.1d8f5528f6569p+0, -0x1.bf9674ed8a000p-4},
    {0x1.1c52311577e7cp+0, -0x1.adc79202f6000p-4}, {0x1.1b17c74cb26e9p+0, -0x1.9c0c3e7288000p-4},
    {0x1.19e010c2c1ab6p+0, -0x1.8a646b372c000p-4}, {0x1.18ab07bb670bdp+0, -0x1.78d01b3ac0000p-4},
    {0x1.1778a25efbcb6p+0, -0x1.674f145380000p-4}, {0x1.1648d354c31dap+0, -0x1.55e0e6d878000p-4},
    {0x1.151b990275fddp+0, -0x1.4485cdea1e000p-4}, {0x1.13f0ea432d24cp+0, -0x1.333d94d6aa000p-4},
    {0x1.12c8b7210f9dap+0, -0x1.22079f8c56000p-4}, {0x1.11a3028ecb531p+0, -0x1.10e4698622000p-4},
    {0x1.107fbda8434afp+0, -0x1.ffa6c6ad20000p-5}, {0x1.0f5ee0f4e6bb3p+0, -0x1.dda8d4a774000p-5},
    {0x1.0e4065d2a9fcep+0, -0x1.bbcece4850000p-5}, {0x1.0d244632ca521p+0, -0x1.9a1894012c000p-5},
    {0x1.0c0a77ce2981ap+0, -0x1.788583302c000p-5}, {0x1.0af2f83c636d1p+0, -0x1.5715e67d68000p-5},
    {0x1.09ddb98a01339p+0, -0x1.35c8a49658000p-5}, {0x1.08cabaf52e7dfp+0, -0x1.149e364154000p-5},
    {0x1.07b9f2f4e28fbp+0, -0x1.e72c082eb8000p-6}, {0x1.06ab58c358f19p+0, -0x1.a55f152528000p-6},
    {0x1.059eea5ecf92cp+0, -0x1.63d62cf818000p-6}, {0x1.04949cdd12c90p+0, -0x1.228fb8caa0000p-6},
    {0x1.038c6c6f0ada9p+0, -0x1.c317b20f90000p-7}, {0x1.02865137932a9p+0, -0x1.419355daa0000p-7},
    {0x1.0182427ea7348p+0, -0x1.81203c2ec0000p-8}, {0x1.008040614b195p+0, -0x1.0040979240000p-9},
    {0x1.fe01ff726fa1ap-1, 0x1.feff384900000p-9}, {0x1.fa11cc261ea74p-1, 0x1.7dc41353d0000p-7},
    {0x1.f6310b081992ep-1, 0x1.3cea3c4c28000p-6}, {0x1.f25f63ceeadcdp-1, 0x1.b9fc114890000p-6},
    {0x1.ee9c8039113e7p-1, 0x1.1b0d8ce110000p-5}, {0x1.eae8078cbb1abp-1, 0x1.58a5bd001c000p-5},
    {0x1.e741aa29d0c9bp-1, 0x1.95c8340d88000p-5}, {0x1.e3a91830a99b5p-1, 0x1.d276aef578000p-5},
    {0x1.e01e009609a56p-1, 0x1.07598e598c000p-4}, {0x1.dca01e577bb98p-1, 0x1.253f5e30d2000p-4},
    {0x1.d92f20b7c9103p-1, 0x1.42edd8b380000p-4}, {0x1.d5cac66fb5ccep-1, 0x1.606598757c000p-4},
    {0x1.d272caa5ede9dp-1, 0x1.7da76356a0000p-4}, {0x1.cf26e3e6b2ccdp-1, 0x1.9ab434e1c6000p-4},
    {0x1.cbe6da2a77902p-1, 0x1.b78c7bb0d6000p-4}, {0x1.c8b266d37086dp-1, 0x1.d431332e72000p-4},
    {0x1.c5894bd5d5804p-1, 0x1.f0a3171de6000p-4}, {0x1.c26b533bb9f8cp-1, 0x1.067152b914000p-3},
    {0x1.bf583eeece73fp-1, 0x1.147858292b000p-3}, {0x1.bc4fd75db96c1p-1, 0x1.2266ecdca3000p-3},
    {0x1.b951e0c864a28p-1, 0x1.303d7a6c55000p-3}, {0x1.b65e2c5ef3e2cp-1, 0x1.3dfc33c331000p-3},
    {0x1.b374867c9888bp-1, 0x1.4ba366b7a8000p-3}, {0x1.b094b211d304ap-1, 0x1.5933928d1f000p-3},
    {0x1.adbe885f2ef7ep-1, 0x1.66acd2418f000p-3}, {0x1.aaf1d31603da2p-1, 0x1.740f8ec669000p-3},
    {0x1.a82e63fd358a7p-1, 0x1.815c0f51af000p-3}, {0x1.a5740ef09738bp-1, 0x1.8e92954f68000p-3},
    {0x1.a2c2a90ab4b27p-1, 0x1.9bb3602f84000p-3}, {0x1.a01a01393f2d1p-1, 0x1.a8bed1c2c0000p-3},
    {0x1.9d79f24db3c1bp-1, 0x1.b5b515c01d000p-3}, {0x1.9ae2505c7b190p-1, 0x1.c2967ccbcc000p-3},
    {0x1.9852ef297ce2fp-1, 0x1.cf635d5486000p-3}, {0x1.95cbaeea44b75p-1, 0x1.dc1bd3446c000p-3},
    {0x1.934c69de74838p-1, 0x1.e8c01b8cfe000p-3}, {0x1.90d4f2f6752e6p-1, 0x1.f5509c0179000p-3},
    {0x1.8e6528effd79dp-1, 0x1.00e6c121fb800p-2}, {0x1.8bfce9fcc007cp-1, 0x1.071b80e93d000p-2},
    {0x1.899c0dabec30ep-1, 0x1.0d46b9e867000p-2}, {0x1.87427aa2317fbp-1, 0x1.13687334bd000p-2},
    {0x1.84f00acb39a08p-1, 0x1.1980d67234800p-2}, {0x1.82a49e8653e55p-1, 0x1.1f8ffe0cc8000p-2},
    {0x1.8060195f40260p-1, 0x1.2595fd7636800p-2}, {0x1.7e22563e0a329p-1, 0x1.2b9300914a800p-2},
    {0x1.7beb377dcb5adp-1, 0x1.3187210436000p-2}, {0x1.79baa679725c2p-1, 0x1.377266dec1800p-2},
    {0x1.77907f2170657p-1, 0x1.3d54ffbaf3000p-2}, {0x1.756cadbd6130cp-1, 0x1.432eee32fe000p-2},
};
DEV double libm_log(double x) {
  const unsigned long long ix = (unsigned long long)__double_as_longlong(x);
  if (ix - 0x3fee000000000000ull < 0x3090000000000ull) {  // 1 - 2^-4 <= x < 1 + 0x1.09p-4
    if (ix == 0x3ff0000000000000ull) return 0.0;
    const double B0 = -0x1p-1, B1 = 0x1.5555555555577p-2, B2 = -0x1.ffffffffffdcbp-3, B3 = 0x1.999999995dd0cp-3, B4 = -0x1.55555556745a7p-3,
                 B5 = 0x1.24924a344de3p-3, B6 = -0x1.fffffa4423d65p-4, B7 = 0x1.c7184282ad6cap-4, B8 = -0x1.999eb43b068ffp-4,
                 B9 = 0x1.78182f7afd085p-4, B10 = -0x1.5521375d145cdp-4;
    const double r = x - 1.0;
    double p2 = __builtin_fma(r, B2, B1), p3 = __builtin_fma(r, B5, B4), p5 = __builtin_fma(r, B8, B7);
    const double r2 = r * r;
    p2 = __builtin_fma(r2, B3, p2); p3 = __builtin_fma(r2, B6, p3);
    const double r3 = r * r2;
    double p = __builtin_fma(r2, B9, p5);
    p = __builtin_fma(r3, B10, p); p = __builtin_fma(p, r3, p3); p = __builtin_fma(p, r3, p2);
    const double t = __builtin_fma(r, 0x1p27, r), rhi = __builtin_fma(-0x1p27, r, t);
    const double rhi2 = rhi * rhi, rlo = r - rhi;
    const double hi = __builtin_fma(rhi2, B0, r);
    double lo = __builtin_fma(rhi2, B0, r - hi);
    lo = __builtin_fma(B0 * rlo, r + rhi, lo);
    return __builtin_fma(p, r3, lo) + hi;
  }
  const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
  const double A0 = -0x1.0000000000001p-1, A1 = 0x1.555555551305bp-2, A2 = -0x1.fffffffeb459p-3, A3 = 0x1.999b324f10111p-3, A4 = -0x1.55575e506c89fp-3;
  const unsigned long long tmp = ix - 0x3fe6000000000000ull;
  const int i = (int)(tmp >> 45) & 127;
  const int k = (int)((long long)tmp >> 52);
  const double z = __longlong_as_double((long long)(ix - (tmp & (0xfffull << 52))));
  const double invc = LIBM_LOG_TAB[i][0], logc = LIBM_LOG_TAB[i][1];
  const double r = __builtin_fma(z, invc, -1.0), kd = (double)k;
  const double w = __builtin_fma(kd, Ln2hi, logc);
  const double p = __builtin_fma(r, A2, A1);
  const double hi = r + w, r2 = r * r;
  double lo = (w - hi) + r;
  lo = __builtin_fma(kd, Ln2lo, lo);
  const double rr2 = r * r2;
  double q = __builtin_fma(r, A4, A3);
  lo = __builtin_fma(r2, A0, lo);
  q = __builtin_fma(q, r2, p);
  return __builtin_fma(rr2, q, lo) + hi;
}
DEV bool is_recon(int t) { return t == MUAVTA_R1 || t == MUAVTA_R2; }
DEV bool is_fighter(int t) { return t == MUAVTA_F1 || t == MUAVTA_F2; }

// Timing experiments only (diagnostic builds; results are wrong unless the ablated phase has nothing to do, as in tools/quiet_probe.py's
// quiet workload): -DMUAVTA_ABLATE=<bit mask> compiles a phase of the step out, so that the launch time without it is its true cost.
#ifndef MUAVTA_ABLATE
#define MUAVTA_ABLATE 0
#endif
#define ABL(bit) ((MUAVTA_ABLATE >> (bit)) & 1)

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

// ====================================================================================================
// CPython random.Random on a per-env tape in HBM: each stream keeps two consecutive raw MT19937
// blocks (2 x 624 words); lane 0 tempers words at the cursor, and the whole wave regenerates a
// consumed block at a step boundary (Modules/_randommodule.c genrand_uint32 / init_by_array).
// ====================================================================================================
enum { ST_AGENT = 0, ST_OBS = 1, ST_TGT = 2, ST_MISSION = 3 };

DEV uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t m) {
  uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
  return m ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
// dst = next MT block after src (src/dst: LDS or global, distinct buffers).  All lanes.
DEV void mt_twist(const uint32_t* src, uint32_t* dst) {
  const int lane = threadIdx.x;
  for (int k = lane; k < 227; k += WG) dst[k] = mt_mix(src[k], src[k + 1], src[k + 397]);
  __syncthreads();
  for (int k = 227 + lane; k < 454; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  __syncthreads();
  for (int k = 454 + lane; k < 623; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  __syncthreads();
  if (lane == 0) dst[623] = mt_mix(src[623], dst[0], dst[396]);
  __syncthreads();
}
// Same on LDS buffers of a single-wave workgroup: phases are ordered by lds_sync() (no vmcnt drain).
DEV void mt_twist_lds(const uint32_t* src, uint32_t* dst) {
  const int lane = threadIdx.x;
  for (int k = lane; k < 227; k += WG) dst[k] = mt_mix(src[k], src[k + 1], src[k + 397]);
  lds_sync();
  for (int k = 227 + lane; k < 454; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  lds_sync();
  for (int k = 454 + lane; k < 623; k += WG) dst[k] = mt_mix(src[k], src[k + 1], dst[k - 227]);
  lds_sync();
  if (lane == 0) dst[623] = mt_mix(src[623], dst[0], dst[396]);
  lds_sync();
}
// init_by_array(key[0..len)) into mt[624] (LDS).  One lane.
DEV void mt_seed(uint32_t* mt, uint32_t k0, uint32_t k1, int len) {
  uint32_t g = 19650218u;  // init_genrand(19650218) generated on the fly
  uint32_t prev = g;
  mt[0] = g;
  int j = 0;
  for (int i = 1; i < 624; i++) {
    g = 1812433253u * (g ^ (g >> 30)) + (uint32_t)i;
    uint32_t key = j ? k1 : k0;
    prev = (g ^ ((prev ^ (prev >> 30)) * 1664525u)) + key + (uint32_t)j;
    mt[i] = prev;
    j++;
    if (j >= len) j = 0;
  }
  // 624th iteration of the first loop: i wrapped to 1 with mt[0] = mt[623]
  mt[0] = prev;
  {
    uint32_t key = j ? k1 : k0;
    prev = (mt[1] ^ ((prev ^ (prev >> 30)) * 1664525u)) + key + (uint32_t)j;
    mt[1] = prev;
  }
  // second loop (i = 2..623): the recurrence is serial in `prev`, but the mt[i] operands are first-loop values
  // whose addresses are known, so they are fetched eight at a time (one LDS wait per 8 steps, not per step)
  {
    int i = 2;
    for (; i + 8 <= 624; i += 8) {
      uint32_t m[8];
#pragma unroll
      for (int q = 0; q < 8; q++) m[q] = mt[i + q];
#pragma unroll
      for (int q = 0; q < 8; q++) {
        prev = (m[q] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)(i + q);
        m[q] = prev;
      }
#pragma unroll
      for (int q = 0; q < 8; q++) mt[i + q] = m[q];
    }
    for (; i < 624; i++) {
      prev = (mt[i] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - (uint32_t)i;
      mt[i] = prev;
    }
  }
  mt[0] = prev;
  prev = (mt[1] ^ ((prev ^ (prev >> 30)) * 1566083941u)) - 1u;
  mt[1] = prev;
  mt[0] = 0x80000000u;
}

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

  // ---------------------------------------------------------------- RNG (lane 0 unless noted)
  DEV uint32_t next32(int st) {
    uint32_t p = S.rng_idx[st];
    uint32_t blk = (p >> 16) & 1u, off = p & 0xffffu;  // high half: which block is "current"
    if (off >= 1248u) { fail(MUAVTA_ERR_POSITION); off = 1247u; }
    uint32_t y;
    const uint32_t w = p - S.rng_win_at[st];  // same block marker => plain cursor difference
    if (w < win_len) {
      y = win_ptr[st * win_stride + w];  // prefetched into LDS at the step boundary
    } else {
      uint32_t b = off >= 624u ? (blk ^ 1u) : blk;
      uint32_t o = off >= 624u ? off - 624u : off;
      y = tape[st * MUAVTA_RNG_WORDS + b * 624u + o];
    }
    S.rng_idx[st] = (blk << 16) | (off + 1u);
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  DEV static uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  DEV double rnd(int st) {  // random.random()
    // fast path (the per-step arrival / spawn / engagement draws): both words sit in the prefetched LDS window — cursor and
    // window base in one round trip, the two words in a second one, instead of a dependent chain per word
    const uint32_t p = S.rng_idx[st], at = S.rng_win_at[st];
    const uint32_t w = p - at, off = p & 0xffffu;
    uint32_t a, b;
    if (w < win_len - 1u && off + 2u <= 1248u) {
      const uint32_t y0 = win_ptr[st * win_stride + w], y1 = win_ptr[st * win_stride + w + 1u];
      S.rng_idx[st] = p + 2u;  // (block marker in the high half is untouched: off + 2 <= 1248)
      a = mt_temper(y0) >> 5; b = mt_temper(y1) >> 6;
    } else {
      a = next32(st) >> 5; b = next32(st) >> 6;
    }
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
  }
  DEV uint64_t getrandbits(int st, int k) {
    if (k <= 32) return (uint64_t)(next32(st) >> (32 - k));
    uint64_t lo = next32(st), hi = next32(st);
    int rem = k - 32;
    if (rem < 32) hi >>= (32 - rem);
    return lo | (hi << 32);
  }
  DEV uint64_t randbelow(int st, uint64_t n) {  // Random._randbelow_with_getrandbits
    int k = 64 - __clzll((long long)n);
    uint64_t r = getrandbits(st, k);
    while (r >= n) r = getrandbits(st, k);
    return r;
  }
  DEV int64_t randint(int st, int64_t a, int64_t b) { return a + (int64_t)randbelow(st, (uint64_t)(b - a) + 1u); }
  DEV double uniform(int st, double a, double b) { return a + (b - a) * rnd(st); }

  // All lanes: regenerate consumed blocks (called at step boundaries, uniform control flow).
  DEV void rng_refill() {
    {  // fast path (almost every step): no stream has consumed its current block — one 16-byte LDS read decides
      const uint4 c = *reinterpret_cast<const uint4*>(&S.rng_idx[0]);
      if ((c.x & 0xffffu) < 624u && (c.y & 0xffffu) < 624u && (c.z & 0xffffu) < 624u && (c.w & 0xffffu) < 624u) return;
    }
    rng_refill_slow(tape, (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&S.rng_idx[0]);
  }
  // The regeneration itself, OUT OF LINE: a stream consumes a 624-word block only in long or busy episodes, but inlined into the
  // step loop its per-lane tape addresses (eleven 64-bit values) were computed at the top of EVERY step and spilled to scratch
  // on the 24-agent tile.  Takes the tape pointer and the LDS address of S.rng_idx by value (no `this`: a Sim whose address
  // escapes to a call would have to live in memory).
  static __device__ __noinline__ void rng_refill_slow(uint32_t* tape_, uint32_t lds_rng_idx) {
    uint32_t* tape = (uint32_t*)(__attribute__((address_space(1))) uint32_t*)tape_;
    uint32_t* idx = (uint32_t*)(__attribute__((address_space(3))) uint32_t*)(uintptr_t)lds_rng_idx;
    const int lane = threadIdx.x & (WG - 1);
    for (int st = 0; st < 4; st++) {
      uint32_t p = idx[st];
      uint32_t blk = (p >> 16) & 1u, off = p & 0xffffu;
      if (off >= 624u) {  // block `blk` fully consumed: it becomes twist(other)
        uint32_t* base = tape + st * MUAVTA_RNG_WORDS;
        mt_twist(base + (blk ^ 1u) * 624u, base + blk * 624u);
        if (lane == 0) idx[st] = ((blk ^ 1u) << 16) | (off - 624u);
        __syncthreads();
      }
    }
  }
  // Lane (st, k) fetches raw word cursor+k of stream st at the START of the step (rng_prefetch_issue) and
  // parks it in the LDS window just before the first consumer (rng_prefetch_commit): the HBM/L2 latency
  // hides behind the action / movement phases, and lane 0's draws of this step come from LDS.
  DEV uint32_t rng_prefetch_issue() {
    uint32_t w = 0;
    if (lane < 32) {
      const int st = lane >> 3, k = lane & 7;
      const uint32_t p = S.rng_idx[st];
      const uint32_t blk = (p >> 16) & 1u, off = (p & 0xffffu) + (uint32_t)k;
      if (off < 1248u) {
        const uint32_t b = off >= 624u ? (blk ^ 1u) : blk;
        const uint32_t o = off >= 624u ? off - 624u : off;
        w = tape[st * MUAVTA_RNG_WORDS + b * 624u + o];
      }
    }
    return w;
  }
  DEV void rng_prefetch_commit(uint32_t w) {
    if (lane < 32) {
      const int st = lane >> 3, k = lane & 7;
      S.rng_win[st][k] = w;
      if (k == 0) S.rng_win_at[st] = S.rng_idx[st];  // no draw happens between issue and commit
    }
    lds_sync();
  }
  // All lanes: seed stream(s).  `scr` = LDS scratch of >= 2*624 words.
  DEV void rng_seed_pair(uint32_t* scr, int stA, uint64_t seedA, int stB, uint64_t seedB) {
    if (lane == 0) mt_seed(scr, (uint32_t)seedA, (uint32_t)(seedA >> 32), (seedA >> 32) ? 2 : 1);
    if (lane == 1 && stB >= 0) mt_seed(scr + 624, (uint32_t)seedB, (uint32_t)(seedB >> 32), (seedB >> 32) ? 2 : 1);
    __syncthreads();
    uint32_t* bA = tape + stA * MUAVTA_RNG_WORDS;
    mt_twist(scr, bA);
    mt_twist(bA, bA + 624);
    if (stB >= 0) {
      uint32_t* bB = tape + stB * MUAVTA_RNG_WORDS;
      mt_twist(scr + 624, bB);
      mt_twist(bB, bB + 624);
    }
    if (lane == 0) { S.rng_idx[stA] = 0; S.rng_win_at[stA] = 0x7fffffffu; if (stB >= 0) { S.rng_idx[stB] = 0; S.rng_win_at[stB] = 0x7fffffffu; } }
    __syncthreads();
  }

  // ---------------------------------------------------------------- references & queues (lane 0)
  DEV bool ref_valid(int id, int slot) const { return slot >= 0 && S.t_id[slot] == id; }
  DEV bool ref_retired(int id, int slot) const { return !ref_valid(id, slot) || S.t_status[slot] == 2; }
  DEV int head_id(int a) const { return S.a_qlen[a] > 0 ? S.a_qid[a][0] : 0; }
  DEV int queue_find(int a, int id) const {
    // every entry of the row is read at once (entries beyond the queue's length are stale but in bounds) and the first
    // match picked from a bit mask: one LDS round trip instead of one per entry
    const int n = S.a_qlen[a];
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < Q; k++) m |= (unsigned)(S.a_qid[a][k] == id) << k;
    m &= (1u << n) - 1u;
    return m ? __ffs((int)m) - 1 : -1;
  }
  DEV void queue_erase(int a, int k) {
    const int n = S.a_qlen[a];
    for (int i = k; i + 1 < n; i++) {
      S.a_qid[a][i] = S.a_qid[a][i + 1];
      S.a_qslot[a][i] = S.a_qslot[a][i + 1];
    }
    // the travel times live in the HBM record: up to four entries per pass with all the loads in front of the stores — a
    // load-then-store per entry is one memory round trip (~1 us) per entry on this serial path
    for (int i = k; i + 1 < n; i += 4) {
      const int r = n - 1 - i;
      double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
      if (r >= 1) t0 = C.a_qtime[a][i + 1];
      if (r >= 2) t1 = C.a_qtime[a][i + 2];
      if (r >= 3) t2 = C.a_qtime[a][i + 3];
      if (r >= 4) t3 = C.a_qtime[a][i + 4];
      if (r >= 1) C.a_qtime[a][i] = t0;
      if (r >= 2) C.a_qtime[a][i + 1] = t1;
      if (r >= 3) C.a_qtime[a][i + 2] = t2;
      if (r >= 4) C.a_qtime[a][i + 3] = t3;
    }
    S.a_qlen[a] = n - 1;
  }
  DEV bool escort_type(int t) const { return (P.escort_mask >> t) & 1u; }

  // Task.removeAgentCap (DroneEnvComponents.py:280-301).  The agent's own queue entry must already be gone.
  // initTime / doneTime are NOT maintained incrementally: the reference's upkeep keeps the invariants
  // initTime == min(details times), doneTime == max(details times) + duration (or -1, -1 when empty), and
  // details == the agents' queue entries, so refresh_task_times() rebuilds them per step in one parallel pass.
  DEV void remove_agent_cap(int s, int a) {
    if (S.t_status[s] == 2) return;
    for (int c = 0; c < 6; c++) C.t_alloc[c][s] -= S.a_caps[c][a];
    S.t_ndet[s] -= 1;
    S.times_dirty = 1; obs_static_clear();
  }
  // UAV.desAllocate (DroneEnvComponents.py:97-113) for a task id that IS in the queue at position k.
  DEV void des_allocate_at(int a, int k) {
    int id = S.a_qid[a][k], slot = S.a_qslot[a][k];
    queue_erase(a, k);
    qs().a_nft[a] = (double)tnow;
    qs().a_nfx[a] = S.a_px[a];
    qs().a_nfy[a] = S.a_py[a];
    S.a_commit[a] = 0;
    if (ref_valid(id, slot)) remove_agent_cap(slot, a);
  }
  DEV bool des_allocate(int a, int id) {
    int k = queue_find(a, id);
    if (k < 0 || id == 0) return false;
    des_allocate_at(a, k);
    return true;
  }
  // `for task in self.tasks: self.desAllocate(task)` over the list being mutated
  // (DroneEnvComponents.py:115-119,122-127): every other queued task survives.
  DEV void iterate_desallocate(int a) {
    int i = 0;
    while (i < S.a_qlen[a]) {
      des_allocate_at(a, i);
      if (S.a_qlen[a] == 0) break;  // python rebinds tasks=[idle]; the running iterator sees an empty list
      i++;
    }
  }
  DEV void desallocate_all(int a) { iterate_desallocate(a); S.a_commit[a] = 0; }
  DEV void out_of_service(int a) { S.a_state[a] = -1; S.a_commit[a] = 0; iterate_desallocate(a); }

  // UAV.taskDone (DroneEnvComponents.py:143-179)
  DEV bool task_done(int a, int id, int type) {
    if (S.a_qlen[a] == 0 || S.a_qid[a][0] != id) return false;
    queue_erase(a, 0);
    S.a_task_start[a] = -1;
    if (type == MUAVTA_ATT) {
      S.a_acap[a] -= 1;
      if (S.a_acap[a] <= 0) S.a_caps[MUAVTA_ATT][a] = 0;
    }
    while (S.a_qlen[a] > 0 && ref_retired(S.a_qid[a][0], S.a_qslot[a][0])) queue_erase(a, 0);
    if (S.a_qlen[a] == 0) {
      if (S.a_reeval[a]) { S.a_last_id[a] = -1; S.a_last_slot[a] = -1; S.a_reeval[a] = 0; }
      qs().a_nft[a] = 0;
      qs().a_nfx[a] = S.a_px[a];
      qs().a_nfy[a] = S.a_py[a];
      S.a_state[a] = 0;
    } else {
      S.a_state[a] = 1;
    }
    return true;
  }
  // UAV.allocate (DroneEnvComponents.py:55-95) for a real task (id != 0) in slot s
  DEV bool uav_allocate(int a, int s, double pre_time = -1.0) {
    int id = S.t_id[s];
    if (queue_find(a, id) >= 0 || S.t_status[s] == 2) return false;
    S.a_reeval[a] = 0;
    S.a_last_id[a] = -1;
    S.a_last_slot[a] = -1;
    double time_to_task = pre_time >= 0 ? pre_time : fdiv(norm2(qs().a_nfx[a] - S.t_px[s], qs().a_nfy[a] - S.t_py[s]), speed_of(S.a_type[a]));
    double start_time = (qs().a_nft[a] - (double)tnow) > 0 ? qs().a_nft[a] : (double)tnow;
    double dur = (double)task_duration(S.t_type[s]);
    double end_time = start_time + time_to_task + dur;
    int n = S.a_qlen[a];
    if (n == 0) {
      S.a_task_start[a] = -1;
      S.a_state[a] = 1;
    }
    if (n >= Q) { fail(MUAVTA_ERR_QUEUE); return false; }
    S.a_qid[a][n] = id;
    S.a_qslot[a][n] = s;
    C.a_qtime[a][n] = time_to_task;
    S.a_qlen[a] = n + 1;
    qs().a_nft[a] = end_time;
    qs().a_nfx[a] = S.t_px[s];
    qs().a_nfy[a] = S.t_py[s];
    // Task.addAgentCap (DroneEnvComponents.py:306-326); status != 2 checked above
    S.t_ndet[s] += 1;
    S.times_dirty = 1; obs_static_clear();
    for (int c = 0; c < 6; c++) C.t_alloc[c][s] += S.a_caps[c][a];
    S.t_status[s] = 1;
    return true;
  }

  // ---------------------------------------------------------------- tasks / events (lane 0)
  DEV void push_event(int tag, int arg) {
    int n = S.n_events;
    if (n >= E) { fail(MUAVTA_ERR_EVENTS); return; }
    S.ev_tag[n] = tag;
    S.ev_arg[n] = arg;
    S.n_events = n + 1;
  }
  // A retired slot may be recycled once no LIVE agent queues it (its type/position are still read by
  // the switch penalty and the expected-distance term, :852-861,:1219-1220).
  DEV bool slot_unreferenced(int id) const {
    for (int a = 0; a < P.n_agents; a++) {
      if (S.a_state[a] == -1) continue;
      for (int k = 0; k < S.a_qlen[a]; k++) if (S.a_qid[a][k] == id) return false;
    }
    return true;
  }
  // Everything a released slot leaves behind except its column of known bits: the caller clears those (serially
  // below, or one agent per lane in the end-of-step GC) AFTER this has read them.  agent_known_tasks keeps the ids
  // of retired tasks and the token builders read len(known_ids): a reveal that is still pending for the id gets the
  // knower set (_wps_process_reveals adds the id to every set).
  DEV void release_slot_record(int s) {
    int h = S.t_threat[s];
    if (h >= 0) { S.h_tflags[h] = S.t_flags[s] & (TF_DEADLINE | TF_COUNTED); S.h_tdeadline[h] = S.t_deadline[s]; }
    if (S.t_bucket[s] == 0) atomicAdd(&S.n_retired_empty_buckets, 1);
    const int id = S.t_id[s];
    S.t_id[s] = -1;
    // a reveal can only be pending while t < created_at + threat_delay (registered at creation, :1491-1501)
    const bool pending = P.share_knowledge && tnow <= S.t_created[s] + (P.threat_delay > 0 ? P.threat_delay : 0) + 1;
    if (!pending && !rel_log) return;
    unsigned long long knowers = 0;
    for (int a = 0; a < P.n_agents; a++) knowers |= (unsigned long long)((S.known[a][s >> 5] >> (s & 31)) & 1u) << a;
    if (pending)
      for (int k = 0; k < S.n_pending; k++)
        if (S.pend_slot[k] == s && S.pend_id[k] == id) S.pend_know[k] = (typename KnowMask<A>::type)knowers;
    if (rel_log) {  // facade only: the task's final record and who knew the id when it left the device
      const int k = atomicAdd(reinterpret_cast<int*>(rel_log), 1);
      if (k < T) {
        double* r = rel_log + 1 + (size_t)k * MUAVTA_REL_ROW;
        const int ty = S.t_type[s];
        r[0] = id; r[1] = (double)(uint32_t)knowers; r[2] = (double)(uint32_t)(knowers >> 32); r[3] = ty;
        r[4] = (S.t_flags[s] & TF_DEADLINE) ? S.t_deadline[s] : -1; r[5] = S.t_created[s]; r[6] = S.t_required[s];
        r[7] = (S.t_flags[s] & TF_ESCORT) ? 1 : 0; r[8] = S.t_ndet[s]; r[9] = S.t_prot_agent[s];
        r[10] = (S.t_flags[s] & TF_ELIGIBLE) ? (double)S.t_elig[s] : -1.0; r[11] = S.t_px[s]; r[12] = S.t_py[s];
        r[13] = qs().t_org[s]; r[14] = qs().t_done[s]; r[15] = C.t_init[s]; r[16] = C.t_dtime[s];
        for (int c = 0; c < 6; c++) { r[17 + c] = C.t_cur[c][s]; r[23 + c] = C.t_alloc[c][s]; }
      }
    }
  }
  DEV void release_slot(int s) {  // serial form (lane 0, on-demand reclaim)
    release_slot_record(s);
    // leave the slot clean for its next tenant: nobody knows it, it is on the free list
    const uint32_t bit = 1u << (s & 31);
    for (int a = 0; a < P.n_agents; a++) {
      const uint32_t old = S.known[a][s >> 5];
      if (old & bit) { S.known[a][s >> 5] = old & ~bit; S.a_gone[a] += 1; }
    }
    S.free_slots[s >> 5] |= bit;
  }
  // One list item -> S.act_*[n] (lane 0; index -> slot through the open list the previous observation returned, which no phase
  // before the end of the step rewrites).  Agent ids beyond the fleet are skipped (the host entry points reject them).
  DEV bool stage_item(int n, int a, int idx) {
    if (a >= P.n_agents) return false;
    if (idx < 0) idx += S.n_open;  // python negative indexing into last_tasks_info
    S.act_agent[n] = (i8)a;
    S.act_slot[n] = (idx >= 0 && idx < S.n_open) ? (i8)S.open_slot[idx] : (i8)-1;
    S.act_index[n] = (i16)(idx < -32768 ? -32768 : idx > 32767 ? 32767 : idx);
    return true;
  }
  // the next up-to-A items of a long list (wave-uniform result: how many were staged)
  DEV int stage_more() {
    lds_sync();
    if (lane == 0) {
      int n = 0;
      while (more_pos < more_cap && n < A) {
        const int a = more_agent[more_pos];
        if (a < 0) { more_pos = more_cap; break; }
        if (stage_item(n, a, more_index[more_pos])) n++;
        more_pos++;
      }
      S.n_act = n;
    }
    lds_sync();
    return S.n_act;
  }
  DEV bool pending_item_names_slot(int s) const {  // (lane 0) an item not staged yet names slot s
    if (!more_agent) return false;
    for (int k = more_pos; k < more_cap; k++) {
      if (more_agent[k] < 0) break;
      int idx = more_index[k];
      if (idx < 0) idx += S.n_open;
      if (idx >= 0 && idx < S.n_open && S.open_slot[idx] == s) return true;
    }
    return false;
  }
  DEV int reclaim_slot_serial() {
    for (int k = 0; k < S.n_order; k++) {
      int s = S.t_order[k];
      if (S.t_status[s] != 2 || !slot_unreferenced(S.t_id[s])) continue;
      // Outside a step (muavta_call: _create_escort_for / _sync_escorts on a full tile) a slot that last_tasks_info still lists is not
      // recycled: nothing rebuilds that list before the next step applies action indices through it, and its row would name the new
      // tenant.  Inside a step the list is only read before the first creation and rebuilt at the end.  No slot left: capacity flag.
      if (!in_step) { const int r = S.t_row[s]; if (r < S.n_open && (int)S.open_slot[r] == s) continue; }
      bool staged = false;  // actions still to be applied this step may name it (-> invalid-action penalty)
      for (int j = 0; j < S.n_act; j++) staged |= (S.act_slot[j] == s);
      if (staged || pending_item_names_slot(s)) continue;
      release_slot(s);
      for (int i = k; i + 1 < S.n_order; i++) S.t_order[i] = S.t_order[i + 1];
      S.n_order--;
      return s;
    }
    return -1;
  }
  // Task.__init__ (DroneEnvComponents.py:224-263) into a free slot; returns the slot or -1
  DEV int new_task(double x, double y, int type, double req) {
    int id = S.next_task_id++;  // _alloc_task_id (:325-328)
    if (id > 32000) { fail(MUAVTA_ERR_TASK_SLOTS); return -1; }  // ids are stored in 16 bits
    int s = -1;
    for (int w = 0; w < KW; w++) if (S.free_slots[w]) { s = (w << 5) + __ffs((int)S.free_slots[w]) - 1; break; }
    if (s >= T) s = -1;
    if (s < 0) s = reclaim_slot_serial();  // tile full mid-step: recycle a retired slot before the end-of-step GC
    if (s < 0) { fail(MUAVTA_ERR_TASK_SLOTS); return -1; }
    S.free_slots[s >> 5] &= ~(1u << (s & 31));
    S.t_id[s] = id;
    S.t_px[s] = x; S.t_py[s] = y;
    for (int c = 0; c < 6; c++) { C.t_cur[c][s] = 0; C.t_alloc[c][s] = 0; }
    C.t_cur[type][s] = req;
    obs_static_clear();
    qs().t_org[s] = req; qs().t_done[s] = 0;
    C.t_init[s] = -1; C.t_dtime[s] = -1;
    S.t_status[s] = 0; S.t_type[s] = type; S.t_created[s] = 0; S.t_deadline[s] = -1; S.t_required[s] = 0;
    S.t_flags[s] = 0; S.t_elig[s] = 0; S.t_threat[s] = -1;
    S.t_prot_agent[s] = -1; S.t_prot_id[s] = -1; S.t_prot_slot[s] = -1;
    S.t_ndet[s] = 0; S.t_bucket[s] = 0;
    S.t_order[S.n_order++] = s;
    return s;
  }
  DEV void know_all(int s) { for (int a = 0; a < P.n_agents; a++) S.known[a][s >> 5] |= 1u << (s & 31); S.t_flags[s] |= TF_KNOWN_ALL; }

  // _register_dynamic_task (:1491-1504)
  DEV void register_dynamic(int s) {
    if (P.hard_windows && !(S.t_flags[s] & TF_DEADLINE)) {
      S.t_flags[s] |= TF_DEADLINE;
      S.t_deadline[s] = tnow + P.window_length;
      S.n_windowed_tasks++;
    }
    if (P.threat_delay > 0 || P.sense_radius > 0) {
      int n = S.n_pending;
      if (n >= R) { fail(MUAVTA_ERR_PENDING); return; }
      S.pend_time[n] = tnow + (P.threat_delay > 0 ? P.threat_delay : 0);
      S.pend_id[n] = S.t_id[s];
      S.pend_slot[n] = (uint8_t)s;
      S.pend_know[n] = 0;
      S.n_pending = n + 1;
    } else {
      know_all(s);
    }
  }
  // _wps_mark_window_outcome (:1543-1555) on explicit (flags, deadline) storage
  DEV void mark_outcome(u8& flags, int deadline, bool success) {
    if (!(flags & TF_DEADLINE)) return;
    if (flags & TF_COUNTED) return;
    flags |= TF_COUNTED;
    if (success && tnow <= deadline) { S.n_on_time++; S.F_Reward += P.on_time_bonus; }
    else { S.n_missed_windows++; S.F_Reward -= P.miss_penalty; }
  }
  DEV void mark_outcome_slot(int s, bool success) { mark_outcome(S.t_flags[s], S.t_deadline[s], success); }

  DEV bool counts_for_mission_done(int s) const {  // :1878-1886
    if (S.t_flags[s] & TF_ESCORT) return true;
    int ty = S.t_type[s];
    if (ty == MUAVTA_DET || ty == MUAVTA_HOLD) return true;
    return S.t_status[s] == 2;
  }
  DEV bool all_mission_done() const {  // freed slots are retired tasks, which always count as done
    for (int k = 0; k < S.n_order; k++) if (!counts_for_mission_done(S.t_order[k])) return false;
    return true;
  }
  DEV bool action_valid(int a, int s) const {  // _is_task_action_valid (:341-363)
    if (S.t_status[s] == 2) return false;
    if (S.a_qlen[a] > 0 && S.a_qid[a][0] == S.t_id[s]) return true;
    if ((S.t_flags[s] & TF_ELIGIBLE) && !((S.t_elig[s] >> S.a_type[a]) & 1u)) return false;
    int ty = S.t_type[s];
    if (P.capability_mask && S.a_caps[ty][a] <= 0) return false;
    if (P.saturate_mask && C.t_alloc[ty][s] >= qs().t_org[s]) return false;
    return true;
  }

  // ---------------------------------------------------------------- escorts (lane 0)
  DEV bool has_escort(int recon) const { return (S.esc_mask >> recon) & 1ull; }  // recon in _escort_by_recon, in one LDS read
  DEV int escort_lookup(int recon) const {
    for (int k = 0; k < S.n_escorts; k++) if (S.esc_agent[k] == recon) return k;
    return -1;
  }
  DEV void create_escort_for(int recon, int rec_slot) {  // _create_escort_for (:1888-1917)
    if (!P.escort_enabled) return;
    if (has_escort(recon)) return;
    int s = new_task(S.a_px[recon], S.a_py[recon], MUAVTA_DEF, P.escort_requirement);
    if (s < 0) return;
    S.t_flags[s] |= TF_ESCORT | TF_ELIGIBLE;
    S.t_elig[s] = P.escort_mask;
    S.t_prot_agent[s] = recon;
    S.t_prot_id[s] = rec_slot >= 0 ? (int)S.t_id[rec_slot] : -1;  // rec_slot < 0: protected_task = None (out-of-step callers only)
    S.t_prot_slot[s] = rec_slot;
    S.t_required[s] = P.escort_required_agents;
    S.t_created[s] = tnow;
    register_dynamic(s);
    int n = S.n_escorts;
    if (n >= A) { fail(MUAVTA_ERR_ESCORTS); return; }
    S.esc_mask |= 1ull << recon;
    S.esc_agent[n] = recon; S.esc_id[n] = S.t_id[s]; S.esc_slot[n] = s;
    S.esc_pid[n] = S.t_prot_id[s]; S.esc_pslot[n] = rec_slot;
    S.n_escorts = n + 1;
    S.escort_requests++;
    push_event(MUAVTA_EV_ESCORT_CREATED, S.t_id[s]);
    push_event(MUAVTA_EV_RESET_ALLOCATION, MUAVTA_DEF);
    S.pending_reset = 1;
  }
  // `holders`: superset of the agents that queue the escort task (callers that ran a wave-wide queue scan pass the
  // exact set; the default visits every agent, as the reference does)
  DEV void retire_escort_entry(int k, bool failed, unsigned long long holders = ~0ull) {  // _retire_escort (:1938-1950) for map entry k
    int s = S.esc_slot[k], id = S.esc_id[k];
    // an escort that expired by its hard window keeps its map entry forever (status == 2 -> early return)
    if (ref_retired(id, s)) return;
    // _release_escort_agents (:1919-1936): agent.desAllocate(escort) for every holder, in agents_obj order.  Every holder's
    // Task.removeAgentCap works on the SAME task: allocatedReqs is fetched once, the holders' capabilities are subtracted in order
    // in registers, and the result is stored once (a read-modify-write per holder is a memory round trip per holder).
    double al[6] = {0, 0, 0, 0, 0, 0};
    bool loaded = false;
    int nd = 0;
    for (int a = 0; a < P.n_agents; a++) {
      if (!((holders >> a) & 1ull)) continue;
      if (S.a_state[a] == -1 || S.a_qlen[a] == 0) continue;
      const int kq = queue_find(a, id);
      if (kq < 0 || id == 0) continue;
      {  // des_allocate_at(a, kq) with the task side deferred
        const int slot = S.a_qslot[a][kq];
        queue_erase(a, kq);
        qs().a_nft[a] = (double)tnow;
        qs().a_nfx[a] = S.a_px[a];
        qs().a_nfy[a] = S.a_py[a];
        S.a_commit[a] = 0;
        if (ref_valid(id, slot) && S.t_status[slot] != 2) {  // (slot == s: the one slot that holds this id)
          if (!loaded) {
#pragma unroll
            for (int c = 0; c < 6; c++) al[c] = C.t_alloc[c][s];
            loaded = true;
          }
#pragma unroll
          for (int c = 0; c < 6; c++) al[c] -= S.a_caps[c][a];
          nd++;
        }
      }
      if (S.a_qlen[a] == 0) {
        S.a_state[a] = 0;
        S.a_commit[a] = 0;
        qs().a_nft[a] = (double)tnow;
        qs().a_nfx[a] = S.a_px[a];
        qs().a_nfy[a] = S.a_py[a];
      }
    }
    if (loaded) {
#pragma unroll
      for (int c = 0; c < 6; c++) C.t_alloc[c][s] = al[c];
      S.t_ndet[s] -= nd;
      S.times_dirty = 1; obs_static_clear();
    }
    S.t_status[s] = 2;
    int recon = S.t_prot_agent[s];
    int kk = escort_lookup(recon);
    if (kk >= 0) {
      S.esc_mask &= ~(1ull << recon);
      for (int i = kk; i + 1 < S.n_escorts; i++) {
        S.esc_agent[i] = S.esc_agent[i + 1]; S.esc_id[i] = S.esc_id[i + 1]; S.esc_slot[i] = S.esc_slot[i + 1];
        S.esc_pid[i] = S.esc_pid[i + 1]; S.esc_pslot[i] = S.esc_pslot[i + 1];
      }
      S.n_escorts--;
    }
    if (failed) S.escort_failed++; else S.escort_completed++;
    push_event(MUAVTA_EV_ESCORT_RETIRED, id);
  }
  // _retire_escort (:1938-1950) of the escort that protects `recon`, with the whole wave (all lanes call it, uniform arguments):
  // every holder erases ITS OWN queue entry (lane = agent: agent.desAllocate touches the agent's own fields), Task.removeAgentCap
  // of the holders runs on one lane in agents_obj order (one fetch of allocatedReqs, one store), and the map entries behind the
  // retired one move up one lane each.  On lane 0 the same work is a chain of dependent LDS round trips: the map look-up, a pass
  // over the fleet, a five-array shift of the map (12-17 thousand cycles per retirement on the 24-agent tile).
  DEV void retire_escort_coop(int recon, bool failed) {
    cold_sync();  // the holders' lanes read queue-time rows other lanes wrote in earlier phases
    const int ne = __builtin_amdgcn_readfirstlane(S.n_escorts), nA = P.n_agents;
    const unsigned long long emk = __builtin_amdgcn_ballot_w64(lane < ne && S.esc_agent[lane < ne ? lane : 0] == recon);  // escort_lookup(recon)
    if (emk == 0ull) return;
    const int kk = __ffsll((long long)emk) - 1;
    const int s = S.esc_slot[kk], id = S.esc_id[kk];
    // an escort that expired by its hard window keeps its map entry forever (status == 2 -> early return)
    if (ref_retired(id, s)) return;
    // _release_escort_agents (:1919-1936)
    bool capped = false;
    if (lane < nA && S.a_state[lane] != -1 && S.a_qlen[lane] > 0) {
      const int a = lane;
      const int kq = queue_find(a, id);
      if (kq >= 0) {
        const int slot = S.a_qslot[a][kq];
        queue_erase(a, kq);
        qs().a_nft[a] = (double)tnow;
        qs().a_nfx[a] = S.a_px[a];
        qs().a_nfy[a] = S.a_py[a];
        S.a_commit[a] = 0;
        capped = ref_valid(id, slot) && S.t_status[slot] != 2;
        if (S.a_qlen[a] == 0) S.a_state[a] = 0;  // (commit_until / next_free_* were just set to the same values)
      }
    }
    const unsigned long long cm = __builtin_amdgcn_ballot_w64(capped);
    // the map entries behind kk, one per lane, before anybody rewrites them
    const int me = lane < ne ? lane : 0;
    const int m_agent = S.esc_agent[me], m_id = S.esc_id[me], m_slot = S.esc_slot[me], m_pid = S.esc_pid[me], m_pslot = S.esc_pslot[me];
    lds_sync();
    if (lane == 0) {
      if (cm) {  // Task.removeAgentCap of every holder, agents ascending
        double al[6];
#pragma unroll
        for (int c = 0; c < 6; c++) al[c] = C.t_alloc[c][s];
        for (unsigned long long m = cm; m; m &= m - 1ull) {
          const int a = __ffsll((long long)m) - 1;
#pragma unroll
          for (int c = 0; c < 6; c++) al[c] -= S.a_caps[c][a];
        }
#pragma unroll
        for (int c = 0; c < 6; c++) C.t_alloc[c][s] = al[c];
        S.t_ndet[s] -= __popcll(cm);
        S.times_dirty = 1; obs_static_clear();
      }
      S.t_status[s] = 2;
      S.esc_mask &= ~(1ull << recon);  // (== S.t_prot_agent[s]: the map is keyed by the protected UAV)
      S.n_escorts = ne - 1;
      if (failed) S.escort_failed++; else S.escort_completed++;
      push_event(MUAVTA_EV_ESCORT_RETIRED, id);
    }
    if (lane > kk && lane < ne) {
      S.esc_agent[lane - 1] = m_agent; S.esc_id[lane - 1] = m_id; S.esc_slot[lane - 1] = m_slot;
      S.esc_pid[lane - 1] = m_pid; S.esc_pslot[lane - 1] = m_pslot;
    }
    cold_sync();
  }
  DEV void retire_escort_for(int recon, bool failed) {  // :1952-1957
    int k = escort_lookup(recon);
    if (k >= 0) retire_escort_entry(k, failed);
  }
  // _escort_fighters_near (:1746-1764): nearest-first list into out[], returns count.
  // Stable insertion sort == python's sort(key=dist) on (dist, agent) pairs built in id order.
  template <class Out>
  DEV int escort_fighters_near(int prot, double radius, Out* out, double* outd) {
    int k = escort_lookup(prot);
    if (k < 0 || ref_retired(S.esc_id[k], S.esc_slot[k])) return 0;
    int eid = S.esc_id[k];
    int n = 0;
    for (int a = 0; a < P.n_agents; a++) {
      if (S.a_state[a] == -1 || !escort_type(S.a_type[a])) continue;
      if (S.a_qlen[a] == 0 || S.a_qid[a][0] != eid) continue;
      double d = norm2(S.a_px[a] - S.a_px[prot], S.a_py[a] - S.a_py[prot]);
      if (d <= radius) {
        int i = n;
        while (i > 0 && outd[i - 1] > d) { outd[i] = outd[i - 1]; out[i] = out[i - 1]; i--; }
        outd[i] = d; out[i] = (Out)a;
        n++;
      }
    }
    return n;
  }
  // closest_escort() for ONE PROTECTED UAV PER LANE (`need` / `prot` vary by lane; all lanes call it).  The serial form walks the
  // escort map and then the whole fleet per call — per threat lane that was n_agents LDS round trips and a square root whenever
  // any lane's filter passed.  Here lane k holds map entry k and lane a holds agent a; both are broadcast with v_readlane in
  // uniform loops, and the distance is only evaluated for an agent that heads some needed escort task.  Same arithmetic, same
  // order (agents ascending: the FIRST at the minimal distance wins).
  DEV void closest_escort_lanes(bool need, int prot, double radius, int& best, int& count) {
    best = -1; count = 0;
    if (__builtin_amdgcn_ballot_w64(need) == 0ull) return;
    const int ne = __builtin_amdgcn_readfirstlane(S.n_escorts), nA = P.n_agents;
    int eid = -1;  // the live escort task of this lane's protected UAV (escort_lookup + ref_retired)
    {
      const int ke = lane < ne ? lane : 0;
      const int ea = lane < ne ? (int)S.esc_agent[ke] : -1, ei = S.esc_id[ke], es = S.esc_slot[ke];
      const int elive = (lane < ne && !ref_retired(ei, es)) ? ei : -1;
      bool found = false;
      for (int k = 0; k < ne; k++) {
        const int ak = __builtin_amdgcn_readlane(ea, k), ik = __builtin_amdgcn_readlane(elive, k);
        if (need && !found && prot == ak) { eid = ik; found = true; }
      }
    }
    const bool want = need && eid >= 0;
    if (__builtin_amdgcn_ballot_w64(want) == 0ull) return;
    const int fa = lane < nA ? lane : 0;
    const bool fok = lane < nA && S.a_state[fa] != -1 && escort_type(S.a_type[fa]) && S.a_qlen[fa] > 0;
    const int fhead = fok ? (int)S.a_qid[fa][0] : -1;
    const double fx = S.a_px[fa], fy = S.a_py[fa];
    const double px = S.a_px[prot], py = S.a_py[prot];
    double bd = 0;
    for (int a = 0; a < nA; a++) {
      const int ha = __builtin_amdgcn_readlane(fhead, a);
      if (ha < 0) continue;
      const bool mt = want && eid == ha;
      if (__builtin_amdgcn_ballot_w64(mt) == 0ull) continue;
      const double d = norm2(readlane_f64(fx, a) - px, readlane_f64(fy, a) - py);
      if (mt && d <= radius) {
        if (count == 0 || d < bd) { bd = d; best = a; }
        count++;
      }
    }
  }
  DEV int closest_escort(int prot, double radius, int* count) {  // first element + count of the list above
    int k = escort_lookup(prot);
    *count = 0;
    if (k < 0 || ref_retired(S.esc_id[k], S.esc_slot[k])) return -1;
    int eid = S.esc_id[k];
    int best = -1, n = 0;
    double bd = 0;
    for (int a = 0; a < P.n_agents; a++) {
      if (S.a_state[a] == -1 || !escort_type(S.a_type[a])) continue;
      if (S.a_qlen[a] == 0 || S.a_qid[a][0] != eid) continue;
      double d = norm2(S.a_px[a] - S.a_px[prot], S.a_py[a] - S.a_py[prot]);
      if (d <= radius) { if (n == 0 || d < bd) { bd = d; best = a; } n++; }
    }
    *count = n;
    return best;
  }

  // ---------------------------------------------------------------- releaseAllTasks (:1442-1480)
  DEV void release_all_tasks(int for_type) {
    int ft = for_type < 0 ? for_type + 6 : for_type;  // python negative index -> caps[-1] == Det
    uint32_t avail = 0;
    for (int a = 0; a < P.n_agents; a++) {
      if (S.a_caps[ft][a] > 0 && S.a_state[a] != -1) {
        S.a_reeval[a] = 1;  // len(agent.tasks) > 0 always holds in python ([task_idle] counts)
        if (S.a_qlen[a] > 0) { S.a_last_id[a] = S.a_qid[a][0]; S.a_last_slot[a] = S.a_qslot[a][0]; }
        else { S.a_last_id[a] = 0; S.a_last_slot[a] = -1; }
        desallocate_all(a);
        avail |= 1u << S.a_type[a];
      }
    }
    for (int k = 0; k < S.n_order; k++) {
      int s = S.t_order[k];
      if (S.t_status[s] != 2 && S.t_type[s] == for_type) {
        double cum = 0;
        for (int ty = 0; ty < 7; ty++) if ((avail >> ty) & 1u) cum += CAP_TABLE[ty][for_type];
        if (cum == 0) {
          S.t_status[s] = 2;
          if (!(S.t_flags[s] & TF_REACHED)) {
            S.t_flags[s] |= TF_REACHED;
            S.n_reached++;
            if (S.n_reached == P.n_tasks) S.conclusion_time = tnow;
          }
        } else {
          S.t_status[s] = 0;
          S.t_bucket[s] = 0;
        }
      }
    }
  }

  // releaseAllTasks with the whole wave: per-agent flags by the agent's lane, queue teardown (shared f64
  // accumulators on the tasks -> agents_obj order) by lane 0 for the agents that actually queue something,
  // then one task per lane for the status/bucket reset.  All lanes must call it.
  // returns whether the call changed any state: a repeat of the same event right after a call that changed nothing
  // is a no-op (f(S) == S), which the drain loop uses to skip the tail of a burst's identical Reset_Allocation events
  DEV bool release_all_tasks_coop(int for_type) {
    PROF_COUNT(53, 1000);
    const int ft = for_type < 0 ? for_type + 6 : for_type;  // python negative index -> caps[-1] == Det
    const int a = lane;
    bool match = false, busy = false, chg = false;
    if (a < P.n_agents && S.a_caps[ft][a] > 0 && S.a_state[a] != -1) {
      match = true;
      chg = S.a_reeval[a] != 1;
      S.a_reeval[a] = 1;  // len(agent.tasks) > 0 always holds in python ([task_idle] counts)
      if (S.a_qlen[a] > 0) { S.a_last_id[a] = S.a_qid[a][0]; S.a_last_slot[a] = S.a_qslot[a][0]; busy = true; chg = true; }
      else {  // desallocateAll on [task_idle]
        chg |= S.a_last_id[a] != 0 || S.a_last_slot[a] != -1 || S.a_commit[a] != 0;
        S.a_last_id[a] = 0; S.a_last_slot[a] = -1; S.a_commit[a] = 0;
      }
    }
    unsigned long long bm = __ballot(busy);
    if (bm) cold_sync();  // the queue teardown below reads / rewrites EnvCold rows other lanes wrote in earlier phases
    uint32_t avail = 0;
    {  // available_agents: set of type indices of the matched agents
      const int ty = a < P.n_agents ? S.a_type[a] : 0;
      // (ballots straight from a compare: the ballot of a combined predicate costs a v_cndmask + v_cmp to materialise it)
      const int tym = match ? ty : -1;
#pragma unroll
      for (int t = 0; t <= MUAVTA_F2; t++) if (__builtin_amdgcn_ballot_w64(tym == t) != 0ull) avail |= 1u << t;
    }
    lds_sync();
    if (Q <= 16) {
      constexpr int HQ = (Q + 1) / 2;
      // desallocateAll of every busy agent at once.  `for task in self.tasks: self.desAllocate(task)` over the list
      // being mutated drops the queue entries at even positions and keeps the odd ones (iterate_desallocate).
      // Agent side (lane = agent): own queue compaction + next-free fields.  Task side (lane = slot): Task.removeAgentCap
      // of the dropped entries in the reference's order (agents ascending), so allocatedReqs sees the same f64 sequence.
      int rid[HQ], rsl[HQ];
#pragma unroll
      for (int i = 0; i < HQ; i++) { rid[i] = -1; rsl[i] = -1; }  // dropped (task id, slot) of this lane's agent
      if (busy) {
        const int n = S.a_qlen[a];
#pragma unroll
        for (int i = 0; i < HQ; i++) if (2 * i < n) { rid[i] = S.a_qid[a][2 * i]; rsl[i] = S.a_qslot[a][2 * i]; }
        int kid[HQ], ksl[HQ]; double ktm[HQ];
#pragma unroll
        for (int i = 0; i < HQ; i++) if (2 * i + 1 < n) { kid[i] = S.a_qid[a][2 * i + 1]; ksl[i] = S.a_qslot[a][2 * i + 1]; ktm[i] = C.a_qtime[a][2 * i + 1]; }
#pragma unroll
        for (int i = 0; i < HQ; i++) if (2 * i + 1 < n) { S.a_qid[a][i] = kid[i]; S.a_qslot[a][i] = ksl[i]; C.a_qtime[a][i] = ktm[i]; }
        S.a_qlen[a] = n >> 1;
        qs().a_nft[a] = (double)tnow; qs().a_nfx[a] = S.a_px[a]; qs().a_nfy[a] = S.a_py[a];
        S.a_commit[a] = 0;
      }
      // Task side: who dropped which slot travels through per-slot agent masks in the scratch tile (LDS atomics), and a slot
      // lane walks only the agents that dropped IT, in ascending order — not every busy agent's HQ dropped entries (r2: 2 x HQ
      // v_readlane + compares per busy agent and pass, ~600 VALU per call on the 24-agent tile).
      typedef typename BucketMask<A>::type DropMask;
      DropMask* dropm = reinterpret_cast<DropMask*>(X.cost);
      static_assert(sizeof(DropMask) * T <= sizeof(double) * Scratch<TL>::COSTN, "drop masks overflow the scratch cost tile");
      for (int sl = lane; sl < T; sl += WG) dropm[sl] = 0;
      lds_sync();
#pragma unroll
      for (int i = 0; i < HQ; i++)
        if (rsl[i] >= 0 && S.t_id[rsl[i]] == rid[i]) atomicOr(&dropm[rsl[i]], (DropMask)1 << a);  // (a live reference: the slot still holds that id)
      lds_sync();
      for (int base = 0; base < T; base += WG) {
        const int sl = base + lane;
        DropMask m = sl < T ? dropm[sl] : (DropMask)0;
        if (m != 0 && S.t_status[sl] == 2) m = 0;  // removeAgentCap ignores concluded tasks
        if (m != 0) {
          double al[6];
#pragma unroll
          for (int c = 0; c < 6; c++) al[c] = C.t_alloc[c][sl];
          int nd = 0;
          for (; m != 0; m &= m - 1) {
            const int b = (sizeof(DropMask) > 4 ? __ffsll((long long)m) : __ffs((int)m)) - 1;
#pragma unroll
            for (int c = 0; c < 6; c++) al[c] -= S.a_caps[c][b];
            nd++;
          }
#pragma unroll
          for (int c = 0; c < 6; c++) C.t_alloc[c][sl] = al[c];
          S.t_ndet[sl] -= nd;
          S.times_dirty = 1; obs_static_clear();
        }
      }
    } else if (lane == 0) {
      while (bm) {
        const int b = __ffsll((long long)bm) - 1;
        bm &= bm - 1ull;
        desallocate_all(b);
      }
    }
    cold_sync();
    if (for_type < 0) return __ballot(chg) != 0ull;  // no task has typeIdx -1
    double cum = 0;
    for (int ty = 0; ty < 7; ty++) if ((avail >> ty) & 1u) cum += CAP_TABLE[ty][for_type];
    bool dead_end = false;
    for (int k = lane; k < S.n_order; k += WG) {
      const int s = S.t_order[k];
      if (S.t_status[s] != 2 && S.t_type[s] == for_type) {
        if (cum == 0) dead_end = true;
        else { chg |= S.t_status[s] != 0 || S.t_bucket[s] != 0; S.t_status[s] = 0; S.t_bucket[s] = 0; }
      }
    }
    chg |= dead_end;
    if (__ballot(dead_end) != 0ull) {  // nobody left who can do this type: retire the tasks, in id order (:1466-1476)
      lds_sync();
      if (lane == 0) {
        for (int k = 0; k < S.n_order; k++) {
          const int s = S.t_order[k];
          if (S.t_status[s] != 2 && S.t_type[s] == for_type) {
            S.t_status[s] = 2;
            if (!(S.t_flags[s] & TF_REACHED)) {
              S.t_flags[s] |= TF_REACHED;
              S.n_reached++;
              if (S.n_reached == P.n_tasks) S.conclusion_time = tnow;
            }
          }
        }
      }
    }
    lds_sync();
    return __ballot(chg) != 0ull;
  }

  // ---------------------------------------------------------------- geometry helpers
  DEV void norm_vector(double& x, double& y) {  // EnvUtils.norm_vector (MultiDroneEnvUtils.py:168-177)
    double m = norm2(x, y);
    if (m == 0) { x = 0; y = 0; return; }
    const double r = frcp_nr(m);  // (m > 0: a norm of finite coordinates)
    x = fdiv_r(x, m, r); y = fdiv_r(y, m, r);
  }
  // core_sim SimCore::avoid_obstacles (core_sim/src/sim_core.rs:25-59); Rust `%` == fmod
  DEV void avoid_obstacles(double px, double py, double mx, double my, double& ax, double& ay) {
    ax = 0.0; ay = 0.0;
    const double PI = 3.14159265358979323846;
    for (int o = 0; o < P.num_obstacles; o++) {
      double dx = C.obst[o][0] - px, dy = C.obst[o][1] - py;
      double d_zone = sqrt(dx * dx + dy * dy) - C.obst[o][2];
      if (d_zone < 40.0) {
        double nx = dx / d_zone, ny = dy / d_zone;
        double force = libm_log(fmax(1.05, d_zone));  // f64::ln of the reference = the host libm's log, bit for bit (libm_log above)
        force = 0.5 / (1.0 - force);
        double ang = atan2(my, mx) - atan2(dy, dx);  // (only the SIGN of the wrapped angle is used: an ulp of atan2 matters on a set of measure zero)
        ang = fmod(ang + PI, 2.0 * PI) - PI;
        double rx, ry;
        if (ang > 0.0) { rx = ny; ry = -nx; } else { rx = -ny; ry = nx; }
        ax += rx * force;
        ay += ry * force;
      }
    }
  }
  // random_position (:1371-1410)
  DEV void random_position(int st, double min_distance, double own_range, bool contact_line, int area, bool check_obs,
                           double& ox, double& oy) {
    double limit_line = contact_line ? CONTACT_LINE : 0;
    for (int tries = 0; tries < 100; tries++) {
      double x, y;
      if (area >= 0) {
        double tlx = qs().area[area][0], tly = qs().area[area][1], w = qs().area[area][2];
        x = uniform(st, tlx, tlx + w);
        y = uniform(st, tly, tly + w);
      } else {
        x = uniform(st, own_range + min_distance, AREA_W - own_range - min_distance);
        y = uniform(st, own_range + min_distance,
                    AREA_H - own_range - min_distance - ((limit_line != 0) ? (AREA_H - limit_line) : 0));
      }
      bool valid = true;
      if (check_obs) {
        for (int o = 0; o < P.num_obstacles && o < 8; o++) {
          if (C.obst[o][2] < 0) break;  // not created yet
          double d = norm2(x - C.obst[o][0], y - C.obst[o][1]) - own_range;
          if (d < C.obst[o][2] + min_distance) { valid = false; break; }
        }
      }
      if (valid) { ox = x; oy = y; return; }
    }
    fail(MUAVTA_ERR_POSITION);
    ox = 0; oy = 0;
  }
  // get_closest_agent with one agent per lane (all lanes call it): the serial scan takes `d < min` strictly, i.e. the FIRST agent
  // at the minimal distance — here the lowest lane among those equal to the wave-wide minimum.  On lane 0 the scan is a chain of
  // n_agents LDS round trips and square roots (16,000 cycles per failed engagement on the 64-agent tile).
  DEV int closest_agent_coop(double x, double y) {
    const double INF = __builtin_huge_val();
    const int a = lane < P.n_agents ? lane : 0;
    const int st = S.a_state[a], ty = S.a_type[a];
    const bool valid = lane < P.n_agents && st != -1 && st != 4;
    const double d = norm2(S.a_px[a] - x, S.a_py[a] - y);
    const bool fighter = is_fighter(ty);
    const double dW = (valid && !fighter) ? d : INF;
    const double mW = wave_min_first(dW, A);  // (A: compile-time — agent lanes only)
    if (__double2hiint(mW) != 0x7ff00000) return __ffsll((long long)__builtin_amdgcn_ballot_w64(dW == mW)) - 1;
    const double dF = (valid && fighter) ? d : INF;
    const double mF = wave_min_first(dF, A);
    if (__double2hiint(mF) != 0x7ff00000) return __ffsll((long long)__builtin_amdgcn_ballot_w64(dF == mF)) - 1;
    return -1;
  }
  // the retarget an engagement replayed on lane 0 asked for (handle_threat_engagement): all lanes, uniform
  DEV void resolve_retarget() {
    const int h = __builtin_amdgcn_readfirstlane(X.retarget_h);
    if (h < 0) return;
    const int tgt = closest_agent_coop(S.h_px[h], S.h_py[h]);
    if (lane == 0) { S.h_target[h] = tgt; S.h_mission[h] = tgt; X.retarget_h = -1; }
    lds_sync();
  }
  DEV int closest_agent(double x, double y) const {  // get_closest_agent (:1691-1723)
    double minF = __builtin_huge_val(), minW = __builtin_huge_val();
    int cF = -1, cW = -1;
    for (int a = 0; a < P.n_agents; a++) {
      int st = S.a_state[a];
      if (st != -1 && st != 4) {
        double d = norm2(S.a_px[a] - x, S.a_py[a] - y);
        if (is_fighter(S.a_type[a])) { if (d < minF) { minF = d; cF = a; } }
        else { if (d < minW) { minW = d; cW = a; } }
      }
    }
    return cW >= 0 ? cW : cF;
  }

  // ====================================================================================================
  // reset (:522-762).  All lanes enter; RNG seeding is cooperative, the rest runs on lane 0.
  // ====================================================================================================
  // reset-time RNG setup for one stream whose seeded MT state (k_seed: CPython's init_by_array, one LANE per stream there —
  // the recurrence is serial) sits in HBM: block0 = twist(seeded) -> `b0`, block1 = twist(block0) -> `b1` (both LDS); both
  // go to the HBM tape with coalesced stores and the head of block0 to the reset window.  Ordered by lds_sync().
  static constexpr int RESET_WIN = 160;  // words per stream staged for the reset (it draws ~100; a draw beyond the window reads the HBM tape)
  DEV void reset_stream(int st, const uint32_t* seeded, uint32_t* b0, uint32_t* b1, uint32_t* win) {
    mt_twist_lds(seeded, b0);
    mt_twist_lds(b0, b1);
    uint32_t* t = tape + st * MUAVTA_RNG_WORDS;
    for (int k = lane; k < 624; k += WG) { t[k] = b0[k]; t[624 + k] = b1[k]; }
    for (int k = lane; k < RESET_WIN; k += WG) win[st * RESET_WIN + k] = b0[k];
    lds_sync();
  }
  // `seeded`: this env's four init_by_array states [4][624] from k_seed
  DEV void reset(uint64_t seed, const uint32_t* seeded) {
    (void)seed;
    // RNG first.  The (not yet initialised) arrays of the state blob double as scratch for two 624-word MT buffers; the
    // stream cursors and the other scalars live beyond them.
    static_assert(offsetof(State, pending_reset) >= 2 * 624 * 4 && offsetof(State, rng_idx) >= 2 * 624 * 4,
                  "MT buffers overlap the scalars that are live while the RNG is set up (error word, stream cursors)");
    static_assert(sizeof(Scratch<TL>) >= 4 * RESET_WIN * 4, "scratch tile too small for the reset RNG window");
    tnow = 0;
    uint32_t* T0 = reinterpret_cast<uint32_t*>(&S);
    uint32_t* T1 = T0 + 624;
    uint32_t* win = reinterpret_cast<uint32_t*>(&X);
    if (lane < 4) { S.rng_idx[lane] = 0; S.rng_win_at[lane] = 0; }
    PROF(39);
    if (lane == 0) S.error = 0;
    lds_sync();
    PROF(42);
    reset_stream(ST_AGENT, seeded + ST_AGENT * 624, T0, T1, win);
    PROF(43);
    win_ptr = win; win_len = RESET_WIN; win_stride = RESET_WIN;
    if (lane == 0) {  // :535-538 — the three stream seeds (rndObsGen, rndTgtGen, rndMissionGen): k_seed drew the same values
      (void)randint(ST_AGENT, 0, INT64_MAX);
      (void)randint(ST_AGENT, 0, INT64_MAX);
      (void)randint(ST_AGENT, 0, INT64_MAX);
    }
    lds_sync();
    const uint32_t agent_cursor = S.rng_idx[ST_AGENT];
    const int err0 = S.error;
    lds_sync();
    PROF(44);
    reset_stream(ST_TGT, seeded + ST_TGT * 624, T0, T1, win);
    reset_stream(ST_MISSION, seeded + ST_MISSION * 624, T0, T1, win);
    if (P.num_obstacles > 0) reset_stream(ST_OBS, seeded + ST_OBS * 624, T0, T1, win);
    PROF(45);
    // now the blob itself: zero it, restore the cursors, build the episode
    {
      uint32_t* w = reinterpret_cast<uint32_t*>(&S);
      for (int i = lane; i < (int)(sizeof(State) / 4); i += WG) w[i] = 0;
    }
    lds_sync();
    if (lane == 0) { S.rng_idx[ST_AGENT] = agent_cursor; S.error = err0; S.obs_rows = -1; }
    lds_sync();
    PROF(46);
    if (lane == 0) reset_serial();
    lds_sync();
    {  // static / initial tasks are known to everyone (:757-758): every agent lane ORs the slots' mask into its row
      const int no = S.n_order;
      for (int w = 0; w < KW; w++) {
        uint32_t m = 0;
        for (int k = 0; k < no; k++) { const int sl = S.t_order[k]; if ((sl >> 5) == w) m |= 1u << (sl & 31); }
        if (lane < P.n_agents) S.known[lane][w] |= m;
      }
      for (int k = lane; k < no; k += WG) S.t_flags[S.t_order[k]] |= TF_KNOWN_ALL;
    }
    lds_sync();
    PROF(47);
    win_ptr = &S.rng_win[0][0]; win_len = 8; win_stride = 8;
    if (lane < 4) S.rng_win_at[lane] = 0x7fffffffu;  // the small window is (re)filled at the next step boundary
    cold_sync();  // obstacles and the initial requirement vectors are read by every lane from here on
    finish_step_parallel(false);
  }

  DEV void reset_serial() {
    const int nA = P.n_agents;
    S.conclusion_time = P.max_time_steps + 1;
    S.next_task_id = 1;
    S.last_plan_step = -1000000000;
    for (int k = 0; k < T; k++) S.t_id[k] = -1;
    const int cap = P.slot_cap > 0 && P.slot_cap < T ? P.slot_cap : T;  // (a requested tile_tasks below the tile's slot count caps the live slots)
    for (int w = 0; w < KW; w++) S.free_slots[w] = cap >= 32 * (w + 1) ? 0xffffffffu : cap > 32 * w ? ((1u << (cap - 32 * w)) - 1u) : 0u;
    for (int h = 0; h < H; h++) { S.h_status[h] = -9; S.h_target[h] = -1; S.h_mission[h] = -1; S.h_intercept[h] = -1; S.h_task_id[h] = -1; S.h_task_slot[h] = -1; }
    for (int a = 0; a < A; a++) { S.a_state[a] = -1; S.a_last_id[a] = -1; S.a_last_slot[a] = -1; S.a_fail[a] = -1; S.a_task_start[a] = -1; S.a_name[a] = -1; S.a_type[a] = 0; }
    // obstacles (:579-583)
    for (int o = 0; o < 8; o++) C.obst[o][2] = -1.0;
    for (int o = 0; o < P.num_obstacles && o < 8; o++) {
      double size = (double)randint(ST_OBS, 30, 100);
      double x, y;
      random_position(ST_OBS, 20, size, true, -1, true, x, y);
      C.obst[o][0] = x; C.obst[o][1] = y; C.obst[o][2] = size;
    }
    // agents (:591-612): shuffle ids, create in config order
    i16* ids = S.act_index;  // scratch: A ints in the blob, idle during a reset (the Scratch tile holds the reset RNG windows)
    for (int i = 0; i < nA; i++) ids[i] = i;
    for (int i = nA - 1; i >= 1; i--) {
      int j = (int)randbelow(ST_AGENT, (uint64_t)i + 1);
      int t = ids[i]; ids[i] = ids[j]; ids[j] = t;
    }
    int pop = 0;
    for (int g = 0; g < P.n_agent_groups; g++)
      for (int i = 0; i < P.agent_count[g]; i++) {
        int a = ids[pop];
        int ty = P.agent_type[g];
        S.a_name[a] = pop;
        pop++;
        S.a_type[a] = ty;
        if (P.random_init_pos) random_position(ST_AGENT, 20, 3, false, -1, true, S.a_px[a], S.a_py[a]);
        else { S.a_px[a] = BASE_X; S.a_py[a] = BASE_Y; }
        for (int c = 0; c < 6; c++) S.a_caps[c][a] = CAP_TABLE[ty][c];
        S.a_acap[a] = is_fighter(ty) ? 10 : 0;
        S.a_state[a] = 0;
        qs().a_nfx[a] = S.a_px[a]; qs().a_nfy[a] = S.a_py[a];
        qs().a_nft[a] = 0.0;  // (the HBM record is not zeroed by a reset)
      }
    for (int i = 0; i < nA; i++) ids[i] = 0;
    // fail events (:616-618)
    for (int a = 0; a < nA; a++)
      if (rnd(ST_AGENT) < P.fail_rate * FAIL_MULT[S.a_type[a]])
        S.a_fail[a] = (int)randint(ST_AGENT, 1, P.max_time_steps == -1 ? 1000 : P.max_time_steps);
    // mission areas (:621-634): SquareArea(center, area_width, area_width)
    for (int i = 0; i < 3; i++) {
      double aw = (double)(1200 * randint(ST_MISSION, 10, 20)) / 100;
      double ah = (double)(700 * randint(ST_MISSION, 10, 20)) / 100;
      double cx, cy;
      random_position(ST_MISSION, fmax(aw, ah), 3, false, -1, false, cx, cy);
      qs().area[i][0] = cx - aw / 2; qs().area[i][1] = cy - aw / 2; qs().area[i][2] = aw;
    }
    // static tasks (:641-667)
    int hold_num = 0;
    for (int g = 0; g < P.n_task_groups; g++)
      for (int i = 0; i < P.task_count[g]; i++) {
        int sel = (int)randbelow(ST_MISSION, 3);
        int ty = P.task_type[g];
        double x, y;
        if (ty != MUAVTA_HOLD) random_position(ST_TGT, 20, 3, true, sel, true, x, y);
        else { x = (double)(int)((hold_num + 1) * AREA_W / 5); y = (double)(int)(AREA_H / 4); hold_num++; }
        new_task(x, y, ty, 1.0);
      }
    // threat groups (:685-729)
    int hid = 0;
    const double wide = AREA_W / 10;
    for (int g = 0; g < P.n_threat_groups; g++) {
      double gx = (double)randint(ST_AGENT, (int)(0 + wide), (int)(AREA_W - wide));
      int det = new_task(gx, AREA_H / 5, MUAVTA_DET, (double)P.threat_count[g]);
      S.g_next[g] = hid;
      for (int k = 0; k < P.threat_count[g]; k++) {
        double sx = (double)randint(ST_AGENT, (int)(gx - wide), (int)(gx + wide));
        if (hid < H) {
          S.h_px[hid] = sx; S.h_py[hid] = 0.0;
          S.h_type[hid] = P.threat_type[g]; S.h_group[hid] = g;
          S.h_det_slot[hid] = det; S.h_acap[hid] = 4; S.h_status[hid] = -9;
        }
        hid++;
      }
      S.g_end[g] = hid;
    }
    S.did_reset = 1;  // (the initial tasks become known to everyone in reset(), one agent per lane)
  }

  // ====================================================================================================
  // step (:774-1206).  n_act staged actions in S.act_agent / S.act_slot (slot < 0: invalid index).
  // ====================================================================================================
  // np.sum of n <= 128 doubles with numpy's pairwise order: 8 strided accumulators r[k] (one per lane),
  // combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n % 8 tail added in order; plain loop for n < 8.
  DEV double np_sum_wave(const double* d, int n) {
    double res;
    if (n < 8) {
      res = 0.;
      for (int i = 0; i < n; i++) res += d[i];
      return res;
    }
    const int body = n - (n % 8);
    double r = 0.;
    if (lane < 8) {
      r = d[lane];
      for (int i = 8 + lane; i < body; i += 8) r += d[i];
    }
    r = r + dpp_xchg(r, 0);   // lanes 2m, 2m+1: r[2m] + r[2m+1]
    r = r + dpp_xchg(r, 1);   // quads: (r0+r1)+(r2+r3) | (r4+r5)+(r6+r7)
    r = r + dpp_xchg(r, 2);   // row_half_mirror pairs the two quads of lanes 0..7
    res = readlane_f64(r, 0);
    for (int i = body; i < n; i++) res += d[i];
    return res;
  }

  // The same sum for up to 16 elements held one per lane (lane i: element i): the second eight join the first through a row
  // shift, the tail comes through v_readlane — no scratch row, no LDS round trips.  Same additions in the same order.
  DEV double np_sum_lanes16(double d, int n_) {
    const int n = __builtin_amdgcn_readfirstlane(n_);
    double res;
    if (n < 8) {
      res = 0.;
      for (int i = 0; i < n; i++) res += readlane_f64(d, i);
      return res;
    }
    const int body = n - (n % 8);
    double r = d;
    if (body == 16) {  // lanes 0..7: d[j] + d[8 + j]   (row_shl:8, lanes 8..15 read beyond the row: 0)
      const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(d), 0x108, 0xf, 0xf, true);
      const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(d), 0x108, 0xf, 0xf, true);
      r = d + __hiloint2double(hi, lo);
    }
    r = r + dpp_xchg(r, 0);
    r = r + dpp_xchg(r, 1);
    r = r + dpp_xchg(r, 2);
    res = readlane_f64(r, 0);
    for (int i = body; i < n; i++) res += readlane_f64(d, i);
    return res;
  }

  DEV void step(bool write_obs_flag) {
    PROF(0);
    in_step = true;
    uint32_t rng_words = 0;
    if (!ABL(0)) { rng_refill(); rng_words = rng_prefetch_issue(); }
    PROF(1);
    // previous positions stay in registers of the lane that owns the agent
    double prev_x = 0, prev_y = 0;
    if (lane < P.n_agents) { prev_x = S.a_px[lane]; prev_y = S.a_py[lane]; }
    lds_sync();
    double r_action = 0, r_distance = 0, r_quality = 0, r_squality = 0;
    {
      // drain the event queue (:800-805): infos['events'] := event_list, every Reset_Allocation in order
      const int nev = S.n_events;
      for (int k = lane; k < nev; k += WG) { S.dev_tag[k] = S.ev_tag[k]; S.dev_arg[k] = S.ev_arg[k]; }
      tnow += 1;
      if (lane == 0) { S.step_reward = 0; S.time_steps = tnow; S.n_dev = nev; S.n_events = 0; }  // :796
      lds_sync();
      int last_arg = -1000;
      bool last_changed = true;
      if (!ABL(1))
      for (int k = 0; k < nev; k++)
        if (S.dev_tag[k] == MUAVTA_EV_RESET_ALLOCATION) {
          const int arg = S.dev_arg[k];
          if (arg == last_arg && !last_changed) continue;  // same call on unchanged state: nothing to do
          last_changed = release_all_tasks_coop(arg);
          last_arg = arg;
        }
    }
    PROF(2);
    if (!ABL(2))
    for (;;) {
      if (S.n_act > 0 && !apply_actions_parallel(r_action, r_distance, r_squality)) {
        precompute_actions();
        lds_sync();
        PROF(33);
        if (lane == 0) step_serial_a(r_action, r_distance, r_quality, r_squality);
        cold_sync();
      }
      // a list longer than the staging arrays (muavta_step_lists) is applied A items at a time, in order: the loop of
      // DroneEnv.py:813-933 is sequential, so chunks applied one after the other are the same sequence
      if (!more_agent || stage_more() == 0) break;
    }
    PROF(3);
    // movement (:965-1129): lanes commit every agent up to the first "event" agent, lane 0 plays that one
    // agent exactly as the reference does, and the wave resumes behind it
    for (int start = ABL(3) ? P.n_agents : 0; start < P.n_agents;) {
      bool popped = false;
      const int first = move_parallel(start, popped);
      // (lanes that dropped a retired head shifted their EnvCold queue-time row: only then do the HBM stores have to be drained.  An
      // unconditional vmcnt(0) here — half a microsecond after the step issued its RNG-window loads and a few after the previous step's
      // observation stores — waited for all of those in every step)
      if (popped) cold_sync(); else lds_sync();
      PROF(26);
#ifdef MUAVTA_PROF
      PROF_COUNT(40, 1000); if (first < P.n_agents) PROF_COUNT(41, 1000);
#endif
      if (first >= P.n_agents) break;
      if (lane == 0) step_serial_move(first, first + 1, r_quality);
      cold_sync();  // (a concluded task / retired escort rewrites other agents' queue rows: the next pass reads them)
      PROF(4);
      start = first + 1;
    }
    // distances (:1131-1138): np.linalg.norm(axis=1) == sqrt(x*x + y*y), no fma
    double dist_sum = 0.0;
    if (!ABL(4)) {
      double d = 0.0;
      if (lane < P.n_agents) {
        double dx = S.a_px[lane] - prev_x, dy = S.a_py[lane] - prev_y;
        d = fsqrt(dx * dx + dy * dy);
        S.a_dist[lane] += d;
        if constexpr (A > 16) X.u[lane] = d;
      }
      if constexpr (A <= 16) dist_sum = np_sum_lanes16(d, P.n_agents);  // np.sum(dists) (:1138)
      else { lds_sync(); dist_sum = np_sum_wave(X.u, P.n_agents); }
    }
    if (!ABL(0)) rng_prefetch_commit(rng_words);
    PROF(5);
    if (!ABL(5)) {
    precompute_threat_targets();
    if (lane == 0) step_serial_b(dist_sum);
    lds_sync();
    }
    if (!ABL(5))
    {  // update_threats (:1725-1744): lanes advance every threat up to the first one that engages or leaves the area,
       // lane 0 plays that one as the reference does, and the wave resumes behind it
      unsigned long long livemask = 0ull;
      if (lane == 0) X.retarget_h = -1;  // (scratch: set before any replay of this phase can post a request, read behind an lds_sync)
      if (!P.escort_enabled) {
        // Without escorts a threat's turn reads nothing another threat's turn writes (own record, own Int task, agent POSITIONS —
        // an engagement changes agents' states and queues, which only _retarget_threat_via_escort looks at): one pass decides
        // every threat, the non-event ones commit at once, and lane 0 replays the event threats in env.threats order (the
        // shared RNG stream and the counters see them in the reference's sequence).
        const unsigned long long em = update_threats_parallel(0, livemask, true);
        lds_sync();
        PROF(23);
        PROF_COUNT(54, 1000);
        if (em) {
          PROF_COUNT(55, 1000 * __popcll(em));
          for (unsigned long long m = em; m; m &= m - 1ull) {
            if (lane == 0) update_threats_serial(__ffsll((long long)m) - 1, livemask);
            lds_sync();
            resolve_retarget();
          }
        }
      } else
      for (int start = 0;;) {
        const int first = (int)update_threats_parallel(start, livemask, false);
        lds_sync();
        PROF(23);
        PROF_COUNT(54, 1000);
        if (first >= S.n_active_threats) break;
        PROF_COUNT(55, 1000);
        if (lane == 0) update_threats_serial(first, livemask);
        lds_sync();
        resolve_retarget();
        start = first + 1;
      }
      if (lane == 0) step_serial_b2();
    }
    lds_sync();
    if (!ABL(6)) {
    if (P.escort_enabled) sync_escorts_coop();
    PROF(6);
    sense_parallel();  // _wps_update_sensing (:1506-1523)
    lds_sync();
    }
    PROF(7);
    if (!ABL(7))
    {
      // wave-wide pre-checks so that lane 0 only walks the lists when something is due this step.  ONE pass over the live
      // slots gives both "a window expired" and "a task still blocks mission completion"; the agents' idle / responding
      // flags are read alongside.  Only a window expiry (status changes, agents freed) invalidates them: evaluated again then.
      bool due = false, expiring = false, idle = false, resp = false, blocking = false;
      if (!ABL(12)) for (int k = lane; k < S.n_pending; k += WG) due |= tnow >= S.pend_time[k];
      auto scan_slots = [&]() {
        expiring = false; blocking = false;
        for (int k = lane; k < S.n_order; k += WG) {
          const int s = S.t_order[k];
          const int fl = S.t_flags[s], st = S.t_status[s], dl = S.t_deadline[s], ty = S.t_type[s];
          expiring |= (fl & TF_DEADLINE) && st != 2 && tnow > dl;
          blocking |= !((fl & TF_ESCORT) || ty == MUAVTA_DET || ty == MUAVTA_HOLD || st == 2);  // !counts_for_mission_done(s)
        }
        idle = false; resp = false;
        if (lane < P.n_agents && S.a_state[lane] != -1) { idle = S.a_qlen[lane] == 0; resp = !idle; }
      };
      if (!ABL(12)) scan_slots();
      const bool any_due = __ballot(due) != 0ull;
      const bool any_exp = P.hard_windows && __ballot(expiring) != 0ull;
      PROF(28);
      if (any_due || any_exp) process_lists_coop(any_due, any_exp);
      lds_sync();
      PROF(29);
      if (any_exp) scan_slots();
      const int n_idle = __popcll(__ballot(idle));
      const bool responding = __ballot(resp) != 0ull;
      const bool all_done_tasks = __ballot(blocking) == 0ull;
      PROF(32);
      if (!ABL(11) && lane == 0) step_serial_c(r_action, r_distance, r_quality, r_squality, n_idle, responding, all_done_tasks);
    }
    lds_sync();
    PROF(8);
    if (!ABL(8)) finish_step_parallel(true);
    PROF(9);
  }

  DEV void displacement(double px, double py, double ux, double uy, double speed, double& ddx, double& ddy) {
    double avx, avy;
    avoid_obstacles(px, py, ux, uy, avx, avy);
    double mx = ux + avx, my = uy + avy;  // :1123
    norm_vector(mx, my);
    ddx = mx * speed; ddy = my * speed;
  }
  // The movement state machine (:965-1129) with one agent per lane.  An agent's turn only touches its own
  // fields unless it (a) fails, (b) engages an Int task (rewrites the threat's target) or (c) concludes a
  // task; those are "events".  Agents in [start, first event) are independent of everything later in the
  // loop and commit their lane's result.  An agent either heads for its current task or (idle) for the
  // base, never both, so each lane runs ONE distance / unit-vector / displacement pipeline.
  // Returns the index of the first event agent >= start (n_agents if none).
  DEV int move_parallel(int start, bool& popped) {
    const int a = lane;
    const bool in_fleet = a >= start && a < P.n_agents;
    const int ai = in_fleet ? a : 0;  // (lanes outside the fleet read agent 0's fields and discard them)
    // Every operand of the lane's agent is read up front — two LDS round trips (agent fields, then the fields of the task
    // they point to) instead of one per branch of the state machine; the values of untaken branches are simply unused.
    const int st0 = S.a_state[ai], fail_at = S.a_fail[ai], ts0 = S.a_task_start[ai], qlen = S.a_qlen[ai], reeval = S.a_reeval[ai];
    const int atype = S.a_type[ai];
    const int last_id = S.a_last_id[ai], last_slot = S.a_last_slot[ai], head_id_ = S.a_qid[ai][0], head_slot = S.a_qslot[ai][0];
    double px = S.a_px[ai], py = S.a_py[ai];
    const bool live = in_fleet && st0 != -1;
    int cid = 0, cs = -1;
    if (reeval) { cid = last_id; cs = last_slot; }
    else if (qlen > 0) { cid = head_id_; cs = head_slot; }
    const int csi = cs >= 0 ? cs : 0;
    const int tid_at = S.t_id[csi], tstat = S.t_status[csi], ty = S.t_type[csi];
    const double tpx = S.t_px[csi], tpy = S.t_py[csi];
    bool evt = false, pop_head = false;
    int new_st = 0, new_ts = 0;
    double ddx = 0.0, ddy = 0.0;
    if (live) {
      if (fail_at == tnow) {
        evt = true;
      } else {
        const double speed = speed_of(atype);
        new_st = st0; new_ts = ts0;
        const bool retired = cid != 0 && (!(cs >= 0 && tid_at == cid) || tstat == 2);  // ref_retired(cid, cs)
        const bool to_task = cid != 0 && !retired;
        const bool idle_check = new_st == 0 && !reeval && qlen == 0;    // :987-993
        const bool to_base = !to_task && (idle_check || new_st == 3);
        if (retired) pop_head = true;  // :1004-1007, own queue only (the task is retired: removeAgentCap is a no-op)
        if (to_task || to_base) {
          // vector to the target, its norm, unit vector, displacement — identical arithmetic on both paths:
          // task: dir/dist with the EPS guard (:1014-1020); base: norm_vector(base - pos) (:1119) whose norm
          // equals norm(pos - base) used by the distance tests (:992,:1116)
          const double tx = to_task ? tpx : BASE_X, ty_ = to_task ? tpy : BASE_Y;
          const double dx = tx - px, dy = ty_ - py;
          const double dist = norm2(dx, dy);
          double ux = 0, uy = 0;
          const bool zero = to_task ? (fabs(dist) < 1e-12) : (dist == 0);
          if (!zero) { const double r = frcp_nr(dist); ux = fdiv_r(dx, dist, r); uy = fdiv_r(dy, dist, r); }
          double ndx, ndy;
          displacement(px, py, ux, uy, speed, ndx, ndy);
          if (to_task) {
            const double engage = engage_range(atype);
            if (new_st == 1) {
              if (ty == MUAVTA_INT) {
                if (dist < engage) evt = true;
                else { ddx = ndx; ddy = ndy; }
              } else if (dist < speed) {
                new_st = 2; new_ts = tnow;
                px = tpx; py = tpy;
              } else { ddx = ndx; ddy = ndy; }
            } else if (new_st == 2) {
              if (ty == MUAVTA_INT && dist >= engage) new_st = 1;
              if (new_ts == -1) {
                new_ts = tnow;
                px = tpx; py = tpy;
              } else if ((tnow - new_ts) >= task_duration(ty) && ty != MUAVTA_HOLD && ty != MUAVTA_DEF &&
                         ty != MUAVTA_INT && ty != MUAVTA_DET) {
                evt = true;
              }
            }
          } else {
            if (idle_check && dist > speed + 5) new_st = 3;
            if (new_st == 3) {  // :1114-1121
              if (dist < speed + 5) new_st = 0;
              else { ddx = ndx; ddy = ndy; }
            }
          }
        }
      }
    }
    const unsigned long long em = __ballot(evt);
    const int first = em ? __ffsll((long long)em) - 1 : P.n_agents;
    PROF(34);
    popped = __ballot(live && a < first && pop_head) != 0ull;  // (uniform)
    if (live && a < first) {  // commit: own fields only
      if (pop_head) {
        des_allocate(a, cid);
        S.a_reeval[a] = 0;
        S.a_last_id[a] = -1; S.a_last_slot[a] = -1;
      }
      S.a_state[a] = new_st;
      S.a_task_start[a] = new_ts;
      px = px + ddx; py = py + ddy;  // :1125-1127
      S.a_px[a] = fmin(fmax(px, 0.0), AREA_W);
      S.a_py[a] = fmin(fmax(py, 0.0), AREA_H);
    }
    return first;
  }

  // events drain, action application, movement state machine
  // Geometry of the staged actions, one action per lane (each agent appears at most once per step on the
  // allocator path; repeated agents fall back to the inline computation): distance to the current head and to
  // the new task (switch penalty :859-861), time_to_task of UAV.allocate (DroneEnvComponents.py:64) and the
  // expected-distance term (:1216-1229).  Agent / task positions do not change while actions are applied.
  DEV double* act_f(int k) { return X.cost + k * A; }   // 4 arrays of A doubles in the idle cost tile
  DEV void precompute_actions() {
    cold_sync();  // next_free_position rows: written by whichever lane ran the previous phases
    const int k = lane;
    if (k >= S.n_act) return;
    const int a = S.act_agent[k], s = S.act_slot[k];
    X.remaining[k] = -1;  // validity tag: agent id
    if (a < 0 || s < 0 || S.a_state[a] == -1) return;
    const double px = S.a_px[a], py = S.a_py[a], tx = S.t_px[s], ty = S.t_py[s];
    double d_old = 0;
    if (S.a_qlen[a] > 0) { const int hs = S.a_qslot[a][0]; d_old = norm2(px - S.t_px[hs], py - S.t_py[hs]); }
    act_f(0)[k] = d_old;
    act_f(1)[k] = norm2(px - tx, py - ty);
    act_f(2)[k] = fdiv(norm2(qs().a_nfx[a] - tx, qs().a_nfy[a] - ty), speed_of(S.a_type[a]));
    const int n = S.a_qlen[a];  // after the append the queue holds n + 1 entries; tasks[-2] is the current last one
    double total;
    if (n >= 1) { const int ps = S.a_qslot[a][n - 1]; total = norm2(tx - S.t_px[ps], ty - S.t_py[ps]); }
    else total = norm2(tx - px, ty - py);
    act_f(3)[k] = -div_coord(total);
    bool dup = false;
    // (kept rolled: unrolled 32-fold with one scalar accumulator pair per element it alone spilled 90 SGPRs)
#pragma clang loop unroll(disable) vectorize(disable)
    for (int q = 0; q < k; q++) dup |= S.act_agent[q] == a;
    if (!dup) X.remaining[k] = a;
  }

  // ---------------------------------------------------------------------------------------------------------
  // Action application (:813-933) with ONE ACTION PER LANE.  Legal when every staged action names a different
  // agent (always true on the allocator path), several tasks per agent are allowed, and the saturation mask (reads
  // other actions' allocations) is off;
  // otherwise the caller uses precompute_actions + step_serial_a.  What is order dependent is kept in order:
  //  * the four reward accumulators get the same addends in the same sequence (replayed from lane registers);
  //  * allocatedReqs of a task taken by several agents this step is summed in action order (same-slot prefix);
  //  * escort creation (new task ids, events) runs on lane 0 afterwards, in action order.
  // Returns false (nothing touched) when the fast path does not apply.
  // ---------------------------------------------------------------------------------------------------------
  DEV bool apply_actions_parallel(double& action_reward, double& distance_reward, double& S_quality_reward) {
    const int n_act = S.n_act;
    if (!P.multiple_tasks_per_agent || P.saturate_mask || n_act > WG) return false;
    cold_sync();  // allocatedReqs / queue times: written by whichever lane ran the previous phases
    const int k = lane;
    const bool mine = k < n_act;
    const int a = mine ? S.act_agent[k] : -1, s = mine ? S.act_slot[k] : -1;
    {  // uniform bail-outs: a terminator inside the list, or an agent named twice — every lane leaves its action number in its
       // agent's cell of a scratch row: where two lanes name one agent, at least one of them reads back the other's number
      bool bad = mine && a < 0;
      if (mine && a >= 0) X.SR[a] = (uint8_t)k;
      lds_sync();
      bad |= mine && a >= 0 && X.SR[a] != (uint8_t)k;
      if (__ballot(bad) != 0ull) return false;
    }
    // ---- per-action part: own agent + read-only task data ----
    double q0 = 0, q1 = 0, q2 = 0, q3 = 0, d0 = 0, d1 = 0;  // addends in program order (S_quality x4, distance x2)
    int nq01 = 0, nq23 = 0, nd0 = 0, nd1 = 0, n_pen = 0;
    bool realloc = false, succ = false, idle_br = false;
    const int pending0 = S.pending_reset;
    double caps[6] = {0, 0, 0, 0, 0, 0};
    int ty = 0;
    if (mine && S.a_state[a] != -1) {
      if (s < 0) n_pen = 1;  // index beyond last_tasks_info (:835-838)
      else {
        const int tid = S.t_id[s];
        const int qlen = S.a_qlen[a];
        const int hid = qlen > 0 ? S.a_qid[a][0] : 0;
        const double px = S.a_px[a], py = S.a_py[a], tx = S.t_px[s], ty_ = S.t_py[s];
        bool cont = false;
        if (hid != tid) {
          if (hid != 0) {
            const int hs = S.a_qslot[a][0];  // kept alive by the GC while it heads a live agent's queue
            q0 = -0.1; q1 = -S.a_caps[S.t_type[hs]][a]; nq01 = 2;
            realloc = true;
            S.a_commit[a] = 0;
            const double dist_old = norm2(px - S.t_px[hs], py - S.t_py[hs]);
            const double dist_new = norm2(px - tx, py - ty_);
            d0 = div_coord(dist_old - dist_new); nd0 = 1;
          } else {
            q0 = 0.05; nq01 = 1;
            idle_br = true;  // the idle penalty (:866-868) looks at pending_reset, which an earlier action's escort creation sets
          }
        } else { q0 = 0.05; nq01 = 1; cont = true; }  // head is a real task (idle can never be indexed)
        if (!cont) {
          if (!action_valid(a, s)) n_pen = 1;
          else if (!(queue_find(a, tid) >= 0 || S.t_status[s] == 2)) {  // UAV.allocate (DroneEnvComponents.py:55-96)
            S.a_reeval[a] = 0; S.a_last_id[a] = -1; S.a_last_slot[a] = -1;
            const double time_to_task = fdiv(norm2(qs().a_nfx[a] - tx, qs().a_nfy[a] - ty_), speed_of(S.a_type[a]));
            const double start_time = (qs().a_nft[a] - (double)tnow) > 0 ? qs().a_nft[a] : (double)tnow;
            ty = S.t_type[s];
            const double end_time = start_time + time_to_task + (double)task_duration(ty);
            if (qlen == 0) { S.a_task_start[a] = -1; S.a_state[a] = 1; }
            if (qlen >= Q) fail(MUAVTA_ERR_QUEUE);
            else {
              // expected-distance term (:1216-1229): from the task back to the previous queue tail (or the agent)
              double total;
              if (qlen >= 1) { const int ps = S.a_qslot[a][qlen - 1]; total = norm2(tx - S.t_px[ps], ty_ - S.t_py[ps]); }
              else total = norm2(tx - px, ty_ - py);
              d1 = -div_coord(total); nd1 = 1;
              S.a_qid[a][qlen] = tid; S.a_qslot[a][qlen] = s; C.a_qtime[a][qlen] = time_to_task; S.a_qlen[a] = qlen + 1;
              qs().a_nft[a] = end_time; qs().a_nfx[a] = tx; qs().a_nfy[a] = ty_;
              if (S.a_state[a] != 1 && S.a_state[a] != -1) S.a_state[a] = 1;
#pragma unroll
              for (int c = 0; c < 6; c++) caps[c] = S.a_caps[c][a];
              succ = true;
            }
          }
        }
      }
    }
    // actions that will create an escort (and with it set pending_reset) further down, in action order
    const bool creates = P.escort_enabled && succ && ty == MUAVTA_REC && is_recon(S.a_type[a]) && !has_escort(a);
    const unsigned long long cm = __ballot(creates);
    if (idle_br && P.dynamic_idle_penalty != 0 && (pending0 || prefix_count(cm) != 0)) { q1 = -P.dynamic_idle_penalty; nq01 = 2; }
    // ---- task side: Task.addAgentCap in action order over the lanes that share a slot ----
    const unsigned long long sm = __ballot(succ);
    if (sm) {
      double pre[6] = {0, 0, 0, 0, 0, 0};
      if (succ) {
#pragma unroll
        for (int c = 0; c < 6; c++) pre[c] = C.t_alloc[c][s];
      }
      unsigned long long same = 0ull;  // successful lanes on my slot
      for (unsigned long long m = sm; m; m &= m - 1ull) {
        const int j = __ffsll((long long)m) - 1;
        const int sj = __builtin_amdgcn_readlane(s, j);
        const int aj = __builtin_amdgcn_readlane(a, j);
        if (succ && sj == s) {
          same |= 1ull << j;
          if (j < k) {
#pragma unroll
            for (int c = 0; c < 6; c++) pre[c] += S.a_caps[c][aj];
          }
        }
      }
      lds_sync();  // every lane has read allocatedReqs before the last one of each slot writes it back
      if (succ) {
        double after[6];
#pragma unroll
        for (int c = 0; c < 6; c++) after[c] = pre[c] + caps[c];
        // (selects, not caps[ty] / after[ty]: a dynamically indexed local array would live in scratch memory)
        double agentCap = caps[0], after_ty = after[0];
#pragma unroll
        for (int c = 1; c < 6; c++) if (ty == c) { agentCap = caps[c]; after_ty = after[c]; }
        double missing = C.t_cur[ty][s] - (after_ty - agentCap);
        missing = missing > 0 ? missing : 0;
        const double addedCap = missing - fmax(missing - agentCap, 0.0);
        if (addedCap <= 0) { q2 = -1.5; q3 = addedCap; nq23 = 2; } else { q2 = addedCap; nq23 = 1; }
        atomicOr(&S.t_bucket[s], 1ull << a);
        if ((same >> k) >> 1 == 0ull) {  // last successful lane on this slot
#pragma unroll
          for (int c = 0; c < 6; c++) C.t_alloc[c][s] = after[c];
          S.t_ndet[s] += __popcll(same);
          S.t_status[s] = 1;
        }
      }
      if (lane == 0) S.times_dirty = 1; obs_static_clear();
    }
    const int n_re = __popcll(__ballot(realloc));
    if (lane == 0 && n_re) { S.n_reallocations += n_re; S.n_task_switches += n_re; }
    if (P.escort_enabled) {  // _create_escort_for (:923-929) creates tasks and events: lane 0, in action order
      lds_sync();
      for (unsigned long long m = cm; m; m &= m - 1ull) {
        const int j = __ffsll((long long)m) - 1;
        const int aj = __builtin_amdgcn_readlane(a, j), sj = __builtin_amdgcn_readlane(s, j);
        if (lane == 0) create_escort_for(aj, sj);
      }
    }
    // ---- rewards: the reference's additions, in its order.  Every lane parks its addends behind those of the lanes before it
    // (counts by ballot + popcount) in two scratch rows; the sums are then plain left-to-right folds over the rows, read four
    // at a time (uniform addresses: every lane folds the same numbers).  r2 replayed them with ~12 v_readlane per action.
    // action_reward only ever receives -1 per penalty: -(count) is the same double whatever the order.
    {
      const unsigned long long b1 = __builtin_amdgcn_ballot_w64(nq01 >= 1), b2 = __builtin_amdgcn_ballot_w64(nq01 >= 2),
                               b3 = __builtin_amdgcn_ballot_w64(nq23 >= 1), b4 = __builtin_amdgcn_ballot_w64(nq23 >= 2),
                               e1 = __builtin_amdgcn_ballot_w64(nd0 != 0), e2 = __builtin_amdgcn_ballot_w64(nd1 != 0);
      double* qrow = X.cost;  // <= 4 per action (COSTN >= 4 * A)
      double* drow = X.v;     // <= 2 per action (T >= 2 * A)
      static_assert(T >= 2 * A, "distance addends need 2 * A scratch doubles");
      int qo = prefix_count(b1) + prefix_count(b2) + prefix_count(b3) + prefix_count(b4);  // (k == lane)
      int dn = prefix_count(e1) + prefix_count(e2);
      lds_sync();  // (the slot-side reads of the scratch rows above are done)
      if (nq01 >= 1) qrow[qo++] = q0;
      if (nq01 >= 2) qrow[qo++] = q1;
      if (nq23 >= 1) qrow[qo++] = q2;
      if (nq23 >= 2) qrow[qo++] = q3;
      if (nd0) drow[dn++] = d0;
      if (nd1) drow[dn++] = d1;
      const int nQ = __popcll(b1) + __popcll(b2) + __popcll(b3) + __popcll(b4), nD = __popcll(e1) + __popcll(e2);
      const int nP = __popcll(__builtin_amdgcn_ballot_w64(n_pen != 0));
      lds_sync();
      auto fold = [](double acc, const double* row, int n) {
        int i = 0;
        for (; i + 4 <= n; i += 4) { const double a0 = row[i], a1 = row[i + 1], a2 = row[i + 2], a3 = row[i + 3]; acc += a0; acc += a1; acc += a2; acc += a3; }
        for (; i < n; i++) acc += row[i];
        return acc;
      };
      S_quality_reward = fold(S_quality_reward, qrow, nQ);
      distance_reward = fold(distance_reward, drow, nD);
      for (int i = 0; i < nP; i++) action_reward += -1;
    }
    cold_sync();
    return true;
  }
  // action application (:813-933), dict order, lane 0
  DEV void step_serial_a(double& action_reward, double& distance_reward, double& quality_reward, double& S_quality_reward) {
    // ---- task allocation (:813-933) ----
    for (int k = 0; k < S.n_act; k++) {
      int a = S.act_agent[k];
      if (a < 0) break;
      if (S.a_state[a] == -1) continue;
      int s = S.act_slot[k];
      if (s < 0) { action_reward += -1; continue; }  // index beyond last_tasks_info (:835-838)
      int tid = S.t_id[s];
      const bool pre = P.multiple_tasks_per_agent && X.remaining[k] == a;  // lane k's geometry is valid for this action
      {
        int hid_ = head_id(a);
        if (hid_ != tid) {
          if (hid_ != 0) {
            int hs = S.a_qslot[a][0];  // kept alive by the GC while it heads a live agent's queue
            S_quality_reward -= 0.1;
            S_quality_reward -= S.a_caps[S.t_type[hs]][a];
            S.n_reallocations += 1;
            S.n_task_switches += 1;
            S.a_commit[a] = 0;
            double dist_old, dist_new;
            if (pre) { dist_old = act_f(0)[k]; dist_new = act_f(1)[k]; }
            else {
              dist_old = norm2(S.a_px[a] - S.t_px[hs], S.a_py[a] - S.t_py[hs]);
              dist_new = norm2(S.a_px[a] - S.t_px[s], S.a_py[a] - S.t_py[s]);
            }
            distance_reward += div_coord(dist_old - dist_new);
          } else {
            S_quality_reward += 0.05;
            if (S.pending_reset && P.dynamic_idle_penalty != 0) S_quality_reward -= P.dynamic_idle_penalty;
          }
        } else {
          S_quality_reward += 0.05;  // head is a real task (idle can never be indexed)
          continue;
        }
      }
      if (!P.multiple_tasks_per_agent) {  // EnvUtils.desallocateAll([agent], env) (MultiDroneEnvUtils.py:183-205)
        while (S.a_qlen[a] > 0) {
          int qs = S.a_qslot[a][0], qi = S.a_qid[a][0];
          des_allocate_at(a, 0);
          if (ref_valid(qi, qs)) S.t_bucket[qs] &= ~(1ull << a);
        }
        qs().a_nft[a] = (double)tnow;
        qs().a_nfx[a] = S.a_px[a];
        qs().a_nfy[a] = S.a_py[a];
      }
      if (!action_valid(a, s)) { action_reward += -1; continue; }
      if (uav_allocate(a, s, pre ? act_f(2)[k] : -1.0)) {
        S.t_bucket[s] |= 1ull << a;
        int ty = S.t_type[s];
        double agentCap = S.a_caps[ty][a];
        double missing = C.t_cur[ty][s] - (C.t_alloc[ty][s] - agentCap);
        missing = missing > 0 ? missing : 0;
        double addedCap = missing - fmax(missing - agentCap, 0.0);
        if (addedCap <= 0) S_quality_reward -= 1.5;
        S_quality_reward += addedCap;
        S.t_status[s] = 1;
        if (pre) {
          distance_reward += act_f(3)[k];
        } else {  // calculate_agent_expected_reward (:1216-1229)
          int n = S.a_qlen[a];
          double total;
          if (n >= 2) {
            int ps = S.a_qslot[a][n - 2];  // live agents' queue entries are never freed by the GC
            total = norm2(qs().a_nfx[a] - S.t_px[ps], qs().a_nfy[a] - S.t_py[ps]);
          } else {
            total = norm2(qs().a_nfx[a] - S.a_px[a], qs().a_nfy[a] - S.a_py[a]);
          }
          distance_reward += -div_coord(total);
        }
        if (S.a_state[a] != 1 && S.a_state[a] != -1) S.a_state[a] = 1;
        if (P.escort_enabled && ty == MUAVTA_REC && is_recon(S.a_type[a]) && !has_escort(a)) create_escort_for(a, s);
      }
    }
  }

  // ---- movement state machine (:965-1129), agents_obj order, lane 0 ----
  // the reference's loop body for agents [first, last) on lane 0
  DEV void step_serial_move(int first, int last, double& quality_reward) { quality_reward += step_serial_move_impl(first, last); }
  DEV double step_serial_move_impl(int first, int last) {
    double quality_reward = 0;
    for (int a = first; a < last; a++) {
      if (S.a_state[a] == -1) continue;
      if (S.a_fail[a] == tnow) {  // :972-981
        S.a_state[a] = -1;
        desallocate_all(a);
        push_event(MUAVTA_EV_RESET_ALLOCATION, -1);
        push_event(MUAVTA_EV_AGENT_FAIL, a);
        S.pending_reset = 1;
        continue;
      }
      const double speed = speed_of(S.a_type[a]);
      double px = S.a_px[a], py = S.a_py[a];
      double ddx = 0.0, ddy = 0.0;  // displacement of this step (movement normalised twice, times max_speed)
      if (S.a_state[a] == 0 && !S.a_reeval[a]) {  // :987-993
        if (S.a_qlen[a] == 0 && norm2(px - BASE_X, py - BASE_Y) > speed + 5) S.a_state[a] = 3;
      }
      {
        // current task: last_task while re_eval, else the head (:996-1002); id 0 == task_idle
        int cid, cs;
        if (S.a_reeval[a]) { cid = S.a_last_id[a]; cs = S.a_last_slot[a]; }
        else if (S.a_qlen[a] > 0) { cid = S.a_qid[a][0]; cs = S.a_qslot[a][0]; }
        else { cid = 0; cs = -1; }
        if (cid != 0 && ref_retired(cid, cs)) {  // :1004-1007 (task_idle.status is never 2)
          des_allocate(a, cid);
          S.a_reeval[a] = 0;
          S.a_last_id[a] = -1; S.a_last_slot[a] = -1;
        } else if (cid != 0) {
          const int ty = S.t_type[cs];
          const double engage = engage_range(S.a_type[a]);
          double dx = S.t_px[cs] - px, dy = S.t_py[cs] - py;
          const double dist = norm2(dx, dy);
          double ux = 0, uy = 0;
          if (!(fabs(dist) < 1e-12)) { const double r = frcp_nr(dist); ux = fdiv_r(dx, dist, r); uy = fdiv_r(dy, dist, r); }
          if (S.a_state[a] == 1) {  // navigating (:1012-1048)
            if (ty == MUAVTA_INT) {
              if (dist < engage) {
                S.a_state[a] = 2;
                S.h_target[S.t_threat[cs]] = a;
                S.a_task_start[a] = tnow;
              } else {
                displacement(px, py, ux, uy, speed, ddx, ddy);
              }
            } else if (dist < speed) {
              S.a_state[a] = 2;
              S.a_task_start[a] = tnow;
              px = S.t_px[cs]; py = S.t_py[cs];
            } else {
              displacement(px, py, ux, uy, speed, ddx, ddy);
            }
          } else if (S.a_state[a] == 2) {  // in task (:1051-1110)
            if (ty == MUAVTA_INT) {
              if (dist >= engage) S.a_state[a] = 1;
            }
            if (S.a_task_start[a] == -1) {
              S.a_task_start[a] = tnow;
              px = S.t_px[cs]; py = S.t_py[cs];
            } else if ((tnow - S.a_task_start[a]) >= task_duration(ty) && ty != MUAVTA_HOLD && ty != MUAVTA_DEF &&
                       ty != MUAVTA_INT && ty != MUAVTA_DET) {
              // task concluded by this agent (:1079-1107).  Everything this block needs from the HBM record is requested first, in
              // one batch (doneReqs / orgReqs live there on the SLIM tile, the requirement vectors on every tile): one memory round
              // trip instead of one per read-modify-write.
              S.a_px[a] = px; S.a_py[a] = py;  // taskDone reads agent.position
              const double org_cs = qs().t_org[cs];
              double done_cs = qs().t_done[cs], cur6[6], al6[6];
#pragma unroll
              for (int c = 0; c < 6; c++) { cur6[c] = C.t_cur[c][cs]; al6[c] = C.t_alloc[c][cs]; }
              bool was_head = task_done(a, cid, ty);
              done_cs += S.a_caps[ty][a];
              qs().t_done[cs] = done_cs;
#pragma unroll
              for (int c = 0; c < 6; c++) C.t_cur[c][cs] = cur6[c] - S.a_caps[c][a];
              obs_static_clear();
              if (was_head && S.t_status[cs] != 2) {  // remove_agent_cap(cs, a)
#pragma unroll
                for (int c = 0; c < 6; c++) C.t_alloc[c][cs] = al6[c] - S.a_caps[c][a];
                S.t_ndet[cs] -= 1;
                S.times_dirty = 1; obs_static_clear();
              }
              if (done_cs >= org_cs) {
                const bool esc = S.t_flags[cs] & TF_ESCORT;
                if (!esc && !(S.t_flags[cs] & TF_REACHED)) { S.t_flags[cs] |= TF_REACHED; S.n_reached++; }
                quality_reward += org_cs * 2;
                S.F_Reward += org_cs * 1 / P.reward_norm_factor;
                if (!esc) mark_outcome_slot(cs, true);
                S.t_status[cs] = 2;
                if (ty == MUAVTA_REC && is_recon(S.a_type[a])) {  // _on_protected_rec_done (:1959-1962)
                  S.protected_rec_completed++;
                  retire_escort_for(a, false);
                }
                if (all_mission_done()) S.conclusion_time = tnow;
              } else {
                quality_reward += S.a_caps[ty][a];
              }
            }
          }
        }
      }
      if (S.a_state[a] == 3) {  // returning to base (:1114-1121)
        if (norm2(px - BASE_X, py - BASE_Y) < speed + 5) {
          S.a_state[a] = 0;
        } else {
          double bx = BASE_X - px, by = BASE_Y - py;
          norm_vector(bx, by);
          displacement(px, py, bx, by, speed, ddx, ddy);
        }
      }
      px = px + ddx; py = py + ddy;  // :1125-1127
      S.a_px[a] = fmin(fmax(px, 0.0), AREA_W);
      S.a_py[a] = fmin(fmax(py, 0.0), AREA_H);
    }
    return quality_reward;
  }

  // total distance, threats, arrivals, escorts
  DEV void step_serial_b(double dist_sum) {
    S.total_distance += dist_sum;
    // :1140-1145 — evaluated here, before this step's spawns / expiries change the counts.  A term whose
    // weight is 0 contributes +-0.0 to the reward sum whatever its value, so it is not evaluated at all.
    S.r_time_penalty = 0;
    if (P.rw[6] != 0) S.r_time_penalty = -(double)(P.n_tasks - S.n_reached) / (double)P.n_tasks * ((double)tnow / (double)P.max_time_steps);
    S.r_alloc = 0;
    if (P.rw[5] != 0 && tnow > P.n_tasks + 1) {  // -len(unallocated_tasks()) (:1434-1440); bucket 0 (idle) is always empty
      int n = 1 + S.n_retired_empty_buckets;
      for (int k = 0; k < S.n_order; k++) if (S.t_bucket[S.t_order[k]] == 0) n++;
      S.r_alloc = -(double)n;
    }
    PROF(21);
    generate_threat();
    PROF(22);
  }
  DEV void step_serial_b2() {
    PROF(24);
    inject_dynamic_arrivals();
    PROF(25);
  }

  // get_closest_agent (:1691-1723) for every threat that can spawn this step, one (threat, agent) pair per lane, so
  // that lane 0's generate_threat only looks the answer up.  Spawn positions are fixed at reset unless
  // dual_region_bursts redraws x (then generate_threat searches itself).  Result: X.roundT[threat id].
  DEV bool threat_spawn_step() const { return tnow > 40 && tnow % 10 == 0; }
  DEV void precompute_threat_targets() {
    if (!threat_spawn_step() || P.dual_region_bursts) return;  // uniform
    const int nA = P.n_agents;
    for (int g = 0; g < P.n_threat_groups; g++) {
      const int start = S.g_next[g], left = S.g_end[g] - start;
      int cnt = P.burst_mode ? (P.burst_size < left ? P.burst_size : left) : (left > 0 ? 1 : 0);
      if (cnt <= 0) continue;
      const int chunk = Scratch<TL>::COSTN / nA;  // threats per pass of the distance scratch (>= 1: COSTN >= 4 * A)
      for (int c0 = 0; c0 < cnt; c0 += chunk) {
        const int cn = cnt - c0 < chunk ? cnt - c0 : chunk;
        for (int p = lane; p < cn * nA; p += WG) {
          const int c = p / nA, a = p - c * nA, h = start + c0 + c;
          const int st = S.a_state[a];
          X.cost[p] = (h < H && st != -1 && st != 4) ? norm2(S.a_px[a] - S.h_px[h], S.a_py[a] - S.h_py[h]) : __builtin_huge_val();
        }
        lds_sync();
        for (int c = lane; c < cn; c += WG) {
          double minF = __builtin_huge_val(), minW = __builtin_huge_val();
          int cF = -1, cW = -1;
          for (int a = 0; a < nA; a++) {
            const double d = X.cost[c * nA + a];
            const int st = S.a_state[a];
            if (st != -1 && st != 4) {
              if (is_fighter(S.a_type[a])) { if (d < minF) { minF = d; cF = a; } }
              else { if (d < minW) { minW = d; cW = a; } }
            }
          }
          if (start + c0 + c < H) X.roundT[start + c0 + c] = cW >= 0 ? cW : cF;
        }
        lds_sync();
      }
    }
  }
  DEV void generate_threat() {  // :1601-1643
    if (!threat_spawn_step()) return;  // (the per-group test below needs it anyway: nine steps in ten end here)
    for (int g = 0; g < P.n_threat_groups; g++) {
      int left = S.g_end[g] - S.g_next[g];
      if (left > 0 && threat_spawn_step()) {
        if (rnd(ST_AGENT) < P.threat_prob) {
          int n_spawn = 1;
          if (P.burst_mode) n_spawn = P.burst_size < left ? P.burst_size : left;
          for (int bi = 0; bi < n_spawn; bi++) {
            if (S.g_next[g] >= S.g_end[g]) break;
            int h = S.g_next[g]++;
            if (h >= H) { fail(MUAVTA_ERR_TASK_SLOTS); break; }
            if (P.dual_region_bursts) {
              double mid = AREA_W * 0.5;
              double wide = fmax(AREA_W / 10, 40.0);
              double x;
              if ((S.burst_region_toggle + bi) % 2 == 0) x = uniform(ST_AGENT, wide, mid - wide);
              else x = uniform(ST_AGENT, mid + wide, AREA_W - wide);
              S.h_px[h] = x;
            }
            const int tgt = P.dual_region_bursts ? closest_agent(S.h_px[h], S.h_py[h]) : X.roundT[h];
            S.h_target[h] = tgt;
            S.h_mission[h] = tgt;
            // TaskFromThreat (:1861-1876)
            int ty = S.h_type[h];
            double attack = threat_attack(ty), defence = threat_defence(ty);
            int s = new_task(S.h_px[h], S.h_py[h], MUAVTA_INT, 2.0);
            if (s >= 0) {
              C.t_cur[MUAVTA_ATT][s] = defence * 2;
              C.t_cur[MUAVTA_DEF][s] = attack * 2;
              obs_static_clear();
              S.t_threat[s] = h;
              S.t_created[s] = tnow;
              if (ty == MUAVTA_T1) { S.t_required[s] = 2; S.t_flags[s] |= TF_ELIGIBLE; S.t_elig[s] = P.escort_mask; }
              S.h_task_id[h] = S.t_id[s];
              S.h_task_slot[h] = s;
            }
            S.h_status[h] = 1;
            S.h_order[S.n_active_threats++] = h;
            C.t_cur[5][S.h_det_slot[h]] -= 1.0;
            if (s >= 0) {
              register_dynamic(s);
              push_event(MUAVTA_EV_NEW_THREAT, S.t_id[s]);
            }
            push_event(MUAVTA_EV_RESET_ALLOCATION, MUAVTA_INT);
            S.pending_reset = 1;
          }
          if (P.dual_region_bursts && n_spawn > 0) S.burst_region_toggle = (S.burst_region_toggle + 1) % 2;
        }
      }
    }
  }

  // the threat's Int task: live slot, or the (flags, deadline) copy kept on the threat once freed
  DEV void threat_task_retire(int h, bool success) {
    int s = S.h_task_slot[h];
    if (ref_valid(S.h_task_id[h], s)) { S.t_status[s] = 2; mark_outcome_slot(s, success); }
    else mark_outcome(S.h_tflags[h], S.h_tdeadline[h], success);
  }

  DEV void handle_threat_engagement(int h) {  // :1781-1858
    int primary = S.h_target[h];
    int mission = S.h_mission[h] >= 0 ? S.h_mission[h] : primary;
    int n_def = 0;
    int16_t* defs = X.remaining;   // scratch lists (T >= A)
    double* defd = X.v;
    if (P.escort_enabled && mission >= 0 && is_recon(S.a_type[mission])) {
      n_def = escort_fighters_near(mission, P.mutual_support_radius, defs, defd);
      if (n_def > 0) {
        primary = defs[0];
        S.h_target[h] = primary;
        S.h_intercept[h] = primary;
      }
    }
    if (primary < 0) return;
    const int hty = S.h_type[h];
    const double t_att = threat_attack(hty), t_def = threat_defence(hty), t_rng = engage_range(hty);
    double attDiff, defDiff, engageDiff;
    if (n_def >= 2) {
      S.mutual_support_engagements++;
      double att_sum = 0, def_sum = 0, eng_sum = 0;
      for (int k = 0; k < n_def; k++) att_sum += S.a_caps[2][defs[k]];
      for (int k = 0; k < n_def; k++) def_sum += S.a_caps[3][defs[k]];
      for (int k = 0; k < n_def; k++) eng_sum += engage_range(S.a_type[defs[k]]);
      eng_sum = eng_sum / (double)n_def;
      attDiff = att_sum / fmax(t_att, 1e-6);
      defDiff = def_sum / fmax(t_def, 1e-6);
      engageDiff = eng_sum / fmax(t_rng, 1e-6);
    } else {
      attDiff = S.a_caps[2][primary] / fmax(t_att, 1e-6);
      defDiff = S.a_caps[3][primary] / fmax(t_def, 1e-6);
      engageDiff = engage_range(S.a_type[primary]) / fmax(t_rng, 1e-6);
    }
    double avg_diff = (attDiff + defDiff + engageDiff) / 3;
    double prob = avg_diff / (avg_diff + 1);
    double r = rnd(ST_AGENT);
    if (r < prob) {
      S.h_status[h] = 2;
      threat_task_retire(h, true);
      S.threats_intercepted++;
      S.a_acap[primary] -= 1;
      if (S.a_acap[primary] <= 0) S.a_caps[3][primary] = 0;
      if (S.a_qlen[primary] > 0 && S.a_qid[primary][0] == S.h_task_id[h]) task_done(primary, S.h_task_id[h], MUAVTA_INT);
      S.step_reward += 1.0;
    } else {
      S.h_acap[h] -= 1;
      S.a_acap[primary] -= 1;
      if (S.a_acap[primary] <= 0) {
        S.a_caps[3][primary] = 0;
        bool was_recon = is_recon(S.a_type[primary]);
        bool was_escort = escort_type(S.a_type[primary]);
        out_of_service(primary);
        if (was_recon) { S.recon_losses++; S.protection_breaches++; retire_escort_for(primary, true); }
        else if (was_escort) S.escort_losses++;
        S.step_reward -= 1.0;
      }
      if (S.h_acap[h] <= 0) {
        S.h_status[h] = 0;
        threat_task_retire(h, false);
      } else {
        X.retarget_h = h;  // threat.target = threat.mission_target = get_closest_agent(position): left to the whole wave (resolve_retarget),
                           // nothing reads either field before the caller gets there
      }
    }
  }

  // update_threats (:1725-1744), one active threat per lane.  A threat's turn touches only its own record
  // (position, target, its Int task's position) unless it ENGAGES (shared RNG stream, kills) or leaves the
  // area with an uncounted window (shared counters): those are events.  Threats before the first event
  // (env.threats order) commit their lane's result; lane 0 replays the reference loop from the event on.
  // Returns (first serial index, snapshot mask of threats with status != 2).
  // lanes [start, n): the threats behind the last serially replayed one; `livemask` is taken once, at start == 0
  // all_at_once (no escorts): every non-event threat commits and the mask of event threats is returned instead of the first one
  DEV unsigned long long update_threats_parallel(int start, unsigned long long& livemask, bool all_at_once) {
    static_assert(H <= 64, "one threat per lane");
    const int n = S.n_active_threats;
    const int k = lane;
    // operands in three LDS round trips (spawn-order slot -> the threat's record -> its target's position and its Int
    // task's slot), whatever branch the lane takes afterwards
    const int h = k < n ? (int)S.h_order[k] : 0;
    const int hst = S.h_status[h], hty = S.h_type[h], slot = S.h_task_slot[h], task_id = S.h_task_id[h], htf = S.h_tflags[h];
    int tgt = S.h_target[h], icpt = S.h_intercept[h];
    const int hmis = S.h_mission[h];
    const double px = S.h_px[h], py = S.h_py[h];
    bool active = k < n && hst != 2;
    if (start == 0) livemask = __ballot(active);  // python snapshots [t for t in self.threats if t.status != 2] before the loop
    active = k >= start && k < n && ((livemask >> k) & 1ull);
    const int sli = slot >= 0 ? slot : 0;
    const int t_id_at = S.t_id[sli], t_stat = S.t_status[sli], t_fl = S.t_flags[sli];
    bool evt = false;
    double npx = 0, npy = 0;
    bool live_task = false;
    if (P.escort_enabled) {  // _retarget_threat_via_escort (:1766-1779): reads agents only.  One protected UAV per threat lane,
                             // the escort map and the fleet walked in uniform loops (closest_escort_lanes)
      const int mission = hmis >= 0 ? hmis : tgt;
      const int mi = mission >= 0 ? mission : 0;
      const bool need = active && !(hst == 0 || tgt < 0) && mission >= 0 && S.a_state[mi] != -1 && is_recon(S.a_type[mi]);
      int e0, cnt;
      closest_escort_lanes(need, mi, P.escort_intercept_radius, e0, cnt);
      if (need) {
        if (cnt == 0) { tgt = mission; icpt = -1; }
        else { tgt = e0; icpt = e0; }
      }
    }
    if (active) {
      const double speed = speed_of(hty);
      if (hst == 0 || tgt < 0) {
        npx = px + speed * 0.0;
        npy = py + speed * -1.0;
      } else {
        const double ax = S.a_px[tgt], ay = S.a_py[tgt];
        double dx = ax - px, dy = ay - py;
        norm_vector(dx, dy);
        npx = px + speed * dx;
        npy = py + speed * dy;
        if (norm2(ax - npx, ay - npy) < engage_range(hty)) evt = true;
      }
      live_task = slot >= 0 && t_id_at == task_id;  // ref_valid(S.h_task_id[h], slot)
      if (npy <= 0) {  // leaving the area: an event only if the retirement would still change something
        if (live_task) evt |= (t_stat != 2) || ((t_fl & TF_DEADLINE) && !(t_fl & TF_COUNTED));
        else evt |= (htf & TF_DEADLINE) && !(htf & TF_COUNTED);
      }
    }
    const unsigned long long em = __ballot(evt);
    const int first = em ? __ffsll((long long)em) - 1 : n;
    if (active && (all_at_once ? !evt : k < first)) {
      S.h_px[h] = npx; S.h_py[h] = npy;
      S.h_target[h] = tgt; S.h_intercept[h] = icpt;
      if (live_task) { S.t_px[slot] = npx; S.t_py[slot] = npy; }
    }
    return all_at_once ? em : (unsigned long long)first;
  }
  DEV void update_threats_serial(int first, unsigned long long livemask) {  // the reference's loop body for ONE threat
    const int n = first + 1 < S.n_active_threats ? first + 1 : S.n_active_threats;
    for (int k = first; k < n; k++) {
      if (!((livemask >> k) & 1ull)) continue;
      int h = S.h_order[k];
      const double speed = speed_of(S.h_type[h]);
      if (S.h_status[h] == 0 || S.h_target[h] < 0) {
        S.h_px[h] = S.h_px[h] + speed * 0.0;
        S.h_py[h] = S.h_py[h] + speed * -1.0;
      } else {
        if (P.escort_enabled) {  // _retarget_threat_via_escort (:1766-1779)
          int mission = S.h_mission[h] >= 0 ? S.h_mission[h] : S.h_target[h];
          if (mission >= 0 && S.a_state[mission] != -1 && is_recon(S.a_type[mission])) {
            int cnt;
            int e0 = closest_escort(mission, P.escort_intercept_radius, &cnt);
            if (cnt == 0) { S.h_target[h] = mission; S.h_intercept[h] = -1; }
            else { S.h_target[h] = e0; S.h_intercept[h] = e0; }
          }
        }
        int tg = S.h_target[h];
        double dx = S.a_px[tg] - S.h_px[h], dy = S.a_py[tg] - S.h_py[h];
        norm_vector(dx, dy);
        S.h_px[h] = S.h_px[h] + speed * dx;
        S.h_py[h] = S.h_py[h] + speed * dy;
        if (norm2(S.a_px[tg] - S.h_px[h], S.a_py[tg] - S.h_py[h]) < engage_range(S.h_type[h])) handle_threat_engagement(h);
      }
      int s = S.h_task_slot[h];
      bool live = ref_valid(S.h_task_id[h], s);
      if (live) { S.t_px[s] = S.h_px[h]; S.t_py[s] = S.h_py[h]; }
      if (S.h_py[h] <= 0) threat_task_retire(h, false);
    }
  }

  DEV void inject_dynamic_arrivals() {  // :1646-1689
    if (P.arrival_rate <= 0 || tnow < 5) return;
    if (rnd(ST_TGT) >= P.arrival_rate) return;
    if (S.next_task_id - 1 >= P.max_tasks - 1) return;  // len(self.tasks) >= max_tasks - 1
    int ty = randbelow(ST_TGT, 2) == 0 ? MUAVTA_ATT : MUAVTA_REC;
    // the reference allocates the id before drawing the mission area; ids are only consumed here
    int sel = (int)randbelow(ST_MISSION, 3);
    double x, y;
    if (P.dual_region_bursts) {
      double mid = AREA_W * 0.5, wide = 40.0;
      if (rnd(ST_TGT) < 0.5) x = uniform(ST_TGT, wide, mid - wide);
      else x = uniform(ST_TGT, mid + wide, AREA_W - wide);
      y = uniform(ST_TGT, AREA_H * 0.2, AREA_H * 0.8);
    } else {
      random_position(ST_TGT, 20, 3, true, sel, true, x, y);
    }
    int s = new_task(x, y, ty, 1.0);
    S.n_arrivals++;
    if (s >= 0) {
      S.t_created[s] = tnow;
      register_dynamic(s);
      push_event(MUAVTA_EV_NEW_THREAT, S.t_id[s]);
    }
    push_event(MUAVTA_EV_RESET_ALLOCATION, ty);
    S.pending_reset = 1;
  }

  DEV void sync_escorts() {  // :1964-2000
    for (int a = 0; a < P.n_agents; a++) {
      if (S.a_state[a] == -1 || !is_recon(S.a_type[a])) continue;
      if (S.a_qlen[a] == 0) continue;
      int cid = S.a_qid[a][0], cs = S.a_qslot[a][0];
      if (ref_retired(cid, cs)) continue;
      if (S.t_type[cs] == MUAVTA_REC && !has_escort(a)) create_escort_for(a, cs);
    }
    // iterate a snapshot of the map (retiring pops entries)
    int n = S.n_escorts;
    int16_t* snap = X.row4col;  // T >= A
    for (int k = 0; k < n; k++) snap[k] = S.esc_agent[k];
    for (int k = 0; k < n; k++) {
      int recon = snap[k];
      int kk = escort_lookup(recon);
      if (kk < 0) continue;
      int es = S.esc_slot[kk];
      int rid = S.esc_pid[kk], rs = S.esc_pslot[kk];
      bool dead = S.a_state[recon] == -1;
      bool idle = S.a_qlen[recon] == 0 || S.a_state[recon] == 0 || S.a_state[recon] == 3;
      bool rec_done = ref_retired(rid, rs);
      bool wrong_task = S.a_qlen[recon] > 0 && S.a_qid[recon][0] != rid;
      if (dead || idle || rec_done || wrong_task) { retire_escort_entry(kk, dead); continue; }
      if (ref_valid(S.esc_id[kk], es)) { S.t_px[es] = S.a_px[recon]; S.t_py[es] = S.a_py[recon]; }
      S.escort_required_steps++;
      int cnt;
      closest_escort(recon, P.escort_radius, &cnt);
      if (cnt > 0) S.escort_covered_steps++;
    }
  }

  // _sync_escorts (:1964-2000) with the whole wave: the per-agent / per-escort predicates are evaluated one per
  // lane (ballots), creations and retirements (shared state, id allocation order) stay on lane 0 in the
  // reference's order, and the coverage test "any escort fighter within radius" is a ballot over agents.
  DEV void sync_escorts_coop() {
    {  // recon UAVs already on a Rec task without an escort (:1967-1974)
      const int a = lane;
      bool need = false;
      int cs = -1;
      if (a < P.n_agents && S.a_state[a] != -1 && is_recon(S.a_type[a]) && S.a_qlen[a] > 0) {
        const int cid = S.a_qid[a][0];
        cs = S.a_qslot[a][0];
        need = !ref_retired(cid, cs) && S.t_type[cs] == MUAVTA_REC && !has_escort(a);
      }
      unsigned long long nm = __ballot(need);
      if (nm) {
        if (lane == 0) {
          while (nm) {
            const int b = __ffsll((long long)nm) - 1;
            nm &= nm - 1ull;
            create_escort_for(b, S.a_qslot[b][0]);
          }
        }
        lds_sync();
      }
    }
    // Snapshot of the map, one entry per lane (retiring pops entries).  Whether an entry retires depends only on its
    // recon UAV and Rec task, which no other entry's retirement touches, so that is decided up front; coverage of
    // the entries BETWEEN two retirements is then evaluated in one pass (fighters' queues only change at a
    // retirement), one fighter per lane, instead of one map entry at a time.
    PROF(59);
    const int n = S.n_escorts;
    if (n == 0) return;
    int recon = -1, es = -1, eid = -1, rid = -1, rs = -1;
    bool retire = false, dead = false, esc_live = false, esc_here = false;
    auto decide = [&]() {  // :1978-1990 for the lane's entry, on the state as it is now
      if (lane < n) {
        dead = S.a_state[recon] == -1;
        const bool idle = S.a_qlen[recon] == 0 || S.a_state[recon] == 0 || S.a_state[recon] == 3;
        const bool rec_done = ref_retired(rid, rs);
        const bool wrong_task = S.a_qlen[recon] > 0 && S.a_qid[recon][0] != rid;
        retire = dead || idle || rec_done || wrong_task;
      }
    };
    if (lane < n) {
      recon = S.esc_agent[lane]; es = S.esc_slot[lane]; eid = S.esc_id[lane];
      rid = S.esc_pid[lane]; rs = S.esc_pslot[lane];
      esc_here = ref_valid(eid, es);  // an escort that expired by its hard window keeps its map entry (status 2, never popped) ...
      esc_live = esc_here && S.t_status[es] != 2;
    }
    decide();
    // With a RECON type among escort_agent_types a protected UAV can itself hold another UAV's escort task: retiring that escort
    // rewrites its queue (head, idle state), i.e. the verdict of ITS OWN entry further down the map.  Then the entries behind a
    // retirement are decided again on the new state, as the reference's loop sees them (found by tests/fuzz_device.py, config 7178).
    const bool recon_escorts = (P.escort_mask & ((1u << MUAVTA_R1) | (1u << MUAVTA_R2))) != 0u;
    const unsigned long long all = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
    unsigned long long rm = __ballot(retire), todo = all;
    PROF_COUNT(56, 1000); PROF_COUNT(57, 1000 * n);
    while (todo) {
      PROF_COUNT(58, 1000);
      const unsigned long long low = todo & (0ull - todo);                 // lowest pending entry
      const unsigned long long nextr = rm & todo;                         // pending retirements
      const unsigned long long seg = nextr ? (todo & ((nextr & (0ull - nextr)) - 1ull)) : todo;  // entries before the next one
      if (seg) {
        // _escort_fighters_near(recon, escort_radius) non-empty? (:1746-1764) for every entry of the segment.  A fighter heads
        // for ONE task, so it can only cover the entry whose escort task that is: each fighter lane finds its entry, runs one
        // distance test (`norm <= radius` on the squared form against the host-side threshold, as in the sensing pass),
        // and the entries collect their fighters' verdicts by ballot.
        const bool fighter = lane < P.n_agents && S.a_state[lane] != -1 && escort_type(S.a_type[lane]) && S.a_qlen[lane] > 0;
        const int head = fighter ? S.a_qid[lane][0] : -1;
        const unsigned long long live_m = __ballot(esc_live);
        int myk = -1, my_recon = 0;
        for (unsigned long long m = seg & live_m; m; m &= m - 1ull) {
          const int k = __ffsll((long long)m) - 1;
          const int ek = __builtin_amdgcn_readlane(eid, k), rk = __builtin_amdgcn_readlane(recon, k);
          if (head == ek) { myk = k; my_recon = rk; }
        }
        bool near = false;
        if (fighter && myk >= 0) {
          const double dx = S.a_px[lane] - S.a_px[my_recon], dy = S.a_py[lane] - S.a_py[my_recon];
          near = fma(dy, dy, dx * dx) <= P.escort_sq_bound;
        }
        unsigned long long cov = 0ull;
        const int neark = near ? myk : -2;
        if (__ballot(near) != 0ull)
          for (unsigned long long m = seg & live_m; m; m &= m - 1ull) {
            const int k = __ffsll((long long)m) - 1;
            if (__builtin_amdgcn_ballot_w64(neark == k) != 0ull) cov |= 1ull << k;
          }
        if ((seg >> lane) & 1ull) {
          if (esc_here) { S.t_px[es] = S.a_px[recon]; S.t_py[es] = S.a_py[recon]; }  // ... and follows the protected UAV (:1995); only coverage asks for status != 2 (:1751)
        }
        if (lane == 0) { S.escort_required_steps += __popcll(seg); S.escort_covered_steps += __popcll(cov); }
        todo &= ~seg;
        lds_sync();
        PROF(60);
      } else {
        const int k = __ffsll((long long)low) - 1;
        const int rk = __builtin_amdgcn_readlane(recon, k), ek = __builtin_amdgcn_readlane(eid, k);
        const bool dk = (__ballot(dead) >> k) & 1ull;
        retire_escort_coop(rk, dk);
        (void)ek;
        todo &= ~low;
        if (recon_escorts && todo) {
          lds_sync();
          decide();
          rm = __ballot(retire);
        }
        PROF(61);
      }
    }
  }

  // _wps_update_sensing (:1506-1523).  First the wave compacts the slots that can be sensed at all (dynamic,
  // open) by ballot; then lane = (agent a, candidate sub-row) with the agent's position in registers.  The
  // test `norm(d) <= sense_radius` is done on the squared form: sqrt is monotone and correctly rounded, so
  // it equals `fma(dy,dy,dx*dx) <= bound` with the bound precomputed on the host (no sqrt per pair).
  DEV void sense_parallel() {
    if (P.sense_radius <= 0) return;
    const int nA = P.n_agents, nO = S.n_order;
    int16_t* cand = X.roundT;  // T entries
    int nc = 0;
    for (int base = 0; base < nO; base += WG) {
      const int k = base + lane;
      int s = -1;
      bool c = false;
      if (k < nO) {
        s = S.t_order[k];
        const int fl = S.t_flags[s];
        c = S.t_status[s] != 2 && !(fl & TF_KNOWN_ALL) && (S.t_created[s] > 0 || (fl & TF_DEADLINE));
      }
      const unsigned long long m = __ballot(c);
      if (c) cand[nc + prefix_count(m)] = s;
      nc += __popcll(m);
    }
    if (nc == 0) return;
    lds_sync();
    int aw = 1;
    while (aw < nA) aw <<= 1;               // 16 / 32 / 64
    const int a = lane & (aw - 1), sub = lane / aw, stride = WG / aw;
    if (!(a < nA && S.a_state[a] != -1)) return;
    const double ax = S.a_px[a], ay = S.a_py[a];
    for (int k = sub; k < nc; k += stride) {
      const int s = cand[k];
      if ((S.known[a][s >> 5] >> (s & 31)) & 1u) continue;
      const double dx = ax - S.t_px[s], dy = ay - S.t_py[s];
      if (fma(dy, dy, dx * dx) <= P.sense_sq_bound) atomicOr(&S.known[a][s >> 5], 1u << (s & 31));
    }
  }


  // _wps_process_reveals (:1525-1541) and _wps_expire_windows (:1557-1573), wave-cooperative: due reveals and expired
  // windows are found by ballot and handled in list order; a reveal sets one known bit per agent lane, an expiry
  // frees the agents heading the task (found by ballot, desallocateAll on lane 0 in agent order — the f64 order of
  // removeAgentCap and the list-mutation quirk stay the reference's).
  DEV void process_lists_coop(bool any_due, bool any_exp) {
    const unsigned long long below = (1ull << lane) - 1ull;
    if (any_due) {
      const int n = S.n_pending;
      int w = 0;
      for (int base = 0; base < n; base += WG) {
        const int k = base + lane;
        int pt = 0, pid = -1, psl = 0;
        typename KnowMask<A>::type pk = 0;
        if (k < n) { pt = S.pend_time[k]; pid = S.pend_id[k]; psl = S.pend_slot[k]; pk = S.pend_know[k]; }
        const bool due = k < n && tnow >= pt;
        const unsigned long long dm = __ballot(due), km = __ballot(k < n && !due);
        if (P.share_knowledge) {
          for (unsigned long long m = dm; m; m &= m - 1ull) {
            const int b = __ffsll((long long)m) - 1;
            const int id = __builtin_amdgcn_readlane(pid, b), sl = __builtin_amdgcn_readlane(psl, b);
            if (ref_valid(id, sl)) {
              if (lane < P.n_agents) S.known[lane][sl >> 5] |= 1u << (sl & 31);
              if (lane == 0) S.t_flags[sl] |= TF_KNOWN_ALL;
            } else if (lane < P.n_agents) {  // released before the reveal: the id still joins the set of everyone who had not sensed it
              const unsigned long long kn = (unsigned long long)S.pend_know[base + b];
              S.a_gone[lane] += !((kn >> lane) & 1ull);
            }
          }
        }
        lds_sync();  // every lane holds its entry before the survivors are packed to the front
        if (k < n && !due) {
          const int d = w + prefix_count(km);
          S.pend_time[d] = pt; S.pend_id[d] = pid; S.pend_slot[d] = (uint8_t)psl; S.pend_know[d] = pk;
        }
        w += __popcll(km);
      }
      if (lane == 0) S.n_pending = w;
      lds_sync();
    }
    if (any_exp) {
      const int n = S.n_order;
      for (int base = 0; base < n; base += WG) {
        const int k = base + lane;
        int sl = -1;
        bool ex = false;
        if (k < n) {
          sl = S.t_order[k];
          ex = (S.t_flags[sl] & TF_DEADLINE) && S.t_status[sl] != 2 && tnow > S.t_deadline[sl];
        }
        for (unsigned long long m = __ballot(ex); m; m &= m - 1ull) {
          const int b = __ffsll((long long)m) - 1;
          const int s_ = __builtin_amdgcn_readlane(sl, b);
          const int id = S.t_id[s_];
          const unsigned long long hm = __ballot(lane < P.n_agents && S.a_qlen[lane] > 0 && S.a_qid[lane][0] == id);
          if (lane == 0) {
            S.t_status[s_] = 2;
            mark_outcome_slot(s_, false);
            if (!(S.t_flags[s_] & TF_REACHED)) { S.t_flags[s_] |= TF_REACHED; S.n_reached++; }
            for (unsigned long long h = hm; h; h &= h - 1ull) desallocate_all(__ffsll((long long)h) - 1);
          }
          lds_sync();  // the next expiry looks at the queues this one just changed
        }
      }
    }
  }
  // reserve tracking (:1575-1580), pending-reset latch (:1156-1160), shared reward (:1162-1178), done flags
  DEV void step_serial_c(double action_reward, double distance_reward, double quality_reward, double S_quality_reward,
                         int idle, bool responding, bool all_done_tasks) {
    // every LDS operand first (one wait), then arithmetic, then the stores
    const int irs = S.idle_reserve_steps, pr = S.pending_reset, ntid = S.next_task_id, ct = S.conclusion_time;
    const double time_penaulty = S.r_time_penalty, alloc_reward = S.r_alloc, step_reward = S.step_reward, FR = S.F_Reward;
    PROF(36);
    // Most steps add nothing to any reward term (no action applied, no task concluded): with finite weights >= 0 every product is then
    // +0.0, their sum is +0.0 and so is the quotient — the 15 dependent multiply-adds and two divisions on lane 0's chain are skipped.
    // (bit patterns, not values: a -0.0 term would make the reference's sum depend on the order of the signed zeros)
    const unsigned long long any_bits = (unsigned long long)__double_as_longlong(action_reward) | (unsigned long long)__double_as_longlong(distance_reward) |
                                        (unsigned long long)__double_as_longlong(quality_reward) | (unsigned long long)__double_as_longlong(S_quality_reward) |
                                        (unsigned long long)__double_as_longlong(alloc_reward) | (unsigned long long)__double_as_longlong(time_penaulty) |
                                        (unsigned long long)__double_as_longlong(step_reward);
    const bool quiet_reward = P.rw_plain && any_bits == 0ull;
    double total = 0.0;
    if (!quiet_reward)
    total = P.rw[0] * action_reward + P.rw[1] * distance_reward + P.rw[2] * quality_reward + P.rw[3] * S_quality_reward +
                   P.rw[4] * (double)P.n_tasks * 0.0 + P.rw[5] * alloc_reward + P.rw[6] * time_penaulty + P.rw[7] * step_reward;
    PROF(37);
    // two divisions on lane 0's chain in EVERY step: the range-restricted sequence (8 VALU each, bit-identical to IEEE inside its
    // domain: `total` is 0 or a sum of reward terms of ordinary magnitude, the divisors are a positive constant >= 0.002 and a step
    // count) instead of the compiler's ~35-instruction expansion.  A configuration without static tasks has reward_norm_factor == 0
    // (the reference raises ZeroDivisionError in its first step): that one keeps the plain division.
    const double shared = quiet_reward ? 0.0
                          : P.reward_norm_factor != 0 ? fdiv(fdiv(total, P.reward_norm_factor), (double)P.max_time_steps)
                                                      : total / P.reward_norm_factor / (double)P.max_time_steps;
    PROF(38);
    const bool all_done = (ntid > 1) && all_done_tasks;
    const bool timed_out = (tnow >= P.max_time_steps) && (P.max_time_steps > 0);
    const bool done = timed_out || (P.early_terminate && all_done);
    S.idle_reserve_steps = irs + idle;
    if (pr && responding) S.pending_reset = 0;
    if (all_done && ct > P.max_time_steps) S.conclusion_time = tnow;
    S.terminated = P.early_terminate && all_done && !timed_out;
    S.truncated = timed_out;
    S.last_reward = done ? FR : shared;  // :1202
  }

  // ---------------------------------------------------------------- end of step: slot GC + open list
  // Retired slots are recycled unless a LIVE agent still queues them (their type/position is read by
  // the switch penalty and expected-distance terms, :852-861,:1219-1220).  Parallel over slots.
  DEV void finish_step_parallel(bool gc) {
    if (gc) {
      // fast path: no live slot is retired and every live slot is already listed as open (no task was created
      // or retired since the lists were built) -> t_order / last_tasks_info are unchanged
      // (... and the two lists really are the same list: with the tile FULL, a task concluded in this step can have had its slot
      // recycled on demand for a task created in the same step — no retired slot is left and the counts agree, but that slot now
      // sits at the END of t_order and still at its old row of last_tasks_info.  Found by tests/fuzz_device.py, config 1000488.)
      const int n = S.n_order, no = S.n_open;
      bool ret = false, diff = false;
      for (int k = lane; k < n; k += WG) {
        const int s = S.t_order[k];
        ret |= S.t_status[s] == 2;
        diff |= k >= no || (int)S.open_slot[k] != s;
      }
      const bool any_ret = __ballot(ret) != 0ull;
      if (!any_ret && no == n && __ballot(diff) == 0ull) {
        if (lane == 0) S.n_act = 0;
        lds_sync();
        return;
      }
#ifdef MUAVTA_PROF
      PROF_COUNT(30, 1000); if (any_ret) PROF_COUNT(31, 1000);
#endif
      if (any_ret) {
        if (rel_log) cold_sync();
        // slots still queued by a live agent (one agent per lane marks its queue entries), then the retired and
        // unreferenced ones are released (one slot per lane), then every agent lane drops the released columns
        // from its known mask and counts them into a_gone
        uint32_t* refmask = reinterpret_cast<uint32_t*>(X.path);
        uint32_t* relmask = refmask + KW;
        if (lane < 2 * KW) refmask[lane] = 0;
        lds_sync();
        if (lane < P.n_agents && S.a_state[lane] != -1) {
          const int ql = S.a_qlen[lane];
          for (int k = 0; k < ql; k++) {
            const int qs = S.a_qslot[lane][k];
            if (ref_valid(S.a_qid[lane][k], qs)) atomicOr(&refmask[qs >> 5], 1u << (qs & 31));
          }
        }
        lds_sync();
        for (int k = lane; k < n; k += WG) {
          const int s = S.t_order[k];
          if (S.t_status[s] == 2 && !((refmask[s >> 5] >> (s & 31)) & 1u)) {
            release_slot_record(s);
            atomicOr(&relmask[s >> 5], 1u << (s & 31));
          }
        }
        lds_sync();
        if (lane < P.n_agents) {
          int gone = 0;
          for (int w = 0; w < KW; w++) {
            const uint32_t m = relmask[w];
            if (m) { const uint32_t old = S.known[lane][w]; gone += __popc(old & m); S.known[lane][w] = old & ~m; }
          }
          if (gone) S.a_gone[lane] += gone;
        }
        if (lane < KW) S.free_slots[lane] |= relmask[lane];
        lds_sync();
      }
      PROF(35);
    }
    {  // compact t_order (drop freed slots) and rebuild last_tasks_info (:492) with ballot + popcount
      const int n = S.n_order, no_old = S.n_open;
      int w = 0, no = 0;
      bool moved = false;  // some row of last_tasks_info holds another task than before (the observation rows' static columns follow the list)
      for (int base = 0; base < n; base += WG) {
        const int k = base + lane;
        int s = -1;
        bool alive = false, open = false;
        if (k < n) {
          s = S.t_order[k];
          alive = S.t_id[s] >= 0;
          open = alive && S.t_status[s] != 2;
        }
        const unsigned long long am = __ballot(alive), om = __ballot(open);
        const unsigned long long below = (1ull << lane) - 1ull;
        lds_sync();  // every lane has read its t_order entry before any is overwritten
        if (alive) S.t_order[w + prefix_count(am)] = s;
        if (open) {
          const int row = no + prefix_count(om);
          moved |= row >= no_old || (int)S.open_slot[row] != s;  // (a row is only ever rewritten by the lane that holds its new content: rows ascend with k)
          S.open_slot[row] = s; S.t_row[s] = (uint8_t)row;
        }
        w += __popcll(am);
        no += __popcll(om);
      }
      const bool list_changed = __ballot(moved) != 0ull || no != no_old;
      if (lane == 0) { S.n_order = w; S.n_open = no; S.n_act = 0; S.list_stale = 0; if (list_changed) obs_static_clear(); }
    }
    lds_sync();
  }

  // ====================================================================================================
  // Observation tensors (:365-415,:468-492), written straight to HBM by all lanes.
  // ====================================================================================================
  // Observation tensors are FEATURE-MAJOR in HBM — tasks f32 [21][max_tasks], legal bit rows u64 [A][ceil(max_tasks/64)] —
  // so with one task row per lane every store instruction is a contiguous run across the wave: no LDS
  // staging, no transposition.  The "no legal action" fallback (:401-408) is a ballot.
  // Task.initTime / doneTime of every live, non-retired slot from the agents' queue entries (== the task's
  // allocationDetails): times are non-negative doubles, so u64 min/max on their bit patterns (LDS atomics)
  // order them numerically.  One queue entry per lane.
  // The caller has passed a cold_sync() (the queue-time rows other lanes wrote are visible).  Besides updating the HBM
  // rows it leaves (initTime, doneTime) of every slot in the scratch tile (obs_times()), where the observation rows of
  // this step pick them up without another trip to memory.  Returns whether it ran.
  DEV double* obs_times() { return X.cost; }  // [2][T] after refresh_task_times() returned true
  // x / c, correctly rounded, for a divisor that is a small integer times a power of two (1200, 6, max_time_steps,
  // max_tasks <= 2^15) and inv = RN(1 / c): q = RN(x * inv) is within 2 ulp of x / c; r = x - q * c is exact in the FMA (q * c
  // has <= 53 + 15 bits and cancels against x to a few units); q + r * inv then differs from x / c by < 2^-51 ulp, and x / c is
  // never that close to a rounding boundary without being on it: both are multiples of 2^g (g = exponent of half an ulp
  // of the quotient) divided by c's odd part m, so they are >= 2^g / m apart, and x / c cannot BE a midpoint because x's
  // last bit sits >= 6 binary places above g.  Three full-rate FMAs instead of the ~12-instruction IEEE division
  // sequence with its quarter-rate v_rcp_f64 (the observation writer divides ~10 times per step and the kernel is
  // VALU-issue bound).  x must be finite (div_small_any keeps inf / NaN as the true division would) and not -0.0 (the
  // result would be +0.0: callers with a negated operand negate the quotient instead, which IEEE division commutes with).
  // *(p + byte_off) for a wave-uniform p, pinned to scalar registers so that the access is emitted in the (SGPR base,
  // 32-bit VGPR offset) addressing form and not with a 64-bit VALU add per access.
  template <class V> static DEV V& at_lane(V* p, uint32_t byte_off) {
#if !MUAVTA_OBS_SADDR
    return *(V*)((char*)p + byte_off);
#endif
    typedef __attribute__((address_space(1))) char GC;
    GC* q = (GC*)p;  // stays a global (not flat) access
    asm("" : "+s"(q));
    return *(V*)(__attribute__((address_space(1))) V*)(q + byte_off);  // zext(32-bit byte offset): what the addressing form takes
  }
  static DEV double div_small(double x, double c, double inv) {
    const double q = x * inv;
    return __builtin_fma(__builtin_fma(-q, c, x), inv, q);
  }
  static DEV double div_coord(double x) { return div_small(x, MAX_COORD, 1.0 / MAX_COORD); }  // x / MAX_COORD, x finite
  static DEV double div40(double x) { return div_small(x, 40.0, 1.0 / 40.0); }  // remaining / 40.0 of the urgency terms (x: a small non-negative integer)
  static DEV double div_small_any(double x, double c, double inv) {
    const double q = x * inv;
    const double q1 = __builtin_fma(__builtin_fma(-q, c, x), inv, q);
    return __builtin_isfinite(q) ? q1 : q;
  }
  DEV bool refresh_task_times() {
    if (!S.times_dirty) return false;  // uniform: LDS word
    lds_sync();
    if (lane == 0) S.times_dirty = 0;
    unsigned long long* tmin = reinterpret_cast<unsigned long long*>(X.cost);
    unsigned long long* tmax = tmin + T;
    for (int s = lane; s < T; s += WG) { tmin[s] = ~0ull; tmax[s] = 0ull; }
    lds_sync();
    if constexpr (A > 32) {
      // 64 agents x Q entries: one AGENT per lane, as many passes as the longest queue holds entries (3-5, not Q = 12), and the
      // lane's queue-time row (HBM) is requested in one go in front of them — one memory latency instead of one per pass.
      const int a = lane < P.n_agents ? lane : 0;
      const int ql = lane < P.n_agents ? (int)S.a_qlen[a] : 0;
      double qt[Q];
#pragma unroll
      for (int k = 0; k < Q; k++) qt[k] = k < ql ? C.a_qtime[a][k] : 0.0;
#pragma unroll
      for (int k = 0; k < Q; k++) {
        if (__ballot(k < ql) == 0ull) break;
        if (k < ql) {
          const int id = S.a_qid[a][k], slot = S.a_qslot[a][k];
          if (ref_valid(id, slot) && S.t_status[slot] != 2) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(qt[k]);
            atomicMin(&tmin[slot], bits);
            atomicMax(&tmax[slot], bits);
          }
        }
      }
    } else {
      const int nE = P.n_agents * Q;
      for (int e = lane; e < nE; e += WG) {
        const int a = e / Q, k = e - a * Q;
        if (k < S.a_qlen[a]) {
          const int id = S.a_qid[a][k], slot = S.a_qslot[a][k];
          if (ref_valid(id, slot) && S.t_status[slot] != 2) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(C.a_qtime[a][k]);
            atomicMin(&tmin[slot], bits);
            atomicMax(&tmax[slot], bits);
          }
        }
      }
    }
    lds_sync();
    double* tt = obs_times();
    for (int s = lane; s < T; s += WG) {
      double ti, td;
      if (S.t_id[s] >= 0 && S.t_status[s] != 2) {
        if (tmin[s] == ~0ull) { ti = -1; td = -1; }
        else {
          ti = __longlong_as_double((long long)tmin[s]);
          td = __longlong_as_double((long long)tmax[s]) + (double)task_duration(S.t_type[s]);
        }
        C.t_init[s] = ti; C.t_dtime[s] = td;
      } else { ti = -1; td = -1; }  // retired / free slots: no observation row reads them
      tt[s] = ti; tt[T + s] = td;
    }
    lds_sync();
    return true;
  }

  // handle_buffer: the tensors are the handle's single observation buffer (overwritten in place every step), not a ring slot —
  // its pad rows are only rewritten where live rows have just disappeared (S.obs_rows)
  DEV void write_obs(float* o_tasks, unsigned long long* o_legal, uint8_t* o_pad, float* o_agents, float* o_flags, bool handle_buffer = false) {
#ifdef MUAVTA_OBS_SKIP  // timing experiments only (results are wrong): bit 0 task rows, 1 legal mask, 2 agent rows + flags, 3 pad flags, 4 everything
    if (MUAVTA_OBS_SKIP & 16) return;
    if (MUAVTA_OBS_SKIP & 1) o_tasks = nullptr;
    if (MUAVTA_OBS_SKIP & 2) o_legal = nullptr;
    if (MUAVTA_OBS_SKIP & 4) { o_agents = nullptr; o_flags = nullptr; }
    if (MUAVTA_OBS_SKIP & 8) o_pad = nullptr;
#endif
    cold_sync();  // the rows below read the requirement vectors the serial phases of this step may have changed
    const int MT = P.max_tasks, nA = P.n_agents;
    const int n = __builtin_amdgcn_readfirstlane(S.n_open);
    const int obs_state = handle_buffer ? __builtin_amdgcn_readfirstlane(S.obs_rows) : -1;
    const int pad_known = obs_state < 0 ? -1 : (obs_state & (OBS_STATIC - 1));
    const int pad_until = pad_known < 0 ? MT : pad_known;  // pad rows below this index have to be (re)written
    // (r4) LIGHT pass: the handle's buffer already holds the static columns of every open row (OBS_STATIC: no requirement /
    // allocation vector and no open-list change since the last write) — only the columns that move every step are rewritten:
    // position (Int and escort tasks follow their threat / UAV), status, init / end time (relative to t), age.  6 stores and 2
    // HBM operands per row instead of 21 and 16.
    const bool light = o_tasks != nullptr && obs_state > 0 && (obs_state & OBS_STATIC) != 0;
    // The HBM rows of the first 64 observation rows are requested NOW, so that their latency runs under the rebuild of the
    // task times below (LDS work); (initTime, doneTime) then come from the scratch tile when they were rebuilt.
    double pc[6], pa[6], pti = 0, ptd = 0;
    const bool dirty = __builtin_amdgcn_readfirstlane(S.times_dirty) != 0;  // (uniform: a scalar branch picks LDS or HBM below, not a per-lane pointer select)
    if (MUAVTA_OBS_PREFETCH && !light) {
      const int s0 = lane < n ? (int)S.open_slot[lane] : 0;
#pragma unroll
      for (int c = 0; c < 6; c++) { pc[c] = lane < n ? C.t_cur[c][s0] : 0.0; pa[c] = lane < n ? C.t_alloc[c][s0] : 0.0; }
      if (!dirty && lane < n) { pti = C.t_init[s0]; ptd = C.t_dtime[s0]; }
    }
    refresh_task_times();
    const double mts = (double)(P.max_time_steps > 1 ? P.max_time_steps : 1), inv_mts = P.inv_mts;
    constexpr double INV_COORD = 1.0 / MAX_COORD, INV6 = 1.0 / 6.0;
    unsigned long long leg0 = 0ull, leg1 = 0ull;  // lane a: legal bits of agent a (rows 0..63, 64..127)
    PROF(15);
    // lane a keeps agent a's (state, head id, type); the agent loop broadcasts them with v_readlane
    int my_st = 0, my_hid = 0, my_ty = 0;
    if (lane < nA) { my_st = S.a_state[lane]; my_hid = head_id(lane); my_ty = S.a_type[lane]; }
    const bool capm = P.capability_mask != 0;
    // (the reference's lists are max_tasks long unless MORE tasks are open — then they grow, DroneEnv.py:410-413 — and its "no legal
    // action" fallback looks at all of them: the loop covers every open row, the stores the first max_tasks)
    const int row_lim = n > MT ? n : MT;
    for (int base = 0; base < row_lim; base += WG) {
      const int j = base + lane;
      const uint32_t ju = (uint32_t)j;  // unsigned lane offset + uniform column pointer: stores take the (SGPR base, VGPR offset) form
      const bool in_n = j < n, in_mt = j < MT;
      int ty = 0;
      uint32_t typemask = 0;  // agent types for which this row is a valid action (before the capability mask)
      if (in_n) {
        const int s = S.open_slot[j];
        const int tid = S.t_id[s];
        ty = S.t_type[s];
        typemask = (S.t_flags[s] & TF_ELIGIBLE) ? S.t_elig[s] : 0xffffffffu;
        if (light) {  // (uniform)
          // (OBS_STATIC implies the task times are not dirty.  These two HBM operands sit on the step's chain: a build without them ran
          // 1.2 % faster, requesting them at the start of step() instead won back 0.3 % — within noise, not adopted: profiles/r04_ab_obs_light_prefetch.txt)
          const double ti = C.t_init[s], td = C.t_dtime[s];
          if (P.saturate_mask && C.t_alloc[ty][s] >= qs().t_org[s]) typemask = 0;
          uint32_t off = ju * 4u;
          const uint32_t cstride = (uint32_t)MT * 4u;
          // (more open tasks than max_tasks: the tensor holds the first max_tasks rows; rows beyond it only feed the legal-mask ballots)
          if (in_mt) {
          if (ty == MUAVTA_INT || (S.t_flags[s] & TF_ESCORT)) {  // the tasks that move: an Int task follows its threat, an escort its UAV
            at_lane(o_tasks, off + cstride) = (float)div_small(S.t_px[s], MAX_COORD, INV_COORD);
            at_lane(o_tasks, off + 2u * cstride) = (float)div_small(S.t_py[s], MAX_COORD, INV_COORD);
          }
          at_lane(o_tasks, off + 3u * cstride) = (float)S.t_status[s];
          if (P.include_time_windows) {
            at_lane(o_tasks, off + 16u * cstride) = (float)div_small_any(ti - (double)tnow, mts, inv_mts);
            at_lane(o_tasks, off + 17u * cstride) = (float)div_small_any(td - (double)tnow, mts, inv_mts);
          }
          at_lane(o_tasks, off + 20u * cstride) = (float)fmin(div_small((double)tnow - (double)S.t_created[s], mts, inv_mts), 1.0);
          }
        } else {
        float r[21];
        r[0] = (float)tid;
        r[1] = (float)div_small(S.t_px[s], MAX_COORD, INV_COORD);
        r[2] = (float)div_small(S.t_py[s], MAX_COORD, INV_COORD);
        r[3] = (float)S.t_status[s];
        double cur[6], alc[6], ti, td;
        if (MUAVTA_OBS_PREFETCH && base == 0) {
#pragma unroll
          for (int c = 0; c < 6; c++) { cur[c] = pc[c]; alc[c] = pa[c]; }
          if (dirty) { ti = obs_times()[s]; td = obs_times()[T + s]; } else { ti = pti; td = ptd; }
        } else {
#pragma unroll
          for (int c = 0; c < 6; c++) { cur[c] = C.t_cur[c][s]; alc[c] = C.t_alloc[c][s]; }
          if (dirty) { ti = obs_times()[s]; td = obs_times()[T + s]; } else { ti = C.t_init[s]; td = C.t_dtime[s]; }
        }
#pragma unroll
        for (int c = 0; c < 6; c++) { r[4 + c] = (float)cur[c]; r[10 + c] = (float)alc[c]; }
        r[16] = r[17] = r[18] = 0.f;
        if (P.include_time_windows) {
          r[16] = (float)div_small_any(ti - (double)tnow, mts, inv_mts);
          r[17] = (float)div_small_any(td - (double)tnow, mts, inv_mts);
          r[18] = (float)div_small((double)ty, 6.0, INV6);
        }
        // the row of the task's own type: two more loads with a per-lane row index (a select chain over the six rows loaded above
        // costs ~30 VALU; a dynamically indexed local array would live in scratch memory)
        const double cur_ty = C.t_cur[ty][s], alc_ty = C.t_alloc[ty][s];
        const double org = qs().t_org[s];
        if (P.saturate_mask && alc_ty >= org) typemask = 0;
        const double unmet = fmax(cur_ty - alc_ty, 0.0);
        r[19] = (float)fdiv(unmet, fmax(org, 1e-6));  // (unmet: 0 or a difference of capability sums, org >= 1e-6: inside fdiv's domain)
        r[20] = (float)fmin(div_small((double)tnow - (double)S.t_created[s], mts, inv_mts), 1.0);
        if (o_tasks && in_mt) {
          // feature-major: column c of row j at (c * MT + j) * 4 — one base in scalar registers, the lane's byte offset stepped by
          // the column stride (one VALU add per store; a column pointer per store was five SALU instructions)
          uint32_t off = ju * 4u;
          const uint32_t cstride = (uint32_t)MT * 4u;
#pragma unroll
          for (int c = 0; c < 21; c++) { at_lane(o_tasks, off) = r[c]; off += cstride; }
        }
        }
      } else if (o_tasks && in_mt && (j < pad_until || j == 0)) {
        // pad rows are {"status": -1}; with no open task row 0 is task_idle (all zeros).  Rows the buffer already holds as pad rows
        // are left alone.
        const float st = (j == 0 && n == 0) ? 0.f : -1.f;
        uint32_t off = ju * 4u;
        const uint32_t cstride = (uint32_t)MT * 4u;
#pragma unroll
        for (int c = 0; c < 21; c++) { at_lane(o_tasks, off) = c == 3 ? st : 0.f; off += cstride; }
      }
      if (o_pad && in_mt && !light) o_pad[ju] = j < (n == 0 ? 1 : n);  // (light: the open list, and with it the pad mask, is what the buffer holds)
      PROF(16);
      if (o_legal) {
        // legal_mask without a per-agent loop: one ballot per agent TYPE gives the rows that type may take
        // (_is_task_action_valid :341-363, eligibility + saturation); lane a then picks its type's row mask.
        unsigned long long okm = 0ull;
#pragma unroll
        for (int t = 0; t <= MUAVTA_F2; t++) {
          const unsigned long long m = __builtin_amdgcn_ballot_w64((typemask & (1u << t)) != 0u);  // (typemask is 0 beyond the open rows)
          if (my_ty == t) okm = m;
        }
        if (capm) {  // capability mask: rows whose task type the agent has capability for
          const int tyx = in_n ? ty : -1;
          unsigned long long capok = 0ull;
#pragma unroll
          for (int tt = 0; tt < 6; tt++) {
            const unsigned long long m = __builtin_amdgcn_ballot_w64(tyx == tt);
            if (lane < nA && S.a_caps[tt][lane] > 0) capok |= m;
          }
          okm &= capok;
        }
        if (lane < nA) { if (base == 0) leg0 = okm; else leg1 = okm; }
      }
      PROF(18);
    }
    if (o_legal && lane < nA) {
      // the agent's current task is always selectable (:346-348); in state 2 it is the ONLY legal row (:475-479)
      unsigned long long cur0 = 0ull, cur1 = 0ull;
      int curpos = -1;
      if (n == 0) {
        curpos = my_hid == 0 ? 0 : -1;   // single task_idle row
        if (my_st != 2) { leg0 = 1ull; }
      } else if (S.a_qlen[lane] > 0) {
        const int hs = S.a_qslot[lane][0];
        if (ref_valid(my_hid, hs) && S.t_status[hs] != 2) curpos = S.t_row[hs];
      }
      if (curpos >= 0) { if (curpos < 64) cur0 = 1ull << curpos; else cur1 = 1ull << (curpos - 64); }
      if (my_st == 2) { leg0 = cur0; leg1 = cur1; }
      else if (n > 0) { leg0 |= cur0; leg1 |= cur1; }
      // "no legal action" fallback (:401-408): the current task if it is open, else row 0
      if ((leg0 | leg1) == 0ull && n > 0 && my_st != 2) leg0 = 1ull;
      if (n > MT) {  // rows beyond the tensor's width took part in the fallback above and are dropped here
        if (MT < 64) { leg0 &= (1ull << MT) - 1ull; leg1 = 0ull; }
        else if (MT < 128) leg1 &= (1ull << (MT - 64)) - 1ull;
      }
      const int KM = (MT + 63) >> 6;
      o_legal[lane * KM] = leg0;
      if (KM > 1) o_legal[lane * KM + 1] = leg1;
    }
    PROF(19);
    if (o_agents && lane < nA) {
      const int a = lane;
      float* r = o_agents + a * 9;
      r[0] = (float)div_small(S.a_px[a], MAX_COORD, INV_COORD);
      r[1] = (float)div_small(S.a_py[a], MAX_COORD, INV_COORD);
#pragma unroll
      for (int c = 0; c < 6; c++) r[2 + c] = (float)S.a_caps[c][a];
      r[8] = (float)head_id(a);
    }
    if (o_flags && lane == 0) {  // _event_flag_vector (:417-438)
      float f = 0, t = 0, r = 0;
      for (int k = 0; k < S.n_events; k++) {
        if (S.ev_tag[k] == MUAVTA_EV_AGENT_FAIL) f = 1; else if (S.ev_tag[k] == MUAVTA_EV_NEW_THREAT) t = 1;
        else if (S.ev_tag[k] == MUAVTA_EV_RESET_ALLOCATION) r = 1;
      }
      o_flags[0] = f; o_flags[1] = t; o_flags[2] = r;
      o_flags[3] = (float)div_small((double)tnow, mts, inv_mts);
      o_flags[4] = (float)div_small((double)n, (double)(P.max_tasks > 1 ? P.max_tasks : 1), P.inv_max_tasks);
    }
    if (handle_buffer && o_tasks && lane == 0) S.obs_rows = (n > 0 ? n : 1) | OBS_STATIC;  // rows from here on hold pad rows now; the open rows' static columns are current
  }

  // ====================================================================================================
  // HungarianAllocator.allocate_tasks (TaskAllocation/OptimizationBased/HungarianAllocator.py:72-208)
  // + _open_tasks / _apply_assign glue (experiments/paper_eval.py:85-101, wps_eval.py:55-61).
  // Result: S.act_agent / S.act_slot / S.act_index, S.n_act.
  // ====================================================================================================
  DEV bool is_escort_task(int s) const { return (S.t_flags[s] & TF_ESCORT) || S.t_required[s] > 0; }
  DEV double residual_demand(int s) const {
    if (is_escort_task(s)) {
      double required = S.t_required[s] ? (double)S.t_required[s] : 1.0;
      return fmax(required - (double)S.t_ndet[s], 0.0);
    }
    int ty = S.t_type[s];
    return fmax(C.t_cur[ty][s] - C.t_alloc[ty][s], 0.0);
  }

  // mode 0 (MUAVTA_ALLOC_HUNGARIAN): Local-/Global-/Coalition-Hungarian as driven by the WPS / escort harness.
  // mode 1 (MUAVTA_ALLOC_URGENCY_PAIR): TaskAllocation/Hybrid/PairCostHybrid.py:31-86,520-550 — engineered
  //   float32 edge scores 0.5*urgency + 0.3*scarcity - 0.4*dist, clipped to +-0.35, for the first 16 live agents x
  //   first 32 underfilled tasks, subtracted from the Hungarian cost; replan gate = experiments/wps_eval.py:64-74.
  DEV int16_t* pair_info() { return T > 64 ? X.pair_info_big : X.remaining; }
  // mode 4 (SC, muavta_allocate_scored): the caller's edge scores / task priorities / reserved agents, indexed in the token layout
  //   (sc.kind, sc.MT, sc.MA) — PairCostHybrid.plan, AttentionRAH.plan, AttentionCommit / AttentionEscort._plan_from_scores
  //   (HungarianAllocator.py:79-92,123-124,170-179).  sc_list: T bytes of LDS behind the scratch tile (the task list handed to
  //   allocate_tasks, as positions in last_tasks_info, in the token builder's order).  Compiled into k_allocate_scored only.
  // The replan gate of the callers' loops as a function of the env state (uniform): MUAVTA_GATE_* of include/muavta.h.  `events` are the
  // ones the last step drained (infos['events'], S.dev_*), the clock is the step about to be taken — what the reference's loops look at
  // between two env.step calls.  allocate<true> evaluates it at its entry; the run-to-the-next-gate kernels (k_run) after every step.
  DEV bool gate_fires(int gate, int interval) const {
    interval = interval < 1 ? 1 : interval;
    bool trig = false;
    for (int k = lane; k < S.n_dev; k += WG) trig |= S.dev_tag[k] <= MUAVTA_EV_AGENT_FAIL;  // Reset_Allocation, New_Threat, Agent_Fail
    if (gate == MUAVTA_GATE_TRAINER) return tnow == 0 || tnow % interval == 0 || __ballot(trig) != 0ull;   // train_pair_cost.py:33-43
    if (gate == MUAVTA_GATE_ESCORT) return tnow == 0 || tnow % interval == 0 || S.n_dev > 0;               // escort_eval.py:52-58
    if (gate == MUAVTA_GATE_ALLOCATOR) return (tnow - S.last_plan_step >= interval) || (S.n_dev > 0);      // should_replan (:27-41)
    return true;
  }
  template <bool SC = false>
  DEV void allocate(int interval, int use_visibility, int mode = 0, const ScoredDev* scp = nullptr, int env = 0, uint8_t* sc_list = nullptr) {
    PROF(10);
    interval = interval < 1 ? 1 : interval;
    if (lane == 0) S.n_act = 0;
    bool go;
    const bool envvis = !(P.sense_radius == 0 && P.threat_delay == 0);  // agent_visibility_map() is not None
    int n_sc = 0;  // SC: length of sc_list
    if constexpr (SC) {
      const ScoredDev& sc = *scp;
      go = gate_fires(sc.gate, interval);
      if ((go || sc.gate == MUAVTA_GATE_ALLOCATOR) && lane == 0) S.n_calls++;
      if (sc.replanned && lane == 0) sc.replanned[env] = go ? 1 : 0;
    } else
    if (mode == 1) {  // _should_replan(env, events, 15); plan(force=True) bypasses the allocator's own gate
      bool trig = false;
      for (int k = lane; k < S.n_dev; k += WG) trig |= S.dev_tag[k] <= MUAVTA_EV_AGENT_FAIL;  // Reset_Allocation, New_Threat, Agent_Fail
      go = tnow == 0 || tnow % 15 == 0 || __ballot(trig) != 0ull;
      if (go && lane == 0) S.n_calls++;
    } else if (mode == 3) {  // trainers' expert: force=True under _should_replan(env, events, interval) (train_pair_cost.py:33-43)
      bool trig = false;
      for (int k = lane; k < S.n_dev; k += WG) trig |= S.dev_tag[k] <= MUAVTA_EV_AGENT_FAIL;
      go = tnow == 0 || tnow % interval == 0 || __ballot(trig) != 0ull;
      if (go && lane == 0) S.n_calls++;
    } else if (mode == 2) {  // escort_eval._should_replan(env, events, interval): all five env tags are REPLAN_EVENTS
      go = tnow == 0 || tnow % interval == 0 || S.n_dev > 0;
      if (go && lane == 0) S.n_calls++;
    } else {
      if (lane == 0) S.n_calls++;
      // should_replan (:27-41): every tag the env emits is in the trigger set
      go = (tnow - S.last_plan_step >= interval) || (S.n_dev > 0);
    }
    if (go && lane == 0) S.gate_step = tnow + 1;
    const bool vis = use_visibility && envvis;  // agent_visibility_map() is None
    int nr = 0, n_live = 1;
    // The allocator's task list is env.tasks filtered at CALL time (_open_tasks, paper_eval.py:96-101): last_tasks_info as the last
    // observation left it, MINUS what was concluded since (the status tests below) PLUS what an out-of-step call created since — ids
    // are monotone, so those follow it in env.tasks order.  They become rows [n_open, n_plan) of open_slot for this plan only (beyond
    // n_open nobody else looks); a pair with such a task consumes its agent and the task's residual like any other and is then dropped,
    // as _apply_assign drops a task that is not in env.last_tasks_info (wps_eval.py:55-61).
    int n_plan = S.n_open;
    if (go && S.list_stale) {  // (uniform; only ever set by muavta_call)
      const int n0 = S.n_open;
      lds_sync();
      const int n_new = compact_to(S.open_slot + n0, S.n_order,
                                   [&](int k) { const int s = S.t_order[k]; const int r = S.t_row[s]; return S.t_status[s] != 2 && !(r < n0 && (int)S.open_slot[r] == s); },
                                   [&](int k) { return (int)S.t_order[k]; });
      lds_sync();
      for (int k = lane; k < n_new; k += WG) S.t_row[S.open_slot[n0 + k]] = (uint8_t)(n0 + k);
      n_plan = n0 + n_new;
      lds_sync();
    }
    if (go) {
      cold_sync();  // residual demand reads currentReqs / allocatedReqs
      if constexpr (SC) {
        const ScoredDev& sc = *scp;
        if (sc.kind == 2) {  // build_escort_tokens' sorted local open list -> sc_list; columns = its first MT entries
          const int n_live_raw = compact_to(X.freeA, P.n_agents, [&](int a) { return S.a_state[a] != -1; }, [&](int a) { return a; });
          const int n_all = compact_to(X.roundT, S.n_order, [&](int k) { const int s = S.t_order[k]; return S.t_status[s] != 2 && residual_demand(s) > 0; },
                                       [&](int k) { return S.t_order[k]; });
          lds_sync();
          const int n_list = escort_sorted_list(envvis, n_live_raw, n_all);
          const int n_cols = n_list < sc.MT ? n_list : sc.MT;  // columns = the first MT entries = the list handed over
          n_sc = n_cols;
          for (int k = lane; k < n_plan; k += WG) { const int s = S.open_slot[k]; pair_info()[s] = 255; X.resid[s] = 0.0; }  // (after the helper: pair_info may alias X.remaining)
          lds_sync();
          for (int j = lane; j < n_cols; j += WG) {
            const int s = X.path[j];
            pair_info()[s] = (int16_t)j; X.resid[s] = residual_demand(s); sc_list[j] = (uint8_t)S.t_row[s];
          }
          lds_sync();
        }
      }
      // live agents -> free list (get_live_agents order), residual demand per open task: one lane each
      // Urgency-Coalition holds committed agents out of the match (committed_names, AttentionCommit.py:24-30)
      unsigned long long held = 0ull;  // SC: reserved_agent_names (HungarianAllocator.py:91-92)
      if constexpr (SC) held = scp->reserved ? scp->reserved[env] : 0ull;
      nr = compact_to(X.freeA, P.n_agents,
                      [&](int a) { return S.a_state[a] != -1 && !(mode == 2 && S.a_commit[a] > tnow) &&
                                          !(SC && (((held >> a) & 1ull) || ((scp->flags & MUAVTA_SC_COMMIT) && S.a_commit[a] > tnow))); },
                      [&](int a) { return a; });
      n_live = nr > 1 ? nr : 1;
      bool any_open = false;
      int n_under = 0;
      if constexpr (SC) {
        const ScoredDev& sc = *scp;
        if (lane < P.n_agents) {  // token row of an agent: its rank among the live ones (tok["live"][:max_agents])
          const bool lv = S.a_state[lane] != -1;
          const unsigned long long lm = __ballot(lv);
          const int rk = prefix_count(lm);
          X.live_rank[lane] = (lv && rk < sc.MA) ? (uint8_t)rk : (uint8_t)255;
        }
        if (sc.kind != 2) {  // build_att_tokens' open_tasks (AttentionRAH.py:69-73): underfilled at the type index, env.tasks order
          const bool full = (sc.flags & MUAVTA_SC_FULL_TASK_LIST) != 0;
          for (int base = 0; base < n_plan; base += WG) {
            const int k = base + lane;
            bool under = false, inl = false;
            int s = 0, rank = 0;
            if (k < n_plan) {
              s = S.open_slot[k];
              const int ty = S.t_type[s];
              // (status: the open list is the one of the last observation — an out-of-step _retire_escort (muavta_call) concludes a task
              // behind it, and the harness lists are built from env.tasks with `status != 2` at call time)
              under = S.t_status[s] != 2 && C.t_alloc[ty][s] < C.t_cur[ty][s];
            }
            const unsigned long long um = __ballot(under);
            if (k < n_plan) {
              rank = n_under + prefix_count(um);
              inl = under && (full || rank < sc.MT);
              pair_info()[s] = (int16_t)(!under ? 255 : rank < sc.MT ? rank : 254);  // token column; 254: in the list, no column
              const double r = inl ? residual_demand(s) : 0.0;
              X.resid[s] = r;
              any_open |= r > 0;
            }
            const unsigned long long im = __ballot(inl);
            if (inl) sc_list[n_sc + prefix_count(im)] = (uint8_t)k;
            n_sc += __popcll(im);
            n_under += __popcll(um);
          }
        } else {
          any_open = n_sc > 0;  // every entry of the sorted list has residual demand
        }
      } else
      for (int base = 0; base < n_plan; base += WG) {
        const int k = base + lane;
        bool under = false;
        int s = 0;
        if (k < n_plan) {
          s = S.open_slot[k];
          const int ty = S.t_type[s];
          // Urgency-Pair only plans over build_att_tokens' open_tasks: underfilled at the type index (AttentionRAH.py:69-73).
          // _open_tasks (paper_eval.py:96-101) reads env.tasks when it is called: a task an out-of-step _retire_escort (muavta_call)
          // concluded since the last observation is still in `open_slot` and not in that list
          under = S.t_status[s] != 2 && (mode != 1 || C.t_alloc[ty][s] < C.t_cur[ty][s]);
        }
        const unsigned long long um = mode == 1 ? __ballot(under) : 0ull;
        if (k < n_plan) {
          // (Urgency-Pair hands the allocator the 32 token rows only: tok["open_tasks"] = kept, PairCostHybrid.py:36,62)
          const double r = (under && !(mode == 1 && n_under + prefix_count(um) >= 32)) ? residual_demand(s) : 0.0;
          X.resid[s] = r;
          any_open |= r > 0;
        }
        if (mode == 1) {
          if (under) {
            const int rank = n_under + prefix_count(um);
            int n_know = 0;
            if (vis) for (int b = 0; b < P.n_agents; b++) n_know += (S.known[b][s >> 5] >> (s & 31)) & 1u;
            pair_info()[s] = (rank < 32 ? rank : 255) | (n_know << 8);
          } else if (k < n_plan) pair_info()[s] = 255;
          n_under += __popcll(um);
        }
      }
      if (mode == 1 && lane < P.n_agents) {
        const bool lv = S.a_state[lane] != -1;
        const unsigned long long lm = __ballot(lv);
        const int rk = prefix_count(lm);
        X.live_rank[lane] = (lv && rk < 16) ? (uint8_t)rk : (uint8_t)255;
      }
      if (nr == 0 || __ballot(any_open) == 0ull) go = false;
    }
    lds_sync();
    PROF(11);
    if (!go) { if constexpr (SC) scored_selected(*scp, env, 0); return; }
    PROF_COUNT(52, 1000);
    int n_act = 0;
    while (true) {
      // round_tasks: open tasks (that had residual > 0 initially) with residual > 1e-9, in last_tasks_info order
      int nc;
      if constexpr (SC) nc = compact_to(X.roundT, n_sc, [&](int i) { return X.resid[S.open_slot[sc_list[i]]] > 1e-9; }, [&](int i) { return (int)sc_list[i]; });
      else nc = compact_to(X.roundT, n_plan, [&](int k) { return X.resid[S.open_slot[k]] > 1e-9; }, [&](int k) { return k; });
      lds_sync();
      if (nr == 0 || nc == 0) break;
      const bool tr = nc < nr;              // scipy transposes so that rows <= cols
      const int Rr = tr ? nc : nr, Cc = tr ? nr : nc;
      if (mode == 2) {  // _threat_stats (AttentionEscort.py:46-66): nearest live threat to the task (escort: to its recon)
        for (int j = lane; j < nc; j += WG) {
          const int s = S.open_slot[X.roundT[j]];
          const int pa = S.t_prot_agent[s];
          const double ax = pa >= 0 ? S.a_px[pa] : S.t_px[s], ay = pa >= 0 ? S.a_py[pa] : S.t_py[s];
          double best = MAX_COORD;
          for (int k = 0; k < S.n_active_threats; k++) {
            const int h = S.h_order[k];
            if (S.h_status[h] == 2) continue;
            const double d = norm2(S.h_px[h] - ax, S.h_py[h] - ay);
            if (d < best) best = d;
          }
          (TL::OTFC ? X.press : X.spc)[j] = 1.0 - fmin(div_coord(best), 1.0);  // threat pressure of round task j (a cost-tile LDS solve may clobber spc afterwards; the register solver leaves it)
        }
        lds_sync();
      }
      // ---- cost (:137-179) of agent a for the task in slot s (jr: its index among the round's tasks) ----
      auto pair_cost = [&](int a, int s, int jr) -> double {
        double c = 1e6;
        bool ok = !(vis && !((S.known[a][s >> 5] >> (s & 31)) & 1u));
        if (ok && (S.t_flags[s] & TF_ELIGIBLE) && !((S.t_elig[s] >> S.a_type[a]) & 1u)) ok = false;
        if (ok) {
          double urgency = 0.0;
          if (S.t_flags[s] & TF_DEADLINE) {
            int remaining = S.t_deadline[s] - tnow;
            remaining = remaining > 0 ? remaining : 0;
            urgency = 1.0 - fmin(div40((double)remaining), 1.0);
          }
          double delivered = is_escort_task(s) ? 1.0 : S.a_caps[S.t_type[s]][a];
          if (delivered > 0) {  // _cost (:43-70), left-to-right, priority = 0
            double dist = norm2(S.a_px[a] - S.t_px[s], S.a_py[a] - S.t_py[s]);
            double missing = fmax(X.resid[s], 1e-6);
            double pri = 0.0;
            if constexpr (SC) { const int col = pair_info()[s]; if (scp->pri && col < scp->MT) pri = scp->pri[(size_t)env * scp->MT + col]; }
            double base = div_coord(dist) - 0.5 * fmin(delivered, missing) - 0.4 * pri - 0.6 * urgency;
            double score = 0.0;
            if constexpr (SC) {  // edge_score_dict (PairCostHybrid.py:283-294 / AttentionEscort.py:472-482)
              const int col = pair_info()[s], row = X.live_rank[a];
              bool edge = col < scp->MT && row < scp->MA && scp->scores != nullptr;
              if (edge && (scp->flags & MUAVTA_SC_EDGE_VALID_ONLY)) {
                if (envvis && !((S.known[a][s >> 5] >> (s & 31)) & 1u)) edge = false;
                if (scp->kind != 2 && !(S.a_caps[S.t_type[s]][a] > 0)) edge = false;  // (eligibility already holds here)
              }
              if (edge) score = (double)scp->scores[((size_t)env * scp->MA + row) * scp->MT + col];
            }
            if (mode == 1) {  // urgency_edge_scores (PairCostHybrid.py:68-86) on the edges build_pair_tokens keeps (:42-60)
              const int info = pair_info()[s];
              if ((info & 255) < 32 && X.live_rank[a] < 16 && S.a_caps[S.t_type[s]][a] > 0) {
                double scar = 0.0;
                if (vis) scar = 1.0 - fmin((double)(info >> 8) / (double)n_live, 1.0);
                double v = 0.5 * urgency + 0.3 * scar - 0.4 * div_coord(dist);
                v = fmin(fmax(v, -0.35), 0.35);
                score = (double)(float)v;  // the scores array is float32
              }
            }
            if (mode == 2) {  // UrgencyCoalition.plan (AttentionEscort.py:729-752), left-to-right in f64
              const int ty = S.t_type[s];
              const bool esc = (S.t_flags[s] & TF_ESCORT) != 0;
              const double cap = S.a_caps[ty][a] > 0 ? S.a_caps[ty][a] : 0.0;
              double v = 0.45 * urgency + 0.35 * (TL::OTFC ? X.press : X.spc)[jr] * (0.5 + 0.5 * (esc ? 1.0 : 0.0)) + 0.3 * fmin(cap, 1.0) - 0.25 * div_coord(dist);
              const bool fighter = is_fighter(S.a_type[a]);
              if (fighter && (esc || ty == MUAVTA_INT)) v += 0.2;
              if (!fighter && ty == MUAVTA_REC) v += 0.2;
              score = fmin(fmax(v, 0.0), 1.0);
            }
            if (base < 1e5 / 2) c = base - score;
          }
        }
        return c;
      };
      static_assert(KW <= 4, "an agent's known mask is cached in four words");
      // pair_cost() split in two for the solvers that keep one column per lane: the column's side (task or agent fields) is
      // read from LDS ONCE per lane, the row's side is the same for every lane (broadcast reads, issued together).  Same arithmetic.
      struct TS { double px, py, urgency, missing, press, pri; int type, info; bool elig_on, esc_task, esc_flag; uint32_t elig; };
      // (the known mask as two 64-bit halves: a four-way select of 32-bit words by a per-lane index is turned into an indexed load
      // from a stack copy of the struct by the optimiser — scratch stores and a dependent scratch load per cost element on the
      // 64-agent tile; a two-way select and a 64-bit shift stay in registers)
      struct AS { double px, py; int type, rank; uint32_t k0, k1; unsigned long long k23; };
      auto load_ts = [&](int sl, int jr) {
        TS t;
        const int fl = S.t_flags[sl];
        t.px = S.t_px[sl]; t.py = S.t_py[sl]; t.type = S.t_type[sl]; t.elig = S.t_elig[sl];
        t.elig_on = (fl & TF_ELIGIBLE) != 0; t.esc_flag = (fl & TF_ESCORT) != 0;
        t.esc_task = t.esc_flag || S.t_required[sl] > 0;
        t.urgency = 0.0;
        if (fl & TF_DEADLINE) {
          int remaining = S.t_deadline[sl] - tnow;
          remaining = remaining > 0 ? remaining : 0;
          t.urgency = 1.0 - fmin(div40((double)remaining), 1.0);
        }
        t.missing = fmax(X.resid[sl], 1e-6);
        t.info = (SC || mode == 1) ? pair_info()[sl] : 0;
        t.press = mode == 2 ? (TL::OTFC ? X.press : X.spc)[jr] : 0.0;
        t.pri = 0.0;
        if constexpr (SC) { if (scp->pri && t.info < scp->MT) t.pri = scp->pri[(size_t)env * scp->MT + t.info]; }
        return t;
      };
      auto load_as = [&](int a) {
        AS g;
        g.px = S.a_px[a]; g.py = S.a_py[a]; g.type = S.a_type[a];
        g.k0 = S.known[a][0]; g.k1 = KW > 1 ? S.known[a][KW > 1 ? 1 : 0] : 0u;
        g.k23 = (KW > 2 ? (unsigned long long)S.known[a][KW > 2 ? 2 : 0] : 0ull) | ((KW > 3 ? (unsigned long long)S.known[a][KW > 3 ? 3 : 0] : 0ull) << 32);
        g.rank = (SC || mode == 1) ? (int)X.live_rank[a] : 0;
        return g;
      };
      // == pair_cost(a, sl, jr); capv = S.a_caps[t.type][a].  STRAIGHT-LINE per lane: every lane evaluates the whole expression and the
      // conditions only select the result.  In SIMT the arithmetic was executed anyway whenever one lane needed it; what the nested
      // `if`s added was an exec-mask region each (s_and_saveexec + branch + s_or: five per cost element), and on the lone wave that
      // ends a launch of configs 4 and 5 every taken branch is an instruction-buffer refill.  Same operations on the same operands
      // for every lane whose result is kept.
      auto pair_eval_cap = [&](const AS& g, int sl, const TS& t, double capv) -> double {
        bool known;
        if constexpr (KW <= 2) known = ((((sl >> 5) ? g.k1 : g.k0) >> (sl & 31)) & 1u) != 0;
        else { const unsigned long long w = (sl & 64) ? g.k23 : ((unsigned long long)g.k0 | ((unsigned long long)g.k1 << 32)); known = ((w >> (sl & 63)) & 1ull) != 0; }
        bool ok = !(vis & !known);
        ok = ok & !(t.elig_on & !((t.elig >> g.type) & 1u));
        const double delivered = t.esc_task ? 1.0 : capv;
        ok = ok & (delivered > 0);
        const double dist = norm2(g.px - t.px, g.py - t.py);
        const double dc = div_coord(dist);
        double base;
        if constexpr (SC) base = dc - 0.5 * fmin(delivered, t.missing) - 0.4 * t.pri - 0.6 * t.urgency;
        else base = dc - 0.5 * fmin(delivered, t.missing) - 0.4 * 0.0 - 0.6 * t.urgency;
        double score = 0.0;
        if constexpr (SC) {  // edge_score_dict: the caller's f32 score of (token row, token column), float()ed
          bool edge = (t.info < scp->MT) & (g.rank < scp->MA) & (scp->scores != nullptr);
          if (scp->flags & MUAVTA_SC_EDGE_VALID_ONLY) edge = edge & (!envvis | known) & !(t.elig_on & !((t.elig >> g.type) & 1u)) & ((scp->kind == 2) | (capv > 0));
          float sv = 0.f;
          if (edge) sv = scp->scores[((size_t)env * scp->MA + g.rank) * scp->MT + t.info];
          score = (double)sv;
        }
        if (mode == 1) {  // (uniform)
          double scar = 0.0;
          if (vis) scar = 1.0 - fmin((double)(t.info >> 8) / (double)n_live, 1.0);
          double v = 0.5 * t.urgency + 0.3 * scar - 0.4 * dc;
          v = fmin(fmax(v, -0.35), 0.35);
          const bool edge = ((t.info & 255) < 32) & (g.rank < 16) & (capv > 0);
          score = edge ? (double)(float)v : 0.0;
        }
        if (mode == 2) {  // (uniform)
          const double cap = capv > 0 ? capv : 0.0;
          double v = 0.45 * t.urgency + 0.35 * t.press * (0.5 + 0.5 * (t.esc_flag ? 1.0 : 0.0)) + 0.3 * fmin(cap, 1.0) - 0.25 * dc;
          const bool fighter = is_fighter(g.type);
          const double v_f = v + 0.2;
          v = (fighter & (t.esc_flag | (t.type == MUAVTA_INT))) ? v_f : v;
          const double v_r = v + 0.2;
          v = (!fighter & (t.type == MUAVTA_REC)) ? v_r : v;
          score = fmin(fmax(v, 0.0), 1.0);
        }
        ok = ok & (base < 1e5 / 2);
        const double c = base - score;
        return ok ? c : 1e6;
      };
      auto pair_eval = [&](int a, const AS& g, int sl, const TS& t) -> double { return pair_eval_cap(g, sl, t, S.a_caps[t.type][a]); };
      bool feasible = false;
      CostCol col;
      // REGC: one LSAP column per lane with its costs in registers, whenever the columns fit the wave (always on the 16- and
      // 24-agent tiles; up to 64 columns on the 64-agent tile, whose kernels are built for 256 VGPRs: four 16-row tuples)
      const bool reg_cols = TL::REGC && (T <= WG || Cc <= WG);
      if (reg_cols) {
        // lane = LSAP column (a task, or an agent when scipy transposes): its whole cost column goes straight into
        // uniformly indexed registers, one row per iteration — no A x T tile in LDS.  pair_cost() split in two: the column's
        // side (task or agent fields) is read from LDS ONCE per lane, the row's side is the same for every lane (broadcast
        // reads, issued together), so an iteration costs one LDS round trip instead of a chain of six.  Same arithmetic.
        const bool incol = lane < Cc;
        {
          // What a row costs the wave that ends a launch is its dependent chain, not its instruction count (§6 of DESIGN.md).  Lane r
          // fetches ROW r's operands up front (every row at once, one LDS round trip), and the row loop broadcasts them with
          // v_readlane: no LDS access on the loop's chain.  Same arithmetic as pair_cost().  (r3: first on the 64-agent tile only;
          // the other two tiles read the row's fields inside the loop — two dependent LDS round trips per row — until the end of r3.)
          auto pack = [](const TS& t) { return t.type | (t.elig_on ? 8 : 0) | (t.esc_task ? 16 : 0) | (t.esc_flag ? 32 : 0) | (int)(t.elig << 8); };
          if (!tr) {  // rows = free agents, lane = task column
            const int my_s = incol ? (int)S.open_slot[X.roundT[lane]] : (int)S.open_slot[X.roundT[0]];
            const TS ts = load_ts(my_s, incol ? lane : 0);
            const int ra = X.freeA[lane < Rr ? lane : 0];
            const AS gr = load_as(ra);
            const int rpk = gr.type | (gr.rank << 8);
            build_cols(col, Rr, [&](int i) -> double {
              AS g;
              const int a = __builtin_amdgcn_readlane(ra, i), pk = __builtin_amdgcn_readlane(rpk, i);
              g.px = readlane_f64(gr.px, i); g.py = readlane_f64(gr.py, i); g.type = pk & 255; g.rank = pk >> 8;
              g.k0 = __builtin_amdgcn_readlane(gr.k0, i); g.k1 = __builtin_amdgcn_readlane(gr.k1, i);
              g.k23 = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)gr.k23, i) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(gr.k23 >> 32), i) << 32);
              double c = 0.0;
              c = pair_eval_cap(g, my_s, ts, S.a_caps[ts.type][a]); feasible |= incol & (c < 1e5 / 2);  // (lanes beyond the columns evaluate column 0's operands; nobody reads their result)
              return c;
            });
          } else {  // rows = round tasks, lane = free-agent column
            const int my_a = incol ? X.freeA[lane] : X.freeA[0];
            const AS g = load_as(my_a);
            typedef double caps_vec __attribute__((ext_vector_type(8)));
            caps_vec mycaps;
#pragma unroll
            for (int c6 = 0; c6 < 6; c6++) mycaps[c6] = S.a_caps[c6][my_a];
            const int rj = lane < Rr ? lane : 0;
            const int rsl = S.open_slot[X.roundT[rj]];
            const TS tr_ = load_ts(rsl, rj);
            const int rpk = pack(tr_);
            build_cols(col, Rr, [&](int i) -> double {
              TS t;
              const int sl = __builtin_amdgcn_readlane(rsl, i), pk = __builtin_amdgcn_readlane(rpk, i);
              t.px = readlane_f64(tr_.px, i); t.py = readlane_f64(tr_.py, i); t.urgency = readlane_f64(tr_.urgency, i);
              t.missing = readlane_f64(tr_.missing, i); t.press = mode == 2 ? readlane_f64(tr_.press, i) : 0.0;
              t.info = (SC || mode == 1) ? __builtin_amdgcn_readlane(tr_.info, i) : 0;
              t.pri = SC ? readlane_f64(tr_.pri, i) : 0.0;
              t.type = pk & 7; t.elig_on = (pk & 8) != 0; t.esc_task = (pk & 16) != 0; t.esc_flag = (pk & 32) != 0; t.elig = (uint32_t)pk >> 8;
              double c = 0.0;
              c = pair_eval_cap(g, sl, t, mycaps[t.type]); feasible |= incol & (c < 1e5 / 2);
              return c;
            });
          }
        }
      } else {
        for (int p = lane; p < nr * nc; p += WG) {  // one (agent, task) pair per lane
          const int i = p / nc, j = p - i * nc;
          const double c = pair_cost(X.freeA[i], S.open_slot[X.roundT[j]], j);
          if constexpr (!TL::OTFC) X.cost[tr ? (j * Cc + i) : (i * Cc + j)] = c;  // OTFC: this pass only answers "any feasible pair?"
          feasible |= c < 1e5 / 2;
        }
      }
      // no pair under the acceptance threshold -> the round accepts nothing whatever the assignment is
      PROF_COUNT(62, 1000 * Rr);
      if (__ballot(feasible) == 0ull) { PROF_COUNT(63, 1000 * Rr); break; }
      lds_sync();
      PROF(12);
      if (reg_cols) lsap_reg_solve(Rr, Cc, [&](int i) -> double { return col.get(i); });
      else if constexpr (TL::OTFC)  // beyond 64 columns: element (row i, column j) of scipy's (possibly transposed) matrix, evaluated when scanned
        lsap(Rr, Cc, [&](int i, int j) { return tr ? pair_cost(X.freeA[j], S.open_slot[X.roundT[i]], i) : pair_cost(X.freeA[i], S.open_slot[X.roundT[j]], j); });
      else if constexpr (T <= WG && A <= 32) lsap_reg(Rr, Cc);
      else lsap(Rr, Cc);
      PROF(13);
      // accept (:182-204): one free agent per lane, actions appended in ascending agent order (scipy returns
      // rows sorted); each task appears at most once per round, so the residual updates are independent
      int n_acc = 0, n_left = 0, n_staged = 0;
      for (int base = 0; base < nr; base += WG) {
        const int i = base + lane;
        bool acc = false, keep = false;
        int a = -1, oi = 0, s = 0;
        if (i < nr) {
          a = X.freeA[i];
          const int j = tr ? X.row4col[i] : X.col4row[i];
          keep = true;
          if (j >= 0) {
            double cij;
            if constexpr (TL::NO_COST_TILE) {  // same value the solver saw, through the same straight-line evaluator
              const int sl = S.open_slot[X.roundT[j]];
              cij = pair_eval_cap(load_as(a), sl, load_ts(sl, j), S.a_caps[S.t_type[sl]][a]);
            }
            else cij = X.cost[tr ? (j * Cc + i) : (i * Cc + j)];
            if (cij < 1e5 / 2) {
              acc = true; keep = false;
              oi = X.roundT[j]; s = S.open_slot[oi];
            }
          }
        }
        // (a task beyond last_tasks_info — created by an out-of-step call since the last observation — takes its agent out of the round and
        // has its residual reduced like any other, but the pair is not an action: _apply_assign only keeps tasks of env.last_tasks_info)
        const bool stage = acc && oi < S.n_open;
        const unsigned long long am = __ballot(acc), km = __ballot(keep), sm = __ballot(stage);
        const unsigned long long below = (1ull << lane) - 1ull;
        lds_sync();  // freeA fully read before it is compacted in place
        if (acc) {
          if (stage) {
            const int n = n_act + n_staged + prefix_count(sm);
            S.act_agent[n] = a; S.act_slot[n] = s; S.act_index[n] = oi;
          }
          const double delivered = is_escort_task(s) ? 1.0 : S.a_caps[S.t_type[s]][a];
          X.resid[s] = fmax(X.resid[s] - delivered, 0.0);
        }
        if (keep) X.freeA[n_left + prefix_count(km)] = a;
        n_acc += __popcll(am);
        n_staged += __popcll(sm);
        n_left += __popcll(km);
      }
      n_act += n_staged;
      nr = n_acc ? n_left : 0;  // no accept -> stop
      lds_sync();
      PROF(14);
    }
    if ((mode == 2 || (SC && (scp->flags & MUAVTA_SC_COMMIT))) && P.commit_horizon > 0 && lane < n_act) {  // apply_agent_commits (AttentionCommit.py:33-44)
      const int a = S.act_agent[lane];
      if (S.a_qlen[a] > 0) S.a_commit[a] = tnow + P.commit_horizon;  // only agents that hold a real task now
    }
    if (lane == 0) {
      S.n_act = n_act;
      S.last_plan_step = tnow;
      S.n_replans++;
    }
    lds_sync();
    if constexpr (SC) scored_selected(*scp, env, n_act);
  }
  // _selected_mask(tok, result) (PairCostHybrid.py:296-310): 1 where the plan pairs token row i with token column j
  DEV void scored_selected(const ScoredDev& sc, int env, int n_act) {
    if (!sc.selected) return;
    float* o = sc.selected + (size_t)env * sc.MA * sc.MT;
    for (int idx = lane; idx < sc.MA * sc.MT; idx += WG) {
      const int i = idx / sc.MT, j = idx - i * sc.MT;
      bool hit = false;
      for (int k = 0; k < n_act; k++) hit |= (int)X.live_rank[S.act_agent[k]] == i && (int)pair_info()[S.act_slot[k]] == j;
      o[idx] = hit ? 1.f : 0.f;
    }
  }

  // ====================================================================================================
  // Token builders: build_pair_tokens = build_att_tokens + edge_valid
  // (TaskAllocation/Hybrid/AttentionRAH.py:50-173, PairCostHybrid.py:31-65; kind 0, kind 1 = raw) and
  // build_escort_tokens (AttentionEscort.py:76-243; kind 2).  One token row per lane; every value is formed in
  // f64 in the reference's order and rounded to f32 once, like `np.float32` array assignment.
  // Rows: task_feats [max_tasks, Dt], agent_feats [max_agents, Da], edge_valid [max_agents, max_tasks]; masks 1 = pad.
  // ====================================================================================================
  struct TokPtrs {
    float* task_feats; uint8_t* task_mask; int32_t* task_ids; float* agent_feats; uint8_t* agent_mask; int32_t* agent_ids;
    float* edge_valid; int32_t* n_urgent;
    float* expert_mask;   // optional [N, max_agents, max_tasks]: _expert_mask of the pairs staged by the last allocate
    int32_t* replanned;   // optional [N]: the allocator planned at this time step
    int kind, max_tasks, max_agents;
  };
  DEV double slot_urgency(int s) const {  // _urgency (AttentionRAH.py:29-34)
    if (!(S.t_flags[s] & TF_DEADLINE)) return 0.0;
    int remaining = S.t_deadline[s] - tnow;
    remaining = remaining > 0 ? remaining : 0;
    return 1.0 - fmin(div40((double)remaining), 1.0);
  }
  DEV void threat_stats(int s, double& pressure, double& dist_n, double& fighter_pressure) const {  // AttentionEscort.py:46-66
    const int pa = S.t_prot_agent[s];
    const double ax = pa >= 0 ? S.a_px[pa] : S.t_px[s], ay = pa >= 0 ? S.a_py[pa] : S.t_py[s];
    double best = MAX_COORD;
    int n_near = 0;
    for (int k = 0; k < S.n_active_threats; k++) {
      const int h = S.h_order[k];
      if (S.h_status[h] == 2) continue;
      const double d = norm2(S.h_px[h] - ax, S.h_py[h] - ay);
      if (d < best) best = d;
      n_near += d < 150.0;
    }
    pressure = 1.0 - fmin(div_coord(best), 1.0);
    dist_n = fmin(div_coord(best), 1.0);
    fighter_pressure = fmin((double)n_near / 4.0, 1.0);
  }
  // build_escort_tokens' task list (AttentionEscort.py:83-96): of the n_all slots in X.roundT (_open_tasks_residual, env.tasks order)
  // those known by at least one live agent (X.freeA[0..n_live_raw); all of them if that leaves nothing or visibility is off),
  // stably sorted by _task_priority_key (:69-74).  Result in X.path, returns its length.  Scratch: X.remaining, X.spc.
  DEV int escort_sorted_list(bool vis, int n_live_raw, int n_all) {
    int n_list = n_all;
    if (vis) {  // local task set: known by at least one live agent; all of open_all if that leaves nothing
      const int n_loc = compact_to(X.remaining, n_all,
                                   [&](int k) {
                                     const int s = X.roundT[k];
                                     bool any = false;
                                     for (int i = 0; i < n_live_raw; i++) any |= (S.known[X.freeA[i]][s >> 5] >> (s & 31)) & 1u;
                                     return any;
                                   },
                                   [&](int k) { return X.roundT[k]; });
      lds_sync();
      if (n_loc > 0) n_list = n_loc;
      else for (int k = lane; k < n_all; k += WG) X.remaining[k] = X.roundT[k];
    } else {
      for (int k = lane; k < n_all; k += WG) X.remaining[k] = X.roundT[k];
    }
    lds_sync();
    // stable sort by _task_priority_key (:69-74): rank = number of entries that sort before this one
    for (int k = lane; k < n_list; k += WG) {
      const int s = X.remaining[k];
      double pr, dn, fp;
      threat_stats(s, pr, dn, fp);
      X.spc[k] = -(1.5 * slot_urgency(s) + 1.2 * pr + 0.8 * ((S.t_flags[s] & TF_ESCORT) ? 1.0 : 0.0) + 0.5 * (S.t_type[s] == MUAVTA_INT ? 1.0 : 0.0));
    }
    lds_sync();
    for (int k = lane; k < n_list; k += WG) {
      const double key = X.spc[k];
      int rank = 0;
      for (int j = 0; j < n_list; j++) rank += (X.spc[j] < key) || (X.spc[j] == key && j < k);
      X.path[rank] = X.remaining[k];
    }
    lds_sync();
    return n_list;
  }
  DEV void tokens(const TokPtrs& K, int env) {
    const int kind = K.kind, MT = K.max_tasks, MA = K.max_agents;
    const int Dt = kind == 0 ? 13 : kind == 1 ? 9 : 22, Da = kind == 0 ? 12 : kind == 1 ? 11 : 16;
    float* o_tf = K.task_feats + (size_t)env * MT * Dt;
    uint8_t* o_tm = K.task_mask + (size_t)env * MT;
    int32_t* o_tid = K.task_ids + (size_t)env * MT;
    float* o_af = K.agent_feats + (size_t)env * MA * Da;
    uint8_t* o_am = K.agent_mask + (size_t)env * MA;
    int32_t* o_aid = K.agent_ids + (size_t)env * MA;
    float* o_ev = K.edge_valid + (size_t)env * MA * MT;
    const bool vis = !(P.sense_radius == 0 && P.threat_delay == 0);  // agent_visibility_map() is not None
    const double horizon = (double)(P.max_time_steps > 1 ? P.max_time_steps : 1);
    const double mid_x = AREA_W * 0.5;
    const double URGENT = 1.0 - 12.0 / 40.0;
    // live agents (get_live_agents order) and the F2 specialists among them
    const int n_live_raw = compact_to(X.freeA, P.n_agents, [&](int a) { return S.a_state[a] != -1; }, [&](int a) { return a; });
    const int n_spec = compact_to(X.col4row, P.n_agents, [&](int a) { return S.a_state[a] != -1 && S.a_type[a] == MUAVTA_F2; }, [&](int a) { return a; });
    const int n_live = n_live_raw > 1 ? n_live_raw : 1;
    // open_all (env.tasks order): underfilled tasks (kind 0/1) or _open_tasks_residual (kind 2)
    int n_all = compact_to(X.roundT, S.n_order,
                           [&](int k) {
                             const int s = S.t_order[k];
                             if (S.t_status[s] == 2) return false;
                             if (kind == 2) return residual_demand(s) > 0;
                             const int ty = S.t_type[s];
                             return C.t_alloc[ty][s] < C.t_cur[ty][s];
                           },
                           [&](int k) { return S.t_order[k]; });
    lds_sync();
    // per open task: urgent-and-windowed flag (for the agents' n_known_urgent), known-by count
    for (int k = lane; k < n_all; k += WG) {
      const int s = X.roundT[k];
      X.SC[k] = (slot_urgency(s) >= URGENT && (S.t_flags[s] & TF_DEADLINE)) ? 1 : 0;
    }
    lds_sync();
    int16_t* list = X.roundT;  // the token rows' task list
    int n_list = n_all;
    if (kind == 2) {
      n_list = escort_sorted_list(vis, n_live_raw, n_all);
      list = X.path;
    }
    const int n_kept = n_list < MT ? n_list : MT;
    // ---- task rows ----
    int n_urgent = 0;
    for (int base = 0; base < MT; base += WG) {
      const int i = base + lane;
      bool urgent = false;
      if (i < MT) {
        float* f = o_tf + (size_t)i * Dt;
        if (i < n_kept) {
          const int s = list[i];
          const int ty = S.t_type[s];
          const double urg = slot_urgency(s);
          int n_know_i = 0;
          for (int b = 0; b < P.n_agents; b++) n_know_i += (S.known[b][s >> 5] >> (s & 31)) & 1u;
          const double scar = vis ? 1.0 - fmin((double)n_know_i / (double)n_live, 1.0) : 0.0;
          const bool dyn = (S.t_flags[s] & TF_DEADLINE) != 0;
          double d_spec = MAX_COORD;
          for (int q = 0; q < n_spec; q++) {
            const int a = X.col4row[q];
            const double d = norm2(S.a_px[a] - S.t_px[s], S.a_py[a] - S.t_py[s]);
            if (q == 0 || d < d_spec) d_spec = d;
          }
          int c = 0;
          f[c++] = (float)(S.t_px[s] / MAX_COORD); f[c++] = (float)(S.t_py[s] / MAX_COORD); f[c++] = (float)((double)ty / 8.0);
          f[c++] = ty == MUAVTA_ATT ? 1.f : 0.f; f[c++] = ty == MUAVTA_REC ? 1.f : 0.f; f[c++] = ty == MUAVTA_INT ? 1.f : 0.f;
          if (kind != 2) {
            const double rem = fmax(C.t_cur[ty][s] - C.t_alloc[ty][s], 0.0);
            urgent = urg >= URGENT && dyn;
            if (kind == 1) {
              int left = S.t_deadline[s] - tnow;
              left = left > 0 ? left : 0;
              f[c++] = (float)(dyn ? fmin((double)left / horizon, 1.0) : 1.0);
              f[c++] = (float)fmin(rem / 4.0, 1.0); f[c++] = dyn ? 1.f : 0.f;
            } else {
              const double n_know = vis ? (double)n_know_i : 1.0;  // _known_by_count
              f[c++] = (float)urg; f[c++] = (float)scar; f[c++] = (float)fmin(rem / 4.0, 1.0); f[c++] = dyn ? 1.f : 0.f;
              f[c++] = (float)fmin(n_know / (double)n_live, 1.0); f[c++] = (float)fmin(d_spec / MAX_COORD, 1.0);
              f[c++] = S.t_px[s] < mid_x ? 0.f : 1.f;
            }
          } else {
            double rem, req_agents = 1.0;
            if (is_escort_task(s)) {
              req_agents = S.t_required[s] ? (double)S.t_required[s] : 1.0;
              rem = fmax(req_agents - (double)S.t_ndet[s], 0.0);
            } else rem = fmax(C.t_cur[ty][s] - C.t_alloc[ty][s], 0.0);
            const double n_know = vis ? (double)n_know_i : 0.0;
            const double deficit = fmin(rem / 4.0, 1.0);
            double pr, dn, fp;
            threat_stats(s, pr, dn, fp);
            const int pa = S.t_prot_agent[s];
            const double prot_x = (pa >= 0 ? S.a_px[pa] : S.t_px[s]) / MAX_COORD, prot_y = (pa >= 0 ? S.a_py[pa] : S.t_py[s]) / MAX_COORD;
            const float prot_alive = (pa >= 0 && S.a_state[pa] != -1) ? 1.f : 0.f;
            f[c++] = (float)urg; f[c++] = (float)scar; f[c++] = (float)deficit; f[c++] = dyn ? 1.f : 0.f;
            f[c++] = (float)fmin(n_know / (double)n_live, 1.0); f[c++] = (float)fmin(d_spec / MAX_COORD, 1.0);
            f[c++] = S.t_px[s] < mid_x ? 0.f : 1.f; f[c++] = (S.t_flags[s] & TF_ESCORT) ? 1.f : 0.f; f[c++] = (float)deficit; f[c++] = (float)pr;
            f[c++] = (float)prot_x; f[c++] = (float)prot_y; f[c++] = (float)fmin(req_agents / 4.0, 1.0); f[c++] = (float)dn;
            f[c++] = prot_alive; f[c++] = (float)fp;
          }
          o_tm[i] = 0; o_tid[i] = S.t_id[s];
        } else {
          for (int c = 0; c < Dt; c++) f[c] = 0.f;
          o_tm[i] = 1; o_tid[i] = -1;
        }
      }
      n_urgent += __popcll(__ballot(urgent));
    }
    if (lane == 0 && K.n_urgent) K.n_urgent[env] = n_urgent;
    if (lane == 0 && K.replanned) K.replanned[env] = S.gate_step == tnow + 1;
    float* o_em = K.expert_mask ? K.expert_mask + (size_t)env * MA * MT : nullptr;
    // ---- agent rows + edge_valid ----
    for (int i = lane; i < MA; i += WG) {
      float* f = o_af + (size_t)i * Da;
      float* ev = o_ev + (size_t)i * MT;
      if (i < n_live_raw) {
        const int a = X.freeA[i];
        int n_known_urgent = 0;
        for (int k = 0; k < n_all; k++) {
          const int s = X.roundT[k];
          if (X.SC[k] && (!vis || ((S.known[a][s >> 5] >> (s & 31)) & 1u))) n_known_urgent++;
        }
        const int ty = S.a_type[a];
        const bool fighter = is_fighter(ty);
        int c = 0;
        f[c++] = (float)(S.a_px[a] / MAX_COORD); f[c++] = (float)(S.a_py[a] / MAX_COORD); f[c++] = fighter ? 1.f : 0.f; f[c++] = fighter ? 0.f : 1.f;
        f[c++] = S.a_qlen[a] == 0 ? 1.f : 0.f;
        f[c++] = (float)fmin(S.a_caps[2][a] / 2.0, 1.0); f[c++] = (float)fmin(S.a_caps[3][a] / 2.0, 1.0); f[c++] = (float)fmin(S.a_caps[1][a] / 2.0, 1.0);
        f[c++] = (float)((double)S.a_state[a] / 5.0); f[c++] = (float)((double)tnow / horizon);
        if (kind == 0) f[c++] = (float)fmin((double)n_known_urgent / (double)(n_all > 1 ? n_all : 1), 1.0);
        if (kind == 2) f[c++] = (float)fmin((double)n_known_urgent / 8.0, 1.0);
        f[c++] = ty == MUAVTA_F2 ? 1.f : 0.f;
        if (kind == 2) {
          double is_escorting = 0.0, dist_prot = 1.0, near_escort = 0.0;
          if (S.a_qlen[a] > 0) {
            const int hs = S.a_qslot[a][0];
            if (ref_valid(S.a_qid[a][0], hs) && (S.t_flags[hs] & TF_ESCORT)) {
              is_escorting = 1.0;
              const int pa = S.t_prot_agent[hs];
              if (pa >= 0) {
                dist_prot = fmin(norm2(S.a_px[a] - S.a_px[pa], S.a_py[a] - S.a_py[pa]) / MAX_COORD, 1.0);
                near_escort = 1.0 - dist_prot;
              }
            }
          }
          int n_known_tasks = 0;  // len(known_ids): ids of released tasks are counted in a_gone
          if (vis) { for (int w = 0; w < KW; w++) n_known_tasks += __popc(S.known[a][w]); n_known_tasks += S.a_gone[a]; }
          const double c_h = (double)(P.commit_horizon ? (P.commit_horizon > 1 ? P.commit_horizon : 1) : 20);
          const double rem_commit = fmax((double)S.a_commit[a] - (double)tnow, 0.0);
          f[c++] = (float)is_escorting; f[c++] = (float)dist_prot; f[c++] = (float)fmin(rem_commit / c_h, 1.0);
          f[c++] = (float)fmin(near_escort + (double)n_known_tasks / 16.0, 1.0);
        }
        o_am[i] = 0; o_aid[i] = a;
        int es = -1;  // the slot the staged plan gives this agent (train_pair_cost.py:54-71: never through the visibility mask)
        if (o_em) for (int k = 0; k < S.n_act; k++) if (S.act_agent[k] == a) es = S.act_slot[k];
        for (int j = 0; j < MT; j++) {
          float v = 0.f;
          int s = -2;
          if (j < n_kept) {
            s = list[j];
            bool ok = !vis || ((S.known[a][s >> 5] >> (s & 31)) & 1u);
            if (ok && (S.t_flags[s] & TF_ELIGIBLE) && !((S.t_elig[s] >> ty) & 1u)) ok = false;
            if (ok && kind != 2 && !(S.a_caps[S.t_type[s]][a] > 0)) ok = false;
            v = ok ? 1.f : 0.f;
          }
          ev[j] = v;
          if (o_em) o_em[(size_t)i * MT + j] = (s == es && v > 0.5f) ? 1.f : 0.f;
        }
      } else {
        for (int c = 0; c < Da; c++) f[c] = 0.f;
        for (int j = 0; j < MT; j++) { ev[j] = 0.f; if (o_em) o_em[(size_t)i * MT + j] = 0.f; }
        o_am[i] = 1; o_aid[i] = -1;
      }
    }
  }
  // Token-ring slot of an env whose episode has ended (muavta_rollout_record): an all-pad sample that is never a training
  // row — masks 1, ids -1, features / edge_valid / expert_mask 0, n_urgent 0, replanned 0.  The reference's episode loops
  // stop at `done` (train_pair_cost.py:108,139), so there is nothing to restate; the slots just must not stay uninitialised.
  DEV void tokens_pad(const TokPtrs& K, int env) {
    const int kind = K.kind, MT = K.max_tasks, MA = K.max_agents;
    const int Dt = kind == 0 ? 13 : kind == 1 ? 9 : 22, Da = kind == 0 ? 12 : kind == 1 ? 11 : 16;
    float* o_tf = K.task_feats + (size_t)env * MT * Dt;
    float* o_af = K.agent_feats + (size_t)env * MA * Da;
    float* o_ev = K.edge_valid + (size_t)env * MA * MT;
    for (int i = lane; i < MT * Dt; i += WG) o_tf[i] = 0.f;
    for (int i = lane; i < MA * Da; i += WG) o_af[i] = 0.f;
    for (int i = lane; i < MA * MT; i += WG) o_ev[i] = 0.f;
    if (K.expert_mask) { float* o_em = K.expert_mask + (size_t)env * MA * MT; for (int i = lane; i < MA * MT; i += WG) o_em[i] = 0.f; }
    for (int i = lane; i < MT; i += WG) { K.task_mask[(size_t)env * MT + i] = 1; K.task_ids[(size_t)env * MT + i] = -1; }
    for (int i = lane; i < MA; i += WG) { K.agent_mask[(size_t)env * MA + i] = 1; K.agent_ids[(size_t)env * MA + i] = -1; }
    if (lane == 0 && K.n_urgent) K.n_urgent[env] = 0;
    if (lane == 0 && K.replanned) K.replanned[env] = 0;
  }
  // number of set bits of a (uniform) lane mask below this lane: v_mbcnt_lo / v_mbcnt_hi, two VALU instructions
  // (`prefix_count(m)` is a 64-bit shift, a 64-bit subtract, two ANDs and two bit counts)
  static DEV int prefix_count(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
  }
  // out[0..n) = f(k) for the k in [0, count) with pred(k), order preserved (ballot + popcount); returns n
  template <class Out, class Pred, class Val>
  DEV int compact_to(Out* out, int count, Pred pred, Val val) {
    int n = 0;
    for (int base = 0; base < count; base += WG) {
      const int k = base + lane;
      const bool p = k < count && pred(k);
      const unsigned long long m = __ballot(p);
      if (p) out[n + prefix_count(m)] = (Out)val(k);
      n += __popcll(m);
    }
    return n;
  }

  // scipy.optimize.linear_sum_assignment (rectangular_lsap, scipy 1.15.3) on X.cost[nr x nc], nr <= nc.
  // Wave-cooperative: lane `it` owns scan position `it` of scipy's `remaining` array (filled in
  // reverse, swap-removed), so the sequential tie rule — a column replaces the running minimum when
  // strictly lower, or when equal and still unassigned — becomes: take the LAST unassigned position
  // among the minima if there is one, else the FIRST minimum.  One f64 wave-min + two ballots per
  // augmenting step; duals are updated one row/column per lane.  Arithmetic order per column is
  // scipy's: minVal + C[i][j] - u[i] - v[j].  All lanes must call this (uniform control flow).
  // f64 min over the wave without LDS traffic: 4 DPP exchange steps inside each 16-lane row (min is
  // idempotent, so mirrors are as good as butterflies), then the 4 row results via v_readlane.
  DEV double dpp_xchg(double v, const int ctrl_sel) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl_sel == 0) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xf, 0xf, true); }        // quad_perm [1,0,3,2]
    else if (ctrl_sel == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x4E, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x4E, 0xf, 0xf, true); }   // quad_perm [2,3,0,1]
    else if (ctrl_sel == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x141, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x141, 0xf, 0xf, true); } // row_half_mirror
    else { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x140, 0xf, 0xf, true); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x140, 0xf, 0xf, true); }                    // row_mirror
    return __hiloint2double(hi, lo);
  }
  DEV double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  }
  // v_min_f64 as such.  fmin() compiles to llvm.minnum, which has to quiet a signalling NaN: every operand the compiler cannot
  // prove canonical — anything that came through a DPP move or a v_readlane — gets a v_max_f64 x, x, x in front, which doubled
  // the VALU count of the wave-wide minimum.  The operands here are finite costs or +inf, never NaN.
  static DEV double vmin(double a, double b) {
    double d;
    asm("v_min_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
  }
  // row_bcast:15 into rows 1 and 3 / row_bcast:31 into rows 2 and 3 (GFX9 DPP); the rows a step does not write keep `keep`
  DEV double dpp_bcast(double v, double keep, const int which) {
    int lo, hi;
    if (which == 15) {
      lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), 0x142, 0xa, 0xf, false);
      hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), 0x142, 0xa, 0xf, false);
    } else {
      lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), 0x143, 0xc, 0xf, false);
      hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), 0x143, 0xc, 0xf, false);
    }
    return __hiloint2double(hi, lo);
  }
  // wave-wide minimum (uniform result): four exchange steps inside each 16-lane row, then the row broadcasts carry the row
  // minima up to lane 63 — 18 VALU + 2 v_readlane (was: 12 + 8 v_readlane + 3 v_min + 9 canonicalising v_max).  In the rows a
  // broadcast step does not write the partner operand stays the previous step's (a value of the lane's own row, >= its minimum).
  DEV double wave_min(double v) {
    double t = dpp_xchg(v, 0); v = vmin(v, t);
    t = dpp_xchg(v, 1); v = vmin(v, t);
    t = dpp_xchg(v, 2); v = vmin(v, t);
    t = dpp_xchg(v, 3); v = vmin(v, t);
    t = dpp_bcast(v, t, 15); v = vmin(v, t);
    t = dpp_bcast(v, t, 31); v = vmin(v, t);
    return readlane_f64(v, 63);
  }
  // The same when only the first `n` lanes can hold anything but +inf (n wave-uniform): the minimum of a 16-lane row is complete
  // after the four exchange steps, so up to 16 lanes need no row broadcast (12 VALU) and up to 32 only the first (15 VALU).
  DEV double wave_min_first(double v, int n) {
    double t = dpp_xchg(v, 0); v = vmin(v, t);
    t = dpp_xchg(v, 1); v = vmin(v, t);
    t = dpp_xchg(v, 2); v = vmin(v, t);
    t = dpp_xchg(v, 3); v = vmin(v, t);
    if (n <= 16) return readlane_f64(v, 0);
    t = dpp_bcast(v, t, 15); v = vmin(v, t);
    if (n <= 32) return readlane_f64(v, 31);
    t = dpp_bcast(v, t, 31); v = vmin(v, t);
    return readlane_f64(v, 63);
  }
  DEV void lsap(int nr, int nc) { const double* c = X.cost; lsap(nr, nc, [c, nc](int i, int j) { return c[i * nc + j]; }); }
  template <class CostAt>
  DEV void lsap(int nr, int nc, CostAt cost_at) {
    const double INF = __builtin_huge_val();
    for (int i = lane; i < nr; i += WG) { X.u[i] = 0; X.col4row[i] = -1; }
    for (int j = lane; j < nc; j += WG) { X.v[j] = 0; X.row4col[j] = -1; X.path[j] = -1; }
    lds_sync();
    for (int cur = 0; cur < nr; cur++) {
      {
        // A row whose first scan step already finds an unassigned column at the minimum needs none of the search state (same
        // reasoning as in lsap_reg_solve: positions are still the initial ones, so the unassigned minimum at the LAST position is
        // the one with the LOWEST column index; u[cur] += minVal, v unchanged, one path edge).
        const double ui = X.u[cur];
        double gmin = INF;
        int g_un = -1;
        for (int base = 0; base < nc; base += WG) {
          const int j = base + lane;
          double val = INF;
          bool un = false;
          if (j < nc) { val = 0.0 + cost_at(cur, j) - ui - X.v[j]; un = X.row4col[j] == -1; }
          const double m = wave_min(val);
          const unsigned long long equ = __ballot(j < nc && val == m && un);
          if (m < gmin) { gmin = m; g_un = equ ? base + __ffsll((long long)equ) - 1 : -1; }
          else if (m == gmin && g_un < 0 && equ) g_un = base + __ffsll((long long)equ) - 1;
        }
        if (gmin != INF && g_un >= 0) {
          lds_sync();  // every lane has read u / v / row4col
          if (lane == 0) { X.u[cur] = ui + gmin; X.row4col[g_un] = cur; X.col4row[cur] = g_un; }
          lds_sync();
          continue;
        }
      }
      for (int it = lane; it < nc; it += WG) { X.remaining[it] = nc - it - 1; X.SC[it] = 0; X.spc[it] = INF; }
      for (int r = lane; r < nr; r += WG) X.SR[r] = 0;
      lds_sync();
      double minVal = 0;
      int i = cur, num_remaining = nc, sink = -1;
      while (sink == -1) {
        const double ui = X.u[i];
        double gmin = INF;
        int g_first = -1, g_lastU = -1;
        for (int base = 0; base < num_remaining; base += WG) {
          const int it = base + lane;
          const bool active = it < num_remaining;
          double val = INF;
          bool un = false;
          if (active) {
            const int j = X.remaining[it];
            const double r = minVal + cost_at(i, j) - ui - X.v[j];
            double sp = X.spc[j];
            if (r < sp) { X.path[j] = i; X.spc[j] = r; sp = r; }
            val = sp;
            un = X.row4col[j] == -1;
          }
          const double m = wave_min(val);
          const unsigned long long eq = __ballot(active && val == m);
          const unsigned long long equ = __ballot(active && val == m && un);
          if (m < gmin) {
            gmin = m;
            g_first = base + __ffsll((long long)eq) - 1;
            g_lastU = equ ? base + 63 - __clzll((long long)equ) : -1;
          } else if (m == gmin && equ) {
            g_lastU = base + 63 - __clzll((long long)equ);
          }
        }
        if (gmin == INF) { if (lane == 0) fail(MUAVTA_ERR_LSAP); lds_sync(); return; }
        minVal = gmin;
        const int index = g_lastU >= 0 ? g_lastU : g_first;
        const int j = X.remaining[index];
        const int rj = X.row4col[j];
        const int last = X.remaining[num_remaining - 1];
        lds_sync();  // everyone has read before lane 0 rewrites `remaining`
        if (lane == 0) { X.SR[i] = 1; X.SC[j] = 1; X.remaining[index] = last; }
        num_remaining--;
        if (rj == -1) sink = j; else i = rj;
        lds_sync();
      }
      // dual updates (one row / column per lane), then the augmentation along `path`
      for (int r = lane; r < nr; r += WG) {
        if (r == cur) X.u[r] += minVal;
        else if (X.SR[r]) X.u[r] += minVal - X.spc[X.col4row[r]];
      }
      for (int j = lane; j < nc; j += WG) if (X.SC[j]) X.v[j] -= minVal - X.spc[j];
      lds_sync();
      if (lane == 0) {
        int j = sink;
        while (true) {
          int r = X.path[j];
          X.row4col[j] = r;
          int t = X.col4row[r]; X.col4row[r] = j; j = t;
          if (r == cur) break;
        }
      }
      lds_sync();
    }
  }

  // Same algorithm with the search state in REGISTERS (nc <= 64): lane `it` carries the column sitting at
  // scan position `it` — its id, dual v, shortest-path cost, predecessor row and assignment — and a
  // swap-remove moves the last position's registers into the vacated lane with v_readlane.  Row duals u and
  // col4row live in the row's lane.  Per scan step only the cost element is read from LDS.
  // Register-resident variant for A <= 32, T <= 64: lane j owns COLUMN j for the whole solve — its cost column
  // (one uniform-indexed register per row: s_set_gpr_idx), v[j], row4col[j], the shortest-path cost and predecessor —
  // and lane r owns row r's u[r] / col4row[r]; no LDS traffic inside the solve.  scipy's `remaining` array is kept
  // only as each column's POSITION in it (filled in reverse, swap-removed), which is all its tie rule looks at:
  // among the minima take the unassigned column at the LAST position if there is one, else the FIRST position.
  // A lane's LSAP cost column in registers: 16-row tuples indexed with a uniform row number (s_set_gpr_idx) — one tuple up
  // to 16 agents, a second of 8 (<= 24 agents) or 16 rows, four on the 64-agent tile (128 VGPRs of its 256).
  typedef double cost_vec __attribute__((ext_vector_type(16)));
  typedef double cost_vec8 __attribute__((ext_vector_type(8)));
  struct CostCol {
    cost_vec v0;
    typename std::conditional<(A > 24), cost_vec, cost_vec8>::type v1;
    cost_vec v2, v3;
    DEV double get(int i) const {  // i uniform
      if constexpr (A <= 16) return v0[i];
      else if constexpr (A <= 32) { if (i < 16) return v0[i]; return v1[i - 16]; }
      else { if (i < 16) return v0[i]; if (i < 32) return v1[i - 16]; if (i < 48) return v2[i - 32]; return v3[i - 48]; }
    }
    DEV void set(int i, double c) {
      if constexpr (A <= 16) v0[i] = c;
      else if constexpr (A <= 32) { if (i < 16) v0[i] = c; else v1[i - 16] = c; }
      else { if (i < 16) v0[i] = c; else if (i < 32) v1[i - 16] = c; else if (i < 48) v2[i - 32] = c; else v3[i - 48] = c; }
    }
  };
  // col[i] = cost(i) for the rows i < n of a column, one register tuple per loop: a loop that picks the tuple by a branch on i
  // makes the register allocator copy a whole tuple around every indexed write (8-16 v_mov_b64 per row on the larger tiles)
  template <class RowCost>
  DEV void build_cols(CostCol& col, int n_, RowCost cost) {
    const int n = __builtin_amdgcn_readfirstlane(n_);
    for (int i = 0; i < (n < 16 ? n : 16); i++) col.v0[i] = cost(i);
    if constexpr (A > 16) for (int i = 16; i < (n < 32 ? n : 32); i++) col.v1[i - 16] = cost(i);
    if constexpr (A > 32) {
      for (int i = 32; i < (n < 48 ? n : 48); i++) col.v2[i - 32] = cost(i);
      for (int i = 48; i < n; i++) col.v3[i - 48] = cost(i);
    }
  }
  DEV void lsap_reg(int nr, int nc) {  // cost tile staged in X.cost (R x C row-major), nr <= nc <= 64
    CostCol col;
    build_cols(col, nr, [&](int i) -> double { return lane < nc ? X.cost[i * nc + lane] : 0.0; });
    lsap_reg_solve(nr, nc, [&](int i) -> double { return col.get(i); });
  }
  // scipy.optimize.linear_sum_assignment (rectangular_lsap.cpp) with the search state in REGISTERS, nr <= nc <= 64: lane j owns
  // COLUMN j for the whole solve — cost_row(i) is its cost in (uniform) row i, v[j], row4col[j], the shortest-path cost and
  // predecessor — and lane r owns row r's u[r] / col4row[r]; no LDS traffic inside the solve.  scipy's `remaining` array is kept
  // only as each column's POSITION in it (filled in reverse, swap-removed), which is all its tie rule looks at: among the
  // minima take the unassigned column at the LAST position if there is one, else the FIRST position.
  // Per scan step (r3): the set of unassigned columns is a uniform bit mask kept on the scalar side (it only changes when a
  // path is augmented), the ballots are taken straight from compares (a ballot of a combined predicate costs a v_cndmask +
  // v_cmp to materialise it), the wave-wide minimum is 18 VALU (wave_min): ~40 VALU per step, from ~75.
  template <class CostRow>
  DEV void lsap_reg_solve(int nr_, int nc_, CostRow cost_row) {
    // the problem size is uniform, but it was counted in loops whose trip count came out of LDS, which makes it divergent in the
    // compiler's eyes — and with it the column masks, the selected column and the row being scanned (VALU selects instead of
    // scalar code, a 16-way select chain instead of an indexed register read)
    const int nr = __builtin_amdgcn_readfirstlane(nr_), nc = __builtin_amdgcn_readfirstlane(nc_);
    const double INF = __builtin_huge_val();
    double u_r = 0, vj = 0;  // lane r < nr: u[r];  lane j < nc: v[j]
    int c4r = -1, r4c = -1;  // lane r: col4row[r];  lane j: row4col[j]
    const bool incol = lane < nc;
    unsigned long long unassigned = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);  // uniform: columns j with row4col[j] == -1
    PROF_COUNT(48, 1000); PROF_COUNT(50, 1000 * nr); PROF_COUNT(51, 1000 * nc);
#ifdef MUAVTA_PROF
    int prof_iters = 0;
#endif
    bool bad = false;  // (uniform) an infeasible matrix
    for (int cur = 0; cur < nr; cur++) {
      {
        // A row whose FIRST scan step already finds an unassigned column at the minimum (most rows: nc > nr leaves most columns
        // free) needs none of the search state below.  Every column is still in `remaining` at its initial position nc-1-j, so
        // "the unassigned minimum at the LAST position" is the one in the LOWEST lane; only row cur was scanned (no other u
        // changes), the selected column's v changes by minVal - spc = 0, and the path is the single edge (cur, sink).  Same values
        // as the general loop (0.0 + c is kept: it turns a cost of -0.0 into +0.0 there as well).  ~45 instructions and 3 branches
        // instead of ~160 / 20 — on a lone wave (the env that ends a launch of configs 4 and 5) a scan step costs ~1,700 cycles.
#ifdef MUAVTA_PROF
        prof_iters++;
#endif
        const double r0 = 0.0 + cost_row(cur) - readlane_f64(u_r, cur) - vj;
        const double val0 = incol ? r0 : INF;
        const double m0 = wave_min(val0);
        if (__double2hiint(m0) != 0x7ff00000) {
          const unsigned long long equ0 = __builtin_amdgcn_ballot_w64(val0 == m0) & unassigned;
          if (equ0) {
            const int sink0 = __ffsll((long long)equ0) - 1;
            if (lane == cur) { u_r += m0; c4r = sink0; }
            if (lane == sink0) r4c = cur;
            unassigned &= ~(1ull << sink0);
            continue;
          }
        }
#ifdef MUAVTA_PROF
        prof_iters--;  // (the general search below counts this scan step again)
#endif
      }
      int pos = nc - 1 - lane;  // scipy fills `remaining` in reverse: column j sits at position nc-1-j
      double sp = INF;
      int pth = -1;
      bool active = incol;
      unsigned long long SRmask = 0ull;
      double minVal = 0;
      int i = cur, nrem = nc, sink;
      // The loop body is the step that CONTINUES the search (the minimum sits in an assigned column); the step that ends it — an
      // unassigned column at the minimum — leaves the loop first and is finished behind it.  One exit, four branches per scan step.
      unsigned long long equ;
      double m;
      for (;;) {
#ifdef MUAVTA_PROF
        prof_iters++;
#endif
        SRmask |= 1ull << i;
        const double ui = readlane_f64(u_r, i);
        const double r = minVal + cost_row(i) - ui - vj;
        const bool upd = active && r < sp;
        sp = upd ? r : sp;
        pth = upd ? i : pth;
        const double val = active ? sp : INF;
        m = wave_min(val);  // (wave_min_first(val, nc) saves 3-6 VALU per scan step for <= 32 columns and loses more to its two scalar branches in the dependent chain: measured r3)
        // m == +inf (never NaN; a scalar compare of the high word): every remaining column is out of reach — the matrix is infeasible.
        // No exit here: the unassigned columns then all sit at the "minimum", so the search ends in this step by itself; the flag
        // is raised behind the solve (a `return` inside the loop made the compiler thread an exit code through every scan step).
        bad |= __double2hiint(m) == 0x7ff00000;
        const unsigned long long eq = __builtin_amdgcn_ballot_w64(val == m);  // scanned columns and lanes beyond nc hold +inf > m
        equ = eq & unassigned;
        if (equ) break;  // an unassigned column ends the search
        int sel = __ffsll((long long)eq) - 1;
        if (eq & (eq - 1ull)) {    // several (assigned) minima: the one at the first position
          int best = 1 << 30;
          for (unsigned long long t = eq; t; t &= t - 1ull) {
            const int b = __ffsll((long long)t) - 1;
            const int pb = __builtin_amdgcn_readlane(pos, b);
            if (pb < best) { best = pb; sel = b; }
          }
        }
        minVal = m;
        const int psel = __builtin_amdgcn_readlane(pos, sel);
        if (pos == nrem - 1) pos = psel;  // remaining[index] = remaining[--num_remaining]
        if (lane == sel) active = false;
        nrem--;
        i = __builtin_amdgcn_readlane(r4c, sel);
      }
      if (bad) break;  // (infeasible: no path to augment — the columns at the "minimum" have no predecessor row)
      {  // the last scan step: among the unassigned minima the one at the last position (positions and `nrem` are not needed any more)
        int sel = 63 - __clzll((long long)equ);
        if (equ & (equ - 1ull)) {
          int best = -1;
          for (unsigned long long t = equ; t; t &= t - 1ull) {
            const int b = __ffsll((long long)t) - 1;
            const int pb = __builtin_amdgcn_readlane(pos, b);
            if (pb > best) { best = pb; sel = b; }
          }
        }
        minVal = m;
        if (lane == sel) active = false;
        sink = sel;
      }
      // dual updates (u over the scanned rows, v over the scanned columns = the ones that left `remaining`)
      const unsigned long long others = SRmask & ~(1ull << cur);
      if (others) {  // (uniform) most searches end in their first row: no other row was scanned, nothing to gather
        const double spc_of_my_col = __shfl(sp, c4r < 0 ? 0 : c4r);  // rows in SR other than cur are assigned
        if ((others >> lane) & 1ull) u_r += minVal - spc_of_my_col;
      }
      if (lane == cur) u_r += minVal;
      if (incol && !active) vj -= minVal - sp;
      // augmentation along `path` (uniform walk)
      unassigned &= ~(1ull << sink);
      int jj = sink;
      while (true) {
        const int r = __builtin_amdgcn_readlane(pth, jj);
        if (lane == jj) r4c = r;
        const int t = __builtin_amdgcn_readlane(c4r, r);
        if (lane == r) c4r = jj;
        jj = t;
        if (r == cur) break;
      }
    }
    if (bad && lane == 0) fail(MUAVTA_ERR_LSAP);  // (scipy: ValueError("cost matrix is infeasible"); the assignment written below is meaningless then)
    if (lane < nr) X.col4row[lane] = c4r;
    if (incol) X.row4col[lane] = r4c;
    PROF_COUNT(49, 1000 * prof_iters);
    lds_sync();
  }

  // compute_s_wps (:1321-1337), same operation order as metrics() below
  DEV double s_wps() const {
    const double dist_term = 0.01 * S.total_distance / fmax(MAX_COORD, 1.0);
    const double rematch = P.reassign_penalty * (double)S.n_task_switches;
    return 12.0 * (double)S.n_on_time - 30.0 * (double)S.n_missed_windows - dist_term - rematch;
  }
  // calculate_metrics (:1231-1319) -> out[30]
  DEV void metrics(double* m) {
    if (lane != 0) return;
    double F_quality = S.next_task_id > 1 ? 0.0 : __builtin_nan("");
    double F_Time = 1.0 / (double)S.conclusion_time * (double)P.max_time_steps;
    double F_distance = S.total_distance > 0 ? 1 / S.total_distance * MAX_COORD : 0;
    int Losses = 0, Kills = 0;
    for (int a = 0; a < P.n_agents; a++) Losses += (S.a_state[a] == -1);
    for (int k = 0; k < S.n_active_threats; k++) Kills += (S.h_status[S.h_order[k]] == 2);
    double dist_term = 0.01 * S.total_distance / fmax(MAX_COORD, 1.0);
    double rematch = P.reassign_penalty * (double)S.n_task_switches;
    double s_wps = 12.0 * (double)S.n_on_time - 30.0 * (double)S.n_missed_windows - dist_term - rematch;
    int req = S.escort_required_steps > 1 ? S.escort_required_steps : 1;
    double escort_cov = (double)S.escort_covered_steps / (double)req;
    double s_esc = s_wps + 20.0 * (double)S.protected_rec_completed - 30.0 * (double)S.recon_losses + 20.0 * escort_cov;
    int k = 0;
    m[k++] = F_Time; m[k++] = F_distance; m[k++] = F_quality; m[k++] = S.F_Reward; m[k++] = s_wps; m[k++] = s_esc;
    m[k++] = Losses; m[k++] = Kills; m[k++] = S.conclusion_time; m[k++] = S.total_distance; m[k++] = S.n_reallocations;
    m[k++] = S.n_task_switches; m[k++] = S.n_arrivals; m[k++] = S.next_task_id - 1; m[k++] = S.n_reached;
    m[k++] = S.n_missed_windows; m[k++] = S.n_on_time; m[k++] = S.n_windowed_tasks;
    int den = S.n_on_time + S.n_missed_windows; den = den > 1 ? den : 1;
    m[k++] = (double)S.n_on_time / (double)den;
    int den2 = tnow * (P.n_agents > 1 ? P.n_agents : 1); den2 = den2 > 1 ? den2 : 1;
    m[k++] = (double)S.idle_reserve_steps / (double)den2;
    m[k++] = escort_cov; m[k++] = S.protected_rec_completed; m[k++] = S.recon_losses; m[k++] = S.escort_losses;
    m[k++] = S.threats_intercepted; m[k++] = S.mutual_support_engagements; m[k++] = S.protection_breaches;
    m[k++] = S.escort_requests; m[k++] = S.escort_completed; m[k++] = S.escort_failed;
  }
};


}  // namespace muavta
