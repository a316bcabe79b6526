// doorbell_probe.hip — builder's probe (not product): what a host <-> resident-kernel handshake costs on this box, i.e. the per-step
// overhead floor of a persistent "mailbox" stepping kernel.  One resident single-wave workgroup per env slot:
//   host:   writes step number t to a host-mapped doorbell                                   (PCIe write, posted)
//   relay:  workgroup 0 polls the doorbell over PCIe and republishes t in a device-memory flag (so that 4096 waves poll L2, not PCIe)
//   waves:  poll the device flag (s_sleep between polls), [optionally burn `work` iterations], bump a device counter;
//           the last one of the step writes t to a host-mapped ack                             (PCIe write)
//   host:   spins on the ack.
// Every loop on both sides is bounded by wall-clock time (s_memrealtime / steady_clock): a lost doorbell ends the run, it cannot hang it.
//   hipcc --offload-arch=gfx950 -O2 -o doorbell_probe doorbell_probe.hip && ./doorbell_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// variant 1: the completion count goes through per-group counters (64 workgroups each, one cache line apart) — the last wave of a group
// bumps the top counter — and the step flag is replicated per group (the relay writes every copy), so no address sees more than 64 clients
template <int SLEEP, int MODE = 0>
__global__ __launch_bounds__(64) void k_serve_probe_tree(volatile uint32_t* doorbell, volatile uint32_t* ack, uint32_t* flags /*[groups][16]*/,
                                                         uint32_t* counters /*[groups + 1][16]*/, uint32_t* gave_up, int n_steps, int work,
                                                         unsigned long long budget_ticks, double* sink) {
  const int wg = blockIdx.x, n_wg = gridDim.x, grp = wg >> 6, n_grp = (n_wg + 63) >> 6;
  const int grp_size = (grp == n_grp - 1) ? n_wg - (grp << 6) : 64;
  double acc = threadIdx.x;
  for (int t = 1; t <= n_steps; t++) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    if (wg == 0) {
      while (true) {
        uint32_t v = 0;
        if (threadIdx.x == 0) v = __hip_atomic_load((uint32_t*)doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= (uint32_t)t) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget_ticks) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (ok) for (int g = threadIdx.x; g < n_grp; g += 64) __hip_atomic_store(flags + g * 16, (uint32_t)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (true) {
        uint32_t v = 0;
        if (threadIdx.x == 0) v = MODE == 0 ? __hip_atomic_load(flags + grp * 16, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                            : __hip_atomic_load(flags + grp * 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= (uint32_t)t) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget_ticks) { ok = false; break; }
        __builtin_amdgcn_s_sleep(SLEEP);
      }
    }
    if (!ok) { if (threadIdx.x == 0) atomicAdd(gave_up, 1u); break; }
    for (int i = 0; i < work; i++) acc = acc * 1.0000001 + 0.5;
    if (threadIdx.x == 0) {
      if (MODE != 1) __threadfence();
      const uint32_t done = (MODE == 0 ? __hip_atomic_fetch_add(counters + (1 + grp) * 16, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                       : __hip_atomic_fetch_add(counters + (1 + grp) * 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u;
      if (done == (uint32_t)grp_size * (uint32_t)t) {
        const uint32_t top = (MODE == 0 ? __hip_atomic_fetch_add(counters, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT)
                                        : __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) + 1u;
        if (top == (uint32_t)n_grp * (uint32_t)t) {
          __threadfence_system();
          __hip_atomic_store((uint32_t*)ack, (uint32_t)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
  }
  if (acc == 12345.678) sink[0] = acc;
}

__global__ __launch_bounds__(64) void k_serve_probe(volatile uint32_t* doorbell /*host*/, volatile uint32_t* ack /*host*/, uint32_t* flag /*device*/,
                                                    uint32_t* counter /*device*/, uint32_t* gave_up /*device*/, int n_steps, int work, unsigned long long budget_ticks,
                                                    double* sink) {
  const int wg = blockIdx.x, n_wg = gridDim.x;
  double acc = threadIdx.x;
  for (int t = 1; t <= n_steps; t++) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    // (one lane polls, the wave follows its verdict: 64 lanes polling one address are 64 requests)
    if (wg == 0) {  // relay: PCIe poll -> device flag
      while (true) {
        uint32_t v = 0;
        if (threadIdx.x == 0) v = __hip_atomic_load((uint32_t*)doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= (uint32_t)t) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget_ticks) { ok = false; break; }
        __builtin_amdgcn_s_sleep(2);
      }
      if (ok && threadIdx.x == 0) __hip_atomic_store(flag, (uint32_t)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (true) {
        uint32_t v = 0;
        if (threadIdx.x == 0) v = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        v = __builtin_amdgcn_readfirstlane(v);
        if (v >= (uint32_t)t) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > budget_ticks) { ok = false; break; }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (!ok) { if (threadIdx.x == 0) atomicAdd(gave_up, 1u); break; }
    for (int i = 0; i < work; i++) acc = acc * 1.0000001 + 0.5;  // stand-in for a step's dependent chain
    if (threadIdx.x == 0) {
      __threadfence();
      const uint32_t done = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u;
      if (done == (uint32_t)n_wg * (uint32_t)t) {  // last wave of step t
        __threadfence_system();
        __hip_atomic_store((uint32_t*)ack, (uint32_t)t, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  if (acc == 12345.678) sink[0] = acc;
}

int main(int argc, char** argv) {
  int n_steps = argc > 1 ? atoi(argv[1]) : 300;
  CK(hipSetDevice(0));
  uint32_t *h_db, *h_ack, *d_db, *d_ack, *d_flag, *d_counter, *d_gave;
  double* d_sink;
  CK(hipHostMalloc((void**)&h_db, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostMalloc((void**)&h_ack, 64, hipHostMallocMapped | hipHostMallocCoherent));
  CK(hipHostGetDevicePointer((void**)&d_db, h_db, 0));
  CK(hipHostGetDevicePointer((void**)&d_ack, h_ack, 0));
  CK(hipMalloc(&d_flag, 64)); CK(hipMalloc(&d_counter, 64)); CK(hipMalloc(&d_gave, 64)); CK(hipMalloc(&d_sink, 64));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  const unsigned long long budget = 100000000ull / 2;  // 0.5 s of the 100 MHz constant clock per wait
  printf("workgroups  work  steps  round trip us: median  p10  p90  max   (kernel gave up: n)\n");
  uint32_t *d_flags, *d_counters;
  CK(hipMalloc(&d_flags, 65 * 64)); CK(hipMalloc(&d_counters, 66 * 64));
  for (int variant = 0; variant < 4; variant++)
  for (int n_wg : {1, 256, 1024, 4096}) {
    for (int work : {0, 2000}) {
      if (variant && n_wg == 1) continue;
      *h_db = 0; *h_ack = 0;
      CK(hipMemsetAsync(d_flag, 0, 64, st)); CK(hipMemsetAsync(d_counter, 0, 64, st)); CK(hipMemsetAsync(d_gave, 0, 64, st));
      CK(hipMemsetAsync(d_flags, 0, 65 * 64, st)); CK(hipMemsetAsync(d_counters, 0, 66 * 64, st));
      CK(hipStreamSynchronize(st));
      if (variant == 0) hipLaunchKernelGGL(k_serve_probe, dim3(n_wg), dim3(64), 0, st, d_db, d_ack, d_flag, d_counter, d_gave, n_steps, work, budget, d_sink);
      else if (variant == 1) hipLaunchKernelGGL(k_serve_probe_tree<8>, dim3(n_wg), dim3(64), 0, st, d_db, d_ack, d_flags, d_counters, d_gave, n_steps, work, budget, d_sink);
      else if (variant == 2) hipLaunchKernelGGL((k_serve_probe_tree<32, 1>), dim3(n_wg), dim3(64), 0, st, d_db, d_ack, d_flags, d_counters, d_gave, n_steps, work, budget, d_sink);
      else hipLaunchKernelGGL((k_serve_probe_tree<32, 2>), dim3(n_wg), dim3(64), 0, st, d_db, d_ack, d_flags, d_counters, d_gave, n_steps, work, budget, d_sink);
      if (n_wg == 256 && work == 0) printf("-- variant %d (%s)\n", variant, variant == 0 ? "one flag, one counter, acquire polls, fence + acq_rel count, s_sleep 8" : variant == 1 ? "per-group flags and counters (64 clients per address), same orderings, s_sleep 8" : variant == 2 ? "per-group, RELAXED polls and counts, no fence, s_sleep 32" : "per-group, relaxed polls and counts + one __threadfence() per wave and step (release of its stores), s_sleep 32");
      CK(hipGetLastError());
      std::vector<double> us;
      bool lost = false;
      for (int t = 1; t <= n_steps && !lost; t++) {
        auto t0 = std::chrono::steady_clock::now();
        __atomic_store_n(h_db, (uint32_t)t, __ATOMIC_RELEASE);
        while (__atomic_load_n(h_ack, __ATOMIC_ACQUIRE) < (uint32_t)t) {
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 1.0) { lost = true; break; }
        }
        us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
      }
      if (lost) __atomic_store_n(h_db, (uint32_t)(n_steps + 1), __ATOMIC_RELEASE);  // let the kernel run to its end
      CK(hipStreamSynchronize(st));
      uint32_t gave = 0;
      CK(hipMemcpy(&gave, d_gave, 4, hipMemcpyDeviceToHost));
      std::vector<double> s(us.begin() + std::min<size_t>(us.size(), 20), us.end());  // skip the first steps (kernel start-up)
      if (s.empty()) s = us;
      std::sort(s.begin(), s.end());
      printf("%10d %5d %6zu  %22.2f %5.2f %5.2f %6.2f   (%u)%s\n", n_wg, work, us.size(), s[s.size() / 2], s[s.size() / 10], s[s.size() * 9 / 10], s.back(), gave,
             lost ? "  HOST TIMED OUT" : "");
      fflush(stdout);
    }
  }
  return 0;
}
