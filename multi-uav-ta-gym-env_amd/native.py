"""Loader for the HIP shared library (csrc/ -> libmuavta.so) behind the C ABI of include/muavta.h.

The library is the product: there is no Python or CPU fallback.  If it is missing, or no MI355X is
visible, the calls fail loudly (``MuavtaError``).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import shutil
import subprocess

from .params import MuavtaDims, MuavtaParams

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
SO_PATH = os.environ.get("MUAVTA_SO") or os.path.join(PKG_DIR, "libmuavta.so")  # MUAVTA_SO: diagnostic builds only
# -O3 / -O2 / -Os are within 2 % of each other on the 128-VGPR kernel (-Oz: -12 %).  MachineLICM is switched off: in the one
# big loop of k_rollout it hoists scalar loads of launch constants and address arithmetic out of the step body, the values then
# live across the whole step at 104 SGPRs / 128 VGPRs and come back as v_readlane / scratch reloads — VALU issue, which is
# what bounds the kernel (headline 171 -> 188 M env-steps/s; profiles/r02_codegen_flags.txt).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ldl", "-mllvm", "-disable-machine-licm",
                "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-invalid-offsetof"]

EXPORTS = [
    "muavta_create", "muavta_destroy", "muavta_last_error", "muavta_dims", "muavta_reset", "muavta_step",
    "muavta_allocate", "muavta_step_staged", "muavta_rollout", "muavta_observe", "muavta_step_result",
    "muavta_metrics", "muavta_get", "muavta_set", "muavta_get_state", "muavta_set_state", "muavta_lsap",
    "muavta_avoid_obstacles", "muavta_device_ptrs", "muavta_last_kernel_ms", "muavta_sync",
    "muavta_rollout_metrics", "muavta_refresh_observation", "muavta_get_rng", "muavta_set_rng", "muavta_abi_sizes",
    "muavta_set_allocator", "muavta_tokens", "muavta_tokens_device", "muavta_set_release_log", "muavta_lsap_impl",
    "muavta_last_seed_ms", "muavta_call", "muavta_rollout_record", "muavta_comm_uid", "muavta_comm_init", "muavta_allreduce_metrics", "muavta_comm_destroy",
    "muavta_kernel_ms_history", "muavta_wait_stream", "muavta_set_parts", "muavta_part_range", "muavta_rollout_part", "muavta_allocate_part",
    "muavta_step_part", "muavta_observe_part", "muavta_wait_part", "muavta_domain_math", "muavta_domain_log", "muavta_domain_atan2", "muavta_step_lists",
    "muavta_allocate_scored", "muavta_allocate_scored_device", "muavta_rl_step_device", "muavta_launch_gaps_ms",
    "muavta_rl_run_device", "muavta_step_run", "muavta_set_lanes", "muavta_lanes", "muavta_rollout_metrics_back", "muavta_error_flags_back", "muavta_set_slot_cap", "muavta_context", "muavta_context_device",
]


class MuavtaError(RuntimeError):
    pass


class MuavtaScored(C.Structure):
    """include/muavta.h: MuavtaScored (muavta_allocate_scored)."""
    _fields_ = [("kind", C.c_int32), ("max_tasks", C.c_int32), ("max_agents", C.c_int32), ("gate", C.c_int32), ("flags", C.c_int32),
                ("replan_interval", C.c_int32), ("use_visibility", C.c_int32), ("reserved0", C.c_int32),
                ("edge_scores", C.c_void_p), ("task_pri", C.c_void_p), ("reserved", C.c_void_p), ("selected", C.c_void_p),
                ("replanned", C.c_void_p)]


def sources():
    """every file the shipped library is compiled from (muavta_diag.h is not one of them: diagnostic builds only)"""
    sim = sorted(os.path.join(CSRC, "sim", f) for f in os.listdir(os.path.join(CSRC, "sim")) if f.endswith(".inc"))
    return [os.path.join(CSRC, f) for f in ("muavta_kernels.hip", "muavta_device.h", "muavta_state.h", "muavta_math.h", "muavta_atan2.h", "muavta_atan2_tab.inc", "muavta_rng.h")] + sim + [
        os.path.join(os.path.dirname(PKG_DIR), "include", "muavta.h")]


def source_hash() -> str:
    """Hash of the kernel sources: PMC figures under profiles/ are only reused for the build they were taken on (bench.py)."""
    import hashlib

    h = hashlib.sha256()
    for p in sources():
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def needs_build() -> bool:
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    return any(os.path.getmtime(s) > t for s in sources())


def build(force: bool = False, verbose: bool = False) -> str:
    """Cross-compile the kernels for gfx950 with hipcc (works without a GPU)."""
    if not force and not needs_build():
        return SO_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc] + HIPCC_FLAGS + ["-o", SO_PATH, os.path.join(CSRC, "muavta_kernels.hip")]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


class MuavtaRlStep(C.Structure):
    """include/muavta.h: MuavtaRlStep (muavta_rl_step_device)."""
    _fields_ = [("plan", MuavtaScored)] + [(n, C.c_void_p) for n in ("task_feats", "task_mask", "task_ids", "agent_feats", "agent_mask", "agent_ids",
                                                                      "edge_valid", "n_urgent", "s_wps", "done")] + [("write_obs", C.c_int32), ("part", C.c_int32)]


class MuavtaRlRun(C.Structure):
    """include/muavta.h: MuavtaRlRun (muavta_rl_run_device)."""
    _fields_ = [("first", MuavtaRlStep)] + [(n, C.c_void_p) for n in ("park_task_feats", "park_task_mask", "park_task_ids", "park_agent_feats", "park_agent_mask",
                                                                       "park_agent_ids", "park_edge_valid", "park_n_urgent", "n_stepped", "park", "reward_sum")] + [
        ("max_steps", C.c_int32), ("reserved", C.c_int32)]


_LIB = None


def lib() -> C.CDLL:
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(SO_PATH):
        raise MuavtaError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; if this library pulls in /opt/rocm's copy
    # first, a later `import torch` in the same process finds no GPU.  Let torch's copy win when torch is installed.
    if "torch" not in sys.modules and os.environ.get("MUAVTA_NO_TORCH") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(SO_PATH)
    vp, i32, u64p = C.c_void_p, C.c_int32, C.c_void_p
    L.muavta_create.argtypes = [C.POINTER(MuavtaParams), i32, i32, C.POINTER(vp)]
    L.muavta_destroy.argtypes = [vp]
    L.muavta_last_error.argtypes = [vp]
    L.muavta_last_error.restype = C.c_char_p
    L.muavta_dims.argtypes = [vp, C.POINTER(MuavtaDims)]
    L.muavta_reset.argtypes = [vp, u64p]
    L.muavta_step.argtypes = [vp, vp, vp]
    L.muavta_step_lists.argtypes = [vp, vp, vp, i32]
    L.muavta_allocate.argtypes = [vp, i32, i32, vp, vp]
    L.muavta_step_staged.argtypes = [vp]
    L.muavta_rollout.argtypes = [vp, u64p, i32, i32, i32, i32]
    L.muavta_observe.argtypes = [vp, vp, vp, vp, vp, vp]
    L.muavta_step_result.argtypes = [vp, vp, vp]
    L.muavta_metrics.argtypes = [vp, vp]
    L.muavta_rollout_metrics.argtypes = [vp, vp]
    L.muavta_get.argtypes = [vp, i32, vp, C.c_size_t]
    L.muavta_set.argtypes = [vp, i32, vp, C.c_size_t]
    L.muavta_get_state.argtypes = [vp, vp, C.c_size_t]
    L.muavta_set_state.argtypes = [vp, vp, C.c_size_t]
    L.muavta_get_rng.argtypes = [vp, vp, C.c_size_t]
    L.muavta_set_rng.argtypes = [vp, vp, C.c_size_t]
    L.muavta_lsap.argtypes = [i32, vp, i32, i32, i32, vp, vp]
    L.muavta_lsap_impl.argtypes = [i32, vp, i32, i32, i32, vp, vp, i32]
    L.muavta_avoid_obstacles.argtypes = [i32, vp, vp, i32, vp, i32, vp]
    L.muavta_domain_math.argtypes = [i32, vp, vp, i32, vp, vp, vp]
    L.muavta_domain_log.argtypes = [i32, vp, i32, vp]
    L.muavta_domain_atan2.argtypes = [i32, vp, vp, i32, vp]
    L.muavta_device_ptrs.argtypes = [vp] + [C.POINTER(vp)] * 6
    L.muavta_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.muavta_kernel_ms_history.argtypes = [vp, vp, C.c_int32]
    L.muavta_last_seed_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.muavta_launch_gaps_ms.argtypes = [vp, vp, C.c_int32]
    L.muavta_comm_uid.argtypes = [vp]
    L.muavta_comm_init.argtypes = [vp, i32, i32, vp]
    L.muavta_allreduce_metrics.argtypes = [vp, vp, i32, vp, i32, vp, vp]
    L.muavta_comm_destroy.argtypes = [vp]
    L.muavta_rollout_record.argtypes = [vp, u64p, i32, i32, i32, i32, vp]
    L.muavta_call.argtypes = [vp, i32, i32, vp, C.c_double, vp]
    L.muavta_sync.argtypes = [vp]
    L.muavta_wait_stream.argtypes = [vp, vp]
    L.muavta_set_parts.argtypes = [vp, i32]
    L.muavta_part_range.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.muavta_rollout_part.argtypes = [vp, i32, i32, i32, i32, i32]
    L.muavta_allocate_part.argtypes = [vp, i32, i32, i32, vp, vp]
    L.muavta_step_part.argtypes = [vp, i32, vp, vp]
    L.muavta_observe_part.argtypes = [vp, i32] + [vp] * 7
    L.muavta_wait_part.argtypes = [vp, i32]
    L.muavta_refresh_observation.argtypes = [vp]
    L.muavta_set_allocator.argtypes = [vp, i32]
    L.muavta_set_release_log.argtypes = [vp, i32]
    L.muavta_tokens.argtypes = [vp, i32, i32, i32] + [vp] * 10
    L.muavta_tokens_device.argtypes = [vp, i32, i32, i32] + [vp] * 10
    L.muavta_allocate_scored.argtypes = [vp, C.POINTER(MuavtaScored), vp, vp]
    L.muavta_allocate_scored_device.argtypes = [vp, C.POINTER(MuavtaScored)]
    L.muavta_rl_step_device.argtypes = [vp, C.POINTER(MuavtaRlStep)]
    L.muavta_rl_run_device.argtypes = [vp, C.POINTER(MuavtaRlRun)]
    L.muavta_step_run.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]
    L.muavta_set_lanes.argtypes = [vp, i32]
    L.muavta_set_slot_cap.argtypes = [vp, i32]
    L.muavta_context.argtypes = [vp, i32, i32, vp]
    L.muavta_context_device.argtypes = [vp, i32, i32, vp]
    L.muavta_lanes.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.muavta_rollout_metrics_back.argtypes = [vp, i32, vp]
    L.muavta_error_flags_back.argtypes = [vp, i32, vp]
    L.muavta_abi_sizes.argtypes = [C.POINTER(i32 * 3)]
    for name in EXPORTS:
        if name != "muavta_last_error":
            getattr(L, name).restype = C.c_int
    sizes = (i32 * 3)()
    L.muavta_abi_sizes(C.byref(sizes))
    if (sizes[0], sizes[1]) != (C.sizeof(MuavtaParams), C.sizeof(MuavtaDims)):
        raise MuavtaError(f"ABI layout mismatch: library {sizes[0]}/{sizes[1]} bytes vs binding "
                          f"{C.sizeof(MuavtaParams)}/{C.sizeof(MuavtaDims)} (MuavtaParams/MuavtaDims)")
    _LIB = L
    return L
