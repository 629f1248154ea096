"""ctypes binding of libedgemarl.so (C ABI: include/edgemarl.h).  Fails loudly when the HIP library
has not been built: there is no CPU path behind this package."""
from __future__ import annotations

import ctypes as C
import os

# torch bundles its own HIP runtime (libamdhip64.so.7).  It must be the first one loaded in the
# process so that libedgemarl.so binds to the same runtime instance (same soname) and shares torch's
# device context, streams and allocations; loading /opt/rocm's copy first gives two runtimes.
import torch  # noqa: F401

from .config import EmConfig, EmmConfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libedgemarl.so")
ABI_VERSION = 8


class EmTape(C.Structure):
    _fields_ = [
        ("explore_u", C.c_void_p), ("explore_res", C.c_void_p), ("bw_u", C.c_void_p), ("net_lat", C.c_void_p),
        ("explore_tgt", C.c_void_p),
        ("rate_u", C.c_void_p), ("dt_raw", C.c_void_p),
        ("n_arrivals", C.c_void_p),
        ("task_workload", C.c_void_p), ("task_data", C.c_void_p), ("task_deadline_offset", C.c_void_p),
        ("task_decider", C.c_void_p),
        ("arrival_slots", C.c_int32),
    ]


class EmStepOut(C.Structure):
    _fields_ = [
        ("obs", C.c_void_p), ("active", C.c_void_p), ("rank", C.c_void_p), ("reward", C.c_void_p),
        ("emitted", C.c_void_p), ("done", C.c_void_p), ("step_latency", C.c_void_p), ("n_draws", C.c_void_p),
        ("reward_seen", C.c_void_p), ("chunk_totals", C.c_void_p),
    ]


class EmsStepOut(C.Structure):
    """ctypes image of ``ems_step_out`` (scaled topology, compact outputs)."""
    _fields_ = [("n_pending", C.c_void_p), ("pend_decider", C.c_void_p), ("pend_obs", C.c_void_p), ("reward", C.c_void_p),
                ("reward_seen", C.c_void_p), ("emitted", C.c_void_p), ("done", C.c_void_p), ("step_latency", C.c_void_p)]


class EmMlp3(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("w3", C.c_void_p), ("b3", C.c_void_p),
                ("in_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32)]


class EmGmlp(C.Structure):
    """ctypes image of ``em_gmlp``: n_groups networks of one shape, ``group_stride`` floats apart."""
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("w3", C.c_void_p), ("b3", C.c_void_p), ("group_stride", C.c_int64),
                ("in_dim", C.c_int32), ("hidden", C.c_int32), ("out_dim", C.c_int32), ("n_groups", C.c_int32)]


class EmRollout(C.Structure):
    _fields_ = [("n_envs", C.c_int64), ("capacity_entries", C.c_int64), ("capacity_steps", C.c_int64),
                ("n_nodes", C.c_int32), ("horizon", C.c_int32)] + [
        (name, C.c_void_p) for name in (
            "ent_obs", "ent_action", "ent_logp", "ent_env", "ent_agent", "ent_adv", "ent_ret",
            "st_state", "st_value", "st_env", "st_first", "st_k",
            "reward_tb", "done_tb", "value_tb", "k_tb", "first_tb", "srow", "scan_scratch", "counters", "err_flag",
            "obs_tb", "st_row")]


class EmsRollout(C.Structure):
    _fields_ = [("n_envs", C.c_int64), ("capacity_entries", C.c_int64), ("n_servers", C.c_int32), ("n_deciders", C.c_int32),
                ("horizon", C.c_int32), ("pad", C.c_int32)] + [
        (name, C.c_void_p) for name in (
            "ent_obs", "ent_action", "ent_logp", "ent_value", "ent_env", "ent_agent", "ent_dst", "ent_adv", "ent_ret",
            "reward_tb", "done_tb", "k_tb", "first_tb", "counters", "group_off", "perm", "err_flag", "ent_state", "value_tb", "index_scratch")]


class EmCentralCritic(C.Structure):
    _fields_ = [("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p), ("w3", C.c_void_p),
                ("b3", C.c_void_p), ("n_deciders", C.c_int32), ("width", C.c_int32), ("ldw", C.c_int32), ("pad", C.c_int32)]


class EmmTape(C.Structure):
    """ctypes image of ``emm_tape`` (EdgeMARLEnv parity mode)."""
    _fields_ = [(name, C.c_void_p) for name in ("rate_u", "dt_raw", "bw_u", "net_lat", "task_workload", "task_data",
                                                "task_deadline_offset")]


class EmmStepOut(C.Structure):
    _fields_ = [(name, C.c_void_p) for name in ("obs", "reward", "terminated", "info", "latency")]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build the HIP kernels first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C marl_amd/csrc).  marl_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.em_abi_version.restype = C.c_int
    if L.em_abi_version() != ABI_VERSION:
        raise RuntimeError("libedgemarl.so ABI version mismatch; rebuild marl_amd/csrc")
    L.em_last_error.restype = C.c_char_p
    vp, i64, u64, i32 = C.c_void_p, C.c_int64, C.c_uint64, C.c_int
    L.em_create.argtypes = [C.POINTER(EmConfig), i64, i32, u64, u64, i64, C.POINTER(vp)]
    L.em_destroy.argtypes = [vp]
    L.em_reset.argtypes = [vp, C.POINTER(EmTape), i32, u64, C.POINTER(EmStepOut), vp]
    L.em_step.argtypes = [vp, vp, vp, C.POINTER(EmTape), C.POINTER(EmStepOut), vp]
    L.em_step_raw.argtypes = [vp, vp, C.POINTER(EmTape), C.POINTER(EmStepOut), vp]
    L.em_step_greedy.argtypes = [vp, C.c_float, C.POINTER(EmTape), C.POINTER(EmStepOut), vp]
    L.em_step_many.argtypes = [vp, i32, i32, vp, vp, C.c_float, i32, C.POINTER(EmStepOut), vp]
    L.em_episode_metrics.argtypes = [vp, vp, vp]
    L.em_set_exploration.argtypes = [vp, i32, i64]
    L.em_get_exploration_counter.argtypes = [vp]
    L.em_get_exploration_counter.restype = i64
    L.em_get_state.argtypes = [vp] + [vp] * 7
    L.em_record_bytes.argtypes = [vp]
    L.em_record_bytes.restype = i64
    L.em_n_envs.argtypes = [vp]
    L.em_n_envs.restype = i64
    L.em_set_lanes_per_env.argtypes = [vp, i32]
    L.em_set_workgroup_waves.argtypes = [vp, i32]
    L.em_get_kernel_variant.argtypes = [vp, vp, vp]
    L.em_get_kernel_variant.restype = C.c_int
    L.ems_create.argtypes = [C.POINTER(EmConfig), C.c_int32, i64, i32, u64, u64, i64, C.POINTER(vp)]
    L.ems_destroy.argtypes = [vp]
    L.ems_reset.argtypes = [vp, i32, u64, C.POINTER(EmsStepOut), vp]
    L.ems_step.argtypes = [vp, vp, vp, i32, C.POINTER(EmsStepOut), vp]
    L.ems_episode_metrics.argtypes = [vp, vp, vp]
    L.ems_set_exploration.argtypes = [vp, i32, i64]
    L.ems_get_exploration_counter.argtypes = [vp]
    L.ems_get_exploration_counter.restype = i64
    L.ems_get_state.argtypes = [vp] + [vp] * 7
    L.ems_rollout_index.argtypes = [C.POINTER(EmsRollout), C.c_int32, C.c_int32, vp, vp, vp, vp]
    L.ems_rollout_index.restype = C.c_int
    L.ems_set_envs_per_wave.argtypes = [vp, i32]
    L.ems_set_envs_per_wave.restype = C.c_int
    L.ems_record_bytes.argtypes = [vp]
    L.ems_record_bytes.restype = i64
    for name in ("ems_create", "ems_destroy", "ems_reset", "ems_step", "ems_episode_metrics", "ems_set_exploration",
                 "ems_get_state"):
        getattr(L, name).restype = C.c_int
    L.em_mlp3_forward.argtypes = [C.POINTER(EmMlp3), vp, i64, i64, vp, vp]
    L.em_gauss_logp.argtypes = [vp, vp, vp, i64, C.c_int32, vp, vp, vp]
    L.em_gauss_logp.restype = C.c_int
    L.em_gauss_sample.argtypes = [vp, vp, vp, i64, C.c_int32, vp, vp]
    L.em_gauss_sample.restype = C.c_int
    L.em_format_actions.argtypes = [vp, i64, C.c_int32, vp, vp, vp]
    L.em_format_actions.restype = C.c_int
    L.em_rollout_index.argtypes = [C.POINTER(EmRollout), C.c_int32, vp, vp, vp, vp]
    L.em_rollout_index_totals.argtypes = [C.POINTER(EmRollout), C.c_int32, vp, vp, vp, vp, vp, vp]
    L.em_rollout_index_totals.restype = C.c_int
    L.em_policy_step.argtypes = [C.POINTER(EmRollout), C.c_int32, C.POINTER(EmMlp3), vp, C.POINTER(EmMlp3),
                                 C.c_float, C.c_float, vp, i32, u64, C.c_uint32, C.c_uint32, u64, vp, vp, vp]
    L.em_rollout_policy_step.argtypes = [C.POINTER(EmRollout), C.c_int32, vp, vp, vp, vp, vp, C.POINTER(EmMlp3), vp, vp, i32, u64,
                                         C.c_uint32, C.c_uint32, u64, vp, vp, vp]
    L.em_rollout_policy_step.restype = C.c_int
    L.em_group_index_scratch_ints.argtypes = [i64, i32]
    L.em_group_index_scratch_ints.restype = i64
    L.em_group_index.argtypes = [vp, i64, i32, vp, vp, vp, vp]
    L.em_group_index.restype = C.c_int
    L.em_value_step.argtypes = [C.POINTER(EmRollout), C.c_int32, C.POINTER(EmMlp3), C.c_float, C.c_float, vp]
    L.em_value_step.restype = C.c_int
    L.em_value_all.argtypes = [C.POINTER(EmRollout), i64, C.POINTER(EmMlp3), C.c_float, C.c_float, vp]
    L.em_value_all.restype = C.c_int
    f32, f64 = C.c_float, C.c_double
    L.em_gae.argtypes = [C.POINTER(EmRollout), f64, f64, vp, i64, vp, vp]
    L.em_gae_static.argtypes = [C.POINTER(EmRollout), f64, f64, vp, vp]
    L.em_gae_static.restype = C.c_int
    L.em_ppo_slab_floats.argtypes = [C.c_int32, C.c_int32, i32]
    L.em_ppo_slab_floats.restype = i64
    L.em_ppo_actor_grad.argtypes = [C.POINTER(EmMlp3), vp, vp, vp, vp, vp, vp, i64, i64, f32, f32, vp, i64, vp, vp, vp]
    L.em_ppo_critic_grad.argtypes = [C.POINTER(EmMlp3), vp, vp, vp, vp, i64, i64, f32, f32, f32, vp, i64, vp, vp, vp]
    L.em_ppo_critic_grad_rows.argtypes = [C.POINTER(EmMlp3), vp, vp, vp, vp, vp, i64, i64, f32, f32, f32, vp, i64, vp, vp, vp]
    L.em_ppo_critic_grad_rows.restype = C.c_int
    L.em_adam_step.argtypes = [vp, vp, vp, vp, i64, f64, vp, vp, vp, vp, i64, f64, i64, f64, f64, f64, f64, vp, vp]
    G = C.POINTER(EmGmlp)
    L.em_gmlp_forward.argtypes = [G, vp, i64, i64, vp, vp, C.c_int32, vp, vp]
    L.em_gmlp_policy.argtypes = [G, vp, vp, i64, i64, vp, vp, C.c_int32, vp, i32, u64, C.c_uint32, C.c_uint32, u64,
                                 vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
    L.em_gmlp_policy_gather.argtypes = [G, vp, vp, vp, i64, i64, vp, vp, C.c_int32, vp, i32, u64, C.c_uint32, C.c_uint32, u64,
                                        vp, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
    L.em_gmlp_value.argtypes = [G, vp, i64, i64, vp, vp, C.c_int32, f32, f32, vp, vp]
    L.em_gmlp_slab_floats.argtypes = [C.c_int32, C.c_int32, i32]
    L.em_gmlp_slab_floats.restype = i64
    L.em_gmlp_actor_grad.argtypes = [G, vp, vp, i64, vp, vp, vp, vp, i64, vp, vp, C.c_int32, f32, f32, vp, vp, vp, vp]
    L.em_gmlp_critic_grad.argtypes = [G, vp, i64, vp, vp, i64, vp, vp, C.c_int32, f32, f32, f32, vp, vp, vp, vp]
    L.em_group_adam.argtypes = [vp, vp, vp, vp, i64, i64, f64, i64, f64, f64, f64, f64, vp, vp]
    L.em_group_moments.argtypes = [vp, vp, vp, C.c_int32, vp, vp]
    L.em_gae_agents.argtypes = [i64, C.c_int32, C.c_int32, vp, vp, vp, vp, vp, vp, f64, f64, vp, vp, vp, vp]
    CC = C.POINTER(EmCentralCritic)
    L.em_cc_forward.argtypes = [CC, vp, vp, vp, C.c_int32, vp, i64, vp, vp, f32, f32, vp, vp]
    L.em_cc_scratch_floats.argtypes = [CC, C.c_int32, i64]
    L.em_cc_scratch_floats.restype = i64
    L.em_cc_grad.argtypes = [CC, vp, vp, vp, vp, vp, C.c_int32, vp, i64, vp, vp, i64, f32, f32, f32, vp, vp, vp, vp, vp]
    L.em_cc_forward.restype = C.c_int
    L.em_cc_grad.restype = C.c_int
    for name in ("em_gmlp_forward", "em_gmlp_policy", "em_gmlp_policy_gather", "em_gmlp_value", "em_gmlp_actor_grad", "em_gmlp_critic_grad",
                 "em_group_adam", "em_group_moments", "em_gae_agents"):
        getattr(L, name).restype = C.c_int
    for name in ("em_mlp3_forward", "em_rollout_index", "em_policy_step", "em_gae", "em_ppo_actor_grad",
                 "em_ppo_critic_grad", "em_adam_step"):
        getattr(L, name).restype = C.c_int
    L.emm_create.argtypes = [C.POINTER(EmmConfig), i64, i32, u64, u64, C.POINTER(vp)]
    L.emm_destroy.argtypes = [vp]
    L.emm_reset.argtypes = [vp, C.POINTER(EmmTape), i32, u64, C.POINTER(EmmStepOut), vp]
    L.emm_step.argtypes = [vp, vp, vp, C.POINTER(EmmTape), C.POINTER(EmmStepOut), vp]
    L.emm_get_state.argtypes = [vp, vp, vp, vp, vp, vp]
    L.emm_set_lanes_per_env.argtypes = [vp, i32]
    L.emm_set_lanes_per_env.restype = C.c_int
    for name in ("emm_create", "emm_destroy", "emm_reset", "emm_step", "emm_get_state"):
        getattr(L, name).restype = C.c_int
    for name in ("em_create", "em_destroy", "em_reset", "em_step", "em_episode_metrics", "em_set_exploration",
                 "em_get_state", "em_set_lanes_per_env", "em_set_workgroup_waves", "em_step_raw", "em_step_greedy", "em_step_many"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def check(rc: int, what: str = "libedgemarl"):
    if rc != 0:
        msg = lib().em_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed ({rc}): {msg}")
