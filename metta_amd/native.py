"""ctypes binding of libmettagrid_hip.so (the C-ABI declared in include/mettagrid_hip.h).

There is deliberately no fallback: if the HIP library is missing or cannot be loaded, importing a symbol from
it raises.  The CPU restatement under oracle/ is test infrastructure and is never reachable from here.
"""

from __future__ import annotations

import ctypes as ct
import os
from pathlib import Path

# METTA_AMD_LIB: another build of the same library (kernel experiments, tools/variants.sh)
_LIB_PATH = Path(os.environ.get("METTA_AMD_LIB") or Path(__file__).resolve().parent / "libmettagrid_hip.so")
# the wide-inventory build of the same sources (-DMGD_MAX_R=32): resource lists of 17..32 names
_LIB_PATH_WIDE = Path(__file__).resolve().parent / "libmettagrid_hip_r32.so"
NARROW_MAX_RESOURCES, WIDE_MAX_RESOURCES = 16, 32
_lib = None
_lib_wide = None


class MgBuffers(ct.Structure):
    _fields_ = [
        ("observations", ct.c_void_p),
        ("rewards", ct.c_void_p),
        ("terminals", ct.c_void_p),
        ("truncations", ct.c_void_p),
        ("actions", ct.c_void_p),
        ("masks", ct.c_void_p),
        ("episode_rewards", ct.c_void_p),
        ("action_success", ct.c_void_p),
        ("num_envs", ct.c_int32),
        ("num_agents", ct.c_int32),
        ("num_tokens", ct.c_int32),
        ("num_actions", ct.c_int32),
    ]


# name -> (restype, argtypes); every symbol include/mettagrid_hip.h declares
_vp, _i32p, _u32p, _u8p, _f32p, _f64p = (
    ct.c_void_p,
    ct.POINTER(ct.c_int32),
    ct.POINTER(ct.c_uint32),
    ct.POINTER(ct.c_uint8),
    ct.POINTER(ct.c_float),
    ct.POINTER(ct.c_double),
)
SYMBOLS = {
    "mg_last_error": (ct.c_char_p, []),
    "mg_version": (ct.c_char_p, []),
    "mg_max_resources": (ct.c_int, []),
    "mg_create": (ct.c_int, [_i32p, ct.c_size_t, ct.c_int, ct.c_int, ct.POINTER(_vp)]),
    "mg_set_maps": (ct.c_int, [_vp, _u8p, ct.c_int, ct.c_int, ct.c_int]),
    "mg_set_maps_ragged": (ct.c_int, [_vp, _u8p, ct.c_int, _i32p, _i32p]),
    "mg_set_map_max_steps": (ct.c_int, [_vp, _i32p]),
    "mg_set_config_slots": (ct.c_int, [_vp, ct.c_int, ct.c_int]),
    "mg_write_config_slot": (ct.c_int, [_vp, ct.c_int, _i32p, ct.c_size_t, _vp]),
    "mg_assign_map_configs": (ct.c_int, [_vp, ct.c_int, _i32p, _i32p, _vp]),
    "mg_finished_ring_capacity": (ct.c_int, [_vp]),
    "mg_finished_ring": (ct.c_int, [_vp, ct.c_uint32, ct.c_uint32, _i32p, _i32p, _vp]),
    "mg_register_config": (ct.c_int, [_vp, _i32p, ct.c_size_t, _i32p]),
    "mg_write_config_slots": (ct.c_int, [_vp, ct.c_int, _i32p, _i32p, _vp]),
    "mg_set_generated_maps": (ct.c_int, [_vp, ct.c_int, ct.c_int, ct.c_int, ct.c_int, _i32p, _i32p, _i32p, _i32p, ct.c_int, ct.c_uint64]),
    "mg_refresh_maps": (ct.c_int, [_vp, ct.c_int, _vp]),
    "mg_get_map": (ct.c_int, [_vp, ct.c_int, _u8p, _vp]),
    "mg_write_map_slot": (ct.c_int, [_vp, ct.c_int, _u8p, _vp]),
    "mg_map_dims": (ct.c_int, [_vp, ct.c_int, _i32p, _i32p]),
    "mg_reset": (ct.c_int, [_vp, _i32p, ct.c_int, _u32p, _i32p, _vp]),
    "mg_step": (ct.c_int, [_vp, _vp]),
    "mg_step_actions": (ct.c_int, [_vp, _vp, _vp]),
    "mg_step_groups": (ct.c_int, [ct.c_int, ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p), ct.POINTER(ct.c_void_p)]),
    "mg_step_noalloc": (ct.c_int, [_vp, _vp, _vp]),
    "mg_status_buffer": (ct.c_int, [_vp, ct.POINTER(ct.POINTER(ct.c_uint32)), _i32p]),
    "mg_check_errors": (ct.c_int, [_vp, _vp]),
    "mg_debug_inject_error": (ct.c_int, [_vp, ct.c_int, ct.c_uint32]),
    "mg_step_path": (ct.c_char_p, [_vp]),
    "mg_set_auto_reset": (ct.c_int, [_vp, ct.c_int, ct.c_int]),
    "mg_set_early_reset": (ct.c_int, [_vp, _u32p]),
    "mg_get_buffers": (ct.c_int, [_vp, ct.POINTER(MgBuffers)]),
    "mg_end_episode": (ct.c_int, [_vp, ct.c_int, _vp]),
    "mg_episode_stats": (ct.c_int, [_vp, ct.c_int, ct.c_int, _f32p, _u8p, _f32p, _u8p, _vp]),
    "mg_stat_slots": (ct.c_int, [_vp, ct.POINTER(ct.c_int), ct.POINTER(ct.c_int)]),
    "mg_agent_state": (ct.c_int, [_vp, ct.c_int, _i32p, _vp]),
    "mg_num_assemblers": (ct.c_int, [_vp, ct.c_int]),
    "mg_agent_stat_reward": (ct.c_int, [_vp, ct.c_int, _f32p, _vp]),
    "mg_chest_state": (ct.c_int, [_vp, ct.c_int, _i32p, ct.c_int, _i32p, _vp]),
    "mg_assembler_state": (ct.c_int, [_vp, ct.c_int, _i32p, _vp]),
    "mg_clipper_state": (ct.c_int, [_vp, ct.c_int, ct.POINTER(ct.c_uint32), _i32p, _vp]),
    "mg_env_info": (ct.c_int, [_vp, ct.c_int, _u32p, _i32p, _u32p, _u32p, _vp]),
    "mg_env_table": (ct.c_int, [_vp, _u32p, _vp]),
    "mg_finished_rewards": (ct.c_int, [_vp, ct.c_int, _f32p, _u32p, _vp]),
    "mg_finished_batch": (ct.c_int, [_vp, ct.c_int, _i32p, _f32p, _u8p, _f32p, _u8p, _f32p, _u32p, _vp]),
    "mg_set_inventory": (ct.c_int, [_vp, ct.c_int, ct.c_int, ct.c_int, _i32p, _i32p, _vp]),
    "mg_summary_buffer": (ct.c_int, [_vp, ct.POINTER(ct.POINTER(ct.c_double)), _i32p]),
    "mg_finished_summary": (ct.c_int, [_vp, _f64p, ct.c_int, _vp]),
    "mg_set_profiling": (ct.c_int, [_vp, ct.c_int]),
    "mg_set_profiling_stride": (ct.c_int, [_vp, ct.c_int]),
    "mg_profile_read": (ct.c_int, [_vp, _f32p, ct.c_int, ct.POINTER(ct.c_int)]),
    "mg_debug_stamps": (ct.c_int, [_vp, ct.POINTER(ct.c_uint64), ct.c_int]),
    "mg_destroy": (ct.c_int, [_vp]),
    "mg_compute_advantage": (
        ct.c_int,
        [_vp, _vp, _vp, _vp, _vp, ct.c_float, ct.c_float, ct.c_float, ct.c_float, ct.c_int, ct.c_int, _vp],
    ),
}


def lib_path(wide: bool = False) -> Path:
    return _LIB_PATH_WIDE if wide else _LIB_PATH


def load(wide: bool = False) -> ct.CDLL:
    """Load the HIP library (built in-tree by `__graft_entry__.build()` / metta_amd/csrc/Makefile).  `wide`: the
    32-resource build."""
    global _lib, _lib_wide
    have = _lib_wide if wide else _lib
    if have is not None:
        return have
    path = lib_path(wide)
    if not path.exists():
        raise ImportError(
            f"{path} is missing: build it with `make -C metta_amd/csrc` (hipcc --offload-arch=gfx950). "
            "The device path has no CPU fallback."
        )
    # torch ships its own libamdhip64: it must be the process's HIP runtime (tensors are exchanged by pointer), so
    # it is loaded first and this library binds to the same one by soname
    import torch  # noqa: F401

    L = ct.CDLL(str(path))
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if wide:
        _lib_wide = L
    else:
        _lib = L
    return L


def load_for(num_resources: int) -> ct.CDLL:
    """The build whose inventories hold `num_resources` (more than 32: the narrow one, whose mg_create says why not)."""
    return load(wide=NARROW_MAX_RESOURCES < num_resources <= WIDE_MAX_RESOURCES)


def check(rc: int, L: ct.CDLL | None = None) -> None:
    if rc != 0:
        raise RuntimeError((L or load()).mg_last_error().decode())
