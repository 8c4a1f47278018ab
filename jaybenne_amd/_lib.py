"""ctypes binding of the C ABI (include/jaybenne_amd.h) exported by ``libjaybenne_amd.so``.

There is no fallback: if the HIP library is missing or a symbol is absent, importing the product
API fails loudly (build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C jaybenne_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JAYBENNE_AMD_LIB selects another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("JAYBENNE_AMD_LIB") or os.path.join(_HERE, "libjaybenne_amd.so")

JB_COMPLETE, JB_ITERATE, JB_INCOMPLETE = 0, 1, 2
JB_ERR_INVALID, JB_ERR_HIP, JB_ERR_CAPACITY, JB_ERR_UNSUPPORTED = -1, -2, -3, -4
JB_SOURCE_THERMAL, JB_SOURCE_EMISSION = 0, 1
JB_STRATEGY_UNIFORM, JB_STRATEGY_ENERGY = 0, 1
JB_ST_ACTIVE, JB_ST_ABSORBED, JB_ST_ESCAPED, JB_ST_OUTGOING, JB_ST_OUTGOING_ABSORBED = 0, 1, 2, 3, 4
JB_RECORD_WORDS = 13

_dpp = C.POINTER(C.c_void_p)


class Params(C.Structure):
    _fields_ = [("num_particles", C.c_int64), ("dt", C.c_double),
                ("min_swarm_occupancy", C.c_double), ("numin", C.c_double), ("numax", C.c_double),
                ("tau_ddmc", C.c_double), ("unique_rank_seeds", C.c_int32), ("seed", C.c_int32),
                ("max_transport_iterations", C.c_int32), ("use_ddmc", C.c_int32),
                ("source_strategy", C.c_int32), ("do_emission", C.c_int32),
                ("do_feedback", C.c_int32), ("rank", C.c_int32)]


class Eos(C.Structure):
    _fields_ = [("model", C.c_int32), ("pad", C.c_int32), ("gm1", C.c_double), ("cv", C.c_double)]


class Opacity(C.Structure):
    _fields_ = [("model", C.c_int32), ("pad", C.c_int32), ("kappa", C.c_double),
                ("c", C.c_double), ("sb", C.c_double), ("time_scale", C.c_double),
                ("mass_scale", C.c_double), ("length_scale", C.c_double),
                ("temperature_scale", C.c_double)]


class Scattering(C.Structure):
    _fields_ = [("model", C.c_int32), ("pad", C.c_int32), ("kappa_s", C.c_double),
                ("apm", C.c_double), ("time_scale", C.c_double), ("mass_scale", C.c_double),
                ("length_scale", C.c_double), ("temperature_scale", C.c_double)]


ARITH_EXACT, ARITH_LEAN = 0, 1   # enum of jb_set_arithmetic

FIELD_IDS = {"rho": 0, "sie": 1, "u": 2, "fleck": 3, "tally": 4, "edelta": 5}   # enum jb_field
FIELD_NAMES = ("rho", "sie", "u", "fleck", "tally", "edelta", "src_ew", "src_num", "P1", "P2", "P3")


class MeshView(C.Structure):
    _fields_ = ([("ndim", C.c_int32), ("ng", C.c_int32), ("nblocks", C.c_int32),
                 ("nblocks_total", C.c_int32), ("nx", C.c_int32 * 3), ("nleaf", C.c_int32 * 3),
                 ("bc", C.c_int32 * 6), ("rank", C.c_int32), ("pad", C.c_int32),
                 ("gmin", C.c_double * 3), ("gmax", C.c_double * 3),
                 ("leaf_map", C.c_void_p), ("owner", C.c_void_p), ("local_index", C.c_void_p),
                 ("gid", C.c_void_p), ("owned", C.c_void_p), ("blk_xmin", C.c_void_p), ("blk_xmax", C.c_void_p),
                 ("blk_dx", C.c_void_p), ("blk_level", C.c_void_p), ("blk_nbr_lev", C.c_void_p)] +
                [(n, C.c_void_p) for n in FIELD_NAMES])


SWARM_F64 = ("x", "y", "z", "vx", "vy", "vz", "t", "w", "e")
SWARM_I32 = ("ip", "jp", "kp", "blk", "status")


class SwarmView(C.Structure):
    _fields_ = ([("n", C.c_int64), ("capacity", C.c_int64)] +
                [(n, C.c_void_p) for n in SWARM_F64 + SWARM_I32] +
                [("id", C.c_void_p), ("rng", C.c_void_p)])


class TransportStats(C.Structure):
    _fields_ = [("n_census", C.c_int64), ("n_absorbed", C.c_int64), ("n_escaped", C.c_int64),
                ("n_outgoing", C.c_int64), ("n_events", C.c_int64),
                ("n_wave_passes", C.c_int64), ("n_wave_services", C.c_int64)]


class DebugStep(C.Structure):
    _fields_ = ([(n, C.c_double) for n in ("t_start", "dt", "ff", "aa", "ss", "vv", "dx_push")] +
                [("multi_d", C.c_int32), ("three_d", C.c_int32)] +
                [(n, C.c_double) for n in ("xl", "yl", "zl", "xu", "yu", "zu", "Px_l", "Py_l",
                                           "Pz_l", "Px_u", "Py_u", "Pz_u", "t", "x", "y", "z",
                                           "vx", "vy", "vz")] +
                [(n, C.c_int32) for n in ("ip", "jp", "kp", "is_absorbed", "is_scattered",
                                          "is_rejected")])


# jb_exchange_transport (include/jaybenne_amd.h): the two collectives of the hand-off on device buffers
ALL_GATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
ALL_TO_ALL_V_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                              C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int, C.c_void_p)


class ExchangeTransport(C.Structure):
    _fields_ = [("handle", C.c_void_p), ("all_gather_u64", ALL_GATHER_FN), ("all_to_all_v", ALL_TO_ALL_V_FN)]


# every entry point include/jaybenne_amd.h declares: name -> (restype, argtypes)
_vp, _i64, _f64, _int = C.c_void_p, C.c_int64, C.c_double, C.c_int
PROTOTYPES = {
    "jb_last_error": (C.c_char_p, []),
    "jb_version": (C.c_char_p, []),
    "jb_initialize": (_int, [C.POINTER(Params), C.POINTER(Eos), C.POINTER(Opacity),
                             C.POINTER(Scattering), _int, C.POINTER(_vp)]),
    "jb_finalize": (_int, [_vp]),
    "jb_set_stream": (_int, [_vp, _vp]),
    "jb_synchronize": (_int, [_vp]),
    "jb_param_seed": (C.c_int32, [_vp]),
    "jb_mesh_create": (_int, [_vp, C.POINTER(MeshView), C.POINTER(_vp)]),
    "jb_mesh_destroy": (_int, [_vp]),
    "jb_update_derived_transport_fields": (_int, [_vp, _vp, _f64]),
    "jb_source_photons_count": (_int, [_vp, _vp, _int, _f64, _int, C.c_uint32, _vp, _vp]),
    "jb_source_photons_fill": (_int, [_vp, _vp, C.POINTER(SwarmView), _int, _f64, _f64, _vp, _vp,
                                      _vp, _vp]),
    "jb_source_photons_fill_range": (_int, [_vp, _vp, C.POINTER(SwarmView), _int, _f64, _f64, _vp, _vp,
                                            _vp, _vp, _vp, _int]),
    "jb_transport_photons": (_int, [_vp, _vp, C.POINTER(SwarmView), _f64, _f64, _i64, _i64, _int]),
    "jb_transport_photons_ddmc": (_int, [_vp, _vp, C.POINTER(SwarmView), _f64, _f64, _i64, _i64,
                                         _int]),
    "jb_last_transport_variant": (C.c_char_p, [_vp]),
    "jb_mesh_exact_geometry": (_int, [_vp]),
    "jb_mesh_ddmc_classes": (_int, [_vp]),
    "jb_get_transport_stats": (_int, [_vp, C.POINTER(TransportStats), _int]),
    "jb_sample_ddmc_block_face": (_int, [_vp, _vp, C.POINTER(SwarmView), _i64, _i64]),
    "jb_check_completion": (_int, [_vp, C.POINTER(SwarmView), _f64, C.POINTER(_i64)]),
    "jb_zero_energy_tally": (_int, [_vp, _vp]),
    "jb_evaluate_radiation_energy": (_int, [_vp, _vp, C.POINTER(SwarmView)]),
    "jb_update_fluid": (_int, [_vp, _vp]),
    "jb_photon_reflect_bc": (_int, [_vp, _vp, C.POINTER(SwarmView), _int]),
    "jb_remove_marked_particles": (_int, [_vp, C.POINTER(SwarmView)]),
    "jb_defrag_particles": (_int, [_vp, _vp, C.POINTER(SwarmView)]),
    "jb_release_scratch": (_int, [_vp]),
    "jb_defrag_policy": (_int, [_vp, _vp, C.POINTER(SwarmView), _i64, C.c_int32, C.POINTER(C.c_int32)]),
    "jb_pack_outgoing": (_int, [_vp, _vp, C.POINTER(SwarmView), _i64, _i64, _int, _vp, _i64, _vp]),
    "jb_unpack_incoming": (_int, [_vp, _vp, C.POINTER(SwarmView), _vp, _i64]),
    "jb_transport_rccl": (_int, [_vp, _int, _int, _vp]),
    "jb_transport_release": (_int, [_vp]),
    "jb_exchange": (_int, [_vp, _vp, C.POINTER(SwarmView), _i64, _i64, _int, _int, _vp, _vp, _i64, _vp, _i64,
                           C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "jb_gather_cells": (_int, [_vp, _vp, _int, _i64, _vp, _vp, _vp]),
    "jb_fill_cells": (_int, [_vp, _vp, _int, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "jb_estimate_timestep": (_f64, [_vp]),
    "jb_radiation_step": (_int, [_vp, _vp, C.POINTER(SwarmView), _f64, _f64,
                                 C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), _vp]),
    "jb_range_push": (_int, [C.c_char_p]),
    "jb_range_pop": (_int, []),
    "jb_ranges_enabled": (_int, []),
    "jb_debug_philox": (_int, [_vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                               C.POINTER(C.c_uint32)]),
    "jb_debug_rocrand_philox": (_int, [_vp, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint32)]),
    "jb_debug_seed_state": (_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "jb_debug_stream_start": (_int, [_vp, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]),
    "jb_debug_draw_stream": (_int, [_vp, C.c_uint64, _int, _vp, C.POINTER(C.c_uint64)]),
    "jb_set_arithmetic": (_int, [_vp, _int]),
    "jb_get_arithmetic": (_int, [_vp]),
    "jb_debug_math": (_int, [_vp, _int, _vp, _int, _vp]),
    "jb_debug_model_coefficients": (_int, [_vp, C.POINTER(C.c_double * 4)]),
    "jb_debug_model_eval": (_int, [_vp, _int, _vp, _int, _vp]),
    "jb_debug_step_call": (_int, [_vp, _int, C.POINTER(DebugStep), _vp, _int, C.POINTER(_int)]),
    "jb_debug_sample_call": (_int, [_vp, _int, _vp, _vp, _vp, _int, _vp, _vp, C.POINTER(_int)]),
}

_lib = None


class JaybenneError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"jaybenne_amd status {status}: {message}")
        self.status = status


def load():
    """Load the shared library and bind every declared symbol (AttributeError if one is absent)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension has not been built "
                "(run __graft_entry__.build() or make -C jaybenne_amd/csrc). "
                "There is no CPU fallback for the product path.")
        # PyTorch-ROCm ships its own HIP runtime; it has to be in the process before this library
        # is, so that both resolve libamdhip64 to the same copy (two runtimes in one process do
        # not see each other's devices or allocations)
        import torch  # noqa: F401
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(status: int) -> int:
    if status < 0:
        raise JaybenneError(status, load().jb_last_error().decode())
    return status
