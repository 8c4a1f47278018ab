"""ctypes front end of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY: imported by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by jaybenne_amd/.

The mesh is described by any object exposing the attributes of ``jaybenne_amd.mesh.Mesh``
(ndim, ng, nx, nleaf, swarm_bc, gmin, gmax, leaf_map, blk_xmin, blk_xmax, blk_dx, blk_level,
blk_nbr_lev, field_shape, ncell); the oracle does not import the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liborc.so")

MATH_LIBM, MATH_PORTABLE = 0, 1
SRC_THERMAL, SRC_EMISSION = 0, 1
ST_ACTIVE, ST_ABSORBED, ST_ESCAPED = 0, 1, 2

FIELD_NAMES = ("rho", "sie", "u", "fleck", "tally", "edelta", "src_ew", "src_num", "P1", "P2", "P3")
SWARM_F64 = ("x", "y", "z", "vx", "vy", "vz", "t", "w", "e")
SWARM_I32 = ("ip", "jp", "kp", "blk", "status")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Params(C.Structure):
    _fields_ = [("num_particles", C.c_int64), ("dt", C.c_double), ("tau_ddmc", C.c_double),
                ("c", C.c_double), ("sb", C.c_double), ("cv", C.c_double),
                ("kappa_a", C.c_double), ("kappa_s", C.c_double), ("apm", C.c_double),
                ("seed", C.c_int32), ("use_ddmc", C.c_int32), ("do_emission", C.c_int32),
                ("do_feedback", C.c_int32), ("opac_model", C.c_int32), ("pad_model", C.c_int32),
                ("ep_A", C.c_double), ("ep_B", C.c_double), ("ep_E", C.c_double)]


class MeshC(C.Structure):
    _fields_ = [("ndim", C.c_int32), ("ng", C.c_int32), ("nblocks", C.c_int32), ("pad0", C.c_int32),
                ("nx", C.c_int32 * 3), ("nleaf", C.c_int32 * 3), ("bc", C.c_int32 * 6),
                ("gmin", C.c_double * 3), ("gmax", C.c_double * 3),
                ("leaf_map", _ip), ("blk_xmin", _dp), ("blk_xmax", _dp), ("blk_dx", _dp),
                ("blk_level", _ip), ("blk_nbr_lev", _ip)] + [(n, _dp) for n in FIELD_NAMES]


class SwarmC(C.Structure):
    _fields_ = ([("n", C.c_int64), ("cap", C.c_int64)] + [(n, _dp) for n in SWARM_F64] +
                [(n, _ip) for n in SWARM_I32] +
                [("id", C.POINTER(C.c_uint64)), ("rng", C.POINTER(C.c_uint64))])


class Step(C.Structure):
    """Mirror of orc_step (oracle/orc_steps.h)."""
    _fields_ = ([(n, C.c_double) for n in ("t_start", "dt", "ff", "aa", "ss", "vv", "dx_push")] +
                [("multi_d", C.c_int), ("three_d", C.c_int)] +
                [(n, C.c_double) for n in ("xl", "yl", "zl", "xu", "yu", "zu", "Px_l", "Py_l",
                                           "Pz_l", "Px_u", "Py_u", "Pz_u", "t", "x", "y", "z",
                                           "vx", "vy", "vz")] +
                [(n, C.c_int) for n in ("ip", "jp", "kp", "is_absorbed", "is_scattered",
                                        "is_rejected")])


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("orc.c", "orc.h", "orc_steps.h", "orc_math.h",
                                             "orc_rng.h", "orc_tables.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH) or
             any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs))
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B", "liborc.so"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_transport_photons.restype = C.c_uint64
        L.orc_transport_photons.argtypes = [C.POINTER(MeshC), C.POINTER(Params), C.POINTER(SwarmC),
                                            C.c_double, C.c_double, C.c_int64, C.c_int64]
        L.orc_check_completion.restype = C.c_int64
        L.orc_check_completion.argtypes = [C.POINTER(SwarmC), C.c_double]
        L.orc_remove_marked.restype = C.c_int64
        L.orc_update_derived_transport_fields.argtypes = [C.POINTER(MeshC), C.POINTER(Params),
                                                          C.c_double]
        L.orc_source_count.argtypes = [C.POINTER(MeshC), C.POINTER(Params), C.c_int, C.c_double,
                                       C.c_int, C.c_uint32, _ip, _ip]
        L.orc_source_fill.argtypes = [C.POINTER(MeshC), C.POINTER(Params), C.POINTER(SwarmC),
                                      C.c_int, C.c_double, C.c_double, _ip,
                                      C.POINTER(C.c_int64), C.POINTER(C.c_uint64)]
        L.orc_seed_state.restype = C.c_uint64
        L.orc_seed_state.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.orc_stream_start.restype = C.c_uint64
        L.orc_stream_start.argtypes = [C.c_uint32, C.c_uint64]
        L.orc_draw_stream.restype = C.c_uint64
        L.orc_draw_stream.argtypes = [C.c_uint64, C.c_int, _dp]
        L.orc_call_scatter.argtypes = [C.c_double, _dp, C.c_int, _dp]
        L.orc_call_face_iso_dir.argtypes = [C.c_double, _dp, C.c_int, _dp]
        L.orc_call_planck.argtypes = [C.c_double, C.c_double, _dp, C.c_int, _dp]
        L.orc_call_face_2d.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, _dp, C.c_int,
                                       C.POINTER(C.c_int), _dp]
        L.orc_call_face_3d.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, _dp, _dp, C.c_int,
                                       C.POINTER(C.c_int), _dp]
        assert L.orc_sizeof_step() == C.sizeof(Step), "orc_step layout mismatch"
        _lib = L
    return _lib


def _d(a: np.ndarray):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _i(a: np.ndarray):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def set_math_mode(mode: int) -> None:
    lib().orc_set_math_mode(int(mode))


def set_threads(n: int) -> None:
    lib().orc_set_threads(int(n))


# ---------------------------------------------------------------------- scalar-level helpers
def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().orc_philox(c, k, o)
    return [int(v) for v in o]


def seed_state(seed: int, domain: int, sid: int) -> int:
    return int(lib().orc_seed_state(seed, domain, sid))


def stream_start(seed: int, pid: int) -> int:
    """First state of the random stream of the particle with creation index ``pid``."""
    return int(lib().orc_stream_start(seed, pid))


def draw_stream(state: int, n: int):
    """n uniforms from a stream that starts in `state`; returns (uniforms, final state)."""
    out = np.empty(n)
    final = lib().orc_draw_stream(state, n, _d(out))
    return out, int(final)


def math_log(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_math_log(_d(x), x.size, _d(out))
    return out


def math_sincos(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    s, c = np.empty_like(x), np.empty_like(x)
    lib().orc_math_sincos(_d(x), x.size, _d(s), _d(c))
    return s, c


def math_sincos2pi(u):
    """sin, cos of 2 pi u (the azimuth form the step functions use)."""
    u = np.ascontiguousarray(u, dtype=np.float64)
    s, c = np.empty_like(u), np.empty_like(u)
    lib().orc_math_sincos2pi(_d(u), u.size, _d(s), _d(c))
    return s, c


def math_one_minus_exp_neg(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_math_one_minus_exp_neg(_d(x), x.size, _d(out))
    return out


def model_coefficients(time_scale=1.0, mass_scale=1.0, length_scale=1.0, temperature_scale=1.0):
    """EPBremss coefficients in code units and the ThomsonS-as-GrayS kappa_s (orc.h)."""
    sc = (C.c_double * 4)(time_scale, mass_scale, length_scale, temperature_scale)
    out = (C.c_double * 4)()
    lib().orc_model_coefficients(sc, out)
    return dict(ep_A=out[0], ep_B=out[1], ep_E=out[2], kappa_s_thomson=out[3])


def model_eval(params: dict, which: int, rho, temp, nu):
    """which: 0 absorption coefficient, 1 emissivity, 2 scattering coefficient."""
    P = Params()
    for k, v in params.items():
        setattr(P, k, v)
    x = np.ascontiguousarray(np.stack([np.asarray(rho, dtype=np.float64),
                                       np.asarray(temp, dtype=np.float64),
                                       np.asarray(nu, dtype=np.float64)], axis=-1).reshape(-1, 3))
    out = np.empty(x.shape[0])
    lib().orc_model_eval(C.byref(P), int(which), _d(x), x.shape[0], _d(out))
    return out


def math_acos(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.empty_like(x)
    lib().orc_math_acos(_d(x), x.size, _d(out))
    return out


def call_step(which: str, st: Step, tape) -> int:
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    fn = {"transport": lib().orc_call_transport_step, "ddmc": lib().orc_call_ddmc_step,
          "albedo": lib().orc_call_ddmc_albedo}[which]
    return int(fn(C.byref(st), _d(tape), tape.size))


def call_scatter(vv, tape):
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    v = np.zeros(3)
    n = lib().orc_call_scatter(vv, _d(tape), tape.size, _d(v))
    return v, n


def call_face_iso_dir(vv, tape):
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    v = np.zeros(3)
    n = lib().orc_call_face_iso_dir(vv, _d(tape), tape.size, _d(v))
    return v, n


def call_planck(sb, temp, tape):
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    e = np.zeros(1)
    n = lib().orc_call_planck(sb, temp, _d(tape), tape.size, _d(e))
    return float(e[0]), n


def call_face_2d(i_l, dx, P_l, P_u, tape, i0, x0):
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    i = C.c_int(i0)
    x = np.array([x0])
    n = lib().orc_call_face_2d(i_l, dx, P_l, P_u, _d(tape), tape.size, C.byref(i), _d(x))
    return i.value, float(x[0]), n


def call_face_3d(i1_l, i2_l, dx1, dx2, P4, tape, ij0, x0):
    tape = np.ascontiguousarray(tape, dtype=np.float64)
    P = np.ascontiguousarray(P4, dtype=np.float64)
    ij = (C.c_int * 2)(*ij0)
    x = np.array(x0, dtype=np.float64)
    n = lib().orc_call_face_3d(i1_l, i2_l, dx1, dx2, _d(P), _d(tape), tape.size, ij, _d(x))
    return [ij[0], ij[1]], x, n


# ---------------------------------------------------------------------- task-level front end
class Oracle:
    """Holds host copies of the fields and the photon swarm of one problem and runs the
    reference's task sequence on them."""

    def __init__(self, mesh, params: Dict[str, float], capacity: int, math_mode: int = MATH_LIBM,
                 threads: int = 1):
        self.mesh = mesh
        self.math_mode = math_mode
        self.threads = threads
        self.P = Params()
        for k, v in params.items():
            setattr(self.P, k, v)
        self.fields = {n: np.zeros(mesh.field_shape) for n in FIELD_NAMES}
        self.cap = int(capacity)
        self.sw = {n: np.zeros(self.cap) for n in SWARM_F64}
        self.sw.update({n: np.zeros(self.cap, dtype=np.int32) for n in SWARM_I32})
        self.sw["id"] = np.zeros(self.cap, dtype=np.uint64)
        self.sw["rng"] = np.zeros(self.cap, dtype=np.uint64)
        self.n = 0
        self.next_id = 0
        self.cycle = 0   # RadiationStep counter: keys the per-cell rounding streams as the hosts do (SourceEpoch)
        self._keep = dict(leaf_map=np.ascontiguousarray(mesh.leaf_map, dtype=np.int32),
                          xmin=np.ascontiguousarray(mesh.blk_xmin), xmax=np.ascontiguousarray(mesh.blk_xmax),
                          dx=np.ascontiguousarray(mesh.blk_dx),
                          lev=np.ascontiguousarray(mesh.blk_level, dtype=np.int32),
                          nbr=np.ascontiguousarray(mesh.blk_nbr_lev, dtype=np.int32))
        self.events = 0

    # -- struct views
    def _mesh_c(self) -> MeshC:
        m, k = self.mesh, self._keep
        M = MeshC()
        M.ndim, M.ng, M.nblocks = m.ndim, m.ng, m.nblocks
        M.nx = (C.c_int32 * 3)(*m.nx)
        M.nleaf = (C.c_int32 * 3)(*m.nleaf)
        M.bc = (C.c_int32 * 6)(*m.swarm_bc)
        M.gmin = (C.c_double * 3)(*m.gmin)
        M.gmax = (C.c_double * 3)(*m.gmax)
        M.leaf_map, M.blk_level, M.blk_nbr_lev = _i(k["leaf_map"]), _i(k["lev"]), _i(k["nbr"])
        M.blk_xmin, M.blk_xmax, M.blk_dx = _d(k["xmin"]), _d(k["xmax"]), _d(k["dx"])
        for n in FIELD_NAMES:
            setattr(M, n, _d(self.fields[n]))
        return M

    def _swarm_c(self) -> SwarmC:
        S = SwarmC()
        S.n, S.cap = self.n, self.cap
        for n in SWARM_F64:
            setattr(S, n, _d(self.sw[n]))
        for n in SWARM_I32:
            setattr(S, n, _i(self.sw[n]))
        S.id = self.sw["id"].ctypes.data_as(C.POINTER(C.c_uint64))
        S.rng = self.sw["rng"].ctypes.data_as(C.POINTER(C.c_uint64))
        return S

    def _enter(self):
        set_math_mode(self.math_mode)
        set_threads(self.threads)

    # -- tasks (names follow reference jaybenne.hpp:59-76)
    def UpdateDerivedTransportFields(self, dt: float) -> None:
        self._enter()
        M = self._mesh_c()
        lib().orc_update_derived_transport_fields(C.byref(M), C.byref(self.P), dt)

    def SourcePhotons(self, source_type: int, t_start: float, dt: float,
                      blocks_in_call: Optional[int] = None) -> int:
        """blocks_in_call = 1 reproduces the per-MeshBlockData initialisation path
        (reference jaybenne.cpp:570-574), nblocks the per-MeshData emission path."""
        self._enter()
        if source_type == SRC_EMISSION and not self.P.do_emission:
            return 0
        m = self.mesh
        if blocks_in_call is None:
            blocks_in_call = m.nblocks
        M = self._mesh_c()
        nper = np.zeros(m.nblocks, dtype=np.int32)
        prefix = np.zeros(m.nblocks * m.ncell, dtype=np.int32)
        lib().orc_source_count(C.byref(M), C.byref(self.P), source_type, dt, blocks_in_call,
                               (self.cycle if source_type == SRC_EMISSION else (0 if self.cycle == 0 else (1 << 19) | self.cycle)),
                               _i(nper), _i(prefix))
        tot = int(nper.sum())
        if self.n + tot > self.cap:
            raise MemoryError("oracle swarm capacity exceeded")
        excl = np.concatenate(([0], np.cumsum(nper)[:-1])).astype(np.int64)
        slot_base = np.ascontiguousarray(self.n + excl, dtype=np.int64)
        id_base = np.ascontiguousarray(self.next_id + excl, dtype=np.uint64)
        S = self._swarm_c()
        lib().orc_source_fill(C.byref(M), C.byref(self.P), C.byref(S), source_type, t_start, dt,
                              _i(prefix), slot_base.ctypes.data_as(C.POINTER(C.c_int64)),
                              id_base.ctypes.data_as(C.POINTER(C.c_uint64)))
        self.n += tot
        self.next_id += tot
        return tot

    def TransportPhotons(self, t_start: float, dt: float, first: int = 0,
                         last: Optional[int] = None) -> int:
        self._enter()
        M, S = self._mesh_c(), self._swarm_c()
        last = self.n if last is None else last
        ev = int(lib().orc_transport_photons(C.byref(M), C.byref(self.P), C.byref(S), t_start, dt,
                                             first, last))
        self.events += ev
        return ev

    def CheckCompletion(self, t_end: float) -> int:
        S = self._swarm_c()
        return int(lib().orc_check_completion(C.byref(S), t_end))

    def RemoveMarkedParticles(self) -> int:
        S = self._swarm_c()
        self.n = int(lib().orc_remove_marked(C.byref(S)))
        return self.n

    def EvaluateRadiationEnergy(self) -> None:
        M, S = self._mesh_c(), self._swarm_c()
        lib().orc_evaluate_radiation_energy(C.byref(M), C.byref(S))

    def UpdateFluid(self) -> None:
        M = self._mesh_c()
        lib().orc_update_fluid(C.byref(M), C.byref(self.P))

    def PhotonReflectBC(self, face: int) -> None:
        M, S = self._mesh_c(), self._swarm_c()
        lib().orc_photon_reflect_bc(C.byref(M), C.byref(S), face)

    def SampleDDMCBlockFace(self) -> None:
        self._enter()
        M, S = self._mesh_c(), self._swarm_c()
        lib().orc_sample_ddmc_block_face(C.byref(M), C.byref(self.P), C.byref(S))

    def InitializeRadiation(self, is_thermal: bool) -> None:
        """reference jaybenne.cpp:570-578"""
        if is_thermal:
            self.SourcePhotons(SRC_THERMAL, 0.0, 0.0, blocks_in_call=1)
        self.EvaluateRadiationEnergy()

    def RadiationStep(self, t_start: float, dt: float) -> None:
        """Task order of reference jaybenne.cpp:104-138."""
        self.cycle += 1
        self.UpdateDerivedTransportFields(dt)
        # emission_blocks_in_call: blocks in the calling rank's MeshData (quirk of
        # sourcing.cpp:68-69: the per-cell count scales with 1 / that number); None = whole mesh
        self.SourcePhotons(SRC_EMISSION, t_start, dt, getattr(self, "emission_blocks_in_call", None))
        self.TransportPhotons(t_start, dt)
        self.RemoveMarkedParticles()
        assert self.CheckCompletion(t_start + dt) == 0
        self.EvaluateRadiationEnergy()
        self.UpdateFluid()
