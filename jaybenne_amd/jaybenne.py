"""Host-side mirror of the reference's package interface (reference src/jaybenne/jaybenne.hpp:48-78)
over the C ABI of ``libjaybenne_amd.so``.

Same task names, same argument meaning (``md, t_start, dt``), same return convention
(``TaskStatus``) and the same failure conditions as the reference (PARTHENON_REQUIRE /
PARTHENON_FAIL become exceptions).  PyTorch appears only as the owner of device memory, the
stream and ``torch.distributed``; every field / particle update runs in the HIP library.
"""
from __future__ import annotations

import ctypes as C
import os
import time
import enum
import sys
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .deck import ParameterInput
from .mesh import Mesh

photons_swarm_name = "photons"           # jaybenne_variables.hpp:23


class TaskStatus(enum.IntEnum):
    complete = _lib.JB_COMPLETE
    iterate = _lib.JB_ITERATE
    incomplete = _lib.JB_INCOMPLETE


class SourceType(enum.IntEnum):           # jaybenne.hpp:56
    thermal = _lib.JB_SOURCE_THERMAL
    emission = _lib.JB_SOURCE_EMISSION


class SourceStrategy(enum.IntEnum):       # jaybenne.hpp:55
    uniform = _lib.JB_STRATEGY_UNIFORM
    energy = _lib.JB_STRATEGY_ENERGY


# ------------------------------------------------------------------------------------------------
class StateDescriptor:
    """What ``jaybenne::Initialize`` returns: the package's parameters (``Param``) plus, here,
    the library context that owns the device-side copies of them."""

    def __init__(self, params: Dict[str, object], ctx: C.c_void_p, device: torch.device):
        self._params = dict(params)
        self.ctx = ctx
        self.device = device
        self.lib = _lib.load()

    def Param(self, name: str):
        return self._params[name]

    # arithmetic of the gray IMC tracking step (include/jaybenne_amd.h): "lean" (default) or
    # "exact" (bit-identical to the CPU oracle's portable flavour)
    def set_arithmetic(self, mode: str) -> None:
        code = {"exact": _lib.ARITH_EXACT, "lean": _lib.ARITH_LEAN}[mode]
        _lib.check(self.lib.jb_set_arithmetic(self.ctx, code))

    def arithmetic(self) -> str:
        return "lean" if self.lib.jb_get_arithmetic(self.ctx) == _lib.ARITH_LEAN else "exact"

    def AllParams(self) -> Dict[str, object]:
        return dict(self._params)

    def close(self) -> None:
        if self.ctx is not None:
            self.lib.jb_finalize(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def Initialize(pin: ParameterInput, opacity, scattering, eos, device: Optional[torch.device] = None,
               my_rank: int = 0) -> StateDescriptor:
    """``jaybenne::Initialize(pin, opacity, scattering, eos)`` -- reference jaybenne.cpp:158-266:
    same keys, defaults and failure conditions."""
    lib = _lib.load()
    num_particles = pin.GetInteger("jaybenne", "num_particles")
    dt = pin.GetOrAddReal("jaybenne", "dt", sys.float_info.max)
    min_occ = pin.GetOrAddReal("jaybenne", "min_swarm_occupancy", 0.0)
    if not (0.0 <= min_occ < 1.0):
        raise ValueError("Minimum allowable swarm occupancy must be >= 0 and less than 1")
    numin = pin.GetOrAddReal("jaybenne", "numin", sys.float_info.min)
    numax = pin.GetOrAddReal("jaybenne", "numax", sys.float_info.max)
    units = opacity.GetRuntimePhysicalConstants()
    unique_rank_seeds = pin.GetOrAddBoolean("jaybenne", "unique_rank_seeds", True)
    seed = pin.GetOrAddInteger("jaybenne", "seed", 123)
    max_iter = pin.GetOrAddInteger("jaybenne", "max_transport_iterations", 10000)
    use_ddmc = pin.GetOrAddBoolean("jaybenne", "use_ddmc", False)
    tau_ddmc = pin.GetOrAddReal("jaybenne", "tau_ddmc", 5.0)
    strategy = pin.GetOrAddString("jaybenne", "source_strategy", "uniform")
    if strategy == "uniform":
        source_strategy = SourceStrategy.uniform
    elif strategy == "energy":
        source_strategy = SourceStrategy.energy
    else:
        raise ValueError("Only uniform or energy source strategies supported!")
    do_emission = pin.GetOrAddBoolean("jaybenne", "do_emission", True)
    do_feedback = pin.GetOrAddBoolean("jaybenne", "do_feedback", True)

    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    p = _lib.Params(num_particles=num_particles, dt=dt, min_swarm_occupancy=min_occ, numin=numin,
                    numax=numax, tau_ddmc=tau_ddmc, unique_rank_seeds=int(unique_rank_seeds),
                    seed=seed, max_transport_iterations=max_iter, use_ddmc=int(use_ddmc),
                    source_strategy=int(source_strategy), do_emission=int(do_emission),
                    do_feedback=int(do_feedback), rank=my_rank)
    e = _lib.Eos(model=eos.model, gm1=eos.gm1, cv=eos.cv)
    scales = {k: float(getattr(opacity, k, 1.0)) for k in
              ("time_scale", "mass_scale", "length_scale", "temperature_scale")}
    o = _lib.Opacity(model=opacity.model, kappa=opacity.kappa, c=units.c, sb=units.sb, **scales)
    scales_s = {k: float(getattr(scattering, k, 1.0)) for k in
                ("time_scale", "mass_scale", "length_scale", "temperature_scale")}
    s = _lib.Scattering(model=scattering.model, kappa_s=scattering.kappa_s, apm=scattering.apm,
                        **scales_s)
    ctx = C.c_void_p()
    _lib.check(lib.jb_initialize(C.byref(p), C.byref(e), C.byref(o), C.byref(s),
                                 device.index or 0, C.byref(ctx)))
    params = dict(num_particles=num_particles, dt=dt, min_swarm_occupancy=min_occ, numin=numin,
                  numax=numax, speed_of_light=units.c, stefan_boltzmann=units.sb,
                  unique_rank_seeds=unique_rank_seeds, seed=int(lib.jb_param_seed(ctx)),
                  max_transport_iterations=max_iter, use_ddmc=use_ddmc, tau_ddmc=tau_ddmc,
                  source_strategy=source_strategy, do_emission=do_emission,
                  do_feedback=do_feedback, eos_d=eos, opacity_d=opacity, scattering_d=scattering)
    return StateDescriptor(params, ctx, device)


# ------------------------------------------------------------------------------------------------
_F64 = _lib.SWARM_F64
_I32 = _lib.SWARM_I32


class MeshData:
    """The blocks of one rank with their fields and photon swarm in HBM (the role of
    Parthenon's ``MeshData<Real>`` + ``SwarmContainer`` for this package).

    Layout in HBM: each field is one ``[nblocks_local, nk, nj, ni]`` float64 tensor (ghosts
    included; the C ABI receives one device pointer per block); the swarm is a structure of
    arrays with ``capacity`` slots, particles ``0..n-1`` valid.
    """

    def __init__(self, pkg: StateDescriptor, mesh: Mesh, capacity: int, rank: int = 0,
                 nranks: int = 1, comm=None, halo_rings: int = 1, replicated: bool = False):
        self.pkg = pkg
        self.mesh = mesh
        self.rank, self.nranks = rank, nranks
        self.comm = comm
        self.lib = pkg.lib
        dev = pkg.device
        self.device = dev
        # Replicated mesh, split particles (the reference has no such mode, its blocks are
        # partitioned: jaybenne.cpp:92-95): every rank holds EVERY block and follows its share of each
        # block's photons -- dealt by stream id at the source (jb_source_photons_fill_range) -- from
        # source to census; nothing is ever handed over, and the ranks' energy_tally / energy_delta
        # are summed once per cycle (one all-reduce each).  For meshes whose fields fit every GPU
        # many times over (the SMR decks: 20 / 32 blocks of 32^2 cells) this balances whatever the
        # blocks cost -- a DDMC block of BASELINE configs[4] costs 1 % of an IMC block --, which a
        # partition of 32 blocks over 8 ranks cannot.
        self.replicated = bool(replicated) and nranks > 1
        if self.replicated:
            owner = np.full(mesh.nblocks, rank, dtype=np.int32)
        else:
            owner = np.ascontiguousarray(mesh.owner, dtype=np.int32)
        self.field_owner = owner           # who holds the authoritative copy of a block's fields (halo.py)
        if owner.max() >= nranks:
            raise ValueError("mesh.owner names a rank >= nranks")
        self.gids = np.nonzero(owner == rank)[0].astype(np.int32)       # the blocks this rank owns
        if len(self.gids) == 0:
            raise ValueError(f"rank {rank} owns no blocks")
        self.nowned = len(self.gids)
        # Halo copies: read-only mirrors of the neighbouring ranks' blocks that touch ours.  A
        # particle that wanders across the rank boundary keeps being tracked here and is handed
        # to its owner once, when its history ends, instead of at every crossing (in the
        # reference every crossing costs a transport iteration with a global sync,
        # jaybenne.cpp:113-131).  HBM is plentiful: one ring costs a few GB at most.
        halo = (mesh.neighbours(self.gids, halo_rings) if (nranks > 1 and halo_rings > 0 and not self.replicated)
                else np.zeros(0, dtype=np.int32))
        self.resident_gids = np.concatenate([self.gids, halo]).astype(np.int32)
        self.owned_flags = np.concatenate([np.ones(self.nowned, dtype=np.int32),
                                           np.zeros(len(halo), dtype=np.int32)])
        self.nblocks = len(self.resident_gids)
        local_index = np.full(mesh.nblocks, -1, dtype=np.int32)
        local_index[self.resident_gids] = np.arange(self.nblocks, dtype=np.int32)
        self.local_index = local_index
        shape = (self.nblocks,) + tuple(mesh.field_shape[1:])
        names = list(_lib.FIELD_NAMES)
        use_ddmc = bool(pkg.Param("use_ddmc"))
        self.fields: Dict[str, torch.Tensor] = {}
        for n in names:
            if n in ("P1", "P2", "P3") and not use_ddmc:
                continue
            self.fields[n] = torch.zeros(shape, dtype=torch.float64, device=dev)
        # swarm
        self.capacity = int(capacity)
        self.swarm: Dict[str, torch.Tensor] = {}
        for n in _F64:
            self.swarm[n] = torch.zeros(self.capacity, dtype=torch.float64, device=dev)
        for n in _I32:
            self.swarm[n] = torch.zeros(self.capacity, dtype=torch.int32, device=dev)
        self.swarm["id"] = torch.zeros(self.capacity, dtype=torch.int64, device=dev)   # uint64 bits
        self.swarm["rng"] = torch.zeros(self.capacity, dtype=torch.int64, device=dev)  # uint64 bits
        self.sv = _lib.SwarmView(n=0, capacity=self.capacity)
        for n in _F64 + _I32 + ("id", "rng"):
            setattr(self.sv, n, self.swarm[n].data_ptr())
        self.max_capacity: Optional[int] = None   # set to forbid growth beyond a slot count
        self.prefix = torch.zeros(self.nblocks * mesh.ncell, dtype=torch.int32, device=dev)
        self.records: Optional[torch.Tensor] = None
        self.next_id = 0     # first unused stream id (global, kept in step on every rank)
        self.cycle = 0       # RadiationStep counter (keys the per-cell rounding streams: source_epoch); 0 = initialisation
        self.events = 0
        self.kernel_events = None   # set to [] to time every transport launch with HIP events
        self._exchange = None       # halo.FieldExchange, built on first use
        self.phase_times = None     # set to {} to account wall time per phase of RadiationStep
        # hand-off accounting (bench.py): records this rank handed to others, wall time spent in
        # the exchange phase, transport iterations, since the caller last zeroed them
        self.handoff_records = 0
        self.exchange_seconds = 0.0        # hand-off proper: from the end of the transport launch on
        self.transport_wait_seconds = 0.0  # host time spent waiting for the transport launch
        self.collective_seconds = 0.0      # part of exchange_seconds inside the two collectives
        self.transport_iterations_total = 0
        self._stats_cache = None           # counters as of the end of the last RadiationStep
        # measurement aid (bench.py --force-exchange): run the hand-off phase of the iterate-sublist
        # also when this rank holds the whole mesh (nothing moves; its fixed cost becomes visible)
        self.force_exchange = False
        # How particles are handed to other ranks: "c" (default) = the library's jb_exchange in one call, over a
        # communicator of its own on an RCCL process group and over torch.distributed callbacks otherwise
        # ("c-rccl" / "c-torch" force one); "python" = the same protocol driven from here (comm.py: count
        # kernel, read-back, all-gather, all-to-all-v from Python), kept for A/B.  JB_HANDOFF overrides.
        self.handoff = os.environ.get("JB_HANDOFF", "c")
        if self.handoff not in ("c", "c-torch", "c-rccl", "python"):
            raise ValueError(f"JB_HANDOFF = {self.handoff!r}: one of c, c-torch, c-rccl, python")
        self._chandoff = None
        # DefragParticles after every k-th RadiationStep (0: never; the reference schedules none)
        # DefragParticles: -1 (default) on the library's schedule (jb_defrag_policy: when a cycle costs
        # 10 % more per event than the best one since the last sort), k > 0 after every k-th cycle,
        # 0 never (the reference: slot order stays what the task list makes it)
        self.defrag_interval = -1
        self.defrags = 0
        self._steps_since_defrag = 0
        self._make_mesh_handle(owner)

    def reserve(self, nslots: int) -> None:
        """Make room for ``nslots`` particles (the role of Parthenon's pool growth inside
        ``Swarm::AddEmptyParticles``, reference sourcing.cpp:123-131): the swarm arrays are
        reallocated at twice the need and the live prefix ``0..n-1`` is carried over."""
        if nslots <= self.capacity:
            return
        if self.max_capacity is not None and nslots > self.max_capacity:
            raise MemoryError(f"swarm capacity {self.max_capacity} too small for {nslots} particles")
        new_cap = 2 * int(nslots)
        if self.max_capacity is not None:
            new_cap = min(new_cap, int(self.max_capacity))
        self._sync_stream()
        n = int(self.sv.n)
        for name in _F64 + _I32 + ("id", "rng"):
            old = self.swarm[name]
            new = torch.zeros(new_cap, dtype=old.dtype, device=old.device)
            new[:n].copy_(old[:n])
            self.swarm[name] = new
            setattr(self.sv, name, new.data_ptr())
        torch.cuda.synchronize(self.device)
        self.capacity = new_cap
        self.sv.capacity = new_cap

    # ---- C views
    def _make_mesh_handle(self, owner: np.ndarray) -> None:
        m = self.mesh
        g = self.resident_gids
        keep = dict(owned=self.owned_flags,
            leaf_map=np.ascontiguousarray(m.leaf_map, dtype=np.int32), owner=owner,
            local_index=self.local_index, gid=np.ascontiguousarray(g, dtype=np.int32),
            blk_xmin=np.ascontiguousarray(m.blk_xmin[g]), blk_xmax=np.ascontiguousarray(m.blk_xmax[g]),
            blk_dx=np.ascontiguousarray(m.blk_dx[g]),
            blk_level=np.ascontiguousarray(m.blk_level[g], dtype=np.int32),
            blk_nbr_lev=np.ascontiguousarray(m.blk_nbr_lev[g], dtype=np.int32))
        v = _lib.MeshView(ndim=m.ndim, ng=m.ng, nblocks=self.nblocks, nblocks_total=m.nblocks,
                          rank=self.rank)
        v.nx = (C.c_int32 * 3)(*m.nx)
        v.nleaf = (C.c_int32 * 3)(*m.nleaf)
        v.bc = (C.c_int32 * 6)(*m.swarm_bc)
        v.gmin = (C.c_double * 3)(*m.gmin)
        v.gmax = (C.c_double * 3)(*m.gmax)
        for k, a in keep.items():
            setattr(v, k, a.ctypes.data)
        ptr_tables = {}
        for n, t in self.fields.items():
            stride = t.stride(0) * t.element_size()
            tab = np.array([t.data_ptr() + b * stride for b in range(self.nblocks)], dtype=np.uint64)
            ptr_tables[n] = tab
            setattr(v, n, tab.ctypes.data)
        handle = C.c_void_p()
        self._sync_stream()
        _lib.check(self.lib.jb_mesh_create(self.pkg.ctx, C.byref(v), C.byref(handle)))
        self.handle = handle

    @property
    def exchange(self):
        """Ghost-zone / halo refresh plan for the host fields (built once per mesh)."""
        if self._exchange is None:
            from .halo import FieldExchange
            self._exchange = FieldExchange(self)
        return self._exchange

    def _sync_stream(self) -> None:
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self.lib.jb_set_stream(self.pkg.ctx, C.c_void_p(stream))

    @property
    def n(self) -> int:
        return int(self.sv.n)

    def ensure_handoff(self) -> None:
        """Makes the C hand-off's transport now (collective over the ranks: jaybenne_amd/handoff.py) instead of
        inside the first exchange."""
        if self.handoff != "python" and self._chandoff is None and self.comm is not None:
            from .handoff import CHandoff
            self._chandoff = CHandoff.make(self, {"c": "auto", "c-torch": "torch", "c-rccl": "rccl"}[self.handoff])

    def handoff_path(self) -> str:
        """What the hand-off of this MeshData runs through (bench.py: ``handoff.path``)."""
        if self.handoff == "python":
            return "python: comm.py (count kernel, read-back, all-gather and all-to-all-v driven from Python)"
        return self._chandoff.path if self._chandoff is not None else "c: jb_exchange (no exchange has run yet)"

    def close(self) -> None:
        if getattr(self, "_chandoff", None) is not None:
            self._chandoff.close()
            self._chandoff = None
        if getattr(self, "handle", None) is not None:
            self.lib.jb_mesh_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host <-> device helpers for the harness
    def set_field(self, name: str, host: np.ndarray, local: bool = False) -> None:
        """host: [nblocks_total, nk, nj, ni] (whole mesh; the resident blocks are picked out) or,
        with ``local=True``, [len(resident_gids), nk, nj, ni]."""
        src = host if local else host[self.resident_gids]
        self.fields[name].copy_(torch.from_numpy(np.ascontiguousarray(src)))

    def get_field(self, name: str) -> np.ndarray:
        """The owned blocks' part of a field, in the order of ``gids``."""
        return self.fields[name][:self.nowned].cpu().numpy()

    def get_swarm(self) -> Dict[str, np.ndarray]:
        n = self.n
        out = {k: v[:n].cpu().numpy() for k, v in self.swarm.items()}
        out["id"] = out["id"].view(np.uint64)
        out["rng"] = out["rng"].view(np.uint64)
        return out

    def stats(self, reset: bool = False) -> Dict[str, int]:
        st = _lib.TransportStats()
        _lib.check(self.lib.jb_get_transport_stats(self.pkg.ctx, C.byref(st), int(reset)))
        self._stats_cache = None
        return {k: int(getattr(st, k)) for k, _ in st._fields_}


# ------------------------------------------------------------------------------------------------
# tasks (reference jaybenne.hpp:59-76)
def UpdateDerivedTransportFields(md: MeshData, dt: float) -> TaskStatus:
    md._sync_stream()
    _lib.check(md.lib.jb_update_derived_transport_fields(md.pkg.ctx, md.handle, dt))
    return TaskStatus.complete


def _global_block_counts(md: MeshData, nper_local: np.ndarray) -> np.ndarray:
    counts = np.zeros(md.mesh.nblocks, dtype=np.int64)
    counts[md.resident_gids] = nper_local        # halo copies source nothing (zeros)
    if md.comm is not None and md.nranks > 1 and not md.replicated:
        counts = md.comm.allreduce_sum_int64(counts)
    return counts     # (replicated mesh: every rank has counted every block itself)


def rank_share(nper: np.ndarray, rank: int, nranks: int):
    """Which of a block's ``nper[b]`` new photons (numbered in cell order, as their stream ids are)
    rank ``rank`` of a replicated-mesh run sources: a contiguous range per block, ``first[b]`` ..
    ``first[b] + count[b]`` -- contiguous cells, so a rank's photons start out as compact in space as
    the whole swarm's do."""
    n = nper.astype(np.int64)
    first = (n * rank) // nranks
    return first.astype(np.int32), ((n * (rank + 1)) // nranks - first).astype(np.int32)


def _sum_over_ranks(md: MeshData, names) -> None:
    """Replicated mesh: the ranks' partial cell fields -> their sum, on every rank."""
    if md.replicated:
        for n in names:
            md.comm.allreduce_sum_tensor(md.fields[n])


def source_epoch(cycle: int, source_type) -> int:
    """Key of the per-cell rounding streams of a source call (include/jaybenne_amd.hpp: SourceEpoch,
    shared by every host): 0 for the initial thermal source, k for the emission source of cycle k."""
    if int(source_type) == int(SourceType.emission):
        return int(cycle)
    return 0 if cycle == 0 else (1 << 19) | int(cycle)


def SourcePhotons(md: MeshData, source_type: SourceType, t_start: float, dt: float,
                  per_block: bool = False) -> TaskStatus:
    """``SourcePhotons<T, ST>(md, t_start, dt)`` -- reference sourcing.cpp:25-208.
    ``per_block=True`` is the ``MeshBlockData`` instantiation used at initialisation (one call
    per block, so the ``nblocks`` of sourcing.cpp:68-69 is 1)."""
    pkg = md.pkg
    if pkg.Param("source_strategy") == SourceStrategy.energy:
        raise NotImplementedError("Energy source strategy not implemented!")
    if source_type == SourceType.emission and not pkg.Param("do_emission"):
        return TaskStatus.complete
    md._sync_stream()
    nper = np.zeros(md.nblocks, dtype=np.int32)
    blocks_in_call = 1 if per_block else md.nowned   # vmesh.GetNBlocks() of this rank's MeshData
    _lib.check(md.lib.jb_source_photons_count(pkg.ctx, md.handle, int(source_type), dt,
                                              blocks_in_call, source_epoch(md.cycle, source_type),
                                              nper.ctypes.data, md.prefix.data_ptr()))
    counts = _global_block_counts(md, nper)
    excl = np.concatenate(([0], np.cumsum(counts)[:-1]))
    id_base = np.ascontiguousarray(md.next_id + excl[md.resident_gids], dtype=np.uint64)
    if md.replicated:    # this rank's share of every block (stream ids stay those of the whole block)
        first, nper = rank_share(nper, md.rank, md.nranks)
    local_excl = np.concatenate(([0], np.cumsum(nper.astype(np.int64))[:-1]))
    slot_base = np.ascontiguousarray(md.n + local_excl, dtype=np.int64)
    tot = int(nper.sum())
    md.reserve(md.n + tot)
    if md.replicated:
        _lib.check(md.lib.jb_source_photons_fill_range(pkg.ctx, md.handle, C.byref(md.sv), int(source_type),
                                                       t_start, dt, nper.ctypes.data, md.prefix.data_ptr(),
                                                       slot_base.ctypes.data, id_base.ctypes.data,
                                                       first.ctypes.data, int(md.rank == 0)))
    else:
        _lib.check(md.lib.jb_source_photons_fill(pkg.ctx, md.handle, C.byref(md.sv), int(source_type),
                                                 t_start, dt, nper.ctypes.data, md.prefix.data_ptr(),
                                                 slot_base.ctypes.data, id_base.ctypes.data))
    md.sv.n += tot
    md.next_id += int(counts.sum())
    return TaskStatus.complete


def _transport(md: MeshData, t_start: float, dt: float, first: int, last: Optional[int],
               fuse_census_tally: bool, ddmc: bool) -> TaskStatus:
    md._sync_stream()
    md._stats_cache = None       # (a transport launch outside RadiationStep moves the counters)
    last = md.n if last is None else last
    fn = md.lib.jb_transport_photons_ddmc if ddmc else md.lib.jb_transport_photons
    timed = md.kernel_events is not None and last > first
    if timed:   # HIP events on the stream the kernel is launched on (bench.py roofline)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record(torch.cuda.current_stream(md.device))
    _lib.check(fn(md.pkg.ctx, md.handle, C.byref(md.sv), t_start, dt, first, last,
                  int(fuse_census_tally)))
    if timed:
        ev1.record(torch.cuda.current_stream(md.device))
        md.kernel_events.append((ev0, ev1, last - first))
    return TaskStatus.complete


def TransportPhotons(md: MeshData, t_start: float, dt: float, first: int = 0,
                     last: Optional[int] = None, fuse_census_tally: bool = False) -> TaskStatus:
    """reference transport.cpp:28-181"""
    return _transport(md, t_start, dt, first, last, fuse_census_tally, False)


def TransportPhotons_DDMC(md: MeshData, t_start: float, dt: float, first: int = 0,
                          last: Optional[int] = None, fuse_census_tally: bool = False) -> TaskStatus:
    """reference transport_ddmc.cpp:28-237"""
    return _transport(md, t_start, dt, first, last, fuse_census_tally, True)


def SampleDDMCBlockFace(md: MeshData, first: int = 0, last: Optional[int] = None) -> TaskStatus:
    """reference sample_ddmc_bface.cpp:81-427"""
    md._sync_stream()
    last = md.n if last is None else last
    _lib.check(md.lib.jb_sample_ddmc_block_face(md.pkg.ctx, md.handle, C.byref(md.sv), first, last))
    return TaskStatus.complete


def CheckCompletion(md: MeshData, t_end: float) -> TaskStatus:
    """reference transport.cpp:187-216 (local part; the global_sync of jaybenne.cpp:130-131 is
    the all-reduce in RadiationStep)"""
    md._sync_stream()
    unfinished = C.c_int64(0)
    st = _lib.check(md.lib.jb_check_completion(md.pkg.ctx, C.byref(md.sv), t_end,
                                               C.byref(unfinished)))
    md.num_unfinished = int(unfinished.value)
    return TaskStatus(st)


def EvaluateRadiationEnergy(md: MeshData) -> TaskStatus:
    """reference jaybenne.cpp:514-564"""
    md._sync_stream()
    _lib.check(md.lib.jb_evaluate_radiation_energy(md.pkg.ctx, md.handle, C.byref(md.sv)))
    _sum_over_ranks(md, ("tally",))
    return TaskStatus.complete


def UpdateFluid(md: MeshData) -> TaskStatus:
    """reference jaybenne.cpp:583-615"""
    md._sync_stream()
    _lib.check(md.lib.jb_update_fluid(md.pkg.ctx, md.handle))
    return TaskStatus.complete


def PhotonReflectBC(md: MeshData, face: int) -> None:
    """reference boundaries.hpp:24-84; face 0..5 = inner_x1, outer_x1, inner_x2, ..."""
    md._sync_stream()
    _lib.check(md.lib.jb_photon_reflect_bc(md.pkg.ctx, md.handle, C.byref(md.sv), face))


def RemoveMarkedParticles(md: MeshData) -> int:
    md._sync_stream()
    _lib.check(md.lib.jb_remove_marked_particles(md.pkg.ctx, C.byref(md.sv)))
    return md.n


def DefragParticles(md: MeshData) -> TaskStatus:
    """reference jaybenne.cpp:499-509 (``Swarm::Defrag``; scheduled by no task list of the
    reference).  The swarm here is always compact after RemoveMarkedParticles; what this task
    restores is its *order*: the photons sorted by (block, cell), as they are sourced -- the order
    that keeps the cell data a wave gathers in the L2 of its XCD, and that diffusion loosens from
    cycle to cycle (include/jaybenne_amd.h: ``jb_defrag_particles``; in place).
    ``MeshData.defrag_interval = k`` makes RadiationStep schedule it after every k-th cycle."""
    if md.n == 0:
        return TaskStatus.complete
    md._sync_stream()
    _lib.check(md.lib.jb_defrag_particles(md.pkg.ctx, md.handle, C.byref(md.sv)))
    md.defrags += 1
    return TaskStatus.complete


def EstimateTimestepMesh(md: MeshData) -> float:
    """reference jaybenne.cpp:271-275"""
    return float(md.lib.jb_estimate_timestep(md.pkg.ctx))


def InitializeRadiation(md: MeshData, is_thermal: bool) -> None:
    """reference jaybenne.cpp:570-578 (called per block by mcblock's ProblemGenerator)"""
    if is_thermal:
        SourcePhotons(md, SourceType.thermal, 0.0, 0.0, per_block=True)
    EvaluateRadiationEnergy(md)


# ------------------------------------------------------------------------------------------------
def _exchange(md: MeshData, first: int, last: int):
    """MeshResetCommunication -> MeshSend -> MeshReceive (reference jaybenne.cpp:26-61) for the
    particles among [first, last) whose destination block lives on another rank.  Returns
    (number received, number handed over anywhere on the node).

    Per call: one count kernel + read-back, one all-gather of the rank x rank count matrix (which
    also answers the completion question), and -- only if anything moved -- one pack kernel, one
    all-to-all-v of 104-byte records and one unpack kernel.  Departed particles stay behind as
    holes until the swarm is compacted."""
    lib, ctx = md.lib, md.pkg.ctx
    if md.handoff != "python":
        # the library's one C call (jb_exchange; jaybenne_amd/handoff.py): the default
        md.ensure_handoff()
        t_c = time.perf_counter()
        out = md._chandoff.exchange(first, last)
        md.collective_seconds += time.perf_counter() - t_c   # (the whole call: the collectives are inside it)
        return out
    md._sync_stream()
    counts = np.zeros(md.nranks, dtype=np.int64)
    if md.records is None:
        md.records = torch.empty((max(4096, (last - first) // 16), _lib.JB_RECORD_WORDS),
                                 dtype=torch.int64, device=md.device)
    st = lib.jb_pack_outgoing(ctx, md.handle, C.byref(md.sv), first, last, md.nranks,
                              md.records.data_ptr(), md.records.shape[0], counts.ctypes.data)
    if st == _lib.JB_ERR_CAPACITY:      # counts are filled in before the capacity check: grow once
        md.records = torch.empty((int(counts.sum() * 1.5) + 4096, _lib.JB_RECORD_WORDS),
                                 dtype=torch.int64, device=md.device)
        st = lib.jb_pack_outgoing(ctx, md.handle, C.byref(md.sv), first, last, md.nranks,
                                  md.records.data_ptr(), md.records.shape[0], counts.ctypes.data)
    _lib.check(st)
    t_c = time.perf_counter()
    matrix = md.comm.gather_count_matrix(counts)
    md.collective_seconds += time.perf_counter() - t_c
    total = int(matrix.sum())
    if total == 0:
        return 0, 0
    nsend = int(counts.sum())
    md.handoff_records += nsend
    t_c = time.perf_counter()
    recv = md.comm.exchange_records(md.records[:nsend] if nsend else None, counts, md.device,
                                    recv_counts=matrix[:, md.rank].copy())
    md.collective_seconds += time.perf_counter() - t_c
    nrecv = 0 if recv is None else int(recv.shape[0])
    if nrecv:
        if md.n + nrecv > md.capacity:
            RemoveMarkedParticles(md)          # close the holes left by earlier departures
        md.reserve(md.n + nrecv)
        md._sync_stream()
        _lib.check(lib.jb_unpack_incoming(ctx, md.handle, C.byref(md.sv), recv.data_ptr(), nrecv))
    return nrecv, total


class _Range:
    """A trace range around a part of the task list (``jb_range_push`` / ``jb_range_pop``: ROCTx, shown by
    ``rocprofv3 --marker-trace``) -- the reference's ``Kokkos::Profiling::pushRegion("Jaybenne::Timestep")``
    and ``("Jaybenne::TransportLoop")``, jaybenne.cpp:87,115,127,145.  Does nothing without the marker library."""

    def __init__(self, md, name: str):
        self.lib, self.name = md.lib, name.encode()

    def __enter__(self):
        self.lib.jb_range_push(self.name)

    def __exit__(self, *exc):
        self.lib.jb_range_pop()
        return False


class _Phase:
    """Optional wall-clock accounting of the phases of RadiationStep (``md.phase_times = {}`` to
    switch it on; each phase then ends with a device synchronisation, so leave it off when
    measuring throughput)."""

    def __init__(self, md, name):
        self.md, self.name = md, name

    def __enter__(self):
        if self.md.phase_times is not None:
            torch.cuda.synchronize(self.md.device)
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if self.md.phase_times is not None:
            torch.cuda.synchronize(self.md.device)
            self.md.phase_times[self.name] = (self.md.phase_times.get(self.name, 0.0)
                                              + time.perf_counter() - self.t0)
        return False


def RadiationStep(md: MeshData, t_start: float, dt: float) -> TaskStatus:
    """One radiation cycle: the task list of ``jaybenne::RadiationStep`` (reference
    jaybenne.cpp:68-151) for this rank's blocks.

    Single rank: one transport launch resolves every block crossing in flight.  Several ranks:
    the iterate-sublist of jaybenne.cpp:113-131 -- transport, hand-off of the particles that left
    for another rank's blocks, SampleDDMCBlockFace on the arrivals, and the global completion
    test (one all-reduced integer) -- repeated until no particle is in flight anywhere.
    """
    with _Range(md, "Jaybenne::Timestep"):          # jaybenne.cpp:87 ... :145
        return _radiation_step(md, t_start, dt)


def _transport_loop(md: MeshData, transport, use_ddmc: bool, t_start: float, dt: float) -> bool:
    """The iterate-sublist of jaybenne.cpp:113-131; False when max_transport_iterations passes did
    not finish it (TaskStatus.iterate)."""
    pkg = md.pkg
    first = 0
    md.transport_iterations = 0
    for it in range(int(pkg.Param("max_transport_iterations"))):
        last = md.n
        with _Phase(md, f"transport[{min(it, 2)}]"):
            transport(md, t_start, dt, first, last, fuse_census_tally=True)
        md.transport_iterations += 1
        md.transport_iterations_total += 1
        if (md.nranks == 1 or md.replicated) and not (md.force_exchange and md.comm is not None):
            return True
        with _Phase(md, "exchange"), _Range(md, "Jaybenne::MeshSendReceive"):
            # the hand-off clock starts when the transport launch has finished (its first device
            # read-back would otherwise be charged with the whole kernel)
            t_w = time.perf_counter()
            done = torch.cuda.Event()
            done.record(torch.cuda.current_stream(md.device))
            done.synchronize()
            t_x = time.perf_counter()
            md.transport_wait_seconds += t_x - t_w
            nrecv, moved = _exchange(md, first, last)
            md.exchange_seconds += time.perf_counter() - t_x
        if moved == 0:
            return True
        first = md.n - nrecv        # the arrivals, appended at the end of the swarm
        if use_ddmc and nrecv:
            with _Phase(md, "block_face"):
                SampleDDMCBlockFace(md, first, md.n)
    return False


def _radiation_step(md: MeshData, t_start: float, dt: float) -> TaskStatus:
    pkg = md.pkg
    use_ddmc = bool(pkg.Param("use_ddmc"))
    transport = TransportPhotons_DDMC if use_ddmc else TransportPhotons
    md.cycle += 1
    with _Phase(md, "derived+source"):
        UpdateDerivedTransportFields(md, dt)
        SourcePhotons(md, SourceType.emission, t_start, dt)
        if md.replicated and md.rank != 0 and not pkg.Param("do_emission"):
            # Replicated mesh without the emission source: nothing resets energy_delta (sourcing.cpp:41-43
            # returns before :165-166 -- SURVEY App. C quirk 5), so behind last cycle's all-reduce EVERY rank
            # holds the global sum of all cycles so far.  One rank keeps carrying it; the others start the
            # cycle at zero and contribute their increment only -- the sum over ranks is then "everything so
            # far + this cycle's absorptions" once, not nranks times.
            md.fields["edelta"].zero_()
        # (the ddmc_face_prob ghost exchange of jaybenne.cpp:108-110 has no consumer: every face
        # the transport and resampling kernels read belongs to the block itself)
        md._sync_stream()
        _lib.check(md.lib.jb_zero_energy_tally(pkg.ctx, md.handle))
        # (the counters are cumulative: the previous step's closing read is this step's opening one,
        # one device synchronisation per step instead of two)
        before = md._stats_cache if md._stats_cache is not None else md.stats()
    with _Range(md, "Jaybenne::TransportLoop"):     # jaybenne.cpp:115 ... :127
        if not _transport_loop(md, transport, use_ddmc, t_start, dt):
            return TaskStatus.iterate
    with _Phase(md, "compaction+fluid"):
        after = md.stats()
        md._stats_cache = after
        if (md.nranks == 1 or md.replicated) and after["n_outgoing"] != before["n_outgoing"]:
            raise RuntimeError("particles left for another rank in a step that holds the whole mesh")
        # replicated mesh: the one exchange of the cycle -- census tally and absorbed / emitted energy
        # of all ranks' photons, summed on every rank (UpdateFluid then does the same on all of them)
        _sum_over_ranks(md, ("tally", "edelta"))
        if any(after[k] != before[k] for k in ("n_absorbed", "n_escaped", "n_outgoing")):
            RemoveMarkedParticles(md)
        md.events += after["n_events"] - before["n_events"]
        UpdateFluid(md)
        md._steps_since_defrag += 1
        if md.defrag_interval < 0:
            sorted_ = C.c_int32(0)
            events = int(after["n_events"] - before["n_events"])
            if md.nranks == 1:
                _lib.check(md.lib.jb_defrag_policy(pkg.ctx, md.handle, C.byref(md.sv), events, 0, C.byref(sorted_)))
            else:
                # the ranks sort together: a cycle is as long as its slowest rank (jaybenne_amd.h)
                _lib.check(md.lib.jb_defrag_policy(pkg.ctx, md.handle, C.byref(md.sv), events, 1, C.byref(sorted_)))
                if md.comm.allreduce_max_float(float(sorted_.value)) > 0.0:
                    _lib.check(md.lib.jb_defrag_policy(pkg.ctx, md.handle, C.byref(md.sv), events, 2, C.byref(sorted_)))
                else:
                    sorted_.value = 0
            if sorted_.value:
                md.defrags += 1
                md._steps_since_defrag = 0
        elif md.defrag_interval > 0 and md._steps_since_defrag >= md.defrag_interval:
            DefragParticles(md)
            md._steps_since_defrag = 0
    return TaskStatus.complete
