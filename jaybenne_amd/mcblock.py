"""Host material package: the part of the reference's ``mcblock`` application that the history
loop needs around it (reference src/mcblock/mcblock.cpp).

* ``Initialize``            parameters, EOS / opacity / scattering models   mcblock.cpp:37-150
* ``ProblemGenerator``      rho = rho0, sie = cv T0, stepdiff step at x >= 0  mcblock.cpp:155-203
* ``PostInitialization``    u = rho sie                                       mcblock.cpp:237-262
* ``UpdateDerived``         sie = u / rho over the entire block               mcblock.cpp:208-232

Everything here is setup-time host code on numpy arrays; the per-cycle work is in
``jaybenne_amd.jaybenne`` (HIP).
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict

import numpy as np

from . import constants
from .deck import ParameterInput
from .mesh import Mesh

EOS_IDEAL_GAS = 0
OPAC_GRAY = 0
OPAC_EPBREMSS = 1
SCAT_GRAY = 0
SCAT_THOMSON = 1


@dataclass
class EOS:
    """singularity::IdealGas(gm1, cv): T = sie / cv (reference mcblock.cpp:78-82)."""
    gm1: float
    cv: float
    model: int = EOS_IDEAL_GAS

    def TemperatureFromDensityInternalEnergy(self, rho, sie):
        return np.maximum(sie / self.cv, 0.0)

    def SpecificHeatFromDensityInternalEnergy(self, rho, sie):
        return self.cv


@dataclass
class Opacity:
    """``NonCGSUnits<Gray | EPBremss>(opac, time, mass, length, temperature)`` (reference
    mcblock.cpp:95-121, opacity.hpp:23-25): ``kappa`` is the Gray opacity in cm^2/g; the scales
    are the deck's code -> CGS conversion factors (mcblock.cpp:84-91).  What the library receives
    is in code units: kappa mass / length^2, and the constants of GetRuntimePhysicalConstants."""
    kappa: float
    model: int = OPAC_GRAY
    time_scale: float = 1.0
    mass_scale: float = 1.0
    length_scale: float = 1.0
    temperature_scale: float = 1.0

    def __post_init__(self):
        # sigma_code = rho_code (mass / length^3) kappa length
        self.kappa = self.kappa * (self.mass_scale / (self.length_scale * self.length_scale))

    @property
    def c(self) -> float:
        return constants.SPEED_OF_LIGHT * (self.time_scale / self.length_scale)

    @property
    def sb(self) -> float:   # erg cm^-2 s^-1 K^-4 = g s^-3 K^-4
        # (products written out, in the order examples/mcblock_amd.cpp uses: `**` goes through pow(),
        # whose last bit may differ from the repeated product for non-unit scales)
        ts, tk = self.time_scale, self.temperature_scale
        return constants.STEFAN_BOLTZMANN * (ts * ts * ts * (tk * tk) * (tk * tk) / self.mass_scale)

    def GetRuntimePhysicalConstants(self):
        return self


@dataclass
class Scattering:
    """``NonCGSUnitsS<GrayS | ThomsonS>`` (reference mcblock.cpp:126-145, opacity.hpp:28-30):
    GrayS(kappa_s, apm) has sigma_s = (rho / apm) kappa_s; ThomsonS(apm) has kappa_s = sigma_T.
    ``apm`` is passed on as the deck gives it (mcblock.cpp:124: "code units")."""
    kappa_s: float
    apm: float
    model: int = SCAT_GRAY
    time_scale: float = 1.0
    mass_scale: float = 1.0
    length_scale: float = 1.0
    temperature_scale: float = 1.0

    def __post_init__(self):
        self.kappa_s = self.kappa_s * (self.mass_scale / (self.length_scale * self.length_scale))


@dataclass
class McblockPackage:
    problem_id: str
    initial_temperature: float
    initial_density: float
    initial_radiation: str
    eos: EOS
    opacity: Opacity
    scattering: Scattering


def Initialize(pin: ParameterInput) -> McblockPackage:
    """reference mcblock.cpp:37-150"""
    if pin.GetString("parthenon/time", "integrator") != "rk1":
        raise ValueError("McBlock driver only supports first order time integration")
    problem_id = pin.GetString("parthenon/job", "problem_id")
    t0 = pin.GetReal("mcblock", "initial_temperature")
    rho0 = pin.GetReal("mcblock", "initial_density")
    initial_radiation = pin.GetString("mcblock", "initial_radiation")
    if initial_radiation not in ("none", "thermal"):
        raise ValueError("Only none or thermal initial radiation supported!")
    gamma = pin.GetOrAddReal("mcblock", "gamma", 1.66666666667)
    cv = pin.GetOrAddReal("mcblock", "cv", 1.0 / (gamma - 1.0))
    eos = EOS(gamma - 1.0, cv)
    scales = {key: pin.GetOrAddReal("mcblock", key, 1.0)   # code -> CGS, mcblock.cpp:84-91
              for key in ("time_scale", "mass_scale", "length_scale", "temperature_scale")}
    name = pin.GetString("mcblock", "opacity_model")
    if name == "none":
        opacity = Opacity(0.0, **scales)
    elif name == "constant":
        opacity = Opacity(pin.GetReal("mcblock", "opacity_constant_value"), **scales)
    elif name == "ep_bremss":
        opacity = Opacity(0.0, model=OPAC_EPBREMSS, **scales)
    else:
        raise ValueError("Only none or constant opacity models supported!")
    apm = pin.GetOrAddReal("mcblock", "apm", 1.0)
    sname = pin.GetOrAddString("mcblock", "scattering_model", "none")
    if sname == "none":
        scattering = Scattering(0.0, apm, **scales)
    elif sname == "constant":
        scattering = Scattering(pin.GetReal("mcblock", "scattering_constant_value"), apm, **scales)
    else:   # (the reference's variant also holds ThomsonS, but its host cannot select it)
        raise ValueError("Only none or constant scattering models supported!")
    return McblockPackage(problem_id, t0, rho0, initial_radiation, eos, opacity, scattering)


def ProblemGenerator(mesh: Mesh, pkg: McblockPackage, gids=None,
                     analytic_ghosts: bool = True) -> Dict[str, np.ndarray]:
    """Initial material state on the host for the blocks ``gids`` (default: all): returns rho,
    sie, u as ``[len(gids), nk, nj, ni]`` arrays with ghost zones filled (reference
    mcblock.cpp:155-203, 237-262 followed by the ghost exchange + FillDerived that Parthenon runs
    at the end of initialisation).

    The stepdiff state is a function of x1 alone with its step on a block boundary, so the ghost
    exchange has a closed form: evaluate the same function at the ghost-cell centre, clamped
    into the domain in x1 (outflow copies the edge cell), periodic in x2/x3.
    ``analytic_ghosts=False`` runs the general by-position exchange of ``Mesh.fill_ghosts``
    instead (whole mesh only); tests check that both agree."""
    cv = pkg.eos.SpecificHeatFromDensityInternalEnergy(pkg.initial_density, 1.0)
    if gids is None:
        gids = np.arange(mesh.nblocks)
    gids = np.asarray(gids)
    shape = (len(gids),) + tuple(mesh.field_shape[1:])
    rho = np.full(shape, pkg.initial_density, dtype=np.float64)
    sie = np.full(shape, cv * pkg.initial_temperature, dtype=np.float64)
    stepdiff = pkg.problem_id == "stepdiff"
    if stepdiff:
        ttlow = 1.0e-5 * pkg.initial_temperature
        for n, b in enumerate(gids):
            x1v = mesh.cell_centers(int(b), 0)
            if analytic_ghosts:
                half = 0.5 * mesh.blk_dx[int(b), 0]
                x1v = np.clip(x1v, mesh.gmin[0] + half, mesh.gmax[0] - half)
            sie[n][:, :, x1v >= 0.0] = cv * ttlow
    u = rho * sie                       # PostInitialization
    if not analytic_ghosts:
        if len(gids) != mesh.nblocks:
            raise ValueError("the by-position ghost exchange needs the whole mesh")
        mesh.fill_ghosts(rho)
        mesh.fill_ghosts(u)
    sie = UpdateDerived(rho, u)
    return {"rho": rho, "sie": sie, "u": u}


def UpdateDerived(rho: np.ndarray, u: np.ndarray) -> np.ndarray:
    """sie = u / rho over the entire block incl. ghosts (reference mcblock.cpp:208-232)."""
    return u / rho


# ------------------------------------------------------------------------------------------------
def block_costs(mesh: Mesh, pin: ParameterInput, pkg: McblockPackage) -> np.ndarray:
    """Tracking work per meshblock for one cycle, in events: what ``Mesh.partition`` balances.

    photons in the block x events per history there.  With the ``uniform`` source strategy every cell
    sources the same number of photons whatever its temperature (reference sourcing.cpp:68-69,99-101:
    ``npc = N / cells per block / nbtotal``; a cold cell's photons carry tiny weights but cost a
    history each), so every block starts with N / nbtotal of them -- the stepdiff decks' hot half
    holds the ENERGY, not the photons.  What differs between blocks is the length of a history:

    * an IMC block (transport.cpp:98-171): collisions + face crossings along a path of c dt,
      ``c dt (sigma_s + sigma_a + sum_d <|Omega_d|> / dx_d)``, ``<|Omega_d|> = 1/2``
      (BASELINE configs[1]: 1000 + 3 * 128 = 1384; measured 1388.5);
    * a DDMC block (``dx_push (sigma_a + sigma_s) > tau_ddmc``, transport_ddmc.cpp:135): one event per
      leak or absorption, ``c dt (f sigma_a + sum_faces P / dx)`` with ``P = 2 / (3 (tau_l + tau_u))``
      (jaybenne.cpp:381) -- a few dozen.
    Material state as the problem generator sets it (rho0; the opacities of the decks are constants)."""
    c = pkg.opacity.c
    dt = pin.GetReal("jaybenne", "dt")
    rho = pkg.initial_density
    sig_a = rho * pkg.opacity.kappa
    sig_s = (rho / pkg.scattering.apm) * pkg.scattering.kappa_s
    use_ddmc = pin.GetOrAddBoolean("jaybenne", "use_ddmc", False)
    tau_ddmc = pin.GetOrAddReal("jaybenne", "tau_ddmc", 5.0)
    cost = np.empty(mesh.nblocks)
    for b in range(mesh.nblocks):
        dx = mesh.blk_dx[b, :mesh.ndim]
        if use_ddmc and dx.min() * (sig_a + sig_s) > tau_ddmc:
            leak = sum(2.0 * (2.0 / (3.0 * 2.0 * (sig_a + sig_s) * d)) / d for d in dx)
            per_history = c * dt * (sig_a + leak)
        else:
            per_history = c * dt * (sig_s + sig_a + sum(0.5 / d for d in dx))
        cost[b] = max(per_history, 1.0)
    return cost


def choose_decomposition(mesh: Mesh, cost: np.ndarray, nranks: int, device=None) -> str:
    if nranks <= 1:
        return "blocks"
    field_bytes = 12 * 8 * mesh.nblocks * mesh.ntot          # every cell field of every block
    try:
        import torch
        total = torch.cuda.get_device_properties(device).total_memory
    except Exception:
        total = 288 << 30
    small = field_bytes * 64 <= total
    owner = mesh.partition(nranks, cost=cost) if mesh.nblocks >= nranks else None
    if owner is None:
        return "replicated" if small else "blocks"
    load = np.bincount(owner, weights=cost, minlength=nranks)
    # (JB_AUTO_IMBALANCE: the threshold, for rehearsals of the replicated-mesh answer on fewer ranks than the 8
    # on which BASELINE configs[4] reaches it)
    unbalanced = load.max() > float(os.environ.get("JB_AUTO_IMBALANCE", "1.15")) * load.mean()
    return "replicated" if (small and unbalanced) else "blocks"


# ------------------------------------------------------------------------------------------------
# device-side driver (needs the HIP library; imported lazily so that the host-only helpers above
# stay usable without a GPU)
class McblockDriver:
    """The cycle loop of the reference application: ``Step()`` = ``jaybenne::RadiationStep`` then
    ``HostUpdateTasks`` (ghost exchange, FillDerived -> UpdateDerived, EstimateTimestep);
    reference src/mcblock/mcblock_driver.cpp:38-74."""

    def __init__(self, pin: ParameterInput, rank: int = 0, nranks: int = 1, comm=None,
                 device=None, capacity_factor: float = 1.3, mesh: Mesh = None,
                 halo_rings: int = 1, decomposition: str = "blocks"):
        """``decomposition`` (several ranks): "blocks" -- the reference's: meshblocks dealt to ranks
        (``Mesh.partition`` by the cost ``block_costs`` estimates), photons handed over where they
        leave a rank's blocks; "replicated" -- every rank holds the whole mesh and follows its share
        of every block's photons (``jaybenne.MeshData``: for meshes that fit every GPU many times
        over); "auto" -- replicated when the mesh's fields take less than 1/64 of the device's memory
        and its blocks are too few or too unequal to balance (< 8 per rank, or cost spread > 1.5)."""
        from . import jaybenne as jb
        self.jb = jb
        self.pin = pin
        self.mesh = mesh if mesh is not None else Mesh.from_deck(pin)
        self.mcb = Initialize(pin)
        self.pkg = jb.Initialize(pin, self.mcb.opacity, self.mcb.scattering, self.mcb.eos,
                                 device=device, my_rank=rank)
        cost = block_costs(self.mesh, pin, self.mcb)
        if decomposition == "auto":
            decomposition = choose_decomposition(self.mesh, cost, nranks, device)
        if decomposition not in ("blocks", "replicated"):
            raise ValueError("decomposition must be 'blocks', 'replicated' or 'auto'")
        self.decomposition = decomposition if nranks > 1 else "blocks"
        replicated = self.decomposition == "replicated"
        if nranks > 1 and not replicated:
            self.mesh.partition(nranks, cost=cost)
        if replicated:
            share = self.pkg.Param("num_particles") / nranks
        else:
            n_local_blocks = int((self.mesh.owner == rank).sum()) if nranks > 1 else self.mesh.nblocks
            share = self.pkg.Param("num_particles") * n_local_blocks / self.mesh.nblocks
        capacity = int(share * capacity_factor) + 4096
        self.md = jb.MeshData(self.pkg, self.mesh, capacity, rank, nranks, comm, halo_rings,
                              replicated=replicated)
        # (not a parameter of the reference: DefragParticles -- here a sort of the swarm by cell, for
        # the locality of the cell gathers -- after every k-th cycle; 0 = never, as the reference)
        self.md.defrag_interval = pin.GetOrAddInteger("jaybenne", "defrag_interval", -1)
        self.tlim = pin.GetReal("parthenon/time", "tlim")
        self.nlim = pin.GetOrAddInteger("parthenon/time", "nlim", -1)
        self.time = 0.0
        self.ncycle = 0
        self.dt = jb.EstimateTimestepMesh(self.md)
        # ProblemGenerator + PostInitialization + initial ghost fill / FillDerived
        ic = ProblemGenerator(self.mesh, self.mcb, gids=self.md.resident_gids)
        for name in ("rho", "sie", "u"):
            self.md.set_field(name, ic[name], local=True)
        jb.InitializeRadiation(self.md, self.mcb.initial_radiation == "thermal")

    def HostUpdateTasks(self) -> None:
        """Ghost exchange of density / internal energy, sie = u / rho, new dt."""
        md = self.md
        if self.pkg.Param("do_feedback"):
            # UpdateFluid changed internal_energy in the owned blocks' interiors: refresh the
            # ghost zones and the halo copies (Parthenon's boundary exchange, mcblock_driver.cpp:68)
            md.exchange.refresh("u")
        md.fields["sie"].copy_(md.fields["u"] / md.fields["rho"])
        self.dt = self.jb.EstimateTimestepMesh(md)

    def Step(self):
        st = self.jb.RadiationStep(self.md, self.time, self.dt)
        if st != self.jb.TaskStatus.complete:
            return st
        self.HostUpdateTasks()
        self.time += self.dt
        self.ncycle += 1
        return st

    def Execute(self) -> None:
        while self.time < self.tlim and (self.nlim < 0 or self.ncycle < self.nlim):
            st = self.Step()
            if st != self.jb.TaskStatus.complete:
                raise RuntimeError(f"radiation step did not complete: {st!r}")
