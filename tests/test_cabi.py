"""The C-ABI shared library loads on a box without a GPU and exports exactly the entry points
include/jaybenne_amd.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "jaybenne_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(jb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from jaybenne_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/jaybenne_amd.h but not exported"
    assert sorted(_lib.PROTOTYPES) == names, "python binding and header disagree"
    assert lib.jb_version().startswith(b"jaybenne_amd")


def test_struct_layouts_match_the_header_sizes():
    """sizeof of every POD crossing the boundary, as the C compiler sees the header."""
    import subprocess
    import tempfile
    from jaybenne_amd import _lib
    src = r'''
#include <stdio.h>
#include "jaybenne_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(jb_params), sizeof(jb_eos), sizeof(jb_opacity),
         sizeof(jb_scattering), sizeof(jb_mesh_view), sizeof(jb_swarm_view),
         sizeof(jb_transport_stats), sizeof(jb_debug_step));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "t.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "t")
        subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), c, "-o", exe], check=True)
        sizes = [int(v) for v in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    py = [ctypes.sizeof(t) for t in (_lib.Params, _lib.Eos, _lib.Opacity, _lib.Scattering, _lib.MeshView,
                                     _lib.SwarmView, _lib.TransportStats, _lib.DebugStep)]
    assert sizes == py


def test_oracle_is_not_reachable_from_the_product():
    """The product package must not import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "jaybenne_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "liborc" not in text and "import orc" not in text and "oracle/" not in text, f
                assert "from oracle" not in text, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    from jaybenne_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def test_plain_c_program_links_and_gets_reference_style_errors(tmp_path):
    import subprocess
    exe = tmp_path / "cabi_example"
    lib_dir = os.path.join(ROOT, "jaybenne_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "cabi_example.c"), "-o", str(exe),
                    "-L", lib_dir, "-ljaybenne_amd", f"-Wl,-rpath,{lib_dir}"], check=True)
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "jaybenne_amd" in res.stdout and "swarm occupancy" in res.stdout


def test_cpp_mirror_header_is_self_contained():
    """include/jaybenne_amd.hpp (the C++ host-side mirror of jaybenne.hpp:48-78) compiles with a
    plain C++17 compiler: it needs the C ABI header only, no HIP, no Parthenon."""
    import subprocess
    src = "#include \"jaybenne_amd.hpp\"\nint main() { return (int)jaybenne_amd::TaskStatus::complete; }\n"
    res = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++",
                          "-I", os.path.join(ROOT, "include"), "-"], input=src, text=True,
                         capture_output=True)
    assert res.returncode == 0, res.stderr
    text = open(os.path.join(ROOT, "include", "jaybenne_amd.hpp")).read()
    for task in ("UpdateDerivedTransportFields", "SourcePhotons", "TransportPhotons",
                 "TransportPhotons_DDMC", "SampleDDMCBlockFace", "CheckCompletion",
                 "EvaluateRadiationEnergy", "UpdateFluid", "InitializeRadiation", "RadiationStep",
                 "EstimateTimestepMesh", "Initialize"):
        assert f" {task}(" in text, task


def test_shared_source_plan_is_partition_independent(tmp_path):
    """include/jaybenne_amd.hpp: PlanSource / SourceEpoch, the host-side arithmetic of SourcePhotons
    that the native hosts AND the (uncompiled) Parthenon adapter call -- compiled with the host
    compiler and run (tests/plan_source_test.cpp)."""
    import shutil
    import subprocess
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    exe = tmp_path / "plan_source_test"
    res = subprocess.run([cxx, "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "plan_source_test.cpp"), "-o", str(exe)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    res = subprocess.run([str(exe)], capture_output=True, text=True)
    assert res.returncode == 0 and "plan_source ok" in res.stdout, res.stdout + res.stderr
    # the adapter really goes through it
    text = open(os.path.join(ROOT, "adapters", "parthenon", "jaybenne_amd_tasks.cpp")).read()
    assert "jaybenne_amd::PlanSource(" in text and "jaybenne_amd::SourceEpoch(" in text


@pytest.mark.parametrize("deck,overrides,nranks", [
    ("stepdiff_smr", {}, 4),                                        # 2-D, 2 levels, 20 blocks
    ("stepdiff_smr_hybrid", {"_level2": True}, 5),                   # 2-D, 3 levels, 32 blocks
    ("stepdiff", {"parthenon/mesh/nx1": 16, "parthenon/mesh/nx2": 16, "parthenon/mesh/nx3": 32,
                  "parthenon/meshblock/nx1": 4, "parthenon/meshblock/nx2": 4,
                  "parthenon/meshblock/nx3": 4}, 8),                 # 3-D, periodic in y and z, 128 blocks
    ("stepdiff", {"parthenon/mesh/nx1": 128, "parthenon/meshblock/nx1": 16}, 3),   # examples/handoff_mpi.cpp
])
def test_shared_halo_plan_matches_the_python_host(tmp_path, deck, overrides, nranks):
    """include/jaybenne_amd.hpp: PlanHalo / FaceNeighbourLevels / PlanHaloRefresh -- the halo copies as the
    C++ hosts (examples/handoff_mpi.cpp, the Parthenon adapter) plan them -- against the Python host's
    Mesh.neighbours / blk_nbr_lev (tests/test_comm_gloo.py, test_mesh_topology.py) on the reference's meshes."""
    import shutil
    import subprocess
    import numpy as np
    from jaybenne_amd.deck import load_deck
    from jaybenne_amd.mesh import BC_PERIODIC, Mesh
    cxx = shutil.which("g++")
    if cxx is None:
        pytest.skip("no g++")
    ov = dict(overrides)
    level2 = ov.pop("_level2", False)
    pin = load_deck(deck, ov)
    if level2:
        pin.load_string("<parthenon/static_refinement2>\nlevel = 2\nx1min = -0.125\nx1max = 0.125\n"
                        "x2min = -0.125\nx2max = 0.125\nx3min = -0.25\nx3max = 0.25\n")
    mesh = Mesh.from_deck(pin)
    owner = mesh.partition(nranks)
    lines = [f"{mesh.ndim} {mesh.nblocks} {nranks}"]
    for d in range(3):
        lines.append(f"{float(mesh.gmin[d])!r} {float(mesh.gmax[d])!r} {mesh.nleaf[d]} {mesh.nx[d]}")
    lines.append(" ".join("1" if mesh.mesh_bc[f] == BC_PERIODIC else "0" for f in range(6)))
    lines.append(" ".join(str(int(v)) for v in np.asarray(mesh.leaf_map).ravel()))
    for g in range(mesh.nblocks):
        lines.append(f"{int(owner[g])} {int(mesh.blk_level[g])} " +
                     " ".join(f"{float(mesh.blk_xmin[g, d])!r} {float(mesh.blk_xmax[g, d])!r}" for d in range(3)))
    exe = tmp_path / "plan_halo_test"
    res = subprocess.run([cxx, "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                          os.path.join(ROOT, "tests", "plan_halo_test.cpp"), "-o", str(exe)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    res = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True)
    assert res.returncode == 0 and "plan_halo ok" in res.stdout, res.stdout[-2000:] + res.stderr
    halos = {int(l.split()[1]): [int(v) for v in l.split()[3:]] for l in res.stdout.splitlines() if l.startswith("rank ")}
    for r in range(nranks):
        mine = np.nonzero(owner == r)[0]
        assert halos[r] == [int(g) for g in mesh.neighbours(mine, 1)], r
    for l in res.stdout.splitlines():
        if l.startswith("nbr "):
            g, *lev = (int(v) for v in l.split()[1:])
            assert lev == [int(v) for v in mesh.blk_nbr_lev[g]], g
    text = open(os.path.join(ROOT, "adapters", "parthenon", "jaybenne_amd_tasks.cpp")).read()
    assert "jaybenne_amd::PlanHalo(" in text and "jaybenne_amd::PlanHaloRefresh(" in text


def test_hot_kernels_fit_three_waves_per_simd(tmp_path):
    """The tracking kernels are tuned for three waves per SIMD (512 / 3 -> 168 vector registers,
    allocated in eights); a change that pushes one of them over the edge silently costs a wave
    (measured: 90 -> 121 ms on the headline workload).  Compile the device code to assembly and
    read each hot kernel's register count and scratch use from its kernel descriptor."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "jaybenne_amd", "csrc", "jb_api.hip")
    flags = None
    for line in open(os.path.join(ROOT, "jaybenne_amd", "csrc", "Makefile")):
        if line.startswith("FLAGS"):
            flags = line.split("?=", 1)[1].replace("\\", " ")
        elif flags is not None and line.startswith(" "):
            flags += " " + line.replace("\\", " ")
        elif flags is not None:
            break
    flags = [f for f in flags.replace("$(ARCH)", "gfx950").split() if f not in ("-fPIC", "-Wall")]
    out = tmp_path / "jb.s"
    res = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", src, "-o", str(out)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    text = out.read_text()
    found = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
        name, body = m.group(1), m.group(2)
        vgpr = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1))
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1))
        found[name] = (vgpr, scratch)
    hot = {
        "IMC, 3-D, exact geometry, lean arithmetic in cell-local coordinates, one cell size (BASELINE configs[1])": "k_imc_cellILi3ELb1ELb1ELb1E",
        "IMC, 3-D, the same, blocks of several sizes": "k_imc_cellILi3ELb1ELb1ELb0E",
        "IMC, 2-D, the same (configs[3])": "k_imc_cellILi2ELb1ELb1ELb0E",
        "IMC, 1-D, one cell size (configs[0])": "k_imc_cellILi1ELb1ELb1ELb1E",
        "IMC, 3-D, one cell size, absorbing material": "k_imc_cellILi3ELb1ELb0ELb1E",
        "IMC, 3-D, exact geometry, lean arithmetic in x-space (JB_NO_IMC_CELL=1)": "k_transportILi3ELb0ELb1ELi2ELb1ELb1E",
        "IMC, 3-D, exact geometry, exact arithmetic": "k_transportILi3ELb0ELb1ELi2ELb1ELb0E",
        "IMC, 2-D, exact geometry, lean arithmetic (configs[3])": "k_transportILi2ELb0ELb1ELi2ELb1ELb1E",
        "IMC, 1-D, exact geometry, lean arithmetic (configs[0])": "k_transportILi1ELb0ELb1ELi2ELb1ELb1E",
        "all-DDMC, 3-D, quad-cooperative gather (configs[2] in 3-D)": "k_ddmc_allILi3ELb1ELi1E",
        "all-DDMC, 3-D, small mesh": "k_ddmc_allILi3ELb1ELi0E",
        "all-DDMC, 1-D, records in LDS (configs[2] as shipped)": "k_ddmc_allILi1ELb1ELi2E",
        "hybrid, 2-D, IMC phase in cell-local coordinates (configs[4])": "k_hybridILi2ELb1ELb1ELi3ELi1E",
        "hybrid, 1-D, IMC phase in cell-local coordinates": "k_hybridILi1ELb1ELb1ELi3ELi1E",
        "hybrid, 3-D, IMC phase in cell-local coordinates": "k_hybridILi3ELb1ELb1ELi3ELi1E",
        "hybrid, 2-D, remainder (both loops), cell-local (configs[4])": "k_hybridILi2ELb1ELb1ELi3ELi0E",
        "hybrid, 2-D, IMC phase, lean in x-space (JB_NO_IMC_CELL=1)": "k_hybridILi2ELb1ELb1ELi2ELi1E",
        "hybrid, 2-D, DDMC phase (configs[4])": "k_hybridILi2ELb1ELb1ELi0ELi2E",
        "hybrid, 3-D, IMC phase, lean in x-space": "k_hybridILi3ELb1ELb1ELi2ELi1E",
        "hybrid, 3-D, DDMC phase": "k_hybridILi3ELb1ELb1ELi0ELi2E",
    }
    def scratch_in_inner_loops(kernel):
        """scratch_load / scratch_store instructions inside a loop nested in the kernel's outer
        (service / event) loop: the event loop itself."""
        body = text[text.index(kernel + ":"):]
        body = body[:body.index("s_endpgm")]
        depth, hits = 0, []
        for line in body.split("\n"):
            m = re.search(r"Depth=(\d+)", line)
            if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
                depth = int(m.group(1)) if m else 0
            elif "scratch_" in line and depth >= 2:
                hits.append(line.strip())
        return hits

    for what, key in hot.items():
        names = [n for n in found if key in n]
        assert len(names) == 1, (what, names)
        vgpr, scratch = found[names[0]]
        assert vgpr <= 168, f"{what}: {vgpr} vector registers (> 168: two waves per SIMD)"
        if "k_imc_cell" in key and "3-D" not in what:
            # 1-D / 2-D: five waves per SIMD (96 registers); registers that do not fit are stored and
            # reloaded around the event loop (BASELINE configs[3]: 43.6 -> 43.0 ms), never inside it
            assert vgpr <= 96, f"{what}: {vgpr} vector registers (> 96: four waves per SIMD)"
            assert not scratch_in_inner_loops(names[0]), f"{what}: register spills inside the event loop"
            continue
        if "k_imc_cell" in key:
            # four waves per SIMD (128 registers): without a spill on wave-uniform geometry, with
            # a few registers stored around the event loop on blocks of several sizes
            assert vgpr <= 128, f"{what}: {vgpr} vector registers (> 128: three waves per SIMD)"
            if "several sizes" in what:
                assert not scratch_in_inner_loops(names[0]), f"{what}: register spills inside the event loop"
                continue
        if "hybrid" in what and "3-D" not in what and ("cell-local" in what or "x-space" in what):
            # The lean IMC phase of the hybrid kernel (cell-local, or in x-space on general geometry)
            # runs four waves per SIMD in 1-D / 2-D and the remainder launch three: the registers they
            # give up are stored and reloaded AROUND the event loops (BASELINE configs[4]: 60.1 ->
            # 55.8 ms; flagged general: 79.5 -> 73.1), never inside one
            if "remainder" not in what:
                assert vgpr <= 128, f"{what}: {vgpr} vector registers (> 128: three waves per SIMD)"
            assert not scratch_in_inner_loops(names[0]), f"{what}: register spills inside an event loop"
        else:
            assert scratch == 0 and not scratch_in_inner_loops(names[0]), f"{what}: {scratch} bytes of scratch per lane (register spills)"
    # the all-DDMC kernel is bound by the latency of its gathers and runs FOUR waves per SIMD:
    # 128 registers, and at most 40 KB of LDS per workgroup (its LDS tally is dynamic shared memory)
    # (static LDS + the most dynamic LDS a launch can ask for -- the tally of <= kLdsTally = 1024 cells,
    # 8 KB, and the step records of <= kLdsRecCells = 256 cells, 16 KB -- within the 64 KB a workgroup may have)
    ddmc = {n: v for n, v in found.items() if "k_ddmc_all" in n}
    assert len(ddmc) > 0
    max_dynamic_lds = 8 * 1024 + 64 * 256
    for n, (vgpr, scratch) in ddmc.items():
        assert vgpr <= 128 and scratch == 0, (n, vgpr, scratch)
        lds = int(re.search(r"\.amdhsa_kernel %s.*?\.amdhsa_group_segment_fixed_size (\d+)" % re.escape(n), text, re.S).group(1))
        assert lds <= 65536 - max_dynamic_lds, (n, lds)
    # no launch of the hybrid IMC/DDMC path touches scratch memory inside an event loop (the IMC and DDMC
    # phases in x-space fit 168 registers outright; the cell-local IMC phase and the remainder launch
    # trade registers that are dead across the loops for a wave per SIMD: above)
    hybrid = {n: v for n, v in found.items() if "k_hybrid" in n}
    assert len(hybrid) > 0
    spilling = {n: scratch_in_inner_loops(n) for n in hybrid if hybrid[n][1] != 0}
    assert not any(spilling.values()), {n: h[:2] for n, h in spilling.items() if h}
