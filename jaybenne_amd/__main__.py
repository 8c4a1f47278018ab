"""Run an input deck the way the reference's regression harness drives `mcblock`:

    python -m jaybenne_amd -i jaybenne_amd/decks/stepdiff.in parthenon/mesh/nx1=128 \\
           parthenon/meshblock/nx1=128 [--tolerance 0.05] [--comparison weighted_mean]

Trailing ``block/key=value`` arguments override deck values (Parthenon's command-line syntax,
which tst/regression_test.py:85-145 emulates by rewriting the deck).  After the last cycle the
energy tally is compared with the analytic solution of tst/stepdiff.py and the same five numbers
as tst/regression_test.py:408-412 are printed; the exit code is 0 iff the chosen criterion is
within the tolerance.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="python -m jaybenne_amd", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-i", "--input", required=True, help="Parthenon-style input deck")
    ap.add_argument("--tolerance", type=float, default=None,
                    help="pass/fail bound on the chosen error (0.05 stepdiff, 0.3 SMR decks in the reference)")
    ap.add_argument("--comparison", default="weighted_mean", choices=["mean", "pointwise", "weighted_mean"])
    ap.add_argument("--transverse-average", action="store_true",
                    help="average the tally over cells of equal x before comparing (uniform meshes)")
    ap.add_argument("--match-total-energy", action="store_true",
                    help="with --transverse-average: scale the solution to the tally's total energy "
                         "(fewer than one source particle per cell under-samples the energy)")
    ap.add_argument("--output", default=None,
                    help="write the final state to this file: .phdf (Parthenon-style HDF5, needs "
                         "libhdf5) or .npz.  Without it a deck with a <parthenon/output0> block of "
                         "file_type = hdf5 writes <problem_id>.out0.final.phdf into --output-dir")
    ap.add_argument("--output-dir", default=None, help="directory for deck-driven dumps (default: none written)")
    ap.add_argument("overrides", nargs="*", help="block/key=value")
    args = ap.parse_args(argv)

    import torch
    from . import analysis, mcblock
    from .deck import ParameterInput

    pin = ParameterInput.from_file(args.input)
    ov = {}
    for item in args.overrides:
        if "=" not in item:
            ap.error(f"override '{item}' is not of the form block/key=value")
        k, v = item.split("=", 1)
        ov[k] = v
    pin.modify(ov)
    if not torch.cuda.is_available():
        print("jaybenne_amd needs a GPU (the history loop runs only as HIP kernels)", file=sys.stderr)
        return 2
    drv = mcblock.McblockDriver(pin, device=torch.device("cuda", 0))
    print(f"problem {drv.mcb.problem_id}: {drv.mesh.ndim}-D, {drv.mesh.nblocks} meshblocks, "
          f"levels {sorted(set(drv.mesh.blk_level.tolist()))}, {drv.md.n} photons")
    t0 = time.perf_counter()
    while drv.time < drv.tlim:
        n0, e0 = drv.md.n, drv.md.events
        c0 = time.perf_counter()
        drv.Step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - c0
        print(f"cycle={drv.ncycle} time={drv.time:.6e} dt={drv.dt:.6e} photons={drv.md.n} "
              f"histories/s={n0 / dt:.3e} events/s={(drv.md.events - e0) / dt:.3e}")
    print(f"walltime used = {time.perf_counter() - t0:.2f} s")
    tally = drv.md.get_field("tally")
    if drv.mcb.problem_id.startswith("inf"):
        from . import constants
        solution = analysis.equilibrium_solution(drv.mcb.initial_temperature,
                                                 constants.STEFAN_BOLTZMANN, constants.SPEED_OF_LIGHT)
        print(f"domain-mean energy tally / (a T0^4): "
              f"{float(np.mean(np.asarray(tally)[drv.mesh.interior()])) / float(solution(0.0, 0.0)):.4f}")
    else:
        solution = analysis.ur_solution
    err = analysis.analytic_errors(drv.mesh, tally, drv.time, solution,
                                   transverse_average=args.transverse_average,
                                   match_total_energy=args.match_total_energy)
    print(f"Mean error:                     {err['mean_error']:.2e}")
    print(f"Mean fractional error:          {err['mean_frac_error']:.2e}")
    print(f"Mean weighted fractional error: {err['mean_frac_error_weighted']:.2e}")
    print(f"Max error:                      {err['max_error']:.2e}")
    print(f"Max fractional error:           {err['max_frac_error']:.2e}")
    out_path = args.output
    if out_path is None and args.output_dir is not None and \
            pin.GetOrAddString("parthenon/output0", "file_type", "none") == "hdf5":
        out_path = os.path.join(args.output_dir, f"{drv.mcb.problem_id}.out0.final.phdf")
    if out_path:
        from . import phdf
        if out_path.endswith(".npz") or not phdf.available():
            if not out_path.endswith(".npz"):
                print("libhdf5 not found: writing .npz instead", file=sys.stderr)
                out_path = os.path.splitext(out_path)[0] + ".npz"
            np.savez(out_path, tally=tally, time=drv.time, blk_xmin=drv.mesh.blk_xmin,
                     blk_dx=drv.mesh.blk_dx, blk_level=drv.mesh.blk_level)
        else:
            # <parthenon/output0> variables / swarm_variables of the deck (inputs/stepdiff_smr.in:86-94)
            names = {"field.material.density": "rho", "field.material.sie": "sie",
                     "field.material.internal_energy": "u", "field.jaybenne.energy_tally": "tally",
                     "field.jaybenne.fleck_factor": "fleck"}
            want = [v.strip() for v in pin.GetOrAddString(
                "parthenon/output0", "variables", "field.jaybenne.energy_tally").replace("&", "").split(",") if v.strip()]
            variables = {v: drv.md.get_field(names[v]) for v in want if v in names}
            sw = drv.md.get_swarm()
            svars = [v.strip() for v in pin.GetOrAddString(
                "parthenon/output0", "swarm_variables", "swarm.x, swarm.y").replace("&", "").split(",") if v.strip()]
            key = {"swarm.x": "x", "swarm.y": "y", "swarm.z": "z", "weight": "w", "time": "t", "energy": "e"}
            swarm = {"blk": drv.md.gids[sw["blk"]], "id": sw["id"]}
            swarm.update({v: sw[key[v]] for v in svars if v in key})
            phdf.write_dump(out_path, drv.mesh, drv.time, drv.dt, drv.ncycle, variables,
                            {"photons": swarm}, input_text=open(args.input).read())
        print(f"wrote {out_path}")
    if args.tolerance is None:
        return 0
    crit = {"mean": err["mean_frac_error"], "pointwise": err["max_frac_error"],
            "weighted_mean": err["mean_frac_error_weighted"]}[args.comparison]
    ok = crit <= args.tolerance
    print("TEST PASSED" if ok else "TEST FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
