#!/usr/bin/env python3
"""Regenerates the committed fixtures under tests/golden/ from the CPU oracle.

The reference holds no golden vectors for this path and cannot be built or imported here
(oracle/orc.h), so these fixtures are outputs of the ORACLE: they pin the
oracle (and through it the HIP kernels) against regressions; what pins the oracle to the reference
is tests/test_oracle_physics.py (the reference's own erf acceptance tests).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import orc  # noqa: E402
from helpers import load_deck, make_oracle, run_oracle_cycles  # noqa: E402
from step_cases import step_cases  # noqa: E402


def step_vectors(mode):
    orc.set_math_mode(mode)
    out = []
    names = [n for n, _ in orc.Step._fields_]
    for kind, d, tape in step_cases():
        st = orc.Step()
        for k, v in d.items():
            setattr(st, k, v)
        n = orc.call_step(kind, st, tape)
        out.append({"kind": kind, "in": d, "tape": tape, "ndraws": n,
                    "out": {k: (getattr(st, k).hex() if isinstance(getattr(st, k), float)
                                else int(getattr(st, k))) for k in names}})
    return out


SMALL_RUNS = [
    ("stepdiff", {"jaybenne/num_particles": 2000}, 2),
    ("stepdiff_ddmc", {"jaybenne/num_particles": 5000}, 2),
    ("stepdiff_smr_hybrid", {"jaybenne/num_particles": 8000}, 1),
]


def small_runs(mode):
    out = {}
    for deck, ov, cyc in SMALL_RUNS:
        pin = load_deck(deck, ov)
        O, mesh, _ = make_oracle(pin, mode, threads=4)
        run_oracle_cycles(O, pin, cyc)
        key = f"{deck}_m{mode}"
        out[key + "_tally"] = O.fields["tally"][mesh.interior()]
        out[key + "_x"] = O.sw["x"][:O.n].copy()
        out[key + "_rng"] = O.sw["rng"][:O.n].copy()
        out[key + "_events"] = np.array([O.events])
    return out


if __name__ == "__main__":
    json.dump({"libm": step_vectors(orc.MATH_LIBM), "portable": step_vectors(orc.MATH_PORTABLE)},
              open(os.path.join(HERE, "step_vectors.json"), "w"), indent=0)
    arrays = {}
    arrays.update(small_runs(orc.MATH_LIBM))
    arrays.update(small_runs(orc.MATH_PORTABLE))
    np.savez_compressed(os.path.join(HERE, "small_runs.npz"), **arrays)
    print("wrote step_vectors.json, small_runs.npz")
