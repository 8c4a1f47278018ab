"""Step functions of the oracle: committed golden vectors (both arithmetic flavours), draw counts
per SURVEY.md Appendix A, and the physical content of each branch."""
import json
import os

import numpy as np
import pytest

from oracle import orc
from step_cases import step_cases

HERE = os.path.dirname(os.path.abspath(__file__))
C = 2.99792458e10


@pytest.fixture(scope="module")
def golden():
    return json.load(open(os.path.join(HERE, "golden", "step_vectors.json")))


@pytest.mark.parametrize("mode_name,mode", [("libm", orc.MATH_LIBM), ("portable", orc.MATH_PORTABLE)])
def test_step_functions_match_golden_vectors(golden, mode_name, mode):
    orc.set_math_mode(mode)
    cases = step_cases()
    vec = golden[mode_name]
    assert len(vec) == len(cases)
    for (kind, d, tape), g in zip(cases, vec):
        assert g["kind"] == kind and g["tape"] == tape
        st = orc.Step()
        for k, v in d.items():
            setattr(st, k, v)
        assert orc.call_step(kind, st, tape) == g["ndraws"]
        for k, v in g["out"].items():
            got = getattr(st, k)
            want = float.fromhex(v) if isinstance(v, str) else v
            assert got == want or (got != got and want != want), (kind, k, got, want)


def _step(**kw):
    base = dict(t_start=0.0, dt=3.335641e-11, ff=1.0, aa=0.0, ss=1.0e3, vv=C, dx_push=1 / 128,
                multi_d=0, three_d=0, xl=-0.5, xu=-0.5 + 1 / 128, yl=-0.5, yu=0.5, zl=-0.5, zu=0.5,
                t=1e-12, x=-0.497, y=0.1, z=-0.2, vx=0.6 * C, vy=0.8 * C, vz=0.0, ip=2, jp=0, kp=0)
    base.update(kw)
    st = orc.Step()
    for k, v in base.items():
        setattr(st, k, v)
    return st


def test_transport_step_scatter_distance_and_draw_order():
    orc.set_math_mode(orc.MATH_LIBM)
    st = _step()
    assert orc.call_step("transport", st, [0.3, 0.7]) == 2      # always two draws
    d_sc = -np.log(0.7) / 1.0e3                                 # second draw -> scattering distance
    assert st.is_scattered == 1 and st.is_absorbed == 0
    assert st.t == pytest.approx(1e-12 + d_sc / C, rel=1e-15)
    assert st.x == pytest.approx(-0.497 + 0.6 * d_sc, rel=1e-15)
    assert st.y == 0.1 and st.z == -0.2                         # 1-D: y, z untouched


def test_transport_step_face_crossing_is_nudged_outside():
    orc.set_math_mode(orc.MATH_LIBM)
    st = _step(ss=1e-3, x=-0.495)
    orc.call_step("transport", st, [0.5, 0.5])
    eps = 1e6 * 10 * np.finfo(float).eps / 128
    assert st.is_scattered == 0 and st.x == (-0.5 + 1 / 128) + eps
    st = _step(ss=1e-3, vx=-0.6 * C)
    orc.call_step("transport", st, [0.5, 0.5])
    assert st.x == -0.5 - eps
    # a step is never longer than the smallest cell extent of the block (dx_push)
    st = _step(ss=1e-3, x=-0.4999, vx=0.6 * C)
    orc.call_step("transport", st, [0.5, 0.5])
    assert st.t == pytest.approx(1e-12 + (1 / 128) / C, rel=1e-14) and st.x < -0.5 + 1 / 128


def test_transport_step_census():
    orc.set_math_mode(orc.MATH_LIBM)
    st = _step(ss=1e-3, t=3.335641e-11 - 1e-15)
    orc.call_step("transport", st, [0.5, 0.5])
    assert st.t >= 3.335641e-11 - 1e-24 and st.is_scattered == 0 and st.is_absorbed == 0


def test_transport_step_absorption_needs_both_comparisons():
    orc.set_math_mode(orc.MATH_LIBM)
    st = _step(aa=2.0e3, ff=1.0, ss=1.0)
    orc.call_step("transport", st, [0.9, 0.5])
    assert st.is_absorbed == 1
    st = _step(aa=2.0e3, ff=0.0, ss=1.0)       # f = 0: absorption becomes effective scattering
    orc.call_step("transport", st, [0.9, 0.9])
    assert st.is_absorbed == 0 and st.is_scattered == 1


def test_ddmc_leak_channels_in_reference_order():
    orc.set_math_mode(orc.MATH_LIBM)
    P = dict(Px_l=0.04, Px_u=0.05, Py_l=0.03, Py_u=0.02, Pz_l=0.06, Pz_u=0.01)
    dx, dy = 1 / 128, 1.0
    leak = [P["Px_l"] / dx, P["Px_u"] / dx, P["Py_l"] / dy, P["Py_u"] / dy, P["Pz_l"] / dy, P["Pz_u"] / dy]
    cum = np.cumsum(leak) / sum(leak)
    want = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    prev = 0.0
    for ch in range(6):
        xi2 = 0.5 * (prev + cum[ch])
        prev = cum[ch]
        st = _step(multi_d=1, three_d=1, ip=5, jp=6, kp=7, **P)
        n = orc.call_step("ddmc", st, [0.5, xi2, 0.3, 0.6])
        assert n == 4
        assert (st.ip - 5, st.jp - 6, st.kp - 7) == want[ch]
        v = np.array([st.vx, st.vy, st.vz])
        assert np.linalg.norm(v) == pytest.approx(C, rel=1e-14)
        axis = ch // 2
        assert np.sign(v[axis]) == (1 if ch % 2 else -1)       # outward


def test_ddmc_census_resamples_position_and_direction():
    orc.set_math_mode(orc.MATH_LIBM)
    st = _step(Px_l=0.01, Px_u=0.01, t=3.335641e-11 - 1e-18)
    n = orc.call_step("ddmc", st, [0.5, 0.25, 0.35, 0.65, 0.15, 0.85])
    assert n == 6
    assert st.z == -0.5 + 0.25 * 1.0 and st.x == -0.5 + 0.35 / 128 and st.y == -0.5 + 0.65
    assert st.vz == C * (1.0 - 2.0 * 0.15)


def test_albedo_admits_to_cell_centre_or_rejects_outward():
    orc.set_math_mode(orc.MATH_LIBM)
    eps = 1e6 * 10 * np.finfo(float).eps
    dx = 1 / 128
    st = _step(x=-0.5 + eps * dx, vx=0.7 * C)
    assert orc.call_step("albedo", st, [0.01, 0.4, 0.8]) == 1
    assert st.is_rejected == 0 and st.x == 0.5 * (-0.5 + (-0.5 + dx))
    st = _step(x=-0.5 + eps * dx, vx=0.7 * C)
    assert orc.call_step("albedo", st, [0.99, 0.4, 0.8]) == 3
    assert st.is_rejected == 1 and st.vx < 0 and st.x == -0.5 - eps * dx
    st = _step()                                   # not at a face: no draw, recentred
    assert orc.call_step("albedo", st, [0.99]) == 0 and st.x == 0.5 * (-0.5 + (-0.5 + dx))


def test_isotropic_samplers_and_planck():
    orc.set_math_mode(orc.MATH_LIBM)
    v, n = orc.call_scatter(C, [0.75, 0.25])
    assert n == 2 and v[2] == C * 0.5 and np.linalg.norm(v) == pytest.approx(C, rel=1e-15)
    v, n = orc.call_face_iso_dir(-C, [0.25, 0.0])
    assert n == 2 and v[0] == -C * 0.5                       # mu = sqrt(xi)
    e, n = orc.call_planck(5.670373e-5, 1.0e5, [0.5, 0.5, 0.5, 0.5, 0.5])
    assert n == 5 and e == pytest.approx(-np.log(0.5 ** 4) * 5.670373e-5 * 1.0e5, rel=1e-15)
    # series index: xi0 * pi^4/90 just above 1 -> l = 2
    e2, _ = orc.call_planck(1.0, 1.0, [0.95, 0.5, 0.5, 0.5, 0.5])
    assert e2 == pytest.approx(-0.5 * np.log(0.5 ** 4), rel=1e-15)


def test_block_face_resampling_helpers():
    i, x, n = orc.call_face_2d(7, 0.01, 0.3, 0.5, [0.1, 0.5], 8, 0.25)
    assert (i, n) == (7, 2) and x == 0.25 - 0.01 * 0.5
    i, x, n = orc.call_face_2d(7, 0.01, 0.3, 0.5, [0.9, 0.5], 7, 0.25)
    assert (i, n) == (8, 2) and x == 0.25 + 0.01 * 0.5
    ij, x12, n = orc.call_face_3d(3, 9, 0.01, 0.02, [0.1, 0.2, 0.3, 0.4], [0.25, 0.5, 0.5], [0, 0], [0.5, -0.25])
    assert n == 3 and ij == [4, 9] and x12[0] == 0.5 + 0.005 and x12[1] == -0.25 - 0.01


# ------------------------------------------------------------------------------------------------
# Vectors the oracle did NOT write: tests/golden/hand_step_vectors.json is worked
# out on paper from the reference's text (dyadic inputs, closed-form results), and carries the one
# vector that came out of the reference's own headers.
def hand_vectors():
    import math
    doc = json.load(open(os.path.join(HERE, "golden", "hand_step_vectors.json")))
    names = {"eps_imc": 1e7 * 2.0 ** -52, "eps_ddmc": 1e9 * 2.0 ** -52,
             "ln2": float.fromhex("0x1.62e42fefa39efp-1"), "sqrt": math.sqrt}
    assert names["eps_imc"] == 1.0e6 * (10.0 * np.finfo(float).eps)       # transport_utils.hpp:24
    assert names["eps_ddmc"] == 1.0e8 * (10.0 * np.finfo(float).eps)      # :25

    def val(v):
        return eval(v, {"__builtins__": {}}, names) if isinstance(v, str) else v

    out = []
    for c in doc["cases"]:
        d = dict(doc["ddmc_common"])
        d.update({k: val(v) for k, v in c["in"].items()})
        exact = {k: val(v) for k, v in c.get("exact", {}).items()}
        close = {k: (val(v[0]), float(v[1])) for k, v in c.get("close", {}).items()}
        out.append((c["name"], c["kind"], d, c["tape"], c["ndraws"], exact, close))
    return out, doc["survey_appendix_d"]


def check_hand_case(name, got, ndraws, want_draws, exact, close):
    assert ndraws == want_draws, (name, ndraws)
    for k, v in exact.items():
        assert getattr(got, k) == v, (name, k, getattr(got, k), v)
    for k, (v, tol) in close.items():
        assert abs(getattr(got, k) - v) <= tol, (name, k, getattr(got, k), v)


@pytest.mark.parametrize("mode", [orc.MATH_LIBM, orc.MATH_PORTABLE])
def test_hand_written_step_vectors(mode):
    orc.set_math_mode(mode)
    cases, _ = hand_vectors()
    kinds = set()
    for name, kind, d, tape, ndraws, exact, close in cases:
        st = orc.Step()
        for k, v in d.items():
            setattr(st, k, v)
        n = orc.call_step(kind, st, tape)
        check_hand_case(name, st, n, ndraws, exact, close)
        kinds.add((kind, ndraws))
    assert len(cases) >= 20 and len(kinds) >= 7


@pytest.mark.parametrize("mode", [orc.MATH_LIBM, orc.MATH_PORTABLE])
def test_survey_appendix_d_vector_from_the_reference_headers(mode):
    """ptcl_transport_step of the reference (transport_utils.hpp:111-160, compiled by
    the survey from the reference's own header) on tape {0.3, 0.7, ...} gave a scatter at
    t = 1.1897395495477489e-14, x = -0.49728599503363674 after two draws."""
    orc.set_math_mode(mode)
    _, app_d = hand_vectors()
    assert app_d["in"]["vx"] == 0.6 * C and app_d["in"]["vy"] == 0.8 * C
    st = orc.Step()
    for k, v in app_d["in"].items():
        setattr(st, k, v)
    assert orc.call_step(app_d["kind"], st, app_d["tape"]) == app_d["ndraws"]
    for k, v in app_d["expect"].items():
        assert getattr(st, k) == v, (k, getattr(st, k), v)
