"""Inputs that drive the three step functions through every branch (shared by the oracle
golden-vector test and the device parity test)."""
import numpy as np


def step_cases():
    """Inputs that reach every branch of the three step functions (tape = uniforms replayed)."""
    c = 2.99792458e10
    base = dict(t_start=0.0, dt=3.335641e-11, ff=1.0, aa=0.0, ss=1.0e3, vv=c, dx_push=1 / 128,
                xl=-0.5, xu=-0.5 + 1 / 128, yl=-0.5, yu=0.5, zl=-0.5, zu=0.5,
                t=1e-12, x=-0.497, y=0.1, z=-0.2, vx=0.6 * c, vy=0.8 * c, vz=0.0, ip=2, jp=0, kp=0)
    cases = []
    tapes = [[0.3, 0.7, 0.11, 0.93], [0.999999, 0.2, 0.5, 0.5], [1e-9, 0.5], [0.5, 1e-9],
             [0.9, 0.9999999], [0.5, 0.3]]
    for nd in (1, 2, 3):
        for tape in tapes:
            for mods in ({}, dict(vx=-0.6 * c), dict(aa=2.0e3, ff=0.4), dict(ss=0.0, aa=0.0),
                         dict(yl=0.09, yu=0.11, zl=-0.21, zu=-0.19, vz=0.3 * c, vy=0.5 * c),
                         dict(t=3.335641e-11 - 1e-16), dict(x=-0.5 + 1 / 128 - 1e-12, vx=c)):
                d = dict(base, multi_d=int(nd >= 2), three_d=int(nd == 3), **mods)
                cases.append(("transport", d, tape))
    # DDMC step: absorption, the six leak directions, census
    P = dict(Px_l=0.04, Px_u=0.05, Py_l=0.03, Py_u=0.02, Pz_l=0.06, Pz_u=0.01)
    for nd in (1, 2, 3):
        gate = {k: (v if (k[1] == "x" or (k[1] == "y" and nd >= 2) or nd == 3) else 0.0)
                for k, v in P.items()}
        for xi2 in (0.0005, 0.05, 0.2, 0.4, 0.55, 0.7, 0.9, 0.999):
            for aa in (0.0, 50.0):
                d = dict(base, multi_d=int(nd >= 2), three_d=int(nd == 3), aa=aa, ff=0.8, **gate)
                cases.append(("ddmc", d, [0.5, xi2, 0.3, 0.6, 0.1, 0.2]))
        d = dict(base, multi_d=int(nd >= 2), three_d=int(nd == 3), **gate)
        cases.append(("ddmc", d, [1e-300 + 1e-12, 0.2, 0.3, 0.6, 0.1, 0.8]))   # far event
        cases.append(("ddmc", d, [1 - 1e-12, 0.25, 0.35, 0.65, 0.15, 0.85]))   # immediate event
        d2 = dict(d, t=3.335641e-11 - 1e-18)
        cases.append(("ddmc", d2, [0.5, 0.25, 0.35, 0.65, 0.15, 0.85]))        # census
    # albedo: at each face (accept / reject), not at a face
    eps_imc = 1e6 * 10 * np.finfo(float).eps
    dx = 1 / 128
    for nd in (1, 2, 3):
        geo = dict(base, multi_d=int(nd >= 2), three_d=int(nd == 3), yl=0.09, yu=0.11, zl=-0.21,
                   zu=-0.19, y=0.1, z=-0.2)
        spots = [dict(x=-0.5 + eps_imc * dx, vx=0.7 * c), dict(x=-0.5 + dx - eps_imc * dx, vx=-0.7 * c),
                 dict(y=0.09 + eps_imc * 0.02, vy=0.7 * c), dict(y=0.11 - eps_imc * 0.02, vy=-0.7 * c),
                 dict(z=-0.21 + eps_imc * 0.02, vz=0.7 * c), dict(z=-0.19 - eps_imc * 0.02, vz=-0.7 * c),
                 dict()]
        for sp in spots:
            for xi in (0.01, 0.99):
                cases.append(("albedo", dict(geo, **sp), [xi, 0.4, 0.8]))
    return cases


