"""Physical constants handed to the package by the host's opacity object
(``opacity.GetRuntimePhysicalConstants()``, reference src/jaybenne/jaybenne.cpp:182-184).
singularity-opac is an empty submodule in the reference tree; these are the CGS CODATA-2010
values its default ``PhysicalConstantsCGS`` carries (assumed; the reference's only numeric pin
is ``ur0 = a T^4 = 7.5646e5`` in tst/stepdiff.py:34, met to 1.5e-4)."""

SPEED_OF_LIGHT = 2.99792458e10   # cm / s
STEFAN_BOLTZMANN = 5.670373e-5   # erg / cm^2 / s / K^4
