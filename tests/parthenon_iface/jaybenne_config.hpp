// declared-interface stand-in for the host coupling that the reference generates from
// jaybenne_config.hpp.in (EOS / Opacity / Scattering model types of the host, HOST_* variable tags):
// only the members adapters/parthenon/jaybenne_amd_tasks.cpp calls.  NOT the reference's file.
#ifndef JB_IFACE_CONFIG_HPP_
#define JB_IFACE_CONFIG_HPP_
#include <string>
#include "parthenon_iface.hpp"

struct EOS {   // singularity IdealGas as mcblock builds it: (gamma - 1, cv)
  double gm1 = 2.0 / 3.0, cv = 1.5;
  double GruneisenParamFromDensityTemperature(double, double) const { return gm1; }
  double SpecificHeatFromDensityTemperature(double, double) const { return cv; }
  EOS GetOnDevice() const { return *this; }
};
struct RuntimePhysicalConstants { double c = 2.99792458e10, sb = 5.670374419e-5; };
struct Opacity {   // Gray(kappa): sigma_a = rho kappa
  double kappa = 0.0;
  RuntimePhysicalConstants GetRuntimePhysicalConstants() const { return {}; }
  double AbsorptionCoefficient(double rho, double, double) const { return rho * kappa; }
  Opacity GetOnDevice() const { return *this; }
};
struct Scattering {   // GrayS(kappa_s, apm): sigma_s = (rho / apm) kappa_s
  double kappa_s = 0.0, apm = 1.0;
  double TotalScatteringCoefficient(double rho, double, double) const { return (rho / apm) * kappa_s; }
  Scattering GetOnDevice() const { return *this; }
};
namespace field { namespace material {
struct density { static std::string name() { return "field.material.density"; } };
struct sie { static std::string name() { return "field.material.sie"; } };
struct internal_energy { static std::string name() { return "field.material.internal_energy"; } };
} }
#define HOST_DENSITY field::material::density
#define HOST_SPECIFIC_INTERNAL_ENERGY field::material::sie
#define HOST_UPDATE_ENERGY field::material::internal_energy
#endif
