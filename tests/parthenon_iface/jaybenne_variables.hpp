// declared-interface stand-in for the variable tags of the reference's jaybenne_variables.hpp (names only:
// each tag is a type with a static name()).  NOT the reference's file.
#ifndef JB_IFACE_VARIABLES_HPP_
#define JB_IFACE_VARIABLES_HPP_
#include <string>
#include "parthenon_iface.hpp"
#include "jaybenne_config.hpp"

static const std::string photons_swarm_name = "photons";
#define JB_IFACE_TAG(ns, v) struct v { static std::string name() { return #ns "." #v; } }
namespace field { namespace jaybenne {
JB_IFACE_TAG(field.jaybenne, energy_tally);
JB_IFACE_TAG(field.jaybenne, fleck_factor);
JB_IFACE_TAG(field.jaybenne, ddmc_face_prob);
JB_IFACE_TAG(field.jaybenne, source_ew_per_cell);
JB_IFACE_TAG(field.jaybenne, source_num_per_cell);
JB_IFACE_TAG(field.jaybenne, energy_delta);
namespace host {
typedef HOST_DENSITY density;
typedef HOST_SPECIFIC_INTERNAL_ENERGY sie;
typedef HOST_UPDATE_ENERGY update_energy;
}
} }
namespace particle { namespace photons {
JB_IFACE_TAG(particle.photons, time);
JB_IFACE_TAG(particle.photons, weight);
JB_IFACE_TAG(particle.photons, energy);
JB_IFACE_TAG(particle.photons, v);
JB_IFACE_TAG(particle.photons, ijk);
} }
#endif
