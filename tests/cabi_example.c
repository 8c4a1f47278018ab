/* Minimal C consumer of the boundary: links libjaybenne_amd.so the way a host application would
 * (no Python, no torch).  Run by tests/test_cabi.py on the CPU; it only calls entry points that
 * need no GPU. */
#include <stdio.h>
#include <string.h>

#include "jaybenne_amd.h"

int main(void) {
  jb_params p;
  memset(&p, 0, sizeof p);
  p.num_particles = 10;
  p.min_swarm_occupancy = 2.0; /* outside [0, 1): the reference's PARTHENON_REQUIRE */
  jb_eos e = {JB_EOS_IDEAL_GAS, 0, 2.0 / 3.0, 1.5};
  jb_opacity o = {JB_OPAC_GRAY, 0, 0.0, 2.99792458e10, 5.670373e-5};
  jb_scattering s = {JB_SCAT_GRAY, 0, 1.0e3, 1.0};
  jb_context *ctx = NULL;
  const jb_status st = jb_initialize(&p, &e, &o, &s, 0, &ctx);
  printf("%s\n", jb_version());
  printf("status %d: %s\n", (int)st, jb_last_error());
  return (st == JB_ERR_INVALID && ctx == NULL) ? 0 : 1;
}
