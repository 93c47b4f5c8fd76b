/* Minimal C host: the Legendre moments of free-gas scattering off an H-1-like target for one
 * incoming energy, through the fine seam ndpp_integrate_freegas_leg (== integrate_freegas_leg,
 * freegas.F90:18).  Build:  gcc examples/freegas_leg.c -Iinclude -Lndpp_amd -lndpp_hip \
 *                           -Wl,-rpath,$PWD/ndpp_amd -o freegas_leg                          */
#include <stdio.h>
#include <stdlib.h>

#include "ndpp_hip.h"

int main(void) {
  ndpp_params p;
  ndpp_default_params(&p);
  p.order = 6;      /* P5 */
  p.mu_bins = 2001;
  double *f = malloc(sizeof(double) * p.mu_bins);
  for (int i = 0; i < p.mu_bins; ++i) f[i] = 0.5; /* isotropic in the CM frame */
  const double e_bins[3] = {0.0, 6.25e-7, 20.0};
  double distro[2 * 6];
  int rc = ndpp_integrate_freegas_leg(&p, 2.53e-8, 0.999167, 2.5301e-8, f, NULL, e_bins, 3, distro);
  if (rc != NDPP_OK) {
    fprintf(stderr, "libndpp_hip: %s\n", ndpp_last_error());
    return 1;
  }
  for (int g = 0; g < 2; ++g) {
    printf("group %d:", g + 1);
    for (int l = 0; l < p.order; ++l) printf(" % .10e", distro[g * p.order + l]);
    printf("\n");
  }
  free(f);
  return 0;
}
