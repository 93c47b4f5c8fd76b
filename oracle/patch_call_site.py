#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Applies the call-site changes of INTEGRATION.md section 5 to a COPY of
the reference's ndpp.F90 (read where it lies, written to the path given, which the Makefile puts
under oracle/_ref/): in preprocess,
  `call calc_scatt(...)`     (ndpp.F90:607-609) becomes `call calc_scatt_hip(..., hip_ierr)`,
  `call calc_chi(...)`       (ndpp.F90:713)     becomes `call calc_chi_hip(..., hip_ierr)`,
  `call calc_scattsab(...)`  (ndpp.F90:773-775) becomes `call calc_scattsab_hip(..., hip_ierr)`,
all three from fortran/ndpp_hip_mod.f90.  Nothing else changes: the ACE reader (neutron and
thermal tables), the XML driver, sab_egrid, tolerance, thinning, group indices and the writers
(print_scatt, print_chi) of the resulting executable are the reference's own.
usage: patch_call_site.py <reference ndpp.F90> <output>"""
import sys

src, dst = sys.argv[1], sys.argv[2]
text = open(src).read()


def once(old, new):
    global text
    assert text.count(old) == 1, f"expected exactly one occurrence of {old!r}"
    text = text.replace(old, new)


once("  use scatt\n", "  use scatt\n  use ndpp_hip_mod, only: calc_scatt_hip, calc_chi_hip, calc_scattsab_hip, ndpp_hip_error\n")
once("      integer                   :: g              ! Energy group index\n",
     "      integer                   :: g              ! Energy group index\n"
     "      integer                   :: hip_ierr       ! status of the libndpp_hip call\n")
once("          call calc_scatt(nuc, self % energy_bins, self % scatt_type, &\n",
     "          call calc_scatt_hip(nuc, self % energy_bins, self % scatt_type, &\n")
once("            self % Ein_inel, el_mat, inel_mat, nuinel_mat)\n",
     "            self % Ein_inel, el_mat, inel_mat, nuinel_mat, hip_ierr)\n"
     "          if (hip_ierr /= 0) call fatal_error(\"libndpp_hip: \" // trim(ndpp_hip_error()))\n")
once("              call calc_chi(nuc, self % energy_bins, Ein_chi, chi_t, chi_p, chi_d)\n",
     "              call calc_chi_hip(nuc, self % energy_bins, Ein_chi, chi_t, chi_p, chi_d, hip_ierr)\n"
     "              if (hip_ierr /= 0) call fatal_error(\"libndpp_hip: \" // trim(ndpp_hip_error()))\n")
once("          call calc_scattsab(sab, self % energy_bins, self % scatt_type, &\n",
     "          call calc_scattsab_hip(sab, self % energy_bins, self % scatt_type, &\n")
once("                             self % Ein_el)\n",
     "                             self % Ein_el, hip_ierr)\n"
     "          if (hip_ierr /= 0) call fatal_error(\"libndpp_hip: \" // trim(ndpp_hip_error()))\n")
open(dst, "w").write(text)
