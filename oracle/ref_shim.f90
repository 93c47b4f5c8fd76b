!===============================================================================
! ref_shim.f90 -- TEST INFRASTRUCTURE ONLY.
!
! C-callable (bind(C)) entry points around the *reference's own* Fortran
! procedures, so that pytest / the golden generator can call the real NDPP
! numerics compiled from /root/reference/src (see oracle/Makefile, target
! _ref/libndpp_ref.so).  Nothing in here is a restatement: every routine simply
! forwards to the reference module procedure named in its comment.
!
! This file is OUR code; the reference sources are compiled where they lie and
! are never copied into the repository.
!===============================================================================
module ref_shim
  use iso_c_binding
  use constants
  use global
  use legendre,         only: calc_pn, calc_int_pn_tablelin
  use freegas
  use search,           only: binary_search
  use scattdata_header, only: integrate_file4_cm_leg, tolab
  implicit none
contains

  ! module global's hidden numerics (global.F90:32-59) -- uninitialised there,
  ! defaults in constants.F90:70-100, set from ndpp.xml at ndpp.F90:355-423
  subroutine ref_set_params(sab_threshold_, brent_mu_thresh_, mu_tol, mu_its, &
                            eout_tol, eout_its, ne_per_grp_, sab_epts, ext_pts, &
                            inel_ext_pts) bind(C, name="ref_set_params")
    real(c_double), value :: sab_threshold_, brent_mu_thresh_, mu_tol, eout_tol
    integer(c_int), value :: mu_its, eout_its, ne_per_grp_, sab_epts, ext_pts
    integer(c_int), value :: inel_ext_pts
    SAB_THRESHOLD     = sab_threshold_
    BRENT_MU_THRESH   = brent_mu_thresh_
    ADAPTIVE_MU_TOL   = mu_tol
    ADAPTIVE_MU_ITS   = mu_its
    ADAPTIVE_EOUT_TOL = eout_tol
    ADAPTIVE_EOUT_ITS = eout_its
    NE_PER_GRP        = ne_per_grp_
    SAB_EPTS_PER_BIN  = sab_epts
    EXTEND_PTS        = ext_pts
    INEL_EXTEND_PTS   = inel_ext_pts
  end subroutine ref_set_params

  ! legendre.F90:349 calc_pn
  function ref_calc_pn(n, x) bind(C, name="ref_calc_pn") result(p)
    integer(c_int), value :: n
    real(c_double), value :: x
    real(c_double) :: p
    p = calc_pn(n, x)
  end function ref_calc_pn

  ! legendre.F90:22 calc_int_pn_tablelin
  subroutine ref_calc_int_pn_tablelin(n, xlo, xhi, flo, fhi, res) &
      bind(C, name="ref_calc_int_pn_tablelin")
    integer(c_int), value :: n
    real(c_double), value :: xlo, xhi, flo, fhi
    real(c_double), intent(out) :: res(n)
    res = calc_int_pn_tablelin(n, xlo, xhi, flo, fhi)
  end subroutine ref_calc_int_pn_tablelin

  ! search.F90:21 binary_search (real)
  function ref_binary_search(a, n, v) bind(C, name="ref_binary_search") result(i)
    integer(c_int), value :: n
    real(c_double), intent(in) :: a(n)
    real(c_double), value :: v
    integer(c_int) :: i
    i = binary_search(a, n, v)
  end function ref_binary_search

  ! freegas.F90:154 calc_FG_Eout_bounds
  subroutine ref_calc_fg_eout_bounds(A, kT, Ein, lo, hi) &
      bind(C, name="ref_calc_fg_eout_bounds")
    real(c_double), value :: A, kT, Ein
    real(c_double), intent(out) :: lo, hi
    call calc_FG_Eout_bounds(A, kT, Ein, lo, hi)
  end subroutine ref_calc_fg_eout_bounds

  ! freegas.F90:188 calc_sab
  function ref_calc_sab(A, kT, Ein, Eout, beta, mu) bind(C, name="ref_calc_sab") result(s)
    real(c_double), value :: A, kT, Ein, Eout, beta, mu
    real(c_double) :: s
    s = calc_sab(A, kT, Ein, Eout, beta, mu)
  end function ref_calc_sab

  ! freegas.F90:235 brent_mu
  function ref_brent_mu(A, kT, Ein, Eout, beta, thresh, lo, hi) &
      bind(C, name="ref_brent_mu") result(m)
    real(c_double), value :: A, kT, Ein, Eout, beta, thresh, lo, hi
    real(c_double) :: m
    m = brent_mu(A, kT, Ein, Eout, beta, thresh, lo, hi)
  end function ref_brent_mu

  ! freegas.F90:356 find_FG_mu
  subroutine ref_find_fg_mu(A, kT, Ein, Eout, mu2) bind(C, name="ref_find_fg_mu")
    real(c_double), value :: A, kT, Ein, Eout
    real(c_double), intent(out) :: mu2(2)
    call find_FG_mu(A, kT, Ein, Eout, mu2)
  end subroutine ref_find_fg_mu

  ! freegas.F90:415 calc_fgk
  function ref_calc_fgk(A, kT, Ein, Eout, l, mu, fEmu, gmu, M) &
      bind(C, name="ref_calc_fgk") result(v)
    real(c_double), value :: A, kT, Ein, Eout, mu
    integer(c_int), value :: l, M
    real(c_double), intent(in) :: fEmu(M), gmu(M)
    real(c_double) :: v
    v = calc_fgk(A, kT, Ein, Eout, l, mu, fEmu, gmu)
  end function ref_calc_fgk

  ! freegas.F90:482 adaptiveSimpsons_mu
  function ref_adaptive_simpsons_mu(A, kT, Ein, Eout, l, fEmu, gmu, M, lo, hi) &
      bind(C, name="ref_adaptive_simpsons_mu") result(v)
    real(c_double), value :: A, kT, Ein, Eout, lo, hi
    integer(c_int), value :: l, M
    real(c_double), intent(in) :: fEmu(M), gmu(M)
    real(c_double) :: v
    v = adaptiveSimpsons_mu(A, kT, Ein, Eout, l, fEmu, gmu, lo, hi)
  end function ref_adaptive_simpsons_mu

  ! freegas.F90:563 adaptiveSimpsons_Eout
  function ref_adaptive_simpsons_eout(A, kT, Ein, l, fEmu, gmu, M, lo, hi) &
      bind(C, name="ref_adaptive_simpsons_eout") result(v)
    real(c_double), value :: A, kT, Ein, lo, hi
    integer(c_int), value :: l, M
    real(c_double), intent(in) :: fEmu(M), gmu(M)
    real(c_double) :: v
    v = adaptiveSimpsons_Eout(A, kT, Ein, l, fEmu, gmu, lo, hi)
  end function ref_adaptive_simpsons_eout

  ! freegas.F90:18 integrate_freegas_leg; distro is (order, G) column-major
  subroutine ref_integrate_freegas_leg(Ein, A, kT, fEmu, gmu, M, E_bins, nb, &
                                       order, distro) &
      bind(C, name="ref_integrate_freegas_leg")
    real(c_double), value :: Ein, A, kT
    integer(c_int), value :: M, nb, order
    real(c_double), intent(in) :: fEmu(M), gmu(M), E_bins(nb)
    real(c_double), intent(out) :: distro(order, nb - 1)
    call integrate_freegas_leg(Ein, A, kT, fEmu, gmu, E_bins, order, distro)
  end subroutine ref_integrate_freegas_leg

  ! scattdata_header.F90:1466 tolab
  function ref_tolab(R, w) bind(C, name="ref_tolab") result(u)
    real(c_double), value :: R, w
    real(c_double) :: u
    u = tolab(R, w)
  end function ref_tolab

  ! scattdata_header.F90:956 integrate_file4_cm_leg.  The reference's callers
  ! pre-zero distro (scattdata_header.F90:545-546,569) and the routine relies on
  ! it (early return :1015) -- the shim does the same.
  subroutine ref_integrate_file4_cm_leg(fw, Ein, awr, Q, E_bins, nb, w, M, &
                                        order, distro) &
      bind(C, name="ref_integrate_file4_cm_leg")
    real(c_double), value :: Ein, awr, Q
    integer(c_int), value :: M, nb, order
    real(c_double), intent(in) :: fw(M), w(M), E_bins(nb)
    real(c_double), intent(out) :: distro(order, nb - 1)
    distro = ZERO
    call integrate_file4_cm_leg(fw, Ein, awr, Q, E_bins, w, order, distro)
  end subroutine ref_integrate_file4_cm_leg

end module ref_shim
