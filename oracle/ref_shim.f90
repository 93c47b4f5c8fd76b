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
  use scattdata_header, only: integrate_file4_cm_leg, tolab, cast_to_unitbase, &
                              interp_unitbase, integrate_file6_cm_leg, &
                              integrate_file6_lab_leg, law9_scatter_lab_leg, ScattData
  use ace_header,       only: DistEnergy, SAlphaBeta, Nuclide, Reaction
  use chi,              only: calc_chi
  use scatt,            only: apply_tol_scatt, create_Ein_grid, calc_scatt, print_scatt_bin, &
                              print_scatt_ascii
  use chi,              only: print_chi_bin, print_chi_ascii
  use string,           only: to_str
  use output,           only: print_ascii_array
  use thin,             only: thin_grid
  use endf_header,      only: Tab1
  use sab,              only: integrate_sab_el, integrate_sab_inel, combine_sab_grid, sab_egrid
  use array_merge,      only: merge
  use interpolation,    only: interpolate_tab1
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


  ! array_merge.F90:13 merge; returns the merged length, result in res(1:n)
  subroutine ref_merge(a, na, b, nb, res, n) bind(C, name="ref_merge")
    integer(c_int), value :: na, nb
    real(c_double), intent(in) :: a(na), b(nb)
    real(c_double), intent(out) :: res(na + nb)
    integer(c_int), intent(out) :: n
    real(8), allocatable :: r(:)
    call merge(a, b, r)
    n = size(r)
    res(1:n) = r
  end subroutine ref_merge

  ! interpolation.F90:24 interpolate_tab1_array
  function ref_interpolate_tab1(data, nd, x) bind(C, name="ref_interpolate_tab1") result(y)
    integer(c_int), value :: nd
    real(c_double), intent(in) :: data(nd)
    real(c_double), value :: x
    real(c_double) :: y
    y = interpolate_tab1(data, x)
  end function ref_interpolate_tab1

  ! scattdata_header.F90:1521 unitbase (= cast_to_unitbase x2 + interp_unitbase)
  ! for two tabulated rows given explicitly.  f1 is (M, np1), f2 is (M, np2).
  ! Outputs: nub, Eout(nub), pdf(nub), fEmu(M, nub) (caller sizes for np1+np2).
  subroutine ref_unitbase(Ein, M, np1, eout1, pdf1, intt1, f1, Ei1, &
                          np2, eout2, pdf2, intt2, f2, Ei2, nub, Eout, pdf, INTT, fEmu) &
      bind(C, name="ref_unitbase")
    real(c_double), value :: Ein, Ei1, Ei2
    integer(c_int), value :: M, np1, np2, intt1, intt2
    real(c_double), intent(in) :: eout1(np1), pdf1(np1), f1(M, np1)
    real(c_double), intent(in) :: eout2(np2), pdf2(np2), f2(M, np2)
    integer(c_int), intent(out) :: nub, INTT
    real(c_double), intent(out) :: Eout(np1 + np2), pdf(np1 + np2), fEmu(M, np1 + np2)
    real(8), allocatable :: e1(:), p1(:), c1(:), e2(:), p2(:), c2(:), ub1(:), ub2(:)
    real(8), allocatable :: fa1(:,:), fa2(:,:), Eo(:), pd(:), fE(:,:)
    integer :: it
    allocate(e1(np1), p1(np1), c1(np1), e2(np2), p2(np2), c2(np2))
    allocate(fa1(M, np1), fa2(M, np2))
    e1 = eout1; p1 = pdf1; c1 = ZERO; e2 = eout2; p2 = pdf2; c2 = ZERO
    fa1 = f1; fa2 = f2
    call cast_to_unitbase(e1, p1, c1, intt1, ub1)
    call cast_to_unitbase(e2, p2, c2, intt2, ub2)
    call interp_unitbase(Ein, ub1, e1, p1, intt1, fa1, Ei1, ub2, e2, p2, intt2, fa2, Ei2, &
                         Eo, pd, it, fE)
    nub = size(Eo)
    INTT = it
    Eout(1:nub) = Eo
    pdf(1:nub) = pd
    fEmu(:, 1:nub) = fE
  end subroutine ref_unitbase

  ! scattdata_header.F90:1085 integrate_file6_cm_leg (distro pre-zeroed like the caller, :529)
  subroutine ref_integrate_file6_cm_leg(fEmu, M, np, mu, Ein, awr, Eout, INTT, pdf, &
                                        E_bins, nb, order, distro) &
      bind(C, name="ref_integrate_file6_cm_leg")
    integer(c_int), value :: M, np, INTT, nb, order
    real(c_double), value :: Ein, awr
    real(c_double), intent(in) :: fEmu(M, np), mu(M), Eout(np), pdf(np), E_bins(nb)
    real(c_double), intent(out) :: distro(order, nb - 1)
    distro = ZERO
    call integrate_file6_cm_leg(fEmu, mu, Ein, awr, Eout, INTT, pdf, E_bins, order, distro)
  end subroutine ref_integrate_file6_cm_leg

  ! scattdata_header.F90:1334 integrate_file6_lab_leg
  subroutine ref_integrate_file6_lab_leg(fEmu, M, np, mu, Eout, INTT, pdf, E_bins, nb, &
                                         order, distro) &
      bind(C, name="ref_integrate_file6_lab_leg")
    integer(c_int), value :: M, np, INTT, nb, order
    real(c_double), intent(in) :: fEmu(M, np), mu(M), Eout(np), pdf(np), E_bins(nb)
    real(c_double), intent(out) :: distro(order, nb - 1)
    distro = ZERO
    call integrate_file6_lab_leg(fEmu, mu, Eout, INTT, pdf, E_bins, order, distro)
  end subroutine ref_integrate_file6_lab_leg

  ! scattdata_header.F90:1274 law9_scatter_lab_leg; edata = edist % data
  subroutine ref_law9_scatter_lab_leg(fmu, M, edata, nd, Ein, E_bins, nb, mu, order, distro) &
      bind(C, name="ref_law9_scatter_lab_leg")
    integer(c_int), value :: M, nd, nb, order
    real(c_double), value :: Ein
    real(c_double), intent(in) :: fmu(M), edata(nd), E_bins(nb), mu(M)
    real(c_double), intent(out) :: distro(order, nb - 1)
    type(DistEnergy), pointer :: ed
    allocate(ed)
    ed % law = 9
    allocate(ed % data(nd))
    ed % data = edata
    distro = ZERO
    call law9_scatter_lab_leg(fmu, ed, Ein, E_bins, mu, order, distro)
    deallocate(ed % data)
    deallocate(ed)
  end subroutine ref_law9_scatter_lab_leg


  ! Build a SAlphaBeta (ace_header.F90:201-235) from flat arrays.  Discrete
  ! secondary modes use e_out(NEo,NEi), mu(NMU,NEo,NEi); the continuous mode the
  ! CSR triple cptr/ce_out/cpdf/cmu(NMU, sum).  n_el_ein = 0: no elastic data.
  subroutine build_sab(t, thr_inel, thr_el, NEi, NEo, NMU, mode, ei, sig, e_out, mu, &
                       cptr, ce_out, cpdf, cmu, el_mode, NEe, NMUe, ee, eP, emu)
    type(SAlphaBeta), pointer, intent(out) :: t
    real(c_double), intent(in) :: thr_inel, thr_el
    integer(c_int), intent(in) :: NEi, NEo, NMU, mode, el_mode, NEe, NMUe
    real(c_double), intent(in) :: ei(*), sig(*), e_out(*), mu(*), ce_out(*), cpdf(*), cmu(*)
    integer(c_int), intent(in) :: cptr(*)
    real(c_double), intent(in) :: ee(*), eP(*), emu(*)
    integer :: k, n, o
    allocate(t)
    t % name = 'synth.00t'; t % awr = ONE; t % kT = 2.53E-8_8; t % n_zaid = 0
    t % threshold_inelastic = thr_inel
    t % threshold_elastic = thr_el
    t % n_inelastic_e_in = NEi; t % n_inelastic_e_out = NEo; t % n_inelastic_mu = NMU
    t % secondary_mode = mode
    allocate(t % inelastic_e_in(NEi), t % inelastic_sigma(NEi))
    t % inelastic_e_in = ei(1:NEi); t % inelastic_sigma = sig(1:NEi)
    if (mode /= SAB_SECONDARY_CONT) then
      allocate(t % inelastic_e_out(NEo, NEi), t % inelastic_mu(NMU, NEo, NEi))
      t % inelastic_e_out = reshape(e_out(1:NEo*NEi), (/ NEo, NEi /))
      t % inelastic_mu = reshape(mu(1:NMU*NEo*NEi), (/ NMU, NEo, NEi /))
    else
      allocate(t % inelastic_data(NEi))
      do k = 1, NEi
        o = cptr(k); n = cptr(k + 1) - cptr(k)
        t % inelastic_data(k) % n_e_out = n
        allocate(t % inelastic_data(k) % e_out(n), t % inelastic_data(k) % e_out_pdf(n))
        allocate(t % inelastic_data(k) % e_out_cdf(n), t % inelastic_data(k) % mu(NMU, n))
        t % inelastic_data(k) % e_out = ce_out(o + 1 : o + n)
        t % inelastic_data(k) % e_out_pdf = cpdf(o + 1 : o + n)
        t % inelastic_data(k) % e_out_cdf = ZERO
        t % inelastic_data(k) % mu = reshape(cmu(o*NMU + 1 : (o + n)*NMU), (/ NMU, n /))
      end do
    end if
    t % elastic_mode = el_mode; t % n_elastic_e_in = NEe; t % n_elastic_mu = NMUe
    if (NEe > 0) then
      allocate(t % elastic_e_in(NEe), t % elastic_P(NEe))
      t % elastic_e_in = ee(1:NEe); t % elastic_P = eP(1:NEe)
      if (NMUe > 0) then
        allocate(t % elastic_mu(NMUe, NEe))
        t % elastic_mu = reshape(emu(1:NMUe*NEe), (/ NMUe, NEe /))
      end if
    end if
  end subroutine build_sab

  ! calc_scattsab's Legendre path (scatt.F90:543-596): integrate_sab_el (sab.F90:21),
  ! integrate_sab_inel (:117), combine_sab_grid (:415).  lorder = scatt_order.
  subroutine ref_calc_scattsab(thr_inel, thr_el, NEi, NEo, NMU, mode, ei, sig, e_out, mu, &
                               cptr, ce_out, cpdf, cmu, el_mode, NEe, NMUe, ee, eP, emu, &
                               E_grid, nE, e_bins, nb, lorder, el, inel, scatt_mat) &
      bind(C, name="ref_calc_scattsab")
    real(c_double), value :: thr_inel, thr_el
    integer(c_int), value :: NEi, NEo, NMU, mode, el_mode, NEe, NMUe, nE, nb, lorder
    real(c_double), intent(in) :: ei(*), sig(*), e_out(*), mu(*), ce_out(*), cpdf(*), cmu(*)
    integer(c_int), intent(in) :: cptr(*)
    real(c_double), intent(in) :: ee(*), eP(*), emu(*), E_grid(nE), e_bins(nb)
    real(c_double), intent(out) :: el(lorder + 1, nb - 1, nE), inel(lorder + 1, nb - 1, nE)
    real(c_double), intent(out) :: scatt_mat(lorder + 1, nb - 1, nE)
    type(SAlphaBeta), pointer :: t
    real(8), allocatable :: sm(:,:,:)
    call build_sab(t, thr_inel, thr_el, NEi, NEo, NMU, mode, ei, sig, e_out, mu, cptr, ce_out, &
                   cpdf, cmu, el_mode, NEe, NMUe, ee, eP, emu)
    omp_threads = 1
    call integrate_sab_el(t, E_grid, e_bins, SCATT_TYPE_LEGENDRE, lorder, el)
    call integrate_sab_inel(t, E_grid, e_bins, SCATT_TYPE_LEGENDRE, lorder, inel)
    call combine_sab_grid(el, inel, sm)
    scatt_mat = sm
    deallocate(t)
  end subroutine ref_calc_scattsab

  ! sab_egrid, sab.F90:460-568; Ein_out sized by the caller, n returned
  subroutine ref_sab_egrid(thr_inel, thr_el, NEi, NEo, NMU, mode, ei, sig, e_out, mu, &
                           cptr, ce_out, cpdf, cmu, el_mode, NEe, NMUe, ee, eP, emu, &
                           e_bins, nb, Ein_out, ncap, n) bind(C, name="ref_sab_egrid")
    real(c_double), value :: thr_inel, thr_el
    integer(c_int), value :: NEi, NEo, NMU, mode, el_mode, NEe, NMUe, nb, ncap
    real(c_double), intent(in) :: ei(*), sig(*), e_out(*), mu(*), ce_out(*), cpdf(*), cmu(*)
    integer(c_int), intent(in) :: cptr(*)
    real(c_double), intent(in) :: ee(*), eP(*), emu(*), e_bins(nb)
    real(c_double), intent(out) :: Ein_out(ncap)
    integer(c_int), intent(out) :: n
    type(SAlphaBeta), pointer :: t
    real(8), allocatable :: Ein(:)
    call build_sab(t, thr_inel, thr_el, NEi, NEo, NMU, mode, ei, sig, e_out, mu, cptr, ce_out, &
                   cpdf, cmu, el_mode, NEe, NMUe, ee, eP, emu)
    call sab_egrid(t, e_bins, Ein)
    n = size(Ein)
    if (n <= ncap) Ein_out(1:n) = Ein
    deallocate(t)
  end subroutine ref_sab_egrid


  ! scatt.F90:786-818 apply_tol_scatt (in place)
  subroutine ref_apply_tol_scatt(L, G, n, data, tol) bind(C, name="ref_apply_tol_scatt")
    integer(c_int), value :: L, G, n
    real(c_double), intent(inout) :: data(L, G, n)
    real(c_double), value :: tol
    real(8), allocatable :: d(:,:,:)
    allocate(d(L, G, n))
    d = data
    call apply_tol_scatt(d, tol)
    data = d
  end subroutine ref_apply_tol_scatt


  ! calc_chi (chi.F90:21-169) on a fissionable nuclide built from flat arrays.
  ! Prompt spectra: n_rxn fission reactions, reaction r has nnest(r) nested
  ! energy distributions (edist % next chain); spectrum s (in chain order) has
  ! law(s), data = sdata(sptr(s)+1 : sptr(s+1)).  Reaction r: MT mts(r),
  ! threshold thr(r), sigma = sig(sigptr(r)+1 : sigptr(r+1)).  Delayed spectra:
  ! n_prec precursor groups, law dlaw(j), data ddata(dptr(j)+1 : dptr(j+1)).
  ! Outputs sized by the caller for ncap incoming energies; nE returned.
  subroutine ref_calc_chi(n_grid, energy, fission_xs, nu_t_type, n_nu_t, nu_t_data, &
                          nu_d_type, n_nu_d, nu_d_data, n_prec, n_pd, prec_data, &
                          n_rxn, mts, thr, sigptr, sig, nnest, law, sptr, sdata, &
                          dlaw, dptr, ddata, e_bins, nb, ncap, nE, E_grid, chi_t, chi_p, chi_d) &
      bind(C, name="ref_calc_chi")
    integer(c_int), value :: n_grid, nu_t_type, n_nu_t, nu_d_type, n_nu_d, n_prec, n_pd
    integer(c_int), value :: n_rxn, nb, ncap
    real(c_double), intent(in) :: energy(n_grid), fission_xs(n_grid), nu_t_data(*), nu_d_data(*)
    real(c_double), intent(in) :: prec_data(*), sig(*), sdata(*), ddata(*), e_bins(nb)
    integer(c_int), intent(in) :: mts(*), thr(*), sigptr(*), nnest(*), law(*), sptr(*), dlaw(*), dptr(*)
    integer(c_int), intent(out) :: nE
    real(c_double), intent(out) :: E_grid(ncap), chi_t(nb - 1, ncap), chi_p(nb - 1, ncap)
    real(c_double), intent(out) :: chi_d(nb - 1, ncap, max(n_prec, 1))
    type(Nuclide), pointer :: nuc
    type(DistEnergy), pointer :: ed, prev
    real(8), allocatable :: Eg(:), ct(:,:), cp(:,:), cd(:,:,:)
    integer :: r, s, k, j
    allocate(nuc)
    nuc % name = 'fiss.00c'; nuc % awr = 233.0_8; nuc % kT = 2.53E-8_8
    nuc % n_grid = n_grid
    allocate(nuc % energy(n_grid), nuc % fission(n_grid))
    nuc % energy = energy; nuc % fission = fission_xs
    nuc % fissionable = .true.; nuc % n_fission = n_rxn
    nuc % nu_t_type = nu_t_type; nuc % nu_d_type = nu_d_type; nuc % nu_p_type = NU_NONE
    allocate(nuc % nu_t_data(n_nu_t)); nuc % nu_t_data = nu_t_data(1:n_nu_t)
    allocate(nuc % nu_d_data(max(n_nu_d, 1)))
    if (n_nu_d > 0) nuc % nu_d_data = nu_d_data(1:n_nu_d)
    nuc % n_precursor = n_prec
    allocate(nuc % nu_d_precursor_data(max(n_pd, 1)))
    if (n_pd > 0) nuc % nu_d_precursor_data = prec_data(1:n_pd)
    nuc % n_reaction = n_rxn
    allocate(nuc % reactions(n_rxn), nuc % index_fission(n_rxn))
    s = 0
    do r = 1, n_rxn
      nuc % index_fission(r) = r
      nuc % reactions(r) % MT = mts(r)
      nuc % reactions(r) % threshold = thr(r)
      nuc % reactions(r) % has_energy_dist = .true.
      allocate(nuc % reactions(r) % sigma(sigptr(r + 1) - sigptr(r)))
      nuc % reactions(r) % sigma = sig(sigptr(r) + 1 : sigptr(r + 1))
      prev => null()
      do k = 1, nnest(r)
        s = s + 1
        allocate(ed)
        ed % law = law(s)
        allocate(ed % data(sptr(s + 1) - sptr(s)))
        ed % data = sdata(sptr(s) + 1 : sptr(s + 1))
        ed % p_valid % n_regions = 0
        ed % p_valid % n_pairs = 0
        if (k == 1) then
          nuc % reactions(r) % edist => ed
        else
          prev % next => ed
        end if
        prev => ed
      end do
    end do
    if (n_prec > 0) then
      allocate(nuc % nu_d_edist(n_prec))
      do j = 1, n_prec
        nuc % nu_d_edist(j) % law = dlaw(j)
        allocate(nuc % nu_d_edist(j) % data(dptr(j + 1) - dptr(j)))
        nuc % nu_d_edist(j) % data = ddata(dptr(j) + 1 : dptr(j + 1))
      end do
    end if
    call calc_chi(nuc, e_bins, Eg, ct, cp, cd)
    nE = size(Eg)
    if (nE <= ncap) then
      E_grid(1:nE) = Eg
      chi_t(:, 1:nE) = ct
      chi_p(:, 1:nE) = cp
      if (n_prec > 0) chi_d(:, 1:nE, 1:n_prec) = cd
    end if
  end subroutine ref_calc_chi

  ! ScattData%init (scattdata_header.F90:78) + %convert_distro (:325) on an
  ! in-memory reaction: MT, optional angular distribution (has_adist), optional
  ! energy distribution of ACE law `law` (0: none).  Returns the tables the way
  ! the C ABI lays them out: f is (M, sum NP), row_ptr 0-based offsets.
  subroutine ref_convert_distro(MT, law, has_adist, na, a_energy, a_type, a_loc, nad, a_data, &
                                ned, edata, nb, e_bins, M, thr_E, cap, is_init, NE, e_grid, &
                                row_ptr, eout, pdf, cdf, intt, f, sd_law, in_cm) &
      bind(C, name="ref_convert_distro")
    integer(c_int), value :: MT, law, has_adist, na, nad, ned, nb, M, cap
    real(c_double), intent(in) :: a_energy(*), a_data(*), edata(*)
    integer(c_int), intent(in) :: a_type(*), a_loc(*)
    real(c_double), intent(in), target :: e_bins(nb)
    real(c_double), value :: thr_E
    integer(c_int), intent(out) :: is_init, NE, row_ptr(*), intt(*), sd_law, in_cm
    real(c_double), intent(out) :: e_grid(*), eout(*), pdf(*), cdf(*), f(M, *)
    type(Nuclide), pointer :: nuc
    type(Reaction), pointer :: rxn
    type(DistEnergy), pointer :: ed
    type(ScattData) :: sd
    integer :: iE, np, o
    allocate(nuc)
    nuc % name = 'conv.00c'; nuc % awr = 236.0058_8; nuc % kT = 2.53E-8_8
    nuc % n_grid = 2
    allocate(nuc % energy(2)); nuc % energy = (/ thr_E, 20.0_8 /)
    nuc % freegas_cutoff = ZERO
    allocate(rxn)
    rxn % MT = MT; rxn % Q_value = ZERO; rxn % multiplicity = 1; rxn % threshold = 1
    rxn % scatter_in_cm = .false.
    rxn % has_angle_dist = (has_adist /= 0); rxn % has_energy_dist = (law /= 0)
    if (has_adist /= 0) then
      rxn % adist % n_energy = na
      allocate(rxn % adist % energy(na), rxn % adist % type(na), rxn % adist % location(na))
      allocate(rxn % adist % data(max(nad, 1)))
      rxn % adist % energy = a_energy(1:na); rxn % adist % type = a_type(1:na)
      rxn % adist % location = a_loc(1:na)
      rxn % adist % data = ZERO
      if (nad > 0) rxn % adist % data = a_data(1:nad)
    end if
    ed => null()
    if (law /= 0) then
      allocate(ed)
      ed % law = law
      allocate(ed % data(ned)); ed % data = edata(1:ned)
      ed % p_valid % n_regions = 0; ed % p_valid % n_pairs = 0
    end if
    is_init = 0; NE = 0; sd_law = -1; in_cm = 0
    call sd % init(nuc, rxn, ed, e_bins, SCATT_TYPE_LEGENDRE, 5, M)
    if (.not. sd % is_init) return
    is_init = 1
    sd_law = sd % law
    if (rxn % scatter_in_cm) in_cm = 1
    call sd % convert_distro()
    NE = sd % NE
    row_ptr(1) = 0
    o = 0
    do iE = 1, NE
      np = size(sd % distro(iE) % data, 2)
      if (o + np > cap) then
        NE = -1
        return
      end if
      e_grid(iE) = sd % E_grid(iE)
      f(:, o + 1 : o + np) = sd % distro(iE) % data
      intt(iE) = sd % INTT(iE)
      if (allocated(sd % pdfs(iE) % data)) then
        eout(o + 1 : o + np) = sd % Eouts(iE) % data
        pdf(o + 1 : o + np) = sd % pdfs(iE) % data
        cdf(o + 1 : o + np) = sd % cdfs(iE) % data
      else
        eout(o + 1 : o + np) = ZERO; pdf(o + 1 : o + np) = ZERO; cdf(o + 1 : o + np) = ZERO
      end if
      o = o + np
      row_ptr(iE + 1) = o
    end do
  end subroutine ref_convert_distro

  ! create_Ein_grid (scatt.F90:166-243) on hand-filled ScattData objects: only
  ! is_init, rxn%MT, rxn%Q_value, E_grid and E_bins are read by the builders.
  subroutine ref_create_ein_grid(n_sd, is_init, MT, Q, eg_ptr, eg, nb, e_bins, n_nuc, nuc_grid, &
                                 awr, kT, cutoff, thresh, cap, n_el, ein_el, n_inel, ein_inel) &
      bind(C, name="ref_create_ein_grid")
    integer(c_int), value :: n_sd, nb, n_nuc, cap
    integer(c_int), intent(in) :: is_init(n_sd), MT(n_sd), eg_ptr(n_sd + 1)
    real(c_double), intent(in) :: Q(n_sd), eg(*), nuc_grid(n_nuc)
    real(c_double), intent(in), target :: e_bins(nb)
    real(c_double), value :: awr, kT, cutoff, thresh
    integer(c_int), intent(out) :: n_el, n_inel
    real(c_double), intent(out) :: ein_el(cap), ein_inel(cap)
    type(ScattData), allocatable, target :: sds(:)
    type(Reaction), pointer :: rx(:)
    real(8), allocatable :: ng(:), el(:), inel(:)
    integer :: k
    allocate(sds(n_sd), rx(n_sd))
    do k = 1, n_sd
      rx(k) % MT = MT(k); rx(k) % Q_value = Q(k)
      sds(k) % is_init = (is_init(k) /= 0)
      sds(k) % rxn => rx(k)
      sds(k) % E_bins => e_bins
      sds(k) % NE = eg_ptr(k + 1) - eg_ptr(k)
      allocate(sds(k) % E_grid(sds(k) % NE))
      sds(k) % E_grid = eg(eg_ptr(k) + 1 : eg_ptr(k + 1))
    end do
    allocate(ng(n_nuc)); ng = nuc_grid
    call create_Ein_grid(sds, e_bins, ng, awr, kT, cutoff, thresh, el, inel)
    n_el = size(el); n_inel = 0
    if (n_el <= cap) ein_el(1:n_el) = el
    if (allocated(inel)) then
      n_inel = size(inel)
      if (n_inel <= cap) ein_inel(1:n_inel) = inel
    end if
  end subroutine ref_create_ein_grid

  ! calc_scatt (scatt.F90:33) on a nuclide unpacked from the flat (I, D) encoding of
  ! tests/synth.py:pack_nuclide.  Matrices come back in Fortran order (L, G, n).
  subroutine ref_calc_scatt(I, D, nb, e_bins, order, mu_bins, nuscatt, cap, n_el, ein_el, &
                            n_inel, ein_inel, el_mat, inel_mat, nuinel_mat) &
      bind(C, name="ref_calc_scatt")
    integer(c_int), intent(in) :: I(*)
    real(c_double), intent(in) :: D(*)
    integer(c_int), value :: nb, order, mu_bins, nuscatt, cap
    real(c_double), intent(in) :: e_bins(nb)
    integer(c_int), intent(out) :: n_el, n_inel
    real(c_double), intent(out) :: ein_el(cap), ein_inel(cap)
    real(c_double), intent(out) :: el_mat(order + 1, nb - 1, cap), inel_mat(order + 1, nb - 1, cap)
    real(c_double), intent(out) :: nuinel_mat(order + 1, nb - 1, cap)
    type(Nuclide), pointer :: nuc
    type(DistEnergy), pointer :: ed, prev
    real(8), allocatable :: Eel(:), Ein(:), el(:,:,:), inel(:,:,:), nuin(:,:,:)
    integer :: ip, dp_, r, k, n_grid, n_rxn, ns, na, nad, ne, nme, nd, npv, ord
    ip = 1; dp_ = 1
    n_grid = I(ip); n_rxn = I(ip + 1); ip = ip + 2
    allocate(nuc)
    nuc % name = 'flat.00c'; nuc % zaid = 8016
    nuc % awr = D(1); nuc % kT = D(2); nuc % freegas_cutoff = D(3); dp_ = 4
    nuc % n_grid = n_grid
    allocate(nuc % energy(n_grid), nuc % elastic(n_grid))
    nuc % energy = D(dp_ : dp_ + n_grid - 1); dp_ = dp_ + n_grid
    nuc % elastic = D(dp_ : dp_ + n_grid - 1); dp_ = dp_ + n_grid
    nuc % n_reaction = n_rxn
    allocate(nuc % reactions(n_rxn))
    do r = 1, n_rxn
      associate (rx => nuc % reactions(r))
        rx % MT = I(ip); rx % multiplicity = I(ip + 1); rx % threshold = I(ip + 2)
        rx % scatter_in_cm = (I(ip + 3) /= 0)
        ns = I(ip + 4); rx % has_angle_dist = (I(ip + 5) /= 0)
        na = I(ip + 6); nad = I(ip + 7); ne = I(ip + 8); nme = I(ip + 9); ip = ip + 10
        rx % Q_value = D(dp_); dp_ = dp_ + 1
        allocate(rx % sigma(max(ns, 1))); rx % sigma = ZERO
        if (ns > 0) rx % sigma = D(dp_ : dp_ + ns - 1)
        dp_ = dp_ + ns
        if (rx % has_angle_dist) then
          rx % adist % n_energy = na
          allocate(rx % adist % energy(na), rx % adist % type(na), rx % adist % location(na))
          allocate(rx % adist % data(nad))
          rx % adist % type = I(ip : ip + na - 1); ip = ip + na
          rx % adist % location = I(ip : ip + na - 1); ip = ip + na
          rx % adist % energy = D(dp_ : dp_ + na - 1); dp_ = dp_ + na
          rx % adist % data = D(dp_ : dp_ + nad - 1); dp_ = dp_ + nad
        end if
        rx % multiplicity_with_E = (nme > 0)
        if (nme > 0) then
          allocate(rx % multiplicity_E)
          rx % multiplicity_E % n_regions = 0; rx % multiplicity_E % n_pairs = nme
          allocate(rx % multiplicity_E % x(nme), rx % multiplicity_E % y(nme))
          rx % multiplicity_E % x = D(dp_ : dp_ + nme - 1); dp_ = dp_ + nme
          rx % multiplicity_E % y = D(dp_ : dp_ + nme - 1); dp_ = dp_ + nme
        end if
        rx % has_energy_dist = (ne > 0)
        prev => null()
        do k = 1, ne
          allocate(ed)
          ed % law = I(ip); nd = I(ip + 1); npv = I(ip + 2); ip = ip + 3
          allocate(ed % data(nd)); ed % data = D(dp_ : dp_ + nd - 1); dp_ = dp_ + nd
          ed % p_valid % n_regions = 0; ed % p_valid % n_pairs = npv
          if (npv > 0) then
            allocate(ed % p_valid % x(npv), ed % p_valid % y(npv))
            ed % p_valid % x = D(dp_ : dp_ + npv - 1); dp_ = dp_ + npv
            ed % p_valid % y = D(dp_ : dp_ + npv - 1); dp_ = dp_ + npv
          end if
          if (k == 1) then
            rx % edist => ed
          else
            prev % next => ed
          end if
          prev => ed
        end do
      end associate
    end do
    ord = order
    call calc_scatt(nuc, e_bins, SCATT_TYPE_LEGENDRE, ord, mu_bins, nuscatt /= 0, Eel, Ein, el, inel, nuin)
    n_el = size(Eel); n_inel = 0
    if (n_el > cap) return
    ein_el(1:n_el) = Eel; el_mat(:, :, 1:n_el) = el
    if (allocated(Ein)) then
      n_inel = size(Ein)
      if (n_inel > cap) return
      ein_inel(1:n_inel) = Ein; inel_mat(:, :, 1:n_inel) = inel
      if (nuscatt /= 0) nuinel_mat(:, :, 1:n_inel) = nuin
    end if
  end subroutine ref_calc_scatt

  ! print_scatt_bin (scatt.F90:1139) and print_chi_bin (chi.F90:319) into a stream file
  ! opened the way ndpp.F90:1309-1311 opens it.  Matrices arrive in Fortran order.
  subroutine ref_print_scatt_bin(path, plen, L, G, n_el, gi_el, ein_el, el_mat, n_inel, gi_inel, &
                                 ein_inel, inel_mat, with_nu, nuinel_mat) &
      bind(C, name="ref_print_scatt_bin")
    integer(c_int), value :: plen, L, G, n_el, n_inel, with_nu
    character(kind=c_char), intent(in) :: path(plen)
    integer(c_int), intent(in) :: gi_el(G + 1), gi_inel(G + 1)
    real(c_double), intent(in) :: ein_el(n_el), el_mat(L, G, n_el), ein_inel(max(n_inel, 1))
    real(c_double), intent(in) :: inel_mat(L, G, max(n_inel, 1)), nuinel_mat(L, G, max(n_inel, 1))
    character(len=plen) :: fname
    real(8), allocatable :: Eel(:), Ein(:), el(:,:,:), inel(:,:,:), nuin(:,:,:)
    integer :: k
    do k = 1, plen
      fname(k:k) = path(k)
    end do
    allocate(Eel(n_el), el(L, G, n_el))
    Eel = ein_el; el = el_mat
    if (n_inel > 0) then
      allocate(Ein(n_inel), inel(L, G, n_inel), nuin(L, G, n_inel))
      Ein = ein_inel; inel = inel_mat; nuin = nuinel_mat
    end if
    open(FILE=fname, UNIT=UNIT_NUC, STATUS='replace', ACTION='write', ACCESS='stream')
    if (n_inel > 0 .and. with_nu /= 0) then
      call print_scatt_bin(gi_el, gi_inel, Eel, Ein, el, inel, nuin)
    else
      call print_scatt_bin(gi_el, gi_inel, Eel, Ein, el, inel)
    end if
    close(UNIT_NUC)
  end subroutine ref_print_scatt_bin

  subroutine ref_print_chi_bin(path, plen, G, NE, nprec, e_grid, chi_t, chi_p, chi_d) &
      bind(C, name="ref_print_chi_bin")
    integer(c_int), value :: plen, G, NE, nprec
    character(kind=c_char), intent(in) :: path(plen)
    real(c_double), intent(in) :: e_grid(NE), chi_t(G, NE), chi_p(G, NE), chi_d(G, NE, max(nprec, 1))
    character(len=plen) :: fname
    real(8), allocatable :: Eg(:), ct(:,:), cp(:,:), cd(:,:,:)
    integer :: k
    do k = 1, plen
      fname(k:k) = path(k)
    end do
    allocate(Eg(NE), ct(G, NE), cp(G, NE), cd(G, NE, nprec))
    Eg = e_grid; ct = chi_t; cp = chi_p
    if (nprec > 0) cd = chi_d(:, :, 1:nprec)
    open(FILE=fname, UNIT=UNIT_NUC, STATUS='replace', ACTION='write', ACCESS='stream')
    call print_chi_bin(Eg, ct, cp, cd)
    close(UNIT_NUC)
  end subroutine ref_print_chi_bin

  ! thin_grid (thin.F90:17) in its one / two / three array forms (mode = 1, 2, 3)
  subroutine ref_thin_grid(mode, L, G, n, x, y, y2, y3, nk, tokeep, tol, n_out, compression, maxerr) &
      bind(C, name="ref_thin_grid")
    integer(c_int), value :: mode, L, G, n, nk
    real(c_double), intent(inout) :: x(n), y(L, G, n), y2(L, G, n), y3(n)
    real(c_double), intent(in) :: tokeep(nk)
    real(c_double), value :: tol
    integer(c_int), intent(out) :: n_out
    real(c_double), intent(out) :: compression, maxerr
    real(8), allocatable :: xa(:), ya(:,:,:), y2a(:,:,:), y3a(:), tk(:)
    allocate(xa(n), ya(L, G, n), tk(nk))
    xa = x; ya = y; tk = tokeep
    compression = ZERO; maxerr = ZERO
    if (mode == 1) then
      call thin_grid(xa, ya, tk, tol, compression, maxerr)
    else if (mode == 2) then
      allocate(y2a(L, G, n)); y2a = y2
      call thin_grid(xa, ya, tk, tol, compression, maxerr, y2a)
    else
      allocate(y2a(L, G, n), y3a(n)); y2a = y2; y3a = y3
      call thin_grid(xa, ya, tk, tol, compression, maxerr, y2a, y3a)
    end if
    n_out = size(xa)
    x(1:n_out) = xa; y(:, :, 1:n_out) = ya
    if (mode >= 2) y2(:, :, 1:n_out) = y2a
    if (mode == 3) y3(1:n_out) = y3a
  end subroutine ref_thin_grid

  ! the formatted writers: to_str (string.F90:408), print_ascii_array (output.F90:221),
  ! print_scatt_ascii (scatt.F90:881), print_chi_ascii (chi.F90:203), into a text file
  ! opened the way ndpp.F90:1285 opens it
  subroutine ref_to_str(x, out, n) bind(C, name="ref_to_str")
    real(c_double), value :: x
    character(kind=c_char), intent(out) :: out(15)
    integer(c_int), intent(out) :: n
    character(15) :: s
    integer :: k
    s = to_str(x)
    n = len_trim(s)
    do k = 1, 15
      out(k) = s(k:k)
    end do
  end subroutine ref_to_str

  subroutine ref_print_ascii_array(path, plen, n, a) bind(C, name="ref_print_ascii_array")
    integer(c_int), value :: plen, n
    character(kind=c_char), intent(in) :: path(plen)
    real(c_double), intent(in) :: a(n)
    character(len=plen) :: fname
    integer :: k
    do k = 1, plen
      fname(k:k) = path(k)
    end do
    open(FILE=fname, UNIT=UNIT_NUC, STATUS='replace', ACTION='write')
    call print_ascii_array(a, UNIT_NUC)
    close(UNIT_NUC)
  end subroutine ref_print_ascii_array

  subroutine ref_print_scatt_ascii(path, plen, L, G, n_el, gi_el, ein_el, el_mat, n_inel, gi_inel, &
                                   ein_inel, inel_mat, with_nu, nuinel_mat) &
      bind(C, name="ref_print_scatt_ascii")
    integer(c_int), value :: plen, L, G, n_el, n_inel, with_nu
    character(kind=c_char), intent(in) :: path(plen)
    integer(c_int), intent(in) :: gi_el(G + 1), gi_inel(G + 1)
    real(c_double), intent(in) :: ein_el(n_el), el_mat(L, G, n_el), ein_inel(max(n_inel, 1))
    real(c_double), intent(in) :: inel_mat(L, G, max(n_inel, 1)), nuinel_mat(L, G, max(n_inel, 1))
    character(len=plen) :: fname
    real(8), allocatable :: Eel(:), Ein(:), el(:,:,:), inel(:,:,:), nuin(:,:,:)
    integer :: k
    do k = 1, plen
      fname(k:k) = path(k)
    end do
    allocate(Eel(n_el), el(L, G, n_el))
    Eel = ein_el; el = el_mat
    if (n_inel > 0) then
      allocate(Ein(n_inel), inel(L, G, n_inel), nuin(L, G, n_inel))
      Ein = ein_inel; inel = inel_mat; nuin = nuinel_mat
    end if
    open(FILE=fname, UNIT=UNIT_NUC, STATUS='replace', ACTION='write')
    if (n_inel > 0 .and. with_nu /= 0) then
      call print_scatt_ascii(gi_el, gi_inel, Eel, Ein, el, inel, nuin)
    else
      call print_scatt_ascii(gi_el, gi_inel, Eel, Ein, el, inel)
    end if
    close(UNIT_NUC)
  end subroutine ref_print_scatt_ascii

  subroutine ref_print_chi_ascii(path, plen, G, NE, nprec, e_grid, chi_t, chi_p, chi_d) &
      bind(C, name="ref_print_chi_ascii")
    integer(c_int), value :: plen, G, NE, nprec
    character(kind=c_char), intent(in) :: path(plen)
    real(c_double), intent(in) :: e_grid(NE), chi_t(G, NE), chi_p(G, NE), chi_d(G, NE, max(nprec, 1))
    character(len=plen) :: fname
    real(8), allocatable :: Eg(:), ct(:,:), cp(:,:), cd(:,:,:)
    integer :: k
    do k = 1, plen
      fname(k:k) = path(k)
    end do
    allocate(Eg(NE), ct(G, NE), cp(G, NE), cd(G, NE, nprec))
    Eg = e_grid; ct = chi_t; cp = chi_p
    if (nprec > 0) cd = chi_d(:, :, 1:nprec)
    open(FILE=fname, UNIT=UNIT_NUC, STATUS='replace', ACTION='write')
    call print_chi_ascii(Eg, ct, cp, cd)
    close(UNIT_NUC)
  end subroutine ref_print_chi_ascii

end module ref_shim
