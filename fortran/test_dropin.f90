!===============================================================================
! test_dropin -- end-to-end drop-in check, Fortran to Fortran.
!
! Builds an H-1-like Nuclide + elastic ScattData in memory (the way the
! reference's own tests/test_scatt does: "fake ACE"), then computes the elastic
! scattering-moment matrix twice:
!   (1) with the reference's calc_elastic_grid (scatt.F90:603), CPU;
!   (2) with calc_elastic_grid_hip (fortran/ndpp_hip_mod.f90) -> libndpp_hip.so.
! Prints the maximum scale-aware difference; exit code 0 iff < 1e-10.
! TEST INFRASTRUCTURE: links the reference objects of oracle/_ref/build.
!===============================================================================
program test_dropin
  use ace_header
  use constants
  use global
  use scatt,            only: calc_elastic_grid, calc_inelastic_grid
  use scattdata_header, only: ScattData
  use ndpp_hip_mod
  implicit none

  type(Nuclide), pointer  :: nuc
  type(ScattData), allocatable, target :: rxn_data(:)
  type(DistEnergy), pointer :: edist_null => null()
  real(8), allocatable, target :: E_bins(:)
  real(8), allocatable :: Ein(:), mu_out(:), ref_mat(:,:,:), hip_mat(:,:,:)
  integer :: i, k, NE, order, mu_bins, ierr, n_grid
  real(8) :: err, scale, worst, dmu, mu
  logical :: ref_only
  character(len=8) :: envv

  ! NDPP_DROPIN_REF_ONLY=1: run the reference side only (sanity check of the
  ! in-memory nuclides on a machine without a GPU); always exits with code 4.
  call get_environment_variable("NDPP_DROPIN_REF_ONLY", envv)
  ref_only = (trim(envv) == "1")

  ! module global's numerics: defaults of constants.F90:70-100
  SAB_THRESHOLD = SAB_THRESH_DEFAULT;  BRENT_MU_THRESH = BRENT_MU_THRESH_DEFAULT
  ADAPTIVE_MU_TOL = ADAPTIVE_MU_TOL_DEFAULT;  ADAPTIVE_MU_ITS = ADAPTIVE_MU_ITS_DEFAULT
  ADAPTIVE_EOUT_TOL = ADAPTIVE_EOUT_TOL_DEFAULT;  ADAPTIVE_EOUT_ITS = ADAPTIVE_EOUT_ITS_DEFAULT
  NE_PER_GRP = NE_PER_GRP_DEFAULT;  SAB_EPTS_PER_BIN = SAB_EPTS_PER_BIN_DEFAULT
  EXTEND_PTS = EXTEND_PTS_DEFAULT;  INEL_EXTEND_PTS = INEL_EXTEND_PTS_DEFAULT
  omp_threads = 1
  verbosity = 0

  order = 3          ! scatt_order: P3 -> 4 moments
  mu_bins = 2001
  allocate(E_bins(3)); E_bins = (/ 0.0_8, 6.25E-7_8, 20.0_8 /)

  ! ---- "fake ACE" H-1: smooth elastic cross section on a 40-point log grid
  allocate(nuc)
  nuc % name = '1001.71c'; nuc % zaid = 1001
  nuc % awr = 0.999167_8;  nuc % kT = 2.5301E-8_8
  n_grid = 40;  nuc % n_grid = n_grid
  allocate(nuc % energy(n_grid), nuc % elastic(n_grid))
  do i = 1, n_grid
    nuc % energy(i) = 1.0E-11_8 * (20.0_8 / 1.0E-11_8) ** (real(i - 1, 8) / real(n_grid - 1, 8))
    nuc % elastic(i) = 20.0_8 / (ONE + nuc % energy(i))
  end do
  nuc % freegas_cutoff = FREEGAS_THRESHOLD_DEFAULT * nuc % kT
  nuc % n_reaction = 1
  allocate(nuc % reactions(1))
  associate (rxn => nuc % reactions(1))
    rxn % MT = ELASTIC;  rxn % Q_value = ZERO;  rxn % multiplicity = 1
    rxn % threshold = 1; rxn % scatter_in_cm = .true.
    rxn % has_angle_dist = .true.;  rxn % has_energy_dist = .false.
    ! three tabulated incoming energies; the tables themselves are filled below
    rxn % adist % n_energy = 3
    allocate(rxn % adist % energy(3), rxn % adist % type(3), rxn % adist % location(3))
    allocate(rxn % adist % data(1))
    rxn % adist % energy = (/ 1.0E-11_8, 1.0E-6_8, 20.0_8 /)
    rxn % adist % type = ANGLE_ISOTROPIC;  rxn % adist % location = 0
    rxn % adist % data = ZERO
  end associate

  ! ---- ScattData as calc_scatt does (scatt.F90:95), then f(mu) rows directly
  allocate(rxn_data(1))
  call rxn_data(1) % init(nuc, nuc % reactions(1), edist_null, E_bins, &
                          SCATT_TYPE_LEGENDRE, order, mu_bins)
  if (.not. rxn_data(1) % is_init) stop 2
  dmu = TWO / real(mu_bins - 1, 8)
  do i = 1, mu_bins
    mu = rxn_data(1) % mu(i)
    rxn_data(1) % distro(1) % data(i, 1) = 0.5_8
    rxn_data(1) % distro(2) % data(i, 1) = 0.5_8 * (ONE + 0.1_8 * mu)
    rxn_data(1) % distro(3) % data(i, 1) = 0.5_8 * (ONE + 0.3_8 * mu)
  end do

  ! ---- incoming grid: free-gas region, above the cutoff, and the extra top point
  NE = 12
  allocate(Ein(NE), mu_out(mu_bins))
  Ein = (/ 1.0E-11_8, 3.0E-10_8, 2.53E-8_8, 6.2499E-7_8, 6.2501E-7_8, 5.0E-6_8, &
           1.01E-5_8, 2.0E-5_8, 1.0E-3_8, 1.5_8, 20.0_8, 20.0_8 * (ONE + 1.0E-3) /)
  mu_out = rxn_data(1) % mu

  call calc_elastic_grid(nuc, mu_out, rxn_data, Ein, order + 1, E_bins, ref_mat)
  if (ref_only) then
    write(*,'(A,4ES14.6)') ' reference elastic, Ein(3): ', ref_mat(1:4, 1, 3)
    call inelastic_part(worst)
    stop 4
  end if
  call calc_elastic_grid_hip(nuc, mu_out, rxn_data, Ein, order + 1, E_bins, hip_mat, ierr)
  if (ierr /= 0) then
    write(*,*) 'libndpp_hip error ', ierr, ': ', trim(ndpp_hip_error())
    stop 3
  end if

  worst = ZERO
  do k = 1, NE
    scale = maxval(abs(ref_mat(:, :, k)))
    if (scale == ZERO) scale = ONE
    err = maxval(abs(hip_mat(:, :, k) - ref_mat(:, :, k))) / scale
    worst = max(worst, err)
    write(*,'(A,ES12.5,A,ES10.3,A,4ES14.6)') ' Ein=', Ein(k), '  err=', err, &
          '  P0..3(g=1)=', hip_mat(1:4, 1, k)
  end do
  write(*,'(A,ES10.3)') ' drop-in check: worst scale-relative difference = ', worst
  if (worst >= 1.0E-10_8) then
    write(*,*) 'FAIL'
    stop 1
  end if

  call inelastic_part(worst)
  write(*,'(A,ES10.3)') ' drop-in check (inelastic + nu-inelastic): worst scale-relative difference = ', worst
  if (worst < 1.0E-10_8) then
    write(*,*) 'PASS'
  else
    write(*,*) 'FAIL'
    stop 1
  end if

contains

  !=============================================================================
  ! A U-238-like nuclide with the four non-elastic integrator families
  ! (SURVEY.md 8d config 3 in miniature): MT 51 level inelastic (file4 CM, Q<0),
  ! MT 91 continuum in the CM frame (Kalbach-Mann-shaped table: unitbase +
  ! integrate_file6_cm_leg), MT 16 (n,2n) evaporation law 9 with an angular
  ! table (multiplicity 2), MT 22 lab-frame tabular law (unitbase +
  ! integrate_file6_lab_leg).  The reference's calc_inelastic_grid against
  ! calc_inelastic_grid_hip, with nu-scatter.
  !=============================================================================
  subroutine inelastic_part(worst_out)
    real(8), intent(out) :: worst_out
    type(Nuclide), pointer :: hv
    type(ScattData), allocatable, target :: sds(:)
    type(DistEnergy), pointer :: ed91, ed16, ed22
    real(8), allocatable, target :: bins(:)
    real(8), allocatable :: Ei(:), ref_in(:,:,:), ref_nu(:,:,:), hip_in(:,:,:), hip_nu(:,:,:), mo(:)
    integer :: n_grid, ii, kk, M, ord, NEi, ier, thr
    real(8) :: e, sc, er

    M = 257;  ord = 7           ! P7 -> 8 moments
    allocate(bins(6)); bins = (/ 0.0_8, 1.0E-3_8, 0.05_8, 0.5_8, 3.0_8, 20.0_8 /)
    allocate(hv)
    hv % name = '92238.71c'; hv % zaid = 92238; hv % awr = 236.0058_8; hv % kT = 2.53E-8_8
    n_grid = 60; hv % n_grid = n_grid
    allocate(hv % energy(n_grid), hv % elastic(n_grid))
    do ii = 1, n_grid
      hv % energy(ii) = 1.0E-5_8 * (20.0_8 / 1.0E-5_8) ** (real(ii - 1, 8) / real(n_grid - 1, 8))
      hv % elastic(ii) = 9.0_8
    end do
    hv % freegas_cutoff = ZERO
    hv % n_reaction = 4
    allocate(hv % reactions(4))
    thr = 34                      ! threshold index on the nuclide grid (~0.04 MeV)

    call set_rxn(hv % reactions(1), 51, -0.035_8, 1, thr, .true.)
    call set_rxn(hv % reactions(2), 91, -1.0_8, 1, 46, .true.)
    call set_rxn(hv % reactions(3), 16, -6.0_8, 2, 54, .false.)
    call set_rxn(hv % reactions(4), 22, -2.0_8, 1, 50, .false.)
    do ii = 1, 4
      associate (r => hv % reactions(ii))
        allocate(r % sigma(n_grid - r % threshold + 1))
        do kk = 1, size(r % sigma)
          e = hv % energy(r % threshold + kk - 1) - hv % energy(r % threshold)
          r % sigma(kk) = (0.5_8 + 0.2_8 * ii) * (ONE - exp(-e / 0.3_8))
        end do
      end associate
    end do

    allocate(sds(4))
    ! MT 51: adist only -> integrate_file4_cm_leg, two tabulated energies
    call manual_sd(sds(1), hv, hv % reactions(1), bins, ord, M, 0, 2, &
                   (/ hv % energy(thr), 20.0_8 /), .true., .false.)
    ! MT 91: CM continuum, law 44, five incoming energies
    allocate(ed91); call set_edist(ed91, 44, 2)
    hv % reactions(2) % edist => ed91
    call manual_sd(sds(2), hv, hv % reactions(2), bins, ord, M, 44, 5, &
                   (/ 1.0_8, 2.5_8, 6.0_8, 12.0_8, 20.0_8 /), .false., .true.)
    ! MT 16: law 9 + adist, lab
    allocate(ed16); call set_edist(ed16, 9, 1)
    deallocate(ed16 % data); allocate(ed16 % data(11))
    ed16 % data = (/ 0.0_8, 4.0_8, 6.0_8, 9.0_8, 14.0_8, 20.0_8, 0.45_8, 0.6_8, 0.8_8, 1.0_8, 6.02_8 /)
    hv % reactions(3) % edist => ed16
    call manual_sd(sds(3), hv, hv % reactions(3), bins, ord, M, 9, 4, &
                   (/ 6.0_8, 9.0_8, 14.0_8, 20.0_8 /), .true., .true.)
    ! MT 22: lab tabular energy-angle table without adist
    allocate(ed22); call set_edist(ed22, 61, 2)
    hv % reactions(4) % edist => ed22
    call manual_sd(sds(4), hv, hv % reactions(4), bins, ord, M, 61, 4, &
                   (/ 2.0_8, 5.0_8, 10.0_8, 20.0_8 /), .false., .true.)

    NEi = 16
    allocate(Ei(NEi), mo(M))
    Ei = (/ 0.03_8, 0.0400001_8, 0.05_8, 0.2_8, 0.9_8, 1.2_8, 2.2_8, 3.0_8, 5.5_8, 6.5_8, 8.0_8, &
            11.0_8, 15.0_8, 19.5_8, 20.0_8, 20.0_8 * (ONE + 1.0E-3) /)
    mo = sds(1) % mu
    call calc_inelastic_grid(hv, mo, sds, Ei, ord + 1, bins, .true., ref_in, ref_nu)
    if (ref_only) then
      do kk = 1, NEi
        write(*,'(A,ES12.5,A,3ES13.5,A,ES13.5)') ' Ein=', Ei(kk), '  reference sum_g P0,P1,P2 =', &
          sum(ref_in(1, :, kk)), sum(ref_in(2, :, kk)), sum(ref_in(3, :, kk)), '  nu:', sum(ref_nu(1, :, kk))
      end do
      worst_out = ZERO
      return
    end if
    call calc_inelastic_grid_hip(hv, mo, sds, Ei, ord + 1, bins, .true., hip_in, hip_nu, ier)
    if (ier /= 0) then
      write(*,*) 'libndpp_hip error ', ier, ': ', trim(ndpp_hip_error())
      stop 3
    end if
    worst_out = ZERO
    do kk = 1, NEi
      sc = max(maxval(abs(ref_in(:, :, kk))), maxval(abs(ref_nu(:, :, kk))))
      if (sc == ZERO) sc = ONE
      er = max(maxval(abs(hip_in(:, :, kk) - ref_in(:, :, kk))), &
               maxval(abs(hip_nu(:, :, kk) - ref_nu(:, :, kk)))) / sc
      worst_out = max(worst_out, er)
      write(*,'(A,ES12.5,A,ES10.3,A,3ES13.5,A,ES13.5)') ' Ein=', Ei(kk), '  err=', er, &
            '  sum_g P0,P1,P2 =', sum(hip_in(1, :, kk)), sum(hip_in(2, :, kk)), &
            sum(hip_in(3, :, kk)), '  nu:', sum(hip_nu(1, :, kk))
    end do
  end subroutine inelastic_part

  subroutine set_rxn(r, MT, Q, mult, threshold, in_cm)
    type(Reaction), intent(inout) :: r
    integer, intent(in) :: MT, mult, threshold
    real(8), intent(in) :: Q
    logical, intent(in) :: in_cm
    r % MT = MT; r % Q_value = Q; r % multiplicity = mult; r % threshold = threshold
    r % scatter_in_cm = in_cm; r % multiplicity_with_E = .false.
    r % has_angle_dist = .false.; r % has_energy_dist = .false.
  end subroutine set_rxn

  subroutine set_edist(ed, law, npv)
    type(DistEnergy), intent(inout) :: ed
    integer, intent(in) :: law, npv
    ed % law = law
    allocate(ed % data(4)); ed % data = ZERO
    ed % p_valid % n_regions = 0
    ed % p_valid % n_pairs = npv
    allocate(ed % p_valid % x(npv), ed % p_valid % y(npv))
    if (npv == 1) then
      ed % p_valid % x = (/ 1.0E-5_8 /); ed % p_valid % y = (/ ONE /)
    else
      ed % p_valid % x = (/ 1.0E-5_8, 20.0_8 /); ed % p_valid % y = (/ ONE, 0.8_8 /)
    end if
  end subroutine set_edist

  ! Fill a ScattData by hand (all components are public, scattdata_header.F90:36-69):
  ! ne_ tabulated incoming energies eg_; with_adist: one f(mu) column per row;
  ! otherwise 5..9 outgoing energies per row with Kalbach-Mann-shaped columns.
  subroutine manual_sd(sd, nu, r, bins_, ord_, M_, law, ne_, eg_, with_adist, with_edist)
    type(ScattData), intent(inout) :: sd
    type(Nuclide), pointer, intent(in) :: nu
    type(Reaction), target, intent(inout) :: r
    real(8), target, intent(in) :: bins_(:)
    integer, intent(in) :: ord_, M_, law, ne_
    real(8), intent(in) :: eg_(:)
    logical, intent(in) :: with_adist, with_edist
    integer :: k_, j_, i_, np_
    real(8) :: dmu_, mu_, A_, R_, emax_, x_, nrm_
    sd % is_init = .true.; sd % NE = ne_
    allocate(sd % E_grid(ne_)); sd % E_grid = eg_(1:ne_)
    allocate(sd % distro(ne_), sd % Eouts(ne_), sd % pdfs(ne_), sd % cdfs(ne_), sd % INTT(ne_))
    allocate(sd % mu(M_))
    dmu_ = TWO / real(M_ - 1, 8)
    do i_ = 1, M_ - 1
      sd % mu(i_) = -ONE + real(i_ - 1, 8) * dmu_
    end do
    sd % mu(M_) = ONE
    sd % E_bins => bins_; sd % groups = size(bins_) - 1
    sd % scatt_type = SCATT_TYPE_LEGENDRE; sd % order = ord_ + 1
    sd % awr = nu % awr; sd % kT = nu % kT; sd % freegas_cutoff = ZERO
    sd % rxn => r; sd % law = law
    if (with_adist) then
      r % has_angle_dist = .true.
      r % adist % n_energy = ne_
      sd % adist => r % adist
    else
      sd % adist => null()
    end if
    if (with_edist) then
      sd % edist => r % edist
    else
      sd % edist => null()
    end if
    do k_ = 1, ne_
      if (with_adist) then
        np_ = 1
      else
        np_ = 5 + mod(3 * k_ + law, 5)
      end if
      allocate(sd % distro(k_) % data(M_, np_), sd % Eouts(k_) % data(np_))
      allocate(sd % pdfs(k_) % data(np_), sd % cdfs(k_) % data(np_))
      sd % INTT(k_) = LINEAR_LINEAR
      if (law == 61 .and. mod(k_, 2) == 0) sd % INTT(k_) = HISTOGRAM
      sd % cdfs(k_) % data = ZERO
      emax_ = 0.85_8 * eg_(k_)
      do j_ = 1, np_
        if (with_adist) then
          sd % Eouts(k_) % data(j_) = ZERO
          sd % pdfs(k_) % data(j_) = ONE
          A_ = 0.15_8 * real(k_, 8); R_ = 0.05_8 * real(k_ - 1, 8)
          do i_ = 1, M_
            mu_ = sd % mu(i_)
            sd % distro(k_) % data(i_, j_) = 0.5_8 * (ONE + A_ * mu_ + R_ * (1.5_8 * mu_ * mu_ - 0.5_8))
          end do
        else
          x_ = real(j_ - 1, 8) / real(np_ - 1, 8)
          sd % Eouts(k_) % data(j_) = emax_ * x_ ** 1.5_8
          sd % pdfs(k_) % data(j_) = (x_ + 0.05_8) * exp(-3.0_8 * x_)
          A_ = 0.5_8 + 2.5_8 * x_; R_ = 0.5_8 * (ONE - x_)
          do i_ = 1, M_
            mu_ = sd % mu(i_)
            sd % distro(k_) % data(i_, j_) = 0.5_8 * A_ / sinh(A_) * (cosh(A_ * mu_) + R_ * sinh(A_ * mu_))
          end do
        end if
      end do
      if (.not. with_adist) then          ! normalise the pdf (trapezoid)
        nrm_ = ZERO
        do j_ = 1, np_ - 1
          nrm_ = nrm_ + 0.5_8 * (sd % pdfs(k_) % data(j_) + sd % pdfs(k_) % data(j_ + 1)) * &
                 (sd % Eouts(k_) % data(j_ + 1) - sd % Eouts(k_) % data(j_))
        end do
        sd % pdfs(k_) % data = sd % pdfs(k_) % data / nrm_
      end if
    end do
  end subroutine manual_sd

end program test_dropin
