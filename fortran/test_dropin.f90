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
  use scatt,            only: calc_elastic_grid, calc_inelastic_grid, calc_scattsab, calc_scatt
  use sab,              only: sab_egrid
  use chi,              only: calc_chi
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
    call sab_part(worst)
    call chi_part(worst)
    call convert_part(worst)
    call scatt_part(worst)
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
  if (worst >= 1.0E-10_8) then
    write(*,*) 'FAIL'
    stop 1
  end if

  call sab_part(worst)
  write(*,'(A,ES10.3)') ' drop-in check (S(a,b) thermal tables): worst scale-relative difference = ', worst
  if (worst >= 1.0E-10_8) then
    write(*,*) 'FAIL'
    stop 1
  end if

  call chi_part(worst)
  write(*,'(A,ES10.3)') ' drop-in check (chi): worst absolute difference = ', worst
  if (worst >= 1.0E-10_8) then
    write(*,*) 'FAIL'
    stop 1
  end if

  call convert_part(worst)
  write(*,'(A,ES10.3)') ' drop-in check (init + convert_distro): worst relative difference = ', worst
  if (worst >= 1.0E-10_8) then
    write(*,*) 'FAIL'
    stop 1
  end if

  call scatt_part(worst)
  write(*,'(A,ES10.3)') ' drop-in check (calc_scatt, whole nuclide from raw ACE): worst scale-relative difference = ', worst
  if (worst < 1.0E-10_8) then
    write(*,*) 'PASS'
  else
    write(*,*) 'FAIL'
    stop 1
  end if

contains

  !=============================================================================
  ! Two in-memory thermal tables, the grids the reference builds for them
  ! (sab_egrid, sab.F90:460, + the extra top point of scatt.F90:426-445), then
  ! calc_scattsab (scatt.F90:543) against calc_scattsab_hip.
  !   table 1: skewed discrete E_out x mu (secondary_mode 1) + coherent elastic
  !   table 2: continuous E_out pdf (secondary_mode 2) + incoherent elastic
  !=============================================================================
  subroutine sab_part(worst_out)
    real(8), intent(out) :: worst_out
    type(SAlphaBeta), pointer :: t
    real(8), allocatable, target :: bins(:)
    real(8), allocatable :: Eg(:), tmpg(:), ref_m(:,:,:), hip_m(:,:,:)
    integer :: which, i, j, k, NEi, NEo, NMU, n, ier, ord, kk
    real(8) :: q, kTt, emax, base, sc, er, nrm

    worst_out = ZERO
    ord = 5
    kTt = 2.53E-8_8
    allocate(bins(3)); bins = (/ 0.0_8, 6.25E-7_8, 20.0_8 /)
    do which = 1, 2
      allocate(t)
      t % name = 'lwtr.10t'; t % awr = 0.999167_8; t % kT = kTt
      NEi = 14; NEo = 8; NMU = 4
      t % n_inelastic_e_in = NEi; t % n_inelastic_e_out = NEo; t % n_inelastic_mu = NMU
      allocate(t % inelastic_e_in(NEi), t % inelastic_sigma(NEi))
      do i = 1, NEi
        t % inelastic_e_in(i) = 1.0E-11_8 * (4.0E-6_8 / 1.0E-11_8) ** (real(i - 1, 8) / real(NEi - 1, 8))
        t % inelastic_sigma(i) = 20.0_8 + 60.0_8 / (ONE + t % inelastic_e_in(i) / 1.0E-8_8)
      end do
      t % threshold_inelastic = t % inelastic_e_in(NEi)
      if (which == 1) then
        t % secondary_mode = SAB_SECONDARY_SKEWED
        allocate(t % inelastic_e_out(NEo, NEi), t % inelastic_mu(NMU, NEo, NEi))
        do i = 1, NEi
          base = ONE - TWO * kTt / (t % inelastic_e_in(i) + kTt)
          do j = 1, NEo
            q = (real(j, 8) - 0.5_8) / real(NEo, 8)
            t % inelastic_e_out(j, i) = t % inelastic_e_in(i) * (0.2_8 + 1.6_8 * q) - 1.5_8 * kTt * log(ONE - q)
            do k = 1, NMU
              t % inelastic_mu(k, j, i) = max(-ONE, min(ONE, 0.3_8 * base + &
                  (-0.9_8 + 1.8_8 * (real(k, 8) - 0.5_8) / real(NMU, 8)) * (0.6_8 + 0.3_8 * sin(real(i + 2 * j, 8)))))
            end do
          end do
        end do
        ! coherent elastic: Bragg edges (exact mode, no cosines)
        t % elastic_mode = SAB_ELASTIC_EXACT
        t % n_elastic_e_in = 6; t % n_elastic_mu = 0
        allocate(t % elastic_e_in(6), t % elastic_P(6))
        t % elastic_e_in = (/ 1.8E-9_8, 4.5E-9_8, 1.1E-8_8, 5.0E-8_8, 1.2E-7_8, 4.0E-6_8 /)
        t % elastic_P = (/ 1.0E-9_8, 2.7E-9_8, 3.9E-9_8, 5.5E-9_8, 6.1E-9_8, 7.4E-9_8 /)
        t % threshold_elastic = t % elastic_e_in(6)
      else
        t % secondary_mode = SAB_SECONDARY_CONT
        allocate(t % inelastic_data(NEi))
        do i = 1, NEi
          n = 10 + mod(3 * i, 7)
          t % inelastic_data(i) % n_e_out = n
          allocate(t % inelastic_data(i) % e_out(n), t % inelastic_data(i) % e_out_pdf(n), &
                   t % inelastic_data(i) % e_out_cdf(n), t % inelastic_data(i) % mu(NMU, n))
          emax = 3.0_8 * t % inelastic_e_in(i) + 12.0_8 * kTt
          do j = 1, n
            q = real(j - 1, 8) / real(n - 1, 8)
            t % inelastic_data(i) % e_out(j) = emax * q * (0.35_8 + 0.65_8 * q)
            t % inelastic_data(i) % e_out_pdf(j) = (t % inelastic_data(i) % e_out(j) + 0.02_8 * emax) * &
                exp(-t % inelastic_data(i) % e_out(j) / (t % inelastic_e_in(i) + TWO * kTt))
            do k = 1, NMU
              t % inelastic_data(i) % mu(k, j) = -0.95_8 + 1.9_8 * (real(k, 8) - 0.5_8) / real(NMU, 8) &
                  + 0.04_8 * cos(real(i + j, 8))
            end do
          end do
          nrm = ZERO
          do j = 1, n - 1
            nrm = nrm + 0.5_8 * (t % inelastic_data(i) % e_out_pdf(j) + t % inelastic_data(i) % e_out_pdf(j + 1)) * &
                  (t % inelastic_data(i) % e_out(j + 1) - t % inelastic_data(i) % e_out(j))
          end do
          t % inelastic_data(i) % e_out_pdf = t % inelastic_data(i) % e_out_pdf / nrm
          t % inelastic_data(i) % e_out_cdf = ZERO
        end do
        ! incoherent elastic: discrete cosines
        t % elastic_mode = SAB_ELASTIC_DISCRETE
        t % n_elastic_e_in = 9; t % n_elastic_mu = 5
        allocate(t % elastic_e_in(9), t % elastic_P(9), t % elastic_mu(5, 9))
        do i = 1, 9
          t % elastic_e_in(i) = 1.0E-11_8 * (4.0E-6_8 / 1.0E-11_8) ** (real(i - 1, 8) / 8.0_8)
          t % elastic_P(i) = 5.0_8 / (ONE + t % elastic_e_in(i) / 1.0E-7_8)
          do k = 1, 5
            t % elastic_mu(k, i) = -0.9_8 + 0.4_8 * real(k - 1, 8) + 0.05_8 * sin(real(i, 8))
          end do
        end do
        t % threshold_elastic = t % elastic_e_in(9)
      end if

      ! the incoming grid exactly as the driver builds it (ndpp.F90:768-771)
      if (allocated(Eg)) deallocate(Eg)
      call sab_egrid(t, bins, Eg)
      allocate(tmpg(size(Eg) + 1))
      tmpg(1:size(Eg)) = Eg
      tmpg(size(Eg) + 1) = Eg(size(Eg)) * (ONE + 1.0E-3)
      deallocate(Eg); allocate(Eg(size(tmpg))); Eg = tmpg; deallocate(tmpg)

      if (allocated(ref_m)) deallocate(ref_m)
      allocate(ref_m(ord + 1, size(bins) - 1, size(Eg)))
      call calc_scattsab(t, bins, SCATT_TYPE_LEGENDRE, ord, ref_m, 2001, Eg)
      if (ref_only) then
        write(*,'(A,I2,A,I6,A,3ES13.5)') ' S(a,b) table ', which, ': ', size(Eg), &
              ' points; reference P0..2(g=1) mid-grid: ', ref_m(1:3, 1, size(Eg) / 2)
      else
        call calc_scattsab_hip(t, bins, SCATT_TYPE_LEGENDRE, ord, hip_m, 2001, Eg, ier)
        if (ier /= 0) then
          write(*,*) 'libndpp_hip error ', ier, ': ', trim(ndpp_hip_error())
          stop 3
        end if
        er = ZERO
        do kk = 1, size(Eg)
          sc = maxval(abs(ref_m(:, :, kk)))
          if (sc == ZERO) sc = ONE
          er = max(er, maxval(abs(hip_m(:, :, kk) - ref_m(:, :, kk))) / sc)
        end do
        write(*,'(A,I2,A,I6,A,ES10.3,A,3ES13.5)') ' S(a,b) table ', which, ': ', size(Eg), &
              ' points  err=', er, '  P0..2(g=1) mid-grid: ', hip_m(1:3, 1, size(Eg) / 2)
        worst_out = max(worst_out, er)
      end if
      deallocate(t)
    end do
  end subroutine sab_part

  !=============================================================================
  ! A fissionable nuclide: MT 19 with two nested spectra (law 4 table, then a
  ! Maxwell law 7), MT 20 (Watt, law 11), MT 21 (evaporation, law 9), two delayed
  ! groups (law 4 table, Maxwell).  calc_chi (chi.F90:21) against calc_chi_hip,
  ! both building the union grid themselves.
  !=============================================================================
  subroutine chi_part(worst_out)
    real(8), intent(out) :: worst_out
    type(Nuclide), pointer :: fn
    type(DistEnergy), pointer :: e1, e2, e3, e4
    real(8), allocatable :: bins(:), Eg_r(:), Eg_h(:), ct_r(:,:), cp_r(:,:), cd_r(:,:,:)
    real(8), allocatable :: ct_h(:,:), cp_h(:,:), cd_h(:,:,:)
    integer :: n_grid, ii, ier, thr(3)

    worst_out = ZERO
    allocate(bins(6)); bins = (/ 0.0_8, 1.0E-3_8, 0.1_8, 1.0_8, 5.0_8, 20.0_8 /)
    allocate(fn)
    fn % name = '94240.71c'; fn % zaid = 94240; fn % awr = 237.992_8; fn % kT = 2.53E-8_8
    n_grid = 50; fn % n_grid = n_grid
    allocate(fn % energy(n_grid), fn % fission(n_grid))
    do ii = 1, n_grid
      fn % energy(ii) = 1.0E-11_8 * (20.0_8 / 1.0E-11_8) ** (real(ii - 1, 8) / real(n_grid - 1, 8))
    end do
    fn % fissionable = .true.; fn % n_fission = 3; fn % n_reaction = 3
    allocate(fn % reactions(3), fn % index_fission(3))
    thr = (/ 1, 36, 42 /)
    fn % fission = ZERO
    do ii = 1, 3
      fn % index_fission(ii) = ii
      associate (r => fn % reactions(ii))
        r % MT = 18 + ii; r % threshold = thr(ii); r % has_energy_dist = .true.
        r % Q_value = 180.0_8; r % multiplicity = 1
        allocate(r % sigma(n_grid - thr(ii) + 1))
        r % sigma = (/ (real(4 - ii, 8) * 0.3_8 * real(k, 8) / real(size(r % sigma), 8) + &
                        merge_real(ii), k = 1, size(r % sigma)) /)
        fn % fission(thr(ii):) = fn % fission(thr(ii):) + r % sigma
      end associate
    end do
    fn % nu_t_type = NU_POLYNOMIAL; fn % nu_p_type = NU_NONE
    allocate(fn % nu_t_data(3)); fn % nu_t_data = (/ 2.0_8, 2.4_8, 0.12_8 /)
    fn % nu_d_type = NU_TABULAR
    allocate(fn % nu_d_data(8))
    fn % nu_d_data = (/ 0.0_8, 3.0_8, 1.0E-11_8, 4.0_8, 20.0_8, 0.016_8, 0.016_8, 0.009_8 /)
    fn % n_precursor = 2
    allocate(fn % nu_d_precursor_data(14))
    fn % nu_d_precursor_data = (/ 0.0125_8, 0.0_8, 2.0_8, 1.0E-11_8, 20.0_8, 0.35_8, 0.40_8, &
                                  0.0318_8, 0.0_8, 2.0_8, 1.0E-11_8, 20.0_8, 0.65_8, 0.60_8 /)

    ! MT 19: law 4 table -> next: Maxwell
    allocate(e1); call table_law4(e1, 3, (/ 1.0E-11_8, 1.0_8, 20.0_8 /), 9, 15.0_8)
    e1 % p_valid % n_regions = 0; e1 % p_valid % n_pairs = 2
    allocate(e1 % p_valid % x(2), e1 % p_valid % y(2))
    e1 % p_valid % x = (/ 1.0E-11_8, 20.0_8 /); e1 % p_valid % y = (/ 0.7_8, 0.4_8 /)
    allocate(e2); e2 % law = 7
    allocate(e2 % data(7)); e2 % data = (/ 0.0_8, 2.0_8, 1.0E-11_8, 20.0_8, 1.30_8, 1.45_8, -20.0_8 /)
    e2 % p_valid % n_regions = 0; e2 % p_valid % n_pairs = 0
    e1 % next => e2
    fn % reactions(1) % edist => e1
    ! MT 20: Watt
    allocate(e3); e3 % law = 11
    allocate(e3 % data(13))
    e3 % data = (/ 0.0_8, 2.0_8, 1.0E-11_8, 20.0_8, 0.95_8, 1.05_8, &
                   0.0_8, 2.0_8, 1.0E-11_8, 20.0_8, 2.2_8, 2.6_8, 3.0_8 /)
    e3 % p_valid % n_regions = 0; e3 % p_valid % n_pairs = 0
    fn % reactions(2) % edist => e3
    ! MT 21: evaporation
    allocate(e4); e4 % law = 9
    allocate(e4 % data(7)); e4 % data = (/ 0.0_8, 2.0_8, 5.0_8, 20.0_8, 0.5_8, 0.9_8, 5.5_8 /)
    e4 % p_valid % n_regions = 0; e4 % p_valid % n_pairs = 0
    fn % reactions(3) % edist => e4
    ! delayed groups
    allocate(fn % nu_d_edist(2))
    call table_law4(fn % nu_d_edist(1), 2, (/ 1.0E-11_8, 20.0_8 /), 7, 3.0_8)
    fn % nu_d_edist(2) % law = 7
    allocate(fn % nu_d_edist(2) % data(7))
    fn % nu_d_edist(2) % data = (/ 0.0_8, 2.0_8, 1.0E-11_8, 20.0_8, 0.4_8, 0.45_8, -20.0_8 /)
    ! Tab1 % n_pairs has no default initialisation (endf_header.F90:13): set every p_valid
    fn % nu_d_edist(1) % p_valid % n_regions = 0; fn % nu_d_edist(1) % p_valid % n_pairs = 0
    fn % nu_d_edist(2) % p_valid % n_regions = 0; fn % nu_d_edist(2) % p_valid % n_pairs = 0

    call calc_chi(fn, bins, Eg_r, ct_r, cp_r, cd_r)
    if (ref_only) then
      write(*,'(A,I4,A,5ES12.4)') ' chi: ', size(Eg_r), ' grid points; reference chi_total(:,2) = ', ct_r(:, 2)
      return
    end if
    call calc_chi_hip(fn, bins, Eg_h, ct_h, cp_h, cd_h, ier)
    if (ier /= 0) then
      write(*,*) 'libndpp_hip error ', ier, ': ', trim(ndpp_hip_error())
      stop 3
    end if
    if (size(Eg_h) /= size(Eg_r)) then
      write(*,*) 'chi grid size differs: ', size(Eg_h), size(Eg_r)
      stop 1
    end if
    if (any(Eg_h /= Eg_r)) stop 1
    worst_out = max(maxval(abs(ct_h - ct_r)), maxval(abs(cp_h - cp_r)), maxval(abs(cd_h - cd_r)))
    write(*,'(A,I4,A,ES10.3,A,5ES12.4)') ' chi: ', size(Eg_h), ' grid points  err=', worst_out, &
          '  chi_total(:,2) = ', ct_h(:, 2)
  end subroutine chi_part

  !=============================================================================
  ! ScattData%init (reference, both sides) followed by the reference's
  ! convert_distro on one object and convert_distro_hip on a second one, for four
  ! reactions given as raw ACE blocks: elastic with isotropic / 32-equiprobable /
  ! tabular histogram / tabular lin-lin angular tables; a law-44 (Kalbach-Mann)
  ! continuum; a law-61 table with histogram, lin-lin and isotropic columns; a
  ! law-4 table with an angular distribution.
  !=============================================================================
  subroutine convert_part(worst_out)
    real(8), intent(out) :: worst_out
    type(Nuclide), pointer :: cn
    type(DistEnergy), pointer :: ed, none_ed
    type(ScattData) :: sref, ship
    real(8), allocatable, target :: bins(:)
    integer :: which, ii, kk, M, ier, lc, npts, iE
    real(8) :: er, x

    worst_out = ZERO
    M = 201
    none_ed => null()
    allocate(bins(3)); bins = (/ 0.0_8, 6.25E-7_8, 20.0_8 /)
    allocate(cn)
    cn % name = '92238.71c'; cn % zaid = 92238; cn % awr = 236.0058_8; cn % kT = 2.53E-8_8
    cn % n_grid = 2
    allocate(cn % energy(2)); cn % energy = (/ 1.0E-5_8, 20.0_8 /)
    cn % freegas_cutoff = ZERO
    cn % n_reaction = 4
    allocate(cn % reactions(4))
    call set_rxn(cn % reactions(1), 2, ZERO, 1, 1, .true.)
    call set_rxn(cn % reactions(2), 91, -1.0_8, 1, 1, .true.)
    call set_rxn(cn % reactions(3), 22, -2.0_8, 1, 1, .false.)
    call set_rxn(cn % reactions(4), 28, -3.0_8, 1, 1, .false.)

    ! elastic: four angular tables
    associate (a => cn % reactions(1) % adist)
      cn % reactions(1) % has_angle_dist = .true.
      a % n_energy = 4
      allocate(a % energy(4), a % type(4), a % location(4))
      a % energy = (/ 1.0E-5_8, 0.1_8, 2.0_8, 20.0_8 /)
      a % type = (/ ANGLE_ISOTROPIC, ANGLE_32_EQUI, ANGLE_TABULAR, ANGLE_TABULAR /)
      allocate(a % data(1 + 33 + 2 * (2 + 3 * 9)))
      a % data(1) = ZERO
      a % location(1) = 0
      a % location(2) = 1
      do ii = 1, 33                       ! edges of a forward-peaked pdf
        x = real(ii - 1, 8) / 32.0_8
        a % data(1 + ii) = -ONE + TWO * sqrt(x)
      end do
      do kk = 0, 1
        lc = 34 + kk * (2 + 3 * 9)
        a % location(3 + kk) = lc
        a % data(lc + 1) = real(1 + kk, 8)       ! histogram, then lin-lin
        a % data(lc + 2) = 9.0_8
        do ii = 1, 9
          x = -ONE + 0.25_8 * real(ii - 1, 8)
          a % data(lc + 2 + ii) = x
          a % data(lc + 2 + 9 + ii) = 0.5_8 * (ONE + 0.6_8 * x + 0.2_8 * x * x) / (ONE + 0.2_8 / 3.0_8)
          a % data(lc + 2 + 18 + ii) = ZERO
        end do
      end do
    end associate

    do which = 1, 4
      ed => none_ed
      if (which == 2) then
        allocate(ed); call ace_block(ed, 44)
        cn % reactions(2) % edist => ed; cn % reactions(2) % has_energy_dist = .true.
      else if (which == 3) then
        allocate(ed); call ace_block(ed, 61)
        cn % reactions(3) % edist => ed; cn % reactions(3) % has_energy_dist = .true.
      else if (which == 4) then
        allocate(ed); call ace_block(ed, 4)
        cn % reactions(4) % edist => ed; cn % reactions(4) % has_energy_dist = .true.
        ! law 4 takes its angles from the reaction's adist: reuse the elastic tables
        cn % reactions(4) % has_angle_dist = .true.
        cn % reactions(4) % adist = cn % reactions(1) % adist
      end if
      call sref % init(cn, cn % reactions(which), ed, bins, SCATT_TYPE_LEGENDRE, 5, M)
      call ship % init(cn, cn % reactions(which), ed, bins, SCATT_TYPE_LEGENDRE, 5, M)
      if (.not. (sref % is_init .and. ship % is_init)) stop 2
      call sref % convert_distro()
      if (ref_only) then
        write(*,'(A,I2,A,I3,A,3ES13.5)') ' convert reaction ', which, ': NE=', sref % NE, &
              '  reference f(mu=0) of the last row: ', sref % distro(sref % NE) % data((M + 1) / 2, 1)
      else
        call convert_distro_hip(ship, ier)
        if (ier /= 0) then
          write(*,*) 'libndpp_hip error ', ier, ': ', trim(ndpp_hip_error())
          stop 3
        end if
        er = ZERO
        do iE = 1, sref % NE
          if (any(shape(ship % distro(iE) % data) /= shape(sref % distro(iE) % data))) stop 1
          er = max(er, maxval(abs(ship % distro(iE) % data - sref % distro(iE) % data) / &
                              max(abs(sref % distro(iE) % data), 1.0E-300_8)))
          if (ship % INTT(iE) /= sref % INTT(iE)) stop 1
          if (any(ship % Eouts(iE) % data /= sref % Eouts(iE) % data)) stop 1
          if (allocated(sref % pdfs(iE) % data)) then
            if (any(ship % pdfs(iE) % data /= sref % pdfs(iE) % data)) stop 1
            if (any(ship % cdfs(iE) % data /= sref % cdfs(iE) % data)) stop 1
          end if
        end do
        write(*,'(A,I2,A,I3,A,ES10.3)') ' convert reaction ', which, ': NE=', sref % NE, '  err=', er
        worst_out = max(worst_out, er)
      end if
      call sref % clear()
      call ship % clear()
    end do
  end subroutine convert_part

  !=============================================================================
  ! calc_scatt (scatt.F90:33) against calc_scatt_hip on an O-16-like nuclide given
  ! as raw ACE blocks: elastic (isotropic / tabular / 32-equiprobable angular tables,
  ! free-gas below 4 kT), an inelastic level MT 51 (law 3 + tabular angles), the
  ! MT 91 continuum (law 44, CM) and MT 22 (law 61, lab); three groups, P1, nu-scatter.
  ! Both sides build their own incoming grids (create_Ein_grid).
  !=============================================================================
  subroutine scatt_part(worst_out)
    real(8), intent(out) :: worst_out
    type(Nuclide), pointer :: on
    type(DistEnergy), pointer :: e51, e91, e22
    real(8), allocatable :: bins(:), Eel_r(:), Ein_r(:), Eel_h(:), Ein_h(:)
    real(8), allocatable :: el_r(:,:,:), in_r(:,:,:), nu_r(:,:,:), el_h(:,:,:), in_h(:,:,:), nu_h(:,:,:)
    integer :: n_grid, ii, kk, ord, ier, lc, save_ext, save_iext, thr(4)
    real(8) :: x, sc, er

    worst_out = ZERO
    save_ext = EXTEND_PTS; save_iext = INEL_EXTEND_PTS
    EXTEND_PTS = 3; INEL_EXTEND_PTS = 4        ! keep the CPU side of this check short
    allocate(bins(4)); bins = (/ 0.0_8, 6.25E-7_8, 0.1_8, 20.0_8 /)
    allocate(on)
    on % name = '8016.71c'; on % zaid = 8016; on % awr = 15.8575_8; on % kT = 2.5301E-8_8
    n_grid = 30; on % n_grid = n_grid
    allocate(on % energy(n_grid), on % elastic(n_grid))
    do ii = 1, n_grid
      on % energy(ii) = 1.0E-11_8 * (20.0_8 / 1.0E-11_8) ** (real(ii - 1, 8) / real(n_grid - 1, 8))
      on % elastic(ii) = 3.8_8 + 0.2_8 / (ONE + on % energy(ii))
    end do
    on % freegas_cutoff = 4.0_8 * on % kT
    on % n_reaction = 4
    allocate(on % reactions(4))
    thr = (/ 1, 29, 29, 28 /)           ! 7.7 MeV and 2.9 MeV on this grid
    call set_rxn(on % reactions(1), 2, ZERO, 1, thr(1), .true.)
    call set_rxn(on % reactions(2), 51, -6.05_8, 1, thr(2), .true.)
    call set_rxn(on % reactions(3), 91, -7.2_8, 1, thr(3), .true.)
    call set_rxn(on % reactions(4), 22, -2.5_8, 1, thr(4), .false.)
    do ii = 2, 4
      associate (r => on % reactions(ii))
        allocate(r % sigma(n_grid - r % threshold + 1))
        do kk = 1, size(r % sigma)
          r % sigma(kk) = 0.1_8 * real(ii, 8) * real(kk - 1, 8) / real(size(r % sigma) - 1, 8) &
                          + 0.01_8 * real(kk - 1, 8)
        end do
      end associate
    end do
    ! elastic angular tables
    associate (a => on % reactions(1) % adist)
      on % reactions(1) % has_angle_dist = .true.
      a % n_energy = 3
      allocate(a % energy(3), a % type(3), a % location(3), a % data(1 + (2 + 3 * 7) + 33))
      a % energy = (/ 1.0E-11_8, 1.0E-3_8, 20.0_8 /)
      a % type = (/ ANGLE_ISOTROPIC, ANGLE_TABULAR, ANGLE_32_EQUI /)
      a % data = ZERO
      a % location(1) = 0
      lc = 1; a % location(2) = lc
      a % data(lc + 1) = TWO; a % data(lc + 2) = 7.0_8
      do ii = 1, 7
        x = -ONE + real(ii - 1, 8) / 3.0_8
        if (ii == 7) x = ONE
        a % data(lc + 2 + ii) = x
        a % data(lc + 2 + 7 + ii) = 0.5_8 * (ONE + 0.3_8 * x)
      end do
      lc = 1 + 2 + 3 * 7; a % location(3) = lc - 1     ! data(lc) .. are the 33 edges
      do ii = 1, 33
        a % data(lc - 1 + ii) = -ONE + TWO * (real(ii - 1, 8) / 32.0_8) ** 0.8_8
      end do
    end associate
    ! MT 51: law 3 (level) + a two-energy tabular angular distribution
    allocate(e51); e51 % law = 3
    allocate(e51 % data(2)); e51 % data = (/ 6.43_8, 0.885_8 /)
    e51 % p_valid % n_regions = 0; e51 % p_valid % n_pairs = 0
    on % reactions(2) % edist => e51; on % reactions(2) % has_energy_dist = .true.
    associate (a => on % reactions(2) % adist)
      on % reactions(2) % has_angle_dist = .true.
      a % n_energy = 2
      allocate(a % energy(2), a % type(2), a % location(2), a % data(1 + 2 * (2 + 3 * 3)))
      a % energy = (/ on % energy(thr(2)), 20.0_8 /)
      a % type = ANGLE_TABULAR
      a % data = ZERO
      do kk = 0, 1
        lc = 1 + kk * 11; a % location(1 + kk) = lc
        a % data(lc + 1) = TWO; a % data(lc + 2) = 3.0_8
        a % data(lc + 3 : lc + 5) = (/ -ONE, ZERO, ONE /)
        a % data(lc + 6 : lc + 8) = (/ 0.5_8 - 0.1_8 * kk, 0.5_8, 0.5_8 + 0.1_8 * kk /)
      end do
    end associate
    ! MT 91 (law 44, CM) and MT 22 (law 61, lab) from the raw blocks of convert_part
    allocate(e91); call ace_block(e91, 44)
    e91 % p_valid % n_pairs = 2
    allocate(e91 % p_valid % x(2), e91 % p_valid % y(2))
    e91 % p_valid % x = (/ 1.0E-5_8, 20.0_8 /); e91 % p_valid % y = (/ ONE, 0.8_8 /)
    on % reactions(3) % edist => e91; on % reactions(3) % has_energy_dist = .true.
    allocate(e22); call ace_block(e22, 61)
    e22 % p_valid % n_pairs = 2
    allocate(e22 % p_valid % x(2), e22 % p_valid % y(2))
    e22 % p_valid % x = (/ 1.0E-5_8, 20.0_8 /); e22 % p_valid % y = (/ 0.9_8, ONE /)
    on % reactions(4) % edist => e22; on % reactions(4) % has_energy_dist = .true.

    ord = 1
    call calc_scatt(on, bins, SCATT_TYPE_LEGENDRE, ord, 129, .true., Eel_r, Ein_r, el_r, in_r, nu_r)
    if (ref_only) then
      write(*,'(A,I4,A,I4,A,2ES13.5)') ' calc_scatt: ', size(Eel_r), ' elastic and ', size(Ein_r), &
            ' inelastic incoming energies; reference el P0,P1 (g=1, point 3): ', el_r(1:2, 1, 3)
    else
      ord = 1
      call calc_scatt_hip(on, bins, SCATT_TYPE_LEGENDRE, ord, 129, .true., Eel_h, Ein_h, el_h, in_h, nu_h, ier)
      if (ier /= 0) then
        write(*,*) 'libndpp_hip error ', ier, ': ', trim(ndpp_hip_error())
        stop 3
      end if
      if (size(Eel_h) /= size(Eel_r) .or. size(Ein_h) /= size(Ein_r)) stop 1
      if (any(Eel_h /= Eel_r) .or. any(Ein_h /= Ein_r)) stop 1
      do kk = 1, size(Eel_r)
        sc = maxval(abs(el_r(:, :, kk))); if (sc == ZERO) sc = ONE
        er = maxval(abs(el_h(:, :, kk) - el_r(:, :, kk))) / sc
        worst_out = max(worst_out, er)
      end do
      write(*,'(A,I4,A,ES10.3)') ' calc_scatt elastic:   ', size(Eel_r), ' points  err=', worst_out
      er = ZERO
      do kk = 1, size(Ein_r)
        sc = max(maxval(abs(in_r(:, :, kk))), maxval(abs(nu_r(:, :, kk)))); if (sc == ZERO) sc = ONE
        er = max(er, max(maxval(abs(in_h(:, :, kk) - in_r(:, :, kk))), &
                         maxval(abs(nu_h(:, :, kk) - nu_r(:, :, kk)))) / sc)
      end do
      write(*,'(A,I4,A,ES10.3)') ' calc_scatt inelastic: ', size(Ein_r), ' points  err=', er
      worst_out = max(worst_out, er)
    end if
    EXTEND_PTS = save_ext; INEL_EXTEND_PTS = save_iext
  end subroutine scatt_part

  ! raw ACE LDAT block of law 4 / 44 / 61 with 3 incoming energies
  subroutine ace_block(ed, law)
    type(DistEnergy), intent(inout) :: ed
    integer, intent(in) :: law
    integer, parameter :: ne_ = 3
    integer :: np_(ne_), kE, j, pos, n_words, per, k, lca
    real(8) :: ein_(ne_), eo, emax, w(4096)
    ein_ = (/ 1.0_8, 5.0_8, 20.0_8 /)
    np_ = (/ 5, 8, 6 /)
    ed % law = law
    ed % p_valid % n_regions = 0; ed % p_valid % n_pairs = 0
    w = ZERO
    w(1) = ZERO; w(2) = real(ne_, 8)
    w(3 : 2 + ne_) = ein_
    pos = 2 + 2 * ne_
    do kE = 1, ne_
      w(2 + ne_ + kE) = real(pos, 8)
      emax = 0.9_8 * ein_(kE)
      w(pos + 1) = TWO; if (law == 44 .and. kE == 2) w(pos + 1) = 22.0_8     ! INTT' > 10
      w(pos + 2) = real(np_(kE), 8)
      do j = 1, np_(kE)
        eo = emax * (real(j - 1, 8) / real(np_(kE) - 1, 8)) ** 1.5_8
        w(pos + 2 + j) = eo
        w(pos + 2 + np_(kE) + j) = exp(-eo / (0.3_8 * emax)) / (0.3_8 * emax)
        w(pos + 2 + 2 * np_(kE) + j) = ONE - exp(-eo / (0.3_8 * emax))
      end do
      per = 2 + 3 * np_(kE)
      if (law == 44) then
        do j = 1, np_(kE)
          w(pos + per + j) = 0.05_8 * real(j, 8)                    ! R
          w(pos + per + np_(kE) + j) = 0.5_8 + 0.3_8 * real(j, 8)   ! A
        end do
        per = per + 2 * np_(kE)
      else if (law == 61) then
        lca = pos + per + np_(kE)           ! angular tables follow the locator list
        do j = 1, np_(kE)
          if (mod(j, 3) == 0) then
            w(pos + per + j) = ZERO        ! isotropic
          else
            w(pos + per + j) = real(lca, 8)
            w(lca + 1) = real(1 + mod(j, 2), 8)     ! histogram / lin-lin
            w(lca + 2) = 5.0_8
            do k = 1, 5
              w(lca + 2 + k) = -ONE + 0.5_8 * real(k - 1, 8)
              w(lca + 7 + k) = 0.5_8 + 0.1_8 * real(j, 8) * (-ONE + 0.5_8 * real(k - 1, 8))
              w(lca + 12 + k) = ZERO
            end do
            lca = lca + 17
          end if
        end do
        per = lca - pos
      end if
      pos = pos + per
    end do
    n_words = pos
    allocate(ed % data(n_words))
    ed % data = w(1:n_words)
  end subroutine ace_block

  pure function merge_real(ii) result(v)
    integer, intent(in) :: ii
    real(8) :: v
    v = 0.1_8 * real(ii, 8)
  end function merge_real

  ! edist%data of an ACE law-4 table: NR=0, NE, E_in(NE), L(NE), then per E_in
  ! INTT, NP, E_out(NP), pdf(NP), cdf(NP)  (layout read at chidata_header.F90:258-350)
  subroutine table_law4(ed, ne_, ein_, np_, emax)
    type(DistEnergy), intent(inout) :: ed
    integer, intent(in) :: ne_, np_
    real(8), intent(in) :: ein_(ne_), emax
    integer :: kE, j, pos, lc
    real(8) :: eo(np_), pd(np_), cd(np_), T
    ed % law = 4
    allocate(ed % data(2 + 2 * ne_ + ne_ * (2 + 3 * np_)))
    ed % data(1) = ZERO; ed % data(2) = real(ne_, 8)
    ed % data(3 : 2 + ne_) = ein_
    pos = 2 + 2 * ne_
    do kE = 1, ne_
      ed % data(2 + ne_ + kE) = real(pos, 8)
      T = 1.2_8 + 0.05_8 * real(kE - 1, 8)
      do j = 1, np_
        eo(j) = emax * (real(j - 1, 8) / real(np_ - 1, 8)) ** 2
        pd(j) = sqrt(eo(j) + 1.0E-3_8) * exp(-eo(j) / T)
      end do
      cd(1) = ZERO
      do j = 2, np_
        cd(j) = cd(j - 1) + 0.5_8 * (pd(j) + pd(j - 1)) * (eo(j) - eo(j - 1))
      end do
      pd = pd / cd(np_); cd = cd / cd(np_)
      lc = pos
      ed % data(lc + 1) = TWO; ed % data(lc + 2) = real(np_, 8)
      ed % data(lc + 3 : lc + 2 + np_) = eo
      ed % data(lc + 3 + np_ : lc + 2 + 2 * np_) = pd
      ed % data(lc + 3 + 2 * np_ : lc + 2 + 3 * np_) = cd
      pos = pos + 2 + 3 * np_
    end do
    ed % p_valid % n_regions = 0; ed % p_valid % n_pairs = 0
  end subroutine table_law4

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
