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
  use scatt,            only: calc_elastic_grid
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
  if (worst < 1.0E-10_8) then
    write(*,*) 'PASS'
  else
    write(*,*) 'FAIL'
    stop 1
  end if
end program test_dropin
