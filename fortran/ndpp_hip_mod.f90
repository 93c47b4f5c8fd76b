!===============================================================================
! ndpp_hip_mod -- the thin Fortran host layer that hands NDPP's elastic
! scattering-moment integration to libndpp_hip.so (MI355X) through ISO_C_BINDING.
!
! CALC_ELASTIC_GRID_HIP has the argument list of the reference's
! calc_elastic_grid (scatt.F90:603) and replaces its loop body (:633-672):
! it flattens the elastic ScattData into contiguous buffers, does the
! bookkeeping of scatt_interp_distro on the host (threshold / sigma / energy
! bracketing, scattdata_header.F90:423-485,:542) and makes ONE call of
! ndpp_elastic_leg_batch for all incoming energies.
!
! Builds against the reference's own modules (ace_header, scattdata_header,
! global, search, constants); nothing of the reference is modified or copied.
!===============================================================================
module ndpp_hip_mod
  use iso_c_binding
  use ace_header,       only: Nuclide, Reaction, DistEnergy, SAlphaBeta
  use scatt,            only: create_Ein_grid
  use array_merge,      only: merge_grids => merge
  use constants
  use global
  use scattdata_header, only: ScattData
  use search,           only: binary_search
  use interpolation,    only: interpolate_tab1
  implicit none
  private
  public :: ndpp_params, calc_elastic_grid_hip, calc_inelastic_grid_hip, ndpp_hip_error
  public :: calc_scattsab_hip, calc_chi_hip, convert_distro_hip, calc_scatt_hip

  ! == struct ndpp_params of include/ndpp_hip.h
  type, bind(C) :: ndpp_params
    integer(c_int) :: order, mu_bins
    real(c_double) :: sab_threshold, brent_mu_thresh, adaptive_mu_tol, adaptive_eout_tol
    integer(c_int) :: adaptive_mu_its, adaptive_eout_its, ne_per_grp
    integer(c_int) :: sab_epts_per_bin, extend_pts, inel_extend_pts
  end type ndpp_params

  ! == struct ndpp_sab_flat: SAlphaBeta (ace_header.F90:201-235) as pointers to its own
  ! arrays; only the continuous mode's jagged inelastic_data(:) is concatenated
  type, bind(C) :: ndpp_sab_flat
    real(c_double) :: threshold_inelastic, threshold_elastic
    integer(c_int) :: n_inelastic_e_in, n_inelastic_e_out, n_inelastic_mu, secondary_mode
    type(c_ptr)    :: inelastic_e_in, inelastic_sigma, inelastic_e_out, inelastic_mu
    type(c_ptr)    :: cont_ptr, cont_e_out, cont_pdf, cont_mu
    integer(c_int) :: elastic_mode, n_elastic_e_in, n_elastic_mu
    type(c_ptr)    :: elastic_e_in, elastic_P, elastic_mu
  end type ndpp_sab_flat

  ! == struct ndpp_chi_spectrum / ndpp_chi_nuclide
  type, bind(C) :: ndpp_chi_spectrum
    integer(c_int) :: law, n_data
    type(c_ptr)    :: data
    integer(c_int) :: threshold, n_sigma
    type(c_ptr)    :: sigma
    integer(c_int) :: has_next, pv_n_regions, pv_n_pairs
    type(c_ptr)    :: pv_nbt, pv_int, pv_x, pv_y
  end type ndpp_chi_spectrum

  type, bind(C) :: ndpp_chi_nuclide
    integer(c_int) :: n_grid
    type(c_ptr)    :: energy, fission
    integer(c_int) :: nu_t_type, n_nu_t
    type(c_ptr)    :: nu_t_data
    integer(c_int) :: nu_d_type, n_nu_d
    type(c_ptr)    :: nu_d_data
    integer(c_int) :: n_precursor, n_prec_data
    type(c_ptr)    :: nu_d_precursor_data
  end type ndpp_chi_nuclide

  ! == struct ndpp_ace_reaction
  type, bind(C) :: ndpp_ace_reaction
    integer(c_int) :: MT, law, has_angle_dist, n_adist
    type(c_ptr)    :: adist_energy, adist_type, adist_location
    integer(c_int) :: n_adist_data
    type(c_ptr)    :: adist_data
    integer(c_int) :: n_edata
    type(c_ptr)    :: edata
    real(c_double) :: threshold_energy
  end type ndpp_ace_reaction

  interface
    ! int ndpp_convert_distro(int mu_bins, const ndpp_ace_reaction*, int G, const double* e_bins,
    !     int NE, int total_np, double* e_grid, int* row_ptr, double* eout, double* pdf,
    !     double* cdf, int* intt, double* f)
    function ndpp_convert_distro(mu_bins, r, G, e_bins, NE, total_np, e_grid, row_ptr, eout, &
                                 pdf, cdf, intt, f) bind(C, name="ndpp_convert_distro") result(rc)
      import :: c_int, c_double, ndpp_ace_reaction
      integer(c_int), value :: mu_bins, G, NE, total_np
      type(ndpp_ace_reaction), intent(in) :: r
      real(c_double), intent(in)  :: e_bins(*)
      real(c_double), intent(out) :: e_grid(*), eout(*), pdf(*), cdf(*), f(*)
      integer(c_int), intent(out) :: row_ptr(*), intt(*)
      integer(c_int) :: rc
    end function ndpp_convert_distro

    ! int ndpp_sab_batch(const ndpp_params*, const ndpp_sab_flat*, int n_ein,
    !     const double* ein, int G, const double* e_bins, double* el, double* inel,
    !     double* scatt_mat)
    function ndpp_sab_batch(p, t, n_ein, ein, G, e_bins, el, inel, scatt_mat) &
        bind(C, name="ndpp_sab_batch") result(rc)
      import :: c_int, c_double, c_ptr, ndpp_params, ndpp_sab_flat
      type(ndpp_params), intent(in) :: p
      type(ndpp_sab_flat), intent(in) :: t
      integer(c_int), value :: n_ein, G
      real(c_double), intent(in) :: ein(*), e_bins(*)
      type(c_ptr), value :: el, inel                  ! c_null_ptr: parts not wanted
      real(c_double), intent(out) :: scatt_mat(*)     ! (order, groups, n_ein)
      integer(c_int) :: rc
    end function ndpp_sab_batch

    ! int ndpp_chi_batch(const ndpp_chi_nuclide*, int n_prompt, const ndpp_chi_spectrum*,
    !     int n_delay, const ndpp_chi_spectrum*, int G, const double* e_bins, int n_ein,
    !     const double* e_grid, double* chi_t, double* chi_p, double* chi_d)
    function ndpp_chi_batch(nuc, n_prompt, prompt, n_delay, delay, G, e_bins, n_ein, &
                            e_grid, chi_t, chi_p, chi_d) bind(C, name="ndpp_chi_batch") result(rc)
      import :: c_int, c_double, ndpp_chi_nuclide, ndpp_chi_spectrum
      type(ndpp_chi_nuclide), intent(in) :: nuc
      integer(c_int), value :: n_prompt, n_delay, G, n_ein
      type(ndpp_chi_spectrum), intent(in) :: prompt(*), delay(*)
      real(c_double), intent(in) :: e_bins(*), e_grid(*)
      real(c_double), intent(out) :: chi_t(*), chi_p(*), chi_d(*)
      integer(c_int) :: rc
    end function ndpp_chi_batch

    ! int ndpp_elastic_leg_batch(const ndpp_params*, double A, double kT,
    !     double freegas_cutoff, double Q, int n_ein, const double* ein,
    !     const int* row_lo, const double* w_hi, int n_rows, const double* f_tab,
    !     int G, const double* e_bins, double* out, int* status, ndpp_stats*)
    function ndpp_elastic_leg_batch(p, A, kT, freegas_cutoff, Q, n_ein, ein, row_lo, &
                                    w_hi, n_rows, f_tab, G, e_bins, out, status, stats) &
        bind(C, name="ndpp_elastic_leg_batch") result(rc)
      import :: c_int, c_double, c_ptr, ndpp_params
      type(ndpp_params), intent(in) :: p
      real(c_double), value :: A, kT, freegas_cutoff, Q
      integer(c_int), value :: n_ein, n_rows, G
      real(c_double), intent(in) :: ein(*), w_hi(*), f_tab(*), e_bins(*)
      integer(c_int), intent(in) :: row_lo(*)
      real(c_double), intent(out) :: out(*)
      integer(c_int), intent(out) :: status(*)
      type(c_ptr), value :: stats
      integer(c_int) :: rc
    end function ndpp_elastic_leg_batch

    ! int ndpp_file6_leg_batch(const ndpp_params*, double awr, int frame_cm, int n_ein,
    !     const double* ein, const int* row_lo, int n_rows, const double* e_grid,
    !     const int* row_ptr, const double* eout, const double* pdf, const int* intt,
    !     const double* f, int G, const double* e_bins, double* out, int* status)
    function ndpp_file6_leg_batch(p, awr, frame_cm, n_ein, ein, row_lo, n_rows, e_grid, &
                                  row_ptr, eout, pdf, intt, f, G, e_bins, out, status) &
        bind(C, name="ndpp_file6_leg_batch") result(rc)
      import :: c_int, c_double, ndpp_params
      type(ndpp_params), intent(in) :: p
      real(c_double), value :: awr
      integer(c_int), value :: frame_cm, n_ein, n_rows, G
      real(c_double), intent(in) :: ein(*), e_grid(*), eout(*), pdf(*), f(*), e_bins(*)
      integer(c_int), intent(in) :: row_lo(*), row_ptr(*), intt(*)
      real(c_double), intent(out) :: out(*)
      integer(c_int), intent(out) :: status(*)
      integer(c_int) :: rc
    end function ndpp_file6_leg_batch

    ! int ndpp_law9_leg_batch(const ndpp_params*, int n_ein, const double* ein,
    !     const int* row_lo, const double* w_hi, int n_rows, const double* f_tab,
    !     int n_edata, const double* edata, int G, const double* e_bins, double* out, int* status)
    function ndpp_law9_leg_batch(p, n_ein, ein, row_lo, w_hi, n_rows, f_tab, n_edata, edata, &
                                 G, e_bins, out, status) &
        bind(C, name="ndpp_law9_leg_batch") result(rc)
      import :: c_int, c_double, ndpp_params
      type(ndpp_params), intent(in) :: p
      integer(c_int), value :: n_ein, n_rows, n_edata, G
      real(c_double), intent(in) :: ein(*), w_hi(*), f_tab(*), edata(*), e_bins(*)
      integer(c_int), intent(in) :: row_lo(*)
      real(c_double), intent(out) :: out(*)
      integer(c_int), intent(out) :: status(*)
      integer(c_int) :: rc
    end function ndpp_law9_leg_batch

    function ndpp_last_error() bind(C, name="ndpp_last_error") result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function ndpp_last_error
  end interface

  real(8), allocatable, save :: pvalid_buf(:)   ! per-point p_valid of the reaction in hand

contains

  ! module global's tunables (global.F90:32-59) -> the C struct
  function params_from_global(order, mu_bins) result(p)
    integer, intent(in) :: order, mu_bins
    type(ndpp_params) :: p
    p % order = order
    p % mu_bins = mu_bins
    p % sab_threshold = SAB_THRESHOLD
    p % brent_mu_thresh = BRENT_MU_THRESH
    p % adaptive_mu_tol = ADAPTIVE_MU_TOL
    p % adaptive_eout_tol = ADAPTIVE_EOUT_TOL
    p % adaptive_mu_its = ADAPTIVE_MU_ITS
    p % adaptive_eout_its = ADAPTIVE_EOUT_ITS
    p % ne_per_grp = NE_PER_GRP
    p % sab_epts_per_bin = SAB_EPTS_PER_BIN
    p % extend_pts = EXTEND_PTS
    p % inel_extend_pts = INEL_EXTEND_PTS
  end function params_from_global

  function ndpp_hip_error() result(msg)
    character(len=512) :: msg
    character(kind=c_char), pointer :: s(:)
    integer :: k
    msg = ''
    call c_f_pointer(ndpp_last_error(), s, [512])
    do k = 1, 512
      if (s(k) == c_null_char) exit
      msg(k:k) = s(k)
    end do
  end function ndpp_hip_error

  subroutine calc_elastic_grid_hip(nuc, mu_out, rxn_data, Ein, order, E_bins, &
                                   scatt_mat, ierr)
    type(Nuclide), pointer, intent(in)     :: nuc
    real(8), intent(inout)                 :: mu_out(:)   ! unused for Legendre output
    type(ScattData), intent(inout), target :: rxn_data(:)
    real(8), allocatable, intent(in)       :: Ein(:)
    integer, intent(in)                    :: order
    real(8), intent(in)                    :: E_bins(:)
    real(8), allocatable, intent(out)      :: scatt_mat(:,:,:)
    integer, intent(out)                   :: ierr

    type(ScattData), pointer :: sd
    type(Reaction),  pointer :: rxn
    type(ndpp_params) :: p
    integer :: groups, NE, irxn, iE, k, nb, iEg, nuc_iE, M
    real(8) :: f, sigS
    real(c_double), allocatable :: f_tab(:,:), ein_b(:), w_hi(:), out(:,:,:)
    integer(c_int), allocatable :: row_lo(:), status(:), where_(:)

    groups = size(E_bins) - 1
    NE = size(Ein)
    allocate(scatt_mat(order, groups, NE))
    scatt_mat = ZERO
    ierr = 0

    do irxn = 1, size(rxn_data)
      sd => rxn_data(irxn)
      if (.not. sd % is_init) cycle
      if (sd % rxn % MT /= ELASTIC) cycle
      rxn => sd % rxn
      M = size(sd % mu)

      ! ---- flatten: this%distro(iE)%data(:,1) rows -> f_tab(M, NE_sd), contiguous
      allocate(f_tab(M, sd % NE))
      do k = 1, sd % NE
        f_tab(:, k) = sd % distro(k) % data(:, 1)
      end do

      ! ---- scatt_interp_distro's bookkeeping, scattdata_header.F90:423-485
      allocate(ein_b(NE), w_hi(NE), row_lo(NE), where_(NE))
      nb = 0
      do iE = 1, NE
        if (Ein(iE) > E_bins(size(E_bins))) cycle          ! top point, copied below
        if (((Ein(iE) <= nuc % energy(rxn % threshold)) .and. (rxn % threshold > 1)) &
            .or. (Ein(iE) > sd % E_bins(size(sd % E_bins)))) cycle       ! :423-431
        if (Ein(iE) >= nuc % energy(nuc % n_grid)) then
          iEg = sd % NE - 1                                              ! :432-442
        else
          if (Ein(iE) <= nuc % energy(1)) then
            nuc_iE = 1
          else
            nuc_iE = binary_search(nuc % energy, nuc % n_grid, Ein(iE))
          end if
          if (nuc % energy(nuc_iE) == nuc % energy(nuc_iE + 1)) nuc_iE = nuc_iE + 1
          f = (Ein(iE) - nuc % energy(nuc_iE)) / &
              (nuc % energy(nuc_iE + 1) - nuc % energy(nuc_iE))
          nuc_iE = nuc_iE - rxn % threshold + 1
          sigS = (ONE - f) * nuc % elastic(nuc_iE) + f * nuc % elastic(nuc_iE + 1)
          if (sigS <= ZERO) cycle                                        ! :466-468
          if (Ein(iE) < sd % E_grid(1)) then
            iEg = 1
          else
            iEg = binary_search(sd % E_grid, sd % NE, Ein(iE))
          end if
          if (sd % E_grid(iEg) >= sd % E_grid(iEg + 1)) iEg = iEg + 1     ! :480-482
        end if
        nb = nb + 1
        where_(nb) = iE
        ein_b(nb) = Ein(iE)
        row_lo(nb) = iEg - 1                                  ! 0-based lower row
        w_hi(nb) = (Ein(iE) - sd % E_grid(iEg)) / &
                   (sd % E_grid(iEg + 1) - sd % E_grid(iEg))             ! :542
      end do

      ! ---- one GPU call for the whole grid (integrate_distro :533-591)
      if (nb > 0) then
        allocate(out(order, groups, nb), status(nb))
        p = params_from_global(order, M)
        ierr = ndpp_elastic_leg_batch(p, sd % awr, sd % kT, sd % freegas_cutoff, &
                 rxn % Q_value, nb, ein_b, row_lo, w_hi, sd % NE, f_tab, groups, &
                 E_bins, out, status, c_null_ptr)
        if (ierr /= 0) return
        do k = 1, nb
          scatt_mat(:, :, where_(k)) = out(:, :, k)   ! elastic: no sigma scaling (:494-497)
        end do
        deallocate(out, status)
      end if
      deallocate(f_tab, ein_b, w_hi, row_lo, where_)
    end do

    ! the extra point above the top group copies its neighbour, scatt.F90:664-670
    do iE = 2, NE
      if (Ein(iE) > E_bins(size(E_bins))) scatt_mat(:, :, iE) = scatt_mat(:, :, iE - 1)
    end do
  end subroutine calc_elastic_grid_hip


  !=============================================================================
  ! CALC_INELASTIC_GRID_HIP replaces calc_inelastic_grid (scatt.F90:682-778): for
  ! every non-elastic ScattData the bookkeeping of scatt_interp_distro
  ! (scattdata_header.F90:423-497: threshold / top tests, sigma lerp, row search,
  ! p_valid) runs here on the host, the integrals of integrate_distro (:513-662)
  ! run on the GPU -- one batch call per reaction for the whole grid -- and the
  ! sigma * p_valid scaling, the reaction sum and the nu-scatter yield weighting
  ! are applied in the reference's order.
  !=============================================================================
  subroutine calc_inelastic_grid_hip(nuc, mu_out, rxn_data, Ein, order, E_bins, nuscatt, &
                                     scatt_mat, nuscatt_mat, ierr)
    type(Nuclide), pointer, intent(in)     :: nuc
    real(8), intent(inout)                 :: mu_out(:)
    type(ScattData), intent(inout), target :: rxn_data(:)
    real(8), allocatable, intent(in)       :: Ein(:)
    integer, intent(in)                    :: order
    real(8), intent(in)                    :: E_bins(:)
    logical, intent(in)                    :: nuscatt
    real(8), allocatable, intent(out)      :: scatt_mat(:,:,:)
    real(8), allocatable, intent(out)      :: nuscatt_mat(:,:,:)
    integer, intent(out)                   :: ierr

    type(ScattData), pointer :: sd
    type(Reaction),  pointer :: rxn
    type(ndpp_params) :: p
    integer :: groups, NE, irxn, iE, k, nb, iEg, nuc_iE, M, j, ntot, kind
    real(8) :: f, sigS, p_valid, yield
    real(c_double), allocatable :: ein_b(:), w_hi(:), scale(:), out(:,:,:), f_tab(:,:)
    real(c_double), allocatable :: eout(:), pdf(:), fcols(:,:)
    integer(c_int), allocatable :: row_lo(:), status(:), where_(:), row_ptr(:), intt(:)
    real(8), allocatable :: temp(:,:)

    groups = size(E_bins) - 1
    NE = size(Ein)
    allocate(scatt_mat(order, groups, NE))
    scatt_mat = ZERO
    if (nuscatt) then
      allocate(nuscatt_mat(order, groups, NE))
      nuscatt_mat = ZERO
    end if
    allocate(temp(order, groups))
    ierr = 0

    do irxn = 1, size(rxn_data)
      sd => rxn_data(irxn)
      if (.not. sd % is_init) cycle
      if (sd % rxn % MT == ELASTIC) cycle
      rxn => sd % rxn
      M = size(sd % mu)

      ! ---- which integrator does integrate_distro pick? (:533-656)
      if (associated(sd % adist) .and. (.not. associated(sd % edist))) then
        kind = 1                                   ! file4 CM, both rows + blend
      else if (rxn % scatter_in_cm) then
        kind = 2                                   ! unitbase + file6 CM
      else if (associated(sd % adist) .and. sd % law == 9) then
        kind = 3                                   ! law 9, both rows + blend
      else
        kind = 4                                   ! unitbase + file6 lab
      end if

      ! ---- scatt_interp_distro's bookkeeping for every incoming energy
      allocate(ein_b(NE), w_hi(NE), row_lo(NE), where_(NE), scale(NE))
      nb = 0
      do iE = 1, NE
        if (Ein(iE) > E_bins(size(E_bins))) cycle
        if (((Ein(iE) <= nuc % energy(rxn % threshold)) .and. (rxn % threshold > 1)) &
            .or. (Ein(iE) > sd % E_bins(size(sd % E_bins)))) cycle
        if (Ein(iE) >= nuc % energy(nuc % n_grid)) then
          sigS = rxn % sigma(size(rxn % sigma))
          iEg = sd % NE - 1
        else
          if (Ein(iE) <= nuc % energy(1)) then
            nuc_iE = 1
          else
            nuc_iE = binary_search(nuc % energy, nuc % n_grid, Ein(iE))
          end if
          if (nuc % energy(nuc_iE) == nuc % energy(nuc_iE + 1)) nuc_iE = nuc_iE + 1
          f = (Ein(iE) - nuc % energy(nuc_iE)) / &
              (nuc % energy(nuc_iE + 1) - nuc % energy(nuc_iE))
          nuc_iE = nuc_iE - rxn % threshold + 1
          sigS = (ONE - f) * rxn % sigma(nuc_iE) + f * rxn % sigma(nuc_iE + 1)
          if (sigS <= ZERO) cycle
          if (Ein(iE) < sd % E_grid(1)) then
            iEg = 1
          else
            iEg = binary_search(sd % E_grid, sd % NE, Ein(iE))
          end if
          if (sd % E_grid(iEg) >= sd % E_grid(iEg + 1)) iEg = iEg + 1
        end if
        if (associated(sd % edist)) then
          p_valid = interpolate_tab1(sd % edist % p_valid, Ein(iE))
        else
          p_valid = ONE
        end if
        nb = nb + 1
        where_(nb) = iE
        ein_b(nb) = Ein(iE)
        row_lo(nb) = iEg - 1
        w_hi(nb) = (Ein(iE) - sd % E_grid(iEg)) / (sd % E_grid(iEg + 1) - sd % E_grid(iEg))
        scale(nb) = sigS          ! distro * sigS * p_valid (:496) is applied below
        call store_pvalid(nb, p_valid)
      end do

      if (nb > 0) then
        allocate(out(order, groups, nb), status(nb))
        p = params_from_global(order, M)
        select case (kind)
        case (1, 3)
          allocate(f_tab(M, sd % NE))
          do k = 1, sd % NE
            f_tab(:, k) = sd % distro(k) % data(:, 1)
          end do
          if (kind == 1) then
            ierr = ndpp_elastic_leg_batch(p, sd % awr, sd % kT, ZERO, rxn % Q_value, nb, &
                     ein_b, row_lo, w_hi, sd % NE, f_tab, groups, E_bins, out, status, c_null_ptr)
          else
            ierr = ndpp_law9_leg_batch(p, nb, ein_b, row_lo, w_hi, sd % NE, f_tab, &
                     size(sd % edist % data), sd % edist % data, groups, E_bins, out, status)
          end if
          deallocate(f_tab)
        case (2, 4)
          ! CSR flattening of Eouts / pdfs / distro (scattdata_header.F90:36-48)
          allocate(row_ptr(sd % NE + 1), intt(sd % NE))
          row_ptr(1) = 0
          do k = 1, sd % NE
            row_ptr(k + 1) = row_ptr(k) + size(sd % Eouts(k) % data)
            intt(k) = sd % INTT(k)
          end do
          ntot = row_ptr(sd % NE + 1)
          allocate(eout(ntot), pdf(ntot), fcols(M, ntot))
          do k = 1, sd % NE
            j = row_ptr(k)
            eout(j + 1 : row_ptr(k + 1)) = sd % Eouts(k) % data
            pdf(j + 1 : row_ptr(k + 1)) = sd % pdfs(k) % data
            fcols(:, j + 1 : row_ptr(k + 1)) = sd % distro(k) % data
          end do
          ierr = ndpp_file6_leg_batch(p, sd % awr, merge(1, 0, kind == 2), nb, ein_b, row_lo, &
                   sd % NE, sd % E_grid, row_ptr, eout, pdf, intt, fcols, groups, E_bins, &
                   out, status)
          deallocate(row_ptr, intt, eout, pdf, fcols)
        end select
        if (ierr /= 0) return
        do k = 1, nb
          iE = where_(k)
          temp = out(:, :, k) * scale(k) * pvalid_buf(k)          ! :496
          scatt_mat(:, :, iE) = scatt_mat(:, :, iE) + temp        ! scatt.F90:753
          if (nuscatt) then
            if (rxn % multiplicity_with_E) then
              yield = interpolate_tab1(rxn % multiplicity_E, Ein(iE))
            else
              yield = real(rxn % multiplicity, 8)
            end if
            nuscatt_mat(:, :, iE) = nuscatt_mat(:, :, iE) + yield * temp   ! :762
          end if
        end do
        deallocate(out, status)
      end if
      deallocate(ein_b, w_hi, row_lo, where_, scale)
      if (allocated(pvalid_buf)) deallocate(pvalid_buf)
    end do

    do iE = 2, NE                                                 ! scatt.F90:766-774
      if (Ein(iE) > E_bins(size(E_bins))) then
        scatt_mat(:, :, iE) = scatt_mat(:, :, iE - 1)
        if (nuscatt) nuscatt_mat(:, :, iE) = nuscatt_mat(:, :, iE - 1)
      end if
    end do

  contains
    subroutine store_pvalid(n, v)
      integer, intent(in) :: n
      real(8), intent(in) :: v
      if (.not. allocated(pvalid_buf)) allocate(pvalid_buf(NE))
      pvalid_buf(n) = v
    end subroutine store_pvalid
  end subroutine calc_inelastic_grid_hip

  !=============================================================================
  ! CALC_SCATTSAB_HIP: the argument list of calc_scattsab (scatt.F90:543) + ierr;
  ! replaces integrate_sab_el + integrate_sab_inel + combine_sab_grid (:574-590)
  ! by one ndpp_sab_batch call.  E_grid is the grid sab_egrid / add_one_more_point
  ! built (the reference's builders are used unchanged, scatt.F90:... ndpp.F90).
  !=============================================================================
  subroutine calc_scattsab_hip(sab, energy_bins, scatt_type, order, scatt_mat, mu_bins, &
                               E_grid, ierr)
    type(SAlphaBeta), pointer, intent(in) :: sab
    real(8), target, intent(in)           :: energy_bins(:)
    integer, intent(in)                   :: scatt_type
    integer, intent(in)                   :: order       ! scatt_order (L = order + 1)
    real(8), allocatable, intent(inout)   :: scatt_mat(:,:,:)
    integer, intent(in)                   :: mu_bins
    real(8), allocatable, target, intent(in) :: E_grid(:)
    integer, intent(out)                  :: ierr

    type(ndpp_params) :: p
    type(ndpp_sab_flat) :: t
    integer(c_int), allocatable, target :: cptr(:)
    real(c_double), allocatable, target :: ce(:), cp(:), cm(:,:)
    integer :: groups, k, n, tot

    ierr = 0
    if (scatt_type /= SCATT_TYPE_LEGENDRE) then
      ierr = -22                       ! the reference has no tabular S(a,b) path either
      return
    end if
    groups = size(energy_bins) - 1
    p = params_from_global(order + 1, mu_bins)
    t % threshold_inelastic = sab % threshold_inelastic
    t % threshold_elastic = sab % threshold_elastic
    t % n_inelastic_e_in = sab % n_inelastic_e_in
    t % n_inelastic_e_out = sab % n_inelastic_e_out
    t % n_inelastic_mu = sab % n_inelastic_mu
    t % secondary_mode = sab % secondary_mode
    t % inelastic_e_in = c_loc(sab % inelastic_e_in)
    t % inelastic_sigma = c_loc(sab % inelastic_sigma)
    t % inelastic_e_out = c_null_ptr;  t % inelastic_mu = c_null_ptr
    t % cont_ptr = c_null_ptr;  t % cont_e_out = c_null_ptr
    t % cont_pdf = c_null_ptr;  t % cont_mu = c_null_ptr
    if (sab % secondary_mode == SAB_SECONDARY_CONT) then
      n = sab % n_inelastic_e_in
      allocate(cptr(n + 1))
      cptr(1) = 0
      do k = 1, n
        cptr(k + 1) = cptr(k) + sab % inelastic_data(k) % n_e_out
      end do
      tot = cptr(n + 1)
      allocate(ce(tot), cp(tot), cm(sab % n_inelastic_mu, tot))
      do k = 1, n
        ce(cptr(k) + 1 : cptr(k + 1)) = sab % inelastic_data(k) % e_out
        cp(cptr(k) + 1 : cptr(k + 1)) = sab % inelastic_data(k) % e_out_pdf
        cm(:, cptr(k) + 1 : cptr(k + 1)) = sab % inelastic_data(k) % mu
      end do
      t % cont_ptr = c_loc(cptr);  t % cont_e_out = c_loc(ce)
      t % cont_pdf = c_loc(cp);    t % cont_mu = c_loc(cm)
    else
      t % inelastic_e_out = c_loc(sab % inelastic_e_out)
      t % inelastic_mu = c_loc(sab % inelastic_mu)
    end if
    t % elastic_mode = sab % elastic_mode
    t % n_elastic_e_in = 0;  t % n_elastic_mu = 0
    t % elastic_e_in = c_null_ptr;  t % elastic_P = c_null_ptr;  t % elastic_mu = c_null_ptr
    if (sab % threshold_elastic > ZERO) then
      t % n_elastic_e_in = sab % n_elastic_e_in
      t % n_elastic_mu = sab % n_elastic_mu
      t % elastic_e_in = c_loc(sab % elastic_e_in)
      t % elastic_P = c_loc(sab % elastic_P)
      if (sab % n_elastic_mu > 0) t % elastic_mu = c_loc(sab % elastic_mu)
    end if

    if (allocated(scatt_mat)) deallocate(scatt_mat)
    allocate(scatt_mat(order + 1, groups, size(E_grid)))
    ierr = ndpp_sab_batch(p, t, size(E_grid), E_grid, groups, energy_bins, c_null_ptr, &
                          c_null_ptr, scatt_mat)
  end subroutine calc_scattsab_hip

  !=============================================================================
  ! CALC_CHI_HIP: the argument list of calc_chi (chi.F90:21) + ierr.  The union
  ! grid is built as calc_chi builds it (:97-113: the spectra's own E_in grids,
  ! merged prompt first, then delayed); the incoming-energy loop (:124-159) is one
  ! ndpp_chi_batch call.
  !=============================================================================
  subroutine calc_chi_hip(nuc, E_bins, E_grid, chi_total, chi_prompt, chi_delay, ierr)
    type(Nuclide), pointer, intent(in)   :: nuc
    real(8), intent(in)                  :: E_bins(:)
    real(8), allocatable, intent(inout)  :: E_grid(:)
    real(8), allocatable, intent(inout)  :: chi_total(:,:), chi_prompt(:,:), chi_delay(:,:,:)
    integer, intent(out)                 :: ierr

    type(ndpp_chi_nuclide) :: cn
    type(ndpp_chi_spectrum), allocatable :: prompt(:), delay(:)
    type(DistEnergy), pointer :: edist
    type(Reaction), pointer :: rxn
    real(8), allocatable :: tmp(:)
    integer :: i, s, num_fiss, groups, NE

    ierr = 0
    groups = size(E_bins) - 1
    num_fiss = 0
    do i = 1, nuc % n_fission
      edist => nuc % reactions(nuc % index_fission(i)) % edist
      num_fiss = num_fiss + 1
      do while (associated(edist % next))
        num_fiss = num_fiss + 1
        edist => edist % next
      end do
    end do
    allocate(prompt(num_fiss), delay(max(nuc % n_precursor, 1)))
    s = 0
    do i = 1, nuc % n_fission
      rxn => nuc % reactions(nuc % index_fission(i))
      edist => rxn % edist
      do
        s = s + 1
        call fill_spectrum(prompt(s), edist)
        prompt(s) % threshold = rxn % threshold
        if (rxn % MT == N_FISSION) then       ! chi.F90:72-76
          prompt(s) % n_sigma = size(nuc % fission)
          prompt(s) % sigma = c_loc(nuc % fission)
        else
          prompt(s) % n_sigma = size(rxn % sigma)
          prompt(s) % sigma = c_loc(rxn % sigma)
        end if
        call grid_union(edist)
        if (.not. associated(edist % next)) exit
        edist => edist % next
      end do
    end do
    do i = 1, nuc % n_precursor
      edist => nuc % nu_d_edist(i)
      call fill_spectrum(delay(i), edist)
      call grid_union(edist)
    end do

    cn % n_grid = nuc % n_grid
    cn % energy = c_loc(nuc % energy);  cn % fission = c_loc(nuc % fission)
    cn % nu_t_type = nuc % nu_t_type;   cn % n_nu_t = 0;  cn % nu_t_data = c_null_ptr
    if (allocated(nuc % nu_t_data)) then
      cn % n_nu_t = size(nuc % nu_t_data);  cn % nu_t_data = c_loc(nuc % nu_t_data)
    end if
    cn % nu_d_type = nuc % nu_d_type;   cn % n_nu_d = 0;  cn % nu_d_data = c_null_ptr
    if (allocated(nuc % nu_d_data)) then
      cn % n_nu_d = size(nuc % nu_d_data);  cn % nu_d_data = c_loc(nuc % nu_d_data)
    end if
    cn % n_precursor = nuc % n_precursor
    cn % n_prec_data = 0;  cn % nu_d_precursor_data = c_null_ptr
    if (allocated(nuc % nu_d_precursor_data)) then
      cn % n_prec_data = size(nuc % nu_d_precursor_data)
      cn % nu_d_precursor_data = c_loc(nuc % nu_d_precursor_data)
    end if

    NE = size(E_grid)
    if (allocated(chi_total)) deallocate(chi_total)
    if (allocated(chi_prompt)) deallocate(chi_prompt)
    if (allocated(chi_delay)) deallocate(chi_delay)
    allocate(chi_total(groups, NE), chi_prompt(groups, NE))
    allocate(chi_delay(groups, NE, nuc % n_precursor))
    block
      real(8), allocatable :: cd(:)
      allocate(cd(max(1, groups * NE * nuc % n_precursor)))
      ierr = ndpp_chi_batch(cn, num_fiss, prompt, nuc % n_precursor, delay, groups, E_bins, &
                            NE, E_grid, chi_total, chi_prompt, cd)
      if (nuc % n_precursor > 0) chi_delay = reshape(cd, shape(chi_delay))
    end block

  contains
    subroutine fill_spectrum(sp, ed)
      type(ndpp_chi_spectrum), intent(out) :: sp
      type(DistEnergy), pointer, intent(in) :: ed
      sp % law = ed % law
      sp % n_data = size(ed % data)
      sp % data = c_loc(ed % data)
      sp % threshold = 1;  sp % n_sigma = 0;  sp % sigma = c_null_ptr
      sp % has_next = 0
      if (associated(ed % next)) sp % has_next = 1
      sp % pv_n_regions = ed % p_valid % n_regions
      sp % pv_n_pairs = ed % p_valid % n_pairs
      sp % pv_nbt = c_null_ptr;  sp % pv_int = c_null_ptr
      sp % pv_x = c_null_ptr;    sp % pv_y = c_null_ptr
      if (allocated(ed % p_valid % nbt)) sp % pv_nbt = c_loc(ed % p_valid % nbt)
      if (allocated(ed % p_valid % int)) sp % pv_int = c_loc(ed % p_valid % int)
      if (allocated(ed % p_valid % x)) sp % pv_x = c_loc(ed % p_valid % x)
      if (allocated(ed % p_valid % y)) sp % pv_y = c_loc(ed % p_valid % y)
    end subroutine fill_spectrum

    ! E_grid := merge(E_grid, the spectrum's incoming energies) (chi.F90:97-113,
    ! chidata_header.F90:98-104)
    subroutine grid_union(ed)
      type(DistEnergy), pointer, intent(in) :: ed
      integer :: NR, n, lc
      NR = int(ed % data(1))
      n = int(ed % data(2 + 2 * NR))
      lc = 2 + 2 * NR
      if (.not. allocated(E_grid)) then
        allocate(E_grid(n))
        E_grid = ed % data(lc + 1 : lc + n)
      else
        call merge_grids(E_grid, ed % data(lc + 1 : lc + n), tmp)
        deallocate(E_grid)
        allocate(E_grid(size(tmp)))
        E_grid = tmp
      end if
    end subroutine grid_union
  end subroutine calc_chi_hip

  !=============================================================================
  ! CONVERT_DISTRO_HIP replaces `call sd % convert_distro()` (scattdata_header.F90
  ! :325) for a ScattData that `init` has set up: the raw ACE blocks the object
  ! points at go to ndpp_convert_distro and the tables come back into
  ! distro(:)%data, Eouts, pdfs, cdfs and INTT exactly as convert_file4/6 leave them.
  !=============================================================================
  subroutine convert_distro_hip(this, ierr)
    class(ScattData), intent(inout) :: this
    integer, intent(out) :: ierr

    type(ndpp_ace_reaction) :: r
    integer(c_int), allocatable :: row_ptr(:), intt(:)
    real(c_double), allocatable :: e_grid(:), eout(:), pdf(:), cdf(:), f(:,:)
    integer :: iE, np, o, tot, M
    logical :: angle_only

    ierr = 0
    if (.not. this % is_init) return
    M = size(this % mu)
    r % MT = this % rxn % MT
    r % law = this % law
    r % has_angle_dist = 0;  r % n_adist = 0;  r % n_adist_data = 0
    r % adist_energy = c_null_ptr;  r % adist_type = c_null_ptr
    r % adist_location = c_null_ptr;  r % adist_data = c_null_ptr
    if (associated(this % adist)) then
      r % has_angle_dist = 1
      r % n_adist = this % adist % n_energy
      r % adist_energy = c_loc(this % adist % energy)
      r % adist_type = c_loc(this % adist % type)
      r % adist_location = c_loc(this % adist % location)
      r % n_adist_data = size(this % adist % data)
      r % adist_data = c_loc(this % adist % data)
    end if
    r % n_edata = 0;  r % edata = c_null_ptr
    if (associated(this % edist)) then
      r % n_edata = size(this % edist % data)
      r % edata = c_loc(this % edist % data)
    end if
    r % threshold_energy = this % E_grid(1)     ! unused: init already built the isotropic adist

    tot = 0
    do iE = 1, this % NE
      tot = tot + size(this % distro(iE) % data, 2)
    end do
    allocate(row_ptr(this % NE + 1), intt(this % NE), e_grid(this % NE))
    allocate(eout(tot), pdf(tot), cdf(tot), f(M, tot))
    ierr = ndpp_convert_distro(M, r, this % groups, this % E_bins, this % NE, tot, e_grid, row_ptr, &
                               eout, pdf, cdf, intt, f)
    if (ierr /= 0) return

    angle_only = (this % law == 0) .or. (this % law == 3) .or. (this % law == 9)
    do iE = 1, this % NE
      o = row_ptr(iE)
      np = row_ptr(iE + 1) - o
      this % distro(iE) % data = f(:, o + 1 : o + np)
      this % INTT(iE) = intt(iE)
      if (allocated(this % Eouts(iE) % data)) deallocate(this % Eouts(iE) % data)
      if (angle_only) then                     ! convert_file4's placeholder, :753-758
        allocate(this % Eouts(iE) % data(2))
        this % Eouts(iE) % data = (/ ZERO, INFINITY /)
      else
        if (allocated(this % pdfs(iE) % data)) deallocate(this % pdfs(iE) % data)
        if (allocated(this % cdfs(iE) % data)) deallocate(this % cdfs(iE) % data)
        allocate(this % Eouts(iE) % data(np), this % pdfs(iE) % data(np), this % cdfs(iE) % data(np))
        this % Eouts(iE) % data = eout(o + 1 : o + np)
        this % pdfs(iE) % data = pdf(o + 1 : o + np)
        this % cdfs(iE) % data = cdf(o + 1 : o + np)
      end if
    end do
  end subroutine convert_distro_hip

  !=============================================================================
  ! CALC_SCATT_HIP: the argument list of calc_scatt (scatt.F90:33) + ierr; the
  ! whole per-nuclide path with every numerical stage on the GPU:
  !   ScattData%init (reference, bookkeeping only)      scatt.F90:87-100
  !   convert_distro          -> convert_distro_hip      :103-106
  !   create_Ein_grid (reference grid builders, host)    :134-135
  !   calc_elastic_grid       -> calc_elastic_grid_hip   :138-139
  !   calc_inelastic_grid     -> calc_inelastic_grid_hip :141-145
  !=============================================================================
  subroutine calc_scatt_hip(nuc, energy_bins, scatt_type, order, mu_bins, nuscatt, Ein_el, &
                            Ein_inel, el_mat, inel_mat, nuinel_mat, ierr)
    type(Nuclide), pointer, intent(in)  :: nuc
    real(8), intent(in)                 :: energy_bins(:)
    integer, intent(in)                 :: scatt_type
    integer, intent(inout)              :: order
    integer, intent(in)                 :: mu_bins
    logical, intent(in)                 :: nuscatt
    real(8), allocatable, intent(inout) :: Ein_el(:), Ein_inel(:)
    real(8), allocatable, intent(inout) :: el_mat(:,:,:), inel_mat(:,:,:), nuinel_mat(:,:,:)
    integer, intent(out)                :: ierr

    type(DistEnergy), pointer :: edist
    type(Reaction),   pointer :: rxn
    type(ScattData), allocatable, target :: rxn_data(:)
    real(8), allocatable :: mu_out(:)
    integer :: num_tot_rxn, i_rxn, k, sd_order
    real(8) :: inel_thresh, cutoff

    ierr = 0
    if (scatt_type /= SCATT_TYPE_LEGENDRE) then
      ierr = -22
      return
    end if
    num_tot_rxn = 0
    do i_rxn = 1, nuc % n_reaction
      rxn => nuc % reactions(i_rxn)
      num_tot_rxn = num_tot_rxn + 1
      if (rxn % has_energy_dist) then
        edist => rxn % edist
        do while (associated(edist % next))
          edist => edist % next
          num_tot_rxn = num_tot_rxn + 1
        end do
      end if
    end do
    allocate(rxn_data(num_tot_rxn))
    k = 0
    do i_rxn = 1, nuc % n_reaction
      k = k + 1
      rxn => nuc % reactions(i_rxn)
      edist => rxn % edist
      call rxn_data(k) % init(nuc, rxn, edist, energy_bins, scatt_type, order, mu_bins)
      if (associated(rxn % edist)) then
        do while (associated(edist % next))
          edist => edist % next
          k = k + 1
          call rxn_data(k) % init(nuc, rxn, edist, energy_bins, scatt_type, order, mu_bins)
        end do
      end if
    end do

    cutoff = ZERO
    inel_thresh = energy_bins(size(energy_bins))
    sd_order = order + 1
    do k = 1, num_tot_rxn
      call convert_distro_hip(rxn_data(k), ierr)
      if (ierr /= 0) return
      if (rxn_data(k) % is_init) then
        sd_order = rxn_data(k) % order
        rxn => rxn_data(k) % rxn
        if (rxn % MT == ELASTIC) then
          cutoff = rxn_data(k) % freegas_cutoff
        else if (nuc % energy(rxn % threshold) < inel_thresh) then
          inel_thresh = nuc % energy(rxn % threshold)
        end if
      end if
    end do

    call create_Ein_grid(rxn_data, energy_bins, nuc % energy, nuc % awr, nuc % kT, cutoff, &
                         inel_thresh, Ein_el, Ein_inel)

    allocate(mu_out(mu_bins))
    mu_out = ZERO
    call calc_elastic_grid_hip(nuc, mu_out, rxn_data, Ein_el, sd_order, energy_bins, el_mat, ierr)
    if (ierr /= 0) return
    if (allocated(Ein_inel)) then
      call calc_inelastic_grid_hip(nuc, mu_out, rxn_data, Ein_inel, sd_order, energy_bins, nuscatt, &
                                   inel_mat, nuinel_mat, ierr)
      if (ierr /= 0) return
    end if
    do k = 1, num_tot_rxn
      call rxn_data(k) % clear()
    end do
  end subroutine calc_scatt_hip

end module ndpp_hip_mod
