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
  use ace_header,       only: Nuclide, Reaction
  use constants
  use global
  use scattdata_header, only: ScattData
  use search,           only: binary_search
  implicit none
  private
  public :: ndpp_params, calc_elastic_grid_hip, ndpp_hip_error

  ! == struct ndpp_params of include/ndpp_hip.h
  type, bind(C) :: ndpp_params
    integer(c_int) :: order, mu_bins
    real(c_double) :: sab_threshold, brent_mu_thresh, adaptive_mu_tol, adaptive_eout_tol
    integer(c_int) :: adaptive_mu_its, adaptive_eout_its, ne_per_grp
    integer(c_int) :: sab_epts_per_bin, extend_pts, inel_extend_pts
  end type ndpp_params

  interface
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

    function ndpp_last_error() bind(C, name="ndpp_last_error") result(msg)
      import :: c_ptr
      type(c_ptr) :: msg
    end function ndpp_last_error
  end interface

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

end module ndpp_hip_mod
