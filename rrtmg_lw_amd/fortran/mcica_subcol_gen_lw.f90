!  Drop-in replacement for the reference's module mcica_subcol_gen_lw (src/mcica_subcol_gen_lw.f90:39-60):
!  same module name, public subroutines get_alpha (:68-70) and mcica_subcol_lw (:183-185), same argument
!  order, kinds, intents and assumed-shape dummies.  The bodies forward to rrtmg_lw_hip_get_alpha and
!  rrtmg_lw_hip_mcica_subcol (include/rrtmg_lw_hip.h).  `iplon` is accepted and ignored exactly as in the
!  reference, whose generator always fills every column.
      module mcica_subcol_gen_lw

      use iso_c_binding
      use parkind, only : im => kind_im, rb => kind_rb
      use rrtmg_lw_init, only : rrtmg_lw_hip_abort, rrtmg_lw_hip_gpoints

      implicit none

      public :: get_alpha, mcica_subcol_lw

      interface
         function rrtmg_lw_hip_get_alpha(ncol, nlay, icld, idcor, decorr_con, dz, lat, juldat, cldfrac, alpha) &
               bind(C, name='rrtmg_lw_hip_get_alpha') result(rc)
            import :: c_int, c_double
            integer(c_int), value :: ncol, nlay, icld, idcor, juldat
            real(c_double), value :: decorr_con
            real(c_double), intent(in) :: dz(*), lat(*), cldfrac(*)
            real(c_double), intent(inout) :: alpha(*)
            integer(c_int) :: rc
         end function rrtmg_lw_hip_get_alpha

         function rrtmg_lw_hip_mcica_subcol(ncol, nlay, icld, permuteseed, irng, play, cldfrac, ciwp, clwp, rei, rel, &
               tauc, alpha, cldfmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl, taucmcl) &
               bind(C, name='rrtmg_lw_hip_mcica_subcol') result(rc)
            import :: c_int, c_double
            integer(c_int), value :: ncol, nlay, icld, permuteseed
            integer(c_int), intent(inout) :: irng
            real(c_double), intent(in) :: play(*), cldfrac(*), ciwp(*), clwp(*), rei(*), rel(*), tauc(*), alpha(*)
            real(c_double), intent(inout) :: cldfmcl(*), ciwpmcl(*), clwpmcl(*), reicmcl(*), relqmcl(*), taucmcl(*)
            integer(c_int) :: rc
         end function rrtmg_lw_hip_mcica_subcol
      end interface

      contains

      subroutine get_alpha(iplon, ncol, nlayers, icld, idcor, decorr_con, &
                           dz, lat, juldat, cldfrac, alpha)

      integer(kind=im), intent(in) :: iplon           ! column/longitude index (unused, as in the reference)
      integer(kind=im), intent(in) :: ncol            ! number of columns
      integer(kind=im), intent(in) :: nlayers         ! number of model layers
      integer(kind=im), intent(in) :: icld            ! clear/cloud, cloud overlap flag
      integer(kind=im), intent(in) :: idcor           ! decorrelation length method (0 constant, 1 latitude-varying)
      integer(kind=im), intent(in) :: juldat          ! Julian day of year
      real(kind=rb), intent(in) :: decorr_con         ! decorrelation length, constant (m)
      real(kind=rb), intent(in) :: dz(:,:)            ! layer thickness (m)                 (ncol,nlayers)
      real(kind=rb), intent(in) :: lat(:)             ! latitude (degrees)                  (ncol)
      real(kind=rb), intent(in) :: cldfrac(:,:)       ! layer cloud fraction                (ncol,nlayers)
      real(kind=rb), intent(out) :: alpha(:,:)        ! vertical cloud fraction correlation parameter

      integer(c_int) :: rc
      real(c_double), allocatable :: a(:,:)

      allocate(a(ncol, nlayers))
      a = alpha(1:ncol, 1:nlayers)                    ! untouched unless icld is 4 or 5
      rc = rrtmg_lw_hip_get_alpha(int(ncol, c_int), int(nlayers, c_int), int(icld, c_int), int(idcor, c_int), &
            real(decorr_con, c_double), dz(1:ncol, 1:nlayers), lat(1:ncol), int(juldat, c_int), &
            cldfrac(1:ncol, 1:nlayers), a)
      if (rc /= 0) call rrtmg_lw_hip_abort('get_alpha')
      alpha(1:ncol, 1:nlayers) = a

      end subroutine get_alpha

      subroutine mcica_subcol_lw(iplon, ncol, nlay, icld, permuteseed, irng, play, &
                       cldfrac, ciwp, clwp, rei, rel, tauc, alpha, cldfmcl, &
                       ciwpmcl, clwpmcl, reicmcl, relqmcl, taucmcl)

      integer(kind=im), intent(in) :: iplon           ! column/longitude index (unused, as in the reference)
      integer(kind=im), intent(in) :: ncol            ! number of columns
      integer(kind=im), intent(in) :: nlay            ! number of model layers
      integer(kind=im), intent(in) :: icld            ! clear/cloud, cloud overlap flag
      integer(kind=im), intent(in) :: permuteseed     ! offsets the random stream between calls (LW vs SW: >= 140 apart)
      integer(kind=im), intent(inout) :: irng         ! 0 = kissvec, 1 = Mersenne Twister
      real(kind=rb), intent(in) :: play(:,:)          ! layer pressures (mb)                (ncol,nlay)
      real(kind=rb), intent(in) :: cldfrac(:,:)       ! layer cloud fraction                (ncol,nlay)
      real(kind=rb), intent(in) :: tauc(:,:,:)        ! in-cloud optical depth              (nbndlw,ncol,nlay)
      real(kind=rb), intent(in) :: ciwp(:,:), clwp(:,:), rei(:,:), rel(:,:)
      real(kind=rb), intent(in) :: alpha(:,:)         ! cloud fraction correlation parameter (ncol,nlay)
      real(kind=rb), intent(out) :: cldfmcl(:,:,:)    ! cloud fraction [mcica]              (ngptlw,ncol,nlay)
      real(kind=rb), intent(out) :: ciwpmcl(:,:,:), clwpmcl(:,:,:), taucmcl(:,:,:)
      real(kind=rb), intent(out) :: relqmcl(:,:), reicmcl(:,:)

      integer(c_int) :: rc, irng_c
      integer :: ng                                   ! sub-columns = g-points of the linked library (ngptlw: 140, or 256)

      if (icld == 0) return                            ! src/mcica_subcol_gen_lw.f90:265
      irng_c = int(irng, c_int)
      ng = int(rrtmg_lw_hip_gpoints())
      if (size(cldfmcl,1) < ng .or. size(cldfmcl,2) < ncol .or. size(cldfmcl,3) < nlay) then
         write(*,'(a,i0,a,i0,a,i0,a)') 'mcica_subcol_lw: the sub-column arrays are smaller than (', ng, ',', ncol, ',', nlay, ')'
         error stop 1
      endif
      ! sections (1:ngptlw, 1:ncol, 1:nlay): exactly-sized arrays (322 KB per 72-layer column) are filled in place, oversized or
      ! strided ones through the compiler's packed temporaries
      rc = rrtmg_lw_hip_mcica_subcol(int(ncol, c_int), int(nlay, c_int), int(icld, c_int), int(permuteseed, c_int), irng_c, &
            play(1:ncol, 1:nlay), cldfrac(1:ncol, 1:nlay), ciwp(1:ncol, 1:nlay), clwp(1:ncol, 1:nlay), &
            rei(1:ncol, 1:nlay), rel(1:ncol, 1:nlay), tauc(1:16, 1:ncol, 1:nlay), alpha(1:ncol, 1:nlay), &
            cldfmcl(1:ng, 1:ncol, 1:nlay), ciwpmcl(1:ng, 1:ncol, 1:nlay), clwpmcl(1:ng, 1:ncol, 1:nlay), &
            reicmcl(1:ncol, 1:nlay), relqmcl(1:ncol, 1:nlay), taucmcl(1:ng, 1:ncol, 1:nlay))
      if (rc /= 0) call rrtmg_lw_hip_abort('mcica_subcol_lw')
      irng = int(irng_c, im)

      end subroutine mcica_subcol_lw

      end module mcica_subcol_gen_lw
