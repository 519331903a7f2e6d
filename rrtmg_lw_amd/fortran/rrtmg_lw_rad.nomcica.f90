!  Drop-in replacement for the reference's module rrtmg_lw_rad, non-McICA flavour
!  (src/rrtmg_lw_rad.nomcica.f90:99-108): same module name, subroutine name, argument order, kinds,
!  intents and assumed-shape dummies, so a host model's
!      use rrtmg_lw_rad, only: rrtmg_lw
!      call rrtmg_lw(ncol, nlay, icld, idrv, play, plev, ..., uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt)
!  compiles unchanged.  The body forwards to rrtmg_lw_hip_run_nomcica (include/rrtmg_lw_hip.h) the sections
!  (1:ncol, 1:nlay[+1]) of its arguments - what the reference's loops touch - so oversized (pcols > ncol) and
!  strided actuals behave as with the reference.
      module rrtmg_lw_rad

      use iso_c_binding
      use parkind, only : im => kind_im, rb => kind_rb
      use rrtmg_lw_init, only : rrtmg_lw_hip_abort

      implicit none

      public :: rrtmg_lw

      interface
         function rrtmg_lw_hip_run_nomcica(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, &
               h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, &
               inflglw, iceflglw, liqflglw, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, &
               uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt) bind(C, name='rrtmg_lw_hip_run_nomcica') result(rc)
            import :: c_int, c_double, c_ptr
            integer(c_int), value :: ncol, nlay, idrv, inflglw, iceflglw, liqflglw
            integer(c_int), intent(inout) :: icld
            real(c_double), intent(in) :: play(*), plev(*), tlay(*), tlev(*), tsfc(*), h2ovmr(*), o3vmr(*), co2vmr(*)
            real(c_double), intent(in) :: ch4vmr(*), n2ovmr(*), o2vmr(*), cfc11vmr(*), cfc12vmr(*), cfc22vmr(*), ccl4vmr(*)
            real(c_double), intent(in) :: emis(*), cldfr(*), taucld(*), cicewp(*), cliqwp(*), reice(*), reliq(*), tauaer(*)
            real(c_double), intent(out) :: uflx(*), dflx(*), hr(*), uflxc(*), dflxc(*), hrc(*)
            type(c_ptr), value :: duflx_dt, duflxc_dt
            integer(c_int) :: rc
         end function rrtmg_lw_hip_run_nomcica
      end interface

      contains

      subroutine rrtmg_lw &
            (ncol    ,nlay    ,icld    ,idrv    , &
             play    ,plev    ,tlay    ,tlev    ,tsfc    , &
             h2ovmr  ,o3vmr   ,co2vmr  ,ch4vmr  ,n2ovmr  ,o2vmr , &
             cfc11vmr,cfc12vmr,cfc22vmr,ccl4vmr ,emis    , &
             inflglw ,iceflglw,liqflglw,cldfr   , &
             taucld  ,cicewp  ,cliqwp  ,reice   ,reliq   , &
             tauaer  , &
             uflx    ,dflx    ,hr      ,uflxc   ,dflxc,  hrc, &
             duflx_dt,duflxc_dt )

      integer(kind=im), intent(in) :: ncol            ! Number of horizontal columns
      integer(kind=im), intent(in) :: nlay            ! Number of model layers
      integer(kind=im), intent(inout) :: icld         ! Cloud overlap method
      integer(kind=im), intent(in) :: idrv            ! Flag for calculation of dFdT
      real(kind=rb), intent(in) :: play(:,:)          ! Layer pressures (hPa, mb)           (ncol,nlay)
      real(kind=rb), intent(in) :: plev(:,:)          ! Interface pressures (hPa, mb)       (ncol,nlay+1)
      real(kind=rb), intent(in) :: tlay(:,:)          ! Layer temperatures (K)
      real(kind=rb), intent(in) :: tlev(:,:)          ! Interface temperatures (K)
      real(kind=rb), intent(in) :: tsfc(:)            ! Surface temperature (K)
      real(kind=rb), intent(in) :: h2ovmr(:,:), o3vmr(:,:), co2vmr(:,:), ch4vmr(:,:), n2ovmr(:,:), o2vmr(:,:)
      real(kind=rb), intent(in) :: cfc11vmr(:,:), cfc12vmr(:,:), cfc22vmr(:,:), ccl4vmr(:,:)
      real(kind=rb), intent(in) :: emis(:,:)          ! Surface emissivity                  (ncol,nbndlw)
      integer(kind=im), intent(in) :: inflglw, iceflglw, liqflglw
      real(kind=rb), intent(in) :: cldfr(:,:)         ! Cloud fraction                      (ncol,nlay)
      real(kind=rb), intent(in) :: cicewp(:,:), cliqwp(:,:), reice(:,:), reliq(:,:)
      real(kind=rb), intent(in) :: taucld(:,:,:)      ! In-cloud optical depth              (nbndlw,ncol,nlay)
      real(kind=rb), intent(in) :: tauaer(:,:,:)      ! Aerosol optical depth               (ncol,nlay,nbndlw)
      real(kind=rb), intent(out) :: uflx(:,:), dflx(:,:), hr(:,:), uflxc(:,:), dflxc(:,:), hrc(:,:)
      real(kind=rb), intent(out), optional, target :: duflx_dt(:,:), duflxc_dt(:,:)

      integer(c_int) :: rc, icld_c
      real(c_double), allocatable, target :: d1(:,:), d2(:,:)
      type(c_ptr) :: p1, p2
      logical :: inplace

      icld_c = int(icld, c_int)
      p1 = c_null_ptr
      p2 = c_null_ptr
      inplace = .false.
      if (idrv == 1) then
         if (.not. (present(duflx_dt) .and. present(duflxc_dt))) then
            write(*,*) 'rrtmg_lw: idrv = 1 requires duflx_dt and duflxc_dt'
            error stop 1
         endif
         ! exactly sized, contiguous actuals go across in place like every other array (and stay page-locked if the host pinned them:
         ! rrtmg_lw_pin); an optional dummy cannot be passed as a section, so anything else goes through a temporary
         inplace = size(duflx_dt,1) == ncol .and. size(duflx_dt,2) == nlay+1 .and. is_contiguous(duflx_dt) .and. &
                   size(duflxc_dt,1) == ncol .and. size(duflxc_dt,2) == nlay+1 .and. is_contiguous(duflxc_dt)
         if (inplace) then
            p1 = c_loc(duflx_dt)
            p2 = c_loc(duflxc_dt)
         else
            if (size(duflx_dt,1) < ncol .or. size(duflx_dt,2) < nlay+1 .or. size(duflxc_dt,1) < ncol .or. size(duflxc_dt,2) < nlay+1) then
               write(*,*) 'rrtmg_lw: duflx_dt / duflxc_dt smaller than (ncol, nlay+1)'
               error stop 1
            endif
            allocate(d1(ncol, nlay+1), d2(ncol, nlay+1))
            p1 = c_loc(d1)
            p2 = c_loc(d2)
         endif
      endif
      ! The reference indexes play(iplon,lay) etc. (src/rrtmg_lw_rad.nomcica.f90:785-910), so a host may pass arrays that are
      ! larger than (ncol,nlay) - e.g. dimensioned (pcols,pver) with ncol < pcols - or strided sections.  The C side strides by
      ! exactly ncol: every array goes across as the section the reference would touch (the compiler packs / unpacks when that
      ! section is not contiguous; exactly-sized contiguous arrays are passed in place).
      call check_extent('play', size(play,1), size(play,2), ncol, nlay)
      call check_extent('plev', size(plev,1), size(plev,2), ncol, nlay+1)
      call check_extent('tauaer', size(tauaer,1), size(tauaer,2), ncol, nlay)
      call check_extent('taucld', size(taucld,2), size(taucld,3), ncol, nlay)
      call check_extent('uflx', size(uflx,1), size(uflx,2), ncol, nlay+1)
      call check_extent('hr', size(hr,1), size(hr,2), ncol, nlay)
      rc = rrtmg_lw_hip_run_nomcica(int(ncol, c_int), int(nlay, c_int), icld_c, int(idrv, c_int), &
            play(1:ncol,1:nlay), plev(1:ncol,1:nlay+1), tlay(1:ncol,1:nlay), tlev(1:ncol,1:nlay+1), tsfc(1:ncol), &
            h2ovmr(1:ncol,1:nlay), o3vmr(1:ncol,1:nlay), co2vmr(1:ncol,1:nlay), ch4vmr(1:ncol,1:nlay), &
            n2ovmr(1:ncol,1:nlay), o2vmr(1:ncol,1:nlay), cfc11vmr(1:ncol,1:nlay), cfc12vmr(1:ncol,1:nlay), &
            cfc22vmr(1:ncol,1:nlay), ccl4vmr(1:ncol,1:nlay), emis(1:ncol,1:16), &
            int(inflglw, c_int), int(iceflglw, c_int), int(liqflglw, c_int), &
            cldfr(1:ncol,1:nlay), taucld(1:16,1:ncol,1:nlay), cicewp(1:ncol,1:nlay), cliqwp(1:ncol,1:nlay), &
            reice(1:ncol,1:nlay), reliq(1:ncol,1:nlay), tauaer(1:ncol,1:nlay,1:16), &
            uflx(1:ncol,1:nlay+1), dflx(1:ncol,1:nlay+1), hr(1:ncol,1:nlay), &
            uflxc(1:ncol,1:nlay+1), dflxc(1:ncol,1:nlay+1), hrc(1:ncol,1:nlay), p1, p2)
      if (rc /= 0) call rrtmg_lw_hip_abort('rrtmg_lw')
      icld = int(icld_c, im)
      if (idrv == 1 .and. .not. inplace) then
         duflx_dt(1:ncol, 1:nlay+1) = d1
         duflxc_dt(1:ncol, 1:nlay+1) = d2
      endif

      end subroutine rrtmg_lw

      subroutine check_extent(name, n1, n2, ncol, nl)
      character(len=*), intent(in) :: name
      integer, intent(in) :: n1, n2
      integer(kind=im), intent(in) :: ncol, nl
      if (n1 < ncol .or. n2 < nl) then
         write(*,'(a,a,a,i0,a,i0,a,i0,a,i0,a)') 'rrtmg_lw: ', name, ' has extents (', n1, ',', n2, &
               '), smaller than (', ncol, ',', nl, ')'
         error stop 1
      endif
      end subroutine check_extent

      end module rrtmg_lw_rad
