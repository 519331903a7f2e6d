!  Aggregation of small calls for a host model that calls rrtmg_lw per chunk of a few columns (rrtmg_lw_hip_queue_* of
!  include/rrtmg_lw_hip.h): the chunks are RECORDED, one device pass solves them all.
!      use rrtmg_lw_queue
!      call rrtmg_lw_queue_begin(nlay, icld, idrv, inflglw, iceflglw, liqflglw)
!      do chunk = 1, nchunks
!         call rrtmg_lw_queue_add(ncol_chunk, icld, play_c, plev_c, ..., uflx_c, dflx_c, hr_c, uflxc_c, dflxc_c, hrc_c [, duflx_dt_c, duflxc_dt_c])
!      enddo
!      call rrtmg_lw_queue_flush()          ! the outputs of every chunk are filled here
!  The argument list of rrtmg_lw_queue_add is rrtmg_lw's (src/rrtmg_lw_rad.nomcica.f90:99-108) without nlay, idrv and the three cloud
!  flags, which belong to the queue.  The arrays of a chunk are NOT copied when it is added: they must be contiguous, exactly
!  (ncol_chunk, nlay[+1]) - emis (ncol_chunk,16), taucld (16,ncol_chunk,nlay), tauaer (ncol_chunk,nlay,16) - and must stay as they are
!  until the flush (explicit-shape dummies: a non-contiguous actual would be passed as a temporary copy that is gone by then).
      module rrtmg_lw_queue

      use iso_c_binding
      use parkind, only : im => kind_im, rb => kind_rb
      use rrtmg_lw_init, only : rrtmg_lw_hip_abort

      implicit none

      public :: rrtmg_lw_queue_begin, rrtmg_lw_queue_add, rrtmg_lw_queue_flush

      interface
         function rrtmg_lw_hip_queue_begin(nlay, icld, idrv, inflglw, iceflglw, liqflglw) bind(C, name='rrtmg_lw_hip_queue_begin') result(rc)
            import :: c_int
            integer(c_int), value :: nlay, icld, idrv, inflglw, iceflglw, liqflglw
            integer(c_int) :: rc
         end function rrtmg_lw_hip_queue_begin
         function rrtmg_lw_hip_queue_add(ncol, icld, play, plev, tlay, tlev, tsfc, &
               h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, &
               cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, &
               uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt) bind(C, name='rrtmg_lw_hip_queue_add') result(rc)
            import :: c_int, c_ptr
            integer(c_int), value :: ncol
            type(c_ptr), value :: icld, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr
            type(c_ptr), value :: cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer
            type(c_ptr), value :: uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt
            integer(c_int) :: rc
         end function rrtmg_lw_hip_queue_add
         function rrtmg_lw_hip_queue_flush() bind(C, name='rrtmg_lw_hip_queue_flush') result(rc)
            import :: c_int
            integer(c_int) :: rc
         end function rrtmg_lw_hip_queue_flush
      end interface

      integer(kind=im), save, private :: q_nlay = 0

      contains

      subroutine rrtmg_lw_queue_begin(nlay, icld, idrv, inflglw, iceflglw, liqflglw)
      integer(kind=im), intent(in) :: nlay, icld, idrv, inflglw, iceflglw, liqflglw
      if (rrtmg_lw_hip_queue_begin(int(nlay, c_int), int(icld, c_int), int(idrv, c_int), int(inflglw, c_int), &
                                   int(iceflglw, c_int), int(liqflglw, c_int)) /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_queue_begin')
      q_nlay = nlay
      end subroutine rrtmg_lw_queue_begin

      subroutine rrtmg_lw_queue_add &
            (ncol    ,icld    , &
             play    ,plev    ,tlay    ,tlev    ,tsfc    , &
             h2ovmr  ,o3vmr   ,co2vmr  ,ch4vmr  ,n2ovmr  ,o2vmr , &
             cfc11vmr,cfc12vmr,cfc22vmr,ccl4vmr ,emis    , &
             cldfr   ,taucld  ,cicewp  ,cliqwp  ,reice   ,reliq   , &
             tauaer  , &
             uflx    ,dflx    ,hr      ,uflxc   ,dflxc,  hrc, &
             duflx_dt,duflxc_dt )
      integer(kind=im), intent(in) :: ncol
      integer(kind=im), intent(inout), target :: icld          ! (reset at the flush like rrtmg_lw's: must stay in scope until then)
      real(kind=rb), intent(in), target :: play(ncol,q_nlay), plev(ncol,q_nlay+1), tlay(ncol,q_nlay), tlev(ncol,q_nlay+1), tsfc(ncol)
      real(kind=rb), intent(in), target :: h2ovmr(ncol,q_nlay), o3vmr(ncol,q_nlay), co2vmr(ncol,q_nlay), ch4vmr(ncol,q_nlay)
      real(kind=rb), intent(in), target :: n2ovmr(ncol,q_nlay), o2vmr(ncol,q_nlay), cfc11vmr(ncol,q_nlay), cfc12vmr(ncol,q_nlay)
      real(kind=rb), intent(in), target :: cfc22vmr(ncol,q_nlay), ccl4vmr(ncol,q_nlay), emis(ncol,16)
      real(kind=rb), intent(in), target :: cldfr(ncol,q_nlay), taucld(16,ncol,q_nlay), cicewp(ncol,q_nlay), cliqwp(ncol,q_nlay)
      real(kind=rb), intent(in), target :: reice(ncol,q_nlay), reliq(ncol,q_nlay), tauaer(ncol,q_nlay,16)
      real(kind=rb), intent(inout), target :: uflx(ncol,q_nlay+1), dflx(ncol,q_nlay+1), hr(ncol,q_nlay)
      real(kind=rb), intent(inout), target :: uflxc(ncol,q_nlay+1), dflxc(ncol,q_nlay+1), hrc(ncol,q_nlay)
      real(kind=rb), intent(inout), target, optional :: duflx_dt(ncol,q_nlay+1), duflxc_dt(ncol,q_nlay+1)
      type(c_ptr) :: pdu, pduc
      integer(c_int) :: rc
      pdu = c_null_ptr; pduc = c_null_ptr
      if (present(duflx_dt)) pdu = c_loc(duflx_dt)
      if (present(duflxc_dt)) pduc = c_loc(duflxc_dt)
      rc = rrtmg_lw_hip_queue_add(int(ncol, c_int), c_loc(icld), c_loc(play), c_loc(plev), c_loc(tlay), c_loc(tlev), c_loc(tsfc), &
            c_loc(h2ovmr), c_loc(o3vmr), c_loc(co2vmr), c_loc(ch4vmr), c_loc(n2ovmr), c_loc(o2vmr), c_loc(cfc11vmr), c_loc(cfc12vmr), &
            c_loc(cfc22vmr), c_loc(ccl4vmr), c_loc(emis), c_loc(cldfr), c_loc(taucld), c_loc(cicewp), c_loc(cliqwp), c_loc(reice), &
            c_loc(reliq), c_loc(tauaer), c_loc(uflx), c_loc(dflx), c_loc(hr), c_loc(uflxc), c_loc(dflxc), c_loc(hrc), pdu, pduc)
      if (rc /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_queue_add')
      end subroutine rrtmg_lw_queue_add

      subroutine rrtmg_lw_queue_flush()
      if (rrtmg_lw_hip_queue_flush() /= 0) call rrtmg_lw_hip_abort('rrtmg_lw_queue_flush')
      end subroutine rrtmg_lw_queue_flush

      end module rrtmg_lw_queue
