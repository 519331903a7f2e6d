! Test driver: a "host model" that uses the drop-in modules exactly as it would use the reference's
! (use rrtmg_lw_init / use rrtmg_lw_rad), reading its inputs from a stream file written by tests/test_fortran_shim.py.
program drive_shim
  use parkind, only: im => kind_im, rb => kind_rb
  use rrtmg_lw_init, only: rrtmg_lw_ini, rrtmg_lw_pin, rrtmg_lw_unpin, rrtmg_lw_static, rrtmg_lw_changed
  use rrtmg_lw_rad, only: rrtmg_lw
  implicit none
  integer(im) :: ncol, nlay, icld, idrv, inflg, iceflg, liqflg, icld0
  integer :: hdr(7), u
  real(rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:), gas(:,:,:), emis(:,:)
  real(rb), allocatable :: cld(:,:,:), taucld(:,:,:), tauaer(:,:,:)
  real(rb), allocatable :: uflx(:,:), dflx(:,:), hr(:,:), uflxc(:,:), dflxc(:,:), hrc(:,:), du(:,:), duc(:,:)
  character(len=512) :: fin, fout
  character(len=8) :: pin
  integer :: lpin

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) hdr
  ncol = hdr(1); nlay = hdr(2); icld = hdr(3); idrv = hdr(4); inflg = hdr(5); iceflg = hdr(6); liqflg = hdr(7)
  allocate(play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay), tlev(ncol,nlay+1), tsfc(ncol), gas(ncol,nlay,10))
  allocate(emis(ncol,16), cld(ncol,nlay,5), taucld(16,ncol,nlay), tauaer(ncol,nlay,16))
  read(u) play, plev, tlay, tlev, tsfc, gas, emis, cld, taucld, tauaer
  close(u)
  allocate(uflx(ncol,nlay+1), dflx(ncol,nlay+1), hr(ncol,nlay), uflxc(ncol,nlay+1), dflxc(ncol,nlay+1), hrc(ncol,nlay))
  allocate(du(ncol,nlay+1), duc(ncol,nlay+1))
  du = 0._rb; duc = 0._rb

  call rrtmg_lw_ini(1004.0_rb)
  ! DRIVE_PIN=1: a host model that page-locks its persistent arrays once (rrtmg_lw_pin); the copies are then direct DMA
  call get_environment_variable('DRIVE_PIN', pin, lpin)
  if (lpin > 0) then
     call rrtmg_lw_pin(play, size(play)); call rrtmg_lw_pin(plev, size(plev)); call rrtmg_lw_pin(tlay, size(tlay))
     call rrtmg_lw_pin(tlev, size(tlev)); call rrtmg_lw_pin(tsfc, size(tsfc)); call rrtmg_lw_pin(gas, size(gas))
     call rrtmg_lw_pin(emis, size(emis)); call rrtmg_lw_pin(cld, size(cld)); call rrtmg_lw_pin(taucld, size(taucld))
     call rrtmg_lw_pin(tauaer, size(tauaer)); call rrtmg_lw_pin(uflx, size(uflx)); call rrtmg_lw_pin(dflx, size(dflx))
     call rrtmg_lw_pin(hr, size(hr)); call rrtmg_lw_pin(uflxc, size(uflxc)); call rrtmg_lw_pin(dflxc, size(dflxc))
     call rrtmg_lw_pin(hrc, size(hrc)); call rrtmg_lw_pin(du, size(du)); call rrtmg_lw_pin(duc, size(duc))
     ! ... and declares what it sets once static (rrtmg_lw_static): a first call fills the scan cache, the aerosol optical depths are then
     ! scaled and the change announced (rrtmg_lw_changed), scaled back and announced again - the call below must see the original values
     call rrtmg_lw_static(tauaer); call rrtmg_lw_static(emis); call rrtmg_lw_static(gas)
     icld0 = icld
     call rrtmg_lw(ncol, nlay, icld0, idrv, play, plev, tlay, tlev, tsfc, &
                   gas(:,:,1), gas(:,:,2), gas(:,:,3), gas(:,:,4), gas(:,:,5), gas(:,:,6), &
                   gas(:,:,7), gas(:,:,8), gas(:,:,9), gas(:,:,10), emis, inflg, iceflg, liqflg, &
                   cld(:,:,1), taucld, cld(:,:,2), cld(:,:,3), cld(:,:,4), cld(:,:,5), tauaer, &
                   uflx, dflx, hr, uflxc, dflxc, hrc, du, duc)
     tauaer = 4._rb * tauaer
     call rrtmg_lw_changed(tauaer)
     icld0 = icld
     call rrtmg_lw(ncol, nlay, icld0, idrv, play, plev, tlay, tlev, tsfc, &
                   gas(:,:,1), gas(:,:,2), gas(:,:,3), gas(:,:,4), gas(:,:,5), gas(:,:,6), &
                   gas(:,:,7), gas(:,:,8), gas(:,:,9), gas(:,:,10), emis, inflg, iceflg, liqflg, &
                   cld(:,:,1), taucld, cld(:,:,2), cld(:,:,3), cld(:,:,4), cld(:,:,5), tauaer, &
                   uflx, dflx, hr, uflxc, dflxc, hrc, du, duc)
     tauaer = 0.25_rb * tauaer
     call rrtmg_lw_changed(tauaer)
  endif
  ! gas order in the file: h2o, o3, co2, ch4, n2o, o2, cfc11, cfc12, cfc22, ccl4 ; cld: cldfr, cicewp, cliqwp, reice, reliq
  call rrtmg_lw(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, &
                gas(:,:,1), gas(:,:,2), gas(:,:,3), gas(:,:,4), gas(:,:,5), gas(:,:,6), &
                gas(:,:,7), gas(:,:,8), gas(:,:,9), gas(:,:,10), emis, inflg, iceflg, liqflg, &
                cld(:,:,1), taucld, cld(:,:,2), cld(:,:,3), cld(:,:,4), cld(:,:,5), tauaer, &
                uflx, dflx, hr, uflxc, dflxc, hrc, du, duc)

  if (lpin > 0) then
     call rrtmg_lw_unpin(play); call rrtmg_lw_unpin(plev); call rrtmg_lw_unpin(tlay); call rrtmg_lw_unpin(tlev); call rrtmg_lw_unpin(tsfc)
     call rrtmg_lw_unpin(gas); call rrtmg_lw_unpin(emis); call rrtmg_lw_unpin(cld); call rrtmg_lw_unpin(taucld); call rrtmg_lw_unpin(tauaer)
     call rrtmg_lw_unpin(uflx); call rrtmg_lw_unpin(dflx); call rrtmg_lw_unpin(hr); call rrtmg_lw_unpin(uflxc); call rrtmg_lw_unpin(dflxc)
     call rrtmg_lw_unpin(hrc); call rrtmg_lw_unpin(du); call rrtmg_lw_unpin(duc)
     call rrtmg_lw_changed(tauaer, .false.); call rrtmg_lw_changed(emis, .false.); call rrtmg_lw_changed(gas, .false.)
  endif
  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) int(icld), uflx, dflx, hr, uflxc, dflxc, hrc, du, duc
  close(u)
end program drive_shim
