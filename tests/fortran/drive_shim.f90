! Test driver: a "host model" that uses the drop-in modules exactly as it would use the reference's
! (use rrtmg_lw_init / use rrtmg_lw_rad), reading its inputs from a stream file written by tests/test_fortran_shim.py.
program drive_shim
  use parkind, only: im => kind_im, rb => kind_rb
  use rrtmg_lw_init, only: rrtmg_lw_ini
  use rrtmg_lw_rad, only: rrtmg_lw
  implicit none
  integer(im) :: ncol, nlay, icld, idrv, inflg, iceflg, liqflg
  integer :: hdr(7), u
  real(rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:), gas(:,:,:), emis(:,:)
  real(rb), allocatable :: cld(:,:,:), taucld(:,:,:), tauaer(:,:,:)
  real(rb), allocatable :: uflx(:,:), dflx(:,:), hr(:,:), uflxc(:,:), dflxc(:,:), hrc(:,:), du(:,:), duc(:,:)
  character(len=512) :: fin, fout

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
  ! gas order in the file: h2o, o3, co2, ch4, n2o, o2, cfc11, cfc12, cfc22, ccl4 ; cld: cldfr, cicewp, cliqwp, reice, reliq
  call rrtmg_lw(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, &
                gas(:,:,1), gas(:,:,2), gas(:,:,3), gas(:,:,4), gas(:,:,5), gas(:,:,6), &
                gas(:,:,7), gas(:,:,8), gas(:,:,9), gas(:,:,10), emis, inflg, iceflg, liqflg, &
                cld(:,:,1), taucld, cld(:,:,2), cld(:,:,3), cld(:,:,4), cld(:,:,5), tauaer, &
                uflx, dflx, hr, uflxc, dflxc, hrc, du, duc)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) int(icld), uflx, dflx, hr, uflxc, dflxc, hrc, du, duc
  close(u)
end program drive_shim
