! Test driver for the McICA flavour: the sequence a host model runs every radiation step with the reference -
! get_alpha, mcica_subcol_lw (module mcica_subcol_gen_lw), then the McICA rrtmg_lw (module rrtmg_lw_rad) -
! against the drop-in modules.  Inputs come from a stream file written by tests/test_fortran_shim.py.
program drive_shim_mcica
  use parkind, only: im => kind_im, rb => kind_rb
  use rrtmg_lw_init, only: rrtmg_lw_ini
  use mcica_subcol_gen_lw, only: get_alpha, mcica_subcol_lw
  use rrtmg_lw_rad, only: rrtmg_lw
  implicit none
  integer(im) :: ncol, nlay, icld, idrv, inflg, iceflg, liqflg, permuteseed, irng, idcor, juldat
  integer :: hdr(11), u
  real(rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:), gas(:,:,:), emis(:,:)
  real(rb), allocatable :: cld(:,:,:), taucld(:,:,:), tauaer(:,:,:), dz(:,:), lat(:), alpha(:,:)
  real(rb), allocatable :: cldfmcl(:,:,:), ciwpmcl(:,:,:), clwpmcl(:,:,:), taucmcl(:,:,:), reicmcl(:,:), relqmcl(:,:)
  real(rb), allocatable :: uflx(:,:), dflx(:,:), hr(:,:), uflxc(:,:), dflxc(:,:), hrc(:,:), du(:,:), duc(:,:)
  character(len=512) :: fin, fout

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) hdr
  ncol = hdr(1); nlay = hdr(2); icld = hdr(3); idrv = hdr(4); inflg = hdr(5); iceflg = hdr(6); liqflg = hdr(7)
  permuteseed = hdr(8); irng = hdr(9); idcor = hdr(10); juldat = hdr(11)
  allocate(play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay), tlev(ncol,nlay+1), tsfc(ncol), gas(ncol,nlay,10))
  allocate(emis(ncol,16), cld(ncol,nlay,5), taucld(16,ncol,nlay), tauaer(ncol,nlay,16), dz(ncol,nlay), lat(ncol), alpha(ncol,nlay))
  read(u) play, plev, tlay, tlev, tsfc, gas, emis, cld, taucld, tauaer, dz, lat
  close(u)
  allocate(cldfmcl(140,ncol,nlay), ciwpmcl(140,ncol,nlay), clwpmcl(140,ncol,nlay), taucmcl(140,ncol,nlay))
  allocate(reicmcl(ncol,nlay), relqmcl(ncol,nlay))
  allocate(uflx(ncol,nlay+1), dflx(ncol,nlay+1), hr(ncol,nlay), uflxc(ncol,nlay+1), dflxc(ncol,nlay+1), hrc(ncol,nlay))
  allocate(du(ncol,nlay+1), duc(ncol,nlay+1))
  du = 0._rb; duc = 0._rb; alpha = 0._rb

  call rrtmg_lw_ini(1004.0_rb)
  ! cld: cldfr, cicewp, cliqwp, reice, reliq
  call get_alpha(1_im, ncol, nlay, icld, idcor, 2500.0_rb, dz, lat, juldat, cld(:,:,1), alpha)
  call mcica_subcol_lw(1_im, ncol, nlay, icld, permuteseed, irng, play, cld(:,:,1), cld(:,:,2), cld(:,:,3), &
                       cld(:,:,4), cld(:,:,5), taucld, alpha, cldfmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl, taucmcl)
  call rrtmg_lw(ncol, nlay, icld, idrv, play, plev, tlay, tlev, tsfc, &
                gas(:,:,1), gas(:,:,2), gas(:,:,3), gas(:,:,4), gas(:,:,5), gas(:,:,6), &
                gas(:,:,7), gas(:,:,8), gas(:,:,9), gas(:,:,10), emis, inflg, iceflg, liqflg, &
                cldfmcl, taucmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl, tauaer, &
                uflx, dflx, hr, uflxc, dflxc, hrc, du, duc)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) int(icld), uflx, dflx, hr, uflxc, dflxc, hrc, du, duc, sum(cldfmcl, dim=1)
  close(u)
end program drive_shim_mcica
