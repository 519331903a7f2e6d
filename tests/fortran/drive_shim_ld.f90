! Test driver: a host model whose arrays are LARGER than (ncol, nlay) - dimensioned (pcols, pver+..) with ncol < pcols, as
! CAM/WRF-style hosts do - and, in a second call, strided sections of still larger arrays.  The reference indexes
! play(iplon,lay) (src/rrtmg_lw_rad.nomcica.f90:785-910) and so accepts both; the drop-in modules must as well.
! Everything outside the (1:ncol, 1:nlay[+1]) part holds a poison value on input and must still hold it on output.
program drive_shim_ld
  use parkind, only: im => kind_im, rb => kind_rb
  use rrtmg_lw_init, only: rrtmg_lw_ini
  use rrtmg_lw_rad, only: rrtmg_lw
  implicit none
  real(rb), parameter :: poison = -7.77e33_rb
  integer(im) :: ncol, nlay, icld, icld2, idrv, inflg, iceflg, liqflg
  integer :: hdr(7), u, pc, pl, k, nbad
  real(rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:), gas(:,:,:), emis(:,:)
  real(rb), allocatable :: cld(:,:,:), taucld(:,:,:), tauaer(:,:,:)
  real(rb), allocatable :: Bplay(:,:), Bplev(:,:), Btlay(:,:), Btlev(:,:), Btsfc(:), Bgas(:,:,:), Bemis(:,:)
  real(rb), allocatable :: Bcld(:,:,:), Btaucld(:,:,:), Btauaer(:,:,:)
  real(rb), allocatable :: O(:,:,:), S(:,:,:)
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
  call rrtmg_lw_ini(1004.0_rb)

  ! ---- call 1: arrays dimensioned (pcols, pver+3) with pcols = ncol + 5
  pc = ncol + 5; pl = nlay + 3
  allocate(Bplay(pc,pl), Bplev(pc,pl), Btlay(pc,pl), Btlev(pc,pl), Btsfc(pc), Bgas(pc,pl,10), Bemis(pc,18))
  allocate(Bcld(pc,pl,5), Btaucld(16,pc,pl), Btauaer(pc,pl,16), O(pc,pl,8))
  Bplay = poison; Bplev = poison; Btlay = poison; Btlev = poison; Btsfc = poison; Bgas = poison; Bemis = poison
  Bcld = poison; Btaucld = poison; Btauaer = poison; O = poison
  Bplay(1:ncol,1:nlay) = play; Bplev(1:ncol,1:nlay+1) = plev; Btlay(1:ncol,1:nlay) = tlay; Btlev(1:ncol,1:nlay+1) = tlev
  Btsfc(1:ncol) = tsfc; Bgas(1:ncol,1:nlay,:) = gas; Bemis(1:ncol,1:16) = emis; Bcld(1:ncol,1:nlay,:) = cld
  Btaucld(:,1:ncol,1:nlay) = taucld; Btauaer(1:ncol,1:nlay,:) = tauaer
  icld2 = icld
  call rrtmg_lw(ncol, nlay, icld2, idrv, Bplay, Bplev, Btlay, Btlev, Btsfc, &
                Bgas(:,:,1), Bgas(:,:,2), Bgas(:,:,3), Bgas(:,:,4), Bgas(:,:,5), Bgas(:,:,6), &
                Bgas(:,:,7), Bgas(:,:,8), Bgas(:,:,9), Bgas(:,:,10), Bemis, inflg, iceflg, liqflg, &
                Bcld(:,:,1), Btaucld, Bcld(:,:,2), Bcld(:,:,3), Bcld(:,:,4), Bcld(:,:,5), Btauaer, &
                O(:,:,1), O(:,:,2), O(:,:,3), O(:,:,4), O(:,:,5), O(:,:,6), O(:,:,7), O(:,:,8))
  nbad = 0
  do k = 1, 8
     nbad = nbad + count(O(ncol+1:pc,:,k) /= poison) + count(O(1:ncol,nlay+2:pl,k) /= poison)
  enddo
  nbad = nbad + count(O(1:ncol,nlay+1,3) /= poison) + count(O(1:ncol,nlay+1,6) /= poison)     ! hr, hrc have nlay rows

  ! ---- call 2: every other row of arrays with 2*ncol rows (strided, non-contiguous sections)
  deallocate(Bplay, Bplev, Btlay, Btlev, Btsfc, Bgas, Bemis, Bcld, Btaucld, Btauaer)
  pc = 2 * ncol; pl = nlay + 1
  allocate(Bplay(pc,pl), Bplev(pc,pl), Btlay(pc,pl), Btlev(pc,pl), Btsfc(pc), Bgas(pc,pl,10), Bemis(pc,16))
  allocate(Bcld(pc,pl,5), Btaucld(16,pc,pl), Btauaer(pc,pl,16), S(pc,pl,8))
  Bplay = poison; Bplev = poison; Btlay = poison; Btlev = poison; Btsfc = poison; Bgas = poison; Bemis = poison
  Bcld = poison; Btaucld = poison; Btauaer = poison; S = poison
  Bplay(1:pc:2,1:nlay) = play; Bplev(1:pc:2,1:nlay+1) = plev; Btlay(1:pc:2,1:nlay) = tlay; Btlev(1:pc:2,1:nlay+1) = tlev
  Btsfc(1:pc:2) = tsfc; Bgas(1:pc:2,1:nlay,:) = gas; Bemis(1:pc:2,1:16) = emis; Bcld(1:pc:2,1:nlay,:) = cld
  Btaucld(:,1:pc:2,1:nlay) = taucld; Btauaer(1:pc:2,1:nlay,:) = tauaer
  icld2 = icld
  call rrtmg_lw(ncol, nlay, icld2, idrv, Bplay(1:pc:2,:), Bplev(1:pc:2,:), Btlay(1:pc:2,:), Btlev(1:pc:2,:), Btsfc(1:pc:2), &
                Bgas(1:pc:2,:,1), Bgas(1:pc:2,:,2), Bgas(1:pc:2,:,3), Bgas(1:pc:2,:,4), Bgas(1:pc:2,:,5), Bgas(1:pc:2,:,6), &
                Bgas(1:pc:2,:,7), Bgas(1:pc:2,:,8), Bgas(1:pc:2,:,9), Bgas(1:pc:2,:,10), Bemis(1:pc:2,:), inflg, iceflg, liqflg, &
                Bcld(1:pc:2,:,1), Btaucld(:,1:pc:2,:), Bcld(1:pc:2,:,2), Bcld(1:pc:2,:,3), Bcld(1:pc:2,:,4), Bcld(1:pc:2,:,5), &
                Btauaer(1:pc:2,:,:), &
                S(1:pc:2,:,1), S(1:pc:2,:,2), S(1:pc:2,:,3), S(1:pc:2,:,4), S(1:pc:2,:,5), S(1:pc:2,:,6), S(1:pc:2,:,7), S(1:pc:2,:,8))
  do k = 1, 8
     nbad = nbad + count(S(2:pc:2,:,k) /= poison)
  enddo

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) int(icld2), int(nbad)
  write(u) O(1:ncol,1:nlay+1,1), O(1:ncol,1:nlay+1,2), O(1:ncol,1:nlay,3), O(1:ncol,1:nlay+1,4), O(1:ncol,1:nlay+1,5), &
           O(1:ncol,1:nlay,6), O(1:ncol,1:nlay+1,7), O(1:ncol,1:nlay+1,8)
  write(u) S(1:pc:2,1:nlay+1,1), S(1:pc:2,1:nlay+1,2), S(1:pc:2,1:nlay,3), S(1:pc:2,1:nlay+1,4), S(1:pc:2,1:nlay+1,5), &
           S(1:pc:2,1:nlay,6), S(1:pc:2,1:nlay+1,7), S(1:pc:2,1:nlay+1,8)
  close(u)
end program drive_shim_ld
