! Test driver: an OpenMP "host model" that calls rrtmg_lw - unchanged, the reference's own interface - per chunk of a few columns from
! several threads at once (SURVEY.md 8b: the reference is serial inside a call, hosts thread OVER calls).  First one thread, chunk after
! chunk; then the same chunks from all threads.  Prints both wall times; the outputs of the threaded pass are written for the test
! (tests/test_fortran_shim.py) and must equal the serial pass bit for bit.
program drive_omp
  use omp_lib
  use parkind, only: im => kind_im, rb => kind_rb
  use rrtmg_lw_init, only: rrtmg_lw_ini
  use rrtmg_lw_rad, only: rrtmg_lw
  implicit none
  type chunk_t
     integer(im) :: n, icld
     real(rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:), gas(:,:,:), emis(:,:)
     real(rb), allocatable :: cld(:,:,:), taucld(:,:,:), tauaer(:,:,:)
     real(rb), allocatable :: uflx(:,:), dflx(:,:), hr(:,:), uflxc(:,:), dflxc(:,:), hrc(:,:), du(:,:), duc(:,:)
  end type chunk_t
  integer(im) :: ncol, nlay, icld, idrv, inflg, iceflg, liqflg, nchunk, nc, c0, k
  integer :: hdr(7), u, rep, nrep
  integer(8) :: t0, t1, rate
  real(rb), allocatable :: play(:,:), plev(:,:), tlay(:,:), tlev(:,:), tsfc(:), gas(:,:,:), emis(:,:)
  real(rb), allocatable :: cld(:,:,:), taucld(:,:,:), tauaer(:,:,:)
  real(rb), allocatable :: uflx(:,:), dflx(:,:), hr(:,:), uflxc(:,:), dflxc(:,:), hrc(:,:), du(:,:), duc(:,:)
  real(rb), allocatable :: ref_uflx(:,:), ref_hr(:,:), ref_dflxc(:,:)
  type(chunk_t), allocatable, target :: ch(:)
  character(len=512) :: fin, fout, arg
  real(rb) :: dmax, ms_serial, ms_omp

  call get_command_argument(1, fin)
  call get_command_argument(2, fout)
  call get_command_argument(3, arg)
  read(arg, *) nc
  nrep = 3
  open(newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
  read(u) hdr
  ncol = hdr(1); nlay = hdr(2); icld = hdr(3); idrv = hdr(4); inflg = hdr(5); iceflg = hdr(6); liqflg = hdr(7)
  allocate(play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay), tlev(ncol,nlay+1), tsfc(ncol), gas(ncol,nlay,10))
  allocate(emis(ncol,16), cld(ncol,nlay,5), taucld(16,ncol,nlay), tauaer(ncol,nlay,16))
  read(u) play, plev, tlay, tlev, tsfc, gas, emis, cld, taucld, tauaer
  close(u)
  allocate(uflx(ncol,nlay+1), dflx(ncol,nlay+1), hr(ncol,nlay), uflxc(ncol,nlay+1), dflxc(ncol,nlay+1), hrc(ncol,nlay))
  allocate(du(ncol,nlay+1), duc(ncol,nlay+1), ref_uflx(ncol,nlay+1), ref_hr(ncol,nlay), ref_dflxc(ncol,nlay+1))
  du = 0._rb; duc = 0._rb

  nchunk = (ncol + nc - 1) / nc
  allocate(ch(nchunk))
  do k = 1, nchunk
     c0 = (k - 1) * nc
     ch(k)%n = min(nc, ncol - c0)
     associate (n => ch(k)%n, a => c0 + 1, b => c0 + ch(k)%n)
       allocate(ch(k)%play(n,nlay), ch(k)%plev(n,nlay+1), ch(k)%tlay(n,nlay), ch(k)%tlev(n,nlay+1), ch(k)%tsfc(n), ch(k)%gas(n,nlay,10))
       allocate(ch(k)%emis(n,16), ch(k)%cld(n,nlay,5), ch(k)%taucld(16,n,nlay), ch(k)%tauaer(n,nlay,16))
       allocate(ch(k)%uflx(n,nlay+1), ch(k)%dflx(n,nlay+1), ch(k)%hr(n,nlay), ch(k)%uflxc(n,nlay+1), ch(k)%dflxc(n,nlay+1), ch(k)%hrc(n,nlay))
       allocate(ch(k)%du(n,nlay+1), ch(k)%duc(n,nlay+1))
       ch(k)%play = play(a:b,:); ch(k)%plev = plev(a:b,:); ch(k)%tlay = tlay(a:b,:); ch(k)%tlev = tlev(a:b,:); ch(k)%tsfc = tsfc(a:b)
       ch(k)%gas = gas(a:b,:,:); ch(k)%emis = emis(a:b,:); ch(k)%cld = cld(a:b,:,:); ch(k)%taucld = taucld(:,a:b,:); ch(k)%tauaer = tauaer(a:b,:,:)
       ch(k)%du = 0._rb; ch(k)%duc = 0._rb
     end associate
  enddo

  call rrtmg_lw_ini(1004.0_rb)
  call system_clock(count_rate=rate)

  ! (1) one thread, chunk after chunk (the last repetition is the timed one)
  do rep = 1, 2
     call system_clock(t0)
     do k = 1, nchunk
        call one_chunk(k)
     enddo
     call system_clock(t1)
  enddo
  ms_serial = 1.e3_rb * real(t1 - t0, rb) / real(rate, rb)
  do k = 1, nchunk
     c0 = (k - 1) * nc
     ref_uflx(c0+1:c0+ch(k)%n,:) = ch(k)%uflx
     ref_hr(c0+1:c0+ch(k)%n,:) = ch(k)%hr
     ref_dflxc(c0+1:c0+ch(k)%n,:) = ch(k)%dflxc
     ch(k)%uflx = -1._rb; ch(k)%hr = -1._rb; ch(k)%dflxc = -1._rb
  enddo

  ! (2) the same calls from every thread of the team
  do rep = 1, nrep
     call system_clock(t0)
     !$omp parallel do schedule(dynamic, 1)
     do k = 1, nchunk
        call one_chunk(k)
     enddo
     !$omp end parallel do
     call system_clock(t1)
  enddo
  ms_omp = 1.e3_rb * real(t1 - t0, rb) / real(rate, rb)

  dmax = 0._rb
  do k = 1, nchunk
     c0 = (k - 1) * nc
     associate (a => c0 + 1, b => c0 + ch(k)%n)
       uflx(a:b,:) = ch(k)%uflx; dflx(a:b,:) = ch(k)%dflx; hr(a:b,:) = ch(k)%hr
       uflxc(a:b,:) = ch(k)%uflxc; dflxc(a:b,:) = ch(k)%dflxc; hrc(a:b,:) = ch(k)%hrc; du(a:b,:) = ch(k)%du; duc(a:b,:) = ch(k)%duc
       dmax = max(dmax, maxval(abs(ch(k)%uflx - ref_uflx(a:b,:))), maxval(abs(ch(k)%hr - ref_hr(a:b,:))), maxval(abs(ch(k)%dflxc - ref_dflxc(a:b,:))))
     end associate
  enddo
  write(*,'(a,i0,a,i0,a,i0,a,f10.3,a,f10.3,a,f12.1,a,es10.3)') 'threads=', omp_get_max_threads(), ' chunks=', nchunk, ' columns_per_chunk=', nc, &
        ' serial_ms=', ms_serial, ' omp_ms=', ms_omp, ' omp_columns_per_s=', real(ncol, rb) / (1.e-3_rb * ms_omp), ' max_abs_diff_omp_vs_serial=', dmax

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
  write(u) int(ch(1)%icld), uflx, dflx, hr, uflxc, dflxc, hrc, du, duc
  close(u)

contains
  subroutine one_chunk(k)
    integer(im), intent(in) :: k
    ch(k)%icld = icld
    call rrtmg_lw(ch(k)%n, nlay, ch(k)%icld, idrv, ch(k)%play, ch(k)%plev, ch(k)%tlay, ch(k)%tlev, ch(k)%tsfc, &
                  ch(k)%gas(:,:,1), ch(k)%gas(:,:,2), ch(k)%gas(:,:,3), ch(k)%gas(:,:,4), ch(k)%gas(:,:,5), ch(k)%gas(:,:,6), &
                  ch(k)%gas(:,:,7), ch(k)%gas(:,:,8), ch(k)%gas(:,:,9), ch(k)%gas(:,:,10), ch(k)%emis, inflg, iceflg, liqflg, &
                  ch(k)%cld(:,:,1), ch(k)%taucld, ch(k)%cld(:,:,2), ch(k)%cld(:,:,3), ch(k)%cld(:,:,4), ch(k)%cld(:,:,5), ch(k)%tauaer, &
                  ch(k)%uflx, ch(k)%dflx, ch(k)%hr, ch(k)%uflxc, ch(k)%dflxc, ch(k)%hrc, ch(k)%du, ch(k)%duc)
  end subroutine one_chunk
end program drive_omp
