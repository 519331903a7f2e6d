! TEST INFRASTRUCTURE.  C-callable entry to the REFERENCE's own RRTATM (src/rrtatm.f, compiled where it lies by oracle/Makefile), used
! to pin rrtmg_lw_amd/atmpth.py.  It does what readprof does around the call (src/rrtmg_lw.1col.f90:896-906, :998-1002): constants into
! /CONSTS/ (the nine members the driver sets - AIRMWT stays unset, as in the reference), units into /IFIL/, the input positioned at
! record 3.1, then copies /PROFILE/, /SPECIES/ and the layer altitudes out.
subroutine ref_rrtatm(path, npath, nskip, tape6, ntape6, mxl, nlayers, pavel, tavel, pz, tz, altz, wkl, wbrodl, nmol_out) bind(C, name='ref_rrtatm')
   use iso_c_binding
   implicit none
   integer, parameter :: mxfsc = 600, mxlay = mxfsc + 3, mxmol = 39
   character(kind=c_char), intent(in) :: path(*), tape6(*)
   integer(c_long), value :: npath, nskip, ntape6, mxl
   integer(c_long), intent(out) :: nlayers, nmol_out
   real(c_double), intent(out) :: pavel(mxl), tavel(mxl), pz(0:mxl), tz(0:mxl), altz(0:mxl), wkl(7, mxl), wbrodl(mxl)
   integer :: ird, ipr, ipu, idum, nlayrs, nmolec, ixsect, ipath, ityl
   real :: pic, planckc, boltzc, clightc, avogadc, alosmtc, gasconc, radcn1c, radcn2c
   real :: pbar, tbar, pzc, tzc, coldry, amount, wn2l, cdum
   real :: dvl, wtotl, albl, adbl, avbl, h2osl, secnta, altzc
   character*4 :: ht1, ht2
   common /consts/ pic, planckc, boltzc, clightc, avogadc, alosmtc, gasconc, radcn1c, radcn2c
   common /ifil/ ird, ipr, ipu, idum(15)
   common /profile/ nlayrs, pbar(mxlay), tbar(mxlay), pzc(0:mxlay), tzc(0:mxlay)
   common /species/ coldry(mxlay), amount(mxmol, mxlay), wn2l(mxlay), cdum(mxlay), nmolec
   common /pathd1/ dvl(mxlay), wtotl(mxlay), albl(mxlay), adbl(mxlay), avbl(mxlay), h2osl(mxlay), ipath(mxlay), ityl(mxlay), &
                   secnta(mxlay), ht1, ht2, altzc(0:mxlay)
   common /xrrtatm/ ixsect
   character(len=1024) :: fn, t6
   character(len=200) :: line
   integer :: i, l, m
   fn = ' '
   t6 = ' '
   do i = 1, int(npath)
      fn(i:i) = path(i)
   end do
   do i = 1, int(ntape6)
      t6(i:i) = tape6(i)
   end do
   ! src/rrtmg_lw_init.f90:240-262 (rrlw_con), handed over as readprof does
   pic = 2.0 * asin(1.0)
   planckc = 6.62606876e-27
   boltzc = 1.3806503e-16
   clightc = 2.99792458e+10
   avogadc = 6.02214199e+23
   alosmtc = 2.6867775e+19
   gasconc = 8.31447200e+07
   radcn1c = 1.191042722e-12
   radcn2c = 1.4387752
   ird = 9
   ipr = 66
   ipu = 7
   ixsect = 0
   open (ird, file=trim(fn), form='formatted', status='old')
   do i = 1, int(nskip)
      read (ird, '(a)') line
   end do
   open (ipr, file=trim(t6), status='unknown')
   call rrtatm
   close (ird)
   close (ipr)
   nlayers = nlayrs
   nmol_out = nmolec
   do l = 1, nlayrs
      pavel(l) = pbar(l)
      tavel(l) = tbar(l)
      wbrodl(l) = wn2l(l)
      do m = 1, 7
         wkl(m, l) = amount(m, l)
      end do
   end do
   do l = 0, nlayrs
      pz(l) = pzc(l)
      tz(l) = tzc(l)
      altz(l) = altzc(l)
   end do
end subroutine ref_rrtatm
