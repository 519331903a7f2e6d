! TEST INFRASTRUCTURE - reference harness, not product code.
!
! C-callable (bind(C)) wrappers around the *reference's own* Fortran, compiled from the sources
! where they lie under /root/reference by oracle/Makefile into oracle/_ref/libref_{nomcica,mcica}.so.
! Nothing of the reference's algorithm is restated here: these wrappers only forward arguments.
!
!   ref_init               -> rrtmg_lw_ini            (src/rrtmg_lw_init.f90:47)
!   ref_rrtmg_lw           -> rrtmg_lw                (src/rrtmg_lw_rad.nomcica.f90:99 | src/rrtmg_lw_rad.f90:99,
!                                                      selected with -DREF_MCICA at build time)
!   ref_column             -> cldprop/setcoef/taumol/rtrn|rtrnmr in the order of the reference's
!                             column driver (src/rrtmg_lw.1col.f90:497-580), i.e. the post-inatm
!                             interface that the golden OUTPUT_RRTM files pin (SURVEY.md 8c).
!
! The k-data comes from an RRLWBLOB file through oracle/_ref/gen/ref_kgb_blob.f90.

module ref_harness
  use iso_c_binding
  use parkind, only: im => kind_im, rb => kind_rb
  use parrrtm, only: nbndlw, ngptlw, mxmol, maxxsec
  implicit none
contains

  subroutine ref_set_kdata_path(path, n) bind(C, name='ref_set_kdata_path')
    use ref_blob_reader, only: kdata_path
    integer(c_int), value :: n
    character(kind=c_char), intent(in) :: path(n)
    integer :: i
    kdata_path = ' '
    do i = 1, n
       kdata_path(i:i) = path(i)
    enddo
  end subroutine ref_set_kdata_path

  subroutine ref_init(cpdair) bind(C, name='ref_init')
    use rrtmg_lw_init, only: rrtmg_lw_ini
    real(c_double), value :: cpdair
    real(rb) :: c
    c = cpdair
    call rrtmg_lw_ini(c)
  end subroutine ref_init

#ifndef REF_MCICA
  subroutine ref_rrtmg_lw(ncol, nlay, icld, idrv, &
       play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
       cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, &
       cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, &
       uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt) bind(C, name='ref_rrtmg_lw')
    use rrtmg_lw_rad, only: rrtmg_lw
    integer(c_int), value :: ncol, nlay, idrv, inflglw, iceflglw, liqflglw
    integer(c_int), intent(inout) :: icld
    real(c_double), intent(in) :: play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay), tlev(ncol,nlay+1)
    real(c_double), intent(in) :: tsfc(ncol), h2ovmr(ncol,nlay), o3vmr(ncol,nlay), co2vmr(ncol,nlay)
    real(c_double), intent(in) :: ch4vmr(ncol,nlay), n2ovmr(ncol,nlay), o2vmr(ncol,nlay)
    real(c_double), intent(in) :: cfc11vmr(ncol,nlay), cfc12vmr(ncol,nlay), cfc22vmr(ncol,nlay), ccl4vmr(ncol,nlay)
    real(c_double), intent(in) :: emis(ncol,nbndlw)
    real(c_double), intent(in) :: cldfr(ncol,nlay), taucld(nbndlw,ncol,nlay), cicewp(ncol,nlay), cliqwp(ncol,nlay)
    real(c_double), intent(in) :: reice(ncol,nlay), reliq(ncol,nlay), tauaer(ncol,nlay,nbndlw)
    real(c_double), intent(out) :: uflx(ncol,nlay+1), dflx(ncol,nlay+1), hr(ncol,nlay)
    real(c_double), intent(out) :: uflxc(ncol,nlay+1), dflxc(ncol,nlay+1), hrc(ncol,nlay)
    real(c_double), intent(out) :: duflx_dt(ncol,nlay+1), duflxc_dt(ncol,nlay+1)
    integer(im) :: icld_f
    icld_f = icld
    call rrtmg_lw(int(ncol,im), int(nlay,im), icld_f, int(idrv,im), &
         play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
         cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, &
         int(inflglw,im), int(iceflglw,im), int(liqflglw,im), &
         cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, &
         uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt)
    icld = icld_f
  end subroutine ref_rrtmg_lw
#else
  subroutine ref_rrtmg_lw(ncol, nlay, icld, idrv, &
       play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
       cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw, &
       cldfmcl, taucmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl, tauaer, &
       uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt) bind(C, name='ref_rrtmg_lw')
    use rrtmg_lw_rad, only: rrtmg_lw
    integer(c_int), value :: ncol, nlay, idrv, inflglw, iceflglw, liqflglw
    integer(c_int), intent(inout) :: icld
    real(c_double), intent(in) :: play(ncol,nlay), plev(ncol,nlay+1), tlay(ncol,nlay), tlev(ncol,nlay+1)
    real(c_double), intent(in) :: tsfc(ncol), h2ovmr(ncol,nlay), o3vmr(ncol,nlay), co2vmr(ncol,nlay)
    real(c_double), intent(in) :: ch4vmr(ncol,nlay), n2ovmr(ncol,nlay), o2vmr(ncol,nlay)
    real(c_double), intent(in) :: cfc11vmr(ncol,nlay), cfc12vmr(ncol,nlay), cfc22vmr(ncol,nlay), ccl4vmr(ncol,nlay)
    real(c_double), intent(in) :: emis(ncol,nbndlw)
    real(c_double), intent(in) :: cldfmcl(ngptlw,ncol,nlay), taucmcl(ngptlw,ncol,nlay)
    real(c_double), intent(in) :: ciwpmcl(ngptlw,ncol,nlay), clwpmcl(ngptlw,ncol,nlay)
    real(c_double), intent(in) :: reicmcl(ncol,nlay), relqmcl(ncol,nlay), tauaer(ncol,nlay,nbndlw)
    real(c_double), intent(out) :: uflx(ncol,nlay+1), dflx(ncol,nlay+1), hr(ncol,nlay)
    real(c_double), intent(out) :: uflxc(ncol,nlay+1), dflxc(ncol,nlay+1), hrc(ncol,nlay)
    real(c_double), intent(out) :: duflx_dt(ncol,nlay+1), duflxc_dt(ncol,nlay+1)
    integer(im) :: icld_f
    icld_f = icld
    call rrtmg_lw(int(ncol,im), int(nlay,im), icld_f, int(idrv,im), &
         play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr, &
         cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, &
         int(inflglw,im), int(iceflglw,im), int(liqflglw,im), &
         cldfmcl, taucmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl, tauaer, &
         uflx, dflx, hr, uflxc, dflxc, hrc, duflx_dt, duflxc_dt)
    icld = icld_f
  end subroutine ref_rrtmg_lw
#endif

#ifndef REF_MCICA
  ! Prepared-column entry: the inputs are what `readprof` (column driver) or `inatm` (GCM entry)
  ! hand to the physics; wkl holds the 7 molecular column amounts (molec/cm2), wx the 4 cross-section
  ! amounts (already x1e-20), exactly as src/rrtmg_lw.1col.f90:1027-1060 leaves them.
  subroutine ref_column(nlayers, istart, iend, iout, icld, idrv, &
       pavel, tavel, pz, tz, tbound, semiss, coldry, wkl7, wbrodl, wx4, pwvcm, &
       inflag, iceflag, liqflag, cldfrac, tauc, ciwp, clwp, rei, rel, taua, &
       totuflux, totdflux, fnet, htr, totuclfl, totdclfl, fnetc, htrc, &
       dtotuflux_dt, dtotuclfl_dt, taug_out, fracs_out, ncbands_out) bind(C, name='ref_column')
    use rrlw_con, only: fluxfac, oneminus, pi
    use rrlw_wvn, only: ngb
    use rrtmg_lw_cldprop, only: cldprop
    use rrtmg_lw_setcoef, only: setcoef
    use rrtmg_lw_taumol, only: taumol
    use rrtmg_lw_rtrn, only: rtrn
    use rrtmg_lw_rtrnmr, only: rtrnmr
    integer(c_int), value :: nlayers, istart, iend, iout, icld, idrv, inflag, iceflag, liqflag
    real(c_double), intent(in) :: pavel(nlayers), tavel(nlayers), pz(0:nlayers), tz(0:nlayers)
    real(c_double), value :: tbound, pwvcm
    real(c_double), intent(in) :: semiss(nbndlw), coldry(nlayers), wkl7(7,nlayers), wbrodl(nlayers), wx4(4,nlayers)
    real(c_double), intent(in) :: cldfrac(nlayers), tauc(nbndlw,nlayers), ciwp(nlayers), clwp(nlayers)
    real(c_double), intent(in) :: rei(nlayers), rel(nlayers), taua(nlayers,nbndlw)
    real(c_double), intent(out) :: totuflux(0:nlayers), totdflux(0:nlayers), fnet(0:nlayers), htr(0:nlayers)
    real(c_double), intent(out) :: totuclfl(0:nlayers), totdclfl(0:nlayers), fnetc(0:nlayers), htrc(0:nlayers)
    real(c_double), intent(out) :: dtotuflux_dt(0:nlayers), dtotuclfl_dt(0:nlayers)
    real(c_double), intent(out) :: taug_out(nlayers,ngptlw), fracs_out(nlayers,ngptlw)
    integer(c_int), intent(out) :: ncbands_out

    integer(im) :: nl, ncbands, laytrop, k, ig
    integer(im) :: jp(nlayers), jt(nlayers), jt1(nlayers), indself(nlayers), indfor(nlayers), indminor(nlayers)
    real(rb) :: wkl(mxmol,nlayers), wx(maxxsec,nlayers)
    real(rb) :: taucloud(nlayers,nbndlw), planklay(nlayers,nbndlw), planklev(0:nlayers,nbndlw)
    real(rb) :: plankbnd(nbndlw), dplankbnd_dt(nbndlw)
    real(rb), dimension(nlayers) :: colh2o, colco2, colo3, coln2o, colco, colch4, colo2, colbrd
    real(rb), dimension(nlayers) :: fac00, fac01, fac10, fac11
    real(rb), dimension(nlayers) :: rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, rat_h2on2o, rat_h2on2o_1
    real(rb), dimension(nlayers) :: rat_h2och4, rat_h2och4_1, rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1
    real(rb), dimension(nlayers) :: selffac, selffrac, forfac, forfrac, minorfrac, scaleminor, scaleminorn2
    real(rb) :: fracs(nlayers,ngptlw), taug(nlayers,ngptlw), taut(nlayers,ngptlw)
    real(rb) :: tb, pw

    nl = nlayers
    oneminus = 1._rb - 1.e-6_rb
    pi = 2._rb * asin(1._rb)
    fluxfac = pi * 2.e4_rb
    wkl = 0._rb
    wkl(1:7,:) = wkl7
    wx = 0._rb
    wx(1:4,:) = wx4
    tb = tbound
    pw = pwvcm
    dplankbnd_dt = 0._rb
    dtotuflux_dt = 0._rb
    dtotuclfl_dt = 0._rb

    call cldprop(nl, int(inflag,im), int(iceflag,im), int(liqflag,im), cldfrac, tauc, &
                 ciwp, clwp, rei, rel, ncbands, taucloud)
    call setcoef(nl, int(istart,im), pavel, tavel, tz, tb, semiss, &
                 coldry, wkl, wbrodl, &
                 laytrop, jp, jt, jt1, planklay, planklev, plankbnd, &
                 int(idrv,im), dplankbnd_dt, &
                 colh2o, colco2, colo3, coln2o, colco, colch4, colo2, &
                 colbrd, fac00, fac01, fac10, fac11, &
                 rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, &
                 rat_h2on2o, rat_h2on2o_1, rat_h2och4, rat_h2och4_1, &
                 rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1, &
                 selffac, selffrac, indself, forfac, forfrac, indfor, &
                 minorfrac, scaleminor, scaleminorn2, indminor)
    call taumol(nl, pavel, wx, coldry, &
                laytrop, jp, jt, jt1, planklay, planklev, plankbnd, &
                colh2o, colco2, colo3, coln2o, colco, colch4, colo2, &
                colbrd, fac00, fac01, fac10, fac11, &
                rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, &
                rat_h2on2o, rat_h2on2o_1, rat_h2och4, rat_h2och4_1, &
                rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1, &
                selffac, selffrac, indself, forfac, forfrac, indfor, &
                minorfrac, scaleminor, scaleminorn2, indminor, &
                fracs, taug)
    do k = 1, nl
       do ig = 1, ngptlw
          taut(k,ig) = taug(k,ig) + taua(k,ngb(ig))
       enddo
    enddo
    if (icld .eq. 1) then
       call rtrn(nl, int(istart,im), int(iend,im), int(iout,im), pz, semiss, ncbands, &
                 cldfrac, taucloud, planklay, planklev, plankbnd, &
                 pw, fracs, taut, &
                 totuflux, totdflux, fnet, htr, &
                 totuclfl, totdclfl, fnetc, htrc, &
                 int(idrv,im), dplankbnd_dt, dtotuflux_dt, dtotuclfl_dt)
    else
       call rtrnmr(nl, int(istart,im), int(iend,im), int(iout,im), pz, semiss, ncbands, &
                   cldfrac, taucloud, planklay, planklev, plankbnd, &
                   pw, fracs, taut, &
                   totuflux, totdflux, fnet, htr, &
                   totuclfl, totdclfl, fnetc, htrc, &
                   int(idrv,im), dplankbnd_dt, dtotuflux_dt, dtotuclfl_dt)
    endif
    taug_out = taug
    fracs_out = fracs
    ncbands_out = ncbands
  end subroutine ref_column
#else
  ! McICA flavour of the prepared-column entry: cldprmc -> setcoef -> taumol -> rtrnmc (src/rrtmg_lw.1col.f90:497-580, imca = 1)
  subroutine ref_column_mc(nlayers, istart, iend, iout, icld, idrv, &
       pavel, tavel, pz, tz, tbound, semiss, coldry, wkl7, wbrodl, wx4, pwvcm, &
       inflag, iceflag, liqflag, cldfmc, taucmc_in, ciwpmc, clwpmc, reicmc, relqmc, taua, &
       totuflux, totdflux, fnet, htr, totuclfl, totdclfl, fnetc, htrc, &
       dtotuflux_dt, dtotuclfl_dt, taug_out, fracs_out, ncbands_out) bind(C, name='ref_column_mc')
    use rrlw_con, only: fluxfac, oneminus, pi
    use rrlw_wvn, only: ngb
    use rrtmg_lw_cldprmc, only: cldprmc
    use rrtmg_lw_setcoef, only: setcoef
    use rrtmg_lw_taumol, only: taumol
    use rrtmg_lw_rtrnmc, only: rtrnmc
    integer(c_int), value :: nlayers, istart, iend, iout, icld, idrv, inflag, iceflag, liqflag
    real(c_double), intent(in) :: pavel(nlayers), tavel(nlayers), pz(0:nlayers), tz(0:nlayers)
    real(c_double), value :: tbound, pwvcm
    real(c_double), intent(in) :: semiss(nbndlw), coldry(nlayers), wkl7(7,nlayers), wbrodl(nlayers), wx4(4,nlayers)
    real(c_double), intent(in) :: cldfmc(ngptlw,nlayers), taucmc_in(ngptlw,nlayers), ciwpmc(ngptlw,nlayers), clwpmc(ngptlw,nlayers)
    real(c_double), intent(in) :: reicmc(nlayers), relqmc(nlayers), taua(nlayers,nbndlw)
    real(c_double), intent(out) :: totuflux(0:nlayers), totdflux(0:nlayers), fnet(0:nlayers), htr(0:nlayers)
    real(c_double), intent(out) :: totuclfl(0:nlayers), totdclfl(0:nlayers), fnetc(0:nlayers), htrc(0:nlayers)
    real(c_double), intent(out) :: dtotuflux_dt(0:nlayers), dtotuclfl_dt(0:nlayers)
    real(c_double), intent(out) :: taug_out(nlayers,ngptlw), fracs_out(nlayers,ngptlw)
    integer(c_int), intent(out) :: ncbands_out

    integer(im) :: nl, ncbands, laytrop, k, ig
    integer(im) :: jp(nlayers), jt(nlayers), jt1(nlayers), indself(nlayers), indfor(nlayers), indminor(nlayers)
    real(rb) :: wkl(mxmol,nlayers), wx(maxxsec,nlayers)
    real(rb) :: taucmc(ngptlw,nlayers), planklay(nlayers,nbndlw), planklev(0:nlayers,nbndlw)
    real(rb) :: plankbnd(nbndlw), dplankbnd_dt(nbndlw)
    real(rb), dimension(nlayers) :: colh2o, colco2, colo3, coln2o, colco, colch4, colo2, colbrd
    real(rb), dimension(nlayers) :: fac00, fac01, fac10, fac11
    real(rb), dimension(nlayers) :: rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, rat_h2on2o, rat_h2on2o_1
    real(rb), dimension(nlayers) :: rat_h2och4, rat_h2och4_1, rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1
    real(rb), dimension(nlayers) :: selffac, selffrac, forfac, forfrac, minorfrac, scaleminor, scaleminorn2
    real(rb) :: fracs(nlayers,ngptlw), taug(nlayers,ngptlw), taut(nlayers,ngptlw)
    real(rb) :: tb, pw

    nl = nlayers
    oneminus = 1._rb - 1.e-6_rb
    pi = 2._rb * asin(1._rb)
    fluxfac = pi * 2.e4_rb
    wkl = 0._rb
    wkl(1:7,:) = wkl7
    wx = 0._rb
    wx(1:4,:) = wx4
    tb = tbound
    pw = pwvcm
    dplankbnd_dt = 0._rb
    dtotuflux_dt = 0._rb
    dtotuclfl_dt = 0._rb

    taucmc = taucmc_in
    call cldprmc(nl, int(inflag,im), int(iceflag,im), int(liqflag,im), cldfmc, ciwpmc, &
                 clwpmc, reicmc, relqmc, ncbands, taucmc)
    call setcoef(nl, int(istart,im), pavel, tavel, tz, tb, semiss, &
                 coldry, wkl, wbrodl, &
                 laytrop, jp, jt, jt1, planklay, planklev, plankbnd, &
                 int(idrv,im), dplankbnd_dt, &
                 colh2o, colco2, colo3, coln2o, colco, colch4, colo2, &
                 colbrd, fac00, fac01, fac10, fac11, &
                 rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, &
                 rat_h2on2o, rat_h2on2o_1, rat_h2och4, rat_h2och4_1, &
                 rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1, &
                 selffac, selffrac, indself, forfac, forfrac, indfor, &
                 minorfrac, scaleminor, scaleminorn2, indminor)
    call taumol(nl, pavel, wx, coldry, &
                laytrop, jp, jt, jt1, planklay, planklev, plankbnd, &
                colh2o, colco2, colo3, coln2o, colco, colch4, colo2, &
                colbrd, fac00, fac01, fac10, fac11, &
                rat_h2oco2, rat_h2oco2_1, rat_h2oo3, rat_h2oo3_1, &
                rat_h2on2o, rat_h2on2o_1, rat_h2och4, rat_h2och4_1, &
                rat_n2oco2, rat_n2oco2_1, rat_o3co2, rat_o3co2_1, &
                selffac, selffrac, indself, forfac, forfrac, indfor, &
                minorfrac, scaleminor, scaleminorn2, indminor, &
                fracs, taug)
    do k = 1, nl
       do ig = 1, ngptlw
          taut(k,ig) = taug(k,ig) + taua(k,ngb(ig))
       enddo
    enddo
    call rtrnmc(nl, int(istart,im), int(iend,im), int(iout,im), pz, semiss, ncbands, &
                cldfmc, taucmc, planklay, planklev, plankbnd, &
                pw, fracs, taut, &
                totuflux, totdflux, fnet, htr, &
                totuclfl, totdclfl, fnetc, htrc, &
                int(idrv,im), dplankbnd_dt, dtotuflux_dt, dtotuclfl_dt)
    taug_out = taug
    fracs_out = fracs
    ncbands_out = ncbands
  end subroutine ref_column_mc
#endif


  ! One-column McICA sub-column generator of the reference (src/mcica_subcol_gen_lw.1col.f90:67-168,:171-280).
  subroutine ref_get_alpha_1col(nlayers, icld, idcor, decorr_con, dz, lat, juldat, cldfrac, alpha) bind(C, name='ref_get_alpha_1col')
    use mcica_subcol_gen_lw, only: get_alpha
    integer(c_int), value :: nlayers, icld, idcor, juldat
    real(c_double), value :: decorr_con, lat
    real(c_double), intent(in) :: dz(nlayers), cldfrac(nlayers)
    real(c_double), intent(out) :: alpha(nlayers)
    real(rb) :: dc, la
    dc = decorr_con
    la = lat
    alpha = 0._rb
    call get_alpha(1_im, int(nlayers,im), int(icld,im), int(idcor,im), dc, dz, la, int(juldat,im), cldfrac, alpha)
  end subroutine ref_get_alpha_1col

  subroutine ref_mcica_subcol_1col(nlayers, icld, ims, irng, play, cldfrac, ciwp, clwp, rei, rel, tauc, alpha, &
       cldfmc, ciwpmc, clwpmc, reicmc, relqmc, taucmc) bind(C, name='ref_mcica_subcol_1col')
    use mcica_subcol_gen_lw, only: mcica_subcol_lw
    integer(c_int), value :: nlayers, icld, ims
    integer(c_int), intent(inout) :: irng
    real(c_double), intent(in) :: play(nlayers), cldfrac(nlayers), ciwp(nlayers), clwp(nlayers), rei(nlayers), rel(nlayers)
    real(c_double), intent(in) :: tauc(nbndlw,nlayers), alpha(nlayers)
    real(c_double), intent(out) :: cldfmc(ngptlw,nlayers), ciwpmc(ngptlw,nlayers), clwpmc(ngptlw,nlayers)
    real(c_double), intent(out) :: reicmc(nlayers), relqmc(nlayers), taucmc(ngptlw,nlayers)
    integer(im) :: irng_f
    irng_f = irng
    call mcica_subcol_lw(1_im, int(nlayers,im), int(icld,im), int(ims,im), irng_f, play, cldfrac, ciwp, clwp, rei, rel, &
                         tauc, alpha, cldfmc, ciwpmc, clwpmc, reicmc, relqmc, taucmc)
    irng = irng_f
  end subroutine ref_mcica_subcol_1col

end module ref_harness
