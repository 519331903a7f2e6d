// CDNA4 (gfx950) kernels of the RRTMG_LW hot path.  Written for MI355X only.
//
// Mapping (DESIGN.md "Kernels"): the caller's arrays are (ncol, nlay) with the COLUMN index fastest
// (reference: src/rrtmg_lw_rad.nomcica.f90:219-276), so lanes = consecutive columns makes every load and
// store of profile, workspace, scratch and flux data a full-width coalesced access.  Spectral work is
// split into "band chunks": one thread owns NGC (2..8) consecutive g-points of one band for one column and
// keeps their optical depths, Planck fractions and radiances in registers, so the g-point reduction of the
// fluxes is an in-register sum in the reference's own order and needs no cross-lane traffic.
//
//   k_prep   inatm + setcoef (+ Planck sources, diffusivity secants)          1 thread / column
//   k_cloud  cldprop + cloud-overlap factors of rtrnmr                         1 thread / column
//   k_band   taumol for the chunk's g-points fused with the rtrn/rtrnmr down- and up-sweeps
//                                                                              1 thread / (column, chunk)
//   k_final  flux scaling, net flux, heating rates                            1 thread / column
//
// Reference lines are cited per routine.  No CPU fallback exists anywhere in this file.
#include <hip/hip_runtime.h>

#include "tables.hpp"

namespace rrlw {

// ------------------------------------------------------------------------------------------------
// device-side views
// ------------------------------------------------------------------------------------------------
struct DevTables {
    const double *ktab;
    const double *stat;
    BandLayout band[NBND];
    StaticLayout sl;
    double absice0[2], abscld1, absliq0;
    double delwave[NBND];
    double refrat[NBND][6];
    double heatfac, fluxfac, oneminus, bpade;
};

// per-(layer, column) double fields of the workspace, each [nlay][ncolb]
enum Field {
    F_FAC00, F_FAC01, F_FAC10, F_FAC11,
    F_COLH2O, F_COLCO2, F_COLO3, F_COLN2O, F_COLCO, F_COLCH4, F_COLO2, F_COLBRD,
    F_COLDRY, F_SELFFAC, F_SELFFRAC, F_FORFAC, F_FORFRAC, F_MINORFRAC, F_SCALEMINOR, F_SCALEMINORN2,
    F_PAVEL, F_WX1, F_WX2, F_WX3, F_WX4,
    NFIELD
};
// per-column doubles, each [ncolb]
enum PerCol { PC_PLANKBND = 0, PC_DPLANKBND = 16, PC_SECDIFF = 32, NPERCOL = 48 };
// rtrnmr overlap factors, each [(nlay+2)][ncolb], level index 0..nlay+1
enum MrFac { MR_FACCLD1, MR_FACCLD2, MR_FACCLR1, MR_FACCLR2, MR_FACCMB1, MR_FACCMB2,
             MR_FACCLD1D, MR_FACCLD2D, MR_FACCLR1D, MR_FACCLR2D, MR_FACCMB1D, MR_FACCMB2D, NMRFAC };

struct Workspace {
    int ncolb;          // column stride (batch capacity)
    int nlay;
    double *f;          // [NFIELD][nlay][ncolb]
    int *idx;           // [nlay][ncolb]   jp | jt<<6 | jt1<<9 | indself<<12 | indfor<<16 | indminor<<18
    double *planklay;   // [16][nlay][ncolb]
    double *planklev;   // [16][nlay+1][ncolb]
    double *percol;     // [NPERCOL][ncolb]
    int *laytrop;       // [ncolb]
    int *ncbands;       // [ncolb]
    double *taucloud;   // [16][nlay][ncolb]   cldprop output
    double *odcld;      // [16][nlay][ncolb]   secdiff(ib) * taucloud
    double *efcl;       // [16][nlay][ncolb]   rtrn: (1 - exp(-odcld)) * cldfrac
    double *mrfac;      // [NMRFAC][nlay+2][ncolb]
    int *cflag;         // [nlay+2][ncolb]  bit0 icldlyr, bit1 istcld, bit2 istcldd; cflag[0] bit3 = column has cloud
    double *scr[4];     // [NGCMAX][nlay][ncolb]: atrans, bbugas, atot, bbutot
    int *err;           // [1] first physics error code
};

// GCM-interface inputs (device pointers, column stride = ncol_total), reference src/rrtmg_lw_rad.nomcica.f90:219-276
struct GcmIn {
    const double *play, *plev, *tlay, *tlev, *tsfc, *h2ovmr, *o3vmr, *co2vmr, *ch4vmr, *n2ovmr, *o2vmr;
    const double *cfc11vmr, *cfc12vmr, *cfc22vmr, *ccl4vmr, *emis;
    const double *cldfr, *taucld, *cicewp, *cliqwp, *reice, *reliq, *tauaer;
};
// prepared-column inputs (device pointers, column stride = ncol_total), reference src/rrtmg_lw.1col.f90:497-580
struct ColIn {
    const double *pavel, *tavel, *pz, *tz, *tbound, *semiss, *coldry, *wkl, *wbrodl, *wx, *pwvcm;
    const double *cldfrac, *tauc, *ciwp, *clwp, *rei, *rel, *taua;
};
struct FluxOut {
    double *uflx, *dflx, *hr, *uflxc, *dflxc, *hrc, *duflx_dt, *duflxc_dt;   // stride ncol_total
    double *fnet, *fnetc;                                                     // optional (column entry)
};

enum ErrCode { E_NONE = 0, E_ICE_SMALL = 1, E_ICE_BOUNDS = 2, E_ICE_GEN_BOUNDS = 3, E_LIQ_BOUNDS = 4, E_BAD_FLAG = 5 };

#define WS_F(F, lev) W.f[((size_t)(F) * W.nlay + ((lev)-1)) * W.ncolb + col]

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ------------------------------------------------------------------------------------------------
// k_prep : inatm (src/rrtmg_lw_rad.nomcica.f90:591-919) + setcoef (src/rrtmg_lw_setcoef.f90:50-434)
//          + secdiff (src/rrtmg_lw_rtrn.f90:280-288); also zeroes the flux accumulators.
// ------------------------------------------------------------------------------------------------
template <bool GCM>
__global__ __launch_bounds__(256) void k_prep(DevTables T, Workspace W, GcmIn g, ColIn c, FluxOut out, int ncol, int col0,
                                              int nct, int idrv, int istart)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const size_t gc = (size_t)col0 + col;       // index into caller arrays
    const int nlay = W.nlay;
    const double *S = T.stat;
    const double *totplnk = S + T.sl.totplnk, *totplnkd = S + T.sl.totplnkderiv;
    const double *totplk16 = S + T.sl.totplk16, *totplk16d = S + T.sl.totplk16deriv;
    const double *preflog = S + T.sl.preflog, *tref = S + T.sl.tref;

    const double amd = 28.9660, amw = 18.0160, avogad = 6.02214199e+23, grav = 9.8066;
    const double stpfac = 296. / 1013.;

    double tbound, pz0, tz0, pwvcm = 0.0;
    double semiss[NBND];
    if (GCM) {
        tbound = g.tsfc[gc];
        pz0 = g.plev[gc];
        tz0 = g.tlev[gc];
#pragma unroll
        for (int b = 0; b < NBND; b++) semiss[b] = g.emis[gc + (size_t)nct * b];
    } else {
        tbound = c.tbound[gc];
        pz0 = c.pz[gc];
        tz0 = c.tz[gc];
        pwvcm = c.pwvcm[gc];
#pragma unroll
        for (int b = 0; b < NBND; b++) semiss[b] = c.semiss[gc + (size_t)nct * b];
    }

    // surface / level-0 Planck terms: setcoef :173-215
    const int indbound = clampi((int)(tbound - 159.), 1, 180);
    const double tbndfrac = tbound - 159. - (double)indbound;
    const int indlev0 = clampi((int)(tz0 - 159.), 1, 180);
    const double t0frac = tz0 - 159. - (double)indlev0;
#pragma unroll
    for (int b = 0; b < NBND; b++) {
        const double *tp = totplnk + 181 * b;
        double pb, pl0, dpb = 0.0;
        if (b == 15 && istart == 16) {      // :233-247 (band 16 alone: 2600-3250 cm-1 table; level 0 mixes tables)
            pb = semiss[b] * (totplk16[indbound - 1] + tbndfrac * (totplk16[indbound] - totplk16[indbound - 1]));
            if (idrv == 1) dpb = semiss[b] * (totplk16d[indbound - 1] + tbndfrac * (totplk16d[indbound] - totplk16d[indbound - 1]));
            pl0 = totplk16[indlev0 - 1] + t0frac * (tp[indlev0] - tp[indlev0 - 1]);
        } else {
            pb = semiss[b] * (tp[indbound - 1] + tbndfrac * (tp[indbound] - tp[indbound - 1]));
            if (idrv == 1) {
                const double *td = totplnkd + 181 * b;
                dpb = semiss[b] * (td[indbound - 1] + tbndfrac * (td[indbound] - td[indbound - 1]));
            }
            pl0 = tp[indlev0 - 1] + t0frac * (tp[indlev0] - tp[indlev0 - 1]);
        }
        W.percol[(size_t)(PC_PLANKBND + b) * W.ncolb + col] = pb;
        W.percol[(size_t)(PC_DPLANKBND + b) * W.ncolb + col] = dpb;
        W.planklev[((size_t)b * (nlay + 1) + 0) * W.ncolb + col] = pl0;
    }

    double amttl = 0.0, wvttl = 0.0;
    double pzlo = pz0;
    int laytrop = 0;
    for (int lay = 1; lay <= nlay; lay++) {
        const size_t gi = gc + (size_t)nct * (lay - 1);
        double pavel, tavel, tzl, coldry, wbrodl, w1, w2, w3, w4, w5, w6, w7, x1, x2, x3, x4;
        if (GCM) {
            // inatm :785-870
            pavel = g.play[gi];
            tavel = g.tlay[gi];
            const double pzl = g.plev[gc + (size_t)nct * lay];
            tzl = g.tlev[gc + (size_t)nct * lay];
            w1 = g.h2ovmr[gi]; w2 = g.co2vmr[gi]; w3 = g.o3vmr[gi]; w4 = g.n2ovmr[gi];
            w5 = 0.0; w6 = g.ch4vmr[gi]; w7 = g.o2vmr[gi];
            const double amm = (1. - w1) * amd + w1 * amw;
            coldry = (pzlo - pzl) * 1.e3 * avogad / (1.e2 * grav * amm * (1. + w1));
            pzlo = pzl;
            double summol = 0.0;
            summol = summol + w2; summol = summol + w3; summol = summol + w4;
            summol = summol + w5; summol = summol + w6; summol = summol + w7;
            wbrodl = coldry * (1. - summol);
            w1 = coldry * w1; w2 = coldry * w2; w3 = coldry * w3; w4 = coldry * w4;
            w5 = coldry * w5; w6 = coldry * w6; w7 = coldry * w7;
            amttl = amttl + coldry + w1;
            wvttl = wvttl + w1;
            x1 = coldry * g.ccl4vmr[gi] * 1.e-20;
            x2 = coldry * g.cfc11vmr[gi] * 1.e-20;
            x3 = coldry * g.cfc12vmr[gi] * 1.e-20;
            x4 = coldry * g.cfc22vmr[gi] * 1.e-20;
        } else {
            pavel = c.pavel[gi];
            tavel = c.tavel[gi];
            tzl = c.tz[gc + (size_t)nct * lay];
            coldry = c.coldry[gi];
            wbrodl = c.wbrodl[gi];
            const size_t wi = gc + (size_t)nct * 7 * (lay - 1);       // wkl (ncol,7,nlayers)
            w1 = c.wkl[wi]; w2 = c.wkl[wi + (size_t)nct]; w3 = c.wkl[wi + (size_t)nct * 2];
            w4 = c.wkl[wi + (size_t)nct * 3]; w5 = c.wkl[wi + (size_t)nct * 4]; w6 = c.wkl[wi + (size_t)nct * 5];
            w7 = c.wkl[wi + (size_t)nct * 6];
            const size_t xi = gc + (size_t)nct * 4 * (lay - 1);       // wx (ncol,4,nlayers)
            x1 = c.wx[xi]; x2 = c.wx[xi + (size_t)nct]; x3 = c.wx[xi + (size_t)nct * 2]; x4 = c.wx[xi + (size_t)nct * 3];
        }

        // Planck functions at layer and level temperatures: setcoef :189-269
        const int indlay = clampi((int)(tavel - 159.), 1, 180);
        const double tlayfrac = tavel - 159. - (double)indlay;
        const int indlev = clampi((int)(tzl - 159.), 1, 180);
        const double tlevfrac = tzl - 159. - (double)indlev;
#pragma unroll
        for (int b = 0; b < NBND; b++) {
            const double *tp = (b == 15 && istart == 16) ? totplk16 : totplnk + 181 * b;
            W.planklay[((size_t)b * nlay + (lay - 1)) * W.ncolb + col] = tp[indlay - 1] + tlayfrac * (tp[indlay] - tp[indlay - 1]);
            W.planklev[((size_t)b * (nlay + 1) + lay) * W.ncolb + col] = tp[indlev - 1] + tlevfrac * (tp[indlev] - tp[indlev - 1]);
        }

        // pressure / temperature interpolation: setcoef :276-306
        const double plog = log(pavel);
        const int jp = clampi((int)(36. - 5 * (plog + 0.04)), 1, 58);
        const double fp = 5. * (preflog[jp - 1] - plog);
        const int jt = clampi((int)(3. + (tavel - tref[jp - 1]) / 15.), 1, 4);
        const double ft = ((tavel - tref[jp - 1]) / 15.) - (double)(jt - 3);
        const int jt1 = clampi((int)(3. + (tavel - tref[jp]) / 15.), 1, 4);
        const double ft1 = ((tavel - tref[jp]) / 15.) - (double)(jt1 - 3);
        const double water = w1 / coldry;
        const double scalefac = pavel * stpfac / tavel;
        int indself = 0, indfor, indminor;
        double forfac, forfrac, selffac, selffrac = 0.0, factor;
        if (!(plog <= 4.56)) {            // :312-334
            laytrop++;
            forfac = scalefac / (1. + water);
            factor = (332.0 - tavel) / 36.0;
            indfor = min(2, max(1, (int)factor));
            forfrac = factor - (double)indfor;
            selffac = water * forfac;
            factor = (tavel - 188.0) / 7.2;
            indself = min(9, max(1, (int)factor - 7));
            selffrac = factor - (double)(indself + 7);
        } else {                          // :369-377
            forfac = scalefac / (1. + water);
            factor = (tavel - 188.0) / 36.0;
            indfor = 3;
            forfrac = factor - 1.0;
            selffac = water * forfac;
        }
        const double scaleminor = pavel / tavel;
        const double scaleminorn2 = (pavel / tavel) * (wbrodl / (coldry + w1));
        factor = (tavel - 180.8) / 7.2;
        indminor = min(18, max(1, (int)factor));
        const double minorfrac = factor - (double)indminor;

        double colh2o = 1.e-20 * w1, colco2 = 1.e-20 * w2, colo3 = 1.e-20 * w3, coln2o = 1.e-20 * w4;
        double colco = 1.e-20 * w5, colch4 = 1.e-20 * w6, colo2 = 1.e-20 * w7;
        if (colco2 == 0.) colco2 = 1.e-32 * coldry;
        if (colo3 == 0.) colo3 = 1.e-32 * coldry;
        if (coln2o == 0.) coln2o = 1.e-32 * coldry;
        if (colco == 0.) colco = 1.e-32 * coldry;
        if (colch4 == 0.) colch4 = 1.e-32 * coldry;
        const double colbrd = 1.e-20 * wbrodl;
        const double compfp = 1. - fp;     // :421-429
        WS_F(F_FAC10, lay) = compfp * ft;
        WS_F(F_FAC00, lay) = compfp * (1. - ft);
        WS_F(F_FAC11, lay) = fp * ft1;
        WS_F(F_FAC01, lay) = fp * (1. - ft1);
        WS_F(F_SELFFAC, lay) = colh2o * selffac;
        WS_F(F_FORFAC, lay) = colh2o * forfac;
        WS_F(F_SELFFRAC, lay) = selffrac;
        WS_F(F_FORFRAC, lay) = forfrac;
        WS_F(F_MINORFRAC, lay) = minorfrac;
        WS_F(F_SCALEMINOR, lay) = scaleminor;
        WS_F(F_SCALEMINORN2, lay) = scaleminorn2;
        WS_F(F_COLH2O, lay) = colh2o; WS_F(F_COLCO2, lay) = colco2; WS_F(F_COLO3, lay) = colo3;
        WS_F(F_COLN2O, lay) = coln2o; WS_F(F_COLCO, lay) = colco; WS_F(F_COLCH4, lay) = colch4;
        WS_F(F_COLO2, lay) = colo2; WS_F(F_COLBRD, lay) = colbrd; WS_F(F_COLDRY, lay) = coldry;
        WS_F(F_PAVEL, lay) = pavel;
        WS_F(F_WX1, lay) = x1; WS_F(F_WX2, lay) = x2; WS_F(F_WX3, lay) = x3; WS_F(F_WX4, lay) = x4;
        W.idx[(size_t)(lay - 1) * W.ncolb + col] = jp | (jt << 6) | (jt1 << 9) | (indself << 12) | (indfor << 16) | (indminor << 18);
    }
    W.laytrop[col] = laytrop;
    if (GCM) {
        const double wvsh = (amw * wvttl) / (amd * amttl);      // inatm :869-870
        pwvcm = wvsh * (1.e3 * pz0) / (1.e2 * grav);
    }
    // diffusivity secants: rtrn :265-288
    const double a0[16] = {1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
    const double a1[16] = {0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
    const double a2[16] = {0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
#pragma unroll
    for (int b = 0; b < NBND; b++) {
        double sd = 1.66;
        if (!(b == 0 || b == 3 || b >= 9)) {
            sd = a0[b] + a1[b] * exp(a2[b] * pwvcm);
            if (sd > 1.80) sd = 1.80;
            if (sd < 1.50) sd = 1.50;
        }
        W.percol[(size_t)(PC_SECDIFF + b) * W.ncolb + col] = sd;
    }
    W.ncbands[col] = 1;
    W.cflag[col] = 0;
    // flux accumulators
    for (int lev = 0; lev <= nlay; lev++) {
        const size_t o = gc + (size_t)nct * lev;
        out.uflx[o] = 0.0; out.dflx[o] = 0.0; out.uflxc[o] = 0.0; out.dflxc[o] = 0.0;
        if (idrv == 1) { out.duflx_dt[o] = 0.0; out.duflxc_dt[o] = 0.0; }
    }
}

// ------------------------------------------------------------------------------------------------
// k_cloud : cldprop (src/rrtmg_lw_cldprop.f90:50-295), cloud optical depth along the diffusivity
//           angle (rtrn :321-334 / rtrnmr :333-343) and the maximum-random overlap factors
//           (src/rrtmg_lw_rtrnmr.f90:347-506).  mode 1 = rtrn, 2 = rtrnmr.
//           Values the reference reads uninitialised (faccmb1/2, faccmb1d/2d: SURVEY.md 0.4) are ZERO.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int icb_map(int ib, int ind)   // cldprop :167-169 (1-based ib)
{
    if (ind == 0) return 1;
    if (ind == 2) return ib;
    return ib <= 2 ? ib : (ib <= 5 ? 3 : (ib <= 8 ? 4 : 5));
}

template <bool GCM>
__global__ __launch_bounds__(256) void k_cloud(DevTables T, Workspace W, GcmIn g, ColIn c, int ncol, int col0, int nct, int mode,
                                               int inflag, int iceflag, int liqflag)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const size_t gc = (size_t)col0 + col;
    const int nlay = W.nlay;
    const double *S = T.stat;
    const double *absice1 = S + T.sl.absice1, *absice2 = S + T.sl.absice2, *absice3 = S + T.sl.absice3, *absliq1 = S + T.sl.absliq1;
    const double *cldfr = GCM ? g.cldfr : c.cldfrac;
    const double *ciwp_ = GCM ? g.cicewp : c.ciwp, *clwp_ = GCM ? g.cliqwp : c.clwp;
    const double *rei_ = GCM ? g.reice : c.rei, *rel_ = GCM ? g.reliq : c.rel;
    const double cldmin = 1.e-20;
    int ncbands = 1, iceind = 0, liqind = 0, err = 0;
    // persistent single-coefficient slots (abscoice(1)/abscoliq(1) of the reference)
    for (int lay = 1; lay <= nlay; lay++) {
        const size_t gi = gc + (size_t)nct * (lay - 1);
        const double cf = cldfr[gi], ciwp = ciwp_[gi], clwp = clwp_[gi];
        double tauc[NBND], tauctot = 0.0;
#pragma unroll
        for (int ib = 0; ib < NBND; ib++) {
            // taucld (16,ncol,nlay) | tauc (ncol,16,nlayers)
            tauc[ib] = GCM ? g.taucld[ib + (size_t)NBND * gi] : c.tauc[gc + (size_t)nct * (ib + (size_t)NBND * (lay - 1))];
            tauctot = tauctot + tauc[ib];
        }
        double tcl[NBND];
#pragma unroll
        for (int ib = 0; ib < NBND; ib++) tcl[ib] = 0.0;
        const double cwp = ciwp + clwp;
        if (cf >= cldmin && (cwp >= cldmin || tauctot >= cldmin)) {
            if (inflag == 0) {
                ncbands = 16;
#pragma unroll
                for (int ib = 0; ib < NBND; ib++) tcl[ib] = tauc[ib];
            } else if (inflag == 1) {
                ncbands = 16;
#pragma unroll
                for (int ib = 0; ib < NBND; ib++) tcl[ib] = T.abscld1 * cwp;
            } else if (inflag == 2) {
                const double radice = rei_[gi];
                int icemode = -1;        // how abscoice(k) is evaluated for this layer
                double ice_single = 0.0, fint_i = 0.0;
                int index_i = 1;
                if (ciwp == 0.0) { icemode = 0; ice_single = 0.0; iceind = 0; }
                else if (iceflag == 0) {
                    if (radice < 10.0) err = E_ICE_SMALL;
                    icemode = 0; ice_single = T.absice0[0] + T.absice0[1] / radice; iceind = 0;
                } else if (iceflag == 1) {
                    if (radice < 13.0 || radice > 130.) err = E_ICE_BOUNDS;
                    ncbands = 5; icemode = 1; iceind = 1;
                } else if (iceflag == 2) {
                    if (radice < 5.0 || radice > 131.0) err = E_ICE_BOUNDS;
                    ncbands = 16;
                    const double factor = (radice - 2.) / 3.;
                    index_i = (int)factor; if (index_i == 43) index_i = 42;
                    fint_i = factor - (double)index_i;
                    icemode = 2; iceind = 2;
                } else if (iceflag == 3) {
                    if (radice < 5.0 || radice > 140.0) err = E_ICE_GEN_BOUNDS;
                    ncbands = 16;
                    const double factor = (radice - 2.) / 3.;
                    index_i = (int)factor; if (index_i == 46) index_i = 45;
                    fint_i = factor - (double)index_i;
                    icemode = 3; iceind = 2;
                } else err = E_BAD_FLAG;
                int liqmode = -1, index_l = 1;
                double liq_single = 0.0, fint_l = 0.0;
                if (clwp == 0.0) { liqmode = 0; liq_single = 0.0; liqind = 0; if (iceind == 1) iceind = 2; }
                else if (liqflag == 0) { liqmode = 0; liq_single = T.absliq0; liqind = 0; if (iceind == 1) iceind = 2; }
                else if (liqflag == 1) {
                    const double radliq = rel_[gi];
                    if (radliq < 2.5 || radliq > 60.) err = E_LIQ_BOUNDS;
                    index_l = (int)(radliq - 1.5);
                    if (index_l == 0) index_l = 1;
                    if (index_l == 58) index_l = 57;
                    fint_l = radliq - 1.5 - (double)index_l;
                    ncbands = 16; liqmode = 1; liqind = 2;
                } else err = E_BAD_FLAG;
                if (err) index_i = clampi(index_i, 1, 42), index_l = clampi(index_l, 1, 57);
                for (int ib = 1; ib <= ncbands; ib++) {
                    const int ki = icb_map(ib, iceind), kl = icb_map(ib, liqind);
                    double ai, al;
                    if (icemode == 0) ai = ice_single;
                    else if (icemode == 1) ai = absice1[2 * (ki - 1)] + absice1[2 * (ki - 1) + 1] / radice;
                    else if (icemode == 2) { const double *t = absice2 + 43 * (ki - 1); ai = t[index_i - 1] + fint_i * (t[index_i] - t[index_i - 1]); }
                    else if (icemode == 3) { const double *t = absice3 + 46 * (ki - 1); ai = t[index_i - 1] + fint_i * (t[index_i] - t[index_i - 1]); }
                    else ai = 0.0;
                    if (liqmode == 0) al = liq_single;
                    else if (liqmode == 1) { const double *t = absliq1 + 58 * (kl - 1); al = t[index_l - 1] + fint_l * (t[index_l] - t[index_l - 1]); }
                    else al = 0.0;
                    tcl[ib - 1] = ciwp * ai + clwp * al;
                }
            }
        }
#pragma unroll
        for (int ib = 0; ib < NBND; ib++) W.taucloud[((size_t)ib * nlay + (lay - 1)) * W.ncolb + col] = tcl[ib];
    }
    W.ncbands[col] = ncbands;
    if (err) atomicCAS(W.err, 0, err);

    // optical depth along the diffusivity angle; note secdiff is indexed by the CLOUD band ib (rtrn :323)
    int anycloud = 0;
    for (int lay = 1; lay <= nlay; lay++) {
        const double cf = cldfr[gc + (size_t)nct * (lay - 1)];
        const int cloudy = cf >= 1.e-6;
        anycloud |= cloudy;
        W.cflag[(size_t)lay * W.ncolb + col] = cloudy;
        for (int ib = 0; ib < NBND; ib++) {
            const size_t o = ((size_t)ib * nlay + (lay - 1)) * W.ncolb + col;
            double od = 0.0, ef = 0.0;
            if (cloudy && ib < ncbands) {
                od = W.percol[(size_t)(PC_SECDIFF + ib) * W.ncolb + col] * W.taucloud[o];
                if (mode == 1) ef = (1. - exp(-od)) * cf;
            }
            W.odcld[o] = od;
            if (mode == 1) W.efcl[o] = ef;
        }
    }
    W.cflag[col] = anycloud ? 8 : 0;
    W.cflag[(size_t)(nlay + 1) * W.ncolb + col] = 0;
    if (mode != 2) return;

    // ---- maximum-random overlap factors: rtrnmr :347-506 ----------------------------------------------
#define CF(l) cldfr[gc + (size_t)nct * ((l)-1)]
#define ICLD(l) (W.cflag[(size_t)(l) * W.ncolb + col] & 1)
#define MR(k, l) W.mrfac[((size_t)(k) * (nlay + 2) + (l)) * W.ncolb + col]
    for (int k = 0; k < NMRFAC; k++)
        for (int l = 0; l <= nlay + 1; l++) MR(k, l) = 0.0;
    double rat1 = 0.0, rat2 = 0.0;
    {
        int ist = 1;           // istcld(lev)
        for (int lev = 1; lev <= nlay; lev++) {
            int ist_next;
            if (ICLD(lev)) {
                ist_next = 0;
                const double cl = CF(lev);
                if (lev == nlay) {
                    // all factors at lev+1 stay zero
                } else {
                    const double cu = CF(lev + 1);
                    if (cu >= cl) {
                        // faccld1/2(lev+1) = 0
                        if (ist == 1) {
                            double v2 = 0.0;
                            if (cl < 1.) v2 = (cu - cl) / (1. - cl);
                            MR(MR_FACCLR1, lev + 1) = 0.0;
                            MR(MR_FACCLR2, lev + 1) = v2;
                            MR(MR_FACCLR2, lev) = 0.0;
                            MR(MR_FACCLD2, lev) = 0.0;
                        } else {
                            const double cd = CF(lev - 1);
                            const double fmx = fmax(cl, cd);
                            if (cu > fmx) {
                                MR(MR_FACCLR1, lev + 1) = rat2;
                                MR(MR_FACCLR2, lev + 1) = (cu - fmx) / (1. - fmx);
                            } else if (cu < fmx) {
                                MR(MR_FACCLR1, lev + 1) = (cu - cl) / (cd - cl);
                                MR(MR_FACCLR2, lev + 1) = 0.0;
                            } else {
                                MR(MR_FACCLR1, lev + 1) = rat2;
                                MR(MR_FACCLR2, lev + 1) = 0.0;
                            }
                        }
                        if (MR(MR_FACCLR1, lev + 1) > 0. || MR(MR_FACCLR2, lev + 1) > 0.) { rat1 = 1.; rat2 = 0.; }
                        else { rat1 = 0.; rat2 = 0.; }
                    } else {
                        // facclr1/2(lev+1) = 0
                        if (ist == 1) {
                            MR(MR_FACCLD1, lev + 1) = 0.0;
                            MR(MR_FACCLD2, lev + 1) = (cl - cu) / cl;
                            MR(MR_FACCLR2, lev) = 0.0;
                            MR(MR_FACCLD2, lev) = 0.0;
                        } else {
                            const double cd = CF(lev - 1);
                            const double fmn = fmin(cl, cd);
                            if (cu <= fmn) {
                                MR(MR_FACCLD1, lev + 1) = rat1;
                                MR(MR_FACCLD2, lev + 1) = (fmn - cu) / fmn;
                            } else {
                                MR(MR_FACCLD1, lev + 1) = (cl - cu) / (cl - fmn);
                                MR(MR_FACCLD2, lev + 1) = 0.0;
                            }
                        }
                        if (MR(MR_FACCLD1, lev + 1) > 0. || MR(MR_FACCLD2, lev + 1) > 0.) { rat1 = 0.; rat2 = 1.; }
                        else { rat1 = 0.; rat2 = 0.; }
                    }
                }
                if (ist != 1) {
                    const double cd = CF(lev - 1);
                    const double cu = lev < nlay ? CF(lev + 1) : 0.0;   // lev == nlay: the reference reads cldfrac(nlayers+1), out of bounds; both
                    // terms are then irrelevant because faccmb*(nlayers+1) is never used (up-sweep reads lev+1 <= nlayers only when lev < nlayers...)
                    MR(MR_FACCMB1, lev + 1) = fmax(0., fmin(cu - cl, cd - cl));
                    MR(MR_FACCMB2, lev + 1) = fmax(0., fmin(cl - cu, cl - cd));
                }
            } else {
                ist_next = 1;
            }
            W.cflag[(size_t)lev * W.ncolb + col] |= (ist << 1);
            ist = ist_next;
        }
    }
    {
        int ist = 1;           // istcldd(lev)
        for (int lev = nlay; lev >= 1; lev--) {
            int ist_next;
            if (ICLD(lev)) {
                ist_next = 0;
                const double cl = CF(lev);
                if (lev == 1) {
                    // factors at lev-1 = 0 stay zero
                } else {
                    const double cd = CF(lev - 1);
                    if (cd >= cl) {
                        if (ist == 1) {
                            double v2 = 0.0;
                            if (cl < 1.) v2 = (cd - cl) / (1. - cl);
                            MR(MR_FACCLR1D, lev - 1) = 0.0;
                            MR(MR_FACCLR2D, lev - 1) = v2;
                            MR(MR_FACCLR2D, lev) = 0.0;
                            MR(MR_FACCLD2D, lev) = 0.0;
                        } else {
                            const double cu = CF(lev + 1);
                            const double fmx = fmax(cl, cu);
                            if (cd > fmx) {
                                MR(MR_FACCLR1D, lev - 1) = rat2;
                                MR(MR_FACCLR2D, lev - 1) = (cd - fmx) / (1. - fmx);
                            } else if (cd < fmx) {
                                MR(MR_FACCLR1D, lev - 1) = (cd - cl) / (cu - cl);
                                MR(MR_FACCLR2D, lev - 1) = 0.0;
                            } else {
                                MR(MR_FACCLR1D, lev - 1) = rat2;
                                MR(MR_FACCLR2D, lev - 1) = 0.0;
                            }
                        }
                        if (MR(MR_FACCLR1D, lev - 1) > 0. || MR(MR_FACCLR2D, lev - 1) > 0.) { rat1 = 1.; rat2 = 0.; }
                        else { rat1 = 0.; rat2 = 0.; }
                    } else {
                        if (ist == 1) {
                            MR(MR_FACCLD1D, lev - 1) = 0.0;
                            MR(MR_FACCLD2D, lev - 1) = (cl - cd) / cl;
                            MR(MR_FACCLR2D, lev) = 0.0;
                            MR(MR_FACCLD2D, lev) = 0.0;
                        } else {
                            const double cu = CF(lev + 1);
                            const double fmn = fmin(cl, cu);
                            if (cd <= fmn) {
                                MR(MR_FACCLD1D, lev - 1) = rat1;
                                MR(MR_FACCLD2D, lev - 1) = (fmn - cd) / fmn;
                            } else {
                                MR(MR_FACCLD1D, lev - 1) = (cl - cd) / (cl - fmn);
                                MR(MR_FACCLD2D, lev - 1) = 0.0;
                            }
                        }
                        if (MR(MR_FACCLD1D, lev - 1) > 0. || MR(MR_FACCLD2D, lev - 1) > 0.) { rat1 = 0.; rat2 = 1.; }
                        else { rat1 = 0.; rat2 = 0.; }
                    }
                }
                if (ist != 1) {
                    const double cu = CF(lev + 1);
                    const double cd = lev > 1 ? CF(lev - 1) : 0.0;
                    MR(MR_FACCMB1D, lev - 1) = fmax(0., fmin(cu - cl, cd - cl));
                    MR(MR_FACCMB2D, lev - 1) = fmax(0., fmin(cl - cu, cl - cd));
                }
            } else {
                ist_next = 1;
            }
            W.cflag[(size_t)lev * W.ncolb + col] |= (ist << 2);
            ist = ist_next;
        }
    }
#undef CF
#undef ICLD
#undef MR
}

// ------------------------------------------------------------------------------------------------
// Band descriptions for taumol (src/rrtmg_lw_taumol.f90:299-3164; compact spec: SURVEY.md appendix B)
// ------------------------------------------------------------------------------------------------
enum Sp { H2O = 0, CO2, O3, N2O, CO, CH4, O2, BRD };                 // F_COLH2O + Sp
enum Key { K_NONE, K_SINGLE, K_BINARY, K_ZERO };
enum Amt { A_COL, A_ADJ, A_BRD_N2, A_O2, A_BRD };
enum Rat { R_H2OCO2 = 0, R_H2OO3, R_H2ON2O, R_H2OCH4, R_N2OCO2, R_O3CO2 };
enum Corr { C_NONE, C_B1LO, C_B1UP, C_B2LO };

struct Minor {
    Amt amt; int sp; bool two_d; int refslot;
    double thr, base, expo;    // A_ADJ parameters
    double chiconst;           // > 0: literal reference mixing ratio (band 13); else chi_mls(sp, jp+1)
};
struct Region {
    Key key; int a, b, rat;
    bool self_, for_;
    int planck_slot;           // >= 0: Planck fractions interpolated in the mixture (refrat slot); -1: constant
    bool frac_from_a;          // upper region takes its (constant) fractions from fracrefa (band 6)
    int nm; Minor m[3];
    int ncfc; int cfc_wx[2];
    Corr corr;
    int mult;                  // 0 none, 4 = band-4 upper multipliers, 7 = band-7 upper multipliers
};
constexpr Minor NOM = {A_COL, 0, false, 0, 0, 0, 0, 0};
constexpr Region ZERO_REGION = {K_ZERO, 0, 0, 0, false, false, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0};

template <int B> struct BT;
#define BAND_TRAITS(B_, NG_, LO, UP) \
    template <> struct BT<B_> { static constexpr int ng = NG_; static constexpr Region lo = LO; static constexpr Region up = UP; };

#define R1(...) Region{__VA_ARGS__}
BAND_TRAITS(1, 10,
    R1(K_SINGLE, H2O, 0, 0, true, true, -1, false, 1, {{A_BRD_N2, BRD, false, 0, 0, 0, 0, 0}, NOM, NOM}, 0, {0, 0}, C_B1LO, 0),
    R1(K_SINGLE, H2O, 0, 0, false, true, -1, false, 1, {{A_BRD_N2, BRD, false, 0, 0, 0, 0, 0}, NOM, NOM}, 0, {0, 0}, C_B1UP, 0))
BAND_TRAITS(2, 12,
    R1(K_SINGLE, H2O, 0, 0, true, true, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_B2LO, 0),
    R1(K_SINGLE, H2O, 0, 0, false, true, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
BAND_TRAITS(3, 16,
    R1(K_BINARY, H2O, CO2, R_H2OCO2, true, true, 0, false, 1, {{A_ADJ, N2O, true, 2, 1.5, 0.5, 0.65, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_BINARY, H2O, CO2, R_H2OCO2, false, true, 1, false, 1, {{A_ADJ, N2O, true, 3, 1.5, 0.5, 0.65, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
BAND_TRAITS(4, 14,
    R1(K_BINARY, H2O, CO2, R_H2OCO2, true, true, 0, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_BINARY, O3, CO2, R_O3CO2, false, false, 1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 4))
BAND_TRAITS(5, 16,
    R1(K_BINARY, H2O, CO2, R_H2OCO2, true, true, 0, false, 1, {{A_COL, O3, true, 2, 0, 0, 0, 0}, NOM, NOM}, 1, {1, 0}, C_NONE, 0),
    R1(K_BINARY, O3, CO2, R_O3CO2, false, false, 1, false, 0, {NOM, NOM, NOM}, 1, {1, 0}, C_NONE, 0))
BAND_TRAITS(6, 8,
    R1(K_SINGLE, H2O, 0, 0, true, true, -1, false, 1, {{A_ADJ, CO2, false, 0, 3.0, 2.0, 0.77, 0}, NOM, NOM}, 2, {2, 3}, C_NONE, 0),
    R1(K_NONE, 0, 0, 0, false, false, -1, true, 0, {NOM, NOM, NOM}, 2, {2, 3}, C_NONE, 0))
BAND_TRAITS(7, 12,
    R1(K_BINARY, H2O, O3, R_H2OO3, true, true, 0, false, 1, {{A_ADJ, CO2, true, 2, 3.0, 3.0, 0.79, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_SINGLE, O3, 0, 0, false, false, -1, false, 1, {{A_ADJ, CO2, false, 0, 3.0, 2.0, 0.79, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 7))
BAND_TRAITS(8, 8,
    R1(K_SINGLE, H2O, 0, 0, true, true, -1, false, 3,
       {{A_ADJ, CO2, false, 0, 3.0, 2.0, 0.65, 0}, {A_COL, O3, false, 0, 0, 0, 0, 0}, {A_COL, N2O, false, 0, 0, 0, 0, 0}}, 2, {3, 4}, C_NONE, 0),
    R1(K_SINGLE, O3, 0, 0, false, false, -1, false, 2,
       {{A_ADJ, CO2, false, 0, 3.0, 2.0, 0.65, 0}, {A_COL, N2O, false, 0, 0, 0, 0, 0}, NOM}, 2, {3, 4}, C_NONE, 0))
BAND_TRAITS(9, 12,
    R1(K_BINARY, H2O, CH4, R_H2OCH4, true, true, 0, false, 1, {{A_ADJ, N2O, true, 2, 1.5, 0.5, 0.65, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_SINGLE, CH4, 0, 0, false, false, -1, false, 1, {{A_ADJ, N2O, false, 0, 1.5, 0.5, 0.65, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
BAND_TRAITS(10, 6,
    R1(K_SINGLE, H2O, 0, 0, true, true, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_SINGLE, H2O, 0, 0, false, true, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
BAND_TRAITS(11, 8,
    R1(K_SINGLE, H2O, 0, 0, true, true, -1, false, 1, {{A_O2, O2, false, 0, 0, 0, 0, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_SINGLE, H2O, 0, 0, false, true, -1, false, 1, {{A_O2, O2, false, 0, 0, 0, 0, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
BAND_TRAITS(12, 8,
    R1(K_BINARY, H2O, CO2, R_H2OCO2, true, true, 0, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    ZERO_REGION)
BAND_TRAITS(13, 4,
    R1(K_BINARY, H2O, N2O, R_H2ON2O, true, true, 0, false, 2,
       {{A_ADJ, CO2, true, 2, 3.0, 2.0, 0.68, 3.55e-4}, {A_COL, CO, true, 4, 0, 0, 0, 0}, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_NONE, 0, 0, 0, false, false, -1, false, 1, {{A_COL, O3, false, 0, 0, 0, 0, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
BAND_TRAITS(14, 2,
    R1(K_SINGLE, CO2, 0, 0, true, true, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_SINGLE, CO2, 0, 0, false, false, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
BAND_TRAITS(15, 2,
    R1(K_BINARY, N2O, CO2, R_N2OCO2, true, true, 0, false, 1, {{A_BRD, BRD, true, 2, 0, 0, 0, 0}, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    ZERO_REGION)
BAND_TRAITS(16, 2,
    R1(K_BINARY, H2O, CH4, R_H2OCH4, true, true, 0, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0),
    R1(K_SINGLE, CH4, 0, 0, false, false, -1, false, 0, {NOM, NOM, NOM}, 0, {0, 0}, C_NONE, 0))
#undef R1

__device__ const double kMult4[14] = {1, 1, 1, 1, 1, 1, 1, 0.92, 0.88, 1.07, 1.1, 0.99, 0.88, 0.943};   // taumol :1028-1034
__device__ const double kMult7[12] = {1, 1, 1, 1, 1, 0.92, 0.88, 1.07, 1.1, 0.99, 0.855, 1};             // taumol :1664-1669

// load NGC consecutive doubles (16-byte aligned when NGC is even: rows have even length and g0 is even)
template <int NGC>
__device__ __forceinline__ void ldrow(const double *__restrict__ p, double (&v)[NGC])
{
    if constexpr (NGC % 2 == 0) {
        const double2 *q = reinterpret_cast<const double2 *>(p);
#pragma unroll
        for (int j = 0; j < NGC / 2; j++) { const double2 t = q[j]; v[2 * j] = t.x; v[2 * j + 1] = t.y; }
    } else {
#pragma unroll
        for (int j = 0; j < NGC; j++) v[j] = p[j];
    }
}

// acc[j] = first ? w * row[j] : acc[j] + w * row[j]
template <int NGC>
__device__ __forceinline__ void axpy(double (&acc)[NGC], double w, const double *__restrict__ row, bool first)
{
    double v[NGC];
    ldrow<NGC>(row, v);
#pragma unroll
    for (int j = 0; j < NGC; j++) acc[j] = first ? w * v[j] : acc[j] + w * v[j];
}

// out[j] = r0[j] + frac * (r1[j] - r0[j]),  r1 = r0 + stride
template <int NGC>
__device__ __forceinline__ void lerp_rows(double (&o)[NGC], const double *__restrict__ r0, int stride, double frac)
{
    double a[NGC], b[NGC];
    ldrow<NGC>(r0, a);
    ldrow<NGC>(r0 + stride, b);
#pragma unroll
    for (int j = 0; j < NGC; j++) o[j] = a[j] + frac * (b[j] - a[j]);
}

struct Spec { double speccomb, specparm, fs; int js; };
__device__ __forceinline__ Spec spec_calc(double cola, double rat, double colb, double mult, double oneminus)
{
    Spec s;                                         // taumol :523-528
    s.speccomb = cola + rat * colb;
    s.specparm = cola / s.speccomb;
    if (s.specparm >= oneminus) s.specparm = oneminus;
    const double specmult = mult * s.specparm;
    s.js = 1 + (int)specmult;
    s.fs = specmult - (double)(int)specmult;        // mod(specmult, 1.0) for specmult >= 0
    return s;
}

// lower-atmosphere binary stencil, branch-free form of taumol :569-598 / :641-663:
// rows ind+off+{0,1,2} and ind+off+9+{0,1,2} with weights w[0..5] (unused points get weight 0)
__device__ __forceinline__ void stencil6(double specparm, double fs, double fa, double fb, int &off, double (&w)[6])
{
    if (specparm < 0.125) {
        const double p = fs - 1, p2 = p * p, p4 = p2 * p2;
        const double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
        off = 0;
        w[0] = fk0 * fa; w[1] = fk1 * fa; w[2] = fk2 * fa; w[3] = fk0 * fb; w[4] = fk1 * fb; w[5] = fk2 * fb;
    } else if (specparm > 0.875) {
        const double p = -fs, p2 = p * p, p4 = p2 * p2;
        const double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
        off = -1;
        w[0] = fk2 * fa; w[1] = fk1 * fa; w[2] = fk0 * fa; w[3] = fk2 * fb; w[4] = fk1 * fb; w[5] = fk0 * fb;
    } else {
        off = 0;
        w[0] = (1. - fs) * fa; w[1] = fs * fa; w[2] = 0.0; w[3] = (1. - fs) * fb; w[4] = fs * fb; w[5] = 0.0;
    }
}

// Gaseous optical depth and Planck fraction of NGC g-points (g0 .. g0+NGC-1 of band B) for one layer.
template <int B, int NGC, bool LOWER>
__device__ __forceinline__ void taumol_layer(const DevTables &T, const Workspace &W, int lay, int col, int g0, int packed,
                                             double (&tau)[NGC], double (&frac)[NGC])
{
    constexpr Region R = LOWER ? BT<B>::lo : BT<B>::up;
    constexpr int ng = BT<B>::ng;
    if constexpr (R.key == K_ZERO) {
#pragma unroll
        for (int j = 0; j < NGC; j++) { tau[j] = 0.0; frac[j] = 0.0; }
        return;
    } else {
        const BandLayout &L = T.band[B - 1];
        const double *__restrict__ kt = T.ktab + g0;
        const int jp = packed & 63, jt = (packed >> 6) & 7, jt1 = (packed >> 9) & 7;
        const int indself = (packed >> 12) & 15, indfor = (packed >> 16) & 3, indminor = (packed >> 18) & 31;
        const double minorfrac = (R.nm > 0) ? WS_F(F_MINORFRAC, lay) : 0.0;
        const double *rat_tab = T.stat + T.sl.rat;
        const double *chi_tab = T.stat + T.sl.chi;

        // ---- key species --------------------------------------------------------------------------------
        if constexpr (R.key == K_SINGLE) {
            const double fac00 = WS_F(F_FAC00, lay), fac01 = WS_F(F_FAC01, lay), fac10 = WS_F(F_FAC10, lay), fac11 = WS_F(F_FAC11, lay);
            const int r0 = LOWER ? ((jp - 1) * 5 + (jt - 1)) : ((jp - 13) * 5 + (jt - 1));
            const int r1 = LOWER ? (jp * 5 + (jt1 - 1)) : ((jp - 12) * 5 + (jt1 - 1));
            const double *tab = kt + (LOWER ? L.absa : L.absb);
            double acc[NGC];
            axpy<NGC>(acc, fac00, tab + (size_t)r0 * ng, true);
            axpy<NGC>(acc, fac10, tab + (size_t)(r0 + 1) * ng, false);
            axpy<NGC>(acc, fac01, tab + (size_t)r1 * ng, false);
            axpy<NGC>(acc, fac11, tab + (size_t)(r1 + 1) * ng, false);
            const double colk = WS_F(F_COLH2O + R.a, lay);
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = colk * acc[j];
        } else if constexpr (R.key == K_BINARY) {
            const double fac00 = WS_F(F_FAC00, lay), fac01 = WS_F(F_FAC01, lay), fac10 = WS_F(F_FAC10, lay), fac11 = WS_F(F_FAC11, lay);
            const double cola = WS_F(F_COLH2O + R.a, lay), colb = WS_F(F_COLH2O + R.b, lay);
            const double rat = rat_tab[R.rat * 59 + (jp - 1)], rat_1 = rat_tab[R.rat * 59 + jp];
            constexpr double mult = LOWER ? 8. : 4.;
            const Spec s = spec_calc(cola, rat, colb, mult, T.oneminus);
            const Spec s1 = spec_calc(cola, rat_1, colb, mult, T.oneminus);
            double acc[NGC];
            if constexpr (LOWER) {
                const double *tab = kt + L.absa;
                const int ind0 = ((jp - 1) * 5 + (jt - 1)) * 9 + s.js - 1;      // 0-based row
                const int ind1 = (jp * 5 + (jt1 - 1)) * 9 + s1.js - 1;
                int off;
                double w[6];
                stencil6(s.specparm, s.fs, fac00, fac10, off, w);
                const double *r = tab + (size_t)(ind0 + off) * ng;
                axpy<NGC>(acc, w[0], r, true);
                axpy<NGC>(acc, w[1], r + ng, false);
                axpy<NGC>(acc, w[2], r + 2 * ng, false);
                axpy<NGC>(acc, w[3], r + 9 * ng, false);
                axpy<NGC>(acc, w[4], r + 10 * ng, false);
                axpy<NGC>(acc, w[5], r + 11 * ng, false);
#pragma unroll
                for (int j = 0; j < NGC; j++) tau[j] = s.speccomb * acc[j];
                stencil6(s1.specparm, s1.fs, fac01, fac11, off, w);
                r = tab + (size_t)(ind1 + off) * ng;
                axpy<NGC>(acc, w[0], r, true);
                axpy<NGC>(acc, w[1], r + ng, false);
                axpy<NGC>(acc, w[2], r + 2 * ng, false);
                axpy<NGC>(acc, w[3], r + 9 * ng, false);
                axpy<NGC>(acc, w[4], r + 10 * ng, false);
                axpy<NGC>(acc, w[5], r + 11 * ng, false);
#pragma unroll
                for (int j = 0; j < NGC; j++) tau[j] = tau[j] + s1.speccomb * acc[j];
            } else {                                                              // taumol :751-771
                const double *tab = kt + L.absb;
                const int ind0 = ((jp - 13) * 5 + (jt - 1)) * 5 + s.js - 1;
                const int ind1 = ((jp - 12) * 5 + (jt1 - 1)) * 5 + s1.js - 1;
                const double *r = tab + (size_t)ind0 * ng;
                axpy<NGC>(acc, (1. - s.fs) * fac00, r, true);
                axpy<NGC>(acc, s.fs * fac00, r + ng, false);
                axpy<NGC>(acc, (1. - s.fs) * fac10, r + 5 * ng, false);
                axpy<NGC>(acc, s.fs * fac10, r + 6 * ng, false);
#pragma unroll
                for (int j = 0; j < NGC; j++) tau[j] = s.speccomb * acc[j];
                r = tab + (size_t)ind1 * ng;
                axpy<NGC>(acc, (1. - s1.fs) * fac01, r, true);
                axpy<NGC>(acc, s1.fs * fac01, r + ng, false);
                axpy<NGC>(acc, (1. - s1.fs) * fac11, r + 5 * ng, false);
                axpy<NGC>(acc, s1.fs * fac11, r + 6 * ng, false);
#pragma unroll
                for (int j = 0; j < NGC; j++) tau[j] = tau[j] + s1.speccomb * acc[j];
            }
            // Planck fractions interpolated in the mixture: taumol :556-561, :692-693
            const Spec sp = spec_calc(cola, T.refrat[B - 1][R.planck_slot], colb, mult, T.oneminus);
            lerp_rows<NGC>(frac, kt + (LOWER ? L.fracrefa : L.fracrefb) + (size_t)(sp.js - 1) * ng, ng, sp.fs);
        } else {
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = 0.0;
        }
        if constexpr (R.key != K_BINARY) {
            ldrow<NGC>(kt + ((LOWER || R.frac_from_a) ? L.fracrefa : L.fracrefb), frac);
        }

        // ---- water-vapour continua: taumol :350-353 -------------------------------------------------------
        if constexpr (R.self_) {
            double t[NGC];
            lerp_rows<NGC>(t, kt + L.selfref + (size_t)(indself - 1) * ng, ng, WS_F(F_SELFFRAC, lay));
            const double selffac = WS_F(F_SELFFAC, lay);
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = tau[j] + selffac * t[j];
        }
        if constexpr (R.for_) {
            double t[NGC];
            lerp_rows<NGC>(t, kt + L.forref + (size_t)(indfor - 1) * ng, ng, WS_F(F_FORFRAC, lay));
            const double forfac = WS_F(F_FORFAC, lay);
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = tau[j] + forfac * t[j];
        }

        // ---- minor gases ---------------------------------------------------------------------------------------
#pragma unroll
        for (int im = 0; im < R.nm; im++) {
            constexpr int dummy = 0; (void)dummy;
            const Minor M = R.m[im];
            double amount;
            if (M.amt == A_COL) amount = WS_F(F_COLH2O + M.sp, lay);
            else if (M.amt == A_BRD_N2) amount = WS_F(F_COLBRD, lay) * WS_F(F_SCALEMINORN2, lay);
            else if (M.amt == A_O2) amount = WS_F(F_COLO2, lay) * WS_F(F_SCALEMINOR, lay);
            else if (M.amt == A_BRD) amount = WS_F(F_COLBRD, lay) * WS_F(F_SCALEMINOR, lay);
            else {                                                       // A_ADJ: taumol :547-554
                const double colx = WS_F(F_COLH2O + M.sp, lay), coldry = WS_F(F_COLDRY, lay);
                const double chiref = M.chiconst > 0. ? M.chiconst : chi_tab[M.sp * 59 + jp];     // chi_mls(sp+1, jp+1)
                const double chi = colx / coldry;
                const double ratx = 1.e20 * chi / chiref;
                amount = colx;
                if (ratx > M.thr) amount = (M.base + pow(ratx - M.base, M.expo)) * chiref * coldry * 1.e-20;
            }
            const int slot = LOWER ? L.minor_lo[im] : L.minor_up[im];
            double ab[NGC];
            if (M.two_d) {                                               // taumol :635-639
                constexpr int nj = LOWER ? 9 : 5;
                const Spec sm = spec_calc(WS_F(F_COLH2O + R.a, lay), T.refrat[B - 1][M.refslot], WS_F(F_COLH2O + R.b, lay), LOWER ? 8. : 4., T.oneminus);
                const double *r = kt + slot + (size_t)((indminor - 1) * nj + (sm.js - 1)) * ng;
                double m1[NGC], m2[NGC];
                lerp_rows<NGC>(m1, r, ng, sm.fs);
                lerp_rows<NGC>(m2, r + (size_t)nj * ng, ng, sm.fs);
#pragma unroll
                for (int j = 0; j < NGC; j++) ab[j] = m1[j] + minorfrac * (m2[j] - m1[j]);
            } else {
                lerp_rows<NGC>(ab, kt + slot + (size_t)(indminor - 1) * ng, ng, minorfrac);
            }
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = tau[j] + amount * ab[j];
        }

        // ---- halocarbons: taumol :1254, :1381-1382, :1753-1754 ---------------------------------------------------------
#pragma unroll
        for (int ic = 0; ic < R.ncfc; ic++) {
            const double wxv = WS_F(F_WX1 + R.cfc_wx[ic] - 1, lay);
            double v[NGC];
            ldrow<NGC>(kt + L.vec[ic], v);
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = tau[j] + wxv * v[j];
        }

        if constexpr (R.corr != C_NONE) {
            const double pp = WS_F(F_PAVEL, lay);
            double corradj = 1.;
            if constexpr (R.corr == C_B1LO) { if (pp < 250.) corradj = 1. - 0.15 * (250. - pp) / 154.4; }
            else if constexpr (R.corr == C_B1UP) corradj = 1. - 0.15 * (pp / 95.6);
            else corradj = 1. - .05 * (pp - 100.) / 900.;
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = corradj * tau[j];
        }
        if constexpr (R.mult == 4) {
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = tau[j] * kMult4[g0 + j];
        } else if constexpr (R.mult == 7) {
#pragma unroll
            for (int j = 0; j < NGC; j++) tau[j] = tau[j] * kMult7[g0 + j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_band : taumol fused with the radiative-transfer sweeps.
//   MODE 0 clear column set (icld = 0, or cloud-free call): rtrn/rtrnmr clear branch
//   MODE 1 rtrn   (random overlap)            src/rrtmg_lw_rtrn.f90:339-574
//   MODE 2 rtrnmr (maximum-random overlap)    src/rrtmg_lw_rtrnmr.f90:510-775
// Per-(layer, g) quantities needed again by the up-sweep go through `scr` ([j][lay][col], coalesced).
// Band fluxes are added to the caller's flux arrays (zeroed by k_prep); chunk kernels run in stream order
// so the additions are deterministic and follow the reference's band order.
// ------------------------------------------------------------------------------------------------
struct BandArgs {
    int ncol, col0, nct, g0, idrv;
    const double *emis;        // semiss (nct,16)
    const double *tauaer;      // (nct,nlay,16) or null
    const double *cldfrac;     // (nct,nlay)
};

__device__ __forceinline__ void gas_layer(double od, const double *__restrict__ lut, double bpade, double &atrans, double &tfn)
{
    // clear-layer transmittance: rtrn :439-451
    if (od <= 0.06) {
        atrans = od - 0.5 * od * od;
        tfn = 0.166667 * od;
    } else {
        const int itr = (int)(10000.0 * (od / (bpade + od)) + 0.5);
        const double2 e = reinterpret_cast<const double2 *>(lut)[itr];
        atrans = 1. - e.x;
        tfn = e.y;
    }
}

template <int B, int NGC, int MODE>
__global__ __launch_bounds__(256) void k_band(DevTables T, Workspace W, BandArgs a, FluxOut out)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= a.ncol) return;
    const size_t gc = (size_t)a.col0 + col;
    const int nlay = W.nlay, nct = a.nct, g0 = a.g0;
    const size_t ncb = W.ncolb;
    const double *__restrict__ lut = T.stat + T.sl.lut;
    const double *__restrict__ tau_tbl = T.stat + T.sl.tau_tbl;
    const double bpade = T.bpade;
    const double secdiff = W.percol[(size_t)(PC_SECDIFF + B - 1) * ncb + col];
    const double wtdelw = T.delwave[B - 1];
    const int laytrop = W.laytrop[col];
    const bool idrv = a.idrv == 1;
    double *__restrict__ sA = W.scr[0], *__restrict__ sB = W.scr[1], *__restrict__ sC = W.scr[2], *__restrict__ sD = W.scr[3];

    int ibc = 0;                    // 0-based cloud band for this spectral band: rtrn :343-349
    bool colcloud = false;
    if constexpr (MODE != 0) {
        const int ncbands = W.ncbands[col];
        ibc = ncbands == 1 ? 0 : (ncbands == 5 ? (B <= 2 ? B - 1 : (B <= 5 ? 2 : (B <= 8 ? 3 : 4))) : B - 1);
        colcloud = (W.cflag[col] & 8) != 0;
    }

    double radld[NGC], radclrd[NGC], frac[NGC], tau[NGC];
    double cldrad[NGC], clrrad[NGC], radmr[NGC];       // rtrnmr partial radiances
#pragma unroll
    for (int j = 0; j < NGC; j++) { radld[j] = 0.0; radclrd[j] = 0.0; cldrad[j] = 0.0; clrrad[j] = 0.0; radmr[j] = 0.0; }
    int iclddn = 0;

    double plev_hi = W.planklev[((size_t)(B - 1) * (nlay + 1) + nlay) * ncb + col];
    // ------------------------------------------------------------------ downward sweep: rtrn :361-466
    for (int lev = nlay; lev >= 1; lev--) {
        const int packed = W.idx[(size_t)(lev - 1) * ncb + col];
        if (lev <= laytrop) taumol_layer<B, NGC, true>(T, W, lev, col, g0, packed, tau, frac);
        else taumol_layer<B, NGC, false>(T, W, lev, col, g0, packed, tau, frac);
        const double taua = a.tauaer ? a.tauaer[gc + (size_t)nct * ((lev - 1) + (size_t)nlay * (B - 1))] : 0.0;
        const double blay = W.planklay[((size_t)(B - 1) * nlay + (lev - 1)) * ncb + col];
        const double plev_lo = W.planklev[((size_t)(B - 1) * (nlay + 1) + (lev - 1)) * ncb + col];
        const double dplankup = plev_hi - blay, dplankdn = plev_lo - blay;
        plev_hi = plev_lo;
        int cloudy = 0;
        double cf = 0.0, odcld = 0.0, efcl = 0.0;
        int flags = 0;
        if constexpr (MODE != 0) {
            flags = W.cflag[(size_t)lev * ncb + col];
            cloudy = flags & 1;
            if (cloudy) {
                cf = a.cldfrac[gc + (size_t)nct * (lev - 1)];
                odcld = W.odcld[((size_t)ibc * nlay + (lev - 1)) * ncb + col];
                if constexpr (MODE == 1) efcl = W.efcl[((size_t)ibc * nlay + (lev - 1)) * ncb + col];
            }
        }
        double dsum = 0.0, dsumc = 0.0;
        if (!cloudy) {
#pragma unroll
            for (int j = 0; j < NGC; j++) {
                double od = secdiff * (tau[j] + taua);
                if (od < 0.0) od = 0.0;
                double atr, tfn;
                gas_layer(od, lut, bpade, atr, tfn);
                const double bbd = frac[j] * (blay + tfn * dplankdn);
                const double bbu = frac[j] * (blay + tfn * dplankup);
                const size_t so = ((size_t)j * nlay + (lev - 1)) * ncb + col;
                sA[so] = atr;
                sB[so] = bbu;
                radld[j] = radld[j] + (bbd - radld[j]) * atr;
                dsum = dsum + radld[j];
                if constexpr (MODE != 0) {
                    if (iclddn) radclrd[j] = radclrd[j] + (bbd - radclrd[j]) * atr;
                    else radclrd[j] = radld[j];
                    dsumc = dsumc + radclrd[j];
                }
            }
        } else {
            if constexpr (MODE != 0) {
                iclddn = 1;
                double fclr1 = 0, fcld1 = 0, fcmb1 = 0, fcmb2 = 0, fclr2 = 0, fcld2 = 0;
                if constexpr (MODE == 2) {
                    const size_t mo = (size_t)(lev - 1) * ncb + col, ms = (size_t)(nlay + 2) * ncb;
                    fclr1 = W.mrfac[MR_FACCLR1D * ms + mo]; fcld1 = W.mrfac[MR_FACCLD1D * ms + mo];
                    fcmb1 = W.mrfac[MR_FACCMB1D * ms + mo]; fcmb2 = W.mrfac[MR_FACCMB2D * ms + mo];
                    fclr2 = W.mrfac[MR_FACCLR2D * ms + mo]; fcld2 = W.mrfac[MR_FACCLD2D * ms + mo];
                }
#pragma unroll
                for (int j = 0; j < NGC; j++) {
                    double od = secdiff * (tau[j] + taua);
                    if (od < 0.0) od = 0.0;
                    double odtot = od + odcld;
                    double atr, tfgas, atot, tftot;
                    // three sub-branches: rtrn :372-435
                    if (odtot < 0.06) {
                        atr = od - 0.5 * od * od; tfgas = 0.166667 * od;
                        atot = odtot - 0.5 * odtot * odtot; tftot = 0.166667 * odtot;
                    } else if (od <= 0.06) {
                        atr = od - 0.5 * od * od; tfgas = 0.166667 * od;
                        const int it = (int)(10000.0 * (odtot / (bpade + odtot)) + 0.5);
                        const double2 e = reinterpret_cast<const double2 *>(lut)[it];
                        atot = 1. - e.x; tftot = e.y;
                    } else {
                        const int ig = (int)(10000.0 * (od / (bpade + od)) + 0.5);
                        const double2 e = reinterpret_cast<const double2 *>(lut)[ig];
                        od = tau_tbl[ig];
                        atr = 1. - e.x; tfgas = e.y;
                        odtot = od + odcld;
                        const int it = (int)(10000.0 * (odtot / (bpade + odtot)) + 0.5);
                        const double2 e2 = reinterpret_cast<const double2 *>(lut)[it];
                        atot = 1. - e2.x; tftot = e2.y;
                    }
                    const double bbd = frac[j] * (blay + tfgas * dplankdn);
                    const double gassrc = bbd * atr;
                    const double bbdtot = frac[j] * (blay + tftot * dplankdn);
                    const size_t so = ((size_t)j * nlay + (lev - 1)) * ncb + col;
                    sA[so] = atr;
                    sB[so] = frac[j] * (blay + tfgas * dplankup);
                    sC[so] = atot;
                    sD[so] = frac[j] * (blay + tftot * dplankup);
                    if constexpr (MODE == 1) {
                        radld[j] = radld[j] - radld[j] * (atr + efcl * (1. - atr)) + gassrc + cf * (bbdtot * atot - gassrc);
                    } else {            // rtrnmr :591-615
                        if (flags & 4) {        // istcldd(lev) == 1
                            cldrad[j] = cf * radld[j];
                            clrrad[j] = radld[j] - cldrad[j];
                            radmr[j] = 0.0;
                        }
                        const double ttot = 1. - atot;
                        const double cldsrc = bbdtot * atot;
                        cldrad[j] = cldrad[j] * ttot + cf * cldsrc;
                        clrrad[j] = clrrad[j] * (1. - atr) + (1. - cf) * gassrc;
                        radld[j] = cldrad[j] + clrrad[j];
                        const double radmod = radmr[j] * (fclr1 * (1. - atr) + fcld1 * ttot) - fcmb1 * gassrc + fcmb2 * cldsrc;
                        const double oldcld = cldrad[j] - radmod;
                        const double oldclr = clrrad[j] + radmod;
                        radmr[j] = -radmod + fclr2 * oldclr - fcld2 * oldcld;
                        cldrad[j] = cldrad[j] + radmr[j];
                        clrrad[j] = clrrad[j] - radmr[j];
                    }
                    dsum = dsum + radld[j];
                    radclrd[j] = radclrd[j] + (bbd - radclrd[j]) * atr;
                    dsumc = dsumc + radclrd[j];
                }
            }
        }
        const size_t oo = gc + (size_t)nct * (lev - 1);
        out.dflx[oo] += (dsum * 0.5) * wtdelw;
        if constexpr (MODE != 0) out.dflxc[oo] += (dsumc * 0.5) * wtdelw;
    }

    // ------------------------------------------------------------------ surface: rtrn :476-495
    const double plankbnd = W.percol[(size_t)(PC_PLANKBND + B - 1) * ncb + col];
    const double dplankbnd = idrv ? W.percol[(size_t)(PC_DPLANKBND + B - 1) * ncb + col] : 0.0;
    const double reflect = 1. - a.emis[gc + (size_t)nct * (B - 1)];
    double radlu[NGC], radclru[NGC], drad[NGC], dradc[NGC];
    double usum = 0.0, usumc = 0.0, dusum = 0.0, dusumc = 0.0;
#pragma unroll
    for (int j = 0; j < NGC; j++) {
        const double rad0 = frac[j] * plankbnd;
        radlu[j] = rad0 + reflect * radld[j];
        radclru[j] = rad0 + reflect * radclrd[j];
        usum = usum + radlu[j];
        usumc = usumc + radclru[j];
        drad[j] = frac[j] * dplankbnd;
        dradc[j] = drad[j];
        dusum = dusum + drad[j];
    }
    out.uflx[gc] += (usum * 0.5) * wtdelw;
    if constexpr (MODE != 0) out.uflxc[gc] += (usumc * 0.5) * wtdelw;
    if (idrv) {
        out.duflx_dt[gc] += ((dusum * 0.5) * wtdelw) * T.fluxfac;
        if constexpr (MODE != 0) out.duflxc_dt[gc] += ((dusum * 0.5) * wtdelw) * T.fluxfac;
    }
#pragma unroll
    for (int j = 0; j < NGC; j++) { cldrad[j] = 0.0; clrrad[j] = 0.0; radmr[j] = 0.0; }

    // ------------------------------------------------------------------ upward sweep: rtrn :497-540
    for (int lev = 1; lev <= nlay; lev++) {
        int cloudy = 0, flags = 0;
        double cf = 0.0, efcl = 0.0;
        if constexpr (MODE != 0) {
            flags = W.cflag[(size_t)lev * ncb + col];
            cloudy = flags & 1;
            if (cloudy) {
                cf = a.cldfrac[gc + (size_t)nct * (lev - 1)];
                if constexpr (MODE == 1) efcl = W.efcl[((size_t)ibc * nlay + (lev - 1)) * ncb + col];
            }
        }
        usum = 0.0; usumc = 0.0; dusum = 0.0; dusumc = 0.0;
        if (!cloudy) {
#pragma unroll
            for (int j = 0; j < NGC; j++) {
                const size_t so = ((size_t)j * nlay + (lev - 1)) * ncb + col;
                const double atr = sA[so], bbu = sB[so];
                radlu[j] = radlu[j] + (bbu - radlu[j]) * atr;
                usum = usum + radlu[j];
                if (idrv) { drad[j] = drad[j] * (1.0 - atr); dusum = dusum + drad[j]; }
                if constexpr (MODE != 0) {
                    if (colcloud) {
                        radclru[j] = radclru[j] + (bbu - radclru[j]) * atr;
                        if (idrv) { dradc[j] = dradc[j] * (1.0 - atr); dusumc = dusumc + dradc[j]; }
                    } else {
                        radclru[j] = radlu[j];
                        if (idrv) { dradc[j] = drad[j]; dusumc = dusumc + dradc[j]; }
                    }
                    usumc = usumc + radclru[j];
                }
            }
        } else {
            if constexpr (MODE != 0) {
                double fclr1 = 0, fcld1 = 0, fcmb1 = 0, fcmb2 = 0, fclr2 = 0, fcld2 = 0;
                if constexpr (MODE == 2) {
                    const size_t mo = (size_t)(lev + 1) * ncb + col, ms = (size_t)(nlay + 2) * ncb;
                    fclr1 = W.mrfac[MR_FACCLR1 * ms + mo]; fcld1 = W.mrfac[MR_FACCLD1 * ms + mo];
                    fcmb1 = W.mrfac[MR_FACCMB1 * ms + mo]; fcmb2 = W.mrfac[MR_FACCMB2 * ms + mo];
                    fclr2 = W.mrfac[MR_FACCLR2 * ms + mo]; fcld2 = W.mrfac[MR_FACCLD2 * ms + mo];
                }
#pragma unroll
                for (int j = 0; j < NGC; j++) {
                    const size_t so = ((size_t)j * nlay + (lev - 1)) * ncb + col;
                    const double atr = sA[so], bbu = sB[so], atot = sC[so], bbutot = sD[so];
                    const double gassrc = bbu * atr;
                    if constexpr (MODE == 1) {
                        radlu[j] = radlu[j] - radlu[j] * (atr + efcl * (1. - atr)) + gassrc + cf * (bbutot * atot - gassrc);
                    } else {            // rtrnmr :680-703
                        if (flags & 2) {        // istcld(lev) == 1
                            cldrad[j] = cf * radlu[j];
                            clrrad[j] = radlu[j] - cldrad[j];
                            radmr[j] = 0.0;
                        }
                        const double ttot = 1. - atot;
                        const double cldsrc = bbutot * atot;
                        cldrad[j] = cldrad[j] * ttot + cf * cldsrc;
                        clrrad[j] = clrrad[j] * (1.0 - atr) + (1. - cf) * gassrc;
                        radlu[j] = cldrad[j] + clrrad[j];
                        const double radmod = radmr[j] * (fclr1 * (1.0 - atr) + fcld1 * ttot) - fcmb1 * gassrc + fcmb2 * cldsrc;
                        const double oldcld = cldrad[j] - radmod;
                        const double oldclr = clrrad[j] + radmod;
                        radmr[j] = -radmod + fclr2 * oldclr - fcld2 * oldcld;
                        cldrad[j] = cldrad[j] + radmr[j];
                        clrrad[j] = clrrad[j] - radmr[j];
                    }
                    usum = usum + radlu[j];
                    if (idrv) {
                        drad[j] = drad[j] * cf * (1.0 - atot) + drad[j] * (1.0 - cf) * (1.0 - atr);
                        dusum = dusum + drad[j];
                        dradc[j] = dradc[j] * (1.0 - atr);
                        dusumc = dusumc + dradc[j];
                    }
                    radclru[j] = radclru[j] + (bbu - radclru[j]) * atr;
                    usumc = usumc + radclru[j];
                }
            }
        }
        const size_t oo = gc + (size_t)nct * lev;
        out.uflx[oo] += (usum * 0.5) * wtdelw;
        if constexpr (MODE != 0) out.uflxc[oo] += (usumc * 0.5) * wtdelw;
        if (idrv) {
            out.duflx_dt[oo] += ((dusum * 0.5) * wtdelw) * T.fluxfac;
            if constexpr (MODE != 0) out.duflxc_dt[oo] += ((dusumc * 0.5) * wtdelw) * T.fluxfac;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_final : rtrn :580-604 (flux scaling, net flux, heating rate) and the output copies of
//           src/rrtmg_lw_rad.nomcica.f90:563-583.  clear_from_total: MODE 0 ran, so the clear-sky
//           stream equals the total-sky stream.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_final(DevTables T, FluxOut out, const double *pz, int ncol, int col0, int nct, int nlay,
                                               int idrv, int clear_from_total)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= ncol) return;
    const size_t gc = (size_t)col0 + col;
    double fnet_lo = 0.0, fnetc_lo = 0.0, pz_lo = 0.0;
    for (int lev = 0; lev <= nlay; lev++) {
        const size_t o = gc + (size_t)nct * lev;
        const double u = out.uflx[o] * T.fluxfac, d = out.dflx[o] * T.fluxfac;
        double uc, dc;
        if (clear_from_total) { uc = u; dc = d; }
        else { uc = out.uflxc[o] * T.fluxfac; dc = out.dflxc[o] * T.fluxfac; }
        out.uflx[o] = u; out.dflx[o] = d; out.uflxc[o] = uc; out.dflxc[o] = dc;
        if (idrv == 1 && clear_from_total) out.duflxc_dt[o] = out.duflx_dt[o];
        const double fnet = u - d, fnetc = uc - dc;
        if (out.fnet) { out.fnet[o] = fnet; out.fnetc[o] = fnetc; }
        const double pzl = pz[o];
        if (lev > 0) {
            const size_t ol = gc + (size_t)nct * (lev - 1);
            out.hr[ol] = T.heatfac * (fnet_lo - fnet) / (pz_lo - pzl);
            out.hrc[ol] = T.heatfac * (fnetc_lo - fnetc) / (pz_lo - pzl);
        }
        fnet_lo = fnet; fnetc_lo = fnetc; pz_lo = pzl;
    }
}

}  // namespace rrlw
