/*
 * TEST INFRASTRUCTURE - CPU oracle for the RRTMG_LW hot path.  NOT product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * A plain-C restatement of the reference algorithm (AER RRTMG_LW v5.0, /root/reference), double
 * precision throughout, same order of operations as the Fortran.  Every function cites the
 * reference file:line it follows.
 *
 * PARITY STATUS: "parity unpinned" against the reference's golden OUTPUT_RRTM files - those need the
 * absorption-coefficient data (src/rrtmg_lw_k_g.f90 | data/rrtmg_lw.nc) that the reference mount
 * lists in .MISSING_LARGE_BLOBS.  What IS pinned: (1) this file agrees with the reference's own
 * Fortran (built by oracle/Makefile into oracle/_ref) on identical inputs and identical stand-in
 * k-tables (tests/test_oracle_vs_ref.py, fixtures tests/golden/ref_*.npz); (2) the k-independent
 * part of the golden files - per-band surface emission, which fixes the Planck tables, setcoef's
 * interpolation, delwave and fluxfac - matches (tests/test_golden_planck.py).  When real k-data is
 * placed under data/, tests/test_golden_examples.py compares with all 13 golden files.
 *
 * Conventions: arrays are 1-based like the Fortran (element 0 unused) unless noted; levels are
 * 0..nlayers.  Reduced tables keep the reference's layout (g-point slowest).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rrlw_blob.h"

#define NBND 16
/* -DRRLW_G256: the 256-g-point model the reference keeps as a commented-out alternative (modules/parrrtm.f90:40-41,77-110;
 * src/rrtmg_lw_init.f90:313-314 "the full 256 g-point set can be restored with ngptlw=256, ngc=16*16, ngn=256*1., etc."):
 * every band keeps its 16 original g-points, the combination step copies. */
#ifdef RRLW_G256
#define NGPT 256
#else
#define NGPT 140
#endif
#define MG 16
#define MXLAY 603
#define NTBL 10000

/* ---------------------------------------------------------------------------------------------
 * Static data (modules rrlw_ref, rrlw_wvn, rrlw_cld, rrlw_tbl, rrlw_con)
 * ------------------------------------------------------------------------------------------- */
static double pref_[59], preflog_[59], tref_[59], chi_mls_[7 * 59];
static double totplnk_[181 * 16], totplk16_[181], totplnkderiv_[181 * 16], totplk16deriv_[181];
static double absice0_[2], absice1_[2 * 5], absice2_[43 * 16], absice3_[46 * 16], absliq1_[58 * 16];
static double abscld1, absliq0;
static int ngc_[16], ngs_[16], ngm_[256], ngn_[256], ngb_[256], nspa_[16], nspb_[16];
static double wt_[16], delwave_[16], rwgt_[256];
static double tau_tbl[NTBL + 1], exp_tbl[NTBL + 1], tfn_tbl[NTBL + 1];
static double bpade, heatfac, oneminus, pi_, fluxfac;
static const double tblint = 10000.0;
static int initialised = 0;
static char errmsg[256];

#define PREF(j) pref_[(j)-1]
#define PREFLOG(j) preflog_[(j)-1]
#define TREF(j) tref_[(j)-1]
#define CHI(i, j) chi_mls_[((i)-1) + 7 * ((j)-1)]
#define TOTPLNK(i, b) totplnk_[((i)-1) + 181 * ((b)-1)]
#define TOTPLNKD(i, b) totplnkderiv_[((i)-1) + 181 * ((b)-1)]
#define NGS(b) ((b) == 0 ? 0 : ngs_[(b)-1])

/* Reduced (140-g) tables, one list per band, Fortran element order (g-point axis where the reference has it) */
typedef struct {
    char name[24];
    int ndim, dims[4];
    double *d;
} tab_t;
static tab_t tabs[17][16];
static int ntabs[17];

static const double *T(int band, const char *name)
{
    for (int i = 0; i < ntabs[band]; i++)
        if (strcmp(tabs[band][i].name, name) == 0) return tabs[band][i].d;
    fprintf(stderr, "oracle: no table %s in band %d\n", name, band);
    abort();
}

#define A2(t, n1, i, ig) (t)[((i)-1) + (n1) * ((ig)-1)]
#define A3(t, n1, n2, i, j, ig) (t)[((i)-1) + (n1) * (((j)-1) + (n2) * ((ig)-1))]
#define FR2(t, ng, ig, j) (t)[((ig)-1) + (ng) * ((j)-1)]

/* ---------------------------------------------------------------------------------------------
 * rrtmg_lw_ini : src/rrtmg_lw_init.f90:47-194  (lwdatinit :197-300, LUTs :125-142, rwgt :149-173,
 * g-point reduction cmbgb1..16 :385-2034 - every cmbgbN is the same weighted group sum; Planck
 * fractions are summed without weights, e.g. :693-712)
 * ------------------------------------------------------------------------------------------- */
static int load_static(const char *path)
{
    rrlw_blob b;
    if (rrlw_blob_open(&b, path) != 0) { snprintf(errmsg, sizeof errmsg, "cannot read %s", path); return -1; }
#define GETD(nm, dst, cnt)                                                                      \
    do {                                                                                         \
        const rrlw_blob_entry *e = rrlw_blob_find(&b, nm);                                       \
        if (!e || e->dtype != 0 || e->nbytes != 8u * (cnt)) { snprintf(errmsg, sizeof errmsg, "static blob: bad %s", nm); return -2; } \
        memcpy(dst, rrlw_blob_data(&b, e), e->nbytes);                                           \
    } while (0)
#define GETI(nm, dst, cnt)                                                                      \
    do {                                                                                         \
        const rrlw_blob_entry *e = rrlw_blob_find(&b, nm);                                       \
        if (!e || e->dtype != 1 || e->nbytes != 4u * (cnt)) { snprintf(errmsg, sizeof errmsg, "static blob: bad %s", nm); return -2; } \
        memcpy(dst, rrlw_blob_data(&b, e), e->nbytes);                                           \
    } while (0)
    GETD("pref", pref_, 59); GETD("preflog", preflog_, 59); GETD("tref", tref_, 59);
    GETD("chi_mls", chi_mls_, 7 * 59);
    GETD("totplnk", totplnk_, 181 * 16); GETD("totplk16", totplk16_, 181);
    GETD("totplnkderiv", totplnkderiv_, 181 * 16); GETD("totplk16deriv", totplk16deriv_, 181);
    GETD("absice0", absice0_, 2); GETD("absice1", absice1_, 10); GETD("absice2", absice2_, 43 * 16);
    GETD("absice3", absice3_, 46 * 16); GETD("absliq1", absliq1_, 58 * 16);
    GETD("abscld1", &abscld1, 1); GETD("absliq0", &absliq0, 1);
    GETD("wt", wt_, 16); GETD("delwave", delwave_, 16);
    GETI("ngc", ngc_, 16); GETI("ngs", ngs_, 16); GETI("ngm", ngm_, 256); GETI("ngn", ngn_, 140);
    GETI("ngb", ngb_, 140); GETI("nspa", nspa_, 16); GETI("nspb", nspb_, 16);
    rrlw_blob_close(&b);
#ifdef RRLW_G256
    for (int ib = 0; ib < 16; ib++) { ngc_[ib] = 16; ngs_[ib] = 16 * (ib + 1); }
    for (int i = 0; i < 256; i++) { ngm_[i] = i % 16 + 1; ngn_[i] = 1; ngb_[i] = i / 16 + 1; }
#endif
    return 0;
}

static void reduced_name(const char *orig, char *out)
{
    /* kao -> ka, kbo_mn2 -> kb_mn2, selfrefo -> selfref, fracrefao -> fracrefa, ccl4o -> ccl4 */
    if ((strncmp(orig, "kao", 3) == 0 || strncmp(orig, "kbo", 3) == 0)) {
        out[0] = 'k'; out[1] = orig[1];
        strcpy(out + 2, orig + 3);
    } else {
        size_t n = strlen(orig);
        memcpy(out, orig, n - 1);
        out[n - 1] = 0;
    }
}

static int load_and_reduce_kdata(const char *path)
{
    rrlw_blob b;
    if (rrlw_blob_open(&b, path) != 0) { snprintf(errmsg, sizeof errmsg, "cannot read k-data blob %s", path); return -1; }
    for (int band = 1; band <= 16; band++) {
        for (int i = 0; i < ntabs[band]; i++) free(tabs[band][i].d);
        ntabs[band] = 0;
    }
    for (uint32_t i = 0; i < b.n; i++) {
        const rrlw_blob_entry *e = &b.ent[i];
        if (e->name[0] != 'b' || e->name[3] != '.') continue;
        int band = (e->name[1] - '0') * 10 + (e->name[2] - '0');
        if (band < 1 || band > 16 || e->dtype != 0) { snprintf(errmsg, sizeof errmsg, "k-data blob: bad entry %s", e->name); return -2; }
        const char *oname = e->name + 4;
        int is_frac = strncmp(oname, "fracref", 7) == 0;
        int gax = is_frac ? 0 : (int)e->ndim - 1;
        if (e->dims[gax] != MG) { snprintf(errmsg, sizeof errmsg, "k-data blob: %s has no 16-g axis", e->name); return -2; }
        size_t inner = 1, outer = 1;
        for (int d = 0; d < gax; d++) inner *= e->dims[d];
        for (int d = gax + 1; d < (int)e->ndim; d++) outer *= e->dims[d];
        int ng = ngc_[band - 1];
        tab_t *t = &tabs[band][ntabs[band]++];
        reduced_name(oname, t->name);
        t->ndim = (int)e->ndim;
        for (int d = 0; d < (int)e->ndim; d++) t->dims[d] = (int)e->dims[d];
        t->dims[gax] = ng;
        t->d = (double *)calloc(inner * outer * (size_t)ng, sizeof(double));
        const double *src = (const double *)rrlw_blob_data(&b, e);
        for (size_t o = 0; o < outer; o++)
            for (size_t in = 0; in < inner; in++) {
                int iprsm = 0;
                for (int igc = 1; igc <= ng; igc++) {
                    double sumk = 0.0;
                    int cnt = ngn_[NGS(band - 1) + igc - 1];
                    for (int ipr = 1; ipr <= cnt; ipr++) {
                        iprsm++;
                        double v = src[in + inner * ((size_t)(iprsm - 1) + MG * o)];
                        if (is_frac) sumk = sumk + v;
                        else sumk = sumk + v * rwgt_[iprsm - 1 + 16 * (band - 1)];
                    }
                    t->d[in + inner * ((size_t)(igc - 1) + (size_t)ng * o)] = sumk;
                }
            }
    }
    rrlw_blob_close(&b);
    return 0;
}

int orc_ngpt(void) { return NGPT; }

int orc_init(const char *static_path, const char *kdata_path, double cpdair)
{
    if (load_static(static_path) != 0) return -1;
    /* lwdatinit: src/rrtmg_lw_init.f90:243,265,298 */
    const double grav = 9.8066, secdy = 8.6400e4;
    heatfac = grav * secdy / (cpdair * 1.e2);
    /* rad driver constants: src/rrtmg_lw_rad.f90:451-453 */
    oneminus = 1.0 - 1.e-6;
    pi_ = 2.0 * asin(1.0);
    fluxfac = pi_ * 2.e4;
    /* LUTs: src/rrtmg_lw_init.f90:125-142 */
    const double pade = 0.278, expeps = 1.e-20;
    tau_tbl[0] = 0.0; tau_tbl[NTBL] = 1.e10;
    exp_tbl[0] = 1.0; exp_tbl[NTBL] = expeps;
    tfn_tbl[0] = 0.0; tfn_tbl[NTBL] = 1.0;
    bpade = 1.0 / pade;
    for (int itr = 1; itr <= NTBL - 1; itr++) {
        double tfn = (double)itr / (double)NTBL;
        tau_tbl[itr] = bpade * tfn / (1.0 - tfn);
        exp_tbl[itr] = exp(-tau_tbl[itr]);
        if (exp_tbl[itr] <= expeps) exp_tbl[itr] = expeps;
        if (tau_tbl[itr] < 0.06) tfn_tbl[itr] = tau_tbl[itr] / 6.0;
        else tfn_tbl[itr] = 1.0 - 2.0 * ((1.0 / tau_tbl[itr]) - (exp_tbl[itr] / (1. - exp_tbl[itr])));
    }
    /* rwgt: src/rrtmg_lw_init.f90:149-173 */
    int igcsm = 0;
    for (int ibnd = 1; ibnd <= NBND; ibnd++) {
        int iprsm = 0;
        double wtsm[MG + 1];
        if (ngc_[ibnd - 1] < MG) {
            for (int igc = 1; igc <= ngc_[ibnd - 1]; igc++) {
                igcsm++;
                double wtsum = 0.0;
                for (int ipr = 1; ipr <= ngn_[igcsm - 1]; ipr++) {
                    iprsm++;
                    wtsum = wtsum + wt_[iprsm - 1];
                }
                wtsm[igc] = wtsum;
            }
            for (int ig = 1; ig <= MG; ig++) {
                int ind = (ibnd - 1) * MG + ig;
                rwgt_[ind - 1] = wt_[ig - 1] / wtsm[ngm_[ind - 1]];
            }
        } else {
            for (int ig = 1; ig <= MG; ig++) {
                igcsm++;
                int ind = (ibnd - 1) * MG + ig;
                rwgt_[ind - 1] = 1.0;
            }
        }
    }
    if (load_and_reduce_kdata(kdata_path) != 0) return -2;
    initialised = 1;
    return 0;
}

const char *orc_errmsg(void) { return errmsg; }

/* copy a reduced table out (for table-layout tests); returns element count or -1 */
long orc_get_table(int band, const char *name, double *out, long cap)
{
    for (int i = 0; i < ntabs[band]; i++)
        if (strcmp(tabs[band][i].name, name) == 0) {
            long n = 1;
            for (int d = 0; d < tabs[band][i].ndim; d++) n *= tabs[band][i].dims[d];
            if (out && cap >= n) memcpy(out, tabs[band][i].d, (size_t)n * 8);
            return n;
        }
    return -1;
}

void orc_get_luts(double *tau, double *ex, double *tfn)
{
    memcpy(tau, tau_tbl, sizeof tau_tbl);
    memcpy(ex, exp_tbl, sizeof exp_tbl);
    memcpy(tfn, tfn_tbl, sizeof tfn_tbl);
}

/* ---------------------------------------------------------------------------------------------
 * Per-column working state (what rrtmg_lw keeps in locals, src/rrtmg_lw_rad.f90:336-440)
 * ------------------------------------------------------------------------------------------- */
#define ML (MXLAY + 2)
typedef struct {
    int nlayers, laytrop, ncbands, inflag, iceflag, liqflag;
    double pavel[ML], tavel[ML], pz[ML], tz[ML], tbound, semiss[NBND + 1], coldry[ML], wbrodl[ML];
    double wkl[ML][8], wx[ML][5], pwvcm;
    double cldfrac[ML], tauc[ML][NBND + 1], ciwp[ML], clwp[ML], rei[ML], rel[ML], taua[ML][NBND + 1];
    double taucloud[ML][NBND + 1];
    int jp[ML], jt[ML], jt1[ML], indself[ML], indfor[ML], indminor[ML];
    double planklay[ML][NBND + 1], planklev[ML][NBND + 1], plankbnd[NBND + 1], dplankbnd_dt[NBND + 1];
    double colh2o[ML], colco2[ML], colo3[ML], coln2o[ML], colco[ML], colch4[ML], colo2[ML], colbrd[ML];
    double fac00[ML], fac01[ML], fac10[ML], fac11[ML];
    double rat_h2oco2[ML], rat_h2oco2_1[ML], rat_h2oo3[ML], rat_h2oo3_1[ML], rat_h2on2o[ML], rat_h2on2o_1[ML];
    double rat_h2och4[ML], rat_h2och4_1[ML], rat_n2oco2[ML], rat_n2oco2_1[ML], rat_o3co2[ML], rat_o3co2_1[ML];
    double selffac[ML], selffrac[ML], forfac[ML], forfrac[ML], minorfrac[ML], scaleminor[ML], scaleminorn2[ML];
    double taug[ML][NGPT + 1], fracs[ML][NGPT + 1], taut[ML][NGPT + 1];
    /* McICA sub-column inputs (g-point resolved) */
    double cldfmc[ML][NGPT + 1], taucmc[ML][NGPT + 1], ciwpmc[ML][NGPT + 1], clwpmc[ML][NGPT + 1];
    /* outputs */
    double totuflux[ML], totdflux[ML], fnet[ML], htr[ML], totuclfl[ML], totdclfl[ML], fnetc[ML], htrc[ML];
    double dtotuflux_dt[ML], dtotuclfl_dt[ML];
} col_t;

static col_t *C = NULL;

/* ---------------------------------------------------------------------------------------------
 * cldprop : src/rrtmg_lw_cldprop.f90:50-295
 * ------------------------------------------------------------------------------------------- */
static int cldprop(col_t *c)
{
    static const int icb[3][16] = {  /* :167-169 */
        {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
        {1, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5},
        {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}};
    const double cldmin = 1.e-20;            /* :110 */
    double abscoice[NBND + 1], abscoliq[NBND + 1], tauctot[ML];
    int iceind = 0, liqind = 0;
    memset(abscoice, 0, sizeof abscoice);
    memset(abscoliq, 0, sizeof abscoliq);
    int nl = c->nlayers;
    c->ncbands = 1;
    for (int lay = 1; lay <= nl; lay++) {
        tauctot[lay] = 0.0;
        for (int ib = 1; ib <= NBND; ib++) {
            c->taucloud[lay][ib] = 0.0;
            tauctot[lay] = tauctot[lay] + c->tauc[lay][ib];
        }
    }
    for (int lay = 1; lay <= nl; lay++) {
        double cwp = c->ciwp[lay] + c->clwp[lay];
        if (c->cldfrac[lay] >= cldmin && (cwp >= cldmin || tauctot[lay] >= cldmin)) {
            if (c->inflag == 0) {
                c->ncbands = 16;
                for (int ib = 1; ib <= c->ncbands; ib++) c->taucloud[lay][ib] = c->tauc[lay][ib];
            } else if (c->inflag == 1) {
                c->ncbands = 16;
                for (int ib = 1; ib <= c->ncbands; ib++) c->taucloud[lay][ib] = abscld1 * cwp;
            } else if (c->inflag == 2) {
                double radice = c->rei[lay];
                if (c->ciwp[lay] == 0.0) {
                    abscoice[1] = 0.0;
                    iceind = 0;
                } else if (c->iceflag == 0) {
                    if (radice < 10.0) { strcpy(errmsg, "ICE RADIUS TOO SMALL"); return 1; }
                    abscoice[1] = absice0_[0] + absice0_[1] / radice;
                    iceind = 0;
                } else if (c->iceflag == 1) {
                    if (radice < 13.0 || radice > 130.) { strcpy(errmsg, "ICE RADIUS OUT OF BOUNDS"); return 1; }
                    c->ncbands = 5;
                    for (int ib = 1; ib <= c->ncbands; ib++)
                        abscoice[ib] = absice1_[0 + 2 * (ib - 1)] + absice1_[1 + 2 * (ib - 1)] / radice;
                    iceind = 1;
                } else if (c->iceflag == 2) {
                    if (radice < 5.0 || radice > 131.0) { strcpy(errmsg, "ICE RADIUS OUT OF BOUNDS"); return 1; }
                    c->ncbands = 16;
                    double factor = (radice - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 43) index = 42;
                    double fint = factor - (double)index;
                    for (int ib = 1; ib <= c->ncbands; ib++)
                        abscoice[ib] = A2(absice2_, 43, index, ib) +
                                       fint * (A2(absice2_, 43, index + 1, ib) - (A2(absice2_, 43, index, ib)));
                    iceind = 2;
                } else if (c->iceflag == 3) {
                    if (radice < 5.0 || radice > 140.0) { strcpy(errmsg, "ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS"); return 1; }
                    c->ncbands = 16;
                    double factor = (radice - 2.) / 3.;
                    int index = (int)factor;
                    if (index == 46) index = 45;
                    double fint = factor - (double)index;
                    for (int ib = 1; ib <= c->ncbands; ib++)
                        abscoice[ib] = A2(absice3_, 46, index, ib) +
                                       fint * (A2(absice3_, 46, index + 1, ib) - (A2(absice3_, 46, index, ib)));
                    iceind = 2;
                }
                if (c->clwp[lay] == 0.0) {
                    abscoliq[1] = 0.0;
                    liqind = 0;
                    if (iceind == 1) iceind = 2;
                } else if (c->liqflag == 0) {
                    abscoliq[1] = absliq0;
                    liqind = 0;
                    if (iceind == 1) iceind = 2;
                } else if (c->liqflag == 1) {
                    double radliq = c->rel[lay];
                    if (radliq < 2.5 || radliq > 60.) { strcpy(errmsg, "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS"); return 1; }
                    int index = (int)(radliq - 1.5);
                    if (index == 0) index = 1;
                    if (index == 58) index = 57;
                    double fint = radliq - 1.5 - (double)index;
                    c->ncbands = 16;
                    for (int ib = 1; ib <= c->ncbands; ib++)
                        abscoliq[ib] = A2(absliq1_, 58, index, ib) +
                                       fint * (A2(absliq1_, 58, index + 1, ib) - (A2(absliq1_, 58, index, ib)));
                    liqind = 2;
                }
                for (int ib = 1; ib <= c->ncbands; ib++)
                    c->taucloud[lay][ib] = c->ciwp[lay] * abscoice[icb[iceind][ib - 1]] +
                                           c->clwp[lay] * abscoliq[icb[liqind][ib - 1]];
            }
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * cldprmc : src/rrtmg_lw_cldprmc.f90:51-273
 * ------------------------------------------------------------------------------------------- */
static int cldprmc(col_t *c)
{
    const double cldmin = 1.e-20;            /* :112 */
    int nl = c->nlayers;
    c->ncbands = 1;                           /* :173 */
    for (int lay = 1; lay <= nl; lay++) {
        for (int ig = 1; ig <= NGPT; ig++) {
            double cwp = c->ciwpmc[lay][ig] + c->clwpmc[lay][ig];
            if (c->cldfmc[lay][ig] >= cldmin && (cwp >= cldmin || c->taucmc[lay][ig] >= cldmin)) {
                if (c->inflag == 0) {
                    return 0;                 /* :186-188 - returns at the first cloudy cell */
                } else if (c->inflag == 1) {
                    strcpy(errmsg, "INFLAG = 1 OPTION NOT AVAILABLE WITH MCICA");
                    return 1;                 /* :190-191 */
                } else if (c->inflag == 2) {
                    double radice = c->rei[lay];
                    double abscoice, abscoliq;
                    int ib = ngb_[ig - 1];
                    if (c->ciwpmc[lay][ig] == 0.0) {
                        abscoice = 0.0;
                    } else if (c->iceflag == 0) {
                        if (radice < 10.0) { strcpy(errmsg, "ICE RADIUS TOO SMALL"); return 1; }
                        abscoice = absice0_[0] + absice0_[1] / radice;
                    } else if (c->iceflag == 1) {
                        if (radice < 13.0 || radice > 130.) { strcpy(errmsg, "ICE RADIUS OUT OF BOUNDS"); return 1; }
                        c->ncbands = 5;
                        static const int ipat1[16] = {1, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5};
                        int icx = ipat1[ib - 1];
                        abscoice = absice1_[0 + 2 * (icx - 1)] + absice1_[1 + 2 * (icx - 1)] / radice;
                    } else if (c->iceflag == 2) {
                        if (radice < 5.0 || radice > 131.0) { strcpy(errmsg, "ICE RADIUS OUT OF BOUNDS"); return 1; }
                        c->ncbands = 16;
                        double factor = (radice - 2.) / 3.;
                        int index = (int)factor;
                        if (index == 43) index = 42;
                        double fint = factor - (double)index;
                        abscoice = A2(absice2_, 43, index, ib) +
                                   fint * (A2(absice2_, 43, index + 1, ib) - (A2(absice2_, 43, index, ib)));
                    } else if (c->iceflag == 3) {
                        if (radice < 5.0 || radice > 140.0) { strcpy(errmsg, "ICE GENERALIZED EFFECTIVE SIZE OUT OF BOUNDS"); return 1; }
                        c->ncbands = 16;
                        double factor = (radice - 2.) / 3.;
                        int index = (int)factor;
                        if (index == 46) index = 45;
                        double fint = factor - (double)index;
                        abscoice = A2(absice3_, 46, index, ib) +
                                   fint * (A2(absice3_, 46, index + 1, ib) - (A2(absice3_, 46, index, ib)));
                    } else {
                        abscoice = 0.0;
                    }
                    if (c->clwpmc[lay][ig] == 0.0) {
                        abscoliq = 0.0;
                    } else if (c->liqflag == 0) {
                        abscoliq = absliq0;
                    } else if (c->liqflag == 1) {
                        double radliq = c->rel[lay];
                        if (radliq < 2.5 || radliq > 60.) { strcpy(errmsg, "LIQUID EFFECTIVE RADIUS OUT OF BOUNDS"); return 1; }
                        int index = (int)(radliq - 1.5);
                        if (index == 0) index = 1;
                        if (index == 58) index = 57;
                        double fint = radliq - 1.5 - (double)index;
                        abscoliq = A2(absliq1_, 58, index, ib) +
                                   fint * (A2(absliq1_, 58, index + 1, ib) - (A2(absliq1_, 58, index, ib)));
                    } else {
                        abscoliq = 0.0;
                    }
                    c->taucmc[lay][ig] = c->ciwpmc[lay][ig] * abscoice + c->clwpmc[lay][ig] * abscoliq;
                }
            }
        }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * setcoef : src/rrtmg_lw_setcoef.f90:50-434
 * ------------------------------------------------------------------------------------------- */
static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static void setcoef(col_t *c, int istart, int idrv)
{
    const double stpfac = 296. / 1013.;
    int nl = c->nlayers;
    int indbound = clampi((int)(c->tbound - 159.), 1, 180);
    double tbndfrac = c->tbound - 159. - (double)indbound;
    int indlev0 = clampi((int)(c->tz[0] - 159.), 1, 180);
    double t0frac = c->tz[0] - 159. - (double)indlev0;
    c->laytrop = 0;
    for (int lay = 1; lay <= nl; lay++) {
        int indlay = clampi((int)(c->tavel[lay] - 159.), 1, 180);
        double tlayfrac = c->tavel[lay] - 159. - (double)indlay;
        int indlev = clampi((int)(c->tz[lay] - 159.), 1, 180);
        double tlevfrac = c->tz[lay] - 159. - (double)indlev;
        double dbdtlev, dbdtlay;
        for (int iband = 1; iband <= 15; iband++) {   /* :203-226 */
            if (lay == 1) {
                dbdtlev = TOTPLNK(indbound + 1, iband) - TOTPLNK(indbound, iband);
                c->plankbnd[iband] = c->semiss[iband] * (TOTPLNK(indbound, iband) + tbndfrac * dbdtlev);
                dbdtlev = TOTPLNK(indlev0 + 1, iband) - TOTPLNK(indlev0, iband);
                c->planklev[0][iband] = TOTPLNK(indlev0, iband) + t0frac * dbdtlev;
                if (idrv == 1) {
                    dbdtlev = TOTPLNKD(indbound + 1, iband) - TOTPLNKD(indbound, iband);
                    c->dplankbnd_dt[iband] = c->semiss[iband] * (TOTPLNKD(indbound, iband) + tbndfrac * dbdtlev);
                }
            }
            dbdtlev = TOTPLNK(indlev + 1, iband) - TOTPLNK(indlev, iband);
            dbdtlay = TOTPLNK(indlay + 1, iband) - TOTPLNK(indlay, iband);
            c->planklay[lay][iband] = TOTPLNK(indlay, iband) + tlayfrac * dbdtlay;
            c->planklev[lay][iband] = TOTPLNK(indlev, iband) + tlevfrac * dbdtlev;
        }
        int iband = 16;                               /* :233-269 */
        if (istart == 16) {
            if (lay == 1) {
                dbdtlev = totplk16_[indbound] - totplk16_[indbound - 1];
                c->plankbnd[iband] = c->semiss[iband] * (totplk16_[indbound - 1] + tbndfrac * dbdtlev);
                if (idrv == 1) {
                    dbdtlev = totplk16deriv_[indbound] - totplk16deriv_[indbound - 1];
                    c->dplankbnd_dt[iband] = c->semiss[iband] * (totplk16deriv_[indbound - 1] + tbndfrac * dbdtlev);
                }
                dbdtlev = TOTPLNK(indlev0 + 1, iband) - TOTPLNK(indlev0, iband);
                c->planklev[0][iband] = totplk16_[indlev0 - 1] + t0frac * dbdtlev;
            }
            dbdtlev = totplk16_[indlev] - totplk16_[indlev - 1];
            dbdtlay = totplk16_[indlay] - totplk16_[indlay - 1];
            c->planklay[lay][iband] = totplk16_[indlay - 1] + tlayfrac * dbdtlay;
            c->planklev[lay][iband] = totplk16_[indlev - 1] + tlevfrac * dbdtlev;
        } else {
            if (lay == 1) {
                dbdtlev = TOTPLNK(indbound + 1, iband) - TOTPLNK(indbound, iband);
                c->plankbnd[iband] = c->semiss[iband] * (TOTPLNK(indbound, iband) + tbndfrac * dbdtlev);
                if (idrv == 1) {
                    dbdtlev = TOTPLNKD(indbound + 1, iband) - TOTPLNKD(indbound, iband);
                    c->dplankbnd_dt[iband] = c->semiss[iband] * (TOTPLNKD(indbound, iband) + tbndfrac * dbdtlev);
                }
                dbdtlev = TOTPLNK(indlev0 + 1, iband) - TOTPLNK(indlev0, iband);
                c->planklev[0][iband] = TOTPLNK(indlev0, iband) + t0frac * dbdtlev;
            }
            dbdtlev = TOTPLNK(indlev + 1, iband) - TOTPLNK(indlev, iband);
            dbdtlay = TOTPLNK(indlay + 1, iband) - TOTPLNK(indlay, iband);
            c->planklay[lay][iband] = TOTPLNK(indlay, iband) + tlayfrac * dbdtlay;
            c->planklev[lay][iband] = TOTPLNK(indlev, iband) + tlevfrac * dbdtlev;
        }

        double plog = log(c->pavel[lay]);            /* :276-284 */
        c->jp[lay] = clampi((int)(36. - 5 * (plog + 0.04)), 1, 58);
        int jp1 = c->jp[lay] + 1;
        double fp = 5. * (PREFLOG(c->jp[lay]) - plog);
        c->jt[lay] = clampi((int)(3. + (c->tavel[lay] - TREF(c->jp[lay])) / 15.), 1, 4);   /* :293-306 */
        double ft = ((c->tavel[lay] - TREF(c->jp[lay])) / 15.) - (double)(c->jt[lay] - 3);
        c->jt1[lay] = clampi((int)(3. + (c->tavel[lay] - TREF(jp1)) / 15.), 1, 4);
        double ft1 = ((c->tavel[lay] - TREF(jp1)) / 15.) - (double)(c->jt1[lay] - 3);
        double water = c->wkl[lay][1] / c->coldry[lay];
        double scalefac = c->pavel[lay] * stpfac / c->tavel[lay];
        double factor;
        int jpl = c->jp[lay];
        if (!(plog <= 4.56)) {                       /* :312-366 lower atmosphere */
            c->laytrop = c->laytrop + 1;
            c->forfac[lay] = scalefac / (1. + water);
            factor = (332.0 - c->tavel[lay]) / 36.0;
            c->indfor[lay] = (int)fmin(2, fmax(1, (int)factor));
            c->forfrac[lay] = factor - (double)c->indfor[lay];
            c->selffac[lay] = water * c->forfac[lay];
            factor = (c->tavel[lay] - 188.0) / 7.2;
            c->indself[lay] = (int)fmin(9, fmax(1, (int)factor - 7));
            c->selffrac[lay] = factor - (double)(c->indself[lay] + 7);
            c->scaleminor[lay] = c->pavel[lay] / c->tavel[lay];
            c->scaleminorn2[lay] = (c->pavel[lay] / c->tavel[lay]) * (c->wbrodl[lay] / (c->coldry[lay] + c->wkl[lay][1]));
            factor = (c->tavel[lay] - 180.8) / 7.2;
            c->indminor[lay] = (int)fmin(18, fmax(1, (int)factor));
            c->minorfrac[lay] = factor - (double)c->indminor[lay];
            c->rat_h2oco2[lay] = CHI(1, jpl) / CHI(2, jpl);
            c->rat_h2oco2_1[lay] = CHI(1, jpl + 1) / CHI(2, jpl + 1);
            c->rat_h2oo3[lay] = CHI(1, jpl) / CHI(3, jpl);
            c->rat_h2oo3_1[lay] = CHI(1, jpl + 1) / CHI(3, jpl + 1);
            c->rat_h2on2o[lay] = CHI(1, jpl) / CHI(4, jpl);
            c->rat_h2on2o_1[lay] = CHI(1, jpl + 1) / CHI(4, jpl + 1);
            c->rat_h2och4[lay] = CHI(1, jpl) / CHI(6, jpl);
            c->rat_h2och4_1[lay] = CHI(1, jpl + 1) / CHI(6, jpl + 1);
            c->rat_n2oco2[lay] = CHI(4, jpl) / CHI(2, jpl);
            c->rat_n2oco2_1[lay] = CHI(4, jpl + 1) / CHI(2, jpl + 1);
        } else {                                     /* :369-412 upper atmosphere */
            c->forfac[lay] = scalefac / (1. + water);
            factor = (c->tavel[lay] - 188.0) / 36.0;
            c->indfor[lay] = 3;
            c->forfrac[lay] = factor - 1.0;
            c->selffac[lay] = water * c->forfac[lay];
            c->scaleminor[lay] = c->pavel[lay] / c->tavel[lay];
            c->scaleminorn2[lay] = (c->pavel[lay] / c->tavel[lay]) * (c->wbrodl[lay] / (c->coldry[lay] + c->wkl[lay][1]));
            factor = (c->tavel[lay] - 180.8) / 7.2;
            c->indminor[lay] = (int)fmin(18, fmax(1, (int)factor));
            c->minorfrac[lay] = factor - (double)c->indminor[lay];
            c->rat_h2oco2[lay] = CHI(1, jpl) / CHI(2, jpl);
            c->rat_h2oco2_1[lay] = CHI(1, jpl + 1) / CHI(2, jpl + 1);
            c->rat_o3co2[lay] = CHI(3, jpl) / CHI(2, jpl);
            c->rat_o3co2_1[lay] = CHI(3, jpl + 1) / CHI(2, jpl + 1);
        }
        c->colh2o[lay] = 1.e-20 * c->wkl[lay][1];    /* :354-366 / :399-411 */
        c->colco2[lay] = 1.e-20 * c->wkl[lay][2];
        c->colo3[lay] = 1.e-20 * c->wkl[lay][3];
        c->coln2o[lay] = 1.e-20 * c->wkl[lay][4];
        c->colco[lay] = 1.e-20 * c->wkl[lay][5];
        c->colch4[lay] = 1.e-20 * c->wkl[lay][6];
        c->colo2[lay] = 1.e-20 * c->wkl[lay][7];
        if (c->colco2[lay] == 0.) c->colco2[lay] = 1.e-32 * c->coldry[lay];
        if (c->colo3[lay] == 0.) c->colo3[lay] = 1.e-32 * c->coldry[lay];
        if (c->coln2o[lay] == 0.) c->coln2o[lay] = 1.e-32 * c->coldry[lay];
        if (c->colco[lay] == 0.) c->colco[lay] = 1.e-32 * c->coldry[lay];
        if (c->colch4[lay] == 0.) c->colch4[lay] = 1.e-32 * c->coldry[lay];
        c->colbrd[lay] = 1.e-20 * c->wbrodl[lay];

        double compfp = 1. - fp;                     /* :421-429 */
        c->fac10[lay] = compfp * ft;
        c->fac00[lay] = compfp * (1. - ft);
        c->fac11[lay] = fp * ft1;
        c->fac01[lay] = fp * (1. - ft1);
        c->selffac[lay] = c->colh2o[lay] * c->selffac[lay];
        c->forfac[lay] = c->colh2o[lay] * c->forfac[lay];
    }
}

/* ---------------------------------------------------------------------------------------------
 * taumol : src/rrtmg_lw_taumol.f90:50-3166
 * helpers for the pieces every taugbN repeats
 * ------------------------------------------------------------------------------------------- */
typedef struct { double speccomb, specparm, fs; int js; } spec_t;

/* e.g. :523-528 (mult = 8 lower / 4 upper) */
static spec_t spec_calc(double cola, double rat, double colb, double mult)
{
    spec_t s;
    s.speccomb = cola + rat * colb;
    s.specparm = cola / s.speccomb;
    if (s.specparm >= oneminus) s.specparm = oneminus;
    double specmult = mult * (s.specparm);
    s.js = 1 + (int)specmult;
    s.fs = fmod(specmult, 1.0);
    return s;
}

typedef struct { int mode; double f000, f100, f200, f010, f110, f210; } fac6_t;

/* lower-atmosphere binary-species stencil weights, :569-598 (first plane; :600-629 second plane) */
static fac6_t fac6(double specparm, double fs, double fa, double fb)
{
    fac6_t r;
    r.f200 = r.f210 = 0.0;
    if (specparm < 0.125) {
        double p = fs - 1;
        double p2 = p * p, p4 = p2 * p2;
        double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
        r.mode = 0;
        r.f000 = fk0 * fa; r.f100 = fk1 * fa; r.f200 = fk2 * fa;
        r.f010 = fk0 * fb; r.f110 = fk1 * fb; r.f210 = fk2 * fb;
    } else if (specparm > 0.875) {
        double p = -fs;
        double p2 = p * p, p4 = p2 * p2;
        double fk0 = p4, fk1 = 1 - p - 2.0 * p4, fk2 = p + p4;
        r.mode = 1;
        r.f000 = fk0 * fa; r.f100 = fk1 * fa; r.f200 = fk2 * fa;
        r.f010 = fk0 * fb; r.f110 = fk1 * fb; r.f210 = fk2 * fb;
    } else {
        r.mode = 2;
        r.f000 = (1. - fs) * fa; r.f010 = (1. - fs) * fb;
        r.f100 = fs * fa; r.f110 = fs * fb;
    }
    return r;
}

/* :641-663 */
static double sum6(const double *absa, int n1, int ind, int ig, const fac6_t *f, double speccomb)
{
    if (f->mode == 0)
        return speccomb * (f->f000 * A2(absa, n1, ind, ig) + f->f100 * A2(absa, n1, ind + 1, ig) +
                           f->f200 * A2(absa, n1, ind + 2, ig) + f->f010 * A2(absa, n1, ind + 9, ig) +
                           f->f110 * A2(absa, n1, ind + 10, ig) + f->f210 * A2(absa, n1, ind + 11, ig));
    if (f->mode == 1)
        return speccomb * (f->f200 * A2(absa, n1, ind - 1, ig) + f->f100 * A2(absa, n1, ind, ig) +
                           f->f000 * A2(absa, n1, ind + 1, ig) + f->f210 * A2(absa, n1, ind + 8, ig) +
                           f->f110 * A2(absa, n1, ind + 9, ig) + f->f010 * A2(absa, n1, ind + 10, ig));
    return speccomb * (f->f000 * A2(absa, n1, ind, ig) + f->f100 * A2(absa, n1, ind + 1, ig) +
                       f->f010 * A2(absa, n1, ind + 9, ig) + f->f110 * A2(absa, n1, ind + 10, ig));
}

/* :350-353 */
static double cont(double fac, double frac, const double *ref, int n1, int ind, int ig)
{
    return fac * (A2(ref, n1, ind, ig) + frac * (A2(ref, n1, ind + 1, ig) - A2(ref, n1, ind, ig)));
}

/* 1-D minor-gas interpolation, e.g. :354-355 */
static double minor1(const double *k, int indm, double minorfrac, int ig)
{
    return A2(k, 19, indm, ig) + minorfrac * (A2(k, 19, indm + 1, ig) - A2(k, 19, indm, ig));
}

/* 2-D (mixture x temperature) minor-gas interpolation, :635-639 */
static double minor2(const double *k, int nj, int jm, double fm, int indm, double minorfrac, int ig)
{
    double m1 = A3(k, nj, 19, jm, indm, ig) + fm * (A3(k, nj, 19, jm + 1, indm, ig) - A3(k, nj, 19, jm, indm, ig));
    double m2 = A3(k, nj, 19, jm, indm + 1, ig) + fm * (A3(k, nj, 19, jm + 1, indm + 1, ig) - A3(k, nj, 19, jm, indm + 1, ig));
    return m1 + minorfrac * (m2 - m1);
}

/* single-key major term, :356-360 */
static double major1(const double *ab, int n1, int ind0, int ind1, int ig, const col_t *c, int lay)
{
    return c->fac00[lay] * A2(ab, n1, ind0, ig) + c->fac10[lay] * A2(ab, n1, ind0 + 1, ig) +
           c->fac01[lay] * A2(ab, n1, ind1, ig) + c->fac11[lay] * A2(ab, n1, ind1 + 1, ig);
}

/* upper-atmosphere binary major term (always linear, stride nspb=5), :762-771 */
static double major_upper5(const double *absb, int n1, int ind0, int ind1, int ig, const spec_t *s, const spec_t *s1,
                           const col_t *c, int lay)
{
    double fac000 = (1. - s->fs) * c->fac00[lay], fac010 = (1. - s->fs) * c->fac10[lay];
    double fac100 = s->fs * c->fac00[lay], fac110 = s->fs * c->fac10[lay];
    double fac001 = (1. - s1->fs) * c->fac01[lay], fac011 = (1. - s1->fs) * c->fac11[lay];
    double fac101 = s1->fs * c->fac01[lay], fac111 = s1->fs * c->fac11[lay];
    return s->speccomb * (fac000 * A2(absb, n1, ind0, ig) + fac100 * A2(absb, n1, ind0 + 1, ig) +
                          fac010 * A2(absb, n1, ind0 + 5, ig) + fac110 * A2(absb, n1, ind0 + 6, ig)) +
           s1->speccomb * (fac001 * A2(absb, n1, ind1, ig) + fac101 * A2(absb, n1, ind1 + 1, ig) +
                           fac011 * A2(absb, n1, ind1 + 5, ig) + fac111 * A2(absb, n1, ind1 + 6, ig));
}

/* "high-concentration" column rescale, e.g. :547-554 */
static double adjcol(double col, double coldry, double chiref, double thresh, double base, double expo)
{
    double chi = col / coldry;
    double rat = 1.e20 * chi / chiref;
    if (rat > thresh) {
        double adjfac = base + pow(rat - base, expo);
        return adjfac * chiref * coldry * 1.e-20;
    }
    return col;
}

#define IND0A(b) (((c->jp[lay] - 1) * 5 + (c->jt[lay] - 1)) * nspa_[(b)-1])
#define IND1A(b) ((c->jp[lay] * 5 + (c->jt1[lay] - 1)) * nspa_[(b)-1])
#define IND0B(b) (((c->jp[lay] - 13) * 5 + (c->jt[lay] - 1)) * nspb_[(b)-1])
#define IND1B(b) (((c->jp[lay] - 12) * 5 + (c->jt1[lay] - 1)) * nspb_[(b)-1])

/* band 1: :299-392 */
static void taugb1(col_t *c)
{
    const int ng = ngc_[0], gs = NGS(0);
    const double *absa = T(1, "ka"), *absb = T(1, "kb"), *ka_mn2 = T(1, "ka_mn2"), *kb_mn2 = T(1, "kb_mn2");
    const double *selfref = T(1, "selfref"), *forref = T(1, "forref"), *fracrefa = T(1, "fracrefa"), *fracrefb = T(1, "fracrefb");
    for (int lay = 1; lay <= c->laytrop; lay++) {
        int ind0 = IND0A(1) + 1, ind1 = IND1A(1) + 1;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        double pp = c->pavel[lay];
        double corradj = 1.;
        if (pp < 250.) corradj = 1. - 0.15 * (250. - pp) / 154.4;
        double scalen2 = c->colbrd[lay] * c->scaleminorn2[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double taun2 = scalen2 * minor1(ka_mn2, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = corradj * (c->colh2o[lay] * major1(absa, 65, ind0, ind1, ig, c, lay) + tauself + taufor + taun2);
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        int ind0 = IND0B(1) + 1, ind1 = IND1B(1) + 1;
        int indf = c->indfor[lay], indm = c->indminor[lay];
        double pp = c->pavel[lay];
        double corradj = 1. - 0.15 * (pp / 95.6);
        double scalen2 = c->colbrd[lay] * c->scaleminorn2[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double taun2 = scalen2 * minor1(kb_mn2, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = corradj * (c->colh2o[lay] * major1(absb, 235, ind0, ind1, ig, c, lay) + taufor + taun2);
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 2: :395-464 */
static void taugb2(col_t *c)
{
    const int ng = ngc_[1], gs = NGS(1);
    const double *absa = T(2, "ka"), *absb = T(2, "kb");
    const double *selfref = T(2, "selfref"), *forref = T(2, "forref"), *fracrefa = T(2, "fracrefa"), *fracrefb = T(2, "fracrefb");
    for (int lay = 1; lay <= c->laytrop; lay++) {
        int ind0 = IND0A(2) + 1, ind1 = IND1A(2) + 1;
        int inds = c->indself[lay], indf = c->indfor[lay];
        double pp = c->pavel[lay];
        double corradj = 1. - .05 * (pp - 100.) / 900.;
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            c->taug[lay][gs + ig] = corradj * (c->colh2o[lay] * major1(absa, 65, ind0, ind1, ig, c, lay) + tauself + taufor);
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        int ind0 = IND0B(2) + 1, ind1 = IND1B(2) + 1;
        int indf = c->indfor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            c->taug[lay][gs + ig] = c->colh2o[lay] * major1(absb, 235, ind0, ind1, ig, c, lay) + taufor;
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 3: :467-779 */
static void taugb3(col_t *c)
{
    const int ng = ngc_[2], gs = NGS(2);
    const double *absa = T(3, "ka"), *absb = T(3, "kb"), *ka_mn2o = T(3, "ka_mn2o"), *kb_mn2o = T(3, "kb_mn2o");
    const double *selfref = T(3, "selfref"), *forref = T(3, "forref"), *fracrefa = T(3, "fracrefa"), *fracrefb = T(3, "fracrefb");
    double refrat_planck_a = CHI(1, 9) / CHI(2, 9), refrat_planck_b = CHI(1, 13) / CHI(2, 13);
    double refrat_m_a = CHI(1, 3) / CHI(2, 3), refrat_m_b = CHI(1, 13) / CHI(2, 13);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2oco2[lay], c->colco2[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2oco2_1[lay], c->colco2[lay], 8.);
        spec_t sm = spec_calc(c->colh2o[lay], refrat_m_a, c->colco2[lay], 8.);
        double adjcoln2o = adjcol(c->coln2o[lay], c->coldry[lay], CHI(4, c->jp[lay] + 1), 1.5, 0.5, 0.65);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->colco2[lay], 8.);
        int ind0 = IND0A(3) + s.js, ind1 = IND1A(3) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double absn2o = minor2(ka_mn2o, 9, sm.js, sm.fs, indm, c->minorfrac[lay], ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor + adjcoln2o * absn2o;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2oco2[lay], c->colco2[lay], 4.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2oco2_1[lay], c->colco2[lay], 4.);
        spec_t sm = spec_calc(c->colh2o[lay], refrat_m_b, c->colco2[lay], 4.);
        double adjcoln2o = adjcol(c->coln2o[lay], c->coldry[lay], CHI(4, c->jp[lay] + 1), 1.5, 0.5, 0.65);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_b, c->colco2[lay], 4.);
        int ind0 = IND0B(3) + s.js, ind1 = IND1B(3) + s1.js;
        int indf = c->indfor[lay], indm = c->indminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double absn2o = minor2(kb_mn2o, 5, sm.js, sm.fs, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = major_upper5(absb, 1175, ind0, ind1, ig, &s, &s1, c, lay) + taufor + adjcoln2o * absn2o;
            c->fracs[lay][gs + ig] = FR2(fracrefb, ng, ig, sp.js) + sp.fs * (FR2(fracrefb, ng, ig, sp.js + 1) - FR2(fracrefb, ng, ig, sp.js));
        }
    }
}

/* band 4: :782-1038 */
static void taugb4(col_t *c)
{
    const int ng = ngc_[3], gs = NGS(3);
    const double *absa = T(4, "ka"), *absb = T(4, "kb");
    const double *selfref = T(4, "selfref"), *forref = T(4, "forref"), *fracrefa = T(4, "fracrefa"), *fracrefb = T(4, "fracrefb");
    double refrat_planck_a = CHI(1, 11) / CHI(2, 11), refrat_planck_b = CHI(3, 13) / CHI(2, 13);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2oco2[lay], c->colco2[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2oco2_1[lay], c->colco2[lay], 8.);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->colco2[lay], 8.);
        int ind0 = IND0A(4) + s.js, ind1 = IND1A(4) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        spec_t s = spec_calc(c->colo3[lay], c->rat_o3co2[lay], c->colco2[lay], 4.);
        spec_t s1 = spec_calc(c->colo3[lay], c->rat_o3co2_1[lay], c->colco2[lay], 4.);
        spec_t sp = spec_calc(c->colo3[lay], refrat_planck_b, c->colco2[lay], 4.);
        int ind0 = IND0B(4) + s.js, ind1 = IND1B(4) + s1.js;
        for (int ig = 1; ig <= ng; ig++) {
            c->taug[lay][gs + ig] = major_upper5(absb, 1175, ind0, ind1, ig, &s, &s1, c, lay);
            c->fracs[lay][gs + ig] = FR2(fracrefb, ng, ig, sp.js) + sp.fs * (FR2(fracrefb, ng, ig, sp.js + 1) - FR2(fracrefb, ng, ig, sp.js));
        }
        /* :1028-1034 empirical stratospheric multipliers */
        c->taug[lay][gs + 8] = c->taug[lay][gs + 8] * 0.92;
        c->taug[lay][gs + 9] = c->taug[lay][gs + 9] * 0.88;
        c->taug[lay][gs + 10] = c->taug[lay][gs + 10] * 1.07;
        c->taug[lay][gs + 11] = c->taug[lay][gs + 11] * 1.1;
        c->taug[lay][gs + 12] = c->taug[lay][gs + 12] * 0.99;
        c->taug[lay][gs + 13] = c->taug[lay][gs + 13] * 0.88;
        c->taug[lay][gs + 14] = c->taug[lay][gs + 14] * 0.943;
    }
}

/* band 5: :1041-1313 */
static void taugb5(col_t *c)
{
    const int ng = ngc_[4], gs = NGS(4);
    const double *absa = T(5, "ka"), *absb = T(5, "kb"), *ka_mo3 = T(5, "ka_mo3"), *ccl4 = T(5, "ccl4");
    const double *selfref = T(5, "selfref"), *forref = T(5, "forref"), *fracrefa = T(5, "fracrefa"), *fracrefb = T(5, "fracrefb");
    double refrat_planck_a = CHI(1, 5) / CHI(2, 5), refrat_planck_b = CHI(3, 43) / CHI(2, 43);
    double refrat_m_a = CHI(1, 7) / CHI(2, 7);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2oco2[lay], c->colco2[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2oco2_1[lay], c->colco2[lay], 8.);
        spec_t sm = spec_calc(c->colh2o[lay], refrat_m_a, c->colco2[lay], 8.);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->colco2[lay], 8.);
        int ind0 = IND0A(5) + s.js, ind1 = IND1A(5) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double abso3 = minor2(ka_mo3, 9, sm.js, sm.fs, indm, c->minorfrac[lay], ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor + abso3 * c->colo3[lay] + c->wx[lay][1] * ccl4[ig - 1];
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        spec_t s = spec_calc(c->colo3[lay], c->rat_o3co2[lay], c->colco2[lay], 4.);
        spec_t s1 = spec_calc(c->colo3[lay], c->rat_o3co2_1[lay], c->colco2[lay], 4.);
        spec_t sp = spec_calc(c->colo3[lay], refrat_planck_b, c->colco2[lay], 4.);
        int ind0 = IND0B(5) + s.js, ind1 = IND1B(5) + s1.js;
        for (int ig = 1; ig <= ng; ig++) {
            c->taug[lay][gs + ig] = major_upper5(absb, 1175, ind0, ind1, ig, &s, &s1, c, lay) + c->wx[lay][1] * ccl4[ig - 1];
            c->fracs[lay][gs + ig] = FR2(fracrefb, ng, ig, sp.js) + sp.fs * (FR2(fracrefb, ng, ig, sp.js + 1) - FR2(fracrefb, ng, ig, sp.js));
        }
    }
}

/* band 6: :1316-1399 */
static void taugb6(col_t *c)
{
    const int ng = ngc_[5], gs = NGS(5);
    const double *absa = T(6, "ka"), *ka_mco2 = T(6, "ka_mco2"), *cfc11adj = T(6, "cfc11adj"), *cfc12 = T(6, "cfc12");
    const double *selfref = T(6, "selfref"), *forref = T(6, "forref"), *fracrefa = T(6, "fracrefa");
    for (int lay = 1; lay <= c->laytrop; lay++) {
        double adjcolco2 = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, c->jp[lay] + 1), 3.0, 2.0, 0.77);
        int ind0 = IND0A(6) + 1, ind1 = IND1A(6) + 1;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double absco2 = minor1(ka_mco2, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colh2o[lay] * major1(absa, 65, ind0, ind1, ig, c, lay) + tauself + taufor +
                                    adjcolco2 * absco2 + c->wx[lay][2] * cfc11adj[ig - 1] + c->wx[lay][3] * cfc12[ig - 1];
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++)
        for (int ig = 1; ig <= ng; ig++) {
            c->taug[lay][gs + ig] = 0.0 + c->wx[lay][2] * cfc11adj[ig - 1] + c->wx[lay][3] * cfc12[ig - 1];
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
}

/* band 7: :1402-1673 */
static void taugb7(col_t *c)
{
    const int ng = ngc_[6], gs = NGS(6);
    const double *absa = T(7, "ka"), *absb = T(7, "kb"), *ka_mco2 = T(7, "ka_mco2"), *kb_mco2 = T(7, "kb_mco2");
    const double *selfref = T(7, "selfref"), *forref = T(7, "forref"), *fracrefa = T(7, "fracrefa"), *fracrefb = T(7, "fracrefb");
    double refrat_planck_a = CHI(1, 3) / CHI(3, 3), refrat_m_a = CHI(1, 3) / CHI(3, 3);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2oo3[lay], c->colo3[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2oo3_1[lay], c->colo3[lay], 8.);
        spec_t sm = spec_calc(c->colh2o[lay], refrat_m_a, c->colo3[lay], 8.);
        double adjcolco2 = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, c->jp[lay] + 1), 3.0, 3.0, 0.79);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->colo3[lay], 8.);
        int ind0 = IND0A(7) + s.js, ind1 = IND1A(7) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double absco2 = minor2(ka_mco2, 9, sm.js, sm.fs, indm, c->minorfrac[lay], ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor + adjcolco2 * absco2;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        double adjcolco2 = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, c->jp[lay] + 1), 3.0, 2.0, 0.79);
        int ind0 = IND0B(7) + 1, ind1 = IND1B(7) + 1;
        int indm = c->indminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double absco2 = minor1(kb_mco2, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colo3[lay] * major1(absb, 235, ind0, ind1, ig, c, lay) + adjcolco2 * absco2;
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
        /* :1664-1669 */
        c->taug[lay][gs + 6] = c->taug[lay][gs + 6] * 0.92;
        c->taug[lay][gs + 7] = c->taug[lay][gs + 7] * 0.88;
        c->taug[lay][gs + 8] = c->taug[lay][gs + 8] * 1.07;
        c->taug[lay][gs + 9] = c->taug[lay][gs + 9] * 1.1;
        c->taug[lay][gs + 10] = c->taug[lay][gs + 10] * 0.99;
        c->taug[lay][gs + 11] = c->taug[lay][gs + 11] * 0.855;
    }
}

/* band 8: :1676-1796 */
static void taugb8(col_t *c)
{
    const int ng = ngc_[7], gs = NGS(7);
    const double *absa = T(8, "ka"), *absb = T(8, "kb"), *ka_mco2 = T(8, "ka_mco2"), *ka_mn2o = T(8, "ka_mn2o");
    const double *ka_mo3 = T(8, "ka_mo3"), *kb_mco2 = T(8, "kb_mco2"), *kb_mn2o = T(8, "kb_mn2o");
    const double *cfc12 = T(8, "cfc12"), *cfc22adj = T(8, "cfc22adj");
    const double *selfref = T(8, "selfref"), *forref = T(8, "forref"), *fracrefa = T(8, "fracrefa"), *fracrefb = T(8, "fracrefb");
    for (int lay = 1; lay <= c->laytrop; lay++) {
        double adjcolco2 = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, c->jp[lay] + 1), 3.0, 2.0, 0.65);
        int ind0 = IND0A(8) + 1, ind1 = IND1A(8) + 1;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double absco2 = minor1(ka_mco2, indm, c->minorfrac[lay], ig);
            double abso3 = minor1(ka_mo3, indm, c->minorfrac[lay], ig);
            double absn2o = minor1(ka_mn2o, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colh2o[lay] * major1(absa, 65, ind0, ind1, ig, c, lay) + tauself + taufor +
                                    adjcolco2 * absco2 + c->colo3[lay] * abso3 + c->coln2o[lay] * absn2o +
                                    c->wx[lay][3] * cfc12[ig - 1] + c->wx[lay][4] * cfc22adj[ig - 1];
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        double adjcolco2 = adjcol(c->colco2[lay], c->coldry[lay], CHI(2, c->jp[lay] + 1), 3.0, 2.0, 0.65);
        int ind0 = IND0B(8) + 1, ind1 = IND1B(8) + 1;
        int indm = c->indminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double absco2 = minor1(kb_mco2, indm, c->minorfrac[lay], ig);
            double absn2o = minor1(kb_mn2o, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colo3[lay] * major1(absb, 235, ind0, ind1, ig, c, lay) + adjcolco2 * absco2 +
                                    c->coln2o[lay] * absn2o + c->wx[lay][3] * cfc12[ig - 1] + c->wx[lay][4] * cfc22adj[ig - 1];
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 9: :1799-2059 */
static void taugb9(col_t *c)
{
    const int ng = ngc_[8], gs = NGS(8);
    const double *absa = T(9, "ka"), *absb = T(9, "kb"), *ka_mn2o = T(9, "ka_mn2o"), *kb_mn2o = T(9, "kb_mn2o");
    const double *selfref = T(9, "selfref"), *forref = T(9, "forref"), *fracrefa = T(9, "fracrefa"), *fracrefb = T(9, "fracrefb");
    double refrat_planck_a = CHI(1, 9) / CHI(6, 9), refrat_m_a = CHI(1, 3) / CHI(6, 3);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2och4[lay], c->colch4[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2och4_1[lay], c->colch4[lay], 8.);
        spec_t sm = spec_calc(c->colh2o[lay], refrat_m_a, c->colch4[lay], 8.);
        double adjcoln2o = adjcol(c->coln2o[lay], c->coldry[lay], CHI(4, c->jp[lay] + 1), 1.5, 0.5, 0.65);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->colch4[lay], 8.);
        int ind0 = IND0A(9) + s.js, ind1 = IND1A(9) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double absn2o = minor2(ka_mn2o, 9, sm.js, sm.fs, indm, c->minorfrac[lay], ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor + adjcoln2o * absn2o;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        double adjcoln2o = adjcol(c->coln2o[lay], c->coldry[lay], CHI(4, c->jp[lay] + 1), 1.5, 0.5, 0.65);
        int ind0 = IND0B(9) + 1, ind1 = IND1B(9) + 1;
        int indm = c->indminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double absn2o = minor1(kb_mn2o, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colch4[lay] * major1(absb, 235, ind0, ind1, ig, c, lay) + adjcoln2o * absn2o;
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 10: :2062-2126 */
static void taugb10(col_t *c)
{
    const int ng = ngc_[9], gs = NGS(9);
    const double *absa = T(10, "ka"), *absb = T(10, "kb");
    const double *selfref = T(10, "selfref"), *forref = T(10, "forref"), *fracrefa = T(10, "fracrefa"), *fracrefb = T(10, "fracrefb");
    for (int lay = 1; lay <= c->laytrop; lay++) {
        int ind0 = IND0A(10) + 1, ind1 = IND1A(10) + 1;
        int inds = c->indself[lay], indf = c->indfor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            c->taug[lay][gs + ig] = c->colh2o[lay] * major1(absa, 65, ind0, ind1, ig, c, lay) + tauself + taufor;
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        int ind0 = IND0B(10) + 1, ind1 = IND1B(10) + 1;
        int indf = c->indfor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            c->taug[lay][gs + ig] = c->colh2o[lay] * major1(absb, 235, ind0, ind1, ig, c, lay) + taufor;
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 11: :2129-2206 */
static void taugb11(col_t *c)
{
    const int ng = ngc_[10], gs = NGS(10);
    const double *absa = T(11, "ka"), *absb = T(11, "kb"), *ka_mo2 = T(11, "ka_mo2"), *kb_mo2 = T(11, "kb_mo2");
    const double *selfref = T(11, "selfref"), *forref = T(11, "forref"), *fracrefa = T(11, "fracrefa"), *fracrefb = T(11, "fracrefb");
    for (int lay = 1; lay <= c->laytrop; lay++) {
        int ind0 = IND0A(11) + 1, ind1 = IND1A(11) + 1;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        double scaleo2 = c->colo2[lay] * c->scaleminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double tauo2 = scaleo2 * minor1(ka_mo2, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colh2o[lay] * major1(absa, 65, ind0, ind1, ig, c, lay) + tauself + taufor + tauo2;
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        int ind0 = IND0B(11) + 1, ind1 = IND1B(11) + 1;
        int indf = c->indfor[lay], indm = c->indminor[lay];
        double scaleo2 = c->colo2[lay] * c->scaleminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double tauo2 = scaleo2 * minor1(kb_mo2, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colh2o[lay] * major1(absb, 235, ind0, ind1, ig, c, lay) + taufor + tauo2;
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 12: :2209-2409 */
static void taugb12(col_t *c)
{
    const int ng = ngc_[11], gs = NGS(11);
    const double *absa = T(12, "ka");
    const double *selfref = T(12, "selfref"), *forref = T(12, "forref"), *fracrefa = T(12, "fracrefa");
    double refrat_planck_a = CHI(1, 10) / CHI(2, 10);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2oco2[lay], c->colco2[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2oco2_1[lay], c->colco2[lay], 8.);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->colco2[lay], 8.);
        int ind0 = IND0A(12) + s.js, ind1 = IND1A(12) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++)
        for (int ig = 1; ig <= ng; ig++) {
            c->taug[lay][gs + ig] = 0.0;
            c->fracs[lay][gs + ig] = 0.0;
        }
}

/* band 13: :2412-2669 */
static void taugb13(col_t *c)
{
    const int ng = ngc_[12], gs = NGS(12);
    const double *absa = T(13, "ka"), *ka_mco2 = T(13, "ka_mco2"), *ka_mco = T(13, "ka_mco"), *kb_mo3 = T(13, "kb_mo3");
    const double *selfref = T(13, "selfref"), *forref = T(13, "forref"), *fracrefa = T(13, "fracrefa"), *fracrefb = T(13, "fracrefb");
    double refrat_planck_a = CHI(1, 5) / CHI(4, 5), refrat_m_a = CHI(1, 1) / CHI(4, 1), refrat_m_a3 = CHI(1, 3) / CHI(4, 3);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2on2o[lay], c->coln2o[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2on2o_1[lay], c->coln2o[lay], 8.);
        spec_t smco2 = spec_calc(c->colh2o[lay], refrat_m_a, c->coln2o[lay], 8.);
        double adjcolco2 = adjcol(c->colco2[lay], c->coldry[lay], 3.55e-4, 3.0, 2.0, 0.68);   /* :2494-2501 */
        spec_t smco = spec_calc(c->colh2o[lay], refrat_m_a3, c->coln2o[lay], 8.);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->coln2o[lay], 8.);
        int ind0 = IND0A(13) + s.js, ind1 = IND1A(13) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double absco2 = minor2(ka_mco2, 9, smco2.js, smco2.fs, indm, c->minorfrac[lay], ig);
            double absco = minor2(ka_mco, 9, smco.js, smco.fs, indm, c->minorfrac[lay], ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor + adjcolco2 * absco2 + c->colco[lay] * absco;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        int indm = c->indminor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double abso3 = minor1(kb_mo3, indm, c->minorfrac[lay], ig);
            c->taug[lay][gs + ig] = c->colo3[lay] * abso3;
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 14: :2672-2730 */
static void taugb14(col_t *c)
{
    const int ng = ngc_[13], gs = NGS(13);
    const double *absa = T(14, "ka"), *absb = T(14, "kb");
    const double *selfref = T(14, "selfref"), *forref = T(14, "forref"), *fracrefa = T(14, "fracrefa"), *fracrefb = T(14, "fracrefb");
    for (int lay = 1; lay <= c->laytrop; lay++) {
        int ind0 = IND0A(14) + 1, ind1 = IND1A(14) + 1;
        int inds = c->indself[lay], indf = c->indfor[lay];
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            c->taug[lay][gs + ig] = c->colco2[lay] * major1(absa, 65, ind0, ind1, ig, c, lay) + tauself + taufor;
            c->fracs[lay][gs + ig] = fracrefa[ig - 1];
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        int ind0 = IND0B(14) + 1, ind1 = IND1B(14) + 1;
        for (int ig = 1; ig <= ng; ig++) {
            c->taug[lay][gs + ig] = c->colco2[lay] * major1(absb, 235, ind0, ind1, ig, c, lay);
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

/* band 15: :2733-2955 */
static void taugb15(col_t *c)
{
    const int ng = ngc_[14], gs = NGS(14);
    const double *absa = T(15, "ka"), *ka_mn2 = T(15, "ka_mn2");
    const double *selfref = T(15, "selfref"), *forref = T(15, "forref"), *fracrefa = T(15, "fracrefa");
    double refrat_planck_a = CHI(4, 1) / CHI(2, 1), refrat_m_a = CHI(4, 1) / CHI(2, 1);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->coln2o[lay], c->rat_n2oco2[lay], c->colco2[lay], 8.);
        spec_t s1 = spec_calc(c->coln2o[lay], c->rat_n2oco2_1[lay], c->colco2[lay], 8.);
        spec_t sm = spec_calc(c->coln2o[lay], refrat_m_a, c->colco2[lay], 8.);
        spec_t sp = spec_calc(c->coln2o[lay], refrat_planck_a, c->colco2[lay], 8.);
        int ind0 = IND0A(15) + s.js, ind1 = IND1A(15) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay], indm = c->indminor[lay];
        double scalen2 = c->colbrd[lay] * c->scaleminor[lay];      /* :2817 (scaleminor, not scaleminorn2) */
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double taun2 = scalen2 * minor2(ka_mn2, 9, sm.js, sm.fs, indm, c->minorfrac[lay], ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor + taun2;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++)
        for (int ig = 1; ig <= ng; ig++) {
            c->taug[lay][gs + ig] = 0.0;
            c->fracs[lay][gs + ig] = 0.0;
        }
}

/* band 16: :2958-3164 */
static void taugb16(col_t *c)
{
    const int ng = ngc_[15], gs = NGS(15);
    const double *absa = T(16, "ka"), *absb = T(16, "kb");
    const double *selfref = T(16, "selfref"), *forref = T(16, "forref"), *fracrefa = T(16, "fracrefa"), *fracrefb = T(16, "fracrefb");
    double refrat_planck_a = CHI(1, 6) / CHI(6, 6);
    for (int lay = 1; lay <= c->laytrop; lay++) {
        spec_t s = spec_calc(c->colh2o[lay], c->rat_h2och4[lay], c->colch4[lay], 8.);
        spec_t s1 = spec_calc(c->colh2o[lay], c->rat_h2och4_1[lay], c->colch4[lay], 8.);
        spec_t sp = spec_calc(c->colh2o[lay], refrat_planck_a, c->colch4[lay], 8.);
        int ind0 = IND0A(16) + s.js, ind1 = IND1A(16) + s1.js;
        int inds = c->indself[lay], indf = c->indfor[lay];
        fac6_t f0 = fac6(s.specparm, s.fs, c->fac00[lay], c->fac10[lay]);
        fac6_t f1 = fac6(s1.specparm, s1.fs, c->fac01[lay], c->fac11[lay]);
        for (int ig = 1; ig <= ng; ig++) {
            double tauself = cont(c->selffac[lay], c->selffrac[lay], selfref, 10, inds, ig);
            double taufor = cont(c->forfac[lay], c->forfrac[lay], forref, 4, indf, ig);
            double tau_major = sum6(absa, 585, ind0, ig, &f0, s.speccomb);
            double tau_major1 = sum6(absa, 585, ind1, ig, &f1, s1.speccomb);
            c->taug[lay][gs + ig] = tau_major + tau_major1 + tauself + taufor;
            c->fracs[lay][gs + ig] = FR2(fracrefa, ng, ig, sp.js) + sp.fs * (FR2(fracrefa, ng, ig, sp.js + 1) - FR2(fracrefa, ng, ig, sp.js));
        }
    }
    for (int lay = c->laytrop + 1; lay <= c->nlayers; lay++) {
        int ind0 = IND0B(16) + 1, ind1 = IND1B(16) + 1;
        for (int ig = 1; ig <= ng; ig++) {
            c->taug[lay][gs + ig] = c->colch4[lay] * major1(absb, 235, ind0, ind1, ig, c, lay);
            c->fracs[lay][gs + ig] = fracrefb[ig - 1];
        }
    }
}

static void taumol(col_t *c)   /* :280-296 */
{
    taugb1(c); taugb2(c); taugb3(c); taugb4(c); taugb5(c); taugb6(c); taugb7(c); taugb8(c);
    taugb9(c); taugb10(c); taugb11(c); taugb12(c); taugb13(c); taugb14(c); taugb15(c); taugb16(c);
}

/* ---------------------------------------------------------------------------------------------
 * rtrn / rtrnmc : src/rrtmg_lw_rtrn.f90:51-606, src/rrtmg_lw_rtrnmc.f90:51-595 (mc != 0)
 * rtrnmr        : src/rrtmg_lw_rtrnmr.f90:51-806 (separate function below)
 * ------------------------------------------------------------------------------------------- */
static const int ipat[3][16] = {   /* rtrn :252-254 */
    {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1},
    {1, 2, 3, 3, 3, 4, 4, 4, 5, 5, 5, 5, 5, 5, 5, 5},
    {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16}};
static const double a0[16] = {1.66, 1.55, 1.58, 1.66, 1.54, 1.454, 1.89, 1.33, 1.668, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66, 1.66};
static const double a1[16] = {0.00, 0.25, 0.22, 0.00, 0.13, 0.446, -0.10, 0.40, -0.006, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
static const double a2[16] = {0.00, -12.0, -11.7, 0.00, -0.72, -0.243, 0.19, -0.062, 0.414, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00, 0.00};
static const double wtdiff = 0.5, rec_6 = 0.166667;

static void calc_secdiff(double pwvcm, double *secdiff)   /* rtrn :280-288 */
{
    for (int ibnd = 1; ibnd <= NBND; ibnd++) {
        if (ibnd == 1 || ibnd == 4 || ibnd >= 10) {
            secdiff[ibnd] = 1.66;
        } else {
            secdiff[ibnd] = a0[ibnd - 1] + a1[ibnd - 1] * exp(a2[ibnd - 1] * pwvcm);
            if (secdiff[ibnd] > 1.80) secdiff[ibnd] = 1.80;
            if (secdiff[ibnd] < 1.50) secdiff[ibnd] = 1.50;
        }
    }
}

static void finish_fluxes(col_t *c)    /* rtrn :580-604 */
{
    int nl = c->nlayers;
    c->totuflux[0] = c->totuflux[0] * fluxfac;
    c->totdflux[0] = c->totdflux[0] * fluxfac;
    c->fnet[0] = c->totuflux[0] - c->totdflux[0];
    c->totuclfl[0] = c->totuclfl[0] * fluxfac;
    c->totdclfl[0] = c->totdclfl[0] * fluxfac;
    c->fnetc[0] = c->totuclfl[0] - c->totdclfl[0];
    for (int lev = 1; lev <= nl; lev++) {
        c->totuflux[lev] = c->totuflux[lev] * fluxfac;
        c->totdflux[lev] = c->totdflux[lev] * fluxfac;
        c->fnet[lev] = c->totuflux[lev] - c->totdflux[lev];
        c->totuclfl[lev] = c->totuclfl[lev] * fluxfac;
        c->totdclfl[lev] = c->totdclfl[lev] * fluxfac;
        c->fnetc[lev] = c->totuclfl[lev] - c->totdclfl[lev];
        int l = lev - 1;
        c->htr[l] = heatfac * (c->fnet[l] - c->fnet[lev]) / (c->pz[l] - c->pz[lev]);
        c->htrc[l] = heatfac * (c->fnetc[l] - c->fnetc[lev]) / (c->pz[l] - c->pz[lev]);
    }
    c->htr[nl] = 0.0;
    c->htrc[nl] = 0.0;
}

static double urad[ML], drad[ML], clrurad[ML], clrdrad[ML], d_urad_dt[ML], d_clrurad_dt[ML];
static double atrans[ML], atot[ML], bbugas[ML], bbutot[ML];

static void zero_accumulators(col_t *c, int idrv)
{
    for (int lev = 0; lev <= c->nlayers; lev++) {
        urad[lev] = drad[lev] = clrurad[lev] = clrdrad[lev] = 0.0;
        c->totuflux[lev] = c->totdflux[lev] = c->totuclfl[lev] = c->totdclfl[lev] = 0.0;
        if (idrv == 1) {
            d_urad_dt[lev] = d_clrurad_dt[lev] = 0.0;
            c->dtotuflux_dt[lev] = c->dtotuclfl_dt[lev] = 0.0;
        }
    }
}

static void band_accumulate(col_t *c, int iband, int idrv)   /* rtrn :549-574 */
{
    for (int lev = c->nlayers; lev >= 0; lev--) {
        double uflux = urad[lev] * wtdiff, dflux = drad[lev] * wtdiff;
        urad[lev] = 0.0; drad[lev] = 0.0;
        c->totuflux[lev] = c->totuflux[lev] + uflux * delwave_[iband - 1];
        c->totdflux[lev] = c->totdflux[lev] + dflux * delwave_[iband - 1];
        double uclfl = clrurad[lev] * wtdiff, dclfl = clrdrad[lev] * wtdiff;
        clrurad[lev] = 0.0; clrdrad[lev] = 0.0;
        c->totuclfl[lev] = c->totuclfl[lev] + uclfl * delwave_[iband - 1];
        c->totdclfl[lev] = c->totdclfl[lev] + dclfl * delwave_[iband - 1];
    }
    if (idrv == 1)
        for (int lev = c->nlayers; lev >= 0; lev--) {
            double duflux_dt = d_urad_dt[lev] * wtdiff;
            d_urad_dt[lev] = 0.0;
            c->dtotuflux_dt[lev] = c->dtotuflux_dt[lev] + duflux_dt * delwave_[iband - 1] * fluxfac;
            double duclfl_dt = d_clrurad_dt[lev] * wtdiff;
            d_clrurad_dt[lev] = 0.0;
            c->dtotuclfl_dt[lev] = c->dtotuclfl_dt[lev] + duclfl_dt * delwave_[iband - 1] * fluxfac;
        }
}

/* gas-only layer quantities shared by all three solvers (clear branch :437-452) */
static inline int lut_index(double od) { return (int)(tblint * (od / (bpade + od)) + 0.5); }

static void rtrn_generic(col_t *c, int istart, int iend, int iout, int idrv, int mc)
{
    int nl = c->nlayers;
    double secdiff[NBND + 1];
    static double odcld[ML][NGPT + 1], abscld[ML][NGPT + 1], efclfrac[ML][NGPT + 1];
    static int icldlyr[ML];
    calc_secdiff(c->pwvcm, secdiff);
    zero_accumulators(c, idrv);
    for (int lay = 1; lay <= nl; lay++) {
        if (mc) {                               /* rtrnmc :307-329 */
            icldlyr[lay] = 0;
            for (int ig = 1; ig <= NGPT; ig++) {
                if (c->cldfmc[lay][ig] == 1.) {
                    int ib = ngb_[ig - 1];
                    odcld[lay][ig] = secdiff[ib] * c->taucmc[lay][ig];
                    double transcld = exp(-odcld[lay][ig]);
                    abscld[lay][ig] = 1. - transcld;
                    efclfrac[lay][ig] = abscld[lay][ig] * c->cldfmc[lay][ig];
                    icldlyr[lay] = 1;
                } else {
                    odcld[lay][ig] = 0.0; abscld[lay][ig] = 0.0; efclfrac[lay][ig] = 0.0;
                }
            }
        } else {                                /* rtrn :321-334 */
            for (int ib = 1; ib <= c->ncbands; ib++) {
                if (c->cldfrac[lay] >= 1.e-6) {
                    odcld[lay][ib] = secdiff[ib] * c->taucloud[lay][ib];
                    double transcld = exp(-odcld[lay][ib]);
                    abscld[lay][ib] = 1. - transcld;
                    efclfrac[lay][ib] = abscld[lay][ib] * c->cldfrac[lay];
                    icldlyr[lay] = 1;
                } else {
                    odcld[lay][ib] = 0.0; abscld[lay][ib] = 0.0; efclfrac[lay][ib] = 0.0;
                    icldlyr[lay] = 0;
                }
            }
        }
    }
    int igc = 1;
    for (int iband = istart; iband <= iend; iband++) {
        if (iout > 0 && iband >= 2) igc = NGS(iband - 1) + 1;
        int ib = 0;
        if (!mc) {
            if (c->ncbands == 1) ib = ipat[0][iband - 1];
            else if (c->ncbands == 5) ib = ipat[1][iband - 1];
            else if (c->ncbands == 16) ib = ipat[2][iband - 1];
        }
        do {
            int ic = mc ? igc : ib;             /* index into odcld/efclfrac */
            double radld = 0., radclrd = 0.;
            int iclddn = 0;
            for (int lev = nl; lev >= 1; lev--) {       /* :361-466 */
                double plfrac = c->fracs[lev][igc];
                double blay = c->planklay[lev][iband];
                double dplankup = c->planklev[lev][iband] - blay;
                double dplankdn = c->planklev[lev - 1][iband] - blay;
                double odepth = secdiff[iband] * c->taut[lev][igc];
                double bbd;
                if (odepth < 0.0) odepth = 0.0;
                if (icldlyr[lev] == 1) {
                    double cf = mc ? c->cldfmc[lev][igc] : c->cldfrac[lev];
                    iclddn = 1;
                    double odtot = odepth + odcld[lev][ic];
                    double gassrc, bbdtot;
                    if (odtot < 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        double odepth_rec = rec_6 * odepth;
                        gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[lev];
                        atot[lev] = odtot - 0.5 * odtot * odtot;
                        double odtot_rec = rec_6 * odtot;
                        bbdtot = plfrac * (blay + dplankdn * odtot_rec);
                        bbd = plfrac * (blay + dplankdn * odepth_rec);
                        radld = radld - radld * (atrans[lev] + efclfrac[lev][ic] * (1. - atrans[lev])) + gassrc +
                                cf * (bbdtot * atot[lev] - gassrc);
                        drad[lev - 1] = drad[lev - 1] + radld;
                        bbugas[lev] = plfrac * (blay + dplankup * odepth_rec);
                        bbutot[lev] = plfrac * (blay + dplankup * odtot_rec);
                    } else if (odepth <= 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        double odepth_rec = rec_6 * odepth;
                        gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[lev];
                        odtot = odepth + odcld[lev][ic];
                        int ittot = lut_index(odtot);
                        double tfactot = tfn_tbl[ittot];
                        bbdtot = plfrac * (blay + tfactot * dplankdn);
                        bbd = plfrac * (blay + dplankdn * odepth_rec);
                        atot[lev] = 1. - exp_tbl[ittot];
                        radld = radld - radld * (atrans[lev] + efclfrac[lev][ic] * (1. - atrans[lev])) + gassrc +
                                cf * (bbdtot * atot[lev] - gassrc);
                        drad[lev - 1] = drad[lev - 1] + radld;
                        bbugas[lev] = plfrac * (blay + dplankup * odepth_rec);
                        bbutot[lev] = plfrac * (blay + tfactot * dplankup);
                    } else {
                        int itgas = lut_index(odepth);
                        odepth = tau_tbl[itgas];
                        atrans[lev] = 1. - exp_tbl[itgas];
                        double tfacgas = tfn_tbl[itgas];
                        gassrc = atrans[lev] * plfrac * (blay + tfacgas * dplankdn);
                        odtot = odepth + odcld[lev][ic];
                        int ittot = lut_index(odtot);
                        double tfactot = tfn_tbl[ittot];
                        bbdtot = plfrac * (blay + tfactot * dplankdn);
                        bbd = plfrac * (blay + tfacgas * dplankdn);
                        atot[lev] = 1. - exp_tbl[ittot];
                        radld = radld - radld * (atrans[lev] + efclfrac[lev][ic] * (1. - atrans[lev])) + gassrc +
                                cf * (bbdtot * atot[lev] - gassrc);
                        drad[lev - 1] = drad[lev - 1] + radld;
                        bbugas[lev] = plfrac * (blay + tfacgas * dplankup);
                        bbutot[lev] = plfrac * (blay + tfactot * dplankup);
                    }
                } else {
                    if (odepth <= 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        odepth = rec_6 * odepth;
                        bbd = plfrac * (blay + dplankdn * odepth);
                        bbugas[lev] = plfrac * (blay + dplankup * odepth);
                    } else {
                        int itr = lut_index(odepth);
                        double transc = exp_tbl[itr];
                        atrans[lev] = 1. - transc;
                        double tausfac = tfn_tbl[itr];
                        bbd = plfrac * (blay + tausfac * dplankdn);
                        bbugas[lev] = plfrac * (blay + tausfac * dplankup);
                    }
                    radld = radld + (bbd - radld) * atrans[lev];
                    drad[lev - 1] = drad[lev - 1] + radld;
                }
                if (iclddn == 1) {
                    radclrd = radclrd + (bbd - radclrd) * atrans[lev];
                    clrdrad[lev - 1] = clrdrad[lev - 1] + radclrd;
                } else {
                    radclrd = radld;
                    clrdrad[lev - 1] = drad[lev - 1];
                }
            }
            double rad0 = c->fracs[1][igc] * c->plankbnd[iband];        /* :476-495 */
            double d_rad0_dt = 0., d_radlu_dt = 0., d_radclru_dt = 0.;
            if (idrv == 1) d_rad0_dt = c->fracs[1][igc] * c->dplankbnd_dt[iband];
            double reflect = 1. - c->semiss[iband];
            double radlu = rad0 + reflect * radld;
            double radclru = rad0 + reflect * radclrd;
            urad[0] = urad[0] + radlu;
            clrurad[0] = clrurad[0] + radclru;
            if (idrv == 1) {
                d_radlu_dt = d_rad0_dt;
                d_urad_dt[0] = d_urad_dt[0] + d_radlu_dt;
                d_radclru_dt = d_rad0_dt;
                d_clrurad_dt[0] = d_clrurad_dt[0] + d_radclru_dt;
            }
            for (int lev = 1; lev <= nl; lev++) {                       /* :497-540 */
                if (icldlyr[lev] == 1) {
                    double cf = mc ? c->cldfmc[lev][igc] : c->cldfrac[lev];
                    double gassrc = bbugas[lev] * atrans[lev];
                    radlu = radlu - radlu * (atrans[lev] + efclfrac[lev][ic] * (1. - atrans[lev])) + gassrc +
                            cf * (bbutot[lev] * atot[lev] - gassrc);
                    urad[lev] = urad[lev] + radlu;
                    if (idrv == 1) {
                        d_radlu_dt = d_radlu_dt * cf * (1.0 - atot[lev]) + d_radlu_dt * (1.0 - cf) * (1.0 - atrans[lev]);
                        d_urad_dt[lev] = d_urad_dt[lev] + d_radlu_dt;
                    }
                } else {
                    radlu = radlu + (bbugas[lev] - radlu) * atrans[lev];
                    urad[lev] = urad[lev] + radlu;
                    if (idrv == 1) {
                        d_radlu_dt = d_radlu_dt * (1.0 - atrans[lev]);
                        d_urad_dt[lev] = d_urad_dt[lev] + d_radlu_dt;
                    }
                }
                if (iclddn == 1) {
                    radclru = radclru + (bbugas[lev] - radclru) * atrans[lev];
                    clrurad[lev] = clrurad[lev] + radclru;
                } else {
                    radclru = radlu;
                    clrurad[lev] = urad[lev];
                }
                if (idrv == 1) {
                    if (iclddn == 1) {
                        d_radclru_dt = d_radclru_dt * (1.0 - atrans[lev]);
                        d_clrurad_dt[lev] = d_clrurad_dt[lev] + d_radclru_dt;
                    } else {
                        d_radclru_dt = d_radlu_dt;
                        d_clrurad_dt[lev] = d_urad_dt[lev];
                    }
                }
            }
            igc = igc + 1;
        } while (igc <= NGS(iband));
        band_accumulate(c, iband, idrv);
    }
    finish_fluxes(c);
}

/* rtrnmr : src/rrtmg_lw_rtrnmr.f90:51-806.  Convention for the reference's uninitialised reads
 * (faccmb1/2, faccmb1d/2d: assigned only at :419-424,:497-502 but read at :605-609,:694-698): ZERO. */
static void rtrnmr(col_t *c, int istart, int iend, int iout, int idrv)
{
    int nl = c->nlayers;
    double secdiff[NBND + 1];
    static double odcld[ML][NBND + 1];
    static int icldlyr[ML], istcld[ML + 1], istcldd[ML + 1];
    static double faccld1[ML + 1], faccld2[ML + 1], facclr1[ML + 1], facclr2[ML + 1], faccmb1[ML + 1], faccmb2[ML + 1];
    static double faccld1d[ML + 1], faccld2d[ML + 1], facclr1d[ML + 1], facclr2d[ML + 1], faccmb1d[ML + 1], faccmb2d[ML + 1];
    const double *cldfrac = c->cldfrac;
    double rat1 = 0., rat2 = 0., fmx, fmn;
    calc_secdiff(c->pwvcm, secdiff);
    zero_accumulators(c, idrv);
    for (int i = 0; i <= nl + 1; i++) {
        faccld1[i] = faccld2[i] = facclr1[i] = facclr2[i] = faccmb1[i] = faccmb2[i] = 0.;
        faccld1d[i] = faccld2d[i] = facclr1d[i] = facclr2d[i] = faccmb1d[i] = faccmb2d[i] = 0.;
        istcld[i] = istcldd[i] = 0;
    }
    for (int lay = 1; lay <= nl; lay++)             /* :333-343 */
        for (int ib = 1; ib <= c->ncbands; ib++) {
            if (cldfrac[lay] >= 1.e-6) { odcld[lay][ib] = secdiff[ib] * c->taucloud[lay][ib]; icldlyr[lay] = 1; }
            else { odcld[lay][ib] = 0.0; icldlyr[lay] = 0; }
        }
    istcld[1] = 1;                                  /* :347-428 */
    istcldd[nl] = 1;
    for (int lev = 1; lev <= nl; lev++) {
        if (icldlyr[lev] == 1) {
            istcld[lev + 1] = 0;
            if (lev == nl) {
                faccld1[lev + 1] = 0.; faccld2[lev + 1] = 0.; facclr1[lev + 1] = 0.;
                facclr2[lev + 1] = 0.; faccmb1[lev + 1] = 0.; faccmb2[lev + 1] = 0.;
            } else if (cldfrac[lev + 1] >= cldfrac[lev]) {
                faccld1[lev + 1] = 0.; faccld2[lev + 1] = 0.;
                if (istcld[lev] == 1) {
                    facclr1[lev + 1] = 0.; facclr2[lev + 1] = 0.;
                    if (cldfrac[lev] < 1.) facclr2[lev + 1] = (cldfrac[lev + 1] - cldfrac[lev]) / (1. - cldfrac[lev]);
                    facclr2[lev] = 0.; faccld2[lev] = 0.;
                } else {
                    fmx = fmax(cldfrac[lev], cldfrac[lev - 1]);
                    if (cldfrac[lev + 1] > fmx) {
                        facclr1[lev + 1] = rat2;
                        facclr2[lev + 1] = (cldfrac[lev + 1] - fmx) / (1. - fmx);
                    } else if (cldfrac[lev + 1] < fmx) {
                        facclr1[lev + 1] = (cldfrac[lev + 1] - cldfrac[lev]) / (cldfrac[lev - 1] - cldfrac[lev]);
                        facclr2[lev + 1] = 0.;
                    } else {
                        facclr1[lev + 1] = rat2;
                        facclr2[lev + 1] = 0.;
                    }
                }
                if (facclr1[lev + 1] > 0. || facclr2[lev + 1] > 0.) { rat1 = 1.; rat2 = 0.; }
                else { rat1 = 0.; rat2 = 0.; }
            } else {
                facclr1[lev + 1] = 0.; facclr2[lev + 1] = 0.;
                if (istcld[lev] == 1) {
                    faccld1[lev + 1] = 0.;
                    faccld2[lev + 1] = (cldfrac[lev] - cldfrac[lev + 1]) / cldfrac[lev];
                    facclr2[lev] = 0.; faccld2[lev] = 0.;
                } else {
                    fmn = fmin(cldfrac[lev], cldfrac[lev - 1]);
                    if (cldfrac[lev + 1] <= fmn) {
                        faccld1[lev + 1] = rat1;
                        faccld2[lev + 1] = (fmn - cldfrac[lev + 1]) / fmn;
                    } else {
                        faccld1[lev + 1] = (cldfrac[lev] - cldfrac[lev + 1]) / (cldfrac[lev] - fmn);
                        faccld2[lev + 1] = 0.;
                    }
                }
                if (faccld1[lev + 1] > 0. || faccld2[lev + 1] > 0.) { rat1 = 0.; rat2 = 1.; }
                else { rat1 = 0.; rat2 = 0.; }
            }
            if (istcld[lev] != 1) {
                faccmb1[lev + 1] = fmax(0., fmin(cldfrac[lev + 1] - cldfrac[lev], cldfrac[lev - 1] - cldfrac[lev]));
                faccmb2[lev + 1] = fmax(0., fmin(cldfrac[lev] - cldfrac[lev + 1], cldfrac[lev] - cldfrac[lev - 1]));
            }
        } else {
            istcld[lev + 1] = 1;
        }
    }
    for (int lev = nl; lev >= 1; lev--) {           /* :430-506 */
        if (icldlyr[lev] == 1) {
            istcldd[lev - 1] = 0;
            if (lev == 1) {
                faccld1d[lev - 1] = 0.; faccld2d[lev - 1] = 0.; facclr1d[lev - 1] = 0.;
                facclr2d[lev - 1] = 0.; faccmb1d[lev - 1] = 0.; faccmb2d[lev - 1] = 0.;
            } else if (cldfrac[lev - 1] >= cldfrac[lev]) {
                faccld1d[lev - 1] = 0.; faccld2d[lev - 1] = 0.;
                if (istcldd[lev] == 1) {
                    facclr1d[lev - 1] = 0.; facclr2d[lev - 1] = 0.;
                    if (cldfrac[lev] < 1.) facclr2d[lev - 1] = (cldfrac[lev - 1] - cldfrac[lev]) / (1. - cldfrac[lev]);
                    facclr2d[lev] = 0.; faccld2d[lev] = 0.;
                } else {
                    fmx = fmax(cldfrac[lev], cldfrac[lev + 1]);
                    if (cldfrac[lev - 1] > fmx) {
                        facclr1d[lev - 1] = rat2;
                        facclr2d[lev - 1] = (cldfrac[lev - 1] - fmx) / (1. - fmx);
                    } else if (cldfrac[lev - 1] < fmx) {
                        facclr1d[lev - 1] = (cldfrac[lev - 1] - cldfrac[lev]) / (cldfrac[lev + 1] - cldfrac[lev]);
                        facclr2d[lev - 1] = 0.;
                    } else {
                        facclr1d[lev - 1] = rat2;
                        facclr2d[lev - 1] = 0.;
                    }
                }
                if (facclr1d[lev - 1] > 0. || facclr2d[lev - 1] > 0.) { rat1 = 1.; rat2 = 0.; }
                else { rat1 = 0.; rat2 = 0.; }
            } else {
                facclr1d[lev - 1] = 0.; facclr2d[lev - 1] = 0.;
                if (istcldd[lev] == 1) {
                    faccld1d[lev - 1] = 0.;
                    faccld2d[lev - 1] = (cldfrac[lev] - cldfrac[lev - 1]) / cldfrac[lev];
                    facclr2d[lev] = 0.; faccld2d[lev] = 0.;
                } else {
                    fmn = fmin(cldfrac[lev], cldfrac[lev + 1]);
                    if (cldfrac[lev - 1] <= fmn) {
                        faccld1d[lev - 1] = rat1;
                        faccld2d[lev - 1] = (fmn - cldfrac[lev - 1]) / fmn;
                    } else {
                        faccld1d[lev - 1] = (cldfrac[lev] - cldfrac[lev - 1]) / (cldfrac[lev] - fmn);
                        faccld2d[lev - 1] = 0.;
                    }
                }
                if (faccld1d[lev - 1] > 0. || faccld2d[lev - 1] > 0.) { rat1 = 0.; rat2 = 1.; }
                else { rat1 = 0.; rat2 = 0.; }
            }
            if (istcldd[lev] != 1) {
                faccmb1d[lev - 1] = fmax(0., fmin(cldfrac[lev + 1] - cldfrac[lev], cldfrac[lev - 1] - cldfrac[lev]));
                faccmb2d[lev - 1] = fmax(0., fmin(cldfrac[lev] - cldfrac[lev + 1], cldfrac[lev] - cldfrac[lev - 1]));
            }
        } else {
            istcldd[lev - 1] = 1;
        }
    }

    int igc = 1;
    for (int iband = istart; iband <= iend; iband++) {
        if (iout > 0 && iband >= 2) igc = NGS(iband - 1) + 1;
        int ib = 0;
        if (c->ncbands == 1) ib = ipat[0][iband - 1];
        else if (c->ncbands == 5) ib = ipat[1][iband - 1];
        else if (c->ncbands == 16) ib = ipat[2][iband - 1];
        do {
            double radld = 0., radclrd = 0.;
            double cldradd = 0., clrradd = 0., cldradu = 0., clrradu = 0., oldcld = 0., oldclr = 0., rad = 0., radmod;
            int iclddn = 0;
            for (int lev = nl; lev >= 1; lev--) {           /* :531-630 */
                double plfrac = c->fracs[lev][igc];
                double blay = c->planklay[lev][iband];
                double dplankup = c->planklev[lev][iband] - blay;
                double dplankdn = c->planklev[lev - 1][iband] - blay;
                double odepth = secdiff[iband] * c->taut[lev][igc];
                double bbd;
                if (odepth < 0.0) odepth = 0.0;
                if (icldlyr[lev] == 1) {
                    iclddn = 1;
                    double odtot = odepth + odcld[lev][ib];
                    double gassrc, bbdtot;
                    if (odtot < 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        double odepth_rec = rec_6 * odepth;
                        gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[lev];
                        atot[lev] = odtot - 0.5 * odtot * odtot;
                        double odtot_rec = rec_6 * odtot;
                        bbdtot = plfrac * (blay + dplankdn * odtot_rec);
                        bbd = plfrac * (blay + dplankdn * odepth_rec);
                        bbugas[lev] = plfrac * (blay + dplankup * odepth_rec);
                        bbutot[lev] = plfrac * (blay + dplankup * odtot_rec);
                    } else if (odepth <= 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        double odepth_rec = rec_6 * odepth;
                        gassrc = plfrac * (blay + dplankdn * odepth_rec) * atrans[lev];
                        odtot = odepth + odcld[lev][ib];
                        int ittot = lut_index(odtot);
                        double tfactot = tfn_tbl[ittot];
                        bbdtot = plfrac * (blay + tfactot * dplankdn);
                        bbd = plfrac * (blay + dplankdn * odepth_rec);
                        atot[lev] = 1. - exp_tbl[ittot];
                        bbugas[lev] = plfrac * (blay + dplankup * odepth_rec);
                        bbutot[lev] = plfrac * (blay + tfactot * dplankup);
                    } else {
                        int itgas = lut_index(odepth);
                        odepth = tau_tbl[itgas];
                        atrans[lev] = 1. - exp_tbl[itgas];
                        double tfacgas = tfn_tbl[itgas];
                        gassrc = atrans[lev] * plfrac * (blay + tfacgas * dplankdn);
                        odtot = odepth + odcld[lev][ib];
                        int ittot = lut_index(odtot);
                        double tfactot = tfn_tbl[ittot];
                        bbdtot = plfrac * (blay + tfactot * dplankdn);
                        bbd = plfrac * (blay + tfacgas * dplankdn);
                        atot[lev] = 1. - exp_tbl[ittot];
                        bbugas[lev] = plfrac * (blay + tfacgas * dplankup);
                        bbutot[lev] = plfrac * (blay + tfactot * dplankup);
                    }
                    if (istcldd[lev] == 1) {
                        cldradd = cldfrac[lev] * radld;
                        clrradd = radld - cldradd;
                        oldcld = cldradd;
                        oldclr = clrradd;
                        rad = 0.;
                    }
                    double ttot = 1. - atot[lev];
                    double cldsrc = bbdtot * atot[lev];
                    cldradd = cldradd * ttot + cldfrac[lev] * cldsrc;
                    clrradd = clrradd * (1. - atrans[lev]) + (1. - cldfrac[lev]) * gassrc;
                    radld = cldradd + clrradd;
                    drad[lev - 1] = drad[lev - 1] + radld;
                    radmod = rad * (facclr1d[lev - 1] * (1. - atrans[lev]) + faccld1d[lev - 1] * ttot) -
                             faccmb1d[lev - 1] * gassrc + faccmb2d[lev - 1] * cldsrc;
                    oldcld = cldradd - radmod;
                    oldclr = clrradd + radmod;
                    rad = -radmod + facclr2d[lev - 1] * oldclr - faccld2d[lev - 1] * oldcld;
                    cldradd = cldradd + rad;
                    clrradd = clrradd - rad;
                } else {
                    if (odepth <= 0.06) {
                        atrans[lev] = odepth - 0.5 * odepth * odepth;
                        odepth = rec_6 * odepth;
                        bbd = plfrac * (blay + dplankdn * odepth);
                        bbugas[lev] = plfrac * (blay + dplankup * odepth);
                    } else {
                        int itr = lut_index(odepth);
                        double transc = exp_tbl[itr];
                        atrans[lev] = 1. - transc;
                        double tausfac = tfn_tbl[itr];
                        bbd = plfrac * (blay + tausfac * dplankdn);
                        bbugas[lev] = plfrac * (blay + tausfac * dplankup);
                    }
                    radld = radld + (bbd - radld) * atrans[lev];
                    drad[lev - 1] = drad[lev - 1] + radld;
                }
                if (iclddn == 1) {
                    radclrd = radclrd + (bbd - radclrd) * atrans[lev];
                    clrdrad[lev - 1] = clrdrad[lev - 1] + radclrd;
                } else {
                    radclrd = radld;
                    clrdrad[lev - 1] = drad[lev - 1];
                }
            }
            double rad0 = c->fracs[1][igc] * c->plankbnd[iband];        /* :640-661 */
            double d_rad0_dt = 0., d_radlu_dt = 0., d_radclru_dt = 0.;
            if (idrv == 1) d_rad0_dt = c->fracs[1][igc] * c->dplankbnd_dt[iband];
            double reflect = 1. - c->semiss[iband];
            double radlu = rad0 + reflect * radld;
            double radclru = rad0 + reflect * radclrd;
            urad[0] = urad[0] + radlu;
            clrurad[0] = clrurad[0] + radclru;
            if (idrv == 1) {
                d_radlu_dt = d_rad0_dt;
                d_urad_dt[0] = d_urad_dt[0] + d_radlu_dt;
                d_radclru_dt = d_rad0_dt;
                d_clrurad_dt[0] = d_clrurad_dt[0] + d_radclru_dt;
            }
            for (int lev = 1; lev <= nl; lev++) {                       /* :663-738 */
                if (icldlyr[lev] == 1) {
                    double gassrc = bbugas[lev] * atrans[lev];
                    if (istcld[lev] == 1) {
                        cldradu = cldfrac[lev] * radlu;
                        clrradu = radlu - cldradu;
                        oldcld = cldradu;
                        oldclr = clrradu;
                        rad = 0.;
                    }
                    double ttot = 1. - atot[lev];
                    double cldsrc = bbutot[lev] * atot[lev];
                    cldradu = cldradu * ttot + cldfrac[lev] * cldsrc;
                    clrradu = clrradu * (1.0 - atrans[lev]) + (1. - cldfrac[lev]) * gassrc;
                    radlu = cldradu + clrradu;
                    urad[lev] = urad[lev] + radlu;
                    radmod = rad * (facclr1[lev + 1] * (1.0 - atrans[lev]) + faccld1[lev + 1] * ttot) -
                             faccmb1[lev + 1] * gassrc + faccmb2[lev + 1] * cldsrc;
                    oldcld = cldradu - radmod;
                    oldclr = clrradu + radmod;
                    rad = -radmod + facclr2[lev + 1] * oldclr - faccld2[lev + 1] * oldcld;
                    cldradu = cldradu + rad;
                    clrradu = clrradu - rad;
                    if (idrv == 1) {
                        d_radlu_dt = d_radlu_dt * cldfrac[lev] * (1.0 - atot[lev]) +
                                     d_radlu_dt * (1.0 - cldfrac[lev]) * (1.0 - atrans[lev]);
                        d_urad_dt[lev] = d_urad_dt[lev] + d_radlu_dt;
                    }
                } else {
                    radlu = radlu + (bbugas[lev] - radlu) * atrans[lev];
                    urad[lev] = urad[lev] + radlu;
                    if (idrv == 1) {
                        d_radlu_dt = d_radlu_dt * (1.0 - atrans[lev]);
                        d_urad_dt[lev] = d_urad_dt[lev] + d_radlu_dt;
                    }
                }
                if (iclddn == 1) {
                    radclru = radclru + (bbugas[lev] - radclru) * atrans[lev];
                    clrurad[lev] = clrurad[lev] + radclru;
                } else {
                    radclru = radlu;
                    clrurad[lev] = urad[lev];
                }
                if (idrv == 1) {
                    if (iclddn == 1) {
                        d_radclru_dt = d_radclru_dt * (1.0 - atrans[lev]);
                        d_clrurad_dt[lev] = d_clrurad_dt[lev] + d_radclru_dt;
                    } else {
                        d_radclru_dt = d_radlu_dt;
                        d_clrurad_dt[lev] = d_urad_dt[lev];
                    }
                }
            }
            igc = igc + 1;
        } while (igc <= NGS(iband));
        band_accumulate(c, iband, idrv);
    }
    finish_fluxes(c);
}

/* ---------------------------------------------------------------------------------------------
 * inatm : src/rrtmg_lw_rad.nomcica.f90:591-919 (McICA flavour src/rrtmg_lw_rad.f90:598-924)
 * all 2-D inputs are Fortran (ncol, nlay[+1]) arrays: element (iplon, l) at [iplon-1 + ncol*(l-1)]
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int ncol, nlay;
    const double *play, *plev, *tlay, *tlev, *tsfc, *h2ovmr, *o3vmr, *co2vmr, *ch4vmr, *n2ovmr, *o2vmr;
    const double *cfc11vmr, *cfc12vmr, *cfc22vmr, *ccl4vmr, *emis;
    int inflglw, iceflglw, liqflglw;
    const double *cldfr, *taucld, *cicewp, *cliqwp, *reice, *reliq, *tauaer;         /* non-McICA */
    const double *cldfmcl, *taucmcl, *ciwpmcl, *clwpmcl, *reicmcl, *relqmcl;         /* McICA */
} gcm_in_t;

#define G2(a, i, l) (a)[((i)-1) + (size_t)in->ncol * ((l)-1)]

static void inatm(col_t *c, const gcm_in_t *in, int iplon, int icld, int iaer, int mcica)
{
    const double amd = 28.9660, amw = 18.0160, avogad = 6.02214199e+23, grav = 9.8066;
    int nl = in->nlay;
    c->nlayers = nl;
    for (int l = 0; l <= nl + 1; l++) {
        for (int m = 0; m < 8; m++) c->wkl[l][m] = 0.0;
        for (int m = 0; m < 5; m++) c->wx[l][m] = 0.0;
        c->cldfrac[l] = c->ciwp[l] = c->clwp[l] = c->rei[l] = c->rel[l] = 0.0;
        for (int b = 0; b <= NBND; b++) { c->tauc[l][b] = 0.0; c->taua[l][b] = 0.0; }
    }
    if (mcica)
        for (int l = 0; l <= nl + 1; l++)
            for (int ig = 0; ig <= NGPT; ig++) c->cldfmc[l][ig] = c->taucmc[l][ig] = c->ciwpmc[l][ig] = c->clwpmc[l][ig] = 0.0;
    double amttl = 0.0, wvttl = 0.0;
    c->tbound = in->tsfc[iplon - 1];
    c->pz[0] = G2(in->plev, iplon, 1);
    c->tz[0] = G2(in->tlev, iplon, 1);
    for (int l = 1; l <= nl; l++) {
        c->pavel[l] = G2(in->play, iplon, l);
        c->tavel[l] = G2(in->tlay, iplon, l);
        c->pz[l] = G2(in->plev, iplon, l + 1);
        c->tz[l] = G2(in->tlev, iplon, l + 1);
        c->wkl[l][1] = G2(in->h2ovmr, iplon, l);
        c->wkl[l][2] = G2(in->co2vmr, iplon, l);
        c->wkl[l][3] = G2(in->o3vmr, iplon, l);
        c->wkl[l][4] = G2(in->n2ovmr, iplon, l);
        c->wkl[l][6] = G2(in->ch4vmr, iplon, l);
        c->wkl[l][7] = G2(in->o2vmr, iplon, l);
        double amm = (1. - c->wkl[l][1]) * amd + c->wkl[l][1] * amw;
        c->coldry[l] = (c->pz[l - 1] - c->pz[l]) * 1.e3 * avogad / (1.e2 * grav * amm * (1. + c->wkl[l][1]));
    }
    for (int l = 1; l <= nl; l++) {
        c->wx[l][1] = G2(in->ccl4vmr, iplon, l);
        c->wx[l][2] = G2(in->cfc11vmr, iplon, l);
        c->wx[l][3] = G2(in->cfc12vmr, iplon, l);
        c->wx[l][4] = G2(in->cfc22vmr, iplon, l);
    }
    for (int l = 1; l <= nl; l++) {
        double summol = 0.0;
        for (int imol = 2; imol <= 7; imol++) summol = summol + c->wkl[l][imol];
        c->wbrodl[l] = c->coldry[l] * (1. - summol);
        for (int imol = 1; imol <= 7; imol++) c->wkl[l][imol] = c->coldry[l] * c->wkl[l][imol];
        amttl = amttl + c->coldry[l] + c->wkl[l][1];
        wvttl = wvttl + c->wkl[l][1];
        for (int ix = 1; ix <= 4; ix++) c->wx[l][ix] = c->coldry[l] * c->wx[l][ix] * 1.e-20;
    }
    double wvsh = (amw * wvttl) / (amd * amttl);
    c->pwvcm = wvsh * (1.e3 * c->pz[0]) / (1.e2 * grav);
    for (int n = 1; n <= NBND; n++) c->semiss[n] = in->emis[(iplon - 1) + (size_t)in->ncol * (n - 1)];
    if (iaer >= 1)
        for (int l = 1; l <= nl; l++)
            for (int ib = 1; ib <= NBND; ib++)
                c->taua[l][ib] = in->tauaer[(iplon - 1) + (size_t)in->ncol * ((l - 1) + (size_t)nl * (ib - 1))];
    if (icld >= 1) {
        c->inflag = in->inflglw; c->iceflag = in->iceflglw; c->liqflag = in->liqflglw;
        if (!mcica) {
            for (int l = 1; l <= nl; l++) {
                c->cldfrac[l] = G2(in->cldfr, iplon, l);
                c->ciwp[l] = G2(in->cicewp, iplon, l);
                c->clwp[l] = G2(in->cliqwp, iplon, l);
                c->rei[l] = G2(in->reice, iplon, l);
                c->rel[l] = G2(in->reliq, iplon, l);
                for (int n = 1; n <= NBND; n++)
                    c->tauc[l][n] = in->taucld[(n - 1) + (size_t)NBND * ((iplon - 1) + (size_t)in->ncol * (l - 1))];
            }
        } else {
            for (int l = 1; l <= nl; l++) {
                for (int ig = 1; ig <= NGPT; ig++) {
                    size_t k = (ig - 1) + (size_t)NGPT * ((iplon - 1) + (size_t)in->ncol * (l - 1));
                    c->cldfmc[l][ig] = in->cldfmcl[k];
                    c->taucmc[l][ig] = in->taucmcl[k];
                    c->ciwpmc[l][ig] = in->ciwpmcl[k];
                    c->clwpmc[l][ig] = in->clwpmcl[k];
                }
                c->rei[l] = G2(in->reicmcl, iplon, l);
                c->rel[l] = G2(in->relqmcl, iplon, l);
            }
        }
    }
}

static void combine_taut(col_t *c)      /* src/rrtmg_lw_rad.f90:542-554 (iaer = 10) */
{
    for (int k = 1; k <= c->nlayers; k++)
        for (int ig = 1; ig <= NGPT; ig++) c->taut[k][ig] = c->taug[k][ig] + c->taua[k][ngb_[ig - 1]];
}

static int ensure_col(void)
{
    if (!initialised) { strcpy(errmsg, "orc_init not called"); return -1; }
    if (!C) C = (col_t *)calloc(1, sizeof(col_t));
    return C ? 0 : -1;
}

/* ---------------------------------------------------------------------------------------------
 * rrtmg_lw, non-McICA : src/rrtmg_lw_rad.nomcica.f90:99-588
 * ------------------------------------------------------------------------------------------- */
int orc_rrtmg_lw_nomcica(int ncol, int nlay, int *icld, int idrv,
                         const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
                         const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr,
                         const double *n2ovmr, const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr,
                         const double *cfc22vmr, const double *ccl4vmr, const double *emis,
                         int inflglw, int iceflglw, int liqflglw,
                         const double *cldfr, const double *taucld, const double *cicewp, const double *cliqwp,
                         const double *reice, const double *reliq, const double *tauaer,
                         double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
                         double *duflx_dt, double *duflxc_dt)
{
    if (ensure_col()) return -1;
    if (nlay > MXLAY) { strcpy(errmsg, "nlay > mxlay"); return -1; }
    gcm_in_t in = {ncol, nlay, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr,
                   cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw,
                   cldfr, taucld, cicewp, cliqwp, reice, reliq, tauaer, 0, 0, 0, 0, 0, 0};
    const int istart = 1, iend = 16, iout = 0, iaer = 10;
    if (*icld < 0 || *icld > 3) *icld = 2;       /* :456 */
    col_t *c = C;
    /* inflag/iceflag/liqflag are only assigned when icld >= 1 (:893-896); with icld = 0 all cloud arrays
       are zero so cldprop's flags are never consulted. */
    c->inflag = c->iceflag = c->liqflag = 0;
    for (int iplon = 1; iplon <= ncol; iplon++) {
        inatm(c, &in, iplon, *icld, iaer, 0);
        if (cldprop(c)) return 1;
        setcoef(c, istart, idrv);
        taumol(c);
        combine_taut(c);
        if (*icld == 1) rtrn_generic(c, istart, iend, iout, idrv, 0);
        else rtrnmr(c, istart, iend, iout, idrv);
        for (int k = 0; k <= nlay; k++) {
            size_t o = (size_t)(iplon - 1) + (size_t)ncol * k;
            uflx[o] = c->totuflux[k]; dflx[o] = c->totdflux[k];
            uflxc[o] = c->totuclfl[k]; dflxc[o] = c->totdclfl[k];
        }
        for (int k = 0; k <= nlay - 1; k++) {
            size_t o = (size_t)(iplon - 1) + (size_t)ncol * k;
            hr[o] = c->htr[k]; hrc[o] = c->htrc[k];
        }
        if (idrv == 1)
            for (int k = 0; k <= nlay; k++) {
                size_t o = (size_t)(iplon - 1) + (size_t)ncol * k;
                duflx_dt[o] = c->dtotuflux_dt[k]; duflxc_dt[o] = c->dtotuclfl_dt[k];
            }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * rrtmg_lw, McICA : src/rrtmg_lw_rad.f90:99-595
 * ------------------------------------------------------------------------------------------- */
int orc_rrtmg_lw_mcica(int ncol, int nlay, int *icld, int idrv,
                       const double *play, const double *plev, const double *tlay, const double *tlev, const double *tsfc,
                       const double *h2ovmr, const double *o3vmr, const double *co2vmr, const double *ch4vmr,
                       const double *n2ovmr, const double *o2vmr, const double *cfc11vmr, const double *cfc12vmr,
                       const double *cfc22vmr, const double *ccl4vmr, const double *emis,
                       int inflglw, int iceflglw, int liqflglw,
                       const double *cldfmcl, const double *taucmcl, const double *ciwpmcl, const double *clwpmcl,
                       const double *reicmcl, const double *relqmcl, const double *tauaer,
                       double *uflx, double *dflx, double *hr, double *uflxc, double *dflxc, double *hrc,
                       double *duflx_dt, double *duflxc_dt)
{
    if (ensure_col()) return -1;
    if (nlay > MXLAY) { strcpy(errmsg, "nlay > mxlay"); return -1; }
    gcm_in_t in = {ncol, nlay, play, plev, tlay, tlev, tsfc, h2ovmr, o3vmr, co2vmr, ch4vmr, n2ovmr, o2vmr,
                   cfc11vmr, cfc12vmr, cfc22vmr, ccl4vmr, emis, inflglw, iceflglw, liqflglw,
                   0, 0, 0, 0, 0, 0, tauaer, cldfmcl, taucmcl, ciwpmcl, clwpmcl, reicmcl, relqmcl};
    const int istart = 1, iend = 16, iout = 0, iaer = 10;
    if (*icld < 0 || *icld > 3) *icld = 2;       /* src/rrtmg_lw_rad.f90:470 */
    col_t *c = C;
    c->inflag = c->iceflag = c->liqflag = 0;
    for (int iplon = 1; iplon <= ncol; iplon++) {
        inatm(c, &in, iplon, *icld, iaer, 1);
        if (cldprmc(c)) return 1;
        setcoef(c, istart, idrv);
        taumol(c);
        combine_taut(c);
        rtrn_generic(c, istart, iend, iout, idrv, 1);
        for (int k = 0; k <= nlay; k++) {
            size_t o = (size_t)(iplon - 1) + (size_t)ncol * k;
            uflx[o] = c->totuflux[k]; dflx[o] = c->totdflux[k];
            uflxc[o] = c->totuclfl[k]; dflxc[o] = c->totdclfl[k];
        }
        for (int k = 0; k <= nlay - 1; k++) {
            size_t o = (size_t)(iplon - 1) + (size_t)ncol * k;
            hr[o] = c->htr[k]; hrc[o] = c->htrc[k];
        }
        if (idrv == 1)
            for (int k = 0; k <= nlay; k++) {
                size_t o = (size_t)(iplon - 1) + (size_t)ncol * k;
                duflx_dt[o] = c->dtotuflux_dt[k]; duflxc_dt[o] = c->dtotuclfl_dt[k];
            }
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * Prepared-column entry = the physics sequence of the column driver (src/rrtmg_lw.1col.f90:497-580),
 * same argument list as oracle/ref_harness.f90:ref_column.  wkl7 is (7,nlayers), wx4 (4,nlayers),
 * tauc (16,nlayers), taua (nlayers,16) in Fortran order; outputs (0:nlayers); taug/fracs (nlayers,140).
 * ------------------------------------------------------------------------------------------- */
int orc_column(int nlayers, int istart, int iend, int iout, int icld, int idrv,
               const double *pavel, const double *tavel, const double *pz, const double *tz, double tbound,
               const double *semiss, const double *coldry, const double *wkl7, const double *wbrodl, const double *wx4,
               double pwvcm, int inflag, int iceflag, int liqflag, const double *cldfrac, const double *tauc,
               const double *ciwp, const double *clwp, const double *rei, const double *rel, const double *taua,
               double *totuflux, double *totdflux, double *fnet, double *htr,
               double *totuclfl, double *totdclfl, double *fnetc, double *htrc,
               double *dtotuflux_dt, double *dtotuclfl_dt, double *taug_out, double *fracs_out, int *ncbands_out)
{
    if (ensure_col()) return -1;
    col_t *c = C;
    int nl = nlayers;
    c->nlayers = nl;
    c->tbound = tbound;
    c->pwvcm = pwvcm;
    c->inflag = inflag; c->iceflag = iceflag; c->liqflag = liqflag;
    for (int b = 1; b <= NBND; b++) c->semiss[b] = semiss[b - 1];
    for (int l = 0; l <= nl; l++) { c->pz[l] = pz[l]; c->tz[l] = tz[l]; }
    for (int l = 1; l <= nl; l++) {
        c->pavel[l] = pavel[l - 1]; c->tavel[l] = tavel[l - 1];
        c->coldry[l] = coldry[l - 1]; c->wbrodl[l] = wbrodl[l - 1];
        for (int m = 1; m <= 7; m++) c->wkl[l][m] = wkl7[(m - 1) + 7 * (l - 1)];
        for (int m = 1; m <= 4; m++) c->wx[l][m] = wx4[(m - 1) + 4 * (l - 1)];
        c->cldfrac[l] = cldfrac[l - 1]; c->ciwp[l] = ciwp[l - 1]; c->clwp[l] = clwp[l - 1];
        c->rei[l] = rei[l - 1]; c->rel[l] = rel[l - 1];
        for (int b = 1; b <= NBND; b++) {
            c->tauc[l][b] = tauc[(b - 1) + NBND * (l - 1)];
            c->taua[l][b] = taua[(l - 1) + nl * (b - 1)];
        }
    }
    for (int b = 1; b <= NBND; b++) c->dplankbnd_dt[b] = 0.0;
    if (cldprop(c)) return 1;
    setcoef(c, istart, idrv);
    taumol(c);
    combine_taut(c);
    if (icld == 1) rtrn_generic(c, istart, iend, iout, idrv, 0);
    else rtrnmr(c, istart, iend, iout, idrv);
    for (int k = 0; k <= nl; k++) {
        totuflux[k] = c->totuflux[k]; totdflux[k] = c->totdflux[k]; fnet[k] = c->fnet[k]; htr[k] = c->htr[k];
        totuclfl[k] = c->totuclfl[k]; totdclfl[k] = c->totdclfl[k]; fnetc[k] = c->fnetc[k]; htrc[k] = c->htrc[k];
        dtotuflux_dt[k] = idrv == 1 ? c->dtotuflux_dt[k] : 0.0;
        dtotuclfl_dt[k] = idrv == 1 ? c->dtotuclfl_dt[k] : 0.0;
    }
    for (int l = 1; l <= nl; l++)
        for (int ig = 1; ig <= NGPT; ig++) {
            taug_out[(l - 1) + nl * (ig - 1)] = c->taug[l][ig];
            fracs_out[(l - 1) + nl * (ig - 1)] = c->fracs[l][ig];
        }
    *ncbands_out = c->ncbands;
    return 0;
}

/* McICA flavour of the prepared-column entry: cldprmc -> setcoef -> taumol -> rtrnmc, the per-sample physics sequence of the
 * column driver with imca = 1 (src/rrtmg_lw.1col.f90:497-580).  cldfmc, taucmc, ciwpmc, clwpmc are (140,nlayers). */
int orc_column_mc(int nlayers, int istart, int iend, int iout, int icld, int idrv,
                  const double *pavel, const double *tavel, const double *pz, const double *tz, double tbound,
                  const double *semiss, const double *coldry, const double *wkl7, const double *wbrodl, const double *wx4,
                  double pwvcm, int inflag, int iceflag, int liqflag, const double *cldfmc, const double *taucmc,
                  const double *ciwpmc, const double *clwpmc, const double *reicmc, const double *relqmc, const double *taua,
                  double *totuflux, double *totdflux, double *fnet, double *htr,
                  double *totuclfl, double *totdclfl, double *fnetc, double *htrc,
                  double *dtotuflux_dt, double *dtotuclfl_dt, double *taug_out, double *fracs_out, int *ncbands_out)
{
    (void)icld;
    if (ensure_col()) return -1;
    col_t *c = C;
    int nl = nlayers;
    c->nlayers = nl;
    c->tbound = tbound;
    c->pwvcm = pwvcm;
    c->inflag = inflag; c->iceflag = iceflag; c->liqflag = liqflag;
    for (int b = 1; b <= NBND; b++) c->semiss[b] = semiss[b - 1];
    for (int l = 0; l <= nl; l++) { c->pz[l] = pz[l]; c->tz[l] = tz[l]; }
    for (int l = 1; l <= nl; l++) {
        c->pavel[l] = pavel[l - 1]; c->tavel[l] = tavel[l - 1];
        c->coldry[l] = coldry[l - 1]; c->wbrodl[l] = wbrodl[l - 1];
        for (int m = 1; m <= 7; m++) c->wkl[l][m] = wkl7[(m - 1) + 7 * (l - 1)];
        for (int m = 1; m <= 4; m++) c->wx[l][m] = wx4[(m - 1) + 4 * (l - 1)];
        c->rei[l] = reicmc[l - 1]; c->rel[l] = relqmc[l - 1];
        for (int ig = 1; ig <= NGPT; ig++) {
            size_t o = (size_t)(ig - 1) + (size_t)NGPT * (l - 1);
            c->cldfmc[l][ig] = cldfmc[o]; c->taucmc[l][ig] = taucmc[o];
            c->ciwpmc[l][ig] = ciwpmc[o]; c->clwpmc[l][ig] = clwpmc[o];
        }
        for (int b = 1; b <= NBND; b++) c->taua[l][b] = taua[(l - 1) + nl * (b - 1)];
    }
    for (int b = 1; b <= NBND; b++) c->dplankbnd_dt[b] = 0.0;
    if (cldprmc(c)) return 1;
    setcoef(c, istart, idrv);
    taumol(c);
    combine_taut(c);
    rtrn_generic(c, istart, iend, iout, idrv, 1);
    for (int k = 0; k <= nl; k++) {
        totuflux[k] = c->totuflux[k]; totdflux[k] = c->totdflux[k]; fnet[k] = c->fnet[k]; htr[k] = c->htr[k];
        totuclfl[k] = c->totuclfl[k]; totdclfl[k] = c->totdclfl[k]; fnetc[k] = c->fnetc[k]; htrc[k] = c->htrc[k];
        dtotuflux_dt[k] = idrv == 1 ? c->dtotuflux_dt[k] : 0.0;
        dtotuclfl_dt[k] = idrv == 1 ? c->dtotuclfl_dt[k] : 0.0;
    }
    for (int l = 1; l <= nl; l++)
        for (int ig = 1; ig <= NGPT; ig++) {
            taug_out[(l - 1) + nl * (ig - 1)] = c->taug[l][ig];
            fracs_out[(l - 1) + nl * (ig - 1)] = c->fracs[l][ig];
        }
    *ncbands_out = c->ncbands;
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * McICA sub-column generator.  The GCM routine (src/mcica_subcol_gen_lw.f90:183-703) does not compile as
 * shipped; its compilable statement is the one-column generator
 * src/mcica_subcol_gen_lw.1col.f90:171-710, whose per-column semantics are restated here behind the GCM
 * argument list:  irng = 0 (kissvec) seeds every column from its own pressures, so columns are independent;
 * irng = 1 (Mersenne Twister) draws from ONE stream in (sub-column, column, layer) order as the GCM routine
 * does (src/mcica_subcol_gen_lw.f90:476-490, :523-530), which for ncol = 1 is the column driver's order.
 * ------------------------------------------------------------------------------------------- */
/* MT19937 on 32-bit words: src/mcica_random_numbers.f90:77-306 (signed arithmetic there, same bits) */
typedef struct { int cur; uint32_t st[624]; } mt_t;

static void mt_init(mt_t *t, int32_t seed)       /* initialize_scalar :157-169 */
{
    t->st[0] = (uint32_t)seed;
    for (int i = 1; i < 624; i++) t->st[i] = 1812433253u * (t->st[i - 1] ^ (t->st[i - 1] >> 30)) + (uint32_t)i;
    t->cur = 624;
}

static uint32_t mt_twist(uint32_t u, uint32_t v)  /* mixbits/twist :112-131 */
{
    uint32_t mix = (u & 0x80000000u) | (v & 0x7fffffffu);
    return (mix >> 1) ^ ((v & 1u) ? 0x9908b0dfu : 0u);
}

static void mt_next(mt_t *t)                      /* nextState :133-147 */
{
    int k;
    for (k = 0; k <= 624 - 397 - 1; k++) t->st[k] = t->st[k + 397] ^ mt_twist(t->st[k], t->st[k + 1]);
    for (k = 624 - 397; k <= 624 - 2; k++) t->st[k] = t->st[k + 397 - 624] ^ mt_twist(t->st[k], t->st[k + 1]);
    t->st[623] = t->st[396] ^ mt_twist(t->st[623], t->st[0]);
    t->cur = 0;
}

static double mt_real(mt_t *t)                    /* getRandomInt/temper/getRandomReal :149-155,:262-295 */
{
    if (t->cur >= 624) mt_next(t);
    uint32_t y = t->st[t->cur++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    int32_t li = (int32_t)y;
    if (li < 0) return ((double)li + 4294967296.0) / (4294967296.0 - 1.0);
    return ((double)li) / (4294967296.0 - 1.0);
}

/* kissvec: src/mcica_subcol_gen_lw.1col.f90:677-710 (32-bit wrap-around, logical shifts) */
static double kissvec(int32_t *s1, int32_t *s2, int32_t *s3, int32_t *s4)
{
    uint32_t a = (uint32_t)*s1, b = (uint32_t)*s2, c = (uint32_t)*s3, d = (uint32_t)*s4;
    a = 69069u * a + 1327217885u;
    b ^= b << 13; b ^= b >> 17; b ^= b << 5;
    c = 18000u * (c & 65535u) + (c >> 16);
    d = 30903u * (d & 65535u) + (d >> 16);
    uint32_t kiss = a + b + (c << 16) + d;
    *s1 = (int32_t)a; *s2 = (int32_t)b; *s3 = (int32_t)c; *s4 = (int32_t)d;
    return (double)(int32_t)kiss * 2.328306e-10 + 0.5;
}

/* get_alpha: src/mcica_subcol_gen_lw.f90:68-180 (.1col :67-168).  dz, cldfrac, alpha are (ncol,nlay); lat (ncol). */
void orc_get_alpha(int ncol, int nlay, int icld, int idcor, double decorr_con, const double *dz, const double *lat,
                   int juldat, const double *cldfrac, double *alpha)
{
    const double am1 = 1.4315, am2 = 2.1219, am4 = -25.584, amr = 7.0;
    for (int i = 0; i < ncol; i++) {
        double decorr_inv = 1.0, decorr_len = 0.0;
        if (icld == 4 || icld == 5) {
            if (idcor == 1) {
                double am3;
                if (juldat > 181) am3 = -4. * amr / 365. * (juldat - 272);
                else am3 = 4. * amr / 365. * (juldat - 91);
                double decorr_lat = am1 + am2 * exp(-((lat[i] - am3) * (lat[i] - am3)) / (am4 * am4));
                decorr_len = decorr_lat * 1.e3;
            } else {
                decorr_len = decorr_con;
            }
            if (decorr_len >= 0.0) decorr_inv = 1.0 / decorr_len;
        }
        if (icld == 4 || icld == 5) {
            alpha[i] = 0.0;
            for (int k = 2; k <= nlay; k++) {
                size_t o = (size_t)i + (size_t)ncol * (k - 1), om = (size_t)i + (size_t)ncol * (k - 2);
                alpha[o] = exp(-(0.5 * (dz[o] + dz[om])) * decorr_inv);
                if (icld == 5 && cldfrac[o] == 0.0 && cldfrac[om] > 0.0) alpha[o] = 0.0;
            }
        }
    }
}

/* mcica_subcol_lw with the GCM argument list (src/mcica_subcol_gen_lw.f90:183-291).  play, cldfrac, ciwp, clwp, rei, rel,
 * alpha: (ncol,nlay); tauc: (16,ncol,nlay); outputs cldfmcl, ciwpmcl, clwpmcl, taucmcl: (140,ncol,nlay); reicmcl, relqmcl (ncol,nlay). */
int orc_mcica_subcol(int ncol, int nlay, int icld, int permuteseed, int *irng, const double *play, const double *cldfrac,
                     const double *ciwp, const double *clwp, const double *rei, const double *rel, const double *tauc,
                     const double *alpha, double *cldfmcl, double *ciwpmcl, double *clwpmcl, double *reicmcl, double *relqmcl,
                     double *taucmcl)
{
    if (!initialised) { strcpy(errmsg, "orc_init not called"); return -1; }
    if (icld == 0) return 0;
    if (icld < 0 || icld > 5) { strcpy(errmsg, "MCICA_SUBCOL_LW: INVALID ICLD"); return 1; }
    if (*irng != 0) *irng = 1;
    const double cldmin = 1.0e-20;
    const int nsub = NGPT;
    for (size_t k = 0; k < (size_t)ncol * nlay; k++) { reicmcl[k] = rei[k]; relqmcl[k] = rel[k]; }
    double *cdf = (double *)malloc(sizeof(double) * nsub * (size_t)ncol * nlay);
    double *cdf2 = (double *)malloc(sizeof(double) * nsub * (size_t)ncol * nlay);
#define CDF(a, s, i, l) (a)[(s) + (size_t)nsub * ((i) + (size_t)ncol * (l))]
#define P2(a, i, l) (a)[(i) + (size_t)ncol * (l)]
    mt_t mt;
    if (*irng == 1) mt_init(&mt, (int32_t)permuteseed);
    const int two = (icld == 4 || icld == 5);
    if (*irng == 0) {
        for (int i = 0; i < ncol; i++) {
            if (nlay < 4) { free(cdf); free(cdf2); strcpy(errmsg, "MCICA_SUBCOL: fewer than four layers"); return 1; }
            if (P2(play, i, 0) * 1.e2 < P2(play, i, 1) * 1.e2) {
                free(cdf); free(cdf2);
                strcpy(errmsg, "MCICA_SUBCOL: KISSVEC SEED GENERATOR REQUIRES PMID FROM BOTTOM FOUR LAYERS.");
                return 1;
            }
            int32_t s[4];
            for (int q = 0; q < 4; q++) {
                double pm = P2(play, i, q) * 1.e2;
                s[q] = (int32_t)((pm - (double)(int)pm) * 1000000000);
            }
            for (int q = 0; q < permuteseed; q++) (void)kissvec(&s[0], &s[1], &s[2], &s[3]);
            for (int isub = 0; isub < nsub; isub++) {
                if (icld == 3) {
                    double r = kissvec(&s[0], &s[1], &s[2], &s[3]);
                    for (int l = 0; l < nlay; l++) CDF(cdf, isub, i, l) = r;
                } else {
                    for (int l = 0; l < nlay; l++) {
                        CDF(cdf, isub, i, l) = kissvec(&s[0], &s[1], &s[2], &s[3]);
                        if (two) CDF(cdf2, isub, i, l) = kissvec(&s[0], &s[1], &s[2], &s[3]);
                    }
                }
            }
        }
    } else {
        for (int isub = 0; isub < nsub; isub++)
            for (int i = 0; i < ncol; i++) {
                if (icld == 3) {
                    double r = mt_real(&mt);
                    for (int l = 0; l < nlay; l++) CDF(cdf, isub, i, l) = r;
                } else {
                    for (int l = 0; l < nlay; l++) {
                        CDF(cdf, isub, i, l) = mt_real(&mt);
                        if (two) CDF(cdf2, isub, i, l) = mt_real(&mt);
                    }
                }
            }
    }
    for (int i = 0; i < ncol; i++) {
        for (int l = 1; l < nlay; l++) {
            double cfl = P2(cldfrac, i, l - 1);
            if (cfl < cldmin) cfl = 0.0;
            for (int isub = 0; isub < nsub; isub++) {
                if (icld == 2) {                       /* maximum-random, .1col :440-448 */
                    if (CDF(cdf, isub, i, l - 1) > 1. - cfl) CDF(cdf, isub, i, l) = CDF(cdf, isub, i, l - 1);
                    else CDF(cdf, isub, i, l) = CDF(cdf, isub, i, l) * (1. - cfl);
                } else if (two) {                      /* exponential(-random), .1col :492-496,:521-525 */
                    if (CDF(cdf2, isub, i, l) < P2(alpha, i, l)) CDF(cdf, isub, i, l) = CDF(cdf, isub, i, l - 1);
                }
            }
        }
        for (int l = 0; l < nlay; l++) {
            double cf = P2(cldfrac, i, l);
            if (cf < cldmin) cf = 0.0;
            for (int isub = 0; isub < nsub; isub++) {
                size_t o = isub + (size_t)nsub * (i + (size_t)ncol * l);
                if (CDF(cdf, isub, i, l) >= 1. - cf) {
                    cldfmcl[o] = 1.0;
                    clwpmcl[o] = P2(clwp, i, l);
                    ciwpmcl[o] = P2(ciwp, i, l);
                    taucmcl[o] = tauc[(ngb_[isub] - 1) + (size_t)NBND * (i + (size_t)ncol * l)];
                } else {
                    cldfmcl[o] = 0.0; clwpmcl[o] = 0.0; ciwpmcl[o] = 0.0; taucmcl[o] = 0.0;
                }
            }
        }
    }
#undef CDF
#undef P2
    free(cdf);
    free(cdf2);
    return 0;
}
