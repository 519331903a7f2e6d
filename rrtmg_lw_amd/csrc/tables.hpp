// Host-side one-time setup for the HIP solver: the work of rrtmg_lw_ini (reference:
// src/rrtmg_lw_init.f90:47-194) re-done for the device layout.
//
//  * reads lw_static.bin (Planck integrals, reference atmosphere, cloud coefficients, g-point maps) and
//    the k-data blob (original 16-g absorption coefficients, rrtmg_lw_amd/kspec.py),
//  * combines 256 -> 140 g-points with the rwgt weights (reference: src/rrtmg_lw_init.f90:149-173 and the
//    cmbgb1..16 routines :385-2034; Planck fractions are plain sums, e.g. :693-712),
//  * stores every reduced table g-point-FASTEST ([table row][g]) - the reference keeps g slowest
//    (absa(index,ig), modules/rrlw_kg03.f90:65), which is the wrong way round for threads that walk
//    consecutive g-points of one row with vector loads,
//  * builds the transmittance / tau-transition look-up tables (:125-142), exp and tfn interleaved so that
//    one 16-byte gather serves both.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace rrlw {

constexpr int NBND = 16;
// -DRRLW_G256 builds the 256-g-point model the reference keeps as a commented-out alternative (modules/parrrtm.f90:40-41,77-110;
// src/rrtmg_lw_init.f90:313-314: "the full 256 g-point set can be restored with ngptlw=256, ngc=16*16, ngn=256*1., etc."): every band
// keeps its 16 original g-points and the combination step copies (McICA: 256 sub-columns, masks of 8 words).
#ifdef RRLW_G256
constexpr int NGPT = 256;
#else
constexpr int NGPT = 140;
#endif
constexpr int MASK_WORDS = (NGPT + 31) / 32;        // 32-bit words of a (column, layer)'s sub-column cloud mask: 5 (140 bits) or 8
constexpr int NTBL = 10000;

struct BlobEntry {
    int dtype, ndim;
    uint32_t dims[6];
    const unsigned char *data;
    size_t nbytes;
    size_t count() const { size_t n = 1; for (int i = 0; i < ndim; i++) n *= dims[i]; return n; }
};

class Blob {
  public:
    bool open(const std::string &path, std::string &err)
    {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) { err = "cannot open " + path; return false; }
        std::fseek(f, 0, SEEK_END);
        long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        buf_.resize((size_t)sz);
        bool ok = std::fread(buf_.data(), 1, (size_t)sz, f) == (size_t)sz;
        std::fclose(f);
        if (!ok || sz < 16 || std::memcmp(buf_.data(), "RRLWBLOB", 8) != 0) { err = path + ": not an RRLWBLOB file"; return false; }
        uint32_t ver, n;
        std::memcpy(&ver, &buf_[8], 4);
        std::memcpy(&n, &buf_[12], 4);
        if (ver != 1) { err = path + ": unsupported blob version"; return false; }
        if (16 + 96 * (uint64_t)n > (uint64_t)sz) { err = path + ": entry table runs past the end of the file"; return false; }
        for (uint32_t i = 0; i < n; i++) {
            const unsigned char *p = &buf_[16 + 96 * (size_t)i];
            char name[49];
            std::memcpy(name, p, 48);
            name[48] = 0;
            BlobEntry e;
            uint32_t dt, nd;
            uint64_t off, nb;
            std::memcpy(&dt, p + 48, 4);
            std::memcpy(&nd, p + 52, 4);
            std::memcpy(e.dims, p + 56, 24);
            std::memcpy(&off, p + 80, 8);
            std::memcpy(&nb, p + 88, 8);
            if (off > (uint64_t)sz || nb > (uint64_t)sz - off || nd > 6) { err = path + ": corrupt entry table"; return false; }   // (no wrap-around of off + nb)
            e.dtype = (int)dt; e.ndim = (int)nd; e.data = &buf_[off]; e.nbytes = (size_t)nb;
            entries_[name] = e;
        }
        return true;
    }
    const BlobEntry *find(const std::string &name) const
    {
        auto it = entries_.find(name);
        return it == entries_.end() ? nullptr : &it->second;
    }
    bool get_f64(const std::string &name, size_t count, std::vector<double> &out, std::string &err) const
    {
        const BlobEntry *e = find(name);
        if (!e || e->dtype != 0 || e->nbytes != count * 8) { err = "table '" + name + "' missing or wrong size"; return false; }
        out.resize(count);
        std::memcpy(out.data(), e->data, e->nbytes);
        return true;
    }
    bool get_i32(const std::string &name, size_t count, std::vector<int> &out, std::string &err) const
    {
        const BlobEntry *e = find(name);
        if (!e || e->dtype != 1 || e->nbytes != count * 4) { err = "table '" + name + "' missing or wrong size"; return false; }
        out.resize(count);
        std::memcpy(out.data(), e->data, e->nbytes);
        return true;
    }

  private:
    std::vector<unsigned char> buf_;
    std::map<std::string, BlobEntry> entries_;
};

// Offsets (in doubles) of one band's tables inside the packed k-table buffer; -1 = band has no such table.
// Rows are contiguous runs of `ng` doubles (g fastest).
struct BandLayout {
    int ng;          // g-points in this band (ngc)
    int gstart;      // 0-based index of the band's first g-point among the 140
    int absa, absb;  // [65*nspa][ng], [235*nspb][ng]
    int selfref;     // [10][ng]
    int forref;      // [4][ng]
    int fracrefa;    // [9][ng] (binary-key bands) or [1][ng]
    int fracrefb;    // [5][ng] or [1][ng]
    int minor_lo[3]; // lower-atmosphere minor-gas tables: [19][ng] or [19*9][ng] (row = (indm-1)*9 + jm-1)
    int minor_up[2]; // upper-atmosphere: [19][ng] or [19*5][ng]
    int vec[2];      // CFC / CCl4 vectors [ng]
};

// names of the reduced tables that fill the minor_lo / minor_up / vec slots of each band
struct BandSlots { const char *lo[3]; const char *up[2]; const char *vec[2]; };
static const BandSlots kSlots[NBND] = {
    /* 1*/ {{"ka_mn2", 0, 0}, {"kb_mn2", 0}, {0, 0}},
    /* 2*/ {{0, 0, 0}, {0, 0}, {0, 0}},
    /* 3*/ {{"ka_mn2o", 0, 0}, {"kb_mn2o", 0}, {0, 0}},
    /* 4*/ {{0, 0, 0}, {0, 0}, {0, 0}},
    /* 5*/ {{"ka_mo3", 0, 0}, {0, 0}, {"ccl4", 0}},
    /* 6*/ {{"ka_mco2", 0, 0}, {0, 0}, {"cfc11adj", "cfc12"}},
    /* 7*/ {{"ka_mco2", 0, 0}, {"kb_mco2", 0}, {0, 0}},
    /* 8*/ {{"ka_mco2", "ka_mo3", "ka_mn2o"}, {"kb_mco2", "kb_mn2o"}, {"cfc12", "cfc22adj"}},
    /* 9*/ {{"ka_mn2o", 0, 0}, {"kb_mn2o", 0}, {0, 0}},
    /*10*/ {{0, 0, 0}, {0, 0}, {0, 0}},
    /*11*/ {{"ka_mo2", 0, 0}, {"kb_mo2", 0}, {0, 0}},
    /*12*/ {{0, 0, 0}, {0, 0}, {0, 0}},
    /*13*/ {{"ka_mco2", "ka_mco", 0}, {"kb_mo3", 0}, {0, 0}},
    /*14*/ {{0, 0, 0}, {0, 0}, {0, 0}},
    /*15*/ {{"ka_mn2", 0, 0}, {0, 0}, {0, 0}},
    /*16*/ {{0, 0, 0}, {0, 0}, {0, 0}},
};

// Offsets (in doubles) inside the "static" device buffer
struct StaticLayout {
    int preflog, tref;            // [59]
    int totplnk, totplnkderiv;    // [16][181]  (band-major: row = band)
    int totplk16, totplk16deriv;  // [181]
    int rat;                      // [6][59] chi ratios: h2o/co2, h2o/o3, h2o/n2o, h2o/ch4, n2o/co2, o3/co2
    int chi;                      // [7][59] chi_mls, species-major
    int absice1;                  // [5][2]
    int absice2, absice3;         // [16][43], [16][46] band-major
    int absliq1;                  // [16][58]
    int lut;                      // [10001][2]  (exp_tbl, tfn_tbl)
    int tau_tbl;                  // [10001]
    int lutf;                     // [10002] float pairs {1 - exp_tbl, tfn_tbl} (8 bytes each; the copy k_sweep stages in LDS)
};

struct HostTables {
    std::vector<double> ktab;     // packed reduced k tables
    BandLayout band[NBND];
    std::vector<double> stat;     // static buffer
    StaticLayout sl;
    double absice0[2], abscld1, absliq0;
    double delwave[NBND];
    int nspa[NBND], nspb[NBND], ngc[NBND], ngs[NBND], ngb[NGPT];
    double heatfac, fluxfac, oneminus, bpade;
    double refrat[NBND][6];       // per-band reference ratios used by taumol (see kernels.hip)
    bool standin = false;
};

inline std::string reduced_name(const std::string &o)
{
    if (o.rfind("kao", 0) == 0 || o.rfind("kbo", 0) == 0) return std::string("k") + o[1] + o.substr(3);
    return o.substr(0, o.size() - 1);
}

inline bool build_tables(const std::string &static_path, const std::string &kdata_path, double cpdair,
                         HostTables &T, std::string &err)
{
    Blob sb, kb;
    if (!sb.open(static_path, err) || !kb.open(kdata_path, err)) return false;
    std::vector<double> pref, preflog, tref, chi, totplnk, totplk16, totplnkd, totplk16d;
    std::vector<double> absice0, absice1, absice2, absice3, absliq1, abscld1, absliq0, wt, delwave;
    std::vector<int> ngc, ngs, ngm, ngn, ngb, nspa, nspb;
    bool ok = sb.get_f64("preflog", 59, preflog, err) && sb.get_f64("tref", 59, tref, err) &&
              sb.get_f64("chi_mls", 7 * 59, chi, err) && sb.get_f64("totplnk", 181 * 16, totplnk, err) &&
              sb.get_f64("totplk16", 181, totplk16, err) && sb.get_f64("totplnkderiv", 181 * 16, totplnkd, err) &&
              sb.get_f64("totplk16deriv", 181, totplk16d, err) && sb.get_f64("absice0", 2, absice0, err) &&
              sb.get_f64("absice1", 10, absice1, err) && sb.get_f64("absice2", 43 * 16, absice2, err) &&
              sb.get_f64("absice3", 46 * 16, absice3, err) && sb.get_f64("absliq1", 58 * 16, absliq1, err) &&
              sb.get_f64("abscld1", 1, abscld1, err) && sb.get_f64("absliq0", 1, absliq0, err) &&
              sb.get_f64("wt", 16, wt, err) && sb.get_f64("delwave", 16, delwave, err) &&
              sb.get_i32("ngc", 16, ngc, err) && sb.get_i32("ngs", 16, ngs, err) && sb.get_i32("ngm", 256, ngm, err) &&
              sb.get_i32("ngn", 140, ngn, err) && sb.get_i32("ngb", 140, ngb, err) &&
              sb.get_i32("nspa", 16, nspa, err) && sb.get_i32("nspb", 16, nspb, err);
    if (!ok) return false;
#ifdef RRLW_G256
    ngn.assign(256, 1);
    ngb.resize(256);
    for (int b = 0; b < NBND; b++) { ngc[b] = 16; ngs[b] = 16 * (b + 1); }
    for (int i = 0; i < 256; i++) { ngm[i] = i % 16 + 1; ngb[i] = i / 16 + 1; }
#endif

    // constants: lwdatinit (src/rrtmg_lw_init.f90:243,265,298) and the rad driver (src/rrtmg_lw_rad.f90:451-453)
    T.heatfac = 9.8066 * 8.6400e4 / (cpdair * 1.e2);
    T.oneminus = 1.0 - 1.e-6;
    T.fluxfac = (2.0 * std::asin(1.0)) * 2.e4;
    T.bpade = 1.0 / 0.278;
    T.absice0[0] = absice0[0]; T.absice0[1] = absice0[1];
    T.abscld1 = abscld1[0]; T.absliq0 = absliq0[0];
    for (int b = 0; b < NBND; b++) {
        T.delwave[b] = delwave[b]; T.nspa[b] = nspa[b]; T.nspb[b] = nspb[b]; T.ngc[b] = ngc[b]; T.ngs[b] = ngs[b];
    }
    for (int g = 0; g < NGPT; g++) T.ngb[g] = ngb[g];
    const BlobEntry *meta = kb.find("meta.standin");
    T.standin = meta != nullptr;

    // ---- g-point groups and combination weights -------------------------------------------------
    // group(b, k) = original g-points [first, first+len) of band b that form reduced g-point k; weight of an
    // original point = wt / (sum of wt over its group)  (all ones when the band keeps its 16 points).
    struct Group { int first, len; };
    std::vector<std::vector<Group>> groups(NBND);
    std::vector<std::vector<double>> weight(NBND, std::vector<double>(16, 1.0));
    {
        int k140 = 0;
        for (int b = 0; b < NBND; b++) {
            int first = 0;
            for (int k = 0; k < ngc[b]; k++, k140++) {
                groups[b].push_back({first, ngn[k140]});
                first += ngn[k140];
            }
            if (first != 16) { err = "g-point map does not cover 16 points in band " + std::to_string(b + 1); return false; }
            if (ngc[b] < 16)
                for (auto &g : groups[b]) {
                    double s = 0.0;
                    for (int i = 0; i < g.len; i++) s = s + wt[g.first + i];
                    for (int i = 0; i < g.len; i++) weight[b][g.first + i] = wt[g.first + i] / s;
                }
        }
    }

    // ---- reduce + transpose every k-data array into the packed buffer -----------------------------
    std::map<std::string, int> where[NBND];
    T.ktab.clear();
    for (int b = 0; b < NBND; b++) {
        char pfx[8];
        std::snprintf(pfx, sizeof pfx, "b%02d.", b + 1);
        const int ng = ngc[b];
        // iterate the blob entries of this band in a fixed order: use the kspec order via known names
        static const char *names[] = {"fracrefao", "fracrefbo", "kao", "kbo", "selfrefo", "forrefo", "kao_mn2", "kbo_mn2",
                                      "kao_mn2o", "kbo_mn2o", "kao_mo3", "kao_mco2", "kbo_mco2", "kao_mo2", "kbo_mo2",
                                      "kao_mco", "kbo_mo3", "ccl4o", "cfc11adjo", "cfc12o", "cfc22adjo"};
        for (const char *nm : names) {
            const BlobEntry *e = kb.find(std::string(pfx) + nm);
            if (!e) continue;
            if (e->dtype != 0) { err = std::string("k-data entry not float64: ") + pfx + nm; return false; }
            const bool frac = std::strncmp(nm, "fracref", 7) == 0;
            const int gax = frac ? 0 : e->ndim - 1;
            if (e->dims[gax] != 16) { err = std::string("k-data entry without 16-g axis: ") + pfx + nm; return false; }
            const size_t rows = e->count() / 16;     // all non-g indices flattened in Fortran order
            const double *src = reinterpret_cast<const double *>(e->data);
            while (T.ktab.size() % 2) T.ktab.push_back(0.0);     // keep every table 16-byte aligned
            const int off = (int)T.ktab.size();
            T.ktab.resize(T.ktab.size() + rows * (size_t)ng);
            for (size_t r = 0; r < rows; r++)
                for (int k = 0; k < ng; k++) {
                    const Group &g = groups[b][k];
                    double s = 0.0;
                    for (int i = 0; i < g.len; i++) {
                        const int ip = g.first + i;
                        const double v = frac ? src[ip + 16 * r] : src[r + rows * ip];
                        s = frac ? s + v : s + v * weight[b][ip];
                    }
                    T.ktab[off + r * ng + k] = s;
                }
            where[b][reduced_name(nm)] = off;
        }
    }
    // stencil rows with zero weight may lie one row past the last table (kernels.hip, stencil6): keep them in bounds
    T.ktab.resize(T.ktab.size() + 64, 0.0);
    auto need = [&](int b, const char *nm, bool required, int &dst) -> bool {
        auto it = where[b].find(nm);
        if (it == where[b].end()) {
            dst = -1;
            if (required) { err = "k-data: band " + std::to_string(b + 1) + " lacks table " + nm; return false; }
            return true;
        }
        dst = it->second;
        return true;
    };
    for (int b = 0; b < NBND; b++) {
        BandLayout &L = T.band[b];
        L.ng = ngc[b];
        L.gstart = b == 0 ? 0 : ngs[b - 1];
        if (!need(b, "ka", true, L.absa) || !need(b, "kb", nspb[b] > 0, L.absb) || !need(b, "selfref", true, L.selfref) ||
            !need(b, "forref", true, L.forref) || !need(b, "fracrefa", true, L.fracrefa) ||
            !need(b, "fracrefb", false, L.fracrefb))
            return false;
        for (int i = 0; i < 3; i++) { L.minor_lo[i] = -1; if (kSlots[b].lo[i] && !need(b, kSlots[b].lo[i], true, L.minor_lo[i])) return false; }
        for (int i = 0; i < 2; i++) { L.minor_up[i] = -1; if (kSlots[b].up[i] && !need(b, kSlots[b].up[i], true, L.minor_up[i])) return false; }
        for (int i = 0; i < 2; i++) { L.vec[i] = -1; if (kSlots[b].vec[i] && !need(b, kSlots[b].vec[i], true, L.vec[i])) return false; }
    }

    // ---- static device buffer ---------------------------------------------------------------------
    auto &S = T.stat;
    S.clear();
    auto put = [&](const double *p, size_t n) { while (S.size() % 2) S.push_back(0.0); int o = (int)S.size(); S.insert(S.end(), p, p + n); return o; };
    T.sl.preflog = put(preflog.data(), 59);
    T.sl.tref = put(tref.data(), 59);
    T.sl.totplnk = put(totplnk.data(), 181 * 16);          // Fortran (181,16) == [band][181]
    T.sl.totplnkderiv = put(totplnkd.data(), 181 * 16);
    T.sl.totplk16 = put(totplk16.data(), 181);
    T.sl.totplk16deriv = put(totplk16d.data(), 181);
    auto CHI = [&](int i, int j) { return chi[(i - 1) + 7 * (j - 1)]; };      // chi_mls(i,j)
    {
        // setcoef's reference ratios (src/rrtmg_lw_setcoef.f90:338-351,392-396) tabulated over jp = 1..59
        std::vector<double> rat(6 * 59);
        const int num[6] = {1, 1, 1, 1, 4, 3}, den[6] = {2, 3, 4, 6, 2, 2};
        for (int k = 0; k < 6; k++)
            for (int j = 1; j <= 59; j++) rat[k * 59 + (j - 1)] = CHI(num[k], j) / CHI(den[k], j);
        T.sl.rat = put(rat.data(), rat.size());
        std::vector<double> chit(7 * 59);
        for (int i = 1; i <= 7; i++)
            for (int j = 1; j <= 59; j++) chit[(i - 1) * 59 + (j - 1)] = CHI(i, j);
        T.sl.chi = put(chit.data(), chit.size());
    }
    {
        std::vector<double> a1(10);
        for (int ib = 0; ib < 5; ib++) { a1[ib * 2] = absice1[0 + 2 * ib]; a1[ib * 2 + 1] = absice1[1 + 2 * ib]; }
        T.sl.absice1 = put(a1.data(), 10);
    }
    T.sl.absice2 = put(absice2.data(), 43 * 16);           // Fortran (43,16) == [band][43]
    T.sl.absice3 = put(absice3.data(), 46 * 16);
    T.sl.absliq1 = put(absliq1.data(), 58 * 16);
    {
        // look-up tables, src/rrtmg_lw_init.f90:125-142
        std::vector<double> lut(2 * (NTBL + 1)), tau(NTBL + 1);
        const double expeps = 1.e-20;
        tau[0] = 0.0; tau[NTBL] = 1.e10;
        lut[0] = 1.0; lut[2 * NTBL] = expeps;
        lut[1] = 0.0; lut[2 * NTBL + 1] = 1.0;
        for (int i = 1; i < NTBL; i++) {
            const double t = (double)i / (double)NTBL;
            tau[i] = T.bpade * t / (1.0 - t);
            double ex = std::exp(-tau[i]);
            if (ex <= expeps) ex = expeps;
            lut[2 * i] = ex;
            lut[2 * i + 1] = tau[i] < 0.06 ? tau[i] / 6.0 : 1.0 - 2.0 * ((1.0 / tau[i]) - (ex / (1. - ex)));
        }
        T.sl.lut = put(lut.data(), lut.size());
        T.sl.tau_tbl = put(tau.data(), tau.size());
        // float copy for k_sweep: the transmittance complement is formed in float64 and rounded once
        std::vector<double> lutf(NTBL + 2, 0.0);
        for (int i = 0; i <= NTBL; i++) {
            const float pr[2] = {(float)(1.0 - lut[2 * i]), (float)lut[2 * i + 1]};
            std::memcpy(&lutf[i], pr, 8);
        }
        T.sl.lutf = put(lutf.data(), lutf.size());
    }
    // per-band reference ratios of taumol (refrat_planck_a/b, refrat_m_a/b/..; src/rrtmg_lw_taumol.f90:504-513,
    // :812-815, :1080-1086, :1442-1445, :1839-1842, :2242, :2454-2460, :2770-2773, :2991)
    for (int b = 0; b < NBND; b++) for (int k = 0; k < 6; k++) T.refrat[b][k] = 0.0;
    // slots: 0 planck_a, 1 planck_b, 2 minor_a (first 2-D minor), 3 minor_b (upper 2-D minor), 4 second lower 2-D minor
    T.refrat[2][0] = CHI(1, 9) / CHI(2, 9);   T.refrat[2][1] = CHI(1, 13) / CHI(2, 13);
    T.refrat[2][2] = CHI(1, 3) / CHI(2, 3);   T.refrat[2][3] = CHI(1, 13) / CHI(2, 13);
    T.refrat[3][0] = CHI(1, 11) / CHI(2, 11); T.refrat[3][1] = CHI(3, 13) / CHI(2, 13);
    T.refrat[4][0] = CHI(1, 5) / CHI(2, 5);   T.refrat[4][1] = CHI(3, 43) / CHI(2, 43);
    T.refrat[4][2] = CHI(1, 7) / CHI(2, 7);
    T.refrat[6][0] = CHI(1, 3) / CHI(3, 3);   T.refrat[6][2] = CHI(1, 3) / CHI(3, 3);
    T.refrat[8][0] = CHI(1, 9) / CHI(6, 9);   T.refrat[8][2] = CHI(1, 3) / CHI(6, 3);
    T.refrat[11][0] = CHI(1, 10) / CHI(2, 10);
    T.refrat[12][0] = CHI(1, 5) / CHI(4, 5);  T.refrat[12][2] = CHI(1, 1) / CHI(4, 1);
    T.refrat[12][4] = CHI(1, 3) / CHI(4, 3);
    T.refrat[14][0] = CHI(4, 1) / CHI(2, 1);  T.refrat[14][2] = CHI(4, 1) / CHI(2, 1);
    T.refrat[15][0] = CHI(1, 6) / CHI(6, 6);
    return true;
}

}  // namespace rrlw
